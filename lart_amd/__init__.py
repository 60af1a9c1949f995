"""lart_amd: MI355X-native Monte Carlo resonance-line transfer -- the photon-packet hot path of LaRT behind a C-ABI.

    lart_amd/csrc   hand-written HIP (gfx950) kernels + the C-ABI of include/lart_gpu.h -> liblart_gpu.so
    lart_amd/capi   ctypes binding of that ABI (fails loudly when the library is missing)
    lart_amd/host   host-side mirror of the reference interface (namelist, grid_create, observers, normalisation)
    lart_amd/fhost/ the Fortran host (ISO_C_BINDING) with the same call sequence as the reference's main.f90
"""
__version__ = "0.1.0"
