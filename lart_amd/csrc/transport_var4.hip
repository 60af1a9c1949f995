// transport kernel instantiations for the clump medium (variant 4: raytrace_clump.f90, clump_mod.f90)
#include "transport_kernel.hpp"
extern "C" void lart_launch_transport_var4(int stokes, int dust, int peel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const KArgs &A) {
    LART_LAUNCH_CASE(4, 0, 0, 0) LART_LAUNCH_CASE(4, 1, 0, 0) LART_LAUNCH_CASE(4, 0, 1, 0) LART_LAUNCH_CASE(4, 1, 1, 0)
    LART_LAUNCH_CASE(4, 0, 0, 1) LART_LAUNCH_CASE(4, 1, 0, 1) LART_LAUNCH_CASE(4, 0, 1, 1) LART_LAUNCH_CASE(4, 1, 1, 1)
}
