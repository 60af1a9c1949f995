// transport kernel instantiations for the ray-march variant 1 (0 plain, 1 xyz-symmetric octant, 2 z-only slab)
#include "transport_kernel.hpp"
extern "C" void lart_launch_transport_var1(int stokes, int dust, int peel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const KArgs &A) {
    LART_LAUNCH_CASE(1, 0, 0, 0) LART_LAUNCH_CASE(1, 1, 0, 0) LART_LAUNCH_CASE(1, 0, 1, 0) LART_LAUNCH_CASE(1, 1, 1, 0)
}
