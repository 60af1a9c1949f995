// transport kernel instantiations for overlapping clumps (variant 7: raytrace_clump.f90:606-918, scattering_car.f90:898-937); single-line profiles
#include "transport_kernel.hpp"
extern "C" void lart_launch_transport_var7(int stokes, int dust, int peel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const KArgs &A) {
    LART_LAUNCH_CASE1(7, 0, 0, 0, 1) LART_LAUNCH_CASE1(7, 1, 0, 0, 1) LART_LAUNCH_CASE1(7, 0, 1, 0, 1) LART_LAUNCH_CASE1(7, 1, 1, 0, 1)
    LART_LAUNCH_CASE1(7, 0, 0, 1, 1) LART_LAUNCH_CASE1(7, 1, 0, 1, 1) LART_LAUNCH_CASE1(7, 0, 1, 1, 1) LART_LAUNCH_CASE1(7, 1, 1, 1, 1)
}
