// transport kernel instantiations for overlapping clumps, populations of at most 64 clumps (variant 9 = variant 7 with the flights that need the clump table executed
// by the wave, one lane per clump: ovl_coop in lart_device.hpp; raytrace_clump.f90:606-918, scattering_car.f90:898-937); single-line profiles
#include "transport_kernel.hpp"
extern "C" void lart_launch_transport_var9(int stokes, int dust, int peel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const KArgs &A) {
    LART_LAUNCH_CASE1(9, 0, 0, 0, 1) LART_LAUNCH_CASE1(9, 1, 0, 0, 1) LART_LAUNCH_CASE1(9, 0, 1, 0, 1) LART_LAUNCH_CASE1(9, 1, 1, 0, 1)
    LART_LAUNCH_CASE1(9, 0, 0, 1, 1) LART_LAUNCH_CASE1(9, 1, 0, 1, 1) LART_LAUNCH_CASE1(9, 0, 1, 1, 1) LART_LAUNCH_CASE1(9, 1, 1, 1, 1)
}
