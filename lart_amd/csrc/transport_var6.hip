// transport kernel instantiations for the ray-march variant 6 (periodic in x and y: raytrace_*_car_xyper)
#include "transport_kernel.hpp"
extern "C" void lart_launch_transport_var6(int stokes, int dust, int peel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const KArgs &A) {
    LART_LAUNCH_CASE(6, 0, 0, 0) LART_LAUNCH_CASE(6, 1, 0, 0) LART_LAUNCH_CASE(6, 0, 1, 0) LART_LAUNCH_CASE(6, 1, 1, 0)
    LART_LAUNCH_CASE(6, 0, 0, 1) LART_LAUNCH_CASE(6, 1, 0, 1) LART_LAUNCH_CASE(6, 0, 1, 1) LART_LAUNCH_CASE(6, 1, 1, 1)
}
