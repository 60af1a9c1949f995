// transport kernel instantiations for the ray-march variant 5 (mirrors in x and y: raytrace_*_car_xysym)
#include "transport_kernel.hpp"
extern "C" void lart_launch_transport_var5(int stokes, int dust, int peel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const KArgs &A) {
    LART_LAUNCH_CASE(5, 0, 0, 0) LART_LAUNCH_CASE(5, 1, 0, 0) LART_LAUNCH_CASE(5, 0, 1, 0) LART_LAUNCH_CASE(5, 1, 1, 0)
    LART_LAUNCH_CASE(5, 0, 0, 1) LART_LAUNCH_CASE(5, 1, 0, 1) LART_LAUNCH_CASE(5, 0, 1, 1) LART_LAUNCH_CASE(5, 1, 1, 1)
}
