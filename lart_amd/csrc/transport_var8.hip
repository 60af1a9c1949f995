// transport kernel instantiations for the ray-march variant 8 (xy-periodic box with a shearing x boundary: raytrace_to_tau_car_xyper_shear;
// its edge / peel marches are the plain Cartesian ones, as setup_procedure leaves them)
#include "transport_kernel.hpp"
extern "C" void lart_launch_transport_var8(int stokes, int dust, int peel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const KArgs &A) {
    LART_LAUNCH_CASE(8, 0, 0, 0) LART_LAUNCH_CASE(8, 1, 0, 0) LART_LAUNCH_CASE(8, 0, 1, 0) LART_LAUNCH_CASE(8, 1, 1, 0)
    LART_LAUNCH_CASE(8, 0, 0, 1) LART_LAUNCH_CASE(8, 1, 0, 1) LART_LAUNCH_CASE(8, 0, 1, 1) LART_LAUNCH_CASE(8, 1, 1, 1)
}
