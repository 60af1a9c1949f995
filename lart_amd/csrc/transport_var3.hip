// transport kernel instantiations for the octree march (variant 3: raytrace_amr.f90 / scattering_amr.f90)
#include "transport_kernel.hpp"
extern "C" void lart_launch_transport_var3(int stokes, int dust, int peel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const KArgs &A) {
    LART_LAUNCH_CASE(3, 0, 0, 0) LART_LAUNCH_CASE(3, 1, 0, 0) LART_LAUNCH_CASE(3, 0, 1, 0) LART_LAUNCH_CASE(3, 1, 1, 0)
    LART_LAUNCH_CASE(3, 0, 0, 1) LART_LAUNCH_CASE(3, 1, 0, 1) LART_LAUNCH_CASE(3, 0, 1, 1) LART_LAUNCH_CASE(3, 1, 1, 1)
}
