// peel kernel instantiations: one per ray-march variant that may have observers (all but the xyz-symmetric octant, setup.f90:198-206) x Stokes
#include "peel_kernel.hpp"
// one march launch + one deposit launch (peel_kernel.hpp)
extern "C" void lart_launch_peel(int var, int stokes, dim3 grid, dim3 grid_dep, dim3 block, size_t shmem, hipStream_t st, const KArgs &A) {
    LART_PEEL_CASE(0, 0) LART_PEEL_CASE(0, 1) LART_PEEL_CASE(2, 0) LART_PEEL_CASE(2, 1) LART_PEEL_CASE(3, 0) LART_PEEL_CASE(3, 1)
    LART_PEEL_CASE(4, 0) LART_PEEL_CASE(4, 1) LART_PEEL_CASE(5, 0) LART_PEEL_CASE(5, 1) LART_PEEL_CASE(6, 0) LART_PEEL_CASE(6, 1)
    LART_PEEL_CASE(7, 0) LART_PEEL_CASE(7, 1)
}
