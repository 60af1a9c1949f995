// transport kernel instantiations for line type 8 ('ly_beta': Ly-beta with the 3p -> 2s conversion to H-alpha, line_mod.f90:293-345) and for the
// pseudo line type 9 = Ly-alpha with the H2 lines next to it (par%h2_model = 'neufeld', h2_mod.f90).  The reference allows these on the plain
// Cartesian grid, the 1-column slab (H2 only) and the octree, without Stokes vectors (setup.f90:239-280, :302-348): no other kernel is built.
#include "transport_kernel.hpp"
#define LART_LT_CASE(VAR, D, P, LT) \
    if (var == VAR && dust == D && peel == P && A.par.line_type == LT) { \
        if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void *)transport_kernel<VAR, false, (bool)D, (bool)P, LT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
        hipLaunchKernelGGL((transport_kernel<VAR, false, (bool)D, (bool)P, LT>), grid, block, shmem, st, A); return 0; }
#define LART_LT_ALL(VAR, LT) LART_LT_CASE(VAR, 0, 0, LT) LART_LT_CASE(VAR, 1, 0, LT) LART_LT_CASE(VAR, 0, 1, LT) LART_LT_CASE(VAR, 1, 1, LT)
// returns 0 when a kernel was launched, 1 when the combination is not one the reference allows
extern "C" int lart_launch_transport_lt89(int var, int dust, int peel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const KArgs &A) {
    LART_LT_ALL(0, 8) LART_LT_ALL(3, 8)
    LART_LT_ALL(0, 9) LART_LT_ALL(2, 9) LART_LT_ALL(3, 9)
    return 1;
}
