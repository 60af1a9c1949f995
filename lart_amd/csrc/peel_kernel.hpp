// peel_kernel.hpp -- the deferred next-event estimator: the reference's peeling_*_outside / _inside routines
// (peelingoff_rect.f90:24-690, peelingoff_heal.f90:12-250, peelingoff_amr.f90) run over the event records the transport kernel
// queued (DPeelQ, lart_device.hpp), one record per lane with a grid-stride loop: every lane marches one ray per observer
// (raytrace_to_edge / _dist with the tau >= 745.2 cut) and deposits with atomics when the attenuated weight is non-zero.
// The arithmetic of a deposit is the peel routine's, unchanged; only the place where it runs moved out of the scattering.
#pragma once
#include "transport_kernel.hpp"

template <int VAR, bool STOKES>
__global__ __launch_bounds__(256, 3) void peel_kernel(KArgs A) {
    extern __shared__ double sh[];
    if (A.rep.nrep > 1) {                                      // this block's copy of the contended tally sections (DRep)
        const unsigned r = blockIdx.x % (unsigned)A.rep.nrep;
        if (r) {
            const ptrdiff_t shift = (A.rep.base + (size_t)(r - 1) * A.rep.len) - (A.t.scal + A.rep.off);
            for (int q = 0; q < 8; q++) { if (A.t.c3[q]) A.t.c3[q] += shift; if (A.t.c2[q]) A.t.c2[q] += shift; }
        }
    }
    Lds l = lds_stage(sh, A.g);
    // the line type at run time (one kernel per march variant serves every line): calc_voigt inside the march reads it
    l.lt = A.par.line_type; l.dnu_hz = A.par.line_type == 7 ? A.par.delta_nu_HD_Hz : A.par.DnuHK_Hz;
    l.hd_ratio = A.par.ratio_Dfreq_HD; l.hd_ratio_a = A.par.ratio_voigta_HD; l.hd_nD = A.par.nD_over_nH;
    if (A.par.line_type >= 4 && A.par.line_type <= 6) { l.lt = 4; lds_set_multiplet(l, A.ml); }
    const size_t n = A.ctl->q_n < A.pq.cap ? A.ctl->q_n : A.pq.cap, C = A.pq.cap;
    const double *f = A.pq.f; const int *qc = A.pq.cell;
    const bool dust = A.par.DGR > 0.0;
    unsigned nsteps = 0, natom = 0;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (size_t)gridDim.x * blockDim.x) {
        Photon p;
        p.x = f[PQ_X * C + r]; p.y = f[PQ_Y * C + r]; p.z = f[PQ_Z * C + r]; p.kx = f[PQ_KX * C + r]; p.ky = f[PQ_KY * C + r]; p.kz = f[PQ_KZ * C + r];
        p.wgt = f[PQ_WGT * C + r]; p.xfreq = f[PQ_XA * C + r]; p.nsg = 0.0; p.nsd = 0.0; p.xfreq_ref = 0.0; p.inside = true;
        if (STOKES) { p.mx = f[PQ_MX * C + r]; p.my = f[PQ_MY * C + r]; p.mz = f[PQ_MZ * C + r]; p.nx = f[PQ_NX * C + r]; p.ny = f[PQ_NY * C + r]; p.nz = f[PQ_NZ * C + r];
                      p.Q = f[PQ_Q * C + r]; p.U = f[PQ_U * C + r]; p.V = f[PQ_V * C + r]; }
        p.ic = qc[r]; p.jc = qc[C + r]; p.kc = qc[2 * C + r]; p.icl = 0;
        if (VAR == 4) { p.icl = p.kc; p.kc = 1; }
        const int kind = qc[3 * C + r];
        if (kind == PEEL_DIRECT) peel_direct<VAR>(A, l, dust, p, nsteps, natom);
        else if (kind == PEEL_DUST) {
            if (STOKES && A.mu.nPDF > 0) peel_dust_stokes<VAR>(A, l, p, nsteps, natom);
            else peel_dust_nostokes<VAR>(A, l, p, nsteps, natom);
        } else {
            const double ux = f[PQ_UX * C + r], uy = f[PQ_UY * C + r], uz = f[PQ_UZ * C + r], E1 = f[PQ_E1 * C + r], E2 = f[PQ_E2 * C + r], E3 = f[PQ_E3 * C + r], grec = f[PQ_GREC * C + r];
            if (STOKES) peel_resonance_stokes<VAR>(A, l, dust, p, p.xfreq, ux, uy, uz, E1, E2, E3, grec, nsteps, natom);
            else peel_resonance_nostokes<VAR>(A, l, dust, p, p.xfreq, ux, uy, uz, E1, E2, grec, nsteps, natom);
        }
    }
    const double s4 = wave_sum((double)nsteps), s6 = wave_sum((double)natom);
    if ((threadIdx.x & 63) == 0) { if (s4 != 0.0) atomicAdd(&A.t.scal[4], s4); if (s6 != 0.0) atomicAdd(&A.t.scal[6], s6); }
}

#define LART_PEEL_CASE(VAR, S) \
    if (var == VAR && stokes == S) { \
        if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void *)peel_kernel<VAR, (bool)S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
        hipLaunchKernelGGL((peel_kernel<VAR, (bool)S>), grid, block, shmem, st, A); return; }
