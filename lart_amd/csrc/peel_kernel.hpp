// peel_kernel.hpp -- the deferred next-event estimator: the reference's peeling_*_outside / _inside routines
// (peelingoff_rect.f90:24-690, peelingoff_heal.f90:12-250, peelingoff_amr.f90) run over the event records the transport kernel
// queued (DPeelQ, lart_device.hpp).
//
// Work item = (record, observer).  A peel ray crosses 1.5 cells on average at tau0 = 1e6 (e^-tau is exactly 0 beyond tau = 745.2, and
// one cell has a line-centre depth of ~1e4), but a ray at a wing frequency crosses a hundred: with one item per lane and a plain
// loop a wave would march at the pace of its longest ray (measured: 13.9 ms per 8e7 records, 10x the arithmetic).  So every lane
// keeps ONE march in flight (EdgeMarch), every iteration advances all marches of the wave by one cell, and a lane whose march is
// over -- or never started, outside the image -- takes the next item from the queue.  The march kernel is bound by the latency of its
// dependent gathers (record -> cell -> next cell), so it carries nothing but the march: a ray that came through with a non-zero
// transmission (0.2 % of the items at tau0 = 1e6) is appended to a deposit list (record, observer, tau), and peel_deposit_kernel
// afterwards calls the routine of the record's kind with that observer and optical depth (its `only` / `tau_given` arguments): the
// arithmetic of a deposit (exp, pixel, Stokes rotations, atomics) is the one-piece routine's, statement for statement.
#pragma once
#include "transport_kernel.hpp"
// waves per SIMD of the march: the register budget of a transport wave of the same run (168 VGPRs with a Stokes vector: the march then fits without spills and into the
// slot a finishing transport wave leaves behind; 128 without).  Measured: 6 and 8 waves (84 / 64 VGPRs, spills) lose 3-30 %, 3 waves for the Stokes marches gains 1.4 % on C3.
#ifndef LART_PEEL_WAVES
#define LART_PEEL_WAVES(S) ((S) ? 3 : 4)
#endif
#ifndef LART_DEP_CHUNK
#define LART_DEP_CHUNK 256ull
#endif

template <int VAR, bool STOKES>
__device__ inline void peel_load(const KArgs &A, const Lds &l, size_t r, bool full, Photon &p, int &kind, double &ux, double &uy, double &uz, double &E1, double &E2, double &E3, double &grec) {
    const double *rec = A.pq.f + r * pq_recd(STOKES);          // one record, 16-byte pieces at constant offsets (lart_device.hpp, PQ_*)
    const double2 *r2 = reinterpret_cast<const double2 *>(rec);
    const double2 a0 = r2[PQ_X / 2], a1 = r2[PQ_Z / 2], a2 = r2[PQ_KY / 2], a3 = r2[PQ_WGT / 2];
    const int4 qi = *reinterpret_cast<const int4 *>(rec + PQ_INTS);
    p.x = a0.x; p.y = a0.y; p.z = a1.x; p.kx = a1.y; p.ky = a2.x; p.kz = a2.y;
    p.xfreq = a3.y; p.nsg = 0.0; p.nsd = 0.0; p.xfreq_ref = 0.0; p.inside = true; p.wgt = 0.0; p.Q = 0.0; p.U = 0.0; p.V = 0.0;
    p.ic = qi.x; p.jc = qi.y; p.kc = qi.z; p.icl = 0;
    if (is_clump(VAR)) { p.icl = p.kc; p.kc = 1; }
    kind = qi.w;
    ux = 0.0; uy = 0.0; uz = 0.0; E1 = 0.0; E2 = 0.0; E3 = 0.0; grec = 0.0;
    if (kind == PEEL_RESONANCE) { const double2 u0 = r2[PQ_UX / 2], u1 = r2[PQ_UZ / 2]; ux = u0.x; uy = u0.y; uz = u1.x; grec = (l.lt != 1) ? u1.y : A.par.g_recoil0; }
    if (STOKES) { const double2 m0 = r2[PQ_MX / 2], m1 = r2[PQ_MZ / 2]; p.mx = m0.x; p.my = m0.y; p.mz = m1.x;
                  if (peel_record_has_n<VAR>(A.par)) { const double2 n0 = r2[PQ_NX / 2]; p.nx = n0.x; p.ny = n0.y; p.nz = rec[PQ_NZ]; }   // (stored: a mirror flips k only)
                  else { p.nx = p.ky * p.mz - p.kz * p.my; p.ny = p.kz * p.mx - p.kx * p.mz; p.nz = p.kx * p.my - p.ky * p.mx; }   // n = k x m: (m, n, k) is a right-handed triad
                  if (full) { const double2 s1 = r2[PQ_U / 2]; p.Q = m1.y; p.U = s1.x; p.V = s1.y; } }
    if (full) {                                                // what only the deposit reads
        p.wgt = a3.x;
        if (kind == PEEL_RESONANCE) { if (l.lt != 1) { const double2 e0 = r2[PQ_E1 / 2]; E1 = e0.x; E2 = e0.y; E3 = rec[PQ_E3]; } else { E1 = A.par.E1; E2 = A.par.E2; E3 = A.par.E3; } }
    }
}

template <int VAR, bool STOKES>
__global__ __launch_bounds__(256, LART_PEEL_WAVES(STOKES)) void peel_kernel(KArgs A) {
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
    const unsigned long long nregions = A.pq.nregions;        // producer waves of the transport launch this kernel follows
    const bool dust = A.par.DGR > 0.0;
    const int lane = threadIdx.x & 63;
    unsigned nsteps = 0, natom = 0, ndrop = 0;
    EdgeMarch m; m.live = false; m.tau = 0.0;
    int st = 0;                                                // 0 no item, 1 marching, 2 march over (weight and deposit pending)
    unsigned long long w_base = 0, w_next = 0, w_end = 0, d_next = 0, d_end = 0; size_t rec = 0; int io = 0, kind = 0; bool exhausted = false, ray_dust = false;
    for (;;) {
        // (1) lanes without a march take the next items.  The wave takes one producer region at a time from the queue head (one atomic per
        // region, ~1e3 records) and hands its records to its idle lanes in order: the record reads of an iteration are consecutive addresses.
        const bool idle = st == 0 && !exhausted;
        const unsigned long long mk = __ballot(idle);
        if (mk) {
            if (w_next >= w_end) {                             // wave-uniform: region used up
                unsigned long long w = 0; const int leader = __ffsll((long long)mk) - 1;
                if (lane == leader) w = atomicAdd(&A.pq.pctl->q_head, 1ull);
                w = __shfl(w, leader, 64);
                if (w >= nregions) { if (idle) exhausted = true; }
                else { w_base = w * A.pq.region; w_next = 0; w_end = (unsigned long long)A.pq.count[w] * (unsigned)A.nobs; }
            }
            const unsigned long long avail = w_end - w_next, want = (unsigned long long)__popcll(mk);
            const unsigned long long mine = w_next + (unsigned long long)__popcll(mk & ((1ull << lane) - 1ull));
            w_next += want < avail ? want : avail;
            if (idle && !exhausted && mine < w_end) {          // (idle lanes beyond the region's end wait for the next region, one iteration later)
                rec = (size_t)(w_base + mine / (unsigned)A.nobs); io = (int)(mine % (unsigned)A.nobs);
                Photon p; double ux, uy, uz, E1, E2, E3, grec;
                peel_load<VAR, STOKES>(A, l, rec, false, p, kind, ux, uy, uz, E1, E2, E3, grec);
                PeelDir q; double xf;
                if (peel_ray_begin<VAR, STOKES>(A, kind, p, ux, uy, uz, grec, A.obs[io], dust, q, xf, ray_dust)) {
                    edge_begin<VAR>(A, l, ray_dust, p, xf, q.kx, q.ky, q.kz, q.dist, m, nsteps);
                    st = m.live ? 1 : 2;                       // (a ray that leaves the medium at once: tau = 0, deposit at full weight)
                }                                              // else outside the image: nothing to deposit, st stays 0
            }
        }
        if (!__ballot(st != 0)) { if (!__ballot(!exhausted)) break; continue; }
        // (2) one cell for every march of the wave; a finished march with a non-zero transmission deposits
        if (st == 1) { edge_step<VAR>(A, l, ray_dust, m, nsteps); if (!m.live) st = 2; }
        // a finished march whose transmission is not exactly zero goes to the deposit list: the wave reserves LART_DEP_CHUNK entries at a
        // time (wave-uniform d_next / d_end) and marks the unused tail of its last chunk when it ends
        const bool dep = st == 2 && (d_exp_neg(-m.tau) != 0.0 || (kind == PEEL_DIRECT && A.par.save_direc0));
        const unsigned long long dm = __ballot(dep);
        if (dm) {
            // the wave fills its chunk to the last entry and continues in a new one: nothing is left unused but the tail of a wave's last chunk
            const unsigned long long want = (unsigned long long)__popcll(dm), left = d_end - d_next;
            unsigned long long nb = 0;
            if (want > left) {                                 // wave-uniform
                const int leader = __ffsll((long long)dm) - 1;
                if (lane == leader) nb = atomicAdd(&A.pq.pctl->d_n, (unsigned long long)LART_DEP_CHUNK);
                nb = __shfl(nb, leader, 64);
            }
            if (dep) {
                const unsigned long long rk = (unsigned long long)__popcll(dm & ((1ull << lane) - 1ull));
                const unsigned long long e = rk < left ? d_next + rk : nb + (rk - left);
                if (e < A.pq.dep_cap) { A.pq.dep_idx[2 * e] = (unsigned)rec; A.pq.dep_idx[2 * e + 1] = (unsigned)io; A.pq.dep_tau[e] = m.tau; }
                else ndrop++;                                  // never silently: lart_gpu_get_tallies refuses the run (scal[LART_SCAL_PEEL_DROPPED])
            }
            if (want > left) { d_next = nb + (want - left); d_end = nb + LART_DEP_CHUNK; } else d_next += want;
        }
        if (st == 2) st = 0;
    }
    for (unsigned long long e = d_next + (unsigned)lane; e < d_end && e < A.pq.dep_cap; e += 64) A.pq.dep_idx[2 * e] = 0xffffffffu;
    const double s4 = wave_sum((double)nsteps), sd = wave_sum((double)ndrop);
    if ((threadIdx.x & 63) == 0) { if (s4 != 0.0) atomicAdd(&A.t.scal[4], s4); if (sd != 0.0) atomicAdd(&A.t.scal[LART_SCAL_PEEL_DROPPED], sd); }
}

// the deposits of the rays on the list: one entry per lane, grid-stride
#ifndef LART_DEPOSIT_WAVES
#define LART_DEPOSIT_WAVES(S) ((S) ? 3 : 4)
#endif
// (round 4: the register budget of a transport wave of the same run -- 128 VGPRs without a Stokes vector, 168 with one.  The kernel has two thousandths of the march's items and no use for its 170-200 registers, but a wave of
//  that size does not fit the slot a finishing transport wave of 128 / 168 registers leaves behind: next to the transport kernel of the following slice the deposit
//  kernel starved -- 12.7 ms per slice in profiles/r04_c3_kernel_stats.csv for 1.8e6 deposits -- and held back the queue buffer it has to release)
template <int VAR, bool STOKES>
__global__ __launch_bounds__(256, LART_DEPOSIT_WAVES(STOKES)) void peel_deposit_kernel(KArgs A) {
    extern __shared__ double sh[];
    if (A.rep.nrep > 1) {                                      // this block's copy of the contended tally sections (DRep)
        const unsigned r = blockIdx.x % (unsigned)A.rep.nrep;
        if (r) {
            const ptrdiff_t shift = (A.rep.base + (size_t)(r - 1) * A.rep.len) - (A.t.scal + A.rep.off);
            for (int q = 0; q < 8; q++) { if (A.t.c3[q]) A.t.c3[q] += shift; if (A.t.c2[q]) A.t.c2[q] += shift; }
        }
    }
    Lds l = lds_stage(sh, A.g);
    l.lt = A.par.line_type; l.dnu_hz = A.par.line_type == 7 ? A.par.delta_nu_HD_Hz : A.par.DnuHK_Hz;
    l.hd_ratio = A.par.ratio_Dfreq_HD; l.hd_ratio_a = A.par.ratio_voigta_HD; l.hd_nD = A.par.nD_over_nH;
    if (A.par.line_type >= 4 && A.par.line_type <= 6) { l.lt = 4; lds_set_multiplet(l, A.ml); }
    const unsigned long long n = A.pq.pctl->d_n < A.pq.dep_cap ? A.pq.pctl->d_n : A.pq.dep_cap;
    const bool dust = A.par.DGR > 0.0;
    unsigned nsteps = 0, natom = 0;
    for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned r = A.pq.dep_idx[2 * e];
        if (r == 0xffffffffu) continue;
        const int io = (int)A.pq.dep_idx[2 * e + 1]; const double tau = A.pq.dep_tau[e];
        Photon p; int kind; double ux, uy, uz, E1, E2, E3, grec;
        peel_load<VAR, STOKES>(A, l, r, true, p, kind, ux, uy, uz, E1, E2, E3, grec);
        if (kind == PEEL_DIRECT) peel_direct<VAR>(A, l, dust, p, nsteps, natom, io, tau);
        else if (kind == PEEL_DUST) {
            if (STOKES && A.mu.nPDF > 0) peel_dust_stokes<VAR>(A, l, p, nsteps, natom, io, tau);
            else peel_dust_nostokes<VAR>(A, l, p, nsteps, natom, io, tau);
        } else if (STOKES) peel_resonance_stokes<VAR>(A, l, dust, p, p.xfreq, ux, uy, uz, E1, E2, E3, grec, nsteps, natom, io, tau);
        else peel_resonance_nostokes<VAR>(A, l, dust, p, p.xfreq, ux, uy, uz, E1, E2, grec, nsteps, natom, io, tau);
    }
    const double s6 = wave_sum((double)natom);
    if ((threadIdx.x & 63) == 0 && s6 != 0.0) atomicAdd(&A.t.scal[6], s6);
}

#define LART_PEEL_CASE(VAR, S) \
    if (var == VAR && stokes == S) { \
        if (shmem > 64 * 1024) { (void)hipFuncSetAttribute((const void *)peel_kernel<VAR, (bool)S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
                                 (void)hipFuncSetAttribute((const void *)peel_deposit_kernel<VAR, (bool)S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); } \
        hipLaunchKernelGGL((peel_kernel<VAR, (bool)S>), grid, block, shmem, st, A); \
        hipLaunchKernelGGL((peel_deposit_kernel<VAR, (bool)S>), grid_dep, block, shmem, st, A); return; }
