// transport_kernel.hpp -- the persistent-lane transport kernel.
//
// Reference loop being replaced: run_equal_number, run_simulation_mod.f90:150-202 (one packet at a
// time per MPI rank).  Here every lane owns one resident packet (state in a SoA pool in HBM between
// launches, in registers inside a launch), advances it by at most `max_events` scattering events per
// launch and, when its packet escapes or is absorbed, draws the next photon ordinal from a device
// counter (replaces the master/worker token protocol, run_simulation_mod.f90:31-64).
#pragma once
#include "lart_device.hpp"

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// waves per SIMD the register allocator must leave room for.  Measured on the headline workload (profiles/README.md):
// the plain kernel fits 128 VGPRs (4 waves) with a handful of spills and is fastest there; the Stokes triad adds ~30
// registers.  Peel-off no longer marches inside this kernel (it queues a record, peel_kernel.hpp), so it costs no occupancy.
// Waves per SIMD: lart_kernel_waves (lart_device.hpp) -- 4 without a Stokes vector (128 VGPRs), 3 with one (168), plus the instantiations the ISA lint demotes.
// Round 3 built ONE Stokes instantiation, the 1-column slab without dust (C1), at 4 waves per SIMD because it measured 8 % faster there.  Round 4 found what
// the squeeze to 128 VGPRs can cost (profiles/r04_exec_restore_miscompile.md): with dozens of spills the register allocator splits live ranges across the
// divergent `if (pending)` of vz_core_coop and puts the split copy and a spill store at the TOP of the join block, in front of the `s_or_b64 exec` that brings
// back the lanes which skipped the branch -- those lanes later read a stale register / scratch slot (C1: garbage event counters; the round-3 memory fault of
// the slab kernel with dust was the same pattern hitting an index).  Every Stokes kernel is back at 3 waves (168 VGPRs, no live-range splits of that kind), and
// tools/isa_lint.py -- run by tests/test_isa_lint.py on the BASELINE instantiations, with --build on every translation unit -- fails the build check if the pattern shows up.
#ifdef LART_MIN_WAVES
#define LART_WAVES_V5(V, S, D, P, LT) LART_MIN_WAVES
#else
#define LART_WAVES_V5(V, S, D, P, LT) lart_kernel_waves(V, S, D, P, LT)
#endif
// largest workgroup: the kernels that fit 128 VGPRs may run as ONE 1024-lane workgroup per CU (the same 4 waves per SIMD), which leaves the LDS that four
// 256-lane workgroups spend on four copies of the tables to the J(x) window of the profile tallies (lart_gpu.hip, tally layout)
#define LART_MAX_BLOCK_V5(V, S, D, P, LT) (LART_WAVES_V5(V, S, D, P, LT) >= 4 ? 1024 : 256)
#ifndef LART_OVL_CELLS_PER_ROUND
#define LART_OVL_CELLS_PER_ROUND 4           // CSR cells a lane in an overlap march walks per round of its wave
#endif


// one flight of a packet: optical depth to the next interaction (forced on the first flight: run_simulation_mod.f90:163-177,
// add_escaped_fraction_to_Jout :208-247), then the march
template <int VAR, bool DUST>
__device__ inline void fly(const KArgs &A, const Lds &l, Rng &g, Photon &p, bool &first, unsigned &nsteps, OvlHold *oh = nullptr, OvlState *os = nullptr) {
    const lart_params &par = A.par;
    double tau;
    if (first) {
        double tau0;
        if (VAR == 7) { double te_; int ow_; tau0 = ovl_march<false>(A, l, DUST, p.x, p.y, p.z, p.kx, p.ky, p.kz, p.xfreq, LART_TAU_HUGE, 0.0, nsteps, te_, ow_, nullptr, nullptr, os); }   // = raytrace_to_edge<7> (edge_begin), in the kernel's own march state
        else tau0 = raytrace_to_edge<VAR>(A, l, DUST, p, p.xfreq, p.kx, p.ky, p.kz, -1.0, nsteps);
        double esc = d_exp_neg(-tau0);
        {
            const Cell &c = A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)];
            double u1 = c.vfx * p.kx + c.vfy * p.ky + c.vfz * p.kz;
            Photon e = p; e.wgt = p.wgt * esc; e.xfreq_ref = (p.xfreq + u1) * (c.Dfreq / par.Dfreq_ref);
            if (e.wgt != 0.0) tally_escape(par, A.t, e);
            if (l.lt == 8 && e.wgt != 0.0) aux_add(A.t.scal, l, 1, e.wgt);      // W_esc1: run_simulation_mod.f90:168
        }
        double wgt1 = 1.0 - esc;
        p.wgt = p.wgt * wgt1;
        tau = (tau0 > 0.0) ? -d_log(1.0 - g.next() * wgt1) : LART_HUGEST;
        first = false;
    } else tau = -d_log(g.next());
    double u_own = 0.0;
    if (VAR == 7) u_own = g.next();                        // the draw of sample_owner_clump (raytrace_clump.f90:648), taken here: a packet that escapes instead has no later draw to shift
    LART_PROF_MARK(p, 2);
#ifdef LART_EXPERIMENT_RAY2                            // timing experiment: cost of one flight (the copy's escapes are tallied twice: timing only)
    { Photon p2 = p; unsigned ns2 = 0; raytrace_to_tau<VAR>(A, l, DUST, p2, tau, ns2); if (p2.x == 1e300) p.x = p2.y; }
#endif
    if (VAR == 7) {                                        // overlapping clumps: the march may be left in flight in *os (the kernel's loop advances it)
        double t_end; int owner;
        if (ovl_start<true>(A, l, DUST, p.x, p.y, p.z, p.kx, p.ky, p.kz, p.xfreq, tau, u_own, nsteps, t_end, owner, oh, *os)) ovl_apply(A, p, t_end, owner);
    } else raytrace_to_tau<VAR>(A, l, DUST, p, tau, nsteps, u_own, oh);
    LART_PROF_MARK(p, 3);
}

// the same for VAR 9 (overlapping clumps, at most 64 of them), called by the whole wave: the flights that need the clump table -- the optical depth to the surface of a
// first flight, a flight that does not end on its carried holder list -- are executed by the wave together (ovl_coop), the rest is each lane's own
template <bool DUST>
__device__ inline void fly_coop(const KArgs &A, const Lds &l, Rng &g, Photon &p, bool act, bool &first, unsigned &nsteps, OvlHold &oh) {
    const lart_params &par = A.par;
    double tau0 = 0.0; int none = 0;
    const bool born = act && first;
    if (__ballot(born)) ovl_coop<false>(A, l, DUST, born, p, LART_TAU_HUGE, 0.0, nsteps, tau0, none, nullptr);        // = raytrace_to_edge (edge_begin) of the first flight
    double tau = 0.0, u_own = 0.0, t_end = 0.0; int owner = 0; bool need = false;
    if (act) {
        if (first) {
            const double esc = d_exp_neg(-tau0);
            const Cell &c = A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)];
            const double u1 = c.vfx * p.kx + c.vfy * p.ky + c.vfz * p.kz;
            Photon e = p; e.wgt = p.wgt * esc; e.xfreq_ref = (p.xfreq + u1) * (c.Dfreq / par.Dfreq_ref);
            if (e.wgt != 0.0) tally_escape(par, A.t, e);
            const double wgt1 = 1.0 - esc;
            p.wgt = p.wgt * wgt1;
            tau = (tau0 > 0.0) ? -d_log(1.0 - g.next() * wgt1) : LART_HUGEST;
            first = false;
        } else tau = -d_log(g.next());
        u_own = g.next();                                      // the draw of sample_owner_clump (raytrace_clump.f90:648)
        need = !ovl_carried(A, l, DUST, p.x, p.y, p.z, p.kx, p.ky, p.kz, p.xfreq, tau, u_own, nsteps, t_end, owner, oh);
    }
    LART_PROF_MARK(p, 2);
    if (__ballot(need)) ovl_coop<true>(A, l, DUST, need, p, tau, u_own, nsteps, t_end, owner, &oh);
    if (act) ovl_apply(A, p, t_end, owner);
    LART_PROF_MARK(p, 3);
}

// LT: line%line_type as a compile-time constant (1 single line, 2 Ly-alpha doublet, 7 hydrogen + deuterium), handed to the device functions through Lds
template <int VAR, bool STOKES, bool DUST, bool PEEL, int LT>
__global__ __launch_bounds__(LART_MAX_BLOCK_V5(VAR, STOKES, DUST, PEEL, LT), LART_WAVES_V5(VAR, STOKES, DUST, PEEL, LT)) void transport_kernel(KArgs A) {
    extern __shared__ double sh[];
    if (A.rep.nrep > 1) {                                      // this block's copy of the contended tally sections (DRep)
        const unsigned r = blockIdx.x % (unsigned)A.rep.nrep;
        if (r) {
            const ptrdiff_t shift = (A.rep.base + (size_t)(r - 1) * A.rep.len) - (A.t.scal + A.rep.off);
            for (int q = 0; q < 8; q++) { if (A.t.c3[q]) A.t.c3[q] += shift; if (A.t.c2[q]) A.t.c2[q] += shift; }
            if (A.jpa.J1) A.jpa.J1 += shift; if (A.jpa.P1) A.jpa.P1 += shift; if (A.jpa.P1_new) A.jpa.P1_new += shift;
            if (A.jpa.Pconv) A.jpa.Pconv += shift; if (A.t.peel_Ha) A.t.peel_Ha += shift;
        }
    }
    Lds l_ = lds_stage(sh, A.g); lds_set_dipole(l_, A.par.E1); l_.lt = LT; l_.dnu_hz = (LT == 2) ? A.par.DnuHK_Hz : (LT == 7) ? A.par.delta_nu_HD_Hz : 0.0;
    if (LT == 4) lds_set_multiplet(l_, A.ml);
    if (PEEL) lds_set_fov(l_, A);
    if (LT == 7) { l_.hd_ratio = A.par.ratio_Dfreq_HD; l_.hd_ratio_a = A.par.ratio_voigta_HD; l_.hd_nD = A.par.nD_over_nH; }
    if (A.jpa.lds_bins > 0) {                                  // per-block copies of the P1 / P1_new profile tallies
        double *jp = sh + VT_END + (A.g.nx + A.g.ny + A.g.nz + 3) + (size_t)LART_RNG_BUF * blockDim.x;
        for (int i = threadIdx.x; i < 2 * A.jpa.lds_bins; i += blockDim.x) jp[i] = 0.0;
        l_.jp_P = jp; l_.jp_Pn = jp + A.jpa.lds_bins;
        __syncthreads();
    }
    const int nJ = A.jpa.lds_j_w * A.jpa.nbins;                 // line-centre window of J1 per block
    if (nJ > 0) {
        double *jj = sh + VT_END + (A.g.nx + A.g.ny + A.g.nz + 3) + (size_t)LART_RNG_BUF * blockDim.x + 2 * (size_t)A.jpa.lds_bins;
        for (int i = threadIdx.x; i < nJ; i += blockDim.x) jj[i] = 0.0;
        l_.jp_J = jj; l_.j_lo = A.jpa.lds_j_lo; l_.j_w = A.jpa.lds_j_w;
        __syncthreads();
    }
    // wave-local scratch of the cooperative sampler, behind everything else: LART_COOP_BYTES per wave
    l_.coop = (unsigned char *)(sh + VT_END + (A.g.nx + A.g.ny + A.g.nz + 3) + (size_t)LART_RNG_BUF * blockDim.x + 2 * (size_t)A.jpa.lds_bins + (size_t)nJ) + LART_COOP_BYTES * (threadIdx.x >> 6);
    if (PEEL) {                                                // this wave's record counter of the peel queue
        l_.pq_cnt = (unsigned *)(l_.coop - LART_COOP_BYTES * (threadIdx.x >> 6) + LART_COOP_BYTES * (blockDim.x >> 6)) + (threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0) *l_.pq_cnt = 0u;
    }
    // behind everything else, in this order (the launch is sized for them by lart_gpu_step, the same sums): the block's accumulators of the ly_beta / H2 weight budget
    // (8 doubles), the clump records of a small overlapping population, the observers' records of the peel prefilter
    size_t tail = (((size_t)(VT_END + (A.g.nx + A.g.ny + A.g.nz + 3) + (size_t)LART_RNG_BUF * blockDim.x + 2 * (size_t)A.jpa.lds_bins + (size_t)nJ) * sizeof(double)
                     + (size_t)(LART_COOP_BYTES + 4) * (blockDim.x >> 6)) + 15) & ~(size_t)15;
    if (LT == 8 || LT == 9) {
        double *ax = (double *)((unsigned char *)sh + tail); tail += 8 * sizeof(double);
        if (threadIdx.x < 8) ax[threadIdx.x] = 0.0;
        __syncthreads();
        l_.aux = ax;
    }
    if (is_ovl(VAR)) { lds_stage_clumps(l_, A.cl, (unsigned char *)sh + tail); tail += (lds_clump_bytes(A.cl) + 15) & ~(size_t)15; }
    if (PEEL && A.nobs > 0 && A.nobs <= LART_LDS_OBSERVERS) {
        double *ob = (double *)((unsigned char *)sh + tail); const double *src = (const double *)A.obs;
        for (int i = threadIdx.x; i < A.nobs * (int)(sizeof(DObs) / sizeof(double)); i += blockDim.x) ob[i] = src[i];
        __syncthreads();
        l_.obs = (const DObs *)ob;
    }
    const Lds l = l_;
    double *rng_lds = sh + VT_END + (A.g.nx + A.g.ny + A.g.nz + 3) + threadIdx.x;      // [LART_RNG_BUF][blockDim.x] draws
    const lart_params &par = A.par;
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // pool.n is a multiple of blockDim.x
    const int64_t P = A.pool.n;
    double *F = A.pool.f;

    Photon p; Rng g; g.attach(rng_lds, blockDim.x);
    OvlHold oh; oh.clear();                                        // VAR 7: holders carried from one flight to the next (not kept between launches)
    OvlState os; os.live = false;                                                 // VAR 7: the march of a flight that left its clump, between two rounds of the wave
    g.k0 = (uint32_t)A.seed; g.k1 = (uint32_t)(A.seed >> 32);                  // (a lane that never holds a packet still lends its ALU to Rng::prefill_sparse: the key is the run's)
#ifdef LART_PROFILE
    for (int i = 0; i < 16; i++) p.pt[i] = 0;
    for (int i = 0; i < 4; i++) p.pc[i] = 0;
    p.plast = clock64();
#endif
    int flags = A.pool.flags[lane];
    bool alive = flags & FLAG_ALIVE, first = flags & FLAG_FIRST;
    p.iband = (flags & FLAG_BAND2) ? 2 : 1;
    long long k = A.pool.k[lane];
    if (alive) {
        p.x = F[F_X * P + lane]; p.y = F[F_Y * P + lane]; p.z = F[F_Z * P + lane];
        p.kx = F[F_KX * P + lane]; p.ky = F[F_KY * P + lane]; p.kz = F[F_KZ * P + lane];
        p.xfreq = F[F_XFREQ * P + lane]; p.wgt = F[F_WGT * P + lane]; p.nsg = F[F_NSG * P + lane]; p.nsd = F[F_NSD * P + lane];
        if (STOKES) {
            p.mx = F[F_MX * P + lane]; p.my = F[F_MY * P + lane]; p.mz = F[F_MZ * P + lane];
            p.nx = F[F_NX * P + lane]; p.ny = F[F_NY * P + lane]; p.nz = F[F_NZ * P + lane];
            p.Q = F[F_Q * P + lane]; p.U = F[F_U * P + lane]; p.V = F[F_V * P + lane];
        }
        p.ic = A.pool.ic[lane]; p.jc = A.pool.jc[lane]; p.kc = A.pool.kc[lane]; p.inside = true; p.xfreq_ref = 0.0; p.icl = 0;
        if (is_clump(VAR)) { p.icl = p.ic; p.ic = 1; }               // clump mode keeps the owner clump in the cell-index plane
        if (VAR == 8) p.shear = F[F_SHEAR * P + lane];
        g.init(A.seed, (uint64_t)(A.first_id + k * A.stride), A.pool.rng_n[lane]);
    }

    LART_PROF_MARK(p, 0);
    unsigned nsteps = 0, npeel = 0, nev = 0, ndone = 0;
    double nsg_done = 0.0, nsd_done = 0.0;
    int budget = A.max_events;

    // The event loop is left by the WAVE, not by the lane: a lane without a packet (none left to draw, or its launch budget used up) stays in and
    // lends its ALU to the packets of the wave -- the random-draw top-up (Rng::prefill_sparse) and the rejection loops of the vz sampler
    // (vz_core_coop / vz_wing_coop) spread over all 64 lanes.  In the drain phase a wave holds a few packets or one (lart_gpu_compact deals the
    // last packets out over the wave slots), and an event of a lone packet is what the tail of a run is made of.
    bool exhausted = false;
    for (;;) {
        if (!alive && !exhausted && budget > 0) {
            unsigned long long kk = atomicAdd(&A.ctl->next_k, 1ULL);
            if (kk >= (unsigned long long)A.count) exhausted = true;
            else {
                k = (long long)kk;
                g.init(A.seed, (uint64_t)(A.first_id + k * A.stride), 0);
                if (VAR == 8) p.shear = 0.0;                   // generate_photon.f90:141
                p.iband = 1;                                   // :139
                oh.na = 0;
                generate_photon<VAR, STOKES, DUST, PEEL>(A, l, g, p, npeel);
                if (A.allph) allph_initial(A, p, k);
                if (A.trace_slot) trace_rec<VAR>(A, k, g.n, p, 0);
                alive = true; first = true;
                if (PEEL) budget--;                            // the birth queued one peel record (direct light): lanes x max_events records bound the queue
                LART_PROF_MARK(p, 1);
            }
        }
        const bool act = alive && budget > 0;
        const unsigned long long am = __ballot(act);
        if (!am) break;
        g.template fill<LART_FILL_TOP(STOKES)>(act);          // draws of the flight and the vz sampler; scatter_event fills again before scatter_post
#ifdef LART_EXPERIMENT_PREFILL2                        // timing experiment: cost of one buffer fill
        if (act) g.template prefill<LART_FILL_TOP(STOKES)>();
#endif
        LART_PROF_TRIP(p, 2);
        if (VAR == 9) fly_coop<DUST>(A, l, g, p, act, first, nsteps, oh);
        else if (act && !(VAR == 7 && os.live)) fly<VAR, DUST>(A, l, g, p, first, nsteps, VAR == 7 ? &oh : nullptr, VAR == 7 ? &os : nullptr);
        if (VAR == 7 && act && os.live) {                  // a flight that left its clump: a few cells of the march per round, the other lanes of the wave go on scattering
            double t_end = 0.0; int owner = 0;
            if (ovl_advance<true>(A, l, DUST, p.x, p.y, p.z, p.kx, p.ky, p.kz, p.xfreq, 0.0, nsteps, t_end, owner, nullptr, os, LART_OVL_CELLS_PER_ROUND)) {
                ovl_finish<true>(A, os, t_end, owner); ovl_apply(A, p, t_end, owner);
            }
        }
        const bool marching = VAR == 7 && os.live;
        { const bool on = act && p.inside && !marching; scatter_event<VAR, STOKES, DUST, PEEL>(A, l, g, p, npeel, on); if (on) { nev++; budget--; } }
        LART_PROF_MARK(p, 6);
        if (act && !marching) {
            if (A.trace_slot) trace_rec<VAR>(A, k, g.n, p, p.inside ? 1 : 2);
            if (!p.inside) {
                nsg_done += p.nsg; nsd_done += p.nsd; ndone++;
                if (A.log_nscatt) { A.log_nscatt[k] = p.nsg + p.nsd; A.log_xfreq[k] = p.xfreq_ref; }
                if (A.allph) allph_final<STOKES>(A, p, k);
                alive = false;
            }
        }
        LART_PROF_MARK(p, 7);
    }

    if (nJ > 0) {                                              // flush the block's J1 window
        __syncthreads();
        for (int i = threadIdx.x; i < nJ; i += blockDim.x) {
            const double v = l.jp_J[i];
            if (v != 0.0) atomicAdd(&A.jpa.J1[(size_t)(A.jpa.lds_j_lo + i % A.jpa.lds_j_w) + (size_t)A.par.nxfreq * (i / A.jpa.lds_j_w)], v);
        }
    }
    if (A.jpa.lds_bins > 0) {                                  // flush the block's profile tallies: one atomic per touched bin
        __syncthreads();
        for (int i = threadIdx.x; i < A.jpa.lds_bins; i += blockDim.x) {
            double v = l.jp_P[i], w = l.jp_Pn[i];
            if (A.jpa.P1 && v != 0.0) atomicAdd(&A.jpa.P1[i], v);
            if (A.jpa.P1_new && w != 0.0) atomicAdd(&A.jpa.P1_new[i], w);
        }
    }
    if (l.aux) {                                               // flush the block's weight budget: one atomic per term
        __syncthreads();
        if (threadIdx.x < 8 && l.aux[threadIdx.x] != 0.0) atomicAdd(&A.t.scal[LART_SCAL_AUX + threadIdx.x], l.aux[threadIdx.x]);
    }
    if (PEEL && (threadIdx.x & 63) == 0) A.pq.count[lane >> 6] = *l.pq_cnt;
    A.pool.flags[lane] = (alive ? FLAG_ALIVE : 0) | (first ? FLAG_FIRST : 0) | ((LT == 8 && p.iband == 2) ? FLAG_BAND2 : 0);
    if (alive) {
        A.pool.k[lane] = k; A.pool.rng_n[lane] = g.n;
        F[F_X * P + lane] = p.x; F[F_Y * P + lane] = p.y; F[F_Z * P + lane] = p.z;
        F[F_KX * P + lane] = p.kx; F[F_KY * P + lane] = p.ky; F[F_KZ * P + lane] = p.kz;
        F[F_XFREQ * P + lane] = p.xfreq; F[F_WGT * P + lane] = p.wgt; F[F_NSG * P + lane] = p.nsg; F[F_NSD * P + lane] = p.nsd;
        if (STOKES) {
            F[F_MX * P + lane] = p.mx; F[F_MY * P + lane] = p.my; F[F_MZ * P + lane] = p.mz;
            F[F_NX * P + lane] = p.nx; F[F_NY * P + lane] = p.ny; F[F_NZ * P + lane] = p.nz;
            F[F_Q * P + lane] = p.Q; F[F_U * P + lane] = p.U; F[F_V * P + lane] = p.V;
        }
        if (VAR == 8) F[F_SHEAR * P + lane] = p.shear;
        A.pool.ic[lane] = is_clump(VAR) ? p.icl : p.ic; A.pool.jc[lane] = p.jc; A.pool.kc[lane] = p.kc;
    }
#ifdef LART_PROFILE
    LART_PROF_MARK(p, 0);
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < 12; i++) atomicAdd(&A.t.scal[8 + i], (double)p.pt[i]);
    if ((threadIdx.x & 63) == 0) for (int i = 12; i < 16; i++) atomicAdd(&A.t.scal[LART_SCAL_AUX + i - 12], (double)p.pt[i]);      // (line type 1 leaves the aux scalars free)
    for (int i = 0; i < 4; i++) { double v = wave_sum((double)p.pc[i]); if ((threadIdx.x & 63) == 0) atomicAdd(&A.t.scal[20 + i], v); }
#endif
#ifdef LART_RNG_STATS
    { double v0 = wave_sum((double)g.nslow), v1 = wave_sum((double)g.nslow_w); if ((threadIdx.x & 63) == 0) { atomicAdd(&A.t.scal[8], v0); atomicAdd(&A.t.scal[9], v1); } }
#endif
    // per-wave pre-reduction, one atomic per wave per scalar
    double s0 = wave_sum(nsg_done), s1 = wave_sum(nsd_done), s2 = wave_sum((double)nev), s3 = wave_sum((double)nsteps),
           s4 = wave_sum((double)npeel), s5 = wave_sum((double)ndone), s6 = wave_sum(alive ? 1.0 : 0.0);
    if ((threadIdx.x & 63) == 0) {
        if (s0 != 0.0) atomicAdd(&A.t.scal[0], s0);
        if (s1 != 0.0) atomicAdd(&A.t.scal[1], s1);
        if (s2 != 0.0) atomicAdd(&A.t.scal[2], s2);
        if (s3 != 0.0) atomicAdd(&A.t.scal[3], s3);
        if (s4 != 0.0) atomicAdd(&A.t.scal[4], s4);
        if (s5 != 0.0) atomicAdd(&A.t.scal[5], s5);
        if (s6 != 0.0) atomicAdd(&A.ctl->n_active, (unsigned long long)(s6 + 0.5));
    }
}

// launcher signature shared by the per-variant translation units
typedef void (*transport_launch_fn)(int stokes, int dust, int peel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const KArgs &A);

#define LART_LAUNCH_CASE1(VAR, S, D, P, LT) \
    if (stokes == S && dust == D && peel == P && (A.par.line_type == LT || (LT == 4 && (A.par.line_type == 5 || A.par.line_type == 6)))) { \
        if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void *)transport_kernel<VAR, (bool)S, (bool)D, (bool)P, LT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
        hipLaunchKernelGGL((transport_kernel<VAR, (bool)S, (bool)D, (bool)P, LT>), grid, block, shmem, st, A); return; }
#define LART_LAUNCH_CASE(VAR, S, D, P) LART_LAUNCH_CASE1(VAR, S, D, P, 1) LART_LAUNCH_CASE1(VAR, S, D, P, 2) LART_LAUNCH_CASE1(VAR, S, D, P, 4) LART_LAUNCH_CASE1(VAR, S, D, P, 7)
