// lart_device.hpp -- CDNA4 (gfx950) device functions of the LaRT photon-packet path.
//
// One photon packet per lane, all arithmetic IEEE f64 (reference: wp = real64, define.f90:25-26).
// Per-lane counter-based RNG (Philox4x32-10 keyed (seed, photon id), counter = draw index) replaces
// the reference's per-rank MT19937-64 stream (random_mt.f90:579-630) so a packet's history does not
// depend on which lane / GPU runs it.  Read-only medium = one 64-byte record per cell (one cache
// line sector per gather); Voigt tables and cell faces live in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lart_gpu.h"
#define LART_HD __device__ inline
#include "lart_math.hpp"

#define LART_PI      3.141592653589793238462643383279502884197
#define LART_TWOPI   6.283185307179586476925286766559005768394
#define LART_FOURPI  12.56637061435917295385057353311801153679
#define LART_HALFPI  (LART_PI / 2.0)
#define LART_RAD2DEG (180.0 / LART_PI)
#define LART_HUGEST  1.7976931348623157e308
// a march in the periodic octree box that keeps wrapping without absorbing anything (|kz| < 1e-7 through vacuum) is ended here: the
// reference would not return from it, and a wave that never finishes takes the GPU with it
#define LART_AMR_MAX_WRAPS 10000000
#define LART_TAU_HUGE 745.2                 // raytrace_car.f90:432 -- exp(-tau_huge) underflows

// ---------------------------------------------------------------------------------------------
// medium: one record per cell, Fortran (i,j,k) order, 64-byte aligned
// ---------------------------------------------------------------------------------------------
struct __attribute__((aligned(64))) Cell {
    double rhokap, voigt_a, Dfreq, vfx, vfy, vfz, rhokapD, pad;
};

struct DGrid {
    int nx, ny, nz, i0, j0, k0;
    double xmin, ymin, zmin, dx, dy, dz;
    const Cell *cell;
    const double *faces;          // xface(nx+1) | yface(ny+1) | zface(nz+1), global copy (staged to LDS)
};

// octree (amr_grid_type, octree_mod.f90:19-138): flat arrays with the reference's 1-based numbering; leaf tables are the
// Cell records of DGrid indexed by the leaf number (photon ic = leaf index, jc = kc = 1)
struct __attribute__((aligned(32))) AmrGeom { double cx, cy, cz, ch; };
struct DAmr {
    int ncells, nleaf;
    const int *children, *neighbor, *ileaf, *icell_of_leaf;    // children(8,ncells), neighbor(6,ncells)
    const AmrGeom *geom;                                       // per cell
    double lo, hi;                                             // cubic box [lo,hi]^3
    int jpa_geometry; double dr_JPa, dz_JPa;                   // position-binned profile tallies (amr_JPa_bin)
    int nr_JPa, nz_JPa;
};

// clump medium (VAR 4): clump_mod.f90:27-118.  Geometry and physics in separate records: the CSR search only touches geometry.
struct ClumpGeo  { double x, y, z, r2; };                                           // centre, radius^2
struct ClumpPhys { double rhokap, rhokapD, voigt_a, Dfreq, vx, vy, vz, pad; };     // v in units of the clump's thermal speed
// VAR 4 (disjoint clumps) and VAR 7 (overlapping clumps) share the owner-clump bookkeeping of the scattering and peel-off routines
__host__ __device__ constexpr bool is_clump(int var) { return var == 4 || var == 7 || var == 9; }
// Waves per SIMD a transport kernel is built for (its __launch_bounds__, transport_kernel.hpp) -- 4 = 128 VGPRs, workgroups up to 1024 lanes; 3 = 168 VGPRs, up to 256 lanes.
// One list for the kernel and the host (lart_gpu_step refuses a workgroup the kernel was not built for).  3 waves: every kernel with a Stokes vector and the
// wave-cooperative overlap kernel (profiles/r04_exec_restore_miscompile.md), and every other instantiation in which tools/isa_lint.py --build finds allocator copies
// in front of an exec restore at 128 VGPRs: a hit means "this build of this kernel must not ship", whatever the copies turn out to be.
__host__ __device__ constexpr int lart_kernel_waves(int var, bool stokes, bool dust, bool peel, int lt) {
    return (stokes || var == 9) ? 3
         : (var == 4 && !dust && peel && lt == 7) ? 3             // round 4: .LBB19_547 of transport_kernel<4,false,false,true,7>
         : 4;
}
__host__ __device__ constexpr bool is_ovl(int var) { return var == 7 || var == 9; }      // overlapping clumps; 9 = at most 64 of them, flights by the wave (ovl_coop)
// VAR 8 = the xy-periodic box with a shearing x boundary (raytrace_to_tau_car_xyper_shear, raytrace_car.f90:2677-2954, chosen at setup.f90:967-969
// when par%Omega /= 0): the tau march of VAR 6 with photon%vfy_shear; setup_procedure leaves raytrace_to_edge at its default there, the
// PLAIN Cartesian march (setup.f90:941, :967-969 assign raytrace_to_tau only), so every edge / peel march of VAR 8 is VAR 0's
__host__ __device__ constexpr bool is_per(int var) { return var == 6 || var == 8; }
__host__ __device__ constexpr int edge_var(int var) { return var == 8 ? 0 : var; }
struct DClumps {
    long long N;
    const ClumpGeo *geo; const ClumpPhys *phys;
    const int *cg_start, *cg_list;                             // CSR over cgx^3 cells, 1-based entries as the reference builds them
    int cgx; double cg_min, cg_d, cg_inv, sphere_R, Dfreq_ref, radius_max;
    const unsigned long long *cg_mask;                          // populations of at most 128 clumps: [2] words per CSR cell, bit icl-1 set when clump icl is registered in it (nullptr otherwise)
};

struct DObs { double x, y, z, rm[9]; };

struct DMueller { int nPDF; const double *coss, *S11, *S12, *S33, *S34, *phase_PDF; const int *alias; };

// fused tally buffer (one RCCL reduce): pointers into it
struct DTally {
    double *scal;                 // [0] nscatt_gas [1] nscatt_dust [2] n_events [3] n_cellsteps [4] n_cellsteps_peel [5] nphotons_done
    double *Jout, *Jin, *Jabs, *Jmu;
    double *c3[8];                // scatt, direc, I, Q, U, V, direc0 (par%save_direc0), scatt_dust (par%save_dust_scattered)   (nobs * nxfreq*nxim*nyim)
    double *c2[8];                // *_2D (no scatt_dust image)      (nobs * nxim*nyim)
    double *Jout_Ha, *Jabs_Ha;    // line type 8: the band-2 (H-alpha) spectra grid%Jout_Ha, grid%Jabs_Ha (nxfreq_Ha)
    double *peel_Ha;              // line type 8: observer(:)%peel_Ha (nobs * nxfreq_Ha*nxim*nyim)
};
// scalars 25..31 of the fused tally buffer: line type 8: W_conv, W_esc1, W_abs1, W_esc2, W_abs2 (define.f90:471-475); H2: W_H2abs, W_H2scat, W_H2pump(1..4)
#define LART_SCAL_AUX 25

// resident photon pool (SoA, coalesced load/store at launch boundaries)
enum { F_X = 0, F_Y, F_Z, F_KX, F_KY, F_KZ, F_XFREQ, F_WGT, F_NSG, F_NSD,
       F_MX, F_MY, F_MZ, F_NX, F_NY, F_NZ, F_Q, F_U, F_V, F_SHEAR /* photon%vfy_shear, VAR 8 only */, F_COUNT };
enum { FLAG_ALIVE = 1, FLAG_FIRST = 2, FLAG_BAND2 = 4 };
struct DPool {
    int64_t n;                    // lanes
    double *f;                    // F_COUNT planes of n doubles
    unsigned long long *rng_n;    // draw counter
    long long *k;                 // photon ordinal (id = first_id + k*stride)
    int *ic, *jc, *kc, *flags;
};

// J(x) / P_alpha profile tallies (compile-time CALCJ/CALCP/CALCPnew in the reference; run-time switches here)
// profile tallies; P1 / P1_new are accumulated per block in LDS when `lds_bins` > 0 and flushed once at kernel end
// J1: the frequency bins [lds_j_lo, lds_j_lo + lds_j_w) around the line centre of every profile bin are accumulated per block in LDS too when they fit
// (lds_j_w > 0): scattered f64 atomics run at the memory side at ~1e10 per second for the whole chip, which is what one deposit per event asks for
struct DJPa { int calc_J, calc_P, calc_Pnew, nbins; const int *ind_bin; double cross0; double *J1, *P1, *P1_new; int lds_bins, lds_j_lo, lds_j_w; double *Pconv; };   // Pconv: add_to_Pconv (line type 8)
// Copies of the contended tally sections (peel cubes and images, J1, P1, P1_new = the tail [off, off+len) of the fused buffer).
// Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2: same-address atomics from different XCDs meet
// at the memory side and serialise there (measured: 70x slow-down of a 64-bin P_alpha profile).  Block b therefore adds into
// copy b % nrep -- copy 0 is the fused buffer itself, copies 1.. live in `base` -- so one XCD's atomics stay in its own L2;
// lart_gpu_get_tallies / lart_gpu_tally_buffer fold the copies back (k_fold_copies) before anything is read.
struct DRep { double *base; long long off, len; int nrep; };

struct DCtl { unsigned long long next_k; unsigned long long n_active; };
struct DPeelCtl { unsigned long long q_head, d_n; };   // per queue buffer: producer regions peel_kernel has taken; entries of the deposit list
//   // q_n: records in the peel queue of this launch; q_head: (record, observer) items peel_kernel has handed out

// Peel-off event queue (deferred next-event estimator).  The reference marches a second ray to every observer inside each scattering
// (peelingoff_rect.f90:576-690 etc.).  Inlined into the transport kernel that second march doubles its live state (2 waves/SIMD, scratch
// spills).  Here a scattering only RECORDS what the peel routine reads -- position, incoming direction, weight, the atom's frame
// (xfreq_atom, u_x, u_y, u_z), E1..E3, the Stokes triad and Q,U,V, the cell -- as one SoA record (coalesced 512-byte stores per plane
// and wave), and peel_kernel runs the unchanged peel routines over the records afterwards, one record per lane, all lanes busy.
// A mirror plane flips the direction of a packet and leaves the vectors m and n of its Stokes frame as they are (the mirrored quadrant, raytrace_car.f90:2235-2244;
// the octree with mirrored root-box faces, raytrace_amr.f90:176-201): after that (m, n, k) is no right-handed triad any more, and a peel record has to carry n
// instead of leaving it to n = k x m (peel_load)
template <int VAR> __device__ __forceinline__ bool peel_record_has_n(const lart_params &par) { return VAR == 5 || (VAR == 3 && (par.xy_symmetry || par.xyz_symmetry)); }
// one record = 18 doubles (144 B) without a Stokes vector, 28 (224 B) with one, 16-byte aligned, fields paired so that a record is written and read with
// 16-byte accesses at constant offsets from ONE address (round 4; rounds 1-3 kept one plane per field: 21 plane bases in scalar registers, 21 address
// computations and 21 narrow stores per record).  12-13: the cell / leaf / clump indices and the kind as four ints.
enum { PQ_X = 0, PQ_Y, PQ_Z, PQ_KX, PQ_KY, PQ_KZ, PQ_WGT, PQ_XA, PQ_UX, PQ_UY, PQ_UZ, PQ_GREC, PQ_INTS, PQ_INTS2, PQ_E1, PQ_E2, PQ_E3, PQ_PAD, PQ_NOSTOKES,
       PQ_MX = PQ_NOSTOKES, PQ_MY, PQ_MZ, PQ_Q, PQ_U, PQ_V, PQ_NX, PQ_NY, PQ_NZ, PQ_PAD2, PQ_COUNT };
static_assert(PQ_NOSTOKES % 2 == 0 && PQ_COUNT % 2 == 0 && PQ_INTS % 2 == 0, "peel records are read and written in 16-byte pieces");
__host__ __device__ constexpr unsigned pq_recd(bool stokes) { return stokes ? (unsigned)PQ_COUNT : (unsigned)PQ_NOSTOKES; }      // doubles per record
enum { PEEL_DIRECT = 0, PEEL_RESONANCE = 1, PEEL_DUST = 2 };
// Every producer wave owns one region of `region` consecutive record slots (64 lanes x (events of the launch + 1)) and fills it from the
// front: the slot counter is a per-wave word in LDS, so queueing costs no global atomic (one counter for the whole grid serialises in
// L2: measured +6 ms on a 9 ms launch); count[w] = records wave w wrote.  peel_kernel takes whole regions from the queue head.
// scalar 24 of the fused tally buffer: deposit-list entries that did not fit (lart_gpu_get_tallies refuses the run; scal[7] likewise for the overlap lists)
#define LART_SCAL_PEEL_DROPPED 24
struct DPeelQ { double *f; unsigned cap; unsigned region, nregions; unsigned *count;
                unsigned *dep_idx; double *dep_tau; unsigned long long dep_cap; DPeelCtl *pctl; };   // deposit list: (record, observer) and the optical depth of the rays that came through (peel_kernel.hpp)   // f: cap records of pq_recd(stokes) doubless (ic, jc, kc | owner clump, kind)

// resonance + fluorescence multiplets (line types 4, 5, 6: lart_line_multiplet).  The per-level constants are wave-uniform and ride
// in the kernel arguments; the per-branch tables are gathered with lane-varying (iup, idown) from global memory:
// tab[k*12 + iup*4 + idown], k = 0 Elow_Hz, 1 E1, 2 E2, 3 E3, 4 P_down;  itab[0..2] = ndown, itab[3 + iup*4 + idown] = A_down
struct DMulti { int nup, coherent; double delE_Hz[3], damping_up[3], f12[3]; const double *tab; const int *itab; };

struct KArgs {
    lart_params par;
    DGrid g;
    DTally t;
    DPool pool;
    DMueller mu;
    const DObs *obs; int nobs, nxim, nyim; double dxim, dyim;
    int heal_nside;               // > 0: HEALPix inside observers (nxim = npix, nyim = 1)
    DJPa jpa;
    DAmr amr;
    DClumps cl;
    DRep rep;
    DCtl *ctl;
    long long first_id, stride, count;
    unsigned long long seed;
    int max_events;
    double *log_nscatt, *log_xfreq;
    double *allph; long long allph_n;   // par%save_all_photons: 10 columns of allph_n records (r, xfreq1, xfreq2, nscatt_gas, nscatt_dust, I, Q, U, V, r0), nullptr = off
    DMulti ml;
    DPeelQ pq;
    // per-event history probe of the parity tests (lart_gpu_test_trace): trace_slot[k] >= 0 selects photon ordinal k; record i of slot s
    // is trace[(s * trace_cap + i) * LART_TRACE_W ..]; trace_n[s] counts the records (also those beyond trace_cap, which are not stored)
    const int *trace_slot; double *trace; int *trace_n; int trace_cap;
    const double *h2tab;          // the H2 line table (see h2_kappa_D), nullptr without H2
};
#define LART_TRACE_W 8

// ---------------------------------------------------------------------------------------------
// LDS image: Voigt tables (606 f64) followed by the three face arrays
// ---------------------------------------------------------------------------------------------
static __device__ const double c_voigt_tab[606] = {   // H0(101) | H1(202) | H2(101) | H3(202)
#include "voigt_tables.inc"
};
#define VT_H0 0
#define VT_H1 101
#define VT_H2 303
#define VT_H3 404
#define VT_END 606

struct Lds { const double *vt; const double *xf, *yf, *zf;   // Voigt tables and cell faces staged in LDS
             double dip_p2, dip_qs;      // Cardano constants of the dipole phase function for par.E1 > 0 (rand_resonance): p, 1/(E1 p^3)
             double *jp_P, *jp_Pn;       // per-block LDS copies of P1 / P1_new (nullptr: deposits go to global memory)
             double *jp_J; int j_lo, j_w; // per-block LDS copy of the line-centre window of J1: [j_w][nbins] (nullptr: off)
             unsigned *pq_cnt;           // this wave's record counter of the peel queue (LDS word), nullptr = no peel-off
             double *aux;                // line type 8 / H2: the weight budget of the block (8 doubles in LDS, flushed once per launch), nullptr: add to the global scalars
             const ClumpGeo *ct_geo; const ClumpPhys *ct_phys;   // VAR 7, populations of at most 128 clumps: the clump records staged in LDS (nullptr: read from global memory)
             const DObs *obs;            // peel prefilter: the observers' records staged in LDS (nullptr: read from global memory)
             double fov_bx, fov_by;      // peel prefilter: tan of the half field of view (+ margin) in x and y, < 0 = no field-of-view test (lds_set_fov)
             unsigned char *coop;        // LART_COOP_BYTES per wave: scratch of the cooperative samplers (vz_core_coop / vz_wing_coop), nullptr = off
             int lt; double dnu_hz;      // line type (1 single line, 2 Ly-alpha doublet, 7 H + D) and DnuHK_Hz / delta_nu_HD_Hz: set from the kernel's template
             int ml_nup, ml_coh; double ml_dE[3], ml_damp[3], ml_f12[3]; const double *ml_tab; const int *ml_itab;   // lt 4 = the multiplets (line types 4, 5, 6): DMulti
             double hd_ratio, hd_ratio_a, hd_nD; };   // line type 7: ratio_Dfreq_HD, ratio_voigta_HD, nD_over_nH (line_mod.f90:1186-1191)
                                         // (lt comes from a template parameter, so that `l.lt == 2` / `== 7` fold away in the single-line kernels)

// one term of the weight budget (W_conv, W_esc1, ...; W_H2abs, W_H2scat, W_H2pump): into the block's LDS accumulators when the kernel set them up -- every lane of a
// launch adding to ONE global address serialises at the memory side --, else (probes) into the global scalars
struct KArgs;
__device__ __forceinline__ void aux_add(double *scal, const Lds &l, int k, double w) { if (l.aux) atomicAdd(&l.aux[k], w); else atomicAdd(&scal[25 + k], w); }
__device__ inline Lds lds_stage(double *sh, const DGrid &g) {
    for (int i = threadIdx.x; i < VT_END; i += blockDim.x) sh[i] = c_voigt_tab[i];
    int nf = g.nx + g.ny + g.nz + 3;
    for (int i = threadIdx.x; i < nf; i += blockDim.x) sh[VT_END + i] = g.faces[i];
    __syncthreads();
    Lds l; l.vt = sh; l.xf = sh + VT_END; l.yf = l.xf + (g.nx + 1); l.zf = l.yf + (g.ny + 1);
    l.dip_p2 = 0.0; l.dip_qs = 0.0; l.jp_P = nullptr; l.jp_Pn = nullptr; l.jp_J = nullptr; l.j_lo = 0; l.j_w = 0; l.coop = nullptr; l.pq_cnt = nullptr; l.fov_bx = -1.0; l.fov_by = -1.0; l.obs = nullptr; l.ct_geo = nullptr; l.ct_phys = nullptr; l.aux = nullptr; l.lt = 1; l.dnu_hz = 0.0; l.hd_ratio = 1.0; l.hd_ratio_a = 1.0; l.hd_nD = 0.0; l.ml_nup = 1; l.ml_coh = 0; l.ml_tab = nullptr; l.ml_itab = nullptr;
    for (int i = 0; i < 3; i++) { l.ml_dE[i] = 0.0; l.ml_damp[i] = 1.0; l.ml_f12[i] = 0.0; }
    return l;
}

// ---------------------------------------------------------------------------------------------
// elementary functions on the arguments this path produces (lart_math.hpp); -DLART_OCML_MATH = device library versions
// ---------------------------------------------------------------------------------------------
#ifdef LART_OCML_MATH
__device__ inline double d_log(double x) { return log(x); }
__device__ inline double d_tan(double x) { return tan(x); }
__device__ inline double d_exp_neg(double x) { return exp(x); }
__device__ inline void d_sincos_2pi(double u, double *s, double *c) { sincos(LART_TWOPI * u, s, c); }
#else
__device__ inline double d_log(double x) { return lartm::log_pos(x); }
__device__ inline double d_tan(double x) { return lartm::tan_halfpi(x); }
__device__ inline double d_exp_neg(double x) { return lartm::exp_neg(x); }
__device__ inline void d_sincos_2pi(double u, double *s, double *c) { lartm::sincos_2pi(u, s, c); }
#endif

// ---------------------------------------------------------------------------------------------
// RNG
// ---------------------------------------------------------------------------------------------
// 32x32 -> 64 bit product in one quarter-rate instruction (the compiler emits v_mul_hi_u32 + v_mul_lo_u32)
__device__ inline void mulhilo32(uint32_t k, uint32_t v, uint32_t &hi, uint32_t &lo) {
#ifdef LART_PHILOX_PLAIN
    hi = __umulhi(k, v); lo = k * v;
#else
    uint64_t p;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "s"(k), "v"(v) : "vcc");
    lo = (uint32_t)p; hi = (uint32_t)(p >> 32);
#endif
}
// Philox4x32-10, key = seed, counter = (block, photon id): draw n of a packet is half (n & 1) of block n >> 1, so any
// lane can produce any draw of its packet without state.  The photon loop uses that to take the generator out of the
// divergent code: at the top of every event round ALL lanes of the wave evaluate LART_RNG_TOPUP Philox blocks (full SIMD
// utilisation, independent blocks interleaved for latency hiding) and append them to a per-lane ring of 16 draws in LDS;
// the rejection loops downstream read the ring.  A lane that runs its ring dry continues with single blocks (`slow`
// path); next2() needs one block evaluation for two draws whatever the parity.
#ifndef LART_RNG_BUF
#define LART_RNG_BUF 16
#endif
// blocks per fill point of the event loop (Rng::fill): before the flight / between the vz sampler and scatter_post; by kernel family (tools/gpu_job.sh perflib sweeps,
// profiles/r04_rng_fill.md).  LART_RNG_TOPUP is the round-1 name of the single fill and still sets the top fill of a build that defines it.
#ifdef LART_RNG_TOPUP
#define LART_FILL_TOP(S) LART_RNG_TOPUP
#define LART_FILL_MID(S) 0
#else
#ifndef LART_FILL_TOP_S
#define LART_FILL_TOP_S 4
#endif
#ifndef LART_FILL_MID_S
#define LART_FILL_MID_S 4
#endif
#ifndef LART_FILL_TOP_N
#define LART_FILL_TOP_N 2
#endif
#ifndef LART_FILL_MID_N
#define LART_FILL_MID_N 3
#endif
#define LART_FILL_TOP(S) ((S) ? LART_FILL_TOP_S : LART_FILL_TOP_N)
#define LART_FILL_MID(S) ((S) ? LART_FILL_MID_S : LART_FILL_MID_N)
#endif
#ifndef LART_PHILOX_ROUNDS
#define LART_PHILOX_ROUNDS 10      // Philox4x32-10 (the KAT-pinned default of Salmon et al. 2011); tuning builds only change this to measure
#endif
// orders LDS exchanges between the lanes of ONE wave (its LDS pipeline is in order: this only constrains the compiler)
#define LART_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define LART_COOP_BYTES 192          // wave-local LDS scratch of the cooperative samplers

struct Rng {
    uint32_t k0, k1; uint64_t id, n; double spare;          // spare: second half of the last slow block, valid while n is odd
    // ring of LART_RNG_BUF (= 16) draws in LDS: draw i of the packet lives in slot i & 15 of this lane's column;
    // `have` draws are buffered ahead of n (n + have is even while have > 0); have == 0: slow path until the next prefill()
    double *lbuf = nullptr; uint32_t stride = 0, have = 0;
#ifdef LART_RNG_STATS                                       // tuning builds: single-block evaluations off the ring (per lane) and how often the WAVE takes that detour
    uint32_t nslow = 0, nslow_w = 0;
#define LART_RNG_SLOW() do { nslow++; if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) nslow_w++; } while (0)
#else
#define LART_RNG_SLOW() do { } while (0)
#endif
    __device__ inline void init(uint64_t seed, uint64_t pid, uint64_t n0) {
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32); id = pid; n = n0 & ~1ULL; spare = 0.0;
        have = 0;
        if (n0 & 1ULL) (void)next();
    }
    __device__ inline void attach(double *lds_lane, uint32_t lds_stride) { lbuf = lds_lane; stride = lds_stride; have = 0; }
    // open-interval uniform: (x>>12 + 0.5) / 2^52  (random_mt.f90:627-629)
    static __device__ inline double open01(uint32_t lo, uint32_t hi) {
        uint64_t x = ((uint64_t)hi << 32) | lo;
        return ((double)(x >> 12) + 0.5) * (1.0 / 4503599627370496.0);
    }
    __device__ inline void block(uint64_t blk, double &first, double &second) const {
        uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32), c2 = (uint32_t)id, c3 = (uint32_t)(id >> 32);
        uint32_t a = k0, b = k1;
        // rolled on purpose: inlined at every slow-path call site, and the ten rounds are one dependency chain anyway
#ifdef LART_PHILOX_PLAIN
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int r = 0; r < LART_PHILOX_ROUNDS; r++) {
            uint32_t h0, l0, h1, l1;
            mulhilo32(0xD2511F53u, c0, h0, l0); mulhilo32(0xCD9E8D57u, c2, h1, l1);
            uint32_t n0 = h1 ^ c1 ^ a, n2 = h0 ^ c3 ^ b;
            c0 = n0; c1 = l1; c2 = n2; c3 = l0;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        first = open01(c0, c1); second = open01(c2, c3);
    }
    // Blocks are named scalars (not arrays, which end up in scratch) and advance together round by round.
    struct Ctr { uint32_t c0, c1, c2, c3; };
    static __device__ inline void round1(Ctr &c, uint32_t a, uint32_t b) {
        uint32_t h0, l0, h1, l1;
        mulhilo32(0xD2511F53u, c.c0, h0, l0); mulhilo32(0xCD9E8D57u, c.c2, h1, l1);
        uint32_t n0 = h1 ^ c.c1 ^ a, n2 = h0 ^ c.c3 ^ b;
        c.c0 = n0; c.c1 = l1; c.c2 = n2; c.c3 = l0;
    }
    __device__ inline Ctr ctr_of(uint64_t blk) const { Ctr c; c.c0 = (uint32_t)blk; c.c1 = (uint32_t)(blk >> 32); c.c2 = (uint32_t)id; c.c3 = (uint32_t)(id >> 32); return c; }
    __device__ inline void store2(uint32_t slot, bool on, const Ctr &c) {
        if (on) { lbuf[(slot & (LART_RNG_BUF - 1)) * stride] = open01(c.c0, c.c1); lbuf[((slot + 1) & (LART_RNG_BUF - 1)) * stride] = open01(c.c2, c.c3); }
    }
    // Top the ring up with K blocks (2 draws each) behind the draws still buffered.  Every lane evaluates the same number of blocks; a lane whose
    // ring is nearly full stores only what fits (those counters are simply produced again later).  The event loop fills twice per round (round 4):
    // LART_FILL_TOP blocks before the flight (tau draw, vz sampler) and LART_FILL_MID blocks between the vz sampler and scatter_post (polar angle,
    // azimuth, perpendicular velocity).  One fill of 5 blocks per round, the round-1 scheme, leaves a lane whose packet sits in the line wing with
    // a Stokes vector (11-12 draws per event) with a dry ring on EVERY round of the excursion: it then evaluates single blocks inside the divergent
    // code, ~100 instructions at a handful of lanes each time -- counted with -DLART_RNG_STATS: 10.4 such detours per wave-round in the Stokes
    // kernels, 1.5 in the headline kernel (profiles/r04_rng_fill.md).
    template <int K>
    __device__ inline void prefill() {
        static_assert(LART_RNG_BUF == 16, "the ring indexes draws by i & 15");
        static_assert(K >= 1 && K <= 8, "1..8 blocks per fill");
        uint64_t e = have ? n + have : (n & ~1ULL);          // first draw not buffered (even); nothing buffered: restart on draw n's block
        uint32_t used = (uint32_t)(e - (n & ~1ULL)), freeb = (LART_RNG_BUF - used) >> 1, slot = (uint32_t)e;
        uint64_t b0 = e >> 1;
        Ctr q0 = ctr_of(b0), q1 = ctr_of(b0 + 1), q2 = ctr_of(b0 + 2), q3 = ctr_of(b0 + 3), q4 = ctr_of(b0 + 4), q5 = ctr_of(b0 + 5), q6 = ctr_of(b0 + 6), q7 = ctr_of(b0 + 7);
        uint32_t a = k0, b = k1;
#pragma unroll 1
        for (int r = 0; r < LART_PHILOX_ROUNDS; r++) {
            round1(q0, a, b);
            if (K >= 2) round1(q1, a, b);
            if (K >= 3) round1(q2, a, b);
            if (K >= 4) round1(q3, a, b);
            if (K >= 5) round1(q4, a, b);
            if (K >= 6) round1(q5, a, b);
            if (K >= 7) round1(q6, a, b);
            if (K >= 8) round1(q7, a, b);
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        store2(slot, freeb > 0, q0);
        if (K >= 2) store2(slot + 2, freeb > 1, q1);
        if (K >= 3) store2(slot + 4, freeb > 2, q2);
        if (K >= 4) store2(slot + 6, freeb > 3, q3);
        if (K >= 5) store2(slot + 8, freeb > 4, q4);
        if (K >= 6) store2(slot + 10, freeb > 5, q5);
        if (K >= 7) store2(slot + 12, freeb > 6, q6);
        if (K >= 8) store2(slot + 14, freeb > 7, q7);
        uint32_t stored = freeb < (uint32_t)K ? freeb : (uint32_t)K;
        have = (uint32_t)(e + 2 * stored - n);
    }
    // The same top-up for a wave that holds few packets (the drain phase: DESIGN.md 3.4).  With at most 64 / K packet lanes in the wave,
    // lane j evaluates block (j mod K) of the (j div K)-th packet lane and stores it into that lane's ring: one block per lane instead of
    // K blocks per packet lane, i.e. a fraction of the latency on a wave that has its SIMD to itself (the generator is 39 % of a lone packet's event:
    // profiles/r03_lone_wave_sections.txt).  Same counters, same slots, same draws as prefill<K>().  Call with the whole wave; `act` = this lane holds a packet.
    template <int K>
    __device__ inline void prefill_sparse(bool act, unsigned long long mask) {
        const int lane = threadIdx.x & 63;
        unsigned long long e = 0; uint32_t freeb = 0;
        if (act) { e = have ? n + have : (n & ~1ULL); const uint32_t used = (uint32_t)(e - (n & ~1ULL)); freeb = (LART_RNG_BUF - used) >> 1; }
        const int t = lane / K, b = lane - t * K;
        unsigned long long mk = mask;
#pragma unroll
        for (int q = 0; q < 64 / K; q++) if (q < t) mk &= mk - 1;       // drop the t lowest packet lanes: the lowest one left is this helper's target
        const bool work = mk != 0 && t < 64 / K;
        const int src = work ? __ffsll((long long)mk) - 1 : lane;
        const unsigned long long sid = __shfl((unsigned long long)id, src, 64), se = __shfl(e, src, 64);
        const uint32_t sfree = __shfl(freeb, src, 64);
        if (work && (uint32_t)b < sfree) {
            const unsigned long long blk = (se >> 1) + (unsigned long long)b;
            uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32), c2 = (uint32_t)sid, c3 = (uint32_t)(sid >> 32), ka = k0, kb = k1;
#pragma unroll 1
            for (int r = 0; r < LART_PHILOX_ROUNDS; r++) {
                uint32_t h0, l0, h1, l1;
                mulhilo32(0xD2511F53u, c0, h0, l0); mulhilo32(0xCD9E8D57u, c2, h1, l1);
                const uint32_t n0 = h1 ^ c1 ^ ka, n2 = h0 ^ c3 ^ kb;
                c0 = n0; c1 = l1; c2 = n2; c3 = l0;
                ka += 0x9E3779B9u; kb += 0xBB67AE85u;
            }
            double *col = lbuf + (src - lane);                 // the target lane's column of the ring
            const uint32_t slot = (uint32_t)se + 2u * (uint32_t)b;
            col[(slot & (LART_RNG_BUF - 1)) * stride] = open01(c0, c1); col[((slot + 1) & (LART_RNG_BUF - 1)) * stride] = open01(c2, c3);
        }
        LART_WAVE_SYNC();
        if (act) { const uint32_t stored = freeb < (uint32_t)K ? freeb : (uint32_t)K; have = (uint32_t)(e + 2 * stored - n); }
    }
    // one fill point of the event loop, called by the whole wave: `act` = this lane holds a packet that draws from its ring in the stage that follows
    template <int K>
    __device__ inline void fill(bool act) {
        if (K <= 0 || lbuf == nullptr) return;                // (the single-event probes of the parity tests run without a ring: every draw is a single block there)
        const unsigned long long am = __ballot(act);
        if (!am) return;
        if (__popcll(am) * (K > 0 ? K : 1) <= 64) prefill_sparse<(K > 0 ? K : 1)>(act, am);
        else if (act) prefill<(K > 0 ? K : 1)>();
    }
    __device__ inline double next() {
        if (have) { double v = lbuf[((uint32_t)n & (LART_RNG_BUF - 1)) * stride]; n++; have--; return v; }
        uint64_t m = n++;
        if (m & 1ULL) return spare;
        LART_RNG_SLOW();
        double first; block(m >> 1, first, spare);
        return first;
    }
    // two consecutive draws; off the ring this is ONE block evaluation for either parity of n
    __device__ inline void next2(double &u1, double &u2) {
        if (have >= 2) {
            u1 = lbuf[((uint32_t)n & (LART_RNG_BUF - 1)) * stride]; u2 = lbuf[(((uint32_t)n + 1) & (LART_RNG_BUF - 1)) * stride];
            n += 2; have -= 2; return;
        }
        bool odd = n & 1ULL;                                  // one buffered draw left <=> n odd (n + have is even)
        double prev = have ? lbuf[((uint32_t)n & (LART_RNG_BUF - 1)) * stride] : spare, first, second;
        LART_RNG_SLOW();
        block((n + 1) >> 1, first, second);
        if (odd) { u1 = prev; u2 = first; spare = second; } else { u1 = first; u2 = second; }
        have = 0; n += 2;
    }
    // Marsaglia polar pair (random_mt.f90:964-988); first = v2*rsq (what the reference returns first)
    __device__ inline double gauss_pair(double &second) {
        double v1, v2, rsq;
        do { double u1, u2; next2(u1, u2); v1 = 2.0 * u1 - 1.0; v2 = 2.0 * u2 - 1.0; rsq = v1 * v1 + v2 * v2; } while (!(rsq > 0.0 && rsq < 1.0));
        rsq = sqrt(-2.0 * d_log(rsq) / rsq);
        second = v1 * rsq;
        return v2 * rsq;
    }
};

// ---------------------------------------------------------------------------------------------
// Voigt H(x,a): voigt_mod.f90:541-733 (four regimes on a 0.05 grid)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double voigt_H(const double *vt, double vin, double a) {
    double v = vin < 0.0 ? -vin : vin;
    if (v < 1.0) {
        double vv = v * 20.0; int k = (int)vv;                // k = k1-1
        double y1 = vt[VT_H0 + k] + a * (vt[VT_H1 + k] + a * vt[VT_H2 + k]);
        double y2 = vt[VT_H0 + k + 1] + a * (vt[VT_H1 + k + 1] + a * vt[VT_H2 + k + 1]);
        return y1 + (y2 - y1) * (vv - (double)k);
    } else if (v < 5.0) {
        double vv = v * 20.0; int k = (int)vv - 1;            // k = k1-1, k1 = int(vv)
        double y1 = vt[VT_H0 + k] + a * (vt[VT_H1 + k] + a * (vt[VT_H2 + k] + a * vt[VT_H3 + k]));
        double y2 = vt[VT_H0 + k + 1] + a * (vt[VT_H1 + k + 1] + a * (vt[VT_H2 + k + 1] + a * vt[VT_H3 + k + 1]));
        double y3 = vt[VT_H0 + k + 2] + a * (vt[VT_H1 + k + 2] + a * (vt[VT_H2 + k + 2] + a * vt[VT_H3 + k + 2]));
        double u1 = (double)k, u2 = (double)(k + 1), u3 = (double)(k + 2);
        return 0.5 * y1 * (vv - u2) * (vv - u3) - y2 * (vv - u1) * (vv - u3) + 0.5 * y3 * (vv - u1) * (vv - u2);
    } else if (v < 10.0) {
        double vv = v * 20.0; int k = (int)vv; double a2 = a * a;
        double y1 = a * (vt[VT_H1 + k] + a2 * vt[VT_H3 + k]);
        double y2 = a * (vt[VT_H1 + k + 1] + a2 * vt[VT_H3 + k + 1]);
        return y1 + (y2 - y1) * (vv - (double)k);
    }
    double v2 = 1.0 / (v * v);
    return 0.56418958354775628695 * a * v2 * (1.0 + ((1.5 - a * a) + 3.75 * v2) * v2);
}
__device__ inline void lds_set_multiplet(Lds &l, const DMulti &m) {
    l.ml_nup = m.nup; l.ml_coh = m.coherent; l.ml_tab = m.tab; l.ml_itab = m.itab;
    for (int i = 0; i < 3; i++) { l.ml_dE[i] = m.delE_Hz[i]; l.ml_damp[i] = m.damping_up[i]; l.ml_f12[i] = m.f12[i]; }
}
// calc_voigt1 / calc_voigt2 (line_mod.f90:38-62): the profile of a cell with Voigt parameter a and Doppler width Dfreq.
// Line type 2 = the Ly-alpha doublet: 1/3 of the strength at the 2P1/2 level, DnuHK below the 2P3/2 level.
__device__ __forceinline__ double calc_voigt(const Lds &l, double xfreq, double a, double Dfreq) {
    if (l.lt == 2) { const double dnu = l.dnu_hz / Dfreq; return voigt_H(l.vt, xfreq + dnu, a) * (1.0 / 3.0) + voigt_H(l.vt, xfreq, a) * (2.0 / 3.0); }
    if (l.lt == 7) {                                           // calc_voigt_HD :82-106: H profile + the D line seen from the H frame
        const double dx = l.dnu_hz / Dfreq, xD = (xfreq - dx) * l.hd_ratio;
        return voigt_H(l.vt, xfreq, a) + l.hd_nD * l.hd_ratio * voigt_H(l.vt, xD, a * l.hd_ratio_a);
    }
    if (l.lt == 4) {                                           // calc_voigt3 :64-80 (types 5, 6; nup = 1 for type 4 = calc_voigt1)
        double v = voigt_H(l.vt, xfreq, a);
        for (int i = 1; i < l.ml_nup; i++) {
            const double Dnu = l.ml_dE[i] / Dfreq, a_ratio = l.ml_damp[i] / l.ml_damp[0], f_ratio = l.ml_f12[i] / l.ml_f12[0];
            v = v + voigt_H(l.vt, xfreq + Dnu, a * a_ratio) * f_ratio;
        }
        return v;
    }
    return voigt_H(l.vt, xfreq, a);
}

// packet state of one lane (photon_type, define.f90)
struct Photon {
    double x, y, z, kx, ky, kz, xfreq, wgt, nsg, nsd;
    double mx, my, mz, nx, ny, nz, Q, U, V;
    int ic, jc, kc; bool inside;
    int icl;                       // owner clump (VAR 4): 0 = vacuum (photon%icell_clump, define.f90:90)
    int iband;                     // photon%iband (define.f90:96): 1 the line's band, 2 H-alpha after a Ly-beta conversion (line type 8); kept in FLAG_BAND2 between launches
    double shear;                  // photon%vfy_shear (define.f90:100): y velocity picked up at the shearing x boundary, in thermal units (VAR 8)
    double xfreq_ref;
#ifdef LART_PROFILE
    long long pt[16], plast;       // tuning builds only: wave clocks per code section (prof_mark)
    unsigned pc[4];                // wave-level trip counts
#endif
};
// section timers of the tuning build (make SUFFIX=_prof EXTRA=-DLART_PROFILE): the clocks since the previous mark are
// charged to section i.  0 pool load/store, 1 refill + birth, 2 tau draw + forced first scattering, 3 raytrace_to_tau,
// 4 scatter: before the vz sampler, 5 vz sampler |x|<=1 loop, 6 scatter: after the vz sampler, 7 bookkeeping of finished
// packets, 8 vz |x|>1 set-up, 9 vz |x|>1 loop, 10 vz |x|>1 final tangent.  LART_PROF_TRIP counts how often the WAVE runs a
// loop body (first active lane counts): 0 |x|<=1 tries, 1 |x|>1 tries, 2 event rounds, 3 rounds that entered |x|>1
#ifdef LART_PROFILE
#define LART_PROF_MARK(p, i) do { long long t_ = clock64(); (p).pt[i] += t_ - (p).plast; (p).plast = t_; } while (0)
#define LART_PROF_TRIP(p, i) do { if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) (p).pc[i]++; } while (0)
#define LART_PROF_ARG(p) , Photon &p
#define LART_PROF_PASS(p) , p
#else
#define LART_PROF_MARK(p, i) do { } while (0)
#define LART_PROF_TRIP(p, i) do { } while (0)
#define LART_PROF_ARG(p)
#define LART_PROF_PASS(p)
#endif

// ---------------------------------------------------------------------------------------------
// samplers
// ---------------------------------------------------------------------------------------------
// The |x| > 1 sampler with its rejection loop run by the whole wave (see vz_core_coop).  Few lanes are in this branch at a
// time (~16 % on the headline workload) and each needs ~1.15 tries, but the wave used to issue 2.4 passes until the last
// of them was done.  Here the branch lanes compute their comparison function (stage A, as before), then every lane of the
// wave evaluates one try of one of them: helper j of a target reads the target's draws n + d j .. (d = 2 or 3 draws per try)
// from its LDS ring and evaluates exactly that try; the first accepted try wins (stage B); the winner's window (pb, delt)
// goes back to the target, which draws the velocity inside it (stage C).  Same tries, same order, same draws.
__device__ inline double vz_wing_coop(Rng &g, bool mine, double x0in, double a, unsigned char *wl) {
    const double xc = 2.414213562373095048801688724209698, two_over_PI = 2.0 / LART_PI;
    const unsigned lane = __lane_id();
    const double x0 = fabs(x0in);
    // --- stage A: the comparison function in beta-space (random_mt.f90:2588-2640), wing lanes only
    double beta0 = 0.0, dbeta = 0.0, beta1 = 0.0, hA = 0.0, hB = 0.0, thr0 = 0.0, thr1 = 0.0;
    int mode = 0;                                 // 0: single segment; 1: three segments; 2: two segments (dbeta = 0)
    if (mine) {
        const double x0sq = x0 * x0;
        beta0 = d_exp_neg(-x0sq / 2.0);
        const double h0_two = beta0 / a, h0 = h0_two / 2.0;
        double S0 = beta0 * h0, S1 = 0.0, S2;
        beta1 = beta0; hA = h0; mode = 1;
        if (x0 < xc) {
            dbeta = sqrt(two_over_PI * a * (1.0 - beta0) * beta0 * x0); beta1 = beta0 + dbeta;
            double pb1 = sqrt(-2.0 * d_log(beta1));
            hB = two_over_PI * beta1 * pb1 / (x0sq - pb1 * pb1);
            S1 = dbeta * h0; S2 = (1.0 - beta1) * hB;
        } else {
            double h2 = 0.3861 / (x0sq - 1.373);
            if (h0_two < h2) { mode = 0; hB = h2; S2 = 1.0; }
            else if (h0 < h2) { mode = 2; hB = h2; S2 = (1.0 - beta0) * h2; }
            else {
                dbeta = sqrt(two_over_PI * a * (1.0 - beta0) * beta0 * x0); beta1 = beta0 + dbeta;
                double pb1 = sqrt(-2.0 * d_log(beta1));
                double h1 = two_over_PI * beta1 * pb1 / (x0sq - pb1 * pb1);
                hB = h1 > h2 ? h1 : h2;
                S1 = dbeta * h0; S2 = (1.0 - beta1) * hB;
            }
        }
        const double Stot = S0 + S1 + S2;
        thr0 = S0 / Stot; thr1 = 1.0 - S2 / Stot;
    }
    // --- stage B: the tries, by everybody
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    lds_u32 *key = (lds_u32 *)wl; lds_u8 *tgt = (lds_u8 *)wl + 128;
    bool pending = mine;
    double pb = 0.0, delt = 0.0;
    for (;;) {
        const unsigned long long F = __ballot(pending);
        if (F == 0ULL) break;
        const unsigned dper = mode == 0 ? 2u : 3u;            // draws per try of this lane
        const bool helpable = pending && g.have >= dper;
        const unsigned long long H = __ballot(helpable), M = __ballot(1);
        const int nH = __popcll(H), nM = __popcll(M);
        const bool solo_pass = (nH == 0 || nH > 32);          // dry rings, or no room in the slot table: everybody for itself
        if (pending && (solo_pass || !helpable)) {            // one classical try (same arithmetic as below)
            double beta, Cb, rS = 2.0, u, acc;
            if (mode == 0) g.next2(u, acc); else { g.next2(rS, u); acc = g.next(); }
            if (mode == 0) { beta = u; Cb = hB; }
            else if (rS < thr0)      { beta = beta0 * sqrt(u); Cb = beta / a; }
            else if (mode == 2)      { beta = beta0 + (1.0 - beta0) * u; Cb = hB; }
            else if (rS < thr1)      { beta = beta0 + dbeta * u; Cb = hA; }
            else                     { beta = beta1 + (1.0 - beta1) * u; Cb = hB; }
            pb = sqrt(-2.0 * d_log(beta));
            const double den = fma(x0 - pb, x0 + pb, a * a);
            delt = atan((2.0 * pb * a) / den); if (den < 0.0) delt = delt + LART_PI;
            if (acc * Cb < (beta / (a * LART_PI)) * delt) pending = false;
        }
        if (solo_pass) continue;
        const unsigned long long below = (1ULL << lane) - 1ULL;
        const int rs = __popcll(H & below);
        int K = nM / nH; if (K > LART_RNG_BUF / 2) K = LART_RNG_BUF / 2;
        const int r = __popcll(M & below), j = r / nH, s = r - j * nH;
        if (helpable) { tgt[rs] = (unsigned char)lane; key[rs] = 0xFFFFFFFFu; }
        LART_WAVE_SYNC();
        const int T = tgt[s];
        // the target's state.  Cross-lane reads only at converged points (a shuffle from a lane that is masked off reads 0):
        // what only segmented comparison functions need is fetched under wave-uniform conditions
        const unsigned nT = __shfl((unsigned)g.n, T), haveT = __shfl(g.have, T); const int modeT = __shfl(mode, T);
        const double x0T = __shfl(x0, T), aT = __shfl(a, T), hBT = __shfl(hB, T);
        double beta0T = 0.0, thr0T = 0.0, thr1T = 0.0, beta1T = 0.0, dbetaT = 0.0, hAT = 0.0;
        if (__ballot(helpable && mode != 0) != 0ULL) { beta0T = __shfl(beta0, T); thr0T = __shfl(thr0, T); }
        if (__ballot(helpable && mode == 1) != 0ULL) { thr1T = __shfl(thr1, T); beta1T = __shfl(beta1, T); dbetaT = __shfl(dbeta, T); hAT = __shfl(hA, T); }
        const unsigned dT = modeT == 0 ? 2u : 3u;
        double pbh = 0.0, delth = 0.0;
        if (j < min(K, (int)(haveT / dT))) {
            const double *col = g.lbuf + (T - (int)lane);
            const unsigned i0 = nT + dT * (unsigned)j;
            const double d0 = col[(i0 & (LART_RNG_BUF - 1)) * g.stride], d1 = col[((i0 + 1u) & (LART_RNG_BUF - 1)) * g.stride];
            double beta, Cb, acc;
            if (modeT == 0) { beta = d0; Cb = hBT; acc = d1; }
            else {
                acc = col[((i0 + 2u) & (LART_RNG_BUF - 1)) * g.stride];
                if (d0 < thr0T)       { beta = beta0T * sqrt(d1); Cb = beta / aT; }
                else if (modeT == 2)  { beta = beta0T + (1.0 - beta0T) * d1; Cb = hBT; }
                else if (d0 < thr1T)  { beta = beta0T + dbetaT * d1; Cb = hAT; }
                else                  { beta = beta1T + (1.0 - beta1T) * d1; Cb = hBT; }
            }
            pbh = sqrt(-2.0 * d_log(beta));
            const double den = fma(x0T - pbh, x0T + pbh, aT * aT);
            delth = atan((2.0 * pbh * aT) / den); if (den < 0.0) delth = delth + LART_PI;
            if (acc * Cb < (beta / (aT * LART_PI)) * delth) __hip_atomic_fetch_min(&key[s], ((unsigned)j << 8) | lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        LART_WAVE_SYNC();
        const unsigned kb = helpable ? key[rs] : 0xFFFFFFFFu;
        const int src = kb != 0xFFFFFFFFu ? (int)(kb & 63u) : (int)lane;
        const double pbw = __shfl(pbh, src), deltw = __shfl(delth, src);
        if (helpable) {
            const unsigned tried = (unsigned)min(K, (int)(g.have / dper));
            const unsigned used = dper * (kb != 0xFFFFFFFFu ? (kb >> 8) + 1u : tried);
            g.n += used; g.have -= used;
            if (kb != 0xFFFFFFFFu) { pb = pbw; delt = deltw; pending = false; }
        }
    }
    // --- stage C: the velocity inside the accepted window (:2683-2693)
    double vz = 0.0;
    if (mine) {
        const double t1 = atan((-pb - x0) / a);
        vz = x0 + a * d_tan(delt * g.next() + t1);
    }
    return x0in < 0.0 ? -vz : vz;
}
// atom velocity parallel to k: random_mt.f90:2562-2696 (rand_resonance_vz_seon)
// |x| <= 1: Lorentzian proposal, Gaussian rejection (random_mt.f90:2579-2586)
__device__ inline double vz_core(Rng &g, double x0in, double a LART_PROF_ARG(pp)) {
    double x0 = fabs(x0in), vz;
    for (;;) {
        LART_PROF_TRIP(pp, 0);
        double u1, u2; g.next2(u1, u2);
        vz = x0 + a * d_tan(LART_PI * (u1 - 0.5));
#ifdef LART_EXPERIMENT_DOUBLE_TRY                              // timing experiment: the arithmetic of a try done twice, same result
        double vzb = x0 + a * d_tan(LART_PI * (u2 - 0.5));
        if (u2 <= d_exp_neg(-(vz * vz)) + 0.0 * d_exp_neg(-(vzb * vzb))) break;
#else
        if (u2 <= d_exp_neg(-(vz * vz))) break;
#endif
    }
    return x0in < 0.0 ? -vz : vz;
}
// The |x| <= 1 rejection loop, run by the whole wave together.  A lane accepts a try with probability ~3/4, so after the first
// try about a quarter of the lanes are still pending while the wave would keep issuing tries until its unluckiest lane is
// done (5.3 passes per event on the headline workload for 1.34 tries per lane).  Here the lanes that are done (and the lanes
// that are not in this branch at all) evaluate the NEXT tries of the pending lanes: with nF pending lanes among nM present,
// each pending lane gets K = nM / nF helpers; helper j reads draws (n + 2j, n + 2j + 1) of its target from the target's draw
// ring in LDS and evaluates exactly the arithmetic of that try; the first accepted j wins, the target advances its counter
// by 2 (j + 1).  Same tries, same order, same draws as the sequential loop -> identical results, ~2 passes instead of ~5.
// Call with every lane of the wave that is inside the event loop (mine = this lane needs a core velocity).
// wl: 192 bytes of wave-local LDS (the headline configuration has < 1 KB of LDS to spare per workgroup at 4 workgroups per
// CU): key[32] u32 = (first accepted try << 8) | lane that evaluated it, per slot; tgt[32] u8 = target lane of the slot.
// With more than half of the lanes pending there is no helper to spare and everybody just takes its own next try.
__device__ inline double vz_core_coop(Rng &g, bool mine, double x0in, double a, unsigned char *wl) {
    const unsigned lane = __lane_id();
    const double x0 = fabs(x0in);
    double vz = 0.0;
    bool pending = mine;
    if (pending) {                                            // first try: every lane for itself
        double u1, u2; g.next2(u1, u2);
        vz = x0 + a * d_tan(LART_PI * (u1 - 0.5));
        pending = !(u2 <= d_exp_neg(-(vz * vz)));
    }
    // LDS address space spelled out (ds_* instead of flat accesses); the exchanges between lanes are ordered by
    // wavefront-scope fences, which cost no instruction: the LDS pipeline of a wave is in order
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    lds_u32 *key = (lds_u32 *)wl; lds_u8 *tgt = (lds_u8 *)wl + 128;
    for (;;) {
        const unsigned long long F = __ballot(pending);
        if (F == 0ULL) break;
        const bool helpable = pending && g.have >= 2;         // the draws of its next try are in the ring
        const unsigned long long H = __ballot(helpable), M = __ballot(1);
        const int nH = __popcll(H), nM = __popcll(M);
        if (2 * nH > nM || nH == 0) {                         // no helpers to spare (or dry rings): one classical pass
            if (pending) {
                double u1, u2; g.next2(u1, u2);
                vz = x0 + a * d_tan(LART_PI * (u1 - 0.5));
                pending = !(u2 <= d_exp_neg(-(vz * vz)));
            }
            continue;
        }
        if (H != F && pending && !helpable) {                 // ring run dry (rare): that lane takes one try by itself
            double u1, u2; g.next2(u1, u2);
            vz = x0 + a * d_tan(LART_PI * (u1 - 0.5));
            pending = !(u2 <= d_exp_neg(-(vz * vz)));
        }
        const unsigned long long below = (1ULL << lane) - 1ULL;
        const int rs = __popcll(H & below);                   // my slot as a target (nH <= 32)
        int K = nM / nH; if (K > LART_RNG_BUF / 2) K = LART_RNG_BUF / 2;
        const int r = __popcll(M & below), j = r / nH, s = r - j * nH;    // the slot I help and the try I evaluate for it
        if (helpable) { tgt[rs] = (unsigned char)lane; key[rs] = 0xFFFFFFFFu; }
        LART_WAVE_SYNC();
        const int T = tgt[s];
        const double x0T = __shfl(x0, T), aT = __shfl(a, T);
        const unsigned nT = __shfl((unsigned)g.n, T), haveT = __shfl(g.have, T);
        double vzh = 0.0;
        if (j < min(K, (int)(haveT >> 1))) {
            const double *col = g.lbuf + (T - (int)lane);                      // the target's column of the draw ring
            const double u1 = col[((nT + 2u * j) & (LART_RNG_BUF - 1)) * g.stride], u2 = col[((nT + 2u * j + 1u) & (LART_RNG_BUF - 1)) * g.stride];
            vzh = x0T + aT * d_tan(LART_PI * (u1 - 0.5));
            if (u2 <= d_exp_neg(-(vzh * vzh))) __hip_atomic_fetch_min(&key[s], ((unsigned)j << 8) | lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // lowest accepted try wins
        }
        LART_WAVE_SYNC();
        const unsigned kb = helpable ? key[rs] : 0xFFFFFFFFu;
        const double vzw = __shfl(vzh, kb != 0xFFFFFFFFu ? (int)(kb & 63u) : (int)lane);
        if (helpable) {                                       // my own slot: tries 0 .. min(K, have/2) - 1 were evaluated
            const unsigned used = kb != 0xFFFFFFFFu ? 2u * ((kb >> 8) + 1u) : 2u * (unsigned)min(K, (int)(g.have >> 1));
            g.n += used; g.have -= used;
            if (kb != 0xFFFFFFFFu) { vz = vzw; pending = false; }
        }
    }
    return x0in < 0.0 ? -vz : vz;
}
// |x| > 1: piecewise-constant comparison function in beta = exp(-u^2/2) space (random_mt.f90:2588-2693)
__device__ inline double vz_wing(Rng &g, double x0in, double a LART_PROF_ARG(pp)) {
    const double xc = 2.414213562373095048801688724209698, two_over_PI = 2.0 / LART_PI;
    double x0 = fabs(x0in), vz;
    LART_PROF_MARK(pp, 5); LART_PROF_TRIP(pp, 3);
    double x0sq = x0 * x0, api = a * LART_PI, beta0 = d_exp_neg(-x0sq / 2.0), h0_two = beta0 / a, h0 = h0_two / 2.0;
    // comparison function in beta-space: up to three segments [0,beta0) ~ beta/a, [beta0,beta1) = hA, [beta1,1) = hB
    double dbeta = 0.0, beta1 = beta0, hA = h0, hB, S0 = beta0 * h0, S1 = 0.0, S2;
    int mode;                                    // 0: single segment (uniform beta, Cb = h2); 1: segmented
    if (x0 < xc) {
        dbeta = sqrt(two_over_PI * a * (1.0 - beta0) * beta0 * x0); beta1 = beta0 + dbeta;
        double pb1 = sqrt(-2.0 * d_log(beta1));
        hB = two_over_PI * beta1 * pb1 / (x0sq - pb1 * pb1);
        S1 = dbeta * h0; S2 = (1.0 - beta1) * hB; mode = 1;
    } else {
        double h2 = 0.3861 / (x0sq - 1.373);
        if (h0_two < h2) { mode = 0; hB = h2; S2 = 1.0; }
        else if (h0 < h2) { mode = 1; hB = h2; S2 = (1.0 - beta0) * h2; }     // two segments: dbeta = 0
        else {
            dbeta = sqrt(two_over_PI * a * (1.0 - beta0) * beta0 * x0); beta1 = beta0 + dbeta;
            double pb1 = sqrt(-2.0 * d_log(beta1));
            double h1 = two_over_PI * beta1 * pb1 / (x0sq - pb1 * pb1);
            hB = h1 > h2 ? h1 : h2;
            S1 = dbeta * h0; S2 = (1.0 - beta1) * hB; mode = 1;
        }
    }
    bool two_seg = (mode == 1 && dbeta == 0.0 && !(x0 < xc));
    double Stot = S0 + S1 + S2, t1, delt, pb = 0.0;
    LART_PROF_MARK(pp, 8);
    for (;;) {
        LART_PROF_TRIP(pp, 1);
        // segment pick rS (not drawn in single-segment mode), then one draw u inside the segment: same order of draws
        // as the reference's three branches, two call sites instead of six
        double beta, Cb, rS = 2.0, u, acc;
        if (mode == 0) g.next2(u, acc); else { g.next2(rS, u); acc = g.next(); }
        if (mode == 0) { beta = u; Cb = hB; }
        else if (rS < S0 / Stot)                      { beta = beta0 * sqrt(u); Cb = beta / a; }
        else if (two_seg)                             { beta = beta0 + (1.0 - beta0) * u; Cb = hB; }
        else if (rS < 1.0 - S2 / Stot)                { beta = beta0 + dbeta * u; Cb = hA; }
        else                                          { beta = beta1 + (1.0 - beta1) * u; Cb = hB; }
        pb = sqrt(-2.0 * d_log(beta));
        // window width delt = atan((pb-x0)/a) - atan((-pb-x0)/a) in one arctangent: tan(delt) = 2 pb a / (a^2 + x0^2 - pb^2),
        // delt in (0, pi).  (The two-arctangent form of the reference cancels near -pi/2 for |x| >> a; this one does not.)
        double den = fma(x0 - pb, x0 + pb, a * a);
        delt = atan((2.0 * pb * a) / den); if (den < 0.0) delt = delt + LART_PI;
        if (acc * Cb < (beta / api) * delt) break;
    }
    t1 = atan((-pb - x0) / a);                               // lower edge of the window of the accepted try only
    LART_PROF_MARK(pp, 9);
    vz = x0 + a * d_tan(delt * g.next() + t1);
    LART_PROF_MARK(pp, 10);
    return x0in < 0.0 ? -vz : vz;
}
// atom velocity parallel to k: random_mt.f90:2562-2696 (rand_resonance_vz_seon)
__device__ inline double rand_resonance_vz(Rng &g, double x0in, double a LART_PROF_ARG(pp)) {
    return fabs(x0in) <= 1.0 ? vz_core(g, x0in, a LART_PROF_PASS(pp)) : vz_wing(g, x0in, a LART_PROF_PASS(pp));
}
// dipole/isotropic mix scattering angle: random_mt.f90:2974-2993
__device__ inline double rand_resonance_u(double u, double E1) {       // the same with the uniform draw handed in
    if (E1 > 0.0) {
        double p2 = sqrt((4.0 - E1) / (3.0 * E1));
        double Q = (4.0 * u - 2.0) / (E1 * (p2 * p2 * p2));
        double W = cbrt(Q + sqrt(Q * Q + 1.0));
        return p2 * (W - 1.0 / W);
    } else if (E1 < 0.0) {
        double p2 = sqrt(fabs((4.0 - E1) / (3.0 * E1)));
        double Q = (4.0 * u - 2.0) / (E1 * (p2 * p2 * p2));
        return 2.0 * p2 * cos((acos(Q) + LART_FOURPI) / 3.0);
    }
    return 2.0 * u - 1.0;
}
__device__ inline double rand_resonance(Rng &g, double E1) { return rand_resonance_u(g.next(), E1); }
// the same with the constants of the E1 > 0 branch computed once per kernel (Lds::dip_p2, dip_qs); they are wave-uniform
__device__ inline double uniform_f64(double v) {
    uint64_t b = __double_as_longlong(v);
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __longlong_as_double(((uint64_t)hi << 32) | lo);
}
__device__ inline void lds_set_dipole(Lds &l, double E1) {
    if (E1 > 0.0) { double p2 = sqrt((4.0 - E1) / (3.0 * E1)); l.dip_p2 = uniform_f64(p2); l.dip_qs = uniform_f64(1.0 / (E1 * (p2 * p2 * p2))); }
}
__device__ inline double dipole_cost(const Lds &l, double u, double E1) {
    if (l.dip_p2 > 0.0) { double Q = (4.0 * u - 2.0) * l.dip_qs, W = cbrt(Q + sqrt(Q * Q + 1.0)); return l.dip_p2 * (W - 1.0 / W); }
    return rand_resonance_u(u, E1);
}
// Henyey-Greenstein: random_mt.f90:3022-3042
__device__ inline double rand_hg(Rng &g, double gg) {
    double x = g.next();
    if (gg == 0.0) return 2.0 * x - 1.0;
    double g2 = gg * gg, twog = 2.0 * gg, t = (1.0 - g2) / (1.0 - gg + twog * x);
    return ((1.0 + g2) - t * t) / twog;
}
// Voigt-distributed offset: random_mt.f90:3062-3083
__device__ inline double rand_voigt(Rng &g, double a) {
    double c = tan(LART_PI * g.next() - LART_HALFPI), s2;
    return a * c + g.gauss_pair(s2) * 0.70710678118654752440084436210485;
}
// equispaced-grid linear interpolation: mathlib.f90:71-106
__device__ inline double interp_eq(int n, const double *x, const double *y, double xnew) {
    double dx = x[1] - x[0];
    int i = (int)((xnew - x[0]) / dx + 1);
    if (i <= 0) return y[0];
    if (i >= n) return y[n - 1];
    return y[i - 1] + (y[i] - y[i - 1]) * (xnew - x[i - 1]) / dx;
}
// alias-table draw with in-bin linear pdf inversion: random_mt.f90:2196-2216
__device__ inline double rand_alias_linear(Rng &g, const DMueller &m) {
    int n = m.nPDF - 1;
    int k = (int)floor(n * g.next()) + 1, idx;
    if (g.next() < m.phase_PDF[k - 1]) idx = k; else idx = m.alias[k - 1];
    double p0 = m.S11[idx - 1], p1 = m.S11[idx];
    return (sqrt(p0 * p0 + (p1 * p1 - p0 * p0) * g.next()) - p0) * (m.coss[idx] - m.coss[idx - 1]) / (p1 - p0) + m.coss[idx - 1];
}

// ---------------------------------------------------------------------------------------------
// photon packet (registers)
// ---------------------------------------------------------------------------------------------

__device__ inline size_t cidx(const DGrid &g, int i, int j, int k) {
    return (size_t)(i - 1) + (size_t)g.nx * ((size_t)(j - 1) + (size_t)g.ny * (size_t)(k - 1));
}
__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------------------------
// ray marches (Amanatides-Woo DDA): raytrace_car.f90
//   VAR 0: plain grid            raytrace_to_tau_car :1425-1648, raytrace_to_edge_car :410-508
//   VAR 1: +++ octant, mirrors   _car_xyzsym :1650-1949 / :584-781
//   VAR 2: slab, z only          _car_zonly  :2519-2675 / :1138-1234
//   VAR 5: mirrors in x and y    _car_xysym  :1951-2250 / :783-969
//   VAR 6: periodic in x and y   _car_xyper  :2252-2517 / :971-1136
// ---------------------------------------------------------------------------------------------
struct Trav { int ic, jc, kc, is, js, ks; double kx, ky, kz, tx, ty, tz, delx, dely, delz; };

// one axis of setup_traversal_car (:12-87); returns true when the ray leaves the grid at once
__device__ inline bool axis_setup_plain(int n, const double *face, double d, double xp, int &cell, int &step, double k, double &t, double &del) {
    if (k > 0.0) {
        if (cell > n && face[cell - 1] <= xp) return true;
        step = 1; t = (face[cell] - xp) / k; del = d / k;
    } else if (k < 0.0) {
        if (face[cell - 1] == xp) { if (cell > 1) cell -= 1; else return true; }
        step = -1; t = (face[cell - 1] - xp) / k; del = -d / k;
    } else { step = 0; t = LART_HUGEST; del = LART_HUGEST; }
    return false;
}
// one axis of the mirror-symmetric set-up (:1691-1721 tau [strict], :631-661 edge)
__device__ inline bool axis_setup_sym(int n, const double *face, double d, int i0, double xp, bool strict, int &cell, int &step, double &k, double &t, double &del) {
    if (k > 0.0) {
        if (cell > n && (strict ? (face[cell - 1] == xp) : (face[cell - 1] <= xp))) return true;
        step = 1; t = (face[cell] - xp) / k; del = d / k;
    } else if (k < 0.0) {
        step = -1; del = -d / k;
        if (face[cell - 1] == xp) { if (cell > 1) cell -= 1; else { cell = i0; step = 1; k = fabs(k); } }
        if (step == 1) t = (face[cell] - xp) / k; else t = (face[cell - 1] - xp) / k;
    } else { step = 0; t = LART_HUGEST; del = LART_HUGEST; }
    return false;
}
// z axis of the slab variants: tau (:2558-2582, strict) / edge (:1182-1201)
__device__ inline bool axis_setup_z(int n, const double *face, double d, double zp, bool strict, int &cell, int &step, double k, double &t, double &del) {
    if (k > 0.0) {
        if (cell > n && (strict ? (face[cell - 1] == zp) : (face[cell - 1] <= zp))) return true;
        step = 1; t = (face[cell] - zp) / k; del = d / k;
    } else if (k < 0.0) {
        if (face[cell - 1] == zp) { if (cell > 1) cell -= 1; else return true; }
        step = -1; t = (face[cell - 1] - zp) / k; del = -d / k;
    } else { step = 0; t = LART_HUGEST; del = LART_HUGEST; }
    return false;
}
// one periodic axis of the xy-periodic box: the tau march re-enters a start on the upper face at the lower one (:2287-2311),
// the edge march leaves there (:996-1017); a start on the lowest face moving down wraps to the top cell.  xp moves with the wrap.
__device__ inline bool axis_setup_per(int n, const double *face, double d, double &xp, bool tau_mode, int &cell, int &step, double k, double &t, double &del) {
    if (k > 0.0) {
        if (tau_mode) { if (cell > n && face[cell - 1] == xp) { cell = 1; xp = face[0]; } }
        else if (cell > n && face[cell - 1] <= xp) return true;
        step = 1; t = (face[cell] - xp) / k; del = d / k;
    } else if (k < 0.0) {
        if (face[cell - 1] == xp) { if (cell > 1) cell -= 1; else { cell = n; xp = face[n]; } }
        step = -1; t = (face[cell - 1] - xp) / k; del = -d / k;
    } else { step = 0; t = LART_HUGEST; del = LART_HUGEST; }
    return false;
}
template <int VAR>
__device__ inline bool trav_setup(const DGrid &g, const Lds &l, double &xp, double &yp, double zp, bool strict, Trav &t) {
    if (VAR == 5) {
        if (axis_setup_sym(g.nx, l.xf, g.dx, g.i0, xp, strict, t.ic, t.is, t.kx, t.tx, t.delx)) return true;
        if (axis_setup_sym(g.ny, l.yf, g.dy, g.j0, yp, strict, t.jc, t.js, t.ky, t.ty, t.dely)) return true;
        if (axis_setup_z(g.nz, l.zf, g.dz, zp, strict, t.kc, t.ks, t.kz, t.tz, t.delz)) return true;
    } else if (is_per(VAR)) {
        if (axis_setup_per(g.nx, l.xf, g.dx, xp, strict, t.ic, t.is, t.kx, t.tx, t.delx)) return true;
        if (axis_setup_per(g.ny, l.yf, g.dy, yp, strict, t.jc, t.js, t.ky, t.ty, t.dely)) return true;
        if (axis_setup_z(g.nz, l.zf, g.dz, zp, strict, t.kc, t.ks, t.kz, t.tz, t.delz)) return true;
    } else if (VAR == 0) {
        if (axis_setup_plain(g.nx, l.xf, g.dx, xp, t.ic, t.is, t.kx, t.tx, t.delx)) return true;
        if (axis_setup_plain(g.ny, l.yf, g.dy, yp, t.jc, t.js, t.ky, t.ty, t.dely)) return true;
        if (axis_setup_plain(g.nz, l.zf, g.dz, zp, t.kc, t.ks, t.kz, t.tz, t.delz)) return true;
    } else if (VAR == 1) {
        if (axis_setup_sym(g.nx, l.xf, g.dx, g.i0, xp, strict, t.ic, t.is, t.kx, t.tx, t.delx)) return true;
        if (axis_setup_sym(g.ny, l.yf, g.dy, g.j0, yp, strict, t.jc, t.js, t.ky, t.ty, t.dely)) return true;
        if (axis_setup_sym(g.nz, l.zf, g.dz, g.k0, zp, strict, t.kc, t.ks, t.kz, t.tz, t.delz)) return true;
    } else {
        t.is = 0; t.js = 0; t.tx = LART_HUGEST; t.ty = LART_HUGEST; t.delx = LART_HUGEST; t.dely = LART_HUGEST;
        if (axis_setup_z(g.nz, l.zf, g.dz, zp, strict, t.kc, t.ks, t.kz, t.tz, t.delz)) return true;
    }
    return false;
}
// advance the traversal across the nearest face (idx as Fortran minloc: first minimum).
// returns false when the ray left the grid.
template <int VAR>
__device__ inline bool trav_cross(const DGrid &g, Trav &t, int idx) {
    // value selects on purpose: the obvious three-way if() is merged by the compiler into pointer selects over the fields
    // of `t`, which moves the whole traversal state into scratch memory
    const bool ax = idx == 1, ay = idx == 2;
    int cell = ax ? t.ic : (ay ? t.jc : t.kc), step = ax ? t.is : (ay ? t.js : t.ks);
    const int n = ax ? g.nx : (ay ? g.ny : g.nz), m0 = VAR == 5 ? (ax ? g.i0 : g.j0) : (ax ? g.i0 : (ay ? g.j0 : g.k0));
    double k = ax ? t.kx : (ay ? t.ky : t.kz), tt = ax ? t.tx : (ay ? t.ty : t.tz);
    const double del = ax ? t.delx : (ay ? t.dely : t.delz);
    cell += step;
    bool ok = true;
    const bool mirror = VAR == 1 || (VAR == 5 && (ax || ay)), wrap = is_per(VAR) && (ax || ay);
    if (mirror && cell < 1) { cell = m0; step = 1; k = -k; }
    else if (wrap) { cell = cell < 1 ? n : (cell > n ? 1 : cell); }
    else if (cell < 1 || cell > n) ok = false;
    if (ok) tt += del;
    t.ic = ax ? cell : t.ic; t.jc = ay ? cell : t.jc; t.kc = (!ax && !ay) ? cell : t.kc;
    if (VAR == 1 || VAR == 5) {
        t.is = ax ? step : t.is; t.js = ay ? step : t.js; t.ks = (!ax && !ay) ? step : t.ks;
        t.kx = ax ? k : t.kx; t.ky = ay ? k : t.ky; t.kz = (!ax && !ay) ? k : t.kz;
    }
    t.tx = ax ? tt : t.tx; t.ty = ay ? tt : t.ty; t.tz = (!ax && !ay) ? tt : t.tz;
    return ok;
}
// h2_kappa_D (h2_mod.f90:58-78): the opacity of the H2 lines next to Ly-alpha as a multiplier of the H I opacity of a cell of Doppler width Dfreq.
// The kernels of this case carry the pseudo line type 9 = line type 1 + H2 (lart_gpu.hip sets it when lart_params.h2_nlines > 0).  The line table is read
// from a small device array (h2[0..3] dnu_Hz, [4..7] strength, [8..11] a_damp, [12..15] p_scat): indexing the kernel arguments with a run-time line number
// would copy the whole parameter block into scratch memory (measured on the peel kernel, which takes the line type at run time: 2.5x slower)
__device__ __forceinline__ double h2_kappa_D(const lart_params &par, const double *h2, const Lds &l, double xfreq, double Dfreq) {
    double kap = 0.0;
    const double ratio = par.h2_hi_width ? 1.0 : Dfreq / par.h2_Dfreq_Hz;
    for (int il = 0; il < par.h2_nlines; il++) { const double dx = h2[il] / Dfreq, x_h2 = (xfreq - dx) * ratio; kap = kap + h2[4 + il] * ratio * voigt_H(l.vt, x_h2, h2[8 + il]); }
    return kap;
}
__device__ inline int minloc3(double a, double b, double c) { int m = 1; double v = a; if (b < v) { m = 2; v = b; } if (c < v) m = 3; return m; }


// ---------------------------------------------------------------------------------------------
// octree marches (VAR 3): octree_mod.f90 + raytrace_amr.f90
// ---------------------------------------------------------------------------------------------
__device__ inline int amr_find_leaf(const DAmr &a, double x, double y, double z) {          // octree_mod.f90:149-171
    if (x < a.lo || x > a.hi || y < a.lo || y > a.hi || z < a.lo || z > a.hi) return 0;
    int ic = 1;
    for (;;) {
        int il = a.ileaf[ic - 1];
        if (il > 0) return il;
        const AmrGeom &q = a.geom[ic - 1];
        int io = (x >= q.cx ? 1 : 0) + (y >= q.cy ? 2 : 0) + (z >= q.cz ? 4 : 0);
        ic = a.children[io + 8 * (size_t)(ic - 1)];
        if (ic == 0) return 0;
    }
}
// distance to the exit face of cell q (iface 1=+x 2=-x 3=+y 4=-y 5=+z 6=-z, first minimum): :412-451
__device__ inline double amr_cell_exit(const AmrGeom &q, double x, double y, double z, double kx, double ky, double kz, int &iface) {
    double tx = LART_HUGEST, ty = LART_HUGEST, tz = LART_HUGEST; int fx = 1, fy = 3, fz = 5;
    if (kx > 0.0) tx = (q.cx + q.ch - x) / kx; else if (kx < 0.0) { tx = (q.cx - q.ch - x) / kx; fx = 2; }
    if (ky > 0.0) ty = (q.cy + q.ch - y) / ky; else if (ky < 0.0) { ty = (q.cy - q.ch - y) / ky; fy = 4; }
    if (kz > 0.0) tz = (q.cz + q.ch - z) / kz; else if (kz < 0.0) { tz = (q.cz - q.ch - z) / kz; fz = 6; }
    // minloc over t(1..6) with the unused direction of each axis = hugest: ties resolve to the lower face index
    double t = tx; iface = fx;
    if (ty < t || (ty == t && fy < iface)) { t = ty; iface = fy; }
    if (tz < t || (tz == t && fz < iface)) { t = tz; iface = fz; }
    return t;
}
// leaf entered through face `iface` of cell ic: neighbour table + descent, normal bit from iface: :718-758
__device__ inline int amr_next_leaf(const DAmr &a, int ic, int iface, double x, double y, double z) {
    int ng = a.neighbor[(iface - 1) + 6 * (size_t)(ic - 1)];
    if (ng == 0) return 0;
    int il;
    while ((il = a.ileaf[ng - 1]) == 0) {
        const AmrGeom &q = a.geom[ng - 1];
        int bx = x >= q.cx ? 1 : 0, by = y >= q.cy ? 2 : 0, bz = z >= q.cz ? 4 : 0;
        if (iface == 1) bx = 0; else if (iface == 2) bx = 1; else if (iface == 3) by = 0; else if (iface == 4) by = 2; else if (iface == 5) bz = 0; else bz = 4;
        int ch = a.children[(bx + by + bz) + 8 * (size_t)(ng - 1)];
        if (ch == 0) break;
        ng = ch;
    }
    return a.ileaf[ng - 1];
}
// a face of the root box (raytrace_amr.f90:176-201): periodic in x and y (wrap by the box size, look the leaf up from the root), mirror
// planes of the +++ octant / of the ++ quadrant (reflect, stay in the leaf); returns the leaf to continue in, 0 = escape
__device__ inline int amr_boundary(const KArgs &A, int il, int iface, double &x, double &y, double z, double &kx, double &ky, double &kz, bool &reflected) {
    const lart_params &par = A.par;
    if (par.xy_periodic) {
        if (iface == 1) x = x - (A.amr.hi - A.amr.lo); else if (iface == 2) x = x + (A.amr.hi - A.amr.lo);
        else if (iface == 3) y = y - (A.amr.hi - A.amr.lo); else if (iface == 4) y = y + (A.amr.hi - A.amr.lo);
        return iface <= 4 ? amr_find_leaf(A.amr, x, y, z) : 0;
    } else if (par.xyz_symmetry) {
        if (iface == 2) { kx = -kx; reflected = true; return il; }
        if (iface == 4) { ky = -ky; reflected = true; return il; }
        if (iface == 6) { kz = -kz; reflected = true; return il; }
    } else if (par.xy_symmetry) {
        if (iface == 2) { kx = -kx; reflected = true; return il; }
        if (iface == 4) { ky = -ky; reflected = true; return il; }
    }
    return 0;
}
// one deposit into J1(ix, ib): the block's LDS window around the line centre, or global memory
__device__ inline void jpa_add_J(const KArgs &A, const Lds &l, int ix, int ib, double v) {
    const int w = ix - 1 - l.j_lo;
    if (l.jp_J && w >= 0 && w < l.j_w) atomicAdd(&l.jp_J[w + l.j_w * (ib - 1)], v);
    else atomicAdd(&A.jpa.J1[(size_t)(ix - 1) + (size_t)A.par.nxfreq * (ib - 1)], v);
}
// amr_JPa_bin: octree_mod.f90:286-311 (3 = the leaf itself)
__device__ inline int amr_JPa_bin(const DAmr &a, int il, double x0, double y0, double z0) {
    if (a.jpa_geometry == 3) return il;
    if (a.jpa_geometry == 1) { int ib = (int)floor(sqrt(x0 * x0 + y0 * y0 + z0 * z0) / a.dr_JPa) + 1; return (ib >= 1 && ib <= a.nr_JPa) ? ib : 0; }
    int ib = (int)floor((z0 - a.lo) / a.dz_JPa) + 1;
    return clampi(ib, 1, a.nz_JPa);
}
// add_to_J_amr / add_to_Pnew_amr: raytrace_amr.f90:848-912 (bin of the segment midpoint)
__device__ inline void jpa_segment_amr(const KArgs &A, const Lds &l, const Photon &p, int il, const Cell &c, double x0, double y0, double z0, double del, double dtauH) {
    if (!(c.rhokap > 0.0)) return;
    int ib = amr_JPa_bin(A.amr, il, x0, y0, z0);
    if (ib < 1) return;
    if (A.jpa.calc_J) {
        double xr = p.xfreq * (c.Dfreq / A.par.Dfreq_ref);
        int ix = (int)floor((xr - A.par.xfreq_min) / A.par.dxfreq) + 1;
        if (ix >= 1 && ix <= A.par.nxfreq) jpa_add_J(A, l, ix, ib, del * p.wgt);
    }
    if (A.jpa.calc_Pnew) atomicAdd(&(l.jp_Pn ? l.jp_Pn : A.jpa.P1_new)[ib - 1], dtauH * p.wgt / (c.rhokap * c.Dfreq / A.jpa.cross0));
}
// State of a tau-to-the-edge march between two cell steps.  raytrace_to_edge runs begin + step until done; the peel kernel keeps one
// march per lane and lets every lane advance by one cell per iteration, so that a lane whose ray is over takes the next record while its
// neighbours are still marching (peel_kernel.hpp).  Same statements in the same order as the one-piece loops they replace.
struct EdgeMarch {
    bool live;
    double tau, d, xfreq, dist, kx, ky, kz;
    Trav t; double Dold, u1, rk, va, rD;          // Cartesian variants
    double x, y, z; int il; long long icl;        // octree leaf / owner clump and the running position
};
// raytrace_to_edge_amr (:265-344) / raytrace_to_dist_amr (:559-640)
__device__ __forceinline__ void edge_begin_amr(const KArgs &A, const Photon &p, EdgeMarch &m) {
    const DAmr &a = A.amr;
    m.x = p.x; m.y = p.y; m.z = p.z; m.il = p.ic;
    if (m.il <= 0) { m.il = amr_find_leaf(a, m.x, m.y, m.z); if (m.il <= 0) { m.live = false; return; } }
    const Cell *c = &A.g.cell[m.il - 1];
    m.u1 = c->vfx * m.kx + c->vfy * m.ky + c->vfz * m.kz;
}
__device__ __forceinline__ void edge_step_amr(const KArgs &A, const Lds &l, bool dust, EdgeMarch &m, unsigned &nsteps) {
    const DAmr &a = A.amr;
    const Cell *c = &A.g.cell[m.il - 1];
    int ic = a.icell_of_leaf[m.il - 1], iface;
    double t_exit = amr_cell_exit(a.geom[ic - 1], m.x, m.y, m.z, m.kx, m.ky, m.kz, iface);
    double rhokap = c->rhokap * calc_voigt(l, m.xfreq, c->voigt_a, c->Dfreq);
    if (l.lt == 9 && m.dist < 0.0) rhokap = rhokap + c->rhokap * h2_kappa_D(A.par, A.h2tab, l, m.xfreq, c->Dfreq);
    if (dust) rhokap = rhokap + c->rhokapD;
    nsteps++;
    if (m.dist >= 0.0 && m.d + t_exit >= m.dist) { m.tau = m.tau + (m.dist - m.d) * rhokap; m.live = false; return; }
    m.tau = m.tau + t_exit * rhokap;
    if (m.tau >= LART_TAU_HUGE) { m.live = false; return; }
    m.d = m.d + t_exit; m.x = m.x + t_exit * m.kx; m.y = m.y + t_exit * m.ky; m.z = m.z + t_exit * m.kz;
    int il_new = amr_next_leaf(a, ic, iface, m.x, m.y, m.z); bool reflected = false;
    if (il_new <= 0) { il_new = amr_boundary(A, m.il, iface, m.x, m.y, m.z, m.kx, m.ky, m.kz, reflected); if (m.d > LART_AMR_MAX_WRAPS * (a.hi - a.lo)) il_new = 0; }
    if (il_new <= 0) { m.live = false; return; }
    const Cell *cn = &A.g.cell[il_new - 1];
    double u2 = cn->vfx * m.kx + cn->vfy * m.ky + cn->vfz * m.kz;
    m.xfreq = reflected ? (m.xfreq + m.u1) - u2 : (m.xfreq + m.u1) * c->Dfreq / cn->Dfreq - u2;
    m.u1 = u2; m.il = il_new;
}


// ---------------------------------------------------------------------------------------------
// clump medium (VAR 4): clump_mod.f90 + raytrace_clump.f90.  xfreq stays in REF Doppler units; -/+ ulos at entry/exit.
// ---------------------------------------------------------------------------------------------
// the records of clump icl: from the workgroup's LDS copy when the population was staged there (lds_stage_clumps), else from global memory
__device__ __forceinline__ ClumpGeo cl_geo(const KArgs &A, const Lds &l, int icl) { return l.ct_geo ? l.ct_geo[icl - 1] : A.cl.geo[icl - 1]; }
__device__ __forceinline__ ClumpPhys cl_phys(const KArgs &A, const Lds &l, int icl) { return l.ct_phys ? l.ct_phys[icl - 1] : A.cl.phys[icl - 1]; }
#define LART_LDS_OBSERVERS 16                           // observers whose records (96 B each) are staged in LDS for the peel prefilter
#define LART_OVL_LDS_CLUMPS 128                          // largest population that is staged (= the populations that carry cg_mask)
__device__ inline size_t lds_clump_bytes(const DClumps &c) { return (c.cg_mask && c.N <= LART_OVL_LDS_CLUMPS) ? (size_t)c.N * (sizeof(ClumpGeo) + sizeof(ClumpPhys)) : 0; }
__device__ inline void lds_stage_clumps(Lds &l, const DClumps &c, unsigned char *at) {
    if (!lds_clump_bytes(c)) return;
    ClumpPhys *ph = (ClumpPhys *)at; ClumpGeo *ge = (ClumpGeo *)(at + (size_t)c.N * sizeof(ClumpPhys));
    const double *sp = (const double *)c.phys, *sg = (const double *)c.geo; double *dp = (double *)ph, *dg = (double *)ge;
    for (int i = threadIdx.x; i < (int)c.N * 8; i += blockDim.x) dp[i] = sp[i];
    for (int i = threadIdx.x; i < (int)c.N * 4; i += blockDim.x) dg[i] = sg[i];
    __syncthreads();
    l.ct_phys = ph; l.ct_geo = ge;
}
__device__ __forceinline__ double voigt_clump(const KArgs &A, const Lds &l, double xfreq, const ClumpPhys &c) { return calc_voigt(l, xfreq * (A.cl.Dfreq_ref / c.Dfreq), c.voigt_a, c.Dfreq); }   // clump_mod.f90:130-170 (line types 1, 2)
__device__ __forceinline__ double kappa_clump(const KArgs &A, const Lds &l, bool dust, double xfreq, const ClumpPhys &c) {                                                                 // :176-182
    double kap = c.rhokap * voigt_clump(A, l, xfreq, c);
    if (dust) kap = kap + c.rhokapD;
    return kap;
}
__device__ inline double ulos_clump(const KArgs &A, const ClumpPhys &c, double kx, double ky, double kz) { return (c.vx * kx + c.vy * ky + c.vz * kz) * (c.Dfreq / A.cl.Dfreq_ref); }   // :184-190
__device__ inline double clump_exit_dist(const ClumpGeo &q, double xp, double yp, double zp, double kx, double ky, double kz) {       // :1511-1526
    double rx = xp - q.x, ry = yp - q.y, rz = zp - q.z, b = rx * kx + ry * ky + rz * kz;
    double disc = b * b - (rx * rx + ry * ry + rz * rz) + q.r2; if (disc < 0.0) disc = 0.0;
    return fmax(0.0, -b + sqrt(disc));
}
__device__ inline double sphere_exit_dist(const DClumps &c, double xp, double yp, double zp, double kx, double ky, double kz) {       // :1530-1540
    double b = xp * kx + yp * ky + zp * kz, disc = b * b - (xp * xp + yp * yp + zp * zp) + c.sphere_R * c.sphere_R; if (disc < 0.0) disc = 0.0;
    return fmax(0.0, -b + sqrt(disc));
}
__device__ inline int cg_clamp(const DClumps &c, double v) { int i = (int)((v - c.cg_min) * c.cg_inv); return i < 0 ? 0 : (i > c.cgx - 1 ? c.cgx - 1 : i); }
// find_next_clump: :1390-1507 (Amanatides-Woo over the CSR grid; nearest entry point wins, stop once the cell starts beyond it)
__device__ inline bool find_next_clump(const DClumps &c, double xp, double yp, double zp, double kx, double ky, double kz, long long skip_icl, double t_max,
                                       double &t_entry, double &t_exit, long long &icl_found, unsigned &nsteps) {
    double best_te = LART_HUGEST, best_tx2 = 0.0, d = 0.0; long long best_icl = 0; const int cg = c.cgx;
    int ci = cg_clamp(c, xp), cj = cg_clamp(c, yp), ck = cg_clamp(c, zp), si, sj, sk;
    double tx, ty, tz, delx, dely, delz;
    if (kx > 0.0) { si = 1; tx = ((c.cg_min + (double)(ci + 1) * c.cg_d) - xp) / kx; delx = c.cg_d / kx; }
    else if (kx < 0.0) { si = -1; tx = ((c.cg_min + (double)ci * c.cg_d) - xp) / kx; delx = -c.cg_d / kx; } else { si = 0; tx = LART_HUGEST; delx = LART_HUGEST; }
    if (ky > 0.0) { sj = 1; ty = ((c.cg_min + (double)(cj + 1) * c.cg_d) - yp) / ky; dely = c.cg_d / ky; }
    else if (ky < 0.0) { sj = -1; ty = ((c.cg_min + (double)cj * c.cg_d) - yp) / ky; dely = -c.cg_d / ky; } else { sj = 0; ty = LART_HUGEST; dely = LART_HUGEST; }
    if (kz > 0.0) { sk = 1; tz = ((c.cg_min + (double)(ck + 1) * c.cg_d) - zp) / kz; delz = c.cg_d / kz; }
    else if (kz < 0.0) { sk = -1; tz = ((c.cg_min + (double)ck * c.cg_d) - zp) / kz; delz = -c.cg_d / kz; } else { sk = 0; tz = LART_HUGEST; delz = LART_HUGEST; }
    for (;;) {
        if (d > best_te || d > t_max) break;
        nsteps++;
        size_t q = (size_t)ci + (size_t)cj * cg + (size_t)ck * cg * cg;
        int ip1 = c.cg_start[q + 1];
        for (int ip = c.cg_start[q]; ip < ip1; ip++) {
            long long icl = c.cg_list[ip - 1];
            if (icl == skip_icl) continue;
            const ClumpGeo g = c.geo[icl - 1];
            double rx = xp - g.x, ry = yp - g.y, rz = zp - g.z, b = rx * kx + ry * ky + rz * kz;            // ray_sphere_isect :1363-1387
            double disc = b * b - (rx * rx + ry * ry + rz * rz) + g.r2;
            if (disc < 0.0) continue;
            disc = sqrt(disc);
            double te = -b - disc, tx2 = -b + disc;
            if (tx2 > 0.0 && te < best_te) { best_te = te; best_tx2 = tx2; best_icl = icl; }
        }
        if (tx <= ty && tx <= tz) { d = tx; ci += si; if (ci < 0 || ci >= cg) break; tx += delx; }
        else if (ty <= tz)        { d = ty; cj += sj; if (cj < 0 || cj >= cg) break; ty += dely; }
        else                      { d = tz; ck += sk; if (ck < 0 || ck >= cg) break; tz += delz; }
    }
    if (best_icl > 0 && best_te <= t_max) { t_entry = best_te; t_exit = fmin(best_tx2, t_max); icl_found = best_icl; return true; }
    return false;
}
// the clump holding a point (first hit of active_set_at_point, :1595-1634): birth clump, generate_photon.f90:325-332
__device__ inline int clump_at_point(const DClumps &c, double xp, double yp, double zp) {
    const int cg = c.cgx; int ci = cg_clamp(c, xp), cj = cg_clamp(c, yp), ck = cg_clamp(c, zp);
    for (int k = (ck > 0 ? ck - 1 : 0); k <= (ck < cg - 1 ? ck + 1 : cg - 1); k++)
        for (int j = (cj > 0 ? cj - 1 : 0); j <= (cj < cg - 1 ? cj + 1 : cg - 1); j++)
            for (int i = (ci > 0 ? ci - 1 : 0); i <= (ci < cg - 1 ? ci + 1 : cg - 1); i++) {
                size_t q = (size_t)i + (size_t)j * cg + (size_t)k * cg * cg;
                int ip1 = c.cg_start[q + 1];
                for (int ip = c.cg_start[q]; ip < ip1; ip++) {
                    int icl = c.cg_list[ip - 1]; const ClumpGeo g = c.geo[icl - 1];
                    double rx = xp - g.x, ry = yp - g.y, rz = zp - g.z;
                    if (rx * rx + ry * ry + rz * rz <= g.r2) return icl;
                }
            }
    return 0;
}
__device__ inline void tally_escape(const lart_params &par, const DTally &T, const Photon &p);
// raytrace_to_tau_clump: raytrace_clump.f90:83-201
__device__ __forceinline__ void raytrace_to_tau_clump(const KArgs &A, const Lds &l, bool dust, Photon &p, double tau_in, unsigned &nsteps) {
    const DClumps &c = A.cl;
    const double kx = p.kx, ky = p.ky, kz = p.kz; double tau_rem = tau_in; long long last_icl = 0;
    while (p.inside) {
        if (p.icl > 0) {
            const ClumpPhys ph = c.phys[p.icl - 1];
            double t_seg = clump_exit_dist(c.geo[p.icl - 1], p.x, p.y, p.z, kx, ky, kz), kap = kappa_clump(A, l, dust, p.xfreq, ph);
            nsteps++;
            if (tau_rem <= kap * t_seg) {
                double ds = tau_rem / fmax(kap, 2.2250738585072014e-308);
                p.x = p.x + ds * kx; p.y = p.y + ds * ky; p.z = p.z + ds * kz;
                return;
            }
            tau_rem = tau_rem - kap * t_seg;
            p.x = p.x + t_seg * kx; p.y = p.y + t_seg * ky; p.z = p.z + t_seg * kz;
            p.xfreq = p.xfreq + ulos_clump(A, ph, kx, ky, kz);
            last_icl = p.icl; p.icl = 0;
            if (p.x * p.x + p.y * p.y + p.z * p.z >= c.sphere_R * c.sphere_R) { p.inside = false; p.xfreq_ref = p.xfreq; tally_escape(A.par, A.t, p); return; }
        } else {
            double t_sp = sphere_exit_dist(c, p.x, p.y, p.z, kx, ky, kz), te, tx2; long long icl_found;
            if (t_sp <= 0.0) { p.inside = false; p.xfreq_ref = p.xfreq; tally_escape(A.par, A.t, p); return; }
            if (find_next_clump(c, p.x, p.y, p.z, kx, ky, kz, last_icl, t_sp, te, tx2, icl_found, nsteps)) {
                te = fmax(0.0, te);
                p.x = p.x + te * kx; p.y = p.y + te * ky; p.z = p.z + te * kz;
                p.xfreq = p.xfreq - ulos_clump(A, c.phys[icl_found - 1], kx, ky, kz);
                last_icl = 0; p.icl = (int)icl_found;
            } else {
                p.x = p.x + t_sp * kx; p.y = p.y + t_sp * ky; p.z = p.z + t_sp * kz;
                p.inside = false; p.xfreq_ref = p.xfreq; tally_escape(A.par, A.t, p); return;
            }
        }
    }
}
// raytrace_to_edge_clump_capped: :494-541.  The uncapped form (:204-267) is what the reference calls for the forced first
// flight; its only consumers are exp(-tau0) and 1 - exp(-tau0), which are exactly 0 and 1 beyond the cap, so one routine serves both.
__device__ __forceinline__ void edge_begin_clump(const KArgs &A, const Lds &l, bool dust, const Photon &p0, EdgeMarch &m, unsigned &nsteps) {
    const DClumps &c = A.cl;
    m.x = p0.x; m.y = p0.y; m.z = p0.z; m.icl = p0.icl;
    if (m.icl > 0) {
        const ClumpPhys ph = c.phys[m.icl - 1];
        double t_seg = clump_exit_dist(c.geo[m.icl - 1], m.x, m.y, m.z, m.kx, m.ky, m.kz);
        nsteps++;
        m.tau = m.tau + kappa_clump(A, l, dust, m.xfreq, ph) * t_seg;
        if (m.tau >= LART_TAU_HUGE) { m.live = false; return; }
        m.x = m.x + t_seg * m.kx; m.y = m.y + t_seg * m.ky; m.z = m.z + t_seg * m.kz;
        m.xfreq = m.xfreq + ulos_clump(A, ph, m.kx, m.ky, m.kz);
        if (m.x * m.x + m.y * m.y + m.z * m.z >= c.sphere_R * c.sphere_R) { m.live = false; return; }
    }
}
// one clump of the vacuum walk: the next entry (find_next_clump over the CSR cells) and the chord through it
__device__ __forceinline__ void edge_step_clump(const KArgs &A, const Lds &l, bool dust, EdgeMarch &m, unsigned &nsteps) {
    const DClumps &c = A.cl;
    double t_sp = sphere_exit_dist(c, m.x, m.y, m.z, m.kx, m.ky, m.kz), te, tx2; long long icl_found;
    if (t_sp <= 0.0) { m.live = false; return; }
    if (!find_next_clump(c, m.x, m.y, m.z, m.kx, m.ky, m.kz, m.icl, t_sp, te, tx2, icl_found, nsteps)) { m.live = false; return; }
    te = fmax(0.0, te);
    m.x = m.x + te * m.kx; m.y = m.y + te * m.ky; m.z = m.z + te * m.kz;
    const ClumpPhys ph = c.phys[icl_found - 1];
    double u_los = ulos_clump(A, ph, m.kx, m.ky, m.kz);
    m.xfreq = m.xfreq - u_los;
    double t_seg = clump_exit_dist(c.geo[icl_found - 1], m.x, m.y, m.z, m.kx, m.ky, m.kz);
    nsteps++;
    m.tau = m.tau + kappa_clump(A, l, dust, m.xfreq, ph) * t_seg;
    if (m.tau >= LART_TAU_HUGE) { m.live = false; return; }
    m.x = m.x + t_seg * m.kx; m.y = m.y + t_seg * m.ky; m.z = m.z + t_seg * m.kz;
    m.xfreq = m.xfreq + u_los;
    m.icl = icl_found;
    if (m.x * m.x + m.y * m.y + m.z * m.z >= c.sphere_R * c.sphere_R) m.live = false;
}

// ---------------------------------------------------------------------------------------------
// overlapping clumps (VAR 7): raytrace_clump.f90:606-918.  The frequency of the flight stays in the GLOBAL frame; inside k overlapping
// clumps the opacity is the sum of the k line profiles, each at its own frame frequency, and the scattering clump is drawn opacity-weighted.
//
// The reference collects every ENTER/EXIT of the ray up to the sphere's surface into arrays of 2048 events, sorts them and walks the
// intervals.  Here the same event sequence is produced on the fly: the CSR cells are visited in ray order, the clumps of a cell put their
// events into a short sorted list, and the list is drained up to the cell's far face before the next cell is read (a clump is registered in
// every cell its sphere touches, so an event inside a cell is known by the time the cell is left).  Same intervals, same order of the active
// list (append on ENTER, swap with the last on EXIT), so the same sums, bit for bit, while the lists fit: LART_OVL_ACT clumps around one point,
// LART_OVL_PEND events ahead of the ray inside one CSR cell row.  An overflow is counted in scal[7] and lart_gpu_get_tallies refuses the run.
// ---------------------------------------------------------------------------------------------
#define LART_OVL_ACT 32
#define LART_OVL_PEND 64
// sum_kap_active: :621-636
__device__ inline double ovl_sum_kap(const KArgs &A, const Lds &l, bool dust, const int *act, int na, double xfreq_g, double kx, double ky, double kz) {
    double sum = 0.0;
    for (int m = 0; m < na; m++) { const ClumpPhys ph = cl_phys(A, l, act[m]); sum = sum + kappa_clump(A, l, dust, xfreq_g - ulos_clump(A, ph, kx, ky, kz), ph); }
    return sum;
}
// TO_TAU: raytrace_to_tau_clump_overlap (:668-788): lim = the optical depth to go; returns with owner > 0 and t_end = the path length to the
//         scattering point, or owner = 0 and t_end = the path to the sphere's surface (escape).  u_own = the draw of sample_owner_clump (:644-664).
// else:   raytrace_to_edge_clump_overlap_capped (:858-918): lim = the cap; returns the optical depth.
//         col != nullptr: also raytrace_to_edge_column_clump_overlap (:935-1001): col[0] += N_gas, col[1] += tau_dust of every interval (cross0 = col[2])
// The holders of a point that took the in-clump fast path, carried to the later flights of the same packet.  reach = a lower bound on the distance from that point
// to anything that could change the list -- the faces of the CSR cell the list was read from (which also fix the list's order), the sphere's surface, every clump
// surface; all bounds hold for every direction, so the list is the one the 27-cell scan would rebuild, in this order, at every point of the BALL of that radius
// around the point.  Round 4: a flight is taken on the list when its end point stays in the ball, |P - P0| + s < reach (rounds 2-3 subtracted the flights' lengths
// from the reach -- the triangle inequality's bound on the same distance: a packet that scatters back and forth inside its clump used the reach up seven flights
// after a rebuild, its displacement grows like the square root of their number): no cell read, no containment tests, no sort.
#define LART_OVL_HOLD 8
// (round 4: the eight clump numbers live in four 64-bit registers, two per word, picked by selects -- an int[8] indexed at run time is a scratch array, and this list is
//  read on 88 % of the flights)
struct OvlHold {
    int na; double reach, x0, y0, z0; unsigned long long w0, w1, w2, w3;       // reach: radius of the ball around (x0, y0, z0), the point the list was built at, inside which it holds
    __device__ __forceinline__ void clear() { na = 0; reach = 0.0; x0 = 0.0; y0 = 0.0; z0 = 0.0; w0 = 0; w1 = 0; w2 = 0; w3 = 0; }
    __device__ __forceinline__ int get(int m) const { const unsigned long long w = m < 2 ? w0 : (m < 4 ? w1 : (m < 6 ? w2 : w3)); return (int)((m & 1) ? (w >> 32) : (w & 0xffffffffull)); }
    __device__ __forceinline__ void set(int m, int id) {
        const unsigned long long v = (unsigned long long)(unsigned)id;
        unsigned long long w = m < 2 ? w0 : (m < 4 ? w1 : (m < 6 ? w2 : w3));
        w = (m & 1) ? ((w & 0xffffffffull) | (v << 32)) : ((w & 0xffffffff00000000ull) | v);
        w0 = m < 2 ? w : w0; w1 = (m >= 2 && m < 4) ? w : w1; w2 = (m >= 4 && m < 6) ? w : w2; w3 = m >= 6 ? w : w3;
    }
};
// The march as a state that survives between calls: ovl_start does everything up to the first cell (holder list, fast paths, the traversal set-up), ovl_advance
// walks at most `max_cells` CSR cells -- each with the intervals up to its far face -- and says whether the march is over, ovl_finish hands out the result.
// ovl_march = start + advance to the end + finish is the one-piece form (peel rays, sight lines, probes); the transport kernel lets a lane that is in a march
// advance by a few cells per round of its wave while the other lanes go on scattering (transport_kernel.hpp): the same statements in the same order either way.
#ifdef LART_OVL_STATS                                   // diagnostic build (make SUFFIX=_st EXTRA=-DLART_OVL_STATS): how the flights of VAR 7 divide, in scal[8..15]
#define OVL_STAT(A, i, v) atomicAdd(&(A).t.scal[8 + (i)], (double)(v))
#else
#define OVL_STAT(A, i, v) do { } while (0)
#endif
// a flight on the carried holder list: true when it ends inside what is left of the list's reach (t_end, owner set); otherwise the list is dropped
__device__ inline bool ovl_carried(const KArgs &A, const Lds &l, bool dust, double xp, double yp, double zp, double kx, double ky, double kz, double xfreq_g, double lim, double u_own,
                                   unsigned &nsteps, double &t_end, int &owner, OvlHold &oh) {
    if (oh.na <= 0) return false;
    // every holder's opacity once: the eight terms live in registers picked by selects (an array indexed at run time would be a scratch array), the owner draw below adds
    // the same numbers in the same order as the sum
    double k0 = 0.0, k1 = 0.0, k2 = 0.0, k3 = 0.0, k4 = 0.0, k5 = 0.0, k6 = 0.0, k7 = 0.0, kap = 0.0;
    for (int m = 0; m < oh.na; m++) {
        const ClumpPhys ph = cl_phys(A, l, oh.get(m)); const double kv = kappa_clump(A, l, dust, xfreq_g - ulos_clump(A, ph, kx, ky, kz), ph);
        kap = kap + kv;
        k0 = m == 0 ? kv : k0; k1 = m == 1 ? kv : k1; k2 = m == 2 ? kv : k2; k3 = m == 3 ? kv : k3; k4 = m == 4 ? kv : k4; k5 = m == 5 ? kv : k5; k6 = m == 6 ? kv : k6; k7 = m == 7 ? kv : k7;
    }
    if (kap > 0.0) {
        const double s_ = lim / kap;
        const double dx = xp - oh.x0, dy = yp - oh.y0, dz = zp - oh.z0, d2 = dx * dx + dy * dy + dz * dz;
        const double room = (oh.reach * (1.0 - 1e-9) - 1e-12 * A.cl.cg_d) - s_;             // |P - P0| + s < reach, without the square root
        if (room > 0.0 && d2 < room * room) {
            nsteps++;
            t_end = s_;
            double cumul = 0.0;                                                               // sample_owner_clump :644-664
            for (int m = 0; m < oh.na; m++) {
                const double kv = m < 4 ? (m < 2 ? (m == 0 ? k0 : k1) : (m == 2 ? k2 : k3)) : (m < 6 ? (m == 4 ? k4 : k5) : (m == 6 ? k6 : k7));
                cumul = cumul + kv; owner = oh.get(m); if (u_own * kap <= cumul) break;
            }
            OVL_STAT(A, 0, 1);                                                                // flights on the carried list
            return true;
        }
    }
    oh.na = 0;
    return false;
}
struct OvlState {
    bool live, over, cells; int na, np, ci, cj, ck, si, sj, sk;
    double tx, ty, tz, delx, dely, delz, d, t_cur, acc, t_sp, u_own;
    unsigned long long vis0, vis1;
    int act[LART_OVL_ACT], pid[LART_OVL_PEND]; double pt[LART_OVL_PEND];        // pid: +icl = ENTER, -icl = EXIT; pt descending, the next event last
};
// returns true when the answer is known without a march (t_end, owner set); otherwise S holds the march at its first cell
template <bool TO_TAU>
__device__ inline bool ovl_start(const KArgs &A, const Lds &l, bool dust, double xp, double yp, double zp, double kx, double ky, double kz,
                                 double xfreq_g, double lim, double u_own, unsigned &nsteps, double &t_end, int &owner, OvlHold *oh, OvlState &S) {
    const DClumps &c = A.cl; const int cg = c.cgx;
    owner = 0; t_end = 0.0; S.live = false; S.over = false; S.acc = TO_TAU ? lim : 0.0; S.t_sp = 0.0; S.u_own = u_own;
    if (TO_TAU && oh && ovl_carried(A, l, dust, xp, yp, zp, kx, ky, kz, xfreq_g, lim, u_own, nsteps, t_end, owner, *oh)) return true;
    const double t_sp = sphere_exit_dist(c, xp, yp, zp, kx, ky, kz);
    S.t_sp = t_sp;
    if (t_sp <= 0.0) return true;
    if (TO_TAU) OVL_STAT(A, 1, 1);                                                            // holder lists rebuilt
    int *act = S.act; int na = 0;
    bool over = false;
    int ci = cg_clamp(c, xp), cj = cg_clamp(c, yp), ck = cg_clamp(c, zp);
    // active_set_at_point (clump_mod.f90:1594-1634) reads the 27 CSR cells around the point and keeps the clumps that hold it in first-seen
    // order.  A clump that holds the point is registered in the point's OWN cell (registration is by bounding box), so one cell gives the set;
    // the order of the 27-cell scan (k outer, i inner; first cell of the scan that registers the clump, then clump number: the CSR lists ascend)
    // is restored by a sort on that key, so the sums over the list and the owner draw are the reference's, bit for bit (tested against the
    // oracle's 27-cell restatement).  The same cell also bounds the distance to every surface from below -- a clump whose surface comes closer
    // than the point's distance to the faces of its cell is registered in it -- which the in-clump fast path below needs.
    double free_in = LART_HUGEST, free_out = LART_HUGEST;                                     // min of R^2 - |r|^2 over the holders, of |r|^2 - R^2 over the others
    // populations of at most 128 clumps (cg_mask): EVERY clump is tested against the point, not only the ones of its cell -- the same holders (a holder is registered
    // in the point's own cell), and lower bounds on the distances to ALL surfaces that do not depend on the size of the CSR cells (with 30 clumps of radius 0.3 the
    // grid has 32^3 cells of 0.06: the distance to the faces of the cell would end the carried list after three flights).  d_srf: min over all clumps of a lower bound
    // of | |r| - R | (single precision, rounded towards zero by a relative 4e-6: a bound, not a result).
    const bool all_clumps = TO_TAU && c.cg_mask != nullptr;
    double d_srf = LART_HUGEST;
    if (all_clumps) {
        for (int icl = 1; icl <= (int)c.N; icl++) {
            const ClumpGeo g = cl_geo(A, l, icl);
            const double rx = xp - g.x, ry = yp - g.y, rz = zp - g.z, rr = rx * rx + ry * ry + rz * rz;
            const bool in = rr <= g.r2;
            if (in) { if (na < LART_OVL_ACT) act[na++] = icl; else over = true; }
            const float num = (float)fabs(rr - g.r2) * 0.999998f, den = (sqrtf((float)rr) + sqrtf((float)g.r2)) * 1.000002f;
            d_srf = fmin(d_srf, (double)(num / den));
        }
    } else {
        const size_t q = (size_t)ci + (size_t)cj * cg + (size_t)ck * cg * cg;
        const int ip1 = c.cg_start[q + 1];
        for (int ip = c.cg_start[q]; ip < ip1; ip++) {
            const int icl = c.cg_list[ip - 1]; const ClumpGeo g = cl_geo(A, l, icl);
            const double rx = xp - g.x, ry = yp - g.y, rz = zp - g.z, rr = rx * rx + ry * ry + rz * rz;
            if (rr <= g.r2) { if (na < LART_OVL_ACT) act[na++] = icl; else over = true; free_in = fmin(free_in, g.r2 - rr); }
            else free_out = fmin(free_out, rr - g.r2);
        }
    }
    // the order of the list: first cell of the 27-cell scan that registers the clump, then the clump number.  order_reach: how far the point may move before that
    // order could change -- when the bounding box of every holder starts below the scan (i0 <= ci - 1 on every axis) all keys share the scan's first cell and the
    // order is the clump number, which stays so while the point does not move below those cells; otherwise the order is tied to this cell.
    double order_reach = LART_HUGEST;
    if (na > 1) {
        long long key[LART_OVL_ACT];
        const double x0 = c.cg_min + (double)ci * c.cg_d, y0 = c.cg_min + (double)cj * c.cg_d, z0 = c.cg_min + (double)ck * c.cg_d;
        int wx = 1 << 20, wy = 1 << 20, wz = 1 << 20;                                         // cells between the holders' bounding boxes and the scan, per axis (< 0: a box starts inside the scan)
        for (int m = 0; m < na; m++) {
            const ClumpGeo g = cl_geo(A, l, act[m]); const double rc = sqrt(g.r2);
            const int b0 = max(0, (int)((g.x - c.cg_min - rc) * c.cg_inv)), b1 = max(0, (int)((g.y - c.cg_min - rc) * c.cg_inv)), b2 = max(0, (int)((g.z - c.cg_min - rc) * c.cg_inv));
            const int i0 = max(b0, ci - 1), j0 = max(b1, cj - 1), k0 = max(b2, ck - 1);
            wx = min(wx, ci - 1 - b0); wy = min(wy, cj - 1 - b1); wz = min(wz, ck - 1 - b2);
            key[m] = ((((long long)k0 * cg + j0) * cg + i0) << 31) | (long long)act[m];
        }
        for (int m = 1; m < na; m++) { const long long km = key[m]; const int am = act[m]; int n = m - 1;
            while (n >= 0 && key[n] > km) { key[n + 1] = key[n]; act[n + 1] = act[n]; n--; }
            key[n + 1] = km; act[n + 1] = am; }
        if (wx >= 0 && wy >= 0 && wz >= 0 && ci >= 1 && cj >= 1 && ck >= 1) order_reach = fmin(fmin((xp - x0) + (double)wx * c.cg_d, (yp - y0) + (double)wy * c.cg_d), (zp - z0) + (double)wz * c.cg_d);
        else order_reach = fmin(fmin(fmin(xp - x0, x0 + c.cg_d - xp), fmin(yp - y0, y0 + c.cg_d - yp)), fmin(zp - z0, z0 + c.cg_d - zp));
    }
    if (TO_TAU && na > 0 && !over) {
        // In-clump fast path (the counterpart of the in-cell one of raytrace_to_tau): deep inside a clump the flight ends long before the next
        // ENTER or EXIT; the march would find one interval [0, t_first) with this list and stop at acc / kap in it.  Lower bounds of the
        // distances to the surfaces: (|r|^2 - R^2) / (|r| + R) outside, (R^2 - |r|^2) / (R + |r|) inside, denominators bounded from above.
        // Taken only when acc / kap lies clearly below all of them; everything else goes through the march.
        double kv[LART_OVL_ACT], kap = 0.0;                                                   // every clump's opacity once: the sum (:621-636) and the owner draw (:644-664) add the same numbers in the same order
        for (int m = 0; m < na; m++) { const ClumpPhys ph = cl_phys(A, l, act[m]); kv[m] = kappa_clump(A, l, dust, xfreq_g - ulos_clump(A, ph, kx, ky, kz), ph); kap = kap + kv[m]; }
        if (kap > 0.0) {
            const double x0 = c.cg_min + (double)ci * c.cg_d, y0 = c.cg_min + (double)cj * c.cg_d, z0 = c.cg_min + (double)ck * c.cg_d;
            const double dface = fmin(fmin(fmin(xp - x0, x0 + c.cg_d - xp), fmin(yp - y0, y0 + c.cg_d - yp)), fmin(zp - z0, z0 + c.cg_d - zp));
            const double s_ = lim / kap;
            const double reach = all_clumps ? fmin(fmin(order_reach, t_sp), d_srf)
                                            : fmin(fmin(dface, t_sp), fmin(free_out / (3.4641016151377544 * c.cg_d + 2.0 * c.radius_max), free_in / (2.0 * c.radius_max)));
#ifdef LART_OVL_STATS
            { const int which = reach == (all_clumps ? order_reach : dface) ? 0 : reach == t_sp ? 1 : 2;
              OVL_STAT(A, 8 + which, 1); OVL_STAT(A, 12, reach); OVL_STAT(A, 13, fmin(s_, 10.0)); OVL_STAT(A, 14, na > LART_OVL_HOLD ? 1 : 0); OVL_STAT(A, 15, dface); }
#endif
            if (s_ < reach * (1.0 - 1e-9)) {
                nsteps++;
                t_end = s_;
                double cumul = 0.0;                                                           // sample_owner_clump :644-664
                for (int m = 0; m < na; m++) { cumul = cumul + kv[m]; owner = act[m]; if (u_own * kap <= cumul) break; }
                if (oh && na <= LART_OVL_HOLD) {                                              // the same bounds without the direction: R - |r| instead of t_sp
                    const double ball = fmin(reach, c.sphere_R - sqrt(xp * xp + yp * yp + zp * zp));
                    if (ball - s_ > 0.0) { oh->na = na; oh->reach = ball; oh->x0 = xp; oh->y0 = yp; oh->z0 = zp; for (int m = 0; m < na; m++) oh->set(m, act[m]); OVL_STAT(A, 3, 1); }
                }
                OVL_STAT(A, 2, 1);                                                            // ... of which the fast path took the flight (3: and left a list behind)
                return true;
            }
        }
    }
    // collect_ray_events_overlap: clump_mod.f90:1664-1693
    if (kx > 0.0) { S.si = 1; S.delx = c.cg_d / kx; S.tx = ((c.cg_min + (double)(ci + 1) * c.cg_d) - xp) / kx; }
    else if (kx < 0.0) { S.si = -1; S.delx = -c.cg_d / kx; S.tx = ((c.cg_min + (double)ci * c.cg_d) - xp) / kx; } else { S.si = 0; S.delx = LART_HUGEST; S.tx = LART_HUGEST; }
    if (ky > 0.0) { S.sj = 1; S.dely = c.cg_d / ky; S.ty = ((c.cg_min + (double)(cj + 1) * c.cg_d) - yp) / ky; }
    else if (ky < 0.0) { S.sj = -1; S.dely = -c.cg_d / ky; S.ty = ((c.cg_min + (double)cj * c.cg_d) - yp) / ky; } else { S.sj = 0; S.dely = LART_HUGEST; S.ty = LART_HUGEST; }
    if (kz > 0.0) { S.sk = 1; S.delz = c.cg_d / kz; S.tz = ((c.cg_min + (double)(ck + 1) * c.cg_d) - zp) / kz; }
    else if (kz < 0.0) { S.sk = -1; S.delz = -c.cg_d / kz; S.tz = ((c.cg_min + (double)ck * c.cg_d) - zp) / kz; } else { S.sk = 0; S.delz = LART_HUGEST; S.tz = LART_HUGEST; }
    S.ci = ci; S.cj = cj; S.ck = ck; S.na = na; S.np = 0; S.over = over;
    S.d = 0.0; S.t_cur = 0.0;                                                                 // acc: optical depth still to go / gone so far
    S.cells = true; S.vis0 = 0ull; S.vis1 = 0ull; S.live = true;
    if (TO_TAU) { OVL_STAT(A, 4, 1); OVL_STAT(A, 7, na); }                                    // marches begun (7: holders at their start)
    return false;
}
// at most max_cells rounds of (one CSR cell, then the intervals up to its far face); true when the march is over
template <bool TO_TAU>
__device__ inline bool ovl_advance(const KArgs &A, const Lds &l, bool dust, double xp, double yp, double zp, double kx, double ky, double kz,
                                   double xfreq_g, double lim, unsigned &nsteps, double &t_end, int &owner, double *col, OvlState &S, int max_cells) {
    const DClumps &c = A.cl; const int cg = c.cgx;
    int *act = S.act, *pid = S.pid; double *pt = S.pt;
    int na = S.na, np = S.np, ci = S.ci, cj = S.cj, ck = S.ck;
    double tx = S.tx, ty = S.ty, tz = S.tz, d = S.d, t_cur = S.t_cur, acc = S.acc;
    const double t_sp = S.t_sp;
    bool over = S.over, cells = S.cells, done = false;
    // A clump of a population of large spheres is registered in many cells of the row: once the ray has dealt with it (events queued, or never to be met: missed,
    // behind the origin, beyond the sphere's surface, EXIT already behind the ray), every later cell would come to the same `continue`.  While the population
    // fits a 128-bit mask (the shipped overlap examples have 30 clumps) the clumps dealt with are remembered exactly, and the cells carry their list as such a
    // mask too (cg_mask; the lists ascend, so the set bits in ascending order ARE the list): a cell costs one 16-byte load and an and-not, a clump one sphere
    // intersection per ray instead of one per cell it is registered in.
    unsigned long long vis0 = S.vis0, vis1 = S.vis1;
    auto visit = [&](int icl) {
        const unsigned long long bit = 1ull << ((icl - 1) & 63); const bool hi = ((icl - 1) & 64) != 0;
        const ClumpGeo g = cl_geo(A, l, icl);
        const double rx = xp - g.x, ry = yp - g.y, rz = zp - g.z, b = rx * kx + ry * ky + rz * kz;     // ray_sphere_isect :1362-1386
        double disc = b * b - (rx * rx + ry * ry + rz * rz) + g.r2;
        bool dealt = disc < 0.0;
        double te = 0.0, tex = 0.0;
        if (!dealt) {
            disc = sqrt(disc);
            te = -b - disc; const double tx2 = -b + disc;
            dealt = !(tx2 > 0.0) || te > t_sp;
            if (!dealt) { tex = fmin(tx2, t_sp); dealt = tex <= t_cur; }                       // its EXIT lies behind the ray: seen and done in an earlier cell
            if (!dealt) for (int m = 0; m < np; m++) dealt = dealt || pid[m] == icl || pid[m] == -icl;
        }
        if (!dealt && np + 2 > LART_OVL_PEND) { over = true; return; }                       // (not marked: a later cell tries again, as the list walk did)
        if (hi) vis1 |= bit; else vis0 |= bit;
        if (dealt) return;
        for (int e = (te > 0.0 ? 0 : 1); e < 2; e++) {                                // ENTER (only when the origin is outside), then EXIT
            const double tn = e == 0 ? te : tex; const int idn = e == 0 ? icl : -icl;
            int pos = 0; while (pos < np && pt[pos] > tn) pos++;                      // after the later ones, before the equal ones (stable: first seen, first out)
            for (int m = np; m > pos; m--) { pt[m] = pt[m - 1]; pid[m] = pid[m - 1]; }
            pt[pos] = tn; pid[pos] = idn; np++;
        }
    };
    for (int round = 0; round < max_cells && !done; round++) {
        double t_lim = LART_HUGEST;
        if (TO_TAU) OVL_STAT(A, 5, 1);                                                        // cells walked
        if (cells && !(d > t_sp) && ci >= 0 && ci < cg && cj >= 0 && cj < cg && ck >= 0 && ck < cg) {
            const size_t q = (size_t)ci + (size_t)cj * cg + (size_t)ck * cg * cg;
            if (c.cg_mask) {
                unsigned long long m0 = c.cg_mask[2 * q] & ~vis0, m1 = c.cg_mask[2 * q + 1] & ~vis1;
                while (m0) { const int bpos = __ffsll((long long)m0) - 1; m0 &= m0 - 1; visit(bpos + 1); }
                while (m1) { const int bpos = __ffsll((long long)m1) - 1; m1 &= m1 - 1; visit(bpos + 65); }
            } else {
                const int ip1 = c.cg_start[q + 1];
                for (int ip = c.cg_start[q]; ip < ip1; ip++) visit(c.cg_list[ip - 1]);
            }
            if (tx <= ty && tx <= tz) { d = tx; tx = tx + S.delx; ci += S.si; }
            else if (ty <= tz)        { d = ty; ty = ty + S.dely; cj += S.sj; }
            else                      { d = tz; tz = tz + S.delz; ck += S.sk; }
            t_lim = d;
        } else cells = false;
        for (;;) {                                                                            // the intervals up to t_lim (:713-743 / :884-897)
            const bool ev = np > 0 && pt[np - 1] <= t_lim;
            if (!ev && cells) break;
            const double t_next = fmin(ev ? pt[np - 1] : t_sp, t_sp), dt = t_next - t_cur;
            if (dt > 0.0) {
                const double kap = ovl_sum_kap(A, l, dust, act, na, xfreq_g, kx, ky, kz);
                nsteps++;
                if (TO_TAU) OVL_STAT(A, 6, 1);                                                // intervals evaluated
                if (TO_TAU) {
                    if (kap > 0.0 && acc <= kap * dt) {
                        t_end = t_cur + acc / kap;
                        double cumul = 0.0;                                                   // sample_owner_clump :644-664
                        for (int m = 0; m < na; m++) {
                            const ClumpPhys ph = cl_phys(A, l, act[m]);
                            cumul = cumul + kappa_clump(A, l, dust, xfreq_g - ulos_clump(A, ph, kx, ky, kz), ph); owner = act[m];
                            if (S.u_own * kap <= cumul) break;
                        }
                        done = true; break;
                    }
                    acc = acc - kap * dt;
                } else {
                    acc = acc + kap * dt; if (acc >= lim) { done = true; break; }
                    if (col) for (int m = 0; m < na; m++) { const ClumpPhys ph = cl_phys(A, l, act[m]); col[0] = col[0] + (ph.rhokap / col[2]) * ph.Dfreq * dt; col[1] = col[1] + ph.rhokapD * dt; }
                }
                t_cur = t_next;
            }
            if (!ev) { done = true; break; }                                                  // that was the last interval, up to the surface
            const int id = pid[--np];                                                         // apply_event :769-786
            if (id > 0) { if (na < LART_OVL_ACT) act[na++] = id; else over = true; }
            else for (int m = 0; m < na; m++) if (act[m] == -id) { act[m] = act[na - 1]; na--; break; }
        }
    }
    S.na = na; S.np = np; S.ci = ci; S.cj = cj; S.ck = ck; S.tx = tx; S.ty = ty; S.tz = tz; S.d = d; S.t_cur = t_cur; S.acc = acc;
    S.over = over; S.cells = cells; S.vis0 = vis0; S.vis1 = vis1;
    if (done) S.live = false;
    return done;
}
template <bool TO_TAU>
__device__ inline double ovl_finish(const KArgs &A, const OvlState &S, double &t_end, int owner) {
    if (S.over && A.t.scal) atomicAdd(&A.t.scal[7], 1.0);
    if (TO_TAU) { if (owner == 0) t_end = S.t_sp; return 0.0; }
    return S.acc;
}
template <bool TO_TAU>
__device__ inline double ovl_march(const KArgs &A, const Lds &l, bool dust, double xp, double yp, double zp, double kx, double ky, double kz,
                                   double xfreq_g, double lim, double u_own, unsigned &nsteps, double &t_end, int &owner, double *col = nullptr, OvlHold *oh = nullptr,
                                   OvlState *room = nullptr) {
    OvlState own; OvlState &S = room ? *room : own;          // (the transport kernel lends its own state: one set of lists in scratch instead of two)
    if (ovl_start<TO_TAU>(A, l, dust, xp, yp, zp, kx, ky, kz, xfreq_g, lim, u_own, nsteps, t_end, owner, oh, S)) return 0.0;
    while (!ovl_advance<TO_TAU>(A, l, dust, xp, yp, zp, kx, ky, kz, xfreq_g, lim, nsteps, t_end, owner, col, S, 1 << 30)) { }
    return ovl_finish<TO_TAU>(A, S, t_end, owner);
}
// the end of raytrace_to_tau_clump_overlap (:668-788): move to the scattering point or out of the sphere
__device__ inline void ovl_apply(const KArgs &A, Photon &p, double t_end, int owner) {
    p.x = p.x + t_end * p.kx; p.y = p.y + t_end * p.ky; p.z = p.z + t_end * p.kz;
    p.icl = owner;
    if (owner == 0) { p.inside = false; p.xfreq_ref = p.xfreq; tally_escape(A.par, A.t, p); }
}
// raytrace_to_tau_clump_overlap: :668-788
__device__ inline void raytrace_to_tau_clump_ovl(const KArgs &A, const Lds &l, bool dust, Photon &p, double tau_in, double u_own, unsigned &nsteps, OvlHold *oh = nullptr) {
    double t_end; int owner;
    ovl_march<true>(A, l, dust, p.x, p.y, p.z, p.kx, p.ky, p.kz, p.xfreq, tau_in, u_own, nsteps, t_end, owner, nullptr, oh);
    ovl_apply(A, p, t_end, owner);
}

// ---------------------------------------------------------------------------------------------
// VAR 9 = VAR 7 for a population of at most 64 clumps whose records are staged in LDS (the shipped overlap inputs have 30).  Round 4: counters of the
// -DLART_PROFILE build put 45 % of the wave clocks of VAR 7 into the marches (3.6 % of the flights) and most of another 35 % into the holder-list rebuilds
// (12 %): one or two lanes of a wave walking scratch-backed lists (sorted event queue, holder list, sort keys: 1.9 KB per lane, every access a memory
// round trip) while the other lanes wait.  Here a flight that cannot stay on its carried list is executed by the WAVE, one flight at a time, ONE LANE PER
// CLUMP: containment test, distance bound, sort key, opacity at the ray's frequency (one line profile per lane instead of one per holder in sequence) and
// the ray's ENTER / EXIT times are computed side by side; the holder list lives in one VGPR (lane m = the m-th holder); the event queue is not stored at
// all -- the next event is the wave minimum of the lanes' next times.  Same event sequence (ENTER before EXIT of one clump; two clumps at exactly equal
// times: the lower number first, where the list walk had first seen, first out), same order of the holder list (first cell of the 27-cell scan that
// registers the clump, then the clump number; append on ENTER, swap with the last on EXIT), the sums over it added in that order: the same numbers as
// ovl_start / ovl_advance, with no list that could overflow and nothing in scratch.
// ---------------------------------------------------------------------------------------------
#define LART_OVL_COOP_CLUMPS 64
__device__ __forceinline__ int rl_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ double rl_f64(double v, int lane) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane)); }
__device__ __forceinline__ long long rl_i64(long long v, int lane) {
    return (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), lane) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane));
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
}
// minimum over the 64 lanes, in every lane: two quad permutes and two row rotations (DPP) leave each row of 16 with its minimum, the four rows meet through v_readlane
__device__ __forceinline__ double wave_min_f64(double v) {
    v = fmin(v, dpp_f64<0xB1>(v));                            // quad_perm:[1,0,3,2]
    v = fmin(v, dpp_f64<0x4E>(v));                            // quad_perm:[2,3,0,1]
    v = fmin(v, dpp_f64<0x124>(v));                           // row_ror:4
    v = fmin(v, dpp_f64<0x128>(v));                           // row_ror:8
    return fmin(fmin(rl_f64(v, 0), rl_f64(v, 16)), fmin(rl_f64(v, 32), rl_f64(v, 48)));
}
// minimum over the lanes [B, B + W) of a value whose rows of 16 lanes already hold their minima in every lane (the four DPP steps of wave_min_f64)
template <int W> __device__ __forceinline__ double group_min_rows(double v, int B) {
    if (W == 16) return rl_f64(v, B);
    if (W == 32) return fmin(rl_f64(v, B), rl_f64(v, B + 16));
    return fmin(fmin(rl_f64(v, 0), rl_f64(v, 16)), fmin(rl_f64(v, 32), rl_f64(v, 48)));
}
__device__ __forceinline__ double row_min_f64(double v) {
    v = fmin(v, dpp_f64<0xB1>(v)); v = fmin(v, dpp_f64<0x4E>(v)); v = fmin(v, dpp_f64<0x124>(v)); v = fmin(v, dpp_f64<0x128>(v));
    return v;
}
// the flights of the lanes with `need` by the whole wave (every lane of the wave calls this, converged): W lanes per flight, one lane per clump, 64 / W flights side by
// side -- what depends on (flight, clump) is computed for all of them at once (containment, distance bound, sort key, opacity at the ray's frequency, ENTER / EXIT
// times), what depends on the flight alone (the order of the holders, the sums over them, the walk over the events) one flight after the other on the lanes
// [B, B + W) of its group.  TO_TAU: res = t_end, owner as ovl_start / ovl_advance / ovl_finish leave them (lim = the optical depth to go), *oh = the list carried to the
// next flights; else res = the optical depth to the sphere's surface (raytrace_to_edge_clump_overlap_capped with lim = the cap).
template <bool TO_TAU, int W>
__device__ inline void ovl_coop_w(const KArgs &A, const Lds &l, bool dust, bool need, const Photon &p, double lim, double u_own, unsigned &nsteps, double &res, int &owner, OvlHold *oh) {
    static_assert(W == 32 || W == 64, "one or two flights per turn");
    constexpr int G = 64 / W;
    const DClumps &c = A.cl; const int cg = c.cgx; const int N = (int)c.N;
    const int lane = (int)(threadIdx.x & 63u), j = lane & (W - 1), grp = lane / W;      // this lane's clump: number j + 1; its group of lanes
    const bool has = j < N;
    const double INF = LART_HUGEST;
    const unsigned long long GM = W == 64 ? ~0ull : ((1ull << (W & 63)) - 1ull);
    // what depends on the flight alone, computed by the flights' own lanes side by side
    double my_tsp = 0.0, my_rp = 0.0; int my_ci = 0, my_cj = 0, my_ck = 0;
    if (need) {
        my_tsp = sphere_exit_dist(c, p.x, p.y, p.z, p.kx, p.ky, p.kz);
        if (TO_TAU) my_rp = sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
        my_ci = cg_clamp(c, p.x); my_cj = cg_clamp(c, p.y); my_ck = cg_clamp(c, p.z);
    }
    unsigned long long todo = __ballot(need);
    while (todo) {
        // the flights of this turn: group q takes the q-th of the waiting lanes
        const int L0 = __ffsll((long long)todo) - 1; todo &= todo - 1;
        int L1 = -1; if (G > 1 && todo) { L1 = __ffsll((long long)todo) - 1; todo &= todo - 1; }
        const int mine = grp == 0 ? L0 : L1;
        const bool live = mine >= 0;
        const int src = live ? mine : lane;
        // ---- per (flight, clump): every lane for its group's flight
        const double xq = __shfl(p.x, src, 64), yq = __shfl(p.y, src, 64), zq = __shfl(p.z, src, 64), kxq = __shfl(p.kx, src, 64), kyq = __shfl(p.ky, src, 64), kzq = __shfl(p.kz, src, 64);
        const double xfq = __shfl(p.xfreq, src, 64), tspq = __shfl(my_tsp, src, 64);
        const int ciq = __shfl(my_ci, src, 64), cjq = __shfl(my_cj, src, 64), ckq = __shfl(my_ck, src, 64);
        const ClumpGeo g = has ? l.ct_geo[j] : ClumpGeo{0.0, 0.0, 0.0, -1.0};      // this lane's clump (read again in every turn: LDS is cheaper than eight registers held across the loop)
        const double rx = xq - g.x, ry = yq - g.y, rz = zq - g.z, rr = rx * rx + ry * ry + rz * rz;
        const bool in = live && has && rr <= g.r2;            // active_set_at_point (clump_mod.f90:1594-1634): a holder is registered in the point's own cell, so "every clump" finds the same set
        const unsigned long long hm_all = __ballot(in);
        const double rc = sqrt(fmax(g.r2, 0.0));
        const int b0 = max(0, (int)((g.x - c.cg_min - rc) * c.cg_inv)), b1 = max(0, (int)((g.y - c.cg_min - rc) * c.cg_inv)), b2 = max(0, (int)((g.z - c.cg_min - rc) * c.cg_inv));
        const int i0 = max(b0, ciq - 1), j0 = max(b1, cjq - 1), k0 = max(b2, ckq - 1);
        const long long key = ((((long long)k0 * cg + j0) * cg + i0) << 31) | (long long)(j + 1);      // the order of the list: first cell of the 27-cell scan that registers the clump, then the clump number (ovl_start)
        const int wxj = ciq - 1 - b0, wyj = cjq - 1 - b1, wzj = ckq - 1 - b2;
        double kapj = 0.0;                                    // every clump's opacity at the ray's frequency: constant along the ray (sum_kap_active :621-636 evaluates it per interval)
        if (live && has) { const ClumpPhys ph = l.ct_phys[j]; kapj = kappa_clump(A, l, dust, xfq - ulos_clump(A, ph, kxq, kyq, kzq), ph); }
        double dsrf = INF;                                    // lower bound of | |r| - R | (single precision, rounded towards zero: a bound, not a result)
        if (TO_TAU && has) { const float num = (float)fabs(rr - g.r2) * 0.999998f, den = (sqrtf((float)rr) + sqrtf((float)g.r2)) * 1.000002f; dsrf = (double)(num / den); }
        if (TO_TAU) dsrf = row_min_f64(dsrf);
        const double bq = rx * kxq + ry * kyq + rz * kzq;       // ray_sphere_isect :1362-1386
        double disc = bq * bq - rr + g.r2;
        bool hit = live && has && !(disc < 0.0);
        disc = sqrt(fmax(disc, 0.0));
        const double te = -bq - disc, tx2 = -bq + disc;
        hit = hit && tx2 > 0.0 && !(te > tspq);
        double t_in = (hit && te > 0.0) ? te : INF, t_out = hit ? fmin(tx2, tspq) : INF;     // ENTER only when the origin is outside
        int v_act = 0;                                        // the holder lists: lane B + m = the m-th holder of the group's flight
        // ---- per flight
        for (int q = 0; q < G; q++) {
            const int L = q == 0 ? L0 : L1; if (L < 0) break;
            const int B = q * W; const bool ing = grp == q;
            const double xp = rl_f64(p.x, L), yp = rl_f64(p.y, L), zp = rl_f64(p.z, L), limL = rl_f64(lim, L), u_ownL = rl_f64(u_own, L), t_sp = rl_f64(my_tsp, L);
            double out = 0.0; int own = 0; unsigned cnt = 0;  // the flight's result (the same in every lane)
            int hna = 0; double hreach = 0.0; unsigned long long hw0 = 0, hw1 = 0, hw2 = 0, hw3 = 0;      // the list it leaves behind
            if (t_sp > 0.0) {
                if (TO_TAU) if (lane == 0) OVL_STAT(A, 1, 1);      // holder lists rebuilt
                const unsigned long long hm = (hm_all >> B) & GM;
                int na = __popcll(hm);
                int rank = 0; double order_reach = INF;
                if (na > 1) {
                    const int ci = rl_i32(my_ci, L), cj = rl_i32(my_cj, L), ck = rl_i32(my_ck, L);
                    int wx = 1 << 20, wy = 1 << 20, wz = 1 << 20;
                    for (unsigned long long m = hm; m; m &= m - 1) {
                        const int h = B + __ffsll((long long)m) - 1;
                        rank += (rl_i64(key, h) < key) ? 1 : 0;
                        wx = min(wx, rl_i32(wxj, h)); wy = min(wy, rl_i32(wyj, h)); wz = min(wz, rl_i32(wzj, h));
                    }
                    const double x0 = c.cg_min + (double)ci * c.cg_d, y0 = c.cg_min + (double)cj * c.cg_d, z0 = c.cg_min + (double)ck * c.cg_d;
                    if (wx >= 0 && wy >= 0 && wz >= 0 && ci >= 1 && cj >= 1 && ck >= 1) order_reach = fmin(fmin((xp - x0) + (double)wx * c.cg_d, (yp - y0) + (double)wy * c.cg_d), (zp - z0) + (double)wz * c.cg_d);
                    else order_reach = fmin(fmin(fmin(xp - x0, x0 + c.cg_d - xp), fmin(yp - y0, y0 + c.cg_d - yp)), fmin(zp - z0, z0 + c.cg_d - zp));
                }
                { const int got = __builtin_amdgcn_ds_permute(((ing && in) ? B + rank : (G > 1 ? ((B + W) & 63) : 63)) << 2, j + 1); if (ing) v_act = got; }      // lane B + m = the m-th holder (everybody else aims at a lane whose value is dropped)
                auto sum_act = [&]() { double s = 0.0; for (int m = 0; m < na; m++) s = s + rl_f64(kapj, B + rl_i32(v_act, B + m) - 1); return s; };
                auto draw_owner = [&](double kap) { double cumul = 0.0; int o = 0;                // sample_owner_clump :644-664
                    for (int m = 0; m < na; m++) { o = rl_i32(v_act, B + m); cumul = cumul + rl_f64(kapj, B + o - 1); if (u_ownL * kap <= cumul) break; } return o; };
                bool finished = false;
                if (TO_TAU && na > 0) {                           // the in-clump fast path of ovl_start
                    const double kap = sum_act();
                    if (kap > 0.0) {
                        const double d_srf = group_min_rows<W>(dsrf, B);
                        const double s_ = limL / kap, reach = fmin(fmin(order_reach, t_sp), d_srf);
                        if (s_ < reach * (1.0 - 1e-9)) {
                            cnt++; out = s_; own = draw_owner(kap);
                            if (oh && na <= LART_OVL_HOLD) {
                                const double ball = fmin(reach, c.sphere_R - rl_f64(my_rp, L));
                                if (ball - s_ > 0.0) {
                                    hna = na; hreach = ball;
                                    for (int m = 0; m < na; m++) { const unsigned long long v = (unsigned long long)(unsigned)rl_i32(v_act, B + m) << ((m & 1) * 32);
                                        if (m < 2) hw0 |= v; else if (m < 4) hw1 |= v; else if (m < 6) hw2 |= v; else hw3 |= v; }
                                    if (lane == 0) OVL_STAT(A, 3, 1);
                                }
                            }
                            if (lane == 0) OVL_STAT(A, 2, 1);
                            finished = true;
                        }
                    }
                }
                if (!finished) {                                  // the march: collect_ray_events_overlap (clump_mod.f90:1664-1693) + the interval walk (:713-743 / :884-897)
                    if (TO_TAU) { if (lane == 0) { OVL_STAT(A, 4, 1); OVL_STAT(A, 7, na); } }
                    double t_cur = 0.0, acc = TO_TAU ? limL : 0.0;
                    for (;;) {
                        const double cand = fmin(t_in, t_out), tmin = group_min_rows<W>(row_min_f64(cand), B);
                        const bool ev = tmin < INF;
                        const double t_next = fmin(ev ? tmin : t_sp, t_sp), dt = t_next - t_cur;
                        if (dt > 0.0) {
                            const double kap = sum_act();
                            cnt++;
                            if (TO_TAU) { if (lane == 0) OVL_STAT(A, 6, 1);
                                if (kap > 0.0 && acc <= kap * dt) { out = t_cur + acc / kap; own = draw_owner(kap); break; }
                                acc = acc - kap * dt;
                            } else { acc = acc + kap * dt; if (acc >= limL) break; }
                            t_cur = t_next;
                        }
                        if (!ev) break;                                                           // that was the last interval, up to the surface
                        const int Wn = __ffsll((long long)((__ballot(cand == tmin) >> B) & GM)) - 1;      // apply_event :769-786 (the lowest clump number of the group at that time)
                        const bool enter = rl_i32((t_in <= t_out) ? 1 : 0, B + Wn) != 0;
                        if (enter) { if (lane == B + na) v_act = Wn + 1; na++; if (lane == B + Wn) t_in = INF; }
                        else {
                            const unsigned long long at = (__ballot(ing && j < na && v_act == Wn + 1) >> B) & GM;
                            if (at) { const int pos = __ffsll((long long)at) - 1, last = rl_i32(v_act, B + na - 1); if (lane == B + pos) v_act = last; na--; }
                            if (lane == B + Wn) t_out = INF;
                        }
                    }
                    if (TO_TAU) { if (own == 0) out = t_sp; } else out = acc;
                }
            }
            if (lane == L) {
                res = out; owner = own; nsteps += cnt;
                if (TO_TAU && oh) { oh->na = hna; oh->reach = hreach; oh->x0 = xp; oh->y0 = yp; oh->z0 = zp; oh->w0 = hw0; oh->w1 = hw1; oh->w2 = hw2; oh->w3 = hw3; }
            }
        }
    }
}
template <bool TO_TAU>
__device__ inline void ovl_coop(const KArgs &A, const Lds &l, bool dust, bool need, const Photon &p, double lim, double u_own, unsigned &nsteps, double &res, int &owner, OvlHold *oh) {
    if (A.cl.N <= 32) ovl_coop_w<TO_TAU, 32>(A, l, dust, need, p, lim, u_own, nsteps, res, owner, oh);       // two flights side by side (the shipped overlap inputs have 30 clumps)
    else ovl_coop_w<TO_TAU, 64>(A, l, dust, need, p, lim, u_own, nsteps, res, owner, oh);
}

// tau to the grid edge (dist < 0) or to a finite distance (raytrace_to_dist_car :89-195, VAR 0 only)
template <int VAR>
__device__ __forceinline__ void edge_begin(const KArgs &A, const Lds &l, bool dust, const Photon &p, double xfreq, double kx, double ky, double kz, double dist, EdgeMarch &m, unsigned &nsteps) {
    if (VAR == 8) { edge_begin<0>(A, l, dust, p, xfreq, kx, ky, kz, dist, m, nsteps); return; }
    m.live = true; m.tau = 0.0; m.d = 0.0; m.xfreq = xfreq; m.dist = dist; m.kx = kx; m.ky = ky; m.kz = kz;
    if (VAR == 3) { edge_begin_amr(A, p, m); return; }
    if (VAR == 4) { edge_begin_clump(A, l, dust, p, m, nsteps); return; }
    if (is_ovl(VAR)) { double te_; int ow_; m.tau = ovl_march<false>(A, l, dust, p.x, p.y, p.z, kx, ky, kz, xfreq, LART_TAU_HUGE, 0.0, nsteps, te_, ow_); m.live = false; return; }   // one piece: the event lists do not survive between steps
    const DGrid &g = A.g;
    m.t.kx = kx; m.t.ky = ky; m.t.kz = kz; m.t.ic = p.ic; m.t.jc = p.jc; m.t.kc = p.kc;
    double xp = p.x, yp = p.y;
    if (trav_setup<VAR>(g, l, xp, yp, p.z, false, m.t)) { m.live = false; return; }
    const Cell *c = &g.cell[cidx(g, m.t.ic, m.t.jc, m.t.kc)];
    m.Dold = c->Dfreq; m.u1 = c->vfx * m.t.kx + c->vfy * m.t.ky + c->vfz * m.t.kz;
    m.rk = c->rhokap; m.va = c->voigt_a; m.rD = c->rhokapD;
}
template <int VAR>
__device__ __forceinline__ void edge_step(const KArgs &A, const Lds &l, bool dust, EdgeMarch &m, unsigned &nsteps) {
    if (VAR == 8) { edge_step<0>(A, l, dust, m, nsteps); return; }
    if (VAR == 3) { edge_step_amr(A, l, dust, m, nsteps); return; }
    if (VAR == 4) { edge_step_clump(A, l, dust, m, nsteps); return; }
    const DGrid &g = A.g; Trav &t = m.t;
    double rhokap = m.rk * calc_voigt(l, m.xfreq, m.va, m.Dold);
    if (l.lt == 9 && m.dist < 0.0) rhokap = rhokap + m.rk * h2_kappa_D(A.par, A.h2tab, l, m.xfreq, m.Dold);
    if (dust) rhokap = rhokap + m.rD;
    nsteps++;
    int idx = (VAR == 2) ? 3 : minloc3(t.tx, t.ty, t.tz);
    double tt = idx == 1 ? t.tx : (idx == 2 ? t.ty : t.tz);
    if (m.dist >= 0.0 && tt >= m.dist) { m.tau = m.tau + (m.dist - m.d) * rhokap; m.live = false; return; }
    m.tau = m.tau + (tt - m.d) * rhokap; m.d = tt;
    if (!trav_cross<VAR>(g, t, idx)) { m.live = false; return; }
    if (m.tau >= LART_TAU_HUGE) { m.live = false; return; }
    const Cell *c = &g.cell[cidx(g, t.ic, t.jc, t.kc)];
    double Dn = c->Dfreq, u2 = c->vfx * t.kx + c->vfy * t.ky + c->vfz * t.kz;
    m.xfreq = (m.xfreq + m.u1) * m.Dold / Dn - u2;
    m.Dold = Dn; m.u1 = u2; m.rk = c->rhokap; m.va = c->voigt_a; m.rD = c->rhokapD;
}
template <int VAR>
__device__ __forceinline__ double raytrace_to_edge(const KArgs &A, const Lds &l, bool dust, const Photon &p, double xfreq, double kx, double ky, double kz, double dist, unsigned &nsteps) {
    EdgeMarch m;
    edge_begin<VAR>(A, l, dust, p, xfreq, kx, ky, kz, dist, m, nsteps);
    while (m.live) edge_step<VAR>(A, l, dust, m, nsteps);
    return m.tau;
}

__device__ inline void tally_escape(const lart_params &par, const DTally &T, const Photon &p) {
    int ix = (int)floor((p.xfreq_ref - par.xfreq_min) / par.dxfreq) + 1;
    if (ix >= 1 && ix <= par.nxfreq) {
        atomicAdd(&T.Jout[ix - 1], p.wgt);
        if (par.save_Jmu) {                                   // add_to_Jmu: raytrace_car.f90:4049-4064
            double mu = p.kz; if (par.xyz_symmetry) mu = fabs(mu);
            int imu = (int)floor((mu - par.mu_min) / par.dmu) + 1;
            imu = clampi(imu, 1, par.nmu);
            atomicAdd(&T.Jmu[(size_t)(ix - 1) + (size_t)par.nxfreq * (imu - 1)], p.wgt);
        }
    }
}

// a band-2 (H-alpha) packet leaves: its frequency is the lab-frame one in reference units already (raytrace_car.f90:1625-1636, raytrace_amr.f90:244-256)
__device__ inline void tally_escape_Ha(const KArgs &A, const Lds &l, Photon &p) {
    p.xfreq_ref = p.xfreq;
    const int ix = (int)floor((p.xfreq_ref - A.par.xfreq_min_Ha) / A.par.dxfreq_Ha) + 1;
    if (ix >= 1 && ix <= A.par.nxfreq_Ha) atomicAdd(&A.t.Jout_Ha[ix - 1], p.wgt);
    aux_add(A.t.scal, l, 3, p.wgt);                            // W_esc2
}
// add_to_J (raytrace_car.f90:3979-4011) + add_to_Pnew (:4015-4045): one path segment of length del in cell `ci`
__device__ inline void jpa_segment(const KArgs &A, const Lds &l, const Photon &p, size_t ci, double rhokap_cell, double Dfreq_cell, double del, double dtauH) {
    if (!(rhokap_cell > 0.0)) return;
    int ib = A.jpa.ind_bin[ci];
    if (ib < 1) return;
    if (A.jpa.calc_J) {
        double xr = p.xfreq * (Dfreq_cell / A.par.Dfreq_ref);
        int ix = (int)floor((xr - A.par.xfreq_min) / A.par.dxfreq) + 1;
        if (ix > 0 && ix <= A.par.nxfreq) jpa_add_J(A, l, ix, ib, del * p.wgt);
    }
    if (A.jpa.calc_Pnew) atomicAdd(&(l.jp_Pn ? l.jp_Pn : A.jpa.P1_new)[ib - 1], dtauH * p.wgt / (rhokap_cell * Dfreq_cell / A.jpa.cross0));
}


// raytrace_to_tau_amr: raytrace_amr.f90:77-259
__device__ __forceinline__ void raytrace_to_tau_amr(const KArgs &A, const Lds &l, bool dust, Photon &p, double tau_in, unsigned &nsteps) {
    const DAmr &a = A.amr;
    double x = p.x, y = p.y, z = p.z, kx = p.kx, ky = p.ky, kz = p.kz, tau = 0.0;
    int il = p.ic;
    if (il <= 0) { il = amr_find_leaf(a, x, y, z); if (il <= 0) { p.inside = false; return; } }
    const Cell *c = &A.g.cell[il - 1];
#ifndef LART_NO_FAST_FLIGHT
    {   // fast path: the flight ends clearly inside the leaf it starts in (see raytrace_to_tau): no exit-face solve (three divisions)
        const AmrGeom q = a.geom[a.icell_of_leaf[il - 1] - 1];
        const bool b2 = l.lt == 8 && p.iband == 2;           // band 2 (H-alpha): dust only, scaled to 6563 A (raytrace_amr.f90:111-127)
        const double kapH0 = b2 ? 0.0 : c->rhokap * calc_voigt(l, p.xfreq, c->voigt_a, c->Dfreq);
        double kap0 = kapH0;
        if (l.lt == 9) kap0 = kap0 + c->rhokap * h2_kappa_D(A.par, A.h2tab, l, p.xfreq, c->Dfreq);
        if (dust) kap0 = kap0 + c->rhokapD;
        if (b2) kap0 = dust ? c->rhokapD * A.par.R_Ha : 0.0;
        if (kap0 > 0.0) {
            const double s = tau_in / kap0, xn = x + s * kx, yn = y + s * ky, zn = z + s * kz, h = q.ch * (1.0 - 1e-9);
            if (fabs(xn - q.cx) < h && fabs(yn - q.cy) < h && fabs(zn - q.cz) < h) {
                nsteps++;
                if (A.jpa.nbins && !b2) jpa_segment_amr(A, l, p, il, *c, x + 0.5 * s * kx, y + 0.5 * s * ky, z + 0.5 * s * kz, s, s * kapH0);
                p.x = xn; p.y = yn; p.z = zn; p.ic = il; p.jc = 1; p.kc = 1;
                return;
            }
        }
    }
#endif
    double u1 = c->vfx * kx + c->vfy * ky + c->vfz * kz;
    int nbound = 0;
    for (;;) {
        int ic = a.icell_of_leaf[il - 1], iface;
        double t_exit = amr_cell_exit(a.geom[ic - 1], x, y, z, kx, ky, kz, iface);
        const bool b2 = l.lt == 8 && p.iband == 2;
        double rhokapH = b2 ? 0.0 : c->rhokap * calc_voigt(l, p.xfreq, c->voigt_a, c->Dfreq), rhokap = rhokapH;
        if (l.lt == 9) rhokap = rhokap + c->rhokap * h2_kappa_D(A.par, A.h2tab, l, p.xfreq, c->Dfreq);
        if (dust) rhokap = rhokap + c->rhokapD;
        if (b2) rhokap = dust ? c->rhokapD * A.par.R_Ha : 0.0;
        nsteps++;
        if (tau + t_exit * rhokap >= tau_in) {
            double d_step = (rhokap > 0.0) ? (tau_in - tau) / rhokap : t_exit;
            if (A.jpa.nbins && !b2) jpa_segment_amr(A, l, p, il, *c, x + 0.5 * d_step * kx, y + 0.5 * d_step * ky, z + 0.5 * d_step * kz, d_step, d_step * rhokapH);
            x = x + d_step * kx; y = y + d_step * ky; z = z + d_step * kz;
            break;
        }
        tau = tau + t_exit * rhokap; x = x + t_exit * kx; y = y + t_exit * ky; z = z + t_exit * kz;
        if (A.jpa.nbins && !b2) jpa_segment_amr(A, l, p, il, *c, x - 0.5 * t_exit * kx, y - 0.5 * t_exit * ky, z - 0.5 * t_exit * kz, t_exit, t_exit * rhokapH);
        int il_new = amr_next_leaf(a, ic, iface, x, y, z); bool reflected = false;
        if (il_new <= 0) { il_new = amr_boundary(A, il, iface, x, y, z, kx, ky, kz, reflected); if (++nbound > LART_AMR_MAX_WRAPS) il_new = 0; }
        if (il_new <= 0) { p.inside = false; break; }
        const Cell *cn = &A.g.cell[il_new - 1];
        double u2 = cn->vfx * kx + cn->vfy * ky + cn->vfz * kz;
        if (!b2) p.xfreq = reflected ? (p.xfreq + u1) - u2 : (p.xfreq + u1) * c->Dfreq / cn->Dfreq - u2;      // band 2 carries a lab-frame frequency
        u1 = u2; il = il_new; c = cn;
    }
    p.x = x; p.y = y; p.z = z; p.kx = kx; p.ky = ky; p.kz = kz; p.ic = il; p.jc = 1; p.kc = 1;
    if (!p.inside && l.lt == 8 && p.iband == 2) tally_escape_Ha(A, l, p);
    else if (!p.inside) {
        u1 = c->vfx * kx + c->vfy * ky + c->vfz * kz;
        p.xfreq = p.xfreq + u1;
        p.xfreq_ref = p.xfreq * (c->Dfreq / A.par.Dfreq_ref);
        tally_escape(A.par, A.t, p);
        if (l.lt == 8) aux_add(A.t.scal, l, 1, p.wgt);      // W_esc1
    }
}

// march to optical depth tau_in; moves the packet, re-expresses xfreq at every crossing, tallies escape
template <int VAR>
__device__ __forceinline__ void raytrace_to_tau(const KArgs &A, const Lds &l, bool dust, Photon &p, double tau_in, unsigned &nsteps, double u_own = 0.0, OvlHold *oh = nullptr) {
    if (VAR == 3) { raytrace_to_tau_amr(A, l, dust, p, tau_in, nsteps); return; }
    if (VAR == 4) { raytrace_to_tau_clump(A, l, dust, p, tau_in, nsteps); return; }
    if (is_ovl(VAR)) { raytrace_to_tau_clump_ovl(A, l, dust, p, tau_in, u_own, nsteps, oh); return; }
    const DGrid &g = A.g;
#ifndef LART_NO_FAST_FLIGHT
    // Fast path: the flight ends inside the cell it starts in (99.6 % of the flights at tau0 = 1e6: the line-centre depth of one cell is
    // ~1e4).  The reference's loop (:1485-1524) would set the traversal up (three divisions and the face fix-ups), accumulate
    // tau = t_face * kappa and back off by (tau - tau_in) / kappa: the same end point up to rounding, tau_in / kappa along k.  Taken only
    // when that point lies clearly inside the cell (1e-9 of the cell size away from every face), so every start on a face and every end
    // near one still goes through the march below and is decided exactly as the reference decides it.
    if (p.ic >= 1 && p.ic <= g.nx && p.jc >= 1 && p.jc <= g.ny && p.kc >= 1 && p.kc <= g.nz) {
        const size_t ci = cidx(g, p.ic, p.jc, p.kc);
        const Cell &c0 = g.cell[ci];
        const double rk0 = c0.rhokap, D0 = c0.Dfreq, vH0 = calc_voigt(l, p.xfreq, c0.voigt_a, D0);
        const bool b2 = l.lt == 8 && p.iband == 2;           // band 2 (H-alpha): dust only, scaled to 6563 A (raytrace_car.f90:1495-1503)
        double kap0 = rk0 * vH0;
        if (l.lt == 9) kap0 = kap0 + rk0 * h2_kappa_D(A.par, A.h2tab, l, p.xfreq, D0);        // :1490-1493
        if (dust) kap0 = kap0 + c0.rhokapD;
        if (b2) kap0 = dust ? c0.rhokapD * A.par.R_Ha : 0.0;
        if (kap0 > 0.0) {
            const double s = tau_in / kap0, xn = p.x + s * p.kx, yn = p.y + s * p.ky, zn = p.z + s * p.kz;
            const double ex = 1e-9 * g.dx, ey = 1e-9 * g.dy, ez = 1e-9 * g.dz;
            bool in = zn > l.zf[p.kc - 1] + ez && zn < l.zf[p.kc] - ez;
            if (VAR != 2) in = in && xn > l.xf[p.ic - 1] + ex && xn < l.xf[p.ic] - ex && yn > l.yf[p.jc - 1] + ey && yn < l.yf[p.jc] - ey;
            if (in) {
                nsteps++;
                if (A.jpa.nbins && !b2) jpa_segment(A, l, p, ci, rk0, D0, s, s * (rk0 * vH0));
                p.x = xn; p.y = yn; p.z = zn;
                return;
            }
        }
    }
#endif
    Trav t; t.kx = p.kx; t.ky = p.ky; t.kz = p.kz; t.ic = p.ic; t.jc = p.jc; t.kc = p.kc;
    double xp = p.x, yp = p.y, zp = p.z;
    if (trav_setup<VAR>(g, l, xp, yp, zp, true, t)) { p.inside = false; return; }
    double tau = 0.0, d = 0.0;
    const Cell *c = &g.cell[cidx(g, t.ic, t.jc, t.kc)];
    double rk = c->rhokap, va = c->voigt_a, rD = c->rhokapD;
    double vx = c->vfx, vy = c->vfy, vz = c->vfz;
    if (VAR == 8) vy = vy + p.shear;                          // :2808-2809 the cell's y velocity + what the packet picked up at the shearing boundary
    double Dold = c->Dfreq, u1 = vx * t.kx + vy * t.ky + vz * t.kz;
    double del = 0.0, vH = 0.0;
    const bool band2 = l.lt == 8 && p.iband == 2;
    for (;;) {
        vH = calc_voigt(l, p.xfreq, va, Dold);
        double rhokap = rk * vH;
        if (l.lt == 9) rhokap = rhokap + rk * h2_kappa_D(A.par, A.h2tab, l, p.xfreq, Dold);
        if (dust) rhokap = rhokap + rD;
        if (band2) { vH = 0.0; rhokap = dust ? rD * A.par.R_Ha : 0.0; }
        nsteps++;
        int idx = (VAR == 2) ? 3 : minloc3(t.tx, t.ty, t.tz);
        double tt = idx == 1 ? t.tx : (idx == 2 ? t.ty : t.tz);
        del = tt - d;
        tau = tau + del * rhokap; d = tt;
        if (tau >= tau_in) {
            if (rhokap > 0.0) { double over = (tau - tau_in) / rhokap; d = d - over; del = del - over; }
            break;
        }
        int io = t.ic, jo = t.jc, ko = t.kc;
        if (!trav_cross<VAR>(g, t, idx)) { p.inside = false; t.ic = io; t.jc = jo; t.kc = ko; break; }
        if (VAR == 8 && idx == 1) {                           // :2843-2852 a packet that wraps in x changes its y velocity by -+ par%Omega (q Omega Lx)
            if (t.is < 0 && io == 1) p.shear = p.shear - A.par.Omega;
            if (t.is > 0 && io == g.nx) p.shear = p.shear + A.par.Omega;
        }
        if (A.jpa.nbins && !band2) jpa_segment(A, l, p, cidx(g, io, jo, ko), rk, Dold, del, del * (rk * vH));      // add_to_J / add_to_Pnew :1579-1584
        c = &g.cell[cidx(g, t.ic, t.jc, t.kc)];
        vx = c->vfx; vy = c->vfy; vz = c->vfz;
        if (VAR == 8) vy = vy + p.shear;
        double Dn = c->Dfreq, u2 = vx * t.kx + vy * t.ky + vz * t.kz;
        if (!band2) p.xfreq = (p.xfreq + u1) * Dold / Dn - u2;
        Dold = Dn; u1 = u2; rk = c->rhokap; va = c->voigt_a; rD = c->rhokapD;
    }
    if (A.jpa.nbins && !band2) jpa_segment(A, l, p, cidx(g, t.ic, t.jc, t.kc), rk, Dold, del, del * (rk * vH));          // last segment :1604-1609
    if (!p.inside && band2) tally_escape_Ha(A, l, p);
    else
    if (!p.inside) {                                          // :1613-1623 fluid frame -> lab frame, escape tally
        // (for VAR 1 the mirrored direction components of the last flight segment are used, :1918)
        u1 = vx * t.kx + vy * t.ky + vz * t.kz;
        p.xfreq = p.xfreq + u1;
        p.xfreq_ref = p.xfreq * (Dold / A.par.Dfreq_ref);
        tally_escape(A.par, A.t, p);
        if (l.lt == 8) aux_add(A.t.scal, l, 1, p.wgt);      // W_esc1 :1624
    }
    if (VAR == 1 || VAR == 5) {                               // :1935-1946 unfolded path, then fold back into the octant (xysym :2238-2248: x and y only)
        p.x = p.x + d * p.kx; p.y = p.y + d * p.ky; p.z = p.z + d * p.kz;
        if (p.x < g.xmin) p.x = -p.x;
        if (p.y < g.ymin) p.y = -p.y;
        if (VAR == 1 && p.z < g.zmin) p.z = -p.z;
        p.kx = t.kx; p.ky = t.ky; p.kz = t.kz;
        p.ic = t.ic; p.jc = t.jc; p.kc = t.kc;
    } else if (is_per(VAR)) {                                 // :2509-2514 (shear :2943-2949) back into the base period (xrange = nx dx)
        if (p.inside) { xp = xp + d * t.kx; yp = yp + d * t.ky; zp = zp + d * t.kz; }
        const double xr = -2.0 * g.xmin, yr = -2.0 * g.ymin;      // grid%xrange = xmax - xmin with xmin = -xmax (grid_mod_car.f90:166-181)
        p.x = xp - floor((xp - g.xmin) / xr) * xr; p.y = yp - floor((yp - g.ymin) / yr) * yr; p.z = zp;
        p.ic = t.ic; p.jc = t.jc; p.kc = t.kc;
    } else {
        if (p.inside) { p.x = xp + d * t.kx; p.y = yp + d * t.ky; p.z = zp + d * t.kz; }
        if (VAR == 0) { p.ic = t.ic; p.jc = t.jc; }
        p.kc = t.kc;
    }
}

// cell-by-cell core-skip threshold (Smith+15 eq. 35): grid_mod_car.f90:1598-1629
// amr_xcrit_local: octree_mod.f90:248-278
__device__ inline void xcrit_local_amr(const KArgs &A, const Photon &p, double &xc, double &xc2) {
    if (A.par.core_skip_global) { xc = A.par.xcrit; xc2 = A.par.xcrit2; return; }
    xc = 0.0; xc2 = 0.0;
    if (p.ic <= 0) return;
    const AmrGeom &q = A.amr.geom[A.amr.icell_of_leaf[p.ic - 1] - 1];
    double dl = fmin(q.ch - fabs(p.x - q.cx), fmin(q.ch - fabs(p.y - q.cy), q.ch - fabs(p.z - q.cz)));
    if (dl <= 0.0) return;
    const Cell &c = A.g.cell[p.ic - 1];
    double atau = c.voigt_a * c.rhokap * dl;
    if (atau > 1.0) { xc = cbrt(atau) / 5.0; xc2 = xc * xc; }
}
__device__ inline void xcrit_local(const lart_params &par, const DGrid &g, const Lds &l, const Photon &p, double &xc, double &xc2) {
    if (par.core_skip_global) { xc = par.xcrit; xc2 = par.xcrit2; return; }
    xc = 0.0; xc2 = 0.0;
    if (p.ic < 1 || p.jc < 1 || p.kc < 1) return;
    double dlx = fmin(p.x - l.xf[p.ic - 1], l.xf[p.ic] - p.x), dly = fmin(p.y - l.yf[p.jc - 1], l.yf[p.jc] - p.y), dlz = fmin(p.z - l.zf[p.kc - 1], l.zf[p.kc] - p.z);
    double dl = fmin(dlx, fmin(dly, dlz));
    if (dl <= 0.0) return;
    const Cell &c = g.cell[cidx(g, p.ic, p.jc, p.kc)];
    double atau = c.voigt_a * c.rhokap * dl;
    if (atau > 1.0) { xc = cbrt(atau) / 5.0; xc2 = xc * xc; }
}

// HEALPix RING pixel (1-based) of a direction: healpix.f90:29-112
__device__ inline int vec2pix(int nside, double vx, double vy, double vz) {
    const double halfpi = 1.570796326794896619231321691639751442099, twothird = 0.6666666666666666666666666666666666666666;
    double dnorm = sqrt(vx * vx + vy * vy + vz * vz), z = vz / dnorm, phi = 0.0;
    if (vx != 0.0 || vy != 0.0) phi = atan2(vy, vx);
    double za = fabs(z);
    if (phi < 0.0) phi = phi + LART_TWOPI;
    double tt = phi / halfpi;
    int nl2 = 2 * nside, nl4 = 4 * nside, ncap = nl2 * (nside - 1), npix = 12 * nside * nside;
    if (za <= twothird) {
        int jp = (int)(nside * (0.5 + tt - z * 0.75)), jm = (int)(nside * (0.5 + tt + z * 0.75));
        int ir = nside + 1 + jp - jm, kshift = ((ir & 1) == 0) ? 1 : 0;
        int ip = (int)((jp + jm - nside + kshift + 1) / 2) + 1;
        if (ip > nl4) ip = ip - nl4;
        return ncap + nl4 * (ir - 1) + ip;
    }
    double tp = tt - (double)(int)tt, tmp = nside * sqrt(3.0 * (1.0 - za));
    int jp = (int)(tp * tmp), jm = (int)((1.0 - tp) * tmp);
    int ir = jp + jm + 1, ip = (int)(tt * ir) + 1;
    if (ip > 4 * ir) ip = ip - 4 * ir;
    return (z <= 0.0) ? npix - 2 * ir * (ir + 1) + ip : 2 * ir * (ir - 1) + ip;
}

// ---------------------------------------------------------------------------------------------
// peel-off (next-event estimator): far observers peelingoff_rect.f90, inside observers peelingoff_heal.f90
// ---------------------------------------------------------------------------------------------
struct PeelDir { double kx, ky, kz, r2, dist; int ix, iy; bool in_image; };   // dist < 0: march to the grid edge
// The peel routines of the reference compute the detector pixel first and the optical depth last.  Here the order is turned
// round: direction -> (cheap, conservative) field-of-view test -> march -> weight; the pixel (two atan2 or vec2pix) and the
// Stokes rotations are only evaluated for a non-zero weight.  Deep inside an optically thick medium e^-tau is exactly 0 for
// almost every event, and a zero weight never reached the tallies before either, so the tallies are unchanged.
__device__ inline PeelDir peel_dir0(const KArgs &A, const DObs &o, const Photon &p) {
    PeelDir q;
    q.kx = o.x - p.x; q.ky = o.y - p.y; q.kz = o.z - p.z;
    q.r2 = q.kx * q.kx + q.ky * q.ky + q.kz * q.kz;
    double r = sqrt(q.r2);
    q.kx = q.kx / r; q.ky = q.ky / r; q.kz = q.kz / r;
    q.dist = A.heal_nside > 0 ? r : -1.0;      // inside observer: tau to the observer (peelingoff_heal.f90:60)
    q.ix = 0; q.iy = 0; q.in_image = true;
    if (A.heal_nside > 0) return q;             // every direction has a HEALPix pixel
    // far observer: clearly outside the image -> no march (as the reference); anything near the border is decided exactly later
    double kx = o.rm[0] * q.kx + o.rm[1] * q.ky + o.rm[2] * q.kz, ky = o.rm[3] * q.kx + o.rm[4] * q.ky + o.rm[5] * q.kz, kz = o.rm[6] * q.kx + o.rm[7] * q.ky + o.rm[8] * q.kz;
    double hx = 0.5 * A.nxim * A.dxim * (1.0 / LART_RAD2DEG), hy = 0.5 * A.nyim * A.dyim * (1.0 / LART_RAD2DEG);
    if (hx < 1.5 && hy < 1.5 && kz > 0.0) {      // (narrow fields only: tan is monotonic there)
        double tx = fabs(kx) / kz, ty = fabs(ky) / kz, mx = hx + 1e-6, my = hy + 1e-6;
        // tan(h) <= h + h^3/3 + 2h^5/15 + h^7 (h < 1.5): an upper bound, so the test never rejects a pixel of the image
        double bx = mx * (1.0 + mx * mx * (1.0 / 3.0 + mx * mx * (2.0 / 15.0 + mx * mx))), by = my * (1.0 + my * my * (1.0 / 3.0 + my * my * (2.0 / 15.0 + my * my)));
        if (mx > 0.7) bx = tan(mx); if (my > 0.7) by = tan(my);
        if (tx > bx * (1.0 + 1e-9) || ty > by * (1.0 + 1e-9)) q.in_image = false;
    }
    return q;
}
// the pixel, exactly as the reference computes it (peelingoff_rect.f90:595-627 / peelingoff_heal.f90:35-45)
__device__ inline void peel_pixel(const KArgs &A, const DObs &o, PeelDir &q) {
    if (A.heal_nside > 0) {
        q.ix = vec2pix(A.heal_nside, -q.kx, -q.ky, -q.kz); q.iy = 1;
        q.in_image = (q.ix >= 1 && q.ix <= A.nxim);
        return;
    }
    double kx = o.rm[0] * q.kx + o.rm[1] * q.ky + o.rm[2] * q.kz, ky = o.rm[3] * q.kx + o.rm[4] * q.ky + o.rm[5] * q.kz, kz = o.rm[6] * q.kx + o.rm[7] * q.ky + o.rm[8] * q.kz;
    q.ix = (int)floor(atan2(-kx, kz) * LART_RAD2DEG / A.dxim + A.nxim / 2.0) + 1;
    q.iy = (int)floor(atan2(-ky, kz) * LART_RAD2DEG / A.dyim + A.nyim / 2.0) + 1;
    q.in_image = (q.ix >= 1 && q.ix <= A.nxim && q.iy >= 1 && q.iy <= A.nyim);
}
__device__ inline size_t pix2(const KArgs &A, int io, int ix, int iy) { return (size_t)io * A.nxim * A.nyim + (size_t)(ix - 1) + (size_t)A.nxim * (size_t)(iy - 1); }
__device__ inline size_t pix3(const KArgs &A, int io, int ixf, int ix, int iy) {
    return ((size_t)io * A.nxim * A.nyim + (size_t)(ix - 1) + (size_t)A.nxim * (size_t)(iy - 1)) * (size_t)A.par.nxfreq + (size_t)(ixf - 1);
}
// e^{-tau} is exactly 0 once tau >= 745.2 in f64: adding it leaves the tally unchanged, so skip the atomics
__device__ inline void add_if(double *arr, size_t i, double v, unsigned &natom) { if (v != 0.0) { atomicAdd(&arr[i], v); natom++; } }

// bulk velocity along the peel direction: the owner clump's in clump mode (peelingoff_rect.f90:65-68, 186-187, 403-404, 534-535, 658-659)
template <int VAR>
__device__ inline double peel_u1(const KArgs &A, const Photon &p, const Cell &c, const PeelDir &q) {
    if (is_clump(VAR) && p.icl > 0) return ulos_clump(A, A.cl.phys[p.icl - 1], q.kx, q.ky, q.kz);
    return c.vfx * q.kx + c.vfy * q.ky + c.vfz * q.kz;
}
// azimuth about the (m,n,k) triad and the peel reference normal: :356-385 (= :204-224)
struct StokesGeom { double cost, sint, cosp, sinp, cos2p, sin2p, nx, ny, nz; };
__device__ inline StokesGeom stokes_geom(const Photon &p, const PeelDir &q) {
    StokesGeom s;
    s.cost = p.kx * q.kx + p.ky * q.ky + p.kz * q.kz; s.sint = sqrt(1.0 - s.cost * s.cost);
    if (s.sint == 0.0) { s.cosp = 1.0; s.sinp = 0.0; s.cos2p = 1.0; s.sin2p = 0.0; }
    else { s.cosp = (q.kx * p.mx + q.ky * p.my + q.kz * p.mz) / s.sint; s.sinp = (q.kx * p.nx + q.ky * p.ny + q.kz * p.nz) / s.sint;
           s.cos2p = 2.0 * s.cosp * s.cosp - 1.0; s.sin2p = 2.0 * s.cosp * s.sinp; }
    s.nx = -s.sinp * p.mx + s.cosp * p.nx; s.ny = -s.sinp * p.my + s.cosp * p.ny; s.nz = -s.sinp * p.mz + s.cosp * p.nz;
    return s;
}
// Every peel routine is: head (direction to the observer, field-of-view test, frequency along that direction) -> optical depth to
// the observer -> weight, pixel, deposit.  The heads are separate functions because the peel kernel marches the rays of many records
// interleaved: it calls the head to start a march (peel_ray_begin) and, when the march of (record, observer `only`) is over, the
// routine itself with `only` and the optical depth it found (`tau_given`); the default arguments give the one-piece routine.
template <int VAR>
__device__ inline bool peel_head_direct(const KArgs &A, const DObs &o, const Photon &p, const Cell &c, PeelDir &q, double &xf, double &xfreq_ref) {
    q = peel_dir0(A, o, p);
    if (!q.in_image) return false;
    xf = p.xfreq;
    if (is_clump(VAR) && p.icl > 0) xfreq_ref = p.xfreq + peel_u1<VAR>(A, p, c, q);      // clump frame -> lab frame along the peel direction
    else if (!A.par.comoving_source) {
        double u1 = c.vfx * p.kx + c.vfy * p.ky + c.vfz * p.kz; xfreq_ref = p.xfreq + u1;
        double u2 = c.vfx * q.kx + c.vfy * q.ky + c.vfz * q.kz; xf = xfreq_ref - u2;
    } else { double u1 = c.vfx * q.kx + c.vfy * q.ky + c.vfz * q.kz; xfreq_ref = p.xfreq + u1; }
    xfreq_ref = xfreq_ref * (c.Dfreq / A.par.Dfreq_ref);
    return true;
}
__device__ inline bool peel_head_res_nostokes(const KArgs &A, const DObs &o, const Photon &p, const Cell &c, double xfreq_atom, double ux, double uy, double uz, double g_rec0,
                                              PeelDir &q, double &cost2, double &xfreq) {
    q = peel_dir0(A, o, p);
    if (!q.in_image) return false;
    double cost = p.kx * q.kx + p.ky * q.ky + p.kz * q.kz, sint; cost2 = cost * cost; sint = sqrt(1.0 - cost2);
    double rho1 = sqrt(1.0 - p.kz * p.kz) * sint, cosp = 1.0, sinp = 0.0;
    if (rho1 != 0.0) { double rho = 1.0 / rho1; cosp = rho * (cost * p.kz - q.kz); sinp = rho * (p.kx * q.ky - q.kx * p.ky); }
    xfreq = xfreq_atom + (ux * cosp + uy * sinp) * sint + uz * cost;
    if (A.par.recoil) xfreq = xfreq - (g_rec0 / c.Dfreq) * (1.0 - cost);
    return true;
}
__device__ inline bool peel_head_res_stokes(const KArgs &A, const DObs &o, const Photon &p, const Cell &c, double xfreq_atom, double ux, double uy, double uz, double g_rec0,
                                            PeelDir &q, StokesGeom &s, double &xfreq) {
    q = peel_dir0(A, o, p);
    if (!q.in_image) return false;
    s = stokes_geom(p, q);
    xfreq = xfreq_atom + (ux * s.cosp + uy * s.sinp) * s.sint + uz * s.cost;
    if (A.par.recoil) xfreq = xfreq - (g_rec0 / c.Dfreq) * (1.0 - s.cost);
    return true;
}
// peeling_direct_outside: peelingoff_rect.f90:24-129
template <int VAR>
__device__ inline void peel_direct(const KArgs &A, const Lds &l, bool dust, const Photon &p, unsigned &nsteps, unsigned &natom, int only = -1, double tau_given = 0.0) {
    const Cell &c = A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)];
    for (int i = 0; i < A.nobs; i++) {
        if (only >= 0 && i != only) continue;
        PeelDir q; double xf, xfreq_ref;
        if (!peel_head_direct<VAR>(A, A.obs[i], p, c, q, xf, xfreq_ref)) continue;
        int ixf = (int)floor((xfreq_ref - A.par.xfreq_min) / A.par.dxfreq) + 1;
        double tau = only >= 0 ? tau_given : raytrace_to_edge<VAR>(A, l, dust, p, xf, q.kx, q.ky, q.kz, q.dist, nsteps);
        const double wgt0 = 1.0 / (LART_FOURPI * q.r2) * p.wgt, wgt = d_exp_neg(-tau) * wgt0;      // wgt0: the unattenuated source (direc0)
        if (wgt == 0.0 && !A.par.save_direc0) continue;
        peel_pixel(A, A.obs[i], q);
        if (!q.in_image) continue;
        if (A.par.save_peeloff_2D) { add_if(A.t.c2[1], pix2(A, i, q.ix, q.iy), wgt, natom); if (A.par.use_stokes) add_if(A.t.c2[2], pix2(A, i, q.ix, q.iy), wgt, natom);
                                     if (A.par.save_direc0) add_if(A.t.c2[6], pix2(A, i, q.ix, q.iy), wgt0, natom); }
        if (A.par.save_peeloff_3D && ixf >= 1 && ixf <= A.par.nxfreq) { add_if(A.t.c3[1], pix3(A, i, ixf, q.ix, q.iy), wgt, natom); if (A.par.use_stokes) add_if(A.t.c3[2], pix3(A, i, ixf, q.ix, q.iy), wgt, natom);
                                     if (A.par.save_direc0) add_if(A.t.c3[6], pix3(A, i, ixf, q.ix, q.iy), wgt0, natom); }
    }
}
// peeling_resonance_nostokes_outside: :576-690
template <int VAR>
__device__ inline void peel_resonance_nostokes(const KArgs &A, const Lds &l, bool dust, const Photon &p, double xfreq_atom, double ux, double uy, double uz, double E1, double E2, double g_rec0, unsigned &nsteps, unsigned &natom,
                                               int only = -1, double tau_given = 0.0) {
    const Cell &c = A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)];
    for (int i = 0; i < A.nobs; i++) {
        if (only >= 0 && i != only) continue;
        PeelDir q; double cost2, xfreq;
        if (!peel_head_res_nostokes(A, A.obs[i], p, c, xfreq_atom, ux, uy, uz, g_rec0, q, cost2, xfreq)) continue;
        double tau = only >= 0 ? tau_given : raytrace_to_edge<VAR>(A, l, dust, p, xfreq, q.kx, q.ky, q.kz, q.dist, nsteps);
        double peel = 0.75 * E1 * (cost2 + 1.0) + E2;
        double wgt = peel / (LART_FOURPI * q.r2) * d_exp_neg(-tau) * p.wgt;
        if (wgt == 0.0) continue;
        peel_pixel(A, A.obs[i], q);
        if (!q.in_image) continue;
        double u1 = peel_u1<VAR>(A, p, c, q);
        double xfreq_ref = (xfreq + u1) * (c.Dfreq / A.par.Dfreq_ref);
        int ixf = (int)floor((xfreq_ref - A.par.xfreq_min) / A.par.dxfreq) + 1;
        if (A.par.save_peeloff_2D) add_if(A.t.c2[0], pix2(A, i, q.ix, q.iy), wgt, natom);
        if (A.par.save_peeloff_3D && ixf >= 1 && ixf <= A.par.nxfreq) add_if(A.t.c3[0], pix3(A, i, ixf, q.ix, q.iy), wgt, natom);
    }
}
__device__ inline void stokes_deposit(const KArgs &A, int io, const PeelDir &q, int ixf, double wgt, double Id, double Qd, double Ud, double Vd, unsigned &natom) {
    if (wgt == 0.0) return;
    if (A.par.save_peeloff_2D) { size_t s = pix2(A, io, q.ix, q.iy);
        atomicAdd(&A.t.c2[0][s], wgt * Id); atomicAdd(&A.t.c2[2][s], wgt * Id); atomicAdd(&A.t.c2[3][s], wgt * Qd); atomicAdd(&A.t.c2[4][s], wgt * Ud); atomicAdd(&A.t.c2[5][s], wgt * Vd); natom += 5; }
    if (A.par.save_peeloff_3D && ixf >= 1 && ixf <= A.par.nxfreq) { size_t s = pix3(A, io, ixf, q.ix, q.iy);
        atomicAdd(&A.t.c3[0][s], wgt * Id); atomicAdd(&A.t.c3[2][s], wgt * Id); atomicAdd(&A.t.c3[3][s], wgt * Qd); atomicAdd(&A.t.c3[4][s], wgt * Ud); atomicAdd(&A.t.c3[5][s], wgt * Vd); natom += 5; }
}
// peeling_resonance_stokes_outside: :303-482
template <int VAR>
__device__ inline void peel_resonance_stokes(const KArgs &A, const Lds &l, bool dust, const Photon &p, double xfreq_atom, double ux, double uy, double uz, double E1, double E2, double E3, double g_rec0, unsigned &nsteps, unsigned &natom,
                                             int only = -1, double tau_given = 0.0) {
    const Cell &c = A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)];
    for (int i = 0; i < A.nobs; i++) {
        if (only >= 0 && i != only) continue;
        const DObs &o = A.obs[i];
        PeelDir q; StokesGeom s; double xfreq;
        if (!peel_head_res_stokes(A, o, p, c, xfreq_atom, ux, uy, uz, g_rec0, q, s, xfreq)) continue;
        double tau = only >= 0 ? tau_given : raytrace_to_edge<VAR>(A, l, dust, p, xfreq, q.kx, q.ky, q.kz, q.dist, nsteps);
        double wgt = 1.0 / q.r2 * d_exp_neg(-tau) * p.wgt;
        if (wgt == 0.0) continue;
        peel_pixel(A, o, q);
        if (!q.in_image) continue;
        double cost2 = s.cost * s.cost;
        double S22 = 0.75 * E1 * (cost2 + 1.0), S11 = S22 + E2, S12 = 0.75 * E1 * (cost2 - 1.0), S33 = 1.5 * E1 * s.cost, S44 = 1.5 * E3 * s.cost;
        double u1 = peel_u1<VAR>(A, p, c, q);
        double xfreq_ref = (xfreq + u1) * (c.Dfreq / A.par.Dfreq_ref);
        int ixf = (int)floor((xfreq_ref - A.par.xfreq_min) / A.par.dxfreq) + 1;
        double Q0 = s.cos2p * p.Q + s.sin2p * p.U, U0 = -s.sin2p * p.Q + s.cos2p * p.U;
        double Iobs = (S11 + S12 * Q0) / LART_FOURPI, Qobs = (S12 + S22 * Q0) / LART_FOURPI, Uobs = (S33 * U0) / LART_FOURPI, Vobs = (S44 * p.V) / LART_FOURPI;
        double cosg = -(o.rm[0] * s.nx + o.rm[1] * s.ny + o.rm[2] * s.nz), sing = o.rm[3] * s.nx + o.rm[4] * s.ny + o.rm[5] * s.nz;
        double cos2g = 2.0 * cosg * cosg - 1.0, sin2g = 2.0 * cosg * sing;
        double Qdet = cos2g * Qobs + sin2g * Uobs, Udet = -sin2g * Qobs + cos2g * Uobs;
        stokes_deposit(A, i, q, ixf, wgt, Iobs, Qdet, Udet, Vobs, natom);
    }
}
// peeling_dust_nostokes_outside: :484-574
template <int VAR>
__device__ inline void peel_dust_nostokes(const KArgs &A, const Lds &l, const Photon &p, unsigned &nsteps, unsigned &natom, int only = -1, double tau_given = 0.0) {
    const Cell &c = A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)];
    for (int i = 0; i < A.nobs; i++) {
        if (only >= 0 && i != only) continue;
        PeelDir q = peel_dir0(A, A.obs[i], p);
        if (!q.in_image) continue;
        double tau = only >= 0 ? tau_given : raytrace_to_edge<VAR>(A, l, true, p, p.xfreq, q.kx, q.ky, q.kz, q.dist, nsteps);
        double cosa = p.kx * q.kx + p.ky * q.ky + p.kz * q.kz, hgg = A.par.hgg;
        double b = (1.0 + hgg * hgg) - 2.0 * hgg * cosa;
        double peel = (1.0 - hgg * hgg) / (b * sqrt(b)) / LART_FOURPI;
        double wgt = peel / q.r2 * d_exp_neg(-tau) * p.wgt;
        if (wgt == 0.0) continue;
        peel_pixel(A, A.obs[i], q);
        if (!q.in_image) continue;
        double u1 = peel_u1<VAR>(A, p, c, q);
        double xfreq_ref = (p.xfreq + u1) * (c.Dfreq / A.par.Dfreq_ref);
        int ixf = (int)floor((xfreq_ref - A.par.xfreq_min) / A.par.dxfreq) + 1;
        if (A.par.save_peeloff_2D) add_if(A.t.c2[0], pix2(A, i, q.ix, q.iy), wgt, natom);
        if (A.par.save_peeloff_3D && ixf >= 1 && ixf <= A.par.nxfreq) { add_if(A.t.c3[0], pix3(A, i, ixf, q.ix, q.iy), wgt, natom);
                                                                          if (A.par.save_dust_scattered) add_if(A.t.c3[7], pix3(A, i, ixf, q.ix, q.iy), wgt, natom); }
    }
}
// peeling_dust_stokes_outside: :131-299
template <int VAR>
__device__ inline void peel_dust_stokes(const KArgs &A, const Lds &l, const Photon &p, unsigned &nsteps, unsigned &natom, int only = -1, double tau_given = 0.0) {
    const Cell &c = A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)];
    for (int i = 0; i < A.nobs; i++) {
        if (only >= 0 && i != only) continue;
        const DObs &o = A.obs[i];
        PeelDir q = peel_dir0(A, o, p);
        if (!q.in_image) continue;
        double tau = only >= 0 ? tau_given : raytrace_to_edge<VAR>(A, l, true, p, p.xfreq, q.kx, q.ky, q.kz, q.dist, nsteps);
        double wgt = 1.0 / q.r2 * d_exp_neg(-tau) * p.wgt;
        if (wgt == 0.0) continue;
        peel_pixel(A, o, q);
        if (!q.in_image) continue;
        double u1 = peel_u1<VAR>(A, p, c, q);
        double xfreq_ref = (p.xfreq + u1) * (c.Dfreq / A.par.Dfreq_ref);
        int ixf = (int)floor((xfreq_ref - A.par.xfreq_min) / A.par.dxfreq) + 1;
        StokesGeom s = stokes_geom(p, q);
        const DMueller &m = A.mu;
        double S11 = interp_eq(m.nPDF, m.coss, m.S11, s.cost), S12 = interp_eq(m.nPDF, m.coss, m.S12, s.cost);
        double S33 = interp_eq(m.nPDF, m.coss, m.S33, s.cost), S34 = interp_eq(m.nPDF, m.coss, m.S34, s.cost);
        double Q0 = s.cos2p * p.Q + s.sin2p * p.U, U0 = -s.sin2p * p.Q + s.cos2p * p.U;
        double Iobs = (S11 * 1.0 + S12 * Q0) / LART_TWOPI, Qobs = (S12 * 1.0 + S11 * Q0) / LART_TWOPI;
        double Uobs = (S33 * U0 + S34 * p.V) / LART_TWOPI, Vobs = (-S34 * U0 + S33 * p.V) / LART_TWOPI;
        double cosg = -(o.rm[0] * s.nx + o.rm[1] * s.ny + o.rm[2] * s.nz), sing = o.rm[3] * s.nx + o.rm[4] * s.ny + o.rm[5] * s.nz;
        double cos2g = 2.0 * cosg * cosg - 1.0, sin2g = 2.0 * cosg * sing;
        double Qdet = cos2g * Qobs + sin2g * Uobs, Udet = -sin2g * Qobs + cos2g * Uobs;
        stokes_deposit(A, i, q, ixf, wgt, Iobs, Qdet, Udet, Vobs, natom);
        if (A.par.save_dust_scattered && A.par.save_peeloff_3D && ixf >= 1 && ixf <= A.par.nxfreq) add_if(A.t.c3[7], pix3(A, i, ixf, q.ix, q.iy), wgt * Iobs, natom);   // :284-287
    }
}
// the head of the routine for `kind`, for one observer: false = nothing to march (outside the image); else the ray (q, frequency xf in the
// frame of the packet's cell, whether dust opacity counts along it)
template <int VAR, bool STOKES>
__device__ __forceinline__ bool peel_ray_begin(const KArgs &A, int kind, const Photon &p, double ux, double uy, double uz, double grec, const DObs &o, bool dust, PeelDir &q, double &xf, bool &ray_dust) {
    const Cell &c = A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)];
    if (kind == PEEL_DIRECT) { double xfreq_ref; ray_dust = dust; return peel_head_direct<VAR>(A, o, p, c, q, xf, xfreq_ref); }
    if (kind == PEEL_DUST) { ray_dust = true; q = peel_dir0(A, o, p); xf = p.xfreq; return q.in_image; }
    ray_dust = dust;
    if (STOKES) { StokesGeom s; return peel_head_res_stokes(A, o, p, c, p.xfreq, ux, uy, uz, grec, q, s, xf); }
    double cost2; return peel_head_res_nostokes(A, o, p, c, p.xfreq, ux, uy, uz, grec, q, cost2, xf);
}

// Is every peel ray of this event extinguished inside the cell (leaf, clump) it starts in?  Deep inside an optically thick medium the
// line-centre depth of one cell is ~1e4, and exp(-tau) is exactly 0 in f64 from tau = 745.2 on (the reference's tau_huge cut): such an
// event can deposit nothing, so it is not queued at all.  The test is the first step of the march without its divisions:
// tau_1 = kappa * min_i(d_i / |k_i|) >= 745.2  <=>  kappa * d_i >= 745.2 |k_i| on the three axes (d_i = distance to the face the ray
// moves towards).  Any other case -- an observer inside the medium, a start in the vacuum between clumps, the direct light of a
// birth -- is queued and decided by the march itself.
// the field-of-view bounds of peel_dir0 (tan of the half image size + 1e-6, its polynomial upper bound below 0.7 rad), once per kernel
__device__ inline void lds_set_fov(Lds &l, const KArgs &A) {
    if (A.heal_nside > 0) return;
    const double hx = 0.5 * A.nxim * A.dxim * (1.0 / LART_RAD2DEG), hy = 0.5 * A.nyim * A.dyim * (1.0 / LART_RAD2DEG);
    if (!(hx < 1.5 && hy < 1.5)) return;
    const double mx = hx + 1e-6, my = hy + 1e-6;
    double bx = mx * (1.0 + mx * mx * (1.0 / 3.0 + mx * mx * (2.0 / 15.0 + mx * mx))), by = my * (1.0 + my * my * (1.0 / 3.0 + my * my * (2.0 / 15.0 + my * my)));
    if (mx > 0.7) bx = tan(mx); if (my > 0.7) by = tan(my);
    l.fov_bx = uniform_f64(bx); l.fov_by = uniform_f64(by);
}
// 1 / sqrt(x) to rounding level without the division sequence: the hardware estimate and two Newton steps (the prefilter below only)
__device__ inline double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y * fma(-0.5 * x * y, y, 1.5);
}
// opacity constants of the cell (leaf, owner clump) an event happens in, as scatter_pre read them: handed to the peel prefilter so that it does not gather the record again
// (the gather sat exposed in front of the record stores: 3 waves per SIMD do not hide it)
struct PeelCell { double rhokap, rhokapD, voigt_a, Dfreq, uni; };      // uni: Cell::pad -- Cartesian medium: cells of the uniform cube around the cell; octree: the length a leaf's box can grow by (lart_gpu.hip, uniform_cubes / uniform_margins_amr)
template <int VAR, bool STOKES>
__device__ __forceinline__ bool peel_all_opaque(const KArgs &A, const Lds &l, int kind, const Photon &p, double xa, double ux, double uy, double uz, double grec, bool dust, const PeelCell *pc = nullptr) {
    if (kind == PEEL_DIRECT || A.heal_nside > 0 || VAR == 1) return false;
    if (is_ovl(VAR)) return false;                           // the ray may cross any number of clumps from its first metre on
    if (VAR == 4 && p.icl <= 0) return false;
    if (VAR != 3 && VAR != 4 && (p.ic < 1 || p.ic > A.g.nx || p.jc < 1 || p.jc > A.g.ny || p.kc < 1 || p.kc > A.g.nz)) return false;
    // Round 4: the test no longer runs the head of the peel routine (peel_ray_begin: two square roots, seven divisions, the polynomial of the field-of-view
    // bound -- 20 % of the C3 transport kernel's time, profiles/r04_sections.md) but a division-free form of the same quantities.  It only has to be
    // CONSERVATIVE -- a record that is queued needlessly is marched and found opaque by the peel kernel, which computes everything as the reference does --
    // so the threshold carries a margin of 1e-9 (the forms differ from the head's by a few ulp; a ray this close to 745.2 would deposit exp(-745.2) ~ 5e-324
    // of its weight): nothing that the exact test would queue is dropped here.
    //   direction q = (o - r) / |o - r| by a Newton-refined reciprocal square root;
    //   frequency along q: x_atom + u_x cos(phi) sin(theta) + u_y sin(phi) sin(theta) + u_z cos(theta) with cos(phi) sin(theta) = q.m, sin(phi) sin(theta) = q.n,
    //   cos(theta) = q.k in the Stokes frame (stokes_geom), and the corresponding projections on the frame of peel_head_res_nostokes otherwise.
    const double T = LART_TAU_HUGE * (1.0 + 1e-9);
    PeelCell cc;
    if (pc) cc = *pc;
    else if (VAR == 4) { const ClumpPhys ph = A.cl.phys[p.icl - 1]; cc.rhokap = ph.rhokap; cc.rhokapD = ph.rhokapD; cc.voigt_a = ph.voigt_a; cc.Dfreq = ph.Dfreq; cc.uni = 0.0; }
    else { const Cell &c = A.g.cell[(VAR == 3 || is_clump(VAR)) ? (size_t)(p.ic - 1) : cidx(A.g, p.ic, p.jc, p.kc)]; cc.rhokap = c.rhokap; cc.rhokapD = c.rhokapD; cc.voigt_a = c.voigt_a; cc.Dfreq = c.Dfreq; cc.uni = is_clump(VAR) ? 0.0 : c.pad; }
    const double Dfreq_bg = (VAR == 4) ? A.g.cell[0].Dfreq : cc.Dfreq;      // (the recoil term of the peel head uses the background record of a clump medium, as peel_head_res_* do)
    for (int i = 0; i < A.nobs; i++) {
        const DObs o = l.obs ? l.obs[i] : A.obs[i];          // (LDS copy: this test runs on every event and must not wait for memory)
        double qx = o.x - p.x, qy = o.y - p.y, qz = o.z - p.z;
        const double rinv = rsqrt_nr(qx * qx + qy * qy + qz * qz);
        qx = qx * rinv; qy = qy * rinv; qz = qz * rinv;
        if (l.fov_bx > 0.0) {                                // clearly outside the image: no march, no deposit (as peel_dir0, with the margin on the other side)
            const double kx = o.rm[0] * qx + o.rm[1] * qy + o.rm[2] * qz, ky = o.rm[3] * qx + o.rm[4] * qy + o.rm[5] * qz, kz = o.rm[6] * qx + o.rm[7] * qy + o.rm[8] * qz;
            if (kz > 0.0 && (fabs(kx) > l.fov_bx * kz * (1.0 + 3e-9) || fabs(ky) > l.fov_by * kz * (1.0 + 3e-9))) continue;
        }
        double xf = xa; bool rd = dust;
        if (kind == PEEL_DUST) rd = true;
        else {
            const double cost = p.kx * qx + p.ky * qy + p.kz * qz;
            if (STOKES) xf = xa + ux * (qx * p.mx + qy * p.my + qz * p.mz) + uy * (qx * p.nx + qy * p.ny + qz * p.nz) + uz * cost;
            else {
                const double s2 = 1.0 - p.kz * p.kz;
                if (!(s2 > 1e-12)) return false;             // (k along z: the head's special case decides)
                xf = xa + (ux * (cost * p.kz - qz) + uy * (p.kx * qy - qx * p.ky)) * rsqrt_nr(s2) + uz * cost;
            }
            if (A.par.recoil) xf = xf - (grec / Dfreq_bg) * (1.0 - cost);
        }
        if (VAR == 4) {
            ClumpPhys ph; ph.rhokap = cc.rhokap; ph.rhokapD = cc.rhokapD; ph.voigt_a = cc.voigt_a; ph.Dfreq = cc.Dfreq;      // (kappa_clump reads these four)
            if (!(kappa_clump(A, l, rd, xf, ph) * clump_exit_dist(A.cl.geo[p.icl - 1], p.x, p.y, p.z, qx, qy, qz) >= T)) return false;
            continue;
        }
        double kap = cc.rhokap * calc_voigt(l, xf, cc.voigt_a, cc.Dfreq);
        if (rd) kap = kap + cc.rhokapD;
        double dx, dy, dz;
        if (VAR == 3) {
            const AmrGeom g = A.amr.geom[A.amr.icell_of_leaf[p.ic - 1] - 1];
            const double gh = g.ch + cc.uni;                     // the leaf's box grown by the margin in which the medium is this leaf's (Cell::pad, uniform_margins_amr)
            dx = qx > 0.0 ? (g.cx + gh) - p.x : p.x - (g.cx - gh); dy = qy > 0.0 ? (g.cy + gh) - p.y : p.y - (g.cy - gh); dz = qz > 0.0 ? (g.cz + gh) - p.z : p.z - (g.cz - gh);
        } else {
            // Round 4: not the rest of the first cell but the way out of the largest cube of copies of this cell around it (Cell::pad, computed by lart_gpu_set_grid_car): the
            // ray keeps its frequency and its opacity until it has crossed R more cells on one axis, so kappa x (that path) is what the march would add up, cell by cell
            const int R = (int)cc.uni;
            dx = qx > 0.0 ? l.xf[p.ic + R] - p.x : p.x - l.xf[p.ic - 1 - R]; dy = qy > 0.0 ? l.yf[p.jc + R] - p.y : p.y - l.yf[p.jc - 1 - R]; dz = qz > 0.0 ? l.zf[p.kc + R] - p.z : p.z - l.zf[p.kc - 1 - R];
        }
        bool opaque = kap * dz >= T * fabs(qz);
        if (VAR != 2) opaque = opaque && kap * dx >= T * fabs(qx) && kap * dy >= T * fabs(qy);
        if (!opaque) return false;
    }
    return true;
}

// one record into the peel queue (see DPeelQ).  Called from divergent code: the lanes of the wave that are here right now share one
// atomic for their slots.  xa = xfreq_atom for a resonance event, the packet's xfreq for a dust event or the direct (birth) peel.
template <int VAR, bool STOKES>
__device__ __forceinline__ void peel_emit(const KArgs &A, const Lds &l, bool dust, int kind, const Photon &p, double xa, double ux, double uy, double uz, double E1, double E2, double E3, double grec, unsigned &npeel, const PeelCell *pc = nullptr) {
#ifndef LART_NO_PEEL_PREFILTER
    if (peel_all_opaque<VAR, STOKES>(A, l, kind, p, xa, ux, uy, uz, grec, dust, pc)) { npeel += (unsigned)A.nobs; return; }   // (the one cell each ray was tested in)
#endif
    const unsigned long long m = __ballot(1);
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) { base = *l.pq_cnt; *l.pq_cnt = base + (unsigned)__popcll(m); }      // the wave's own counter: no atomic needed, the wave runs in lockstep
    base = __shfl(base, leader, 64);
    const unsigned k = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    if (k >= A.pq.region) return;                             // cannot happen: the host sizes a region for 64 lanes x (events + 1) records
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, r = wave * A.pq.region + k;
    double *rec = A.pq.f + r * pq_recd(STOKES);                // 16-byte aligned: hipMalloc base, record sizes of 144 / 224 B
    double2 *r2 = reinterpret_cast<double2 *>(rec);
    r2[PQ_X / 2] = make_double2(p.x, p.y); r2[PQ_Z / 2] = make_double2(p.z, p.kx); r2[PQ_KY / 2] = make_double2(p.ky, p.kz); r2[PQ_WGT / 2] = make_double2(p.wgt, xa);
    // what a single line (line type 1) would only repeat is left out: E1..E3 are line%E1..E3 and the recoil constant line%g_recoil0 there
    // (lart_params), and the third vector of the right-handed triad is n = k x m (peel_load)
    if (kind == PEEL_RESONANCE) { r2[PQ_UX / 2] = make_double2(ux, uy); r2[PQ_UZ / 2] = make_double2(uz, grec);
        if (l.lt != 1) { r2[PQ_E1 / 2] = make_double2(E1, E2); rec[PQ_E3] = E3; } }
    *reinterpret_cast<int4 *>(rec + PQ_INTS) = make_int4(p.ic, p.jc, is_clump(VAR) ? p.icl : p.kc, kind);
    if (STOKES) { r2[PQ_MX / 2] = make_double2(p.mx, p.my); r2[PQ_MZ / 2] = make_double2(p.mz, p.Q); r2[PQ_U / 2] = make_double2(p.U, p.V);
                  if (peel_record_has_n<VAR>(A.par)) { r2[PQ_NX / 2] = make_double2(p.nx, p.ny); rec[PQ_NZ] = p.nz; } }
}

// ---------------------------------------------------------------------------------------------
// line type 8 ('ly_beta'), band 2 = H-alpha: the sight line is dust only and a band-2 packet only meets dust.  A history has one conversion
// at most and a handful of band-2 dust events, so these peels march inside the transport kernel instead of going through the peel queue.
// ---------------------------------------------------------------------------------------------
// raytrace_to_edge_car_dust_Ha (raytrace_car.f90:510-582) / raytrace_to_edge_amr_dust_Ha (raytrace_amr.f90:353-390)
template <int VAR>
__device__ inline double edge_dust_Ha(const KArgs &A, const Lds &l, bool dust, const Photon &p, double kx, double ky, double kz, unsigned &nsteps) {
    double tau = 0.0;
    if (!dust) return tau;
    if (VAR == 3) {
        const DAmr &a = A.amr; double x = p.x, y = p.y, z = p.z; int il = p.ic;
        if (il <= 0) { il = amr_find_leaf(a, x, y, z); if (il <= 0) return tau; }
        for (;;) {
            const int ic = a.icell_of_leaf[il - 1]; int iface;
            const double t_exit = amr_cell_exit(a.geom[ic - 1], x, y, z, kx, ky, kz, iface);
            nsteps++;
            tau = tau + t_exit * (A.g.cell[il - 1].rhokapD * A.par.R_Ha);
            if (tau >= LART_TAU_HUGE) return tau;
            x = x + t_exit * kx; y = y + t_exit * ky; z = z + t_exit * kz;
            const int il_new = amr_next_leaf(a, ic, iface, x, y, z);
            if (il_new <= 0) return tau;
            il = il_new;
        }
    }
    const DGrid &g = A.g;
    Trav t; t.kx = kx; t.ky = ky; t.kz = kz; t.ic = p.ic; t.jc = p.jc; t.kc = p.kc;
    double xp = p.x, yp = p.y, d = 0.0;
    if (trav_setup<0>(g, l, xp, yp, p.z, false, t)) return tau;
    for (;;) {
        const double rhokap = g.cell[cidx(g, t.ic, t.jc, t.kc)].rhokapD * A.par.R_Ha;
        nsteps++;
        const int idx = minloc3(t.tx, t.ty, t.tz);
        const double tt = idx == 1 ? t.tx : (idx == 2 ? t.ty : t.tz);
        tau = tau + (tt - d) * rhokap; d = tt;
        if (!trav_cross<0>(g, t, idx)) break;
        if (tau >= LART_TAU_HUGE) break;
    }
    return tau;
}
__device__ inline size_t pix3_Ha(const KArgs &A, int io, int ixf, int ix, int iy) {
    return ((size_t)io * A.nxim * A.nyim + (size_t)(ix - 1) + (size_t)A.nxim * (size_t)(iy - 1)) * (size_t)A.par.nxfreq_Ha + (size_t)(ixf - 1);
}
// peeling_conversion_Ha_outside: peelingoff_rect.f90:692-792 (peelingoff_amr.f90 twin): the newborn H-alpha packet, emitted at line centre in the
// atom frame; dipole phase function of the conversion channel (E1, E2); dust-only sight line; no recoil; into observer%peel_Ha
template <int VAR>
__device__ inline void peel_conversion_Ha(const KArgs &A, const Lds &l, bool dust, const Photon &p, const Cell &c, double ux, double uy, double uz, double E1, double E2, unsigned &nsteps, unsigned &natom) {
    if (!A.t.peel_Ha) return;
    for (int i = 0; i < A.nobs; i++) {
        PeelDir q = peel_dir0(A, A.obs[i], p);
        if (!q.in_image) continue;
        peel_pixel(A, A.obs[i], q);
        if (!q.in_image) continue;
        const double cost = p.kx * q.kx + p.ky * q.ky + p.kz * q.kz, cost2 = cost * cost, sint = sqrt(1.0 - cost2), rho1 = sqrt(1.0 - p.kz * p.kz) * sint;
        double cosp = 1.0, sinp = 0.0;
        if (rho1 != 0.0) { const double rho = 1.0 / rho1; cosp = rho * (cost * p.kz - q.kz); sinp = rho * (p.kx * q.ky - q.kx * p.ky); }
        const double xfreq = (ux * cosp + uy * sinp) * sint + uz * cost;
        const double u1 = c.vfx * q.kx + c.vfy * q.ky + c.vfz * q.kz, xfreq_ref = (xfreq + u1) * (c.Dfreq / A.par.Dfreq_ref);
        const int ixf = (int)floor((xfreq_ref - A.par.xfreq_min_Ha) / A.par.dxfreq_Ha) + 1;
        const double tau = edge_dust_Ha<VAR>(A, l, dust, p, q.kx, q.ky, q.kz, nsteps);
        const double peel = 0.75 * E1 * (cost2 + 1.0) + E2, wgt = peel / (LART_FOURPI * q.r2) * d_exp_neg(-tau) * p.wgt;
        if (ixf >= 1 && ixf <= A.par.nxfreq_Ha) add_if(A.t.peel_Ha, pix3_Ha(A, i, ixf, q.ix, q.iy), wgt, natom);
    }
}
// peeling_dust_Ha_outside: :794-870 (HG phase function of par%hgg_Ha; the packet's lab-frame frequency as it is)
template <int VAR>
__device__ inline void peel_dust_Ha(const KArgs &A, const Lds &l, bool dust, const Photon &p, unsigned &nsteps, unsigned &natom) {
    if (!A.t.peel_Ha) return;
    for (int i = 0; i < A.nobs; i++) {
        PeelDir q = peel_dir0(A, A.obs[i], p);
        if (!q.in_image) continue;
        peel_pixel(A, A.obs[i], q);
        if (!q.in_image) continue;
        const int ixf = (int)floor((p.xfreq - A.par.xfreq_min_Ha) / A.par.dxfreq_Ha) + 1;
        const double tau = edge_dust_Ha<VAR>(A, l, dust, p, q.kx, q.ky, q.kz, nsteps);
        const double cosa = p.kx * q.kx + p.ky * q.ky + p.kz * q.kz, gg = A.par.hgg_Ha;
        const double peel = (1.0 - gg * gg) / pow((1.0 + gg * gg) - 2.0 * gg * cosa, 1.5) / LART_FOURPI, wgt = peel / q.r2 * d_exp_neg(-tau) * p.wgt;
        if (ixf >= 1 && ixf <= A.par.nxfreq_Ha) add_if(A.t.peel_Ha, pix3_Ha(A, i, ixf, q.ix, q.iy), wgt, natom);
    }
}
// ---------------------------------------------------------------------------------------------
// scattering: scattering_car.f90
// ---------------------------------------------------------------------------------------------
// closed-form rotation of k by (theta, phi) about the old k: :795-809
__device__ inline void rotate_k(Photon &p, double cost, double sint, double cosp, double sinp) {
    if (fabs(p.kz) >= 0.99999999999) { p.kx = sint * cosp; p.ky = sint * sinp; p.kz = cost; }
    else {
        double kx1 = p.kx, ky1 = p.ky, kz1 = p.kz, kr = sqrt(kx1 * kx1 + ky1 * ky1), s_kr = sint / kr;   // one division for both components
        p.kx = cost * kx1 + (kz1 * kx1 * cosp - ky1 * sinp) * s_kr;
        p.ky = cost * ky1 + (kz1 * ky1 * cosp + kx1 * sinp) * s_kr;
        p.kz = cost * kz1 - sint * cosp * kr;
    }
}
// (m,n,k) triad rotation, fixed order: :470-484
__device__ inline void rotate_triad(Photon &p, double cost, double sint, double cosp, double sinp) {
    double px = cosp * p.mx + sinp * p.nx, py = cosp * p.my + sinp * p.ny, pz = cosp * p.mz + sinp * p.nz;
    p.nx = cosp * p.nx - sinp * p.mx; p.ny = cosp * p.ny - sinp * p.my; p.nz = cosp * p.nz - sinp * p.mz;
    p.mx = cost * px - sint * p.kx; p.my = cost * py - sint * p.ky; p.mz = cost * pz - sint * p.kz;
    p.kx = sint * px + cost * p.kx; p.ky = sint * py + cost * p.ky; p.kz = sint * pz + cost * p.kz;
}
// azimuth by rejection on 1 + S12/S11 (Q cos 2phi + U sin 2phi): :364-371.  Returns sin/cos of the accepted phi = 2 pi u;
// the double angle comes from the same pair (cos 2phi = (c-s)(c+s), sin 2phi = 2sc), one sincos per try.
__device__ inline void phi_rejection(Rng &g, const Photon &p, double r, double &sinp, double &cosp) {
    double amp = 1.0 + fabs(r) * sqrt(p.Q * p.Q + p.U * p.U);
    for (;;) {
        double u, u2; g.next2(u, u2);
        double Prand = amp * u2, s, c;
        d_sincos_2pi(u, &s, &c);
        double c2 = (c - s) * (c + s), s2 = 2.0 * s * c;
        if (Prand <= 1.0 + r * (p.Q * c2 + p.U * s2)) { sinp = s; cosp = c; return; }
    }
}
// dust absorption / reduced weight: :504-555 (= :217-264). true when the packet is destroyed
__device__ __forceinline__ bool dust_absorb(const KArgs &A, const Lds &l, Rng &g, Photon &p, const Cell &c) {
    p.nsd = p.nsd + p.wgt;
    const lart_params &par = A.par;
    double dep = 0.0; bool killed = false;
    if (!par.use_reduced_wgt) { if (g.next() > par.albedo) { dep = p.wgt; killed = true; } }
    else { dep = p.wgt * (1.0 - par.albedo); }
    if (par.line_type == 8 && dep != 0.0) aux_add(A.t.scal, l, 2, dep);      // W_abs1 (scattering_car.f90:524, :553)
    if (dep != 0.0 && par.save_Jabs) {
        double uu1 = c.vfx * p.kx + c.vfy * p.ky + c.vfz * p.kz;
        double xr = (p.xfreq + uu1) * (c.Dfreq / par.Dfreq_ref);
        int ix = (int)floor((xr - par.xfreq_min) / par.dxfreq) + 1;
        if (ix >= 1 && ix <= par.nxfreq) atomicAdd(&A.t.Jabs[ix - 1], dep);
        p.xfreq_ref = xr;
    }
    if (killed) {
        if (A.allph) {                                         // par%save_all_photons: the absorbed photon's lab-frame frequency, weight 0 (scattering_car.f90:236-246, 525-535)
            if (!par.save_Jabs) { const double uu1 = c.vfx * p.kx + c.vfy * p.ky + c.vfz * p.kz; p.xfreq_ref = (p.xfreq + uu1) * (c.Dfreq / par.Dfreq_ref); }
            p.wgt = 0.0;
        }
        p.inside = false; return true;
    }
    if (par.use_reduced_wgt) p.wgt = p.wgt * par.albedo;
    return false;
}

// one scattering event (scattering :14-120 Cartesian branch, then the selected routine), in three pieces so that the
// atom-velocity sampler between them can also run somewhere else (transport_kernel.hpp batches its |x| > 1 branch per block):
//   scatter_pre  -> dust-or-gas decision; a dust event completes here; a resonance event returns true with the sampler's
//                   arguments in ScatCtx (x in the local Doppler units of the cell / clump)
//   scatter_post -> everything after do_resonance's `uz = rand_resonance_vz(x, a)`
struct ScatCtx { double va, xloc, cl_scale, cl_Dfreq, xsamp, dnu, vsamp, rhokap, rhokapD; int species, iup, h2; double uni; };   // h2: 1-based H2 line of an H2 scattering (pseudo line type 9), 0 = an H I event;   // xsamp / vsamp: the sampler's frequency and Voigt parameter (2P1/2 level: xloc + dnu; deuterium, species 2: the D frame)
template <int VAR, bool STOKES, bool DUST, bool PEEL>
__device__ __forceinline__ bool scatter_pre(const KArgs &A, const Lds &l, Rng &g, Photon &p, unsigned &npeel, ScatCtx &sc) {
    const lart_params &par = A.par;
    const Cell &c = A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)];
    double va = c.voigt_a;
    // clump mode (scattering_car.f90:69-84, line_clump_mod.f90:30-59): opacities, Voigt parameter and Doppler width of the owner clump;
    // scale = Dfreq_ref / Dfreq_clump moves x into the clump's local Doppler units for the sampler and back
    double cl_scale = 1.0, cl_Dfreq = c.Dfreq, cl_rhokap = c.rhokap, cl_rhokapD = c.rhokapD;
    if (is_clump(VAR)) { const ClumpPhys ph = cl_phys(A, l, p.icl); va = ph.voigt_a; cl_Dfreq = ph.Dfreq; cl_rhokap = ph.rhokap; cl_rhokapD = ph.rhokapD; cl_scale = A.cl.Dfreq_ref / ph.Dfreq; }
    sc.h2 = 0;
    if (l.lt == 8 && p.iband == 2) {                          // scatter_dust_Ha_nostokes (scattering_car.f90:586-658; _amr scattering_amr.f90:147-209): every event of a band-2 packet is a dust event
        p.nsd = p.nsd + p.wgt;
        if (!par.use_reduced_wgt) {
            if (g.next() > par.albedo_Ha) {
                if (par.save_Jabs) { const int ix = (int)floor((p.xfreq - par.xfreq_min_Ha) / par.dxfreq_Ha) + 1; if (ix >= 1 && ix <= par.nxfreq_Ha) atomicAdd(&A.t.Jabs_Ha[ix - 1], p.wgt); }
                p.inside = false; aux_add(A.t.scal, l, 4, p.wgt);       // W_abs2
                if (A.allph) { p.xfreq_ref = p.xfreq; p.wgt = 0.0; }
                return false;
            }
        } else {
            if (par.save_Jabs) { const int ix = (int)floor((p.xfreq - par.xfreq_min_Ha) / par.dxfreq_Ha) + 1; if (ix >= 1 && ix <= par.nxfreq_Ha) atomicAdd(&A.t.Jabs_Ha[ix - 1], p.wgt * (1.0 - par.albedo_Ha)); }
            aux_add(A.t.scal, l, 4, p.wgt * (1.0 - par.albedo_Ha));
            p.wgt = p.wgt * par.albedo_Ha;
        }
        if (PEEL) { unsigned natom = 0; peel_dust_Ha<VAR>(A, l, DUST, p, npeel, natom); }
        const double cost = rand_hg(g, par.hgg_Ha), sint = sqrt(1.0 - cost * cost);
        double sinp, cosp; d_sincos_2pi(g.next(), &sinp, &cosp);
        rotate_k(p, cost, sint, cosp, sinp);
        return false;
    }
    if (l.lt == 9) {                                           // the H2 channel (scattering_car.f90:92-104, octree :44-57): three-way split H I / H2 / dust
        const double kap_HI = c.rhokap * calc_voigt(l, p.xfreq, va, c.Dfreq), kap_H2 = c.rhokap * h2_kappa_D(par, A.h2tab, l, p.xfreq, c.Dfreq), kap_D = DUST ? c.rhokapD : 0.0;
        if (g.next() * (kap_HI + kap_H2 + kap_D) <= kap_H2) {  // scatter_h2 :122-199
            const double ratio = par.h2_hi_width ? 1.0 : c.Dfreq / par.h2_Dfreq_Hz;       // h2_select_line_D (h2_mod.f90:81-115)
            double w[4], wtot = 0.0; int il = 1;
            for (int q = 0; q < par.h2_nlines; q++) { const double dx = A.h2tab[q] / c.Dfreq, x_h2 = (p.xfreq - dx) * ratio; w[q] = A.h2tab[4 + q] * ratio * voigt_H(l.vt, x_h2, A.h2tab[8 + q]); wtot = wtot + w[q]; }
            if (wtot > 0.0) { const double r = g.next() * wtot; double acc = 0.0; il = par.h2_nlines; for (int q = 0; q < par.h2_nlines; q++) { acc = acc + w[q]; if (r <= acc) { il = q + 1; break; } } }
            aux_add(A.t.scal, l, 2 + il - 1, p.wgt);                      // W_H2pump(il)
            if (g.next() > A.h2tab[12 + il - 1]) {            // fluorescence to another wavelength: the packet leaves Ly-alpha
                aux_add(A.t.scal, l, 0, p.wgt); p.inside = false; if (A.allph) p.wgt = 0.0;
                return false;
            }
            aux_add(A.t.scal, l, 1, p.wgt); p.nsg = p.nsg + p.wgt;
            const double rt = c.Dfreq / par.h2_Dfreq_Hz, dx = A.h2tab[il - 1] / c.Dfreq;       // (scatter_h2 always uses the H2 Doppler width: :163-166)
            sc.h2 = il; sc.cl_Dfreq = c.Dfreq; sc.xsamp = (p.xfreq - dx) * rt; sc.vsamp = A.h2tab[8 + il - 1]; sc.xloc = sc.xsamp;
            return true;
        }
    }
    if (DUST) {
        double rD = cl_rhokapD;
        double p_dust = rD / (cl_rhokap * calc_voigt(l, p.xfreq * cl_scale, va, cl_Dfreq) + rD);
        if (g.next() <= p_dust) {
            if (dust_absorb(A, l, g, p, c)) return false;
            if (STOKES && A.mu.nPDF > 0) {                     // scatter_dust_stokes :201-329
                if (PEEL) peel_emit<VAR, STOKES>(A, l, DUST, PEEL_DUST, p, p.xfreq, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, npeel);      // peeling_dust_stokes_outside, deferred
                double cost = rand_alias_linear(g, A.mu), sint = sqrt(1.0 - cost * cost);
                const DMueller &m = A.mu;
                double S11 = interp_eq(m.nPDF, m.coss, m.S11, cost), S12 = interp_eq(m.nPDF, m.coss, m.S12, cost);
                double S33 = interp_eq(m.nPDF, m.coss, m.S33, cost), S34 = interp_eq(m.nPDF, m.coss, m.S34, cost);
                double sinp, cosp; phi_rejection(g, p, S12 / S11, sinp, cosp);
                double cos2p = 2.0 * cosp * cosp - 1.0, sin2p = 2.0 * sinp * cosp;
                double Q0 = cos2p * p.Q + sin2p * p.U, U0 = -sin2p * p.Q + cos2p * p.U;
                double I1 = S11 + S12 * Q0, Q1 = S12 + S11 * Q0, U1 = S33 * U0 + S34 * p.V, V1 = -S34 * U0 + S33 * p.V;
                p.Q = Q1 / I1; p.U = U1 / I1; p.V = V1 / I1;
                rotate_triad(p, cost, sint, cosp, sinp);
            } else {                                           // scatter_dust_nostokes :488-584
                if (PEEL) peel_emit<VAR, STOKES>(A, l, DUST, PEEL_DUST, p, p.xfreq, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, npeel);      // peeling_dust_nostokes_outside, deferred
                double cost = rand_hg(g, par.hgg), sint = sqrt(1.0 - cost * cost);
                double sinp, cosp; d_sincos_2pi(g.next(), &sinp, &cosp);
                rotate_k(p, cost, sint, cosp, sinp);
            }
            return false;
        }
    }
    // resonance scattering: scatter_resonance_nostokes :660-827 / _stokes :331-486
    if (is_ovl(VAR)) p.xfreq = p.xfreq - ulos_clump(A, cl_phys(A, l, p.icl), p.kx, p.ky, p.kz);   // scatter_resonance_clump_*: global -> owner frame (:898-937)
    p.nsg = p.nsg + p.wgt;
    if (A.jpa.calc_P && c.rhokap > 0.0) {                      // add_to_Pa: scattering_car.f90:830-859
        int ib = (VAR == 3) ? amr_JPa_bin(A.amr, p.ic, p.x, p.y, p.z) : A.jpa.ind_bin[cidx(A.g, p.ic, p.jc, p.kc)];   // add_to_Pa_amr: scattering_amr.f90:917-942
        if (ib >= 1) atomicAdd(&(l.jp_P ? l.jp_P : A.jpa.P1)[ib - 1], p.wgt / (c.rhokap * c.Dfreq / A.jpa.cross0));
    }
    sc.va = va; sc.cl_scale = cl_scale; sc.cl_Dfreq = cl_Dfreq; sc.xloc = is_clump(VAR) ? p.xfreq * cl_scale : p.xfreq;
    sc.uni = is_clump(VAR) ? 0.0 : c.pad;
    sc.rhokap = cl_rhokap; sc.rhokapD = DUST ? cl_rhokapD : 0.0;   // (the peel prefilter of scatter_post reads them: no second gather of the cell / clump record)
    sc.xsamp = sc.xloc; sc.dnu = 0.0; sc.vsamp = va; sc.species = 1; sc.iup = 0;
    if (l.lt == 4 && l.ml_nup > 1) {                           // do_resonance5 / 6 (line_mod.f90:358-376, :421-449): which upward transition
        const double Dx2 = l.ml_dE[1] / cl_Dfreq, va2 = va * l.ml_damp[1] / l.ml_damp[0];
        const double p1 = voigt_H(l.vt, sc.xloc, va) * l.ml_f12[0], p2 = voigt_H(l.vt, sc.xloc + Dx2, va2) * l.ml_f12[1];
        const double xi = g.next();
        if (l.ml_nup == 2) {
            if (!(xi < p1 / (p2 + p1))) { sc.iup = 1; sc.xsamp = sc.xloc + Dx2; sc.vsamp = va2; }
        } else {
            const double Dx3 = l.ml_dE[2] / cl_Dfreq, va3 = va * l.ml_damp[2] / l.ml_damp[0];
            const double p3 = voigt_H(l.vt, sc.xloc + Dx3, va3) * l.ml_f12[2], ptot = p1 + p2 + p3;
            if (xi < p1 / ptot) {}
            else if (xi < (p1 + p2) / ptot) { sc.iup = 1; sc.xsamp = sc.xloc + Dx2; sc.vsamp = va2; }
            else { sc.iup = 2; sc.xsamp = sc.xloc + Dx3; sc.vsamp = va3; }
        }
    }
    if (l.lt == 7) {                                           // do_resonance_HD (line_mod.f90:490-510): a hydrogen or a deuterium atom
        const double dx = l.dnu_hz / cl_Dfreq, xD = (sc.xloc - dx) * l.hd_ratio;
        const double pH = voigt_H(l.vt, sc.xloc, va), pD = l.hd_nD * l.hd_ratio * voigt_H(l.vt, xD, va * l.hd_ratio_a);
        if (!(g.next() < pH / (pH + pD))) { sc.species = 2; sc.xsamp = xD; sc.vsamp = va * l.hd_ratio_a; }
    }
    if (l.lt == 2) {                                           // do_resonance2 (line_mod.f90:152-163): which level scatters
        sc.dnu = l.dnu_hz / cl_Dfreq;
        double pH = voigt_H(l.vt, sc.xloc + sc.dnu, va) * (1.0 / 3.0), pK = voigt_H(l.vt, sc.xloc, va) * (2.0 / 3.0);
        pH = pH / (pH + pK);
        if (g.next() < pH) sc.xsamp = sc.xloc + sc.dnu;
    }
    return true;
}
template <int VAR, bool STOKES, bool DUST, bool PEEL>
__device__ __forceinline__ void scatter_post(const KArgs &A, const Lds &l, Rng &g, Photon &p, unsigned &npeel, const ScatCtx &sc, double uz_loc_in) {
    double uz_loc = uz_loc_in;
    const lart_params &par = A.par;
    if (l.lt == 9 && sc.h2 > 0) {                              // scatter_h2 (scattering_car.f90:160-199): coherent in the molecule's frame, isotropic re-emission, H2 Doppler units and back
        const double ratio = sc.cl_Dfreq / par.h2_Dfreq_Hz, dx = A.h2tab[sc.h2 - 1] / sc.cl_Dfreq, xatom = sc.xsamp - uz_loc;
        double u0, u1, u2, u3; g.next2(u0, u1); g.next2(u2, u3);
        const double cost = 2.0 * u0 - 1.0, sint = sqrt(1.0 - cost * cost);
        double sinp, cosp, s2, c2; d_sincos_2pi(u1, &sinp, &cosp); d_sincos_2pi(u2, &s2, &c2);
        const double uxy = sqrt(-d_log(u3)), ux = uxy * c2, uy = uxy * s2;
        const double x_h2_new = xatom + uz_loc * cost + (ux * cosp + uy * sinp) * sint;
        p.xfreq = x_h2_new / ratio + dx;
        rotate_k(p, cost, sint, cosp, sinp);
        return;
    }
    // do_resonance1 (line_mod.f90:108-139; do_resonance1_clump: local Doppler units and back)
    double uz, xfreq_atom;
    const bool deut = (l.lt == 7) && sc.species == 2;          // a deuterium atom scattered (line%selected_species_HD = 2)
    if (deut) uz_loc = uz_loc / l.hd_ratio;                    // D Doppler units -> the photon's H units (line_mod.f90:507)
    if (is_clump(VAR)) { double scale_inv = 1.0 / sc.cl_scale; xfreq_atom = (sc.xloc - uz_loc) * scale_inv; uz = uz_loc * scale_inv; }
    else { uz = uz_loc; xfreq_atom = p.xfreq - uz; }
    double S11 = 0, S22 = 0, S12 = 0, S33 = 0, S44 = 0, sinp, cosp, cost, cost2, sint;
    double E1 = par.E1, E2 = par.E2, E3 = par.E3;              // photon%E1..E3 of this event (the peel routines read them)
    if (l.lt == 4) {                                           // do_resonance4 / 5 / 6[_clump]: the downward channel (alias draw when the level has > 1)
        int idown = 0; const int nd = l.ml_itab[sc.iup];
        if (nd > 1) {
            double u1, u2; g.next2(u1, u2);
            const int k = (int)floor(nd * u1);                 // rand_alias_choise (random_mt.f90:2122-2135), 0-based here
            idown = (u2 < l.ml_tab[48 + sc.iup * 4 + k]) ? k : l.ml_itab[3 + sc.iup * 4 + k] - 1;
            const double del_xfreq = l.ml_tab[sc.iup * 4 + idown] / sc.cl_Dfreq;
            if (is_clump(VAR)) { const double scale_inv = 1.0 / sc.cl_scale; xfreq_atom = (sc.xloc - uz_loc - del_xfreq) * scale_inv; }
            else xfreq_atom = xfreq_atom - del_xfreq;
        }
        E1 = l.ml_tab[12 + sc.iup * 4 + idown]; E2 = l.ml_tab[24 + sc.iup * 4 + idown]; E3 = l.ml_tab[36 + sc.iup * 4 + idown];
        if (l.ml_coh && l.ml_nup == 3) {                        // par%HeI_coherent: compute_HeI_E_coherent (line_mod.f90:8-34) of the atom-frame frequency
            const double D2 = is_clump(VAR) ? sc.xloc - uz_loc : xfreq_atom, D1 = D2 + l.ml_dE[1] / sc.cl_Dfreq, D0 = D2 + l.ml_dE[2] / sc.cl_Dfreq;
            const double D2D0 = D2 * D0, D2D1 = D2 * D1, D0D1 = D0 * D1, pp = D2 * D0 * D1;
            const double den = 4.0 * (D2D1 * D2D1 + 3.0 * D2D0 * D2D0 + 5.0 * D0D1 * D0D1);
            E1 = (3.0 * D2D0 * D2D0 + 7.0 * D0D1 * D0D1 + 8.0 * pp * D1 + 18.0 * pp * D0) / den;
            E3 = (3.0 * D2D0 * D2D0 + 15.0 * D0D1 * D0D1 + 8.0 * D2 * pp + 10.0 * pp * D0) / den;
            E2 = 1.0 - E1;
        }
    }
    if (l.lt == 2) {                                           // do_resonance2 :166-171: E1 of the atom-frame frequency
        const double qK = is_clump(VAR) ? sc.xloc - uz_loc : xfreq_atom, qH = qK + sc.dnu;
        E1 = (2.0 * qK * qH + qH * qH) / (qK * qK + 2.0 * qH * qH); E2 = 1.0 - E1; E3 = (E1 + 2.0) / 3.0;
    }
    bool converted = false;
    if (l.lt == 8) {                                           // do_resonance8 (line_mod.f90:318-329): 3p -> 2s with probability P_down(2): the packet turns into H-alpha
        if (g.next() < par.P_conv) { converted = true; p.iband = 2; E1 = par.E_conv[0]; E2 = par.E_conv[1]; E3 = par.E_conv[2]; }
        if (converted) {                                       // scattering_car.f90:697-705: W_conv and the conversion-rate profile (add_to_Pconv :861-894)
            aux_add(A.t.scal, l, 0, p.wgt);
            const Cell &cc = A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)];
            if (A.jpa.Pconv && cc.rhokap > 0.0) {
                const int ib = (VAR == 3) ? amr_JPa_bin(A.amr, p.ic, p.x, p.y, p.z) : A.jpa.ind_bin[cidx(A.g, p.ic, p.jc, p.kc)];
                if (ib >= 1) atomicAdd(&A.jpa.Pconv[ib - 1], p.wgt / (cc.rhokap * cc.Dfreq / A.jpa.cross0));
            }
        }
    }
    if (STOKES) {
        cost = (l.lt == 2 || l.lt == 4) ? rand_resonance_u(g.next(), E1) : dipole_cost(l, g.next(), E1); cost2 = cost * cost; sint = sqrt(1.0 - cost2);
        S22 = 0.75 * E1 * (cost2 + 1.0); S11 = S22 + E2; S12 = 0.75 * E1 * (cost2 - 1.0); S33 = 1.5 * E1 * cost;
        S44 = (l.lt == 2) ? (E1 + 2.0) / 2.0 * cost : 1.5 * E3 * cost;
        LART_PROF_MARK(p, 12);
        phi_rejection(g, p, S12 / S11, sinp, cosp);
        LART_PROF_MARK(p, 13);
    } else {
        double ud, up; g.next2(ud, up);                       // polar angle, then azimuth: the reference's order of draws
        cost = (l.lt == 2 || l.lt == 4) ? rand_resonance_u(ud, E1) : dipole_cost(l, ud, E1); cost2 = cost * cost; sint = sqrt(1.0 - cost2);
        d_sincos_2pi(up, &sinp, &cosp);
    }
    // perpendicular atom velocity; truncated tail when core-skipping (:726-752, :400-430)
    double ux, uy, xc = 0.0, xc2 = 0.0;
    if (par.core_skip) { if (VAR == 3) xcrit_local_amr(A, p, xc, xc2); else xcrit_local(par, A.g, l, p, xc, xc2); }
    if (par.core_skip && fabs(p.xfreq) < xc) {
        double u2, u3; g.next2(u2, u3); double uxy = sqrt(xc2 - d_log(u3)), s2, c2; d_sincos_2pi(u2, &s2, &c2);
        ux = uxy * c2; uy = uxy * s2;
    } else if (STOKES) {
        double g2, g1 = g.gauss_pair(g2);
        ux = g1 * 0.70710678118654752440084436210485; uy = g2 * 0.70710678118654752440084436210485;
    } else {
        double u2, u3; g.next2(u2, u3); double uxy = sqrt(-d_log(u3)), s2, c2; d_sincos_2pi(u2, &s2, &c2);
        ux = uxy * c2; uy = uxy * s2;
    }
    LART_PROF_MARK(p, 14);
    if (deut) { ux = ux / l.hd_ratio; uy = uy / l.hd_ratio; }  // scattering_car.f90:402-409, 417-424
    if (is_clump(VAR)) { double vth_ratio = sc.cl_Dfreq / A.cl.Dfreq_ref; ux = ux * vth_ratio; uy = uy * vth_ratio; }   // scattering_car.f90:384-388, 715-719
    p.xfreq = xfreq_atom + uz * cost + (ux * cosp + uy * sinp) * sint;
    const double g_rec0 = deut ? par.g_recoil0_D : par.g_recoil0;      // :427-437
    if (par.recoil && !converted) p.xfreq = p.xfreq - (g_rec0 / sc.cl_Dfreq) * (1.0 - cost);      // (no recoil on the conversion channel: :768-770)
    if (STOKES) {
        LART_PROF_MARK(p, 6);
        if (PEEL) { const PeelCell pc = {sc.rhokap, sc.rhokapD, sc.va, sc.cl_Dfreq, sc.uni}; peel_emit<VAR, STOKES>(A, l, DUST, PEEL_RESONANCE, p, xfreq_atom, ux, uy, uz, E1, E2, E3, g_rec0, npeel, &pc); }   // peeling_resonance_stokes_outside, deferred
        LART_PROF_MARK(p, 15);
        double cos2p = 2.0 * cosp * cosp - 1.0, sin2p = 2.0 * sinp * cosp;
        double Q0 = cos2p * p.Q + sin2p * p.U, U0 = -sin2p * p.Q + cos2p * p.U;
        double I1 = S11 + S12 * Q0, Q1 = S12 + S22 * Q0, U1 = S33 * U0, V1 = S44 * p.V;
        p.Q = Q1 / I1; p.U = U1 / I1; p.V = V1 / I1;
        rotate_triad(p, cost, sint, cosp, sinp);
    } else {
        if (PEEL && converted) { unsigned natom = 0; peel_conversion_Ha<VAR>(A, l, DUST, p, A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)], ux, uy, uz, E1, E2, npeel, natom); }      // peeling_conversion_Ha (:784-788)
        else if (PEEL) { const PeelCell pc = {sc.rhokap, sc.rhokapD, sc.va, sc.cl_Dfreq, sc.uni}; peel_emit<VAR, STOKES>(A, l, DUST, PEEL_RESONANCE, p, xfreq_atom, ux, uy, uz, E1, E2, E3, g_rec0, npeel, &pc); }   // peeling_resonance_nostokes_outside, deferred
        rotate_k(p, cost, sint, cosp, sinp);
        if (converted) {                                       // :811-827: emitted at line centre in the atom frame -> the lab frame along the new direction, reference Doppler units
            const Cell &cc = A.g.cell[cidx(A.g, p.ic, p.jc, p.kc)];
            const double u1_fluid = cc.vfx * p.kx + cc.vfy * p.ky + cc.vfz * p.kz;
            p.xfreq = (p.xfreq - xfreq_atom + u1_fluid) * (cc.Dfreq / par.Dfreq_ref);
        }
    }
    if (is_ovl(VAR)) p.xfreq = p.xfreq + ulos_clump(A, cl_phys(A, l, p.icl), p.kx, p.ky, p.kz);   // ... and back to the global frame along the new direction
}
// par%save_all_photons (run_simulation_mod.f90:249-358): the record of photon k.  The impact parameter |r - (r.k)k| does not change along
// the ray, so the reference's move to the sphere's surface (:302-324) drops out.
__device__ inline double impact_parameter(const Photon &p, double &mx, double &my, double &mz) {
    const double rk = p.x * p.kx + p.y * p.ky + p.z * p.kz;
    mx = p.x - rk * p.kx; my = p.y - rk * p.ky; mz = p.z - rk * p.kz;
    return sqrt(mx * mx + my * my + mz * mz);
}
// one record of the history probe: the state a packet is in after its birth (kind 0), after a scattering (1) or when it has left / was absorbed (2).
// [0] draws consumed, [1] xfreq, [2..4] position, [5] cell code (ic + 2^21 jc + 2^42 kc; the owner clump in clump media), [6] kz, [7] kind
template <int VAR>
__device__ inline void trace_rec(const KArgs &A, long long k, unsigned long long rng_n, const Photon &p, int kind) {
    const int sl = A.trace_slot[k];
    if (sl < 0) return;
    const int i = A.trace_n[sl]; A.trace_n[sl] = i + 1;            // one lane owns a packet: no atomic
    if (i >= A.trace_cap) return;
    double *r = A.trace + ((size_t)sl * A.trace_cap + i) * LART_TRACE_W;
    r[0] = (double)rng_n; r[1] = p.inside ? p.xfreq : p.xfreq_ref; r[2] = p.x; r[3] = p.y; r[4] = p.z;
    r[5] = is_clump(VAR) ? (double)p.icl : (double)p.ic + 2097152.0 * (double)p.jc + 4398046511104.0 * (double)p.kc;
    r[6] = p.kz; r[7] = (double)kind;
}
__device__ inline void allph_initial(const KArgs &A, const Photon &p, long long k) {          // make_all_initial_photons :249-292
    A.allph[1 * A.allph_n + k] = p.xfreq;
    if (A.par.source_geometry != 0) { double mx, my, mz; A.allph[9 * A.allph_n + k] = impact_parameter(p, mx, my, mz); }
}
template <bool STOKES>
__device__ inline void allph_final(const KArgs &A, const Photon &p, long long k) {            // make_all_photons :294-358
    const long long n = A.allph_n; double mx, my, mz; const double mm = impact_parameter(p, mx, my, mz);
    A.allph[k] = mm; A.allph[2 * n + k] = p.xfreq_ref; A.allph[3 * n + k] = p.nsg; A.allph[4 * n + k] = p.nsd;
    if (STOKES) {
        double cos2p = 1.0, sin2p = 0.0;
        if (mm > 0.0) { mx = mx / mm; my = my / mm; mz = mz / mm;
            const double cosp = mx * p.mx + my * p.my + mz * p.mz, sinp = mx * p.nx + my * p.ny + mz * p.nz;
            cos2p = 2.0 * cosp * cosp - 1.0; sin2p = 2.0 * sinp * cosp; }
        A.allph[5 * n + k] = p.wgt; A.allph[6 * n + k] = (cos2p * p.Q + sin2p * p.U) * p.wgt;
        A.allph[7 * n + k] = (-sin2p * p.Q + cos2p * p.U) * p.wgt; A.allph[8 * n + k] = p.V * p.wgt;
    }
}
// `on`: this lane has a packet to scatter.  Every lane of the wave that is inside the event loop calls this (the |x| <= 1
// sampler is a wave-cooperative stage, vz_core_coop).
template <int VAR, bool STOKES, bool DUST, bool PEEL>
__device__ __forceinline__ void scatter_event(const KArgs &A, const Lds &l, Rng &g, Photon &p, unsigned &npeel, bool on) {
    ScatCtx sc; sc.xloc = 0.0; sc.xsamp = 0.0; sc.va = 1.0; sc.vsamp = 1.0; sc.species = 1; sc.iup = 0; sc.h2 = 0;
#if defined(LART_NO_COOP)
    if (!on || !scatter_pre<VAR, STOKES, DUST, PEEL>(A, l, g, p, npeel, sc)) return;
    double uz_loc = rand_resonance_vz(g, sc.xsamp, sc.vsamp LART_PROF_PASS(p));
    scatter_post<VAR, STOKES, DUST, PEEL>(A, l, g, p, npeel, sc, uz_loc);
#else
    bool go = false;
    if (on) go = scatter_pre<VAR, STOKES, DUST, PEEL>(A, l, g, p, npeel, sc);
    LART_PROF_MARK(p, 4);
    const bool core = go && fabs(sc.xsamp) <= 1.0;
    double uz_loc;
    if (l.coop) {
        uz_loc = vz_core_coop(g, core, sc.xsamp, sc.vsamp, l.coop);
        LART_PROF_MARK(p, 5);
#ifdef LART_NO_WING_COOP
        if (go && !core) uz_loc = vz_wing(g, sc.xsamp, sc.vsamp LART_PROF_PASS(p));
#else
        const double uw = vz_wing_coop(g, go && !core, sc.xsamp, sc.vsamp, l.coop);
        if (go && !core) uz_loc = uw;
        LART_PROF_MARK(p, 9);
#endif
    } else { uz_loc = core ? vz_core(g, sc.xsamp, sc.vsamp LART_PROF_PASS(p)) : 0.0; if (go && !core) uz_loc = vz_wing(g, sc.xsamp, sc.vsamp LART_PROF_PASS(p)); }
    g.template fill<LART_FILL_MID(STOKES)>(go);                // the draws of scatter_post: polar angle, azimuth (tries), perpendicular velocity (tries)
    LART_PROF_MARK(p, 11);
    if (go) scatter_post<VAR, STOKES, DUST, PEEL>(A, l, g, p, npeel, sc, uz_loc);
#endif
}

// ---------------------------------------------------------------------------------------------
// photon birth: generate_photon.f90:3-339, setup_isotropic_injection :342-408
// ---------------------------------------------------------------------------------------------
template <int VAR, bool STOKES, bool DUST, bool PEEL>
__device__ __forceinline__ void generate_photon(const KArgs &A, const Lds &l, Rng &g, Photon &p, unsigned &npeel) {
    const lart_params &par = A.par; const DGrid &gr = A.g;
    if (par.source_geometry == 1) {
        double rp = cbrt(g.next()) * par.source_rmax, cost = 2.0 * g.next() - 1.0, sint = sqrt(1.0 - cost * cost), sp, cp;
        d_sincos_2pi(g.next(), &sp, &cp);
        p.x = rp * sint * cp; p.y = rp * sint * sp; p.z = rp * cost;
    } else if (par.source_geometry == 2) {                     // uniform_cylinder :43-49
        double rp = sqrt(g.next()) * par.source_rmax, sp, cp; d_sincos_2pi(g.next(), &sp, &cp);
        p.x = rp * cp; p.y = rp * sp; p.z = par.zrange * g.next() + gr.zmin;
    } else if (par.source_geometry == 3) {                     // uniform over the box :50-54
        p.x = par.xrange * g.next() + gr.xmin; p.y = par.yrange * g.next() + gr.ymin; p.z = par.zrange * g.next() + gr.zmin;
    } else if (par.source_geometry == 4) {                     // uniform_xy: a sheet at z = 0 :55-67
        if (par.source_rmax > 0.0) { double rp = sqrt(g.next()) * par.source_rmax, sp, cp; d_sincos_2pi(g.next(), &sp, &cp); p.x = rp * cp; p.y = rp * sp; }
        else { p.x = par.xrange * g.next() + gr.xmin; p.y = par.yrange * g.next() + gr.ymin; }
        p.z = 0.0;
    } else if (par.source_geometry == 5) {                     // gaussian in z :68-72 (one deviate of the polar pair)
        p.x = par.xrange * g.next() + gr.xmin; p.y = par.yrange * g.next() + gr.ymin;
        double second; p.z = par.source_zscale / 1.4142135623730951 * g.gauss_pair(second);
    } else { p.x = par.xs_point; p.y = par.ys_point; p.z = par.zs_point; }
    p.wgt = 1.0;
    if (VAR == 1) { if (p.x < gr.xmin) p.x = -p.x; if (p.y < gr.ymin) p.y = -p.y; if (p.z < gr.zmin) p.z = -p.z; }
    double cost = 2.0 * g.next() - 1.0, sint = sqrt(1.0 - cost * cost), sinp, cosp;
    d_sincos_2pi(g.next(), &sinp, &cosp);
    p.kx = sint * cosp; p.ky = sint * sinp; p.kz = cost;
    if (VAR == 3) { p.ic = amr_find_leaf(A.amr, p.x, p.y, p.z); p.jc = 1; p.kc = 1; if (p.ic <= 0) p.ic = 1; }   // generate_photon.f90:375-379 (source inside the box)
    else if (is_clump(VAR)) { p.ic = 1; p.jc = 1; p.kc = 1; }                       // the background grid is uniform (grid_mod_clump.f90:101-108): one record
    else {
    p.ic = (int)floor((p.x - gr.xmin) / gr.dx) + 1; p.jc = (int)floor((p.y - gr.ymin) / gr.dy) + 1; p.kc = (int)floor((p.z - gr.zmin) / gr.dz) + 1;
    if (p.kx < 0.0 && p.ic == gr.nx + 1) p.ic = gr.nx;
    if (p.ky < 0.0 && p.jc == gr.ny + 1) p.jc = gr.ny;
    if (p.kz < 0.0 && p.kc == gr.nz + 1) p.kc = gr.nz;
    if (p.kx > 0.0 && p.ic < 1) p.ic = 1;
    if (p.ky > 0.0 && p.jc < 1) p.jc = 1;
    if (p.kz > 0.0 && p.kc < 1) p.kc = 1;
    }
    if (STOKES) { p.mx = cost * cosp; p.my = cost * sinp; p.mz = -sint; p.nx = -sinp; p.ny = cosp; p.nz = 0.0; p.Q = 0.0; p.U = 0.0; p.V = 0.0; }
    p.nsg = 0.0; p.nsd = 0.0; p.inside = true; p.xfreq = par.xfreq0; p.xfreq_ref = 0.0;
    // a source outside the medium box never reaches a table in the reference's configs; clamp the gather
    const Cell &c = (VAR == 3 || is_clump(VAR)) ? gr.cell[p.ic - 1] : gr.cell[cidx(gr, clampi(p.ic, 1, gr.nx), clampi(p.jc, 1, gr.ny), clampi(p.kc, 1, gr.nz))];
    if (l.lt == 2 && par.spectral_type != 2) { const double dnu = l.dnu_hz / c.Dfreq; if (g.next() <= 1.0 / 3.0) p.xfreq = p.xfreq - dnu; }   // the source feeds both levels (generate_photon.f90:160-167; not for a continuum)
    if (l.lt == 4 && par.spectral_type != 2) {                 // :168-243: the source feeds the levels by f12 and every downward channel by its branching ratio
        int iup = 0;
        if (l.ml_nup > 1) {
            const double p1 = l.ml_f12[0], p2 = l.ml_f12[1], xi = g.next();
            if (l.ml_nup == 2) { if (!(xi < p1 / (p2 + p1))) iup = 1; }
            else { const double ptot = p1 + p2 + l.ml_f12[2]; if (xi < p1 / ptot) {} else if (xi < (p1 + p2) / ptot) iup = 1; else iup = 2; }
            if (iup > 0) p.xfreq = p.xfreq - l.ml_dE[iup] / c.Dfreq;
        }
        const int nd = l.ml_itab[iup];
        if (nd > 1) {
            double u1, u2; g.next2(u1, u2);
            const int k = (int)floor(nd * u1);
            const int idown = (u2 < l.ml_tab[48 + iup * 4 + k]) ? k : l.ml_itab[3 + iup * 4 + k] - 1;
            p.xfreq = p.xfreq - l.ml_tab[iup * 4 + idown] / c.Dfreq;
        }
    }
    if (par.spectral_type == 1) p.xfreq = p.xfreq + rand_voigt(g, c.voigt_a);
    else if (par.spectral_type == 2) {                         // continuum: flat over the frequency axis, in units of the local cell (:244-247)
        p.xfreq = g.next() * (par.xfreq_max - par.xfreq_min) + par.xfreq_min;
        p.xfreq = p.xfreq / (c.Dfreq / par.Dfreq_ref);
    } else if (par.spectral_type == 3) {                       // gaussian: sigma in reference Doppler units (:254-266); one deviate of the polar pair
        double second; p.xfreq = p.xfreq + g.gauss_pair(second) * par.source_sigma_x;
        p.xfreq = p.xfreq / (c.Dfreq / par.Dfreq_ref);
    }
    double u1 = c.vfx * p.kx + c.vfy * p.ky + c.vfz * p.kz;
    if (!par.comoving_source) p.xfreq = p.xfreq - u1;
    if (par.save_Jin) {
        double xlab = (p.xfreq + u1) * (c.Dfreq / par.Dfreq_ref);
        int ix = (int)floor((xlab - par.xfreq_min) / par.dxfreq) + 1;
        if (ix >= 1 && ix <= par.nxfreq) atomicAdd(&A.t.Jin[ix - 1], p.wgt);
    }
    p.icl = 0;
    if (VAR == 4) {                                           // birth clump: generate_photon.f90:325-332
        p.icl = clump_at_point(A.cl, p.x, p.y, p.z);
        if (p.icl > 0) p.xfreq = p.xfreq - ulos_clump(A, A.cl.phys[p.icl - 1], p.kx, p.ky, p.kz);
    }
    if (PEEL) peel_emit<VAR, STOKES>(A, l, DUST, PEEL_DIRECT, p, p.xfreq, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, npeel);            // peeling_direct_outside, deferred
}
