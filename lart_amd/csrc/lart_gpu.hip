// lart_gpu.hip -- C-ABI (include/lart_gpu.h) over the HIP transport kernels + device-function probes.
// Host side is plain HIP runtime calls on one stream per context; no torch types, no CPU fallback.
#include "transport_kernel.hpp"
#include "sightline.hpp"
#include <rccl/rccl.h>
#include <unistd.h>
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <ctime>

extern "C" void lart_launch_transport_var0(int, int, int, dim3, dim3, size_t, hipStream_t, const KArgs &);
extern "C" void lart_launch_transport_var1(int, int, int, dim3, dim3, size_t, hipStream_t, const KArgs &);
extern "C" void lart_launch_transport_var2(int, int, int, dim3, dim3, size_t, hipStream_t, const KArgs &);
extern "C" void lart_launch_transport_var3(int, int, int, dim3, dim3, size_t, hipStream_t, const KArgs &);
extern "C" void lart_launch_transport_var4(int, int, int, dim3, dim3, size_t, hipStream_t, const KArgs &);
extern "C" void lart_launch_transport_var5(int, int, int, dim3, dim3, size_t, hipStream_t, const KArgs &);
extern "C" void lart_launch_transport_var6(int, int, int, dim3, dim3, size_t, hipStream_t, const KArgs &);
extern "C" void lart_launch_transport_var7(int, int, int, dim3, dim3, size_t, hipStream_t, const KArgs &);
extern "C" void lart_launch_transport_var8(int, int, int, dim3, dim3, size_t, hipStream_t, const KArgs &);
extern "C" void lart_launch_transport_var9(int, int, int, dim3, dim3, size_t, hipStream_t, const KArgs &);
extern "C" int lart_launch_transport_lt89(int var, int dust, int peel, dim3, dim3, size_t, hipStream_t, const KArgs &);
extern "C" void lart_launch_peel(int var, int stokes, dim3, dim3, dim3, size_t, hipStream_t, const KArgs &);

static thread_local std::string g_err;
// Every fill and copy goes through the context's OWN stream.  The streams are created non-blocking, so the null stream is not ordered with them: a hipMemset on it is a
// fill KERNEL that may run long after the host call has returned -- with other contexts keeping the GPU busy, after the first transport kernel of the run, zeroing
// tallies in mid-run (round 4: three "equals the other host" tests failed next to the full-size runs of the device suite; tools/dbg/bg_race.py).  A fill is stream-ordered
// in front of the kernels that use the memory; a copy is followed by a synchronisation of that stream (the host buffers are pageable and often temporaries).
static hipError_t lart_put(hipStream_t st, void *dst, const void *src, size_t bytes) {
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st); return e != hipSuccess ? e : hipStreamSynchronize(st); }
static hipError_t lart_get(hipStream_t st, void *dst, const void *src, size_t bytes) {
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st); return e != hipSuccess ? e : hipStreamSynchronize(st); }
static int fail(const std::string &m) { g_err = m; return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct lart_gpu {
    int device = 0;
    hipStream_t stream = nullptr;
    lart_params par{};
    // medium
    DGrid g{}; Cell *d_cell = nullptr; double *d_faces = nullptr; size_t ncell = 0; bool has_dust_table = false;
    // Mueller
    DMueller mu{}; double *d_mu = nullptr; int *d_alias = nullptr;
    // resonance + fluorescence multiplet (line types 4, 5, 6)
    DMulti ml{}; double *d_ml_tab = nullptr; int *d_ml_itab = nullptr; bool has_ml = false;
    // observers
    DObs *d_obs = nullptr; int nobs = 0, nxim = 0, nyim = 0; double dxim = 0, dyim = 0; int heal_nside = 0;
    std::vector<DObs> h_obs; std::vector<double> h_faces;    // host copies for the sight-line ray set-up
    // J/P profile tallies
    DJPa jpa{}; int *d_ind_bin = nullptr;
    // octree (VAR 3)
    bool is_amr = false; DAmr amr{}; int *d_amr_int = nullptr; AmrGeom *d_amr_geom = nullptr;
    // clump medium (VAR 4)
    bool is_clump = false, clump_overlap = false; DClumps cl{}; ClumpGeo *d_cl_geo = nullptr; ClumpPhys *d_cl_phys = nullptr; int *d_cl_int = nullptr; unsigned long long *d_cl_mask = nullptr;
    // fused tally buffer
    double *d_tally = nullptr; int64_t ntally = 0; DTally t{};
    DRep rep{}; bool rep_dirty = false;                      // per-XCD copies of the contended tally sections
    // pool
    DPool pool{}; void *d_pool = nullptr; int64_t want_lanes = 0; int block = 256; bool block_user = false;
    DCtl *d_ctl = nullptr;
    // run
    long long first_id = 0, stride = 1, count = 0; unsigned long long seed = 0; bool begun = false;
    int log_on = 0; double *d_log = nullptr; int64_t nlog = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; double kernel_ms = 0.0; int64_t nlaunch = 0;
    double sight_ms = 0.0;                                   // device time of the last sight-line map
    // lane compaction: the launch covers lanes [0, live_lanes); state of the last lart_gpu_sync
    int64_t live_lanes = 0; bool state_valid = false; int64_t last_alive = 0, last_next_k = 0; int auto_compact = 1;
    int *d_compact = nullptr; int64_t compact_lanes = 0; int64_t n_compactions = 0; int64_t compact_alive = 0, compact_waves = 0; int spread_drain = 1;
         // d_compact holds the slot lists of a pool of compact_lanes lanes
    // peel-off event queue (deferred next-event estimator): records of one transport launch, consumed by peel_kernel right after it
    // two buffers: while peel_kernel works through the records of slice k on stream2, the transport kernel of slice k + 1 fills the other one
    DPeelQ pq[2]{}; void *d_pq = nullptr; size_t pq_bytes_max = 0; int peel_slice = 0; int n_cu = 0; size_t pq_nobs = 0, pq_lanes = 0;   // (the queue was laid out for pq_nobs observers and pq_lanes lanes)
    hipStream_t stream2 = nullptr; hipEvent_t ev_filled[2] = {nullptr, nullptr}, ev_drained[2] = {nullptr, nullptr}; bool drained_pending[2] = {false, false};
    // output_reduce across GPUs: RCCL communicator of this context (one rank per context)
    ncclComm_t comm = nullptr; int comm_n = 0, comm_rank = -1;
    // the root's receive buffer of output_reduce (MPI_REDUCE has separate send / receive buffers, memory_mod_mpi.f90:377-393: reducing twice gives
    // the same totals).  Valid until this context adds events or zeroes its tallies; lart_gpu_get_tallies hands out the totals while it is.
    double *d_reduced = nullptr; bool reduced_valid = false;
    // history probe (lart_gpu_test_trace)
    int *d_trace_slot = nullptr, *d_trace_n = nullptr; double *d_trace = nullptr; int trace_cap = 0;
    double *d_h2tab = nullptr;                               // the H2 line table of h2_kappa_D (16 doubles)
};

extern "C" const char *lart_gpu_last_error(void) { return g_err.c_str(); }

extern "C" int lart_gpu_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

extern "C" int lart_gpu_create(lart_gpu **out, const lart_params *par, int device) {
    if (!out || !par) return fail("lart_gpu_create: null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail("lart_gpu_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= n) return fail("lart_gpu_create: device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    if (par->source_geometry < 0 || par->source_geometry > 5) return fail("lart_gpu_create: source_geometry must be 0..5 (point, uniform_sphere, uniform_cylinder, uniform, uniform_xy, gaussian)");
    if (par->spectral_type < 0 || par->spectral_type > 3) return fail("lart_gpu_create: spectral_type must be 0 monochromatic, 1 voigt, 2 continuum or 3 gaussian");
    if (par->line_type != 1 && par->line_type != 2 && par->line_type != 7 && par->line_type != 8 && !(par->line_type >= 4 && par->line_type <= 6))
        return fail("lart_gpu_create: line_type must be 1 (single line), 2 (S-P doublet, par%fine_structure), 4 / 5 / 6 (resonance + fluorescence multiplets, with lart_gpu_set_line_multiplet), 7 (H + D, ly_alpha_HD) or 8 (ly_beta)");
    if (par->line_type == 8) {                                  // the reference's vetoes for ly_beta (setup.f90:239-280) and what the band-2 path needs
        if (par->use_stokes) return fail("lart_gpu_create: ly_beta: Stokes polarization not yet supported (setup.f90:254-258)");
        if (par->xyz_symmetry || par->xy_symmetry || par->xy_periodic || par->Omega != 0.0) return fail("lart_gpu_create: ly_beta: xyz/xy symmetry, xy_periodic and shear not yet supported (setup.f90:270-275)");
        if (par->core_skip || par->core_skip_global) return fail("lart_gpu_create: ly_beta forces core_skip = .false. (setup.f90:281-290): the host has to clear it");
        if (!(par->P_conv >= 0.0 && par->P_conv <= 1.0) || par->nxfreq_Ha < 1 || !(par->dxfreq_Ha > 0.0) || !(par->R_Ha >= 0.0) || !(par->albedo_Ha >= 0.0 && par->albedo_Ha <= 1.0) || !(fabs(par->hgg_Ha) < 1.0))
            return fail("lart_gpu_create: line_type 8 needs P_conv in [0,1], the H-alpha axis (nxfreq_Ha, xfreq_min_Ha, dxfreq_Ha) and the band-2 dust constants (R_Ha, albedo_Ha, hgg_Ha)");
    }
    if (par->h2_nlines != 0) {                                  // H2 lines next to Ly-alpha (setup.f90:302-348)
        if (par->h2_nlines < 0 || par->h2_nlines > 4) return fail("lart_gpu_create: h2_nlines must be 0..4");
        if (par->line_type != 1) return fail("lart_gpu_create: H2: only supported for line_id = 'ly_alpha' (line_type 1)");
        if (par->use_stokes) return fail("lart_gpu_create: H2: Stokes polarization not yet supported");
        if (par->xyz_symmetry || par->xy_symmetry || par->Omega != 0.0) return fail("lart_gpu_create: H2: xyz/xy symmetry and shear not yet supported");
        if (!(par->h2_Dfreq_Hz > 0.0)) return fail("lart_gpu_create: H2: h2_Dfreq_Hz must be > 0");
        for (int i = 0; i < par->h2_nlines; i++) if (!(par->h2_strength[i] >= 0.0) || !(par->h2_a_damp[i] > 0.0) || !(par->h2_p_scat[i] >= 0.0 && par->h2_p_scat[i] <= 1.0)) return fail("lart_gpu_create: H2: bad line record (strength >= 0, a_damp > 0, p_scat in [0,1])");
    }
    if (par->line_type == 7 && !(par->ratio_Dfreq_HD > 0.0 && par->ratio_voigta_HD > 0.0 && par->delta_nu_HD_Hz > 0.0 && par->nD_over_nH >= 0.0)) return fail("lart_gpu_create: line_type 7 needs the cross-species constants of line% (delta_nu_HD_Hz, ratio_Dfreq_HD, ratio_voigta_HD, nD_over_nH)");
    if (par->line_type == 2 && !(par->DnuHK_Hz > 0.0)) return fail("lart_gpu_create: line_type 2 needs DnuHK_Hz > 0 (line%DnuHK_Hz)");
    lart_gpu *c = new lart_gpu();
    c->device = device; c->par = *par;
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&c->d_ctl, sizeof(DCtl)));
    HIPCHK(hipMemsetAsync(c->d_ctl, 0, sizeof(DCtl), c->stream));
    if (par->h2_nlines > 0) {
        double h[16] = {0}; for (int i = 0; i < par->h2_nlines; i++) { h[i] = par->h2_dnu_Hz[i]; h[4 + i] = par->h2_strength[i]; h[8 + i] = par->h2_a_damp[i]; h[12 + i] = par->h2_p_scat[i]; }
        HIPCHK(hipMalloc(&c->d_h2tab, sizeof h)); HIPCHK(lart_put(c->stream, c->d_h2tab, h, sizeof h));
    }
    *out = c;
    return 0;
}

static void free_tally(lart_gpu *c) {
    if (c->d_tally) { (void)hipFree(c->d_tally); c->d_tally = nullptr; }
    if (c->d_reduced) { (void)hipFree(c->d_reduced); c->d_reduced = nullptr; } c->reduced_valid = false;
    if (c->rep.base) { (void)hipFree(c->rep.base); } c->rep = DRep{}; c->rep_dirty = false;
}

// a setter replaces device tables a resident photon pool refers to: wait for the stream, drop the tally layout and end the current
// run, so that lart_gpu_step cannot launch with stale pointers (the caller has to lart_gpu_begin again)
static void invalidate_run(lart_gpu *c) {
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_tally(c);
    c->begun = false; c->state_valid = false;
}

extern "C" int lart_gpu_destroy(lart_gpu *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &e : c->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    (void)hipFree(c->d_cell); (void)hipFree(c->d_faces); (void)hipFree(c->d_mu); (void)hipFree(c->d_alias); (void)hipFree(c->d_ml_tab); (void)hipFree(c->d_ml_itab); (void)hipFree(c->d_obs); (void)hipFree(c->d_ind_bin); (void)hipFree(c->d_amr_int); (void)hipFree(c->d_amr_geom); (void)hipFree(c->d_cl_geo); (void)hipFree(c->d_cl_phys); (void)hipFree(c->d_cl_int); (void)hipFree(c->d_cl_mask);
    free_tally(c); (void)hipFree(c->d_h2tab); (void)hipFree(c->d_pool); (void)hipFree(c->d_ctl); (void)hipFree(c->d_log); (void)hipFree(c->d_compact); (void)hipFree(c->d_pq);
    if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
    for (int b = 0; b < 2; b++) { if (c->ev_filled[b]) (void)hipEventDestroy(c->ev_filled[b]); if (c->ev_drained[b]) (void)hipEventDestroy(c->ev_drained[b]); }
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

// lay the fused tally buffer out: scalars | Jout | Jin | Jabs | Jmu | per-observer cubes and images
static int build_tally(lart_gpu *c) {
    (void)hipSetDevice(c->device);
    free_tally(c);
    const lart_params &p = c->par;
    int64_t nxf = p.nxfreq, nmu = p.save_Jmu ? p.nmu : 0;
    int64_t n2 = (int64_t)c->nobs * c->nxim * c->nyim, n3 = n2 * nxf;
    bool peel = p.save_peeloff && c->nobs > 0;
    int64_t off = 32;                                       // scalars 0..6; 8..31 are section clocks / trip counts of -DLART_PROFILE builds
    int64_t oJout = off; off += nxf; int64_t oJin = off; off += nxf; int64_t oJabs = off; off += nxf; int64_t oJmu = off; off += nxf * nmu;
    const bool lyb = p.line_type == 8; const int64_t nxh = lyb ? p.nxfreq_Ha : 0;
    int64_t oJoutHa = off; off += nxh; int64_t oJabsHa = off; off += nxh;
    const int64_t off_rep = off;                            // everything from here on is contended: gets per-XCD copies
    int64_t o3[8], o2[8];
    auto wanted = [&](int k) { return k < 2 || (k < 6 && p.use_stokes) || (k == 6 && p.save_direc0) || (k == 7 && p.save_dust_scattered); };
    for (int k = 0; k < 8; k++) { bool on = peel && p.save_peeloff_3D && wanted(k); o3[k] = on ? off : -1; if (on) off += n3; }
    for (int k = 0; k < 8; k++) { bool on = peel && p.save_peeloff_2D && wanted(k) && k != 7; o2[k] = on ? off : -1; if (on) off += n2; }
    int64_t oPeelHa = -1; if (lyb && peel && p.save_peeloff_3D) { oPeelHa = off; off += n2 * nxh; }       // observer(:)%peel_Ha (observer_rect.f90:306-311)
    int64_t oJ1 = -1, oP1 = -1, oPn = -1, oPc = -1;
    c->jpa.lds_bins = 0;
    if (c->jpa.nbins > 0) {
        // P1 / P1_new: per-block LDS copies while they fit
        c->jpa.lds_bins = ((c->jpa.calc_P || c->jpa.calc_Pnew) && c->jpa.nbins <= 2048) ? c->jpa.nbins : 0;
        c->jpa.lds_j_w = 0; c->jpa.lds_j_lo = 0;             // the J1 window is sized at launch, when the workgroup size is known (size_j_window)
        if (c->jpa.calc_J)    { oJ1 = off; off += nxf * c->jpa.nbins; }
        if (c->jpa.calc_P)    { oP1 = off; off += c->jpa.nbins; }
        if (c->jpa.calc_Pnew) { oPn = off; off += c->jpa.nbins; }
        if (c->jpa.calc_P && lyb) { oPc = off; off += c->jpa.nbins; }
    }
    c->ntally = off;
    HIPCHK(hipMalloc(&c->d_tally, off * sizeof(double)));
    HIPCHK(hipMemsetAsync(c->d_tally, 0, off * sizeof(double), c->stream));
    const int64_t len_rep = off - off_rep;
    if (len_rep > 0 && len_rep * 7 * 8 <= (int64_t)24 << 30) {       // 7 more copies, one per XCD, while they stay under 24 GiB
        HIPCHK(hipMalloc(&c->rep.base, (size_t)len_rep * 7 * sizeof(double)));
        HIPCHK(hipMemsetAsync(c->rep.base, 0, (size_t)len_rep * 7 * sizeof(double), c->stream));
        c->rep.off = off_rep; c->rep.len = len_rep; c->rep.nrep = 8;
    }
    c->t.scal = c->d_tally; c->t.Jout = c->d_tally + oJout; c->t.Jin = c->d_tally + oJin; c->t.Jabs = c->d_tally + oJabs; c->t.Jmu = c->d_tally + oJmu;
    for (int k = 0; k < 8; k++) { c->t.c3[k] = o3[k] >= 0 ? c->d_tally + o3[k] : nullptr; c->t.c2[k] = o2[k] >= 0 ? c->d_tally + o2[k] : nullptr; }
    c->jpa.J1 = oJ1 >= 0 ? c->d_tally + oJ1 : nullptr; c->jpa.P1 = oP1 >= 0 ? c->d_tally + oP1 : nullptr; c->jpa.P1_new = oPn >= 0 ? c->d_tally + oPn : nullptr;
    c->jpa.Pconv = oPc >= 0 ? c->d_tally + oPc : nullptr;
    c->t.Jout_Ha = lyb ? c->d_tally + oJoutHa : nullptr; c->t.Jabs_Ha = lyb ? c->d_tally + oJabsHa : nullptr; c->t.peel_Ha = oPeelHa >= 0 ? c->d_tally + oPeelHa : nullptr;
    return 0;
}

// Cell::pad of a Cartesian medium = R, the half width in cells of the largest cube around the cell that lies inside the grid and holds nothing but copies of this
// cell (opacities, Voigt parameter, Doppler width, velocity: compared bit for bit).  A peel ray that starts anywhere in the cell keeps its frequency and its opacity
// for the whole way out of that cube, at least R more cells on the axis it leaves through: the prefilter of peel_emit (peel_all_opaque) tests that path instead of
// the rest of the first cell.  Chebyshev distance to the nearest cell that differs from one of its six neighbours or touches the grid boundary, minus one, by the
// two-pass chamfer transform over the 26-neighbourhood (exact for that metric); capped at 1000.
// Chebyshev distance of every cell of an nx x ny x nz array to the nearest cell with d = 0: the two-pass chamfer transform over the 26-neighbourhood (exact for that metric)
static void chamfer_chebyshev(std::vector<unsigned short> &d, int nx, int ny, int nz) {
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    for (int pass = 0; pass < 2; pass++)
        for (int kk = 0; kk < nz; kk++) for (int jj = 0; jj < ny; jj++) for (int ii = 0; ii < nx; ii++) {
            const int k = pass ? nz - 1 - kk : kk, j = pass ? ny - 1 - jj : jj, i = pass ? nx - 1 - ii : ii, f = pass ? -1 : 1;      // backward pass: the mirrored offsets
            const size_t q = (size_t)i + sy * j + sz * k;
            unsigned m = d[q]; if (m == 0) continue;
            for (int dk = -1; dk <= 0; dk++) for (int dj = -1; dj <= 1; dj++) for (int di = -1; di <= 1; di++) {      // the 13 neighbours in front of this one in scan order
                if (dk == 0 && (dj > 0 || (dj == 0 && di >= 0))) continue;
                const int k2 = k + f * dk, j2 = j + f * dj, i2 = i + f * di;
                if (k2 < 0 || k2 >= nz || j2 < 0 || j2 >= ny || i2 < 0 || i2 >= nx) continue;
                const unsigned v = d[(size_t)i2 + sy * j2 + sz * k2] + 1u; if (v < m) m = v;
            }
            d[q] = (unsigned short)m;
        }
}
static bool same_medium(const Cell &a, const Cell &b) { return std::memcmp(&a, &b, 7 * sizeof(double)) == 0; }      // the seven fields in front of pad, bit for bit
static void uniform_cubes(std::vector<Cell> &h, int nx, int ny, int nz) {
    const size_t nc = h.size(), sx = 1, sy = (size_t)nx, sz = (size_t)nx * ny;
    std::vector<unsigned short> d(nc, 1001);
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
        const size_t q = (size_t)i + sy * j + sz * k;
        bool bad = i == 0 || j == 0 || k == 0 || i == nx - 1 || j == ny - 1 || k == nz - 1;
        if (!bad) bad = !same_medium(h[q], h[q - sx]) || !same_medium(h[q], h[q + sx]) || !same_medium(h[q], h[q - sy]) || !same_medium(h[q], h[q + sy]) || !same_medium(h[q], h[q - sz]) || !same_medium(h[q], h[q + sz]);
        if (bad) d[q] = 0;
    }
    chamfer_chebyshev(d, nx, ny, nz);
    for (size_t q = 0; q < nc; q++) h[q].pad = d[q] > 0 ? (double)(d[q] - 1) : 0.0;
}
// The same for the leaves of an octree: Cell::pad = a length M such that the leaf's box grown by M on every side lies inside the root box and holds nothing but leaves
// with this leaf's medium.  The leaves are painted onto the uniform grid of the finest level (at most 256^3 cells, else no margins), the cube radius R of every fine
// cell comes from the same transform, and M = (the smallest R on the leaf's surface layer) x the fine cell width: every point within M of the leaf is within M of one
// of its surface cells, i.e. inside that cell's cube.
static void uniform_margins_amr(std::vector<Cell> &h, int nleaf, double lo, double hi, const double *cx, const double *cy, const double *cz, const double *ch, const int32_t *icell_of_leaf) {
    double hmin = 1e300; for (int i = 0; i < nleaf; i++) hmin = std::min(hmin, ch[icell_of_leaf[i] - 1]);
    const double w = 2.0 * hmin, fn = (hi - lo) / w; const int n = (int)std::llround(fn);
    if (n < 3 || n > 256 || std::fabs(fn - n) > 1e-6 * n) return;
    const size_t sy = (size_t)n, sz = (size_t)n * n;
    std::vector<int> id((size_t)n * n * n, -1);
    struct Box { int i0, j0, k0, s; }; std::vector<Box> box(nleaf);
    for (int L = 0; L < nleaf; L++) {
        const int ic = icell_of_leaf[L] - 1; const double hw = ch[ic];
        const int s = (int)std::llround(hw / hmin), i0 = (int)std::llround((cx[ic] - hw - lo) / w), j0 = (int)std::llround((cy[ic] - hw - lo) / w), k0 = (int)std::llround((cz[ic] - hw - lo) / w);
        if (s < 1 || i0 < 0 || j0 < 0 || k0 < 0 || i0 + s > n || j0 + s > n || k0 + s > n) return;      // not the regular refinement this needs: no margins
        box[L] = {i0, j0, k0, s};
        for (int k = k0; k < k0 + s; k++) for (int j = j0; j < j0 + s; j++) for (int i = i0; i < i0 + s; i++) id[(size_t)i + sy * j + sz * k] = L;
    }
    std::vector<unsigned short> d(id.size(), 1001);
    for (int k = 0; k < n; k++) for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
        const size_t q = (size_t)i + sy * j + sz * k; const int a = id[q];
        bool bad = a < 0 || i == 0 || j == 0 || k == 0 || i == n - 1 || j == n - 1 || k == n - 1;
        if (!bad) { const size_t nb[6] = {q - 1, q + 1, q - sy, q + sy, q - sz, q + sz};
            for (int t = 0; t < 6 && !bad; t++) { const int b = id[nb[t]]; bad = b < 0 || (b != a && !same_medium(h[a], h[b])); } }
        if (bad) d[q] = 0;
    }
    chamfer_chebyshev(d, n, n, n);
    for (int L = 0; L < nleaf; L++) {
        const Box &b = box[L]; unsigned m = 1001;
        for (int k = b.k0; k < b.k0 + b.s; k++) for (int j = b.j0; j < b.j0 + b.s; j++) for (int i = b.i0; i < b.i0 + b.s; i++) {
            if (k != b.k0 && k != b.k0 + b.s - 1 && j != b.j0 && j != b.j0 + b.s - 1 && i != b.i0 && i != b.i0 + b.s - 1) continue;      // the surface layer only
            m = std::min(m, (unsigned)d[(size_t)i + sy * j + sz * k]);
        }
        h[L].pad = m > 1 ? (double)(m - 1) * w * (1.0 - 1e-12) : 0.0;
    }
}

extern "C" int lart_gpu_set_grid_car(lart_gpu *c, int32_t nx, int32_t ny, int32_t nz, double xmin, double ymin, double zmin,
                                     double dx, double dy, double dz, int32_t i0, int32_t j0, int32_t k0,
                                     const double *xface, const double *yface, const double *zface,
                                     const double *rhokap, const double *voigt_a, const double *Dfreq,
                                     const double *vfx, const double *vfy, const double *vfz, const double *rhokapD) {
    if (!c) return fail("set_grid_car: null context");
    if (nx < 1 || ny < 1 || nz < 1) return fail("set_grid_car: nx, ny, nz must be >= 1");
    if (!xface || !yface || !zface || !rhokap || !voigt_a || !Dfreq || !vfx || !vfy || !vfz) return fail("set_grid_car: null table");
    if (c->par.DGR > 0.0 && !rhokapD) return fail("set_grid_car: DGR > 0 needs rhokapD");
    if (c->par.xyz_symmetry && (i0 < 1 || j0 < 1 || k0 < 1 || i0 > nx || j0 > ny || k0 > nz)) return fail("set_grid_car: xyz_symmetry needs mirror indices i0,j0,k0 inside the grid");
    HIPCHK(hipSetDevice(c->device));
    size_t nc = (size_t)nx * ny * nz;
    std::vector<Cell> h(nc);
    for (size_t i = 0; i < nc; i++) {
        h[i].rhokap = rhokap[i]; h[i].voigt_a = voigt_a[i]; h[i].Dfreq = Dfreq[i];
        h[i].vfx = vfx[i]; h[i].vfy = vfy[i]; h[i].vfz = vfz[i]; h[i].rhokapD = rhokapD ? rhokapD[i] : 0.0; h[i].pad = 0.0;
        if (!(Dfreq[i] > 0.0)) return fail("set_grid_car: Dfreq must be > 0 in every cell (grid_mod_car.f90:276-277)");
    }
    uniform_cubes(h, nx, ny, nz);
    (void)hipFree(c->d_cell); (void)hipFree(c->d_faces); c->d_cell = nullptr; c->d_faces = nullptr;
    HIPCHK(hipMalloc(&c->d_cell, nc * sizeof(Cell)));
    HIPCHK(lart_put(c->stream, c->d_cell, h.data(), nc * sizeof(Cell)));
    size_t nf = (size_t)nx + ny + nz + 3;
    std::vector<double> f(nf);
    std::memcpy(&f[0], xface, (nx + 1) * sizeof(double)); std::memcpy(&f[nx + 1], yface, (ny + 1) * sizeof(double)); std::memcpy(&f[nx + ny + 2], zface, (nz + 1) * sizeof(double));
    HIPCHK(hipMalloc(&c->d_faces, nf * sizeof(double)));
    HIPCHK(lart_put(c->stream, c->d_faces, f.data(), nf * sizeof(double)));
    c->h_faces = f;
    c->g.nx = nx; c->g.ny = ny; c->g.nz = nz; c->g.i0 = i0; c->g.j0 = j0; c->g.k0 = k0;
    c->g.xmin = xmin; c->g.ymin = ymin; c->g.zmin = zmin; c->g.dx = dx; c->g.dy = dy; c->g.dz = dz;
    c->g.cell = c->d_cell; c->g.faces = c->d_faces; c->ncell = nc; c->has_dust_table = rhokapD != nullptr;
    c->is_amr = false; c->is_clump = false; c->jpa.nbins = 0; invalidate_run(c);
    return 0;
}

extern "C" int lart_gpu_set_line_multiplet(lart_gpu *c, const lart_line_multiplet *m) {
    if (!c || !m) return fail("set_line_multiplet: null argument");
    if (m->line_type != c->par.line_type || m->line_type < 4 || m->line_type > 6) return fail("set_line_multiplet: line_type must be 4, 5 or 6 and equal lart_params.line_type");
    if (m->nup < 1 || m->nup > LART_LINE_MAX_UP || (m->line_type == 4 && m->nup != 1) || (m->line_type == 5 && m->nup != 2) || (m->line_type == 6 && m->nup != 3))
        return fail("set_line_multiplet: nup must be 1 (type 4), 2 (type 5) or 3 (type 6)");
    double tab[60] = {0}; int itab[15] = {0};
    for (int i = 0; i < m->nup; i++) {
        const int nd = m->ndown[i];
        if (nd < 1 || nd > LART_LINE_MAX_DOWN || !(m->f12[i] > 0.0) || !(m->damping_up[i] > 0.0)) return fail("set_line_multiplet: bad level record (ndown 1..4, f12 > 0, damping_up > 0)");
        itab[i] = nd;
        for (int d = 0; d < nd; d++) {
            if (nd > 1 && (m->A_down[i][d] < 0 || m->A_down[i][d] > nd || !(m->P_down[i][d] >= 0.0))) return fail("set_line_multiplet: bad alias table (A_down 0..ndown, 1-based; P_down >= 0)");
            itab[3 + i * 4 + d] = m->A_down[i][d] > 0 ? m->A_down[i][d] : d + 1;
            tab[i * 4 + d] = m->Elow_Hz[i][d]; tab[12 + i * 4 + d] = m->E1[i][d]; tab[24 + i * 4 + d] = m->E2[i][d]; tab[36 + i * 4 + d] = m->E3[i][d];
            tab[48 + i * 4 + d] = m->P_down[i][d];
        }
    }
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_ml_tab) { HIPCHK(hipMalloc(&c->d_ml_tab, sizeof tab)); HIPCHK(hipMalloc(&c->d_ml_itab, sizeof itab)); }
    HIPCHK(lart_put(c->stream, c->d_ml_tab, tab, sizeof tab)); HIPCHK(lart_put(c->stream, c->d_ml_itab, itab, sizeof itab));
    c->ml = DMulti{}; c->ml.nup = m->nup; c->ml.coherent = (m->HeI_coherent != 0 && m->line_type == 6) ? 1 : 0; c->ml.tab = c->d_ml_tab; c->ml.itab = c->d_ml_itab;
    for (int i = 0; i < 3; i++) { c->ml.delE_Hz[i] = i < m->nup ? m->delE_Hz[i] : 0.0; c->ml.damping_up[i] = i < m->nup ? m->damping_up[i] : 1.0; c->ml.f12[i] = i < m->nup ? m->f12[i] : 0.0; }
    c->has_ml = true;
    return 0;
}

extern "C" int lart_gpu_set_mueller(lart_gpu *c, int32_t nPDF, const double *coss, const double *S11, const double *S12,
                                    const double *S33, const double *S34, const double *phase_PDF, const int32_t *alias) {
    if (!c) return fail("set_mueller: null context");
    if (nPDF < 2 || !coss || !S11 || !S12 || !S33 || !S34 || !phase_PDF || !alias) return fail("set_mueller: bad table");
    for (int i = 0; i < nPDF - 1; i++) if (alias[i] < 0 || alias[i] > nPDF - 1) return fail("set_mueller: alias entries must be 0..nPDF-1 (1-based, 0 = unused)");
    HIPCHK(hipSetDevice(c->device));
    (void)hipFree(c->d_mu); (void)hipFree(c->d_alias);
    size_t n = nPDF;
    std::vector<double> h(6 * n, 0.0);
    std::memcpy(&h[0], coss, n * 8); std::memcpy(&h[n], S11, n * 8); std::memcpy(&h[2 * n], S12, n * 8);
    std::memcpy(&h[3 * n], S33, n * 8); std::memcpy(&h[4 * n], S34, n * 8); std::memcpy(&h[5 * n], phase_PDF, (n - 1) * 8);
    HIPCHK(hipMalloc(&c->d_mu, 6 * n * 8)); HIPCHK(lart_put(c->stream, c->d_mu, h.data(), 6 * n * 8));
    HIPCHK(hipMalloc(&c->d_alias, n * 4)); HIPCHK(hipMemsetAsync(c->d_alias, 0, n * 4, c->stream));
    HIPCHK(lart_put(c->stream, c->d_alias, alias, (n - 1) * 4));
    c->mu.nPDF = nPDF; c->mu.coss = c->d_mu; c->mu.S11 = c->d_mu + n; c->mu.S12 = c->d_mu + 2 * n; c->mu.S33 = c->d_mu + 3 * n; c->mu.S34 = c->d_mu + 4 * n;
    c->mu.phase_PDF = c->d_mu + 5 * n; c->mu.alias = c->d_alias;
    return 0;
}

extern "C" int lart_gpu_set_observers(lart_gpu *c, int32_t nobs, const double *xyz, const double *rmatrix, double dxim, double dyim, int32_t nxim, int32_t nyim) {
    if (!c) return fail("set_observers: null context");
    if (nobs < 0 || nobs > LART_MAX_OBSERVERS) return fail("set_observers: nobs out of range");
    if (nobs > 0 && (!xyz || !rmatrix || nxim < 1 || nyim < 1 || !(dxim > 0.0) || !(dyim > 0.0))) return fail("set_observers: bad image geometry");
    HIPCHK(hipSetDevice(c->device));
    (void)hipFree(c->d_obs); c->d_obs = nullptr;
    std::vector<DObs> h(nobs > 0 ? nobs : 1);
    for (int i = 0; i < nobs; i++) { h[i].x = xyz[3 * i]; h[i].y = xyz[3 * i + 1]; h[i].z = xyz[3 * i + 2]; for (int q = 0; q < 9; q++) h[i].rm[q] = rmatrix[9 * i + q]; }
    HIPCHK(hipMalloc(&c->d_obs, h.size() * sizeof(DObs)));
    HIPCHK(lart_put(c->stream, c->d_obs, h.data(), h.size() * sizeof(DObs)));
    c->nobs = nobs; c->nxim = nxim; c->nyim = nyim; c->dxim = dxim; c->dyim = dyim; c->heal_nside = 0; h.resize(nobs > 0 ? nobs : 0); c->h_obs = h;
    invalidate_run(c);
    return 0;
}

extern "C" int lart_gpu_set_observers_heal(lart_gpu *c, int32_t nobs, const double *xyz, int32_t nside) {
    if (!c) return fail("set_observers_heal: null context");
    if (nobs < 1 || nobs > LART_MAX_OBSERVERS || !xyz) return fail("set_observers_heal: nobs out of range");
    bool pow2 = nside >= 1 && nside <= 8192 && (nside & (nside - 1)) == 0;
    if (!pow2) return fail("set_observers_heal: nside must be a power of 2 in 1..8192 (nside2npix, healpix.f90:5-27)");
    if (c->par.use_stokes) return fail("set_observers_heal: Stokes parameters are not defined for the HEALPix path (setup.f90:181)");
    if (c->par.xyz_symmetry || c->par.xy_periodic) return fail("set_observers_heal: inside observers need the plain Cartesian grid");
    HIPCHK(hipSetDevice(c->device));
    (void)hipFree(c->d_obs); c->d_obs = nullptr;
    std::vector<DObs> h(nobs);
    for (int i = 0; i < nobs; i++) { h[i].x = xyz[3 * i]; h[i].y = xyz[3 * i + 1]; h[i].z = xyz[3 * i + 2]; for (int q = 0; q < 9; q++) h[i].rm[q] = 0.0; }
    HIPCHK(hipMalloc(&c->d_obs, h.size() * sizeof(DObs)));
    HIPCHK(lart_put(c->stream, c->d_obs, h.data(), h.size() * sizeof(DObs)));
    c->nobs = nobs; c->nxim = 12 * nside * nside; c->nyim = 1; c->dxim = 1.0; c->dyim = 1.0; c->heal_nside = nside; c->h_obs = h;
    invalidate_run(c);
    return 0;
}

extern "C" int lart_gpu_set_JPa(lart_gpu *c, int32_t calc_J, int32_t calc_P, int32_t calc_Pnew, int32_t nbins, const int32_t *ind_bin, double cross0) {
    if (!c) return fail("set_JPa: null context");
    if (c->is_clump) return fail("set_JPa: profile tallies are not defined for the clump medium");
    if (!c->d_cell || c->is_amr) return fail("set_JPa: set the Cartesian medium first (lart_gpu_set_grid_car); the octree uses lart_gpu_set_JPa_amr");
    if (nbins < 1 || !ind_bin || !(cross0 > 0.0)) return fail("set_JPa: bad arguments");
    for (size_t i = 0; i < c->ncell; i++) if (ind_bin[i] < 0 || ind_bin[i] > nbins) return fail("set_JPa: ind_bin entries must be 0..nbins");
    HIPCHK(hipSetDevice(c->device));
    (void)hipFree(c->d_ind_bin); c->d_ind_bin = nullptr;
    HIPCHK(hipMalloc(&c->d_ind_bin, c->ncell * sizeof(int)));
    HIPCHK(lart_put(c->stream, c->d_ind_bin, ind_bin, c->ncell * sizeof(int)));
    c->jpa.calc_J = calc_J ? 1 : 0; c->jpa.calc_P = calc_P ? 1 : 0; c->jpa.calc_Pnew = calc_Pnew ? 1 : 0;
    c->jpa.nbins = (calc_J || calc_P || calc_Pnew) ? nbins : 0; c->jpa.ind_bin = c->d_ind_bin; c->jpa.cross0 = cross0;
    invalidate_run(c);
    return 0;
}

extern "C" int lart_gpu_set_grid_amr(lart_gpu *c, int32_t ncells, int32_t nleaf, double lo, double hi,
                                     const double *cx, const double *cy, const double *cz, const double *ch,
                                     const int32_t *children, const int32_t *neighbor, const int32_t *ileaf, const int32_t *icell_of_leaf,
                                     const double *rhokap, const double *voigt_a, const double *Dfreq,
                                     const double *vfx, const double *vfy, const double *vfz, const double *rhokapD) {
    if (!c) return fail("set_grid_amr: null context");
    if (ncells < 1 || nleaf < 1 || nleaf > ncells || !(hi > lo)) return fail("set_grid_amr: need 1 <= nleaf <= ncells and hi > lo");
    if (!cx || !cy || !cz || !ch || !children || !neighbor || !ileaf || !icell_of_leaf) return fail("set_grid_amr: null tree table");
    if (!rhokap || !voigt_a || !Dfreq || !vfx || !vfy || !vfz) return fail("set_grid_amr: null medium table");
    if (c->par.DGR > 0.0 && !rhokapD) return fail("set_grid_amr: DGR > 0 needs rhokapD");
    // every index the kernels follow is checked here, so a bad table cannot send a wave out of bounds
    for (size_t i = 0; i < (size_t)8 * ncells; i++) if (children[i] < 0 || children[i] > ncells) return fail("set_grid_amr: children entries must be 0..ncells");
    for (size_t i = 0; i < (size_t)6 * ncells; i++) if (neighbor[i] < 0 || neighbor[i] > ncells) return fail("set_grid_amr: neighbor entries must be 0..ncells");
    for (int i = 0; i < ncells; i++) {
        if (ileaf[i] < 0 || ileaf[i] > nleaf) return fail("set_grid_amr: ileaf entries must be 0..nleaf");
        if (!(ch[i] > 0.0)) return fail("set_grid_amr: cell half-widths must be > 0");
        if (ileaf[i] == 0) { bool any = false; for (int o = 0; o < 8; o++) any = any || children[8 * (size_t)i + o] > 0; if (!any) return fail("set_grid_amr: an internal cell has no children"); }
        else if (icell_of_leaf[ileaf[i] - 1] != i + 1) return fail("set_grid_amr: ileaf / icell_of_leaf are not inverse of each other");
        for (int o = 0; o < 8; o++) { int k = children[8 * (size_t)i + o]; if (k != 0 && k <= i + 1) return fail("set_grid_amr: children must have larger cell numbers than their parent (amr_build_tree order)"); }
    }
    for (int i = 0; i < nleaf; i++) {
        if (icell_of_leaf[i] < 1 || icell_of_leaf[i] > ncells) return fail("set_grid_amr: icell_of_leaf entries must be 1..ncells");
        if (!(Dfreq[i] > 0.0)) return fail("set_grid_amr: Dfreq must be > 0 in every leaf");
    }
    HIPCHK(hipSetDevice(c->device));
    std::vector<Cell> h(nleaf);
    for (int i = 0; i < nleaf; i++) {
        h[i].rhokap = rhokap[i]; h[i].voigt_a = voigt_a[i]; h[i].Dfreq = Dfreq[i];
        h[i].vfx = vfx[i]; h[i].vfy = vfy[i]; h[i].vfz = vfz[i]; h[i].rhokapD = rhokapD ? rhokapD[i] : 0.0; h[i].pad = 0.0;
    }
    uniform_margins_amr(h, nleaf, lo, hi, cx, cy, cz, ch, icell_of_leaf);
    (void)hipFree(c->d_cell); (void)hipFree(c->d_faces); (void)hipFree(c->d_amr_int); (void)hipFree(c->d_amr_geom);
    c->d_cell = nullptr; c->d_faces = nullptr; c->d_amr_int = nullptr; c->d_amr_geom = nullptr;
    HIPCHK(hipMalloc(&c->d_cell, (size_t)nleaf * sizeof(Cell)));
    HIPCHK(lart_put(c->stream, c->d_cell, h.data(), (size_t)nleaf * sizeof(Cell)));
    double f[6] = {lo, hi, lo, hi, lo, hi};
    HIPCHK(hipMalloc(&c->d_faces, sizeof f)); HIPCHK(lart_put(c->stream, c->d_faces, f, sizeof f));
    std::vector<AmrGeom> gq(ncells);
    for (int i = 0; i < ncells; i++) { gq[i].cx = cx[i]; gq[i].cy = cy[i]; gq[i].cz = cz[i]; gq[i].ch = ch[i]; }
    HIPCHK(hipMalloc(&c->d_amr_geom, (size_t)ncells * sizeof(AmrGeom)));
    HIPCHK(lart_put(c->stream, c->d_amr_geom, gq.data(), (size_t)ncells * sizeof(AmrGeom)));
    size_t ni = (size_t)15 * ncells + nleaf;
    HIPCHK(hipMalloc(&c->d_amr_int, ni * sizeof(int)));
    int *d = c->d_amr_int;
    HIPCHK(lart_put(c->stream, d, children, (size_t)8 * ncells * 4));
    HIPCHK(lart_put(c->stream, d + (size_t)8 * ncells, neighbor, (size_t)6 * ncells * 4));
    HIPCHK(lart_put(c->stream, d + (size_t)14 * ncells, ileaf, (size_t)ncells * 4));
    HIPCHK(lart_put(c->stream, d + (size_t)15 * ncells, icell_of_leaf, (size_t)nleaf * 4));
    c->amr = DAmr{};
    c->amr.ncells = ncells; c->amr.nleaf = nleaf; c->amr.children = d; c->amr.neighbor = d + (size_t)8 * ncells;
    c->amr.ileaf = d + (size_t)14 * ncells; c->amr.icell_of_leaf = d + (size_t)15 * ncells; c->amr.geom = c->d_amr_geom;
    c->amr.lo = lo; c->amr.hi = hi;
    // leaf records are addressed as cell (il,1,1) of a 1x1x1 "grid": cidx() = il-1
    c->g = DGrid{}; c->g.nx = 1; c->g.ny = 1; c->g.nz = 1; c->g.i0 = 1; c->g.j0 = 1; c->g.k0 = 1;
    c->g.xmin = lo; c->g.ymin = lo; c->g.zmin = lo; c->g.dx = hi - lo; c->g.dy = hi - lo; c->g.dz = hi - lo;
    c->g.cell = c->d_cell; c->g.faces = c->d_faces; c->ncell = nleaf; c->has_dust_table = rhokapD != nullptr;
    c->is_amr = true; c->is_clump = false; c->jpa.nbins = 0; invalidate_run(c);
    return 0;
}

extern "C" int lart_gpu_set_JPa_amr(lart_gpu *c, int32_t calc_J, int32_t calc_P, int32_t calc_Pnew, int32_t geometry_JPa, int32_t nbins, double dbin, double cross0) {
    if (!c) return fail("set_JPa_amr: null context");
    if (!c->is_amr) return fail("set_JPa_amr: set the octree first (lart_gpu_set_grid_amr)");
    if (!(cross0 > 0.0) || nbins < 1) return fail("set_JPa_amr: bad arguments");
    if (geometry_JPa != 1 && geometry_JPa != -1 && geometry_JPa != 3) return fail("set_JPa_amr: geometry_JPa must be 1 (shells), -1 (z slabs) or 3 (leaves)");
    if (geometry_JPa == 3 && nbins != c->amr.nleaf) return fail("set_JPa_amr: geometry_JPa = 3 needs nbins = nleaf");
    if (geometry_JPa != 3 && !(dbin > 0.0)) return fail("set_JPa_amr: bin width must be > 0");
    c->amr.jpa_geometry = geometry_JPa; c->amr.dr_JPa = dbin; c->amr.dz_JPa = dbin; c->amr.nr_JPa = nbins; c->amr.nz_JPa = nbins;
    c->jpa.calc_J = calc_J ? 1 : 0; c->jpa.calc_P = calc_P ? 1 : 0; c->jpa.calc_Pnew = calc_Pnew ? 1 : 0;
    c->jpa.nbins = (calc_J || calc_P || calc_Pnew) ? nbins : 0; c->jpa.ind_bin = nullptr; c->jpa.cross0 = cross0;
    invalidate_run(c);
    return 0;
}

extern "C" int lart_gpu_set_clumps(lart_gpu *c, int64_t n, const double *x, const double *y, const double *z,
                                   const double *vx, const double *vy, const double *vz, const double *radius,
                                   const double *rhokap, const double *rhokapD, const double *voigt_a, const double *Dfreq,
                                   double sphere_R, double Dfreq_ref, double voigt_a_ref,
                                   int32_t cgx, double cg_min, double cg_d, const int32_t *cg_start, const int32_t *cg_list) {
    if (!c) return fail("set_clumps: null context");
    if (n < 1 || n > 2147483647LL) return fail("set_clumps: need 1 <= n < 2^31 clumps");
    if (!x || !y || !z || !vx || !vy || !vz || !radius || !rhokap || !voigt_a || !Dfreq || !cg_start || !cg_list) return fail("set_clumps: null table");
    if (c->par.DGR > 0.0 && !rhokapD) return fail("set_clumps: DGR > 0 needs rhokapD");
    if (!(sphere_R > 0.0) || !(Dfreq_ref > 0.0) || !(voigt_a_ref > 0.0) || cgx < 1 || cgx > 1024 || !(cg_d > 0.0)) return fail("set_clumps: bad scalar argument");
    if (c->par.xyz_symmetry || c->par.xy_periodic) return fail("set_clumps: the clump medium has no symmetry / periodic variant (grid_mod_clump.f90:62-65)");
    if (cg_min > -sphere_R || cg_min + cgx * cg_d < sphere_R) return fail("set_clumps: the CSR grid must cover the sphere");
    // every index the kernels follow is checked here
    size_t ncells = (size_t)cgx * cgx * cgx;
    if (cg_start[0] != 1) return fail("set_clumps: cg_start[0] must be 1 (1-based offsets)");
    for (size_t q = 0; q < ncells; q++) if (cg_start[q + 1] < cg_start[q]) return fail("set_clumps: cg_start must be non-decreasing");
    int64_t nreg = (int64_t)cg_start[ncells] - 1;
    for (int64_t i = 0; i < nreg; i++) if (cg_list[i] < 1 || cg_list[i] > n) return fail("set_clumps: cg_list entries must be 1..n");
    for (int64_t i = 0; i < n; i++) {
        if (!(radius[i] > 0.0) || !(Dfreq[i] > 0.0) || !(rhokap[i] >= 0.0) || !(voigt_a[i] > 0.0)) return fail("set_clumps: radius, Dfreq, voigt_a must be > 0 and rhokap >= 0 for every clump");
        if (!std::isfinite(x[i]) || !std::isfinite(y[i]) || !std::isfinite(z[i]) || !std::isfinite(vx[i]) || !std::isfinite(vy[i]) || !std::isfinite(vz[i])) return fail("set_clumps: non-finite clump position / velocity");
    }
    HIPCHK(hipSetDevice(c->device));
    std::vector<ClumpGeo> hg((size_t)n); std::vector<ClumpPhys> hp((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        hg[i].x = x[i]; hg[i].y = y[i]; hg[i].z = z[i]; hg[i].r2 = radius[i] * radius[i];
        hp[i].rhokap = rhokap[i]; hp[i].rhokapD = rhokapD ? rhokapD[i] : 0.0; hp[i].voigt_a = voigt_a[i]; hp[i].Dfreq = Dfreq[i];
        hp[i].vx = vx[i]; hp[i].vy = vy[i]; hp[i].vz = vz[i]; hp[i].pad = 0.0;
    }
    (void)hipFree(c->d_cl_geo); (void)hipFree(c->d_cl_phys); (void)hipFree(c->d_cl_int); c->d_cl_geo = nullptr; c->d_cl_phys = nullptr; c->d_cl_int = nullptr;
    (void)hipFree(c->d_cell); (void)hipFree(c->d_faces); c->d_cell = nullptr; c->d_faces = nullptr;
    HIPCHK(hipMalloc(&c->d_cl_geo, (size_t)n * sizeof(ClumpGeo))); HIPCHK(lart_put(c->stream, c->d_cl_geo, hg.data(), (size_t)n * sizeof(ClumpGeo)));
    HIPCHK(hipMalloc(&c->d_cl_phys, (size_t)n * sizeof(ClumpPhys))); HIPCHK(lart_put(c->stream, c->d_cl_phys, hp.data(), (size_t)n * sizeof(ClumpPhys)));
    size_t ni = ncells + 1 + (size_t)(nreg > 0 ? nreg : 1);
    HIPCHK(hipMalloc(&c->d_cl_int, ni * sizeof(int)));
    HIPCHK(lart_put(c->stream, c->d_cl_int, cg_start, (ncells + 1) * sizeof(int)));
    if (nreg > 0) HIPCHK(lart_put(c->stream, c->d_cl_int + ncells + 1, cg_list, (size_t)nreg * sizeof(int)));
    c->cl = DClumps{};
    c->cl.N = n; c->cl.geo = c->d_cl_geo; c->cl.phys = c->d_cl_phys; c->cl.cg_start = c->d_cl_int; c->cl.cg_list = c->d_cl_int + ncells + 1;
    c->cl.cgx = cgx; c->cl.cg_min = cg_min; c->cl.cg_d = cg_d; c->cl.cg_inv = 1.0 / cg_d; c->cl.sphere_R = sphere_R; c->cl.Dfreq_ref = Dfreq_ref;
    c->cl.radius_max = 0.0; for (int64_t i = 0; i < n; i++) c->cl.radius_max = std::fmax(c->cl.radius_max, radius[i]);
    (void)hipFree(c->d_cl_mask); c->d_cl_mask = nullptr;
    if (n <= 128) {                                            // membership masks of the CSR cells for the overlap march (lart_device.hpp, ovl_march)
        std::vector<unsigned long long> mk(2 * ncells, 0ull); bool ascending = true;
        for (size_t q = 0; q < ncells && ascending; q++)
            for (int ip = cg_start[q]; ip < cg_start[q + 1]; ip++) {
                const int icl = cg_list[ip - 1];
                if (ip > cg_start[q] && cg_list[ip - 2] >= icl) { ascending = false; break; }      // the walk's order is the list's: only an ascending list is its own mask
                mk[2 * q + ((icl - 1) >> 6)] |= 1ull << ((icl - 1) & 63);
            }
        if (ascending) { HIPCHK(hipMalloc(&c->d_cl_mask, mk.size() * 8)); HIPCHK(lart_put(c->stream, c->d_cl_mask, mk.data(), mk.size() * 8)); c->cl.cg_mask = c->d_cl_mask; }
    }
    // the background medium of grid_create_clump is uniform (rhokap = 0, Dfreq = Dfreq_ref, voigt_a = voigt_a_ref, v = 0): one record
    Cell bg{}; bg.rhokap = 0.0; bg.voigt_a = voigt_a_ref; bg.Dfreq = Dfreq_ref; bg.vfx = bg.vfy = bg.vfz = 0.0; bg.rhokapD = 0.0; bg.pad = 0.0;
    HIPCHK(hipMalloc(&c->d_cell, sizeof(Cell))); HIPCHK(lart_put(c->stream, c->d_cell, &bg, sizeof(Cell)));
    double f[6] = {-sphere_R, sphere_R, -sphere_R, sphere_R, -sphere_R, sphere_R};
    HIPCHK(hipMalloc(&c->d_faces, sizeof f)); HIPCHK(lart_put(c->stream, c->d_faces, f, sizeof f));
    c->g = DGrid{}; c->g.nx = 1; c->g.ny = 1; c->g.nz = 1; c->g.i0 = 1; c->g.j0 = 1; c->g.k0 = 1;
    c->g.xmin = -sphere_R; c->g.ymin = -sphere_R; c->g.zmin = -sphere_R; c->g.dx = 2 * sphere_R; c->g.dy = 2 * sphere_R; c->g.dz = 2 * sphere_R;
    c->g.cell = c->d_cell; c->g.faces = c->d_faces; c->ncell = 1; c->has_dust_table = rhokapD != nullptr;
    c->is_clump = true; c->clump_overlap = false; c->is_amr = false; c->jpa.nbins = 0; invalidate_run(c);
    return 0;
}
// setup_clump_overlap (setup.f90:1053-1082): the overlap-aware marches and scattering wrappers for a population with has_overlap = .true.
extern "C" int lart_gpu_set_clump_overlap(lart_gpu *c, int32_t has_overlap) {
    if (!c) return fail("set_clump_overlap: null context");
    if (!c->is_clump) return fail("set_clump_overlap: call lart_gpu_set_clumps first");
    if (has_overlap && c->par.line_type != 1) return fail("set_clump_overlap: overlapping clumps are built for single-line profiles (line_type 1) only");
    c->clump_overlap = has_overlap != 0; invalidate_run(c);
    return 0;
}

extern "C" int lart_gpu_set_pool(lart_gpu *c, int64_t nlanes, int32_t block) {
    if (!c) return fail("set_pool: null context");
    if (block != 0 && (block % 64 != 0 || block < 64 || block > 1024)) return fail("set_pool: block must be a multiple of 64 up to 1024");
    if (c->begun) return fail("set_pool: call before lart_gpu_begin");
    c->want_lanes = nlanes; if (block) { c->block = block; c->block_user = true; }
    return 0;
}

extern "C" int lart_gpu_test_cell_margin(lart_gpu *c, double *out, int64_t n) {
    if (!c || !out) return fail("test_cell_margin: null argument");
    if (!c->d_cell || n != (int64_t)c->ncell) return fail("test_cell_margin: n must be the number of cells / leaves of the medium");
    HIPCHK(hipSetDevice(c->device));
    std::vector<Cell> h((size_t)n);
    HIPCHK(lart_get(c->stream, h.data(), c->d_cell, (size_t)n * sizeof(Cell)));
    for (int64_t i = 0; i < n; i++) out[i] = h[(size_t)i].pad;
    return 0;
}

extern "C" int lart_gpu_pool_info(lart_gpu *c, int64_t *nlanes, int32_t *block) {
    if (!c) return fail("pool_info: null context");
    if (nlanes) *nlanes = c->pool.n; if (block) *block = c->block;
    return 0;
}

extern "C" int lart_gpu_enable_photon_log(lart_gpu *c, int32_t on) { if (!c) return fail("null context"); c->log_on = on ? (c->log_on == 2 ? 2 : 1) : 0; return 0; }
extern "C" int lart_gpu_enable_all_photons(lart_gpu *c, int32_t on) { if (!c) return fail("null context"); c->log_on = on ? 2 : 0; return 0; }

// J1(x, bin): a window of frequency bins around the line centre is accumulated per workgroup in LDS (ds_add_f64) and flushed once per launch: scattered
// f64 atomics execute at the memory side at ~1e10 per second for the whole chip, which is what one deposit per flight asks for.  The window is as wide as the
// LDS of a CU allows at FULL occupancy: four 256-lane workgroups leave 2 KiB each (four copies of the tables, 32 KiB of draws each), one 1024-lane workgroup
// leaves ~19 KiB -- so a run with J(x) and no Stokes vector uses 1024-lane workgroups unless lart_gpu_set_pool chose the size.  When that is not possible (Stokes:
// 3 waves per SIMD, or a forced size), one workgroup per CU less buys the room.  Octree, 178 480 leaves, events/s on one MI355X: no tallies 17.1e9; J + P + P_new
// 9.3e9 with global atomics, 10.7e9 with the window at three 256-lane workgroups per CU, 12.6e9 with one 1024-lane workgroup.  With J alone the memory-side
// atomics are not yet the limit (14.4e9 against 13.3e9 with the window), so J alone keeps them.
static void size_j_window(lart_gpu *c) {
    c->jpa.lds_j_w = 0; c->jpa.lds_j_lo = 0;
    if (!(c->jpa.nbins > 0 && c->jpa.calc_J) || std::getenv("LART_NO_LDS_J")) return;
    if (!(c->jpa.calc_P || c->jpa.calc_Pnew) && !std::getenv("LART_LDS_J")) return;          // J alone: see above
    const int64_t nxf = c->par.nxfreq;
    const long long base = (long long)(VT_END + c->g.nx + c->g.ny + c->g.nz + 3 + LART_RNG_BUF * c->block + 2 * c->jpa.lds_bins) * 8 + (LART_COOP_BYTES + 4) * ((c->block + 63) / 64);
    const long long tail = (c->par.save_peeloff && c->nobs > 0 && c->nobs <= LART_LDS_OBSERVERS) ? (long long)c->nobs * (long long)sizeof(DObs) + 16 : 0;      // the observers' records behind the window (lart_gpu_step)
    const int waves_cu = c->par.use_stokes ? 12 : 16;
    int per_cu = waves_cu * 64 / c->block; if (per_cu < 1) per_cu = 1;
    long long w = 0;
    for (int give = 0; give < 2 && w < 8; give++) {
        if (give && !(c->jpa.calc_P || c->jpa.calc_Pnew)) break;
        if (per_cu - give < 1) break;
        w = ((long long)160 * 1024 / (per_cu - give) - 256 - tail - base) / (8LL * c->jpa.nbins);
    }
    if (w > 64) w = 64; if (w > nxf) w = nxf;
    if (w < 8) return;
    const int ixc = (int)std::floor((0.0 - c->par.xfreq_min) / c->par.dxfreq);       // 0-based bin of the line centre in reference units
    int lo = ixc - (int)w / 2; if (lo < 0) lo = 0; if (lo + w > nxf) lo = (int)(nxf - w);
    c->jpa.lds_j_w = (int)w; c->jpa.lds_j_lo = lo;
}

// overlapping clumps, at most 64 of them (with their records staged in LDS): the transport kernel whose marches are executed by the wave, one lane per clump (variant 9, ovl_coop).
// LART_OVL_NO_COOP=1 keeps the per-lane lists of variant 7 -- the same histories, for the tests that say so
static bool ovl_coop9(const lart_gpu *c) {
    return c->is_clump && c->clump_overlap && c->cl.cg_mask && c->cl.N <= LART_OVL_COOP_CLUMPS && !std::getenv("LART_OVL_NO_COOP");
}

static int march_variant(const lart_gpu *c);
// waves per SIMD of the transport kernel this context's runs launch (lart_kernel_waves)
static int kernel_waves_of(const lart_gpu *c) {
    const int var = march_variant(c); const bool h2 = c->par.h2_nlines > 0;
    return lart_kernel_waves((var == 7 && ovl_coop9(c)) ? 9 : var, c->par.use_stokes != 0, c->par.DGR > 0.0, c->par.save_peeloff && c->nobs > 0, h2 ? 9 : (int)c->par.line_type);
}

static int ensure_pool(lart_gpu *c) {
    if (!c->block_user) {
        c->block = 256;
        if (c->jpa.nbins > 0 && c->jpa.calc_J && (c->jpa.calc_P || c->jpa.calc_Pnew) && !c->par.use_stokes) {
            // two 512-lane workgroups per CU (round 4; round 3 chose one of 1024): the J(x) window is half as wide, but a finishing workgroup frees half a CU instead of all
            // or nothing -- for the next workgroup and for the peel kernels of the previous slice.  The octree with CALCJ + CALCP + CALCPnew: 12.9e9 events/s at 1024 lanes,
            // 13.9e9 at 512; with a 100 x 100 observer 6.8e9 at 256, 8.2e9 at 512, 7.9e9 at 1024
            c->block = 512; size_j_window(c); if (c->jpa.lds_j_w < 8) c->block = 256; }
        // overlapping clumps, a population small enough for its records to live in LDS (lds_stage_clumps: 96 B per clump): four 256-lane workgroups per CU have no room
        // for four copies, one 1024-lane workgroup has
        // (the wave-cooperative kernel of the smallest populations runs 3 waves per SIMD: three 256-lane workgroups, which do have the room)
        if (c->is_clump && c->clump_overlap && c->cl.cg_mask && c->cl.N <= LART_OVL_LDS_CLUMPS && !c->par.use_stokes && !ovl_coop9(c)) c->block = 1024;
        if (c->block > 256 && kernel_waves_of(c) < 4) { c->block = 256; size_j_window(c); }      // a kernel built for 3 waves per SIMD takes 256-lane workgroups
    }
    int64_t n = c->want_lanes;
    if (n <= 0) {
        hipDeviceProp_t pr; HIPCHK(hipGetDeviceProperties(&pr, c->device));
        n = (int64_t)pr.multiProcessorCount * 6144;            // 96 waves per CU: 6 full residencies at 4 waves/SIMD (lanes sweep in profiles/README.md)
    }
    if (c->count > 0 && n > c->count) n = c->count;
    if (n < c->block) n = c->block;
    n = ((n + c->block - 1) / c->block) * c->block;
    if (c->d_pool && c->pool.n == n) return 0;
    (void)hipFree(c->d_pool); c->d_pool = nullptr;
    (void)hipFree(c->d_compact); c->d_compact = nullptr; c->compact_lanes = 0;     // the slot lists are sized with the pool
    size_t bytes = (size_t)n * (F_COUNT * 8 + 8 + 8 + 4 * 4);
    HIPCHK(hipMalloc(&c->d_pool, bytes));
    char *b = (char *)c->d_pool;
    c->pool.n = n;
    c->pool.f = (double *)b; b += (size_t)n * F_COUNT * 8;
    c->pool.rng_n = (unsigned long long *)b; b += (size_t)n * 8;
    c->pool.k = (long long *)b; b += (size_t)n * 8;
    c->pool.ic = (int *)b; b += (size_t)n * 4; c->pool.jc = (int *)b; b += (size_t)n * 4; c->pool.kc = (int *)b; b += (size_t)n * 4; c->pool.flags = (int *)b;
    return 0;
}

// the ray-march variant, chosen as setup_procedure does (setup.f90:951-980)
static int march_variant(const lart_gpu *c) {
    if (c->is_clump) return c->clump_overlap ? 7 : 4;
    if (c->is_amr) return 3;
    if (c->par.xyz_symmetry) return 1;
    if (c->par.xy_symmetry) return 5;
    if (c->par.xy_periodic) return (c->g.nx == 1 && c->g.ny == 1) ? 2 : (c->par.Omega != 0.0 ? 8 : 6);     // setup.f90:959-973
    return 0;
}

extern "C" int lart_gpu_begin(lart_gpu *c, int64_t first_id, int64_t stride, int64_t count, uint64_t seed) {
    if (!c) return fail("begin: null context");
    if (!c->d_cell) return fail("begin: no medium (call lart_gpu_set_grid_car first)");
    if (count < 0 || stride < 1 || first_id < 0) return fail("begin: need count >= 0, stride >= 1, first_id >= 0");
    if (c->par.save_peeloff && c->nobs == 0) return fail("begin: save_peeloff set but no observers (lart_gpu_set_observers)");
    if (c->heal_nside > 0 && (c->par.use_stokes || c->par.xyz_symmetry || c->par.xy_periodic)) return fail("begin: HEALPix observers need the plain grid without Stokes");
    if (c->par.save_peeloff && c->par.xyz_symmetry) return fail("begin: peel-off is not allowed with xyz_symmetry (setup.f90:198-206)");
    if (c->par.DGR > 0.0 && !c->has_dust_table) return fail("begin: DGR > 0 but the grid has no rhokapD");
    if (c->par.line_type >= 4 && c->par.line_type <= 6 && !c->has_ml) return fail("begin: line_type 4/5/6 needs lart_gpu_set_line_multiplet first");
    if (c->is_clump && c->heal_nside > 0) return fail("begin: HEALPix inside observers are not supported with the clump medium (setup.f90:177)");
    if (c->is_amr && c->heal_nside > 0 && c->par.xyz_symmetry) return fail("begin: inside observers need the full octree");
    if (c->par.xy_symmetry && (c->par.xyz_symmetry || c->par.xy_periodic || c->is_clump)) return fail("begin: xy_symmetry excludes xyz_symmetry, xy_periodic and the clump medium");
    if (c->par.xy_symmetry && !c->is_amr && (c->g.i0 < 1 || c->g.j0 < 1 || c->g.i0 > c->g.nx || c->g.j0 > c->g.ny)) return fail("begin: xy_symmetry needs mirror indices i0, j0 inside the grid");
    if (c->heal_nside > 0 && c->par.xy_symmetry) return fail("begin: HEALPix observers need the plain grid");
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_tally) { if (build_tally(c)) return 1; }
    c->first_id = first_id; c->stride = stride; c->count = count; c->seed = seed;
    if (ensure_pool(c)) return 1;
    HIPCHK(hipMemsetAsync(c->pool.flags, 0, (size_t)c->pool.n * 4, c->stream));
    HIPCHK(hipMemsetAsync(c->d_ctl, 0, sizeof(DCtl), c->stream));
    (void)hipFree(c->d_log); c->d_log = nullptr; c->nlog = 0;
    if (c->log_on && count > 0) { const size_t bytes = (size_t)count * 8 * (c->log_on == 2 ? 12 : 2);
        HIPCHK(hipMalloc(&c->d_log, bytes)); HIPCHK(hipMemsetAsync(c->d_log, 0, bytes, c->stream)); c->nlog = count; }
    for (auto &e : c->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    c->ev.clear(); c->kernel_ms = 0.0; c->nlaunch = 0;
    c->live_lanes = c->pool.n; c->state_valid = false; c->n_compactions = 0; c->compact_alive = c->pool.n; c->compact_waves = 0;
    if (const char *e = std::getenv("LART_DENSE_DRAIN")) c->spread_drain = std::atoi(e) ? 0 : 1;      // tuning knob: pack the last packets densely (round 2 behaviour)
    c->begun = true;
    return 0;
}

extern "C" int lart_gpu_sync(lart_gpu *c, int64_t *done, int64_t *active);
extern "C" int lart_gpu_compact(lart_gpu *c);
static int collect_events(lart_gpu *c) {
    for (auto &e : c->ev) {
        float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
        c->kernel_ms += ms; c->nlaunch++;
        (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
    }
    c->ev.clear();
    return 0;
}

// the peel queue: two buffers of `cap` records of pq_recd(stokes) doubles (224 / 144 B: lart_device.hpp, PQ_*), the producer waves' counts and
// the deposit list; sized once per run for pq_bytes_max of HBM (default: a quarter of the free memory, at most 64 GiB; LART_PEEL_QUEUE_GB)
static int ensure_peel_queue(lart_gpu *c, int max_events) {
    if (!c->n_cu) { hipDeviceProp_t pr; HIPCHK(hipGetDeviceProperties(&pr, c->device)); c->n_cu = pr.multiProcessorCount; }
    if (!c->stream2) {
        // the peel kernels' stream gets the highest priority the device offers: their workgroups are short and few, and the queue buffer they drain is needed again two slices later
        int prio_lo = 0, prio_hi = 0; HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIPCHK(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_hi));
        for (int b = 0; b < 2; b++) { HIPCHK(hipEventCreateWithFlags(&c->ev_filled[b], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_drained[b], hipEventDisableTiming)); }
    }
    const size_t nobs = (size_t)(c->nobs > 0 ? c->nobs : 1);
    const size_t rec_bytes = (size_t)pq_recd(c->par.use_stokes != 0) * 8;
    const size_t lanes = (size_t)c->pool.n;
    if (!c->pq_bytes_max) {
        size_t fr = 0, tot = 0; HIPCHK(hipMemGetInfo(&fr, &tot));
        c->pq_bytes_max = fr / 4 < ((size_t)64 << 30) ? fr / 4 : ((size_t)64 << 30);
        if (const char *e = std::getenv("LART_PEEL_QUEUE_GB")) { const double gb = std::atof(e); if (gb >= 0.25 && gb <= 200.0) c->pq_bytes_max = (size_t)(gb * 1073741824.0); }   // tuning knob
    }
    size_t fit = c->pq_bytes_max / (2 * (rec_bytes + 16 * nobs) * lanes);      // records per lane and buffer the byte budget allows (+ one deposit-list entry per observer)
    int slice = max_events;
    if ((size_t)slice + 1 > fit) slice = fit > 1 ? (int)(fit - 1) : 1;
    const size_t cap = lanes * ((size_t)slice + 1);
    if (cap > 0xfffffff0ull) return fail("peel queue: more than 2^32 records per launch; lower the pool size or max_events");
    if (!c->d_pq || c->pq[0].cap < cap || c->pq_nobs != nobs || c->pq_lanes != lanes) {     // the deposit list scales with nobs, the waves' counters with the lanes
        HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipStreamSynchronize(c->stream2));
        (void)hipFree(c->d_pq); c->d_pq = nullptr;
        // the deposit list: (record, observer, tau) of every (record, observer) pair, plus the unused tail of every marching wave's last chunk
        // (peel_kernel: n_cu x 4 workgroups of 4 waves, LART_DEP_CHUNK = 256 entries; a wave fills its chunks to the last entry)
        const size_t dep_cap = cap * nobs + (size_t)c->n_cu * 16 * 256;
        const size_t cnt_bytes = ((lanes / 64 + 1) * sizeof(unsigned) + 15) / 16 * 16;
        const size_t one = (cap * rec_bytes + cnt_bytes + dep_cap * 16 + sizeof(DPeelCtl) + 255) / 256 * 256;
        HIPCHK(hipMalloc(&c->d_pq, 2 * one));
        for (int b = 0; b < 2; b++) {
            char *base = (char *)c->d_pq + (size_t)b * one; DPeelQ &q = c->pq[b];
            q.f = (double *)base; q.cap = (unsigned)cap;
            q.count = (unsigned *)(base + cap * rec_bytes);
            q.dep_tau = (double *)(base + cap * rec_bytes + cnt_bytes); q.dep_idx = (unsigned *)(q.dep_tau + dep_cap); q.dep_cap = dep_cap;
            q.pctl = (DPeelCtl *)(base + cap * rec_bytes + cnt_bytes + dep_cap * 16);
        }
        c->drained_pending[0] = c->drained_pending[1] = false; c->pq_nobs = nobs; c->pq_lanes = lanes;
    } else slice = (int)(c->pq[0].cap / lanes) - 1 < max_events ? (int)(c->pq[0].cap / lanes) - 1 : max_events;
    c->peel_slice = slice < 1 ? 1 : slice;
    for (int b = 0; b < 2; b++) c->pq[b].region = 64u * ((unsigned)c->peel_slice + 1u);      // one producer wave's slots: every lane queues at most one record per event and one per birth
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Lane compaction.  The reference keeps every worker busy with a dynamic master/worker scheduler (run_simulation_mod.f90:31-64);
// here a lane refills itself from the photon counter, and once the counter is exhausted the pool drains: packets end at very
// different times (<N_scatt> ~ 1e6 at tau0 = 1e6 with a spread of the same size), so waves would run on with a few live lanes.
// Between launches the live lanes beyond the first `m` (= number alive) move into the dead slots below m and the grid shrinks to m.
// A packet's history depends only on (seed, id), so the move changes nothing but the lane a packet sits in.
// ---------------------------------------------------------------------------------------------
__device__ inline int wave_slot(bool pred, unsigned *counter) {           // one atomic per wave: this lane's slot among the lanes with pred
    const unsigned long long b = __ballot(pred);
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == 0 && b) base = atomicAdd(counter, (unsigned)__popcll(b));
    base = __shfl(base, 0, 64);
    return (int)(base + (unsigned)__popcll(b & ((1ull << lane) - 1ull)));
}
// the lanes the m live packets are to sit in: dealt round-robin over `waves` waves -- packet j goes to wave j mod waves, row j div waves -- so that the
// last packets of a run have a wave (and, below one wave per SIMD, a SIMD) each instead of sharing the divergence of a full wave: one event of a lone
// packet takes 5.2 us, one round of a full wave 10.1 us (profiles/r03_lone_wave_sections.txt).  waves = ceil(m / 64): the dense layout.
__device__ inline bool compact_wanted(long long i, long long m, long long waves) {
    const long long w = i >> 6, r = i & 63, q = m / waves, rem = m - q * waves;
    return w < waves && (r < q || (r == q && w < rem));
}
__global__ void k_compact_scan(DPool pool, long long live, long long m, long long waves, int *holes, int *movers, unsigned *cnt) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < live, alive = in && (pool.flags[i] & FLAG_ALIVE), want = compact_wanted(i, m, waves);
    const bool hole = in && want && !alive, mover = in && !want && alive;
    const int sh = wave_slot(hole, &cnt[0]), sm = wave_slot(mover, &cnt[1]);
    if (hole) holes[sh] = (int)i;
    if (mover) movers[sm] = (int)i;
}
__global__ void k_compact_move(DPool pool, const int *holes, const int *movers, const unsigned *cnt, int nplanes, int extra_plane) {
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cnt[1] || j >= cnt[0]) return;
    const long long src = movers[j], dst = holes[j], P = pool.n;
    for (int q = 0; q < nplanes; q++) pool.f[(size_t)q * P + dst] = pool.f[(size_t)q * P + src];
    if (extra_plane >= 0) pool.f[(size_t)extra_plane * P + dst] = pool.f[(size_t)extra_plane * P + src];
    pool.rng_n[dst] = pool.rng_n[src]; pool.k[dst] = pool.k[src];
    pool.ic[dst] = pool.ic[src]; pool.jc[dst] = pool.jc[src]; pool.kc[dst] = pool.kc[src];
    pool.flags[dst] = pool.flags[src]; pool.flags[src] = 0;
}

extern "C" int lart_gpu_compact(lart_gpu *c) {
    if (!c || !c->begun) return fail("compact: call lart_gpu_begin first");
    HIPCHK(hipSetDevice(c->device));
    if (!c->state_valid) { if (lart_gpu_sync(c, nullptr, nullptr)) return 1; }
    if (!c->state_valid) return 0;                                  // nothing launched yet
    const int64_t m = c->last_alive, live = c->live_lanes;
    // wave slots that run side by side: 4 per SIMD (the occupancy the transport kernel is built for); fewer packets than that -> one packet per wave
    if (!c->n_cu) { hipDeviceProp_t pr; HIPCHK(hipGetDeviceProperties(&pr, c->device)); c->n_cu = pr.multiProcessorCount; }
    // Measured on the headline workload (profiles/r03_drain_timeline.txt): one round of a wave costs 4.7 us with one packet in it and 10.1 us with 64
    // when the wave has its SIMD to itself (latency-bound), and 4 us of issue time per full wave otherwise.  So: full waves while there are at least
    // two of them per SIMD; below that the packets are dealt out over two waves per SIMD (half the divergence, the issue slots still suffice: 7.5 us
    // against 10.1 us at 32 against 64 packets per wave), and from one packet per SIMD on every packet has a wave and a SIMD of its own.
    const int64_t wpb = c->block / 64, simds = (int64_t)c->n_cu * 4, slots = 2 * simds;
    int64_t waves = (m + 63) / 64;                                   // dense
    if (c->spread_drain && waves < slots) waves = m < slots ? m : slots;
    if (waves < 1) waves = 1;
    waves = ((waves + wpb - 1) / wpb) * wpb;                         // whole workgroups
    int64_t new_live = waves * 64;
    if (new_live > live) { waves = live / 64; new_live = waves * 64; }       // (the layout never grows the launch)
    if (m > new_live) return 0;
    if (new_live == live && c->compact_waves == waves) { c->compact_alive = m; return 0; }
    if (!c->d_compact || c->compact_lanes < c->pool.n) {             // the pool may have been re-created larger by a later lart_gpu_begin
        (void)hipFree(c->d_compact); c->d_compact = nullptr; c->compact_lanes = 0;
        HIPCHK(hipMalloc(&c->d_compact, ((size_t)2 * c->pool.n + 2) * sizeof(int))); c->compact_lanes = c->pool.n;
    }
    int *holes = c->d_compact + 2, *movers = holes + c->pool.n; unsigned *cnt = (unsigned *)c->d_compact;
    HIPCHK(hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(k_compact_scan, dim3((unsigned)((live + 255) / 256)), dim3(256), 0, c->stream, c->pool, (long long)live, (long long)m, (long long)waves, holes, movers, cnt);
    HIPCHK(hipGetLastError());
    const int64_t nmove_max = m;                                     // at most every live packet moves
    if (nmove_max > 0) {
        hipLaunchKernelGGL(k_compact_move, dim3((unsigned)((nmove_max + 255) / 256)), dim3(256), 0, c->stream, c->pool, holes, movers, cnt, c->par.use_stokes ? (int)F_SHEAR : (int)F_MX, march_variant(c) == 8 ? (int)F_SHEAR : -1);
        HIPCHK(hipGetLastError());
    }
    unsigned h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h, cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (h[0] != h[1]) return fail("compact: live-lane bookkeeping is inconsistent (holes != movers)");
    c->live_lanes = new_live; c->n_compactions++; c->compact_alive = m; c->compact_waves = waves;
    return 0;
}
extern "C" int lart_gpu_set_auto_compact(lart_gpu *c, int32_t on) { if (!c) return fail("null context"); c->auto_compact = on ? 1 : 0; return 0; }
extern "C" int lart_gpu_live_lanes(lart_gpu *c, int64_t *live, int64_t *ncompactions) { if (!c) return fail("null context"); if (live) *live = c->live_lanes; if (ncompactions) *ncompactions = c->n_compactions; return 0; }

extern "C" int lart_gpu_step(lart_gpu *c, int32_t nsteps, int32_t max_events) {
    if (!c || !c->begun) return fail("step: call lart_gpu_begin first (a set_* call ends the current run)");
    if (max_events < 1) return fail("step: max_events must be >= 1");
    if (!c->d_tally || (!c->d_cell && !c->is_clump) || !c->d_pool) return fail("step: no medium / tally buffer / photon pool on the device");
    HIPCHK(hipSetDevice(c->device));
    // drain phase (no photon left to refill a lane with): pack the live lanes to the front of the pool and shrink the grid
    // (again whenever a quarter of the packets of the last layout have finished)
    if (c->auto_compact && c->state_valid && c->last_next_k >= (int64_t)c->count && c->last_alive * 4 < c->compact_alive * 3 && c->live_lanes > c->block) {
        if (lart_gpu_compact(c)) return 1;
    }
    KArgs A{};
    A.par = c->par; A.g = c->g; A.t = c->t; A.pool = c->pool; A.mu = c->mu; A.obs = c->d_obs; A.nobs = c->par.save_peeloff ? c->nobs : 0;
    A.nxim = c->nxim; A.nyim = c->nyim; A.dxim = c->dxim; A.dyim = c->dyim; A.ctl = c->d_ctl; A.heal_nside = c->heal_nside; A.jpa = c->jpa; A.amr = c->amr; A.cl = c->cl; A.rep = c->rep; A.ml = c->ml;
    A.first_id = c->first_id; A.stride = c->stride; A.count = c->count; A.seed = c->seed; A.max_events = max_events;
    A.log_nscatt = c->d_log; A.log_xfreq = c->d_log ? c->d_log + c->nlog : nullptr;
    A.allph = (c->d_log && c->log_on == 2) ? c->d_log + 2 * c->nlog : nullptr; A.allph_n = c->nlog;
    A.trace_slot = c->d_trace_slot; A.trace = c->d_trace; A.trace_n = c->d_trace_n; A.trace_cap = c->trace_cap; A.h2tab = c->d_h2tab;
    int var = march_variant(c);
    const bool h2 = c->par.h2_nlines > 0;
    if (h2) A.par.line_type = 9;                               // pseudo line type of the kernels: Ly-alpha + the H2 lines (lart_device.hpp, h2_kappa_D)
    if ((c->par.line_type == 8 && var != 0 && var != 3) || (h2 && var != 0 && var != 2 && var != 3)) return fail("step: ly_beta runs on the plain Cartesian grid and the octree, the H2 lines also on the 1-column slab (setup.f90:239-280, 302-348)");
    if (c->par.line_type == 8 && (c->heal_nside > 0)) return fail("step: ly_beta: inside-observer (HEALPix) not yet supported (setup.f90:249-253)");
    if (h2 && c->heal_nside > 0) return fail("step: H2: inside-observer (HEALPix) not yet supported");
    int stokes = c->par.use_stokes ? 1 : 0, dust = c->par.DGR > 0.0 ? 1 : 0, peel = (c->par.save_peeloff && c->nobs > 0) ? 1 : 0;
    dim3 block(c->block), grid((unsigned)(c->live_lanes / c->block));
    size_j_window(c); A.jpa = c->jpa;
    // Voigt tables + faces | ring of random draws [LART_RNG_BUF][block] | per-block P1, P1_new
    size_t shmem = (size_t)(VT_END + c->g.nx + c->g.ny + c->g.nz + 3 + LART_RNG_BUF * c->block + 2 * c->jpa.lds_bins + c->jpa.lds_j_w * c->jpa.nbins) * sizeof(double) + (size_t)(LART_COOP_BYTES + 4) * ((c->block + 63) / 64);   // ... + the wave-local scratch of vz_core_coop + the waves' peel-queue counters
    shmem = (shmem + 15) & ~(size_t)15;                        // the tail of the kernel's LDS layout (transport_kernel.hpp), in its order:
    if (c->par.line_type == 8 || h2) shmem += 8 * sizeof(double);                                                           // the block's accumulators of the ly_beta / H2 weight budget
    if (var == 7 && c->cl.cg_mask && c->cl.N <= LART_OVL_LDS_CLUMPS) shmem += (((size_t)c->cl.N * (sizeof(ClumpGeo) + sizeof(ClumpPhys))) + 15) & ~(size_t)15;   // the clump records (lds_stage_clumps)
    if (peel && c->nobs <= LART_LDS_OBSERVERS) shmem += (size_t)c->nobs * sizeof(DObs);                                     // the observers' records of the peel prefilter
    if (shmem > 160 * 1024) return fail("step: grid faces do not fit the 160 KiB LDS");
    const bool coop9 = var == 7 && ovl_coop9(c);
    if (c->block > 256 && kernel_waves_of(c) < 4)
        return fail("step: this transport kernel is built for 3 waves per SIMD (Stokes vector, wave-cooperative overlap kernel, see lart_kernel_waves): workgroups of at most 256 lanes (lart_gpu_set_pool)");
    // peel-off: a launch of `slice` events per lane queues at most live_lanes x (slice + 1) records (one per event, one per birth);
    // the queue is sized once per run for pq_bytes_max (16 GiB by default) and a step of max_events is cut into slices that fit
    int slice = max_events;
    if (peel) {
        if (ensure_peel_queue(c, max_events)) return 1;
        slice = c->peel_slice;
    }
    const size_t peel_shmem = (size_t)(VT_END + c->g.nx + c->g.ny + c->g.nz + 3) * sizeof(double);
    const bool no_march = std::getenv("LART_TIMING_NO_MARCH") != nullptr;      // TIMING EXPERIMENTS ONLY (tools/perf_kernel.py): the queued records are dropped, the peel tallies stay empty
    int nslice = 0;
    for (int s = 0; s < nsteps; s++) {
        hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, c->stream));
        for (int left = max_events; left > 0; left -= slice, nslice++) {
            const int b = nslice & 1;
            A.max_events = left < slice ? left : slice;
            if (peel) {                                           // this buffer's previous records must have been worked through
                A.pq = c->pq[b]; A.pq.nregions = (unsigned)(c->live_lanes / 64);
                if (c->drained_pending[b]) { HIPCHK(hipStreamWaitEvent(c->stream, c->ev_drained[b], 0)); c->drained_pending[b] = false; }
            }
            HIPCHK(hipMemsetAsync(&c->d_ctl->n_active, 0, sizeof(unsigned long long), c->stream));
            if (c->par.line_type == 8 || h2) { if (lart_launch_transport_lt89(var, dust, peel, grid, block, shmem, c->stream, A)) return fail("step: no kernel for this combination of line type and medium"); }
            else if (var == 0) lart_launch_transport_var0(stokes, dust, peel, grid, block, shmem, c->stream, A);
            else if (var == 1) lart_launch_transport_var1(stokes, dust, peel, grid, block, shmem, c->stream, A);
            else if (var == 2) lart_launch_transport_var2(stokes, dust, peel, grid, block, shmem, c->stream, A);
            else if (var == 3) lart_launch_transport_var3(stokes, dust, peel, grid, block, shmem, c->stream, A);
            else if (var == 4) lart_launch_transport_var4(stokes, dust, peel, grid, block, shmem, c->stream, A);
            else if (var == 5) lart_launch_transport_var5(stokes, dust, peel, grid, block, shmem, c->stream, A);
            else if (var == 6) lart_launch_transport_var6(stokes, dust, peel, grid, block, shmem, c->stream, A);
            else if (var == 7 && coop9) lart_launch_transport_var9(stokes, dust, peel, grid, block, shmem, c->stream, A);
            else if (var == 7) lart_launch_transport_var7(stokes, dust, peel, grid, block, shmem, c->stream, A);
            else lart_launch_transport_var8(stokes, dust, peel, grid, block, shmem, c->stream, A);
            HIPCHK(hipGetLastError());
            if (peel) {
                // the peel kernels of this slice run on the second stream, next to the transport kernel of the next slice: the march is bound
                // by gather latency (VALU a quarter busy), the transport kernel by VALU issue
                HIPCHK(hipEventRecord(c->ev_filled[b], c->stream));
                HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_filled[b], 0));
                HIPCHK(hipMemsetAsync(A.pq.pctl, 0, sizeof(DPeelCtl), c->stream2));
                if (!no_march) lart_launch_peel(edge_var(var), stokes, dim3((unsigned)(c->n_cu * 4)), dim3((unsigned)(c->n_cu * 2)), dim3(256), peel_shmem, c->stream2, A);
                HIPCHK(hipGetLastError());
                HIPCHK(hipEventRecord(c->ev_drained[b], c->stream2)); c->drained_pending[b] = true;
            }
        }
        if (peel) for (int b = 0; b < 2; b++)                     // the step is over when its deposits are in: everything after it on `stream` sees the tallies
            if (c->drained_pending[b]) { HIPCHK(hipStreamWaitEvent(c->stream, c->ev_drained[b], 0)); c->drained_pending[b] = false; }
        HIPCHK(hipEventRecord(e1, c->stream));
        c->ev.emplace_back(e0, e1);
    }
    if (nsteps > 0 && c->rep.nrep > 1) c->rep_dirty = true;
    if (nsteps > 0) { c->state_valid = false; c->reduced_valid = false; }
    return 0;
}

extern "C" int lart_gpu_sync(lart_gpu *c, int64_t *done, int64_t *active) {
    if (!c) return fail("sync: null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (collect_events(c)) return 1;
    DCtl h{}; double nd = 0.0;
    HIPCHK(lart_get(c->stream, &h, c->d_ctl, sizeof h));
    if (c->d_tally) HIPCHK(lart_get(c->stream, &nd, c->d_tally + 5, 8));
    if (done) *done = (int64_t)(nd + 0.5);
    if (active) *active = (int64_t)h.n_active + ((int64_t)h.next_k < c->count ? (c->count - (int64_t)h.next_k) : 0);
    if (c->begun && c->nlaunch > 0) { c->last_alive = (int64_t)h.n_active; c->last_next_k = (int64_t)h.next_k; c->state_valid = true; }
    return 0;
}

extern "C" int lart_gpu_kernel_time(lart_gpu *c, double *total_ms, int64_t *nlaunches) {
    if (!c) return fail("kernel_time: null context");
    if (total_ms) *total_ms = c->kernel_ms; if (nlaunches) *nlaunches = c->nlaunch;
    return 0;
}

extern "C" int lart_gpu_run(lart_gpu *c, int64_t first_id, int64_t stride, int64_t count, uint64_t seed) {
    if (lart_gpu_begin(c, first_id, stride, count, seed)) return 1;
    // photons still to be processed before this call (device nphotons_done is cumulative over runs)
    for (;;) {
        if (lart_gpu_step(c, 1, 4096)) return 1;
        int64_t done = 0, active = 0;
        if (lart_gpu_sync(c, &done, &active)) return 1;
        if (active == 0) break;
    }
    return 0;
}


// add the per-XCD copies back into the fused buffer and clear them (DRep)
__global__ void k_fold_copies(double *dst, double *copies, long long len, int ncopies) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double a = 0.0;
    for (int r = 0; r < ncopies; r++) { double v = copies[(size_t)r * len + i]; if (v != 0.0) { a += v; copies[(size_t)r * len + i] = 0.0; } }
    if (a != 0.0) dst[i] += a;
}
static int fold_copies(lart_gpu *c) {
    if (c->rep.nrep > 1 && c->rep_dirty && c->d_tally) {
        unsigned nb = (unsigned)((c->rep.len + 255) / 256);
        hipLaunchKernelGGL(k_fold_copies, dim3(nb), dim3(256), 0, c->stream, c->d_tally + c->rep.off, c->rep.base, (long long)c->rep.len, c->rep.nrep - 1);
        HIPCHK(hipGetLastError());
        c->rep_dirty = false;
    }
    return 0;
}

extern "C" int lart_gpu_tally_buffer(lart_gpu *c, void **p, int64_t *n) {
    if (!c) return fail("tally_buffer: null context");
    if (!c->d_tally) { if (build_tally(c)) return 1; }
    HIPCHK(hipSetDevice(c->device));
    if (fold_copies(c)) return 1;                            // the buffer handed out (RCCL reduce) holds the totals
    HIPCHK(hipStreamSynchronize(c->stream));
    if (p) *p = c->d_tally; if (n) *n = c->ntally;
    return 0;
}
extern "C" int lart_gpu_zero_tallies(lart_gpu *c) {
    if (!c) return fail("zero_tallies: null context");
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_tally) { if (build_tally(c)) return 1; }
    HIPCHK(hipMemsetAsync(c->d_tally, 0, c->ntally * 8, c->stream));
    if (c->rep.base) HIPCHK(hipMemsetAsync(c->rep.base, 0, (size_t)c->rep.len * (c->rep.nrep - 1) * 8, c->stream));
    c->rep_dirty = false; c->reduced_valid = false;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int lart_gpu_get_tallies(lart_gpu *c, lart_tallies *o) {
    if (!c || !o) return fail("get_tallies: null argument");
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_tally) { if (build_tally(c)) return 1; }
    if (fold_copies(c)) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (collect_events(c)) return 1;
    // after output_reduce the root hands out the totals over all ranks (its receive buffer), until it adds events of its own again
    // (the section pointers of c->t point into d_tally; a section of the receive buffer is addressed as base + its OFFSET in d_tally -- no arithmetic between the two allocations)
    const double *base = (c->reduced_valid && c->d_reduced) ? c->d_reduced : c->d_tally;
    double s[16]; HIPCHK(lart_get(c->stream, s, base, sizeof s));
    { double dr = 0.0; HIPCHK(lart_get(c->stream, &dr, base + LART_SCAL_PEEL_DROPPED, 8));
      if (dr > 0.0) return fail("get_tallies: the peel-off deposit list overflowed (" + std::to_string((long long)dr) + " rays dropped): the images have lost flux; this is a sizing bug of the peel queue, not a property of the input"); }
    if (s[7] > 0.0) return fail("get_tallies: a ray met more overlapping clumps than the march lists hold (LART_OVL_ACT clumps around a point, LART_OVL_PEND events ahead): the result is not the reference's");
    o->nscatt_gas = s[0]; o->nscatt_dust = s[1]; o->n_events = s[2]; o->n_cellsteps = s[3]; o->n_cellsteps_peel = s[4]; o->n_atomics = s[6]; o->nphotons_done = (int64_t)(s[5] + 0.5);
    DCtl h{}; HIPCHK(lart_get(c->stream, &h, c->d_ctl, sizeof h)); o->nphotons_active = (int64_t)h.n_active;
    int64_t nxf = c->par.nxfreq, n2 = (int64_t)c->nobs * c->nxim * c->nyim, n3 = n2 * nxf;
#define CP(dst, src, cnt) if ((dst) && (src) && (cnt) > 0) HIPCHK(lart_get(c->stream, (dst), base + ((src) - c->d_tally), (size_t)(cnt) * 8))
    CP(o->Jout, c->t.Jout, nxf); CP(o->Jin, c->t.Jin, nxf); CP(o->Jabs, c->t.Jabs, nxf); CP(o->Jmu, c->t.Jmu, c->par.save_Jmu ? nxf * c->par.nmu : 0);
    double *h3[8] = {o->scatt, o->direc, o->I, o->Q, o->U, o->V, o->direc0, o->scatt_dust}, *h2[8] = {o->scatt_2D, o->direc_2D, o->I_2D, o->Q_2D, o->U_2D, o->V_2D, o->direc0_2D, nullptr};
    for (int k = 0; k < 8; k++) { CP(h3[k], c->t.c3[k], n3); CP(h2[k], c->t.c2[k], n2); }
    CP(o->J1, c->jpa.J1, nxf * c->jpa.nbins); CP(o->P1, c->jpa.P1, c->jpa.nbins); CP(o->P1_new, c->jpa.P1_new, c->jpa.nbins);
    CP(o->Jout_Ha, c->t.Jout_Ha, c->par.nxfreq_Ha); CP(o->Jabs_Ha, c->t.Jabs_Ha, c->par.nxfreq_Ha); CP(o->peel_Ha, c->t.peel_Ha, n2 * c->par.nxfreq_Ha); CP(o->Pconv, c->jpa.Pconv, c->jpa.nbins);
    { double w[8] = {0}; HIPCHK(lart_get(c->stream, w, base + LART_SCAL_AUX, 7 * 8)); for (int i = 0; i < 8; i++) o->aux[i] = w[i]; }
#undef CP
    return 0;
}

// ---------------------------------------------------------------------------------------------
// output_reduce across GPUs (reference: reduce_mem = MPI_REDUCE(SUM) to rank 0, memory_mod_mpi.f90:366-394, called for every
// tally in output_reduce_outside, output_sum_rect.f90:7-149).  Here: ONE ncclReduce(sum, f64) of the fused tally buffer to
// the root over xGMI.  Two ways to form the communicator: one context per process (ranks started by mpirun / torchrun; the
// host broadcasts the 128-byte unique id) or all contexts in this process (ncclCommInitAll).
// ---------------------------------------------------------------------------------------------
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(std::string(#x) + ": " + ncclGetErrorString(r_)); } while (0)

extern "C" int lart_gpu_comm_unique_id(uint8_t *id128) {
    if (!id128) return fail("comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u; NCCLCHK(ncclGetUniqueId(&u));
    std::memcpy(id128, &u, sizeof u);
    return 0;
}
extern "C" int lart_gpu_comm_init(lart_gpu *c, const uint8_t *id128, int32_t nranks, int32_t rank) {
    if (!c || !id128) return fail("comm_init: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("comm_init: need 0 <= rank < nranks");
    HIPCHK(hipSetDevice(c->device));
    if (c->comm) { NCCLCHK(ncclCommDestroy(c->comm)); c->comm = nullptr; }
    ncclUniqueId u; std::memcpy(&u, id128, sizeof u);
    NCCLCHK(ncclCommInitRank(&c->comm, nranks, u, rank));
    int n = 0, r = -1; NCCLCHK(ncclCommCount(c->comm, &n)); NCCLCHK(ncclCommUserRank(c->comm, &r));
    c->comm_n = n; c->comm_rank = r;
    return 0;
}
extern "C" int lart_gpu_comm_info(lart_gpu *c, int32_t *nranks, int32_t *rank) {
    if (!c) return fail("comm_info: null context");
    if (nranks) *nranks = c->comm ? c->comm_n : 0; if (rank) *rank = c->comm ? c->comm_rank : -1;
    return 0;
}
extern "C" int lart_gpu_comm_destroy(lart_gpu *c) {
    if (!c) return fail("comm_destroy: null context");
    if (c->comm) { (void)hipSetDevice(c->device); NCCLCHK(ncclCommDestroy(c->comm)); c->comm = nullptr; c->comm_n = 0; c->comm_rank = -1; }
    return 0;
}
static int reduce_prepare(lart_gpu *c) {
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_tally) { if (build_tally(c)) return 1; }
    return fold_copies(c);                                    // the buffer that travels holds this GPU's totals
}
extern "C" int lart_gpu_reduce_tallies(lart_gpu *c, int32_t root) {
    if (!c) return fail("reduce_tallies: null context");
    if (!c->comm) return fail("reduce_tallies: no communicator (lart_gpu_comm_init first)");
    if (root < 0 || root >= c->comm_n) return fail("reduce_tallies: root out of range");
    if (reduce_prepare(c)) return 1;
    const bool is_root = c->comm_rank == root;
    if (is_root && !c->d_reduced) HIPCHK(hipMalloc(&c->d_reduced, (size_t)c->ntally * sizeof(double)));
    c->reduced_valid = false;
    NCCLCHK(ncclReduce(c->d_tally, is_root ? c->d_reduced : c->d_tally, (size_t)c->ntally, ncclDouble, ncclSum, root, c->comm, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->reduced_valid = is_root;                               // (RCCL writes the receive buffer on the root only) every rank's own sums stay untouched: a second reduce gives the same totals
    return 0;
}
// all contexts live in this process (one host thread drives the GPUs of the node)
extern "C" int lart_gpu_reduce_tallies_multi(lart_gpu **ctx, int32_t n, int32_t root) {
    if (!ctx || n < 1 || root < 0 || root >= n) return fail("reduce_tallies_multi: bad arguments");
    for (int i = 0; i < n; i++) { if (!ctx[i]) return fail("reduce_tallies_multi: null context"); if (reduce_prepare(ctx[i])) return 1; }
    for (int i = 1; i < n; i++) if (ctx[i]->ntally != ctx[0]->ntally) return fail("reduce_tallies_multi: contexts have different tally layouts");
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (ctx[i]->device == ctx[j]->device) return fail("reduce_tallies_multi: two contexts share a device");
    if (n == 1) { HIPCHK(hipStreamSynchronize(ctx[0]->stream)); return 0; }
    { lart_gpu *r = ctx[root]; HIPCHK(hipSetDevice(r->device)); if (!r->d_reduced) HIPCHK(hipMalloc(&r->d_reduced, (size_t)r->ntally * sizeof(double))); }
    for (int i = 0; i < n; i++) ctx[i]->reduced_valid = false;
    bool have = true;
    for (int i = 0; i < n; i++) have = have && ctx[i]->comm && ctx[i]->comm_n == n && ctx[i]->comm_rank == i;
    if (!have) {
        std::vector<ncclComm_t> comms(n); std::vector<int> devs(n);
        for (int i = 0; i < n; i++) { devs[i] = ctx[i]->device; if (ctx[i]->comm) { (void)hipSetDevice(ctx[i]->device); (void)ncclCommDestroy(ctx[i]->comm); ctx[i]->comm = nullptr; } }
        NCCLCHK(ncclCommInitAll(comms.data(), n, devs.data()));
        for (int i = 0; i < n; i++) { ctx[i]->comm = comms[i]; ctx[i]->comm_n = n; ctx[i]->comm_rank = i; }
    }
    NCCLCHK(ncclGroupStart());
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(ctx[i]->device);
        ncclResult_t r = ncclReduce(ctx[i]->d_tally, i == root ? ctx[i]->d_reduced : ctx[i]->d_tally, (size_t)ctx[i]->ntally, ncclDouble, ncclSum, root, ctx[i]->comm, ctx[i]->stream);
        if (r != ncclSuccess) { (void)ncclGroupEnd(); return fail(std::string("ncclReduce: ") + ncclGetErrorString(r)); }
    }
    NCCLCHK(ncclGroupEnd());
    for (int i = 0; i < n; i++) { HIPCHK(hipSetDevice(ctx[i]->device)); HIPCHK(hipStreamSynchronize(ctx[i]->stream)); }
    ctx[root]->reduced_valid = true;
    return 0;
}
// run_simulation over the GPUs of the node from one host thread: context r transports the photons r+1, r+1+n, ... <= nphotons
// (the reference's static decomposition `do ip = rank+1, nphotons, nproc`, run_simulation_mod.f90:150-151); launches are
// asynchronous per device, so the loop enqueues one slice on every GPU and then waits for all of them.
extern "C" int lart_gpu_run_multi(lart_gpu **ctx, int32_t n, int64_t nphotons, uint64_t seed, int32_t max_events) {
    if (!ctx || n < 1 || nphotons < 0) return fail("run_multi: bad arguments");
    if (max_events < 1) max_events = 4096;
    for (int r = 0; r < n; r++) {
        if (!ctx[r]) return fail("run_multi: null context");
        const int64_t cnt = nphotons > r ? (nphotons - 1 - r) / n + 1 : 0;
        if (lart_gpu_begin(ctx[r], r + 1, n, cnt, seed)) return 1;
    }
    for (;;) {
        for (int r = 0; r < n; r++) if (lart_gpu_step(ctx[r], 1, max_events)) return 1;
        int64_t left = 0;
        for (int r = 0; r < n; r++) { int64_t done = 0, active = 0; if (lart_gpu_sync(ctx[r], &done, &active)) return 1; left += active; }
        if (left == 0) break;
    }
    return 0;
}

// par%iseed = 0: "seed from the operating system" (init_random_mt_seed reads /dev/urandom, falls back to clock xor pid,
// random_mt.f90:889-947).  The host calls this once (rank 0) and shares the value, so that all ranks use one Philox key.
extern "C" int lart_host_entropy_seed(uint64_t *seed) {
    if (!seed) return 1;
    uint64_t t = 0;
    FILE *f = std::fopen("/dev/urandom", "rb");
    if (f) { if (std::fread(&t, sizeof t, 1, f) != 1) t = 0; std::fclose(f); }
    if (t == 0) { t = (uint64_t)std::time(nullptr) * 1000003ull ^ (uint64_t)getpid(); }
    t &= 0x1fffffffffffffull; if (t == 0) t = 1;              // 53 bits, non-zero: exact in the hosts' integer(int64) AND in the f64 header keyword ISEED
    *seed = t;
    return 0;
}

extern "C" int lart_gpu_get_photon_log(lart_gpu *c, double *nscatt, double *xfreq_ref, int64_t n) {
    if (!c) return fail("photon_log: null context");
    if (!c->d_log) return fail("photon_log: not enabled (lart_gpu_enable_photon_log before lart_gpu_begin)");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    int64_t m = n < c->nlog ? n : c->nlog;
    if (nscatt) HIPCHK(lart_get(c->stream, nscatt, c->d_log, (size_t)m * 8));
    if (xfreq_ref) HIPCHK(lart_get(c->stream, xfreq_ref, c->d_log + c->nlog, (size_t)m * 8));
    return 0;
}

// the per-event history probe of the parity tests: the photons first_id + k * stride, k < count, run to the last one (as lart_gpu_run
// does, tallies included), and for every k with slot[k] >= 0 the state after the birth and after every event is recorded
// (LART_TRACE_W = 8 doubles per record, at most cap records per slot; n_out = records the packet produced)
extern "C" int lart_gpu_test_trace(lart_gpu *c, int64_t first_id, int64_t stride, int64_t count, uint64_t seed, const int32_t *slot, int32_t nslots, int32_t cap,
                                   double *trace_out, int32_t *n_out) {
    if (!c || !slot || !trace_out || !n_out || count < 1 || nslots < 1 || cap < 1) return fail("test_trace: bad arguments");
    for (int64_t k = 0; k < count; k++) if (slot[k] >= nslots) return fail("test_trace: slot entries must be < nslots");
    HIPCHK(hipSetDevice(c->device));
    const size_t nt = (size_t)nslots * cap * LART_TRACE_W;
    HIPCHK(hipMalloc(&c->d_trace_slot, (size_t)count * 4)); HIPCHK(hipMalloc(&c->d_trace_n, (size_t)nslots * 4)); HIPCHK(hipMalloc(&c->d_trace, nt * 8));
    HIPCHK(lart_put(c->stream, c->d_trace_slot, slot, (size_t)count * 4));
    HIPCHK(hipMemsetAsync(c->d_trace_n, 0, (size_t)nslots * 4, c->stream)); HIPCHK(hipMemsetAsync(c->d_trace, 0, nt * 8, c->stream));
    c->trace_cap = cap;
    int rc = lart_gpu_run(c, first_id, stride, count, seed);
    if (!rc) { if (lart_get(c->stream, trace_out, c->d_trace, nt * 8) != hipSuccess || lart_get(c->stream, n_out, c->d_trace_n, (size_t)nslots * 4) != hipSuccess) rc = fail("test_trace: copy back failed"); }
    (void)hipFree(c->d_trace_slot); (void)hipFree(c->d_trace_n); (void)hipFree(c->d_trace);
    c->d_trace_slot = nullptr; c->d_trace_n = nullptr; c->d_trace = nullptr; c->trace_cap = 0;
    return rc;
}

extern "C" int lart_gpu_get_all_photons(lart_gpu *c, double *out, int64_t n) {
    if (!c || !out) return fail("all_photons: null argument");
    if (!c->d_log || c->log_on != 2) return fail("all_photons: not enabled (lart_gpu_enable_all_photons before lart_gpu_begin)");
    if (n != c->nlog) return fail("all_photons: n must equal the photon count of lart_gpu_begin");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(lart_get(c->stream, out, c->d_log + 2 * c->nlog, (size_t)n * 80));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// device-function probes (parity tests): one lane per element
// ---------------------------------------------------------------------------------------------
__global__ void k_test_voigt(DGrid g, int n, const double *x, const double *a, double *h) {
    extern __shared__ double sh[];
    Lds l = lds_stage(sh, g);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) h[i] = voigt_H(l.vt, x[i], a[i]);
}
// what: 0 uniform, 1 rand_resonance_vz(p1=x,p2=a), 2 rand_resonance(p1=E1), 3 HG(p1=g), 4 rand_voigt(p1=a), 5 gauss (first of pair)
__global__ void k_test_rand(int what, unsigned long long seed, unsigned long long id, int n, double p1, double p2, double *out) {
    if (blockIdx.x * blockDim.x + threadIdx.x != 0) return;
    Rng g; g.init(seed, id, 0);
    for (int i = 0; i < n; i++) {
        double v, s2;
        switch (what) {
        case 0: v = g.next(); break;
        case 1: { Photon pp{}; v = rand_resonance_vz(g, p1, p2 LART_PROF_PASS(pp)); } break;
        case 2: v = rand_resonance(g, p1); break;
        case 3: v = rand_hg(g, p1); break;
        case 4: v = rand_voigt(g, p1); break;
        default: v = g.gauss_pair(s2); break;
        }
        out[i] = v;
    }
}
// mode = 10*kind + variant; kind 0 = to_tau, 1 = to_edge, 2 = to_dist (variant 0)
template <int VAR>
__device__ void test_ray_one(const KArgs &A, const Lds &l, int kind, double *r, int *ic, double tq, double *tau_out) {
    Photon p{}; p.x = r[0]; p.y = r[1]; p.z = r[2]; p.kx = r[3]; p.ky = r[4]; p.kz = r[5]; p.xfreq = r[6]; p.wgt = r[7];
    p.ic = ic[0]; p.jc = ic[1]; p.kc = ic[2]; p.inside = ic[3] != 0;
    if (is_clump(VAR)) { p.icl = ic[0]; p.ic = 1; p.jc = 1; p.kc = 1; }          // clump mode: ic[0] carries the owner clump (0 = vacuum)
    if (VAR == 8 && kind == 0) p.shear = *tau_out;                               // photon%vfy_shear comes in through tau_out and leaves through it
    unsigned ns = 0; bool dust = A.par.DGR > 0.0;
    if (kind == 3) { p.iband = 2; kind = 0; }                                    // the tau march of a band-2 (H-alpha) packet of line type 8
    if (kind == 4) { *tau_out = edge_dust_Ha<VAR>(A, l, dust, p, p.kx, p.ky, p.kz, ns); return; }      // raytrace_to_edge_*_dust_Ha
    if (kind == 0) {
        raytrace_to_tau<VAR>(A, l, dust, p, tq, ns);
        r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = p.kx; r[4] = p.ky; r[5] = p.kz; r[6] = (A.par.line_type == 8 && !p.inside) ? p.xfreq_ref : p.xfreq; r[7] = p.wgt;      // (line type 8 probes: the exit frequency in reference units, 0 for a ray that starts outside)
        ic[0] = is_clump(VAR) ? p.icl : p.ic; ic[1] = p.jc; ic[2] = p.kc; ic[3] = p.inside ? 1 : 0;
        if (VAR == 8) *tau_out = p.shear;
    } else *tau_out = raytrace_to_edge<VAR>(A, l, dust, p, p.xfreq, p.kx, p.ky, p.kz, kind == 2 ? tq : -1.0, ns);
}
__global__ void k_test_raytrace(KArgs A, int mode, int n, double *r8, int *ic4, const double *tq, double *tau_out) {
    extern __shared__ double sh[];
    Lds l = lds_stage(sh, A.g); l.lt = A.par.line_type; l.dnu_hz = A.par.line_type == 7 ? A.par.delta_nu_HD_Hz : A.par.DnuHK_Hz; l.hd_ratio = A.par.ratio_Dfreq_HD; l.hd_ratio_a = A.par.ratio_voigta_HD; l.hd_nD = A.par.nD_over_nH;       // probes: the line type at run time
    if (A.par.line_type >= 4 && A.par.line_type <= 6) { l.lt = 4; lds_set_multiplet(l, A.ml); }
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int kind = mode / 10, var = mode % 10;
    if (var == 0) test_ray_one<0>(A, l, kind, r8 + 8 * i, ic4 + 4 * i, tq ? tq[i] : 0.0, tau_out + i);
    else if (var == 1) test_ray_one<1>(A, l, kind, r8 + 8 * i, ic4 + 4 * i, tq ? tq[i] : 0.0, tau_out + i);
    else if (var == 2) test_ray_one<2>(A, l, kind, r8 + 8 * i, ic4 + 4 * i, tq ? tq[i] : 0.0, tau_out + i);
    else if (var == 3) test_ray_one<3>(A, l, kind, r8 + 8 * i, ic4 + 4 * i, tq ? tq[i] : 0.0, tau_out + i);
    else if (var == 4) test_ray_one<4>(A, l, kind, r8 + 8 * i, ic4 + 4 * i, tq ? tq[i] : 0.0, tau_out + i);
    else if (var == 5) test_ray_one<5>(A, l, kind, r8 + 8 * i, ic4 + 4 * i, tq ? tq[i] : 0.0, tau_out + i);
    else if (var == 8) test_ray_one<8>(A, l, kind, r8 + 8 * i, ic4 + 4 * i, tq ? tq[i] : 0.0, tau_out + i);
    else test_ray_one<6>(A, l, kind, r8 + 8 * i, ic4 + 4 * i, tq ? tq[i] : 0.0, tau_out + i);
}
__global__ void k_test_vec2pix(int nside, int n, const double *v, int *ipix) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ipix[i] = vec2pix(nside, v[3 * i], v[3 * i + 1], v[3 * i + 2]);
}
template <int VAR, bool STOKES, bool DUST>
__device__ void test_scatter_one(const KArgs &A, const Lds &l, double *st, int *ic, unsigned long long seed, unsigned long long id, long long *nd) {
    Photon p{}; p.x = st[0]; p.y = st[1]; p.z = st[2]; p.kx = st[3]; p.ky = st[4]; p.kz = st[5]; p.xfreq = st[6]; p.wgt = st[7];
    p.mx = st[8]; p.my = st[9]; p.mz = st[10]; p.nx = st[11]; p.ny = st[12]; p.nz = st[13]; p.Q = st[15]; p.U = st[16]; p.V = st[17];
    p.ic = ic[0]; p.jc = ic[1]; p.kc = ic[2]; p.inside = true; p.iband = (ic[3] == 2) ? 2 : 1;
    if (is_clump(VAR)) { p.icl = ic[0]; p.ic = 1; p.jc = 1; p.kc = 1; }
    Rng g; g.init(seed, id, 0); unsigned np = 0;
    scatter_event<VAR, STOKES, DUST, false>(A, l, g, p, np, true);
    st[0] = p.x; st[1] = p.y; st[2] = p.z; st[3] = p.kx; st[4] = p.ky; st[5] = p.kz; st[6] = p.xfreq; st[7] = p.wgt;
    st[8] = p.mx; st[9] = p.my; st[10] = p.mz; st[11] = p.nx; st[12] = p.ny; st[13] = p.nz; st[14] = 1.0; st[15] = p.Q; st[16] = p.U; st[17] = p.V;
    ic[3] = p.inside ? (p.iband == 2 ? 2 : 1) : 0; *nd = (long long)g.n;
}
__global__ void k_test_scatter(KArgs A, int n, double *st18, int *ic4, unsigned long long seed, const unsigned long long *ids, long long *nd) {
    extern __shared__ double sh[];
    Lds l = lds_stage(sh, A.g); l.lt = A.par.line_type; l.dnu_hz = A.par.line_type == 7 ? A.par.delta_nu_HD_Hz : A.par.DnuHK_Hz; l.hd_ratio = A.par.ratio_Dfreq_HD; l.hd_ratio_a = A.par.ratio_voigta_HD; l.hd_nD = A.par.nD_over_nH;       // probes: the line type at run time
    if (A.par.line_type >= 4 && A.par.line_type <= 6) { l.lt = 4; lds_set_multiplet(l, A.ml); }
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool stokes = A.par.use_stokes, dust = A.par.DGR > 0.0;
    if (A.cl.N > 0) {
        if (stokes && dust) test_scatter_one<4, true, true>(A, l, st18 + 18 * i, ic4 + 4 * i, seed, ids[i], nd + i);
        else if (stokes) test_scatter_one<4, true, false>(A, l, st18 + 18 * i, ic4 + 4 * i, seed, ids[i], nd + i);
        else if (dust) test_scatter_one<4, false, true>(A, l, st18 + 18 * i, ic4 + 4 * i, seed, ids[i], nd + i);
        else test_scatter_one<4, false, false>(A, l, st18 + 18 * i, ic4 + 4 * i, seed, ids[i], nd + i);
        return;
    }
    if (stokes && dust) test_scatter_one<0, true, true>(A, l, st18 + 18 * i, ic4 + 4 * i, seed, ids[i], nd + i);
    else if (stokes) test_scatter_one<0, true, false>(A, l, st18 + 18 * i, ic4 + 4 * i, seed, ids[i], nd + i);
    else if (dust) test_scatter_one<0, false, true>(A, l, st18 + 18 * i, ic4 + 4 * i, seed, ids[i], nd + i);
    else test_scatter_one<0, false, false>(A, l, st18 + 18 * i, ic4 + 4 * i, seed, ids[i], nd + i);
}

// ONE event round of the photon loop (flight to the next interaction, then the scattering) from a given packet state and draw counter: what one pass of
// transport_kernel's loop does to a packet, with the generator off the ring.  st27 in the layout of the oracle's state record (oracle/lart_oracle.c,
// ORC_STATE_W): x y z kx ky kz xfreq wgt m(3) n(3) Q U V nscatt_gas nscatt_dust draws icell jcell kcell owner_clump vfy_shear iband kind; `first` = the
// packet has not flown yet (forced first scattering).  On return the same record after the event (kind 1 scattered, 2 left or absorbed).
template <int VAR, bool STOKES, bool DUST>
__device__ void test_event_one(const KArgs &A, const Lds &l, double *st, unsigned long long seed, unsigned long long id, int first_in) {
    Photon p{}; p.x = st[0]; p.y = st[1]; p.z = st[2]; p.kx = st[3]; p.ky = st[4]; p.kz = st[5]; p.xfreq = st[6]; p.wgt = st[7];
    p.mx = st[8]; p.my = st[9]; p.mz = st[10]; p.nx = st[11]; p.ny = st[12]; p.nz = st[13]; p.Q = st[14]; p.U = st[15]; p.V = st[16]; p.nsg = st[17]; p.nsd = st[18];
    p.ic = (int)st[20]; p.jc = (int)st[21]; p.kc = (int)st[22]; p.icl = (int)st[23]; p.shear = st[24]; p.iband = st[25] == 2.0 ? 2 : 1; p.inside = true; p.xfreq_ref = 0.0;
    if (is_clump(VAR)) { p.ic = 1; p.jc = 1; p.kc = 1; }
    Rng g; g.init(seed, id, (unsigned long long)st[19]);
    bool first = first_in != 0; unsigned ns = 0, np = 0;
    OvlHold oh; oh.clear(); OvlState os; os.live = false;
    fly<VAR, DUST>(A, l, g, p, first, ns, VAR == 7 ? &oh : nullptr, VAR == 7 ? &os : nullptr);
    if (VAR == 7) while (os.live) {                            // the march of a flight that left its clump, to its end (the kernel advances it a few cells per round)
        double t_end = 0.0; int owner = 0;
        if (ovl_advance<true>(A, l, DUST, p.x, p.y, p.z, p.kx, p.ky, p.kz, p.xfreq, 0.0, ns, t_end, owner, nullptr, os, 1 << 20)) { ovl_finish<true>(A, os, t_end, owner); ovl_apply(A, p, t_end, owner); }
    }
    scatter_event<VAR, STOKES, DUST, false>(A, l, g, p, np, p.inside);
    st[0] = p.x; st[1] = p.y; st[2] = p.z; st[3] = p.kx; st[4] = p.ky; st[5] = p.kz; st[6] = p.inside ? p.xfreq : p.xfreq_ref; st[7] = p.wgt;
    st[8] = p.mx; st[9] = p.my; st[10] = p.mz; st[11] = p.nx; st[12] = p.ny; st[13] = p.nz; st[14] = p.Q; st[15] = p.U; st[16] = p.V; st[17] = p.nsg; st[18] = p.nsd;
    st[19] = (double)g.n; st[20] = is_clump(VAR) ? 1.0 : (double)p.ic; st[21] = (double)p.jc; st[22] = (double)p.kc; st[23] = (double)p.icl; st[24] = p.shear; st[25] = (double)p.iband;
    st[26] = p.inside ? 1.0 : 2.0;
}
template <int VAR>
__device__ void test_event_var(const KArgs &A, const Lds &l, double *st, unsigned long long seed, unsigned long long id, int first) {
    const bool stokes = A.par.use_stokes, dust = A.par.DGR > 0.0;
    if (stokes && dust) test_event_one<VAR, true, true>(A, l, st, seed, id, first);
    else if (stokes) test_event_one<VAR, true, false>(A, l, st, seed, id, first);
    else if (dust) test_event_one<VAR, false, true>(A, l, st, seed, id, first);
    else test_event_one<VAR, false, false>(A, l, st, seed, id, first);
}
__global__ void k_test_event(KArgs A, int var, int n, double *st27, unsigned long long seed, const unsigned long long *ids, const int *first) {
    extern __shared__ double sh[];
    Lds l = lds_stage(sh, A.g); l.lt = 1; lds_set_dipole(l, A.par.E1);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (var == 7) test_event_var<7>(A, l, st27 + 27 * i, seed, ids[i], first[i]);
    else if (var == 4) test_event_var<4>(A, l, st27 + 27 * i, seed, ids[i], first[i]);
    else test_event_var<0>(A, l, st27 + 27 * i, seed, ids[i], first[i]);
}

template <typename T> struct DevBuf {                      // a probe's scratch buffer; copies on the stream the probe's kernel runs on (given to alloc)
    T *p = nullptr; size_t n = 0; hipStream_t st = nullptr;
    int alloc(hipStream_t stream, size_t cnt, const T *src) { st = stream; n = cnt; if (hipMalloc(&p, cnt * sizeof(T) + 8) != hipSuccess) return 1; if (src && lart_put(st, p, src, cnt * sizeof(T)) != hipSuccess) return 1; return 0; }
    int back(T *dst) { return lart_get(st, dst, p, n * sizeof(T)) != hipSuccess; }
    ~DevBuf() { if (p) (void)hipFree(p); }
};
static KArgs probe_args(lart_gpu *c) {
    KArgs A{}; A.par = c->par; if (c->par.h2_nlines > 0) A.par.line_type = 9; A.h2tab = c->d_h2tab; A.g = c->g; A.t = c->t; A.mu = c->mu; A.obs = c->d_obs; A.nobs = 0; A.nxim = c->nxim; A.nyim = c->nyim; A.dxim = c->dxim; A.dyim = c->dyim; A.ctl = c->d_ctl; A.heal_nside = c->heal_nside; A.jpa = c->jpa; A.amr = c->amr; A.cl = c->cl; A.ml = c->ml;
    return A;
}
static size_t probe_shmem(lart_gpu *c) { return (size_t)(VT_END + c->g.nx + c->g.ny + c->g.nz + 3) * sizeof(double); }

extern "C" int lart_gpu_test_voigt(lart_gpu *c, int32_t n, const double *x, const double *a, double *h) {
    if (!c || n < 0) return fail("test_voigt: bad argument");
    HIPCHK(hipSetDevice(c->device));
    DevBuf<double> dx, da, dh; if (dx.alloc(c->stream, n, x) || da.alloc(c->stream, n, a) || dh.alloc(c->stream, n, nullptr)) return fail("test_voigt: hipMalloc failed");
    DGrid g{}; g.nx = g.ny = g.nz = 0; double zero[3] = {0, 0, 0}; DevBuf<double> df; if (df.alloc(c->stream, 3, zero)) return fail("test_voigt: hipMalloc failed"); g.faces = df.p;
    hipLaunchKernelGGL(k_test_voigt, dim3((n + 255) / 256), dim3(256), (VT_END + 3) * sizeof(double), c->stream, g, n, dx.p, da.p, dh.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(c->stream));
    if (dh.back(h)) return fail("test_voigt: copy back failed");
    return 0;
}
extern "C" int lart_gpu_test_rand(lart_gpu *c, int32_t what, uint64_t seed, uint64_t id, int32_t n, double p1, double p2, double *out) {
    if (!c || n < 0) return fail("test_rand: bad argument");
    HIPCHK(hipSetDevice(c->device));
    DevBuf<double> d; if (d.alloc(c->stream, n, nullptr)) return fail("test_rand: hipMalloc failed");
    hipLaunchKernelGGL(k_test_rand, dim3(1), dim3(64), 0, c->stream, what, (unsigned long long)seed, (unsigned long long)id, n, p1, p2, d.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(c->stream));
    if (d.back(out)) return fail("test_rand: copy back failed");
    return 0;
}
extern "C" int lart_gpu_test_raytrace(lart_gpu *c, int32_t mode, int32_t n, double *r8, int32_t *ic4, const double *tq, double *tau_out) {
    if (!c || n < 0 || !c->d_cell) return fail("test_raytrace: needs a context with a medium");
    int kind = mode / 10, var = mode % 10;
    if (kind < 0 || kind > 4 || var < 0 || (var > 6 && var != 8) || (kind == 2 && var != 0 && var != 3) || (kind >= 3 && (c->par.line_type != 8 || (var != 0 && var != 3)))) return fail("test_raytrace: bad mode");
    if ((var == 3) != c->is_amr || (var == 4) != c->is_clump) return fail("test_raytrace: variant 3 needs the octree medium, 4 the clump medium, 0-2, 5, 6 the Cartesian one");
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_tally) { if (build_tally(c)) return 1; }
    // host-side shape check before touching the kernel: every start cell must be a valid gather target
    for (int i = 0; i < n; i++) {
        const int *q = ic4 + 4 * i;
        if (var == 3) { if (q[0] < 1 || q[0] > c->amr.nleaf) return fail("test_raytrace: start leaf must be 1..nleaf"); continue; }
        if (var == 4) { if (q[0] < 0 || q[0] > c->cl.N) return fail("test_raytrace: owner clump must be 0..n"); continue; }
        if (q[0] < 1 || q[0] > c->g.nx + 1 || q[1] < 1 || q[1] > c->g.ny + 1 || q[2] < 1 || q[2] > c->g.nz + 1) return fail("test_raytrace: start cell outside the grid");
        if (q[0] > c->g.nx || q[1] > c->g.ny || q[2] > c->g.nz) return fail("test_raytrace: probe start cells must be interior (1..n)");
    }
    DevBuf<double> dr, dt, dq; DevBuf<int> di;
    if (dr.alloc(c->stream, (size_t)8 * n, r8) || di.alloc(c->stream, (size_t)4 * n, ic4) || dt.alloc(c->stream, n, (var == 8 && tau_out) ? tau_out : nullptr) || dq.alloc(c->stream, n, tq)) return fail("test_raytrace: hipMalloc failed");
    if (c->par.line_type >= 4 && c->par.line_type <= 6 && !c->has_ml) return fail("line_type 4/5/6 needs lart_gpu_set_line_multiplet first");
    KArgs A = probe_args(c);
    hipLaunchKernelGGL(k_test_raytrace, dim3((n + 63) / 64), dim3(64), probe_shmem(c), c->stream, A, mode, n, dr.p, di.p, tq ? dq.p : nullptr, dt.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(c->stream));
    if (dr.back(r8) || di.back(ic4) || (tau_out && dt.back(tau_out))) return fail("test_raytrace: copy back failed");
    return 0;
}
extern "C" int lart_gpu_test_vec2pix(lart_gpu *c, int32_t nside, int32_t n, const double *v3, int32_t *ipix) {
    if (!c || n < 0 || nside < 1) return fail("test_vec2pix: bad argument");
    HIPCHK(hipSetDevice(c->device));
    DevBuf<double> dv; DevBuf<int> di;
    if (dv.alloc(c->stream, (size_t)3 * n, v3) || di.alloc(c->stream, n, nullptr)) return fail("test_vec2pix: hipMalloc failed");
    hipLaunchKernelGGL(k_test_vec2pix, dim3((n + 255) / 256), dim3(256), 0, c->stream, nside, n, dv.p, di.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(c->stream));
    if (di.back(ipix)) return fail("test_vec2pix: copy back failed");
    return 0;
}
extern "C" int lart_gpu_test_scatter(lart_gpu *c, int32_t n, double *st18, int32_t *ic4, uint64_t seed, const uint64_t *ids, int64_t *ndraws) {
    if (!c || n < 0 || !c->d_cell) return fail("test_scatter: needs a context with a medium");
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_tally) { if (build_tally(c)) return 1; }
    for (int i = 0; i < n; i++) { const int *q = ic4 + 4 * i;
        if (c->is_clump) { if (q[0] < 1 || q[0] > c->cl.N) return fail("test_scatter: clump mode scatters inside a clump (owner 1..n)"); continue; }
        if (q[0] < 1 || q[0] > c->g.nx || q[1] < 1 || q[1] > c->g.ny || q[2] < 1 || q[2] > c->g.nz) return fail("test_scatter: cell outside the grid"); }
    if (c->par.use_stokes && c->par.DGR > 0.0 && c->mu.nPDF == 0) return fail("test_scatter: Stokes dust scattering needs a Mueller table");
    DevBuf<double> ds; DevBuf<int> di; DevBuf<unsigned long long> dd; DevBuf<long long> dn;
    if (ds.alloc(c->stream, (size_t)18 * n, st18) || di.alloc(c->stream, (size_t)4 * n, ic4) || dd.alloc(c->stream, n, (const unsigned long long *)ids) || dn.alloc(c->stream, n, nullptr)) return fail("test_scatter: hipMalloc failed");
    if (c->par.line_type >= 4 && c->par.line_type <= 6 && !c->has_ml) return fail("line_type 4/5/6 needs lart_gpu_set_line_multiplet first");
    KArgs A = probe_args(c);
    hipLaunchKernelGGL(k_test_scatter, dim3((n + 63) / 64), dim3(64), probe_shmem(c), c->stream, A, n, ds.p, di.p, (unsigned long long)seed, dd.p, dn.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(c->stream));
    if (ds.back(st18) || di.back(ic4) || dn.back((long long *)ndraws)) return fail("test_scatter: copy back failed");
    return 0;
}

extern "C" int lart_gpu_test_event(lart_gpu *c, int32_t n, double *st27, uint64_t seed, const uint64_t *ids, const int32_t *first) {
    if (!c || n < 0 || !st27 || !ids || !first || !c->d_cell) return fail("test_event: needs a context with a medium and the state, id and first-flight arrays");
    const int var = march_variant(c);
    if (var != 0 && var != 4 && var != 7) return fail("test_event: built for the plain grid and the clump media (march variants 0, 4, 7)");
    if (c->par.line_type != 1 || c->par.h2_nlines > 0) return fail("test_event: single-line profiles (line_type 1) only");
    if (c->par.use_stokes && c->par.DGR > 0.0 && c->mu.nPDF == 0) return fail("test_event: Stokes dust scattering needs a Mueller table");
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_tally) { if (build_tally(c)) return 1; }
    for (int i = 0; i < n; i++) { const double *q = st27 + 27 * (size_t)i;
        if (c->is_clump) { if (q[23] < 0 || q[23] > (double)c->cl.N) return fail("test_event: owner clump out of range"); continue; }
        if (q[20] < 1 || q[20] > c->g.nx || q[21] < 1 || q[21] > c->g.ny || q[22] < 1 || q[22] > c->g.nz) return fail("test_event: cell outside the grid"); }
    DevBuf<double> ds; DevBuf<unsigned long long> dd; DevBuf<int> df;
    if (ds.alloc(c->stream, (size_t)27 * n, st27) || dd.alloc(c->stream, n, (const unsigned long long *)ids) || df.alloc(c->stream, n, first)) return fail("test_event: hipMalloc failed");
    KArgs A = probe_args(c);
    hipLaunchKernelGGL(k_test_event, dim3((n + 63) / 64), dim3(64), probe_shmem(c), c->stream, A, var, n, ds.p, (unsigned long long)seed, dd.p, df.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(c->stream));
    if (ds.back(st27)) return fail("test_event: copy back failed");
    return 0;
}

// make_sightline_tau: one ray per observer pixel from the far side of the medium, marched once per frequency bin
extern "C" int lart_gpu_sightline_tau(lart_gpu *c, double cross0, double *tau_gas, double *N_gas, double *tau_dust) {
    if (!c || !tau_gas || !N_gas) return fail("sightline_tau: null argument");
    if (!c->d_cell && !c->is_clump) return fail("sightline_tau: set the medium first");
    if (c->nobs < 1 || c->h_obs.size() != (size_t)c->nobs) return fail("sightline_tau: set the observers first (lart_gpu_set_observers / _heal)");
    if (!(cross0 > 0.0)) return fail("sightline_tau: cross0 (line%cross0) must be > 0");
    if (c->par.xyz_symmetry || c->par.xy_periodic || c->par.xy_symmetry) return fail("sightline_tau: needs the plain medium (no observers with xyz_symmetry / xy_periodic, setup.f90:198-206)");
    if (c->is_clump && c->heal_nside > 0) return fail("sightline_tau: HEALPix inside observers are not supported with the clump medium (setup.f90:177)");
    if (c->par.nxfreq < 1) return fail("sightline_tau: nxfreq must be >= 1");
    HIPCHK(hipSetDevice(c->device));
    SightGeom g{};
    g.var = c->is_clump ? 4 : (c->is_amr ? 3 : 0);
    g.nside = c->heal_nside; g.nobs = c->nobs; g.nxim = c->nxim; g.nyim = c->nyim; g.dxim = c->dxim; g.dyim = c->dyim; g.obs = c->h_obs.data();
    if (g.var == 0) {
        if (c->h_faces.size() != (size_t)c->g.nx + c->g.ny + c->g.nz + 3) return fail("sightline_tau: no host copy of the cell faces");
        g.nx = c->g.nx; g.ny = c->g.ny; g.nz = c->g.nz; g.xmin = c->g.xmin; g.ymin = c->g.ymin; g.zmin = c->g.zmin; g.dx = c->g.dx; g.dy = c->g.dy; g.dz = c->g.dz;
        g.xface = c->h_faces.data(); g.yface = g.xface + g.nx + 1; g.zface = g.yface + g.ny + 1;
    }
    g.lo = c->amr.lo; g.hi = c->amr.hi; g.sphere_R = c->cl.sphere_R;
    std::vector<SightRay> rays;
    lart_sight_build_rays(g, rays);
    const size_t nray = rays.size(), nxf = (size_t)c->par.nxfreq;
    const bool dust = c->par.DGR > 0.0 && c->has_dust_table;
    DevBuf<SightRay> dr; DevBuf<double> dt, dn, dd;
    if (dr.alloc(c->stream, nray, rays.data()) || dt.alloc(c->stream, nray * nxf, nullptr) || dn.alloc(c->stream, nray, nullptr) || dd.alloc(c->stream, nray, nullptr)) return fail("sightline_tau: hipMalloc failed");
    HIPCHK(hipMemsetAsync(dt.p, 0, nray * nxf * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(dn.p, 0, nray * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(dd.p, 0, nray * sizeof(double), c->stream));
    if (c->par.line_type >= 4 && c->par.line_type <= 6 && !c->has_ml) return fail("line_type 4/5/6 needs lart_gpu_set_line_multiplet first");
    const bool ovl = c->is_clump && c->clump_overlap;                    // setup_clump_overlap: the overlap forms of the two marches (raytrace_clump.f90:922-1001)
    if (ovl && !c->d_tally) { if (build_tally(c)) return 1; }            // the list-overflow counter lives with the scalar tallies
    KArgs A = probe_args(c);
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, c->stream));
    HIPCHK(lart_sight_launch(A, ovl ? 7 : g.var, dr.p, (int)nray, cross0, dust, dt.p, dn.p, dd.p, probe_shmem(c), c->stream));
    HIPCHK(hipEventRecord(e1, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0.0f; (void)hipEventElapsedTime(&ms, e0, e1); c->sight_ms = ms; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (dt.back(tau_gas) || dn.back(N_gas) || (tau_dust && dd.back(tau_dust))) return fail("sightline_tau: copy back failed");
    if (ovl) { double over = 0.0; HIPCHK(lart_get(c->stream, &over, c->d_tally + 7, sizeof over));
        if (over > 0.0) return fail("sightline_tau: a ray met more overlapping clumps than the march lists hold"); }
    return 0;
}
extern "C" int lart_gpu_sightline_time(lart_gpu *c, double *ms) { if (!c || !ms) return fail("sightline_time: null argument"); *ms = c->sight_ms; return 0; }

