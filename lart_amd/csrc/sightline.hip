// sightline.hip -- sight-line maps: host ray set-up + the marching kernel (see sightline.hpp for the reference routines).
//
// Work decomposition: one wavefront = one sight line x 64 consecutive frequency bins.  All 64 lanes walk the same cells
// (the traversal is wave-uniform: no divergence, the Cell / clump records are broadcast loads) and differ only in the
// frequency they evaluate the Voigt profile at, so the march is bound by the H(x,a) evaluations, not by memory.
// The column density and the dust depth do not depend on frequency; the first chunk of every ray carries them along.
#include "sightline.hpp"
#include <cmath>

// ---------------------------------------------------------------------------------------------
// host: pixel -> ray
// ---------------------------------------------------------------------------------------------
static const double kPi = 3.141592653589793238462643383279502884197, kRad2Deg = 180.0 / kPi, kHugest = 1.7976931348623157e308;

// pix2vec (healpix.f90:114-187): centre of RING pixel ipix = 1..12 nside^2
void lart_sight_pix2vec(int nside, int ipix, double v[3]) {
    const int nl2 = 2 * nside, nl4 = 4 * nside, npix = 12 * nside * nside, ncap = 2 * nside * (nside - 1);
    const double fact1 = 1.5 * nside, fact2 = 3.0 * nside * nside;
    double z, phi;
    if (ipix <= ncap) {                                    // north polar cap
        double hip = ipix / 2.0;
        int iring = (int)std::sqrt(std::fmax(0.0, hip - std::sqrt(std::trunc(hip)))) + 1;
        int iphi = ipix - 2 * iring * (iring - 1);
        z = 1.0 - (double)(iring * iring) / fact2;
        phi = ((double)iphi - 0.5) * kPi / (2.0 * iring);
    } else if (ipix <= nl2 * (5 * nside + 1)) {            // equatorial belt
        int ip = ipix - ncap - 1, iring = ip / nl4 + nside, iphi = ip % nl4 + 1;
        double fodd = 0.5 * (1 + (iring + nside) % 2);
        z = (double)(nl2 - iring) / fact1;
        phi = ((double)iphi - fodd) * kPi / (2.0 * nside);
    } else {                                               // south polar cap
        int ip = npix - ipix + 1;
        double hip = ip / 2.0;
        int iring = (int)std::sqrt(std::fmax(0.0, hip - std::sqrt(std::trunc(hip)))) + 1;
        int iphi = 4 * iring + 1 - (ip - 2 * iring * (iring - 1));
        z = -1.0 + (double)(iring * iring) / fact2;
        phi = ((double)iphi - 0.5) * kPi / (2.0 * iring);
    }
    const double sth = std::sqrt((1.0 - z) * (1.0 + z));
    v[0] = sth * std::cos(phi); v[1] = sth * std::sin(phi); v[2] = z;
}

namespace {
struct Box { double lo[3], hi[3]; };

// direction from the observer out through pixel (ix,iy): TAN image (sightline_tau_rect.f90:44-54) or HEALPix pixel centre
inline void pixel_direction(const SightGeom &g, const DObs &o, int ix, int iy, double k[3]) {
    if (g.nside > 0) { lart_sight_pix2vec(g.nside, ix, k); return; }
    double kx = std::tan((ix - (g.nxim + 1.0) / 2.0) * g.dxim / kRad2Deg), ky = std::tan((iy - (g.nyim + 1.0) / 2.0) * g.dyim / kRad2Deg), kz = -1.0;
    const double kr = std::sqrt(kx * kx + ky * ky + kz * kz);
    kx = kx / kr; ky = ky / kr; kz = kz / kr;
    for (int c = 0; c < 3; c++) k[c] = o.rm[c] * kx + o.rm[3 + c] * ky + o.rm[6 + c] * kz;       // transpose of rmatrix
}

// clump medium: far intersection with the bounding sphere, nudged inwards (sightline_tau_clump.f90:78-116)
inline bool start_clump(const SightGeom &g, const DObs &o, const double k[3], SightRay &r) {
    const double R = g.sphere_R, eps = 1.0e-9 * R;
    const double b = o.x * k[0] + o.y * k[1] + o.z * k[2], c = o.x * o.x + o.y * o.y + o.z * o.z - R * R, disc = b * b - c;
    if (disc < 0.0) return false;
    const double t_far = -b + std::sqrt(disc);
    if (t_far <= 0.0) return false;
    r.x = o.x + k[0] * t_far; r.y = o.y + k[1] * t_far; r.z = o.z + k[2] * t_far;
    r.kx = -k[0]; r.ky = -k[1]; r.kz = -k[2];
    r.x = r.x + eps * r.kx; r.y = r.y + eps * r.ky; r.z = r.z + eps * r.kz;
    r.ic = r.jc = r.kc = 0; r.dist = -1.0;
    return true;
}

// box media: the six face distances, then the far face (outside observers) or the near one (inside)
inline bool start_box(const SightGeom &g, const DObs &o, const double k[3], SightRay &r) {
    const bool inside = g.nside > 0, amr = g.var == 3;
    Box bx;
    if (amr) { for (int a = 0; a < 3; a++) { bx.lo[a] = g.lo; bx.hi[a] = g.hi; } }
    else { bx.lo[0] = g.xmin; bx.lo[1] = g.ymin; bx.lo[2] = g.zmin; bx.hi[0] = -g.xmin; bx.hi[1] = -g.ymin; bx.hi[2] = -g.zmin; }   // grid%xmax = -grid%xmin on the plain grid (grid_mod_car.f90:138-171)
    const double ob[3] = {o.x, o.y, o.z}, none = inside ? -999.0 : kHugest;
    double delt[6];
    for (int a = 0; a < 3; a++) {
        if (k[a] == 0.0) { delt[2 * a] = none; delt[2 * a + 1] = none; }
        else { delt[2 * a] = (bx.hi[a] - ob[a]) / k[a]; delt[2 * a + 1] = (bx.lo[a] - ob[a]) / k[a]; }
    }
    const int n[3] = {g.nx, g.ny, g.nz}; const double d[3] = {g.dx, g.dy, g.dz}, mn[3] = {g.xmin, g.ymin, g.zmin};
    int j0 = 0; double dd;
    if (inside) {                                          // sightline_tau_heal.f90:70-79
        dd = kHugest;
        for (int j = 0; j < 6; j++) if (delt[j] > 0.0 && delt[j] < kHugest && delt[j] < dd) { dd = delt[j]; j0 = j + 1; }
        if (!(dd > 0.0 && dd < kHugest)) return false;
    } else {
        dd = -999.9;
        const double eps = amr ? 1.0e-9 * (g.hi - g.lo) : 0.0;
        for (int j = 0; j < 6; j++) {
            if (!(delt[j] > 0.0 && delt[j] < kHugest)) continue;
            const double p[3] = {o.x + k[0] * delt[j], o.y + k[1] * delt[j], o.z + k[2] * delt[j]};
            bool on_box = true;
            if (amr) {                                     // sightline_tau_rect.f90:259-277
                for (int a = 0; a < 3; a++) on_box = on_box && p[a] >= bx.lo[a] - eps && p[a] <= bx.hi[a] + eps;
            } else {                                       // :79-107 (the face's own axis is pinned to its index)
                for (int a = 0; a < 3; a++) {
                    int cell = (int)std::floor((p[a] - mn[a]) / d[a]) + 1;
                    if (j == 2 * a) cell = n[a] + 1; else if (j == 2 * a + 1) cell = 1;
                    on_box = on_box && cell >= 1 && cell <= n[a] + 1;
                }
            }
            if (on_box && delt[j] > dd) { dd = delt[j]; j0 = j + 1; }
        }
        if (!(dd > 0.0 && dd < kHugest) || j0 == 0) return false;
    }
    double p[3] = {o.x + k[0] * dd, o.y + k[1] * dd, o.z + k[2] * dd};
    r.kx = -k[0]; r.ky = -k[1]; r.kz = -k[2];
    r.dist = inside ? dd : -1.0;
    const int ax = (j0 - 1) / 2; const bool hi_face = (j0 - 1) % 2 == 0;
    if (amr) {
        if (inside) p[ax] = hi_face ? bx.hi[ax] : bx.lo[ax];                                       // sightline_tau_heal.f90:236-249
        else {                                                                                     // sightline_tau_rect.f90:283-297
            const double eps = 1.0e-9 * (g.hi - g.lo), kk[3] = {r.kx, r.ky, r.kz};
            for (int a = 0; a < 3; a++) { p[a] = std::fmax(bx.lo[a], std::fmin(bx.hi[a], p[a])); p[a] = p[a] + eps * kk[a]; }
        }
        r.x = p[0]; r.y = p[1]; r.z = p[2]; r.ic = r.jc = r.kc = 0;
        return true;
    }
    int cell[3];
    for (int a = 0; a < 3; a++) cell[a] = (int)std::floor((p[a] - mn[a]) / d[a]) + 1;
    const double *face[3] = {g.xface, g.yface, g.zface};
    cell[ax] = hi_face ? n[ax] + 1 : 1; p[ax] = face[ax][hi_face ? n[ax] : 0];                     // :117-138
    const double kk[3] = {r.kx, r.ky, r.kz};
    for (int a = 0; a < 3; a++) if (cell[a] == n[a] + 1 && kk[a] < 0.0) cell[a] = n[a];
    for (int a = 0; a < 3; a++) if (cell[a] < 1 || cell[a] > n[a]) return false;                   // the medium tables end at n
    r.x = p[0]; r.y = p[1]; r.z = p[2]; r.ic = cell[0]; r.jc = cell[1]; r.kc = cell[2];
    return true;
}
}  // namespace

void lart_sight_build_rays(const SightGeom &g, std::vector<SightRay> &rays) {
    const size_t n2 = (size_t)g.nxim * g.nyim;
    rays.assign(n2 * g.nobs, SightRay{});
    for (int io = 0; io < g.nobs; io++)
        for (int iy = 1; iy <= g.nyim; iy++)
            for (int ix = 1; ix <= g.nxim; ix++) {
                SightRay &r = rays[(size_t)io * n2 + (size_t)(ix - 1) + (size_t)g.nxim * (iy - 1)];
                double k[3];
                pixel_direction(g, g.obs[io], ix, iy, k);
                r.ok = (g.var == 4 ? start_clump(g, g.obs[io], k, r) : start_box(g, g.obs[io], k, r)) ? 1 : 0;
            }
}

// ---------------------------------------------------------------------------------------------
// device: the march
// ---------------------------------------------------------------------------------------------
struct SightAcc { double tau, N, td; };

// raytrace_to_edge_car_tau_gas + _column (raytrace_car.f90:1236-1423); with dist: raytrace_to_dist_tau_gas_car + _column_car (:197-408)
__device__ inline void sight_car(const KArgs &A, const Lds &l, const SightRay &r, double xf0, double cross0, bool dust, SightAcc &s) {
    const DGrid &g = A.g;
    const Cell *c = &g.cell[cidx(g, r.ic, r.jc, r.kc)];
    // the local frame of the start cell (sightline_tau_rect.f90:148-153)
    double xfreq = xf0 * A.par.Dfreq_ref / c->Dfreq - (c->vfx * r.kx + c->vfy * r.ky + c->vfz * r.kz);
    Trav t; t.kx = r.kx; t.ky = r.ky; t.kz = r.kz; t.ic = r.ic; t.jc = r.jc; t.kc = r.kc;
    double xs = r.x, ys = r.y;
    if (trav_setup<0>(g, l, xs, ys, r.z, false, t)) return;
    c = &g.cell[cidx(g, t.ic, t.jc, t.kc)];
    double d = 0.0, Dold = c->Dfreq, u1 = c->vfx * t.kx + c->vfy * t.ky + c->vfz * t.kz;
    for (;;) {
        const double rhokap = c->rhokap * calc_voigt(l, xfreq, c->voigt_a, c->Dfreq);
        const int idx = minloc3(t.tx, t.ty, t.tz);
        const double tt = idx == 1 ? t.tx : (idx == 2 ? t.ty : t.tz);
        const bool last = r.dist >= 0.0 && tt >= r.dist;
        const double del = last ? r.dist - d : tt - d;
        s.tau = s.tau + del * rhokap;
        s.N = s.N + del * (c->rhokap * c->Dfreq / cross0);
        if (dust) s.td = s.td + del * c->rhokapD;
        if (last) break;
        d = tt;
        if (!trav_cross<0>(g, t, idx)) break;
        c = &g.cell[cidx(g, t.ic, t.jc, t.kc)];
        const double Dn = c->Dfreq, u2 = c->vfx * t.kx + c->vfy * t.ky + c->vfz * t.kz;
        xfreq = (xfreq + u1) * Dold / Dn - u2;
        Dold = Dn; u1 = u2;
    }
}

// raytrace_to_{edge,dist}_tau_gas_amr + _column_amr (raytrace_amr.f90:395-554, :650-823)
__device__ inline void sight_amr(const KArgs &A, const Lds &l, const SightRay &r, double xf0, double cross0, bool dust, SightAcc &s) {
    const DAmr &a = A.amr;
    double x = r.x, y = r.y, z = r.z, kx = r.kx, ky = r.ky, kz = r.kz, d = 0.0;
    int il = amr_find_leaf(a, x, y, z);
    if (il <= 0) return;
    const Cell *c = &A.g.cell[il - 1];
    double u1 = c->vfx * kx + c->vfy * ky + c->vfz * kz;
    double xfreq = xf0 * A.par.Dfreq_ref / c->Dfreq - u1;
    for (;;) {
        int ic = a.icell_of_leaf[il - 1], iface;
        const double t_exit = amr_cell_exit(a.geom[ic - 1], x, y, z, kx, ky, kz, iface);
        const bool last = r.dist >= 0.0 && d + t_exit >= r.dist;
        const double del = last ? r.dist - d : t_exit;
        s.tau = s.tau + del * (c->rhokap * calc_voigt(l, xfreq, c->voigt_a, c->Dfreq));
        s.N = s.N + del * c->rhokap * c->Dfreq / cross0;
        if (dust) s.td = s.td + del * c->rhokapD;
        if (last) return;
        d = d + t_exit; x = x + t_exit * kx; y = y + t_exit * ky; z = z + t_exit * kz;
        int il_new = amr_next_leaf(a, ic, iface, x, y, z); bool reflected = false;
        if (il_new <= 0) { il_new = amr_boundary(A, il, iface, x, y, z, kx, ky, kz, reflected); if (d > LART_AMR_MAX_WRAPS * (a.hi - a.lo)) il_new = 0; }
        if (il_new <= 0) return;
        const Cell *cn = &A.g.cell[il_new - 1];
        const double u2 = cn->vfx * kx + cn->vfy * ky + cn->vfz * kz;
        xfreq = reflected ? (xfreq + u1) - u2 : (xfreq + u1) * c->Dfreq / cn->Dfreq - u2;
        u1 = u2; il = il_new; c = cn;
    }
}

// raytrace_to_edge_tau_gas_clump (= raytrace_to_edge_clump with clump_gas_only, raytrace_clump.f90:274-285, :196-270)
// + raytrace_to_edge_column_clump (:289-334).  The ray starts in vacuum on the bounding sphere.
__device__ inline void sight_clump(const KArgs &A, const Lds &l, const SightRay &r, double xf, double cross0, bool dust, SightAcc &s) {
    const DClumps &c = A.cl;
    double xp = r.x, yp = r.y, zp = r.z; const double kx = r.kx, ky = r.ky, kz = r.kz;
    long long icl_cur = 0, icl_found; unsigned nsteps = 0;
    for (;;) {
        const double t_sp = sphere_exit_dist(c, xp, yp, zp, kx, ky, kz);
        double te, tx2;
        if (t_sp <= 0.0) break;
        if (!find_next_clump(c, xp, yp, zp, kx, ky, kz, icl_cur, t_sp, te, tx2, icl_found, nsteps)) break;
        te = fmax(0.0, te);
        xp = xp + te * kx; yp = yp + te * ky; zp = zp + te * kz;
        const ClumpPhys ph = c.phys[icl_found - 1];
        const double u_los = ulos_clump(A, ph, kx, ky, kz);
        xf = xf - u_los;
        const double t_seg = clump_exit_dist(c.geo[icl_found - 1], xp, yp, zp, kx, ky, kz);
        s.tau = s.tau + ph.rhokap * voigt_clump(A, l, xf, ph) * t_seg;
        s.N = s.N + (ph.rhokap / cross0) * ph.Dfreq * t_seg;
        if (dust) s.td = s.td + ph.rhokapD * t_seg;
        xp = xp + t_seg * kx; yp = yp + t_seg * ky; zp = zp + t_seg * kz;
        xf = xf + u_los;
        icl_cur = icl_found;
        if (xp * xp + yp * yp + zp * zp >= c.sphere_R * c.sphere_R) break;
    }
}

template <int VAR>
__global__ __launch_bounds__(256) void k_sightline(const KArgs A, const SightRay *__restrict__ rays, int nray, int nchunk, double cross0, int dust,
                                                   double *__restrict__ tau_gas, double *__restrict__ N_gas, double *__restrict__ tau_dust) {
    extern __shared__ double sh[];
    Lds l_ = lds_stage(sh, A.g); l_.lt = A.par.line_type; l_.dnu_hz = A.par.line_type == 7 ? A.par.delta_nu_HD_Hz : A.par.DnuHK_Hz; l_.hd_ratio = A.par.ratio_Dfreq_HD; l_.hd_ratio_a = A.par.ratio_voigta_HD; l_.hd_nD = A.par.nD_over_nH;
    if (A.par.line_type >= 4 && A.par.line_type <= 6) { l_.lt = 4; lds_set_multiplet(l_, A.ml); }
    const Lds l = l_;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= (long long)nray * nchunk) return;
    const int ir = (int)(wave / nchunk), chunk = (int)(wave % nchunk), kf = chunk * 64 + (threadIdx.x & 63);
    const SightRay r = rays[ir];
    if (!r.ok) return;                                                    // the pixel sees no medium: the maps keep their zeros
    const double xf0 = ((double)(kf + 1) - 0.5) * A.par.dxfreq + A.par.xfreq_min;      // grid%xfreq(kf+1), grid_mod_car.f90:1506
    SightAcc s{0.0, 0.0, 0.0};
    if (VAR == 0) sight_car(A, l, r, xf0, cross0, dust != 0, s);
    else if (VAR == 3) sight_amr(A, l, r, xf0, cross0, dust != 0, s);
    else if (VAR == 7) {                                                  // raytrace_to_edge_tau_gas_clump_overlap + _column_clump_overlap (raytrace_clump.f90:922-1001): gas only, no cap
        double col[3] = {0.0, 0.0, cross0}, te_; int ow_; unsigned ns_ = 0;
        s.tau = ovl_march<false>(A, l, false, r.x, r.y, r.z, r.kx, r.ky, r.kz, xf0, LART_HUGEST, 0.0, ns_, te_, ow_, col);
        s.N = col[0]; if (dust) s.td = col[1];
    }
    else sight_clump(A, l, r, xf0, cross0, dust != 0, s);
    if (kf < A.par.nxfreq) tau_gas[(size_t)ir * A.par.nxfreq + kf] = s.tau;
    if (kf == 0) { N_gas[ir] = s.N; if (tau_dust) tau_dust[ir] = s.td; }
}

hipError_t lart_sight_launch(const KArgs &A, int var, const SightRay *d_rays, int nray, double cross0, bool dust,
                             double *d_tau_gas, double *d_N_gas, double *d_tau_dust, size_t shmem, hipStream_t stream) {
    if (nray <= 0) return hipSuccess;
    const int nchunk = (A.par.nxfreq + 63) / 64;
    const long long nwave = (long long)nray * nchunk;
    const dim3 block(256), grid((unsigned)((nwave + 3) / 4));
    if (var == 0) hipLaunchKernelGGL(k_sightline<0>, grid, block, shmem, stream, A, d_rays, nray, nchunk, cross0, dust ? 1 : 0, d_tau_gas, d_N_gas, d_tau_dust);
    else if (var == 3) hipLaunchKernelGGL(k_sightline<3>, grid, block, shmem, stream, A, d_rays, nray, nchunk, cross0, dust ? 1 : 0, d_tau_gas, d_N_gas, d_tau_dust);
    else if (var == 7) hipLaunchKernelGGL(k_sightline<7>, grid, block, shmem, stream, A, d_rays, nray, nchunk, cross0, dust ? 1 : 0, d_tau_gas, d_N_gas, d_tau_dust);
    else hipLaunchKernelGGL(k_sightline<4>, grid, block, shmem, stream, A, d_rays, nray, nchunk, cross0, dust ? 1 : 0, d_tau_gas, d_N_gas, d_tau_dust);
    return hipGetLastError();
}
