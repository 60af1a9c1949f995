// Host-side set-up of the clump population (the part of init_clumps that is a loop over clumps, clump_mod.f90): random
// sequential adsorption placement with the reference's linked-list grid and its MT19937-64 stream, systematic velocity
// fields, and the CSR acceleration grid.  Plain C++ (no device code): this is what grid_create_clump does on the CPU in
// the reference too; the hosts (lart_amd/clump.py, Fortran) call it through the C-ABI and hand the arrays to
// lart_gpu_set_clumps.
#include "../../include/lart_gpu.h"
#include <cmath>
#include <cstdint>
#include <random>
#include <string>
#include <vector>

namespace {
thread_local std::string h_err;
int hfail(const char *m) { h_err = m; return 1; }
// rand_number(): (x >> 12 + 0.5) / 2^52 on the MT19937-64 output (random_mt.f90:579-630); std::mt19937_64 is that generator
struct Mt {
    std::mt19937_64 g; bool stored = false; double gset = 0.0;
    explicit Mt(uint64_t seed) : g(seed) {}
    double rnd() { return ((double)(g() >> 12) + 0.5) * (1.0 / 4503599627370496.0); }
    double gauss() {                                               // rand_gauss with its cached second deviate: random_mt.f90:964-988
        if (stored) { stored = false; return gset; }
        double v1, v2, rsq;
        do { v1 = 2.0 * rnd() - 1.0; v2 = 2.0 * rnd() - 1.0; rsq = v1 * v1 + v2 * v2; } while (!(rsq > 0.0 && rsq < 1.0));
        rsq = std::sqrt(-2.0 * std::log(rsq) / rsq);
        gset = v1 * rsq; stored = true;
        return v2 * rsq;
    }
};
}  // namespace

extern "C" const char *lart_host_last_error(void) { return h_err.c_str(); }

// generate_clumps, uniform path: clump_mod.f90:897-1151
extern "C" int lart_host_clumps_generate(int64_t n, double sphere_R, double r_min, double clump_radius, int32_t flags,
                                         double sigma_v_over_vtherm, int64_t iseed, double *x, double *y, double *z,
                                         double *vx, double *vy, double *vz, double *acceptance) {
    return lart_host_clumps_generate_cone(n, sphere_R, r_min, clump_radius, flags, 0.0, sigma_v_over_vtherm, iseed, x, y, z, vx, vy, vz, acceptance);
}
// ... with par%cone_opening [deg]: centres drawn directly inside the bicone about the z axis (:1029-1048), 0 or >= 90 = the whole shell
extern "C" int lart_host_clumps_generate_cone(int64_t n, double sphere_R, double r_min, double clump_radius, int32_t flags, double cone_opening_deg,
                                              double sigma_v_over_vtherm, int64_t iseed, double *x, double *y, double *z,
                                              double *vx, double *vy, double *vz, double *acceptance) {
    if (n < 1 || !(sphere_R > 0.0) || !(clump_radius > 0.0) || r_min < 0.0 || r_min >= sphere_R || !x || !y || !z || !vx || !vy || !vz)
        return hfail("clumps_generate: bad argument");
    const bool fully_inside = flags & 1, allow_overlap = flags & 2;
    const double cos_cone = (cone_opening_deg > 0.0 && cone_opening_deg < 90.0) ? std::cos(cone_opening_deg * (3.14159265358979323846264338327950288 / 180.0)) : -1.0;      // par%clump_fully_inside, par%clump_allow_overlap (:1090: the overlap test runs, the trial is kept)
    if (fully_inside && (clump_radius >= sphere_R || r_min + 2.0 * clump_radius > sphere_R))
        return hfail("clumps_generate: clump_fully_inside but no clump fits inside the shell (clump_mod.f90:937-951)");
    int rg = (int)std::pow((double)n, 1.0 / 3.0) + 1; if (rg < 32) rg = 32; if (rg > 512) rg = 512;
    double rg_cell = std::fmax(2.0 * sphere_R / rg, 2.0 * clump_radius);
    rg = (int)((2.0 * sphere_R) / rg_cell) + 1; if (rg < 2) rg = 2;
    rg_cell = (2.0 * sphere_R) / rg;
    const double min_sep2 = (2.0 * clump_radius) * (2.0 * clump_radius);
    std::vector<int> head((size_t)rg * rg * rg, -1), nxt((size_t)n, -1);
    const double r_max_c = fully_inside ? sphere_R - clump_radius : sphere_R, r_min_c = fully_inside ? r_min + clump_radius : r_min;
    const double r_max_c2 = r_max_c * r_max_c, r_min_c2 = r_min_c * r_min_c;
    // a layout that cannot be completed (packing too dense) must not spin forever
    const int64_t max_attempts = 2000 * n + 1000000;
    Mt mt((uint64_t)iseed);
    int64_t icl = 0, attempts = 0;
    while (icl < n) {
        if (++attempts > max_attempts) return hfail("clumps_generate: random sequential adsorption did not converge (volume filling too high)");
        double xc, yc, zc, d2;
        if (cos_cone > 0.0) {                                      // direct sampling inside the bicone: clump_mod.f90:1029-1048
            double r_trial, cos_theta, sin_theta;
            for (;;) {
                r_trial = std::pow(r_min_c * r_min_c * r_min_c + (r_max_c * r_max_c * r_max_c - r_min_c * r_min_c * r_min_c) * mt.rnd(), 1.0 / 3.0);
                cos_theta = cos_cone + (1.0 - cos_cone) * mt.rnd();
                if (mt.rnd() < 0.5) cos_theta = -cos_theta;
                sin_theta = std::sqrt(std::fmax(0.0, 1.0 - cos_theta * cos_theta));
                if (fully_inside && r_trial > clump_radius) {
                    const double q = clump_radius / r_trial, cos_theta_min = std::fabs(cos_theta) * std::sqrt(1.0 - q * q) - sin_theta * q;
                    if (cos_theta_min < cos_cone) continue;
                }
                break;
            }
            const double phi = 6.283185307179586476925286766559 * mt.rnd();
            xc = r_trial * sin_theta * std::cos(phi); yc = r_trial * sin_theta * std::sin(phi); zc = r_trial * cos_theta;
        } else
        for (;;) {
            xc = (2.0 * mt.rnd() - 1.0) * r_max_c; yc = (2.0 * mt.rnd() - 1.0) * r_max_c; zc = (2.0 * mt.rnd() - 1.0) * r_max_c;
            d2 = xc * xc + yc * yc + zc * zc;
            if (d2 <= r_max_c2 && d2 >= r_min_c2) break;
        }
        auto cell = [&](double v) { int i = (int)((v + sphere_R) / rg_cell); return i < 0 ? 0 : (i > rg - 1 ? rg - 1 : i); };
        int ig = cell(xc), jg = cell(yc), kg = cell(zc);
        bool overlap = false;
        for (int k2 = std::max(0, kg - 1); k2 <= std::min(rg - 1, kg + 1) && !overlap; k2++)
            for (int j2 = std::max(0, jg - 1); j2 <= std::min(rg - 1, jg + 1) && !overlap; j2++)
                for (int i2 = std::max(0, ig - 1); i2 <= std::min(rg - 1, ig + 1) && !overlap; i2++)
                    for (int jnb = head[(size_t)i2 + (size_t)j2 * rg + (size_t)k2 * rg * rg]; jnb > 0; jnb = nxt[jnb - 1]) {
                        double dx = xc - x[jnb - 1], dy = yc - y[jnb - 1], dz = zc - z[jnb - 1];
                        if (dx * dx + dy * dy + dz * dz < min_sep2) { overlap = true; break; }
                    }
        if (overlap && !allow_overlap) continue;
        icl++;
        x[icl - 1] = xc; y[icl - 1] = yc; z[icl - 1] = zc;
        vx[icl - 1] = 0.0; vy[icl - 1] = 0.0; vz[icl - 1] = 0.0;
        if (sigma_v_over_vtherm > 0.0) { vx[icl - 1] = sigma_v_over_vtherm * mt.gauss(); vy[icl - 1] = sigma_v_over_vtherm * mt.gauss(); vz[icl - 1] = sigma_v_over_vtherm * mt.gauss(); }
        size_t q = (size_t)ig + (size_t)jg * rg + (size_t)kg * rg * rg;
        nxt[icl - 1] = head[q]; head[q] = (int)icl;
    }
    if (acceptance) *acceptance = (double)n / (double)attempts;
    return 0;
}

// build_clump_csr + clump_cell_range: clump_mod.f90:1267-1360.  cg_start needs cgx^3+1 entries (cgx from
// lart_host_clumps_csr_dims); call once with cg_list = NULL to get the registration count, then with cg_list[count].
extern "C" int lart_host_clumps_csr_dims(int64_t n, double sphere_R, double radius_max, int32_t *cgx, double *cg_min, double *cg_d) {
    if (n < 1 || !(sphere_R > 0.0) || !(radius_max > 0.0) || !cgx || !cg_min || !cg_d) return hfail("clumps_csr_dims: bad argument");
    int cg = (int)std::pow((double)n, 1.0 / 3.0) + 1; if (cg < 32) cg = 32; if (cg > 512) cg = 512;
    *cgx = cg; *cg_min = -(sphere_R + radius_max); *cg_d = (2.0 * (sphere_R + radius_max)) / cg;
    return 0;
}
extern "C" int lart_host_clumps_build_csr(int64_t n, const double *x, const double *y, const double *z, const double *radius,
                                          int32_t cgx, double cg_min, double cg_d, int32_t *cg_start, int32_t *cg_list, int64_t *nreg) {
    if (n < 1 || !x || !y || !z || !radius || cgx < 1 || !(cg_d > 0.0) || !cg_start || !nreg) return hfail("clumps_build_csr: bad argument");
    const double inv = 1.0 / cg_d; const size_t ncells = (size_t)cgx * cgx * cgx;
    auto range = [&](int64_t i, int r[6]) {
        double rc = radius[i]; int m = cgx - 1;
        r[0] = std::max(0, (int)((x[i] - cg_min - rc) * inv)); r[1] = std::min(m, (int)((x[i] - cg_min + rc) * inv));
        r[2] = std::max(0, (int)((y[i] - cg_min - rc) * inv)); r[3] = std::min(m, (int)((y[i] - cg_min + rc) * inv));
        r[4] = std::max(0, (int)((z[i] - cg_min - rc) * inv)); r[5] = std::min(m, (int)((z[i] - cg_min + rc) * inv));
    };
    std::vector<int> cnt(ncells, 0); int r[6];
    for (int64_t i = 0; i < n; i++) { range(i, r);
        for (int k = r[4]; k <= r[5]; k++) for (int j = r[2]; j <= r[3]; j++) for (int ii = r[0]; ii <= r[1]; ii++) cnt[(size_t)ii + (size_t)j * cgx + (size_t)k * cgx * cgx]++; }
    int64_t tot = 0; for (size_t q = 0; q < ncells; q++) tot += cnt[q];
    if (tot > 2147483646LL) return hfail("clumps_build_csr: more than 2^31 registrations");
    cg_start[0] = 1;
    for (size_t q = 0; q < ncells; q++) cg_start[q + 1] = cg_start[q] + cnt[q];
    *nreg = tot;
    if (!cg_list) return 0;
    std::fill(cnt.begin(), cnt.end(), 0);
    for (int64_t i = 0; i < n; i++) { range(i, r);
        for (int k = r[4]; k <= r[5]; k++) for (int j = r[2]; j <= r[3]; j++) for (int ii = r[0]; ii <= r[1]; ii++) {
            size_t q = (size_t)ii + (size_t)j * cgx + (size_t)k * cgx * cgx; cg_list[cg_start[q] - 1 + cnt[q]] = (int32_t)(i + 1); cnt[q]++; } }
    return 0;
}

// check_has_overlap: clump_mod.f90:1544-1591 (any pair of clumps closer than the sum of its radii, searched through the CSR cells)
extern "C" int lart_host_clumps_has_overlap(int64_t n, const double *x, const double *y, const double *z, const double *radius,
                                            int32_t cgx, double cg_min, double cg_d, const int32_t *cg_start, const int32_t *cg_list, int32_t *has_overlap) {
    if (n < 1 || !x || !y || !z || !radius || cgx < 1 || !(cg_d > 0.0) || !cg_start || !cg_list || !has_overlap) return hfail("clumps_has_overlap: bad argument");
    const double inv = 1.0 / cg_d; const int m = cgx - 1;
    *has_overlap = 0;
    for (int64_t i = 0; i < n; i++) {
        const double rc = radius[i];
        const int i0 = std::max(0, (int)((x[i] - cg_min - rc) * inv)), i1 = std::min(m, (int)((x[i] - cg_min + rc) * inv));
        const int j0 = std::max(0, (int)((y[i] - cg_min - rc) * inv)), j1 = std::min(m, (int)((y[i] - cg_min + rc) * inv));
        const int k0 = std::max(0, (int)((z[i] - cg_min - rc) * inv)), k1 = std::min(m, (int)((z[i] - cg_min + rc) * inv));
        for (int k = k0; k <= k1; k++) for (int j = j0; j <= j1; j++) for (int ii = i0; ii <= i1; ii++) {
            const size_t q = (size_t)ii + (size_t)j * cgx + (size_t)k * cgx * cgx;
            for (int32_t ip = cg_start[q]; ip < cg_start[q + 1]; ip++) {
                const int64_t jc = cg_list[ip - 1] - 1; if (jc <= i) continue;
                const double dx = x[jc] - x[i], dy = y[jc] - y[i], dz = z[jc] - z[i], rs = radius[i] + radius[jc];
                if (dx * dx + dy * dy + dz * dz < rs * rs) { *has_overlap = 1; return 0; }
            }
        }
    }
    return 0;
}
