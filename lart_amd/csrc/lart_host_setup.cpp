// Host-side set-up arithmetic shared by the two hosts of this path (lart_amd/host.py through ctypes, lart_amd/fhost/lart_host.f90 through
// ISO_C_BINDING): what the reference does on the CPU before and after the photon loop for the media this path supports.  Plain C++, no
// device code.  One implementation instead of two transcriptions; written array-wise around the facts that make
// these media simple -- one temperature, hence ONE Doppler width and Voigt parameter for the whole grid, and a density that is a mask
// (inside the shell / bicone or not) times a constant -- not as the reference's nested loops.
//
//   lart_host_line_setup     setup_resonance_line, the single lines / doublets / ly_alpha_HD / ly_beta branches   line_mod.f90:551-1272
//   lart_host_h2_init        h2_init                                                                              h2_mod.f90:120-296
//   lart_host_input_rules    read_input, the rules between the namelist fields                                    setup.f90:160-500
//   lart_host_car_medium     grid_create, analytic-medium subset (axes, mask, opacity normalisation, velocity)    grid_mod_car.f90:73-946
//   lart_host_freq_grid      car_setup_freq_grid + the core-skip scale                                            grid_mod_car.f90:1150-1548
//   lart_host_jpa_car        create_JPa_mem                                                                       grid_mod_car.f90:1242-1434
//   lart_host_observers      observer_create_outside / _inside                                                    observer_rect.f90:24-326, observer_heal.f90:24-80
//   lart_host_clump_scalars  init_clumps (population size and opacity) + compute_clump_scalars                    clump_mod.f90:646-760, 2316-2378
//   lart_host_alias_setup    random_alias_setup                                                                   random_mt.f90:2016-2067
//   lart_host_norm_scales    the divisors of output_normalize_outside                                             output_sum_rect.f90:151-487
#include "../../include/lart_gpu.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
thread_local std::string s_err;
int sfail(const std::string &m) { s_err = m; return 1; }
constexpr double PI = 3.141592653589793238462643383279502884197, FOURPI = 4.0 * PI, TWOPI = 2.0 * PI;
constexpr double UM2KM = 1.0e-9, UM2M = 1.0e-6, H_PLANCK = 6.62607004e-34, SPEEDC = 2.99792458e5, AMU_G = 1.67262192e-24;
constexpr double DEG2RAD = PI / 180.0, RAD2DEG = 180.0 / PI;
inline bool fin(double v) { return std::isfinite(v); }
// H(0, a) from the first terms of the series on the x = 0 node of the table (voigt_mod.f90:691-700 with v = 0)
inline double voigt0(double a) { return 1.0 + a * (-1.1283791671 + a * 1.0); }

struct MetalLine { const char *id; const char *ion; int type; double wl, f1, damping, mass, mass_recoil, k_hi, k_lo; };
// par%line_id -> line type, wavelength0 [um], f12(1), damping, mass of vtherm1, mass of g_recoil0, the two level wavenumbers [1/cm] of DnuHK_Hz.
// As in the reference, CIV_1548 recoils with line%mass_amu's default 1.00797 (define.f90:646, line_mod.f90:605) and AlII_1671's mass is a single
// precision literal (:706).
const MetalLine METALS[] = {
    {"CIV_1548", "C IV", 2, 0.1548187, 0.190, 2.647e8, 12.011, 1.00797, 64591.7, 64484.0},
    {"NV_1239", "N V", 2, 0.1238821, 0.156, 3.390e8, 14.0067, 14.0067, 80721.9, 80463.2},
    {"OVI_1032", "O VI", 2, 0.1031912, 0.133, 4.137e8, 15.9994, 15.9994, 96907.5, 96375.0},
    {"NaI_D", "Na I", 2, 0.5891583253, 0.641, 6.153e7, 22.98977, 22.98977, 16973.36619, 16956.17025},
    {"CaII_HK", "Ca II", 2, 0.3934777, 0.682, 1.446667e8, 40.078, 40.078, 25414.40, 25191.51},
    {"MgII_2796", "Mg II", 2, 0.2796352, 0.608, 2.590e8, 24.305, 24.305, 35760.88, 35669.31},
    {"SiIV_1394", "Si IV", 2, 0.1393755, 0.513, 8.743e8, 28.0855, 28.0855, 71748.64, 71287.54},
    {"AlII_1671", "Al II", 1, 0.16707874, 1.77, 1.41e9, (double)26.98154f, (double)26.98154f, 0.0, 0.0},
};
}  // namespace

extern "C" const char *lart_host_setup_last_error(void) { return s_err.c_str(); }

// ---------------------------------------------------------------------------------------------------------------------------------
// lines
// ---------------------------------------------------------------------------------------------------------------------------------
// returns 0 and fills *L; 2 = the id is one of the resonance + fluorescence multiplets (lart_host_line_lookup has them); 1 = unknown id
extern "C" int lart_host_line_setup(const char *line_id, int32_t fine_structure, int32_t include_deuterium, double D_to_H_ratio, lart_host_line *L) {
    if (!line_id || !L) return sfail("line_setup: null argument");
    std::memset(L, 0, sizeof *L);
    std::string id(line_id);
    // Ly-alpha (line_mod.f90:1241-1270) is the default of every field
    L->line_type = 1; L->wavelength0 = 0.1215668237310; L->damping = 6.2649e8;
    L->cross0 = 0.026540083434 / std::sqrt(PI) * (0.27760 + 0.13881 + 0.0);
    L->mass_amu = 1.00797; L->vtherm1 = 0.12895319011972164 / std::sqrt(1.00797);
    L->g_recoil0 = (H_PLANCK / AMU_G / 1.00797) / ((0.1215668237310 * UM2M) * (0.1215668237310 * UM2M));
    L->E1 = 1.0; L->E2 = 0.0; L->E3 = 1.0; L->ratio_Dfreq_HD = 1.0; L->ratio_voigta_HD = 1.0;
    L->E_conv[0] = 1.0; L->E_conv[1] = 0.0; L->E_conv[2] = 1.0; L->wavelength0_Ha = 0.6564553;
    L->fine_structure = fine_structure != 0; L->include_deuterium = include_deuterium != 0;
    std::snprintf(L->line_id, sizeof L->line_id, "%s", line_id); std::snprintf(L->ion_id, sizeof L->ion_id, "H  I");
    const double speedc_cm = SPEEDC * 1.0e5;
    bool lya = id == "ly_alpha" || id == "ly_alpha_HD";
    if (!lya && id != "ly_beta") {
        const MetalLine *m = nullptr;
        for (const MetalLine &q : METALS) if (id == q.id) m = &q;
        if (!m) { lart_line_multiplet ml; return lart_host_line_lookup(line_id, &ml) == 0 ? 2 : sfail("par%line_id = '" + id + "' is not supported by this path"); }
        L->line_type = m->type; L->wavelength0 = m->wl; L->damping = m->damping; std::snprintf(L->ion_id, sizeof L->ion_id, "%s", m->ion);
        L->cross0 = 0.026540083434 / std::sqrt(PI) * m->f1;
        L->vtherm1 = 0.12895319011972164 / std::sqrt(m->mass);          // (line%mass_amu keeps its default, as in the reference: define.f90:646)
        L->g_recoil0 = (H_PLANCK / AMU_G / m->mass_recoil) / ((m->wl * UM2M) * (m->wl * UM2M));
        L->DnuHK_Hz = m->type == 2 ? speedc_cm * (m->k_hi - m->k_lo) : 0.0;
        L->fine_structure = 0; L->include_deuterium = 0;         // (both only act on the ly_alpha branches)
        return 0;
    }
    if (id == "ly_beta") {                                        // line_mod.f90:1192-1240: H I Lyman beta with the 3p -> 2s conversion to H-alpha
        L->line_type = 8; L->wavelength0 = 0.10257222; L->damping = 1.8970e8;
        L->cross0 = 0.026540083434 / std::sqrt(PI) * 0.07910;
        L->g_recoil0 = (H_PLANCK / AMU_G / 1.00797) / ((L->wavelength0 * UM2M) * (L->wavelength0 * UM2M));
        const double A31 = 1.6725e8, A32 = 2.2448e7;
        L->P_conv = A32 / (A31 + A32);
        if (fine_structure) return sfail("ly_beta: fine structure not yet supported");      // setup.f90:259-263
        L->include_deuterium = 0;
        return 0;
    }
    if (include_deuterium && id == "ly_alpha") std::snprintf(L->line_id, sizeof L->line_id, "ly_alpha_HD");      // line_mod.f90:580-588
    if (std::string(L->line_id) == "ly_alpha_HD") {               // :1155-1191
        L->fine_structure = 0;
        const double wl_D = 0.1215337431, mass_D = 2.01410177812, damping_D = 6.2649e8;
        L->line_type = 7; L->nD_over_nH = D_to_H_ratio; std::snprintf(L->ion_id, sizeof L->ion_id, "H+D");
        L->delta_nu_HD_Hz = speedc_cm * (1.0 / (wl_D * UM2M * 1.0e2) - 1.0 / (L->wavelength0 * UM2M * 1.0e2));
        L->ratio_Dfreq_HD = (wl_D / L->wavelength0) * std::sqrt(mass_D / 1.00797);
        L->ratio_voigta_HD = (damping_D / L->damping) * L->ratio_Dfreq_HD;
        L->g_recoil0_D = (H_PLANCK / AMU_G / mass_D) / ((wl_D * UM2M) * (wl_D * UM2M));
    }
    if (L->fine_structure) { L->line_type = 2; L->DnuHK_Hz = 82259.2850014 * speedc_cm - 82258.9191133 * speedc_cm; }      // :1246-1253
    return 0;
}

// define.f90:928-933: thermal width and par%bturb in quadrature
extern "C" double lart_host_vtherm_total(double vtherm1, double bturb, double T) {
    double vt = vtherm1 * std::sqrt(T);
    if (bturb > 0.0) vt = std::sqrt(vt * vt + bturb * bturb);
    return vt;
}

// h2_init: the two B-X lines next to Ly-alpha, R(6) and P(5) of v = 1 <- 2 (Neufeld 1990); LTE populations of the lower levels at T_H2 from the
// X-state level table (rows V, J, E[1/cm] of <h2_data_dir>/energy_X.dat, read by the caller or here from `path`)
extern "C" int lart_host_h2_init(const char *energy_X_path, double wavelength0_um, double vtherm1, double mass_amu, double bturb, double T_H2, double f_H2,
                                 int32_t pure_absorption, lart_host_h2 *H) {
    if (!energy_X_path || !H) return sfail("h2_init: null argument");
    std::memset(H, 0, sizeof *H);
    FILE *f = std::fopen(energy_X_path, "r");
    if (!f) return sfail(std::string("h2_mod: cannot find ") + energy_X_path);
    std::vector<double> V, J, E; char buf[512];
    while (std::fgets(buf, sizeof buf, f)) {                      // read_energy_X (h2_mod.f90:250-296): a magic-number line with '//', '#' comments, then rows
        std::string q(buf); size_t a = q.find_first_not_of(" \t\r\n");
        if (a == std::string::npos || q[a] == '#' || q.find("//") != std::string::npos) continue;
        double v, j, e;
        if (std::sscanf(q.c_str(), "%lf %lf %lf", &v, &j, &e) == 3) { V.push_back(v); J.push_back(j); E.push_back(e); }
    }
    std::fclose(f);
    const double f_osc_Lya = 0.4162, hc_over_k = 1.4387769, f_osc_const = 1.4992e-16;
    const double nu_Lya = (SPEEDC * 1.0e5) / (wavelength0_um * UM2M * 1.0e2);
    const double vth1 = vtherm1 * std::sqrt(mass_amu / (2.0 * mass_amu)), bt2 = bturb > 0.0 ? bturb * bturb : 0.0;
    const double vth = std::sqrt((vth1 * std::sqrt(T_H2)) * (vth1 * std::sqrt(T_H2)) + bt2);
    H->Dfreq_Hz = nu_Lya * vth / SPEEDC; H->n = 2;
    const double dv[2] = {14.140, 99.229}, lam[2] = {1215.72534, 1216.07038}, A_ul[2] = {1.36e8, 1.59e8}, A_tot[2] = {1.6825e9, 1.7199e9};
    const int vl[2] = {2, 2}, Jl[2] = {6, 5}, Ju[2] = {7, 4};
    double Z = 0.0;
    for (size_t i = 0; i < V.size(); i++) { const double gns = ((long)std::llround(J[i]) % 2 != 0) ? 3.0 : 1.0; Z = Z + gns * (2.0 * J[i] + 1.0) * std::exp(-hc_over_k * E[i] / T_H2); }
    for (int i = 0; i < 2; i++) {
        double el = 0.0; bool found = false;
        for (size_t q = 0; q < V.size() && !found; q++) if ((int)std::llround(V[q]) == vl[i] && (int)std::llround(J[q]) == Jl[i]) { el = E[q]; found = true; }
        if (!found) { char m[128]; std::snprintf(m, sizeof m, "h2_mod: X-level (v=%d, J=%d) not found in energy_X.dat", vl[i], Jl[i]); return sfail(m); }
        const double f_osc = f_osc_const * (2.0 * Ju[i] + 1.0) / (2.0 * Jl[i] + 1.0) * (lam[i] * lam[i]) * A_ul[i];
        const double gns = Jl[i] % 2 != 0 ? 3.0 : 1.0, pop = gns * (2.0 * Jl[i] + 1.0) * std::exp(-hc_over_k * el / T_H2) / Z;
        H->dv_kms[i] = dv[i]; H->dnu_Hz[i] = -(dv[i] / SPEEDC) * nu_Lya; H->f_osc[i] = f_osc; H->pop[i] = pop;
        H->strength[i] = f_H2 * pop * (f_osc / f_osc_Lya); H->a_damp[i] = A_tot[i] / (4.0 * PI * H->Dfreq_Hz);
        H->p_scat[i] = pure_absorption ? 0.0 : A_ul[i] / A_tot[i];
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// read_input: what follows from the namelist before anything is built (setup.f90:160-500) -- which switches exclude each other, which optical-depth /
// column input stands for which, the box of a sphere, the angular bins, the profile geometry, the length unit.  In and out: the record; `notes`
// has a bit for every message the reference prints on the way (LART_NOTE_*).
// ---------------------------------------------------------------------------------------------------------------------------------
extern "C" int lart_host_input_rules(lart_host_rules *r) {
    if (!r) return sfail("input_rules: null argument");
    r->notes = 0;
    if (r->nx == 1 || r->ny == 1 || r->nz == 1) r->xyz_symmetry = 0;
    if (r->nside > 0 && r->use_clump_medium) return sfail("ERROR: nside>0 (HEALPIX inside observer) is not supported with clump mode");
    if (r->use_clump_medium && r->use_amr_grid) return sfail("par%use_clump_medium and par%use_amr_grid are mutually exclusive");
    if (r->nside > 0) { r->use_stokes = 0; r->save_peeloff = 1; r->xyz_symmetry = 0; r->nxim = 0; r->nyim = 0; }      // an observer inside the medium sees a HEALPix sphere
    if (!(r->save_peeloff_2D || r->save_peeloff_3D)) r->save_peeloff = 0;
    if (r->nxim > 0 && r->nyim > 0) r->save_peeloff = 1;
    if (r->save_peeloff && r->xyz_symmetry) { r->save_peeloff = 0; r->notes |= LART_NOTE_NO_PEEL_WITH_XYZSYM; }
    if (!r->save_peeloff) { r->save_peeloff_2D = 0; r->save_peeloff_3D = 0; }
    // one optical depth and one column density, however they were named in the input
    if (r->tau0 > 0.0 && r->taumax < 0.0) r->taumax = r->tau0;
    if (r->N_HI > 0.0 && r->N_HImax < 0.0) r->N_HImax = r->N_HI;
    if (r->N_HImax > 0.0 && r->N_gasmax < 0.0) r->N_gasmax = r->N_HImax;
    if (r->N_HIhomo > 0.0 && r->N_gashomo < 0.0) r->N_gashomo = r->N_HIhomo;
    if (r->cext_dust <= 0.0) r->DGR = 0.0;
    if (r->DGR == 0.0) r->save_Jabs = 0;
    if (r->core_skip_global) r->core_skip = 1;
    if (r->save_Jmu) {
        if (r->nmu < 1) return sfail("ERROR: par%nmu must be >= 1 when par%save_Jmu = .true.");
        r->mu_min = r->xyz_symmetry ? 0.0 : -1.0; r->dmu = (r->xyz_symmetry ? 1.0 : 2.0) / r->nmu;
    }
    if (r->nr > 1) { r->nx = r->nr; r->ny = r->nr; r->nz = r->nr; }
    if (r->sphere) {                                               // a sphere lives in the cube that holds it, with the same cell count along every axis
        if (!r->xy_periodic && !r->use_amr_grid) {
            double m = 0.0;
            for (double v : {r->rmax, r->xmax, r->ymax, r->zmax}) if (v > 0.0 && v > m) m = v;
            if (m > 0.0) { r->rmax = m; r->xmax = m; r->ymax = m; r->zmax = m; }
            if (!r->xy_symmetry) { const int n = std::max(r->nx, std::max(r->ny, r->nz)); r->nx = n; r->ny = n; r->nz = n; }
        }
    } else r->rmax = -1.0;
    if (r->source_rmax < 0.0) r->source_rmax = r->rmax;
    if (r->rmax <= 0.0) r->geometry_JPa = 3;                       // profiles of a box: cell by cell
    if (r->geometry_JPa < -1 || r->geometry_JPa > 3) r->geometry_JPa = r->save_all ? 3 : (r->xy_periodic ? -1 : 1);
    if (r->distance2cm < 0.0) {
        switch (r->distance_unit) { case LART_UNIT_KPC: r->distance2cm = 3.0856776e18 * 1e3; break; case LART_UNIT_PC: r->distance2cm = 3.0856776e18; break;
                                    case LART_UNIT_AU: r->distance2cm = 1.4960e13; break; case LART_UNIT_NONE: r->distance2cm = 1.0; break;
                                    default: r->distance2cm = 3.0856776e18 * 1e3; }
    } else r->distance_unit = LART_UNIT_USER;
    if (r->intensity_unit < 0) r->intensity_unit = r->distance2cm != 1.0 ? 1 : 0;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Cartesian medium
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
// a mirrored axis: an even count starts on the mirror plane, an odd one has its first cell centred on it
inline void sym_axis(int n, double vmax, double &d, double &vmin, int &i0) {
    if ((n / 2) * 2 == n) { d = vmax / n; vmin = 0.0; i0 = 1; } else { d = vmax / (n - 0.5); vmin = -d / 2.0; i0 = 2; }
}
}  // namespace

extern "C" int lart_host_car_medium(lart_host_medium *m, double *xface, double *yface, double *zface, double *rhokap, double *voigt_a, double *Dfreq,
                                    double *vfx, double *vfy, double *vfz, double *rhokapD) {
    if (!m || !xface || !yface || !zface || !rhokap || !voigt_a || !Dfreq || !vfx || !vfy || !vfz) return sfail("car_medium: null argument");
    const int nx = m->nx, ny = m->ny, nz = m->nz;
    if (nx < 1 || ny < 1 || nz < 1) return sfail("car_medium: nx, ny, nz must be >= 1");
    const bool sym3 = m->xyz_symmetry, sym2 = !sym3 && m->xy_symmetry;
    m->i0 = m->j0 = m->k0 = 0;
    if (sym3) { sym_axis(nx, m->xmax, m->dx, m->xmin, m->i0); sym_axis(ny, m->ymax, m->dy, m->ymin, m->j0); sym_axis(nz, m->zmax, m->dz, m->zmin, m->k0); }
    else if (sym2) { sym_axis(nx, m->xmax, m->dx, m->xmin, m->i0); sym_axis(ny, m->ymax, m->dy, m->ymin, m->j0); m->dz = 2.0 * m->zmax / nz; m->zmin = -m->zmax; }
    else { m->dx = 2.0 * m->xmax / nx; m->dy = 2.0 * m->ymax / ny; m->dz = 2.0 * m->zmax / nz; m->xmin = -m->xmax; m->ymin = -m->ymax; m->zmin = -m->zmax; }
    for (int i = 0; i <= nx; i++) xface[i] = i * m->dx + m->xmin;
    for (int j = 0; j <= ny; j++) yface[j] = j * m->dy + m->ymin;
    for (int k = 0; k <= nz; k++) zface[k] = k * m->dz + m->zmin;
    std::vector<double> xx(nx), yy(ny), zz(nz);
    for (int i = 0; i < nx; i++) xx[i] = (xface[i] + xface[i + 1]) / 2.0;
    for (int j = 0; j < ny; j++) yy[j] = (yface[j] + yface[j + 1]) / 2.0;
    for (int k = 0; k < nz; k++) zz[k] = (zface[k] + zface[k + 1]) / 2.0;
    const size_t N = (size_t)nx * ny * nz;
    const double vth = m->vtherm, Df = vth / (m->wavelength0 * UM2KM), va = (m->damping / FOURPI) / Df, H0 = voigt0(va);
    m->Dfreq_ref = Df; m->voigt_amean = (m->damping / FOURPI) / Df;
    // the density: 1 inside the shell [rmin, rmax] (and inside the bicone about z), 0 outside; the opacity per unit length of a gas cell before the
    // normalisation is ONE number, k0 = cross0 / Dfreq
    const double k0 = 1.0 / Df * m->cross0, cosc = (m->cone_opening > 0.0 && m->cone_opening < 90.0) ? std::cos(m->cone_opening * PI / 180.0) : -1.0;
    auto at = [&](int i, int j, int k) { return (size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * (size_t)k); };
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
        const double rr = std::sqrt(xx[i] * xx[i] + yy[j] * yy[j] + zz[k] * zz[k]);
        bool gas = !(m->rmax > 0.0 && (rr < m->rmin || rr > m->rmax));
        if (gas && cosc > -1.0 && rr > 0.0 && std::fabs(zz[k]) / rr < cosc) gas = false;
        const size_t q = at(i, j, k);
        rhokap[q] = gas ? k0 : 0.0; Dfreq[q] = Df; voigt_a[q] = va; vfx[q] = 0.0; vfy[q] = 0.0; vfz[q] = 0.0;
        if (rhokapD) rhokapD[q] = gas ? m->dust_opacity : 0.0;
    }
    // weights of the homogeneous means: a cell cut by a mirror plane counts half per plane
    auto wgt = [&](int i, int j, int k) {
        double w = 1.0;
        if (sym3 || sym2) { if (i == 0 && nx % 2) w /= 2.0; if (j == 0 && ny % 2) w /= 2.0; }
        if (sym3 && k == 0 && nz % 2) w /= 2.0;
        return w;
    };
    const double opac_length = (m->rmax > 0.0 && m->rmin > 0.0) ? m->rmax - m->rmin : (m->rmax > 0.0 ? m->rmax : (m->zmax == -m->zmin ? (m->zmax - m->zmin) / 2.0 : m->zmax - m->zmin));
    const int ic = (sym3 || sym2) ? 0 : (nx + 1) / 2 - 1, jc = (sym3 || sym2) ? 0 : (ny + 1) / 2 - 1;
    // column along the pole (x = y = 0): sum of c(k) dz, halved where the column runs through the whole box, its first cell halved on an odd mirrored axis
    auto pole = [&](double per_cell) {
        double s = 0.0;
        for (int k = 0; k < nz; k++) s = s + rhokap[at(ic, jc, k)] * per_cell;
        if (sym3) return (nz % 2) ? (s - rhokap[at(ic, jc, 0)] * per_cell / 2.0) * m->dz : s * m->dz;
        return (m->zmax == -m->zmin) ? s * m->dz / 2.0 : s * m->dz;
    };
    auto homo = [&](double per_cell) {                             // weighted mean over the gas cells, in storage order
        double s = 0.0, c = 0.0;
        for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
            const double r = rhokap[at(i, j, k)];
            if (r > 0.0) { const double w = wgt(i, j, k); s = s + r * per_cell * w; c = c + w; }
        }
        return s / c;
    };
    double norm = 0.0; bool scaled = true;
    if (m->taumax > 0.0) norm = m->taumax / pole(H0);
    else if (m->tauhomo > 0.0) norm = m->tauhomo / (homo(H0) * opac_length);
    else if (m->N_gasmax > 0.0) norm = m->N_gasmax / (pole(Df) / m->cross0);
    else if (m->N_gashomo > 0.0) norm = m->N_gashomo / (homo(Df) / m->cross0 * opac_length);
    else scaled = false;
    if (scaled) for (size_t q = 0; q < N; q++) { rhokap[q] *= norm; if (rhokapD) rhokapD[q] *= norm; }
    m->taupole_out = pole(H0); m->tauhomo_out = homo(H0) * opac_length;
    m->N_gaspole_out = pole(Df) / m->cross0; m->N_gashomo_out = homo(Df) / m->cross0 * opac_length;
    if (m->taumax <= 0.0) m->taumax = m->taupole_out;
    if (m->tauhomo <= 0.0) m->tauhomo = m->tauhomo_out;
    if (m->N_gasmax <= 0.0) m->N_gasmax = m->N_gaspole_out;
    if (m->N_gashomo <= 0.0) m->N_gashomo = m->N_gashomo_out;
    // bulk velocity of the gas cells, in units of the thermal speed
    if (m->velocity_type != LART_VEL_NONE) {
        const double rpk = m->rmax <= 0.0 ? std::max(m->xmax, std::max(m->ymax, m->zmax)) : m->rmax;
        for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
            const size_t q = at(i, j, k);
            if (!(rhokap[q] > 0.0)) continue;
            const double X = xx[i], Y = yy[j], Z = zz[k], rr = std::sqrt(X * X + Y * Y + Z * Z);
            switch (m->velocity_type) {
            case LART_VEL_HUBBLE: vfx[q] = (m->Vexp / vth) * X / rpk; vfy[q] = (m->Vexp / vth) * Y / rpk; vfz[q] = (m->Vexp / vth) * Z / rpk; break;
            case LART_VEL_CONSTANT_RADIAL:
                if (rr > m->dz / 10.0) { vfx[q] = m->Vexp / vth * X / rr; vfy[q] = m->Vexp / vth * Y / rr; vfz[q] = m->Vexp / vth * Z / rr; } break;
            case LART_VEL_PARALLEL: vfx[q] = m->Vx / vth; vfy[q] = m->Vy / vth; vfz[q] = m->Vz / vth; break;
            case LART_VEL_ROT_SOLID: vfx[q] = -m->Vrot / vth * Y / m->rmax; vfy[q] = m->Vrot / vth * X / m->rmax; break;
            case LART_VEL_ROT_HALO: { const double rc = std::sqrt(X * X + Y * Y), den = rc < m->rinner ? m->rinner : rc;
                vfx[q] = -m->Vrot / vth * Y / den; vfy[q] = m->Vrot / vth * X / den; break; }
            case LART_VEL_POWER_LAW: case LART_VEL_LINEAR_DECEL:
                if (rr > m->dz / 10.0) {
                    const double Vs = m->velocity_type == LART_VEL_POWER_LAW ? m->Vexp * std::pow(rr / rpk, m->velocity_alpha)
                                                                             : m->Vexp * std::max(0.0, (rpk - rr) / (rpk - std::max(m->rmin, 0.0)));
                    const double ur = 1.0 / rr;
                    vfx[q] = Vs / vth * X * ur; vfy[q] = Vs / vth * Y * ur; vfz[q] = Vs / vth * Z * ur;
                } break;
            case LART_VEL_SSH: { const double rp = m->rpeak, Vs = rr < rp ? m->Vpeak / rp * rr : m->Vpeak + m->DeltaV * (rr - rp) / (m->rmax - rp);
                const double ur = rr > 0.0 ? 1.0 / rr : 0.0;
                vfx[q] = Vs / vth * X * ur; vfy[q] = Vs / vth * Y * ur; vfz[q] = Vs / vth * Z * ur; break; }
            default: return sfail("car_medium: unknown velocity_type code");
            }
        }
    }
    return 0;
}

// the spectral axis (car_setup_freq_grid) and the core-skip scale of the grid (grid_mod_car.f90:1186-1219)
extern "C" int lart_host_freq_grid(lart_host_freq *f) {
    if (!f) return sfail("freq_grid: null argument");
    const double vth = f->vtherm, wl0 = f->wavelength0 * 1e4;
    f->atau3 = std::pow(f->voigt_amean * f->tauhomo, 1.0 / 3.0);      // (a power, not cbrt: the floor() of its multiples below sets the axis)
    if (fin(f->wavelength_min) && fin(f->wavelength_max)) {
        if (f->nwavelength == 0 && f->nxfreq > 0) f->nwavelength = f->nxfreq;
        if (f->nwavelength > 0) f->nxfreq = f->nwavelength;
        f->xfreq_min = -(f->wavelength_max - wl0) / wl0 * (SPEEDC / vth);
        f->xfreq_max = -(f->wavelength_min - wl0) / wl0 * (SPEEDC / vth);
    } else if (fin(f->velocity_min) && fin(f->velocity_max)) {
        if (f->nvelocity == 0 && f->nxfreq > 0) f->nvelocity = f->nxfreq;
        if (f->nvelocity > 0) f->nxfreq = f->nvelocity;
        f->xfreq_min = -f->velocity_max / vth; f->xfreq_max = -f->velocity_min / vth;
    }
    if (!(fin(f->xfreq_max) && fin(f->xfreq_min))) {
        double xs = f->taumax <= 5e1 ? 25.0 : (f->taumax <= 5e2 ? 14.0 : (f->taumax <= 5e3 ? 10.0 : 5.0));
        const double a3 = f->atau3, vex = std::fabs(f->Vexp) / vth, split = f->DnuHK_Hz / f->Dfreq_ref;
        if (f->Vexp == 0.0) { f->xfreq_max = std::floor(xs * a3) + 1; f->xfreq_min = -(std::floor(xs * a3 + split) + 1); }
        else if (f->Vexp > 0.0) { f->xfreq_max = std::floor(xs * a3) + 1; f->xfreq_min = -(std::floor(xs * a3 + vex + split) + 1); }
        else { f->xfreq_max = std::floor(xs * a3 + vex) + 1; f->xfreq_min = -(std::floor(xs * a3 + split) + 1); }
        if (f->continuum) { xs = 4.0 * xs; f->xfreq_max = std::floor(xs * a3 + vex) + 1; f->xfreq_min = -(std::floor(xs * a3 + vex + split) + 1); }
    }
    if (f->nxfreq < 1) return sfail("freq_grid: nxfreq must be >= 1");
    f->dxfreq = (f->xfreq_max - f->xfreq_min) / f->nxfreq;
    f->dwave = vth / SPEEDC * wl0 * f->dxfreq;
    if (f->line_type == 8) {                                       // :1518-1529 the band-2 (H-alpha) axis
        if (f->nxfreq_Ha <= 0) f->nxfreq_Ha = f->nxfreq;
        if (f->xfreq_max_Ha > 0.0) f->xfreq_min_Ha = -f->xfreq_max_Ha; else { f->xfreq_min_Ha = f->xfreq_min; f->xfreq_max_Ha = f->xfreq_max; }
        f->dxfreq_Ha = (f->xfreq_max_Ha - f->xfreq_min_Ha) / f->nxfreq_Ha;
        f->dwave_Ha = vth / SPEEDC * (f->wavelength0_Ha * 1e4) * f->dxfreq_Ha;
    }
    f->xcrit = 0.0; f->xcrit2 = 0.0;
    const double atau0 = f->voigt_amean * f->tauhomo, at = f->core_skip_global ? atau0 : atau0 / f->cells_per_radius;
    if (at > 1.0) {
        const double xi = at <= 60.0 ? 0.6 : 1.4, chi = at <= 60.0 ? 1.2 : 0.6;
        f->xcrit = 0.02 * std::exp(xi * std::pow(std::log(at), chi)); f->xcrit2 = f->xcrit * f->xcrit;
    }
    return 0;
}
// bin centres of the axis: x, Doppler velocity [km/s], wavelength [A]
extern "C" int lart_host_freq_axis(int32_t n, double xfreq_min, double dxfreq, double vtherm, double wavelength0_um, double *xfreq, double *velocity, double *wavelength) {
    if (n < 1 || !xfreq) return sfail("freq_axis: bad argument");
    for (int i = 0; i < n; i++) {
        xfreq[i] = ((i + 1) - 0.5) * dxfreq + xfreq_min;
        const double v = -vtherm * xfreq[i];
        if (velocity) velocity[i] = v;
        if (wavelength) wavelength[i] = (v / SPEEDC + 1.0) * (wavelength0_um * 1e4);
    }
    return 0;
}

// profile bin of every cell (1-based, 0 = none) and the gas cells per bin, mirror images included: geometry_JPa 1 shells, -1 layers, 3 cells
extern "C" int lart_host_jpa_car(int32_t geometry_JPa, int32_t xyz_symmetry, int32_t nx, int32_t ny, int32_t nz, const double *xface, const double *yface, const double *zface,
                                 double xmax, double ymax, double zmax, const double *rhokap, int32_t *nbins, int32_t *ind_bin, double *ncount) {
    if (!xface || !yface || !zface || !rhokap || !nbins || !ind_bin || !ncount) return sfail("jpa_car: null argument");
    auto at = [&](int i, int j, int k) { return (size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * (size_t)k); };
    auto images = [&](int i, int j, int k) {                        // how many cells of the full box one stored cell stands for
        if (!xyz_symmetry) return 1.0;
        double w = 8.0; if (i == 0 && nx % 2) w /= 2.0; if (j == 0 && ny % 2) w /= 2.0; if (k == 0 && nz % 2) w /= 2.0; return w;
    };
    const size_t N = (size_t)nx * ny * nz;
    if (geometry_JPa == 1) {
        int nr = std::max(nx, std::max(ny, nz));
        if (!xyz_symmetry) nr = nr % 2 == 0 ? nr / 2 : (nr - 1) / 2 + 1;
        const double rmax = std::min(xmax, std::min(ymax, zmax));
        const double dr = nr % 2 == 0 ? rmax / nr : rmax / (nr - 0.5), roff = nr % 2 == 0 ? 0.0 : -dr / 2.0;
        for (int b = 0; b < nr; b++) ncount[b] = 0.0;
        for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
            const double X = (xface[i] + xface[i + 1]) / 2.0, Y = (yface[j] + yface[j + 1]) / 2.0, Z = (zface[k] + zface[k + 1]) / 2.0;
            const long ib = (long)std::floor((std::sqrt(X * X + Y * Y + Z * Z) - roff) / dr) + 1;
            const bool ok = ib >= 1 && ib <= nr;
            ind_bin[at(i, j, k)] = ok ? (int32_t)ib : 0;
            if (ok && rhokap[at(i, j, k)] > 0.0) ncount[ib - 1] += images(i, j, k);
        }
        *nbins = nr; return 0;
    }
    if (geometry_JPa == -1) {
        for (int k = 0; k < nz; k++) { double c = 0.0;
            for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) { ind_bin[at(i, j, k)] = k + 1; if (rhokap[at(i, j, k)] > 0.0) c += 1.0; }
            ncount[k] = c; }
        *nbins = nz; return 0;
    }
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) { ind_bin[at(i, j, k)] = (int32_t)(at(i, j, k) + 1); ncount[at(i, j, k)] = images(i, j, k); }
    *nbins = (int32_t)N; return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// observers
// ---------------------------------------------------------------------------------------------------------------------------------
// In / out arrays of LART_MAX_OBSERVERS entries, NaN = not given: alpha, beta, gamma [deg], obsx, obsy, obsz; *distance, *dxim, *dyim likewise.
// Out: nobs, xyz[3 nobs], rmatrix[9 nobs] (row-major: rm[3 r + c] = rmatrix(r+1, c+1)), steradian_pix.
extern "C" int lart_host_observers(lart_host_obs *o, double *alpha, double *beta, double *gamma, double *obsx, double *obsy, double *obsz, double *xyz, double *rmatrix) {
    if (!o || !alpha || !beta || !gamma || !obsx || !obsy || !obsz || !xyz || !rmatrix) return sfail("observers: null argument");
    const int M = LART_MAX_OBSERVERS;
    if (o->nside > 0) {                                            // an observer inside the medium looks in every direction of a HEALPix sphere
        if (!(fin(obsx[0]) && fin(obsy[0]) && fin(obsz[0]))) { obsx[0] = 0.0; obsy[0] = 0.0; obsz[0] = 0.0; }
        int n = 0;
        for (int i = 0; i < M; i++) if (fin(obsx[i]) && fin(obsy[i]) && fin(obsz[i])) n++;
        for (int i = 0; i < n; i++) { xyz[3 * i] = obsx[i]; xyz[3 * i + 1] = obsy[i]; xyz[3 * i + 2] = obsz[i]; }
        o->nobs = n; o->npix = 12 * o->nside * o->nside; o->nxim = o->npix; o->nyim = 1; o->steradian_pix = 4.0 * PI / o->npix;
        return 0;
    }
    const double maxsz = std::max(o->xmax, std::max(o->ymax, o->zmax));
    for (int i = 0; i < M; i++) { if (fin(beta[i]) && !fin(alpha[i])) alpha[i] = 0.0; if (fin(alpha[i]) && !fin(beta[i])) beta[i] = 0.0; }
    const bool by_angle0 = fin(alpha[0]) && fin(beta[0]), by_pos0 = fin(obsx[0]) && fin(obsy[0]) && fin(obsz[0]);
    if (!by_angle0 && !by_pos0) {                                  // nothing given: one observer on the +z axis
        if (!fin(o->distance)) o->distance = maxsz * 100.0;
        obsx[0] = 0.0; obsy[0] = 0.0; obsz[0] = 1.0; alpha[0] = 0.0; beta[0] = 0.0;
    }
    auto default_gamma = [&](int i) { if (!fin(gamma[i])) gamma[i] = (beta[i] > 0.0 && beta[i] <= 90.0) ? 90.0 : (beta[i] > 90.0 ? -90.0 : 0.0); };
    int n = 0; double tr[LART_MAX_OBSERVERS][6];
    if (fin(alpha[0]) && fin(beta[0])) {
        for (int i = 0; i < M; i++) if (fin(alpha[i]) && fin(beta[i])) n++;
        if (!fin(o->distance)) o->distance = maxsz * 100.0;
        for (int i = 0; i < n; i++) {
            default_gamma(i);
            const double ca = std::cos(alpha[i] * DEG2RAD), sa = std::sin(alpha[i] * DEG2RAD), cb = std::cos(beta[i] * DEG2RAD), sb = std::sin(beta[i] * DEG2RAD);
            xyz[3 * i] = o->distance * ca * sb; xyz[3 * i + 1] = o->distance * sa * sb; xyz[3 * i + 2] = o->distance * cb;
            tr[i][0] = ca; tr[i][1] = sa; tr[i][2] = cb; tr[i][3] = sb; tr[i][4] = std::cos(gamma[i] * DEG2RAD); tr[i][5] = std::sin(gamma[i] * DEG2RAD);
        }
    } else {
        for (int i = 0; i < M; i++) if (fin(obsx[i]) && fin(obsy[i]) && fin(obsz[i])) n++;
        if (!fin(o->distance)) {
            o->distance = std::sqrt(obsx[0] * obsx[0] + obsy[0] * obsy[0] + obsz[0] * obsz[0]);
            if (o->distance < 10.0 * maxsz) o->distance = maxsz * 100.0;
        }
        const double eps = 2.220446049250313e-16;
        for (int i = 0; i < n; i++) {
            default_gamma(i);
            const double ds = o->distance / std::sqrt(obsx[i] * obsx[i] + obsy[i] * obsy[i] + obsz[i] * obsz[i]);
            double p[3] = {obsx[i], obsy[i], obsz[i]};
            if (ds > 1.001) { p[0] *= ds; p[1] *= ds; p[2] *= ds; }   // a position closer than par%distance is pushed out along its direction
            double cb = p[2] / o->distance;
            if (std::fabs(cb - 1.0) < eps) cb = 1.0;
            if (std::fabs(cb + 1.0) < eps) cb = -1.0;
            const double sb = std::sqrt(1.0 - cb * cb);
            beta[i] = std::atan2(sb, cb) * RAD2DEG;
            double ca = 1.0, sa = 0.0;
            if (sb == 0.0) alpha[i] = 0.0; else { const double al = std::atan2(p[1], p[0]); ca = std::cos(al); sa = std::sin(al); alpha[i] = al * RAD2DEG; }
            xyz[3 * i] = p[0]; xyz[3 * i + 1] = p[1]; xyz[3 * i + 2] = p[2];
            tr[i][0] = ca; tr[i][1] = sa; tr[i][2] = cb; tr[i][3] = sb; tr[i][4] = std::cos(gamma[i] * DEG2RAD); tr[i][5] = std::sin(gamma[i] * DEG2RAD);
        }
    }
    o->nobs = n;
    for (int i = 0; i < n; i++) {                                  // the Euler rotation z(alpha) y(beta) z(gamma) that puts the observer on the +z axis of its own frame
        const double ca = tr[i][0], sa = tr[i][1], cb = tr[i][2], sb = tr[i][3], cg = tr[i][4], sg = tr[i][5];
        double *r = rmatrix + 9 * i;
        r[0] = ca * cb * cg - sa * sg; r[1] = sa * cb * cg + ca * sg; r[2] = -sb * cg;
        r[3] = -ca * cb * sg - sa * cg; r[4] = -sa * cb * sg + ca * cg; r[5] = sb * sg;
        r[6] = ca * sb; r[7] = sa * sb; r[8] = cb;
    }
    if (!(fin(o->dxim) && fin(o->dyim))) {                          // pixel size so that the image just holds the medium
        if (o->sphere) {
            o->dxim = std::asin(o->rmax / o->distance) / (o->nxim / 2.0) * RAD2DEG;
            o->dyim = std::asin(o->rmax / o->distance) / (o->nyim / 2.0) * RAD2DEG;
        } else {
            const int vx[8] = {1, 1, 1, -1, -1, -1, 1, -1}, vy[8] = {1, 1, -1, 1, -1, 1, -1, -1}, vz[8] = {1, -1, 1, 1, 1, -1, -1, -1};
            double mx = -999.0, my = -999.0;
            for (int i = 0; i < n; i++) for (int c = 0; c < 8; c++) {      // the corners of the box as the observers see them
                const double pv[3] = {xyz[3 * i] - vx[c] * o->xmax, xyz[3 * i + 1] - vy[c] * o->ymax, xyz[3 * i + 2] - vz[c] * o->zmax};
                const double *r = rmatrix + 9 * i;
                const double kx = r[0] * pv[0] + r[1] * pv[1] + r[2] * pv[2], ky = r[3] * pv[0] + r[4] * pv[1] + r[5] * pv[2], kz = r[6] * pv[0] + r[7] * pv[1] + r[8] * pv[2];
                mx = std::max(mx, std::fabs(std::atan2(-kx, kz))); my = std::max(my, std::fabs(std::atan2(-ky, kz)));
            }
            if (o->nxim == o->nyim) { o->dxim = std::max(mx, my) / (o->nxim / 2.0) * RAD2DEG; o->dyim = std::max(mx, my) / (o->nyim / 2.0) * RAD2DEG; }
            else { if (!fin(o->dxim)) o->dxim = mx / (o->nxim / 2.0) * RAD2DEG; if (!fin(o->dyim)) o->dyim = my / (o->nyim / 2.0) * RAD2DEG; }
        }
    }
    o->steradian_pix = o->dxim * o->dyim * DEG2RAD * DEG2RAD;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// clump population scalars
// ---------------------------------------------------------------------------------------------------------------------------------
// population size, filling / covering factors and the opacity of one clump from whichever of the input quantities is given
extern "C" int lart_host_clump_population(lart_host_clumps_in *c) {
    if (!c) return sfail("clump_population: null argument");
    if (!(c->clump_radius > 0.0)) return sfail("ERROR: clump_radius must be > 0");
    const double R = c->sphere_R, r0 = std::max(0.0, c->rmin), rc = c->clump_radius;
    if (r0 >= R) return sfail("ERROR: par%rmin must be < par%rmax for clump placement");
    c->r_min = r0;
    const double shell2 = R * R + R * r0 + r0 * r0, tiny = 2.2250738585072014e-308;
    auto nint = [](double v) { return v >= 0.0 ? (long long)std::floor(v + 0.5) : -(long long)std::floor(-v + 0.5); };
    long long N;
    if (c->N_clumps_in > 0.0) N = (long long)c->N_clumps_in;
    else if (c->f_vol_in > 0.0) N = nint(c->f_vol_in * (R * R * R - r0 * r0 * r0) / (rc * rc * rc));
    else if (c->f_cov_in > 0.0) N = nint((4.0 / 3.0) * c->f_cov_in * shell2 / (rc * rc));
    else return sfail("ERROR: specify clump_N_clumps, clump_f_vol, or clump_f_cov");
    c->N = std::max(N, 1LL);
    c->fvol = c->N * (rc * rc * rc) / std::max(R * R * R - r0 * r0 * r0, tiny);
    c->fcov = 0.75 * c->N * rc * rc / std::max(shell2, tiny);
    c->Dfreq_ref = c->vtherm / (c->wavelength0 * UM2KM); c->voigt_a_ref = (c->damping / FOURPI) / c->Dfreq_ref;
    const double H0 = voigt0(c->voigt_a_ref);
    if (c->clump_tau0 > 0.0) c->rhokap_ref = c->clump_tau0 / (H0 * rc);
    else if (c->clump_NHI > 0.0) c->rhokap_ref = c->clump_NHI * c->cross0 / (c->Dfreq_ref * rc);
    else if (c->clump_nH > 0.0) { if (c->distance2cm <= 0.0) return sfail("ERROR: clump_nH requires distance_unit"); c->rhokap_ref = c->clump_nH * c->cross0 * c->distance2cm / c->Dfreq_ref; }
    else if (c->taumax > 0.0 || c->N_gasmax > 0.0) {
        const double GF = c->N * (rc * rc * rc) / std::max(shell2, tiny);
        c->rhokap_ref = c->taumax > 0.0 ? c->taumax / (GF * H0) : c->N_gasmax * c->cross0 / (GF * c->Dfreq_ref);
    } else return sfail("ERROR: specify clump_tau0, clump_NHI, clump_nH, taumax, or N_HImax");
    return 0;
}
// the four system-level scalars of a population (sequential sums in clump order): out[0..3] = tauhomo, taumax, N_gashomo, N_gasmax
extern "C" int lart_host_clump_scalars(int64_t n, double sphere_R, double r_min, double cross0, const double *x, const double *y, const double *z, const double *radius,
                                       const double *rhokap, const double *voigt_a, const double *Dfreq, double *out4) {
    if (n < 1 || !x || !y || !z || !radius || !rhokap || !voigt_a || !Dfreq || !out4) return sfail("clump_scalars: bad argument");
    const double shell2 = sphere_R * sphere_R + sphere_R * r_min + r_min * r_min;
    double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
    for (int64_t i = 0; i < n; i++) {
        const double r3 = radius[i] * radius[i] * radius[i], di2 = std::max(x[i] * x[i] + y[i] * y[i] + z[i] * z[i], radius[i] * radius[i]);
        const double wh = r3 / shell2, wm = r3 / (3.0 * di2), H0 = voigt0(voigt_a[i]);
        a = a + rhokap[i] * H0 * wh; b = b + rhokap[i] * H0 * wm; c = c + rhokap[i] * Dfreq[i] * wh / cross0; d = d + rhokap[i] * Dfreq[i] * wm / cross0;
    }
    out4[0] = a; out4[1] = b; out4[2] = c; out4[3] = d;
    return 0;
}
// systematic velocity field of a clump population, added to v (units of the clumps' thermal speed)
extern "C" int lart_host_clump_velocities(int64_t n, int32_t velocity_type, double sphere_R, double Vexp, double Vx, double Vy, double Vz, double Vrot, double rinner,
                                          const double *x, const double *y, const double *z, const double *vtherm, double *vx, double *vy, double *vz) {
    if (n < 1 || !x || !y || !z || !vtherm || !vx || !vy || !vz) return sfail("clump_velocities: bad argument");
    for (int64_t i = 0; i < n; i++) {
        double ax = 0.0, ay = 0.0, az = 0.0; const double X = x[i], Y = y[i], Z = z[i];
        switch (velocity_type) {
        case LART_VEL_HUBBLE: ax = Vexp * X / sphere_R; ay = Vexp * Y / sphere_R; az = Vexp * Z / sphere_R; break;
        case LART_VEL_CONSTANT_RADIAL: { const double rr = std::sqrt(X * X + Y * Y + Z * Z); if (rr > 0.0) { ax = Vexp * X / rr; ay = Vexp * Y / rr; az = Vexp * Z / rr; } break; }
        case LART_VEL_PARALLEL: ax = Vx; ay = Vy; az = Vz; break;
        case LART_VEL_ROT_SOLID: ax = -Vrot * Y / sphere_R; ay = Vrot * X / sphere_R; break;
        case LART_VEL_ROT_HALO: { const double rc = std::sqrt(X * X + Y * Y);
            if (rc > 0.0 && rc < rinner) { ax = -Vrot * Y / rinner; ay = Vrot * X / rinner; } else if (rc >= rinner) { ax = -Vrot * Y / rc; ay = Vrot * X / rc; } break; }
        default: break;
        }
        vx[i] = vx[i] + ax / vtherm[i]; vy[i] = vy[i] + ay / vtherm[i]; vz[i] = vz[i] + az / vtherm[i];
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dust phase matrix: alias table of the scattering-angle bins
// ---------------------------------------------------------------------------------------------------------------------------------
// Walker / Vose alias table: probs[n] in (sum 1), out scaled probs (x n) and the 1-based alias of every bin
extern "C" int lart_host_alias_setup(int32_t n, double *probs, int32_t *alias) {
    if (n < 1 || !probs || !alias) return sfail("alias_setup: bad argument");
    std::vector<int> small, large;
    for (int i = 0; i < n; i++) { probs[i] = probs[i] * n; alias[i] = 0; }
    for (int i = 0; i < n; i++) (probs[i] < 1.0 ? small : large).push_back(i + 1);
    while (!small.empty() && !large.empty()) {
        const int ks = small.back(), kl = large.back(); small.pop_back(); large.pop_back();
        alias[ks - 1] = kl;
        probs[kl - 1] = probs[kl - 1] + probs[ks - 1] - 1.0;
        (probs[kl - 1] < 1.0 ? small : large).push_back(kl);
    }
    return 0;
}
// S11..S34 normalised by the trapezoid integral of S11 over cos(theta), the bin probabilities and their alias table
extern "C" int lart_host_mueller_setup(int32_t n, const double *coss, double *S11, double *S12, double *S33, double *S34, double *phase_PDF, int32_t *alias) {
    if (n < 2 || !coss || !S11 || !S12 || !S33 || !S34 || !phase_PDF || !alias) return sfail("mueller_setup: bad argument");
    double norm = 0.0;
    for (int i = 1; i < n; i++) norm = norm + 0.5 * (S11[i] + S11[i - 1]) * std::fabs(coss[i] - coss[i - 1]);
    for (int i = 0; i < n; i++) { S11[i] /= norm; S12[i] /= norm; S33[i] /= norm; S34[i] /= norm; }
    double tot = 0.0;
    for (int i = 0; i < n - 1; i++) { phase_PDF[i] = (S11[i] + S11[i + 1]) / 2.0; }
    // (pairwise sum as numpy's: the bins of the shipped table are ~1e3, the order matters in the last bit of every probability)
    { std::vector<double> v(phase_PDF, phase_PDF + (n - 1));
      struct P { static double sum(const double *a, size_t m) { if (m < 8) { double s = 0.0; for (size_t i = 0; i < m; i++) s += a[i]; return s; }
          if (m <= 128) { double r[8]; for (int q = 0; q < 8; q++) r[q] = a[q]; size_t i = 8; for (; i + 8 <= m; i += 8) for (int q = 0; q < 8; q++) r[q] += a[i + q];
              double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7])); for (; i < m; i++) s += a[i]; return s; }
          size_t h = m / 2; h -= h % 8; return sum(a, h) + sum(a + h, m - h); } };
      tot = P::sum(v.data(), v.size()); }
    for (int i = 0; i < n - 1; i++) phase_PDF[i] = phase_PDF[i] / tot;
    return lart_host_alias_setup(n - 1, phase_PDF, alias);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// output_normalize: the divisors
// ---------------------------------------------------------------------------------------------------------------------------------
extern "C" int lart_host_norm_scales(lart_host_norm *s) {
    if (!s) return sfail("norm_scales: null argument");
    double area;
    if (s->xy_periodic) area = 2.0;                                // a slab radiates through its two faces, one photon per unit area
    else if (s->sphere) area = FOURPI * s->rmax * s->rmax * s->distance2cm * s->distance2cm;
    else if (s->amr_L_box > 0.0) area = 6.0 * s->amr_L_box * s->amr_L_box * s->distance2cm * s->distance2cm;
    else area = (s->xmax * s->ymax + s->ymax * s->zmax + s->zmax * s->xmax) * 8.0 * s->distance2cm * s->distance2cm;
    s->area = area;
    s->scale_J = s->nphotons * s->bin_unit * TWOPI * area;         // Jout, Jin, Jabs; Jmu: x nmu / scale_J
    s->scale_J_Ha = s->nphotons * s->bin_unit_Ha * TWOPI * area;   // the band-2 spectra of line type 8
    s->scale_cube = s->nphotons * s->steradian_pix * s->bin_unit * s->distance2cm * s->distance2cm;
    s->scale_image = s->nphotons * s->steradian_pix * s->distance2cm * s->distance2cm;
    s->scale_cube_Ha = s->nphotons * s->steradian_pix * s->bin_unit_Ha * s->distance2cm * s->distance2cm;
    return 0;
}
