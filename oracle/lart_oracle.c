/* TEST INFRASTRUCTURE ONLY -- see lart_oracle.h.  Plain C restatement of the LaRT (v2.00) photon-packet
 * hot path for Cartesian media.  Every routine cites the reference file:line it follows
 * (paths relative to /root/reference/src).  Arithmetic is IEEE double, no FMA contraction
 * (-ffp-contract=off), evaluation order as in the reference. */
#define _GNU_SOURCE
#include "lart_oracle.h"
#include "voigt_tables.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants: define.f90:43-78 ---- */
#define PI      3.141592653589793238462643383279502884197
#define TWOPI   6.283185307179586476925286766559005768394
#define FOURPI  12.56637061435917295385057353311801153679
#define HALFPI  (PI/2.0)
#define RAD2DEG (180.0/PI)
#define DEG2RAD (PI/180.0)
#define HUGEST  DBL_MAX
#define SPEEDC  2.99792458e5
#define UM2KM   1.0e-9
#define UM2M    1.0e-6
#define H_PLANCK 6.62607004e-34
#define TAU_HUGE 745.2                      /* raytrace_car.f90:106,184,432,497 */
#define MAX_OBS 16

static char g_err[512];
const char *orc_last_error(void) { return g_err; }

/* =====================================================================================
 * RNG
 * ===================================================================================== */
/* MT19937-64 (Matsumoto & Nishimura 2004), as random_mt.f90:486-498 (init_mt) and :579-630 (rand_mt) */
#define MT_N 312
#define MT_M 156
typedef struct { uint64_t mt[MT_N]; int mti; } mt_t;
static void mt_init(mt_t *s, int64_t seed) {
    s->mt[0] = (uint64_t)seed;
    for (int i = 1; i < MT_N; i++)
        s->mt[i] = 6364136223846793005ULL * (s->mt[i-1] ^ (s->mt[i-1] >> 62)) + (uint64_t)i;
    s->mti = MT_N;
}
static uint64_t mt_next(mt_t *s) {
    static const uint64_t mag01[2] = {0ULL, 0xB5026F5AA96619E9ULL};
    const uint64_t UM = 0xFFFFFFFF80000000ULL, LM = 0x7FFFFFFFULL;
    if (s->mti >= MT_N) {
        int i; uint64_t x;
        for (i = 0; i < MT_N - MT_M; i++) {
            x = (s->mt[i] & UM) | (s->mt[i+1] & LM);
            s->mt[i] = s->mt[i+MT_M] ^ (x >> 1) ^ mag01[x & 1ULL];
        }
        for (; i < MT_N - 1; i++) {
            x = (s->mt[i] & UM) | (s->mt[i+1] & LM);
            s->mt[i] = s->mt[i+(MT_M-MT_N)] ^ (x >> 1) ^ mag01[x & 1ULL];
        }
        x = (s->mt[MT_N-1] & UM) | (s->mt[0] & LM);
        s->mt[MT_N-1] = s->mt[MT_M-1] ^ (x >> 1) ^ mag01[x & 1ULL];
        s->mti = 0;
    }
    uint64_t x = s->mt[s->mti++];
    x ^= (x >> 29) & 0x5555555555555555ULL;
    x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
    x ^= (x << 37) & 0xFFF7EEE000000000ULL;
    x ^= (x >> 43);
    return x;
}

/* Philox4x32-10 (Salmon et al. 2011, "Parallel random numbers: as easy as 1, 2, 3") */
void orc_philox4x32_10(const uint32_t c_in[4], const uint32_t k_in[2], uint32_t out[4]) {
    uint32_t c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3], k0 = k_in[0], k1 = k_in[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct {
    int kind;                 /* 0 MT stream, 1 philox per photon */
    mt_t *mt;                 /* kind 0 */
    int gaus_stored; double gset;   /* rand_gauss1 cache (random_mt.f90:968-987), kind 0 only */
    uint64_t seed, id, n;     /* kind 1: key, stream id, draw index */
    double spare;             /* second half of the current philox block */
} rng_t;

/* uniform deviate on the OPEN interval (0,1): (x>>12 + 0.5)/2^52   (random_mt.f90:627-629) */
/* -------------------------------------------------------------------------------------
 * Decision margins (divergence audit of the photon-by-photon parity tests, tests/test_gpu_divergence.py).  While a history is
 * traced (orc_trace) every data-dependent comparison on the path notes how close it was: the relative distance of the two
 * compared numbers, divided by the condition number of the ill-conditioned side where there is one (the acceptance window of the
 * wing sampler is a difference of two arctangents).  The trace keeps, per event, the smallest margin and which comparison had it:
 * a device history may only part from the oracle's at an event whose smallest margin is at rounding level.
 * ids: 10-19 rand_resonance_vz, 20 gauss pair, 21-22 Voigt regime / table node, 30 alias table, 40-42 Cartesian march,
 * 43-45 octree, 46-55 clumps, 60-66 scattering, 70-72 birth / first flight
 * ------------------------------------------------------------------------------------- */
static __thread int g_mrg_on = 0, g_mrg_id = 0; static __thread double g_mrg = 1e300;
static inline void mrg_note(double l, double r, int id, double cond) {
    if (!g_mrg_on) return;
    double sc = fmax(fabs(l), fabs(r)); if (!(sc > 0.0)) sc = 1e-300;
    double m = fabs(l - r) / (sc * (cond > 1.0 ? cond : 1.0));
    if (m < g_mrg) { g_mrg = m; g_mrg_id = id; }
}
#define MRG(l, r, id) mrg_note((l), (r), (id), 1.0)
static inline double u64_to_open01(uint64_t x) { return ((double)(x >> 12) + 0.5) * (1.0 / 4503599627370496.0); }

static double rnd(rng_t *g) {
    if (g->kind == 0) return u64_to_open01(mt_next(g->mt));
    uint64_t n = g->n++;
    if (n & 1ULL) return g->spare;
    uint64_t blk = n >> 1;
    uint32_t c[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)g->id, (uint32_t)(g->id >> 32)};
    uint32_t k[2] = {(uint32_t)g->seed, (uint32_t)(g->seed >> 32)}, o[4];
    orc_philox4x32_10(c, k, o);
    g->spare = u64_to_open01((uint64_t)o[2] | ((uint64_t)o[3] << 32));
    return u64_to_open01((uint64_t)o[0] | ((uint64_t)o[1] << 32));
}
static void rng_philox(rng_t *g, uint64_t seed, uint64_t id, uint64_t n0) {
    memset(g, 0, sizeof *g); g->kind = 1; g->seed = seed; g->id = id; g->n = n0 & ~1ULL;
    if (n0 & 1ULL) { (void)rnd(g); }          /* re-derive the spare half when resuming on an odd draw */
}

/* Marsaglia polar pair (random_mt.f90:964-988).  Returns the deviate the reference returns first
 * (v2*rsq); *second = v1*rsq is the one the reference caches. */
static double gauss_pair(rng_t *g, double *second) {
    double v1, v2, rsq;
    do {
        v1 = 2.0 * rnd(g) - 1.0;
        v2 = 2.0 * rnd(g) - 1.0;
        rsq = v1 * v1 + v2 * v2;
        MRG(rsq, 1.0, 20);
    } while (!(rsq > 0.0 && rsq < 1.0));
    rsq = sqrt(-2.0 * log(rsq) / rsq);
    *second = v1 * rsq;
    return v2 * rsq;
}
/* rand_gauss: MT mode keeps the reference's hidden cache; the counter-based mode has none
 *  and discards the second deviate. */
static double rand_gauss(rng_t *g) {
    double s;
    if (g->kind == 0) {
        if (g->gaus_stored) { g->gaus_stored = 0; return g->gset; }
        double r = gauss_pair(g, &s); g->gset = s; g->gaus_stored = 1; return r;
    }
    return gauss_pair(g, &s);
}

/* =====================================================================================
 * Voigt H(x,a): voigt_mod.f90:541-733 (voigt_seon2).  u(k) = k-1 so (vv - u(k1)) = vv - (k1-1).
 * ===================================================================================== */
double orc_voigt(double vin, double a) {
    double v = (vin < 0.0) ? -vin : vin, vv, y1, y2, y3;
    int k1, k2, k3;
    if (g_mrg_on) { MRG(v, 1.0, 21); MRG(v, 5.0, 21); MRG(v, 10.0, 21); if (v >= 1.0 && v < 5.0) { double vq = v * 20.0; MRG(vq, rint(vq), 22); } }
    if (v < 1.0) {
        vv = v * 20.0; k1 = (int)vv + 1; k2 = k1 + 1;
        y1 = VOIGT_H0[k1-1] + a * (VOIGT_H1[k1-1] + a * VOIGT_H2[k1-1]);
        y2 = VOIGT_H0[k2-1] + a * (VOIGT_H1[k2-1] + a * VOIGT_H2[k2-1]);
        return y1 + (y2 - y1) * (vv - (double)(k1 - 1));
    } else if (v < 5.0) {
        vv = v * 20.0; k1 = (int)vv; k2 = k1 + 1; k3 = k1 + 2;
        y1 = VOIGT_H0[k1-1] + a * (VOIGT_H1[k1-1] + a * (VOIGT_H2[k1-1] + a * VOIGT_H3[k1-1]));
        y2 = VOIGT_H0[k2-1] + a * (VOIGT_H1[k2-1] + a * (VOIGT_H2[k2-1] + a * VOIGT_H3[k2-1]));
        y3 = VOIGT_H0[k3-1] + a * (VOIGT_H1[k3-1] + a * (VOIGT_H2[k3-1] + a * VOIGT_H3[k3-1]));
        double u1 = (double)(k1 - 1), u2 = (double)(k2 - 1), u3 = (double)(k3 - 1);
        return 0.5 * y1 * (vv - u2) * (vv - u3) - y2 * (vv - u1) * (vv - u3) + 0.5 * y3 * (vv - u1) * (vv - u2);
    } else if (v < 10.0) {
        vv = v * 20.0; k1 = (int)vv + 1; k2 = k1 + 1;
        double a2 = a * a;
        y1 = a * (VOIGT_H1[k1-1] + a2 * VOIGT_H3[k1-1]);
        y2 = a * (VOIGT_H1[k2-1] + a2 * VOIGT_H3[k2-1]);
        return y1 + (y2 - y1) * (vv - (double)(k1 - 1));
    }
    double v2 = 1.0 / (v * v);
    return 0.56418958354775628695 * a * v2 * (1.0 + ((1.5 - a * a) + 3.75 * v2) * v2);
}
void orc_voigt_v(int n, const double *x, const double *a, double *h) { for (int i = 0; i < n; i++) h[i] = orc_voigt(x[i], a[i]); }

/* =====================================================================================
 * Samplers
 * ===================================================================================== */
/* rand_resonance_vz_seon: random_mt.f90:2562-2696 */
static double rand_resonance_vz(rng_t *g, double x0in, double a) {
    const double xc = 1.0 + sqrt(2.0), two_over_PI = 2.0 / PI;
    double x0 = fabs(x0in), vz, beta = 0, Cb = 0, pb, t1 = 0, t2, delt = 0;
    MRG(x0, 1.0, 10);
    if (x0 <= 1.0) {
        for (;;) {
            vz = x0 + a * tan(PI * (rnd(g) - 0.5));
            const double r_ = rnd(g), e_ = exp(-(vz * vz)); MRG(r_, e_, 11);
            if (r_ <= e_) break;
        }
        return (x0in < 0.0) ? -vz : vz;
    }
    double x0sq = x0 * x0, api = a * PI, beta0 = exp(-x0sq / 2.0), h0_two = beta0 / a, h0 = h0_two / 2.0;
    MRG(x0, xc, 12);
    if (x0 < xc) {
        double dbeta = sqrt(two_over_PI * a * (1.0 - beta0) * beta0 * x0), beta1 = beta0 + dbeta, one_b1 = 1.0 - beta1;
        double pb1 = sqrt(-2.0 * log(beta1)), h1 = two_over_PI * beta1 * pb1 / (x0sq - pb1 * pb1);
        double S0 = beta0 * h0, S1 = dbeta * h0, S2 = one_b1 * h1, Stot = S0 + S1 + S2;
        for (;;) {
            double rS = rnd(g); MRG(rS, S0 / Stot, 13); MRG(rS, 1.0 - S2 / Stot, 14);
            if (rS < S0 / Stot)              { beta = beta0 * sqrt(rnd(g)); Cb = beta / a; }
            else if (rS < 1.0 - S2 / Stot)   { beta = beta0 + dbeta * rnd(g); Cb = h0; }
            else                             { beta = beta1 + one_b1 * rnd(g); Cb = h1; }
            pb = sqrt(-2.0 * log(beta));
            t2 = atan((pb - x0) / a); t1 = atan((-pb - x0) / a); delt = t2 - t1;
            { const double l_ = rnd(g) * Cb, r_ = (beta / api) * delt; mrg_note(l_, r_, 15, (fabs(t1) + fabs(t2)) / fabs(delt)); if (l_ < r_) break; }
        }
    } else {
        double h2 = 0.3861 / (x0sq - 1.373);
        MRG(h0_two, h2, 16); MRG(h0, h2, 17);
        if (h0_two < h2) {
            for (;;) {
                beta = rnd(g); Cb = h2;
                pb = sqrt(-2.0 * log(beta));
                t2 = atan((pb - x0) / a); t1 = atan((-pb - x0) / a); delt = t2 - t1;
                { const double l_ = rnd(g) * Cb, r_ = (beta / api) * delt; mrg_note(l_, r_, 15, (fabs(t1) + fabs(t2)) / fabs(delt)); if (l_ < r_) break; }
            }
        } else if (h0 < h2) {
            double S0 = beta0 * h0, one_b0 = 1.0 - beta0, S1 = one_b0 * h2, Stot = S0 + S1;
            for (;;) {
                const double rS = rnd(g); MRG(rS, S0 / Stot, 18);
                if (rS < S0 / Stot) { beta = beta0 * sqrt(rnd(g)); Cb = beta / a; }
                else                    { beta = beta0 + one_b0 * rnd(g); Cb = h2; }
                pb = sqrt(-2.0 * log(beta));
                t2 = atan((pb - x0) / a); t1 = atan((-pb - x0) / a); delt = t2 - t1;
                { const double l_ = rnd(g) * Cb, r_ = (beta / api) * delt; mrg_note(l_, r_, 15, (fabs(t1) + fabs(t2)) / fabs(delt)); if (l_ < r_) break; }
            }
        } else {
            double dbeta = sqrt(two_over_PI * a * (1.0 - beta0) * beta0 * x0), beta1 = beta0 + dbeta, one_b1 = 1.0 - beta1;
            double pb1 = sqrt(-2.0 * log(beta1)), h1 = two_over_PI * beta1 * pb1 / (x0sq - pb1 * pb1);
            double hmax = (h1 > h2) ? h1 : h2;
            double S0 = beta0 * h0, S1 = dbeta * h0, S2 = one_b1 * hmax, Stot = S0 + S1 + S2;
            for (;;) {
                double rS = rnd(g); MRG(rS, S0 / Stot, 13); MRG(rS, 1.0 - S2 / Stot, 14);
                if (rS < S0 / Stot)              { beta = beta0 * sqrt(rnd(g)); Cb = beta / a; }
                else if (rS < 1.0 - S2 / Stot)   { beta = beta0 + dbeta * rnd(g); Cb = h0; }
                else                             { beta = beta1 + one_b1 * rnd(g); Cb = hmax; }
                pb = sqrt(-2.0 * log(beta));
                t2 = atan((pb - x0) / a); t1 = atan((-pb - x0) / a); delt = t2 - t1;
                { const double l_ = rnd(g) * Cb, r_ = (beta / api) * delt; mrg_note(l_, r_, 15, (fabs(t1) + fabs(t2)) / fabs(delt)); if (l_ < r_) break; }
            }
        }
    }
    vz = x0 + a * tan(delt * rnd(g) + t1);
    return (x0in < 0.0) ? -vz : vz;
}
/* rand_resonance: random_mt.f90:2974-2993 */
static double rand_resonance(rng_t *g, double E1) {
    if (E1 > 0.0) {
        double p2 = sqrt((4.0 - E1) / (3.0 * E1));
        double Q = (4.0 * rnd(g) - 2.0) / (E1 * (p2 * p2 * p2));
        double W = pow(Q + sqrt(Q * Q + 1.0), 1.0 / 3.0);
        return p2 * (W - 1.0 / W);
    } else if (E1 < 0.0) {
        double p2 = sqrt(fabs((4.0 - E1) / (3.0 * E1)));
        double Q = (4.0 * rnd(g) - 2.0) / (E1 * (p2 * p2 * p2));
        return 2.0 * p2 * cos((acos(Q) + FOURPI) / 3.0);
    }
    return 2.0 * rnd(g) - 1.0;
}
/* rand_henyey_greenstein: random_mt.f90:3022-3042 */
static double rand_hg(rng_t *g, double gg) {
    double x = rnd(g);
    if (gg == 0.0) return 2.0 * x - 1.0;
    double g2 = gg * gg, twog = 2.0 * gg, t = (1.0 - g2) / (1.0 - gg + twog * x);
    return ((1.0 + g2) - t * t) / twog;
}
/* rand_voigt: random_mt.f90:3062-3083  (a*tan(pi*xi - pi/2) + gauss/sqrt2; Cauchy drawn first) */
static double rand_voigt(rng_t *g, double a) {
    double c = tan(PI * rnd(g) - HALFPI);
    return a * c + rand_gauss(g) * (1.0 / sqrt(2.0));
}

/* =====================================================================================
 * Simulation state
 * ===================================================================================== */
typedef struct {
    double x, y, z, rm[9];            /* rmatrix row-major rm[3*r+c] = rmatrix(r+1,c+1) */
    double *scatt, *direc, *I, *Q, *U, *V;
    double *scatt_2D, *direc_2D, *I_2D, *Q_2D, *U_2D, *V_2D;
    double *direc0, *direc0_2D, *scatt_dust;      /* par%save_direc0, par%save_dust_scattered */
    double *peel_Ha;                              /* line type 8: observer%peel_Ha(nxfreq_Ha, nxim, nyim) (define.f90:567-568) */
} obs_t;

typedef struct {
    double id; double nscatt_gas, nscatt_dust;
    double x, y, z, kx, ky, kz, mx, my, mz, nx, ny, nz;
    int icell, jcell, kcell;
    int icell_clump;                  /* 0 = vacuum, > 0 = owner clump (define.f90:90) */
    double vfy_shear;                 /* y velocity picked up at the shearing x boundary (define.f90:100) */
    int iband;                        /* 1 = the line's own band, 2 = H-alpha after a Ly-beta conversion (define.f90:96); 0 in a zeroed packet counts as 1 */
    int species;                      /* line%selected_species_HD of the last resonance scattering: 1 H, 2 D */
    double xfreq, xfreq_ref, wgt;
    int inside;
    double I, Q, U, V, E1, E2, E3;
} photon_t;

/* amr_grid_type (octree_mod.f90:19-138): flat octree, 1-based cell and leaf indices stored as in the reference */
typedef struct {
    int ncells, nleaf, levelmax;
    int *parent, *children, *level, *ileaf, *icell_of_leaf, *neighbor;   /* children(8,ncells), neighbor(6,ncells) column-major */
    double *cx, *cy, *cz, *ch;
    double xmin, xmax, ymin, ymax, zmin, zmax, L_box;
    int nr_JPa, nz_JPa; double rmax_JPa, dr_JPa, dz_JPa;
} amr_t;

/* clump population and CSR acceleration grid: clump_mod.f90:27-118 */
typedef struct {
    int64_t N; double sphere_R, r_min, radius_max;
    double rhokap_ref, voigt_a_ref, Dfreq_ref, vtherm_ref, temperature_ref;
    double *x, *y, *z, *vx, *vy, *vz, *radius, *radius2, *rhokap, *rhokapD, *voigt_a, *Dfreq, *vtherm;
    int cgx; double cg_min, cg_d, cg_inv; int32_t *cg_start, *cg_list; int64_t nreg;
    double fvol, fcov; int rsa_grid; double rsa_acceptance;
    int has_overlap;                                         /* clump_mod.f90:88, set by check_has_overlap */
} clumps_t;

struct orc_sim {
    int use_amr_grid; char amr_file[512]; amr_t amr;
    int use_clump_medium, clump_fully_inside, clump_allow_overlap; clumps_t cl;
    double cone_opening;                                    /* par%cone_opening [deg]: bicone about the z axis, 0 = everything */
    double clump_radius, clump_N_clumps, clump_f_vol, clump_f_cov, clump_tau0, clump_NHI, clump_nH, clump_temperature, clump_sigma_v;
    int HeI_coherent;                        /* par%HeI_coherent (line type 6) */
    double bturb;                            /* par%bturb [km/s], <= 0: none */
    double wavelength_min, wavelength_max;   /* angstrom: the spectral window, takes precedence over velocity_min/max (grid_mod_car.f90:1457-1461) */
    double Vrot, rinner, velocity_min, velocity_max, Vpeak, rpeak, DeltaV, velocity_alpha;
    /* --- par% (define.f90:209-544) --- */
    double no_photons; int iseed;
    double temperature, taumax, tauhomo, N_gasmax, N_gashomo, Vexp;
    int velocity_type;               /* 0 none, 1 hubble, 2 constant_radial, 3 parallel_velocity */
    double Vx, Vy, Vz;
    int comoving_source, recoil, core_skip, core_skip_global, xyz_symmetry, xy_periodic, xy_symmetry;
    int nx, ny, nz; double xmax, ymax, zmax, rmax, rmin;
    double xs_point, ys_point, zs_point, source_rmax; int source_geometry;   /* 0 point, 1 uniform_sphere, 2 uniform_cylinder, 3 uniform, 4 uniform_xy, 5 gaussian */
    double source_zscale;
    double xfreq0, par_xfreq_min, par_xfreq_max; int nxfreq;
    int spectral_type;               /* 0 monochromatic, 1 voigt, 2 continuum, 3 gaussian */
    double gaussian_sigma_vel, gaussian_FWHM_vel;
    int use_stokes, use_reduced_wgt, save_Jin, save_Jabs, save_Jmu, nmu;
    double mu_min, dmu;
    double hgg, albedo, cext_dust, DGR, distance2cm;
    double Omega;                    /* par%Omega (define.f90:308-312) as grid_create leaves it without a dens_file: added to vfy per x wrap (raytrace_car.f90:2843-2852) */
    int save_peeloff, save_peeloff_2D, save_peeloff_3D, save_direc0, save_dust_scattered, nobs, nxim, nyim;
    double distance, alpha[MAX_OBS], beta[MAX_OBS], gamma[MAX_OBS], obsx[MAX_OBS], obsy[MAX_OBS], obsz[MAX_OBS], dxim, dyim;
    char scatt_mat_file[512];
    int geometry;                    /* 0 sphere, 1 rectangle */
    double atau3;
    /* --- line (line_mod.f90:1241-1270) --- */
    double cross0, vtherm1, damping, wavelength0, g_recoil0, E1, E2, E3;
    /* line types 4, 5, 6 (resonance + fluorescence multiplets, line%b(:) of define.f90:639-680; set by setup_resonance_line, line_mod.f90:713-1153) */
    struct { int nup, ndown[3], A_down[3][4]; double f12[3], delE_Hz[3], damping_up[3], A21[3][4], Elow_Hz[3][4], E1[3][4], E2[3][4], E3[3][4], P_down[3][4]; } ml;
    int fine_structure, line_type; double DnuHK_Hz; char line_id[24];      /* par%line_id: '' = ly_alpha */
    int include_deuterium; double D_to_H_ratio, nD_over_nH, delta_nu_HD_Hz, ratio_Dfreq_HD, ratio_voigta_HD, g_recoil0_D;   /* line_type 7: H + D (line_mod.f90:1155-1191) */     /* line_type 2: Ly-alpha doublet 2P1/2 (H) + 2P3/2 (K), line_mod.f90:1241-1270 */
    /* --- grid (define.f90:117-206) --- */
    int i0, j0, k0;
    double xmin, ymin, zmin, dx, dy, dz, xrange, yrange, zrange;
    double *xface, *yface, *zface;
    double *rhokap, *voigt_a, *Dfreq, *vfx, *vfy, *vfz, *rhokapD;
    double Dfreq_ref, voigt_amean, xfreq_min, xfreq_max, dxfreq, xcrit, xcrit2;
    double *Jout, *Jin, *Jabs, *Jmu;
    /* --- Mueller table (define.f90:616-625) --- */
    int nPDF; double *coss, *S11, *S12, *S33, *S34, *phase_PDF; int *alias;
    obs_t obs[MAX_OBS];
    /* --- run results --- */
    double nscatt_gas, nscatt_dust; int64_t nphotons_done;
    double n_cellsteps, n_cellsteps_peel, n_events;
    int raytrace_variant;            /* 0 car, 1 xyzsym, 2 zonly */
    /* --- CALCJ / CALCP / CALCPnew profile tallies (compile-time switches in the reference, Makefile:33-61) --- */
    int calc_J, calc_P, calc_Pnew, geometry_JPa, nbins;
    int *ind_bin; double *ncount, *J1, *P1, *P1_new;
    /* --- HEALPix inside observers (observer_heal.f90) --- */
    int nside, npix;
    int64_t log_n; double *log_nscatt, *log_xfreq;
    int save_all_photons; double *allph;     /* par%save_all_photons: 10 columns x log_n (r, xfreq1, xfreq2, nscatt_gas, nscatt_dust, I, Q, U, V, r0) */
    /* --- line type 8, 'ly_beta' (line_mod.f90:1192-1240, define.f90:461-475): band-2 (H-alpha) dust constants and spectral axis, the conversion
     *     branch of the 3p level, the weight budget W_conv, W_esc1, W_abs1, W_esc2, W_abs2 --- */
    double cext_dust_Ha, albedo_Ha, hgg_Ha, R_Ha, par_xfreq_max_Ha; int par_nxfreq_Ha;
    int nxfreq_Ha; double xfreq_min_Ha, xfreq_max_Ha, dxfreq_Ha, P_conv, E_conv[3];
    double *Jout_Ha, *Jabs_Ha, *Pconv, W[5];
    /* --- H2 lines next to Ly-alpha (h2_mod.f90; par%h2_model = 'neufeld'): per line dnu_Hz, strength, a_damp, p_scat; W_H2abs, W_H2scat, W_H2pump(:) --- */
    char h2_model[16], h2_data_dir[512]; double f_H2, h2_temperature; int h2_pure_absorption, h2_hi_width;
    int h2_on, h2_nlines; double h2_Dfreq_Hz, h2_dnu_Hz[4], h2_strength[4], h2_a_damp[4], h2_p_scat[4], h2_pop[4], h2_f_osc[4], W_H2[6];
};

static int grid_create_clump(orc_sim *s);
static int grid_create_amr(orc_sim *s);
static int create_JPa_amr_mem(orc_sim *s);
static int amr_JPa_bin(const orc_sim *s, int il, double x0, double y0, double z0);
static void amr_xcrit_local(const orc_sim *s, int il, double x, double y, double z, double *xc, double *xc2);
/* calc_voigt1 / calc_voigt2 (line_mod.f90:38-62): the line profile of a cell with Voigt parameter a and Doppler width Dfreq */
static inline double calc_voigt(const orc_sim *s, double xfreq, double a, double Dfreq) {
    if (s->line_type == 2) { double DnuHK = s->DnuHK_Hz / Dfreq;
        return orc_voigt(xfreq + DnuHK, a) * (1.0 / 3.0) + orc_voigt(xfreq, a) * (2.0 / 3.0); }
    if (s->line_type == 7) {                                /* calc_voigt_HD :82-106: H profile + the D line seen from the H frame */
        double dx_HD = s->delta_nu_HD_Hz / Dfreq, xfreq_D = (xfreq - dx_HD) * s->ratio_Dfreq_HD;
        return orc_voigt(xfreq, a) + s->nD_over_nH * s->ratio_Dfreq_HD * orc_voigt(xfreq_D, a * s->ratio_voigta_HD); }
    if (s->line_type == 5 || s->line_type == 6) {           /* calc_voigt3 :64-80 (octree_mod.f90:329-347, clump_mod.f90:151-158 alike) */
        double v = orc_voigt(xfreq, a);
        for (int i = 2; i <= s->ml.nup; i++) {
            double Dnu = s->ml.delE_Hz[i-1] / Dfreq, a_ratio = s->ml.damping_up[i-1] / s->ml.damping_up[0], f_ratio = s->ml.f12[i-1] / s->ml.f12[0];
            v = v + orc_voigt(xfreq + Dnu, a * a_ratio) * f_ratio; }
        return v; }
    return orc_voigt(xfreq, a);                             /* types 1, 4: calc_voigt1 */
}
#define IDX(s,i,j,k) ((size_t)((i)-1) + (size_t)(s)->nx * ((size_t)((j)-1) + (size_t)(s)->ny * (size_t)((k)-1)))

/* per-thread accumulators, merged after the run */
typedef struct { orc_sim *s; double *Jout, *Jin, *Jabs, *Jmu, *J1, *P1, *P1_new; obs_t obs[MAX_OBS];
                 double *Jout_Ha, *Jabs_Ha, *Pconv, W[5], W_H2[6];
                 double nscatt_gas, nscatt_dust, n_cellsteps, n_cellsteps_peel, n_events;
                 rng_t *g; } acc_t;                          /* the packet's stream, for the one draw inside the overlap march (sample_owner_clump) */

/* =====================================================================================
 * Parameter parsing + read_input post-processing (setup.f90:42-562)
 * ===================================================================================== */
static int parse_bool(const char *v) { return (v[0] == '1' || v[0] == 't' || v[0] == 'T' || !strncmp(v, ".t", 2) || !strncmp(v, ".T", 2)); }

static void set_defaults(orc_sim *s) {
    memset(s, 0, sizeof *s);
    s->no_photons = 1e5; s->iseed = 0; s->temperature = 1e4;
    s->taumax = -999.0; s->tauhomo = -999.0; s->N_gasmax = -999.0; s->N_gashomo = -999.0;
    s->comoving_source = 1; s->nx = 1; s->ny = 1; s->nz = 11; s->xmax = s->ymax = s->zmax = 1.0;
    s->rmin = -999.0; s->rmax = -999.0; s->source_rmax = -999.0;
    s->par_xfreq_min = NAN; s->par_xfreq_max = NAN; s->nxfreq = 121; s->spectral_type = 1;
    s->save_Jin = 1; s->save_Jabs = 1; s->nmu = 11; s->save_peeloff_3D = 1;
    /* float32 literals widened to double (define.f90:454-456) */
    s->hgg = (double)0.6761f; s->albedo = (double)0.3253f; s->cext_dust = (double)1.6059e-21f;
    s->DGR = 0.0; s->distance2cm = -999.9; s->distance = NAN; s->dxim = NAN; s->dyim = NAN;
    for (int i = 0; i < MAX_OBS; i++) s->alpha[i] = s->beta[i] = s->gamma[i] = s->obsx[i] = s->obsy[i] = s->obsz[i] = NAN;
    s->geometry = 0; s->geometry_JPa = 2147483647;
    s->clump_radius = s->clump_N_clumps = s->clump_f_vol = s->clump_f_cov = s->clump_tau0 = s->clump_NHI = s->clump_nH = s->clump_temperature = -1.0;
    s->clump_fully_inside = 1; s->velocity_min = NAN; s->velocity_max = NAN; s->wavelength_min = NAN; s->wavelength_max = NAN; s->velocity_alpha = 1.0;
    s->D_to_H_ratio = 1.5e-5;
    s->gaussian_sigma_vel = 12.843374; s->gaussian_FWHM_vel = -999.0;     /* define.f90:317-318 */
    s->cext_dust_Ha = 3.801e-22; s->albedo_Ha = 0.6741; s->hgg_Ha = 0.4967;    /* define.f90:464-466 (double precision literals) */
    strcpy(s->h2_model, "none"); s->h2_temperature = 1000.0;               /* define.f90:480-485 */
}

/* the single resonance lines (type 1) and S1/2-P1/2,3/2 doublets (type 2) of setup_resonance_line, line_mod.f90:590-713:
   wavelength0 [um], f12(1), damping, the mass of vtherm1, the mass of g_recoil0, and the two level wavenumbers [cm^-1] of DnuHK_Hz.
   Two quirks of the reference are restated as they are: CIV_1548 takes g_recoil0 from line%mass_amu, which still holds its default
   1.00797 (define.f90:646; line_mod.f90:605), and AlII_1671's mass is a default-kind (single precision) literal (:706). */
typedef struct { const char *id; int type; double wl, f1, damping, mass, mass_recoil, k_hi, k_lo; } metal_line;
static const metal_line METAL_LINES[] = {
    {"CIV_1548",  2, 0.1548187,    0.190, 2.647e8,    12.011,   1.00797,  64591.7,     64484.0},
    {"NV_1239",   2, 0.1238821,    0.156, 3.390e8,    14.0067,  14.0067,  80721.9,     80463.2},
    {"OVI_1032",  2, 0.1031912,    0.133, 4.137e8,    15.9994,  15.9994,  96907.5,     96375.0},
    {"NaI_D",     2, 0.5891583253, 0.641, 6.153e7,    22.98977, 22.98977, 16973.36619, 16956.17025},
    {"CaII_HK",   2, 0.3934777,    0.682, 1.446667e8, 40.078,   40.078,   25414.40,    25191.51},
    {"MgII_2796", 2, 0.2796352,    0.608, 2.590e8,    24.305,   24.305,   35760.88,    35669.31},
    {"SiIV_1394", 2, 0.1393755,    0.513, 8.743e8,    28.0855,  28.0855,  71748.64,    71287.54},
    {"AlII_1671", 1, 0.16707874,   1.77,  1.41e9,     (double)26.98154f, (double)26.98154f, 0.0, 0.0},
};
typedef struct { const char *id, *id2, *ion; int type; double wl, mass; int nup; double f12[3], level_cm[3]; int ndown[3];
                 double A21[3][4], Elow_cm[3][4], E1[3][4], E2[3][4], E3[3][4]; } multiplet_row;
static const multiplet_row MULTIPLETS[] = {
#include "line_table.inc"
};
static const multiplet_row *find_multiplet(const char *id) {
    for (size_t i = 0; i < sizeof MULTIPLETS / sizeof MULTIPLETS[0]; i++)
        if (!strcmp(id, MULTIPLETS[i].id) || (MULTIPLETS[i].id2[0] && !strcmp(id, MULTIPLETS[i].id2))) return &MULTIPLETS[i];
    return NULL; }
static const metal_line *find_metal_line(const char *id) {
    for (size_t i = 0; i < sizeof METAL_LINES / sizeof METAL_LINES[0]; i++) if (!strcmp(id, METAL_LINES[i].id)) return &METAL_LINES[i];
    return NULL; }

static int set_param(orc_sim *s, const char *k, const char *v) {
#define D(name) if (!strcmp(k, #name)) { s->name = atof(v); return 0; }
#define I(name) if (!strcmp(k, #name)) { s->name = atoi(v); return 0; }
#define B(name) if (!strcmp(k, #name)) { s->name = parse_bool(v); return 0; }
    D(no_photons) I(iseed) D(temperature) D(taumax) D(tauhomo) D(Vexp) D(Vx) D(Vy) D(Vz)
    B(comoving_source) B(recoil) B(core_skip) B(core_skip_global) B(xyz_symmetry) B(xy_periodic) B(xy_symmetry) B(fine_structure) B(include_deuterium) B(HeI_coherent) B(save_all_photons)
    I(nx) I(ny) I(nz) D(xmax) D(ymax) D(zmax) D(rmax) D(rmin) D(xs_point) D(ys_point) D(zs_point) D(source_rmax)
    D(xfreq0) I(nxfreq) B(use_stokes) B(use_reduced_wgt) B(save_Jin) B(save_Jabs) B(save_Jmu) I(nmu)
    D(hgg) D(albedo) D(cext_dust) D(DGR) D(distance2cm) D(Omega) B(save_peeloff) B(save_peeloff_2D) B(save_peeloff_3D) B(save_direc0) B(save_dust_scattered)
    I(nxim) I(nyim) D(distance) D(dxim) D(dyim) B(calc_J) B(calc_P) B(calc_Pnew) I(geometry_JPa) I(nside)
    if (!strcmp(k, "xfreq_min")) { s->par_xfreq_min = atof(v); return 0; }
    if (!strcmp(k, "xfreq_max")) { s->par_xfreq_max = atof(v); return 0; }
    if (!strcmp(k, "N_HImax") || !strcmp(k, "N_gasmax") || !strcmp(k, "N_HI")) { s->N_gasmax = atof(v); return 0; }
    if (!strcmp(k, "N_HIhomo") || !strcmp(k, "N_gashomo")) { s->N_gashomo = atof(v); return 0; }
    if (!strcmp(k, "tau0")) { s->taumax = atof(v); return 0; }
    B(use_clump_medium) B(clump_fully_inside) B(clump_allow_overlap) D(cone_opening) D(clump_radius) D(clump_N_clumps) D(clump_f_vol) D(clump_f_cov) D(clump_tau0) D(clump_NHI) D(clump_nH)
    D(clump_temperature) D(clump_sigma_v) D(Vrot) D(rinner) D(velocity_min) D(velocity_max) D(Vpeak) D(rpeak) D(DeltaV) D(velocity_alpha) D(gaussian_sigma_vel) D(gaussian_FWHM_vel) D(source_zscale) D(D_to_H_ratio) D(wavelength_min) D(wavelength_max) D(bturb)
    if (!strcmp(k, "use_amr_grid")) { s->use_amr_grid = parse_bool(v); return 0; }
    if (!strcmp(k, "amr_file")) { snprintf(s->amr_file, sizeof s->amr_file, "%s", v); return 0; }
    if (!strcmp(k, "amr_type")) return 0;
    if (!strcmp(k, "scatt_mat_file")) { snprintf(s->scatt_mat_file, sizeof s->scatt_mat_file, "%s", v); return 0; }
    if (!strcmp(k, "nxfreq_Ha")) { s->par_nxfreq_Ha = atoi(v); return 0; }
    if (!strcmp(k, "xfreq_max_Ha")) { s->par_xfreq_max_Ha = atof(v); return 0; }
    D(cext_dust_Ha) D(albedo_Ha) D(hgg_Ha) D(f_H2) D(h2_temperature) B(h2_pure_absorption) B(h2_hi_width)
    if (!strcmp(k, "h2_model")) { snprintf(s->h2_model, sizeof s->h2_model, "%s", v); for (char *q = s->h2_model; *q; q++) if (*q >= 'A' && *q <= 'Z') *q += 32; return 0; }
    if (!strcmp(k, "h2_data_dir")) { snprintf(s->h2_data_dir, sizeof s->h2_data_dir, "%s", v); return 0; }
    if (!strcmp(k, "line_id")) { if (!strcmp(v, "ly_alpha")) { s->line_id[0] = 0; return 0; }
        if (!strcmp(v, "ly_beta")) { snprintf(s->line_id, sizeof s->line_id, "%s", v); return 0; } if (!strcmp(v, "ly_alpha_HD")) { s->include_deuterium = 1; return 0; }
        if (find_metal_line(v) || find_multiplet(v)) { snprintf(s->line_id, sizeof s->line_id, "%s", v); return 0; }
        snprintf(g_err, sizeof g_err, "line_id '%s' not supported by the oracle", v); return -1; }
    if (!strcmp(k, "spectral_type")) {
        if (!strcmp(v, "voigt")) s->spectral_type = 1; else if (!strcmp(v, "monochromatic")) s->spectral_type = 0;
        else if (!strcmp(v, "continuum")) s->spectral_type = 2; else if (!strcmp(v, "gaussian")) s->spectral_type = 3;
        else { snprintf(g_err, sizeof g_err, "spectral_type '%s' not supported by the oracle", v); return -1; }
        return 0; }
    if (!strcmp(k, "source_geometry")) {
        if (!strcmp(v, "point")) s->source_geometry = 0; else if (!strcmp(v, "uniform_sphere") || !strcmp(v, "sphere")) s->source_geometry = 1;
        else if (!strcmp(v, "uniform_cylinder") || !strcmp(v, "cylinder")) s->source_geometry = 2; else if (!strcmp(v, "uniform")) s->source_geometry = 3;
        else if (!strcmp(v, "uniform_xy")) s->source_geometry = 4; else if (!strcmp(v, "gaussian")) s->source_geometry = 5;
        else { snprintf(g_err, sizeof g_err, "source_geometry '%s' not supported by the oracle", v); return -1; }
        return 0; }
    if (!strcmp(k, "velocity_type")) {
        if (!strcmp(v, "hubble")) s->velocity_type = 1; else if (!strcmp(v, "constant_radial")) s->velocity_type = 2;
        else if (!strcmp(v, "parallel_velocity")) s->velocity_type = 3; else if (!v[0]) s->velocity_type = 0;
        else if (!strcmp(v, "rotating_solid_body")) s->velocity_type = 4; else if (!strcmp(v, "rotating_galaxy_halo")) s->velocity_type = 5;
        else if (!strcmp(v, "power_law")) s->velocity_type = 6; else if (!strcmp(v, "linear_decelerate")) s->velocity_type = 7; else if (!strcmp(v, "ssh")) s->velocity_type = 8;   /* Cartesian grid only */
        else { snprintf(g_err, sizeof g_err, "velocity_type '%s' not supported by the oracle", v); return -1; }
        return 0; }
    if (!strcmp(k, "geometry")) {
        if (!strcmp(v, "sphere") || !v[0]) s->geometry = 0; else if (!strcmp(v, "rectangle") || !strcmp(v, "box")) s->geometry = 1;
        else { snprintf(g_err, sizeof g_err, "geometry '%s' not supported by the oracle", v); return -1; }
        return 0; }
    /* observer arrays: alpha(1)=..  written as alpha1=..; bare name sets element 1 */
    { const char *names[6] = {"alpha", "beta", "gamma", "obsx", "obsy", "obsz"};
      double *arrs[6] = {s->alpha, s->beta, s->gamma, s->obsx, s->obsy, s->obsz};
      for (int a = 0; a < 6; a++) { size_t L = strlen(names[a]);
        if (!strncmp(k, names[a], L)) { int idx = k[L] ? atoi(k + L) : 1;
            if (idx >= 1 && idx <= MAX_OBS && (k[L] == 0 || (k[L] >= '0' && k[L] <= '9'))) { arrs[a][idx-1] = atof(v); return 0; } } } }
    snprintf(g_err, sizeof g_err, "unknown parameter '%s' (unknown names fail the namelist read in the reference too)", k);
    return -1;
#undef D
#undef I
#undef B
}

/* line constants, Ly-alpha default branch: line_mod.f90:1241-1270 */
static void alias_setup(int n, double *probs, int *alias);
static void setup_line(orc_sim *s) {
    const multiplet_row *mr = s->line_id[0] ? find_multiplet(s->line_id) : NULL;
    if (mr) {                                               /* line types 4, 5, 6: line_mod.f90:713-1153 */
        const double sigma_0 = 0.026540083434, amu = 1.67262192e-24, vtherm1_amu = 0.12895319011972164, speedc_cm = 2.99792458e5 * 1.0e5;
        double wl = mr->wl * UM2M;
        memset(&s->ml, 0, sizeof s->ml);
        s->line_type = mr->type; s->wavelength0 = mr->wl; s->cross0 = sigma_0 / sqrt(PI) * mr->f12[0];
        s->vtherm1 = vtherm1_amu / sqrt(mr->mass); s->g_recoil0 = (H_PLANCK / amu / mr->mass) / (wl * wl);
        s->ml.nup = mr->nup;
        for (int i = 0; i < mr->nup; i++) {
            s->ml.f12[i] = mr->f12[i]; s->ml.ndown[i] = mr->ndown[i];
            s->ml.delE_Hz[i] = mr->nup > 1 ? mr->level_cm[0] * speedc_cm - mr->level_cm[i] * speedc_cm : 0.0;       /* e.g. :833-834 */
            double sum = 0.0;
            for (int d = 0; d < mr->ndown[i]; d++) { s->ml.A21[i][d] = mr->A21[i][d]; sum += mr->A21[i][d]; s->ml.Elow_Hz[i][d] = mr->Elow_cm[i][d] * speedc_cm;
                s->ml.E1[i][d] = mr->E1[i][d]; s->ml.E2[i][d] = mr->E2[i][d]; s->ml.E3[i][d] = mr->E3[i][d]; }
            s->ml.damping_up[i] = sum;
            if (mr->ndown[i] > 1) { for (int d = 0; d < mr->ndown[i]; d++) s->ml.P_down[i][d] = mr->A21[i][d] / sum;
                                    alias_setup(mr->ndown[i], s->ml.P_down[i], s->ml.A_down[i]); }
        }
        s->damping = s->ml.damping_up[0]; s->DnuHK_Hz = 0.0;
        s->E1 = 1.0; s->E2 = 0.0; s->E3 = 1.0; s->fine_structure = 0; s->include_deuterium = 0;
        return;
    }
    if (!strcmp(s->line_id, "ly_beta")) {                   /* :1192-1240: H I Lyman beta, 3p -> 1s re-emission or 3p -> 2s conversion to H-alpha (+ two photons) */
        const double sigma_0 = 0.026540083434, amu = 1.67262192e-24, vtherm1_amu = 0.12895319011972164, mass_amu = 1.00797;
        s->line_type = 8; s->wavelength0 = 0.10257222; s->damping = 1.8970e8; s->cross0 = sigma_0 / sqrt(PI) * 0.07910;
        s->vtherm1 = vtherm1_amu / sqrt(mass_amu);
        { double wl = s->wavelength0 * UM2M; s->g_recoil0 = (H_PLANCK / amu / mass_amu) / (wl * wl); }
        s->DnuHK_Hz = 0.0; s->E1 = 1.0; s->E2 = 0.0; s->E3 = 1.0; s->fine_structure = 0; s->include_deuterium = 0;
        { const double A21[2] = {1.6725e8, 2.2448e7}; s->P_conv = A21[1] / (A21[0] + A21[1]); }      /* P_down = A21 / sum(A21), :1233-1236 */
        s->E_conv[0] = 1.0; s->E_conv[1] = 0.0; s->E_conv[2] = 1.0;                                  /* both channels are s -> p -> s ladders */
        return;
    }
    if (s->line_id[0]) {                                    /* :590-713; par%fine_structure plays no part for these ids */
        const double sigma_0 = 0.026540083434, amu = 1.67262192e-24, vtherm1_amu = 0.12895319011972164, speedc_cm = 2.99792458e5 * 1.0e5;
        const metal_line *m = find_metal_line(s->line_id);
        double wl = m->wl * UM2M;
        s->line_type = m->type; s->wavelength0 = m->wl; s->damping = m->damping; s->cross0 = sigma_0 / sqrt(PI) * m->f1;
        s->vtherm1 = vtherm1_amu / sqrt(m->mass); s->g_recoil0 = (H_PLANCK / amu / m->mass_recoil) / (wl * wl);
        s->DnuHK_Hz = m->type == 2 ? speedc_cm * (m->k_hi - m->k_lo) : 0.0;
        s->E1 = 1.0; s->E2 = 0.0; s->E3 = 1.0; s->fine_structure = 0; s->include_deuterium = 0;
        return;
    }
    const double sigma_0 = 0.026540083434, amu = 1.67262192e-24, vtherm1_amu = 0.12895319011972164, mass_amu = 1.00797;
    s->E1 = 1.0; s->E2 = 0.0; s->E3 = 1.0; s->line_type = 1; s->DnuHK_Hz = 0.0;
    if (s->fine_structure) {                                /* :1250-1253: delE_Hz(2) = (82259.2850014 - 82258.9191133) cm^-1 * c */
        const double speedc_cm = 2.99792458e5 * 1.0e5;
        double d1 = 82259.2850014 * speedc_cm, d2 = 82258.9191133 * speedc_cm;
        s->line_type = 2; s->DnuHK_Hz = d1 - d2;
    }
    s->wavelength0 = 0.1215668237310;
    s->damping = 6.2649e8;
    s->cross0 = sigma_0 / sqrt(PI) * (0.27760 + 0.13881 + 0.0);
    s->vtherm1 = vtherm1_amu / sqrt(mass_amu);
    double wl = s->wavelength0 * UM2M;
    s->g_recoil0 = (H_PLANCK / amu / mass_amu) / (wl * wl);
    if (s->include_deuterium) {                             /* 'ly_alpha_HD' (:580-588 promotion, fine structure switched off; :1155-1191) */
        const double speedc_cm = 2.99792458e5 * 1.0e5, um2m = UM2M;
        const double wl_D = 0.1215337431, mass_D = 2.01410177812, damping_D = 6.2649e8;
        s->fine_structure = 0; s->line_type = 7; s->DnuHK_Hz = 0.0;
        s->nD_over_nH = s->D_to_H_ratio;
        s->delta_nu_HD_Hz = speedc_cm * (1.0 / (wl_D * um2m * 1.0e2) - 1.0 / (s->wavelength0 * um2m * 1.0e2));
        s->ratio_Dfreq_HD = (wl_D / s->wavelength0) * sqrt(mass_D / mass_amu);
        s->ratio_voigta_HD = (damping_D / s->damping) * s->ratio_Dfreq_HD;
        s->g_recoil0_D = (H_PLANCK / amu / mass_D) / ((wl_D * um2m) * (wl_D * um2m));
    }
}
void orc_line_constants(double *cross0, double *vtherm1, double *damping, double *wavelength0, double *g_recoil0) {
    orc_sim t; t.line_id[0] = 0; t.fine_structure = 0; t.include_deuterium = 0; setup_line(&t);
    *cross0 = t.cross0; *vtherm1 = t.vtherm1; *damping = t.damping; *wavelength0 = t.wavelength0; *g_recoil0 = t.g_recoil0;
}
/* vtherm_total: define.f90:928-933: thermal width and the turbulent b parameter par%bturb in quadrature */
static double vtherm_total(const orc_sim *s, double T) { double vt = s->vtherm1 * sqrt(T); if (s->bturb > 0.0) vt = sqrt(vt * vt + s->bturb * s->bturb); return vt; }

/* random_alias_setup64: random_mt.f90:2016-2067 */
static void alias_setup(int n, double *probs, int *alias) {
    int *small = calloc(n, sizeof(int)), *large = calloc(n, sizeof(int)), ns = 0, nl = 0;
    for (int i = 1; i <= n; i++) { alias[i-1] = 0; probs[i-1] = n * probs[i-1];
        if (probs[i-1] < 1.0) small[ns++] = i; else large[nl++] = i; }
    while (ns > 0 && nl > 0) {
        int ks = small[--ns], kl = large[--nl];
        alias[ks-1] = kl;
        probs[kl-1] = probs[kl-1] + probs[ks-1] - 1.0;
        if (probs[kl-1] < 1.0) small[ns++] = kl; else large[nl++] = kl;
    }
    free(small); free(large);
}
/* setup_scattering_matrix: setup.f90:581-649 */
static int setup_mueller(orc_sim *s) {
    FILE *f = fopen(s->scatt_mat_file, "r");
    if (!f) { snprintf(g_err, sizeof g_err, "cannot open scatt_mat_file '%s'", s->scatt_mat_file); return -1; }
    char line[512]; double wl, cext, alb, hgg; int n;
    if (!fgets(line, sizeof line, f) || !fgets(line, sizeof line, f) || sscanf(line, "%lf %lf %lf %lf %d", &wl, &cext, &alb, &hgg, &n) != 5)
        { fclose(f); snprintf(g_err, sizeof g_err, "bad Mueller header"); return -1; }
    s->albedo = alb; s->hgg = hgg; s->cext_dust = cext; s->nPDF = n;
    if (!fgets(line, sizeof line, f)) { fclose(f); return -1; }
    s->coss = malloc(n * sizeof(double)); s->S11 = malloc(n * sizeof(double)); s->S12 = malloc(n * sizeof(double));
    s->S33 = malloc(n * sizeof(double)); s->S34 = malloc(n * sizeof(double));
    s->phase_PDF = malloc((n - 1) * sizeof(double)); s->alias = malloc((n - 1) * sizeof(int));
    for (int i = 0; i < n; i++)
        if (fscanf(f, "%lf %lf %lf %lf %lf", &s->coss[i], &s->S11[i], &s->S12[i], &s->S33[i], &s->S34[i]) != 5)
            { fclose(f); snprintf(g_err, sizeof g_err, "bad Mueller row %d", i); return -1; }
    fclose(f);
    double norm = 0.0;                                           /* calc_Integral: mathlib.f90:218-240 */
    for (int i = 1; i < n; i++) norm = norm + 0.5 * (s->S11[i] + s->S11[i-1]) * fabs(s->coss[i] - s->coss[i-1]);
    for (int i = 0; i < n; i++) { s->S11[i] /= norm; s->S12[i] /= norm; s->S33[i] /= norm; s->S34[i] /= norm; }
    double sum = 0.0;
    for (int i = 0; i < n - 1; i++) { s->phase_PDF[i] = (s->S11[i] + s->S11[i+1]) / 2.0; sum += s->phase_PDF[i]; }
    for (int i = 0; i < n - 1; i++) s->phase_PDF[i] /= sum;
    alias_setup(n - 1, s->phase_PDF, s->alias);
    return 0;
}
/* interp_eq: mathlib.f90:71-106 */
static double interp_eq(int n, const double *x, const double *y, double xnew) {
    double dx = x[1] - x[0];
    int i = (int)((xnew - x[0]) / dx + 1);         /* Fortran real->integer assignment truncates toward zero */
    if (i <= 0) return y[0];
    if (i >= n) return y[n-1];
    return y[i-1] + (y[i] - y[i-1]) * (xnew - x[i-1]) / dx;
}
/* rand_alias_linear64: random_mt.f90:2196-2216 */
static double rand_alias_linear(rng_t *g, int n, const double *prob_alias, const int *alias, const double *xaxis, const double *prob) {
    int k = (int)floor(n * rnd(g)) + 1, idx;
    { const double r_ = rnd(g); MRG(r_, prob_alias[k-1], 30); if (r_ < prob_alias[k-1]) idx = k; else idx = alias[k-1]; }
    double p0 = prob[idx-1], p1 = prob[idx];
    return (sqrt(p0 * p0 + (p1 * p1 - p0 * p0) * rnd(g)) - p0) * (xaxis[idx] - xaxis[idx-1]) / (p1 - p0) + xaxis[idx-1];
}

/* h2_init: h2_mod.f90:120-222 (the two B-X lines closest to Ly-alpha, R(6) and P(5) of v = 1 <- 2; LTE populations from the X-state level table
 * <h2_data_dir>/energy_X.dat: a magic-number first line, '#' comments, rows "V J E[cm^-1]") */
static int h2_init(orc_sim *s) {
    const double speedc = 2.99792458e5, f_osc_Lya = 0.4162, hc_over_k = 1.4387769, f_osc_const = 1.4992e-16;
    const double nu_Lya_Hz = (speedc * 1.0e5) / (s->wavelength0 * UM2M * 1.0e2);
    const double mass_amu = 1.00797, vth1_H2 = s->vtherm1 * sqrt(mass_amu / (2.0 * mass_amu)), T = s->h2_temperature;
    const double bturb2 = s->bturb > 0.0 ? s->bturb * s->bturb : 0.0, vth_H2 = sqrt((vth1_H2 * sqrt(T)) * (vth1_H2 * sqrt(T)) + bturb2);
    s->h2_Dfreq_Hz = nu_Lya_Hz * vth_H2 / speedc;
    const double dv[2] = {14.140, 99.229}, lam[2] = {1215.72534, 1216.07038}, A_ul[2] = {1.36e8, 1.59e8}, A_tot[2] = {1.6825e9, 1.7199e9};
    const int vl[2] = {2, 2}, Jl[2] = {6, 5}, Ju[2] = {7, 4};
    char fn[600]; snprintf(fn, sizeof fn, "%s/energy_X.dat", s->h2_data_dir[0] ? s->h2_data_dir : "data/h2");
    FILE *f = fopen(fn, "r"); if (!f) { snprintf(g_err, sizeof g_err, "h2_init: cannot find %s", fn); return -1; }
    char line[300]; double Z = 0.0, El[2] = {-1.0, -1.0};
    while (fgets(line, sizeof line, f)) {
        char *q = line; while (*q == ' ' || *q == '\t') q++;
        if (!*q || *q == '\n' || *q == '#' || strstr(q, "//")) continue;
        double vv, jj, ee; if (sscanf(q, "%lf %lf %lf", &vv, &jj, &ee) != 3) continue;
        const double gns = (((long)lrint(jj)) % 2 != 0) ? 3.0 : 1.0;
        Z = Z + gns * (2.0 * jj + 1.0) * exp(-hc_over_k * ee / T);
        for (int i = 0; i < 2; i++) if ((long)lrint(vv) == vl[i] && (long)lrint(jj) == Jl[i] && El[i] < 0.0) El[i] = ee;
    }
    fclose(f);
    s->h2_nlines = 2;
    for (int i = 0; i < 2; i++) {
        if (El[i] < 0.0) { snprintf(g_err, sizeof g_err, "h2_init: X-level (v=%d, J=%d) not found in %s", vl[i], Jl[i], fn); return -1; }
        s->h2_dnu_Hz[i] = -(dv[i] / speedc) * nu_Lya_Hz;
        s->h2_f_osc[i] = f_osc_const * (2.0 * Ju[i] + 1.0) / (2.0 * Jl[i] + 1.0) * (lam[i] * lam[i]) * A_ul[i];
        const double gns = (Jl[i] % 2 != 0) ? 3.0 : 1.0;
        s->h2_pop[i] = gns * (2.0 * Jl[i] + 1.0) * exp(-hc_over_k * El[i] / T) / Z;
        s->h2_strength[i] = s->f_H2 * s->h2_pop[i] * (s->h2_f_osc[i] / f_osc_Lya);
        s->h2_a_damp[i] = A_tot[i] / (4.0 * PI * s->h2_Dfreq_Hz);
        s->h2_p_scat[i] = s->h2_pure_absorption ? 0.0 : A_ul[i] / A_tot[i];
    }
    s->h2_on = 1;
    return 0;
}
/* h2_kappa_D: h2_mod.f90:58-78 (multiplier of the H I opacity of a cell of Doppler width Dfreq) */
static inline double h2_kappa_D(const orc_sim *s, double xfreq, double Dfreq) {
    double kap = 0.0; if (!s->h2_on) return kap;
    const double ratio = s->h2_hi_width ? 1.0 : Dfreq / s->h2_Dfreq_Hz;
    for (int il = 0; il < s->h2_nlines; il++) { double dx = s->h2_dnu_Hz[il] / Dfreq, x_h2 = (xfreq - dx) * ratio; kap = kap + s->h2_strength[il] * ratio * orc_voigt(x_h2, s->h2_a_damp[il]); }
    return kap;
}
static double rnd(rng_t *g);
/* h2_select_line_D: h2_mod.f90:81-115 (1-based line number) */
static int h2_select_line_D(const orc_sim *s, rng_t *g, double xfreq, double Dfreq) {
    const double ratio = s->h2_hi_width ? 1.0 : Dfreq / s->h2_Dfreq_Hz; double w[4], wtot = 0.0;
    for (int il = 0; il < s->h2_nlines; il++) { double dx = s->h2_dnu_Hz[il] / Dfreq, x_h2 = (xfreq - dx) * ratio; w[il] = s->h2_strength[il] * ratio * orc_voigt(x_h2, s->h2_a_damp[il]); wtot = wtot + w[il]; }
    if (wtot <= 0.0) return 1;
    const double r = rnd(g) * wtot; double acc = 0.0;
    for (int il = 0; il < s->h2_nlines; il++) { acc = acc + w[il]; MRG(r, acc, 67); if (r <= acc) return il + 1; }
    return s->h2_nlines;
}
static int read_input_post(orc_sim *s) {
    if (s->spectral_type == 2) s->comoving_source = 0;          /* setup.f90:115: a continuum is flat in the lab frame */
    /* setup.f90:161-165: a 1-cell axis disables xyz_symmetry */
    if (s->nx == 1 || s->ny == 1 || s->nz == 1) s->xyz_symmetry = 0;
    /* :169-186 observer inside the medium (HEALPix): nostokes, no symmetry, no TAN image */
    if (s->nside > 0 && s->use_clump_medium) { snprintf(g_err, sizeof g_err, "ERROR: nside>0 (HEALPIX inside observer) is not supported with clump mode"); return -1; }
    if (s->nside > 0) { s->use_stokes = 0; s->save_peeloff = 1; s->xyz_symmetry = 0; s->nxim = 0; s->nyim = 0; }
    /* :188-206 peel-off switches */
    if (!(s->save_peeloff_2D || s->save_peeloff_3D)) s->save_peeloff = 0;
    if (s->nxim > 0 && s->nyim > 0) s->save_peeloff = 1;
    if (s->save_peeloff && s->xyz_symmetry) s->save_peeloff = 0;
    if (!s->save_peeloff) { s->save_peeloff_2D = 0; s->save_peeloff_3D = 0; }
    /* :236-238 */
    if (s->cext_dust <= 0.0) s->DGR = 0.0;
    if (s->DGR == 0.0) s->save_Jabs = 0;
    if (s->core_skip_global) s->core_skip = 1;
    if (s->line_type == 8) {                                     /* setup.f90:239-300 */
        const char *why = s->use_clump_medium ? "clump medium" : (s->nside > 0) ? "inside-observer (HEALPix)" : s->use_stokes ? "Stokes polarization" :
            s->fine_structure ? "fine structure" : (s->xyz_symmetry || s->xy_symmetry || s->xy_periodic || s->Omega != 0.0) ? "xyz/xy symmetry, xy_periodic and shear" : NULL;
        if (why) { snprintf(g_err, sizeof g_err, "ly_beta: %s not yet supported", why); return -1; }
        s->core_skip = 0; s->core_skip_global = 0;               /* every skipped core scattering would carry the conversion chance */
        if (s->albedo == 0.3253 && s->hgg == 0.6761 && s->cext_dust == 1.6059e-21) {      /* :282-291; NB these are double-precision literals while the defaults are float32
            literals widened to double (define.f90:454-456): the comparison only holds when the input sets the double values itself */
            s->albedo = 0.2804; s->hgg = 0.6483; s->cext_dust = 2.047e-21; }
        s->R_Ha = (s->cext_dust_Ha <= 0.0 || s->cext_dust <= 0.0) ? 0.0 : s->cext_dust_Ha / s->cext_dust;
    }
    if (strcmp(s->h2_model, "none")) {                            /* setup.f90:302-365 */
        const char *why = (s->line_type != 1) ? "only supported for line_id = 'ly_alpha' (line_type 1)" : strcmp(s->h2_model, "neufeld") ? "Phase 1 supports only h2_model='neufeld'" :
            s->use_clump_medium ? "clump medium not yet supported" : (s->nside > 0) ? "inside-observer (HEALPix) not yet supported" : s->use_stokes ? "Stokes polarization not yet supported" :
            (s->xyz_symmetry || s->xy_symmetry || s->Omega != 0.0) ? "xyz/xy symmetry and shear not yet supported" :
            (s->xy_periodic && (s->nx > 1 || s->ny > 1)) ? "xy_periodic supported only for the nx=ny=1 slab" : !(s->f_H2 > 0.0) ? "par%f_H2 must be > 0 when h2_model /= 'none'" : NULL;
        if (why) { snprintf(g_err, sizeof g_err, "H2: %s", why); return -1; }
        if (h2_init(s)) return -1;
    }
    /* :367-381 Jmu */
    if (s->save_Jmu) { if (s->xyz_symmetry) { s->mu_min = 0.0; s->dmu = 1.0 / s->nmu; } else { s->mu_min = -1.0; s->dmu = 2.0 / s->nmu; } }
    /* :398-428 geometry normalisation */
    if (s->geometry == 0) {
        if (!s->use_amr_grid && !s->xy_periodic) {
            double c[4] = {s->rmax, s->xmax, s->ymax, s->zmax}, r0 = -HUGEST;
            for (int i = 0; i < 4; i++) if (c[i] > 0.0 && c[i] > r0) r0 = c[i];
            if (r0 > 0.0) { s->rmax = s->xmax = s->ymax = s->zmax = r0; }
            if (!s->xy_symmetry) { int n = s->nx; if (s->ny > n) n = s->ny; if (s->nz > n) n = s->nz; s->nx = s->ny = s->nz = n; }
        }
    } else s->rmax = -1.0;
    if (s->source_rmax < 0.0) s->source_rmax = s->rmax;
    /* :438-459 geometry of the J/P profiles: 3 = per cell, 1 = spherical shells, -1 = plane-parallel layers */
    if (s->rmax <= 0.0) s->geometry_JPa = 3;                     /* evaluated on the INPUT rmax, also for the octree (grid_create_amr sets par%rmax later) */
    if (s->geometry_JPa < -1 || s->geometry_JPa > 3) { s->geometry_JPa = 1;
        if (s->xy_periodic) s->geometry_JPa = -1; else if (s->xyz_symmetry) s->geometry_JPa = 1; else if (s->xy_symmetry) s->geometry_JPa = 2; }
    if (s->xy_symmetry && (s->xyz_symmetry || s->xy_periodic)) { snprintf(g_err, sizeof g_err, "xy_symmetry excludes xyz_symmetry and xy_periodic (setup.f90:951-958 takes the first)"); return -1; }
    if ((s->geometry_JPa == 2 || s->geometry_JPa == 0) && (s->calc_J || s->calc_P || s->calc_Pnew)) { snprintf(g_err, sizeof g_err, "geometry_JPa = %d (cylinder) not supported by the oracle", s->geometry_JPa); return -1; }
    /* :473-489 distance unit ('' -> 1) */
    if (s->distance2cm < 0.0) s->distance2cm = 1.0;
    /* :496-500 Mueller table / Stokes without table forces DGR = 0 */
    if (s->scatt_mat_file[0] && s->DGR > 0.0) { if (setup_mueller(s)) return -1; }
    else if (s->use_stokes) s->DGR = 0.0;
    if (s->DGR == 0.0) s->save_Jabs = 0;
    return 0;
}

/* =====================================================================================
 * grid_create, uniform-medium subset: grid_mod_car.f90:73-1238
 * ===================================================================================== */
static void sym_axis(int n, double vmax, double *d, double *vmin, int *i0) {      /* :85-112 */
    if ((n / 2) * 2 == n) { *d = vmax / n; *vmin = 0.0; *i0 = 1; }
    else { *d = vmax / (n - 0.5); *vmin = -(*d) / 2.0; *i0 = 2; }
}
static int grid_create(orc_sim *s) {
    int nx = s->nx, ny = s->ny, nz = s->nz; size_t nc = (size_t)nx * ny * nz;
    if (s->xyz_symmetry) {
        sym_axis(nx, s->xmax, &s->dx, &s->xmin, &s->i0);
        sym_axis(ny, s->ymax, &s->dy, &s->ymin, &s->j0);
        sym_axis(nz, s->zmax, &s->dz, &s->zmin, &s->k0);
    } else if (s->xy_symmetry) {                                                 /* :113-134: mirrors in x and y, the full z range */
        sym_axis(nx, s->xmax, &s->dx, &s->xmin, &s->i0);
        sym_axis(ny, s->ymax, &s->dy, &s->ymin, &s->j0);
        s->dz = 2.0 * s->zmax / nz; s->zmin = -s->zmax; s->k0 = 0;
    } else {                                                                     /* :166-176 */
        s->dx = 2.0 * s->xmax / nx; s->dy = 2.0 * s->ymax / ny; s->dz = 2.0 * s->zmax / nz;
        s->xmin = -s->xmax; s->ymin = -s->ymax; s->zmin = -s->zmax; s->i0 = s->j0 = s->k0 = 0;
    }
    s->xrange = s->xmax - s->xmin; s->yrange = s->ymax - s->ymin; s->zrange = s->zmax - s->zmin;
    s->xface = malloc((nx + 1) * sizeof(double)); s->yface = malloc((ny + 1) * sizeof(double)); s->zface = malloc((nz + 1) * sizeof(double));
    for (int i = 1; i <= nx + 1; i++) s->xface[i-1] = (i - 1) * s->dx + s->xmin;   /* :184-187 */
    for (int i = 1; i <= ny + 1; i++) s->yface[i-1] = (i - 1) * s->dy + s->ymin;
    for (int i = 1; i <= nz + 1; i++) s->zface[i-1] = (i - 1) * s->dz + s->zmin;
    s->vfx = calloc(nc, sizeof(double)); s->vfy = calloc(nc, sizeof(double)); s->vfz = calloc(nc, sizeof(double));
    s->Dfreq = malloc(nc * sizeof(double)); s->voigt_a = malloc(nc * sizeof(double)); s->rhokap = malloc(nc * sizeof(double));
    s->rhokapD = calloc(nc, sizeof(double));
    double *xx = malloc(nx * sizeof(double)), *yy = malloc(ny * sizeof(double)), *zz = malloc(nz * sizeof(double));
    for (int i = 0; i < nx; i++) xx[i] = (s->xface[i] + s->xface[i+1]) / 2.0;      /* :224-232 */
    for (int i = 0; i < ny; i++) yy[i] = (s->yface[i] + s->yface[i+1]) / 2.0;
    for (int i = 0; i < nz; i++) zz[i] = (s->zface[i] + s->zface[i+1]) / 2.0;
    s->Dfreq_ref = vtherm_total(s, s->temperature) / (s->wavelength0 * UM2KM);      /* :238 */
    /* (1) temperature -> Dfreq, voigt_a :273-284 */
    double vth = vtherm_total(s, s->temperature);
    for (size_t c = 0; c < nc; c++) { s->Dfreq[c] = vth / (s->wavelength0 * UM2KM); s->voigt_a[c] = (s->damping / FOURPI) / s->Dfreq[c]; }
    /* (2) density: :352-365 synthetic medium, distance2cm reset to 1 */
    if (!s->use_clump_medium) s->distance2cm = 1.0;              /* grid_mod_car.f90:352-363: the clump population carries the length unit */
    for (size_t c = 0; c < nc; c++) { s->rhokap[c] = 1.0; if (s->DGR > 0.0) s->rhokapD[c] = s->cext_dust * s->DGR; }
    if (s->rmax > 0.0) {                                                            /* :367-395 sphere carved by cell centre */
        for (int k = 1; k <= nz; k++) for (int j = 1; j <= ny; j++) for (int i = 1; i <= nx; i++) {
            double rr = sqrt(xx[i-1] * xx[i-1] + yy[j-1] * yy[j-1] + zz[k-1] * zz[k-1]);
            if (rr < s->rmin || rr > s->rmax) { s->rhokap[IDX(s,i,j,k)] = 0.0; if (s->DGR > 0.0) s->rhokapD[IDX(s,i,j,k)] = 0.0; }
        }
    }
    if (s->cone_opening > 0.0 && s->cone_opening < 90.0) {                          /* :397-412 biconical mask about the z axis */
        const double cos_cone = cos(s->cone_opening * (PI / 180.0));
        for (int k = 1; k <= nz; k++) for (int j = 1; j <= ny; j++) for (int i = 1; i <= nx; i++) {
            double rr = sqrt(xx[i-1] * xx[i-1] + yy[j-1] * yy[j-1] + zz[k-1] * zz[k-1]);
            if (rr > 0.0 && fabs(zz[k-1]) / rr < cos_cone) { s->rhokap[IDX(s,i,j,k)] = 0.0; if (s->DGR > 0.0) s->rhokapD[IDX(s,i,j,k)] = 0.0; }
        }
    }
    for (size_t c = 0; c < nc; c++) s->rhokap[c] = s->rhokap[c] / s->Dfreq[c] * s->cross0;    /* :487-493 */
    double opac_length;                                                             /* :495-503 */
    if (s->rmax > 0.0 && s->rmin > 0.0) opac_length = s->rmax - s->rmin;
    else if (s->rmax > 0.0) opac_length = s->rmax;
    else if (s->zmax == -s->zmin) opac_length = s->zrange / 2.0;
    else opac_length = s->zrange;
    int nxcen, nycen;                                                               /* :505-515 */
    if (s->xyz_symmetry || s->xy_symmetry) { nxcen = 1; nycen = 1; } else { nxcen = (nx + 1) / 2; nycen = (ny + 1) / 2; }
#define NADD(i,j,k,nadd) do { nadd = 1.0; if (s->xyz_symmetry) { \
        if ((i) == 1 && (nx/2)*2 != nx) nadd = nadd / 2.0; if ((j) == 1 && (ny/2)*2 != ny) nadd = nadd / 2.0; \
        if ((k) == 1 && (nz/2)*2 != nz) nadd = nadd / 2.0; } else if (s->xy_symmetry) { \
        if ((i) == 1 && (nx/2)*2 != nx) nadd = nadd / 2.0; if ((j) == 1 && (ny/2)*2 != ny) nadd = nadd / 2.0; } } while (0)
    double opacity_sum, opacity1, opac_norm, nopac, nadd;
    if (s->taumax > 0.0) {                                                          /* :518-538 */
        opacity_sum = 0.0;
        for (int k = 1; k <= nz; k++) opacity_sum = opacity_sum + s->rhokap[IDX(s,nxcen,nycen,k)] * orc_voigt(0.0, s->voigt_a[IDX(s,nxcen,nycen,k)]);
        if (s->xyz_symmetry) {
            if ((nz / 2) * 2 == nz) opac_norm = s->taumax / (opacity_sum * s->dz);
            else { opacity1 = s->rhokap[IDX(s,nxcen,nycen,1)] * orc_voigt(0.0, s->voigt_a[IDX(s,nxcen,nycen,1)]);
                   opac_norm = s->taumax / ((opacity_sum - opacity1 / 2.0) * s->dz); }
        } else if (s->zmax == -s->zmin) opac_norm = (2.0 * s->taumax) / (opacity_sum * s->dz);
        else opac_norm = s->taumax / (opacity_sum * s->dz);
        for (size_t c = 0; c < nc; c++) { s->rhokap[c] *= opac_norm; if (s->DGR > 0.0) s->rhokapD[c] *= opac_norm; }
    } else if (s->tauhomo > 0.0) {                                                  /* :539-566 */
        opacity_sum = 0.0; nopac = 0.0;
        for (int k = 1; k <= nz; k++) for (int j = 1; j <= ny; j++) for (int i = 1; i <= nx; i++) {
            NADD(i, j, k, nadd);
            if (s->rhokap[IDX(s,i,j,k)] > 0.0) { opacity_sum += s->rhokap[IDX(s,i,j,k)] * orc_voigt(0.0, s->voigt_a[IDX(s,i,j,k)]) * nadd; nopac += nadd; } }
        opac_norm = s->tauhomo / (opacity_sum / nopac * opac_length);
        for (size_t c = 0; c < nc; c++) { s->rhokap[c] *= opac_norm; if (s->DGR > 0.0) s->rhokapD[c] *= opac_norm; }
    } else if (s->N_gasmax > 0.0) {                                                 /* :567-586 */
        opacity_sum = 0.0;
        for (int k = 1; k <= nz; k++) opacity_sum += s->rhokap[IDX(s,nxcen,nycen,k)] * s->Dfreq[IDX(s,nxcen,nycen,k)];
        if (s->xyz_symmetry) {
            if ((nz / 2) * 2 == nz) opac_norm = s->N_gasmax / (opacity_sum * s->dz / s->cross0);
            else { opacity1 = s->rhokap[IDX(s,nxcen,nycen,1)] * s->Dfreq[IDX(s,nxcen,nycen,1)];
                   opac_norm = s->N_gasmax / ((opacity_sum - opacity1 / 2.0) * s->dz / s->cross0); }
        } else if (s->zmax == -s->zmin) opac_norm = (2.0 * s->N_gasmax) / (opacity_sum * s->dz / s->cross0);
        else opac_norm = s->N_gasmax / (opacity_sum * s->dz / s->cross0);
        for (size_t c = 0; c < nc; c++) { s->rhokap[c] *= opac_norm; if (s->DGR > 0.0) s->rhokapD[c] *= opac_norm; }
    } else if (s->N_gashomo > 0.0) {                                                /* :587-615 */
        opacity_sum = 0.0; nopac = 0.0;
        for (int k = 1; k <= nz; k++) for (int j = 1; j <= ny; j++) for (int i = 1; i <= nx; i++) {
            NADD(i, j, k, nadd);
            if (s->rhokap[IDX(s,i,j,k)] > 0.0) { opacity_sum += s->rhokap[IDX(s,i,j,k)] * s->Dfreq[IDX(s,i,j,k)] * nadd; nopac += nadd; } }
        double dens = opacity_sum / nopac / s->cross0;
        opac_norm = s->N_gashomo / (dens * opac_length);
        for (size_t c = 0; c < nc; c++) { s->rhokap[c] *= opac_norm; if (s->DGR > 0.0) s->rhokapD[c] *= opac_norm; }
    }
    /* homogeneous / pole optical depths and columns :617-737 */
    double tauhomo, taupole, N_gashomo, N_gaspole, os2 = 0.0;
    opacity_sum = 0.0; nopac = 0.0;
    for (int k = 1; k <= nz; k++) for (int j = 1; j <= ny; j++) for (int i = 1; i <= nx; i++) {
        NADD(i, j, k, nadd);
        if (s->rhokap[IDX(s,i,j,k)] > 0.0) { opacity_sum += s->rhokap[IDX(s,i,j,k)] * orc_voigt(0.0, s->voigt_a[IDX(s,i,j,k)]) * nadd;
                                              os2 += s->rhokap[IDX(s,i,j,k)] * s->Dfreq[IDX(s,i,j,k)] * nadd; nopac += nadd; } }
    tauhomo = opacity_sum / nopac * opac_length;
    N_gashomo = os2 / nopac / s->cross0 * opac_length;
    opacity_sum = 0.0; os2 = 0.0;
    for (int k = 1; k <= nz; k++) { opacity_sum += s->rhokap[IDX(s,nxcen,nycen,k)] * orc_voigt(0.0, s->voigt_a[IDX(s,nxcen,nycen,k)]);
                                    os2 += s->rhokap[IDX(s,nxcen,nycen,k)] * s->Dfreq[IDX(s,nxcen,nycen,k)]; }
    if (s->xyz_symmetry) {
        taupole = opacity_sum * s->dz; N_gaspole = os2 * s->dz / s->cross0;
        if ((nz / 2) * 2 != nz) {
            taupole -= s->rhokap[IDX(s,nxcen,nycen,1)] * orc_voigt(0.0, s->voigt_a[IDX(s,nxcen,nycen,1)]) * s->dz / 2.0;
            N_gaspole -= s->rhokap[IDX(s,nxcen,nycen,1)] * s->Dfreq[IDX(s,nxcen,nycen,1)] * s->dz / 2.0 / s->cross0; }
    } else if (s->zmax == -s->zmin) { taupole = opacity_sum * s->dz / 2.0; N_gaspole = os2 * s->dz / 2.0 / s->cross0; }
    else { taupole = opacity_sum * s->dz; N_gaspole = os2 * s->dz / s->cross0; }
    if (s->taumax <= 0.0) s->taumax = taupole;
    if (s->tauhomo <= 0.0) s->tauhomo = tauhomo;
    if (s->N_gasmax <= 0.0) s->N_gasmax = N_gaspole;
    if (s->N_gashomo <= 0.0) s->N_gashomo = N_gashomo;
    /* (3) velocity field :786-946 (velocities in units of the local thermal speed) */
    if (s->velocity_type) {
        double rpeak = (s->rmax <= 0.0) ? fmax(s->xmax, fmax(s->ymax, s->zmax)) : s->rmax;
        if (s->velocity_type == 8) rpeak = s->rpeak;                                /* 'ssh': par%rpeak is an input (Song, Seon & Hwang 2020) */
        for (int k = 1; k <= nz; k++) for (int j = 1; j <= ny; j++) for (int i = 1; i <= nx; i++) {
            size_t c = IDX(s,i,j,k); if (!(s->rhokap[c] > 0.0)) continue;
            double vt = vtherm_total(s, s->temperature), x = xx[i-1], y = yy[j-1], z = zz[k-1], rr = sqrt(x * x + y * y + z * z), Vs;
            switch (s->velocity_type) {
            case 1:                                                                 /* hubble :786-803 */
                s->vfx[c] = (s->Vexp / vt) * x / rpeak; s->vfy[c] = (s->Vexp / vt) * y / rpeak; s->vfz[c] = (s->Vexp / vt) * z / rpeak; break;
            case 2:                                                                 /* constant_radial :839-852 */
                if (rr > s->dz / 10.0) { s->vfx[c] = s->Vexp / vt * x / rr; s->vfy[c] = s->Vexp / vt * y / rr; s->vfz[c] = s->Vexp / vt * z / rr; } break;
            case 3: s->vfx[c] = s->Vx / vt; s->vfy[c] = s->Vy / vt; s->vfz[c] = s->Vz / vt; break;   /* parallel_velocity :804-816 */
            case 4:                                                                 /* rotating_solid_body :905-916 */
                s->vfx[c] = -s->Vrot / vt * y / s->rmax; s->vfy[c] = s->Vrot / vt * x / s->rmax; break;
            case 5: {                                                               /* rotating_galaxy_halo :917-934 */
                double rc = sqrt(x * x + y * y);
                if (rc < s->rinner) { s->vfx[c] = -s->Vrot / vt * y / s->rinner; s->vfy[c] = s->Vrot / vt * x / s->rinner; }
                else { s->vfx[c] = -s->Vrot / vt * y / rc; s->vfy[c] = s->Vrot / vt * x / rc; } } break;
            case 6:                                                                 /* power_law :853-874 */
                if (rr > s->dz / 10.0) { Vs = s->Vexp * pow(rr / rpeak, s->velocity_alpha);
                    s->vfx[c] = Vs / vt * x / rr; s->vfy[c] = Vs / vt * y / rr; s->vfz[c] = Vs / vt * z / rr; } break;
            case 7:                                                                 /* linear_decelerate :875-904 */
                if (rr > s->dz / 10.0) { Vs = s->Vexp * fmax(0.0, (rpeak - rr) / (rpeak - fmax(s->rmin, 0.0)));
                    s->vfx[c] = Vs / vt * x / rr; s->vfy[c] = Vs / vt * y / rr; s->vfz[c] = Vs / vt * z / rr; } break;
            case 8:                                                                 /* ssh :817-838 */
                if (rr < rpeak) { Vs = s->Vpeak / rpeak; s->vfx[c] = Vs / vt * x; s->vfy[c] = Vs / vt * y; s->vfz[c] = Vs / vt * z; }
                else { Vs = s->Vpeak + s->DeltaV * (rr - rpeak) / (s->rmax - rpeak);
                       s->vfx[c] = Vs / vt * x / rr; s->vfy[c] = Vs / vt * y / rr; s->vfz[c] = Vs / vt * z / rr; } break;
            }
        }
    }
    free(xx); free(yy); free(zz);
    /* :1150-1154 */
    s->voigt_amean = (s->damping / FOURPI) / s->Dfreq_ref;
    double atau0 = s->voigt_amean * s->tauhomo;
    /* car_setup_freq_grid :1442-1548 */
    s->atau3 = pow(s->voigt_amean * s->tauhomo, 1.0 / 3.0);
    if (isfinite(s->wavelength_min) && isfinite(s->wavelength_max)) {               /* :1457-1461 */
        double vt0 = vtherm_total(s, s->temperature), speedc = 2.99792458e5;
        s->par_xfreq_min = -(s->wavelength_max - s->wavelength0 * 1e4) / (s->wavelength0 * 1e4) * (speedc / vt0);
        s->par_xfreq_max = -(s->wavelength_min - s->wavelength0 * 1e4) / (s->wavelength0 * 1e4) * (speedc / vt0);
    } else
    if (isfinite(s->velocity_min) && isfinite(s->velocity_max)) {                   /* :1462-1467 */
        double vt0 = vtherm_total(s, s->temperature);
        s->par_xfreq_min = -s->velocity_max / vt0; s->par_xfreq_max = -s->velocity_min / vt0;
    }
    if (!(isfinite(s->par_xfreq_max) && isfinite(s->par_xfreq_min))) {
        double xscale = (s->taumax <= 5e1) ? 25.0 : (s->taumax <= 5e2) ? 14.0 : (s->taumax <= 5e3) ? 10.0 : 5.0;
        double vt = vtherm_total(s, s->temperature);
        if (s->Vexp == 0.0) { s->par_xfreq_max = floor(xscale * s->atau3) + 1; s->par_xfreq_min = -(floor(xscale * s->atau3 + s->DnuHK_Hz / s->Dfreq_ref) + 1); }
        else if (s->Vexp > 0.0) { s->par_xfreq_max = floor(xscale * s->atau3) + 1; s->par_xfreq_min = -(floor(xscale * s->atau3 + fabs(s->Vexp) / vt + s->DnuHK_Hz / s->Dfreq_ref) + 1); }
        else { s->par_xfreq_max = floor(xscale * s->atau3 + fabs(s->Vexp) / vt) + 1; s->par_xfreq_min = -(floor(xscale * s->atau3 + s->DnuHK_Hz / s->Dfreq_ref) + 1); }
        if (s->spectral_type == 2) { xscale = 4.0 * xscale;                         /* :1490-1494 */
            s->par_xfreq_max = floor(xscale * s->atau3 + fabs(s->Vexp) / vt) + 1; s->par_xfreq_min = -(floor(xscale * s->atau3 + fabs(s->Vexp) / vt + s->DnuHK_Hz / s->Dfreq_ref) + 1); }
    }
    s->dxfreq = (s->par_xfreq_max - s->par_xfreq_min) / s->nxfreq;
    s->xfreq_max = s->par_xfreq_max; s->xfreq_min = s->par_xfreq_min;
    if (s->line_type == 8) {                                    /* car_setup_freq_grid :1518-1529 / amr_setup_freq_grid: the band-2 (H-alpha) axis */
        s->nxfreq_Ha = s->par_nxfreq_Ha > 0 ? s->par_nxfreq_Ha : s->nxfreq;
        if (s->par_xfreq_max_Ha > 0.0) { s->xfreq_max_Ha = s->par_xfreq_max_Ha; s->xfreq_min_Ha = -s->par_xfreq_max_Ha; } else { s->xfreq_min_Ha = s->xfreq_min; s->xfreq_max_Ha = s->xfreq_max; }
        s->dxfreq_Ha = (s->xfreq_max_Ha - s->xfreq_min_Ha) / s->nxfreq_Ha;
        s->Jout_Ha = calloc(s->nxfreq_Ha, 8); s->Jabs_Ha = calloc(s->nxfreq_Ha, 8);
    }
    /* tallies :1161-1183 */
    s->Jout = calloc(s->nxfreq, sizeof(double)); s->Jin = calloc(s->nxfreq, sizeof(double)); s->Jabs = calloc(s->nxfreq, sizeof(double));
    s->Jmu = calloc((size_t)s->nxfreq * (s->nmu > 0 ? s->nmu : 1), sizeof(double));
    /* global core-skip parameters :1186-1219 */
    double at = s->core_skip_global ? atau0 : atau0 / (s->xmax / s->dx);
    if (at <= 1.0) { s->xcrit = 0.0; s->xcrit2 = 0.0; }
    else { double xi = (at <= 60.0) ? 0.6 : 1.4, chi = (at <= 60.0) ? 1.2 : 0.6;
           s->xcrit = 0.02 * exp(xi * pow(log(at), chi)); s->xcrit2 = s->xcrit * s->xcrit; }
    /* dispatch as setup_procedure, setup.f90:967-990 */
    if (s->xyz_symmetry) s->raytrace_variant = 1;
    else if (s->xy_symmetry) s->raytrace_variant = 5;                               /* raytrace_to_{tau,edge}_car_xysym */
    else if (s->xy_periodic && s->nx == 1 && s->ny == 1) s->raytrace_variant = 2;
    else if (s->xy_periodic) s->raytrace_variant = (s->Omega != 0.0) ? 8 : 6;       /* raytrace_to_{tau,edge}_car_xyper; 8 = raytrace_to_tau_car_xyper_shear (setup.f90:959-973) */
    else s->raytrace_variant = 0;
    return 0;
}

/* create_JPa_mem: grid_mod_car.f90:1242-1434 (geometry_JPa = 1 spherical shells, -1 layers, 3 per cell).
 * ind_bin(cell) = 1-based profile bin of a cell, 0 = none; ncount(bin) = cells per bin (x8/4/2 in the folded octant). */
static int create_JPa_mem(orc_sim *s) {
    if (!(s->calc_J || s->calc_P || s->calc_Pnew)) return 0;
    int nx = s->nx, ny = s->ny, nz = s->nz; size_t nc = (size_t)nx * ny * nz;
    s->ind_bin = calloc(nc, sizeof(int));
    if (s->geometry_JPa == 1) {
        int nr = nx > ny ? nx : ny; if (nz > nr) nr = nz;
        if (!s->xyz_symmetry) { if ((nr / 2) * 2 == nr) nr = nr / 2; else nr = (nr - 1) / 2 + 1; }
        double rmax = fmin(s->xmax, fmin(s->ymax, s->zmax)), dr, roff;
        if ((nr / 2) * 2 == nr) { dr = rmax / nr; roff = 0.0; } else { dr = rmax / (nr - 0.5); roff = -dr / 2.0; }
        s->nbins = nr; s->ncount = calloc(nr, sizeof(double));
        for (int k = 1; k <= nz; k++) for (int j = 1; j <= ny; j++) for (int i = 1; i <= nx; i++) {
            double xx = (s->xface[i-1] + s->xface[i]) / 2.0, yy = (s->yface[j-1] + s->yface[j]) / 2.0, zz = (s->zface[k-1] + s->zface[k]) / 2.0;
            double rr = sqrt(xx * xx + yy * yy + zz * zz);
            int ib = (int)floor((rr - roff) / dr) + 1;
            s->ind_bin[IDX(s,i,j,k)] = (ib >= 1 && ib <= nr) ? ib : 0;        /* the reference indexes past nr here; guard */
            if (s->rhokap[IDX(s,i,j,k)] > 0.0 && ib >= 1 && ib <= nr) {
                int nadd = 1;
                if (s->xyz_symmetry) { nadd = 8; if (i == 1 && (nx/2)*2 != nx) nadd /= 2; if (j == 1 && (ny/2)*2 != ny) nadd /= 2; if (k == 1 && (nz/2)*2 != nz) nadd /= 2; }
                s->ncount[ib-1] += nadd; }
        }
    } else if (s->geometry_JPa == -1) {
        s->nbins = nz; s->ncount = calloc(nz, sizeof(double));
        for (int k = 1; k <= nz; k++) for (int j = 1; j <= ny; j++) for (int i = 1; i <= nx; i++) {
            s->ind_bin[IDX(s,i,j,k)] = k; if (s->rhokap[IDX(s,i,j,k)] > 0.0) s->ncount[k-1] += 1.0; }
    } else {
        s->nbins = (int)nc; s->ncount = calloc(nc, sizeof(double));
        for (int k = 1; k <= nz; k++) for (int j = 1; j <= ny; j++) for (int i = 1; i <= nx; i++) {
            size_t c = IDX(s,i,j,k); s->ind_bin[c] = (int)c + 1; int nadd = 1;
            if (s->xyz_symmetry) { nadd = 8; if (i == 1 && (nx/2)*2 != nx) nadd /= 2; if (j == 1 && (ny/2)*2 != ny) nadd /= 2; if (k == 1 && (nz/2)*2 != nz) nadd /= 2; }
            s->ncount[c] = nadd; }
    }
    if (s->calc_J) s->J1 = calloc((size_t)s->nxfreq * s->nbins, sizeof(double));
    if (s->calc_P) s->P1 = calloc(s->nbins, sizeof(double));
    if (s->calc_P && s->line_type == 8) s->Pconv = calloc(s->nbins, sizeof(double));
    if (s->calc_Pnew) s->P1_new = calloc(s->nbins, sizeof(double));
    return 0;
}
/* add_to_J (raytrace_car.f90:3979-4011) and add_to_Pnew (:4015-4045) for one path segment in cell c */
static inline void add_to_JPnew(const orc_sim *s, acc_t *A, const photon_t *p, size_t c, double del, double dtauH) {
    if (!(s->rhokap[c] > 0.0)) return;
    int ib = s->ind_bin[c]; if (ib < 1) return;
    if (s->calc_J) {
        double xr = p->xfreq * (s->Dfreq[c] / s->Dfreq_ref);
        int ix = (int)floor((xr - s->xfreq_min) / s->dxfreq) + 1;
        if (ix > 0 && ix <= s->nxfreq) A->J1[(size_t)(ix - 1) + (size_t)s->nxfreq * (ib - 1)] += del * p->wgt;
    }
    if (s->calc_Pnew) { double n = s->rhokap[c] * s->Dfreq[c] / s->cross0; A->P1_new[ib-1] += dtauH * p->wgt / n; }
}
/* add_to_Pa: scattering_car.f90:830-859 */
static inline void add_to_Pa(const orc_sim *s, acc_t *A, const photon_t *p) {
    if (!s->calc_P) return;
    size_t c = IDX(s,p->icell,p->jcell,p->kcell);
    if (!(s->rhokap[c] > 0.0)) return;
    int ib = s->use_amr_grid ? amr_JPa_bin(s, p->icell, p->x, p->y, p->z) : s->ind_bin[c];   /* add_to_Pa_amr: scattering_amr.f90:917-942 */
    if (ib < 1) return;
    A->P1[ib-1] += p->wgt / (s->rhokap[c] * s->Dfreq[c] / s->cross0);
}

/* add_to_Pconv (scattering_car.f90:861-894) / add_to_Pconv_amr: the conversion-rate profile of line type 8, binned and normalised like add_to_Pa */
static inline void add_to_Pconv(const orc_sim *s, acc_t *A, const photon_t *p) {
    if (!s->calc_P || !A->Pconv) return;
    size_t c = IDX(s,p->icell,p->jcell,p->kcell);
    if (!(s->rhokap[c] > 0.0)) return;
    int ib = s->use_amr_grid ? amr_JPa_bin(s, p->icell, p->x, p->y, p->z) : s->ind_bin[c];
    if (ib < 1) return;
    A->Pconv[ib-1] += p->wgt / (s->rhokap[c] * s->Dfreq[c] / s->cross0);
}

/* observer_create_outside: observer_rect.f90:24-326 */
static int observer_create(orc_sim *s) {
    if (!s->save_peeloff) { s->nobs = 0; return 0; }
    if (s->nside > 0) {                                                             /* observer_create_inside: observer_heal.f90:24-80 */
        if (s->raytrace_variant != 0 && s->raytrace_variant != 3) { snprintf(g_err, sizeof g_err, "HEALPix inside observers need the plain Cartesian grid or the full octree"); return -1; }
        if (!(isfinite(s->obsx[0]) && isfinite(s->obsy[0]) && isfinite(s->obsz[0]))) { s->obsx[0] = 0.0; s->obsy[0] = 0.0; s->obsz[0] = 0.0; }
        int n = 0; for (int i = 0; i < MAX_OBS; i++) if (isfinite(s->obsx[i]) && isfinite(s->obsy[i]) && isfinite(s->obsz[i])) n++;
        s->nobs = n; s->npix = 12 * s->nside * s->nside; s->nxim = s->npix; s->nyim = 1;
        size_t n2 = s->npix, n3 = n2 * s->nxfreq;
        for (int i = 0; i < n; i++) { obs_t *o = &s->obs[i]; o->x = s->obsx[i]; o->y = s->obsy[i]; o->z = s->obsz[i];
            if (s->save_peeloff_2D) { o->scatt_2D = calloc(n2, 8); o->direc_2D = calloc(n2, 8); if (s->save_direc0) o->direc0_2D = calloc(n2, 8); }
            if (s->save_peeloff_3D) { o->scatt = calloc(n3, 8); o->direc = calloc(n3, 8); if (s->save_direc0) o->direc0 = calloc(n3, 8); if (s->save_dust_scattered) o->scatt_dust = calloc(n3, 8); } }
        return 0;
    }
    double cosa[MAX_OBS], sina[MAX_OBS], cosb[MAX_OBS], sinb[MAX_OBS], cosg[MAX_OBS], sing[MAX_OBS];
    double rcx = 0.0, rcy = 0.0, rcz = 0.0;
    double maxsz = fmax(s->xmax, fmax(s->ymax, s->zmax));
    for (int i = 0; i < MAX_OBS; i++) {                                            /* :54-55 */
        if (isfinite(s->beta[i]) && !isfinite(s->alpha[i])) s->alpha[i] = 0.0;
        if (isfinite(s->alpha[i]) && !isfinite(s->beta[i])) s->beta[i] = 0.0; }
    if (!(isfinite(s->alpha[0]) && isfinite(s->beta[0])) && !(isfinite(s->obsx[0]) && isfinite(s->obsy[0]) && isfinite(s->obsz[0]))) {
        if (!isfinite(s->distance)) s->distance = maxsz * 100.0;                   /* :58-71 default observer */
        s->obsx[0] = 0.0; s->obsy[0] = 0.0; s->obsz[0] = 1.0; s->alpha[0] = 0.0; s->beta[0] = 0.0; }
    if (isfinite(s->alpha[0]) && isfinite(s->beta[0])) {                           /* :73-122 */
        int n = 0; for (int i = 0; i < MAX_OBS; i++) if (isfinite(s->alpha[i]) && isfinite(s->beta[i])) n++;
        s->nobs = n;
        if (!isfinite(s->distance)) s->distance = maxsz * 100.0;
        for (int i = 0; i < n; i++) {
            if (!isfinite(s->gamma[i])) { if (s->beta[i] > 0.0 && s->beta[i] <= 90.0) s->gamma[i] = 90.0; else if (s->beta[i] > 90.0) s->gamma[i] = -90.0; else s->gamma[i] = 0.0; }
            cosa[i] = cos(s->alpha[i] * DEG2RAD); sina[i] = sin(s->alpha[i] * DEG2RAD);
            cosb[i] = cos(s->beta[i] * DEG2RAD);  sinb[i] = sin(s->beta[i] * DEG2RAD);
            cosg[i] = cos(s->gamma[i] * DEG2RAD); sing[i] = sin(s->gamma[i] * DEG2RAD);
            s->obs[i].x = s->distance * cosa[i] * sinb[i] + rcx; s->obs[i].y = s->distance * sina[i] * sinb[i] + rcy; s->obs[i].z = s->distance * cosb[i] + rcz;
        }
    } else {                                                                       /* :123-199 */
        int n = 0; for (int i = 0; i < MAX_OBS; i++) if (isfinite(s->obsx[i]) && isfinite(s->obsy[i]) && isfinite(s->obsz[i])) n++;
        s->nobs = n;
        if (!isfinite(s->distance)) { s->distance = sqrt(s->obsx[0] * s->obsx[0] + s->obsy[0] * s->obsy[0] + s->obsz[0] * s->obsz[0]);
            if (s->distance < 10.0 * maxsz) s->distance = maxsz * 100.0; }
        for (int i = 0; i < n; i++) {
            if (!isfinite(s->gamma[i])) { if (s->beta[i] > 0.0 && s->beta[i] <= 90.0) s->gamma[i] = 90.0; else if (s->beta[i] > 90.0) s->gamma[i] = -90.0; else s->gamma[i] = 0.0; }
            double ds = s->distance / sqrt(s->obsx[i] * s->obsx[i] + s->obsy[i] * s->obsy[i] + s->obsz[i] * s->obsz[i]);
            if (ds > 1.001) { s->obs[i].x = s->obsx[i] * ds + rcx; s->obs[i].y = s->obsy[i] * ds + rcy; s->obs[i].z = s->obsz[i] * ds + rcz; }
            else { s->obs[i].x = s->obsx[i]; s->obs[i].y = s->obsy[i]; s->obs[i].z = s->obsz[i]; }
            cosb[i] = (s->obs[i].z - rcz) / s->distance;
            if (fabs(cosb[i] - 1.0) < DBL_EPSILON) cosb[i] = 1.0;
            if (fabs(cosb[i] + 1.0) < DBL_EPSILON) cosb[i] = -1.0;
            sinb[i] = sqrt(1.0 - cosb[i] * cosb[i]);
            s->beta[i] = atan2(sinb[i], cosb[i]) * RAD2DEG;
            cosg[i] = cos(s->gamma[i] * DEG2RAD); sing[i] = sin(s->gamma[i] * DEG2RAD);
            if (sinb[i] == 0.0) { cosa[i] = 1.0; sina[i] = 0.0; s->alpha[i] = 0.0; }
            else { double al = atan2(s->obs[i].y - rcy, s->obs[i].x - rcx); cosa[i] = cos(al); sina[i] = sin(al); s->alpha[i] = al * RAD2DEG; }
        }
    }
    for (int i = 0; i < s->nobs; i++) {                                            /* :207-217 */
        double *m = s->obs[i].rm;
        m[0] =  cosa[i] * cosb[i] * cosg[i] - sina[i] * sing[i]; m[1] =  sina[i] * cosb[i] * cosg[i] + cosa[i] * sing[i]; m[2] = -sinb[i] * cosg[i];
        m[3] = -cosa[i] * cosb[i] * sing[i] - sina[i] * cosg[i]; m[4] = -sina[i] * cosb[i] * sing[i] + cosa[i] * cosg[i]; m[5] =  sinb[i] * sing[i];
        m[6] =  cosa[i] * sinb[i]; m[7] = sina[i] * sinb[i]; m[8] = cosb[i];
    }
    if (!(isfinite(s->dxim) && isfinite(s->dyim))) {                               /* :244-276 */
        if (s->geometry == 0) { s->dxim = asin(s->rmax / s->distance) / (s->nxim / 2.0) * RAD2DEG; s->dyim = asin(s->rmax / s->distance) / (s->nyim / 2.0) * RAD2DEG; }
        else {
            static const double vx[8] = {1, 1, 1, -1, -1, -1, 1, -1}, vy[8] = {1, 1, -1, 1, -1, 1, -1, -1}, vz[8] = {1, -1, 1, 1, 1, -1, -1, -1};
            double max_x = -999.0, max_y = -999.0;
            for (int i = 0; i < s->nobs; i++) for (int iv = 0; iv < 8; iv++) {
                double px = s->obs[i].x - vx[iv] * s->xmax, py = s->obs[i].y - vy[iv] * s->ymax, pz = s->obs[i].z - vz[iv] * s->zmax, *m = s->obs[i].rm;
                double kx = m[0] * px + m[1] * py + m[2] * pz, ky = m[3] * px + m[4] * py + m[5] * pz, kz = m[6] * px + m[7] * py + m[8] * pz;
                double ax = fabs(atan2(-kx, kz)), ay = fabs(atan2(-ky, kz));
                if (max_x < ax) max_x = ax; if (max_y < ay) max_y = ay; }
            if (s->nxim == s->nyim) { double m = fmax(max_x, max_y); s->dxim = m / (s->nxim / 2.0) * RAD2DEG; s->dyim = m / (s->nyim / 2.0) * RAD2DEG; }
            else { if (!isfinite(s->dxim)) s->dxim = max_x / (s->nxim / 2.0) * RAD2DEG; if (!isfinite(s->dyim)) s->dyim = max_y / (s->nyim / 2.0) * RAD2DEG; }
        }
    }
    size_t n2 = (size_t)s->nxim * s->nyim, n3 = n2 * s->nxfreq;                    /* :287-324 */
    for (int i = 0; i < s->nobs; i++) { obs_t *o = &s->obs[i];
        if (s->save_peeloff_2D) { o->scatt_2D = calloc(n2, 8); o->direc_2D = calloc(n2, 8); if (s->save_direc0) o->direc0_2D = calloc(n2, 8);
            if (s->use_stokes) { o->I_2D = calloc(n2, 8); o->Q_2D = calloc(n2, 8); o->U_2D = calloc(n2, 8); o->V_2D = calloc(n2, 8); } }
        if (s->save_peeloff_3D && s->line_type == 8) o->peel_Ha = calloc(n2 * s->nxfreq_Ha, 8);      /* observer_rect.f90:306-311 */
        if (s->save_peeloff_3D) { o->scatt = calloc(n3, 8); o->direc = calloc(n3, 8); if (s->save_direc0) o->direc0 = calloc(n3, 8); if (s->save_dust_scattered) o->scatt_dust = calloc(n3, 8);
            if (s->use_stokes) { o->I = calloc(n3, 8); o->Q = calloc(n3, 8); o->U = calloc(n3, 8); o->V = calloc(n3, 8); } } }
    return 0;
}

orc_sim *orc_create(const char *params) {
    orc_sim *s = malloc(sizeof *s); set_defaults(s); setup_line(s);
    char *buf = strdup(params), *save = NULL;
    for (char *tok = strtok_r(buf, " \t\n,", &save); tok; tok = strtok_r(NULL, " \t\n,", &save)) {
        char *eq = strchr(tok, '='); if (!eq) { snprintf(g_err, sizeof g_err, "bad token '%s'", tok); free(buf); free(s); return NULL; }
        *eq = 0; const char *k = tok; if (!strncmp(k, "par%", 4)) k += 4;
        if (set_param(s, k, eq + 1)) { free(buf); free(s); return NULL; }
    }
    free(buf);
    setup_line(s);                                          /* again: par%fine_structure selects the line type */
    if (read_input_post(s) || (s->use_clump_medium ? grid_create_clump(s) : s->use_amr_grid ? (grid_create_amr(s) || create_JPa_amr_mem(s)) : (grid_create(s) || create_JPa_mem(s))) || observer_create(s)) { free(s); return NULL; }
    return s;
}
static void free_obs(obs_t *o) { free(o->peel_Ha); o->peel_Ha = NULL; free(o->scatt); free(o->direc); free(o->I); free(o->Q); free(o->U); free(o->V);
    free(o->scatt_2D); free(o->direc_2D); free(o->I_2D); free(o->Q_2D); free(o->U_2D); free(o->V_2D);
    free(o->direc0); free(o->direc0_2D); free(o->scatt_dust); }
void orc_destroy(orc_sim *s) {
    if (!s) return;
    free(s->xface); free(s->yface); free(s->zface); free(s->rhokap); free(s->voigt_a); free(s->Dfreq); free(s->vfx); free(s->vfy); free(s->vfz); free(s->rhokapD);
    free(s->Jout); free(s->Jin); free(s->Jabs); free(s->Jmu); free(s->coss); free(s->S11); free(s->S12); free(s->S33); free(s->S34); free(s->phase_PDF); free(s->alias);
    for (int i = 0; i < MAX_OBS; i++) free_obs(&s->obs[i]);
    free(s->ind_bin); free(s->ncount); free(s->J1); free(s->P1); free(s->P1_new);
    free(s->log_nscatt); free(s->log_xfreq); free(s);
}


/* =====================================================================================
 * Octree AMR medium: octree_mod.f90, grid_mod_amr.f90, read_generic_amr.f90, raytrace_amr.f90
 * ===================================================================================== */
#define CH(a,io,ic) ((a)->children[(size_t)((io)-1) + 8 * (size_t)((ic)-1)])
#define NB(a,f,ic)  ((a)->neighbor[(size_t)((f)-1) + 6 * (size_t)((ic)-1)])

/* amr_find_leaf: octree_mod.f90:149-171 */
static int amr_find_leaf(const amr_t *a, double x, double y, double z) {
    if (x < a->xmin || x > a->xmax || y < a->ymin || y > a->ymax || z < a->zmin || z > a->zmax) return 0;
    int ic = 1;
    for (;;) {
        if (a->ileaf[ic-1] > 0) return a->ileaf[ic-1];
        int io = 1; if (x >= a->cx[ic-1]) io += 1; if (y >= a->cy[ic-1]) io += 2; if (z >= a->cz[ic-1]) io += 4;
        ic = CH(a, io, ic);
        if (ic == 0) return 0;
    }
}
/* amr_find_cell_at_level: :827-849 */
static int amr_find_cell_at_level(const amr_t *a, double x, double y, double z, int target) {
    if (x < a->xmin || x > a->xmax || y < a->ymin || y > a->ymax || z < a->zmin || z > a->zmax) return 0;
    int ic = 1;
    for (;;) {
        if (a->level[ic-1] >= target) return ic;
        if (a->ileaf[ic-1] > 0) return ic;
        int io = 1; if (x >= a->cx[ic-1]) io += 1; if (y >= a->cy[ic-1]) io += 2; if (z >= a->cz[ic-1]) io += 4;
        int ch = CH(a, io, ic);
        if (ch == 0) return ic;
        ic = ch;
    }
}
/* amr_cell_exit: :412-451 (iface 1=+x 2=-x 3=+y 4=-y 5=+z 6=-z; first minimum) */
static void amr_cell_exit(const amr_t *a, int ic, double x, double y, double z, double kx, double ky, double kz, double *t_exit, int *iface) {
    double cx = a->cx[ic-1], cy = a->cy[ic-1], cz = a->cz[ic-1], h = a->ch[ic-1], t[6];
    if (kx > 0.0) { t[0] = (cx + h - x) / kx; t[1] = HUGEST; } else if (kx < 0.0) { t[0] = HUGEST; t[1] = (cx - h - x) / kx; } else { t[0] = HUGEST; t[1] = HUGEST; }
    if (ky > 0.0) { t[2] = (cy + h - y) / ky; t[3] = HUGEST; } else if (ky < 0.0) { t[2] = HUGEST; t[3] = (cy - h - y) / ky; } else { t[2] = HUGEST; t[3] = HUGEST; }
    if (kz > 0.0) { t[4] = (cz + h - z) / kz; t[5] = HUGEST; } else if (kz < 0.0) { t[4] = HUGEST; t[5] = (cz - h - z) / kz; } else { t[4] = HUGEST; t[5] = HUGEST; }
    int m = 0; for (int i = 1; i < 6; i++) if (t[i] < t[m]) m = i;
    *iface = m + 1; *t_exit = t[m];
}
/* amr_next_leaf: :718-758 (normal sub-octant bit from iface, transverse bits from the position) */
static int amr_next_leaf(const amr_t *a, int ic, int iface, double x, double y, double z) {
    int ng = NB(a, iface, ic);
    if (ng == 0) return 0;
    while (a->ileaf[ng-1] == 0) {
        int io = 1;
        switch (iface) {
        case 1: if (y >= a->cy[ng-1]) io += 2; if (z >= a->cz[ng-1]) io += 4; break;
        case 2: io += 1; if (y >= a->cy[ng-1]) io += 2; if (z >= a->cz[ng-1]) io += 4; break;
        case 3: if (x >= a->cx[ng-1]) io += 1; if (z >= a->cz[ng-1]) io += 4; break;
        case 4: if (x >= a->cx[ng-1]) io += 1; io += 2; if (z >= a->cz[ng-1]) io += 4; break;
        case 5: if (x >= a->cx[ng-1]) io += 1; if (y >= a->cy[ng-1]) io += 2; break;
        default: if (x >= a->cx[ng-1]) io += 1; if (y >= a->cy[ng-1]) io += 2; io += 4; break;
        }
        int ch = CH(a, io, ng);
        if (ch == 0) break;
        ng = ch;
    }
    return a->ileaf[ng-1];
}
/* amr_build_tree: :460-569 (insert leaves in file order, creating internal nodes on the way down) */
static void amr_build_tree(amr_t *a, int nleaf, const double *xl, const double *yl, const double *zl, const int *lev, double lo, double hi) {
    int cap = nleaf * 2 > 16 ? nleaf * 2 : 16, nc = 1;
    a->parent = calloc(cap, 4); a->children = calloc((size_t)8 * cap, 4); a->level = calloc(cap, 4); a->ileaf = calloc(cap, 4);
    a->cx = calloc(cap, 8); a->cy = calloc(cap, 8); a->cz = calloc(cap, 8); a->ch = calloc(cap, 8); a->icell_of_leaf = calloc(nleaf, 4);
    a->xmin = a->ymin = a->zmin = lo; a->xmax = a->ymax = a->zmax = hi; a->L_box = hi - lo;
    a->cx[0] = a->cy[0] = a->cz[0] = (lo + hi) * 0.5; a->ch[0] = a->L_box * 0.5;
    a->levelmax = 0;
    for (int il = 1; il <= nleaf; il++) {
        int ic = 1;
        for (int l = 0; l <= lev[il-1] - 1; l++) {
            int io = 1; if (xl[il-1] >= a->cx[ic-1]) io += 1; if (yl[il-1] >= a->cy[ic-1]) io += 2; if (zl[il-1] >= a->cz[ic-1]) io += 4;
            if (CH(a, io, ic) == 0) {
                nc++;
                if (nc > cap) { int nc2 = cap * 2;
                    a->parent = realloc(a->parent, nc2 * 4); a->children = realloc(a->children, (size_t)8 * nc2 * 4); a->level = realloc(a->level, nc2 * 4); a->ileaf = realloc(a->ileaf, nc2 * 4);
                    a->cx = realloc(a->cx, nc2 * 8); a->cy = realloc(a->cy, nc2 * 8); a->cz = realloc(a->cz, nc2 * 8); a->ch = realloc(a->ch, nc2 * 8);
                    memset(a->children + (size_t)8 * cap, 0, (size_t)8 * (nc2 - cap) * 4); memset(a->ileaf + cap, 0, (nc2 - cap) * 4); cap = nc2; }
                a->parent[nc-1] = ic; a->level[nc-1] = l + 1; a->ileaf[nc-1] = 0; a->ch[nc-1] = a->ch[ic-1] * 0.5;
                for (int q = 1; q <= 8; q++) CH(a, q, nc) = 0;
                int ix = (io - 1) % 2, iy = ((io - 1) / 2) % 2, iz = (io - 1) / 4;
                a->cx[nc-1] = a->cx[ic-1] + (2 * ix - 1) * a->ch[nc-1]; a->cy[nc-1] = a->cy[ic-1] + (2 * iy - 1) * a->ch[nc-1]; a->cz[nc-1] = a->cz[ic-1] + (2 * iz - 1) * a->ch[nc-1];
                CH(a, io, ic) = nc;
            }
            ic = CH(a, io, ic);
        }
        a->ileaf[ic-1] = il; a->icell_of_leaf[il-1] = ic;
        if (lev[il-1] > a->levelmax) a->levelmax = lev[il-1];
    }
    a->ncells = nc; a->nleaf = nleaf;
}
static int amr_is_ancestor(const amr_t *a, int anc, int desc) { int c = desc; while (c > 0) { c = a->parent[c-1]; if (c == anc) return 1; } return 0; }
/* amr_build_neighbors: :619-682 */
static void amr_build_neighbors(amr_t *a) {
    a->neighbor = calloc((size_t)6 * a->ncells, 4);
    for (int ic = 1; ic <= a->ncells; ic++) {
        double cx = a->cx[ic-1], cy = a->cy[ic-1], cz = a->cz[ic-1], hp = 2.0 * a->ch[ic-1]; int lv = a->level[ic-1];
        if (cx + hp <= a->xmax) NB(a, 1, ic) = amr_find_cell_at_level(a, cx + hp, cy, cz, lv);
        if (cx - hp >= a->xmin) NB(a, 2, ic) = amr_find_cell_at_level(a, cx - hp, cy, cz, lv);
        if (cy + hp <= a->ymax) NB(a, 3, ic) = amr_find_cell_at_level(a, cx, cy + hp, cz, lv);
        if (cy - hp >= a->ymin) NB(a, 4, ic) = amr_find_cell_at_level(a, cx, cy - hp, cz, lv);
        if (cz + hp <= a->zmax) NB(a, 5, ic) = amr_find_cell_at_level(a, cx, cy, cz + hp, lv);
        if (cz - hp >= a->zmin) NB(a, 6, ic) = amr_find_cell_at_level(a, cx, cy, cz - hp, lv);
        for (int f = 1; f <= 6; f++) { int ng = NB(a, f, ic); if (ng > 0 && ng != ic && amr_is_ancestor(a, ng, ic)) NB(a, f, ic) = 0; }
    }
}
/* generic AMR text reader: read_generic_amr.f90:112-185 ("NLEAF n" / "BOXLEN L" or legacy "n L"; rows x y z level nH T vx vy vz) */
static int amr_read_text(const char *fn, int *nleaf, double *boxlen, double **col) {
    FILE *f = fopen(fn, "r"); if (!f) { snprintf(g_err, sizeof g_err, "generic_amr_read: cannot open file: %s", fn); return -1; }
    char line[1024]; int n = 0; double bl = -1.0;
    while (fgets(line, sizeof line, f)) {
        char *q = line; while (*q == ' ' || *q == '\t') q++;
        if (*q == 0 || *q == '\n' || *q == '#') continue;
        if (!strncmp(q, "BOXLEN", 6) || !strncmp(q, "boxlen", 6)) bl = atof(q + 6);
        else if (!strncmp(q, "NLEAF", 5) || !strncmp(q, "nleaf", 5)) n = atoi(q + 5);
        else if (sscanf(q, "%d %lf", &n, &bl) != 2) { fclose(f); snprintf(g_err, sizeof g_err, "generic_amr_read: cannot parse text header"); return -1; }
        if (n > 0 && bl > 0.0) break;
    }
    if (n <= 0 || bl <= 0.0) { fclose(f); snprintf(g_err, sizeof g_err, "generic_amr_read: text header must provide NLEAF and BOXLEN"); return -1; }
    for (int c = 0; c < 9; c++) col[c] = malloc((size_t)n * 8);
    int m = 0;
    while (m < n && fgets(line, sizeof line, f)) {
        char *q = line; while (*q == ' ' || *q == '\t') q++;
        if (*q == 0 || *q == '\n' || *q == '#') continue;
        double v[9];
        if (sscanf(q, "%lf %lf %lf %lf %lf %lf %lf %lf %lf", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7], &v[8]) != 9) break;
        for (int c = 0; c < 9; c++) col[c][m] = v[c];
        m++;
    }
    fclose(f);
    if (m != n) { snprintf(g_err, sizeof g_err, "generic_amr_read: expected %d rows, got %d", n, m); return -1; }
    *nleaf = n; *boxlen = bl; return 0;
}
/* grid_create_amr: grid_mod_amr.f90:34-526 (text input, all hydrogen neutral, no dust model / emissivity) */
static int grid_create_amr(orc_sim *s) {
    int nleaf; double boxlen, *col[9];
    if (amr_read_text(s->amr_file, &nleaf, &boxlen, col)) return -1;
    int *lev = malloc(nleaf * sizeof(int)); for (int i = 0; i < nleaf; i++) lev[i] = (int)col[3][i];
    amr_t *a = &s->amr;
    amr_build_tree(a, nleaf, col[0], col[1], col[2], lev, -0.5 * boxlen, 0.5 * boxlen);      /* text files: centred box (:179-181) */
    amr_build_neighbors(a);
    if (s->geometry == 0) { s->rmax = a->L_box * 0.5; s->xmax = s->ymax = s->zmax = s->rmax; }   /* :202-216 */
    else { s->rmax = -1.0; s->xmax = a->xmax; s->ymax = a->ymax; s->zmax = a->zmax; }
    if (s->source_rmax < 0.0) s->source_rmax = s->rmax;
    /* leaf tables live in the Cartesian arrays with nx = nleaf, ny = nz = 1 (photon%icell = leaf index) */
    s->nx = nleaf; s->ny = 1; s->nz = 1; s->xmin = a->xmin; s->ymin = a->ymin; s->zmin = a->zmin; s->dx = s->dy = s->dz = a->L_box;
    s->xface = calloc(2, 8); s->yface = calloc(2, 8); s->zface = calloc(2, 8);
    s->vfx = calloc(nleaf, 8); s->vfy = calloc(nleaf, 8); s->vfz = calloc(nleaf, 8);
    s->Dfreq = malloc(nleaf * 8); s->voigt_a = malloc(nleaf * 8); s->rhokap = malloc(nleaf * 8); s->rhokapD = calloc(nleaf, 8);
    s->Dfreq_ref = vtherm_total(s, s->temperature) / (s->wavelength0 * UM2KM);              /* :219-221 */
    s->voigt_amean = (s->damping / FOURPI) / s->Dfreq_ref;
    for (int il = 0; il < nleaf; il++) {                                                    /* :236-318 */
        double T = fmax(col[5][il], 10.0), vt = vtherm_total(s, T);
        s->Dfreq[il] = vt / (s->wavelength0 * UM2KM); s->voigt_a[il] = (s->damping / FOURPI) / s->Dfreq[il];
        s->rhokap[il] = col[4][il] * s->cross0 / s->Dfreq[il] * s->distance2cm;
        if (s->DGR > 0.0) s->rhokapD[il] = col[4][il] * s->cext_dust * s->DGR * s->distance2cm;
        s->vfx[il] = col[6][il] / vt; s->vfy[il] = col[7][il] / vt; s->vfz[il] = col[8][il] / vt;
    }
    for (int c = 0; c < 9; c++) free(col[c]); free(lev);
    if (s->cone_opening > 0.0 && s->cone_opening < 90.0) {                                  /* Step 4b :336-350: biconical mask by leaf centre */
        const double cos_cone = cos(s->cone_opening * (PI / 180.0));
        for (int il = 0; il < nleaf; il++) { int ic = a->icell_of_leaf[il]; double rr = sqrt(a->cx[ic-1] * a->cx[ic-1] + a->cy[ic-1] * a->cy[ic-1] + a->cz[ic-1] * a->cz[ic-1]);
            if (rr > 0.0 && fabs(a->cz[ic-1]) / rr < cos_cone) { s->rhokap[il] = 0.0; if (s->DGR > 0.0) s->rhokapD[il] = 0.0; } }
    }
    /* Step 5 (:355-446): tau_homo and the +z pole traversal from the box centre */
    double opacity_sum = 0.0, nopac = 0.0, opac_length = boxlen / 2.0, tauhomo = 0.0, taupole, NHI_pole_raw = 0.0, tau_raw = 0.0, opac_norm = 1.0;
    for (int il = 0; il < nleaf; il++) { opacity_sum += s->rhokap[il] * orc_voigt(0.0, s->voigt_a[il]); if (s->rhokap[il] > 0.0) nopac += 1.0; }
    if (nopac > 0.0) tauhomo = (opacity_sum / nopac) * opac_length;
    { double xc = a->cx[0], yc = a->cy[0], zc = a->cz[0];
      for (int it = 0; it < 10000000 && zc < a->zmax; it++) {
          int il = amr_find_leaf(a, xc, yc, zc);
          if (il <= 0) break;                                    /* gaps do not occur in complete octrees */
          double t; int f; amr_cell_exit(a, a->icell_of_leaf[il-1], xc, yc, zc, 0.0, 0.0, 1.0, &t, &f);
          tau_raw += s->rhokap[il-1] * orc_voigt(0.0, s->voigt_a[il-1]) * t;
          NHI_pole_raw += s->rhokap[il-1] * s->Dfreq[il-1] / s->cross0 * t;
          zc += t; } }
    taupole = tau_raw > 0.0 ? tau_raw : tauhomo;
    if (s->taumax > 0.0) { if (taupole > 0.0) opac_norm = s->taumax / taupole; }
    else if (s->N_gasmax > 0.0) { if (NHI_pole_raw > 0.0) opac_norm = s->N_gasmax / NHI_pole_raw; }
    for (int il = 0; il < nleaf; il++) { s->rhokap[il] *= opac_norm; if (s->DGR > 0.0) s->rhokapD[il] *= opac_norm; }
    opacity_sum = 0.0; for (int il = 0; il < nleaf; il++) opacity_sum += s->rhokap[il] * orc_voigt(0.0, s->voigt_a[il]);
    if (nopac > 0.0) tauhomo = (opacity_sum / nopac) * opac_length;
    taupole = s->taumax > 0.0 ? s->taumax : tauhomo;
    s->N_gasmax = NHI_pole_raw * opac_norm; s->N_gashomo = s->N_gasmax; s->tauhomo = tauhomo; s->taumax = taupole;
    /* Step 6 core-skip (:479-492) */
    double atau0 = s->voigt_amean * s->tauhomo, at = atau0 / (boxlen / (2.0 * a->ch[0]));
    if (at <= 1.0) { s->xcrit = 0.0; s->xcrit2 = 0.0; } else { double xi = (at <= 60.0) ? 0.6 : 1.4, chi = (at <= 60.0) ? 1.2 : 0.6; s->xcrit = 0.02 * exp(xi * pow(log(at), chi)); s->xcrit2 = s->xcrit * s->xcrit; }
    /* amr_setup_freq_grid (:726-800): same rules as the Cartesian grid */
    s->atau3 = pow(s->voigt_amean * s->tauhomo, 1.0 / 3.0);
    if (isfinite(s->wavelength_min) && isfinite(s->wavelength_max)) {               /* :1457-1461 */
        double vt0 = vtherm_total(s, s->temperature), speedc = 2.99792458e5;
        s->par_xfreq_min = -(s->wavelength_max - s->wavelength0 * 1e4) / (s->wavelength0 * 1e4) * (speedc / vt0);
        s->par_xfreq_max = -(s->wavelength_min - s->wavelength0 * 1e4) / (s->wavelength0 * 1e4) * (speedc / vt0);
    } else
    if (isfinite(s->velocity_min) && isfinite(s->velocity_max)) {                   /* :1462-1467 */
        double vt0 = vtherm_total(s, s->temperature);
        s->par_xfreq_min = -s->velocity_max / vt0; s->par_xfreq_max = -s->velocity_min / vt0;
    }
    if (!(isfinite(s->par_xfreq_max) && isfinite(s->par_xfreq_min))) {
        double xscale = (s->taumax <= 5e1) ? 25.0 : (s->taumax <= 5e2) ? 14.0 : (s->taumax <= 5e3) ? 10.0 : 5.0, vt = vtherm_total(s, s->temperature);
        if (s->Vexp == 0.0) { s->par_xfreq_max = floor(xscale * s->atau3) + 1.0; s->par_xfreq_min = -(floor(xscale * s->atau3 + s->DnuHK_Hz / s->Dfreq_ref) + 1.0); }
        else if (s->Vexp > 0.0) { s->par_xfreq_max = floor(xscale * s->atau3) + 1.0; s->par_xfreq_min = -(floor(xscale * s->atau3 + fabs(s->Vexp) / vt + s->DnuHK_Hz / s->Dfreq_ref) + 1.0); }
        else { s->par_xfreq_max = floor(xscale * s->atau3 + fabs(s->Vexp) / vt) + 1.0; s->par_xfreq_min = -(floor(xscale * s->atau3 + s->DnuHK_Hz / s->Dfreq_ref) + 1.0); }
        if (s->spectral_type == 2) { xscale = 4.0 * xscale;                         /* grid_mod_amr.f90:773-778 */
            s->par_xfreq_max = floor(xscale * s->atau3 + fabs(s->Vexp) / vt) + 1.0; s->par_xfreq_min = -(floor(xscale * s->atau3 + fabs(s->Vexp) / vt + s->DnuHK_Hz / s->Dfreq_ref) + 1.0); }
    }
    s->dxfreq = (s->par_xfreq_max - s->par_xfreq_min) / s->nxfreq; s->xfreq_max = s->par_xfreq_max; s->xfreq_min = s->par_xfreq_min;
    if (s->line_type == 8) {                                    /* car_setup_freq_grid :1518-1529 / amr_setup_freq_grid: the band-2 (H-alpha) axis */
        s->nxfreq_Ha = s->par_nxfreq_Ha > 0 ? s->par_nxfreq_Ha : s->nxfreq;
        if (s->par_xfreq_max_Ha > 0.0) { s->xfreq_max_Ha = s->par_xfreq_max_Ha; s->xfreq_min_Ha = -s->par_xfreq_max_Ha; } else { s->xfreq_min_Ha = s->xfreq_min; s->xfreq_max_Ha = s->xfreq_max; }
        s->dxfreq_Ha = (s->xfreq_max_Ha - s->xfreq_min_Ha) / s->nxfreq_Ha;
        s->Jout_Ha = calloc(s->nxfreq_Ha, 8); s->Jabs_Ha = calloc(s->nxfreq_Ha, 8);
    }
    s->Jout = calloc(s->nxfreq, 8); s->Jin = calloc(s->nxfreq, 8); s->Jabs = calloc(s->nxfreq, 8); s->Jmu = calloc((size_t)s->nxfreq * (s->nmu > 0 ? s->nmu : 1), 8);
    s->raytrace_variant = 3;
    return 0;
}
/* create_JPa_amr_mem: grid_mod_amr.f90:543-724 (position-binned deposits, gas volume per bin by ns^3 sub-sampling) */
static int create_JPa_amr_mem(orc_sim *s) {
    if (!(s->calc_J || s->calc_P || s->calc_Pnew)) return 0;
    amr_t *a = &s->amr;
    double rmax = s->rmax; if (rmax <= 0.0) rmax = a->L_box * 0.5;
    int nr = 1; { int e = a->levelmax - 1; if (e < 0) e = 0; nr = 1 << e; if (nr > 512) nr = 512; if (nr < 1) nr = 1; }
    int nz = 1; { int e = a->levelmax; if (e < 0) e = 0; nz = 1 << e; if (nz > 512) nz = 512; }
    a->nr_JPa = nr; a->nz_JPa = nz; a->rmax_JPa = rmax; a->dr_JPa = rmax / nr; a->dz_JPa = (a->zmax - a->zmin) / nz;
    int nadd = s->xyz_symmetry ? 8 : (s->xy_symmetry ? 4 : 1);              /* :577-588: the periodic slab does not fold */
    if (s->geometry_JPa == 1) s->nbins = nr; else if (s->geometry_JPa == -1) s->nbins = nz; else s->nbins = a->nleaf;
    s->ncount = calloc(s->nbins, 8);                             /* here: gas volume per bin (vol_sph / vol_plane / vol_leaf) */
    for (int il = 1; il <= a->nleaf; il++) {
        int ic = a->icell_of_leaf[il-1]; double hc = a->ch[ic-1], vleaf = pow(2.0 * hc, 3) * nadd;
        if (s->geometry_JPa == 3) { s->ncount[il-1] = vleaf; continue; }
        if (s->rhokap[il-1] <= 0.0) continue;
        if (s->geometry_JPa == 1) {
            int ns = (int)lround(2.0 * hc / a->dr_JPa); if (ns < 2) ns = 2; if (ns > 8) ns = 8;
            double wsub = vleaf / (double)(ns * ns * ns);
            for (int ks = 1; ks <= ns; ks++) for (int js = 1; js <= ns; js++) for (int is = 1; is <= ns; is++) {
                double xs = a->cx[ic-1] + (2.0 * is - ns - 1.0) * hc / ns, ys = a->cy[ic-1] + (2.0 * js - ns - 1.0) * hc / ns, zs = a->cz[ic-1] + (2.0 * ks - ns - 1.0) * hc / ns;
                int ir = (int)floor(sqrt(xs * xs + ys * ys + zs * zs) / a->dr_JPa) + 1;
                if (ir >= 1 && ir <= nr) s->ncount[ir-1] += wsub; }
        } else {
            double z1 = a->cz[ic-1] - hc, z2 = a->cz[ic-1] + hc, zlo = a->zmin, dz = a->dz_JPa;
            int iz1 = (int)floor((z1 - zlo) / dz) + 1, iz2 = (int)floor((z2 - zlo) / dz) + 1; if (iz1 < 1) iz1 = 1; if (iz2 > nz) iz2 = nz;
            for (int iz = iz1; iz <= iz2; iz++) { double ov = fmin(z2, zlo + iz * dz) - fmax(z1, zlo + (iz - 1) * dz); if (ov > 0.0) s->ncount[iz-1] += vleaf * ov / (2.0 * hc); }
        }
    }
    if (s->calc_J) s->J1 = calloc((size_t)s->nxfreq * s->nbins, 8);
    if (s->calc_P) s->P1 = calloc(s->nbins, 8);
    if (s->calc_P && s->line_type == 8) s->Pconv = calloc(s->nbins, 8);
    if (s->calc_Pnew) s->P1_new = calloc(s->nbins, 8);
    return 0;
}
/* amr_JPa_bin: octree_mod.f90:286-311 (geometry 1 and -1; 3 = the leaf itself) */
static int amr_JPa_bin(const orc_sim *s, int il, double x0, double y0, double z0) {
    const amr_t *a = &s->amr;
    if (s->geometry_JPa == 3) return il;
    if (s->geometry_JPa == 1) { int ib = (int)floor(sqrt(x0 * x0 + y0 * y0 + z0 * z0) / a->dr_JPa) + 1; return (ib >= 1 && ib <= a->nr_JPa) ? ib : 0; }
    int ib = (int)floor((z0 - a->zmin) / a->dz_JPa) + 1; if (ib < 1) ib = 1; if (ib > a->nz_JPa) ib = a->nz_JPa; return ib;
}
/* add_to_J_amr / add_to_Pnew_amr: raytrace_amr.f90:848-912 (segment midpoint decides the bin) */
static inline void add_to_JPnew_amr(const orc_sim *s, acc_t *A, const photon_t *p, int il, double x0, double y0, double z0, double del, double dtauH) {
    if (s->rhokap[il-1] <= 0.0) return;
    int ib = amr_JPa_bin(s, il, x0, y0, z0); if (ib < 1) return;
    if (s->calc_J) { double xr = p->xfreq * (s->Dfreq[il-1] / s->Dfreq_ref); int ix = (int)floor((xr - s->xfreq_min) / s->dxfreq) + 1;
        if (ix >= 1 && ix <= s->nxfreq) A->J1[(size_t)(ix - 1) + (size_t)s->nxfreq * (ib - 1)] += del * p->wgt; }
    if (s->calc_Pnew) A->P1_new[ib-1] += dtauH * p->wgt / (s->rhokap[il-1] * s->Dfreq[il-1] / s->cross0);
}
/* amr_xcrit_local: octree_mod.f90:248-278 */
static void amr_xcrit_local(const orc_sim *s, int il, double x, double y, double z, double *xc, double *xc2) {
    if (s->core_skip_global) { *xc = s->xcrit; *xc2 = s->xcrit2; return; }
    *xc = 0.0; *xc2 = 0.0; if (il <= 0) return;
    const amr_t *a = &s->amr; int ic = a->icell_of_leaf[il-1]; double h = a->ch[ic-1];
    double dl = fmin(h - fabs(x - a->cx[ic-1]), fmin(h - fabs(y - a->cy[ic-1]), h - fabs(z - a->cz[ic-1])));
    if (dl <= 0.0) return;
    double atau = s->voigt_a[il-1] * s->rhokap[il-1] * dl;
    MRG(atau, 1.0, 65);
    if (atau > 1.0) { *xc = pow(atau, 1.0 / 3.0) / 5.0; *xc2 = (*xc) * (*xc); }
}
/* boundary handling shared by the three AMR marches (raytrace_amr.f90:176-201): periodic in x and y (wrap by the box size and look the leaf up
 * from the root), mirror planes of the +++ octant (xyz_symmetry) or the ++ quadrant (xy_symmetry); the leaf to go on in, 0 = the packet leaves */
static inline int amr_boundary(const orc_sim *s, int il, int iface, double *x, double *y, double z, double *kx, double *ky, double *kz, int *reflected) {
    const amr_t *a = &s->amr;
    if (s->xy_periodic) {
        switch (iface) {
        case 1: *x = *x - (a->xmax - a->xmin); break;
        case 2: *x = *x + (a->xmax - a->xmin); break;
        case 3: *y = *y - (a->ymax - a->ymin); break;
        case 4: *y = *y + (a->ymax - a->ymin); break;
        }
        return iface <= 4 ? amr_find_leaf(a, *x, *y, z) : 0;
    } else if (s->xyz_symmetry) {
        if (iface == 2) { *kx = -*kx; *reflected = 1; return il; }
        if (iface == 4) { *ky = -*ky; *reflected = 1; return il; }
        if (iface == 6) { *kz = -*kz; *reflected = 1; return il; }
    } else if (s->xy_symmetry) {
        if (iface == 2) { *kx = -*kx; *reflected = 1; return il; }
        if (iface == 4) { *ky = -*ky; *reflected = 1; return il; }
    }
    return 0;
}
static void add_to_Jmu(const orc_sim *s, acc_t *A, const photon_t *p, int ix);
/* raytrace_to_tau_amr: raytrace_amr.f90:77-259 */
static void raytrace_to_tau_amr(const orc_sim *s, acc_t *A, photon_t *p, double tau_in) {
    const amr_t *a = &s->amr;
    double x = p->x, y = p->y, z = p->z, kx = p->kx, ky = p->ky, kz = p->kz, tau = 0.0;
    int il = p->icell;
    if (il <= 0) { il = amr_find_leaf(a, x, y, z); if (il <= 0) { p->inside = 0; return; } }
    double u1 = s->vfx[il-1] * kx + s->vfy[il-1] * ky + s->vfz[il-1] * kz, u2;
    while (p->inside) {
        int ic = a->icell_of_leaf[il-1], iface; double t_exit;
        amr_cell_exit(a, ic, x, y, z, kx, ky, kz, &t_exit, &iface);
        double rhokapH, rhokap;
        if (p->iband != 2) {
            rhokapH = s->rhokap[il-1] * calc_voigt(s, p->xfreq, s->voigt_a[il-1], s->Dfreq[il-1]); rhokap = rhokapH;
            if (s->h2_on) rhokap = rhokap + s->rhokap[il-1] * h2_kappa_D(s, p->xfreq, s->Dfreq[il-1]);
            if (s->DGR > 0.0) rhokap = rhokap + s->rhokapD[il-1];
        } else { rhokapH = 0.0; rhokap = (s->DGR > 0.0) ? s->rhokapD[il-1] * s->R_Ha : 0.0; }
        A->n_cellsteps += 1.0;
        MRG(tau + t_exit * rhokap, tau_in, 43);
        if (tau + t_exit * rhokap >= tau_in) {
            double d_step = (rhokap > 0.0) ? (tau_in - tau) / rhokap : t_exit;
            if (s->nbins && p->iband != 2) add_to_JPnew_amr(s, A, p, il, x + 0.5 * d_step * kx, y + 0.5 * d_step * ky, z + 0.5 * d_step * kz, d_step, d_step * rhokapH);
            x = x + d_step * kx; y = y + d_step * ky; z = z + d_step * kz; tau = tau_in;
            break;
        }
        tau = tau + t_exit * rhokap; x = x + t_exit * kx; y = y + t_exit * ky; z = z + t_exit * kz;
        if (s->nbins && p->iband != 2) add_to_JPnew_amr(s, A, p, il, x - 0.5 * t_exit * kx, y - 0.5 * t_exit * ky, z - 0.5 * t_exit * kz, t_exit, t_exit * rhokapH);
        int il_new = amr_next_leaf(a, ic, iface, x, y, z), reflected = 0;
        if (il_new <= 0) il_new = amr_boundary(s, il, iface, &x, &y, z, &kx, &ky, &kz, &reflected);
        if (il_new <= 0) { p->inside = 0; break; }
        u2 = s->vfx[il_new-1] * kx + s->vfy[il_new-1] * ky + s->vfz[il_new-1] * kz;
        if (p->iband != 2) { if (reflected) p->xfreq = (p->xfreq + u1) - u2; else p->xfreq = (p->xfreq + u1) * s->Dfreq[il-1] / s->Dfreq[il_new-1] - u2; }
        u1 = u2; il = il_new;
    }
    p->x = x; p->y = y; p->z = z; p->kx = kx; p->ky = ky; p->kz = kz; p->icell = il; p->jcell = 1; p->kcell = 1;
    if (!p->inside && p->iband == 2) {                       /* :244-256 */
        p->xfreq_ref = p->xfreq;
        int ix = (int)floor((p->xfreq - s->xfreq_min_Ha) / s->dxfreq_Ha) + 1;
        if (ix >= 1 && ix <= s->nxfreq_Ha) A->Jout_Ha[ix-1] += p->wgt;
        A->W[3] += p->wgt;
    } else if (!p->inside) {
        u1 = s->vfx[il-1] * kx + s->vfy[il-1] * ky + s->vfz[il-1] * kz;
        p->xfreq = p->xfreq + u1;
        p->xfreq_ref = p->xfreq * (s->Dfreq[il-1] / s->Dfreq_ref);
        int ix = (int)floor((p->xfreq_ref - s->xfreq_min) / s->dxfreq) + 1;
        if (ix >= 1 && ix <= s->nxfreq) { A->Jout[ix-1] += p->wgt; if (s->save_Jmu) add_to_Jmu(s, A, p, ix); }
        if (s->line_type == 8) A->W[1] += p->wgt;
    }
}
/* raytrace_to_edge_amr (:265-344) and raytrace_to_dist_amr (:559-640, dist >= 0) */
static double raytrace_to_edge_amr(const orc_sim *s, const photon_t *p, double dist, double *nsteps) {
    const amr_t *a = &s->amr;
    double x = p->x, y = p->y, z = p->z, kx = p->kx, ky = p->ky, kz = p->kz, tau = 0.0, d = 0.0, xfreq = p->xfreq;
    int il = p->icell;
    if (il <= 0) { il = amr_find_leaf(a, x, y, z); if (il <= 0) return tau; }
    double u1 = s->vfx[il-1] * kx + s->vfy[il-1] * ky + s->vfz[il-1] * kz, u2;
    for (;;) {
        int ic = a->icell_of_leaf[il-1], iface; double t_exit;
        amr_cell_exit(a, ic, x, y, z, kx, ky, kz, &t_exit, &iface);
        double rhokap = s->rhokap[il-1] * calc_voigt(s, xfreq, s->voigt_a[il-1], s->Dfreq[il-1]);
        if (s->h2_on && dist < 0.0) rhokap = rhokap + s->rhokap[il-1] * h2_kappa_D(s, xfreq, s->Dfreq[il-1]);
        if (s->DGR > 0.0) rhokap = rhokap + s->rhokapD[il-1];
        if (nsteps) *nsteps += 1.0;
        if (dist >= 0.0 && d + t_exit >= dist) { tau = tau + (dist - d) * rhokap; return tau; }
        tau = tau + t_exit * rhokap;
        if (tau >= TAU_HUGE) return tau;
        d = d + t_exit; x = x + t_exit * kx; y = y + t_exit * ky; z = z + t_exit * kz;
        int il_new = amr_next_leaf(a, ic, iface, x, y, z), reflected = 0;
        if (il_new <= 0) il_new = amr_boundary(s, il, iface, &x, &y, z, &kx, &ky, &kz, &reflected);
        if (il_new <= 0) return tau;
        u2 = s->vfx[il_new-1] * kx + s->vfy[il_new-1] * ky + s->vfz[il_new-1] * kz;
        if (reflected) xfreq = (xfreq + u1) - u2; else xfreq = (xfreq + u1) * s->Dfreq[il-1] / s->Dfreq[il_new-1] - u2;
        u1 = u2; il = il_new;
    }
}


/* =====================================================================================
 * Clump medium: clump_mod.f90, grid_mod_clump.f90, raytrace_clump.f90, line_clump_mod.f90
 * (non-overlapping spheres of uniform interior in vacuum; uniform clump properties unless set per clump;
 *  radial profiles, file input and overlapping clumps are outside this path: SURVEY 8f-3)
 * photon%xfreq is in REF Doppler units (cl_Dfreq_ref) throughout; shifted by -/+ ulos_clump at clump entry/exit.
 * ===================================================================================== */
/* voigt_clump (line type 1): clump_mod.f90:130-170; kappa_clump :176-182; ulos_clump :184-190 */
static inline double voigt_clump(const orc_sim *s, double xfreq, int64_t icl) {
    const clumps_t *c = &s->cl; double xloc = xfreq * (c->Dfreq_ref / c->Dfreq[icl-1]);
    if (s->line_type == 2) { double Dnu = s->DnuHK_Hz / c->Dfreq[icl-1];
        return orc_voigt(xloc + Dnu, c->voigt_a[icl-1]) * (1.0 / 3.0) + orc_voigt(xloc, c->voigt_a[icl-1]) * (2.0 / 3.0); }
    if (s->line_type == 7) { double Dnu = s->delta_nu_HD_Hz / c->Dfreq[icl-1], xD = (xloc - Dnu) * s->ratio_Dfreq_HD;      /* clump_mod.f90:159-164 */
        return orc_voigt(xloc, c->voigt_a[icl-1]) + s->nD_over_nH * s->ratio_Dfreq_HD * orc_voigt(xD, c->voigt_a[icl-1] * s->ratio_voigta_HD); }
    if (s->line_type == 5 || s->line_type == 6) return calc_voigt(s, xloc, c->voigt_a[icl-1], c->Dfreq[icl-1]);     /* clump_mod.f90:151-158 = calc_voigt3 */
    return orc_voigt(xloc, c->voigt_a[icl-1]);
}
static inline double kappa_clump(const orc_sim *s, double xfreq, int64_t icl) {
    const clumps_t *c = &s->cl; double kap = c->rhokap[icl-1] * voigt_clump(s, xfreq, icl);
    if (s->DGR > 0.0) kap = kap + c->rhokapD[icl-1];
    return kap;
}
static inline double ulos_clump(const clumps_t *c, int64_t icl, double kx, double ky, double kz) {
    return (c->vx[icl-1] * kx + c->vy[icl-1] * ky + c->vz[icl-1] * kz) * (c->Dfreq[icl-1] / c->Dfreq_ref);
}
#define CGIDX(c,i,j,k) ((size_t)(i) + (size_t)(j) * (c)->cgx + (size_t)(k) * (c)->cgx * (c)->cgx)      /* 0-based cell -> 0-based slot of cg_start */

/* generate_clumps, uniform path: clump_mod.f90:897-1151 (RSA with linked-list grid; MT stream of rank 0) */
static int generate_clumps(orc_sim *s, rng_t *g) {
    clumps_t *c = &s->cl; int64_t N = c->N;
    int rg = (int)pow((double)N, 1.0 / 3.0) + 1; if (rg < 32) rg = 32; if (rg > 512) rg = 512;
    double rg_cell = fmax(2.0 * c->sphere_R / rg, 2.0 * c->radius_max);
    rg = (int)((2.0 * c->sphere_R) / rg_cell) + 1; if (rg < 2) rg = 2;
    rg_cell = (2.0 * c->sphere_R) / rg;
    c->rsa_grid = rg;
    double min_sep2 = (2.0 * c->radius_max) * (2.0 * c->radius_max);
    if (s->clump_fully_inside) {
        if (c->radius_max >= c->sphere_R) { snprintf(g_err, sizeof g_err, "clump_fully_inside: max clump radius >= sphere_R"); return -1; }
        if (c->r_min + 2.0 * c->radius_max > c->sphere_R) { snprintf(g_err, sizeof g_err, "clump_fully_inside: rmin + 2*cl_radius_max > rmax"); return -1; }
    }
    size_t ncell = (size_t)rg * rg * rg;
    int *head = malloc(ncell * sizeof(int)), *nxt = malloc((size_t)N * sizeof(int));
    for (size_t i = 0; i < ncell; i++) head[i] = -1;
    for (int64_t i = 0; i < N; i++) nxt[i] = -1;
    int64_t n_attempts = 0, icl = 0;
    double r_max_center = s->clump_fully_inside ? c->sphere_R - s->clump_radius : c->sphere_R;
    double r_min_center = s->clump_fully_inside ? c->r_min + s->clump_radius : c->r_min;
    double r_max_center2 = r_max_center * r_max_center, r_min_center2 = r_min_center * r_min_center;
    const double cos_cone = (s->cone_opening > 0.0 && s->cone_opening < 90.0) ? cos(s->cone_opening * (PI / 180.0)) : -1.0;      /* :970-975 */
    while (icl < N) {
        n_attempts++;
        double xc, yc, zc, d2;
        if (cos_cone > 0.0) {                                       /* direct sampling inside the bicone :1029-1048 */
            double r_trial, cos_theta, sin_theta;
            for (;;) {
                r_trial = pow(r_min_center * r_min_center * r_min_center + (r_max_center * r_max_center * r_max_center - r_min_center * r_min_center * r_min_center) * rnd(g), 1.0 / 3.0);
                cos_theta = cos_cone + (1.0 - cos_cone) * rnd(g);
                if (rnd(g) < 0.5) cos_theta = -cos_theta;
                sin_theta = sqrt(fmax(0.0, 1.0 - cos_theta * cos_theta));
                if (s->clump_fully_inside && r_trial > s->clump_radius) {
                    double q = s->clump_radius / r_trial, cmin = fabs(cos_theta) * sqrt(1.0 - q * q) - sin_theta * q;
                    if (cmin < cos_cone) continue;
                }
                break;
            }
            double phi = TWOPI * rnd(g);
            xc = r_trial * sin_theta * cos(phi); yc = r_trial * sin_theta * sin(phi); zc = r_trial * cos_theta;
        } else
        for (;;) {                                                  /* box rejection :1049-1056 */
            xc = (2.0 * rnd(g) - 1.0) * r_max_center; yc = (2.0 * rnd(g) - 1.0) * r_max_center; zc = (2.0 * rnd(g) - 1.0) * r_max_center;
            d2 = xc * xc + yc * yc + zc * zc;
            if (d2 <= r_max_center2 && d2 >= r_min_center2) break;
        }
        int ig = (int)((xc + c->sphere_R) / rg_cell), jg = (int)((yc + c->sphere_R) / rg_cell), kg = (int)((zc + c->sphere_R) / rg_cell);
        if (ig < 0) ig = 0; if (ig > rg - 1) ig = rg - 1; if (jg < 0) jg = 0; if (jg > rg - 1) jg = rg - 1; if (kg < 0) kg = 0; if (kg > rg - 1) kg = rg - 1;
        int overlap = 0;
        for (int k2 = (kg > 0 ? kg - 1 : 0); k2 <= (kg < rg - 1 ? kg + 1 : rg - 1) && !overlap; k2++)
            for (int j2 = (jg > 0 ? jg - 1 : 0); j2 <= (jg < rg - 1 ? jg + 1 : rg - 1) && !overlap; j2++)
                for (int i2 = (ig > 0 ? ig - 1 : 0); i2 <= (ig < rg - 1 ? ig + 1 : rg - 1) && !overlap; i2++) {
                    int jnb = head[(size_t)i2 + (size_t)j2 * rg + (size_t)k2 * rg * rg];
                    while (jnb > 0) {
                        double ddx = xc - c->x[jnb-1], ddy = yc - c->y[jnb-1], ddz = zc - c->z[jnb-1];
                        if (ddx * ddx + ddy * ddy + ddz * ddz < min_sep2) { overlap = 1; break; }
                        jnb = nxt[jnb-1];
                    }
                }
        if (overlap && !s->clump_allow_overlap) continue;            /* :1090 */
        icl++;
        c->x[icl-1] = xc; c->y[icl-1] = yc; c->z[icl-1] = zc;
        if (s->clump_sigma_v > 0.0) {                                /* :1126-1131 */
            c->vx[icl-1] = s->clump_sigma_v / c->vtherm[icl-1] * rand_gauss(g);
            c->vy[icl-1] = s->clump_sigma_v / c->vtherm[icl-1] * rand_gauss(g);
            c->vz[icl-1] = s->clump_sigma_v / c->vtherm[icl-1] * rand_gauss(g);
        }
        size_t cell = (size_t)ig + (size_t)jg * rg + (size_t)kg * rg * rg;
        nxt[icl-1] = head[cell]; head[cell] = (int)icl;
    }
    c->rsa_acceptance = (double)N / (double)n_attempts;
    free(head); free(nxt);
    return 0;
}
/* assign_clump_velocities_from_type: :1153-1264 */
static void assign_clump_velocities(orc_sim *s) {
    clumps_t *c = &s->cl;
    for (int64_t i = 0; i < c->N; i++) {
        double xc = c->x[i], yc = c->y[i], zc = c->z[i], rr = sqrt(xc * xc + yc * yc + zc * zc), vx = 0, vy = 0, vz = 0;
        switch (s->velocity_type) {
        case 1: vx = s->Vexp * xc / c->sphere_R; vy = s->Vexp * yc / c->sphere_R; vz = s->Vexp * zc / c->sphere_R; break;
        case 2: if (rr > 0.0) { vx = s->Vexp * xc / rr; vy = s->Vexp * yc / rr; vz = s->Vexp * zc / rr; } break;
        case 3: vx = s->Vx; vy = s->Vy; vz = s->Vz; break;
        case 4: vx = -s->Vrot * yc / c->sphere_R; vy = s->Vrot * xc / c->sphere_R; break;
        case 5: { double rc = sqrt(xc * xc + yc * yc);
                  if (rc > 0.0) { if (rc < s->rinner) { vx = -s->Vrot * yc / s->rinner; vy = s->Vrot * xc / s->rinner; } else { vx = -s->Vrot * yc / rc; vy = s->Vrot * xc / rc; } } } break;
        default: break; }
        c->vx[i] += vx / c->vtherm[i]; c->vy[i] += vy / c->vtherm[i]; c->vz[i] += vz / c->vtherm[i];
    }
}
/* clump_cell_range :1345-1360 and build_clump_csr :1267-1342 */
static void clump_cell_range(const clumps_t *c, int64_t i, int r[6]) {
    double rcl = c->radius[i]; int m = c->cgx - 1;
    r[0] = (int)((c->x[i] - c->cg_min - rcl) * c->cg_inv); if (r[0] < 0) r[0] = 0;
    r[1] = (int)((c->x[i] - c->cg_min + rcl) * c->cg_inv); if (r[1] > m) r[1] = m;
    r[2] = (int)((c->y[i] - c->cg_min - rcl) * c->cg_inv); if (r[2] < 0) r[2] = 0;
    r[3] = (int)((c->y[i] - c->cg_min + rcl) * c->cg_inv); if (r[3] > m) r[3] = m;
    r[4] = (int)((c->z[i] - c->cg_min - rcl) * c->cg_inv); if (r[4] < 0) r[4] = 0;
    r[5] = (int)((c->z[i] - c->cg_min + rcl) * c->cg_inv); if (r[5] > m) r[5] = m;
}
static void build_clump_csr(clumps_t *c) {
    int cg = (int)pow((double)c->N, 1.0 / 3.0) + 1; if (cg < 32) cg = 32; if (cg > 512) cg = 512;
    c->cgx = cg; size_t ncells = (size_t)cg * cg * cg;
    c->cg_min = -(c->sphere_R + c->radius_max);
    c->cg_d = (2.0 * (c->sphere_R + c->radius_max)) / cg; c->cg_inv = 1.0 / c->cg_d;
    free(c->cg_start); free(c->cg_list);
    c->cg_start = malloc((ncells + 1) * sizeof(int32_t));
    int *cnt = calloc(ncells, sizeof(int)); int r[6];
    for (int64_t i = 0; i < c->N; i++) { clump_cell_range(c, i, r);
        for (int k = r[4]; k <= r[5]; k++) for (int j = r[2]; j <= r[3]; j++) for (int ii = r[0]; ii <= r[1]; ii++) cnt[CGIDX(c,ii,j,k)]++; }
    c->cg_start[0] = 1;
    for (size_t q = 0; q < ncells; q++) c->cg_start[q+1] = c->cg_start[q] + cnt[q];
    c->nreg = c->cg_start[ncells] - 1;
    c->cg_list = malloc((size_t)(c->nreg > 0 ? c->nreg : 1) * sizeof(int32_t));
    memset(cnt, 0, ncells * sizeof(int));
    for (int64_t i = 0; i < c->N; i++) { clump_cell_range(c, i, r);
        for (int k = r[4]; k <= r[5]; k++) for (int j = r[2]; j <= r[3]; j++) for (int ii = r[0]; ii <= r[1]; ii++) {
            size_t q = CGIDX(c,ii,j,k); c->cg_list[c->cg_start[q] - 1 + cnt[q]] = (int32_t)(i + 1); cnt[q]++; } }
    free(cnt);
}
/* check_has_overlap: clump_mod.f90:1544-1591 (any pair closer than the sum of its radii) */
static void check_has_overlap(clumps_t *c) {
    c->has_overlap = 0;
    for (int64_t i = 1; i <= c->N; i++) {
        int r[6]; clump_cell_range(c, i - 1, r);
        for (int k = r[4]; k <= r[5]; k++) for (int j = r[2]; j <= r[3]; j++) for (int ii = r[0]; ii <= r[1]; ii++) {
            size_t q = CGIDX(c,ii,j,k);
            for (int32_t ip = c->cg_start[q]; ip <= c->cg_start[q+1] - 1; ip++) {
                int64_t jc = c->cg_list[ip-1]; if (jc <= i) continue;
                double dx = c->x[jc-1] - c->x[i-1], dy = c->y[jc-1] - c->y[i-1], dz = c->z[jc-1] - c->z[i-1], rs = c->radius[i-1] + c->radius[jc-1];
                if (dx * dx + dy * dy + dz * dz < rs * rs) { c->has_overlap = 1; return; }
            }
        }
    }
}
/* init_clumps, uniform path: clump_mod.f90:646-891 */
static int init_clumps(orc_sim *s, double R_sphere) {
    clumps_t *c = &s->cl; memset(c, 0, sizeof *c);
    c->sphere_R = R_sphere;
    if (s->clump_radius <= 0.0) { snprintf(g_err, sizeof g_err, "ERROR: clump_radius must be > 0"); return -1; }
    c->r_min = s->rmin > 0.0 ? s->rmin : 0.0;
    if (c->r_min >= R_sphere) { snprintf(g_err, sizeof g_err, "ERROR: par%%rmin must be < par%%rmax for clump placement"); return -1; }
    c->radius_max = s->clump_radius;
    double temp_cl = s->clump_temperature < 0.0 ? s->temperature : s->clump_temperature, vth = vtherm_total(s, temp_cl);
    c->Dfreq_ref = vth / (s->wavelength0 * UM2KM); c->vtherm_ref = vth; c->temperature_ref = temp_cl;
    c->voigt_a_ref = (s->damping / FOURPI) / c->Dfreq_ref;
    double R = R_sphere, r0 = c->r_min, rc = c->radius_max, shell2 = R * R + R * r0 + r0 * r0;
    if (s->clump_N_clumps > 0.0) c->N = (int64_t)s->clump_N_clumps;
    else if (s->clump_f_vol > 0.0) c->N = (int64_t)llround(s->clump_f_vol * (R * R * R - r0 * r0 * r0) / (rc * rc * rc));
    else if (s->clump_f_cov > 0.0) c->N = (int64_t)llround((4.0 / 3.0) * s->clump_f_cov * shell2 / (rc * rc));
    else { snprintf(g_err, sizeof g_err, "ERROR: specify clump_N_clumps, clump_f_vol, or clump_f_cov"); return -1; }
    if (c->N <= 0) c->N = 1;
    c->fvol = (double)c->N * rc * rc * rc / fmax(R * R * R - r0 * r0 * r0, 2.2250738585072014e-308);
    c->fcov = 0.75 * (double)c->N * rc * rc / fmax(shell2, 2.2250738585072014e-308);
    if (s->clump_tau0 > 0.0) c->rhokap_ref = s->clump_tau0 / (orc_voigt(0.0, c->voigt_a_ref) * s->clump_radius);
    else if (s->clump_NHI > 0.0) c->rhokap_ref = s->clump_NHI * s->cross0 / (c->Dfreq_ref * s->clump_radius);
    else if (s->clump_nH > 0.0) c->rhokap_ref = s->clump_nH * s->cross0 * s->distance2cm / c->Dfreq_ref;
    else if (s->taumax > 0.0 || s->N_gasmax > 0.0) {
        double GF = (double)c->N * rc * rc * rc / fmax(shell2, 2.2250738585072014e-308);
        if (GF <= 0.0) { snprintf(g_err, sizeof g_err, "ERROR: cannot back-solve clump opacity"); return -1; }
        if (s->taumax > 0.0) c->rhokap_ref = s->taumax / (GF * orc_voigt(0.0, c->voigt_a_ref));
        else c->rhokap_ref = s->N_gasmax * s->cross0 / (GF * c->Dfreq_ref);
    } else { snprintf(g_err, sizeof g_err, "ERROR: specify clump_tau0, clump_NHI, clump_nH, taumax, or N_HImax"); return -1; }
    size_t n = (size_t)c->N;
    c->x = calloc(n, 8); c->y = calloc(n, 8); c->z = calloc(n, 8); c->vx = calloc(n, 8); c->vy = calloc(n, 8); c->vz = calloc(n, 8);
    c->radius = malloc(n * 8); c->radius2 = malloc(n * 8); c->rhokap = malloc(n * 8); c->rhokapD = calloc(n, 8);
    c->voigt_a = malloc(n * 8); c->Dfreq = malloc(n * 8); c->vtherm = malloc(n * 8);
    for (size_t i = 0; i < n; i++) { c->radius[i] = rc; c->radius2[i] = rc * rc; c->rhokap[i] = c->rhokap_ref; c->voigt_a[i] = c->voigt_a_ref;
        c->Dfreq[i] = c->Dfreq_ref; c->vtherm[i] = c->vtherm_ref;
        if (s->DGR > 0.0) c->rhokapD[i] = c->rhokap_ref * s->cext_dust * s->DGR * c->Dfreq_ref / s->cross0; }
    mt_t *mt = malloc(sizeof *mt); mt_init(mt, (int64_t)s->iseed);      /* rank 0's stream: init_random_seed(par%iseed), setup.f90:775 */
    rng_t g; memset(&g, 0, sizeof g); g.kind = 0; g.mt = mt;
    int rc_ = generate_clumps(s, &g);
    free(mt);
    if (rc_) return -1;
    if (s->velocity_type) assign_clump_velocities(s);
    build_clump_csr(c);
    if (s->clump_allow_overlap) check_has_overlap(c);           /* :889-891 */
    return 0;
}
/* compute_clump_scalars: :2316-2378 */
static void compute_clump_scalars(orc_sim *s) {
    const clumps_t *c = &s->cl; double shell2 = c->sphere_R * c->sphere_R + c->sphere_R * c->r_min + c->r_min * c->r_min, v[4] = {0, 0, 0, 0};
    if (shell2 <= 0.0) shell2 = c->sphere_R * c->sphere_R;
    for (int64_t i = 0; i < c->N; i++) {
        double voigt0 = orc_voigt(0.0, c->voigt_a[i]), r3 = c->radius[i] * c->radius[i] * c->radius[i];
        double di2 = fmax(c->x[i] * c->x[i] + c->y[i] * c->y[i] + c->z[i] * c->z[i], c->radius[i] * c->radius[i]);
        double wh = r3 / shell2, wm = r3 / (3.0 * di2);
        v[0] += c->rhokap[i] * voigt0 * wh; v[1] += c->rhokap[i] * voigt0 * wm;
        v[2] += c->rhokap[i] * c->Dfreq[i] * wh / s->cross0; v[3] += c->rhokap[i] * c->Dfreq[i] * wm / s->cross0;
    }
    s->tauhomo = v[0]; s->taumax = v[1]; s->N_gashomo = v[2]; s->N_gasmax = v[3];
}
/* grid_create_clump: grid_mod_clump.f90:33-131 */
static int grid_create_clump(orc_sim *s) {
    if (s->rmax <= 0.0) { snprintf(g_err, sizeof g_err, "ERROR: par%%rmax must be > 0 for clump medium"); return -1; }
    s->xmax = s->ymax = s->zmax = s->rmax;
    if (s->nx < 1) s->nx = 11; if (s->ny < 1) s->ny = 11; if (s->nz < 1) s->nz = 11;
    s->xyz_symmetry = 0; s->xy_periodic = 0;
    if (init_clumps(s, s->rmax)) return -1;
    compute_clump_scalars(s);
    if (grid_create(s)) return -1;
    size_t nc = (size_t)s->nx * s->ny * s->nz;
    for (size_t i = 0; i < nc; i++) { s->rhokap[i] = 0.0; s->Dfreq[i] = s->cl.Dfreq_ref; s->voigt_a[i] = s->cl.voigt_a_ref; s->vfx[i] = s->vfy[i] = s->vfz[i] = 0.0; }
    s->Dfreq_ref = s->cl.Dfreq_ref; s->voigt_amean = s->cl.voigt_a_ref;
    s->raytrace_variant = s->cl.has_overlap ? 7 : 4;               /* setup_clump_overlap: setup.f90:1053-1082 */
    return 0;
}
/* find_next_clump: clump_mod.f90:1390-1507 (DDA over the CSR grid, nearest entry first) */
static int find_next_clump(const clumps_t *c, double xp, double yp, double zp, double kx, double ky, double kz, int64_t skip_icl, double t_max,
                           double *t_entry, double *t_exit, int64_t *icl_found, double *ncell) {
    double best_te = HUGEST, best_tx2 = 0.0, d = 0.0; int64_t best_icl = 0; int cg = c->cgx;
    int ci = (int)((xp - c->cg_min) * c->cg_inv), cj = (int)((yp - c->cg_min) * c->cg_inv), ck = (int)((zp - c->cg_min) * c->cg_inv);
    if (ci > cg - 1) ci = cg - 1; if (ci < 0) ci = 0; if (cj > cg - 1) cj = cg - 1; if (cj < 0) cj = 0; if (ck > cg - 1) ck = cg - 1; if (ck < 0) ck = 0;
    int si, sj, sk; double tx, ty, tz, delx, dely, delz;
    if (kx > 0.0) { si = 1; tx = ((c->cg_min + (double)(ci + 1) * c->cg_d) - xp) / kx; delx = c->cg_d / kx; }
    else if (kx < 0.0) { si = -1; tx = ((c->cg_min + (double)ci * c->cg_d) - xp) / kx; delx = -c->cg_d / kx; } else { si = 0; tx = HUGEST; delx = HUGEST; }
    if (ky > 0.0) { sj = 1; ty = ((c->cg_min + (double)(cj + 1) * c->cg_d) - yp) / ky; dely = c->cg_d / ky; }
    else if (ky < 0.0) { sj = -1; ty = ((c->cg_min + (double)cj * c->cg_d) - yp) / ky; dely = -c->cg_d / ky; } else { sj = 0; ty = HUGEST; dely = HUGEST; }
    if (kz > 0.0) { sk = 1; tz = ((c->cg_min + (double)(ck + 1) * c->cg_d) - zp) / kz; delz = c->cg_d / kz; }
    else if (kz < 0.0) { sk = -1; tz = ((c->cg_min + (double)ck * c->cg_d) - zp) / kz; delz = -c->cg_d / kz; } else { sk = 0; tz = HUGEST; delz = HUGEST; }
    for (;;) {
        if (d > best_te || d > t_max) break;
        if (ncell) *ncell += 1.0;
        size_t q = CGIDX(c,ci,cj,ck);
        for (int32_t ip = c->cg_start[q]; ip <= c->cg_start[q+1] - 1; ip++) {
            int64_t icl = c->cg_list[ip-1];
            if (icl == skip_icl) continue;
            double rx = xp - c->x[icl-1], ry = yp - c->y[icl-1], rz = zp - c->z[icl-1];       /* ray_sphere_isect :1363-1387 */
            double b = rx * kx + ry * ky + rz * kz, disc = b * b - (rx * rx + ry * ry + rz * rz) + c->radius2[icl-1];
            if (g_mrg_on) mrg_note(b * b + c->radius2[icl-1], rx * rx + ry * ry + rz * rz, 47, 1.0);      /* a grazing ray: disc = 0 */
            if (disc < 0.0) continue;
            disc = sqrt(disc);
            double te = -b - disc, tx2 = -b + disc;
            if (g_mrg_on && tx2 > 0.0 && best_icl > 0) MRG(te, best_te, 48);
            if (tx2 > 0.0 && te < best_te) { best_te = te; best_tx2 = tx2; best_icl = icl; }       /* (te > 0 .or. icl /= skip) holds: skip was excluded above */
        }
        if (tx <= ty && tx <= tz) { d = tx; ci += si; if (ci < 0 || ci >= cg) break; tx += delx; }
        else if (ty <= tz)        { d = ty; cj += sj; if (cj < 0 || cj >= cg) break; ty += dely; }
        else                      { d = tz; ck += sk; if (ck < 0 || ck >= cg) break; tz += delz; }
    }
    if (best_icl > 0) MRG(best_te, t_max, 51);
    if (best_icl > 0 && best_te <= t_max) { *t_entry = best_te; *t_exit = fmin(best_tx2, t_max); *icl_found = best_icl; return 1; }
    return 0;
}
/* clump_exit_dist :1511-1526, sphere_exit_dist :1530-1540 */
static inline double clump_exit_dist(const clumps_t *c, double xp, double yp, double zp, double kx, double ky, double kz, int64_t icl) {
    double rx = xp - c->x[icl-1], ry = yp - c->y[icl-1], rz = zp - c->z[icl-1], b = rx * kx + ry * ky + rz * kz;
    double disc = b * b - (rx * rx + ry * ry + rz * rz) + c->radius2[icl-1]; if (disc < 0.0) disc = 0.0;
    return fmax(0.0, -b + sqrt(disc));
}
static inline double sphere_exit_dist(const clumps_t *c, double xp, double yp, double zp, double kx, double ky, double kz) {
    double b = xp * kx + yp * ky + zp * kz, disc = b * b - (xp * xp + yp * yp + zp * zp) + c->sphere_R * c->sphere_R; if (disc < 0.0) disc = 0.0;
    return fmax(0.0, -b + sqrt(disc));
}
/* active_set_at_point (first hit): :1595-1634 -- the birth clump, generate_photon.f90:325-332 */
static int64_t clump_at_point(const clumps_t *c, double xp, double yp, double zp) {
    int cg = c->cgx;
    int ci = (int)((xp - c->cg_min) * c->cg_inv), cj = (int)((yp - c->cg_min) * c->cg_inv), ck = (int)((zp - c->cg_min) * c->cg_inv);
    if (ci > cg - 1) ci = cg - 1; if (ci < 0) ci = 0; if (cj > cg - 1) cj = cg - 1; if (cj < 0) cj = 0; if (ck > cg - 1) ck = cg - 1; if (ck < 0) ck = 0;
    for (int k = (ck > 0 ? ck - 1 : 0); k <= (ck < cg - 1 ? ck + 1 : cg - 1); k++)
        for (int j = (cj > 0 ? cj - 1 : 0); j <= (cj < cg - 1 ? cj + 1 : cg - 1); j++)
            for (int i = (ci > 0 ? ci - 1 : 0); i <= (ci < cg - 1 ? ci + 1 : cg - 1); i++) {
                size_t q = CGIDX(c,i,j,k);
                for (int32_t ip = c->cg_start[q]; ip <= c->cg_start[q+1] - 1; ip++) {
                    int64_t icl = c->cg_list[ip-1];
                    double rx = xp - c->x[icl-1], ry = yp - c->y[icl-1], rz = zp - c->z[icl-1];
                    if (rx * rx + ry * ry + rz * rz <= c->radius2[icl-1]) return icl;
                }
            }
    return 0;
}
static inline void update_cell_idx(const orc_sim *s, photon_t *p) {             /* raytrace_clump.f90:69-76 */
    int i = (int)floor((p->x - s->xmin) / s->dx) + 1, j = (int)floor((p->y - s->ymin) / s->dy) + 1, k = (int)floor((p->z - s->zmin) / s->dz) + 1;
    p->icell = i < 1 ? 1 : (i > s->nx ? s->nx : i); p->jcell = j < 1 ? 1 : (j > s->ny ? s->ny : j); p->kcell = k < 1 ? 1 : (k > s->nz ? s->nz : k);
}
static void add_to_Jmu(const orc_sim *s, acc_t *A, const photon_t *p, int ix);
static inline void clump_escape(const orc_sim *s, acc_t *A, const photon_t *p) {
    int ix = (int)floor((p->xfreq - s->xfreq_min) / s->dxfreq) + 1;
    if (ix >= 1 && ix <= s->nxfreq) { A->Jout[ix-1] += p->wgt; if (s->save_Jmu) add_to_Jmu(s, A, p, ix); }
}
/* raytrace_to_tau_clump: raytrace_clump.f90:83-201 */
static void raytrace_to_tau_clump(const orc_sim *s, acc_t *A, photon_t *p, double tau_in) {
    const clumps_t *c = &s->cl;
    double kx = p->kx, ky = p->ky, kz = p->kz, tau_rem = tau_in; int64_t last_icl = 0;
    while (p->inside) {
        if (p->icell_clump > 0) {
            int64_t icl = p->icell_clump;
            double t_seg = clump_exit_dist(c, p->x, p->y, p->z, kx, ky, kz, icl), kap = kappa_clump(s, p->xfreq, icl);
            A->n_cellsteps += 1.0;
            MRG(tau_rem, kap * t_seg, 46);
            if (tau_rem <= kap * t_seg) {
                double ds = tau_rem / fmax(kap, 2.2250738585072014e-308);
                p->x = p->x + ds * kx; p->y = p->y + ds * ky; p->z = p->z + ds * kz;
                update_cell_idx(s, p); return;
            }
            tau_rem = tau_rem - kap * t_seg;
            p->x = p->x + t_seg * kx; p->y = p->y + t_seg * ky; p->z = p->z + t_seg * kz;
            p->xfreq = p->xfreq + ulos_clump(c, icl, kx, ky, kz);
            last_icl = icl; p->icell_clump = 0;
            MRG(p->x * p->x + p->y * p->y + p->z * p->z, c->sphere_R * c->sphere_R, 50);
            if (p->x * p->x + p->y * p->y + p->z * p->z >= c->sphere_R * c->sphere_R) { p->inside = 0; update_cell_idx(s, p); p->xfreq_ref = p->xfreq; clump_escape(s, A, p); return; }
        } else {
            double t_sp = sphere_exit_dist(c, p->x, p->y, p->z, kx, ky, kz), te, tx2; int64_t icl_found;
            if (t_sp <= 0.0) { p->inside = 0; p->xfreq_ref = p->xfreq; clump_escape(s, A, p); return; }
            if (find_next_clump(c, p->x, p->y, p->z, kx, ky, kz, last_icl, t_sp, &te, &tx2, &icl_found, &A->n_cellsteps)) {
                te = fmax(0.0, te);
                p->x = p->x + te * kx; p->y = p->y + te * ky; p->z = p->z + te * kz;
                p->xfreq = p->xfreq - ulos_clump(c, icl_found, kx, ky, kz);
                last_icl = 0; p->icell_clump = (int)icl_found; update_cell_idx(s, p);
            } else {
                p->x = p->x + t_sp * kx; p->y = p->y + t_sp * ky; p->z = p->z + t_sp * kz;
                p->inside = 0; update_cell_idx(s, p); p->xfreq_ref = p->xfreq; clump_escape(s, A, p); return;
            }
        }
    }
}
/* raytrace_to_edge_clump (:204-267, tau_max < 0) and raytrace_to_edge_clump_capped (:494-541, peel-off, tau_max = 745.2) */
static double raytrace_to_edge_clump(const orc_sim *s, const photon_t *p0, double tau_max, double *nsteps) {
    const clumps_t *c = &s->cl;
    double tau = 0.0, xp = p0->x, yp = p0->y, zp = p0->z, kx = p0->kx, ky = p0->ky, kz = p0->kz, xf = p0->xfreq;
    int64_t icl_cur = p0->icell_clump, icl_found; int capped = tau_max >= 0.0;
    if (icl_cur > 0) {
        double t_seg = clump_exit_dist(c, xp, yp, zp, kx, ky, kz, icl_cur), kap = kappa_clump(s, xf, icl_cur);
        if (nsteps) *nsteps += 1.0;
        tau = tau + kap * t_seg;
        if (capped && tau >= tau_max) return tau;
        xp = xp + t_seg * kx; yp = yp + t_seg * ky; zp = zp + t_seg * kz;
        xf = xf + ulos_clump(c, icl_cur, kx, ky, kz);
        if (xp * xp + yp * yp + zp * zp >= c->sphere_R * c->sphere_R) return tau;
    }
    for (;;) {
        double t_sp = sphere_exit_dist(c, xp, yp, zp, kx, ky, kz), te, tx2;
        if (t_sp <= 0.0) break;
        if (!find_next_clump(c, xp, yp, zp, kx, ky, kz, icl_cur, t_sp, &te, &tx2, &icl_found, nsteps)) break;
        te = fmax(0.0, te);
        xp = xp + te * kx; yp = yp + te * ky; zp = zp + te * kz;
        double u_los = ulos_clump(c, icl_found, kx, ky, kz);
        xf = xf - u_los;
        double t_seg = clump_exit_dist(c, xp, yp, zp, kx, ky, kz, icl_found), kap = kappa_clump(s, xf, icl_found);
        if (nsteps) *nsteps += 1.0;
        tau = tau + kap * t_seg;
        if (capped && tau >= tau_max) return tau;
        xp = xp + t_seg * kx; yp = yp + t_seg * ky; zp = zp + t_seg * kz;
        xf = xf + u_los;
        icl_cur = icl_found;
        if (xp * xp + yp * yp + zp * zp >= c->sphere_R * c->sphere_R) break;
    }
    return tau;
}

/* ---- overlapping clumps (raytrace_clump.f90:606-1208).  photon%xfreq stays in the GLOBAL frame during the flight; the Doppler shift
 * into the owner clump's frame is applied by the scattering wrappers (scattering_car.f90:898-937). ---- */
#define OVL_MAX_EVT 2048
/* active_set_at_point: clump_mod.f90:1594-1634 (all clumps containing the point, 27 CSR cells, first-seen order) */
static int active_set_at_point(const clumps_t *c, double xp, double yp, double zp, int64_t *active) {
    int cg = c->cgx, n = 0;
    int ci = (int)((xp - c->cg_min) * c->cg_inv), cj = (int)((yp - c->cg_min) * c->cg_inv), ck = (int)((zp - c->cg_min) * c->cg_inv);
    if (ci > cg - 1) ci = cg - 1; if (ci < 0) ci = 0; if (cj > cg - 1) cj = cg - 1; if (cj < 0) cj = 0; if (ck > cg - 1) ck = cg - 1; if (ck < 0) ck = 0;
    for (int k = (ck > 0 ? ck - 1 : 0); k <= (ck < cg - 1 ? ck + 1 : cg - 1); k++)
        for (int j = (cj > 0 ? cj - 1 : 0); j <= (cj < cg - 1 ? cj + 1 : cg - 1); j++)
            for (int i = (ci > 0 ? ci - 1 : 0); i <= (ci < cg - 1 ? ci + 1 : cg - 1); i++) {
                size_t q = CGIDX(c,i,j,k);
                for (int32_t ip = c->cg_start[q]; ip <= c->cg_start[q+1] - 1; ip++) {
                    int64_t icl = c->cg_list[ip-1];
                    double rx = xp - c->x[icl-1], ry = yp - c->y[icl-1], rz = zp - c->z[icl-1];
                    MRG(rx * rx + ry * ry + rz * rz, c->radius2[icl-1], 52);
                    if (rx * rx + ry * ry + rz * rz <= c->radius2[icl-1]) {
                        int dup = 0; for (int m = 0; m < n; m++) if (active[m] == icl) { dup = 1; break; }
                        if (!dup && n < OVL_MAX_EVT) active[n++] = icl;
                    }
                }
            }
    return n;
}
/* collect_ray_events_overlap: clump_mod.f90:1637-1759 (every ENTER/EXIT along the ray up to t_max, sorted by t; stable insertion sort) */
static int collect_ray_events_overlap(const clumps_t *c, double xp, double yp, double zp, double kx, double ky, double kz, double t_max,
                                      double *ev_t, int64_t *ev_icl, int *ev_type, double *ncell) {
    int n_ev = 0, cg = c->cgx; double d = 0.0;
    int ci = (int)((xp - c->cg_min) * c->cg_inv), cj = (int)((yp - c->cg_min) * c->cg_inv), ck = (int)((zp - c->cg_min) * c->cg_inv);
    if (ci > cg - 1) ci = cg - 1; if (ci < 0) ci = 0; if (cj > cg - 1) cj = cg - 1; if (cj < 0) cj = 0; if (ck > cg - 1) ck = cg - 1; if (ck < 0) ck = 0;
    int si, sj, sk; double tx, ty, tz, delx, dely, delz;
    if (kx > 0.0) { si = 1; delx = c->cg_d / kx; tx = ((c->cg_min + (double)(ci + 1) * c->cg_d) - xp) / kx; }
    else if (kx < 0.0) { si = -1; delx = -c->cg_d / kx; tx = ((c->cg_min + (double)ci * c->cg_d) - xp) / kx; } else { si = 0; delx = HUGEST; tx = HUGEST; }
    if (ky > 0.0) { sj = 1; dely = c->cg_d / ky; ty = ((c->cg_min + (double)(cj + 1) * c->cg_d) - yp) / ky; }
    else if (ky < 0.0) { sj = -1; dely = -c->cg_d / ky; ty = ((c->cg_min + (double)cj * c->cg_d) - yp) / ky; } else { sj = 0; dely = HUGEST; ty = HUGEST; }
    if (kz > 0.0) { sk = 1; delz = c->cg_d / kz; tz = ((c->cg_min + (double)(ck + 1) * c->cg_d) - zp) / kz; }
    else if (kz < 0.0) { sk = -1; delz = -c->cg_d / kz; tz = ((c->cg_min + (double)ck * c->cg_d) - zp) / kz; } else { sk = 0; delz = HUGEST; tz = HUGEST; }
    for (;;) {
        if (d > t_max) break;
        if (ci < 0 || ci >= cg || cj < 0 || cj >= cg || ck < 0 || ck >= cg) break;
        if (ncell) *ncell += 1.0;
        size_t q = CGIDX(c,ci,cj,ck);
        for (int32_t ip = c->cg_start[q]; ip <= c->cg_start[q+1] - 1; ip++) {
            int64_t icl = c->cg_list[ip-1];
            double rx = xp - c->x[icl-1], ry = yp - c->y[icl-1], rz = zp - c->z[icl-1];       /* ray_sphere_isect :1362-1386 */
            double b = rx * kx + ry * ky + rz * kz, disc = b * b - (rx * rx + ry * ry + rz * rz) + c->radius2[icl-1];
            if (g_mrg_on) mrg_note(b * b + c->radius2[icl-1], rx * rx + ry * ry + rz * rz, 47, 1.0);
            if (disc < 0.0) continue;
            disc = sqrt(disc);
            double te = -b - disc, tx2 = -b + disc;
            if (g_mrg_on) { mrg_note(b, -disc, 53, 1.0); mrg_note(b, disc, 53, 1.0); MRG(te, t_max, 51); }      /* t_exit > 0, t_entry > 0: -b against -+ sqrt(disc) */
            if (!(tx2 > 0.0)) continue;                                             /* hit = (t_exit > 0); then t_exit <= 0 is skipped again */
            if (te > t_max) continue;
            int dup = 0; for (int ie = 0; ie < n_ev; ie++) if (ev_icl[ie] == icl) { dup = 1; break; }
            if (dup || n_ev + 2 > OVL_MAX_EVT) continue;
            if (te > 0.0) { ev_t[n_ev] = te; ev_icl[n_ev] = icl; ev_type[n_ev] = +1; n_ev++; }
            ev_t[n_ev] = fmin(tx2, t_max); ev_icl[n_ev] = icl; ev_type[n_ev] = -1; n_ev++;
        }
        if (tx <= ty && tx <= tz) { d = tx; tx = tx + delx; ci += si; }
        else if (ty <= tz)        { d = ty; ty = ty + dely; cj += sj; }
        else                      { d = tz; tz = tz + delz; ck += sk; }
    }
    for (int ie = 1; ie < n_ev; ie++) {
        double t = ev_t[ie]; int64_t ic = ev_icl[ie]; int ty_ = ev_type[ie]; int ii = ie - 1;
        if (g_mrg_on) for (int iq = 0; iq < ie; iq++) MRG(ev_t[iq], t, 54);                /* the order of two events along the ray */
        while (ii >= 0 && ev_t[ii] > t) { ev_t[ii+1] = ev_t[ii]; ev_icl[ii+1] = ev_icl[ii]; ev_type[ii+1] = ev_type[ii]; ii--; }
        ev_t[ii+1] = t; ev_icl[ii+1] = ic; ev_type[ii+1] = ty_;
    }
    return n_ev;
}
/* sum_kap_active: raytrace_clump.f90:621-636 (each clump at its own frame frequency, summed in the order of the active list) */
static double sum_kap_active(const orc_sim *s, const int64_t *active, int n, double xfreq_g, double kx, double ky, double kz) {
    double sum = 0.0;
    for (int m = 0; m < n; m++) { int64_t icl = active[m]; double x_cl = xfreq_g - ulos_clump(&s->cl, icl, kx, ky, kz); sum = sum + kappa_clump(s, x_cl, icl); }
    return sum;
}
/* sample_owner_clump: :644-664 (one draw; the total is re-summed for every candidate exactly as written there) */
static int64_t sample_owner_clump(const orc_sim *s, rng_t *g, const int64_t *active, int n, double xfreq_g, double kx, double ky, double kz) {
    double cumul = 0.0, r = rnd(g); int64_t owner = 0;
    for (int m = 0; m < n; m++) {
        int64_t icl = active[m]; double x_cl = xfreq_g - ulos_clump(&s->cl, icl, kx, ky, kz);
        cumul = cumul + kappa_clump(s, x_cl, icl); owner = icl;
        MRG(r * sum_kap_active(s, active, n, xfreq_g, kx, ky, kz), cumul, 55);
        if (r * sum_kap_active(s, active, n, xfreq_g, kx, ky, kz) <= cumul) return owner;
    }
    return owner;
}
static inline void ovl_apply_event(int64_t icl, int type, int64_t *aset, int *na) {           /* apply_event :769-786 */
    if (type == +1) { if (*na < OVL_MAX_EVT) { aset[*na] = icl; (*na)++; } }
    else for (int m = 0; m < *na; m++) if (aset[m] == icl) { aset[m] = aset[*na - 1]; (*na)--; return; }
}
/* raytrace_to_tau_clump_overlap: :668-788 */
static void raytrace_to_tau_clump_overlap(const orc_sim *s, acc_t *A, rng_t *g, photon_t *p, double tau_in) {
    const clumps_t *c = &s->cl;
    static __thread double ev_t[OVL_MAX_EVT]; static __thread int64_t ev_icl[OVL_MAX_EVT], active[OVL_MAX_EVT]; static __thread int ev_type[OVL_MAX_EVT];
    double kx = p->kx, ky = p->ky, kz = p->kz, xfreq_g = p->xfreq, tau_rem = tau_in;
    while (p->inside) {
        double t_sp = sphere_exit_dist(c, p->x, p->y, p->z, kx, ky, kz);
        if (t_sp <= 0.0) { p->inside = 0; update_cell_idx(s, p); p->xfreq_ref = xfreq_g; p->xfreq = xfreq_g; clump_escape(s, A, p); return; }
        int n_ev = collect_ray_events_overlap(c, p->x, p->y, p->z, kx, ky, kz, t_sp, ev_t, ev_icl, ev_type, NULL);
        int n_active = active_set_at_point(c, p->x, p->y, p->z, active);
        double t_cur = 0.0;
        for (int ie = 1; ie <= n_ev + 1; ie++) {
            double t_next = ie <= n_ev ? ev_t[ie-1] : t_sp; t_next = fmin(t_next, t_sp);
            double dt = t_next - t_cur;
            MRG(t_next, t_cur, 56);
            if (dt <= 0.0) { if (ie <= n_ev) ovl_apply_event(ev_icl[ie-1], ev_type[ie-1], active, &n_active); continue; }
            double kap_tot = sum_kap_active(s, active, n_active, xfreq_g, kx, ky, kz);
            A->n_cellsteps += 1.0;
            MRG(tau_rem, kap_tot * dt, 46);
            if (kap_tot > 0.0 && tau_rem <= kap_tot * dt) {
                double ds = tau_rem / kap_tot;
                p->x = p->x + (t_cur + ds) * kx; p->y = p->y + (t_cur + ds) * ky; p->z = p->z + (t_cur + ds) * kz;
                p->xfreq = xfreq_g;
                p->icell_clump = (int)sample_owner_clump(s, g, active, n_active, xfreq_g, kx, ky, kz);
                update_cell_idx(s, p); return;
            }
            tau_rem = tau_rem - kap_tot * dt; t_cur = t_next;
            if (ie <= n_ev) ovl_apply_event(ev_icl[ie-1], ev_type[ie-1], active, &n_active);
        }
        p->x = p->x + t_sp * kx; p->y = p->y + t_sp * ky; p->z = p->z + t_sp * kz;
        p->xfreq = xfreq_g; p->icell_clump = 0; p->inside = 0; update_cell_idx(s, p); p->xfreq_ref = xfreq_g; clump_escape(s, A, p); return;
    }
}
/* raytrace_to_edge_clump_overlap (:792-854, tau_max < 0) and _capped (:858-918, the peel-off march) */
static double raytrace_to_edge_clump_overlap(const orc_sim *s, const photon_t *p0, double tau_max, double *nsteps) {
    const clumps_t *c = &s->cl;
    static __thread double ev_t[OVL_MAX_EVT]; static __thread int64_t ev_icl[OVL_MAX_EVT], active[OVL_MAX_EVT]; static __thread int ev_type[OVL_MAX_EVT];
    double tau = 0.0, kx = p0->kx, ky = p0->ky, kz = p0->kz, xfreq_g = p0->xfreq; int capped = tau_max >= 0.0;
    double t_sp = sphere_exit_dist(c, p0->x, p0->y, p0->z, kx, ky, kz);
    if (t_sp <= 0.0) return tau;
    int n_ev = collect_ray_events_overlap(c, p0->x, p0->y, p0->z, kx, ky, kz, t_sp, ev_t, ev_icl, ev_type, NULL);
    int n_active = active_set_at_point(c, p0->x, p0->y, p0->z, active);
    double t_cur = 0.0;
    for (int ie = 1; ie <= n_ev + 1; ie++) {
        double t_next = ie <= n_ev ? ev_t[ie-1] : t_sp; t_next = fmin(t_next, t_sp);
        double dt = t_next - t_cur;
        if (dt > 0.0) {
            double kap_tot = sum_kap_active(s, active, n_active, xfreq_g, kx, ky, kz);
            if (nsteps) *nsteps += 1.0;
            tau = tau + kap_tot * dt;
            if (capped && tau >= tau_max) return tau;
        }
        t_cur = t_next;
        if (ie <= n_ev) ovl_apply_event(ev_icl[ie-1], ev_type[ie-1], active, &n_active);
    }
    return tau;
}

/* =====================================================================================
 * Ray marches: raytrace_car.f90
 * ===================================================================================== */
typedef struct { int icell, jcell, kcell, istep, jstep, kstep; double kx, ky, kz, tx, ty, tz, delx, dely, delz; } trav_t;

/* setup_traversal_car: raytrace_car.f90:12-87.  Returns 1 when out_of_grid. */
static int setup_traversal_car(const orc_sim *s, double xp, double yp, double zp, trav_t *t) {
    if (t->kx > 0.0) {
        if (t->icell > s->nx && s->xface[t->icell-1] <= xp) return 1;
        t->istep = 1; t->tx = (s->xface[t->icell] - xp) / t->kx; t->delx = s->dx / t->kx;
    } else if (t->kx < 0.0) {
        if (s->xface[t->icell-1] == xp) { if (t->icell > 1) t->icell -= 1; else return 1; }
        t->istep = -1; t->tx = (s->xface[t->icell-1] - xp) / t->kx; t->delx = -s->dx / t->kx;
    } else { t->istep = 0; t->tx = HUGEST; t->delx = HUGEST; }
    if (t->ky > 0.0) {
        if (t->jcell > s->ny && s->yface[t->jcell-1] <= yp) return 1;
        t->jstep = 1; t->ty = (s->yface[t->jcell] - yp) / t->ky; t->dely = s->dy / t->ky;
    } else if (t->ky < 0.0) {
        if (s->yface[t->jcell-1] == yp) { if (t->jcell > 1) t->jcell -= 1; else return 1; }
        t->jstep = -1; t->ty = (s->yface[t->jcell-1] - yp) / t->ky; t->dely = -s->dy / t->ky;
    } else { t->jstep = 0; t->ty = HUGEST; t->dely = HUGEST; }
    if (t->kz > 0.0) {
        if (t->kcell > s->nz && s->zface[t->kcell-1] <= zp) return 1;
        t->kstep = 1; t->tz = (s->zface[t->kcell] - zp) / t->kz; t->delz = s->dz / t->kz;
    } else if (t->kz < 0.0) {
        if (s->zface[t->kcell-1] == zp) { if (t->kcell > 1) t->kcell -= 1; else return 1; }
        t->kstep = -1; t->tz = (s->zface[t->kcell-1] - zp) / t->kz; t->delz = -s->dz / t->kz;
    } else { t->kstep = 0; t->tz = HUGEST; t->delz = HUGEST; }
    return 0;
}
/* xyz-symmetric set-up of one axis: raytrace_car.f90:1691-1721 (tau; strict: face == xp leaves) and
 * :631-661 (edge; face <= xp leaves).  Returns 1 when the ray leaves the grid at once. */
static int setup_axis_sym(int n, const double *face, double d, int i0, double xp, int strict_eq,
                          int *cell, int *step, double *k, double *t, double *del) {
    if (*k > 0.0) {
        if (*cell > n && (strict_eq ? (face[*cell-1] == xp) : (face[*cell-1] <= xp))) return 1;
        *step = 1; *t = (face[*cell] - xp) / *k; *del = d / *k;
    } else if (*k < 0.0) {
        *step = -1; *del = -d / *k;
        if (face[*cell-1] == xp) { if (*cell > 1) *cell -= 1; else { *cell = i0; *step = 1; *k = fabs(*k); } }
        if (*step == 1) *t = (face[*cell] - xp) / *k; else *t = (face[*cell-1] - xp) / *k;
    } else { *step = 0; *t = HUGEST; *del = HUGEST; }
    return 0;
}
/* one plain (open) axis, as the z axis of _car_xysym / _car_xyper / _car_zonly: the tau routines leave only when the start
 * sits exactly on the upper face (strict), the edge routines when it is on or beyond it */
static int setup_axis_plain(int n, const double *face, double d, double xp, int strict_eq, int *cell, int *step, double k, double *t, double *del) {
    if (k > 0.0) {
        if (*cell > n && (strict_eq ? (face[*cell-1] == xp) : (face[*cell-1] <= xp))) return 1;
        *step = 1; *t = (face[*cell] - xp) / k; *del = d / k;
    } else if (k < 0.0) {
        if (face[*cell-1] == xp) { if (*cell > 1) *cell -= 1; else return 1; }
        *step = -1; *t = (face[*cell-1] - xp) / k; *del = -d / k;
    } else { *step = 0; *t = HUGEST; *del = HUGEST; }
    return 0;
}
/* one periodic axis: raytrace_to_tau_car_xyper :2287-2311 (a start on the upper face re-enters at the lower one),
 * raytrace_to_edge_car_xyper :996-1017 (such a start leaves).  *xp is moved with the wrap. */
static int setup_axis_per(int n, const double *face, double d, double *xp, int tau_mode, int *cell, int *step, double k, double *t, double *del) {
    if (k > 0.0) {
        if (tau_mode) { if (*cell > n && face[*cell-1] == *xp) { *cell = 1; *xp = face[0]; } }
        else if (*cell > n && face[*cell-1] <= *xp) return 1;
        *step = 1; *t = (face[*cell] - *xp) / k; *del = d / k;
    } else if (k < 0.0) {
        if (face[*cell-1] == *xp) { if (*cell > 1) *cell -= 1; else { *cell = n; *xp = face[n]; } }
        *step = -1; *t = (face[*cell-1] - *xp) / k; *del = -d / k;
    } else { *step = 0; *t = HUGEST; *del = HUGEST; }
    return 0;
}
static inline int minloc3(double a, double b, double c) { int m = 1; double v = a; if (b < v) { m = 2; v = b; } if (c < v) m = 3; return m; }
static inline double udot(const orc_sim *s, size_t c, double kx, double ky, double kz) { return s->vfx[c] * kx + s->vfy[c] * ky + s->vfz[c] * kz; }

static void add_to_Jmu(const orc_sim *s, acc_t *A, const photon_t *p, int ix) {        /* raytrace_car.f90:4049-4064 */
    double mu = p->kz; if (s->xyz_symmetry) mu = fabs(mu);
    int imu = (int)floor((mu - s->mu_min) / s->dmu) + 1;
    if (imu < 1) imu = 1; if (imu > s->nmu) imu = s->nmu;
    A->Jmu[(size_t)(ix - 1) + (size_t)s->nxfreq * (imu - 1)] += p->wgt;
}

/* raytrace_to_tau_car (:1425-1648), _car_xyzsym (:1650-1949), _car_zonly (:2519-2675) */
static void raytrace_to_tau(const orc_sim *s, acc_t *A, photon_t *p, double tau_in, int variant) {
    if (variant == 7) { raytrace_to_tau_clump_overlap(s, A, A->g, p, tau_in); return; }
    if (variant == 3) { raytrace_to_tau_amr(s, A, p, tau_in); return; }
    if (variant == 4) { raytrace_to_tau_clump(s, A, p, tau_in); return; }
    trav_t t; double xp = p->x, yp = p->y, zp = p->z;
    t.kx = p->kx; t.ky = p->ky; t.kz = p->kz; t.icell = p->icell; t.jcell = p->jcell; t.kcell = p->kcell;
    if (variant == 0) { if (setup_traversal_car(s, xp, yp, zp, &t)) { p->inside = 0; return; } }
    else if (variant == 1) {
        if (setup_axis_sym(s->nx, s->xface, s->dx, s->i0, xp, 1, &t.icell, &t.istep, &t.kx, &t.tx, &t.delx)) { p->inside = 0; return; }
        if (setup_axis_sym(s->ny, s->yface, s->dy, s->j0, yp, 1, &t.jcell, &t.jstep, &t.ky, &t.ty, &t.dely)) { p->inside = 0; return; }
        if (setup_axis_sym(s->nz, s->zface, s->dz, s->k0, zp, 1, &t.kcell, &t.kstep, &t.kz, &t.tz, &t.delz)) { p->inside = 0; return; }
    } else if (variant == 5) {                                                       /* xysym :1985-2066 */
        if (setup_axis_sym(s->nx, s->xface, s->dx, s->i0, xp, 1, &t.icell, &t.istep, &t.kx, &t.tx, &t.delx)) { p->inside = 0; return; }
        if (setup_axis_sym(s->ny, s->yface, s->dy, s->j0, yp, 1, &t.jcell, &t.jstep, &t.ky, &t.ty, &t.dely)) { p->inside = 0; return; }
        if (setup_axis_plain(s->nz, s->zface, s->dz, zp, 1, &t.kcell, &t.kstep, t.kz, &t.tz, &t.delz)) { p->inside = 0; return; }
    } else if (variant == 6 || variant == 8) {                                       /* xyper :2287-2361, shear :2718-2770 */
        setup_axis_per(s->nx, s->xface, s->dx, &xp, 1, &t.icell, &t.istep, t.kx, &t.tx, &t.delx);
        setup_axis_per(s->ny, s->yface, s->dy, &yp, 1, &t.jcell, &t.jstep, t.ky, &t.ty, &t.dely);
        if (setup_axis_plain(s->nz, s->zface, s->dz, zp, 1, &t.kcell, &t.kstep, t.kz, &t.tz, &t.delz)) { p->inside = 0; return; }
    } else {                                                                         /* zonly :2558-2582 */
        t.istep = t.jstep = 0; t.tx = t.ty = HUGEST; t.delx = t.dely = HUGEST;
        if (setup_axis_plain(s->nz, s->zface, s->dz, zp, 1, &t.kcell, &t.kstep, t.kz, &t.tz, &t.delz)) { p->inside = 0; return; }
    }
    const int mirror_xy = (variant == 1 || variant == 5), mirror_z = (variant == 1), wrap_xy = (variant == 6 || variant == 8), shear = (variant == 8);
    double tau = 0.0, d = 0.0, del = 0.0, rhokap, rhokapH = 0.0;
    int iold = t.icell, jold = t.jcell, kold = t.kcell;
    double u1 = udot(s, IDX(s,iold,jold,kold), t.kx, t.ky, t.kz), u2;
    if (shear) u1 = s->vfx[IDX(s,iold,jold,kold)] * t.kx + (s->vfy[IDX(s,iold,jold,kold)] + p->vfy_shear) * t.ky + s->vfz[IDX(s,iold,jold,kold)] * t.kz;   /* :2808-2809 */
    while (p->inside) {
        size_t c = IDX(s,t.icell,t.jcell,t.kcell);
        if (p->iband != 2) {
            rhokapH = s->rhokap[c] * calc_voigt(s, p->xfreq, s->voigt_a[c], s->Dfreq[c]);
            rhokap = rhokapH;
            if (s->h2_on) rhokap = rhokap + s->rhokap[c] * h2_kappa_D(s, p->xfreq, s->Dfreq[c]);       /* :1490-1493 (kept out of rhokapH) */
            if (s->DGR > 0.0) rhokap = rhokap + s->rhokapD[c];
        } else { rhokapH = 0.0; rhokap = (s->DGR > 0.0) ? s->rhokapD[c] * s->R_Ha : 0.0; }            /* :1495-1503 band 2: dust only, scaled to 6563 A */
        A->n_cellsteps += 1.0;
        int idx = (variant == 2) ? 3 : minloc3(t.tx, t.ty, t.tz);
        double *tt = (idx == 1) ? &t.tx : (idx == 2) ? &t.ty : &t.tz;
        del = *tt - d; tau = tau + del * rhokap; d = *tt;
        MRG(tau, tau_in, 40);
        if (tau >= tau_in) {
            if (rhokap > 0.0) { double d_over = (tau - tau_in) / rhokap; d = d - d_over; del = del - d_over; }
            if (!mirror_xy) { xp = xp + d * t.kx; yp = yp + d * t.ky; zp = zp + d * t.kz; }
            break;
        }
        if (idx == 1) { t.icell += t.istep;
            if (mirror_xy && t.icell < 1) { t.icell = s->i0; t.istep = 1; t.kx = -t.kx; }
            else if (wrap_xy) { if (t.icell < 1) { t.icell = s->nx; if (shear) p->vfy_shear = p->vfy_shear - s->Omega; }            /* :2843-2852 */
                                if (t.icell > s->nx) { t.icell = 1; if (shear) p->vfy_shear = p->vfy_shear + s->Omega; } }
            else if (t.icell < 1 || t.icell > s->nx) { p->inside = 0; break; }
            t.tx += t.delx; }
        else if (idx == 2) { t.jcell += t.jstep;
            if (mirror_xy && t.jcell < 1) { t.jcell = s->j0; t.jstep = 1; t.ky = -t.ky; }
            else if (wrap_xy) { if (t.jcell < 1) t.jcell = s->ny; if (t.jcell > s->ny) t.jcell = 1; }
            else if (t.jcell < 1 || t.jcell > s->ny) { p->inside = 0; break; }
            t.ty += t.dely; }
        else { t.kcell += t.kstep;
            if (mirror_z && t.kcell < 1) { t.kcell = s->k0; t.kstep = 1; t.kz = -t.kz; }
            else if (t.kcell < 1 || t.kcell > s->nz) { p->inside = 0; break; }
            t.tz += t.delz; }
        size_t cn = IDX(s,t.icell,t.jcell,t.kcell), co = IDX(s,iold,jold,kold);
        if (s->nbins && p->iband != 2) add_to_JPnew(s, A, p, co, del, del * rhokapH);                   /* :1579-1584 */
        u2 = shear ? s->vfx[cn] * t.kx + (s->vfy[cn] + p->vfy_shear) * t.ky + s->vfz[cn] * t.kz : udot(s, cn, t.kx, t.ky, t.kz);
        if (p->iband != 2) p->xfreq = (p->xfreq + u1) * s->Dfreq[co] / s->Dfreq[cn] - u2;            /* band 2 carries a lab-frame frequency: :1587-1589 */
        iold = t.icell; jold = t.jcell; kold = t.kcell; u1 = u2;
    }
    if (!p->inside) { t.icell = iold; t.jcell = jold; t.kcell = kold; }
    if (s->nbins && p->iband != 2) add_to_JPnew(s, A, p, IDX(s,t.icell,t.jcell,t.kcell), del, del * rhokapH);   /* :1604-1609 last segment */
    if (!p->inside && p->iband == 2) {                                               /* :1625-1636 band-2 escape: no fluid shift, the H-alpha axis */
        p->xfreq_ref = p->xfreq;
        int ix = (int)floor((p->xfreq_ref - s->xfreq_min_Ha) / s->dxfreq_Ha) + 1;
        if (ix >= 1 && ix <= s->nxfreq_Ha) A->Jout_Ha[ix-1] += p->wgt;
        A->W[3] += p->wgt;
    } else
    if (!p->inside) {                                                                /* escape tally :1613-1623 */
        size_t c = IDX(s,t.icell,t.jcell,t.kcell);
        u1 = shear ? s->vfx[c] * t.kx + (s->vfy[c] + p->vfy_shear) * t.ky + s->vfz[c] * t.kz : udot(s, c, t.kx, t.ky, t.kz);
        p->xfreq = p->xfreq + u1;
        p->xfreq_ref = p->xfreq * (s->Dfreq[c] / s->Dfreq_ref);
        int ix = (int)floor((p->xfreq_ref - s->xfreq_min) / s->dxfreq) + 1;
        if (ix >= 1 && ix <= s->nxfreq) { A->Jout[ix-1] += p->wgt; if (s->save_Jmu) add_to_Jmu(s, A, p, ix); }
        if (s->line_type == 8) A->W[1] += p->wgt;                                    /* W_esc1 :1624 */
    }
    if (mirror_xy) {                                                                 /* :1935-1946, xysym :2238-2248 (no fold in z) */
        p->x = p->x + d * p->kx; p->y = p->y + d * p->ky; p->z = p->z + d * p->kz;
        if (p->x < s->xmin) p->x = -p->x; if (p->y < s->ymin) p->y = -p->y; if (mirror_z && p->z < s->zmin) p->z = -p->z;
        p->kx = t.kx; p->ky = t.ky; p->kz = t.kz;
        p->icell = t.icell; p->jcell = t.jcell; p->kcell = t.kcell;
    } else if (variant == 6 || variant == 8) {                                       /* :2509-2514, :2943-2949: back into the base period */
        p->x = xp - floor((xp - s->xmin) / s->xrange) * s->xrange; p->y = yp - floor((yp - s->ymin) / s->yrange) * s->yrange; p->z = zp;
        p->icell = t.icell; p->jcell = t.jcell; p->kcell = t.kcell;
    } else if (variant == 0) { p->x = xp; p->y = yp; p->z = zp; p->icell = t.icell; p->jcell = t.jcell; p->kcell = t.kcell; }
    else { p->x = xp; p->y = yp; p->z = zp; p->kcell = t.kcell; }                     /* zonly :2669-2673 */
}

/* raytrace_to_edge_car (:410-508), _car_xyzsym (:584-781), _car_zonly (:1138-1234); dist >= 0 selects
 * raytrace_to_dist_car (:89-195) on the plain grid */
static double raytrace_to_edge(const orc_sim *s, const photon_t *p, int variant, double dist, double *nsteps) {
    if (variant == 3) return raytrace_to_edge_amr(s, p, dist, nsteps);
    if (variant == 4) return raytrace_to_edge_clump(s, p, -1.0, nsteps);                 /* uncapped: first flight (run_simulation_mod.f90:165) */
    if (variant == 7) return raytrace_to_edge_clump_overlap(s, p, -1.0, nsteps);
    if (variant == 8) variant = 0;            /* setup_procedure assigns raytrace_to_tau only for the shearing box (setup.f90:967-969): raytrace_to_edge stays raytrace_to_edge_car (:941) */
    trav_t t; double xp = p->x, yp = p->y, zp = p->z, tau = 0.0, d = 0.0;
    t.kx = p->kx; t.ky = p->ky; t.kz = p->kz; t.icell = p->icell; t.jcell = p->jcell; t.kcell = p->kcell;
    if (variant == 0) { if (setup_traversal_car(s, xp, yp, zp, &t)) return tau; }
    else if (variant == 1) {
        if (setup_axis_sym(s->nx, s->xface, s->dx, s->i0, xp, 0, &t.icell, &t.istep, &t.kx, &t.tx, &t.delx)) return tau;
        if (setup_axis_sym(s->ny, s->yface, s->dy, s->j0, yp, 0, &t.jcell, &t.jstep, &t.ky, &t.ty, &t.dely)) return tau;
        if (setup_axis_sym(s->nz, s->zface, s->dz, s->k0, zp, 0, &t.kcell, &t.kstep, &t.kz, &t.tz, &t.delz)) return tau;
    } else if (variant == 5) {                                                       /* raytrace_to_edge_car_xysym :783-969 */
        if (setup_axis_sym(s->nx, s->xface, s->dx, s->i0, xp, 0, &t.icell, &t.istep, &t.kx, &t.tx, &t.delx)) return tau;
        if (setup_axis_sym(s->ny, s->yface, s->dy, s->j0, yp, 0, &t.jcell, &t.jstep, &t.ky, &t.ty, &t.dely)) return tau;
        if (setup_axis_plain(s->nz, s->zface, s->dz, zp, 0, &t.kcell, &t.kstep, t.kz, &t.tz, &t.delz)) return tau;
    } else if (variant == 6) {                                                       /* raytrace_to_edge_car_xyper :971-1136 */
        if (setup_axis_per(s->nx, s->xface, s->dx, &xp, 0, &t.icell, &t.istep, t.kx, &t.tx, &t.delx)) return tau;
        if (setup_axis_per(s->ny, s->yface, s->dy, &yp, 0, &t.jcell, &t.jstep, t.ky, &t.ty, &t.dely)) return tau;
        if (setup_axis_plain(s->nz, s->zface, s->dz, zp, 0, &t.kcell, &t.kstep, t.kz, &t.tz, &t.delz)) return tau;
    } else {
        t.istep = t.jstep = 0; t.tx = t.ty = HUGEST; t.delx = t.dely = HUGEST;
        if (setup_axis_plain(s->nz, s->zface, s->dz, zp, 0, &t.kcell, &t.kstep, t.kz, &t.tz, &t.delz)) return tau;
    }
    const int mirror_xy = (variant == 1 || variant == 5), mirror_z = (variant == 1), wrap_xy = (variant == 6);
    int iold = t.icell, jold = t.jcell, kold = t.kcell;
    double u1 = udot(s, IDX(s,iold,jold,kold), t.kx, t.ky, t.kz), u2, xfreq = p->xfreq, rhokap;
    for (;;) {
        size_t c = IDX(s,t.icell,t.jcell,t.kcell);
        rhokap = s->rhokap[c] * calc_voigt(s, xfreq, s->voigt_a[c], s->Dfreq[c]);
        if (s->h2_on && dist < 0.0) rhokap = rhokap + s->rhokap[c] * h2_kappa_D(s, xfreq, s->Dfreq[c]);      /* H2 peel attenuation: _edge_car :473, _zonly :1216 (not in raytrace_to_dist) */
        if (s->DGR > 0.0) rhokap = rhokap + s->rhokapD[c];
        if (nsteps) *nsteps += 1.0;
        int idx = (variant == 2) ? 3 : minloc3(t.tx, t.ty, t.tz);
        double *tt = (idx == 1) ? &t.tx : (idx == 2) ? &t.ty : &t.tz;
        if (dist >= 0.0) { if (*tt >= dist) { tau = tau + (dist - d) * rhokap; break; } }
        tau = tau + (*tt - d) * rhokap; d = *tt;
        if (idx == 1) { t.icell += t.istep;
            if (mirror_xy && t.icell < 1) { t.icell = s->i0; t.istep = 1; t.kx = -t.kx; }
            else if (wrap_xy) { if (t.icell < 1) t.icell = s->nx; if (t.icell > s->nx) t.icell = 1; }
            else if (t.icell < 1 || t.icell > s->nx) break;
            t.tx += t.delx; }
        else if (idx == 2) { t.jcell += t.jstep;
            if (mirror_xy && t.jcell < 1) { t.jcell = s->j0; t.jstep = 1; t.ky = -t.ky; }
            else if (wrap_xy) { if (t.jcell < 1) t.jcell = s->ny; if (t.jcell > s->ny) t.jcell = 1; }
            else if (t.jcell < 1 || t.jcell > s->ny) break;
            t.ty += t.dely; }
        else { t.kcell += t.kstep;
            if (mirror_z && t.kcell < 1) { t.kcell = s->k0; t.kstep = 1; t.kz = -t.kz; }
            else if (t.kcell < 1 || t.kcell > s->nz) break;
            t.tz += t.delz; }
        if (tau >= TAU_HUGE) break;
        size_t cn = IDX(s,t.icell,t.jcell,t.kcell), co = IDX(s,iold,jold,kold);
        u2 = udot(s, cn, t.kx, t.ky, t.kz);
        xfreq = (xfreq + u1) * s->Dfreq[co] / s->Dfreq[cn] - u2;
        iold = t.icell; jold = t.jcell; kold = t.kcell; u1 = u2;
    }
    return tau;
}

/* car_xcrit_local: grid_mod_car.f90:1598-1629 */
static void car_xcrit_local(const orc_sim *s, int i, int j, int k, double x, double y, double z, double *xc, double *xc2) {
    if (s->use_amr_grid) { amr_xcrit_local(s, i, x, y, z, xc, xc2); return; }
    if (s->core_skip_global) { *xc = s->xcrit; *xc2 = s->xcrit2; return; }
    *xc = 0.0; *xc2 = 0.0;
    if (i < 1 || j < 1 || k < 1) return;
    double dlx = fmin(x - s->xface[i-1], s->xface[i] - x), dly = fmin(y - s->yface[j-1], s->yface[j] - y), dlz = fmin(z - s->zface[k-1], s->zface[k] - z);
    double dl = fmin(dlx, fmin(dly, dlz));
    if (dl <= 0.0) return;
    double atau = s->voigt_a[IDX(s,i,j,k)] * s->rhokap[IDX(s,i,j,k)] * dl;
    MRG(atau, 1.0, 65);
    if (atau > 1.0) { *xc = pow(atau, 1.0 / 3.0) / 5.0; *xc2 = (*xc) * (*xc); }
}
double orc_xcrit_local(const orc_sim *s, int i, int j, int k, double x, double y, double z) { double a, b; car_xcrit_local(s, i, j, k, x, y, z, &a, &b); return a; }

/* vec2pix, RING scheme, 1-based pixel: healpix.f90:29-112 */
int orc_vec2pix(int nside, double vx, double vy, double vz) {
    const double halfpi = 1.570796326794896619231321691639751442099, twothird = 0.6666666666666666666666666666666666666666;
    double dnorm = sqrt(vx * vx + vy * vy + vz * vz), z = vz / dnorm, phi = 0.0;
    if (vx != 0.0 || vy != 0.0) phi = atan2(vy, vx);
    double za = fabs(z);
    if (phi < 0.0) phi = phi + TWOPI;
    double tt = phi / halfpi;
    int nl2 = 2 * nside, nl4 = 4 * nside, ncap = nl2 * (nside - 1), npix = 12 * nside * nside, ipix1;
    if (za <= twothird) {
        int jp = (int)(nside * (0.5 + tt - z * 0.75)), jm = (int)(nside * (0.5 + tt + z * 0.75));
        int ir = nside + 1 + jp - jm, kshift = 0;
        if (((ir % 2) + 2) % 2 == 0) kshift = 1;
        int ip = (int)((jp + jm - nside + kshift + 1) / 2) + 1;
        if (ip > nl4) ip = ip - nl4;
        ipix1 = ncap + nl4 * (ir - 1) + ip;
    } else {
        double tp = tt - (int)tt, tmp = nside * sqrt(3.0 * (1.0 - za));
        int jp = (int)(tp * tmp), jm = (int)((1.0 - tp) * tmp);
        int ir = jp + jm + 1, ip = (int)(tt * ir) + 1;
        if (ip > 4 * ir) ip = ip - 4 * ir;
        if (z <= 0.0) ipix1 = npix - 2 * ir * (ir + 1) + ip; else ipix1 = 2 * ir * (ir - 1) + ip;
    }
    return ipix1;
}

/* =====================================================================================
 * Peel-off to far observers: peelingoff_rect.f90
 * ===================================================================================== */
static int g_heal_nside = 0;      /* set per call site through obs_dir_s() */
static inline void obs_dir(const obs_t *o, const photon_t *p, photon_t *pobs, double *r2, int nxim, int nyim, double dxim, double dyim, int *ix, int *iy);
/* peel direction + pixel: TAN image (peelingoff_rect.f90:595-627) or HEALPix pixel of -k_obs (peelingoff_heal.f90:35-45) */
static inline void obs_dir_s(const orc_sim *s, int i, const photon_t *p, photon_t *pobs, double *r2, int *ix, int *iy) {
    if (s->nside > 0) {
        const obs_t *o = &s->obs[i];
        *pobs = *p;
        pobs->kx = o->x - p->x; pobs->ky = o->y - p->y; pobs->kz = o->z - p->z;
        *r2 = pobs->kx * pobs->kx + pobs->ky * pobs->ky + pobs->kz * pobs->kz;
        double r = sqrt(*r2);
        pobs->kx = pobs->kx / r; pobs->ky = pobs->ky / r; pobs->kz = pobs->kz / r;
        *ix = orc_vec2pix(s->nside, -pobs->kx, -pobs->ky, -pobs->kz); *iy = 1;
    } else obs_dir(&s->obs[i], p, pobs, r2, s->nxim, s->nyim, s->dxim, s->dyim, ix, iy);
}
/* tau along the peel ray: to the grid edge (far observers) or to the observer (inside: raytrace_to_dist, peelingoff_heal.f90:60) */
/* raytrace_to_edge_car_dust_Ha (raytrace_car.f90:510-582) and raytrace_to_edge_amr_dust_Ha (raytrace_amr.f90:353-390): the band-2 sight line, dust only */
static double raytrace_to_edge_dust_Ha(const orc_sim *s, const photon_t *p, double *nsteps) {
    double tau = 0.0;
    if (s->DGR <= 0.0) return tau;
    if (s->use_amr_grid) {
        const amr_t *a = &s->amr; double x = p->x, y = p->y, z = p->z; int il = p->icell;
        if (il <= 0) { il = amr_find_leaf(a, x, y, z); if (il <= 0) return tau; }
        for (;;) {
            int ic = a->icell_of_leaf[il-1], iface; double t_exit;
            amr_cell_exit(a, ic, x, y, z, p->kx, p->ky, p->kz, &t_exit, &iface);
            if (nsteps) *nsteps += 1.0;
            tau = tau + t_exit * (s->rhokapD[il-1] * s->R_Ha);
            if (tau >= TAU_HUGE) return tau;
            x = x + t_exit * p->kx; y = y + t_exit * p->ky; z = z + t_exit * p->kz;
            int il_new = amr_next_leaf(a, ic, iface, x, y, z);
            if (il_new <= 0) return tau;
            il = il_new;
        }
    }
    trav_t t; double d = 0.0;
    t.kx = p->kx; t.ky = p->ky; t.kz = p->kz; t.icell = p->icell; t.jcell = p->jcell; t.kcell = p->kcell;
    if (setup_traversal_car(s, p->x, p->y, p->z, &t)) return tau;
    for (;;) {
        const double rhokap = s->rhokapD[IDX(s,t.icell,t.jcell,t.kcell)] * s->R_Ha;
        if (nsteps) *nsteps += 1.0;
        const int idx = minloc3(t.tx, t.ty, t.tz);
        if (idx == 1) { tau = tau + (t.tx - d) * rhokap; d = t.tx; t.icell += t.istep; if (t.icell < 1 || t.icell > s->nx) break; t.tx += t.delx; }
        else if (idx == 2) { tau = tau + (t.ty - d) * rhokap; d = t.ty; t.jcell += t.jstep; if (t.jcell < 1 || t.jcell > s->ny) break; t.ty += t.dely; }
        else { tau = tau + (t.tz - d) * rhokap; d = t.tz; t.kcell += t.kstep; if (t.kcell < 1 || t.kcell > s->nz) break; t.tz += t.delz; }
        if (tau >= TAU_HUGE) break;
    }
    return tau;
}
static inline double peel_tau(const orc_sim *s, acc_t *A, const photon_t *po, double r2) {
    if (po->iband == 2) return raytrace_to_edge_dust_Ha(s, po, &A->n_cellsteps_peel);          /* peel_raytrace_to_edge: peelingoff_rect.f90:900-902 */
    if (s->use_clump_medium && s->cl.has_overlap) return raytrace_to_edge_clump_overlap(s, po, TAU_HUGE, &A->n_cellsteps_peel);      /* :895-896 */
    if (s->use_clump_medium) return raytrace_to_edge_clump(s, po, TAU_HUGE, &A->n_cellsteps_peel);      /* peel_raytrace_to_edge: peelingoff_rect.f90:872-906 */
    return raytrace_to_edge(s, po, s->raytrace_variant, s->nside > 0 ? sqrt(r2) : -1.0, &A->n_cellsteps_peel);
}
static inline void obs_dir(const obs_t *o, const photon_t *p, photon_t *pobs, double *r2, int nxim, int nyim, double dxim, double dyim, int *ix, int *iy) {
    *pobs = *p;
    pobs->kx = o->x - p->x; pobs->ky = o->y - p->y; pobs->kz = o->z - p->z;
    *r2 = pobs->kx * pobs->kx + pobs->ky * pobs->ky + pobs->kz * pobs->kz;
    double r = sqrt(*r2);
    pobs->kx = pobs->kx / r; pobs->ky = pobs->ky / r; pobs->kz = pobs->kz / r;
    const double *m = o->rm;
    double kx = m[0] * pobs->kx + m[1] * pobs->ky + m[2] * pobs->kz, ky = m[3] * pobs->kx + m[4] * pobs->ky + m[5] * pobs->kz, kz = m[6] * pobs->kx + m[7] * pobs->ky + m[8] * pobs->kz;
    *ix = (int)floor(atan2(-kx, kz) * RAD2DEG / dxim + nxim / 2.0) + 1;
    *iy = (int)floor(atan2(-ky, kz) * RAD2DEG / dyim + nyim / 2.0) + 1;
}
#define PIX2(s,ix,iy)      ((size_t)((ix)-1) + (size_t)(s)->nxim * (size_t)((iy)-1))
#define PIX3(s,ixf,ix,iy)  ((size_t)((ixf)-1) + (size_t)(s)->nxfreq * PIX2(s,ix,iy))

/* bulk line-of-sight velocity towards the observer: the owner clump's in clump mode (peelingoff_rect.f90:65-68,186-187,403-404,534-535,658-659) */
static inline double peel_u1(const orc_sim *s, const photon_t *p, size_t c, const photon_t *po) {
    if (s->use_clump_medium && p->icell_clump > 0) return ulos_clump(&s->cl, p->icell_clump, po->kx, po->ky, po->kz);
    return udot(s, c, po->kx, po->ky, po->kz);
}
/* peeling_direct_outside: peelingoff_rect.f90:24-129 */
static void peeling_direct(const orc_sim *s, acc_t *A, const photon_t *p) {
    for (int i = 0; i < s->nobs; i++) {
        photon_t po; double r2; int ix, iy;
        obs_dir_s(s, i, p, &po, &r2, &ix, &iy);
        size_t c = IDX(s,p->icell,p->jcell,p->kcell);
        double xfreq_ref, u1, u2;
        if (s->use_clump_medium && p->icell_clump > 0) { u1 = peel_u1(s, p, c, &po); xfreq_ref = p->xfreq + u1; }
        else if (!s->comoving_source) { u1 = udot(s, c, p->kx, p->ky, p->kz); xfreq_ref = p->xfreq + u1; u2 = udot(s, c, po.kx, po.ky, po.kz); po.xfreq = xfreq_ref - u2; }
        else { u1 = udot(s, c, po.kx, po.ky, po.kz); xfreq_ref = p->xfreq + u1; }
        xfreq_ref = xfreq_ref * (s->Dfreq[c] / s->Dfreq_ref);
        int ixf = (int)floor((xfreq_ref - s->xfreq_min) / s->dxfreq) + 1;
        if (ix >= 1 && ix <= s->nxim && iy >= 1 && iy <= s->nyim) {
            double tau = peel_tau(s, A, &po, r2);
            double wgt0 = 1.0 / (FOURPI * r2) * p->wgt, wgt = exp(-tau) * wgt0;
            obs_t *o = &A->obs[i];
            if (s->save_peeloff_2D) { o->direc_2D[PIX2(s,ix,iy)] += wgt; if (s->use_stokes) o->I_2D[PIX2(s,ix,iy)] += wgt; if (s->save_direc0) o->direc0_2D[PIX2(s,ix,iy)] += wgt0; }
            if (s->save_peeloff_3D && ixf >= 1 && ixf <= s->nxfreq) { o->direc[PIX3(s,ixf,ix,iy)] += wgt; if (s->use_stokes) o->I[PIX3(s,ixf,ix,iy)] += wgt; if (s->save_direc0) o->direc0[PIX3(s,ixf,ix,iy)] += wgt0; }
        }
    }
}
/* peeling_resonance_nostokes_outside: :576-690 */
static void peeling_resonance_nostokes(const orc_sim *s, acc_t *A, const photon_t *p, double xfreq_atom, const double va[3]) {
    for (int i = 0; i < s->nobs; i++) {
        photon_t po; double r2; int ix, iy;
        obs_dir_s(s, i, p, &po, &r2, &ix, &iy);
        if (!(ix >= 1 && ix <= s->nxim && iy >= 1 && iy <= s->nyim)) continue;
        double cost = p->kx * po.kx + p->ky * po.ky + p->kz * po.kz, cost2 = cost * cost, sint = sqrt(1.0 - cost2);
        double rho1 = sqrt(1.0 - p->kz * p->kz) * sint, cosp, sinp, xfreq;
        if (rho1 == 0.0) { cosp = 1.0; sinp = 0.0; }
        else { double rho = 1.0 / rho1; cosp = rho * (cost * p->kz - po.kz); sinp = rho * (p->kx * po.ky - po.kx * p->ky); }
        xfreq = xfreq_atom + (va[0] * cosp + va[1] * sinp) * sint + va[2] * cost;
        size_t c = IDX(s,p->icell,p->jcell,p->kcell);
        if (s->recoil) { double g_recoil = ((s->line_type == 7 && p->species == 2) ? s->g_recoil0_D : s->g_recoil0) / s->Dfreq[c]; xfreq = xfreq - g_recoil * (1.0 - cost); }
        double u1 = peel_u1(s, p, c, &po), xfreq_ref = xfreq + u1;
        xfreq_ref = xfreq_ref * (s->Dfreq[c] / s->Dfreq_ref);
        int ixf = (int)floor((xfreq_ref - s->xfreq_min) / s->dxfreq) + 1;
        po.xfreq = xfreq;
        double tau = peel_tau(s, A, &po, r2);
        double peel = 0.75 * p->E1 * (cost2 + 1.0) + p->E2;
        double wgt = peel / (FOURPI * r2) * exp(-tau) * p->wgt;
        obs_t *o = &A->obs[i];
        if (s->save_peeloff_2D) o->scatt_2D[PIX2(s,ix,iy)] += wgt;
        if (s->save_peeloff_3D && ixf >= 1 && ixf <= s->nxfreq) o->scatt[PIX3(s,ixf,ix,iy)] += wgt;
    }
}
static inline void stokes_deposit(const orc_sim *s, obs_t *o, int ix, int iy, int ixf, double wgt, double Id, double Qd, double Ud, double Vd) {
    if (s->save_peeloff_2D) { size_t q = PIX2(s,ix,iy);
        o->scatt_2D[q] += wgt * Id; o->I_2D[q] += wgt * Id; o->Q_2D[q] += wgt * Qd; o->U_2D[q] += wgt * Ud; o->V_2D[q] += wgt * Vd; }
    if (s->save_peeloff_3D && ixf >= 1 && ixf <= s->nxfreq) { size_t q = PIX3(s,ixf,ix,iy);
        o->scatt[q] += wgt * Id; o->I[q] += wgt * Id; o->Q[q] += wgt * Qd; o->U[q] += wgt * Ud; o->V[q] += wgt * Vd; }
}
/* common geometry of the two Stokes peels (:356-385 / :204-224): azimuth about the (m,n,k) triad and n_obs */
static inline void stokes_peel_geom(const photon_t *p, photon_t *po, double *cost, double *sint, double *cosp, double *sinp, double *cos2p, double *sin2p) {
    *cost = p->kx * po->kx + p->ky * po->ky + p->kz * po->kz; *sint = sqrt(1.0 - (*cost) * (*cost));
    if (*sint == 0.0) { *cosp = 1.0; *sinp = 0.0; *cos2p = 1.0; *sin2p = 0.0; }
    else { *cosp = (po->kx * p->mx + po->ky * p->my + po->kz * p->mz) / *sint; *sinp = (po->kx * p->nx + po->ky * p->ny + po->kz * p->nz) / *sint;
           *cos2p = 2.0 * (*cosp) * (*cosp) - 1.0; *sin2p = 2.0 * (*cosp) * (*sinp); }
    po->nx = -(*sinp) * p->mx + (*cosp) * p->nx; po->ny = -(*sinp) * p->my + (*cosp) * p->ny; po->nz = -(*sinp) * p->mz + (*cosp) * p->nz;
}
/* peeling_resonance_stokes_outside: :303-482 */
static void peeling_resonance_stokes(const orc_sim *s, acc_t *A, const photon_t *p, double xfreq_atom, const double va[3]) {
    for (int i = 0; i < s->nobs; i++) {
        photon_t po; double r2; int ix, iy;
        obs_dir_s(s, i, p, &po, &r2, &ix, &iy);
        if (!(ix >= 1 && ix <= s->nxim && iy >= 1 && iy <= s->nyim)) continue;
        double cost, sint, cosp, sinp, cos2p, sin2p;
        stokes_peel_geom(p, &po, &cost, &sint, &cosp, &sinp, &cos2p, &sin2p);
        double cost2 = cost * cost, S22 = 0.75 * p->E1 * (cost2 + 1.0), S11 = S22 + p->E2, S12 = 0.75 * p->E1 * (cost2 - 1.0), S33 = 1.5 * p->E1 * cost, S44 = 1.5 * p->E3 * cost;
        double xfreq = xfreq_atom + (va[0] * cosp + va[1] * sinp) * sint + va[2] * cost;
        size_t c = IDX(s,p->icell,p->jcell,p->kcell);
        if (s->recoil) { double g_recoil = ((s->line_type == 7 && p->species == 2) ? s->g_recoil0_D : s->g_recoil0) / s->Dfreq[c]; xfreq = xfreq - g_recoil * (1.0 - cost); }
        double u1 = peel_u1(s, p, c, &po), xfreq_ref = (xfreq + u1) * (s->Dfreq[c] / s->Dfreq_ref);
        int ixf = (int)floor((xfreq_ref - s->xfreq_min) / s->dxfreq) + 1;
        double Q0 = cos2p * p->Q + sin2p * p->U, U0 = -sin2p * p->Q + cos2p * p->U;
        double Iobs = (S11 + S12 * Q0) / FOURPI, Qobs = (S12 + S22 * Q0) / FOURPI, Uobs = (S33 * U0) / FOURPI, Vobs = (S44 * p->V) / FOURPI;
        const double *m = s->obs[i].rm;
        double cosg = -(m[0] * po.nx + m[1] * po.ny + m[2] * po.nz), sing = m[3] * po.nx + m[4] * po.ny + m[5] * po.nz;
        double cos2g = 2.0 * cosg * cosg - 1.0, sin2g = 2.0 * cosg * sing;
        double Idet = Iobs, Qdet = cos2g * Qobs + sin2g * Uobs, Udet = -sin2g * Qobs + cos2g * Uobs, Vdet = Vobs;
        po.xfreq = xfreq;
        double tau = peel_tau(s, A, &po, r2);
        double wgt = 1.0 / r2 * exp(-tau) * p->wgt;
        stokes_deposit(s, &A->obs[i], ix, iy, ixf, wgt, Idet, Qdet, Udet, Vdet);
    }
}
/* peeling_dust_nostokes_outside: :484-574 */
static void peeling_dust_nostokes(const orc_sim *s, acc_t *A, const photon_t *p) {
    for (int i = 0; i < s->nobs; i++) {
        photon_t po; double r2; int ix, iy;
        obs_dir_s(s, i, p, &po, &r2, &ix, &iy);
        if (!(ix >= 1 && ix <= s->nxim && iy >= 1 && iy <= s->nyim)) continue;
        size_t c = IDX(s,p->icell,p->jcell,p->kcell);
        double u1 = peel_u1(s, p, c, &po), xfreq_ref = (p->xfreq + u1) * (s->Dfreq[c] / s->Dfreq_ref);
        int ixf = (int)floor((xfreq_ref - s->xfreq_min) / s->dxfreq) + 1;
        double tau = peel_tau(s, A, &po, r2);
        double cosa = p->kx * po.kx + p->ky * po.ky + p->kz * po.kz;
        double peel = (1.0 - s->hgg * s->hgg) / pow((1.0 + s->hgg * s->hgg) - 2.0 * s->hgg * cosa, 1.5) / FOURPI;
        double wgt = peel / r2 * exp(-tau) * p->wgt;
        obs_t *o = &A->obs[i];
        if (s->save_peeloff_2D) o->scatt_2D[PIX2(s,ix,iy)] += wgt;
        if (s->save_peeloff_3D && ixf >= 1 && ixf <= s->nxfreq) { o->scatt[PIX3(s,ixf,ix,iy)] += wgt; if (s->save_dust_scattered) o->scatt_dust[PIX3(s,ixf,ix,iy)] += wgt; }   /* :567-570 */
    }
}
/* peeling_dust_stokes_outside: :131-299 */
static void peeling_dust_stokes(const orc_sim *s, acc_t *A, const photon_t *p) {
    for (int i = 0; i < s->nobs; i++) {
        photon_t po; double r2; int ix, iy;
        obs_dir_s(s, i, p, &po, &r2, &ix, &iy);
        size_t c = IDX(s,p->icell,p->jcell,p->kcell);
        double u1 = peel_u1(s, p, c, &po), xfreq_ref = (p->xfreq + u1) * (s->Dfreq[c] / s->Dfreq_ref);
        int ixf = (int)floor((xfreq_ref - s->xfreq_min) / s->dxfreq) + 1;
        if (!(ix >= 1 && ix <= s->nxim && iy >= 1 && iy <= s->nyim)) continue;
        double cost, sint, cosp, sinp, cos2p, sin2p;
        stokes_peel_geom(p, &po, &cost, &sint, &cosp, &sinp, &cos2p, &sin2p);
        double S11 = interp_eq(s->nPDF, s->coss, s->S11, cost), S12 = interp_eq(s->nPDF, s->coss, s->S12, cost);
        double S33 = interp_eq(s->nPDF, s->coss, s->S33, cost), S34 = interp_eq(s->nPDF, s->coss, s->S34, cost);
        double Q0 = cos2p * p->Q + sin2p * p->U, U0 = -sin2p * p->Q + cos2p * p->U;
        double Iobs = (S11 * p->I + S12 * Q0) / TWOPI, Qobs = (S12 * p->I + S11 * Q0) / TWOPI;
        double Uobs = (S33 * U0 + S34 * p->V) / TWOPI, Vobs = (-S34 * U0 + S33 * p->V) / TWOPI;
        const double *m = s->obs[i].rm;
        double cosg = -(m[0] * po.nx + m[1] * po.ny + m[2] * po.nz), sing = m[3] * po.nx + m[4] * po.ny + m[5] * po.nz;
        double cos2g = 2.0 * cosg * cosg - 1.0, sin2g = 2.0 * cosg * sing;
        double Idet = Iobs, Qdet = cos2g * Qobs + sin2g * Uobs, Udet = -sin2g * Qobs + cos2g * Uobs, Vdet = Vobs;
        double tau = peel_tau(s, A, &po, r2);
        double wgt = 1.0 / r2 * exp(-tau) * p->wgt;
        stokes_deposit(s, &A->obs[i], ix, iy, ixf, wgt, Idet, Qdet, Udet, Vdet);
        if (s->save_dust_scattered && s->save_peeloff_3D && ixf >= 1 && ixf <= s->nxfreq && ix >= 1 && ix <= s->nxim && iy >= 1 && iy <= s->nyim) A->obs[i].scatt_dust[PIX3(s,ixf,ix,iy)] += wgt * Idet;   /* :284-287 */
    }
}

/* =====================================================================================
 * Scattering: scattering_car.f90
 * ===================================================================================== */
/* do_resonance1: line_mod.f90:108-139 */
/* do_resonance1 / do_resonance2 (line_mod.f90:108-192) and their clump forms (line_clump_mod.f90:30-110: local Doppler units and back) */
static void do_resonance1(const orc_sim *s, rng_t *g, photon_t *p, double *uz, double *xfreq_atom, double *cost, double *sint,
                          double *S11, double *S22, double *S12, double *S33, double *S44) {
    double scale = 1.0, scale_inv = 1.0, va, Df;
    if (s->use_clump_medium) { const clumps_t *c = &s->cl; int64_t icl = p->icell_clump;
        scale = c->Dfreq_ref / c->Dfreq[icl-1]; scale_inv = 1.0 / scale; va = c->voigt_a[icl-1]; Df = c->Dfreq[icl-1]; }
    else { size_t c = IDX(s,p->icell,p->jcell,p->kcell); va = s->voigt_a[c]; Df = s->Dfreq[c]; }
    const double xfreq_loc = s->use_clump_medium ? p->xfreq * scale : p->xfreq;
    double uz_loc, xatom_loc;
    if (s->line_type == 2) {                                /* fine structure: pick the 2P1/2 (H) or 2P3/2 (K) level by its share of the profile */
        double DnuHK = s->DnuHK_Hz / Df;
        double pH = orc_voigt(xfreq_loc + DnuHK, va) * (1.0 / 3.0), pK = orc_voigt(xfreq_loc, va) * (2.0 / 3.0);
        pH = pH / (pH + pK);
        const double rl_ = rnd(g); MRG(rl_, pH, 63);
        if (rl_ < pH) uz_loc = rand_resonance_vz(g, xfreq_loc + DnuHK, va); else uz_loc = rand_resonance_vz(g, xfreq_loc, va);
        xatom_loc = xfreq_loc - uz_loc;
        double qH = xatom_loc + DnuHK, qK = xatom_loc;
        p->E1 = (2.0 * qK * qH + qH * qH) / (qK * qK + 2.0 * qH * qH);             /* E1 of the atom-frame frequency */
        p->E2 = 1.0 - p->E1; p->E3 = (p->E1 + 2.0) / 3.0;
    } else if (s->line_type >= 4 && s->line_type <= 6) {   /* do_resonance4 :249-291, 5 :347-406, 6 :408-474 and their clump forms (line_clump_mod.f90:105-260) */
        int iup = 1, idown = 1;
        if (s->line_type == 4) uz_loc = rand_resonance_vz(g, xfreq_loc, va);
        else {
            double Dx2 = s->ml.delE_Hz[1] / Df, va2 = va * s->ml.damping_up[1] / s->ml.damping_up[0];
            double p1 = orc_voigt(xfreq_loc, va) * s->ml.f12[0], p2 = orc_voigt(xfreq_loc + Dx2, va2) * s->ml.f12[1];
            if (s->line_type == 5) {
                p1 = p1 / (p2 + p1);
                const double rl_ = rnd(g); MRG(rl_, p1, 63);
                if (rl_ < p1) { uz_loc = rand_resonance_vz(g, xfreq_loc, va); iup = 1; } else { uz_loc = rand_resonance_vz(g, xfreq_loc + Dx2, va2); iup = 2; }
            } else {
                double Dx3 = s->ml.delE_Hz[2] / Df, va3 = va * s->ml.damping_up[2] / s->ml.damping_up[0];
                double p3 = orc_voigt(xfreq_loc + Dx3, va3) * s->ml.f12[2], ptot = p1 + p2 + p3, pcum1 = p1 / ptot, pcum2 = (p1 + p2) / ptot, xi = rnd(g); MRG(xi, pcum1, 63); MRG(xi, pcum2, 63);
                if (xi < pcum1) { uz_loc = rand_resonance_vz(g, xfreq_loc, va); iup = 1; }
                else if (xi < pcum2) { uz_loc = rand_resonance_vz(g, xfreq_loc + Dx2, va2); iup = 2; }
                else { uz_loc = rand_resonance_vz(g, xfreq_loc + Dx3, va3); iup = 3; }
            }
        }
        xatom_loc = xfreq_loc - uz_loc;
        if (s->line_type == 4 || (s->line_type == 5 && s->ml.ndown[iup-1] > 1)) {           /* rand_alias_choise: random_mt.f90:2122-2135 */
            int n = s->ml.ndown[iup-1], k = (int)floor(n * rnd(g)) + 1;
            { const double rl_ = rnd(g); MRG(rl_, s->ml.P_down[iup-1][k-1], 63); idown = (rl_ < s->ml.P_down[iup-1][k-1]) ? k : s->ml.A_down[iup-1][k-1]; }
            if (s->line_type == 5 || idown != 1) xatom_loc = xatom_loc - s->ml.Elow_Hz[iup-1][idown-1] / Df;
        }
        p->E1 = s->ml.E1[iup-1][idown-1]; p->E2 = s->ml.E2[iup-1][idown-1]; p->E3 = s->ml.E3[iup-1][idown-1];
        if (s->line_type == 6 && s->HeI_coherent) {         /* compute_HeI_E_coherent :8-34 (clump form line_clump_mod.f90:236-238): offsets from P2, P1, P0 */
            double D2 = xatom_loc, D1 = xatom_loc + s->ml.delE_Hz[1] / Df, D0 = xatom_loc + s->ml.delE_Hz[2] / Df;
            double D2D0 = D2 * D0, D2D1 = D2 * D1, D0D1 = D0 * D1, pp = D2 * D0 * D1;
            double den = 4.0 * (D2D1 * D2D1 + 3.0 * D2D0 * D2D0 + 5.0 * D0D1 * D0D1);
            p->E1 = (3.0 * D2D0 * D2D0 + 7.0 * D0D1 * D0D1 + 8.0 * pp * D1 + 18.0 * pp * D0) / den;
            p->E3 = (3.0 * D2D0 * D2D0 + 15.0 * D0D1 * D0D1 + 8.0 * D2 * pp + 10.0 * pp * D0) / den;
            p->E2 = 1.0 - p->E1;
        }
    } else if (s->line_type == 7) {                         /* do_resonance_HD :476-549 (clump form line_clump_mod.f90:262-300): H or D atom */
        double dx_HD = s->delta_nu_HD_Hz / Df, xfreq_D = (xfreq_loc - dx_HD) * s->ratio_Dfreq_HD;
        double pH = orc_voigt(xfreq_loc, va), pD = s->nD_over_nH * s->ratio_Dfreq_HD * orc_voigt(xfreq_D, va * s->ratio_voigta_HD);
        const double rl_ = rnd(g); MRG(rl_, pH / (pH + pD), 63);
        if (rl_ < pH / (pH + pD)) { p->species = 1; uz_loc = rand_resonance_vz(g, xfreq_loc, va); }
        else { p->species = 2; uz_loc = rand_resonance_vz(g, xfreq_D, va * s->ratio_voigta_HD) / s->ratio_Dfreq_HD; }
        xatom_loc = xfreq_loc - uz_loc;
        p->E1 = s->E1; p->E2 = s->E2; p->E3 = s->E3;
    } else if (s->line_type == 8) {                         /* do_resonance8 :293-345: the upward step of do_resonance1, then the downward channel of the 3p level */
        uz_loc = rand_resonance_vz(g, xfreq_loc, va);
        xatom_loc = xfreq_loc - uz_loc;
        const double rc_ = rnd(g); MRG(rc_, s->P_conv, 63);
        if (rc_ < s->P_conv) { p->iband = 2; p->E1 = s->E_conv[0]; p->E2 = s->E_conv[1]; p->E3 = s->E_conv[2]; }      /* 3p -> 2s: H-alpha */
        else { p->E1 = s->E1; p->E2 = s->E2; p->E3 = s->E3; }                                                         /* 3p -> 1s: Ly-beta again */
    } else {
        uz_loc = rand_resonance_vz(g, xfreq_loc, va);
        xatom_loc = xfreq_loc - uz_loc;
        p->E1 = s->E1; p->E2 = s->E2; p->E3 = s->E3;
    }
    if (s->use_clump_medium) { *xfreq_atom = xatom_loc * scale_inv; *uz = uz_loc * scale_inv; }
    else { *uz = uz_loc; *xfreq_atom = xatom_loc; }
    *cost = rand_resonance(g, p->E1);
    double cost2 = (*cost) * (*cost);
    *sint = sqrt(1.0 - cost2);
    if (S44) { *S22 = 0.75 * p->E1 * (cost2 + 1.0); *S11 = *S22 + p->E2; *S12 = 0.75 * p->E1 * (cost2 - 1.0); *S33 = 1.5 * p->E1 * (*cost);
               *S44 = (s->line_type == 2) ? (p->E1 + 2.0) / 2.0 * (*cost) : 1.5 * p->E3 * (*cost); }
}
static inline void rotate_k(photon_t *p, double cost, double sint, double cosp, double sinp) {
    if (fabs(p->kz) >= 0.99999999999) { p->kx = sint * cosp; p->ky = sint * sinp; p->kz = cost; }
    else { double kx1 = p->kx, ky1 = p->ky, kz1 = p->kz, kr = sqrt(kx1 * kx1 + ky1 * ky1);
        p->kx = cost * kx1 + sint * (kz1 * kx1 * cosp - ky1 * sinp) / kr;
        p->ky = cost * ky1 + sint * (kz1 * ky1 * cosp + kx1 * sinp) / kr;
        p->kz = cost * kz1 - sint * cosp * kr; }
}
/* triad rotation, fixed order ("Do not change the calculation order"): :470-484 and :315-328 */
static inline void rotate_triad(photon_t *p, double cost, double sint, double cosp, double sinp) {
    double px = cosp * p->mx + sinp * p->nx, py = cosp * p->my + sinp * p->ny, pz = cosp * p->mz + sinp * p->nz;
    p->nx = cosp * p->nx - sinp * p->mx; p->ny = cosp * p->ny - sinp * p->my; p->nz = cosp * p->nz - sinp * p->mz;
    p->mx = cost * px - sint * p->kx; p->my = cost * py - sint * p->ky; p->mz = cost * pz - sint * p->kz;
    p->kx = sint * px + cost * p->kx; p->ky = sint * py + cost * p->ky; p->kz = sint * pz + cost * p->kz;
}
/* clump thermal-width ratio (scattering_car.f90:384-388, 715-719) and recoil constant (:428-433, 769-774) of the owner clump */
static inline double clump_vth_ratio(const orc_sim *s, const photon_t *p) { return (s->use_clump_medium && p->icell_clump > 0) ? s->cl.Dfreq[p->icell_clump-1] / s->cl.Dfreq_ref : 1.0; }
static inline double recoil_const(const orc_sim *s, const photon_t *p) {
    const double g0 = (s->line_type == 7 && p->species == 2) ? s->g_recoil0_D : s->g_recoil0;      /* a deuterium atom recoils less (scattering_car.f90:427-437) */
    if (s->use_clump_medium && p->icell_clump > 0) return g0 / s->cl.Dfreq[p->icell_clump-1];
    return g0 / s->Dfreq[IDX(s,p->icell,p->jcell,p->kcell)];
}
/* D scatter: the perpendicular velocity comes in D Doppler units -> the photon's H units (scattering_car.f90:402-409, :417-424) */
/* perpendicular atom velocity: core-skip branch :733-737, plain :749-752 */
static inline void perp_velocity_rayleigh(const orc_sim *s, rng_t *g, const photon_t *p, double *ux, double *uy) {
    double xc = 0.0, xc2 = 0.0, phi2, uxy;
    if (s->core_skip) car_xcrit_local(s, p->icell, p->jcell, p->kcell, p->x, p->y, p->z, &xc, &xc2);
    if (s->core_skip) MRG(fabs(p->xfreq), xc, 64);
    if (s->core_skip && fabs(p->xfreq) < xc) { phi2 = TWOPI * rnd(g); uxy = sqrt(xc2 - log(rnd(g))); }
    else                                     { phi2 = TWOPI * rnd(g); uxy = sqrt(-log(rnd(g))); }
    *ux = uxy * cos(phi2); *uy = uxy * sin(phi2);
    if (s->line_type == 7 && p->species == 2) { *ux = *ux / s->ratio_Dfreq_HD; *uy = *uy / s->ratio_Dfreq_HD; }
    double vr = clump_vth_ratio(s, p); *ux = *ux * vr; *uy = *uy * vr;
}
/* peeling_conversion_Ha_outside: peelingoff_rect.f90:692-792 (the newborn H-alpha packet of a conversion: emitted at line centre in the atom frame,
 * dipole phase function of the conversion channel, dust-only sight line, no recoil; into observer%peel_Ha) */
static void peeling_conversion_Ha(const orc_sim *s, acc_t *A, const photon_t *p, const double va[3]) {
    for (int i = 0; i < s->nobs; i++) {
        photon_t po; double r2; int ix, iy;
        obs_dir_s(s, i, p, &po, &r2, &ix, &iy);
        if (!(ix >= 1 && ix <= s->nxim && iy >= 1 && iy <= s->nyim)) continue;
        double cost = p->kx * po.kx + p->ky * po.ky + p->kz * po.kz, cost2 = cost * cost, sint = sqrt(1.0 - cost2), rho1 = sqrt(1.0 - p->kz * p->kz) * sint, cosp, sinp;
        if (rho1 == 0.0) { cosp = 1.0; sinp = 0.0; } else { double rho = 1.0 / rho1; cosp = rho * (cost * p->kz - po.kz); sinp = rho * (p->kx * po.ky - po.kx * p->ky); }
        double xfreq = (va[0] * cosp + va[1] * sinp) * sint + va[2] * cost;
        size_t c = IDX(s,p->icell,p->jcell,p->kcell);
        double u1 = udot(s, c, po.kx, po.ky, po.kz), xfreq_ref = (xfreq + u1) * (s->Dfreq[c] / s->Dfreq_ref);
        int ixf = (int)floor((xfreq_ref - s->xfreq_min_Ha) / s->dxfreq_Ha) + 1;
        po.xfreq = xfreq_ref;                                   /* (po.iband = 2 came with the copy of the packet) */
        double tau = peel_tau(s, A, &po, r2);
        double peel = 0.75 * p->E1 * (cost2 + 1.0) + p->E2, wgt = peel / (FOURPI * r2) * exp(-tau) * p->wgt;
        obs_t *o = &A->obs[i];
        if (o->peel_Ha && ixf >= 1 && ixf <= s->nxfreq_Ha) o->peel_Ha[(size_t)(ixf - 1) + (size_t)s->nxfreq_Ha * PIX2(s,ix,iy)] += wgt;
    }
}
/* peeling_dust_Ha_outside: :794-870 (band-2 dust scattering: HG phase function of par%hgg_Ha, the packet's lab-frame frequency as it is) */
static void peeling_dust_Ha(const orc_sim *s, acc_t *A, const photon_t *p) {
    for (int i = 0; i < s->nobs; i++) {
        photon_t po; double r2; int ix, iy;
        obs_dir_s(s, i, p, &po, &r2, &ix, &iy);
        if (!(ix >= 1 && ix <= s->nxim && iy >= 1 && iy <= s->nyim)) continue;
        int ixf = (int)floor((p->xfreq - s->xfreq_min_Ha) / s->dxfreq_Ha) + 1;
        double tau = peel_tau(s, A, &po, r2);
        double cosa = p->kx * po.kx + p->ky * po.ky + p->kz * po.kz, gg = s->hgg_Ha;
        double peel = (1.0 - gg * gg) / pow((1.0 + gg * gg) - 2.0 * gg * cosa, 1.5) / FOURPI, wgt = peel / r2 * exp(-tau) * p->wgt;
        obs_t *o = &A->obs[i];
        if (o->peel_Ha && ixf >= 1 && ixf <= s->nxfreq_Ha) o->peel_Ha[(size_t)(ixf - 1) + (size_t)s->nxfreq_Ha * PIX2(s,ix,iy)] += wgt;
    }
}
/* scatter_dust_Ha_nostokes: scattering_car.f90:586-658 (scatter_dust_Ha_amr: scattering_amr.f90:147-209): a band-2 packet only meets dust */
static void scatter_dust_Ha(const orc_sim *s, acc_t *A, rng_t *g, photon_t *p) {
    p->nscatt_dust = p->nscatt_dust + p->wgt;
    if (!s->use_reduced_wgt) {
        const double ra_ = rnd(g); MRG(ra_, s->albedo_Ha, 61);
        if (ra_ > s->albedo_Ha) {
            if (s->save_Jabs) { int ix = (int)floor((p->xfreq - s->xfreq_min_Ha) / s->dxfreq_Ha) + 1; if (ix >= 1 && ix <= s->nxfreq_Ha) A->Jabs_Ha[ix-1] += p->wgt; }
            p->inside = 0; A->W[4] += p->wgt;
            if (s->save_all_photons) { p->xfreq_ref = p->xfreq; p->wgt = 0.0; }
            return; }
    } else {
        if (s->save_Jabs) { int ix = (int)floor((p->xfreq - s->xfreq_min_Ha) / s->dxfreq_Ha) + 1; if (ix >= 1 && ix <= s->nxfreq_Ha) A->Jabs_Ha[ix-1] += p->wgt * (1.0 - s->albedo_Ha); }
        A->W[4] += p->wgt * (1.0 - s->albedo_Ha);
        p->wgt = p->wgt * s->albedo_Ha;
    }
    if (s->save_peeloff) peeling_dust_Ha(s, A, p);
    double cost = rand_hg(g, s->hgg_Ha), sint = sqrt(1.0 - cost * cost);
    double phi = TWOPI * rnd(g), cosp = cos(phi), sinp = sin(phi);
    rotate_k(p, cost, sint, cosp, sinp);
}
/* scatter_h2: scattering_car.f90:122-199 (an H2 pumping event: fluorescence destroys the packet, or the molecule scatters it back near Ly-alpha) */
static void scatter_h2(const orc_sim *s, acc_t *A, rng_t *g, photon_t *p) {
    const size_t c = IDX(s,p->icell,p->jcell,p->kcell);
    const double Dfreq_cell = s->Dfreq[c];
    const int il = h2_select_line_D(s, g, p->xfreq, Dfreq_cell);
    A->W_H2[2 + il - 1] += p->wgt;
    { const double rs_ = rnd(g); MRG(rs_, s->h2_p_scat[il-1], 68);
      if (rs_ > s->h2_p_scat[il-1]) { A->W_H2[0] += p->wgt; p->inside = 0; if (s->save_all_photons) p->wgt = 0.0; return; } }
    A->W_H2[1] += p->wgt; p->nscatt_gas = p->nscatt_gas + p->wgt;
    const double ratio = Dfreq_cell / s->h2_Dfreq_Hz, dx = s->h2_dnu_Hz[il-1] / Dfreq_cell, x_h2 = (p->xfreq - dx) * ratio;
    const double uz = rand_resonance_vz(g, x_h2, s->h2_a_damp[il-1]), xatom = x_h2 - uz;
    const double cost = 2.0 * rnd(g) - 1.0, sint = sqrt(1.0 - cost * cost), phi = TWOPI * rnd(g), cosp = cos(phi), sinp = sin(phi);
    const double phi2 = TWOPI * rnd(g), uxy = sqrt(-log(rnd(g))), ux = uxy * cos(phi2), uy = uxy * sin(phi2);
    const double x_h2_new = xatom + uz * cost + (ux * cosp + uy * sinp) * sint;
    p->xfreq = x_h2_new / ratio + dx;
    rotate_k(p, cost, sint, cosp, sinp);
}
/* scatter_resonance_nostokes: :660-827 */
static void scatter_resonance_nostokes(const orc_sim *s, acc_t *A, rng_t *g, photon_t *p) {
    double uz, xfreq_atom, cost, sint, ux, uy;
    p->nscatt_gas = p->nscatt_gas + p->wgt;
    add_to_Pa(s, A, p);
    do_resonance1(s, g, p, &uz, &xfreq_atom, &cost, &sint, 0, 0, 0, 0, 0);
    const int converted = (s->line_type == 8 && p->iband == 2);        /* :697-705 */
    if (converted) { A->W[0] += p->wgt; add_to_Pconv(s, A, p); }
    double phi = TWOPI * rnd(g), cosp = cos(phi), sinp = sin(phi);
    perp_velocity_rayleigh(s, g, p, &ux, &uy);
    p->xfreq = xfreq_atom + uz * cost + (ux * cosp + uy * sinp) * sint;
    if (s->recoil && !converted) { double g_recoil = recoil_const(s, p); p->xfreq = p->xfreq - g_recoil * (1.0 - cost); }
    if (s->save_peeloff) { double va[3] = {ux, uy, uz}; if (converted) peeling_conversion_Ha(s, A, p, va); else peeling_resonance_nostokes(s, A, p, xfreq_atom, va); }
    rotate_k(p, cost, sint, cosp, sinp);
    if (converted) {                                                   /* :811-827: the H-alpha packet leaves at line centre in the atom frame; lab frame, reference units */
        const size_t c = IDX(s,p->icell,p->jcell,p->kcell);
        const double u1_fluid = udot(s, c, p->kx, p->ky, p->kz);
        p->xfreq = (p->xfreq - xfreq_atom + u1_fluid) * (s->Dfreq[c] / s->Dfreq_ref);
    }
}
/* azimuth by rejection on 1 + S12/S11 (Q cos2phi + U sin2phi): :364-371 and :281-288 */
static inline double phi_rejection(rng_t *g, const photon_t *p, double S12overS11) {
    for (;;) {
        double phi = TWOPI * rnd(g), phi1 = 2.0 * phi;
        double Prand = (1.0 + fabs(S12overS11) * sqrt(p->Q * p->Q + p->U * p->U)) * rnd(g);
        double Pcomp = 1.0 + S12overS11 * (p->Q * cos(phi1) + p->U * sin(phi1));
        MRG(Prand, Pcomp, 62);
        if (Prand <= Pcomp) return phi;
    }
}
/* scatter_resonance_stokes: :331-486 */
static void scatter_resonance_stokes(const orc_sim *s, acc_t *A, rng_t *g, photon_t *p) {
    double uz, xfreq_atom, cost, sint, S11, S22, S12, S33, S44, ux, uy;
    p->nscatt_gas = p->nscatt_gas + p->wgt;
    add_to_Pa(s, A, p);
    do_resonance1(s, g, p, &uz, &xfreq_atom, &cost, &sint, &S11, &S22, &S12, &S33, &S44);
    double phi = phi_rejection(g, p, S12 / S11), cosp = cos(phi), sinp = sin(phi);
    double xc = 0.0, xc2 = 0.0;
    if (s->core_skip) car_xcrit_local(s, p->icell, p->jcell, p->kcell, p->x, p->y, p->z, &xc, &xc2);
    if (s->core_skip) MRG(fabs(p->xfreq), xc, 64);
    if (s->core_skip && fabs(p->xfreq) < xc) { double phi2 = TWOPI * rnd(g), uxy = sqrt(xc2 - log(rnd(g))); ux = uxy * cos(phi2); uy = uxy * sin(phi2); }
    else {                                                                            /* :413-414 two gaussians / sqrt2 */
        if (g->kind == 0) { ux = rand_gauss(g) * (1.0 / sqrt(2.0)); uy = rand_gauss(g) * (1.0 / sqrt(2.0)); }
        else { double g2, g1 = gauss_pair(g, &g2); ux = g1 * (1.0 / sqrt(2.0)); uy = g2 * (1.0 / sqrt(2.0)); }
    }
    if (s->line_type == 7 && p->species == 2) { ux = ux / s->ratio_Dfreq_HD; uy = uy / s->ratio_Dfreq_HD; }
    { double vr = clump_vth_ratio(s, p); ux = ux * vr; uy = uy * vr; }
    p->xfreq = xfreq_atom + uz * cost + (ux * cosp + uy * sinp) * sint;
    if (s->recoil) { double g_recoil = recoil_const(s, p); p->xfreq = p->xfreq - g_recoil * (1.0 - cost); }
    if (s->save_peeloff) { double va[3] = {ux, uy, uz}; peeling_resonance_stokes(s, A, p, xfreq_atom, va); }
    double cos2p = 2.0 * cosp * cosp - 1.0, sin2p = 2.0 * sinp * cosp;
    double Q0 = cos2p * p->Q + sin2p * p->U, U0 = -sin2p * p->Q + cos2p * p->U;
    double I1 = S11 + S12 * Q0, Q1 = S12 + S22 * Q0, U1 = S33 * U0, V1 = S44 * p->V;
    p->I = 1.0; p->Q = Q1 / I1; p->U = U1 / I1; p->V = V1 / I1;
    rotate_triad(p, cost, sint, cosp, sinp);
}
/* absorption / reduced weight shared by both dust routines: :504-555 and :217-264. Returns 1 if absorbed */
static int dust_absorb(const orc_sim *s, acc_t *A, rng_t *g, photon_t *p) {
    p->nscatt_dust = p->nscatt_dust + p->wgt;
    size_t c = IDX(s,p->icell,p->jcell,p->kcell);
    if (!s->use_reduced_wgt) {
        const double ra_ = rnd(g); MRG(ra_, s->albedo, 61);
        if (ra_ > s->albedo) {
            if (s->save_Jabs) { double uu1 = udot(s, c, p->kx, p->ky, p->kz), xr = (p->xfreq + uu1) * (s->Dfreq[c] / s->Dfreq_ref);
                int ix = (int)floor((xr - s->xfreq_min) / s->dxfreq) + 1; if (ix >= 1 && ix <= s->nxfreq) A->Jabs[ix-1] += p->wgt; }
            if (s->save_all_photons) {                      /* scattering_car.f90:236-246, :525-535: lab-frame frequency at absorption, weight 0 */
                double uu1 = udot(s, c, p->kx, p->ky, p->kz); p->xfreq_ref = (p->xfreq + uu1) * (s->Dfreq[c] / s->Dfreq_ref); p->wgt = 0.0; }
            if (s->line_type == 8) A->W[2] += p->wgt;           /* W_abs1 :524 */
            p->inside = 0; return 1; }
    } else {
        if (s->line_type == 8) A->W[2] += p->wgt * (1.0 - s->albedo);       /* :553 */
        if (s->save_Jabs) { double uu1 = udot(s, c, p->kx, p->ky, p->kz), xr = (p->xfreq + uu1) * (s->Dfreq[c] / s->Dfreq_ref);
            int ix = (int)floor((xr - s->xfreq_min) / s->dxfreq) + 1; if (ix >= 1 && ix <= s->nxfreq) A->Jabs[ix-1] += p->wgt * (1.0 - s->albedo); }
        p->wgt = p->wgt * s->albedo;
    }
    return 0;
}
/* scatter_dust_nostokes: :488-584 */
static void scatter_dust_nostokes(const orc_sim *s, acc_t *A, rng_t *g, photon_t *p) {
    if (dust_absorb(s, A, g, p)) return;
    if (s->save_peeloff) peeling_dust_nostokes(s, A, p);
    double cost = rand_hg(g, s->hgg), sint = sqrt(1.0 - cost * cost);
    double phi = TWOPI * rnd(g), cosp = cos(phi), sinp = sin(phi);
    rotate_k(p, cost, sint, cosp, sinp);
}
/* scatter_dust_stokes: :201-329 */
static void scatter_dust_stokes(const orc_sim *s, acc_t *A, rng_t *g, photon_t *p) {
    if (dust_absorb(s, A, g, p)) return;
    if (s->save_peeloff) peeling_dust_stokes(s, A, p);
    double cost = rand_alias_linear(g, s->nPDF - 1, s->phase_PDF, s->alias, s->coss, s->S11), sint = sqrt(1.0 - cost * cost);
    double S11 = interp_eq(s->nPDF, s->coss, s->S11, cost), S12 = interp_eq(s->nPDF, s->coss, s->S12, cost);
    double S33 = interp_eq(s->nPDF, s->coss, s->S33, cost), S34 = interp_eq(s->nPDF, s->coss, s->S34, cost);
    double phi = phi_rejection(g, p, S12 / S11), cosp = cos(phi), sinp = sin(phi);
    double cos2p = 2.0 * cosp * cosp - 1.0, sin2p = 2.0 * sinp * cosp;
    double Q0 = cos2p * p->Q + sin2p * p->U, U0 = -sin2p * p->Q + cos2p * p->U;
    double I1 = S11 + S12 * Q0, Q1 = S12 + S11 * Q0, U1 = S33 * U0 + S34 * p->V, V1 = -S34 * U0 + S33 * p->V;
    p->I = 1.0; p->Q = Q1 / I1; p->U = U1 / I1; p->V = V1 / I1;
    rotate_triad(p, cost, sint, cosp, sinp);
}
/* scattering: scattering_car.f90:14-120 (Cartesian branch :106-118) */
static void scattering(const orc_sim *s, acc_t *A, rng_t *g, photon_t *p) {
    A->n_events += 1.0;
    if (p->iband == 2) { scatter_dust_Ha(s, A, g, p); return; }          /* :30-36: every interaction of a band-2 packet is a dust event */
    if (s->h2_on && !s->use_clump_medium) {                                /* :44-57 (octree), :92-104 (Cartesian): three-way split H I / H2 / dust */
        const size_t c = IDX(s,p->icell,p->jcell,p->kcell);
        const double kap_HI = s->rhokap[c] * calc_voigt(s, p->xfreq, s->voigt_a[c], s->Dfreq[c]), kap_H2 = s->rhokap[c] * h2_kappa_D(s, p->xfreq, s->Dfreq[c]);
        const double kap_D = (s->DGR > 0.0) ? s->rhokapD[c] : 0.0;
        const double l_ = rnd(g) * (kap_HI + kap_H2 + kap_D); MRG(l_, kap_H2, 69);
        if (l_ <= kap_H2) { scatter_h2(s, A, g, p); return; }
    }
    if (s->use_clump_medium) {                                   /* scattering_car.f90:69-84 */
        if (s->DGR > 0.0) {
            int64_t icl = p->icell_clump; const clumps_t *c = &s->cl;
            double p_dust = c->rhokapD[icl-1] / (c->rhokap[icl-1] * voigt_clump(s, p->xfreq, icl) + c->rhokapD[icl-1]);
            const double rd_ = rnd(g); MRG(rd_, p_dust, 60);
            if (rd_ <= p_dust) { if (s->use_stokes && s->nPDF > 0) scatter_dust_stokes(s, A, g, p); else scatter_dust_nostokes(s, A, g, p); return; }
        }
        if (s->cl.has_overlap) {                                  /* scatter_resonance_clump_{no,}stokes: scattering_car.f90:898-937 */
            const int64_t icl = p->icell_clump;
            p->xfreq = p->xfreq - ulos_clump(&s->cl, icl, p->kx, p->ky, p->kz);
            if (s->use_stokes) scatter_resonance_stokes(s, A, g, p); else scatter_resonance_nostokes(s, A, g, p);
            p->xfreq = p->xfreq + ulos_clump(&s->cl, icl, p->kx, p->ky, p->kz);
            return;
        }
        if (s->use_stokes) scatter_resonance_stokes(s, A, g, p); else scatter_resonance_nostokes(s, A, g, p);
        return;
    }
    if (s->DGR > 0.0) {
        size_t c = IDX(s,p->icell,p->jcell,p->kcell);
        double p_dust = s->rhokapD[c] / (s->rhokap[c] * calc_voigt(s, p->xfreq, s->voigt_a[c], s->Dfreq[c]) + s->rhokapD[c]);
        const double rd_ = rnd(g); MRG(rd_, p_dust, 60);
            if (rd_ <= p_dust) { if (s->use_stokes && s->nPDF > 0) scatter_dust_stokes(s, A, g, p); else scatter_dust_nostokes(s, A, g, p); return; }
    }
    if (s->use_stokes) scatter_resonance_stokes(s, A, g, p); else scatter_resonance_nostokes(s, A, g, p);
}

/* =====================================================================================
 * Photon birth and the photon loop
 * ===================================================================================== */
/* setup_isotropic_injection: generate_photon.f90:342-408 */
static void setup_isotropic_injection(const orc_sim *s, rng_t *g, photon_t *p) {
    p->wgt = 1.0;
    if (!s->use_amr_grid && s->xyz_symmetry) { if (p->x < s->xmin) p->x = -p->x; if (p->y < s->ymin) p->y = -p->y; if (p->z < s->zmin) p->z = -p->z; }
    double cost = 2.0 * rnd(g) - 1.0, sint = sqrt(1.0 - cost * cost), phi = TWOPI * rnd(g), cosp = cos(phi), sinp = sin(phi);
    p->kx = sint * cosp; p->ky = sint * sinp; p->kz = cost;
    if (s->use_amr_grid) { p->icell = amr_find_leaf(&s->amr, p->x, p->y, p->z); p->jcell = 1; p->kcell = 1; goto triad; }   /* :375-379 */
    p->icell = (int)floor((p->x - s->xmin) / s->dx) + 1; p->jcell = (int)floor((p->y - s->ymin) / s->dy) + 1; p->kcell = (int)floor((p->z - s->zmin) / s->dz) + 1;
    if (p->kx < 0.0 && p->icell == s->nx + 1) p->icell = s->nx;
    if (p->ky < 0.0 && p->jcell == s->ny + 1) p->jcell = s->ny;
    if (p->kz < 0.0 && p->kcell == s->nz + 1) p->kcell = s->nz;
    if (p->kx > 0.0 && p->icell < 1) p->icell = 1;
    if (p->ky > 0.0 && p->jcell < 1) p->jcell = 1;
    if (p->kz > 0.0 && p->kcell < 1) p->kcell = 1;
triad:
    if (s->use_stokes) { p->mx = cost * cosp; p->my = cost * sinp; p->mz = -sint; p->nx = -sinp; p->ny = cosp; p->nz = 0.0; p->I = 1.0; p->Q = 0.0; p->U = 0.0; p->V = 0.0; }
}
/* generate_photon: generate_photon.f90:3-339 (point / uniform_sphere sources, voigt / monochromatic spectra) */
static void generate_photon(const orc_sim *s, acc_t *A, rng_t *g, photon_t *p) {
    if (s->source_geometry == 1) {                                                   /* :34-42 */
        double rp = pow(rnd(g), 1.0 / 3.0) * s->source_rmax, cost = 2.0 * rnd(g) - 1.0, sint = sqrt(1.0 - cost * cost), phi = TWOPI * rnd(g);
        p->x = rp * sint * cos(phi); p->y = rp * sint * sin(phi); p->z = rp * cost;
    } else if (s->source_geometry == 2) {                                            /* uniform_cylinder :43-49 */
        double rp = sqrt(rnd(g)) * s->source_rmax, phi = TWOPI * rnd(g);
        p->x = rp * cos(phi); p->y = rp * sin(phi); p->z = s->zrange * rnd(g) + s->zmin;
    } else if (s->source_geometry == 3) {                                            /* uniform :50-54 */
        p->x = s->xrange * rnd(g) + s->xmin; p->y = s->yrange * rnd(g) + s->ymin; p->z = s->zrange * rnd(g) + s->zmin;
    } else if (s->source_geometry == 4) {                                            /* uniform_xy :55-67 */
        if (s->source_rmax > 0.0) { double rp = sqrt(rnd(g)) * s->source_rmax, phi = TWOPI * rnd(g); p->x = rp * cos(phi); p->y = rp * sin(phi); }
        else { p->x = s->xrange * rnd(g) + s->xmin; p->y = s->yrange * rnd(g) + s->ymin; }
        p->z = 0.0;
    } else if (s->source_geometry == 5) {                                            /* gaussian (in z) :68-72 */
        p->x = s->xrange * rnd(g) + s->xmin; p->y = s->yrange * rnd(g) + s->ymin; p->z = s->source_zscale / sqrt(2.0) * rand_gauss(g);
    } else { p->x = s->xs_point; p->y = s->ys_point; p->z = s->zs_point; }             /* :126-131 */
    setup_isotropic_injection(s, g, p);
    p->nscatt_gas = 0.0; p->nscatt_dust = 0.0; p->inside = 1; p->xfreq = s->xfreq0;
    p->E1 = 0.0; p->E2 = 1.0; p->E3 = 2.0 / 3.0;                                       /* photon_type defaults, define.f90:108-110 */
    size_t c = IDX(s,p->icell,p->jcell,p->kcell);
    if (s->spectral_type != 2) {                            /* :160-243 (not for a continuum source) */
        const double Dfreq_local = s->Dfreq[c];
        if (s->line_type == 2) { double DnuHK = s->DnuHK_Hz / Dfreq_local; if (rnd(g) <= 1.0 / 3.0) p->xfreq = p->xfreq - DnuHK; }   /* :162-167: the source feeds both levels */
        else if (s->line_type == 4) {                       /* :181-188 */
            int n = s->ml.ndown[0], k = (int)floor(n * rnd(g)) + 1, idown = (rnd(g) < s->ml.P_down[0][k-1]) ? k : s->ml.A_down[0][k-1];
            if (idown != 1) p->xfreq = p->xfreq - s->ml.Elow_Hz[0][idown-1] / Dfreq_local;
        } else if (s->line_type == 5) {                     /* :189-215 */
            double p1 = s->ml.f12[0], p2 = s->ml.f12[1]; int iup;
            p1 = p1 / (p2 + p1);
            if (rnd(g) < p1) iup = 1; else { iup = 2; p->xfreq = p->xfreq - s->ml.delE_Hz[1] / Dfreq_local; }
            if (s->ml.ndown[iup-1] > 1) {
                int n = s->ml.ndown[iup-1], k = (int)floor(n * rnd(g)) + 1, idown = (rnd(g) < s->ml.P_down[iup-1][k-1]) ? k : s->ml.A_down[iup-1][k-1];
                p->xfreq = p->xfreq - s->ml.Elow_Hz[iup-1][idown-1] / Dfreq_local;
            }
        } else if (s->line_type == 6) {                     /* :216-242 */
            double p1 = s->ml.f12[0], p2 = s->ml.f12[1], p3 = s->ml.f12[2], ptot = p1 + p2 + p3, pcum1 = p1 / ptot, pcum2 = (p1 + p2) / ptot, xi = rnd(g); MRG(xi, pcum1, 63); MRG(xi, pcum2, 63);
            if (xi < pcum1) {} else if (xi < pcum2) p->xfreq = p->xfreq - s->ml.delE_Hz[1] / Dfreq_local; else p->xfreq = p->xfreq - s->ml.delE_Hz[2] / Dfreq_local;
        }
    }
    if (s->spectral_type == 1) p->xfreq = p->xfreq + rand_voigt(g, s->voigt_a[c]);     /* :252-253 */
    else if (s->spectral_type == 2) {                                                /* continuum :244-247, in units of the local cell */
        p->xfreq = rnd(g) * (s->xfreq_max - s->xfreq_min) + s->xfreq_min;
        p->xfreq = p->xfreq / (s->Dfreq[c] / s->Dfreq_ref);
    } else if (s->spectral_type == 3) {                                              /* gaussian :254-266 */
        double gsig = s->gaussian_FWHM_vel > 0.0 ? s->gaussian_FWHM_vel / 2.3548200450309493 : s->gaussian_sigma_vel;
        p->xfreq = p->xfreq + rand_gauss(g) * (gsig / vtherm_total(s, s->temperature));
        p->xfreq = p->xfreq / (s->Dfreq[c] / s->Dfreq_ref);
    }
    double u1;
    if (!s->comoving_source) { u1 = udot(s, c, p->kx, p->ky, p->kz); p->xfreq = p->xfreq - u1; }   /* :303-306 */
    if (s->save_Jin) {                                                               /* :309-322 */
        u1 = udot(s, c, p->kx, p->ky, p->kz);
        double xlab = (p->xfreq + u1) * (s->Dfreq[c] / s->Dfreq_ref);
        int ix = (int)floor((xlab - s->xfreq_min) / s->dxfreq) + 1;
        if (ix >= 1 && ix <= s->nxfreq) A->Jin[ix-1] += p->wgt;
    }
    p->icell_clump = 0;
    if (s->use_clump_medium && !s->cl.has_overlap) {                                 /* :325-332 */
        int64_t icl = clump_at_point(&s->cl, p->x, p->y, p->z);
        if (icl > 0) { p->icell_clump = (int)icl; p->xfreq = p->xfreq - ulos_clump(&s->cl, icl, p->kx, p->ky, p->kz); }
    }
    if (s->save_peeloff) peeling_direct(s, A, p);                                      /* :334-336 */
}
/* add_escaped_fraction_to_Jout: run_simulation_mod.f90:208-247 */
static void add_escaped_fraction_to_Jout(const orc_sim *s, acc_t *A, const photon_t *p, double tau0) {
    double wgt_esc = p->wgt * exp(-tau0);
    size_t c = IDX(s,p->icell,p->jcell,p->kcell);
    double u1 = udot(s, c, p->kx, p->ky, p->kz), xr = (p->xfreq + u1) * (s->Dfreq[c] / s->Dfreq_ref);
    int ix = (int)floor((xr - s->xfreq_min) / s->dxfreq) + 1;
    if (ix >= 1 && ix <= s->nxfreq) { A->Jout[ix-1] += wgt_esc;
        if (s->save_Jmu) { double mu = p->kz; if (s->xyz_symmetry) mu = fabs(mu);
            int imu = (int)floor((mu - s->mu_min) / s->dmu) + 1; if (imu < 1) imu = 1; if (imu > s->nmu) imu = s->nmu;
            A->Jmu[(size_t)(ix - 1) + (size_t)s->nxfreq * (imu - 1)] += wgt_esc; } }
}
/* one photon: run_equal_number body, run_simulation_mod.f90:155-192 */
/* make_all_initial_photons / make_all_photons: run_simulation_mod.f90:249-358 */
static double allph_impact(const orc_sim *s, const photon_t *p, double *mx, double *my, double *mz) {
    double xp = p->x, yp = p->y, zp = p->z;
    if (s->rmax > 0.0) {
        double rr = p->x * p->x + p->y * p->y + p->z * p->z, dist = 0.0;
        if (rr > s->rmax * s->rmax) { double rk = p->x * p->kx + p->y * p->ky + p->z * p->kz, det = rk * rk - (rr - s->rmax * s->rmax);
            dist = det < 0.0 ? 0.0 : -rk + sqrt(fmax(0.0, det)); }
        xp = p->x + dist * p->kx; yp = p->y + dist * p->ky; zp = p->z + dist * p->kz;
    }
    double rk = xp * p->kx + yp * p->ky + zp * p->kz;
    *mx = xp - rk * p->kx; *my = yp - rk * p->ky; *mz = zp - rk * p->kz;
    return sqrt((*mx) * (*mx) + (*my) * (*my) + (*mz) * (*mz));
}
/* history probe (orc_trace): one record after the birth (kind 0), after every scattering (1) and when the packet has left or was absorbed (2):
 * draws consumed, xfreq (xfreq_ref once outside), x, y, z, cell code (icell + 2^21 jcell + 2^42 kcell; the owner clump in clump media), kz,
 * kind, and the smallest decision margin of the event with the id of that comparison (mrg_note) */
#define ORC_TRACE_W 10
static __thread double *g_tr = NULL; static __thread int g_tr_n = 0, g_tr_cap = 0;
/* orc_trace_state: the FULL packet state at record g_st_at (what a single event needs to start from: the one-step re-synchronised check of the
 * divergence audit, tests/test_gpu_divergence.py) -- [0..5] x y z kx ky kz, [6] xfreq, [7] wgt, [8..13] m n, [14..16] Q U V, [17..18] nscatt_gas / _dust,
 * [19] draws consumed, [20..22] icell jcell kcell, [23] owner clump, [24] vfy_shear, [25] iband, [26] kind of the record */
#define ORC_STATE_W 27
static __thread double *g_st_out = NULL; static __thread int g_st_at = -1;
static inline void trace_rec(const orc_sim *s, const rng_t *g, const photon_t *p, int kind) {
    if (g_st_out && g_tr_n == g_st_at) { double *q = g_st_out;
        q[0] = p->x; q[1] = p->y; q[2] = p->z; q[3] = p->kx; q[4] = p->ky; q[5] = p->kz; q[6] = p->xfreq; q[7] = p->wgt;
        q[8] = p->mx; q[9] = p->my; q[10] = p->mz; q[11] = p->nx; q[12] = p->ny; q[13] = p->nz; q[14] = p->Q; q[15] = p->U; q[16] = p->V;
        q[17] = p->nscatt_gas; q[18] = p->nscatt_dust; q[19] = (double)g->n; q[20] = p->icell; q[21] = p->jcell; q[22] = p->kcell; q[23] = p->icell_clump;
        q[24] = p->vfy_shear; q[25] = p->iband == 2 ? 2.0 : 1.0; q[26] = (double)kind; }
    if (!g_tr) return;
    if (g_tr_n < g_tr_cap) { double *r = g_tr + (size_t)g_tr_n * ORC_TRACE_W;
        r[0] = (double)g->n; r[1] = p->inside ? p->xfreq : p->xfreq_ref; r[2] = p->x; r[3] = p->y; r[4] = p->z;
        r[5] = s->use_clump_medium ? (double)p->icell_clump : (double)p->icell + 2097152.0 * (double)p->jcell + 4398046511104.0 * (double)p->kcell;
        r[6] = p->kz; r[7] = (double)kind; r[8] = g_mrg; r[9] = (double)g_mrg_id; }
    g_tr_n++; g_mrg = 1e300; g_mrg_id = 0;
}
static void run_photon(const orc_sim *s, acc_t *A, rng_t *g, photon_t *p, int64_t k) {
    A->g = g;
    generate_photon(s, A, g, p);
    trace_rec(s, g, p, 0);
    const int64_t n = s->log_n; double *R = s->save_all_photons ? s->allph : NULL;
    if (R) { R[1 * n + k] = p->xfreq; if (s->source_geometry != 0) { double mx, my, mz; R[9 * n + k] = allph_impact(s, p, &mx, &my, &mz); } }
    int first = 1; double tau, tau0, wgt1;
    while (p->inside) {
        if (first) {
            tau0 = raytrace_to_edge(s, p, s->raytrace_variant, -1.0, &A->n_cellsteps);
            add_escaped_fraction_to_Jout(s, A, p, tau0);
            if (s->line_type == 8) A->W[1] += p->wgt * exp(-tau0);             /* W_esc1: run_simulation_mod.f90:168 */
            wgt1 = 1.0 - exp(-tau0);
            p->wgt = p->wgt * wgt1;
            if (tau0 > 0.0) tau = -log(1.0 - rnd(g) * wgt1); else tau = HUGEST;
            first = 0;
        } else tau = -log(rnd(g));
        raytrace_to_tau(s, A, p, tau, s->raytrace_variant);
        if (p->inside) scattering(s, A, g, p);
        trace_rec(s, g, p, p->inside ? 1 : 2);
    }
    A->nscatt_gas += p->nscatt_gas; A->nscatt_dust += p->nscatt_dust;
    if (R) { double mx, my, mz, mm = allph_impact(s, p, &mx, &my, &mz), cos2p = 1.0, sin2p = 0.0;
        R[k] = mm; R[2 * n + k] = p->xfreq_ref; R[3 * n + k] = p->nscatt_gas; R[4 * n + k] = p->nscatt_dust;
        if (s->use_stokes) {
            if (mm > 0.0) { mx = mx / mm; my = my / mm; mz = mz / mm;
                double cosp = mx * p->mx + my * p->my + mz * p->mz, sinp = mx * p->nx + my * p->ny + mz * p->nz;
                cos2p = 2.0 * cosp * cosp - 1.0; sin2p = 2.0 * sinp * cosp; }
            R[5 * n + k] = p->wgt; R[6 * n + k] = (cos2p * p->Q + sin2p * p->U) * p->wgt; R[7 * n + k] = (-sin2p * p->Q + cos2p * p->U) * p->wgt; R[8 * n + k] = p->V * p->wgt; } }
}

static void acc_alloc(acc_t *A, orc_sim *s) {
    memset(A, 0, sizeof *A); A->s = s;
    A->Jout = calloc(s->nxfreq, 8); A->Jin = calloc(s->nxfreq, 8); A->Jabs = calloc(s->nxfreq, 8); A->Jmu = calloc((size_t)s->nxfreq * (s->nmu > 0 ? s->nmu : 1), 8);
    if (s->J1) A->J1 = calloc((size_t)s->nxfreq * s->nbins, 8); if (s->P1) A->P1 = calloc(s->nbins, 8); if (s->P1_new) A->P1_new = calloc(s->nbins, 8);
    if (s->Jout_Ha) { A->Jout_Ha = calloc(s->nxfreq_Ha, 8); A->Jabs_Ha = calloc(s->nxfreq_Ha, 8); } if (s->Pconv) A->Pconv = calloc(s->nbins, 8);
    size_t n2 = (size_t)s->nxim * s->nyim, n3 = n2 * s->nxfreq;
    for (int i = 0; i < s->nobs; i++) { obs_t *o = &A->obs[i], *q = &s->obs[i];
        if (q->scatt_2D) { o->scatt_2D = calloc(n2, 8); o->direc_2D = calloc(n2, 8); }
        if (q->I_2D) { o->I_2D = calloc(n2, 8); o->Q_2D = calloc(n2, 8); o->U_2D = calloc(n2, 8); o->V_2D = calloc(n2, 8); }
        if (q->scatt) { o->scatt = calloc(n3, 8); o->direc = calloc(n3, 8); }
        if (q->direc0) o->direc0 = calloc(n3, 8); if (q->direc0_2D) o->direc0_2D = calloc(n2, 8); if (q->scatt_dust) o->scatt_dust = calloc(n3, 8);
        if (q->peel_Ha) o->peel_Ha = calloc(n2 * s->nxfreq_Ha, 8);
        if (q->I) { o->I = calloc(n3, 8); o->Q = calloc(n3, 8); o->U = calloc(n3, 8); o->V = calloc(n3, 8); } }
}
static void addv(double *dst, const double *src, size_t n) { if (dst && src) for (size_t i = 0; i < n; i++) dst[i] += src[i]; }
static void acc_merge_free(acc_t *A, orc_sim *s) {
    addv(s->Jout, A->Jout, s->nxfreq); addv(s->Jin, A->Jin, s->nxfreq); addv(s->Jabs, A->Jabs, s->nxfreq); addv(s->Jmu, A->Jmu, (size_t)s->nxfreq * (s->nmu > 0 ? s->nmu : 1));
    size_t n2 = (size_t)s->nxim * s->nyim, n3 = n2 * s->nxfreq;
    for (int i = 0; i < s->nobs; i++) { obs_t *o = &A->obs[i], *q = &s->obs[i];
        addv(q->scatt_2D, o->scatt_2D, n2); addv(q->direc_2D, o->direc_2D, n2); addv(q->I_2D, o->I_2D, n2); addv(q->Q_2D, o->Q_2D, n2); addv(q->U_2D, o->U_2D, n2); addv(q->V_2D, o->V_2D, n2);
        addv(q->direc0, o->direc0, n3); addv(q->direc0_2D, o->direc0_2D, n2); addv(q->scatt_dust, o->scatt_dust, n3); addv(q->peel_Ha, o->peel_Ha, n2 * s->nxfreq_Ha);
        addv(q->scatt, o->scatt, n3); addv(q->direc, o->direc, n3); addv(q->I, o->I, n3); addv(q->Q, o->Q, n3); addv(q->U, o->U, n3); addv(q->V, o->V, n3);
        free_obs(o); }
    s->nscatt_gas += A->nscatt_gas; s->nscatt_dust += A->nscatt_dust;
    s->n_cellsteps += A->n_cellsteps; s->n_cellsteps_peel += A->n_cellsteps_peel; s->n_events += A->n_events;
    addv(s->J1, A->J1, (size_t)s->nxfreq * s->nbins); addv(s->P1, A->P1, s->nbins); addv(s->P1_new, A->P1_new, s->nbins);
    addv(s->Jout_Ha, A->Jout_Ha, s->nxfreq_Ha); addv(s->Jabs_Ha, A->Jabs_Ha, s->nxfreq_Ha); addv(s->Pconv, A->Pconv, s->nbins);
    for (int i = 0; i < 5; i++) s->W[i] += A->W[i]; for (int i = 0; i < 6; i++) s->W_H2[i] += A->W_H2[i];
    free(A->Jout); free(A->Jin); free(A->Jabs); free(A->Jmu); free(A->J1); free(A->P1); free(A->P1_new); free(A->Jout_Ha); free(A->Jabs_Ha); free(A->Pconv);
}

int orc_run(orc_sim *s, int64_t first_id, int64_t stride, int64_t count, int rng_kind, uint64_t seed, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    free(s->log_nscatt); free(s->log_xfreq);
    free(s->allph); s->allph = s->save_all_photons ? calloc((size_t)(count > 0 ? count : 1) * 10, 8) : NULL;
    s->log_n = count; s->log_nscatt = calloc(count > 0 ? count : 1, 8); s->log_xfreq = calloc(count > 0 ? count : 1, 8);
    acc_t *accs = malloc(nthreads * sizeof(acc_t));
    for (int t = 0; t < nthreads; t++) acc_alloc(&accs[t], s);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        acc_t *A = &accs[tid]; rng_t g; mt_t mt; photon_t p;
        if (rng_kind == 0) {   /* static interleave over workers, seed + 9999*worker: run_simulation_mod.f90:150-151, random_mt.f90:940-953 */
            memset(&g, 0, sizeof g); g.kind = 0; g.mt = &mt; mt_init(&mt, (int64_t)(int32_t)((int32_t)seed + 9999 * tid));
            for (int64_t k = tid; k < count; k += nthreads) { memset(&p, 0, sizeof p); p.id = (double)(first_id + k * stride);
                run_photon(s, A, &g, &p, k); s->log_nscatt[k] = p.nscatt_gas + p.nscatt_dust; s->log_xfreq[k] = p.xfreq_ref; }
        } else {
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
            for (int64_t k = 0; k < count; k++) { memset(&p, 0, sizeof p); uint64_t id = (uint64_t)(first_id + k * stride); p.id = (double)id;
                rng_philox(&g, seed, id, 0); run_photon(s, A, &g, &p, k); s->log_nscatt[k] = p.nscatt_gas + p.nscatt_dust; s->log_xfreq[k] = p.xfreq_ref; }
        }
    }
    for (int t = 0; t < nthreads; t++) acc_merge_free(&accs[t], s);
    free(accs);
    s->nphotons_done += count;
    return 0;
}
/* the history of ONE photon on the counter-based stream (seed, id), with the tallies going into a scratch accumulator that is
 * thrown away: out[ORC_TRACE_W * cap]; returns the number of records the history has (those beyond cap are not stored) */
int64_t orc_trace(orc_sim *s, uint64_t seed, uint64_t id, int64_t cap, double *out) {
    acc_t A; rng_t g; photon_t p;
    double *ln = s->log_nscatt, *lx = s->log_xfreq, *al = s->allph; int64_t lnn = s->log_n;
    double one[2] = {0.0, 0.0}; s->log_nscatt = one; s->log_xfreq = one + 1; s->log_n = 1; s->allph = NULL;
    const int sap = s->save_all_photons; s->save_all_photons = 0;
    acc_alloc(&A, s);
    memset(&p, 0, sizeof p); p.id = (double)id; rng_philox(&g, seed, id, 0);
    g_tr = out; g_tr_n = 0; g_tr_cap = (int)cap; g_mrg_on = 1; g_mrg = 1e300; g_mrg_id = 0;
    run_photon(s, &A, &g, &p, 0);
    const int64_t n = g_tr_n;
    g_tr = NULL; g_tr_n = 0; g_tr_cap = 0; g_mrg_on = 0;
    /* discard the scratch tallies (acc_merge_free would add them): free what acc_alloc allocated */
    for (int i = 0; i < s->nobs; i++) free_obs(&A.obs[i]);
    free(A.Jout); free(A.Jin); free(A.Jabs); free(A.Jmu); free(A.J1); free(A.P1); free(A.P1_new); free(A.Jout_Ha); free(A.Jabs_Ha); free(A.Pconv);
    s->log_nscatt = ln; s->log_xfreq = lx; s->log_n = lnn; s->allph = al; s->save_all_photons = sap;
    return n;
}
/* the full state of photon (seed, id) at record `at` of its history (0 = after the birth): out[ORC_STATE_W]; returns the number of records of the history */
int64_t orc_trace_state(orc_sim *s, uint64_t seed, uint64_t id, int64_t at, double *out) {
    double scratch[ORC_TRACE_W];
    g_st_out = out; g_st_at = (int)at;
    const int64_t n = orc_trace(s, seed, id, 1, scratch);
    g_st_out = NULL; g_st_at = -1;
    return n;
}
int64_t orc_get_photon_log(const orc_sim *s, double *nscatt, double *xfreq_out, int64_t n) {
    int64_t m = s->log_n < n ? s->log_n : n;
    for (int64_t i = 0; i < m; i++) { if (nscatt) nscatt[i] = s->log_nscatt[i]; if (xfreq_out) xfreq_out[i] = s->log_xfreq[i]; }
    return s->log_n;
}

void orc_zero_tallies(orc_sim *s) {
    memset(s->Jout, 0, s->nxfreq * 8); memset(s->Jin, 0, s->nxfreq * 8); memset(s->Jabs, 0, s->nxfreq * 8); memset(s->Jmu, 0, (size_t)s->nxfreq * (s->nmu > 0 ? s->nmu : 1) * 8);
    size_t n2 = (size_t)s->nxim * s->nyim, n3 = n2 * s->nxfreq;
    for (int i = 0; i < s->nobs; i++) { obs_t *o = &s->obs[i];
        double *a2[6] = {o->scatt_2D, o->direc_2D, o->I_2D, o->Q_2D, o->U_2D, o->V_2D}, *a3[6] = {o->scatt, o->direc, o->I, o->Q, o->U, o->V};
        for (int k = 0; k < 6; k++) { if (a2[k]) memset(a2[k], 0, n2 * 8); if (a3[k]) memset(a3[k], 0, n3 * 8); }
        if (o->direc0) memset(o->direc0, 0, n3 * 8); if (o->direc0_2D) memset(o->direc0_2D, 0, n2 * 8); if (o->scatt_dust) memset(o->scatt_dust, 0, n3 * 8);
        if (o->peel_Ha) memset(o->peel_Ha, 0, n2 * s->nxfreq_Ha * 8); }
    if (s->Jout_Ha) { memset(s->Jout_Ha, 0, s->nxfreq_Ha * 8); memset(s->Jabs_Ha, 0, s->nxfreq_Ha * 8); } if (s->Pconv) memset(s->Pconv, 0, s->nbins * 8);
    memset(s->W, 0, sizeof s->W); memset(s->W_H2, 0, sizeof s->W_H2);
    if (s->J1) memset(s->J1, 0, (size_t)s->nxfreq * s->nbins * 8); if (s->P1) memset(s->P1, 0, s->nbins * 8); if (s->P1_new) memset(s->P1_new, 0, s->nbins * 8);
    s->nscatt_gas = s->nscatt_dust = 0.0; s->nphotons_done = 0; s->n_cellsteps = s->n_cellsteps_peel = s->n_events = 0.0;
}

/* output_normalize_outside, Jout only: output_sum_rect.f90:185-202 */
int orc_normalized_Jout(const orc_sim *s, double *out, int n) {
    double nph = (double)s->nphotons_done, area, scale;
    if (s->xy_periodic) area = 2.0;
    else if (s->geometry == 0) area = FOURPI * s->rmax * s->rmax * s->distance2cm * s->distance2cm;
    else if (s->use_amr_grid) area = 6.0 * s->amr.L_box * s->amr.L_box * s->distance2cm * s->distance2cm;   /* output_normalize_amr :808-813: the six faces of the box */
    else area = (s->xmax * s->ymax + s->ymax * s->zmax + s->zmax * s->xmax) * 8.0 * s->distance2cm * s->distance2cm;
    scale = nph * s->dxfreq * TWOPI * area;
    for (int i = 0; i < n && i < s->nxfreq; i++) out[i] = s->Jout[i] / scale;
    return s->nxfreq;
}

double orc_get_scalar(const orc_sim *s, const char *k) {
#define G(name) if (!strcmp(k, #name)) return (double)s->name;
    G(nscatt_gas) G(nscatt_dust) G(nphotons_done) G(n_cellsteps) G(n_cellsteps_peel) G(n_events)
    G(dx) G(dy) G(dz) G(xmin) G(ymin) G(zmin) G(i0) G(j0) G(k0) G(nx) G(ny) G(nz) G(Dfreq_ref) G(xfreq_min) G(xfreq_max) G(dxfreq) G(nxfreq)
    G(taumax) G(tauhomo) G(N_gasmax) G(N_gashomo) G(voigt_amean) G(atau3) G(nobs) G(dxim) G(dyim) G(distance) G(nxim) G(nyim)
    G(line_type) G(nxfreq_Ha) G(xfreq_min_Ha) G(xfreq_max_Ha) G(dxfreq_Ha) G(R_Ha) G(albedo_Ha) G(hgg_Ha) G(cext_dust_Ha) G(P_conv) G(h2_on) G(h2_nlines) G(h2_Dfreq_Hz) G(h2_hi_width) G(Omega) G(damping) G(wavelength0) G(vtherm1)
    G(rmax) G(xmax) G(ymax) G(zmax) G(DGR) G(albedo) G(hgg) G(cext_dust) G(cross0) G(g_recoil0) G(xcrit) G(raytrace_variant) G(save_peeloff)
    G(use_stokes) G(nPDF) G(save_Jabs) G(nmu) G(mu_min) G(dmu) G(distance2cm) G(source_rmax) G(comoving_source) G(recoil) G(core_skip) G(core_skip_global)
    G(xyz_symmetry) G(xy_periodic) G(save_Jin) G(save_Jmu) G(use_reduced_wgt) G(xs_point) G(ys_point) G(zs_point) G(xfreq0) G(spectral_type) G(source_geometry) G(xrange) G(yrange) G(zrange) G(source_zscale)
    if (s->use_clump_medium) { const clumps_t *c = &s->cl;
        if (!strcmp(k, "cl_N")) return (double)c->N; if (!strcmp(k, "cl_nreg")) return (double)c->nreg; if (!strcmp(k, "cl_cgx")) return c->cgx;
        if (!strcmp(k, "cl_rhokap_ref")) return c->rhokap_ref; if (!strcmp(k, "cl_voigt_a_ref")) return c->voigt_a_ref; if (!strcmp(k, "cl_Dfreq_ref")) return c->Dfreq_ref;
        if (!strcmp(k, "cl_vtherm_ref")) return c->vtherm_ref; if (!strcmp(k, "cl_fvol")) return c->fvol; if (!strcmp(k, "cl_fcov")) return c->fcov; if (!strcmp(k, "cl_has_overlap")) return c->has_overlap;
        if (!strcmp(k, "cl_rsa_grid")) return c->rsa_grid; if (!strcmp(k, "cl_rsa_acceptance")) return c->rsa_acceptance;
        if (!strcmp(k, "cg_min")) return c->cg_min; if (!strcmp(k, "cg_d")) return c->cg_d; if (!strcmp(k, "cl_sphere_R")) return c->sphere_R; if (!strcmp(k, "cl_radius_max")) return c->radius_max; }
    if (!strcmp(k, "amr_ncells")) return s->amr.ncells; if (!strcmp(k, "amr_nleaf")) return s->amr.nleaf; if (!strcmp(k, "amr_levelmax")) return s->amr.levelmax;
    if (!strcmp(k, "amr_dr_JPa")) return s->amr.dr_JPa; if (!strcmp(k, "amr_dz_JPa")) return s->amr.dz_JPa; if (!strcmp(k, "amr_boxlen")) return s->amr.L_box;
    G(wavelength0) G(damping) G(cross0) G(vtherm1) G(g_recoil0)
    G(line_type) G(DnuHK_Hz) G(fine_structure) G(include_deuterium) G(delta_nu_HD_Hz) G(ratio_Dfreq_HD) G(ratio_voigta_HD) G(nD_over_nH) G(g_recoil0_D)
    G(use_clump_medium) G(tauhomo) G(N_gashomo) G(use_amr_grid) G(save_peeloff_2D) G(save_peeloff_3D) G(E1) G(E2) G(E3) G(nbins) G(geometry_JPa) G(calc_J) G(calc_P) G(calc_Pnew) G(nside) G(npix)
#undef G
    return NAN;
}
static int64_t copy_out(const double *src, int64_t cnt, double *out, int64_t n) { if (!src) return -1; for (int64_t i = 0; i < cnt && i < n; i++) out[i] = src[i]; return cnt; }
int64_t orc_get_array(const orc_sim *s, const char *k, double *out, int64_t n) {
    int64_t nc = (int64_t)s->nx * s->ny * s->nz;
    if (!strcmp(k, "allph")) return s->allph ? copy_out(s->allph, s->log_n * 10, out, n) : 0;
    if (!strcmp(k, "line_multiplet")) {                     /* type, nup, ndown[3], f12[3], delE_Hz[3], damping_up[3], then [iup][idown]: A21, Elow_Hz, E1, E2, E3, P_down, A_down */
        double v[98]; int q = 0;
        v[q++] = s->line_type; v[q++] = s->ml.nup;
        for (int i = 0; i < 3; i++) v[q++] = s->ml.ndown[i];
        for (int i = 0; i < 3; i++) v[q++] = s->ml.f12[i];
        for (int i = 0; i < 3; i++) v[q++] = s->ml.delE_Hz[i];
        for (int i = 0; i < 3; i++) v[q++] = s->ml.damping_up[i];
        for (int i = 0; i < 3; i++) for (int d = 0; d < 4; d++) { v[q++] = s->ml.A21[i][d]; v[q++] = s->ml.Elow_Hz[i][d]; v[q++] = s->ml.E1[i][d]; v[q++] = s->ml.E2[i][d];
            v[q++] = s->ml.E3[i][d]; v[q++] = s->ml.P_down[i][d]; v[q++] = s->ml.A_down[i][d]; }
        return copy_out(v, 98, out, n); }
    if (!strcmp(k, "Jout_Ha")) return copy_out(s->Jout_Ha, s->nxfreq_Ha, out, n);
    if (!strcmp(k, "Jabs_Ha")) return copy_out(s->Jabs_Ha, s->nxfreq_Ha, out, n);
    if (!strcmp(k, "Pconv")) return copy_out(s->Pconv, s->nbins, out, n);
    if (!strcmp(k, "W")) return copy_out(s->W, 5, out, n);                       /* W_conv, W_esc1, W_abs1, W_esc2, W_abs2 */
    if (!strcmp(k, "W_H2")) return copy_out(s->W_H2, 2 + s->h2_nlines, out, n);   /* W_H2abs, W_H2scat, W_H2pump(:) */
    if (!strcmp(k, "h2_lines")) { double v[26] = {0}; v[0] = s->h2_nlines; v[1] = s->h2_Dfreq_Hz;
        for (int i = 0; i < s->h2_nlines; i++) { v[2+i*6] = s->h2_dnu_Hz[i]; v[3+i*6] = s->h2_strength[i]; v[4+i*6] = s->h2_a_damp[i]; v[5+i*6] = s->h2_p_scat[i]; v[6+i*6] = s->h2_pop[i]; v[7+i*6] = s->h2_f_osc[i]; }
        return copy_out(v, 26, out, n); }
    if (!strcmp(k, "Jout")) return copy_out(s->Jout, s->nxfreq, out, n);
    if (!strcmp(k, "Jin")) return copy_out(s->Jin, s->nxfreq, out, n);
    if (!strcmp(k, "Jabs")) return copy_out(s->Jabs, s->nxfreq, out, n);
    if (!strcmp(k, "Jmu")) return copy_out(s->Jmu, (int64_t)s->nxfreq * s->nmu, out, n);
    if (!strcmp(k, "rhokap")) return copy_out(s->rhokap, nc, out, n);
    if (!strcmp(k, "voigt_a")) return copy_out(s->voigt_a, nc, out, n);
    if (!strcmp(k, "Dfreq")) return copy_out(s->Dfreq, nc, out, n);
    if (!strcmp(k, "vfx")) return copy_out(s->vfx, nc, out, n);
    if (!strcmp(k, "vfy")) return copy_out(s->vfy, nc, out, n);
    if (!strcmp(k, "vfz")) return copy_out(s->vfz, nc, out, n);
    if (!strcmp(k, "rhokapD")) return copy_out(s->rhokapD, nc, out, n);
    if (!strcmp(k, "xface")) return copy_out(s->xface, s->nx + 1, out, n);
    if (!strcmp(k, "yface")) return copy_out(s->yface, s->ny + 1, out, n);
    if (!strcmp(k, "zface")) return copy_out(s->zface, s->nz + 1, out, n);
    if (s->use_clump_medium) { const clumps_t *c = &s->cl; const double *da = NULL;
        if (!strcmp(k, "cl_x")) da = c->x; else if (!strcmp(k, "cl_y")) da = c->y; else if (!strcmp(k, "cl_z")) da = c->z;
        else if (!strcmp(k, "cl_vx")) da = c->vx; else if (!strcmp(k, "cl_vy")) da = c->vy; else if (!strcmp(k, "cl_vz")) da = c->vz;
        else if (!strcmp(k, "cl_radius")) da = c->radius; else if (!strcmp(k, "cl_rhokap")) da = c->rhokap; else if (!strcmp(k, "cl_rhokapD")) da = c->rhokapD;
        else if (!strcmp(k, "cl_voigt_a")) da = c->voigt_a; else if (!strcmp(k, "cl_Dfreq")) da = c->Dfreq;
        if (da) return copy_out(da, c->N, out, n);
        if (!strcmp(k, "cg_start")) { int64_t cnt = (int64_t)c->cgx * c->cgx * c->cgx + 1; for (int64_t i = 0; i < cnt && i < n; i++) out[i] = c->cg_start[i]; return cnt; }
        if (!strcmp(k, "cg_list")) { for (int64_t i = 0; i < c->nreg && i < n; i++) out[i] = c->cg_list[i]; return c->nreg; } }
    if (s->use_amr_grid) { const amr_t *a = &s->amr;
        if (!strcmp(k, "amr_cx")) return copy_out(a->cx, a->ncells, out, n); if (!strcmp(k, "amr_cy")) return copy_out(a->cy, a->ncells, out, n);
        if (!strcmp(k, "amr_cz")) return copy_out(a->cz, a->ncells, out, n); if (!strcmp(k, "amr_ch")) return copy_out(a->ch, a->ncells, out, n);
        const int *ia = NULL; int64_t cnt = 0;
        if (!strcmp(k, "amr_children")) { ia = a->children; cnt = 8 * (int64_t)a->ncells; } else if (!strcmp(k, "amr_neighbor")) { ia = a->neighbor; cnt = 6 * (int64_t)a->ncells; }
        else if (!strcmp(k, "amr_ileaf")) { ia = a->ileaf; cnt = a->ncells; } else if (!strcmp(k, "amr_icell_of_leaf")) { ia = a->icell_of_leaf; cnt = a->nleaf; }
        else if (!strcmp(k, "amr_level")) { ia = a->level; cnt = a->ncells; } else if (!strcmp(k, "amr_parent")) { ia = a->parent; cnt = a->ncells; }
        if (ia) { for (int64_t i = 0; i < cnt && i < n; i++) out[i] = ia[i]; return cnt; } }
    if (!strcmp(k, "J1")) return copy_out(s->J1, (int64_t)s->nxfreq * s->nbins, out, n);
    if (!strcmp(k, "P1")) return copy_out(s->P1, s->nbins, out, n);
    if (!strcmp(k, "P1_new")) return copy_out(s->P1_new, s->nbins, out, n);
    if (!strcmp(k, "ncount")) return copy_out(s->ncount, s->nbins, out, n);
    if (!strcmp(k, "ind_bin")) { if (!s->ind_bin) return -1; for (int64_t i = 0; i < nc && i < n; i++) out[i] = s->ind_bin[i]; return nc; }
    if (!strcmp(k, "coss")) return copy_out(s->coss, s->nPDF, out, n);
    if (!strcmp(k, "S11")) return copy_out(s->S11, s->nPDF, out, n);
    if (!strcmp(k, "S12")) return copy_out(s->S12, s->nPDF, out, n);
    if (!strcmp(k, "S33")) return copy_out(s->S33, s->nPDF, out, n);
    if (!strcmp(k, "S34")) return copy_out(s->S34, s->nPDF, out, n);
    if (!strcmp(k, "phase_PDF")) return copy_out(s->phase_PDF, s->nPDF - 1, out, n);
    if (!strcmp(k, "alias")) { if (!s->alias) return -1; for (int i = 0; i < s->nPDF - 1 && i < n; i++) out[i] = s->alias[i]; return s->nPDF - 1; }
    if (!strcmp(k, "obs_xyz")) { for (int i = 0; i < s->nobs && 3 * i + 2 < n; i++) { out[3*i] = s->obs[i].x; out[3*i+1] = s->obs[i].y; out[3*i+2] = s->obs[i].z; } return 3 * s->nobs; }
    if (!strcmp(k, "rmatrix")) { for (int i = 0; i < s->nobs; i++) for (int q = 0; q < 9 && 9 * i + q < n; q++) out[9*i+q] = s->obs[i].rm[q]; return 9 * s->nobs; }
    const char *col = strchr(k, ':');
    if (col) { int io = atoi(col + 1); if (io < 0 || io >= s->nobs) return -1; const obs_t *o = &s->obs[io];
        int64_t n2 = (int64_t)s->nxim * s->nyim, n3 = n2 * s->nxfreq; size_t L = col - k;
#define M(name, ptr, cnt) if (L == strlen(name) && !strncmp(k, name, L)) return copy_out(ptr, cnt, out, n);
        M("peel_Ha", o->peel_Ha, n2 * s->nxfreq_Ha)
        M("direc0_2D", o->direc0_2D, n2) M("direc0", o->direc0, n3) M("scatt_dust", o->scatt_dust, n3)
        M("scatt", o->scatt, n3) M("direc", o->direc, n3) M("I", o->I, n3) M("Q", o->Q, n3) M("U", o->U, n3) M("V", o->V, n3)
        M("scatt_2D", o->scatt_2D, n2) M("direc_2D", o->direc_2D, n2) M("I_2D", o->I_2D, n2) M("Q_2D", o->Q_2D, n2) M("U_2D", o->U_2D, n2) M("V_2D", o->V_2D, n2)
#undef M
    }
    return -1;
}

/* =====================================================================================
 * Function-level entry points
 * ===================================================================================== */
static mt_t g_mt; static rng_t g_rng_mt;
static rng_t *pick_rng(rng_t *tmp, int kind, uint64_t seed, uint64_t stream) {
    if (kind == 0) { if (seed != UINT64_MAX) { memset(&g_rng_mt, 0, sizeof g_rng_mt); g_rng_mt.mt = &g_mt; mt_init(&g_mt, (int64_t)(int32_t)seed); } return &g_rng_mt; }
    rng_philox(tmp, seed, stream, 0); return tmp;
}
void orc_rand_number_v(int kind, uint64_t seed, uint64_t id, int n, double *u) { rng_t t, *g = pick_rng(&t, kind, seed, id); for (int i = 0; i < n; i++) u[i] = rnd(g); }
void orc_rand_resonance_vz_v(int kind, uint64_t seed, uint64_t id, int n, double x, double a, double *vz) { rng_t t, *g = pick_rng(&t, kind, seed, id); for (int i = 0; i < n; i++) vz[i] = rand_resonance_vz(g, x, a); }
void orc_rand_resonance_v(int kind, uint64_t seed, uint64_t id, int n, double E1, double *c) { rng_t t, *g = pick_rng(&t, kind, seed, id); for (int i = 0; i < n; i++) c[i] = rand_resonance(g, E1); }
void orc_rand_hg_v(int kind, uint64_t seed, uint64_t id, int n, double gg, double *c) { rng_t t, *g = pick_rng(&t, kind, seed, id); for (int i = 0; i < n; i++) c[i] = rand_hg(g, gg); }
void orc_rand_voigt_v(int kind, uint64_t seed, uint64_t id, int n, double a, double *c) { rng_t t, *g = pick_rng(&t, kind, seed, id); for (int i = 0; i < n; i++) c[i] = rand_voigt(g, a); }
void orc_rand_gauss_v(int kind, uint64_t seed, uint64_t id, int n, double *c) { rng_t t, *g = pick_rng(&t, kind, seed, id); for (int i = 0; i < n; i++) c[i] = rand_gauss(g); }

/* do_resonance1 / do_resonance2 in cell (ic,jc,kc) (clump mode: ic = owner clump): n draws on one stream;
 * out[12*i + ..] = uz, xfreq_atom, cost, sint, S11, S22, S12, S33, S44, E1, E2, E3 (line type 7: E3 slot <- species 1 H / 2 D) */
void orc_do_resonance_v(const orc_sim *s, int kind, uint64_t seed, uint64_t id, int n, double xfreq, int ic, int jc, int kc, double *out) {
    rng_t t, *g = pick_rng(&t, kind, seed, id);
    for (int i = 0; i < n; i++) {
        photon_t p; memset(&p, 0, sizeof p); p.xfreq = xfreq; p.icell = ic; p.jcell = jc; p.kcell = kc; p.icell_clump = ic;
        if (s->use_clump_medium) { p.icell = p.jcell = p.kcell = 1; }
        double *o = out + 12 * (size_t)i;
        do_resonance1(s, g, &p, &o[0], &o[1], &o[2], &o[3], &o[4], &o[5], &o[6], &o[7], &o[8]);
        o[9] = p.E1; o[10] = p.E2; o[11] = (s->line_type == 7) ? (double)p.species : p.E3;
        if (s->line_type == 8) o[4] = (double)(p.iband == 2 ? 2 : 1);              /* do_resonance8: the band after the event (no Stokes matrix with ly_beta) */
    }
}
void orc_calc_voigt_v(const orc_sim *s, int n, const double *x, int ic, int jc, int kc, double *h) {
    size_t c = s->use_amr_grid ? (size_t)(ic - 1) : IDX(s, ic, jc, kc);
    for (int i = 0; i < n; i++) h[i] = calc_voigt(s, x[i], s->voigt_a[c], s->Dfreq[c]);
}
static void unpack(const double r[8], const int ic[4], photon_t *p) {
    memset(p, 0, sizeof *p); p->x = r[0]; p->y = r[1]; p->z = r[2]; p->kx = r[3]; p->ky = r[4]; p->kz = r[5]; p->xfreq = r[6]; p->wgt = r[7];
    p->icell = ic[0]; p->jcell = ic[1]; p->kcell = ic[2]; p->inside = ic[3];
}
/* clump mode: ic[0] carries photon%icell_clump (0 = vacuum) in and out */
static void unpack_s(const orc_sim *s, const double r[8], const int ic[4], photon_t *p) {
    unpack(r, ic, p);
    if (s->use_clump_medium) { p->icell_clump = ic[0]; p->icell = p->jcell = p->kcell = 1; }
}
/* the shearing-box march (variant 8) with photon%vfy_shear in / out */
void orc_raytrace_to_tau_shear(orc_sim *s, double r[8], int ic[4], double tau_in, double *shear) {
    photon_t p; unpack_s(s, r, ic, &p); acc_t A; acc_alloc(&A, s);
    p.vfy_shear = *shear;
    raytrace_to_tau(s, &A, &p, tau_in, 8);
    acc_merge_free(&A, s);
    *shear = p.vfy_shear;
    r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = p.kx; r[4] = p.ky; r[5] = p.kz; r[6] = p.xfreq; r[7] = p.wgt;
    ic[0] = p.icell; ic[1] = p.jcell; ic[2] = p.kcell; ic[3] = p.inside;
}
void orc_raytrace_to_tau(orc_sim *s, int variant, double r[8], int ic[4], double tau_in) {
    photon_t p; unpack_s(s, r, ic, &p); acc_t A; acc_alloc(&A, s);
    rng_t g; memset(&g, 0, sizeof g); rng_philox(&g, (uint64_t)s->iseed, (uint64_t)(r[7] > 0.0 && variant == 7 ? (int64_t)r[7] : 1), 0); A.g = &g;   /* variant 7 draws the owner clump: stream (iseed, id = wgt slot) */
    raytrace_to_tau(s, &A, &p, tau_in, variant);
    acc_merge_free(&A, s);
    r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = p.kx; r[4] = p.ky; r[5] = p.kz; r[6] = p.xfreq; r[7] = p.wgt;
    ic[0] = s->use_clump_medium ? p.icell_clump : p.icell; ic[1] = p.jcell; ic[2] = p.kcell; ic[3] = p.inside;
}
/* probes of the ly_beta / H2 pieces: the tau march of a packet of band `iband` (1 or 2; r[6] = xfreq, xfreq_ref once outside), the dust-only band-2
 * sight line, h2_kappa_D and h2_select_line_D (MT stream kind 0 / Philox kind 1) */
void orc_raytrace_to_tau_band(orc_sim *s, int iband, double r[8], int ic[4], double tau_in) {
    photon_t p; unpack_s(s, r, ic, &p); p.iband = iband; acc_t A; acc_alloc(&A, s);
    raytrace_to_tau(s, &A, &p, tau_in, s->raytrace_variant);
    acc_merge_free(&A, s);
    r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = p.kx; r[4] = p.ky; r[5] = p.kz; r[6] = p.inside ? p.xfreq : p.xfreq_ref; r[7] = p.wgt;
    ic[0] = p.icell; ic[1] = p.jcell; ic[2] = p.kcell; ic[3] = p.inside;
}
void orc_raytrace_to_edge_dust_Ha(const orc_sim *s, const double r[8], const int ic[4], double *tau) { photon_t p; unpack(r, ic, &p); p.iband = 2; *tau = raytrace_to_edge_dust_Ha(s, &p, 0); }
void orc_h2_kappa_v(const orc_sim *s, int n, const double *x, double Dfreq, double *kap) { for (int i = 0; i < n; i++) kap[i] = h2_kappa_D(s, x[i], Dfreq); }
void orc_h2_select_v(const orc_sim *s, int kind, uint64_t seed, uint64_t id, int n, const double *x, double Dfreq, int *il) {
    rng_t t, *g = pick_rng(&t, kind, seed, id); for (int i = 0; i < n; i++) il[i] = h2_select_line_D(s, g, x[i], Dfreq); }
int orc_amr_find_leaf(const orc_sim *s, double x, double y, double z) { return s->use_amr_grid ? amr_find_leaf(&s->amr, x, y, z) : 0; }
void orc_raytrace_to_edge(const orc_sim *s, int variant, const double r[8], const int ic[4], double *tau) {
    photon_t p; unpack_s(s, r, ic, &p);
    if (variant == 14) *tau = raytrace_to_edge_clump(s, &p, TAU_HUGE, 0);         /* the capped peel-off form */
    else if (variant == 17) *tau = raytrace_to_edge_clump_overlap(s, &p, TAU_HUGE, 0);
    else *tau = raytrace_to_edge(s, &p, variant, -1.0, 0);
}
void orc_raytrace_to_dist(const orc_sim *s, const double r[8], const int ic[4], double dist, double *tau) { photon_t p; unpack(r, ic, &p); *tau = raytrace_to_edge(s, &p, s->use_amr_grid ? 3 : 0, dist, 0); }

/* =====================================================================================
 * Sight-line maps (par%save_sightline_tau): sightline_tau_rect.f90, sightline_tau_heal.f90, sightline_tau_clump.f90
 * tau_gas(nu) and N(HI) from the far side of the medium towards each observer pixel.  Gas only, and no tau_huge cut
 * ("sightline tau output must be the full integrated value", raytrace_amr.f90:424-425).
 * ===================================================================================== */
/* raytrace_to_edge_car_tau_gas: raytrace_car.f90:1236-1327; dist >= 0: raytrace_to_dist_tau_gas_car :197-297 */
static double tau_gas_car(const orc_sim *s, const photon_t *p, double dist) {
    trav_t t; double tau = 0.0, d = 0.0;
    t.kx = p->kx; t.ky = p->ky; t.kz = p->kz; t.icell = p->icell; t.jcell = p->jcell; t.kcell = p->kcell;
    if (setup_traversal_car(s, p->x, p->y, p->z, &t)) return tau;
    int iold = t.icell, jold = t.jcell, kold = t.kcell;
    double u1 = udot(s, IDX(s,iold,jold,kold), t.kx, t.ky, t.kz), u2, xfreq = p->xfreq;
    for (;;) {
        size_t c = IDX(s,t.icell,t.jcell,t.kcell);
        double rhokap = s->rhokap[c] * calc_voigt(s, xfreq, s->voigt_a[c], s->Dfreq[c]);
        int idx = minloc3(t.tx, t.ty, t.tz);
        double *tt = (idx == 1) ? &t.tx : (idx == 2) ? &t.ty : &t.tz;
        if (dist >= 0.0 && *tt >= dist) { tau = tau + (dist - d) * rhokap; break; }
        tau = tau + (*tt - d) * rhokap; d = *tt;
        if (idx == 1)      { t.icell += t.istep; if (t.icell < 1 || t.icell > s->nx) break; t.tx += t.delx; }
        else if (idx == 2) { t.jcell += t.jstep; if (t.jcell < 1 || t.jcell > s->ny) break; t.ty += t.dely; }
        else               { t.kcell += t.kstep; if (t.kcell < 1 || t.kcell > s->nz) break; t.tz += t.delz; }
        size_t cn = IDX(s,t.icell,t.jcell,t.kcell), co = IDX(s,iold,jold,kold);
        u2 = udot(s, cn, t.kx, t.ky, t.kz);
        xfreq = (xfreq + u1) * s->Dfreq[co] / s->Dfreq[cn] - u2;
        iold = t.icell; jold = t.jcell; kold = t.kcell; u1 = u2;
    }
    return tau;
}
/* raytrace_to_edge_car_column: raytrace_car.f90:1329-1423; dist >= 0: raytrace_to_dist_column_car :299-408 */
static void column_car(const orc_sim *s, const photon_t *p, double dist, double *N_gas, double *tau_dust) {
    trav_t t; double d = 0.0; *N_gas = 0.0; *tau_dust = 0.0;
    t.kx = p->kx; t.ky = p->ky; t.kz = p->kz; t.icell = p->icell; t.jcell = p->jcell; t.kcell = p->kcell;
    if (setup_traversal_car(s, p->x, p->y, p->z, &t)) return;
    for (;;) {
        size_t c = IDX(s,t.icell,t.jcell,t.kcell);
        double rho = s->rhokap[c] * s->Dfreq[c] / s->cross0, rhokapD = (s->DGR > 0.0) ? s->rhokapD[c] : 0.0, del;
        int idx = minloc3(t.tx, t.ty, t.tz), last = 0;
        double *tt = (idx == 1) ? &t.tx : (idx == 2) ? &t.ty : &t.tz;
        if (dist >= 0.0 && *tt >= dist) { del = dist - d; last = 1; } else del = *tt - d;
        *N_gas = *N_gas + del * rho;
        if (s->DGR > 0.0) *tau_dust = *tau_dust + del * rhokapD;
        if (last) break;
        d = *tt;
        if (idx == 1)      { t.icell += t.istep; if (t.icell < 1 || t.icell > s->nx) break; t.tx += t.delx; }
        else if (idx == 2) { t.jcell += t.jstep; if (t.jcell < 1 || t.jcell > s->ny) break; t.ty += t.dely; }
        else               { t.kcell += t.kstep; if (t.kcell < 1 || t.kcell > s->nz) break; t.tz += t.delz; }
    }
}
/* raytrace_to_edge_tau_gas_amr: raytrace_amr.f90:395-472; dist >= 0: raytrace_to_dist_tau_gas_amr :650-732.
 * want_column: raytrace_to_{edge,dist}_column_amr :477-554 / :737-823 on the same walk (the frequency plays no part in it) */
static double sight_amr(const orc_sim *s, const photon_t *p, double dist, int want_column, double *N_gas, double *tau_dust) {
    const amr_t *a = &s->amr;
    double x = p->x, y = p->y, z = p->z, kx = p->kx, ky = p->ky, kz = p->kz, tau = 0.0, d = 0.0, xfreq = p->xfreq;
    int il = p->icell;
    if (want_column) { *N_gas = 0.0; *tau_dust = 0.0; }
    if (il <= 0) { il = amr_find_leaf(a, x, y, z); if (il <= 0) return tau; }
    double u1 = s->vfx[il-1] * kx + s->vfy[il-1] * ky + s->vfz[il-1] * kz, u2;
    for (;;) {
        int ic = a->icell_of_leaf[il-1], iface, last = 0; double t_exit, del;
        amr_cell_exit(a, ic, x, y, z, kx, ky, kz, &t_exit, &iface);
        if (dist >= 0.0 && d + t_exit >= dist) { del = dist - d; last = 1; } else del = t_exit;
        if (want_column) {
            *N_gas = *N_gas + del * s->rhokap[il-1] * s->Dfreq[il-1] / s->cross0;
            if (s->DGR > 0.0) *tau_dust = *tau_dust + del * s->rhokapD[il-1];
        } else tau = tau + del * (s->rhokap[il-1] * calc_voigt(s, xfreq, s->voigt_a[il-1], s->Dfreq[il-1]));
        if (last) return tau;
        d = d + t_exit; x = x + t_exit * kx; y = y + t_exit * ky; z = z + t_exit * kz;
        int il_new = amr_next_leaf(a, ic, iface, x, y, z), reflected = 0;
        if (il_new <= 0) il_new = amr_boundary(s, il, iface, &x, &y, z, &kx, &ky, &kz, &reflected);
        if (il_new <= 0) return tau;
        u2 = s->vfx[il_new-1] * kx + s->vfy[il_new-1] * ky + s->vfz[il_new-1] * kz;
        if (reflected) xfreq = (xfreq + u1) - u2; else xfreq = (xfreq + u1) * s->Dfreq[il-1] / s->Dfreq[il_new-1] - u2;
        u1 = u2; il = il_new;
    }
}
/* raytrace_to_edge_tau_gas_clump (raytrace_clump.f90:274-285: raytrace_to_edge_clump with clump_gas_only) and
 * raytrace_to_edge_column_clump (:289-334) on the same walk */
/* raytrace_to_edge_tau_gas_clump_overlap (:922-931: the uncapped overlap march with clump_gas_only) and raytrace_to_edge_column_clump_overlap (:935-1001) */
static double sight_clump_overlap(const orc_sim *s, const photon_t *p0, int want_column, double *N_gas, double *tau_dust) {
    const clumps_t *c = &s->cl;
    static __thread double ev_t[OVL_MAX_EVT]; static __thread int64_t ev_icl[OVL_MAX_EVT], active[OVL_MAX_EVT]; static __thread int ev_type[OVL_MAX_EVT];
    double tau = 0.0, kx = p0->kx, ky = p0->ky, kz = p0->kz, xfreq_g = p0->xfreq;
    if (want_column) { *N_gas = 0.0; *tau_dust = 0.0; }
    double t_sp = sphere_exit_dist(c, p0->x, p0->y, p0->z, kx, ky, kz);
    if (t_sp <= 0.0) return tau;
    int n_ev = collect_ray_events_overlap(c, p0->x, p0->y, p0->z, kx, ky, kz, t_sp, ev_t, ev_icl, ev_type, NULL);
    int n_active = active_set_at_point(c, p0->x, p0->y, p0->z, active);
    double t_cur = 0.0;
    for (int ie = 1; ie <= n_ev + 1; ie++) {
        double t_next = ie <= n_ev ? ev_t[ie-1] : t_sp; t_next = fmin(t_next, t_sp);
        double dt = t_next - t_cur;
        if (dt > 0.0) {
            if (want_column) for (int m = 0; m < n_active; m++) { int64_t icl = active[m];
                *N_gas = *N_gas + (c->rhokap[icl-1] / s->cross0) * c->Dfreq[icl-1] * dt;
                if (s->DGR > 0.0 && c->rhokapD) *tau_dust = *tau_dust + c->rhokapD[icl-1] * dt; }
            else { double kap = 0.0;
                for (int m = 0; m < n_active; m++) { int64_t icl = active[m]; kap = kap + c->rhokap[icl-1] * voigt_clump(s, xfreq_g - ulos_clump(c, icl, kx, ky, kz), icl); }
                tau = tau + kap * dt; }
        }
        t_cur = t_next;
        if (ie <= n_ev) ovl_apply_event(ev_icl[ie-1], ev_type[ie-1], active, &n_active);
    }
    return tau;
}
static double sight_clump(const orc_sim *s, const photon_t *p0, int want_column, double *N_gas, double *tau_dust) {
    const clumps_t *c = &s->cl;
    if (c->has_overlap) return sight_clump_overlap(s, p0, want_column, N_gas, tau_dust);
    double tau = 0.0, xp = p0->x, yp = p0->y, zp = p0->z, kx = p0->kx, ky = p0->ky, kz = p0->kz, xf = p0->xfreq;
    int64_t icl_cur = p0->icell_clump, icl_found;
    if (want_column) { *N_gas = 0.0; *tau_dust = 0.0; }
    if (icl_cur > 0) {
        double t_seg = clump_exit_dist(c, xp, yp, zp, kx, ky, kz, icl_cur);
        if (want_column) {
            *N_gas = *N_gas + (c->rhokap[icl_cur-1] / s->cross0) * c->Dfreq[icl_cur-1] * t_seg;
            if (s->DGR > 0.0 && c->rhokapD) *tau_dust = *tau_dust + c->rhokapD[icl_cur-1] * t_seg;
        } else tau = tau + c->rhokap[icl_cur-1] * voigt_clump(s, xf, icl_cur) * t_seg;
        xp = xp + t_seg * kx; yp = yp + t_seg * ky; zp = zp + t_seg * kz;
        xf = xf + ulos_clump(c, icl_cur, kx, ky, kz);
        if (xp * xp + yp * yp + zp * zp >= c->sphere_R * c->sphere_R) return tau;
    }
    for (;;) {
        double t_sp = sphere_exit_dist(c, xp, yp, zp, kx, ky, kz), te, tx2;
        if (t_sp <= 0.0) break;
        if (!find_next_clump(c, xp, yp, zp, kx, ky, kz, icl_cur, t_sp, &te, &tx2, &icl_found, 0)) break;
        te = fmax(0.0, te);
        xp = xp + te * kx; yp = yp + te * ky; zp = zp + te * kz;
        double u_los = ulos_clump(c, icl_found, kx, ky, kz);
        xf = xf - u_los;
        double t_seg = clump_exit_dist(c, xp, yp, zp, kx, ky, kz, icl_found);
        if (want_column) {
            *N_gas = *N_gas + (c->rhokap[icl_found-1] / s->cross0) * c->Dfreq[icl_found-1] * t_seg;
            if (s->DGR > 0.0 && c->rhokapD) *tau_dust = *tau_dust + c->rhokapD[icl_found-1] * t_seg;
        } else tau = tau + c->rhokap[icl_found-1] * voigt_clump(s, xf, icl_found) * t_seg;
        xp = xp + t_seg * kx; yp = yp + t_seg * ky; zp = zp + t_seg * kz;
        xf = xf + u_los;
        icl_cur = icl_found;
        if (xp * xp + yp * yp + zp * zp >= c->sphere_R * c->sphere_R) break;
    }
    return tau;
}
/* pix2vec: healpix.f90:114-187 (RING scheme, ipix = 1..npix) */
void orc_pix2vec(int nside, int ipix, double v[3]) {
    int nl2 = 2 * nside, nl4 = 4 * nside, npix = 12 * nside * nside, ncap = 2 * nside * (nside - 1), iring, iphi, ip;
    double fact1 = 1.5 * nside, fact2 = 3.0 * nside * nside, z, phi, hip, fihip, fodd;
    if (ipix <= ncap) {
        hip = ipix / 2.0; fihip = trunc(hip);
        iring = (int)sqrt(fmax(0.0, hip - sqrt(fihip))) + 1;
        iphi = ipix - 2 * iring * (iring - 1);
        z = 1.0 - (double)(iring * iring) / fact2;
        phi = ((double)iphi - 0.5) * PI / (2.0 * iring);
    } else if (ipix <= nl2 * (5 * nside + 1)) {
        ip = ipix - ncap - 1;
        iring = ip / nl4 + nside;
        iphi = ip % nl4 + 1;
        fodd = 0.5 * (1 + (iring + nside) % 2);
        z = (double)(nl2 - iring) / fact1;
        phi = ((double)iphi - fodd) * PI / (2.0 * nside);
    } else {
        ip = npix - ipix + 1;
        hip = ip / 2.0; fihip = trunc(hip);
        iring = (int)sqrt(fmax(0.0, hip - sqrt(fihip))) + 1;
        iphi = 4 * iring + 1 - (ip - 2 * iring * (iring - 1));
        z = -1.0 + (double)(iring * iring) / fact2;
        phi = ((double)iphi - 0.5) * PI / (2.0 * iring);
    }
    double sth = sqrt((1.0 - z) * (1.0 + z));
    v[0] = sth * cos(phi); v[1] = sth * sin(phi); v[2] = z;
}

/* the start of one sight line: position on the far boundary, direction towards the observer, start cell.
 * Returns 0 when the pixel sees no medium.  *dist = march length (inside observers), -1 = to the edge. */
static int sight_start(const orc_sim *s, int io, int ix, int iy, photon_t *p, double *dist) {
    const obs_t *o = &s->obs[io];
    double delt[6], kx, ky, kz;
    memset(p, 0, sizeof *p); *dist = -1.0;
    if (s->nside > 0) { double v[3]; orc_pix2vec(s->nside, ix, v); p->kx = v[0]; p->ky = v[1]; p->kz = v[2]; }
    else {                                                                                  /* sightline_tau_rect.f90:44-54 */
        kx = tan((ix - (s->nxim + 1.0) / 2.0) * s->dxim / RAD2DEG); ky = tan((iy - (s->nyim + 1.0) / 2.0) * s->dyim / RAD2DEG); kz = -1.0;
        double kr = sqrt(kx * kx + ky * ky + kz * kz); kx = kx / kr; ky = ky / kr; kz = kz / kr;
        const double *m = o->rm;
        p->kx = m[0] * kx + m[3] * ky + m[6] * kz; p->ky = m[1] * kx + m[4] * ky + m[7] * kz; p->kz = m[2] * kx + m[5] * ky + m[8] * kz;
    }
    if (s->use_clump_medium) {                                                              /* sightline_tau_clump.f90:78-116 */
        if (s->nside > 0) return 0;
        double R = s->cl.sphere_R, eps = 1.0e-9 * R;
        double b = o->x * p->kx + o->y * p->ky + o->z * p->kz, c = o->x * o->x + o->y * o->y + o->z * o->z - R * R, disc = b * b - c;
        if (disc < 0.0) return 0;
        double t_far = -b + sqrt(disc);
        if (t_far <= 0.0) return 0;
        p->x = o->x + p->kx * t_far; p->y = o->y + p->ky * t_far; p->z = o->z + p->kz * t_far;
        p->kx = -p->kx; p->ky = -p->ky; p->kz = -p->kz;
        p->x = p->x + eps * p->kx; p->y = p->y + eps * p->ky; p->z = p->z + eps * p->kz;
        p->icell_clump = 0; p->icell = p->jcell = p->kcell = 1;
        return 1;
    }
    double lo[3], hi[3];
    if (s->use_amr_grid) { lo[0] = s->amr.xmin; lo[1] = s->amr.ymin; lo[2] = s->amr.zmin; hi[0] = s->amr.xmax; hi[1] = s->amr.ymax; hi[2] = s->amr.zmax; }
    else { lo[0] = s->xmin; lo[1] = s->ymin; lo[2] = s->zmin; hi[0] = s->xmax; hi[1] = s->ymax; hi[2] = s->zmax; }
    const double none = (s->nside > 0) ? -999.0 : HUGEST, ob[3] = {o->x, o->y, o->z}, kk[3] = {p->kx, p->ky, p->kz};
    for (int a = 0; a < 3; a++) {
        if (kk[a] == 0.0) { delt[2*a] = none; delt[2*a+1] = none; }
        else { delt[2*a] = (hi[a] - ob[a]) / kk[a]; delt[2*a+1] = (lo[a] - ob[a]) / kk[a]; }
    }
    int j0 = 0; double dd;
    if (s->nside > 0) {                                                                     /* nearest positive face: sightline_tau_heal.f90:70-79 */
        dd = HUGEST;
        for (int jj = 1; jj <= 6; jj++) if (delt[jj-1] > 0.0 && delt[jj-1] < HUGEST && delt[jj-1] < dd) { dd = delt[jj-1]; j0 = jj; }
        if (!(dd > 0.0 && dd < HUGEST)) return 0;
    } else if (s->use_amr_grid) {                                                           /* farthest face point inside the box: sightline_tau_rect.f90:259-277 */
        double eps = 1.0e-9 * s->amr.L_box; dd = -999.9;
        for (int jj = 1; jj <= 6; jj++) if (delt[jj-1] > 0.0 && delt[jj-1] < HUGEST) {
            double x = o->x + p->kx * delt[jj-1], y = o->y + p->ky * delt[jj-1], z = o->z + p->kz * delt[jj-1];
            if (x >= lo[0] - eps && x <= hi[0] + eps && y >= lo[1] - eps && y <= hi[1] + eps && z >= lo[2] - eps && z <= hi[2] + eps)
                if (delt[jj-1] > dd) { dd = delt[jj-1]; j0 = jj; }
        }
        if (dd <= 0.0 || dd >= HUGEST || j0 == 0) return 0;
    } else {                                                                                /* :79-107 */
        dd = -999.9;
        for (int jj = 1; jj <= 6; jj++) if (delt[jj-1] > 0.0 && delt[jj-1] < HUGEST) {
            double x = o->x + p->kx * delt[jj-1], y = o->y + p->ky * delt[jj-1], z = o->z + p->kz * delt[jj-1];
            int ic = (int)floor((x - s->xmin) / s->dx) + 1, jc = (int)floor((y - s->ymin) / s->dy) + 1, kc = (int)floor((z - s->zmin) / s->dz) + 1;
            if (jj == 1) ic = s->nx + 1; if (jj == 2) ic = 1; if (jj == 3) jc = s->ny + 1; if (jj == 4) jc = 1; if (jj == 5) kc = s->nz + 1; if (jj == 6) kc = 1;
            if (ic >= 1 && ic <= s->nx + 1 && jc >= 1 && jc <= s->ny + 1 && kc >= 1 && kc <= s->nz + 1)
                if (delt[jj-1] > dd) { dd = delt[jj-1]; j0 = jj; }
        }
        if (!(dd > 0.0 && dd < HUGEST)) return 0;
    }
    p->x = o->x + p->kx * dd; p->y = o->y + p->ky * dd; p->z = o->z + p->kz * dd;
    if (s->nside > 0) *dist = dd;
    if (s->use_amr_grid) {
        if (s->nside > 0) {                                                                 /* sightline_tau_heal.f90:230-251 (the tiny() nudge is below one ulp) */
            p->kx = -p->kx; p->ky = -p->ky; p->kz = -p->kz;
            if (j0 == 1) p->x = hi[0]; else if (j0 == 2) p->x = lo[0]; else if (j0 == 3) p->y = hi[1]; else if (j0 == 4) p->y = lo[1]; else if (j0 == 5) p->z = hi[2]; else p->z = lo[2];
        } else {                                                                            /* sightline_tau_rect.f90:281-297 */
            double eps = 1.0e-9 * s->amr.L_box;
            p->x = fmax(lo[0], fmin(hi[0], p->x)); p->y = fmax(lo[1], fmin(hi[1], p->y)); p->z = fmax(lo[2], fmin(hi[2], p->z));
            p->kx = -p->kx; p->ky = -p->ky; p->kz = -p->kz;
            p->x = p->x + eps * p->kx; p->y = p->y + eps * p->ky; p->z = p->z + eps * p->kz;
        }
        p->icell = amr_find_leaf(&s->amr, p->x, p->y, p->z); p->jcell = p->kcell = 1;
        return p->icell > 0;
    }
    p->icell = (int)floor((p->x - s->xmin) / s->dx) + 1; p->jcell = (int)floor((p->y - s->ymin) / s->dy) + 1; p->kcell = (int)floor((p->z - s->zmin) / s->dz) + 1;
    p->kx = -p->kx; p->ky = -p->ky; p->kz = -p->kz;                                         /* :117-142 */
    if (j0 == 1) { p->icell = s->nx + 1; p->x = s->xface[s->nx]; } else if (j0 == 2) { p->icell = 1; p->x = s->xface[0]; }
    else if (j0 == 3) { p->jcell = s->ny + 1; p->y = s->yface[s->ny]; } else if (j0 == 4) { p->jcell = 1; p->y = s->yface[0]; }
    else if (j0 == 5) { p->kcell = s->nz + 1; p->z = s->zface[s->nz]; } else { p->kcell = 1; p->z = s->zface[0]; }
    if (p->icell == s->nx + 1 && p->kx < 0.0) p->icell = s->nx;
    if (p->jcell == s->ny + 1 && p->ky < 0.0) p->jcell = s->ny;
    if (p->kcell == s->nz + 1 && p->kz < 0.0) p->kcell = s->nz;
    /* the reference reads grid%vfx(icell,jcell,kcell) next (:149): a start index of n+1 (a ray through an edge of the box)
     * would be out of bounds there, so such pixels are left empty here */
    return p->icell >= 1 && p->icell <= s->nx && p->jcell >= 1 && p->jcell <= s->ny && p->kcell >= 1 && p->kcell <= s->nz;
}

/* make_sightline_tau_{outside,outside_amr,inside,inside_amr,clump}: tau_gas(nxfreq,nxim,nyim,nobs), N_gas(nxim,nyim,nobs),
 * tau_dust(nxim,nyim,nobs) (filled when DGR > 0).  HEALPix observers: nxim = npix, nyim = 1. */
int orc_sightline_tau(const orc_sim *s, double *tau_gas, double *N_gas, double *tau_dust) {
    if (s->nobs <= 0) { snprintf(g_err, sizeof g_err, "orc_sightline_tau: no observers"); return 1; }
    const int nxi = s->nside > 0 ? s->npix : s->nxim, nyi = s->nside > 0 ? 1 : s->nyim;
    const size_t n2 = (size_t)nxi * nyi;
    memset(tau_gas, 0, sizeof(double) * n2 * s->nxfreq * s->nobs); memset(N_gas, 0, sizeof(double) * n2 * s->nobs);
    if (tau_dust) memset(tau_dust, 0, sizeof(double) * n2 * s->nobs);
    for (int io = 0; io < s->nobs; io++) {
        #pragma omp parallel for schedule(dynamic, 8)
        for (long q = 0; q < (long)n2; q++) {
            int ix = (int)(q % nxi) + 1, iy = (int)(q / nxi) + 1;
            photon_t p; double dist, N, td;
            if (!sight_start(s, io, ix, iy, &p, &dist)) continue;
            double u1 = 0.0, Dloc = s->Dfreq_ref;
            if (!s->use_clump_medium) {
                size_t c = s->use_amr_grid ? (size_t)(p.icell - 1) : IDX(s, p.icell, p.jcell, p.kcell);
                u1 = s->vfx[c] * p.kx + s->vfy[c] * p.ky + s->vfz[c] * p.kz; Dloc = s->Dfreq[c];
            }
            double *out = tau_gas + ((size_t)io * n2 + q) * s->nxfreq;
            for (int kf = 1; kf <= s->nxfreq; kf++) {
                double xf = (kf - 0.5) * s->dxfreq + s->xfreq_min;
                if (s->use_clump_medium) { p.xfreq = xf; out[kf-1] = sight_clump(s, &p, 0, 0, 0); }
                else {
                    p.xfreq = xf * s->Dfreq_ref / Dloc - u1;                                  /* the local cell frame: :153 */
                    out[kf-1] = s->use_amr_grid ? sight_amr(s, &p, dist, 0, 0, 0) : tau_gas_car(s, &p, dist);
                }
            }
            if (s->use_clump_medium) sight_clump(s, &p, 1, &N, &td);
            else if (s->use_amr_grid) sight_amr(s, &p, dist, 1, &N, &td);
            else column_car(s, &p, dist, &N, &td);
            N_gas[(size_t)io * n2 + q] = N;
            if (tau_dust && s->DGR > 0.0) tau_dust[(size_t)io * n2 + q] = td;
        }
    }
    return 0;
}
void orc_raytrace_tau_gas(const orc_sim *s, const double r[8], const int ic[4], double dist, double *tau) {
    photon_t p; unpack_s(s, r, ic, &p);
    *tau = s->use_clump_medium ? sight_clump(s, &p, 0, 0, 0) : s->use_amr_grid ? sight_amr(s, &p, dist, 0, 0, 0) : tau_gas_car(s, &p, dist);
}
void orc_raytrace_column(const orc_sim *s, const double r[8], const int ic[4], double dist, double *N_gas, double *tau_dust) {
    photon_t p; unpack_s(s, r, ic, &p);
    if (s->use_clump_medium) sight_clump(s, &p, 1, N_gas, tau_dust); else if (s->use_amr_grid) sight_amr(s, &p, dist, 1, N_gas, tau_dust); else column_car(s, &p, dist, N_gas, tau_dust);
}

int64_t orc_scatter_event(orc_sim *s, double st[18], int ic[4], uint64_t seed, uint64_t id, uint64_t n0) {
    photon_t p; unpack_s(s, st, ic, &p);
    p.mx = st[8]; p.my = st[9]; p.mz = st[10]; p.nx = st[11]; p.ny = st[12]; p.nz = st[13]; p.I = st[14]; p.Q = st[15]; p.U = st[16]; p.V = st[17];
    p.iband = (ic[3] == 2) ? 2 : 1;                          /* ic[3]: 1 = a packet of the line's band, 2 = a band-2 (H-alpha) packet of line type 8 */
    rng_t g; rng_philox(&g, seed, id, n0); acc_t A; acc_alloc(&A, s);
    scattering(s, &A, &g, &p);
    acc_merge_free(&A, s);
    st[0] = p.x; st[1] = p.y; st[2] = p.z; st[3] = p.kx; st[4] = p.ky; st[5] = p.kz; st[6] = p.xfreq; st[7] = p.wgt;
    st[8] = p.mx; st[9] = p.my; st[10] = p.mz; st[11] = p.nx; st[12] = p.ny; st[13] = p.nz; st[14] = p.I; st[15] = p.Q; st[16] = p.U; st[17] = p.V;
    ic[0] = p.icell; ic[1] = p.jcell; ic[2] = p.kcell; ic[3] = p.inside ? (p.iband == 2 ? 2 : 1) : 0;
    return (int64_t)(g.n - n0);
}

int64_t orc_set_array(orc_sim *s, const char *k, const double *in, int64_t n) {
    int64_t nc = (int64_t)s->nx * s->ny * s->nz; double *dst = NULL;
    if (s->use_clump_medium && !strncmp(k, "cl_", 3)) {            /* per-clump properties (tests: clumps of different temperature / opacity) */
        clumps_t *c = &s->cl; double *d = NULL;
        if (!strcmp(k, "cl_has_overlap") && n == 1) {                 /* tests: run a non-overlapping population through the overlap forms */
            c->has_overlap = in[0] != 0.0; s->raytrace_variant = c->has_overlap ? 7 : 4; return 1; }
        if (!strcmp(k, "cl_rebuild") && n == 1) {                     /* tests: a population loaded from a clump file (read_clumps_info :2040-2312): new positions / radii ->
                                                                        * build_clump_csr again, check_has_overlap as init_clumps :678-679 does for a loaded file, the scalars of grid_create_clump */
            c->radius_max = 0.0; for (int64_t i = 0; i < c->N; i++) { c->radius2[i] = c->radius[i] * c->radius[i]; if (c->radius[i] > c->radius_max) c->radius_max = c->radius[i]; }
            c->Dfreq_ref = c->Dfreq[0]; c->voigt_a_ref = c->voigt_a[0]; c->rhokap_ref = c->rhokap[0]; c->vtherm_ref = c->vtherm[0];
            build_clump_csr(c); check_has_overlap(c); s->raytrace_variant = c->has_overlap ? 7 : 4; compute_clump_scalars(s);
            size_t nc = (size_t)s->nx * s->ny * s->nz;
            for (size_t i = 0; i < nc; i++) { s->Dfreq[i] = c->Dfreq_ref; s->voigt_a[i] = c->voigt_a_ref; }
            s->Dfreq_ref = c->Dfreq_ref; s->voigt_amean = c->voigt_a_ref;
            return 1; }
        if (!strcmp(k, "cl_x")) d = c->x; else if (!strcmp(k, "cl_y")) d = c->y; else if (!strcmp(k, "cl_z")) d = c->z; else if (!strcmp(k, "cl_radius")) d = c->radius; else if (!strcmp(k, "cl_vtherm")) d = c->vtherm; else
        if (!strcmp(k, "cl_rhokap")) d = c->rhokap; else if (!strcmp(k, "cl_rhokapD")) d = c->rhokapD; else if (!strcmp(k, "cl_voigt_a")) d = c->voigt_a;
        else if (!strcmp(k, "cl_Dfreq")) d = c->Dfreq; else if (!strcmp(k, "cl_vx")) d = c->vx; else if (!strcmp(k, "cl_vy")) d = c->vy; else if (!strcmp(k, "cl_vz")) d = c->vz;
        if (!d || n != c->N) return -1;
        memcpy(d, in, (size_t)n * 8); return n;
    }
    if (!strcmp(k, "rhokap")) dst = s->rhokap; else if (!strcmp(k, "voigt_a")) dst = s->voigt_a; else if (!strcmp(k, "Dfreq")) dst = s->Dfreq;
    else if (!strcmp(k, "vfx")) dst = s->vfx; else if (!strcmp(k, "vfy")) dst = s->vfy; else if (!strcmp(k, "vfz")) dst = s->vfz; else if (!strcmp(k, "rhokapD")) dst = s->rhokapD;
    if (!dst || n != nc) return -1;
    memcpy(dst, in, nc * 8); return nc;
}

