/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the LaRT photon-packet hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / reported CPU baseline.  The product path (lart_amd/csrc, include/lart_gpu.h)
 * never links, imports or falls back to it.
 *
 * Parity status: PINNED at function level against oracle/_ref (the reference's own MPI-free
 * modules compiled from /root/reference: voigt, MT19937-64, all samplers, raytrace_to_{tau,edge,dist}_car*)
 * through tests/golden/, and at whole-path level against the known answers in the reference's
 * shipped run logs (<N_scatt>, N(HI)_pole; tests/test_oracle_known_answers.py).  The routines that
 * cannot be compiled here without an MPI stand-in (scattering_car, generate_photon, peelingoff_rect,
 * grid_mod_car, run_simulation_mod) are restated below with file:line citations.
 */
#ifndef LART_ORACLE_H
#define LART_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_sim orc_sim;

/* Create a simulation from a whitespace-separated "name=value" list whose names are the reference's
 * par% namelist fields (define.f90:209-544), e.g. "temperature=10 taumax=1e6 xyz_symmetry=1 nx=101".
 * Applies the read_input post-processing (setup.f90:42-562), builds the uniform Cartesian medium
 * (grid_mod_car.f90:11-1238 subset), the frequency grid and the observers.  NULL on error
 * (orc_last_error()). */
orc_sim *orc_create(const char *params);
void     orc_destroy(orc_sim *);
const char *orc_last_error(void);

/* rng_kind: 0 = MT19937-64 stream per worker seeded iseed+9999*worker (random_mt.f90:889-956; photons
 * statically interleaved over workers exactly like run_equal_number, run_simulation_mod.f90:150-151);
 * 1 = Philox4x32-10 keyed (seed, photon id), counter = draw index (the GPU build's generator:
 * results independent of the number of workers). Photon ids are first_id + k*stride, k = 0..count-1. */
int orc_run(orc_sim *, int64_t first_id, int64_t stride, int64_t count, int rng_kind, uint64_t seed, int nthreads);

/* scalars by name: nscatt_gas nscatt_dust (weight sums over photons), nphotons_done, n_cellsteps,
 * n_cellsteps_peel, n_events, dx dy dz xmin ymin zmin i0 j0 k0 Dfreq_ref xfreq_min xfreq_max dxfreq nxfreq
 * taumax tauhomo N_gasmax N_gashomo voigt_amean atau3 nobs dxim dyim distance ...  returns NaN if unknown */
double orc_get_scalar(const orc_sim *, const char *name);
/* arrays by name (raw tallies, not normalised): Jout Jin Jabs Jmu rhokap voigt_a Dfreq vfx vfy vfz rhokapD
 * xface yface zface obs_xyz rmatrix; peel cubes "scatt:<iobs>" "direc:<iobs>" "I:.." "Q:.." "U:.." "V:.."
 * and "*_2D:<iobs>".  Returns element count (copies at most n elements), -1 if unknown. */
int64_t orc_get_array(const orc_sim *, const char *name, double *out, int64_t n);
/* overwrite a medium table (tests: non-uniform fields for ray-march parity) */
int64_t orc_set_array(orc_sim *, const char *name, const double *in, int64_t n);
void    orc_zero_tallies(orc_sim *);
/* normalised spectrum like output_normalize_outside (output_sum_rect.f90:185-202) */
int     orc_normalized_Jout(const orc_sim *, double *out, int n);

/* per-photon record of the last orc_run (count entries, indexed k): nscatt, xfreq_ref at exit, flag */
int64_t orc_get_photon_log(const orc_sim *, double *nscatt, double *xfreq_out, int64_t n);
/* history probe of the divergence audit: 10 doubles per record (birth, every event): draws consumed, xfreq, x, y, z, cell code, kz, kind,
 * smallest decision margin of the event, id of that comparison; counter-based stream (seed, id); tallies are not touched */
int64_t orc_trace(orc_sim *, uint64_t seed, uint64_t id, int64_t cap, double *out);
int64_t orc_trace_state(orc_sim *, uint64_t seed, uint64_t id, int64_t at, double *out27);

/* ---- function-level entry points (pinned against oracle/_ref, and used to check device functions) ---- */
double orc_voigt(double x, double a);                                   /* voigt_mod.f90:541-733 */
void   orc_voigt_v(int n, const double *x, const double *a, double *h);
void   orc_line_constants(double *cross0, double *vtherm1, double *damping, double *wavelength0, double *g_recoil0);
/* samplers: kind/seed as in orc_run; for kind 1 the stream is (seed, id=stream_id) */
void   orc_rand_number_v(int kind, uint64_t seed, uint64_t stream_id, int n, double *u);
void   orc_rand_resonance_vz_v(int kind, uint64_t seed, uint64_t stream_id, int n, double x, double a, double *vz);
void   orc_rand_resonance_v(int kind, uint64_t seed, uint64_t stream_id, int n, double E1, double *c);
void   orc_rand_hg_v(int kind, uint64_t seed, uint64_t stream_id, int n, double g, double *c);
void   orc_rand_voigt_v(int kind, uint64_t seed, uint64_t stream_id, int n, double a, double *c);
void   orc_rand_gauss_v(int kind, uint64_t seed, uint64_t stream_id, int n, double *c);
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* ray marches on the sim's grid. r = [x,y,z,kx,ky,kz,xfreq,wgt], ic = [icell,jcell,kcell,inside] (1-based)
 * variant 0 = plain, 1 = xyz_symmetry, 2 = z-only */
void   orc_raytrace_to_tau(orc_sim *, int variant, double r[8], int ic[4], double tau_in);
/* amr_find_leaf (octree_mod.f90:149-171): 1-based leaf holding the point, 0 = outside the box */
int orc_amr_find_leaf(const orc_sim *s, double x, double y, double z);
void   orc_raytrace_to_edge(const orc_sim *, int variant, const double r[8], const int ic[4], double *tau);
void   orc_raytrace_to_dist(const orc_sim *, const double r[8], const int ic[4], double dist, double *tau);
/* sight-line maps (make_sightline_tau, sightline_tau_{rect,heal,clump}.f90): tau_gas[nobs][nyim][nxim][nxfreq] in C order
 * (= Fortran (nxfreq,nxim,nyim) per observer), N_gas[nobs][nyim][nxim], tau_dust likewise (may be NULL).  Needs observers. */
int    orc_sightline_tau(const orc_sim *, double *tau_gas, double *N_gas, double *tau_dust);
/* the marches behind it: raytrace_to_{edge,dist}_*_tau_gas / _column (gas only, no tau_huge cut); dist < 0 = to the edge */
void   orc_raytrace_tau_gas(const orc_sim *, const double r[8], const int ic[4], double dist, double *tau);
void   orc_raytrace_column(const orc_sim *, const double r[8], const int ic[4], double dist, double *N_gas, double *tau_dust);
void   orc_pix2vec(int nside, int ipix, double v[3]);                    /* healpix.f90:114-187 */
/* do_resonance1 / do_resonance2 (line_mod.f90:108-192; par%fine_structure selects type 2) and calc_voigt1/2 (:38-62) in a cell:
 * out[12*i + ..] = uz, xfreq_atom, cost, sint, S11, S22, S12, S33, S44, E1, E2, E3 */
void   orc_do_resonance_v(const orc_sim *, int kind, uint64_t seed, uint64_t stream_id, int n, double xfreq, int ic, int jc, int kc, double *out);
void   orc_calc_voigt_v(const orc_sim *, int n, const double *x, int ic, int jc, int kc, double *h);
int    orc_vec2pix(int nside, double vx, double vy, double vz);          /* healpix.f90:29-112 */
double orc_xcrit_local(const orc_sim *, int i, int j, int k, double x, double y, double z); /* grid_mod_car.f90:1598-1629 */
/* one full scattering event from a given photon state (for GPU device-function parity):
 * st = [x,y,z,kx,ky,kz,xfreq,wgt, mx,my,mz,nx,ny,nz, I,Q,U,V], ic as above; Philox stream (seed,id) from draw n0;
 * returns draws consumed */
void orc_raytrace_to_tau_shear(orc_sim *, double r[8], int ic[4], double tau_in, double *shear);   /* raytrace_to_tau_car_xyper_shear, vfy_shear in / out */
void orc_raytrace_to_tau_band(orc_sim *, int iband, double r[8], int ic[4], double tau_in);
void orc_raytrace_to_edge_dust_Ha(const orc_sim *, const double r[8], const int ic[4], double *tau);
void orc_h2_kappa_v(const orc_sim *, int n, const double *x, double Dfreq, double *kap);
void orc_h2_select_v(const orc_sim *, int kind, uint64_t seed, uint64_t id, int n, const double *x, double Dfreq, int *il);
int64_t orc_scatter_event(orc_sim *, double st[18], int ic[4], uint64_t seed, uint64_t id, uint64_t n0);
#ifdef __cplusplus
}
#endif
#endif
