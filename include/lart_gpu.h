/* lart_gpu.h -- C-ABI of the MI355X-native LaRT photon-packet path (liblart_gpu.so).
 *
 * Drop-in boundary: the reference crosses into its hot path through the Fortran procedure pointer
 *     procedure(run_sim), pointer :: run_simulation      (define.f90:832-838, called at main.f90:42)
 * after grid_create / observer_create have filled the read-only tables, and leaves it through
 *     output_reduce(grid)                                 (define.f90:897-904, main.f90:46).
 * The entry points below are what a maintainer binds with ISO_C_BINDING in place of those two calls
 * (INTEGRATION.md shows the Fortran interface block).  Plain pointers and sizes only; arrays are
 * borrowed for the duration of the call and copied to the device inside; Fortran column-major order
 * and 1-based cell indices are kept as the reference stores them.
 *
 * Every function returns 0 on success, non-zero on error (lart_gpu_last_error() has the text);
 * the host prints it and stops, like the reference's "message + MPI_ABORT" convention (setup.f90:132-139).
 * There is no CPU fallback: without a visible gfx950 device lart_gpu_create() fails.
 */
#ifndef LART_GPU_H
#define LART_GPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LART_MAX_OBSERVERS 16

/* Run-time switches and scalars the kernels read.  Field names are the reference's par%/grid%/line%
 * names (define.f90:209-544, :117-206, :639-680). */
typedef struct lart_params {
    /* geometry / boundary variant: selects raytrace_to_{tau,edge}_car | _car_xyzsym | _car_xysym | _car_zonly (xy_periodic
     * with nx = ny = 1) | _car_xyper (xy_periodic otherwise) exactly as setup_procedure does (setup.f90:951-980) */
    int32_t xyz_symmetry, xy_periodic, xy_symmetry;
    /* source: generate_photon.f90:33-132, :243-306 */
    int32_t source_geometry;       /* 0 'point', 1 'uniform_sphere', 2 'uniform_cylinder', 3 'uniform', 4 'uniform_xy', 5 'gaussian' (generate_photon.f90:31-72) */
    int32_t spectral_type;         /* 0 'monochromatic', 1 'voigt', 2 'continuum', 3 'gaussian' (generate_photon.f90:243-266) */
    int32_t comoving_source;
    /* scattering physics: scattering_car.f90 */
    int32_t recoil, core_skip, core_skip_global, use_stokes, use_reduced_wgt;
    /* tallies */
    int32_t save_Jin, save_Jabs, save_Jmu, nmu;
    int32_t save_peeloff, save_peeloff_2D, save_peeloff_3D;
    int32_t save_direc0, save_dust_scattered;  /* extra peel-off tallies: unattenuated source (direc0), dust-scattered light (scatt_dust) */
    int32_t nxfreq;
    int32_t line_type;             /* line%line_type: 1 single resonance line, 2 S-P doublet sharing one lower level (par%fine_structure and the
                                    * doublet line ids, line_mod.f90:590-713, 1250-1253), 4 / 5 / 6 resonance + fluorescence multiplets with 1 / 2 / 3
                                    * upper levels (tables through lart_gpu_set_line_multiplet; 6 = He I 10833 with par%HeI_coherent),
                                    * 7 hydrogen + deuterium ('ly_alpha_HD' / par%include_deuterium, :1155-1191) */
    double  xs_point, ys_point, zs_point, source_rmax, xfreq0;
    double  xfreq_min, dxfreq, Dfreq_ref;          /* grid%xfreq_min, grid%dxfreq, grid%Dfreq_ref */
    double  xfreq_max, source_sigma_x;             /* grid%xfreq_max (continuum source); Gaussian source sigma / vtherm_total(par%temperature) */
    double  xrange, yrange, zrange, source_zscale; /* grid%xrange.. (extended sources), par%source_zscale */
    double  DGR, hgg, albedo;
    double  mu_min, dmu;
    double  g_recoil0, E1, E2, E3;                 /* line%g_recoil0, line%E1..E3 (line_mod.f90:1255-1270) */
    double  DnuHK_Hz;                              /* line%DnuHK_Hz: 2P3/2 - 2P1/2 splitting in Hz (line type 2; 0 otherwise) */
    double  delta_nu_HD_Hz, ratio_Dfreq_HD, ratio_voigta_HD, nD_over_nH, g_recoil0_D;   /* line type 7: the cross-species constants of line% */
    double  xcrit, xcrit2;                         /* grid%xcrit, grid%xcrit2 (core_skip_global) */
    /* shearing box (TIGRESS-type boxes): par%Omega as grid_create leaves it (grid_mod_car.f90:346-351), i.e. the change of the y velocity, in
     * units of the thermal speed like grid%vfy, of a packet that crosses the periodic x boundary.  Non-zero with xy_periodic and nx, ny > 1
     * selects raytrace_to_tau_car_xyper_shear (setup.f90:967-969, raytrace_car.f90:2677-2954); raytrace_to_edge then stays the plain march */
    double  Omega;
    /* line type 8, 'ly_beta' (do_resonance8 line_mod.f90:293-345): a scattering converts the packet to H-alpha (band 2) with probability
     * P_conv = line%b(1)%P_down(2); E1..E3 of the two channels; the band-2 spectral axis grid%xfreq_min_Ha / dxfreq_Ha / nxfreq_Ha
     * (grid_mod_car.f90:1518-1529) and dust constants par%R_Ha = cext_dust_Ha / cext_dust, par%albedo_Ha, par%hgg_Ha (define.f90:461-468) */
    int32_t nxfreq_Ha, h2_nlines;                  /* h2_nlines > 0: the H2 lines next to Ly-alpha are on (h2_mod.f90: h2_on), at most 4 */
    double  xfreq_min_Ha, dxfreq_Ha, R_Ha, albedo_Ha, hgg_Ha, P_conv, E_conv[3];
    /* H2 pumping by Ly-alpha (par%h2_model = 'neufeld', h2_mod.f90:139-222): per line h2l(:)%dnu_Hz, %strength, %a_damp, %p_scat, the
     * H2 Doppler width h2_Dfreq_Hz, and par%h2_hi_width (1: the lines take the H I Doppler width of the cell) */
    int32_t h2_hi_width, pad_;
    double  h2_Dfreq_Hz, h2_dnu_Hz[4], h2_strength[4], h2_a_damp[4], h2_p_scat[4];
} lart_params;

/* Host views of the tallies the path adds into (reference: grid%Jout/Jin/Jabs/Jmu, par%nscatt_*,
 * observer(:)%scatt/direc/I/Q/U/V and *_2D).  NULL members are skipped. Peel arrays are
 * [nobs][nyim][nxim][nxfreq] in C order = Fortran (nxfreq,nxim,nyim) per observer. */
typedef struct lart_tallies {
    double *Jout, *Jin, *Jabs, *Jmu;                         /* nxfreq, nxfreq, nxfreq, nxfreq*nmu */
    double *scatt, *direc, *I, *Q, *U, *V;                   /* nobs*nxfreq*nxim*nyim each */
    double *scatt_2D, *direc_2D, *I_2D, *Q_2D, *U_2D, *V_2D; /* nobs*nxim*nyim each */
    double *direc0, *direc0_2D, *scatt_dust;                 /* observer(:)%direc0, %direc0_2D (peelingoff_rect.f90:108-125), %scatt_dust (:284-287, :567-570) */
    double *J1, *P1, *P1_new;                                /* nxfreq*nbins, nbins, nbins (lart_gpu_set_JPa) */
    double nscatt_gas, nscatt_dust;                          /* weight sums over finished photons (par%nscatt_*) */
    double n_events;                                         /* scattering events executed */
    double n_cellsteps, n_cellsteps_peel;                    /* cells visited by flight / peel rays (roofline accounting) */
    double n_atomics;                                        /* non-zero peel-off deposits issued as tally atomics (roofline accounting) */
    int64_t nphotons_done, nphotons_active;
    /* line type 8: grid%Jout_Ha, grid%Jabs_Ha (nxfreq_Ha each), observer(:)%peel_Ha (nobs*nxfreq_Ha*nxim*nyim), the conversion-rate profile
     * of add_to_Pconv (nbins), and the weight budget W_conv, W_esc1, W_abs1, W_esc2, W_abs2 (define.f90:471-475) in aux[0..4];
     * H2: W_H2abs, W_H2scat (define.f90:490-491) in aux[0..1], W_H2pump(1..h2_nlines) (h2_mod.f90:49) in aux[2..] */
    double *Jout_Ha, *Jabs_Ha, *peel_Ha, *Pconv;
    double aux[8];
} lart_tallies;

typedef struct lart_gpu lart_gpu;   /* one context = one GPU */

int  lart_gpu_device_count(void);
/* replaces the set-up half of setup_procedure + the table hand-over; device = HIP ordinal */
int  lart_gpu_create(lart_gpu **ctx, const lart_params *par, int device);
int  lart_gpu_destroy(lart_gpu *ctx);
const char *lart_gpu_last_error(void);

/* medium: grid_type tables built by grid_create (grid_mod_car.f90:184-203, :273-538).
 * rhokapD may be NULL when DGR == 0. */
int  lart_gpu_set_grid_car(lart_gpu *ctx, int32_t nx, int32_t ny, int32_t nz,
                           double xmin, double ymin, double zmin, double dx, double dy, double dz,
                           int32_t i0, int32_t j0, int32_t k0,
                           const double *xface, const double *yface, const double *zface,
                           const double *rhokap, const double *voigt_a, const double *Dfreq,
                           const double *vfx, const double *vfy, const double *vfz, const double *rhokapD);
/* Mueller table as setup_scattering_matrix leaves it (setup.f90:581-649): normalised S11..S34 on the
 * equispaced coss grid (nPDF points), phase_PDF/alias with nPDF-1 entries (alias 1-based). */
int  lart_gpu_set_mueller(lart_gpu *ctx, int32_t nPDF, const double *coss, const double *S11, const double *S12,
                          const double *S33, const double *S34, const double *phase_PDF, const int32_t *alias);
/* far observers as observer_create_outside leaves them (observer_rect.f90:207-285): xyz[3*nobs],
 * rmatrix[9*nobs] with rmatrix(r,c) at [9*i + 3*(r-1) + (c-1)] */
int  lart_gpu_set_observers(lart_gpu *ctx, int32_t nobs, const double *xyz, const double *rmatrix,
                            double dxim, double dyim, int32_t nxim, int32_t nyim);

/* observers inside the medium, HEALPix all-sky maps (observer_create_inside, observer_heal.f90:24-80): xyz[3*nobs], nside.
 * Tallies use the scatt/direc (+_2D) members of lart_tallies with nxim = 12*nside^2, nyim = 1.  Plain Cartesian grid or octree
 * (peeling_*_inside_amr, peelingoff_amr.f90:656-849); no Stokes vectors (setup.f90:181), no symmetry / periodic variant, no clumps. */
int  lart_gpu_set_observers_heal(lart_gpu *ctx, int32_t nobs, const double *xyz, int32_t nside);
/* J(x) / P_alpha profile tallies, the reference's compile-time switches CALCJ / CALCP / CALCPnew as run-time flags
 * (add_to_J raytrace_car.f90:3979-4011, add_to_Pnew :4015-4045, add_to_Pa scattering_car.f90:830-859).
 * ind_bin[ncell] = 1-based profile bin of each cell as create_JPa_mem builds it (grid_mod_car.f90:1242-1434:
 * ind_sph for geometry_JPa = 1, kcell for -1, the cell index for 3), 0 = cell belongs to no bin; cross0 = line%cross0. */
int  lart_gpu_set_JPa(lart_gpu *ctx, int32_t calc_J, int32_t calc_P, int32_t calc_Pnew, int32_t nbins, const int32_t *ind_bin, double cross0);

/* Octree AMR medium as grid_create_amr leaves it (grid_mod_amr.f90:34-526; tree tables of amr_grid_type,
 * octree_mod.f90:19-138).  All indices 1-based, 0 = none:  children[8*ncells] (sub-octant fastest; bit0 = x >= cx,
 * bit1 = y, bit2 = z), neighbor[6*ncells] (face fastest; faces +x,-x,+y,-y,+z,-z), ileaf[ncells] (leaf number of a
 * cell, 0 for internal nodes), icell_of_leaf[nleaf]; cell centres cx,cy,cz and half-widths ch per cell; the box is
 * the cube [lo,hi]^3.  Media tables are per LEAF (length nleaf), same meaning as lart_gpu_set_grid_car.
 * Replaces raytrace_{to_tau,to_edge,to_dist}_amr and scattering_amr for the kernels; par.xyz_symmetry selects the
 * mirror planes of the +++ octant (raytrace_amr.f90:176-201). */
int  lart_gpu_set_grid_amr(lart_gpu *ctx, int32_t ncells, int32_t nleaf, double lo, double hi,
                           const double *cx, const double *cy, const double *cz, const double *ch,
                           const int32_t *children, const int32_t *neighbor, const int32_t *ileaf, const int32_t *icell_of_leaf,
                           const double *rhokap, const double *voigt_a, const double *Dfreq,
                           const double *vfx, const double *vfy, const double *vfz, const double *rhokapD);
/* profile tallies on the octree: bins by POSITION (amr_JPa_bin, octree_mod.f90:286-311): geometry_JPa 1 = radial
 * shells of width dbin (nbins of them), -1 = z slabs of width dbin from lo, 3 = one bin per leaf (nbins = nleaf). */
int  lart_gpu_set_JPa_amr(lart_gpu *ctx, int32_t calc_J, int32_t calc_P, int32_t calc_Pnew, int32_t geometry_JPa, int32_t nbins, double dbin, double cross0);

/* Clump medium as init_clumps leaves it (clump_mod.f90:646-891; arrays cl_x..cl_Dfreq :37-71, CSR grid :110-118):
 * n spherical clumps in vacuum inside the sphere of radius sphere_R.  Per clump: centre x,y,z, bulk velocity vx,vy,vz in
 * units of the clump's thermal speed, radius, line-centre opacity rhokap per unit length, dust opacity rhokapD (NULL when
 * par.DGR = 0), voigt_a, Dfreq [Hz].  Dfreq_ref / voigt_a_ref = cl_Dfreq_ref / cl_voigt_a_ref (the units of xfreq and the
 * uniform background record of grid_create_clump, grid_mod_clump.f90:101-113).  CSR acceleration grid: cgx^3 cells of edge
 * cg_d from cg_min on every axis, cg_start[cgx^3+1] (1-based offsets, cg_start[0] = 1), cg_list[cg_start[cgx^3]-1] (1-based
 * clump numbers).  Replaces raytrace_to_{tau,edge}_clump[_capped], do_resonance1_clump and the clump branches of scattering /
 * generate_photon / peeling_*_outside for the kernels.
 * lart_gpu_set_clump_overlap(ctx, 1) after it = setup_clump_overlap (setup.f90:1053-1082) when check_has_overlap found a pair of
 * overlapping clumps: the kernels then run raytrace_to_{tau,edge}_clump_overlap[_capped] (raytrace_clump.f90:606-918; the frequency of the
 * flight stays in the global frame, the opacity inside k clumps is the sum of their k profiles) and scatter_resonance_clump_{no,}stokes
 * (scattering_car.f90:898-937; the scattering clump drawn opacity-weighted), and lart_gpu_sightline_tau the overlap forms of its two
 * marches (:922-1001).  Single-line profiles (line type 1). */
int  lart_gpu_set_clumps(lart_gpu *ctx, int64_t n, const double *x, const double *y, const double *z,
                         const double *vx, const double *vy, const double *vz, const double *radius,
                         const double *rhokap, const double *rhokapD, const double *voigt_a, const double *Dfreq,
                         double sphere_R, double Dfreq_ref, double voigt_a_ref,
                         int32_t cgx, double cg_min, double cg_d, const int32_t *cg_start, const int32_t *cg_list);
int  lart_gpu_set_clump_overlap(lart_gpu *ctx, int32_t has_overlap);

/* ---- resonance + fluorescence multiplets (line types 4, 5, 6 of setup_resonance_line, line_mod.f90:713-1153) ----
 * nup <= 3 upward transitions from the ground level (f12, delE_Hz = frequency offset of the level from the first one, damping_up =
 * sum of its A21), each with ndown <= 4 downward branches: P_down/A_down = the alias table of random_alias_setup (P_down scaled
 * by ndown, A_down 1-based; random_mt.f90:2016-2067) for A21/damping_up, Elow_Hz = the lower level above the ground level (the
 * fluorescence photon comes out that much redder), E1..E3 = the phase-function coefficients of that channel.  The per-line
 * scalars (wavelength0, cross0, vtherm1, g_recoil0, damping) are what the hosts use for the grid set-up.
 * lart_host_line_lookup fills the record for a par%line_id (0, or -1 if the id is not one of the multiplets; CPU, no device);
 * lart_gpu_set_line_multiplet hands it to the kernels -- required before a run when lart_params.line_type is 4, 5 or 6
 * (replaces line%b(:) read by calc_voigt3 / do_resonance4,5,6[_clump] / generate_photon, line_mod.f90:64-80, :249-474,
 * generate_photon.f90:161-243). */
#define LART_LINE_MAX_UP   3
#define LART_LINE_MAX_DOWN 4
typedef struct lart_line_multiplet {
    int32_t line_type, nup;
    int32_t HeI_coherent;        /* par%HeI_coherent (type 6 only): E1..E3 of the atom-frame frequency, compute_HeI_E_coherent line_mod.f90:8-34 */
    int32_t ndown[LART_LINE_MAX_UP];
    int32_t A_down[LART_LINE_MAX_UP][LART_LINE_MAX_DOWN];
    char    ion_id[8];
    double  wavelength0, cross0, vtherm1, g_recoil0, damping;
    double  f12[LART_LINE_MAX_UP], delE_Hz[LART_LINE_MAX_UP], damping_up[LART_LINE_MAX_UP];
    double  A21[LART_LINE_MAX_UP][LART_LINE_MAX_DOWN], Elow_Hz[LART_LINE_MAX_UP][LART_LINE_MAX_DOWN];
    double  E1[LART_LINE_MAX_UP][LART_LINE_MAX_DOWN], E2[LART_LINE_MAX_UP][LART_LINE_MAX_DOWN], E3[LART_LINE_MAX_UP][LART_LINE_MAX_DOWN];
    double  P_down[LART_LINE_MAX_UP][LART_LINE_MAX_DOWN];
} lart_line_multiplet;
int  lart_host_line_lookup(const char *line_id, lart_line_multiplet *out);
int  lart_gpu_set_line_multiplet(lart_gpu *ctx, const lart_line_multiplet *m);

/* ---- host-side set-up helpers for the clump medium (CPU, no device): the per-clump loops of init_clumps ----
 * lart_host_clumps_generate: generate_clumps, uniform path (clump_mod.f90:897-1151): n non-overlapping centres in the shell
 *   [r_min, sphere_R] by random sequential adsorption with the reference's grid and its MT19937-64 stream seeded with
 *   par%iseed (rank 0); sigma_v_over_vtherm = par%clump_sigma_v / cl_vtherm adds Gaussian bulk velocities (rand_gauss).
 *   flags: bit 0 = par%clump_fully_inside, bit 1 = par%clump_allow_overlap (a trial that overlaps an earlier clump is kept, :1090).
 * lart_host_clumps_csr_dims / _build_csr: build_clump_csr (:1267-1360); call _build_csr with cg_list = NULL first to
 *   learn the registration count.  lart_host_clumps_has_overlap: check_has_overlap (:1544-1591).  Errors: non-zero + lart_host_last_error(). */
const char *lart_host_last_error(void);
int  lart_host_clumps_generate(int64_t n, double sphere_R, double r_min, double clump_radius, int32_t flags,
                               double sigma_v_over_vtherm, int64_t iseed, double *x, double *y, double *z,
                               double *vx, double *vy, double *vz, double *acceptance);
int  lart_host_clumps_generate_cone(int64_t n, double sphere_R, double r_min, double clump_radius, int32_t flags, double cone_opening_deg,
                                    double sigma_v_over_vtherm, int64_t iseed, double *x, double *y, double *z,
                                    double *vx, double *vy, double *vz, double *acceptance);   /* ... inside the bicone of par%cone_opening about the z axis (:1029-1048) */
int  lart_host_clumps_csr_dims(int64_t n, double sphere_R, double radius_max, int32_t *cgx, double *cg_min, double *cg_d);
int  lart_host_clumps_build_csr(int64_t n, const double *x, const double *y, const double *z, const double *radius,
                                int32_t cgx, double cg_min, double cg_d, int32_t *cg_start, int32_t *cg_list, int64_t *nreg);
int  lart_host_clumps_has_overlap(int64_t n, const double *x, const double *y, const double *z, const double *radius,
                                  int32_t cgx, double cg_min, double cg_d, const int32_t *cg_start, const int32_t *cg_list, int32_t *has_overlap);

/* ---- host-side set-up helpers for the octree (CPU): amr_build_tree + amr_build_neighbors (octree_mod.f90:460-569, 619-682)
 * from a list of leaf cells (centre, refinement level) in the cube [lo,hi]^3, numbered like the reference (root = 1, cells in
 * creation order while the leaves are inserted in input order).  lart_host_amr_count gives ncells; lart_host_amr_build fills
 * the tables lart_gpu_set_grid_amr takes (children[8*ncells], neighbor[6*ncells], ...; cell_level and levelmax optional).
 * Errors: non-zero + lart_host_amr_last_error(). */
const char *lart_host_amr_last_error(void);
int  lart_host_amr_count(int32_t nleaf, const double *x, const double *y, const double *z, const int32_t *level,
                         double lo, double hi, int32_t *ncells);
int  lart_host_amr_build(int32_t nleaf, const double *x, const double *y, const double *z, const int32_t *level,
                         double lo, double hi, int32_t ncells, double *cx, double *cy, double *cz, double *ch,
                         int32_t *children, int32_t *neighbor, int32_t *ileaf, int32_t *icell_of_leaf,
                         int32_t *cell_level, int32_t *levelmax);

/* replaces `call run_simulation(grid)` (main.f90:42): photons first_id, first_id+stride, ... (count of
 * them; the reference's static decomposition `do ip = rank+1, nphotons, nproc`, run_simulation_mod.f90:150).
 * The random stream of a photon depends only on (seed, id). Runs to completion. */
int  lart_gpu_run(lart_gpu *ctx, int64_t first_id, int64_t stride, int64_t count, uint64_t seed);

/* the same in bounded slices (bench / progress printing, cf. par%nprint): begin, then step until done.
 * One step = one kernel launch in which every resident photon lane advances by at most
 * max_events scattering events (lanes refill from the photon counter when a packet escapes). */
int  lart_gpu_begin(lart_gpu *ctx, int64_t first_id, int64_t stride, int64_t count, uint64_t seed);
int  lart_gpu_step(lart_gpu *ctx, int32_t nsteps, int32_t max_events);
int  lart_gpu_sync(lart_gpu *ctx, int64_t *nphotons_done, int64_t *nphotons_active);
/* average device time of the transport kernel launches since begin (HIP events on the launch stream) */
int  lart_gpu_kernel_time(lart_gpu *ctx, double *total_ms, int64_t *nlaunches);
/* optional per-photon record (reference: allph%nscatt_gas / xfreq2, run_simulation_mod.f90:249-358):
 * device keeps nscatt and exit xfreq_ref for photon k = 0..count-1 of the current begin() */
int  lart_gpu_enable_photon_log(lart_gpu *ctx, int32_t on);
int  lart_gpu_get_photon_log(lart_gpu *ctx, double *nscatt, double *xfreq_ref, int64_t n);
/* par%save_all_photons: the per-photon table of make_all_initial_photons / make_all_photons (run_simulation_mod.f90:249-358) that
 * write_output_allph (write_output_rect.f90:1353-1515) writes.  Enable before lart_gpu_begin; afterwards out[10][n] (n = the photon
 * count of lart_gpu_begin, record k = the k-th photon id of this context) holds the columns r (impact parameter of the escaping
 * ray), xfreq1 (at birth), xfreq2 (lab frame at escape or absorption), NSCATT_gas, NSCATT_dust, I, Q, U, V (Stokes runs; weight 0
 * for a photon absorbed by dust) and r0 (impact parameter of the first ray; extended sources).  Costs 96 bytes of HBM per photon. */
int  lart_gpu_enable_all_photons(lart_gpu *ctx, int32_t on);
int  lart_gpu_get_all_photons(lart_gpu *ctx, double *out, int64_t n);
/* pool geometry knobs (0 = default): resident photon lanes and threads per workgroup (a multiple of 64; at most 256 for runs with a Stokes vector,
 * at most 1024 otherwise).  With block = 0 the library chooses: 256, or 1024 when J(x) profiles are tallied together with P / P_new (one workgroup per
 * CU leaves the LDS for the line-centre window of J; results do not depend on the choice). */
int  lart_gpu_set_pool(lart_gpu *ctx, int64_t nlanes, int32_t block);
int  lart_gpu_pool_info(lart_gpu *ctx, int64_t *nlanes, int32_t *block);   /* valid after lart_gpu_begin */
/* test probe: the spare word of every cell record (Cartesian medium: cells, Fortran order; octree: leaves) = the size of the region around the cell in which the
   medium is a copy of it, as the peel prefilter uses it (csrc/lart_gpu.hip: uniform_cubes / uniform_margins_amr).  n = number of cells / leaves given to set_grid_* */
int  lart_gpu_test_cell_margin(lart_gpu *ctx, double *out, int64_t n);
/* Lane compaction = the role of the reference's dynamic master/worker scheduler (run_simulation_mod.f90:31-64: no worker
 * idles while photons are left).  A lane refills itself from the photon counter; once the counter is exhausted the pool
 * drains, and lart_gpu_step packs the live lanes to the front of the pool and launches a smaller grid whenever fewer than
 * 3/4 of the launched lanes are still alive (after a lart_gpu_sync told it so).  lart_gpu_compact forces one pass;
 * lart_gpu_set_auto_compact(ctx, 0) switches the automatic pass off; lart_gpu_live_lanes reports the lanes the next launch
 * covers and the number of passes since lart_gpu_begin.  Results do not depend on it: a history is a function of (seed, id). */
int  lart_gpu_compact(lart_gpu *ctx);
int  lart_gpu_set_auto_compact(lart_gpu *ctx, int32_t on);
int  lart_gpu_live_lanes(lart_gpu *ctx, int64_t *live_lanes, int64_t *ncompactions);

/* replaces `call make_sightline_tau(grid)` (main.f90:36; make_sightline_tau_outside / _outside_amr sightline_tau_rect.f90:11-345,
 * make_sightline_tau_inside / _inside_amr sightline_tau_heal.f90:11-290, make_sightline_tau_clump sightline_tau_clump.f90:33-155):
 * gas optical depth per frequency bin and HI column density along the sight line of every observer pixel, from the far
 * side of the medium towards the observer (gas only, no tau_huge cut).  Needs the medium and the observers; independent of
 * lart_gpu_begin.  Host arrays in Fortran order: tau_gas(nxfreq,nxim,nyim,nobs), N_gas(nxim,nyim,nobs),
 * tau_dust(nxim,nyim,nobs) (NULL to skip; zeros unless DGR > 0); HEALPix observers: nxim = 12 nside^2, nyim = 1.
 * cross0 = line%cross0 (N_gas = sum rhokap Dfreq / cross0 dl, raytrace_car.f90:1383).  Pixels that miss the medium stay 0. */
int  lart_gpu_sightline_tau(lart_gpu *ctx, double cross0, double *tau_gas, double *N_gas, double *tau_dust);
int  lart_gpu_sightline_time(lart_gpu *ctx, double *kernel_ms);   /* device time of the last map (HIP events) */

/* replaces `call output_reduce(grid)` (main.f90:46; output_reduce_outside output_sum_rect.f90:7-149 -> reduce_mem =
 * MPI_REDUCE(SUM) to rank 0, memory_mod_mpi.f90:366-394): ONE ncclReduce(sum, f64) over xGMI of the fused device tally buffer
 * [scalars | Jout | Jin | Jabs | Jmu | peel cubes and images | J1 | P1 | P1_new] to the root context, then lart_gpu_get_tallies
 * on the root copies the totals out.
 *   one process per GPU (mpirun / torchrun ranks): rank 0 calls lart_gpu_comm_unique_id, the host shares the 128 bytes
 *     (MPI_BCAST in a Fortran host), every rank calls lart_gpu_comm_init(ctx, id, nranks, rank), then lart_gpu_reduce_tallies(ctx, root);
 *   one process, all GPUs of the node: lart_gpu_run_multi(ctx[], n, nphotons, seed, max_events) transports the photons
 *     r+1, r+1+n, ... on context r (run_simulation_mod.f90:150-151) and lart_gpu_reduce_tallies_multi(ctx[], n, root) sums them
 *     (communicator from ncclCommInitAll on first use).
 * The root receives into a buffer of its own (as MPI_REDUCE does): every rank's sums stay as they are, so a second reduce gives the
 * same totals, and lart_gpu_get_tallies on the root hands out the totals until that context adds events or zeroes its tallies.
 * lart_gpu_comm_info reports the rank count RCCL itself saw (ncclCommCount).  The device pointer is still available for a
 * caller-owned collective (lart_gpu_tally_buffer: this rank's own sums). */
int  lart_gpu_comm_unique_id(uint8_t *id128);
int  lart_gpu_comm_init(lart_gpu *ctx, const uint8_t *id128, int32_t nranks, int32_t rank);
int  lart_gpu_comm_info(lart_gpu *ctx, int32_t *nranks, int32_t *rank);
int  lart_gpu_comm_destroy(lart_gpu *ctx);
int  lart_gpu_reduce_tallies(lart_gpu *ctx, int32_t root);
int  lart_gpu_reduce_tallies_multi(lart_gpu **ctx, int32_t n, int32_t root);
int  lart_gpu_run_multi(lart_gpu **ctx, int32_t n, int64_t nphotons, uint64_t seed, int32_t max_events);
int  lart_gpu_tally_buffer(lart_gpu *ctx, void **device_ptr, int64_t *n_doubles);
int  lart_gpu_zero_tallies(lart_gpu *ctx);
int  lart_gpu_get_tallies(lart_gpu *ctx, lart_tallies *out);   /* fails (instead of handing out numbers) when a ray of the overlap forms met more
                                                                  * clumps than its event lists hold, or when the peel-off deposit list dropped
                                                                  * an entry: lart_gpu_zero_tallies clears the condition */
/* par%iseed = 0 means "seed from the operating system" in the reference (init_random_mt_seed: /dev/urandom, else clock xor pid,
 * random_mt.f90:889-947).  The hosts call this once, print the value and use it as the Philox key on every rank. CPU, no device. */
int  lart_host_entropy_seed(uint64_t *seed);

/* ---------------------------------------------------------------------------------------------------------------------------------
 * Host set-up helpers (CPU, lart_amd/csrc/lart_host_setup.cpp): the arithmetic the reference does around the photon loop for the media this
 * path supports, ONCE for both hosts (lart_amd/host.py via ctypes, lart_amd/fhost/lart_host.f90 via ISO_C_BINDING).  In/out structs: the
 * members above the "out" comment are inputs.  Errors: non-zero return + lart_host_setup_last_error().
 * --------------------------------------------------------------------------------------------------------------------------------- */
#define LART_MAX_OBSERVERS 16
enum { LART_VEL_NONE = 0, LART_VEL_HUBBLE, LART_VEL_CONSTANT_RADIAL, LART_VEL_PARALLEL, LART_VEL_ROT_SOLID, LART_VEL_ROT_HALO,
       LART_VEL_POWER_LAW, LART_VEL_LINEAR_DECEL, LART_VEL_SSH };          /* par%velocity_type (grid_mod_car.f90:786-946) */
/* line%... of setup_resonance_line (line_mod.f90:551-1272) for 'ly_alpha' (+ par%fine_structure / par%include_deuterium), 'ly_alpha_HD', 'ly_beta' and the
 * single lines / S-P doublets; lart_host_line_setup returns 2 for the ids of lart_host_line_lookup (the multiplets) */
typedef struct lart_host_line {
    int32_t line_type, fine_structure, include_deuterium, pad_;
    double  wavelength0, damping, cross0, vtherm1, mass_amu, g_recoil0, DnuHK_Hz, E1, E2, E3;
    double  delta_nu_HD_Hz, ratio_Dfreq_HD, ratio_voigta_HD, nD_over_nH, g_recoil0_D;
    double  P_conv, E_conv[3], wavelength0_Ha;
    char    line_id[32];                                                    /* 'ly_alpha' + par%include_deuterium becomes 'ly_alpha_HD' */
    char    ion_id[8];                                                      /* label of the log lines: 'H  I', 'C IV', ... (grid_mod_car.f90:1231-1234) */
} lart_host_line;
typedef struct lart_host_h2 { int32_t n, pad_; double Dfreq_Hz, dv_kms[4], dnu_Hz[4], strength[4], a_damp[4], p_scat[4], pop[4], f_osc[4]; } lart_host_h2;      /* h2l(:) of h2_mod.f90:30-49 */
typedef struct lart_host_medium {
    int32_t nx, ny, nz, xyz_symmetry, xy_symmetry, velocity_type, pad0_, pad1_;
    double  xmax, ymax, zmax, rmin, rmax, cone_opening;
    double  vtherm, wavelength0, damping, cross0;                           /* vtherm_total(par%temperature) [km/s]; line%wavelength0 [um], %damping, %cross0 */
    double  dust_opacity;                                                   /* par%cext_dust * par%DGR */
    double  taumax, tauhomo, N_gasmax, N_gashomo;                           /* the first one that is > 0 normalises the opacity; all four hold the medium's values on return */
    double  Vexp, Vx, Vy, Vz, Vrot, rinner, velocity_alpha, rpeak, Vpeak, DeltaV;
    /* out */
    double  dx, dy, dz, xmin, ymin, zmin, Dfreq_ref, voigt_amean, taupole_out, tauhomo_out, N_gaspole_out, N_gashomo_out;
    int32_t i0, j0, k0, pad2_;
} lart_host_medium;
typedef struct lart_host_freq {
    int32_t nxfreq, nwavelength, nvelocity, continuum, line_type, nxfreq_Ha, core_skip_global, pad_;      /* nxfreq, nwavelength, nvelocity, nxfreq_Ha: in and out */
    double  wavelength_min, wavelength_max, velocity_min, velocity_max, xfreq_min, xfreq_max;             /* NaN = not given; xfreq_min / _max: in and out */
    double  vtherm, wavelength0, wavelength0_Ha, Dfreq_ref, DnuHK_Hz, voigt_amean, tauhomo, taumax, Vexp, xfreq_max_Ha, cells_per_radius;   /* cells_per_radius = grid%xmax / grid%dx */
    /* out */
    double  dxfreq, dwave, atau3, xfreq_min_Ha, dxfreq_Ha, dwave_Ha, xcrit, xcrit2;
} lart_host_freq;
typedef struct lart_host_obs {
    int32_t nside, sphere, nxim, nyim, nobs, npix;                          /* nobs, npix (and nxim, nyim for nside > 0): out */
    double  xmax, ymax, zmax, rmax, distance, dxim, dyim;                   /* distance, dxim, dyim: NaN = not given, set on return */
    double  steradian_pix;                                                  /* out */
} lart_host_obs;
typedef struct lart_host_clumps_in {
    double  sphere_R, rmin, clump_radius, N_clumps_in, f_vol_in, f_cov_in, clump_tau0, clump_NHI, clump_nH, taumax, N_gasmax, distance2cm;
    double  vtherm, wavelength0, damping, cross0;                           /* vtherm_total(clump temperature) */
    /* out */
    int64_t N; double r_min, fvol, fcov, Dfreq_ref, voigt_a_ref, rhokap_ref;
} lart_host_clumps_in;
typedef struct lart_host_norm {
    int32_t xy_periodic, sphere;
    double  nphotons, bin_unit, bin_unit_Ha, distance2cm, rmax, xmax, ymax, zmax, amr_L_box, steradian_pix;
    /* out */
    double  area, scale_J, scale_J_Ha, scale_cube, scale_image, scale_cube_Ha;
} lart_host_norm;
/* read_input (setup.f90:160-500): the fields the rules between namelist entries read and change; notes = LART_NOTE_* bits of the messages to print */
enum { LART_UNIT_NONE = 0, LART_UNIT_KPC, LART_UNIT_PC, LART_UNIT_AU, LART_UNIT_OTHER, LART_UNIT_USER };      /* par%distance_unit */
#define LART_NOTE_NO_PEEL_WITH_XYZSYM 1
typedef struct lart_host_rules {
    int32_t nx, ny, nz, nr, nside, nxim, nyim, nmu, geometry_JPa, distance_unit, intensity_unit, notes;
    int32_t xyz_symmetry, xy_symmetry, xy_periodic, use_clump_medium, use_amr_grid, use_stokes, sphere, save_all;
    int32_t save_peeloff, save_peeloff_2D, save_peeloff_3D, save_Jabs, save_Jmu, core_skip, core_skip_global, pad_;
    double  tau0, taumax, N_HI, N_HImax, N_HIhomo, N_gasmax, N_gashomo, cext_dust, DGR, mu_min, dmu;
    double  rmax, xmax, ymax, zmax, source_rmax, distance2cm;
} lart_host_rules;
int    lart_host_input_rules(lart_host_rules *r);
const char *lart_host_setup_last_error(void);
int    lart_host_line_setup(const char *line_id, int32_t fine_structure, int32_t include_deuterium, double D_to_H_ratio, lart_host_line *out);
double lart_host_vtherm_total(double vtherm1, double bturb, double T);      /* define.f90:928-933 */
int    lart_host_h2_init(const char *energy_X_path, double wavelength0_um, double vtherm1, double mass_amu, double bturb, double T_H2, double f_H2,
                         int32_t pure_absorption, lart_host_h2 *out);       /* h2_mod.f90:120-296 */
/* grid_create, analytic media (grid_mod_car.f90:73-946): arrays in Fortran (i,j,k) order, xface(nx+1) ...; rhokapD may be NULL */
int    lart_host_car_medium(lart_host_medium *m, double *xface, double *yface, double *zface, double *rhokap, double *voigt_a, double *Dfreq,
                            double *vfx, double *vfy, double *vfz, double *rhokapD);
int    lart_host_freq_grid(lart_host_freq *f);                              /* car_setup_freq_grid + the core-skip scale (grid_mod_car.f90:1150-1548) */
int    lart_host_freq_axis(int32_t n, double xfreq_min, double dxfreq, double vtherm, double wavelength0_um, double *xfreq, double *velocity, double *wavelength);
int    lart_host_jpa_car(int32_t geometry_JPa, int32_t xyz_symmetry, int32_t nx, int32_t ny, int32_t nz, const double *xface, const double *yface, const double *zface,
                         double xmax, double ymax, double zmax, const double *rhokap, int32_t *nbins, int32_t *ind_bin, double *ncount);      /* create_JPa_mem :1242-1434 */
/* observer_create (observer_rect.f90:24-326, observer_heal.f90:24-80): arrays of LART_MAX_OBSERVERS entries, NaN = not given, completed on return;
 * xyz[3 nobs], rmatrix[9 nobs] row-major */
int    lart_host_observers(lart_host_obs *o, double *alpha, double *beta, double *gamma, double *obsx, double *obsy, double *obsz, double *xyz, double *rmatrix);
int    lart_host_clump_population(lart_host_clumps_in *c);                  /* init_clumps: population size and clump opacity (clump_mod.f90:646-760) */
int    lart_host_clump_scalars(int64_t n, double sphere_R, double r_min, double cross0, const double *x, const double *y, const double *z, const double *radius,
                               const double *rhokap, const double *voigt_a, const double *Dfreq, double *out4);      /* compute_clump_scalars :2316-2378 */
int    lart_host_clump_velocities(int64_t n, int32_t velocity_type, double sphere_R, double Vexp, double Vx, double Vy, double Vz, double Vrot, double rinner,
                                  const double *x, const double *y, const double *z, const double *vtherm, double *vx, double *vy, double *vz);      /* :1153-1264 */
int    lart_host_alias_setup(int32_t n, double *probs, int32_t *alias);     /* random_alias_setup (random_mt.f90:2016-2067) */
int    lart_host_mueller_setup(int32_t n, const double *coss, double *S11, double *S12, double *S33, double *S34, double *phase_PDF, int32_t *alias);   /* setup.f90:615-649 */
int    lart_host_norm_scales(lart_host_norm *s);                            /* the divisors of output_normalize_outside (output_sum_rect.f90:151-487) */

/* device-function probes used by the parity tests (each runs a tiny kernel, one lane per element) */
int  lart_gpu_test_voigt(lart_gpu *ctx, int32_t n, const double *x, const double *a, double *h);
int  lart_gpu_test_rand(lart_gpu *ctx, int32_t what, uint64_t seed, uint64_t id, int32_t n, double p1, double p2, double *out);
int  lart_gpu_test_raytrace(lart_gpu *ctx, int32_t mode, int32_t n, double *r8, int32_t *ic4, const double *tau_or_dist, double *tau_out);
int  lart_gpu_test_vec2pix(lart_gpu *ctx, int32_t nside, int32_t n, const double *v3, int32_t *ipix);
int  lart_gpu_test_scatter(lart_gpu *ctx, int32_t n, double *st18, int32_t *ic4, uint64_t seed, const uint64_t *ids, int64_t *ndraws);
/* ONE event round of the photon loop (run_simulation_mod.f90:160-200: optical depth draw [forced on the first flight], raytrace_to_tau, scattering) from a
 * given packet state and draw counter; st27 per packet, in and out: x y z kx ky kz xfreq wgt m(3) n(3) Q U V nscatt_gas nscatt_dust draws icell jcell kcell
 * owner_clump vfy_shear iband kind (out: 1 scattered, 2 left or absorbed).  The one-step re-synchronised check of tests/test_gpu_divergence.py feeds the
 * oracle's state of the event before a divergence in here.  Plain grid and clump media, line type 1. */
int  lart_gpu_test_event(lart_gpu *ctx, int32_t n, double *st27, uint64_t seed, const uint64_t *ids, const int32_t *first);
/* per-event history probe (tests/test_gpu_divergence.py): runs the photons first_id + k*stride, k < count, like lart_gpu_run and records, for
 * every k with slot[k] >= 0, 8 doubles after the birth and after every event into trace[(slot*cap + i)*8 ..]: draws consumed, xfreq
 * (xfreq_ref once outside), x, y, z, cell code (icell + 2^21 jcell + 2^42 kcell; the owner clump in clump media), kz, kind (0 birth,
 * 1 scattered, 2 left or absorbed); n_out[slot] = records produced (those beyond cap are counted, not stored) */
int  lart_gpu_test_trace(lart_gpu *ctx, int64_t first_id, int64_t stride, int64_t count, uint64_t seed, const int32_t *slot, int32_t nslots,
                         int32_t cap, double *trace, int32_t *n_out);

#ifdef __cplusplus
}
#endif
#endif
