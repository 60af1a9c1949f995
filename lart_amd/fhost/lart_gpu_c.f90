!------------------------------------------------------------------------------
! lart_gpu_c -- ISO_C_BINDING view of include/lart_gpu.h (liblart_gpu.so).
!
! This is the whole Fortran<->HIP boundary: one derived type per C struct and one
! interface per entry point.  A maintainer of the reference adds this module and
! replaces `call run_simulation(grid)` / `call output_reduce(grid)` in main.f90:42,46
! by lart_gpu_run / lart_gpu_get_tallies (see INTEGRATION.md).
!------------------------------------------------------------------------------
module lart_gpu_c
  use, intrinsic :: iso_c_binding
  implicit none
  public

  !--- struct lart_params (field names = par% / grid% / line% names of the reference)
  type, bind(C) :: lart_params
     integer(c_int32_t) :: xyz_symmetry, xy_periodic, xy_symmetry
     integer(c_int32_t) :: source_geometry, spectral_type, comoving_source
     integer(c_int32_t) :: recoil, core_skip, core_skip_global, use_stokes, use_reduced_wgt
     integer(c_int32_t) :: save_Jin, save_Jabs, save_Jmu, nmu
     integer(c_int32_t) :: save_peeloff, save_peeloff_2D, save_peeloff_3D
     integer(c_int32_t) :: save_direc0, save_dust_scattered
     integer(c_int32_t) :: nxfreq
     integer(c_int32_t) :: line_type
     real(c_double) :: xs_point, ys_point, zs_point, source_rmax, xfreq0
     real(c_double) :: xfreq_min, dxfreq, Dfreq_ref
     real(c_double) :: xfreq_max, source_sigma_x
     real(c_double) :: xrange, yrange, zrange, source_zscale
     real(c_double) :: DGR, hgg, albedo
     real(c_double) :: mu_min, dmu
     real(c_double) :: g_recoil0, E1, E2, E3
     real(c_double) :: DnuHK_Hz
     real(c_double) :: delta_nu_HD_Hz, ratio_Dfreq_HD, ratio_voigta_HD, nD_over_nH, g_recoil0_D
     real(c_double) :: xcrit, xcrit2
     real(c_double) :: Omega
     integer(c_int32_t) :: nxfreq_Ha = 0, h2_nlines = 0
     real(c_double) :: xfreq_min_Ha = 0, dxfreq_Ha = 0, R_Ha = 0, albedo_Ha = 0, hgg_Ha = 0, P_conv = 0, E_conv(3) = 0
     integer(c_int32_t) :: h2_hi_width = 0, pad_ = 0
     real(c_double) :: h2_Dfreq_Hz = 0, h2_dnu_Hz(4) = 0, h2_strength(4) = 0, h2_a_damp(4) = 0, h2_p_scat(4) = 0
  end type lart_params

  !--- struct lart_tallies: host pointers (c_null_ptr = skip) + scalars
  !--- lart_line_multiplet of include/lart_gpu.h (line types 4, 5, 6: line%b(:) of the reference); C [3][4] arrays = Fortran (4,3)
  type, bind(C) :: lart_line_multiplet
     integer(c_int32_t) :: line_type, nup, HeI_coherent
     integer(c_int32_t) :: ndown(3)
     integer(c_int32_t) :: A_down(4,3)
     character(kind=c_char) :: ion_id(8)
     real(c_double) :: wavelength0, cross0, vtherm1, g_recoil0, damping
     real(c_double) :: f12(3), delE_Hz(3), damping_up(3)
     real(c_double) :: A21(4,3), Elow_Hz(4,3), E1(4,3), E2(4,3), E3(4,3), P_down(4,3)
  end type lart_line_multiplet

  type, bind(C) :: lart_tallies
     type(c_ptr) :: Jout, Jin, Jabs, Jmu
     type(c_ptr) :: scatt, direc, I, Q, U, V
     type(c_ptr) :: scatt_2D, direc_2D, I_2D, Q_2D, U_2D, V_2D
     type(c_ptr) :: direc0, direc0_2D, scatt_dust
     type(c_ptr) :: J1, P1, P1_new
     real(c_double) :: nscatt_gas, nscatt_dust, n_events, n_cellsteps, n_cellsteps_peel, n_atomics
     integer(c_int64_t) :: nphotons_done, nphotons_active
     type(c_ptr) :: Jout_Ha = c_null_ptr, Jabs_Ha = c_null_ptr, peel_Ha = c_null_ptr, Pconv = c_null_ptr
     real(c_double) :: aux(8)
  end type lart_tallies

  !--- host set-up helpers of the library (include/lart_gpu.h, "Host set-up helpers"; csrc/lart_host_setup.cpp): in/out records, member for member
  integer(c_int32_t), parameter :: LART_VEL_NONE = 0, LART_VEL_HUBBLE = 1, LART_VEL_CONSTANT_RADIAL = 2, LART_VEL_PARALLEL = 3, LART_VEL_ROT_SOLID = 4, &
                                   LART_VEL_ROT_HALO = 5, LART_VEL_POWER_LAW = 6, LART_VEL_LINEAR_DECEL = 7, LART_VEL_SSH = 8
  integer(c_int32_t), parameter :: LART_UNIT_NONE = 0, LART_UNIT_KPC = 1, LART_UNIT_PC = 2, LART_UNIT_AU = 3, LART_UNIT_OTHER = 4, LART_UNIT_USER = 5, &
                                   LART_NOTE_NO_PEEL_WITH_XYZSYM = 1
  type, bind(C) :: lart_host_rules
     integer(c_int32_t) :: nx, ny, nz, nr, nside, nxim, nyim, nmu, geometry_JPa, distance_unit, intensity_unit, notes
     integer(c_int32_t) :: xyz_symmetry, xy_symmetry, xy_periodic, use_clump_medium, use_amr_grid, use_stokes, sphere, save_all
     integer(c_int32_t) :: save_peeloff, save_peeloff_2D, save_peeloff_3D, save_Jabs, save_Jmu, core_skip, core_skip_global, pad_ = 0
     real(c_double) :: tau0, taumax, N_HI, N_HImax, N_HIhomo, N_gasmax, N_gashomo, cext_dust, DGR, mu_min, dmu
     real(c_double) :: rmax, xmax, ymax, zmax, source_rmax, distance2cm
  end type lart_host_rules
  type, bind(C) :: lart_host_line
     integer(c_int32_t) :: line_type, fine_structure, include_deuterium, pad_
     real(c_double) :: wavelength0, damping, cross0, vtherm1, mass_amu, g_recoil0, DnuHK_Hz, E1, E2, E3
     real(c_double) :: delta_nu_HD_Hz, ratio_Dfreq_HD, ratio_voigta_HD, nD_over_nH, g_recoil0_D
     real(c_double) :: P_conv, E_conv(3), wavelength0_Ha
     character(kind=c_char) :: line_id(32), ion_id(8)
  end type lart_host_line
  type, bind(C) :: lart_host_h2
     integer(c_int32_t) :: n, pad_
     real(c_double) :: Dfreq_Hz, dv_kms(4), dnu_Hz(4), strength(4), a_damp(4), p_scat(4), pop(4), f_osc(4)
  end type lart_host_h2
  type, bind(C) :: lart_host_medium
     integer(c_int32_t) :: nx, ny, nz, xyz_symmetry, xy_symmetry, velocity_type, pad0_ = 0, pad1_ = 0
     real(c_double) :: xmax, ymax, zmax, rmin, rmax, cone_opening
     real(c_double) :: vtherm, wavelength0, damping, cross0, dust_opacity
     real(c_double) :: taumax, tauhomo, N_gasmax, N_gashomo
     real(c_double) :: Vexp, Vx, Vy, Vz, Vrot, rinner, velocity_alpha, rpeak, Vpeak, DeltaV
     real(c_double) :: dx, dy, dz, xmin, ymin, zmin, Dfreq_ref, voigt_amean, taupole_out, tauhomo_out, N_gaspole_out, N_gashomo_out
     integer(c_int32_t) :: i0, j0, k0, pad2_ = 0
  end type lart_host_medium
  type, bind(C) :: lart_host_freq
     integer(c_int32_t) :: nxfreq, nwavelength, nvelocity, continuum, line_type, nxfreq_Ha, core_skip_global, pad_ = 0
     real(c_double) :: wavelength_min, wavelength_max, velocity_min, velocity_max, xfreq_min, xfreq_max
     real(c_double) :: vtherm, wavelength0, wavelength0_Ha, Dfreq_ref, DnuHK_Hz, voigt_amean, tauhomo, taumax, Vexp, xfreq_max_Ha, cells_per_radius
     real(c_double) :: dxfreq, dwave, atau3, xfreq_min_Ha, dxfreq_Ha, dwave_Ha, xcrit, xcrit2
  end type lart_host_freq
  type, bind(C) :: lart_host_obs
     integer(c_int32_t) :: nside, sphere, nxim, nyim, nobs, npix
     real(c_double) :: xmax, ymax, zmax, rmax, distance, dxim, dyim, steradian_pix
  end type lart_host_obs
  type, bind(C) :: lart_host_clumps_in
     real(c_double) :: sphere_R, rmin, clump_radius, N_clumps_in, f_vol_in, f_cov_in, clump_tau0, clump_NHI, clump_nH, taumax, N_gasmax, distance2cm
     real(c_double) :: vtherm, wavelength0, damping, cross0
     integer(c_int64_t) :: N
     real(c_double) :: r_min, fvol, fcov, Dfreq_ref, voigt_a_ref, rhokap_ref
  end type lart_host_clumps_in
  type, bind(C) :: lart_host_norm
     integer(c_int32_t) :: xy_periodic, sphere
     real(c_double) :: nphotons, bin_unit, bin_unit_Ha, distance2cm, rmax, xmax, ymax, zmax, amr_L_box, steradian_pix
     real(c_double) :: area, scale_J, scale_J_Ha, scale_cube, scale_image, scale_cube_Ha
  end type lart_host_norm

  interface
     function lart_gpu_device_count() bind(C, name='lart_gpu_device_count') result(n)
       import; integer(c_int) :: n
     end function
     function lart_gpu_create(ctx, par, device) bind(C, name='lart_gpu_create') result(rc)
       import; type(c_ptr), intent(out) :: ctx; type(lart_params), intent(in) :: par
       integer(c_int), value :: device; integer(c_int) :: rc
     end function
     function lart_gpu_destroy(ctx) bind(C, name='lart_gpu_destroy') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int) :: rc
     end function
     function lart_gpu_last_error() bind(C, name='lart_gpu_last_error') result(p)
       import; type(c_ptr) :: p
     end function
     function lart_gpu_set_grid_car(ctx, nx, ny, nz, xmin, ymin, zmin, dx, dy, dz, i0, j0, k0, xface, yface, zface, &
                                    rhokap, voigt_a, Dfreq, vfx, vfy, vfz, rhokapD) bind(C, name='lart_gpu_set_grid_car') result(rc)
       import; type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nx, ny, nz, i0, j0, k0
       real(c_double), value :: xmin, ymin, zmin, dx, dy, dz
       real(c_double), intent(in) :: xface(*), yface(*), zface(*), rhokap(*), voigt_a(*), Dfreq(*), vfx(*), vfy(*), vfz(*)
       type(c_ptr), value :: rhokapD          ! c_loc(array) or c_null_ptr when DGR = 0
       integer(c_int) :: rc
     end function
     function lart_host_line_lookup(line_id, m) bind(C, name='lart_host_line_lookup') result(rc)
       import; character(kind=c_char), intent(in) :: line_id(*); type(lart_line_multiplet), intent(out) :: m; integer(c_int) :: rc
     end function
     function lart_gpu_set_line_multiplet(ctx, m) bind(C, name='lart_gpu_set_line_multiplet') result(rc)
       import; type(c_ptr), value :: ctx; type(lart_line_multiplet), intent(in) :: m; integer(c_int) :: rc
     end function
     function lart_gpu_set_mueller(ctx, nPDF, coss, S11, S12, S33, S34, phase_PDF, alias) bind(C, name='lart_gpu_set_mueller') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: nPDF
       real(c_double), intent(in) :: coss(*), S11(*), S12(*), S33(*), S34(*), phase_PDF(*)
       integer(c_int32_t), intent(in) :: alias(*); integer(c_int) :: rc
     end function
     function lart_gpu_set_observers(ctx, nobs, xyz, rmatrix, dxim, dyim, nxim, nyim) bind(C, name='lart_gpu_set_observers') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: nobs, nxim, nyim
       real(c_double), intent(in) :: xyz(*), rmatrix(*); real(c_double), value :: dxim, dyim; integer(c_int) :: rc
     end function
     function lart_gpu_set_observers_heal(ctx, nobs, xyz, nside) bind(C, name='lart_gpu_set_observers_heal') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: nobs, nside
       real(c_double), intent(in) :: xyz(*); integer(c_int) :: rc
     end function
     function lart_gpu_set_JPa(ctx, calc_J, calc_P, calc_Pnew, nbins, ind_bin, cross0) bind(C, name='lart_gpu_set_JPa') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: calc_J, calc_P, calc_Pnew, nbins
       integer(c_int32_t), intent(in) :: ind_bin(*); real(c_double), value :: cross0; integer(c_int) :: rc
     end function
     function lart_gpu_sightline_tau(ctx, cross0, tau_gas, N_gas, tau_dust) bind(C, name='lart_gpu_sightline_tau') result(rc)
       import; type(c_ptr), value :: ctx; real(c_double), value :: cross0
       real(c_double), intent(out) :: tau_gas(*), N_gas(*), tau_dust(*); integer(c_int) :: rc
     end function
     function lart_gpu_set_clumps(ctx, n, x, y, z, vx, vy, vz, radius, rhokap, rhokapD, voigt_a, Dfreq, sphere_R, Dfreq_ref, voigt_a_ref, &
                                  cgx, cg_min, cg_d, cg_start, cg_list) bind(C, name='lart_gpu_set_clumps') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n
       real(c_double), intent(in) :: x(*), y(*), z(*), vx(*), vy(*), vz(*), radius(*), rhokap(*), voigt_a(*), Dfreq(*)
       type(c_ptr), value :: rhokapD          ! c_loc(cl_rhokapD) or c_null_ptr when DGR = 0
       real(c_double), value :: sphere_R, Dfreq_ref, voigt_a_ref, cg_min, cg_d
       integer(c_int32_t), value :: cgx
       integer(c_int32_t), intent(in) :: cg_start(*), cg_list(*)
       integer(c_int) :: rc
     end function
     !--- host-side helpers (CPU): the per-clump loops of generate_clumps / build_clump_csr (clump_mod.f90:897-1151, 1267-1360)
     function lart_host_last_error() bind(C, name='lart_host_last_error') result(p)
       import; type(c_ptr) :: p
     end function
     function lart_host_clumps_generate(n, sphere_R, r_min, clump_radius, fully_inside, sigma_v_over_vtherm, iseed, &
                                        x, y, z, vx, vy, vz, acceptance) bind(C, name='lart_host_clumps_generate') result(rc)
       import; integer(c_int64_t), value :: n, iseed; real(c_double), value :: sphere_R, r_min, clump_radius, sigma_v_over_vtherm
       integer(c_int32_t), value :: fully_inside
       real(c_double), intent(out) :: x(*), y(*), z(*), vx(*), vy(*), vz(*), acceptance
       integer(c_int) :: rc
     end function
     function lart_host_clumps_generate_cone(n, sphere_R, r_min, clump_radius, flags, cone_opening_deg, sigma_v_over_vtherm, iseed, &
                                             x, y, z, vx, vy, vz, acceptance) bind(C, name='lart_host_clumps_generate_cone') result(rc)
       import; integer(c_int64_t), value :: n, iseed; real(c_double), value :: sphere_R, r_min, clump_radius, cone_opening_deg, sigma_v_over_vtherm
       integer(c_int32_t), value :: flags
       real(c_double), intent(out) :: x(*), y(*), z(*), vx(*), vy(*), vz(*), acceptance
       integer(c_int) :: rc
     end function
     function lart_host_clumps_csr_dims(n, sphere_R, radius_max, cgx, cg_min, cg_d) bind(C, name='lart_host_clumps_csr_dims') result(rc)
       import; integer(c_int64_t), value :: n; real(c_double), value :: sphere_R, radius_max
       integer(c_int32_t), intent(out) :: cgx; real(c_double), intent(out) :: cg_min, cg_d
       integer(c_int) :: rc
     end function
     function lart_host_clumps_build_csr(n, x, y, z, radius, cgx, cg_min, cg_d, cg_start, cg_list, nreg) bind(C, name='lart_host_clumps_build_csr') result(rc)
       import; integer(c_int64_t), value :: n; real(c_double), intent(in) :: x(*), y(*), z(*), radius(*)
       integer(c_int32_t), value :: cgx; real(c_double), value :: cg_min, cg_d
       integer(c_int32_t), intent(out) :: cg_start(*)
       type(c_ptr), value :: cg_list          ! c_null_ptr on the counting pass
       integer(c_int64_t), intent(out) :: nreg
       integer(c_int) :: rc
     end function
     function lart_host_clumps_has_overlap(n, x, y, z, radius, cgx, cg_min, cg_d, cg_start, cg_list, has_overlap) bind(C, name='lart_host_clumps_has_overlap') result(rc)
       import; integer(c_int64_t), value :: n; real(c_double), intent(in) :: x(*), y(*), z(*), radius(*)
       integer(c_int32_t), value :: cgx; real(c_double), value :: cg_min, cg_d
       integer(c_int32_t), intent(in) :: cg_start(*), cg_list(*)
       integer(c_int32_t), intent(out) :: has_overlap
       integer(c_int) :: rc
     end function
     function lart_gpu_set_clump_overlap(ctx, has_overlap) bind(C, name='lart_gpu_set_clump_overlap') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: has_overlap; integer(c_int) :: rc
     end function
     function lart_gpu_set_grid_amr(ctx, ncells, nleaf, lo, hi, cx, cy, cz, ch, children, neighbor, ileaf, icell_of_leaf, &
                                    rhokap, voigt_a, Dfreq, vfx, vfy, vfz, rhokapD) bind(C, name='lart_gpu_set_grid_amr') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: ncells, nleaf; real(c_double), value :: lo, hi
       real(c_double), intent(in) :: cx(*), cy(*), cz(*), ch(*), rhokap(*), voigt_a(*), Dfreq(*), vfx(*), vfy(*), vfz(*)
       integer(c_int32_t), intent(in) :: children(*), neighbor(*), ileaf(*), icell_of_leaf(*)
       type(c_ptr), value :: rhokapD
       integer(c_int) :: rc
     end function
     function lart_gpu_set_JPa_amr(ctx, calc_J, calc_P, calc_Pnew, geometry_JPa, nbins, dbin, cross0) bind(C, name='lart_gpu_set_JPa_amr') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: calc_J, calc_P, calc_Pnew, geometry_JPa, nbins
       real(c_double), value :: dbin, cross0; integer(c_int) :: rc
     end function
     function lart_host_amr_last_error() bind(C, name='lart_host_amr_last_error') result(p)
       import; type(c_ptr) :: p
     end function
     function lart_host_amr_count(nleaf, x, y, z, level, lo, hi, ncells) bind(C, name='lart_host_amr_count') result(rc)
       import; integer(c_int32_t), value :: nleaf; real(c_double), intent(in) :: x(*), y(*), z(*); integer(c_int32_t), intent(in) :: level(*)
       real(c_double), value :: lo, hi; integer(c_int32_t), intent(out) :: ncells; integer(c_int) :: rc
     end function
     function lart_host_amr_build(nleaf, x, y, z, level, lo, hi, ncells, cx, cy, cz, ch, children, neighbor, ileaf, icell_of_leaf, &
                                  cell_level, levelmax) bind(C, name='lart_host_amr_build') result(rc)
       import; integer(c_int32_t), value :: nleaf, ncells; real(c_double), intent(in) :: x(*), y(*), z(*); integer(c_int32_t), intent(in) :: level(*)
       real(c_double), value :: lo, hi
       real(c_double), intent(out) :: cx(*), cy(*), cz(*), ch(*)
       integer(c_int32_t), intent(out) :: children(*), neighbor(*), ileaf(*), icell_of_leaf(*), levelmax
       type(c_ptr), value :: cell_level
       integer(c_int) :: rc
     end function
     function lart_gpu_enable_all_photons(ctx, on) bind(C, name='lart_gpu_enable_all_photons') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: on; integer(c_int) :: rc
     end function
     function lart_gpu_get_all_photons(ctx, out, n) bind(C, name='lart_gpu_get_all_photons') result(rc)
       import; type(c_ptr), value :: ctx; real(c_double), intent(out) :: out(*); integer(c_int64_t), value :: n; integer(c_int) :: rc
     end function
     function lart_gpu_run(ctx, first_id, stride, count, seed) bind(C, name='lart_gpu_run') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: first_id, stride, count, seed; integer(c_int) :: rc
     end function
     function lart_gpu_begin(ctx, first_id, stride, count, seed) bind(C, name='lart_gpu_begin') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: first_id, stride, count, seed; integer(c_int) :: rc
     end function
     function lart_gpu_step(ctx, nsteps, max_events) bind(C, name='lart_gpu_step') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: nsteps, max_events; integer(c_int) :: rc
     end function
     function lart_gpu_sync(ctx, nphotons_done, nphotons_active) bind(C, name='lart_gpu_sync') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int64_t), intent(out) :: nphotons_done, nphotons_active; integer(c_int) :: rc
     end function
     !--- output_reduce over the GPUs of this process / over ranks (include/lart_gpu.h; reduce_mem, memory_mod_mpi.f90:366-394)
     function lart_gpu_reduce_tallies_multi(ctxs, n, root) bind(C, name='lart_gpu_reduce_tallies_multi') result(rc)
       import; type(c_ptr), intent(in) :: ctxs(*); integer(c_int32_t), value :: n, root; integer(c_int) :: rc
     end function
     function lart_gpu_run_multi(ctxs, n, nphotons, seed, max_events) bind(C, name='lart_gpu_run_multi') result(rc)
       import; type(c_ptr), intent(in) :: ctxs(*); integer(c_int32_t), value :: n, max_events; integer(c_int64_t), value :: nphotons, seed; integer(c_int) :: rc
     end function
     function lart_gpu_comm_unique_id(id128) bind(C, name='lart_gpu_comm_unique_id') result(rc)
       import; integer(c_int8_t), intent(out) :: id128(128); integer(c_int) :: rc
     end function
     function lart_gpu_comm_init(ctx, id128, nranks, rank) bind(C, name='lart_gpu_comm_init') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int8_t), intent(in) :: id128(128); integer(c_int32_t), value :: nranks, rank; integer(c_int) :: rc
     end function
     function lart_gpu_reduce_tallies(ctx, root) bind(C, name='lart_gpu_reduce_tallies') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: root; integer(c_int) :: rc
     end function
     function lart_gpu_compact(ctx) bind(C, name='lart_gpu_compact') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int) :: rc
     end function
     function lart_host_setup_last_error() bind(C, name='lart_host_setup_last_error') result(p)
       import; type(c_ptr) :: p
     end function
     function lart_host_input_rules(r) bind(C, name='lart_host_input_rules') result(rc)
       import; type(lart_host_rules), intent(inout) :: r; integer(c_int) :: rc
     end function
     function lart_host_line_setup(line_id, fine_structure, include_deuterium, D_to_H_ratio, L) bind(C, name='lart_host_line_setup') result(rc)
       import; character(kind=c_char), intent(in) :: line_id(*); integer(c_int32_t), value :: fine_structure, include_deuterium
       real(c_double), value :: D_to_H_ratio; type(lart_host_line), intent(out) :: L; integer(c_int) :: rc
     end function
     function lart_host_h2_init(path, wavelength0, vtherm1, mass_amu, bturb, T_H2, f_H2, pure_absorption, H) bind(C, name='lart_host_h2_init') result(rc)
       import; character(kind=c_char), intent(in) :: path(*); real(c_double), value :: wavelength0, vtherm1, mass_amu, bturb, T_H2, f_H2
       integer(c_int32_t), value :: pure_absorption; type(lart_host_h2), intent(out) :: H; integer(c_int) :: rc
     end function
     function lart_host_car_medium(m, xface, yface, zface, rhokap, voigt_a, Dfreq, vfx, vfy, vfz, rhokapD) bind(C, name='lart_host_car_medium') result(rc)
       import; type(lart_host_medium), intent(inout) :: m; real(c_double), intent(out) :: xface(*), yface(*), zface(*), rhokap(*), voigt_a(*), Dfreq(*), vfx(*), vfy(*), vfz(*)
       type(c_ptr), value :: rhokapD; integer(c_int) :: rc
     end function
     function lart_host_freq_grid(f) bind(C, name='lart_host_freq_grid') result(rc)
       import; type(lart_host_freq), intent(inout) :: f; integer(c_int) :: rc
     end function
     function lart_host_freq_axis(n, xfreq_min, dxfreq, vtherm, wavelength0, xfreq, velocity, wavelength) bind(C, name='lart_host_freq_axis') result(rc)
       import; integer(c_int32_t), value :: n; real(c_double), value :: xfreq_min, dxfreq, vtherm, wavelength0
       real(c_double), intent(out) :: xfreq(*), velocity(*), wavelength(*); integer(c_int) :: rc
     end function
     function lart_host_jpa_car(geometry_JPa, xyz_symmetry, nx, ny, nz, xface, yface, zface, xmax, ymax, zmax, rhokap, nbins, ind_bin, ncount) &
                                bind(C, name='lart_host_jpa_car') result(rc)
       import; integer(c_int32_t), value :: geometry_JPa, xyz_symmetry, nx, ny, nz; real(c_double), intent(in) :: xface(*), yface(*), zface(*), rhokap(*)
       real(c_double), value :: xmax, ymax, zmax; integer(c_int32_t), intent(out) :: nbins, ind_bin(*); real(c_double), intent(out) :: ncount(*); integer(c_int) :: rc
     end function
     function lart_host_observers(o, alpha, beta, gamma, obsx, obsy, obsz, xyz, rmatrix) bind(C, name='lart_host_observers') result(rc)
       import; type(lart_host_obs), intent(inout) :: o; real(c_double), intent(inout) :: alpha(*), beta(*), gamma(*), obsx(*), obsy(*), obsz(*)
       real(c_double), intent(out) :: xyz(*), rmatrix(*); integer(c_int) :: rc
     end function
     function lart_host_clump_population(c) bind(C, name='lart_host_clump_population') result(rc)
       import; type(lart_host_clumps_in), intent(inout) :: c; integer(c_int) :: rc
     end function
     function lart_host_clump_scalars(n, sphere_R, r_min, cross0, x, y, z, radius, rhokap, voigt_a, Dfreq, out4) bind(C, name='lart_host_clump_scalars') result(rc)
       import; integer(c_int64_t), value :: n; real(c_double), value :: sphere_R, r_min, cross0
       real(c_double), intent(in) :: x(*), y(*), z(*), radius(*), rhokap(*), voigt_a(*), Dfreq(*); real(c_double), intent(out) :: out4(4); integer(c_int) :: rc
     end function
     function lart_host_clump_velocities(n, velocity_type, sphere_R, Vexp, Vx, Vy, Vz, Vrot, rinner, x, y, z, vtherm, vx_, vy_, vz_) &
                                         bind(C, name='lart_host_clump_velocities') result(rc)
       import; integer(c_int64_t), value :: n; integer(c_int32_t), value :: velocity_type; real(c_double), value :: sphere_R, Vexp, Vx, Vy, Vz, Vrot, rinner
       real(c_double), intent(in) :: x(*), y(*), z(*), vtherm(*); real(c_double), intent(inout) :: vx_(*), vy_(*), vz_(*); integer(c_int) :: rc
     end function
     function lart_host_mueller_setup(n, coss, S11, S12, S33, S34, phase_PDF, alias) bind(C, name='lart_host_mueller_setup') result(rc)
       import; integer(c_int32_t), value :: n; real(c_double), intent(in) :: coss(*); real(c_double), intent(inout) :: S11(*), S12(*), S33(*), S34(*)
       real(c_double), intent(out) :: phase_PDF(*); integer(c_int32_t), intent(out) :: alias(*); integer(c_int) :: rc
     end function
     function lart_host_norm_scales(s) bind(C, name='lart_host_norm_scales') result(rc)
       import; type(lart_host_norm), intent(inout) :: s; integer(c_int) :: rc
     end function
     function lart_host_entropy_seed(seed) bind(C, name='lart_host_entropy_seed') result(rc)
       import; integer(c_int64_t), intent(out) :: seed; integer(c_int) :: rc
     end function
     function lart_gpu_zero_tallies(ctx) bind(C, name='lart_gpu_zero_tallies') result(rc)
       import; type(c_ptr), value :: ctx; integer(c_int) :: rc
     end function
     function lart_gpu_get_tallies(ctx, t) bind(C, name='lart_gpu_get_tallies') result(rc)
       import; type(c_ptr), value :: ctx; type(lart_tallies), intent(inout) :: t; integer(c_int) :: rc
     end function
  end interface

contains

  !--- error convention of the reference: print the message and stop (setup.f90:132-139)
  subroutine lart_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: n
    if (rc == 0) return
    p = lart_gpu_last_error()
    write(*,'(3a)', advance='no') 'ERROR in ', where, ': '
    if (c_associated(p)) then
       call c_f_pointer(p, msg, [512])
       n = 1
       do while (n <= 512)
          if (msg(n) == c_null_char) exit
          write(*,'(a)', advance='no') msg(n)
          n = n + 1
       enddo
    endif
    write(*,*)
    stop 1
  end subroutine lart_check

end module lart_gpu_c
