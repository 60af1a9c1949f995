!------------------------------------------------------------------------------
! Fortran host of the MI355X photon path: the reference's main.f90 sequence
!   read_input -> setup -> grid_create -> observer_create -> run_simulation -> output_reduce
!   -> output_normalize -> write_output
! with the same namelist surface (&parameters par%... /), the uniform-medium subset of
! grid_create, and run_simulation/output_reduce crossing the C-ABI (module lart_gpu_c).
! Citations: files under /root/reference/src.
!------------------------------------------------------------------------------
module lart_define
  use, intrinsic :: iso_fortran_env, only: real64, int32, int64
  use lart_gpu_c, only: lart_line_multiplet, lart_host_h2
  use, intrinsic :: ieee_arithmetic
  implicit none
  public
  integer, parameter :: wp = real64
  real(wp), parameter :: pi = 3.141592653589793238462643383279502884197_wp
  real(wp), parameter :: twopi = 6.283185307179586476925286766559005768394_wp
  real(wp), parameter :: fourpi = 12.56637061435917295385057353311801153679_wp
  real(wp), parameter :: rad2deg = 180.0_wp/pi, deg2rad = pi/180.0_wp
  real(wp), parameter :: um2m = 1.0e-6_wp, um2km = 1.0e-9_wp
  real(wp), parameter :: speedc = 2.99792458e5_wp, h_planck = 6.62607004e-34_wp
  real(wp), parameter :: nan64 = transfer(-2251799813685248_int64, 1._real64)
  integer,  parameter :: MAX_OBSERVERS = 16

  !--- params_type: the fields of define.f90:209-544 that this path reads, same names and defaults
  !--- (hgg, albedo, cext_dust defaults are single-precision literals widened to f64, define.f90:454-456).
  !--- Fields that select physics outside the path are kept so example inputs parse; read_input rejects
  !--- non-default values with "not supported".
  type params_type
     integer(int64) :: nphotons = 100000_int64
     integer        :: nprint   = 10000000
     real(wp) :: no_photons = 1e5
     real(wp) :: no_print   = 0
     integer  :: iseed      = 0
     real(wp) :: temperature = 1d4
     real(wp) :: taumax = -999.0, tauhomo = -999.0, tau0 = -999.0
     real(wp) :: N_HImax = -999.0, N_HIhomo = -999.0, N_HI = -999.0, N_gasmax = -999.0, N_gashomo = -999.0
     real(wp) :: atau3 = 0.0_wp
     real(wp) :: Vexp = 0.0_wp, Vx = 0.0_wp, Vy = 0.0_wp, Vz = 0.0_wp
     logical  :: comoving_source = .true., recoil = .false., core_skip = .false., core_skip_global = .false.
     logical  :: xyz_symmetry = .false., xy_symmetry = .false., xy_periodic = .false., z_symmetry = .false.
     character(len=128) :: geometry = '', velocity_type = ''
     integer  :: nx = 1, ny = 1, nz = 11, nr = -999
     real(wp) :: xmax = 1.0_wp, ymax = 1.0_wp, zmax = 1.0_wp
     real(wp) :: rmin = -999.0, rmax = -999.0, source_rmax = -999.0_wp
     real(wp) :: xs_point = 0.0_wp, ys_point = 0.0_wp, zs_point = 0.0_wp
     real(wp) :: xfreq0 = 0.0_wp, xfreq_min = nan64, xfreq_max = nan64
     integer  :: nxfreq = 121
     real(wp) :: velocity_min = nan64, velocity_max = nan64
     integer  :: nvelocity = 0
     real(wp) :: distance2cm = -999.9_wp
     character(len=128) :: distance_unit = '', source_geometry = 'point', spectral_type = 'voigt'
     character(len=128) :: base_name = '', out_file = ''
     logical  :: use_master_slave = .true., use_reduced_wgt = .false., use_stokes = .false.
     logical  :: save_all = .false., save_Jin = .true., save_Jabs = .true., save_Jmu = .false.
     integer  :: nmu = 11
     real(wp) :: mu_min = -1.0_wp, dmu = 0.0_wp
     logical  :: save_peeloff = .false., save_peeloff_2D = .false., save_peeloff_3D = .true.
     logical  :: save_sightline_tau = .false., save_direc0 = .false., save_all_photons = .false., save_radial_profile = .false., save_dust_scattered = .false.
     real(wp) :: hgg = 0.6761, albedo = 0.3253, cext_dust = 1.6059e-21, DGR = 0.0
     real(wp) :: nscatt_gas = 0.0_wp, nscatt_dust = 0.0_wp, nscatt_tot = 0.0_wp, exetime = 0.0_wp
     integer  :: nobs = 1, nxim = 0, nyim = 0
     real(wp) :: distance = nan64
     real(wp) :: alpha(MAX_OBSERVERS) = nan64, beta(MAX_OBSERVERS) = nan64, gamma(MAX_OBSERVERS) = nan64
     real(wp) :: obsx(MAX_OBSERVERS) = nan64, obsy(MAX_OBSERVERS) = nan64, obsz(MAX_OBSERVERS) = nan64
     real(wp) :: dxim = nan64, dyim = nan64
     integer  :: nside = 0, npix = 0
     !--- outside this path (accepted at default only)
     logical  :: fine_structure = .false., include_deuterium = .false.
     !--- octree medium (generic text leaf table, read_generic_amr.f90:112-185) and the run-time form of CALCJ / CALCP / CALCPnew
     logical  :: use_amr_grid = .false., calc_J = .false., calc_P = .false., calc_Pnew = .false.
     character(len=16) :: amr_type = 'generic'
     !--- clump medium (define.f90:325-352): uniform clump properties, non-overlapping clumps
     logical  :: use_clump_medium = .false., clump_fully_inside = .true., save_clump_info = .false.
     real(wp) :: clump_radius = -1.0_wp, clump_N_clumps = -1.0_wp, clump_f_vol = -1.0_wp, clump_f_cov = -1.0_wp
     real(wp) :: clump_tau0 = -1.0_wp, clump_NHI = -1.0_wp, clump_nH = -1.0_wp, clump_temperature = -1.0_wp, clump_sigma_v = 0.0_wp
     real(wp) :: Vrot = 0.0_wp, rinner = 0.0_wp, Vpeak = 0.0_wp, rpeak = 0.0_wp, DeltaV = 0.0_wp, velocity_alpha = 1.0_wp
     real(wp) :: gaussian_sigma_vel = 12.843374_wp, gaussian_FWHM_vel = -999.0_wp, source_zscale = 0.0_wp
     character(len=8)   :: h2_model = 'none'
     character(len=128) :: dens_file = '', temp_file = '', velo_file = '', emiss_file = '', cart_file = '', input_field = '', star_file = ''
     character(len=256) :: amr_file = '', clump_input_file = ''
     real(wp) :: bturb = -999.0
     !--- GPU host additions (not in the reference): number of GPUs driven by this process, slice length
     integer  :: ngpu = 0                        ! GPUs of this node to use: 0 = all visible HIP devices (the role of mpirun -np N)
     integer(int64) :: run_seed = 0              ! the Philox key of the run (= iseed, or the operating-system seed when iseed = 0)
     integer  :: gpu_events_per_step = 4096
       !--- the rest of the reference's namelist surface (define.f90:209-544): readable, rejected when set (check_unsupported_fields)
     real(wp) :: luminosity = 1.0_wp
     real(wp) :: temperature0 = -999.0_wp
     real(wp) :: Dfreq0 = -999.0_wp
     real(wp) :: voigt_a0 = -999.0_wp
     real(wp) :: D_to_H_ratio = 1.5e-5_wp
     real(wp) :: source_rscale = 0.0_wp
     real(wp) :: sersic_m = 1.0_wp
     real(wp) :: Reff = 0.0_wp
     real(wp) :: xmin = nan64
     real(wp) :: ymin = nan64
     real(wp) :: zmin = nan64
     real(wp) :: density_rscale = -999.9_wp
     real(wp) :: density_zscale = -999.9_wp
     real(wp) :: density_alpha = 0.0_wp, cone_opening = 0.0_wp, wavelength_min = nan64, wavelength_max = nan64
     integer :: nwavelength = 0
     real(wp) :: q = 1.0_wp
     real(wp) :: Omega = 0.0_wp
     real(wp) :: EW_line = 0.0_wp
     real(wp) :: f_line = 0.0_wp
     character(len=32) :: clump_radius_profile = 'constant', clump_density_profile = 'constant', clump_number_profile = 'constant'
     real(wp) :: clump_radius_alpha = 0.0_wp, clump_radius_r0 = 0.0_wp, clump_density_alpha = 0.0_wp, clump_density_r0 = 0.0_wp, clump_number_alpha = 0.0_wp
     real(wp) :: clump_number_r0 = 0.0_wp, clump_radius_min = -1.0_wp, clump_radius_max_in = -1.0_wp
     character(len=256) :: clump_profile_file = ''
     integer :: amr_snapnum = 0
     character(len=32) :: ionization_model = 'cie_formula', dust_model = 'global_dgr', emissivity_model = 'none', ion_model = 'none'
     real(wp) :: metallicity_global = -1.0_wp
     real(wp) :: Z_ref = 0.0134_wp
     real(wp) :: f_ion_dust = 0.01_wp
     integer :: out_bitpix = 0, sampling_method = 1
     real(wp) :: f_composite = 0.5_wp
     logical  :: save_backup = .false., save_input_grid = .false., use_cie_condition = .false.
     integer :: nxfreq_Ha = 0
     real(wp) :: xfreq_max_Ha = 0.0_wp
     integer :: ny_2gam = 101
     real(wp) :: cext_dust_Ha = 3.801e-22_wp
     real(wp) :: albedo_Ha = 0.6741_wp
     real(wp) :: hgg_Ha = 0.4967_wp
     logical :: transport_2gamma = .false.
     real(wp) :: R_Ha = 0.0_wp
     real(wp) :: W_conv = 0.0_wp
     real(wp) :: W_esc1 = 0.0_wp
     real(wp) :: W_abs1 = 0.0_wp
     real(wp) :: W_esc2 = 0.0_wp
     real(wp) :: W_abs2 = 0.0_wp
     real(wp) :: f_H2 = 0.0_wp
     real(wp) :: h2_temperature = 1000.0_wp
     logical :: h2_pure_absorption = .false.
     logical :: h2_hi_width = .false.
     character(len=128) :: h2_data_dir = ''
     real(wp) :: W_H2abs = 0.0_wp
     real(wp) :: W_H2scat = 0.0_wp
     integer  :: line_prof_file_type = 0, reduce_factor = 1, centering = 0
     real(wp) :: nrejected = 0.0_wp
     real(wp) :: acceptance_rate = 1.0_wp
     real(wp) :: distance_star_to_planet = 0.0_wp
     real(wp) :: stellar_radius = 0.0_wp
     real(wp) :: flux_factor = 0.0_wp
     real(wp) :: inclination_angle(MAX_OBSERVERS) = nan64
     real(wp) :: position_angle(MAX_OBSERVERS) = nan64
     real(wp) :: phase_angle(MAX_OBSERVERS) = nan64
     real(wp) :: rotation_center_x = nan64
     real(wp) :: rotation_center_y = nan64
     real(wp) :: rotation_center_z = nan64
     !--- (fields declared one per line in the reference, gathered by type)
     logical :: out_merge = .false., HeI_coherent = .false., continuum_normalize = .true., clump_allow_overlap = .false., observer_located_inside = .false.
     character(len=8) :: file_format = 'hdf5';  integer(int64) :: num_send_at_once = 100
     integer :: intensity_unit = -999, geometry_JPa = huge(1), stellar_limb_darkening = 2
     character(len=128) :: scatt_mat_file = '', line_prof_file = ''
     character(len=20) :: line_id = 'ly_alpha'
  end type params_type

  type line_type          ! define.f90:639-680, Ly-alpha values of line_mod.f90:1241-1270
     real(wp) :: wavelength0 = 0.1215668237310_wp
     real(wp) :: damping     = 6.2649e8_wp
     real(wp) :: cross0, vtherm1, g_recoil0
     real(wp) :: E1 = 1.0_wp, E2 = 0.0_wp, E3 = 1.0_wp
     integer  :: line_type = 1
     type(lart_line_multiplet) :: ml        ! line types 4, 5, 6: the record of lart_host_line_lookup (line%b(:) of the reference)
     character(len=8) :: ion_id = 'H  I'    ! label of the log lines (grid_mod_car.f90:1231-1234)
     real(wp) :: DnuHK_Hz = 0.0_wp          ! 2P3/2 - 2P1/2 splitting, line type 2 (par%fine_structure)
     real(wp) :: delta_nu_HD_Hz = 0.0_wp, ratio_Dfreq_HD = 1.0_wp, ratio_voigta_HD = 1.0_wp, nD_over_nH = 0.0_wp, g_recoil0_D = 0.0_wp   ! line type 7
     real(wp) :: mass_amu = 1.00797_wp, P_conv = 0.0_wp, E_conv(3) = [1.0_wp, 0.0_wp, 1.0_wp], wavelength0_Ha = 0.6564553_wp                 ! line type 8 (ly_beta)
     type(lart_host_h2) :: h2                ! the H2 lines next to Ly-alpha (h2_init), h2%n = 0: off
  end type line_type

  type grid_type          ! define.f90:117-206 (tables are contiguous, Fortran order = what the C-ABI takes)
     integer  :: nx, ny, nz, nxfreq, i0 = 0, j0 = 0, k0 = 0
     real(wp) :: rmin, rmax, xmin, xmax, ymin, ymax, zmin, zmax, xrange, yrange, zrange, dx, dy, dz
     real(wp) :: xfreq_min, xfreq_max, dxfreq, dwave, Dfreq_ref, voigt_amean, xcrit = 0.0_wp, xcrit2 = 0.0_wp
     real(wp), allocatable :: xfreq(:), velocity(:), wavelength(:), xface(:), yface(:), zface(:)
     real(wp), allocatable :: vfx(:,:,:), vfy(:,:,:), vfz(:,:,:), Dfreq(:,:,:), voigt_a(:,:,:), rhokap(:,:,:)
     real(wp), allocatable :: rhokapD(:,:,:)
     real(wp), allocatable :: Jin(:), Jout(:), Jabs(:), Jmu(:,:)
     integer  :: nxfreq_Ha = 0                                              ! band 2 (H-alpha) of ly_beta: grid_mod_car.f90:1518-1529
     real(wp) :: xfreq_min_Ha = 0.0_wp, xfreq_max_Ha = 0.0_wp, dxfreq_Ha = 0.0_wp, dwave_Ha = 0.0_wp
     real(wp), allocatable :: Jout_Ha(:), Jabs_Ha(:)
  end type grid_type

  type observer_type      ! define.f90:547-600
     real(wp) :: x, y, z, rmatrix(3,3), dxim, dyim, steradian_pix
     integer  :: nxim, nyim
  end type observer_type

  type scattering_matrix_type   ! define.f90:616-625
     integer :: nPDF = 0
     real(wp), allocatable :: coss(:), S11(:), S12(:), S33(:), S34(:), phase_PDF(:)
     integer(int32), allocatable :: alias(:)
  end type scattering_matrix_type

  type(params_type) :: par
  type(line_type)   :: line
  type(scattering_matrix_type) :: scatt_mat
  type(observer_type), allocatable :: observer(:)
  !--- peel-off tallies of all observers: (nxfreq,nxim,nyim,nobs) and (nxim,nyim,nobs)
  real(wp), allocatable, target :: obs_scatt(:,:,:,:), obs_direc(:,:,:,:), obs_I(:,:,:,:), obs_Q(:,:,:,:), obs_U(:,:,:,:), obs_V(:,:,:,:)
  real(wp), allocatable, target :: obs_scatt_2D(:,:,:), obs_direc_2D(:,:,:), obs_I_2D(:,:,:), obs_Q_2D(:,:,:), obs_U_2D(:,:,:), obs_V_2D(:,:,:)
  real(wp), allocatable, target :: obs_direc0(:,:,:,:), obs_direc0_2D(:,:,:), obs_scatt_dust(:,:,:,:)      ! par%save_direc0, par%save_dust_scattered
  real(wp), allocatable, target :: obs_peel_Ha(:,:,:,:), prof_Pconv(:)                                      ! ly_beta: observer(:)%peel_Ha, the conversion-rate profile
  !--- octree (amr_grid_type, octree_mod.f90:19-138), filled by grid_create_amr
  integer(int32) :: amr_ncells = 0, amr_nleaf = 0, amr_levelmax = 0, amr_nbins = 0
  real(wp) :: amr_lo = 0.0_wp, amr_hi = 0.0_wp, amr_dbin = 0.0_wp
  real(wp), allocatable, target :: amr_cx(:), amr_cy(:), amr_cz(:), amr_ch(:)
  integer(int32), allocatable, target :: amr_children(:,:), amr_neighbor(:,:), amr_ileaf(:), amr_icell_of_leaf(:)
  real(wp), allocatable, target :: amr_rhokap(:), amr_rhokapD(:), amr_Dfreq(:), amr_voigt_a(:), amr_vfx(:), amr_vfy(:), amr_vfz(:)
  real(wp), allocatable, target :: amr_vol(:), prof_J1(:,:), prof_P1(:), prof_P1_new(:)
  !--- Cartesian J(x)/P_alpha profile bins (create_JPa_mem, grid_mod_car.f90:1242-1434): bin of each cell, cells per bin
  integer(int32) :: car_nbins = 0
  integer(int32), allocatable, target :: car_ind_bin(:,:,:)
  real(wp), allocatable :: car_ncount(:)
  !--- clump population (clump_mod.f90:27-118), filled by grid_create_clump
  integer(int64) :: N_clumps = 0_int64
  real(wp) :: sphere_R = 0.0_wp, r_min_clump = 0.0_wp, cl_radius_max = 0.0_wp
  real(wp) :: cl_rhokap_ref = 0.0_wp, cl_voigt_a_ref = 0.0_wp, cl_Dfreq_ref = 0.0_wp, cl_vtherm_ref = 0.0_wp
  real(wp), allocatable, target :: cl_x(:), cl_y(:), cl_z(:), cl_vx(:), cl_vy(:), cl_vz(:)
  real(wp), allocatable, target :: cl_radius(:), cl_rhokap(:), cl_rhokapD(:), cl_voigt_a(:), cl_Dfreq(:), cl_vtherm(:), cl_temperature(:)
  real(wp) :: cl_temperature_ref = 0.0_wp
  logical :: clumps_from_file = .false.          ! clump_mod.f90:86-87
  integer(int32) :: cgx = 0
  real(wp) :: cg_xmin = 0.0_wp, cg_dx = 0.0_wp
  integer(int32), allocatable, target :: cg_start(:), cg_list(:)
  logical :: has_overlap = .false.                ! clump_mod.f90:88

  interface say                       ! one labelled value per log line, in the reference's formats (the labels are what tools parse)
     module procedure say_r, say_i8, say_i
  end interface say

contains
  subroutine say_r(label, v, fmt)
    character(len=*), intent(in) :: label, fmt
    real(wp), intent(in) :: v
    write(*,'(a,'//fmt//')') label, v
  end subroutine say_r
  subroutine say_i8(label, v, fmt)
    character(len=*), intent(in) :: label, fmt
    integer(int64), intent(in) :: v
    write(*,'(a,'//fmt//')') label, v
  end subroutine say_i8
  subroutine say_i(label, v, fmt)
    character(len=*), intent(in) :: label, fmt
    integer, intent(in) :: v
    write(*,'(a,'//fmt//')') label, v
  end subroutine say_i

  elemental function is_finite(x) result(r)
    real(wp), intent(in) :: x
    logical :: r
    r = ieee_is_finite(x)
  end function is_finite

  elemental function vtherm_total(T) result(vt)       ! define.f90:928-933
    real(wp), intent(in) :: T
    real(wp) :: vt
    vt = line%vtherm1 * sqrt(T)
    if (par%bturb > 0.0_wp) vt = sqrt(vt*vt + par%bturb*par%bturb)
  end function vtherm_total

  function strlowcase(s) result(t)
    character(len=*), intent(in) :: s
    character(len=len(s)) :: t
    integer :: i, c
    t = s
    do i = 1, len(s)
       c = iachar(s(i:i))
       if (c >= 65 .and. c <= 90) t(i:i) = achar(c + 32)
    enddo
  end function strlowcase
end module lart_define

!==============================================================================
module lart_host
  use, intrinsic :: iso_c_binding
  use lart_define
  use lart_gpu_c
  implicit none
contains

  !--- message of the host helpers of the library, then stop (reference convention: print and abort)
  subroutine setup_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: n
    if (rc == 0) return
    p = lart_host_setup_last_error()
    call c_f_pointer(p, msg, [1024])
    n = 0
    do while (n < 1024)
       if (msg(n+1) == c_null_char) exit
       n = n + 1
    enddo
    write(*,'(64a)') msg(1:min(n,1024))
    write(*,'(2a)') '   in ', where
    stop 1
  end subroutine setup_check

  pure function cstring(c) result(t)
    character(kind=c_char), intent(in) :: c(:)
    character(len=size(c)) :: t
    integer :: i
    t = ''
    do i = 1, size(c)
       if (c(i) == c_null_char) exit
       t(i:i) = c(i)
    enddo
  end function cstring

  !--- setup_resonance_line (line_mod.f90:551-1272): the constants of par%line_id from the library -- lart_host_line_setup for 'ly_alpha' (with
  !--- par%fine_structure / par%include_deuterium), 'ly_alpha_HD', 'ly_beta' and the single lines / doublets, lart_host_line_lookup for the multiplets
  subroutine setup_resonance_line()
    type(lart_host_line) :: h
    integer(c_int) :: rc
    rc = lart_host_line_setup(trim(par%line_id)//c_null_char, merge(1, 0, par%fine_structure), merge(1, 0, par%include_deuterium), par%D_to_H_ratio, h)
    if (rc == 2) then                                                   ! resonance + fluorescence multiplets, line types 4, 5, 6
       rc = lart_host_line_lookup(trim(par%line_id)//c_null_char, line%ml)
       line%line_type = line%ml%line_type; line%wavelength0 = line%ml%wavelength0; line%damping = line%ml%damping; line%cross0 = line%ml%cross0
       line%vtherm1 = line%ml%vtherm1; line%g_recoil0 = line%ml%g_recoil0; line%DnuHK_Hz = 0.0_wp
       line%E1 = 1.0_wp; line%E2 = 0.0_wp; line%E3 = 1.0_wp
       line%ion_id = cstring(line%ml%ion_id)
       line%ml%HeI_coherent = merge(1, 0, par%HeI_coherent .and. line%line_type == 6)
       par%fine_structure = .false.; par%include_deuterium = .false.
       return
    endif
    call setup_check(rc, 'setup_resonance_line')
    line%line_type = h%line_type; line%wavelength0 = h%wavelength0; line%damping = h%damping; line%cross0 = h%cross0; line%vtherm1 = h%vtherm1
    line%g_recoil0 = h%g_recoil0; line%DnuHK_Hz = h%DnuHK_Hz; line%E1 = h%E1; line%E2 = h%E2; line%E3 = h%E3; line%mass_amu = h%mass_amu
    line%delta_nu_HD_Hz = h%delta_nu_HD_Hz; line%ratio_Dfreq_HD = h%ratio_Dfreq_HD; line%ratio_voigta_HD = h%ratio_voigta_HD
    line%nD_over_nH = h%nD_over_nH; line%g_recoil0_D = h%g_recoil0_D; line%P_conv = h%P_conv; line%E_conv = h%E_conv; line%wavelength0_Ha = h%wavelength0_Ha
    line%ion_id = cstring(h%ion_id); par%line_id = cstring(h%line_id)
    par%fine_structure = h%fine_structure /= 0; par%include_deuterium = h%include_deuterium /= 0
  end subroutine setup_resonance_line

  !--- fields of the reference's params_type that select physics or inputs outside this path: any non-default value stops the run
  subroutine check_unsupported_fields()
    if (par%temperature0 /= -999.0_wp) call not_supported('par%temperature0')
    if (par%Dfreq0 /= -999.0_wp) call not_supported('par%Dfreq0')
    if (par%voigt_a0 /= -999.0_wp) call not_supported('par%voigt_a0')
    if (par%source_rscale /= 0.0_wp) call not_supported('par%source_rscale')
    if (par%sersic_m /= 1.0_wp) call not_supported('par%sersic_m')
    if (par%Reff /= 0.0_wp) call not_supported('par%Reff')
    if (is_finite(par%xmin)) call not_supported('par%xmin')
    if (is_finite(par%ymin)) call not_supported('par%ymin')
    if (is_finite(par%zmin)) call not_supported('par%zmin')
    if (par%density_rscale /= -999.9_wp) call not_supported('par%density_rscale')
    if (par%density_zscale /= -999.9_wp) call not_supported('par%density_zscale')
    if (par%density_alpha /= 0.0_wp) call not_supported('par%density_alpha')
    !--- par%q only enters through par%Omega = q Omega xrange in the dens_file branch of grid_create (grid_mod_car.f90:346-351): without a
    !--- dens_file par%Omega is used as it stands (the shear of grid%vfy per x wrap, setup.f90:967-969)
    if (.not. par%continuum_normalize) call not_supported('par%continuum_normalize')
    if (par%EW_line /= 0.0_wp) call not_supported('par%EW_line')
    if (par%f_line /= 0.0_wp) call not_supported('par%f_line')
    if (trim(par%clump_radius_profile) /= 'constant') call not_supported('par%clump_radius_profile')
    if (trim(par%clump_density_profile) /= 'constant') call not_supported('par%clump_density_profile')
    if (trim(par%clump_number_profile) /= 'constant') call not_supported('par%clump_number_profile')
    if (par%clump_radius_alpha /= 0.0_wp) call not_supported('par%clump_radius_alpha')
    if (par%clump_radius_r0 /= 0.0_wp) call not_supported('par%clump_radius_r0')
    if (par%clump_density_alpha /= 0.0_wp) call not_supported('par%clump_density_alpha')
    if (par%clump_density_r0 /= 0.0_wp) call not_supported('par%clump_density_r0')
    if (par%clump_number_alpha /= 0.0_wp) call not_supported('par%clump_number_alpha')
    if (par%clump_number_r0 /= 0.0_wp) call not_supported('par%clump_number_r0')
    if (par%clump_radius_min /= -1.0_wp) call not_supported('par%clump_radius_min')
    if (par%clump_radius_max_in /= -1.0_wp) call not_supported('par%clump_radius_max_in')
    if (trim(par%clump_profile_file) /= '') call not_supported('par%clump_profile_file')
    if (par%amr_snapnum /= 0) call not_supported('par%amr_snapnum')
    if (trim(par%ionization_model) /= 'cie_formula') call not_supported('par%ionization_model')
    if (trim(par%dust_model) /= 'global_dgr') call not_supported('par%dust_model')
    if (trim(par%emissivity_model) /= 'none') call not_supported('par%emissivity_model')
    if (trim(par%ion_model) /= 'none') call not_supported('par%ion_model')
    if (par%metallicity_global /= -1.0_wp) call not_supported('par%metallicity_global')
    if (par%Z_ref /= 0.0134_wp) call not_supported('par%Z_ref')
    if (par%f_ion_dust /= 0.01_wp) call not_supported('par%f_ion_dust')
    if (par%out_bitpix /= 0 .and. par%out_bitpix /= -32 .and. par%out_bitpix /= -64) call not_supported('par%out_bitpix (0, -32 or -64)')
    if (par%sampling_method /= 1) call not_supported('par%sampling_method')
    if (par%f_composite /= 0.5_wp) call not_supported('par%f_composite')
    if (par%save_backup) call not_supported('par%save_backup')
    if (par%save_input_grid) call not_supported('par%save_input_grid')
    if (par%use_cie_condition) call not_supported('par%use_cie_condition')
    if (par%ny_2gam /= 101) call not_supported('par%ny_2gam')
    if (par%transport_2gamma) call not_supported('par%transport_2gamma')
    if (trim(par%line_prof_file) /= '') call not_supported('par%line_prof_file')
    if (par%line_prof_file_type /= 0) call not_supported('par%line_prof_file_type')
    if (par%reduce_factor /= 1) call not_supported('par%reduce_factor')
    if (par%centering /= 0) call not_supported('par%centering')
    if (par%stellar_limb_darkening /= 2) call not_supported('par%stellar_limb_darkening')
    if (par%distance_star_to_planet /= 0.0_wp) call not_supported('par%distance_star_to_planet')
    if (par%stellar_radius /= 0.0_wp) call not_supported('par%stellar_radius')
    if (any(is_finite(par%inclination_angle))) call not_supported('par%inclination_angle')
    if (any(is_finite(par%position_angle))) call not_supported('par%position_angle')
    if (any(is_finite(par%phase_angle))) call not_supported('par%phase_angle')
    if (is_finite(par%rotation_center_x)) call not_supported('par%rotation_center_x')
    if (is_finite(par%rotation_center_y)) call not_supported('par%rotation_center_y')
    if (is_finite(par%rotation_center_z)) call not_supported('par%rotation_center_z')
  end subroutine check_unsupported_fields

  subroutine not_supported(what)
    character(len=*), intent(in) :: what
    write(*,'(3a)') 'ERROR: ', what, ' is not supported by the MI355X photon path (see DESIGN.md, scope).'
    stop 1
  end subroutine not_supported

  !--- read_input: setup.f90:4-579 (namelist + the post-processing this path needs)
  subroutine read_input(model_infile)
    character(len=*), intent(in) :: model_infile
    integer :: unit, ios
    character(len=256) :: msg
    namelist /parameters/ par

    call open_sanitized(model_infile, unit)
    read(unit, parameters, iostat=ios, iomsg=msg)
    close(unit)
    if (ios /= 0) then          ! unknown names fail the namelist read, as in the reference
       write(*,'(2a)') 'ERROR reading &parameters: ', trim(msg); stop 1
    endif

    if (par%no_photons >= 1) par%nphotons = int(par%no_photons, int64)
    if (par%no_print   > 1) par%nprint = int(par%no_print)
    if (par%nprint >= par%nphotons) par%nprint = int(par%nphotons/10)
    if (par%no_photons <= 10) par%nprint = 1

    par%geometry = strlowcase(par%geometry); par%source_geometry = strlowcase(par%source_geometry); par%velocity_type = strlowcase(par%velocity_type)
    par%distance_unit = strlowcase(par%distance_unit); par%spectral_type = strlowcase(par%spectral_type); par%h2_model = strlowcase(par%h2_model)

    call check_unsupported_fields()
    if (par%use_amr_grid .and. trim(strlowcase(par%amr_type)) /= 'generic') call not_supported('par%amr_type = '//trim(par%amr_type))
    if (par%use_amr_grid .and. len_trim(par%amr_file) == 0) then
       write(*,'(a)') 'ERROR: par%use_amr_grid = .true. needs par%amr_file'; stop 1
    endif
    if (par%use_amr_grid .and. par%use_clump_medium) call not_supported('use_amr_grid together with use_clump_medium')
    if ((par%calc_J .or. par%calc_P .or. par%calc_Pnew) .and. par%use_clump_medium) &
       call not_supported('calc_J / calc_P / calc_Pnew with par%use_clump_medium')
    select case (trim(par%line_id))
    case ('CII_1334', 'SiII_1527', 'SiII_1260', 'SiII_1193', 'SiII_1190', 'SiII_1304', 'FeII_2250', 'FeII_2261', 'FeII_UV3', 'FeII_2344', 'FeII_UV1', 'FeII_2600', &
          'FeII_UV2', 'FeII_2383', 'HeI_10833')
    case ('ly_alpha', 'ly_alpha_HD', 'ly_beta', 'CIV_1548', 'NV_1239', 'OVI_1032', 'NaI_D', 'CaII_HK', 'MgII_2796', 'SiIV_1394', 'AlII_1671')
    case default
       call not_supported('par%line_id = '//trim(par%line_id))
    end select
    if (par%z_symmetry) call not_supported('par%z_symmetry')
    if (par%xy_symmetry) then                       ! mirrors in x and y, the full z range (grid_mod_car.f90:113-134)
       if (par%xyz_symmetry .or. par%xy_periodic .or. par%use_clump_medium) &
          call not_supported('par%xy_symmetry together with xyz_symmetry / xy_periodic / clumps')
       if (trim(par%source_geometry) /= 'point' .or. par%xs_point /= 0.0_wp .or. par%ys_point /= 0.0_wp) &
          call not_supported('par%xy_symmetry with a source off the z axis (sources are folded for xyz_symmetry only)')
       if (par%calc_J .or. par%calc_P .or. par%calc_Pnew) call not_supported('par%geometry_JPa = 2 (cylindrical profiles, the xy_symmetry default)')
    endif
    if (len_trim(par%dens_file) + len_trim(par%temp_file) + len_trim(par%velo_file) + len_trim(par%emiss_file) + &
        len_trim(par%cart_file) + len_trim(par%input_field) + len_trim(par%star_file) > 0) call not_supported('an input field file')

    select case(trim(par%geometry))
    case ('');    par%geometry = 'sphere'
    case ('box'); par%geometry = 'rectangle'
    end select
    if (trim(par%geometry) /= 'sphere' .and. trim(par%geometry) /= 'rectangle') call not_supported('par%geometry = '//trim(par%geometry))
    if (trim(par%source_geometry) == 'sphere')   par%source_geometry = 'uniform_sphere'
    if (trim(par%source_geometry) == 'cylinder') par%source_geometry = 'uniform_cylinder'
    select case (trim(par%source_geometry))
    case ('point', 'uniform_sphere')
    case ('uniform_cylinder', 'uniform', 'uniform_xy', 'gaussian')                             ! generate_photon.f90:43-72
       if (par%use_amr_grid .or. par%use_clump_medium) call not_supported('par%source_geometry = '//trim(par%source_geometry)//' with the octree / clump medium')
    case default; call not_supported('par%source_geometry = '//trim(par%source_geometry))
    end select
    select case (trim(par%spectral_type))
    case ('voigt', 'monochromatic', 'gaussian')
    case ('continuum'); par%comoving_source = .false.                                            ! setup.f90:115
    case default; call not_supported('par%spectral_type = '//trim(par%spectral_type))
    end select
    select case(trim(par%velocity_type))
    case ('', 'hubble', 'constant_radial', 'parallel_velocity')
    case ('rotating_solid_body', 'rotating_galaxy_halo')       ! Cartesian grid, or per-clump fields of assign_clump_velocities_from_type
    case ('power_law', 'linear_decelerate', 'ssh')
       if (par%use_clump_medium) call not_supported('par%velocity_type = '//trim(par%velocity_type)//' with the clump medium')
    case default; call not_supported('par%velocity_type = '//trim(par%velocity_type))
    end select

    call setup_resonance_line()
    par%nscatt_dust = 0.0_wp; par%nscatt_gas = 0.0_wp; par%nscatt_tot = 0.0_wp

    if (par%temperature <= 0.0_wp .and. par%bturb <= 0.0_wp) then
       write(*,'(a)') 'ERROR: par%temperature must be > 0 K (or set par%bturb > 0)'; stop 1
    endif
    call input_rules()                                                   ! the rules between the namelist fields (setup.f90:160-500): lart_host_input_rules, once for both hosts

    if (line%line_type == 8) call ly_beta_rules()                                               ! setup.f90:239-300
    if (trim(par%h2_model) /= 'none') call h2_rules()                                           ! setup.f90:302-365
    if ((par%geometry_JPa == 0 .or. par%geometry_JPa == 2) .and. (par%calc_J .or. par%calc_P .or. par%calc_Pnew)) &
       call not_supported('par%geometry_JPa = 2 (cylindrical profiles)')
    if (len_trim(par%scatt_mat_file) > 0 .and. par%DGR > 0.0_wp) then                          ! :496-500
       call setup_scattering_matrix(trim(par%scatt_mat_file))
    else
       if (par%use_stokes) par%DGR = 0.0_wp
    endif
    if (par%DGR == 0.0_wp) par%save_Jabs = .false.

    if (len_trim(par%out_file) == 0) then
       par%base_name = base_of(model_infile, .true.)
    else
       par%base_name = base_of(par%out_file, .false.)
    endif
    par%file_format = strlowcase(par%file_format)
    par%out_file = trim(par%base_name)

    write(*,'(a)')        ''
    write(*,'(3a)')       '+++++ ',trim(model_infile),' +++++'
    write(*,'(2a)')       'Line ID                   : ', trim(par%line_id)
    write(*,'(a,2f7.4)')  'Dust Parameters(a, g)     : ', par%albedo, par%hgg
    call say('Dust-to-Gas Ratio         : ', par%DGR, 'es12.3')
    if (par%DGR > 0.0) write(*,'(a,es12.3)') 'Dust Extinction per H     : ', par%cext_dust
    call say('Total number of photons   : ', par%no_photons, 'es12.3')
  contains
    !--- the namelist fields the rules read and change go through the library's record and come back
    subroutine input_rules()
      type(lart_host_rules) :: r
      r%nx = par%nx; r%ny = par%ny; r%nz = par%nz; r%nr = par%nr; r%nside = par%nside; r%nxim = par%nxim; r%nyim = par%nyim; r%nmu = par%nmu
      r%geometry_JPa = par%geometry_JPa; r%intensity_unit = par%intensity_unit
      select case (trim(par%distance_unit))
      case ('');    r%distance_unit = LART_UNIT_NONE
      case ('kpc'); r%distance_unit = LART_UNIT_KPC
      case ('pc');  r%distance_unit = LART_UNIT_PC
      case ('au');  r%distance_unit = LART_UNIT_AU
      case default; r%distance_unit = LART_UNIT_OTHER
      end select
      r%xyz_symmetry = b2i(par%xyz_symmetry); r%xy_symmetry = b2i(par%xy_symmetry); r%xy_periodic = b2i(par%xy_periodic)
      r%use_clump_medium = b2i(par%use_clump_medium); r%use_amr_grid = b2i(par%use_amr_grid); r%use_stokes = b2i(par%use_stokes)
      r%sphere = b2i(trim(par%geometry) == 'sphere'); r%save_all = b2i(par%save_all)
      r%save_peeloff = b2i(par%save_peeloff); r%save_peeloff_2D = b2i(par%save_peeloff_2D); r%save_peeloff_3D = b2i(par%save_peeloff_3D)
      r%save_Jabs = b2i(par%save_Jabs); r%save_Jmu = b2i(par%save_Jmu); r%core_skip = b2i(par%core_skip); r%core_skip_global = b2i(par%core_skip_global)
      r%tau0 = par%tau0; r%taumax = par%taumax; r%N_HI = par%N_HI; r%N_HImax = par%N_HImax; r%N_HIhomo = par%N_HIhomo; r%N_gasmax = par%N_gasmax
      r%N_gashomo = par%N_gashomo; r%cext_dust = par%cext_dust; r%DGR = par%DGR; r%mu_min = par%mu_min; r%dmu = par%dmu
      r%rmax = par%rmax; r%xmax = par%xmax; r%ymax = par%ymax; r%zmax = par%zmax; r%source_rmax = par%source_rmax; r%distance2cm = par%distance2cm
      call setup_check(lart_host_input_rules(r), 'read_input')
      par%nx = r%nx; par%ny = r%ny; par%nz = r%nz; par%nxim = r%nxim; par%nyim = r%nyim; par%geometry_JPa = r%geometry_JPa; par%intensity_unit = r%intensity_unit
      if (r%distance_unit == LART_UNIT_USER) par%distance_unit = 'user'
      par%xyz_symmetry = r%xyz_symmetry /= 0; par%use_stokes = r%use_stokes /= 0
      par%save_peeloff = r%save_peeloff /= 0; par%save_peeloff_2D = r%save_peeloff_2D /= 0; par%save_peeloff_3D = r%save_peeloff_3D /= 0
      par%save_Jabs = r%save_Jabs /= 0; par%core_skip = r%core_skip /= 0
      par%taumax = r%taumax; par%N_HImax = r%N_HImax; par%N_gasmax = r%N_gasmax; par%N_gashomo = r%N_gashomo; par%DGR = r%DGR; par%mu_min = r%mu_min; par%dmu = r%dmu
      par%rmax = r%rmax; par%xmax = r%xmax; par%ymax = r%ymax; par%zmax = r%zmax; par%source_rmax = r%source_rmax; par%distance2cm = r%distance2cm
      if (iand(r%notes, LART_NOTE_NO_PEEL_WITH_XYZSYM) /= 0) then
         write(*,'(a)') '---> peelingoff is not allowed for a geometry with the xyz_symmetry.'; write(*,'(a)') '---> setting par%save_peeloff = .false.'
      endif
    end subroutine input_rules
    pure function b2i(l) result(i)
      logical, intent(in) :: l
      integer(c_int32_t) :: i
      i = merge(1, 0, l)
    end function b2i
    !--- 'ly_beta' (line type 8): what the band-2 path cannot do is refused, core-skipping would bias the conversion rate, the dust of band 1 takes the
    !--- Draine values at 1026 A when the input left the three defaults alone, and R_Ha is the ratio of the two extinction cross-sections
    subroutine ly_beta_rules()
      character(len=64) :: why
      why = ''
      if (par%use_clump_medium) then
         why = 'clump medium'
      else if (par%nside > 0) then
         why = 'inside-observer (HEALPix)'
      else if (par%use_stokes) then
         why = 'Stokes polarization'
      else if (par%xyz_symmetry .or. par%xy_symmetry .or. par%xy_periodic .or. par%Omega /= 0.0_wp) then
         why = 'xyz/xy symmetry, xy_periodic and shear'
      endif
      if (len_trim(why) > 0) then
         write(*,'(3a)') 'ly_beta: ', trim(why), ' not yet supported'; stop 1
      endif
      if (par%core_skip .or. par%core_skip_global) write(*,'(a)') 'NOTE: ly_beta forces core_skip = .false. (core-skip would bias the conversion rate).'
      par%core_skip = .false.; par%core_skip_global = .false.
      !--- (the reference compares with double-precision literals while the defaults are single-precision ones widened to double, define.f90:454-456:
      !---  the re-default only happens when the input sets those double values itself)
      if (par%albedo == 0.3253_wp .and. par%hgg == 0.6761_wp .and. par%cext_dust == 1.6059e-21_wp) then
         par%albedo = 0.2804_wp; par%hgg = 0.6483_wp; par%cext_dust = 2.047e-21_wp
         write(*,'(a)') 'NOTE: ly_beta re-defaults band-1 dust constants to Draine 1026A values (cext=2.047e-21, albedo=0.2804, hgg=0.6483).'
      endif
      par%R_Ha = 0.0_wp
      if (par%cext_dust_Ha > 0.0_wp .and. par%cext_dust > 0.0_wp) par%R_Ha = par%cext_dust_Ha/par%cext_dust
    end subroutine ly_beta_rules
    !--- the H2 Lyman-band lines next to Ly-alpha (par%h2_model = 'neufeld'): refusals as the reference's, then h2_init through the library
    subroutine h2_rules()
      character(len=96) :: why
      character(len=256) :: dir
      why = ''
      if (line%line_type /= 1) then
         why = "only supported for line_id = 'ly_alpha' (line_type 1)"
      else if (trim(par%h2_model) /= 'neufeld') then
         why = "Phase 1 supports only h2_model='neufeld'; got "//trim(par%h2_model)
      else if (par%use_clump_medium) then
         why = 'clump medium not yet supported'
      else if (par%nside > 0) then
         why = 'inside-observer (HEALPix) not yet supported'
      else if (par%use_stokes) then
         why = 'Stokes polarization not yet supported'
      else if (par%xyz_symmetry .or. par%xy_symmetry .or. par%Omega /= 0.0_wp) then
         why = 'xyz/xy symmetry and shear not yet supported'
      else if (par%xy_periodic .and. (par%nx > 1 .or. par%ny > 1)) then
         why = 'xy_periodic supported only for the nx=ny=1 slab'
      else if (.not. par%f_H2 > 0.0_wp) then
         why = "par%f_H2 must be > 0 when h2_model /= 'none'"
      endif
      if (len_trim(why) > 0) then
         write(*,'(2a)') 'H2: ', trim(why); stop 1
      endif
      dir = par%h2_data_dir
      if (len_trim(dir) == 0) dir = 'data/h2'
      call setup_check(lart_host_h2_init(trim(dir)//'/energy_X.dat'//c_null_char, line%wavelength0, line%vtherm1, line%mass_amu, par%bturb, par%h2_temperature, &
                                         par%f_H2, merge(1, 0, par%h2_pure_absorption), line%h2), 'h2_init')
    end subroutine h2_rules
    !--- the shipped example inputs assign real literals to integer fields (par%nprint = 1e5), which ifort and
    !--- gfortran accept and flang rejects: copy the file to a scratch unit with those literals made integer.
    subroutine open_sanitized(fname, uout)
      character(len=*), intent(in) :: fname
      integer, intent(out) :: uout
      character(len=*), parameter :: intfields(14) = [character(len=10) :: 'nprint', 'nx', 'ny', 'nz', 'nr', 'nxfreq', 'nxim', &
           'nyim', 'nmu', 'nside', 'iseed', 'nobs', 'nvelocity', 'ngpu']
      character(len=512) :: ln, name, val
      integer :: uin, io, ip, ie, k
      real(wp) :: rv
      open(newunit=uin, file=trim(fname), status='old', iostat=io)
      if (io /= 0) then
         write(*,'(2a)') 'ERROR: cannot open input file ', trim(fname); stop 1
      endif
      open(newunit=uout, status='scratch', form='formatted')
      do
         read(uin, '(a)', iostat=io) ln
         if (io /= 0) exit
         ip = index(ln, 'par%'); ie = index(ln, '=')
         if (ip > 0 .and. ie > ip .and. (index(ln, '!') == 0 .or. index(ln, '!') > ie)) then
            name = adjustl(ln(ip+4:ie-1)); val = adjustl(ln(ie+1:))
            if (index(val, '!') > 0) val = val(1:index(val, '!')-1)
            do k = 1, size(intfields)
               if (trim(strlowcase(name)) == trim(intfields(k)) .and. scan(trim(val), '.eEdD') > 0) then
                  read(val, *, iostat=io) rv
                  if (io == 0) write(ln, '(a,a,a,i0)') ' par%', trim(name), ' = ', nint(rv)
               endif
            enddo
         endif
         write(uout, '(a)') trim(ln)
      enddo
      close(uin)
      rewind(uout)
    end subroutine open_sanitized
    !--- get_base_input_name (utility.f90:250-258: everything before the first '.in') for the input file, get_base_name (:220-248: only a
    !    recognised output extension is stripped) for par%out_file; the directory part goes in both
    function base_of(f, is_input) result(b)
      character(len=*), intent(in) :: f
      logical, intent(in) :: is_input
      character(len=128) :: b
      integer :: i1, i2, n
      i1 = index(f, '/', back=.true.) + 1
      b = f(i1:)
      n = len_trim(b)
      if (is_input) then
         i2 = index(b, '.in')
         if (i2 > 1) b = b(1:i2-1)
         return
      endif
      i2 = n
      if (n >= 8) then
         if (b(n-7:n) == '.fits.gz') i2 = n - 8
      endif
      if (i2 == n .and. n >= 5) then
         if (b(n-4:n) == '.hdf5' .or. b(n-4:n) == '.fits') i2 = n - 5
      endif
      if (i2 == n .and. n >= 3) then
         if (b(n-2:n) == '.h5') i2 = n - 3
      endif
      if (i2 >= 1) b = b(1:i2)
    end function base_of
  end subroutine read_input

  !--- setup_scattering_matrix (setup.f90:581-649): the header of the table overrides albedo / hgg / cext_dust; lart_host_mueller_setup normalises the
  !--- columns by the integral of S11 and builds the alias table of the scattering-angle bins
  subroutine setup_scattering_matrix(fname)
    character(len=*), intent(in) :: fname
    character(len=128) :: skip
    real(wp) :: wl
    integer :: i, u, n, ios
    open(newunit=u, file=fname, status='old', iostat=ios)
    if (ios /= 0) then
       write(*,'(2a)') 'ERROR: cannot open par%scatt_mat_file ', fname; stop 1
    endif
    read(u,*) skip
    read(u,*) wl, par%cext_dust, par%albedo, par%hgg, n
    read(u,*) skip
    scatt_mat%nPDF = n
    allocate(scatt_mat%coss(n), scatt_mat%S11(n), scatt_mat%S12(n), scatt_mat%S33(n), scatt_mat%S34(n), scatt_mat%phase_PDF(n-1), scatt_mat%alias(n-1))
    read(u,*) (scatt_mat%coss(i), scatt_mat%S11(i), scatt_mat%S12(i), scatt_mat%S33(i), scatt_mat%S34(i), i = 1, n)
    close(u)
    call setup_check(lart_host_mueller_setup(int(n, c_int32_t), scatt_mat%coss, scatt_mat%S11, scatt_mat%S12, scatt_mat%S33, scatt_mat%S34, &
                                             scatt_mat%phase_PDF, scatt_mat%alias), 'setup_scattering_matrix')
  end subroutine setup_scattering_matrix

  !--- H(0,a) on the x = 0 node of the Voigt table (voigt_mod.f90:691-700 with v = 0)
  elemental function voigt0(a) result(h)
    real(wp), intent(in) :: a
    real(wp) :: h
    h = 1.0_wp + a*(-1.1283791671_wp + a*1.0_wp)
  end function voigt0

  function velocity_code(name) result(c)
    character(len=*), intent(in) :: name
    integer(c_int32_t) :: c
    select case (trim(name))
    case ('');                     c = LART_VEL_NONE
    case ('hubble');               c = LART_VEL_HUBBLE
    case ('constant_radial');      c = LART_VEL_CONSTANT_RADIAL
    case ('parallel_velocity');    c = LART_VEL_PARALLEL
    case ('rotating_solid_body');  c = LART_VEL_ROT_SOLID
    case ('rotating_galaxy_halo'); c = LART_VEL_ROT_HALO
    case ('power_law');            c = LART_VEL_POWER_LAW
    case ('linear_decelerate');    c = LART_VEL_LINEAR_DECEL
    case ('ssh');                  c = LART_VEL_SSH
    case default;                  c = -1
    end select
  end function velocity_code

  !--- grid_create (grid_mod_car.f90:73-1238), analytic-medium subset: the library builds the axes, the density mask of the shell / bicone, the opacity
  !--- normalisation and the velocity field (lart_host_car_medium), then the spectral axis and the core-skip scale (car_setup_freq_grid)
  subroutine grid_create(grid)
    type(grid_type), intent(inout), target :: grid
    type(lart_host_medium) :: m
    type(c_ptr) :: pD
    integer :: nx, ny, nz
    nx = par%nx; ny = par%ny; nz = par%nz
    grid%nx = nx; grid%ny = ny; grid%nz = nz
    grid%xmax = par%xmax; grid%ymax = par%ymax; grid%zmax = par%zmax; grid%rmin = par%rmin; grid%rmax = par%rmax
    m%nx = nx; m%ny = ny; m%nz = nz; m%xyz_symmetry = merge(1, 0, par%xyz_symmetry); m%xy_symmetry = merge(1, 0, par%xy_symmetry)
    m%velocity_type = velocity_code(par%velocity_type)
    m%xmax = par%xmax; m%ymax = par%ymax; m%zmax = par%zmax; m%rmin = par%rmin; m%rmax = par%rmax; m%cone_opening = par%cone_opening
    m%vtherm = vtherm_total(par%temperature); m%wavelength0 = line%wavelength0; m%damping = line%damping; m%cross0 = line%cross0
    m%dust_opacity = par%cext_dust*par%DGR
    m%taumax = par%taumax; m%tauhomo = par%tauhomo; m%N_gasmax = par%N_gasmax; m%N_gashomo = par%N_gashomo
    m%Vexp = par%Vexp; m%Vx = par%Vx; m%Vy = par%Vy; m%Vz = par%Vz; m%Vrot = par%Vrot; m%rinner = par%rinner
    m%velocity_alpha = par%velocity_alpha; m%rpeak = par%rpeak; m%Vpeak = par%Vpeak; m%DeltaV = par%DeltaV
    if (.not. par%use_clump_medium) then                                 ! the clump population carries the physical length unit
       par%distance_unit = ''; par%distance2cm = 1.0_wp
    endif
    allocate(grid%xface(nx+1), grid%yface(ny+1), grid%zface(nz+1))
    allocate(grid%vfx(nx,ny,nz), grid%vfy(nx,ny,nz), grid%vfz(nx,ny,nz), grid%Dfreq(nx,ny,nz), grid%voigt_a(nx,ny,nz), grid%rhokap(nx,ny,nz))
    pD = c_null_ptr
    if (par%DGR > 0.0_wp) then
       allocate(grid%rhokapD(nx,ny,nz)); pD = c_loc(grid%rhokapD)
    endif
    call setup_check(lart_host_car_medium(m, grid%xface, grid%yface, grid%zface, grid%rhokap, grid%voigt_a, grid%Dfreq, grid%vfx, grid%vfy, grid%vfz, pD), 'grid_create')
    grid%dx = m%dx; grid%dy = m%dy; grid%dz = m%dz; grid%xmin = m%xmin; grid%ymin = m%ymin; grid%zmin = m%zmin; grid%i0 = m%i0; grid%j0 = m%j0; grid%k0 = m%k0
    grid%xrange = grid%xmax - grid%xmin; grid%yrange = grid%ymax - grid%ymin; grid%zrange = grid%zmax - grid%zmin
    grid%Dfreq_ref = m%Dfreq_ref; grid%voigt_amean = m%voigt_amean
    par%taumax = m%taumax; par%tauhomo = m%tauhomo; par%N_gasmax = m%N_gasmax; par%N_gashomo = m%N_gashomo       ! what was not given is the medium's own
    call car_setup_freq_grid(grid, .true.)
    allocate(grid%Jout(grid%nxfreq), grid%Jin(grid%nxfreq), grid%Jabs(grid%nxfreq), grid%Jmu(grid%nxfreq, max(par%nmu,1)))
    grid%Jout = 0.0_wp; grid%Jin = 0.0_wp; grid%Jabs = 0.0_wp; grid%Jmu = 0.0_wp
    if (line%line_type == 8) then                                        ! grid%Jout_Ha, grid%Jabs_Ha: the band-2 spectra
       allocate(grid%Jout_Ha(grid%nxfreq_Ha), grid%Jabs_Ha(grid%nxfreq_Ha)); grid%Jout_Ha = 0.0_wp; grid%Jabs_Ha = 0.0_wp
    endif
    call say('voigt_a          : ', grid%voigt_amean, 'es16.3')
    call say('temperature (K)  : ', par%temperature, 'es16.3')
    write(6,'(a,es16.3)') 'N('//trim(line%ion_id)//')_pole     : ', m%N_gaspole_out
    write(6,'(a,es16.3)') 'N('//trim(line%ion_id)//')_homo     : ', m%N_gashomo_out
    write(6,'(a,es16.3)') 'tau_pole  ('//trim(line%ion_id)//') : ', m%taupole_out
    write(6,'(a,es16.3)') 'tau_homo  ('//trim(line%ion_id)//') : ', m%tauhomo_out
  end subroutine grid_create

  !--- host-side error of the lart_host_* helpers: print and stop (reference convention)
  subroutine host_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: n
    if (rc == 0) return
    p = lart_host_last_error()
    write(*,'(3a)', advance='no') 'ERROR in ', where, ': '
    if (c_associated(p)) then
       call c_f_pointer(p, msg, [512])
       do n = 1, 512
          if (msg(n) == c_null_char) exit
          write(*,'(a)', advance='no') msg(n)
       enddo
    endif
    write(*,*)
    stop 1
  end subroutine host_check

  !--- init_clumps, uniform path (clump_mod.f90:646-891); the per-clump loops (generate_clumps :897-1151,
  !--- build_clump_csr :1267-1342) run in liblart_gpu's host helpers on the reference's MT19937-64 stream
  subroutine init_clumps(R_sphere)
    real(wp), intent(in) :: R_sphere
    real(wp) :: temp_cl, rc, sig, acc
    type(lart_host_clumps_in) :: c
    integer :: n
    sphere_R = R_sphere
    if (len_trim(par%clump_input_file) > 0) then      ! a pre-built population replaces the generator (clump_mod.f90:674-680)
       if (par%clump_radius <= 0.0_wp) then
          write(*,*) 'ERROR: clump_radius must be > 0'; stop 1
       endif
       r_min_clump = max(0.0_wp, par%rmin)
       if (r_min_clump >= R_sphere) then
          write(*,'(a,2es12.4)') 'ERROR: par%rmin must be < par%rmax for clump placement; rmin, rmax = ', r_min_clump, R_sphere; stop 1
       endif
       call read_clumps_info(trim(par%clump_input_file), R_sphere)
       return
    endif
    temp_cl = merge(par%temperature, par%clump_temperature, par%clump_temperature < 0.0_wp)
    c%sphere_R = R_sphere; c%rmin = par%rmin; c%clump_radius = par%clump_radius; c%N_clumps_in = par%clump_N_clumps; c%f_vol_in = par%clump_f_vol
    c%f_cov_in = par%clump_f_cov; c%clump_tau0 = par%clump_tau0; c%clump_NHI = par%clump_NHI; c%clump_nH = par%clump_nH; c%taumax = par%taumax
    c%N_gasmax = par%N_gasmax; c%distance2cm = par%distance2cm
    c%vtherm = vtherm_total(temp_cl); c%wavelength0 = line%wavelength0; c%damping = line%damping; c%cross0 = line%cross0
    call setup_check(lart_host_clump_population(c), 'init_clumps')      ! population size, f_vol / f_cov, the opacity of one clump
    r_min_clump = c%r_min; cl_radius_max = par%clump_radius; rc = cl_radius_max
    cl_Dfreq_ref = c%Dfreq_ref; cl_vtherm_ref = c%vtherm; cl_temperature_ref = temp_cl; cl_voigt_a_ref = c%voigt_a_ref
    N_clumps = c%N; cl_rhokap_ref = c%rhokap_ref
    call say(' Clumps: N_clumps  = ', N_clumps, 'i14')
    write(*,'(a,f12.6)')  ' Clumps: f_vol     = ', c%fvol
    write(*,'(a,f12.5)')  ' Clumps: f_cov     = ', c%fcov
    write(*,'(a,2f12.5)') ' Clumps: rmin/rmax = ', r_min_clump, R_sphere
    call say(' Clumps: cl_rhokap = ', cl_rhokap_ref, 'es12.4')
    call say(' Clumps: voigt_a   = ', cl_voigt_a_ref, 'f12.5')
    call say(' Clumps: cl_Dfreq  = ', cl_Dfreq_ref, 'es12.4')
    n = int(N_clumps)
    allocate(cl_x(n), cl_y(n), cl_z(n), cl_vx(n), cl_vy(n), cl_vz(n), cl_radius(n), cl_rhokap(n), cl_voigt_a(n), cl_Dfreq(n), cl_vtherm(n))
    cl_radius = rc; cl_rhokap = cl_rhokap_ref; cl_voigt_a = cl_voigt_a_ref; cl_Dfreq = cl_Dfreq_ref; cl_vtherm = cl_vtherm_ref
    allocate(cl_temperature(n)); cl_temperature = cl_temperature_ref
    if (par%DGR > 0.0_wp) then
       allocate(cl_rhokapD(n)); cl_rhokapD = cl_rhokap_ref*par%cext_dust*par%DGR*cl_Dfreq_ref/line%cross0
    endif
    sig = 0.0_wp
    if (par%clump_sigma_v > 0.0_wp) sig = par%clump_sigma_v/cl_vtherm_ref
    call host_check(lart_host_clumps_generate_cone(N_clumps, sphere_R, r_min_clump, rc, &
         merge(1_c_int32_t, 0_c_int32_t, par%clump_fully_inside) + merge(2_c_int32_t, 0_c_int32_t, par%clump_allow_overlap), par%cone_opening, &
         sig, int(par%iseed, c_int64_t), cl_x, cl_y, cl_z, cl_vx, cl_vy, cl_vz, acc), 'lart_host_clumps_generate')
    if (par%clump_allow_overlap) then
       write(*,'(a)') ' Random placement done (overlap allowed).'
    else
       write(*,'(a,f6.1,a)') ' RSA done, acceptance rate = ', acc*100.0_wp, '%'
    endif
    if (len_trim(par%velocity_type) > 0) call assign_clump_velocities_from_type()
    call build_clump_csr(par%clump_allow_overlap)
  end subroutine init_clumps


  !--- build_clump_csr (clump_mod.f90:1267-1360) + check_has_overlap (:1544-1591; init_clumps runs it for par%clump_allow_overlap and for every loaded file)
  subroutine build_clump_csr(check_overlap)
    logical, intent(in) :: check_overlap
    real(c_double) :: cmin, cd
    integer(c_int32_t) :: cg, ovl
    integer(c_int64_t) :: nreg
    call host_check(lart_host_clumps_csr_dims(N_clumps, sphere_R, cl_radius_max, cg, cmin, cd), 'lart_host_clumps_csr_dims')
    cgx = cg; cg_xmin = cmin; cg_dx = cd
    if (allocated(cg_start)) deallocate(cg_start)
    if (allocated(cg_list)) deallocate(cg_list)
    allocate(cg_start(cgx**3 + 1))
    call host_check(lart_host_clumps_build_csr(N_clumps, cl_x, cl_y, cl_z, cl_radius, cgx, cg_xmin, cg_dx, cg_start, c_null_ptr, nreg), 'lart_host_clumps_build_csr')
    allocate(cg_list(max(1_int64, nreg)))
    call host_check(lart_host_clumps_build_csr(N_clumps, cl_x, cl_y, cl_z, cl_radius, cgx, cg_xmin, cg_dx, cg_start, c_loc(cg_list), nreg), 'lart_host_clumps_build_csr')
    write(*,'(a,i12,a,i5,a)') ' CSR grid: ', nreg, ' registrations in ', cgx, '^3 cells.'
    has_overlap = .false.
    if (check_overlap) then
       call host_check(lart_host_clumps_has_overlap(N_clumps, cl_x, cl_y, cl_z, cl_radius, cgx, cg_xmin, cg_dx, cg_start, cg_list, ovl), 'lart_host_clumps_has_overlap')
       has_overlap = ovl /= 0
       if (has_overlap) then
          write(*,'(a)') ' Overlap check: overlapping clumps detected — using overlap-aware raytrace.'
       else
          write(*,'(a)') ' Overlap check: no overlaps found — using standard raytrace.'
       endif
    endif
  end subroutine build_clump_csr

  !--- io_resolve_filename: iofile_mod.f90:143-177
  function resolve_filename(fname) result(res)
    character(len=*), intent(in) :: fname
    character(len=512) :: res
    character(len=8), parameter :: exts(4) = [character(len=8) :: '.h5', '.fits.gz', '.fits', '.dat']
    integer :: i, k
    logical :: ex
    res = fname
    do k = len_trim(fname), 1, -1
       if (fname(k:k) == '/' .or. fname(k:k) == char(92)) exit
       if (fname(k:k) == '.') return
    enddo
    do i = 1, size(exts)
       res = trim(fname)//trim(exts(i))
       inquire(file=trim(res), exist=ex)
       if (ex) return
    enddo
    write(*,'(3a)') 'io_resolve_filename: no file found for "', trim(fname), '"'; stop 1
  end function resolve_filename

  !--- the four sums of compute_clump_scalars (clump_mod.f90:2316-2378): tauhomo, taumax, N_gashomo, N_gasmax
  subroutine clump_scalars(vals)
    real(wp), intent(out) :: vals(4)
    call setup_check(lart_host_clump_scalars(N_clumps, sphere_R, r_min_clump, line%cross0, cl_x, cl_y, cl_z, cl_radius, cl_rhokap, cl_voigt_a, cl_Dfreq, vals), 'clump_scalars')
  end subroutine clump_scalars

  !--- read_clumps_info + rescale_loaded_clumps_to_target: clump_mod.f90:2040-2312, 2385-2449
  subroutine read_clumps_info(fname, R_sphere)
    character(len=*), intent(in) :: fname
    real(wp), intent(in) :: R_sphere
    interface
       function lart_clumps_file_read(path, n, data, present, keys, distunit) bind(C, name='lart_clumps_file_read') result(rc)
         import; character(kind=c_char), intent(in) :: path(*); integer(c_long_long), intent(out) :: n; type(c_ptr), intent(out) :: data
         integer(c_int), intent(out) :: present(9); real(c_double), intent(out) :: keys(5); character(kind=c_char), intent(out) :: distunit(64)
         integer(c_int) :: rc
       end function
       subroutine lart_free(p) bind(C, name='lart_free')
         import; type(c_ptr), value :: p
       end subroutine
    end interface
    character(len=512) :: path
    type(c_ptr) :: cdata
    real(c_double), pointer :: tab(:,:)
    real(c_double) :: keys(5)
    integer(c_int) :: present(9), rc
    integer(c_long_long) :: nn
    character(kind=c_char) :: cunit(64)
    character(len=64) :: distunit_file
    integer :: n, i
    real(wp) :: vals(4), realized, target_val, alpha
    character(len=12) :: which
    logical :: user_set
    sphere_R = R_sphere; clumps_from_file = .true.
    r_min_clump = max(0.0_wp, par%rmin)
    path = resolve_filename(fname)
    write(*,'(2a)') ' Clumps: reading from ', trim(path)
    rc = lart_clumps_file_read(trim(path)//c_null_char, nn, cdata, present, keys, cunit)
    if (rc /= 0) then
       write(*,'(3a,i0,a)') 'ERROR: cannot read clump_input_file: ', trim(path), ' (code ', rc, ': 1 open, 2 short file, 3 no table section, 4 columns, 6 no HDF5 in this build)'; stop 1
    endif
    if (keys(1) /= keys(1) .or. keys(1) <= 0.0_wp) then
       write(*,'(a)') 'ERROR: clump_input_file: N_CLUMPS keyword missing or <= 0'; stop 1
    endif
    N_clumps = int(keys(1), int64); n = int(N_clumps)
    if (n > nn .or. any(present(1:6) == 0)) then
       write(*,'(a)') 'ERROR: clump_input_file: the table needs the columns X, Y, Z, VX, VY, VZ with N_CLUMPS rows'; stop 1
    endif
    call c_f_pointer(cdata, tab, [int(nn), 9])
    allocate(cl_x(n), cl_y(n), cl_z(n), cl_vx(n), cl_vy(n), cl_vz(n), cl_radius(n), cl_rhokap(n), cl_voigt_a(n), cl_Dfreq(n), cl_vtherm(n), cl_temperature(n))
    cl_x = tab(1:n,1); cl_y = tab(1:n,2); cl_z = tab(1:n,3); cl_vx = tab(1:n,4); cl_vy = tab(1:n,5); cl_vz = tab(1:n,6)
    !--- read_perclump_or_keyword (:1965-2036): the column when the file has it, else the keyword for every clump
    call col_or_key(7, 2, 'R_CLUMP', 'CL_RAD', cl_radius)
    call col_or_key(8, 3, 'RHOKAP', 'RHOKAP', cl_rhokap)
    call col_or_key(9, 4, 'TEMP', 'TEMP_CL', cl_temperature)
    call lart_free(cdata)
    distunit_file = ''
    do i = 1, 64
       if (cunit(i) == c_null_char) exit
       distunit_file(i:i) = cunit(i)
    enddo
    user_set = len_trim(par%distance_unit) > 0 .and. par%distance2cm > 1.0_wp                  ! :2164-2176
    if (.not. user_set) then
       if (len_trim(distunit_file) > 0) then
          par%distance_unit = trim(distunit_file)
          write(*,'(3a)') " Clumps: adopting DISTUNIT = '", trim(par%distance_unit), "' from clump_input_file"
       endif
       if (keys(5) == keys(5) .and. keys(5) > 0.0_wp) then
          par%distance2cm = keys(5)
          write(*,'(a,es12.5,a)') ' Clumps: adopting DIST_CM = ', par%distance2cm, ' cm/code-unit from clump_input_file'
       endif
    endif
    do i = 1, n
       cl_vtherm(i) = vtherm_total(cl_temperature(i)); cl_Dfreq(i) = cl_vtherm(i)/(line%wavelength0*um2km); cl_voigt_a(i) = (line%damping/fourpi)/cl_Dfreq(i)
       cl_vx(i) = cl_vx(i)/cl_vtherm(i); cl_vy(i) = cl_vy(i)/cl_vtherm(i); cl_vz(i) = cl_vz(i)/cl_vtherm(i)
    enddo
    if (present(8) >= 2) then                                                                     ! DENSITY / DENS: :2190-2207
       if (len_trim(par%distance_unit) > 0 .or. par%distance2cm > 1.0_wp) then
          cl_rhokap = cl_rhokap*line%cross0/cl_Dfreq*par%distance2cm
          write(*,'(a,es12.4,a)') ' Clumps: density column converted to opacity using par%distance2cm = ', par%distance2cm, ' cm/code-unit'
       else
          write(*,'(a)') ' Clumps: WARNING -- a density column found but par%distance_unit/distance2cm is not set; treating the values as RHOKAP'
       endif
    endif
    cl_radius_max = maxval(cl_radius); cl_temperature_ref = cl_temperature(1); cl_vtherm_ref = cl_vtherm(1)
    cl_Dfreq_ref = cl_Dfreq(1); cl_voigt_a_ref = cl_voigt_a(1); cl_rhokap_ref = cl_rhokap(1)
    if (par%taumax > 0.0_wp .or. par%tauhomo > 0.0_wp .or. par%N_gasmax > 0.0_wp .or. par%N_gashomo > 0.0_wp) then
       call clump_scalars(vals)
       block                                                            ! the first given target in the order taumax, tauhomo, N_gasmax, N_gashomo
         integer, parameter :: slot(4) = [2, 1, 4, 3]
         character(len=9), parameter :: names(4) = [character(len=9) :: 'taumax', 'tauhomo', 'N_gasmax', 'N_gashomo']
         real(wp) :: given(4)
         integer :: k
         given = [par%taumax, par%tauhomo, par%N_gasmax, par%N_gashomo]
         do k = 1, 4
            if (given(k) > 0.0_wp) exit
         enddo
         realized = vals(slot(k)); target_val = given(k); which = names(k)
       end block
       if (realized <= 0.0_wp) then
          write(*,'(a)') ' Clumps: WARNING -- realized tau/NHI from loaded clumps is zero; cannot rescale. Leaving cl_rhokap unchanged.'
       else
          alpha = target_val/realized
          cl_rhokap = cl_rhokap*alpha; cl_rhokap_ref = cl_rhokap_ref*alpha
          write(*,'(3a,es12.4,a,es12.4)') ' Clumps: rescaled to par%', trim(which), ' = ', target_val, ', alpha = ', alpha
       endif
    endif
    if (par%DGR > 0.0_wp) then
       allocate(cl_rhokapD(n)); cl_rhokapD = cl_rhokap*par%cext_dust*par%DGR*cl_Dfreq/line%cross0
    endif
    call say(' Clumps: N_clumps  = ', N_clumps, 'i14')
    call say(' Clumps: cl_rhokap = ', cl_rhokap_ref, 'es12.4')
    call say(' Clumps: voigt_a   = ', cl_voigt_a_ref, 'f12.5')
    call say(' Clumps: cl_Dfreq  = ', cl_Dfreq_ref, 'es12.4')
    write(*,'(a,2es12.4)')' Clumps: r_cl min/max  = ', minval(cl_radius), cl_radius_max
    call build_clump_csr(.true.)
  contains
    subroutine col_or_key(icol, ikey, cname, kname, dst)
      integer, intent(in) :: icol, ikey
      character(len=*), intent(in) :: cname, kname
      real(wp), intent(out) :: dst(:)
      if (present(icol) /= 0) then
         dst = tab(1:n, icol)
      else if (keys(ikey) == keys(ikey)) then
         dst = keys(ikey)
      else
         write(*,'(5a)') 'ERROR: clump_input_file: neither a column ', cname, ' nor the keyword ', kname, ' in the file'; stop 1
      endif
    end subroutine col_or_key
  end subroutine read_clumps_info

  !--- assign_clump_velocities_from_type (clump_mod.f90:1153-1264, the fields this path supports): the systematic field added to the clumps' velocities
  subroutine assign_clump_velocities_from_type()
    call setup_check(lart_host_clump_velocities(N_clumps, velocity_code(par%velocity_type), sphere_R, par%Vexp, par%Vx, par%Vy, par%Vz, par%Vrot, par%rinner, &
                                                cl_x, cl_y, cl_z, cl_vtherm, cl_vx, cl_vy, cl_vz), 'assign_clump_velocities_from_type')
    write(*,'(2a)') ' Clumps: velocity_type   = ', trim(par%velocity_type)
  end subroutine assign_clump_velocities_from_type

  !--- compute_clump_scalars: clump_mod.f90:2316-2378
  subroutine compute_clump_scalars()
    real(wp) :: vals(4)
    call clump_scalars(vals)
    par%tauhomo = vals(1); par%taumax = vals(2); par%N_gashomo = vals(3); par%N_gasmax = vals(4)
  end subroutine compute_clump_scalars

  !--- grid_create_clump: grid_mod_clump.f90:33-131
  subroutine grid_create_clump(grid)
    type(grid_type), intent(inout) :: grid
    character(len=128) :: vtype
    if (par%rmax <= 0.0_wp) then
       write(*,*) 'ERROR: par%rmax must be > 0 for clump medium'; stop 1
    endif
    par%xmax = par%rmax; par%ymax = par%rmax; par%zmax = par%rmax
    if (par%nx < 1) par%nx = 11
    if (par%ny < 1) par%ny = 11
    if (par%nz < 1) par%nz = 11
    par%xyz_symmetry = .false.; par%xy_symmetry = .false.; par%xy_periodic = .false.; par%z_symmetry = .false.
    call init_clumps(par%rmax)
    call compute_clump_scalars()
    call say(' Clump derived: tauhomo   = ', par%tauhomo, 'es14.5')
    call say(' Clump derived: taumax    = ', par%taumax, 'es14.5')
    call say(' Clump derived: N_gashomo = ', par%N_gashomo, 'es14.5')
    call say(' Clump derived: N_gasmax  = ', par%N_gasmax, 'es14.5')
    vtype = par%velocity_type; par%velocity_type = ''       ! the background grid is static; velocities live on the clumps
    call grid_create(grid)
    par%velocity_type = vtype
    grid%rhokap = 0.0_wp; grid%Dfreq = cl_Dfreq_ref; grid%voigt_a = cl_voigt_a_ref
    grid%vfx = 0.0_wp; grid%vfy = 0.0_wp; grid%vfz = 0.0_wp
    if (allocated(grid%rhokapD)) grid%rhokapD = 0.0_wp
    grid%Dfreq_ref = cl_Dfreq_ref; grid%voigt_amean = cl_voigt_a_ref
    call say(' Clump grid: rmax      = ', par%rmax, 'f12.5')
    write(*,'(a,3i5)')    ' Clump grid: nx,ny,nz  = ', grid%nx, grid%ny, grid%nz
    call say(' Clump grid: cl_Dfreq_ref  = ', cl_Dfreq_ref, 'es12.4')
    write(*,'(a,f12.5,a)')' Clump grid: cl_vtherm_ref = ', cl_vtherm_ref, ' km/s'
    call say(' Clump grid: voigt_a   = ', cl_voigt_a_ref, 'f12.5')
  end subroutine grid_create_clump

  !--- grid_create_amr: grid_mod_amr.f90:34-526 for the generic text leaf table (all hydrogen neutral, no dust model file)
  subroutine grid_create_amr(grid)
    type(grid_type), intent(inout) :: grid
    integer :: unit, ios, n, i, il, ic, nread
    real(wp) :: amr_origin, boxlen, row(9), vt, opac_length, nopac, tauhomo, tau_raw, NHI, taupole, opac_norm, xc, zc, t, atau0_cell, xi, chi, H0
    real(wp), allocatable, target :: lx(:), ly(:), lz(:), nH(:), Tl(:), lvx(:), lvy(:), lvz(:)
    integer(int32), allocatable, target :: lev(:)
    integer(c_int32_t) :: nc, lm
    character(len=512) :: line_txt
    character(len=512) :: q

    !--- binary containers (read_generic_amr.f90:186-279): FITS binary table (.fits, .fits.gz) or HDF5 group (.h5, .hdf5)
    if (amr_is_binary(trim(par%amr_file))) then
       call amr_read_binary_file()
    else
    !--- read_generic_amr.f90:112-185: "NLEAF n" / "BOXLEN L" or legacy "n L", rows x y z level nH T vx vy vz, '#' comments
    open(newunit=unit, file=trim(par%amr_file), status='old', action='read', iostat=ios)
    if (ios /= 0) then
       write(*,'(2a)') 'generic_amr_read: cannot open file: ', trim(par%amr_file); stop 1
    endif
    n = 0; boxlen = -1.0_wp
    do
       read(unit,'(a)', iostat=ios) line_txt
       if (ios /= 0) exit
       q = adjustl(line_txt)
       if (len_trim(q) == 0 .or. q(1:1) == '#') cycle
       if (strlowcase(q(1:6)) == 'boxlen') then
          read(q(7:), *) boxlen
       else if (strlowcase(q(1:5)) == 'nleaf') then
          read(q(6:), *) n
       else
          read(q, *, iostat=ios) n, boxlen
          if (ios /= 0) then
             write(*,'(a)') 'generic_amr_read: cannot parse text header'; stop 1
          endif
       endif
       if (n > 0 .and. boxlen > 0.0_wp) exit
    enddo
    if (n <= 0 .or. boxlen <= 0.0_wp) then
       write(*,'(a)') 'generic_amr_read: text header must provide NLEAF and BOXLEN'; stop 1
    endif
    allocate(lx(n), ly(n), lz(n), lev(n), nH(n), Tl(n), lvx(n), lvy(n), lvz(n))
    nread = 0
    do while (nread < n)
       read(unit,'(a)', iostat=ios) line_txt
       if (ios /= 0) exit
       q = adjustl(line_txt)
       if (len_trim(q) == 0 .or. q(1:1) == '#') cycle
       read(q, *, iostat=ios) row
       if (ios /= 0) exit
       nread = nread + 1
       lx(nread) = row(1); ly(nread) = row(2); lz(nread) = row(3); lev(nread) = int(row(4)); nH(nread) = row(5); Tl(nread) = row(6)
       lvx(nread) = row(7); lvy(nread) = row(8); lvz(nread) = row(9)
    enddo
    close(unit)
    if (nread /= n) then
       write(*,'(a,i0,a,i0)') 'generic_amr_read: expected ', n, ' rows, got ', nread; stop 1
    endif
    amr_origin = -0.5_wp*boxlen                           ! text files carry no origin: centred box (:177-181)
    endif

    !--- tree + neighbour tables (octree_mod.f90:460-569, 619-682) by the host helper; text files: box centred on the origin (:179-181)
    amr_nleaf = n; amr_lo = amr_origin; amr_hi = amr_origin + boxlen
    call amr_check(lart_host_amr_count(amr_nleaf, lx, ly, lz, lev, amr_lo, amr_hi, nc), 'lart_host_amr_count')
    amr_ncells = nc
    allocate(amr_cx(nc), amr_cy(nc), amr_cz(nc), amr_ch(nc), amr_children(8,nc), amr_neighbor(6,nc), amr_ileaf(nc), amr_icell_of_leaf(n))
    call amr_check(lart_host_amr_build(amr_nleaf, lx, ly, lz, lev, amr_lo, amr_hi, nc, amr_cx, amr_cy, amr_cz, amr_ch, &
         amr_children, amr_neighbor, amr_ileaf, amr_icell_of_leaf, c_null_ptr, lm), 'lart_host_amr_build')
    amr_levelmax = lm
    write(*,'(a,i0)')     'AMR nleaf    : ', amr_nleaf
    write(*,'(a,es12.4,a)') 'AMR boxlen   : ', boxlen, ' (code units)'

    !--- geometry (:186-199)
    if (trim(par%geometry) == 'sphere') then
       par%rmax = boxlen*0.5_wp; par%xmax = par%rmax; par%ymax = par%rmax; par%zmax = par%rmax
    else
       par%rmax = -1.0_wp; par%xmax = amr_hi; par%ymax = amr_hi; par%zmax = amr_hi
    endif
    if (par%source_rmax < 0.0_wp) par%source_rmax = par%rmax

    !--- leaf physics (:230-330): T floor 10 K, all H neutral
    grid%Dfreq_ref = vtherm_total(par%temperature)/(line%wavelength0*um2km); grid%voigt_amean = (line%damping/fourpi)/grid%Dfreq_ref
    allocate(amr_Dfreq(n), amr_voigt_a(n), amr_rhokap(n), amr_vfx(n), amr_vfy(n), amr_vfz(n))
    if (par%DGR > 0.0_wp) allocate(amr_rhokapD(n))
    par%distance_unit = ''; par%distance2cm = merge(par%distance2cm, 1.0_wp, par%distance2cm > 0.0_wp)
    do il = 1, n
       vt = vtherm_total(max(Tl(il), 10.0_wp))
       amr_Dfreq(il)   = vt/(line%wavelength0*um2km)
       amr_voigt_a(il) = (line%damping/fourpi)/amr_Dfreq(il)
       amr_rhokap(il)  = nH(il)*line%cross0/amr_Dfreq(il)*par%distance2cm
       if (par%DGR > 0.0_wp) amr_rhokapD(il) = nH(il)*par%cext_dust*par%DGR*par%distance2cm
       amr_vfx(il) = lvx(il)/vt; amr_vfy(il) = lvy(il)/vt; amr_vfz(il) = lvz(il)/vt
    enddo

    if (par%cone_opening > 0.0_wp .and. par%cone_opening < 90.0_wp) then     ! Step 4b (:336-350): biconical mask by leaf centre
       do il = 1, n
          xc = sqrt(amr_cx(amr_icell_of_leaf(il))**2 + amr_cy(amr_icell_of_leaf(il))**2 + amr_cz(amr_icell_of_leaf(il))**2)
          if (xc > 0.0_wp) then
             if (abs(amr_cz(amr_icell_of_leaf(il)))/xc < cos(par%cone_opening*deg2rad)) then
                amr_rhokap(il) = 0.0_wp
                if (allocated(amr_rhokapD)) amr_rhokapD(il) = 0.0_wp
             endif
          endif
       enddo
    endif
    !--- homogeneous and +z pole optical depths, normalisation (:352-470)
    opac_length = boxlen/2.0_wp
    nopac = real(count(amr_rhokap > 0.0_wp), wp)
    tauhomo = 0.0_wp
    do il = 1, n
       tauhomo = tauhomo + amr_rhokap(il)*voigt0(amr_voigt_a(il))
    enddo
    if (nopac > 0.0_wp) then
       tauhomo = tauhomo/nopac*opac_length
    else
       tauhomo = 0.0_wp
    endif
    xc = amr_cx(1); zc = amr_cz(1); tau_raw = 0.0_wp; NHI = 0.0_wp
    do while (zc < amr_hi)
       il = amr_find_leaf(xc, xc, zc)
       if (il <= 0) exit
       ic = amr_icell_of_leaf(il)
       t  = (amr_cz(ic) + amr_ch(ic) - zc)/1.0_wp
       tau_raw = tau_raw + amr_rhokap(il)*voigt0(amr_voigt_a(il))*t
       NHI     = NHI + amr_rhokap(il)*amr_Dfreq(il)/line%cross0*t
       zc = zc + t
    enddo
    taupole = tau_raw
    if (taupole <= 0.0_wp) taupole = tauhomo
    opac_norm = 1.0_wp
    if (par%taumax > 0.0_wp) then
       if (taupole > 0.0_wp) opac_norm = par%taumax/taupole
    else if (par%N_gasmax > 0.0_wp .and. NHI > 0.0_wp) then
       opac_norm = par%N_gasmax/NHI
    endif
    amr_rhokap = amr_rhokap*opac_norm
    if (allocated(amr_rhokapD)) amr_rhokapD = amr_rhokapD*opac_norm
    tauhomo = 0.0_wp
    do il = 1, n
       tauhomo = tauhomo + amr_rhokap(il)*voigt0(amr_voigt_a(il))
    enddo
    if (nopac > 0.0_wp) tauhomo = tauhomo/nopac*opac_length
    if (par%taumax <= 0.0_wp) par%taumax = tauhomo
    par%N_gasmax = NHI*opac_norm; par%N_gashomo = par%N_gasmax; par%tauhomo = tauhomo
    write(*,'(a,es12.4)') 'AMR voigt_a      : ', grid%voigt_amean
    write(*,'(a,es12.4)') 'AMR temperature  : ', par%temperature
    write(*,'(a,es12.4)') 'AMR N(HI)_pole   : ', par%N_gasmax
    write(*,'(a,es12.4)') 'AMR tau_pole      : ', par%taumax

    !--- spectral axis and global core-skip threshold (amr_setup_freq_grid :726-800: the Cartesian rules)
    grid%nx = n; grid%ny = 1; grid%nz = 1
    grid%xmax = par%xmax; grid%ymax = par%ymax; grid%zmax = par%zmax; grid%xmin = amr_lo; grid%ymin = amr_lo; grid%zmin = amr_lo
    grid%rmax = par%rmax; grid%rmin = par%rmin; grid%dx = boxlen; grid%dy = boxlen; grid%dz = boxlen
    call car_setup_freq_grid(grid)
    allocate(grid%Jout(grid%nxfreq), grid%Jin(grid%nxfreq), grid%Jabs(grid%nxfreq), grid%Jmu(grid%nxfreq, max(par%nmu,1)))
    grid%Jout = 0.0_wp; grid%Jin = 0.0_wp; grid%Jabs = 0.0_wp; grid%Jmu = 0.0_wp
    atau0_cell = grid%voigt_amean*par%tauhomo/(boxlen/(2.0_wp*amr_ch(1)))
    if (atau0_cell > 1.0_wp) then
       if (atau0_cell <= 60.0_wp) then
          xi = 0.6_wp; chi = 1.2_wp
       else
          xi = 1.4_wp; chi = 0.6_wp
       endif
       grid%xcrit = 0.02_wp*exp(xi*(log(atau0_cell))**chi); grid%xcrit2 = grid%xcrit**2
    endif
    call create_JPa_amr_mem(grid)
  contains
    logical function amr_is_binary(f)
      character(len=*), intent(in) :: f
      integer :: n
      character(len=len(f)) :: g
      g = strlowcase(f); n = len_trim(g); amr_is_binary = .false.
      if (n > 3) amr_is_binary = g(n-2:n) == '.h5'
      if (n > 5) amr_is_binary = amr_is_binary .or. g(n-4:n) == '.fits' .or. g(n-4:n) == '.hdf5'
      if (n > 8) amr_is_binary = amr_is_binary .or. g(n-7:n) == '.fits.gz'
    end function amr_is_binary
    subroutine amr_read_binary_file()
      interface
         function lart_amr_read_binary(path, nleaf, boxlen, origin, data) bind(C, name='lart_amr_read_binary') result(rc)
           import; character(kind=c_char), intent(in) :: path(*); integer(c_int), intent(out) :: nleaf
           real(c_double), intent(out) :: boxlen, origin(3); type(c_ptr), intent(out) :: data; integer(c_int) :: rc
         end function
         subroutine lart_free(p) bind(C, name='lart_free')
           import; type(c_ptr), value :: p
         end subroutine
      end interface
      type(c_ptr) :: cdata
      real(c_double), pointer :: tab(:,:)
      real(c_double) :: org(3)
      integer(c_int) :: nl, rc
      rc = lart_amr_read_binary(trim(par%amr_file)//c_null_char, nl, boxlen, org, cdata)
      if (rc /= 0) then
         write(*,'(3a,i0,a)') 'generic_amr_read_binary: cannot read ', trim(par%amr_file), ' (code ', rc, &
              ': 1 open, 2 short file, 3 no leaf table / BOXLEN, 4 columns, 6 no HDF5 in this build)'; stop 1
      endif
      if (org(1) /= org(2) .or. org(1) /= org(3)) then
         write(*,'(a)') 'ERROR: an AMR box with different ORIGINX/Y/Z is not supported by the MI355X photon path (cubic box [lo,hi]^3).'; stop 1
      endif
      n = nl; amr_origin = org(1)
      call c_f_pointer(cdata, tab, [n, 9])
      allocate(lx(n), ly(n), lz(n), lev(n), nH(n), Tl(n), lvx(n), lvy(n), lvz(n))
      lx = tab(:,1); ly = tab(:,2); lz = tab(:,3); lev = nint(tab(:,4)); nH = tab(:,5); Tl = tab(:,6)
      lvx = tab(:,7); lvy = tab(:,8); lvz = tab(:,9)
      call lart_free(cdata)
    end subroutine amr_read_binary_file
  end subroutine grid_create_amr

  subroutine amr_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: n
    if (rc == 0) return
    p = lart_host_amr_last_error()
    write(*,'(3a)', advance='no') 'ERROR in ', where, ': '
    if (c_associated(p)) then
       call c_f_pointer(p, msg, [512])
       do n = 1, 512
          if (msg(n) == c_null_char) exit
          write(*,'(a)', advance='no') msg(n)
       enddo
    endif
    write(*,*)
    stop 1
  end subroutine amr_check

  !--- amr_find_leaf: octree_mod.f90:149-171
  function amr_find_leaf(x, y, z) result(il)
    real(wp), intent(in) :: x, y, z
    integer :: il, ic, io
    il = 0
    if (x < amr_lo .or. x > amr_hi .or. y < amr_lo .or. y > amr_hi .or. z < amr_lo .or. z > amr_hi) return
    ic = 1
    do
       if (amr_ileaf(ic) > 0) then
          il = amr_ileaf(ic); return
       endif
       io = 1
       if (x >= amr_cx(ic)) io = io + 1
       if (y >= amr_cy(ic)) io = io + 2
       if (z >= amr_cz(ic)) io = io + 4
       ic = amr_children(io, ic)
       if (ic == 0) return
    enddo
  end function amr_find_leaf

  !--- create_JPa_amr_mem: grid_mod_amr.f90:543-724 (number of position bins and gas volume per bin by ns^3 sub-sampling)
  subroutine create_JPa_amr_mem(grid)
    type(grid_type), intent(in) :: grid
    integer :: nr, nz, il, ic, ns, i, j, k, ir, iz, iz1, iz2
    real(wp) :: rmax, hc, vleaf, nadd, dr, dz, wsub, xs, ys, zs, z1, z2, ov
    if (.not. (par%calc_J .or. par%calc_P .or. par%calc_Pnew)) return
    rmax = par%rmax
    if (rmax <= 0.0_wp) rmax = (amr_hi - amr_lo)*0.5_wp
    nr = min(512, max(1, 2**max(0, amr_levelmax - 1))); nz = min(512, max(1, 2**max(0, amr_levelmax)))
    dr = rmax/nr; dz = (amr_hi - amr_lo)/nz
    nadd = 1.0_wp                                   ! grid_mod_amr.f90:577-588: only the mirror symmetries fold, the periodic slab is stored whole
    if (par%xyz_symmetry) then
       nadd = 8.0_wp
    else if (par%xy_symmetry) then
       nadd = 4.0_wp
    endif
    select case (par%geometry_JPa)
    case (3);     amr_nbins = amr_nleaf; amr_dbin = 0.0_wp
    case (-1);    amr_nbins = nz;        amr_dbin = dz
    case default; amr_nbins = nr;        amr_dbin = dr
    end select
    allocate(amr_vol(amr_nbins)); amr_vol = 0.0_wp
    do il = 1, amr_nleaf
       ic = amr_icell_of_leaf(il); hc = amr_ch(ic); vleaf = (2.0_wp*hc)**3*nadd
       if (par%geometry_JPa == 3) then
          amr_vol(il) = vleaf; cycle
       endif
       if (.not. (amr_rhokap(il) > 0.0_wp)) cycle
       if (par%geometry_JPa == -1) then
          z1 = amr_cz(ic) - hc; z2 = amr_cz(ic) + hc
          iz1 = max(1, int(floor((z1 - amr_lo)/dz)) + 1); iz2 = min(nz, int(floor((z2 - amr_lo)/dz)) + 1)
          do iz = iz1, iz2
             ov = min(z2, amr_lo + iz*dz) - max(z1, amr_lo + (iz-1)*dz)
             if (ov > 0.0_wp) amr_vol(iz) = amr_vol(iz) + vleaf*ov/(2.0_wp*hc)
          enddo
       else
          ns = max(2, min(8, nint(2.0_wp*hc/dr))); wsub = vleaf/real(ns, wp)**3
          do k = 1, ns; do j = 1, ns; do i = 1, ns
             xs = amr_cx(ic) + (2.0_wp*i - ns - 1.0_wp)/ns*hc; ys = amr_cy(ic) + (2.0_wp*j - ns - 1.0_wp)/ns*hc; zs = amr_cz(ic) + (2.0_wp*k - ns - 1.0_wp)/ns*hc
             ir = int(floor(sqrt(xs*xs + ys*ys + zs*zs)/dr)) + 1
             if (ir >= 1 .and. ir <= nr) amr_vol(ir) = amr_vol(ir) + wsub
          enddo; enddo; enddo
       endif
    enddo
    if (par%calc_J) then
       allocate(prof_J1(grid%nxfreq, amr_nbins)); prof_J1 = 0.0_wp
    endif
    if (par%calc_P) then
       allocate(prof_P1(amr_nbins)); prof_P1 = 0.0_wp
    endif
    if (par%calc_Pnew) then
       allocate(prof_P1_new(amr_nbins)); prof_P1_new = 0.0_wp
    endif
    if (par%calc_J)    write(6,'(a,i0)') 'AMR CALCJ enabled, geometry_JPa = ', par%geometry_JPa
    if (par%calc_P)    write(6,'(a,i0)') 'AMR CALCP enabled, geometry_JPa = ', par%geometry_JPa
    if (par%calc_Pnew) write(6,'(a,i0)') 'AMR CALCPnew enabled, geometry_JPa = ', par%geometry_JPa
  end subroutine create_JPa_amr_mem

  !--- create_JPa_mem: grid_mod_car.f90:1242-1434 (profile bin of every cell + the number of gas cells per bin)
  subroutine create_JPa_mem(grid)
    type(grid_type), intent(in) :: grid
    integer :: nx, ny, nz
    real(wp), allocatable :: cnt(:)
    if (.not. (par%calc_J .or. par%calc_P .or. par%calc_Pnew)) return
    nx = grid%nx; ny = grid%ny; nz = grid%nz
    allocate(car_ind_bin(nx, ny, nz), cnt(nx*ny*nz))
    call setup_check(lart_host_jpa_car(int(par%geometry_JPa, c_int32_t), merge(1, 0, par%xyz_symmetry), nx, ny, nz, grid%xface, grid%yface, grid%zface, &
                                       grid%xmax, grid%ymax, grid%zmax, grid%rhokap, car_nbins, car_ind_bin, cnt), 'create_JPa_mem')
    allocate(car_ncount(car_nbins)); car_ncount = cnt(1:car_nbins)
    if (par%calc_J) then
       allocate(prof_J1(grid%nxfreq, car_nbins)); prof_J1 = 0.0_wp
    endif
    if (par%calc_P) then
       allocate(prof_P1(car_nbins)); prof_P1 = 0.0_wp
    endif
    if (par%calc_Pnew) then
       allocate(prof_P1_new(car_nbins)); prof_P1_new = 0.0_wp
    endif
    if (line%line_type == 8 .and. par%calc_P) then                      ! the conversion-rate profile of ly_beta (add_to_Pconv)
       allocate(prof_Pconv(car_nbins)); prof_Pconv = 0.0_wp
    endif
  end subroutine create_JPa_mem

  !--- car_setup_freq_grid (grid_mod_car.f90:1442-1548) through lart_host_freq_grid: the spectral axis from the wavelength / velocity window or from
  !--- (a tau)^(1/3), the band-2 axis of ly_beta; core_skip_scale: also grid%xcrit of the Cartesian grid (:1186-1219; the octree sets its own)
  subroutine car_setup_freq_grid(grid, core_skip_scale)
    type(grid_type), intent(inout) :: grid
    logical, intent(in), optional :: core_skip_scale
    type(lart_host_freq) :: f
    logical :: cs
    cs = .false.
    if (present(core_skip_scale)) cs = core_skip_scale
    f%nxfreq = par%nxfreq; f%nwavelength = par%nwavelength; f%nvelocity = par%nvelocity; f%continuum = merge(1, 0, trim(par%spectral_type) == 'continuum')
    f%line_type = line%line_type; f%nxfreq_Ha = par%nxfreq_Ha; f%core_skip_global = merge(1, 0, par%core_skip_global)
    f%wavelength_min = par%wavelength_min; f%wavelength_max = par%wavelength_max; f%velocity_min = par%velocity_min; f%velocity_max = par%velocity_max
    f%xfreq_min = par%xfreq_min; f%xfreq_max = par%xfreq_max; f%vtherm = vtherm_total(par%temperature); f%wavelength0 = line%wavelength0
    f%wavelength0_Ha = line%wavelength0_Ha; f%Dfreq_ref = grid%Dfreq_ref; f%DnuHK_Hz = line%DnuHK_Hz; f%voigt_amean = grid%voigt_amean
    f%tauhomo = par%tauhomo; f%taumax = par%taumax; f%Vexp = par%Vexp; f%xfreq_max_Ha = par%xfreq_max_Ha
    f%cells_per_radius = 1.0_wp
    if (cs .and. grid%dx > 0.0_wp) f%cells_per_radius = grid%xmax/grid%dx
    call setup_check(lart_host_freq_grid(f), 'car_setup_freq_grid')
    par%atau3 = f%atau3; par%nxfreq = f%nxfreq; par%nwavelength = f%nwavelength; par%nvelocity = f%nvelocity; par%xfreq_min = f%xfreq_min; par%xfreq_max = f%xfreq_max
    grid%nxfreq = f%nxfreq; grid%dxfreq = f%dxfreq; grid%xfreq_min = f%xfreq_min; grid%xfreq_max = f%xfreq_max; grid%dwave = f%dwave
    if (line%line_type == 8) then
       par%nxfreq_Ha = f%nxfreq_Ha; grid%nxfreq_Ha = f%nxfreq_Ha
       grid%xfreq_min_Ha = f%xfreq_min_Ha; grid%xfreq_max_Ha = f%xfreq_max_Ha; grid%dxfreq_Ha = f%dxfreq_Ha; grid%dwave_Ha = f%dwave_Ha
    endif
    if (cs) then
       grid%xcrit = f%xcrit; grid%xcrit2 = f%xcrit2
    endif
    allocate(grid%xfreq(grid%nxfreq), grid%velocity(grid%nxfreq), grid%wavelength(grid%nxfreq))
    call setup_check(lart_host_freq_axis(grid%nxfreq, grid%xfreq_min, grid%dxfreq, f%vtherm, line%wavelength0, grid%xfreq, grid%velocity, grid%wavelength), 'freq axis')
  end subroutine car_setup_freq_grid

  !--- observer_create_outside (observer_rect.f90:24-326, rotation centre at the origin) / observer_create_inside (observer_heal.f90:24-80) through
  !--- lart_host_observers; par%alpha/beta/gamma, par%obsx/y/z, par%distance, par%dxim/dyim come back completed, as the reference leaves them
  subroutine observer_create()
    type(lart_host_obs) :: o
    real(c_double) :: xyz(3*MAX_OBSERVERS), rm(9*MAX_OBSERVERS)
    integer :: i
    if (.not. par%save_peeloff) then
       par%nobs = 0; return
    endif
    o%nside = par%nside; o%sphere = merge(1, 0, trim(par%geometry) == 'sphere'); o%nxim = par%nxim; o%nyim = par%nyim
    o%xmax = par%xmax; o%ymax = par%ymax; o%zmax = par%zmax; o%rmax = par%rmax; o%distance = par%distance; o%dxim = par%dxim; o%dyim = par%dyim
    call setup_check(lart_host_observers(o, par%alpha, par%beta, par%gamma, par%obsx, par%obsy, par%obsz, xyz, rm), 'observer_create')
    par%nobs = o%nobs
    allocate(observer(par%nobs))
    do i = 1, par%nobs
       observer(i)%x = xyz(3*i-2); observer(i)%y = xyz(3*i-1); observer(i)%z = xyz(3*i)
       observer(i)%rmatrix = transpose(reshape(rm(9*i-8:9*i), [3, 3]))   ! the library's rows are Fortran's first index
       observer(i)%steradian_pix = o%steradian_pix
    enddo
    if (par%nside > 0) then
       par%npix = o%npix
       do i = 1, par%nobs
          observer(i)%rmatrix = 0.0_wp; observer(i)%dxim = 0.0_wp; observer(i)%dyim = 0.0_wp; observer(i)%nxim = par%npix; observer(i)%nyim = 1
       enddo
       call alloc_peel(par%npix, 1)
       return
    endif
    par%distance = o%distance; par%dxim = o%dxim; par%dyim = o%dyim
    do i = 1, par%nobs
       observer(i)%dxim = par%dxim; observer(i)%dyim = par%dyim; observer(i)%nxim = par%nxim; observer(i)%nyim = par%nyim
    enddo
    call alloc_peel(par%nxim, par%nyim)                                                         ! :287-324
  contains
    subroutine alloc_peel(nxi, nyi)
      integer, intent(in) :: nxi, nyi
      integer :: no
      no = par%nobs
      if (par%save_peeloff_2D) then
         allocate(obs_scatt_2D(nxi,nyi,no), obs_direc_2D(nxi,nyi,no)); obs_scatt_2D = 0; obs_direc_2D = 0
         if (par%save_direc0) then
            allocate(obs_direc0_2D(nxi,nyi,no)); obs_direc0_2D = 0
         endif
         if (par%use_stokes) then
            allocate(obs_I_2D(nxi,nyi,no), obs_Q_2D(nxi,nyi,no), obs_U_2D(nxi,nyi,no), obs_V_2D(nxi,nyi,no))
         endif
      endif
      if (par%save_peeloff_3D) then
         allocate(obs_scatt(par%nxfreq,nxi,nyi,no), obs_direc(par%nxfreq,nxi,nyi,no)); obs_scatt = 0; obs_direc = 0
         if (par%save_direc0) then
            allocate(obs_direc0(par%nxfreq,nxi,nyi,no)); obs_direc0 = 0
         endif
         if (par%save_dust_scattered) then
            allocate(obs_scatt_dust(par%nxfreq,nxi,nyi,no)); obs_scatt_dust = 0
         endif
         if (line%line_type == 8) then                                   ! observer(:)%peel_Ha: the band-2 cube of ly_beta (define.f90:567-568)
            allocate(obs_peel_Ha(par%nxfreq_Ha,nxi,nyi,no)); obs_peel_Ha = 0
         endif
         if (par%use_stokes) then
            allocate(obs_I(par%nxfreq,nxi,nyi,no), obs_Q(par%nxfreq,nxi,nyi,no), obs_U(par%nxfreq,nxi,nyi,no), obs_V(par%nxfreq,nxi,nyi,no))
         endif
      endif
    end subroutine alloc_peel
  end subroutine observer_create

  !--- output_normalize_outside (output_sum_rect.f90:151-487): the divisors come from lart_host_norm_scales (Jout/Jin/Jabs/Jmu, the peel cubes and
  !--- images, and for ly_beta the band-2 sections with the H-alpha bin unit :210-236, :433-435)
  subroutine output_normalize(grid)
    type(grid_type), intent(inout) :: grid
    type(lart_host_norm) :: q
    real(wp) :: intensity_bin_unit
    par%nscatt_dust = par%nscatt_dust/par%nphotons; par%nscatt_gas = par%nscatt_gas/par%nphotons; par%nscatt_tot = par%nscatt_gas + par%nscatt_dust
    intensity_bin_unit = merge(grid%dwave, grid%dxfreq, par%intensity_unit == 1)
    q%xy_periodic = merge(1, 0, par%xy_periodic); q%sphere = merge(1, 0, trim(par%geometry) == 'sphere'); q%nphotons = real(par%nphotons, wp)
    q%bin_unit = intensity_bin_unit; q%bin_unit_Ha = intensity_bin_unit
    if (line%line_type == 8) q%bin_unit_Ha = merge(grid%dwave_Ha, grid%dxfreq_Ha, par%intensity_unit == 1)
    q%distance2cm = par%distance2cm; q%rmax = grid%rmax; q%xmax = grid%xmax; q%ymax = grid%ymax; q%zmax = grid%zmax
    q%amr_L_box = merge(amr_hi - amr_lo, 0.0_wp, par%use_amr_grid); q%steradian_pix = 0.0_wp
    if (par%nobs > 0 .and. allocated(observer)) q%steradian_pix = observer(1)%steradian_pix
    call setup_check(lart_host_norm_scales(q), 'output_normalize')
    grid%Jout = grid%Jout/q%scale_J
    if (par%save_Jin) grid%Jin = grid%Jin/q%scale_J
    if (par%save_Jmu) grid%Jmu = grid%Jmu*par%nmu/q%scale_J
    if (par%DGR > 0.0_wp .and. par%save_Jabs) grid%Jabs = grid%Jabs/q%scale_J
    if (allocated(grid%Jout_Ha)) grid%Jout_Ha = grid%Jout_Ha/q%scale_J_Ha
    if (allocated(grid%Jabs_Ha)) grid%Jabs_Ha = grid%Jabs_Ha/q%scale_J_Ha
    if (par%save_peeloff_2D .and. par%nobs > 0) then
       obs_scatt_2D = obs_scatt_2D/q%scale_image; obs_direc_2D = obs_direc_2D/q%scale_image
       if (allocated(obs_direc0_2D)) obs_direc0_2D = obs_direc0_2D/q%scale_image
       if (par%use_stokes) then
          obs_I_2D = obs_I_2D/q%scale_image; obs_Q_2D = obs_Q_2D/q%scale_image; obs_U_2D = obs_U_2D/q%scale_image; obs_V_2D = obs_V_2D/q%scale_image
       endif
    endif
    if (par%save_peeloff_3D .and. par%nobs > 0) then
       obs_scatt = obs_scatt/q%scale_cube; obs_direc = obs_direc/q%scale_cube
       if (allocated(obs_direc0)) obs_direc0 = obs_direc0/q%scale_cube
       if (allocated(obs_scatt_dust)) obs_scatt_dust = obs_scatt_dust/q%scale_cube
       if (allocated(obs_peel_Ha)) obs_peel_Ha = obs_peel_Ha/q%scale_cube_Ha
       if (par%use_stokes) then
          obs_I = obs_I/q%scale_cube; obs_Q = obs_Q/q%scale_cube; obs_U = obs_U/q%scale_cube; obs_V = obs_V/q%scale_cube
       endif
    endif
    !--- Cartesian profiles (output_sum_rect.f90:275-360): per gas cell of the bin, per unit volume
    if (car_nbins > 0) then
       block
         integer :: ib
         real(wp) :: dVol, f
         dVol = grid%dx*grid%dy*grid%dz*par%distance2cm**2
         f = 1.0_wp
         if (par%xy_periodic .and. (par%geometry_JPa == 3 .or. par%geometry_JPa == -1)) &
            f = (grid%xmax - grid%xmin)*(grid%ymax - grid%ymin)*par%distance2cm**2
         do ib = 1, car_nbins
            if (car_ncount(ib) > 0.0_wp) then
               if (allocated(prof_J1))     prof_J1(:,ib)   = prof_J1(:,ib)/car_ncount(ib)*(f/(fourpi*dVol*par%nphotons*intensity_bin_unit))
               if (allocated(prof_P1))     prof_P1(ib)     = prof_P1(ib)/car_ncount(ib)*(f/(dVol*par%nphotons))
               if (allocated(prof_P1_new)) prof_P1_new(ib) = prof_P1_new(ib)/car_ncount(ib)*(f/(dVol*par%nphotons))
               if (allocated(prof_Pconv))  prof_Pconv(ib)  = prof_Pconv(ib)/car_ncount(ib)*(f/(dVol*par%nphotons))
            else
               if (allocated(prof_J1))     prof_J1(:,ib)   = 0.0_wp
               if (allocated(prof_P1))     prof_P1(ib)     = 0.0_wp
               if (allocated(prof_P1_new)) prof_P1_new(ib) = 0.0_wp
               if (allocated(prof_Pconv))  prof_Pconv(ib)  = 0.0_wp
            endif
         enddo
       end block
    endif
    !--- octree profiles: volume-weighted normalisation (output_sum_heal.f90:476-590)
    if (allocated(amr_vol)) then
       block
         integer :: ib
         real(wp) :: d2
         d2 = par%distance2cm**2
         !--- periodic slab (one photon per cm^2): the slab area replaces 1/d2 for the leaf and plane bins (output_sum_rect.f90:874-1000)
         if (par%xy_periodic .and. (par%geometry_JPa == 3 .or. par%geometry_JPa == -1)) d2 = 1.0_wp/(amr_hi - amr_lo)**2
         do ib = 1, amr_nbins
            if (amr_vol(ib) > 0.0_wp) then
               if (allocated(prof_J1))     prof_J1(:,ib)   = prof_J1(:,ib)/(fourpi*amr_vol(ib)*d2*par%nphotons*intensity_bin_unit)
               if (allocated(prof_P1))     prof_P1(ib)     = prof_P1(ib)/(amr_vol(ib)*d2*par%nphotons)
               if (allocated(prof_P1_new)) prof_P1_new(ib) = prof_P1_new(ib)/(amr_vol(ib)*d2*par%nphotons)
               if (allocated(prof_Pconv))  prof_Pconv(ib)  = prof_Pconv(ib)/(amr_vol(ib)*d2*par%nphotons)
            endif
         enddo
       end block
    endif
  end subroutine output_normalize
end module lart_host
