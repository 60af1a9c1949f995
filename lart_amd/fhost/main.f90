!------------------------------------------------------------------------------
! LaRT_amd.x <input.in> [--dry-run | --dry-run-write]
! Same sequence as the reference's main.f90:19-69; run_simulation and output_reduce cross the
! C-ABI (include/lart_gpu.h) into the HIP kernels.  --dry-run stops after the grid/observer set-up
! (prints the same grid summary the reference prints) so the host logic can be checked without a GPU.
!------------------------------------------------------------------------------
program LaRT_amd
  use, intrinsic :: iso_c_binding
  use lart_define
  use lart_host
  use lart_gpu_c
  use lart_fits
  use lart_out
  implicit none
  type(grid_type), target :: grid
  type(lart_params) :: cpar
  type(lart_tallies) :: t
  type(c_ptr) :: ctx
  type(c_ptr), allocatable :: ctxs(:)
  character(len=256) :: infile, arg
  logical :: dry_run, dry_write
  real(wp) :: nph_key = 0.0_wp                   ! the photon count a peel-off file's keyword reports (this run, or this run + the merged file)
  interface
     function lart_out_read(path, section, name, dst, n) bind(C, name='lart_out_read') result(rc)
       import; character(kind=c_char), intent(in) :: path(*), section(*), name(*); real(c_double), intent(out) :: dst(*); integer(c_long_long), value :: n; integer(c_int) :: rc
     end function
     function lart_out_read_key(path, section, key, v) bind(C, name='lart_out_read_key') result(rc)
       import; character(kind=c_char), intent(in) :: path(*), section(*), key(*); real(c_double), intent(out) :: v; integer(c_int) :: rc
     end function
  end interface
  integer :: i, ndev, ngpu, r
  integer(int64) :: done, active, left, t0, t1, rate, run_seed, cnt_r
  real(wp), allocatable, target :: xyz(:), rmat(:)
  real(wp) :: dtime

  if (command_argument_count() < 1) then
     call get_command_argument(0, arg)
     write(*,*) 'Usage: ', trim(arg), ' input_file [--dry-run]'
     stop
  endif
  call get_command_argument(1, infile)
  dry_run = .false.; dry_write = .false.
  do i = 2, command_argument_count()
     call get_command_argument(i, arg)
     if (trim(arg) == '--dry-run') dry_run = .true.
     if (trim(arg) == '--dry-run-write') then       ! set-up + the output files with empty tallies (container check without a GPU)
        dry_run = .true.; dry_write = .true.
     endif
  enddo
  call system_clock(t0, rate)
  ngpu = 1

  call read_input(trim(infile))
  !--- par%iseed = 0: seed from the operating system, as init_random_mt_seed does (random_mt.f90:889-947); printed and written
  !    to the output header (keyword ISEED) so that the run can be repeated.  A non-zero iseed is used as it is.
  run_seed = int(par%iseed, int64)
  if (par%iseed == 0) then
     if (lart_host_entropy_seed(run_seed) /= 0) run_seed = 1_int64
     par%iseed = int(mod(run_seed, 2147483647_int64))      ! the host-side MT stream (clump placement) takes a default integer
     if (par%iseed == 0) par%iseed = 1
     write(*,'(a,i0)') 'par%iseed = 0: seeded from the operating system, seed = ', run_seed
  endif
  par%run_seed = run_seed
  if (par%use_clump_medium) then            ! main.f90:25-27
     call grid_create_clump(grid)
     if (par%save_clump_info) call write_clumps_info()       ! grid_mod_clump.f90:123-129
  else if (par%use_amr_grid) then           ! main.f90:28-30
     call grid_create_amr(grid)
  else
     call grid_create(grid)
     call create_JPa_mem(grid)
     if (car_nbins > 0) write(*,'(a,i0,a,i0,a,es13.5)') 'profile bins: geometry_JPa = ', par%geometry_JPa, '  nbins = ', car_nbins, '  sum(ncount) = ', sum(car_ncount)
  endif
  call observer_create()
  if (par%save_peeloff .and. par%nside > 0) write(*,'(a,i0,a,i0)') 'HEALPix inside observers: nside = ', par%nside, '  npix = ', par%npix
  if (par%save_peeloff) write(*,'(a,i4,a,2i6,a,2es13.5)') 'observers: ', par%nobs, '  image ', par%nxim, par%nyim, '  pixel (deg) ', par%dxim, par%dyim
  write(*,'(a,2f10.3,i6)') 'xfreq_min, xfreq_max, nxfreq : ', grid%xfreq_min, grid%xfreq_max, grid%nxfreq
  if (dry_run) then
     out_bitpix = par%out_bitpix
     if (dry_write) call write_output(trim(par%out_file), grid)
     if (dry_write .and. par%save_sightline_tau .and. par%save_peeloff .and. par%nobs > 0) call make_sightline_tau(.false.)
     write(*,'(a)') '--dry-run: set-up only, no transport.'
     stop
  endif

  !--- hand the read-only tables to the device (replaces the MPI shared-memory windows, memory_mod_mpi.f90:96-130)
  ndev = lart_gpu_device_count()
  if (ndev < 1) then
     write(*,'(a)') 'ERROR: no MI355X/HIP device visible; this build has no CPU transport path.'; stop 1
  endif
  call fill_params(cpar)
  out_bitpix = par%out_bitpix
  !--- one context per GPU, all driven by this host thread (par%ngpu: 0 = every visible device; HIP_VISIBLE_DEVICES narrows it).
  !    Replaces `mpirun -np N`: context r transports the photons r, r+N, ... (run_simulation_mod.f90:150-151).
  ngpu = par%ngpu
  if (ngpu <= 0) ngpu = ndev
  if (ngpu > ndev) then
     write(*,'(a,i0,a,i0,a)') 'ERROR: par%ngpu = ', ngpu, ' but only ', ndev, ' HIP device(s) are visible.'; stop 1
  endif
  if (par%save_all_photons .and. ngpu > 1) then
     write(*,'(a)') 'par%save_all_photons: the per-photon table is kept by one context; using one GPU.'
     ngpu = 1
  endif
  if (int(ngpu, int64) > max(par%nphotons, 1_int64)) ngpu = int(max(par%nphotons, 1_int64))
  allocate(ctxs(ngpu))
  do r = 1, ngpu
     call lart_check(lart_gpu_create(ctxs(r), cpar, int(r - 1, c_int)), 'lart_gpu_create')
     call setup_device(ctxs(r))
  enddo
  ctx = ctxs(1)
  write(*,'(a,i0)') 'GPUs used                    : ', ngpu

  if (par%save_sightline_tau .and. par%save_peeloff .and. par%nobs > 0) call make_sightline_tau(.true.)      ! main.f90:34-37

  !--- run_simulation (main.f90:42): bounded slices so progress can be printed like par%nprint does
  call time_stamp(dtime)
  write(6,'(a,f8.3,a)') '---> Now starting simulation...  @ ', dtime/60.0_wp, ' mins'
  if (par%save_all_photons) call lart_check(lart_gpu_enable_all_photons(ctx, 1_c_int32_t), 'lart_gpu_enable_all_photons')
  do r = 1, ngpu
     cnt_r = 0
     if (par%nphotons >= r) cnt_r = (par%nphotons - r)/ngpu + 1
     call lart_check(lart_gpu_begin(ctxs(r), int(r, c_int64_t), int(ngpu, c_int64_t), cnt_r, run_seed), 'lart_gpu_begin')
  enddo
  do
     do r = 1, ngpu        ! launches are asynchronous: one slice is enqueued on every GPU before the first wait
        call lart_check(lart_gpu_step(ctxs(r), 1_c_int32_t, int(par%gpu_events_per_step, c_int32_t)), 'lart_gpu_step')
     enddo
     left = 0
     do r = 1, ngpu
        call lart_check(lart_gpu_sync(ctxs(r), done, active), 'lart_gpu_sync')
        left = left + active
     enddo
     if (left == 0) exit
  enddo
  call time_stamp(dtime)
  write(6,'(es14.5,a,f8.3,a)') dble(par%nphotons), ' photons calculated: ', dtime/60.0_wp, ' mins'

  !--- output_reduce (main.f90:46): one RCCL reduce(sum, f64) of the fused tally buffers onto GPU 0 (reduce_mem,
  !    memory_mod_mpi.f90:366-394), then the device->host copy
  write(6,'(a)') '---> Now Gathering Results...'
  if (ngpu > 1) call lart_check(lart_gpu_reduce_tallies_multi(ctxs, int(ngpu, c_int32_t), 0_c_int32_t), 'lart_gpu_reduce_tallies_multi')
  call null_tallies(t)
  t%Jout = c_loc(grid%Jout); t%Jin = c_loc(grid%Jin)
  if (par%save_Jabs) t%Jabs = c_loc(grid%Jabs)
  if (par%save_Jmu)  t%Jmu  = c_loc(grid%Jmu)
  if (allocated(obs_scatt)) then
     t%scatt = c_loc(obs_scatt); t%direc = c_loc(obs_direc)
  endif
  if (allocated(obs_I)) then
     t%I = c_loc(obs_I); t%Q = c_loc(obs_Q); t%U = c_loc(obs_U); t%V = c_loc(obs_V)
  endif
  if (allocated(obs_scatt_2D)) then
     t%scatt_2D = c_loc(obs_scatt_2D); t%direc_2D = c_loc(obs_direc_2D)
  endif
  if (allocated(obs_I_2D)) then
     t%I_2D = c_loc(obs_I_2D); t%Q_2D = c_loc(obs_Q_2D); t%U_2D = c_loc(obs_U_2D); t%V_2D = c_loc(obs_V_2D)
  endif
  if (allocated(obs_direc0))     t%direc0     = c_loc(obs_direc0)
  if (allocated(obs_direc0_2D))  t%direc0_2D  = c_loc(obs_direc0_2D)
  if (allocated(obs_scatt_dust)) t%scatt_dust = c_loc(obs_scatt_dust)
  if (allocated(prof_J1))     t%J1     = c_loc(prof_J1)
  if (allocated(prof_P1))     t%P1     = c_loc(prof_P1)
  if (allocated(prof_P1_new)) t%P1_new = c_loc(prof_P1_new)
  if (allocated(grid%Jout_Ha)) then
     t%Jout_Ha = c_loc(grid%Jout_Ha); t%Jabs_Ha = c_loc(grid%Jabs_Ha)
  endif
  if (allocated(obs_peel_Ha)) t%peel_Ha = c_loc(obs_peel_Ha)
  if (allocated(prof_Pconv))  t%Pconv   = c_loc(prof_Pconv)
  call lart_check(lart_gpu_get_tallies(ctx, t), 'lart_gpu_get_tallies')
  par%nscatt_gas = t%nscatt_gas; par%nscatt_dust = t%nscatt_dust
  if (line%line_type == 8) then                                         ! the weight budget of the two bands (define.f90:471-475)
     par%W_conv = t%aux(1); par%W_esc1 = t%aux(2); par%W_abs1 = t%aux(3); par%W_esc2 = t%aux(4); par%W_abs2 = t%aux(5)
  else if (line%h2%n > 0) then                                          ! H2: destroyed / re-emitted Ly-alpha weight (define.f90:490-491)
     par%W_H2abs = t%aux(1); par%W_H2scat = t%aux(2)
  endif

  call output_normalize(grid)
  call time_stamp(dtime); par%exetime = dtime/60.0_wp
  call write_output(trim(par%out_file), grid)
  if (par%save_all_photons) call write_output_allph()

  write(6,'(a,es12.4)')   'Average Number of scattering : ', par%nscatt_tot
  write(6,'(a,2es12.4)')  'Nscatt_dust, Nscatt_gas      : ', par%nscatt_dust, par%nscatt_gas
  write(6,'(a,es12.4)')   'Scattering events per second : ', t%n_events/max(dtime, 1e-9_wp)
  write(6,'(a,es12.4)')   'Scattering events per photon : ', t%n_events/real(max(par%nphotons, 1_int64), wp)      ! unweighted count (the <N_scatt> of the reference's logs before v2.00)
  if (line%line_type == 8) then                                         ! output_sum_rect.f90:446-466: band1_esc + band1_abs + conv = 1, band2_esc + band2_abs = conv
     write(*,'(a)')        '--- ly_beta bookkeeping (weights per source photon) ---'
     write(*,'(a,es14.6)') '  band-1 (Ly-beta) escaped        : ', par%W_esc1/par%nphotons
     write(*,'(a,es14.6)') '  band-1 (Ly-beta) dust-absorbed  : ', par%W_abs1/par%nphotons
     write(*,'(a,es14.6)') '  conversions (3p->2s)            : ', par%W_conv/par%nphotons
     write(*,'(a,es14.6)') '  band-2 (H-alpha) escaped        : ', par%W_esc2/par%nphotons
     write(*,'(a,es14.6)') '  band-2 (H-alpha) dust-absorbed  : ', par%W_abs2/par%nphotons
     write(*,'(a,es14.6)') '  band1_esc + band1_abs + conv (=1)     : ', (par%W_esc1 + par%W_abs1 + par%W_conv)/par%nphotons
     write(*,'(a,es14.6)') '  band2_esc + band2_abs (= conv)        : ', (par%W_esc2 + par%W_abs2)/par%nphotons
  else if (line%h2%n > 0) then
     write(*,'(a,es14.6)') 'H2: Ly-alpha weight destroyed by H2 per photon : ', par%W_H2abs/par%nphotons
     write(*,'(a,es14.6)') 'H2: Ly-alpha weight re-emitted by H2 per photon: ', par%W_H2scat/par%nphotons
  endif
  write(6,'(a,f8.3,a)')   'Total Excution Time          : ', par%exetime, ' mins'
  do r = 1, ngpu
     call lart_check(lart_gpu_destroy(ctxs(r)), 'lart_gpu_destroy')
  enddo

contains
  !--- hand the read-only tables to one device
  subroutine setup_device(ctx)
    type(c_ptr), intent(in) :: ctx
    integer :: i
  if (par%use_amr_grid) then
       if (par%DGR > 0.0_wp) then
          call lart_check(lart_gpu_set_grid_amr(ctx, amr_ncells, amr_nleaf, amr_lo, amr_hi, amr_cx, amr_cy, amr_cz, amr_ch, amr_children, amr_neighbor, &
               amr_ileaf, amr_icell_of_leaf, amr_rhokap, amr_voigt_a, amr_Dfreq, amr_vfx, amr_vfy, amr_vfz, c_loc(amr_rhokapD)), 'lart_gpu_set_grid_amr')
       else
          call lart_check(lart_gpu_set_grid_amr(ctx, amr_ncells, amr_nleaf, amr_lo, amr_hi, amr_cx, amr_cy, amr_cz, amr_ch, amr_children, amr_neighbor, &
               amr_ileaf, amr_icell_of_leaf, amr_rhokap, amr_voigt_a, amr_Dfreq, amr_vfx, amr_vfy, amr_vfz, c_null_ptr), 'lart_gpu_set_grid_amr')
       endif
       if (amr_nbins > 0) call lart_check(lart_gpu_set_JPa_amr(ctx, merge(1, 0, par%calc_J), merge(1, 0, par%calc_P), merge(1, 0, par%calc_Pnew), &
            par%geometry_JPa, amr_nbins, amr_dbin, line%cross0), 'lart_gpu_set_JPa_amr')
    else if (par%use_clump_medium) then
       if (par%DGR > 0.0_wp) then
          call lart_check(lart_gpu_set_clumps(ctx, N_clumps, cl_x, cl_y, cl_z, cl_vx, cl_vy, cl_vz, cl_radius, cl_rhokap, c_loc(cl_rhokapD), &
               cl_voigt_a, cl_Dfreq, sphere_R, cl_Dfreq_ref, cl_voigt_a_ref, cgx, cg_xmin, cg_dx, cg_start, cg_list), 'lart_gpu_set_clumps')
       else
          call lart_check(lart_gpu_set_clumps(ctx, N_clumps, cl_x, cl_y, cl_z, cl_vx, cl_vy, cl_vz, cl_radius, cl_rhokap, c_null_ptr, &
               cl_voigt_a, cl_Dfreq, sphere_R, cl_Dfreq_ref, cl_voigt_a_ref, cgx, cg_xmin, cg_dx, cg_start, cg_list), 'lart_gpu_set_clumps')
       endif
       if (has_overlap) call lart_check(lart_gpu_set_clump_overlap(ctx, 1), 'lart_gpu_set_clump_overlap')      ! setup_clump_overlap: setup.f90:1053-1082
    else if (par%DGR > 0.0_wp) then
       call lart_check(lart_gpu_set_grid_car(ctx, grid%nx, grid%ny, grid%nz, grid%xmin, grid%ymin, grid%zmin, grid%dx, grid%dy, grid%dz, &
            grid%i0, grid%j0, grid%k0, grid%xface, grid%yface, grid%zface, grid%rhokap, grid%voigt_a, grid%Dfreq, &
            grid%vfx, grid%vfy, grid%vfz, c_loc(grid%rhokapD)), 'lart_gpu_set_grid_car')
    else
       call lart_check(lart_gpu_set_grid_car(ctx, grid%nx, grid%ny, grid%nz, grid%xmin, grid%ymin, grid%zmin, grid%dx, grid%dy, grid%dz, &
            grid%i0, grid%j0, grid%k0, grid%xface, grid%yface, grid%zface, grid%rhokap, grid%voigt_a, grid%Dfreq, &
            grid%vfx, grid%vfy, grid%vfz, c_null_ptr), 'lart_gpu_set_grid_car')
    endif
    if (car_nbins > 0) call lart_check(lart_gpu_set_JPa(ctx, merge(1, 0, par%calc_J), merge(1, 0, par%calc_P), merge(1, 0, par%calc_Pnew), &
         car_nbins, car_ind_bin, line%cross0), 'lart_gpu_set_JPa')
    if (line%line_type >= 4 .and. line%line_type <= 6) call lart_check(lart_gpu_set_line_multiplet(ctx, line%ml), 'lart_gpu_set_line_multiplet')
    if (scatt_mat%nPDF > 0) call lart_check(lart_gpu_set_mueller(ctx, scatt_mat%nPDF, scatt_mat%coss, scatt_mat%S11, scatt_mat%S12, &
         scatt_mat%S33, scatt_mat%S34, scatt_mat%phase_PDF, scatt_mat%alias), 'lart_gpu_set_mueller')
    if (par%nobs > 0 .and. par%save_peeloff) then
       if (.not. allocated(xyz)) allocate(xyz(3*par%nobs), rmat(9*par%nobs))
       do i = 1, par%nobs
          xyz(3*i-2:3*i) = [observer(i)%x, observer(i)%y, observer(i)%z]
          rmat(9*i-8:9*i) = [observer(i)%rmatrix(1,:), observer(i)%rmatrix(2,:), observer(i)%rmatrix(3,:)]
       enddo
       if (par%nside > 0) then
          call lart_check(lart_gpu_set_observers_heal(ctx, par%nobs, xyz, par%nside), 'lart_gpu_set_observers_heal')
       else
          call lart_check(lart_gpu_set_observers(ctx, par%nobs, xyz, rmat, par%dxim, par%dyim, par%nxim, par%nyim), 'lart_gpu_set_observers')
       endif
    endif
  end subroutine setup_device

  subroutine time_stamp(dt)
    real(wp), intent(out) :: dt
    call system_clock(t1)
    dt = real(t1 - t0, wp)/real(rate, wp)
  end subroutine time_stamp

  !--- make_sightline_tau (sightline_tau_rect.f90:11-190, sightline_tau_heal.f90:11-165, sightline_tau_clump.f90:33-155) + the
  !    writers write_sightline_tau_outside (sightline_tau_rect.f90:347-407) / _inside (sightline_tau_heal.f90:292-345):
  !    one file <base>_tau[_kkk] per observer with sections TAU_gas, N_gas[, TAU_dust]
  subroutine make_sightline_tau(on_gpu)
    logical, intent(in) :: on_gpu
    real(wp), allocatable, target :: tau_gas(:,:,:,:), N_gas(:,:,:), tau_dust(:,:,:)
    integer :: nxi, nyi, io, fmt
    type(out_file) :: o
    type(key_list) :: kw, k0
    character(len=8) :: tail
    nxi = par%nxim; nyi = par%nyim
    if (par%nside > 0) then
       nxi = par%npix; nyi = 1
    endif
    allocate(tau_gas(par%nxfreq, nxi, nyi, par%nobs), N_gas(nxi, nyi, par%nobs), tau_dust(nxi, nyi, par%nobs))
    tau_gas = 0.0_wp; N_gas = 0.0_wp; tau_dust = 0.0_wp
    if (on_gpu) call lart_check(lart_gpu_sightline_tau(ctx, line%cross0, tau_gas, N_gas, tau_dust), 'lart_gpu_sightline_tau')
    fmt = out_format_of(par%file_format)
    do io = 1, par%nobs
       tail = ''
       if (par%nobs > 1) write(tail,'(a,i3.3)') '_', io
       kw%n = 0; k0%n = 0
       call kadd(kw, 'EQUINOX', 2000, 'Equinox of Ref. Coord.')
       if (par%nside > 0) then
          call kadd(kw, 'NSIDE', par%nside, 'HEALPIX nside')
          call kadd(kw, 'NPIX',  par%npix,  'HEALPIX npix')
          call kadd(kw, 'OBSX',  observer(io)%x, 'observer x')
          call kadd(kw, 'OBSY',  observer(io)%y, 'observer y')
          call kadd(kw, 'OBSZ',  observer(io)%z, 'observer z')
       else
          call kadd(kw, 'CD1_1',  par%dxim, 'Degree / Pixel')
          call kadd(kw, 'CD2_1',  0.0_wp,   'Degree / Pixel')
          call kadd(kw, 'CD1_2',  0.0_wp,   'Degree / Pixel')
          call kadd(kw, 'CD2_2',  par%dyim, 'Degree / Pixel')
          call kadd(kw, 'CRPIX1', (par%nxim + 1)/2.0_wp, 'Reference Pixel in X')
          call kadd(kw, 'CRPIX2', (par%nyim + 1)/2.0_wp, 'Reference Pixel in Y')
          call kadd(kw, 'CRVAL1', 0.0_wp, 'R.A. (Degree)')
          call kadd(kw, 'CRVAL2', 0.0_wp, 'Dec  (Degree)')
          call kadd(kw, 'CTYPE1', 'RA--TAN', 'Coordinate Type')
          call kadd(kw, 'CTYPE2', 'DEC-TAN', 'Coordinate Type')
          call kadd(kw, 'DISTANCE', par%distance, 'Distance')
       endif
       call kadd(kw, 'DISTUNIT', trim(par%distance_unit), 'Distance Unit')
       call out_open(o, trim(par%base_name)//'_tau'//trim(tail), fmt)
       if (par%nside > 0) then
          call out_image(o, 'TAU_gas', [par%nxfreq, nxi], tau_gas(:,:,1,io), kw)
          call out_image(o, 'N_gas', [nxi], N_gas(:,1,io), k0)
          if (par%DGR > 0.0_wp) call out_image(o, 'TAU_dust', [nxi], tau_dust(:,1,io), k0)
       else
          call out_image(o, 'TAU_gas', [par%nxfreq, nxi, nyi], tau_gas(:,:,:,io), kw)
          call out_image(o, 'N_gas', [nxi, nyi], N_gas(:,:,io), k0)
          if (par%DGR > 0.0_wp) call out_image(o, 'TAU_dust', [nxi, nyi], tau_dust(:,:,io), k0)
       endif
       call out_close(o, .true.)
       write(6,'(2a)') 'sight-line maps written: ', trim(o%path)
    enddo
  end subroutine make_sightline_tau

  subroutine null_tallies(tt)
    type(lart_tallies), intent(out) :: tt
    tt%Jout = c_null_ptr; tt%Jin = c_null_ptr; tt%Jabs = c_null_ptr; tt%Jmu = c_null_ptr
    tt%scatt = c_null_ptr; tt%direc = c_null_ptr; tt%I = c_null_ptr; tt%Q = c_null_ptr; tt%U = c_null_ptr; tt%V = c_null_ptr
    tt%scatt_2D = c_null_ptr; tt%direc_2D = c_null_ptr; tt%I_2D = c_null_ptr; tt%Q_2D = c_null_ptr; tt%U_2D = c_null_ptr; tt%V_2D = c_null_ptr
    tt%direc0 = c_null_ptr; tt%direc0_2D = c_null_ptr; tt%scatt_dust = c_null_ptr
    tt%J1 = c_null_ptr; tt%P1 = c_null_ptr; tt%P1_new = c_null_ptr
  end subroutine null_tallies

  subroutine fill_params(p)
    type(lart_params), intent(out) :: p
    p%xyz_symmetry = merge(1, 0, par%xyz_symmetry); p%xy_periodic = merge(1, 0, par%xy_periodic); p%xy_symmetry = merge(1, 0, par%xy_symmetry)
    select case (trim(par%source_geometry))
    case ('uniform_sphere');   p%source_geometry = 1
    case ('uniform_cylinder'); p%source_geometry = 2
    case ('uniform');          p%source_geometry = 3
    case ('uniform_xy');       p%source_geometry = 4
    case ('gaussian');         p%source_geometry = 5
    case default;              p%source_geometry = 0
    end select
    p%xrange = grid%xrange; p%yrange = grid%yrange; p%zrange = grid%zrange; p%source_zscale = par%source_zscale
    select case (trim(par%spectral_type))
    case ('voigt');     p%spectral_type = 1
    case ('continuum'); p%spectral_type = 2
    case ('gaussian');  p%spectral_type = 3
    case default;       p%spectral_type = 0
    end select
    p%xfreq_max = grid%xfreq_max
    if (par%gaussian_FWHM_vel > 0.0_wp) then
       p%source_sigma_x = par%gaussian_FWHM_vel/2.3548200450309493_wp/vtherm_total(par%temperature)
    else
       p%source_sigma_x = par%gaussian_sigma_vel/vtherm_total(par%temperature)
    endif
    p%comoving_source = merge(1, 0, par%comoving_source)
    p%recoil = merge(1, 0, par%recoil); p%core_skip = merge(1, 0, par%core_skip); p%core_skip_global = merge(1, 0, par%core_skip_global)
    p%use_stokes = merge(1, 0, par%use_stokes); p%use_reduced_wgt = merge(1, 0, par%use_reduced_wgt)
    p%save_Jin = merge(1, 0, par%save_Jin); p%save_Jabs = merge(1, 0, par%save_Jabs); p%save_Jmu = merge(1, 0, par%save_Jmu); p%nmu = par%nmu
    p%save_peeloff = merge(1, 0, par%save_peeloff); p%save_peeloff_2D = merge(1, 0, par%save_peeloff_2D); p%save_peeloff_3D = merge(1, 0, par%save_peeloff_3D)
    p%save_direc0 = merge(1, 0, par%save_direc0); p%save_dust_scattered = merge(1, 0, par%save_dust_scattered)
    p%nxfreq = grid%nxfreq
    p%xs_point = par%xs_point; p%ys_point = par%ys_point; p%zs_point = par%zs_point; p%source_rmax = par%source_rmax; p%xfreq0 = par%xfreq0
    p%xfreq_min = grid%xfreq_min; p%dxfreq = grid%dxfreq; p%Dfreq_ref = grid%Dfreq_ref
    p%DGR = par%DGR; p%hgg = par%hgg; p%albedo = par%albedo; p%mu_min = par%mu_min; p%dmu = par%dmu
    p%g_recoil0 = line%g_recoil0; p%E1 = line%E1; p%E2 = line%E2; p%E3 = line%E3
    p%line_type = line%line_type; p%DnuHK_Hz = line%DnuHK_Hz
    p%delta_nu_HD_Hz = line%delta_nu_HD_Hz; p%ratio_Dfreq_HD = line%ratio_Dfreq_HD; p%ratio_voigta_HD = line%ratio_voigta_HD
    p%nD_over_nH = line%nD_over_nH; p%g_recoil0_D = line%g_recoil0_D
    p%xcrit = grid%xcrit; p%xcrit2 = grid%xcrit2
    p%Omega = par%Omega
    if (line%line_type == 8) then                                       ! ly_beta: the band-2 (H-alpha) axis, the dust of band 2, the conversion channel
       p%nxfreq_Ha = grid%nxfreq_Ha; p%xfreq_min_Ha = grid%xfreq_min_Ha; p%dxfreq_Ha = grid%dxfreq_Ha
       p%R_Ha = par%R_Ha; p%albedo_Ha = par%albedo_Ha; p%hgg_Ha = par%hgg_Ha; p%P_conv = line%P_conv; p%E_conv = line%E_conv
    endif
    if (line%h2%n > 0) then                                             ! the H2 lines next to Ly-alpha (h2l(:) of h2_mod.f90)
       p%h2_nlines = line%h2%n; p%h2_hi_width = merge(1, 0, par%h2_hi_width); p%h2_Dfreq_Hz = line%h2%Dfreq_Hz
       p%h2_dnu_Hz = line%h2%dnu_Hz; p%h2_strength = line%h2%strength; p%h2_a_damp = line%h2%a_damp; p%h2_p_scat = line%h2%p_scat
    endif
  end subroutine fill_params

  !--- write_output_outside: `Spectrum` table + keywords (write_output_rect.f90:423-498), Jmu image (:524-539), profile
  !--- sections (:541-650), peel-off sections Scattered/Direct/Stokes_* (:1113-1344) in <base>_obs.* / <base>_stokes.*;
  !--- container by par%file_format: HDF5 <base>.h5 (reference default) or gzip FITS <base>.fits.gz
  !--- write_output_allph (write_output_rect.f90:1353-1515): <base>_allph, one row per photon
  !--- write_clumps_info (clump_mod.f90:1779-1961): <base>_clumps.{h5,fits.gz}; one table section of float32 columns X Y Z [code units],
  !--- VX VY VZ [km/s] and, only where they vary by more than 1e-3 of their mean, R_CLUMP RHOKAP TEMP; the scalars as keywords of the section
  subroutine write_clumps_info()
    type(out_file) :: o
    type(key_list) :: kl
    character(len=16) :: names(9)
    real(wp), allocatable :: cols(:,:)
    real(wp) :: f_vol, f_cov, shell2, vshell
    integer :: n, ncol, keep_bitpix
    n = int(N_clumps); allocate(cols(n, 9))
    names(1:6) = [character(len=16) :: 'X', 'Y', 'Z', 'VX', 'VY', 'VZ']; ncol = 6
    cols(:,1) = cl_x; cols(:,2) = cl_y; cols(:,3) = cl_z
    cols(:,4) = cl_vx*cl_vtherm; cols(:,5) = cl_vy*cl_vtherm; cols(:,6) = cl_vz*cl_vtherm
    if (varies(cl_radius)) then
       ncol = ncol + 1; names(ncol) = 'R_CLUMP'; cols(:,ncol) = cl_radius
    endif
    if (varies(cl_rhokap)) then
       ncol = ncol + 1; names(ncol) = 'RHOKAP'; cols(:,ncol) = cl_rhokap
    endif
    if (varies(cl_temperature)) then
       ncol = ncol + 1; names(ncol) = 'TEMP'; cols(:,ncol) = cl_temperature
    endif
    vshell = max(sphere_R**3 - r_min_clump**3, tiny(1.0_wp)); shell2 = max(sphere_R**2 + sphere_R*r_min_clump + r_min_clump**2, tiny(1.0_wp))
    if (clumps_from_file) then
       f_vol = sum(cl_radius**3)/vshell; f_cov = 0.75_wp*sum(cl_radius**2)/shell2
    else
       f_vol = real(N_clumps, wp)*cl_radius_max**3/vshell; f_cov = 0.75_wp*real(N_clumps, wp)*cl_radius_max**2/shell2
    endif
    write(*,'(a,es12.4)') ' Clumps: realized f_vol      = ', f_vol
    write(*,'(a,es12.4)') ' Clumps: realized f_cov      = ', f_cov
    write(*,'(a,3l2)') ' Clumps: clump columns saved (R_CLUMP/RHOKAP/TEMP) = ', varies(cl_radius), varies(cl_rhokap), varies(cl_temperature)
    kl%n = 0
    call kadd(kl, 'N_CLUMPS', n,                    'number of clumps (realized)')
    call kadd(kl, 'SPHERE_R', sphere_R,             'outer sphere radius [code units]')
    call kadd(kl, 'RMIN',     r_min_clump,          'inner placement radius [code units]')
    call kadd(kl, 'CL_RAD',   cl_radius_max,        'clump radius (max) [code units]')
    call kadd(kl, 'F_VOL',    f_vol,                'volume filling factor (realized)')
    call kadd(kl, 'F_COV',    f_cov,                'covering factor (realized)')
    call kadd(kl, 'TAU0',     par%clump_tau0,       'line-center tau (center to surface)')
    call kadd(kl, 'SIGMA_V',  par%clump_sigma_v,    'bulk velocity sigma [km/s]')
    call kadd(kl, 'TEMP_CL',  cl_temperature_ref,   'clump temperature [K] (reference)')
    call kadd(kl, 'RHOKAP',   cl_rhokap_ref,        'opacity/code-unit inside clumps (ref)')
    call kadd(kl, 'CL_DFREQ', cl_Dfreq_ref,         'Doppler frequency [Hz] (reference)')
    call kadd(kl, 'VTHERM',   cl_vtherm_ref,        'thermal velocity [km/s] (reference)')
    call kadd(kl, 'VOIGT_A',  cl_voigt_a_ref,       'Voigt damping parameter (reference)')
    call kadd(kl, 'RMAX',     par%rmax,             'outer sphere radius input (par%rmax)')
    call kadd(kl, 'IN_FCOV',  par%clump_f_cov,      'covering factor (input)')
    call kadd(kl, 'IN_FVOL',  par%clump_f_vol,      'volume filling factor (input)')
    call kadd(kl, 'IN_NCL',   par%clump_N_clumps,   'N_clumps (input)')
    call kadd(kl, 'IN_NHI',   par%clump_NHI,        'clump NHI input [cm^-2] (clump center to surface)')
    call kadd(kl, 'IN_NH',    par%clump_nH,         'clump nH density input [cm^-3]')
    call kadd(kl, 'IN_TEMP',  par%clump_temperature,'clump temperature input [K]')
    call kadd(kl, 'DISTUNIT', trim(par%distance_unit), 'Distance Unit')
    call kadd(kl, 'DIST_CM',  par%distance2cm,      'Distance Unit (cm)')
    keep_bitpix = out_bitpix; out_bitpix = -32              ! float32 columns (:1805)
    call out_open(o, trim(par%base_name)//'_clumps', out_format_of(trim(strlowcase(par%file_format))))
    call out_table(o, 'section_001', names(1:ncol), cols(:, 1:ncol), kl)
    call out_close(o, .true.)
    out_bitpix = keep_bitpix
    write(*,'(2a)') ' Clumps saved to ', trim(o%path)
  contains
    logical function varies(a)
      real(wp), intent(in) :: a(:)
      varies = (maxval(a) - minval(a)) > 1.0e-3_wp*max(abs(sum(a)/max(real(size(a), wp), 1.0_wp)), tiny(1.0_wp))
    end function varies
  end subroutine write_clumps_info

  subroutine write_output_allph()
    real(wp), allocatable :: rec(:,:)
    character(len=16) :: names(10)
    integer :: cols(10), ncol, fmt, j
    type(out_file) :: o
    type(key_list) :: k0
    allocate(rec(par%nphotons, 10))
    call lart_check(lart_gpu_get_all_photons(ctx, rec, int(par%nphotons, c_int64_t)), 'lart_gpu_get_all_photons')
    names(1:5) = [character(len=16) :: 'r', 'xfreq1', 'xfreq2', 'NSCATT_gas', 'NSCATT_dust']; cols(1:5) = [1, 2, 3, 4, 5]; ncol = 5
    if (par%use_stokes) then
       names(ncol+1:ncol+4) = [character(len=16) :: 'I', 'Q', 'U', 'V']; cols(ncol+1:ncol+4) = [6, 7, 8, 9]; ncol = ncol + 4
    endif
    if (trim(par%source_geometry) /= 'point') then
       ncol = ncol + 1; names(ncol) = 'r0'; cols(ncol) = 10
    endif
    fmt = out_format_of(par%file_format)
    call out_open(o, trim(par%base_name)//'_allph', fmt)
    call out_table(o, 'AllPhotons', names(1:ncol), rec(:, cols(1:ncol)), k0)
    call out_close(o, .true.)
    write(*,'(2a)') 'all photons  -> ', trim(o%path)
  end subroutine write_output_allph

  subroutine write_output(filename, g)
    character(len=*), intent(in) :: filename
    type(grid_type), intent(inout) :: g
    type(out_file) :: o
    real(wp) :: nph_main, exetime_tot
    type(key_list) :: kl, ki, ko, k0, kh
    character(len=16) :: names(6)
    character(len=8) :: tail
    real(wp), allocatable :: cols(:,:), rad(:,:)
    integer :: ncol, fmt, io, nxi, nyi
    k0%n = 0
    fmt = out_format_of(trim(strlowcase(par%file_format)))
    nph_main = real(par%nphotons, wp); exetime_tot = par%exetime; nph_key = nph_main
    if (par%out_merge) call merge_main_file()          ! write_output_rect.f90:75-78, 112-420
    ncol = 4; names(1:4) = [character(len=16) :: 'Xfreq', 'velocity', 'wavelength', 'Jout']
    if (par%save_Jabs) then
       ncol = ncol + 1; names(ncol) = 'Jabs'
    endif
    if (par%save_Jin) then
       ncol = ncol + 1; names(ncol) = 'Jin'
    endif
    allocate(cols(g%nxfreq, ncol))
    cols(:,1) = g%xfreq; cols(:,2) = g%velocity; cols(:,3) = g%wavelength; cols(:,4) = g%Jout
    ncol = 4
    if (par%save_Jabs) then
       ncol = ncol + 1; cols(:,ncol) = g%Jabs
    endif
    if (par%save_Jin) then
       ncol = ncol + 1; cols(:,ncol) = g%Jin
    endif
    call kadd(kl, 'ExeTime',  exetime_tot,        'Excution Time (min)')
    call kadd(kl, 'Nprocs',   max(ngpu, 1),       'No. of Threads')          ! here: GPUs that shared the photons
    call kadd(kl, 'recoil',   par%recoil,         'recoil')
    call kadd(kl, 'coreskip', par%core_skip,      'coreskip algorithm')
    call kadd(kl, 'xyz_sym',  par%xyz_symmetry,   'xyz_symmetry')
    call kadd(kl, 'xy_per',   par%xy_periodic,    'xy_periodic')
    call kadd(kl, 'save_all', par%save_all,       'save all 3D output?')
    call kadd(kl, 'save_Jin', par%save_Jin,       'save input J?')
    call kadd(kl, 'save_Jab', par%save_Jabs,      'save absorbed J?')
    call kadd(kl, 'nphotons', nph_main, 'number of photons')
    call kadd(kl, 'taumax',   par%taumax,         'tau_max')
    call kadd(kl, 'tauhomo',  par%tauhomo,        'tau_homo')
    call kadd(kl, 'Ngasmax',  par%N_gasmax,       'N(gas)_max cm^-2')
    call kadd(kl, 'Ngashomo', par%N_gashomo,      'N(gas)_homo cm^-2')
    call kadd(kl, 'temp',     par%temperature,    'temperature (K)')
    call kadd(kl, 'Vexp',     par%Vexp,           'Velocity (km/s)')
    call kadd(kl, 'DGR',      par%DGR,            'Dust-to-Gas Ratio')
    call kadd(kl, 'atau3',    par%atau3,          '(a*tau_max)^(1/3)')
    call kadd(kl, 'voigta',   g%voigt_amean,      'voigt_a')
    call kadd(kl, 'Xfreq1',   g%xfreq_min,        'Xfreq_min')
    call kadd(kl, 'Xfreq2',   g%xfreq_max,        'Xfreq_max')
    call kadd(kl, 'Dxfreq',   g%dxfreq,           'Dxfreq')
    call kadd(kl, 'Dwave',    g%dwave,            'Dwavelength (angstrom)')
    call kadd(kl, 'I_unit',   par%intensity_unit, 'Intensity Unit (0:no dimension, 1:cm^-2 A^-1)')
    call kadd(kl, 'Dfreq',    g%Dfreq_ref,        'Doppler Freq. (Hz)')
    call kadd(kl, 'Nsc_dust', par%nscatt_dust,    'Nscatt_dust/photon')
    call kadd(kl, 'Nsc_gas',  par%nscatt_gas,     'Nscatt_gas/photon')
    call kadd(kl, 'Nsc_tot',  par%nscatt_tot,     'Nscatt_tot/photon')
    call kadd(kl, 'nx',       g%nx,               'No. of x cells')
    call kadd(kl, 'ny',       g%ny,               'No. of y cells')
    call kadd(kl, 'nz',       g%nz,               'No. of z cells')
    call kadd(kl, 'xmax',     par%xmax,           'xmax')
    call kadd(kl, 'ymax',     par%ymax,           'ymax')
    call kadd(kl, 'zmax',     par%zmax,           'zmax')
    call kadd(kl, 'calc_P',   par%calc_P,         'calculated Pa within media?')
    call kadd(kl, 'calc_Pnew', par%calc_Pnew,     'calculated Pa within media?')
    call kadd(kl, 'calc_J',   par%calc_J,         'calculated J within media?')
    call kadd(kl, 'ISEED',    real(par%run_seed, wp), 'random seed of this run (host addition; iseed = 0: drawn from the OS)')
    call kadd(ki, 'nphotons', nph_main, 'number of photons')
    call kadd(ki, 'dxim', par%dxim, 'pixel size along x (deg)')
    call kadd(ki, 'dyim', par%dyim, 'pixel size along y (deg)')
    call kadd(ki, 'distance', par%distance, 'observer distance')

    call out_open(o, trim(par%base_name), fmt)
    call out_table(o, 'Spectrum', names(1:ncol), cols, kl)
    if (par%save_Jmu) call out_image(o, 'Jmu', [g%nxfreq, par%nmu], g%Jmu, ki)
    !--- volume-normalised profiles on the octree (write_output_rect.f90:541-650, sections Jx_1D / Pa_1D / Pa_1D_new)
    if (allocated(prof_J1))     call out_image(o, 'Jx_1D', shape(prof_J1), prof_J1, ki)
    if (allocated(prof_P1))     call out_image(o, 'Pa_1D', shape(prof_P1), prof_P1, ki)
    if (allocated(prof_P1_new)) call out_image(o, 'Pa_1D_new', shape(prof_P1_new), prof_P1_new, ki)
    if (line%line_type == 8) then                                       ! write_output_rect.f90:652-733: the band-2 sections, behind every older one
       kh%n = 0
       call kadd(kh, 'nxfreq_Ha', g%nxfreq_Ha,    'No. of band-2 frequency bins')
       call kadd(kh, 'Xfreq1_Ha', g%xfreq_min_Ha, 'band-2 Xfreq_min')
       call kadd(kh, 'Xfreq2_Ha', g%xfreq_max_Ha, 'band-2 Xfreq_max')
       call kadd(kh, 'Dxfreq_Ha', g%dxfreq_Ha,    'band-2 Dxfreq')
       call kadd(kh, 'Dwave_Ha',  g%dwave_Ha,     'band-2 Dwavelength (angstrom)')
       call kadd(kh, 'lambda0_Ha', line%wavelength0_Ha*1e4_wp, 'H-alpha rest wavelength (angstrom)')
       call kadd(kh, 'W_esc1', par%W_esc1/nph_main, 'band-1 escaped weight/photon (this run)')
       call kadd(kh, 'W_abs1', par%W_abs1/nph_main, 'band-1 dust-absorbed weight/photon (this run)')
       call kadd(kh, 'W_conv', par%W_conv/nph_main, 'conversion weight/photon (this run)')
       call kadd(kh, 'W_esc2', par%W_esc2/nph_main, 'band-2 escaped weight/photon (this run)')
       call kadd(kh, 'W_abs2', par%W_abs2/nph_main, 'band-2 dust-absorbed weight/photon (this run)')
       if (allocated(g%Jout_Ha)) call out_image(o, 'Jout_Ha', [g%nxfreq_Ha], g%Jout_Ha, kh)
       if (allocated(g%Jabs_Ha)) call out_image(o, 'Jabs_Ha', [g%nxfreq_Ha], g%Jabs_Ha, kh)
       if (allocated(prof_Pconv)) call out_image(o, 'Pconv_1D', shape(prof_Pconv), prof_Pconv, ki)
    endif
    call out_close(o, .true.)
    write(6,'(2a)') 'output written: ', trim(o%path)
    !--- peel-off images: one set of files per observer (write_output_rect.f90:742-1350), <base>_obs[_kkk] / <base>_stokes[_kkk]
    !    for the (nxfreq,nxim,nyim) cubes, <base>_obs2D[_kkk] / <base>_stokes_2D[_kkk] for the frequency-integrated images
    if (par%save_peeloff .and. par%nobs > 0) then
       do io = 1, par%nobs
          tail = ''
          if (par%nobs > 1) write(tail,'(a,i3.3)') '_', io
          nph_key = real(par%nphotons, wp)
          if (par%out_merge .and. allocated(obs_scatt)) call merge_peel_files(io, trim(tail), .false.)     ! write_output_rect.f90:1031-1340
          call peel_keys(ko, io, g, merge(3, 1, par%nside > 0))
          if (allocated(obs_scatt)) then
             nxi = size(obs_scatt, 2); nyi = size(obs_scatt, 3)
             call out_open(o, trim(par%base_name)//'_obs'//trim(tail), fmt)
             if (par%nside > 0) then                     ! write_output_heal.f90: (nxfreq, npix) maps
                call out_image(o, 'Scattered', [g%nxfreq, nxi], obs_scatt(:,:,1,io), ko)
                call out_image(o, 'Direct',    [g%nxfreq, nxi], obs_direc(:,:,1,io), k0)
             else
                call out_image(o, 'Scattered', [g%nxfreq, nxi, nyi], obs_scatt(:,:,:,io), ko)
                call out_image(o, 'Direct',    [g%nxfreq, nxi, nyi], obs_direc(:,:,:,io), k0)
                if (allocated(obs_direc0))     call out_image(o, 'Direct0',  [g%nxfreq, nxi, nyi], obs_direc0(:,:,:,io), k0)
                if (allocated(obs_scatt_dust)) call out_image(o, 'ScatDust', [g%nxfreq, nxi, nyi], obs_scatt_dust(:,:,:,io), k0)
                if (allocated(obs_peel_Ha))    call out_image(o, 'Peel_Ha',  [g%nxfreq_Ha, nxi, nyi], obs_peel_Ha(:,:,:,io), k0)      ! observer(:)%peel_Ha (ly_beta)
             endif
             if (par%save_radial_profile .and. par%nside == 0) then
                call radial_profiles(g, io, .false., rad)
                call out_table(o, 'RadialI', [character(len=16) :: 'r', 'I'], rad(:,1:2), k0)
             endif
             call out_close(o, .true.)
             if (par%use_stokes .and. allocated(obs_I)) then
                call out_open(o, trim(par%base_name)//'_stokes'//trim(tail), fmt)
                call out_image(o, 'Stokes_I', [g%nxfreq, nxi, nyi], obs_I(:,:,:,io), ko)
                call out_image(o, 'Stokes_Q', [g%nxfreq, nxi, nyi], obs_Q(:,:,:,io), k0)
                call out_image(o, 'Stokes_U', [g%nxfreq, nxi, nyi], obs_U(:,:,:,io), k0)
                call out_image(o, 'Stokes_V', [g%nxfreq, nxi, nyi], obs_V(:,:,:,io), k0)
                if (par%save_radial_profile .and. par%nside == 0) then
                   call radial_profiles(g, io, .false., rad, stokes=.true.)
                   call out_table(o, 'Stokes_radial', [character(len=16) :: 'r', 'I', 'Q', 'U', 'V', 'pol'], rad, k0)
                endif
                call out_close(o, .true.)
             endif
          endif
          if (allocated(obs_scatt_2D)) then
             nph_key = real(par%nphotons, wp)
             if (par%out_merge) call merge_peel_files(io, trim(tail), .true.)                                 ! write_output_rect.f90:770-1000
             call peel_keys(ko, io, g, merge(4, 2, par%nside > 0))
             nxi = size(obs_scatt_2D, 1); nyi = size(obs_scatt_2D, 2)
             call out_open(o, trim(par%base_name)//'_obs2D'//trim(tail), fmt)
             if (par%nside > 0) then
                call out_image(o, 'Scattered', [nxi], obs_scatt_2D(:,1,io), ko)
                call out_image(o, 'Direct',    [nxi], obs_direc_2D(:,1,io), k0)
             else
                call out_image(o, 'Scattered', [nxi, nyi], obs_scatt_2D(:,:,io), ko)
                call out_image(o, 'Direct',    [nxi, nyi], obs_direc_2D(:,:,io), k0)
                if (allocated(obs_direc0_2D)) call out_image(o, 'Direct0', [nxi, nyi], obs_direc0_2D(:,:,io), k0)
             endif
             if (par%save_radial_profile .and. par%nside == 0) then
                call radial_profiles(g, io, .true., rad)
                call out_table(o, 'RadialI', [character(len=16) :: 'r', 'I'], rad(:,1:2), k0)
             endif
             call out_close(o, .true.)
             if (par%use_stokes .and. allocated(obs_I_2D)) then
                call out_open(o, trim(par%base_name)//'_stokes_2D'//trim(tail), fmt)
                call out_image(o, 'Stokes_I', [nxi, nyi], obs_I_2D(:,:,io), ko)
                call out_image(o, 'Stokes_Q', [nxi, nyi], obs_Q_2D(:,:,io), k0)
                call out_image(o, 'Stokes_U', [nxi, nyi], obs_U_2D(:,:,io), k0)
                call out_image(o, 'Stokes_V', [nxi, nyi], obs_V_2D(:,:,io), k0)
                if (par%save_radial_profile .and. par%nside == 0) then
                   call radial_profiles(g, io, .true., rad, stokes=.true.)
                   call out_table(o, 'Stokes_radial', [character(len=16) :: 'r', 'I', 'Q', 'U', 'V', 'pol'], rad, k0)
                endif
                call out_close(o, .true.)
             endif
          endif
       enddo
    endif
  contains
    !--- par%out_merge: a previous output of the same name is read and combined, photon-weighted, with this run before the file is written
    !--- again: a = (a_old nph_old + a_new nph_new) / (nph_old + nph_new) for every array, the same for the scattering counts, the execution
    !--- times added (write_output_rect.f90:112-420).  Here for both containers (the reference's HDF5 backend cannot read-and-rewrite).
    function cstr(x) result(c)
      character(len=*), intent(in) :: x
      character(kind=c_char, len=:), allocatable :: c
      c = trim(x)//c_null_char
    end function cstr
    function out_path(base) result(path)
      character(len=*), intent(in) :: base
      character(len=512) :: path
      path = trim(base)//'.fits.gz'
      if (fmt == OUT_HDF5) path = trim(base)//'.h5'
    end function out_path
    logical function old_nph(path, section, nph1)
      character(len=*), intent(in) :: path, section
      real(wp), intent(out) :: nph1
      logical :: ex
      old_nph = .false.; nph1 = 0.0_wp
      inquire(file=trim(path), exist=ex)
      if (.not. ex) return
      if (lart_out_read_key(cstr(path), cstr(section), cstr('nphotons'), nph1) /= 0 .or. nph1 <= 0.0_wp) then
         write(*,'(4a)') 'out_merge: no nphotons keyword in section ', section, ' of ', trim(path); return
      endif
      old_nph = .true.
    end function old_nph
    subroutine merge_arr(path, section, name, a, n, nph1)
      character(len=*), intent(in) :: path, section, name
      integer, intent(in) :: n
      real(wp), intent(inout) :: a(n)
      real(wp), intent(in) :: nph1
      real(wp), allocatable :: old(:)
      real(wp) :: nph2
      integer :: rc
      allocate(old(n)); nph2 = real(par%nphotons, wp)
      rc = lart_out_read(cstr(path), cstr(section), cstr(name), old, int(n, c_long_long))
      if (rc /= 0) then
         write(*,'(6a,i0,a)') 'out_merge: cannot read ', name, ' of section ', section, ' in ', trim(path), rc, ': kept this run only'; return
      endif
      a = (old*nph1 + a*nph2)/(nph1 + nph2)
    end subroutine merge_arr
    subroutine merge_main_file()
      character(len=512) :: path
      real(wp) :: nph1, nph2, v
      path = out_path(trim(par%base_name))
      if (.not. old_nph(path, 'Spectrum', nph1)) return
      nph2 = real(par%nphotons, wp); nph_main = nph1 + nph2
      if (lart_out_read_key(cstr(path), cstr('Spectrum'), cstr('ExeTime'), v) == 0) exetime_tot = par%exetime + v
      if (lart_out_read_key(cstr(path), cstr('Spectrum'), cstr('Nsc_gas'), v) == 0)  par%nscatt_gas  = (v*nph1 + par%nscatt_gas*nph2)/nph_main
      if (lart_out_read_key(cstr(path), cstr('Spectrum'), cstr('Nsc_dust'), v) == 0) par%nscatt_dust = (v*nph1 + par%nscatt_dust*nph2)/nph_main
      if (lart_out_read_key(cstr(path), cstr('Spectrum'), cstr('Nsc_tot'), v) == 0)  par%nscatt_tot  = (v*nph1 + par%nscatt_tot*nph2)/nph_main
      call merge_arr(path, 'Spectrum', 'Jout', g%Jout, g%nxfreq, nph1)
      if (par%save_Jabs) call merge_arr(path, 'Spectrum', 'Jabs', g%Jabs, g%nxfreq, nph1)
      if (par%save_Jin)  call merge_arr(path, 'Spectrum', 'Jin',  g%Jin,  g%nxfreq, nph1)
      if (par%save_Jmu)  call merge_arr(path, 'Jmu', 'data', g%Jmu, size(g%Jmu), nph1)
      if (allocated(prof_J1))     call merge_arr(path, 'Jx_1D', 'data', prof_J1, size(prof_J1), nph1)
      if (allocated(prof_P1))     call merge_arr(path, 'Pa_1D', 'data', prof_P1, size(prof_P1), nph1)
      if (allocated(prof_P1_new)) call merge_arr(path, 'Pa_1D_new', 'data', prof_P1_new, size(prof_P1_new), nph1)
      write(*,'(3a,es12.4,a,es12.4)') 'out_merge: combined with ', trim(path), ': photons ', nph1, ' + ', nph2
    end subroutine merge_main_file
    subroutine merge_peel_files(io, tail, two_d)
      integer, intent(in) :: io
      character(len=*), intent(in) :: tail
      logical, intent(in) :: two_d
      character(len=512) :: path
      real(wp) :: nph1
      integer :: n
      if (two_d) then
         path = out_path(trim(par%base_name)//'_obs2D'//tail); n = size(obs_scatt_2D, 1)*size(obs_scatt_2D, 2)
         if (old_nph(path, 'Scattered', nph1)) then
            call merge_arr(path, 'Scattered', 'data', obs_scatt_2D(:,:,io), n, nph1); call merge_arr(path, 'Direct', 'data', obs_direc_2D(:,:,io), n, nph1)
            if (allocated(obs_direc0_2D)) call merge_arr(path, 'Direct0', 'data', obs_direc0_2D(:,:,io), n, nph1)
            nph_key = nph1 + real(par%nphotons, wp)
         endif
         if (par%use_stokes .and. allocated(obs_I_2D)) then
            path = out_path(trim(par%base_name)//'_stokes_2D'//tail)
            if (old_nph(path, 'Stokes_I', nph1)) then
               call merge_arr(path, 'Stokes_I', 'data', obs_I_2D(:,:,io), n, nph1); call merge_arr(path, 'Stokes_Q', 'data', obs_Q_2D(:,:,io), n, nph1)
               call merge_arr(path, 'Stokes_U', 'data', obs_U_2D(:,:,io), n, nph1); call merge_arr(path, 'Stokes_V', 'data', obs_V_2D(:,:,io), n, nph1)
            endif
         endif
      else
         path = out_path(trim(par%base_name)//'_obs'//tail); n = size(obs_scatt, 1)*size(obs_scatt, 2)*size(obs_scatt, 3)
         if (old_nph(path, 'Scattered', nph1)) then
            call merge_arr(path, 'Scattered', 'data', obs_scatt(:,:,:,io), n, nph1); call merge_arr(path, 'Direct', 'data', obs_direc(:,:,:,io), n, nph1)
            if (allocated(obs_direc0))     call merge_arr(path, 'Direct0',  'data', obs_direc0(:,:,:,io), n, nph1)
            if (allocated(obs_scatt_dust)) call merge_arr(path, 'ScatDust', 'data', obs_scatt_dust(:,:,:,io), n, nph1)
            nph_key = nph1 + real(par%nphotons, wp)
         endif
         if (par%use_stokes .and. allocated(obs_I)) then
            path = out_path(trim(par%base_name)//'_stokes'//tail)
            if (old_nph(path, 'Stokes_I', nph1)) then
               call merge_arr(path, 'Stokes_I', 'data', obs_I(:,:,:,io), n, nph1); call merge_arr(path, 'Stokes_Q', 'data', obs_Q(:,:,:,io), n, nph1)
               call merge_arr(path, 'Stokes_U', 'data', obs_U(:,:,:,io), n, nph1); call merge_arr(path, 'Stokes_V', 'data', obs_V(:,:,:,io), n, nph1)
            endif
         endif
      endif
    end subroutine merge_peel_files
  end subroutine write_output

  !--- keywords of the first section of a peel-off file, in the reference's order:
  !    mode 1 = (nxfreq,nxim,nyim) cube  write_output_rect.f90:1093-1150     mode 2 = (nxim,nyim) image  write_output_rect.f90:829-853
  !    mode 3 = (nxfreq,npix) HEALPix    write_output_heal.f90:757-771       mode 4 = (npix) HEALPix     write_output_heal.f90:592-602
  subroutine peel_keys(k, io, g, mode)
    type(key_list), intent(out) :: k
    integer, intent(in) :: io, mode
    type(grid_type), intent(in) :: g
    k%n = 0
    call kadd(k, 'EQUINOX', 2000, 'Equinox of Ref. Coord.')
    if (mode == 1) then
       call kadd(k, 'CTYPE1', 'WAVE', 'vacuum wavelength (FITS WCS)')
       call kadd(k, 'CUNIT1', 'Angstrom', 'wavelength unit')
       call kadd(k, 'CRPIX1', 1.0_wp, 'reference pixel for spectral axis')
       call kadd(k, 'CRVAL1', g%wavelength(1), 'wavelength at CRPIX1 (Angstrom)')
       call kadd(k, 'CD1_1', -g%dwave, 'wavelength step (Angstrom/pixel)')
       call kadd(k, 'CTYPE2', 'RA--TAN', 'Coordinate Type')
       call kadd(k, 'CUNIT2', 'deg', 'RA unit')
       call kadd(k, 'CRPIX2', (par%nxim + 1)/2.0_wp, 'Reference Pixel in X')
       call kadd(k, 'CRVAL2', 0.0_wp, 'R.A. (Degree)')
       call kadd(k, 'CD2_2', par%dxim, 'Degree / Pixel')
       call kadd(k, 'CTYPE3', 'DEC-TAN', 'Coordinate Type')
       call kadd(k, 'CUNIT3', 'deg', 'DEC unit')
       call kadd(k, 'CRPIX3', (par%nyim + 1)/2.0_wp, 'Reference Pixel in Y')
       call kadd(k, 'CRVAL3', 0.0_wp, 'Dec  (Degree)')
       call kadd(k, 'CD3_3', par%dyim, 'Degree / Pixel')
    else if (mode == 2) then
       call kadd(k, 'CD1_1',  par%dxim, 'Degree / Pixel')
       call kadd(k, 'CD2_1',  0.0_wp,   'Degree / Pixel')
       call kadd(k, 'CD1_2',  0.0_wp,   'Degree / Pixel')
       call kadd(k, 'CD2_2',  par%dyim, 'Degree / Pixel')
       call kadd(k, 'CRPIX1', (par%nxim + 1)/2.0_wp, 'Reference Pixel in X')
       call kadd(k, 'CRPIX2', (par%nyim + 1)/2.0_wp, 'Reference Pixel in Y')
       call kadd(k, 'CRVAL1', 0.0_wp, 'R.A. (Degree)')
       call kadd(k, 'CRVAL2', 0.0_wp, 'Dec  (Degree)')
       call kadd(k, 'CTYPE1', 'RA--TAN', 'Coordinate Type')
       call kadd(k, 'CTYPE2', 'DEC-TAN', 'Coordinate Type')
    else if (mode == 4) then
       call kadd(k, 'NSIDE', par%nside, 'HEALPIX NSIDE')
       call kadd(k, 'NPIX',  par%npix,  'HEALPIX NPIX')
    endif
    if (mode <= 2) call kadd(k, 'DISTANCE', par%distance, 'Distance')
    call kadd(k, 'DISTUNIT', trim(par%distance_unit), 'Distance Unit')
    call kadd(k, 'DIST_CM', par%distance2cm, 'Distance Unit (cm)')
    if (mode == 1 .or. mode == 3) then
       call kadd(k, 'Xfreq1', g%xfreq_min, 'Xfreq_min')
       call kadd(k, 'Xfreq2', g%xfreq_max, 'Xfreq_max')
       call kadd(k, 'Dxfreq', g%dxfreq, 'Dxfreq')
       call kadd(k, 'Dwave', g%dwave, 'Dwavelength (angstrom)')
       call kadd(k, 'I_unit', par%intensity_unit, 'Intensity Unit (0:no dimension, 1:cm^-2 A^-1)')
       call kadd(k, 'Dfreq', g%Dfreq_ref, 'Doppler Freq.  (Hz)')
    endif
    call kadd(k, 'nphotons', nph_key, 'number of photons')
    if (mode <= 2) then
       call kadd(k, 'alpha', par%alpha(io), 'alpha (degree)')
       call kadd(k, 'beta',  par%beta(io),  'beta (degree)')
       call kadd(k, 'gamma', par%gamma(io), 'gamma (degree)')
    endif
    call kadd(k, 'obsx', observer(io)%x, 'Observer X coordinate')
    call kadd(k, 'obsy', observer(io)%y, 'Observer Y coordinate')
    call kadd(k, 'obsz', observer(io)%z, 'Observer Z coordinate')
    if (mode <= 2) then                                  ! observer_rect.f90:36-38: the rotation centre defaults to the origin
       call kadd(k, 'rot_cenx', 0.0_wp, 'Rotation Center X coordinate')
       call kadd(k, 'rot_ceny', 0.0_wp, 'Rotation Center Y coordinate')
       call kadd(k, 'rot_cenz', 0.0_wp, 'Rotation Center Z coordinate')
    endif
    if (mode == 3) then                                  ! (the all-sky maps also say how they are pixelised; the reference's 3-D file leaves it to the 2-D one)
       call kadd(k, 'NSIDE', par%nside, 'HEALPIX NSIDE')
       call kadd(k, 'NPIX',  par%npix,  'HEALPIX NPIX')
    endif
  end subroutine peel_keys

  !--- make_radial_intensity / make_radial_stokes (output_sum_rect.f90:489-658): azimuthal means of the (normalised) peel-off
  !    image about its centre; rad(:,1) = r, (:,2) = I, and with stokes (:,3:6) = Q, U, V, pol in the radial frame
  subroutine radial_profiles(g, io, use_2D, rad, stokes)
    type(grid_type), intent(in) :: g
    integer, intent(in) :: io
    logical, intent(in) :: use_2D
    real(wp), allocatable, intent(out) :: rad(:,:)
    logical, intent(in), optional :: stokes
    integer :: nr, i, j, ir, nxi, nyi
    integer, allocatable :: ncount(:)
    real(wp) :: xcen, ycen, xx, yy, rr, cosp, sinp, cos2p, sin2p, unit, vi, vq, vu, vv
    logical :: st
    st = .false.
    if (present(stokes)) st = stokes
    unit = g%dxfreq
    if (par%intensity_unit == 1) unit = g%dwave
    nxi = par%nxim; nyi = par%nyim
    nr = (max(nxi, nyi) + 1)/2
    allocate(rad(nr, 6), ncount(nr)); rad = 0.0_wp; ncount = 0
    if (mod(nr, 2) == 0) then
       rad(:,1) = [((i - 0.5_wp)/nr, i = 1, nr)]
    else
       rad(:,1) = [((i - 1.0_wp)/(nr - 0.5_wp), i = 1, nr)]
    endif
    xcen = (nxi + 1.0_wp)/2.0_wp; ycen = (nxi + 1.0_wp)/2.0_wp          ! (the reference uses nxim for both)
    do j = 1, nyi
       yy = j - ycen
       do i = 1, nxi
          xx = i - xcen
          rr = sqrt(xx**2 + yy**2)
          if (mod(nr, 2) == 0) then
             ir = floor(rr) + 1
          else
             ir = floor(rr + 0.5_wp) + 1
          endif
          if (ir < 1 .or. ir > nr) cycle
          ncount(ir) = ncount(ir) + 1
          if (.not. st) then
             if (use_2D) then
                rad(ir,2) = rad(ir,2) + obs_scatt_2D(i,j,io) + obs_direc_2D(i,j,io)
             else
                rad(ir,2) = rad(ir,2) + (sum(obs_scatt(:,i,j,io)) + sum(obs_direc(:,i,j,io)))*unit
             endif
             cycle
          endif
          if (rr == 0.0_wp) then
             cosp = 1.0_wp; sinp = 0.0_wp
          else
             cosp = yy/rr; sinp = -xx/rr
          endif
          cos2p = 2.0_wp*cosp**2 - 1.0_wp; sin2p = 2.0_wp*cosp*sinp
          if (use_2D) then
             vi = obs_I_2D(i,j,io); vq = obs_Q_2D(i,j,io)*cos2p + obs_U_2D(i,j,io)*sin2p
             vu = -obs_Q_2D(i,j,io)*sin2p + obs_U_2D(i,j,io)*cos2p; vv = obs_V_2D(i,j,io)
          else
             vi = sum(obs_I(:,i,j,io))*unit; vq = sum(obs_Q(:,i,j,io)*cos2p + obs_U(:,i,j,io)*sin2p)*unit
             vu = sum(-obs_Q(:,i,j,io)*sin2p + obs_U(:,i,j,io)*cos2p)*unit; vv = sum(obs_V(:,i,j,io))*unit
          endif
          rad(ir,2) = rad(ir,2) + vi; rad(ir,3) = rad(ir,3) + vq; rad(ir,4) = rad(ir,4) + vu; rad(ir,5) = rad(ir,5) + vv
       enddo
    enddo
    do ir = 1, nr
       if (ncount(ir) > 0) then
          rad(ir,2:5) = rad(ir,2:5)/ncount(ir)
          if (st .and. rad(ir,2) > 0.0_wp) rad(ir,6) = sqrt(rad(ir,3)**2 + rad(ir,4)**2)/rad(ir,2)
       endif
    enddo
  end subroutine radial_profiles
end program LaRT_amd
