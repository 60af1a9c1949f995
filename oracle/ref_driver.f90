!------------------------------------------------------------------------------
! TEST INFRASTRUCTURE ONLY (never linked into, imported by or shipped with the
! product path).
!
! bind(C) entry points over the reference's OWN compiled modules, built by
! oracle/Makefile from the sources where they lie under /root/reference/src
! (define, random_mt [without -DMPI], voigt_mod, mathlib, healpix, line_mod,
! h2_mod, raytrace_car).  Nothing here restates reference arithmetic: each
! wrapper only marshals C arguments into the reference's derived types and
! calls the reference routine.  Used by tools/gen_golden.py to produce the
! vectors under tests/golden/ and by tests that run in the container where
! /root/reference is mounted.
!
! The rest of LaRT (grid_mod_car, scattering_car, generate_photon, peelingoff_*,
! run_simulation_mod, ...) needs `use mpi` and cannot be built here without
! stand-ins, so those routines are pinned through oracle/lart_oracle.c +
! the known answers in the shipped logs (DESIGN.md, "Oracle").
!------------------------------------------------------------------------------
module ref_driver
  use iso_c_binding
  use define
  use random
  use voigt_mod
  use mathlib
  use line_mod
  use raytrace
  implicit none
  type(grid_type), save, target :: g
contains

  !--- line constants: line_mod.f90:551-1272 (setup_resonance_line), Ly-alpha default
  subroutine ref_init_line(cross0, vtherm1, damping, wavelength0, g_recoil0) bind(C, name='ref_init_line')
    real(c_double), intent(out) :: cross0, vtherm1, damping, wavelength0, g_recoil0
    par%line_id = 'ly_alpha'
    par%fine_structure = .false.
    call setup_resonance_line()
    calc_voigt   => calc_voigt1
    do_resonance => do_resonance1
    cross0      = line%cross0
    vtherm1     = line%vtherm1
    damping     = line%damping
    wavelength0 = line%wavelength0
    g_recoil0   = line%g_recoil0
  end subroutine ref_init_line

  !--- par%fine_structure: line type 2 (Ly-alpha doublet) or back to type 1; sets the profile pointer as setup_procedure does
  subroutine ref_set_fine_structure(on, DnuHK_Hz) bind(C, name='ref_set_fine_structure')
    integer(c_int), value :: on
    real(c_double), intent(out) :: DnuHK_Hz
    par%line_id = 'ly_alpha'
    par%fine_structure = (on /= 0)
    call setup_resonance_line()
    if (par%fine_structure) then
       calc_voigt => calc_voigt2; do_resonance => do_resonance2
    else
       calc_voigt => calc_voigt1; do_resonance => do_resonance1
    endif
    DnuHK_Hz = line%DnuHK_Hz
  end subroutine ref_set_fine_structure

  !--- do_resonance2 (line_mod.f90:141-192) with the Stokes matrix elements and the per-photon E1..E3
  subroutine ref_do_resonance2_v(n, xfreq, ic, jc, kc, out) bind(C, name='ref_do_resonance2_v')
    integer(c_int), value :: n, ic, jc, kc
    real(c_double), value :: xfreq
    real(c_double), intent(out) :: out(12, n)      ! uz, xatom, cost, sint, S11, S22, S12, S33, S44, E1, E2, E3
    type(photon_type) :: ph
    integer :: i
    do i = 1, n
       ph%xfreq = xfreq
       ph%icell = ic; ph%jcell = jc; ph%kcell = kc
       call do_resonance2(ph, g, out(1,i), out(2,i), out(3,i), out(4,i), out(5,i), out(6,i), out(7,i), out(8,i), out(9,i))
       out(10,i) = ph%E1; out(11,i) = ph%E2; out(12,i) = ph%E3
    enddo
  end subroutine ref_do_resonance2_v

  !--- par%line_id = one of the single-line / doublet ids of setup_resonance_line (line_mod.f90:590-713), or 'ly_alpha' to go back;
  !--- out = line_type, wavelength0, damping, cross0, vtherm1, g_recoil0, DnuHK_Hz, E1, E2, E3
  subroutine ref_set_line_id(name, out) bind(C, name='ref_set_line_id')
    character(kind=c_char), intent(in) :: name(*)
    real(c_double), intent(out) :: out(10)
    integer :: i
    par%line_id = ''
    do i = 1, len(par%line_id)
       if (name(i) == c_null_char) exit
       par%line_id(i:i) = name(i)
    enddo
    par%fine_structure = .false.
    if (allocated(line%b)) deallocate(line%b)
    line%delE_Hz = 0.0_wp; line%f12 = 0.0_wp; line%nup = 1
    call setup_resonance_line()
    select case (line%line_type)                     ! as setup_procedure does (setup.f90:778-803)
    case (2); calc_voigt => calc_voigt2; do_resonance => do_resonance2
    case (4); calc_voigt => calc_voigt1; do_resonance => do_resonance4
    case (5); calc_voigt => calc_voigt3; do_resonance => do_resonance5
    case (6); calc_voigt => calc_voigt3; do_resonance => do_resonance6
    case default; calc_voigt => calc_voigt1; do_resonance => do_resonance1
    end select
    out = [real(line%line_type, c_double), line%wavelength0, line%damping, line%cross0, line%vtherm1, line%g_recoil0, line%DnuHK_Hz, &
           line%E1, line%E2, line%E3]
  end subroutine ref_set_line_id

  subroutine ref_set_HeI_coherent(on) bind(C, name='ref_set_HeI_coherent')
    integer(c_int), value :: on
    par%HeI_coherent = (on /= 0)
  end subroutine ref_set_HeI_coherent

  !--- line%b(:) of the current line (types 4, 5, 6): type, nup, ndown(3), f12(3), delE_Hz(3), damping(3), then for every (iup, idown) of a 3 x 4
  !--- table: A21, Elow_Hz, E1, E2, E3, P_down, A_down (zeros where the reference has no entry)
  subroutine ref_get_multiplet(out) bind(C, name='ref_get_multiplet')
    real(c_double), intent(out) :: out(98)
    integer :: i, d, q
    out = 0.0_c_double
    out(1) = line%line_type; out(2) = line%nup
    do i = 1, line%nup
       out(2+i) = line%b(i)%ndown; out(5+i) = line%f12(i); out(11+i) = line%b(i)%damping
       if (line%nup > 1) out(8+i) = line%delE_Hz(i)
       do d = 1, line%b(i)%ndown
          q = 14 + ((i-1)*4 + (d-1))*7
          out(q+1) = line%b(i)%A21(d); out(q+2) = line%b(i)%Elow_Hz(d); out(q+3) = line%b(i)%E1(d); out(q+4) = line%b(i)%E2(d); out(q+5) = line%b(i)%E3(d)
          if (allocated(line%b(i)%P_down)) then
             out(q+6) = line%b(i)%P_down(d); out(q+7) = line%b(i)%A_down(d)
          endif
       enddo
    enddo
  end subroutine ref_get_multiplet

  !--- do_resonance through the pointer (do_resonance4 / 5 / 6 ...): out(:,i) = uz, xatom, cost, sint, S11, S22, S12, S33, S44, E1, E2, E3
  subroutine ref_do_resonance_ptr_v(n, xfreq, ic, jc, kc, out) bind(C, name='ref_do_resonance_ptr_v')
    integer(c_int), value :: n, ic, jc, kc
    real(c_double), value :: xfreq
    real(c_double), intent(out) :: out(12, n)
    type(photon_type) :: ph
    integer :: i
    do i = 1, n
       ph%xfreq = xfreq
       ph%icell = ic; ph%jcell = jc; ph%kcell = kc
       call do_resonance(ph, g, out(1,i), out(2,i), out(3,i), out(4,i), out(5,i), out(6,i), out(7,i), out(8,i), out(9,i))
       out(10,i) = ph%E1; out(11,i) = ph%E2; out(12,i) = ph%E3
    enddo
  end subroutine ref_do_resonance_ptr_v

  !--- par%include_deuterium: 'ly_alpha_HD' (line type 7) or back to plain Ly-alpha; out = delta_nu_HD_Hz, ratio_Dfreq_HD, ratio_voigta_HD, g_recoil0_D
  subroutine ref_set_deuterium(on, D_to_H, out) bind(C, name='ref_set_deuterium')
    integer(c_int), value :: on
    real(c_double), value :: D_to_H
    real(c_double), intent(out) :: out(4)
    par%fine_structure = .false.
    par%D_to_H_ratio = D_to_H
    if (on /= 0) then
       par%line_id = 'ly_alpha_HD'
    else
       par%line_id = 'ly_alpha'
    endif
    call setup_resonance_line()
    if (on /= 0) then
       calc_voigt => calc_voigt_HD; do_resonance => do_resonance_HD
    else
       calc_voigt => calc_voigt1; do_resonance => do_resonance1
    endif
    out = [line%delta_nu_HD_Hz, line%ratio_Dfreq_HD, line%ratio_voigta_HD, line%g_recoil0_D]
  end subroutine ref_set_deuterium

  !--- do_resonance_HD (line_mod.f90:476-549): out(:,i) = uz, xatom, cost, sint, species (1 H, 2 D)
  subroutine ref_do_resonance_HD_v(n, xfreq, ic, jc, kc, out) bind(C, name='ref_do_resonance_HD_v')
    integer(c_int), value :: n, ic, jc, kc
    real(c_double), value :: xfreq
    real(c_double), intent(out) :: out(5, n)
    type(photon_type) :: ph
    integer :: i
    do i = 1, n
       ph%xfreq = xfreq
       ph%icell = ic; ph%jcell = jc; ph%kcell = kc
       call do_resonance_HD(ph, g, out(1,i), out(2,i), out(3,i), out(4,i))
       out(5,i) = real(line%selected_species_HD, c_double)
    enddo
  end subroutine ref_do_resonance_HD_v

  !--- calc_voigt through the pointer (calc_voigt1 or calc_voigt2, line_mod.f90:38-62) in cell (ic,jc,kc)
  subroutine ref_calc_voigt_v(n, x, ic, jc, kc, h) bind(C, name='ref_calc_voigt_v')
    integer(c_int), value :: n, ic, jc, kc
    real(c_double), intent(in) :: x(n)
    real(c_double), intent(out) :: h(n)
    integer :: i
    do i = 1, n
       h(i) = calc_voigt(g, x(i), ic, jc, kc)
    enddo
  end subroutine ref_calc_voigt_v

  function ref_voigt(x, a) bind(C, name='ref_voigt') result(h)
    real(c_double), value :: x, a
    real(c_double) :: h
    h = voigt(x, a)
  end function ref_voigt

  subroutine ref_voigt_v(n, x, a, h) bind(C, name='ref_voigt_v')
    integer(c_int), value :: n
    real(c_double), intent(in)  :: x(n), a(n)
    real(c_double), intent(out) :: h(n)
    integer :: i
    do i = 1, n
       h(i) = voigt(x(i), a(i))
    enddo
  end subroutine ref_voigt_v

  !--- RNG: random_mt.f90 init_random_mt_seed (:889-956) / rand_mt (:579-630)
  subroutine ref_seed(iseed) bind(C, name='ref_seed')
    integer(c_int), value :: iseed
    integer :: is
    is = iseed
    call init_random_seed(is)
  end subroutine ref_seed

  subroutine ref_rand_number_v(n, u) bind(C, name='ref_rand_number_v')
    integer(c_int), value :: n
    real(c_double), intent(out) :: u(n)
    integer :: i
    do i = 1, n
       u(i) = rand_number()
    enddo
  end subroutine ref_rand_number_v

  !--- samplers, n draws each from the current MT state
  subroutine ref_rand_resonance_vz_v(n, x, a, vz) bind(C, name='ref_rand_resonance_vz_v')
    integer(c_int), value :: n
    real(c_double), value :: x, a
    real(c_double), intent(out) :: vz(n)
    integer :: i
    do i = 1, n
       vz(i) = rand_resonance_vz(x, a)
    enddo
  end subroutine ref_rand_resonance_vz_v

  subroutine ref_rand_resonance_v(n, E1, c) bind(C, name='ref_rand_resonance_v')
    integer(c_int), value :: n
    real(c_double), value :: E1
    real(c_double), intent(out) :: c(n)
    integer :: i
    do i = 1, n
       c(i) = rand_resonance(E1)
    enddo
  end subroutine ref_rand_resonance_v

  subroutine ref_rand_hg_v(n, gg, c) bind(C, name='ref_rand_hg_v')
    integer(c_int), value :: n
    real(c_double), value :: gg
    real(c_double), intent(out) :: c(n)
    integer :: i
    do i = 1, n
       c(i) = rand_henyey_greenstein(gg)
    enddo
  end subroutine ref_rand_hg_v

  subroutine ref_rand_voigt_v(n, a, c) bind(C, name='ref_rand_voigt_v')
    integer(c_int), value :: n
    real(c_double), value :: a
    real(c_double), intent(out) :: c(n)
    integer :: i
    do i = 1, n
       c(i) = rand_voigt(a)
    enddo
  end subroutine ref_rand_voigt_v

  subroutine ref_rand_gauss_v(n, c) bind(C, name='ref_rand_gauss_v')
    integer(c_int), value :: n
    real(c_double), intent(out) :: c(n)
    integer :: i
    do i = 1, n
       c(i) = rand_gauss()
    enddo
  end subroutine ref_rand_gauss_v

  !--- alias tables (random_mt.f90:2016-2238)
  subroutine ref_alias_setup(n, probs, alias) bind(C, name='ref_alias_setup')
    integer(c_int), value :: n
    real(c_double), intent(inout) :: probs(n)
    integer(c_int), intent(out)   :: alias(n)
    integer :: al(n)
    call random_alias_setup(probs, al)
    alias(:) = al(:)
  end subroutine ref_alias_setup

  subroutine ref_rand_alias_linear_v(m, n, prob_alias, alias, xaxis, prob, xout) bind(C, name='ref_rand_alias_linear_v')
    integer(c_int), value :: m, n
    real(c_double), intent(in) :: prob_alias(n), xaxis(n+1), prob(n+1)
    integer(c_int), intent(in) :: alias(n)
    real(c_double), intent(out) :: xout(m)
    integer :: i, al(n)
    al(:) = alias(:)
    do i = 1, m
       xout(i) = rand_alias_linear(prob_alias, al, xaxis, prob)
    enddo
  end subroutine ref_rand_alias_linear_v

  !--- mathlib.f90:71-106
  subroutine ref_interp_eq(n, x, y, xnew, ynew) bind(C, name='ref_interp_eq')
    integer(c_int), value :: n
    real(c_double), intent(in) :: x(n), y(n)
    real(c_double), value :: xnew
    real(c_double), intent(out) :: ynew
    call interp_eq(x, y, xnew, ynew)
  end subroutine ref_interp_eq

  !--- healpix.f90:29-112
  subroutine ref_vec2pix(nside, vx, vy, vz, ipix) bind(C, name='ref_vec2pix')
    use healpix, only: vec2pix
    integer(c_int), value :: nside
    real(c_double), value :: vx, vy, vz
    integer(c_int), intent(out) :: ipix
    integer :: ip
    call vec2pix(nside, vx, vy, vz, ip)
    ipix = ip
  end subroutine ref_vec2pix

  !--- do_resonance1: line_mod.f90:108-139 (uses grid%voigt_a at the photon cell)
  subroutine ref_do_resonance1_v(n, xfreq, ic, jc, kc, uz, xatom, cost, sint) bind(C, name='ref_do_resonance1_v')
    integer(c_int), value :: n, ic, jc, kc
    real(c_double), value :: xfreq
    real(c_double), intent(out) :: uz(n), xatom(n), cost(n), sint(n)
    type(photon_type) :: ph
    integer :: i
    do i = 1, n
       ph%xfreq = xfreq
       ph%icell = ic; ph%jcell = jc; ph%kcell = kc
       call do_resonance1(ph, g, uz(i), xatom(i), cost(i), sint(i))
    enddo
  end subroutine ref_do_resonance1_v

  !--- switches of par% that the compiled raytrace routines read
  subroutine ref_set_par(DGR, save_Jmu, nmu, xyz_symmetry) bind(C, name='ref_set_par')
    real(c_double), value :: DGR
    integer(c_int), value :: save_Jmu, nmu, xyz_symmetry
    par%DGR = DGR
    par%save_Jmu = (save_Jmu /= 0)
    par%nmu = nmu
    par%xyz_symmetry = (xyz_symmetry /= 0)
    if (par%xyz_symmetry) then
       par%mu_min = 0.0_wp;  par%dmu = 1.0_wp/par%nmu
    else
       par%mu_min = -1.0_wp; par%dmu = 2.0_wp/par%nmu
    endif
  end subroutine ref_set_par

  !--- fill the reference grid_type from caller arrays (column-major nx,ny,nz)
  subroutine ref_set_grid(nx, ny, nz, xmin, ymin, zmin, dx, dy, dz, i0, j0, k0, Dfreq_ref, &
                          xfreq_min, dxfreq, nxfreq, rhokap, voigt_a, Dfreq, vfx, vfy, vfz, rhokapD) &
                          bind(C, name='ref_set_grid')
    integer(c_int), value :: nx, ny, nz, i0, j0, k0, nxfreq
    real(c_double), value :: xmin, ymin, zmin, dx, dy, dz, Dfreq_ref, xfreq_min, dxfreq
    real(c_double), intent(in) :: rhokap(nx,ny,nz), voigt_a(nx,ny,nz), Dfreq(nx,ny,nz)
    real(c_double), intent(in) :: vfx(nx,ny,nz), vfy(nx,ny,nz), vfz(nx,ny,nz), rhokapD(nx,ny,nz)
    integer :: i
    call ref_free_grid()
    g%nx = nx; g%ny = ny; g%nz = nz
    g%i0 = i0; g%j0 = j0; g%k0 = k0
    g%dx = dx; g%dy = dy; g%dz = dz
    g%xmin = xmin; g%ymin = ymin; g%zmin = zmin
    g%xmax = xmin + nx*dx; g%ymax = ymin + ny*dy; g%zmax = zmin + nz*dz
    g%xrange = g%xmax - g%xmin; g%yrange = g%ymax - g%ymin; g%zrange = g%zmax - g%zmin
    g%Dfreq_ref = Dfreq_ref
    g%xfreq_min = xfreq_min; g%dxfreq = dxfreq; g%nxfreq = nxfreq
    g%xfreq_max = xfreq_min + nxfreq*dxfreq
    allocate(g%xface(nx+1), g%yface(ny+1), g%zface(nz+1))
    !-- same expression as grid_mod_car.f90:184-187
    g%xface(:) = [ ((i-1)*g%dx + g%xmin, i=1,nx+1) ]
    g%yface(:) = [ ((i-1)*g%dy + g%ymin, i=1,ny+1) ]
    g%zface(:) = [ ((i-1)*g%dz + g%zmin, i=1,nz+1) ]
    allocate(g%rhokap(nx,ny,nz), g%voigt_a(nx,ny,nz), g%Dfreq(nx,ny,nz))
    allocate(g%vfx(nx,ny,nz), g%vfy(nx,ny,nz), g%vfz(nx,ny,nz), g%rhokapD(nx,ny,nz))
    g%rhokap = rhokap; g%voigt_a = voigt_a; g%Dfreq = Dfreq
    g%vfx = vfx; g%vfy = vfy; g%vfz = vfz; g%rhokapD = rhokapD
    allocate(g%Jout(nxfreq)); g%Jout = 0.0_wp
    allocate(g%Jmu(nxfreq, max(par%nmu,1))); g%Jmu = 0.0_wp
  end subroutine ref_set_grid

  subroutine ref_free_grid() bind(C, name='ref_free_grid')
    if (associated(g%xface))   deallocate(g%xface)
    if (associated(g%yface))   deallocate(g%yface)
    if (associated(g%zface))   deallocate(g%zface)
    if (associated(g%rhokap))  deallocate(g%rhokap)
    if (associated(g%voigt_a)) deallocate(g%voigt_a)
    if (associated(g%Dfreq))   deallocate(g%Dfreq)
    if (associated(g%vfx))     deallocate(g%vfx)
    if (associated(g%vfy))     deallocate(g%vfy)
    if (associated(g%vfz))     deallocate(g%vfz)
    if (associated(g%rhokapD)) deallocate(g%rhokapD)
    if (associated(g%Jout))    deallocate(g%Jout)
    if (associated(g%Jmu))     deallocate(g%Jmu)
  end subroutine ref_free_grid

  subroutine ref_get_faces(xf, yf, zf) bind(C, name='ref_get_faces')
    real(c_double), intent(out) :: xf(g%nx+1), yf(g%ny+1), zf(g%nz+1)
    xf = g%xface; yf = g%yface; zf = g%zface
  end subroutine ref_get_faces

  subroutine ref_get_Jout(J) bind(C, name='ref_get_Jout')
    real(c_double), intent(out) :: J(g%nxfreq)
    J = g%Jout
  end subroutine ref_get_Jout

  subroutine ref_zero_Jout() bind(C, name='ref_zero_Jout')
    g%Jout = 0.0_wp
    g%Jmu  = 0.0_wp
  end subroutine ref_zero_Jout

  !--- photon state marshalling: r = [x,y,z,kx,ky,kz,xfreq,wgt], ic = [icell,jcell,kcell,inside]
  subroutine unpack(r, ic, ph)
    real(c_double), intent(in) :: r(8)
    integer(c_int), intent(in) :: ic(4)
    type(photon_type), intent(out) :: ph
    ph%x = r(1); ph%y = r(2); ph%z = r(3)
    ph%kx = r(4); ph%ky = r(5); ph%kz = r(6)
    ph%xfreq = r(7); ph%wgt = r(8)
    ph%icell = ic(1); ph%jcell = ic(2); ph%kcell = ic(3)
    ph%inside = (ic(4) /= 0)
    ph%iband = 1
    ph%vfy_shear = 0.0_wp
  end subroutine unpack

  subroutine pack(ph, r, ic)
    type(photon_type), intent(in) :: ph
    real(c_double), intent(out) :: r(8)
    integer(c_int), intent(out) :: ic(4)
    r(1) = ph%x; r(2) = ph%y; r(3) = ph%z
    r(4) = ph%kx; r(5) = ph%ky; r(6) = ph%kz
    r(7) = ph%xfreq; r(8) = ph%wgt
    ic(1) = ph%icell; ic(2) = ph%jcell; ic(3) = ph%kcell
    ic(4) = merge(1, 0, ph%inside)
  end subroutine pack

  !--- variant: 0 = raytrace_to_tau_car (:1425), 1 = _car_xyzsym (:1650), 2 = _car_zonly (:2519), 5 = _car_xysym (:1951), 6 = _car_xyper (:2252)
  subroutine ref_raytrace_to_tau(variant, r, ic, tau_in) bind(C, name='ref_raytrace_to_tau')
    integer(c_int), value :: variant
    real(c_double), intent(inout) :: r(8)
    integer(c_int), intent(inout) :: ic(4)
    real(c_double), value :: tau_in
    type(photon_type) :: ph
    call unpack(r, ic, ph)
    select case (variant)
    case (0); call raytrace_to_tau_car(ph, g, tau_in)
    case (1); call raytrace_to_tau_car_xyzsym(ph, g, tau_in)
    case (2); call raytrace_to_tau_car_zonly(ph, g, tau_in)
    case (5); call raytrace_to_tau_car_xysym(ph, g, tau_in)
    case (6); call raytrace_to_tau_car_xyper(ph, g, tau_in)
    end select
    call pack(ph, r, ic)
  end subroutine ref_raytrace_to_tau

  !--- raytrace_to_tau_car_xyper_shear (:2677-2954): the xy-periodic march with the shearing x boundary; Omega = par%Omega as grid_create
  !--- leaves it (the change of vfy per x wrap, thermal units); shear = photon%vfy_shear in / out
  subroutine ref_raytrace_to_tau_shear(r, ic, tau_in, Omega, shear) bind(C, name='ref_raytrace_to_tau_shear')
    real(c_double), intent(inout) :: r(8)
    integer(c_int), intent(inout) :: ic(4)
    real(c_double), value :: tau_in, Omega
    real(c_double), intent(inout) :: shear
    type(photon_type) :: ph
    call unpack(r, ic, ph)
    par%Omega = Omega
    ph%vfy_shear = shear
    call raytrace_to_tau_car_xyper_shear(ph, g, tau_in)
    shear = ph%vfy_shear
    call pack(ph, r, ic)
  end subroutine ref_raytrace_to_tau_shear

  !--- line type 8 ('ly_beta', line_mod.f90:1192-1240): do_resonance8 (:293-345); out(:,i) = uz, xatom, cost, sint, E1, E2, E3, iband
  subroutine ref_do_resonance8_v(n, xfreq, ic, jc, kc, out) bind(C, name='ref_do_resonance8_v')
    integer(c_int), value :: n, ic, jc, kc
    real(c_double), value :: xfreq
    real(c_double), intent(out) :: out(8, n)
    type(photon_type) :: ph
    integer :: i
    do i = 1, n
       ph%xfreq = xfreq; ph%iband = 1
       ph%icell = ic; ph%jcell = jc; ph%kcell = kc
       call do_resonance8(ph, g, out(1,i), out(2,i), out(3,i), out(4,i))
       out(5,i) = ph%E1; out(6,i) = ph%E2; out(7,i) = ph%E3; out(8,i) = ph%iband
    enddo
  end subroutine ref_do_resonance8_v

  !--- band 2 of line type 8: the H-alpha spectral axis and dust ratio the marches read (grid%xfreq_min_Ha, dxfreq_Ha, nxfreq_Ha, par%R_Ha)
  subroutine ref_set_band2(xfreq_min_Ha, dxfreq_Ha, nxfreq_Ha, R_Ha) bind(C, name='ref_set_band2')
    real(c_double), value :: xfreq_min_Ha, dxfreq_Ha, R_Ha
    integer(c_int), value :: nxfreq_Ha
    g%xfreq_min_Ha = xfreq_min_Ha; g%dxfreq_Ha = dxfreq_Ha; g%nxfreq_Ha = nxfreq_Ha
    g%xfreq_max_Ha = xfreq_min_Ha + nxfreq_Ha*dxfreq_Ha
    par%R_Ha = R_Ha
    if (associated(g%Jout_Ha)) deallocate(g%Jout_Ha)
    allocate(g%Jout_Ha(nxfreq_Ha)); g%Jout_Ha = 0.0_wp
    par%W_esc1 = 0.0_wp; par%W_esc2 = 0.0_wp
  end subroutine ref_set_band2

  !--- raytrace_to_tau_car (:1425-1648) with photon%iband given (2: dust-only opacity rhokapD R_Ha, lab-frame frequency, Jout_Ha / W_esc2)
  subroutine ref_raytrace_to_tau_band(iband, r, ic, tau_in) bind(C, name='ref_raytrace_to_tau_band')
    integer(c_int), value :: iband
    real(c_double), intent(inout) :: r(8)
    integer(c_int), intent(inout) :: ic(4)
    real(c_double), value :: tau_in
    type(photon_type) :: ph
    call unpack(r, ic, ph)
    ph%iband = iband
    call raytrace_to_tau_car(ph, g, tau_in)
    call pack(ph, r, ic)
    if (.not. ph%inside) r(7) = ph%xfreq_ref
  end subroutine ref_raytrace_to_tau_band

  !--- raytrace_to_edge_car_dust_Ha (:510-582)
  subroutine ref_raytrace_to_edge_dust_Ha(r, ic, tau) bind(C, name='ref_raytrace_to_edge_dust_Ha')
    real(c_double), intent(in) :: r(8)
    integer(c_int), intent(in) :: ic(4)
    real(c_double), intent(out) :: tau
    type(photon_type) :: ph
    call unpack(r, ic, ph)
    ph%iband = 2
    call raytrace_to_edge_car_dust_Ha(ph, g, tau)
  end subroutine ref_raytrace_to_edge_dust_Ha

  subroutine ref_get_band2(Jout_Ha, W) bind(C, name='ref_get_band2')
    real(c_double), intent(out) :: Jout_Ha(g%nxfreq_Ha), W(2)
    Jout_Ha = g%Jout_Ha; W = [par%W_esc1, par%W_esc2]
  end subroutine ref_get_band2

  !--- H2 lines next to Ly-alpha (h2_mod.f90): h2_init with par%h2_model = 'neufeld' reads <data_dir>/energy_X.dat; T_h2 < 0 switches the lines off.
  !--- out = n_h2_lines, h2_Dfreq_Hz, then per line dnu_Hz, strength, a_damp, p_scat, pop, f_osc
  subroutine ref_h2_init(data_dir, T_h2, f_H2, hi_width, pure_absorption, bturb, out) bind(C, name='ref_h2_init')
    use h2_mod
    character(kind=c_char), intent(in) :: data_dir(*)
    real(c_double), value :: T_h2, f_H2, bturb
    integer(c_int), value :: hi_width, pure_absorption
    real(c_double), intent(out) :: out(26)
    character(len=256) :: dir
    integer :: i
    dir = ''
    do i = 1, len(dir)
       if (data_dir(i) == c_null_char) exit
       dir(i:i) = data_dir(i)
    enddo
    out = 0.0_c_double
    if (T_h2 < 0.0_wp) then
       par%h2_model = 'none'; call h2_init(trim(dir)); return
    endif
    par%h2_model = 'neufeld'; par%h2_temperature = T_h2; par%f_H2 = f_H2
    par%h2_hi_width = (hi_width /= 0); par%h2_pure_absorption = (pure_absorption /= 0); par%bturb = bturb
    call h2_init(trim(dir))
    out(1) = n_h2_lines; out(2) = h2_Dfreq_Hz
    do i = 1, n_h2_lines
       out(2+(i-1)*6+1) = h2l(i)%dnu_Hz; out(2+(i-1)*6+2) = h2l(i)%strength; out(2+(i-1)*6+3) = h2l(i)%a_damp
       out(2+(i-1)*6+4) = h2l(i)%p_scat; out(2+(i-1)*6+5) = h2l(i)%pop; out(2+(i-1)*6+6) = h2l(i)%f_osc
    enddo
  end subroutine ref_h2_init

  !--- h2_kappa_D (:58-78) and h2_select_line_D (:81-115; consumes one random number per call)
  subroutine ref_h2_kappa_v(n, x, Dfreq, kap) bind(C, name='ref_h2_kappa_v')
    use h2_mod
    integer(c_int), value :: n
    real(c_double), value :: Dfreq
    real(c_double), intent(in) :: x(n)
    real(c_double), intent(out) :: kap(n)
    integer :: i
    do i = 1, n
       kap(i) = h2_kappa_D(x(i), Dfreq)
    enddo
  end subroutine ref_h2_kappa_v

  subroutine ref_h2_select_v(n, x, Dfreq, il) bind(C, name='ref_h2_select_v')
    use h2_mod
    integer(c_int), value :: n
    real(c_double), value :: Dfreq
    real(c_double), intent(in) :: x(n)
    integer(c_int), intent(out) :: il(n)
    integer :: i
    do i = 1, n
       il(i) = h2_select_line_D(x(i), Dfreq)
    enddo
  end subroutine ref_h2_select_v

  !--- variant: 0 = raytrace_to_edge_car (:410), 1 = _car_xyzsym (:584), 2 = _car_zonly (:1138), 5 = _car_xysym (:783), 6 = _car_xyper (:971)
  subroutine ref_raytrace_to_edge(variant, r, ic, tau) bind(C, name='ref_raytrace_to_edge')
    integer(c_int), value :: variant
    real(c_double), intent(in) :: r(8)
    integer(c_int), intent(in) :: ic(4)
    real(c_double), intent(out) :: tau
    type(photon_type) :: ph
    call unpack(r, ic, ph)
    select case (variant)
    case (0); call raytrace_to_edge_car(ph, g, tau)
    case (1); call raytrace_to_edge_car_xyzsym(ph, g, tau)
    case (2); call raytrace_to_edge_car_zonly(ph, g, tau)
    case (5); call raytrace_to_edge_car_xysym(ph, g, tau)
    case (6); call raytrace_to_edge_car_xyper(ph, g, tau)
    end select
  end subroutine ref_raytrace_to_edge

  !--- raytrace_to_dist_car (:89-195)
  subroutine ref_raytrace_to_dist(r, ic, dist, tau) bind(C, name='ref_raytrace_to_dist')
    real(c_double), intent(in) :: r(8)
    integer(c_int), intent(in) :: ic(4)
    real(c_double), value :: dist
    real(c_double), intent(out) :: tau
    type(photon_type) :: ph
    call unpack(r, ic, ph)
    call raytrace_to_dist_car(ph, g, dist, tau)
  end subroutine ref_raytrace_to_dist

  !--- sight-line marches: raytrace_to_edge_car_tau_gas (:1236-1327), raytrace_to_dist_tau_gas_car (:197-297);
  !    dist < 0 selects the to-edge routine
  subroutine ref_raytrace_tau_gas(r, ic, dist, tau) bind(C, name='ref_raytrace_tau_gas')
    real(c_double), intent(in) :: r(8)
    integer(c_int), intent(in) :: ic(4)
    real(c_double), value :: dist
    real(c_double), intent(out) :: tau
    type(photon_type) :: ph
    call unpack(r, ic, ph)
    if (dist < 0.0_wp) then
       call raytrace_to_edge_car_tau_gas(ph, g, tau)
    else
       call raytrace_to_dist_tau_gas_car(ph, g, dist, tau)
    endif
  end subroutine ref_raytrace_tau_gas

  !--- raytrace_to_edge_car_column (:1329-1423), raytrace_to_dist_column_car (:299-408)
  subroutine ref_raytrace_column(r, ic, dist, N_gas, tau_dust) bind(C, name='ref_raytrace_column')
    real(c_double), intent(in) :: r(8)
    integer(c_int), intent(in) :: ic(4)
    real(c_double), value :: dist
    real(c_double), intent(out) :: N_gas, tau_dust
    type(photon_type) :: ph
    call unpack(r, ic, ph)
    if (dist < 0.0_wp) then
       call raytrace_to_edge_car_column(ph, g, N_gas, tau_dust)
    else
       call raytrace_to_dist_column_car(ph, g, dist, N_gas, tau_dust)
    endif
  end subroutine ref_raytrace_column

  !--- healpix.f90:114-187 (RING pixel centre)
  subroutine ref_pix2vec(nside, ipix, v) bind(C, name='ref_pix2vec')
    use healpix, only: pix2vec
    integer(c_int), value :: nside, ipix
    real(c_double), intent(out) :: v(3)
    real(wp) :: vx, vy, vz
    call pix2vec(nside, ipix, vx, vy, vz)
    v = [vx, vy, vz]
  end subroutine ref_pix2vec

#if defined(CALCJ) && defined(CALCPnew)
  !--- profile tallies compiled into raytrace_car.f90 by -DCALCJ -DCALCPnew (add_to_J :3979-4011, add_to_Pnew :4015-4045)
  subroutine ref_set_JPa(geometry_JPa, nbins, ind) bind(C, name='ref_set_JPa')
    integer(c_int), value :: geometry_JPa, nbins
    integer(c_int), intent(in) :: ind(g%nx, g%ny, g%nz)
    g%geometry_JPa = geometry_JPa
    g%nr = nbins
    if (associated(g%ind_sph)) deallocate(g%ind_sph)
    if (associated(g%J1))      deallocate(g%J1)
    if (associated(g%P1_new))  deallocate(g%P1_new)
    allocate(g%ind_sph(g%nx, g%ny, g%nz)); g%ind_sph = ind
    allocate(g%J1(g%nxfreq, nbins), g%P1_new(nbins)); g%J1 = 0.0_wp; g%P1_new = 0.0_wp
  end subroutine ref_set_JPa

  subroutine ref_get_JP(nbins, J1, P1_new) bind(C, name='ref_get_JP')
    integer(c_int), value :: nbins
    real(c_double), intent(out) :: J1(g%nxfreq, nbins), P1_new(nbins)
    J1 = g%J1; P1_new = g%P1_new
  end subroutine ref_get_JP
#endif

end module ref_driver
