"""Python mirror of the reference's host-side interface around the photon path.

Same names, argument meaning and error behaviour as the Fortran host of the reference:
    read_input        setup.f90:4-579            namelist &parameters par%... (lart_amd/params.py) + post-processing
    setup_resonance_line  line_mod.f90:551-1272  -> lart_host_line_setup / lart_host_line_lookup
    grid_create       grid_mod_car.f90:11-1238   -> lart_host_car_medium + lart_host_freq_grid (analytic sphere / slab / box subset)
    observer_create   observer_rect.f90:24-326   -> lart_host_observers
    run_simulation    main.f90:42                -> lart_gpu_run through the C-ABI (lart_amd.capi)
    output_reduce     main.f90:46                -> RCCL sum of the fused device tally buffer
    output_normalize  output_sum_rect.f90:151-487 -> lart_host_norm_scales (Jout/Jin/Jabs/Jmu, peel cubes, the band-2 sections of ly_beta)
This module is plumbing: the set-up arithmetic lives once, in the library (lart_amd/csrc/lart_host_setup.cpp, bound by lart_amd/hostlib.py and by the
Fortran host lart_amd/fhost/*.f90 alike); all transport arithmetic runs in liblart_gpu.so.
"""
import ctypes as C
import math
import os
from dataclasses import dataclass
import numpy as np

from . import capi, hostlib
from .params import LartInputError, Params, MAX_OBSERVERS, NAN, read_namelist, set_param, _f32, _parse_value     # noqa: F401  (re-exported: the namelist surface)
from .postproc import radial_profiles                                                               # noqa: F401  (re-exported)

PI = 3.141592653589793238462643383279502884197
TWOPI, FOURPI = 2 * PI, 4 * PI
UM2KM, UM2M = 1.0e-9, 1.0e-6
SPEEDC = 2.99792458e5
DEG2RAD, RAD2DEG = PI / 180.0, 180.0 / PI
_HL = hostlib.lib


def _chk(rc):
    hostlib.check(rc, LartInputError)


@dataclass
class Line:
    """line_type (define.f90:639-680) as lart_host_line_setup / lart_host_line_lookup fill it (Ly-alpha by default)"""
    wavelength0: float = 0.0
    damping: float = 0.0
    cross0: float = 0.0
    vtherm1: float = 0.0
    g_recoil0: float = 0.0
    E1: float = 1.0
    E2: float = 0.0
    E3: float = 1.0
    line_type: int = 1
    delta_nu_HD_Hz: float = 0.0     # line type 7 (H + D): the cross-species constants of line_mod.f90:1186-1191
    ratio_Dfreq_HD: float = 1.0
    ratio_voigta_HD: float = 1.0
    nD_over_nH: float = 0.0
    g_recoil0_D: float = 0.0
    bturb: float = -999.0           # par%bturb (km/s)
    multiplet: object = None        # capi.LartLineMultiplet for line types 4, 5, 6 (line%b(:))
    DnuHK_Hz: float = 0.0           # 2P3/2 - 2P1/2 splitting (par%fine_structure: line type 2, line_mod.f90:1250-1253)
    P_conv: float = 0.0             # line type 8: line%b(1)%P_down(2), the 3p -> 2s branch (line_mod.f90:1230-1238)
    E_conv: tuple = (1.0, 0.0, 1.0)
    wavelength0_Ha: float = 0.6564553
    mass_amu: float = 1.00797
    h2: object = None               # H2Lines (h2_init) when par%h2_model /= 'none'

    def __post_init__(self):
        if self.wavelength0 == 0.0:
            _fill_line(self, "ly_alpha", False, False, 0.0)


def _fill_line(line, line_id, fine_structure, include_deuterium, D_to_H):
    """setup_resonance_line through the library; returns (rc, record): rc = 2 for the ids of the multiplet table"""
    h = hostlib.HostLine()
    rc = _HL().lart_host_line_setup(line_id.encode(), int(fine_structure), int(include_deuterium), float(D_to_H), C.byref(h))
    if rc == 0:
        for n in ("line_type", "wavelength0", "damping", "cross0", "vtherm1", "g_recoil0", "DnuHK_Hz", "E1", "E2", "E3", "delta_nu_HD_Hz", "ratio_Dfreq_HD",
                  "ratio_voigta_HD", "nD_over_nH", "g_recoil0_D", "P_conv", "wavelength0_Ha", "mass_amu"):
            setattr(line, n, getattr(h, n))
        line.E_conv = tuple(h.E_conv)
    return rc, h


def vtherm_total(line, T):
    """define.f90:928-933: thermal width and par%bturb (line.bturb, set by read_input) in quadrature; T may be an array"""
    vt = line.vtherm1 * np.sqrt(T)
    if line.bturb > 0.0:
        vt = np.sqrt(vt * vt + line.bturb * line.bturb)
    return vt if isinstance(vt, np.ndarray) else float(vt)


@dataclass
class ScattMat:
    nPDF: int = 0
    coss: np.ndarray = None
    S11: np.ndarray = None
    S12: np.ndarray = None
    S33: np.ndarray = None
    S34: np.ndarray = None
    phase_PDF: np.ndarray = None
    alias: np.ndarray = None


def random_alias_setup(probs):
    """random_mt.f90:2016-2067; returns (scaled probs, 1-based alias)"""
    p = np.array(probs, dtype=np.float64); alias = np.zeros(len(p), dtype=np.int32)
    _chk(_HL().lart_host_alias_setup(len(p), hostlib.p(p), hostlib.p(alias)))
    return p, alias


def setup_scattering_matrix(par, path):
    """setup.f90:581-649: the header overrides albedo / hgg / cext_dust; the table is normalised and given its alias table by lart_host_mueller_setup"""
    with open(path) as f:
        f.readline()
        h = f.readline().split()
        f.readline()
        par.cext_dust, par.albedo, par.hgg, n = float(h[1]), float(h[2]), float(h[3]), int(h[4])
        tab = np.array([[float(t) for t in f.readline().split()[:5]] for _ in range(n)])
    sm = ScattMat(nPDF=n, coss=tab[:, 0].copy(), S11=tab[:, 1].copy(), S12=tab[:, 2].copy(), S33=tab[:, 3].copy(), S34=tab[:, 4].copy())
    sm.phase_PDF, sm.alias = np.zeros(n - 1), np.zeros(n - 1, dtype=np.int32)
    _chk(_HL().lart_host_mueller_setup(n, *[hostlib.p(a) for a in (sm.coss, sm.S11, sm.S12, sm.S33, sm.S34, sm.phase_PDF, sm.alias)]))
    return sm





# DnuHK_Hz (setup_resonance_line, line_mod.f90:590-713).  As in the reference, CIV_1548 recoils with line%mass_amu's default 1.00797
# (define.f90:646, line_mod.f90:605) and AlII_1671's mass is a single precision literal (:706).






_SOURCE_GEOMETRIES = {"point": 0, "uniform_sphere": 1, "uniform_cylinder": 2, "uniform": 3, "uniform_xy": 4, "gaussian": 5}


@dataclass
class H2Lines:
    """h2_mod.f90:30-49: the H2 lines next to Ly-alpha as h2_init leaves them"""
    n: int = 0
    Dfreq_Hz: float = 0.0
    dv_kms: tuple = ()
    dnu_Hz: tuple = ()
    strength: tuple = ()
    a_damp: tuple = ()
    p_scat: tuple = ()
    pop: tuple = ()
    f_osc: tuple = ()





def h2_init(par, line):
    """h2_init (h2_mod.f90:120-296) through lart_host_h2_init: the two B-X lines next to Ly-alpha with LTE populations at par%h2_temperature from
    <par%h2_data_dir>/energy_X.dat, strengths relative to the H I line-centre opacity"""
    h = hostlib.HostH2()
    _chk(_HL().lart_host_h2_init(os.path.join(par.h2_data_dir or "data/h2", "energy_X.dat").encode(), line.wavelength0, line.vtherm1, line.mass_amu, float(par.bturb),
                                 float(par.h2_temperature), float(par.f_H2), int(par.h2_pure_absorption), C.byref(h)))
    v = lambda a: tuple(a[i] for i in range(h.n))
    return H2Lines(n=h.n, Dfreq_Hz=h.Dfreq_Hz, dv_kms=v(h.dv_kms), dnu_Hz=v(h.dnu_Hz), strength=v(h.strength), a_damp=v(h.a_damp), p_scat=v(h.p_scat), pop=v(h.pop), f_osc=v(h.f_osc))


def read_input(par, line=None):
    """post-processing part of read_input (setup.f90:42-562) for the fields this path uses. Returns (par, line, scatt_mat)."""
    line = line or Line()
    rc, h = _fill_line(line, par.line_id, par.fine_structure, par.include_deuterium, par.D_to_H_ratio)       # setup_resonance_line, line_mod.f90:551-1272
    if rc == 2:                                                           # resonance + fluorescence multiplets, types 4-6 (:713-1153): the table of lart_host_line_lookup
        ml = capi.line_lookup(par.line_id)
        line.line_type, line.wavelength0, line.damping, line.cross0 = ml.line_type, ml.wavelength0, ml.damping, ml.cross0
        line.vtherm1, line.g_recoil0, line.DnuHK_Hz, line.multiplet = ml.vtherm1, ml.g_recoil0, 0.0, ml
        line.E1, line.E2, line.E3 = 1.0, 0.0, 1.0
        ml.HeI_coherent = 1 if (par.HeI_coherent and ml.line_type == 6) else 0
        par.fine_structure = par.include_deuterium = False
    elif rc != 0:
        raise LartInputError(_HL().lart_host_setup_last_error().decode())
    else:
        par.fine_structure, par.include_deuterium, par.line_id = bool(h.fine_structure), bool(h.include_deuterium), h.line_id.decode()
    if par.no_photons >= 1:
        par.nphotons = int(par.no_photons)
    for n in ("geometry", "source_geometry", "velocity_type", "spectral_type"):
        setattr(par, n, getattr(par, n).lower())
    if par.geometry == "":
        par.geometry = "sphere"
    elif par.geometry == "box":
        par.geometry = "rectangle"
    if par.geometry not in ("sphere", "rectangle"):
        raise LartInputError("par%%geometry = '%s' is not supported by this path" % par.geometry)
    par.source_geometry = {"sphere": "uniform_sphere", "cylinder": "uniform_cylinder"}.get(par.source_geometry, par.source_geometry)
    if par.source_geometry not in _SOURCE_GEOMETRIES or (par.source_geometry not in ("point", "uniform_sphere") and (par.use_amr_grid or par.use_clump_medium)):
        raise LartInputError("par%%source_geometry = '%s' is not supported by this path" % par.source_geometry)
    if par.spectral_type not in ("voigt", "monochromatic", "continuum", "gaussian"):
        raise LartInputError("par%%spectral_type = '%s' is not supported by this path" % par.spectral_type)
    if par.velocity_type in ("rotating_solid_body", "rotating_galaxy_halo"):
        pass                                                              # (clump medium: per-clump fields of assign_clump_velocities_from_type)
    elif par.velocity_type in ("power_law", "linear_decelerate", "ssh") and not par.use_clump_medium:
        pass
    elif par.velocity_type not in ("", "hubble", "constant_radial", "parallel_velocity"):
        raise LartInputError("par%%velocity_type = '%s' is not supported by this path" % par.velocity_type)
    if par.spectral_type == "continuum":
        par.comoving_source = False                                       # setup.f90:115
    line.bturb = par.bturb
    if par.temperature <= 0.0 and par.bturb <= 0.0:
        raise LartInputError("ERROR: par%temperature must be > 0 K (or set par%bturb > 0)")
    r = hostlib.HostRules(sphere=int(par.geometry == "sphere"), distance_unit=hostlib.DISTANCE_UNITS.get(par.distance_unit, 4), intensity_unit=0)
    rule_fields = ("nx", "ny", "nz", "nr", "nside", "nxim", "nyim", "nmu", "geometry_JPa", "xyz_symmetry", "xy_symmetry", "xy_periodic", "use_clump_medium", "use_amr_grid",
                   "use_stokes", "save_all", "save_peeloff", "save_peeloff_2D", "save_peeloff_3D", "save_Jabs", "save_Jmu", "core_skip", "core_skip_global", "tau0", "taumax",
                   "N_HI", "N_HImax", "N_HIhomo", "N_gasmax", "N_gashomo", "cext_dust", "DGR", "mu_min", "dmu", "rmax", "xmax", "ymax", "zmax", "source_rmax", "distance2cm")
    for n in rule_fields:
        setattr(r, n, getattr(par, n))
    _chk(_HL().lart_host_input_rules(C.byref(r)))                        # the rules between the namelist fields (setup.f90:160-500), once for both hosts
    for n in rule_fields:
        setattr(par, n, type(getattr(par, n))(getattr(r, n)))
    if r.notes & 1:
        print("---> peelingoff is not allowed for a geometry with the xyz_symmetry.\n---> setting par%save_peeloff = .false.")
    if line.line_type == 8:                                               # setup.f90:239-300: vetoes, no core-skip, band-1 dust re-defaults, R_Ha
        why = ("clump medium" if par.use_clump_medium else "inside-observer (HEALPix)" if par.nside > 0 else "Stokes polarization" if par.use_stokes else
               "fine structure" if par.fine_structure else "xyz/xy symmetry, xy_periodic and shear" if (par.xyz_symmetry or par.xy_symmetry or par.xy_periodic or par.Omega != 0.0) else None)
        if why:
            raise LartInputError("ly_beta: %s not yet supported" % why)
        if par.core_skip or par.core_skip_global:
            print("NOTE: ly_beta forces core_skip = .false. (core-skip would bias the conversion rate).")
        par.core_skip = par.core_skip_global = False
        # (:282-291 compares with the double-precision literals 0.3253_wp, 0.6761_wp, 1.6059e-21_wp while the defaults are single-precision literals
        #  widened to double, define.f90:454-456: the re-default only happens when the input sets those double values itself)
        if par.albedo == 0.3253 and par.hgg == 0.6761 and par.cext_dust == 1.6059e-21:
            par.albedo, par.hgg, par.cext_dust = 0.2804, 0.6483, 2.047e-21
            print("NOTE: ly_beta re-defaults band-1 dust constants to Draine 1026A values (cext=2.047e-21, albedo=0.2804, hgg=0.6483).")
        par.R_Ha = 0.0 if (par.cext_dust_Ha <= 0.0 or par.cext_dust <= 0.0) else par.cext_dust_Ha / par.cext_dust
    par.h2_model = par.h2_model.lower()
    if par.h2_model != "none":                                            # setup.f90:302-365
        why = ("only supported for line_id = 'ly_alpha' (line_type 1)" if line.line_type != 1 else "Phase 1 supports only h2_model='neufeld'; got " + par.h2_model if par.h2_model != "neufeld" else
               "clump medium not yet supported" if par.use_clump_medium else "inside-observer (HEALPix) not yet supported" if par.nside > 0 else
               "Stokes polarization not yet supported" if par.use_stokes else "xyz/xy symmetry and shear not yet supported" if (par.xyz_symmetry or par.xy_symmetry or par.Omega != 0.0) else
               "xy_periodic supported only for the nx=ny=1 slab" if (par.xy_periodic and (par.nx > 1 or par.ny > 1)) else "par%f_H2 must be > 0 when h2_model /= 'none'" if not par.f_H2 > 0.0 else None)
        if why:
            raise LartInputError("H2: " + why)
        line.bturb = par.bturb
        line.h2 = h2_init(par, line)
    if par.use_amr_grid:
        if par.amr_type.lower() != "generic":
            raise LartInputError("par%%amr_type = '%s' is not supported by this path (generic text files only)" % par.amr_type)
        if not par.amr_file:
            raise LartInputError("par%use_amr_grid = .true. needs par%amr_file")
    if par.xy_symmetry:                                                   # mirrors in x and y, the full z range (grid_mod_car.f90:113-134)
        if par.xyz_symmetry or par.xy_periodic or par.use_clump_medium:
            raise LartInputError("par%xy_symmetry excludes xyz_symmetry / xy_periodic / clumps")
        if par.source_geometry != "point" or par.xs_point != 0.0 or par.ys_point != 0.0:
            raise LartInputError("par%xy_symmetry needs a point source on the z axis (sources are folded for xyz_symmetry only, generate_photon.f90:352-356)")
        if par.calc_J or par.calc_P or par.calc_Pnew:
            raise LartInputError("par%geometry_JPa = 2 (cylindrical profiles, the xy_symmetry default) is not supported by this path")
    if par.geometry_JPa in (0, 2) and (par.calc_J or par.calc_P or par.calc_Pnew):
        raise LartInputError("par%geometry_JPa = 2 (cylindrical profiles) is not supported by this path")
    scatt_mat = ScattMat()
    if par.scatt_mat_file and par.DGR > 0.0:                              # :496-500
        scatt_mat = setup_scattering_matrix(par, par.scatt_mat_file)
    elif par.use_stokes:
        par.DGR = 0.0
    if par.DGR == 0.0:
        par.save_Jabs = False
    return par, line, scatt_mat


@dataclass
class Grid:
    """grid_type (define.f90:117-206): geometry scalars + read-only tables + frequency axis"""
    nx: int = 0
    ny: int = 0
    nz: int = 0
    i0: int = 0
    j0: int = 0
    k0: int = 0
    xmin: float = 0.0
    ymin: float = 0.0
    zmin: float = 0.0
    xmax: float = 0.0
    ymax: float = 0.0
    zmax: float = 0.0
    dx: float = 0.0
    dy: float = 0.0
    dz: float = 0.0
    rmin: float = 0.0
    rmax: float = 0.0
    xface: np.ndarray = None
    yface: np.ndarray = None
    zface: np.ndarray = None
    rhokap: np.ndarray = None
    voigt_a: np.ndarray = None
    Dfreq: np.ndarray = None
    vfx: np.ndarray = None
    vfy: np.ndarray = None
    vfz: np.ndarray = None
    rhokapD: np.ndarray = None
    Dfreq_ref: float = 0.0
    voigt_amean: float = 0.0
    nxfreq: int = 0
    xfreq_min: float = 0.0
    xfreq_max: float = 0.0
    dxfreq: float = 0.0
    dwave: float = 0.0
    xfreq: np.ndarray = None
    velocity: np.ndarray = None
    wavelength: np.ndarray = None
    xcrit: float = 0.0
    xcrit2: float = 0.0
    taupole: float = 0.0
    tauhomo: float = 0.0





def voigt_center(a):
    """H(0,a) from the first Harris terms on the x=0 table node: voigt_mod.f90:691-700 with v = 0"""
    return 1.0 + a * (-1.1283791671 + a * 1.0)


def grid_create(par, line):
    """grid_mod_car.f90:73-1238, analytic-medium subset (no input files): lart_host_car_medium builds the axes, the density mask of the shell / bicone, the
    opacity normalisation (taumax | tauhomo | N_gasmax | N_gashomo) and the velocity field, car_setup_freq_grid the spectral axis and the core-skip scale"""
    g = Grid(nx=par.nx, ny=par.ny, nz=par.nz, xmax=par.xmax, ymax=par.ymax, zmax=par.zmax, rmin=par.rmin, rmax=par.rmax)
    if par.velocity_type not in hostlib.VELOCITY_TYPES:
        raise LartInputError("par%%velocity_type = '%s' is not supported by this path" % par.velocity_type)
    m = hostlib.HostMedium(nx=g.nx, ny=g.ny, nz=g.nz, xyz_symmetry=int(par.xyz_symmetry), xy_symmetry=int(par.xy_symmetry), velocity_type=hostlib.VELOCITY_TYPES[par.velocity_type])
    for n in ("xmax", "ymax", "zmax", "rmin", "rmax", "cone_opening", "taumax", "tauhomo", "N_gasmax", "N_gashomo", "Vexp", "Vx", "Vy", "Vz", "Vrot", "rinner", "velocity_alpha",
              "rpeak", "Vpeak", "DeltaV"):
        setattr(m, n, float(getattr(par, n)))
    m.vtherm, m.wavelength0, m.damping, m.cross0 = vtherm_total(line, par.temperature), line.wavelength0, line.damping, line.cross0
    m.dust_opacity = par.cext_dust * par.DGR
    if not par.use_clump_medium:                                          # :352-363: the clump population carries the physical length unit
        par.distance_unit, par.distance2cm = "", 1.0
    shape = (g.nx, g.ny, g.nz)
    g.xface, g.yface, g.zface = np.zeros(g.nx + 1), np.zeros(g.ny + 1), np.zeros(g.nz + 1)
    for n in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz"):
        setattr(g, n, np.zeros(shape, order="F"))
    g.rhokapD = np.zeros(shape, order="F") if par.DGR > 0.0 else None
    _chk(_HL().lart_host_car_medium(C.byref(m), *[hostlib.p(a) for a in (g.xface, g.yface, g.zface, g.rhokap, g.voigt_a, g.Dfreq, g.vfx, g.vfy, g.vfz, g.rhokapD)]))
    for n in ("dx", "dy", "dz", "xmin", "ymin", "zmin", "i0", "j0", "k0", "Dfreq_ref", "voigt_amean"):
        setattr(g, n, getattr(m, n))
    g.taupole, g.tauhomo = m.taupole_out, m.tauhomo_out
    par.taumax, par.tauhomo, par.N_gasmax, par.N_gashomo = m.taumax, m.tauhomo, m.N_gasmax, m.N_gashomo      # what was not given is the medium's own (:617-737)
    car_setup_freq_grid(par, line, g, core_skip_scale=True)
    return g


@dataclass
class JPa:
    """create_JPa_mem (grid_mod_car.f90:1242-1434): profile bin of every cell + cells per bin"""
    nbins: int = 0
    ind_bin: np.ndarray = None      # (nx,ny,nz) int32, 1-based, 0 = none
    ncount: np.ndarray = None


def create_JPa_mem(par, g):
    if not (par.calc_J or par.calc_P or par.calc_Pnew):
        return JPa()
    ind = np.zeros((g.nx, g.ny, g.nz), dtype=np.int32, order="F"); ncount = np.zeros(g.nx * g.ny * g.nz); nb = C.c_int32()
    _chk(_HL().lart_host_jpa_car(int(par.geometry_JPa), int(par.xyz_symmetry), g.nx, g.ny, g.nz, hostlib.p(g.xface), hostlib.p(g.yface), hostlib.p(g.zface),
                                 g.xmax, g.ymax, g.zmax, hostlib.p(g.rhokap), C.byref(nb), hostlib.p(ind), hostlib.p(ncount)))
    return JPa(nb.value, ind, ncount[:nb.value].copy())


def car_setup_freq_grid(par, line, g, core_skip_scale=False):
    """grid_mod_car.f90:1442-1548 through lart_host_freq_grid: the spectral axis from the wavelength / velocity window or from (a tau)^(1/3), the band-2 axis of
    ly_beta; core_skip_scale: also grid%xcrit of the Cartesian grid (:1186-1219; the octree sets its own)"""
    vth = vtherm_total(line, par.temperature)
    f = hostlib.HostFreq(nxfreq=par.nxfreq, nwavelength=par.nwavelength, nvelocity=par.nvelocity, continuum=int(par.spectral_type == "continuum"), line_type=line.line_type,
                         nxfreq_Ha=par.nxfreq_Ha, core_skip_global=int(par.core_skip_global))
    for n in ("wavelength_min", "wavelength_max", "velocity_min", "velocity_max", "xfreq_min", "xfreq_max", "tauhomo", "taumax", "Vexp", "xfreq_max_Ha"):
        setattr(f, n, float(getattr(par, n)))
    f.vtherm, f.wavelength0, f.wavelength0_Ha, f.Dfreq_ref, f.DnuHK_Hz, f.voigt_amean = vth, line.wavelength0, line.wavelength0_Ha, g.Dfreq_ref, line.DnuHK_Hz, g.voigt_amean
    f.cells_per_radius = g.xmax / g.dx if (core_skip_scale and g.dx > 0.0) else 1.0
    _chk(_HL().lart_host_freq_grid(C.byref(f)))
    par.atau3, par.nxfreq, par.nwavelength, par.nvelocity, par.xfreq_min, par.xfreq_max = f.atau3, f.nxfreq, f.nwavelength, f.nvelocity, f.xfreq_min, f.xfreq_max
    g.nxfreq, g.dxfreq, g.xfreq_max, g.xfreq_min, g.dwave = f.nxfreq, f.dxfreq, f.xfreq_max, f.xfreq_min, f.dwave
    if line.line_type == 8:
        g.nxfreq_Ha = par.nxfreq_Ha = f.nxfreq_Ha
        g.xfreq_min_Ha, g.xfreq_max_Ha, g.dxfreq_Ha, g.dwave_Ha = f.xfreq_min_Ha, f.xfreq_max_Ha, f.dxfreq_Ha, f.dwave_Ha
    if core_skip_scale:
        g.xcrit, g.xcrit2 = f.xcrit, f.xcrit2
    g.xfreq, g.velocity, g.wavelength = np.zeros(g.nxfreq), np.zeros(g.nxfreq), np.zeros(g.nxfreq)
    _chk(_HL().lart_host_freq_axis(g.nxfreq, g.xfreq_min, g.dxfreq, vth, line.wavelength0, hostlib.p(g.xfreq), hostlib.p(g.velocity), hostlib.p(g.wavelength)))


@dataclass
class Observers:
    nobs: int = 0
    xyz: np.ndarray = None
    rmatrix: np.ndarray = None      # (nobs, 3, 3) row r, column c = rmatrix(r+1, c+1)
    dxim: float = 0.0
    dyim: float = 0.0
    nxim: int = 0
    nyim: int = 0
    steradian_pix: float = 0.0


def observer_create(par):
    """observer_create_outside (observer_rect.f90:24-326, rotation centre at the origin) / observer_create_inside (observer_heal.f90:24-80) through
    lart_host_observers; par%alpha/beta/gamma, par%obsx/y/z, par%distance, par%dxim/dyim come back completed, as the reference leaves them"""
    if not par.save_peeloff:
        return Observers()
    o = hostlib.HostObs(nside=par.nside, sphere=int(par.geometry == "sphere"), nxim=par.nxim, nyim=par.nyim)
    for n in ("xmax", "ymax", "zmax", "rmax", "distance", "dxim", "dyim"):
        setattr(o, n, float(getattr(par, n)))
    arr = {n: hostlib.nan_array(getattr(par, n)) for n in ("alpha", "beta", "gamma", "obsx", "obsy", "obsz")}
    xyz, rm = np.zeros(3 * MAX_OBSERVERS), np.zeros(9 * MAX_OBSERVERS)
    _chk(_HL().lart_host_observers(C.byref(o), *[hostlib.p(arr[n]) for n in ("alpha", "beta", "gamma", "obsx", "obsy", "obsz")], hostlib.p(xyz), hostlib.p(rm)))
    for n, a in arr.items():
        getattr(par, n)[:] = [float(v) for v in a]
    par.nobs, n = o.nobs, o.nobs
    if par.nside > 0:
        par.npix = o.npix
        return Observers(nobs=n, xyz=xyz[:3 * n].reshape(n, 3).copy(), nxim=o.nxim, nyim=1, steradian_pix=o.steradian_pix)
    par.distance, par.dxim, par.dyim = o.distance, o.dxim, o.dyim
    return Observers(nobs=n, xyz=xyz[:3 * n].reshape(n, 3).copy(), rmatrix=rm[:9 * n].reshape(n, 3, 3).copy(), dxim=o.dxim, dyim=o.dyim, nxim=par.nxim, nyim=par.nyim,
                     steradian_pix=o.steradian_pix)


def make_lart_params(par, line, g):
    p = capi.LartParams()
    for n in ("xyz_symmetry", "xy_periodic", "xy_symmetry", "comoving_source", "recoil", "core_skip", "core_skip_global", "use_stokes",
              "use_reduced_wgt", "save_Jin", "save_Jabs", "save_Jmu", "nmu", "save_peeloff", "save_peeloff_2D", "save_peeloff_3D",
              "save_direc0", "save_dust_scattered"):
        setattr(p, n, int(getattr(par, n)))
    p.source_geometry = _SOURCE_GEOMETRIES[par.source_geometry]
    p.xrange, p.yrange, p.zrange, p.source_zscale = g.xmax - g.xmin, g.ymax - g.ymin, g.zmax - g.zmin, par.source_zscale
    p.spectral_type = {"monochromatic": 0, "voigt": 1, "continuum": 2, "gaussian": 3}[par.spectral_type]
    gsig = par.gaussian_FWHM_vel / 2.3548200450309493 if par.gaussian_FWHM_vel > 0.0 else par.gaussian_sigma_vel
    p.xfreq_max, p.source_sigma_x = g.xfreq_max, gsig / vtherm_total(line, par.temperature)
    p.nxfreq = g.nxfreq
    for n in ("xs_point", "ys_point", "zs_point", "source_rmax", "xfreq0", "DGR", "hgg", "albedo", "mu_min", "dmu"):
        setattr(p, n, float(getattr(par, n)))
    p.xfreq_min, p.dxfreq, p.Dfreq_ref = g.xfreq_min, g.dxfreq, g.Dfreq_ref
    p.g_recoil0, p.E1, p.E2, p.E3 = line.g_recoil0, line.E1, line.E2, line.E3
    p.line_type, p.DnuHK_Hz = line.line_type, line.DnuHK_Hz
    p.delta_nu_HD_Hz, p.ratio_Dfreq_HD, p.ratio_voigta_HD, p.nD_over_nH, p.g_recoil0_D = (line.delta_nu_HD_Hz, line.ratio_Dfreq_HD, line.ratio_voigta_HD,
                                                                                           line.nD_over_nH, line.g_recoil0_D)
    p.xcrit, p.xcrit2 = g.xcrit, g.xcrit2
    p.Omega = float(par.Omega)                 # raytrace_to_tau_car_xyper_shear when xy_periodic and nx, ny > 1 (setup.f90:967-969)
    if line.line_type == 8:
        p.nxfreq_Ha, p.xfreq_min_Ha, p.dxfreq_Ha = int(g.nxfreq_Ha), float(g.xfreq_min_Ha), float(g.dxfreq_Ha)
        p.R_Ha, p.albedo_Ha, p.hgg_Ha, p.P_conv = float(par.R_Ha), float(par.albedo_Ha), float(par.hgg_Ha), float(line.P_conv)
        for i in range(3):
            p.E_conv[i] = float(line.E_conv[i])
    if line.h2 is not None:
        h = line.h2
        p.h2_nlines, p.h2_hi_width, p.h2_Dfreq_Hz = h.n, int(par.h2_hi_width), h.Dfreq_Hz
        for i in range(h.n):
            p.h2_dnu_Hz[i], p.h2_strength[i], p.h2_a_damp[i], p.h2_p_scat[i] = h.dnu_Hz[i], h.strength[i], h.a_damp[i], h.p_scat[i]
    return p


def job_seed(iseed, dist=None):
    """(Philox key of the job, seed of the host-side MT stream).  A non-zero par%iseed is both.  par%iseed = 0: one value from the operating system
    (lart_host_entropy_seed), drawn by rank 0 and received by the others when `dist` is an initialised torch.distributed; the MT seed is its residue
    mod 2^31 - 1, never 0 (main.f90 of the Fortran host does the same)"""
    if iseed != 0:
        return int(iseed), int(iseed)
    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    box = [capi.entropy_seed() if (not multi or dist.get_rank() == 0) else None]
    if multi:
        dist.broadcast_object_list(box, src=0)
    run_seed = int(box[0])
    print("par%%iseed = 0: seeded from the operating system, seed = %d" % run_seed)
    return run_seed, (run_seed % 2147483647 or 1)


class Simulation:
    """main.f90:19-69 sequence: read_input -> setup -> grid_create -> observer_create -> run_simulation ->
    output_reduce -> output_normalize, with run_simulation/output_reduce crossing the C-ABI."""

    def __init__(self, par, device=0, dist=None):
        if isinstance(par, str):
            par = read_namelist(par)
        elif isinstance(par, dict):
            d = par; par = Params()
            for k, v in d.items():
                set_param(par, k, v)
        self.par, self.line, self.scatt_mat = read_input(par)
        self.amr = None; self.clumps = None
        # par%iseed = 0: seed from the operating system (init_random_mt_seed, random_mt.f90:889-947), ONE value for the whole job: with several ranks (dist = an
        # initialised torch.distributed) rank 0 draws it and the others receive it before anything random is generated -- the clump population has to be the same
        # on every rank and the photon ids of the ranks interleave on one Philox key.  As main.f90 of the Fortran host: the 53-bit value is the Philox key, its
        # residue mod 2^31 - 1 (never 0) the seed of the host-side MT stream that places the clumps.
        self.run_seed, self.par.iseed = job_seed(self.par.iseed, dist)
        if self.par.use_clump_medium:                                     # grid_create_clump (main.f90:25-27)
            from . import clump as clump_mod
            self.clumps, self.grid = clump_mod.grid_create_clump(self.par, self.line)
        elif self.par.use_amr_grid:                                       # grid_create_amr (main.f90:31-35)
            from . import amr as amr_mod
            self.amr, self.grid = amr_mod.grid_create_amr(self.par, self.line, self.par.amr_file)
        else:
            self.grid = grid_create(self.par, self.line)
        self.obs = observer_create(self.par)
        self.cpar = make_lart_params(self.par, self.line, self.grid)
        self.ctx = capi.Context(self.cpar, device)
        if self.line.multiplet is not None:
            self.ctx.set_line_multiplet(self.line.multiplet)
        g = self.grid
        if self.clumps is not None:
            c = self.clumps
            self.ctx.set_clumps(c.x, c.y, c.z, c.vx, c.vy, c.vz, c.radius, c.rhokap, c.rhokapD, c.voigt_a, c.Dfreq,
                                c.sphere_R, c.Dfreq_ref, c.voigt_a_ref, c.cgx, c.cg_min, c.cg_d, c.cg_start, c.cg_list)
            if c.has_overlap:                                             # setup_clump_overlap: setup.f90:1053-1082
                self.ctx.set_clump_overlap(True)
        elif self.amr is not None:
            a = self.amr
            self.ctx.set_grid_amr(a.xmin, a.xmax, a.cx, a.cy, a.cz, a.ch, a.children, a.neighbor, a.ileaf, a.icell_of_leaf,
                                  a.rhokap, a.voigt_a, a.Dfreq, a.vfx, a.vfy, a.vfz, a.rhokapD)
        else:
            self.ctx.set_grid_car(g.nx, g.ny, g.nz, g.xmin, g.ymin, g.zmin, g.dx, g.dy, g.dz, g.i0, g.j0, g.k0,
                                  g.xface, g.yface, g.zface, g.rhokap, g.voigt_a, g.Dfreq, g.vfx, g.vfy, g.vfz, g.rhokapD)
        if self.scatt_mat.nPDF:
            s = self.scatt_mat
            self.ctx.set_mueller(s.coss, s.S11, s.S12, s.S33, s.S34, s.phase_PDF, s.alias)
        if self.obs.nobs:
            o = self.obs
            if self.par.nside > 0:
                self.ctx.set_observers_heal(o.xyz, self.par.nside)
            else:
                self.ctx.set_observers(o.xyz, o.rmatrix, o.dxim, o.dyim, o.nxim, o.nyim)
        if self.clumps is not None:
            if self.par.calc_J or self.par.calc_P or self.par.calc_Pnew:
                raise LartInputError("CALCJ/CALCP profile tallies are not defined for the clump medium")
            self.jpa = JPa(); self.nphotons_run = 0
            return
        if self.amr is not None:
            amr_mod.create_JPa_amr_mem(self.par, self.amr)
            self.jpa = JPa(nbins=self.amr.nbins)
            if self.jpa.nbins:
                dbin = self.amr.dr_JPa if self.par.geometry_JPa == 1 else self.amr.dz_JPa
                self.ctx.set_JPa_amr(self.par.calc_J, self.par.calc_P, self.par.calc_Pnew, self.par.geometry_JPa, self.jpa.nbins, dbin, self.line.cross0)
            self.nphotons_run = 0
            return
        self.jpa = create_JPa_mem(self.par, g)
        if self.jpa.nbins:
            self.ctx.set_JPa(self.par.calc_J, self.par.calc_P, self.par.calc_Pnew, self.jpa.nbins, self.jpa.ind_bin, self.line.cross0)
        self.nphotons_run = 0

    def seed(self):
        """the Philox key of this run: par%iseed as given, or the value __init__ drew (and shared between the ranks) for par%iseed = 0"""
        return self.run_seed

    def run_simulation(self, nphotons=None, rank=0, nproc=1, seed=None):
        """photon ids rank+1, rank+1+nproc, ... <= nphotons (run_equal_number, run_simulation_mod.f90:150-151).
        The device tallies are cumulative over calls (lart_gpu_zero_tallies resets them); output_normalize divides by the
        photons the device has finished, so a second call adds statistics instead of skewing the normalisation."""
        n = self.par.nphotons if nphotons is None else int(nphotons)
        count = photons_of_rank(n, rank, nproc)
        if self.par.save_all_photons:
            self.ctx.enable_all_photons(True)
        self.ctx.run(count, first_id=rank + 1, stride=nproc, seed=self.seed() if seed is None else seed)
        self.nphotons_run = n
        return count

    def make_sightline_tau(self):
        """make_sightline_tau (main.f90:36, par%save_sightline_tau): tau_gas(nu), N(HI) and tau_dust maps of every observer"""
        if not self.obs.nobs:
            raise LartInputError("make_sightline_tau needs observers (par%save_peeloff)")
        return self.ctx.sightline_tau(self.line.cross0)

    def output_reduce(self, dist=None, force=False, root=0):
        """output_reduce (main.f90:46): the fused tally buffer summed onto the root rank (reduce_mem = MPI_REDUCE to rank 0,
        memory_mod_mpi.f90:366-394) by the library's own RCCL reduce (lart_gpu_comm_init + lart_gpu_reduce_tallies, one
        ncclReduce(sum, f64) over xGMI).  torch.distributed only carries the 128-byte communicator id from rank 0 to the others."""
        if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
            return
        world, rank = dist.get_world_size(), dist.get_rank()
        if self.ctx.comm_info()[0] != world:
            box = [capi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            self.ctx.comm_init(box[0], world, rank)
        self.ctx.reduce_tallies(root)

    def output_normalize(self, t=None, nphotons=None):
        """output_normalize_outside (output_sum_rect.f90:151-487): physical units for Jout/Jin/Jabs/Jmu, the peel cubes and images, the J / P profiles, and for
        ly_beta the band-2 sections (Jout_Ha, Jabs_Ha by the H-alpha bin unit :210-236, peel_Ha :433-435, Pconv like P1 :345-372) and the weight budget per photon"""
        t = t or self.ctx.tallies()
        par, g = self.par, self.grid
        nph = float(nphotons or t["nphotons_done"] or self.nphotons_run)     # photons the device(s) finished (cumulative, summed by output_reduce)
        out = dict(t)
        phys = par.distance2cm != 1.0                                     # par%intensity_unit default, setup.f90:491-497: per Angstrom when a length unit is set
        s = hostlib.HostNorm(xy_periodic=int(par.xy_periodic), sphere=int(par.geometry == "sphere"), nphotons=nph, bin_unit=g.dwave if phys else g.dxfreq,
                             distance2cm=par.distance2cm, rmax=g.rmax, xmax=g.xmax, ymax=g.ymax, zmax=g.zmax, amr_L_box=self.amr.L_box if self.amr is not None else 0.0,
                             steradian_pix=self.obs.steradian_pix if self.obs.nobs else 0.0)
        s.bin_unit_Ha = (g.dwave_Ha if phys else g.dxfreq_Ha) if self.line.line_type == 8 else s.bin_unit
        _chk(_HL().lart_host_norm_scales(C.byref(s)))
        unit = s.bin_unit
        for k in ("Jout", "Jin", "Jabs"):
            out[k] = t[k] / s.scale_J
        if "Jmu" in t:
            out["Jmu"] = t["Jmu"] * par.nmu / s.scale_J                  # :188-190, :206-208
        for k in ("Jout_Ha", "Jabs_Ha"):
            if k in t:
                out[k] = t[k] / s.scale_J_Ha
        if self.obs.nobs:
            for k in ("scatt", "direc", "I", "Q", "U", "V", "direc0", "scatt_dust"):
                if k in t:
                    out[k] = t[k] / s.scale_cube                          # :428-430
                if k + "_2D" in t:
                    out[k + "_2D"] = t[k + "_2D"] / s.scale_image
            if "peel_Ha" in t:
                out["peel_Ha"] = t["peel_Ha"] / s.scale_cube_Ha
        for k in ("W_conv", "W_esc1", "W_abs1", "W_esc2", "W_abs2", "W_H2abs", "W_H2scat"):     # :455-465: the budget is printed per photon
            if k in t:
                out[k] = t[k] / nph
        if self.jpa.nbins and self.amr is not None:                       # octree: volume-weighted, output_sum_heal.f90:476-590
            vol = np.where(self.amr.vol > 0, self.amr.vol, np.inf) * par.distance2cm ** 2
            if par.xy_periodic and par.geometry_JPa in (3, -1):           # slab, one photon per cm^2: the slab area replaces 1/d^2 (:874-1000)
                vol = np.where(self.amr.vol > 0, self.amr.vol, np.inf) / self.amr.L_box ** 2
            if "J1" in t:
                out["J1"] = t["J1"] / (FOURPI * vol[None, :] * nph * unit)
            for k in ("P1", "P1_new", "Pconv"):
                if k in t:
                    out[k] = t[k] / (vol * nph)
        elif self.jpa.nbins:                                              # :275-372 J(x), P_alpha and conversion-rate profiles
            dVol = g.dx * g.dy * g.dz * par.distance2cm ** 2
            cnt = np.where(self.jpa.ncount > 0, self.jpa.ncount, np.inf)
            geo_area = (g.xmax - g.xmin) * (g.ymax - g.ymin) * par.distance2cm ** 2 if par.xy_periodic else 1.0
            f = geo_area if (par.xy_periodic and par.geometry_JPa in (3, -1)) else 1.0
            if "J1" in t:
                out["J1"] = t["J1"] / cnt[None, :] * (f / (FOURPI * dVol * nph * unit))
            for k in ("P1", "P1_new", "Pconv"):
                if k in t:
                    out[k] = t[k] / cnt * (f / (dVol * nph))
        out["nscatt_tot"] = (t["nscatt_gas"] + t["nscatt_dust"]) / nph
        return out


def photons_of_rank(nphotons, rank, nproc):
    """how many of the ids 1..nphotons fall on `rank` under `do ip = rank+1, nphotons, nproc` (run_simulation_mod.f90:150-151)"""
    return (nphotons - rank - 1) // nproc + 1 if nphotons > rank else 0


def reduce_tally_buffer(t, dist, root=0):
    """caller-owned variant of the exchange step on a torch tensor (lart_gpu_tally_buffer wrapped by _as_cuda_tensor, or a CPU
    copy in the gloo tests): element-wise SUM onto the root rank, as reduce_mem does (MPI_REDUCE, memory_mod_mpi.f90:377-393)"""
    dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
    return t


def _as_cuda_tensor(ptr, n, keep):
    """wrap the device tally buffer (owned by liblart_gpu) as a torch f64 tensor for torch.distributed"""
    import torch

    class _Wrap:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3, "strides": None}
        _keep = keep
    return torch.as_tensor(_Wrap(), device="cuda")
