"""Python mirror of the reference's host-side interface around the photon path.

Same names, argument meaning and error behaviour as the Fortran host of the reference:
    read_input        setup.f90:4-579            namelist &parameters par%... / + post-processing
    setup_resonance_line  line_mod.f90:551-1272  (Ly-alpha branch)
    grid_create       grid_mod_car.f90:11-1238   (analytic uniform sphere / slab / box subset)
    observer_create   observer_rect.f90:24-326
    run_simulation    main.f90:42                -> lart_gpu_run through the C-ABI (lart_amd.capi)
    output_reduce     main.f90:46                -> RCCL sum of the fused device tally buffer
    output_normalize  output_sum_rect.f90:151-487 (Jout/Jin/Jabs/Jmu + peel cubes)
This module is host plumbing (numpy); all transport arithmetic runs in liblart_gpu.so.  The Fortran host
with the same sequence lives in lart_amd/fhost/*.f90.
"""
import math
import os
import re
from dataclasses import dataclass, field
import numpy as np

from . import capi

PI = 3.141592653589793238462643383279502884197
TWOPI, FOURPI = 2 * PI, 4 * PI
UM2KM, UM2M = 1.0e-9, 1.0e-6
H_PLANCK = 6.62607004e-34
SPEEDC = 2.99792458e5
DEG2RAD, RAD2DEG = PI / 180.0, 180.0 / PI
NAN = float("nan")
MAX_OBSERVERS = capi_max = 16


class LartInputError(ValueError):
    """raised where the reference prints a message and aborts (setup.f90:132-139, :244-280)"""


def _f32(x):
    return float(np.float32(x))


@dataclass
class Params:
    """subset of params_type (define.f90:209-544) that the Cartesian hot path reads; defaults as there.
    hgg / albedo / cext_dust defaults are float32 literals widened to f64 (define.f90:454-456)."""
    no_photons: float = 1e5
    iseed: int = 0
    temperature: float = 1e4
    taumax: float = -999.0
    tauhomo: float = -999.0
    tau0: float = -999.0
    N_HImax: float = -999.0
    N_HIhomo: float = -999.0
    N_HI: float = -999.0
    Vexp: float = 0.0
    Vx: float = 0.0
    Vy: float = 0.0
    Vz: float = 0.0
    comoving_source: bool = True
    recoil: bool = False
    core_skip: bool = False
    core_skip_global: bool = False
    xyz_symmetry: bool = False
    xy_symmetry: bool = False
    fine_structure: bool = False
    include_deuterium: bool = False
    D_to_H_ratio: float = 1.5e-5
    line_id: str = "ly_alpha"
    xy_periodic: bool = False
    geometry: str = ""
    velocity_type: str = ""
    nx: int = 1
    ny: int = 1
    nz: int = 11
    nr: int = -999
    xmax: float = 1.0
    ymax: float = 1.0
    zmax: float = 1.0
    rmin: float = -999.0
    rmax: float = -999.0
    source_rmax: float = -999.0
    xs_point: float = 0.0
    ys_point: float = 0.0
    zs_point: float = 0.0
    xfreq0: float = 0.0
    xfreq_min: float = NAN
    xfreq_max: float = NAN
    nxfreq: int = 121
    velocity_min: float = NAN
    save_all_photons: bool = False  # the per-photon table of write_output_allph (Context.all_photons)
    HeI_coherent: bool = False      # line type 6: frequency-dependent E1..E3 (compute_HeI_E_coherent, line_mod.f90:8-34)
    bturb: float = -999.0           # turbulent b parameter [km/s], in quadrature with the thermal width (define.f90:928-933)
    wavelength_min: float = NAN     # angstrom; takes precedence over the velocity window (grid_mod_car.f90:1457-1461)
    wavelength_max: float = NAN
    nwavelength: int = 0
    velocity_max: float = NAN
    nvelocity: int = 0
    source_geometry: str = "point"
    spectral_type: str = "voigt"
    distance2cm: float = -999.9
    distance_unit: str = ""
    out_file: str = ""
    out_merge: bool = False
    file_format: str = "hdf5"
    use_master_slave: bool = True
    num_send_at_once: int = 100
    use_reduced_wgt: bool = False
    use_stokes: bool = False
    save_all: bool = False
    save_Jin: bool = True
    save_Jabs: bool = True
    save_Jmu: bool = False
    nmu: int = 11
    save_peeloff: bool = False
    save_peeloff_2D: bool = False
    save_peeloff_3D: bool = True
    save_sightline_tau: bool = False
    save_direc0: bool = False
    save_dust_scattered: bool = False
    save_radial_profile: bool = False
    hgg: float = _f32(0.6761)
    albedo: float = _f32(0.3253)
    cext_dust: float = _f32(1.6059e-21)
    DGR: float = 0.0
    scatt_mat_file: str = ""
    nprint: int = 10000000
    no_print: float = 0.0
    nobs: int = 1
    nxim: int = 0
    nyim: int = 0
    distance: float = NAN
    alpha: list = field(default_factory=lambda: [NAN] * MAX_OBSERVERS)
    beta: list = field(default_factory=lambda: [NAN] * MAX_OBSERVERS)
    gamma: list = field(default_factory=lambda: [NAN] * MAX_OBSERVERS)
    obsx: list = field(default_factory=lambda: [NAN] * MAX_OBSERVERS)
    obsy: list = field(default_factory=lambda: [NAN] * MAX_OBSERVERS)
    obsz: list = field(default_factory=lambda: [NAN] * MAX_OBSERVERS)
    dxim: float = NAN
    dyim: float = NAN
    nside: int = 0
    npix: int = 0
    # the reference's compile-time tally switches (Makefile:33-61), run-time here
    calc_J: bool = False
    calc_P: bool = False
    calc_Pnew: bool = False
    geometry_JPa: int = 2147483647
    # clump medium (define.f90:325-352; uniform clump properties, no overlap, no radial profiles / file input)
    use_clump_medium: bool = False
    clump_radius: float = -1.0
    clump_N_clumps: float = -1.0
    clump_f_vol: float = -1.0
    clump_f_cov: float = -1.0
    clump_tau0: float = -1.0
    clump_NHI: float = -1.0
    clump_nH: float = -1.0
    clump_temperature: float = -1.0
    clump_sigma_v: float = 0.0
    clump_fully_inside: bool = True
    clump_allow_overlap: bool = False
    cone_opening: float = 0.0       # bicone half-opening angle about the z axis [deg], 0 = everything
    clump_input_file: str = ""
    save_clump_info: bool = False
    Vrot: float = 0.0
    rinner: float = 0.0
    Vpeak: float = 0.0
    rpeak: float = 0.0
    DeltaV: float = 0.0
    velocity_alpha: float = 1.0
    gaussian_sigma_vel: float = 12.843374
    gaussian_FWHM_vel: float = -999.0
    source_zscale: float = 0.0
    # octree medium (grid_mod_amr.f90); only the generic text format of read_generic_amr.f90:112-185 is read here
    use_amr_grid: bool = False
    amr_type: str = "generic"
    amr_file: str = ""
    # derived (not inputs)
    nphotons: int = 100000
    mu_min: float = -1.0
    dmu: float = 0.0
    atau3: float = 0.0
    N_gasmax: float = -999.0
    N_gashomo: float = -999.0
    # shearing box (define.f90:308-312): background y velocity -q Omega x.  Without a dens_file grid_create leaves par%Omega as it is
    # (grid_mod_car.f90:346-351 rescales it only there), i.e. the change of grid%vfy, in thermal-speed units, per wrap around the x boundary
    q: float = 1.0
    Omega: float = 0.0
    # line_id = 'ly_beta' (line type 8): band-2 (H-alpha) spectral axis and dust constants (define.f90:461-468; Draine 2003, R_V = 3.1 at 6562 A)
    nxfreq_Ha: int = 0
    xfreq_max_Ha: float = 0.0
    cext_dust_Ha: float = 3.801e-22
    albedo_Ha: float = 0.6741
    hgg_Ha: float = 0.4967
    R_Ha: float = 0.0
    # H2 lines next to Ly-alpha (define.f90:478-491; h2_mod.f90): 'none' | 'neufeld'
    h2_model: str = "none"
    f_H2: float = 0.0
    h2_temperature: float = 1000.0
    h2_pure_absorption: bool = False
    h2_hi_width: bool = False
    h2_data_dir: str = ""


# fields of params_type that exist in the reference but select physics outside this path: accepted at their
# default, rejected ("not supported") when set to something that would change the path
_UNSUPPORTED_DEFAULTS = {
    "z_symmetry": False,
    "dens_file": "", "temp_file": "", "velo_file": "", "emiss_file": "", "cart_file": "", "input_field": "",
    "star_file": "", "use_cie_condition": False, "density_rscale": -999.9, "density_zscale": -999.9,
    "density_alpha": 0.0, "temperature0": -999.0,
    "line_prof_file": "", "luminosity": 1.0, "save_backup": False, "save_input_grid": False,
    "intensity_unit": -999, "out_bitpix": 0,
    # the rest of the reference's namelist surface (define.f90:209-544): readable, rejected when set
    "Dfreq0": -999.0,
    "voigt_a0": -999.0,
    "source_rscale": 0.0,
    "sersic_m": 1.0,
    "Reff": 0.0,
    "xmin": float("nan"),
    "ymin": float("nan"),
    "zmin": float("nan"),
    "continuum_normalize": True,
    "EW_line": 0.0,
    "f_line": 0.0,
    "clump_radius_profile": "constant",
    "clump_density_profile": "constant",
    "clump_number_profile": "constant",
    "clump_radius_alpha": 0.0,
    "clump_radius_r0": 0.0,
    "clump_density_alpha": 0.0,
    "clump_density_r0": 0.0,
    "clump_number_alpha": 0.0,
    "clump_number_r0": 0.0,
    "clump_radius_min": -1.0,
    "clump_radius_max_in": -1.0,
    "clump_profile_file": "",
    "amr_snapnum": 0,
    "ionization_model": "cie_formula",
    "dust_model": "global_dgr",
    "emissivity_model": "none",
    "ion_model": "none",
    "metallicity_global": -1.0,
    "Z_ref": 0.0134,
    "f_ion_dust": 0.01,
    "base_name": "",
    "sampling_method": 1,
    "f_composite": 0.5,
    "ny_2gam": 101,
    "transport_2gamma": False,
    "W_conv": 0.0,
    "W_esc1": 0.0,
    "W_abs1": 0.0,
    "W_esc2": 0.0,
    "W_abs2": 0.0,
    "W_H2abs": 0.0,
    "W_H2scat": 0.0,
    "line_prof_file_type": 0,
    "reduce_factor": 1,
    "centering": 0,
    "nscatt_gas": 0.0,
    "nscatt_dust": 0.0,
    "nscatt_tot": 0.0,
    "exetime": 0.0,
    "nrejected": 0.0,
    "acceptance_rate": 1.0,
    "stellar_limb_darkening": 2,
    "distance_star_to_planet": 0.0,
    "stellar_radius": 0.0,
    "flux_factor": 0.0,
    "inclination_angle": float("nan"),
    "position_angle": float("nan"),
    "phase_angle": float("nan"),
    "rotation_center_x": float("nan"),
    "rotation_center_y": float("nan"),
    "rotation_center_z": float("nan"),
    "observer_located_inside": False,
}


def _parse_value(s):
    s = s.strip().rstrip(",")
    if s.lower() in (".true.", "t", ".t."):
        return True
    if s.lower() in (".false.", "f", ".f."):
        return False
    if s[:1] in "'\"":
        return s.strip("'\"")
    t = re.sub(r"[dD]", "e", s)
    try:
        return int(t)
    except ValueError:
        return float(t)


def read_namelist(path):
    """&parameters par%name = value ... /  (setup.f90:27-40).  Unknown names fail like a Fortran namelist read."""
    par = Params()
    txt = open(path).read()
    m = re.search(r"&parameters(.*?)^\s*/", txt, re.S | re.M | re.I)
    if not m:
        raise LartInputError("no &parameters namelist group in " + path)
    # strip comments, then split the group body at every `par%name[(i)] =`: several assignments may share a line
    # (par%nx = 11, par%ny = 11) and a value list may follow one name (par%alpha = -90.0, -90.0)
    body = "\n".join(ln.split("!")[0] for ln in m.group(1).splitlines())
    parts = re.split(r"(par%\w+\s*(?:\(\s*\d+\s*\))?\s*=)", body, flags=re.I)
    if parts[0].strip().strip(","):
        raise LartInputError("cannot parse namelist text: " + parts[0].strip())
    for key, val in zip(parts[1::2], parts[2::2]):
        mm = re.match(r"par%(\w+)\s*(?:\(\s*(\d+)\s*\))?\s*=", key, re.I)
        val = val.strip().rstrip(",").strip()
        if not val:
            raise LartInputError("no value for par%" + mm.group(1))
        if val[:1] in "'\"":
            v = _parse_value(val)
        else:
            items = [t for t in re.split(r"[,\s]+", val) if t]
            v = _parse_value(items[0]) if len(items) == 1 else [_parse_value(t) for t in items]
        set_param(par, mm.group(1), v, mm.group(2))
    return par


def set_param(par, name, val, idx=None):
    if name in _UNSUPPORTED_DEFAULTS:
        if val != _UNSUPPORTED_DEFAULTS[name]:
            raise LartInputError("par%%%s = %r is not supported by this path (only the default %r)" % (name, val, _UNSUPPORTED_DEFAULTS[name]))
        return
    if not hasattr(par, name):
        raise LartInputError("unknown namelist variable par%" + name)
    cur = getattr(par, name)
    if isinstance(cur, list):
        vals = val if isinstance(val, (list, tuple)) else [val]
        i0 = int(idx) - 1 if idx else 0
        for k, v in enumerate(vals):
            cur[i0 + k] = float(v)
    elif isinstance(cur, bool):
        setattr(par, name, bool(val))
    elif isinstance(cur, int):
        setattr(par, name, int(val))
    elif isinstance(cur, float):
        setattr(par, name, float(val))
    else:
        setattr(par, name, str(val))


@dataclass
class Line:
    """line_type (define.f90:639-680) for Ly-alpha: line_mod.f90:1241-1270"""
    wavelength0: float = 0.1215668237310
    damping: float = 6.2649e8
    cross0: float = 0.026540083434 / math.sqrt(PI) * (0.27760 + 0.13881 + 0.0)
    vtherm1: float = 0.12895319011972164 / math.sqrt(1.00797)
    g_recoil0: float = (H_PLANCK / 1.67262192e-24 / 1.00797) / (0.1215668237310 * UM2M) ** 2
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
    n = len(probs)
    probs = np.array(probs, dtype=np.float64) * n
    alias = np.zeros(n, dtype=np.int32)
    small = [i + 1 for i in range(n) if probs[i] < 1.0]
    large = [i + 1 for i in range(n) if not probs[i] < 1.0]
    while small and large:
        ks, kl = small.pop(), large.pop()
        alias[ks - 1] = kl
        probs[kl - 1] = probs[kl - 1] + probs[ks - 1] - 1.0
        (small if probs[kl - 1] < 1.0 else large).append(kl)
    return probs, alias


def setup_scattering_matrix(par, path):
    """setup.f90:581-649: header overrides albedo/hgg/cext_dust; S11..S34 normalised by the trapezoid integral of S11"""
    with open(path) as f:
        f.readline()
        h = f.readline().split()
        f.readline()
        par.cext_dust, par.albedo, par.hgg, n = float(h[1]), float(h[2]), float(h[3]), int(h[4])
        tab = np.array([[float(t) for t in f.readline().split()[:5]] for _ in range(n)])
    sm = ScattMat(nPDF=n, coss=tab[:, 0].copy(), S11=tab[:, 1].copy(), S12=tab[:, 2].copy(), S33=tab[:, 3].copy(), S34=tab[:, 4].copy())
    norm = 0.0
    for i in range(1, n):
        norm = norm + 0.5 * (sm.S11[i] + sm.S11[i - 1]) * abs(sm.coss[i] - sm.coss[i - 1])
    for a in (sm.S11, sm.S12, sm.S33, sm.S34):
        a /= norm
    pdf = (sm.S11[:-1] + sm.S11[1:]) / 2.0
    pdf = pdf / pdf.sum()
    sm.phase_PDF, sm.alias = random_alias_setup(pdf)
    return sm


# par%line_id -> line type, wavelength0 [um], f12(1), damping, mass of vtherm1, mass of g_recoil0, the two level wavenumbers [1/cm] of
# DnuHK_Hz (setup_resonance_line, line_mod.f90:590-713).  As in the reference, CIV_1548 recoils with line%mass_amu's default 1.00797
# (define.f90:646, line_mod.f90:605) and AlII_1671's mass is a single precision literal (:706).
_AL_MASS = float(np.float32(26.98154))
_METAL_LINES = {
    "CIV_1548":  (2, 0.1548187,    0.190, 2.647e8,    12.011,   1.00797,  64591.7,     64484.0),
    "NV_1239":   (2, 0.1238821,    0.156, 3.390e8,    14.0067,  14.0067,  80721.9,     80463.2),
    "OVI_1032":  (2, 0.1031912,    0.133, 4.137e8,    15.9994,  15.9994,  96907.5,     96375.0),
    "NaI_D":     (2, 0.5891583253, 0.641, 6.153e7,    22.98977, 22.98977, 16973.36619, 16956.17025),
    "CaII_HK":   (2, 0.3934777,    0.682, 1.446667e8, 40.078,   40.078,   25414.40,    25191.51),
    "MgII_2796": (2, 0.2796352,    0.608, 2.590e8,    24.305,   24.305,   35760.88,    35669.31),
    "SiIV_1394": (2, 0.1393755,    0.513, 8.743e8,    28.0855,  28.0855,  71748.64,    71287.54),
    "AlII_1671": (1, 0.16707874,   1.77,  1.41e9,     _AL_MASS, _AL_MASS, 0.0,         0.0),
}

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


def read_energy_X(fname):
    """read_energy_X (h2_mod.f90:250-296): rows 'V J E[cm^-1]' after a magic-number line and '#' comments"""
    if not os.path.exists(fname):
        raise LartInputError("h2_mod: cannot find " + fname)
    rows = []
    for ln in open(fname):
        q = ln.strip()
        if not q or q.startswith("#") or "//" in q:
            continue
        t = q.split()
        try:
            rows.append((float(t[0]), float(t[1]), float(t[2])))
        except (ValueError, IndexError):
            continue
    return np.array(rows)


def h2_init(par, line):
    """h2_init (h2_mod.f90:120-222): the two B-X lines closest to Ly-alpha, R(6) and P(5) of v = 1 <- 2 (Neufeld 1990), LTE populations of the lower
    levels at par%h2_temperature from the X-state level table <par%h2_data_dir>/energy_X.dat, strengths relative to the H I line-centre opacity"""
    speedc, f_osc_Lya, hc_over_k, f_osc_const = SPEEDC, 0.4162, 1.4387769, 1.4992e-16
    nu_Lya = (speedc * 1.0e5) / (line.wavelength0 * UM2M * 1.0e2)
    vth1 = line.vtherm1 * math.sqrt(line.mass_amu / (2.0 * line.mass_amu)); T = par.h2_temperature
    bt2 = par.bturb ** 2 if par.bturb > 0.0 else 0.0
    vth = math.sqrt((vth1 * math.sqrt(T)) ** 2 + bt2)
    Dfreq = nu_Lya * vth / speedc
    dv, lam, A_ul, A_tot = (14.140, 99.229), (1215.72534, 1216.07038), (1.36e8, 1.59e8), (1.6825e9, 1.7199e9)
    vl, Jl, Ju = (2, 2), (6, 5), (7, 4)
    lev = read_energy_X(os.path.join(par.h2_data_dir or "data/h2", "energy_X.dat"))
    Z = 0.0
    for vv, jj, ee in lev:                                                # the reference's summation order
        gns = 3.0 if int(round(jj)) % 2 != 0 else 1.0
        Z = Z + gns * (2.0 * jj + 1.0) * math.exp(-hc_over_k * ee / T)
    out = dict(dnu=[], s=[], a=[], p=[], pop=[], f=[])
    for i in range(2):
        sel = [ee for vv, jj, ee in lev if int(round(vv)) == vl[i] and int(round(jj)) == Jl[i]]
        if not sel:
            raise LartInputError("h2_mod: X-level (v=%d, J=%d) not found in energy_X.dat" % (vl[i], Jl[i]))
        f_osc = f_osc_const * (2.0 * Ju[i] + 1.0) / (2.0 * Jl[i] + 1.0) * lam[i] ** 2 * A_ul[i]
        gns = 3.0 if Jl[i] % 2 != 0 else 1.0
        pop = gns * (2.0 * Jl[i] + 1.0) * math.exp(-hc_over_k * sel[0] / T) / Z
        out["dnu"].append(-(dv[i] / speedc) * nu_Lya); out["f"].append(f_osc); out["pop"].append(pop)
        out["s"].append(par.f_H2 * pop * (f_osc / f_osc_Lya)); out["a"].append(A_tot[i] / (4.0 * PI * Dfreq))
        out["p"].append(0.0 if par.h2_pure_absorption else A_ul[i] / A_tot[i])
    return H2Lines(n=2, Dfreq_Hz=Dfreq, dv_kms=dv, dnu_Hz=tuple(out["dnu"]), strength=tuple(out["s"]), a_damp=tuple(out["a"]), p_scat=tuple(out["p"]),
                   pop=tuple(out["pop"]), f_osc=tuple(out["f"]))


def read_input(par, line=None):
    """post-processing part of read_input (setup.f90:42-562) for the fields this path uses. Returns (par, line, scatt_mat)."""
    line = line or Line()
    ml = capi.line_lookup(par.line_id) if par.line_id not in ("ly_alpha", "ly_alpha_HD", "ly_beta") and par.line_id not in _METAL_LINES else None
    if ml is not None:                                                    # resonance + fluorescence multiplets, types 4-6: line_mod.f90:713-1153
        line.line_type, line.wavelength0, line.damping, line.cross0 = ml.line_type, ml.wavelength0, ml.damping, ml.cross0
        line.vtherm1, line.g_recoil0, line.DnuHK_Hz, line.multiplet = ml.vtherm1, ml.g_recoil0, 0.0, ml
        line.E1, line.E2, line.E3 = 1.0, 0.0, 1.0
        ml.HeI_coherent = 1 if (par.HeI_coherent and ml.line_type == 6) else 0
        par.fine_structure = par.include_deuterium = False
    elif par.line_id in _METAL_LINES:                                     # single lines / S-P doublets: line_mod.f90:590-713
        typ, wl, f1, damping, mass, mass_recoil, k_hi, k_lo = _METAL_LINES[par.line_id]
        line.line_type, line.wavelength0, line.damping = typ, wl, damping
        line.cross0 = 0.026540083434 / math.sqrt(PI) * f1
        line.vtherm1 = 0.12895319011972164 / math.sqrt(mass)
        line.g_recoil0 = (H_PLANCK / 1.67262192e-24 / mass_recoil) / (wl * UM2M) ** 2
        line.DnuHK_Hz = (2.99792458e5 * 1.0e5) * (k_hi - k_lo) if typ == 2 else 0.0
        line.E1, line.E2, line.E3 = 1.0, 0.0, 1.0
        par.fine_structure = par.include_deuterium = False               # (both only act on the ly_alpha branches)
    elif par.line_id == "ly_beta":                                        # line_mod.f90:1192-1240: H I Lyman beta with the 3p -> 2s conversion to H-alpha
        line.line_type, line.wavelength0, line.damping = 8, 0.10257222, 1.8970e8
        line.cross0 = 0.026540083434 / math.sqrt(PI) * 0.07910
        line.vtherm1 = 0.12895319011972164 / math.sqrt(1.00797)
        line.g_recoil0 = (H_PLANCK / 1.67262192e-24 / 1.00797) / (line.wavelength0 * UM2M) ** 2
        line.DnuHK_Hz, line.E1, line.E2, line.E3 = 0.0, 1.0, 0.0, 1.0
        A21 = (1.6725e8, 2.2448e7)
        line.P_conv = A21[1] / (A21[0] + A21[1])
        if par.fine_structure:                                            # setup.f90:259-263
            raise LartInputError("ly_beta: fine structure not yet supported")
        par.include_deuterium = False
    elif par.line_id not in ("ly_alpha", "ly_alpha_HD"):
        raise LartInputError("par%%line_id = '%s' is not supported by this path (ly_alpha, ly_alpha_HD, %s)" % (par.line_id, ", ".join(_METAL_LINES) + " and the multiplets of lart_host_line_lookup"))
    if par.include_deuterium and par.line_id == "ly_alpha":               # line_mod.f90:580-588
        par.line_id = "ly_alpha_HD"
    if par.line_id == "ly_alpha_HD":                                      # :1155-1191
        par.fine_structure = False
        wl_D, mass_D, damping_D = 0.1215337431, 2.01410177812, 6.2649e8
        speedc_cm = 2.99792458e5 * 1.0e5
        line.line_type = 7; line.nD_over_nH = par.D_to_H_ratio
        line.delta_nu_HD_Hz = speedc_cm * (1.0 / (wl_D * UM2M * 1.0e2) - 1.0 / (line.wavelength0 * UM2M * 1.0e2))
        line.ratio_Dfreq_HD = (wl_D / line.wavelength0) * math.sqrt(mass_D / 1.00797)
        line.ratio_voigta_HD = (damping_D / line.damping) * line.ratio_Dfreq_HD
        line.g_recoil0_D = (H_PLANCK / 1.67262192e-24 / mass_D) / (wl_D * UM2M) ** 2
    if par.fine_structure:                                                # setup_resonance_line, line_mod.f90:1246-1253
        speedc_cm = 2.99792458e5 * 1.0e5
        line.line_type, line.DnuHK_Hz = 2, 82259.2850014 * speedc_cm - 82258.9191133 * speedc_cm
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
    if par.nx == 1 or par.ny == 1 or par.nz == 1:
        par.xyz_symmetry = False                                          # setup.f90:167
    if par.nside > 0 and par.use_clump_medium:                           # setup.f90:177-181
        raise LartInputError("ERROR: nside>0 (HEALPIX inside observer) is not supported with clump mode")
    if par.use_clump_medium and par.use_amr_grid:
        raise LartInputError("par%use_clump_medium and par%use_amr_grid are mutually exclusive")
    if par.nside > 0:                                                    # observer inside the medium: setup.f90:169-186
        par.use_stokes = False
        par.save_peeloff = True
        par.xyz_symmetry = False
        par.nxim = par.nyim = 0
    if not (par.save_peeloff_2D or par.save_peeloff_3D):                  # :188-206
        par.save_peeloff = False
    if par.nxim > 0 and par.nyim > 0:
        par.save_peeloff = True
    if par.save_peeloff and par.xyz_symmetry:
        par.save_peeloff = False
        print("---> peelingoff is not allowed for a geometry with the xyz_symmetry.\n---> setting par%save_peeloff = .false.")
    if not par.save_peeloff:
        par.save_peeloff_2D = par.save_peeloff_3D = False
    if par.tau0 > 0.0 and par.taumax < 0.0:                               # :228-233
        par.taumax = par.tau0
    if par.N_HI > 0.0 and par.N_HImax < 0.0:
        par.N_HImax = par.N_HI
    if par.N_HImax > 0.0 and par.N_gasmax < 0.0:
        par.N_gasmax = par.N_HImax
    if par.N_HIhomo > 0.0 and par.N_gashomo < 0.0:
        par.N_gashomo = par.N_HIhomo
    if par.cext_dust <= 0.0:
        par.DGR = 0.0
    if par.DGR == 0.0:
        par.save_Jabs = False
    if par.core_skip_global:
        par.core_skip = True
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
    if par.save_Jmu:                                                      # :367-381
        if par.nmu < 1:
            raise LartInputError("ERROR: par%nmu must be >= 1 when par%save_Jmu = .true.")
        if par.xyz_symmetry:
            par.mu_min, par.dmu = 0.0, 1.0 / par.nmu
        else:
            par.mu_min, par.dmu = -1.0, 2.0 / par.nmu
    if par.nr > 1:
        par.nx = par.ny = par.nz = par.nr
    if par.use_amr_grid:
        if par.amr_type.lower() != "generic":
            raise LartInputError("par%%amr_type = '%s' is not supported by this path (generic text files only)" % par.amr_type)
        if not par.amr_file:
            raise LartInputError("par%use_amr_grid = .true. needs par%amr_file")
    if par.geometry == "sphere":                                          # :398-428
        if not par.xy_periodic and not par.use_amr_grid:
            c = [v for v in (par.rmax, par.xmax, par.ymax, par.zmax) if v > 0.0]
            if c:
                par.rmax = par.xmax = par.ymax = par.zmax = max(c)
            if not par.xy_symmetry:
                par.nx = par.ny = par.nz = max(par.nx, par.ny, par.nz)
    else:
        par.rmax = -1.0
    if par.xy_symmetry:                                                   # mirrors in x and y, the full z range (grid_mod_car.f90:113-134)
        if par.xyz_symmetry or par.xy_periodic or par.use_clump_medium:
            raise LartInputError("par%xy_symmetry excludes xyz_symmetry / xy_periodic / clumps")
        if par.source_geometry != "point" or par.xs_point != 0.0 or par.ys_point != 0.0:
            raise LartInputError("par%xy_symmetry needs a point source on the z axis (sources are folded for xyz_symmetry only, generate_photon.f90:352-356)")
        if par.calc_J or par.calc_P or par.calc_Pnew:
            raise LartInputError("par%geometry_JPa = 2 (cylindrical profiles, the xy_symmetry default) is not supported by this path")
    if par.source_rmax < 0.0:
        par.source_rmax = par.rmax
    if par.rmax <= 0.0:                                                   # geometry of the J/P profiles: setup.f90:438-459
        par.geometry_JPa = 3
    if par.geometry_JPa < -1 or par.geometry_JPa > 3:
        par.geometry_JPa = 3 if par.save_all else (-1 if par.xy_periodic else 1)
    if par.geometry_JPa in (0, 2) and (par.calc_J or par.calc_P or par.calc_Pnew):
        raise LartInputError("par%geometry_JPa = 2 (cylindrical profiles) is not supported by this path")
    if par.distance2cm < 0.0:                                             # :473-489
        par.distance2cm = {"kpc": 3.0856776e21, "pc": 3.0856776e18, "au": 1.4960e13, "": 1.0}.get(par.distance_unit, 3.0856776e21)
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


def _sym_axis(n, vmax):
    if (n // 2) * 2 == n:
        return vmax / n, 0.0, 1
    d = vmax / (n - 0.5)
    return d, -d / 2.0, 2


def voigt_center(a):
    """H(0,a) from the first Harris terms on the x=0 table node: voigt_mod.f90:691-700 with v = 0"""
    return 1.0 + a * (-1.1283791671 + a * 1.0)


def grid_create(par, line):
    """grid_mod_car.f90:73-1238, analytic-medium subset (no input files)."""
    g = Grid(nx=par.nx, ny=par.ny, nz=par.nz, xmax=par.xmax, ymax=par.ymax, zmax=par.zmax, rmin=par.rmin, rmax=par.rmax)
    nx, ny, nz = g.nx, g.ny, g.nz
    if par.xyz_symmetry:
        g.dx, g.xmin, g.i0 = _sym_axis(nx, g.xmax)
        g.dy, g.ymin, g.j0 = _sym_axis(ny, g.ymax)
        g.dz, g.zmin, g.k0 = _sym_axis(nz, g.zmax)
    elif par.xy_symmetry:                                                 # :113-134
        g.dx, g.xmin, g.i0 = _sym_axis(nx, g.xmax)
        g.dy, g.ymin, g.j0 = _sym_axis(ny, g.ymax)
        g.dz, g.zmin, g.k0 = 2.0 * par.zmax / nz, -g.zmax, 0
    else:
        g.dx, g.dy, g.dz = 2.0 * par.xmax / nx, 2.0 * par.ymax / ny, 2.0 * par.zmax / nz
        g.xmin, g.ymin, g.zmin = -g.xmax, -g.ymax, -g.zmax
    g.xface = np.arange(nx + 1) * g.dx + g.xmin
    g.yface = np.arange(ny + 1) * g.dy + g.ymin
    g.zface = np.arange(nz + 1) * g.dz + g.zmin
    xx = (g.xface[:-1] + g.xface[1:]) / 2.0
    yy = (g.yface[:-1] + g.yface[1:]) / 2.0
    zz = (g.zface[:-1] + g.zface[1:]) / 2.0
    shape = (nx, ny, nz)
    vth = vtherm_total(line, par.temperature)
    g.Dfreq_ref = vth / (line.wavelength0 * UM2KM)
    g.Dfreq = np.full(shape, vth / (line.wavelength0 * UM2KM), order="F")
    g.voigt_a = (line.damping / FOURPI) / g.Dfreq
    if not par.use_clump_medium:                                          # :352-363: the clump population carries the physical length unit
        par.distance_unit, par.distance2cm = "", 1.0
    g.rhokap = np.ones(shape, order="F")
    g.rhokapD = np.full(shape, par.cext_dust * par.DGR, order="F") if par.DGR > 0.0 else None
    X, Y, Z = np.meshgrid(xx, yy, zz, indexing="ij")
    rr = np.sqrt(X ** 2 + Y ** 2 + Z ** 2)
    if par.rmax > 0.0:                                                    # :367-395
        out = (rr < par.rmin) | (rr > par.rmax)
        g.rhokap[out] = 0.0
        if g.rhokapD is not None:
            g.rhokapD[out] = 0.0
    if 0.0 < par.cone_opening < 90.0:                                     # :397-412 biconical mask about the z axis
        with np.errstate(invalid="ignore", divide="ignore"):
            out = (rr > 0.0) & (np.abs(Z) / rr < math.cos(par.cone_opening * math.pi / 180.0))
        g.rhokap[out] = 0.0
        if g.rhokapD is not None:
            g.rhokapD[out] = 0.0
    g.rhokap = g.rhokap / g.Dfreq * line.cross0                           # :487-493
    if par.rmax > 0.0 and par.rmin > 0.0:
        opac_length = par.rmax - par.rmin
    elif par.rmax > 0.0:
        opac_length = par.rmax
    elif g.zmax == -g.zmin:
        opac_length = (g.zmax - g.zmin) / 2.0
    else:
        opac_length = g.zmax - g.zmin
    nxcen, nycen = (1, 1) if (par.xyz_symmetry or par.xy_symmetry) else ((nx + 1) // 2, (ny + 1) // 2)
    H0 = voigt_center(g.voigt_a)
    pole_k = g.rhokap[nxcen - 1, nycen - 1, :] * H0[nxcen - 1, nycen - 1, :]
    pole_n = g.rhokap[nxcen - 1, nycen - 1, :] * g.Dfreq[nxcen - 1, nycen - 1, :]
    nadd = np.ones(shape)
    if par.xyz_symmetry:
        if nx % 2:
            nadd[0, :, :] /= 2.0
        if ny % 2:
            nadd[:, 0, :] /= 2.0
        if nz % 2:
            nadd[:, :, 0] /= 2.0
    elif par.xy_symmetry:
        if nx % 2:
            nadd[0, :, :] /= 2.0
        if ny % 2:
            nadd[:, 0, :] /= 2.0
    gas = g.rhokap > 0.0

    def pole(v):
        s = _seqsum(v)
        if par.xyz_symmetry:
            return (s - v[0] / 2.0) * g.dz if nz % 2 else s * g.dz
        return s * g.dz / 2.0 if g.zmax == -g.zmin else s * g.dz

    opac_norm = None
    if par.taumax > 0.0:                                                  # :518-538
        opac_norm = par.taumax / pole(pole_k)
    elif par.tauhomo > 0.0:                                               # :539-566
        opac_norm = par.tauhomo / ((g.rhokap * H0 * nadd)[gas].sum() / nadd[gas].sum() * opac_length)
    elif par.N_gasmax > 0.0:                                              # :567-586
        opac_norm = par.N_gasmax / (pole(pole_n) / line.cross0)
    elif par.N_gashomo > 0.0:                                             # :587-615
        opac_norm = par.N_gashomo / ((g.rhokap * g.Dfreq * nadd)[gas].sum() / nadd[gas].sum() / line.cross0 * opac_length)
    if opac_norm is not None:
        g.rhokap *= opac_norm
        if g.rhokapD is not None:
            g.rhokapD *= opac_norm
        pole_k = pole_k * opac_norm
        pole_n = pole_n * opac_norm
    g.tauhomo = (g.rhokap * H0 * nadd)[gas].sum() / nadd[gas].sum() * opac_length      # :617-640
    g.taupole = pole(pole_k)
    N_gashomo = (g.rhokap * g.Dfreq * nadd)[gas].sum() / nadd[gas].sum() / line.cross0 * opac_length
    N_gaspole = pole(pole_n) / line.cross0
    if par.taumax <= 0.0:
        par.taumax = g.taupole
    if par.tauhomo <= 0.0:
        par.tauhomo = g.tauhomo
    if par.N_gasmax <= 0.0:
        par.N_gasmax = N_gaspole
    if par.N_gashomo <= 0.0:
        par.N_gashomo = N_gashomo
    g.vfx = np.zeros(shape, order="F"); g.vfy = np.zeros(shape, order="F"); g.vfz = np.zeros(shape, order="F")
    if par.velocity_type:                                                 # :786-946
        rpeak = max(g.xmax, g.ymax, g.zmax) if par.rmax <= 0.0 else par.rmax
        if par.velocity_type == "hubble":
            g.vfx[gas] = ((par.Vexp / vth) * X / rpeak)[gas]; g.vfy[gas] = ((par.Vexp / vth) * Y / rpeak)[gas]; g.vfz[gas] = ((par.Vexp / vth) * Z / rpeak)[gas]
        elif par.velocity_type == "constant_radial":
            m = gas & (rr > g.dz / 10.0)
            with np.errstate(invalid="ignore", divide="ignore"):
                g.vfx[m] = (par.Vexp / vth * X / rr)[m]; g.vfy[m] = (par.Vexp / vth * Y / rr)[m]; g.vfz[m] = (par.Vexp / vth * Z / rr)[m]
        elif par.velocity_type == "parallel_velocity":
            g.vfx[gas] = par.Vx / vth; g.vfy[gas] = par.Vy / vth; g.vfz[gas] = par.Vz / vth
        elif par.velocity_type == "rotating_solid_body":                  # :905-916
            g.vfx[gas] = (-par.Vrot / vth * Y / par.rmax)[gas]; g.vfy[gas] = (par.Vrot / vth * X / par.rmax)[gas]
        elif par.velocity_type == "rotating_galaxy_halo":                 # :917-934
            rc = np.sqrt(X ** 2 + Y ** 2)
            with np.errstate(invalid="ignore", divide="ignore"):
                den = np.where(rc < par.rinner, par.rinner, rc)
                g.vfx[gas] = (-par.Vrot / vth * Y / den)[gas]; g.vfy[gas] = (par.Vrot / vth * X / den)[gas]
        else:                                                             # power_law :853-874, linear_decelerate :875-904, ssh :817-838
            m = gas & (rr > g.dz / 10.0)
            with np.errstate(invalid="ignore", divide="ignore"):
                if par.velocity_type == "power_law":
                    Vs = par.Vexp * (rr / rpeak) ** par.velocity_alpha
                elif par.velocity_type == "linear_decelerate":
                    Vs = par.Vexp * np.maximum(0.0, (rpeak - rr) / (rpeak - max(par.rmin, 0.0)))
                else:
                    rp = par.rpeak; m = gas
                    Vs = np.where(rr < rp, par.Vpeak / rp * rr, par.Vpeak + par.DeltaV * (rr - rp) / (par.rmax - rp))
                ur = np.where(rr > 0.0, 1.0 / np.where(rr > 0.0, rr, 1.0), 0.0)
                g.vfx[m] = (Vs / vth * X * ur)[m]; g.vfy[m] = (Vs / vth * Y * ur)[m]; g.vfz[m] = (Vs / vth * Z * ur)[m]
    g.voigt_amean = (line.damping / FOURPI) / g.Dfreq_ref                 # :1150-1154
    atau0 = g.voigt_amean * par.tauhomo
    car_setup_freq_grid(par, line, g)
    at = atau0 if par.core_skip_global else atau0 / (g.xmax / g.dx)       # :1186-1219
    if at > 1.0:
        xi, chi = (0.6, 1.2) if at <= 60.0 else (1.4, 0.6)
        g.xcrit = 0.02 * math.exp(xi * math.log(at) ** chi)
        g.xcrit2 = g.xcrit ** 2
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
    nx, ny, nz = g.nx, g.ny, g.nz
    nadd = np.ones((nx, ny, nz))
    if par.xyz_symmetry:
        nadd[:] = 8.0
        if nx % 2:
            nadd[0, :, :] /= 2.0
        if ny % 2:
            nadd[:, 0, :] /= 2.0
        if nz % 2:
            nadd[:, :, 0] /= 2.0
    gas = g.rhokap > 0.0
    if par.geometry_JPa == 1:
        nr = max(nx, ny, nz)
        if not par.xyz_symmetry:
            nr = nr // 2 if nr % 2 == 0 else (nr - 1) // 2 + 1
        rmax = min(g.xmax, g.ymax, g.zmax)
        dr, roff = (rmax / nr, 0.0) if nr % 2 == 0 else (rmax / (nr - 0.5), -(rmax / (nr - 0.5)) / 2.0)
        xx = (g.xface[:-1] + g.xface[1:]) / 2.0; yy = (g.yface[:-1] + g.yface[1:]) / 2.0; zz = (g.zface[:-1] + g.zface[1:]) / 2.0
        X, Y, Z = np.meshgrid(xx, yy, zz, indexing="ij")
        ib = np.floor((np.sqrt(X ** 2 + Y ** 2 + Z ** 2) - roff) / dr).astype(np.int64) + 1
        ok = (ib >= 1) & (ib <= nr)
        ind = np.where(ok, ib, 0).astype(np.int32)
        ncount = np.zeros(nr)
        np.add.at(ncount, ib[ok & gas] - 1, nadd[ok & gas])
        return JPa(nr, np.asfortranarray(ind), ncount)
    if par.geometry_JPa == -1:
        ind = np.broadcast_to(np.arange(1, nz + 1, dtype=np.int32)[None, None, :], (nx, ny, nz))
        return JPa(nz, np.asfortranarray(ind), gas.sum(axis=(0, 1)).astype(float))
    ind = (np.arange(nx * ny * nz, dtype=np.int32) + 1).reshape((nx, ny, nz), order="F")
    return JPa(nx * ny * nz, np.asfortranarray(ind), (nadd if par.xyz_symmetry else np.ones((nx, ny, nz))).ravel(order="F"))


def _seqsum(v):
    s = 0.0
    for t in v:
        s = s + float(t)
    return s


def car_setup_freq_grid(par, line, g):
    """grid_mod_car.f90:1442-1548"""
    par.atau3 = (g.voigt_amean * par.tauhomo) ** (1.0 / 3.0)
    vth = vtherm_total(line, par.temperature)
    if math.isfinite(par.wavelength_min) and math.isfinite(par.wavelength_max):
        if par.nwavelength == 0 and par.nxfreq > 0:
            par.nwavelength = par.nxfreq
        if par.nwavelength > 0:
            par.nxfreq = par.nwavelength
        speedc = 2.99792458e5
        par.xfreq_min = -(par.wavelength_max - line.wavelength0 * 1e4) / (line.wavelength0 * 1e4) * (speedc / vth)
        par.xfreq_max = -(par.wavelength_min - line.wavelength0 * 1e4) / (line.wavelength0 * 1e4) * (speedc / vth)
    elif math.isfinite(par.velocity_min) and math.isfinite(par.velocity_max):
        if par.nvelocity == 0 and par.nxfreq > 0:
            par.nvelocity = par.nxfreq
        if par.nvelocity > 0:
            par.nxfreq = par.nvelocity
        par.xfreq_min, par.xfreq_max = -par.velocity_max / vth, -par.velocity_min / vth
    g.nxfreq = par.nxfreq
    if not (math.isfinite(par.xfreq_max) and math.isfinite(par.xfreq_min)):
        xscale = 25.0 if par.taumax <= 5e1 else 14.0 if par.taumax <= 5e2 else 10.0 if par.taumax <= 5e3 else 5.0
        if par.Vexp == 0.0:
            par.xfreq_max = math.floor(xscale * par.atau3) + 1
            par.xfreq_min = -(math.floor(xscale * par.atau3 + line.DnuHK_Hz / g.Dfreq_ref) + 1)
        elif par.Vexp > 0.0:
            par.xfreq_max = math.floor(xscale * par.atau3) + 1
            par.xfreq_min = -(math.floor(xscale * par.atau3 + abs(par.Vexp) / vth + line.DnuHK_Hz / g.Dfreq_ref) + 1)
        else:
            par.xfreq_max = math.floor(xscale * par.atau3 + abs(par.Vexp) / vth) + 1
            par.xfreq_min = -(math.floor(xscale * par.atau3 + line.DnuHK_Hz / g.Dfreq_ref) + 1)
        if par.spectral_type == "continuum":                               # :1490-1494
            xscale = 4.0 * xscale
            par.xfreq_max = math.floor(xscale * par.atau3 + abs(par.Vexp) / vth) + 1
            par.xfreq_min = -(math.floor(xscale * par.atau3 + abs(par.Vexp) / vth + line.DnuHK_Hz / g.Dfreq_ref) + 1)
    g.dxfreq = (par.xfreq_max - par.xfreq_min) / par.nxfreq
    g.xfreq_max, g.xfreq_min = float(par.xfreq_max), float(par.xfreq_min)
    if line.line_type == 8:                                                # :1518-1529 the band-2 (H-alpha) axis
        g.nxfreq_Ha = par.nxfreq_Ha if par.nxfreq_Ha > 0 else g.nxfreq
        par.nxfreq_Ha = g.nxfreq_Ha
        if par.xfreq_max_Ha > 0.0:
            g.xfreq_max_Ha, g.xfreq_min_Ha = float(par.xfreq_max_Ha), -float(par.xfreq_max_Ha)
        else:
            g.xfreq_min_Ha, g.xfreq_max_Ha = g.xfreq_min, g.xfreq_max
        g.dxfreq_Ha = (g.xfreq_max_Ha - g.xfreq_min_Ha) / g.nxfreq_Ha
        g.dwave_Ha = vth / SPEEDC * (line.wavelength0_Ha * 1e4) * g.dxfreq_Ha
    g.xfreq = (np.arange(1, g.nxfreq + 1) - 0.5) * g.dxfreq + g.xfreq_min
    g.velocity = -vth * g.xfreq
    g.dwave = vth / SPEEDC * (line.wavelength0 * 1e4) * g.dxfreq
    g.wavelength = (g.velocity / SPEEDC + 1.0) * (line.wavelength0 * 1e4)


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
    """observer_create_outside: observer_rect.f90:24-326 (rotation centre at the origin)"""
    if not par.save_peeloff:
        return Observers()
    fin = math.isfinite
    if par.nside > 0:                                                     # observer_create_inside: observer_heal.f90:24-80
        if not (fin(par.obsx[0]) and fin(par.obsy[0]) and fin(par.obsz[0])):
            par.obsx[0] = par.obsy[0] = par.obsz[0] = 0.0
        n = sum(1 for i in range(MAX_OBSERVERS) if fin(par.obsx[i]) and fin(par.obsy[i]) and fin(par.obsz[i]))
        par.nobs, par.npix = n, 12 * par.nside ** 2
        return Observers(nobs=n, xyz=np.array([[par.obsx[i], par.obsy[i], par.obsz[i]] for i in range(n)]), nxim=par.npix, nyim=1,
                         steradian_pix=4.0 * PI / par.npix)
    maxsz = max(par.xmax, par.ymax, par.zmax)
    for i in range(MAX_OBSERVERS):
        if fin(par.beta[i]) and not fin(par.alpha[i]):
            par.alpha[i] = 0.0
        if fin(par.alpha[i]) and not fin(par.beta[i]):
            par.beta[i] = 0.0
    if not (fin(par.alpha[0]) and fin(par.beta[0])) and not (fin(par.obsx[0]) and fin(par.obsy[0]) and fin(par.obsz[0])):
        if not fin(par.distance):
            par.distance = maxsz * 100.0
        par.obsx[0], par.obsy[0], par.obsz[0] = 0.0, 0.0, 1.0
        par.alpha[0], par.beta[0] = 0.0, 0.0

    def default_gamma(i):
        if not fin(par.gamma[i]):
            par.gamma[i] = 90.0 if 0.0 < par.beta[i] <= 90.0 else (-90.0 if par.beta[i] > 90.0 else 0.0)

    xyz, trig = [], []
    if fin(par.alpha[0]) and fin(par.beta[0]):
        n = sum(1 for i in range(MAX_OBSERVERS) if fin(par.alpha[i]) and fin(par.beta[i]))
        if not fin(par.distance):
            par.distance = maxsz * 100.0
        for i in range(n):
            default_gamma(i)
            ca, sa = math.cos(par.alpha[i] * DEG2RAD), math.sin(par.alpha[i] * DEG2RAD)
            cb, sb = math.cos(par.beta[i] * DEG2RAD), math.sin(par.beta[i] * DEG2RAD)
            cg, sg = math.cos(par.gamma[i] * DEG2RAD), math.sin(par.gamma[i] * DEG2RAD)
            xyz.append((par.distance * ca * sb, par.distance * sa * sb, par.distance * cb))
            trig.append((ca, sa, cb, sb, cg, sg))
    else:
        n = sum(1 for i in range(MAX_OBSERVERS) if fin(par.obsx[i]) and fin(par.obsy[i]) and fin(par.obsz[i]))
        if not fin(par.distance):
            par.distance = math.sqrt(par.obsx[0] ** 2 + par.obsy[0] ** 2 + par.obsz[0] ** 2)
            if par.distance < 10.0 * maxsz:
                par.distance = maxsz * 100.0
        eps = np.finfo(np.float64).eps
        for i in range(n):
            default_gamma(i)
            ds = par.distance / math.sqrt(par.obsx[i] ** 2 + par.obsy[i] ** 2 + par.obsz[i] ** 2)
            p = (par.obsx[i] * ds, par.obsy[i] * ds, par.obsz[i] * ds) if ds > 1.001 else (par.obsx[i], par.obsy[i], par.obsz[i])
            cb = p[2] / par.distance
            if abs(cb - 1.0) < eps:
                cb = 1.0
            if abs(cb + 1.0) < eps:
                cb = -1.0
            sb = math.sqrt(1.0 - cb * cb)
            par.beta[i] = math.atan2(sb, cb) * RAD2DEG
            cg, sg = math.cos(par.gamma[i] * DEG2RAD), math.sin(par.gamma[i] * DEG2RAD)
            if sb == 0.0:
                ca, sa, par.alpha[i] = 1.0, 0.0, 0.0
            else:
                al = math.atan2(p[1], p[0]); ca, sa = math.cos(al), math.sin(al); par.alpha[i] = al * RAD2DEG
            xyz.append(p); trig.append((ca, sa, cb, sb, cg, sg))
    par.nobs = n
    rm = np.zeros((n, 3, 3))
    for i, (ca, sa, cb, sb, cg, sg) in enumerate(trig):                   # :207-217
        rm[i] = [[ca * cb * cg - sa * sg, sa * cb * cg + ca * sg, -sb * cg],
                 [-ca * cb * sg - sa * cg, -sa * cb * sg + ca * cg, sb * sg],
                 [ca * sb, sa * sb, cb]]
    if not (fin(par.dxim) and fin(par.dyim)):                             # :244-276
        if par.geometry == "sphere":
            par.dxim = math.asin(par.rmax / par.distance) / (par.nxim / 2.0) * RAD2DEG
            par.dyim = math.asin(par.rmax / par.distance) / (par.nyim / 2.0) * RAD2DEG
        else:
            vx = [1, 1, 1, -1, -1, -1, 1, -1]; vy = [1, 1, -1, 1, -1, 1, -1, -1]; vz = [1, -1, 1, 1, 1, -1, -1, -1]
            mx = my = -999.0
            for i in range(n):
                for iv in range(8):
                    pv = np.array(xyz[i]) - np.array([vx[iv] * par.xmax, vy[iv] * par.ymax, vz[iv] * par.zmax])
                    k = rm[i] @ pv
                    mx = max(mx, abs(math.atan2(-k[0], k[2]))); my = max(my, abs(math.atan2(-k[1], k[2])))
            if par.nxim == par.nyim:
                par.dxim = max(mx, my) / (par.nxim / 2.0) * RAD2DEG; par.dyim = max(mx, my) / (par.nyim / 2.0) * RAD2DEG
            else:
                if not fin(par.dxim):
                    par.dxim = mx / (par.nxim / 2.0) * RAD2DEG
                if not fin(par.dyim):
                    par.dyim = my / (par.nyim / 2.0) * RAD2DEG
    return Observers(nobs=n, xyz=np.array(xyz), rmatrix=rm, dxim=par.dxim, dyim=par.dyim, nxim=par.nxim, nyim=par.nyim,
                     steradian_pix=par.dxim * par.dyim * DEG2RAD ** 2)


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
        """output_normalize_outside (output_sum_rect.f90:151-487): physical units for Jout/Jin/Jabs/Jmu and peel cubes"""
        t = t or self.ctx.tallies()
        par, g = self.par, self.grid
        nph = float(nphotons or t["nphotons_done"] or self.nphotons_run)     # photons the device(s) finished (cumulative, summed by output_reduce)
        out = dict(t)
        unit = g.dwave if par.distance2cm != 1.0 else g.dxfreq            # par%intensity_unit default, setup.f90:491-497
        if par.xy_periodic:                                               # slab, two faces: :174-190
            area = 2.0
        elif par.geometry == "sphere":                                    # :191-196
            area = FOURPI * g.rmax ** 2 * par.distance2cm ** 2
        else:
            area = (g.xmax * g.ymax + g.ymax * g.zmax + g.zmax * g.xmax) * 8.0 * par.distance2cm ** 2
            if self.amr is not None:                                      # the six faces of the octree box (output_sum_rect.f90:808-813)
                area = 6.0 * self.amr.L_box ** 2 * par.distance2cm ** 2
        scale = nph * unit * TWOPI * area
        for k in ("Jout", "Jin", "Jabs"):
            out[k] = t[k] / scale
        if "Jmu" in t:
            out["Jmu"] = t["Jmu"] * par.nmu / scale                       # :188-190, :206-208
        if self.obs.nobs:
            sc3 = nph * self.obs.steradian_pix * unit * par.distance2cm ** 2                    # :428-430
            sc2 = nph * self.obs.steradian_pix * par.distance2cm ** 2
            for k in ("scatt", "direc", "I", "Q", "U", "V", "direc0", "scatt_dust"):
                if k in t:
                    out[k] = t[k] / sc3
                if k + "_2D" in t:
                    out[k + "_2D"] = t[k + "_2D"] / sc2
        if self.jpa.nbins and self.amr is not None:                       # octree: volume-weighted, output_sum_heal.f90:476-590
            vol = np.where(self.amr.vol > 0, self.amr.vol, np.inf) * par.distance2cm ** 2
            if par.xy_periodic and par.geometry_JPa in (3, -1):           # slab, one photon per cm^2: the slab area replaces 1/d^2 (:874-1000)
                vol = np.where(self.amr.vol > 0, self.amr.vol, np.inf) / self.amr.L_box ** 2
            if "J1" in t:
                out["J1"] = t["J1"] / (FOURPI * vol[None, :] * nph * unit)
            for k in ("P1", "P1_new"):
                if k in t:
                    out[k] = t[k] / (vol * nph)
        elif self.jpa.nbins:                                              # :275-360 J(x) and P_alpha profiles
            dVol = g.dx * g.dy * g.dz * par.distance2cm ** 2
            cnt = np.where(self.jpa.ncount > 0, self.jpa.ncount, np.inf)
            geo_area = (g.xmax - g.xmin) * (g.ymax - g.ymin) * par.distance2cm ** 2 if par.xy_periodic else 1.0
            f = geo_area if (par.xy_periodic and par.geometry_JPa in (3, -1)) else 1.0
            if "J1" in t:
                out["J1"] = t["J1"] / cnt[None, :] * (f / (FOURPI * dVol * nph * unit))
            if "P1" in t:
                out["P1"] = t["P1"] / cnt * (f / (dVol * nph))
            if "P1_new" in t:
                out["P1_new"] = t["P1_new"] / cnt * (f / (dVol * nph))
        out["nscatt_tot"] = (t["nscatt_gas"] + t["nscatt_dust"]) / nph
        return out


def radial_profiles(par, g, out, io=0, use_2D=False, stokes=False):
    """make_radial_intensity / make_radial_stokes (output_sum_rect.f90:489-658) on the normalised tallies `out` of
    Simulation.output_normalize: columns r, I[, Q, U, V, pol] of the azimuthal means about the image centre"""
    unit = g.dwave if par.distance2cm != 1.0 else g.dxfreq
    nxi, nyi = par.nxim, par.nyim
    nr = (max(nxi, nyi) + 1) // 2
    r = (np.arange(1, nr + 1) - 0.5) / nr if nr % 2 == 0 else (np.arange(1, nr + 1) - 1.0) / (nr - 0.5)
    xx, yy = np.meshgrid(np.arange(1, nxi + 1) - (nxi + 1.0) / 2.0, np.arange(1, nyi + 1) - (nxi + 1.0) / 2.0, indexing="ij")
    rr = np.sqrt(xx ** 2 + yy ** 2)
    ir = (np.floor(rr) if nr % 2 == 0 else np.floor(rr + 0.5)).astype(int)          # 0-based bin
    ok = ir < nr
    cnt = np.bincount(ir[ok], minlength=nr).astype(float)

    def img(k):
        return out[k + "_2D"][:, :, io] if use_2D else out[k][:, :, :, io].sum(axis=0) * unit

    def mean(a):
        return np.where(cnt > 0, np.bincount(ir[ok], weights=a[ok], minlength=nr) / np.maximum(cnt, 1.0), 0.0)

    if not stokes:
        return np.column_stack([r, mean(img("scatt") + img("direc"))])
    with np.errstate(invalid="ignore", divide="ignore"):
        cosp = np.where(rr == 0.0, 1.0, yy / rr); sinp = np.where(rr == 0.0, 0.0, -xx / rr)
    c2, s2 = 2.0 * cosp ** 2 - 1.0, 2.0 * cosp * sinp
    if use_2D:
        I, Q, U, V = img("I"), img("Q") * c2 + img("U") * s2, -img("Q") * s2 + img("U") * c2, img("V")
    else:
        q3, u3 = out["Q"][:, :, :, io], out["U"][:, :, :, io]
        I, V = img("I"), img("V")
        Q = (q3 * c2[None] + u3 * s2[None]).sum(axis=0) * unit; U = (-q3 * s2[None] + u3 * c2[None]).sum(axis=0) * unit
    mI, mQ, mU, mV = mean(I), mean(Q), mean(U), mean(V)
    with np.errstate(invalid="ignore", divide="ignore"):
        pol = np.where(mI > 0.0, np.sqrt(mQ ** 2 + mU ** 2) / mI, 0.0)
    return np.column_stack([r, mI, mQ, mU, mV, pol])


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
