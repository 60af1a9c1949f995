"""The reference's namelist surface: params_type (define.f90:209-544) as a dataclass with the reference's defaults, the fields this path refuses
to change from their defaults, and the reader of `&parameters par%name = value ... /` groups (setup.f90:27-40).  A schema, no arithmetic: what is
derived from these fields is computed by the library's host helpers (lart_amd/hostlib.py)."""
import re
from dataclasses import dataclass, field
import numpy as np

NAN = float("nan")
MAX_OBSERVERS = 16


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


