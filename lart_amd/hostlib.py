"""ctypes binding of the host set-up helpers of liblart_gpu.so (include/lart_gpu.h, "Host set-up helpers"; lart_amd/csrc/lart_host_setup.cpp).

The arithmetic of the reference's set-up routines for the media this path supports lives ONCE, in the library; this module and the Fortran host
(lart_amd/fhost/lart_gpu_c.f90) are its two callers.  Struct members mirror the header one for one."""
import ctypes as C
import numpy as np

from . import capi

_d, _i32, _i64, _vp = C.c_double, C.c_int32, C.c_int64, C.c_void_p
MAX_OBSERVERS = 16
VELOCITY_TYPES = {"": 0, "hubble": 1, "constant_radial": 2, "parallel_velocity": 3, "rotating_solid_body": 4, "rotating_galaxy_halo": 5,
                  "power_law": 6, "linear_decelerate": 7, "ssh": 8}


class HostRules(C.Structure):
    _fields_ = [(n, _i32) for n in ("nx", "ny", "nz", "nr", "nside", "nxim", "nyim", "nmu", "geometry_JPa", "distance_unit", "intensity_unit", "notes",
                                    "xyz_symmetry", "xy_symmetry", "xy_periodic", "use_clump_medium", "use_amr_grid", "use_stokes", "sphere", "save_all",
                                    "save_peeloff", "save_peeloff_2D", "save_peeloff_3D", "save_Jabs", "save_Jmu", "core_skip", "core_skip_global", "pad_")] + \
        [(n, _d) for n in ("tau0", "taumax", "N_HI", "N_HImax", "N_HIhomo", "N_gasmax", "N_gashomo", "cext_dust", "DGR", "mu_min", "dmu",
                           "rmax", "xmax", "ymax", "zmax", "source_rmax", "distance2cm")]


DISTANCE_UNITS = {"": 0, "kpc": 1, "pc": 2, "au": 3}


class HostLine(C.Structure):
    _fields_ = [(n, _i32) for n in ("line_type", "fine_structure", "include_deuterium", "pad_")] + \
        [(n, _d) for n in ("wavelength0", "damping", "cross0", "vtherm1", "mass_amu", "g_recoil0", "DnuHK_Hz", "E1", "E2", "E3",
                           "delta_nu_HD_Hz", "ratio_Dfreq_HD", "ratio_voigta_HD", "nD_over_nH", "g_recoil0_D", "P_conv")] + \
        [("E_conv", _d * 3), ("wavelength0_Ha", _d), ("line_id", C.c_char * 32), ("ion_id", C.c_char * 8)]


class HostH2(C.Structure):
    _fields_ = [("n", _i32), ("pad_", _i32), ("Dfreq_Hz", _d)] + [(n, _d * 4) for n in ("dv_kms", "dnu_Hz", "strength", "a_damp", "p_scat", "pop", "f_osc")]


class HostMedium(C.Structure):
    _fields_ = [(n, _i32) for n in ("nx", "ny", "nz", "xyz_symmetry", "xy_symmetry", "velocity_type", "pad0_", "pad1_")] + \
        [(n, _d) for n in ("xmax", "ymax", "zmax", "rmin", "rmax", "cone_opening", "vtherm", "wavelength0", "damping", "cross0", "dust_opacity",
                           "taumax", "tauhomo", "N_gasmax", "N_gashomo", "Vexp", "Vx", "Vy", "Vz", "Vrot", "rinner", "velocity_alpha", "rpeak", "Vpeak", "DeltaV",
                           "dx", "dy", "dz", "xmin", "ymin", "zmin", "Dfreq_ref", "voigt_amean", "taupole_out", "tauhomo_out", "N_gaspole_out", "N_gashomo_out")] + \
        [(n, _i32) for n in ("i0", "j0", "k0", "pad2_")]


class HostFreq(C.Structure):
    _fields_ = [(n, _i32) for n in ("nxfreq", "nwavelength", "nvelocity", "continuum", "line_type", "nxfreq_Ha", "core_skip_global", "pad_")] + \
        [(n, _d) for n in ("wavelength_min", "wavelength_max", "velocity_min", "velocity_max", "xfreq_min", "xfreq_max", "vtherm", "wavelength0", "wavelength0_Ha",
                           "Dfreq_ref", "DnuHK_Hz", "voigt_amean", "tauhomo", "taumax", "Vexp", "xfreq_max_Ha", "cells_per_radius",
                           "dxfreq", "dwave", "atau3", "xfreq_min_Ha", "dxfreq_Ha", "dwave_Ha", "xcrit", "xcrit2")]


class HostObs(C.Structure):
    _fields_ = [(n, _i32) for n in ("nside", "sphere", "nxim", "nyim", "nobs", "npix")] + \
        [(n, _d) for n in ("xmax", "ymax", "zmax", "rmax", "distance", "dxim", "dyim", "steradian_pix")]


class HostClumpsIn(C.Structure):
    _fields_ = [(n, _d) for n in ("sphere_R", "rmin", "clump_radius", "N_clumps_in", "f_vol_in", "f_cov_in", "clump_tau0", "clump_NHI", "clump_nH", "taumax", "N_gasmax",
                                  "distance2cm", "vtherm", "wavelength0", "damping", "cross0")] + [("N", _i64)] + \
        [(n, _d) for n in ("r_min", "fvol", "fcov", "Dfreq_ref", "voigt_a_ref", "rhokap_ref")]


class HostNorm(C.Structure):
    _fields_ = [("xy_periodic", _i32), ("sphere", _i32)] + \
        [(n, _d) for n in ("nphotons", "bin_unit", "bin_unit_Ha", "distance2cm", "rmax", "xmax", "ymax", "zmax", "amr_L_box", "steradian_pix",
                           "area", "scale_J", "scale_J_Ha", "scale_cube", "scale_image", "scale_cube_Ha")]


SYMBOLS = ["lart_host_setup_last_error", "lart_host_input_rules", "lart_host_line_setup", "lart_host_vtherm_total", "lart_host_h2_init", "lart_host_car_medium", "lart_host_freq_grid",
           "lart_host_freq_axis", "lart_host_jpa_car", "lart_host_observers", "lart_host_clump_population", "lart_host_clump_scalars", "lart_host_clump_velocities",
           "lart_host_alias_setup", "lart_host_mueller_setup", "lart_host_norm_scales"]
_ready = False


def lib():
    global _ready
    L = capi.lib()
    if not _ready:
        L.lart_host_setup_last_error.restype = C.c_char_p
        L.lart_host_input_rules.argtypes = [C.POINTER(HostRules)]
        L.lart_host_line_setup.argtypes = [C.c_char_p, _i32, _i32, _d, C.POINTER(HostLine)]
        L.lart_host_vtherm_total.restype = _d; L.lart_host_vtherm_total.argtypes = [_d, _d, _d]
        L.lart_host_h2_init.argtypes = [C.c_char_p, _d, _d, _d, _d, _d, _d, _i32, C.POINTER(HostH2)]
        L.lart_host_car_medium.argtypes = [C.POINTER(HostMedium)] + [_vp] * 10
        L.lart_host_freq_grid.argtypes = [C.POINTER(HostFreq)]
        L.lart_host_freq_axis.argtypes = [_i32, _d, _d, _d, _d, _vp, _vp, _vp]
        L.lart_host_jpa_car.argtypes = [_i32] * 5 + [_vp] * 3 + [_d] * 3 + [_vp] * 4
        L.lart_host_observers.argtypes = [C.POINTER(HostObs)] + [_vp] * 8
        L.lart_host_clump_population.argtypes = [C.POINTER(HostClumpsIn)]
        L.lart_host_clump_scalars.argtypes = [_i64, _d, _d, _d] + [_vp] * 8
        L.lart_host_clump_velocities.argtypes = [_i64, _i32] + [_d] * 7 + [_vp] * 7
        L.lart_host_alias_setup.argtypes = [_i32, _vp, _vp]
        L.lart_host_mueller_setup.argtypes = [_i32] + [_vp] * 7
        L.lart_host_norm_scales.argtypes = [C.POINTER(HostNorm)]
        _ready = True
    return L


def check(rc, exc):
    """raise `exc` with the library's message for a non-zero return code"""
    if rc != 0:
        raise exc(lib().lart_host_setup_last_error().decode())


def p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def nan_array(values):
    return np.array([float(v) for v in values], dtype=np.float64)
