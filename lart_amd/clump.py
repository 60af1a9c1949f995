"""Clump medium, host side: the scalar part of init_clumps / grid_create_clump (clump_mod.f90:646-891,
grid_mod_clump.f90:33-131) in numpy; the per-clump loops (RSA placement, CSR grid) run in liblart_gpu's host helpers
(csrc/lart_host_clumps.cpp).  Uniform clump properties, non-overlapping clumps (the reference's default path)."""
import ctypes as C
import math
from dataclasses import dataclass

import numpy as np

from . import capi
from .host import FOURPI, UM2KM, LartInputError, vtherm_total, voigt_center


@dataclass
class Clumps:
    N: int = 0
    sphere_R: float = 0.0
    r_min: float = 0.0
    radius_max: float = 0.0
    rhokap_ref: float = 0.0
    voigt_a_ref: float = 0.0
    Dfreq_ref: float = 0.0
    vtherm_ref: float = 0.0
    temperature_ref: float = 0.0
    fvol: float = 0.0
    fcov: float = 0.0
    rsa_acceptance: float = 0.0
    x: np.ndarray = None
    y: np.ndarray = None
    z: np.ndarray = None
    vx: np.ndarray = None
    vy: np.ndarray = None
    vz: np.ndarray = None
    radius: np.ndarray = None
    rhokap: np.ndarray = None
    rhokapD: np.ndarray = None
    voigt_a: np.ndarray = None
    Dfreq: np.ndarray = None
    vtherm: np.ndarray = None
    cgx: int = 0
    cg_min: float = 0.0
    cg_d: float = 0.0
    cg_start: np.ndarray = None
    cg_list: np.ndarray = None
    has_overlap: bool = False                              # clump_mod.f90:88, check_has_overlap


def _hchk(rc):
    if rc != 0:
        raise LartInputError(capi.lib().lart_host_last_error().decode())


def build_csr(cl):
    """build_clump_csr: clump_mod.f90:1267-1342"""
    L = capi.lib()
    cgx, cmin, cd = C.c_int32(), C.c_double(), C.c_double()
    _hchk(L.lart_host_clumps_csr_dims(cl.N, cl.sphere_R, cl.radius_max, C.byref(cgx), C.byref(cmin), C.byref(cd)))
    cl.cgx, cl.cg_min, cl.cg_d = cgx.value, cmin.value, cd.value
    cl.cg_start = np.zeros(cl.cgx ** 3 + 1, dtype=np.int32)
    nreg = C.c_int64()
    args = (cl.N, capi._p(cl.x), capi._p(cl.y), capi._p(cl.z), capi._p(cl.radius), cl.cgx, cl.cg_min, cl.cg_d, capi._p(cl.cg_start))
    _hchk(L.lart_host_clumps_build_csr(*args, None, C.byref(nreg)))
    cl.cg_list = np.zeros(max(1, nreg.value), dtype=np.int32)
    _hchk(L.lart_host_clumps_build_csr(*args, capi._p(cl.cg_list), C.byref(nreg)))
    cl.cg_list = cl.cg_list[:nreg.value]
    return cl


def assign_clump_velocities(par, cl):
    """assign_clump_velocities_from_type: clump_mod.f90:1153-1264 (the fields this path supports)"""
    x, y, z = cl.x, cl.y, cl.z
    rr = np.sqrt(x * x + y * y + z * z)
    vx = np.zeros_like(x); vy = np.zeros_like(x); vz = np.zeros_like(x)
    vt = par.velocity_type
    if vt == "hubble":
        vx, vy, vz = par.Vexp * x / cl.sphere_R, par.Vexp * y / cl.sphere_R, par.Vexp * z / cl.sphere_R
    elif vt == "constant_radial":
        ok = rr > 0
        vx[ok], vy[ok], vz[ok] = par.Vexp * x[ok] / rr[ok], par.Vexp * y[ok] / rr[ok], par.Vexp * z[ok] / rr[ok]
    elif vt == "parallel_velocity":
        vx[:], vy[:], vz[:] = par.Vx, par.Vy, par.Vz
    elif vt == "rotating_solid_body":
        vx, vy = -par.Vrot * y / cl.sphere_R, par.Vrot * x / cl.sphere_R
    elif vt == "rotating_galaxy_halo":
        rc = np.sqrt(x * x + y * y)
        inner = (rc > 0) & (rc < par.rinner); outer = rc >= par.rinner
        vx[inner], vy[inner] = -par.Vrot * y[inner] / par.rinner, par.Vrot * x[inner] / par.rinner
        vx[outer], vy[outer] = -par.Vrot * y[outer] / rc[outer], par.Vrot * x[outer] / rc[outer]
    cl.vx = cl.vx + vx / cl.vtherm; cl.vy = cl.vy + vy / cl.vtherm; cl.vz = cl.vz + vz / cl.vtherm


def init_clumps(par, line, R_sphere):
    """init_clumps, uniform path: clump_mod.f90:646-891"""
    cl = Clumps(sphere_R=float(R_sphere))
    if par.clump_radius <= 0.0:
        raise LartInputError("ERROR: clump_radius must be > 0")
    cl.r_min = max(0.0, par.rmin)
    if cl.r_min >= R_sphere:
        raise LartInputError("ERROR: par%rmin must be < par%rmax for clump placement")
    cl.radius_max = rc = par.clump_radius
    temp_cl = par.temperature if par.clump_temperature < 0.0 else par.clump_temperature
    vth = vtherm_total(line, temp_cl)
    cl.Dfreq_ref = vth / (line.wavelength0 * UM2KM); cl.vtherm_ref = vth; cl.temperature_ref = temp_cl
    cl.voigt_a_ref = (line.damping / FOURPI) / cl.Dfreq_ref
    R, r0 = R_sphere, cl.r_min
    shell2 = R * R + R * r0 + r0 * r0
    nint = lambda v: int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))
    if par.clump_N_clumps > 0.0:
        cl.N = int(par.clump_N_clumps)
    elif par.clump_f_vol > 0.0:
        cl.N = nint(par.clump_f_vol * (R ** 3 - r0 ** 3) / rc ** 3)
    elif par.clump_f_cov > 0.0:
        cl.N = nint((4.0 / 3.0) * par.clump_f_cov * shell2 / (rc * rc))
    else:
        raise LartInputError("ERROR: specify clump_N_clumps, clump_f_vol, or clump_f_cov")
    cl.N = max(cl.N, 1)
    tiny = 2.2250738585072014e-308
    cl.fvol = cl.N * rc ** 3 / max(R ** 3 - r0 ** 3, tiny)
    cl.fcov = 0.75 * cl.N * rc * rc / max(shell2, tiny)
    H0 = float(voigt_center(np.float64(cl.voigt_a_ref)))
    if par.clump_tau0 > 0.0:
        cl.rhokap_ref = par.clump_tau0 / (H0 * par.clump_radius)
    elif par.clump_NHI > 0.0:
        cl.rhokap_ref = par.clump_NHI * line.cross0 / (cl.Dfreq_ref * par.clump_radius)
    elif par.clump_nH > 0.0:
        if par.distance2cm <= 0.0:
            raise LartInputError("ERROR: clump_nH requires distance_unit")
        cl.rhokap_ref = par.clump_nH * line.cross0 * par.distance2cm / cl.Dfreq_ref
    elif par.taumax > 0.0 or par.N_gasmax > 0.0:
        GF = cl.N * rc ** 3 / max(shell2, tiny)
        cl.rhokap_ref = par.taumax / (GF * H0) if par.taumax > 0.0 else par.N_gasmax * line.cross0 / (GF * cl.Dfreq_ref)
    else:
        raise LartInputError("ERROR: specify clump_tau0, clump_NHI, clump_nH, taumax, or N_HImax")
    n = cl.N
    cl.x, cl.y, cl.z = np.zeros(n), np.zeros(n), np.zeros(n)
    cl.vx, cl.vy, cl.vz = np.zeros(n), np.zeros(n), np.zeros(n)
    cl.radius = np.full(n, rc); cl.rhokap = np.full(n, cl.rhokap_ref); cl.voigt_a = np.full(n, cl.voigt_a_ref)
    cl.Dfreq = np.full(n, cl.Dfreq_ref); cl.vtherm = np.full(n, cl.vtherm_ref)
    cl.rhokapD = np.full(n, cl.rhokap_ref * par.cext_dust * par.DGR * cl.Dfreq_ref / line.cross0) if par.DGR > 0.0 else None
    acc = C.c_double()
    sig = par.clump_sigma_v / cl.vtherm_ref if par.clump_sigma_v > 0.0 else 0.0
    _hchk(capi.lib().lart_host_clumps_generate(n, cl.sphere_R, cl.r_min, rc, int(par.clump_fully_inside) | (2 if par.clump_allow_overlap else 0), sig, int(par.iseed),
                                               capi._p(cl.x), capi._p(cl.y), capi._p(cl.z), capi._p(cl.vx), capi._p(cl.vy), capi._p(cl.vz), C.byref(acc)))
    cl.rsa_acceptance = acc.value
    if par.velocity_type:
        assign_clump_velocities(par, cl)
    build_csr(cl)
    if par.clump_allow_overlap:                            # clump_mod.f90:889-891: only then is the population searched for overlapping pairs
        has = C.c_int32()
        _hchk(capi.lib().lart_host_clumps_has_overlap(cl.N, capi._p(cl.x), capi._p(cl.y), capi._p(cl.z), capi._p(cl.radius), cl.cgx, cl.cg_min, cl.cg_d,
                                                      capi._p(cl.cg_start), capi._p(cl.cg_list), C.byref(has)))
        cl.has_overlap = bool(has.value)
        print(" Overlap check: overlapping clumps detected — using overlap-aware raytrace." if cl.has_overlap else " Overlap check: no overlaps found — using standard raytrace.")
    return cl


def compute_clump_scalars(par, line, cl):
    """compute_clump_scalars: clump_mod.f90:2316-2378 (sequential sums in clump order, as the reference loops)"""
    shell2 = cl.sphere_R ** 2 + cl.sphere_R * cl.r_min + cl.r_min ** 2
    H0 = voigt_center(cl.voigt_a)
    r3 = cl.radius ** 3
    di2 = np.maximum(cl.x ** 2 + cl.y ** 2 + cl.z ** 2, cl.radius ** 2)
    wh, wm = r3 / shell2, r3 / (3.0 * di2)
    seq = lambda v: float(np.cumsum(v)[-1])               # left-to-right double sum
    par.tauhomo = seq(cl.rhokap * H0 * wh); par.taumax = seq(cl.rhokap * H0 * wm)
    par.N_gashomo = seq(cl.rhokap * cl.Dfreq * wh / line.cross0); par.N_gasmax = seq(cl.rhokap * cl.Dfreq * wm / line.cross0)


def grid_create_clump(par, line):
    """grid_create_clump: grid_mod_clump.f90:33-131.  Returns (Clumps, Grid): the Cartesian grid only carries the spectral
    axis and the uniform background record (rhokap = 0, Dfreq = cl_Dfreq_ref, voigt_a = cl_voigt_a_ref, v = 0)."""
    from . import host
    if par.rmax <= 0.0:
        raise LartInputError("ERROR: par%rmax must be > 0 for clump medium")
    par.xmax = par.ymax = par.zmax = par.rmax
    par.nx, par.ny, par.nz = max(par.nx, 1) if par.nx >= 1 else 11, par.ny if par.ny >= 1 else 11, par.nz if par.nz >= 1 else 11
    par.xyz_symmetry = False; par.xy_periodic = False
    cl = init_clumps(par, line, par.rmax)
    compute_clump_scalars(par, line, cl)
    vt, par.velocity_type = par.velocity_type, ""          # the background grid is static; velocities live on the clumps
    g = host.grid_create(par, line)
    par.velocity_type = vt
    g.rhokap[...] = 0.0; g.Dfreq[...] = cl.Dfreq_ref; g.voigt_a[...] = cl.voigt_a_ref
    g.vfx[...] = 0.0; g.vfy[...] = 0.0; g.vfz[...] = 0.0
    if g.rhokapD is not None:
        g.rhokapD[...] = 0.0
    g.Dfreq_ref = cl.Dfreq_ref; g.voigt_amean = cl.voigt_a_ref
    return cl, g
