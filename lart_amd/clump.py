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
    """assign_clump_velocities_from_type (clump_mod.f90:1153-1264, the fields this path supports) through lart_host_clump_velocities"""
    from . import hostlib
    vt = hostlib.VELOCITY_TYPES.get(par.velocity_type, 0)
    hostlib.check(hostlib.lib().lart_host_clump_velocities(cl.N, vt, cl.sphere_R, par.Vexp, par.Vx, par.Vy, par.Vz, par.Vrot, par.rinner,
                                                             *[hostlib.p(a) for a in (cl.x, cl.y, cl.z, cl.vtherm, cl.vx, cl.vy, cl.vz)]), LartInputError)


def init_clumps(par, line, R_sphere):
    """init_clumps, uniform path: clump_mod.f90:646-891"""
    from . import hostlib
    cl = Clumps(sphere_R=float(R_sphere))
    temp_cl = par.temperature if par.clump_temperature < 0.0 else par.clump_temperature
    c = hostlib.HostClumpsIn(sphere_R=float(R_sphere), rmin=par.rmin, clump_radius=par.clump_radius, N_clumps_in=par.clump_N_clumps, f_vol_in=par.clump_f_vol,
                             f_cov_in=par.clump_f_cov, clump_tau0=par.clump_tau0, clump_NHI=par.clump_NHI, clump_nH=par.clump_nH, taumax=par.taumax, N_gasmax=par.N_gasmax,
                             distance2cm=par.distance2cm, vtherm=vtherm_total(line, temp_cl), wavelength0=line.wavelength0, damping=line.damping, cross0=line.cross0)
    hostlib.check(hostlib.lib().lart_host_clump_population(C.byref(c)), LartInputError)      # population size, f_vol / f_cov, the opacity of one clump (:646-760)
    cl.r_min, cl.radius_max, rc = c.r_min, par.clump_radius, par.clump_radius
    cl.Dfreq_ref, cl.vtherm_ref, cl.temperature_ref, cl.voigt_a_ref = c.Dfreq_ref, c.vtherm, temp_cl, c.voigt_a_ref
    cl.N, cl.fvol, cl.fcov, cl.rhokap_ref = int(c.N), c.fvol, c.fcov, c.rhokap_ref
    n = cl.N
    cl.x, cl.y, cl.z = np.zeros(n), np.zeros(n), np.zeros(n)
    cl.vx, cl.vy, cl.vz = np.zeros(n), np.zeros(n), np.zeros(n)
    cl.radius = np.full(n, rc); cl.rhokap = np.full(n, cl.rhokap_ref); cl.voigt_a = np.full(n, cl.voigt_a_ref)
    cl.Dfreq = np.full(n, cl.Dfreq_ref); cl.vtherm = np.full(n, cl.vtherm_ref)
    cl.rhokapD = np.full(n, cl.rhokap_ref * par.cext_dust * par.DGR * cl.Dfreq_ref / line.cross0) if par.DGR > 0.0 else None
    acc = C.c_double()
    sig = par.clump_sigma_v / cl.vtherm_ref if par.clump_sigma_v > 0.0 else 0.0
    _hchk(capi.lib().lart_host_clumps_generate_cone(n, cl.sphere_R, cl.r_min, rc, int(par.clump_fully_inside) | (2 if par.clump_allow_overlap else 0), float(par.cone_opening), sig, int(par.iseed),
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


# ---------------------------------------------------------------------------------------------------------------------------------
# clump population files: write_clumps_info / read_clumps_info (clump_mod.f90:1779-1961, 2040-2312).  One table section with float32
# columns X, Y, Z [code units], VX, VY, VZ [km/s] and -- only when they vary by more than 1e-3 of their mean -- R_CLUMP, RHOKAP, TEMP;
# the scalars as keywords of that section.  FITS (.fits, .fits.gz: empty primary HDU + BINTABLE) or HDF5 (.h5: /section_001 with one
# dataset per column and the keywords as attributes, hdf5io_mod.f90:8-13).
# ---------------------------------------------------------------------------------------------------------------------------------
_CL_KEYS = ("N_CLUMPS", "SPHERE_R", "RMIN", "CL_RAD", "F_VOL", "F_COV", "TAU0", "SIGMA_V", "TEMP_CL", "RHOKAP", "CL_DFREQ", "VTHERM", "VOIGT_A", "RMAX",
            "IN_FCOV", "IN_FVOL", "IN_NCL", "IN_NHI", "IN_NH", "IN_TEMP", "DISTUNIT", "DIST_CM")


def clumps_file_tables(par, line, cl):
    """the columns and keywords write_clumps_info puts out for a population (:1880-1956)"""
    T = getattr(cl, "temperature", None)
    if T is None:
        T = np.full(cl.N, cl.temperature_ref)
    cols = [("X", cl.x), ("Y", cl.y), ("Z", cl.z), ("VX", cl.vx * cl.vtherm), ("VY", cl.vy * cl.vtherm), ("VZ", cl.vz * cl.vtherm)]
    tiny = 2.2250738585072014e-308
    for name, a in (("R_CLUMP", cl.radius), ("RHOKAP", cl.rhokap), ("TEMP", T)):
        if a.max() - a.min() > 1.0e-3 * max(abs(a.sum() / max(cl.N, 1)), tiny):
            cols.append((name, a))
    R, r0 = cl.sphere_R, cl.r_min
    if getattr(cl, "from_file", False):
        fvol = float(np.cumsum(cl.radius ** 3)[-1]) / max(R ** 3 - r0 ** 3, tiny)
        fcov = 0.75 * float(np.cumsum(cl.radius ** 2)[-1]) / max(R * R + R * r0 + r0 * r0, tiny)
    else:
        fvol = cl.N * cl.radius_max ** 3 / max(R ** 3 - r0 ** 3, tiny); fcov = 0.75 * cl.N * cl.radius_max ** 2 / max(R * R + R * r0 + r0 * r0, tiny)
    keys = dict(N_CLUMPS=int(cl.N), SPHERE_R=R, RMIN=r0, CL_RAD=cl.radius_max, F_VOL=fvol, F_COV=fcov, TAU0=par.clump_tau0, SIGMA_V=par.clump_sigma_v,
                TEMP_CL=cl.temperature_ref, RHOKAP=cl.rhokap_ref, CL_DFREQ=cl.Dfreq_ref, VTHERM=cl.vtherm_ref, VOIGT_A=cl.voigt_a_ref, RMAX=par.rmax,
                IN_FCOV=par.clump_f_cov, IN_FVOL=par.clump_f_vol, IN_NCL=par.clump_N_clumps, IN_NHI=par.clump_NHI, IN_NH=par.clump_nH,
                IN_TEMP=par.clump_temperature, DISTUNIT=par.distance_unit, DIST_CM=par.distance2cm)
    return cols, keys


def write_table_file(path, cols, keys):
    """one table section of float32 columns + keywords, FITS or HDF5 by the extension (iofile_mod.f90: io_detect_format)"""
    from .amr import _fits_card, _h5
    n = len(cols[0][1])
    if path.endswith(".h5") or path.endswith(".hdf5"):
        C, lib, T = _h5()
        lib.H5Tcopy.restype = C.c_int64; lib.H5Tcopy.argtypes = [C.c_int64]; lib.H5Tset_size.argtypes = [C.c_int64, C.c_size_t]; lib.H5Tclose.argtypes = [C.c_int64]
        tf32 = C.c_int64.in_dll(lib, "H5T_NATIVE_FLOAT_g").value; tstr = C.c_int64.in_dll(lib, "H5T_C_S1_g").value
        ti64 = C.c_int64.in_dll(lib, "H5T_NATIVE_LLONG_g").value
        f = lib.H5Fcreate(path.encode(), 2, 0, 0); g = lib.H5Gcreate2(f, b"section_001", 0, 0, 0)
        dims = (C.c_uint64 * 1)(n); sp = lib.H5Screate_simple(1, dims, None)
        for name, a in cols:
            a32 = np.ascontiguousarray(a, dtype=np.float32)
            ds = lib.H5Dcreate2(g, name.encode(), tf32, sp, 0, 0, 0); lib.H5Dwrite(ds, tf32, 0, 0, 0, a32.ctypes.data_as(C.c_void_p)); lib.H5Dclose(ds)
        lib.H5Sclose(sp); sc = lib.H5Screate(0)
        for k, v in keys.items():
            if isinstance(v, str):
                t = lib.H5Tcopy(tstr); lib.H5Tset_size(t, max(len(v), 1)); a = lib.H5Acreate2(g, k.encode(), t, sc, 0, 0)
                lib.H5Awrite(a, t, C.c_char_p(v.encode() or b" ")); lib.H5Aclose(a); lib.H5Tclose(t)
            elif isinstance(v, (int, np.integer)):
                a = lib.H5Acreate2(g, k.encode(), ti64, sc, 0, 0); lib.H5Awrite(a, ti64, C.byref(C.c_longlong(int(v)))); lib.H5Aclose(a)
            else:
                a = lib.H5Acreate2(g, k.encode(), T["DOUBLE"], sc, 0, 0); lib.H5Awrite(a, T["DOUBLE"], C.byref(C.c_double(float(v)))); lib.H5Aclose(a)
        lib.H5Sclose(sc); lib.H5Gclose(g); lib.H5Fclose(f)
        return
    import gzip
    dt = np.dtype([(name, ">f4") for name, _ in cols]); tab = np.zeros(n, dtype=dt)
    for name, a in cols:
        tab[name] = a
    def block(cards):
        h = "".join(cards) + "END".ljust(80)
        return (h + " " * (-len(h) % 2880)).encode("ascii")
    prim = block([_fits_card("SIMPLE", True), _fits_card("BITPIX", 8), _fits_card("NAXIS", 0), _fits_card("EXTEND", True)])
    cards = [_fits_card("XTENSION", "BINTABLE"), _fits_card("BITPIX", 8), _fits_card("NAXIS", 2), _fits_card("NAXIS1", dt.itemsize), _fits_card("NAXIS2", n),
             _fits_card("PCOUNT", 0), _fits_card("GCOUNT", 1), _fits_card("TFIELDS", len(cols))]
    for i, (name, _) in enumerate(cols):
        cards += [_fits_card("TTYPE%d" % (i + 1), name), _fits_card("TFORM%d" % (i + 1), "1E")]
    cards += [_fits_card(k, v) for k, v in keys.items()]
    body = tab.tobytes(); body += b"\0" * (-len(body) % 2880)
    with (gzip.open(path, "wb") if path.endswith(".gz") else open(path, "wb")) as f:
        f.write(prim + block(cards) + body)


def read_table_file(path):
    """-> (columns {NAME: float64 array}, keywords {NAME: value}) of the first table section of a FITS / HDF5 file"""
    from .amr import _h5
    if path.endswith(".h5") or path.endswith(".hdf5"):
        C, lib, T = _h5()
        f = lib.H5Fopen(path.encode(), 0, 0)
        if f < 0:
            raise LartInputError("ERROR: cannot open clump_input_file: " + path)
        lib.H5Literate.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        names = []
        CB = C.CFUNCTYPE(C.c_int, C.c_int64, C.c_char_p, C.c_void_p, C.c_void_p)
        cb = CB(lambda g, name, info, u: (names.append(name.decode()), 0)[1])
        lib.H5Literate(f, 0, 0, None, cb, None)
        cols, keys = {}, {}
        for gname in sorted(names):
            g = lib.H5Gopen2(f, gname.encode(), 0)
            if g < 0:
                continue
            if lib.H5Lexists(g, b"X", 0) > 0:
                dn = []
                cb2 = CB(lambda gg, name, info, u: (dn.append(name.decode()), 0)[1]); lib.H5Literate(g, 0, 0, None, cb2, None)
                for d in dn:
                    ds = lib.H5Dopen2(g, d.encode(), 0)
                    if ds < 0:
                        continue
                    sp = lib.H5Dget_space(ds); dims = (C.c_uint64 * 4)(); rk = lib.H5Sget_simple_extent_dims(sp, dims, None); lib.H5Sclose(sp)
                    if rk == 1:
                        a = np.zeros(dims[0]); lib.H5Dread(ds, T["DOUBLE"], 0, 0, 0, a.ctypes.data_as(C.c_void_p)); cols[d.upper()] = a
                    lib.H5Dclose(ds)
                lib.H5Aget_type.restype = C.c_int64; lib.H5Aget_type.argtypes = [C.c_int64]; lib.H5Tget_class.argtypes = [C.c_int64]; lib.H5Tget_size.restype = C.c_size_t
                lib.H5Tget_size.argtypes = [C.c_int64]; lib.H5Tclose.argtypes = [C.c_int64]
                for k in _CL_KEYS:
                    if lib.H5Aexists(g, k.encode()) > 0:
                        a = lib.H5Aopen(g, k.encode(), 0); t = lib.H5Aget_type(a)
                        if lib.H5Tget_class(t) == 3:                              # H5T_STRING
                            buf = C.create_string_buffer(lib.H5Tget_size(t) + 1); lib.H5Aread(a, t, buf); keys[k] = buf.value.decode().strip()
                        else:
                            v = C.c_double(); lib.H5Aread(a, T["DOUBLE"], C.byref(v)); keys[k] = v.value
                        lib.H5Tclose(t); lib.H5Aclose(a)
                lib.H5Gclose(g)
                break
            lib.H5Gclose(g)
        lib.H5Fclose(f)
        if not cols:
            raise LartInputError("ERROR: clump_input_file: cannot reach the table section of " + path)
        return cols, keys
    import gzip
    try:
        with (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")) as fh:
            raw = fh.read()
    except OSError:
        raise LartInputError("ERROR: cannot open clump_input_file: " + path)
    pos = 0
    def header():
        nonlocal pos
        kw = {}
        while True:
            blk = raw[pos:pos + 2880].decode("ascii"); pos += 2880
            if len(blk) < 2880:
                raise LartInputError("ERROR: clump_input_file: cannot reach binary-table HDU 2")
            for i in range(36):
                card = blk[80 * i:80 * i + 80]; key = card[:8].strip()
                if key == "END":
                    return kw
                if card[8:10] == "= ":
                    kw[key] = card[10:].split("/")[0].strip() if "'" not in card[10:] else card[10:].split("'")[1].strip()
    h0 = header()
    if int(h0.get("NAXIS", 0)) > 0:
        nb = abs(int(h0["BITPIX"])) // 8
        for i in range(int(h0["NAXIS"])):
            nb *= int(h0["NAXIS%d" % (i + 1)])
        pos += (nb + 2879) // 2880 * 2880
    h = header()
    if h.get("XTENSION", "") != "BINTABLE":
        raise LartInputError("ERROR: clump_input_file: cannot reach binary-table HDU 2")
    codes = {"D": ">f8", "E": ">f4", "J": ">i4", "K": ">i8", "I": ">i2"}
    nf = int(h["TFIELDS"])
    dt = np.dtype([("c%d" % i, codes[h["TFORM%d" % (i + 1)].lstrip("0123456789")[0]]) for i in range(nf)])
    tab = np.frombuffer(raw, dtype=dt, count=int(h["NAXIS2"]), offset=pos)
    cols = {h["TTYPE%d" % (i + 1)].strip().upper(): tab["c%d" % i].astype(np.float64) for i in range(nf)}
    keys = {}
    for k in _CL_KEYS:
        if k in h:
            try:
                keys[k] = float(h[k])
            except ValueError:
                keys[k] = h[k]
    return cols, keys


def write_clumps_info(path, par, line, cl):
    """write_clumps_info: clump_mod.f90:1779-1961"""
    cols, keys = clumps_file_tables(par, line, cl)
    write_table_file(path, cols, keys)
    print(" Clumps: clump columns saved (R_CLUMP/RHOKAP/TEMP) = " + " ".join("T" if any(c[0] == k for c in cols) else "F" for k in ("R_CLUMP", "RHOKAP", "TEMP")))
    print(" Clumps saved to " + path)


def resolve_filename(name):
    """io_resolve_filename: iofile_mod.f90:143-177"""
    import os
    base = name.replace("\\", "/").rsplit("/", 1)[-1]
    if "." in base:
        return name
    for ext in (".h5", ".fits.gz", ".fits", ".dat"):
        if os.path.exists(name + ext):
            return name + ext
    raise LartInputError('io_resolve_filename: no file found for "%s"' % name)


def read_clumps_info(par, line, fname, R_sphere):
    """read_clumps_info + rescale_loaded_clumps_to_target: clump_mod.f90:2040-2312, 2385-2449"""
    cl = Clumps(sphere_R=float(R_sphere)); cl.from_file = True
    cl.r_min = max(0.0, par.rmin)
    path = resolve_filename(fname)
    print(" Clumps: reading from " + path)
    cols, keys = read_table_file(path)
    n = int(keys.get("N_CLUMPS", 0))
    if n <= 0:
        raise LartInputError("ERROR: clump_input_file: N_CLUMPS keyword missing or <= 0")
    cl.N = n
    need = lambda k: cols[k][:n].copy() if k in cols else (_ for _ in ()).throw(LartInputError("ERROR: clump_input_file: column %s missing" % k))
    cl.x, cl.y, cl.z = need("X"), need("Y"), need("Z")
    vx, vy, vz = need("VX"), need("VY"), need("VZ")

    def col_or_key(names, key):                                  # read_perclump_or_keyword :1965-2036
        for nm in names:
            if nm in cols:
                return cols[nm][:n].copy(), nm
        if key not in keys:
            raise LartInputError("ERROR: clump_input_file: neither a column %s nor the keyword %s" % ("/".join(names), key))
        return np.full(n, float(keys[key])), ""
    cl.radius, _ = col_or_key(("R_CLUMP",), "CL_RAD")
    cl.rhokap, rk_name = col_or_key(("RHOKAP", "DENSITY", "DENS"), "RHOKAP")
    cl.temperature, _ = col_or_key(("TEMP",), "TEMP_CL")
    user_set = bool(par.distance_unit) and par.distance2cm > 1.0
    if not user_set:                                             # :2164-2176
        if str(keys.get("DISTUNIT", "")).strip():
            par.distance_unit = str(keys["DISTUNIT"]).strip(); print(" Clumps: adopting DISTUNIT = '%s' from clump_input_file" % par.distance_unit)
        if float(keys.get("DIST_CM", 0.0) or 0.0) > 0.0:
            par.distance2cm = float(keys["DIST_CM"]); print(" Clumps: adopting DIST_CM = %12.5E cm/code-unit from clump_input_file" % par.distance2cm)
    cl.vtherm = vtherm_total(line, cl.temperature)
    cl.Dfreq = cl.vtherm / (line.wavelength0 * UM2KM); cl.voigt_a = (line.damping / FOURPI) / cl.Dfreq
    cl.vx, cl.vy, cl.vz = vx / cl.vtherm, vy / cl.vtherm, vz / cl.vtherm
    if rk_name in ("DENSITY", "DENS"):                           # :2190-2207
        if bool(par.distance_unit) or par.distance2cm > 1.0:
            cl.rhokap = cl.rhokap * line.cross0 / cl.Dfreq * par.distance2cm
        else:
            print(" Clumps: WARNING -- column %s found but par%%distance_unit/distance2cm is not set; treating the values as RHOKAP" % rk_name)
    cl.radius_max = float(cl.radius.max()); cl.temperature_ref = float(cl.temperature[0]); cl.vtherm_ref = float(cl.vtherm[0])
    cl.Dfreq_ref = float(cl.Dfreq[0]); cl.voigt_a_ref = float(cl.voigt_a[0]); cl.rhokap_ref = float(cl.rhokap[0])
    R, r0 = cl.sphere_R, cl.r_min; tiny = 2.2250738585072014e-308
    cl.fvol = float(np.cumsum(cl.radius ** 3)[-1]) / max(R ** 3 - r0 ** 3, tiny)
    cl.fcov = 0.75 * float(np.cumsum(cl.radius ** 2)[-1]) / max(R * R + R * r0 + r0 * r0, tiny)
    if par.taumax > 0.0 or par.tauhomo > 0.0 or par.N_gasmax > 0.0 or par.N_gashomo > 0.0:       # rescale_loaded_clumps_to_target
        targets = dict(taumax=par.taumax, tauhomo=par.tauhomo, N_gasmax=par.N_gasmax, N_gashomo=par.N_gashomo)
        real = clump_scalars(line, cl)
        which = next(k for k in ("taumax", "tauhomo", "N_gasmax", "N_gashomo") if targets[k] > 0.0)
        if real[which] <= 0.0:
            print(" Clumps: WARNING -- realized tau/NHI from loaded clumps is zero; cannot rescale. Leaving cl_rhokap unchanged.")
        else:
            alpha = targets[which] / real[which]
            cl.rhokap = cl.rhokap * alpha; cl.rhokap_ref = cl.rhokap_ref * alpha
            print(" Clumps: rescaled to par%%%s = %12.4E, alpha = %12.4E" % (which, targets[which], alpha))
    cl.rhokapD = cl.rhokap * par.cext_dust * par.DGR * cl.Dfreq / line.cross0 if par.DGR > 0.0 else None
    build_csr(cl)
    has = C.c_int32()                                            # init_clumps :678-679: a loaded population is always searched for overlaps
    _hchk(capi.lib().lart_host_clumps_has_overlap(cl.N, capi._p(cl.x), capi._p(cl.y), capi._p(cl.z), capi._p(cl.radius), cl.cgx, cl.cg_min, cl.cg_d,
                                                  capi._p(cl.cg_start), capi._p(cl.cg_list), C.byref(has)))
    cl.has_overlap = bool(has.value)
    print(" Overlap check: overlapping clumps detected — using overlap-aware raytrace." if cl.has_overlap else " Overlap check: no overlaps found — using standard raytrace.")
    return cl


def clump_scalars(line, cl):
    """compute_clump_scalars: clump_mod.f90:2316-2378 (sequential sums in clump order) through lart_host_clump_scalars"""
    from . import hostlib
    out = np.zeros(4)
    hostlib.check(hostlib.lib().lart_host_clump_scalars(cl.N, cl.sphere_R, cl.r_min, line.cross0, *[hostlib.p(np.ascontiguousarray(a, dtype=np.float64)) for a in
                                                          (cl.x, cl.y, cl.z, cl.radius, cl.rhokap, cl.voigt_a, cl.Dfreq)], hostlib.p(out)), LartInputError)
    return dict(tauhomo=float(out[0]), taumax=float(out[1]), N_gashomo=float(out[2]), N_gasmax=float(out[3]))


def compute_clump_scalars(par, line, cl):
    """the four system-level scalars into par (grid_mod_clump.f90:77-93)"""
    r = clump_scalars(line, cl)
    par.tauhomo, par.taumax, par.N_gashomo, par.N_gasmax = r["tauhomo"], r["taumax"], r["N_gashomo"], r["N_gasmax"]


def grid_create_clump(par, line):
    """grid_create_clump: grid_mod_clump.f90:33-131.  Returns (Clumps, Grid): the Cartesian grid only carries the spectral
    axis and the uniform background record (rhokap = 0, Dfreq = cl_Dfreq_ref, voigt_a = cl_voigt_a_ref, v = 0)."""
    from . import host
    if par.rmax <= 0.0:
        raise LartInputError("ERROR: par%rmax must be > 0 for clump medium")
    par.xmax = par.ymax = par.zmax = par.rmax
    par.nx, par.ny, par.nz = max(par.nx, 1) if par.nx >= 1 else 11, par.ny if par.ny >= 1 else 11, par.nz if par.nz >= 1 else 11
    par.xyz_symmetry = False; par.xy_periodic = False
    if par.clump_input_file:                              # init_clumps :674-680: a pre-built population replaces the generator
        cl = read_clumps_info(par, line, par.clump_input_file, par.rmax)
    else:
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
