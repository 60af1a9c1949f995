"""Octree (AMR) host side: input generator + the host mirror of the reference's octree set-up.

    make_amr_sphere / write_generic_amr   our own generator of the plain-text generic format the reference reads
                                          (read_generic_amr.f90:112-185); replaces python/AMR_grid tooling
    read_generic_amr                      text reader, same header rules
    amr_build_tree / amr_build_neighbors  octree_mod.f90:460-569, :619-682 -- same cell numbering as the reference
                                          (cells are created in the order the leaves of the file first reach them)
    grid_create_amr                       grid_mod_amr.f90:34-526 (text input, neutral hydrogen, global DGR)
    create_JPa_amr_mem                    grid_mod_amr.f90:543-724
numpy only; the transport itself runs in liblart_gpu.so (lart_gpu_set_grid_amr).
"""
import math
from dataclasses import dataclass
import numpy as np

from .host import (LartInputError, FOURPI, UM2KM, vtherm_total, voigt_center)


def _seqsum(v):
    s = 0.0
    for t in v:
        s = s + float(t)
    return s


# ---------------------------------------------------------------------------------------------------------------
# input generation
# ---------------------------------------------------------------------------------------------------------------
def make_amr_sphere(level_min=3, level_max=7, boxlen=2.0, radius=None, nH=1.0, T=1.0e4, refine_boundary=True):
    """uniform sphere in a cubic box centred on the origin: leaves at level_min everywhere, and every leaf cut by the
    sphere surface split down to level_max (a cell is cut when it touches the ball but is not wholly inside it).
    Gas density nH inside (cell centre within the radius), 0 outside.  Returns columns x,y,z,level,nH,T,vx,vy,vz."""
    radius = boxlen / 2.0 if radius is None else radius
    r2 = radius * radius
    out = []

    def rec(cx, cy, cz, h, lev):
        split = lev < level_min
        if not split and refine_boundary and lev < level_max:
            # farthest corner inside? / nearest point inside?
            fx, fy, fz = abs(cx) + h, abs(cy) + h, abs(cz) + h
            inside = fx * fx + fy * fy + fz * fz <= r2
            nx, ny, nz = max(abs(cx) - h, 0.0), max(abs(cy) - h, 0.0), max(abs(cz) - h, 0.0)
            touches = nx * nx + ny * ny + nz * nz <= r2
            split = touches and not inside
        if split:
            q = h / 2.0
            for iz in (-1, 1):
                for iy in (-1, 1):
                    for ix in (-1, 1):
                        rec(cx + ix * q, cy + iy * q, cz + iz * q, q, lev + 1)
        else:
            out.append((cx, cy, cz, lev, nH if cx * cx + cy * cy + cz * cz <= r2 else 0.0))

    rec(0.0, 0.0, 0.0, boxlen / 2.0, 0)
    a = np.array(out)
    n = len(a)
    return dict(x=a[:, 0], y=a[:, 1], z=a[:, 2], level=a[:, 3].astype(np.int32), nH=a[:, 4], T=np.full(n, T),
                vx=np.zeros(n), vy=np.zeros(n), vz=np.zeros(n), boxlen=boxlen)


def make_amr_slab(level_min=2, level_max=3, boxlen=2.0, zgas=None, nH=1.0, T=1.0e4):
    """plane-parallel medium in a cubic box for par%xy_periodic runs: uniform gas for |z| <= zgas (the whole box when None), leaves at
    level_min, the x > 0, y > 0 quadrant split down to level_max -- so a packet that wraps in x or y lands on leaves of another size
    than the ones it left, while the physics is that of the uniform slab.  Columns as make_amr_sphere."""
    out = []

    def rec(cx, cy, cz, h, lev):
        if lev < level_min or (lev < level_max and cx > 0.0 and cy > 0.0):
            q = h / 2.0
            for iz in (-1, 1):
                for iy in (-1, 1):
                    for ix in (-1, 1):
                        rec(cx + ix * q, cy + iy * q, cz + iz * q, q, lev + 1)
        else:
            out.append((cx, cy, cz, lev, nH if (zgas is None or abs(cz) <= zgas) else 0.0))

    rec(0.0, 0.0, 0.0, boxlen / 2.0, 0)
    a = np.array(out)
    n = len(a)
    return dict(x=a[:, 0], y=a[:, 1], z=a[:, 2], level=a[:, 3].astype(np.int32), nH=a[:, 4], T=np.full(n, T),
                vx=np.zeros(n), vy=np.zeros(n), vz=np.zeros(n), boxlen=boxlen)


def write_generic_amr(path, d):
    with open(path, "w") as f:
        f.write("# generic AMR leaf table: x y z level nH T vx vy vz (box centred on the origin)\n")
        f.write("NLEAF %d\nBOXLEN %.17g\n" % (len(d["x"]), d["boxlen"]))
        for i in range(len(d["x"])):
            f.write("%.17g %.17g %.17g %d %.17g %.17g %.17g %.17g %.17g\n" %
                    (d["x"][i], d["y"][i], d["z"][i], d["level"][i], d["nH"][i], d["T"][i], d["vx"][i], d["vy"][i], d["vz"][i]))


_COLS = ("x", "y", "z", "level", "dens", "T", "vx", "vy", "vz")


def _fits_card(key, val, comment=""):
    if isinstance(val, str):
        v = "'%-8s'" % val
        card = "%-8s= %-20s" % (key, v)
    elif isinstance(val, (bool, np.bool_)):
        card = "%-8s= %20s" % (key, "T" if val else "F")
    elif isinstance(val, (int, np.integer)):
        card = "%-8s= %20d" % (key, val)
    else:
        card = "%-8s= %20s" % (key, ("%.15E" % val))
    if comment:
        card += " / " + comment
    return card[:80].ljust(80)


def write_generic_amr_fits(path, d, origin=None, single=False):
    """FITS binary table as python/AMR_grid/AMR_grid.py:_write_fits lays it out (primary HDU + BINTABLE 'AMR_GRID' with columns
    x y z level dens T vx vy vz, keywords BOXLEN, ORIGINX/Y/Z, NLEAF); .gz paths are gzip-compressed"""
    import gzip
    n = len(d["x"]); bl = float(d["boxlen"]); org = [-0.5 * bl] * 3 if origin is None else list(origin)
    ft = ">f4" if single else ">f8"; form = "1E" if single else "1D"
    dt = np.dtype([(c, ">i4" if c == "level" else ft) for c in _COLS])
    tab = np.zeros(n, dtype=dt)
    for c, k in zip(_COLS, ("x", "y", "z", "level", "nH", "T", "vx", "vy", "vz")):
        tab[c] = d[k]
    def block(cards):
        h = "".join(cards) + "END".ljust(80)
        return (h + " " * (-len(h) % 2880)).encode("ascii")
    prim = block([_fits_card("SIMPLE", True), _fits_card("BITPIX", 8), _fits_card("NAXIS", 0), _fits_card("EXTEND", True)])
    cards = [_fits_card("XTENSION", "BINTABLE"), _fits_card("BITPIX", 8), _fits_card("NAXIS", 2), _fits_card("NAXIS1", dt.itemsize),
             _fits_card("NAXIS2", n), _fits_card("PCOUNT", 0), _fits_card("GCOUNT", 1), _fits_card("TFIELDS", len(_COLS))]
    for i, c in enumerate(_COLS):
        cards += [_fits_card("TTYPE%d" % (i + 1), c), _fits_card("TFORM%d" % (i + 1), "1J" if c == "level" else form)]
    cards += [_fits_card("EXTNAME", "AMR_GRID"), _fits_card("BOXLEN", bl, "Simulation box length"), _fits_card("ORIGINX", org[0]),
              _fits_card("ORIGINY", org[1]), _fits_card("ORIGINZ", org[2]), _fits_card("NLEAF", n)]
    body = tab.tobytes(); body += b"\0" * (-len(body) % 2880)
    with (gzip.open(path, "wb") if path.endswith(".gz") else open(path, "wb")) as f:
        f.write(prim + block(cards) + body)


def _read_generic_amr_fits(path):
    import gzip
    with (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")) as f:
        raw = f.read()
    pos = 0
    def header():
        nonlocal pos
        kw = {}
        while True:
            blk = raw[pos:pos + 2880].decode("ascii"); pos += 2880
            for i in range(36):
                card = blk[80 * i:80 * i + 80]; key = card[:8].strip()
                if key == "END":
                    return kw
                if card[8:10] == "= ":
                    v = card[10:].split("/")[0].strip() if "'" not in card[10:] else card[10:].split("'")[1].strip()
                    kw[key] = v
    h0 = header()
    if int(h0.get("NAXIS", 0)) > 0:
        nb = abs(int(h0["BITPIX"])) // 8
        for i in range(int(h0["NAXIS"])):
            nb *= int(h0["NAXIS%d" % (i + 1)])
        pos += (nb + 2879) // 2880 * 2880
    h = header()
    if h.get("XTENSION", "") != "BINTABLE" or "BOXLEN" not in h:
        raise LartInputError("generic_amr_read_binary: no BINTABLE with BOXLEN in " + path)
    codes = {"D": ">f8", "E": ">f4", "J": ">i4", "K": ">i8", "I": ">i2"}
    dt = np.dtype([("c%d" % i, codes[h["TFORM%d" % (i + 1)].lstrip("0123456789")[0]]) for i in range(int(h["TFIELDS"]))])
    n = int(h["NAXIS2"]); tab = np.frombuffer(raw, dtype=dt, count=n, offset=pos)
    bl = float(h["BOXLEN"]); org = [float(h.get(k, -0.5 * bl)) for k in ("ORIGINX", "ORIGINY", "ORIGINZ")]
    a = [tab["c%d" % i].astype(np.float64) for i in range(9)]
    return dict(x=a[0], y=a[1], z=a[2], level=a[3].astype(np.int32), nH=a[4], T=a[5], vx=a[6], vy=a[7], vz=a[8], boxlen=bl, origin=org)


def _h5():
    import ctypes as C
    lib = None
    for cand in ("/opt/conda/lib/libhdf5.so", "libhdf5.so"):
        try:
            lib = C.CDLL(cand); break
        except OSError:
            pass
    if lib is None:
        raise LartInputError("no HDF5 C library found (generic AMR .h5 files)")
    hid = C.c_int64; lib.H5open()
    for fn in ("H5Fcreate", "H5Fopen", "H5Gcreate2", "H5Gopen2", "H5Screate_simple", "H5Screate", "H5Dcreate2", "H5Dopen2", "H5Acreate2", "H5Aopen", "H5Dget_space"):
        getattr(lib, fn).restype = hid
    lib.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, hid, hid]; lib.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    lib.H5Gcreate2.argtypes = [hid, C.c_char_p, hid, hid, hid]; lib.H5Gopen2.argtypes = [hid, C.c_char_p, hid]
    lib.H5Screate_simple.argtypes = [C.c_int, C.c_void_p, C.c_void_p]; lib.H5Screate.argtypes = [C.c_int]
    lib.H5Dcreate2.argtypes = [hid, C.c_char_p, hid, hid, hid, hid, hid]; lib.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    lib.H5Dwrite.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]; lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    lib.H5Acreate2.argtypes = [hid, C.c_char_p, hid, hid, hid, hid]; lib.H5Aopen.argtypes = [hid, C.c_char_p, hid]
    lib.H5Awrite.argtypes = [hid, hid, C.c_void_p]; lib.H5Aread.argtypes = [hid, hid, C.c_void_p]
    lib.H5Dget_space.argtypes = [hid]; lib.H5Sget_simple_extent_dims.argtypes = [hid, C.c_void_p, C.c_void_p]
    lib.H5Aexists.argtypes = [hid, C.c_char_p]; lib.H5Lexists.argtypes = [hid, C.c_char_p, hid]
    for fn in ("H5Fclose", "H5Gclose", "H5Sclose", "H5Dclose", "H5Aclose"):
        getattr(lib, fn).argtypes = [hid]
    T = {k: hid.in_dll(lib, "H5T_NATIVE_%s_g" % k).value for k in ("DOUBLE", "INT")}
    return C, lib, T


def write_generic_amr_h5(path, d, origin=None, group="AMR_GRID"):
    """HDF5 form of python/AMR_grid/AMR_grid.py:_write_hdf5: one group of 1-D datasets + attributes BOXLEN, ORIGINX/Y/Z, NLEAF"""
    C, lib, T = _h5()
    n = len(d["x"]); bl = float(d["boxlen"]); org = [-0.5 * bl] * 3 if origin is None else list(origin)
    f = lib.H5Fcreate(path.encode(), 2, 0, 0)                      # H5F_ACC_TRUNC
    g = lib.H5Gcreate2(f, group.encode(), 0, 0, 0)
    dims = (C.c_uint64 * 1)(n); sp = lib.H5Screate_simple(1, dims, None)
    for c, k in zip(_COLS, ("x", "y", "z", "level", "nH", "T", "vx", "vy", "vz")):
        if c == "level":
            a = np.ascontiguousarray(d[k], dtype=np.int32); t = T["INT"]
        else:
            a = np.ascontiguousarray(d[k], dtype=np.float64); t = T["DOUBLE"]
        ds = lib.H5Dcreate2(g, c.encode(), t, sp, 0, 0, 0); lib.H5Dwrite(ds, t, 0, 0, 0, a.ctypes.data_as(C.c_void_p)); lib.H5Dclose(ds)
    lib.H5Sclose(sp)
    sc = lib.H5Screate(0)                                          # H5S_SCALAR
    for k, v in (("BOXLEN", bl), ("ORIGINX", org[0]), ("ORIGINY", org[1]), ("ORIGINZ", org[2])):
        a = lib.H5Acreate2(g, k.encode(), T["DOUBLE"], sc, 0, 0); lib.H5Awrite(a, T["DOUBLE"], C.byref(C.c_double(v))); lib.H5Aclose(a)
    for k in ("NLEAF", "NAXIS2"):
        a = lib.H5Acreate2(g, k.encode(), T["INT"], sc, 0, 0); lib.H5Awrite(a, T["INT"], C.byref(C.c_int(n))); lib.H5Aclose(a)
    lib.H5Sclose(sc); lib.H5Gclose(g); lib.H5Fclose(f)


def _read_generic_amr_h5(path, group="AMR_GRID"):
    C, lib, T = _h5()
    f = lib.H5Fopen(path.encode(), 0, 0)
    if f < 0:
        raise LartInputError("generic_amr_read_binary: cannot open " + path)
    g = lib.H5Gopen2(f, group.encode(), 0)
    if g < 0:
        raise LartInputError("generic_amr_read_binary: no group %s in %s" % (group, path))
    def attr(k, default=None):
        if lib.H5Aexists(g, k.encode()) <= 0:
            return default
        a = lib.H5Aopen(g, k.encode(), 0); v = C.c_double(); lib.H5Aread(a, T["DOUBLE"], C.byref(v)); lib.H5Aclose(a); return v.value
    bl = attr("BOXLEN"); org = [attr(k, -0.5 * bl) for k in ("ORIGINX", "ORIGINY", "ORIGINZ")]
    cols = []
    for c in _COLS:
        ds = lib.H5Dopen2(g, c.encode(), 0); sp = lib.H5Dget_space(ds); dims = (C.c_uint64 * 1)(); lib.H5Sget_simple_extent_dims(sp, dims, None)
        a = np.zeros(dims[0]); lib.H5Dread(ds, T["DOUBLE"], 0, 0, 0, a.ctypes.data_as(C.c_void_p)); lib.H5Sclose(sp); lib.H5Dclose(ds)
        cols.append(a)
    lib.H5Gclose(g); lib.H5Fclose(f)
    return dict(x=cols[0], y=cols[1], z=cols[2], level=cols[3].astype(np.int32), nH=cols[4], T=cols[5], vx=cols[6], vy=cols[7], vz=cols[8],
                boxlen=bl, origin=org)


def read_generic_amr(path):
    """generic_amr_read (read_generic_amr.f90:52-279): text, FITS binary table (.fits, .fits.gz) or HDF5 (.h5, .hdf5)"""
    low = path.lower()
    if low.endswith((".fits", ".fits.gz")):
        return _read_generic_amr_fits(path)
    if low.endswith((".h5", ".hdf5")):
        return _read_generic_amr_h5(path)
    return _read_generic_amr_text(path)


def _read_generic_amr_text(path):
    n, bl, rows = 0, -1.0, []
    with open(path) as f:
        lines = iter(f)
        for ln in lines:
            q = ln.strip()
            if not q or q.startswith("#"):
                continue
            if q[:6].upper() == "BOXLEN":
                bl = float(q[6:])
            elif q[:5].upper() == "NLEAF":
                n = int(q[5:])
            else:
                t = q.split(); n, bl = int(t[0]), float(t[1])
            if n > 0 and bl > 0:
                break
        if n <= 0 or bl <= 0:
            raise LartInputError("generic_amr_read: text header must provide NLEAF and BOXLEN")
        for ln in lines:
            q = ln.strip()
            if not q or q.startswith("#"):
                continue
            rows.append([float(t) for t in q.split()[:9]])
            if len(rows) == n:
                break
    if len(rows) != n:
        raise LartInputError("generic_amr_read: expected %d rows, got %d" % (n, len(rows)))
    a = np.array(rows)
    return dict(x=a[:, 0], y=a[:, 1], z=a[:, 2], level=a[:, 3].astype(np.int32), nH=a[:, 4], T=a[:, 5], vx=a[:, 6], vy=a[:, 7], vz=a[:, 8], boxlen=bl)


# ---------------------------------------------------------------------------------------------------------------
# octree
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class AmrGrid:
    ncells: int = 0
    nleaf: int = 0
    levelmax: int = 0
    parent: np.ndarray = None
    children: np.ndarray = None        # (8, ncells) int32, 1-based, 0 = none
    level: np.ndarray = None
    ileaf: np.ndarray = None
    icell_of_leaf: np.ndarray = None
    neighbor: np.ndarray = None        # (6, ncells)
    cx: np.ndarray = None
    cy: np.ndarray = None
    cz: np.ndarray = None
    ch: np.ndarray = None
    xmin: float = 0.0
    xmax: float = 0.0
    L_box: float = 0.0
    # leaf tables
    rhokap: np.ndarray = None
    rhokapD: np.ndarray = None
    Dfreq: np.ndarray = None
    voigt_a: np.ndarray = None
    vfx: np.ndarray = None
    vfy: np.ndarray = None
    vfz: np.ndarray = None
    # profile tallies
    nr_JPa: int = 0
    nz_JPa: int = 0
    dr_JPa: float = 0.0
    dz_JPa: float = 0.0
    vol: np.ndarray = None
    nbins: int = 0


def amr_build_tree(x, y, z, level, lo, hi):
    """same numbering as octree_mod.f90:460-569: root = 1; a cell's number is its creation rank when the leaves are
    inserted in file order (sort key: first leaf that passes through it, then depth)."""
    nleaf = len(x)
    L = hi - lo
    lmax = int(level.max())
    # integer coordinates at the finest level of every leaf centre
    scale = (1 << lmax) / L
    fx = np.floor((x - lo) * scale).astype(np.int64); fy = np.floor((y - lo) * scale).astype(np.int64); fz = np.floor((z - lo) * scale).astype(np.int64)
    keys, first, lev_of = [], [], []
    for l in range(1, lmax + 1):
        m = np.nonzero(level >= l)[0]
        sh = lmax - l
        k = ((fx[m] >> sh) << 42) | ((fy[m] >> sh) << 21) | (fz[m] >> sh)
        uk, idx = np.unique(k, return_index=True)
        keys.append(uk); first.append(m[idx]); lev_of.append(np.full(len(uk), l, dtype=np.int64))
    allk = np.concatenate(keys); allf = np.concatenate(first); alll = np.concatenate(lev_of)
    order = np.lexsort((alll, allf))                     # by first leaf, then by depth: the insertion order
    allk, alll = allk[order], alll[order]
    ncells = len(allk) + 1
    cell_level = np.concatenate([[0], alll]).astype(np.int32)
    ix = np.concatenate([[0], allk >> 42]); iy = np.concatenate([[0], (allk >> 21) & ((1 << 21) - 1)]); iz = np.concatenate([[0], allk & ((1 << 21) - 1)])
    h = (L / 2.0) / (2.0 ** cell_level)
    # centres by the reference's recurrence c_child = c_parent +- h_child, accumulated level by level
    lookup = {}
    for l in range(0, lmax + 1):
        sel = np.nonzero(cell_level == l)[0]
        lookup[l] = dict(zip(((ix[sel] << 42) | (iy[sel] << 21) | iz[sel]).tolist(), (sel + 1).tolist()))
    parent = np.zeros(ncells, dtype=np.int32)
    children = np.zeros((8, ncells), dtype=np.int32, order="F")
    cx = np.zeros(ncells); cy = np.zeros(ncells); cz = np.zeros(ncells)
    cx[0] = cy[0] = cz[0] = (lo + hi) * 0.5
    for l in range(1, lmax + 1):
        sel = np.nonzero(cell_level == l)[0]
        pk = ((ix[sel] >> 1) << 42) | ((iy[sel] >> 1) << 21) | (iz[sel] >> 1)
        par = np.array([lookup[l - 1][int(k)] for k in pk.tolist()], dtype=np.int32) if l > 1 else np.ones(len(sel), dtype=np.int32)
        bx, by, bz = ix[sel] & 1, iy[sel] & 1, iz[sel] & 1
        parent[sel] = par
        children[bx + 2 * by + 4 * bz, par - 1] = sel + 1
        cx[sel] = cx[par - 1] + (2 * bx - 1) * h[sel]; cy[sel] = cy[par - 1] + (2 * by - 1) * h[sel]; cz[sel] = cz[par - 1] + (2 * bz - 1) * h[sel]
    # leaves
    lk = ((fx >> (lmax - level)) << 42) | ((fy >> (lmax - level)) << 21) | (fz >> (lmax - level))
    icell_of_leaf = np.array([lookup[int(l)][int(k)] for l, k in zip(level.tolist(), lk.tolist())], dtype=np.int32)
    ileaf = np.zeros(ncells, dtype=np.int32)
    ileaf[icell_of_leaf - 1] = np.arange(1, nleaf + 1, dtype=np.int32)
    g = AmrGrid(ncells=ncells, nleaf=nleaf, levelmax=lmax, parent=parent, children=children, level=cell_level, ileaf=ileaf,
                icell_of_leaf=icell_of_leaf, cx=cx, cy=cy, cz=cz, ch=h, xmin=lo, xmax=hi, L_box=L)
    g._lookup, g._ijk = lookup, (ix, iy, iz)
    return g


def amr_build_neighbors(g):
    """octree_mod.f90:619-682: same-level face neighbour, else the coarser cell that covers it; ancestors of the cell
    itself (sparse trees) and the outside are 0."""
    ix, iy, iz = g._ijk
    nb = np.zeros((6, g.ncells), dtype=np.int32, order="F")
    n_at = 1 << g.level.astype(np.int64)
    for f, (dx, dy, dz) in enumerate(((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))):
        jx, jy, jz = ix + dx, iy + dy, iz + dz
        ok = (jx >= 0) & (jx < n_at) & (jy >= 0) & (jy < n_at) & (jz >= 0) & (jz < n_at)
        ok[0] = False
        res = np.zeros(g.ncells, dtype=np.int32)
        todo = np.nonzero(ok)[0]
        lev = g.level[todo].astype(np.int64); kx, ky, kz = jx[todo], jy[todo], jz[todo]
        found = np.zeros(len(todo), dtype=np.int32)
        for up in range(0, g.levelmax + 1):          # walk up until an existing cell covers the position
            pend = np.nonzero((found == 0) & (lev - up >= 0))[0]
            if len(pend) == 0:
                break
            for i in pend.tolist():
                l = int(lev[i] - up)
                c = g._lookup[l].get(int(((kx[i] >> up) << 42) | ((ky[i] >> up) << 21) | (kz[i] >> up))) if l > 0 else 1
                if c:
                    # find_cell_at_level stops at a leaf or at the target level; a coarser INTERNAL cell means a gap below it
                    found[i] = c
        res[todo] = found
        nb[f, :] = res
    # drop neighbours that are ancestors of the cell (gap in a sparse tree)
    for f in range(6):
        cand = np.nonzero(nb[f] > 0)[0]
        for c in cand.tolist():
            a, t = g.parent[c], nb[f, c]
            if g.level[t - 1] >= g.level[c]:
                continue
            while a > 0:
                if a == t:
                    nb[f, c] = 0; break
                a = g.parent[a - 1]
    g.neighbor = nb
    return g


def amr_find_leaf(g, x, y, z):
    if not (g.xmin <= x <= g.xmax and g.xmin <= y <= g.xmax and g.xmin <= z <= g.xmax):
        return 0
    ic = 1
    while True:
        if g.ileaf[ic - 1] > 0:
            return int(g.ileaf[ic - 1])
        io = (1 if x >= g.cx[ic - 1] else 0) + (2 if y >= g.cy[ic - 1] else 0) + (4 if z >= g.cz[ic - 1] else 0)
        ic = int(g.children[io, ic - 1])
        if ic == 0:
            return 0


def grid_create_amr(par, line, path):
    """grid_mod_amr.f90:34-526 for the generic text format: tree, leaf physics, +z pole normalisation, frequency grid.
    Returns (AmrGrid, Grid-like namespace with the spectral axis)."""
    from . import host
    d = read_generic_amr(path)
    bl = d["boxlen"]; org = d.get("origin", [-0.5 * bl] * 3)
    if not (org[0] == org[1] == org[2]):
        raise LartInputError("an AMR box with different ORIGINX/Y/Z is not supported by this path (cubic box [lo,hi]^3)")
    g = amr_build_tree(d["x"], d["y"], d["z"], d["level"], org[0], org[0] + bl)
    amr_build_neighbors(g)
    if par.geometry == "sphere":
        par.rmax = g.L_box * 0.5; par.xmax = par.ymax = par.zmax = par.rmax
    else:
        par.rmax = -1.0; par.xmax = par.ymax = par.zmax = g.xmax
    if par.source_rmax < 0.0:
        par.source_rmax = par.rmax
    Dfreq_ref = vtherm_total(line, par.temperature) / (line.wavelength0 * UM2KM)
    voigt_amean = (line.damping / FOURPI) / Dfreq_ref
    T = np.maximum(d["T"], 10.0)
    vt = vtherm_total(line, T)                                            # grid_mod_amr.f90:220-221
    g.Dfreq = vt / (line.wavelength0 * UM2KM)
    g.voigt_a = (line.damping / FOURPI) / g.Dfreq
    g.rhokap = d["nH"] * line.cross0 / g.Dfreq * par.distance2cm
    g.rhokapD = d["nH"] * par.cext_dust * par.DGR * par.distance2cm if par.DGR > 0.0 else None
    g.vfx, g.vfy, g.vfz = d["vx"] / vt, d["vy"] / vt, d["vz"] / vt
    if 0.0 < par.cone_opening < 90.0:                    # Step 4b (:336-350): biconical mask by leaf centre
        ic = g.icell_of_leaf - 1
        rr = np.sqrt(g.cx[ic] ** 2 + g.cy[ic] ** 2 + g.cz[ic] ** 2)
        with np.errstate(invalid="ignore", divide="ignore"):
            out = (rr > 0.0) & (np.abs(g.cz[ic]) / rr < math.cos(par.cone_opening * math.pi / 180.0))
        g.rhokap[out] = 0.0
        if g.rhokapD is not None:
            g.rhokapD[out] = 0.0
    H0 = voigt_center(g.voigt_a)
    nopac = float((g.rhokap > 0).sum())
    opac_length = bl / 2.0
    tauhomo = _seqsum(g.rhokap * H0) / nopac * opac_length if nopac > 0 else 0.0
    # pole traversal +z from the box centre
    xc = yc = zc = float(g.cx[0]); tau_raw = NHI = 0.0
    while zc < g.xmax:
        il = amr_find_leaf(g, xc, yc, zc)
        if il <= 0:
            break
        ic = g.icell_of_leaf[il - 1]
        t = (g.cz[ic - 1] + g.ch[ic - 1] - zc) / 1.0
        tau_raw += g.rhokap[il - 1] * H0[il - 1] * t
        NHI += g.rhokap[il - 1] * g.Dfreq[il - 1] / line.cross0 * t
        zc += t
    taupole = tau_raw if tau_raw > 0 else tauhomo
    opac_norm = 1.0
    if par.taumax > 0.0:
        if taupole > 0:
            opac_norm = par.taumax / taupole
    elif par.N_gasmax > 0.0 and NHI > 0:
        opac_norm = par.N_gasmax / NHI
    g.rhokap = g.rhokap * opac_norm
    if g.rhokapD is not None:
        g.rhokapD = g.rhokapD * opac_norm
    tauhomo = _seqsum(g.rhokap * H0) / nopac * opac_length if nopac > 0 else 0.0
    taupole = par.taumax if par.taumax > 0 else tauhomo
    par.N_gasmax = NHI * opac_norm; par.N_gashomo = par.N_gasmax; par.tauhomo = tauhomo; par.taumax = taupole
    sg = host.Grid(nx=g.nleaf, ny=1, nz=1, Dfreq_ref=Dfreq_ref, voigt_amean=voigt_amean, xmax=par.xmax, ymax=par.ymax, zmax=par.zmax,
                   xmin=g.xmin, ymin=g.xmin, zmin=g.xmin, rmax=par.rmax, dx=g.L_box, dy=g.L_box, dz=g.L_box)
    at = voigt_amean * par.tauhomo / (bl / (2.0 * g.ch[0]))
    if at > 1.0:
        xi, chi = (0.6, 1.2) if at <= 60.0 else (1.4, 0.6)
        sg.xcrit = 0.02 * math.exp(xi * math.log(at) ** chi); sg.xcrit2 = sg.xcrit ** 2
    host.car_setup_freq_grid(par, line, sg)              # amr_setup_freq_grid (:726-800) applies the same rules
    return g, sg


def create_JPa_amr_mem(par, g):
    """grid_mod_amr.f90:543-724: number of bins and the gas volume per bin (ns^3 sub-sampling of every gas leaf)"""
    if not (par.calc_J or par.calc_P or par.calc_Pnew):
        return g
    rmax = par.rmax if par.rmax > 0 else g.L_box * 0.5
    nr = min(512, max(1, 2 ** max(0, g.levelmax - 1))); nz = min(512, max(1, 2 ** max(0, g.levelmax)))
    g.nr_JPa, g.nz_JPa, g.dr_JPa, g.dz_JPa = nr, nz, rmax / nr, g.L_box / nz
    nadd = 8 if par.xyz_symmetry else (4 if par.xy_symmetry else 1)      # :577-588: the periodic slab does not fold
    ic = g.icell_of_leaf - 1
    hc = g.ch[ic]; vleaf = (2.0 * hc) ** 3 * nadd
    if par.geometry_JPa == 3:
        g.nbins, g.vol = g.nleaf, vleaf
        return g
    gas = g.rhokap > 0
    if par.geometry_JPa == 1:
        vol = np.zeros(nr)
        ns_all = np.clip(np.rint(2.0 * hc / g.dr_JPa).astype(int), 2, 8)
        for ns in np.unique(ns_all[gas]):
            m = np.nonzero(gas & (ns_all == ns))[0]
            off = (2.0 * np.arange(1, ns + 1) - ns - 1.0) / ns
            ox, oy, oz = np.meshgrid(off, off, off, indexing="ij")
            xs = g.cx[ic[m], None] + ox.ravel()[None, :] * hc[m, None]
            ys = g.cy[ic[m], None] + oy.ravel()[None, :] * hc[m, None]
            zs = g.cz[ic[m], None] + oz.ravel()[None, :] * hc[m, None]
            ir = np.floor(np.sqrt(xs * xs + ys * ys + zs * zs) / g.dr_JPa).astype(int)
            w = np.broadcast_to((vleaf[m] / ns ** 3)[:, None], ir.shape)
            ok = ir < nr
            np.add.at(vol, ir[ok], w[ok])
        g.nbins, g.vol = nr, vol
    else:
        vol = np.zeros(nz); dz, zlo = g.dz_JPa, g.xmin
        for i in np.nonzero(gas)[0].tolist():
            z1, z2 = g.cz[ic[i]] - hc[i], g.cz[ic[i]] + hc[i]
            for iz in range(max(1, int(math.floor((z1 - zlo) / dz)) + 1), min(nz, int(math.floor((z2 - zlo) / dz)) + 1) + 1):
                ov = min(z2, zlo + iz * dz) - max(z1, zlo + (iz - 1) * dz)
                if ov > 0:
                    vol[iz - 1] += vleaf[i] * ov / (2.0 * hc[i])
        g.nbins, g.vol = nz, vol
    return g
