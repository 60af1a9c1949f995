"""ctypes binding of the C-ABI in include/lart_gpu.h (liblart_gpu.so, built in-tree by lart_amd/csrc/Makefile).

Plumbing only: the product is the shared library.  Loading fails loudly when the library is missing --
there is no CPU path behind this module.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LART_GPU_LIB") or os.path.join(_HERE, "csrc", "liblart_gpu.so")   # override: tuning builds only

_d, _i32, _i64, _u64, _vp = C.c_double, C.c_int32, C.c_int64, C.c_uint64, C.c_void_p


class LartParams(C.Structure):
    """struct lart_params (include/lart_gpu.h) -- field names are the reference's par%/grid%/line% names."""
    _fields_ = [(n, _i32) for n in (
        "xyz_symmetry", "xy_periodic", "xy_symmetry", "source_geometry", "spectral_type", "comoving_source",
        "recoil", "core_skip", "core_skip_global", "use_stokes", "use_reduced_wgt",
        "save_Jin", "save_Jabs", "save_Jmu", "nmu", "save_peeloff", "save_peeloff_2D", "save_peeloff_3D", "save_direc0", "save_dust_scattered", "nxfreq", "line_type")] + \
        [(n, _d) for n in (
            "xs_point", "ys_point", "zs_point", "source_rmax", "xfreq0", "xfreq_min", "dxfreq", "Dfreq_ref", "xfreq_max", "source_sigma_x", "xrange", "yrange", "zrange", "source_zscale",
            "DGR", "hgg", "albedo", "mu_min", "dmu", "g_recoil0", "E1", "E2", "E3", "DnuHK_Hz", "delta_nu_HD_Hz", "ratio_Dfreq_HD", "ratio_voigta_HD", "nD_over_nH", "g_recoil0_D", "xcrit", "xcrit2",
            "Omega")] + [("nxfreq_Ha", _i32), ("h2_nlines", _i32)] + \
        [(n, _d) for n in ("xfreq_min_Ha", "dxfreq_Ha", "R_Ha", "albedo_Ha", "hgg_Ha", "P_conv")] + [("E_conv", _d * 3), ("h2_hi_width", _i32), ("pad_", _i32), ("h2_Dfreq_Hz", _d)] + \
        [(n, _d * 4) for n in ("h2_dnu_Hz", "h2_strength", "h2_a_damp", "h2_p_scat")]


class LartTallies(C.Structure):
    _fields_ = [(n, _vp) for n in ("Jout", "Jin", "Jabs", "Jmu", "scatt", "direc", "I", "Q", "U", "V",
                                   "scatt_2D", "direc_2D", "I_2D", "Q_2D", "U_2D", "V_2D", "direc0", "direc0_2D", "scatt_dust", "J1", "P1", "P1_new")] + \
        [(n, _d) for n in ("nscatt_gas", "nscatt_dust", "n_events", "n_cellsteps", "n_cellsteps_peel", "n_atomics")] + \
        [("nphotons_done", _i64), ("nphotons_active", _i64)] + [(n, _vp) for n in ("Jout_Ha", "Jabs_Ha", "peel_Ha", "Pconv")] + [("aux", _d * 8)]


class LartLineMultiplet(C.Structure):
    """lart_line_multiplet of include/lart_gpu.h (line types 4, 5, 6: line%b(:), line_mod.f90:713-1153)"""
    _fields_ = [("line_type", _i32), ("nup", _i32), ("HeI_coherent", _i32), ("ndown", _i32 * 3), ("A_down", (_i32 * 4) * 3), ("ion_id", C.c_char * 8),
                ("wavelength0", _d), ("cross0", _d), ("vtherm1", _d), ("g_recoil0", _d), ("damping", _d),
                ("f12", _d * 3), ("delE_Hz", _d * 3), ("damping_up", _d * 3)] + \
        [(n, (_d * 4) * 3) for n in ("A21", "Elow_Hz", "E1", "E2", "E3", "P_down")]


SYMBOLS = ["lart_host_line_lookup", "lart_gpu_set_line_multiplet", "lart_gpu_device_count", "lart_gpu_create", "lart_gpu_destroy", "lart_gpu_last_error",
           "lart_gpu_set_grid_car", "lart_gpu_set_mueller", "lart_gpu_set_observers", "lart_gpu_set_observers_heal",
           "lart_gpu_set_JPa", "lart_gpu_set_grid_amr", "lart_gpu_set_JPa_amr", "lart_gpu_set_clumps", "lart_gpu_set_clump_overlap", "lart_host_last_error", "lart_host_clumps_generate", "lart_host_clumps_generate_cone", "lart_host_clumps_csr_dims", "lart_host_clumps_build_csr", "lart_host_clumps_has_overlap", "lart_host_amr_last_error", "lart_host_amr_count", "lart_host_amr_build", "lart_gpu_run",
           "lart_gpu_begin", "lart_gpu_step", "lart_gpu_sync", "lart_gpu_kernel_time", "lart_gpu_sightline_tau", "lart_gpu_sightline_time",
           "lart_gpu_enable_photon_log", "lart_gpu_get_photon_log", "lart_gpu_enable_all_photons", "lart_gpu_get_all_photons", "lart_gpu_set_pool", "lart_gpu_pool_info", "lart_gpu_test_cell_margin",
           "lart_gpu_tally_buffer", "lart_gpu_zero_tallies", "lart_gpu_get_tallies",
           "lart_gpu_compact", "lart_gpu_set_auto_compact", "lart_gpu_live_lanes",
           "lart_gpu_comm_unique_id", "lart_gpu_comm_init", "lart_gpu_comm_info", "lart_gpu_comm_destroy", "lart_gpu_reduce_tallies",
           "lart_gpu_reduce_tallies_multi", "lart_gpu_run_multi", "lart_host_entropy_seed",
           "lart_gpu_test_voigt", "lart_gpu_test_rand", "lart_gpu_test_raytrace", "lart_gpu_test_vec2pix", "lart_gpu_test_scatter", "lart_gpu_test_trace", "lart_gpu_test_event"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("liblart_gpu.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C lart_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.lart_gpu_last_error.restype = C.c_char_p
        L.lart_gpu_create.argtypes = [C.POINTER(_vp), C.POINTER(LartParams), C.c_int]
        L.lart_gpu_destroy.argtypes = [_vp]
        L.lart_gpu_set_grid_car.argtypes = [_vp, _i32, _i32, _i32] + [_d] * 6 + [_i32] * 3 + [_vp] * 10
        L.lart_gpu_set_mueller.argtypes = [_vp, _i32] + [_vp] * 7
        L.lart_gpu_set_observers.argtypes = [_vp, _i32, _vp, _vp, _d, _d, _i32, _i32]
        L.lart_gpu_set_observers_heal.argtypes = [_vp, _i32, _vp, _i32]
        L.lart_gpu_set_JPa.argtypes = [_vp, _i32, _i32, _i32, _i32, _vp, _d]
        L.lart_gpu_set_grid_amr.argtypes = [_vp, _i32, _i32, _d, _d] + [_vp] * 15
        L.lart_gpu_set_JPa_amr.argtypes = [_vp, _i32, _i32, _i32, _i32, _i32, _d, _d]
        L.lart_host_last_error.restype = C.c_char_p
        L.lart_host_clumps_generate.argtypes = [_i64, _d, _d, _d, _i32, _d, _i64] + [_vp] * 7
        L.lart_host_clumps_generate_cone.argtypes = [_i64, _d, _d, _d, _i32, _d, _d, _i64] + [_vp] * 7
        L.lart_host_clumps_csr_dims.argtypes = [_i64, _d, _d, _vp, _vp, _vp]
        L.lart_host_clumps_build_csr.argtypes = [_i64, _vp, _vp, _vp, _vp, _i32, _d, _d, _vp, _vp, _vp]
        L.lart_host_clumps_has_overlap.argtypes = [_i64, _vp, _vp, _vp, _vp, _i32, _d, _d, _vp, _vp, _vp]
        L.lart_gpu_set_clump_overlap.argtypes = [_vp, _i32]
        L.lart_host_amr_last_error.restype = C.c_char_p
        L.lart_host_amr_count.argtypes = [_i32, _vp, _vp, _vp, _vp, _d, _d, _vp]
        L.lart_host_amr_build.argtypes = [_i32, _vp, _vp, _vp, _vp, _d, _d, _i32] + [_vp] * 10
        L.lart_gpu_set_clumps.argtypes = [_vp, _i64] + [_vp] * 11 + [_d, _d, _d, _i32, _d, _d, _vp, _vp]
        L.lart_gpu_test_vec2pix.argtypes = [_vp, _i32, _i32, _vp, _vp]
        L.lart_gpu_run.argtypes = [_vp, _i64, _i64, _i64, _u64]
        L.lart_gpu_begin.argtypes = [_vp, _i64, _i64, _i64, _u64]
        L.lart_gpu_step.argtypes = [_vp, _i32, _i32]
        L.lart_gpu_sync.argtypes = [_vp, C.POINTER(_i64), C.POINTER(_i64)]
        L.lart_gpu_kernel_time.argtypes = [_vp, C.POINTER(_d), C.POINTER(_i64)]
        L.lart_gpu_sightline_tau.argtypes = [_vp, _d, _vp, _vp, _vp]
        L.lart_gpu_sightline_time.argtypes = [_vp, C.POINTER(_d)]
        L.lart_gpu_enable_photon_log.argtypes = [_vp, _i32]
        L.lart_gpu_get_photon_log.argtypes = [_vp, _vp, _vp, _i64]
        L.lart_gpu_enable_all_photons.argtypes = [_vp, _i32]
        L.lart_gpu_get_all_photons.argtypes = [_vp, _vp, _i64]
        L.lart_gpu_set_pool.argtypes = [_vp, _i64, _i32]
        L.lart_gpu_pool_info.argtypes = [_vp, C.POINTER(_i64), C.POINTER(_i32)]
        L.lart_gpu_test_cell_margin.argtypes = [_vp, C.POINTER(C.c_double), _i64]
        L.lart_gpu_tally_buffer.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_i64)]
        L.lart_gpu_zero_tallies.argtypes = [_vp]
        L.lart_gpu_get_tallies.argtypes = [_vp, C.POINTER(LartTallies)]
        L.lart_gpu_compact.argtypes = [_vp]
        L.lart_gpu_set_auto_compact.argtypes = [_vp, _i32]
        L.lart_gpu_live_lanes.argtypes = [_vp, C.POINTER(_i64), C.POINTER(_i64)]
        L.lart_gpu_comm_unique_id.argtypes = [_vp]
        L.lart_gpu_comm_init.argtypes = [_vp, _vp, _i32, _i32]
        L.lart_gpu_comm_info.argtypes = [_vp, C.POINTER(_i32), C.POINTER(_i32)]
        L.lart_gpu_comm_destroy.argtypes = [_vp]
        L.lart_gpu_reduce_tallies.argtypes = [_vp, _i32]
        L.lart_gpu_reduce_tallies_multi.argtypes = [_vp, _i32, _i32]
        L.lart_gpu_run_multi.argtypes = [_vp, _i32, _i64, _u64, _i32]
        L.lart_host_entropy_seed.argtypes = [C.POINTER(_u64)]
        L.lart_gpu_test_voigt.argtypes = [_vp, _i32, _vp, _vp, _vp]
        L.lart_gpu_test_rand.argtypes = [_vp, _i32, _u64, _u64, _i32, _d, _d, _vp]
        L.lart_gpu_test_raytrace.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp, _vp]
        L.lart_gpu_test_scatter.argtypes = [_vp, _i32, _vp, _vp, _u64, _vp, _vp]
        L.lart_gpu_test_trace.argtypes = [_vp, _i64, _i64, _i64, _u64, _vp, _i32, _i32, _vp, _vp]
        L.lart_gpu_test_event.argtypes = [_vp, _i32, _vp, _u64, _vp, _vp]
        _lib = L
    return _lib


def line_lookup(line_id):
    """lart_host_line_lookup: the multiplet record of a par%line_id, or None when the id is not one of line types 4, 5, 6"""
    m = LartLineMultiplet()
    L = lib(); L.lart_host_line_lookup.argtypes = [C.c_char_p, _vp]
    return m if L.lart_host_line_lookup(line_id.encode(), C.byref(m)) == 0 else None


def pack_multiplet(m):
    """the record as the 98 numbers tests/golden/multiplets.npz holds (oracle 'line_multiplet' / ref_get_multiplet order)"""
    v = [m.line_type, m.nup] + list(m.ndown) + list(m.f12) + list(m.delE_Hz) + list(m.damping_up)
    for i in range(3):
        for d in range(4):
            real = i < m.nup and d < m.ndown[i] and m.ndown[i] > 1
            v += [m.A21[i][d], m.Elow_Hz[i][d], m.E1[i][d], m.E2[i][d], m.E3[i][d], m.P_down[i][d] if real else 0.0, m.A_down[i][d] if real else 0.0]
    return np.array(v, dtype=np.float64)


class LartError(RuntimeError):
    pass


def entropy_seed():
    """lart_host_entropy_seed: the Philox key for par%iseed = 0 (the reference seeds from /dev/urandom then, random_mt.f90:889-947)"""
    v = _u64()
    if lib().lart_host_entropy_seed(C.byref(v)) != 0:
        raise LartError("lart_host_entropy_seed failed")
    return int(v.value)


def comm_unique_id():
    """lart_gpu_comm_unique_id: the 128 bytes rank 0 shares with the other ranks before lart_gpu_comm_init"""
    b = (C.c_uint8 * 128)()
    _chk(lib().lart_gpu_comm_unique_id(b))
    return bytes(b)


def run_multi(contexts, nphotons, seed, max_events=4096):
    """lart_gpu_run_multi: one host thread, context r transports the photons r+1, r+1+n, ..."""
    arr = (_vp * len(contexts))(*[c.h for c in contexts])
    _chk(lib().lart_gpu_run_multi(arr, len(contexts), int(nphotons), int(seed), int(max_events)))


def reduce_tallies_multi(contexts, root=0):
    """lart_gpu_reduce_tallies_multi: ncclReduce(sum, f64) of the fused tally buffers of the contexts of this process"""
    arr = (_vp * len(contexts))(*[c.h for c in contexts])
    _chk(lib().lart_gpu_reduce_tallies_multi(arr, len(contexts), int(root)))


def _chk(rc):
    if rc != 0:
        raise LartError(lib().lart_gpu_last_error().decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _f64(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))


class Context:
    """One GPU.  Thin, argument-for-argument wrapper of the lart_gpu_* entry points."""

    def __init__(self, params: LartParams, device=0):
        self.h = _vp()
        self.par = params
        _chk(lib().lart_gpu_create(C.byref(self.h), C.byref(params), device))
        self.nobs = self.nxim = self.nyim = 0
        self.nbins = 0; self.jpa = (0, 0, 0)

    def close(self):
        if getattr(self, "h", None):
            lib().lart_gpu_destroy(self.h); self.h = None

    __del__ = close

    def set_grid_car(self, nx, ny, nz, xmin, ymin, zmin, dx, dy, dz, i0, j0, k0, xface, yface, zface,
                     rhokap, voigt_a, Dfreq, vfx, vfy, vfz, rhokapD=None):
        arrs = [_f64(a) for a in (xface, yface, zface, rhokap, voigt_a, Dfreq, vfx, vfy, vfz, rhokapD)]
        nc = nx * ny * nz
        for a, n in zip(arrs, (nx + 1, ny + 1, nz + 1) + (nc,) * 7):
            if a is not None and a.size != n:
                raise ValueError("table size %d != %d" % (a.size, n))
        _chk(lib().lart_gpu_set_grid_car(self.h, nx, ny, nz, xmin, ymin, zmin, dx, dy, dz, i0, j0, k0, *[_p(a) for a in arrs]))

    def set_line_multiplet(self, m):
        _chk(lib().lart_gpu_set_line_multiplet(self.h, C.byref(m)))

    def set_mueller(self, coss, S11, S12, S33, S34, phase_PDF, alias):
        n = len(coss)
        a = [_f64(x) for x in (coss, S11, S12, S33, S34, phase_PDF)]
        al = np.ascontiguousarray(alias, dtype=np.int32)
        if a[5].size != n - 1 or al.size != n - 1:
            raise ValueError("phase_PDF / alias must have nPDF-1 entries")
        _chk(lib().lart_gpu_set_mueller(self.h, n, *[_p(x) for x in a], _p(al)))

    def set_observers(self, xyz, rmatrix, dxim, dyim, nxim, nyim):
        xyz = _f64(np.asarray(xyz).reshape(-1)); rm = _f64(np.asarray(rmatrix).reshape(-1))
        nobs = xyz.size // 3
        if rm.size != 9 * nobs:
            raise ValueError("rmatrix must have 9*nobs entries")
        _chk(lib().lart_gpu_set_observers(self.h, nobs, _p(xyz), _p(rm), dxim, dyim, nxim, nyim))
        self.nobs, self.nxim, self.nyim = nobs, nxim, nyim

    def set_observers_heal(self, xyz, nside):
        xyz = _f64(np.asarray(xyz).reshape(-1)); nobs = xyz.size // 3
        _chk(lib().lart_gpu_set_observers_heal(self.h, nobs, _p(xyz), nside))
        self.nobs, self.nxim, self.nyim = nobs, 12 * nside * nside, 1

    def set_JPa(self, calc_J, calc_P, calc_Pnew, nbins, ind_bin, cross0):
        ind = np.ascontiguousarray(np.asarray(ind_bin, dtype=np.int32).ravel(order="F"))
        _chk(lib().lart_gpu_set_JPa(self.h, int(calc_J), int(calc_P), int(calc_Pnew), nbins, _p(ind), cross0))
        self.nbins = nbins; self.jpa = (int(calc_J), int(calc_P), int(calc_Pnew))

    def set_grid_amr(self, lo, hi, cx, cy, cz, ch, children, neighbor, ileaf, icell_of_leaf,
                     rhokap, voigt_a, Dfreq, vfx, vfy, vfz, rhokapD=None):
        """octree tables as amr_grid_type holds them (children (8,ncells), neighbor (6,ncells), 1-based, 0 = none)"""
        ncells, nleaf = int(np.asarray(ileaf).size), int(np.asarray(icell_of_leaf).size)
        geo = [_f64(a) for a in (cx, cy, cz, ch)]
        ch8 = np.ascontiguousarray(np.asarray(children, dtype=np.int32).ravel(order="F"))
        nb6 = np.ascontiguousarray(np.asarray(neighbor, dtype=np.int32).ravel(order="F"))
        il = np.ascontiguousarray(ileaf, dtype=np.int32); icl = np.ascontiguousarray(icell_of_leaf, dtype=np.int32)
        med = [_f64(a) for a in (rhokap, voigt_a, Dfreq, vfx, vfy, vfz, rhokapD)]
        if any(a.size != ncells for a in geo) or ch8.size != 8 * ncells or nb6.size != 6 * ncells:
            raise ValueError("tree tables must have ncells (x8 children, x6 neighbor) entries")
        if any(a is not None and a.size != nleaf for a in med):
            raise ValueError("medium tables must have nleaf entries")
        _chk(lib().lart_gpu_set_grid_amr(self.h, ncells, nleaf, lo, hi, *[_p(a) for a in geo], _p(ch8), _p(nb6), _p(il), _p(icl),
                                         *[_p(a) for a in med]))

    def set_JPa_amr(self, calc_J, calc_P, calc_Pnew, geometry_JPa, nbins, dbin, cross0):
        _chk(lib().lart_gpu_set_JPa_amr(self.h, int(calc_J), int(calc_P), int(calc_Pnew), int(geometry_JPa), int(nbins), float(dbin), cross0))
        self.nbins = nbins; self.jpa = (int(calc_J), int(calc_P), int(calc_Pnew))

    def set_clumps(self, x, y, z, vx, vy, vz, radius, rhokap, rhokapD, voigt_a, Dfreq, sphere_R, Dfreq_ref, voigt_a_ref,
                   cgx, cg_min, cg_d, cg_start, cg_list):
        """clump population + CSR grid as clump_mod.f90 holds them (1-based cg_start / cg_list)"""
        a = [_f64(v) for v in (x, y, z, vx, vy, vz, radius, rhokap, rhokapD, voigt_a, Dfreq)]
        n = a[0].size
        if any(v is not None and v.size != n for v in a):
            raise ValueError("clump tables must all have n entries")
        st = np.ascontiguousarray(cg_start, dtype=np.int32); li = np.ascontiguousarray(cg_list, dtype=np.int32)
        if st.size != cgx ** 3 + 1 or li.size < st[-1] - 1:
            raise ValueError("cg_start must have cgx^3+1 entries and cg_list cg_start[-1]-1")
        _chk(lib().lart_gpu_set_clumps(self.h, n, *[_p(v) for v in a], sphere_R, Dfreq_ref, voigt_a_ref, cgx, cg_min, cg_d, _p(st), _p(li)))

    def set_clump_overlap(self, has_overlap):
        _chk(lib().lart_gpu_set_clump_overlap(self.h, int(bool(has_overlap))))

    def set_pool(self, nlanes=0, block=0):
        _chk(lib().lart_gpu_set_pool(self.h, nlanes, block))

    def pool_info(self):
        n, b = _i64(), _i32()
        _chk(lib().lart_gpu_pool_info(self.h, C.byref(n), C.byref(b)))
        return n.value, b.value

    def cell_margin(self, n):
        """Cell::pad of the n cells / leaves: the uniform cube (cells) or margin (length) the peel prefilter uses"""
        out = np.zeros(int(n))
        _chk(lib().lart_gpu_test_cell_margin(self.h, out.ctypes.data_as(C.POINTER(C.c_double)), int(n)))
        return out

    def compact(self):
        _chk(lib().lart_gpu_compact(self.h))

    def set_auto_compact(self, on=True):
        _chk(lib().lart_gpu_set_auto_compact(self.h, int(on)))

    def live_lanes(self):
        n, k = _i64(), _i64()
        _chk(lib().lart_gpu_live_lanes(self.h, C.byref(n), C.byref(k)))
        return n.value, k.value

    def comm_init(self, unique_id, nranks, rank):
        b = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _chk(lib().lart_gpu_comm_init(self.h, b, int(nranks), int(rank)))

    def comm_info(self):
        n, r = _i32(), _i32()
        _chk(lib().lart_gpu_comm_info(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def comm_destroy(self):
        _chk(lib().lart_gpu_comm_destroy(self.h))

    def reduce_tallies(self, root=0):
        _chk(lib().lart_gpu_reduce_tallies(self.h, int(root)))

    def enable_photon_log(self, on=True):
        _chk(lib().lart_gpu_enable_photon_log(self.h, int(on)))

    def enable_all_photons(self, on=True):
        _chk(lib().lart_gpu_enable_all_photons(self.h, int(on)))

    def all_photons(self, count):
        """par%save_all_photons: dict of the ten columns write_output_allph writes (r, xfreq1, xfreq2, NSCATT_gas, NSCATT_dust, I, Q, U, V, r0)"""
        a = np.zeros((10, count))
        _chk(lib().lart_gpu_get_all_photons(self.h, _p(a), count))
        return dict(zip(("r", "xfreq1", "xfreq2", "NSCATT_gas", "NSCATT_dust", "I", "Q", "U", "V", "r0"), a))

    def run(self, count, first_id=1, stride=1, seed=1234):
        _chk(lib().lart_gpu_run(self.h, first_id, stride, int(count), seed))

    def begin(self, count, first_id=1, stride=1, seed=1234):
        _chk(lib().lart_gpu_begin(self.h, first_id, stride, int(count), seed))

    def step(self, nsteps=1, max_events=1024):
        _chk(lib().lart_gpu_step(self.h, nsteps, max_events))

    def sync(self):
        d, a = _i64(), _i64()
        _chk(lib().lart_gpu_sync(self.h, C.byref(d), C.byref(a)))
        return d.value, a.value

    def kernel_time(self):
        ms, n = _d(), _i64()
        _chk(lib().lart_gpu_kernel_time(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def sightline_tau(self, cross0):
        """make_sightline_tau: tau_gas (nxfreq,nxim,nyim,nobs), N_gas (nxim,nyim,nobs), tau_dust (nxim,nyim,nobs), Fortran order"""
        nxf = int(self.par.nxfreq)
        tg = np.zeros((nxf, self.nxim, self.nyim, self.nobs), order="F"); ng = np.zeros((self.nxim, self.nyim, self.nobs), order="F")
        td = np.zeros((self.nxim, self.nyim, self.nobs), order="F")
        _chk(lib().lart_gpu_sightline_tau(self.h, float(cross0), tg.ctypes.data_as(_vp), ng.ctypes.data_as(_vp), td.ctypes.data_as(_vp)))
        return tg, ng, td

    def sightline_time(self):
        ms = _d()
        _chk(lib().lart_gpu_sightline_time(self.h, C.byref(ms)))
        return ms.value

    def photon_log(self, count):
        a, b = np.zeros(count), np.zeros(count)
        _chk(lib().lart_gpu_get_photon_log(self.h, _p(a), _p(b), count))
        return a, b

    def tally_buffer(self):
        p, n = _vp(), _i64()
        _chk(lib().lart_gpu_tally_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def zero_tallies(self):
        _chk(lib().lart_gpu_zero_tallies(self.h))

    def tallies(self):
        """dict of numpy arrays (Fortran shapes) + scalars"""
        p = self.par
        nxf, nmu = p.nxfreq, (p.nmu if p.save_Jmu else 0)
        t = LartTallies()
        out = {"Jout": np.zeros(nxf), "Jin": np.zeros(nxf), "Jabs": np.zeros(nxf)}
        if nmu:
            out["Jmu"] = np.zeros((nxf, nmu), order="F")
        if p.save_peeloff and self.nobs:
            names3 = ["scatt", "direc"] + (["I", "Q", "U", "V"] if p.use_stokes else [])
            if p.save_peeloff_3D:
                for n in names3:
                    out[n] = np.zeros((nxf, self.nxim, self.nyim, self.nobs), order="F")
            if p.save_peeloff_2D:
                for n in names3:
                    out[n + "_2D"] = np.zeros((self.nxim, self.nyim, self.nobs), order="F")
            if p.save_direc0 and p.save_peeloff_3D:
                out["direc0"] = np.zeros((nxf, self.nxim, self.nyim, self.nobs), order="F")
            if p.save_direc0 and p.save_peeloff_2D:
                out["direc0_2D"] = np.zeros((self.nxim, self.nyim, self.nobs), order="F")
            if p.save_dust_scattered and p.save_peeloff_3D:
                out["scatt_dust"] = np.zeros((nxf, self.nxim, self.nyim, self.nobs), order="F")
        if self.nbins:
            if self.jpa[0]:
                out["J1"] = np.zeros((nxf, self.nbins), order="F")
            if self.jpa[1]:
                out["P1"] = np.zeros(self.nbins)
            if self.jpa[2]:
                out["P1_new"] = np.zeros(self.nbins)
        if p.line_type == 8:                                               # band 2 (H-alpha): grid%Jout_Ha, grid%Jabs_Ha, observer%peel_Ha, the conversion-rate profile
            out["Jout_Ha"] = np.zeros(p.nxfreq_Ha); out["Jabs_Ha"] = np.zeros(p.nxfreq_Ha)
            if p.save_peeloff and self.nobs and p.save_peeloff_3D:
                out["peel_Ha"] = np.zeros((p.nxfreq_Ha, self.nxim, self.nyim, self.nobs), order="F")
            if self.nbins and self.jpa[1]:
                out["Pconv"] = np.zeros(self.nbins)
        for k, a in out.items():
            setattr(t, k, a.ctypes.data)
        _chk(lib().lart_gpu_get_tallies(self.h, C.byref(t)))
        for k in ("nscatt_gas", "nscatt_dust", "n_events", "n_cellsteps", "n_cellsteps_peel", "n_atomics", "nphotons_done", "nphotons_active"):
            out[k] = getattr(t, k)
        aux = [t.aux[i] for i in range(8)]
        if p.line_type == 8:                                               # the weight budget of output_sum_rect.f90:452-467
            out.update(W_conv=aux[0], W_esc1=aux[1], W_abs1=aux[2], W_esc2=aux[3], W_abs2=aux[4])
        if p.h2_nlines > 0:                                                # :470-485
            out.update(W_H2abs=aux[0], W_H2scat=aux[1], W_H2pump=np.array(aux[2:2 + p.h2_nlines]))
        return out

    # ---- device-function probes -------------------------------------------------------------
    def test_voigt(self, x, a):
        x = _f64(x); a = _f64(np.broadcast_to(a, x.shape)); h = np.zeros_like(x)
        _chk(lib().lart_gpu_test_voigt(self.h, x.size, _p(x), _p(a), _p(h))); return h

    def test_rand(self, what, seed, pid, n, p1=0.0, p2=0.0):
        out = np.zeros(n)
        _chk(lib().lart_gpu_test_rand(self.h, what, seed, pid, n, p1, p2, _p(out))); return out

    def test_raytrace(self, kind, variant, r8, ic4, tq=None, shear=None):
        """kind 0 raytrace_to_tau, 1 raytrace_to_edge, 2 raytrace_to_dist; variant 8 (shearing box): photon%vfy_shear goes in through
        `shear` and comes back in the third return value"""
        r = np.ascontiguousarray(r8, dtype=np.float64).copy(); ic = np.ascontiguousarray(ic4, dtype=np.int32).copy()
        n = r.shape[0]; tau = np.zeros(n) if shear is None else np.ascontiguousarray(shear, dtype=np.float64).copy()
        q = None if tq is None else np.ascontiguousarray(tq, dtype=np.float64)
        _chk(lib().lart_gpu_test_raytrace(self.h, 10 * kind + variant, n, _p(r), _p(ic), _p(q), _p(tau)))
        return r, ic, tau

    def test_vec2pix(self, nside, v3):
        v = np.ascontiguousarray(v3, dtype=np.float64); out = np.zeros(v.shape[0], dtype=np.int32)
        _chk(lib().lart_gpu_test_vec2pix(self.h, nside, v.shape[0], _p(v), _p(out))); return out

    def test_trace(self, count, select, cap, seed, first_id=1, stride=1):
        """history probe: runs photons first_id + k*stride (k < count) to the last one; for the ordinals in `select` returns
        (records[len(select), cap, 8], n_records[len(select)]): draws consumed, xfreq, x, y, z, cell code, kz, kind per event"""
        select = np.asarray(select, dtype=np.int64)
        slot = np.full(int(count), -1, dtype=np.int32); slot[select] = np.arange(len(select), dtype=np.int32)
        tr = np.zeros((len(select), int(cap), 8)); n = np.zeros(len(select), dtype=np.int32)
        _chk(lib().lart_gpu_test_trace(self.h, int(first_id), int(stride), int(count), int(seed), _p(slot), len(select), int(cap), _p(tr), _p(n)))
        return tr, n

    def test_event(self, st27, seed, ids, first):
        """lart_gpu_test_event: one event round (flight + scattering) per row of st27 (the oracle's state-record layout); returns the records after the event"""
        st = np.ascontiguousarray(st27, dtype=np.float64).copy().reshape(-1, 27)
        ids = np.ascontiguousarray(ids, dtype=np.uint64); first = np.ascontiguousarray(first, dtype=np.int32)
        _chk(lib().lart_gpu_test_event(self.h, st.shape[0], _p(st), int(seed), _p(ids), _p(first)))
        return st

    def test_scatter(self, st18, ic4, seed, ids):
        st = np.ascontiguousarray(st18, dtype=np.float64).copy(); ic = np.ascontiguousarray(ic4, dtype=np.int32).copy()
        ids = np.ascontiguousarray(ids, dtype=np.uint64); nd = np.zeros(st.shape[0], dtype=np.int64)
        _chk(lib().lart_gpu_test_scatter(self.h, st.shape[0], _p(st), _p(ic), seed, _p(ids), _p(nd)))
        return st, ic, nd
