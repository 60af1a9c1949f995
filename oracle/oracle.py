"""TEST INFRASTRUCTURE: ctypes bindings of oracle/liblart_oracle.so (the CPU restatement) and, when it has
been built in the container that mounts /root/reference, of oracle/_ref/libref.so (the reference's own
MPI-free modules).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_d, _i, _u64, _i64, _vp = C.c_double, C.c_int, C.c_uint64, C.c_int64, C.c_void_p


def build(ref=True):
    tgt = ["liblart_oracle.so"]
    if ref and os.path.exists("/root/reference/src/define.f90"):
        tgt.append("_ref/libref.so")
    subprocess.check_call(["make", "-s", "-C", HERE] + tgt)


def usable_cores():
    """cores this process may really use: the cgroup CPU quota when there is one (a 1-GPU box gets a 16-core share of a 256-thread
    host; 256 spinning OpenMP threads on 16 cores crawl), else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 64))


def _ptr(a):
    return a.ctypes.data_as(_vp)


class Oracle:
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            p = os.path.join(HERE, "liblart_oracle.so")
            if not os.path.exists(p):
                build(ref=False)
            L = C.CDLL(p)
            L.orc_create.restype = _vp; L.orc_create.argtypes = [C.c_char_p]
            L.orc_destroy.argtypes = [_vp]
            L.orc_last_error.restype = C.c_char_p
            L.orc_run.argtypes = [_vp, _i64, _i64, _i64, _i, _u64, _i]; L.orc_run.restype = _i
            L.orc_get_scalar.restype = _d; L.orc_get_scalar.argtypes = [_vp, C.c_char_p]
            L.orc_get_array.restype = _i64; L.orc_get_array.argtypes = [_vp, C.c_char_p, _vp, _i64]
            L.orc_zero_tallies.argtypes = [_vp]
            L.orc_set_array.restype = _i64; L.orc_set_array.argtypes = [_vp, C.c_char_p, _vp, _i64]
            L.orc_normalized_Jout.argtypes = [_vp, _vp, _i]
            L.orc_get_photon_log.restype = _i64; L.orc_get_photon_log.argtypes = [_vp, _vp, _vp, _i64]
            L.orc_trace.restype = _i64; L.orc_trace.argtypes = [_vp, _u64, _u64, _i64, _vp]
            L.orc_trace_state.restype = _i64; L.orc_trace_state.argtypes = [_vp, _u64, _u64, _i64, _vp]
            L.orc_voigt.restype = _d; L.orc_voigt.argtypes = [_d, _d]
            L.orc_voigt_v.argtypes = [_i, _vp, _vp, _vp]
            L.orc_line_constants.argtypes = [_vp] * 5
            L.orc_rand_number_v.argtypes = [_i, _u64, _u64, _i, _vp]
            L.orc_rand_resonance_vz_v.argtypes = [_i, _u64, _u64, _i, _d, _d, _vp]
            L.orc_rand_resonance_v.argtypes = [_i, _u64, _u64, _i, _d, _vp]
            L.orc_rand_hg_v.argtypes = [_i, _u64, _u64, _i, _d, _vp]
            L.orc_rand_voigt_v.argtypes = [_i, _u64, _u64, _i, _d, _vp]
            L.orc_rand_gauss_v.argtypes = [_i, _u64, _u64, _i, _vp]
            L.orc_philox4x32_10.argtypes = [_vp, _vp, _vp]
            L.orc_raytrace_to_tau.argtypes = [_vp, _i, _vp, _vp, _d]
            L.orc_raytrace_to_edge.argtypes = [_vp, _i, _vp, _vp, _vp]
            L.orc_raytrace_to_tau_shear.argtypes = [_vp, _vp, _vp, _d, _vp]
            L.orc_raytrace_to_tau_band.argtypes = [_vp, _i, _vp, _vp, _d]
            L.orc_raytrace_to_edge_dust_Ha.argtypes = [_vp, _vp, _vp, _vp]
            L.orc_h2_kappa_v.argtypes = [_vp, _i, _vp, _d, _vp]
            L.orc_h2_select_v.argtypes = [_vp, _i, _u64, _u64, _i, _vp, _d, _vp]
            L.orc_raytrace_to_dist.argtypes = [_vp, _vp, _vp, _d, _vp]
            L.orc_amr_find_leaf.argtypes = [_vp, _d, _d, _d]
            L.orc_xcrit_local.restype = _d; L.orc_xcrit_local.argtypes = [_vp, _i, _i, _i, _d, _d, _d]
            L.orc_scatter_event.restype = _i64; L.orc_scatter_event.argtypes = [_vp, _vp, _vp, _u64, _u64, _u64]
            L.orc_sightline_tau.restype = _i; L.orc_sightline_tau.argtypes = [_vp, _vp, _vp, _vp]
            L.orc_raytrace_tau_gas.argtypes = [_vp, _vp, _vp, _d, _vp]
            L.orc_raytrace_column.argtypes = [_vp, _vp, _vp, _d, _vp, _vp]
            L.orc_pix2vec.argtypes = [_i, _i, _vp]
            L.orc_do_resonance_v.argtypes = [_vp, _i, _u64, _u64, _i, _d, _i, _i, _i, _vp]
            L.orc_calc_voigt_v.argtypes = [_vp, _i, _vp, _i, _i, _i, _vp]
            cls._lib = L
        return cls._lib

    def __init__(self, params=None, **kw):
        L = self.lib()
        if params is None:
            params = ""
        if isinstance(params, dict):
            kw = {**params, **kw}; params = ""
        toks = [params] if params else []
        for k, v in kw.items():
            if isinstance(v, bool):
                v = int(v)
            toks.append("%s=%s" % (k, repr(float(v)) if isinstance(v, float) else v))
        self.params = " ".join(toks)
        self.h = L.orc_create(self.params.encode())
        if not self.h:
            raise RuntimeError("orc_create failed: " + L.orc_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib().orc_destroy(self.h); self.h = None

    def scalar(self, name):
        return self.lib().orc_get_scalar(self.h, name.encode())

    def array(self, name, shape=None):
        L = self.lib()
        n = L.orc_get_array(self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        a = np.zeros(n)
        L.orc_get_array(self.h, name.encode(), _ptr(a), n)
        return a.reshape(shape, order="F") if shape else a

    def set_array(self, name, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order='F'))
        if self.lib().orc_set_array(self.h, name.encode(), _ptr(a), a.size) < 0:
            raise KeyError(name)

    def grid3(self, name):
        n = [int(self.scalar(k)) for k in ("nx", "ny", "nz")]
        return self.array(name, tuple(n))

    def run(self, count, first_id=1, stride=1, rng_kind=1, seed=1234, nthreads=None):
        if nthreads is None:
            nthreads = usable_cores()
        return self.lib().orc_run(self.h, first_id, stride, int(count), rng_kind, seed, nthreads)

    def zero(self):
        self.lib().orc_zero_tallies(self.h)

    def normalized_Jout(self):
        n = int(self.scalar("nxfreq")); a = np.zeros(n)
        self.lib().orc_normalized_Jout(self.h, _ptr(a), n); return a

    def photon_log(self):
        L = self.lib(); n = L.orc_get_photon_log(self.h, None, None, 0)
        a, b = np.zeros(n), np.zeros(n); L.orc_get_photon_log(self.h, _ptr(a), _ptr(b), n); return a, b

    def trace(self, seed, pid, cap):
        """history of one photon on the Philox stream (seed, pid): (records[n, 10], n_total) -- draws consumed, xfreq, x, y, z, cell code, kz,
        kind (0 birth, 1 scattered, 2 left / absorbed), smallest decision margin of the event, id of that comparison"""
        out = np.zeros((int(cap), 10)); n = self.lib().orc_trace(self.h, int(seed), int(pid), int(cap), _ptr(out))
        return out[:min(n, cap)], int(n)

    def trace_state(self, seed, pid, at):
        """full packet state at record `at` of the history of photon (seed, pid) -- x y z kx ky kz xfreq wgt m(3) n(3) Q U V nscatt_gas nscatt_dust draws icell jcell kcell
        owner_clump vfy_shear iband kind (lart_oracle.c, ORC_STATE_W = 27) -- and the number of records of the history"""
        out = np.zeros(27); n = self.lib().orc_trace_state(self.h, int(seed), int(pid), int(at), _ptr(out))
        return out, int(n)

    def xfreq_axis(self):
        n = int(self.scalar("nxfreq"))
        return (np.arange(n) + 0.5) * self.scalar("dxfreq") + self.scalar("xfreq_min")

    def raytrace_to_tau(self, variant, r, ic, tau):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32)
        self.lib().orc_raytrace_to_tau(self.h, variant, _ptr(r), _ptr(ic), tau); return r, ic

    def raytrace_to_tau_band(self, iband, r, ic, tau):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32)
        self.lib().orc_raytrace_to_tau_band(self.h, int(iband), _ptr(r), _ptr(ic), tau); return r, ic

    def raytrace_to_edge_dust_Ha(self, r, ic):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32); t = _d()
        self.lib().orc_raytrace_to_edge_dust_Ha(self.h, _ptr(r), _ptr(ic), C.byref(t)); return t.value

    def h2_kappa(self, x, Dfreq):
        x = np.ascontiguousarray(x, dtype=np.float64); k = np.zeros_like(x)
        self.lib().orc_h2_kappa_v(self.h, len(x), _ptr(x), float(Dfreq), _ptr(k)); return k

    def h2_select(self, kind, seed, sid, x, Dfreq):
        x = np.ascontiguousarray(x, dtype=np.float64); il = np.zeros(len(x), dtype=np.int32)
        self.lib().orc_h2_select_v(self.h, kind, seed, sid, len(x), _ptr(x), float(Dfreq), _ptr(il)); return il

    def raytrace_to_tau_shear(self, r, ic, tau, shear=0.0):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32); sh = _d(shear)
        self.lib().orc_raytrace_to_tau_shear(self.h, _ptr(r), _ptr(ic), tau, C.byref(sh)); return r, ic, sh.value

    def raytrace_to_edge(self, variant, r, ic):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32); t = _d()
        self.lib().orc_raytrace_to_edge(self.h, variant, _ptr(r), _ptr(ic), C.byref(t)); return t.value

    def amr_find_leaf(self, x, y, z):
        return int(self.lib().orc_amr_find_leaf(self.h, float(x), float(y), float(z)))

    def raytrace_to_dist(self, r, ic, dist):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32); t = _d()
        self.lib().orc_raytrace_to_dist(self.h, _ptr(r), _ptr(ic), dist, C.byref(t)); return t.value

    def do_resonance(self, kind, seed, sid, n, xfreq, ic, jc, kc):
        out = np.zeros((n, 12)); self.lib().orc_do_resonance_v(self.h, kind, seed, sid, n, float(xfreq), ic, jc, kc, _ptr(out)); return out

    def calc_voigt(self, x, ic, jc, kc):
        x = np.ascontiguousarray(x, dtype=np.float64); h = np.zeros_like(x)
        self.lib().orc_calc_voigt_v(self.h, len(x), _ptr(x), ic, jc, kc, _ptr(h)); return h

    def raytrace_tau_gas(self, r, ic, dist=-1.0):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32); t = _d()
        self.lib().orc_raytrace_tau_gas(self.h, _ptr(r), _ptr(ic), dist, C.byref(t)); return t.value

    def raytrace_column(self, r, ic, dist=-1.0):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32); n = _d(); t = _d()
        self.lib().orc_raytrace_column(self.h, _ptr(r), _ptr(ic), dist, C.byref(n), C.byref(t)); return n.value, t.value

    def sightline_tau(self):
        """tau_gas (nxfreq,nxim,nyim,nobs), N_gas (nxim,nyim,nobs), tau_dust (nxim,nyim,nobs) in Fortran order"""
        ns = int(self.scalar("nside")); nob = int(self.scalar("nobs")); nxf = int(self.scalar("nxfreq"))
        nxi, nyi = (int(self.scalar("npix")), 1) if ns > 0 else (int(self.scalar("nxim")), int(self.scalar("nyim")))
        tg = np.zeros((nxf, nxi, nyi, nob), order="F"); ng = np.zeros((nxi, nyi, nob), order="F"); td = np.zeros((nxi, nyi, nob), order="F")
        if self.lib().orc_sightline_tau(self.h, _ptr(tg), _ptr(ng), _ptr(td)):
            raise RuntimeError(self.lib().orc_last_error().decode())
        return tg, ng, td

    @classmethod
    def pix2vec(cls, nside, ipix):
        v = np.zeros(3); cls.lib().orc_pix2vec(int(nside), int(ipix), _ptr(v)); return v

    def scatter_event(self, st, ic, seed, pid, n0=0):
        st = np.array(st, dtype=np.float64); ic = np.array(ic, dtype=np.int32)
        nd = self.lib().orc_scatter_event(self.h, _ptr(st), _ptr(ic), seed, pid, n0); return st, ic, nd


def voigt(x, a):
    x = np.ascontiguousarray(np.broadcast_to(x, np.broadcast(x, a).shape), dtype=np.float64).ravel()
    a = np.ascontiguousarray(np.broadcast_to(a, x.shape), dtype=np.float64).ravel()
    h = np.zeros_like(x); Oracle.lib().orc_voigt_v(len(x), _ptr(x), _ptr(a), _ptr(h)); return h


def line_constants():
    v = [_d() for _ in range(5)]
    Oracle.lib().orc_line_constants(*[C.byref(t) for t in v]); return [t.value for t in v]


def philox(ctr, key):
    c = np.array(ctr, dtype=np.uint32); k = np.array(key, dtype=np.uint32); o = np.zeros(4, dtype=np.uint32)
    Oracle.lib().orc_philox4x32_10(_ptr(c), _ptr(k), _ptr(o)); return o


def _sampler(fn, kind, seed, stream, n, *args):
    out = np.zeros(n); getattr(Oracle.lib(), fn)(kind, seed, stream, n, *args, _ptr(out)); return out


MT_CONTINUE = 2 ** 64 - 1     # seed value meaning "continue the global MT stream"
rand_number = lambda kind, seed, stream, n: _sampler("orc_rand_number_v", kind, seed, stream, n)
rand_resonance_vz = lambda kind, seed, stream, n, x, a: _sampler("orc_rand_resonance_vz_v", kind, seed, stream, n, x, a)
rand_resonance = lambda kind, seed, stream, n, E1: _sampler("orc_rand_resonance_v", kind, seed, stream, n, E1)
rand_hg = lambda kind, seed, stream, n, g: _sampler("orc_rand_hg_v", kind, seed, stream, n, g)
rand_voigt = lambda kind, seed, stream, n, a: _sampler("orc_rand_voigt_v", kind, seed, stream, n, a)
rand_gauss = lambda kind, seed, stream, n: _sampler("orc_rand_gauss_v", kind, seed, stream, n)


class Ref:
    """oracle/_ref/libref.so -- only exists in the container that mounts /root/reference."""
    _lib = None
    SO = "libref.so"

    @classmethod
    def available(cls):
        return os.path.exists(os.path.join(HERE, "_ref", cls.SO))

    @classmethod
    def lib(cls):
        if cls.__dict__.get("_lib") is None:
            L = C.CDLL(os.path.join(HERE, "_ref", cls.SO))
            L.ref_voigt.restype = _d; L.ref_voigt.argtypes = [_d, _d]
            L.ref_voigt_v.argtypes = [_i, _vp, _vp, _vp]
            L.ref_seed.argtypes = [_i]
            L.ref_rand_number_v.argtypes = [_i, _vp]
            L.ref_rand_resonance_vz_v.argtypes = [_i, _d, _d, _vp]
            L.ref_rand_resonance_v.argtypes = [_i, _d, _vp]
            L.ref_rand_hg_v.argtypes = [_i, _d, _vp]
            L.ref_rand_voigt_v.argtypes = [_i, _d, _vp]
            L.ref_rand_gauss_v.argtypes = [_i, _vp]
            L.ref_alias_setup.argtypes = [_i, _vp, _vp]
            L.ref_rand_alias_linear_v.argtypes = [_i, _i, _vp, _vp, _vp, _vp, _vp]
            L.ref_interp_eq.argtypes = [_i, _vp, _vp, _d, _vp]
            L.ref_vec2pix.argtypes = [_i, _d, _d, _d, _vp]
            L.ref_do_resonance1_v.argtypes = [_i, _d, _i, _i, _i, _vp, _vp, _vp, _vp]
            L.ref_set_par.argtypes = [_d, _i, _i, _i]
            L.ref_set_grid.argtypes = [_i, _i, _i] + [_d] * 6 + [_i] * 3 + [_d] * 3 + [_i] + [_vp] * 7
            L.ref_raytrace_to_tau.argtypes = [_i, _vp, _vp, _d]
            L.ref_raytrace_to_edge.argtypes = [_i, _vp, _vp, _vp]
            L.ref_raytrace_to_dist.argtypes = [_vp, _vp, _d, _vp]
            L.ref_get_Jout.argtypes = [_vp]
            if hasattr(L, "ref_raytrace_to_tau_shear"):
                L.ref_raytrace_to_tau_shear.argtypes = [_vp, _vp, _d, _d, _vp]
                L.ref_do_resonance8_v.argtypes = [_i, _d, _i, _i, _i, _vp]
                L.ref_set_band2.argtypes = [_d, _d, _i, _d]
                L.ref_raytrace_to_tau_band.argtypes = [_i, _vp, _vp, _d]
                L.ref_raytrace_to_edge_dust_Ha.argtypes = [_vp, _vp, _vp]
                L.ref_get_band2.argtypes = [_vp, _vp]
                L.ref_h2_init.argtypes = [C.c_char_p, _d, _d, _i, _i, _d, _vp]
                L.ref_h2_kappa_v.argtypes = [_i, _vp, _d, _vp]
                L.ref_h2_select_v.argtypes = [_i, _vp, _d, _vp]
            if hasattr(L, "ref_set_deuterium"):
                L.ref_set_deuterium.argtypes = [_i, _d, _vp]
                L.ref_do_resonance_HD_v.argtypes = [_i, _d, _i, _i, _i, _vp]
            if hasattr(L, "ref_set_fine_structure"):
                L.ref_set_fine_structure.argtypes = [_i, _vp]
                L.ref_do_resonance2_v.argtypes = [_i, _d, _i, _i, _i, _vp]
                L.ref_do_resonance_ptr_v.argtypes = [_i, _d, _i, _i, _i, _vp]
                L.ref_calc_voigt_v.argtypes = [_i, _vp, _i, _i, _i, _vp]
            if hasattr(L, "ref_raytrace_tau_gas"):
                L.ref_raytrace_tau_gas.argtypes = [_vp, _vp, _d, _vp]
                L.ref_raytrace_column.argtypes = [_vp, _vp, _d, _vp, _vp]
                L.ref_pix2vec.argtypes = [_i, _i, _vp]
            L.ref_get_faces.argtypes = [_vp, _vp, _vp]
            v = [_d() for _ in range(5)]
            L.ref_init_line(*[C.byref(t) for t in v])
            cls.line = [t.value for t in v]
            if hasattr(L, "ref_set_JPa"):
                L.ref_set_JPa.argtypes = [_i, _i, _vp]
                L.ref_get_JP.argtypes = [_i, _vp, _vp]
            cls._lib = L
        return cls.__dict__["_lib"]

    @classmethod
    def voigt(cls, x, a):
        x = np.ascontiguousarray(x, dtype=np.float64).ravel(); a = np.ascontiguousarray(np.broadcast_to(a, x.shape), dtype=np.float64)
        h = np.zeros_like(x); cls.lib().ref_voigt_v(len(x), _ptr(x), _ptr(a), _ptr(h)); return h

    @classmethod
    def seed(cls, s):
        cls.lib().ref_seed(s)

    @classmethod
    def draw(cls, fn, n, *args):
        out = np.zeros(n); getattr(cls.lib(), fn)(n, *args, _ptr(out)); return out

    @classmethod
    def set_grid_from(cls, o):
        """load the oracle-built medium into the reference's grid_type"""
        L = cls.lib()
        g = lambda k: o.scalar(k)
        L.ref_set_par(g("DGR"), int(g("save_Jmu")), int(g("nmu")), int(g("xyz_symmetry")))
        arrs = [np.ascontiguousarray(o.array(k)) for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD")]
        L.ref_set_grid(int(g("nx")), int(g("ny")), int(g("nz")), g("xmin"), g("ymin"), g("zmin"), g("dx"), g("dy"), g("dz"),
                       int(g("i0")), int(g("j0")), int(g("k0")), g("Dfreq_ref"), g("xfreq_min"), g("dxfreq"), int(g("nxfreq")),
                       *[_ptr(a) for a in arrs])
        cls._keep = arrs

    @classmethod
    def raytrace_to_tau(cls, variant, r, ic, tau):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32)
        cls.lib().ref_raytrace_to_tau(variant, _ptr(r), _ptr(ic), tau); return r, ic

    @classmethod
    def raytrace_to_edge(cls, variant, r, ic):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32); t = _d()
        cls.lib().ref_raytrace_to_edge(variant, _ptr(r), _ptr(ic), C.byref(t)); return t.value

    @classmethod
    def raytrace_to_tau_shear(cls, r, ic, tau, Omega, shear=0.0):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32); sh = _d(shear)
        cls.lib().ref_raytrace_to_tau_shear(_ptr(r), _ptr(ic), tau, Omega, C.byref(sh)); return r, ic, sh.value

    @classmethod
    def raytrace_to_dist(cls, r, ic, dist):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32); t = _d()
        cls.lib().ref_raytrace_to_dist(_ptr(r), _ptr(ic), dist, C.byref(t)); return t.value


    @classmethod
    def raytrace_tau_gas(cls, r, ic, dist=-1.0):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32); t = _d()
        cls.lib().ref_raytrace_tau_gas(_ptr(r), _ptr(ic), dist, C.byref(t)); return t.value

    @classmethod
    def raytrace_column(cls, r, ic, dist=-1.0):
        r = np.array(r, dtype=np.float64); ic = np.array(ic, dtype=np.int32); n = _d(); t = _d()
        cls.lib().ref_raytrace_column(_ptr(r), _ptr(ic), dist, C.byref(n), C.byref(t)); return n.value, t.value

    @classmethod
    def pix2vec(cls, nside, ipix):
        v = np.zeros(3); cls.lib().ref_pix2vec(int(nside), int(ipix), _ptr(v)); return v


    @classmethod
    def set_fine_structure(cls, on):
        v = _d(); cls.lib().ref_set_fine_structure(int(on), C.byref(v)); return v.value

    @classmethod
    def do_resonance2(cls, n, xfreq, ic, jc, kc):
        out = np.zeros((n, 12)); cls.lib().ref_do_resonance2_v(n, float(xfreq), ic, jc, kc, _ptr(out)); return out

    @classmethod
    def calc_voigt(cls, x, ic, jc, kc):
        x = np.ascontiguousarray(x, dtype=np.float64); h = np.zeros_like(x)
        cls.lib().ref_calc_voigt_v(len(x), _ptr(x), ic, jc, kc, _ptr(h)); return h


    @classmethod
    def set_line_id(cls, name):
        out = np.zeros(10); cls.lib().ref_set_line_id(name.encode() + b"\0", _ptr(out)); return out

    @classmethod
    def set_HeI_coherent(cls, on):
        cls.lib().ref_set_HeI_coherent(int(on))

    @classmethod
    def get_multiplet(cls):
        out = np.zeros(98); cls.lib().ref_get_multiplet(_ptr(out)); return out

    @classmethod
    def do_resonance_ptr(cls, n, xfreq, ic, jc, kc):
        out = np.zeros((n, 12)); cls.lib().ref_do_resonance_ptr_v(n, float(xfreq), ic, jc, kc, _ptr(out)); return out

    @classmethod
    def set_deuterium(cls, on, D_to_H=1.5e-5):
        out = np.zeros(4); cls.lib().ref_set_deuterium(int(on), float(D_to_H), _ptr(out)); return out

    @classmethod
    def do_resonance_HD(cls, n, xfreq, ic, jc, kc):
        out = np.zeros((n, 5)); cls.lib().ref_do_resonance_HD_v(n, float(xfreq), ic, jc, kc, _ptr(out)); return out


class RefCalc(Ref):
    """the same reference subset compiled with -DCALCJ -DCALCP -DCALCPnew (oracle/_ref/libref_calc.so)"""
    _lib = None
    SO = "libref_calc.so"

    @classmethod
    def set_JPa(cls, geometry_JPa, nbins, ind):
        ind = np.ascontiguousarray(np.asarray(ind, dtype=np.int32).ravel(order="F"))
        cls.lib().ref_set_JPa(geometry_JPa, nbins, _ptr(ind)); cls._keep_ind = ind

    @classmethod
    def get_JP(cls, nxfreq, nbins):
        J = np.zeros(nxfreq * nbins); P = np.zeros(nbins)
        cls.lib().ref_get_JP(nbins, _ptr(J), _ptr(P)); return J.reshape((nxfreq, nbins), order="F"), P
