#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own compiled modules (oracle/_ref/libref.so).

Runs only in the container that mounts /root/reference (the .so is built by `make -C oracle _ref/libref.so`
from the reference sources where they lie).  The fixtures are plain input/output vectors:
    voigt.npz        x, a, H                      voigt_seon2 (voigt_mod.f90:541-733)
    mt19937.npz      seeds, first 64 uniforms     rand_mt (random_mt.f90:579-630)
    samplers.npz     per (x,a): 256 draws of rand_resonance_vz / rand_resonance / HG / gauss / rand_voigt
                     from MT seeds (random_mt.f90:2562-2696, :2974-2993, :3022-3042, :964-988, :3075-3083)
    raytrace_*.npz   random rays through a perturbed 9^3 (plain, xyz-symmetric) / 21-cell slab medium:
                     inputs, outputs of raytrace_to_{tau,edge,dist}_car* (raytrace_car.f90) and the Jout tally
    line.npz         Ly-alpha constants from setup_resonance_line (line_mod.f90:1241-1270)
The medium tables for the ray fixtures are produced by the oracle's own grid builder and then perturbed, and are
stored in the fixture so the test needs nothing but numpy.
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
R = O.Ref
R.lib()
rng = np.random.default_rng(20261003)

np.savez(os.path.join(OUT, "line.npz"), line=np.array(R.line))

# Voigt
x = np.concatenate([np.linspace(-12, 12, 4801), rng.uniform(-40, 40, 2000), [0.0, 1.0, 5.0, 10.0, 0.999999, 4.999999, 9.999999]])
for_a = [1.492e-2, 4.7186e-4, 1.0e-3, 5.0e-2]
xs = np.tile(x, len(for_a)); as_ = np.repeat(for_a, len(x))
np.savez_compressed(os.path.join(OUT, "voigt.npz"), x=xs, a=as_, H=R.voigt(xs, as_))

# MT19937-64 stream
seeds = np.array([1, 1234, 4321, 99, 2**31 - 1, -7], dtype=np.int64)
U = []
for s in seeds:
    R.seed(int(s)); U.append(R.draw("ref_rand_number_v", 64))
np.savez(os.path.join(OUT, "mt19937.npz"), seeds=seeds, u=np.array(U))

# samplers (bit-exact given the MT stream)
xa = [(0.0, 1.492e-2), (0.7, 1.492e-2), (1.0, 1.492e-2), (1.7, 1.492e-2), (2.3, 4.7186e-4), (2.5, 4.7186e-4), (3.0, 1.492e-2),
      (4.0, 4.7186e-4), (8.0, 1.492e-2), (30.0, 1.492e-2), (-2.0, 1.492e-2), (-6.0, 4.7186e-4), (2.45, 1.0e-3), (2.6, 5e-2)]
vz = []
for i, (xx, aa) in enumerate(xa):
    R.seed(100 + i); vz.append(R.draw("ref_rand_resonance_vz_v", 256, xx, aa))
R.seed(7); cosr = R.draw("ref_rand_resonance_v", 256, 1.0)
R.seed(7); cosr2 = R.draw("ref_rand_resonance_v", 256, 0.5)
R.seed(7); hg = R.draw("ref_rand_hg_v", 256, float(np.float32(0.6761)))
R.seed(7); ga = R.draw("ref_rand_gauss_v", 256)
R.seed(7); rv = R.draw("ref_rand_voigt_v", 256, 1.492e-2)
np.savez(os.path.join(OUT, "samplers.npz"), xa=np.array(xa), vz=np.array(vz), cos_E1_1=cosr, cos_E1_05=cosr2, hg=hg, gauss=ga, rand_voigt=rv)


def perturb(o):
    n = [int(o.scalar(k)) for k in ("nx", "ny", "nz")]
    o.set_array("rhokap", o.grid3("rhokap") * rng.uniform(0.5, 1.5, n))
    o.set_array("Dfreq", o.grid3("Dfreq") * rng.uniform(0.8, 1.2, n))
    o.set_array("voigt_a", o.grid3("voigt_a") * rng.uniform(0.8, 1.2, n))
    for k in ("vfx", "vfy", "vfz"):
        o.set_array(k, rng.normal(0, 2.0, n))
    o.set_array("rhokapD", o.grid3("rhokap") * 0.01)


cases = {"car": (0, "nx=9 ny=9 nz=9 rmax=1 taumax=50 temperature=1e4 geometry=rectangle DGR=1 cext_dust=1e-21"),
         "xyzsym_odd": (1, "nx=9 ny=9 nz=9 rmax=1 taumax=50 temperature=1e4 xyz_symmetry=1 geometry=rectangle"),
         "xyzsym_even": (1, "nx=8 ny=8 nz=8 rmax=1 taumax=50 temperature=1e4 xyz_symmetry=1 geometry=rectangle"),
         "zonly": (2, "nx=1 ny=1 nz=21 xy_periodic=1 taumax=50 temperature=1e4 geometry=rectangle")}
for name, (variant, params) in cases.items():
    o = O.Oracle(params); perturb(o); R.set_grid_from(o)
    g = o.scalar
    nx, ny, nz = int(g("nx")), int(g("ny")), int(g("nz"))
    N = 600
    rin, icin, tauin, rout, icout, tedge, dist, tdist = [], [], [], [], [], [], [], []
    for it in range(N):
        ic = [int(rng.integers(1, nx + 1)), int(rng.integers(1, ny + 1)), int(rng.integers(1, nz + 1))]
        pos = [g("xmin") + (ic[0] - 1 + rng.uniform()) * g("dx"), g("ymin") + (ic[1] - 1 + rng.uniform()) * g("dy"), g("zmin") + (ic[2] - 1 + rng.uniform()) * g("dz")]
        if it % 7 == 0:
            ax = int(rng.integers(0, 3)); pos[ax] = o.array(["xface", "yface", "zface"][ax])[ic[ax] - 1]
        mu = rng.uniform(-1, 1); ph = rng.uniform(0, 2 * np.pi); st = np.sqrt(1 - mu * mu)
        k = [st * np.cos(ph), st * np.sin(ph), mu]
        if it % 11 == 0:
            k = [0.0, 0.0, 1.0 if mu > 0 else -1.0]
        r = pos + k + [rng.normal(0, 3.0), 1.0]; icc = ic + [1]
        tau = rng.exponential(1.0) * (10 if it % 3 else 0.3)
        rr, icr = R.raytrace_to_tau(variant, r, icc, tau)
        rin.append(r); icin.append(icc); tauin.append(tau); rout.append(rr); icout.append(icr)
        tedge.append(R.raytrace_to_edge(variant, r, icc))
        d = rng.uniform(0, 2); dist.append(d); tdist.append(R.raytrace_to_dist(r, icc, d) if variant == 0 else np.nan)
    J = np.zeros(int(g("nxfreq"))); R.lib().ref_get_Jout(J.ctypes.data)
    np.savez_compressed(os.path.join(OUT, "raytrace_%s.npz" % name), params=params, variant=variant,
                        rhokap=o.array("rhokap"), voigt_a=o.array("voigt_a"), Dfreq=o.array("Dfreq"), vfx=o.array("vfx"), vfy=o.array("vfy"),
                        vfz=o.array("vfz"), rhokapD=o.array("rhokapD"),
                        r_in=np.array(rin), ic_in=np.array(icin), tau_in=np.array(tauin), r_out=np.array(rout), ic_out=np.array(icout),
                        tau_edge=np.array(tedge), dist=np.array(dist), tau_dist=np.array(tdist), Jout=J)
    print(name, "escaped", int(np.sum(np.array(icout)[:, 3] == 0)), "Jout sum", J.sum())

# J(x) / P_alpha(new) path-length deposits compiled into the reference's raytrace_car.f90 by -DCALCJ -DCALCPnew
RC = O.RefCalc
RC.lib()
jcases = {"car_sph": (0, "nx=9 ny=9 nz=9 rmax=1 taumax=50 temperature=1e4"),
          "xyzsym_sph": (1, "nx=9 ny=9 nz=9 rmax=1 taumax=50 temperature=1e4 xyz_symmetry=1"),
          "zonly_plane": (2, "nx=1 ny=1 nz=21 xy_periodic=1 taumax=50 temperature=1e4 geometry_JPa=-1 rmax=1")}
for name, (variant, params) in jcases.items():
    o = O.Oracle(params + " calc_J=1 calc_Pnew=1"); perturb(o); RC.set_grid_from(o)
    g = o.scalar
    nx, ny, nz, nb, nxf = int(g("nx")), int(g("ny")), int(g("nz")), int(g("nbins")), int(g("nxfreq"))
    geo = int(g("geometry_JPa"))
    ind = np.maximum(o.array("ind_bin"), 1).astype(np.int32)          # cells outside every bin carry no gas; keep the index legal
    RC.set_JPa(1 if geo == 1 else geo, nb, ind)
    rin, icin, tauin = [], [], []
    for it in range(400):
        ic = [int(rng.integers(1, nx + 1)), int(rng.integers(1, ny + 1)), int(rng.integers(1, nz + 1))]
        pos = [g("xmin") + (ic[0] - 1 + rng.uniform()) * g("dx"), g("ymin") + (ic[1] - 1 + rng.uniform()) * g("dy"), g("zmin") + (ic[2] - 1 + rng.uniform()) * g("dz")]
        mu = rng.uniform(-1, 1); ph = rng.uniform(0, 2 * np.pi); st = np.sqrt(1 - mu * mu)
        r = pos + [st * np.cos(ph), st * np.sin(ph), mu] + [rng.normal(0, 3.0), rng.uniform(0.2, 1.0)]; icc = ic + [1]
        tau = rng.exponential(1.0) * (10 if it % 3 else 0.3)
        RC.raytrace_to_tau(variant, r, icc, tau)
        rin.append(r); icin.append(icc); tauin.append(tau)
    J1, P1n = RC.get_JP(nxf, nb)
    np.savez_compressed(os.path.join(OUT, "jpa_%s.npz" % name), params=params + " calc_J=1 calc_Pnew=1", variant=variant,
                        rhokap=o.array("rhokap"), voigt_a=o.array("voigt_a"), Dfreq=o.array("Dfreq"), vfx=o.array("vfx"), vfy=o.array("vfy"),
                        vfz=o.array("vfz"), r_in=np.array(rin), ic_in=np.array(icin), tau_in=np.array(tauin), J1=J1, P1_new=P1n, geometry_JPa=geo)
    print("jpa", name, "geometry", geo, "nbins", nb, "J1 sum", J1.sum(), "P1_new sum", P1n.sum())

# HEALPix pixel index (healpix.f90:29-112)
vv = rng.normal(size=(4000, 3)); vv[::50, 2] = 0.0; vv[1::50, :2] = 0.0
pix = {}
ip = __import__("ctypes").c_int()
for nside in (1, 2, 8, 64):
    out = []
    for v in vv:
        R.lib().ref_vec2pix(nside, float(v[0]), float(v[1]), float(v[2]), __import__("ctypes").byref(ip)); out.append(ip.value)
    pix["nside%d" % nside] = np.array(out, dtype=np.int32)
np.savez_compressed(os.path.join(OUT, "vec2pix.npz"), v=vv, **pix)

# alias table + in-bin sampling (random_mt.f90:2016-2238) on the Mueller phase function shape
mu_file = "/root/reference/data/mueller_Lyalpha.dat"
tab = np.loadtxt(mu_file, skiprows=3)
coss, S11 = tab[:, 0].copy(), tab[:, 1].copy()
pdf = (S11[:-1] + S11[1:]) / 2; pdf = pdf / pdf.sum()
pa = pdf.copy(); al = np.zeros(len(pa), dtype=np.int32)
R.lib().ref_alias_setup(len(pa), pa.ctypes.data, al.ctypes.data)
R.seed(5); xo = np.zeros(512)
R.lib().ref_rand_alias_linear_v(512, len(pa), pa.ctypes.data, al.ctypes.data, coss.ctypes.data, S11.ctypes.data, xo.ctypes.data)
yn = np.zeros(1); ys = []
xq = rng.uniform(-1.05, 1.05, 200)
for v in xq:
    R.lib().ref_interp_eq(len(coss), coss.ctypes.data, S11.ctypes.data, float(v), yn.ctypes.data); ys.append(yn[0])
np.savez(os.path.join(OUT, "alias_mueller.npz"), pdf=pdf, prob_alias=pa, alias=al, draws=xo, xq=xq, yq=np.array(ys))
# the Mueller table is a runtime input of the dust configs (par%scatt_mat_file): keep its numbers as arrays
hdr = [float(t) for t in open(mu_file).read().splitlines()[1].split()[:4]]
np.savez(os.path.join(OUT, "mueller_Lyalpha.npz"), header=np.array(hdr), table=tab)
print("golden fixtures written to", OUT)
