#!/usr/bin/env python3
"""Generate tests/golden/sightline_car.npz and pix2vec.npz from the reference's own compiled modules (oracle/_ref/libref.so):
    sightline_car.npz  random rays through a perturbed 9^3 medium with dust: raytrace_to_edge_car_tau_gas / _column
                       (raytrace_car.f90:1236-1423) and raytrace_to_dist_tau_gas_car / _column_car (:197-408)
    pix2vec.npz        pixel centres of pix2vec (healpix.f90:114-187) for nside = 1, 2, 4, 16, 64
Separate from gen_golden.py so the older fixtures (one shared random stream) stay byte-identical."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden")
R = O.Ref
R.lib()
rng = np.random.default_rng(20261004)
params = "nx=9 ny=9 nz=9 rmax=1 taumax=50 temperature=1e4 geometry=rectangle DGR=1 cext_dust=1e-21"
o = O.Oracle(params)
n = [9, 9, 9]
o.set_array("rhokap", o.grid3("rhokap") * rng.uniform(0.5, 1.5, n))
o.set_array("Dfreq", o.grid3("Dfreq") * rng.uniform(0.8, 1.2, n))
o.set_array("voigt_a", o.grid3("voigt_a") * rng.uniform(0.8, 1.2, n))
for k in ("vfx", "vfy", "vfz"):
    o.set_array(k, rng.normal(0, 2.0, n))
o.set_array("rhokapD", o.grid3("rhokap") * 0.01)
R.set_grid_from(o)
g = o.scalar
rin, icin, dist, tg_e, tg_d, col_e, col_d = [], [], [], [], [], [], []
for it in range(500):
    ic = [int(rng.integers(1, 10)) for _ in range(3)]
    pos = [g("xmin") + (ic[0] - 1 + rng.uniform()) * g("dx"), g("ymin") + (ic[1] - 1 + rng.uniform()) * g("dy"), g("zmin") + (ic[2] - 1 + rng.uniform()) * g("dz")]
    mu = rng.uniform(-1, 1); ph = rng.uniform(0, 2 * np.pi); st = np.sqrt(1 - mu * mu)
    k = [st * np.cos(ph), st * np.sin(ph), mu]
    if it % 5 == 0:          # start on a boundary face moving inwards, like the sight-line set-up does
        ax = int(rng.integers(0, 3)); hi = bool(rng.integers(0, 2))
        pos[ax] = o.array(["xface", "yface", "zface"][ax])[9 if hi else 0]; ic[ax] = 9 if hi else 1
        k[ax] = -abs(k[ax]) if hi else abs(k[ax])
    r = pos + k + [rng.normal(0, 3.0), 1.0]; icc = ic + [1]
    d = rng.uniform(0, 2)
    rin.append(r); icin.append(icc); dist.append(d)
    tg_e.append(R.raytrace_tau_gas(r, icc)); tg_d.append(R.raytrace_tau_gas(r, icc, d))
    col_e.append(R.raytrace_column(r, icc)); col_d.append(R.raytrace_column(r, icc, d))
np.savez_compressed(os.path.join(OUT, "sightline_car.npz"), params=params,
                    rhokap=o.array("rhokap"), voigt_a=o.array("voigt_a"), Dfreq=o.array("Dfreq"), vfx=o.array("vfx"), vfy=o.array("vfy"),
                    vfz=o.array("vfz"), rhokapD=o.array("rhokapD"), r_in=np.array(rin), ic_in=np.array(icin), dist=np.array(dist),
                    tau_gas_edge=np.array(tg_e), tau_gas_dist=np.array(tg_d), column_edge=np.array(col_e), column_dist=np.array(col_d))
pv = {}
for ns in (1, 2, 4, 16, 64):
    pv["nside%d" % ns] = np.array([R.pix2vec(ns, ip) for ip in range(1, 12 * ns * ns + 1)])
np.savez_compressed(os.path.join(OUT, "pix2vec.npz"), **pv)
print("sightline_car: tau range", min(tg_e), max(tg_e), " N range", min(c[0] for c in col_e), max(c[0] for c in col_e))
