#!/usr/bin/env python3
"""Generate tests/golden/raytrace_{xysym_odd,xysym_even,xyper}.npz from the reference's own compiled raytrace_car.f90
(oracle/_ref/libref.so): raytrace_to_{tau,edge}_car_xysym (:1951-2250, :783-969) and _car_xyper (:2252-2517, :971-1136) on
perturbed media, with starts on faces / mirror planes / period boundaries and axis-parallel rays.
Separate from gen_golden.py so the older fixtures (one shared random stream) stay byte-identical."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden")
R = O.Ref
R.lib()
rng = np.random.default_rng(20261005)
cases = {"xysym_odd": (5, "nx=7 ny=7 nz=10 xmax=1 ymax=1 zmax=1.5 taumax=50 temperature=1e4 xy_symmetry=1 geometry=rectangle DGR=1 cext_dust=1e-21"),
         "xysym_even": (5, "nx=6 ny=8 nz=9 xmax=1 ymax=1.2 zmax=1 taumax=50 temperature=1e4 xy_symmetry=1 geometry=rectangle"),
         "xyper": (6, "nx=5 ny=7 nz=9 xmax=0.5 ymax=0.8 zmax=1 taumax=30 temperature=1e4 xy_periodic=1 geometry=rectangle DGR=1 cext_dust=1e-21")}
for name, (variant, params) in cases.items():
    o = O.Oracle(params)
    n = [int(o.scalar(k)) for k in ("nx", "ny", "nz")]
    o.set_array("rhokap", o.grid3("rhokap") * rng.uniform(0.5, 1.5, n))
    o.set_array("Dfreq", o.grid3("Dfreq") * rng.uniform(0.8, 1.2, n))
    o.set_array("voigt_a", o.grid3("voigt_a") * rng.uniform(0.8, 1.2, n))
    for k in ("vfx", "vfy", "vfz"):
        o.set_array(k, rng.normal(0, 2.0, n))
    o.set_array("rhokapD", o.grid3("rhokap") * 0.01)
    R.set_grid_from(o)
    g = o.scalar
    nx, ny, nz = n
    faces = [o.array("xface"), o.array("yface"), o.array("zface")]
    rin, icin, tauin, rout, icout, tedge = [], [], [], [], [], []
    for it in range(700):
        ic = [int(rng.integers(1, nx + 1)), int(rng.integers(1, ny + 1)), int(rng.integers(1, nz + 1))]
        pos = [g("xmin") + (ic[0] - 1 + rng.uniform()) * g("dx"), g("ymin") + (ic[1] - 1 + rng.uniform()) * g("dy"), g("zmin") + (ic[2] - 1 + rng.uniform()) * g("dz")]
        mu = rng.uniform(-1, 1); ph = rng.uniform(0, 2 * np.pi); st = np.sqrt(1 - mu * mu)
        k = [st * np.cos(ph), st * np.sin(ph), mu]
        if it % 7 == 0:                       # start exactly on a cell face (lower face of the cell)
            ax = int(rng.integers(0, 3)); pos[ax] = faces[ax][ic[ax] - 1]
        if it % 9 == 0:                       # start on the lowest x or y face moving outwards: mirror plane / period boundary
            ax = int(rng.integers(0, 2)); ic[ax] = 1; pos[ax] = faces[ax][0]; k[ax] = -abs(k[ax])
        if it % 11 == 0:
            k = [0.0, 0.0, 1.0 if mu > 0 else -1.0]
        if it % 13 == 0:
            k = [1.0 if mu > 0 else -1.0, 0.0, 0.0]
        r = pos + k + [rng.normal(0, 3.0), 1.0]; icc = ic + [1]
        tau = rng.exponential(1.0) * (10 if it % 3 else 0.3)
        if it % 13 == 0:
            tau = min(tau, 5.0) if variant == 6 else tau    # an x-parallel ray never leaves a periodic box: keep the target reachable
        rr, icr = R.raytrace_to_tau(variant, r, icc, tau)
        rin.append(r); icin.append(icc); tauin.append(tau); rout.append(rr); icout.append(icr)
        tedge.append(R.raytrace_to_edge(variant, r, icc) if not (variant == 6 and it % 13 == 0) else np.nan)
    J = np.zeros(int(g("nxfreq"))); R.lib().ref_get_Jout(J.ctypes.data)
    np.savez_compressed(os.path.join(OUT, "raytrace_%s.npz" % name), params=params, variant=variant,
                        rhokap=o.array("rhokap"), voigt_a=o.array("voigt_a"), Dfreq=o.array("Dfreq"), vfx=o.array("vfx"), vfy=o.array("vfy"),
                        vfz=o.array("vfz"), rhokapD=o.array("rhokapD"),
                        r_in=np.array(rin), ic_in=np.array(icin), tau_in=np.array(tauin), r_out=np.array(rout), ic_out=np.array(icout),
                        tau_edge=np.array(tedge), Jout=J)
    print(name, "escaped", int(np.sum(np.array(icout)[:, 3] == 0)), "Jout sum", J.sum())
