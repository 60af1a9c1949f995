#!/usr/bin/env python3
"""Generate tests/golden/deuterium.npz from the reference's own compiled modules (oracle/_ref/libref.so) with line_id =
'ly_alpha_HD' (line type 7, H + D Lyman-alpha): the cross-species constants (line_mod.f90:1176-1191), calc_voigt_HD (:82-106),
do_resonance_HD on the reference's MT19937-64 stream (:476-549), and raytrace_to_{tau,edge}_car through calc_voigt_HD."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden")
R = O.Ref
R.lib()
rng = np.random.default_rng(20261007)
DH = 3e-3                                   # (enhanced abundance so that both species are picked often)
params = "nx=9 ny=9 nz=9 rmax=1 taumax=2000 temperature=100 geometry=rectangle DGR=1 cext_dust=1e-21 include_deuterium=1 D_to_H_ratio=%g" % DH
o = O.Oracle(params)
n = [9, 9, 9]
o.set_array("rhokap", o.grid3("rhokap") * rng.uniform(0.5, 1.5, n))
o.set_array("Dfreq", o.grid3("Dfreq") * rng.uniform(0.6, 1.8, n))
o.set_array("voigt_a", o.grid3("voigt_a") * rng.uniform(0.8, 1.2, n))
for k in ("vfx", "vfy", "vfz"):
    o.set_array(k, rng.normal(0, 2.0, n))
o.set_array("rhokapD", o.grid3("rhokap") * 0.01)
R.set_grid_from(o)
const = R.set_deuterium(1, DH)
g = o.scalar
cells = [(1, 1, 1), (5, 5, 5), (9, 3, 7)]
Df = o.grid3("Dfreq")
xs = np.concatenate([np.linspace(-10, 80, 451), rng.uniform(-30, 120, 200)])
cv = np.array([R.calc_voigt(xs, *c) for c in cells])
xres = [0.0, 2.0, const[0] / Df[0, 0, 0], const[0] / Df[4, 4, 4] + 0.5, const[0] / Df[8, 2, 6] - 1.0, 30.0, -4.0]
res = []
for i, x in enumerate(xres):
    R.seed(400 + i); res.append(R.do_resonance_HD(256, x, *cells[i % 3]))
rin, icin, tauin, rout, icout, tedge = [], [], [], [], [], []
for it in range(400):
    ic = [int(rng.integers(1, 10)) for _ in range(3)]
    pos = [g("xmin") + (ic[0] - 1 + rng.uniform()) * g("dx"), g("ymin") + (ic[1] - 1 + rng.uniform()) * g("dy"), g("zmin") + (ic[2] - 1 + rng.uniform()) * g("dz")]
    mu = rng.uniform(-1, 1); ph = rng.uniform(0, 2 * np.pi); st = np.sqrt(1 - mu * mu)
    x0 = rng.normal(0, 3.0) if it % 2 else const[0] / Df.mean() + rng.normal(0, 3.0)
    r = pos + [st * np.cos(ph), st * np.sin(ph), mu] + [x0, 1.0]; icc = ic + [1]
    tau = rng.exponential(1.0) * (10 if it % 3 else 0.3)
    rr, icr = R.raytrace_to_tau(0, r, icc, tau)
    rin.append(r); icin.append(icc); tauin.append(tau); rout.append(rr); icout.append(icr); tedge.append(R.raytrace_to_edge(0, r, icc))
R.set_deuterium(0)
np.savez_compressed(os.path.join(OUT, "deuterium.npz"), params=params, constants=const,
                    rhokap=o.array("rhokap"), voigt_a=o.array("voigt_a"), Dfreq=o.array("Dfreq"), vfx=o.array("vfx"), vfy=o.array("vfy"),
                    vfz=o.array("vfz"), rhokapD=o.array("rhokapD"), x=xs, cells=np.array(cells), calc_voigt=cv,
                    xres=np.array(xres), resonance=np.array(res),
                    r_in=np.array(rin), ic_in=np.array(icin), tau_in=np.array(tauin), r_out=np.array(rout), ic_out=np.array(icout), tau_edge=np.array(tedge))
print("constants", const, "D picks", [int((np.array(r)[:, 4] == 2).sum()) for r in res])
