#!/usr/bin/env python3
"""Generate tests/golden/fine_structure.npz from the reference's own compiled modules (oracle/_ref/libref.so) with
par%fine_structure = .true. (line type 2, the Ly-alpha doublet): DnuHK_Hz (line_mod.f90:1250-1253), calc_voigt2 (:49-62),
do_resonance2 on the reference's MT19937-64 stream (:141-192), and raytrace_to_{tau,edge}_car through calc_voigt2 on a
perturbed 9^3 medium at 10-100 K (where the level splitting is a few Doppler widths)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden")
R = O.Ref
R.lib()
rng = np.random.default_rng(20261006)
params = "nx=9 ny=9 nz=9 rmax=1 taumax=200 temperature=30 geometry=rectangle DGR=1 cext_dust=1e-21 fine_structure=1"
o = O.Oracle(params)
n = [9, 9, 9]
o.set_array("rhokap", o.grid3("rhokap") * rng.uniform(0.5, 1.5, n))
o.set_array("Dfreq", o.grid3("Dfreq") * rng.uniform(0.6, 1.8, n))
o.set_array("voigt_a", o.grid3("voigt_a") * rng.uniform(0.8, 1.2, n))
for k in ("vfx", "vfy", "vfz"):
    o.set_array(k, rng.normal(0, 2.0, n))
o.set_array("rhokapD", o.grid3("rhokap") * 0.01)
R.set_grid_from(o)
Dnu = R.set_fine_structure(1)
g = o.scalar
xs = np.concatenate([np.linspace(-12, 8, 401), rng.uniform(-30, 30, 200)])
cells = [(1, 1, 1), (5, 5, 5), (9, 3, 7)]
cv = np.array([R.calc_voigt(xs, *c) for c in cells])
xres = [0.0, -1.5, -3.0, -3.5, 2.0, -6.0, 8.0]
res = []
for i, x in enumerate(xres):
    R.seed(300 + i); res.append(R.do_resonance2(256, x, *cells[i % 3]))
rin, icin, tauin, rout, icout, tedge = [], [], [], [], [], []
for it in range(400):
    ic = [int(rng.integers(1, 10)) for _ in range(3)]
    pos = [g("xmin") + (ic[0] - 1 + rng.uniform()) * g("dx"), g("ymin") + (ic[1] - 1 + rng.uniform()) * g("dy"), g("zmin") + (ic[2] - 1 + rng.uniform()) * g("dz")]
    mu = rng.uniform(-1, 1); ph = rng.uniform(0, 2 * np.pi); st = np.sqrt(1 - mu * mu)
    r = pos + [st * np.cos(ph), st * np.sin(ph), mu] + [rng.normal(-1.5, 3.0), 1.0]; icc = ic + [1]
    tau = rng.exponential(1.0) * (10 if it % 3 else 0.3)
    rr, icr = R.raytrace_to_tau(0, r, icc, tau)
    rin.append(r); icin.append(icc); tauin.append(tau); rout.append(rr); icout.append(icr); tedge.append(R.raytrace_to_edge(0, r, icc))
R.set_fine_structure(0)
np.savez_compressed(os.path.join(OUT, "fine_structure.npz"), params=params, DnuHK_Hz=Dnu,
                    rhokap=o.array("rhokap"), voigt_a=o.array("voigt_a"), Dfreq=o.array("Dfreq"), vfx=o.array("vfx"), vfy=o.array("vfy"),
                    vfz=o.array("vfz"), rhokapD=o.array("rhokapD"), x=xs, cells=np.array(cells), calc_voigt=cv,
                    xres=np.array(xres), resonance=np.array(res),
                    r_in=np.array(rin), ic_in=np.array(icin), tau_in=np.array(tauin), r_out=np.array(rout), ic_out=np.array(icout), tau_edge=np.array(tedge))
print("DnuHK_Hz", Dnu, "E1 range", np.array(res)[..., 9].min(), np.array(res)[..., 9].max())
