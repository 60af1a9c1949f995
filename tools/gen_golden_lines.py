#!/usr/bin/env python3
"""Generate tests/golden/line_ids.npz from the reference's own compiled modules (oracle/_ref/libref.so): the constants
setup_resonance_line (line_mod.f90:590-713) gives for every single-line (type 1) and S1/2-P doublet (type 2) line_id, and for two
of the doublets (MgII_2796, CIV_1548) calc_voigt2, do_resonance2 on the reference's MT19937-64 stream and the marches through the
two-component profile on a perturbed 9^3 medium."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden")
R = O.Ref
R.lib()
IDS = ["CIV_1548", "NV_1239", "OVI_1032", "NaI_D", "CaII_HK", "MgII_2796", "SiIV_1394", "AlII_1671"]
consts = np.array([R.set_line_id(i) for i in IDS])
out = dict(ids=np.array(IDS), constants=consts)
for tag, lid, T in (("mg", "MgII_2796", 3e4), ("c4", "CIV_1548", 1e5)):
    rng = np.random.default_rng(20261007 + len(tag) + ord(tag[0]))
    params = "nx=9 ny=9 nz=9 rmax=1 taumax=300 temperature=%g geometry=rectangle DGR=1 cext_dust=1e-21 line_id=%s" % (T, lid)
    o = O.Oracle(params)
    n = [9, 9, 9]
    o.set_array("rhokap", o.grid3("rhokap") * rng.uniform(0.5, 1.5, n))
    o.set_array("Dfreq", o.grid3("Dfreq") * rng.uniform(0.6, 1.8, n))
    o.set_array("voigt_a", o.grid3("voigt_a") * rng.uniform(0.8, 1.2, n))
    for k in ("vfx", "vfy", "vfz"):
        o.set_array(k, rng.normal(0, 2.0, n))
    o.set_array("rhokapD", o.grid3("rhokap") * 0.01)
    R.set_grid_from(o)
    c = R.set_line_id(lid)
    g = o.scalar
    dx = c[6] / g("Dfreq_ref")                                    # the H component sits at x = -DnuHK/Dfreq
    xs = np.concatenate([np.linspace(-8, 8, 161), -dx + np.linspace(-8, 8, 161), rng.uniform(-1.5 * dx, 0.5 * dx, 200)])
    cells = [(1, 1, 1), (5, 5, 5), (9, 3, 7)]
    cv = np.array([R.calc_voigt(xs, *cc) for cc in cells])
    xres = [0.0, -1.5, 3.0, -dx, -dx + 2.0, -0.5 * dx, -dx - 4.0]
    res = []
    for i, x in enumerate(xres):
        R.seed(500 + i); res.append(R.do_resonance2(256, x, *cells[i % 3]))
    rin, icin, tauin, rout, icout, tedge = [], [], [], [], [], []
    for it in range(300):
        ic = [int(rng.integers(1, 10)) for _ in range(3)]
        pos = [g("xmin") + (ic[0] - 1 + rng.uniform()) * g("dx"), g("ymin") + (ic[1] - 1 + rng.uniform()) * g("dy"), g("zmin") + (ic[2] - 1 + rng.uniform()) * g("dz")]
        mu = rng.uniform(-1, 1); ph = rng.uniform(0, 2 * np.pi); st = np.sqrt(1 - mu * mu)
        x0 = rng.normal(0, 3.0) if it % 2 else rng.normal(-dx, 3.0)
        r = pos + [st * np.cos(ph), st * np.sin(ph), mu] + [x0, 1.0]; icc = ic + [1]
        tau = rng.exponential(1.0) * (10 if it % 3 else 0.3)
        rr, icr = R.raytrace_to_tau(0, r, icc, tau)
        rin.append(r); icin.append(icc); tauin.append(tau); rout.append(rr); icout.append(icr); tedge.append(R.raytrace_to_edge(0, r, icc))
    out.update({tag + "_params": params, tag + "_x": xs, tag + "_cells": np.array(cells), tag + "_calc_voigt": cv, tag + "_xres": np.array(xres),
                tag + "_resonance": np.array(res), tag + "_r_in": np.array(rin), tag + "_ic_in": np.array(icin), tag + "_tau_in": np.array(tauin),
                tag + "_r_out": np.array(rout), tag + "_ic_out": np.array(icout), tag + "_tau_edge": np.array(tedge)})
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD"):
        out[tag + "_" + k] = o.array(k)
    print(lid, "DnuHK/Dfreq", dx, "E1 range", np.array(res)[..., 9].min(), np.array(res)[..., 9].max())
R.set_line_id("ly_alpha")
np.savez_compressed(os.path.join(OUT, "line_ids.npz"), **out)
for i, c in zip(IDS, consts):
    print(i, c)
