#!/usr/bin/env python3
"""Generate tests/golden/multiplets.npz from the reference's own compiled modules (oracle/_ref/libref.so): for every line_id of types
4, 5 and 6 (resonance + fluorescence multiplets, line_mod.f90:713-1153) the ten line constants and the whole line%b(:) record
(f12, delE_Hz, dampings, A21, Elow_Hz, E1..E3, the alias tables), and for SiII_1527 (type 4), FeII_UV1 (type 5) and HeI_10833
(type 6) calc_voigt1/3, do_resonance4/5/6 on the reference's MT19937-64 stream and the marches on a perturbed 9^3 medium."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden")
R = O.Ref
R.lib()
IDS = ["CII_1334", "SiII_1527", "SiII_1260", "SiII_1193", "SiII_1190", "SiII_1304", "FeII_2250", "FeII_2261", "FeII_UV3", "FeII_2344", "FeII_UV1", "FeII_2600",
       "FeII_UV2", "FeII_2383", "HeI_10833"]
consts, tabs = [], []
for i in IDS:
    consts.append(R.set_line_id(i)); tabs.append(R.get_multiplet())
out = dict(ids=np.array(IDS), constants=np.array(consts), tables=np.array(tabs))
for tag, lid, T in (("si", "SiII_1527", 1e5), ("fe", "FeII_UV1", 3e4), ("he", "HeI_10833", 1e4)):
    rng = np.random.default_rng(20261008 + ord(tag[0]))
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
    R.set_line_id(lid); t = R.get_multiplet()
    g = o.scalar
    offs = [0.0] + [-t[8 + i] / g("Dfreq_ref") for i in range(1, int(t[1]))]       # line centres: x = -delE_Hz(i)/Dfreq
    xs = np.concatenate([c + np.linspace(-8, 8, 161) for c in offs] + [rng.uniform(min(offs) - 20, 20, 200)])
    cells = [(1, 1, 1), (5, 5, 5), (9, 3, 7)]
    cv = np.array([R.calc_voigt(xs, *cc) for cc in cells])
    xres = [0.0, -1.5, 3.0] + [c + d for c in offs[1:] for d in (0.0, 1.0)] + [0.5 * min(offs) - 0.3, 12.0]
    res = []
    for i, x in enumerate(xres):
        R.seed(700 + i); res.append(R.do_resonance_ptr(256, x, *cells[i % 3]))
    rin, icin, tauin, rout, icout, tedge = [], [], [], [], [], []
    for it in range(300):
        ic = [int(rng.integers(1, 10)) for _ in range(3)]
        pos = [g("xmin") + (ic[0] - 1 + rng.uniform()) * g("dx"), g("ymin") + (ic[1] - 1 + rng.uniform()) * g("dy"), g("zmin") + (ic[2] - 1 + rng.uniform()) * g("dz")]
        mu = rng.uniform(-1, 1); ph = rng.uniform(0, 2 * np.pi); st = np.sqrt(1 - mu * mu)
        x0 = rng.normal(offs[it % len(offs)], 3.0)
        r = pos + [st * np.cos(ph), st * np.sin(ph), mu] + [x0, 1.0]; icc = ic + [1]
        tau = rng.exponential(1.0) * (10 if it % 3 else 0.3)
        rr, icr = R.raytrace_to_tau(0, r, icc, tau)
        rin.append(r); icin.append(icc); tauin.append(tau); rout.append(rr); icout.append(icr); tedge.append(R.raytrace_to_edge(0, r, icc))
    out.update({tag + "_params": params, tag + "_x": xs, tag + "_cells": np.array(cells), tag + "_calc_voigt": cv, tag + "_xres": np.array(xres),
                tag + "_resonance": np.array(res), tag + "_r_in": np.array(rin), tag + "_ic_in": np.array(icin), tag + "_tau_in": np.array(tauin),
                tag + "_r_out": np.array(rout), tag + "_ic_out": np.array(icout), tag + "_tau_edge": np.array(tedge)})
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD"):
        out[tag + "_" + k] = o.array(k)
    if tag == "he":                                                               # par%HeI_coherent: E1..E3 of the atom-frame frequency
        R.set_HeI_coherent(1); coh = []
        for i, x in enumerate(xres):
            R.seed(800 + i); coh.append(R.do_resonance_ptr(256, x, *cells[i % 3]))
        R.set_HeI_coherent(0)
        out["he_coherent_resonance"] = np.array(coh)
    rr = np.array(res)
    print(lid, "offsets", offs, "E1 values", np.unique(np.round(rr[..., 9], 6)))
R.set_line_id("ly_alpha")
np.savez_compressed(os.path.join(OUT, "multiplets.npz"), **out)
