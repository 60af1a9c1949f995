#!/usr/bin/env python3
"""print the device and oracle history records around the first difference of the mismatching photons of one audit case
(tests/test_gpu_divergence.py): python tools/debug_divergence.py CASE [nshow]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_divergence as T
from oracle import oracle as O
from lart_amd import host
np.set_printoptions(linewidth=250, precision=12)
case = sys.argv[1]; nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cases = {"clumps": (dict(T.CLUMP), 100000, 11), "clumps_moving": (dict(T.CLUMP, clump_sigma_v=30.0, velocity_type="hubble", Vexp=80.0), 100000, 11),
         "overlap_dust": (dict(T.OVERLAP, clump_sigma_v=20.0, DGR=1.0, cext_dust=3e-18), 50000, 11),
         "overlap_sigv": (dict(T.OVERLAP, clump_sigma_v=100.0), 100000, 11),
         "plain": (dict(T.CAR[0][1], iseed=5), 100000, 5)}
kw, n, seed = cases[case]
sim = host.Simulation(dict(kw)); o = O.Oracle(**({k: v for k, v in kw.items() if k != "iseed"} if case == "plain" else kw))
sim.ctx.enable_photon_log(True); sim.run_simulation(n, seed=seed); ng, xg = sim.ctx.photon_log(n)
o.run(n, seed=seed); no, xo = o.photon_log()
same = np.isclose(ng, no, rtol=1e-12, atol=0) & (np.isclose(xg, xo, rtol=1e-7, atol=1e-9) | ((xo == 0) & (xg == 0)))
bad = np.nonzero(~same)[0]
print(case, "mismatching", len(bad), "of", n)
shown = 0
for k in bad:
    cap = int(min(ng[k], no[k]) * 1.05 + 64)
    tr, nrec = sim.ctx.test_trace(n, [k], cap, seed)
    otr, on = o.trace(seed, int(k) + 1, cap)
    L = float(kw.get('clump_radius', 1.0)) if kw.get('use_clump_medium') else 1.0
    i, drift, dmax = T.first_difference(tr[0], int(nrec[0]), otr, on, L)
    mg = otr[i, 8] if (i is not None and i < len(otr)) else -1.0
    if i is None or mg < max(1e-9, 1e3 * drift):
        continue
    print('drift before', drift, 'max drift', dmax)
    print("photon", k, "log dev (%.15g, %.12g) orc (%.15g, %.12g)" % (ng[k], xg[k], no[k], xo[k]), "records dev", nrec[0], "orc", on, "first difference", i, "margin", mg)
    m = min(int(nrec[0]), on, cap)
    dpos = np.abs(tr[0][:m, 2:5] - otr[:m, 2:5]).max(axis=1)
    j0 = np.nonzero(dpos > 1e-12)[0]
    if len(j0):
        j = int(j0[0]); print("  first |dpos| > 1e-12 at record", j, "dpos", dpos[max(0, j - 1):j + 3])
        for q in range(max(0, j - 1), min(j + 2, m)):
            print("  dev", q, tr[0][q]); print("  orc", q, otr[q])
    if i is not None:
        for j in range(max(0, i - 1), min(i + 2, int(nrec[0]), on)):
            print("  dev", j, tr[0][j]); print("  orc", j, otr[j])
    shown += 1
    if shown >= nshow:
        break
