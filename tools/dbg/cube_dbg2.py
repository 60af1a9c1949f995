import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
torch.cuda.init()
from lart_amd import host
from oracle import oracle as O
from conftest_data import mueller_path
import tempfile
mf = mueller_path(tempfile.mkdtemp())
CLUMP = dict(use_clump_medium=True, rmax=1.0, clump_radius=0.01, clump_f_cov=5.0, clump_tau0=1e3, temperature=1e4, nxfreq=301,
             velocity_min=-100.0, velocity_max=100.0, nx=11, ny=11, nz=11)
kw = dict(CLUMP, clump_tau0=1e2, use_stokes=True, DGR=1.0, scatt_mat_file=mf, cext_dust=1.6e-21, nxim=15, nyim=15, save_peeloff_2D=True, alpha=20.0, beta=50.0, iseed=1)
sim = host.Simulation(dict(kw)); o = O.Oracle(**kw)
sim.ctx.enable_photon_log(True)
nxf = sim.cpar.nxfreq; per = 16; prev = None
for b in range(60):
    first = 1 + b * per
    sim.ctx.run(per, first_id=first, stride=1, seed=2024); t = sim.ctx.tallies(); ng, xg = sim.ctx.photon_log(per)
    o.zero(); o.run(per, first_id=first, stride=1, seed=2024); no, xo = o.photon_log()
    clean = bool(np.all(np.isclose(ng, no, rtol=1e-12, atol=0) & (np.isclose(xg, xo, rtol=1e-9, atol=1e-12) | (xo == 0.0))))
    cur = {k: t[k].copy() for k in ("scatt", "I", "Q", "direc")}
    dev = {k: (cur[k] - prev[k] if prev is not None else cur[k]) for k in cur}; prev = cur
    if not clean: continue
    for k in dev:
        ora = o.array(k + ":0").reshape((nxf, 15, 15), order="F"); d = dev[k][..., 0]
        scale = np.abs(o.array("I:0")).max()
        bad = np.abs(d - ora) > 1e-9 * scale
        if bad.any():
            idx = np.argwhere(bad)
            print("batch", b, k, "nbad", bad.sum(), "scale", scale)
            for i in idx[:6]:
                print("   ", tuple(i), d[tuple(i)], ora[tuple(i)], (d[tuple(i)] - ora[tuple(i)]) / scale)
print("done")
