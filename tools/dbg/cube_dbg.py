import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
from lart_amd import host
from oracle import oracle as O
kw = dict(temperature=1e4, taumax=3e1, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=61, nxim=16, nyim=16, DGR=1.0, cext_dust=2e-15,
          save_peeloff_2D=True, save_direc0=True, save_dust_scattered=True, alpha=15.0, beta=40.0, source_geometry="uniform_sphere", source_rmax=0.7)
for variant in ("full", "point", "nodirec0"):
    k2 = dict(kw)
    if variant == "point": k2.pop("source_geometry"); k2.pop("source_rmax")
    if variant == "nodirec0": k2.pop("save_direc0"); k2.pop("save_dust_scattered")
    sim = host.Simulation(dict(k2, iseed=1)); o = O.Oracle(**k2)
    sim.ctx.enable_photon_log(True)
    n = 64
    sim.ctx.run(n, first_id=1, stride=1, seed=2024); ng, xg = sim.ctx.photon_log(n); t = sim.ctx.tallies()
    o.run(n, first_id=1, stride=1, seed=2024); no, xo = o.photon_log()
    bad = ~(np.isclose(ng, no, rtol=1e-12, atol=0) & np.isclose(xg, xo, rtol=1e-9, atol=1e-12))
    print(variant, "mismatch", bad.sum(), "of", n, "nsd gpu/orc", t["nscatt_dust"], o.scalar("nscatt_dust"))
    for i in np.where(bad)[0][:8]:
        print("  id", i + 1, "nscatt", ng[i], no[i], "xfreq", xg[i], xo[i])
