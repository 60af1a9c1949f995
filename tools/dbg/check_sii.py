#!/usr/bin/env python3
"""which convention do the 2022 / 2023 logs of examples/SiII_1193 and SiII_1527 count <N_scatt> in?  (GPU box)  prints weighted (v2.00) and unweighted means"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
from lart_amd import host
SETS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_example_params.json")))
LOG = {"SiII_1193/tau1e+0_V000.in": 1.0155, "SiII_1193/tau1e+0_V050.in": 1.0126, "SiII_1193/tau1e+0_V100.in": 1.0116, "SiII_1193/tau1e+0_V200.in": 1.0103,
       "SiII_1193/tau1e+3_V000.in": None, "SiII_1527/t1e5tau1e0.in": 1.0570, "SiII_1527/t1e5tau1e0_V050.in": 1.0147, "SiII_1527/t1e5tau1e0_V100.in": 1.0075, "SiII_1527/t1e5tau2e0.in": 1.0943,
       "SiII_1527/t1e5tau1e1.in": 1.1597}
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
for name, want in LOG.items():
    kw = dict(SETS[name]); kw["no_photons"] = n; kw["iseed"] = 7
    for k in ("nprint", "num_send_at_once", "out_merge"):
        kw.pop(k, None)
    t0 = time.time(); sim = host.Simulation(kw); sim.run_simulation(n); t = sim.ctx.tallies()
    print(name, "logged", want, "weighted %.5f" % ((t["nscatt_gas"] + t["nscatt_dust"]) / n), "unweighted %.5f" % (t["n_events"] / n), "%.1fs" % (time.time() - t0), flush=True)
    sim.ctx.close()
