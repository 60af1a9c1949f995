#!/usr/bin/env python3
"""cost of the profile tallies on the octree config (C5): events/s with each of CALCJ / CALCP / CALCPnew alone"""
import json, os, sys, tempfile, time
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lart_amd import host, amr
tmp = tempfile.mkdtemp(); afile = os.path.join(tmp, "sphere.dat"); amr.write_generic_amr(afile, amr.make_amr_sphere())
base = dict(use_amr_grid=True, amr_file=afile, temperature=1e4, taumax=1e4, rmax=1.0, nxfreq=241, xfreq_min=-15.0, xfreq_max=15.0, no_photons=1e7, iseed=1)
for flags in (dict(), dict(calc_P=True), dict(calc_Pnew=True), dict(calc_J=True), dict(calc_J=True, calc_P=True, calc_Pnew=True), dict(calc_P=True, geometry_JPa=3)):
    sim = host.Simulation(dict(base, **flags))
    sim.ctx.begin(int(1e7), seed=1)
    sim.ctx.step(1, 256); sim.ctx.sync()
    t0 = sim.ctx.tallies(); w0 = time.perf_counter()
    sim.ctx.step(2, 1024); sim.ctx.sync()
    dt = time.perf_counter() - w0; t1 = sim.ctx.tallies()
    print(json.dumps({"flags": flags, "events_per_s": (t1["n_events"] - t0["n_events"]) / dt, "nbins": sim.jpa.nbins}), flush=True)
    sim.ctx.close()
