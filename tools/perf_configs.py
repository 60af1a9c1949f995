#!/usr/bin/env python3
"""events/s of the transport kernels on BASELINE.json's other configurations (C1, C3, C4, C5 of SURVEY section 8), GPU box.
Not the bench line (bench.py measures configs[1]); recorded in DESIGN.md for orientation."""
import json, os, sys, tempfile, time
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lart_amd import host, amr
from conftest_data import mueller_path

tmp = tempfile.mkdtemp()
afile = os.path.join(tmp, "sphere.dat"); amr.write_generic_amr(afile, amr.make_amr_sphere())
mfile = mueller_path(tmp)
CONFIGS = {
    "C1 slab tau0=1e4 T=10K xy_periodic Stokes": dict(temperature=10.0, taumax=1e4, xy_periodic=True, nx=1, ny=1, nz=201, use_stokes=True, nxfreq=121, no_photons=1e7, iseed=1),
    "C3 sphere tau0=1e6 201^3 Stokes peel 129x129x201": dict(temperature=10.0, taumax=1e6, comoving_source=True, nx=201, ny=201, nz=201, rmax=1.0, nxfreq=201,
                                                            use_stokes=True, save_Jin=True, nxim=129, nyim=129, no_photons=1e7, iseed=1),
    "C4 clumps f_cov=5 tau_cl=1e3 dust Mueller Stokes core_skip": dict(use_clump_medium=True, rmax=1.0, clump_radius=0.01, clump_f_cov=5.0, clump_tau0=1e3, temperature=1e4,
                                                                    use_stokes=True, DGR=1.0, scatt_mat_file=mfile, core_skip=True, nxfreq=301, velocity_min=-100.0, velocity_max=100.0,
                                                                    nx=11, ny=11, nz=11, no_photons=1e7, iseed=1),
    "C5 octree sphere levels 3-7 (178480 leaves) tau0=1e4 CALCJ+CALCP+CALCPnew": dict(use_amr_grid=True, amr_file=afile, temperature=1e4, taumax=1e4, rmax=1.0, nxfreq=241,
                                                                                   xfreq_min=-15.0, xfreq_max=15.0, calc_J=True, calc_P=True, calc_Pnew=True, no_photons=1e7, iseed=1),
    # line types beyond plain Ly-alpha on the headline geometry (tau0 = 1e6 octant): the LT = 2 / 7 / 4 kernel families
    "L2 headline sphere with par%fine_structure (LT=2)": dict(temperature=10.0, taumax=1e6, xyz_symmetry=True, nx=101, ny=101, nz=101, rmax=1.0, nxfreq=201, fine_structure=True,
                                                              no_photons=1e7, iseed=1),
    "L7 headline sphere T=1e4 with deuterium D/H=1.5e-5 (LT=7)": dict(temperature=1e4, taumax=1e6, xyz_symmetry=True, nx=101, ny=101, nz=101, rmax=1.0, nxfreq=201,
                                                                     include_deuterium=True, no_photons=1e7, iseed=1),
    "L4 examples/FeII_test-like: FeII_UV1 tau0=1e2 T=1e4 sphere 101^3 (LT=4, fluorescence)": dict(temperature=1e4, taumax=1e2, nx=101, ny=101, nz=101, rmax=1.0, nxfreq=600,
                                                                                              line_id="FeII_UV1", wavelength_min=2575.0, wavelength_max=2640.0, no_photons=4e9, iseed=1),   # (few events per photon: enough photons to keep every lane busy)
    "L6 HeI_10833 tau0=1e3 T=1e4 sphere 101^3 (LT=4, three blended levels)": dict(temperature=1e4, taumax=1e3, nx=101, ny=101, nz=101, rmax=1.0, nxfreq=201, line_id="HeI_10833",
                                                                                velocity_min=-300.0, velocity_max=300.0, no_photons=1e7, iseed=1),
}
for name, kw in CONFIGS.items():
    sim = host.Simulation(dict(kw))
    sim.ctx.begin(int(kw["no_photons"]), seed=1)
    sim.ctx.step(1, 256); sim.ctx.sync()
    t0 = sim.ctx.tallies(); w0 = time.perf_counter()
    sim.ctx.step(2, 1024); sim.ctx.sync()
    dt = time.perf_counter() - w0; t1 = sim.ctx.tallies()
    ev = t1["n_events"] - t0["n_events"]
    print(json.dumps({"config": name, "events_per_s": ev / dt, "cells_per_event": (t1["n_cellsteps"] - t0["n_cellsteps"]) / max(ev, 1),
                      "peel_cells_per_event": (t1["n_cellsteps_peel"] - t0["n_cellsteps_peel"]) / max(ev, 1), "lanes": sim.ctx.pool_info()[0]}), flush=True)
    sim.ctx.close()
