#!/usr/bin/env python3
"""sight-line map kernel on the C3 shapes (examples/sphere_peel: 201^3 cells, 129x129 image, 201 frequency bins) and on an
octree / clump medium: device time (HIP events), Voigt evaluations per second, and the CPU oracle on a sample of pixels."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lart_amd import host

cases = {
    "car_201^3_129x129x201": dict(temperature=1e1, taumax=1e6, nx=201, ny=201, nz=201, rmax=1.0, nxfreq=201, nxim=129, nyim=129, save_peeloff=True),
    "car_201^3_512x512x201": dict(temperature=1e1, taumax=1e6, nx=201, ny=201, nz=201, rmax=1.0, nxfreq=201, nxim=512, nyim=512, save_peeloff=True),
    "clumps_1.3M_256x256x301": dict(use_clump_medium=True, rmax=1.0, clump_radius=0.003, clump_f_cov=16.0, clump_tau0=1e3, temperature=1e4, nxfreq=301,
                                    velocity_min=-100.0, velocity_max=100.0, nxim=256, nyim=256, save_peeloff=True, iseed=7),
}
for name, kw in cases.items():
    t0 = time.perf_counter(); sim = host.Simulation(dict(kw)); t_setup = time.perf_counter() - t0
    sim.make_sightline_tau()                                       # warm-up (module load)
    t0 = time.perf_counter(); tg, ng, td = sim.make_sightline_tau(); wall = time.perf_counter() - t0
    ms = sim.ctx.sightline_time()
    nray = int((ng > 0).sum()); nxf = kw["nxfreq"]
    if "nx" in kw and not kw.get("use_clump_medium"):
        # cells crossed per ray ~ chord length / dx summed over axes; count exactly from N_gas is not possible: estimate 1.5 * nx * chord fraction
        steps = None
    out = dict(case=name, setup_s=round(t_setup, 2), kernel_ms=round(ms, 3), wall_ms_incl_copies=round(wall * 1e3, 1), rays_hitting=nray,
               ray_freq_per_s=nray * nxf / (ms * 1e-3), bytes_out=tg.nbytes)
    print(json.dumps(out), flush=True)
if "--cpu" in sys.argv:
    from oracle import oracle as O
    kw = dict(cases["car_201^3_129x129x201"]); kw.update(nxim=16, nyim=16)
    kw["dxim"] = None
    kw = {k: v for k, v in kw.items() if v is not None}
    o = O.Oracle(**kw)
    t0 = time.perf_counter(); tg, ng, td = o.sightline_tau(); dt = time.perf_counter() - t0
    print(json.dumps(dict(case="cpu_oracle_car_201^3_16x16x201", seconds=dt, ray_freq_per_s=int((ng > 0).sum()) * 201 / dt, threads=os.cpu_count())))
