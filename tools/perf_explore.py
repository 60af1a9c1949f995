#!/usr/bin/env python3
"""throughput of the transport kernel vs photon age / pool geometry on the headline workload (GPU box)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()            # before liblart_gpu touches HIP (torch's lazy init fails afterwards)
import bench
from lart_amd import host

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--events", type=int, default=8192)
ap.add_argument("--lanes", type=int, default=0)
ap.add_argument("--block", type=int, default=0)
ap.add_argument("--tau", type=float, default=1e6)
ap.add_argument("--photons", type=float, default=1e7)
ap.add_argument("--every", type=int, default=1)
ap.add_argument("--config", default="sphere")
a = ap.parse_args()
w = dict(bench.WORKLOAD); w["taumax"] = a.tau; w["no_photons"] = a.photons
if a.config == "peel":      # examples/sphere_peel/t1tau6.in (C3): 201^3, Stokes, 129x129x201 peel cube
    w.update(xyz_symmetry=False, comoving_source=True, nx=201, ny=201, nz=201, use_stokes=True, save_Jin=True, nxim=129, nyim=129)
sim = host.Simulation(w)
if a.lanes or a.block:
    sim.ctx.set_pool(a.lanes, a.block)
sim.ctx.begin(int(a.photons), seed=w["iseed"])
prev = sim.ctx.tallies(); tprev = time.perf_counter(); t00 = tprev
for s in range(a.steps):
    sim.ctx.step(1, a.events)
    done, active = sim.ctx.sync()
    if (s + 1) % a.every == 0 or active == 0:
        t = sim.ctx.tallies(); now = time.perf_counter()
        ev = t["n_events"] - prev["n_events"]; cs = t["n_cellsteps"] - prev["n_cellsteps"]
        cp = t["n_cellsteps_peel"] - prev["n_cellsteps_peel"]
        print(json.dumps(dict(step=s + 1, t=round(now - t00, 3), events_per_s=ev / (now - tprev), cells_per_event=cs / max(ev, 1), peel_cells_per_event=cp / max(ev, 1), done=done, active=active,
                              mean_nscatt_done=(t["nscatt_gas"] / max(t["nphotons_done"], 1)), lanes=sim.ctx.pool_info()[0])), flush=True)
        prev, tprev = t, now
    if active == 0:
        break
if os.environ.get("LART_GPU_LIB", "").endswith("_prof.so"):            # section clocks of the -DLART_PROFILE tuning build
    ptr, nt = sim.ctx.tally_buffer()
    buf = host._as_cuda_tensor(ptr, 32, sim.ctx).cpu().numpy()
    sc, tr = buf[8:20], buf[20:24]
    names = ["pool", "birth", "tau+first", "raytrace_to_tau", "scatter_pre", "vz_core_loop", "scatter_post", "finish", "vz_wing_setup", "vz_wing_loop", "vz_wing_tan", "-"]
    print(json.dumps({"section_clock_share": {k: round(float(v / sc.sum()), 4) for k, v in zip(names, sc)},
                      "wave_trips_per_round": {"core_tries": tr[0] / tr[2], "wing_tries": tr[1] / tr[2], "wing_entered": tr[3] / tr[2]},
                      "clocks_per_round": float(sc.sum() / tr[2])}))
ms, n = sim.ctx.kernel_time()
print(json.dumps(dict(total_kernel_ms=ms, launches=n, total_events=prev["n_events"], overall_events_per_s=prev["n_events"] / (time.perf_counter() - t00))))
