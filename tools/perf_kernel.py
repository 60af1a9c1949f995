#!/usr/bin/env python3
"""Steady-state launches of ONE transport-kernel variant for rocprofv3 (tools/gpu_job.sh profcfg:TAG:CONFIG) and for the
roofline of the configurations bench.py does not time (BASELINE.json configs[0], [2], [3], [4] = C1, C3, C4, C5).

  python tools/perf_kernel.py --config c3 [--warm 3 --steps 4 --events 4096]

prints one JSON line: events/s, cells per event (flight and peel rays), tally atomics per event as the kernels count them, the
algorithmic bytes of SURVEY section 8(d) (48 B per visited cell of flight and peel rays + 16 B per tally atomic) per launch, the
average launch duration from HIP events on the launch stream, and the HBM roofline fraction."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def configs(tmp):
    from lart_amd import amr
    from conftest_data import mueller_path
    afile = os.path.join(tmp, "sphere.dat")
    c = {
        "c1": dict(temperature=10.0, taumax=1e4, xy_periodic=True, nx=1, ny=1, nz=201, use_stokes=True, save_Jin=True, hgg=0.6592, albedo=0.3899, no_photons=1e8, iseed=1),
        "c2": dict(temperature=10.0, taumax=1e6, xyz_symmetry=True, nx=101, ny=101, nz=101, rmax=1.0, nxfreq=201, no_photons=1e7, iseed=1),
        "c3": dict(temperature=10.0, taumax=1e6, nx=201, ny=201, nz=201, rmax=1.0, nxfreq=201, use_stokes=True, save_Jin=True, nxim=129, nyim=129, no_photons=1e8, iseed=1),
        "c3n": dict(temperature=10.0, taumax=1e6, nx=201, ny=201, nz=201, rmax=1.0, nxfreq=201, use_stokes=True, save_Jin=True, no_photons=1e8, iseed=1),   # C3 without its observer
        "c4": dict(use_clump_medium=True, rmax=1.0, clump_radius=0.01, clump_f_cov=5.0, clump_tau0=1e3, temperature=1e4, use_stokes=True, DGR=1.0,
                   scatt_mat_file="MUELLER", core_skip=True, nxfreq=301, velocity_min=-100.0, velocity_max=100.0, nx=11, ny=11, nz=11, no_photons=1e8, iseed=1),
        "c4o": dict(use_clump_medium=True, rmax=1.0, clump_radius=0.3, clump_f_cov=2.0, clump_tau0=1e3, temperature=1e4, clump_sigma_v=100.0, clump_allow_overlap=True,
                    nxfreq=101, velocity_min=-200.0, velocity_max=200.0, nx=11, ny=11, nz=11, no_photons=1e8, iseed=1),        # examples/clump_sphere/clump_overlap_sigv.in
        "c4os": dict(use_clump_medium=True, rmax=1.0, clump_radius=0.3, clump_f_cov=2.0, clump_tau0=1e3, temperature=1e4, clump_allow_overlap=True,
                     nxfreq=101, velocity_min=-100.0, velocity_max=100.0, nx=11, ny=11, nz=11, no_photons=1e8, iseed=1),       # examples/clump_sphere/clump_overlap_mono.in with a Voigt source: static clumps
        "c5": dict(use_amr_grid=True, amr_file=afile, temperature=1e4, taumax=1e4, rmax=1.0, nxfreq=241, xfreq_min=-15.0, xfreq_max=15.0,
                   calc_J=True, calc_P=True, calc_Pnew=True, no_photons=1e8, iseed=1),
        "c5obs": dict(use_amr_grid=True, amr_file=afile, temperature=1e4, taumax=1e4, rmax=1.0, nxfreq=121, nxim=100, nyim=100, calc_J=True, calc_P=True, calc_Pnew=True,
                      no_photons=1e8, iseed=1),                # the C5 leg of bench.py (other_configs: c5_octree_JP, with its 100 x 100 observer)
        "c5j": dict(use_amr_grid=True, amr_file=afile, temperature=1e4, taumax=1e4, rmax=1.0, nxfreq=241, xfreq_min=-15.0, xfreq_max=15.0, calc_J=True, no_photons=1e8, iseed=1),
        "c5p": dict(use_amr_grid=True, amr_file=afile, temperature=1e4, taumax=1e4, rmax=1.0, nxfreq=241, xfreq_min=-15.0, xfreq_max=15.0, calc_P=True, no_photons=1e8, iseed=1),
        "c5q": dict(use_amr_grid=True, amr_file=afile, temperature=1e4, taumax=1e4, rmax=1.0, nxfreq=241, xfreq_min=-15.0, xfreq_max=15.0, calc_Pnew=True, no_photons=1e8, iseed=1),
        "c5n": dict(use_amr_grid=True, amr_file=afile, temperature=1e4, taumax=1e4, rmax=1.0, nxfreq=241, xfreq_min=-15.0, xfreq_max=15.0, no_photons=1e8, iseed=1),
        # line type 8 (Ly-beta -> H-alpha) and the H2 lines on the C2-like medium at a depth where packets still convert / meet H2 (plain grid: setup.f90:239-280)
        "lyb": dict(line_id="ly_beta", temperature=1e4, taumax=1e4, nx=101, ny=101, nz=101, rmax=1.0, nxfreq=201, DGR=1.0, cext_dust=1e-21, no_photons=1e8, iseed=1),
        "lya0": dict(temperature=1e4, taumax=1e4, nx=101, ny=101, nz=101, rmax=1.0, nxfreq=201, DGR=1.0, cext_dust=1e-21, no_photons=1e8, iseed=1),      # the same medium in Ly-alpha
        "h2": dict(temperature=1e4, taumax=1e4, nx=101, ny=101, nz=101, rmax=1.0, nxfreq=201, DGR=1.0, cext_dust=1e-21, h2_model="neufeld", f_H2=0.05, h2_temperature=8000.0,
                   h2_data_dir="H2DATA", no_photons=1e8, iseed=1),
    }
    return c, afile, mueller_path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--events", type=int, default=4096)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    a = ap.parse_args()
    import torch
    torch.cuda.init()
    from lart_amd import host, amr
    tmp = tempfile.mkdtemp()
    cfgs, afile, mueller_path = configs(tmp)
    kw = dict(cfgs[a.config])
    if kw.get("use_amr_grid"):
        amr.write_generic_amr(afile, amr.make_amr_sphere())
    if kw.get("h2_data_dir") == "H2DATA":                      # the level table h2_init reads, written back from the fixture (tests/conftest.py::h2_data_dir)
        import numpy as np
        z = np.load(os.path.join(ROOT, "tests", "golden", "h2_energy_X.npz")); kw["h2_data_dir"] = tmp
        with open(os.path.join(tmp, "energy_X.dat"), "w") as f:
            f.write("2 4 29 // magic number\n#H2 electronic ground state rovibrational level energies [cm^-1]\n")
            for v, j, e in zip(z["V"], z["J"], z["E_cm"]):
                f.write("%d\t%d\t%.17g\n" % (int(v), int(j), float(e)))
    if kw.get("scatt_mat_file") == "MUELLER":
        kw["scatt_mat_file"] = mueller_path(tmp)
    sim = host.Simulation(kw)
    if a.lanes or a.block:
        sim.ctx.set_pool(a.lanes, a.block)
    sim.ctx.begin(int(kw["no_photons"]), seed=1)
    for _ in range(a.warm):
        sim.ctx.step(1, a.events)
    sim.ctx.sync()
    t0 = sim.ctx.tallies(); ms0, nl0 = sim.ctx.kernel_time(); w0 = time.perf_counter()
    sim.ctx.step(a.steps, a.events); sim.ctx.sync()
    dt = time.perf_counter() - w0; t1 = sim.ctx.tallies(); ms1, nl1 = sim.ctx.kernel_time()
    d = lambda k: t1[k] - t0[k]
    ev, nl = d("n_events"), max(nl1 - nl0, 1)
    k_ms = (ms1 - ms0) / nl
    atom = d("n_atomics") if "n_atomics" in t1 else 0.0
    bytes_launch = (48.0 * (d("n_cellsteps") + d("n_cellsteps_peel")) + 16.0 * atom) / nl
    out = {"config": a.config, "events_per_s": ev / dt, "events_per_launch": ev / nl, "avg_launch_ms": k_ms, "cells_per_event": d("n_cellsteps") / max(ev, 1),
           "peel_cells_per_event": d("n_cellsteps_peel") / max(ev, 1), "atomics_per_event": atom / max(ev, 1), "lanes": sim.ctx.pool_info()[0], "block": sim.ctx.pool_info()[1],
           "photons_finished": d("nphotons_done"), "events_per_step_per_lane": a.events,
           "roofline": {"bound": "hbm", "achieved": bytes_launch / (k_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": bytes_launch / (k_ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_launch": bytes_launch}}
    print(json.dumps(out), flush=True)
    lib = os.environ.get("LART_GPU_LIB", "")
    if lib.endswith("_prof.so"):                                           # -DLART_PROFILE build: wave clocks per code section (lart_device.hpp, LART_PROF_MARK)
        ptr, nt = sim.ctx.tally_buffer()
        b = host._as_cuda_tensor(ptr, 32, sim.ctx).cpu().numpy()
        sc = list(b[8:20]) + list(b[25:29]); tr = b[20:24]; tot = sum(sc)
        names = ["pool", "birth", "tau+first", "raytrace_to_tau", "scatter_pre", "vz_core_loop", "post_rest(ux..emit)", "finish+rotate", "vz_wing_setup", "vz_wing_loop", "vz_wing_tan", "mid_fill",
                 "post_head(E,cost,S)", "phi_rejection", "perp_velocity", "peel_emit"]
        print(json.dumps({"section_clock_share": {k: round(float(v / tot), 4) for k, v in zip(names, sc)},
                          "wave_trips_per_round": {"core_tries": tr[0] / tr[2], "wing_tries": tr[1] / tr[2], "wing_entered": tr[3] / tr[2]}, "clocks_per_round": float(tot / tr[2])}), flush=True)
    if lib.endswith("_rs.so"):                                             # -DLART_RNG_STATS build: single-block evaluations off the draw ring
        ptr, nt = sim.ctx.tally_buffer()
        b = host._as_cuda_tensor(ptr, 32, sim.ctx).cpu().numpy(); ev_all = t1["n_events"]
        print(json.dumps({"slow_blocks_per_event": float(b[8] / ev_all), "wave_detours_per_wave_event": float(b[9] / (ev_all / 64.0))}), flush=True)
    if os.environ.get("LART_GPU_LIB", "").endswith("_st.so"):             # -DLART_OVL_STATS build: how the flights of the overlap march divide
        ptr, nt = sim.ctx.tally_buffer()
        b = host._as_cuda_tensor(ptr, 32, sim.ctx).cpu().numpy(); ev_all = t1["n_events"]
        names = ["carried_list", "rebuilds", "rebuild_fast", "rebuild_fast_cached", "marches", "cells", "intervals", "holders_at_march_start"]
        print(json.dumps({"per_event": {k: float(b[8 + i] / ev_all) for i, k in enumerate(names)}, "cells_per_march": float(b[13] / max(b[12], 1)), "intervals_per_march": float(b[14] / max(b[12], 1)),
                          "binding_bound_share": {k: float(b[16 + i] / max(b[16:20].sum(), 1)) for i, k in enumerate(["cell_face", "sphere", "other_clump", "own_clump"])},
                          "mean_reach": float(b[20] / max(b[16:20].sum(), 1)), "mean_flight_capped10": float(b[21] / max(b[16:20].sum(), 1)), "more_than_4_holders": float(b[22] / max(b[16:20].sum(), 1)),
                          "mean_dface": float(b[23] / max(b[16:20].sum(), 1)),
                          "cgx": int(sim.clumps.cgx), "cg_d": float(sim.clumps.cg_d) if hasattr(sim.clumps, "cg_d") else None}))
    sim.ctx.close()


if __name__ == "__main__":
    main()
