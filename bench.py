#!/usr/bin/env python3
"""bench.py -- scattering events/s of the MI355X photon-packet path on BASELINE.json's headline workload.

Workload (config.workload): examples/sphere/t1tau6.in physics = static uniform sphere, tau0 = 1e6, T = 10 K,
101^3 cells in the +++ octant (xyz_symmetry), point source, Voigt emission, 201 frequency bins, 1e7 photons
per GPU ("weak" scaling: each rank owns the photon ids rank+1, rank+1+N, ... of an N x 1e7 photon job, no
data-path collective; one RCCL sum of the fused tally buffer after the timed region).

A "step" = one transport-kernel launch in which every resident photon lane advances by at most
--events-per-step scattering events (lanes refill from the photon counter when their packet escapes).
The timed region holds exactly K steps between barrier + device synchronisation; inputs are resident in HBM.
value = scattering events executed by all ranks / max-over-ranks wall time.

roofline.achieved = algorithmic bytes of the transport kernel per launch (48 B per cell visited by a flight:
6 f64 medium gathers; + 16 B per tally atomic) / its average launch duration from HIP
events on the launch stream.  The kernel is fp64-ALU/latency bound, not HBM bound: the fraction
is expected to be tiny and is reported as such.
cpu_baseline = the CPU oracle (a port, oracle/lart_oracle.c) on a bounded sample of the same workload on the
host cores of this box (rank 0, N = 1 only).

After the timed region (not part of `value`): --completion-photons photons per GPU of the same workload are run to the last
photon.  That gives (a) the other half of BASELINE.json's metric, "spectrum_rms_vs_ref" = rms difference of the normalised
emergent spectrum Jout against the committed tau0 = 1e6 reference spectrum tests/golden/c2_tau6.npz (CPU oracle on the
reference's MT19937-64 stream, tools/gen_golden_bigtau.py), quoted next to the Poisson-expected rms of the two runs, and
(b) "run_to_completion" = total events / wall time including the drain tail (lane compaction).

--gpus N > 1 without a torchrun environment: this process starts `python -m torch.distributed.run --nproc-per-node N` on
itself as a child BEFORE touching the GPU and returns its exit code; the printed n_gpus is the rank count RCCL reported for
the library's own communicator (lart_gpu_comm_info), and the run fails if that differs from --gpus.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(temperature=10.0, taumax=1.0e6, DGR=0.0, Vexp=0.0, xyz_symmetry=True, comoving_source=False, recoil=False,
                spectral_type="voigt", nx=101, ny=101, nz=101, nxfreq=201, rmax=1.0, no_photons=1.0e7, iseed=20261003)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--events-per-step", type=int, default=4096)
    ap.add_argument("--lanes", type=int, default=0, help="resident photon lanes per GPU (0 = library default)")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--cpu-photons", type=int, default=0, help="oracle sample size (0 = auto, ~20 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the tally reduction even with one rank (rehearsal of the N>1 path)")
    ap.add_argument("--completion-photons", type=int, default=3000000, help="photons per GPU run to the last photon after the timed region "
                    "(spectrum_rms_vs_ref + run_to_completion); 0 = skip.  3e6 keeps the whole command within ~6 minutes; BASELINE.json's 1e7 takes ~12")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short steady-state legs of BASELINE.json configs[2] (C3) and configs[4] (C5)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start one rank per GPU as a child job (nothing in this process has touched the GPU yet) and hand its code back
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s)" % (a.gpus, world))
    import torch
    import torch.distributed as dist
    use_dist = world > 1 or a.force_dist
    if use_dist:
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)

    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "lart_amd", "csrc", "liblart_gpu.so")):
        ge.build()
    from lart_amd import host

    sim = host.Simulation(dict(WORKLOAD), device=local)
    if a.lanes or a.block:
        sim.ctx.set_pool(a.lanes, a.block)
    nph = int(WORKLOAD["no_photons"])
    # rank r of an (N x 1e7)-photon job: ids r+1, r+1+N, ...   (run_simulation_mod.f90:150-151)
    sim.ctx.begin(nph, first_id=rank + 1, stride=world, seed=WORKLOAD["iseed"])

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    lanes_timed, block_timed = sim.ctx.pool_info()
    for _ in range(a.warmup):
        sim.ctx.step(1, a.events_per_step)
    sim.ctx.sync()
    t_a = sim.ctx.tallies()
    ms_a, nl_a = sim.ctx.kernel_time()
    barrier()
    t0 = time.perf_counter()
    sim.ctx.step(a.steps, a.events_per_step)
    sim.ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    t_b = sim.ctx.tallies()
    ms_b, nl_b = sim.ctx.kernel_time()

    ev = t_b["n_events"] - t_a["n_events"]
    cells = t_b["n_cellsteps"] - t_a["n_cellsteps"]
    done = t_b["nphotons_done"] - t_a["nphotons_done"]
    tot = torch.tensor([ev, cells, float(done), dt], dtype=torch.float64, device="cuda")
    if use_dist:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(mx[3])
    ev_all, cells_all, done_all = float(tot[0]), float(tot[1]), float(tot[2])

    # the path's only exchange step: one RCCL reduce(sum, f64) of the fused tally buffer onto rank 0 (output_reduce), done by the
    # library's own communicator; the first call also creates it, so time a second one
    sim.output_reduce(dist if use_dist else None, force=a.force_dist)
    n_rccl = sim.ctx.comm_info()[0] if use_dist else 1
    if n_rccl != world:
        sys.exit("bench.py: RCCL reports %d rank(s), expected %d" % (n_rccl, world))
    t_r0 = time.perf_counter()
    sim.output_reduce(dist if use_dist else None, force=a.force_dist)
    reduce_ms = (time.perf_counter() - t_r0) * 1e3
    _, ntally = sim.ctx.tally_buffer()

    # BASELINE.json configs[2] (sphere_peel, the designated 1 -> 8 GPU workload) runs on every rank of an N-GPU job too: rank r takes the ids r+1, r+1+N, ...
    # of the 1e8-photon job; configs[4] (octree) only at N = 1
    others = other_configs(local, a.events_per_step, rank, world, dist if use_dist else None, barrier) if not a.no_other_configs else None
    completion = run_to_completion(sim, a, rank, world, dist if use_dist else None, barrier) if a.completion_photons > 0 else None

    if rank == 0:
        launches = max(nl_b - nl_a, 1)
        k_ms = (ms_b - ms_a) / launches
        # algorithmic bytes per launch on THIS rank: 48 B per visited cell + 16 B per tally atomic
        # (Jout escape + forced-first-scatter fraction: 2 per finished photon; scalar counters are per wave)
        bytes_launch = (48.0 * cells + 16.0 * 2.0 * done) / launches
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(a.events_per_step)
        out = {
            "metric": "scattering events/sec (whole node), tau0=1e6 sphere",
            "value": ev_all / dt, "unit": "events/s", "n_gpus": n_rccl, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "examples/sphere/t1tau6.in: static uniform sphere tau0=1e6, T=10K, 101^3 xyz_symmetry octant, "
                                   "point source, Voigt line, nxfreq=201, 1e7 photons per GPU (BASELINE.json configs[1])",
                       "events_per_step_per_lane": a.events_per_step, "lanes_per_gpu": lanes_timed, "block": block_timed,
                       "cells_per_event": cells_all / max(ev_all, 1.0), "photons_finished_in_timed_region": done_all,
                       "tally_reduce_ms": reduce_ms, "tally_doubles": ntally, "parallelism": "photon-dp%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "transport_kernel<1,false,false,false,1>", "avg_launch_ms": k_ms,
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "note": "fp64 ALU/latency-bound path; see DESIGN.md 'Roofline' for the VALU accounting"},
        }
        if others is not None:
            out["other_configs"] = others
        if completion is not None:
            out["spectrum_rms_vs_ref"] = completion.pop("spectrum_rms_vs_ref")
            out["spectrum"] = completion.pop("spectrum")
            out["run_to_completion"] = completion
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_photons)
            ref_core = 2.8e6                                                  # the reference's own Fortran, 1 core, this workload (measured where /root/reference is mounted)
            out["cpu_baseline"]["reference_events_per_s_per_core"] = ref_core
            out["cpu_baseline"]["port_vs_reference_per_core"] = out["cpu_baseline"]["value"] / out["cpu_baseline"]["cores"] / ref_core
            out["cpu_baseline"]["note"] = "kind 'port': the builder's C restatement, faster per core than the reference's Fortran by the factor port_vs_reference_per_core; a GPU / cpu_baseline ratio is not GPU vs reference"
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def run_to_completion(sim, a, rank, world, dist, barrier):
    """--completion-photons photons per GPU of the same workload, to the last photon: throughput with the drain tail, and the
    emergent spectrum against the committed tau0 = 1e6 reference spectrum (outside the timed region of `value`)."""
    import numpy as np
    import torch
    nph = a.completion_photons * world
    sim.ctx.zero_tallies()
    count = (nph - rank - 1) // world + 1 if nph > rank else 0
    barrier()
    t0 = time.perf_counter()
    sim.ctx.begin(count, first_id=rank + 1, stride=world, seed=WORKLOAD["iseed"] + 1)
    launches = 0; t_msg = t0
    while True:
        sim.ctx.step(1, a.events_per_step); launches += 1
        done, left = sim.ctx.sync()
        if left == 0:
            break
        if rank == 0 and time.perf_counter() - t_msg > 30.0:          # a long leg stays visibly alive (stderr: stdout carries the one JSON line)
            t_msg = time.perf_counter()
            print("bench.py: run to completion, %.0f s, %d photons of this rank still in flight or waiting" % (t_msg - t0, left), file=sys.stderr, flush=True)
    barrier()
    dt = time.perf_counter() - t0
    live, ncompact = sim.ctx.live_lanes()
    if dist is not None:
        mx = torch.tensor([dt], dtype=torch.float64, device="cuda"); dist.all_reduce(mx, op=dist.ReduceOp.MAX); dt = float(mx[0])
    sim.output_reduce(dist, force=a.force_dist)
    if rank != 0:
        return None
    t = sim.ctx.tallies()
    out = {"photons": int(t["nphotons_done"]), "events": t["n_events"], "seconds": dt, "value": t["n_events"] / dt, "unit": "events/s",
           "launches": launches, "lane_compactions": ncompact, "mean_scatterings": t["nscatt_gas"] / max(t["nphotons_done"], 1)}
    fx = os.path.join(ROOT, "tests", "golden", "c2_tau6.npz")
    out["spectrum_rms_vs_ref"] = None; out["spectrum"] = {"fixture": None}
    if os.path.exists(fx):
        d = np.load(fx)
        ng, nr = float(t["nphotons_done"]), float(d["nphotons"])
        jg = t["Jout"] / (float(d["Jout_scale_per_photon"]) * ng)
        jr = d["Jout"]
        pk = jr.max()
        # Poisson expectation of the difference of two independent runs: a bin that holds the fraction p_i of the photons has the
        # variance p_i / N in each run (weights are 1 - exp(-tau0) = 1 here); p_i from the two runs pooled
        p_i = (t["Jout"] + d["Jout_raw"]) / (ng + nr)
        var = p_i * (1.0 / ng + 1.0 / nr) / float(d["Jout_scale_per_photon"]) ** 2
        out["spectrum_rms_vs_ref"] = float(np.sqrt(np.mean((jg - jr) ** 2)) / pk)
        out["spectrum"] = {"fixture": "tests/golden/c2_tau6.npz", "definition": "sqrt(mean_i (J_gpu,i - J_ref,i)^2) / max_i J_ref,i over the nxfreq bins of normalised Jout",
                           "rms_expected_poisson": float(np.sqrt(np.mean(var)) / pk), "ref_photons": int(nr), "gpu_photons": int(ng),
                           "chi2_per_bin": float(np.mean((jg - jr)[var > 0] ** 2 / var[var > 0])),
                           "mean_scatterings_gpu": out["mean_scatterings"], "mean_scatterings_ref": float(d["nscatt_mean"]), "mean_scatterings_ref_err": float(d["nscatt_err"])}
    return out


def pmc_traffic(events_per_step):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/*pmc*.json, separate
    FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled as the guide's gfx950 correction prescribes).  None when no pass
    exists for this events-per-step setting."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary*.json")), reverse=True):
        try:
            d = json.load(open(f))
            if "bench.py" in d.get("command", "") and int(d.get("events_per_step", -1)) == int(events_per_step) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                return (2.0 * d["FETCH_SIZE"]["mean_per_launch"] + d["WRITE_SIZE"]["mean_per_launch"]) * 1024.0, "profiles/" + os.path.basename(f) + " (committed rocprofv3 --pmc passes of this command, not measured in this run)"
        except Exception:
            pass
    return None, None


def config_traffic(tag):
    """HBM bytes per launch of the transport kernel of one of the other configurations, from the committed rocprofv3 PMC passes of `tools/perf_kernel.py --config TAG`
    (newest profiles/r*_TAG_pmc_summary.json; 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes, as pmc_traffic)"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc_summary.json" % tag)), reverse=True):
        try:
            d = json.load(open(f))
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                return (2.0 * d["FETCH_SIZE"]["mean_per_launch"] + d["WRITE_SIZE"]["mean_per_launch"]) * 1024.0, "profiles/" + os.path.basename(f)
        except Exception:
            pass
    return None, None


def other_configs(device, events, rank=0, world=1, dist=None, barrier=None):
    """short steady-state legs of the two other GPU configurations of BASELINE.json that fit one device: configs[2] (C3: sphere_peel/t1tau6.in, 201^3, Stokes,
    129 x 129 x 201 peel cube) and configs[4] (C5: the 178 480-leaf octree with calc_J + calc_P + calc_Pnew).  3 warm-up steps (the pool has to reach its
    steady mix of packet ages), 3 timed steps of `events` scattering events per lane; each with the roofline of SURVEY section 8(d): 48 B per cell visited by a
    flight or a peel ray + 16 B per tally atomic, over the average launch time from HIP events on the launch stream.  Not part of `value`.

    With N > 1 ranks the C3 leg is BASELINE.json configs[2] as written -- ONE 1e8-photon job split over the GPUs (rank r: ids r+1, r+1+N, ...; run_simulation_mod.f90:150-151),
    timed between barriers, max over ranks, followed by the job's one exchange step: the library's ncclReduce of the fused buffer with the 129 x 129 x 201 cubes
    (output_sum_rect.f90:107-127), timed on its second call."""
    import tempfile
    import torch
    from lart_amd import host, amr
    out = {}
    tmp = tempfile.mkdtemp()
    afile = os.path.join(tmp, "sphere.dat")
    cfgs = {
        "c3_sphere_peel_tau6": dict(temperature=10.0, taumax=1e6, nx=201, ny=201, nz=201, rmax=1.0, nxfreq=201, use_stokes=True, save_Jin=True, nxim=129, nyim=129, no_photons=1e8, iseed=1),
        "c5_octree_JP": dict(use_amr_grid=True, amr_file=afile, temperature=1e4, taumax=1e4, rmax=1.0, nxfreq=121, nxim=100, nyim=100, calc_J=True, calc_P=True, calc_Pnew=True,
                             no_photons=1e8, iseed=1),
    }
    tags = {"c3_sphere_peel_tau6": "c3", "c5_octree_JP": "c5obs"}
    for name, kw in cfgs.items():
        if world > 1 and name != "c3_sphere_peel_tau6":
            continue
        if kw.get("use_amr_grid"):
            amr.write_generic_amr(afile, amr.make_amr_sphere())
        sim = host.Simulation(dict(kw), device=device)
        nph = int(kw["no_photons"])
        sim.ctx.begin((nph - rank - 1) // world + 1, first_id=rank + 1, stride=world, seed=1)
        for _ in range(3):
            sim.ctx.step(1, events)
        sim.ctx.sync()
        t0 = sim.ctx.tallies(); ms0, nl0 = sim.ctx.kernel_time()
        if barrier:
            barrier()
        w0 = time.perf_counter()
        sim.ctx.step(3, events); sim.ctx.sync()
        if barrier:
            barrier()
        dt = time.perf_counter() - w0; t1 = sim.ctx.tallies(); ms1, nl1 = sim.ctx.kernel_time()
        d = lambda k: t1[k] - t0[k]
        ev_all = d("n_events"); reduce_ms = None; ntally = sim.ctx.tally_buffer()[1]
        if dist is not None:
            tot = torch.tensor([ev_all, dt], dtype=torch.float64, device="cuda")
            mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            ev_all, dt = float(tot[0]), float(mx[1])
            sim.output_reduce(dist)                              # creates the communicator of this context; the second call is the exchange step alone
            barrier(); r0 = time.perf_counter(); sim.output_reduce(dist); barrier()
            reduce_ms = (time.perf_counter() - r0) * 1e3
        nl = max(nl1 - nl0, 1); k_ms = (ms1 - ms0) / nl
        by = (48.0 * (d("n_cellsteps") + d("n_cellsteps_peel")) + 16.0 * d("n_atomics")) / nl
        ach = by / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = config_traffic(tags[name])
        out[name] = {"value": ev_all / dt, "unit": "events/s", "n_gpus": world, "scaling": "strong" if world > 1 else None, "steps": 3, "events_per_step_per_lane": events,
                     "lanes": sim.ctx.pool_info()[0],
                     "cells_per_event": d("n_cellsteps") / max(d("n_events"), 1), "peel_cells_per_event": d("n_cellsteps_peel") / max(d("n_events"), 1),
                     "atomics_per_event": d("n_atomics") / max(d("n_events"), 1), "tally_doubles": ntally, "tally_reduce_ms": reduce_ms,
                     "roofline": {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                                  "avg_launch_ms": k_ms, "algorithmic_bytes_per_launch": by}}
        sim.ctx.close()
    return out if rank == 0 else None


def host_cores():
    """usable host cores: the cgroup CPU quota when there is one (a 1-GPU box gets a 16-core share of a 256-thread host),
    else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return min(n, int(os.environ.get("LART_CPU_CORES", "16")))


def cpu_baseline(nphot):
    """the CPU oracle (kind 'port') on the first photons of the same workload, all host cores"""
    from oracle import oracle as O
    O.build(ref=False)
    cores = host_cores()
    kw = {k: v for k, v in WORKLOAD.items() if k not in ("iseed", "no_photons")}
    o = O.Oracle(**kw)
    if nphot <= 0:
        # ~0.4 core-seconds per photon at tau0 = 1e6 (1.0e6 events at ~2.8e6 events/s/core): aim at ~20 s of wall time
        nphot = max(cores * 2, int(20.0 * cores / 0.4))
        nphot = (nphot // cores) * cores
    t0 = time.perf_counter()
    o.run(nphot, first_id=1, stride=1, rng_kind=1, seed=WORKLOAD["iseed"], nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": o.scalar("n_events") / dt, "unit": "events/s", "cores": cores, "kind": "port",
            "sample": "first %d photons (ids 1..%d) of the same workload run to completion by oracle/lart_oracle.c with %d OpenMP threads, %.1f s"
                      % (nphot, nphot, cores, dt),
            "mean_scatterings": o.scalar("nscatt_gas") / nphot}


if __name__ == "__main__":
    main()
