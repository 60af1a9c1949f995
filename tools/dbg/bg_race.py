#!/usr/bin/env python3
"""Does a run give the same tallies when other contexts of the same process keep the GPU busy?  (round 4: three 'LaRT_amd.x equals the Python host' tests
failed while the full-size configurations ran in the background.)  Runs each small case alone, then again next to a tau0 = 1e6 sphere on a second context
driven by its own thread, and prints what differs."""
import os, sys, threading, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
from lart_amd import host

HEAD = ("&parameters\n par%no_photons = 20000\n par%temperature = 1e4\n par%N_HImax = 1e16\n par%rmax = 1.0\n par%nx = 31\n par%ny = 31\n par%nz = 31\n"
        " par%comoving_source = .false.\n par%spectral_type = 'monochromatic'\n par%velocity_min = -400.0\n par%velocity_max = 400.0\n par%nvelocity = 80\n"
        " par%nxim = 9\n par%nyim = 9\n par%distance = 1e4\n par%alpha = -90.0, -90.0\n par%beta = 0.0, 60.0\n par%iseed = 21\n")
CASES = {"rot": HEAD + " par%velocity_type = 'rotating_galaxy_halo'\n par%Vrot = 100.0\n par%rinner = 0.2\n/\n",
         "ssh": HEAD + " par%velocity_type = 'ssh'\n par%Vpeak = 120.0\n par%rpeak = 0.3\n par%DeltaV = -60.0\n/\n",
         "ovl": ("&parameters\n par%no_photons = 20000\n par%use_clump_medium = .true.\n par%rmax = 1.0\n par%clump_radius = 0.3\n par%clump_f_cov = 2.0\n par%clump_tau0 = 30.0\n"
                 " par%temperature = 1e4\n par%clump_allow_overlap = .true.\n par%clump_sigma_v = 100.0\n par%nxfreq = 101\n par%velocity_min = -200.0\n par%velocity_max = 200.0\n"
                 " par%nx = 11\n par%ny = 11\n par%nz = 11\n par%iseed = 7\n/\n")}


def run_case(name, tmp):
    fn = os.path.join(tmp, name + ".in"); open(fn, "w").write(CASES[name])
    sim = host.Simulation(host.read_namelist(fn)); sim.ctx.enable_photon_log(True)
    sim.run_simulation()
    t = sim.ctx.tallies()
    return t, sim.ctx.photon_log(20000)


def main():
    tmp = tempfile.mkdtemp(prefix="bgrace")
    alone = {k: run_case(k, tmp) for k in CASES}
    stop = threading.Event()
    import time
    t_start = time.time()
    if len(sys.argv) > 1 and sys.argv[1] == "bigtau":          # the background of the device suite: the five full-size configurations of tests/test_gpu_bigtau.py
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import conftest, test_gpu_bigtau
        def drive():
            test_gpu_bigtau.run_all()
            print("background finished at %.1f s" % (time.time() - t_start), flush=True)
        stop = conftest.STOP
    else:
        big = host.Simulation(dict(temperature=10.0, taumax=1e6, xyz_symmetry=True, nx=101, ny=101, nz=101, rmax=1.0, nxfreq=201, iseed=5))

        def drive():
            big.ctx.begin(200000, first_id=1, stride=1, seed=5)
            while not stop.is_set():
                big.ctx.step(1, 4096)
                if big.ctx.sync()[1] == 0:
                    break
    th = threading.Thread(target=drive); th.start()
    try:
        for rep in range(40 if len(sys.argv) > 1 else 2):
            if not th.is_alive() and rep > 2:
                break
            print("--- pass %d at %.1f s" % (rep, time.time() - t_start), flush=True)
            for k in CASES:
                t, (ns, xs) = run_case(k, tmp); t0, (ns0, xs0) = alone[k]
                line = ["%s rep %d:" % (k, rep), "done %d/%d" % (t["nphotons_done"], t0["nphotons_done"]), "histories identical %.6f" % np.mean((ns == ns0) & (xs == xs0))]
                for key in sorted(t0):
                    a, b = np.asarray(t0[key], float), np.asarray(t[key], float)
                    if a.shape != b.shape:
                        line.append("%s shape %s/%s" % (key, a.shape, b.shape)); continue
                    if a.size and not np.allclose(a, b, rtol=1e-9, atol=1e-12 * max(1.0, float(np.abs(a).max()) if a.size else 1.0)):
                        line.append("%s DIFFERS (max abs %.3e, sums %.6e / %.6e)" % (key, float(np.abs(a - b).max()), float(a.sum()), float(b.sum())))
                print(" ".join(line), flush=True)
    finally:
        stop.set(); th.join()


if __name__ == "__main__":
    main()
