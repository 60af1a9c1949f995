"""bench.py / host logic that needs no GPU: the rank-count check that keeps `n_gpus` honest, the photon-id decomposition of the two hosts,
the seed rule for par%iseed = 0."""
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """`--gpus N` must equal the ranks the launcher started; checked before anything touches torch or the GPU"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and "--gpus 2" in (r.stdout + r.stderr) and "1 rank" in (r.stdout + r.stderr)


def test_photon_ids_of_a_rank_cover_the_job_once():
    """rank r of N transports the ids r+1, r+1+N, ... <= nphotons (run_simulation_mod.f90:150-151): the Python formula, the Fortran host's
    `(nphotons - r)/ngpu + 1` (main.f90, 1-based r) and lart_gpu_run_multi's `(nphotons - 1 - r)/n + 1` agree and partition 1..nphotons"""
    from lart_amd import host
    for nph in (0, 1, 7, 8, 9, 1000, 12345):
        for n in (1, 2, 3, 8):
            py = [host.photons_of_rank(nph, r, n) for r in range(n)]
            fortran = [((nph - r) // n + 1) if nph >= r else 0 for r in range(1, n + 1)]
            c_abi = [((nph - 1 - r) // n + 1) if nph > r else 0 for r in range(n)]
            assert py == fortran == c_abi and sum(py) == nph
            ids = sorted(i for r in range(n) for i in range(r + 1, nph + 1, n))
            assert ids == list(range(1, nph + 1))


def test_entropy_seed_fits_both_hosts():
    """par%iseed = 0: lart_host_entropy_seed (CPU entry point of the library) gives a non-zero 53-bit value, different from call to call"""
    import __graft_entry__ as g
    g.build()
    from lart_amd import capi
    a, b = capi.entropy_seed(), capi.entropy_seed()
    assert 0 < a < 2 ** 53 and 0 < b < 2 ** 53 and a != b
    assert float(a) == a                                   # exact in the f64 header keyword ISEED


def test_oracle_thread_count_respects_the_cpu_quota():
    from oracle import oracle as O
    n = O.usable_cores()
    assert 1 <= n <= min(64, os.cpu_count() or 1)
