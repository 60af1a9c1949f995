"""The N > 1 path without GPUs: world_size-2 gloo processes.  Each rank transports the photon ids the product's
decomposition gives it (here with the CPU oracle standing in for the device, same Philox streams), the fused tally
buffers are summed by the product's reduce_tally_buffer(), and the result must equal a single-rank run."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARAMS = "temperature=1e4 taumax=30 nx=21 ny=21 nz=21 rmax=1 nxfreq=61 xyz_symmetry=1"
N, SEED = 4001, 77        # odd count: the two ranks get different shares


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lart_amd import host
    from oracle import oracle as O
    count = host.photons_of_rank(N, rank, world)
    o = O.Oracle(PARAMS)
    o.run(count, first_id=rank + 1, stride=world, rng_kind=1, seed=SEED, nthreads=2)
    buf = torch.from_numpy(np.concatenate([[o.scalar("nscatt_gas"), float(count)], o.array("Jout"), o.array("Jin")]))
    host.reduce_tally_buffer(buf, dist)
    # par%iseed = 0: one seed for the job, drawn on rank 0 (clump placement and photon streams must agree between the ranks); a given seed is left alone
    s0 = host.job_seed(0, dist); own = host.job_seed(0)
    both = [None, None]; dist.all_gather_object(both, (s0, own, host.job_seed(12345, dist)))
    if rank == 0:
        np.save(out, buf.numpy())
        assert both[0][0] == both[1][0] and 0 < s0[0] < 2 ** 53 and 0 < s0[1] < 2 ** 31 and s0[1] == (s0[0] % 2147483647 or 1)
        assert both[0][1] != both[1][1] and both[0][2] == both[1][2] == (12345, 12345)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank(oracle_mod, tmp_path):
    from lart_amd import host
    assert host.photons_of_rank(N, 0, 2) + host.photons_of_rank(N, 1, 2) == N
    assert [host.photons_of_rank(10, r, 4) for r in range(4)] == [3, 3, 2, 2]
    assert host.photons_of_rank(2, 3, 4) == 0
    out = str(tmp_path / "sum.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    o = oracle_mod.Oracle(PARAMS)
    o.run(N, first_id=1, stride=1, rng_kind=1, seed=SEED)
    ref = np.concatenate([[o.scalar("nscatt_gas"), float(N)], o.array("Jout"), o.array("Jin")])
    # same photon histories (streams keyed by id), only the f64 summation order differs
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-9)
