"""Scheduling / exchange features of the HIP path on a real MI355X: lane compaction in the drain phase, the library's own RCCL
reduce (one rank and the one-process multi-context entry point), the setters' "end the current run" rule, run_multi.

None of them may change a result: a packet's history is a function of (seed, id) only, so every check here is bit-for-bit
against the same library without the feature (and, for the histories, against the CPU oracle)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(temperature=1e4, taumax=3e3, xyz_symmetry=True, nx=31, ny=31, nz=31, rmax=1.0, nxfreq=101, iseed=9)


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from lart_amd import capi
    assert capi.lib().lart_gpu_device_count() > 0, "no GPU visible: the HIP path cannot run (there is no fallback)"
    return capi


def make(kw, device=0):
    from lart_amd import host
    return host.Simulation(dict(kw), device)


def drain(sim, n, events, compact):
    sim.ctx.set_pool(8192, 256); sim.ctx.set_auto_compact(compact); sim.ctx.enable_photon_log(True)
    sim.ctx.begin(n, seed=9)
    lanes = [sim.ctx.pool_info()[0]]
    while True:
        sim.ctx.step(1, events)
        done, active = sim.ctx.sync()
        lanes.append(sim.ctx.live_lanes()[0])
        if active == 0:
            break
    return sim.ctx.photon_log(n), sim.ctx.tallies(), lanes


def test_lane_compaction_changes_nothing_but_the_grid(capi, oracle_mod):
    """8192 lanes, 8192 photons: no refill at all, the whole run is a drain.  With compaction the launches shrink with the live
    lanes; per-photon scattering counts and exit frequencies are identical with and without it, and equal the oracle's."""
    n = 8192
    (na, xa), ta, la = drain(make(KW), n, 256, True)
    (nb, xb), tb, lb = drain(make(KW), n, 256, False)
    assert np.array_equal(na, nb) and np.array_equal(xa, xb)
    assert np.allclose(ta["Jout"], tb["Jout"], rtol=0, atol=1e-9) and ta["nphotons_done"] == tb["nphotons_done"] == n
    assert ta["n_events"] == tb["n_events"]
    assert min(lb) == max(lb) == 8192                      # feature off: the grid never changes
    assert la[-1] <= 512 and sorted(la, reverse=True) == la and len(set(la)) >= 4, la
    o = oracle_mod.Oracle(**{k: v for k, v in KW.items() if k != "iseed"}); o.run(n, seed=9)
    no, xo = o.photon_log()
    same = np.isclose(na, no, rtol=1e-12, atol=0) & np.isclose(xa, xo, rtol=1e-7, atol=1e-9)
    assert same.mean() > 0.999


def test_drain_compaction_layouts_with_the_shear_plane(capi, monkeypatch):
    """the drain-phase re-layout (lart_gpu_compact: round-robin spread over the wave slots below 64 x W packets, dense above; holes / movers lists) on a pool that carries
    the extra state plane of the shearing box (F_SHEAR, VAR 8) and a Stokes vector: the run with automatic compaction, the same with LART_DENSE_DRAIN=1 (dense layout only)
    and the run without compaction give identical per-photon histories, and the first two did compact (the bench contract test no longer pins this)"""
    kw = dict(temperature=1e4, taumax=3e2, nx=6, ny=4, nz=21, xmax=0.5, ymax=0.3, zmax=1.0, geometry="rectangle", xy_periodic=True, Omega=2.5, nxfreq=81, use_stokes=True, iseed=9)
    n = 8192
    (na, xa), ta, la = drain(make(kw), n, 64, True)
    monkeypatch.setenv("LART_DENSE_DRAIN", "1")
    (nd, xd), td, ld = drain(make(kw), n, 64, True)
    monkeypatch.delenv("LART_DENSE_DRAIN")
    (nb, xb), tb, lb = drain(make(kw), n, 64, False)
    assert np.array_equal(na, nb) and np.array_equal(xa, xb) and np.array_equal(nd, nb) and np.array_equal(xd, xb)
    assert ta["nphotons_done"] == td["nphotons_done"] == tb["nphotons_done"] == n and ta["n_events"] == tb["n_events"] == td["n_events"]
    assert np.allclose(ta["Jout"], tb["Jout"], rtol=0, atol=1e-9) and np.allclose(td["Jout"], tb["Jout"], rtol=0, atol=1e-9)
    assert min(lb) == max(lb) == 8192 and la[-1] < 8192 and ld[-1] < 8192 and len(set(la)) >= 3 and len(set(ld)) >= 3, (la[-5:], ld[-5:])


def test_forced_compaction_mid_run_with_stokes_planes(capi):
    """lart_gpu_compact by hand in the middle of a Stokes run (19 state planes move): same histories as an undisturbed run"""
    kw = dict(KW, taumax=300.0, use_stokes=True, xyz_symmetry=False, nx=21, ny=21, nz=21)
    n = 4096
    a = make(kw); a.ctx.set_pool(4096, 64); a.ctx.set_auto_compact(False); a.ctx.enable_photon_log(True); a.ctx.begin(n, seed=5)
    k = 0
    while True:
        a.ctx.step(1, 40); k += 1
        if a.ctx.sync()[1] == 0:
            break
        if k % 3 == 0:
            a.ctx.compact()
    assert a.ctx.live_lanes()[1] >= 2
    b = make(kw); b.ctx.set_auto_compact(False); b.ctx.enable_photon_log(True); b.run_simulation(n, seed=5)
    na, xa = a.ctx.photon_log(n); nb, xb = b.ctx.photon_log(n)
    assert np.array_equal(na, nb) and np.array_equal(xa, xb)


def test_rccl_reduce_entry_points_one_rank(capi):
    """the library's RCCL communicator with the one rank a 1-GPU box has: lart_gpu_comm_init / _reduce_tallies and the one-process
    multi-context entry point leave the totals unchanged; RCCL itself reports the rank count"""
    sim = make(KW); sim.run_simulation(3000)
    before = sim.ctx.tallies()
    sim.ctx.comm_init(capi.comm_unique_id(), 1, 0)
    assert sim.ctx.comm_info() == (1, 0)
    sim.ctx.reduce_tallies(0)
    capi.reduce_tallies_multi([sim.ctx], 0)
    after = sim.ctx.tallies()
    assert np.array_equal(before["Jout"], after["Jout"]) and before["nscatt_gas"] == after["nscatt_gas"]
    with pytest.raises(capi.LartError):
        sim.ctx.reduce_tallies(1)                          # root outside the communicator
    sim.ctx.comm_destroy()
    assert sim.ctx.comm_info()[0] == 0
    with pytest.raises(capi.LartError):
        sim.ctx.reduce_tallies(0)                          # no communicator any more


def test_reduce_twice_gives_the_same_totals(capi):
    """output_reduce is idempotent (MPI_REDUCE has separate send and receive buffers, memory_mod_mpi.f90:377-393): the root receives into a buffer of
    its own, this rank's sums stay as they are, and new events invalidate the received totals"""
    sim = make(KW); sim.run_simulation(3000)
    before = sim.ctx.tallies()
    sim.ctx.comm_init(capi.comm_unique_id(), 1, 0)
    for _ in range(3):
        sim.ctx.reduce_tallies(0)
        after = sim.ctx.tallies()
        assert np.array_equal(before["Jout"], after["Jout"]) and before["nscatt_gas"] == after["nscatt_gas"] and after["nphotons_done"] == 3000
    sim.run_simulation(2000, seed=99)                      # the context adds events: what get_tallies hands out is its own sums again
    more = sim.ctx.tallies()
    assert more["nphotons_done"] == 5000 and more["Jout"].sum() > before["Jout"].sum()
    sim.ctx.reduce_tallies(0)
    assert np.array_equal(sim.ctx.tallies()["Jout"], more["Jout"])
    sim.ctx.comm_destroy()


def test_a_larger_run_after_a_small_one_on_the_same_context(capi):
    """the pool (and with it the slot lists of the lane compaction) is re-created when a later lart_gpu_begin asks for more lanes; every run
    ends in a drain, so the compaction runs in both: the second run equals the same run on a fresh context"""
    kw = dict(KW, taumax=300.0)
    a = make(kw); a.ctx.enable_photon_log(True)
    a.run_simulation(500, seed=3)
    lanes_small = a.ctx.pool_info()[0]
    a.ctx.zero_tallies(); a.run_simulation(50000, seed=4)
    assert a.ctx.pool_info()[0] > lanes_small and a.ctx.live_lanes()[1] >= 1
    b = make(kw); b.ctx.enable_photon_log(True); b.run_simulation(50000, seed=4)
    assert np.array_equal(a.ctx.photon_log(50000)[0], b.ctx.photon_log(50000)[0])
    assert np.allclose(a.ctx.tallies()["Jout"], b.ctx.tallies()["Jout"], rtol=0, atol=1e-9)


def test_peel_deposit_list_holds_every_ray_of_a_thin_medium(capi):
    """an optically thin medium with several observers and save_direc0: nearly every (record, observer) pair deposits, the case in which the
    deposit list of the peel pass is fullest; lart_gpu_get_tallies would refuse the run if an entry had been dropped.  The image sums equal
    those of a run with a small pool (many short slices), which fills the list in a different pattern"""
    kw = dict(temperature=1e4, taumax=0.5, nx=11, ny=11, nz=11, rmax=1.0, nxfreq=41, nxim=8, nyim=8, save_peeloff_2D=True, save_direc0=True,
              obsx=[1.0, 0.0, 0.0, -1.0], obsy=[0.0, 1.0, 0.0, 1.0], obsz=[0.0, 0.0, 1.0, 1.0], iseed=12)
    a = make(kw); a.run_simulation(400000); ta = a.ctx.tallies()
    b = make(kw); b.ctx.set_pool(16384, 256); b.run_simulation(400000); tb = b.ctx.tallies()
    assert ta["nphotons_done"] == tb["nphotons_done"] == 400000
    for k in ("scatt", "direc", "direc0"):
        assert np.isclose(ta[k].sum(), tb[k].sum(), rtol=1e-9), k
        assert ta[k].sum() > 0.0


def test_run_multi_one_context_equals_run(capi):
    a = make(KW); a.ctx.enable_photon_log(True); capi.run_multi([a.ctx], 5000, 9)
    b = make(KW); b.ctx.enable_photon_log(True); b.run_simulation(5000, seed=9)
    assert np.array_equal(a.ctx.photon_log(5000)[0], b.ctx.photon_log(5000)[0])
    assert np.allclose(a.ctx.tallies()["Jout"], b.ctx.tallies()["Jout"], rtol=0, atol=1e-9)


def test_run_multi_two_contexts_on_one_device_sum_to_the_one_context_run(capi):
    """the N > 1 decomposition executed on the one GPU a test box has (run_simulation_mod.f90:150-151: rank r of 2 transports the ids r+1, r+3, ...;
    output_sum_rect.f90:107-127: the element-wise sum of the ranks' buffers): lart_gpu_run_multi drives two contexts -- two pools, two HIP streams, two
    fused tally buffers with their per-XCD copies, peel queues included -- on device 0; the two lart_gpu_tally_buffer images summed on the host equal the
    buffer of ONE context that ran all ids (up to the order of the atomic additions), and photon k of context r is photon 2k + r of that run."""
    from lart_amd import host
    kw = dict(temperature=1e4, taumax=3e2, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=61, use_stokes=True, save_Jin=True, nxim=16, nyim=16, iseed=9)
    n = 6001                                                   # odd: the ranks get 3001 and 3000 photons
    a, b = make(kw), make(kw)
    for s in (a, b):
        s.ctx.enable_photon_log(True)
    capi.run_multi([a.ctx, b.ctx], n, 9)
    one = make(kw); one.ctx.enable_photon_log(True); one.run_simulation(n, seed=9)

    def image(s):
        ptr, cnt = s.ctx.tally_buffer()
        return host._as_cuda_tensor(ptr, cnt, s.ctx).cpu().numpy().copy()
    ia, ib, io = image(a), image(b), image(one)
    assert ia.shape == ib.shape == io.shape and ia[32:].sum() > 0 and ib[32:].sum() > 0
    assert ia[5] == 3001 and ib[5] == 3000 and io[5] == n      # nphotons_done per "rank"
    tot = ia + ib
    assert np.allclose(tot[:7], io[:7], rtol=1e-12, atol=0)    # nscatt_gas, nscatt_dust, events, cell steps, peel cell steps, photons
    assert np.allclose(tot[32:], io[32:], rtol=0, atol=1e-12 * np.abs(io[32:]).max())
    ta, to = a.ctx.tallies(), one.ctx.tallies()
    assert to["scatt"].sum() > 0 and to["Q"].any()
    (na, xa), (nb, xb), (no, xo) = a.ctx.photon_log(3001), b.ctx.photon_log(3000), one.ctx.photon_log(n)
    assert np.array_equal(na, no[0::2]) and np.array_equal(nb, no[1::2]) and np.array_equal(xa, xo[0::2]) and np.array_equal(xb, xo[1::2])


def test_setter_after_begin_ends_the_run(capi):
    """a set_* call frees tables the resident pool refers to: the run is over, lart_gpu_step must refuse instead of launching"""
    sim = make(KW)
    sim.ctx.begin(1000, seed=1); sim.ctx.step(1, 16); sim.ctx.sync()
    g = sim.grid
    sim.ctx.set_grid_car(g.nx, g.ny, g.nz, g.xmin, g.ymin, g.zmin, g.dx, g.dy, g.dz, g.i0, g.j0, g.k0,
                         g.xface, g.yface, g.zface, g.rhokap, g.voigt_a, g.Dfreq, g.vfx, g.vfy, g.vfz, g.rhokapD)
    with pytest.raises(capi.LartError, match="begin"):
        sim.ctx.step(1, 16)
    sim.ctx.begin(1000, seed=1); sim.ctx.step(1, 16)      # a fresh begin works
    assert sim.ctx.sync()[0] >= 0


def test_second_run_accumulates_and_normalises_by_finished_photons(capi):
    """device tallies are cumulative over runs; output_normalize divides by the photons the device finished"""
    a = make(KW); a.run_simulation(4000, rank=0, nproc=2); a.run_simulation(4000, rank=1, nproc=2)
    b = make(KW); b.run_simulation(4000)
    ja, jb = a.output_normalize()["Jout"], b.output_normalize()["Jout"]
    assert a.ctx.tallies()["nphotons_done"] == 4000
    assert np.allclose(ja, jb, rtol=1e-12, atol=1e-18)


def test_J_window_in_lds_changes_nothing(capi, monkeypatch):
    """CALCJ: the frequency bins around the line centre are accumulated per workgroup in LDS and flushed once per launch (scattered f64
    atomics run at the memory side at ~1e10/s for the whole chip); the same run with every deposit as a global atomic gives the same J(x, r)
    up to the order of the additions, in a window that does not start at bin 0 and with deposits on both sides of it"""
    kw = dict(temperature=1e4, taumax=3e2, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=161, xfreq_min=-12.0, xfreq_max=20.0, calc_J=True, calc_P=True, calc_Pnew=True, iseed=3)
    a = make(kw); a.run_simulation(20000); ta = a.ctx.tallies()
    monkeypatch.setenv("LART_NO_LDS_J", "1")
    b = make(kw); b.run_simulation(20000); tb = b.ctx.tallies()
    assert ta["J1"].sum() > 0 and np.allclose(ta["J1"], tb["J1"], rtol=1e-11, atol=1e-12 * ta["J1"].max())
    assert np.allclose(ta["P1"], tb["P1"], rtol=1e-11) and np.allclose(ta["P1_new"], tb["P1_new"], rtol=1e-11)
    # the window really is narrower than the axis: mass on both sides of it
    j = ta["J1"].sum(axis=1)
    assert j[:40].sum() > 0 and j[-60:].sum() > 0
    # the run above chose two 512-lane workgroups per CU by itself (round 4; one of 1024 in round 3), the run without the window 256; a forced 256 with the
    # window (one workgroup per CU less) and a forced 1024 (the widest window) are two more layouts of the same sums
    assert a.ctx.pool_info()[1] == 512 and b.ctx.pool_info()[1] == 256
    monkeypatch.delenv("LART_NO_LDS_J")
    for block in (256, 1024):
        c = make(kw); c.ctx.set_pool(0, block); c.run_simulation(20000); tc = c.ctx.tallies()
        assert c.ctx.pool_info()[1] == block
        assert np.allclose(tc["J1"], tb["J1"], rtol=1e-11, atol=1e-12 * ta["J1"].max()) and np.allclose(tc["P1_new"], tb["P1_new"], rtol=1e-11)
    # a Stokes run cannot use the large workgroup (3 waves per SIMD): refused when forced
    d = make(dict(kw, use_stokes=True)); d.ctx.set_pool(0, 1024)
    with pytest.raises(RuntimeError, match="256 lanes"):
        d.run_simulation(1000)


def test_runs_next_to_busy_contexts_of_the_same_process(capi):
    """every fill and copy of a context goes through that context's own stream (round 4: set-up fills on the NULL stream landed after the first kernels of a run
    when other contexts kept the GPU busy and zeroed tallies in mid-run; tools/dbg/bg_race.py).  Four contexts driven by their own threads keep the device busy
    while fresh contexts are created, run and read out on the main thread: every one of those runs has to equal the run made alone, tally for tally"""
    import threading
    kw = dict(temperature=1e4, taumax=3e2, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=81, nxim=9, nyim=9, alpha=10.0, beta=30.0, iseed=21)
    n = 20000
    alone = make(kw); alone.ctx.enable_photon_log(True); alone.run_simulation(n); t0 = alone.ctx.tallies(); ns0, xs0 = alone.ctx.photon_log(n)
    stop = threading.Event(); errors = []
    busy = [make(dict(temperature=10.0, taumax=1e5, xyz_symmetry=True, nx=41, ny=41, nz=41, rmax=1.0, nxfreq=101, iseed=5 + i)) for i in range(4)]
    for b in busy:
        b.ctx.set_pool(262144, 256)                          # (busy, not saturating: the test has to stay short)

    def drive(sim, seed):
        try:
            sim.ctx.begin(10 ** 9, first_id=1, stride=1, seed=seed)
            while not stop.is_set():
                sim.ctx.step(1, 512); sim.ctx.sync()
        except Exception as e:                               # noqa: BLE001 -- reported below
            errors.append(e)
    threads = [threading.Thread(target=drive, args=(s, 40 + i)) for i, s in enumerate(busy)]
    for th in threads:
        th.start()
    try:
        for rep in range(3):
            sim = make(kw); sim.ctx.enable_photon_log(True); sim.run_simulation(n); t = sim.ctx.tallies(); ns, xs = sim.ctx.photon_log(n)
            assert t["nphotons_done"] == n, (rep, t["nphotons_done"])
            assert np.array_equal(ns, ns0) and np.array_equal(xs, xs0), rep
            for key in ("Jout", "scatt", "direc"):
                assert np.allclose(t[key], t0[key], rtol=1e-9, atol=1e-12 * max(1.0, float(np.abs(t0[key]).max()))), (rep, key)
            sim.ctx.close()
    finally:
        stop.set()
        for th in threads:
            th.join()
    assert not errors, errors
