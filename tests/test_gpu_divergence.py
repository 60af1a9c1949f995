"""Divergence audit of the photon-by-photon parity tests.

Device and oracle run the same counter-based streams (seed, photon id), so whole histories are identical -- except where a last-bit difference
(FMA contraction, libm) flips one data-dependent comparison, after which the two histories consume different draws and have nothing in common
any more.  The photon-by-photon tests used to forgive 3-5 % of mismatching photons without looking at them; a real bug that hits a rare branch
would hide in that allowance.  Here every photon whose logged scattering count or exit frequency differs is examined:

  * both sides record the state after the birth and after every event (lart_gpu_test_trace / orc_trace: draws consumed, frequency, position,
    cell or owner clump, direction, kind of event);
  * the first record in which the two histories differ DISCRETELY (draw counter, cell, kind) is located;
  * the oracle's record of THAT event carries the smallest margin of all comparisons the event made (oracle/lart_oracle.c, mrg_note: relative
    distance of the two compared numbers, divided by the condition number of the acceptance window of the wing sampler, a difference of two
    arctangents).  The divergence is explained when that margin is at rounding level (MARGIN) -- see audit() for the allowance in chaotic media.

What the audit found (MI355X, 1e5 photons per case, profiles/r03_divergence_audit.txt): in every Cartesian and octree case (12 + 3 configurations:
plain, octant, slab, quadrant, periodic box, dust HG / Mueller, Stokes, recoil, outflows, doublet, H + D, He I triplet, core-skip) there is NOT ONE
discrete divergence; the 1-17 photons per 1e5 whose logs differ have identical histories and differ in the last digits of the weight
1 - exp(-tau0) of the forced first scattering (a packet born in the far wing has a tiny tau0; the subtraction cancels on both sides).  In the
clump media 0.1-0.7 % of the histories do part: flights between spheres are a Sinai billiard, the positions of two arithmetically equivalent
marches drift apart geometrically from rounding level and finally flip a hit / miss (comparisons 46-48, 53, 54); the drift never jumps.

Any mismatching photon that is not explained in this way fails the test.
References: run_simulation_mod.f90:150-202 (loop), random_mt.f90:2562-2696 (rejection sampler), raytrace_car.f90:1485-1596,
raytrace_amr.f90:77-259, raytrace_clump.f90:83-201, 668-788, scattering_car.f90:14-120."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MARGIN = 1e-9          # relative; the flips seen are at 1e-16 ... 1e-12 (see the printed histogram)
W = 8


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from lart_amd import capi
    assert capi.lib().lart_gpu_device_count() > 0, "no GPU visible: the HIP path cannot run (there is no fallback)"
    return capi


def state_distance(dev, orc, length):
    """relative distance of the floating-point state of two records (position in units of `length`: the clump radius or 1, frequency, direction);
    the frequency of a packet that was absorbed is not defined (the oracle logs 0) and is left out"""
    d, o = dev, orc[:, :W]
    dpos = np.abs(d[:, 2:5] - o[:, 2:5]).max(axis=1) / length
    skipx = (d[:, 7] == 2) & ((o[:, 1] == 0.0) | (d[:, 1] == 0.0))
    dx = np.where(skipx, 0.0, np.abs(d[:, 1] - o[:, 1]) / np.maximum(1.0, np.abs(o[:, 1])))
    return np.maximum(np.maximum(dpos, dx), np.abs(d[:, 6] - o[:, 6]))


def first_difference(dev, nd, orc, no, length=1.0):
    """(i, drift): index of the first record in which the two histories differ DISCRETELY -- draws consumed, cell / owner clump, kind of event --
    or, failing that, by more than 1e-7 in the floating-point state of a medium whose marches are not chaotic; drift = the state distance of the
    record before it (the allowance a flipped comparison gets, see audit).  i = None: one history is a prefix of the other and both end there"""
    m = min(nd, no, len(dev), len(orc))
    d, o = dev[:m], orc[:m]
    delta = state_distance(d, o, length)
    # (the draw counter of a packet that has left is not compared: the overlap forms take the owner-clump draw of sample_owner_clump before the
    #  march -- transport_kernel.hpp, fly() -- so a packet that escapes instead has consumed one draw it never uses)
    bad = ((d[:, 0] != o[:, 0]) & ~((d[:, 7] == 2) & (o[:, 7] == 2))) | (d[:, 5] != o[:, 5]) | (d[:, 7] != o[:, 7])
    if length == 1.0:
        bad |= delta > 1e-7
    i = np.nonzero(bad)[0]
    if len(i):
        i = int(i[0]); return i, (float(delta[i - 1]) if i > 0 else 0.0), float(delta[:i].max()) if i > 0 else 0.0
    return (None, 0.0, float(delta.max()) if m else 0.0) if nd == no else (m, float(delta[m - 1]) if m else 0.0, float(delta.max()) if m else 0.0)


def drift_jumps(dev, orc, upto, length, floor=1e-11, growth=1e3, cond=1e-3):
    """events before record `upto` in which the distance of the two states grows by more than `growth` x (from above `floor`) although every
    comparison of the event was farther than `cond` from its threshold: a drift may only grow geometrically through ill-conditioned encounters
    (a ray grazing a sphere: comparison 47), it must not jump -- a jump is two different formulas, not rounding"""
    m = min(upto, len(dev), len(orc))
    if m < 2:
        return []
    delta = state_distance(dev[:m], orc[:m], length)
    prev = np.maximum(delta[:-1], floor)
    j = np.nonzero((delta[1:] > growth * prev) & (orc[1:m, 8] > cond))[0] + 1
    return [(int(q), float(delta[q - 1]), float(delta[q]), float(orc[q, 8]), int(orc[q, 9])) for q in j]


def resync_agrees(sim, o, seed, pid, i, otr, length):
    """the one-step re-synchronised check: the oracle's full state after event i - 1 of photon `pid` (orc_trace_state) goes through ONE event round on the
    device (lart_gpu_test_event: optical-depth draw, flight, scattering -- the statements of the transport kernel's loop); the result has to be the oracle's
    record i: same draws consumed, same owner clump / cell, same kind of event, and the same frequency, position and direction to 1e-9 (positions in units
    of `length`).  True means: given the same input the device takes exactly the oracle's step, so the two histories parted because their inputs had
    drifted apart (a sphere is a dispersing billiard), not because the step functions differ."""
    st, _ = o.trace_state(seed, pid, i - 1)
    dev = sim.ctx.test_event(st[None, :], seed, [pid], [1 if i == 1 else 0])[0]
    want = otr[i]
    kind_ok = dev[26] == want[7]
    cell = dev[23] if sim.par.use_clump_medium else dev[20] + 2097152.0 * dev[21] + 4398046511104.0 * dev[22]
    draws_ok = dev[19] == want[0] or (dev[26] == 2 and want[7] == 2)
    dpos = np.abs(dev[0:3] - want[2:5]).max() / length
    dx = 0.0 if (want[7] == 2 and want[1] == 0.0) else abs(dev[6] - want[1]) / max(1.0, abs(want[1]))
    return bool(kind_ok and cell == want[5] and draws_ok and dpos < 1e-9 and dx < 1e-9 and abs(dev[5] - want[6]) < 1e-9)


def audit(sim, o, n, seed, label, budget_bytes=1 << 30, length=1.0):
    """run n photons on both sides, then explain every mismatching photon; returns (fraction of identical histories, unexplained, report).

    A mismatching photon is explained when
      (a) its two histories are discretely identical to the end (same draws, same cells, same events): the logs differ by the rounding of the
          weight 1 - exp(-tau0) of the forced first scattering or by the drift below; or
      (b) the first discrete difference happens in an event whose smallest comparison margin (oracle, mrg_note) is below MARGIN: a rounding flip; or
      (c) clump media only (`length` = the clump radius): flights between spheres are a Sinai billiard, every ray-sphere encounter multiplies a position
          difference, so two arithmetically equivalent marches drift apart geometrically from rounding level (measured: 1e-15 -> 1e-8 of the box in 3000
          events) until the drift flips a hit / miss decision whose own margin is NOT small.  Round 3 forgave such a flip when its margin was below
          1e3 x the drift, which at a drift of 0.1 forgives anything.  Now the step is repeated: the oracle's state of the event
          before (orc_trace_state) goes through one event round on the device (lart_gpu_test_event) and must come out as the oracle's next record
          (resync_agrees).  A step function that differs from the oracle's fails here whatever the drift was."""
    sim.ctx.enable_photon_log(True)
    sim.run_simulation(n, seed=seed)
    ng, xg = sim.ctx.photon_log(n)
    o.run(n, seed=seed); no, xo = o.photon_log()
    same = np.isclose(ng, no, rtol=1e-12, atol=0) & (np.isclose(xg, xo, rtol=1e-7, atol=1e-9) | (xo == 0))      # (an absorbed packet has no exit frequency: the oracle logs 0)
    bad = np.nonzero(~same)[0]
    unexplained, ids, margins, drifts, n_same_history, n_resync = [], [], [], [], 0, 0
    if len(bad):
        est = (np.minimum(ng[bad], no[bad]) * 1.05 + 64).astype(np.int64)
        order = np.argsort(est)
        bad, est = bad[order], est[order]
        i0 = 0
        while i0 < len(bad):
            cap = int(est[i0]); i1 = i0
            while i1 < len(bad) and (i1 - i0 + 1) * max(cap, int(est[i1])) * W * 8 <= budget_bytes:
                cap = max(cap, int(est[i1])); i1 += 1
            i1 = max(i1, i0 + 1); cap = int(max(cap, est[i1 - 1]))
            sel = bad[i0:i1]
            sim.ctx.zero_tallies()
            tr, nrec = sim.ctx.test_trace(n, sel, cap, seed)
            for j, k in enumerate(sel):
                otr, on = o.trace(seed, int(k) + 1, cap)
                i, drift, drift_max = first_difference(tr[j], int(nrec[j]), otr, on, length)
                if i is None:                                   # (a): the same history; the logged weight sum / exit frequency differ by rounding or drift only
                    n_same_history += 1
                    # (the weight of the forced first scattering is 1 - exp(-tau0), run_simulation_mod.f90:169: for a packet born in the far wing tau0 is
                    #  tiny and the subtraction loses log10(1 / tau0) digits on both sides alike; every later nscatt increment carries that weight)
                    if not (np.isclose(ng[k], no[k], rtol=1e-6, atol=0) and (xo[k] == 0 or np.isclose(xg[k], xo[k], rtol=1e-6 + 10 * drift_max, atol=1e-9))):
                        unexplained.append((int(k), -1, drift_max, 0))
                    continue
                if i >= len(otr) or i >= cap:
                    unexplained.append((int(k), -2, 0.0, 0)); continue       # the difference lies beyond what was recorded
                mg, mid = float(otr[i, 8]), int(otr[i, 9])
                ids.append(mid); margins.append(mg); drifts.append(drift)
                if mg < MARGIN:                                 # the oracle itself sat at rounding distance from a decision in this event
                    pass
                elif length != 1.0 and i >= 1 and resync_agrees(sim, o, seed, int(k) + 1, i, otr, length):
                    n_resync += 1                               # the two histories had drifted apart; from the SAME state the device takes the oracle's step
                else:
                    unexplained.append((int(k), i, mg, mid, drift))
                jumps = drift_jumps(tr[j], otr, i, length)
                if jumps:
                    unexplained.append((int(k), "drift jump", jumps[:3]))
            i0 = i1
    rep = "%s: %d photons, %.3f %% identical logs, %d mismatching of which %d with identical histories (weight rounding) and %d whose step the device repeats from the oracle's state; flipped comparisons by id %s; margins max %.2e median %.2e; drift before the flip max %.2e" % (
        label, n, 100.0 * same.mean(), len(bad), n_same_history, n_resync, dict(zip(*[a.tolist() for a in np.unique(ids, return_counts=True)])) if ids else {},
        max(margins) if margins else 0.0, float(np.median(margins)) if margins else 0.0, max(drifts) if drifts else 0.0)
    print(rep)
    return same.mean(), unexplained, rep


def make(kw):
    from lart_amd import host
    return host.Simulation(dict(kw))


def oracle_of(oracle_mod, kw):
    return oracle_mod.Oracle(**{k: v for k, v in kw.items() if k != "iseed"})


CAR = [
    ("plain", dict(temperature=1e4, taumax=1e2, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=81), 100000),
    ("plain_stokes_recoil_outflow", dict(temperature=1e4, taumax=1e2, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=81, use_stokes=True, recoil=True,
                                         velocity_type="hubble", Vexp=40.0, source_geometry="uniform_sphere"), 100000),
    ("octant_tau3", dict(temperature=1e4, taumax=1e3, xyz_symmetry=True, nx=41, ny=41, nz=41, rmax=1.0, nxfreq=121), 100000),
    ("octant_T10_tau4", dict(temperature=10.0, taumax=1e4, xyz_symmetry=True, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=121), 20000),
    ("slab_stokes", dict(temperature=1e2, taumax=1e2, nx=1, ny=1, nz=51, xmax=1.0, ymax=1.0, zmax=1.0, xy_periodic=True, use_stokes=True, save_Jin=True,
                         nxfreq=81), 100000),
    ("quadrant_outflow", dict(temperature=1e4, taumax=3e1, nx=20, ny=14, nz=31, rmax=1.0, xy_symmetry=True, velocity_type="hubble", Vexp=30.0, nxfreq=81), 100000),
    ("periodic_box_recoil", dict(temperature=1e4, taumax=2e1, nx=6, ny=4, nz=21, xmax=0.5, ymax=0.3, zmax=1.0, geometry="rectangle", xy_periodic=True,
                                 recoil=True, nxfreq=81), 100000),
    ("shearing_box", dict(temperature=1e4, taumax=2e1, nx=6, ny=4, nz=21, xmax=0.5, ymax=0.3, zmax=1.0, geometry="rectangle", xy_periodic=True, Omega=2.5, nxfreq=81), 100000),
    ("dust_hg_reduced_weight", dict(temperature=1e4, taumax=1e2, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=81, DGR=1.0, cext_dust=2e-19, use_reduced_wgt=True), 100000),
    ("fine_structure_stokes", dict(temperature=3e1, taumax=3e1, xyz_symmetry=True, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=121, fine_structure=True, use_stokes=True,
                                   velocity_type="hubble", Vexp=5.0), 100000),
    ("deuterium_recoil", dict(temperature=1e4, taumax=3e1, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=121, include_deuterium=True, D_to_H_ratio=0.05, recoil=True), 100000),
    ("HeI_triplet_stokes", dict(temperature=1e4, taumax=2e1, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=161, line_id="HeI_10833", velocity_min=-200.0,
                                velocity_max=200.0, use_stokes=True, velocity_type="hubble", Vexp=20.0), 100000),
]


@pytest.mark.parametrize("label,kw,n", CAR, ids=[c[0] for c in CAR])
def test_cartesian_divergences_are_rounding_flips(capi, oracle_mod, label, kw, n):
    sim = make(dict(kw, iseed=5)); o = oracle_of(oracle_mod, kw)
    frac, unexplained, rep = audit(sim, o, n, 5, label)
    assert not unexplained, (rep, unexplained[:10])
    assert frac > 0.9, rep


def test_dust_mueller_stokes_divergences(capi, oracle_mod, mueller_file):
    kw = dict(temperature=1e4, taumax=3e1, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=81, use_stokes=True, DGR=1.0, scatt_mat_file=mueller_file, cext_dust=5e-19,
              velocity_type="hubble", Vexp=20.0)
    sim = make(dict(kw, iseed=6)); o = oracle_of(oracle_mod, kw)
    frac, unexplained, rep = audit(sim, o, 100000, 6, "dust_mueller_stokes")
    assert not unexplained, (rep, unexplained[:10])


@pytest.fixture(scope="module")
def amr_file(tmp_path_factory):
    from lart_amd import amr
    fn = str(tmp_path_factory.mktemp("amr") / "sphere_small.dat")
    amr.write_generic_amr(fn, amr.make_amr_sphere(level_min=2, level_max=5))
    return fn


@pytest.mark.parametrize("extra,n", [(dict(taumax=1e2), 100000), (dict(taumax=3e1, velocity_type="hubble", Vexp=40.0, use_stokes=True, recoil=True), 100000),
                                     (dict(taumax=1e5, temperature=10.0, core_skip=True), 1000)], ids=["static", "outflow_stokes", "core_skip"])
def test_octree_divergences_are_rounding_flips(capi, oracle_mod, amr_file, extra, n):
    kw = dict(dict(use_amr_grid=True, amr_file=amr_file, temperature=1e4, rmax=1.0, nxfreq=81), **extra)
    sim = make(dict(kw, iseed=7)); o = oracle_of(oracle_mod, kw)
    frac, unexplained, rep = audit(sim, o, n, 7, "octree " + str(extra))
    assert not unexplained, (rep, unexplained[:10])


CLUMP = dict(use_clump_medium=True, rmax=1.0, clump_radius=0.01, clump_f_cov=5.0, clump_tau0=1e2, temperature=1e4, nxfreq=301,
             velocity_min=-100.0, velocity_max=100.0, nx=11, ny=11, nz=11, iseed=7)
OVERLAP = dict(use_clump_medium=True, rmax=1.0, clump_radius=0.3, clump_f_cov=2.0, clump_tau0=30.0, temperature=1e4, clump_allow_overlap=True,
               nxfreq=101, velocity_min=-200.0, velocity_max=200.0, nx=11, ny=11, nz=11, iseed=7)


@pytest.mark.parametrize("base,extra,n", [(CLUMP, dict(), 100000), (CLUMP, dict(clump_sigma_v=30.0, velocity_type="hubble", Vexp=80.0), 100000),
                                          (CLUMP, dict(use_stokes=True, clump_tau0=30.0, recoil=True), 100000),
                                          (OVERLAP, dict(clump_sigma_v=100.0), 100000), (OVERLAP, dict(clump_sigma_v=20.0, DGR=1.0, cext_dust=3e-18), 50000),
                                          (OVERLAP, dict(clump_radius=0.05, clump_f_cov=6.0, clump_sigma_v=30.0), 50000)],
                         ids=["clumps", "clumps_moving", "clumps_stokes", "overlap_sigv", "overlap_dust", "overlap_many"])
def test_clump_divergences_are_rounding_flips(capi, oracle_mod, base, extra, n):
    kw = dict(base, **extra)                               # (the oracle gets par%iseed too: it seeds the clump placement)
    sim = make(kw); o = oracle_mod.Oracle(**kw)
    frac, unexplained, rep = audit(sim, o, n, 11, "clump " + str(extra), length=float(kw["clump_radius"]))
    assert not unexplained, (rep, unexplained[:10])


def test_core_skip_divergences(capi, oracle_mod):
    kw = dict(temperature=10.0, taumax=2e5, nx=5, ny=5, nz=5, rmax=1.0, nxfreq=121, xyz_symmetry=True, core_skip=True)
    sim = make(dict(kw, iseed=31)); o = oracle_of(oracle_mod, kw)
    frac, unexplained, rep = audit(sim, o, 2000, 31, "core_skip")
    assert not unexplained, (rep, unexplained[:10])
