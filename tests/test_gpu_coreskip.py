"""Core-skipping (north star: "Voigt-profile frequency redistribution with cell-by-cell core-skip") on the GPU against the CPU oracle:
car_xcrit_local (grid_mod_car.f90:1598-1629), amr_xcrit_local (octree_mod.f90:248-278), the truncated perpendicular velocity
(scattering_car.f90:397-411 with Stokes, :733-752 without; scattering_amr.f90:254-270, :465-480) and the global recipe
par%core_skip_global (grid_mod_car.f90:1186-1219, setup.f90:237).  Whole histories on the same (seed, id) streams."""
import numpy as np
import pytest
from conftest import rms_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from lart_amd import capi
    assert capi.lib().lart_gpu_device_count() > 0, "no GPU visible: the HIP path cannot run (there is no fallback)"
    return capi


@pytest.fixture(scope="module")
def amr_file(tmp_path_factory):
    from lart_amd import amr
    p = tmp_path_factory.mktemp("amr") / "sphere_small.dat"
    amr.write_generic_amr(str(p), amr.make_amr_sphere(level_min=2, level_max=3, T=10.0))      # leaves at par%temperature: the frequency axis is in their Doppler units
    return str(p)


def make(kw):
    from lart_amd import host
    return host.Simulation(dict(kw), 0)


def histories(capi, oracle_mod, kw, n, seed):
    sim = make(dict(kw, iseed=seed)); sim.ctx.enable_photon_log(True); sim.run_simulation(n)
    o = oracle_mod.Oracle(**(dict(kw, iseed=seed) if kw.get("use_clump_medium") else kw)); o.run(n, seed=seed)      # (clumps: the same placement stream)
    ng, xg = sim.ctx.photon_log(n); no, xo = o.photon_log()
    same = np.isclose(ng, no, rtol=1e-12, atol=0) & np.isclose(xg, xo, rtol=1e-7, atol=1e-9)
    return sim, o, ng, no, same


def flipped_rms(same, n, nbins_eff=40):
    """bound on the spectrum rms-diff of two runs that share all but the fraction f of their histories: the f n flipped photons land
    in other bins, ~ sqrt(2 f nb / n) of the peak for nb populated bins (x2 slack, + 1 % of f for the isclose window)"""
    f = 1.0 - same.mean() + 0.01
    return 2.0 * np.sqrt(2.0 * f * nbins_eff / n)


CAR = dict(temperature=10.0, taumax=2e5, nx=5, ny=5, nz=5, rmax=1.0, nxfreq=121)      # coarse cells: a tau_cell dl/dx ~ 6e2, xcrit up to ~1.7
# (a device run of a few hundred photons lasts as long as its longest history: ~9 us per scattering, hence the moderate optical depth)


@pytest.mark.parametrize("extra,n", [(dict(xyz_symmetry=True, core_skip=True), 250),
                                     (dict(core_skip=True, use_stokes=True), 140),
                                     (dict(core_skip=True, recoil=True, velocity_type="hubble", Vexp=20.0), 600),
                                     (dict(xyz_symmetry=True, core_skip_global=True), 1000),
                                     (dict(core_skip_global=True, use_stokes=True), 500)],
                         ids=["cell", "cell_stokes", "cell_recoil_outflow", "global", "global_stokes"])
def test_cartesian_core_skip_histories(capi, oracle_mod, extra, n):
    """tau0 = 2e5, T = 10 K on a coarse grid (xcrit > 1 away from the cell faces): photon by photon against the oracle, and the
    skipping really happens: the same photon ids without it have different histories and scatter several times more often"""
    kw = dict(CAR, **extra)
    sim, o, ng, no, same = histories(capi, oracle_mod, kw, n, 31)
    assert sim.cpar.core_skip == 1 and (sim.cpar.core_skip_global == 1) == bool(extra.get("core_skip_global"))
    if extra.get("core_skip_global"):
        assert sim.cpar.xcrit > 0.0 and np.isclose(sim.cpar.xcrit, o.scalar("xcrit"), rtol=1e-12)
    assert same.mean() > 0.99, same.mean()
    t = sim.ctx.tallies()
    assert rms_diff(t["Jout"], o.array("Jout")) < flipped_rms(same, n)
    assert abs(ng.mean() / no.mean() - 1.0) < 0.03
    m = min(300, n)
    plain = make(dict({k: v for k, v in kw.items() if not k.startswith("core_skip")}, iseed=31)); plain.ctx.enable_photon_log(True); plain.run_simulation(m)
    npl = plain.ctx.photon_log(m)[0]
    assert np.mean(npl != ng[:m]) > 0.9                     # same ids, same streams: the truncated-tail branch changed the histories
    assert npl.mean() > (2.0 if extra.get("core_skip_global") else 1.3) * ng.mean()


@pytest.mark.parametrize("extra,n", [(dict(core_skip=True), 200), (dict(core_skip=True, use_stokes=True), 170), (dict(core_skip_global=True), 1400)],
                         ids=["cell", "cell_stokes", "global"])
def test_octree_core_skip_histories(capi, oracle_mod, amr_file, extra, n):
    """amr_xcrit_local: the distance to the nearest face of the LEAF sets xcrit (octree_mod.f90:248-278)"""
    kw = dict(use_amr_grid=True, amr_file=amr_file, temperature=10.0, taumax=1e5, rmax=1.0, nxfreq=121, **extra)
    sim, o, ng, no, same = histories(capi, oracle_mod, kw, n, 33)
    assert same.mean() > 0.99, same.mean()
    assert rms_diff(sim.ctx.tallies()["Jout"], o.array("Jout")) < flipped_rms(same, n)
    m = min(300, n)
    plain = make(dict({k: v for k, v in kw.items() if not k.startswith("core_skip")}, iseed=33)); plain.ctx.enable_photon_log(True); plain.run_simulation(m)
    npl = plain.ctx.photon_log(m)[0]
    assert np.mean(npl != ng[:m]) > 0.5                     # same ids, same streams: most histories met a cell with xcrit > |x|
    if extra.get("core_skip_global"):
        assert npl.mean() > 2.0 * ng.mean()


def test_clump_core_skip_stress(capi, oracle_mod):
    """C4-like stress: clumps with tau_cl = 1e3, f_cov = 5.  Cell-by-cell core-skip reads grid%rhokap,
    which grid_create_clump zeroes (grid_mod_clump.f90:101), so par%core_skip alone changes nothing in the reference -- reproduced:
    identical histories with and without it; par%core_skip_global does skip, and matches the oracle photon by photon."""
    base = dict(use_clump_medium=True, rmax=1.0, clump_radius=0.02, clump_f_cov=5.0, clump_tau0=1e3, temperature=1e4, nxfreq=301,
                velocity_min=-150.0, velocity_max=150.0, nx=11, ny=11, nz=11)
    n = 4000
    a = make(dict(base, core_skip=True, iseed=5)); a.ctx.enable_photon_log(True); a.run_simulation(n)
    b = make(dict(base, iseed=5)); b.ctx.enable_photon_log(True); b.run_simulation(n)
    assert np.array_equal(a.ctx.photon_log(n)[0], b.ctx.photon_log(n)[0])
    kw = dict(base, core_skip_global=True)
    sim, o, ng, no, same = histories(capi, oracle_mod, kw, n, 5)
    assert sim.cpar.xcrit > 0.0 and np.isclose(sim.cpar.xcrit, o.scalar("xcrit"), rtol=1e-12)
    assert same.mean() > 0.99, same.mean()
    assert np.mean(ng != b.ctx.photon_log(n)[0]) > 0.3       # grid%xcrit of the clump set-up is small (a tau_homo / ... < 1 per cell): few events skip, many histories change
    assert rms_diff(sim.ctx.tallies()["Jout"], o.array("Jout")) < flipped_rms(same, n)
