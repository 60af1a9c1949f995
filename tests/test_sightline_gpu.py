"""Sight-line tau(nu) / N(HI) maps on the GPU (lart_gpu_sightline_tau through the C-ABI) against the CPU oracle, pixel by
pixel and frequency by frequency: far TAN observers over the Cartesian grid, the octree and the clump medium, HEALPix
observers inside the Cartesian grid and the octree.  Tolerance: the march is the same sequence of sums; the device differs
by FMA contraction and last-bit libm results only (rtol 1e-10 on tau, 1e-12 on the columns)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from lart_amd import capi
    assert capi.lib().lart_gpu_device_count() > 0, "no GPU visible: the HIP path cannot run (there is no fallback)"
    return capi


def pair(oracle_mod, kw):
    """the same namelist for the host mirror and for the oracle (observer lists are alpha1, alpha2, ... there)"""
    from lart_amd import host
    sim = host.Simulation(dict(kw))
    ok = {}
    for k, v in kw.items():
        if isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                ok["%s%d" % (k, i + 1)] = x
        else:
            ok[k] = v
    return sim, oracle_mod.Oracle(**ok)


def compare(sim, o, dust=False):
    tg, ng, td = sim.make_sightline_tau()
    to, no, do = o.sightline_tau()
    assert tg.shape == to.shape and ng.shape == no.shape
    assert np.array_equal(ng == 0.0, no == 0.0)                          # the same pixels see the medium
    assert np.array_equal(tg == 0.0, to == 0.0)
    assert (no > 0).sum() > 0.2 * no.size
    assert np.allclose(ng, no, rtol=1e-12, atol=0)
    assert np.allclose(tg, to, rtol=1e-10, atol=1e-300)
    if dust:
        assert do.max() > 0 and np.allclose(td, do, rtol=1e-12, atol=0)
    else:
        assert td.max() == 0.0 and do.max() == 0.0
    return tg, ng, td


def test_cartesian_far_observers(capi, oracle_mod):
    # two observers, Hubble flow, dust, 70 bins (one full chunk of 64 frequencies and a ragged one), non-square image
    kw = dict(temperature=1e4, taumax=1e4, nx=31, ny=31, nz=31, rmax=1.0, nxfreq=70, nxim=24, nyim=17, velocity_type="hubble", Vexp=60.0,
              DGR=1.0, alpha=[25.0, -110.0], beta=[60.0, 130.0], save_peeloff=True)
    sim, o = pair(oracle_mod, kw)
    tg, ng, td = compare(sim, o, dust=True)
    assert tg.shape == (70, 24, 17, 2)
    assert sim.ctx.sightline_time() > 0.0


def test_cartesian_rectangle_and_axis_aligned_rays(capi, oracle_mod):
    # observer on the z axis of a box: the central column of pixels has kx = 0 or ky = 0 exactly (odd image), rays cross faces and edges
    kw = dict(temperature=2e4, taumax=50.0, nx=8, ny=12, nz=20, xmax=0.4, ymax=0.6, zmax=1.0, geometry="rectangle", nxfreq=33, nxim=15, nyim=15,
              save_peeloff=True)
    sim, o = pair(oracle_mod, kw)
    tg, ng, _ = compare(sim, o)
    a = o.array("voigt_a")[0]
    H = oracle_mod.voigt(o.xfreq_axis(), a) / oracle_mod.voigt(0.0, a)
    assert np.allclose(tg[:, 7, 7, 0], 2 * 50.0 * H, rtol=1e-9)           # straight down the z axis: twice the pole depth


def test_static_sphere_known_answer_on_device(capi, oracle_mod):
    from lart_amd import host
    sim = host.Simulation(dict(temperature=1e4, taumax=1e3, rmax=1.0, nx=41, ny=41, nz=41, nxfreq=41, nxim=9, nyim=9, save_peeloff=True))
    tg, ng, td = sim.make_sightline_tau()
    x = sim.xfreq_axis() if hasattr(sim, "xfreq_axis") else (np.arange(41) + 0.5) * sim.grid.dxfreq + sim.grid.xfreq_min
    a = float(np.asarray(sim.grid.voigt_a).ravel()[sim.grid.voigt_a.size // 2])
    H = oracle_mod.voigt(x, a) / oracle_mod.voigt(0.0, a)
    assert np.allclose(tg[:, 4, 4, 0], 2e3 * H, rtol=1e-9) and np.isclose(ng[4, 4, 0], 2 * sim.par.N_gasmax, rtol=1e-12)
    assert ng[0, 0, 0] == 0.0 and tg[:, 0, 0, 0].max() == 0.0


def test_cartesian_inside_observers(capi, oracle_mod):
    kw = dict(temperature=1e4, taumax=1e3, nx=31, ny=31, nz=31, rmax=1.0, nxfreq=61, nside=4, obsx=[0.3, -0.2], obsy=[0.0, 0.45], obsz=[0.0, 0.1],
              velocity_type="hubble", Vexp=30.0)
    sim, o = pair(oracle_mod, kw)
    tg, ng, _ = compare(sim, o)
    assert tg.shape == (61, 192, 1, 2)


@pytest.fixture(scope="module")
def amr_file(tmp_path_factory):
    from lart_amd import amr
    fn = str(tmp_path_factory.mktemp("amr") / "sphere_small.dat")
    amr.write_generic_amr(fn, amr.make_amr_sphere(level_min=2, level_max=5))
    return fn


def test_octree_far_and_inside_observers(capi, oracle_mod, amr_file):
    kw = dict(use_amr_grid=True, amr_file=amr_file, temperature=1e4, taumax=1e3, rmax=1.0, nxfreq=61, velocity_type="hubble", Vexp=40.0,
              nxim=21, nyim=21, alpha=[0.0, 35.0], beta=[0.0, 70.0], save_peeloff=True)
    sim, o = pair(oracle_mod, kw)
    compare(sim, o)
    kw = dict(use_amr_grid=True, amr_file=amr_file, temperature=1e4, taumax=1e3, rmax=1.0, nxfreq=61, velocity_type="hubble", Vexp=40.0,
              nside=4, obsx=0.25, obsy=-0.1, obsz=0.3)
    sim, o = pair(oracle_mod, kw)
    compare(sim, o)


def test_clump_medium(capi, oracle_mod):
    kw = dict(use_clump_medium=True, rmax=1.0, clump_radius=0.03, clump_f_cov=4.0, clump_tau0=50.0, clump_sigma_v=30.0, temperature=1e4, nxfreq=90,
              velocity_min=-150.0, velocity_max=150.0, velocity_type="hubble", Vexp=80.0, nxim=20, nyim=20, alpha=15.0, beta=40.0, save_peeloff=True, iseed=5)
    sim, o = pair(oracle_mod, kw)
    tg, ng, _ = compare(sim, o)
    # mean number of clumps along a sight line through the centre region ~ f_cov-like: columns come in units of one clump chord
    assert ng.max() > 0


def test_overlapping_clumps(capi, oracle_mod):
    """the overlap forms of the two marches (raytrace_clump.f90:922-1001) through the device's streaming event walk: examples/clump_sphere/
    clump_overlap_peel.in has par%save_sightline_tau"""
    kw = dict(use_clump_medium=True, rmax=1.0, clump_radius=0.3, clump_f_cov=2.0, clump_tau0=1e3, temperature=1e4, clump_sigma_v=100.0, clump_allow_overlap=True,
              nxfreq=101, velocity_min=-200.0, velocity_max=200.0, nx=11, ny=11, nz=11, nxim=33, nyim=33, alpha=[0.0, 25.0], beta=[0.0, 60.0], save_peeloff=True,
              DGR=1.0, cext_dust=3e-18, iseed=5)
    sim, o = pair(oracle_mod, kw)
    assert sim.clumps.has_overlap and int(o.scalar("raytrace_variant")) == 7
    tg, ng, td = compare(sim, o, dust=True)
    assert ng.max() > 0 and td.max() > 0


def test_boundary_errors(capi):
    from lart_amd import host, capi as cp
    sim = host.Simulation(dict(temperature=1e4, taumax=10.0, nx=9, ny=9, nz=9, rmax=1.0))          # no observers
    with pytest.raises(host.LartInputError):
        sim.make_sightline_tau()
    with pytest.raises(cp.LartError) as e:
        sim.ctx.sightline_tau(sim.line.cross0)
    assert "observers" in str(e.value)
    sim = host.Simulation(dict(temperature=1e4, taumax=10.0, nx=9, ny=9, nz=9, rmax=1.0, nxim=4, nyim=4, save_peeloff=True))
    with pytest.raises(cp.LartError):
        sim.ctx.sightline_tau(0.0)
