"""Octree (AMR) medium on the CPU side: the oracle's restatement of octree_mod.f90 / grid_mod_amr.f90 / raytrace_amr.f90
against the reference's own known answers (examples/amr_sphere_generic/log_amr_1M.txt) and geometric invariants, and
the host mirror lart_amd/amr.py against the oracle, table for table.  No GPU."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def small_amr(tmp_path_factory):
    from lart_amd import amr
    d = amr.make_amr_sphere(level_min=2, level_max=5)
    fn = str(tmp_path_factory.mktemp("amr") / "sphere_small.dat")
    amr.write_generic_amr(fn, d)
    return fn, d


def host_grid(kw):
    from lart_amd import amr, host
    par = host.Params()
    for k, v in kw.items():
        host.set_param(par, k, v)
    par, line, _ = host.read_input(par)
    g, sg = amr.grid_create_amr(par, line, par.amr_file)
    amr.create_JPa_amr_mem(par, g)
    return par, line, g, sg


@pytest.mark.parametrize("geo", [1, -1, 3])
def test_host_tree_and_tables_equal_oracle(oracle_mod, small_amr, geo):
    fn, d = small_amr
    kw = dict(use_amr_grid=True, amr_file=fn, temperature=1e4, taumax=1e3, rmax=1.0, nxfreq=61,
              calc_J=True, calc_P=True, calc_Pnew=True, geometry_JPa=geo)
    o = oracle_mod.Oracle(**kw)
    par, line, g, sg = host_grid(kw)
    assert (g.ncells, g.nleaf) == (int(o.scalar("amr_ncells")), int(o.scalar("amr_nleaf"))) and g.nleaf == len(d["x"])
    for k in ("children", "neighbor", "ileaf", "icell_of_leaf"):          # same numbering as amr_build_tree: bit-exact
        assert np.array_equal(o.array("amr_" + k), np.asarray(getattr(g, k)).ravel(order="F")), k
    for k in ("cx", "cy", "cz", "ch"):
        assert np.array_equal(o.array("amr_" + k), getattr(g, k)), k
    for k in ("rhokap", "Dfreq", "voigt_a", "vfx"):
        assert np.allclose(o.array(k), getattr(g, k), rtol=1e-14, atol=0), k
    assert par.geometry_JPa == int(o.scalar("geometry_JPa")) == geo
    assert g.nbins == int(o.scalar("nbins"))
    assert np.allclose(o.array("ncount"), g.vol, rtol=1e-12, atol=1e-15)   # gas volume per profile bin
    assert sg.nxfreq == int(o.scalar("nxfreq")) and np.isclose(sg.xfreq_min, o.scalar("xfreq_min")) and np.isclose(sg.dxfreq, o.scalar("dxfreq"))
    assert np.isclose(sg.xcrit, o.scalar("xcrit"), rtol=1e-12)


def test_input_rules(small_amr):
    from lart_amd import host
    fn, _ = small_amr
    # setup.f90:438: the profile geometry is decided on the INPUT rmax, before grid_create_amr sets par%rmax
    par, _, g, _ = host_grid(dict(use_amr_grid=True, amr_file=fn, temperature=1e4, taumax=1e2, calc_P=True))
    assert par.geometry_JPa == 3 and g.nbins == g.nleaf and par.rmax == 1.0
    for bad in (dict(use_amr_grid=True), dict(use_amr_grid=True, amr_file=fn, amr_type="ramses")):
        p = host.Params()
        for k, v in dict(temperature=1e4, taumax=10.0, **bad).items():
            host.set_param(p, k, v)
        with pytest.raises(host.LartInputError):
            host.read_input(p)


def test_reference_log_known_answers(oracle_mod, tmp_path):
    """examples/amr_sphere_generic/log_amr_1M.txt: the data set of make_amr_sphere_data.py (levels 3..7 refined on the
    sphere surface) has nleaf = 178480, voigt_a = 4.7186E-04, N(HI)_pole = 1.6954E+17 at tau_pole = 1e4"""
    from lart_amd import amr
    d = amr.make_amr_sphere()
    assert len(d["x"]) == 178480
    fn = str(tmp_path / "sphere.dat"); amr.write_generic_amr(fn, d)
    o = oracle_mod.Oracle(use_amr_grid=True, amr_file=fn, temperature=1e4, taumax=1e4, rmax=1.0, nxfreq=121)
    assert int(o.scalar("amr_nleaf")) == 178480 and o.scalar("amr_boxlen") == 2.0
    assert abs(o.array("voigt_a")[0] / 4.7186e-4 - 1) < 2e-5
    assert abs(o.scalar("N_gasmax") / 1.6954e17 - 1) < 5e-5
    assert o.scalar("taumax") == 1e4


def test_octree_geometry_invariants(oracle_mod, small_amr):
    """raytrace_to_edge_amr from the centre along any direction through a uniform sphere = tau0 * H(x)/H(0) up to the
    staircase of the refined surface; to_dist + remaining edge = edge; to_tau lands where to_dist says"""
    fn, _ = small_amr
    o = oracle_mod.Oracle(use_amr_grid=True, amr_file=fn, temperature=1e4, taumax=1e2, rmax=1.0, nxfreq=61)
    rng = np.random.default_rng(3)
    for _ in range(50):
        k = rng.normal(size=3); k /= np.linalg.norm(k)
        pos = rng.uniform(-0.5, 0.5, 3)
        il = int(o.amr_find_leaf(*pos))
        assert il >= 1
        r = np.array([*pos, *k, rng.uniform(-4, 4), 1.0]); ic = np.array([il, 1, 1, 1], dtype=np.int32)
        tau_edge = o.raytrace_to_edge(3, r, ic)
        dist = rng.uniform(0.05, 0.4)
        t1 = o.raytrace_to_dist(r, ic, dist)
        assert 0 <= t1 <= tau_edge * (1 + 1e-12)
        r2, ic2 = o.raytrace_to_tau(3, r.copy(), ic.copy(), t1)
        if t1 > 0:
            assert np.allclose(r2[:3], pos + dist * k, rtol=0, atol=1e-9)         # the same path, inverted
        rest = o.raytrace_to_edge(3, r2, ic2)
        assert np.isclose(t1 + rest, tau_edge, rtol=1e-9, atol=1e-12)
    # static uniform sphere: centre-to-edge depth along the axes is exactly tau0 at line centre
    r = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 1.0]); ic = np.array([int(o.amr_find_leaf(0, 0, 0)), 1, 1, 1], dtype=np.int32)
    assert np.isclose(o.raytrace_to_edge(3, r, ic), 1e2, rtol=1e-12)


@pytest.mark.slow
def test_whole_path_known_answer(oracle_mod, small_amr):
    """log_amr_1M.txt: <N_scatt> = 2.8225E+04 for the tau0 = 1e4, T = 1e4 K static sphere (point source).  The oracle on
    the coarser octree (levels 2..5) must agree within the Monte-Carlo error + the ~1% staircase effect."""
    fn, _ = small_amr
    o = oracle_mod.Oracle(use_amr_grid=True, amr_file=fn, temperature=1e4, taumax=1e4, rmax=1.0, nxfreq=121)
    n = 400
    o.run(n, seed=5)
    ns, _ = o.photon_log()
    assert abs(ns.mean() - 2.8225e4) < 0.02 * 2.8225e4 + 4 * ns.std() / np.sqrt(n)


def test_periodic_octree_slab(oracle_mod, tmp_path):
    """par%xy_periodic on the octree (raytrace_amr.f90:176-187, :309-319): a side face of the root box wraps to the opposite one and the
    leaf is looked up from the root.  Uniform gas on leaves of two sizes (the wrap lands on another refinement level): the physics is the
    plane-parallel slab, so <N_scatt> and the emergent spectrum equal those of the Cartesian nx = ny = 1 slab of the same tau0; without
    the wrap the same box leaks through its sides and scatters less."""
    from lart_amd import amr
    fn = str(tmp_path / "slab.dat")
    amr.write_generic_amr(fn, amr.make_amr_slab(level_min=2, level_max=3))
    n = 20000
    kw = dict(use_amr_grid=True, amr_file=fn, temperature=1e4, taumax=3e2, nxfreq=61, xfreq_min=-12.0, xfreq_max=12.0)
    a = oracle_mod.Oracle(xy_periodic=True, **kw); a.run(n, seed=5)
    assert int(a.scalar("amr_nleaf")) == 48 + 16 * 8 and a.scalar("taumax") == 3e2
    c = oracle_mod.Oracle(temperature=1e4, taumax=3e2, xy_periodic=True, nx=1, ny=1, nz=64, nxfreq=61, xfreq_min=-12.0, xfreq_max=12.0); c.run(n, seed=6)
    b = oracle_mod.Oracle(**kw); b.run(n, seed=5)
    na, nc, nb = a.photon_log()[0], c.photon_log()[0], b.photon_log()[0]
    assert abs(na.mean() - nc.mean()) < 4.0 * np.hypot(na.std(), nc.std()) / np.sqrt(n), (na.mean(), nc.mean())
    assert nb.mean() < 0.85 * na.mean()
    ja, jc = a.array("Jout"), c.array("Jout")
    assert abs(ja.sum() - n) < 1e-9 * n and abs(jc.sum() - n) < 1e-9 * n   # nothing is lost through the sides (every packet leaves through z; esc + (1 - esc) is 1 up to rounding)
    chi2 = np.sum((ja - jc) ** 2 / np.maximum(ja + jc, 1.0)) / np.count_nonzero(ja + jc)
    assert chi2 < 1.6, chi2
    assert np.allclose(a.normalized_Jout(), ja / (n * a.scalar("dxfreq") * 2.0 * np.pi * 2.0))      # two faces, no area (output_sum_rect.f90:802-804)
    # the plane profile of the scattering rate (geometry_JPa = -1 by default for a periodic box): symmetric in z, same on both refinement levels
    p = oracle_mod.Oracle(xy_periodic=True, calc_P=True, rmax=1.0, **kw); p.run(n, seed=5)
    assert int(p.scalar("geometry_JPa")) == -1 and int(p.scalar("nbins")) == 8
    P = p.array("P1") / p.array("ncount")
    assert np.allclose(P, P[::-1], rtol=0.08) and P[3] > P[0]


def test_mirrored_octree_quadrant(oracle_mod, tmp_path):
    """par%xy_symmetry on the octree (raytrace_amr.f90:193-197): the -x and -y faces of the root box reflect, +x, +y and both z faces let
    the packet out.  The box [0, L]^2 x [-L/2.., ..] cannot be written as a generic file centred on the origin, so the check is on the
    boundary rule itself: a uniform box with mirrors on two sides scatters more than the open box and less than the periodic slab."""
    from lart_amd import amr
    fn = str(tmp_path / "box.dat")
    amr.write_generic_amr(fn, amr.make_amr_slab(level_min=2, level_max=3))
    n = 10000
    kw = dict(use_amr_grid=True, amr_file=fn, temperature=1e4, taumax=3e2, nxfreq=61, xfreq_min=-12.0, xfreq_max=12.0)
    means = []
    for extra in (dict(), dict(xy_symmetry=True), dict(xy_periodic=True)):
        o = oracle_mod.Oracle(**kw, **extra); o.run(n, seed=5)
        assert o.array("Jout").sum() == n
        means.append(o.photon_log()[0].mean())
    assert means[0] < 0.95 * means[1] and means[1] < 0.95 * means[2], means
