"""Sight-line tau(nu) / N(HI) maps (par%save_sightline_tau) on the CPU side: the oracle's restatement of the marches and of
pix2vec against vectors from the reference's own compiled modules, and the map driver against geometric known answers."""
import numpy as np
import pytest


def test_sightline_marches_vs_reference(oracle_mod, golden):
    """raytrace_to_{edge,dist}_car_tau_gas / _column (raytrace_car.f90:197-408, :1236-1423), vectors from oracle/_ref"""
    z = golden("sightline_car.npz")
    o = oracle_mod.Oracle(str(z["params"]))
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD"):
        o.set_array(k, z[k])
    for r, ic, d, te, td, ce, cd in zip(z["r_in"], z["ic_in"], z["dist"], z["tau_gas_edge"], z["tau_gas_dist"], z["column_edge"], z["column_dist"]):
        assert np.isclose(o.raytrace_tau_gas(r, ic), te, rtol=1e-10, atol=1e-300)          # the Voigt-table tolerance
        assert np.isclose(o.raytrace_tau_gas(r, ic, float(d)), td, rtol=1e-10, atol=1e-300)
        assert np.allclose(o.raytrace_column(r, ic), ce, rtol=1e-14, atol=0)                # no Voigt profile in it
        assert np.allclose(o.raytrace_column(r, ic, float(d)), cd, rtol=1e-14, atol=0)


def test_pix2vec_vs_reference(oracle_mod, golden):
    z = golden("pix2vec.npz")
    for ns in (1, 2, 4, 16, 64):
        ref = z["nside%d" % ns]
        got = np.array([oracle_mod.Oracle.pix2vec(ns, ip) for ip in range(1, 12 * ns * ns + 1)])
        assert np.allclose(got, ref, rtol=0, atol=2e-16)
        L = oracle_mod.Oracle.lib()                                                       # pix2vec and vec2pix are inverse maps
        import ctypes
        L.orc_vec2pix.restype = ctypes.c_int; L.orc_vec2pix.argtypes = [ctypes.c_int] + [ctypes.c_double] * 3
        assert [L.orc_vec2pix(ns, *map(float, v)) for v in got] == list(range(1, 12 * ns * ns + 1))


def test_uniform_sphere_map_known_answers(oracle_mod):
    """static uniform sphere seen from far away: the central pixel's line-centre depth is 2 tau0 (the whole diameter),
    N(HI) = 2 N_pole, tau(x) follows H(x,a)/H(0,a); pixels outside the projected disc are empty"""
    o = oracle_mod.Oracle(temperature=1e4, taumax=1e3, rmax=1.0, nx=41, ny=41, nz=41, nxfreq=41, nxim=9, nyim=9, save_peeloff=1)
    tg, ng, td = o.sightline_tau()
    assert tg.shape == (41, 9, 9, 1) and ng.shape == (9, 9, 1)
    x = o.xfreq_axis(); a = o.array("voigt_a")[0]
    H = oracle_mod.voigt(x, a) / oracle_mod.voigt(0.0, a)
    assert np.allclose(tg[:, 4, 4, 0], 2e3 * H, rtol=1e-9)
    assert np.isclose(ng[4, 4, 0], 2 * o.scalar("N_gasmax"), rtol=1e-12)
    assert tg[:, 0, 0, 0].max() == 0.0 and ng[0, 0, 0] == 0.0                            # the image corner misses the sphere
    assert np.allclose(ng[:, :, 0], ng[::-1, :, 0]) and np.allclose(ng[:, :, 0], ng[:, :, 0].T)
    assert td.max() == 0.0


def test_hubble_flow_shifts_the_profile(oracle_mod):
    """Hubble-like outflow: along the central sight line the gas spans -Vexp..+Vexp, so tau(x) becomes a plateau of width
    2 Vexp/v_th, symmetric about x = 0, with the frequency integral of the static sphere"""
    kw = dict(temperature=1e4, taumax=1e3, rmax=1.0, nx=41, ny=41, nz=41, nxfreq=81, nxim=5, nyim=5, save_peeloff=1)
    o0 = oracle_mod.Oracle(**kw); o1 = oracle_mod.Oracle(Vexp=100.0, velocity_type="hubble", **kw)
    t0 = o0.sightline_tau()[0][:, 2, 2, 0]; t1 = o1.sightline_tau()[0][:, 2, 2, 0]
    x1 = o1.xfreq_axis()
    centre = np.interp(0.0, x1, t1)
    assert centre < 0.2 * t0[40]
    m = np.abs(x1) < 6.0
    assert np.allclose(np.interp(-x1[m], x1, t1), t1[m], rtol=0.02)                       # symmetric (cell-staircase level)
    assert np.isclose(centre, 2e3 * np.sqrt(np.pi) / (2 * 100.0 / (0.12895319011972164 / np.sqrt(1.00797) * 100.0)), rtol=0.02)   # plateau = tau0 sqrt(pi) / (2 Vexp/v_th)
    assert np.isclose(t0.sum() * o0.scalar("dxfreq"), t1.sum() * o1.scalar("dxfreq"), rtol=0.02)


def test_inside_observer_map(oracle_mod):
    """HEALPix observer at the centre of a static uniform sphere: every pixel sees tau0 at line centre up to the staircase
    of the cell-centre-carved boundary, and N(HI) likewise"""
    o = oracle_mod.Oracle(temperature=1e4, taumax=1e2, rmax=1.0, nx=41, ny=41, nz=41, nxfreq=21, nside=2, obsx=0.0, obsy=0.0, obsz=0.0)
    tg, ng, td = o.sightline_tau()
    assert tg.shape == (21, 48, 1, 1)
    a = o.array("voigt_a")[0]
    assert np.all(np.abs(tg[10, :, 0, 0] / 1e2 - 1) < 0.05) and np.all(np.abs(ng[:, 0, 0] / o.scalar("N_gasmax") - 1) < 0.05)


def test_octree_and_clump_maps(oracle_mod, tmp_path):
    from lart_amd import amr
    fn = str(tmp_path / "s.dat"); amr.write_generic_amr(fn, amr.make_amr_sphere(level_min=2, level_max=5))
    o = oracle_mod.Oracle(use_amr_grid=True, amr_file=fn, temperature=1e4, taumax=1e2, rmax=1.0, nxfreq=21, nxim=7, nyim=7, save_peeloff=1)
    tg, ng, _ = o.sightline_tau()
    a = o.array("voigt_a")[0]
    assert np.isclose(tg[10, 3, 3, 0], 2e2, rtol=1e-6) and np.isclose(ng[3, 3, 0], 2 * o.scalar("N_gasmax"), rtol=1e-6)
    assert ng[0, 0, 0] == 0.0
    oi = oracle_mod.Oracle(use_amr_grid=True, amr_file=fn, temperature=1e4, taumax=1e2, rmax=1.0, nxfreq=21, nside=2, obsx=0.0, obsy=0.0, obsz=0.0)
    ti, ni, _ = oi.sightline_tau()                                                           # make_sightline_tau_inside_amr
    assert ti.shape == (21, 48, 1, 1) and np.all(np.abs(ti[10, :, 0, 0] / 1e2 - 1) < 0.05) and np.all(np.abs(ni[:, 0, 0] / oi.scalar("N_gasmax") - 1) < 0.05)
    # clumps: <N(HI)> over the disc = f_cov-weighted clump columns; here only: non-negative, zero outside the sphere, and
    # tau at line centre = N * cross0 / Dfreq * H(0) for static clumps (sigma_v = 0)
    oc = oracle_mod.Oracle(use_clump_medium=True, clump_radius=0.05, clump_f_cov=3.0, clump_tau0=10.0, clump_sigma_v=0.0, temperature=1e4,
                           rmax=1.0, nxfreq=21, nxim=9, nyim=9, save_peeloff=1, iseed=3)
    tc, nc, _ = oc.sightline_tau()
    assert nc.min() >= 0.0 and nc[0, 0, 0] == 0.0 and nc[4, 4, 0] > 0.0
    H0 = oracle_mod.voigt(0.0, oc.scalar("cl_voigt_a_ref"))
    assert np.allclose(tc[10, :, :, 0], nc[:, :, 0] * oc.scalar("cross0") / oc.scalar("cl_Dfreq_ref") * H0, rtol=1e-9)


def test_overlap_forms_of_the_clump_maps(oracle_mod):
    """raytrace_to_edge_tau_gas_clump_overlap / _column_clump_overlap (raytrace_clump.f90:922-1001).  Both forms are deterministic, so on a
    population WITHOUT overlaps the overlap forms must give the maps of the standard forms to rounding (moving clumps, dust column); on
    overlapping clumps at rest the column is additive: tau(line centre) = N cross0 / Dfreq H(0), pixel by pixel"""
    kw = dict(use_clump_medium=True, clump_radius=0.05, clump_f_cov=3.0, clump_tau0=10.0, clump_sigma_v=40.0, temperature=1e4, rmax=1.0, nxfreq=41,
              velocity_min=-150.0, velocity_max=150.0, nxim=9, nyim=9, save_peeloff=1, iseed=3, DGR=1.0, cext_dust=3e-18, alpha=20.0, beta=50.0)
    a = oracle_mod.Oracle(**kw); ta, na, da = a.sightline_tau()
    b = oracle_mod.Oracle(**kw); b.set_array("cl_has_overlap", np.array([1.0])); tb, nb, db = b.sightline_tau()
    assert na.max() > 0 and np.allclose(na, nb, rtol=1e-11, atol=1e-12 * na.max()) and np.allclose(da, db, rtol=1e-11, atol=1e-12 * da.max())
    assert np.allclose(ta, tb, rtol=1e-9, atol=1e-12 * ta.max())
    ov = oracle_mod.Oracle(use_clump_medium=True, clump_radius=0.3, clump_f_cov=2.0, clump_tau0=10.0, clump_allow_overlap=True, temperature=1e4, rmax=1.0,
                           nxfreq=21, nxim=9, nyim=9, save_peeloff=1, iseed=7)
    assert bool(ov.scalar("cl_has_overlap"))
    tc, nc, _ = ov.sightline_tau()
    H0 = oracle_mod.voigt(0.0, ov.scalar("cl_voigt_a_ref"))
    assert np.allclose(tc[10, :, :, 0], nc[:, :, 0] * ov.scalar("cross0") / ov.scalar("cl_Dfreq_ref") * H0, rtol=1e-9)
    # the central pixel looks through the whole sphere: its column exceeds that of any single clump chord (2 r = 0.6 of tau0 / r per unit length)
    assert tc[10, 4, 4, 0] > 2 * 10.0
