"""Line type 8 ('ly_beta': Ly-beta with the 3p -> 2s conversion to H-alpha) and the H2 lines next to Ly-alpha on the device.

Function level: the ray marches against vectors from the reference's own compiled raytrace_car.f90 / h2_mod.f90 (tests/golden/lyb_h2.npz), scattering
events against the oracle (whose do_resonance8 / h2_select_line_D are pinned on the compiled line_mod.f90 / h2_mod.f90, tests/test_oracle_lyb_h2.py).
Whole path: photon by photon against the oracle on the same streams, the weight budget of output_sum_rect.f90:452-485 as a conservation check, the
band-2 spectra, the H-alpha peel cube and the conversion-rate profile; the reference's vetoes (setup.f90:239-280, 302-348) as errors."""
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


def sim_from_params(params, z, prefix, h2_dir=None):
    from lart_amd import host
    kw = dict(tok.split("=") for tok in str(params).split())
    par = host.Params()
    for k, v in kw.items():
        if k == "h2_data_dir":
            v = h2_dir
        host.set_param(par, k, host._parse_value(v) if k not in ("geometry", "line_id", "h2_model", "h2_data_dir") else v)
    sim = host.Simulation(par)
    g = sim.grid
    sim.ctx.set_grid_car(g.nx, g.ny, g.nz, g.xmin, g.ymin, g.zmin, g.dx, g.dy, g.dz, g.i0, g.j0, g.k0, g.xface, g.yface, g.zface,
                         z[prefix + "rhokap"], z[prefix + "voigt_a"], z[prefix + "Dfreq"], z[prefix + "vfx"], z[prefix + "vfy"], z[prefix + "vfz"], z[prefix + "rhokapD"])
    return sim


def test_ly_beta_marches_device_vs_reference_vectors(capi, golden):
    """raytrace_to_tau_car (raytrace_car.f90:1425-1648) for band-1 and band-2 packets and raytrace_to_edge_car_dust_Ha (:510-582): vectors from the compiled reference"""
    z = golden("lyb_h2.npz")
    sim = sim_from_params(z["lyb_params"], z, "lyb_")
    assert sim.cpar.line_type == 8 and np.isclose(sim.cpar.P_conv, z["lyb_multiplet"][14 + 7 + 5], rtol=0, atol=0)
    sim.ctx.zero_tallies()
    b1 = z["lyb_band"] == 1
    for band, kind in ((1, 0), (2, 3)):
        m = z["lyb_band"] == band
        r, ic, _ = sim.ctx.test_raytrace(kind, 0, z["lyb_r_in"][m], z["lyb_ic_in"][m], z["lyb_tau_in"][m])
        assert np.array_equal(ic, z["lyb_ic_out"][m]), band
        assert np.allclose(r, z["lyb_r_out"][m], rtol=1e-10, atol=1e-12), band
    _, _, te = sim.ctx.test_raytrace(4, 0, z["lyb_r_in"], z["lyb_ic_in"])
    assert np.allclose(te, z["lyb_tau_edge_Ha"], rtol=1e-11, atol=1e-300)
    t = sim.ctx.tallies()
    assert np.allclose(t["Jout_Ha"], z["lyb_Jout_Ha"], rtol=0, atol=1e-12) and np.allclose(t["Jout"], z["lyb_Jout"], rtol=0, atol=1e-12)
    assert np.isclose(t["W_esc1"], z["lyb_W_esc"][0]) and np.isclose(t["W_esc2"], z["lyb_W_esc"][1])


def test_h2_marches_device_vs_reference_vectors(capi, golden, h2_data_dir):
    """raytrace_to_tau_car / raytrace_to_edge_car with the H2 opacity rhokap h2_kappa (raytrace_car.f90:1490-1493, :473): vectors from the compiled reference"""
    z = golden("lyb_h2.npz")
    sim = sim_from_params(z["h2_params"], z, "h2_", h2_data_dir)
    ref = z["h2_init"][1]
    c = sim.cpar
    assert c.h2_nlines == 2 and np.isclose(c.h2_Dfreq_Hz, ref[1], rtol=1e-13)
    for i in range(2):
        assert np.allclose([c.h2_dnu_Hz[i], c.h2_strength[i], c.h2_a_damp[i], c.h2_p_scat[i]], ref[2 + 6 * i:6 + 6 * i], rtol=1e-12)       # the host's h2_init = the reference's
    sim.ctx.zero_tallies()
    r, ic, _ = sim.ctx.test_raytrace(0, 0, z["h2_r_in"], z["h2_ic_in"], z["h2_tau_in"])
    assert np.array_equal(ic, z["h2_ic_out"])
    assert np.allclose(r, z["h2_r_out"], rtol=1e-10, atol=1e-12)
    _, _, te = sim.ctx.test_raytrace(1, 0, z["h2_r_in"], z["h2_ic_in"])
    assert np.allclose(te, z["h2_tau_edge"], rtol=1e-10, atol=1e-300)
    assert np.allclose(sim.ctx.tallies()["Jout"], z["h2_Jout"], rtol=0, atol=1e-12)


LYB = dict(line_id="ly_beta", temperature=1e4, taumax=3e1, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=81, DGR=1.0, cext_dust=1e-15, cext_dust_Ha=5e-16)


def make(kw):
    from lart_amd import host
    return host.Simulation(dict(kw))


def oracle_of(oracle_mod, kw):
    return oracle_mod.Oracle(**{k: v for k, v in kw.items() if k != "iseed"})


@pytest.mark.parametrize("kw", [dict(LYB), dict(LYB, recoil=True, velocity_type="hubble", Vexp=30.0, use_reduced_wgt=True)])
def test_ly_beta_scatter_events_vs_oracle(capi, oracle_mod, kw):
    """scattering (scattering_car.f90:14-120) for band-1 packets (resonance with the conversion branch, or band-1 dust) and band-2 packets (dust only)"""
    sim = make(kw); o = oracle_of(oracle_mod, kw)
    rng = np.random.default_rng(5); n = 3000
    st = np.zeros((n, 18)); st[:, 0:3] = rng.uniform(-0.5, 0.5, (n, 3))
    k = rng.normal(size=(n, 3)); k /= np.linalg.norm(k, axis=1)[:, None]; st[:, 3:6] = k
    st[:, 6] = rng.uniform(-4, 4, n); st[:, 7] = 1.0; st[:, 14] = 1.0
    g = sim.grid
    cells = np.column_stack([np.floor((st[:, 0] - g.xmin) / g.dx) + 1, np.floor((st[:, 1] - g.ymin) / g.dy) + 1, np.floor((st[:, 2] - g.zmin) / g.dz) + 1]).astype(np.int32)
    band = np.where(np.arange(n) % 3 == 0, 2, 1).astype(np.int32)
    ic = np.column_stack([cells, band]).astype(np.int32)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    sg, icg, ndg = sim.ctx.test_scatter(st, ic, 55, ids)
    same = conv = 0
    for i in range(n):
        so, ico, ndo = o.scatter_event(st[i], ic[i], 55, int(ids[i]))
        assert ico[3] == icg[i][3], i                                            # left / band 1 / band 2: the same outcome
        if ndo == ndg[i] and np.allclose(sg[i][:8], so[:8], rtol=1e-9, atol=1e-11):
            same += 1
        conv += int(band[i] == 1 and ico[3] == 2)
    assert same >= 0.995 * n, same
    assert conv > 20                                                             # 11.8 % of the band-1 events that scatter off gas (not dust) convert


def test_h2_scatter_events_vs_oracle(capi, oracle_mod, h2_data_dir):
    kw = dict(temperature=1e3, taumax=1e3, nx=11, ny=11, nz=11, rmax=1.0, nxfreq=81, DGR=1.0, cext_dust=1e-19, h2_model="neufeld", f_H2=0.5, h2_temperature=4000.0,
              h2_data_dir=h2_data_dir)
    sim = make(kw); o = oracle_of(oracle_mod, kw)
    rng = np.random.default_rng(6); n = 4000
    st = np.zeros((n, 18)); st[:, 0:3] = rng.uniform(-0.5, 0.5, (n, 3))
    k = rng.normal(size=(n, 3)); k /= np.linalg.norm(k, axis=1)[:, None]; st[:, 3:6] = k
    vth = 0.12895319011972164 / np.sqrt(1.00797) * np.sqrt(1e3)
    st[:, 6] = np.where(np.arange(n) % 2 == 0, rng.normal(-14.14 / vth, 0.6, n), rng.normal(-99.229 / vth, 0.6, n)); st[:, 7] = 1.0; st[:, 14] = 1.0      # around the two H2 lines
    g = sim.grid
    ic = np.column_stack([np.floor((st[:, 0] - g.xmin) / g.dx) + 1, np.floor((st[:, 1] - g.ymin) / g.dy) + 1, np.floor((st[:, 2] - g.zmin) / g.dz) + 1, np.ones(n)]).astype(np.int32)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    sim.ctx.zero_tallies()
    sg, icg, ndg = sim.ctx.test_scatter(st, ic, 56, ids)
    same = 0; o.zero()
    for i in range(n):
        so, ico, ndo = o.scatter_event(st[i], ic[i], 56, int(ids[i]))
        assert ico[3] == icg[i][3], i
        same += int(ndo == ndg[i] and np.allclose(sg[i][:8], so[:8], rtol=1e-9, atol=1e-11))
    assert same >= 0.99 * n, same
    t = sim.ctx.tallies(); W = o.array("W_H2")
    assert W[0] > 0.2 * n and W[2] > 0 and W[3] > 0                             # H2 events dominate at these frequencies: destroyed packets, both lines pumped
    assert np.isclose(t["W_H2abs"], W[0]) and np.isclose(t["W_H2scat"], W[1]) and np.allclose(t["W_H2pump"], W[2:4])


def budget(t, n):
    return (t["W_esc1"] + t["W_abs1"] + t["W_conv"]) / n, (t["W_esc2"] + t["W_abs2"]) - t["W_conv"]


@pytest.mark.parametrize("extra,n", [(dict(nxim=12, nyim=12, calc_P=True), 40000), (dict(taumax=3e2, velocity_type="hubble", Vexp=40.0, recoil=True, use_reduced_wgt=True), 40000)])
def test_ly_beta_histories_photon_by_photon(capi, oracle_mod, extra, n):
    kw = dict(LYB, iseed=61, **extra)
    sim = make(kw); sim.ctx.enable_photon_log(True); sim.run_simulation(n); t = sim.ctx.tallies()
    o = oracle_of(oracle_mod, kw); o.run(n, seed=kw["iseed"])
    ng, xg = sim.ctx.photon_log(n); no, xo = o.photon_log()
    same = np.isclose(ng, no, rtol=1e-12, atol=0) & (np.isclose(xg, xo, rtol=1e-7, atol=1e-9) | (xo == 0))
    assert same.mean() > 0.999, same.mean()
    b1, b2 = budget(t, n)
    assert abs(b1 - 1.0) < 1e-9 and abs(b2) < 1e-6 * n                           # output_sum_rect.f90:452-467: nothing is lost in either band
    W = o.array("W")
    for k, i in (("W_conv", 0), ("W_esc1", 1), ("W_abs1", 2), ("W_esc2", 3), ("W_abs2", 4)):
        assert abs(t[k] - W[i]) < 2e-3 * n, (k, t[k], W[i])
    assert rms_diff(t["Jout_Ha"], o.array("Jout_Ha")) < 0.01 and rms_diff(t["Jout"], o.array("Jout")) < 0.02
    assert abs(t["Jabs_Ha"].sum() - o.array("Jabs_Ha").sum()) < 2e-3 * n
    if "nxim" in extra:
        nh = int(o.scalar("nxfreq_Ha"))
        gH, oH = t["peel_Ha"][..., 0], o.array("peel_Ha:0").reshape((nh, 12, 12), order="F")
        assert abs(gH.sum() / oH.sum() - 1.0) < 0.01 and rms_diff(gH.sum(axis=(1, 2)), oH.sum(axis=(1, 2))) < 0.02
        gs, os_ = t["scatt"][..., 0], o.array("scatt:0").reshape((81, 12, 12), order="F")
        assert abs(gs.sum() / os_.sum() - 1.0) < 0.02
        assert np.allclose(t["Pconv"], o.array("Pconv"), rtol=0.05, atol=0.01 * o.array("Pconv").max())
        assert abs(t["Pconv"].sum() / t["P1"].sum() - sim.cpar.P_conv) < 0.01     # Pconv / Pa -> P_down(2) (scattering_car.f90:865)


@pytest.fixture(scope="module")
def amr_file(tmp_path_factory):
    from lart_amd import amr
    fn = str(tmp_path_factory.mktemp("amr") / "sphere_small.dat")
    amr.write_generic_amr(fn, amr.make_amr_sphere(level_min=2, level_max=5))
    return fn


def test_ly_beta_on_the_octree(capi, oracle_mod, amr_file):
    """do_resonance8_amr / scatter_dust_Ha_amr / raytrace_*_amr with photon%iband (scattering_amr.f90:147-209, 579-625; raytrace_amr.f90:111-127, 353-390)"""
    kw = dict(line_id="ly_beta", use_amr_grid=True, amr_file=amr_file, temperature=1e4, taumax=3e1, rmax=1.0, nxfreq=81, DGR=1.0, cext_dust=1e-15, cext_dust_Ha=5e-16,
              nxim=10, nyim=10, iseed=62)
    n = 30000
    sim = make(kw); sim.ctx.enable_photon_log(True); sim.run_simulation(n); t = sim.ctx.tallies()
    o = oracle_of(oracle_mod, kw); o.run(n, seed=62)
    ng, xg = sim.ctx.photon_log(n); no, xo = o.photon_log()
    same = np.isclose(ng, no, rtol=1e-12, atol=0) & (np.isclose(xg, xo, rtol=1e-7, atol=1e-9) | (xo == 0))
    assert same.mean() > 0.999, same.mean()
    b1, b2 = budget(t, n)
    assert abs(b1 - 1.0) < 1e-9 and abs(b2) < 1e-6 * n
    assert rms_diff(t["Jout_Ha"], o.array("Jout_Ha")) < 0.01
    nh = int(o.scalar("nxfreq_Ha"))
    gH, oH = t["peel_Ha"][..., 0], o.array("peel_Ha:0").reshape((nh, 10, 10), order="F")
    assert abs(gH.sum() / oH.sum() - 1.0) < 0.01


@pytest.mark.parametrize("extra,n", [(dict(nx=1, ny=1, nz=51, xmax=1.0, ymax=1.0, zmax=1.0, xy_periodic=True), 7000),
                                     (dict(nx=21, ny=21, nz=21, rmax=1.0, nxim=10, nyim=10, DGR=1.0, cext_dust=1e-19), 7000)], ids=["slab", "sphere_peel_dust"])
def test_h2_histories_photon_by_photon(capi, oracle_mod, h2_data_dir, extra, n):
    """Ly-alpha with the H2 lines (Neufeld 1990 slab set-up and a dusty sphere with an observer): histories, the H2 budget of output_sum_rect.f90:470-485"""
    kw = dict(dict(temperature=1e3, taumax=1e4, nxfreq=121, h2_model="neufeld", f_H2=0.3, h2_temperature=3000.0, h2_data_dir=h2_data_dir, iseed=63), **extra)
    sim = make(kw); sim.ctx.enable_photon_log(True); sim.run_simulation(n); t = sim.ctx.tallies()
    o = oracle_of(oracle_mod, kw); o.run(n, seed=63)
    ng, xg = sim.ctx.photon_log(n); no, xo = o.photon_log()
    same = np.isclose(ng, no, rtol=1e-12, atol=0) & (np.isclose(xg, xo, rtol=1e-7, atol=1e-9) | (xo == 0))
    assert same.mean() > 0.995, same.mean()
    W = o.array("W_H2")
    assert W[0] > 0.01 * n                                                        # H2 destroys a visible share of the packets
    assert abs(t["W_H2abs"] - W[0]) < 3e-3 * n and abs(t["W_H2scat"] - W[1]) < 3e-3 * n + 1e-3 * W[1] and np.allclose(t["W_H2pump"], W[2:4], rtol=0.02, atol=3e-3 * n)
    absorbed = t["Jabs"].sum() if "DGR" in extra else 0.0
    assert abs(t["Jout"].sum() + t["W_H2abs"] + absorbed - n) < 2e-3 * n            # escaped + destroyed by H2 (+ absorbed by dust) = emitted (a few leave the frequency axis)
    assert rms_diff(t["Jout"], o.array("Jout")) < 0.01
    if "nxim" in extra:
        gs, os_ = t["scatt"][..., 0], o.array("scatt:0").reshape((121, 10, 10), order="F")
        assert abs(gs.sum() / os_.sum() - 1.0) < 0.02 and rms_diff(gs.sum(axis=(1, 2)), os_.sum(axis=(1, 2))) < 0.03
    noh2 = make({k: v for k, v in kw.items() if not k.startswith("h2_") and k != "f_H2"}); noh2.run_simulation(5000)
    assert noh2.ctx.tallies()["Jout"].sum() > t["Jout"].sum() / n * 5000 + 50         # the lines matter


@pytest.mark.parametrize("case", ["lyb_peel_profiles", "lyb_outflow_reduced_weight", "lyb_octree_peel", "h2_sphere_peel_dust", "h2_slab", "h2_octree"])
def test_tallies_equal_the_oracle_on_identical_histories(capi, oracle_mod, h2_data_dir, amr_file, case):
    """every tally of the two line types bin for bin on batches of photons whose histories are the oracle's: both spectra of either band, the H-alpha peel cube, the
    conversion profile, the scattered-light cube, the weight budgets -- the comparisons above allow 1-3 %"""
    H2 = dict(temperature=1e3, taumax=1e3, nxfreq=121, h2_model="neufeld", f_H2=0.3, h2_temperature=3000.0, h2_data_dir=h2_data_dir)
    if case == "lyb_peel_profiles":
        kw = dict(LYB, nxim=12, nyim=12, calc_P=True)
    elif case == "lyb_outflow_reduced_weight":
        kw = dict(LYB, taumax=1e2, velocity_type="hubble", Vexp=40.0, recoil=True, use_reduced_wgt=True, nxim=12, nyim=12)
    elif case == "lyb_octree_peel":
        kw = dict(line_id="ly_beta", use_amr_grid=True, amr_file=amr_file, temperature=1e4, taumax=3e1, rmax=1.0, nxfreq=81, DGR=1.0, cext_dust=1e-15, cext_dust_Ha=5e-16, nxim=12, nyim=12)
    elif case == "h2_sphere_peel_dust":
        kw = dict(H2, nx=21, ny=21, nz=21, rmax=1.0, nxim=12, nyim=12, DGR=1.0, cext_dust=1e-19)
    elif case == "h2_slab":
        kw = dict(H2, nx=1, ny=1, nz=51, xmax=1.0, ymax=1.0, zmax=1.0, xy_periodic=True)
    else:
        kw = dict(H2, use_amr_grid=True, amr_file=amr_file, rmax=1.0, nxim=12, nyim=12)
    kw["iseed"] = 1
    sim = make(kw); sim.ctx.enable_photon_log(True)
    o = oracle_of(oracle_mod, kw)
    lyb = kw.get("line_id") == "ly_beta"
    names = ["Jout", "Jin", "Jabs"] + (["Jout_Ha", "Jabs_Ha"] if lyb else []) + (["Pconv", "P1"] if kw.get("calc_P") else [])
    cubes = (["scatt", "direc"] + (["peel_Ha"] if lyb else [])) if "nxim" in kw else []
    scal = ["W_conv", "W_esc1", "W_abs1", "W_esc2", "W_abs2"] if lyb else ["W_H2abs", "W_H2scat"]
    per, nclean, prev, tot = 16, 0, None, 0.0
    for b in range(40):
        first = 1 + b * per
        sim.ctx.run(per, first_id=first, stride=1, seed=99); t = sim.ctx.tallies()
        ng, xg = sim.ctx.photon_log(per)
        o.zero(); o.run(per, first_id=first, stride=1, seed=99); no, xo = o.photon_log()
        cur = {k: np.array(t[k], dtype=float).copy() for k in names + cubes + scal + (["W_H2pump"] if not lyb else [])}
        dev = {k: (cur[k] - prev[k] if prev is not None else cur[k]) for k in cur}; prev = cur
        if not (np.all(np.isclose(ng, no, rtol=1e-12, atol=0)) and np.all(np.isclose(xg, xo, rtol=1e-9, atol=1e-12) | (xo == 0.0))):
            continue
        nclean += 1
        for k in names:
            r = o.array(k).reshape(dev[k].shape, order="F")
            assert np.allclose(dev[k], r, rtol=1e-9, atol=1e-12 * per), (case, k, b, float(np.abs(dev[k] - r).max()))
        scale = max(np.abs(o.array("scatt:0")).max(), 1e-300) if cubes else 1.0     # (one scale for all cubes: the direct light of an opaque medium is e^-700, denormal numbers)
        for k in cubes:
            r = o.array(k + ":0"); d = dev[k][..., 0].ravel(order="F")
            assert np.allclose(d, r, rtol=1e-9, atol=1e-9 * scale), (case, k, b, float(np.abs(d - r).max() / scale))
            tot += np.abs(r).sum() if k != "direc" else 0.0
        W = o.array("W") if lyb else o.array("W_H2")
        for i, k in enumerate(scal):
            assert abs(dev[k] - W[i]) < 1e-9 * per, (case, k, b, float(dev[k]), float(W[i]))
        if not lyb:
            assert np.allclose(dev["W_H2pump"], W[2:2 + len(np.atleast_1d(dev["W_H2pump"]))], rtol=1e-9, atol=1e-9 * per)
    assert nclean >= 8 and (tot > 0 or not cubes), (nclean, tot)


def test_ly_beta_and_h2_divergence_audit(capi, oracle_mod, h2_data_dir):
    from test_gpu_divergence import audit
    kw = dict(LYB, taumax=1e2)
    frac, unexplained, rep = audit(make(dict(kw, iseed=5)), oracle_of(oracle_mod, kw), 100000, 5, "ly_beta")
    assert not unexplained and frac > 0.999, (rep, unexplained[:5])
    kw = dict(temperature=1e3, taumax=1e3, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=121, h2_model="neufeld", f_H2=0.3, h2_temperature=3000.0, h2_data_dir=h2_data_dir)
    frac, unexplained, rep = audit(make(dict(kw, iseed=5)), oracle_of(oracle_mod, kw), 100000, 5, "H2")
    assert not unexplained and frac > 0.999, (rep, unexplained[:5])


def test_vetoes_of_the_reference(capi, h2_data_dir):
    """setup.f90:239-280 (ly_beta) and :302-348 (H2): what the reference aborts on is an error here, at the host and at the C-ABI"""
    from lart_amd import host
    for bad in (dict(use_stokes=True), dict(xyz_symmetry=True), dict(xy_periodic=True, nx=1, ny=1), dict(use_clump_medium=True, clump_radius=0.1, clump_f_cov=1.0, clump_tau0=1.0),
                dict(fine_structure=True), dict(nside=4)):
        with pytest.raises((host.LartInputError, capi.LartError, RuntimeError)):
            make(dict(LYB, **bad))
    for bad in (dict(use_stokes=True), dict(xyz_symmetry=True), dict(line_id="CIV_1548", N_gasmax=1e14), dict(f_H2=0.0), dict(h2_model="lte"), dict(xy_periodic=True, nx=4, ny=4)):
        with pytest.raises((host.LartInputError, capi.LartError, RuntimeError)):
            make(dict(dict(temperature=1e3, taumax=1e2, nx=11, ny=11, nz=11, rmax=1.0, h2_model="neufeld", f_H2=0.1, h2_data_dir=h2_data_dir), **bad))
    sim = make(dict(LYB, core_skip=True))                     # ly_beta forces core_skip off (setup.f90:281-290)
    assert sim.cpar.core_skip == 0 and sim.cpar.line_type == 8
    p = capi.LartParams(); 
    for f, _ in sim.cpar._fields_:
        setattr(p, f, getattr(sim.cpar, f))
    p.use_stokes = 1
    with pytest.raises(capi.LartError, match="ly_beta"):
        capi.Context(p)
