"""BASELINE.json's configurations at their FULL optical depth on the GPU against the committed reference spectra
(tests/golden/c1_slab_tau4.npz, c2_tau6.npz, c3_peel_tau6.npz: CPU oracle on the reference's MT19937-64 stream, tools/gen_golden_bigtau.py):

  C1  examples/slab/t1tau4.in          slab tau0 = 1e4, nz = 201, Stokes                         (run_simulation_mod.f90:150-202, output_sum_rect.f90:185)
  C2  examples/sphere/t1tau6.in        sphere tau0 = 1e6, T = 10 K, 101^3 octant                   (the headline; output_sum_rect.f90:194-202)
  C3  examples/sphere_peel/t1tau6.in   sphere tau0 = 1e6, 201^3, Stokes, 129 x 129 x 201 peel cube (peelingoff_rect.f90:303-482, output_sum_rect.f90:428-430)
  C4  examples/clump_sphere/clump_tau3_fcov5_peel.in + the dust lines of sphere_peel/t1tau7dust.in:5-10 + core_skip: 66 667 clumps, tau_cl = 1e3,
      Mueller Ly-alpha dust, Stokes, 129 x 129 x 301 cube (raytrace_clump.f90:83-201, scattering_car.f90:201-329)      -> c4_clump_dust.npz
  C5  examples/amr_sphere_generic/sphere_amr_static.in on the 178 480-leaf octree, tau0 = 1e4, calc_J + calc_P + calc_Pnew, geometry_JPa = 1
      (raytrace_amr.f90:77-259, :848-912, scattering_amr.f90:917-942)                                                   -> c5_amr_JP.npz

All five run on the one GPU at the same time (conftest.run_together: one context, stream and host thread each), because a run lasts as long as
its longest history on a lone wave and those tails overlap; the results are those of the same runs made one after the other.

Metric (BASELINE.json / SURVEY section 8d): rms_diff = sqrt(mean_i (J_gpu,i - J_ref,i)^2) / max_i J_ref,i over the nxfreq bins, to be <= 1 %,
quoted next to the Poisson-expected rms of the two independent runs; chi^2 per bin says whether the difference IS the noise; <N_scatt> within
the errors of the two means."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from lart_amd import capi
    assert capi.lib().lart_gpu_device_count() > 0, "no GPU visible: the HIP path cannot run (there is no fallback)"
    return capi


def params_of(fx):
    import json
    return json.loads(str(fx["params"]))


def compare(jg_raw, ng, jr_raw, nr, label):
    """per-photon spectra of two independent runs: rms-diff of the metric, its Poisson expectation (unit weights), chi^2 per populated bin"""
    jg, jr = jg_raw / ng, jr_raw / nr
    p = (jg_raw + jr_raw) / (ng + nr)
    var = p * (1.0 / ng + 1.0 / nr)
    pk = jr.max()
    rms = float(np.sqrt(np.mean((jg - jr) ** 2)) / pk); exp = float(np.sqrt(np.mean(var)) / pk)
    ok = var > 0
    chi2 = float(np.mean((jg - jr)[ok] ** 2 / var[ok]))
    print("%s: rms-diff %.4f  Poisson-expected %.4f  chi2/bin %.2f  (N_gpu %d, N_ref %d)" % (label, rms, exp, chi2, ng, nr))
    return rms, exp, chi2


N_DEVICE = {"c1": 1000000, "c2": 400000, "c3": 80000, "c4": 1000000, "c5": 300000}       # (the run time is the longest history of c2 / c3, not the photon count)
SEED = {"c1": 20261005, "c2": 20261004, "c3": 20261006, "c4": 20261007, "c5": 20261008}
FIXTURE = {"c1": "c1_slab_tau4.npz", "c2": "c2_tau6.npz", "c3": "c3_peel_tau6.npz", "c4": "c4_clump_dust.npz", "c5": "c5_amr_JP.npz"}


def run_all():
    """the five BASELINE configurations, run together; {name: (fixture, simulation, tallies, per-photon scattering counts)}.  Called by conftest.Background when the
    collection ends (host threads and contexts of its own, next to the other device tests), or by the fixture below when this file is imported some other way"""
    import sys
    import tempfile
    import __graft_entry__ as g
    g.build()
    from conftest import run_together, GOLDEN
    from lart_amd import host, capi
    assert capi.lib().lart_gpu_device_count() > 0, "no GPU visible: the HIP path cannot run (there is no fallback)"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_golden_bigtau import materialize
    tmp = tempfile.mkdtemp(prefix="bigtau")
    sims, fxs = {}, {}
    for name, f in FIXTURE.items():
        if not os.path.exists(os.path.join(GOLDEN, f)):
            continue
        fxs[name] = np.load(os.path.join(GOLDEN, f), allow_pickle=False)
        kw = materialize(params_of(fxs[name]), tmp)            # iseed of c4 = the clump-placement stream of the fixture; the photon stream is SEED[name]
        sims[name] = host.Simulation(dict(kw, iseed=kw.get("iseed", SEED[name])))
        sims[name].ctx.enable_photon_log(True)
    run_together([(sims[k], N_DEVICE[k], SEED[k]) for k in sims])
    out = {}
    for k, sim in sims.items():
        out[k] = (fxs[k], sim, sim.ctx.tallies(), sim.ctx.photon_log(N_DEVICE[k])[0])
    return out


@pytest.fixture(scope="module")
def runs(capi):
    import conftest
    job = conftest.BACKGROUND.get("bigtau")
    return job.join() if job is not None else run_all()


def test_c2_sphere_tau6_spectrum_within_one_percent(runs):
    """the headline at tau0 = 1e6 on the device: 4e5 photons here (bench.py runs 3e6 against the same fixture)"""
    fx, sim, t, ns = runs["c2"]; n = N_DEVICE["c2"]
    assert t["nphotons_done"] == n
    rms, exp, chi2 = compare(t["Jout"], n, fx["Jout_raw"], float(fx["nphotons"]), "C2 Jout")
    assert rms <= 0.01, (rms, exp)
    assert chi2 < 1.6 and rms < 1.5 * exp + 0.002, (rms, exp, chi2)
    err = np.hypot(ns.std() / np.sqrt(n), float(fx["nscatt_err"]))
    assert abs(ns.mean() - float(fx["nscatt_mean"])) < 4 * err, (ns.mean(), float(fx["nscatt_mean"]), err)
    # normalisation of output_normalize (output_sum_rect.f90:194-202): integral of Jout over x and the sphere's surface = 1 photon / (2 pi)
    out = sim.output_normalize(t)
    assert np.isclose(out["Jout"].sum() * float(fx["Jout_scale_per_photon"]), 1.0, rtol=1e-3)      # (a few 1e-4 leave the frequency axis)
    assert np.allclose(out["Jout"], t["Jout"] / (n * float(fx["Jout_scale_per_photon"])), rtol=1e-12)


def test_c1_slab_tau4_spectrum(runs):
    fx, sim, t, ns = runs["c1"]; n = N_DEVICE["c1"]
    rms, exp, chi2 = compare(t["Jout"], n, fx["Jout_raw"], float(fx["nphotons"]), "C1 Jout")
    assert rms <= 0.01 and chi2 < 1.6, (rms, exp, chi2)
    rj, ej, cj = compare(t["Jin"], n, fx["Jin_raw"], float(fx["nphotons"]), "C1 Jin")
    assert rj <= 0.01 and cj < 1.6
    err = np.hypot(ns.std() / np.sqrt(n), float(fx["nscatt_err"]))
    assert abs(ns.mean() - float(fx["nscatt_mean"])) < 4 * err


def weighted(g_raw, n, fx, key, nr, label, floor=0.05):
    """a weighted estimator (peel-off, path-length and event tallies) against the fixture: the noise comes from the fixture's chunk-to-chunk
    scatter (err of the raw sum), scaled by the photon numbers for the device run; rms-diff of the metric, its expectation, chi^2 over the
    bins that carry the quantity (> floor of the peak)"""
    g = np.asarray(g_raw, float).ravel(order="F") / n; r = fx[key + "_raw"] / nr
    sig = fx[key + "_raw_err"] / nr * np.sqrt(1.0 + nr / n)
    pk = r.max()
    rms = float(np.sqrt(np.mean((g - r) ** 2)) / pk); exp = float(np.sqrt(np.mean(sig ** 2)) / pk)
    ok = (sig > 0) & (r > floor * pk)
    chi2 = float(np.mean((g - r)[ok] ** 2 / sig[ok] ** 2))
    print("%s: rms-diff %.4f expected %.4f chi2/bin %.2f (%d bins)" % (label, rms, exp, chi2, int(ok.sum())))
    return rms, exp, chi2, g, r, sig


def peel_checks(t, n, fx, nr, names, label, nimg):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_golden_bigtau import radial_profile
    for name in names:
        rms, exp, chi2, g, r, sig = weighted(t[name][..., 0].sum(axis=(1, 2)), n, fx, name + "_spec", nr, "%s peel %s spectrum" % (label, name))
        assert rms < 2.0 * exp + 0.003 and chi2 < 3.0, (name, rms, exp, chi2)
        assert abs(g.sum() / r.sum() - 1.0) < 4.0 * np.sqrt(np.sum(sig ** 2)) / r.sum() + 0.003, name
        rms, exp, chi2, g, r, sig = weighted(radial_profile(t[name][..., 0].sum(axis=0), 32), n, fx, name + "_img_r", nr, "%s peel %s radial profile" % (label, name))
        assert chi2 < 3.0, (name, chi2)


def test_c3_sphere_peel_tau6_full_size(runs):
    """201^3 cells, Stokes, one far observer with a 129 x 129 x 201 cube, tau0 = 1e6: emergent spectrum, peel-integrated spectra of the
    scattered light and of Stokes I, the radial surface-brightness profile, and Q, U consistent with zero when integrated over a sphere's image"""
    fx, sim, t, ns = runs["c3"]; n = N_DEVICE["c3"]
    nr = float(fx["nphotons"])
    rms, exp, chi2 = compare(t["Jout"], n, fx["Jout_raw"], nr, "C3 Jout")
    assert rms <= 0.02 and chi2 < 1.6 and rms < 1.5 * exp + 0.003, (rms, exp, chi2)
    err = np.hypot(ns.std() / np.sqrt(n), float(fx["nscatt_err"]))
    assert abs(ns.mean() - float(fx["nscatt_mean"])) < 4 * err
    peel_checks(t, n, fx, nr, ("scatt", "I"), "C3", 129)
    # direct light: tau0 = 1e6 lets only the far Lorentz wing of the Voigt emission through
    assert t["direc"].sum() / n < 1e-3 * t["scatt"].sum() / n and float(fx["direc_spec_raw"].sum()) / nr < 1e-3 * float(fx["scatt_spec_raw"].sum()) / nr
    # Q, U: the image of a sphere is radially polarised, the pixel pattern changes sign across the image: pixel-by-pixel against the fixture
    for name in ("Q", "U"):
        g = t[name][..., 0].sum(axis=0).ravel() / n; r = fx[name + "_img_raw"] / nr
        sig = fx[name + "_img_raw_err"] / nr * np.sqrt(1.0 + nr / n)
        ok = sig > np.percentile(sig, 50)                   # the better-sampled half of the pixels
        assert float(np.mean((g - r)[ok] ** 2 / sig[ok] ** 2)) < 3.0, name
        assert abs(g.sum()) < 0.02 * (t["I"][..., 0].sum() / n)


def test_c4_clumps_dust_stokes_full_size(runs):
    """BASELINE configs[3]: 66 667 clumps of tau_cl = 1e3 (f_cov = 5), dust with the Mueller Ly-alpha table, Stokes, core_skip switch, one far
    observer 129 x 129 x 301: the same clump population as the fixture (same placement stream), emergent and absorbed spectra, <N_scatt>, the
    dust share, peel-integrated spectra and radial profiles of the scattered light and of Stokes I"""
    if "c4" not in runs:
        pytest.skip("tests/golden/c4_clump_dust.npz not generated yet (tools/gen_golden_bigtau.py run c4)")
    fx, sim, t, ns = runs["c4"]; n = N_DEVICE["c4"]
    nr = float(fx["nphotons"])
    assert sim.clumps.N == int(fx["cl_N"]) == 66667 and sim.cpar.core_skip == 1 and sim.cpar.use_stokes == 1 and sim.cpar.DGR == 1.0
    rms, exp, chi2 = compare(t["Jout"], n, fx["Jout_raw"], nr, "C4 Jout")
    assert rms <= 0.01 and chi2 < 1.6 and rms < 1.5 * exp + 0.002, (rms, exp, chi2)
    rj, ej, cj = compare(t["Jin"], n, fx["Jin_raw"], nr, "C4 Jin")
    assert rj <= 0.01 and cj < 1.6
    err = np.hypot(ns.std() / np.sqrt(n), float(fx["nscatt_err"]))
    assert abs(ns.mean() - float(fx["nscatt_mean"])) < 4 * err, (ns.mean(), float(fx["nscatt_mean"]), err)
    # dust: tau_dust of a clump is 2.7e-5, so absorption is rare; absorbed + escaped = emitted, and the absorbed share agrees within its counting error
    ja, jr = t["Jabs"].sum(), float(fx["Jabs_raw"].sum())
    lost_g, lost_r = 1.0 - (t["Jout"].sum() + ja) / n, 1.0 - (float(fx["Jout_raw"].sum()) + jr) / nr      # packets that leave through the ends of the +-100 km/s axis
    assert 0.0 <= lost_g < 1e-3 and abs(lost_g - lost_r) < 4.0 * np.sqrt(max(lost_r, 1e-6) * (1.0 / n + 1.0 / nr)) + 1e-6, (lost_g, lost_r)
    assert abs(ja / n - jr / nr) < 4.0 * np.sqrt(jr / nr * (1.0 / n + 1.0 / nr)) + 1e-7, (ja / n, jr / nr)
    peel_checks(t, n, fx, nr, ("scatt", "I"), "C4", 129)
    rms, exp, chi2, g, r, sig = weighted(t["direc"][..., 0].sum(axis=(1, 2)), n, fx, "direc_spec", nr, "C4 peel direct spectrum")
    assert chi2 < 3.0 and abs(g.sum() / r.sum() - 1.0) < 4.0 * np.sqrt(np.sum(sig ** 2)) / r.sum() + 0.003
    for name in ("Q", "U"):
        g = t[name][..., 0].sum(axis=0).ravel() / n; r = fx[name + "_img_raw"] / nr
        sig = fx[name + "_img_raw_err"] / nr * np.sqrt(1.0 + nr / n)
        ok = sig > np.percentile(sig, 50)
        assert float(np.mean((g - r)[ok] ** 2 / sig[ok] ** 2)) < 3.0, name


def test_c5_octree_J_P_profiles_full_size(runs):
    """BASELINE configs[4]: the 178 480-leaf sphere at tau0 = 1e4 with the three profile tallies in radial shells (geometry_JPa = 1): J(x, r) of
    add_to_J_amr, P_alpha(r) of add_to_Pa_amr, P_alpha_new(r) of add_to_Pnew_amr, the emergent spectrum, the peel cube, and the reference's
    logged <N_scatt> = 2.8225e4 (examples/amr_sphere_generic/log_amr_1M.txt)"""
    if "c5" not in runs:
        pytest.skip("tests/golden/c5_amr_JP.npz not generated yet (tools/gen_golden_bigtau.py run c5)")
    fx, sim, t, ns = runs["c5"]; n = N_DEVICE["c5"]
    nr = float(fx["nphotons"])
    assert sim.amr.nleaf == int(fx["nleaf"]) == 178480 and sim.jpa.nbins == int(fx["nbins"])
    rms, exp, chi2 = compare(t["Jout"], n, fx["Jout_raw"], nr, "C5 Jout")
    assert rms <= 0.01 and chi2 < 1.6 and rms < 1.5 * exp + 0.002, (rms, exp, chi2)
    err = np.hypot(ns.std() / np.sqrt(n), float(fx["nscatt_err"]))
    assert abs(ns.mean() - float(fx["nscatt_mean"])) < 4 * err
    assert abs(ns.mean() / 2.8225e4 - 1.0) < 0.01 + 4 * ns.std() / np.sqrt(n) / 2.8225e4
    for name, tol in (("J1", 0.02), ("P1", 0.02), ("P1_new", 0.02)):
        rms, exp, chi2, g, r, sig = weighted(t[name], n, fx, name, nr, "C5 " + name, floor=0.02)
        assert rms <= tol and chi2 < 2.0, (name, rms, exp, chi2)
        assert abs(g.sum() / r.sum() - 1.0) < 4.0 * np.sqrt(np.sum(sig ** 2)) / r.sum() + 0.002, name
    # P1 and P1_new estimate the same scattering rate per atom (events against path length x opacity): their shell profiles agree
    pk = t["P1"].max()
    ok = t["P1"] > 0.05 * pk
    assert np.allclose(t["P1"][ok], t["P1_new"][ok], rtol=0.03)
    peel_checks(t, n, fx, nr, ("scatt",), "C5", 100)
