"""BASELINE.json's configurations at their FULL optical depth on the GPU against the committed reference spectra
(tests/golden/c1_slab_tau4.npz, c2_tau6.npz, c3_peel_tau6.npz: CPU oracle on the reference's MT19937-64 stream, tools/gen_golden_bigtau.py):

  C1  examples/slab/t1tau4.in          slab tau0 = 1e4, nz = 201, Stokes                         (run_simulation_mod.f90:150-202, output_sum_rect.f90:185)
  C2  examples/sphere/t1tau6.in        sphere tau0 = 1e6, T = 10 K, 101^3 octant                   (the headline; output_sum_rect.f90:194-202)
  C3  examples/sphere_peel/t1tau6.in   sphere tau0 = 1e6, 201^3, Stokes, 129 x 129 x 201 peel cube (peelingoff_rect.f90:303-482, output_sum_rect.f90:428-430)

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


def run(capi, fx, n, seed):
    from lart_amd import host
    sim = host.Simulation(dict(params_of(fx), iseed=seed))
    sim.ctx.enable_photon_log(True)
    sim.run_simulation(n)
    ns, _ = sim.ctx.photon_log(n)
    return sim, sim.ctx.tallies(), ns


def test_c2_sphere_tau6_spectrum_within_one_percent(capi, golden):
    """the headline: 1e6 photons at tau0 = 1e6 on the device (about a minute of transport + the drain tail)"""
    fx = golden("c2_tau6.npz")
    n = 1000000
    sim, t, ns = run(capi, fx, n, 20261004)
    rms, exp, chi2 = compare(t["Jout"], n, fx["Jout_raw"], float(fx["nphotons"]), "C2 Jout")
    assert rms <= 0.01, (rms, exp)
    assert chi2 < 1.6 and rms < 1.5 * exp + 0.002, (rms, exp, chi2)
    err = np.hypot(ns.std() / np.sqrt(n), float(fx["nscatt_err"]))
    assert abs(ns.mean() - float(fx["nscatt_mean"])) < 4 * err, (ns.mean(), float(fx["nscatt_mean"]), err)
    # normalisation of output_normalize (output_sum_rect.f90:194-202): integral of Jout over x and the sphere's surface = 1 photon / (2 pi)
    out = sim.output_normalize(t)
    assert np.isclose(out["Jout"].sum() * float(fx["Jout_scale_per_photon"]), 1.0, rtol=1e-3)      # (a few 1e-4 leave the frequency axis)
    assert np.allclose(out["Jout"], t["Jout"] / (n * float(fx["Jout_scale_per_photon"])), rtol=1e-12)


def test_c1_slab_tau4_spectrum(capi, golden):
    fx = golden("c1_slab_tau4.npz")
    n = 1000000
    sim, t, ns = run(capi, fx, n, 20261005)
    rms, exp, chi2 = compare(t["Jout"], n, fx["Jout_raw"], float(fx["nphotons"]), "C1 Jout")
    assert rms <= 0.01 and chi2 < 1.6, (rms, exp, chi2)
    rj, ej, cj = compare(t["Jin"], n, fx["Jin_raw"], float(fx["nphotons"]), "C1 Jin")
    assert rj <= 0.01 and cj < 1.6
    err = np.hypot(ns.std() / np.sqrt(n), float(fx["nscatt_err"]))
    assert abs(ns.mean() - float(fx["nscatt_mean"])) < 4 * err


def test_c3_sphere_peel_tau6_full_size(capi, golden):
    """201^3 cells, Stokes, one far observer with a 129 x 129 x 201 cube, tau0 = 1e6: emergent spectrum, peel-integrated spectra of the
    scattered light and of Stokes I, the radial surface-brightness profile, and Q, U consistent with zero when integrated over a sphere's image"""
    fx = golden("c3_peel_tau6.npz")
    n = 100000
    nr = float(fx["nphotons"])
    sim, t, ns = run(capi, fx, n, 20261006)
    rms, exp, chi2 = compare(t["Jout"], n, fx["Jout_raw"], nr, "C3 Jout")
    assert rms <= 0.02 and chi2 < 1.6 and rms < 1.5 * exp + 0.003, (rms, exp, chi2)
    err = np.hypot(ns.std() / np.sqrt(n), float(fx["nscatt_err"]))
    assert abs(ns.mean() - float(fx["nscatt_mean"])) < 4 * err
    # peel-off: a weighted estimator, so its noise comes from the fixture's chunk-to-chunk scatter (err of the raw sum) and the same relative
    # error scaled by the photon numbers for the device run
    for name in ("scatt", "I"):
        g = t[name][..., 0].sum(axis=(1, 2)) / n; r = fx[name + "_spec_raw"] / nr
        sig = fx[name + "_spec_raw_err"] / nr * np.sqrt(1.0 + nr / n)
        pk = r.max()
        rms = float(np.sqrt(np.mean((g - r) ** 2)) / pk); exp = float(np.sqrt(np.mean(sig ** 2)) / pk)
        # (chi^2 over the bins that carry the spectrum: a peel-off estimate is a sum of a few large weights in the far wings, where the
        #  chunk-to-chunk scatter of a handful of deposits is no error estimate)
        ok = (sig > 0) & (r > 0.05 * pk)
        chi2 = float(np.mean((g - r)[ok] ** 2 / sig[ok] ** 2))
        print("C3 peel %s spectrum: rms-diff %.4f expected %.4f chi2/bin(core) %.2f" % (name, rms, exp, chi2))
        assert rms < 2.0 * exp + 0.003 and chi2 < 3.0, (name, rms, exp, chi2)
        assert abs(g.sum() / r.sum() - 1.0) < 4.0 * np.sqrt(np.sum(sig ** 2)) / r.sum() + 0.003, name
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_golden_bigtau import radial_profile
    for name in ("scatt", "I"):
        g = radial_profile(t[name][..., 0].sum(axis=0), 32) / n; r = fx[name + "_img_r_raw"] / nr
        sig = fx[name + "_img_r_raw_err"] / nr * np.sqrt(1.0 + nr / n)
        ok = (sig > 0) & (r > 0.05 * r.max())
        assert float(np.mean((g - r)[ok] ** 2 / sig[ok] ** 2)) < 3.0, name
        assert np.all(g[:28] > 0)
    # direct light: tau0 = 1e6 lets only the far Lorentz wing of the Voigt emission through
    assert t["direc"].sum() / n < 1e-3 * t["scatt"].sum() / n and float(fx["direc_spec_raw"].sum()) / nr < 1e-3 * float(fx["scatt_spec_raw"].sum()) / nr
    # Q, U: the image of a sphere is radially polarised, the pixel pattern changes sign across the image: pixel-by-pixel against the fixture
    for name in ("Q", "U"):
        g = t[name][..., 0].sum(axis=0).ravel() / n; r = fx[name + "_img_raw"] / nr
        sig = fx[name + "_img_raw_err"] / nr * np.sqrt(1.0 + nr / n)
        ok = sig > np.percentile(sig, 50)                   # the better-sampled half of the pixels
        assert float(np.mean((g - r)[ok] ** 2 / sig[ok] ** 2)) < 3.0, name
        assert abs(g.sum()) < 0.02 * (t["I"][..., 0].sum() / n)
