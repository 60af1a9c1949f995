"""line_id = 'ly_alpha_HD' / par%include_deuterium (line type 7: hydrogen + deuterium Lyman-alpha, line_mod.f90:82-106, :476-549,
:1155-1191): the oracle's restatement against vectors from the reference's own compiled modules, and whole-path sanity."""
import numpy as np


def load(oracle_mod, golden):
    z = golden("deuterium.npz")
    o = oracle_mod.Oracle(str(z["params"]))
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD"):
        o.set_array(k, z[k])
    return z, o


def test_cross_species_constants_and_profile(oracle_mod, golden):
    z, o = load(oracle_mod, golden)
    got = [o.scalar(k) for k in ("delta_nu_HD_Hz", "ratio_Dfreq_HD", "ratio_voigta_HD", "g_recoil0_D")]
    assert int(o.scalar("line_type")) == 7 and np.allclose(got, z["constants"], rtol=1e-13, atol=0)
    for c, ref in zip(z["cells"], z["calc_voigt"]):
        assert np.allclose(o.calc_voigt(z["x"], *map(int, c)), ref, rtol=2e-11, atol=0)


def test_do_resonance_HD_on_the_reference_stream(oracle_mod, golden):
    z, o = load(oracle_mod, golden)
    npicked = 0
    for i, (x, ref) in enumerate(zip(z["xres"], z["resonance"])):
        got = o.do_resonance(0, 400 + i, 0, 256, float(x), *map(int, z["cells"][i % 3]))
        assert np.array_equal(got[:, 11], ref[:, 4])                                      # the same species, draw by draw
        assert np.allclose(got[:, :4], ref[:, :4], rtol=1e-9, atol=1e-12), (i, x)
        npicked += int((ref[:, 4] == 2).sum())
    assert npicked > 50                                                                   # deuterium atoms do scatter in these vectors


def test_marches_through_the_combined_profile(oracle_mod, golden):
    z, o = load(oracle_mod, golden)
    for r, ic, tau, r_ref, ic_ref, te in zip(z["r_in"], z["ic_in"], z["tau_in"], z["r_out"], z["ic_out"], z["tau_edge"]):
        ro, ico = o.raytrace_to_tau(0, r, ic, float(tau))
        assert np.array_equal(ico, ic_ref) and np.allclose(ro, r_ref, rtol=1e-10, atol=1e-13)
        assert np.isclose(o.raytrace_to_edge(0, r, ic), te, rtol=1e-10, atol=1e-300)


def test_whole_path_with_deuterium(oracle_mod):
    """cosmic D/H = 1.5e-5 at tau0 = 1e4: the D line, 82 km/s to the blue, absorbs a notch out of the blue peak (Dijkstra+06);
    with D/H = 0 the run equals plain Ly-alpha draw for draw except for the species pick"""
    kw = dict(temperature=1e4, taumax=1e4, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=121, velocity_min=-250.0, velocity_max=250.0)
    o0 = oracle_mod.Oracle(**kw); o0.run(4000, seed=4)
    o1 = oracle_mod.Oracle(include_deuterium=True, **kw); o1.run(4000, seed=4)
    assert int(o1.scalar("line_type")) == 7 and abs(o1.scalar("delta_nu_HD_Hz") / 6.7125e11 - 1) < 1e-4
    n0, _ = o0.photon_log(); n1, _ = o1.photon_log()
    assert abs(n1.mean() / n0.mean() - 1) < 0.1
    assert o1.array("Jout").sum() == 4000.0
