"""par%fine_structure = .true. (line type 2: the Ly-alpha doublet 2P1/2 + 2P3/2, line_mod.f90:49-62, :141-192, :1250-1253):
the oracle's restatement against vectors from the reference's own compiled modules, and whole-path sanity."""
import numpy as np


def load(oracle_mod, golden):
    z = golden("fine_structure.npz")
    o = oracle_mod.Oracle(str(z["params"]))
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD"):
        o.set_array(k, z[k])
    return z, o


def test_level_splitting_and_profile(oracle_mod, golden):
    z, o = load(oracle_mod, golden)
    assert o.scalar("line_type") == 2 and o.scalar("DnuHK_Hz") == float(z["DnuHK_Hz"])          # exact: same constants, same order
    for c, ref in zip(z["cells"], z["calc_voigt"]):
        assert np.allclose(o.calc_voigt(z["x"], *map(int, c)), ref, rtol=2e-11, atol=0)           # the Voigt-table tolerance


def test_do_resonance2_on_the_reference_stream(oracle_mod, golden):
    z, o = load(oracle_mod, golden)
    for i, (x, ref) in enumerate(zip(z["xres"], z["resonance"])):
        got = o.do_resonance(0, 300 + i, 0, 256, float(x), *map(int, z["cells"][i % 3]))
        # same MT19937-64 stream: the level pick, the velocity, the angle, E1(q_H, q_K) and the matrix elements
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-12), (i, x)
        same = np.sum(np.all(np.isclose(got, ref, rtol=1e-12, atol=1e-14), axis=1))
        assert same >= 250


def test_marches_through_the_doublet_profile(oracle_mod, golden):
    z, o = load(oracle_mod, golden)
    for r, ic, tau, r_ref, ic_ref, te in zip(z["r_in"], z["ic_in"], z["tau_in"], z["r_out"], z["ic_out"], z["tau_edge"]):
        ro, ico = o.raytrace_to_tau(0, r, ic, float(tau))
        assert np.array_equal(ico, ic_ref) and np.allclose(ro, r_ref, rtol=1e-10, atol=1e-13)
        assert np.isclose(o.raytrace_to_edge(0, r, ic), te, rtol=1e-10, atol=1e-300)


def test_whole_path_doublet(oracle_mod):
    """T = 10 K (splitting 3.3 Doppler widths): photons injected 1/3 at the H and 2/3 at the K level; the emergent spectrum is
    the usual double peak around the blend; at 1e4 K (splitting 0.1) the doublet is indistinguishable from the single line"""
    kw = dict(taumax=1e3, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=121, spectral_type="monochromatic")
    o = oracle_mod.Oracle(temperature=10.0, fine_structure=True, **kw); o.run(20000, seed=3)
    J = o.array("Jout"); x = o.xfreq_axis()
    assert J.sum() > 19000 and o.scalar("xfreq_min") < -(o.scalar("xfreq_max") - 1)     # the axis reaches past the H level
    hot1 = oracle_mod.Oracle(temperature=1e4, fine_structure=True, **kw); hot1.run(20000, seed=3)
    hot0 = oracle_mod.Oracle(temperature=1e4, **kw); hot0.run(20000, seed=3)
    n1, _ = hot1.photon_log(); n0, _ = hot0.photon_log()
    assert abs(n1.mean() / n0.mean() - 1) < 0.08
