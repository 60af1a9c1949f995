"""par%line_id: the single resonance lines (type 1) and S1/2-P1/2,3/2 doublets (type 2) of setup_resonance_line
(line_mod.f90:590-713) -- C IV, N V, O VI, Na I D, Ca II H&K, Mg II, Si IV, Al II.  Constants, calc_voigt2, do_resonance2 and the
marches of the oracle against vectors from the reference's own compiled modules (tools/gen_golden_lines.py)."""
import numpy as np
import pytest

ORDER = ("line_type", "wavelength0", "damping", "cross0", "vtherm1", "g_recoil0", "DnuHK_Hz", "E1", "E2", "E3")


def test_line_constants(oracle_mod, golden):
    z = golden("line_ids.npz")
    for lid, ref in zip(z["ids"], z["constants"]):
        o = oracle_mod.Oracle(temperature=1e4, taumax=10.0, nx=5, ny=5, nz=5, rmax=1.0, line_id=str(lid))
        got = [o.scalar(k) for k in ORDER]
        assert got[0] == ref[0] and np.allclose(got, ref, rtol=1e-14, atol=0), (lid, got, ref)
    # the two restated quirks: C IV recoils with the hydrogen mass, Al II's mass is a single precision literal
    c4 = dict(zip(z["ids"], z["constants"]))["CIV_1548"]
    assert np.isclose(c4[5], (6.62607004e-34 / 1.67262192e-24 / 1.00797) / (0.1548187e-6) ** 2, rtol=1e-12)


@pytest.mark.parametrize("tag", ["mg", "c4"])
def test_doublet_profile_scattering_and_marches(oracle_mod, golden, tag):
    z = golden("line_ids.npz")
    o = oracle_mod.Oracle(str(z[tag + "_params"]))
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD"):
        o.set_array(k, z[tag + "_" + k])
    assert int(o.scalar("line_type")) == 2
    cells = z[tag + "_cells"]
    for c, ref in zip(cells, z[tag + "_calc_voigt"]):
        assert np.allclose(o.calc_voigt(z[tag + "_x"], *map(int, c)), ref, rtol=2e-11, atol=0)
    for i, (x, ref) in enumerate(zip(z[tag + "_xres"], z[tag + "_resonance"])):
        got = o.do_resonance(0, 500 + i, 0, 256, float(x), *map(int, cells[i % 3]))
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-12), (i, x)                       # level pick, velocity, angle, E1..E3 of the level
    for r, ic, tau, r_ref, ic_ref, te in zip(z[tag + "_r_in"], z[tag + "_ic_in"], z[tag + "_tau_in"], z[tag + "_r_out"], z[tag + "_ic_out"],
                                             z[tag + "_tau_edge"]):
        ro, ico = o.raytrace_to_tau(0, r, ic, float(tau))
        assert np.array_equal(ico, ic_ref) and np.allclose(ro, r_ref, rtol=1e-10, atol=1e-13)
        assert np.isclose(o.raytrace_to_edge(0, r, ic), te, rtol=1e-10, atol=1e-300)


def test_whole_path_metal_doublet(oracle_mod):
    """Mg II 2796/2803 through an expanding sphere (examples/MgII-like set-up): both components appear in the escaping spectrum,
    about 2:1 in the optically thin limit of a flat continuum source; an unknown id is refused"""
    kw = dict(temperature=1e4, N_gasmax=1e13, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=200, line_id="MgII_2796", spectral_type="continuum",
              velocity_min=-1200.0, velocity_max=400.0, velocity_type="hubble", Vexp=100.0)
    o = oracle_mod.Oracle(**kw); o.run(20000, seed=3)
    assert int(o.scalar("line_type")) == 2 and abs(o.array("Jout").sum() - 20000.0) < 1.0
    n, _ = o.photon_log()
    assert 0.0 < n.mean() < 10.0
    with pytest.raises(Exception):
        oracle_mod.Oracle(line_id="XeIX_1", temperature=1e4, taumax=1.0, nx=5, ny=5, nz=5, rmax=1.0)
