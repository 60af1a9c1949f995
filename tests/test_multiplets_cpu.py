"""par%line_id of line types 4, 5, 6 -- resonance + fluorescence multiplets (C II 1334, Si II 1527/1260/1193/1304, Fe II 2250/2261/
UV1/UV2/UV3, He I 10833; setup_resonance_line, line_mod.f90:713-1153): every constant and table of line%b(:), calc_voigt1/3,
do_resonance4/5/6 and the marches of the oracle against vectors from the reference's own compiled modules
(tools/gen_golden_multiplets.py), and the product's table (lart_host_line_lookup) against the same vectors."""
import ctypes as C
import numpy as np
import pytest

ORDER = ("line_type", "wavelength0", "damping", "cross0", "vtherm1", "g_recoil0")


def test_every_table_of_the_oracle(oracle_mod, golden):
    z = golden("multiplets.npz")
    for lid, ref, tab in zip(z["ids"], z["constants"], z["tables"]):
        o = oracle_mod.Oracle(temperature=1e4, taumax=10.0, nx=5, ny=5, nz=5, rmax=1.0, line_id=str(lid))
        assert np.array_equal([o.scalar(k) for k in ORDER], ref[:6]), lid
        assert np.array_equal(o.array("line_multiplet"), tab), lid               # bit for bit, the alias tables included


def test_every_table_of_the_product(golden):
    from lart_amd import capi
    z = golden("multiplets.npz")
    for lid, ref, tab in zip(z["ids"], z["constants"], z["tables"]):
        m = capi.line_lookup(str(lid))
        assert m is not None and [m.line_type, m.wavelength0, m.damping, m.cross0, m.vtherm1, m.g_recoil0] == list(ref[:6]), lid
        assert np.array_equal(capi.pack_multiplet(m), tab), lid
    assert capi.line_lookup("ly_alpha") is None and capi.line_lookup("XeIX_1") is None


@pytest.mark.parametrize("tag,typ", [("si", 4), ("fe", 5), ("he", 6)])
def test_profile_scattering_and_marches(oracle_mod, golden, tag, typ):
    z = golden("multiplets.npz")
    o = oracle_mod.Oracle(str(z[tag + "_params"]))
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD"):
        o.set_array(k, z[tag + "_" + k])
    assert int(o.scalar("line_type")) == typ
    cells = z[tag + "_cells"]
    for c, ref in zip(cells, z[tag + "_calc_voigt"]):
        assert np.allclose(o.calc_voigt(z[tag + "_x"], *map(int, c)), ref, rtol=2e-11, atol=0)
    nfluor = 0
    for i, (x, ref) in enumerate(zip(z[tag + "_xres"], z[tag + "_resonance"])):
        got = o.do_resonance(0, 700 + i, 0, 256, float(x), *map(int, cells[i % 3]))
        # the same MT19937-64 stream: level pick, velocity, downward channel (fluorescence shift of xatom), angle, matrix elements, E1..E3
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-12), (i, x)
        nfluor += int((np.abs(ref[:, 1] - (x - ref[:, 0])) > 1.0).sum())
    assert (nfluor > 100) == (typ != 6)                                           # He I 10833 has no fluorescent channel
    for r, ic, tau, r_ref, ic_ref, te in zip(z[tag + "_r_in"], z[tag + "_ic_in"], z[tag + "_tau_in"], z[tag + "_r_out"], z[tag + "_ic_out"],
                                             z[tag + "_tau_edge"]):
        ro, ico = o.raytrace_to_tau(0, r, ic, float(tau))
        assert np.array_equal(ico, ic_ref) and np.allclose(ro, r_ref, rtol=1e-10, atol=1e-13)
        assert np.isclose(o.raytrace_to_edge(0, r, ic), te, rtol=1e-10, atol=1e-300)


def test_whole_path_fluorescence(oracle_mod):
    """Si II 1527 -> 1533 (examples/SiII_1527): with a Voigt line source about A21(2)/(A21(1)+A21(2)) = 66 % of the scatterings
    send the photon to the fluorescent line 287.24 cm^-1 to the red, where the medium is transparent"""
    kw = dict(temperature=1e5, taumax=10.0, nx=21, ny=21, nz=21, rmax=1.0, line_id="SiII_1527", nxfreq=400, wavelength_min=1526.0, wavelength_max=1536.0)
    o = oracle_mod.Oracle(**kw); o.run(20000, seed=5)
    J = o.array("Jout"); lam = np.linspace(1536.0, 1526.0, 400)                     # the frequency axis runs blue-wards
    assert abs(J.sum() - 20000.0) < 1.0
    red = J[lam > 1532.0].sum() / J.sum()
    assert 0.55 < red < 1.0


def test_HeI_coherent_on_the_reference_stream(oracle_mod, golden):
    """par%HeI_coherent: E1..E3 as the polynomial of the atom-frame offsets from the three 2^3P levels (compute_HeI_E_coherent)"""
    z = golden("multiplets.npz")
    o = oracle_mod.Oracle(str(z["he_params"]) + " HeI_coherent=1")
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD"):
        o.set_array(k, z["he_" + k])
    ref_all = z["he_coherent_resonance"]
    assert np.ptp(ref_all[..., 9]) > 0.3                                          # E1 now varies from event to event
    for i, (x, ref) in enumerate(zip(z["he_xres"], ref_all)):
        got = o.do_resonance(0, 800 + i, 0, 256, float(x), *map(int, z["he_cells"][i % 3]))
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-12), (i, x)
