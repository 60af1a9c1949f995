"""The oracle's restatement of the Ly-beta (line type 8) and H2 pieces against vectors from the reference's own compiled modules
(tests/golden/lyb_h2.npz, tools/gen_golden_lyb_h2.py: line_mod.f90:293-345, 1192-1240; raytrace_car.f90:510-582, 1425-1648; h2_mod.f90)."""
import numpy as np
import pytest


def test_ly_beta_line_constants_and_do_resonance8(oracle_mod, golden):
    z = golden("lyb_h2.npz")
    o = oracle_mod.Oracle(str(z["lyb_params"]))
    lc = z["lyb_line"]
    assert int(lc[0]) == 8 == int(o.scalar("line_type"))
    for k, i in (("wavelength0", 1), ("damping", 2), ("cross0", 3), ("vtherm1", 4), ("g_recoil0", 5)):
        assert o.scalar(k) == lc[i], k                                       # exact: the same literals, the same expressions
    assert o.scalar("P_conv") == z["lyb_multiplet"][14 + 1 * 7 + 5]          # line%b(1)%P_down(2)
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD"):
        o.set_array(k, z["lyb_" + k])
    ic, jc, kc = [int(v) for v in z["res8_cell"]]
    for x, ref in zip(z["res8_x"], z["res8"]):
        r = o.do_resonance(0, int(z["res8_seed"]), 0, len(ref), float(x), ic, jc, kc)       # the reference's MT19937-64 stream
        assert np.array_equal(r[:, 4], ref[:, 7])                            # band after the event: the same conversions
        assert np.allclose(r[:, 0:4], ref[:, 0:4], rtol=1e-12, atol=1e-14)    # uz, atom-frame frequency, cos, sin
        assert np.array_equal(r[:, 9:12], ref[:, 4:7])                       # E1..E3 of the channel
        assert 0.08 < np.mean(ref[:, 7] == 2) < 0.16


def test_ly_beta_band_marches_vs_reference(oracle_mod, golden):
    z = golden("lyb_h2.npz")
    o = oracle_mod.Oracle(str(z["lyb_params"]))
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD"):
        o.set_array(k, z["lyb_" + k])
    for r, ic, tau, ib, r_ref, ic_ref, te in zip(z["lyb_r_in"], z["lyb_ic_in"], z["lyb_tau_in"], z["lyb_band"], z["lyb_r_out"], z["lyb_ic_out"], z["lyb_tau_edge_Ha"]):
        ro, ico = o.raytrace_to_tau_band(int(ib), r, ic, float(tau))
        assert np.array_equal(ico, ic_ref)
        assert np.allclose(ro, r_ref, rtol=1e-10, atol=1e-13)                  # band 2: the frequency never changes, the opacity is rhokapD R_Ha
        assert np.isclose(o.raytrace_to_edge_dust_Ha(r, ic), te, rtol=1e-12, atol=1e-300)
    assert np.array_equal(o.array("Jout_Ha"), z["lyb_Jout_Ha"]) and np.array_equal(o.array("Jout"), z["lyb_Jout"])
    W = o.array("W")
    assert W[1] == z["lyb_W_esc"][0] and W[3] == z["lyb_W_esc"][1]            # W_esc1, W_esc2


def test_h2_lines_vs_reference(oracle_mod, golden, h2_data_dir):
    z = golden("lyb_h2.npz")
    for (T, f, hiw, pure, bt), ref in zip(z["h2_cases"], z["h2_init"]):
        kw = dict(temperature=1e3, taumax=10.0, nx=3, ny=3, nz=3, rmax=1.0, h2_model="neufeld", f_H2=float(f), h2_temperature=float(T), h2_data_dir=h2_data_dir,
                  h2_hi_width=bool(hiw), h2_pure_absorption=bool(pure))
        if bt > 0:
            kw["bturb"] = float(bt)
        o = oracle_mod.Oracle(**kw)
        assert np.allclose(o.array("h2_lines")[:14], ref[:14], rtol=1e-13, atol=0), T       # n, Dfreq, then dnu, strength, a_damp, p_scat, pop, f_osc per line
    o = oracle_mod.Oracle(str(z["h2_params"]).replace("/root/reference/data/h2", h2_data_dir))
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz", "rhokapD"):
        o.set_array(k, z["h2_" + k])
    Df = float(z["h2_Dfreq_cell"])
    assert np.allclose(o.h2_kappa(z["h2_x"], Df), z["h2_kappa"], rtol=1e-10, atol=1e-300)
    assert np.array_equal(o.h2_select(0, int(z["h2_sel_seed"]), 0, z["h2_xsel"], Df), z["h2_sel"])
    for r, ic, tau, r_ref, ic_ref, te in zip(z["h2_r_in"], z["h2_ic_in"], z["h2_tau_in"], z["h2_r_out"], z["h2_ic_out"], z["h2_tau_edge"]):
        ro, ico = o.raytrace_to_tau(0, r, ic, float(tau))
        assert np.array_equal(ico, ic_ref)
        assert np.allclose(ro, r_ref, rtol=1e-10, atol=1e-13)
        assert np.isclose(o.raytrace_to_edge(0, r, ic), te, rtol=1e-10, atol=1e-300)
    assert np.array_equal(o.array("Jout"), z["h2_Jout"])


def test_ly_beta_weight_budget_and_conversion_rate(oracle_mod):
    """output_sum_rect.f90:452-467: every Ly-beta packet ends escaped, absorbed or converted, every converted one escaped or absorbed in band 2;
    the conversion-rate profile is P_down(2) = 0.11834 of the scattering-rate profile (scattering_car.f90:865)"""
    o = oracle_mod.Oracle(line_id="ly_beta", temperature=1e4, taumax=3e1, nx=21, ny=21, nz=21, rmax=1.0, nxfreq=81, DGR=1.0, cext_dust=1e-15, cext_dust_Ha=5e-16, nxim=8, nyim=8,
                          calc_P=True)
    n = 40000
    o.run(n, seed=3)
    W = o.array("W")
    assert abs((W[0] + W[1] + W[2]) / n - 1.0) < 1e-12 and abs((W[3] + W[4]) - W[0]) < 1e-9 * n
    assert np.isclose(o.array("Jout_Ha").sum(), W[3], rtol=1e-3) and np.isclose(o.array("Jabs_Ha").sum(), W[4], rtol=1e-12)
    P1, Pc = o.array("P1"), o.array("Pconv")
    assert abs(Pc.sum() / P1.sum() - o.scalar("P_conv")) < 0.01
    assert o.array("peel_Ha:0").sum() > 0.0 and o.array("scatt:0").sum() > 0.0


def test_h2_vetoes(oracle_mod, h2_data_dir):
    with pytest.raises(RuntimeError, match="H2"):
        oracle_mod.Oracle(temperature=1e3, taumax=10.0, nx=3, ny=3, nz=3, rmax=1.0, h2_model="neufeld", f_H2=0.1, use_stokes=True, h2_data_dir=h2_data_dir)
    with pytest.raises(RuntimeError, match="ly_beta"):
        oracle_mod.Oracle(line_id="ly_beta", temperature=1e3, taumax=10.0, nx=3, ny=3, nz=3, rmax=1.0, xyz_symmetry=True)
