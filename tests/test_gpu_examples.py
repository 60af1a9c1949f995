"""The parameter combinations of the reference's shipped inputs (tests/golden/reference_example_params.json, written by
tools/gen_example_params.py from examples/*/*.in) on the device, at reduced optical depth, grid and photon number: every accepted input
must set up, run to its last photon and conserve photons.  Property checks, not parity: every third set (72 of the 214, spread over the example
directories) runs in the default device suite, the rest carry the `slow` marker (pytest -m "gpu and slow")."""
import json
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SETS = json.load(open(os.path.join(HERE, "golden", "reference_example_params.json")))
NAMES = sorted(SETS)


def reduced(d, tmp):
    kw = dict(d)
    kw["no_photons"] = 2000
    for k in ("nx", "ny", "nz"):
        if kw.get(k, 0) > 41:
            kw[k] = 41 if kw[k] % 2 else 40
    for k in ("nxim", "nyim"):
        if kw.get(k, 0) > 33:
            kw[k] = 33
    if kw.get("nside", 0) > 8:
        kw["nside"] = 8
    for k in ("nprint", "no_print", "num_send_at_once", "out_merge", "save_sightline_tau", "save_clump_info", "out_file", "nphotons"):
        kw.pop(k, None)
    if kw.get("scatt_mat_file") == "MUELLER":
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
        from conftest_data import mueller_path
        kw["scatt_mat_file"] = mueller_path(tmp)
    if kw.get("use_clump_medium") and kw.get("clump_radius", 1.0) < 0.01:
        kw["clump_radius"] = 0.01                                  # 1e6-1e8 clumps in the shipped inputs: the population, not its size, is what the path sees
    return kw


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from lart_amd import capi
    return capi


@pytest.mark.parametrize("name", [pytest.param(n, marks=() if i % 3 == 0 else pytest.mark.slow) for i, n in enumerate(NAMES)])
def test_shipped_parameter_set_runs(capi, tmp_path, name):
    from lart_amd import host
    kw = reduced(SETS[name], str(tmp_path))
    sim = host.Simulation(dict(kw))
    if sim.par.taumax > 300.0:                                     # tau0 up to 1e7 in the shipped inputs: one packet would scatter 1e7 times
        for k in ("N_HI", "N_HImax", "N_HIhomo", "tauhomo", "clump_NHI", "clump_nH", "clump_tau0"):
            kw.pop(k, None)
        if kw.get("use_clump_medium"):
            kw["clump_tau0"] = 30.0
        else:
            kw["taumax"] = 300.0
        sim.ctx.close()
        sim = host.Simulation(dict(kw))
    sim.run_simulation(2000)
    t = sim.ctx.tallies()
    assert t["nphotons_done"] == 2000
    tot = t["Jout"].sum() + (t["Jabs"].sum() if "Jabs" in t and t["Jabs"] is not None else 0.0)
    assert np.isfinite(tot) and tot <= 2000 * (1 + 1e-9)
    out = sim.output_normalize(t)
    assert np.all(np.isfinite(out["Jout"]))
    for k in ("scatt", "I", "Q", "J1", "P1"):
        if k in out and out[k] is not None:
            assert np.all(np.isfinite(out[k])), k
    sim.ctx.close()
