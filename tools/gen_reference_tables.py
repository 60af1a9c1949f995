#!/usr/bin/env python3
"""Collect the known-answer tables the reference ships with its examples into tests/golden/reference_tables.json (data only):

  examples/HeI_coherent_test/results_summary.txt:7-37   20 runs (He I 10833, Stokes, 101x101 peel): <N_scat>, peak Jout, sum Jout, img.sum
  examples/HeI_sphere/Nscatt_sphere_uniform_voigt.txt    18 rows (T, tau0, <N_scatt>) of the uniform-sphere He I 10833 models (mk_models.py)
  examples/FeII_test/out.txt                             9 runs (Fe II UV1/UV2/UV3 multiplets, continuum source): 'Average Number of scattering'

For every case the namelist assignments of the shipped input file are stored as {name: value-text} so that the tests can write
the same input (only par%no_photons and the output name change) on a box where /root/reference does not exist.
Runs in the container that mounts /root/reference:   python tools/gen_reference_tables.py
"""
import json
import os
import re

REF = "/root/reference/examples"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_namelist(path):
    """{field: value text} of the par%... assignments (comments and blank lines dropped; order kept)"""
    out = {}
    for line in open(path):
        line = line.split("!")[0].strip()
        m = re.match(r"par%([A-Za-z0-9_()]+)\s*=\s*(.+)$", line)
        if m:
            out[m.group(1)] = m.group(2).strip()
    return out


def main():
    tab = {"source": "seoncafe/LaRT v2.00 examples (known-answer tables and shipped logs)"}
    # ---- HeI_coherent_test
    rows = []
    for line in open(os.path.join(REF, "HeI_coherent_test", "results_summary.txt")):
        m = re.match(r"\s*((?:pt|un)_tau[0-9.]+_(?:inc|coh))\s+([0-9.eE+-]+)\s+([0-9.eE+-]+)\s+([0-9.eE+-]+)\s+([0-9.eE+-]+)\s*$", line)
        if m:
            name = m.group(1)
            rows.append({"case": name, "Nscat": float(m.group(2)), "peak_Jout": float(m.group(3)), "sum_Jout": float(m.group(4)), "img_sum": float(m.group(5)),
                         "namelist": parse_namelist(os.path.join(REF, "HeI_coherent_test", name + ".in"))})
    assert len(rows) == 20
    tab["HeI_coherent_test"] = {"cite": "examples/HeI_coherent_test/results_summary.txt:7-37 (5e6 photons per case, 64 MPI ranks)", "rows": rows}
    # ---- HeI_sphere: the table was written from t{1,4}tau{-3..5} models of mk_models.py; t4tau0..3.in are shipped, the others differ in
    # temperature / taumax / no_photons only (mk_models.py make_input)
    base = parse_namelist(os.path.join(REF, "HeI_sphere", "t4tau2.in"))
    rows = []
    for line in open(os.path.join(REF, "HeI_sphere", "Nscatt_sphere_uniform_voigt.txt")):
        t, tau, n = line.split()
        rows.append({"temperature": float(t), "taumax": float(tau), "Nscatt": float(n)})
    assert len(rows) == 18
    for k in ("no_photons", "temperature", "taumax", "num_send_at_once"):
        base.pop(k, None)
    tab["HeI_sphere"] = {"cite": "examples/HeI_sphere/Nscatt_sphere_uniform_voigt.txt:1-18 (NSC_GAS header keyword of each run, write_nscatt2.py)",
                         "namelist_common": base, "rows": rows}
    # ---- FeII_test
    rows, name = [], None
    for line in open(os.path.join(REF, "FeII_test", "out.txt")):
        m = re.match(r"\+\+\+\+\+ (\S+)\.in \+\+\+\+\+", line)
        if m:
            name = m.group(1)
        m = re.match(r"Average Number of scattering :\s+([0-9.eE+-]+)", line)
        if m and name:
            rows.append({"case": name, "Nscat": float(m.group(1)), "namelist": parse_namelist(os.path.join(REF, "FeII_test", name + ".in"))}); name = None
    assert len(rows) == 9
    tab["FeII_test"] = {"cite": "examples/FeII_test/out.txt ('Average Number of scattering', 1e8 photons per case)", "rows": rows}
    # ---- SiII_1527 (log of 2023/07/25) and SiII_1193 (2022/06/06): 'Average Number of scattering' of every run of the two logs.  Both logs are older than v2.00 and count
    # every scattering as 1 (v2.00 adds the packet weight, scattering_car.f90:690): the comparable quantity of a v2.00 run is the UNWEIGHTED number of scattering events per photon
    for ex in ("SiII_1527", "SiII_1193"):
        rows, name, setup = [], None, {}
        for line in open(os.path.join(REF, ex, "out.txt")):
            m = re.match(r"\+\+\+\+\+ (\S+)\.in \+\+\+\+\+", line)
            if m:
                name = m.group(1); setup = {}
            m = re.match(r"(voigt_a|temperature \(K\)|N\(Si II\)_pole|tau_pole  \(Si II\))\s*:\s+([0-9.eE+-]+)", line)
            if m:
                setup[m.group(1)] = float(m.group(2))
            m = re.match(r"Average Number of scattering :\s+([0-9.eE+-]+)", line)
            if m and name:
                rows.append({"case": name, "Nscat_unweighted": float(m.group(1)), "setup": setup, "namelist": parse_namelist(os.path.join(REF, ex, name + ".in"))}); name = None
        tab[ex] = {"cite": "examples/%s/out.txt ('Average Number of scattering' of every run; logs older than v2.00: one count per scattering, not the packet weight)" % ex, "rows": rows}
    dst = os.path.join(ROOT, "tests", "golden", "reference_tables.json")
    json.dump(tab, open(dst, "w"), indent=1)
    print("wrote", dst, {k: len(v["rows"]) for k, v in tab.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
