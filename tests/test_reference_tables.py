"""Known-answer tables the reference ships with its examples (tests/golden/reference_tables.json, collected by
tools/gen_reference_tables.py), reproduced through the Fortran host LaRT_amd.x on the GPU:

  examples/HeI_coherent_test/results_summary.txt:7-37   20 runs: <N_scat>, peak Jout, sum Jout, img.sum  -- He I 10833 (line type 6,
      par%HeI_coherent on and off), Stokes resonance scattering, point and uniform-sphere sources, 101 x 101 x 201 peel cube.  These
      are reference-held numbers for output_normalize (sum Jout = 1 / (dxfreq 2 pi 4 pi rmax^2), output_sum_rect.f90:194-202) and
      for the peel-cube normalisation (output_sum_rect.f90:428-430: img.sum = sum of the `Scattered` cube)
  examples/HeI_sphere/Nscatt_sphere_uniform_voigt.txt:1-18   (T, tau0, <N_scatt>) of the uniform He I spheres
  examples/FeII_test/out.txt   'Average Number of scattering' and the grid lines of nine Fe II UV1/UV2/UV3 runs (line types 4 / 5)

The inputs are the shipped namelists (field by field; only par%no_photons changes -- and par%iseed, which the examples leave at
"seed from the clock").  CPU part: the stored namelists equal the shipped files where /root/reference is mounted, and the set-up
lines of the logs are reproduced by --dry-run."""
import json
import os
import re
import numpy as np
import pytest
from test_fortran_host import EXE, read_h5, run

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAB = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))
REF = "/root/reference/examples"


@pytest.fixture(scope="module")
def exe():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(EXE)
    return EXE


def namelist(fields, **over):
    f = dict(fields); f.update({k: str(v) for k, v in over.items()})
    return "&parameters\n" + "".join(" par%%%s = %s\n" % kv for kv in f.items()) + "/\n"


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"par%([A-Za-z0-9_()]+)\s*=\s*(.+)$", line.split("!")[0].strip())
        if m:
            out[m.group(1)] = m.group(2).strip()
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only mounted in the build container")
def test_stored_namelists_equal_the_shipped_inputs():
    for r in TAB["HeI_coherent_test"]["rows"]:
        assert r["namelist"] == parse(os.path.join(REF, "HeI_coherent_test", r["case"] + ".in")), r["case"]
    for r in TAB["FeII_test"]["rows"]:
        assert r["namelist"] == parse(os.path.join(REF, "FeII_test", r["case"] + ".in")), r["case"]
    for k in range(4):                                        # the four shipped members of the HeI_sphere family
        f = parse(os.path.join(REF, "HeI_sphere", "t4tau%d.in" % k))
        assert float(f["temperature"]) == 1e4 and float(f["taumax"]) == 10.0 ** k
        for key, v in TAB["HeI_sphere"]["namelist_common"].items():
            assert f[key] == v, key
    want = [tuple(float(x) for x in l.split()) for l in open(os.path.join(REF, "HeI_sphere", "Nscatt_sphere_uniform_voigt.txt"))]
    assert want == [(r["temperature"], r["taumax"], r["Nscatt"]) for r in TAB["HeI_sphere"]["rows"]]


def test_feii_set_up_lines_of_the_shipped_log(exe, tmp_path):
    """examples/FeII_test/out.txt (a log of 2024/09/21): the set-up lines LaRT_amd.x prints for FeII_UV1.in (--dry-run, no GPU).
    The log predates v2.00's Fe II table: it shows voigt_a 2.226E-03 and N(Fe II)_pole 1.176E+14 where line_mod.f90:998-1052 of the
    reference as shipped (A21 = 2.35e8 + 3.52e7, f12 = 0.239; the product's table is pinned on it bit for bit,
    tests/test_multiplets_cpu.py) gives a = 2.702e8 / (4 pi Dfreq) = 1.025E-03 and 1.174E+14.  Asserted: what both agree on, and
    the v2.00 values."""
    row = TAB["FeII_test"]["rows"][0]
    assert row["case"] == "FeII_UV1"
    r = run(exe, namelist(row["namelist"], no_photons="1e3"), tmp_path, "--dry-run", name="FeII_UV1.in")
    assert r.returncode == 0, r.stdout + r.stderr
    for line in ("Line ID                   : FeII_UV1", "temperature (K)  :        1.000E+05", "tau_pole  (Fe II) :        2.000E+01",
                 "tau_homo  (Fe II) :        2.000E+01", "Dust-to-Gas Ratio         :    0.000E+00", "Total number of photons   :    1.000E+03"):
        assert line in r.stdout, line
    val = lambda key: float([l for l in r.stdout.splitlines() if l.startswith(key)][0].split(":")[1].split()[0])
    assert abs(val("N(Fe II)_pole") / 1.176e14 - 1.0) < 3e-3          # the log's column density (0.2 % from the newer f-value)
    vth = 0.12895319011972164 / np.sqrt(55.847) * np.sqrt(1e5)           # km/s: vtherm1_amu / sqrt(mass_amu) sqrt(T)
    a_v2 = (2.35e8 + 3.52e7) / (4.0 * np.pi * vth / (0.260017206 * 1e-9))
    assert abs(val("voigt_a") / a_v2 - 1.0) < 1e-3


def nscat_of(stdout):
    return float(re.search(r"Average Number of scattering :\s+(\S+)", stdout).group(1))


@pytest.mark.gpu
@pytest.mark.parametrize("row", TAB["HeI_coherent_test"]["rows"], ids=lambda r: r["case"])
def test_hei_coherent_summary(exe, tmp_path, row):
    """results_summary.txt:7-37, one row.  2e6 photons here against 5e6 there; tolerances = the Monte-Carlo error of the two runs:
    <N_scat> is a mean over photons (relative error ~ 1.2 / sqrt(N)), sum Jout is exact photon conservation (4 digits, as printed),
    peak Jout is one of 201 bins (counting error of the bin), img.sum a peel-off estimator summed over the cube."""
    n = 2.0e6
    name = row["case"]
    r = run(exe, namelist(row["namelist"], no_photons="%.1e" % n, iseed=1234), tmp_path, name=name + ".in")
    assert r.returncode == 0, r.stdout + r.stderr
    S = read_h5(str(tmp_path / (name + ".h5")))["Spectrum"]
    jout = S[1]["Jout"]
    nsc = float(S[0]["Nsc_tot"])
    assert abs(nsc - nscat_of(r.stdout)) < 1e-3 * max(nsc, 1e-3)
    tol_n = 0.005 * row["Nscat"] + 0.006                     # the table prints 2 decimals; 0.5 % covers 4 sigma of both runs
    assert abs(nsc - row["Nscat"]) < tol_n, (nsc, row["Nscat"])
    assert abs(jout.sum() / row["sum_Jout"] - 1.0) < 2e-4, (jout.sum(), row["sum_Jout"])
    # peak bin: counts ~ peak/sum * N photons in it
    cnt = jout.max() / jout.sum() * n
    assert abs(jout.max() / row["peak_Jout"] - 1.0) < 4.0 * np.sqrt(1.0 / cnt + 1.0 / (cnt * 2.5)) + 2e-3, (jout.max(), row["peak_Jout"])
    O = read_h5(str(tmp_path / (name + "_obs.h5")))
    img = O["Scattered"][1].sum()
    assert abs(img / row["img_sum"] - 1.0) < 0.01, (img, row["img_sum"])
    K = read_h5(str(tmp_path / (name + "_stokes.h5")))
    I, Q, U = K["Stokes_I"][1], K["Stokes_Q"][1], K["Stokes_U"][1]
    # sanity check 4 of the summary: image-integrated polarisation ~ 0 for a sphere seen from afar
    assert abs(Q.sum()) < 0.01 * I.sum() and abs(U.sum()) < 0.01 * I.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("row", TAB["HeI_sphere"]["rows"], ids=lambda r: "T%g_tau%g" % (r["temperature"], r["taumax"]))
def test_hei_sphere_nscatt_table(exe, tmp_path, row):
    """Nscatt_sphere_uniform_voigt.txt: <N_scatt>(T, tau0) of a uniform He I 10833 sphere with a uniform source"""
    tau = row["taumax"]
    n = 1e6 if tau <= 1e3 else (3e5 if tau <= 1e4 else 1e5)
    f = dict(TAB["HeI_sphere"]["namelist_common"])
    text = namelist(f, no_photons="%.1e" % n, temperature="%.5e" % row["temperature"], taumax="%.5e" % tau, iseed=99)
    r = run(exe, text, tmp_path, name="hei.in")
    assert r.returncode == 0, r.stdout + r.stderr
    nsc = nscat_of(r.stdout)
    # scatter of N_scatt per photon ~ its mean (thin: Poisson-like, thick: broad), so 4 sigma ~ 4 * 1.3 / sqrt(n); + the 4 digits printed
    tol = (5.2 / np.sqrt(n) + 2e-3) * row["Nscatt"]
    assert abs(nsc - row["Nscatt"]) < tol, (nsc, row["Nscatt"], tol)


@pytest.mark.gpu
@pytest.mark.parametrize("row", TAB["FeII_test"]["rows"], ids=lambda r: r["case"])
def test_feii_log_nscatt(exe, tmp_path, row):
    """examples/FeII_test/out.txt: 'Average Number of scattering' of the nine Fe II runs (1e8 photons there, 4e6 here)"""
    n = 4e6
    r = run(exe, namelist(row["namelist"], no_photons="%.1e" % n, iseed=7), tmp_path, name=row["case"] + ".in")
    assert r.returncode == 0, r.stdout + r.stderr
    nsc = nscat_of(r.stdout)
    # first scattering is forced (weight 1 - exp(-tau)), the rest is geometric-like: var ~ e (1 + e) with e = <N> - 1; 4 sigma with a
    # factor 1.5 of slack, plus the last printed digit of the log and the 1e8-photon error of the log itself
    e = row["Nscat"] - 1.0
    assert abs(nsc - row["Nscat"]) < 6.0 * np.sqrt(e * (1.0 + e) / n) + 1.2e-4, (nsc, row["Nscat"])
