"""Clump medium on the CPU side: the oracle's restatement of clump_mod.f90 / raytrace_clump.f90 against the deterministic
numbers of the reference's shipped log (examples/clump_sphere/log_back) and geometric invariants, and the host mirror
(lart_amd/clump.py + csrc/lart_host_clumps.cpp) against the oracle, table for table.  No GPU.

The clump modules of the reference `use mpi`, so oracle/_ref cannot hold them: this part of the oracle is
pinned by the log's known answers, not by compiled reference code."""
import numpy as np
import pytest

# examples/clump_sphere/clump_NHI18_fcov1.in as run for log_back:5-52 (that run printed no rmin line: the shell started at 0)
NHI18 = dict(use_clump_medium=True, rmax=1.0, clump_radius=0.001, clump_f_cov=1.0, N_HImax=1e18, temperature=1e4, clump_fully_inside=False,
             spectral_type="monochromatic", velocity_type="rotating_galaxy_halo", Vrot=300.0, rinner=0.1, nxfreq=500,
             velocity_min=-1000.0, velocity_max=1000.0, nx=11, ny=11, nz=11, iseed=1)
C4 = dict(use_clump_medium=True, rmax=1.0, clump_radius=0.01, clump_f_cov=5.0, clump_tau0=1e3, temperature=1e4, nxfreq=301,
          velocity_min=-100.0, velocity_max=100.0, nx=11, ny=11, nz=11, iseed=7)


def host_clumps(kw):
    from lart_amd import host, clump
    par = host.Params()
    for k, v in kw.items():
        host.set_param(par, k, v)
    par, line, _ = host.read_input(par)
    cl, g = clump.grid_create_clump(par, line)
    return par, line, cl, g


def test_reference_log_known_answers(oracle_mod):
    o = oracle_mod.Oracle(**NHI18)
    assert int(o.scalar("cl_N")) == 1333333                                   # "Clumps: N_clumps  =        1333333"
    assert abs(o.scalar("cl_rhokap_ref") / 4.4261e7 - 1) < 2e-5               # "Clumps: cl_rhokap =   4.4261E+07"
    assert abs(o.scalar("cl_voigt_a_ref") - 0.00047) < 5e-6                   # "Clumps: voigt_a   =      0.00047"
    assert abs(o.scalar("cl_Dfreq_ref") / 1.0566e11 - 1) < 5e-5               # "Clumps: cl_Dfreq  =   1.0566E+11"
    assert abs(o.scalar("cl_fvol") - 0.001333) < 1e-6 and abs(o.scalar("cl_fcov") - 1.0) < 1e-5
    assert abs(o.scalar("tauhomo") / 5.89826e4 - 1) < 2e-6                    # "Clump derived: tauhomo  =    5.89826E+04"
    assert abs(o.scalar("N_gashomo") / 1.0e18 - 1) < 1e-9                     # "Clump derived: N_gashomo=    1.00000E+18"
    assert abs(o.scalar("cl_rsa_acceptance") - 0.995) < 0.002                 # "RSA done, acceptance rate =   99.5%"
    assert int(o.scalar("cl_cgx")) == 111                                     # "registrations in   111^3 cells"
    assert abs(o.scalar("cl_nreg") / 1827254 - 1) < 2e-3                      # another realisation of the same statistics


@pytest.mark.parametrize("extra", [dict(), dict(clump_sigma_v=20.0, velocity_type="hubble", Vexp=50.0), dict(DGR=1.0, rmin=0.2, clump_f_vol=0.03, clump_f_cov=-1.0)])
def test_host_population_equals_oracle(oracle_mod, extra):
    kw = dict(C4, **extra)
    o = oracle_mod.Oracle(**kw)
    par, line, cl, g = host_clumps(kw)
    assert cl.N == int(o.scalar("cl_N")) and cl.cgx == int(o.scalar("cl_cgx")) and len(cl.cg_list) == int(o.scalar("cl_nreg"))
    for k in ("x", "y", "z", "vx", "vy", "vz", "radius", "rhokap", "voigt_a", "Dfreq"):
        assert np.array_equal(o.array("cl_" + k), getattr(cl, k)), k                # the same MT19937-64 stream, the same placement: bit-exact
    if kw.get("DGR", 0) > 0:
        assert np.allclose(o.array("cl_rhokapD"), cl.rhokapD, rtol=1e-15)
    assert np.array_equal(o.array("cg_start"), cl.cg_start) and np.array_equal(o.array("cg_list"), cl.cg_list)
    for k in ("tauhomo", "taumax", "N_gashomo", "N_gasmax"):
        assert np.isclose(o.scalar(k), getattr(par, k), rtol=1e-13), k
    assert np.isclose(g.xfreq_min, o.scalar("xfreq_min")) and np.isclose(g.dxfreq, o.scalar("dxfreq")) and g.Dfreq_ref == o.scalar("Dfreq_ref")
    # non-overlap (random sequential adsorption) and containment of the centres
    r = np.sqrt(cl.x ** 2 + cl.y ** 2 + cl.z ** 2)
    assert r.max() <= 1.0 - 0.01 + 1e-12 and r.min() >= kw.get("rmin", 0.0) + (0.01 if kw.get("rmin") else 0.0) - 1e-12
    from scipy.spatial import cKDTree
    d, _ = cKDTree(np.column_stack([cl.x, cl.y, cl.z])).query(np.column_stack([cl.x, cl.y, cl.z]), k=2)
    assert d[:, 1].min() >= 2 * 0.01


def test_ray_march_invariants(oracle_mod):
    o = oracle_mod.Oracle(**C4)
    N = int(o.scalar("cl_N")); x, y, z = o.array("cl_x"), o.array("cl_y"), o.array("cl_z")
    rng = np.random.default_rng(5)
    taus = []
    for _ in range(200):
        k = rng.normal(size=3); k /= np.linalg.norm(k)
        xf = rng.uniform(-3, 3)
        r = np.array([0.0, 0.0, 0.0, *k, xf, 1.0]); ic = np.array([0, 1, 1, 1], dtype=np.int32)
        if rng.random() < 0.3:                                     # start inside a clump
            j = rng.integers(0, N); pos = np.array([x[j], y[j], z[j]]) + rng.uniform(-0.004, 0.004, 3)
            r[:3] = pos; ic[0] = j + 1
        tau_edge = o.raytrace_to_edge(4, r, ic)
        taus.append(tau_edge)
        # brute force: sum over all clumps of chord x kappa (static clumps: no frequency shifts)
        rx, ry, rz = r[0] - x, r[1] - y, r[2] - z
        b = rx * k[0] + ry * k[1] + rz * k[2]; disc = b * b - (rx * rx + ry * ry + rz * rz) + 1e-4
        hit = disc > 0
        t1 = np.maximum(-b[hit] - np.sqrt(disc[hit]), 0.0); t2 = -b[hit] + np.sqrt(disc[hit])
        chord = np.clip(t2, 0, None) - t1; chord = chord[t2 > 0]
        kap = o.scalar("cl_rhokap_ref") * oracle_mod.voigt(np.array([xf]), o.scalar("cl_voigt_a_ref"))[0]
        assert np.isclose(tau_edge, kap * chord.sum(), rtol=1e-9, atol=1e-9)
        assert np.isclose(min(tau_edge, 745.2), min(o.raytrace_to_edge(14, r, ic), 1e300) if tau_edge < 745.2 else 745.2, rtol=1e-12) or tau_edge >= 745.2
        # to_tau lands at the requested depth: the remaining depth to the edge is the difference
        if tau_edge > 0:
            tq = tau_edge * rng.uniform(0.1, 0.9)
            r2, ic2 = o.raytrace_to_tau(4, r, ic, tq)
            assert ic2[3] == 1 and ic2[0] >= 1
            assert np.isclose(o.raytrace_to_edge(4, r2, ic2), tau_edge - tq, rtol=1e-8, atol=1e-8)
        r3, ic3 = o.raytrace_to_tau(4, r, ic, tau_edge * 1.0001 + 1e-6)
        assert ic3[3] == 0 and abs(np.linalg.norm(r3[:3]) - 1.0) < 1e-9                   # escapes on the sphere
    # mean line-centre depth from the centre ~ tauhomo for an isotropic set of rays (f_cov = 5 clumps of tau0 = 1e3 ... x 4/3)
    assert len(taus) == 200


@pytest.mark.slow
def test_whole_path_known_answer(oracle_mod):
    """log_back:52: <N_scatt> = 4.3454E+03 (1e6 photons) for clump_NHI18_fcov1 with the flat rotation curve; the
    distribution is heavy-tailed (median 3), so 6e4 photons give ~2 % and another clump realisation adds its own scatter"""
    o = oracle_mod.Oracle(**NHI18)
    n = 60000
    o.run(n, seed=4)
    ns, _ = o.photon_log()
    assert abs(ns.mean() - 4345.4) < 0.03 * 4345.4 + 4 * ns.std() / np.sqrt(n), (ns.mean(), ns.std() / np.sqrt(n))


# examples/clump_sphere/clump_overlap_mono.in (static) and clump_overlap_peel.in (sigma_v = 100 km/s): 30 clumps of radius 0.3 in the unit sphere
OVERLAP = dict(use_clump_medium=True, rmax=1.0, clump_radius=0.3, clump_f_cov=2.0, clump_tau0=1e3, temperature=1e4, clump_allow_overlap=True,
               nxfreq=101, velocity_min=-100.0, velocity_max=100.0, nx=11, ny=11, nz=11, iseed=7)


def test_overlapping_population_host_equals_oracle(oracle_mod):
    """par%clump_allow_overlap (clump_mod.f90:1090): the trial that overlaps is kept, check_has_overlap (:1544-1591) finds the pairs"""
    for extra, want in ((dict(), True), (dict(clump_sigma_v=100.0), True), (dict(clump_radius=0.01, clump_f_cov=0.02), False)):
        kw = dict(OVERLAP, **extra)
        o = oracle_mod.Oracle(**kw)
        par, line, cl, g = host_clumps(kw)
        assert cl.N == int(o.scalar("cl_N")) and (cl.N == 30 or not want)
        for k in ("x", "y", "z", "vx", "vy", "vz"):
            assert np.array_equal(o.array("cl_" + k), getattr(cl, k)), k
        assert cl.has_overlap == bool(o.scalar("cl_has_overlap")) == want
        assert int(o.scalar("raytrace_variant")) == (7 if want else 4)
    # the same placement draws with and without the switch while no trial is rejected; a dense population differs after the first rejection
    a = oracle_mod.Oracle(**dict(OVERLAP, clump_allow_overlap=False, clump_radius=0.05, clump_f_cov=1.0))
    b = oracle_mod.Oracle(**dict(OVERLAP, clump_radius=0.05, clump_f_cov=1.0))
    xa, xb = a.array("cl_x"), b.array("cl_x")
    assert xa[0] == xb[0] and not np.array_equal(xa, xb) and bool(b.scalar("cl_has_overlap")) and not bool(a.scalar("cl_has_overlap"))


def test_overlap_forms_on_a_disjoint_population_equal_the_standard_forms(oracle_mod):
    """raytrace_to_{tau,edge}_clump_overlap + scatter_resonance_clump_* (raytrace_clump.f90:606-918, scattering_car.f90:898-937) against
    raytrace_to_{tau,edge}_clump on clumps that do not overlap: the same physics through two different bookkeepings (global-frame frequency
    and event lists against clump-frame frequency and nearest-entry search), so the results agree statistically -- with moving clumps, dust
    and a far observer"""
    kw = dict(use_clump_medium=True, rmax=1.0, clump_radius=0.08, clump_f_cov=3.0, clump_tau0=1e2, temperature=1e4, clump_sigma_v=30.0, nxfreq=101,
              velocity_min=-150.0, velocity_max=150.0, nx=11, ny=11, nz=11, iseed=7)
    n = 20000

    def pair(kw, n, seeds=(3, 4)):
        a = oracle_mod.Oracle(**kw); a.run(n, seed=seeds[0])
        b = oracle_mod.Oracle(**kw); b.set_array("cl_has_overlap", np.array([1.0])); assert int(b.scalar("raytrace_variant")) == 7
        b.run(n, seed=seeds[1])
        return a, b
    a, b = pair(kw, n)
    na, nb = a.photon_log()[0], b.photon_log()[0]
    assert abs(na.mean() - nb.mean()) < 4.0 * np.hypot(na.std(), nb.std()) / np.sqrt(n), (na.mean(), nb.mean())
    ja, jb = a.array("Jout"), b.array("Jout")
    assert np.sum((ja - jb) ** 2 / np.maximum(ja + jb, 1.0)) / np.count_nonzero(ja + jb) < 1.6
    # dust and a far observer on static clumps.  (With moving clumps the two forms differ BY CONSTRUCTION in the reference: the dust
    # probability of scattering_car.f90:75-76 and the capped peel march see photon%xfreq, which is the clump-frame frequency in the
    # standard forms and the global-frame one in the overlap forms.  Restated as written; visible as Jabs 1e-3 against 6e2 at sigma_v = 30.)
    kw0 = dict(kw, clump_sigma_v=0.0, DGR=1.0, cext_dust=3e-18, nxim=8, nyim=8)
    a0, b0 = pair(kw0, n)
    Ja, Jb = a0.array("Jabs").sum(), b0.array("Jabs").sum()
    assert Ja > 0.02 * n and abs(Ja - Jb) < 0.03 * Ja + 4 * np.sqrt(Ja), (Ja, Jb)
    sa, sb = a0.array("scatt:0").sum(), b0.array("scatt:0").sum()
    assert abs(sa - sb) < 0.05 * sa, (sa, sb)
    a1, b1 = pair(kw0, 2000, seeds=(3, 3))                        # direct light: no draw of the scattering involved, only the edge march
    assert np.isclose(a1.array("direc:0").sum(), b1.array("direc:0").sum(), rtol=1e-9)


def test_overlapping_clumps_whole_path(oracle_mod):
    """30 clumps of radius 0.3: inside k clumps the opacity is the sum of k profiles; merged clumps trap more than the same clumps would
    one after the other (fewer free surfaces), every packet still leaves, and the covering factor keeps its meaning: a direct ray from the centre
    sees exp(-sum of chord optical depths)"""
    n = 3000
    o = oracle_mod.Oracle(**dict(OVERLAP, clump_tau0=30.0)); o.run(n, seed=5)
    assert o.array("Jout").sum() == n and o.photon_log()[0].mean() > 5.0
    # optical depth of a ray through the overlap forms = sum over clumps of kappa * chord (static clumps, line centre): additive by construction
    x, y, z, r = o.array("cl_x"), o.array("cl_y"), o.array("cl_z"), 0.3
    k = np.array([0.3, -0.5, 0.81]); k /= np.linalg.norm(k)
    b = -(x * k[0] + y * k[1] + z * k[2]); disc = b * b - (x * x + y * y + z * z) + r * r
    hit = disc > 0
    t1 = np.clip(-b[hit] - np.sqrt(disc[hit]), 0.0, 1.0); t2 = np.clip(-b[hit] + np.sqrt(disc[hit]), 0.0, 1.0)
    from oracle import oracle as om
    kap = o.array("cl_rhokap")[0] * om.voigt(np.array([0.0]), o.scalar("cl_voigt_a_ref"))[0]
    tau = o.raytrace_to_edge(7, np.array([0, 0, 0, k[0], k[1], k[2], 0.0, 1.0]), np.array([0, 1, 1, 1], dtype=np.int32))
    assert np.isclose(tau, kap * np.sum(t2 - t1), rtol=1e-10)


@pytest.mark.parametrize("ext", [".fits.gz", ".fits", ".h5"])
def test_clump_file_round_trip(tmp_path, ext):
    """write_clumps_info / read_clumps_info (clump_mod.f90:1779-1961, 2040-2312): float32 columns X..VZ in code units and km/s, the constant
    columns R_CLUMP / RHOKAP / TEMP dropped in favour of the keywords CL_RAD / RHOKAP / TEMP_CL, par%clump_input_file without an extension
    resolved like io_resolve_filename, the loaded population searched for overlaps, rescaling to par%taumax"""
    from lart_amd import host, clump
    kw = dict(C4, clump_sigma_v=25.0, clump_radius=0.05, clump_f_cov=2.0)
    par, line, cl, g = host_clumps(kw)
    fn = str(tmp_path / ("pop_clumps" + ext))
    clump.write_clumps_info(fn, par, line, cl)
    cols, keys = clump.read_table_file(fn)
    assert sorted(cols) == sorted(["X", "Y", "Z", "VX", "VY", "VZ"]) and int(keys["N_CLUMPS"]) == cl.N            # all three clump columns are constant here
    assert np.isclose(keys["CL_RAD"], 0.05) and np.isclose(keys["RHOKAP"], cl.rhokap_ref) and keys["TEMP_CL"] == 1e4 and np.isclose(keys["F_COV"], cl.fcov)
    par2, line2, cl2, g2 = host_clumps(dict(kw, clump_input_file=str(tmp_path / "pop_clumps") if ext == ".fits.gz" else fn))
    assert cl2.N == cl.N and not cl2.has_overlap and cl2.from_file
    assert np.array_equal(cl2.x, cl.x.astype(np.float32).astype(np.float64))                       # what float32 keeps
    assert np.allclose(cl2.vx * cl2.vtherm, cl.vx * cl.vtherm, rtol=1e-6, atol=1e-5) and np.allclose(cl2.rhokap, cl.rhokap, rtol=1e-6)
    assert np.isclose(par2.taumax, par.taumax, rtol=1e-5) and cl2.cgx == cl.cgx
    par3, _, cl3, _ = host_clumps(dict(kw, clump_input_file=fn, taumax=50.0))                      # rescale_loaded_clumps_to_target :2385-2449
    assert np.isclose(par3.taumax, 50.0, rtol=1e-12) and np.isclose(cl3.rhokap[0] / cl2.rhokap[0], 50.0 / par2.taumax, rtol=1e-12)


def test_clump_file_with_clump_columns_and_overlaps(tmp_path):
    """a hand-made population: varying radii and temperatures go into R_CLUMP / TEMP columns, a DENSITY column is converted with DIST_CM
    (:2190-2207), overlapping clumps switch the overlap forms on (init_clumps :678-679 checks every loaded file)"""
    from lart_amd import host, clump
    rng = np.random.default_rng(3); n = 40
    pos = rng.normal(0, 0.35, (n, 3)); pos = pos[np.linalg.norm(pos, axis=1) < 0.7][:25]; n = len(pos)
    cols = [("X", pos[:, 0]), ("Y", pos[:, 1]), ("Z", pos[:, 2]), ("VX", rng.normal(0, 30, n)), ("VY", rng.normal(0, 30, n)), ("VZ", rng.normal(0, 30, n)),
            ("R_CLUMP", rng.uniform(0.1, 0.3, n)), ("DENSITY", rng.uniform(0.5, 2.0, n) * 1e-2), ("TEMP", rng.uniform(0.8e4, 2e4, n))]
    fn = str(tmp_path / "made.fits.gz")
    clump.write_table_file(fn, cols, dict(N_CLUMPS=n, DISTUNIT="kpc", DIST_CM=3.0857e21))
    par, line, cl, g = host_clumps(dict(use_clump_medium=True, rmax=1.0, clump_radius=0.2, clump_input_file=fn, temperature=1e4, nxfreq=101,
                                        velocity_min=-300.0, velocity_max=300.0, nx=11, ny=11, nz=11))
    assert cl.N == n and cl.has_overlap and par.distance_unit == "kpc" and par.distance2cm == 3.0857e21
    r32 = cols[6][1].astype(np.float32).astype(np.float64)
    assert np.array_equal(cl.radius, r32) and cl.radius_max == r32.max()
    T32 = cols[8][1].astype(np.float32).astype(np.float64)
    assert np.allclose(cl.vtherm, host.vtherm_total(line, T32), rtol=1e-14)
    assert np.allclose(cl.voigt_a * cl.Dfreq, line.damping / (4 * np.pi))
    nH32 = cols[7][1].astype(np.float32).astype(np.float64)
    assert np.allclose(cl.rhokap, nH32 * line.cross0 / cl.Dfreq * 3.0857e21, rtol=1e-12)
    assert np.allclose(cl.vx * cl.vtherm, cols[3][1].astype(np.float32), rtol=1e-12)
    with pytest.raises(host.LartInputError):
        host_clumps(dict(use_clump_medium=True, rmax=1.0, clump_radius=0.2, clump_input_file=str(tmp_path / "nothing_here"), temperature=1e4))


def test_bicone(oracle_mod, tmp_path):
    """par%cone_opening (examples/bicone): clump centres drawn directly inside the bicone about the z axis (clump_mod.f90:1029-1048), cells and
    leaves outside it emptied (grid_mod_car.f90:397-412, grid_mod_amr.f90:336-350); hosts equal the oracle"""
    from lart_amd import host, amr
    kw = dict(use_clump_medium=True, rmax=1.0, clump_radius=0.02, clump_f_cov=2.0, clump_tau0=1e3, temperature=1e4, cone_opening=45.0, rmin=0.025, nxfreq=121, iseed=3)
    o = oracle_mod.Oracle(**kw); par, line, cl, g = host_clumps(kw)
    assert cl.N == int(o.scalar("cl_N")) and np.array_equal(o.array("cl_x"), cl.x) and np.array_equal(o.array("cl_z"), cl.z)
    r = np.sqrt(cl.x ** 2 + cl.y ** 2 + cl.z ** 2)
    mu = np.abs(cl.z) / r                                                       # fully inside: the whole clump, not only its centre
    assert np.all(mu * np.sqrt(1 - (0.02 / r) ** 2) - np.sqrt(1 - mu ** 2) * 0.02 / r >= np.cos(np.pi / 4) - 1e-12) and (cl.z > 0).any() and (cl.z < 0).any()
    kc = dict(temperature=1e4, taumax=1e2, nx=21, ny=21, nz=21, rmax=1.0, cone_opening=40.0, nxfreq=61, DGR=1.0)
    oc = oracle_mod.Oracle(**kc); p = host.Params()
    for k, v in kc.items():
        host.set_param(p, k, v)
    p, ln, _ = host.read_input(p); gc = host.grid_create(p, ln)
    assert np.allclose(oc.array("rhokap"), gc.rhokap.ravel(order="F"), rtol=1e-14) and np.allclose(oc.array("rhokapD"), gc.rhokapD.ravel(order="F"), rtol=1e-14)
    assert 0.1 < (gc.rhokap > 0).mean() < 0.2 and gc.rhokap[10, 10, 15] > 0 and gc.rhokap[15, 10, 10] == 0
    fn = str(tmp_path / "s.dat"); amr.write_generic_amr(fn, amr.make_amr_sphere(level_min=2, level_max=4))
    ka = dict(use_amr_grid=True, amr_file=fn, temperature=1e4, taumax=1e2, rmax=1.0, cone_opening=40.0, nxfreq=61)
    oa = oracle_mod.Oracle(**ka); pa = host.Params()
    for k, v in ka.items():
        host.set_param(pa, k, v)
    pa, la, _ = host.read_input(pa); ga, sg = amr.grid_create_amr(pa, la, fn)
    assert np.allclose(oa.array("rhokap"), ga.rhokap, rtol=1e-14) and 0 < (ga.rhokap > 0).mean() < 0.3 and oa.scalar("taumax") == 1e2
