"""The Fortran host (LaRT_amd.x): namelist surface + grid set-up without a GPU (--dry-run), end to end on the GPU."""
import os
import re
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "lart_amd", "fhost", "LaRT_amd.x")

T1TAU3 = """&parameters
 par%no_photons  = 1e6
 par%temperature = 1.0e1
 par%taumax      = 1.0e3
 par%scatt_mat_file = '../data/mueller_Lyalpha.dat'
 par%DGR         = 0.0
 par%Vexp        = 0.0
 par%use_stokes   = .true.
 par%save_Jin     = .true.
 par%nx = 201
 par%ny = 201
 par%nz = 201
 par%rmax   = 1.0
 par%nxfreq = 201
 par%nxim   = 129
 par%nyim   = 129
 par%nprint = 1e6
/
"""


@pytest.fixture(scope="module")
def exe():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(EXE)
    return EXE


def run(exe, text, tmp_path, *args, name="case.in"):
    f = tmp_path / name
    f.write_text(text)
    return subprocess.run([exe, name, *args], cwd=tmp_path, capture_output=True, text=True, timeout=600)


def test_dry_run_reproduces_the_shipped_log_header(exe, tmp_path):
    # examples/sphere_peel/out.txt:8-20 prints these lines for t1tau3.in
    r = run(exe, T1TAU3, tmp_path, "--dry-run")
    out = r.stdout
    assert "voigt_a          :        1.492E-02" in out
    assert "N(H  I)_pole     :        5.449E+14" in out
    assert "tau_pole  (H  I) :        1.000E+03" in out
    assert "tau_homo  (H  I) :        1.000E+03" in out
    assert re.search(r"observers:\s+1\s+image\s+129\s+129", out)


def test_unknown_field_and_unsupported_mode_fail(exe, tmp_path):
    r = run(exe, "&parameters\n par%no_such_thing = 1\n/\n", tmp_path, "--dry-run", name="bad.in")
    assert r.returncode != 0 and "ERROR" in r.stdout
    r = run(exe, "&parameters\n par%use_amr_grid = .true.\n/\n", tmp_path, "--dry-run", name="amr.in")
    assert r.returncode != 0 and "needs par%amr_file" in r.stdout
    r = run(exe, "&parameters\n par%density_rscale = 0.5\n/\n", tmp_path, "--dry-run", name="fs.in")
    assert r.returncode != 0 and "not supported" in r.stdout


def read_fits(path):
    """tiny reader for the uncompressed FITS the host writes: returns {EXTNAME: (header dict, array or table dict)}"""
    import gzip
    raw = (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")).read()
    pos, hdus = 0, {}
    while pos < len(raw):
        hdr = {}
        while True:
            block = raw[pos:pos + 2880].decode("ascii"); pos += 2880
            cards = [block[i:i + 80] for i in range(0, 2880, 80)]
            done = False
            for c in cards:
                k = c[:8].strip()
                if k == "END":
                    done = True; break
                if c[8:10] == "= ":
                    hdr[k] = c[10:].split(" / ")[0].strip().strip("'").strip()
            if done:
                break
        naxis = int(hdr.get("NAXIS", 0))
        dims = [int(hdr["NAXIS%d" % (i + 1)]) for i in range(naxis)]
        nbytes = abs(int(hdr.get("BITPIX", 8))) // 8 * int(np.prod(dims)) if naxis else 0
        data = raw[pos:pos + nbytes]; pos += (nbytes + 2879) // 2880 * 2880
        if hdr.get("XTENSION", "").startswith("BINTABLE"):
            ncol = int(hdr["TFIELDS"]); arr = np.frombuffer(data, dtype=">f8").reshape(dims[1], ncol)
            hdus[hdr["EXTNAME"]] = (hdr, {hdr["TTYPE%d" % (i + 1)]: arr[:, i] for i in range(ncol)})
        elif naxis:
            hdus[hdr["EXTNAME"]] = (hdr, np.frombuffer(data, dtype=">f8").reshape(dims[::-1]).T)
    return hdus


def read_h5(path):
    """reader for the host's HDF5 files through the HDF5 C library (no h5py in the image): {group: (attrs, table dict | array)}
    in creation order; arrays come back in Fortran index order (the file holds the reversed dimensions)"""
    import ctypes as C
    lib = None
    for cand in ("/opt/conda/lib/libhdf5.so", "libhdf5.so"):
        try:
            lib = C.CDLL(cand); break
        except OSError:
            pass
    if lib is None:
        pytest.skip("no HDF5 C library in this image")
    hid = C.c_int64
    lib.H5open()
    NATIVE_DOUBLE = hid.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value; NATIVE_INT = hid.in_dll(lib, "H5T_NATIVE_INT_g").value
    lib.H5Fopen.restype = hid; lib.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    lib.H5Gopen2.restype = hid; lib.H5Gopen2.argtypes = [hid, C.c_char_p, hid]
    lib.H5Dopen2.restype = hid; lib.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    lib.H5Dget_space.restype = hid; lib.H5Dget_space.argtypes = [hid]
    lib.H5Sget_simple_extent_ndims.argtypes = [hid]; lib.H5Sget_simple_extent_dims.argtypes = [hid, C.c_void_p, C.c_void_p]
    lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    lib.H5Aopen.restype = hid; lib.H5Aopen.argtypes = [hid, C.c_char_p, hid]
    lib.H5Aget_type.restype = hid; lib.H5Aget_type.argtypes = [hid]; lib.H5Tget_class.argtypes = [hid]; lib.H5Tget_size.restype = C.c_size_t; lib.H5Tget_size.argtypes = [hid]
    lib.H5Aread.argtypes = [hid, hid, C.c_void_p]
    for fn in ("H5Fclose", "H5Gclose", "H5Dclose", "H5Aclose", "H5Sclose", "H5Tclose"):
        getattr(lib, fn).argtypes = [hid]                     # 64-bit identifiers: without this every close fails and the file stays locked
    lib.H5Literate.argtypes = [hid, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.H5Aiterate2.argtypes = [hid, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    LCB = C.CFUNCTYPE(C.c_int, hid, C.c_char_p, C.c_void_p, C.c_void_p)

    def names_of(loc, attrs=False):
        out = []
        cb = LCB(lambda g, name, info, data: out.append(name.decode()) or 0)
        if attrs:
            lib.H5Aiterate2(loc, 0, 0, None, cb, None)          # H5_INDEX_NAME, H5_ITER_INC
        else:
            lib.H5Literate(loc, 1, 0, None, cb, None)           # H5_INDEX_CRT_ORDER (tracked by the writer), H5_ITER_INC
        return out
    f = lib.H5Fopen(path.encode(), 0, 0)
    assert f >= 0, path
    res = {}
    for gname in names_of(f):
        g = lib.H5Gopen2(f, gname.encode(), 0)
        attrs = {}
        for an in names_of(g, attrs=True):
            a = lib.H5Aopen(g, an.encode(), 0); t = lib.H5Aget_type(a); cls = lib.H5Tget_class(t)
            if cls == 1:                                         # H5T_FLOAT
                v = C.c_double(); lib.H5Aread(a, NATIVE_DOUBLE, C.byref(v)); attrs[an] = v.value
            elif cls == 0:                                       # H5T_INTEGER
                v = C.c_int(); lib.H5Aread(a, NATIVE_INT, C.byref(v)); attrs[an] = v.value
            else:
                buf = C.create_string_buffer(lib.H5Tget_size(t) + 1); lib.H5Aread(a, t, buf); attrs[an] = buf.value.decode().strip()
            lib.H5Tclose(t); lib.H5Aclose(a)
        cols = {}
        for dn in names_of(g):
            d = lib.H5Dopen2(g, dn.encode(), 0); sp = lib.H5Dget_space(d); nd = lib.H5Sget_simple_extent_ndims(sp)
            dims = (C.c_uint64 * nd)(); lib.H5Sget_simple_extent_dims(sp, dims, None)
            arr = np.zeros(tuple(dims), dtype=np.float64); lib.H5Dread(d, NATIVE_DOUBLE, 0, 0, 0, arr.ctypes.data_as(C.c_void_p))
            cols[dn] = arr.T if nd > 1 else arr                  # back to Fortran index order
            lib.H5Sclose(sp); lib.H5Dclose(d)
        res[gname] = (attrs, cols["data"] if list(cols) == ["data"] else cols)
        lib.H5Gclose(g)
    lib.H5Fclose(f)
    return res


@pytest.mark.gpu
def test_end_to_end_matches_python_host(exe, tmp_path):
    text = ("&parameters\n par%no_photons = 30000\n par%iseed = 5\n par%temperature = 1e4\n par%taumax = 1e2\n par%nx = 31\n par%ny = 31\n"
            " par%nz = 31\n par%rmax = 1.0\n par%nxfreq = 81\n par%nxim = 12\n par%nyim = 12\n par%use_stokes = .true.\n par%save_peeloff_2D = .true.\n par%save_radial_profile = .true.\n par%file_format = 'fits'\n/\n")
    r = run(exe, text, tmp_path, name="e2e.in")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Average Number of scattering" in r.stdout
    spec = read_fits(str(tmp_path / "e2e.fits.gz"))["Spectrum"]
    assert list(spec[1].keys()) == ["Xfreq", "velocity", "wavelength", "Jout", "Jin"]
    from lart_amd import host
    sim = host.Simulation(dict(no_photons=30000, iseed=5, temperature=1e4, taumax=1e2, nx=31, ny=31, nz=31, rmax=1.0, nxfreq=81,
                               nxim=12, nyim=12, use_stokes=True, save_peeloff_2D=True))
    sim.run_simulation()
    t = sim.output_normalize()
    # same tables, same seed, same photon ids -> same histories; tallies equal up to atomic summation order
    assert np.allclose(spec[1]["Jout"], t["Jout"], rtol=1e-9, atol=1e-15)
    assert np.allclose(spec[1]["Jin"], t["Jin"], rtol=1e-9, atol=1e-15)
    assert np.isclose(float(spec[0]["Nsc_tot"]), t["nscatt_tot"], rtol=1e-9)
    # one set of files per observer, cubes in <base>_obs / <base>_stokes, images in <base>_obs2D / <base>_stokes_2D
    obs = read_fits(str(tmp_path / "e2e_obs.fits.gz"))
    assert list(obs) == ["Scattered", "Direct", "RadialI"] and obs["Scattered"][1].shape == (81, 12, 12)
    assert np.allclose(obs["Scattered"][1], t["scatt"][..., 0], rtol=1e-6, atol=1e-12 * t["scatt"].max())
    assert obs["Scattered"][0]["CTYPE2"].strip() == "RA--TAN" and float(obs["Scattered"][0]["CRPIX2"]) == 6.5
    stk = read_fits(str(tmp_path / "e2e_stokes.fits.gz"))
    assert list(stk) == ["Stokes_I", "Stokes_Q", "Stokes_U", "Stokes_V", "Stokes_radial"]
    assert np.allclose(stk["Stokes_Q"][1], t["Q"][..., 0], rtol=1e-6, atol=1e-9 * np.abs(t["I"]).max())
    o2 = read_fits(str(tmp_path / "e2e_obs2D.fits.gz")); s2 = read_fits(str(tmp_path / "e2e_stokes_2D.fits.gz"))
    assert np.allclose(o2["Direct"][1], t["direc_2D"][..., 0], rtol=1e-6, atol=1e-12 * t["direc_2D"].max())
    assert np.allclose(s2["Stokes_U"][1], t["U_2D"][..., 0], rtol=1e-6, atol=1e-9 * np.abs(t["I_2D"]).max())
    # make_radial_intensity / make_radial_stokes (output_sum_rect.f90:489-658)
    ri = host.radial_profiles(sim.par, sim.grid, t); rs = host.radial_profiles(sim.par, sim.grid, t, stokes=True)
    r2 = host.radial_profiles(sim.par, sim.grid, t, use_2D=True, stokes=True)
    assert np.allclose(obs["RadialI"][1]["r"], ri[:, 0]) and np.allclose(obs["RadialI"][1]["I"], ri[:, 1], rtol=1e-6, atol=1e-12 * ri[:, 1].max())
    for c, k in enumerate(("r", "I", "Q", "U", "V", "pol")):
        assert np.allclose(stk["Stokes_radial"][1][k], rs[:, c], rtol=1e-5, atol=1e-8 * np.abs(rs[:, 1]).max()), k
        assert np.allclose(s2["Stokes_radial"][1][k], r2[:, c], rtol=1e-5, atol=1e-8 * np.abs(r2[:, 1]).max()), k
    assert rs[:, 5].max() > 0.0                                         # a resolved sphere is radially polarised


CLUMP_IN = """&parameters
 par%use_clump_medium = .true.
 par%rmax             = 1.0
 par%clump_radius     = 0.01
 par%clump_f_cov      = 5.0
 par%clump_tau0       = 1.0e2
 par%temperature      = 1.0e4
 par%clump_sigma_v    = 15.0
 par%velocity_type    = 'hubble'
 par%Vexp             = 40.0
 par%no_photons       = 40000
 par%spectral_type    = 'voigt'
 par%nxfreq           = 151
 par%velocity_min     = -300.0
 par%velocity_max     =  300.0
 par%nx = 11, par%ny = 11, par%nz = 11
 par%nxim = 16, par%nyim = 16
 par%save_peeloff_2D  = .true.
 par%iseed            = 7
 par%file_format      = 'fits'
/
"""


def test_clump_dry_run_equals_python_host(exe, tmp_path):
    """the clump population of the Fortran host (C helpers on the reference's MT19937-64 stream) = the Python host's"""
    r = run(exe, CLUMP_IN, tmp_path, "--dry-run", name="clump.in")
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    from lart_amd import host, clump
    par = host.read_namelist(str(tmp_path / "clump.in")); par, line, _ = host.read_input(par)
    cl, g = clump.grid_create_clump(par, line)
    assert re.search(r"Clumps: N_clumps\s+=\s+%d\b" % cl.N, out) and cl.N == 66667
    assert re.search(r"CSR grid:\s+%d registrations in\s+%d\^3 cells" % (len(cl.cg_list), cl.cgx), out)
    m = re.search(r"Clump derived: taumax\s+=\s+(\S+)", out); assert abs(float(m.group(1)) / par.taumax - 1) < 1e-5
    m = re.search(r"Clumps: cl_rhokap =\s+(\S+)", out); assert abs(float(m.group(1)) / cl.rhokap_ref - 1) < 1e-4
    assert "Clumps: velocity_type   = hubble" in out


@pytest.mark.gpu
def test_clump_end_to_end_matches_python_host(exe, tmp_path):
    r = run(exe, CLUMP_IN, tmp_path, name="clump.in")
    assert r.returncode == 0, r.stdout + r.stderr
    spec = read_fits(str(tmp_path / "clump.fits.gz"))["Spectrum"]
    from lart_amd import host
    sim = host.Simulation(str(tmp_path / "clump.in"))
    sim.run_simulation()
    t = sim.output_normalize()
    assert np.allclose(spec[1]["Jout"], t["Jout"], rtol=1e-9, atol=1e-15)           # same clumps, same seed, same photon ids
    assert np.isclose(float(spec[0]["Nsc_tot"]), t["nscatt_tot"], rtol=1e-9)
    obs = read_fits(str(tmp_path / "clump_obs.fits.gz"))
    assert np.allclose(obs["Scattered"][1], t["scatt"][..., 0], rtol=1e-6, atol=1e-12 * t["scatt"].max())


# the parameters of examples/clump_sphere/clump_overlap_peel.in (fewer photons, a fixed seed, a smaller image)
OVERLAP_IN = """&parameters
 par%use_clump_medium    = .true.
 par%rmax                = 1.0
 par%clump_radius        = 0.3
 par%clump_f_cov         = 2.0
 par%clump_tau0          = 1.0e3
 par%temperature         = 1.0e4
 par%clump_sigma_v       = 100.0
 par%clump_allow_overlap = .true.
 par%no_photons          = 2e4
 par%spectral_type       = 'voigt'
 par%nxfreq              = 101
 par%velocity_min        = -200.0
 par%velocity_max        =  200.0
 par%nx = 11, par%ny = 11, par%nz = 11
 par%save_peeloff        = .true.
 par%nxim                = 17
 par%nyim                = 17
 par%iseed               = 5
/
"""


def test_overlapping_clumps_dry_run(exe, tmp_path):
    """par%clump_allow_overlap: placement without rejection and the overlap check of init_clumps (clump_mod.f90:889-891, :1544-1591)"""
    r = run(exe, OVERLAP_IN, tmp_path, "--dry-run", name="ovl.in")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Random placement done (overlap allowed)." in r.stdout and "overlapping clumps detected" in r.stdout
    assert re.search(r"Clumps: N_clumps\s+=\s+30\b", r.stdout)
    from lart_amd import host, clump
    par = host.read_namelist(str(tmp_path / "ovl.in")); par, line, _ = host.read_input(par)
    cl, g = clump.grid_create_clump(par, line)
    assert cl.N == 30 and cl.has_overlap
    m = re.search(r"Clump derived: taumax\s+=\s+(\S+)", r.stdout); assert abs(float(m.group(1)) / par.taumax - 1) < 1e-5


@pytest.mark.gpu
def test_overlapping_clumps_end_to_end_matches_python_host(exe, tmp_path):
    r = run(exe, OVERLAP_IN, tmp_path, name="ovl.in")
    assert r.returncode == 0, r.stdout + r.stderr
    H = read_h5(str(tmp_path / "ovl.h5"))
    from lart_amd import host
    sim = host.Simulation(str(tmp_path / "ovl.in")); sim.run_simulation(); t = sim.output_normalize()
    np.testing.assert_allclose(H["Spectrum"][1]["Jout"], t["Jout"], rtol=1e-9, atol=1e-15)
    O = read_h5(str(tmp_path / "ovl_obs.h5"))
    assert np.allclose(O["Scattered"][1], t["scatt"][..., 0], rtol=1e-6, atol=1e-12 * t["scatt"].max())


@pytest.mark.parametrize("fmt", ["hdf5", "fits"])
def test_clump_file_written_and_read_back(exe, tmp_path, fmt):
    """par%save_clump_info -> <base>_clumps.{h5,fits.gz} (write_clumps_info, clump_mod.f90:1779-1961); par%clump_input_file (read_clumps_info
    :2040-2312) loads it again, also without its extension, and rescales to par%taumax (:2385-2449).  Both hosts read both hosts' files."""
    from lart_amd import host, clump
    base = OVERLAP_IN.replace(" par%no_photons          = 2e4\n", " par%no_photons = 100\n par%save_clump_info = .true.\n par%file_format = '" + fmt + "'\n")
    r = run(exe, base, tmp_path, "--dry-run", name="w.in")
    assert r.returncode == 0, r.stdout + r.stderr
    fn = str(tmp_path / ("w_clumps" + (".h5" if fmt == "hdf5" else ".fits.gz")))
    assert "Clumps saved to" in r.stdout and os.path.exists(fn) and "(R_CLUMP/RHOKAP/TEMP) =  F F F" in r.stdout
    cols, keys = clump.read_table_file(fn)
    assert sorted(cols) == ["VX", "VY", "VZ", "X", "Y", "Z"] and int(keys["N_CLUMPS"]) == 30 and keys["CL_RAD"] == 0.3 and keys["SIGMA_V"] == 100.0
    par = host.read_namelist(str(tmp_path / "w.in")); par, line, _ = host.read_input(par)
    cl, g = clump.grid_create_clump(par, line)                                        # the population the file was written from
    assert np.array_equal(cols["X"], cl.x.astype(np.float32).astype(np.float64)) and np.allclose(cols["VZ"], cl.vz * cl.vtherm, rtol=1e-6, atol=1e-4)
    assert np.isclose(keys["RHOKAP"], cl.rhokap_ref, rtol=1e-12) and np.isclose(keys["F_COV"], cl.fcov, rtol=1e-12)
    # read back by the Fortran host (extension left to io_resolve_filename) and by the Python host: the same population, the same scalars
    rd = base.replace(" par%save_clump_info = .true.\n", " par%clump_input_file = 'w_clumps'\n par%taumax = 2.0e3\n")
    r2 = run(exe, rd, tmp_path, "--dry-run", name="r.in")
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert "Clumps: reading from w_clumps" in r2.stdout and "rescaled to par%taumax" in r2.stdout and "overlapping clumps detected" in r2.stdout
    par2 = host.read_namelist(str(tmp_path / "r.in")); par2.clump_input_file = fn; par2, line2, _ = host.read_input(par2)
    cl2, g2 = clump.grid_create_clump(par2, line2)
    assert cl2.N == 30 and cl2.has_overlap and np.isclose(par2.taumax, 2.0e3, rtol=1e-12)
    m = re.search(r"Clump derived: taumax\s+=\s+(\S+)", r2.stdout); assert abs(float(m.group(1)) / 2.0e3 - 1) < 1e-5
    m = re.search(r"Clumps: cl_rhokap =\s+(\S+)", r2.stdout); assert abs(float(m.group(1)) / cl2.rhokap_ref - 1) < 1e-4
    m = re.search(r"CSR grid:\s+(\d+) registrations", r2.stdout); assert int(m.group(1)) == len(cl2.cg_list)
    # a file written by the Python host is read by the Fortran host
    clump.write_clumps_info(str(tmp_path / ("py_clumps" + (".h5" if fmt == "hdf5" else ".fits"))), par, line, cl)
    r3 = run(exe, rd.replace("'w_clumps'", "'py_clumps'"), tmp_path, "--dry-run", name="r3.in")
    assert r3.returncode == 0 and re.search(r"CSR grid:\s+%d registrations" % len(cl2.cg_list), r3.stdout), r3.stdout + r3.stderr


MERGE_IN = ("&parameters\n par%no_photons = {n}\n par%temperature = 1e4\n par%taumax = 1e2\n par%nx = 21\n par%ny = 21\n par%nz = 21\n par%rmax = 1.0\n par%nxfreq = 51\n"
            " par%nxim = 8\n par%nyim = 8\n par%use_stokes = .true.\n par%save_peeloff_2D = .true.\n par%save_Jin = .true.\n par%save_Jmu = .true.\n par%nmu = 4\n"
            " par%save_radial_profile = .true.\n par%iseed = {seed}\n par%out_merge = {merge}\n par%file_format = '{fmt}'\n/\n")


@pytest.mark.parametrize("fmt", ["hdf5", "fits"])
def test_out_merge_dry_run(exe, tmp_path, fmt):
    """par%out_merge (write_output_rect.f90:75-78, 112-143): no previous file -> a plain write; with one -> the photon counts add up in every file"""
    rd = read_h5 if fmt == "hdf5" else read_fits; ext = ".h5" if fmt == "hdf5" else ".fits.gz"
    r = run(exe, MERGE_IN.format(n=1000, seed=3, merge=".true.", fmt=fmt), tmp_path, "--dry-run-write", name="m.in")
    assert r.returncode == 0 and "out_merge: combined" not in r.stdout, r.stdout + r.stderr
    k = {a.lower(): b for a, b in rd(str(tmp_path / ("m" + ext)))["Spectrum"][0].items()}
    assert float(k["nphotons"]) == 1000
    r = run(exe, MERGE_IN.format(n=3000, seed=4, merge=".true.", fmt=fmt), tmp_path, "--dry-run-write", name="m.in")
    assert r.returncode == 0 and "out_merge: combined with m" + ext in r.stdout, r.stdout + r.stderr
    k = {a.lower(): b for a, b in rd(str(tmp_path / ("m" + ext)))["Spectrum"][0].items()}
    assert float(k["nphotons"]) == 4000
    for f, sec in (("m_obs", "Scattered"), ("m_stokes", "Stokes_I"), ("m_obs2D", "Scattered"), ("m_stokes_2D", "Stokes_I")):
        kk = {a.lower(): b for a, b in rd(str(tmp_path / (f + ext)))[sec][0].items()}
        assert float(kk["nphotons"]) == 4000, f
    r = run(exe, MERGE_IN.format(n=500, seed=5, merge=".false.", fmt=fmt), tmp_path, "--dry-run-write", name="m.in")      # switched off: overwritten
    k = {a.lower(): b for a, b in rd(str(tmp_path / ("m" + ext)))["Spectrum"][0].items()}
    assert r.returncode == 0 and float(k["nphotons"]) == 500


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["hdf5", "fits"])
def test_out_merge_end_to_end(exe, tmp_path, fmt):
    """two runs with different seeds merged by the host = the photon-weighted mean of the two runs made separately (every array of every file)"""
    rd = read_h5 if fmt == "hdf5" else read_fits; ext = ".h5" if fmt == "hdf5" else ".fits.gz"
    out = {}
    for name, n, seed in (("a", 20000, 3), ("b", 60000, 4)):
        r = run(exe, MERGE_IN.format(n=n, seed=seed, merge=".false.", fmt=fmt), tmp_path, name=name + ".in")
        assert r.returncode == 0, r.stdout + r.stderr
        out[name] = {f: rd(str(tmp_path / (name + f + ext))) for f in ("", "_obs", "_stokes", "_obs2D", "_stokes_2D")}
    for n, seed in ((20000, 3), (60000, 4)):
        r = run(exe, MERGE_IN.format(n=n, seed=seed, merge=".true.", fmt=fmt), tmp_path, name="m.in")
        assert r.returncode == 0, r.stdout + r.stderr
    M = {f: rd(str(tmp_path / ("m" + f + ext))) for f in ("", "_obs", "_stokes", "_obs2D", "_stokes_2D")}
    w = lambda x, y: (np.asarray(x) * 20000 + np.asarray(y) * 60000) / 80000
    for col in ("Jout", "Jin"):
        assert np.allclose(M[""]["Spectrum"][1][col], w(out["a"][""]["Spectrum"][1][col], out["b"][""]["Spectrum"][1][col]), rtol=1e-6, atol=1e-12), col
    assert np.allclose(M[""]["Jmu"][1], w(out["a"][""]["Jmu"][1], out["b"][""]["Jmu"][1]), rtol=1e-6, atol=1e-12)
    for f, secs in (("_obs", ("Scattered", "Direct")), ("_stokes", ("Stokes_I", "Stokes_Q", "Stokes_U")), ("_obs2D", ("Scattered", "Direct")), ("_stokes_2D", ("Stokes_I", "Stokes_Q"))):
        for sec in secs:
            A, B, C_ = out["a"][f][sec][1], out["b"][f][sec][1], M[f][sec][1]
            assert np.allclose(C_, w(A, B), rtol=1e-5, atol=1e-9 * np.abs(C_).max()), (f, sec)
    k = {a.lower(): b for a, b in M[""]["Spectrum"][0].items()}
    ka = {a.lower(): b for a, b in out["a"][""]["Spectrum"][0].items()}; kb = {a.lower(): b for a, b in out["b"][""]["Spectrum"][0].items()}
    assert float(k["nphotons"]) == 80000 and np.isclose(float(k["nsc_tot"]), (float(ka["nsc_tot"]) * 20000 + float(kb["nsc_tot"]) * 60000) / 80000, rtol=1e-6)
    # the radial profiles are recomputed from the merged images
    ra, rb, rm = out["a"]["_obs"]["RadialI"][1]["I"], out["b"]["_obs"]["RadialI"][1]["I"], M["_obs"]["RadialI"][1]["I"]
    assert np.allclose(rm, w(ra, rb), rtol=1e-5, atol=1e-9 * np.abs(rm).max())


AMR_IN = """&parameters
 par%use_amr_grid  = .true.
 par%amr_type      = 'generic'
 par%amr_file      = 'sphere_small.dat'
 par%no_photons    = 3e4
 par%temperature   = 1.0e4
 par%taumax        = 1.0e2
 par%rmax          = 1.0
 par%nxfreq = 121
 par%calc_J = .true., par%calc_P = .true., par%calc_Pnew = .true.
 par%iseed = 3
/
"""


def write_amr(tmp_path):
    from lart_amd import amr
    amr.write_generic_amr(str(tmp_path / "sphere_small.dat"), amr.make_amr_sphere(level_min=2, level_max=5))


def test_amr_dry_run_equals_python_host(exe, tmp_path):
    write_amr(tmp_path)
    r = run(exe, AMR_IN, tmp_path, "--dry-run", name="amr.in")
    assert r.returncode == 0, r.stdout + r.stderr
    from lart_amd import host, amr
    par = host.read_namelist(str(tmp_path / "amr.in")); par.amr_file = str(tmp_path / "sphere_small.dat")
    par, line, _ = host.read_input(par)
    g, sg = amr.grid_create_amr(par, line, par.amr_file)
    assert re.search(r"AMR nleaf\s+:\s+%d\b" % g.nleaf, r.stdout)
    m = re.search(r"AMR N\(HI\)_pole\s+:\s+(\S+)", r.stdout); assert abs(float(m.group(1)) / par.N_gasmax - 1) < 1e-4
    m = re.search(r"xfreq_min, xfreq_max, nxfreq :\s+(\S+)\s+(\S+)\s+(\d+)", r.stdout)
    assert abs(float(m.group(1)) - sg.xfreq_min) < 1e-3 and abs(float(m.group(2)) - sg.xfreq_max) < 1e-3 and int(m.group(3)) == 121


@pytest.mark.gpu
def test_amr_end_to_end_matches_python_host(exe, tmp_path):
    write_amr(tmp_path)
    r = run(exe, AMR_IN, tmp_path, name="amr.in")
    assert r.returncode == 0, r.stdout + r.stderr
    hdus = read_h5(str(tmp_path / "amr.h5"))           # par%file_format defaults to 'hdf5' like the reference
    assert list(hdus)[:2] == ["Spectrum", "Jx_1D"] and hdus["Spectrum"][0]["EXTNAME"] == "Spectrum"
    from lart_amd import host
    par = host.read_namelist(str(tmp_path / "amr.in")); par.amr_file = str(tmp_path / "sphere_small.dat")
    sim = host.Simulation(par)
    sim.run_simulation()
    t = sim.output_normalize()
    np.testing.assert_allclose(hdus["Spectrum"][1]["Jout"], t["Jout"], rtol=1e-9, atol=1e-15)
    assert np.allclose(hdus["Jx_1D"][1], t["J1"], rtol=1e-8, atol=1e-12 * t["J1"].max())
    assert np.allclose(hdus["Pa_1D"][1].ravel(), t["P1"], rtol=1e-8) and np.allclose(hdus["Pa_1D_new"][1].ravel(), t["P1_new"], rtol=1e-8)


PERIODIC_AMR_IN = ("&parameters\n par%use_amr_grid = .true.\n par%amr_file = 'slab.dat'\n par%no_photons = 20000\n par%taumax = 1e2\n par%temperature = 1e4\n"
                   " par%xy_periodic = .true.\n par%nxfreq = 61\n par%xfreq_min = -12.0\n par%xfreq_max = 12.0\n par%calc_P = .true.\n par%calc_J = .true.\n"
                   " par%rmax = 1.0\n par%iseed = 23\n/\n")


def test_periodic_octree_dry_run(exe, tmp_path):
    """par%xy_periodic with an octree (raytrace_amr.f90:176-187): accepted, plane profiles by default (setup.f90:446-447)"""
    from lart_amd import amr
    amr.write_generic_amr(str(tmp_path / "slab.dat"), amr.make_amr_slab(level_min=2, level_max=3))
    r = run(exe, PERIODIC_AMR_IN, tmp_path, "--dry-run-write", name="pamr.in")
    assert r.returncode == 0, r.stdout + r.stderr
    assert re.search(r"AMR nleaf\s+:\s+176\b", r.stdout) and "geometry_JPa = -1" in r.stdout
    H = read_h5(str(tmp_path / "pamr.h5"))
    assert H["Spectrum"][0]["xy_per"] == 1 and H["Pa_1D"][1].size == 8 and H["Jx_1D"][1].shape == (61, 8)


@pytest.mark.gpu
def test_periodic_octree_end_to_end_matches_python_host(exe, tmp_path):
    from lart_amd import amr, host
    amr.write_generic_amr(str(tmp_path / "slab.dat"), amr.make_amr_slab(level_min=2, level_max=3))
    r = run(exe, PERIODIC_AMR_IN, tmp_path, name="pamr.in")
    assert r.returncode == 0, r.stdout + r.stderr
    H = read_h5(str(tmp_path / "pamr.h5"))
    par = host.read_namelist(str(tmp_path / "pamr.in")); par.amr_file = str(tmp_path / "slab.dat")
    sim = host.Simulation(par); sim.run_simulation(); t = sim.output_normalize()
    np.testing.assert_allclose(H["Spectrum"][1]["Jout"], t["Jout"], rtol=1e-9, atol=1e-15)
    assert np.allclose(H["Pa_1D"][1].ravel(), t["P1"], rtol=1e-8) and np.allclose(H["Jx_1D"][1], t["J1"], rtol=1e-8, atol=1e-12 * t["J1"].max())
    assert np.isclose(H["Spectrum"][1]["Jout"].sum() * (24.0 / 61) * 2.0 * np.pi * 2.0, 1.0, rtol=1e-9)     # every packet leaves through z: two faces, no area


def test_output_containers_hdf5_and_gzip_fits(exe, tmp_path):
    """the same sections, columns and keywords in both containers (reference: hdf5io_mod.f90:8-13, write_output_rect.f90:427-498)"""
    base = ("&parameters\n par%no_photons = 1000\n par%temperature = 1e4\n par%taumax = 1e2\n par%nx = 11\n par%ny = 11\n par%nz = 11\n"
            " par%rmax = 1.0\n par%nxfreq = 41\n par%nxim = 8\n par%nyim = 8\n par%use_stokes = .true.\n par%save_Jmu = .true.\n par%nmu = 5\n")
    r = run(exe, base + " par%file_format = 'fits'\n/\n", tmp_path, "--dry-run-write", name="cf.in")
    assert r.returncode == 0, r.stdout + r.stderr
    r = run(exe, base + "/\n", tmp_path, "--dry-run-write", name="ch.in")          # default container: HDF5
    assert r.returncode == 0, r.stdout + r.stderr
    F = read_fits(str(tmp_path / "cf.fits.gz")); H = read_h5(str(tmp_path / "ch.h5"))
    assert list(F) == list(H) == ["Spectrum", "Jmu"]
    assert list(F["Spectrum"][1]) == list(H["Spectrum"][1]) == ["Xfreq", "velocity", "wavelength", "Jout", "Jin"]
    assert np.array_equal(F["Spectrum"][1]["Xfreq"], H["Spectrum"][1]["Xfreq"]) and len(H["Spectrum"][1]["Xfreq"]) == 41
    fk = {k.lower(): v for k, v in F["Spectrum"][0].items()}
    for k in ("taumax", "voigta", "dfreq", "xfreq1"):
        assert np.isclose(float(fk[k]), float({kk.lower(): vv for kk, vv in H["Spectrum"][0].items()}[k]), rtol=1e-9), k
    assert H["Spectrum"][0]["taumax"] == 100.0 and H["Spectrum"][0]["nx"] == 11 and H["Spectrum"][0]["xyz_sym"] == 0
    # the full keyword set of the Spectrum section, in the reference's order (write_output_rect.f90:448-497; EXTNAME is the group name /
    # the FITS extension name here), + the host's ISEED
    want = ["ExeTime", "Nprocs", "recoil", "coreskip", "xyz_sym", "xy_per", "save_all", "save_Jin", "save_Jab", "nphotons", "taumax", "tauhomo", "Ngasmax",
            "Ngashomo", "temp", "Vexp", "DGR", "atau3", "voigta", "Xfreq1", "Xfreq2", "Dxfreq", "Dwave", "I_unit", "Dfreq", "Nsc_dust", "Nsc_gas", "Nsc_tot",
            "nx", "ny", "nz", "xmax", "ymax", "zmax", "calc_P", "calc_Pnew", "calc_J", "ISEED"]
    fits_keys = [k for k in F["Spectrum"][0] if k.upper() in {w.upper()[:8] for w in want}]          # FITS keywords hold 8 characters (calc_Pnew -> calc_Pne)
    assert [k.upper() for k in fits_keys] == [w.upper()[:8] for w in want]
    assert sorted(k.lower() for k in H["Spectrum"][0] if k.lower() != "extname") == sorted(w.lower() for w in want)
    hk = {k.lower(): v for k, v in H["Spectrum"][0].items()}
    assert hk["nphotons"] == 1000 and hk["temp"] == 1e4 and hk["nprocs"] == 1 and hk["i_unit"] == 0 and hk["coreskip"] == 0 and hk["save_jin"] == 1
    assert 0 < hk["iseed"] < 2.0 ** 53 and hk["iseed"] == int(hk["iseed"])   # par%iseed = 0 in this input: drawn from the operating system and recorded exactly
    assert np.isclose(hk["dxfreq"], (hk["xfreq2"] - hk["xfreq1"]) / 41) and np.isclose(hk["atau3"], (hk["voigta"] * 100.0) ** (1.0 / 3.0), rtol=1e-6)
    assert H["Jmu"][1].shape == (41, 5) and F["Jmu"][1].shape == (41, 5)
    S = read_h5(str(tmp_path / "ch_stokes.h5")); O = read_h5(str(tmp_path / "ch_obs.h5"))
    assert list(S) == ["Stokes_I", "Stokes_Q", "Stokes_U", "Stokes_V"] and S["Stokes_Q"][1].shape == (41, 8, 8)
    assert list(O) == ["Scattered", "Direct"] and O["Direct"][1].shape == (41, 8, 8)
    import gzip
    raw = gzip.open(str(tmp_path / "cf.fits.gz")).read()
    cards = [raw[i:i + 80].decode("ascii") for i in range(0, 2880 * 3, 80)]
    assert any(c.startswith("BITPIX  =") for c in cards) and any(c.startswith("NAXIS1  =") for c in cards) and any(c.startswith("taumax  =") for c in cards)
    assert not any(c[0] == " " and c[8:10] == "= " for c in cards)                # keyword names left-justified in columns 1-8 (FITS standard 4.1.2)
    FO = read_fits(str(tmp_path / "cf_obs.fits.gz"))
    assert list(FO) == ["Scattered", "Direct"]
    # keywords of the first section of a peel-off cube, in the reference's order (write_output_rect.f90:1093-1150)
    want3 = ["EQUINOX", "CTYPE1", "CUNIT1", "CRPIX1", "CRVAL1", "CD1_1", "CTYPE2", "CUNIT2", "CRPIX2", "CRVAL2", "CD2_2", "CTYPE3", "CUNIT3", "CRPIX3", "CRVAL3", "CD3_3",
             "DISTANCE", "DISTUNIT", "DIST_CM", "Xfreq1", "Xfreq2", "Dxfreq", "Dwave", "I_unit", "Dfreq", "nphotons", "alpha", "beta", "gamma", "obsx", "obsy", "obsz",
             "rot_cenx", "rot_ceny", "rot_cenz"]
    assert [k.upper() for k in FO["Scattered"][0] if k.upper() in {w.upper() for w in want3}] == [w.upper() for w in want3]
    assert sorted(k.lower() for k in O["Scattered"][0] if k.lower() != "extname") == sorted(w.lower() for w in want3)
    ok = {k.lower(): v for k, v in O["Scattered"][0].items()}
    assert ok["ctype1"] == "WAVE" and ok["crpix2"] == 4.5 and ok["rot_cenx"] == 0.0 and ok["nphotons"] == 1000 and ok["dist_cm"] > 0
    r = run(exe, base + " par%DGR = 1.0\n par%use_stokes = .false.\n par%save_peeloff_2D = .true.\n par%save_direc0 = .true.\n par%save_dust_scattered = .true.\n/\n", tmp_path,
            "--dry-run-write", name="cd.in")
    assert r.returncode == 0, r.stdout + r.stderr
    assert list(read_h5(str(tmp_path / "cd_obs.h5"))) == ["Scattered", "Direct", "Direct0", "ScatDust"]
    O2 = read_h5(str(tmp_path / "cd_obs2D.h5"))
    assert list(O2) == ["Scattered", "Direct", "Direct0"]
    # the (nxim, nyim) images carry a two-axis WCS of their own (write_output_rect.f90:829-853)
    want2 = ["EQUINOX", "CD1_1", "CD2_1", "CD1_2", "CD2_2", "CRPIX1", "CRPIX2", "CRVAL1", "CRVAL2", "CTYPE1", "CTYPE2", "DISTANCE", "DISTUNIT", "DIST_CM", "nphotons",
             "alpha", "beta", "gamma", "obsx", "obsy", "obsz", "rot_cenx", "rot_ceny", "rot_cenz"]
    assert sorted(k.lower() for k in O2["Scattered"][0] if k.lower() != "extname") == sorted(w.lower() for w in want2)
    k2 = {k.lower(): v for k, v in O2["Scattered"][0].items()}
    assert k2["ctype1"] == "RA--TAN" and k2["ctype2"] == "DEC-TAN" and k2["cd2_1"] == 0.0 and k2["crpix1"] == 4.5
    # par%out_bitpix = -32: float32 storage in both containers (define.f90:418); -64 / 0 keep float64
    r = run(exe, base + " par%out_bitpix = -32\n/\n", tmp_path, "--dry-run-write", name="c32.in")
    assert r.returncode == 0, r.stdout + r.stderr
    import ctypes
    raw = open(str(tmp_path / "c32.h5"), "rb").read()
    H32 = read_h5(str(tmp_path / "c32.h5"))
    assert np.allclose(H32["Spectrum"][1]["Xfreq"], H["Spectrum"][1]["Xfreq"], rtol=1e-6) and not np.array_equal(H32["Spectrum"][1]["Xfreq"], H["Spectrum"][1]["Xfreq"])
    assert len(raw) < os.path.getsize(str(tmp_path / "ch.h5"))
    r = run(exe, base + " par%out_bitpix = -32\n par%file_format = 'fits'\n/\n", tmp_path, "--dry-run-write", name="f32.in")
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(str(tmp_path / "f32.fits.gz"))


JPA_IN = ("&parameters\n par%no_photons = 20000\n par%temperature = 1e4\n par%taumax = 1e3\n par%nx = 21\n par%ny = 21\n par%nz = 21\n"
          " par%rmax = 1.0\n par%nxfreq = 61\n par%calc_J = .true.\n par%calc_P = .true.\n par%calc_Pnew = .true.\n"
          " par%geometry_JPa = {geo}\n par%xyz_symmetry = {sym}\n par%iseed = 7\n/\n")


@pytest.mark.parametrize("geo,sym", [(1, ".false."), (1, ".true."), (3, ".true."), (-1, ".false.")])
def test_cartesian_profile_bins_equal_python_host(exe, tmp_path, geo, sym):
    """create_JPa_mem (grid_mod_car.f90:1242-1434): same bin count and the same number of gas cells behind the bins"""
    r = run(exe, JPA_IN.format(geo=geo, sym=sym), tmp_path, "--dry-run", name="jpa.in")
    assert r.returncode == 0, r.stdout + r.stderr
    from lart_amd import host
    par = host.read_namelist(str(tmp_path / "jpa.in"))
    par, line, _ = host.read_input(par)
    g = host.grid_create(par, line)
    jp = host.create_JPa_mem(par, g)
    m = re.search(r"profile bins: geometry_JPa = (-?\d+)\s+nbins = (\d+)\s+sum\(ncount\) =\s+(\S+)", r.stdout)
    assert m, r.stdout
    assert int(m.group(1)) == geo and int(m.group(2)) == jp.nbins and abs(float(m.group(3)) / jp.ncount.sum() - 1) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("geo,sym", [(1, ".true."), (-1, ".false.")])
def test_cartesian_profiles_end_to_end_match_python_host(exe, tmp_path, geo, sym):
    r = run(exe, JPA_IN.format(geo=geo, sym=sym), tmp_path, name="jpa.in")
    assert r.returncode == 0, r.stdout + r.stderr
    H = read_h5(str(tmp_path / "jpa.h5"))
    assert list(H) == ["Spectrum", "Jx_1D", "Pa_1D", "Pa_1D_new"]
    from lart_amd import host
    sim = host.Simulation(host.read_namelist(str(tmp_path / "jpa.in")))
    sim.run_simulation()
    t = sim.output_normalize()
    assert t["J1"].max() > 0 and t["P1"].max() > 0
    np.testing.assert_allclose(H["Spectrum"][1]["Jout"], t["Jout"], rtol=1e-9, atol=1e-15)
    assert np.allclose(H["Jx_1D"][1], t["J1"], rtol=1e-8, atol=1e-12 * t["J1"].max())
    assert np.allclose(H["Pa_1D"][1].ravel(), t["P1"], rtol=1e-8) and np.allclose(H["Pa_1D_new"][1].ravel(), t["P1_new"], rtol=1e-8)


HEAL_IN = ("&parameters\n par%no_photons = 5000\n par%temperature = 1e4\n par%taumax = 1e2\n par%nx = 21\n par%ny = 21\n par%nz = 21\n"
           " par%rmax = 1.0\n par%nxfreq = 41\n par%nside = 4\n par%obsx = 0.0, 0.3\n par%obsy = 0.0, 0.1\n par%obsz = 0.0, -0.2\n"
           " par%use_stokes = .true.\n par%iseed = 11\n/\n")


def test_healpix_inside_observers_dry_run(exe, tmp_path):
    """observer_create_inside (observer_heal.f90:24-80): nobs from the finite obsx/obsy/obsz triples, npix = 12 nside^2"""
    r = run(exe, HEAL_IN, tmp_path, "--dry-run-write", name="heal.in")
    assert r.returncode == 0, r.stdout + r.stderr
    assert re.search(r"HEALPix inside observers: nside = 4\s+npix = 192", r.stdout) and re.search(r"observers:\s+2\b", r.stdout)
    for k in (1, 2):                                                     # one file per observer, (nxfreq, npix) maps
        O = read_h5(str(tmp_path / ("heal_obs_%03d.h5" % k)))
        assert O["Scattered"][1].shape == (41, 192) and O["Direct"][1].shape == (41, 192) and O["Scattered"][0]["NSIDE"] == 4
        hk = {a.lower() for a in O["Scattered"][0]} - {"extname"}            # write_output_heal.f90:757-771 (+ NSIDE/NPIX): no WCS, no alpha/beta/gamma
        assert hk == {"equinox", "distunit", "dist_cm", "xfreq1", "xfreq2", "dxfreq", "dwave", "i_unit", "dfreq", "nphotons", "obsx", "obsy", "obsz", "nside", "npix"}
    assert not os.path.exists(tmp_path / "heal_stokes_001.h5")               # setup.f90:183: no Stokes vectors for an inside observer


@pytest.mark.gpu
def test_healpix_end_to_end_matches_python_host(exe, tmp_path):
    r = run(exe, HEAL_IN, tmp_path, name="heal.in")
    assert r.returncode == 0, r.stdout + r.stderr
    from lart_amd import host
    sim = host.Simulation(host.read_namelist(str(tmp_path / "heal.in")))
    sim.run_simulation()
    t = sim.output_normalize()
    assert t["scatt"].max() > 0 and t["direc"].max() > 0
    for io in (0, 1):
        O = read_h5(str(tmp_path / ("heal_obs_%03d.h5" % (io + 1))))
        for sec, k in (("Scattered", "scatt"), ("Direct", "direc")):
            assert np.allclose(O[sec][1], t[k][:, :, 0, io], rtol=1e-8, atol=1e-12 * t[k].max()), sec


SIGHT_IN = ("&parameters\n par%no_photons = 2000\n par%temperature = 1e4\n par%taumax = 1e3\n par%nx = 21\n par%ny = 21\n par%nz = 21\n"
            " par%rmax = 1.0\n par%nxfreq = 70\n par%nxim = 12\n par%nyim = 9\n par%alpha = 30.0, -100.0\n par%beta = 50.0, 120.0\n"
            " par%velocity_type = 'hubble'\n par%Vexp = 40.0\n par%DGR = 1.0\n par%save_sightline_tau = .true.\n par%iseed = 3\n{extra}/\n")


def test_sightline_files_dry_run(exe, tmp_path):
    """write_sightline_tau_outside (sightline_tau_rect.f90:347-407): <base>_tau_kkk per observer, sections and WCS keywords"""
    r = run(exe, SIGHT_IN.format(extra=""), tmp_path, "--dry-run-write", name="sl.in")
    assert r.returncode == 0, r.stdout + r.stderr
    for k in (1, 2):
        H = read_h5(str(tmp_path / ("sl_tau_%03d.h5" % k)))
        assert list(H) == ["TAU_gas", "N_gas", "TAU_dust"]
        assert H["TAU_gas"][1].shape == (70, 12, 9) and H["N_gas"][1].shape == (12, 9) and H["TAU_dust"][1].shape == (12, 9)
        kw = {a.upper(): b for a, b in H["TAU_gas"][0].items()}
        assert kw["CTYPE1"] == "RA--TAN" and kw["CRPIX1"] == 6.5 and kw["CRPIX2"] == 5.0 and kw["EQUINOX"] == 2000
    r = run(exe, SIGHT_IN.format(extra=" par%file_format = 'fits'\n"), tmp_path, "--dry-run-write", name="sf.in")
    assert r.returncode == 0, r.stdout + r.stderr
    F = read_fits(str(tmp_path / "sf_tau_002.fits.gz"))
    assert list(F) == ["TAU_gas", "N_gas", "TAU_dust"] and F["TAU_gas"][1].shape == (70, 12, 9)


@pytest.mark.gpu
def test_sightline_end_to_end_matches_python_host(exe, tmp_path):
    r = run(exe, SIGHT_IN.format(extra=""), tmp_path, name="sl.in")
    assert r.returncode == 0, r.stdout + r.stderr
    from lart_amd import host
    sim = host.Simulation(host.read_namelist(str(tmp_path / "sl.in")))
    tg, ng, td = sim.make_sightline_tau()
    assert ng.max() > 0 and td.max() > 0
    for k in (0, 1):
        H = read_h5(str(tmp_path / ("sl_tau_%03d.h5" % (k + 1))))
        assert np.allclose(H["TAU_gas"][1], tg[..., k], rtol=1e-12, atol=0) and np.allclose(H["N_gas"][1], ng[..., k], rtol=1e-13, atol=0)
        assert np.allclose(H["TAU_dust"][1], td[..., k], rtol=1e-13, atol=0)
    assert os.path.exists(tmp_path / "sl.h5")                           # the transport run still follows


def test_xy_symmetry_dry_run_equals_python_host(exe, tmp_path):
    """grid_create with par%xy_symmetry (grid_mod_car.f90:113-134): mirror-plane cells count half in the homogeneous depth"""
    text = ("&parameters\n par%no_photons = 1000\n par%temperature = 1e4\n par%taumax = 1e2\n par%nx = 21\n par%ny = 14\n par%nz = 31\n"
            " par%rmax = 1.0\n par%nxfreq = 41\n par%xy_symmetry = .true.\n/\n")
    r = run(exe, text, tmp_path, "--dry-run", name="xys.in")
    assert r.returncode == 0, r.stdout + r.stderr
    from lart_amd import host
    par, line, _ = host.read_input(host.read_namelist(str(tmp_path / "xys.in")))
    host.grid_create(par, line)
    m = re.search(r"tau_homo  \(H  I\) :\s+(\S+)", r.stdout)
    assert m and abs(float(m.group(1)) / par.tauhomo - 1) < 6e-4
    m = re.search(r"N\(H  I\)_pole     :\s+(\S+)", r.stdout)
    assert m and abs(float(m.group(1)) / par.N_gasmax - 1) < 6e-4
    r = run(exe, text.replace("/\n", " par%xs_point = 0.2\n/\n"), tmp_path, "--dry-run", name="xys2.in")
    assert r.returncode != 0 and "not supported" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("vel", [" par%velocity_type = 'rotating_galaxy_halo'\n par%Vrot = 100.0\n par%rinner = 0.2\n",
                                 " par%velocity_type = 'ssh'\n par%Vpeak = 120.0\n par%rpeak = 0.3\n par%DeltaV = -60.0\n",
                                 " par%velocity_type = 'hubble'\n par%Vexp = 80.0\n par%include_deuterium = .true.\n par%D_to_H_ratio = 0.02\n par%recoil = .true.\n",
                                 " par%velocity_type = 'hubble'\n par%Vexp = 40.0\n par%line_id = 'FeII_UV1'\n par%bturb = 4.0\n",
                                 " par%velocity_type = 'hubble'\n par%Vexp = 20.0\n par%save_all_photons = .true.\n"])
def test_velocity_fields_end_to_end_match_python_host(exe, tmp_path, vel):
    """the analytic velocity fields of grid_create (grid_mod_car.f90:786-946) as examples/rotating_galaxy_halo uses them"""
    text = ("&parameters\n par%no_photons = 20000\n par%temperature = 1e4\n par%N_HImax = 1e16\n par%rmax = 1.0\n par%nx = 31\n par%ny = 31\n par%nz = 31\n"
            " par%comoving_source = .false.\n par%spectral_type = 'monochromatic'\n par%velocity_min = -400.0\n par%velocity_max = 400.0\n par%nvelocity = 80\n"
            " par%nxim = 9\n par%nyim = 9\n par%distance = 1e4\n par%alpha = -90.0, -90.0\n par%beta = 0.0, 60.0\n par%iseed = 21\n" + vel + "/\n")
    r = run(exe, text, tmp_path, name="rot.in")
    assert r.returncode == 0, r.stdout + r.stderr
    from lart_amd import host
    sim = host.Simulation(host.read_namelist(str(tmp_path / "rot.in")))
    assert np.abs(sim.grid.vfx).max() > 1.0
    sim.run_simulation()
    t = sim.output_normalize()
    H = read_h5(str(tmp_path / "rot.h5"))
    np.testing.assert_allclose(H["Spectrum"][1]["Jout"], t["Jout"], rtol=1e-9, atol=1e-15)
    if "save_all_photons" in vel:                              # <base>_allph (write_output_allph): one row per photon, the columns of a non-Stokes point source
        P = read_h5(str(tmp_path / "rot_allph.h5"))["AllPhotons"][1]; a = sim.ctx.all_photons(20000)
        assert list(P) == ["r", "xfreq1", "xfreq2", "NSCATT_gas", "NSCATT_dust"] and all(np.array_equal(P[k], a[k]) for k in P)
    for k in (0, 1):
        O = read_h5(str(tmp_path / ("rot_obs_%03d.h5" % (k + 1))))
        assert np.allclose(O["Scattered"][1], t["scatt"][..., k], rtol=1e-6, atol=1e-12 * t["scatt"].max())


def test_whole_namelist_surface_is_readable(exe, tmp_path):
    """every field of the reference's params_type (define.f90:209-544) can be named in an input file: defaults pass, physics or
    inputs outside the path stop with 'not supported' instead of a namelist error (and names the reference lacks still fail)"""
    ok = ("&parameters\n par%temperature = 1e4\n par%taumax = 1e2\n par%f_H2 = 0.0\n par%cone_opening = 0.0\n par%clump_allow_overlap = .false.\n"
          " par%amr_snapnum = 0\n par%stellar_radius = 0.0\n par%EW_line = 0.0\n par%Omega = 0.0\n par%q = 1.0\n par%D_to_H_ratio = 1.5e-5\n/\n")
    r = run(exe, ok, tmp_path, "--dry-run", name="all.in")
    assert r.returncode == 0, r.stdout + r.stderr
    for bad in ("par%density_rscale = 0.5", "par%clump_radius_alpha = 1.0", "par%stellar_radius = 0.1",
                "par%xmin = -1.0", "par%ionization_model = 'file'", "par%phase_angle = 10.0, 20.0"):
        r = run(exe, "&parameters\n par%temperature = 1e4\n par%taumax = 1e2\n " + bad + "\n/\n", tmp_path, "--dry-run", name="bad.in")
        assert r.returncode != 0 and "not supported" in r.stdout and bad.split()[0][4:] in r.stdout, bad
    r = run(exe, "&parameters\n par%temperature = 1e4\n par%taumax = 1e2\n par%xy_periodic = .true.\n par%nx = 4\n par%ny = 4\n par%nz = 11\n par%Omega = 28.0\n/\n", tmp_path,
            "--dry-run", name="shear.in")                                                  # the shearing box is a parameter of the march since round 3 (VAR 8)
    assert r.returncode == 0, r.stdout + r.stderr
    r = run(exe, "&parameters\n par%metalz = 1.0\n/\n", tmp_path, "--dry-run", name="bad2.in")
    assert r.returncode != 0 and "ERROR reading &parameters" in r.stdout
    r = run(exe, "&parameters\n par%temperature = 1e4\n par%taumax = 1e2\n par%include_deuterium = .true.\n par%fine_structure = .true.\n/\n", tmp_path,
            "--dry-run", name="hd.in")                                                    # promoted to ly_alpha_HD, the doublet switched off
    assert r.returncode == 0 and "ly_alpha_HD" in r.stdout, r.stdout
    # round 4: the Fortran host drives ly_beta (line type 8) and the H2 lines as the Python host does; what the reference refuses is refused with its message
    r = run(exe, "&parameters\n par%temperature = 1e4\n par%taumax = 1e2\n par%line_id = 'ly_beta'\n par%nx = 11\n par%ny = 11\n par%nz = 11\n par%rmax = 1.0\n/\n", tmp_path,
            "--dry-run", name="lyb.in")
    assert r.returncode == 0 and "Line ID                   : ly_beta" in r.stdout, r.stdout + r.stderr
    r = run(exe, "&parameters\n par%temperature = 1e4\n par%taumax = 1e2\n par%line_id = 'ly_beta'\n par%use_stokes = .true.\n/\n", tmp_path, "--dry-run", name="lyb2.in")
    assert r.returncode != 0 and "ly_beta: Stokes polarization not yet supported" in r.stdout
    r = run(exe, "&parameters\n par%temperature = 1e4\n par%taumax = 1e2\n par%h2_model = 'simple'\n par%f_H2 = 1e-3\n/\n", tmp_path, "--dry-run", name="h2bad.in")
    assert r.returncode != 0 and "H2: Phase 1 supports only h2_model='neufeld'" in r.stdout
    r = run(exe, "&parameters\n par%temperature = 1e4\n par%taumax = 1e2\n par%h2_model = 'neufeld'\n/\n", tmp_path, "--dry-run", name="h2bad2.in")
    assert r.returncode != 0 and "par%f_H2 must be > 0" in r.stdout
    from lart_amd import host
    (tmp_path / "all.in").write_text(ok)
    host.read_namelist(str(tmp_path / "all.in"))
    (tmp_path / "bad.in").write_text("&parameters\n par%density_rscale = 0.5\n/\n")
    with pytest.raises(host.LartInputError, match="not supported"):
        host.read_namelist(str(tmp_path / "bad.in"))


@pytest.mark.parametrize("ext", ["fits.gz", "fits", "h5"])
def test_amr_binary_containers_equal_text(exe, tmp_path, ext):
    """generic_amr_read_binary (read_generic_amr.f90:186-279): the FITS binary table / HDF5 group written by the reference's
    python/AMR_grid tools gives the same octree and normalisation as the text table; an off-centre origin moves the box"""
    from lart_amd import amr, host
    d = amr.make_amr_sphere(level_min=2, level_max=5)
    amr.write_generic_amr(str(tmp_path / "sphere_small.dat"), d)
    if ext == "h5":
        amr.write_generic_amr_h5(str(tmp_path / "s.h5"), d)
    else:
        amr.write_generic_amr_fits(str(tmp_path / ("s." + ext)), d, single=(ext == "fits"))
    a = run(exe, AMR_IN, tmp_path, "--dry-run", name="amr.in")
    b = run(exe, AMR_IN.replace("sphere_small.dat", "s." + ext), tmp_path, "--dry-run", name="amrb.in")
    assert a.returncode == 0 and b.returncode == 0, b.stdout + b.stderr
    pick = lambda out: [ln for ln in out.splitlines() if ln.startswith(("AMR ", "xfreq_min"))]
    assert pick(a.stdout) == pick(b.stdout) and len(pick(a.stdout)) >= 4
    # the python mirror reads the same containers
    par = host.read_namelist(str(tmp_path / "amrb.in")); par.amr_file = str(tmp_path / ("s." + ext))
    par, line, _ = host.read_input(par)
    g, _ = amr.grid_create_amr(par, line, par.amr_file)
    assert g.nleaf == len(d["x"])
    # corner-based origin: the same leaves shifted by +1 in a box [0, 2]^3
    d2 = dict(d, x=d["x"] + 1.0, y=d["y"] + 1.0, z=d["z"] + 1.0)
    if ext == "h5":
        amr.write_generic_amr_h5(str(tmp_path / "o.h5"), d2, origin=[0.0, 0.0, 0.0])
    else:
        amr.write_generic_amr_fits(str(tmp_path / ("o." + ext)), d2, origin=[0.0, 0.0, 0.0])
    c = run(exe, AMR_IN.replace("sphere_small.dat", "o." + ext), tmp_path, "--dry-run", name="amro.in")
    assert c.returncode == 0, c.stdout + c.stderr
    assert [ln for ln in pick(c.stdout) if "nleaf" in ln] == [ln for ln in pick(a.stdout) if "nleaf" in ln]


@pytest.mark.gpu
def test_reference_example_amr_sphere_reproduces_the_shipped_log(exe, tmp_path):
    """examples/amr_sphere_generic: the parameters of sphere_amr_inside_test1M.in (the run behind log_amr_1M.txt) on the leaf
    table of the example's generator (levels 3..7 refined on the surface, written here as uniform_amr_sphere.fits.gz) through
    LaRT_amd.x: the log's header lines, <N_scatt> = 2.8225E+04 within the Monte-Carlo error of 1e5 photons, and the HEALPix files"""
    from lart_amd import amr
    amr.write_generic_amr_fits(str(tmp_path / "uniform_amr_sphere.fits.gz"), amr.make_amr_sphere())
    text = ("&parameters\n par%use_amr_grid = .true.\n par%amr_type = 'generic'\n par%amr_file = 'uniform_amr_sphere.fits.gz'\n par%no_photons = 1e5\n"
            " par%taumax = 1.0e4\n par%DGR = 0.0\n par%source_geometry = 'point'\n par%spectral_type = 'voigt'\n par%rmax = 1.0\n par%nxfreq = 121\n"
            " par%nside = 8\n par%obsx = 0.0\n par%obsy = 0.0\n par%obsz = 0.0\n par%save_peeloff_2D = .true.\n par%save_Jin = .true.\n par%nprint = 100000\n"
            " par%out_merge = .false.\n par%iseed = 1\n/\n")
    r = run(exe, text, tmp_path, name="sphere_amr_inside_test.in")
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    for line in ("AMR nleaf    : 178480", "AMR voigt_a      :   4.7186E-04", "AMR N(HI)_pole   :   1.6954E+17", "AMR tau_pole      :   1.0000E+04"):
        assert line in out, line
    m = re.search(r"Average Number of scattering :\s+(\S+)", out)
    assert m and abs(float(m.group(1)) / 2.8225e4 - 1) < 0.02
    O = read_h5(str(tmp_path / "sphere_amr_inside_test_obs.h5"))
    assert O["Scattered"][1].shape == (121, 768) and O["Scattered"][1].sum() > 0


@pytest.mark.parametrize("lid", ["CIV_1548", "NV_1239", "OVI_1032", "NaI_D", "CaII_HK", "MgII_2796", "SiIV_1394", "AlII_1671"])
def test_line_ids_set_up_like_the_python_host(exe, tmp_path, lid):
    """par%line_id for the single lines and S-P doublets (line_mod.f90:590-713): the log of the set-up (Voigt parameter, optical depth
    for a column density, frequency window of a continuum source: damping, cross0, vtherm1, wavelength0, DnuHK_Hz) equals the Python host's"""
    text = ("&parameters\n par%%temperature = 2e4\n par%%N_HImax = 3e13\n par%%rmax = 1.0\n par%%nx = 11\n par%%ny = 11\n par%%nz = 11\n par%%line_id = '%s'\n"
            " par%%spectral_type = 'continuum'\n par%%velocity_min = -1500.0\n par%%velocity_max = 500.0\n par%%nvelocity = 100\n/\n" % lid)
    r = run(exe, text, tmp_path, "--dry-run", name="line.in")
    assert r.returncode == 0, r.stdout + r.stderr
    from lart_amd import host
    par, line, _ = host.read_input(host.read_namelist(str(tmp_path / "line.in")))
    g = host.grid_create(par, line)
    val = lambda key: float([l for l in r.stdout.splitlines() if l.startswith(key)][0].split(":")[1].split()[0])
    assert np.isclose(val("voigt_a"), g.voigt_amean, rtol=2e-3) and np.isclose(val("tau_pole"), par.taumax, rtol=2e-3)
    xmin = float([l for l in r.stdout.splitlines() if l.startswith("xfreq_min")][0].split(":")[1].split()[0])
    assert np.isclose(xmin, par.xfreq_min, rtol=1e-5)
    assert "Line ID                   : " + lid in r.stdout
