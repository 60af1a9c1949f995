"""The drop-in boundary without a GPU: the shared library loads, exports every symbol include/lart_gpu.h declares,
the ctypes structs match the C layout, and the host-side mirror builds the same tables as the oracle."""
import ctypes
import os
import re
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from lart_amd import capi
    return capi


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "lart_gpu.h")).read()
    declared = sorted(set(re.findall(r"\b(lart_(?:gpu|host)_\w+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = ctypes.CDLL(built.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), "missing export " + s
    from lart_amd import hostlib
    assert sorted(built.SYMBOLS + hostlib.SYMBOLS) == declared        # every declaration is bound: device ABI in capi.py, host set-up helpers in hostlib.py


def test_struct_layout_matches_header(built, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lart_gpu.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(lart_params), '
                   'offsetof(lart_params, xs_point), offsetof(lart_params, xcrit2), sizeof(lart_tallies), offsetof(lart_tallies, nphotons_done));'
                   'printf("%zu %zu %zu %zu\\n", offsetof(lart_params, Omega), offsetof(lart_params, E_conv), offsetof(lart_params, h2_p_scat), offsetof(lart_tallies, aux));return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sz = [int(t) for t in subprocess.check_output([str(exe)]).split()]
    P, T = built.LartParams, built.LartTallies
    assert sz == [ctypes.sizeof(P), P.xs_point.offset, P.xcrit2.offset, ctypes.sizeof(T), T.nphotons_done.offset,
                  P.Omega.offset, P.E_conv.offset, P.h2_p_scat.offset, T.aux.offset]


def test_no_device_fails_loudly(built):
    # this container has no GPU: the product path must refuse, not fall back
    if built.lib().lart_gpu_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(built.LartError, match="no HIP device"):
        built.Context(built.LartParams(), 0)


def test_product_does_not_touch_the_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "lart_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".f90", ".inc")):
                txt = open(os.path.join(dp, f)).read()
                assert "liblart_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


CASES = [dict(temperature=10.0, taumax=1e6, xyz_symmetry=True, nx=101, ny=101, nz=101, rmax=1.0, nxfreq=201),
         dict(temperature=10.0, taumax=1e4, use_stokes=True, xy_periodic=True, nx=1, ny=1, nz=201, hgg=0.6592, albedo=0.3899),
         dict(temperature=1e4, taumax=1e3, nx=30, ny=30, nz=30, rmax=1.0, nxim=16, nyim=16, use_stokes=True, Vexp=20.0, velocity_type="hubble"),
         dict(temperature=1e4, N_HI=1e19, nx=21, ny=21, nz=21, geometry="rectangle", nxim=8, nyim=12),
         dict(temperature=1e4, taumax=1e1, nx=21, ny=21, nz=31, xmax=1.0, ymax=1.0, zmax=1.5, geometry="rectangle", xy_symmetry=True, nxfreq=81),
         dict(temperature=1e4, tauhomo=1e1, nx=20, ny=14, nz=31, rmax=1.0, xy_symmetry=True, velocity_type="hubble", Vexp=30.0, nxfreq=81),
         dict(temperature=1e4, taumax=2e1, nx=6, ny=4, nz=21, xmax=0.5, ymax=0.3, zmax=1.0, geometry="rectangle", xy_periodic=True, nxfreq=81),
         dict(temperature=1e4, N_HImax=1e18, nx=21, ny=21, nz=21, rmax=1.0, velocity_type="rotating_solid_body", Vrot=100.0, velocity_min=-300.0, velocity_max=300.0, nxfreq=80),
         dict(temperature=1e4, N_HImax=1e18, nx=21, ny=21, nz=21, rmax=1.0, velocity_type="rotating_galaxy_halo", Vrot=100.0, rinner=0.2, velocity_min=-300.0, velocity_max=300.0, nxfreq=80),
         dict(temperature=1e4, taumax=1e3, nx=21, ny=21, nz=21, rmax=1.0, rmin=0.2, velocity_type="power_law", Vexp=80.0, velocity_alpha=0.5),
         dict(temperature=1e4, taumax=1e3, nx=21, ny=21, nz=21, rmax=1.0, rmin=0.2, velocity_type="linear_decelerate", Vexp=80.0),
         dict(temperature=1e4, taumax=1e3, nx=21, ny=21, nz=21, rmax=1.0, spectral_type="continuum", velocity_type="hubble", Vexp=-50.0),
         dict(temperature=1e4, taumax=1e3, nx=21, ny=21, nz=21, rmax=1.0, velocity_type="ssh", Vpeak=120.0, rpeak=0.3, DeltaV=-60.0, velocity_min=-400.0, velocity_max=400.0, nxfreq=80)]


@pytest.mark.parametrize("kw", CASES)
def test_host_tables_match_oracle(built, oracle_mod, kw):
    from lart_amd import host
    par = host.Params()
    for k, v in kw.items():
        host.set_param(par, k, v)
    par, line, _ = host.read_input(par)
    g = host.grid_create(par, line)
    ob = host.observer_create(par)
    o = oracle_mod.Oracle(**kw)
    for k in ("rhokap", "voigt_a", "Dfreq", "vfx", "vfy", "vfz"):
        assert np.allclose(getattr(g, k), o.grid3(k), rtol=1e-13, atol=0), k
    for k in ("dx", "dy", "dz", "xmin", "ymin", "zmin", "i0", "j0", "k0", "Dfreq_ref", "xfreq_min", "dxfreq", "nxfreq"):
        assert getattr(g, k) == o.scalar(k), k
    assert np.isclose(g.xcrit, o.scalar("xcrit"), rtol=1e-12)
    if ob.nobs:
        assert np.allclose(ob.xyz.ravel(), o.array("obs_xyz"), rtol=1e-14)
        assert np.allclose(ob.rmatrix.ravel(), o.array("rmatrix"), atol=1e-15)
        assert np.isclose(ob.dxim, o.scalar("dxim"), rtol=1e-13) and np.isclose(ob.dyim, o.scalar("dyim"), rtol=1e-13)


def test_namelist_surface(built, tmp_path):
    from lart_amd import host
    f = tmp_path / "t1tau6.in"
    f.write_text("&parameters\n par%no_photons  = 1e6\n par%temperature = 1.0e1\n par%taumax      = 1.0e6\n par%DGR = 0.0\n"
                 " par%xyz_symmetry    = .true.\n par%comoving_source = .false.\n par%spectral_type   = 'voigt'\n"
                 " par%nx = 101\n par%ny = 101\n par%nz = 101\n par%nxfreq = 201\n par%rmax   = 1.0\n par%nprint = 100000\n/\n")
    par = host.read_namelist(str(f))
    assert par.xyz_symmetry and not par.comoving_source and par.nx == 101 and par.taumax == 1e6 and par.no_photons == 1e6
    bad = tmp_path / "bad.in"
    bad.write_text("&parameters\n par%not_a_field = 1\n/\n")
    with pytest.raises(host.LartInputError):
        host.read_namelist(str(bad))
    fs = tmp_path / "fs.in"
    fs.write_text("&parameters\n par%density_rscale = 0.5\n/\n")
    with pytest.raises(host.LartInputError, match="not supported"):
        host.read_namelist(str(fs))
    fs.write_text("&parameters\n par%include_deuterium = .true.\n par%fine_structure = .true.\n par%temperature = 1e4\n par%taumax = 1e3\n/\n")
    par, line, _ = host.read_input(host.read_namelist(str(fs)))
    assert line.line_type == 7 and par.line_id == "ly_alpha_HD" and not par.fine_structure and line.nD_over_nH == 1.5e-5
    assert abs(line.delta_nu_HD_Hz / 6.7104e11 - 1) < 2e-3 and abs(line.ratio_Dfreq_HD / 1.41318 - 1) < 1e-3
    fs.write_text("&parameters\n par%fine_structure = .true.\n par%temperature = 10.0\n par%taumax = 1e3\n/\n")
    par, line, _ = host.read_input(host.read_namelist(str(fs)))
    assert line.line_type == 2 and abs(line.DnuHK_Hz / 1.0969049e10 - 1) < 1e-6
    clump = tmp_path / "clump.in"
    clump.write_text("&parameters\n par%use_clump_medium = .true.\n par%clump_radius = 0.01\n par%clump_f_cov = 5.0\n par%clump_tau0 = 1.0e3\n/\n")
    par = host.read_namelist(str(clump))
    assert par.use_clump_medium and par.clump_f_cov == 5.0 and par.clump_fully_inside
    amr = tmp_path / "amr.in"
    amr.write_text("&parameters\n par%use_amr_grid = .true.\n par%amr_type = 'generic'\n par%amr_file = 'leaves.dat'\n/\n")
    par = host.read_namelist(str(amr))
    assert par.use_amr_grid and par.amr_file == "leaves.dat"


def test_line_id_constants_of_the_python_host(golden):
    """host.read_input's table of single lines and doublets (setup_resonance_line, line_mod.f90:590-713) against the constants the
    reference's compiled module gives (tests/golden/line_ids.npz)"""
    from lart_amd import host
    z = golden("line_ids.npz")
    for lid, ref in zip(z["ids"], z["constants"]):
        par = host.Params(); par.line_id = str(lid); par.temperature = 1e4; par.taumax = 10.0; par.fine_structure = True
        par, L, _ = host.read_input(par)
        got = [L.line_type, L.wavelength0, L.damping, L.cross0, L.vtherm1, L.g_recoil0, L.DnuHK_Hz, L.E1, L.E2, L.E3]
        assert np.allclose(got, ref, rtol=1e-14, atol=0) and not par.fine_structure, (lid, got, ref)
    par = host.Params(); par.line_id = "ly_beta"; par.taumax = 10.0             # line type 8 since round 3 (line_mod.f90:330-372): 3p -> 2s conversion to H-alpha
    par, L, _ = host.read_input(par)
    assert L.line_type == 8 and abs(L.P_conv - 0.11833545951986842) < 1e-15 and abs(L.wavelength0_Ha - 0.6563) < 1e-3
    par = host.Params(); par.line_id = "ly_gamma"; par.taumax = 10.0
    with pytest.raises(host.LartInputError):
        host.read_input(par)
