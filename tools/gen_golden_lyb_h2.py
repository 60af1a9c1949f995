#!/usr/bin/env python3
"""Generate tests/golden/lyb_h2.npz from the reference's own compiled modules (oracle/_ref/libref.so: line_mod, h2_mod, raytrace_car):

  * line type 8 ('ly_beta', line_mod.f90:1192-1240): the line constants, do_resonance8 (:293-345) on the MT19937-64 stream (uz, atom-frame
    frequency, polar angle, E1..E3, band after the event), raytrace_to_tau_car (:1425-1648) for band-1 and band-2 packets on a perturbed dusty
    medium (band 2: dust-only opacity rhokapD R_Ha, lab-frame frequency, Jout_Ha / W_esc2), raytrace_to_edge_car_dust_Ha (:510-582);
  * the H2 lines next to Ly-alpha (h2_mod.f90): h2_init at three temperatures (strengths, damping, branching, populations from data/h2/energy_X.dat),
    h2_kappa_D, h2_select_line_D on the MT stream, raytrace_to_tau_car / raytrace_to_edge_car with the H2 opacity switched on.

Also writes tests/golden/h2_energy_X.npz: the X-state level table (V, J, E) that h2_init reads at run time (a data file of the reference)."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden")
R = O.Ref
L = R.lib()
_p = lambda a: a.ctypes.data_as(C.c_void_p)
rng = np.random.default_rng(20261008)
out = {}

# ---- the level table (data)
rows = []
for ln in open("/root/reference/data/h2/energy_X.dat"):
    q = ln.strip()
    if not q or q.startswith("#") or "//" in q:
        continue
    t = q.split()
    try:
        rows.append((float(t[0]), float(t[1]), float(t[2])))
    except (ValueError, IndexError):
        pass
lev = np.array(rows)
np.savez_compressed(os.path.join(OUT, "h2_energy_X.npz"), V=lev[:, 0], J=lev[:, 1], E_cm=lev[:, 2],
                    source="data/h2/energy_X.dat of the reference (Komasa et al. 2011, Table 1, compiled by K. Kaplan): X-state rovibrational level energies [cm^-1]")

# ---- ly_beta: constants and do_resonance8
lc = R.set_line_id("ly_beta")
out["lyb_line"] = lc                      # line_type, wavelength0, damping, cross0, vtherm1, g_recoil0, DnuHK_Hz, E1, E2, E3
mult = R.get_multiplet(); out["lyb_multiplet"] = mult
params = "line_id=ly_beta nx=7 ny=6 nz=8 xmax=1 ymax=0.9 zmax=1.1 taumax=40 temperature=1e4 geometry=rectangle DGR=1 cext_dust=4e-16 cext_dust_Ha=1.5e-16"
o = O.Oracle(params)
n3 = [int(o.scalar(k)) for k in ("nx", "ny", "nz")]
o.set_array("rhokap", o.grid3("rhokap") * rng.uniform(0.5, 1.5, n3))
o.set_array("Dfreq", o.grid3("Dfreq") * rng.uniform(0.8, 1.2, n3))
o.set_array("voigt_a", o.grid3("voigt_a") * rng.uniform(0.8, 1.2, n3))
for k in ("vfx", "vfy", "vfz"):
    o.set_array(k, rng.normal(0, 2.0, n3))
o.set_array("rhokapD", o.grid3("rhokap") * rng.uniform(0.02, 0.2, n3))
R.set_grid_from(o)
L.ref_set_band2(o.scalar("xfreq_min_Ha"), o.scalar("dxfreq_Ha"), int(o.scalar("nxfreq_Ha")), o.scalar("R_Ha"))
xs = [0.0, 0.6, -1.7, 3.2, 11.0]
res = []
for x in xs:
    R.seed(77)
    r8 = np.zeros((400, 8)); L.ref_do_resonance8_v(400, float(x), 3, 2, 4, _p(r8)); res.append(r8)
out["res8_x"] = np.array(xs); out["res8"] = np.array(res); out["res8_cell"] = np.array([3, 2, 4]); out["res8_seed"] = 77
g = o.scalar
nx, ny, nz = n3
faces = [o.array("xface"), o.array("yface"), o.array("zface")]
rin, icin, tauin, band, rout, icout, tedge_ha = [], [], [], [], [], [], []
for it in range(600):
    ic = [int(rng.integers(1, nx + 1)), int(rng.integers(1, ny + 1)), int(rng.integers(1, nz + 1))]
    pos = [g("xmin") + (ic[0] - 1 + rng.uniform()) * g("dx"), g("ymin") + (ic[1] - 1 + rng.uniform()) * g("dy"), g("zmin") + (ic[2] - 1 + rng.uniform()) * g("dz")]
    mu = rng.uniform(-1, 1); ph = rng.uniform(0, 2 * np.pi); st = np.sqrt(1 - mu * mu)
    k = [st * np.cos(ph), st * np.sin(ph), mu]
    if it % 7 == 0:
        ax = int(rng.integers(0, 3)); pos[ax] = faces[ax][ic[ax] - 1]
    if it % 11 == 0:
        k = [0.0, 0.0, 1.0 if mu > 0 else -1.0]
    ib = 2 if it % 2 else 1
    r = pos + k + [rng.normal(0, 3.0), 1.0]; icc = ic + [1]
    tau = rng.exponential(1.0) * (3.0 if it % 3 else 0.2)
    rr = np.array(r, dtype=np.float64); ii = np.array(icc, dtype=np.int32)
    L.ref_raytrace_to_tau_band(ib, _p(rr), _p(ii), float(tau))
    te = C.c_double(); r0 = np.array(r, dtype=np.float64); i0 = np.array(icc, dtype=np.int32)
    L.ref_raytrace_to_edge_dust_Ha(_p(r0), _p(i0), C.byref(te))
    rin.append(r); icin.append(icc); tauin.append(tau); band.append(ib); rout.append(rr); icout.append(ii); tedge_ha.append(te.value)
JH = np.zeros(int(g("nxfreq_Ha"))); W2 = np.zeros(2); L.ref_get_band2(_p(JH), _p(W2))
J1 = np.zeros(int(g("nxfreq"))); L.ref_get_Jout(_p(J1))
out.update(lyb_params=params, lyb_rhokap=o.array("rhokap"), lyb_voigt_a=o.array("voigt_a"), lyb_Dfreq=o.array("Dfreq"), lyb_vfx=o.array("vfx"), lyb_vfy=o.array("vfy"),
           lyb_vfz=o.array("vfz"), lyb_rhokapD=o.array("rhokapD"), lyb_r_in=np.array(rin), lyb_ic_in=np.array(icin), lyb_tau_in=np.array(tauin), lyb_band=np.array(band),
           lyb_r_out=np.array(rout), lyb_ic_out=np.array(icout), lyb_tau_edge_Ha=np.array(tedge_ha), lyb_Jout_Ha=JH, lyb_W_esc=W2, lyb_Jout=J1)
print("ly_beta: P_down(2) =", mult[14 + 1 * 7 + 5], "band-2 rays", int(np.sum(np.array(band) == 2)), "escaped band 2 weight", W2[1], "band 1", W2[0], "Jout_Ha sum", JH.sum())

# ---- H2
R.set_line_id("ly_alpha")
h2 = []
cases = [(1000.0, 1e-3, 0, 0, -1.0), (3000.0, 0.3, 0, 0, -1.0), (5000.0, 0.5, 1, 1, 4.0)]
for T, f, hiw, pure, bt in cases:
    v = np.zeros(26); L.ref_h2_init(b"/root/reference/data/h2", T, f, hiw, pure, bt, _p(v)); h2.append(v)
out["h2_cases"] = np.array(cases); out["h2_init"] = np.array(h2)
T, f, hiw, pure, bt = cases[1]
v = np.zeros(26); L.ref_h2_init(b"/root/reference/data/h2", T, f, hiw, pure, bt, _p(v))
params2 = "nx=6 ny=7 nz=8 xmax=1 ymax=1 zmax=1 taumax=300 temperature=1e3 geometry=rectangle DGR=1 cext_dust=2e-19 h2_model=neufeld f_H2=%r h2_temperature=%r h2_data_dir=/root/reference/data/h2" % (f, T)
o2 = O.Oracle(params2)
n3 = [int(o2.scalar(k)) for k in ("nx", "ny", "nz")]
o2.set_array("rhokap", o2.grid3("rhokap") * rng.uniform(0.5, 1.5, n3))
o2.set_array("Dfreq", o2.grid3("Dfreq") * rng.uniform(0.8, 1.2, n3))
o2.set_array("voigt_a", o2.grid3("voigt_a") * rng.uniform(0.8, 1.2, n3))
for k in ("vfx", "vfy", "vfz"):
    o2.set_array(k, rng.normal(0, 1.0, n3))
o2.set_array("rhokapD", o2.grid3("rhokap") * 1e-3)
R.set_grid_from(o2)
Df = float(o2.array("Dfreq").mean())
xk = np.concatenate([np.linspace(-40, 10, 1201), rng.uniform(-40, 10, 300)])
kap = np.zeros_like(xk); L.ref_h2_kappa_v(len(xk), _p(xk), Df, _p(kap))
R.seed(91)
xsel = rng.uniform(-30, 5, 2000); ils = np.zeros(2000, dtype=np.int32); L.ref_h2_select_v(2000, _p(xsel), Df, _p(ils))
g = o2.scalar; nx, ny, nz = n3
rin, icin, tauin, rout, icout, tedge = [], [], [], [], [], []
for it in range(500):
    ic = [int(rng.integers(1, nx + 1)), int(rng.integers(1, ny + 1)), int(rng.integers(1, nz + 1))]
    pos = [g("xmin") + (ic[0] - 1 + rng.uniform()) * g("dx"), g("ymin") + (ic[1] - 1 + rng.uniform()) * g("dy"), g("zmin") + (ic[2] - 1 + rng.uniform()) * g("dz")]
    mu = rng.uniform(-1, 1); ph = rng.uniform(0, 2 * np.pi); st = np.sqrt(1 - mu * mu)
    k = [st * np.cos(ph), st * np.sin(ph), mu]
    xf = rng.uniform(-25, 3) if it % 2 else rng.normal(0, 3.0)                 # half of the rays near the H2 lines (x ~ -4 and ~ -27 at 1e3 K)
    r = pos + k + [xf, 1.0]; icc = ic + [1]
    tau = rng.exponential(1.0) * (5.0 if it % 3 else 0.3)
    rr, icr = R.raytrace_to_tau(0, r, icc, tau)
    rin.append(r); icin.append(icc); tauin.append(tau); rout.append(rr); icout.append(icr); tedge.append(R.raytrace_to_edge(0, r, icc))
J2 = np.zeros(int(g("nxfreq"))); L.ref_get_Jout(_p(J2))
out.update(h2_params=params2, h2_Dfreq_cell=Df, h2_x=xk, h2_kappa=kap, h2_xsel=xsel, h2_sel=ils, h2_sel_seed=91,
           h2_rhokap=o2.array("rhokap"), h2_voigt_a=o2.array("voigt_a"), h2_Dfreq=o2.array("Dfreq"), h2_vfx=o2.array("vfx"), h2_vfy=o2.array("vfy"), h2_vfz=o2.array("vfz"),
           h2_rhokapD=o2.array("rhokapD"), h2_r_in=np.array(rin), h2_ic_in=np.array(icin), h2_tau_in=np.array(tauin), h2_r_out=np.array(rout), h2_ic_out=np.array(icout),
           h2_tau_edge=np.array(tedge), h2_Jout=J2)
v0 = np.zeros(26); L.ref_h2_init(b"/root/reference/data/h2", -1.0, 0.0, 0, 0, -1.0, _p(v0))      # switch the lines off again
print("H2: lines", h2[1][:2], "strengths", h2[1][3], h2[1][9], "kappa max", kap.max(), "selected line 2 share", float(np.mean(ils == 2)))
np.savez_compressed(os.path.join(OUT, "lyb_h2.npz"), **out)
