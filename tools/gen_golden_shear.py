#!/usr/bin/env python3
"""Generate tests/golden/raytrace_shear.npz from the reference's own compiled raytrace_car.f90 (oracle/_ref/libref.so):
raytrace_to_tau_car_xyper_shear (:2677-2954) on a perturbed medium -- the xy-periodic march in which a packet that wraps around the x
boundary changes its y velocity by -+ par%Omega (photon%vfy_shear).  Rays with small optical depth per box length wrap several times;
starts on faces / period boundaries and axis-parallel rays are included, and a third of the rays start with a shear already picked up.
The edge march of this configuration is the plain raytrace_to_edge_car (setup.f90:941, :967-969): tau_edge comes from there."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden")
R = O.Ref
R.lib()
rng = np.random.default_rng(20261007)
OMEGA = 1.7
params = "nx=5 ny=7 nz=9 xmax=0.5 ymax=0.8 zmax=1 taumax=3 temperature=1e4 xy_periodic=1 geometry=rectangle DGR=1 cext_dust=1e-21 Omega=%r" % OMEGA
o = O.Oracle(params)
assert int(o.scalar("raytrace_variant")) == 8
n = [int(o.scalar(k)) for k in ("nx", "ny", "nz")]
o.set_array("rhokap", o.grid3("rhokap") * rng.uniform(0.5, 1.5, n))
o.set_array("Dfreq", o.grid3("Dfreq") * rng.uniform(0.8, 1.2, n))
o.set_array("voigt_a", o.grid3("voigt_a") * rng.uniform(0.8, 1.2, n))
for k in ("vfx", "vfy", "vfz"):
    o.set_array(k, rng.normal(0, 2.0, n))
o.set_array("rhokapD", o.grid3("rhokap") * 0.01)
R.set_grid_from(o)
g = o.scalar
nx, ny, nz = n
faces = [o.array("xface"), o.array("yface"), o.array("zface")]
rin, icin, tauin, shin, rout, icout, shout, tedge = [], [], [], [], [], [], [], []
for it in range(900):
    ic = [int(rng.integers(1, nx + 1)), int(rng.integers(1, ny + 1)), int(rng.integers(1, nz + 1))]
    pos = [g("xmin") + (ic[0] - 1 + rng.uniform()) * g("dx"), g("ymin") + (ic[1] - 1 + rng.uniform()) * g("dy"), g("zmin") + (ic[2] - 1 + rng.uniform()) * g("dz")]
    mu = rng.uniform(-1, 1) * (0.15 if it % 2 else 1.0); ph = rng.uniform(0, 2 * np.pi); st = np.sqrt(1 - mu * mu)      # half of the rays nearly in the plane: many wraps
    k = [st * np.cos(ph), st * np.sin(ph), mu]
    if it % 7 == 0:
        ax = int(rng.integers(0, 3)); pos[ax] = faces[ax][ic[ax] - 1]
    if it % 9 == 0:
        ax = int(rng.integers(0, 2)); ic[ax] = 1; pos[ax] = faces[ax][0]; k[ax] = -abs(k[ax])
    if it % 11 == 0:
        k = [0.0, 0.0, 1.0 if mu > 0 else -1.0]
    if it % 13 == 0:
        k = [1.0 if mu > 0 else -1.0, 0.0, 0.0]
    r = pos + k + [rng.normal(0, 3.0), 1.0]; icc = ic + [1]
    tau = rng.exponential(1.0) * (10 if it % 3 else 0.3)
    if it % 13 == 0:
        tau = min(tau, 2.0)                   # an x-parallel ray never leaves the box: keep the target reachable
    sh0 = 0.0 if it % 3 else float(rng.integers(-3, 4)) * OMEGA
    rr, icr, sh1 = R.raytrace_to_tau_shear(r, icc, tau, OMEGA, sh0)
    rin.append(r); icin.append(icc); tauin.append(tau); shin.append(sh0); rout.append(rr); icout.append(icr); shout.append(sh1)
    tedge.append(R.raytrace_to_edge(0, r, icc))
J = np.zeros(int(g("nxfreq"))); R.lib().ref_get_Jout(J.ctypes.data)
np.savez_compressed(os.path.join(OUT, "raytrace_shear.npz"), params=params, variant=8, Omega=OMEGA,
                    rhokap=o.array("rhokap"), voigt_a=o.array("voigt_a"), Dfreq=o.array("Dfreq"), vfx=o.array("vfx"), vfy=o.array("vfy"),
                    vfz=o.array("vfz"), rhokapD=o.array("rhokapD"),
                    r_in=np.array(rin), ic_in=np.array(icin), tau_in=np.array(tauin), shear_in=np.array(shin), r_out=np.array(rout), ic_out=np.array(icout),
                    shear_out=np.array(shout), tau_edge=np.array(tedge), Jout=J)
sh = np.array(shout) - np.array(shin)
print("shear: escaped", int(np.sum(np.array(icout)[:, 3] == 0)), "Jout sum", J.sum(), "rays that wrapped in x", int(np.sum(sh != 0)), "max |wraps|", np.abs(sh).max() / OMEGA)
