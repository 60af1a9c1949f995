#!/usr/bin/env python3
"""Golden spectra of BASELINE.json's configurations at their FULL optical depth, from the CPU oracle (oracle/lart_oracle.c,
the restatement of the reference pinned on the compiled reference modules), on the reference's own MT19937-64 stream
(rng_kind 0: seed + 9999*worker, run_simulation_mod.f90:150-151, random_mt.f90:940-953).

    python tools/gen_golden_bigtau.py run   c2 --workers 7 --chunk 200 --hours 8      # background workers, checkpoint per chunk
    python tools/gen_golden_bigtau.py merge c2                                        # -> tests/golden/c2_tau6.npz

Configurations (namelist physics of the shipped example files, only no_photons / iseed changed):
  c1  examples/slab/t1tau4.in          slab tau0 = 1e4, nz = 201, xy_periodic, use_stokes, save_Jin
  c2  examples/sphere/t1tau6.in        sphere tau0 = 1e6, T = 10 K, 101^3 octant (xyz_symmetry), nxfreq = 201
  c3  examples/sphere_peel/t1tau6.in   sphere tau0 = 1e6, 201^3, use_stokes, save_Jin, one far observer 129 x 129 x 201
  c4  examples/clump_sphere/clump_tau3_fcov5_peel.in (N_cl = 66 667 clumps of radius 0.01, tau_cl = 1e3, T = 1e4 K, one far observer
      129 x 129 x 301) + the dust lines of examples/sphere_peel/t1tau7dust.in:5-10 (Mueller Ly-alpha table, DGR = 1, use_stokes,
      save_Jabs) + core_skip (no shipped input combines clumps and dust)
  c5  examples/amr_sphere_generic/sphere_amr_static.in on the 178 480-leaf sphere of the example's generator (levels 3..7, refined
      on the surface; lart_amd.amr.make_amr_sphere), tau0 = 1e4, T = 1e4 K, calc_J + calc_P + calc_Pnew, geometry_JPa = 1, observer 100 x 100 x 121

Each worker process runs chunks of `chunk` photons with its own MT seed and after every chunk rewrites one checkpoint file
(raw tally sums + per-chunk sums of squares for the error estimate), so the fixture can be merged from whatever has
accumulated.  The fixture stores: raw and normalised Jout (output_sum_rect.f90:185,194-202), the per-bin standard error
from the scatter between chunks, <N_scatt> and its error, and for c3 the peel-integrated spectra and radial profiles.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c1": dict(temperature=10.0, taumax=1.0e4, Vexp=0.0, hgg=0.6592, albedo=0.3899, DGR=0.0, use_stokes=True, save_Jin=True,
               spectral_type="voigt", recoil=False, xy_periodic=True, nx=1, ny=1, nz=201, xmax=1.0, ymax=1.0, zmax=1.0),
    "c2": dict(temperature=10.0, taumax=1.0e6, DGR=0.0, Vexp=0.0, xyz_symmetry=True, comoving_source=False, recoil=False,
               spectral_type="voigt", nx=101, ny=101, nz=101, nxfreq=201, rmax=1.0),
    "c3": dict(temperature=10.0, taumax=1.0e6, DGR=0.0, Vexp=0.0, use_stokes=True, save_Jin=True, nx=201, ny=201, nz=201,
               rmax=1.0, nxfreq=201, nxim=129, nyim=129),
    # "MUELLER" / "AMR_SPHERE" stand for files written at run time from committed data (tests/golden/mueller_Lyalpha.npz) and from the
    # deterministic generator (lart_amd.amr.make_amr_sphere); iseed of c4 is the clump-placement stream (init_clumps), not the photon stream
    "c4": dict(use_clump_medium=True, rmax=1.0, clump_radius=0.01, clump_f_cov=5.0, clump_tau0=1.0e3, temperature=1.0e4, clump_sigma_v=0.0,
               spectral_type="voigt", nxfreq=301, velocity_min=-100.0, velocity_max=100.0, nx=11, ny=11, nz=11, nxim=129, nyim=129,
               scatt_mat_file="MUELLER", DGR=1.0, use_reduced_wgt=False, use_stokes=True, save_Jin=True, save_Jabs=True, core_skip=True, iseed=1),
    "c5": dict(use_amr_grid=True, amr_file="AMR_SPHERE", taumax=1.0e4, DGR=0.0, spectral_type="voigt", source_geometry="point",
               temperature=1.0e4, rmax=1.0, nxfreq=121, nxim=100, nyim=100, calc_J=True, calc_P=True, calc_Pnew=True),
}
SCRATCH = os.environ.get("LART_GOLDEN_SCRATCH", "/tmp/lart_golden")
FIXTURE = {"c1": "c1_slab_tau4.npz", "c2": "c2_tau6.npz", "c3": "c3_peel_tau6.npz", "c4": "c4_clump_dust.npz", "c5": "c5_amr_JP.npz"}
PEEL = {"c3": ("scatt", "direc", "I", "Q", "U"), "c4": ("scatt", "direc", "I", "Q", "U"), "c5": ("scatt", "direc")}


def materialize(kw, tmp=None):
    """replace the MUELLER / AMR_SPHERE placeholders of a parameter set by files written under tmp"""
    kw = dict(kw); tmp = tmp or SCRATCH
    os.makedirs(tmp, exist_ok=True)
    if kw.get("scatt_mat_file") == "MUELLER":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from conftest_data import mueller_path
        kw["scatt_mat_file"] = mueller_path(tmp)
    if kw.get("amr_file") == "AMR_SPHERE":
        from lart_amd import amr
        fn = os.path.join(tmp, "amr_sphere_3_7_%d.dat" % os.getpid())
        amr.write_generic_amr(fn, amr.make_amr_sphere())
        kw["amr_file"] = fn
    return kw


def radial_profile(img, nbin):
    """azimuthal average of an (nxim, nyim) image about its centre in nbin equal-width rings out to the half width"""
    nx, ny = img.shape
    x = np.arange(nx) - (nx - 1) / 2.0; y = np.arange(ny) - (ny - 1) / 2.0
    r = np.hypot(x[:, None], y[None, :]) / (nx / 2.0)
    k = np.minimum((r * nbin).astype(int), nbin)          # ring nbin = corners, dropped
    s = np.bincount(k.ravel(), weights=img.ravel(), minlength=nbin + 1)[:nbin]
    c = np.bincount(k.ravel(), minlength=nbin + 1)[:nbin]
    return s / np.maximum(c, 1)


def reductions(name, o, kw):
    """the small arrays one chunk contributes (raw sums, not normalised)"""
    out = {"Jout": o.array("Jout"), "nscatt": np.array([o.scalar("nscatt_gas") + o.scalar("nscatt_dust")]),
           "n_events": np.array([o.scalar("n_events")]), "n_cellsteps": np.array([o.scalar("n_cellsteps")]),
           "n_cellsteps_peel": np.array([o.scalar("n_cellsteps_peel")])}
    if kw.get("save_Jin"):
        out["Jin"] = o.array("Jin")
    if kw.get("save_Jabs"):
        out["Jabs"] = o.array("Jabs")
    if name == "c5":
        out["J1"] = o.array("J1"); out["P1"] = o.array("P1"); out["P1_new"] = o.array("P1_new")
    if name in PEEL:
        sh = (kw["nxfreq"], kw["nxim"], kw["nyim"])
        for t in PEEL[name]:
            c = o.array(t + ":0").reshape(sh, order="F")
            out[t + "_spec"] = c.sum(axis=(1, 2))
            out[t + "_img_r"] = radial_profile(c.sum(axis=0), 32)
            if t in ("Q", "U"):
                out[t + "_img"] = c.sum(axis=0).ravel()    # 129 x 129: the polarisation pattern changes sign over the image
    return out


def worker(name, wid, chunk, t_end, seed0):
    from oracle import oracle as O
    kw = materialize(CONFIGS[name])
    o = O.Oracle(**kw)
    os.makedirs(os.path.join(SCRATCH, name), exist_ok=True)
    path = os.path.join(SCRATCH, name, "w%02d.npz" % wid)
    acc, acc2, nchunk, nph, log_n = {}, {}, 0, 0, []
    if os.path.exists(path):                                # resume after an interruption
        d = np.load(path)
        nchunk, nph = int(d["nchunk"]), int(d["nphotons"]); log_n = list(d["log_nscatt"])
        for k in d.files:
            if k.startswith("sum_"): acc[k[4:]] = d[k].copy()
            if k.startswith("sq_"): acc2[k[3:]] = d[k].copy()
    while time.time() < t_end:
        seed = seed0 + 100003 * wid + nchunk                # one MT stream per chunk
        o.zero()
        o.run(chunk, first_id=1, stride=1, rng_kind=0, seed=seed, nthreads=1)
        r = reductions(name, o, kw)
        for k, v in r.items():
            acc[k] = acc.get(k, 0.0) + v
            acc2[k] = acc2.get(k, 0.0) + v * v
        ns, _ = o.photon_log()
        log_n.extend(ns.tolist())
        nchunk += 1; nph += chunk
        tmp = path + ".tmp.npz"
        np.savez(tmp, nchunk=nchunk, nphotons=nph, chunk=chunk, log_nscatt=np.array(log_n),
                 **{"sum_" + k: v for k, v in acc.items()}, **{"sq_" + k: v for k, v in acc2.items()})
        os.replace(tmp, path)


def run(a):
    import multiprocessing as mp
    t_end = time.time() + a.hours * 3600.0
    os.nice(19)
    ps = [mp.Process(target=worker, args=(a.config, w, a.chunk, t_end, a.seed)) for w in range(a.workers)]
    for p in ps: p.start()
    for p in ps: p.join()


def merge(a):
    from oracle import oracle as O
    name = a.config; kw = materialize(CONFIGS[name])
    files = sorted(glob.glob(os.path.join(SCRATCH, name, "w[0-9][0-9].npz")))
    if not files:
        raise SystemExit("no checkpoints under %s/%s" % (SCRATCH, name))
    acc, acc2, nchunk, nph, chunk, logn = {}, {}, 0, 0, None, []
    for f in files:
        d = np.load(f)
        nchunk += int(d["nchunk"]); nph += int(d["nphotons"]); chunk = int(d["chunk"]); logn.append(d["log_nscatt"])
        for k in d.files:
            if k.startswith("sum_"): acc[k[4:]] = acc.get(k[4:], 0.0) + d[k]
            if k.startswith("sq_"): acc2[k[3:]] = acc2.get(k[3:], 0.0) + d[k]
    logn = np.concatenate(logn)
    o = O.Oracle(**kw)
    nxf = int(o.scalar("nxfreq")); dxf = o.scalar("dxfreq")
    # output_normalize_outside: output_sum_rect.f90:185 (slab: area 2), :194-202 (sphere: 4 pi rmax^2 distance2cm^2)
    area = 2.0 if kw.get("xy_periodic") else 4.0 * np.pi * o.scalar("rmax") ** 2 * o.scalar("distance2cm") ** 2
    scale = nph * dxf * 2.0 * np.pi * area
    out = {"nphotons": nph, "nchunk": nchunk, "chunk": chunk, "xfreq": o.xfreq_axis(), "dxfreq": dxf, "Jout_scale_per_photon": dxf * 2.0 * np.pi * area,
           "nscatt_mean": logn.mean(), "nscatt_err": logn.std() / np.sqrt(len(logn)),
           "params": json.dumps(CONFIGS[name], sort_keys=True), "rng": "MT19937-64, seed0 + 100003*worker + chunk"}
    if name == "c4":
        out["cl_N"] = int(o.scalar("cl_N"))
    if name == "c5":
        out["nbins"] = int(o.scalar("nbins")); out["nleaf"] = int(o.scalar("amr_nleaf")); out["dr_JPa"] = o.scalar("amr_dr_JPa")
    for k, v in acc.items():
        mean_chunk = v / nchunk
        var_chunk = np.maximum(acc2[k] / nchunk - mean_chunk ** 2, 0.0) * nchunk / max(nchunk - 1, 1)
        out[k + "_raw"] = v                                           # raw weight sums over nphotons
        out[k + "_raw_err"] = np.sqrt(var_chunk * nchunk)             # standard error of the raw sum (scatter between chunks)
    out["Jout"] = out["Jout_raw"] / scale; out["Jout_err"] = out["Jout_raw_err"] / scale
    out["cells_per_event"] = float(acc["n_cellsteps"][0] / max(acc["n_events"][0], 1.0))
    if name in PEEL:
        out["peel_cells_per_event"] = float(acc["n_cellsteps_peel"][0] / max(acc["n_events"][0], 1.0))
    dst = os.path.join(ROOT, "tests", "golden", FIXTURE[name])
    np.savez_compressed(dst, **out)
    j = out["Jout"]; pk = j.max()
    print("%s: %d photons in %d chunks from %d workers; <N_scatt> = %.5g +- %.3g; peak Jout %.4g; Poisson rms of the fixture %.4f of the peak -> %s"
          % (name, nph, nchunk, len(files), out["nscatt_mean"], out["nscatt_err"], pk, np.sqrt(np.mean(out["Jout_err"] ** 2)) / pk, dst))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["run", "merge"])
    ap.add_argument("config", choices=sorted(CONFIGS))
    ap.add_argument("--workers", type=int, default=7)
    ap.add_argument("--chunk", type=int, default=200)
    ap.add_argument("--hours", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=20261004)
    a = ap.parse_args()
    (run if a.cmd == "run" else merge)(a)
