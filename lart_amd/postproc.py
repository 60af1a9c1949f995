"""Post-processing of the normalised outputs (host side, numpy): the azimuthal means the reference writes next to its peel-off images."""
import numpy as np


def radial_profiles(par, g, out, io=0, use_2D=False, stokes=False):
    """make_radial_intensity / make_radial_stokes (output_sum_rect.f90:489-658) on the normalised tallies `out` of
    Simulation.output_normalize: columns r, I[, Q, U, V, pol] of the azimuthal means about the image centre"""
    unit = g.dwave if par.distance2cm != 1.0 else g.dxfreq
    nxi, nyi = par.nxim, par.nyim
    nr = (max(nxi, nyi) + 1) // 2
    r = (np.arange(1, nr + 1) - 0.5) / nr if nr % 2 == 0 else (np.arange(1, nr + 1) - 1.0) / (nr - 0.5)
    xx, yy = np.meshgrid(np.arange(1, nxi + 1) - (nxi + 1.0) / 2.0, np.arange(1, nyi + 1) - (nxi + 1.0) / 2.0, indexing="ij")
    rr = np.sqrt(xx ** 2 + yy ** 2)
    ir = (np.floor(rr) if nr % 2 == 0 else np.floor(rr + 0.5)).astype(int)          # 0-based bin
    ok = ir < nr
    cnt = np.bincount(ir[ok], minlength=nr).astype(float)

    def img(k):
        return out[k + "_2D"][:, :, io] if use_2D else out[k][:, :, :, io].sum(axis=0) * unit

    def mean(a):
        return np.where(cnt > 0, np.bincount(ir[ok], weights=a[ok], minlength=nr) / np.maximum(cnt, 1.0), 0.0)

    if not stokes:
        return np.column_stack([r, mean(img("scatt") + img("direc"))])
    with np.errstate(invalid="ignore", divide="ignore"):
        cosp = np.where(rr == 0.0, 1.0, yy / rr); sinp = np.where(rr == 0.0, 0.0, -xx / rr)
    c2, s2 = 2.0 * cosp ** 2 - 1.0, 2.0 * cosp * sinp
    if use_2D:
        I, Q, U, V = img("I"), img("Q") * c2 + img("U") * s2, -img("Q") * s2 + img("U") * c2, img("V")
    else:
        q3, u3 = out["Q"][:, :, :, io], out["U"][:, :, :, io]
        I, V = img("I"), img("V")
        Q = (q3 * c2[None] + u3 * s2[None]).sum(axis=0) * unit; U = (-q3 * s2[None] + u3 * c2[None]).sum(axis=0) * unit
    mI, mQ, mU, mV = mean(I), mean(Q), mean(U), mean(V)
    with np.errstate(invalid="ignore", divide="ignore"):
        pol = np.where(mI > 0.0, np.sqrt(mQ ** 2 + mU ** 2) / mI, 0.0)
    return np.column_stack([r, mI, mQ, mU, mV, pol])
