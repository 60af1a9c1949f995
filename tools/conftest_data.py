"""the Mueller Ly-alpha table of tests/golden written back as the text file par%scatt_mat_file expects (setup.f90:597-627)"""
import os
import numpy as np


def mueller_path(tmpdir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, "tests", "golden", "mueller_Lyalpha.npz"))
    p = os.path.join(tmpdir, "mueller_Lyalpha.dat")
    h, tab = z["header"], z["table"]
    with open(p, "w") as f:
        f.write("lambda(um), Cext(cm^2/H), albedo, <cos>, # of angles\n")
        f.write("  %.5f  %.5E  %.5f  %.5f  %d\n" % (h[0], h[1], h[2], h[3], len(tab)))
        f.write("cos(theta),     S11,               S12,             S33,             S34\n")
        for r in tab:
            f.write("%.4f  %.10E  %.10E  %.10E  %.10E\n" % tuple(r))
    return p
