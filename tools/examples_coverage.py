#!/usr/bin/env python3
"""Which of the reference's shipped inputs (examples/*/*.in) the Fortran host accepts: `LaRT_amd.x <input> --dry-run` (set-up only, no GPU)
on a scratch copy of every input, the reason of a refusal taken from the host's own message.  Needs /root/reference (this container only);
writes profiles/examples_coverage.md.  Inputs whose data files are not shipped (octree tables, density cubes) are reported as such."""
import glob
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
EXE = os.path.join(ROOT, "lart_amd", "fhost", "LaRT_amd.x")


def main():
    scratch = sys.argv[1] if len(sys.argv) > 1 else "/tmp/exrun"
    shutil.rmtree(scratch, ignore_errors=True)
    os.makedirs(os.path.join(scratch, "examples"))
    os.symlink(os.path.join(REF, "data"), os.path.join(scratch, "data"))
    rows = []
    for d in sorted(glob.glob(os.path.join(REF, "examples", "*"))):
        if not os.path.isdir(d):
            continue
        w = os.path.join(scratch, "examples", os.path.basename(d)); os.makedirs(w)
        for f in os.listdir(d):                                 # the inputs and whatever small data files sit next to them, as links
            if not f.endswith(".in"):
                os.symlink(os.path.join(d, f), os.path.join(w, f))
        for f in sorted(glob.glob(os.path.join(d, "*.in"))):
            shutil.copy(f, w)
            try:
                r = subprocess.run([EXE, os.path.basename(f), "--dry-run"], cwd=w, capture_output=True, text=True, timeout=90)
            except subprocess.TimeoutExpired:                   # 1e8 clumps and the like: the input is read and accepted, the set-up is just long on a busy CPU
                rows.append((os.path.basename(d), os.path.basename(f), True, "accepted (set-up not finished within 90 s here)")); continue
            out = r.stdout + r.stderr
            if r.returncode == 0:
                why = "accepted"
            else:
                m = re.search(r"(not supported[^\n]*|ERROR[^\n]*|cannot [^\n]*|No such file[^\n]*|io_resolve_filename[^\n]*)", out)
                why = (m.group(1) if m else out.strip().splitlines()[-1] if out.strip() else "rc %d" % r.returncode).strip()
            rows.append((os.path.basename(d), os.path.basename(f), r.returncode == 0, why[:150]))
    ok = sum(1 for r in rows if r[2])
    with open(os.path.join(ROOT, "profiles", "examples_coverage.md"), "w") as fh:
        fh.write("# Shipped inputs of the reference through `LaRT_amd.x --dry-run`\n\n`tools/examples_coverage.py` (needs /root/reference; set-up only, no GPU). "
                 "%d of %d inputs are accepted; the others stop with the message in the last column.\n\n| example | input | set-up | message |\n|---|---|---|---|\n" % (ok, len(rows)))
        for d, f, good, why in rows:
            fh.write("| %s | %s | %s | %s |\n" % (d, f, "ok" if good else "refused", ("" if why == "accepted" else why).replace("|", "/")))
    print("%d of %d accepted" % (ok, len(rows)))
    from collections import Counter
    for k, v in Counter(r[3] for r in rows if not r[2]).most_common(25):
        print("%4d  %s" % (v, k))


if __name__ == "__main__":
    main()
