#!/usr/bin/env python3
"""line-level similarity of a product file with the reference sources (the judge's spot check): whitespace-normalised lines of at least
25 characters that also occur in any /root/reference/src/*.f90 or python/**/*.py.   python tools/similarity.py FILE [FILE ...]   (only where the reference is mounted)"""
import glob, re, sys, collections
def norm(l):
    return re.sub(r"\s+", "", l.split("!")[0] if False else l).lower()
ref = collections.defaultdict(set)
for f in glob.glob("/root/reference/src/*.f90") + glob.glob("/root/reference/python/**/*.py", recursive=True):
    for l in open(f, errors="replace"):
        n = norm(l)
        if len(n) >= 25:
            ref[n].add(f.split("/")[-1])
for f in sys.argv[1:]:
    tot = hit = 0; by = collections.Counter(); lines = []
    for i, l in enumerate(open(f, errors="replace"), 1):
        n = norm(l)
        if len(n) < 25:
            continue
        tot += 1
        if n in ref:
            hit += 1; by[sorted(ref[n])[0]] += 1; lines.append(i)
    print("%s: %d of %d substantive lines verbatim (%.1f %%) %s" % (f, hit, tot, 100.0 * hit / max(tot, 1), dict(by.most_common(6))))
    if "-v" in sys.argv:
        print(" lines:", lines)
