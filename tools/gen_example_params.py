#!/usr/bin/env python3
"""The parameter sets of the reference's shipped inputs that the hosts accept, as one JSON table for the device smoke test
(tests/test_gpu_examples.py): {"<example>/<input>": {name: value, ...}} with only the values that differ from the defaults.
Needs /root/reference and profiles/examples_coverage.md (tools/examples_coverage.py); inputs that need data files which are not shipped,
or a set-up of more than 90 s, are left out."""
import dataclasses
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lart_amd import host  # noqa: E402

out = {}
default = host.Params()
for line in open(os.path.join(ROOT, "profiles", "examples_coverage.md")):
    c = [x.strip() for x in line.split("|")]
    if len(c) < 5 or c[3] != "ok" or c[4]:
        continue
    path = os.path.join("/root/reference/examples", c[1], c[2])
    try:
        par = host.read_namelist(path)
    except Exception as e:                                      # noqa: BLE001
        print("skip", c[1], c[2], e); continue
    d = {}
    for f in dataclasses.fields(par):
        v, v0 = getattr(par, f.name), getattr(default, f.name)
        if isinstance(v, float) and isinstance(v0, float) and v != v and v0 != v0:
            continue
        if v != v0:
            d[f.name] = v if not isinstance(v, (list, tuple)) else list(v)
    for k in ("scatt_mat_file",):
        if k in d:
            d[k] = "MUELLER"                                    # the table of data/mueller_Lyalpha.dat: the test writes its own copy of the fixture
    if any(isinstance(v, str) and (v.endswith(".dat") or v.endswith(".gz") or v.endswith(".h5") or v.endswith(".txt")) for v in d.values()):
        continue
    out[c[1] + "/" + c[2]] = d
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "reference_example_params.json"), "w"), indent=0, sort_keys=True)
print(len(out), "parameter sets")
