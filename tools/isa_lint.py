#!/usr/bin/env python3
"""Lint of the generated gfx950 assembly for ONE miscompile pattern of the register allocator (found in round 4, profiles/r04_exec_restore_miscompile.md):

    .LBB0_352:                                   ; join block of `if (cond) { ... }` -- entered with exec = the lanes that took the branch
        v_mov_b32_e32 v36, v68                   ; <- live-range-split COPY inserted by the allocator
        scratch_store_dword off, v69, off offset:40 ; 4-byte Folded Spill
        s_or_b64 exec, exec, s[6:7]              ; <- only HERE do the lanes that skipped the branch come back

The copy / spill runs for the lanes of the branch only although the value is live in every lane; the lanes that skipped the branch later get a stale
register or scratch slot back (transport_kernel<2,true,false,false,1> at 128 VGPRs: the per-lane event counter, i.e. garbage n_events; the round-3 fault of
transport_kernel<2,true,true,...> at 128 VGPRs is the same thing hitting an index).  It needs live-range splitting across a divergent region, i.e. a kernel
squeezed below what it wants (dozens of VGPR spills).

    tools/isa_lint.py FILE.s [...]        exit code 1 and one line per hit when a labelled block has allocator copies / spills / reloads in front of its exec restore
    tools/isa_lint.py --build             compile every translation unit of lart_amd/csrc to assembly (device only, the flags of the Makefile) and lint it
"""
import os, re, subprocess, sys, tempfile
RA = re.compile(r"^\s*(v_mov_b32|v_mov_b64|v_accvgpr_(read|write)|scratch_(store|load)_\w+|buffer_(store|load)_\w+ .*(Spill|Reload))")
LANE_IO = re.compile(r"^\s*(v_writelane_b32|v_readlane_b32|s_mov_b\d+|s_nop|s_waitcnt)")        # SGPR spill traffic and scalar moves ignore exec: harmless in front of the restore (they usually reload the saved mask itself)
EXEC_RESTORE = re.compile(r"^\s*s_or_b64 exec, exec, s\[")
LABEL = re.compile(r"^\.LBB\d+_\d+:")
FUNC = re.compile(r"^(_Z\w+):")


def lint(path):
    hits = []; func = "?"; lines = open(path, errors="replace").read().splitlines()
    i = 0
    while i < len(lines):
        m = FUNC.match(lines[i])
        if m:
            func = m.group(1)
        if LABEL.match(lines[i]):
            j = i + 1; pre = []
            while j < len(lines) and j < i + 24:
                t = lines[j]
                if not t.strip() or t.lstrip().startswith(";"):
                    j += 1; continue
                if EXEC_RESTORE.match(t):
                    if pre:
                        hits.append((func, i + 1, lines[i].split(":")[0], [p.strip().split(";")[0].strip() for p in pre]))
                    break
                if RA.match(t):
                    pre.append(t); j += 1; continue
                if LANE_IO.match(t):
                    j += 1; continue
                break
        i += 1
    return hits


def main():
    files = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--build" in sys.argv:
        root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lart_amd", "csrc")
        tmp = tempfile.mkdtemp()
        flags = "--offload-arch=gfx950 -O3 -mllvm -disable-machine-licm -mllvm -simplifycfg-sink-common=false -std=c++17 -fPIC -munsafe-fp-atomics --cuda-device-only -S".split()
        flags += [a[8:] for a in sys.argv if a.startswith("--extra=")]
        srcs = sorted(f for f in os.listdir(root) if f.endswith(".hip"))
        procs = []
        for f in srcs:
            out = os.path.join(tmp, f[:-4] + ".s"); files.append(out)
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + flags + [os.path.join(root, f), "-o", out], stderr=subprocess.DEVNULL))
            if len(procs) >= 8:
                procs.pop(0).wait()
        for p in procs:
            p.wait()
    bad = 0
    for f in files:
        for func, ln, label, pre in lint(f):
            bad += 1
            name = subprocess.run(["c++filt", func], capture_output=True, text=True).stdout.strip()
            print("%s:%d %s in %s: %s  BEFORE the exec restore" % (os.path.basename(f), ln, label, name, " ; ".join(pre)))
    print("isa_lint: %d hit(s) in %d file(s)" % (bad, len(files)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
