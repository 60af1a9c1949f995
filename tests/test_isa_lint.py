"""Build check on the generated gfx950 code (no GPU needed: hipcc cross-compiles): the instantiations of the five BASELINE configurations, the overlap kernel and
the peel march are compiled to assembly one by one and run through tools/isa_lint.py, which looks for register-allocator copies / spills placed in front of the
exec restore of a join block -- the miscompile that corrupted transport_kernel<2,true,false,false,1> at 128 VGPRs in round 4 and faulted
transport_kernel<2,true,true,...> in round 3 (profiles/r04_exec_restore_miscompile.md).  A hit means: this build of this kernel must not ship."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
KERNELS = ["transport_kernel<1, false, false, false, 1>",      # C2 (bench.py)
           "transport_kernel<2, true, false, false, 1>",       # C1
           "transport_kernel<0, true, false, true, 1>",        # C3
           "transport_kernel<4, true, true, true, 1>",         # C4
           "transport_kernel<3, false, false, false, 1>",      # C5
           "transport_kernel<3, false, false, true, 1>",       # C5 with an observer (bench.py leg)
           "transport_kernel<7, false, false, false, 1>",      # overlapping clumps
           "peel_kernel<0, true>", "peel_deposit_kernel<0, true>"]
FLAGS = "--offload-arch=gfx950 -O3 -mllvm -disable-machine-licm -mllvm -simplifycfg-sink-common=false -std=c++17 -fPIC -munsafe-fp-atomics --cuda-device-only -S".split()


def compile_one(args):
    i, k, tmp = args
    src = os.path.join(tmp, "k%d.hip" % i); out = os.path.join(tmp, "k%d.s" % i)
    with open(src, "w") as f:
        f.write('#include "%s"\ntemplate __global__ void %s(KArgs);\n' % (os.path.join(ROOT, "lart_amd", "csrc", "peel_kernel.hpp"), k))
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_allocator_code_in_front_of_an_exec_restore(tmp_path):
    import isa_lint
    with ThreadPoolExecutor(max_workers=4) as ex:
        outs = list(ex.map(compile_one, [(i, k, str(tmp_path)) for i, k in enumerate(KERNELS)]))
    hits = [h for o in outs for h in isa_lint.lint(o)]
    assert not hits, hits


def test_the_lint_sees_the_pattern(tmp_path):
    """the three lines of the round-4 find (and a harmless block: SGPR spill reloads ignore exec) through the matcher"""
    import isa_lint
    bad = tmp_path / "bad.s"; ok = tmp_path / "ok.s"
    bad.write_text("_Z1kv:\n.LBB0_352:\n\tv_mov_b32_e32 v36, v68\n\tscratch_store_dword off, v69, off offset:40 ; 4-byte Folded Spill\n\ts_or_b64 exec, exec, s[6:7]\n\tv_mbcnt_lo_u32_b32 v1, -1, 0\n")
    ok.write_text("_Z1kv:\n.LBB0_301:\n\tv_readlane_b32 s52, v127, 19\n\tv_readlane_b32 s53, v127, 20\n\ts_or_b64 exec, exec, s[52:53]\n\tv_mov_b32_e32 v36, v68\n")
    assert len(isa_lint.lint(str(bad))) == 1 and isa_lint.lint(str(bad))[0][2] == ".LBB0_352"
    assert isa_lint.lint(str(ok)) == []


@pytest.mark.skipif(not os.environ.get("LART_RUN_SLOW"), reason="compiles every translation unit to assembly (minutes of eight cores): LART_RUN_SLOW=1, or tools/isa_lint.py --build by hand")
def test_every_translation_unit_is_lint_clean():
    """tools/isa_lint.py --build: all instantiations the library ships, not only the BASELINE ones (round 4: a hit in transport_kernel<4,false,false,true,7>
    at 128 VGPRs -> that instantiation is built for 3 waves per SIMD, lart_kernel_waves in csrc/lart_device.hpp)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), "--build"], capture_output=True, text=True)
    assert r.returncode == 0 and "0 hit(s)" in r.stdout, r.stdout[-3000:]
