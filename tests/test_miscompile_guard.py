"""CPU: keep the round-1 wrong-result build noticed.

`tools/repro_min_stage1.hip`, spelling A (eight field elements into a per-thread array, multiplied up under a
`#pragma unroll` that hipcc only partially honours) returns wrong products on the MI355X at -O3 and right ones at -O2
(profiles/r02e_min_stage1_miscompile_O3_only.txt). Without a GPU the result cannot be re-checked here; what CAN be held
is (1) the shape that triggers it — the array lives in scratch memory (dynamic index), the product loop is unrolled
3 + 1, and -O2 / -O3 schedule the scratch stores around the loads differently — so that a hipcc change that removes or
moves the pattern fails this test and sends someone back to the GPU reproducer; and (2) that NO kernel the library
ships has that shape: the kernels that use scratch memory at all are the ones DESIGN.md §6 lists, none of them on the
per-proof path."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "anon-aadhaar-halo2_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


def kernel_meta(asm):
    """kernel name -> (private segment bytes, vgpr spills) from the .amdgpu_metadata block."""
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", asm, re.S):
        blk = m.group(2)
        out[m.group(1)] = (int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)),
                           int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)))
    return out


def body(asm, name_part):
    m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end\d+:" % name_part, asm, re.S | re.M)
    assert m, name_part
    return [ln.split(";")[0].strip() for ln in m.group(2).splitlines() if ln.split(";")[0].strip() and not ln.strip().startswith(".")]


@pytest.fixture(scope="module")
def repro_asm(tmp_path_factory):
    d = tmp_path_factory.mktemp("repro")
    out = {}
    for opt in ("O2", "O3"):
        p = str(d / ("r_%s.s" % opt))
        subprocess.run([HIPCC, "-" + opt, "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                        os.path.join(ROOT, "tools", "repro_min_stage1.hip"), "-o", p], check=True, cwd=str(d), timeout=900,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out[opt] = open(p).read()
    return out


def test_reproducer_still_has_the_triggering_shape(repro_asm):
    for opt in ("O2", "O3"):
        meta = {k: v for k, v in kernel_meta(repro_asm[opt]).items() if "stage1" in k}
        a = [v for k, v in meta.items() if "stage1ILi0E" in k][0]
        b = [v for k, v in meta.items() if "stage1ILi1E" in k][0]
        c = [v for k, v in meta.items() if "stage1ILi2E" in k][0]
        assert a[0] > 0 and b[0] > 0, "spellings A and B keep m[] in scratch memory (dynamically indexed array)"
        assert c == (0, 0), "spelling C has no array, hence no scratch"
        ins = body(repro_asm[opt], "stage1ILi0E")
        mads = sum(1 for x in ins if x.startswith("v_mad_u64_u32"))
        rolled = sum(1 for x in body(repro_asm[opt], "stage1ILi1E") if x.startswith("v_mad_u64_u32"))
        assert rolled > 0 and 3.5 < mads / rolled < 4.5, "spelling A: the product loop is unrolled 3 + 1 (4 inlined products), B: rolled"
        assert any(x.startswith("scratch_store") for x in ins) and any(x.startswith("scratch_load") for x in ins)
    # the two pipelines differ in how the loads of stage one and the scratch stores are interleaved — the only difference
    # between the right (-O2) and the wrong (-O3) build; if they become identical the bug has moved or gone: re-run
    # tools/repro_min_stage1.hip on a GPU and update DESIGN.md §6
    assert body(repro_asm["O2"], "stage1ILi0E") != body(repro_asm["O3"], "stage1ILi0E")


# Kernels of the shipped library that use scratch memory (hipcc -O3, gfx950), all start-up work (SRS table build, the
# G1 FFT of g_to_lagrange / downsize) or a 12-byte register spill — DESIGN.md §6 names them. Per-proof kernels: none.
SCRATCH_ALLOWED = {"table_next_kernel": "start-up: window-table build", "msm_fold_kernel": "16-byte spill, no array",
                   "ecfft_round_kernel": "start-up: G1 FFT", "ecfft_finish_kernel": "start-up: G1 FFT", "ecfft_load_kernel": "start-up: G1 FFT"}


def test_no_shipped_per_proof_kernel_uses_scratch(tmp_path):
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-DAMDZK_ASM_PRODUCT", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
             "--cuda-device-only", "-S"]
    procs = []
    for f in ("capi", "ntt", "msm", "poly", "plonk_kernels", "prover"):
        out = str(tmp_path / (f + ".s"))
        procs.append((out, subprocess.Popen([HIPCC] + flags + [os.path.join(CSRC, f + ".hip"), "-o", out], cwd=str(tmp_path),
                                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    users = {}
    for out, p in procs:
        assert p.wait(timeout=1200) == 0, out
        for name, (priv, spill) in kernel_meta(open(out).read()).items():
            if priv:
                users[name] = (priv, spill)
    for name in users:
        assert any(ok in name for ok in SCRATCH_ALLOWED), "kernel %s now uses %d bytes of scratch memory: a per-thread array next to inlined " \
            "products is the shape that was miscompiled (DESIGN.md §6) — restructure it or extend the list with a reason" % (name, users[name][0])
