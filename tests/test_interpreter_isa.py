"""CPU: the compiled expression interpreters, walked instruction by instruction (tools/isa_walk.py) — hipcc cross-compiles
plonk_kernels.hip for gfx950 here, the walk follows the wave's control flow through the assembly for a given program and
counts what is issued. Guards what a source-level test cannot see: the per-instruction overhead hipcc's structurizer adds
around the dispatch (~130 VALU instructions per interpreted instruction before the state was moved out of loop-carried
registers, ~45 after), and the shape of the hot paths (one in-place product = 171 multiplier instructions)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_walk  # noqa: E402

CSRC = os.path.join(ROOT, "anon-aadhaar-halo2_amd", "csrc")


@pytest.fixture(scope="module")
def walker(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-DAMDZK_ASM_PRODUCT", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", os.path.join(CSRC, "plonk_kernels.hip"), "-o", str(out / "pk.s")],
                   check=True, cwd=str(out), timeout=900)
    return isa_walk.Walker(open(out / "pk.s").read(), "expr_eval_limbs_kernel")


def cost(wk, pre, body, suf, reps=16):
    a, b = wk.run(pre + body * reps + suf), wk.run(pre + body * 2 * reps + suf)
    return {k: (b[k] - a[k]) / reps for k in ("valu", "mad64", "lds", "vmem")}


W = isa_walk.W_


def test_products_are_one_in_place_product_plus_a_bounded_overhead(walker):
    push, acc = [W("PUSH_COL", 1)], [W("WACC", 0), W("WFLUSH", 4 | 16)]
    for op in ("MUL_COL", "MUL_CONST"):
        c = cost(walker, push, [W(op, 2)], acc)
        assert c["mad64"] == 171, (op, c)          # 81 (a*b) + 81 (m*p) + 9 (m = t * p' mod 2^29)
        assert c["valu"] <= 171 + 95, (op, c)      # product's 35 other instructions + unpack + dispatch overhead
    c = cost(walker, push, [W("SQR")], acc)
    assert c["mad64"] == 135 and c["valu"] <= 135 + 100, c


def test_sums_cost_an_unpack_a_carry_pass_and_the_dispatch(walker):
    push, acc = [W("PUSH_COL", 1)], [W("WACC", 0), W("WFLUSH", 4 | 16)]
    for op, lim in (("ADD_COL", 80), ("ADD_CONST", 80), ("SUB_COL", 100), ("NEG", 85), ("REDUCE", 100)):
        c = cost(walker, push, [W(op, 2)], acc)
        assert c["mad64"] == 0 and c["valu"] <= lim, (op, c)


def test_a_term_costs_81_multiplier_instructions_to_accumulate(walker):
    c = cost(walker, [], [W("PUSH_COL", 1), W("WACC", 0)], [W("WFLUSH", 4 | 16)])
    assert c["mad64"] == 81 and c["valu"] <= 240 and c["lds"] <= 40, c
    carry = cost(walker, [], [W("PUSH_COL", 1), W("WACC", 1 << 23)], [W("WFLUSH", 4 | 16)])
    assert carry["mad64"] == 81 and carry["valu"] - c["valu"] <= 70, (c, carry)   # the carry pass rides along


def test_the_instruction_stream_is_scalar(walker):
    # no vector load of instruction words: per interpreted instruction only the operand fetch (<= 3 loads of 32 B)
    c = cost(walker, [W("PUSH_COL", 1)], [W("NEG")], [W("WACC", 0), W("WFLUSH", 4 | 16)])
    assert c["vmem"] <= 3, c
