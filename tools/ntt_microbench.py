#!/usr/bin/env python3
"""NTT-only workload for counter collection: `reps` batched transforms of `ncols` resident columns of 2^k (both steps of
the 2-step plan at k = 15), timed with HIP events. Usage: python tools/ntt_microbench.py [--k 15] [--ncols 822] [--reps 5]"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--ncols", type=int, default=822)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch

    import __graft_entry__ as ge

    pkg = ge.load_package()
    ctx = pkg.Context(0)
    n = 1 << a.k
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    x = torch.randint(0, 2 ** 62, (a.ncols, n, 4), dtype=torch.int64, device="cuda", generator=g)
    x[:, :, 3] >>= 2
    torch.cuda.synchronize()
    dom = pkg.domain.EvaluationDomain(ctx, 3, a.k)

    class V:
        ptr = ctypes.c_void_p(x.data_ptr())

    pkg.arithmetic.best_fft_dev(ctx, V, dom.omega, a.k, ncols=a.ncols)
    ts = []
    for _ in range(a.reps):
        ctx.timer_start()
        pkg.arithmetic.best_fft_dev(ctx, V, dom.omega, a.k, ncols=a.ncols)
        ts.append(ctx.timer_stop())
    ts.sort()
    print(json.dumps({"k": a.k, "ncols": a.ncols, "ms_median": round(ts[len(ts) // 2], 4), "ns_per_element": round(ts[len(ts) // 2] * 1e6 / (a.ncols * n), 4)}))
    dom.free()
    ctx.close()


if __name__ == "__main__":
    main()
