#!/usr/bin/env python3
"""BASELINE.json configs[4]: synthetic k=22 KZG stress — one 2^22-point BN254 G1 MSM and one 2^22 Fr
NTT (+ inverse), inputs resident, with the HBM-roofline fractions (algorithmic bytes: 96*n for the
MSM, 64*n for the NTT). Usage: python tools/stress_k22.py [--k 22]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=22)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch

    import __graft_entry__ as ge

    pkg = ge.load_package()
    ctx = pkg.Context(0)
    k, n = args.k, 1 << args.k
    s_int = 7 ** 20 % R  # tau from "seed 7" 
    s_mont = np.array([((s_int << 256) % R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    t0 = time.perf_counter()
    params = pkg.kzg.ParamsKZG.setup(ctx, k, s_mont)
    ctx.sync()
    t_setup = time.perf_counter() - t0

    class V:
        def __init__(self, t):
            import ctypes
            self.ptr = ctypes.c_void_p(t.data_ptr())

    g = torch.Generator(device="cuda")
    g.manual_seed(42)
    uni = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    uni[:, 3] >>= 2
    sel = torch.randint(0, 4, (n,), device="cuda", generator=g)
    skew = uni.clone()
    skew[sel < 2] = 0               # 50% zero
    skew[sel == 2, 1:] = 0          # 25% < 2^64 (canonical)
    ctx._chk(ctx.L.amdzk_fr_from_raw_dev(ctx.h, skew.data_ptr(), n))
    ctx.sync()
    torch.cuda.synchronize()
    out = {"k": k, "srs_setup_s": round(t_setup, 2)}
    for name, col in (("msm_uniform", uni), ("msm_50zero_25small", skew)):
        pkg.arithmetic.best_multiexp_dev(ctx, params.h, 0, V(col), 1, n)
        ts = []
        for _ in range(args.reps):
            ctx.timer_start()
            pkg.arithmetic.best_multiexp_dev(ctx, params.h, 0, V(col), 1, n)
            ts.append(ctx.timer_stop())
        ms = float(np.median(ts))
        out[name] = {"ms": round(ms, 3), "algorithmic_GBps": round(96.0 * n / ms / 1e6, 2), "frac_of_8TBps": round(96.0 * n / ms / 1e6 / 8000, 5)}
    dom = pkg.domain.EvaluationDomain(ctx, 3, k)
    a = uni.clone()
    pkg.arithmetic.best_fft_dev(ctx, V(a), dom.omega, k)
    for name, w, flags in (("ntt_forward", dom.omega, 0), ("ntt_inverse_scaled", dom.omega_inv, 1)):
        ts = []
        for _ in range(args.reps):
            ctx.timer_start()
            pkg.arithmetic.best_fft_dev(ctx, V(a), w, k, flags=flags)
            ts.append(ctx.timer_stop())
        ms = float(np.median(ts))
        out[name] = {"ms": round(ms, 3), "algorithmic_GBps": round(64.0 * n / ms / 1e6, 2), "frac_of_8TBps": round(64.0 * n / ms / 1e6 / 8000, 5)}
    ctx.prof_reset()
    ctx.prof_enable(True)
    pkg.arithmetic.best_multiexp_dev(ctx, params.h, 0, V(uni), 1, n)
    pkg.arithmetic.best_fft_dev(ctx, V(a), dom.omega, k)
    ctx.prof_enable(False)
    out["per_kernel_ms"] = {kname: round(v[1], 3) for kname, v in sorted(ctx.prof_dump().items(), key=lambda kv: -kv[1][1])}
    print(json.dumps(out))
    params.free()
    dom.free()
    ctx.close()


if __name__ == "__main__":
    main()
