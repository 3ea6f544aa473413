#!/usr/bin/env python3
"""bench.py — proofs/s of the create_proof hot path on MI355X (contract: see the task brief).

One "step" = the device work of ONE proof of the RSA-SHA256 circuit shape
(/root/reference/src/lib.rs:263-274,295-326: k=15, 80 gate advice + 16 range-lookup advice +
16 SHA spread advice, 24 lookups, 115 permutation columns -> 58 permutation products, degree 4 so
extended_k = 17), on synthetic witness columns already resident in HBM:
  * 248 MSMs of 2^15   (112 advice + 48 permuted lookup columns + 82 grand products + 1 random +
                        3 quotient pieces + 2 SHPLONK openings), batched per protocol phase,
  * 244 iNTTs of 2^15  (lagrange_to_coeff of every committed column),
  * 244 coset NTTs 2^15 -> 2^17 (coeff_to_extended) and 1 extended iNTT (extended_to_coeff).
This is the round-1 workload ("proof_shape_proxy"): the quotient evaluation, grand products,
evaluations and the Fiat-Shamir host driver are not yet inside the step, and `config.workload`
says so. Nothing in the timed region is cached: every step recomputes every MSM and NTT.

N > 1: independent proofs shard one-per-GPU (weak scaling); the only collective is the gather of
the finished commitments (RCCL all_gather of fixed-size byte strings).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = 15
N_ADVICE, N_LOOKUP_PERM, N_PRODUCTS, N_G_BASIS = 112, 48, 82, 6
N_POLYS = N_ADVICE + N_LOOKUP_PERM + N_PRODUCTS + 2  # + 2 instance columns
HBM_PEAK_GBS = 8000.0


def make_columns(torch, ctx, ncols, n, kind, seed):
    """Synthetic witness columns on the device, Montgomery form (BASELINE.md §3 distributions)."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    a = torch.randint(0, 2 ** 62, (ncols, n, 4), dtype=torch.int64, device="cuda", generator=g)
    a[..., 3] >>= 2  # < 2^60 < top limb of r: a valid canonical value
    if kind == "uniform":
        return a  # any value < r is a valid Montgomery representation of a uniform element
    sel = torch.randint(0, 10, (ncols, n), device="cuda", generator=g)
    small = sel < 7
    zero = (sel >= 7) & (sel < 9)
    if kind == "lookup":  # permuted lookup columns: table-sized values (12-bit range table)
        a[..., 0] &= 0xFFF
        a[..., 1:] = 0
    else:  # witness-like: 70% < 2^64, 20% zero, 10% uniform
        a[..., 1:][small] = 0
        a[zero] = 0
    ctx._chk(ctx.L.amdzk_fr_from_raw_dev(ctx.h, a.data_ptr(), a.numel() // 4))  # canonical -> Montgomery
    ctx.sync()
    return a


class DevView:
    """Lets pkg helpers address a torch tensor's storage."""

    def __init__(self, t):
        import ctypes
        self.ptr = ctypes.c_void_p(t.data_ptr())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge

    pkg = ge.load_package()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctx = pkg.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # one stream for torch copies and amdzk kernels
    n = 1 << K

    # ---- SRS: synthetic bases. Any set of curve points exercises the same arithmetic; take
    # pseudo-random multiples of the generator made by the device itself (MSM of unit vectors would
    # be circular), here: small-multiple ladder i*G via repeated addition on the host is too slow,
    # so use the oracle-free closed form: hash-to-x + square-root on the host with python ints.
    bases = synth_bases(2 * n, seed=7)
    params = pkg.kzg.ParamsKZG(ctx, K, g=bases[:n].copy(), g_lagrange=bases[n:].copy())
    dom = pkg.domain.EvaluationDomain(ctx, 4, K)
    en = dom.extended_len()

    adv = make_columns(torch, ctx, N_ADVICE, n, "witness", 1 + rank)
    lkp = make_columns(torch, ctx, N_LOOKUP_PERM, n, "lookup", 2 + rank)
    prod = make_columns(torch, ctx, N_PRODUCTS + N_G_BASIS + 2, n, "uniform", 3 + rank)
    coeff = torch.empty((N_POLYS, n, 4), dtype=torch.int64, device="cuda")
    ext = torch.empty((N_POLYS, en, 4), dtype=torch.int64, device="cuda")
    hq = torch.randint(0, 2 ** 60, (1, en, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()

    A = pkg.arithmetic
    commits = {}

    def step():
        # phase 1-3: commitments (Lagrange basis), one batched submission per protocol phase
        commits["advice"] = A.best_multiexp_dev(ctx, params.h, 1, DevView(adv), N_ADVICE, n)
        commits["lookup"] = A.best_multiexp_dev(ctx, params.h, 1, DevView(lkp), N_LOOKUP_PERM, n)
        commits["products"] = A.best_multiexp_dev(ctx, params.h, 1, DevView(prod), N_PRODUCTS, n)
        # lagrange -> coeff of every committed column (+ instance), then to the extended coset
        coeff[:N_ADVICE].copy_(adv)
        coeff[N_ADVICE:N_ADVICE + N_LOOKUP_PERM].copy_(lkp)
        coeff[N_ADVICE + N_LOOKUP_PERM:].copy_(prod[:N_PRODUCTS + 2])
        dom.lagrange_to_coeff_dev(DevView(coeff), ncols=N_POLYS)
        dom.coeff_to_extended_dev(DevView(coeff), DevView(ext), ncols=N_POLYS)
        # quotient: /Z_H, back to coefficients, commit random + 3 pieces + 2 openings (monomial basis)
        dom.divide_by_vanishing_poly_dev(DevView(hq))
        dom.extended_to_coeff_dev(DevView(hq))
        commits["g"] = A.best_multiexp_dev(ctx, params.h, 0, DevView(prod[N_PRODUCTS + 2:]), N_G_BASIS, n)
        ctx.sync()

    for _ in range(args.warmup):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the one exchange step: gather every rank's commitments (fixed-size byte strings) on all ranks
        blob = torch.from_numpy(np.concatenate([commits[k].reshape(-1) for k in sorted(commits)]).view(np.int64)).cuda()
        out = [torch.empty_like(blob) for _ in range(world)]
        dist.all_gather(out, blob)

    # ---- roofline of the dominant kernel, measured live with HIP events on the ctx stream
    roof = None
    cpu = None
    if rank == 0:
        ctx.prof_reset()
        ctx.prof_enable(True)
        step()
        ctx.prof_enable(False)
        prof = ctx.prof_dump()
        dom_name = max(prof, key=lambda kname: prof[kname][1])
        launches, total_ms = prof[dom_name]
        step_ms = sum(v[1] for v in prof.values())
        # algorithmic bytes of one MSM = 96*n (32 B scalar + 64 B base); this step's msm_accum_l1
        # launches cover (112, 48, 82, 6) columns -> mean columns per launch:
        cols_per_launch = (N_ADVICE + N_LOOKUP_PERM + N_PRODUCTS + N_G_BASIS) / 4.0
        if dom_name.startswith("msm"):
            alg_bytes = 96.0 * n * cols_per_launch
        else:
            alg_bytes = 64.0 * n * N_POLYS
        avg_s = total_ms / launches * 1e-3
        roof = {"bound": "hbm", "kernel": dom_name, "achieved": round(alg_bytes / avg_s / 1e9, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                "avg_launch_ms": round(total_ms / launches, 4), "launches_per_step": launches,
                "kernel_share_of_step": round(total_ms / step_ms, 3),
                "per_kernel_ms": {kname: round(v[1], 3) for kname, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}}
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(adv, lkp, prod, bases, n)

    if rank == 0:
        ms = dt / args.steps * 1e3
        line = {"metric": "create_proof proofs/sec (device work of one proof, RSA-SHA256 circuit shape k=15)",
                "value": round(world * args.steps / dt, 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u32x8 (254-bit Montgomery integers)", "data": "synthetic",
                "config": {"workload": "proof_shape_proxy: rsa_sha256 shape k=15 — 248 MSM(2^15) + 244 iNTT(2^15) + "
                                       "244 coset NTT(2^15->2^17) + 1 extended iNTT; quotient evaluation, grand products, "
                                       "evaluations and transcript NOT yet in the step",
                           "k": K, "extended_k": dom.extended_k, "proofs_per_gpu_per_step": 1, "parallelism": "proofs sharded 1/GPU"},
                "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    params.free()
    dom.free()
    ctx.close()


def synth_bases(count, seed):
    """Deterministic curve points: x from splitmix64, y = sqrt(x^3+3) (q = 3 mod 4), Montgomery form.
    Pure python integers; ~20 us per point."""
    q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    mont = (1 << 256) % q
    out = np.zeros((count, 8), dtype=np.uint64)
    s = seed & 0xFFFFFFFFFFFFFFFF
    i = 0

    def nxt():
        nonlocal s
        s = (s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    while i < count:
        x = (nxt() | (nxt() << 64) | (nxt() << 128) | ((nxt() >> 3) << 192)) % q
        rhs = (x * x * x + 3) % q
        y = pow(rhs, (q + 1) // 4, q)
        if y * y % q != rhs:
            continue
        xm, ym = x * mont % q, y * mont % q
        for l in range(4):
            out[i, l] = (xm >> (64 * l)) & 0xFFFFFFFFFFFFFFFF
            out[i, 4 + l] = (ym >> (64 * l)) & 0xFFFFFFFFFFFFFFFF
        i += 1
    return out


def cpu_baseline(adv, lkp, prod, bases, n):
    """CPU leg: the oracle (port of halo2's best_multiexp / best_fft, OpenMP over all host cores) on a
    bounded sample of the same step: 6 MSMs (2 per column kind), 4 iNTT 2^15, 4 coset NTT 2^17; scaled
    to the step's counts. The oracle is only the measured baseline here, never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import zkutil as zu

    O = zu.Oracle()
    cores = O.threads
    gl = bases[n:]

    def t_msm(t):
        col = np.ascontiguousarray(t.cpu().numpy().view(np.uint64))
        t0 = time.perf_counter()
        O.best_multiexp(col, gl)
        return time.perf_counter() - t0

    m_adv = (t_msm(adv[0]) + t_msm(adv[1])) / 2
    m_lkp = (t_msm(lkp[0]) + t_msm(lkp[1])) / 2
    m_uni = (t_msm(prod[0]) + t_msm(prod[1])) / 2
    od = zu.OracleDomain(O, 4, K)
    cols = [np.ascontiguousarray(prod[i].cpu().numpy().view(np.uint64)) for i in range(4)]
    t0 = time.perf_counter()
    for c in cols:
        od.lagrange_to_coeff(c)
    t_intt = (time.perf_counter() - t0) / 4
    t0 = time.perf_counter()
    for c in cols:
        od.coeff_to_extended(c)
    t_ext = (time.perf_counter() - t0) / 4
    per_proof = (N_ADVICE * m_adv + N_LOOKUP_PERM * m_lkp + (N_PRODUCTS + N_G_BASIS) * m_uni + N_POLYS * (t_intt + t_ext) + t_ext)
    return {"value": round(1.0 / per_proof, 5), "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "6 MSM(2^15) + 4 iNTT(2^15) + 4 coset NTT(2^17) timed with the C++ oracle (OpenMP, %d threads), "
                      "scaled to the step's 248/244/245 counts" % cores,
            "ms_per_msm": {"witness": round(m_adv * 1e3, 2), "lookup": round(m_lkp * 1e3, 2), "uniform": round(m_uni * 1e3, 2)},
            "ms_per_intt": round(t_intt * 1e3, 2), "ms_per_coset_ntt": round(t_ext * 1e3, 2)}


if __name__ == "__main__":
    main()
