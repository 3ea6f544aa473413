#!/usr/bin/env python3
"""bench.py — create_proof throughput on MI355X (contract: see the task brief).

One "step" = ONE full `create_proof` (KZG/SHPLONK/Blake2b, halo2_proofs v2023_01_20 semantics) of the composite
Aadhaar verifier circuit's budget (default --shape full: /root/reference/src/aadhaar_verifier_circuit.rs:49-56 = the
RSA-SHA256 shape of /root/reference/src/lib.rs:263-274,295-326 at k = 15 — 80 vertical-gate advice + 16 range-lookup
advice + 16 SHA spread advice columns, 24 lookups, 115 permutation columns -> 58 grand products, degree 4 so
extended_k = 17 — plus the IdentityCircuit / TimestampCircuit / SquareCircuit columns and gates: 141 advice, 118
permutation columns; --shape k15 / k18 = the RSA-SHA256 sub-circuit alone; the shapes live in
anon-aadhaar-halo2_amd/workloads.py), on synthetic satisfying witnesses. Each step draws fresh blinding
(seed = step index), takes the next of the rank's witnesses and recomputes everything: 278 commitments (multi-scalar
multiplications, in 8 batches), 274 inverse transforms, 274 x 3 coset transforms, the h(X) numerator on 3 cosets of 2^15
rows, 59 + 24 grand products, ~900 evaluations, SHPLONK. Witness synthesis (the reference's Rust chips) and keygen are
outside the step, as in upstream's own split. The SRS is a real one (g[i] = s^i G, g_lagrange[i] = L_i(s) G, built on
the device), so the proofs are valid; tests/test_gpu_prover.py verifies this circuit's proof with the oracle's verifier.

Launching: `python bench.py --gpus N` starts N ranks itself (one child process per GPU, before anything in the parent
touches HIP); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment). N > 1: independent proofs, one shard per GPU (weak
scaling); the only collective is the gather of the finished proof bytes (fixed length), an RCCL all_gather.
`--batch B` is BASELINE config 4: B independent witnesses (seeds 0..B-1), proof i on rank i mod N, gathered on all ranks.

The timed region runs `--regions` times (5 by default: same steps, fresh blinding seeds); `value` and `ms_per_step` are
those of the MEDIAN region and `config.value_samples` lists them all. `config.host_cpu_s_per_proof` is the process's CPU
time over that region divided by its proofs (all host threads: the per-proof drivers that block on the GPU), and
`--host-cores N` confines the rank to N cores before anything touches the GPU (one GPU's share of an 8-GPU host).

What is timed (round 4): the regions run with ONE HOST->DEVICE WITNESS UPLOAD PER PROOF — pinned host memory -> one of the
context's two device buffers on its copy stream, the next proof's upload under the current proof
(anon-aadhaar-halo2_amd/feeder.py), the upload for a region's first proofs issued under the previous region's last proofs —
so every region issues exactly `steps` uploads and the first P witnesses are in HBM when it starts: that is `value`, what
a caller of create_proof with a freshly synthesized witness gets. The same regions with the witnesses kept resident in HBM
are config.resident_proofs_per_s (the figure rounds 1-3 called `value`); region 0 of both passes must give the same bytes.
--no-stream-pass times the resident pass only (profiling).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
SHAPE_NAMES = ("full", "k15", "k18")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48,
                    help="timed proofs per GPU; the default gives four full rounds of 12 in flight, so pipeline fill/drain is a small share")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape", default="full", choices=SHAPE_NAMES,
                    help="full = composite Aadhaar verifier budget at k = 15 (the metric's configuration); k15 / k18 = RSA-SHA256 sub-circuit shapes")
    ap.add_argument("--batch", type=int, default=0,
                    help="BASELINE config 4: this many independent witnesses (seeds 0..B-1) over all ranks, proof i on rank i mod N; "
                         "overrides --steps with B / N")
    ap.add_argument("--witnesses", type=int, default=0, help="distinct resident witnesses per GPU the steps cycle through (0 = auto: 4, or all of them with --batch)")
    ap.add_argument("--concurrency", type=int, default=int(os.environ.get("AMDZK_BENCH_CONCURRENCY", "0")),
                    help="proofs in flight per GPU (each on its own amdzk context / HIP stream / proving-key workspace); "
                         "0 = auto: the largest divisor of --steps between 6 and 12 (so the timed steps form whole rounds), else "
                         "one of 5, 4, 3, else 8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream-pass", action="store_true", help="skip the PCIe-inclusive pass (one witness upload per proof)")
    ap.add_argument("--no-serial-latency", action="store_true",
                    help="skip the single-proof latency of a key made with AMDZK_KEYGEN_SERIAL (one stream per proof, as in rounds 1-2; "
                         "one more keygen and a few proofs; tools/profile_gpu.sh)")
    ap.add_argument("--stagger-ms", type=float, default=float(os.environ.get("AMDZK_BENCH_STAGGER_MS", "0")),
                    help="worker w starts its share of a timed region w x this many ms late (inside the timed region): the proofs in flight then sit "
                         "in different phases instead of in step")
    ap.add_argument("--regions", type=int, default=5, help="how many times the timed region (--steps proofs per GPU) runs; the median is reported")
    ap.add_argument("--host-cores", type=int, default=0,
                    help="confine this rank to the first N cores of its affinity mask (sched_setaffinity before any GPU call); 0 = leave it alone")
    ap.add_argument("--host-wait", default="auto", choices=("auto", "spin", "block"),
                    help="how the per-proof driver threads wait for the GPU during the timed steps: spin (hipStreamSynchronize), block "
                         "(poll a completion event with 50-us sleeps), auto = block with 4 or more proofs in flight or more than one rank: same rate on 16 cores for a sixth of "
                         "the CPU time, and 73 instead of 63 proofs/s on 2 cores (profiles/r03e_host_wait_and_cu_mask.txt)")
    ap.add_argument("--no-k22", action="store_true", help="skip BASELINE config 5 (2^22-point MSM and NTT, config.k22_stress) and the CPU kernel baselines")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, worker=None, env_extra=None, timeout=None):
    """Start n ranks of this script (one child per GPU) and wait for them. Runs BEFORE the parent has imported torch
    or touched HIP; the children are fresh interpreters (never an exec of a process that initialised the GPU). Rank 0's
    stdout is ours (the one JSON line); the other ranks' stdout goes to stderr. Returns the exit code."""
    port = free_port()
    cmd = list(worker) if worker else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    t_end = None if timeout is None else time.time() + timeout
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:  # one rank failed: the others would wait in a collective forever
                    q.terminate()
        if live:
            if t_end is not None and time.time() > t_end:
                rc = rc or 124
                for q in live:
                    q.kill()
            time.sleep(0.05)
    return rc


def cgroup_cpu_max():
    """(text, cores) of this process's CPU bandwidth limit: cgroup v2 cpu.max, else v1 cfs quota / period; cores None = no limit.
    (The pool's one-GPU boxes: affinity mask 256 cores, cpu.max "1600000 100000" = 16 cores' worth of time.)"""
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().strip()
        q, per = txt.split()[:2]
        return txt, (None if q == "max" else float(q) / float(per))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return "%d %d" % (q, per), (None if q <= 0 else q / per)
    except (OSError, ValueError):
        return None, None


def usable_cores(affinity_cores):
    quota = cgroup_cpu_max()[1]
    if affinity_cores is None:
        return None
    return affinity_cores if quota is None else max(1, min(affinity_cores, int(quota + 0.999)))


def rank_core_slice(allowed, local_rank, local_world):
    """The cores rank `local_rank` of `local_world` ranks on one host keeps: the r-th contiguous slice of the sorted ids
    the launcher itself may use. Slices are disjoint and cover `allowed`; with fewer cores than ranks the ranks share
    cores round-robin (a rank never ends up with an empty mask)."""
    allowed = sorted(allowed)
    n = len(allowed)
    if local_world <= 1 or n == 0:
        return allowed
    if n < local_world:
        return [allowed[local_rank % n]]
    lo, hi = local_rank * n // local_world, (local_rank + 1) * n // local_world
    return allowed[lo:hi]


# ------------------------------------------------------------------------------------------ provers
class DevView:
    def __init__(self, ptr):
        import ctypes
        self.ptr = ctypes.c_void_p(ptr)


def mont_limbs(v):
    import numpy as np
    return np.array([((v << 256) % R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


class GpuProver:
    """P proving contexts on one GPU over one circuit layout and `nw` resident witnesses."""

    def __init__(self, args, rank, local_rank, P, witness_seeds, want_host_srs):
        import numpy as np
        import torch
        import __graft_entry__ as ge

        self.np, self.torch = np, torch
        pkg = self.pkg = ge.load_package()
        check_library_stamp(pkg)
        self.plonk = pkg.plonk
        self.trace = None
        if os.environ.get("AMDZK_BENCH_TRACE"):
            # diagnosis only: (context index, start, end) of every create_proof, dumped as JSON at close()
            self.trace = [(-3, time.time(), time.perf_counter())]  # wall-clock anchor for the perf_counter stamps
            self.plonk = _TracedPlonk(pkg.plonk, self.trace)
        wl = pkg.workloads
        self.P = P
        self.ctxs = [pkg.Context(local_rank) for _ in range(P)]
        ctx = self.ctx = self.ctxs[0]
        t0 = time.perf_counter()
        c = self.circuit = wl.make(args.shape, seed=witness_seeds[0])
        self.desc = c.desc
        self.K = c.k
        self.n = 1 << c.k
        self.s_int = 0x0123456789ABCDEF0123456789ABCDEF % R
        self.tr_int = 0xA11CE
        self.params = pkg.kzg.ParamsKZG.setup(ctx, self.K, mont_limbs(self.s_int), want_host_copy=want_host_srs)
        fixed_host = self.to_mont_dev(c.fixed).cpu().numpy().view(np.uint64)
        tr = mont_limbs(self.tr_int)
        # Key modes (include/amdzk.h AMDZK_KEYGEN_*): with several proofs in flight per GPU every proof stays on ONE stream
        # (serial keys: the proofs already fill the chip between them; spreading each over three streams as well costs
        # 1-2 % of the rate, profiles/r03b_*); the single-proof latency is measured with a default (lanes) key.
        self.serial_keys = P >= 4 and os.environ.get("AMDZK_BENCH_LANES_KEYS") != "1"
        flags = self.plonk.KEYGEN_SERIAL if self.serial_keys else None  # None: the library's default (lanes, unless AMDZK_SERIAL=1)
        self.pks = [self.plonk.ProvingKey(cx, self.params, self.desc, fixed_host, c.assembly.mapping, tr, flags=flags) for cx in self.ctxs]
        self.fixed_host, self.tr = fixed_host, tr
        self._lat_keys = {}
        # resident witnesses (Montgomery form, device) + the same in pinned host memory for the PCIe-inclusive pass
        self.witness_ints = []
        self.adv, self.inst = [], []
        for j, ws in enumerate(witness_seeds):
            advice, instances = (c.advice, c.instances) if j == 0 else c.witness(ws)
            if j == 0:
                self.witness_ints.append((advice, instances))
            self.adv.append(self.to_mont_dev(advice))  # (A, n, 4) int64 view of u64 limbs
            self.inst.append([self.to_mont_dev([col]).cpu().numpy().view(np.uint64)[0] if col else np.zeros((0, 4), np.uint64)
                              for col in instances])
        self.d_adv = [DevView(t.data_ptr()) for t in self.adv]
        self.pinned, self.streams = None, None
        self.setup_s = time.perf_counter() - t0

    def to_mont_dev(self, cols):
        lim = self.pkg.workloads.canon_limbs(cols)
        t = self.torch.from_numpy(lim.view(self.np.int64)).cuda()
        self.ctx._chk(self.ctx.L.amdzk_fr_from_raw_dev(self.ctx.h, t.data_ptr(), t.numel() // 4))
        self.ctx.sync()
        return t

    def prove(self, w, wi, seed):
        return self.plonk.create_proof(self.ctxs[w], self.pks[w], self.inst[wi], self.d_adv[wi], seed=seed)

    def latency(self, seed, serial, reps=int(os.environ.get("AMDZK_BENCH_LATENCY_REPS", "3"))):
        """One proof alone on the GPU with a key of the given mode — default: the proof's independent work on three streams
        (lanes); serial: one stream, as in rounds 1-2 — `reps` times after one untimed proof: (median ms, the last proof)."""
        want_flags = self.plonk.KEYGEN_SERIAL if serial else None  # None: the library's default (lanes, unless AMDZK_SERIAL=1)
        if bool(serial) == bool(self.serial_keys):
            pk, own = self.pks[0], False
        else:
            pk, own = self.plonk.ProvingKey(self.ctx, self.params, self.desc, self.fixed_host, self.circuit.assembly.mapping, self.tr, flags=want_flags), True
        try:
            self.plonk.create_proof(self.ctx, pk, self.inst[0], self.d_adv[0], seed=seed + 1)  # workspaces in steady state
            ms = []
            for r in range(reps):
                self.ctx.sync()
                t = time.perf_counter()
                proof = self.plonk.create_proof(self.ctx, pk, self.inst[0], self.d_adv[0], seed=seed)
                ms.append((time.perf_counter() - t) * 1e3)
        finally:
            if own:
                pk.free()
        return sorted(ms)[len(ms) // 2], proof

    def sync(self):
        for cx in self.ctxs:
            cx.sync()
        self.torch.cuda.synchronize()

    def check_affinity(self):
        """Every context's stream / workspaces and every key's buffers must live on this rank's GPU."""
        for cx, pk in zip(self.ctxs, self.pks):
            cx.check_affinity(pk=pk.h)
        for t in self.adv:
            self.ctx.check_affinity(ptr=t.data_ptr())

    # ---- PCIe-inclusive pass
    def stream_setup(self):
        fd = self.pkg.feeder
        A = self.desc["num_advice"]
        self.pinned = []
        for t in self.adv:
            pw = fd.PinnedWitness(self.ctx, A, self.n)
            pw.array[...] = t.cpu().numpy().view(self.np.uint64)
            self.pinned.append(pw)
        self.streams = [fd.WitnessStream(cx, A * self.n * 32) for cx in self.ctxs]

    def prove_stream(self, w, jobs, then=None):
        """jobs: [(witness index, seed)] for worker w, proved with one upload per proof; `then` = the witness index the
        next call on this worker starts with (its upload is issued before this call's last proof: feeder.prove_stream)."""
        items = [(self.pinned[wi], self.inst[wi], seed) for wi, seed in jobs]
        return self.pkg.feeder.prove_stream(self.plonk, self.ctxs[w], self.pks[w], self.streams[w], items,
                                            then=None if then is None else self.pinned[then])

    def close(self):
        if self.trace is not None:
            idx = {id(cx): i for i, cx in enumerate(self.ctxs)}
            with open(os.environ["AMDZK_BENCH_TRACE"], "w") as f:
                json.dump([[idx.get(c, c if c < 0 else -9), round(a, 6), round(b, 6)] for c, a, b in self.trace], f)
        if self.streams:
            for s in self.streams:
                s.free()
        if self.pinned:
            for p in self.pinned:
                p.free()
        for q in self.pks:
            q.free()
        self.params.free()
        for cx in self.ctxs:
            cx.close()


class _TracedPlonk:
    """AMDZK_BENCH_TRACE: the plonk module with create_proof timed on the host clock (everything else passes through)."""

    def __init__(self, plonk, log):
        self._plonk, self._log = plonk, log

    def __getattr__(self, name):
        return getattr(self._plonk, name)

    def create_proof(self, ctx, *a, **kw):
        t0 = time.perf_counter()
        out = self._plonk.create_proof(ctx, *a, **kw)
        self._log.append((id(ctx), t0, time.perf_counter()))
        return out


def check_library_stamp(pkg):
    """libamdzk.so must have been built from THIS tree's kernel sources (amdzk_build_info() carries their hash): a stale
    binary that travelled with the snapshot would otherwise be measured under the new sources' name."""
    have, want = pkg.build_info().get("src"), kernel_src_hash()
    if have != want and os.environ.get("AMDZK_BENCH_ALLOW_STALE_LIB") != "1":
        raise SystemExit("bench.py: libamdzk.so was built from kernel sources %s, this tree's are %s: rebuild "
                         "(python -c 'import __graft_entry__ as g; g.build()')" % (have, want))
    return have


class StubProver:
    """TEST ONLY (AMDZK_BENCH_STUB=1, set by tests/test_bench_launcher.py, never by the driver): stands in for the
    GPU so that the launcher, the rank plumbing, the barriers, the gather and the JSON line can be exercised on CPU
    with gloo. Its "proofs" are hashes; the line it produces says data = "stub" and carries no roofline."""

    def __init__(self, args, rank, local_rank, P, witness_seeds, want_host_srs):
        self.P, self.K, self.n = P, 4, 16
        self.desc = {"num_advice": 0, "lookups": [], "permutation_columns": [], "cs_degree": 3, "num_instance": 0, "num_fixed": 0}
        self.witness_seeds = witness_seeds
        self.setup_s = 0.0
        self.streams = None

    def prove(self, w, wi, seed):
        time.sleep(0.002)
        return (hashlib.sha256(b"stub-%d-%d" % (self.witness_seeds[wi], seed)).digest() * 3)[:96]

    def sync(self):
        pass

    def check_affinity(self):
        pass

    def stream_setup(self):
        self.streams = True

    def prove_stream(self, w, jobs, then=None):
        return [self.prove(w, wi, seed) for wi, seed in jobs]

    def close(self):
        pass


def run_pool(P, jobs, fn):
    """Run fn(worker, job) over `jobs` with up to P workers (worker w owns context w; ctypes drops the GIL inside the C
    call, so the host drivers of different proofs overlap and their kernels interleave on the GPU). Results in job order."""
    nxt = iter(range(len(jobs)))
    lock = threading.Lock()
    results = [None] * len(jobs)
    errors = []

    def work(w):
        try:
            while True:
                with lock:
                    i = next(nxt, None)
                if i is None:
                    return
                results[i] = fn(w, jobs[i])
        except BaseException as e:  # noqa: BLE001 - re-raised on the main thread
            errors.append(e)

    th = [threading.Thread(target=work, args=(w,)) for w in range(min(P, max(1, len(jobs))))]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    if errors:
        raise errors[0]
    return results


# ------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    stub = os.environ.get("AMDZK_BENCH_STUB") == "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or under "
                         "torch.distributed.run with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    # The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default): with more
    # proofs in flight than queues, kernels of different proofs queue up behind each other. Must be set before the
    # runtime initialises (libamdzk.so's own initialiser does the same for hosts that load it first).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    cores_allowed = core_slice = None
    if hasattr(os, "sched_getaffinity"):
        # before torch / HIP: no exec, no taskset hop; threads started later inherit the mask.
        # N > 1 ranks on one host: rank r keeps the r-th slice of the cores the launcher was allowed (one host worker
        # set per GPU) — otherwise N x (P + 1) driver threads and N OpenMP pools share one mask.
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if lw > 1 and os.environ.get("AMDZK_BENCH_NO_CORE_SLICE") != "1":
            core_slice = rank_core_slice(sorted(os.sched_getaffinity(0)), int(os.environ.get("LOCAL_RANK", "0")), lw)
            os.sched_setaffinity(0, core_slice)
        if args.host_cores > 0:
            os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:args.host_cores])
        cores_allowed = len(os.sched_getaffinity(0))
    import torch
    import torch.distributed as dist

    # AMDZK_BENCH_FORCE_DEVICE / AMDZK_BENCH_BACKEND exist only to rehearse the N>1 code path on a one-GPU box (all
    # ranks on device 0, gloo instead of RCCL); the driver never sets them.
    backend = os.environ.get("AMDZK_BENCH_BACKEND", "gloo" if stub else "nccl")
    if not stub:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a gfx950 GPU (there is no CPU fallback in the product path)")
        if os.environ.get("AMDZK_BENCH_FORCE_DEVICE") is not None:
            local_rank = int(os.environ["AMDZK_BENCH_FORCE_DEVICE"])
        torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    # AMDZK_BENCH_DIST_SELF=1 (rehearsal on a one-GPU box, never set by the driver): a ONE-rank process group is made anyway
    # and every collective of the N > 1 path — barrier, the max-over-ranks all_reduce, the all_gather of the proofs — runs
    # on it, i.e. on RCCL with device tensors, which the gloo rehearsals of several ranks on one device cannot exercise.
    dist_on = world > 1 or os.environ.get("AMDZK_BENCH_DIST_SELF") == "1"
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if dist_on:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    steps = args.steps
    if args.batch:
        if args.batch % world:
            raise SystemExit("bench.py: --batch %d is not a multiple of the %d ranks" % (args.batch, world))
        steps = args.batch // world
    P = args.concurrency
    if P <= 0:
        # throughput grows slowly past 8 in flight once the runtime has as many hardware queues (profiles/r02j_hw_queues_sweep.txt:
        # 4 / 8 / 12 / 16 in flight = 69 / 74 / 74 / 75 proofs/s); a last round with idle workers costs more than that
        divs = [d for d in (12, 11, 10, 9, 8, 7, 6, 5, 4, 3) if steps % d == 0]
        P = divs[0] if divs else min(8, max(1, steps))
    # global proof index of (step s, rank r) = s*world + r (round-robin, batch.shard_indices). With --batch the witness
    # of proof g has seed g; otherwise the rank cycles through nw witnesses of its own.
    nw = args.witnesses or (steps if args.batch else min(4, steps))
    nw = max(1, min(nw, steps))
    if args.batch:
        witness_seeds = [s * world + rank for s in range(nw)]
    else:
        witness_seeds = [7 + 1000 * rank + j for j in range(nw)]
    # the CPU baseline and the k = 22 stress belong to the N = 1 line only (rank 0 of N > 1 has 1/N of the host's cores,
    # and the other ranks would sit in the closing barrier meanwhile)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and not stub
    if world > 1:
        args.no_cpu_baseline = args.no_k22 = True
    prover = (StubProver if stub else GpuProver)(args, rank, local_rank, P, witness_seeds, want_cpu)
    desc = prover.desc

    jobs = [(s % nw, 1000 * rank + s) for s in range(steps)]  # (witness index, blinding seed)

    def barrier():
        prover.sync()
        if dist_on:
            dist.barrier()

    def max_over_ranks(dt):
        if not dist_on:
            return dt
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # untimed: W proofs on EVERY in-flight context, so that each proving key's workspace, pinned staging and lazily
    # loaded kernels are in steady state before the timed region. (Per-context tasks: the JOB index is the context, so
    # whichever thread picks a task up, no two tasks share a context.)
    def warm_context(_, cx):
        for r in range(max(args.warmup, 0)):
            prover.prove(cx, (cx + r) % nw, 1000 * rank + 500000 + r * P + cx)

    # host waits: with more proofs in flight than cores to spare the driver threads sleep instead of spinning
    block_waits = args.host_wait == "block" or (args.host_wait == "auto" and (P >= 4 or world > 1))
    if not stub:
        for cx in prover.ctxs:
            cx.set_host_wait(block_waits)
    run_pool(P, list(range(P)), warm_context)
    prover.check_affinity()
    # The timed region, R times over: same steps, fresh blinding seeds; the median region is reported. Worker w (one
    # context, one host thread) proves steps w, w + P, w + 2P, ... of a region.
    #   streamed (the default, and `value`): every proof's witness crosses PCIe once — pinned host memory -> one of the
    #     context's two device buffers on its copy stream — while the context's previous proof runs; the upload for a
    #     region's first proof is issued before the previous region's last proof (a service's steady state), so each region
    #     issues exactly `steps` uploads and the first P witnesses are in HBM when a region starts.
    #   resident (`config.resident_proofs_per_s`, the figure rounds 1-3 called `value`): the witnesses stay in HBM.
    # Region 0 of both passes uses the same seeds: the proofs must be the same bytes.
    per_worker = [[j for i, j in enumerate(jobs) if i % P == w] for w in range(P)]
    R_ = max(1, args.regions)

    def region_items(reg):
        return [[(wi, seed + 100000 * reg) for wi, seed in per_worker[w]] for w in range(P)]

    def run_regions(streamed):
        regs, first = [], None
        if streamed:
            # untimed: both halves of every double buffer touched once, and the pipeline primed with region 0's first witness
            run_pool(P, list(range(P)), lambda _, cx: prover.prove_stream(
                cx, [(wi, seed + 900000) for wi, seed in per_worker[cx][:2]], then=per_worker[cx][0][0] if per_worker[cx] else None))
        for reg in range(R_):
            items = region_items(reg)
            barrier()
            c0 = time.process_time()
            t0 = time.perf_counter()
            if getattr(prover, "trace", None) is not None:
                prover.trace.append((-1 if streamed else -2, t0, t0))  # region start marker
            def lag(cx):
                if args.stagger_ms > 0:
                    time.sleep(cx * args.stagger_ms * 1e-3)

            if streamed:
                nxt = reg + 1 < R_
                pr = run_pool(P, list(range(P)), lambda _, cx: (lag(cx), prover.prove_stream(
                    cx, items[cx], then=per_worker[cx][0][0] if (nxt and per_worker[cx]) else None))[1])
            else:
                pr = run_pool(P, list(range(P)), lambda _, cx: (lag(cx), [prover.prove(cx, wi, seed) for wi, seed in items[cx]])[1])
            barrier()
            regs.append((max_over_ranks(time.perf_counter() - t0), time.process_time() - c0))
            if reg == 0:
                first = [None] * steps
                for w in range(P):
                    for q, proof in enumerate(pr[w]):
                        first[w + q * P] = proof
        return regs, first

    stream_equal = None
    streamed_value = not args.no_stream_pass
    if streamed_value:
        prover.stream_setup()
        regions, proofs = run_regions(True)
        res_regions, res_proofs = run_regions(False)
        stream_equal = res_proofs == proofs
        if not stream_equal:
            raise SystemExit("bench.py: proofs from streamed witnesses differ from the resident-witness proofs")
    else:
        regions, proofs = run_regions(False)
        res_regions = regions
    dt, host_cpu_s = sorted(regions)[len(regions) // 2]
    res_dt = sorted(res_regions)[len(res_regions) // 2][0]
    gathered_ok = None
    if dist_on:
        # the one exchange step: every rank's proofs (equal length) gathered on all ranks (RCCL all_gather)
        if stub:
            import __graft_entry__ as ge
            batch_mod = ge.load_package().batch
        else:
            batch_mod = prover.pkg.batch
        gathered = batch_mod.gather_proofs(proofs, world * steps, device=coll_dev, force=True)
        gathered_ok = len(gathered) == world * steps and all(len(p) == len(proofs[0]) for p in gathered) and \
            all(gathered[s * world + rank] == proofs[s] for s in range(steps))
        if not gathered_ok:
            raise SystemExit("bench.py: gathered proofs do not match this rank's proofs")

    # host cores of every rank (count, first id, last id) for the line
    my_mask = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    rank_cores = [[len(my_mask), my_mask[0] if my_mask else -1, my_mask[-1] if my_mask else -1]]
    if dist_on and world > 1:
        mine = torch.tensor(rank_cores[0], device=coll_dev, dtype=torch.int64)
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        rank_cores = [[int(v) for v in t.cpu().tolist()] for t in allm]

    roof = cpu = None
    wall_prof = lat_ms = lat_serial_ms = k22 = None
    if rank == 0 and not stub:
        roof, wall_prof = roofline(prover, desc)
        # single-proof latency: one proof alone on the GPU, no per-kernel events (median of 3); then the same with a
        # serial-mode key (one stream), whose proof must be the same bytes
        prover.sync()
        prover.ctx.set_host_wait(False)  # one proof alone: spinning waits are the low-latency choice
        lat_ms, ref = prover.latency(777001, serial=False)
        if not args.no_serial_latency:
            lat_serial_ms, other = prover.latency(777001, serial=True)
            if other != ref:
                raise SystemExit("bench.py: the proof of the serial-mode key differs from the default (lanes) key's")
        if not args.no_k22:
            k22 = k22_stress(prover, want_cpu=not args.no_cpu_baseline)
        if not args.no_cpu_baseline:
            gpu_proof = prover.prove(0, 0, 424242)
            cpu = cpu_baseline(prover, gpu_proof)
            if k22 and k22.get("cpu_kernels"):
                cpu["kernels"] = k22.pop("cpu_kernels")

    if rank == 0:
        ms = dt / steps * 1e3
        metric = "create_proof wall-clock (ms) + proofs/sec, full Aadhaar circuit, 1/2/4/8 GPU" if args.shape == "full" \
            else "create_proof wall-clock (ms) + proofs/sec, RSA-SHA256 circuit shape"
        line = {"metric": ("STUB " if stub else "") + metric,
                "value": round(world * steps / dt, 4), "unit": "proofs/s", "n_gpus": world, "steps": steps,
                "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None,
                "dtype": "u32 limbs (254-bit Montgomery integers: 9 x 29-bit in the hot products, 8 x 32-bit elsewhere)",
                "data": "stub" if stub else "synthetic",
                "config": {"workload": "create_proof, %s: %d advice, %d lookups, %d permutation columns, degree %d, "
                                       "KZG/SHPLONK/Blake2b, %d distinct witnesses per GPU cycled through"
                                       % (args.shape, desc["num_advice"], len(desc["lookups"]), len(desc["permutation_columns"]),
                                          desc["cs_degree"], nw),
                           "k": prover.K, "extended_k": prover.K + max(0, (desc["cs_degree"] - 2).bit_length()),
                           "proof_bytes": len(proofs[-1]), "proofs_in_flight_per_gpu": P,
                           "batch": args.batch or None, "proofs_total": world * steps,
                           "warmup_proofs_untimed": max(args.warmup, 0) * P,
                           "value_samples": [round(world * steps / r[0], 4) for r in regions],
                           "value_is": ("streamed: one %.0f MiB witness upload per proof (pinned host memory -> device on the context's copy "
                                        "stream, double-buffered, the next proof's upload under the current proof; the first %d witnesses "
                                        "are in HBM when a region starts)" % (desc["num_advice"] * prover.n * 32 / 2 ** 20, P)) if streamed_value
                           else "resident witnesses (--no-stream-pass)",
                           "resident_proofs_per_s": round(world * steps / res_dt, 4),
                           "resident_samples": [round(world * steps / r[0], 4) for r in res_regions],
                           "streamed_over_resident": round(res_dt / dt, 4),
                           "streamed_equals_resident_bytes": stream_equal,
                           "timed_regions": len(regions),
                           "host_cpu_s_per_proof": round(host_cpu_s / steps, 5), "host_threads": min(P, steps) + 1,
                           "host_cores_allowed": cores_allowed, "host_cgroup_cpu_max": cgroup_cpu_max()[0],
                           "host_cores_usable": usable_cores(cores_allowed),
                           "host_cores_per_rank": [{"rank": r, "cores": c[0], "first": c[1], "last": c[2]} for r, c in enumerate(rank_cores)],
                           "host_cores_rule": ("rank r keeps the r-th contiguous slice of the launcher's allowed cores (sched_setaffinity before "
                                               "any GPU call)") if core_slice is not None else "the launcher's mask, unchanged",
                           "host_wait": "block (driver threads poll a completion event with 50-us sleeps)" if block_waits else "spin (hipStreamSynchronize)",
                           "host_note": "rank 0's process CPU time (all threads) over the median region / its proofs; one host thread "
                                        "per proof in flight, blocked in the HIP runtime while the GPU works",
                           "single_proof_latency_ms": round(lat_ms, 3) if lat_ms else None,
                           "single_proof_latency_ms_serial_key": round(lat_serial_ms, 3) if lat_serial_ms else None,
                           "latency_note": "one proof in flight, median of 3; default key = the proof's independent work on three "
                                           "streams (lanes), serial key (AMDZK_KEYGEN_SERIAL) = one stream; same proof bytes",
                           "k22_stress": k22,
                           "pcie_inclusive_proofs_per_s": round(world * steps / dt, 4) if streamed_value else None,
                           "gather": ("all_gather of %d proofs, every rank's own proofs found in place" % (world * steps)) if gathered_ok else None,
                           "parallelism": "independent proofs sharded across GPUs, %d in flight per GPU" % P,
                           "key_mode": ("serial keys for the timed steps (one stream per proof; %d proofs in flight fill the chip), "
                                        "default lanes key for single_proof_latency_ms" % P) if getattr(prover, "serial_keys", False)
                           else "default (lanes) keys",
                           "library_build": None if stub else prover.pkg.build_info()["text"],
                           "setup_s_excluded": round(prover.setup_s, 1)},
                "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    prover.close()
    if cpu is not None and not cpu["proof_bytes_equal_gpu"]:
        raise SystemExit("bench.py: the CPU oracle's proof differs from the GPU's at the benchmarked shape")


def roofline(prover, desc):
    """Per-kernel timing of one more proof with HIP events on the ctx stream (the stream the kernels run on)."""
    ctx = prover.ctx
    n = prover.n
    ctx.prof_reset()
    ctx.prof_enable(True)
    t1 = time.perf_counter()
    prover.prove(0, 0, 10 ** 6)
    wall_prof = (time.perf_counter() - t1) * 1e3
    ctx.prof_enable(False)
    prof = ctx.prof_dump()
    dom_name = max(prof, key=lambda kname: prof[kname][1])
    launches, total_ms = prof[dom_name]
    gpu_ms = sum(v[1] for v in prof.values())
    A, L, S = desc["num_advice"], len(desc["lookups"]), len(desc["permutation_columns"])
    nsets = (S + desc["cs_degree"] - 3) // (desc["cs_degree"] - 2)
    msm_cols = A + 2 * L + nsets + L + 1 + (desc["cs_degree"] - 1) + 2
    npolys = A + desc["num_instance"] + 3 * L + nsets
    if dom_name.startswith("msm"):
        # algorithmic bytes of an MSM = 96 B per (scalar, base) pair; this kernel's launches cover all
        # msm_cols committed columns of the proof
        alg_bytes = 96.0 * n * msm_cols / launches
    elif dom_name.startswith("ntt"):
        alg_bytes = 64.0 * (n * npolys + (n << 2) * npolys + (n << 2)) / launches
    else:  # h(X) evaluation: every coset column read once + h written
        alg_bytes = 32.0 * (n << 2) * (npolys + desc["num_fixed"] + S + 4 + 1) / launches
    avg_s = total_ms / launches * 1e-3
    pmc = pmc_counters(dom_name)
    roof = {"bound": "hbm", "kernel": dom_name, "achieved": round(alg_bytes / avg_s / 1e9, 3), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS, 6), "traffic": pmc.get("traffic"),
            "traffic_source": pmc.get("source"), "algorithmic_bytes_per_launch": round(alg_bytes),
            "avg_launch_ms": round(total_ms / launches, 4), "avg_launch_ms_profile": pmc.get("avg_ms_profile"),
            "launches_per_step": launches,
            # what the kernel is actually limited by: VALU issue. cycles per VALU wave-instruction =
            # launch time x 1024 SIMDs x 2.4 GHz / SQ_INSTS_VALU per launch (of a proof's launches); the product block alone
            # reaches 4.3 with three wavefronts per SIMD, the kernel 5.3 (profiles/r03w_wide_product_and_l1_prefetch.txt)
            "valu_wave_insts_per_launch": pmc.get("valu"), "valu_wave_insts_source": pmc.get("valu_source", "whole-run average" if pmc.get("valu") else None),
            "valu_issue_cycles_per_inst": round(avg_s * 1024 * 2.4e9 / pmc["valu"], 2) if pmc.get("valu") else None,
            "issue_bound": issue_bound(pmc, avg_s),
            "gpu_busy_ms_per_step": round(gpu_ms, 3), "wall_ms_profiled_step": round(wall_prof, 3),
            "per_kernel_ms": {kname: round(v[1], 3) for kname, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}}
    return roof, wall_prof


MAD_CYCLES = 4.3  # v_mad_u64_u32, cycles per wave-instruction and SIMD at its peak (8 independent chains, 4-8 waves per SIMD:
                  # profiles/r02k_instruction_rates_mb_isa.jsonl; 5.3-5.8 with 4 chains, whatever register takes the carry-out)


def issue_bound(pmc, avg_s):
    """The limit the dominant kernel is actually at: its multiply-adds cannot issue faster than MAD_CYCLES each on
    1024 SIMDs at 2.4 GHz. mads per launch = SQ_INSTS_VALU per launch (PMC) x the multiply-add share of the kernel's
    inner loop (static, from the assembly; stamped into the same summary by tools/summarize_prof.py). frac = that
    lower bound / the measured launch time: the rest is the other VALU instructions (carry extraction, sums,
    unpacking) and the dependent-issue stalls of the one-accumulator product."""
    if not pmc.get("valu") or pmc.get("mad_share") is None:
        return None
    share = pmc["mad_share"]
    model_s = pmc["valu"] * share * MAD_CYCLES / (1024 * 2.4e9)
    return {"mad_share_of_valu": share, "peak_cycles_per_mad": MAD_CYCLES, "mads_per_launch": round(pmc["valu"] * share),
            "mad_issue_ms_per_launch": round(model_s * 1e3, 4), "frac": round(model_s / avg_s, 4)}


def strip_comments(text):
    """C/C++ source without comments and with runs of blanks collapsed (string and character literals kept as they are),
    so that editing a comment does not change kernel_src_hash()."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            while i < n and text[i] != "\n":
                i += 1
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    lines = [" ".join(ln.split()) for ln in "".join(out).split("\n")]
    return "\n".join(ln for ln in lines if ln)


def kernel_src_hash():
    """sha256 over the kernel sources (csrc/*.hip, *.cuh, *.hpp, *.inc, comments and blank space stripped, and the Makefile
    with its flags), the same way tools/summarize_prof.py stamps a PMC summary: counters measured on other kernel code
    are not this build's counters, and a comment edit is not other kernel code."""
    import glob
    h = hashlib.sha256()
    d = os.path.join(ROOT, "anon-aadhaar-halo2_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.cuh")) + glob.glob(os.path.join(d, "*.hpp")) +
                    glob.glob(os.path.join(d, "*.inc")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(p).encode() + b"\0")
        text = open(p, "r", errors="replace").read()
        h.update((text if p.endswith("Makefile") else strip_comments(text)).encode())
    return h.hexdigest()[:16]


def pmc_counters(kernel):
    """HBM bytes and VALU wave-instructions per launch of the dominant kernel from the committed rocprofv3 PMC passes
    of this same command (counters cannot be read from inside the process): the newest profiles/*_kernel_summary.csv
    whose first line records THIS build's kernel-source hash. A summary taken on other kernel code is refused and the
    fields stay null. FETCH_SIZE + WRITE_SIZE in KiB; the gfx950 x2 FETCH correction is for wide coalesced streams and
    is NOT applied to this kernel's 64-byte random gathers (uncalibrated pattern, stated as such)."""
    import csv
    import glob
    names = {"msm_accum_l1": "msm_accum_seg_kernel<true>", "expr_evaluate_h": "expr_eval_kernel<true>",
             "ntt_step": "ntt_step_kernel<false>", "ntt_step_last": "ntt_step_kernel<true>"}
    want = kernel_src_hash()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_summary.csv")), reverse=True):
        try:
            with open(path) as f:
                first = f.readline()
                if not first.startswith("# kernel_src_sha256="):
                    continue
                if first.strip().split("=", 1)[1] != want:
                    stale = stale or os.path.basename(path)
                    continue
                mad_share = None
                body = []
                for line in f:  # further '#' lines are notes; only the mad share is read
                    if line.startswith("# msm_accum_l1_mad_share="):
                        mad_share = float(line.split("=", 1)[1].split()[0])
                    elif not line.startswith("#"):
                        body.append(line)
                if not body or "SQ_INSTS_VALU_per_launch" not in body[0]:
                    continue  # another command's summary (tools/summarize_k22.py writes *_k22_kernel_summary.csv)
                for row in csv.DictReader(body):
                    if row.get("kernel") == names.get(kernel, kernel):
                        out = {"source": "profiles/%s (rocprofv3 --pmc, raw, per launch; kernel sources %s)" % (os.path.basename(path), want)}
                        if row.get("FETCH_SIZE_KiB_per_launch_raw") and row.get("WRITE_SIZE_KiB_per_launch_raw"):
                            out["traffic"] = round((float(row["FETCH_SIZE_KiB_per_launch_raw"]) + float(row["WRITE_SIZE_KiB_per_launch_raw"])) * 1024)
                        if row.get("SQ_INSTS_VALU_per_launch"):
                            out["valu"] = round(float(row["SQ_INSTS_VALU_per_launch"]))
                        if mad_share is not None and kernel == "msm_accum_l1":
                            out["mad_share"] = mad_share
                        out["avg_ms_profile"] = steady_avg_ms(path, names.get(kernel, kernel))
                        # instructions per launch of a PROOF's launches (the summary's whole-run average also holds keygen's
                        # larger commitment batches: 3.70e8 against 3.36e8 for the level-1 kernel)
                        sv = steady_avg_ms(path, names.get(kernel, kernel), "SQ_INSTS_VALU_per_launch")
                        if sv:
                            out["valu"] = round(sv)
                            out["valu_source"] = "steady-state proof"
                        return out
        except OSError:
            continue
    return {"source": "none for kernel sources %s%s" % (want, " (newest stamped summary, %s, is of other sources: refused)" % stale if stale else "")}


def k22_stress(prover, want_cpu):
    """BASELINE config 5: one 2^22-point BN254 G1 multi-scalar multiplication (uniform scalars; then 50 % zero / 25 %
    below 2^64) and one 2^22 Fr transform forward + inverse on resident inputs, with their fractions of the 8 TB/s HBM
    peak on algorithmic bytes (96 n and 64 n). With the CPU leg: the same kernels through the oracle
    (liboracle.so: Pippenger as halo2's best_multiexp, radix-2 best_fft; OpenMP on the box's host threads) at 2^15,
    2^18 and 2^22 — and the 2^22 results of both sides must be equal."""
    import ctypes
    np, torch, pkg, ctx = prover.np, prover.torch, prover.pkg, prover.ctx
    k, n = 22, 1 << 22
    s_int = 7 ** 20 % R  # tau from "seed 7"
    t0 = time.perf_counter()
    params = pkg.kzg.ParamsKZG.setup(ctx, k, mont_limbs(s_int), want_host_copy=want_cpu)
    ctx.sync()
    out = {"k": k, "srs_setup_s": round(time.perf_counter() - t0, 2)}

    class V:
        def __init__(self, t):
            self.ptr = ctypes.c_void_p(t.data_ptr())

    g = torch.Generator(device="cuda")
    g.manual_seed(42)
    uni = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    uni[:, 3] >>= 2  # any 252-bit pattern is a valid Montgomery representative
    sel = torch.randint(0, 4, (n,), device="cuda", generator=g)
    skew = uni.clone()
    skew[sel < 2] = 0
    skew[sel == 2, 1:] = 0
    ctx._chk(ctx.L.amdzk_fr_from_raw_dev(ctx.h, skew.data_ptr(), n))
    ctx.sync()
    torch.cuda.synchronize()

    def med(fn, reps):
        fn()
        ts = []
        for _ in range(reps):
            ctx.timer_start()
            fn()
            ts.append(ctx.timer_stop())
        return float(np.median(ts))

    gpu_msm = None
    for name, col in (("msm", uni), ("msm_skewed", skew)):
        res = []
        ms = med(lambda: res.append(pkg.arithmetic.best_multiexp_dev(ctx, params.h, 0, V(col), 1, n)), 3)
        if name == "msm":
            gpu_msm = res[-1][0]
        out[name + "_ms"] = round(ms, 3)
        out[name + "_frac_of_8TBps"] = round(96.0 * n / ms / 1e6 / HBM_PEAK_GBS, 5)
    dom = pkg.domain.EvaluationDomain(ctx, 3, k)
    a = uni.clone()
    for name, w, flags in (("ntt", dom.omega, 0), ("intt", dom.omega_inv, 1)):
        ms = med(lambda: pkg.arithmetic.best_fft_dev(ctx, V(a), w, k, flags=flags), 5)
        out[name + "_ms"] = round(ms, 3)
        out[name + "_frac_of_8TBps"] = round(64.0 * n / ms / 1e6 / HBM_PEAK_GBS, 5)
    out["note"] = "resident inputs, HIP-event timed, medians; fractions on algorithmic bytes (96 n per MSM, 64 n per transform)"
    if want_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import plonk_fast as PF

        L, th = PF.lib(), PF.threads()
        vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)
        scal = np.ascontiguousarray(uni.cpu().numpy().view(np.uint64))
        bases = params._g
        kern = {"cores": th, "host_cores": PF.host_cores(),
                "note": "liboracle.so (C++/OpenMP restatement of best_multiexp / best_fft), one run each, on the first 2^k "
                        "scalars and SRS points of the 2^22 stress inputs; *_16_threads = the same on a 16-thread pool (rounds 1-3's figure)"}
        for kk in (15, 18, 22):
            m = 1 << kk
            res = np.zeros(8, np.uint64)
            t0 = time.perf_counter()
            L.oracle_best_multiexp(vp(scal), vp(bases), ctypes.c_size_t(m), ctypes.c_int(th), vp(res))
            kern["msm_2^%d_ms" % kk] = round((time.perf_counter() - t0) * 1e3, 2)
            if kk == k:
                jac = gpu_msm.reshape(12)
                kern["msm_2^22_equal_gpu"] = bool(jac[8:].any()) and bool((jac[:8] == res).all())
            w = np.zeros(4, np.uint64)
            L.oracle_fr_omega(ctypes.c_uint32(kk), vp(w))
            v = np.ascontiguousarray(scal[:m].copy())
            t0 = time.perf_counter()
            L.oracle_best_fft(vp(v), vp(w), ctypes.c_uint32(kk), ctypes.c_int(th))
            kern["ntt_2^%d_ms" % kk] = round((time.perf_counter() - t0) * 1e3, 2)
            if th > 16:
                res16 = np.zeros(8, np.uint64)
                t0 = time.perf_counter()
                L.oracle_best_multiexp(vp(scal), vp(bases), ctypes.c_size_t(m), ctypes.c_int(16), vp(res16))
                kern["msm_2^%d_ms_16_threads" % kk] = round((time.perf_counter() - t0) * 1e3, 2)
                v16 = np.ascontiguousarray(scal[:m].copy())
                t0 = time.perf_counter()
                L.oracle_best_fft(vp(v16), vp(w), ctypes.c_uint32(kk), ctypes.c_int(16))
                kern["ntt_2^%d_ms_16_threads" % kk] = round((time.perf_counter() - t0) * 1e3, 2)
            if kk == k:
                b = uni.clone()
                pkg.arithmetic.best_fft_dev(ctx, V(b), w, k)
                ctx.sync()
                kern["ntt_2^22_equal_gpu"] = bool((b.cpu().numpy().view(np.uint64) == v).all())
        out["cpu_kernels"] = kern
        if not (kern["msm_2^22_equal_gpu"] and kern["ntt_2^22_equal_gpu"]):
            raise SystemExit("bench.py: the 2^22 stress results of the GPU and of the CPU oracle differ: %r" % kern)
    dom.free()
    params.free()
    return out


def steady_avg_ms(summary_path, kernel, column="avg_ms"):
    """Average launch time (or, column = "SQ_INSTS_VALU_per_launch", the VALU wave-instructions per launch) of `kernel` over
    ONE steady-state proof of the rocprofv3 passes that belong to a PMC summary (profiles/<tag>_kernel_stats_steady.csv,
    written by tools/summarize_prof.py next to it)."""
    import csv
    path = summary_path.replace("_kernel_summary.csv", "_kernel_stats_steady.csv")
    try:
        with open(path) as f:
            rows = [ln for ln in f if not ln.startswith("#")]
        for row in csv.DictReader(rows):
            if row["kernel"] == kernel and row.get(column):
                return round(float(row[column]), 4)
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(prover, gpu_proof):
    """CPU leg: full create_proof runs of the same circuit, witness, SRS and RNG seed on the host cores with the oracle
    prover (oracle/plonk_fast.py: upstream's step order; every O(n) loop — Pippenger MSM per commitment as halo2's
    best_multiexp, radix-2 FFTs, the h(X) evaluation, permutation / lookup products, evaluations — in the C++ oracle under
    OpenMP; transcript, RNG and glue in Python, which inflates the CPU time somewhat). Its proof must equal the GPU's byte
    for byte (bench.py exits non-zero otherwise); keygen is excluded on both sides. The oracle is the measured baseline
    here, never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import plonk_fast as PF

    c = prover.circuit
    advice, instances = prover.witness_ints[0]
    t0 = time.perf_counter()
    fpk = PF.FastKey(c.desc, c.fixed, c.assembly.mapping, prover.s_int, prover.tr_int, msm_bases=(prover.params._g, prover.params._gl))
    t_keygen = time.perf_counter() - t0
    hc = PF.host_cores()
    times, equal = [], True
    for _ in range(2):  # two samples on every usable core: the first also warms the OpenMP pool
        t0 = time.perf_counter()
        proof = PF.create_proof(fpk, instances, advice, seed=424242)
        times.append(time.perf_counter() - t0)
        equal = equal and proof == gpu_proof
    dt = min(times)
    th = PF.threads()
    out = {"value": round(1.0 / dt, 5), "unit": "proofs/s", "cores": th, "kind": "port",
           "sample": "2 full create_proof runs (same circuit/witness/SRS/seed as the GPU run) with the C++/OpenMP oracle prover on "
                     "%d threads = every core this process may use (affinity mask %d cores, cgroup cpu.max %r), the faster one reported; "
                     "Python transcript/RNG/glue included; keygen (%.0f s) excluded"
                     % (th, hc["affinity_cores"], hc["cgroup_cpu_max"], t_keygen),
           "seconds_per_proof": round(dt, 2), "seconds_per_proof_samples": [round(t, 2) for t in times],
           "host_cores": hc}
    if th > 16 and not os.environ.get("ORACLE_THREADS"):
        # the figure of rounds 1-3 (a 16-thread pool) beside it: one more sample
        PF.set_threads(16)
        try:
            t0 = time.perf_counter()
            proof = PF.create_proof(fpk, instances, advice, seed=424242)
            out["seconds_per_proof_16_threads"] = round(time.perf_counter() - t0, 2)
            equal = equal and proof == gpu_proof
        finally:
            PF.set_threads(None)
    out["proof_bytes_equal_gpu"] = equal
    return out


def main(argv=None):
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # parent of N ranks: nothing here has imported torch or loaded libamdzk
        sys.exit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv)))
    run_rank(args)


if __name__ == "__main__":
    main()
