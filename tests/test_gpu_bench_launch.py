"""GPU: `python bench.py --gpus 2` end to end with the REAL prover — launcher, two ranks, rendezvous, barriers,
max-over-ranks timing, the gather of the proofs and its check, one JSON line from rank 0. Both ranks share the box's
one GPU (AMDZK_BENCH_FORCE_DEVICE=0) and the collective backend is gloo: two NCCL ranks cannot sit on one device.
On the driver's 8-GPU node the same code runs with one GPU per rank over RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_batch_mode():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"AMDZK_BENCH_FORCE_DEVICE": "0", "AMDZK_BENCH_BACKEND": "gloo"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--shape", "k15", "--warmup", "1",
                        "--concurrency", "2", "--no-cpu-baseline", "--no-k22", "--regions", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    ln = lines[0]
    assert ln["n_gpus"] == 2 and ln["steps"] == 4 and ln["config"]["batch"] == 8 and ln["config"]["proofs_total"] == 8
    assert ln["data"] == "synthetic" and ln["config"]["proof_bytes"] == 34976
    assert "all_gather of 8 proofs" in ln["config"]["gather"]
    assert ln["config"]["pcie_inclusive_proofs_per_s"] > 0  # the streamed pass ran and its proofs equalled the resident ones
    assert ln["roofline"]["kernel"].startswith("msm") and ln["value"] > 1.0
    assert len(ln["config"]["value_samples"]) == 2 and ln["config"]["host_cpu_s_per_proof"] > 0 and ln["config"]["host_threads"] == 3


def test_one_rank_rccl_collectives():
    """AMDZK_BENCH_DIST_SELF=1: bench.py's N > 1 collectives — init_process_group("nccl", device_id=...), barrier, the
    max-over-ranks all_reduce on a device tensor, the all_gather of the proofs as device tensors — on a ONE-rank RCCL
    group: the part of the multi-GPU path that several gloo ranks on one device cannot reach."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "AMDZK_BENCH_BACKEND", "AMDZK_BENCH_FORCE_DEVICE"):
        env.pop(k, None)
    env["AMDZK_BENCH_DIST_SELF"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--shape", "k15", "--warmup", "1", "--concurrency", "4",
                        "--regions", "2", "--no-cpu-baseline", "--no-k22", "--no-serial-latency"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ln = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")][-1]
    assert ln["n_gpus"] == 1 and ln["config"]["proofs_total"] == 8
    assert "all_gather of 8 proofs" in ln["config"]["gather"]
