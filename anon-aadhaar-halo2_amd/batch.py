"""Proof-level data parallelism: independent proofs of one circuit are sharded over
the ranks (one process per GPU), proof i -> rank i mod world; the only exchange is the gather of the
finished proof byte strings, which all have the same length for one circuit. Backend-agnostic:
`nccl` (= RCCL over xGMI) on GPUs, `gloo` in the CPU tests."""
import torch
import torch.distributed as dist


def shard_indices(num_proofs, rank, world):
    """Indices of the proofs this rank produces (round-robin, as §8(e): proof i -> GPU i mod N)."""
    return list(range(rank, num_proofs, world))


def gather_proofs(local_proofs, num_proofs, device="cpu", force=False):
    """all_gather the ranks' proofs; returns the list of all proofs in index order on every rank.
    local_proofs: this rank's proofs in the order of shard_indices(). force: run the collectives even in a one-rank
    group (bench.py's AMDZK_BENCH_DIST_SELF rehearsal)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        assert len(local_proofs) == num_proofs
        return list(local_proofs)
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = shard_indices(num_proofs, rank, world)
    assert len(local_proofs) == len(mine)
    plen = len(local_proofs[0]) if local_proofs else 0
    # every rank contributes the same number of slots (pad the short ones) so one all_gather suffices
    slots = (num_proofs + world - 1) // world
    meta = torch.tensor([plen], dtype=torch.int64, device=device)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    plen = int(meta.item())
    buf = bytearray(slots * plen)
    for j, p in enumerate(local_proofs):
        assert len(p) == plen, "proofs of one circuit must have equal length"
        buf[j * plen:(j + 1) * plen] = p
    send = torch.frombuffer(buf, dtype=torch.uint8).to(device)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    out = [None] * num_proofs
    for r in range(world):
        data = bytes(recv[r].cpu().numpy().tobytes())
        for j, idx in enumerate(shard_indices(num_proofs, r, world)):
            out[idx] = data[j * plen:(j + 1) * plen]
    return out
