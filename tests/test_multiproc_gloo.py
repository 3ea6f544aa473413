"""CPU, world_size 2, gloo: the N>1 path of the batch driver — round-robin sharding of independent
proofs and the single gather of fixed-length proof byte strings (the only collective of the design)."""
import hashlib
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_proof(i, plen=96):
    return (hashlib.sha256(b"proof-%d" % i).digest() * 4)[:plen]


def worker(rank, world, port, num_proofs, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as ge

    pkg = ge.load_package()
    from anon_aadhaar_halo2_amd import batch

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    mine = batch.shard_indices(num_proofs, rank, world)
    local = [fake_proof(i) for i in mine]
    allp = batch.gather_proofs(local, num_proofs)
    q.put((rank, mine, [p.hex() for p in allp]))
    dist.destroy_process_group()
    del pkg


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run(world, num_proofs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, num_proofs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_ranks_shard_and_gather_even():
    res = run(2, 8)
    want = [fake_proof(i).hex() for i in range(8)]
    for rank, mine, allp in res:
        assert mine == list(range(rank, 8, 2))
        assert allp == want


def test_two_ranks_shard_and_gather_ragged():
    res = run(2, 5)  # rank 0 makes 3 proofs, rank 1 makes 2: padded slot must not leak into the result
    want = [fake_proof(i).hex() for i in range(5)]
    for rank, mine, allp in res:
        assert allp == want
