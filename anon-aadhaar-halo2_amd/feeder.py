"""Streaming witnesses: the device half of what a caller sees after `Circuit::synthesize`
(/root/reference/src/lib.rs:328-397) — SURVEY.md §8(f) rank 2.

The reference synthesizes one witness per proof on the host (Rust chips; not built here). What the backend owes
that caller is that handing over a fresh host witness per proof costs as little as possible: each in-flight
proving context owns two device buffers; while proof i runs out of one, the witness of proof i+1 is copied from
pinned host memory into the other on the context's copy stream (amdzk_dev_upload_async), and the create_proof
that reads it is ordered behind the copy on the device (amdzk_upload_fence) — no host wait, no pageable staging.
"""
import os
import queue
import threading
import time

import numpy as np


class PinnedWitness:
    """One witness (num_advice x n Fr, Montgomery form — what halo2's Assigned::evaluate leaves on the host) in
    hipHostMalloc'd memory."""

    def __init__(self, ctx, num_advice, n):
        self.shape = (num_advice, n, 4)
        self.nbytes = num_advice * n * 32
        self.buf = ctx.alloc_pinned(self.nbytes)
        self.array = self.buf.array(self.shape, np.uint64)

    @property
    def ptr(self):
        return self.buf.ptr.value

    def free(self):
        self.array = None
        self.buf.free()


class PacedUploader:
    """One thread per process that issues the witness uploads of ALL contexts in arrival order, a few MiB at a time and no
    faster than `rate` bytes/s in total. Uploads have a whole proof's time to arrive; sent at full speed they share the PCIe
    link with the command packets and kernel arguments the GPU fetches from host memory, and when several proofs run in step
    all their uploads fall on the same instant (profiles/r04y_streamed_regions_in_step.txt). Enabled by
    AMDZK_UPLOAD_RATE_GBS > 0; each job's `issued` event is set when its last copy has been handed to the runtime."""

    CHUNK = 4 << 20
    _inst = None
    _lock = threading.Lock()

    @classmethod
    def get(cls, rate):
        with cls._lock:
            if cls._inst is None:
                cls._inst = cls(rate)
            return cls._inst

    def __init__(self, rate):
        self.rate = float(rate)
        self.q = queue.Queue()
        self.t = threading.Thread(target=self._run, name="amdzk-uploader", daemon=True)
        self.t.start()

    def submit(self, ctx, dst, pinned):
        job = (ctx, dst, pinned.ptr, pinned.nbytes, threading.Event(), [])
        self.q.put(job)
        return job

    def _run(self):
        due = time.perf_counter()
        while True:
            ctx, dst, ptr, nbytes, issued, err = self.q.get()
            try:
                for off in range(0, nbytes, self.CHUNK):
                    n = min(self.CHUNK, nbytes - off)
                    now = time.perf_counter()
                    if due > now:
                        time.sleep(due - now)
                    else:
                        due = now
                    ctx.upload_async_at(dst, off, ptr + off, n)
                    due += n / self.rate
            except BaseException as e:  # noqa: BLE001 - handed to the thread that acquires the buffer
                err.append(e)
            issued.set()


class WitnessStream:
    """Double-buffered device staging for ONE proving context (one proof at a time per context)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        self.bufs = [ctx.alloc(nbytes), ctx.alloc(nbytes)]
        self.cur = 0
        self.pending = False
        self.pending_src = None
        self.chunk = int(float(os.environ.get("AMDZK_UPLOAD_CHUNK_MIB", "0")) * (1 << 20))
        rate = float(os.environ.get("AMDZK_UPLOAD_RATE_GBS", "0")) * 1e9
        self.uploader = PacedUploader.get(rate) if rate > 0 else None
        self.job = None

    def prefetch(self, pinned):
        """Start copying `pinned` (PinnedWitness) into the idle buffer. The previous reader of that buffer — the
        proof before the current one — has returned, so the buffer is free."""
        assert not self.pending, "one prefetch per acquire"
        assert pinned.nbytes <= self.nbytes
        dst = self.bufs[self.cur ^ 1]
        if self.uploader is not None:
            self.job = self.uploader.submit(self.ctx, dst, pinned)
        elif self.chunk and self.chunk < pinned.nbytes:
            for off in range(0, pinned.nbytes, self.chunk):
                self.ctx.upload_async_at(dst, off, pinned.ptr + off, min(self.chunk, pinned.nbytes - off))
        else:
            self.ctx.upload_async(dst, pinned.ptr, pinned.nbytes)
        self.pending = True
        self.pending_src = pinned

    def acquire(self):
        """The buffer holding the most recently prefetched witness; everything submitted to the ctx from now on
        runs after its copy."""
        assert self.pending, "acquire without prefetch"
        if self.job is not None:
            self.job[4].wait()
            err, self.job = self.job[5], None
            if err:
                raise err[0]
        self.ctx.upload_fence()
        self.cur ^= 1
        self.pending = False
        self.pending_src = None
        return self.bufs[self.cur]

    def free(self):
        for b in self.bufs:
            b.free()


def prove_stream(plonk, ctx, pk, stream, items, transcript=0, then=None):
    """create_proof for each (pinned_witness, instances, seed) of `items` on one context, with the next witness's
    upload overlapped with the current proof. Returns the proofs in order.

    `then`: the pinned witness the NEXT call on this stream starts with. Its upload is issued before this call's last
    proof, so a caller that proves in rounds (bench.py's timed regions, a service's batches) keeps the pipeline full
    across rounds: every call issues exactly len(items) uploads, none of them exposed. A stream that a previous call
    primed this way must be continued with that same witness."""
    out = []
    if not items:
        return out
    if stream.pending:
        assert stream.pending_src is items[0][0], "the stream was primed with another witness than this call starts with"
    else:
        stream.prefetch(items[0][0])
    for j, (_, inst, seed) in enumerate(items):
        buf = stream.acquire()
        nxt = items[j + 1][0] if j + 1 < len(items) else then
        if nxt is not None:
            stream.prefetch(nxt)
        out.append(plonk.create_proof(ctx, pk, inst, buf, seed=seed, transcript=transcript))
    return out
