"""N>1 path on CPU: world_size-2 gloo run of the shard + all-gather logic (the same code that runs
over RCCL on the GPUs).  Each rank's local detection lists come from the oracle on its shard; the
gathered tuple must equal the oracle's result on the unsharded batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vortex_amd import parallel
from oracle import vortex_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, C, B = 64, 80, 6
        pred = O.yolo_head_eval(O.synth_yolo_heads(B, S, C, seed=7), S)
        st, it = torch.tensor([0.06]), torch.tensor([0.2])
        full = O.yolo_postprocess(pred, st, it, S)
        shard = parallel.shard_batch(pred, rank, world)
        lo, hi = parallel.shard_range(B, rank, world)
        assert shard.shape[0] == hi - lo == B // world
        local = O.yolo_postprocess(shard, st, it, S)
        counts = torch.tensor([r.shape[0] for r in local], dtype=torch.int32)
        packed = torch.cat(local + (torch.zeros(5, 6),), 0)      # extra rows beyond sum(counts) are ignored
        meta = torch.cat((torch.zeros(len(local), dtype=torch.int32), counts, torch.zeros(1, dtype=torch.int32)))
        got = parallel.allgather_detections(packed, meta, len(local))
        ok = len(got) == B and all(torch.equal(a, b) for a, b in zip(got, full))
        got2 = parallel.allgather_detections(packed, counts, len(local))
        ok = ok and all(torch.equal(a, b) for a, b in zip(got2, full))
        # an empty rank still participates
        if rank == 1:
            packed, counts = torch.zeros(0, 6), torch.zeros(len(local), dtype=torch.int32)
        got3 = parallel.allgather_detections(packed, counts, len(local))
        ok = ok and all(torch.equal(a, b) for a, b in zip(got3[:len(local)], full[:len(local)]))
        ok = ok and all(t.shape == (0, 6) for t in got3[len(local):])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_allgather_detections_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 256, 257):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
