"""The all-gather of detections over RCCL (torch.distributed backend "nccl") on the one GPU of the test box:
a world of one rank exercises the same calls the N-GPU run makes (all_gather_into_tensor of the counts and of
the padded rows, device tensors in, device tensors out).  The world-size-2 logic is covered on CPU by
tests/test_parallel_gloo.py.  Runs in a child process so that a stuck rendezvous cannot hang the suite."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    import torch.distributed as dist
    import vortex_amd as V
    from vortex_amd import parallel
    from oracle import vortex_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        S, C, B = 64, 80, 4
        heads = [h.to(dev) for h in O.synth_yolo_heads(B, S, C, seed=7)]
        st, it = torch.tensor([0.06]), torch.tensor([0.2])
        post = V.YoloV3FusedPostProcess(S)
        ref = post(heads, st, it)
        packed, meta = post.forward_device(heads, st, it)
        got = parallel.allgather_detections(packed, meta, B)
        ok = len(got) == B and all(a.is_cuda and torch.equal(a, b) for a, b in zip(got, ref))
        shard = parallel.shard_batch(heads[0])
        ok = ok and shard.shape == heads[0].shape
        dist.barrier()
        q.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_allgather_detections_over_rccl_world1():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    try:
        assert q.get(timeout=150) is True
    finally:
        p.join(30)
        if p.is_alive():
            p.kill()


@pytest.mark.parametrize("allgather", [False, True])
def test_bench_spawns_its_own_ranks_without_a_launcher(allgather):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts two fresh rank processes itself (before
    any GPU call in the parent); on a one-GPU box they share the GPU over gloo and the line says so.  With --allgather
    every step ends in parallel.allgather_detections over the two ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--batch", "16", "--no-cpu-baseline", "--no-variants", "--configs", "none", "--repeats", "2"]
                         + (["--allgather"] if allgather else []),
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    import torch
    assert rec["n_gpus"] == 2 and len(rec["per_rank_images_per_s"]) == 2
    assert rec["backend"] == ("rccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert rec["config"]["global_batch"] == 32 and rec["value"] > 0
    assert rec["step_ms_min_max_over_ranks"][0] <= rec["step_ms_min_max_over_ranks"][1]
    assert len(rec["rank_devices"]) == 2 and rec["spread"]["repeats"] >= 2
    if allgather:              # the collective is rehearsed too (gloo stages the device tensors through the host)
        assert rec["allgather_ms_per_step"] > 0 and "all-gather" in rec["config"]["parallelism"]
