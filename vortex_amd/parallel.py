"""Multi-GPU layer: one process per GPU, images sharded across ranks, no data-path collective.

The reference has no distributed code on this path; images are independent
(per-image loop, base_postprocess.py:134-137), so every rank runs K1..NMS on its own contiguous
batch shard.  The only (optional) exchange is an all-gather of the variable-length detection
lists, done with torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU
tests): phase 1 gathers the per-image keep counts, phase 2 gathers the packed rows padded to the
largest per-rank total.  Payload is tiny (<= top_k*24 B per image), so it is latency-bound."""
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n items for `rank`; the first n % world ranks get one extra."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(t: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_range(t.size(0), rank, world)
    return t[lo:hi]


def _all_gather_flat(x: torch.Tensor, group=None) -> torch.Tensor:
    """x [n, ...] (same shape on every rank) -> [world*n, ...], rank-major."""
    world = dist.get_world_size(group)
    out = torch.empty((world * x.size(0),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if dist.get_backend(group) == "gloo":
        if x.is_cuda:
            # rehearsal on a box with fewer GPUs than ranks: gloo has no device all-gather, so the payload (a few KB per
            # image) is staged through host tensors; the production path below never leaves the device
            hx = x.contiguous().cpu()
            hout = torch.empty((world * hx.size(0),) + tuple(hx.shape[1:]), dtype=hx.dtype)
            dist.all_gather(list(hout.chunk(world, 0)), hx, group=group)
            out.copy_(hout)
        else:
            dist.all_gather(list(out.chunk(world, 0)), x.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


def allgather_detections(packed: torch.Tensor, meta_or_counts: torch.Tensor, B: int, group=None):
    """packed f32[>=sum(counts), D]: this rank's kept rows, image after image (vx_pack_rows layout);
    meta_or_counts: the i32 meta vector of the post-process (counts|keep_counts|status) or i32[B] keep counts.
    Returns, on every rank, the tuple of per-image f32[K'_i, D] over the GLOBAL batch (rank-major order).
    Every rank must hold the same number B of images (pad the last shard with empty images)."""
    if meta_or_counts.numel() == 2 * B + 1:
        counts = meta_or_counts[B:2 * B]
    else:
        counts = meta_or_counts
        assert counts.numel() == B
    if not (dist.is_available() and dist.is_initialized()):      # single process: nothing to gather
        kc = counts.cpu().tolist()
        return tuple(torch.split(packed[:sum(kc)], kc))
    world = dist.get_world_size(group)
    D = packed.size(1)
    all_counts = _all_gather_flat(counts.to(torch.int32), group).cpu()           # phase 1 (+ the D2H sync)
    per_rank = all_counts.view(world, B).sum(1).tolist()
    max_rows = max(max(per_rank), 1)
    mine = per_rank[dist.get_rank(group)]
    buf = torch.zeros((max_rows, D), dtype=packed.dtype, device=packed.device)
    buf[:mine] = packed[:mine]
    rows = _all_gather_flat(buf, group).view(world, max_rows, D)                  # phase 2
    out: List[torch.Tensor] = []
    cl = all_counts.view(world, B).tolist()
    for r in range(world):
        out.extend(torch.split(rows[r, :per_rank[r]], cl[r]))
    return tuple(out)
