"""Batch-sharded multi-GPU execution: one process per GPU, torch.distributed over
RCCL/xGMI ("nccl" backend on ROCm; "gloo" in the CPU tests).

The path shards by independent samples — every (prompt, latent) pair is its own 50-step
trajectory — so there is NO data-path collective.  The only communication is
  * one broadcast of the weights from rank 0 at load (23.8 GB bf16 for FLUX.1-dev), tensor by
    tensor so each message is a large contiguous buffer (xGMI is point-to-point: few, big messages);
  * one gather of the decoded u8 images (3 MB/sample at 1024^2) to rank 0 per batch.
The reference has no distributed code at all (pipelines/mod.rs:214-217 uses device 0 only).
"""
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n: int, rank: int, world_size: int) -> List[int]:
    """Sample i runs on rank i % world_size (SURVEY §8e partitioning); returns this rank's sample ids."""
    return list(range(rank, n, world_size))


def broadcast_tensors(shapes: Dict[str, tuple], make: Callable[[str, tuple], torch.Tensor], sink: Callable[[str, torch.Tensor], None],
                      device, dtype=torch.bfloat16, src: int = 0) -> int:
    """Rank `src` materialises each tensor with `make(name, shape)`; everyone receives it and hands it
    to `sink(name, tensor)` (e.g. FluxModel.set_tensor).  Returns the number of bytes broadcast."""
    rank, ws = world()
    total = 0
    for name, shape in shapes.items():
        if rank == src:
            t = make(name, shape).to(device=device, dtype=dtype).contiguous()
        else:
            t = torch.empty(shape, dtype=dtype, device=device)
        if ws > 1:
            dist.broadcast(t, src=src)
        sink(name, t)
        total += t.numel() * t.element_size()
        del t
    return total


def gather_to_rank0(local: torch.Tensor, n_total: int) -> Optional[torch.Tensor]:
    """Gather per-rank sample stacks (n_local, ...) to rank 0 and restore the global sample order
    (sample i was produced by rank i % world).  Ranks may hold different counts (ragged batch)."""
    rank, ws = world()
    if ws == 1:
        return local
    counts = [len(shard_indices(n_total, r, ws)) for r in range(ws)]
    cmax = max(counts)
    pad = local
    if local.shape[0] < cmax:  # pad so every rank sends the same shape
        fill = torch.zeros((cmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad = torch.cat([local, fill], 0)
    bufs = [torch.empty_like(pad) for _ in range(ws)] if rank == 0 else None
    dist.gather(pad.contiguous(), bufs, dst=0)
    if rank != 0:
        return None
    out = [None] * n_total
    for r in range(ws):
        for j, i in enumerate(shard_indices(n_total, r, ws)):
            out[i] = bufs[r][j]
    return torch.stack(out, 0)


def generate_sharded(prompts: Sequence[str], run_local: Callable[[List[str], List[int]], torch.Tensor]) -> Optional[torch.Tensor]:
    """Shard `prompts` across ranks, run `run_local(my_prompts, my_sample_ids)` -> (n_local, ...) on each,
    gather to rank 0 in prompt order."""
    rank, ws = world()
    ids = shard_indices(len(prompts), rank, ws)
    local = run_local([prompts[i] for i in ids], ids)
    return gather_to_rank0(local, len(prompts))
