"""Batch-sharded multi-GPU execution: one process per GPU, torch.distributed over
RCCL/xGMI ("nccl" backend on ROCm; "gloo" in the CPU tests).

The path shards by independent samples — every (prompt, latent) pair is its own 50-step
trajectory — so there is NO data-path collective.  The only communication is
  * one broadcast of the weights from rank 0 at load: the model's flat weight arenas
    (23.8 GB bf16 / 6.7 GB nf4 for FLUX.1-dev) go out in a few messages of up to 1 GiB each
    (`broadcast_state`; xGMI is point-to-point: few, big messages), after a ~1 KB description of the layout;
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


STATE_CHUNK = 1 << 30  # bytes per broadcast message of the weight arenas


class StateExportUnsupported(RuntimeError):
    """The source rank's checkpoint cannot travel as flat arenas (LLM.int8 matrices): every rank loads it itself."""


def agree_or_raise(error: Optional[BaseException], what: str = "load", group=None) -> None:
    """Collective error check over `group` (default: all ranks): every rank reports whether its `what` step failed; if any did,
    ALL of them raise (instead of the healthy ones blocking in the next collective until the RCCL timeout).  No-op without
    torch.distributed."""
    if not (dist.is_available() and dist.is_initialized()):
        if error is not None:
            raise error
        return
    ws = dist.get_world_size(group)
    if ws == 1:
        if error is not None:
            raise error
        return
    msgs = [None] * ws
    dist.all_gather_object(msgs, None if error is None else f"{type(error).__name__}: {error}", group=group)
    bad = [(r, m) for r, m in enumerate(msgs) if m is not None]
    if bad:
        if error is not None:
            raise error
        raise RuntimeError(f"{what} failed on rank {bad[0][0]} of the group: {bad[0][1]}")


def broadcast_state(model, device, src: int = 0, chunk_bytes: int = STATE_CHUNK, comm: "Optional[RcclComm]" = None) -> dict:
    """Replicate `model`'s weights from rank `src` to every rank: the layout blob first (state_export / state_adopt), then every
    weight arena IN PLACE — the arenas are wrapped as torch views of their device pointers (no staging copy) and go out in
    messages of `chunk_bytes`, enqueued back to back on the current stream with one synchronisation at the end.  With `comm`
    (an RcclComm) the messages are fmi_comm_broadcast calls instead of torch.distributed ones.  `model` needs state_export(),
    state_adopt(blob), state_buffers() -> [(ptr, bytes)] and state_views(device).  An export failure on `src` (StateExportUnsupported for LLM.int8
    checkpoints, or any load error recorded by the caller) is announced to every rank before the first data collective, so
    nobody blocks.  Returns {"bytes", "messages", "seconds"}."""
    # (`model.state_views(device)` -> one flat uint8 tensor per arena aliasing its memory; FluxModel builds them from
    # state_buffers() through __cuda_array_interface__)
    import time
    rank, ws = world()
    t0 = time.perf_counter()
    if ws == 1:
        return {"bytes": 0, "messages": 0, "seconds": 0.0}
    head = [None]
    local_error = None
    if rank == src:
        try:
            head = [("ok", model.state_export())]
        except Exception as e:
            # Only FMI_ERR_UNSUPPORTED (LLM.int8 matrices are not part of the flat state) means "every rank loads the checkpoint
            # itself"; anything else (a HIP error, out of memory, a bug) is a hard error on ALL ranks, announced through the
            # same head message so that nobody blocks in the data collectives.
            from ._lib import ERR_UNSUPPORTED
            kind = "unsupported" if getattr(e, "code", None) == ERR_UNSUPPORTED or isinstance(e, StateExportUnsupported) else "error"
            head = [(kind, f"{type(e).__name__}: {e}")]
            local_error = e
    dist.broadcast_object_list(head, src=src)
    kind, blob = head[0]
    if kind == "unsupported":
        raise StateExportUnsupported(blob)
    if kind != "ok":
        if local_error is not None:
            raise local_error
        raise RuntimeError(f"state export failed on rank {src}: {blob}")
    if rank != src:
        model.state_adopt(blob)
    total = msgs = 0
    dev = torch.device(device)
    views = model.state_views(dev)
    for (ptr, n), view in zip(model.state_buffers(), views):
        if not n:
            continue
        off = 0
        while off < n:
            k = min(chunk_bytes, n - off)
            if comm is not None:
                comm.broadcast(ptr + off, k, src)
            else:
                dist.broadcast(view[off:off + k], src=src)
            off += k
            total += k
            msgs += 1
    _sync(device)
    return {"bytes": total, "messages": msgs, "seconds": time.perf_counter() - t0}


def _sync(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


class RcclComm:
    """fmi_comm (csrc/rccl_comm.hip): this process's RCCL communicator behind the C-ABI, created collectively — rank 0 draws the
    128-byte id, torch.distributed (whatever its backend) carries it to the others, every rank calls fmi_comm_create on its
    current device.  Operations are enqueued on the current torch stream unless a stream pointer is given.

    THE CONSTRUCTOR IS A COLLECTIVE over `group`: every rank must construct it, in the same order relative to other collectives.
    It runs in three agreed stages so that a rank that cannot take part never leaves the others blocked inside ncclCommInitRank:
    (1) a non-collective probe on every rank (fmi_comm_probe: librccl opens, a device is current) + agree_or_raise;
    (2) rank 0 draws the id, broadcast_object_list carries it (or the failure) + agree_or_raise;
    (3) fmi_comm_create (ncclCommInitRank) + agree_or_raise."""

    def __init__(self, device, group=None):
        import ctypes as C
        from . import _lib as L
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("RcclComm needs an initialised torch.distributed process group (it carries the communicator id)")
        self.lib = L.load()
        self.device = torch.device(device)
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)
        self.h = None
        err = None
        try:  # stage 1: can this rank take part at all?  (no collective inside)
            with torch.cuda.device(self.device):
                L.check(self.lib.fmi_comm_probe(), self.lib)
        except Exception as e:
            err = e
        agree_or_raise(err, "RCCL availability probe", group)
        ident, err = [None], None
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            try:
                L.check(self.lib.fmi_comm_unique_id(buf), self.lib)
                ident = [bytes(buf)]
            except Exception as e:
                err = e
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(ident, src=src, group=group)
        h = C.c_void_p()
        if ident[0] is None:
            err = err or RuntimeError("rank 0 could not create an RCCL id")
        agree_or_raise(err, "RCCL id creation", group)  # stage 2: nobody enters ncclCommInitRank unless everybody holds the id
        try:
            with torch.cuda.device(self.device):
                L.check(self.lib.fmi_comm_create((C.c_uint8 * 128).from_buffer_copy(ident[0]), self.rank, self.world_size, C.byref(h)), self.lib)
        except Exception as e:
            err = e
        agree_or_raise(err, "RCCL communicator creation", group)
        self.h = h

    def _stream(self, stream=None):
        import ctypes as C
        return C.c_void_p(int(stream) if stream else torch.cuda.current_stream(self.device).cuda_stream)

    def broadcast(self, ptr: int, nbytes: int, root: int = 0, stream=None):
        import ctypes as C
        from . import _lib as L
        L.check(self.lib.fmi_comm_broadcast(self.h, C.c_void_p(ptr), nbytes, root, self._stream(stream)), self.lib)

    def gather(self, send: torch.Tensor, recv: Optional[torch.Tensor], root: int = 0, stream=None):
        import ctypes as C
        from . import _lib as L
        n = send.numel() * send.element_size()
        L.check(self.lib.fmi_comm_gather(self.h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()) if recv is not None else None, n, root,
                                         self._stream(stream)))

    def all_to_all(self, send: torch.Tensor, recv: torch.Tensor, stream=None):
        import ctypes as C
        from . import _lib as L
        n = send.numel() * send.element_size()
        assert n % self.world_size == 0 and recv.numel() * recv.element_size() == n
        L.check(self.lib.fmi_comm_all_to_all(self.h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), n // self.world_size, self._stream(stream)), self.lib)

    def stats(self) -> Tuple[int, int]:
        import ctypes as C
        calls, sent = C.c_ulonglong(), C.c_ulonglong()
        self.lib.fmi_comm_stats(self.h, C.byref(calls), C.byref(sent))
        return calls.value, sent.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.fmi_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
    if cmax == 0:
        return local if rank == 0 else None
    bufs = [torch.empty_like(pad) for _ in range(ws)] if rank == 0 else None
    dist.gather(pad.contiguous(), bufs, dst=0)
    if rank != 0:
        return None
    out = [None] * n_total
    for r in range(ws):
        for j, i in enumerate(shard_indices(n_total, r, ws)):
            out[i] = bufs[r][j]
    return torch.stack(out, 0)


def generate_sharded(prompts: Sequence[str], run_local: Callable[[List[str], List[int]], torch.Tensor],
                     empty: Optional[Callable[[], torch.Tensor]] = None) -> Optional[torch.Tensor]:
    """Shard `prompts` across ranks, run `run_local(my_prompts, my_sample_ids)` -> (n_local, ...) on each,
    gather to rank 0 in prompt order.  A rank without samples (fewer prompts than ranks) does not call
    `run_local`; it contributes `empty()` — a (0, ...) tensor of the output's trailing shape and dtype."""
    rank, ws = world()
    ids = shard_indices(len(prompts), rank, ws)
    if ids or empty is None:
        local = run_local([prompts[i] for i in ids], ids)
    else:
        local = empty()
    return gather_to_rank0(local, len(prompts))


# ---------------------------------------------------------------------------------------------
# Single-image sequence parallelism: Ulysses-style head sharding of the
# joint attention.  Tokens are sharded across the P ranks of a sequence-parallel group for everything
# that is per-token (LayerNorm, the q|k|v / MLP GEMMs, the residual updates); attention needs every
# key for a head, so q, k, v are redistributed  (tokens/P, all H heads) -> (all tokens, H/P heads)  by
# one all-to-all, and the attention output goes back by a second one.  These helpers are the index
# math of that exchange (layout (tokens, heads, 128)); they run on any backend (tested with gloo).
def ulysses_token_range(n_tokens: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous token shard of a rank: ceil-split, the last ranks may hold fewer (or no) tokens."""
    per = (n_tokens + world_size - 1) // world_size
    a = min(rank * per, n_tokens)
    return a, min(a + per, n_tokens)


def ulysses_head_range(n_heads: int, rank: int, world_size: int) -> Tuple[int, int]:
    if n_heads % world_size:
        raise ValueError(f"{n_heads} heads do not split over {world_size} ranks")
    per = n_heads // world_size
    return rank * per, (rank + 1) * per


def ulysses_scatter_heads(x_local: torch.Tensor, n_tokens: int, group=None) -> torch.Tensor:
    """(my tokens, H, d) -> (n_tokens, H/P, d): every rank ends with ALL tokens of ITS heads.
    One all_to_all; message to rank r = my tokens x r's heads (bytes = tokens/P * H/P * d * 2 for bf16)."""
    rank, ws = world()
    if ws == 1:
        return x_local
    H, d = x_local.shape[1], x_local.shape[2]
    send = [x_local[:, slice(*ulysses_head_range(H, r, ws)), :].contiguous() for r in range(ws)]
    recv = [torch.empty((ulysses_token_range(n_tokens, r, ws)[1] - ulysses_token_range(n_tokens, r, ws)[0], H // ws, d),
                        dtype=x_local.dtype, device=x_local.device) for r in range(ws)]
    _all_to_all(recv, send, group)
    return torch.cat(recv, 0)


def ulysses_gather_heads(o_heads: torch.Tensor, n_tokens: int, n_heads: int, group=None) -> torch.Tensor:
    """Inverse of ulysses_scatter_heads: (n_tokens, H/P, d) -> (my tokens, H, d)."""
    rank, ws = world()
    if ws == 1:
        return o_heads
    d = o_heads.shape[2]
    send = [o_heads[slice(*ulysses_token_range(n_tokens, r, ws))].contiguous() for r in range(ws)]
    a, b = ulysses_token_range(n_tokens, rank, ws)
    recv = [torch.empty((b - a, n_heads // ws, d), dtype=o_heads.dtype, device=o_heads.device) for _ in range(ws)]
    _all_to_all(recv, send, group)
    return torch.cat(recv, 1)


def _all_to_all(recv: List[torch.Tensor], send: List[torch.Tensor], group=None):
    """One all-to-all with ragged pieces: all_to_all_single over flattened buffers (RCCL and gloo both have it)."""
    flat_send = torch.cat([t.reshape(-1) for t in send])
    flat_recv = torch.empty(sum(t.numel() for t in recv), dtype=flat_send.dtype, device=flat_send.device)
    dist.all_to_all_single(flat_recv, flat_send, [t.numel() for t in recv], [t.numel() for t in send], group=group)
    off = 0
    for t in recv:
        t.copy_(flat_recv[off:off + t.numel()].view_as(t))
        off += t.numel()


# ---------------------------------------------------------------------------------------------
# The wired path: FluxModel.set_sequence_parallel + this class.  The C library packs the exchange buffers and calls
# back for the collective (include/flux_mi355x.h: fmi_all_to_all_fn); here the collective is torch.distributed's
# all_to_all_single — RCCL over xGMI with the "nccl" backend (equal splits: one message of bytes_per_peer to each peer),
# a host-staged exchange with gloo (tests: two ranks sharing one GPU).
class _DeviceBytes:
    """A device buffer known by address, visible to torch without a copy (__cuda_array_interface__)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2, "strides": None}


class SequenceParallel:
    """Sequence-parallel group for ONE image: rank r holds txt tokens [r*T/N, (r+1)*T/N) and img tokens [r*S/N, (r+1)*S/N).

    On the "nccl" backend THE CONSTRUCTOR IS A COLLECTIVE over `group` (it creates the library's own RCCL communicator, a second
    one next to torch's: RcclComm) — every rank of the group must construct it.  FMI_SP_TORCH_A2A (1 = keep the Python
    all_to_all_single callback instead) must have the SAME value on every rank: ranks that disagree would wait for each other in
    different collectives; the constructor checks that and raises on all ranks."""

    def __init__(self, device, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("SequenceParallel needs an initialised torch.distributed process group")
        self.group = group
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device(device)
        self.backend = dist.get_backend(group)
        self._views = {}
        self._exchanges = 0
        self._bytes_sent = 0
        # RCCL: the library's own communicator does the exchange from C (fmi_comm_all_to_all); FMI_SP_TORCH_A2A=1 keeps the
        # round-2 path (a Python callback per exchange that calls torch.distributed.all_to_all_single) for A/B runs
        import os
        self.comm = None
        native = self.backend == "nccl" and os.environ.get("FMI_SP_TORCH_A2A", "0") != "1"
        if self.backend == "nccl" and self.world_size > 1:  # the choice must be unanimous (see the class docstring)
            votes = [None] * self.world_size
            dist.all_gather_object(votes, bool(native), group=group)
            if len(set(votes)) != 1:
                raise RuntimeError(f"FMI_SP_TORCH_A2A differs between the ranks of the sequence-parallel group: native exchange = {votes}")
        if native:
            self.comm = RcclComm(self.device, group)

    @property
    def exchanges(self) -> int:
        return self.comm.stats()[0] if self.comm is not None else self._exchanges

    @property
    def bytes_sent(self) -> int:
        return self.comm.stats()[1] if self.comm is not None else self._bytes_sent

    def _view(self, ptr: int, nbytes: int) -> torch.Tensor:
        key = (ptr, nbytes)
        t = self._views.get(key)
        if t is None:
            t = self._views[key] = torch.as_tensor(_DeviceBytes(ptr, nbytes), device=self.device)
        return t

    def all_to_all(self, send: int, recv: int, bytes_per_peer: int, stream) -> None:
        n = bytes_per_peer * self.world_size
        st, rt = self._view(send, n), self._view(recv, n)
        ext = torch.cuda.ExternalStream(int(stream), device=self.device) if stream else torch.cuda.default_stream(self.device)
        with torch.cuda.stream(ext):
            if self.backend == "nccl":
                dist.all_to_all_single(rt, st, group=self.group)  # ordered on `ext` by the process group
            else:  # gloo has no device all-to-all: stage through the host (st.cpu() waits for the packing kernel on `ext`)
                hs = st.cpu()
                hr = torch.empty_like(hs)
                dist.all_to_all_single(hr, hs, group=self.group)
                rt.copy_(hr)
        self._exchanges += 1
        self._bytes_sent += n - bytes_per_peer

    def attach(self, flux_model) -> None:
        if self.comm is not None:
            flux_model.set_sequence_parallel_native(self.rank, self.world_size, self.comm)
        else:
            flux_model.set_sequence_parallel(self.rank, self.world_size, self.all_to_all)

    def detach(self, flux_model) -> None:
        flux_model.set_sequence_parallel(0, 1, None)

    def shard(self, x: torch.Tensor, dim: int = 1) -> torch.Tensor:
        """This rank's equal share of axis `dim` (the token axis)."""
        n = x.shape[dim]
        if n % self.world_size:
            raise ValueError(f"{n} tokens do not split evenly over {self.world_size} ranks")
        per = n // self.world_size
        return x.narrow(dim, self.rank * per, per).contiguous()

    def gather(self, x_local: torch.Tensor, dim: int = 1) -> torch.Tensor:
        """Inverse of shard: every rank gets the full axis back (all_gather; host-staged for gloo)."""
        if self.backend == "nccl":
            parts = [torch.empty_like(x_local) for _ in range(self.world_size)]
            dist.all_gather(parts, x_local.contiguous(), group=self.group)
            return torch.cat(parts, dim)
        h = x_local.contiguous().cpu()
        parts = [torch.empty_like(h) for _ in range(self.world_size)]
        dist.all_gather(parts, h, group=self.group)
        return torch.cat(parts, dim).to(x_local.device)


def sequence_parallel_rank_time(flux, world_size: int, timesteps, device, S: int = 4096, T: int = 512, seed: int = 0) -> dict:
    """What ONE rank of an N-rank sequence-parallel group computes per denoise step, measured on one GPU: the model on 1/N of the
    tokens, the pack / unpack kernels and the attention of H/N heads over all tokens, with the all-to-all replaced by a loopback
    (the rank's own first send block copied into every receive block: right sizes, one kernel per exchange, meaningless pixels).
    Wire time of the 2 exchanges per block comes on top; `MB_sent_per_step` says how much would travel.  Measurement aid
    (tools/sp_rank_time.py, bench.py's `secondary.sequence_parallel_rank`), not a substitute for a multi-GPU run."""
    import time
    dev = torch.device(device)
    cfg = flux.cfg
    N = int(world_size)
    stats = {"calls": 0, "bytes": 0}
    views = {}

    def loopback(send, recv, nbytes, stream):
        key = (send, recv, nbytes)
        if key not in views:
            views[key] = (torch.as_tensor(_DeviceBytes(send, nbytes), device=dev).expand(N, nbytes),
                          torch.as_tensor(_DeviceBytes(recv, nbytes * N), device=dev).view(N, nbytes))
        src, dst = views[key]
        dst.copy_(src)
        stats["calls"] += 1
        stats["bytes"] += nbytes * (N - 1)

    flux.set_sequence_parallel(0, N, loopback if N > 1 else None)
    try:
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        Sl, Tl = S // N, T // N
        lat = torch.randn((1, Sl, cfg["in_channels"]), generator=g, device=dev)
        ids = torch.zeros((1, Sl, 3), device=dev)
        txt = torch.randn((1, Tl, cfg["joint_attention_dim"]), generator=g, device=dev).to(torch.bfloat16)
        tids = torch.zeros((1, Tl, 3), device=dev)
        y = torch.randn((1, cfg["pooled_projection_dim"]), generator=g, device=dev)
        gd = torch.full((1,), 3.5, device=dev) if flux.is_guidance() else None
        ns = len(timesteps) - 1
        flux.denoise(lat, ids, txt, tids, y, gd, list(timesteps[:3]))  # warm-up: workspace and exchange buffers
        torch.cuda.synchronize(dev)
        stats["calls"] = stats["bytes"] = 0
        t0 = time.perf_counter()
        flux.denoise(lat, ids, txt, tids, y, gd, list(timesteps))
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 1e3 / ns
        calls, sent = stats["calls"], stats["bytes"]
        flux.set_profiling(True)  # second pass with a device sync per phase: where the time goes
        flux.denoise(lat, ids, txt, tids, y, gd, list(timesteps))
        phases = {k: round(v / ns, 2) for k, v in flux.phase_ms().items() if v > 0}
        flux.set_profiling(False)
    finally:
        flux.set_sequence_parallel(0, 1, None)
    return {"ranks": N, "tokens_per_rank": Sl + Tl, "heads_per_rank": cfg["num_attention_heads"] // N, "ms_per_step_compute": round(ms, 2),
            "exchanges_per_step": calls // ns, "MB_sent_per_step": round(sent / ns / 1e6, 1), "phase_ms_per_step": phases}
