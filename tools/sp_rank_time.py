#!/usr/bin/env python3
"""Per-rank time of the sequence-parallel denoise step on ONE GPU.

A 1-GPU box cannot run the collective, but it can run everything else a rank does: the model on 1/N of the tokens, the pack /
unpack kernels, and the attention over all L tokens of 24/N heads.  The all-to-all is replaced by a loopback (the rank's own
send block copied into every receive block: right sizes and kernels, meaningless pixels), so the number printed is the
compute a rank adds per denoise step; the wire time of the two exchanges per block comes on top (sizes printed)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import diffusion_rs_amd as d  # noqa: E402
from diffusion_rs_amd import synth  # noqa: E402
from diffusion_rs_amd.dist import _DeviceBytes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--split-k", action="store_true", help="latency mode: FluxModel.set_split_k(True)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    flux = d.FluxModel(d.FLUX_DEV, 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    for name, shape in synth.flux_tensor_shapes(d.FLUX_DEV).items():
        if "norm_q" in name or "norm_k" in name or "norm_added" in name:
            t = torch.ones(shape, dtype=torch.bfloat16, device=dev)
        elif name.endswith(".bias"):
            t = torch.zeros(shape, dtype=torch.bfloat16, device=dev)
        else:
            t = torch.randn(shape, generator=g, device=dev, dtype=torch.bfloat16)
            t.mul_(synth._std_for(name, 0.02, 0.01))
        flux.set_tensor(name, t)
        del t
    flux.assert_complete()
    flux.set_split_k(args.split_k)
    S, T, NS = 4096, 512, args.steps
    sched = d.SchedulerConfig()
    ts = sched.get_timesteps(NS, sched.calculate_shift(S))
    out = []
    for N in [int(x) for x in args.worlds.split(",")]:
        stats = {"calls": 0, "bytes": 0}
        views = {}

        def loopback(send, recv, nbytes, stream, N=N, stats=stats, views=views):
            key = (send, recv, nbytes)
            if key not in views:  # block 0 of the send buffer stands in for every peer's block: ONE copy kernel per exchange
                views[key] = (torch.as_tensor(_DeviceBytes(send, nbytes), device=dev).expand(N, nbytes),
                              torch.as_tensor(_DeviceBytes(recv, nbytes * N), device=dev).view(N, nbytes))
            src, dst = views[key]
            dst.copy_(src)
            stats["calls"] += 1
            stats["bytes"] += nbytes * (N - 1)

        flux.set_sequence_parallel(0, N, loopback if N > 1 else None)
        Sl, Tl = S // N, T // N
        lat = torch.randn((1, Sl, 64), generator=g, device=dev)
        ids = torch.zeros((1, Sl, 3), device=dev)
        txt = torch.randn((1, Tl, 4096), generator=g, device=dev).to(torch.bfloat16)
        tids = torch.zeros((1, Tl, 3), device=dev)
        y = torch.randn((1, 768), generator=g, device=dev)
        gd = torch.full((1,), 3.5, device=dev)
        flux.denoise(lat, ids, txt, tids, y, gd, ts[:3])  # warm-up (workspace, exchange buffers)
        torch.cuda.synchronize()
        stats["calls"] = stats["bytes"] = 0
        t0 = time.time()
        flux.denoise(lat, ids, txt, tids, y, gd, ts)
        torch.cuda.synchronize()
        ms = (time.time() - t0) * 1e3 / NS
        row = {"ranks": N, "split_k": bool(args.split_k), "tokens_per_rank": Sl + Tl, "heads_per_rank": 24 // N, "ms_per_step_compute": round(ms, 2),
               "exchanges_per_step": stats["calls"] // NS, "MB_sent_per_rank_per_step": round(stats["bytes"] / NS / 1e6, 1)}
        flux.set_profiling(True)  # second pass with a device sync per phase: where the time goes
        flux.denoise(lat, ids, txt, tids, y, gd, ts)
        row["phase_ms_per_step"] = {k: round(v / NS, 2) for k, v in flux.phase_ms().items() if v > 0}
        flux.set_profiling(False)
        if out:
            row["speedup_vs_1_before_wire_time"] = round(out[0]["ms_per_step_compute"] / ms, 2)
        out.append(row)
        print(json.dumps(row), flush=True)
    flux.set_sequence_parallel(0, 1, None)


if __name__ == "__main__":
    main()
