#!/usr/bin/env python3
"""Per-rank time of the sequence-parallel denoise step on ONE GPU.

A 1-GPU box cannot run the collective, but it can run everything else a rank does: the model on 1/N of the tokens, the pack /
unpack kernels, and the attention over all L tokens of 24/N heads.  The all-to-all is replaced by a loopback (the rank's own
send block copied into every receive block: right sizes and kernels, meaningless pixels), so the number printed is the
compute a rank adds per denoise step; the wire time of the two exchanges per block comes on top (sizes printed)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import diffusion_rs_amd as d  # noqa: E402
from diffusion_rs_amd import synth  # noqa: E402
from diffusion_rs_amd import dist as fdist  # noqa: E402


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
    NS = args.steps
    sched = d.SchedulerConfig()
    ts = sched.get_timesteps(NS, sched.calculate_shift(4096))
    first = None
    for N in [int(x) for x in args.worlds.split(",")]:
        row = fdist.sequence_parallel_rank_time(flux, N, ts, dev)
        row["split_k"] = bool(args.split_k)
        if first is None:
            first = row["ms_per_step_compute"]
        else:
            row["speedup_vs_first_row_before_wire_time"] = round(first / row["ms_per_step_compute"], 2)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
