#!/usr/bin/env python
"""A/B timing of a library toggle inside ONE process on ONE box (box-to-box variance is +-2 %):
    python tools/ab_toggle.py fused_qkv      # fmi_flux_set_fused_qkv_relayout 0 / 1
    python tools/ab_toggle.py mod_gemm [steps]   # fmi_flux_set_modulation_gemm 0 / 1 (default 50 steps: the precompute is per image)
FLUX.1-dev at the C2 shape, random weights, 10 denoise steps per measurement, alternating A,B,A,B."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import diffusion_rs_amd as d  # noqa: E402
from diffusion_rs_amd import _lib as L  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "fused_qkv"
    lib = L.load()
    flux = d.FluxModel(d.FLUX_DEV, 0)
    d.synth.fill_flux_random_device(flux, seed=0)
    dev = torch.device("cuda", 0)
    B, h, w, T = 1, 128, 128, 512
    g = torch.Generator(device=dev).manual_seed(1)
    txt = torch.randn((B, T, 4096), generator=g, device=dev).to(torch.bfloat16)
    y = torch.randn((B, 768), generator=g, device=dev)
    guidance = torch.full((B,), 3.5, device=dev)
    txt_ids = torch.zeros((B, T, 3), device=dev)
    sched = d.SchedulerConfig()
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else (50 if what == "mod_gemm" else 10)
    ts = sched.get_timesteps(50, sched.calculate_shift(4096))[:nsteps + 1]
    lat = d.randn_latents(B, 16, h, w, seed=3, device=dev)
    img, img_ids = d.pack_latents(lat)

    def setmode(v):
        if what == "fused_qkv":
            L.check(lib.fmi_flux_set_fused_qkv_relayout(flux.h, v))
        elif what == "mod_gemm":
            L.check(lib.fmi_flux_set_modulation_gemm(flux.h, v))
        else:
            raise SystemExit("unknown toggle")

    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = flux.denoise(img, img_ids, txt, txt_ids, y, guidance, ts)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / nsteps * 1e3, out

    setmode(1)
    run()
    res = {0: [], 1: []}
    outs = {}
    for rep in range(4 if nsteps <= 10 else 2):
        for v in (0, 1):
            setmode(v)
            ms, out = run()
            res[v].append(ms)
            outs[v] = out
    setmode(1)
    for v in (0, 1):
        print(f"{what}={v}: ms per denoise step {['%.2f' % x for x in res[v]]}  best {min(res[v]):.2f}")
    print("outputs bit-identical:", bool(torch.equal(outs[0], outs[1])))


if __name__ == "__main__":
    main()
