#!/usr/bin/env python3
"""Full-size soak of the "results do not depend on other work on the GPU" guarantee (DESIGN §4.4 rule 3): FLUX.1-dev at the C2
shape, 10 denoise steps per repetition, while a second process keeps the device busy; every repetition must reproduce the latents
obtained on the idle device bit for bit.  (tests/test_gpu_shared_device.py does the same on a small model in every test run.)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import diffusion_rs_amd as d  # noqa: E402
from diffusion_rs_amd import synth  # noqa: E402
from tests.test_gpu_shared_device import CoRunner  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
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
    sched = d.SchedulerConfig()
    ts = sched.get_timesteps(10, sched.calculate_shift(4096))
    lat = torch.randn((1, 4096, 64), generator=g, device=dev)
    ids = torch.zeros((1, 4096, 3), device=dev)
    txt = torch.randn((1, 512, 4096), generator=g, device=dev).to(torch.bfloat16)
    tids = torch.zeros((1, 512, 3), device=dev)
    y = torch.randn((1, 768), generator=g, device=dev)
    gd = torch.full((1,), 3.5, device=dev)
    run = lambda: flux.denoise(lat, ids, txt, tids, y, gd, ts)
    ref = run()
    torch.cuda.synchronize()
    t0 = time.time()
    run()
    torch.cuda.synchronize()
    idle_ms = (time.time() - t0) * 1e3 / 10
    co = CoRunner(seconds + 30)
    bad = reps = 0
    t0 = time.time()
    try:
        while time.time() - t0 < seconds:
            bad += int(not torch.equal(run(), ref))
            reps += 1
        alive = co.alive()
    finally:
        co.stop()
    busy_ms = (time.time() - t0) * 1e3 / (10 * reps)
    print(json.dumps({"model": "FLUX.1-dev, 4096 + 512 tokens, 10 denoise steps per repetition", "repetitions_next_to_the_co_runner": reps,
                      "repetitions_differing_from_the_idle_result": bad, "co_runner_alive_until_the_end": alive,
                      "ms_per_step_idle": round(idle_ms, 1), "ms_per_step_next_to_the_co_runner": round(busy_ms, 1)}))


if __name__ == "__main__":
    main()
