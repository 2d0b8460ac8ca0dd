#!/usr/bin/env python3
"""The VAE decode of a 1024 x 1024 image (latent 128 x 128) alone, FLUX AutoencoderKL config, random weights: milliseconds per decode over events on the
launch stream.  `python tools/vae_bench.py [lib.so ...]`: each library in its own process (FMI_LIB), two rounds — A/B of builds on one box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    import diffusion_rs_amd as d
    from diffusion_rs_amd import synth
    dev = torch.device("cuda", 0)
    vae = d.AutoEncoderKl(d.VAE_FLUX, 0)
    synth.fill_vae_random_device(vae, seed=1, device=dev)
    z = torch.randn(1, 16, 128, 128, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    for _ in range(3):
        out = vae.decode(z)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        out = vae.decode(z)
    e1.record()
    torch.cuda.synchronize()
    print(f"  {os.environ.get('FMI_LIB', 'in-tree library')}: {e0.elapsed_time(e1) / n:.2f} ms per 1024^2 decode   checksum {float(out.float().abs().sum()):.6e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        libs = sys.argv[1:] or [""]
        for rep in range(2):
            for lib in libs:
                env = dict(os.environ)
                if lib:
                    env["FMI_LIB"] = os.path.abspath(lib)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], check=True, env=env)
