#!/usr/bin/env python
"""CPU only (oracle): which of the HIP VAE decoder's bf16 roundings cost its u8 agreement with the f32 decoder?  The oracle's decoder with bf16 roundings
switched on at chosen places (orc_vae_set_study_rounding: bit 0 conv outputs inside a ResnetBlock, 1 GroupNorm outputs = conv operands, 2 the residual stream,
3 the mid attention's operands, 4 the final image) against itself in f32, real AutoencoderKL config.   python tools/vae_rounding_study.py [latent_side]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusion_rs_amd as d  # noqa: E402
from oracle import oracle as orc  # noqa: E402

h = int(sys.argv[1]) if len(sys.argv) > 1 else 32
orc.set_threads(orc.usable_cpus())
sd = d.synth.vae_state_dict_numpy(d.VAE_FLUX, seed=4)
ov = orc.Vae(d.VAE_FLUX)
ov.load(sd)
z = np.random.default_rng(h).standard_normal((1, 16, h, h)).astype(np.float32)
lib = orc.lib()
lib.orc_vae_set_study_rounding(0)
ref = ov.decode(z)
u8 = lambda x: np.clip(np.rint((np.clip(x, -1, 1) + 1) * 127.5), 0, 255).astype(np.int32)
ru = u8(ref)
names = {5: "residual stream at the last level only", 0: "conv1 outputs (norm2 inputs)", 1: "GroupNorm outputs (conv operands)", 2: "residual stream", 3: "mid-attention operands", 4: "final image"}
print(f"# VAE decode, real FLUX AutoencoderKL config, latent {h}x{h} -> {8 * h}^2; the oracle with bf16 roundings at the named places vs itself in f32")
for mask in (0b011111, 0b000001, 0b000010, 0b000100, 0b001000, 0b010000, 0b011011, 0b011010, 0b001010, 0b000111, 0b100000, 0b111011):
    lib.orc_vae_set_study_rounding(mask)
    got = ov.decode(z)
    du = np.abs(u8(got) - ru)
    rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    what = " + ".join(names[b] for b in range(6) if mask >> b & 1)
    print(f"  mask {mask:06b}  rel-L2 {rel:.3e}  u8 max |d| {int(du.max())}  within 2: {float((du <= 2).mean()):.5f}  within 1: {float((du <= 1).mean()):.5f}   [{what}]", flush=True)
lib.orc_vae_set_study_rounding(0)
