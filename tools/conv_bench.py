#!/usr/bin/env python3
"""The VAE decoder's convolution shapes at a 1024 x 1024 decode (latent 128 x 128), one by one through the C-ABI
(fmi_conv2d_nhwc), timed with events on the launch stream.  `python tools/conv_bench.py [lib.so ...]`: several libraries
= A/B of builds on one box (each in its own process; the first argument of a child is the library it loads).

Shapes (vaes/vae.rs:371-456 at the public FLUX AutoencoderKL config 128-256-512-512, 3 resnets per level):
(in_h, in_w, Cin, Cout, k, upsample, count per decode)."""
import ctypes as C
import os
import subprocess
import sys

SHAPES = [
    (128, 128, 64, 512, 3, 0, 1),     # conv_in (16 channels padded to 64)
    (128, 128, 512, 512, 3, 0, 10),   # mid block (4) + level 0 (6)
    (128, 128, 512, 512, 3, 1, 1),    # upsample conv -> 256^2
    (256, 256, 512, 512, 3, 0, 6),
    (256, 256, 512, 512, 3, 1, 1),    # -> 512^2
    (512, 512, 512, 256, 3, 0, 1),
    (512, 512, 512, 256, 1, 0, 1),    # shortcut
    (512, 512, 256, 256, 3, 0, 5),
    (512, 512, 256, 256, 3, 1, 1),    # -> 1024^2
    (1024, 1024, 256, 128, 3, 0, 1),
    (1024, 1024, 256, 128, 1, 0, 1),  # shortcut
    (1024, 1024, 128, 128, 3, 0, 5),
    (1024, 1024, 128, 64, 3, 0, 1),   # conv_out (3 channels padded)
]


def child(path, iters):
    import torch
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    lib.fmi_last_error.restype = C.c_char_p
    lib.fmi_conv2d_nhwc.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p]
    assert lib.fmi_init(0) == 0, lib.fmi_last_error()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    tot_us = tot_fl = 0.0
    print(f"== {path}")
    for (H, W, Cin, Cout, ks, up, cnt) in SHAPES:
        oh, ow = (H * 2, W * 2) if up else (H, W)
        x = (torch.randn(1, H, W, Cin, device=dev, generator=g) * 0.5).bfloat16()
        w = (torch.randn(Cout, ks, ks, Cin, device=dev, generator=g) * 0.02).bfloat16()
        b = torch.zeros(Cout, device=dev).bfloat16()
        out = torch.empty(1, oh, ow, Cout, device=dev, dtype=torch.bfloat16)
        run = lambda: lib.fmi_conv2d_nhwc(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), 1, H, W, Cin, Cout, ks, up, None)
        for _ in range(3):
            assert run() == 0, lib.fmi_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        fl = 2.0 * oh * ow * Cout * ks * ks * Cin
        tot_us += us * cnt
        tot_fl += fl * cnt
        chk = float(out.float().abs().sum())
        print(f"  in {H:4d}x{W:<4d} {Cin:3d}->{Cout:3d} k{ks} up{up} x{cnt:2d}: {us:8.1f} us  {fl / us * 1e-6:7.1f} TF   checksum {chk:.6e}")
    print(f"  per decode: {tot_us * 1e-3:.2f} ms for {tot_fl * 1e-12:.2f} TFLOP = {tot_fl / tot_us * 1e-6:.1f} TF")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        libs = [a for a in sys.argv[1:] if not a.isdigit()] or [os.path.join(root, "diffusion-rs_amd", "libflux_mi355x.so")]
        iters = next((a for a in sys.argv[1:] if a.isdigit()), "20")
        for rep in range(2):
            for lib in libs:
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(lib), iters], check=True)
