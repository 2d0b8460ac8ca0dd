"""BASELINE config C3 (FLUX.1-dev nf4, 1024x1024: M = 4608 rows) — the fused dequant-GEMM at the real shapes.

BnbLinear::forward (diffusion_rs_backend/src/bitsandbytes/mod.rs:301-312) = dequantise the weight to bf16, then
matmul.  The fused kernel (gemm_w4q.h) must therefore equal "stand-alone dequant kernel + dense MFMA GEMM" BIT FOR BIT
(same bf16 operands, same accumulation order), and both must equal the CPU oracle's linear on the oracle-dequantised
weights within the GEMM tolerance (rel-L2 <= 4e-3, bf16 output).  The oracle handles a sample of 256 output rows per
shape (a full 4608 x 21504 x 3072 f32 GEMM would take minutes on the host)."""
import ctypes as C

import numpy as np
import pytest

from tests.util import dev, host, rel_l2

pytestmark = pytest.mark.gpu

QT = {"fp4": 1, "nf4": 2}


@pytest.fixture(scope="module")
def env():
    import torch
    import diffusion_rs_amd as d
    from diffusion_rs_amd import _lib as L
    from oracle import oracle as orc
    lib = L.load()
    L.check(lib.fmi_init(0))
    return torch, d, L, lib, orc


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _quantize(torch, orc, w, blocksize, qt):
    """bitsandbytes blockwise 4-bit quantisation on the host oracle (quantisation is a loader-side operation;
    the reference only ever dequantises)."""
    packed, absmax = orc.quantize_blockwise_4bit(w.float().cpu().numpy().ravel(), blocksize, qt)
    return torch.from_numpy(packed).cuda(), torch.from_numpy(absmax).cuda()


def _run(torch, L, lib, orc, M, N, K, blocksize=64, qt="nf4", bias=True, epi=0, seed=0, oracle_rows=256):
    g = torch.Generator(device="cuda").manual_seed(1000 + seed)
    w = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    b = (torch.randn((N,), generator=g, device="cuda") * 0.1).to(torch.bfloat16) if bias else None
    packed, absmax = _quantize(torch, orc, w, blocksize, qt)
    # path 1: fused dequant-GEMM on the packed codes
    y_fused = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    L.check(lib.fmi_linear_bnb4_bf16(_p(x), _p(packed), _p(absmax), blocksize, QT[qt], _p(b), _p(y_fused), M, N, K, epi, None))
    # path 2: the reference's own structure — stand-alone dequant kernel (bit-exact vs dequant.cu), then the dense GEMM
    wdq = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
    getattr(lib, f"dequantize_blockwise_bf16_{qt}")(None, _p(packed), _p(absmax), _p(wdq), blocksize, N * K, None)
    y_dense = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    L.check(lib.fmi_linear_bf16(_p(x), _p(wdq), _p(b), _p(y_dense), M, N, K, epi, None))
    torch.cuda.synchronize()
    nbad = int((y_fused.view(torch.int16) != y_dense.view(torch.int16)).sum())
    # oracle: dequantise on the CPU (bit-exact check of the GPU dequant), linear on a row sample
    wdq_o = orc.dequantize_blockwise(None, packed.cpu().numpy(), absmax.cpu().numpy(), blocksize, N * K, qt, "bf16").reshape(N, K)
    assert np.array_equal(host(wdq), wdq_o), "GPU dequant differs from the oracle"
    rows = np.unique(np.linspace(0, M - 1, min(M, oracle_rows)).astype(np.int64))
    ref = orc.linear(host(x)[rows], wdq_o, None if b is None else host(b))
    if epi == 1:
        ref = orc.gelu(ref)
    err = rel_l2(host(y_fused)[rows], ref)
    return nbad, err


@pytest.mark.parametrize("N,K", [(9216, 3072), (21504, 3072), (3072, 3072), (3072, 15360), (12288, 3072), (3072, 12288)])
def test_fused_nf4_gemm_at_c3_shapes(env, N, K):
    """Every block-linear shape of FLUX.1-dev at M = 4608 (S = 4096 image + T = 512 text tokens)."""
    torch, d, L, lib, orc = env
    nbad, err = _run(torch, L, lib, orc, 4608, N, K, seed=N + K)
    print(f"nf4 fused vs dequant+dense at M=4608 N={N} K={K}: {nbad} differing outputs; vs oracle (256 rows) rel-L2 {err:.3e}")
    assert nbad == 0
    assert err <= 4e-3


@pytest.mark.parametrize("M,N,K,bs,qt,epi", [
    (4096, 9216, 3072, 64, "fp4", 0),     # fp4 table
    (512, 9216, 3072, 64, "nf4", 0),      # the text stream of a double block
    (1000, 3072, 128, 64, "nf4", 1),      # ragged M (clamped rows), two K tiles, GELU epilogue
    (300, 512, 64, 64, "nf4", 0),         # a single K tile: prologue + tail only
    (777, 1280, 192, 64, "fp4", 1),       # three K tiles, ragged M
    (2048, 3072, 4096, 128, "nf4", 0),    # absmax block spans two K tiles
    (1024, 768, 2048, 1024, "nf4", 0),    # ... sixteen K tiles (bitsandbytes' largest common blocksize)
    (4608, 3328, 3072, 64, "nf4", 0),     # N = 13 tiles: the last W tile's rows 3328.. are clamped
])
def test_fused_4bit_gemm_edge_shapes(env, M, N, K, bs, qt, epi):
    torch, d, L, lib, orc = env
    nbad, err = _run(torch, L, lib, orc, M, N, K, blocksize=bs, qt=qt, epi=epi, seed=M + N + K + bs)
    print(f"{qt} bs={bs} epi={epi} M={M} N={N} K={K}: {nbad} differing outputs; vs oracle rel-L2 {err:.3e}")
    assert nbad == 0
    assert err <= 4e-3


def test_fused_nf4_gemm_is_deterministic_and_survives_small_row_dispatch(env):
    """Run-to-run determinism at the C3 shape (the pipeline mixes LDS-DMA, register loads and LDS stores: a missed wait
    shows up as run-to-run differences), and the row threshold between the two fused kernels changes nothing."""
    torch, d, L, lib, orc = env
    M, N, K = 4608, 3072, 3072
    g = torch.Generator(device="cuda").manual_seed(5)
    w = torch.randn((N, K), generator=g, device="cuda") / K ** 0.5
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    packed, absmax = d.synth.quantize_nf4_device(w.to(torch.bfloat16), 64)
    outs = []
    for rep in range(6):
        y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        L.check(lib.fmi_linear_bnb4_bf16(_p(x), _p(packed), _p(absmax), 64, 2, None, _p(y), M, N, K, 0, None))
        outs.append(y)
    torch.cuda.synchronize()
    for y in outs[1:]:
        assert torch.equal(y.view(torch.int16), outs[0].view(torch.int16))
    try:
        L.check(lib.fmi_set_bnb4_onewave_min_rows(1 << 30))  # everything on the two-workgroups-per-CU kernel
        y2 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        L.check(lib.fmi_linear_bnb4_bf16(_p(x), _p(packed), _p(absmax), 64, 2, None, _p(y2), M, N, K, 0, None))
        torch.cuda.synchronize()
    finally:
        L.check(lib.fmi_set_bnb4_onewave_min_rows(256))
    assert torch.equal(y2.view(torch.int16), outs[0].view(torch.int16))


def test_nf4_model_at_c3_tokens_fused_equals_dense_cache_and_holds_no_bf16_copy(env):
    """FLUX.1 width (D = 3072) with nf4 block AND modulation linears at the C3 token counts (S = 4096, T = 512), one
    double + one single block: the fused dequant-GEMM (packed weights only), the default policy (per-call expansion into a
    scratch for launches this large) and the opt-in expanded cache give the same prediction bit for bit, and only the cache
    allocates the bf16 arenas."""
    torch, d, L, lib, orc = env
    from tests.test_gpu_fullsize import WIDE
    from tests.util import flux_inputs
    gm = d.FluxModel(WIDE)
    g = torch.Generator(device="cuda").manual_seed(11)
    nq = 0
    for name, shape in d.synth.flux_tensor_shapes(WIDE).items():
        if "norm_q" in name or "norm_k" in name or "norm_added" in name:
            t = torch.ones(shape, dtype=torch.bfloat16, device="cuda")
        elif name.endswith(".bias"):
            t = (torch.randn(shape, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
        else:
            t = (torch.randn(shape, generator=g, device="cuda") * d.synth._std_for(name, 0.02, 0.01)).to(torch.bfloat16)
        quant = name.endswith(".weight") and (d.synth.is_block_linear(name) or "norm" in name and "linear" in name)
        if quant:
            packed, absmax = d.synth.quantize_nf4_device(t, 64)
            gm.set_linear_bnb4(name[:-len(".weight")], packed, absmax, 64, "nf4", shape[0], shape[1])
            nq += 1
        else:
            gm.set_tensor(name, t)
    gm.assert_complete()
    assert nq == 21  # double block: 2 modulation + 12 block linears; single block: 1 + 5; norm_out.linear
    img, ids, txt, txt_ids, y = flux_inputs(WIDE, 1, (64, 64), 512, seed=3)
    t = np.array([0.7], np.float32)
    gd = np.array([3.5], np.float32)
    args = (dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(gd))
    gm.set_quant_dense_cache(2)  # every launch on the fused kernels
    a = host(gm.forward(*args))
    bufs = gm.state_buffers()
    assert bufs[1][1] == 0 and bufs[2][1] == 0 and bufs[3][1] > 0  # no MOD / BLOCKS (bf16) arena, only BASE + Q4
    fused_bytes = gm.size_in_bytes()
    gm.set_quant_dense_cache(0)  # default: by size — these 4096 / 4608-row launches expand per call into the scratch
    c = host(gm.forward(*args))
    bufs = gm.state_buffers()
    assert bufs[1][1] == 0 and bufs[2][1] == 0
    packed_bytes = gm.size_in_bytes()
    assert 0 < packed_bytes - fused_bytes <= 2 * (3 * 3072 + 12288) * 3072 * 2 + 4096  # the two scratch matrices, nothing else
    gm.set_quant_dense_cache(1)
    b = host(gm.forward(*args))
    bufs = gm.state_buffers()
    assert bufs[1][1] > 0 and bufs[2][1] > 0
    assert np.isfinite(a).all() and np.array_equal(a, b) and np.array_equal(a, c)
    print(f"nf4 D=3072 1+1 blocks at S=4096,T=512: fused == per-call expansion == dense cache bit for bit; resident {fused_bytes / 2**20:.0f} MiB fused, "
          f"{packed_bytes / 2**20:.0f} MiB with the scratch, {gm.size_in_bytes() / 2**20:.0f} MiB with the expanded cache")
    gm.close()


def test_nf4_full_model_modulation_matrix_above_2e31_elements_with_400_rows(env):
    """FLUX.1-dev in full (19 + 38 blocks) as an nf4 checkpoint — the fused modulation matrix is
    344 D x D = 3.25e9 weights, more than 2^31 — with 8 prompts x 50 steps = 400 rows in the modulation precompute of
    fmi_flux_denoise (Pipeline.MAX_BATCH = 8 makes this reachable).  The launch must stay on the fused dequant-GEMM (the
    per-call scratch and the 32-bit dequant launchers cannot hold that matrix): the 8-sample denoise equals the same samples
    run 4 + 4 (200 rows, always fused) bit for bit, and the expanded-cache mode (row-chunked one-time expansion) gives the same
    bits.  Few tokens, so the 50 steps stay cheap."""
    torch, d, L, lib, orc = env
    cfg = dict(d.FLUX_DEV)
    gm = d.FluxModel(cfg)
    g = torch.Generator(device="cuda").manual_seed(21)
    for name, shape in d.synth.flux_tensor_shapes(cfg).items():
        if "norm_q" in name or "norm_k" in name or "norm_added" in name:
            t = torch.ones(shape, dtype=torch.bfloat16, device="cuda")
        elif name.endswith(".bias"):
            t = (torch.randn(shape, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
        else:
            t = torch.randn(shape, generator=g, device="cuda", dtype=torch.bfloat16)
            t.mul_(d.synth._std_for(name, 0.02, 0.01))
        if name.endswith(".weight") and (d.synth.is_block_linear(name) or "norm" in name and "linear" in name):
            packed, absmax = d.synth.quantize_nf4_device(t, 64)
            gm.set_linear_bnb4(name[:-len(".weight")], packed, absmax, 64, "nf4", shape[0], shape[1])
            del packed, absmax
        else:
            gm.set_tensor(name, t)
        del t
    gm.assert_complete()
    gm.set_quant_dense_cache(0)  # packed only: the single blocks' 512-row launches expand per call (the default would expand them once on this part)
    B, T, hw, steps = 8, 32, (4, 8), 50
    from tests.util import flux_inputs
    img, ids, txt, txt_ids, y = flux_inputs(cfg, B, hw, T, seed=6)
    gd = np.full((B,), 3.5, np.float32)
    sched = d.SchedulerConfig()
    ts = sched.get_timesteps(steps, sched.calculate_shift(hw[0] * hw[1]))
    run = lambda sl: host(gm.denoise(dev(img[sl]), dev(ids[sl]), dev(txt[sl], torch.bfloat16), dev(txt_ids[sl]), dev(y[sl]), dev(gd[sl]), ts))
    full = run(slice(0, 8))  # 400 modulation rows
    halves = np.concatenate([run(slice(0, 4)), run(slice(4, 8))], 0)  # 200 + 200
    assert np.isfinite(full).all() and np.array_equal(full, halves)
    bufs = gm.state_buffers()
    assert bufs[1][1] == 0 and bufs[2][1] == 0  # still no bf16 arena
    gm.set_quant_dense_cache(1)  # one-time expansion of every matrix, the 3.25e9-weight one in row chunks
    cached = run(slice(0, 8))
    assert np.array_equal(full, cached)
    print(f"nf4 FLUX.1-dev, 8 x 50 = 400 modulation rows over a 3.25e9-weight matrix: fused == 4+4 == expanded cache bit for bit "
          f"({gm.size_in_bytes() / 2**30:.1f} GiB with the cache)")
    gm.close()
