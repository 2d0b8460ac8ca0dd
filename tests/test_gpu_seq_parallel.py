"""Single-image sequence parallelism (include/flux_mi355x.h: fmi_flux_set_sequence_parallel).

The tokens of ONE image are sharded over N ranks; the joint attention trades (local tokens, all heads) for (all tokens,
H/N heads) through the caller's all-to-all.  Every kernel involved moves or computes rows independently of which other
rows share its launch, so the sharded forward has to reproduce the single-device forward BIT FOR BIT — that is the bar here.

Two harnesses, both on one GPU:
  * N ranks as N threads of this process, each with its own model handle; the "collective" is a barrier plus device
    copies between the ranks' exchange buffers (tests the C side: pack / unpack kernels, buffer sizes, head split);
  * 2 processes under torch.distributed (gloo, host-staged exchange) through Pipeline.enable_sequence_parallel
    (tests dist.SequenceParallel and the pipeline front door).
"""
import os
import socket
import threading

import numpy as np
import pytest

from tests.util import SMALL_FLUX, dev, flux_inputs

pytestmark = pytest.mark.gpu

FLUX4 = dict(SMALL_FLUX, num_attention_heads=4)  # D = 512: splits over 2 and 4 ranks


class ThreadRanks:
    """N model handles + an in-process all-to-all: rank r's recv block p <- rank p's send block r."""

    def __init__(self, d, torch, cfg, sd, N):
        from diffusion_rs_amd.dist import _DeviceBytes
        self.torch, self.N = torch, N
        self.view = lambda ptr, n: torch.as_tensor(_DeviceBytes(ptr, n), device="cuda:0")
        self.models = []
        for r in range(N):
            m = d.FluxModel(cfg)
            m.load_state_dict(sd)
            self.models.append(m)
        self.barrier = threading.Barrier(N, timeout=120)
        self.posted = [None] * N
        self.calls = [0] * N
        self.bytes_per_peer = [set() for _ in range(N)]
        for r, m in enumerate(self.models):
            m.set_sequence_parallel(r, N, self._make_a2a(r))

    def _make_a2a(self, r):
        def a2a(send, recv, nbytes, stream):
            torch = self.torch
            torch.cuda.synchronize()  # my packing kernel is done
            self.posted[r] = (send, nbytes)
            self.barrier.wait()
            for p in range(self.N):
                src_ptr, src_n = self.posted[p]
                assert src_n == nbytes, "ranks disagree on the message size"
                self.view(recv + p * nbytes, nbytes).copy_(self.view(src_ptr + r * nbytes, nbytes))
            torch.cuda.synchronize()
            self.barrier.wait()  # nobody repacks its send buffer before everyone has read it
            self.calls[r] += 1
            self.bytes_per_peer[r].add(nbytes)
        return a2a

    def run(self, fn):
        """fn(rank, model) on every rank concurrently; returns the list of results (re-raises the first failure)."""
        out, err = [None] * self.N, [None] * self.N

        def body(r):
            try:
                self.torch.cuda.set_device(0)
                out[r] = fn(r, self.models[r])
            except BaseException as e:  # noqa: BLE001 — reported below; break the barrier so the other ranks do not hang
                err[r] = e
                self.barrier.abort()

        ts = [threading.Thread(target=body, args=(r,)) for r in range(self.N)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        first = next((e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or next((e for e in err if e is not None), None)
        if first is not None:
            raise first
        return out


def _shard(x, r, N, axis=1):
    per = x.shape[axis] // N
    return np.ascontiguousarray(np.take(x, range(r * per, (r + 1) * per), axis=axis))


@pytest.fixture(scope="module")
def env():
    import torch
    import diffusion_rs_amd as d
    return dict(torch=torch, d=d)


@pytest.mark.parametrize("cfg_name,N,S_hw,T", [("small", 2, (8, 8), 32), ("h4", 2, (8, 12), 48), ("h4", 4, (16, 16), 64), ("h4", 4, (6, 6), 8)])
def test_sequence_parallel_forward_is_bit_identical_to_one_device(env, cfg_name, N, S_hw, T):
    torch, d = env["torch"], env["d"]
    cfg = SMALL_FLUX if cfg_name == "small" else FLUX4
    sd = d.synth.flux_state_dict_numpy(cfg, seed=3)
    img, ids, txt, txt_ids, y = flux_inputs(cfg, 1, S_hw, T, seed=11)
    t, g = np.array([0.7], np.float32), np.array([3.5], np.float32)
    one = d.FluxModel(cfg)
    one.load_state_dict(sd)
    ref = one.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g))
    torch.cuda.synchronize()

    ranks = ThreadRanks(d, torch, cfg, sd, N)

    def fwd(r, m):
        o = m.forward(dev(_shard(img, r, N)), dev(_shard(ids, r, N)), dev(_shard(txt, r, N), torch.bfloat16), dev(_shard(txt_ids, r, N)), dev(t), dev(y), dev(g))
        torch.cuda.synchronize()
        return o

    got = torch.cat(ranks.run(fwd), 1)
    assert torch.isfinite(got).all()
    blocks = cfg["num_layers"] + cfg["num_single_layers"]
    assert ranks.calls == [2 * blocks] * N  # two exchanges per transformer block, on every rank
    H, Ll = cfg["num_attention_heads"], (S_hw[0] * S_hw[1] + T) // N
    Hr, Lpl = H // N, (Ll + 63) // 64 * 64
    assert ranks.bytes_per_peer[0] == {(2 * Hr * Ll * 128 + Hr * 128 * Lpl) * 2, Ll * Hr * 128 * 2}
    mism = int((got.view(torch.int32) != ref.view(torch.int32)).sum())
    # and against the CPU oracle directly ("the SP tests compare the HIP path with the HIP path"): the sharded forward is
    # Flux::forward (model.rs:790-833) of the whole image, at the tolerance of the single-device forward
    from oracle import oracle as orc
    from tests.util import host, rel_l2
    om = orc.Flux(cfg)
    om.load(sd)
    err = rel_l2(host(got), om.forward(img, ids, txt, txt_ids, t, y, g))
    print(f"SP forward {cfg_name} N={N} S={S_hw} T={T}: {mism} mismatching elements of {got.numel()}, max |diff| {float((got - ref).abs().max()):.3e}; vs oracle {err:.3e}")
    assert mism == 0 and err <= 1e-2


def test_sequence_parallel_denoise_is_bit_identical_and_switches_off(env):
    torch, d = env["torch"], env["d"]
    cfg, N, S_hw, T, steps = FLUX4, 2, (8, 8), 32, 3
    sd = d.synth.flux_state_dict_numpy(cfg, seed=5)
    img, ids, txt, txt_ids, y = flux_inputs(cfg, 1, S_hw, T, seed=2)
    g = np.array([3.5], np.float32)
    sched = d.SchedulerConfig()
    ts = sched.get_timesteps(steps, sched.calculate_shift(S_hw[0] * S_hw[1]))
    one = d.FluxModel(cfg)
    one.load_state_dict(sd)
    ref = one.denoise(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(y), dev(g), ts)
    ranks = ThreadRanks(d, torch, cfg, sd, N)
    got = torch.cat(ranks.run(lambda r, m: m.denoise(dev(_shard(img, r, N)), dev(_shard(ids, r, N)), dev(_shard(txt, r, N), torch.bfloat16),
                                                     dev(_shard(txt_ids, r, N)), dev(y), dev(g), ts)), 1)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    # switched off again, a handle is an ordinary single-device model
    m0 = ranks.models[0]
    m0.set_sequence_parallel(0, 1, None)
    again = m0.denoise(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(y), dev(g), ts)
    assert torch.equal(again, ref)


def test_split_k_latency_mode_is_deterministic_and_equal_to_rounding(env):
    """fmi_flux_set_split_k: residual projections of small launches are cut along K and reduced in a fixed order — the same
    result to f32 rounding (stated: rel-L2 <= 1e-3 on a forward), identical from run to run, and the oracle tolerance holds."""
    torch, d = env["torch"], env["d"]
    from oracle import oracle as orc
    from tests.util import host, rel_l2
    cfg = FLUX4
    sd = d.synth.flux_state_dict_numpy(cfg, seed=8)
    img, ids, txt, txt_ids, y = flux_inputs(cfg, 1, (16, 16), 64, seed=4)
    t, g = np.array([0.8], np.float32), np.array([3.5], np.float32)
    args = (dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g))
    m = d.FluxModel(cfg)
    m.load_state_dict(sd)
    plain = m.forward(*args)
    m.set_split_k(True)
    a, b = m.forward(*args), m.forward(*args)
    assert torch.equal(a, b)
    err = rel_l2(host(a), host(plain))
    n_diff = int((a != plain).sum())
    om = orc.Flux(cfg)
    om.load(sd)
    err_o = rel_l2(host(a), om.forward(img, ids, txt, txt_ids, t, y, g))
    print(f"split-K vs unsplit: rel-L2 {err:.2e}, {n_diff} of {a.numel()} f32 outputs differ; vs oracle {err_o:.2e}")
    assert n_diff > 0, "the latency mode did not engage on this shape"
    assert err <= 1e-3 and err_o <= 1e-2
    # with sequence parallelism on top (2 ranks): equal to the single-device result to the same tolerance
    ranks = ThreadRanks(d, torch, cfg, sd, 2)
    for mm in ranks.models:
        mm.set_split_k(True)
    got = torch.cat(ranks.run(lambda r, mm: mm.forward(dev(_shard(img, r, 2)), dev(_shard(ids, r, 2)), dev(_shard(txt, r, 2), torch.bfloat16),
                                                       dev(_shard(txt_ids, r, 2)), dev(t), dev(y), dev(g))), 1)
    torch.cuda.synchronize()
    assert rel_l2(host(got), host(plain)) <= 1e-3
    m.set_split_k(False)
    assert torch.equal(m.forward(*args), plain)
    # long enough for the attention's key split (>= 16 KV tiles, few heads per rank): 4 ranges of whole tiles + log-sum-exp merge
    img, ids, txt, txt_ids, y = flux_inputs(cfg, 1, (32, 34), 64, seed=6)  # L = 1152: 18 tiles
    args = (dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g))
    plain = m.forward(*args)
    ref_o = om.forward(img, ids, txt, txt_ids, t, y, g)
    outs = []
    for rep in range(2):
        outs.append(torch.cat(ranks.run(lambda r, mm: mm.forward(dev(_shard(img, r, 2)), dev(_shard(ids, r, 2)), dev(_shard(txt, r, 2), torch.bfloat16),
                                                                 dev(_shard(txt_ids, r, 2)), dev(t), dev(y), dev(g))), 1))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    e_plain, e_or, e_plain_or = rel_l2(host(outs[0]), host(plain)), rel_l2(host(outs[0]), ref_o), rel_l2(host(plain), ref_o)
    print(f"2 ranks, split-K + key-split attention, L = 1152: vs unsplit single device {e_plain:.2e}; vs oracle {e_or:.2e} (unsplit vs oracle {e_plain_or:.2e})")
    assert e_plain <= 3e-3 and e_or <= 1e-2


def test_rank_time_measurement_aid_runs_and_counts_the_exchanges(env):
    """dist.sequence_parallel_rank_time (bench.py's `secondary.sequence_parallel_rank`, tools/sp_rank_time.py)."""
    torch, d = env["torch"], env["d"]
    from diffusion_rs_amd import dist as fdist
    m = d.FluxModel(FLUX4)
    m.load_state_dict(d.synth.flux_state_dict_numpy(FLUX4, seed=0))
    sched = d.SchedulerConfig()
    ts = sched.get_timesteps(4, sched.calculate_shift(64))
    row = fdist.sequence_parallel_rank_time(m, 4, ts, "cuda:0", S=64, T=32)
    assert row["ranks"] == 4 and row["tokens_per_rank"] == 24 and row["heads_per_rank"] == 1
    assert row["exchanges_per_step"] == 2 * (FLUX4["num_layers"] + FLUX4["num_single_layers"])
    Ll, Hr, Lpl = 24, 1, 64
    assert abs(row["MB_sent_per_step"] - 4 * 3 * ((2 * Hr * Ll * 128 + Hr * 128 * Lpl) * 2 + Ll * Hr * 128 * 2) / 1e6) < 0.06
    assert row["ms_per_step_compute"] > 0 and "attention" in row["phase_ms_per_step"]
    # the handle is an ordinary single-device model again afterwards
    img, ids, txt, txt_ids, y = flux_inputs(FLUX4, 1, (8, 8), 32, seed=1)
    out = m.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(np.array([0.5], np.float32)), dev(y), dev(np.array([3.5], np.float32)))
    assert torch.isfinite(out).all()


def test_sequence_parallel_rejects_what_it_cannot_split(env):
    torch, d = env["torch"], env["d"]
    m = d.FluxModel(SMALL_FLUX)  # 2 heads
    with pytest.raises(d.FmiError):
        m.set_sequence_parallel(0, 3, lambda *a: None)  # 2 heads over 3 ranks
    with pytest.raises(d.FmiError):
        m.set_sequence_parallel(2, 2, lambda *a: None)  # rank outside the group
    with pytest.raises(d.FmiError):
        m.set_sequence_parallel(0, 2, None)  # no collective
    m.load_state_dict(d.synth.flux_state_dict_numpy(SMALL_FLUX, seed=0))
    m.set_sequence_parallel(0, 2, lambda *a: None)
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, 2, (4, 4), 16)
    with pytest.raises(d.FmiError):  # one image at a time
        m.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(np.ones(2, np.float32)), dev(y), dev(np.ones(2, np.float32)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pipeline_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import diffusion_rs_amd as d
        from tests.util import SMALL_VAE
        pipe = d.Pipeline.load(d.ModelSource.Synthetic("dev", seed=4, flux_cfg=FLUX4, vae_cfg=SMALL_VAE))
        params = d.DiffusionGenerationParams(height=128, width=128, num_steps=3, guidance_scale=3.5)
        prompts = ["a red fox", "a blue heron"]
        # single device, one image per call like the sequence-parallel mode (both ranks compute the same)
        one_by_one = lambda: torch.cat([pipe.generate_tensor(prompts[i:i + 1], params, seed=9, sample_ids=[i]) for i in range(len(prompts))], 0).cpu()
        ref = one_by_one()
        sp = pipe.enable_sequence_parallel()
        got = pipe.generate_tensor(prompts, params, seed=9).cpu()  # all ranks hold the full images
        front = pipe.forward(prompts, params, seed=9, output="tensor")
        pipe.disable_sequence_parallel()
        after = one_by_one()
        print(f"rank {rank}: SP vs single device: {int((ref != got).sum())} of {ref.numel()} u8 values differ; after switching off: {int((ref != after).sum())}", flush=True)
        q.put((rank, bool(torch.equal(ref, got)), front is None if rank else bool(torch.equal(front.cpu(), ref)), bool(torch.equal(after, ref)),
               sp.exchanges, sp.bytes_sent, None))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, False, False, False, 0, 0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_pipeline_sequence_parallel_two_processes_gloo_one_gpu(env):
    """dist.SequenceParallel + Pipeline.enable_sequence_parallel: 2 ranks sharing this GPU, gloo (host-staged exchange)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, same, front_ok, after_ok, exchanges, sent, err in sorted(res):
        assert err is None, err
        assert same, f"rank {rank}: sequence-parallel image differs from the single-device one"
        assert front_ok and after_ok
        assert exchanges == 4 * 3 * 4 * 2, exchanges  # (2 + 2 images) x 3 steps x (2 + 2 blocks) x 2 exchanges
        print(f"rank {rank}: {exchanges} exchanges, {sent / 1e6:.2f} MB sent")
