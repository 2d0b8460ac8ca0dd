"""world_size-2 tests of the multi-GPU paths on CPU (gloo): sample -> rank mapping, weight broadcast from rank 0
(tensor-wise and as flat arenas), ragged gather back to rank 0 in prompt order, Pipeline.forward's sharded front door
(SURVEY §8e), the index math of the Ulysses head redistribution, the token shards of dist.SequenceParallel and the
collective error agreement (one rank's failure raises on every rank)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_prompts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from diffusion_rs_amd import dist as fd

    # 1. weights: rank 0 makes them, everyone receives identical tensors
    shapes = {"a.weight": (4, 8), "b.bias": (8,), "c.weight": (3, 5)}
    got = {}
    g = torch.Generator().manual_seed(123)
    nbytes = fd.broadcast_tensors(shapes, lambda n, s: torch.randn(s, generator=g), lambda n, t: got.__setitem__(n, t.clone()), "cpu",
                                  dtype=torch.float32)
    assert nbytes == (32 + 8 + 15) * 4
    # 2. sharded generation: "image" of sample i is a deterministic function of (prompt, sample id)
    prompts = [f"p{i}" for i in range(n_prompts)]

    def run_local(my_prompts, ids):
        assert my_prompts == [prompts[i] for i in ids]
        if not ids:
            return torch.zeros((0, 3, 4, 4), dtype=torch.uint8)
        return torch.stack([torch.full((3, 4, 4), 10 * i + len(p), dtype=torch.uint8) for p, i in zip(my_prompts, ids)])

    out = fd.generate_sharded(prompts, run_local)

    # 3. flat-arena broadcast (dist.broadcast_state) with a stand-in for FluxModel's state_* methods
    class StubModel:
        def __init__(self):
            self.blob = None
            self.bufs = [torch.zeros(0, dtype=torch.uint8)] * 4

        def state_export(self):
            return b"FMIS-stub:" + bytes([1, 0, 1, 1])

        def state_adopt(self, blob):
            self.blob = blob
            self.bufs = [torch.zeros(n, dtype=torch.uint8) for n in (1000, 0, 2500, 77)]

        def state_buffers(self):
            return [(1 if b.numel() else 0, b.numel()) for b in self.bufs]

        def state_views(self, device):
            return [b if b.numel() else None for b in self.bufs]

    sm = StubModel()
    if rank == 0:
        gg = torch.Generator().manual_seed(7)
        sm.bufs = [torch.randint(0, 256, (n,), dtype=torch.uint8, generator=gg) for n in (1000, 0, 2500, 77)]
    st = fd.broadcast_state(sm, "cpu", chunk_bytes=1024)  # forces several messages per arena
    assert st["bytes"] == 3577 and st["messages"] == 1 + 3 + 1
    assert rank == 0 or sm.blob == b"FMIS-stub:" + bytes([1, 0, 1, 1])
    state_sum = [int(b.to(torch.int64).sum()) for b in sm.bufs]

    # 4. the front door: Pipeline.forward shards prompts when torch.distributed is initialised
    from diffusion_rs_amd import pipeline as pl

    class StubPipeline(pl.Pipeline):
        def __init__(self):  # no GPU: only forward()'s sharding / gather logic is under test
            self.device = torch.device("cpu")
            self.calls = []

        def generate_tensor(self, my_prompts, params, *, embeddings=None, latents=None, seed=None, sample_ids=None, **kw):
            self.calls.append((list(my_prompts), list(sample_ids)))
            assert embeddings is None or embeddings[0].shape[0] == len(my_prompts)
            e = torch.zeros(len(my_prompts)) if embeddings is None else embeddings[0][:, 0]
            return torch.stack([torch.full((3, params.height, params.width), (7 * i + len(p) + int(e[j])) % 256, dtype=torch.uint8)
                                for j, (p, i) in enumerate(zip(my_prompts, sample_ids))]) if my_prompts else torch.empty((0, 3, params.height, params.width), dtype=torch.uint8)

    sp = StubPipeline()
    params = pl.DiffusionGenerationParams(8, 8, 2, 3.5)
    emb = (torch.arange(n_prompts, dtype=torch.float32)[:, None].repeat(1, 4), torch.zeros(n_prompts, 2))
    fwd = sp.forward(prompts, params, output="tensor", embeddings=emb, first_sample=100)
    assert (fwd is None) == (rank != 0)
    assert sp.calls == ([([prompts[i] for i in fd.shard_indices(n_prompts, rank, world)], [100 + i for i in fd.shard_indices(n_prompts, rank, world)])]
                        if fd.shard_indices(n_prompts, rank, world) else [])

    # 5. Ulysses head redistribution: scatter tokens->heads, "attention" as a per-head reduction over all tokens, gather back
    H, d, n_tok = 4, 8, 11  # 11 tokens over 2 ranks: ragged (6 + 5)
    full = torch.arange(n_tok * H * d, dtype=torch.float32).reshape(n_tok, H, d)
    a, b = fd.ulysses_token_range(n_tok, rank, world)
    heads = fd.ulysses_scatter_heads(full[a:b].clone(), n_tok)
    h0, h1 = fd.ulysses_head_range(H, rank, world)
    assert torch.equal(heads, full[:, h0:h1])  # every token of my heads, in token order
    attn = heads + heads.sum(0, keepdim=True)  # needs all tokens of a head, nothing of other heads
    back = fd.ulysses_gather_heads(attn, n_tok, H)
    assert torch.equal(back, (full + full.sum(0, keepdim=True))[a:b])

    # 6. the wired sequence-parallel group (dist.SequenceParallel): token shards and their inverse, and the callback handed to the
    #    C library — block p of the send buffer to rank p, block p of the receive buffer from rank p — on host memory here
    spg = fd.SequenceParallel("cpu")
    assert (spg.rank, spg.world_size, spg.backend) == (rank, world, "gloo")
    tok = torch.arange(2 * 8 * 3, dtype=torch.float32).reshape(2, 8, 3)
    mine = spg.shard(tok)
    assert torch.equal(mine, tok[:, rank * 4:(rank + 1) * 4])
    assert torch.equal(spg.gather(mine), tok)
    try:
        spg.shard(torch.zeros(1, 7, 3))
        raise AssertionError("7 tokens must not split over 2 ranks")
    except ValueError:
        pass

    # 7. collective error agreement: a failure on ONE rank raises on ALL of them (nobody is left waiting in the next collective),
    #    and a source whose state cannot travel as flat arenas (LLM.int8) is announced before the first data message
    try:
        fd.agree_or_raise(ValueError("boom") if rank == 1 else None, "test step")
        raise AssertionError("agree_or_raise must raise on every rank")
    except ValueError:
        assert rank == 1
    except RuntimeError as e:
        assert rank == 0 and "rank 1" in str(e) and "boom" in str(e)
    fd.agree_or_raise(None, "nothing failed")

    from diffusion_rs_amd._lib import ERR_HIP, ERR_UNSUPPORTED, FmiError

    class Int8Stub(StubModel):
        def state_export(self):  # what FluxModel.state_export raises for LLM.int8 matrices: FMI_ERR_UNSUPPORTED
            raise FmiError("fmi status -4: state_export: LLM.int8 matrices are not part of the flat state", code=ERR_UNSUPPORTED)

    try:
        fd.broadcast_state(Int8Stub(), "cpu")
        raise AssertionError("broadcast_state must announce an unexportable state")
    except fd.StateExportUnsupported as e:
        assert "LLM.int8" in str(e)

    # ... but ONLY that status means "every rank loads the checkpoint itself": any other failure of the export (a HIP error, out of
    # memory, a bug) is a hard error on every rank, never a silent fall-back to N local loads
    class BrokenStub(StubModel):
        def state_export(self):
            raise FmiError("fmi status -2: hipMemcpy: an illegal memory access was encountered", code=ERR_HIP)

    try:
        fd.broadcast_state(BrokenStub(), "cpu")
        raise AssertionError("a failed export must raise")
    except fd.StateExportUnsupported:
        raise AssertionError("a HIP error is not 'unsupported'")
    except FmiError as e:
        assert rank == 0 and e.code == ERR_HIP
    except RuntimeError as e:
        assert rank == 1 and "illegal memory access" in str(e) and "rank 0" in str(e)

    q.put((rank, {k: v.numpy() for k, v in got.items()}, None if out is None else out.numpy(), fd.shard_indices(n_prompts, rank, world),
           state_sum, None if fwd is None else fwd.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_prompts, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_prompts, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


def test_shard_broadcast_gather_world2():
    for n_prompts in (5, 1, 4):  # ragged, fewer prompts than ranks, even
        r0, r1 = _run(n_prompts)
        for k in r0[1]:
            np.testing.assert_array_equal(r0[1][k], r1[1][k])  # identical weights on both ranks
        assert r1[2] is None
        out = r0[2]
        assert out.shape == (n_prompts, 3, 4, 4)
        for i in range(n_prompts):
            assert (out[i] == 10 * i + len(f"p{i}")).all()  # prompt order restored
        assert sorted(r0[3] + r1[3]) == list(range(n_prompts)) and r0[3] == list(range(0, n_prompts, 2))
        assert r0[4] == r1[4] and r0[4][1] == 0 and r0[4][0] > 0  # flat arenas identical on both ranks
        assert r1[5] is None and r0[5].shape == (n_prompts, 3, 8, 8)
        for i in range(n_prompts):  # Pipeline.forward: prompt order, per-sample embeddings and Philox stream ids followed the shard
            assert (r0[5][i] == (7 * (100 + i) + len(f"p{i}") + i) % 256).all()
