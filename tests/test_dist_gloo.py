"""world_size-2 test of the batch-sharding path on CPU (gloo): sample -> rank mapping, weight
broadcast from rank 0, ragged gather back to rank 0 in prompt order."""
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
    q.put((rank, {k: v.numpy() for k, v in got.items()}, None if out is None else out.numpy(), fd.shard_indices(n_prompts, rank, world)))
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
