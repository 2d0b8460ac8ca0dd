"""fmi_comm (csrc/rccl_comm.hip): the library's own RCCL communicator behind the C-ABI.  One GPU here, so the group has one rank:
this checks that librccl is found and bound (dlopen), that creation / all-to-all / broadcast / gather run as stream-ordered
operations with the documented semantics in the degenerate group, and the error paths.  The N > 1 semantics are RCCL's own
(ncclAllToAll / ncclBroadcast / grouped send-recv); the byte layout the library expects from the all-to-all is pinned by the
sequence-parallel tests, which run the same packing kernels against an in-process exchange."""
import ctypes as C
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_comm_single_rank_group():
    import torch
    import torch.distributed as dist
    from diffusion_rs_amd import _lib as L
    from diffusion_rs_amd import dist as fd
    lib = L.load()
    L.check(lib.fmi_init(0))
    own_group = not dist.is_initialized()
    if own_group:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        comm = fd.RcclComm("cuda:0")
        assert lib.fmi_comm_rank(comm.h) == 0 and lib.fmi_comm_world_size(comm.h) == 1
        g = torch.Generator(device="cuda").manual_seed(3)
        a = torch.randint(0, 255, (1 << 20,), dtype=torch.uint8, device="cuda", generator=g)
        b = torch.zeros_like(a)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):  # enqueued on the caller's stream, behind the producer of `a2`
            a2 = a + 1
            comm.all_to_all(a2, b)
            c = b.clone()
        side.synchronize()
        assert torch.equal(c, a + 1)
        keep = a.clone()
        comm.broadcast(a.data_ptr(), a.numel(), 0)
        out = torch.zeros_like(a)
        comm.gather(a, out, 0)
        torch.cuda.synchronize()
        assert torch.equal(a, keep) and torch.equal(out, keep)
        calls, sent = comm.stats()
        assert calls == 3 and sent == a.numel()  # a one-rank all-to-all / gather sends nothing; the broadcast root counts its bytes
        # error paths: status + message, no crash
        assert lib.fmi_comm_broadcast(comm.h, C.c_void_p(a.data_ptr()), 16, 5, None) < 0 and b"root" in lib.fmi_last_error()
        assert lib.fmi_comm_all_to_all(None, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), 16, None) < 0
        h = C.c_void_p()
        assert lib.fmi_comm_create((C.c_uint8 * 128)(), 2, 2, C.byref(h)) < 0  # rank out of range
        comm.close()
    finally:
        if own_group:
            dist.destroy_process_group()
