"""The data-parallel gradient path on CPU with two gloo ranks: the hook protocol of the engine modules
(hook(stage, final_flat_gradient_slice)) driven by a stand-in model, so the collective logic is covered without a GPU.
The real-engine equivalence (2 ranks == average of 2 sequential micro-batches) is in test_ddp_gpu.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class StandInModel(torch.nn.Module):
    """Flat gradient buffer + two-stage hook, like stcd_amd.modules.HipChangeDetector._run_backward."""

    def __init__(self, n0=1000, n1=500):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(n0 + n1))
        self.stage1_range, self.stage0_range = (0, n1), (n1, n0 + n1)   # encoder first in the flat buffer
        self.grad_stage_hook = None

    def fake_backward(self, rank):
        g = torch.arange(self.w.numel(), dtype=torch.float32) * (rank + 1)
        for stage, (b, e) in ((0, self.stage0_range), (1, self.stage1_range)):
            if self.grad_stage_hook is not None:
                self.grad_stage_hook(stage, g[b:e])
        self.w.grad = g


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from stcd_amd.ddp import FlatGradReducer, broadcast_parameters, init_distributed

    r, lr, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    m = StandInModel()
    with torch.no_grad():
        m.w.fill_(float(rank + 3))
    broadcast_parameters(m)
    assert float(m.w[0]) == 3.0                       # rank 0's replica everywhere
    FlatGradReducer(m)
    assert m.grad_stage_hook is not None
    m.fake_backward(rank)
    want = torch.arange(m.w.numel(), dtype=torch.float32) * sum(range(1, world + 1)) / world
    ok = torch.allclose(m.w.grad, want, rtol=1e-6)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gradient_average_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}
