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


def _worker_bf16(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from stcd_amd.ddp import FlatGradReducer, init_distributed

    init_distributed(backend="gloo")
    m = StandInModel(n0=1001, n1=333)                 # bucket sizes that do not divide by the world size
    FlatGradReducer(m, bucket_dtype="bf16")
    gen = torch.Generator().manual_seed(5)
    base = torch.randn(m.w.numel(), generator=gen)    # the same on both ranks; rank r contributes base * (r + 1) + r * 1e-3
    g = base * (rank + 1) + rank * 1e-3
    for stage, (b, e) in ((0, m.stage0_range), (1, m.stage1_range)):
        m.grad_stage_hook(stage, g[b:e])
    per_rank = [(base * (r + 1) + r * 1e-3).to(torch.bfloat16).float() for r in range(world)]
    want = (sum(per_rank) / world).to(torch.bfloat16).float()          # one rounding of the inputs, fp32 mean, one rounding of the mean
    q.put((rank, bool(torch.equal(g, want))))
    dist.destroy_process_group()


def test_two_rank_bf16_buckets_accumulate_in_fp32():
    """bucket_dtype="bf16": the result is bf16(mean_fp32(bf16(g_rank))) on every rank, bit for bit (SURVEY 8e option)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bf16, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(2)) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------------------------------
# The REAL stage ranges of every engine family (stcd_grad_stage_range through the C ABI: no GPU needed for queries) drive
# the reducer: the two slices must be disjoint, cover the flat gradient buffer, stage 0 must hold exactly the decoder's
# tensors (final first, so its collective overlaps the encoder's backward), and a 2-rank run over those slices must
# deliver the mean everywhere -- also with a trainer-shaped epoch metric all-reduce.
ARCHES = [("diff", 0), ("conc", 1), ("sub", 2), ("fcef", 9), ("xconc", 10), ("snunet", 3), ("segcd", 4), ("segcd_resnet18", 5), ("segcd_resnet101", 7),
          ("unetseg_resnet34", 18), ("ffctlcd_resnet50", 32), ("changeformer", 64)]
SINGLE_STAGE = ("snunet", "segcd", "unetseg", "ffctlcd")


def _engine_layout(arch):
    from stcd_amd.engine import Engine
    e = Engine(arch, 3, 2, "bf16")
    return e.param_floats, e.stage0_range, e.stage1_range, [(p.name, p.offset, p.numel) for p in e.params]


@pytest.mark.parametrize("arch,_id", ARCHES)
def test_stage_ranges_partition_the_gradient_buffer(arch, _id):
    total, s0, s1, params = _engine_layout(arch)
    assert s0[0] <= s0[1] and s1[0] <= s1[1]
    assert {s0[0], s1[0]} >= {0} and max(s0[1], s1[1]) == total          # one of them starts at 0, one ends at the end
    assert s1[1] == s0[0] or s0[1] == s1[0]                              # adjacent
    assert (s0[1] - s0[0]) + (s1[1] - s1[0]) == total                    # disjoint + covering
    for name, off, numel in params:                                       # no tensor straddles the boundary
        in0 = s0[0] <= off and off + numel <= s0[1]
        in1 = s1[0] <= off and off + numel <= s1[1]
        assert in0 != in1, name
        if not arch.startswith(SINGLE_STAGE):
            dec = name.startswith("TDec_x2.") if arch == "changeformer" else (name.split(".")[0].endswith("d") or name.startswith(("upconv", "cross_conc")))
            assert in0 == dec, f"{name}: stage 0 must finalise exactly the decoder's gradients"
    if arch.startswith(SINGLE_STAGE):
        assert s1[1] - s1[0] == 0                                         # single-stage plan: everything final after stage 0


class RangedModel(torch.nn.Module):
    def __init__(self, total, s0, s1):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(total))
        self.stage0_range, self.stage1_range = s0, s1
        self.grad_stage_hook = None

    def fake_backward(self, rank):
        g = (torch.arange(self.w.numel(), dtype=torch.float32) % 977) * (rank + 1)
        for stage, (b, e) in ((0, self.stage0_range), (1, self.stage1_range)):
            self.grad_stage_hook(stage, g[b:e])
        self.w.grad = g


def _ranged_worker(rank, world, port, q, layouts):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from stcd_amd.ddp import FlatGradReducer, init_distributed
    init_distributed(backend="gloo")
    ok = True
    for total, s0, s1 in layouts:
        for overlap in (True, False):
            m = RangedModel(total, s0, s1)
            FlatGradReducer(m, overlap=overlap)
            m.fake_backward(rank)
            want = (torch.arange(total, dtype=torch.float32) % 977) * sum(range(1, world + 1)) / world
            ok = ok and bool(torch.allclose(m.w.grad, want, rtol=1e-6))
    cm = torch.tensor([[10.0 + rank, 1.0], [2.0, 3.0 * (rank + 1)]], dtype=torch.float64)     # trainer-shaped metric sync
    dist.all_reduce(cm, op=dist.ReduceOp.SUM)
    ok = ok and cm.tolist() == [[21.0, 2.0], [4.0, 9.0]]
    q.put((rank, ok))
    dist.destroy_process_group()


def test_two_ranks_over_real_stage_ranges_gloo():
    layouts = []
    for arch, _ in ARCHES:
        total, s0, s1, _p = _engine_layout(arch)
        layouts.append((total, s0, s1))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ranged_worker, args=(r, 2, port, q, layouts)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}
