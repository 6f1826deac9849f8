"""Two data-parallel ranks (sharing cuda:0, gloo transport) through the real engine == the average of the same two
micro-batches run one after the other in one process (BatchNorm statistics are per replica in both, like the
reference's DataParallel -- SURVEY.md section 8e)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _data(rank):
    rng = np.random.default_rng(50 + rank)
    x1 = torch.from_numpy(rng.standard_normal((2, 3, 32, 32)).astype(np.float32))
    x2 = torch.from_numpy(rng.standard_normal((2, 3, 32, 32)).astype(np.float32))
    y = torch.from_numpy((rng.random((2, 32, 32)) < 0.3).astype(np.int64))
    return x1, x2, y


def _grads_single(rank_data, state):
    from stcd_amd.modules import SiamUnet_diff
    m = SiamUnet_diff(3, 2, dtype="fp32")
    m.load_state_dict(state)
    m.to("cuda:0").train()
    m.set_dropout_p(0.0)
    x1, x2, y = rank_data
    torch.nn.functional.cross_entropy(m(x1.cuda(), x2.cuda()), y.cuda()).backward()
    return torch.cat([p.grad.flatten() for p in m.parameters()]).cpu()


def _worker(rank, world, port, q, state):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from stcd_amd.ddp import FlatGradReducer, broadcast_parameters
    from stcd_amd.modules import SiamUnet_diff

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = SiamUnet_diff(3, 2, dtype="fp32")
    m.load_state_dict(state)
    m.to("cuda:0").train()
    m.set_dropout_p(0.0)
    broadcast_parameters(m)
    FlatGradReducer(m, overlap=True)
    x1, x2, y = _data(rank)
    torch.nn.functional.cross_entropy(m(x1.cuda(), x2.cuda()), y.cuda()).backward()
    torch.cuda.synchronize()
    q.put((rank, torch.cat([p.grad.flatten() for p in m.parameters()]).cpu().numpy()))
    dist.destroy_process_group()


def test_two_ranks_equal_mean_of_micro_batches():
    from oracle import fcsiam_ref as R

    state = R.synth_state("diff", 3, 2, seed=9)
    want = (_grads_single(_data(0), state) + _grads_single(_data(1), state)) / 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, state)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        np.testing.assert_allclose(got[r], want.numpy(), rtol=1e-4, atol=1e-6 * float(want.abs().max()) + 1e-9)
    np.testing.assert_array_equal(got[0], got[1])


# ---------------------------------------------------------------------------------------------------------------------
# The RCCL path itself (backend "nccl": side-stream async all_reduce with ReduceOp.AVG, stage-1 wait ordering the
# optimizer after both collectives).  Needs two GPUs: one process per GPU, spawned before any GPU call in the child.
def _nccl_worker(rank, world, port, q, state, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from stcd_amd.ddp import FlatGradReducer, broadcast_parameters
    from stcd_amd.modules import SiamUnet_diff
    from stcd_amd.optim import FlatAdamW

    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    m = SiamUnet_diff(3, 2, dtype="fp32")
    m.load_state_dict(state)
    m.to(dev).train()
    m.set_dropout_p(0.0)
    broadcast_parameters(m)
    FlatGradReducer(m, overlap=overlap)
    opt = FlatAdamW(m, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
    x1, x2, y = _data(rank)
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(m(x1.to(dev), x2.to(dev)), y.to(dev)).backward()
    grads = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    opt.step()                                   # must see fully reduced gradients (stage-1 wait)
    torch.cuda.synchronize()
    q.put((rank, grads.cpu().numpy(), torch.cat([p.detach().flatten() for p in m.parameters()]).cpu().numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_over_rccl(overlap):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the RCCL branch of FlatGradReducer); run on a multi-GPU lease")
    from oracle import fcsiam_ref as R
    from stcd_amd.modules import SiamUnet_diff
    from stcd_amd.optim import FlatAdamW

    state = R.synth_state("diff", 3, 2, seed=9)
    want = (_grads_single(_data(0), state) + _grads_single(_data(1), state)) / 2
    # single-process AdamW step from the averaged gradient
    m = SiamUnet_diff(3, 2, dtype="fp32")
    m.load_state_dict(state)
    m.to("cuda:0").train()
    m.set_dropout_p(0.0)
    opt = FlatAdamW(m, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
    x1, x2, y = _data(0)
    torch.nn.functional.cross_entropy(m(x1.cuda(), x2.cuda()), y.cuda()).backward()
    off = 0
    for p in m.parameters():
        p.grad.copy_(want[off:off + p.numel()].view_as(p).cuda())
        off += p.numel()
    opt.step()
    want_params = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu().numpy()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q, state, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, g, w = q.get(timeout=180)
        got[r] = (g, w)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        np.testing.assert_allclose(got[r][0], want.numpy(), rtol=1e-4, atol=1e-6 * float(want.abs().max()) + 1e-9)
        np.testing.assert_allclose(got[r][1], want_params, rtol=1e-5, atol=2e-6)
    np.testing.assert_array_equal(got[0][1], got[1][1])


# ---------------------------------------------------------------------------------------------------------------------
# The same RCCL branch on ONE GPU: a one-rank "nccl" group.  ReduceOp.AVG over one rank is the identity, so gradients and the
# optimizer step must equal the plain single-process ones bit for bit -- while the side-stream wait, record_stream, the two
# asynchronous RCCL all_reduce calls per backward and the stage-1 wait all really execute (several steps, both families'
# stage layouts: FC-Siam has two gradient stages, SegCD one).
def _nccl_one_rank_worker(port, q, state, overlap, family):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    from stcd_amd.ddp import FlatGradReducer, broadcast_parameters
    from stcd_amd.optim import FlatAdamW

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {}
    for reduced in (False, True):
        m = _family_model(family, state).to(dev).train()
        if reduced:
            broadcast_parameters(m)
            red = FlatGradReducer(m, overlap=overlap, force=True)
            assert m.grad_stage_hook is not None and red.backend == "nccl"
        opt = FlatAdamW(m, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
        for step in range(3):
            x1, x2, y = _data(step)
            opt.zero_grad(set_to_none=True)
            o = m(x1.to(dev), x2.to(dev))
            o = o[-1] if isinstance(o, (list, tuple)) else o
            loss = torch.nn.functional.cross_entropy(o, y.to(dev)) if family == "diff" else o.square().mean()
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        out[reduced] = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu().numpy()
    q.put(out)
    dist.destroy_process_group()


def _family_model(family, state):
    if family == "diff":
        from stcd_amd.modules import SiamUnet_diff
        m = SiamUnet_diff(3, 2, dtype="fp32")
        m.load_state_dict(state)
        m.set_dropout_p(0.0)
        return m
    from stcd_amd.segcd import SegCD
    m = SegCD(encoder_name="resnet18", dtype="fp32")
    m.load_state_dict(state)
    return m


@pytest.mark.parametrize("family,overlap", [("diff", True), ("diff", False), ("segcd", True)])
def test_one_rank_rccl_group_runs_the_collective_path(family, overlap):
    from oracle import fcsiam_ref as R
    from oracle import segcd_ref as G
    state = R.synth_state("diff", 3, 2, seed=9) if family == "diff" else G.synth_state(3, 1, 9, encoder="resnet18")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = ctx.Process(target=_nccl_one_rank_worker, args=(port, q, state, overlap, family))
    p.start()
    out = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    np.testing.assert_array_equal(out[True], out[False])


@pytest.mark.parametrize("model,want_bucket", [("diff", "fp32"), ("snunet", "bf16")])
def test_bench_contract_under_torchrun_two_ranks_on_one_card(tmp_path, model, want_bucket):
    """The driver's multi-GPU launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W`) rehearsed with N = 2 on this one-GPU box (STCD_DDP_REHEARSAL=1: both
    ranks share cuda:0 and meet over gloo -- RCCL cannot host two ranks on one device): exactly ONE JSON line from rank 0, n_gpus 2,
    value = global pairs / max-over-ranks time, per-rank rates present, the gradient buckets really averaged (the ranks end with the
    same loss trajectory on different data shards only if the parameters stayed in lock-step: checked through `last_loss` finiteness
    and the collective backend field).  `--bucket-dtype auto` (round 4): SNUNet (48 MB of gradients) and ChangeFormer (164 MB) travel
    as bf16 buckets with fp32 accumulation, the FC-Siam family (5-6 MB) stays fp32 -- so the first multi-GPU run of the
    driver exercises the path section 8e describes."""
    import json
    import socket
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, STCD_DDP_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "4",
           "--no-cpu-baseline", "--no-roofline"] + (["--model", model, "--size", "64"] if model != "diff" else [])
    out = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["collective_backend"] == "gloo" and d["rccl_ranks"] == 0
    assert d["bucket_dtype"] == want_bucket
    assert d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert len(d["pairs_per_sec_per_rank"]) == 2 and all(v > 0 for v in d["pairs_per_sec_per_rank"])
    assert abs(d["value"] - 2 * 4 * 4 / (d["ms_per_step"] * 4 / 1e3)) <= 0.01 * d["value"]        # whole-job rate over the max time
    assert np.isfinite(d["config"]["last_loss"])
