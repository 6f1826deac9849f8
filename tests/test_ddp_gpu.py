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
