"""Data-parallel training over the 8 GPUs of one node: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm, over xGMI), image pairs sharded across ranks, ONE flat-bucket gradient all-reduce per backward stage.

The reference has no distributed code (nn.DataParallel pinned to one device,
/root/reference/train_pse_cd.py:405-417), so this is new capability judged against the oracle: N ranks must equal
the average of N sequential micro-batches on one device (BatchNorm statistics stay per replica, like the
reference's DataParallel).

The engine finishes the decoder's gradients first (backward stage 0): their all-reduce is launched on a side
stream while the encoder's backward (stage 1) still runs; the encoder's bucket follows, and the compute stream
waits for both before ``loss.backward()`` returns control to the optimizer.  Two collectives per step
(diff: 3.5 MB + 1.9 MB fp32) -- latency-bound on xGMI either way, so they are not split further.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None):
    """Rendezvous from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Returns (rank, local_rank, world_size).  A single process needs no env and no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # STCD_DDP_REHEARSAL=1: a multi-rank launch on a box with FEWER GPUs than ranks (the one-GPU development boxes): the ranks share
    # the cards round-robin and meet over gloo (RCCL cannot put two ranks on one device).  Everything above the transport -- the
    # launch contract, barriers, the reducer's hook protocol, rank-0 reporting -- runs as it will on a real node.
    rehearsal = os.environ.get("STCD_DDP_REHEARSAL") == "1"
    if torch.cuda.is_available():
        if rehearsal:
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if (torch.cuda.is_available() and not rehearsal) else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, even split of a global batch / index list (remainder to the first ranks)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class FlatGradReducer:
    """Averages gradients across ranks through the model's ``grad_stage_hook`` protocol
    (stcd_amd.modules.HipChangeDetector): hook(stage, flat_gradient_slice) is called when that slice is final.

    RCCL path (backend "nccl", GPU tensors): each bucket's all-reduce (ReduceOp.AVG, no extra scaling kernel) is issued
    asynchronously from a side stream that first waits for the compute stream, so the decoder bucket travels over xGMI
    while the encoder's backward kernels run; after the last bucket the compute stream waits on both collectives
    (stream-side wait, the host never blocks).  gloo path (CPU tests, or GPU tensors in a CPU-transport test):
    staged through host memory, synchronous."""

    def __init__(self, model, group=None, overlap: bool = True, force: bool = False, bucket_dtype: str = "fp32"):
        """force: install the hook even in a one-rank group (the collectives then average over one rank: an identity that still
        runs the whole RCCL path -- side stream, async all_reduce, final wait -- on a single GPU).
        bucket_dtype "bf16" (SNUNet's 48 MB of gradients): the buckets travel as bf16 with fp32 ACCUMULATION -- a direct
        reduce-scatter (all_to_all of bf16 shards: every rank sends shard j straight to rank j over its own xGMI link), an fp32 mean of
        the received shards, and an all_gather of the bf16 result: half the bytes of the fp32 all-reduce, one rounding of the inputs and
        one of the mean, no bf16 running sums."""
        if bucket_dtype not in ("fp32", "bf16"):
            raise ValueError("bucket_dtype must be 'fp32' or 'bf16'")
        self.bucket_dtype = bucket_dtype
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.overlap = overlap
        self._comm = None
        self._works = []
        if self.world > 1 or (force and dist.is_initialized()):
            model.grad_stage_hook = self._hook

    def _reduce_bf16(self, g: torch.Tensor):
        """g <- bf16(mean over ranks, in fp32, of bf16(g_rank)): direct reduce-scatter + all-gather over bf16 payloads."""
        w, n = self.world, g.numel()
        per = (n + w - 1) // w
        send = torch.zeros(per * w, dtype=torch.bfloat16, device=g.device)
        send[:n].copy_(g)
        if g.is_cuda and self.backend == "nccl":
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=self.group)               # shard j of every rank -> rank j
            shard = (recv.view(w, per).float().sum(0) / w).to(torch.bfloat16)
            out = torch.empty_like(send)
            dist.all_gather_into_tensor(out, shard, group=self.group)
        else:      # gloo (CPU-transport tests): no all_to_all there -- gather the bf16 buckets, same arithmetic per shard
            h = send.cpu() if send.is_cuda else send
            parts = [torch.empty_like(h) for _ in range(w)]
            dist.all_gather(parts, h, group=self.group)
            out = (torch.stack([p.float() for p in parts]).sum(0) / w).to(torch.bfloat16)
        g.copy_(out[:n].to(device=g.device, dtype=g.dtype))

    def _hook(self, stage: int, g: torch.Tensor):
        if g.numel() > 0 and self.bucket_dtype == "bf16":
            if g.is_cuda and self.backend == "nccl" and self.overlap:
                if self._comm is None:
                    self._comm = torch.cuda.Stream(device=g.device)
                self._comm.wait_stream(torch.cuda.current_stream(g.device))
                g.record_stream(self._comm)
                with torch.cuda.stream(self._comm):
                    self._reduce_bf16(g)
                if stage == 1:
                    torch.cuda.current_stream(g.device).wait_stream(self._comm)
            else:
                self._reduce_bf16(g)
            return
        if g.numel() > 0:
            if g.is_cuda and self.backend == "nccl":
                if self.overlap:
                    if self._comm is None:
                        self._comm = torch.cuda.Stream(device=g.device)
                    self._comm.wait_stream(torch.cuda.current_stream(g.device))
                    g.record_stream(self._comm)     # g may be a slice of a temporary (gradient accumulation): keep its
                    with torch.cuda.stream(self._comm):   # memory from being recycled while the collective still reads it
                        self._works.append(dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
                else:
                    dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group)
            else:
                t = g.cpu() if g.is_cuda else g
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                t.mul_(1.0 / self.world)
                if g.is_cuda:
                    g.copy_(t)
        if stage == 1:                          # last bucket: order the compute stream after every collective
            for w in self._works:
                w.wait()
            self._works.clear()


def broadcast_parameters(model, src: int = 0, group=None):
    """Same initial replica everywhere (parameters and BN buffers)."""
    if not dist.is_initialized():
        return
    staged = dist.get_backend(group) != "nccl"
    for t in list(model.parameters()) + list(model.buffers()):
        if staged and t.is_cuda:
            h = t.data.cpu()
            dist.broadcast(h, src=src, group=group)
            t.data.copy_(h)
        else:
            dist.broadcast(t.data, src=src, group=group)
