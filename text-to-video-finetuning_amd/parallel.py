"""Data parallelism for the train step: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference gets plain DDP from accelerate (`train.py:518-523,661-667`): bucketed all-reduce(mean) of every trainable
gradient plus an all_gather of the loss for logging (`train.py:856`).  Here the trainable (LoRA) gradients already
live in ONE flat fp32 buffer (training.FlatAdamW), so the whole exchange is a single all-reduce(SUM) of 29-58 M floats
(117-234 MB) per step — at xGMI ring rates ~1-3 ms against a >100 ms step, so it is issued once after backward; the
1/world scaling is folded into the fused AdamW kernel.  The loss scalar rides in the tail of the same buffer when
`with_loss=True`, replacing the reference's per-micro-step all_gather + sync.
Device-agnostic on purpose: the same code path is exercised on CPU with gloo (tests/test_dp_gloo.py).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun-style env (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def allreduce_flat_grads(flat_g, world, group=None, loss=None, tail=None, always=False):
    """SUM-all-reduce the flat gradient buffer; returns (grad_scale, mean_loss_or_None), grad_scale = 1/world being applied
    by the optimizer kernel.  `tail`: index of a spare slot INSIDE `flat_g` (behind the gradients) that carries the loss
    through the same collective — no concatenation copy, no second collective (the reference all_gathers the loss per
    micro-step, train.py:856).  Without `tail` the loss is appended to a copy of the buffer (helper form)."""
    if world <= 1 and not (always and dist.is_initialized()):      # `always`: run the collective at world size 1 too (tests)
        return 1.0, loss
    if loss is None:
        dist.all_reduce(flat_g, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / world, None
    if tail is not None:
        flat_g[tail] = loss.detach().to(flat_g.dtype)
        dist.all_reduce(flat_g, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / world, flat_g[tail] / world
    buf = torch.cat([flat_g, loss.detach().reshape(1).to(flat_g.dtype)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    flat_g.copy_(buf[:-1])
    return 1.0 / world, buf[-1] / world


def broadcast_params(flat_p, group=None, src=0):
    """Make every rank start from rank-0's trainable parameters (DDP does this at wrap time)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_p, src=src, group=group)
