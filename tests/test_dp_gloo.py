"""Data-parallel gradient exchange on CPU: 2 processes, gloo.  The same `allreduce_flat_grads` runs over RCCL on GPUs."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = dict(block_out_channels=(64, 64, 64, 64), cross_attention_dim=64, attention_head_dim=64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model_and_grads(seeds):
    """LoRA-gradient of the summed two-pass eps-MSE on the CPU oracle for the given batch seeds (mean over seeds)."""
    from oracle.lora import inject_trainable_lora_extended
    from oracle.unet3d import UNet3DConditionModel
    from oracle.weights import randomize_lora_up
    torch.manual_seed(0)
    m = UNet3DConditionModel(**SMALL)
    m.requires_grad_(False)
    inject_trainable_lora_extended(m, {"Transformer2DModel"}, r=4)
    randomize_lora_up(m)
    m.eval()
    params = [p for p in m.parameters() if p.requires_grad]
    flat = torch.zeros(sum(p.numel() for p in params))
    losses = []
    for s in seeds:
        g = torch.Generator().manual_seed(s)
        x = torch.randn(1, 4, 2, 8, 8, generator=g); t = torch.randint(0, 1000, (1,), generator=g)
        ehs = torch.randn(1, 77, 64, generator=g); tgt = torch.randn(1, 4, 2, 8, 8, generator=g)
        loss = sum(torch.nn.functional.mse_loss(m(x, t, ehs).sample, tgt) for _ in range(2))
        grads = torch.autograd.grad(loss, params)
        flat += torch.cat([gr.flatten() for gr in grads]) / len(seeds)
        losses.append(loss.detach())
    return flat, torch.stack(losses).mean()


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import t2v_amd  # noqa: F401
    from t2v_amd.parallel import allreduce_flat_grads, broadcast_params, init_from_env
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    flat, loss = _model_and_grads([100 + rank])              # each rank: its own clip
    p = torch.full((5,), float(rank))
    broadcast_params(p)
    assert torch.equal(p, torch.zeros(5))
    scale, mean_loss = allreduce_flat_grads(flat, world, None, loss)
    if rank == 0:
        torch.save(dict(g=flat * scale, loss=mean_loss), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_exchange_equals_single_process_mean(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    old = torch.get_num_threads()
    torch.set_num_threads(2)                                  # same oneDNN kernel selection as the workers
    try:
        ref_g, ref_loss = _model_and_grads([100, 101])        # single process, both clips, mean
    finally:
        torch.set_num_threads(old)
    err = float((got["g"] - ref_g).norm() / ref_g.norm())
    assert err < 1e-3, err                                    # fp32 CPU conv algorithms differ slightly run to run
    assert torch.allclose(got["loss"], ref_loss, rtol=1e-4)


def test_world_size_one_is_identity():
    sys.path.insert(0, ROOT)
    from t2v_amd.parallel import allreduce_flat_grads
    g = torch.arange(4.0)
    scale, loss = allreduce_flat_grads(g, 1, None, torch.tensor(2.0))
    assert scale == 1.0 and loss.item() == 2.0 and torch.equal(g, torch.arange(4.0))
