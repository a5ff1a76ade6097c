"""Data-parallel train step for real (GPU): two processes on the one available device, each running the native
`DenoiseTrainer` on its own clip with `world_size=2` — the reference's accelerate/DDP step (train.py:661-667: gradient
all-reduce(mean); train.py:856: loss gather) on the product's own DP path (`training._exchange_and_update` ->
`parallel.allreduce_flat_grads`: ONE all-reduce of the flat LoRA-gradient buffer whose tail slot carries the loss).

RCCL refuses two ranks on one device, so the process group is gloo (it stages CUDA tensors through the host); the code path
in the trainer is the one RCCL runs on a multi-GPU node.  Checked: the reduced gradient equals the single-process sum of the
two clips' gradients, both ranks apply the identical update (= single-process AdamW on the mean gradient), and every rank
reports the rank-mean loss.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build_trainer(world, pg=None):
    import parity_utils as pu
    from t2v_amd.training import DenoiseTrainer
    ounet, ovae, _ = pu.build_oracle(False, 4, 0.05)
    dunet, dvae = pu.build_native(ounet, ovae, False, 4)
    params = [p for p in dunet.parameters() if p.requires_grad]
    return DenoiseTrainer(dunet, dvae, params, lr=1e-3, world_size=world, process_group=pg)


def _batch(rank):
    from oracle.weights import synthetic_batch
    return {k: v.cuda() for k, v in synthetic_batch(4, 64, 64, seed=100 + rank, text_dim=64).items()}


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        probe = torch.ones(4, device="cuda")
        dist.all_reduce(probe)
        assert float(probe[0]) == world
    except Exception as e:   # noqa: BLE001
        torch.save(dict(unsupported=f"{type(e).__name__}: {e}"), os.path.join(out_dir, f"r{rank}.pt"))
        dist.destroy_process_group()
        return
    tr = _build_trainer(world)
    loss = tr.train_step(_batch(rank))
    torch.cuda.synchronize()
    torch.save(dict(loss=float(loss), flat_g=tr.opt.flat_g.cpu(), flat_p=tr.opt.flat_p.cpu()), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_dp_step_equals_single_process_two_clip_step(tmp_path):
    import socket
    import torch.multiprocessing as mp
    from conftest import relerr
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{r}.pt") for r in (0, 1))
    if "unsupported" in r0:
        pytest.skip(f"gloo cannot all-reduce device tensors in this build: {r0['unsupported']}")
    # single process: the same two clips, gradients summed in the flat buffer, AdamW on the mean (grad_scale = 1/2)
    tr = _build_trainer(1)
    tr.opt.zero_grad()
    l0 = tr._fwd_bwd(_batch(0))
    l1 = tr._fwd_bwd(_batch(1))           # no zero_grad in between: the flat buffer accumulates g0 + g1
    torch.cuda.synchronize()
    g_sum = tr.opt.flat_g.cpu().clone()
    p_before = tr.opt.flat_p.cpu().clone()
    tr.opt.step(grad_scale=0.5)
    torch.cuda.synchronize()
    assert float(g_sum.norm()) > 0
    assert relerr(r0["flat_g"], g_sum) < 1e-3 and torch.equal(r0["flat_g"], r1["flat_g"])     # fp32-atomic reduction order only
    assert torch.equal(r0["flat_p"], r1["flat_p"])                                           # replicas stay in lock-step
    upd_dp, upd_sp = r0["flat_p"] - p_before, tr.opt.flat_p.cpu() - p_before
    assert float(upd_sp.norm()) > 0 and relerr(upd_dp, upd_sp) < 1e-2      # AdamW on the mean gradient, same clip as single process
    mean = 0.5 * (float(l0) + float(l1))
    assert abs(r0["loss"] - mean) < 1e-4 * abs(mean) and abs(r1["loss"] - mean) < 1e-4 * abs(mean)


# ---- the reference's own loop under data parallelism: `loss.backward(); optimizer.step()` with FlatAdamW doing the exchange
def _loop_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    import parity_utils as pu
    from t2v_amd.training import DenoiseTrainer, FlatAdamW
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        probe = torch.ones(4, device="cuda")
        dist.all_reduce(probe)
    except Exception as e:   # noqa: BLE001
        torch.save(dict(unsupported=f"{type(e).__name__}: {e}"), os.path.join(out_dir, f"l{rank}.pt"))
        dist.destroy_process_group()
        return
    ounet, ovae, _ = pu.build_oracle(False, 4, 0.05)
    dunet, dvae = pu.build_native(ounet, ovae, False, 4)
    params = [p for p in dunet.parameters() if p.requires_grad]
    # train.py:661-667 without the DDP wrap of the UNet: the optimizer owns the exchange
    opt = FlatAdamW(params, lr=1e-3, model=dunet)                      # world size from the default process group
    assert opt.world == world
    helper = DenoiseTrainer.__new__(DenoiseTrainer)                    # only its loss_fn (= finetune_unet, train.py:720-836)
    helper.__dict__.update(unet=dunet, vae=dvae, text_encoder=None, _aux_stream=None, batch_passes=True, use_offset_noise=False,
                           offset_noise_strength=0.1, cache_latents=False)
    from t2v_amd.schedulers import DDPMScheduler
    helper.scheduler = DDPMScheduler()
    losses = []
    for _ in range(2):                                                 # two optimisation steps of the reference's loop
        loss = helper.loss_fn(_batch(rank))
        loss.backward()
        opt.note_loss(loss)
        opt.step()
        losses.append(float(opt.last_mean_loss))
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    torch.save(dict(losses=losses, flat_p=opt.flat_p.cpu()), os.path.join(out_dir, f"l{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_reference_style_loop_under_dp_keeps_replicas_in_lock_step(tmp_path):
    """`loss.backward(); optimizer.step(); optimizer.zero_grad(set_to_none=True)` on two ranks with FlatAdamW performing the
    flat all-reduce itself: replicas stay bit-equal, the logged loss is the rank mean, and the parameters equal what the
    trainer's own DP path (`DenoiseTrainer.train_step`, same clips) produces."""
    import socket
    import torch.multiprocessing as mp
    from conftest import relerr
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_loop_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"l{r}.pt") for r in (0, 1))
    if "unsupported" in r0:
        pytest.skip(f"gloo cannot all-reduce device tensors in this build: {r0['unsupported']}")
    assert torch.equal(r0["flat_p"], r1["flat_p"])
    assert r0["losses"] == r1["losses"] and all(l > 0 for l in r0["losses"])
    # single process reference: two clips accumulated, AdamW on the mean, twice
    tr = _build_trainer(1)
    p0 = tr.opt.flat_p.cpu().clone()
    for _ in range(2):
        tr.opt.zero_grad()
        tr._fwd_bwd(_batch(0))
        tr._fwd_bwd(_batch(1))
        tr.opt.step(grad_scale=0.5)
    torch.cuda.synchronize()
    upd_dp, upd_sp = r0["flat_p"] - p0, tr.opt.flat_p.cpu() - p0
    # AdamW's first steps are sign-like: coordinates at the fp32-atomic noise level may flip between the two reduction orders
    cos = float((upd_dp.double() * upd_sp.double()).sum() / (upd_dp.double().norm() * upd_sp.double().norm()))
    assert float(upd_sp.norm()) > 0 and relerr(upd_dp, upd_sp) < 0.15 and cos > 0.99, (relerr(upd_dp, upd_sp), cos)


def test_rccl_allreduce_of_the_flat_gradient_buffer_around_a_graph_replay():
    """backend="nccl" (= RCCL) at world size 1 on the real device: the collective the 8-GPU run issues every step, on the very
    buffer (`flat_g_full`, loss in the tail slot), before and after a HIP-graph replay of the step."""
    import socket
    import torch.distributed as dist
    from t2v_amd.parallel import allreduce_flat_grads
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        tr = _build_trainer(1)
        batch = _batch(0)
        tr.capture(batch, warmup=1)
        l0 = tr.replay_step()
        torch.cuda.synchronize()
        tr.opt.zero_grad()
        tr._graph.replay()
        g_before = tr.opt.flat_g.clone()
        scale, mean = allreduce_flat_grads(tr.opt.flat_g_full, 1, None, loss=tr._static_loss, tail=tr.opt.numel, always=True)
        torch.cuda.synchronize()
        assert scale == 1.0 and torch.equal(tr.opt.flat_g, g_before) and float(g_before.norm()) > 0
        assert abs(float(mean) - float(tr._static_loss)) < 1e-6 * abs(float(tr._static_loss))
        tr.opt.step(grad_scale=scale, refresh=False)
        l1 = tr.replay_step()                       # and the graph still replays after the collective touched its buffers
        torch.cuda.synchronize()
        assert torch.isfinite(l1) and float(l0) > 0
    finally:
        dist.destroy_process_group()


def test_bench_gpus_2_self_spawns_and_prints_one_line():
    """`python bench.py --gpus 2` without a rendezvous environment re-launches itself under torch.distributed.run (bench._self_spawn:
    what the driver's multi-GPU run relies on when it starts the script directly).  Two ranks share the one device of this box
    over gloo (RCCL refuses two ranks on one GPU): rank 0 must print ONE json line with n_gpus = 2 and the whole-job rate."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, T2V_BENCH_BACKEND="gloo", T2V_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "c1",
                          "--no-cpu-baseline", "--no-roofline", "--no-other-mode"], env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]       # whole-job rate: both ranks' clips per step time
