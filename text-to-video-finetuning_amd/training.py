"""The denoising train step on MI355X: the device-side equivalent of `train.py:720-836` (`finetune_unet`) and
`train.py:848-879` (backward, global-norm clip, AdamW), plus the data-parallel gradient exchange the reference gets
from accelerate/DDP (`train.py:661-667`).

  * trainable tensors (LoRA factors, or any subset) are re-homed into ONE flat fp32 buffer; their `.grad`s are views of
    one flat gradient buffer, so autograd accumulates straight into it, the DP exchange is a single RCCL all-reduce of
    that buffer over xGMI, and clip + AdamW are two fused kernels (the reference builds one param group per tensor,
    train.py:221-234);
  * the two UNet passes of `train.py:814-834` are kept (loss = mse0 + mse1);
  * `capture()` records forward+backward of a step into a HIP graph (static shapes) and replays it, removing the
    ~10^4 Python-side launches from the critical path.
"""
import os

import torch

from . import native as nv
from .models.clip_text import encode as encode_text
from .models.vae import tensor_to_vae_latent
from .schedulers import DDPMScheduler


class _Mse(torch.autograd.Function):
    """F.mse_loss(pred.float(), target.float()) (train.py:827) with its gradient produced in the same pass."""

    @staticmethod
    def forward(ctx, pred, target):
        nv.require_cuda(pred, target)
        pred = pred.contiguous().float()
        target = target.contiguous().float()
        loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        nv.call("t2v_mse_fwd_bwd", pred.data_ptr(), target.data_ptr(), pred.numel(), loss.data_ptr(), dpred.data_ptr(),
                1.0, nv.stream())
        ctx.save_for_backward(dpred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None


def mse_loss(pred, target):
    return _Mse.apply(pred, target)


def lr_lambda(name="constant", num_warmup_steps=0, num_training_steps=None, base_lr=None, num_cycles=1, power=1.0, lr_end=1e-7):
    """Multiplier of the base learning rate at optimiser step `k` (0-based) — the schedules `diffusers.get_scheduler` builds for
    the reference's `lr_scheduler` / `lr_warmup_steps` / `max_train_steps` options (train.py:481,519,606-612): "constant"
    (the default), "constant_with_warmup", "linear", "cosine", "cosine_with_restarts" (hard restarts, `num_cycles`) and
    "polynomial" (decay from `base_lr` to `lr_end` with `power`; needs `base_lr`, the optimiser's initial rate).  diffusers is not
    installable here: the six formulas are restated from its optimization.py and pinned by closed-form tests
    (tests/test_host_logic.py)."""
    import math
    w, total = int(num_warmup_steps), num_training_steps
    names = ("constant", "constant_with_warmup", "linear", "cosine", "cosine_with_restarts", "polynomial")
    if name not in names:                                     # (checked when the schedule is BUILT, not mid-training)
        raise ValueError(f"unknown lr schedule {name!r} (supported: {', '.join(names)})")
    if name in ("linear", "cosine", "cosine_with_restarts", "polynomial") and not total:
        raise ValueError(f"lr schedule {name!r} needs num_training_steps")
    if name == "polynomial" and not (base_lr and base_lr > lr_end):
        raise ValueError(f"lr schedule 'polynomial' needs base_lr > lr_end ({lr_end}), got {base_lr!r}")

    def f(k):
        if name == "constant":
            return 1.0
        if k < w:
            return float(k) / float(max(1, w))
        if name == "constant_with_warmup":
            return 1.0
        if name == "polynomial":
            if k > total:
                return lr_end / base_lr
            pct = 1.0 - float(k - w) / float(total - w)
            return ((base_lr - lr_end) * pct ** power + lr_end) / base_lr
        prog = float(k - w) / float(max(1, total - w))
        if name == "linear":
            return max(0.0, 1.0 - prog)
        if name == "cosine":
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * min(1.0, prog))))
        if prog >= 1.0:                                       # cosine_with_restarts
            return 0.0
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(num_cycles) * prog) % 1.0))))
    return f


class FlatAdamW:
    """torch.optim.AdamW semantics (train.py:238-249: betas (0.9,0.999), wd 1e-2, eps 1e-8) on one flat buffer.

    Every trainable tensor is re-homed into `flat_p` (its `.grad` into `flat_g`).  If `model` is given, LoRA factors of
    cloneofsimo-style wrappers are stored in GEMM layout (lora_bank.py) and a bf16 shadow `flat_p16` is kept, refreshed by
    `refresh_bf16()` (one cast kernel per step)."""

    def __init__(self, params, lr=5e-6, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0, model=None,
                 process_group=None, world_size=None):
        """`process_group` / `world_size`: data parallelism under a plain `loss.backward(); optimizer.step()` loop (the
        reference's, train.py:848-879).  The LoRA factor gradients are written straight into the flat buffer by side kernels —
        they never pass through autograd hooks, so a DistributedDataParallel wrapper (what `accelerator.prepare(unet)` builds,
        train.py:661-667) would not reduce them.  With world_size > 1 `step()` therefore performs the exchange itself: ONE
        all-reduce(SUM) of the flat gradient buffer (whose tail slot carries the loss handed to `note_loss`), then clip + AdamW on
        the mean.  world_size=None: taken from the initialised default process group (1 if there is none)."""
        from . import lora_bank
        if world_size is None:
            import torch.distributed as dist
            world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.pg, self.world = process_group, int(world_size)
        self.last_mean_loss = None
        self.lr_schedule = None            # optional multiplier of `lr` per optimiser step: training.lr_lambda(...)
        self.steps_done = 0
        seen, plist = set(), []
        for p in params:
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                plist.append(p)
        if not plist:
            raise ValueError("FlatAdamW: no trainable parameters")
        dev = plist[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW runs on a ROCm device only")
        self.params = plist
        self.lr, self.betas, self.weight_decay, self.eps, self.max_grad_norm = lr, betas, weight_decay, eps, max_grad_norm
        if model is not None:
            from .models.leaves import assign_dropout_names
            assign_dropout_names(model)          # dropout seeds keyed by module name (models/leaves.py)
            for name, p in model.named_parameters():
                p.__dict__.setdefault("_t2v_pname", name)     # (layout fingerprint of state_dict)
        plans = lora_bank.plan(model)
        plist = lora_bank.reorder(plist, plans)
        self.params = plist
        sizes = []
        for p in plist:
            pl = plans.get(id(p))
            k = (pl[0].down_numel if pl[1] == "down" else pl[0].up_numel) if pl else p.numel()
            sizes.append((k + 7) // 8 * 8)                       # keep every slice 16-byte aligned in bf16
        n = sum(sizes)
        self.numel = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        # 8 spare floats behind the gradients: slot n carries the loss through the SAME all-reduce (train.py:856 gathers it
        # with a separate collective per micro-step)
        self.flat_g_full = torch.zeros(n + 8, dtype=torch.float32, device=dev)
        self.flat_g = self.flat_g_full[:n]
        self.flat_p16 = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._sumsq_ws = torch.zeros(2048, dtype=torch.float32, device=dev)
        off, offsets = 0, {}
        self._homes = []
        for p, k in zip(plist, sizes):
            pl = plans.get(id(p))
            if pl:
                view = lora_bank.param_view(self.flat_p[off:off + k], p, pl[0], pl[1])
                gview = lora_bank.param_view(self.flat_g[off:off + k], p, pl[0], pl[1])
            else:
                view = self.flat_p[off:off + p.numel()].view(p.shape)
                gview = self.flat_g[off:off + p.numel()].view(p.shape)
            view.copy_(p.detach().float())
            p.data = view
            p.grad = gview
            p.__dict__["_t2v_flat_grad"] = True          # opt-in marker: kernels may accumulate straight into `.grad` (functional)
            self._homes.append((p, view, gview.detach()))   # own alias of the gradient slot (p.grad's object can be re-pointed)
            offsets[id(p)] = off
            off += k
        lora_bank.attach(plans, self.flat_p16, self.flat_g, offsets)
        # merged-weight execution of the wrapped layers (lora_bank.MergePlan); T2V_LORA_MERGE=0 keeps the rank columns in
        # the layer launches (the round-1 path) for A/B runs
        self.merge = None
        if plans and os.environ.get("T2V_LORA_MERGE", "1") != "0":
            self.merge = lora_bank.MergePlan(plans, self.flat_p)
        # wrappers whose dropout is active (the reference's default train mode) keep the branch apart: its rank-wide terms ride
        # in the base layers' epilogues and read transposed factor copies (lora_bank.PrepPlan); T2V_LORA_EPI=0: separate passes
        self.prep = None
        if plans and os.environ.get("T2V_LORA_EPI", "1") != "0":
            self.prep = lora_bank.PrepPlan(plans, self.flat_p)
        self.refresh_bf16()

    def _ensure_homed(self):
        from . import lora_bank
        if not lora_bank.is_homed(self._homes):      # e.g. after the reference's save_pipe (`unet.cpu()` ... `.to(device)`)
            lora_bank.rehome(self._homes)

    def refresh_bf16(self):
        """Derive everything the layer launches read from the fp32 parameters: the bf16 factor copies (one cast kernel) and
        the merged weights W_eff = W + s U D of the wrapped layers (one merge kernel)."""
        self._ensure_homed()
        nv.call("t2v_cast_f32_to_bf16", self.flat_p.data_ptr(), self.flat_p16.data_ptr(), self.numel, nv.stream())
        if self.merge is not None:
            self.merge.run()
        if self.prep is not None:
            self.prep.run()

    def _layout_fingerprint(self):
        """Hash of what the flat buffers MEAN: every trainable tensor's qualified name (where a model was given), shape and slot
        size in bank order.  Two trainers with equal `numel` but another trainable set / lora_bank plan read different hashes."""
        import hashlib
        h = hashlib.sha256()
        for p, view, _ in self._homes:
            h.update(f"{p.__dict__.get('_t2v_pname', '?')}|{tuple(p.shape)}|{view.numel()}|{tuple(view.shape)};".encode())
        return h.hexdigest()[:32]

    def state_dict(self):
        """Optimiser state for checkpoint / resume: the AdamW moments in the flat layout of THIS trainer (same model, same trainable
        set, same lora_bank plan — `layout` fingerprints that), the device step counter, `steps_done` — the position of the LR
        schedule, which the reference restores through its scheduler's state (train.py:606-612) — and the dropout position (base
        seed + optimisation step of the host protocol, models/leaves.py), so that a resumed run does not redraw the masks of
        steps 0..k."""
        from .models import leaves
        return {"steps_done": int(self.steps_done), "numel": int(self.numel), "layout": self._layout_fingerprint(),
                "exp_avg": self.exp_avg.detach().clone(), "exp_avg_sq": self.exp_avg_sq.detach().clone(),
                "step_count": self.step_count.detach().clone(),
                "dropout": {"base": int(leaves._seed_state["base"]), "step": int(leaves._seed_state["step"]),
                            "fwd": int(leaves._seed_state["fwd"])}}

    def load_state_dict(self, sd):
        if int(sd["numel"]) != int(self.numel):
            raise RuntimeError(f"t2v_amd: optimiser state of {sd['numel']} elements does not fit this trainer ({self.numel})")
        if "layout" in sd and sd["layout"] != self._layout_fingerprint():
            raise RuntimeError("t2v_amd: optimiser state was saved for another trainable set / LoRA bank layout (same element count, "
                               "different tensors): its moments would be applied to the wrong parameters")
        self.steps_done = int(sd["steps_done"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count.copy_(sd["step_count"])
        if "dropout" in sd:
            from .models import leaves
            leaves._seed_state["base"], leaves._seed_state["step"] = int(sd["dropout"]["base"]), int(sd["dropout"]["step"])
            # (index of the root forward inside an accumulation window: a mid-window resume must not redraw micro-step 0's masks)
            leaves._seed_state["fwd"] = int(sd["dropout"].get("fwd", -1))

    def zero_grad(self, set_to_none=False):
        from .functional import clear_bwd_colsums, drop_pending_wgrads
        drop_pending_wgrads()              # leftovers of a backward pass that died half-way must not run against freed operands
        clear_bwd_colsums()
        self._ensure_homed()
        self.flat_g_full.zero_()

    def grad_norm(self):
        return self.sumsq.sqrt()

    def note_loss(self, loss):
        """Put this rank's loss into the tail slot of the flat gradient buffer: the next `step()` all-reduces it together with
        the gradients and leaves the rank mean in `last_mean_loss` (the reference gathers the loss with its own collective per
        micro-step, train.py:856)."""
        self.flat_g_full[self.numel] = loss.detach().to(self.flat_g_full.dtype)

    def step(self, grad_scale=1.0, refresh=True, exchange=True):
        """(all-reduce when world_size > 1) + clip + AdamW on the flat buffers (2 kernels).  `refresh`: re-derive the bf16 copies
        of the LoRA factors right away, so that a plain `loss.backward(); optimizer.step()` loop (the reference's) needs no extra
        call; DenoiseTrainer refreshes inside its captured step instead (and exchanges itself: exchange=False)."""
        from .functional import join_side_stream
        join_side_stream()                 # no-op after a normal backward (its end-of-backward callback already joined)
        self._ensure_homed()
        if exchange and self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.flat_g_full, op=dist.ReduceOp.SUM, group=self.pg)
            grad_scale = grad_scale / self.world
            self.last_mean_loss = self.flat_g_full[self.numel] / self.world
        s = nv.stream()
        self.sumsq.zero_()
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        if clip:
            nv.call("t2v_sumsq", self.flat_g.data_ptr(), self.numel, self.sumsq.data_ptr(), self._sumsq_ws.data_ptr(), s)
        lr = self.lr * (self.lr_schedule(self.steps_done) if self.lr_schedule is not None else 1.0)
        self.steps_done += 1
        from .functional import note_weights_changed
        note_weights_changed()
        nv.call("t2v_adamw", self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                self.exp_avg_sq.data_ptr(), self.numel, lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                self.sumsq.data_ptr() if clip else None, float(self.max_grad_norm or 0.0), float(grad_scale),
                self.step_count.data_ptr(), s)
        from .models.leaves import advance_dropout_step
        advance_dropout_step()
        if refresh:
            self.refresh_bf16()


class DenoiseTrainer:
    """One optimisation step = VAE encode -> noise/add_noise -> 2x UNet forward -> eps-MSE -> backward ->
    (RCCL all-reduce of the flat LoRA gradient) -> clip -> AdamW."""

    # capture state (class-level defaults: helpers that borrow `loss_fn` build the object without __init__, tests/test_dp_gpu.py)
    _inline_aux = False
    _pipe = None
    _cap_stream = None
    _aux_stream = None

    def __init__(self, unet, vae, params, lr=5e-6, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0,
                 scheduler=None, process_group=None, world_size=1, text_encoder=None, use_offset_noise=False,
                 offset_noise_strength=0.1, rescale_schedule=False, cache_latents=False, gradient_accumulation_steps=1,
                 lr_scheduler="constant", lr_warmup_steps=0, max_train_steps=None):
        self.unet, self.vae = unet, vae
        self.text_encoder = text_encoder           # CLIPTextModel (train.py:784-790): native forward (models/clip_text.py)
        self._aux_stream = None
        self.batch_passes = True                   # evaluate the two UNet passes of train.py:814 as one stacked forward
        self.use_offset_noise = use_offset_noise and not rescale_schedule      # train.py:750
        self.offset_noise_strength = offset_noise_strength
        self.cache_latents = cache_latents         # batches come from utils/latent_cache.py: 'pixel_values' already ARE latents
        self.scheduler = scheduler or DDPMScheduler()
        if rescale_schedule:
            self.scheduler.rescale_betas()         # train.py:689-690 (betas only; see schedulers.enforce_zero_terminal_snr)
        self.opt = FlatAdamW(params, lr, betas, weight_decay, eps, max_grad_norm, model=unet, world_size=1)   # (exchange: below)
        if lr_scheduler != "constant":      # train.py:606-612
            self.opt.lr_schedule = lr_lambda(lr_scheduler, lr_warmup_steps, max_train_steps, base_lr=lr)
        self.gas = max(1, int(gradient_accumulation_steps))      # train.py:481,519,848 (`accelerator.accumulate`)
        self._micro = 0
        self._window_loss = None
        self._flag_every = max(0, int(os.environ.get("T2V_FLAG_CHECK_EVERY", "64")))     # steps between device-flag reads (0: never)
        self._flag_steps = 0
        self._flag_pending = None                  # (event, pinned words) of the read enqueued at the previous check point
        self.pg, self.world = process_group, world_size
        self.rank = 0
        if world_size > 1:
            import torch.distributed as dist
            self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self._drop_epoch = None                    # device-side dropout epoch (ranks start 2^32 apart: different masks per rank)
        self._graph = None
        self._static = None
        self._pipe = None
        self._cap_stream = None
        self._inline_aux = False

    # ---- train.py:720-836
    def _text_trainable(self):
        return self.text_encoder is not None and any(p.requires_grad for p in self.text_encoder.parameters())

    def _prepare(self, batch, side_clip=True):
        """The part of a step that does not depend on the trainable parameters (train.py:720-800): frozen text encoder, VAE
        encode, noise, timesteps, add_noise.  -> dict(noisy, target, timesteps, ehs, ids).  `side_clip`: run the frozen CLIP
        tower on the auxiliary stream beside the VAE encode (joined by `_unet_loss`); False = everything on the current stream
        (the pipelined capture records this whole part as ONE single-stream graph that runs on the auxiliary stream)."""
        ehs, ids = batch.get("encoder_hidden_states"), batch.get("prompt_ids")
        if ids is not None and ids.dim() > 2:
            ids = ids[0]
        if self._text_trainable():
            ehs = None                                   # (train.py:763-790: encoded inside the autograd graph, `_unet_loss`)
        elif ehs is None:
            # train.py:784-790: frozen text encoder (no grad) — independent of the VAE encode
            if side_clip:
                aux = self._aux()
                aux.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(aux), torch.no_grad():
                    ehs = encode_text(self.text_encoder, ids)
            else:
                with torch.no_grad():
                    ehs = encode_text(self.text_encoder, ids)
        if "latents" in batch or self.cache_latents:     # cache_latents path (train.py:741-746)
            latents = batch["latents"] if "latents" in batch else batch["pixel_values"]
        else:
            latents = tensor_to_vae_latent(batch["pixel_values"], self.vae, batch.get("vae_eps"))
        noise = batch["noise"] if "noise" in batch else self.sample_noise(latents)
        bsz = latents.shape[0]
        if "timesteps" in batch:
            timesteps = batch["timesteps"]
        else:
            timesteps = torch.randint(0, self.scheduler.num_train_timesteps, (bsz,), device=latents.device).long()
        noisy = self.scheduler.add_noise(latents, noise, timesteps)
        if self.scheduler.prediction_type == "epsilon":
            target = noise
        elif self.scheduler.prediction_type == "v_prediction":
            target = self.scheduler.get_velocity(latents, noise, timesteps)
        else:
            raise ValueError(f"Unknown prediction type {self.scheduler.prediction_type}")
        return {"noisy": noisy, "target": target, "timesteps": timesteps, "ehs": ehs, "ids": ids}

    def _unet_loss(self, prep, join_aux=True):
        """train.py:800-836: the UNet passes and the eps-MSE on a prepared batch."""
        noisy, target, timesteps, ehs = prep["noisy"], prep["target"], prep["timesteps"], prep["ehs"]
        bsz = noisy.shape[0]
        text_trainable = self._text_trainable()
        if text_trainable and ehs is None:           # train.py:763-790: text encoder in the autograd graph (main stream)
            ehs = encode_text(self.text_encoder, prep["ids"])
        if join_aux and not self._inline_aux and self._aux_stream is not None:  # CLIP and/or the parameter refresh of _fwd_bwd ran beside the VAE encode
            torch.cuda.current_stream().wait_stream(self._aux_stream)
        video_length = noisy.shape[2]
        if text_trainable and video_length > 1:
            # train.py:805-828: pass 0 = whole clip with DETACHED text states; pass 1 = frame 1 only with the trainable
            # states ("train text information only on the spatial layers")
            pred = self.unet(noisy, timesteps, encoder_hidden_states=ehs.detach()).sample
            l0 = mse_loss(pred.float(), target.float())
            pred = self.unet(noisy[:, :, 1:2], timesteps, encoder_hidden_states=ehs).sample
            return l0 + mse_loss(pred.float(), target[:, :, 1:2].float())
        if video_length > 1 and self.batch_passes:
            # train.py:814-834 runs the UNet twice on the same (noisy, t, text) and sums the two MSEs (text not trainable).
            # The two passes are independent, so they are evaluated as ONE forward over the stacked pair — the same
            # arithmetic per pass (every statistic in the net is per sample), twice the rows per kernel launch.
            pred = self.unet(torch.cat([noisy, noisy], 0), torch.cat([timesteps, timesteps], 0),
                             encoder_hidden_states=torch.cat([ehs, ehs], 0)).sample
            return mse_loss(pred[:bsz].float(), target.float()) + mse_loss(pred[bsz:].float(), target.float())
        losses = []
        for i in range(2):                            # train.py:814: two passes, losses summed
            pred = self.unet(noisy, timesteps, encoder_hidden_states=ehs).sample
            losses.append(mse_loss(pred.float(), target.float()))
            if video_length == 1 and i == 0:
                break
        return losses[0] if len(losses) == 1 else losses[0] + losses[1]

    def loss_fn(self, batch):
        return self._unet_loss(self._prepare(batch))

    def sample_noise(self, latents):
        """train.py:349-358: eps ~ N(0,1), optionally plus a per-(b,c,f) offset."""
        noise = torch.randn_like(latents)
        if self.use_offset_noise:
            b, c, f = latents.shape[:3]
            noise = noise + self.offset_noise_strength * torch.randn(b, c, f, 1, 1, device=latents.device)
        return noise

    def _aux(self):
        if self._inline_aux:                       # single-stream capture: the "auxiliary" work stays on the launch stream
            return torch.cuda.current_stream()
        if self._aux_stream is None:
            self._aux_stream = torch.cuda.Stream()
        return self._aux_stream

    def _fwd_bwd(self, batch, prep=None):
        """Forward + backward of one step.  `prep`: the output of `_prepare` when that part has already run (pipelined replay:
        on the auxiliary stream, as its own graph) — then everything here stays on the current stream."""
        # Dropout epoch: seeds reach the kernels by value (host counter, models/leaves.py::_next_seed), which a captured graph
        # would freeze; the launches of this step therefore also carry the address of a device counter that the step bumps
        # first thing (one captured add), so every replay draws fresh masks while forward and backward of a step agree.
        if self._drop_epoch is None:
            self._drop_epoch = torch.full((1,), (int(self.rank) << 32) + 1, dtype=torch.int64, device=self.opt.flat_p.device)
        self._drop_epoch += 1
        nv.call("t2v_set_dropout_epoch", self._drop_epoch.data_ptr())
        try:
            if prep is None:
                # bf16 factor copies + merged weights W_eff for this step (one cast + one merge kernel, HBM-bound): on the
                # auxiliary stream, beside the MFMA-bound VAE encode; `_unet_loss` joins it before the UNet
                aux = self._aux()
                aux.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(aux):
                    self.opt.refresh_bf16()
                loss = self.loss_fn(batch)
            else:
                self.opt.refresh_bf16()
                loss = self._unet_loss(prep, join_aux=False)
            loss.backward()
            from .functional import join_side_stream
            join_side_stream()             # the last factor-gradient batch is enqueued before clip / AdamW / all-reduce
        finally:
            nv.call("t2v_set_dropout_epoch", None)
        return loss.detach()

    def _exchange_and_update(self, loss):
        """DP exchange (train.py:661-667,856): ONE all-reduce(SUM) of the flat gradient buffer over RCCL/xGMI whose tail slot
        carries this rank's loss; returns the rank-mean loss (what the reference logs after `accelerator.gather`).  With gradient
        accumulation the buffer holds the SUM over the window's micro-steps: the optimiser scales by 1/window (accelerate divides
        each micro-loss instead, train.py:848-861 — the same gradient)."""
        from .parallel import allreduce_flat_grads
        scale, mean_loss = allreduce_flat_grads(self.opt.flat_g_full, self.world, self.pg, loss=loss, tail=self.opt.numel)
        self.opt.step(grad_scale=scale / self.gas, refresh=False, exchange=False)     # (_fwd_bwd refreshes the bf16 copies)
        return mean_loss

    def _micro_step(self, run):
        """One micro-step of an accumulation window: gradients are zeroed at the window's start, the exchange + update happen
        at its end (in between the flat buffer just accumulates and the loss is returned as is)."""
        if self._micro == 0:
            self.opt.zero_grad()
            self._window_loss = None
        try:
            loss = run()
        except BaseException:
            # a failed forward / backward leaves a partial gradient in the flat buffer: the window is abandoned (the next call
            # starts a new one with zero_grad) instead of being applied at its end
            self._micro = 0
            raise
        self._micro += 1
        # the reference logs / gathers the MEAN loss of the window's micro-steps (accelerate divides each by the window length)
        # (a replayed graph hands back the SAME static tensor every time: the window's running sum must own its memory)
        if self.gas > 1:
            self._window_loss = loss.clone() if self._window_loss is None else self._window_loss + loss
        if self._micro < self.gas:
            return loss
        self._micro = 0
        mean = self._window_loss / self.gas if self.gas > 1 else loss
        self._window_loss = None
        out = self._exchange_and_update(mean)
        self._poll_device_flags()
        return out

    def _poll_device_flags(self):
        """Split-K give-up word (gemm_w8.hip: a reducer that timed out stores 0xdead and sums incomplete slabs) WITHOUT stalling
        the step pipeline: every `T2V_FLAG_CHECK_EVERY` optimiser steps the flag words of all GEMM workspaces are copied to pinned
        host memory behind the step just issued, and the copy enqueued at the PREVIOUS check point — long complete — is read.  A
        time-out therefore raises at most two intervals after the step it spoiled instead of staying silent until the next
        checkpoint (`state_dict` still reads the flags synchronously)."""
        if not self._flag_every:
            return
        self._flag_steps += 1
        if self._flag_steps % self._flag_every:
            return
        from .functional import gemm_workspace_flags_async, raise_on_gemm_flags
        prev, self._flag_pending = self._flag_pending, gemm_workspace_flags_async()
        if prev is not None:
            raise_on_gemm_flags(prev)

    def train_step(self, batch):
        return self._micro_step(lambda: self._fwd_bwd(batch))

    @staticmethod
    def check_device_flags():
        """Raise if a kernel of an earlier step flagged a device-side failure (split-K hand-off time-out).  Synchronises."""
        from .functional import check_gemm_workspaces
        check_gemm_workspaces()

    def state_dict(self):
        """Everything a resumed run needs beyond the parameters themselves (which the reference saves through its pipeline /
        LoRA files, train.py:911-935): the optimiser state (moments, step counter, LR-schedule position, dropout step) and the
        device-side dropout epoch of this trainer's captured / eager steps."""
        self.check_device_flags()          # a checkpoint is a natural sync point: no split-K hand-off gave up since the last one
        # the epoch is stored WITHOUT this rank's 2^32 offset: every rank of a resumed run loads the usual rank-0 file and must
        # keep drawing its own masks
        ep = None if self._drop_epoch is None else int(self._drop_epoch.item()) - (int(self.rank) << 32)
        sd = {"opt": self.opt.state_dict(), "drop_epoch": ep, "drop_epoch_rank_free": True, "micro": int(self._micro)}
        if self._micro > 0:
            # mid-window checkpoint (gradient accumulation): the partial gradient sum and the running window loss belong to the
            # state — without them a resumed run would apply a partial gradient scaled by 1/window
            sd["window_grad"] = self.opt.flat_g_full.detach().clone()
            sd["window_loss"] = None if self._window_loss is None else self._window_loss.detach().clone()
        return sd

    def load_state_dict(self, sd):
        self.opt.load_state_dict(sd["opt"])
        if sd.get("drop_epoch") is not None:
            if self._drop_epoch is None:
                self._drop_epoch = torch.zeros(1, dtype=torch.int64, device=self.opt.flat_p.device)
            ep = int(sd["drop_epoch"])
            if sd.get("drop_epoch_rank_free"):
                ep += int(self.rank) << 32                      # this rank's own mask stream
            self._drop_epoch.fill_(ep)                          # in place: a captured step reads this address
        self._micro = int(sd.get("micro", 0))
        self._window_loss = None
        if self._micro > 0:
            if sd.get("window_grad") is None:                   # (a file of an older build: the window cannot be continued)
                self._micro = 0
            else:
                self.opt._ensure_homed()
                self.opt.flat_g_full.copy_(sd["window_grad"])
                wl = sd.get("window_loss")
                self._window_loss = None if wl is None else wl.to(self.opt.flat_p.device).clone()
        self.opt.refresh_bf16()

    # ---- HIP-graph replay of forward+backward (static shapes)
    def capture(self, batch, warmup=2, pipelined=None):
        """Record the step as HIP graphs (static shapes).  Active dropout is captured too: the seeds frozen into the graph are
        offset per replay by the device-side dropout epoch that `_fwd_bwd` bumps inside the captured region.

        Every form records SINGLE-STREAM graphs.  Why (profiles/r04_host_timeline.txt, r04_graph_launch_probe.txt): a graph with
        a forked branch inside (the eager step forks the frozen CLIP tower and the parameter refresh onto the auxiliary stream)
        takes the runtime's per-node launch path — 27 ms of host time per replay here, begun only once the previous replay has
        drained — while a single-stream graph is submitted as pre-built packets in under 1 ms without waiting, so the host runs
        steps ahead of the device (what the data-parallel exchange and a Python training loop around the step need).

        pipelined=False: ONE graph on ONE stream — the frozen CLIP tower and the parameter refresh, which the eager step forks onto
        the auxiliary stream, stay on the launch stream while the capture is open.  T2V_GRAPH_FORK=1 restores round 3's forked
        single graph.
        pipelined=True (the default since round 5; T2V_GRAPH_PIPELINE=0 / pipelined=False = the one-graph form): the step as TWO
        single-stream graphs, twice over —
          P_k  `_prepare`: frozen CLIP tower, VAE encode, noise, timesteps, add_noise -> prepared batch k     (k = 0, 1)
          U_k  `_fwd_bwd` on prepared batch k: parameter refresh, UNet forward + backward, factor gradients
        with step i replaying P_{i%2} on the auxiliary stream (as soon as U of step i-2 has released the slot, i.e. beside U of
        step i-1) and U_{i%2} on the launch stream.  The two kinds of graph may run concurrently and therefore record into
        separate memory pools and on separate streams (GEMM / GroupNorm scratch is per stream); the two copies of one kind are
        serialised by stream order.  Every P graph records into a pool of its OWN: round 4 measured the form with both P graphs
        in one pool, where slot 1's prepared batch sat on blocks the other P graph uses for intermediates (P of step 4 wrote over
        the batch U of step 3 was reading: 3e-3 on that step, 5 runs of 5); with separate pools the host-batch and device-batch
        runs of tests/test_train_gpu.py::test_pipelined_replays_without_host_sync_follow_the_eager_trajectory replay identical
        losses (profiles/r05_pytest_first_run.log).  A second lesson from the same test: a replay right behind an asynchronous
        host-to-device copy started before the copy had landed (host inputs go up through a temporary + copy kernel).
        Same box, ms per step (profiles/r04_capture_modes.txt): forked 80.4 (host 74 per step), one stream 81.4 (host 1), pipelined
        80.5 (host 1; the small CLIP kernels slip in beside the UNet's — the large kernels of two streams do not overlap, each
        fills the CUs: sum of kernel durations = step time in the trace)."""
        if pipelined is None:
            pipelined = os.environ.get("T2V_GRAPH_PIPELINE", "1") == "1" and os.environ.get("T2V_GRAPH_FORK", "0") != "1"
        self._inline_aux = not pipelined and os.environ.get("T2V_GRAPH_FORK", "0") != "1"
        try:
            return self._capture(batch, warmup, pipelined)
        finally:
            self._inline_aux = False

    def _capture(self, batch, warmup, pipelined):
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream()
        # eager warm-up on the streams the graphs are recorded on: every lazily created per-stream buffer (GEMM / GroupNorm scratch,
        # pinned staging reserve, merge tables of this dropout mode, tile-table look-ups) exists before a capture is open
        side = self._cap_stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.opt.zero_grad()
                self._fwd_bwd(static)
        torch.cuda.current_stream().wait_stream(side)
        if pipelined:
            aux = self._aux()
            aux.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(aux):
                self._prepare(static, side_clip=False)
            torch.cuda.current_stream().wait_stream(aux)
        torch.cuda.synchronize()
        self._replays = 0
        self._pipe = None
        mods = [self.unet, self.vae] + ([self.text_encoder] if self.text_encoder is not None else [])
        # frozen parameters whose cached copies the graph reads: GEMM-layout bf16 copies (`_t2v_prep`) AND the fp32 copies of bf16
        # bias / LayerNorm vectors (`_t2v_f32`: CLIP and VAE towers kept in bf16), plus the fused q/k/v copies of the CLIP tower,
        # which live in the attention modules' __dict__, not in parameters()
        from .models import clip_text
        self._frozen = [p for m in mods if m is not None for p in m.parameters()
                        if not p.requires_grad and ("_t2v_prep" in p.__dict__ or "_t2v_f32" in p.__dict__)]
        self._frozen += clip_text.fused_params(self.text_encoder, refresh=False)
        if not pipelined:
            self.opt.zero_grad()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._cap_stream):
                self._static_loss = self._fwd_bwd(static)
            self._graph, self._static = g, static
            return self
        aux = self._aux()
        pipe, unet_pool = [], None
        for k in range(2):
            st = static if k == 0 else {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in static.items()}
            g_pre = torch.cuda.CUDAGraph()
            # every P graph records into a pool of its OWN: P of step i+1 (the other slot) runs beside U of step i, which reads this
            # slot's prepared batch — in a shared pool that batch may sit on memory the other P graph uses for its intermediates
            # (blocks freed at the end of the first capture are handed out again in the second).  The U graphs are serialised on
            # the launch stream and share theirs.
            with torch.cuda.graph(g_pre, stream=aux):
                prep = self._prepare(st, side_clip=False)
            self.opt.zero_grad()
            g_unet = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_unet, pool=unet_pool, stream=self._cap_stream):
                loss = self._fwd_bwd(st, prep=prep)
            unet_pool = g_unet.pool()
            pipe.append({"static": st, "pre": g_pre, "unet": g_unet, "prep": prep, "loss": loss,
                         "ev_pre": torch.cuda.Event(), "ev_free": None})
        self._pipe = pipe
        self._graph, self._static, self._static_loss = pipe[0]["unet"], pipe[0]["static"], pipe[0]["loss"]
        return self

    def replay_step(self, batch=None):
        """One optimisation step from the recorded graphs.  `batch`: new inputs for the static buffers (device tensors are
        copied behind everything already queued on the current stream; host tensors — pinned, from a loader — go up on the
        auxiliary stream without waiting for the running step); None = the captured batch again."""
        if self._graph is None:
            raise RuntimeError("call capture() first")
        if getattr(self.opt, "merge", None) is not None:
            self.opt.merge.sync_base()             # base weights re-loaded since the last step? (masters refresh in place)
        from .functional import resync_prepared
        from .models import clip_text
        clip_text.fused_params(self.text_encoder)  # re-loaded q/k/v weights of the CLIP tower -> its fused copy (in place)
        resynced = resync_prepared(self._frozen)   # ... and the cached bf16 / fp32 copies of the unwrapped frozen layers
        if self._pipe is None:
            if batch is not None:
                for k, v in batch.items():
                    if torch.is_tensor(v):
                        self._static[k].copy_(v)

            def run():
                self._graph.replay()
                return self._static_loss
            return self._micro_step(run)
        slot = self._pipe[self._replays % 2]
        self._replays += 1
        cur, aux = torch.cuda.current_stream(), self._aux()
        if slot["ev_free"] is not None:
            aux.wait_event(slot["ev_free"])        # U of two steps ago has read this slot's prepared batch
        else:
            aux.wait_stream(cur)                   # first use: behind the capture / whatever produced the static inputs
        if resynced or (batch is not None and any(torch.is_tensor(v) and v.is_cuda for v in batch.values())):
            aux.wait_stream(cur)                   # device inputs / refreshed frozen copies: produced on the caller's stream
        with torch.cuda.stream(aux):
            if batch is not None:
                for k, v in batch.items():
                    if torch.is_tensor(v):
                        if not v.is_cuda:
                            # host input: upload into a temporary, then a device-side copy into the static buffer.  A graph
                            # replayed right behind an asynchronous host-to-device copy on the same stream started before the
                            # copy had landed (packet-path launch; tests/test_train_gpu.py::test_pipelined_replays..[True]
                            # read the previous contents) — the copy KERNEL is ordered like every other dispatch.
                            v = v.to(slot["static"][k].device, non_blocking=True)
                        slot["static"][k].copy_(v)
            slot["pre"].replay()
            slot["ev_pre"].record(aux)

        def run():
            cur.wait_event(slot["ev_pre"])
            slot["unet"].replay()
            if slot["ev_free"] is None:
                slot["ev_free"] = torch.cuda.Event()
            slot["ev_free"].record(cur)
            return slot["loss"]
        return self._micro_step(run)
