"""LoRA bank: storage layout of the trainable LoRA factors inside the trainer's flat buffers.

The reference trains `lora_down` / `lora_up` of every wrapped layer (utils/lora.py:46-62,97-139,179-216).  Here every
factor is stored in the flat fp32 parameter buffer DIRECTLY in the layout the GEMM kernels consume
  down  [rp, taps, Cin_p]   (K ordered (tap, c); the Parameter is a permuted, sliced VIEW with the module's shape)
  up    [rp, Np]            (transposed: the backward `dt = dy U` reads it as a plain [N, K] weight and rides in the
                             base layer's backward-data launch; the forward `y += s t U^T` reads it K-major)
(rp / Np / Cin_p = sizes rounded up to 8, zero padded), so that
  * one cast kernel per step refreshes the bf16 copies of ALL factors (no per-layer permute/cast launches),
  * weight gradients are accumulated by the TN GEMM straight into the flat fp32 gradient buffer (no per-layer
    zeros / un-permute / AccumulateGrad launches), where clip + AdamW + the RCCL all-reduce operate.

Projection groups.  `to_q/to_k/to_v` of a self-attention (and `to_k/to_v` of the text cross-attention) read the same
input, so they are evaluated as ONE GEMM (models/leaves.py::Attention).  Their factors are stored adjacently:
  downs  [n*rp, Cin_p]          = the members' down slots back to back            (D_cat)
  ups    [n*rp, n*Np]           member i owns the row block [i*rp, (i+1)*rp) — its slot — and, inside it, the column block
                                [i*Np, (i+1)*Np); the rest of the slot is structurally zero (never receives a gradient),
                                so the whole region IS the block-diagonal up matrix the fused launches consume.
"""
import torch

from .functional import ceil8

MERGE_MAX_RANK = 32      # padded rank window of csrc/lora_merge.hip (RMAX)


class LoraEntry:
    __slots__ = ("r", "rp", "n", "npad", "cin", "cin_p", "taps", "down_off", "up_off", "down_numel", "up_numel",
                 "down_w16", "up_w16", "down_g", "up_g", "flat_ptr", "group", "gidx", "weff_fwd", "weff_bwd", "merge_scale",
                 "up_t16", "down_t16", "rk", "prep_scale")


class LoraGroup:
    """Projections that share their input (see module docstring).  `n` members of equal (rp, Np, Cin_p)."""
    __slots__ = ("entries", "mods", "owner", "n", "rp_each", "rp", "npad_each", "npad", "cin_p", "down_w16", "up_w16", "down_g",
                 "up_g", "weff_fwd", "weff_bwd", "merge_scale", "up_t16", "down_t16", "rk", "prep_scale")


def _dissolve(g):
    for e in g.entries:
        e.group, e.gidx = None, 0
        e.up_numel = e.npad * e.rp
    g.owner.__dict__.pop("_t2v_group", None)


def _find_groups(model, plans):
    """[(attention_module, [wrapper modules])]: q,k,v (same input width) or k,v of every module exposing to_q/to_k/to_v."""
    out = []
    for mod in model.modules():
        names = ("to_q", "to_k", "to_v")
        if not all(hasattr(mod, a) for a in names):
            continue
        ws = [getattr(mod, a) for a in names]
        ents = []
        for w in ws:
            parts = _wrapper_parts(w)
            pl = plans.get(id(parts[1].weight)) if parts is not None else None
            ents.append(pl[0] if pl is not None and pl[2] is w else None)
        if any(e is None or e.taps != 1 for e in ents):
            continue
        if any(_wrapper_parts(w)[0].bias is not None for w in ws):
            continue
        same = lambda a, b: (a.rp, a.npad, a.cin_p, a.r) == (b.rp, b.npad, b.cin_p, b.r)
        if same(ents[0], ents[1]) and same(ents[1], ents[2]):
            out.append((mod, ws))
        elif same(ents[1], ents[2]):
            out.append((mod, ws[1:]))
    return out


def _wrapper_parts(mod):
    base = getattr(mod, "linear", None)
    if base is None:
        base = getattr(mod, "conv", None)
    if base is None or not hasattr(mod, "lora_down") or not hasattr(mod, "lora_up"):
        return None
    return base, mod.lora_down, mod.lora_up


def plan(model, fuse_groups=True):
    """Map id(param) -> (entry, 'down'|'up', wrapper) for every cloneofsimo-style wrapper in `model`."""
    plans = {}
    if model is None:
        return plans
    for mod in model.modules():
        mod.__dict__.pop("_t2v_group", None)
    for mod in model.modules():
        parts = _wrapper_parts(mod)
        if parts is None:
            continue
        base, down, up = parts
        wd, wu = down.weight, up.weight
        if not (wd.requires_grad and wu.requires_grad):
            continue
        e = LoraEntry()
        e.r = wd.shape[0]
        e.rp = ceil8(e.r)
        e.n, e.npad = wu.shape[0], ceil8(wu.shape[0])
        e.cin, e.cin_p = wd.shape[1], ceil8(wd.shape[1])
        e.taps = 1
        for s in wd.shape[2:]:
            e.taps *= s
        if wd.dim() == 5 and (wd.shape[3] != 1 or wd.shape[4] != 1):
            continue    # only the (k,1,1) temporal conv is a supported 3-D window
        e.down_numel = e.rp * e.taps * e.cin_p
        e.up_numel = e.npad * e.rp
        e.group, e.gidx = None, 0
        plans[id(wd)] = (e, "down", mod)
        plans[id(wu)] = (e, "up", mod)
    if fuse_groups:
        for attn, ws in _find_groups(model, plans):
            g = LoraGroup()
            g.mods, g.owner = ws, attn
            g.entries = [plans[id(w.lora_down.weight)][0] for w in ws]
            g.n = len(ws)
            e0 = g.entries[0]
            g.rp_each, g.rp, g.npad_each, g.npad, g.cin_p = e0.rp, e0.rp * g.n, e0.npad, e0.npad * g.n, e0.cin_p
            if g.rp > 96 or e0.rp > 32:          # (the streaming factor-gradient kernel takes padded ranks up to 32: wider members
                continue                          #  run one by one, functional._lora_side_grads has the K-major GEMM fallback)
            for i, e in enumerate(g.entries):
                e.group, e.gidx = g, i
                e.up_numel = e.rp * g.npad            # the member's slot is a full row block of the group's up matrix
            attn._t2v_group = g
    return plans


def reorder(plist, plans):
    """Parameter order of the flat buffer: group members' downs back to back, then their ups back to back (member order)."""
    by_id = {id(p): p for p in plist}
    emitted, out = set(), []
    for p in plist:
        if id(p) in emitted:
            continue
        pl = plans.get(id(p))
        g = pl[0].group if pl else None
        if g is None:
            out.append(p)
            emitted.add(id(p))
            continue
        members = [w.lora_down.weight for w in g.mods] + [w.lora_up.weight for w in g.mods]
        if not all(id(m) in by_id for m in members):      # partially trained group: evaluate its members one by one
            _dissolve(g)
            out.append(p)
            emitted.add(id(p))
            continue
        for m in members:
            out.append(m)
            emitted.add(id(m))
    return out


def param_view(flat_slice, p, entry, role):
    """The tensor (shape == p.shape) through which the Parameter sees its slice of the flat buffer."""
    if role == "down":
        s3 = flat_slice.view(entry.rp, entry.taps, entry.cin_p)[: entry.r, :, : entry.cin]
        if p.dim() == 2:
            return s3[:, 0, :]
        if p.dim() == 4:
            return s3.reshape(entry.r, p.shape[2], p.shape[3], entry.cin).permute(0, 3, 1, 2) \
                if entry.cin == entry.cin_p else \
                flat_slice.view(entry.rp, p.shape[2], p.shape[3], entry.cin_p)[: entry.r, :, :, : entry.cin].permute(0, 3, 1, 2)
        v = s3.permute(0, 2, 1)
        return v[:, :, :, None, None]
    if entry.group is not None:
        c0 = entry.gidx * entry.npad
        s2 = flat_slice.view(entry.rp, entry.group.npad)[: entry.r, c0: c0 + entry.n].t()
    else:
        s2 = flat_slice.view(entry.rp, entry.npad)[: entry.r, : entry.n].t()
    for _ in range(p.dim() - 2):
        s2 = s2.unsqueeze(-1)
    return s2


def attach(plans, flat_p16, flat_g, offsets):
    """Create the prepared-layout views and hang the entry on its wrapper module (`mod._t2v_bank`)."""
    done = set()
    for pid, (e, role, mod) in plans.items():
        off = offsets.get(pid)
        if off is None:
            continue
        if role == "down":
            e.down_off = off
            e.down_w16 = flat_p16[off: off + e.down_numel].view(e.rp, e.taps * e.cin_p)
            e.down_g = flat_g[off: off + e.down_numel].view(e.rp, e.taps * e.cin_p)
        else:
            e.up_off = off
            if e.group is not None:     # column block of the member's row-block slot (leading dimension = group width)
                c0 = e.gidx * e.npad
                e.up_w16 = flat_p16[off: off + e.up_numel].view(e.rp, e.group.npad)[:, c0: c0 + e.npad]
                e.up_g = flat_g[off: off + e.up_numel].view(e.rp, e.group.npad)[:, c0: c0 + e.npad]
            else:
                e.up_w16 = flat_p16[off: off + e.up_numel].view(e.rp, e.npad)
                e.up_g = flat_g[off: off + e.up_numel].view(e.rp, e.npad)
        done.add(id(mod))
    groups = {}
    for pid, (e, role, mod) in plans.items():
        if all(hasattr(e, a) for a in ("down_w16", "up_w16")):
            e.flat_ptr = flat_g.data_ptr()
            mod._t2v_bank = e
            if e.group is not None:
                groups[id(e.group)] = e.group
    for g in groups.values():
        es = g.entries
        ok = all(hasattr(e, "down_off") and hasattr(e, "up_off") for e in es)
        for a, b in zip(es, es[1:]):
            ok = ok and b.down_off == a.down_off + a.down_numel and b.up_off == a.up_off + a.up_numel
        if not ok:
            raise RuntimeError("t2v_amd: fused projection group is not contiguous in the flat buffer")
        d0, u0 = es[0].down_off, es[0].up_off
        g.down_w16 = flat_p16[d0: d0 + g.rp * g.cin_p].view(g.rp, g.cin_p)
        g.down_g = flat_g[d0: d0 + g.rp * g.cin_p].view(g.rp, g.cin_p)
        g.up_w16 = flat_p16[u0: u0 + g.rp * g.npad].view(g.rp, g.npad)
        g.up_g = flat_g[u0: u0 + g.rp * g.npad].view(g.rp, g.npad)


class MergePlan:
    """Merged weights of every wrapped layer: W_eff = W + s U D, refreshed by ONE `t2v_lora_merge` launch per step.

    With dropout off and the identity selector, `base(x) + scale*up(down(x))` (utils/lora.py:57-62,134-139,211-216) is
    `x (*) W_eff^T`: each wrapped layer then runs as a plain N = C_out implicit GEMM forward and backward-data — no rank
    columns in the tile grid (N = C+16 knocks out the wide tiles), no rank-update passes over y / dx.  The factor gradients
    keep their own formulas (dU = s t^T dy, dD = s dt^T x) and are computed on the side stream.  Per layer this holds
      w32       fp32 master of the frozen base weight in forward GEMM layout [Np, taps*Cin_p]   (read-only)
      weff_fwd  bf16 [Np, taps*Cin_p]            weff_bwd  bf16 [Cin_p, taps_flipped*Np]
    (projection groups: the members' W_eff are row / column blocks of one group buffer).  The device job table is built once;
    it stays valid because every buffer it points to (flat fp32 parameters, masters, outputs) is allocated once."""

    def __init__(self, plans, flat_p):
        import ctypes as C

        from . import native as nv
        from .functional import _prep_compute
        dev = flat_p.device
        BF16 = torch.bfloat16
        seen, entries = set(), []
        for pid, (e, role, mod) in plans.items():
            if id(e) in seen or not all(hasattr(e, a) for a in ("down_off", "up_off")):
                continue
            seen.add(id(e))
            if e.rp > MERGE_MAX_RANK:
                # beyond the merge kernel's rank window: the layer keeps its LoRA branch apart (leaves.run_layer falls back to
                # functional.lora_layer / the GEMM composition because `merge_scale` stays unset) — any rank the reference accepts
                continue
            entries.append((e, mod))
        self.entries = entries
        self._keep = []
        self._base_tags = []
        jobs = (nv.LoraMergeJob * max(1, len(entries)))()
        groups_done = {}
        for k, (e, mod) in enumerate(entries):
            base = _wrapper_parts(mod)[0]
            w32 = _prep_compute(base.weight, "fwd32", None)
            K = e.taps * e.cin_p
            if tuple(w32.shape) != (e.npad, K):
                raise RuntimeError(f"t2v_amd: LoRA merge plan: prepared base weight {tuple(w32.shape)} does not match the bank "
                                   f"entry ({e.npad}, {K})")
            g = e.group
            if g is not None:
                if id(g) not in groups_done:
                    g.weff_fwd = torch.empty(g.npad, g.cin_p, dtype=BF16, device=dev)
                    g.weff_bwd = torch.empty(g.cin_p, g.npad, dtype=BF16, device=dev)
                    g.merge_scale = float(g.mods[0].scale)
                    groups_done[id(g)] = g
                r0 = e.gidx * e.npad
                e.weff_fwd = g.weff_fwd[r0: r0 + e.npad]
                e.weff_bwd = g.weff_bwd[:, r0: r0 + e.npad]
                up_ptr = flat_p.data_ptr() + 4 * (e.up_off + r0)
                ldu = g.npad
            else:
                e.weff_fwd = torch.empty(e.npad, K, dtype=BF16, device=dev)
                e.weff_bwd = torch.empty(e.cin_p, e.taps * e.npad, dtype=BF16, device=dev)
                up_ptr = flat_p.data_ptr() + 4 * e.up_off
                ldu = e.npad
            e.merge_scale = float(mod.scale)
            j = jobs[k]
            j.w32, j.up, j.ldu, j.down = w32.data_ptr(), up_ptr, ldu, flat_p.data_ptr() + 4 * e.down_off
            j.wf, j.ldwf = e.weff_fwd.data_ptr(), e.weff_fwd.stride(0)
            j.wb, j.ldwb = e.weff_bwd.data_ptr(), e.weff_bwd.stride(0)
            j.Np, j.Cp, j.taps, j.rp, j.scale = e.npad, e.cin_p, e.taps, e.rp, e.merge_scale
            self._keep.append(w32)
            self._base_tags.append(self._tag(base.weight))
        self.njobs = len(entries)
        self.ntiles = 0
        self._jobs = jobs
        self._tables = {}        # subset of entries (tuple of indices, None = all) -> (jobs_dev, tile_job_dev, njobs, ntiles)
        if self.njobs:
            self.jobs_dev, self.tile_job_dev, _, self.ntiles = self._table(None)
        self.bytes = sum(w.numel() * 4 for w in self._keep) + sum(e.weff_fwd.numel() * 4 for e, _ in entries)

    def _table(self, subset):
        """Device job table + tile list of the entries `subset` (tuple of indices; None = every entry), built once per subset."""
        tab = self._tables.get(subset)
        if tab is not None:
            return tab
        import ctypes as C

        from . import native as nv
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("t2v_amd: the LoRA merge table for this set of dropping wrappers has not been built yet; run one "
                               "eager step in this mode before capturing (DenoiseTrainer.capture does)")
        idx = range(self.njobs) if subset is None else subset
        jobs = (nv.LoraMergeJob * max(1, len(idx)))()
        for q, k in enumerate(idx):
            C.memmove(C.byref(jobs[q]), C.byref(self._jobs[k]), C.sizeof(nv.LoraMergeJob))
        lib = nv.lib()
        n = len(idx)
        total = lib.t2v_lora_merge_plan(jobs, n, None, 0)
        if total <= 0:
            nv.check(int(total) if total < 0 else -1, "t2v_lora_merge_plan")
        tile_job = (C.c_int * total)()
        if lib.t2v_lora_merge_plan(jobs, n, tile_job, total) != total:
            nv.check(-1, "t2v_lora_merge_plan")
        dev = self._keep[0].device
        tab = (torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev),
               torch.frombuffer(bytearray(bytes(tile_job)), dtype=torch.int32).to(dev), n, int(total))
        self._tables[subset] = tab
        return tab

    @staticmethod
    def _tag(w):
        return (w.data_ptr(), w._version, w.device)

    def sync_base(self):
        """The fp32 masters are snapshots of the FROZEN base weights: after a `load_state_dict` / resume / `.to()` that changes
        a base weight (new version counter or storage), re-derive that layer's master IN PLACE (the device job table keeps
        pointing at the same buffers).  Also re-checks every wrapper's `scale` against the value baked into the job table —
        a changed scale needs a new plan, and is refused rather than trained on silently.  Returns the number of refreshed
        masters.  Host-side only (a few hundred attribute reads): the trainers call it once per step before the launch."""
        from .functional import _prep_compute
        n = 0
        for k, (e, mod) in enumerate(self.entries):
            w = _wrapper_parts(mod)[0].weight
            tag = self._tag(w)
            if tag != self._base_tags[k]:
                self._keep[k].copy_(_prep_compute(w, "fwd32", None))
                self._base_tags[k] = tag
                n += 1
            if getattr(self, "_current", None) and e.merge_scale is not None and float(mod.scale) != e.merge_scale:
                raise RuntimeError("t2v_amd: a LoRA wrapper's scale changed after the merge plan was built "
                                   f"({e.merge_scale} -> {float(mod.scale)}); rebuild the optimiser / trainer")
        return n

    @staticmethod
    def _drops(mod):
        d = getattr(mod, "dropout", None)
        return isinstance(d, torch.nn.Dropout) and d.training and d.p > 0.0

    def wanted(self):
        """Indices of the entries whose wrapper can take the merged path right now: wrappers whose dropout is active (the
        reference's default train mode, utils/lora.py:35,89) evaluate the branch apart and never read W_eff — refreshing theirs
        is HBM traffic for nothing (all 574 layers: 15 GB per step; in the default mode only the Conv3d wrappers, whose
        dropout_p is 0, stay merged).  A projection group is refreshed as a whole or not at all."""
        ok = [not self._drops(mod) for _, mod in self.entries]
        if not all(ok):
            if getattr(self, "_members", None) is None:
                self._members = {}
                for k, (e, _) in enumerate(self.entries):
                    if e.group is not None:
                        self._members.setdefault(id(e.group), []).append(k)
            for idx in self._members.values():
                if not all(ok[k] for k in idx):
                    for k in idx:
                        ok[k] = False
        return tuple(k for k, v in enumerate(ok) if v)

    def _mark(self, current):
        """W_eff valid / stale per entry (`current`: tuple of entry indices): a stale merged weight switches the layer to the
        branch-apart path (leaves.run_layer compares `merge_scale` with the wrapper's scale), so skipping a refresh can never
        feed stale weights to a forward."""
        if current == getattr(self, "_current", None):
            return
        self._current = current
        cur = set(current)
        groups = {}
        for k, (e, mod) in enumerate(self.entries):
            e.merge_scale = float(mod.scale) if k in cur else None
            if e.group is not None:
                groups[id(e.group)] = (e.group, k in cur)
        for g, on in groups.values():
            g.merge_scale = float(g.mods[0].scale) if on else None

    def run(self):
        """Refresh the W_eff that are read this step from the current fp32 factors (asynchronous on the current stream;
        graph-capture safe once the subset's table exists)."""
        if not self.njobs:
            return
        if not torch.cuda.is_current_stream_capturing():
            self.sync_base()
        want = self.wanted()
        if not want:
            self._mark(())
            return
        jobs_dev, tile_job_dev, n, ntiles = self._table(None if len(want) == self.njobs else want)
        self._mark(want)
        from . import native as nv
        nv.call("t2v_lora_merge", jobs_dev.data_ptr(), n, tile_job_dev.data_ptr(), ntiles, nv.stream())


class PrepPlan:
    """Transposed bf16 factor copies for the wrapped layers whose dropout is ACTIVE (the reference's default train mode,
    utils/lora.py:35,49,89,119): the branch cannot be merged into W, the base layer's launch adds it in its epilogue instead
    (T2VGemm.lr_mode; functional._LoraLayer) and reads the factors row-per-output-column:
      up_t16   bf16 [Np, rk]         = U^T                       forward   y  += s mask (t U^T)
      down_t16 bf16 [Cin_p, taps*rk] = s D^T, flipped taps       backward  dx += s dt (*) D^T
    refreshed by ONE `t2v_lora_prep` launch per step (device job table built once; every buffer it points to is allocated
    once).  `prep_scale` on an entry = the scale baked into its copies while they are current, else None (the layers then take
    the separate rank-update passes)."""

    def __init__(self, plans, flat_p):
        from . import native as nv
        dev = flat_p.device
        BF16 = torch.bfloat16
        seen, entries = set(), []
        for pid, (e, role, mod) in plans.items():
            if id(e) in seen or not all(hasattr(e, a) for a in ("down_off", "up_off")):
                continue
            seen.add(id(e))
            e.prep_scale = None
            if e.rp > 32:
                continue
            entries.append((e, mod))
        self.entries = entries
        jobs = (nv.LoraPrepJob * max(1, len(entries)))()
        total = 0
        lib = nv.lib()
        groups = {}
        for k, (e, mod) in enumerate(entries):
            e.rk = 16 if e.rp <= 16 else 32
            g = e.group
            if g is not None and g.rp_each * g.n <= 48 and e.rk == 16:
                # members of a projection group: row blocks of ONE up^T buffer [n*Np, rk] and column blocks of ONE (s D)^T buffer
                # [Cin_p, n*rk] — what the grouped launches read (functional._LoraGroupDrop); a member evaluated on its own
                # reads its block through the strides
                # (the (s D)^T blocks sit at stride rp_each — the stride of the members' dt columns in dt_cat, the other operand
                #  of dx += s dt_cat D_cat; 16 spare columns so that a member read on its own may run 16 ranks wide)
                if id(g) not in groups:
                    g.rk = e.rk
                    g.up_t16 = torch.zeros(g.npad, e.rk, dtype=BF16, device=dev)
                    g.down_t16 = torch.zeros(g.cin_p, (g.n * g.rp_each + 15) // 16 * 16 + 16, dtype=BF16, device=dev)
                    g.prep_scale = None
                    groups[id(g)] = g
                e.up_t16 = g.up_t16[e.gidx * e.npad:(e.gidx + 1) * e.npad]
                e.down_t16 = g.down_t16[:, e.gidx * e.rp:]
                rkd = e.rp
            else:
                e.up_t16 = torch.zeros(e.npad, e.rk, dtype=BF16, device=dev)
                e.down_t16 = torch.zeros(e.cin_p, e.taps * e.rk, dtype=BF16, device=dev)
                rkd = e.rk
            j = jobs[k]
            if e.group is not None:          # the member's column block of the group's up matrix
                j.up, j.ldu = flat_p.data_ptr() + 4 * (e.up_off + e.gidx * e.npad), e.group.npad
            else:
                j.up, j.ldu = flat_p.data_ptr() + 4 * e.up_off, e.npad
            j.down = flat_p.data_ptr() + 4 * e.down_off
            j.upT, j.dnT = e.up_t16.data_ptr(), e.down_t16.data_ptr()
            j.Np, j.Cp, j.taps, j.rp, j.rk, j.scale = e.npad, e.cin_p, e.taps, e.rp, e.rk, float(mod.scale)
            j.ldt, j.rkd = e.down_t16.stride(0), rkd
            j.chunk0 = total
            total += int(lib.t2v_lora_prep_chunks(e.npad, e.cin_p, e.taps, e.rk, rkd))
        self.njobs, self.total = len(entries), total
        if self.njobs:
            self.jobs_dev = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        self._scales = [float(mod.scale) for _, mod in entries]
        self.groups = list(groups.values())

    def wanted(self):
        for _, mod in self.entries:
            d = getattr(mod, "dropout", None)
            if isinstance(d, torch.nn.Dropout) and d.training and d.p > 0.0:
                return True
        return False

    def run(self):
        if not self.njobs:
            return
        want = self.wanted()
        if not torch.cuda.is_current_stream_capturing():
            for (e, mod), s0 in zip(self.entries, self._scales):
                if float(mod.scale) != s0:
                    raise RuntimeError("t2v_amd: a LoRA wrapper's scale changed after the trainer was built "
                                       f"({s0} -> {float(mod.scale)}); rebuild the optimiser / trainer")
        for (e, mod), s0 in zip(self.entries, self._scales):
            e.prep_scale = s0 if want else None
        for g in self.groups:
            g.prep_scale = float(g.mods[0].scale) if want else None
        if not want:
            return
        from . import native as nv
        nv.call("t2v_lora_prep", self.jobs_dev.data_ptr(), self.njobs, self.total, nv.stream())


def is_homed(homes):
    """True if EVERY Parameter still aliases its flat-buffer view (a partial move — `text_encoder.to(...)`, a sub-module
    `.float()` — would otherwise detach parameters mid-list from the optimizer unnoticed; the check is ~1e3 pointer compares)."""
    for p, v, g in homes:
        if p.data_ptr() != v.data_ptr() or p.grad is None or p.grad.data_ptr() != g.data_ptr():
            return False
    return True


def rehome(homes):
    """Point every Parameter (and its .grad) back at its flat-buffer view, keeping the CURRENT parameter values.
    Needed after `model.cpu()` / `.to(device)` round trips — the reference's `save_pipe` does one at every checkpoint
    (train.py:417-442) — which re-allocate the parameter storages and would silently detach them from the optimizer."""
    with torch.no_grad():
        for p, v, g in homes:
            if p.data_ptr() != v.data_ptr():
                v.copy_(p.detach().to(device=v.device, dtype=v.dtype))
                p.data = v
            # `g` itself is never handed out: Module._apply swaps the .data of the very tensor object that is p.grad
            if p.grad is None:
                p.grad = g.detach()
            elif p.grad.data_ptr() != g.data_ptr():      # a gradient autograd (or a device move) put elsewhere: keep its values
                g.copy_(p.grad.detach().to(device=g.device, dtype=g.dtype))
                p.grad = g.detach()
