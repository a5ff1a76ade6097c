#!/usr/bin/env python
"""Summarise the two counter passes of scripts/pmc_step.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over the C2 train step)
into profiles/<name>.json: HBM-side bytes per launch and per step for every kernel family.

    python scripts/pmc_summary.py gpurun_out/r2_pmc_step profiles/r02_pmc_step.json [executions]

FETCH_SIZE / WRITE_SIZE are reported in KB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide (16 B/lane)
streaming read (MI355X_MICROARCH.md, HBM section): reads are doubled; WRITE_SIZE is taken as is.  `executions` = how many
times the step ran in the profiled command (1 eager capture warm-up + warm-up + timed replays)."""
import json
import re
import sys

GEMM_LABEL = "gemm_kernel_dma<...> + gemm_w8_kernel<...> (Linear / Conv2d / Conv3d forward + backward-data, LDS-DMA ring)"
FAMILIES = [("gemm_kernel_dma", GEMM_LABEL), ("gemm_w8_kernel", GEMM_LABEL), ("gemm_skinny_kernel", GEMM_LABEL), ("lora_drop_dt", "lora_drop_dt_kernel (masked dt of the dropped LoRA branches)"),
            ("lora_prep", "lora_prep_kernel"),
            ("gemm_kernel<", "gemm_kernel<..,AT|BT> (K-major operands)"), ("gemm_pair", "gemm_pair_kernel"),
            ("gemm_finalize", "gemm_finalize_kernel (split-K)"), ("lora_wgrad", "lora_wgrad_kernel (factor gradients)"),
            ("lora_merge", "lora_merge_kernel"), ("temporal_fused", "temporal_fused_fwd_kernel"), ("gn_stats_kernelILb0", "gn_stats (forward)"), ("gn_stats_kernelILb1", "gn_stats (backward)"),
            ("gn_apply_kernelILb0", "gn_apply (forward)"), ("gn_apply_kernelILb1", "gn_apply (backward)"),
            ("ln_fwd", "ln_fwd"), ("ln_bwd", "ln_bwd"), ("attn_fwd_packed", "attn_fwd_packed (temporal)"),
            ("attn_bwd_packed", "attn_bwd_packed (temporal)"), ("attn_fwd_wg", "attn_fwd_wg (spatial, shared K/V tiles)"),
            ("attn_fwd_kernel", "attn_fwd (spatial / text)"),
            ("attn_bwd_dq", "attn_bwd_dq"), ("attn_bwd_dkdv", "attn_bwd_dkdv"), ("geglu_fwd", "geglu_fwd"), ("geglu_bwd", "geglu_bwd")]


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+total=([\d.e+]+)\s+per_launch=([\d.]+)\s+avg_us=([\d.]+)", line)
        if m:
            out[m.group(1).strip()] = (int(m.group(3)), float(m.group(4)), float(m.group(6)))
    return out


def demangle_family(name):
    key = name.replace("_ZN12_GLOBAL__N_1", "")
    key = re.sub(r"^\d+", "", key)
    for pat, label in FAMILIES:
        p = pat.replace("<", "I").replace("gemm_kernelI", "gemm_kernelI")
        if p in key or pat in key:
            if pat == "gemm_kernel<" and "gemm_kernel_dma" in key:
                continue
            return label
    return None


def main():
    base, dst = sys.argv[1], sys.argv[2]
    f, w = parse(base + "_FETCH_SIZE.txt"), parse(base + "_WRITE_SIZE.txt")
    # executions of the step in the profiled command: the LoRA merge kernel runs exactly once per step
    # steps inside the counted window: scripts/pmc_step.sh appends it ("steps_in_window N"); older logs: the LoRA merge kernel runs
    # exactly once per step
    once = [v[0] for k, v in f.items() if "lora_merge_kernel" in k]
    win = re.search(r"steps_in_window (\d+)", open(base + "_FETCH_SIZE.txt").read())
    execs = int(sys.argv[3]) if len(sys.argv) > 3 else (int(win.group(1)) if win else (once[0] if once else 4))
    fam = {}
    for name in set(f) | set(w):
        label = demangle_family(name)
        if label is None:
            continue
        n = (f.get(name) or w.get(name))[0]
        d = fam.setdefault(label, dict(launches=0, fetch_kb=0.0, write_kb=0.0, us=0.0))
        d["launches"] += n
        d["fetch_kb"] += f.get(name, (0, 0.0, 0.0))[1]
        d["write_kb"] += w.get(name, (0, 0.0, 0.0))[1]
        d["us"] += (f.get(name) or w.get(name))[2] * n
    rows = {}
    for label, d in sorted(fam.items(), key=lambda kv: -(2 * kv[1]["fetch_kb"] + kv[1]["write_kb"])):
        read_b, write_b = 2.0 * d["fetch_kb"] * 1024, d["write_kb"] * 1024
        rows[label] = dict(launches_per_step=round(d["launches"] / execs, 1), avg_us_under_pmc=round(d["us"] / d["launches"], 1),
                           hbm_read_MB_per_launch=round(read_b / d["launches"] / 1e6, 2),
                           hbm_write_MB_per_launch=round(write_b / d["launches"] / 1e6, 2),
                           hbm_bytes_per_launch=int((read_b + write_b) / d["launches"]),
                           hbm_GB_per_step=round((read_b + write_b) / execs / 1e9, 2),
                           GBps_while_running=round((read_b + write_b) / d["us"] / 1e3, 1))
    out = dict(what="rocprofv3 --kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE (separate passes, scripts/pmc_step.sh) over "
                    "`python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-default-mode --no-host-timing` (config C2 in the "
                    f"reference's default train mode, shipped tile table, EAGER step — the counter tool serialises dispatches and crashed under a "
                    f"captured CLIP tower, scripts/pmc_step.sh; same kernels on the same tiles as the replayed graph): the last {execs} steps of the "
                    "trace (the timed steps); per-kernel-family totals",
               note="counters sit at the eight XCD L2s' memory side: operands shared by workgroups on different XCDs are counted once per XCD",
               fetch_correction="reads = 2 x FETCH_SIZE: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a 16 B/lane streaming "
                                "read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported; both in KB",
               families=rows)
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in rows.items():
        print(f"{k[:60]:60s} {v['launches_per_step']:7.1f}/step  {v['hbm_bytes_per_launch'] / 1e6:8.2f} MB/launch  {v['hbm_GB_per_step']:6.2f} GB/step  {v['GBps_while_running']:7.1f} GB/s")


if __name__ == "__main__":
    main()
