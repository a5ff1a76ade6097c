"""Per-shape timing of every GEMM-family launch in one C2 train step (eager, HIP events)."""
import sys, os, collections; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F
from bench import build_models, synthetic_batch, CONFIGS
from t2v_amd.training import DenoiseTrainer
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
frames, H, W, r = CONFIGS[cfg]
dev = torch.device('cuda', 0)
unet, vae, trainable = build_models(frames, r, dev, 0)
tr = DenoiseTrainer(unet, vae, trainable)
batch = synthetic_batch(frames, H, W, dev, 1234)
tr.opt.zero_grad(); tr._fwd_bwd(batch); torch.cuda.synchronize()
recs = []
orig = F.launch_gemm
def timed(**kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); orig(**kw); e.record()
    g = kw.get('geom')
    key = (kw['M'], kw['N'], kw['K'], kw.get('a_mode', 0), kw.get('a_trans', 0), kw.get('b_trans', 0), kw.get('b_conv', 0), kw.get('split_k', 1), kw.get('batch', 1), (g.KH, g.KW, g.sy, g.tdiv, g.up) if g is not None else None)
    recs.append((key, s, e))
F.launch_gemm = timed
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
tr.opt.zero_grad(); t0.record(); tr._fwd_bwd(batch); t1.record(); torch.cuda.synchronize()
F.launch_gemm = orig
agg = collections.OrderedDict()
for key, s, e in recs:
    ms = s.elapsed_time(e)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += ms
tot = sum(a[1] for a in agg.values())
print(f'total step (eager, instrumented) {t0.elapsed_time(t1):.1f} ms; gemm {tot:.1f} ms in {len(recs)} launches')
print('M N K amode at bt bconv splitk batch geom | count total_ms avg_us TF/s')
for key, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    M, N, K = key[:3]; fl = 2.0 * M * N * K * max(1, key[8])
    if key[9] is not None and key[9][3] == 2: fl /= 4
    print(key, '|', cnt, f'{ms:.2f} {ms / cnt * 1e3:.1f} {fl * cnt / (ms * 1e-3) / 1e12:.1f}')
