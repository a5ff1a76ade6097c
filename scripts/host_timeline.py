"""Host-side timeline of the graph-replayed C2 step (no synchronisation inside the loop): how long the host spends in zero_grad,
in the graph launch and in the exchange + update of every step, against the device's step time.  Shows whether the host runs
ahead of the device or is held by it (profiles/r04_host_timeline.txt).

    python scripts/host_timeline.py [steps]        (T2V_GRAPH_PIPELINE=0: the single forked graph)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
import t2v_amd  # noqa: F401
from t2v_amd.training import DenoiseTrainer

dev = torch.device("cuda", 0)
frames, H, W, r = bench.CONFIGS["c2"]
unet, vae, trainable = bench.build_models(frames, r, dev, seed=0, dropout=True, grad_ckpt=False)
te = bench.build_text_encoder(dev)
tr = DenoiseTrainer(unet, vae, trainable, lr=5e-6, world_size=1, text_encoder=te)
batch = bench.synthetic_batch(frames, H, W, dev, seed=1234, with_ids=True)
tr.capture(batch, warmup=1)
marks = []


class Timed:
    def __init__(self, g):
        self.g = g

    def replay(self):
        t = time.perf_counter()
        self.g.replay()
        marks.append(("replay", time.perf_counter() - t))


if getattr(tr, "_pipe", None) is not None:
    for slot in tr._pipe:
        slot["unet"] = Timed(slot["unet"])
else:
    tr._graph = Timed(tr._graph)
_zg, _ex = tr.opt.zero_grad, tr._exchange_and_update


def zg(*a, **k):
    t = time.perf_counter()
    _zg(*a, **k)
    marks.append(("zero_grad", time.perf_counter() - t))


def ex(loss):
    t = time.perf_counter()
    out = _ex(loss)
    marks.append(("update", time.perf_counter() - t))
    return out


tr.opt.zero_grad, tr._exchange_and_update = zg, ex
for _ in range(3):
    tr.replay_step()
torch.cuda.synchronize()
marks.clear()
t0 = time.perf_counter()
stamps = []
for i in range(steps):
    ts = time.perf_counter()
    tr.replay_step()
    stamps.append((ts - t0, time.perf_counter() - t0))
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"pipelined={getattr(tr, '_pipe', None) is not None} steps={steps}: device {t_all / steps * 1e3:.2f} ms/step, "
      f"host issued everything after {t_issue * 1e3:.1f} ms ({t_issue / steps * 1e3:.2f} ms/step)")
for i in range(steps):
    ph = marks[3 * i:3 * i + 3]
    print(f"  step {i}: host enters at {stamps[i][0] * 1e3:7.1f} ms, leaves at {stamps[i][1] * 1e3:7.1f} ms; " +
          ", ".join(f"{n} {d * 1e3:.2f} ms" for n, d in ph))
