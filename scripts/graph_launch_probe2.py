"""Device idle time per hipGraphLaunch on this runtime, by launch pattern (profiles/r04_graph_launch_probe2.txt).  A single-stream
graph of N kernels (~T1 ms) is replayed L times without host synchronisation; idle per launch = (total - L*T1) / L."""
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2600
L = 8
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 24, device=dev)
y = torch.zeros(1 << 20, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)


def capture(pool=None, stream=None):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool, stream=stream):
        cnt.add_(1)
        for i in range(N):
            x.add_(1.0)
    return g


main = torch.cuda.Stream()
aux = torch.cuda.Stream()
cap = torch.cuda.Stream()
with torch.cuda.stream(main):
    x.add_(1.0)
torch.cuda.synchronize()
gs = [capture(stream=cap)]
for _ in range(2):
    gs.append(capture(pool=gs[0].pool(), stream=cap))


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    ti = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, ti * 1e3


with torch.cuda.stream(main):
    for g in gs:
        g.replay()
    torch.cuda.synchronize()
    t1s = []
    for _ in range(3):
        t, _ = timed(lambda: gs[0].replay())
        t1s.append(t)
    T1 = min(t1s)
    print(f"N={N}: one replay alone {T1:.2f} ms (of {[round(v, 2) for v in t1s]})")

    def pattern(tag, nexec, eager=False, evwait=False, other_stream_work=False):
        def run():
            for i in range(L):
                if evwait:
                    with torch.cuda.stream(aux):
                        y.add_(1.0)
                        ev = torch.cuda.Event()
                        ev.record(aux)
                    main.wait_event(ev)
                gs[i % nexec].replay()
                if eager:
                    y.mul_(1.0)
                    y.mul_(1.0)
        tot, issue = timed(run)
        print(f"  {tag:58s} total {tot:7.2f} ms, host issue {issue:6.2f} ms, idle per launch {(tot - L * T1) / L:6.2f} ms")

    pattern("1 exec, back to back", 1)
    pattern("2 execs alternating", 2)
    pattern("3 execs rotating", 3)
    pattern("2 execs + two eager kernels after each", 2, eager=True)
    pattern("1 exec + two eager kernels after each", 1, eager=True)
    pattern("2 execs + event wait (other stream) before each", 2, evwait=True)
    pattern("2 execs + event wait + eager kernels", 2, eager=True, evwait=True)
