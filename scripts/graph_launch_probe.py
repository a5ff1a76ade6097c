"""How hipGraphLaunch behaves on this runtime when the previous launch is still running (profiles/r04_graph_launch_probe.txt):
host time of each of 4 back-to-back replays of a graph of N small kernels, for a single-stream graph, a graph with a forked side
branch, alternating launch streams, and alternating executables."""
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 22, device=dev)          # 16 MB: ~15 us per add
y = torch.zeros(1 << 22, device=dev)


def body(fork):
    side = torch.cuda.Stream() if fork else None
    for i in range(N):
        x.add_(1.0)
        if fork and i % 500 == 10:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(20):
                    y.add_(1.0)
        if fork and i % 500 == 400:
            torch.cuda.current_stream().wait_stream(side)
    if fork:
        torch.cuda.current_stream().wait_stream(side)


def capture(fork, pool=None):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool):
        body(fork)
    return g


def run(tag, graphs, streams):
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    times = []
    prev = None
    for i in range(4):
        s = streams[i % len(streams)]
        with torch.cuda.stream(s):
            if prev is not None and len(streams) > 1:
                s.wait_event(prev)
            t = time.perf_counter()
            graphs[i % len(graphs)].replay()
            times.append((time.perf_counter() - t) * 1e3)
            x.mul_(1.0)                     # an eager kernel behind the graph (the optimizer's place)
            prev = torch.cuda.Event()
            prev.record(s)
    t_issue = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) * 1e3
    print(f"{tag:55s} host per launch {[round(v, 1) for v in times]} ms; all issued after {t_issue:.1f} ms; device done after {t_all:.1f} ms")


main = torch.cuda.Stream()
s2 = torch.cuda.Stream()
with torch.cuda.stream(main):
    body(False)
torch.cuda.synchronize()
for fork in (False, True):
    with torch.cuda.stream(main):
        g1 = capture(fork)
        g2 = capture(fork, pool=g1.pool())
    nm = "forked graph" if fork else "single-stream graph"
    run(f"{nm}, {N} nodes: one exec, one stream", [g1], [main])
    run(f"{nm}: two execs, one stream", [g1, g2], [main])
    run(f"{nm}: one exec, two streams (event-chained)", [g1], [main, s2])
    run(f"{nm}: two execs, two streams (event-chained)", [g1, g2], [main, s2])
