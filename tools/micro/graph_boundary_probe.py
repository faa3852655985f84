"""What a hipGraph boundary costs on the device: N tiny kernels as ONE single-stream graph, as K back-to-back graphs on one stream,
and as a fork / join of two graphs on two streams (event record + wait between graph launches).
usage: python tools/micro/graph_boundary_probe.py"""
import time

import torch

x = torch.zeros(1024, device="cuda")
y = torch.zeros(1024, device="cuda")


def cap(fn, stream=None, pool=None):
    g = torch.cuda.CUDAGraph()
    kw = {}
    if pool is not None:
        kw["pool"] = pool
    if stream is not None:
        kw["stream"] = stream
    with torch.cuda.graph(g, **kw):
        fn()
    return g


def k(t, n):
    for _ in range(n):
        t.add_(1.0)


def timed(fn, n=2000):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, t_issue / n * 1e6


torch.cuda.synchronize()
g30 = cap(lambda: k(x, 30))
print("one graph, 30 kernels: %.1f us per replay (host issue %.1f)" % timed(g30.replay))
parts = [cap(lambda: k(x, 10)) for _ in range(3)]
print("three graphs x 10 kernels, one stream: %.1f us (host %.1f)" % timed(lambda: [g.replay() for g in parts]))
s2 = torch.cuda.Stream()
ga = cap(lambda: k(x, 10))
gb1 = cap(lambda: k(x, 10))
with torch.cuda.stream(s2):
    gb2 = cap(lambda: k(y, 10), stream=s2)
gc = cap(lambda: k(x, 10))
ev1, ev2 = torch.cuda.Event(), torch.cuda.Event()


def forked():
    cur = torch.cuda.current_stream()
    ga.replay()
    ev1.record(cur)
    with torch.cuda.stream(s2):
        s2.wait_event(ev1)
        gb2.replay()
        ev2.record(s2)
    gb1.replay()
    cur.wait_event(ev2)
    gc.replay()


print("10 | (10 || 10 on a second stream) | 10: %.1f us (host %.1f)  -- serial would be 40 kernels" % timed(forked))
g40 = cap(lambda: k(x, 40))
print("one graph, 40 kernels: %.1f us (host %.1f)" % timed(g40.replay))
