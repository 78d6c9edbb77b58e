#!/usr/bin/env python3
"""Run ON THE GPU BOX: what a HIP-graph node costs on this stack (ROCm 7.2, MI355X) — the numbers the small-per-GPU-batch work is
planned with.

1. synthetic graphs of N tiny kernels: a linear chain on one stream; the same chain with a fork / join to a side stream every
   `period` kernels (the shape `main.wait_stream(side)` gives the captured step) -> us per node, us per fork/join;
2. the real bs-2 KD step: host time of replay() (no sync) vs device time per replay, for the default streams and with every
   side stream off (one linear chain)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

dev = torch.device("cuda:0")


def timed_replay(g, n=30):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return host, e0.elapsed_time(e1) / n * 1e3


def synthetic(N, numel, period=0, nside=1, deferred=False):
    """deferred: the side stream is forked from ONCE-per-period events of the main stream (side.wait_event) but joins the main
    stream only ONCE at the end — the shape 'all weight gradients as one side chain, one join before the optimiser' would give."""
    x = torch.zeros(numel, device=dev)
    ys = [torch.zeros(numel, device=dev) for _ in range(nside)]
    sides = [torch.cuda.Stream() for _ in range(nside)]
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        x.add_(1.0)
    torch.cuda.current_stream().wait_stream(warm)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for i in range(N):
            if period and i % period == 0:
                for s, y in zip(sides, ys):
                    s.wait_stream(main)
                    with torch.cuda.stream(s):
                        y.add_(1.0)
                x.add_(1.0)
                if not deferred:
                    for s in sides:
                        main.wait_stream(s)
            else:
                x.add_(1.0)
        if deferred:
            for s in sides:
                main.wait_stream(s)
    return timed_replay(g)


def main():
    print("| graph | nodes | host us/replay | device us/replay | device us/node |")
    print("|---|---|---|---|---|")
    for numel in (1024, 1 << 20):
        for N, period, nside in ((100, 0, 1), (500, 0, 1), (500, 10, 1), (500, 4, 1), (500, 4, 2)):
            h, d = synthetic(N, numel, period, nside)
            forks = (N // period) if period else 0
            print(f"| chain of add_({numel}) period {period} sides {nside} | {N + forks * nside} | {h:.0f} | {d:.0f} | {d / (N + forks * nside):.2f} |")
        for N, period in ((500, 10), (500, 4)):
            h, d = synthetic(N, numel, period, 1, deferred=True)
            forks = N // period
            print(f"| chain of add_({numel}) period {period}, side chain joined ONCE at the end | {N + forks} | {h:.0f} | {d:.0f} | {d / (N + forks):.2f} |")
    from cagc import kd
    from cagc.op import modconv as mc
    for bs in (2, 4):
        for label, overlap, side in (("default streams", True, 1 << 40), ("teacher stream only", True, 0), ("one stream", False, 0)):
            kd.OVERLAP_TEACHER = overlap
            mc._SIDE_LIMIT = side
            student, teacher, disc = kd.build_synthetic_workload(256, dev, seed=0)
            mask = kd.ellipse_mask(bs, 256, dev)
            st = kd.GraphedKDStep(student, teacher, disc, bs, mask, random_noise=True)
            rng = random.Random(0)
            for _ in range(3):
                st.sample_and_step(bs, mask, rng, None)
            torch.cuda.synchronize()
            n = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(n):
                st.sample_and_step(bs, mask, rng, None)
            e1.record()
            host = (time.perf_counter() - t0) / n * 1e3
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n * 1e3
            print(f"KD step bs {bs}, {label}: host {host:.2f} ms/step to enqueue, device {e0.elapsed_time(e1) / n:.2f} ms/step, wall {wall:.2f}")
            del st, student, teacher, disc
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
