#!/usr/bin/env python3
"""Run ON THE GPU BOX: who launches what on one eager KD generator step (torch.profiler, one stream, with Python stacks).

    python scripts/launch_census.py [--local-batch 2] [--top 60]

Prints, for every device kernel of a step, the launch count and total time, attributed to the innermost Python frame inside
this repository (cagc/*.py file:line) — the list the launch-count work of DESIGN §5 "small per-GPU batch" is driven by."""
import argparse
import collections
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--local-batch", type=int, default=2)
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--size", type=int, default=256)
    args = ap.parse_args()
    from cagc import kd
    from cagc.op import modconv as mc
    dev = torch.device("cuda:0")
    student, teacher, disc = kd.build_synthetic_workload(args.size, dev, seed=0)
    kd.OVERLAP_TEACHER = False
    mc._SIDE_LIMIT = 0
    bs = args.local_batch
    mask = kd.ellipse_mask(bs, args.size, dev)
    step = kd.KDStep(student, teacher, disc)
    rng = random.Random(0)
    for _ in range(3):
        step.sample_and_step(bs, mask, rng, None)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step.sample_and_step(bs, mask, rng, None)
        torch.cuda.synchronize()
    # map every device kernel to the CPU op that launched it (correlation via the profiler's own linkage)
    by_site = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    n_k, t_k = 0, 0.0
    for ev in prof.events():
        ks = getattr(ev, "kernels", None)
        if not ks:
            continue
        site = None
        for fr in (ev.stack or []):
            if "content-aware-gan-compression_amd" in fr or "/bench.py" in fr:
                site = fr.split("content-aware-gan-compression_amd/")[-1].strip()
                break
        site = site or ("<torch> " + ev.name)
        for k in ks:
            d = by_site[site]
            d[0] += 1
            d[1] += k.duration
            d[2][k.name.split("(")[0][:60]] += 1
            n_k += 1
            t_k += k.duration
    print(f"per-GPU batch {bs}, {args.size} px: {n_k} device kernels, {t_k / 1e3:.3f} ms of kernel time in one eager single-stream step")
    print("| launches | total us | launched from | kernels |")
    print("|---|---|---|---|")
    for site, (n, t, names) in sorted(by_site.items(), key=lambda kv: -kv[1][0])[:args.top]:
        print(f"| {n} | {t:.0f} | {site} | {', '.join(f'{k} x{c}' for k, c in names.most_common(4))} |")


if __name__ == "__main__":
    main()
