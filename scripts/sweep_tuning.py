#!/usr/bin/env python3
"""Run ON THE GPU BOX: one-at-a-time sweep of the launch-policy knobs (environment form of cagc_set_tuning) on the graph-replayed KD
step at small per-GPU batches.  python scripts/sweep_tuning.py [batches...]   ->  ms/step per (knob, value, batch)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS = {"CAGC_RD_MIN_WGS": [256, 512, 768, 1024], "CAGC_RD_ATOMIC_BELOW": [0, 96, 160, 320], "CAGC_RD_SPLIT_WGS": [256, 512, 768],
         "CAGC_WINO4_MIN_WGS": [128, 256, 512], "CAGC_WGRAD_RD_WGS": [256, 384, 768]}       # WGRAD_RD_WGS default 0 = the launch model
if os.environ.get("SWEEP_KNOBS"):      # e.g. SWEEP_KNOBS=CAGC_RD_ATOMIC_BELOW,CAGC_RD_SPLIT_WGS
    KNOBS = {k: v for k, v in KNOBS.items() if k in os.environ["SWEEP_KNOBS"].split(",")}


def run(env_extra, bs):
    env = dict(os.environ, **{k: str(v) for k, v in env_extra.items()})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--graph", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-roofline",
           "--no-full-iteration", "--no-proxy", "--no-config3", "--sweep", "0", "--local-batch", str(bs)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[0])["median_ms_per_step"] if line else None


def main():
    batches = [int(a) for a in sys.argv[1:]] or [2, 4]
    base = {b: run({}, b) for b in batches}
    print("baseline", base, flush=True)
    for k, vals in KNOBS.items():
        for v in vals:
            row = {b: run({k: v}, b) for b in batches}
            print(k, v, row, {b: round(row[b] / base[b] - 1, 4) if row[b] and base[b] else None for b in batches}, flush=True)
    print("baseline again", {b: run({}, b) for b in batches}, flush=True)


if __name__ == "__main__":
    main()
