import json, os, subprocess, sys
def run(env_extra, bs):
    env = dict(os.environ, **{k: str(v) for k, v in env_extra.items()})
    cmd = [sys.executable, os.path.join(os.environ["GRAFT_REPO_ROOT"], "bench.py"), "--graph", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-roofline",
           "--no-full-iteration", "--no-proxy", "--no-config3", "--sweep", "0", "--local-batch", str(bs)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[0])["median_ms_per_step"] if line else None
cfgs = {"default": {}, "side_wgrad=0": {"CAGC_SIDE_WGRAD": 0}, "side_wgrad=2^22": {"CAGC_SIDE_WGRAD": 1 << 22}, "side_wgrad=2^25": {"CAGC_SIDE_WGRAD": 1 << 25},
        "skip_gemm_main": {"CAGC_SIDE_SKIP_GEMM": 0}, "blur_w64=0": {"CAGC_BLUR_W64": 0}, "blur_w64=2": {"CAGC_BLUR_W64": 2},
        "wino4_hv=1": {"CAGC_WINO4_HV": 1}, "wino4_hv=2": {"CAGC_WINO4_HV": 2}, "default2": {}}
for name, e in cfgs.items():
    print(name, {b: run(e, b) for b in (2, 4, 8, 16)}, flush=True)
