"""Print the headline fields of a bench.py JSON line:  python scripts/show_bench.py gpurun_out/bench.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], "ms/step  roofline.frac", r["frac"], r["kernel"], "clock", r.get("shader_clock_mhz"))
for k, v in r["all_mfma_entry_points"].items():
    print("   ", k, v)
for k, v in r.get("hbm_bound_entry_points", {}).items():
    print("   ", k, v["ms_per_step"], v["achieved_GBps"], v["largest_launch_GBps"])
for k in ("strong_scaling_proxy_1gpu", "full_iteration", "saliency_sweep", "config3_1024", "configs0_cpu_forward", "cpu_baseline"):
    print(k, json.dumps(d.get(k))[:400])
