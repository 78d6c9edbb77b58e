"""Sweep the register-direct weight-gradient kernel's wave-tile plan (CAGC_WGRAD_RD_PLAN=mb,nb, read once per process) over the
student's layer shapes: one subprocess per plan.  python scripts/sweep_wgrad_plan.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for plan in ["1,1", "2,1", "1,2", "2,2", "3,1", "3,2", "4,1"]:
    env = dict(os.environ, CAGC_WGRAD_RD_PLAN=plan, LAYERS=os.environ.get("LAYERS", "6,7,8,9,10,11,12"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_wgrad.py")], env=env, capture_output=True, text=True)
    print("== plan", plan)
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("cin")))
