"""Sweep the register-direct weight-gradient wave-tile plan over the discriminator's layer shapes (time_wgrad_d.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for plan in ["2,1", "3,1", "4,1", "2,2", "1,2", "1,4", "2,4"]:
    env = dict(os.environ, CAGC_WGRAD_RD_PLAN=plan)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_wgrad_d.py")], env=env, capture_output=True, text=True)
    print("== plan", plan)
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("s") or l.startswith("total")), r.stderr[-300:] if r.returncode else "")
