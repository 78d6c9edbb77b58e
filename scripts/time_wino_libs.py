"""Run ON THE GPU BOX: python scripts/time_wino_libs.py lib1.so lib2.so ... — one process per (library, shape) of scripts/time_wino.py,
interleaved so that every library sees the same clocks; prints a table (ms)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = [(16, 512, 64), (16, 256, 128), (16, 128, 256), (16, 512, 32)] if not os.environ.get("SHAPES") else \
    [tuple(int(v) for v in s.split(",")) for s in os.environ["SHAPES"].split(";")]
res = {}
for rep in range(int(os.environ.get("REPS", 2))):
    for (B, C, H) in shapes:
        for lib in sys.argv[1:]:
            env = dict(os.environ, B=str(B), C=str(C), H=str(H))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_wino.py"), lib], env=env, capture_output=True, text=True).stdout
            import re
            m_ = re.search(r"([0-9.]+) ms  direct-equiv", out)
            ms = float(m_.group(1)) if m_ else float("nan")
            mhz = int(out.split("this kernel: ")[1].split(" MHz")[0]) if "this kernel: " in out else 0
            res.setdefault((lib, (B, C, H)), []).append((ms, mhz))
print("| library | " + " | ".join(f"B{B} C{C} H{H}" for (B, C, H) in shapes) + " |")
print("|---|" + "---|" * len(shapes))
for lib in sys.argv[1:]:
    print(f"| {lib} | " + " | ".join("%.3f @%d MHz" % min(res[(lib, s)]) for s in shapes) + " |")
