"""In-step phase times of the vanilla training step (development aid): bench.py's kernels_in_step, three runs."""
import json, subprocess, sys
for _ in range(3):
    r = subprocess.run([sys.executable, "bench.py", "--no-instant", "--no-render", "--no-cpu-baseline", "--steps", "100"], capture_output=True, text=True)
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    k = d["kernels_in_step"]
    print(f"step {d['ms_per_step']:.4f} ms  fwd {k['fwd']['ms']:.4f}  dgrad {k['dgrad']['ms']:.4f}  wgrad {k['wgrad']['ms']:.4f}", flush=True)
