"""Kernel timings of the Instant-NGP step (development aid): bench.py --workload instant, two runs, one line each."""
import json, subprocess, sys
for _ in range(2):
    r = subprocess.run([sys.executable, "bench.py", "--workload", "instant", "--steps", "60"], capture_output=True, text=True)
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    k = d["kernels"]
    print(f"step {d['ms_per_step']:.4f} ms  " + "  ".join(f"{n} {v['ms']*1e3:.1f}" for n, v in k.items()), flush=True)
