"""How fast does Part 4 training amplify a last-bit difference?  (GPU; python tests/studies/part4_sensitivity.py [steps])

With option "deterministic" two runs of the same steps are bit-equal (tests/test_gpu_deterministic.py), so any difference
between the two runs below comes from ONE cause: a single network weight changed by one unit in the last place before the
first step.  The growth of that difference is what every other source of last-bit noise sees as well -- summation order of
float atomics in the default mode, another partition of the batch over data-parallel ranks -- and explains why loss
trajectories of such runs part ways within tens of steps while their first gradients agree to 1e-5: AdamW turns every
gradient entry, however small, into a step of +-lr (2x lr for the hash tables), so an entry whose sum cancels to rounding
noise moves by +-lr with the sign of that noise."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import project_nerf_amd  # noqa: F401,E402
from project_nerf_amd import ops  # noqa: E402
from test_gpu_deterministic import _part4_engine, _probes, _rays  # noqa: E402


def run(steps, nudge):
    eng = _part4_engine()
    if nudge:
        with torch.no_grad():
            w = eng.net[5568 + 17:5568 + 18]                       # one weight of the displacement decoder's first layer
            w.copy_(torch.nextafter(w, w + 1.0))
        eng.repack()
    R, S = 1024, 32
    o, d, target, g = _rays(R, 6)
    t = torch.rand(R, 1, generator=g).cuda()
    out = []
    for step in range(1, steps + 1):
        loss = float(eng.train_step(o, d, target, t, S, probes=_probes(step) if step % 16 == 0 else None))
        out.append((loss, eng.net.clone(), eng.tables.clone()))
    return out


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    ops.set_deterministic(True)
    a, a2, b = run(steps, False), run(steps, False), run(steps, True)
    assert all(x[0] == y[0] and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2]) for x, y in zip(a, a2)), "deterministic runs differ"
    print("two unperturbed runs with the ordered sums: bit-equal over", steps, "steps")
    print("step   loss            loss (one weight + 1 ulp)   rel diff    |d net| / |net|   |d tables| / |tables|   table entries that differ")
    for k, (x, y) in enumerate(zip(a, b), 1):
        if k <= 8 or k % 4 == 0:
            dn = float((x[1] - y[1]).norm() / x[1].norm())
            dt = float((x[2] - y[2]).norm() / x[2].norm())
            print(f"{k:4d}   {x[0]:.8f}      {y[0]:.8f}                {abs(x[0] - y[0]) / x[0]:.2e}    {dn:.2e}          {dt:.2e}"
                  f"                {int((x[2] != y[2]).sum())} of {x[2].numel()}")
