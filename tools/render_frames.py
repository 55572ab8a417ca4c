"""800 x 800 x 128-sample frames through the vanilla engine's render chain (for rocprofv3 --stats: where a frame's time goes).
    python tools/render_frames.py [n_frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from project_nerf_amd.engine import VanillaNerfEngine  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
eng = VanillaNerfEngine(seed=0, device="cuda")
H = W = 800
o = torch.nn.functional.normalize(torch.randn(1, 3, device="cuda"), dim=-1).expand(H * W, 3).contiguous() * 4.0
d = torch.nn.functional.normalize(-o + 0.4 * torch.randn(H * W, 3, device="cuda"), dim=-1)
eng.render_image(o, d, 128)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_frames):
    eng.render_image(o, d, 128)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n_frames
print(f"{n_frames} frames: {dt * 1e3:.2f} ms per frame = {1 / dt:.2f} FPS")
