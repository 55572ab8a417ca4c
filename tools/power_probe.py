"""Board power and shader clock (rocm-smi) while the decoder chain kernels run back to back (development aid):
evidence for 'the chains are power-limited'.  usage: python tools/power_probe.py [infer|train|dgrad|wgrad]"""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import project_nerf_amd
from project_nerf_amd import ops
from project_nerf_amd.engine import default_init
mode = sys.argv[1] if len(sys.argv) > 1 else "infer"
packed = ops.mlp_pack(default_init(0).cuda())
R, S = 8192, 64
n = R * S
o = torch.randn(R, 3, device="cuda"); d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1)
z = ops.sample_rays(o, d, 2.0, 6.0, S)
stash = torch.empty(ops.mlp_stash_bytes(n), dtype=torch.uint8, device="cuda") if mode != "infer" else None
lib = ops._lib.load(); st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
if mode in ("dgrad", "wgrad"):
    rgb, sigma = ops.mlp_fwd(packed, o, d, z, stash)
    ws = torch.empty(ops.mlp_bwd_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    grads = torch.empty(ops.MLP_PARAM_COUNT, device="cuda")
    d_rgb, d_sigma = torch.randn_like(rgb), torch.randn_like(sigma)
    lib.nerf_mlp_bwd_dgrad(P(packed), P(stash), P(rgb), P(sigma), P(d_rgb), P(d_sigma), n, P(ws), st)
def launch():
    if mode == "dgrad":
        lib.nerf_mlp_bwd_dgrad(P(packed), P(stash), P(rgb), P(sigma), P(d_rgb), P(d_sigma), n, P(ws), st)
    elif mode == "wgrad":
        lib.nerf_mlp_bwd_wgrad(P(stash), P(ws), n, P(grads), st)
    else:
        ops.mlp_fwd(packed, o, d, z, stash) if stash is not None else ops.mlp_fwd(packed, o, d, z)
stop = False
def smi():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
        keep = [ln.strip() for ln in r.splitlines() if any(k in ln for k in ("Power", "sclk", "fclk", "mclk"))]
        print(" | ".join(k.split("GPU[0]")[-1].strip(" :\t") for k in keep), flush=True)
        time.sleep(1.0)
print("idle:"); t = threading.Thread(target=smi); t.start(); time.sleep(2.5)
print(f"{mode} kernel, 524,288 samples per launch, back to back:", flush=True)
t0 = time.time()
while time.time() - t0 < 8.0:
    for _ in range(200):
        launch()
    torch.cuda.synchronize()
stop = True; t.join()
