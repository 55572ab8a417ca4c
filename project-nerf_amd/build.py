"""Builds libnerf_hip.so (the C-ABI shared library of include/nerf_hip.h) for gfx950.

hipcc cross-compiles without a GPU; objects go to project-nerf_amd/build/, the library
next to this file so that it travels with the source tree.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libnerf_hip.so")
COMM_SRC = os.path.join(HERE, "csrc_comm", "comm.cpp")
COMM_LIB = os.path.join(HERE, "libnerf_comm.so")
ROCM_LIB = os.environ.get("ROCM_LIB", "/opt/rocm/lib")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-Wall",
          "-Wno-unused-function", "-I", os.path.join(HERE, "..", "include")]
PER_FILE = {"sample.hip": ["-ffp-contract=off"],
            "mlp_fwd.hip": ["-Wno-inline-asm"], "mlp_bwd.hip": ["-Wno-inline-asm"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "nerf_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, verbose):
    obj = os.path.join(OUT, src + ".o")
    path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), _headers_mtime()):
        return obj, False
    cmd = [HIPCC, "-x", "hip", "-c", path, "-o", obj] + COMMON + PER_FILE.get(src, [])
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip() and verbose:
        print(r.stderr)
    return obj, True


GENERATED = {"mlp_mtile_asm.h": "gen_mtile_asm.py", "mlp_stream_asm.h": "gen_stream_asm.py"}


def _default_config(header):
    """False for a header generated with timing ablations or a non-default window depth
    (tools/ablate_stream.sh): such a stream is numerically wrong and must never be reused."""
    with open(header) as f:
        head = f.read(400)
    cfg = [ln for ln in head.splitlines() if ln.startswith("// GEN_CONFIG")]
    return not cfg or cfg[0].strip() == "// GEN_CONFIG D=4 NO="


def _generate(verbose):
    """inline-asm headers are generated (and git-ignored): csrc/gen_*.py -> csrc/*.h"""
    for header, script in GENERATED.items():
        h, g = os.path.join(CSRC, header), os.path.join(CSRC, script)
        keep = os.environ.get("NERF_BUILD_KEEP_HEADERS") is not None      # tools/ablate_stream.sh only
        if os.path.exists(h) and os.path.getmtime(h) > os.path.getmtime(g) and (keep or _default_config(h)):
            continue
        if verbose:
            print(f"{script} -> {header}", flush=True)
        r = subprocess.run([sys.executable, g], capture_output=True, text=True, cwd=CSRC,
                           env={k: v for k, v in os.environ.items() if not k.startswith("GEN_")})
        if r.returncode != 0:
            raise RuntimeError(f"{script} failed:\n{r.stderr}")
        with open(h, "w") as f:
            f.write(r.stdout)


def build(verbose=False, jobs=None):
    os.makedirs(OUT, exist_ok=True)
    _generate(verbose)
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, verbose), srcs))
    objs = [o for o, _ in res]
    if any(changed for _, changed in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    try:
        build_comm(verbose)
    except RuntimeError as e:
        # libnerf_hip.so does not need RCCL: without librccl only the optional exchange library is missing, and
        # _comm.NativeComm raises when something tries to load it
        print(f"warning: libnerf_comm.so was not built ({str(e).splitlines()[0]})", file=sys.stderr)
    return LIB


def build_comm(verbose=False):
    """libnerf_comm.so: the RCCL exchange step behind include/nerf_comm.h (host code only; links librccl)"""
    header = os.path.join(HERE, "..", "include", "nerf_comm.h")
    if os.path.exists(COMM_LIB) and os.path.getmtime(COMM_LIB) > max(os.path.getmtime(COMM_SRC), os.path.getmtime(header)):
        return COMM_LIB
    cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", os.path.join(HERE, "..", "include"), COMM_SRC, "-o", COMM_LIB,
           "-L" + ROCM_LIB, "-lrccl", "-Wl,-rpath," + ROCM_LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building libnerf_comm.so failed:\n{r.stdout}\n{r.stderr}")
    return COMM_LIB


if __name__ == "__main__":
    print(build(verbose="-q" not in sys.argv))
