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
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-Wall",
          "-Wno-unused-function", "-I", os.path.join(HERE, "..", "include")]
PER_FILE = {"sample.hip": ["-ffp-contract=off"],
            # the stream asm clobbers a0..a159; hipcc calls AGPRs past its default split "reserved"
            "mlp_fwd.hip": ["-Wno-inline-asm"]}


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


def build(verbose=False, jobs=None):
    os.makedirs(OUT, exist_ok=True)
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
    return LIB


if __name__ == "__main__":
    print(build(verbose="-q" not in sys.argv))
