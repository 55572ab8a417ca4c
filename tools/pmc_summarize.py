#!/usr/bin/env python3
"""Reduce the three rocprofv3 --pmc passes of tools/pmc_bench.sh to profiles/rNN_pmc_summary.json.

Per kernel (median over its launches of one bench run):
  * FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE under-counts by 2x
    (MI355X_MICROARCH.md, HBM/rocprofv3 section), so
        hbm_bytes_per_launch_corrected = 2 * FETCH_SIZE_KiB * 1024 + WRITE_SIZE_KiB * 1024
  * SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs, so
        mfma_busy_frac = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024)

usage: python tools/pmc_summarize.py gpurun_out/pmc_r01 profiles/r01_pmc_summary.json
"""
import collections
import csv
import os
import json
import re
import statistics
import sys


_DEMANGLED = {}


def demangle(name):
    """rocprofv3 leaves kernels with _Float16 parameters mangled, and this image's c++filt does not know DF16_: the few
    shapes the library has are decoded here (namespace nerf [:: p4], name, bool / fp16-vector template arguments)"""
    if not name.startswith("_ZN4nerf"):
        return name
    rest, parts = name[len("_ZN4nerf"):], ["nerf"]
    while rest and rest[0].isdigit():
        m = re.match(r"(\d+)", rest)
        n = int(m.group(1))
        parts.append(rest[m.end():m.end() + n])
        rest = rest[m.end() + n:]
    targs = ""
    if rest.startswith("ILb1E"):
        targs = "<true>"
    elif rest.startswith("ILb0E"):
        targs = "<false>"
    elif rest.startswith("IDv2_DF16_"):
        targs = "<_Float16 vector>"
    return "::".join(parts) + targs + "("


def short(name):
    name = demangle(name)
    return _short(name)


def _short(name):
    """key of a kernel in the summary: its name inside namespace nerf (nerf::p4::x -> p4::x) with its template arguments;
    every kernel of the library is kept (round 2 dropped tv_normsq_kernel<true> & co. here, not in rocprofv3)"""
    if "nerf::" not in name and "_ZN4nerf" not in name:
        return None
    m = re.search(r"nerf::((?:\w+::)*\w+)", name)
    if m is None:
        return name.split("(")[0]
    base = m.group(1)
    if base in ("hash_fwd_kernel", "hash_bwd_input_kernel"):     # template argument is a vector type: nested brackets
        return base + ("<fp16 table>" if ("_Float16" in name.split("(")[0] or "DF16_" in name.split("(")[0]) else "<fp32 table>")
    t = re.match(r"<[^()]*>", name[m.end():])
    return base + (t.group(0) if t else "")


def load(path):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            key = short(r["Kernel_Name"])
            if key is not None:
                per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


def main(src, dst):
    import os
    fetch, write = (load(f"{src}/{n}_counter_collection.csv") for n in ("fetch", "write"))
    sq = load(f"{src}/sq_counter_collection.csv") if os.path.exists(f"{src}/sq_counter_collection.csv") else {}
    out = {}
    for k in fetch:
        f = statistics.median(fetch[k]["FETCH_SIZE"])
        w = statistics.median(write[k]["WRITE_SIZE"]) if k in write else 0.0
        n = len(fetch[k]["FETCH_SIZE"])
        total = 2 * sum(fetch[k]["FETCH_SIZE"]) * 1024 + (sum(write[k]["WRITE_SIZE"]) * 1024 if k in write else 0.0)
        # median: the steady-state launch of a kernel that always runs at one size; mean / total: kernels launched at several
        # sizes per step (the four hash grids of Part 4, the optimiser over tables and tiny networks)
        e = {"launches": n, "FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w,
             "hbm_bytes_per_launch_corrected": 2 * f * 1024 + w * 1024, "hbm_bytes_per_launch_mean": total / n, "hbm_bytes_total": total}
        if k in sq and sq[k].get("GRBM_GUI_ACTIVE"):
            s = {c: statistics.median(v) for c, v in sq[k].items()}
            e["mfma_busy_frac"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (s["GRBM_GUI_ACTIVE"] / 8 * 1024)
            e["wait_any/wave_cycles"] = s["SQ_WAIT_ANY"] / max(s["SQ_WAVE_CYCLES"], 1.0)
        out[k] = e
    # what the counters were taken on: bench.py refuses to quote them as `traffic` once the kernels' sources have changed
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import kernel_sources_sha16
    out["_meta"] = {"kernel_sources_sha16": kernel_sources_sha16(), "git_head": os.environ.get("NERF_GIT_HEAD", "unknown"),
                    "collected_by": "tools/pmc_summarize.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ passes (separate passes)"}
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    for k, e in out.items():
        if k == "_meta":
            continue
        print(f"{k:28s} {e['hbm_bytes_per_launch_corrected'] / 1e6:10.1f} MB  mfma_busy {e.get('mfma_busy_frac', 0):.3f}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
