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
import json
import re
import statistics
import sys


def short(name):
    """key of a kernel in the summary: its name inside namespace nerf (nerf::p4::x -> p4::x) with its template arguments;
    every kernel of the library is kept (round 2 dropped tv_normsq_kernel<true> & co. here, not in rocprofv3)"""
    if "nerf::" not in name and "_ZN4nerf" not in name:
        return None
    m = re.search(r"nerf::((?:\w+::)*\w+)", name)
    if m is None:
        return name.split("(")[0]
    base = m.group(1)
    if base == "hash_fwd_kernel":                        # template argument is a vector type: nested brackets
        return "hash_fwd_kernel<fp16 table>" if "_Float16" in name else "hash_fwd_kernel<fp32 table>"
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
        e = {"launches": len(fetch[k]["FETCH_SIZE"]), "FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w,
             "hbm_bytes_per_launch_corrected": 2 * f * 1024 + w * 1024}
        if k in sq and sq[k].get("GRBM_GUI_ACTIVE"):
            s = {c: statistics.median(v) for c, v in sq[k].items()}
            e["mfma_busy_frac"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (s["GRBM_GUI_ACTIVE"] / 8 * 1024)
            e["wait_any/wave_cycles"] = s["SQ_WAIT_ANY"] / max(s["SQ_WAVE_CYCLES"], 1.0)
        out[k] = e
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    for k, e in out.items():
        print(f"{k:28s} {e['hbm_bytes_per_launch_corrected'] / 1e6:10.1f} MB  mfma_busy {e.get('mfma_busy_frac', 0):.3f}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
