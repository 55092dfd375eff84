#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) of bench.py
(tools/profile_round.sh).  gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at
64 B, so bytes = FETCH_SIZE[KiB] * 1024 * 2 + WRITE_SIZE[KiB] * 1024.
    python tools/pmc_traffic.py gpurun_out/pmc_<tag>_fetch gpurun_out/pmc_<tag>_write <batch> [precision] > profiles/<tag>_pmc_traffic.json"""
import csv
import glob
import json
import os
import sys


def load(d, counter):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            key = "cross_attn_kernel" if "cross_attn_kernel" in k else "gemm_all" if "gemm_" in k else None
            did = r.get("Dispatch_Id") or r.get("Correlation_Id")
            short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()
            for kk in ([key] if key else []) + (["kernel:" + short] if ("gemm_" in k or "attn" in k) else []):
                per.setdefault(kk, {})
                per[kk][did] = per[kk].get(did, 0.0) + float(r["Counter_Value"])  # summed over XCDs / SEs
    return {k: dict(launches=len(v), avg_KiB_raw=sum(v.values()) / max(len(v), 1)) for k, v in per.items()}


def main():
    fetch, write, batch = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 1 --cpu-users 0 --no-prof",
           "correction": "bytes = FETCH_SIZE[KiB]*1024*2 + WRITE_SIZE[KiB]*1024 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE tallies 128-B requests at 64 B)",
           "batch": batch, "precision": sys.argv[4] if len(sys.argv) > 4 else "bf16x3"}
    for key, name in (("cross_attn_kernel", "hbm_bytes_per_launch"), ("gemm_all", "hbm_bytes_per_launch_avg")):
        f, w = fetch.get(key), write.get(key)
        if f and w:
            out[key] = {name: f["avg_KiB_raw"] * 1024 * 2 + w["avg_KiB_raw"] * 1024, "raw": {"FETCH_SIZE": f, "WRITE_SIZE": w}}
    out["per_kernel"] = {}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith("kernel:"):
            f, w = fetch.get(k, {"launches": 0, "avg_KiB_raw": 0.0}), write.get(k, {"launches": 0, "avg_KiB_raw": 0.0})
            out["per_kernel"][k[7:]] = {"launches": f["launches"] or w["launches"], "fetch_bytes_per_launch": f["avg_KiB_raw"] * 2048,
                                        "write_bytes_per_launch": w["avg_KiB_raw"] * 1024}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
