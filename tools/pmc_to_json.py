"""profiles/<tag>_pmc_{FETCH,WRITE}_SIZE.csv (rocprofv3 --pmc, one counter per pass) -> profiles/pmc_traffic.json
usage: python tools/pmc_to_json.py r01e"""
import collections
import csv
import json
import sys

tag = sys.argv[1]
out = {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean KiB per launch of bench.py's "
                "workload (N=1e6, d=2).  gfx950: FETCH_SIZE reads 1/2 of the bytes of wide coalesced streaming reads "
                "(MI355X_MICROARCH.md §HBM) -> doubled for the streaming kernel (k_scan) only; kernels dominated by random "
                "line/sector reads (k_search, k_step with the fused gather) are uncalibrated and reported raw.",
       "tag": tag, "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"profiles/{tag}_pmc_{c}.csv")):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        # the instantiations bench.py's timed loop launches (multinomial, fused gather, no sum q^2)
        short = next((name for name, pat in (("k_step", "k_step<1, 2, false, true"), ("k_scan", "k_scan<gpf::InFixQ, 1>"),
                                             ("k_search", "k_search<0>"), ("k_gather", "k_gather<2>")) if pat in k), None)
        if short:
            out["kernels"].setdefault(short, {})[c] = round(sum(v) / len(v), 2)
for k, d in out["kernels"].items():
    d["traffic_bytes"] = int((d["FETCH_SIZE"] * (2 if k == "k_scan" else 1) + d["WRITE_SIZE"]) * 1024)
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"]))
