#!/usr/bin/env python
"""HBM traffic of the GEMM family per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit in
one pass: 4 TCC slots).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced
stream's bytes (MI355X_MICROARCH.md §HBM), so the read side is doubled.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out_f -- python bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out_w -- python bench.py ...   (same)
    python tools/pmc_traffic.py out_f/*/*counter_collection.csv out_w/*/*counter_collection.csv profiles/r01_gemm_traffic.json"""
import csv
import json
import sys


def collect(path, counter):
    tot, n = {}, {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        fam = ("gemm" if ("gemm_kernel" in name or "splitk_reduce" in name or "conv_halo_kernel" in name or "gemm2_kernel" in name or "linear_pr_kernel" in name) else
               "other_t2v" if "anonymous namespace" in name and "at::" not in name else None)
        if fam is None:
            continue
        tot[fam] = tot.get(fam, 0.0) + float(r["Counter_Value"])
        if "splitk_reduce" not in name:
            n[fam] = n.get(fam, 0) + 1
    return tot, n


f_tot, f_n = collect(sys.argv[1], "FETCH_SIZE")
w_tot, w_n = collect(sys.argv[2], "WRITE_SIZE")
out = {}
for fam in f_tot:
    launches = f_n[fam]
    rd = 2.0 * f_tot[fam] * 1024 / launches
    wr = w_tot.get(fam, 0.0) * 1024 / max(w_n.get(fam, 1), 1)
    out[fam] = {"launches_seen": launches, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                "hbm_bytes_per_launch": rd + wr}
out["method"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB units, gfx950 FETCH_SIZE x2 correction), eager "
                 "replay of the UNet step; per launch = sum over dispatches / t2v_gemm launches (split-K reduce folded in)")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
