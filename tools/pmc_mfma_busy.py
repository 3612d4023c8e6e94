#!/usr/bin/env python
"""Whole-step matrix-core utilisation per kernel class from ONE rocprofv3 --pmc pass:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -- python bench.py ...
    python tools/pmc_mfma_busy.py DIR/*/*counter_collection.csv > profiles/rNN_mfma_busy_by_class.csv

SQ_VALU_MFMA_BUSY_CYCLES counts cycles (summed over the 1024 SIMDs), GRBM_GUI_ACTIVE the cycles the dispatch kept the chip busy —
reported by this rocprofv3 summed over the 8 XCDs, hence the / 8 (a 90 us GEMM reads 1.59 M = 8 x 199 k cycles at 2.2 GHz).
MFMA busy fraction of a class = sum(MFMA busy) / (1024 SIMDs x sum(GUI active / 8))."""
import collections
import csv
import re
import sys

CLASSES = [("conv_halo_kernel", "t2v_conv_halo (3x3 conv, halo slab)"), ("linear_pr_kernel", "t2v_linear_pr (short-K linear, resident panel)"), ("gemm_kernel", "t2v_gemm (implicit-GEMM conv / linear)"),
           ("splitk_reduce", "t2v_gemm split-K reduce"), ("attn_spatial", "flash attention (spatial / text)"), ("attn_temporal", "temporal attention"),
           ("gn_", "GroupNorm"), ("group_norm", "GroupNorm"), ("layernorm", "LayerNorm"), ("ffn", "fused FFN")]


def klass(name):
    for key, label in CLASSES:
        if key in name:
            return label
    return "other (elementwise, layout, ATen)"


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        c = klass(r["Kernel_Name"])
        per[c][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[c].add(r["Dispatch_Id"])
    tot_busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in per.values())
    tot_gui = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in per.values())
    print("class,dispatches,gui_active_cycles_div8,share_of_gpu_time,mfma_busy_cycles,mfma_busy_frac_of_1024_simds")
    for c, v in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
        gui = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        print(f"{c},{len(disp[c])},{gui:.0f},{gui / (tot_gui / 8.0):.4f},{busy:.0f},{busy / (1024.0 * gui) if gui else 0.0:.4f}")
    print(f"ALL,{sum(len(d) for d in disp.values())},{tot_gui / 8.0:.0f},1.0000,{tot_busy:.0f},{tot_busy / (1024.0 * tot_gui / 8.0):.4f}")


if __name__ == "__main__":
    main()
