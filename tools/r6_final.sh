#!/bin/bash
# Round-6 measurement call: the whole GPU suite + smoke, the default bench line, the rocprofv3 kernel summary of the same UNet step, the two
# PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) behind roofline.traffic, and the whole-step MFMA-busy pass by class.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${RUN:-r6final}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err | cut -c1-300
cd /tmp
rm -rf /tmp/prof_stats /tmp/prof_f /tmp/prof_w /tmp/prof_m
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --steps 10 --warmup 2 --clip 0 --cpu-baseline 0 --distill 0 --full-finetune 0 --graph 0 > $O/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/unet_bench_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 --full-finetune 0 > $O/prof_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 --full-finetune 0 > $O/prof_write.log 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) $O/gemm_traffic.json; head -c 600 $O/gemm_traffic.json
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_m -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 --full-finetune 0 > $O/prof_mfma.log 2>&1
python $R/tools/pmc_mfma_busy.py $(find /tmp/prof_m -name "*counter_collection.csv" | head -1) > $O/mfma_busy_by_class.csv 2>$O/mfma_busy.err; cat $O/mfma_busy_by_class.csv | head -12
head -14 $O/unet_bench_kernel_stats.csv | cut -c1-220
