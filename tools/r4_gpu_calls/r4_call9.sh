#!/bin/bash
# round 4, call 9: new GPU tests (train-mode teacher, full-size VAE decode, mid-width LoRA fixture, LoRA-epilogue default) +
# whole-step MFMA-busy PMC pass by kernel class + rocprofv3 kernel stats of the UNet loop
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c9
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_engine.py -q -x -s -k "teacher or vae_decode_full_size" 2>&1 | grep -E "rel-L2|passed|failed|Error|assert" | tail -8 | tee $O/tests.txt
timeout 1500 python -m pytest tests/test_gpu_train_parity.py -q -x -s -k "mid_width or replayed_masks or one_gemm" 2>&1 | grep -E "fixture|passed|failed|Error|assert" | tail -8 | tee -a $O/tests.txt
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_step /tmp/stats_step
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_step -- python $R/bench.py --steps 3 --warmup 1 --graph 0 --clip 0 --distill 0 --cpu-baseline 0 --breakdown 0 > $O/pmc_bench.json 2> $O/pmc_bench.err
python3 $R/tools/pmc_mfma_busy.py /tmp/pmc_step/*/*counter_collection.csv > $O/mfma_busy_by_class.csv
cat $O/mfma_busy_by_class.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_step -- python $R/bench.py --steps 14 --warmup 2 --graph 0 --clip 0 --distill 0 --cpu-baseline 0 --breakdown 0 > $O/stats_bench.json 2> $O/stats_bench.err
cp /tmp/stats_step/*/*kernel_stats.csv $O/unet_bench_kernel_stats.csv
head -12 $O/unet_bench_kernel_stats.csv
