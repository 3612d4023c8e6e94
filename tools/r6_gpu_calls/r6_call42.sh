#!/bin/bash
# round 6, call 42: counters on t2v_wgrad_tn's two output tiles at three base-weight gradient shapes (separate --pmc passes, kernel-trace only)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c42
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
args=""
for t in 0 1; do
  i=0
  for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rm -rf /tmp/pw
    T2V_WGRAD_TILE128=$t timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pw -- python $R/tools/wgrad_full_pmc_target.py > $O/pmc_${t}_$i.log 2>&1
    f=$(find /tmp/pw -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then cp $f $O/cc_${t}_$i.csv; args="$args tile$((64+64*t))=$O/cc_${t}_$i.csv"; else echo "pass $t/$i: no counter file"; tail -3 $O/pmc_${t}_$i.log; fi
  done
done
python $R/tools/pmc_table.py $args > $O/wgrad_tile_pmc.csv
rm -f $O/cc_*.csv
grep -c . $O/wgrad_tile_pmc.csv; grep "wgrad_tn_kernel" $O/wgrad_tile_pmc.csv | head -60
