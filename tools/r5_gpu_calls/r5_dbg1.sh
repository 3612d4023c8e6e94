set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5dbg1
mkdir -p $O
cd $R
timeout 600 python -m pytest -q -m gpu -s "tests/test_gpu_train_parity.py::test_overlapped_gradient_exchange_over_rccl_one_rank" > $O/alone.txt 2>&1; echo "alone rc=$?"; tail -5 $O/alone.txt | cut -c1-300
timeout 900 python -m pytest -q -m gpu -s tests/test_gpu_train_parity.py -k "native_checkpointing or overlapped or trainer_route" > $O/three.txt 2>&1; echo "three rc=$?"; tail -5 $O/three.txt | cut -c1-300
