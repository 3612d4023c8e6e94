#!/bin/bash
# round 3, call 13: per-shape time of every t2v_gemm launch of the full-size student's forward / backward lists
set -u
mkdir -p gpurun_out/r3c13
timeout 900 python tools/student_gemm_profile.py > gpurun_out/r3c13/student_gemm_shapes.csv 2> gpurun_out/r3c13/err.txt
tail -2 gpurun_out/r3c13/err.txt | cut -c1-300
head -45 gpurun_out/r3c13/student_gemm_shapes.csv; tail -1 gpurun_out/r3c13/student_gemm_shapes.csv
