#!/bin/bash
# round 5, call 19 (the last GPU seconds): device tests that run an inference engine at widths where the one-launch GroupNorm form is
# taken and that call 17 did not cover (mid-width reference-gradient fixture, pipeline, frozen train-mode teacher, DDIM inversion, trainer route)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c19
mkdir -p $O
cd $R
timeout 150 python -m pytest -v -m gpu \
  "tests/test_gpu_train_parity.py::test_mid_width_student_on_device_vs_the_reference_lora_gradient_fixture" \
  "tests/test_gpu_train_parity.py::test_tiny_student_on_device_vs_the_reference_lora_gradient_fixture" \
  "tests/test_gpu_engine.py::test_pipeline_on_gpu_vs_reference_pipeline_fixture" \
  "tests/test_gpu_engine.py::test_train_mode_frozen_teacher_on_device_no_warning_and_mask_replay" \
  "tests/test_gpu_engine.py::test_ddim_inversion_and_motion_prior_score_on_device" \
  "tests/test_gpu_engine.py::test_alternating_input_signatures_keep_their_plans" \
  "tests/test_gpu_train_parity.py::test_trainer_route_two_distill_steps_with_an_optimizer_update_between" \
  "tests/test_gpu_train_parity.py::test_train_mode_student_on_device_with_replayed_masks" \
  > $O/tests.txt 2>&1
grep -E "PASSED|FAILED|ERROR|passed|failed" $O/tests.txt | cut -c1-200 | tail -12
