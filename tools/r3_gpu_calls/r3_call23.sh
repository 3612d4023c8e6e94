#!/bin/bash
# round 3, call 23 (the last 100 s of the budget): the gradient engine with the LoRA branch in the base leaf's epilogue
# (T2V_LORA_EPILOGUE=1) on the device against the reference's own LoRA gradients and with replayed train-mode masks
T2V_LORA_EPILOGUE=1 timeout 75 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -x -s -k "tiny_student_on_device or train_mode_student" 2>&1 | grep -E "fixture|train mode|passed|failed|Error|assert" | tail -6 | cut -c1-260
