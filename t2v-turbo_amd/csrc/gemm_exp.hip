// Tile ids 24-29 of t2v_gemm (4-wave 256x256 with 128x128 wave tiles; register-staged operand path): the same kernel template
// as gemm.hip, instantiated in a translation unit of their own - see the note above t2v_gemm_launch_experimental in gemm.hip.
// They never win on an inference shape (profiles/r02_gemm_experimental_cfgs.csv), but the tuned table (gemm_tune.json) does pick the
// register-staged ids 25-29 for 23 of the training path's K = 64 / N = 64 LoRA shapes, so they ship; only id 24 is unused.
#define T2V_GEMM_EXP_ONLY
#include "gemm.hip"
