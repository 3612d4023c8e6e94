// Tile ids 24-29 of t2v_gemm (4-wave 256x256 with 128x128 wave tiles; register-staged operand path): the same kernel template
// as gemm.hip, instantiated in a translation unit of their own - see the note above t2v_gemm_launch_experimental in gemm.hip.
#define T2V_GEMM_EXP_ONLY
#include "gemm.hip"
