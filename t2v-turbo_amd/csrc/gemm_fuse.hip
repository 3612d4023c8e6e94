// The fast t2v_gemm kernels with fused normalisation statistics (row statistics for the next LayerNorm, column statistics per
// 32-row slab for the next GroupNorm, a LayerNorm folded into the consuming GEMM): the same kernel template as gemm.hip,
// instantiated in a translation unit of their own so that the code of the validated kernels does not depend on them — see the
// note above t2v_gemm_launch_fused in gemm.hip.
#define T2V_GEMM_FUSE_ONLY
#include "gemm.hip"
