// Launch parameters / entry points of the second t2v_gemm kernel family (gemm2.hip), shared with the router in gemm.hip.
#pragma once
#include "common.h"

struct Gemm2Params {
    t2v_gemm_desc d;
    int tiles_m, tiles_n;
    int taps, nsub;        // K = taps * nsub pairs (nsub = channels / 32 per tap)
    int nq, nstage;
    int tap_stride;        // TCONV3: rows between consecutive frames (H*W); 0 for LINEAR
    int frames;
    int xcd_m, xcd_n, nblk;
    int blk_start[8], blk_r0[8], blk_c0[8], blk_w[8];
};

// cfg > 0: the family takes the launch (tile id 49 + cfg); forced: 1 / 2 = tile 50 / 51 asked for, 0 = the library's own rule
int t2v_gemm2_prepare(const t2v_gemm_desc* dd, Gemm2Params& p, int forced, int& cfg);
int t2v_gemm2_dispatch(int cfg, Gemm2Params& p, hipStream_t s);
