// Torch-free lab bench for t2v_gemm (saves the 1-2 minutes of `import torch` on a fresh GPU box per call).
//
//   hipcc -O2 -std=c++17 -I include tools/gemm_lab.cpp -o tools/gemm_lab -ldl
//   tools/gemm_lab spec.txt            (or `-` for stdin)
//
// Every spec line is one experiment (blank lines and #-comments are echoed):
//   lib name mode nimg h w frames cin n act res rv cfg split debug [iters]
//     lib    index into T2V_LAB_LIBS (colon-separated .so paths; default: the product library)
//     mode   T2V_GEMM_* (0 linear: M = nimg*h*w rows, 1 conv3x3, 4 tconv3)
//     act    0 none, 1 GEGLU, 2 SiLU;  res / rv: 1 = residual operand / time-embedding row vector present;
//            res 2 / 3 = residual + the LoRA branch in the epilogue (rank 64, one leaf) without / with dropout 0.1 (no reference check)
//     cfg    tile id (0 = library heuristic; 39 = t2v_conv_halo with its own choice, 40..42 = t2v_conv_halo tiles), split = split-K factor (0 = heuristic), debug = ablation bits (ablate builds)
// Output: one CSV row per experiment: name, M, N, K, cfg, split, debug, us (best of 3 runs of `iters` back-to-back launches),
// TFLOP/s, max |err| / tolerance of 256 sampled outputs against an fp64 host reference (debug == 0 only).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "t2v_hip.h"

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s failed: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

static inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Lib {
    void* h = nullptr;
    int (*gemm)(const t2v_gemm_desc*, void*) = nullptr;
    int (*halo)(const t2v_gemm_desc*, void*) = nullptr;
    int (*init)() = nullptr;
    int (*debug)(int) = nullptr;
    int (*hdebug)(int) = nullptr;
    int (*g2enable)(int) = nullptr;
    const char* (*last_error)() = nullptr;
};

static Lib load_lib(const std::string& path) {
    Lib l;
    l.h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", path.c_str(), dlerror()); exit(1); }
    l.gemm = (int (*)(const t2v_gemm_desc*, void*))dlsym(l.h, "t2v_gemm");
    l.halo = (int (*)(const t2v_gemm_desc*, void*))dlsym(l.h, "t2v_conv_halo");
    l.init = (int (*)())dlsym(l.h, "t2v_init");
    l.debug = (int (*)(int))dlsym(l.h, "t2v_gemm_debug");
    l.hdebug = (int (*)(int))dlsym(l.h, "t2v_conv_halo_debug");
    l.g2enable = (int (*)(int))dlsym(l.h, "t2v_gemm2_enable");
    l.last_error = (const char* (*)())dlsym(l.h, "t2v_last_error");
    if (!l.gemm || !l.init) { fprintf(stderr, "%s: missing symbols\n", path.c_str()); exit(1); }
    if (l.init() != 0) { fprintf(stderr, "%s: t2v_init failed\n", path.c_str()); exit(1); }
    return l;
}

struct Buf {
    void* d = nullptr;
    size_t bytes = 0;
    void need(size_t b) {
        if (b <= bytes) return;
        if (d) CHECK(hipFree(d));
        CHECK(hipMalloc(&d, b));
        bytes = b;
    }
};

int main(int argc, char** argv) {
    const char* libs_env = getenv("T2V_LAB_LIBS");
    std::vector<Lib> libs;
    {
        std::string s = libs_env ? libs_env : "t2v-turbo_amd/libt2v_hip.so";
        std::stringstream ss(s);
        std::string item;
        while (std::getline(ss, item, ':'))
            if (!item.empty()) libs.push_back(load_lib(item));
    }
    FILE* in = (argc > 1 && strcmp(argv[1], "-") != 0) ? fopen(argv[1], "r") : stdin;
    if (!in) { perror("spec"); return 1; }
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    Buf bx, bw, bw2, bo, bres, bbias, brv, bws, blt, blu, bseed;
    std::vector<uint16_t> hx, hw, hres, hout;
    std::vector<float> hbias, hrv;
    printf("name,lib,mode,M,N,K,act,res,rv,cfg,split,debug,us,tflops,err_over_tol\n");
    char line[1024];
    // the last generated problem is kept when consecutive lines share the geometry (same data for A/B rows)
    long long last_key[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int K2 = 0;
    while (fgets(line, sizeof line, in)) {
        if (line[0] == '#' || line[0] == '\n') { fputs(line, stdout); continue; }
        int lib = 0, mode, nimg, h, w, frames, cin, n, act, res, rv, cfg, split, debug, iters = 20;
        char name[128];
        const int got = sscanf(line, "%d %127s %d %d %d %d %d %d %d %d %d %d %d %d %d %d", &lib, name, &mode, &nimg, &h, &w, &frames, &cin, &n,
                               &act, &res, &rv, &cfg, &split, &debug, &iters);
        if (got < 15) { fprintf(stderr, "bad spec line: %s", line); continue; }
        if (lib < 0 || lib >= (int)libs.size()) { fprintf(stderr, "no lib %d\n", lib); continue; }
        const int taps = mode == 0 ? 1 : (mode == 4 ? 3 : 9);
        const long long rows = (long long)nimg * h * w;
        const long long M = rows;  // modes 0, 1, 4 keep the grid
        if (mode != 0 && mode != 1 && mode != 4) { fprintf(stderr, "mode %d not supported by the lab\n", mode); continue; }
        const int K = taps * cin;
        const int n_out = act == 1 ? n / 2 : n;
        const long long key[8] = {mode, nimg, h, w, frames, cin, n, act};
        if (memcmp(key, last_key, sizeof key) != 0) {
            memcpy(last_key, key, sizeof key);
            std::mt19937 rng(1234u + (unsigned)(M * 31 + n * 7 + K));
            std::uniform_real_distribution<float> U(-1.f, 1.f);
            hx.resize((size_t)rows * cin);
            for (auto& v : hx) v = f2bf(U(rng));
            hw.resize((size_t)n * K);
            const float ws = 1.0f / std::sqrt((float)K) * 1.7f;
            for (auto& v : hw) v = f2bf(U(rng) * ws);
            hres.resize((size_t)M * n_out);
            for (auto& v : hres) v = f2bf(U(rng));
            hbias.resize(n);
            for (auto& v : hbias) v = U(rng);
            hrv.resize((size_t)nimg * n);
            for (auto& v : hrv) v = U(rng);
            bx.need(hx.size() * 2); bw.need(hw.size() * 2); bres.need(hres.size() * 2); bbias.need(hbias.size() * 4); brv.need(hrv.size() * 4);
            bo.need((size_t)M * n_out * 2);
            CHECK(hipMemcpy(bx.d, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(bw.d, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
            if (mode != 0) {   // slab-major pack of the same weights for t2v_conv_halo: [N][C/32][taps][32]
                K2 = ((cin / 32 * taps + 7) / 8) * 8 * 32;   // rows zero-padded to an even number of 4-pair weight stages
                std::vector<uint16_t> hw2((size_t)n * K2, 0);
                for (int r = 0; r < n; ++r)
                    for (int t = 0; t < taps; ++t)
                        for (int c = 0; c < cin; ++c)
                            hw2[(size_t)r * K2 + ((size_t)(c / 32) * taps + t) * 32 + (c % 32)] = hw[(size_t)r * K + (size_t)t * cin + c];
                bw2.need(hw2.size() * 2);
                CHECK(hipMemcpy(bw2.d, hw2.data(), hw2.size() * 2, hipMemcpyHostToDevice));
            }
            CHECK(hipMemcpy(bres.d, hres.data(), hres.size() * 2, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(bbias.d, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(brv.d, hrv.data(), hrv.size() * 4, hipMemcpyHostToDevice));
        }
        bws.need((size_t)256 << 20);
        t2v_gemm_desc d;
        memset(&d, 0, sizeof d);
        d.a0 = bx.d; d.c0 = cin; d.lda0 = cin; d.mode = mode; d.n_img = nimg; d.h_in = h; d.w_in = w; d.frames = frames;
        d.M = (int)M; d.N = n; d.w = bw.d; d.ldw = K; d.batch = 1; d.batch_inner = 1; d.alpha = 1.0f; d.bias = (const float*)bbias.d;
        if (rv) { d.rowvec = (const float*)brv.d; d.rowvec_div = h * w; d.ld_rowvec = n; }
        if (res && act != 1) { d.residual = bres.d; d.ldr = n_out; }
        if (res >= 2) {   // LoRA epilogue operands: t [M][64], U [N][64] (values do not matter for the timing: small random)
            std::vector<uint16_t> ht((size_t)M * 64), hu((size_t)n * 64);
            std::mt19937 r2(7);
            std::uniform_real_distribution<float> U2(-0.1f, 0.1f);
            for (auto& v : ht) v = f2bf(U2(r2));
            for (auto& v : hu) v = f2bf(U2(r2));
            blt.need(ht.size() * 2); blu.need(hu.size() * 2); bseed.need(8);
            CHECK(hipMemcpy(blt.d, ht.data(), ht.size() * 2, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(blu.d, hu.data(), hu.size() * 2, hipMemcpyHostToDevice));
            const uint64_t seed = 0x1234567ull;
            CHECK(hipMemcpy(bseed.d, &seed, 8, hipMemcpyHostToDevice));
            d.lora_t = blt.d; d.ld_lora_t = 64; d.lora_u = blu.d; d.ld_lora_u = 64; d.lora_n_leaf = n; d.lora_scale = 1.0f;
            if (res == 3) {
                d.drop_seed = bseed.d; d.drop_thr = (unsigned)(0.1 * 4294967296.0); d.drop_site = 3; d.drop_inv_keep = 1.0f / 0.9f;
                d.drop_ncols = n; d.drop_col0 = 0;
            }
        }
        d.act = act; d.out = bo.d; d.ldo = n_out; d.tile_cfg = cfg == 49 ? 0 : cfg; d.split_k = split; d.ws = bws.d; d.ws_bytes = (long long)bws.bytes;
        Lib& L = libs[lib];
        if (L.debug) L.debug(debug);
        if (L.hdebug) L.hdebug(debug);
        if (L.g2enable) L.g2enable(cfg == 50 || cfg == 51 || cfg == 49);   // 49: the library's own routing between the two families
        const bool halo = cfg >= 39 && cfg < 44;
        int (*run)(const t2v_gemm_desc*, void*) = L.gemm;
        if (halo) {
            if (!L.halo || mode == 0) { fprintf(stderr, "no t2v_conv_halo for this line\n"); continue; }
            run = L.halo;
            d.w = bw2.d;
            d.ldw = K2;
            d.tile_cfg = cfg == 39 ? 0 : cfg;
        }
        CHECK(hipMemsetAsync(bo.d, 0, (size_t)M * n_out * 2, stream));
        int rc = 0;
        for (int i = 0; i < 3 && rc == 0; ++i) rc = run(&d, stream);
        if (rc != 0) {
            printf("%s,%d,%d,%lld,%d,%d,%d,%d,%d,%d,%d,%d,rc=%d %s,,\n", name, lib, mode, M, n, K, act, res, rv, cfg, split, debug, rc,
                   L.last_error ? L.last_error() : "");
            continue;
        }
        CHECK(hipStreamSynchronize(stream));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, stream));
            for (int i = 0; i < iters; ++i) run(&d, stream);
            CHECK(hipEventRecord(e1, stream));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double us = best * 1e3 / iters;
        double err_ratio = -1.0;
        if (debug == 0 && res < 2) {  // sampled fp64 reference
            hout.resize((size_t)M * n_out);
            CHECK(hipMemcpy(hout.data(), bo.d, hout.size() * 2, hipMemcpyDeviceToHost));
            std::mt19937 rng(99);
            err_ratio = 0.0;
            auto dot = [&](long long m, int ch) {
                double acc = 0.0;
                const long long hw_ = (long long)h * w;
                if (mode == 0) {
                    for (int c = 0; c < cin; ++c) acc += (double)bf2f(hx[m * cin + c]) * bf2f(hw[(size_t)ch * K + c]);
                } else if (mode == 1) {
                    const long long img = m / hw_, rem = m % hw_;
                    const int y = (int)(rem / w), x = (int)(rem % w);
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const int yy = y + ky - 1, xx = x + kx - 1;
                            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                            const uint16_t* xr = &hx[((img * h + yy) * w + xx) * cin];
                            const uint16_t* wr = &hw[(size_t)ch * K + (ky * 3 + kx) * cin];
                            for (int c = 0; c < cin; ++c) acc += (double)bf2f(xr[c]) * bf2f(wr[c]);
                        }
                } else {  // (3,1,1) over frames
                    const long long img = m / hw_, pix = m % hw_;
                    const long long clip = img / frames;
                    const int f = (int)(img % frames);
                    for (int kt = 0; kt < 3; ++kt) {
                        const int ff = f + kt - 1;
                        if (ff < 0 || ff >= frames) continue;
                        const uint16_t* xr = &hx[((clip * frames + ff) * hw_ + pix) * cin];
                        const uint16_t* wr = &hw[(size_t)ch * K + kt * cin];
                        for (int c = 0; c < cin; ++c) acc += (double)bf2f(xr[c]) * bf2f(wr[c]);
                    }
                }
                return acc + hbias[ch];
            };
            for (int s = 0; s < 256; ++s) {
                long long m = (s < 32) ? (s % 2 ? M - 1 - s : s) : (long long)(rng() % M);
                int co = (s < 32) ? (s % 2 ? n_out - 1 - (s % n_out) : s % n_out) : (int)(rng() % n_out);
                double ref;
                if (act == 1) {  // packed rows: 64-row groups [32 value | 32 gate]
                    const int grp = co / 32, j = co % 32;
                    const double v = dot(m, grp * 64 + j), g = dot(m, grp * 64 + 32 + j);
                    ref = v * 0.5 * g * (1.0 + std::erf(g * 0.7071067811865476));
                } else {
                    ref = dot(m, co);
                    if (rv) ref += hrv[(m / ((long long)h * w)) * n + co];
                    if (res) ref += bf2f(hres[m * n_out + co]);
                    if (act == 2) ref = ref / (1.0 + std::exp(-ref));
                }
                const double got = bf2f(hout[m * n_out + co]);
                const double tol = 0.02 + 0.01 * std::fabs(ref);
                const double r = std::fabs(got - ref) / tol;
                if (r > err_ratio) err_ratio = r;
            }
        }
        printf("%s,%d,%d,%lld,%d,%d,%d,%d,%d,%d,%d,%d,%.2f,%.1f,%.3f\n", name, lib, mode, M, n, K, act, res, rv, cfg, split, debug, us,
               2.0 * M * n * K / us / 1e6, err_ratio);
        fflush(stdout);
    }
    return 0;
}
