// Torch-free lab bench: t2v_linear_pr (csrc/linear_pr.hip) against t2v_gemm on the same operands.
//
//   hipcc -O2 -std=c++17 -I include tools/linear_lab.cpp -o tools/linear_lab -ldl
//   T2V_LAB_LIBS=lib0.so[:lib1.so...] tools/linear_lab spec.txt      (default library: t2v-turbo_amd/libt2v_hip.so)
//
// Spec lines (blank lines and #-comments are echoed):
//   lib name M K N act res gemm_cfg ny debug [iters]
//     act 0 none / 1 GEGLU; res 1 = residual operand; gemm_cfg = tile id for the t2v_gemm side (0 = library heuristic, -1 = skip t2v_gemm);
//     ny = column splits of t2v_linear_pr (0 = library rule); debug = t2v_linear_pr ablation bits (ablate builds)
// Output per line: name, M, K, N, us_gemm, TF_gemm, us_lpr, TF_lpr, max |lpr - gemm| over the whole output, max |lpr - fp64 reference| / tolerance
// over 512 sampled outputs (debug == 0 only).  Times: best of 3 runs of `iters` back-to-back launches between two events.
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
    int (*lpr)(const t2v_gemm_desc*, void*) = nullptr;
    int (*lpr_ok)(const t2v_gemm_desc*) = nullptr;
    int (*lpr_debug)(int) = nullptr;
    int (*lpr_split)(int) = nullptr;
    int (*init)() = nullptr;
    const char* (*last_error)() = nullptr;
};

static Lib load_lib(const std::string& path) {
    Lib l;
    l.h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", path.c_str(), dlerror()); exit(1); }
    l.gemm = (int (*)(const t2v_gemm_desc*, void*))dlsym(l.h, "t2v_gemm");
    l.lpr = (int (*)(const t2v_gemm_desc*, void*))dlsym(l.h, "t2v_linear_pr");
    l.lpr_ok = (int (*)(const t2v_gemm_desc*))dlsym(l.h, "t2v_linear_pr_supported");
    l.lpr_debug = (int (*)(int))dlsym(l.h, "t2v_linear_pr_debug");
    l.lpr_split = (int (*)(int))dlsym(l.h, "t2v_linear_pr_force_split");
    l.init = (int (*)())dlsym(l.h, "t2v_init");
    l.last_error = (const char* (*)())dlsym(l.h, "t2v_last_error");
    if (!l.gemm || !l.lpr || !l.init) { fprintf(stderr, "%s: missing symbols\n", path.c_str()); exit(1); }
    if (l.init() != 0) { fprintf(stderr, "%s: t2v_init failed\n", path.c_str()); exit(1); }
    return l;
}

template <class T>
static T* to_dev(const std::vector<T>& v) {
    T* d = nullptr;
    CHECK(hipMalloc(&d, v.size() * sizeof(T)));
    CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: linear_lab spec.txt|-\n"); return 2; }
    std::vector<Lib> libs;
    {
        const char* env = getenv("T2V_LAB_LIBS");
        std::string s = env ? env : "t2v-turbo_amd/libt2v_hip.so";
        std::stringstream ss(s);
        std::string item;
        while (std::getline(ss, item, ':')) if (!item.empty()) libs.push_back(load_lib(item));
    }
    FILE* f = strcmp(argv[1], "-") == 0 ? stdin : fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 2; }
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    void* ws = nullptr;
    const long long ws_bytes = 96ll << 20;
    CHECK(hipMalloc(&ws, ws_bytes));
    printf("name,M,K,N,act,res,gemm_cfg,ny,debug,us_gemm,tf_gemm,us_lpr,tf_lpr,max_abs_diff_vs_gemm,ref_err_over_tol\n");
    char line[1024];
    while (fgets(line, sizeof line, f)) {
        if (line[0] == '#' || line[0] == '\n') { fputs(line, stdout); continue; }
        int li, M, K, N, act, res, gcfg, ny, debug, iters = 20;
        char name[128];
        const int n = sscanf(line, "%d %127s %d %d %d %d %d %d %d %d %d", &li, name, &M, &K, &N, &act, &res, &gcfg, &ny, &debug, &iters);
        if (n < 10) { fprintf(stderr, "bad spec line: %s", line); continue; }
        if (li < 0 || li >= (int)libs.size()) { fprintf(stderr, "lib index %d out of range\n", li); continue; }
        Lib& L = libs[li];
        const int n_out = act == 1 ? N / 2 : N;
        std::mt19937 rng(1234 + M + N);
        std::uniform_real_distribution<float> ud(-1.f, 1.f);
        std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K), hWp((size_t)N * K), hR((size_t)M * n_out);
        std::vector<float> hB(N);
        for (auto& v : hA) v = f2bf(ud(rng));
        const float wsc = 1.0f / sqrtf((float)K) * 1.7f;
        for (auto& v : hW) v = f2bf(ud(rng) * wsc);
        for (auto& v : hB) v = ud(rng) * 0.5f;
        for (auto& v : hR) v = f2bf(ud(rng));
        // fragment pack (include/t2v_hip.h, t2v_linear_pr)
        for (int q = 0; q < N / 64; ++q)
            for (int s = 0; s < K / 16; ++s)
                for (int b = 0; b < 2; ++b)
                    for (int l = 0; l < 64; ++l) {
                        const int i = l & 31;
                        const int row = 64 * q + 32 * b + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
                        const int col = 16 * s + 8 * (l >> 5);
                        memcpy(&hWp[((((size_t)q * (K / 16) + s) * 2 + b) * 64 + l) * 8], &hW[(size_t)row * K + col], 16);
                    }
        uint16_t *dA = to_dev(hA), *dW = to_dev(hW), *dWp = to_dev(hWp), *dR = to_dev(hR);
        float* dB = to_dev(hB);
        uint16_t *o1 = nullptr, *o2 = nullptr;
        CHECK(hipMalloc(&o1, (size_t)M * n_out * 2));
        CHECK(hipMalloc(&o2, (size_t)M * n_out * 2));
        CHECK(hipMemset(o1, 0xff, (size_t)M * n_out * 2));
        CHECK(hipMemset(o2, 0xff, (size_t)M * n_out * 2));
        t2v_gemm_desc d;
        memset(&d, 0, sizeof d);
        d.a0 = dA; d.c0 = K; d.lda0 = K; d.mode = T2V_GEMM_LINEAR; d.M = M; d.N = N; d.w = dW; d.ldw = K;
        d.batch = 1; d.batch_inner = 1; d.alpha = 1.0f; d.bias = dB; d.residual = res ? dR : nullptr; d.ldr = n_out; d.act = act;
        d.out = o1; d.ldo = n_out; d.tile_cfg = gcfg > 0 ? gcfg : 0; d.ws = ws; d.ws_bytes = ws_bytes;
        t2v_gemm_desc d2 = d;
        d2.w = dWp; d2.out = o2; d2.tile_cfg = 0;
        auto time_it = [&](auto&& launch) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                launch();
                CHECK(hipStreamSynchronize(stream));
                CHECK(hipEventRecord(e0, stream));
                for (int i = 0; i < iters; ++i) launch();
                CHECK(hipEventRecord(e1, stream));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms * 1000.f / iters);
            }
            return best;
        };
        float us_g = 0.f, us_l = 0.f;
        if (gcfg >= 0) {
            us_g = time_it([&] {
                const int rc = L.gemm(&d, stream);
                if (rc != 0) { fprintf(stderr, "t2v_gemm rc %d: %s\n", rc, L.last_error ? L.last_error() : ""); exit(1); }
            });
        }
        if (L.lpr_debug) L.lpr_debug(debug);
        if (L.lpr_split) L.lpr_split(ny);
        const int ok = L.lpr_ok ? L.lpr_ok(&d2) : 1;
        if (ok == 1) {
            us_l = time_it([&] {
                const int rc = L.lpr(&d2, stream);
                if (rc != 0) { fprintf(stderr, "t2v_linear_pr rc %d: %s\n", rc, L.last_error ? L.last_error() : ""); exit(1); }
            });
        }
        if (L.lpr_debug) L.lpr_debug(0);
        if (L.lpr_split) L.lpr_split(0);
        double max_diff = -1.0, ref_ratio = -1.0;
        if (ok == 1 && debug == 0) {
            std::vector<uint16_t> h1((size_t)M * n_out), h2((size_t)M * n_out);
            CHECK(hipMemcpy(h2.data(), o2, h2.size() * 2, hipMemcpyDeviceToHost));
            if (gcfg >= 0) {
                CHECK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost));
                max_diff = 0.0;
                for (size_t i = 0; i < h1.size(); ++i) {
                    const double dd = fabs((double)bf2f(h1[i]) - (double)bf2f(h2[i]));
                    if (!(dd <= max_diff)) max_diff = dd;   // (NaN propagates)
                }
            }
            std::uniform_int_distribution<int> rm(0, M - 1), rn(0, n_out - 1);
            ref_ratio = 0.0;
            for (int sidx = 0; sidx < 512; ++sidx) {
                const int m = sidx < 8 ? (sidx < 4 ? sidx : M - 1 - (sidx - 4)) : rm(rng), c = rn(rng);
                auto dot = [&](int row) {
                    double acc = hB[row];
                    for (int k = 0; k < K; ++k) acc += (double)bf2f(hA[(size_t)m * K + k]) * (double)bf2f(hW[(size_t)row * K + k]);
                    return acc;
                };
                double want;
                if (act == 1) {
                    const int grp = c / 32, j = c % 32;
                    const double v = dot(64 * grp + j), g = dot(64 * grp + 32 + j);
                    want = v * 0.5 * g * (1.0 + erf(g * 0.70710678118654752));
                } else {
                    want = dot(c) + (res ? (double)bf2f(hR[(size_t)m * n_out + c]) : 0.0);
                }
                const double got = bf2f(h2[(size_t)m * n_out + c]);
                const double tol = 1e-2 * fabs(want) + 2e-2;
                const double r = fabs(got - want) / tol;
                if (!(r <= ref_ratio)) ref_ratio = r;
            }
        }
        if (getenv("LPR_TRACE") && ok == 1) {   // a -DT2V_LPR_TRACE library: one more launch on a zeroed workspace, then the stamps of three workgroups
            CHECK(hipMemset(ws, 0, 3 * 8 * 32 * 8 + 4096 * 32));
            L.lpr(&d2, stream);
            CHECK(hipStreamSynchronize(stream));
            std::vector<unsigned long long> tr(3 * 8 * 32);
            CHECK(hipMemcpy(tr.data(), ws, tr.size() * 8, hipMemcpyDeviceToHost));
            {   // every workgroup: duration (ticks) by XCC
                std::vector<unsigned long long> wt(4096 * 4);
                CHECK(hipMemcpy(wt.data(), (char*)ws + 3 * 8 * 32 * 8, wt.size() * 8, hipMemcpyDeviceToHost));
                printf("# wg durations %s (ticks; xcc:duration:start offset within its xcc):", name);
                unsigned long long x0[16];
                for (auto& v : x0) v = ~0ull;
                for (int i = 0; i < 4096 && wt[i * 4]; ++i) { const unsigned xcc = (unsigned)wt[i * 4 + 3] & 15; if (wt[i * 4] < x0[xcc]) x0[xcc] = wt[i * 4]; }
                for (int i = 0; i < 4096 && wt[i * 4]; ++i) { const unsigned xcc = (unsigned)wt[i * 4 + 3] & 15; printf(" %u:%llu:%llu", xcc, wt[i * 4 + 1] - wt[i * 4], wt[i * 4] - x0[xcc]); }
                printf("\n");
            }
            unsigned long long t0 = ~0ull;
            for (int i = 0; i < 3 * 8; ++i) if (tr[i * 32] && tr[i * 32] < t0) t0 = tr[i * 32];
            for (int i = 0; i < 3 * 8; ++i) {
                const unsigned hw = (unsigned)tr[i * 32 + 31];
                printf("# trace %s wg-slot %d wave %d simd %u cu %u se %u:", name, i / 8, i % 8, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7);
                for (int k2 = 0; k2 < 31 && tr[i * 32 + k2]; ++k2) printf(" %llu", tr[i * 32 + k2] - t0);
                printf("\n");
            }
        }
        const double flop = 2.0 * M * N * K;
        printf("%s,%d,%d,%d,%d,%d,%d,%d,%d,%.2f,%.1f,%.2f,%.1f,%.4g,%.3f\n", name, M, K, N, act, res, gcfg, ny, debug, us_g,
               us_g > 0 ? flop / us_g * 1e-6 : 0.0, us_l, us_l > 0 ? flop / us_l * 1e-6 : 0.0, max_diff, ref_ratio);
        fflush(stdout);
        CHECK(hipFree(dA)); CHECK(hipFree(dW)); CHECK(hipFree(dWp)); CHECK(hipFree(dR)); CHECK(hipFree(dB)); CHECK(hipFree(o1)); CHECK(hipFree(o2));
    }
    return 0;
}
