// t2v_replay: run a recorded list of C-ABI launches from ONE call.
//
// The engines record a forward / backward once (function + arguments per launch) and replay it on later calls.  Replayed from
// Python that is one ctypes call per launch — 8 800 calls and ~230 ms of host time per distillation step (round 3: plain replay
// was paced by the host on every box).  Here the recorded list is a flat array of 64-bit words
//     [function id, argument count n, n argument slots] ...
// and the library walks it itself: ~0.1 us of host time per launch instead of ~26 us.  A slot holds an integer or pointer as
// is, a float as its 32 bits; each entry point has a generated thunk that reads its slots with the types of ITS signature (the
// stream, always the last parameter, is the one given to t2v_replay).  Host code only: nothing here runs on the device.
#include "common.h"

#include <cstring>
#include <tuple>
#include <type_traits>
#include <utility>

namespace {

template <class T>
inline T slot_as(uint64_t v) {
    if constexpr (std::is_pointer_v<T>) {
        return (T)(uintptr_t)v;
    } else if constexpr (std::is_same_v<T, float>) {
        const uint32_t b = (uint32_t)v;
        float f;
        memcpy(&f, &b, 4);
        return f;
    } else if constexpr (std::is_same_v<T, double>) {
        double d;
        memcpy(&d, &v, 8);
        return d;
    } else {
        return (T)(int64_t)v;
    }
}

template <auto Fn>
struct Thunk;
template <class... A, int (*Fn)(A...)>
struct Thunk<Fn> {
    static constexpr size_t N = sizeof...(A);   // incl. the stream
    template <size_t I>
    static auto arg(const uint64_t* s, void* stream) {
        using T = std::tuple_element_t<I, std::tuple<A...>>;
        if constexpr (I + 1 == N) return (T)stream;
        else return slot_as<T>(s[I]);
    }
    template <size_t... I>
    static int go(const uint64_t* s, void* stream, std::index_sequence<I...>) { return Fn(arg<I>(s, stream)...); }
    static int call(const uint64_t* s, int n, void* stream) {
        if (n + 1 != (int)N) return T2V_EINVAL;
        return go(s, stream, std::make_index_sequence<N>{});
    }
};

struct Entry { const char* name; int (*call)(const uint64_t*, int, void*); };
#define T2V_ENTRY(f) {#f, &Thunk<&f>::call}
// every launching entry point of include/t2v_hip.h (stream last)
const Entry kTable[] = {
    T2V_ENTRY(t2v_gemm), T2V_ENTRY(t2v_conv_halo), T2V_ENTRY(t2v_linear_pr),
#ifdef T2V_EXPERIMENTAL
    T2V_ENTRY(t2v_ffn_fused),
#endif
    T2V_ENTRY(t2v_conv3x3_small_cin), T2V_ENTRY(t2v_gn_stats),
    T2V_ENTRY(t2v_gn_apply), T2V_ENTRY(t2v_group_norm), T2V_ENTRY(t2v_group_norm_cs), T2V_ENTRY(t2v_gn_stats_cs), T2V_ENTRY(t2v_layernorm),
    T2V_ENTRY(t2v_softmax_rows), T2V_ENTRY(t2v_attn_spatial), T2V_ENTRY(t2v_attn_temporal), T2V_ENTRY(t2v_ncfhw_to_tokens),
    T2V_ENTRY(t2v_tokens_to_ncfhw), T2V_ENTRY(t2v_timestep_embedding), T2V_ENTRY(t2v_silu), T2V_ENTRY(t2v_fill_zero), T2V_ENTRY(t2v_cast),
    T2V_ENTRY(t2v_lincomb3), T2V_ENTRY(t2v_lcm_step), T2V_ENTRY(t2v_gn_bwd), T2V_ENTRY(t2v_softmax_bwd_rows), T2V_ENTRY(t2v_transpose_bf16),
    T2V_ENTRY(t2v_sumpool2x2), T2V_ENTRY(t2v_gn_bwd2), T2V_ENTRY(t2v_layernorm_bwd), T2V_ENTRY(t2v_geglu_fwd), T2V_ENTRY(t2v_geglu_bwd),
    T2V_ENTRY(t2v_scatter2x), T2V_ENTRY(t2v_add_bf16), T2V_ENTRY(t2v_attn_temporal_bwd), T2V_ENTRY(t2v_adamw_step), T2V_ENTRY(t2v_ema_update),
    T2V_ENTRY(t2v_sumsq), T2V_ENTRY(t2v_gather_f32), T2V_ENTRY(t2v_attn_spatial_bwd), T2V_ENTRY(t2v_wgrad_tn), T2V_ENTRY(t2v_wgrad_tn_group),
    T2V_ENTRY(t2v_transpose_pad_bf16), T2V_ENTRY(t2v_dropout_bf16), T2V_ENTRY(t2v_im2col_bf16), T2V_ENTRY(t2v_gn_coef_cs), T2V_ENTRY(t2v_norm_affine_grad),
};
constexpr int kEntries = (int)(sizeof(kTable) / sizeof(kTable[0]));

}  // namespace

extern "C" int t2v_replay_lookup(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < kEntries; ++i)
        if (strcmp(kTable[i].name, name) == 0) return i;
    return -1;
}


extern "C" int t2v_replay(const unsigned long long* prog, long long nwords, void* stream, int* failed_index) {
    T2V_REQUIRE(prog && nwords >= 0, T2V_EINVAL, "t2v_replay: null program");
    long long i = 0;
    int idx = 0;
    while (i < nwords) {
        T2V_REQUIRE(i + 2 <= nwords, T2V_EINVAL, "t2v_replay: truncated program");
        const long long fn = (long long)prog[i], n = (long long)prog[i + 1];
        T2V_REQUIRE(fn >= 0 && fn < kEntries && n >= 0 && n <= 64 && i + 2 + n <= nwords, T2V_EINVAL, "t2v_replay: malformed entry");
        const int rc = kTable[fn].call((const uint64_t*)(prog + i + 2), (int)n, stream);
        if (rc != T2V_OK) {
            if (failed_index) *failed_index = idx;
            return rc;
        }
        i += 2 + n;
        ++idx;
    }
    if (failed_index) *failed_index = -1;
    return T2V_OK;
}
