// selftest.hip -- dumps the lane maps of the hardware primitives the kernels rely on, so a GPU test can
// assert them (tests/test_gpu_layouts.py): MFMA 32x32 C/D layout (bf16 and f32 forms) and the
// ds_read_b64_tr_b16 gather.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace {
__global__ __launch_bounds__(64) void k_selftest(int32_t* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    const int l = threadIdx.x, h = l >> 5, i = l & 31;
    // (a) bf16 MFMA: A[i][k*] = i + 1, B[k*][j] = j + 1 on one k-slot -> D[i][j] = (i+1)(j+1)
    {
        Frag<bf16> a, b;
        a.v[0] = a.v[1] = b.v[0] = b.v[1] = u32x4{0u, 0u, 0u, 0u};
        if (h == 0) {
            bf16 x = (bf16)(float)(i + 1);
            uint32_t bits = (uint32_t)__builtin_bit_cast(uint16_t, x);
            a.v[0][0] = bits;
            b.v[0][0] = bits;
        }
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        mtl_mma(a, b, c);
        for (int r = 0; r < 16; ++r) out[l * 16 + r] = (int32_t)c[r];
    }
    // (b) f32 MFMA
    {
        Frag<float> a, b;
        a.v[0] = a.v[1] = b.v[0] = b.v[1] = u32x4{0u, 0u, 0u, 0u};
        if (h == 0) {
            a.v[0][0] = __builtin_bit_cast(uint32_t, (float)(i + 1));
            b.v[0][0] = a.v[0][0];
        }
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        mtl_mma(a, b, c);
        for (int r = 0; r < 16; ++r) out[1024 + l * 16 + r] = (int32_t)c[r];
    }
    // (c) ds_read_b64_tr_b16 with the linear address pattern lane -> 8 bytes at 8*lane
    for (int x = l; x < 1024; x += 64) lds[x] = (short)x;
    __syncthreads();
    {
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 4));
        for (int e = 0; e < 4; ++e) out[2048 + l * 4 + e] = (int32_t)v[e];
    }
    // (d) the same with the [row][stride 72] block pattern used by the kernels:
    //     lane i of a 16-lane group addresses row (i>>2), cols 4*(i&3) of the group's block
    {
        const int g = l >> 4, ig = l & 15;
        const int row = 2 * g + (ig >> 2);          // arbitrary distinct blocks per group
        const int col = 16 * (g & 1) + 4 * (ig & 3);
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(lds + row * 72 + col));
        for (int e = 0; e < 4; ++e) out[2304 + l * 4 + e] = (int32_t)v[e];
    }
}
}  // namespace

extern "C" int mtlora_selftest_layouts(int32_t* out, void* stream) {
    if (!out) return MTLORA_ERR_NULL;
    hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

// ---------------------------------------------------------------------------------------------
// profiling facility
// ---------------------------------------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <vector>
namespace {
struct ProfRec {
    int kind;
    double bytes, s8d, flops;
    hipEvent_t a, b;
    char tag[56];
};
thread_local char g_next_tag[56] = {0};
std::atomic<int> g_prof_on{0};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
size_t g_prof_cap = 0;
}  // namespace

int mtl_prof_start(int kind, double alg_bytes, hipStream_t s, double s8d_bytes, double flops) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return -1;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.size() >= g_prof_cap) return -1;
    ProfRec r;
    r.kind = kind;
    r.bytes = alg_bytes;
    r.s8d = s8d_bytes;
    r.flops = flops;
    memcpy(r.tag, g_next_tag, sizeof(r.tag));
    g_next_tag[0] = 0;
    if (hipEventCreate(&r.a) != hipSuccess) return -1;
    if (hipEventCreate(&r.b) != hipSuccess) {
        (void)hipEventDestroy(r.a);
        return -1;
    }
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}
void mtl_prof_tag(const char* fmt, ...) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_next_tag, sizeof(g_next_tag), fmt, ap);
    va_end(ap);
}
void mtl_prof_stop(int idx, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (idx >= 0 && (size_t)idx < g_prof.size()) (void)hipEventRecord(g_prof[idx].b, s);
}

extern "C" int mtlora_prof_begin(int max_records) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    g_prof_cap = max_records > 0 ? (size_t)max_records : 0;
    g_prof.reserve(g_prof_cap);
    g_prof_on.store(1);
    return MTLORA_OK;
}

extern "C" int mtlora_prof_end(mtlora_prof_summary* out) {
    g_prof_on.store(0);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (out) {
        for (int k = 0; k < MTLORA_PROF_KINDS; ++k) {
            out->count[k] = 0;
            out->ms[k] = 0.0;
            out->alg_bytes[k] = 0.0;
            out->s8d_bytes[k] = 0.0;
            out->flops[k] = 0.0;
        }
    }
    int st = MTLORA_OK;
    const char* dump = getenv("MTLORA_PROF_DUMP");  // developer aid: one line per record (kind, alg bytes, ms, shape note)
    FILE* df = dump && dump[0] ? fopen(dump, "w") : nullptr;
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) st = MTLORA_ERR_HIP;
        if (df) fprintf(df, "%s,%.0f,%.0f,%.0f,%.4f,%s\n", mtlora_prof_kind_name(r.kind), r.bytes, r.s8d, r.flops, ms, r.tag);
        if (out && r.kind >= 0 && r.kind < MTLORA_PROF_KINDS) {
            out->count[r.kind] += 1;
            out->ms[r.kind] += ms;
            out->alg_bytes[r.kind] += r.bytes;
            out->s8d_bytes[r.kind] += r.s8d;
            out->flops[r.kind] += r.flops;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    if (df) fclose(df);
    g_prof.clear();
    return st;
}

extern "C" const char* mtlora_prof_kind_name(int kind) {
    switch (kind) {
        case PK_NT_FWD_MAIN: return "k_nt:fwd_outputs";
        case PK_NT_FWD_P: return "k_nt:fwd_lowrank_P";
        case PK_NT_BWD_Q: return "k_nt:bwd_lowrank_Q";
        case PK_NT_BWD_DX: return "k_nt:bwd_dX";
        case PK_TN: return "k_tn:dA_dB";
        case PK_ATTN_FWD: return "k_attn_fwd";
        case PK_ATTN_BWD: return "k_attn_bwd";
        case PK_PACK: return "k_pack";
        case PK_REDUCE: return "k_tn_reduce";
        case PK_WINDOW: return "k_window_process";
        case PK_LN_FWD: return "k_ln_fwd";
        case PK_LN_BWD: return "k_ln_bwd";
        case PK_BN: return "k_bn";
        case PK_RESIDUAL: return "k_residual";
        case PK_LOSS: return "k_up_loss";
        case PK_SUM: return "k_sum";
        case PK_UPSAMPLE: return "k_upsample";
        case PK_NT_PLAIN_FWD: return "k_nt:plain_fwd";
        case PK_NT_PLAIN_DX: return "k_nt:plain_dX";
        case PK_TN_PLAIN: return "k_tn:plain_dW";
        default: return "";
    }
}
