// pq.h -- k_pq: the P = alpha D(X) A^T / Q = alpha dY B passes of a layer whose projection rows (R x K) do not fit in LDS and whose row
// count gives the tiled kernels too few workgroups (Swin-B stage 2 at rank 128: M = 12 544 -> 98 tiles of 128 rows on 256 CUs).
//
// What was wrong with k_ntl / k_sp_projk there (round 4, tools/projk_force.sh): both are LATENCY chains at one workgroup per CU --
// k_ntl stages one 96-wide k-tile ahead (2.1 us per k-tile: 21 us at K = 512, 47 us at K = 2048), k_sp_projk re-reads all R x K
// projection rows per 32-row item (5 x the activation bytes through a 27 GB/s per-CU fill path).  The fill-rate model of DESIGN 4.1d says
// what a launch needs: per CU, (TM + R) x K x 2 bytes at ~27 GB/s with enough bytes in flight to cover ~2 us of latency (>= 54 KB).
//
// k_pq: one workgroup (4 waves) per tile of TM = 64 or 128 rows x TN = 64 or 128 rank columns; 32-wide k-tiles stream through a
// ring of NST stages by LDS-DMA with NST - 1 stages (>= 60 KB) in flight; rows are dense 64-byte lines with the 16-byte chunks permuted
// on the source side (chunk ^ ((row >> 2) & 3), as k_nte); the dropout keep-mask of the P pass is applied to the activation fragments
// after the LDS read; the tile leaves through a per-wave LDS image as whole 128 / 256-byte row segments.  Single source, all R columns
// (layers without task inputs; the Q pass of layers without task outputs).
#pragma once

struct PqSrc {
    const void* act;     // (M x K) contiguous rows
    int col_lo, col_hi;  // columns of Out (= rows of wproj) this source owns (multiples of 8)
    int mask, pad_;      // mask: dropout keep-mask on the activation (keyed by (m, k))
};
struct PqParams {
    const void* wproj;  // (R x K) alpha-scaled projection rows
    void* out;          // (M x ld_out)
    int64_t ld_out;
    int M, K, R, n_src;
    DropoutCfg drop;
    PqSrc src[MAXO];    // blockIdx.z = source (P: x and the tasks' own inputs; Q: the outputs' gradients)
};

template <int N>
__device__ __forceinline__ void pq_wait_vm(int n) {  // s_waitcnt vmcnt(n * N), n wave-uniform in [0, 7]
    switch (n) {
        case 0: SP_WAIT_VM(0); break;
        case 1: SP_WAIT_VM(N); break;
        case 2: SP_WAIT_VM(2 * N); break;
        case 3: SP_WAIT_VM(3 * N); break;
        case 4: SP_WAIT_VM(4 * N); break;
        case 5: SP_WAIT_VM(5 * N); break;
        case 6: SP_WAIT_VM(6 * N); break;
        default: SP_WAIT_VM(7 * N); break;
    }
}

// WM = row waves (32 rows each): 4 -> TM = 128, every wave walks both k-steps of a k-tile; 2 -> TM = 64, the two waves of a row block split
// the k-steps (wk = 0 / 1) and add their accumulators through LDS at the end.  Either way a wave owns 32 rows x all NB column blocks, so
// every activation fragment is read -- and, in the P pass, hashed against the dropout mask -- by exactly ONE wave (a 2 x 2 wave grid hashed
// each fragment twice: 60 VALU instructions per fragment made the masked pass VALU-bound at 1.6 x the unmasked one).
// KSPLIT = false at WM = 2: the two waves of a row block split the COLUMN blocks instead (both walk both k-steps, no reduction): the
// unmasked Q passes measure 20 - 25 % faster that way (tools/pq_times.py), the masked P passes slower (every fragment hashed twice).
template <typename T, int WM, int NB, int NST, bool KSPLIT>
__global__ __launch_bounds__(256) void k_pq(const PqParams P) {
    constexpr bool SPLITN = WM == 2 && !KSPLIT;
    constexpr int NBW = SPLITN ? NB / 2 : NB;  // column blocks per wave
    constexpr int WK = KSPLIT ? 4 / WM : 1, TM = 32 * WM, TN = 32 * NB, ROWS = TN + TM, STAGE = ROWS * 64, DPW = ROWS / 64, KE = 32;
    constexpr int ORS = NBW * 64 + 8, CPR = NBW * 4;  // output image: row stride (bytes), 16-byte chunks per row
    static_assert((WM == 2 || WM == 4) && (WM == 2 || !KSPLIT) && NB % 2 == 0, "wave grid");
    static_assert(ROWS % 64 == 0 && NST >= 3 && NST <= 9 && (NST - 2) * DPW <= 63, "stage geometry / vmcnt range");
    static_assert(NST * STAGE >= 2 * NB * 16 * 256 && NST * STAGE >= 4 * 32 * ORS, "the ring holds the reduction and output images");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wk = KSPLIT ? wave / WM : 0, wn = SPLITN ? wave / WM : 0, h = lane >> 5, rl = lane & 31;
    const int M = P.M, K = P.K, R = P.R;
    const PqSrc& S = P.src[blockIdx.z];
    const int c_hi = S.col_hi;
    const int m0 = (int)blockIdx.x * TM, n0 = S.col_lo + (int)blockIdx.y * TN;  // (blockIdx.y: column tile of a segment wider than TN)
    if (n0 >= c_hi) return;  // (uniform: the whole workgroup leaves before any barrier)
    const unsigned char* const wgt = reinterpret_cast<const unsigned char*>(P.wproj);
    const unsigned char* const act = reinterpret_cast<const unsigned char*>(S.act);
    DropoutCfg drop = P.drop;
    mtl_dropout_resolve(drop);
    const bool masked = S.mask != 0 && drop.thr16 != 0;
    const int total = (K + KE - 1) / KE;

    // loader: wave w issues the DMA instructions j = w + 4 t (t < DPW) of a stage; instruction j covers stage rows 16 j .. 16 j + 15
    // (projection rows for 16 j < TN, activation rows after), lane l -> row 16 j + (l >> 2), physical chunk l & 3
    int64_t src[DPW];
    int q8[DPW];
    bool isw[DPW];
#pragma unroll
    for (int t = 0; t < DPW; ++t) {
        const int j = wave + 4 * t, row = 16 * j + (lane >> 2), p = lane & 3;
        q8[t] = (p ^ ((row >> 2) & 3)) * 8;  // logical k offset (elements) of this lane's 16 bytes
        isw[t] = 16 * j < TN;
        if (isw[t]) {
            const int wr = n0 + row < R ? n0 + row : R - 1;
            src[t] = (int64_t)wr * K + q8[t];
        } else {
            const int ar = m0 + row - TN;
            src[t] = (int64_t)(ar < M ? ar : M - 1) * K + q8[t];
        }
    }
    auto issue = [&](int i) __attribute__((always_inline)) {
        const int k0 = i * KE;
        unsigned char* dst = smem + (i % NST) * STAGE;
        if (k0 + KE <= K) {
#pragma unroll
            for (int t = 0; t < DPW; ++t) sp_dma16((isw[t] ? wgt : act) + (src[t] + k0) * 2, dst + (wave + 4 * t) * 1024);
        } else {  // ragged last k-tile: chunks past K read the zero page
#pragma unroll
            for (int t = 0; t < DPW; ++t) {
                const void* g = k0 + q8[t] < K ? (const void*)((isw[t] ? wgt : act) + (src[t] + k0) * 2) : (const void*)g_zero16;
                sp_dma16(g, dst + (wave + 4 * t) * 1024);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
        if (i < total) issue(i);

    f32x16 acc[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)(m0 + wm * 32 + rl));
    int co[2];  // fragment addressing: row (block base + rl) * 64 + ((2 ks + h) ^ ((rl >> 2) & 3)) * 16
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) co[ks] = (((2 * ks + h) ^ ((rl >> 2) & 3)) << 4) + rl * 64;
    const int a_off = (TN + wm * 32) * 64, w_off = wn * (NBW * 32) * 64;

    for (int i = 0; i < total; ++i) {
        const int rest = total - 1 - i;
        pq_wait_vm<DPW>(rest < NST - 2 ? rest : NST - 2);  // stage i of this wave has landed (younger stages may still fly)
        // ... of every wave, and every wave is done reading stage i - 1.  A raw s_barrier: __syncthreads() would add a workgroup fence,
        // which the compiler lowers to s_waitcnt vmcnt(0) while LDS-DMA writes are pending -- the whole ring would drain every k-tile
        asm volatile("" ::: "memory");
        SP_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (i + NST - 1 < total) issue(i + NST - 1);
        const unsigned char* st = smem + (i % NST) * STAGE;
#pragma unroll
        for (int kq = 0; kq < 2 / WK; ++kq) {
            const int ks = KSPLIT ? wk : kq;
            const int cofs = KSPLIT ? (wk ? co[1] : co[0]) : co[kq];
            u32x4 fw[NBW], fa;
            fa = *reinterpret_cast<const u32x4*>(st + a_off + cofs);
#pragma unroll
            for (int b = 0; b < NBW; ++b) fw[b] = *reinterpret_cast<const u32x4*>(st + w_off + b * 32 * 64 + cofs);
            if (masked) VOps<T>::drop(fa, drop, rh, (uint32_t)(i * KE + ks * 16 + 8 * h));
#pragma unroll
            for (int b = 0; b < NBW; ++b) sp_mma1<T>(fw[b], fa, acc[b]);
        }
    }

    __syncthreads();  // (every DMA has landed and been read: the ring is free)
    if constexpr (KSPLIT) {  // the k-halves of a row block meet in LDS: element (b, e) of lane l at ((b * 16 + e) * 64 + l) * 4
        float* red = reinterpret_cast<float*>(smem) + wm * (NB * 16 * 64);
        if (wk == 1) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[(b * 16 + e) * 64 + lane] = acc[b][e];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][e] += red[(b * 16 + e) * 64 + lane];
        }
        __syncthreads();
        if (wk == 1) return;
    }

    // ---- epilogue: the wave's 32 x TN tile through a private LDS image, whole row segments per store
    const __amdgpu_buffer_rsrc_t orsrc = sp_rsrc(P.out, (int64_t)M * P.ld_out * 2);
    const uint32_t ldo2 = (uint32_t)(P.ld_out * 2);
    unsigned char* img = smem + wave * (32 * ORS);
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nl = b * 32 + 8 * q + 4 * h;
            u32x2 pk = {mtl_pk2<T>(acc[b][q * 4], acc[b][q * 4 + 1]), mtl_pk2<T>(acc[b][q * 4 + 2], acc[b][q * 4 + 3])};
            *reinterpret_cast<u32x2*>(img + rl * ORS + nl * 2) = pk;
        }
    SP_WAIT_LGKM0();  // the image is private to this wave
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < CPR / 2; ++it) {
        const int idx = it * 64 + lane, ml = idx / CPR, c16 = idx - ml * CPR;
        const int m = m0 + wm * 32 + ml, n = n0 + wn * (NBW * 32) + c16 * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(img + ml * ORS + c16 * 16);
        sp_bstore(v, orsrc, (m < M && n < c_hi) ? (uint32_t)m * ldo2 + (uint32_t)n * 2u : 0xFFFFFFFFu);
    }
}
