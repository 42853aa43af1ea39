// dense.h -- k_ntd: the MFMA-dense single-output launches of MTLoRALinear (bf16): stage-2 / 3 forward outputs and dX, the decoder
// heads' and PatchMerging GEMMs -- every launch whose reduction is long enough for a pipeline to pay (K + R >= 256) and that has
// enough tiles.  Included by linear.hip after stream.h (sp_dma16 / SP_WAIT_VM / sp_mma1) inside its anonymous namespace.
//
// What was wrong with k_ntl on these shapes (400-460 TFLOP/s, MFMA pipe 17 % busy): ONE k-tile of register prefetch,
// two barriers per k-tile, and a workgroup's serial chain  global -> registers -> LDS -> barrier -> fragments -> MFMA  exposed once
// per k-tile; two workgroups per CU were the only overlap.  Here:
//   * tile 256 (activation rows m) x 128 (weight rows n) x 64 (k), 8 waves as 4 (m) x 2 (n), wave tile 64 x 64 = 2 x 2 MFMA blocks
//     (64 accumulator registers): 4 ds_read_b128 per 4 MFMAs (k_ntl: 6 per 4);
//   * operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers) into a RING of three 48 KB stages: the
//     loads of k-tile i + 2 are issued before k-tile i is multiplied, ONE barrier per k-tile, counted vmcnt waits;
//   * rows are dense 128-byte lines in LDS, the 16-byte chunks permuted on the source side (chunk ^ ((row >> 1) & 7)): the
//     ds_read_b128 fragment reads are conflict-free (the layout of stream.h's 64-wide slabs);
//   * the k-stream is [rank tiles of (P | Q) x (B | A^T)  |  base tiles of (X | dY) x (W | W^T)] as in k_ntl; MLR multiplies the
//     accumulators by the dropout keep-mask of (m, n) when the rank tiles are done (dX launches);
//   * epilogue as k_ntl (per-wave LDS transposition, whole 128-byte row segments per 8 lanes), plus the GELU' gate (GATE: the Mlp's
//     fc2 dX) and the GELU second output (ACT: fc1 forward); stores / gate loads go through buffer descriptors (unconditional:
//     out-of-range rows are dropped by the bounds check), so the number of vector-memory operations of an epilogue is a constant;
//   * one workgroup per CU (144 KB of LDS), PERSISTENT over tiles in the XCD-aware order of k_ntl, and the ring runs ACROSS tiles:
//     the first two stages of the next tile are issued before the epilogue of the current one (whose images live in the ring slot
//     of the last k-tile), so a tile's load prologue hides under the previous tile's stores.  vmcnt: operations complete in issue
//     order (gfx9), so the wait for stage i allows [stage i + 1] + [the epilogue issued after stage i] outstanding.
// Measured and dropped: a contiguous run of tiles per workgroup (see tile_coords); register double-buffering of the fragments across
// k-tiles (+5 %: 240 VGPRs, no gain -- the k loop is bound by the latency of the stage loads, two stages = 96 KB in flight per CU,
// not by LDS / MFMA overlap); 4-byte "touch" loads two k-tiles ahead to warm the L2 (1.5x SLOWER: they double the number of line
// requests in the CU's miss queue and, returning in order, sit in front of the stage loads); full-width warm-up DMAs into a scratch KB two
// k-tiles ahead (1.25x slower); the operands staged through three register sets instead of LDS-DMA -- THREE stages = 144 KB in flight
// per CU on the ordinary vector-load path (correct, 7-10 % slower).  More bytes in flight do not help and more requests hurt: the
// limit is how fast the XCD's L2 serves 32 CUs that ask for 48 KB each per k-tile (1.5 MB per XCD and k-tile, the 12 workgroups of a
// row block hitting the same lines at once), i.e. bytes per flop -- a 256 x 256 tile with stream-K balancing is the next step.
// (Starting each column tile's base k loop at a different k-tile, so that the workgroups of a row block do not ask for the same lines
// at once: +-0.)
#pragma once

constexpr int ND_TM = 256, ND_TN = 128, ND_KE = 64;
constexpr int ND_ROWS = ND_TM + ND_TN;                 // rows of a stage: [weights 0..127 | activation 128..383]
constexpr int ND_STAGE = ND_ROWS * 128;                // 48 KB
constexpr int ND_NST = 3;
constexpr int ND_LDS = ND_NST * ND_STAGE;              // 144 KB
constexpr int ND_DPW = ND_ROWS / 8 / 8;                // DMA instructions per wave per stage (48 / 8 = 6)
constexpr int ND_OPS = ND_DPW;                         // vector-memory operations per wave and stage
constexpr int ND_ORS = 128 + 8;                        // epilogue image row stride (64 columns of bf16 + pad)

template <bool ACT, bool MLR, bool GATE>
__global__ __launch_bounds__(512, 2) void k_ntd(const NlParams P) {
    constexpr int EPI_OPS = 8 * (1 + (ACT ? 1 : 0));  // vector-memory operations of one epilogue (its stores), per wave.  [The gate
    // values are loaded BEFORE the last k-tile is multiplied -- older than the next tile's first stages, so they do not count here.]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, rl = lane & 31;
    // every parameter the loops touch, read once (a by-value parameter block is re-read with dependent scalar loads otherwise)
    const int M = P.M, n_rows = P.n_rows, Kb = P.K, seg_lo = P.seg_lo, seg_hi = P.seg_hi, n_tiles = P.n_tiles;
    const uint32_t q8 = P.q8, r8 = P.r8, nt_magic = P.nt_magic;
    const unsigned char* const wgt_b = reinterpret_cast<const unsigned char*>(P.wgt);
    const unsigned char* const act_b = reinterpret_cast<const unsigned char*>(P.act);
    const unsigned char* const wgt_r = reinterpret_cast<const unsigned char*>(P.Rm);
    const unsigned char* const act_r = reinterpret_cast<const unsigned char*>(P.L);
    const uint32_t ldw_b = (uint32_t)(P.ld_wgt * 2), lda_b = (uint32_t)(P.ld_act * 2), ldw_r = (uint32_t)(P.ldR * 2), lda_r = (uint32_t)(P.ldL * 2);
    const uint32_t ldo2 = (uint32_t)(P.ld_out * 2);
    const float* const bias = P.bias;
    const float* const alpha = P.alpha;
    const int use_base = P.use_base;
    DropoutCfg drop = P.drop;
    mtl_dropout_resolve(drop);
    const int dbg = P.dbg & NT_DBG_MASK;  // developer ablation bits (0 at compile time unless -DMTL_NT_ABLATE=1): 1 no stores, 2 no stage loads, 4 no MFMA, 8 no epilogue
    const __amdgpu_buffer_rsrc_t orsrc = sp_rsrc(P.out, (int64_t)M * P.ld_out * 2);
    const __amdgpu_buffer_rsrc_t arsrc = sp_rsrc(P.act2, ACT ? (int64_t)M * P.ld_out * 2 : 0);
    const __amdgpu_buffer_rsrc_t grsrc = sp_rsrc(const_cast<bf16*>(P.gate), GATE ? (int64_t)M * P.ld_out * 2 : 0);
    (void)arsrc;
    (void)grsrc;

    const int n1 = seg_hi > seg_lo ? (seg_hi - seg_lo + ND_KE - 1) / ND_KE : 0;
    const int n2 = (use_base && Kb > 0) ? (Kb + ND_KE - 1) / ND_KE : 0;
    const int total = n1 + n2;
    const uint32_t n_wg_tiles = q8 * 8u + r8;  // tiles of the launch
    if (total == 0) return;

    // fragment addressing: row (block base + rl) * 128 + ((2 ks + h) ^ ((rl >> 1) & 7)) * 16 -- the permutation term only depends on rl
    int co[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) co[ks] = (((2 * ks + h) ^ ((rl >> 1) & 7)) << 4) + rl * 128;
    const int w_off = (wn * 64) * 128, a_off = (ND_TN + wm * 64) * 128;

    // ---- loader: wave w issues the DMA instructions j = w + 8 t (t < 6) of a stage; instruction j covers stage rows 8 j .. 8 j + 7
    // (weights for j < 16, activation rows after), lane l -> row 8 j + (l >> 3), physical chunk l & 7
    uint32_t rowc[ND_DPW], q16[ND_DPW];  // clamped matrix row (of the tile being LOADED) and byte offset of the logical chunk
#pragma unroll
    for (int t = 0; t < ND_DPW; ++t) {
        const int row = 8 * (wave + 8 * t) + (lane >> 3), p = lane & 7;
        q16[t] = (uint32_t)((p ^ ((row >> 1) & 7)) * 16);
    }
    auto tile_coords = [&](uint32_t tile, int& m0, int& n0) __attribute__((always_inline)) {
        uint32_t b = tile;  // XCD-aware tile order (as k_ntl): logical tile index from the dispatch slot.  [A contiguous run of tiles per
        // workgroup -- the n-tiles of one row block back to back on one CU -- measured 20 % SLOWER: 32 distinct row blocks per XCD
        // do not fit its L2, while here the 12 workgroups that share a row block fetch it once.]
        b = ((b & 7u) < r8 ? (b & 7u) * (q8 + 1u) : r8 * (q8 + 1u) + ((b & 7u) - r8) * q8) + (b >> 3);
        const uint32_t bm = n_tiles == 1 ? b : __umulhi(b, nt_magic);
        m0 = (int)bm * ND_TM;
        n0 = (int)(b - bm * (uint32_t)n_tiles) * ND_TN;
    };
    auto set_rows = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < ND_DPW; ++t) {
            const int j = wave + 8 * t, row = 8 * j + (lane >> 3);
            if (j < ND_TN / 8) {
                const int wr = n0 + row;
                rowc[t] = (uint32_t)(wr < n_rows ? wr : n_rows - 1);
            } else {
                const int ar = m0 + row - ND_TN;
                rowc[t] = (uint32_t)(ar < M ? ar : M - 1);
            }
        }
    };
    auto issue = [&](int i, int slot) __attribute__((always_inline)) {  // k-tile i of the tile whose rows are in rowc -> ring slot
        if (dbg & 2) return;
        const bool lr = i < n1;
        const int k0 = lr ? seg_lo + i * ND_KE : (i - n1) * ND_KE;
        const int khi = lr ? seg_hi : Kb;
        unsigned char* dst = smem + slot * ND_STAGE;
        const unsigned char* wb = (lr ? wgt_r : wgt_b) + (int64_t)k0 * 2;
        const unsigned char* ab = (lr ? act_r : act_b) + (int64_t)k0 * 2;
        const uint32_t ldw2 = lr ? ldw_r : ldw_b, lda2 = lr ? lda_r : lda_b;
        if (k0 + ND_KE <= khi) {
#pragma unroll
            for (int t = 0; t < ND_DPW; ++t) {
                const int j = wave + 8 * t;
                const bool isw = j < ND_TN / 8;
                sp_dma16((isw ? wb : ab) + (rowc[t] * (isw ? ldw2 : lda2) + q16[t]), dst + j * 1024);
            }
        } else {  // ragged last k-tile of a part: chunks past the end read the zero page
#pragma unroll
            for (int t = 0; t < ND_DPW; ++t) {
                const int j = wave + 8 * t;
                const bool isw = j < ND_TN / 8;
                const void* g = k0 + (int)(q16[t] >> 1) < khi ? (const void*)((isw ? wb : ab) + (rowc[t] * (isw ? ldw2 : lda2) + q16[t]))
                                                              : (const void*)g_zero16;
                sp_dma16(g, dst + j * 1024);
            }
        }
    };

    int base_slot = 0;  // ring slot of k-tile 0 of the current tile
    bool first = true;
    int m0 = 0, n0 = 0;
    if (blockIdx.x < n_wg_tiles) {
        tile_coords(blockIdx.x, m0, n0);
        set_rows(m0, n0);
        issue(0, 0);
        if (total > 1) issue(1, 1);
    }
    for (uint32_t tile = blockIdx.x; tile < n_wg_tiles; tile += gridDim.x) {
        f32x16 acc[2][2];  // [n block][m block]
        const bool bias_first = !MLR && bias != nullptr && alpha == nullptr && use_base != 0;
#pragma unroll
        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bi = {0.f, 0.f, 0.f, 0.f};
                if (bias_first) {
                    int n = n0 + wn * 64 + sn * 32 + 8 * q + 4 * h;
                    n = n < n_rows - 4 ? n : n_rows - 4;
                    bi = *reinterpret_cast<const f32x4*>(bias + n);
                }
#pragma unroll
                for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[sn][sm][q * 4 + e] = bi[e];
            }

        int slot = base_slot;
        u32x4 hv[2][4];  // GATE: the pre-activations of this wave's 64 x 64 tile, in the order the epilogue stores it
        (void)hv;
        for (int i = 0; i < total; ++i) {
            // operations issued after stage i's loads: stage i + 1 (if any) and, for the first two stages of a tile that follows
            // another one, that tile's epilogue
            const bool nxt = i + 1 < total, epi = !first && i < 2;
            if (nxt && epi)
                SP_WAIT_VM(ND_OPS + EPI_OPS);
            else if (epi)
                SP_WAIT_VM(EPI_OPS);
            else if (nxt)
                SP_WAIT_VM(ND_OPS);
            else
                SP_WAIT_VM(0);
            __syncthreads();  // stage i is complete for every wave; every wave is done reading stage i - 1 (and its epilogue images)
            if (i + 2 < total) issue(i + 2, slot >= 1 ? slot - 1 : slot + 2);
            if constexpr (GATE) {
                if (i + 1 == total) {  // in flight while the last k-tile is multiplied; consumed by the epilogue
                    const int n = n0 + wn * 64 + (lane & 7) * 8;
#pragma unroll
                    for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int m = m0 + wm * 64 + sm * 32 + it * 8 + (lane >> 3);
                            const uint32_t off = (m < M && n < n_rows) ? (uint32_t)m * ldo2 + (uint32_t)n * 2u : 0xFFFFFFFFu;
                            hv[sm][it] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, (int)off, 0, 0);
                        }
                }
            }
            const unsigned char* st = smem + slot * ND_STAGE;
            const unsigned char* sw = st + w_off;
            const unsigned char* sa = st + a_off;
            u32x4 fw[4][2], fa[4][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int bq = 0; bq < 2; ++bq) {
                    fw[ks][bq] = *reinterpret_cast<const u32x4*>(sw + bq * 32 * 128 + co[ks]);
                    fa[ks][bq] = *reinterpret_cast<const u32x4*>(sa + bq * 32 * 128 + co[ks]);
                }
            if (!(dbg & 4)) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                        for (int sm = 0; sm < 2; ++sm) sp_mma1<bf16>(fw[ks][sn], fa[ks][sm], acc[sn][sm]);
            }
            if constexpr (MLR) {
                if (i + 1 == n1 && drop.thr16 != 0) {  // the rank part is complete: acc *= keep(m, n)
#pragma unroll
                    for (int sm = 0; sm < 2; ++sm) {
                        const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)(m0 + wm * 64 + sm * 32 + rl));
#pragma unroll
                        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int n = n0 + wn * 64 + sn * 32 + 8 * q + 4 * h;
                                const uint32_t h0 = mtl_dropout_pairbits(drop, rh, (uint32_t)n);
                                const uint32_t h1 = mtl_dropout_pairbits(drop, rh, (uint32_t)(n + 2));
                                if ((h0 & 0xFFFFu) < drop.thr16) acc[sn][sm][q * 4 + 0] = 0.f;
                                if ((h0 >> 16) < drop.thr16) acc[sn][sm][q * 4 + 1] = 0.f;
                                if ((h1 & 0xFFFFu) < drop.thr16) acc[sn][sm][q * 4 + 2] = 0.f;
                                if ((h1 >> 16) < drop.thr16) acc[sn][sm][q * 4 + 3] = 0.f;
                            }
                    }
                }
            }
            slot = slot == ND_NST - 1 ? 0 : slot + 1;
        }
        // `slot` is now the ring slot after the last k-tile = slot of the next tile's k-tile 0; the last k-tile sat in slot - 1
        const int img_slot = slot >= 1 ? slot - 1 : ND_NST - 1;
        const int m0c = m0, n0c = n0;
        // ---- next tile: its first two stages go out before this tile's epilogue (their slots are free: every wave passed the
        // last barrier after it had finished with them)
        {
            const uint32_t nt = tile + gridDim.x;
            if (nt < n_wg_tiles) {
                tile_coords(nt, m0, n0);
                set_rows(m0, n0);
                issue(0, slot);
                if (total > 1) issue(1, slot == ND_NST - 1 ? 0 : slot + 1);
            }
        }
        base_slot = slot;
        first = false;

        if (!bias_first && (alpha || bias) && use_base) {  // acc = acc * alpha[n] + bias[n]
#pragma unroll
            for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int n = n0c + wn * 64 + sn * 32 + 8 * q + 4 * h;
                    n = n < n_rows - 4 ? n : n_rows - 4;
                    f32x4 al = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
                    if (alpha) al = *reinterpret_cast<const f32x4*>(alpha + n);
                    if (bias) bi = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
                    for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[sn][sm][q * 4 + e] = acc[sn][sm][q * 4 + e] * al[e] + bi[e];
                }
        }

        // ---- epilogue: the wave's 64 (m) x 64 (n) tile, 32 rows at a time, through a private image in the ring slot of the last
        // k-tile (free once every wave has left the k loop); then whole 128-byte row segments per 8 lanes.  EPI_OPS operations.
        __syncthreads();
        if (!(dbg & 8)) {
            unsigned char* img = smem + img_slot * ND_STAGE + wave * (32 * ND_ORS);
            const int c16 = lane & 7;
            const int n = n0c + wn * 64 + c16 * 8;
#pragma unroll
            for (int sm = 0; sm < 2; ++sm) {
#pragma unroll
                for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = sn * 32 + 8 * q + 4 * h;
                        u32x2 pk = {mtl_pack_bf16(acc[sn][sm][q * 4], acc[sn][sm][q * 4 + 1]),
                                    mtl_pack_bf16(acc[sn][sm][q * 4 + 2], acc[sn][sm][q * 4 + 3])};
                        *reinterpret_cast<u32x2*>(img + rl * ND_ORS + nl * 2) = pk;
                    }
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the image is private to this wave
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int ml = it * 8 + (lane >> 3);
                    const int m = m0c + wm * 64 + sm * 32 + ml;
                    u32x4 v = *reinterpret_cast<const u32x4*>(img + ml * ND_ORS + c16 * 16);
                    const uint32_t off = (m < M && n < n_rows && !(dbg & 1)) ? (uint32_t)m * ldo2 + (uint32_t)n * 2u : 0xFFFFFFFFu;
                    if constexpr (GATE) {  // the bf16-rounded gradient times gelu'(pre-activation), rounded once (as ATen does)
                        v = mtl_gelu_gate_pk4<bf16, false>(v, hv[sm][it]);
                    }
                    sp_bstore(v, orsrc, off);
                    if constexpr (ACT) {
                        const u32x4 av = mtl_gelu_pk4<bf16, false>(v);
                        sp_bstore(av, arsrc, off);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// k_nte (round 4) -- the same launches with TWO workgroups per CU.  The ablation of k_ntd (skeleton 15 + epilogue 12 +
// stage loads 13 + MFMA 18-23 us ADD UP to the 62 us of s2.fc1; the loads alone run at 27 TB/s out of L2) says a lock-step 8-wave
// workgroup overlaps none of its phases.  Here a workgroup is 4 waves (one per SIMD) on a 256 (m) x 128 (n) tile with 32-wide k-tiles:
//   * wave tile 128 (m) x 64 (n) = 4 x 2 MFMA blocks (128 accumulator registers): 6 ds_read_b128 per 8 MFMAs (k_ntd: 4 per 4);
//   * stage = 384 rows x 64 B = 24 KB, ring of three = 72 KB: two workgroups per CU, each with two stages in flight, drifting out of
//     phase -- one multiplies while the other waits for a stage, sits in its barrier or stores its tile;
//   * rows are dense 64-byte lines in LDS, 16-byte chunks permuted on the source side (chunk ^ ((row >> 2) & 3)): the 16 rows a
//     ds_read_b128 lane group reads (stride 64 B) cover the 16 four-bank groups exactly once;
//   * k-stream, masking (MLR), bias-as-initial-value, ACT second output, persistent XCD-aware tile loop and the counted vmcnt waits as in
//     k_ntd; the GATE epilogue loads the pre-activations of a 32-row block right before that block is stored (64 more registers for a
//     whole-tile prefetch do not fit next to 128 accumulators).
// ------------------------------------------------------------------------------------------------
constexpr int NE_TM = 256, NE_TN = 128, NE_KE = 32;
constexpr int NE_ROWS = NE_TM + NE_TN;                 // [weights 0..127 | activation 128..383]
constexpr int NE_STAGE = NE_ROWS * 64;                 // 24 KB
constexpr int NE_NST = 3;
constexpr int NE_LDS = NE_NST * NE_STAGE;              // 72 KB
constexpr int NE_DPW = NE_ROWS / 16 / 4;               // DMA instructions per wave per stage (24 / 4 = 6)
constexpr int NE_ORS = 128 + 8;                        // epilogue image row stride

template <bool ACT, bool MLR, bool GATE>
__global__ __launch_bounds__(256, 2) void k_nte(const NlParams P) {
    constexpr int EPI_OPS = 16 * (1 + (ACT ? 1 : 0)) + (GATE ? 16 : 0);  // vector-memory operations of one epilogue, per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, rl = lane & 31;
    const int M = P.M, n_rows = P.n_rows, Kb = P.K, seg_lo = P.seg_lo, seg_hi = P.seg_hi, n_tiles = P.n_tiles;
    const uint32_t q8 = P.q8, r8 = P.r8, nt_magic = P.nt_magic;
    const unsigned char* const wgt_b = reinterpret_cast<const unsigned char*>(P.wgt);
    const unsigned char* const act_b = reinterpret_cast<const unsigned char*>(P.act);
    const unsigned char* const wgt_r = reinterpret_cast<const unsigned char*>(P.Rm);
    const unsigned char* const act_r = reinterpret_cast<const unsigned char*>(P.L);
    const uint32_t ldw_b = (uint32_t)(P.ld_wgt * 2), lda_b = (uint32_t)(P.ld_act * 2), ldw_r = (uint32_t)(P.ldR * 2), lda_r = (uint32_t)(P.ldL * 2);
    const uint32_t ldo2 = (uint32_t)(P.ld_out * 2);
    const float* const bias = P.bias;
    const float* const alpha = P.alpha;
    const int use_base = P.use_base;
    DropoutCfg drop = P.drop;
    mtl_dropout_resolve(drop);
    const int dbg = P.dbg & NT_DBG_MASK;  // developer ablation bits as in k_ntd (0 at compile time in the shipped library)
    const __amdgpu_buffer_rsrc_t orsrc = sp_rsrc(P.out, (int64_t)M * P.ld_out * 2);
    const __amdgpu_buffer_rsrc_t arsrc = sp_rsrc(P.act2, ACT ? (int64_t)M * P.ld_out * 2 : 0);
    const __amdgpu_buffer_rsrc_t grsrc = sp_rsrc(const_cast<bf16*>(P.gate), GATE ? (int64_t)M * P.ld_out * 2 : 0);
    (void)arsrc;
    (void)grsrc;

    const int n1 = seg_hi > seg_lo ? (seg_hi - seg_lo + NE_KE - 1) / NE_KE : 0;
    const int n2 = (use_base && Kb > 0) ? (Kb + NE_KE - 1) / NE_KE : 0;
    const int total = n1 + n2;
    const uint32_t n_wg_tiles = q8 * 8u + r8;
    if (total == 0) return;

    // fragment addressing: row (block base + rl) * 64 + ((2 ks + h) ^ ((rl >> 2) & 3)) * 16
    int co[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) co[ks] = (((2 * ks + h) ^ ((rl >> 2) & 3)) << 4) + rl * 64;
    const int w_off = (wn * 64) * 64, a_off = (NE_TN + wm * 128) * 64;

    // loader: wave w issues the DMA instructions j = w + 4 t (t < 6) of a stage; instruction j covers stage rows 16 j .. 16 j + 15
    // (weights for j < 8, activation rows after), lane l -> row 16 j + (l >> 2), physical chunk l & 3
    uint32_t rowc[NE_DPW], q16[NE_DPW];
#pragma unroll
    for (int t = 0; t < NE_DPW; ++t) {
        const int row = 16 * (wave + 4 * t) + (lane >> 2), p = lane & 3;
        q16[t] = (uint32_t)((p ^ ((row >> 2) & 3)) * 16);
    }
    auto tile_coords = [&](uint32_t tile, int& m0, int& n0) __attribute__((always_inline)) {
        uint32_t b = tile;
        b = ((b & 7u) < r8 ? (b & 7u) * (q8 + 1u) : r8 * (q8 + 1u) + ((b & 7u) - r8) * q8) + (b >> 3);
        const uint32_t bm = n_tiles == 1 ? b : __umulhi(b, nt_magic);
        m0 = (int)bm * NE_TM;
        n0 = (int)(b - bm * (uint32_t)n_tiles) * NE_TN;
    };
    auto set_rows = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NE_DPW; ++t) {
            const int j = wave + 4 * t, row = 16 * j + (lane >> 2);
            if (j < NE_TN / 16) {
                const int wr = n0 + row;
                rowc[t] = (uint32_t)(wr < n_rows ? wr : n_rows - 1);
            } else {
                const int ar = m0 + row - NE_TN;
                rowc[t] = (uint32_t)(ar < M ? ar : M - 1);
            }
        }
    };
    auto issue = [&](int i, int slot) __attribute__((always_inline)) {
        if (dbg & 2) return;
        const bool lr = i < n1;
        const int k0 = lr ? seg_lo + i * NE_KE : (i - n1) * NE_KE;
        const int khi = lr ? seg_hi : Kb;
        unsigned char* dst = smem + slot * NE_STAGE;
        const unsigned char* wb = (lr ? wgt_r : wgt_b) + (int64_t)k0 * 2;
        const unsigned char* ab = (lr ? act_r : act_b) + (int64_t)k0 * 2;
        const uint32_t ldw2 = lr ? ldw_r : ldw_b, lda2 = lr ? lda_r : lda_b;
        if (k0 + NE_KE <= khi) {
#pragma unroll
            for (int t = 0; t < NE_DPW; ++t) {
                const int j = wave + 4 * t;
                const bool isw = j < NE_TN / 16;
                sp_dma16((isw ? wb : ab) + (rowc[t] * (isw ? ldw2 : lda2) + q16[t]), dst + j * 1024);
            }
        } else {  // ragged last k-tile of a part: chunks past the end read the zero page
#pragma unroll
            for (int t = 0; t < NE_DPW; ++t) {
                const int j = wave + 4 * t;
                const bool isw = j < NE_TN / 16;
                const void* g = k0 + (int)(q16[t] >> 1) < khi ? (const void*)((isw ? wb : ab) + (rowc[t] * (isw ? ldw2 : lda2) + q16[t]))
                                                              : (const void*)g_zero16;
                sp_dma16(g, dst + j * 1024);
            }
        }
    };

    int base_slot = 0;
    bool first = true;
    int m0 = 0, n0 = 0;
    if (blockIdx.x < n_wg_tiles) {
        tile_coords(blockIdx.x, m0, n0);
        set_rows(m0, n0);
        issue(0, 0);
        if (total > 1) issue(1, 1);
    }
    for (uint32_t tile = blockIdx.x; tile < n_wg_tiles; tile += gridDim.x) {
        f32x16 acc[2][4];  // [n block][m block]
        const bool bias_first = !MLR && bias != nullptr && alpha == nullptr && use_base != 0;
#pragma unroll
        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bi = {0.f, 0.f, 0.f, 0.f};
                if (bias_first) {
                    int n = n0 + wn * 64 + sn * 32 + 8 * q + 4 * h;
                    n = n < n_rows - 4 ? n : n_rows - 4;
                    bi = *reinterpret_cast<const f32x4*>(bias + n);
                }
#pragma unroll
                for (int sm = 0; sm < 4; ++sm)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[sn][sm][q * 4 + e] = bi[e];
            }

        int slot = base_slot;
        for (int i = 0; i < total; ++i) {
            const bool nxt = i + 1 < total, epi = !first && i < 2;
            if (nxt && epi)
                SP_WAIT_VM(NE_DPW + EPI_OPS);
            else if (epi)
                SP_WAIT_VM(EPI_OPS);
            else if (nxt)
                SP_WAIT_VM(NE_DPW);
            else
                SP_WAIT_VM(0);
            __syncthreads();  // stage i is complete for every wave; every wave is done reading stage i - 1 (and its epilogue images)
            if (i + 2 < total) issue(i + 2, slot >= 1 ? slot - 1 : slot + 2);
            const unsigned char* st = smem + slot * NE_STAGE;
            const unsigned char* sw = st + w_off;
            const unsigned char* sa = st + a_off;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 fw[2], fa[4];
#pragma unroll
                for (int bq = 0; bq < 2; ++bq) fw[bq] = *reinterpret_cast<const u32x4*>(sw + bq * 32 * 64 + co[ks]);
#pragma unroll
                for (int bq = 0; bq < 4; ++bq) fa[bq] = *reinterpret_cast<const u32x4*>(sa + bq * 32 * 64 + co[ks]);
                if (!(dbg & 4)) {
#pragma unroll
                    for (int sm = 0; sm < 4; ++sm)
#pragma unroll
                        for (int sn = 0; sn < 2; ++sn) sp_mma1<bf16>(fw[sn], fa[sm], acc[sn][sm]);
                }
            }
            if constexpr (MLR) {
                if (i + 1 == n1 && drop.thr16 != 0) {  // the rank part is complete: acc *= keep(m, n)
#pragma unroll
                    for (int sm = 0; sm < 4; ++sm) {
                        const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)(m0 + wm * 128 + sm * 32 + rl));
#pragma unroll
                        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int n = n0 + wn * 64 + sn * 32 + 8 * q + 4 * h;
                                const uint32_t h0 = mtl_dropout_pairbits(drop, rh, (uint32_t)n);
                                const uint32_t h1 = mtl_dropout_pairbits(drop, rh, (uint32_t)(n + 2));
                                if ((h0 & 0xFFFFu) < drop.thr16) acc[sn][sm][q * 4 + 0] = 0.f;
                                if ((h0 >> 16) < drop.thr16) acc[sn][sm][q * 4 + 1] = 0.f;
                                if ((h1 & 0xFFFFu) < drop.thr16) acc[sn][sm][q * 4 + 2] = 0.f;
                                if ((h1 >> 16) < drop.thr16) acc[sn][sm][q * 4 + 3] = 0.f;
                            }
                    }
                }
            }
            slot = slot == NE_NST - 1 ? 0 : slot + 1;
        }
        const int img_slot = slot >= 1 ? slot - 1 : NE_NST - 1;
        const int m0c = m0, n0c = n0;
        {
            const uint32_t nt = tile + gridDim.x;
            if (nt < n_wg_tiles) {
                tile_coords(nt, m0, n0);
                set_rows(m0, n0);
                issue(0, slot);
                if (total > 1) issue(1, slot == NE_NST - 1 ? 0 : slot + 1);
            }
        }
        base_slot = slot;
        first = false;

        if (!bias_first && (alpha || bias) && use_base) {  // acc = acc * alpha[n] + bias[n]
#pragma unroll
            for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int n = n0c + wn * 64 + sn * 32 + 8 * q + 4 * h;
                    n = n < n_rows - 4 ? n : n_rows - 4;
                    f32x4 al = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
                    if (alpha) al = *reinterpret_cast<const f32x4*>(alpha + n);
                    if (bias) bi = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
                    for (int sm = 0; sm < 4; ++sm)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[sn][sm][q * 4 + e] = acc[sn][sm][q * 4 + e] * al[e] + bi[e];
                }
        }

        // ---- epilogue: the wave's 128 (m) x 64 (n) tile, 32 rows at a time, through a private image in the ring slot of the last
        // k-tile (free once every wave has left the k loop); whole 128-byte row segments per 8 lanes.  EPI_OPS operations.
        __syncthreads();
        if (!(dbg & 8)) {
            unsigned char* img = smem + img_slot * NE_STAGE + wave * (32 * NE_ORS);
            const int c16 = lane & 7;
            const int n = n0c + wn * 64 + c16 * 8;
#pragma unroll
            for (int sm = 0; sm < 4; ++sm) {
                u32x4 hv[4];
                (void)hv;
                if constexpr (GATE) {  // pre-activations of this 32-row block, in the order the stores below use them
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int m = m0c + wm * 128 + sm * 32 + it * 8 + (lane >> 3);
                        const uint32_t off = (m < M && n < n_rows) ? (uint32_t)m * ldo2 + (uint32_t)n * 2u : 0xFFFFFFFFu;
                        hv[it] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, (int)off, 0, 0);
                    }
                }
#pragma unroll
                for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = sn * 32 + 8 * q + 4 * h;
                        u32x2 pk = {mtl_pack_bf16(acc[sn][sm][q * 4], acc[sn][sm][q * 4 + 1]),
                                    mtl_pack_bf16(acc[sn][sm][q * 4 + 2], acc[sn][sm][q * 4 + 3])};
                        *reinterpret_cast<u32x2*>(img + rl * NE_ORS + nl * 2) = pk;
                    }
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the image is private to this wave
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int ml = it * 8 + (lane >> 3);
                    const int m = m0c + wm * 128 + sm * 32 + ml;
                    u32x4 v = *reinterpret_cast<const u32x4*>(img + ml * NE_ORS + c16 * 16);
                    const uint32_t off = (m < M && n < n_rows) ? (uint32_t)m * ldo2 + (uint32_t)n * 2u : 0xFFFFFFFFu;
                    if constexpr (GATE) {
                        v = mtl_gelu_gate_pk4<bf16, false>(v, hv[it]);
                    }
                    sp_bstore(v, orsrc, off);
                    if constexpr (ACT) {
                        const u32x4 av = mtl_gelu_pk4<bf16, false>(v);
                        sp_bstore(av, arsrc, off);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

