// stream.h -- k_sp*: wave-streaming kernels for the HBM-side MTLoRALinear launches (16-bit types).  Included by linear.hip
// inside its anonymous namespace (after Segs / CtxLayout / g_zero16 / gelu_* / MtlProfScope are visible).
//
// Why another family: the tiled kernels (k_nt / k_ntl) run a workgroup as ONE serial chain
//     global -> registers -> LDS -> barrier -> MFMA -> barrier -> ... -> epilogue -> exit
// with two workgroups per CU to overlap it: at K = 96 a tile is one or two k-steps, so a CU has ~20 KB of HBM reads in flight
// on average and nothing in flight while its workgroups store -- 2.6-3.8 TB/s at best, and a layer needs three launches
// (P pass, outputs, and the Q pass on the way back) that re-read X / dY.  Here a WAVE is the unit of work:
//   * every wave owns 32-row slabs of the activation (slab s of wave w: w + i * n_waves), streams them through its PRIVATE
//     LDS slots with LDS-DMA loads (global_load_lds_dwordx4, no staging registers), reads them back as MFMA fragments and
//     keeps going -- no workgroup barrier after the prologue, so the load, MFMA and store phases of the 8 (x1 or x2) waves
//     of a CU interleave freely and the next slab's DMA is in flight while the current one is multiplied and stored;
//   * the weight-like operands are STATIONARY: loaded once per workgroup into LDS in fragment-major order (one 1 KB,
//     conflict-free ds_read_b128 per fragment), the grid is persistent (<= 1-2 workgroups per CU);
//   * the low-rank projection never leaves registers: P^T = (alpha A) D(X)^T comes out of the MFMA with a lane holding,
//     for its row m, rank rows r = 8 q + 4 h + e -- exactly a B-operand fragment of the NEXT MFMA once the expansion
//     factor's fragments are packed with the same k permutation (k_pack does that): no P image in LDS, no P pass, no
//     re-read of X, and (with the Z-form factor gradients) no P / Q in HBM at all;
//   * the epilogue writes 16 bytes per lane straight from the accumulators (v_permlane32_swap pairs the two half-waves'
//     8-byte pieces, T21 of the CDNA guide): no LDS transpose, no barrier.
// Slab layout in a slot: 32 rows x CH elements, dense (LDS-DMA writes lane-linear), with the 16-byte chunks of a row
// permuted on the SOURCE side so that the ds_read_b128 fragment reads (16 lanes = 16 rows of one logical chunk per LDS
// cycle) are conflict-free:  CH = 96 (192-byte rows, 12 chunks): physical = (logical + (row >> 2)) mod 12;
//                            CH = 64 (128-byte rows,  8 chunks): physical = logical ^ ((row >> 1) & 7).
// vmcnt discipline: a wave's loads return in order, its stores may not be ordered against them, so a wait for chunk t uses
// vmcnt(#DMA instructions of the YOUNGER chunks only): whatever the stores do, the count can only be reached once every
// older load has landed (it may additionally wait for store acknowledgements -- the other waves of the CU cover that).
#pragma once

#ifndef MTL_SP_WAVES
#define MTL_SP_WAVES 8
#endif
constexpr int SP_WAVES = MTL_SP_WAVES;
constexpr int SP_LDS_MAX = 160 * 1024;
constexpr int SP_MAXB = 4;  // 32-row weight blocks per projection source (rank segment <= 128 columns)

template <int CH>
struct SpGeom {
    static constexpr int CPR = CH / 8;        // 16-byte chunks per row
    static constexpr int ROWB = CH * 2;       // bytes per row
    static constexpr int SLOT = 32 * ROWB;    // bytes per slab chunk (6 KB / 4 KB)
    static constexpr int NDMA = SLOT / 1024;  // LDS-DMA wave instructions per chunk
    static constexpr int KS = CH / 16;        // MFMA k-steps per chunk
    static __device__ __forceinline__ int phys(int row, int q) {
        if constexpr (CH == 96) {
            const int p = q + (row >> 2);
            return p >= 12 ? p - 12 : p;
        } else {
            return q ^ ((row >> 1) & 7);
        }
    }
    static __device__ __forceinline__ int logical(int row, int p) {
        if constexpr (CH == 96) {
            const int q = p - (row >> 2);
            return q < 0 ? q + 12 : q;
        } else {
            return p ^ ((row >> 1) & 7);
        }
    }
};

constexpr int sp_vmcnt(int n) { return (n & 0xF) | ((n >> 4) << 14) | 0x0F70; }  // s_waitcnt vmcnt(n) only
#define SP_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(sp_vmcnt(n))
#define SP_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f)

__device__ __forceinline__ void sp_dma16(const void* g, unsigned char* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0,
                                     0);
}

// 16-byte store through a buffer descriptor: 32-bit byte offset, rows past the end of the tensor are dropped by the bounds
// check (offset >= num_records), an invalid column is expressed as offset 0xFFFFFFFF -- no exec-masked branches around the
// stores, no 64-bit address arithmetic, and every store instruction is issued (the vmcnt bookkeeping relies on that)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sp_rsrc(void* p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : bytes), 0x00020000);
}
__device__ __forceinline__ void sp_bstore(const u32x4& v, __amdgpu_buffer_rsrc_t r, uint32_t off) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, 0);
}

template <typename T>
__device__ __forceinline__ void sp_mma1(const u32x4& a, const u32x4& b, f32x16& c) {
    if constexpr (__is_same(T, f16))
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// wait for the oldest outstanding chunk given how many YOUNGER chunks of NDMA instructions each are still in flight
template <int NDMA>
__device__ __forceinline__ void sp_wait_chunk(int younger) {
    if (younger >= 2)
        SP_WAIT_VM(2 * NDMA);
    else if (younger == 1)
        SP_WAIT_VM(NDMA);
    else
        SP_WAIT_VM(0);
}

// accumulator block (32 weight rows x 32 activation rows) -> global rows m (lane & 31), columns col0 + [0, 32):
// lane (m, h) holds, for q = 0..3, the four consecutive columns 8 q + 4 h + e.  permlane32_swap pairs the half-waves'
// 8-byte pieces: lanes < 32 end up with columns 8 q .. 8 q + 7 (q even), lanes >= 32 with 8 q + 8 .. 8 q + 15 -> one
// 16-byte store per lane and pair.  `lo` / `hi`: only columns in [lo, hi) are written (multiples of 8).
template <typename T>
__device__ __forceinline__ void sp_pack_pair(const f32x16& a, int q, int hl, u32x4& v) {
    uint32_t a0 = mtl_pk2<T>(a[4 * q + 0], a[4 * q + 1]), a1 = mtl_pk2<T>(a[4 * q + 2], a[4 * q + 3]);
    uint32_t b0 = mtl_pk2<T>(a[4 * q + 4], a[4 * q + 5]), b1 = mtl_pk2<T>(a[4 * q + 6], a[4 * q + 7]);
    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    (void)hl;
    v = u32x4{r0[0], r1[0], r0[1], r1[1]};
}

// ------------------------------------------------------------------------------------------------
// k_sp_proj : Out[m][c] = sum_k Wp[c][k] * f_s(X_s[m][k])   for the columns c of source s   (P = alpha D(X) A^T per source,
// Q = alpha dY_o B_o per output: the P / Q passes).  Work item = (slab of 32 rows, source); the reduction streams through
// the wave's slots in chunks of CH, the weights (all sources' rows) are stationary in LDS.
// ------------------------------------------------------------------------------------------------
struct SpSrc {
    const void* act;  // (M x K) contiguous
    int blk_lo, n_blk;  // 32-row weight blocks that cover the source's columns
    int col_lo, col_hi; // columns of Out (= rows of Wp) this source owns
    int mask;           // dropout keep-mask on the activation
    int pad_;
};
struct SpProjParams {
    const void* wproj;  // (Rw x K) row-major
    void* out;          // (M x ld_out)
    int64_t ld_out;
    int64_t M;
    int K, Rw, n_src, n_blk_total;
    int n_slabs, n_items;
    DropoutCfg drop;
    SpSrc src[MAXO];
};
typedef const __attribute__((address_space(4))) SpProjParams* SpProjPtr;

template <typename T, int CH, int NS>
__global__ __launch_bounds__(64 * SP_WAVES, SP_WAVES / 4) void k_sp_proj(const SpProjParams Pv) {
    typedef SpGeom<CH> G;
    (void)Pv;
    SpProjPtr P = (SpProjPtr)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, rl = lane & 31;
    const int K = P->K, KST = K >> 4, NCH = K / CH;
    const int n_items = P->n_items, n_src = P->n_src;
    const int64_t M = P->M;
    unsigned char* wl = smem;                                                     // weights, fragment-major
    unsigned char* slots = smem + (size_t)P->n_blk_total * KST * 1024 + (size_t)wave * NS * G::SLOT;
    DropoutCfg drop;
    drop.seed_lo = P->drop.seed_lo;
    drop.seed_hi = P->drop.seed_hi;
    drop.thr16 = P->drop.thr16;
    drop.off = P->drop.off;
    mtl_dropout_resolve(drop);

    // ---- stationary weights: fragment f = blk * KST + ks <- rows blk*32 + (lane & 31), elements ks*16 + 8 h .. + 8
    {
        const T* wp = reinterpret_cast<const T*>(P->wproj);
        const int nfrag = P->n_blk_total * KST;
        for (int f = wave; f < nfrag; f += SP_WAVES) {
            const int blk = f / KST, ks = f - blk * KST;
            const int row = blk * 32 + rl;
            const void* g = row < P->Rw ? (const void*)(wp + (int64_t)row * K + ks * 16 + 8 * h) : (const void*)g_zero16;
            sp_dma16(g, wl + (size_t)f * 1024);
        }
    }
    // per-lane constants of the slab traffic
    int goff[G::NDMA], grow[G::NDMA], fo[G::KS];
#pragma unroll
    for (int j = 0; j < G::NDMA; ++j) {
        const int c = j * 64 + lane, row = c / G::CPR, p = c - row * G::CPR;
        grow[j] = row;
        goff[j] = row * K + G::logical(row, p) * 8;
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) fo[ks] = rl * G::ROWB + G::phys(rl, 2 * ks + h) * 16;


    const int n_waves = gridDim.x * SP_WAVES;
    const int w_gid = blockIdx.x * SP_WAVES + wave;
    // loader cursor
    int l_item = w_gid, l_ch = 0;
    auto issue = [&](int slot) __attribute__((always_inline)) -> bool {
        if (l_item >= n_items) return false;
        const int slab = l_item / n_src, s = l_item - slab * n_src;
        const T* base = reinterpret_cast<const T*>(P->src[s].act) + (int64_t)slab * 32 * K + l_ch * CH;
        unsigned char* dst = slots + slot * G::SLOT;
        if ((int64_t)slab * 32 + 32 <= M) {
#pragma unroll
            for (int j = 0; j < G::NDMA; ++j) sp_dma16(base + goff[j], dst + j * 1024);
        } else {  // last slab: rows past M re-read row M - 1 (never stored)
            const int last = (int)(M - 1 - (int64_t)slab * 32);
#pragma unroll
            for (int j = 0; j < G::NDMA; ++j) {
                const int r = grow[j] < last ? grow[j] : last;
                sp_dma16(base + goff[j] + (r - grow[j]) * K, dst + j * 1024);
            }
        }
        if (++l_ch == NCH) {
            l_ch = 0;
            l_item += n_waves;
        }
        return true;
    };
    int ahead = 0;  // chunks in flight
#pragma unroll
    for (int i = 0; i < NS; ++i) ahead += issue(i) ? 1 : 0;
    // (the first slab requests above are in flight together with the stationary operands: one latency, not two)
    SP_WAIT_VM(0);
    __syncthreads();

    f32x16 acc[SP_MAXB];
    int slot = 0;
    const __amdgpu_buffer_rsrc_t orsrc = sp_rsrc(P->out, M * P->ld_out * 2);
    for (int item = w_gid; item < n_items; item += n_waves) {
        const int slab = item / n_src, s = item - slab * n_src;
        const int blk_lo = P->src[s].blk_lo, n_blk = P->src[s].n_blk;
        const bool masked = P->src[s].mask != 0 && drop.thr16 != 0;
        const int64_t m = (int64_t)slab * 32 + rl;
        const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)m);
#pragma unroll
        for (int b = 0; b < SP_MAXB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        for (int ch = 0; ch < NCH; ++ch) {
            sp_wait_chunk<G::NDMA>(ahead - 1);
            const unsigned char* sl = slots + slot * G::SLOT;
            u32x4 xf[G::KS];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) xf[ks] = *reinterpret_cast<const u32x4*>(sl + fo[ks]);
            SP_WAIT_LGKM0();
            --ahead;
            ahead += issue(slot) ? 1 : 0;
            slot = slot + 1 == NS ? 0 : slot + 1;
            if (masked) {
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) VOps<T>::drop(xf[ks], drop, rh, (uint32_t)(ch * CH + ks * 16 + 8 * h));
            }
            const unsigned char* wb = wl + ((size_t)blk_lo * KST + ch * G::KS) * 1024 + lane * 16;
#pragma unroll
            for (int b = 0; b < SP_MAXB; ++b) {
                if (b < n_blk) {
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) {
                        const u32x4 wf = *reinterpret_cast<const u32x4*>(wb + ((size_t)b * KST + ks) * 1024);
                        sp_mma1<T>(wf, xf[ks], acc[b]);
                    }
                }
            }
        }
        // epilogue: columns [col_lo, col_hi) of the rows of this slab
        const int col_lo = P->src[s].col_lo, col_hi = P->src[s].col_hi;
        const uint32_t rowoff = (uint32_t)m * (uint32_t)(P->ld_out * 2);
#pragma unroll
        for (int b = 0; b < SP_MAXB; ++b) {
            if (b < n_blk) {
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    u32x4 v;
                    sp_pack_pair<T>(acc[b], q, h, v);
                    const int col = (blk_lo + b) * 32 + 8 * q + 8 * h;
                    sp_bstore(v, orsrc, (col >= col_lo && col < col_hi) ? rowoff + (uint32_t)col * 2u : 0xFFFFFFFFu);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_sp_xres : fused MTLoRALinear launch, "activation-resident" form (short reduction: K = NKC * CH <= 192).
//   forward (T = 0):   Y = X W^T + b + Bp (alpha A D(X)^T)          one read of X, one write of Y [+ gelu(Y)] [+ P for the backward]
//   dX of a wide layer: dX = dY W + keep .* (At (alpha Bt dY^T))     (reduction over the layer's N <= 192, K output columns)
// A wave loads its 32-row slab (NKC chunks through its slot), keeps the fragments in registers, forms the projection
// P^T / Q^T = proj . act^T with the stationary projection rows (NRB 32-row blocks), converts the accumulators in place into the
// B-operand fragments of the rank step (k-permuted expansion factors, see k_pack) and then walks the output blocks of its
// workgroup's part:  acc = bias | 0 ;  acc += expand[nb] . pf ;  [acc *= keep] ;  acc += w[nb] . xf ;  store.
// The fragment reads of block nb + 1 are issued before the epilogue of block nb (they land while it converts and stores).
// The output columns are split into n_parts parts (a workgroup owns one: its weights are stationary in LDS); the parts of
// one slab group sit on the same XCD (blockIdx b -> XCD b % 8) so that the slab is re-read from that L2.
// STG: the epilogue goes through a per-wave LDS image (32 rows x 64 columns) so that a store instruction writes 8 rows x
// 128 contiguous bytes: a row-per-lane store (32 rows x 32 bytes) costs the texture-addresser ~80 cycles per instruction --
// with 20 of them per slab the store ISSUE alone is 45 us of a 57 us launch (tools/sp_ablate.sh).
// ------------------------------------------------------------------------------------------------
struct SpLinParams {
    const void* act;     // (M x K) contiguous
    const void* w;       // (n_cols x K) row-major: output column -> reduction row
    const void* proj;    // (R x K) row-major, alpha-scaled projection rows
    const void* expand;  // fragment-major k-permuted expansion factors: [ceil(n_cols/32)][estep][64][8]
    const float* bias;   // (n_cols) or null
    void* out;           // (M x ld_out)
    void* out2;          // ACT: gelu(out), same layout
    const void* gate;    // GATE: out *= gelu'(gate[m][n]), same layout as out (the Mlp's fc2 dX)
    void* pout;          // (M x ldp) projection image for the factor gradients, nullable
    int64_t ld_out, ldp, M;
    int n_cols, R, n_parts, blk_per_part;
    int n_slabs, mask_act, mask_lr, dbg;  // dbg: developer ablation bits (MTLORA_SP_DBG, -DMTL_NT_ABLATE=1 builds only): 1 no output stores, 2 no slab loads, 4 no block MFMAs, 8 no P store
    int estep, estep2;   // rank steps per block in `expand` (2 * ceil(R / 32)); k_sp_ares: the reduction length K
    int xsh, pad_;       // blockIdx -> (part, slab group): part = (b >> xsh) % n_parts, group = (b & (2^xsh - 1)) + ((b / (n_parts << xsh)) << xsh);
                         // xsh = 3: the parts of a group share an XCD (b % 8); xsh = 0: grids smaller than 8 * n_parts (desc.max_cu)
    DropoutCfg drop;
};
typedef const __attribute__((address_space(4))) SpLinParams* SpLinPtr;

constexpr int SP_STG_ROW = 64 * 2 + 8;        // row stride of a wave's 32 x 64 output image
constexpr int SP_STG = 32 * SP_STG_ROW;       // 4352 bytes per wave

// wait until at most min(n, known ladder step) vector-memory operations are outstanding (n = operations issued AFTER the one
// waited for: loads return in order, stores are counted in order with them -- the compiler's own model on gfx9 -- and an
// under-estimate only waits longer)
__device__ __forceinline__ void sp_wait_younger(int n) {
    if (n >= 48)
        SP_WAIT_VM(48);
    else if (n >= 32)
        SP_WAIT_VM(32);
    else if (n >= 20)
        SP_WAIT_VM(20);
    else if (n >= 12)
        SP_WAIT_VM(12);
    else if (n >= 6)
        SP_WAIT_VM(6);
    else
        SP_WAIT_VM(0);
}

template <typename T, int CH, int NKC, int NRB, bool ACT, bool STG, bool GATE = false>
__global__ __launch_bounds__(64 * SP_WAVES, SP_WAVES / 4) void k_sp_xres(const SpLinParams Pv) {
    typedef SpGeom<CH> G;
    constexpr int KST = NKC * G::KS;  // MFMA k-steps of the whole reduction
    constexpr int K = NKC * CH;
    constexpr int RST = 2 * NRB;      // rank steps
    (void)Pv;
    SpLinPtr P = (SpLinPtr)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, rl = lane & 31;
    const int64_t M = P->M;
    const int n_cols = P->n_cols;
    // (part, slab group) of this workgroup: b, b + 8, b + 16, ... (one XCD) are the parts of one group
    const int n_parts = P->n_parts;
    const int b = blockIdx.x, xsh = P->xsh;
    const int part = (b >> xsh) % n_parts;
    const int grp = (b & ((1 << xsh) - 1)) + ((b / (n_parts << xsh)) << xsh);
    const int n_grp = gridDim.x / n_parts;
    const int nb_all = (n_cols + 31) >> 5;
    const int bpp = P->blk_per_part;
    const int nb0 = part * bpp;
    const int nbn = nb0 + bpp <= nb_all ? bpp : nb_all - nb0;  // blocks of this part (> 0 by construction)
    // LDS: [w frags bpp*KST][proj frags NRB*KST][expand frags bpp*RST][bias 32*bpp floats][slots][output images]
    unsigned char* wl = smem;
    unsigned char* pl = wl + (size_t)bpp * KST * 1024;
    unsigned char* el = pl + (size_t)NRB * KST * 1024;
    float* bl = reinterpret_cast<float*>(el + (size_t)bpp * RST * 1024);
    unsigned char* slots = reinterpret_cast<unsigned char*>(bl + bpp * 32) + (size_t)wave * G::SLOT;
    unsigned char* img = reinterpret_cast<unsigned char*>(bl + bpp * 32) + (size_t)SP_WAVES * G::SLOT + (size_t)wave * SP_STG;
    (void)img;
    DropoutCfg drop;
    drop.seed_lo = P->drop.seed_lo;
    drop.seed_hi = P->drop.seed_hi;
    drop.thr16 = P->drop.thr16;
    drop.off = P->drop.off;
    mtl_dropout_resolve(drop);
    const bool mask_act = P->mask_act != 0 && drop.thr16 != 0, mask_lr = P->mask_lr != 0 && drop.thr16 != 0;
    const int dbg = P->dbg & NT_DBG_MASK;  // (0 at compile time unless -DMTL_NT_ABLATE=1)

    {   // stationary operands
        const T* wp = reinterpret_cast<const T*>(P->w);
        for (int f = wave; f < nbn * KST; f += SP_WAVES) {
            const int blk = f / KST, ks = f - blk * KST;
            const int row = (nb0 + blk) * 32 + rl;
            const void* g = row < n_cols ? (const void*)(wp + (int64_t)row * K + ks * 16 + 8 * h) : (const void*)g_zero16;
            sp_dma16(g, wl + (size_t)f * 1024);
        }
        const T* pp = reinterpret_cast<const T*>(P->proj);
        for (int f = wave; f < NRB * KST; f += SP_WAVES) {
            const int blk = f / KST, ks = f - blk * KST;
            const int row = blk * 32 + rl;
            const void* g = row < P->R ? (const void*)(pp + (int64_t)row * K + ks * 16 + 8 * h) : (const void*)g_zero16;
            sp_dma16(g, pl + (size_t)f * 1024);
        }
        const unsigned char* ep = reinterpret_cast<const unsigned char*>(P->expand);
        const int estep = P->estep;
        for (int f = wave; f < nbn * RST; f += SP_WAVES) {
            const int blk = f / RST, t = f - blk * RST;
            const void* g = t < estep ? (const void*)(ep + ((size_t)(nb0 + blk) * estep + t) * 1024 + lane * 16) : (const void*)g_zero16;
            sp_dma16(g, el + (size_t)f * 1024);
        }
        for (int i = tid; i < bpp * 32; i += 64 * SP_WAVES) {
            const int c = nb0 * 32 + i;
            bl[i] = (P->bias && c < n_cols) ? P->bias[c] : 0.f;
        }
    }
    int goff[G::NDMA], grow[G::NDMA], fo[G::KS];
#pragma unroll
    for (int j = 0; j < G::NDMA; ++j) {
        const int c = j * 64 + lane, row = c / G::CPR, p = c - row * G::CPR;
        grow[j] = row;
        goff[j] = row * K + G::logical(row, p) * 8;
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) fo[ks] = rl * G::ROWB + G::phys(rl, 2 * ks + h) * 16;

    const int n_slabs = P->n_slabs;
    const int stride = n_grp * SP_WAVES;
    // loader cursor (one slot per wave: the next chunk is issued as soon as the current one is in registers)
    int l_slab = grp * SP_WAVES + wave, l_ch = 0;
    auto issue = [&]() __attribute__((always_inline)) -> bool {
        if (l_slab >= n_slabs) return false;
        const T* base = reinterpret_cast<const T*>(P->act) + (int64_t)l_slab * 32 * K + l_ch * CH;
        if (dbg & 2) {
        } else if ((int64_t)l_slab * 32 + 32 <= M) {
#pragma unroll
            for (int j = 0; j < G::NDMA; ++j) sp_dma16(base + goff[j], slots + j * 1024);
        } else {
            const int last = (int)(M - 1 - (int64_t)l_slab * 32);
#pragma unroll
            for (int j = 0; j < G::NDMA; ++j) {
                const int r = grow[j] < last ? grow[j] : last;
                sp_dma16(base + goff[j] + (r - grow[j]) * K, slots + j * 1024);
            }
        }
        if (++l_ch == NKC) {
            l_ch = 0;
            l_slab += stride;
        }
        return true;
    };
    issue();
    // (the first slab requests above are in flight together with the stationary operands: one latency, not two)
    SP_WAIT_VM(0);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t orsrc = sp_rsrc(P->out, M * P->ld_out * 2), o2rsrc = sp_rsrc(P->out2, P->out2 ? M * P->ld_out * 2 : 0),
                                 prsrc = sp_rsrc(P->pout, P->pout ? M * P->ldp * 2 : 0);
    (void)o2rsrc;
    const __amdgpu_buffer_rsrc_t grsrc = sp_rsrc(const_cast<void*>(P->gate), (GATE && P->gate) ? M * P->ld_out * 2 : 0);
    (void)grsrc;
    const uint32_t ldo2 = (uint32_t)(P->ld_out * 2);
    int st_since = 0;  // store instructions certainly issued since the chunk now in flight was requested

    for (int slab = grp * SP_WAVES + wave; slab < n_slabs; slab += stride) {
        const int64_t m = (int64_t)slab * 32 + rl;
        u32x4 xf[KST];
#pragma unroll
        for (int c = 0; c < NKC; ++c) {
            sp_wait_younger(st_since);
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) xf[c * G::KS + ks] = *reinterpret_cast<const u32x4*>(slots + fo[ks]);
            SP_WAIT_LGKM0();
            issue();
            st_since = 0;
        }
        // ---- projection: accP[rb] = proj[rb] . f(act)^T
        f32x16 accP[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) accP[rb][r] = 0.f;
        const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)m);
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            u32x4 pw[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) pw[rb] = *reinterpret_cast<const u32x4*>(pl + ((size_t)rb * KST + ks) * 1024 + lane * 16);
            u32x4 a = xf[ks];
            if (mask_act) VOps<T>::drop(a, drop, rh, (uint32_t)(ks * 16 + 8 * h));
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) sp_mma1<T>(pw[rb], a, accP[rb]);
        }
        // the accumulators ARE the rank-step fragments: step t = 2 rb + t', slot s <-> register 8 t' + s
        u32x4 pf[RST];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                pf[2 * rb + t] = u32x4{mtl_pk2<T>(accP[rb][8 * t + 0], accP[rb][8 * t + 1]), mtl_pk2<T>(accP[rb][8 * t + 2], accP[rb][8 * t + 3]),
                                       mtl_pk2<T>(accP[rb][8 * t + 4], accP[rb][8 * t + 5]), mtl_pk2<T>(accP[rb][8 * t + 6], accP[rb][8 * t + 7])};
        if (P->pout && part == 0 && !(dbg & 8)) {
            const uint32_t prow = (uint32_t)m * (uint32_t)(P->ldp * 2);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    u32x4 v;
                    sp_pack_pair<T>(accP[rb], q, h, v);
                    const int col = rb * 32 + 8 * q + 8 * h;
                    sp_bstore(v, prsrc, col < P->R ? prow + (uint32_t)col * 2u : 0xFFFFFFFFu);
                }
            st_since += 2 * NRB;
        }
        // ---- output blocks of this part
        u32x4 wf[KST], ef[RST];
        f32x4 bv[4];
        auto load_blk = [&](int nb) __attribute__((always_inline)) {
            const unsigned char* eb = el + (size_t)nb * RST * 1024 + lane * 16;
#pragma unroll
            for (int t = 0; t < RST; ++t) ef[t] = *reinterpret_cast<const u32x4*>(eb + t * 1024);
            const unsigned char* wb = wl + (size_t)nb * KST * 1024 + lane * 16;
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) wf[ks] = *reinterpret_cast<const u32x4*>(wb + ks * 1024);
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<const f32x4*>(bl + nb * 32 + 8 * q + 4 * h);
        };
        constexpr bool PF = KST <= 8 && SP_WAVES <= 8;  // register room for the next block's fragments next to the epilogue's temporaries
        if constexpr (PF) load_blk(0);
        u32x4 hv[4];  // GATE: pre-activations of the block (pair) about to be produced, loaded before its MFMAs
        (void)hv;
        for (int nb = 0; nb < nbn; ++nb) {
            if constexpr (!PF) load_blk(nb);
            if constexpr (GATE) {
                if constexpr (STG) {
                    if (!(nb & 1)) {  // first block of a pair: the 4 x (8 rows x 128 bytes) the flush will store
                        const int c16 = lane & 7, col = (nb0 + nb) * 32 + c16 * 8;
                        const bool colok = col < n_cols && (c16 < 4 || nb + 1 < nbn);
                        const uint32_t o0 = ((uint32_t)slab * 32u + (uint32_t)(lane >> 3)) * ldo2 + (uint32_t)col * 2u;
#pragma unroll
                        for (int it = 0; it < 4; ++it)
                            hv[it] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, (int)(colok ? o0 + (uint32_t)(it * 8) * ldo2 : 0xFFFFFFFFu), 0, 0);
                        st_since += 4;
                    }
                } else {
                    const uint32_t rowoff = (uint32_t)m * ldo2;
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        const int col = (nb0 + nb) * 32 + 8 * q + 8 * h;
                        hv[q >> 1] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, (int)(col < n_cols ? rowoff + (uint32_t)col * 2u : 0xFFFFFFFFu), 0, 0);
                    }
                    st_since += 2;
                }
            }
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * q + e] = bv[q][e];
#pragma unroll
            for (int t = 0; t < RST; ++t) sp_mma1<T>(ef[t], pf[t], acc);
            const int col0 = (nb0 + nb) * 32;
            if (mask_lr) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = col0 + 8 * q + 4 * h;
                    const uint32_t h0 = mtl_dropout_pairbits(drop, rh, (uint32_t)n);
                    const uint32_t h1 = mtl_dropout_pairbits(drop, rh, (uint32_t)(n + 2));
                    if ((h0 & 0xFFFFu) < drop.thr16) acc[4 * q + 0] = 0.f;
                    if ((h0 >> 16) < drop.thr16) acc[4 * q + 1] = 0.f;
                    if ((h1 & 0xFFFFu) < drop.thr16) acc[4 * q + 2] = 0.f;
                    if ((h1 >> 16) < drop.thr16) acc[4 * q + 3] = 0.f;
                }
            }
            if (!(dbg & 4)) {
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) sp_mma1<T>(wf[ks], xf[ks], acc);
            }
            if constexpr (PF) {
                if (nb + 1 < nbn) load_blk(nb + 1);  // lands while this block is converted and stored
            }
            if constexpr (STG) {
                // image columns (nb & 1) * 32 + 8 q + 4 h .. + 4 of row rl; flushed after the odd block of a pair / the last block
                unsigned char* ip = img + rl * SP_STG_ROW + ((nb & 1) * 32 + 4 * h) * 2;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<u32x2*>(ip + q * 16) =
                        u32x2{mtl_pk2<T>(acc[4 * q + 0], acc[4 * q + 1]), mtl_pk2<T>(acc[4 * q + 2], acc[4 * q + 3])};
                if ((nb & 1) || nb + 1 == nbn) {
                    const int c0 = (nb0 + (nb & ~1)) * 32;  // first column of the pair
                    SP_WAIT_LGKM0();
                    __builtin_amdgcn_wave_barrier();
                    u32x4 v[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const u32x4*>(img + (it * 8 + (lane >> 3)) * SP_STG_ROW + (lane & 7) * 16);
                    const int c16 = lane & 7, col = c0 + c16 * 8;
                    const bool colok = col < n_cols && (c16 < 4 || (nb & 1));
                    const uint32_t o0 = ((uint32_t)slab * 32u + (uint32_t)(lane >> 3)) * ldo2 + (uint32_t)col * 2u;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const uint32_t off = (colok && !(dbg & 1)) ? o0 + (uint32_t)(it * 8) * ldo2 : 0xFFFFFFFFu;
                        if constexpr (GATE) {  // the rounded gradient times gelu'(pre-activation), rounded once (as ATen does)
                            v[it] = mtl_gelu_gate_pk4<T, true>(v[it], hv[it]);
                        }
                        sp_bstore(v[it], orsrc, off);
                        if constexpr (ACT) {
                            const u32x4 av = mtl_gelu_pk4<T, true>(v[it]);
                            sp_bstore(av, o2rsrc, off);
                        }
                    }
                    st_since += ACT ? 8 : 4;
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                const uint32_t rowoff = (uint32_t)m * ldo2;
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    u32x4 v;
                    sp_pack_pair<T>(acc, q, h, v);
                    const int col = col0 + 8 * q + 8 * h;
                    const uint32_t off = (col < n_cols && !(dbg & 1)) ? rowoff + (uint32_t)col * 2u : 0xFFFFFFFFu;
                    if constexpr (GATE) {
                        v = mtl_gelu_gate_pk4<T, true>(v, hv[q >> 1]);
                    }
                    sp_bstore(v, orsrc, off);
                    if constexpr (ACT) {
                        const u32x4 av = mtl_gelu_pk4<T, true>(v);
                        sp_bstore(av, o2rsrc, off);
                    }
                }
                st_since += ACT ? 4 : 2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_sp_ares : fused MTLoRALinear launch, "accumulator-resident" form (few output columns: n_cols <= 32 * NOB per part, any
// reduction length K = n_ch * CH): the dX of a T = 0 layer whose output is wider than its input,
//     dX = dY W + keep .* ((alpha dY B) A)        reduction over the layer's N, K <= 192 output columns,
// in ONE pass over dY.  A wave streams its 32-row slab of the activation chunk by chunk through its slot; every chunk feeds
// the NOB output accumulators (w fragments) AND the NRB projection accumulators (proj fragments: Q^T = alpha Bt dY^T); after
// the last chunk the projection accumulators become the rank-step fragments (as in k_sp_xres), the rank part of every
// output block is formed in a scratch accumulator, masked, added, and the block is stored.  [mask_act: the projection sees
// the dropout-masked activation -- the forward of a narrow layer, Y = X W^T + b + Bp (alpha A D(X)^T).]
// ------------------------------------------------------------------------------------------------
// EXPG: the expansion-factor fragments are read from global memory (L2-resident: <= 12 KB shared by every workgroup) instead of LDS,
// the block after next being requested while the current one is multiplied -- frees the 12 KB that let a reduction of 384 with a
// 64-wide rank (stage-0 fc2 forward / fc1 dX: 72 KB of weight fragments + 48 KB of projection fragments) fit next to the slots.
template <typename T, int CH, int NOB, int NRB, bool EXPG = false>
__global__ __launch_bounds__(64 * SP_WAVES, SP_WAVES / 4) void k_sp_ares(const SpLinParams Pv) {
    typedef SpGeom<CH> G;
    constexpr int RST = 2 * NRB;
    (void)Pv;
    SpLinPtr P = (SpLinPtr)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, rl = lane & 31;
    const int64_t M = P->M;
    const int n_cols = P->n_cols;
    const int K = P->estep2;              // reduction length (elements), a multiple of CH
    const int KST = K >> 4, NCH = K / CH;
    const int n_parts = P->n_parts;
    const int b = blockIdx.x, xsh = P->xsh;
    const int part = (b >> xsh) % n_parts;
    const int grp = (b & ((1 << xsh) - 1)) + ((b / (n_parts << xsh)) << xsh);
    const int n_grp = gridDim.x / n_parts;
    const int nb_all = (n_cols + 31) >> 5;
    constexpr int bpp = NOB;              // blocks per part (blocks past the last column have zero weights and are not stored)
    const int nb0 = part * bpp;
    (void)nb_all;
    // LDS: [w frags bpp*KST][proj frags NRB*KST][expand frags bpp*RST (not with EXPG)][bias 32*bpp floats][slots]
    unsigned char* wl = smem;
    unsigned char* pl = wl + (size_t)bpp * KST * 1024;
    unsigned char* el = pl + (size_t)NRB * KST * 1024;
    float* bl = reinterpret_cast<float*>(el + (EXPG ? (size_t)0 : (size_t)bpp * RST * 1024));
    unsigned char* slots = reinterpret_cast<unsigned char*>(bl + bpp * 32) + (size_t)wave * G::SLOT;
    DropoutCfg drop;
    drop.seed_lo = P->drop.seed_lo;
    drop.seed_hi = P->drop.seed_hi;
    drop.thr16 = P->drop.thr16;
    drop.off = P->drop.off;
    mtl_dropout_resolve(drop);
    const bool mask_act = P->mask_act != 0 && drop.thr16 != 0, mask_lr = P->mask_lr != 0 && drop.thr16 != 0;
    const int dbg = P->dbg & NT_DBG_MASK;  // (0 at compile time unless -DMTL_NT_ABLATE=1)

    {   // stationary operands
        const T* wp = reinterpret_cast<const T*>(P->w);
        for (int f = wave; f < bpp * KST; f += SP_WAVES) {
            const int blk = f / KST, ks = f - blk * KST;
            const int row = (nb0 + blk) * 32 + rl;
            const void* g = row < n_cols ? (const void*)(wp + (int64_t)row * K + ks * 16 + 8 * h) : (const void*)g_zero16;
            sp_dma16(g, wl + (size_t)f * 1024);
        }
        const T* pp = reinterpret_cast<const T*>(P->proj);
        for (int f = wave; f < NRB * KST; f += SP_WAVES) {
            const int blk = f / KST, ks = f - blk * KST;
            const int row = blk * 32 + rl;
            const void* g = row < P->R ? (const void*)(pp + (int64_t)row * K + ks * 16 + 8 * h) : (const void*)g_zero16;
            sp_dma16(g, pl + (size_t)f * 1024);
        }
        const unsigned char* ep = reinterpret_cast<const unsigned char*>(P->expand);
        const int estep = P->estep;
        for (int f = wave; !EXPG && f < bpp * RST; f += SP_WAVES) {
            const int blk = f / RST, t = f - blk * RST;
            const void* g = (t < estep && (nb0 + blk) * 32 < n_cols) ? (const void*)(ep + ((size_t)(nb0 + blk) * estep + t) * 1024 + lane * 16) : (const void*)g_zero16;
            sp_dma16(g, el + (size_t)f * 1024);
        }
        for (int i = tid; i < bpp * 32; i += 64 * SP_WAVES) {
            const int c = nb0 * 32 + i;
            bl[i] = (P->bias && c < n_cols) ? P->bias[c] : 0.f;
        }
    }
    int goff[G::NDMA], grow[G::NDMA], fo[G::KS];
#pragma unroll
    for (int j = 0; j < G::NDMA; ++j) {
        const int c = j * 64 + lane, row = c / G::CPR, p = c - row * G::CPR;
        grow[j] = row;
        goff[j] = row * K + G::logical(row, p) * 8;
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) fo[ks] = rl * G::ROWB + G::phys(rl, 2 * ks + h) * 16;

    const int n_slabs = P->n_slabs;
    const int stride = n_grp * SP_WAVES;
    int l_slab = grp * SP_WAVES + wave, l_ch = 0;
    auto issue = [&]() __attribute__((always_inline)) -> bool {
        if (l_slab >= n_slabs) return false;
        const T* base = reinterpret_cast<const T*>(P->act) + (int64_t)l_slab * 32 * K + l_ch * CH;
        if (dbg & 2) {
        } else if ((int64_t)l_slab * 32 + 32 <= M) {
#pragma unroll
            for (int j = 0; j < G::NDMA; ++j) sp_dma16(base + goff[j], slots + j * 1024);
        } else {
            const int last = (int)(M - 1 - (int64_t)l_slab * 32);
#pragma unroll
            for (int j = 0; j < G::NDMA; ++j) {
                const int r = grow[j] < last ? grow[j] : last;
                sp_dma16(base + goff[j] + (r - grow[j]) * K, slots + j * 1024);
            }
        }
        if (++l_ch == NCH) {
            l_ch = 0;
            l_slab += stride;
        }
        return true;
    };
    issue();
    // (the first slab requests above are in flight together with the stationary operands: one latency, not two)
    SP_WAIT_VM(0);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t orsrc = sp_rsrc(P->out, M * P->ld_out * 2), prsrc = sp_rsrc(P->pout, P->pout ? M * P->ldp * 2 : 0);
    const uint32_t ldo2 = (uint32_t)(P->ld_out * 2);
    int st_since = 0;

    for (int slab = grp * SP_WAVES + wave; slab < n_slabs; slab += stride) {
        const int64_t m = (int64_t)slab * 32 + rl;
        const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)m);
        f32x16 acc[NOB], accP[NRB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                bv = *reinterpret_cast<const f32x4*>(bl + ob * 32 + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[ob][4 * q + e] = bv[e];
            }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) accP[rb][r] = 0.f;
        for (int ch = 0; ch < NCH; ++ch) {
            sp_wait_younger(st_since);
            u32x4 xf[G::KS];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) xf[ks] = *reinterpret_cast<const u32x4*>(slots + fo[ks]);
            SP_WAIT_LGKM0();
            issue();
            st_since = 0;
            const unsigned char* wb = wl + (size_t)(ch * G::KS) * 1024 + lane * 16;
            const unsigned char* pb = pl + (size_t)(ch * G::KS) * 1024 + lane * 16;
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                if (ks & 1) __builtin_amdgcn_sched_barrier(0);  // at most two k-steps' fragments in flight (register room)
                u32x4 wfr[NOB], pfr[NRB];
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob)
                    wfr[ob] = *reinterpret_cast<const u32x4*>(wb + ((size_t)ob * KST + ks) * 1024);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) pfr[rb] = *reinterpret_cast<const u32x4*>(pb + ((size_t)rb * KST + ks) * 1024);
                u32x4 a = xf[ks];
                if (!(dbg & 4)) {
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob)
                        sp_mma1<T>(wfr[ob], a, acc[ob]);
                }
                if (mask_act) VOps<T>::drop(a, drop, rh, (uint32_t)(ch * CH + ks * 16 + 8 * h));
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) sp_mma1<T>(pfr[rb], a, accP[rb]);
            }
        }
        u32x4 pf[RST];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                pf[2 * rb + t] = u32x4{mtl_pk2<T>(accP[rb][8 * t + 0], accP[rb][8 * t + 1]), mtl_pk2<T>(accP[rb][8 * t + 2], accP[rb][8 * t + 3]),
                                       mtl_pk2<T>(accP[rb][8 * t + 4], accP[rb][8 * t + 5]), mtl_pk2<T>(accP[rb][8 * t + 6], accP[rb][8 * t + 7])};
        if (P->pout && part == 0 && !(dbg & 8)) {
            const uint32_t prow = (uint32_t)m * (uint32_t)(P->ldp * 2);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    u32x4 v;
                    sp_pack_pair<T>(accP[rb], q, h, v);
                    const int col = rb * 32 + 8 * q + 8 * h;
                    sp_bstore(v, prsrc, col < P->R ? prow + (uint32_t)col * 2u : 0xFFFFFFFFu);
                }
            st_since += 2 * NRB;
        }
        const uint32_t rowoff = (uint32_t)m * ldo2;
        u32x4 ef[RST], efn[RST];  // expansion fragments of the current / the next block
        auto load_ef = [&](u32x4 (&e)[RST], int ob) __attribute__((always_inline)) {
            if constexpr (EXPG) {
                const unsigned char* ep = reinterpret_cast<const unsigned char*>(P->expand);
                const int estep = P->estep;
                const bool live = (nb0 + ob) * 32 < n_cols;
#pragma unroll
                for (int t = 0; t < RST; ++t)
                    e[t] = (live && t < estep) ? *reinterpret_cast<const u32x4*>(ep + ((size_t)(nb0 + ob) * estep + t) * 1024 + lane * 16)
                                               : u32x4{0u, 0u, 0u, 0u};
            } else {
                const unsigned char* eb = el + (size_t)ob * RST * 1024 + lane * 16;
#pragma unroll
                for (int t = 0; t < RST; ++t) e[t] = *reinterpret_cast<const u32x4*>(eb + t * 1024);
            }
        };
        if constexpr (EXPG) {
            load_ef(efn, 0);
            st_since += RST;  // (vector-memory operations behind the chunk in flight: an under-count would only wait longer)
        }
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            {
                const int col0 = (nb0 + ob) * 32;
                if constexpr (EXPG) {
#pragma unroll
                    for (int t = 0; t < RST; ++t) ef[t] = efn[t];
                    if (ob + 1 < NOB) {
                        load_ef(efn, ob + 1);
                        st_since += RST;
                    }
                } else {
                    load_ef(ef, ob);
                }
                if (mask_lr) {
                    f32x16 lr;
#pragma unroll
                    for (int r = 0; r < 16; ++r) lr[r] = 0.f;
#pragma unroll
                    for (int t = 0; t < RST; ++t) sp_mma1<T>(ef[t], pf[t], lr);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = col0 + 8 * q + 4 * h;
                        const uint32_t h0 = mtl_dropout_pairbits(drop, rh, (uint32_t)n);
                        const uint32_t h1 = mtl_dropout_pairbits(drop, rh, (uint32_t)(n + 2));
                        if ((h0 & 0xFFFFu) >= drop.thr16) acc[ob][4 * q + 0] += lr[4 * q + 0];
                        if ((h0 >> 16) >= drop.thr16) acc[ob][4 * q + 1] += lr[4 * q + 1];
                        if ((h1 & 0xFFFFu) >= drop.thr16) acc[ob][4 * q + 2] += lr[4 * q + 2];
                        if ((h1 >> 16) >= drop.thr16) acc[ob][4 * q + 3] += lr[4 * q + 3];
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < RST; ++t) sp_mma1<T>(ef[t], pf[t], acc[ob]);
                }
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    u32x4 v;
                    sp_pack_pair<T>(acc[ob], q, h, v);
                    const int col = col0 + 8 * q + 8 * h;
                    sp_bstore(v, orsrc, (col < n_cols && !(dbg & 1)) ? rowoff + (uint32_t)col * 2u : 0xFFFFFFFFu);
                }
                st_since += 2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_sp_projsum : the Q pass of a layer WITH task outputs, fused with the sum of the output gradients:
//     Q[:, seg_o] = alpha_o dY_o B_o  (o = shared, tasks)      and      G = sum_o dY_o   (fp32 sum, rounded once)
// in ONE pass over the 1 + T gradient tensors.  The tiled path reads them twice here (k_sum or the multi-source staging of the dX
// kernel, and the Q pass); with G written next to Q the dX launch becomes single-source (dX = G W + keep .* (Q_s A_s), dX_t =
// Q_t A_t) and k_sum disappears.  Work item = slab of 32 rows; per reduction chunk the wave walks the sources (one DMA chunk
// each, same slot ring as k_sp_proj), adds the fragments into the chunk's fp32 sum and feeds that source's projection blocks
// (shared: <= 2 blocks of 32 rank rows; the task segments lie in <= 4 blocks, the tasks of one block sharing its accumulator through
// row-masked projection fragments).  Any number of task outputs up to MTLORA_MAX_TASKS.
// ------------------------------------------------------------------------------------------------
constexpr int SP_PS_MAXT = 4;  // 32-row blocks the task segments may span (the tasks of a block share one accumulator)
template <typename T, int CH, int NS>
__global__ __launch_bounds__(64 * SP_WAVES, SP_WAVES / 4) void k_sp_projsum(const SpProjParams Pv, T* __restrict__ gsum) {
    typedef SpGeom<CH> G;
    (void)Pv;
    SpProjPtr P = (SpProjPtr)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, rl = lane & 31;
    const int K = P->K, KST = K >> 4, NCH = K / CH;
    const int n_src = P->n_src, n_slabs = P->n_slabs;
    const int64_t M = P->M;
    unsigned char* wl = smem;
    unsigned char* slots = smem + (size_t)P->n_blk_total * KST * 1024 + (size_t)wave * NS * G::SLOT;
    {
        const T* wp = reinterpret_cast<const T*>(P->wproj);
        const int nfrag = P->n_blk_total * KST;
        for (int f = wave; f < nfrag; f += SP_WAVES) {
            const int blk = f / KST, ks = f - blk * KST;
            const int row = blk * 32 + rl;
            const void* g = row < P->Rw ? (const void*)(wp + (int64_t)row * K + ks * 16 + 8 * h) : (const void*)g_zero16;
            sp_dma16(g, wl + (size_t)f * 1024);
        }
    }
    int goff[G::NDMA], grow[G::NDMA], fo[G::KS];
#pragma unroll
    for (int j = 0; j < G::NDMA; ++j) {
        const int c = j * 64 + lane, row = c / G::CPR, p = c - row * G::CPR;
        grow[j] = row;
        goff[j] = row * K + G::logical(row, p) * 8;
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) fo[ks] = rl * G::ROWB + G::phys(rl, 2 * ks + h) * 16;

    // task sources 1 .. n_src - 1: consecutive rank segments, each inside ONE 32-row block; blocks tblk0 .. tblk0 + n_tblk - 1
    const int tblk0 = n_src > 1 ? P->src[1].blk_lo : 0, n_tblk = n_src > 1 ? P->src[n_src - 1].blk_lo - tblk0 + 1 : 0;
    const int t_lo = n_src > 1 ? P->src[1].col_lo : 0, t_hi = n_src > 1 ? P->src[n_src - 1].col_hi : 0;
    const int n_waves = gridDim.x * SP_WAVES;
    const int w_gid = blockIdx.x * SP_WAVES + wave;
    // loader cursor over (slab, chunk, source)
    int l_slab = w_gid, l_ch = 0, l_s = 0;
    auto issue = [&](int slot) __attribute__((always_inline)) -> bool {
        if (l_slab >= n_slabs) return false;
        const T* base = reinterpret_cast<const T*>(P->src[l_s].act) + (int64_t)l_slab * 32 * K + l_ch * CH;
        unsigned char* dst = slots + slot * G::SLOT;
        if ((int64_t)l_slab * 32 + 32 <= M) {
#pragma unroll
            for (int j = 0; j < G::NDMA; ++j) sp_dma16(base + goff[j], dst + j * 1024);
        } else {
            const int last = (int)(M - 1 - (int64_t)l_slab * 32);
#pragma unroll
            for (int j = 0; j < G::NDMA; ++j) {
                const int r = grow[j] < last ? grow[j] : last;
                sp_dma16(base + goff[j] + (r - grow[j]) * K, dst + j * 1024);
            }
        }
        if (++l_s == n_src) {
            l_s = 0;
            if (++l_ch == NCH) {
                l_ch = 0;
                l_slab += n_waves;
            }
        }
        return true;
    };
    int ahead = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) ahead += issue(i) ? 1 : 0;
    // (the first slab requests above are in flight together with the stationary operands: one latency, not two)
    SP_WAIT_VM(0);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t orsrc = sp_rsrc(P->out, M * P->ld_out * 2), grsrc = sp_rsrc(gsum, M * (int64_t)K * 2);

    int slot = 0;
    for (int slab = w_gid; slab < n_slabs; slab += n_waves) {
        const uint32_t m = (uint32_t)slab * 32u + (uint32_t)rl;
        f32x16 accS[2], accT[SP_PS_MAXT];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accS[0][r] = accS[1][r] = 0.f;
#pragma unroll
            for (int t = 0; t < SP_PS_MAXT; ++t) accT[t][r] = 0.f;
        }
        for (int ch = 0; ch < NCH; ++ch) {
            float gs[G::KS][8];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) gs[ks][e] = 0.f;
            for (int s = 0; s < n_src; ++s) {
                {
                    sp_wait_chunk<G::NDMA>(ahead - 1);
                    const unsigned char* sl = slots + slot * G::SLOT;
                    u32x4 xf[G::KS];
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) xf[ks] = *reinterpret_cast<const u32x4*>(sl + fo[ks]);
                    SP_WAIT_LGKM0();
                    --ahead;
                    ahead += issue(slot) ? 1 : 0;
                    slot = slot + 1 == NS ? 0 : slot + 1;
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) {
                        float f[8];
                        VOps<T>::unpack(xf[ks], f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) gs[ks][e] += f[e];
                    }
                    const int blk_lo = P->src[s].blk_lo;
                    const unsigned char* wb = wl + ((size_t)blk_lo * KST + ch * G::KS) * 1024 + lane * 16;
                    if (s == 0) {
                        const int nb = P->src[0].n_blk;
#pragma unroll
                        for (int ks = 0; ks < G::KS; ++ks) {
                            sp_mma1<T>(*reinterpret_cast<const u32x4*>(wb + (size_t)ks * 1024), xf[ks], accS[0]);
                            if (nb > 1) sp_mma1<T>(*reinterpret_cast<const u32x4*>(wb + ((size_t)KST + ks) * 1024), xf[ks], accS[1]);
                        }
                    } else {
                        // the tasks of one 32-row block share ONE accumulator: this task's projection rows only (the other rows of
                        // the fragment are zeroed), so its product lands in its own rank rows and nowhere else
                        const int tb = blk_lo - tblk0;
                        const bool mine = blk_lo * 32 + rl >= P->src[s].col_lo && blk_lo * 32 + rl < P->src[s].col_hi;
#pragma unroll
                        for (int ks = 0; ks < G::KS; ++ks) {
                            u32x4 wf = *reinterpret_cast<const u32x4*>(wb + (size_t)ks * 1024);
                            wf = mine ? wf : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                            for (int b = 0; b < SP_PS_MAXT; ++b)
                                if (b == tb) sp_mma1<T>(wf, xf[ks], accT[b]);
                        }
                    }
                }
            }
            // the chunk of G: lane (row m, half h) holds columns ch*CH + ks*16 + 8 h .. + 8 of its row
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                u32x4 v;
                if constexpr (sizeof(T) == 2) v = u32x4{mtl_pk2<T>(gs[ks][0], gs[ks][1]), mtl_pk2<T>(gs[ks][2], gs[ks][3]), mtl_pk2<T>(gs[ks][4], gs[ks][5]),
                                                        mtl_pk2<T>(gs[ks][6], gs[ks][7])};
                sp_bstore(v, grsrc, m * (uint32_t)(K * 2) + (uint32_t)(ch * CH + ks * 16 + 8 * h) * 2u);
            }
        }
        // Q segments: the shared one from accS, the task blocks from accT (columns of the task range only)
        const uint32_t rowoff = m * (uint32_t)(P->ld_out * 2);
        {
            const int col_lo = P->src[0].col_lo, col_hi = P->src[0].col_hi, blk_lo = P->src[0].blk_lo;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (b == 0 || P->src[0].n_blk > 1) {
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        u32x4 v;
                        sp_pack_pair<T>(accS[b], q, h, v);
                        const int col = (blk_lo + b) * 32 + 8 * q + 8 * h;
                        sp_bstore(v, orsrc, (col >= col_lo && col < col_hi) ? rowoff + (uint32_t)col * 2u : 0xFFFFFFFFu);
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < SP_PS_MAXT; ++b) {
            if (b < n_tblk) {
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    u32x4 v;
                    sp_pack_pair<T>(accT[b], q, h, v);
                    const int col = (tblk0 + b) * 32 + 8 * q + 8 * h;
                    sp_bstore(v, orsrc, (col >= t_lo && col < t_hi) ? rowoff + (uint32_t)col * 2u : 0xFFFFFFFFu);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_sp_tn : the factor gradients as a wave-streaming pass (replaces the tiled k_tn for 16-bit types):
//     Out[a][b] = sum_m SrcA[m][a0 + a] * SrcB[m][b]     dA_o = Q_o^T D(X_o) (r_o x K),   dB_o^T = P_o^T dY_o (r_o x N)
// SrcA is the narrow (rank-side) matrix, SrcB the wide one.  Work unit "pp" = (problem, 64-column tile of the narrow window,
// part of 32 NB wide columns); a pp is served by G workgroups, every wave of which owns 32-row slabs (slab = wave id + i * 8 G) and keeps the
// 2 x NB accumulator blocks (32 x 32 fp32 each) of its pp in registers for the whole launch -- nothing is written until the end.
// Per slab the wave DMAs the narrow piece (32 rows x <= 128 B) and the wide piece (32 rows x 64 NB B) into its private slot and
// reads both back TRANSPOSED (ds_read_b64_tr_b16: a lane gets 8 consecutive m of one column = the k axis of the MFMA), so the
// same slot image serves as A operand (narrow) and B operand (wide).  LDS layout (dense rows, 16-byte chunks permuted on the
// source side so that the 4 rows x 64 B a transposed read touches per cycle cover all 64 banks):
//     narrow rows of 128 B: physical chunk = logical ^ (4 * ((row >> 1) & 1));
//     wide rows of 192 B (NB = 3): identity (row stride = 48 dwords);  256 B (NB = 4): logical ^ (4 * (row & 3)).
// The dropout keep-mask of D(X) is applied in LDS by the lane that loaded the chunk (once per element).  At the end the 8 waves
// of a workgroup are summed through LDS in a fixed order and the workgroup's partial goes to HBM in raw accumulator layout
// ([block][reg][lane]); k_sp_tn_reduce sums the G partials of a pp in a fixed order and scatters to dA / dB (deterministic).
// ------------------------------------------------------------------------------------------------
constexpr int SP_TN_MAP = 8 * 40;
struct SpTnProb {
    const void* A;   // narrow: (M x lda), window [a0, a0 + Na)
    const void* B;   // wide:   (M x ldb), columns [0, Nb)
    int64_t lda, ldb;
    int a0, Na, Nb, b_mask;
    int tiles_a, parts, pp_lo, transpose;
    float* out;      // element (a, b) at out[a * ldo + b], or out[b * ldo + a] when transpose
    int out_a, out_b, ldo, pad_;
};
struct SpTnParams {
    SpTnProb p[2 * MAXO];
    int n_prob, n_pp, G, n_slabs;
    int64_t M;
    float* part;  // [pp][g][2 * NB blocks][1024]
    DropoutCfg drop;
    uint32_t map[SP_TN_MAP];  // blockIdx -> pp | g << 16 (0xFFFFFFFF: idle)
};
typedef const __attribute__((address_space(4))) SpTnParams* SpTnPtr;

constexpr int SP_TN_NARROW = 32 * 128;  // narrow area of a slot (32 rows x 64 columns)

template <typename T, int NB, int NS>
__global__ __launch_bounds__(64 * SP_WAVES, SP_WAVES / 4) void k_sp_tn(const SpTnParams Pv) {
    constexpr int RB = NB * 64;            // wide row bytes
    constexpr int CPR = NB * 4;            // 16-byte chunks per wide row
    constexpr int NDW = NB * 2;            // DMA instructions of the wide piece
    constexpr int NDN = 4;                 // DMA instructions of the narrow piece
    constexpr int SLOT = SP_TN_NARROW + 32 * RB;
    (void)Pv;
    SpTnPtr P = (SpTnPtr)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- which pp / row group: a host-built table.  The pps that share a narrow slab (same problem, same narrow tile, same row
    // group) are placed on ONE XCD (blockIdx b -> XCD b % 8) next to each other in dispatch order, so the narrow slab is re-read
    // from that L2 (10-25 % of the launch time on the T = 4 layers), and the XCDs are loaded evenly (<= 32 workgroups each)
    const int G = P->G;
    const uint32_t ent = P->map[blockIdx.x];
    if (ent == 0xFFFFFFFFu) return;
    const int pp = (int)(ent & 0xFFFFu), g = (int)(ent >> 16);
    int pi = 0;
    for (int i = 1; i < P->n_prob; ++i)
        if (pp >= P->p[i].pp_lo) pi = i;
    const int local = pp - P->p[pi].pp_lo;
    const int parts = P->p[pi].parts;
    const int ta = local / parts, pc = local - ta * parts;
    const int64_t lda = P->p[pi].lda, ldb = P->p[pi].ldb;
    const int na_valid = min(64, P->p[pi].Na - ta * 64);
    const bool two = na_valid > 32;
    const int nch_valid = (na_valid + 7) >> 3;
    const T* Ap = reinterpret_cast<const T*>(P->p[pi].A) + P->p[pi].a0 + ta * 64;
    const T* Bp = reinterpret_cast<const T*>(P->p[pi].B) + pc * (NB * 32);
    const int64_t M = P->M;
    const int n_slabs = P->n_slabs;
    DropoutCfg drop;
    drop.seed_lo = P->drop.seed_lo;
    drop.seed_hi = P->drop.seed_hi;
    drop.thr16 = P->drop.thr16;
    drop.off = P->drop.off;
    mtl_dropout_resolve(drop);
    const bool bmask = P->p[pi].b_mask != 0 && drop.thr16 != 0;

    unsigned char* slots = smem + (size_t)wave * NS * SLOT;
    // ---- per-lane DMA constants (32-bit byte offsets from the slab's first row: scalar base + vector offset addressing).  Narrow
    // chunks past the window re-read its last valid chunk: they only feed accumulator rows nobody reads.
    uint32_t goA[NDN], goB[NDW];
#pragma unroll
    for (int j = 0; j < NDN; ++j) {
        const int c = j * 64 + lane, row = c >> 3, p = c & 7;
        int q = p ^ (((row >> 1) & 1) << 2);
        q = q < nch_valid ? q : nch_valid - 1;
        goA[j] = (uint32_t)(row * (int)lda + q * 8) * 2u;
    }
#pragma unroll
    for (int j = 0; j < NDW; ++j) {
        const int c = j * 64 + lane, row = c / CPR, p = c - row * CPR;
        const int q = NB == 4 ? (p ^ ((row & 3) << 2)) : p;
        goB[j] = (uint32_t)(row * (int)ldb + q * 8) * 2u;
    }
    auto row_b = [&](int j) __attribute__((always_inline)) { return (j * 64 + lane) / CPR; };
    auto col_b = [&](int j) __attribute__((always_inline)) {
        const int c = j * 64 + lane, row = c / CPR, p = c - row * CPR;
        return pc * (NB * 32) + (NB == 4 ? (p ^ ((row & 3) << 2)) : p) * 8;
    };
    // ---- transposed fragment addressing: read j (0..3) of a fragment covers rows rj = 16 (j >> 1) + 8 h + 4 (j & 1) + (i >> 2)
    // of the slab, 8 bytes at logical chunk 4 blk + 2 (g4 & 1) + ((i & 3) >> 1), byte (i & 1) * 8 of that chunk
    const int g4 = lane >> 4, i16 = lane & 15, hh = g4 >> 1;
    const int clow = 2 * (g4 & 1) + ((i16 & 3) >> 1), cbyte = (i16 & 1) * 8;
    int offA[4], offB[4], swB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rj = 16 * (j >> 1) + 8 * hh + 4 * (j & 1) + (i16 >> 2);
        offA[j] = rj * 128 + ((clow ^ (((rj >> 1) & 1) << 2)) << 4) + cbyte;  // block 0; block 1: ^ 64 (chunk bit 2)
        offB[j] = SP_TN_NARROW + rj * RB + cbyte;
        swB[j] = NB == 4 ? (rj & 3) : 0;
    }
    auto frag_a = [&](const unsigned char* sl, int blk) __attribute__((always_inline)) -> Frag<T> {
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sl + (offA[j] ^ (blk << 6))));
            u32x2 u = __builtin_bit_cast(u32x2, v);
            w[2 * j] = u[0];
            w[2 * j + 1] = u[1];
        }
        Frag<T> f;
        f.v[0] = u32x4{w[0], w[1], w[2], w[3]};
        f.v[1] = u32x4{w[4], w[5], w[6], w[7]};
        return f;
    };
    auto frag_b = [&](const unsigned char* sl, int blk) __attribute__((always_inline)) -> Frag<T> {
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = (((blk ^ swB[j]) << 2) | clow) << 4;
            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sl + offB[j] + ch));
            u32x2 u = __builtin_bit_cast(u32x2, v);
            w[2 * j] = u[0];
            w[2 * j + 1] = u[1];
        }
        Frag<T> f;
        f.v[0] = u32x4{w[0], w[1], w[2], w[3]};
        f.v[1] = u32x4{w[4], w[5], w[6], w[7]};
        return f;
    };

    const int stride = G * SP_WAVES;
    const int w_gid = g * SP_WAVES + wave;
    int l_slab = w_gid;
    auto issue = [&](int slot) __attribute__((always_inline)) -> bool {
        if (l_slab >= n_slabs) return false;
        const int64_t m0 = (int64_t)l_slab * 32;
        const unsigned char* a = reinterpret_cast<const unsigned char*>(Ap + m0 * lda);
        const unsigned char* b = reinterpret_cast<const unsigned char*>(Bp + m0 * ldb);
        unsigned char* dst = slots + slot * SLOT;
        if (m0 + 32 <= M) {
#pragma unroll
            for (int j = 0; j < NDN; ++j) sp_dma16(a + goA[j], dst + j * 1024);
#pragma unroll
            for (int j = 0; j < NDW; ++j) sp_dma16(b + goB[j], dst + SP_TN_NARROW + j * 1024);
        } else {  // last slab: rows past M re-read row M - 1; the narrow rows are zeroed in LDS before they are used
            const int last = (int)(M - m0) - 1;
#pragma unroll
            for (int j = 0; j < NDN; ++j) {
                const int row = (j * 64 + lane) >> 3;
                sp_dma16(a + (goA[j] - (uint32_t)((row > last ? row - last : 0) * (int)lda) * 2u), dst + j * 1024);
            }
#pragma unroll
            for (int j = 0; j < NDW; ++j) {
                const int row = row_b(j);
                sp_dma16(b + (goB[j] - (uint32_t)((row > last ? row - last : 0) * (int)ldb) * 2u), dst + SP_TN_NARROW + j * 1024);
            }
        }
        l_slab += stride;
        return true;
    };
    int ahead = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) ahead += issue(i) ? 1 : 0;

    f32x16 acc[2][NB];
#pragma unroll
    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ia][jb][r] = 0.f;

    int slot = 0;
    for (int slab = w_gid; slab < n_slabs; slab += stride) {
        if (ahead >= 2)
            SP_WAIT_VM(NDN + NDW);
        else
            SP_WAIT_VM(0);
        unsigned char* sl = slots + slot * SLOT;
        if ((int64_t)slab * 32 + 32 > M) {  // ragged tail: zero the narrow rows past M (uniform branch, last slab only)
            const int live = (int)(M - (int64_t)slab * 32);
#pragma unroll
            for (int j = 0; j < NDN; ++j)
                if (((j * 64 + lane) >> 3) >= live) *reinterpret_cast<u32x4*>(sl + j * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        if (bmask) {  // keep-mask of D(X): every lane masks the chunks it loaded
#pragma unroll
            for (int j = 0; j < NDW; ++j) {
                u32x4* cp = reinterpret_cast<u32x4*>(sl + SP_TN_NARROW + j * 1024 + lane * 16);
                u32x4 v = *cp;
                const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)(slab * 32 + row_b(j)));
                VOps<T>::drop(v, drop, rh, (uint32_t)col_b(j));
                *cp = v;
            }
        }
        Frag<T> fa0 = frag_a(sl, 0), fa1;
        if (two) fa1 = frag_a(sl, 1);
        Frag<T> fb[NB];
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) fb[jb] = frag_b(sl, jb);
        SP_WAIT_LGKM0();
        --ahead;
        ahead += issue(slot) ? 1 : 0;
        slot = slot + 1 == NS ? 0 : slot + 1;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) mtl_mma(fa0, fb[jb], acc[0][jb]);
        if (two) {
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) mtl_mma(fa1, fb[jb], acc[1][jb]);
        }
    }

    // ---- workgroup sum (fixed order over the waves) and the partial in raw accumulator layout
    SP_WAIT_VM(0);
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [wave][reg][lane]
    float* dst = P->part + ((int64_t)pp * G + g) * (2 * NB * 1024);
    const int nblk = two ? 2 * NB : NB;
    for (int blk = 0; blk < nblk; ++blk) {
        const int ia = blk / NB, jb = blk - ia * NB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = 0.f;
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < NB; ++b2)
                    if (a2 == ia && b2 == jb) v = acc[a2][b2][r];
            red[(wave * 16 + r) * 64 + lane] = v;
        }
        __syncthreads();
#pragma unroll
        for (int e = tid; e < 1024; e += 64 * SP_WAVES) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < SP_WAVES; ++w) sum += red[w * 1024 + e];
            dst[blk * 1024 + e] = sum;
        }
        __syncthreads();
    }
}

constexpr int SP_TN_RG = 4;
template <int NB>
__global__ __launch_bounds__(256 * SP_TN_RG) void k_sp_tn_reduce(const SpTnParams P) {
    // one workgroup per (pp, accumulator block): thread t of a group owns 4 consecutive lanes of one accumulator register (16-byte
    // loads, 4 consecutive wide columns of one narrow row); the G partials are dealt round-robin to SP_TN_RG thread groups, each
    // summing its share in a fixed order with 4 independent chains, then a fixed-order LDS combine
    __shared__ f32x4 sm[SP_TN_RG][256];
    const int pp = blockIdx.y, blk = blockIdx.x;
    int pi = 0;
    for (int i = 1; i < P.n_prob; ++i)
        if (pp >= P.p[i].pp_lo) pi = i;
    const SpTnProb& pr = P.p[pi];
    const int local = pp - pr.pp_lo;
    const int ta = local / pr.parts, pc = local - ta * pr.parts;
    const int ia = blk / NB, jb = blk - ia * NB;
    const int na_valid = min(64, pr.Na - ta * 64);
    if (ia * 32 >= na_valid) return;
    const int t = threadIdx.x & 255, grp = threadIdx.x >> 8;
    const int G = P.G;
    const float* src = P.part + (int64_t)pp * G * (2 * NB * 1024) + blk * 1024 + t * 4;
    const int64_t stride = 2 * NB * 1024;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    int sp = grp;
    for (; sp + 3 * SP_TN_RG < G; sp += 4 * SP_TN_RG) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += *reinterpret_cast<const f32x4*>(src + (int64_t)(sp + u * SP_TN_RG) * stride);
    }
    for (; sp < G; sp += SP_TN_RG) acc[0] += *reinterpret_cast<const f32x4*>(src + (int64_t)sp * stride);
    sm[grp][t] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (grp != 0) return;
    f32x4 tsum = sm[0][t];
#pragma unroll
    for (int gI = 1; gI < SP_TN_RG; ++gI) tsum += sm[gI][t];
    const int e = t * 4, r = e >> 6, lane = e & 63;
    const int a = ta * 64 + ia * 32 + mtl_d_row(lane, r);
    const int b0 = pc * (NB * 32) + jb * 32 + (lane & 31);
    if (a >= pr.out_a) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (b0 + k < pr.out_b) {
            const int64_t at = pr.transpose ? (int64_t)(b0 + k) * pr.ldo + a : (int64_t)a * pr.ldo + b0 + k;
            pr.out[at] = tsum[k];
        }
}

// ------------------------------------------------------------------------------------------------
// k_sp_projk : the P / Q passes whose projection rows do NOT fit in LDS next to the slots (K R too large: the fc2 forward and the
// fc1 / qkv backward of stages 2 / 3, every layer of the r = 128 configurations) and whose row count is small.  The tiled kernel runs
// these on ceil(M / 128) workgroups (98 - 196 of 256 CUs) with one k-tile of prefetch: 35 - 60 us for 20 - 50 MB.  Here ONE work
// item (32-row slab, source) is a WORKGROUP and the reduction is split over its 8 waves: wave w takes the CH-wide chunks w, w + 8,
// ... of the slab (DMA into its private slots, all of them in flight at once: a whole slab row block of 32 x K is requested in one
// go), multiplies them with the matching columns of the projection rows -- read as MFMA fragments straight from global memory (the
// rows are L2-resident: 0.1 - 0.5 MB shared by every workgroup) -- and the 8 partial accumulator sets are summed through LDS in a
// fixed order (deterministic).  3 barriers per item; items are dealt round-robin to a persistent grid.
// ------------------------------------------------------------------------------------------------
template <typename T, int CH, int NSL>
__global__ __launch_bounds__(64 * SP_WAVES, SP_WAVES / 4) void k_sp_projk(const SpProjParams Pv) {
    typedef SpGeom<CH> G;
    (void)Pv;
    SpProjPtr P = (SpProjPtr)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, rl = lane & 31;
    const int K = P->K, NCH = K / CH;
    const int n_items = P->n_items, n_src = P->n_src;
    const int64_t M = P->M;
    unsigned char* slots = smem + (size_t)wave * NSL * G::SLOT;
    float* red = reinterpret_cast<float*>(smem);  // [wave][blk][reg][lane], aliases the slots (between barriers)
    DropoutCfg drop;
    drop.seed_lo = P->drop.seed_lo;
    drop.seed_hi = P->drop.seed_hi;
    drop.thr16 = P->drop.thr16;
    drop.off = P->drop.off;
    mtl_dropout_resolve(drop);
    const T* wp = reinterpret_cast<const T*>(P->wproj);
    const int Rw = P->Rw;

    int goff[G::NDMA], grow[G::NDMA], fo[G::KS];
#pragma unroll
    for (int j = 0; j < G::NDMA; ++j) {
        const int c = j * 64 + lane, row = c / G::CPR, p = c - row * G::CPR;
        grow[j] = row;
        goff[j] = row * K + G::logical(row, p) * 8;
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) fo[ks] = rl * G::ROWB + G::phys(rl, 2 * ks + h) * 16;
    const __amdgpu_buffer_rsrc_t orsrc = sp_rsrc(P->out, M * P->ld_out * 2);
    const int nmy = wave < NCH ? (NCH - wave + SP_WAVES - 1) / SP_WAVES : 0;  // chunks of this wave: wave, wave + 8, ...

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int slab = item / n_src, s = item - slab * n_src;
        const int blk_lo = P->src[s].blk_lo, n_blk = P->src[s].n_blk;
        const bool masked = P->src[s].mask != 0 && drop.thr16 != 0;
        const T* base = reinterpret_cast<const T*>(P->src[s].act) + (int64_t)slab * 32 * K;
        const int last = (int)(M - 1 - (int64_t)slab * 32);  // >= 31 for a full slab
        const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)(slab * 32 + rl));
        f32x16 acc[SP_MAXB];
#pragma unroll
        for (int b = 0; b < SP_MAXB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        for (int j0 = 0; j0 < nmy; j0 += NSL) {  // groups of NSL chunks: all of a group's loads are in flight together
            const int ng = nmy - j0 < NSL ? nmy - j0 : NSL;
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                if (u < ng) {
                    const T* cb = base + (wave + (j0 + u) * SP_WAVES) * CH;
#pragma unroll
                    for (int j = 0; j < G::NDMA; ++j) {
                        const int r = grow[j] < last ? grow[j] : last;  // rows past M re-read row M - 1 (never stored)
                        sp_dma16(cb + goff[j] + (r - grow[j]) * K, slots + u * G::SLOT + j * 1024);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                if (u < ng) {
                    const int ch = wave + (j0 + u) * SP_WAVES;
                    // projection fragments of this chunk, straight from global memory (issued before the wait for the slab chunk)
                    u32x4 wf[SP_MAXB][G::KS];
#pragma unroll
                    for (int b = 0; b < SP_MAXB; ++b) {
                        if (b < n_blk) {
                            const int row = (blk_lo + b) * 32 + rl;
                            const T* wr = wp + (int64_t)(row < Rw ? row : Rw - 1) * K + ch * CH + 8 * h;
#pragma unroll
                            for (int ks = 0; ks < G::KS; ++ks) {
                                wf[b][ks] = *reinterpret_cast<const u32x4*>(wr + ks * 16);
                                if (row >= Rw) wf[b][ks] = u32x4{0u, 0u, 0u, 0u};
                            }
                        }
                    }
                    SP_WAIT_VM(0);
                    const unsigned char* sl = slots + u * G::SLOT;
                    u32x4 xf[G::KS];
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) xf[ks] = *reinterpret_cast<const u32x4*>(sl + fo[ks]);
                    if (masked) {
#pragma unroll
                        for (int ks = 0; ks < G::KS; ++ks) VOps<T>::drop(xf[ks], drop, rh, (uint32_t)(ch * CH + ks * 16 + 8 * h));
                    }
#pragma unroll
                    for (int b = 0; b < SP_MAXB; ++b) {
                        if (b < n_blk) {
#pragma unroll
                            for (int ks = 0; ks < G::KS; ++ks) sp_mma1<T>(wf[b][ks], xf[ks], acc[b]);
                        }
                    }
                }
            }
        }
        // ---- sum of the 8 waves' partial accumulators (fixed order), conversion, store of the source's columns
        SP_WAIT_LGKM0();
        __syncthreads();  // every wave is done with its slots: the reduction image may alias them
#pragma unroll
        for (int b = 0; b < SP_MAXB; ++b) {
            if (b < n_blk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave * SP_MAXB + b) * 16 + r) * 64 + lane] = acc[b][r];
            }
        }
        __syncthreads();
        {
            // thread (m = tid & 31, cg = tid >> 5): the 8 output columns 8 cg .. 8 cg + 7 (relative to block blk_lo) of row m:
            // column c' = 8 q + 4 hh + e of a block sits in register 4 q + e of lane m + 32 hh
            const int m = tid & 31, cg = tid >> 5;
            const int b = cg >> 2, q = cg & 3;
            if (b < n_blk) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int hh = j >> 2, e = j & 3;
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < SP_WAVES; ++w) sum += red[((w * SP_MAXB + b) * 16 + 4 * q + e) * 64 + m + 32 * hh];
                    v[j] = sum;
                }
                const int col = (blk_lo + b) * 32 + 8 * q;
                const int64_t row = (int64_t)slab * 32 + m;
                const u32x4 o = u32x4{mtl_pk2<T>(v[0], v[1]), mtl_pk2<T>(v[2], v[3]), mtl_pk2<T>(v[4], v[5]), mtl_pk2<T>(v[6], v[7])};
                const bool ok = col >= P->src[s].col_lo && col < P->src[s].col_hi && row < M;
                sp_bstore(o, orsrc, ok ? (uint32_t)row * (uint32_t)(P->ld_out * 2) + (uint32_t)col * 2u : 0xFFFFFFFFu);
            }
        }
        __syncthreads();  // the image is read: the next item's loads may overwrite it
    }
}
