// panel.h -- k_pnl: persistent row-panel engine for the HBM-side MTLoRALinear launches (bf16).  Included by linear.hip
// (inside its anonymous namespace, after NtOut / EPI_ROW / gelu_* are defined).
//
// One launch = forward of a layer (all 1+T outputs), or its dX (all of dX, dX_t), with the low-rank projection formed
// by the SAME workgroup -- the P = alpha D(X) A^T / Q = alpha dY B passes of k_nt, their re-read of X / dY and the round
// trip of P / Q through HBM disappear (north star: "one LDS-staged kernel"; reference models/lora.py:253-284).
//
//   * a workgroup (8 waves, one per CU, persistent) owns 128-row panels  m0 = 128 * (blockIdx.x + i * gridDim.x);
//   * per panel it runs a PROGRAM of parts built by the host (PnPart): first the projection parts (one per activation
//     source: weight = alpha-scaled factor rows of that source's rank segment, zero rows elsewhere, so all sources
//     accumulate into one 128 x R accumulator), whose result is parked as a bf16 image  P[m][r]  in LDS (and copied to
//     HBM once, for the factor gradients); then, for every 128-column output tile, base parts (activation x weight tile,
//     one per gradient source in the backward: sum_o dY_o W costs MFMA time the HBM-bound launch has to spare, not a
//     k_sum pass) and rank parts (P image x expansion-factor tile) chained on one accumulator set (+ a second "base" set
//     when several outputs share the base GEMM), each output leaving through the transposing epilogue of k_nt;
//   * every part is consumed as 64-element k-steps from ONE ring of LDS stages filled by global_load_lds (no staging
//     registers; XOR-swizzled 128-byte rows as k_nt2): the loader runs NS-1 steps ahead of the MFMAs ACROSS parts, tiles
//     and panels, so a workgroup never pays a cold start after its first step and the loads of the next tile are in
//     flight while the current one is stored.  One bare s_barrier per step.
//   * vmcnt discipline: a wave waits for its own loads of step t with vmcnt(#loads of step t+1); epilogue stores are
//     issued BEFORE the loads of the next step in that iteration, so that the counter argument holds whatever order
//     stores and loads retire in.  Bias comes in through scalar loads (lgkmcnt) for the same reason.
constexpr int PN_BM = 128, PN_TN = 128, PN_BK = 64, PN_ROWB = 128;
constexpr int PN_STAGE = (PN_BM + PN_TN) * PN_ROWB;  // 32 KB per ring stage: weight rows 0..127, activation rows 128..255
constexpr int PN_EPI_WAVE = 32 * EPI_ROW;            // a wave's 32 (m) x 64 (n) bf16 output image
constexpr int PN_EPI = 8 * PN_EPI_WAVE;
constexpr int PN_PPAD = 48;                          // P image row = R * 2 + 48 bytes: 32 zero bytes (a 32-wide k sub-step may
                                                     // run 16 columns past R) + 16 of skew (row stride = 4 * odd dwords)
constexpr int PN_MAXPARTS = 2 * MAXO + 2;
constexpr int PN_LDS_MAX = 160 * 1024;

enum : int {
    PF_ZERO = 1,       // zero the accumulators before the part (PF_BIAS: initialise them with the bias instead)
    PF_BIAS = 2,
    PF_LOADBASE = 4,   // acc = base before the part
    PF_SAVEBASE = 8,   // base = acc after the part
    PF_MASK = 16,      // acc *= dropout keep(m, n) after the part (dX = G W + keep .* (Q A): rank part first)
    PF_EPI = 32,       // store acc as output `out` after the part
    PF_DROPACT = 64,   // the activation operand of the part is D(x): keep-mask applied to the fragments
    PF_RANK = 128      // activation operand = the LDS image, columns [k_lo, k_hi)
};

struct PnPart {
    const void* act;   // (M x ld_act) activation source; null for PF_RANK parts
    const void* wgt;   // weight-like operand: rows = output columns (or rank rows for projection parts)
    int64_t ld_act, ld_wgt;
    int w_lo, w_hi;    // valid weight rows (others read as zero)
    int k_lo, k_hi;    // reduction range
    int flags, out;
    int step0, pad_;   // first step of the part inside its group (projection parts / the parts of one tile); set by launch_pnl
};
struct PnOut {
    void* ptr;
    const void* gate;  // GATE: out *= gelu'(gate[m][n])
    void* act;         // ACT: second output gelu(out)
};
struct PnParams {
    int64_t M;
    int n_rows;        // output columns
    int n_proj, n_parts;
    int R;             // columns of the projected image (multiple of 16; 0: none)
    int nstage;        // ring depth (2 or 3)
    int dbg;           // developer toggles (MTLORA_PNL_DBG): 1 no output stores, 2 no loads, 4 no MFMA, 8 no vmcnt waits,
                       // 16 no epilogue at all, 32 no dropout mask, 64 no accumulator zero / bias
    int n_proj_steps, n_tile_steps;  // k-steps of the projection parts / of one tile's parts (launch_pnl)
    int64_t ld_out;
    const float* bias;
    void* pout;        // HBM copy of the image (M x R), nullable
    DropoutCfg drop;
    PnOut out[MAXO];
    PnPart proj[MAXO];
    PnPart part[PN_MAXPARTS];
};
typedef const __attribute__((address_space(4))) PnParams* PnPtr;
typedef const __attribute__((address_space(4))) PnPart* PnPartPtr;
typedef const __attribute__((address_space(4))) float* PnCF;
typedef __attribute__((ext_vector_type(8))) float f32x8;

// one k-step of the per-panel program, as the workgroup keeps it in LDS (built once at kernel start from the parts)
struct PnStep {
    uint64_t wgt;   // byte address of the weight window's row 0 at this step's k0
    uint64_t act;   // byte address of activation row 0 (of panel 0) at k0; 0: the activation operand is the LDS image
    uint32_t ld;    // row strides in bytes: weight | activation << 16
    uint32_t win;   // valid window rows [lo, hi): lo | hi << 8;  k0 << 16
    uint32_t ctl;   // flags (12 bits) | 16-byte chunks with data (1..8) << 12 | output index << 16 | n-tile << 20
    uint32_t pad_;
};
static_assert(sizeof(PnStep) == 32, "PnStep is two 16-byte LDS reads");
constexpr int PN_MAXSTEPS = 256;
constexpr int PN_TABLE = PN_MAXSTEPS * 32;
enum : int { PF_PARK = 256, PF_PROJ = 512 };  // kernel-internal: last projection step / projection step

template <int V>
struct PnIC {
    static constexpr int value = V;
};

template <bool MULTI, bool MLR, bool GATE, bool ACT, int NS>
__global__ __launch_bounds__(512, 2) void k_pnl(const PnParams Pv) {
    (void)Pv;
    PnPtr P = (PnPtr)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3, h = lane >> 5, rl32 = lane & 31;
    const int R = P->R, PRS = R * 2 + PN_PPAD, dbg = P->dbg;
    const int n_rows = P->n_rows, n_proj = P->n_proj, n_parts = P->n_parts;
    const int n_tiles = (n_rows + PN_TN - 1) / PN_TN;
    const int64_t M = P->M;
    const int n_panels = (int)((M + PN_BM - 1) / PN_BM);
    const int n_steps = P->n_proj_steps + n_tiles * P->n_tile_steps;  // per panel
    unsigned char* epi = smem + NS * PN_STAGE + wave * PN_EPI_WAVE;
    unsigned char* sP = smem + NS * PN_STAGE + PN_EPI;
    const PnStep* table = reinterpret_cast<const PnStep*>(smem + NS * PN_STAGE + PN_EPI + (R > 0 ? PN_BM * PRS : 0));
    if ((int)blockIdx.x >= n_panels) return;

    DropoutCfg drop;
    drop.seed_lo = P->drop.seed_lo;
    drop.seed_hi = P->drop.seed_hi;
    drop.thr16 = P->drop.thr16;
    drop.off = P->drop.off;
    mtl_dropout_resolve(drop);

    if (R > 0)  // zero pad columns of the image: written once, never touched again
        for (int i = tid; i < PN_BM * 3; i += 512)
            *reinterpret_cast<u32x4*>(sP + (i / 3) * PRS + R * 2 + (i % 3) * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- build the step table: thread t expands sequence part t (projection parts, then every tile's parts)
    {
        const int n_seq = n_proj + n_tiles * n_parts;
        for (int t = tid; t < n_seq; t += 512) {
            const bool is_proj = t < n_proj;
            const int bn = is_proj ? 0 : (t - n_proj) / n_parts;
            const int j = is_proj ? t : (t - n_proj) - bn * n_parts;
            PnPartPtr pt = is_proj ? (PnPartPtr)&P->proj[j] : (PnPartPtr)&P->part[j];
            const int k_lo = pt->k_lo, k_hi = pt->k_hi, flags = pt->flags;
            const int rbase = is_proj ? 0 : bn * PN_TN;
            int lo = pt->w_lo - rbase, hi = pt->w_hi - rbase;
            lo = lo < 0 ? 0 : lo;
            hi = hi > PN_TN ? PN_TN : (hi < 0 ? 0 : hi);
            const uint64_t wbase = (uint64_t)(uintptr_t)pt->wgt + (uint64_t)rbase * (uint64_t)pt->ld_wgt * 2u;
            const uint64_t abase = (uint64_t)(uintptr_t)pt->act;
            int sidx = is_proj ? pt->step0 : P->n_proj_steps + bn * P->n_tile_steps + pt->step0;
            PnStep* out = const_cast<PnStep*>(table) + sidx;
            for (int k0 = k_lo; k0 < k_hi; k0 += PN_BK, ++out) {
                const bool first = k0 == k_lo, last = k0 + PN_BK >= k_hi;
                int f = flags & (PF_RANK | PF_DROPACT);
                if (first) f |= flags & (PF_ZERO | PF_BIAS | PF_LOADBASE);
                if (last) f |= flags & (PF_MASK | PF_SAVEBASE | PF_EPI);
                if (is_proj) f |= PF_PROJ | ((last && j == n_proj - 1) ? PF_PARK : 0) | ((first && j == 0) ? PF_ZERO : 0);
                const int left = k_hi - k0, nch = left >= PN_BK ? 8 : left >> 3;
                PnStep e;
                e.wgt = wbase + (uint64_t)k0 * 2u;
                e.act = abase ? abase + (uint64_t)k0 * 2u : 0;
                e.ld = (uint32_t)(pt->ld_wgt * 2) | ((uint32_t)(pt->ld_act * 2) << 16);
                e.win = (uint32_t)lo | ((uint32_t)hi << 8) | ((uint32_t)k0 << 16);
                e.ctl = (uint32_t)f | ((uint32_t)nch << 12) | ((uint32_t)pt->out << 16) | ((uint32_t)bn << 20);
                e.pad_ = 0;
                *out = e;
            }
        }
    }
    __syncthreads();

    // ---------------- loader ----------------
    // wave w < 4 fills weight rows [32 w, 32 w + 32) of the stage, wave w >= 4 the activation rows [32 (w - 4), ...): four
    // wave instructions of 8 rows each; lane -> (row = lane >> 3, physical 16-byte chunk = lane & 7), logical chunk =
    // physical ^ ((row >> 1) & 7) (the ds_read_b128 fragment reads then touch every bank once)
    const bool ld_w = wave < 4;
    int lrow[4], lkc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lrow[j] = (wave & 3) * 32 + j * 8 + (lane >> 3);
        lkc[j] = (lane & 7) ^ ((lrow[j] >> 1) & 7);
    }
    int ls = 0, lp = blockIdx.x;  // loader position: step within the panel, panel
    const uint64_t zpage = (uint64_t)(uintptr_t)g_zero16;
    // per-lane row offsets / row validity are cached and only recomputed when the row stride / the window (weight side) or
    // the panel (activation side) changes: the steps of one part differ in k0 alone
    uint32_t c_ld = 0xFFFFFFFFu, c_key = 0xFFFFFFFFu, rowoff[4] = {0u, 0u, 0u, 0u};
    uint32_t okrow = 0u;  // bit j: row j of this lane is inside the window / the matrix (a per-lane VGPR, not four lane masks)
    // issue the loads of the loader's step into ring slot SLOT, advance; returns this wave's load count
    auto issue_next = [&](auto slot_tag) __attribute__((always_inline)) -> int {
        constexpr int SLOT = decltype(slot_tag)::value;
        if (lp >= n_panels) return 0;
        const u32x4 e0 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(table) + ls * 32);
        const u32x4 e1 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(table) + ls * 32 + 16);
        unsigned char* stage = smem + SLOT * PN_STAGE;
        const uint32_t ctl = (uint32_t)__builtin_amdgcn_readfirstlane((int)e1[2]);
        const int nch = (ctl >> 12) & 15;
        int n = 0;
        if (dbg & 2) {
        } else if (ld_w) {
            const uint64_t wb = ((uint64_t)e0[1] << 32) | e0[0];
            const uint32_t ldw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(e1[0] & 0xFFFFu));
            const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)(e1[1] & 0xFFFFu));
            if (ldw != c_ld) {
                c_ld = ldw;
#pragma unroll
                for (int j = 0; j < 4; ++j) rowoff[j] = (uint32_t)lrow[j] * ldw + (uint32_t)lkc[j] * 16u;
            }
            if (key != c_key) {
                c_key = key;
                const uint32_t lo = key & 0xFF, span = ((key >> 8) & 0xFF) - lo;
                okrow = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) okrow |= ((uint32_t)lrow[j] - lo < span ? 1u : 0u) << j;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = ((okrow >> j) & 1u) && lkc[j] < nch;
                const uint64_t a = ok ? wb + rowoff[j] : zpage;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)a,
                                                 (__attribute__((address_space(3))) void*)(stage + (wave * 4 + j) * 1024), 16, 0, 0);
            }
            n = 4;
        } else if (!(ctl & PF_RANK)) {
            const uint32_t lda = (uint32_t)__builtin_amdgcn_readfirstlane((int)(e1[0] >> 16));
            const uint64_t ab = (((uint64_t)e0[3] << 32) | e0[2]) + (uint64_t)lp * (uint64_t)(PN_BM * lda);
            if (lda != c_ld) {
                c_ld = lda;
#pragma unroll
                for (int j = 0; j < 4; ++j) rowoff[j] = (uint32_t)lrow[j] * lda + (uint32_t)lkc[j] * 16u;
            }
            if ((uint32_t)lp != c_key) {
                c_key = (uint32_t)lp;
                okrow = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) okrow |= ((int64_t)lp * PN_BM + lrow[j] < M ? 1u : 0u) << j;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = ((okrow >> j) & 1u) && lkc[j] < nch;
                const uint64_t a = ok ? ab + rowoff[j] : zpage;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)a,
                                                 (__attribute__((address_space(3))) void*)(stage + (wave * 4 + j) * 1024), 16, 0, 0);
            }
            n = 4;
        }
        if (++ls == n_steps) {
            ls = 0;
            lp += gridDim.x;
        }
        return n;
    };

    // ---------------- consumer ----------------
    f32x16 acc[2], base[2];
    (void)base;
    int n1 = 0;  // this wave's load count of the youngest issued step (NS == 3: the one allowed to stay in flight)
    int64_t m0 = 0;
    int n0 = 0;
    uint32_t rh = 0;  // dropout row hash of this lane's activation row
    bool p_copy = false;

    // fragment addressing inside a stage: byte offsets of this lane's eight weight-side and four activation-side 16-byte
    // pieces per step (u = 32-element half, hh = slot), XOR swizzle applied once
    const int ra = wm * 32 + rl32;
    int wofs[2][2][2], aofs[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int rw = wn * 64 + s2 * 32 + rl32;
                wofs[u][s2][hh] = rw * PN_ROWB + (((4 * u + 2 * hh + h) ^ ((rw >> 1) & 7)) << 4);
            }
            aofs[u][hh] = PN_TN * PN_ROWB + ra * PN_ROWB + (((4 * u + 2 * hh + h) ^ ((ra >> 1) & 7)) << 4);
        }

    auto apply_mask = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + sn * 32 + 8 * q + 4 * h;
                const uint32_t h0 = mtl_dropout_pairbits(drop, rh, (uint32_t)n);
                const uint32_t h1 = mtl_dropout_pairbits(drop, rh, (uint32_t)(n + 2));
                if ((h0 & 0xFFFFu) < drop.thr16) acc[sn][q * 4 + 0] = 0.f;
                if ((h0 >> 16) < drop.thr16) acc[sn][q * 4 + 1] = 0.f;
                if ((h1 & 0xFFFFu) < drop.thr16) acc[sn][q * 4 + 2] = 0.f;
                if ((h1 >> 16) < drop.thr16) acc[sn][q * 4 + 3] = 0.f;
            }
    };
    // acc -> output o: transpose the wave's 64 (n) x 32 (m) tile through its private LDS image, 128-byte row segments out
    auto epilogue = [&](int o) __attribute__((always_inline)) {
        bf16* outp = reinterpret_cast<bf16*>(P->out[o].ptr);
        const bf16* gate = reinterpret_cast<const bf16*>(P->out[o].gate);
        bf16* actp = reinterpret_cast<bf16*>(P->out[o].act);
        (void)gate;
        (void)actp;
        if (!outp || n0 + wn * 64 >= n_rows || (dbg & 16)) return;
#pragma unroll
        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = sn * 32 + 8 * q + 4 * h;
                u32x2 pk = {mtl_pack_bf16(acc[sn][q * 4], acc[sn][q * 4 + 1]), mtl_pack_bf16(acc[sn][q * 4 + 2], acc[sn][q * 4 + 3])};
                *reinterpret_cast<u32x2*>(epi + rl32 * EPI_ROW + nl * 2) = pk;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the image is private to this wave
        __builtin_amdgcn_wave_barrier();
        const int64_t ldo = P->ld_out;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int ml = it * 8 + (lane >> 3), c16 = lane & 7;
            const int64_t m = m0 + wm * 32 + ml;
            const int n = n0 + wn * 64 + c16 * 8;
            u32x4 v = *reinterpret_cast<const u32x4*>(epi + ml * EPI_ROW + c16 * 16);
            if (m < M && n < n_rows) {
                if constexpr (GATE) {
                    if (gate) {
                        const u32x4 hv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(gate + m * ldo + n));
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float g0 = __builtin_bit_cast(float, v[q] << 16) * gelu_grad(__builtin_bit_cast(float, hv[q] << 16));
                            const float g1 = __builtin_bit_cast(float, v[q] & 0xFFFF0000u) *
                                             gelu_grad(__builtin_bit_cast(float, hv[q] & 0xFFFF0000u));
                            v[q] = mtl_pack_bf16(g0, g1);
                        }
                    }
                }
                if (!(dbg & 1)) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(outp + m * ldo + n));
                if constexpr (ACT) {
                    if (actp) {
                        u32x4 av;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            av[q] = mtl_pack_bf16(gelu_fwd(__builtin_bit_cast(float, v[q] << 16)),
                                                  gelu_fwd(__builtin_bit_cast(float, v[q] & 0xFFFF0000u)));
                        if (!(dbg & 1)) __builtin_nontemporal_store(av, reinterpret_cast<u32x4*>(actp + m * ldo + n));
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // prologue: NS - 1 steps in flight
    {
        const int first = issue_next(PnIC<0>{});
        n1 = first;
        if constexpr (NS == 3) n1 = issue_next(PnIC<1>{});
    }

    int cs = 0, cp = blockIdx.x;
    // one k-step out of ring slot SLOT (a compile-time constant: the main loop below is unrolled over the ring); returns
    // false after the workgroup's last step
    auto do_step = [&](auto slot_tag) __attribute__((always_inline)) -> bool {
        constexpr int SLOT = decltype(slot_tag)::value;
        typedef PnIC<(SLOT + NS - 1) % NS> LdSlot;  // the slot the loader fills while this one is multiplied
        if (cs == 0) {
            m0 = (int64_t)cp * PN_BM;
            if (drop.enabled()) rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)(m0 + ra));
        }
        const u32x4 e1 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(table) + cs * 32 + 16);
        const int ctl = __builtin_amdgcn_readfirstlane((int)e1[2]);
        const int win = __builtin_amdgcn_readfirstlane((int)e1[1]);
        const int k0 = (int)((uint32_t)win >> 16), w_hi = (win >> 8) & 0xFF, nch = (ctl >> 12) & 15;
        const bool active = wn * 64 < w_hi;
        // ---- before the step
        if (ctl & (PF_ZERO | PF_LOADBASE)) {
            n0 = ((ctl >> 20) & 0xFFF) * PN_TN;
            if ((ctl & PF_ZERO) && !(dbg & 64)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
                if ((ctl & PF_BIAS) && active) {
                    // wave-uniform addresses: scalar loads (lgkmcnt) -- a vector load here would sit behind the ring's
                    // in-flight LDS-DMA loads in vmcnt order and drain them.  All sixteen are issued before the first use
                    // (addresses clamped into the row instead of branching around each load); columns >= n_rows are never
                    // stored.
                    PnCF bs = (PnCF)(uintptr_t)P->bias;
                    f32x4 bv[2][4][2];
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            int nb = n0 + wn * 64 + sn * 32 + 8 * q;
                            nb = __builtin_amdgcn_readfirstlane(nb > n_rows - 8 ? n_rows - 8 : nb);
                            bv[sn][q][0] = *reinterpret_cast<const __attribute__((address_space(4))) f32x4*>(bs + nb);
                            bv[sn][q][1] = *reinterpret_cast<const __attribute__((address_space(4))) f32x4*>(bs + nb + 4);
                        }
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[sn][q * 4 + e] = h ? bv[sn][q][1][e] : bv[sn][q][0][e];
                }
            }
            if constexpr (MULTI) {
                if (ctl & PF_LOADBASE) {
                    acc[0] = base[0];
                    acc[1] = base[1];
                }
            }
        }
        // ---- the step: wait for its tile, barrier, keep the loader NS - 1 steps ahead, multiply
        if (dbg & 8) {
        } else if (NS == 3 && n1 == 4)
            __builtin_amdgcn_s_waitcnt(0x0F74);  // vmcnt(4): the younger issued step may stay in flight
        else
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        asm volatile("s_barrier" ::: "memory");
        if (p_copy) {  // copy the freshly parked image to HBM (128-byte+ row segments); older than the loads issued below
            p_copy = false;
            bf16* pg = reinterpret_cast<bf16*>(P->pout);
            const int cpr = R >> 3;  // 16-byte chunks per row
            for (int c = tid; c < PN_BM * cpr; c += 512) {
                const int row = c / cpr, ch = c - row * cpr;
                const int64_t m = m0 + row;
                if (m < M)
                    *reinterpret_cast<u32x4*>(pg + m * R + ch * 8) = *reinterpret_cast<const u32x4*>(sP + row * PRS + ch * 16);
            }
        }
        const bool epi_iter = (ctl & PF_EPI) != 0;
        if (!epi_iter) n1 = issue_next(LdSlot{});
        if (active && !(dbg & 4)) {
            const unsigned char* stage = smem + SLOT * PN_STAGE;
            const bool from_p = (ctl & PF_RANK) != 0;
            const bool dropact = (ctl & PF_DROPACT) && drop.enabled() && !(dbg & 32);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u * 4 >= nch) break;
                Frag<bf16> fw[2], fa;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    fw[s2].v[0] = *reinterpret_cast<const u32x4*>(stage + wofs[u][s2][0]);
                    fw[s2].v[1] = *reinterpret_cast<const u32x4*>(stage + wofs[u][s2][1]);
                }
                if (from_p) {
                    const unsigned char* pr = sP + ra * PRS + (k0 + 32 * u) * 2;
                    fa.v[0] = *reinterpret_cast<const u32x4*>(pr + h * 16);
                    fa.v[1] = *reinterpret_cast<const u32x4*>(pr + (2 + h) * 16);
                } else {
                    fa.v[0] = *reinterpret_cast<const u32x4*>(stage + aofs[u][0]);
                    fa.v[1] = *reinterpret_cast<const u32x4*>(stage + aofs[u][1]);
                    if (dropact) {
                        VOps<bf16>::drop(fa.v[0], drop, rh, (uint32_t)(k0 + (4 * u + h) * 8));
                        VOps<bf16>::drop(fa.v[1], drop, rh, (uint32_t)(k0 + (4 * u + 2 + h) * 8));
                    }
                }
                mtl_mma(fw[0], fa, acc[0]);
                mtl_mma(fw[1], fa, acc[1]);
            }
        }
        // ---- after the step
        if (ctl & (PF_MASK | PF_SAVEBASE | PF_EPI | PF_PARK)) {
            if constexpr (MLR) {
                if ((ctl & PF_MASK) && drop.enabled()) apply_mask();
            }
            if constexpr (MULTI) {
                if (ctl & PF_SAVEBASE) {
                    base[0] = acc[0];
                    base[1] = acc[1];
                }
            }
            if (epi_iter) {
                epilogue((ctl >> 16) & 15);
                n1 = issue_next(LdSlot{});  // after the stores: see the vmcnt note at the top
            }
            if (ctl & PF_PARK) {  // acc holds image^T [rank r][row m] (alpha is folded into the projection weights)
                if (wn * 64 < R) {
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int r = wn * 64 + sn * 32 + 8 * q + 4 * h;
                            if (r < R) {
                                const u32x2 pk = {mtl_pack_bf16(acc[sn][q * 4], acc[sn][q * 4 + 1]),
                                                  mtl_pack_bf16(acc[sn][q * 4 + 2], acc[sn][q * 4 + 3])};
                                *reinterpret_cast<u32x2*>(sP + ra * PRS + r * 2) = pk;
                            }
                        }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): visible to the other waves after the next step's barrier
                p_copy = P->pout != nullptr;
            }
        }
        if (++cs == n_steps) {
            cs = 0;
            cp += gridDim.x;
        }
        return cp < n_panels;
    };
    for (;;) {
        if (!do_step(PnIC<0>{})) break;
        if (!do_step(PnIC<1>{})) break;
        if constexpr (NS == 3) {
            if (!do_step(PnIC<2>{})) break;
        }
    }
}
