// hid_params.h -- parameter blocks and launch geometry of the k_hid_* kernels (hid.hip), shared with their host side in linear.hip
#pragma once
#include <stdint.h>
#include <stddef.h>

constexpr int HID_TG = 4;   // tasks per launch (rank <= 4); launches with rank <= 8 take HID_TG / 2
constexpr int HID_RB = 4;   // rows between two workgroup barriers
constexpr int HID_MAXW = 16;

struct HidParams {
    const void* hbase;    // (M x H)
    const void* p1;       // (M x ldp1)  P of fc1; task i of this launch owns columns [off1[i], off1[i] + 8)
    void* p2;             // forward out (M x ldp2): columns [off2[i], off2[i] + 8) (alpha-scaled; columns past the rank are written as zeros)
    const void* q2;       // backward in (M x ldq2): Q of fc2, columns off2[i]
    void* q1;             // backward out (M x ldq1): Q of fc1, columns off1[i]
    const void* gsrc;     // backward in (M x H): dH_s for the first task group, the running G for the following ones (may be null: zeros)
    void* g;              // backward out (M x H)
    const void* b1t;      // (R1 x H) bt_cat of fc1: row rr = B1[:, rr], unscaled
    const void* a2;       // (R2 x H) a_cat of fc2: row rr = A2[rr, :], unscaled
    const float* alpha1;  // (R1)
    const float* alpha2;  // (R2)
    float* part;          // backward: [workgroup][nt][2][RR][chunk columns]  (kind 0: dB1^T[rho][j], kind 1: dA2[rho][j])
    float* rowpart;       // MFMA forms: [chunk][M][16] unscaled fp32 row sums (P2 forward, Q1 backward), finished by k_hid_rows_finish
    int64_t M;
    int H, nt;
    int ldp1, ldp2, ldq1, ldq2;
    int off1[HID_TG], off2[HID_TG];
};


struct HidOff {
    int v[HID_TG];
};
struct HidRedParams {
    const float* part;  // [chunk][n_wg][nt][2][RR][chunk_cols]
    int n_wg, nt, RR, H;
    int chunk_cols;     // columns per chunk (H: one chunk)
    int r[HID_TG];         // un-padded ranks
    float* dB1[HID_TG];    // (H x r) row-major, nullable
    float* dA2[HID_TG];    // (r x H) row-major, nullable
};

// MFMA forms: LDS bytes of a workgroup serving a chunk of hc columns (hc / 32 waves)
static inline size_t hid_d_lds_bytes(int hc, bool bwd) {
    // (last term: the per-wave images, overlaid by the [waves][16][33] fp32 row-sum table)
    return (size_t)(bwd ? 2 : 1) * HID_TG * hc * 8 + (size_t)17 * (hc * 2 + 16) + (bwd ? 4096 : 2048) + (size_t)(hc / 32) * (16 * 33 * 4);
}
// chunk width (columns per workgroup, 32 per wave).  Forward: one 12-wave workgroup per CU over 384 columns (128-column chunks: 0.33 ->
// 0.35 ms at stage 0 of c2: three times the P1 / row-sum traffic buys nothing, the kernel is VALU-bound at 128 registers either way).
// Backward: three independent 4-wave workgroups per CU over 128 columns each (0.65 -> 0.56 ms at stage 0, 0.34 -> 0.30 at stage 1:
// the barriers of a row block stall 4 waves instead of 12; the 256-column form needs 180 registers = 2 waves per SIMD).
// (192-column chunks were measured once with two waves per SIMD by mistake: 1.00 vs 0.76 ms.)
static inline int hid_d_chunk(int H, bool bwd) {
    if (bwd) return H % 128 == 0 ? 128 : 0;
    return H % 384 == 0 ? 384 : (H % 256 == 0 ? 256 : (H % 128 == 0 ? 128 : 0));
}

// launch descriptor filled by linear.hip (which owns the layer layouts), executed by hid.hip
struct HidLaunch {
    int kind;        // 0 k_hid_proj (VALU), 1 k_hid_bwd (VALU), 2 k_hid_fwd_d, 3 k_hid_bwd_d (MFMA, accumulator layout)
    int hc, n_chunk; // MFMA forms: chunk width (384 / 256 / 128 columns) and chunks (grid y)
    int dtype;       // MTLORA_BF16 / MTLORA_F16
    int tg, rr;      // VALU forms: tasks per launch / rank block (4 or 8); MFMA forms: rr
    int nthr, n_wg;  // threads per workgroup, workgroups
    size_t lds;      // dynamic LDS bytes (MFMA forms)
};
