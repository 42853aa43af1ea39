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
    float* part;          // backward: [gridDim.x][nt][2][RR][H]  (kind 0: dB1^T[rho][j], kind 1: dA2[rho][j])
    int64_t M;
    int H, nt;
    int ldp1, ldp2, ldq1, ldq2;
    int off1[HID_TG], off2[HID_TG];
};


struct HidRedParams {
    const float* part;  // [n_wg][nt][2][RR][H]
    int n_wg, nt, RR, H;
    int r[HID_TG];         // un-padded ranks
    float* dB1[HID_TG];    // (H x r) row-major, nullable
    float* dA2[HID_TG];    // (r x H) row-major, nullable
};

constexpr int HIDM_CW = 128;  // columns per wave
constexpr int HIDB_CW = 32;

template <typename T, int RR>
struct HidMGeom {
    static constexpr int NV = HID_TG * RR;                  // accumulator columns in use (<= 32)
    static __host__ __device__ int a2_stride(int H) { return H * 2 + 16; }
    static __host__ __device__ size_t lds_bytes(int H) {
        const int NW = H / HIDM_CW;
        return (size_t)NV * H * 4 + (size_t)(NV + 1) * a2_stride(H) + (size_t)2 * NW * NV * 32 * 4;
    }
};

template <typename T, int RR>
struct HidBGeom {
    static constexpr int NV = HID_TG * RR;
    static __host__ __device__ int b_stride(int H) { return H * 2 + 16; }
    static __host__ __device__ size_t lds_bytes(int H) {
        const int NW = H / HIDB_CW;
        const size_t tiles = (size_t)NW * 2 * 32 * 64, red = (size_t)NW * NV * 32 * 4;
        return (size_t)2 * NV * H * 4 + (size_t)(NV + 1) * b_stride(H) + 2 * 32 * 64 + (tiles > red ? tiles : red);
    }
};

// launch descriptor filled by linear.hip (which owns the layer layouts), executed by hid.hip
struct HidLaunch {
    int kind;        // 0 k_hid_proj (VALU), 1 k_hid_bwd (VALU), 2 k_hid_proj_m, 3 k_hid_bwd_m, 4 k_hid_reduce
    int dtype;       // MTLORA_BF16 / MTLORA_F16
    int tg, rr;      // VALU forms: tasks per launch / rank block (4 or 8); MFMA forms: rr
    int nthr, n_wg;  // threads per workgroup, workgroups
    size_t lds;      // dynamic LDS bytes (MFMA forms)
};
