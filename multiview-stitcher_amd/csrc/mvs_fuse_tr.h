// mvs_fuse_tr.h -- internal: per-view record and blend-weight arithmetic of the translation fast path,
// shared by the column kernel (mvs_fuse.hip), the region kernel (mvs_fuse_region.hip) and the host-side
// region classification (same float formulas on both sides).
#pragma once
#include "mvs_internal.h"

#include <cmath>

constexpr float kPiHalf = 1.57079632679489661923f;

// Compact per-view record of the fast path (176 bytes; the region / column kernels read the first 160 = 10 x 16-byte
// loads per lane, the row kernels (mvs_fuse_rows*) read the fields with scalar loads).
struct alignas(16) TrView {
    int lo[3], hi[3];            // valid chunk-index box (inclusive), exact per scipy's in-bounds test
    int io[3];                   // input index = chunk index + io
    int wnz;                     // 5 (3D support) or 1 (2D)
    float fw[3];                 // fractional interpolation weights (0: single tap)
    int xtab_off;                // index of this view's x-weight table entry for chunk x = lo[2] (see xweight_table_kernel)
    unsigned long long data;     // device pointer of the slab
    long long span;              // elements from data[0] to the last voxel, + 1
    int stride_y, stride_z;      // elements
    int sup_ilo[3], sup_ihi[3];  // support nodes 0 / 4 in chunk-index units: ilo + flo, ihi - fhi
    float sup_flo[3], sup_fhi[3];
    float sup_k[3];              // support nodes per output pixel
    float ws[3];                 // tent scales of the closed-form support table
    int pad1[2];
    int n[3];                    // slab shape z, y, x (second-tap mirroring at the upper border, float tiles)
    int linear;                  // 1: interpolation order 1 (two taps per axis), 0: order 0 (one tap)
};
static_assert(sizeof(TrView) == 176, "TrView layout");

// ---- blend weight in "distance" form -----------------------------------------------------------
// Along one axis the support grid has nodes 0..4 at chunk indices sup_lo .. sup_hi.  With
// dl = x - sup_lo and dh = sup_hi - x (output pixels; computed as (float)(x - ilo) - flo so that the
// subtraction of the large parts is exact), the folded grid coordinate is u = min(dl, dh) * k in
// [0,2] (k = nodes per pixel; u < 0: outside the support, weight 0).  The table is symmetric, so
// nodes 3,4 fold onto 1,0.
__host__ __device__ __forceinline__ float fold_u(int x, int ilo, float flo, int ihi, float fhi, float k) {
    const float dl = (float)(x - ilo) - flo;
    const float dh = (float)(ihi - x) - fhi;
    return fminf(dl, dh) * k;
}
// Tent nodes bracketing folded coordinate u: a0 = s*i, a1 = s*(i+1), weight f = u - i, i in {0,1}.
__host__ __device__ __forceinline__ void tent_cell(float u, float s, float& a0, float& a1, float& f) {
    const float i = (u >= 1.f) ? 1.f : 0.f;
    a0 = s * i;
    a1 = s * (i + 1.f);
    f = u - i;
}
// W along x from the row nodes: lerp over {0, G1, G2} at folded coordinate u.
__host__ __device__ __forceinline__ float row_profile(float u, float G1, float dG) {
    return (u <= 1.f) ? u * G1 : fmaf(u - 1.f, dG, G1);
}
// branch-free cosine ramp (same arithmetic as blend_ramp)
__host__ __device__ __forceinline__ float blend_ramp_nb(float x) {
    const float xc = fminf(fmaxf(x, 0.f), 1.f);
    const float a = xc * kPiHalf;
    const float a2 = a * a;
    float s = fmaf(a2, 1.6059043836821613e-10f, -2.5052108385441720e-08f);
    s = fmaf(s, a2, 2.7557319223985893e-06f);
    s = fmaf(s, a2, -1.9841269841269841e-04f);
    s = fmaf(s, a2, 8.3333333333333333e-03f);
    s = fmaf(s, a2, -1.6666666666666666e-01f);
    s = fmaf(s * a2, a, a);
    const float c = fmaf(2.f, s * s, -1.f);
    const float w = (c + 1.f) * 0.5f;
    return (x >= 1.f) ? 1.f : w;
}


// Row nodes of view V at (z, y): the x profile of the row is lerp{0, G1, G2} over the folded x coordinate (row_profile).
// Returns false when the row lies outside the support along z or y (weight 0 on the whole row).
__host__ __device__ __forceinline__ bool tr_row_nodes(const TrView& V, int z, int y, float& G1, float& G2) {
    float az0 = INFINITY, az1 = INFINITY, fz = 0.f, uz = 0.f;
    const bool has_z = V.wnz > 1;
    if (has_z) {
        uz = fold_u(z, V.sup_ilo[0], V.sup_flo[0], V.sup_ihi[0], V.sup_fhi[0], V.sup_k[0]);
        if (uz < 0.f) return false;
        tent_cell(uz, V.ws[0], az0, az1, fz);
    }
    const float uy = fold_u(y, V.sup_ilo[1], V.sup_flo[1], V.sup_ihi[1], V.sup_fhi[1], V.sup_k[1]);
    if (uy < 0.f) return false;
    float ay0, ay1, fy;
    tent_cell(uy, V.ws[1], ay0, ay1, fy);
    const float uz_ = 1.f - fz, uy_ = 1.f - fy;
    const float m00 = fminf(az0, ay0), m01 = fminf(az0, ay1), m10 = fminf(az1, ay0), m11 = fminf(az1, ay1);
    const float a1 = V.ws[2], a2 = 2.f * V.ws[2];
    float g0 = fmaf(fminf(m01, a1), fy, fminf(m00, a1) * uy_);
    float g1 = fmaf(fminf(m11, a1), fy, fminf(m10, a1) * uy_);
    G1 = has_z ? fmaf(g1, fz, g0 * uz_) : g0;
    g0 = fmaf(fminf(m01, a2), fy, fminf(m00, a2) * uy_);
    g1 = fmaf(fminf(m11, a2), fy, fminf(m10, a2) * uy_);
    G2 = has_z ? fmaf(g1, fz, g0 * uz_) : g0;
    return true;
}

// Blend weight profile value W at chunk index (z, y, x) of view V (before the ramp): the closed form of the
// trilinear interpolation of edt = min_d(ws_d * tent(i_d)); < 0 is never returned, outside the support -> 0.
__host__ __device__ __forceinline__ float tr_weight_profile(const TrView& V, int z, int y, int x) {
    float G1, G2;
    if (!tr_row_nodes(V, z, y, G1, G2)) return 0.f;
    const float ux = fold_u(x, V.sup_ilo[2], V.sup_flo[2], V.sup_ihi[2], V.sup_fhi[2], V.sup_k[2]);
    if (ux < 0.f) return 0.f;
    return row_profile(ux, G1, G2 - G1);
}
