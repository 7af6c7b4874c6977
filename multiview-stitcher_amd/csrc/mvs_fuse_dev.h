// mvs_fuse_dev.h -- internal: device-side view record of the generic fuse path and launch helpers
// shared between mvs_fuse.hip and mvs_gauss.hip.
#pragma once
#include "mvs_internal.h"

struct DevView {
    const void* data;
    long long stride_z, stride_y;   // elements
    int nz, ny, nx;                 // slab shape
    int wnz;                        // z extent of the support table: 5 (3D) or 1 (2D)
    double m[9];
    double off[3];
    double wm[9];
    double woff[3];
    float edt[125];
    // ---- translation fast path (valid when tr_ok): matrix == I, diagonal support map ----
    int tr_ok;
    int io[3];        // input index = chunk index + io
    float fw[3];      // fractional interpolation weights (0 => single tap on that axis)
    int lo[3], hi[3]; // chunk-index range where the view is in bounds, exact per scipy's test
    float ws[3];      // tent scales of the closed-form support table edt = min_d(ws_d * tent(i_d))
    float sup_k[3];   // support nodes per output pixel (= w_matrix diagonal)
    // chunk-index coordinates of support nodes 0 and 4, split as ilo + flo and ihi - fhi with
    // integer ilo/ihi and fractions in [0,1): distances to them are exact in float
    int sup_ilo[3], sup_ihi[3];
    float sup_flo[3], sup_fhi[3];
    long long span;   // elements from data[0] to the last voxel of the slab, + 1
    float pad[3];     // pad[0]: 1 = linear interpolation (set with tr_ok)
};


int mvs_fill_dev_view(MvsContext* c, const mvs_view_t& v, int ndim, const void* dev_data, DevView* d);
// `box0` (may be null): chunk index of the output array's first voxel -- the output is a sub-box of the chunk, evaluated with the
// chunk's own coordinates
void mvs_launch_resample(MvsContext* c, const DevView& d, int dtype, int order, float cval, float* out, const int64_t shape[3], const int* box0 = nullptr,
                         MvsCropStats* crop_stats = nullptr, int crop_stats_k = 0);
void mvs_launch_blend(MvsContext* c, const DevView& d, float* out, const int64_t shape[3], const int* box0 = nullptr);
// both for the boxes of up to 8 views in two launches (`dviews`: the same records in device memory)
// (`tr`: the views' translation-path records, or NULL -- with them the blend weights come from the closed form in float arithmetic)
struct TrView;
void mvs_launch_boxes_batch(MvsContext* c, const DevView* hviews, const DevView* dviews, int n_views, int dtype, int order, float cval,
                            float* const* res_out, float* const* blend_out, const int64_t (*shapes)[3], const int (*box0)[3],
                            const TrView* tr = nullptr);
bool mvs_prepare_tr_view(DevView* d, int order, const int64_t chunk_shape[3], size_t elem_size, const int64_t org[3], const int64_t ioff[3],
                         TrView* out);
void mvs_view_chunk_box(const DevView& d, const int64_t shape[3], int lo[3], int hi[3]);
void mvs_view_to_chunk_frame(DevView* d, const int64_t org[3], const int64_t ioff[3]);
