// mvs_fuse_rows_f32.hip -- float instantiations of the row-owning fuse kernels (mvs_fuse_rows_dev.h); one translation
// unit per dtype so that the six kernels of each compile side by side.
#include "mvs_fuse_rows_dev.h"

void mvs_launch_rows_f32(int nvclass, bool frac, int nblocks, int wpg, hipStream_t s, const TrView* views, const mvsrows::Strip* strips,
                        const mvsrows::Cell* cells, const mvsrows::RowItem* items, int nitems, void* out, int oy, int ox, int tz, int ty, int tx) {
    mvsrows::launch_rows<float>(nvclass, frac, nblocks, wpg, s, views, strips, cells, items, nitems, out, oy, ox, tz, ty, tx);
}
