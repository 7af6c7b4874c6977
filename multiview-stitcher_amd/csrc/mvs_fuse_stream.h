// mvs_fuse_stream.h -- internal: the streaming row kernel of the translation fast path (mvs_fuse_stream.hip), driven by the
// region planner (mvs_fuse_region.hip).
#pragma once
#include "mvs_fuse_tr.h"

#include <vector>

// rows [z0, z1) x [y0, y1) of the chunk over its whole x range; `views` (ascending): the views that reach into them, each
// covering the rows completely in z and y
struct StreamStrip { int z0, z1, y0, y1; std::vector<int> views; };

// addressing limits of the kernel: integer offsets, slab < 1 GiB
bool mvs_stream_view_ok(const TrView& V);

// Builds (strips != nullptr) or reuses (strips == nullptr: the plan built last for `hash`) the tables of the strips and launches
// the kernel on `stream` (nullptr: tables only).  Uploads go through c->stream, so `stream` must be ordered after it.
int mvs_fuse_stream(MvsContext* c, const TrView* htr, const TrView* dtr, int n_views, const std::vector<StreamStrip>* strips,
                    unsigned long long hash, void* dout, const int o[3], const int t[3], hipStream_t stream);
