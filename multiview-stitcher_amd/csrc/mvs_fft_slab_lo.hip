#include "mvs_fft_slab.inc"
bool mvs_launch_slab_lo(MvsContext* c, const SlabArgs& A, unsigned grid, size_t lds_bytes, hipError_t* err) { return launch_slab<44, 17>(c, A, grid, lds_bytes, err); }
