#include "mvs_fft_lines.inc"
bool mvs_launch_dft_line_hi(MvsContext* c, const FftArgs& A, unsigned grid) { return launch_dft_line<64, 45>(c, A, grid); }
