#include "mvs_fft_lines.inc"
bool mvs_launch_dft_line_lo(MvsContext* c, const FftArgs& A, unsigned grid) { return launch_dft_line<44, 17>(c, A, grid); }
