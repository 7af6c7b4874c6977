"""C3 at size (tests/test_at_size_parity_gpu.py::test_c3_register_and_content_based_sampled_oracle_parity) under the content-based
options given as KEY=VALUE arguments (cb_exact, cb_taps_f64, ...): prints the statistics check_boxes returns."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_at_size_parity_gpu as T
from multiview_stitcher_amd import _lib
_lib.init(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.set_option(k, int(v))
orig = T.at_size.check_boxes
def spy(*a, **kw):
    st = orig(*a, **kw)
    print("options", sys.argv[1:], "->", {k: st[k] for k in ("voxels", "lsb_flips", "beyond_plain_bar", "marginal_voxels", "max_floor_used") if k in st}, flush=True)
    return st
T.at_size.check_boxes = spy
try:
    T.test_c3_register_and_content_based_sampled_oracle_parity(0)
except AssertionError as e:
    print("assertion:", str(e)[:200])
