"""What register() itself spends outside the pair registrations: cProfile callees of registration.register (main thread)."""
import cProfile, pstats, sys, io
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from multiview_stitcher_amd import _lib, registration
from multiview_stitcher_amd import spatial_image_utils as si

dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, np.round(tile * 0.2).astype(int), seed=0, max_jitter=4)[:3]
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
key = si.DEFAULT_TRANSFORM_KEY
import gc; gc.collect(); gc.freeze()
f = lambda: registration.register(sims, transform_key=key, new_transform_key="reg", device=0)
f(); f()
pr = cProfile.Profile(); pr.enable(); f(); f(); f(); f(); pr.disable()
s = io.StringIO(); st = pstats.Stats(pr, stream=s); st.sort_stats("cumulative"); st.print_callees("registration.py.*\\(register\\)")
print(s.getvalue()[:6000])
s = io.StringIO(); st = pstats.Stats(pr, stream=s); st.sort_stats("cumulative"); st.print_callees("compute_pairwise_registrations")
print(s.getvalue()[:3000])
