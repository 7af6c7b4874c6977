"""The pruned arg-max search against the full scoring on north-star mosaics of several seeds and noise levels: every pairwise
translation and quality must be identical.  python tools/prune_stress.py [seeds] """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_stitcher_amd.executors import pin_process_to_compact_cpus
pin_process_to_compact_cpus()
import numpy as np, torch
import bench
from multiview_stitcher_amd import _lib, registration
from multiview_stitcher_amd import spatial_image_utils as si

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
overlap = np.round(tile * 0.2).astype(int)
key = si.DEFAULT_TRANSFORM_KEY
lanes = [lane << 8 for lane in range(16)]
bad = 0
for seed in range(nseeds):
    tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=500 + seed, max_jitter=3 + seed % 3)[:3]
    for noise in (0, 100, 400):
        if noise:
            g = torch.Generator(device=dev); g.manual_seed(seed)
            for t in tiles:
                ti = t.view(torch.int16)
                n = torch.randn(t.shape, generator=g, device=dev, dtype=torch.float16) * noise
                ti.copy_((ti.to(torch.float32) + n.to(torch.float32)).clamp_(0, 30000).to(torch.int16)); del n
        sims = bench.build_sims(tiles, origins, 0)
        torch.cuda.synchronize()
        out = {}
        for flag in (1, 0):
            for d in lanes:
                _lib.init(d); _lib.set_option("ssim_prune", flag, d)
                for k in ("reg_pruned", "reg_cand_volumes", "reg_candidates"): _lib.get_counter(k, d, reset=True)
            res = registration.register(sims, transform_key=key, new_transform_key="reg", device=0, return_dict=True)
            st = {k: sum(_lib.get_counter(k, d, reset=True) for d in lanes) for k in ("reg_pruned", "reg_cand_volumes", "reg_candidates")}
            out[flag] = (res, st)
        r1, r0 = out[1][0]["pairwise_registration"]["results"][0], out[0][0]["pairwise_registration"]["results"][0]
        same = all(np.array_equal(np.asarray(a["transform"]), np.asarray(b["transform"])) and (a["quality"] == b["quality"] or (a["quality"] != a["quality"] and b["quality"] != b["quality"])) for a, b in zip(r1, r0))
        q = np.array([a["quality"] for a in r1])
        st = out[1][1]
        print(f"seed {seed} noise {noise}: identical {same}; pairs {len(r1)}; candidates {st['reg_candidates']:.0f}, left unfinished {st['reg_pruned']:.0f}, "
              f"volumes walked {st['reg_cand_volumes']:.1f}; quality min {np.nanmin(q):.3f} median {np.nanmedian(q):.3f}", flush=True)
        bad += 0 if same else 1
for d in lanes:
    _lib.set_option("ssim_prune", 1, d)
print("mismatching runs:", bad)
