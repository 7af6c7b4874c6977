"""One mosaic sharded over two ranks, emulated in one process on one GPU (two context lanes): every rank holds only its
tiles + halo (the other views are metadata-only RemoteArrays), registers the pairs whose fixed view it owns and fuses its
sub-box.  The gathered registration and the union of the fused sub-boxes must equal the single-device results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_rank_shard_equals_single_device(hip_device):
    from multiview_stitcher_amd import fusion, mv_graph, param_utils, registration, sample_data, sharding
    from multiview_stitcher_amd import spatial_image_utils as si
    from multiview_stitcher_amd.device import DeviceArray

    key = sample_data.METADATA_TRANSFORM_KEY
    sims, jit, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(48, 64, 64), tiles=(2, 2, 2), overlap=(12, 16, 16),
                                                      dtype=np.uint16, max_jitter=2, seed=9)
    sims = [s.isel({"c": 0, "t": 0}) for s in sims]
    world = 2

    # reference: everything on one device
    full = [s.copy(data=DeviceArray.from_host(np.ascontiguousarray(s.data), 0)) for s in sims]
    ref = registration.register(full, transform_key=key, new_transform_key="reg", device=0, return_dict=True)
    want = np.asarray(fusion.fuse(full, transform_key="reg", output_chunksize={d: 64 for d in "zyx"}, merge_chunks=False).data)

    # partition from the stage metadata
    sps = [si.get_stack_properties_from_sim(s) for s in sims]
    affs = [param_utils.select_time(si.get_affine_from_sim(s, key), 0) for s in sims]
    osp0 = fusion._bb_dicts(fusion.process_output_stack_properties(sims, None, None, None, None, "union", key), ["z", "y", "x"])
    boxes, counts = sharding.output_subboxes(osp0, world)
    owners = sharding.tile_owners(sps, affs, boxes)
    edges = ref["pairwise_registration"]["edges"]
    assert sorted(set(owners)) == [0, 1]

    # every "rank": resident tiles + halo on its own context lane, the rest metadata only
    rank_sims, mailbox = [], [None] * world
    for r in range(world):
        need = set(sharding.rank_tiles(sps, affs, boxes, edges, owners, r, margin=4.0))
        dev = 0 | (r << 8)
        rank_sims.append([
            s.copy(data=DeviceArray.from_host(np.ascontiguousarray(s.data), dev)) if v in need
            else s.copy(data=sharding.RemoteArray(s.data.shape, s.data.dtype, owner=owners[v]))
            for v, s in enumerate(sims)])

    # registration: two passes emulate the all-gather (pass 1 fills the mailbox, pass 2 reads all parts)
    class Gather:
        def __init__(self, r):
            self.r = r

        def __call__(self, payload):
            mailbox[self.r] = payload
            return [p if p is not None else {} for p in mailbox]

    for r in range(world):        # pass 1: only to fill the mailbox (the incomplete gather of rank 0 is discarded)
        ex = sharding.ShardedPairExecutor(r, world, owners, device=0 | (r << 8), gather=Gather(r))
        try:
            registration.register(rank_sims[r], transform_key=key, new_transform_key="reg", device=0 | (r << 8), pairwise_executor=ex)
        except RuntimeError:
            assert r == 0
    n_local = []
    for r in range(world):        # pass 2: every rank sees the complete mailbox
        ex = sharding.ShardedPairExecutor(r, world, owners, device=0 | (r << 8), gather=Gather(r))
        registration.register(rank_sims[r], transform_key=key, new_transform_key="reg", device=0 | (r << 8), pairwise_executor=ex)
        n_local.append(ex.last_local_count)
    assert sum(n_local) == len(edges) and all(0 < n < len(edges) for n in n_local)
    for r in range(world):
        for a, b in zip(rank_sims[r], full):
            np.testing.assert_array_equal(si.get_affine_from_sim(a, "reg"), si.get_affine_from_sim(b, "reg"))

    # fusion: every rank its sub-box; the union is the whole mosaic
    got = np.zeros_like(want)
    covered = np.zeros(want.shape[-3:], dtype=bool)
    for r in range(world):
        fused, box = sharding.fuse_shard(rank_sims[r], r, world, "reg", output_chunksize={d: 64 for d in "zyx"}, device=0 | (r << 8),
                                         merge_chunks=False)
        sl = tuple(slice(box["index_offset"][d], box["index_offset"][d] + box["shape"][d]) for d in "zyx")
        got[(Ellipsis,) + sl] = np.asarray(fused.data)
        assert not covered[sl].any()
        covered[sl] = True
    assert covered.all()
    # the merged launch block of a rank (the bench's form: output on the device) is derived once and replayed afterwards, metadata-only
    # tiles included in the geometry key: same sub-box, bit for bit
    for r in range(world):
        fusion._REPLAY_MEMO.clear()
        a, _ = sharding.fuse_shard(rank_sims[r], r, world, "reg", output_chunksize={d: 1 << 20 for d in "zyx"}, device=0 | (r << 8), output_on_backend=True)
        assert len(fusion._REPLAY_MEMO) == 1
        b, box = sharding.fuse_shard(rank_sims[r], r, world, "reg", output_chunksize={d: 1 << 20 for d in "zyx"}, device=0 | (r << 8), output_on_backend=True)
        assert len(fusion._REPLAY_MEMO) == 1
        np.testing.assert_array_equal(np.asarray(a.data), np.asarray(b.data))
        sl = tuple(slice(box["index_offset"][d], box["index_offset"][d] + box["shape"][d]) for d in "zyx")
        np.testing.assert_array_equal(np.asarray(b.data), want[(Ellipsis,) + sl].reshape(np.asarray(b.data).shape))
    np.testing.assert_array_equal(got, want)
    # one launch block per sub-box (merge_chunks, the default): the same mosaic voxel for voxel -- the sub-boxes are fused in
    # the index frame of the whole mosaic, chunks and launch blocks only shift integer indices
    for r in range(world):
        fused, box = sharding.fuse_shard(rank_sims[r], r, world, "reg", device=0 | (r << 8))
        sl = tuple(slice(box["index_offset"][d], box["index_offset"][d] + box["shape"][d]) for d in "zyx")
        np.testing.assert_array_equal(np.asarray(fused.data), want[(Ellipsis,) + sl])
