"""GPU tests of the streaming ends of the fuse path: zarr-backed tiles in, Zarr v2 / NGFF 0.4 out,
pyramid levels by the device block-mean kernel.  Everything is compared with the in-memory workflow on the same tiles."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dataset(ndim):
    from multiview_stitcher_amd import sample_data

    if ndim == 2:
        return sample_data.generate_tiled_dataset(ndim=2, tile_shape=(256, 224), tiles=(2, 2), overlap=(48, 40), dtype=np.uint16,
                                                  max_jitter=2, seed=4, jitter_in_metadata=True)[0]
    return sample_data.generate_tiled_dataset(ndim=3, tile_shape=(40, 128, 144), tiles=(1, 2, 2), overlap=(0, 24, 32), dtype=np.uint16,
                                              max_jitter=2, seed=6, jitter_in_metadata=True)[0]


def _block_mean(a, f):
    """np.mean over blocks, trimmed, cast back (ngff_utils.py:1284-1330)."""
    sl = tuple(slice(0, (n // k) * k) for n, k in zip(a.shape, f))
    a = a[sl]
    shp = []
    for n, k in zip(a.shape, f):
        shp += [n // k, k]
    return a.reshape(shp).mean(axis=tuple(range(1, 2 * a.ndim, 2))).astype(a.dtype)


@pytest.mark.parametrize("ndim", [2, 3])
def test_fuse_streams_zarr_in_and_out(hip_device, tmp_path, ndim):
    from multiview_stitcher_amd import fusion, ngff_utils, sample_data, zarr_io, spatial_image_utils as si

    key = sample_data.METADATA_TRANSFORM_KEY
    sims = _dataset(ndim)
    chunks = {"y": 128, "x": 96} if ndim == 2 else {"z": 32, "y": 96, "x": 128}
    want = np.asarray(fusion.fuse(sims, transform_key=key, output_chunksize=chunks).data)      # (chunks merged into one launch block)

    # tiles -> OME-Zarr stores -> lazy sims carrying the stage transform
    lazy = []
    for i, s in enumerate(sims):
        z = ngff_utils.write_sim_to_ome_zarr(s, str(tmp_path / f"tile{i}.zarr"))
        assert zarr_io.is_zarr_backed(z.data)
        si.set_sim_affine(z, si.get_affine_from_sim(s, key), key)
        lazy.append(z)
    out_url = str(tmp_path / "fused.zarr")
    fused = fusion.fuse(lazy, transform_key=key, output_chunksize=chunks, output_zarr_url=out_url,
                        zarr_options={"ome_zarr": True})
    assert zarr_io.is_zarr_backed(fused.data)
    np.testing.assert_array_equal(np.asarray(fused.data), want)
    # chunk by chunk (merge_chunks=False): the streamed result equals the chunk-by-chunk in-memory result
    fused_c = fusion.fuse(lazy, transform_key=key, output_chunksize=chunks, output_zarr_url=str(tmp_path / "fused_c.zarr"), merge_chunks=False)
    np.testing.assert_array_equal(np.asarray(fused_c.data),
                                  np.asarray(fusion.fuse(sims, transform_key=key, output_chunksize=chunks, merge_chunks=False).data))
    assert si.get_origin_from_sim(fused) == pytest.approx(si.get_origin_from_sim(fusion.fuse(sims, transform_key=key, output_chunksize=chunks)))

    # store layout: group, level arrays with '/' keys and the fuse chunk grid, multiscales document
    meta0 = json.load(open(os.path.join(out_url, "0", ".zarray")))
    sd = ["z", "y", "x"][-ndim:]
    assert meta0["chunks"] == [1, 1] + [chunks[d] for d in sd] and meta0["dimension_separator"] == "/"
    ms = json.load(open(os.path.join(out_url, ".zattrs")))["multiscales"][0]
    assert [a["name"] for a in ms["axes"]] == ["c", "t"] + sd
    nlev = len(ms["datasets"])
    assert nlev >= 2
    # pyramid: every level is the block mean of the previous one (device kernel vs numpy)
    msim = ngff_utils.read_msim_from_ome_zarr(out_url)
    prev = want
    for lev in range(1, nlev):
        cur = np.asarray(msim[f"scale{lev}"].data)
        f = [1, 1] + [p // c for p, c in zip(prev.shape[2:], cur.shape[2:])]
        np.testing.assert_array_equal(cur, _block_mean(prev, f))
        s0, s1 = ms["datasets"][lev - 1]["coordinateTransformations"], ms["datasets"][lev]["coordinateTransformations"]
        for ax in range(2, 2 + ndim):
            assert s1[0]["scale"][ax] == pytest.approx(s0[0]["scale"][ax] * f[ax])
            assert s1[1]["translation"][ax] == pytest.approx(s0[1]["translation"][ax] + (f[ax] - 1) * s0[0]["scale"][ax] / 2)
        prev = cur


def test_plain_zarr_output_and_chunk_farm(hip_device, tmp_path):
    """Without ome_zarr the array sits directly under the url; two workers with complementary chunk filters fill one
    store (the multi-GPU farm's pattern: disjoint chunk files, no merge step)."""
    from multiview_stitcher_amd import fusion, sample_data, zarr_io

    key = sample_data.METADATA_TRANSFORM_KEY
    sims = _dataset(2)
    chunks = {"y": 128, "x": 96}
    want = np.asarray(fusion.fuse(sims, transform_key=key, output_chunksize=chunks, merge_chunks=False).data)
    url = str(tmp_path / "plain.zarr")
    for part in (0, 1):
        fusion.fuse(sims, transform_key=key, output_chunksize=chunks, output_zarr_url=url,
                    chunk_filter=lambda bi, part=part: (bi[-1] + bi[-2]) % 2 == part)
    z = zarr_io.ZarrArray.open(url)
    assert "dimension_separator" not in z.meta
    np.testing.assert_array_equal(np.asarray(z), want)


def test_register_reads_zarr_backed_tiles(hip_device, tmp_path):
    from multiview_stitcher_amd import msi_utils, ngff_utils, param_utils, registration, sample_data, spatial_image_utils as si

    key = sample_data.METADATA_TRANSFORM_KEY
    sims, jit, _ = sample_data.generate_tiled_dataset(ndim=2, tile_shape=(128, 160), tiles=(2, 2), overlap=(40, 48), dtype=np.uint16,
                                                      max_jitter=3, seed=2)
    want = registration.register([msi_utils.get_msim_from_sim(s) for s in sims], transform_key=key, new_transform_key="reg", reg_channel_index=0)
    lazy = []
    for i, s in enumerate(sims):
        z = ngff_utils.write_sim_to_ome_zarr(s, str(tmp_path / f"t{i}.zarr"))
        si.set_sim_affine(z, si.get_affine_from_sim(s, key), key)
        lazy.append(msi_utils.get_msim_from_sim(z))
    got = registration.register(lazy, transform_key=key, new_transform_key="reg", reg_channel_index=0)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(param_utils.select_time(a, 0), param_utils.select_time(b, 0))


def test_device_farm_streams_into_one_store(hip_device, tmp_path):
    """executors.fuse_on_devices with output_zarr_url: the workers' chunk files add up to the single-device result and the
    pyramid is completed once at the end."""
    from multiview_stitcher_amd import executors, fusion, ngff_utils, sample_data

    key = sample_data.METADATA_TRANSFORM_KEY
    sims = _dataset(2)
    chunks = {"y": 128, "x": 96}
    want = np.asarray(fusion.fuse(sims, transform_key=key, output_chunksize=chunks, merge_chunks=False).data)
    url = str(tmp_path / "farm.zarr")
    os.makedirs(url)
    open(os.path.join(url, "stale"), "w").write("x")      # overwrite=True (default) clears what was there
    out = executors.fuse_on_devices(sims, devices=(0, 0 | 1 << 8), transform_key=key, output_chunksize=chunks, output_zarr_url=url,
                                    zarr_options={"ome_zarr": True})
    assert not os.path.exists(os.path.join(url, "stale"))
    np.testing.assert_array_equal(np.asarray(out.data), want)
    assert len(ngff_utils.read_msim_from_ome_zarr(url).keys()) >= 2


def test_batch_options_drive_block_wise_zarr_output(hip_device, tmp_path):
    """fuse(..., output_zarr_url=, batch_options={"batch_func": f, "n_batch": k}) calls f(fuse_chunk, block_ids, **kw)
    with every block id of the output chunk grid exactly once, in batches of k (fusion/_core.py:1123-1141;
    misc_utils.py:150-158), and fuse_chunk(block_id) writes that block's region.  The built-in GPU batch function gives
    the same store."""
    from multiview_stitcher_amd import executors, fusion, sample_data

    key = sample_data.METADATA_TRANSFORM_KEY
    sims = _dataset(3)
    chunks = {"z": 32, "y": 96, "x": 128}
    want = np.asarray(fusion.fuse(sims, transform_key=key, output_chunksize=chunks, merge_chunks=False).data)
    seen, batch_sizes = [], []

    def spy(fuse_chunk, block_ids, tag=None):
        assert tag == "t"
        batch_sizes.append(len(block_ids))
        for b in block_ids:
            seen.append(tuple(b))
            assert fuse_chunk(b) is None

    fused = fusion.fuse(sims, transform_key=key, output_chunksize=chunks, output_zarr_url=str(tmp_path / "a.zarr"),
                        batch_options={"batch_func": spy, "n_batch": 4, "batch_func_kwargs": {"tag": "t"}})
    grid = tuple(-(-n // c) for n, c in zip(want.shape, [1, 1] + [chunks[d] for d in "zyx"]))
    assert sorted(seen) == sorted(np.ndindex(*grid)) and len(seen) == len(set(seen))
    assert all(b == 4 for b in batch_sizes[:-1]) and 1 <= batch_sizes[-1] <= 4
    np.testing.assert_array_equal(np.asarray(fused.data), want)

    fused2 = fusion.fuse(sims, transform_key=key, output_chunksize=chunks, output_zarr_url=str(tmp_path / "b.zarr"),
                         zarr_options={"ome_zarr": True},
                         batch_options={"batch_func": executors.process_batch_using_gpus, "n_batch": 6,
                                        "batch_func_kwargs": {"devices": (0, 0 | (1 << 8))}})
    np.testing.assert_array_equal(np.asarray(fused2.data), want)

    with pytest.raises(ValueError):
        fusion.fuse(sims, transform_key=key, batch_options={"n_batch": 2})           # needs output_zarr_url
    with pytest.raises(TypeError):
        fusion.fuse(sims, transform_key=key, output_zarr_url=str(tmp_path / "c.zarr"), batch_options={"nbatch": 2})


def test_fuse_streams_into_ngff_05_zarr_v3(hip_device, tmp_path):
    """zarr_options={"ome_zarr": True, "ngff_version": "0.5"}: the fused chunks go into a Zarr v3 hierarchy (zarr.json nodes,
    c/<i>/... chunk keys, metadata under the "ome" key) with a compressed pyramid; reading it back gives the in-memory result."""
    from multiview_stitcher_amd import fusion, ngff_utils, sample_data, zarr_io

    key = sample_data.METADATA_TRANSFORM_KEY
    sims = _dataset(3)
    chunks = {"z": 32, "y": 96, "x": 128}
    want = np.asarray(fusion.fuse(sims, transform_key=key, output_chunksize=chunks).data)
    url = str(tmp_path / "v3.zarr")
    fused = fusion.fuse(sims, transform_key=key, output_chunksize=chunks, output_zarr_url=url,
                        zarr_options={"ome_zarr": True, "ngff_version": "0.5",
                                      "zarr_array_creation_kwargs": {"compressor": {"id": "blosc", "cname": "zstd", "clevel": 3, "shuffle": 1}}})
    assert zarr_io.is_zarr_backed(fused.data)
    np.testing.assert_array_equal(np.asarray(fused.data), want)
    grp = json.load(open(os.path.join(url, "zarr.json")))
    assert grp["node_type"] == "group" and grp["attributes"]["ome"]["version"] == "0.5"
    arr = json.load(open(os.path.join(url, "0", "zarr.json")))
    assert arr["chunk_grid"]["configuration"]["chunk_shape"] == [1, 1, 32, 96, 128] and [c["name"] for c in arr["codecs"]] == ["bytes", "blosc"]
    assert os.path.isdir(os.path.join(url, "0", "c"))
    ms = ngff_utils.read_msim_from_ome_zarr(url)
    lvl1 = ms["scale1"] if "scale1" in ms.keys() else None
    if lvl1 is not None:
        np.testing.assert_array_equal(np.asarray(lvl1.data)[0, 0], _block_mean(want[0, 0], (1, 2, 2))[: lvl1.data.shape[2]])


def test_fuse_zarr_creation_kwargs_chunks_and_format(hip_device, tmp_path):
    """zarr_array_creation_kwargs={"chunks": ...} names the store's chunk grid (spatial or full rank), as in
    write_sim_to_ome_zarr; a zarr_format that contradicts the NGFF version is refused."""
    from multiview_stitcher_amd import fusion, sample_data, zarr_io

    key = sample_data.METADATA_TRANSFORM_KEY
    sims = _dataset(2)
    want = np.asarray(fusion.fuse(sims, transform_key=key, output_chunksize={"y": 128, "x": 96}).data)
    url = str(tmp_path / "f.zarr")
    fused = fusion.fuse(sims, transform_key=key, output_chunksize={"y": 128, "x": 96}, output_zarr_url=url,
                        zarr_options={"ome_zarr": True, "zarr_array_creation_kwargs": {"chunks": (64, 48)}})
    np.testing.assert_array_equal(np.asarray(fused.data), want)
    assert list(zarr_io.ZarrArray.open(os.path.join(url, "0")).chunks[-2:]) == [64, 48]
    with pytest.raises(ValueError):
        fusion.fuse(sims, transform_key=key, output_chunksize={"y": 128, "x": 96}, output_zarr_url=str(tmp_path / "g.zarr"),
                    zarr_options={"ome_zarr": True, "ngff_version": "0.4", "zarr_array_creation_kwargs": {"zarr_format": 3}})
    with pytest.raises(ValueError):
        fusion.fuse(sims, transform_key=key, output_chunksize={"y": 128, "x": 96}, output_zarr_url=str(tmp_path / "h.zarr"),
                    zarr_options={"zarr_array_creation_kwargs": {"chunks": (50, 50)}})
