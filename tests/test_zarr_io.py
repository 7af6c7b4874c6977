"""CPU tests of the Zarr-v2 / NGFF-0.4 restatement: byte layout against the storage specification
(known-answer files written by hand), windowed reads / writes against numpy, and the multiscales document against the
reference's conventions (ngff_utils.py:1185-1230, 1493-1561; msi_utils.py:279-325)."""
import gzip
import json
import os
import zlib

import numpy as np
import pytest

from multiview_stitcher_amd import msi_utils, ngff_utils, zarr_io
from multiview_stitcher_amd import spatial_image_utils as si


def test_known_answer_store_layout(tmp_path):
    # a 5x7 uint16 array in 2x4 chunks, "." separator: 3x2 chunk files, edge chunks padded to the full chunk shape
    a = np.arange(35, dtype="<u2").reshape(5, 7)
    z = zarr_io.ZarrArray.create(tmp_path / "a.zarr", a.shape, (2, 4), a.dtype, fill_value=9)
    z[...] = a
    meta = json.load(open(tmp_path / "a.zarr" / ".zarray"))
    assert meta == {"zarr_format": 2, "shape": [5, 7], "chunks": [2, 4], "dtype": "<u2", "order": "C", "fill_value": 9,
                    "compressor": None, "filters": None}
    assert sorted(f for f in os.listdir(tmp_path / "a.zarr") if not f.startswith(".")) == ["0.0", "0.1", "1.0", "1.1", "2.0", "2.1"]
    raw = np.frombuffer(open(tmp_path / "a.zarr" / "2.1", "rb").read(), dtype="<u2").reshape(2, 4)
    np.testing.assert_array_equal(raw, [[32, 33, 34, 9], [9, 9, 9, 9]])   # row 4, cols 4..6, then fill
    raw = np.frombuffer(open(tmp_path / "a.zarr" / "0.1", "rb").read(), dtype="<u2").reshape(2, 4)
    np.testing.assert_array_equal(raw, [[4, 5, 6, 9], [11, 12, 13, 9]])


def test_reads_a_store_written_by_hand(tmp_path):
    # what zarr-python 2 writes for zarr.open(..., shape=(3, 4), chunks=(2, 2), dtype="f4", compressor=Zlib(1),
    # dimension_separator="/"): nested chunk keys, zlib streams, NaN fill as the string "NaN"; chunk 1/0 missing
    root = tmp_path / "h.zarr"
    os.makedirs(root / "0")
    os.makedirs(root / "1")
    json.dump({"zarr_format": 2, "shape": [3, 4], "chunks": [2, 2], "dtype": "<f4", "order": "C", "fill_value": "NaN",
               "compressor": {"id": "zlib", "level": 1}, "filters": None, "dimension_separator": "/"}, open(root / ".zarray", "w"))
    full = np.arange(12, dtype="<f4").reshape(3, 4)
    pad = np.full((4, 4), np.nan, "<f4")
    pad[:3] = full
    for i, j in [(0, 0), (0, 1), (1, 1)]:
        open(root / str(i) / str(j), "wb").write(zlib.compress(np.ascontiguousarray(pad[2 * i:2 * i + 2, 2 * j:2 * j + 2]).tobytes(), 1))
    z = zarr_io.ZarrArray.open(root)
    got = np.asarray(z)
    want = full.copy()
    want[2, 0:2] = np.nan
    np.testing.assert_array_equal(got, want)
    assert np.asarray(z[1:, 1:3]).shape == (2, 2)
    np.testing.assert_array_equal(np.asarray(z[1, 2:]), [6, 7])


@pytest.mark.parametrize("compressor", [None, {"id": "zlib", "level": 1}, {"id": "gzip", "level": 1}])
@pytest.mark.parametrize("sep", [".", "/"])
def test_windows_against_numpy(tmp_path, compressor, sep):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 60000, (3, 37, 29, 41)).astype(np.uint16)
    z = zarr_io.ZarrArray.create(tmp_path / "w.zarr", a.shape, (1, 16, 8, 16), a.dtype, compressor=compressor, dimension_separator=sep)
    ref = np.zeros_like(a)
    for _ in range(20):    # random, mostly unaligned writes: exercises the read-patch-write path
        lo = [int(rng.integers(0, s)) for s in a.shape]
        hi = [int(rng.integers(l + 1, s + 1)) for l, s in zip(lo, a.shape)]
        sl = tuple(slice(l, h) for l, h in zip(lo, hi))
        z[sl] = a[sl]
        ref[sl] = a[sl]
    np.testing.assert_array_equal(np.asarray(z), ref)
    z[...] = a
    for _ in range(20):
        lo = [int(rng.integers(0, s)) for s in a.shape]
        hi = [int(rng.integers(l, s + 1)) for l, s in zip(lo, a.shape)]
        sl = tuple(slice(l, h) for l, h in zip(lo, hi))
        np.testing.assert_array_equal(np.asarray(z[sl]), a[sl])
    # composed lazy views, ints drop axes, negative indices, ellipsis
    v = z[1][5:30, :, 3:]
    assert v.shape == (25, 29, 38) and v.dtype == np.uint16
    np.testing.assert_array_equal(np.asarray(v[2:4, -1, ...]), a[1, 7:9, -1, 3:])
    if compressor and compressor["id"] == "gzip":
        assert gzip.decompress(open(z.chunk_path((0, 0, 0, 0)), "rb").read())[:2] == a[0, 0, 0, :1].tobytes()


def test_unknown_codec_and_format_fail_loudly(tmp_path):
    with pytest.raises(NotImplementedError, match="lzma"):
        zarr_io.ZarrArray.create(tmp_path / "b.zarr", (4,), (2,), "u1", compressor={"id": "lzma"})
    os.makedirs(tmp_path / "v4.zarr")
    json.dump({"zarr_format": 4, "shape": [1], "chunks": [1], "dtype": "|u1"}, open(tmp_path / "v4.zarr" / ".zarray", "w"))
    with pytest.raises(NotImplementedError):
        zarr_io.ZarrArray.open(tmp_path / "v4.zarr")
    z = zarr_io.ZarrArray.create(tmp_path / "c.zarr", (4, 4), (2, 2), "u1")
    with pytest.raises(IndexError):
        z.write([3, 3], np.zeros((2, 2), "u1"))


def test_resolution_levels_kat():
    # msi_utils.py:279-325: halve while shape // 2 > 100
    shapes, rel, ab = msi_utils.calc_resolution_levels({"z": 120, "y": 1000, "x": 450})
    assert shapes == [{"z": 120, "y": 1000, "x": 450}, {"z": 120, "y": 500, "x": 225}, {"z": 120, "y": 250, "x": 112}, {"z": 120, "y": 125, "x": 112}]
    assert rel[1:] == [{"z": 1, "y": 2, "x": 2}, {"z": 1, "y": 2, "x": 2}, {"z": 1, "y": 2, "x": 1}]
    assert ab[-1] == {"z": 1, "y": 8, "x": 4}
    assert msi_utils.calc_resolution_levels({"y": 150, "x": 150})[0] == [{"y": 150, "x": 150}]


def test_multiscales_document(tmp_path):
    sp, o = {"z": 2.0, "y": 0.5, "x": 0.5}, {"z": 10.0, "y": -3.0, "x": 4.0}
    _, _, ab = msi_utils.calc_resolution_levels({"z": 50, "y": 420, "x": 420})
    tfs, axes = ngff_utils.calc_ngff_coordinate_transformations_and_axes({"spacing": sp, "origin": o}, ab, nsdims=["t", "c"])
    assert axes == [{"name": "t", "type": "time"}, {"name": "c", "type": "channel"},
                    {"name": "z", "type": "space", "unit": "micrometer"}, {"name": "y", "type": "space", "unit": "micrometer"},
                    {"name": "x", "type": "space", "unit": "micrometer"}]
    assert tfs[0] == [{"type": "scale", "scale": [1.0, 1.0, 2.0, 0.5, 0.5]}, {"type": "translation", "translation": [0.0, 0, 10.0, -3.0, 4.0]}]
    # level 2: y, x factor 4 -> spacing 2.0, origin shifted by (4 - 1) * 0.5 / 2
    assert tfs[2] == [{"type": "scale", "scale": [1.0, 1.0, 2.0, 2.0, 2.0]}, {"type": "translation", "translation": [0.0, 0, 10.0, -2.25, 4.75]}]
    g = zarr_io.create_group(str(tmp_path / "g.zarr"))
    ngff_utils.write_multiscales_metadata(g, axes, [{"path": str(i), "coordinateTransformations": t} for i, t in enumerate(tfs)])
    doc = json.load(open(tmp_path / "g.zarr" / ".zattrs"))
    assert list(doc) == ["multiscales"] and len(doc["multiscales"]) == 1
    ms = doc["multiscales"][0]
    assert sorted(ms) == ["axes", "datasets", "name", "version"] and ms["version"] == "0.4"
    assert [d["path"] for d in ms["datasets"]] == ["0", "1", "2"]
    assert json.load(open(tmp_path / "g.zarr" / ".zgroup")) == {"zarr_format": 2}


def test_lazy_sim_round_trip_without_gpu(tmp_path):
    # level 0 only (shape too small for a pyramid -> no kernel needed): write, read back lazily, select like fuse does
    a = np.random.default_rng(1).integers(0, 4000, (1, 1, 20, 90, 80)).astype(np.uint16)
    sim = si.to_spatial_image(a, dims=["c", "t", "z", "y", "x"], scale={"z": 2.0, "y": 1.0, "x": 1.0}, translation={"z": 5.0, "y": 7.0, "x": -2.0})
    back = ngff_utils.write_sim_to_ome_zarr(sim, str(tmp_path / "s.zarr"))
    assert zarr_io.is_zarr_backed(back.data) and back.dims == ("c", "t", "z", "y", "x")
    assert si.get_spacing_from_sim(back) == {"z": 2.0, "y": 1.0, "x": 1.0}
    assert si.get_origin_from_sim(back) == {"z": 5.0, "y": 7.0, "x": -2.0}
    assert json.load(open(tmp_path / "s.zarr" / "0" / ".zarray"))["dimension_separator"] == "/"
    assert os.path.exists(tmp_path / "s.zarr" / "0" / "0" / "0" / "0" / "0" / "0")
    field = back.isel({"c": 0, "t": 0})
    slab = field.sel({"z": slice(9.0, 21.0), "y": slice(None, 20.0), "x": slice(0.0, None)})
    assert zarr_io.is_zarr_backed(slab.data)
    np.testing.assert_array_equal(np.asarray(slab.data), a[0, 0, 2:9, :14, 2:])
    ms = ngff_utils.read_msim_from_ome_zarr(str(tmp_path / "s.zarr"))
    assert ms.keys() == ["scale0"]


# ---- zstd / lz4 / blosc chunk codecs (zarr_codecs.py) -------------------------------------------------------------------
def test_zstd_and_lz4_frames_written_by_pyarrow_decode():
    """numcodecs.Zstd is a plain zstd frame, numcodecs.LZ4 a 4-byte little-endian size + one LZ4 block: frames produced by
    pyarrow's bundled encoders (third-party code) must decode, and chunks stored with these compressors must read back."""
    pa = pytest.importorskip("pyarrow")
    from multiview_stitcher_amd import zarr_codecs as zc

    rng = np.random.default_rng(0)
    raw = np.cumsum(rng.integers(-3, 4, 5000), dtype=np.int64).astype(np.uint16).tobytes()
    z = pa.Codec("zstd", compression_level=3).compress(raw, asbytes=True)
    assert zc.zstd_decode(z, len(raw)) == raw and len(z) < len(raw)
    l4 = len(raw).to_bytes(4, "little") + pa.Codec("lz4_raw").compress(raw, asbytes=True)
    assert zc.lz4_decode(l4) == raw
    assert zc.lz4_decode(zc.lz4_encode(raw)) == raw and zc.zstd_decode(zc.zstd_encode(raw, 5), len(raw)) == raw


@pytest.mark.parametrize("comp", [{"id": "zstd", "level": 3}, {"id": "lz4"},
                                  {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0},
                                  {"id": "blosc", "cname": "zstd", "clevel": 3, "shuffle": 2, "blocksize": 0},
                                  {"id": "blosc", "cname": "zlib", "clevel": 4, "shuffle": 0, "blocksize": 4096}])
def test_compressed_store_round_trip(tmp_path, comp):
    pytest.importorskip("pyarrow")
    from multiview_stitcher_amd import zarr_io

    rng = np.random.default_rng(1)
    data = np.cumsum(rng.integers(-2, 3, (37, 50, 41)), axis=2).astype(np.uint16)
    arr = zarr_io.ZarrArray.create(str(tmp_path / "a"), data.shape, (16, 32, 24), data.dtype, compressor=comp)
    arr.write([0, 0, 0], data)
    back = zarr_io.ZarrArray.open(str(tmp_path / "a"))
    assert back.meta["compressor"] == comp
    np.testing.assert_array_equal(back[...], data)
    sizes = [os.path.getsize(os.path.join(str(tmp_path / "a"), f)) for f in os.listdir(str(tmp_path / "a")) if not f.startswith(".")]
    assert max(sizes) < 16 * 32 * 24 * 2                       # the chunks really are compressed


def test_blosc_container_hand_assembled_frames():
    """Frames put together byte by byte from the published header layout (c-blosc README_HEADER.rst): a memcpyed frame,
    a one-block frame whose single stream is stored raw, and a byte-shuffled two-stream block."""
    import struct

    from multiview_stitcher_amd import zarr_codecs as zc

    payload = bytes(range(200))
    memcpyed = struct.pack("<BBBBIII", 2, 1, 0x2 | 0x1 | (1 << 5), 2, 200, 200, 216) + payload
    assert zc.blosc_decode(memcpyed) == payload
    # one block, dont-split flag, stream stored raw (cbytes == block size), no shuffle
    raw1 = struct.pack("<BBBBIII", 2, 1, 0x10 | (1 << 5), 4, 200, 200, 16 + 4 + 4 + 200) + struct.pack("<i", 20) + struct.pack("<i", 200) + payload
    assert zc.blosc_decode(raw1) == payload
    # typesize 2, 256 elements, byte shuffle, split into 2 streams (256 >= MIN_BUFFERSIZE 128), both stored raw
    elems = np.arange(256, dtype="<u2") * 259
    shuffled = elems.view(np.uint8).reshape(256, 2).T.tobytes()
    frame = struct.pack("<BBBBIII", 2, 1, 0x1 | (1 << 5), 2, 512, 512, 16 + 4 + 2 * (4 + 256)) + struct.pack("<i", 20)
    frame = frame + struct.pack("<i", 256) + shuffled[:256] + struct.pack("<i", 256) + shuffled[256:]
    assert zc.blosc_decode(frame) == elems.tobytes()
    with pytest.raises(NotImplementedError):
        zc.blosc_decode(struct.pack("<BBBBIII", 2, 1, 0x10, 1, 8, 8, 16 + 4 + 4 + 3) + struct.pack("<i", 20) + struct.pack("<i", 3) + b"abc")   # blosclz


# ---- Zarr v3 / NGFF 0.5 (reference: ngff_utils.py:1185-1281, 1820-1905) -------------------------------------------------
def _v3_meta(shape, chunks, data_type, codecs, fill=0, enc=None):
    return {"zarr_format": 3, "node_type": "array", "shape": shape, "data_type": data_type,
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": chunks}},
            "chunk_key_encoding": enc or {"name": "default", "configuration": {"separator": "/"}},
            "fill_value": fill, "codecs": codecs, "attributes": {"note": "by hand"}, "dimension_names": ["y", "x"]}


def test_v3_reads_a_store_written_by_hand(tmp_path):
    # what zarr-python 3 writes for a (3, 4) float32 array in (2, 2) chunks with [bytes, gzip]: zarr.json, chunk keys c/<i>/<j>,
    # little-endian payload, NaN fill as the string "NaN"; chunk c/1/0 missing
    root = tmp_path / "h3.zarr"
    os.makedirs(root / "c" / "0")
    os.makedirs(root / "c" / "1")
    json.dump(_v3_meta([3, 4], [2, 2], "float32", [{"name": "bytes", "configuration": {"endian": "little"}},
                                                     {"name": "gzip", "configuration": {"level": 1}}], fill="NaN"),
              open(root / "zarr.json", "w"))
    full = np.arange(12, dtype="<f4").reshape(3, 4)
    pad = np.full((4, 4), np.nan, "<f4")
    pad[:3] = full
    for i, j in [(0, 0), (0, 1), (1, 1)]:
        open(root / "c" / str(i) / str(j), "wb").write(gzip.compress(np.ascontiguousarray(pad[2 * i:2 * i + 2, 2 * j:2 * j + 2]).tobytes(), 1))
    z = zarr_io.ZarrArray.open(root)
    assert z.zarr_format == 3 and z.chunks == (2, 2) and z.dtype == np.float32
    want = full.copy()
    want[2, 0:2] = np.nan
    np.testing.assert_array_equal(np.asarray(z), want)
    assert zarr_io.read_attrs(str(root)) == {"note": "by hand"} and zarr_io.array_exists(str(root))
    # "v2" chunk key encoding: <i>.<j> next to zarr.json
    root2 = tmp_path / "k.zarr"
    os.makedirs(root2)
    json.dump(_v3_meta([2, 2], [2, 2], "uint8", [{"name": "bytes"}], enc={"name": "v2", "configuration": {"separator": "."}}),
              open(root2 / "zarr.json", "w"))
    open(root2 / "0.0", "wb").write(bytes([1, 2, 3, 4]))
    np.testing.assert_array_equal(np.asarray(zarr_io.ZarrArray.open(root2)), [[1, 2], [3, 4]])


def test_v3_known_answer_store_layout_and_unsupported_pieces(tmp_path):
    a = np.arange(35, dtype="<u2").reshape(5, 7)
    z = zarr_io.ZarrArray.create(tmp_path / "a3.zarr", a.shape, (2, 4), a.dtype, fill_value=9, zarr_format=3, dimension_names=["y", "x"])
    z[...] = a
    meta = json.load(open(tmp_path / "a3.zarr" / "zarr.json"))
    assert meta == {"zarr_format": 3, "node_type": "array", "shape": [5, 7], "data_type": "uint16",
                    "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": [2, 4]}},
                    "chunk_key_encoding": {"name": "default", "configuration": {"separator": "/"}}, "fill_value": 9,
                    "codecs": [{"name": "bytes", "configuration": {"endian": "little"}}], "attributes": {}, "dimension_names": ["y", "x"]}
    assert sorted(os.listdir(tmp_path / "a3.zarr")) == ["c", "zarr.json"] and sorted(os.listdir(tmp_path / "a3.zarr" / "c")) == ["0", "1", "2"]
    raw = np.frombuffer(open(tmp_path / "a3.zarr" / "c" / "2" / "1", "rb").read(), dtype="<u2").reshape(2, 4)
    np.testing.assert_array_equal(raw, [[32, 33, 34, 9], [9, 9, 9, 9]])
    np.testing.assert_array_equal(np.asarray(zarr_io.ZarrArray.open(tmp_path / "a3.zarr")), a)
    for comp in ({"id": "zstd", "level": 3}, {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1}, {"id": "gzip", "level": 2}):
        zc = zarr_io.ZarrArray.create(tmp_path / f"c_{comp['id']}.zarr", a.shape, (2, 4), a.dtype, compressor=comp, zarr_format=3)
        zc[...] = a
        back = zarr_io.ZarrArray.open(tmp_path / f"c_{comp['id']}.zarr")
        assert [c["name"] for c in back.meta_v3["codecs"]] == ["bytes", comp["id"]]
        np.testing.assert_array_equal(np.asarray(back), a)
    for bad in ([{"name": "transpose", "configuration": {"order": [1, 0]}}, {"name": "bytes"}],
                [{"name": "sharding_indexed", "configuration": {}}],
                [{"name": "bytes", "configuration": {"endian": "little"}}, {"name": "crc32c"}]):
        os.makedirs(tmp_path / "bad.zarr", exist_ok=True)
        json.dump(_v3_meta([2, 2], [2, 2], "uint16", bad), open(tmp_path / "bad.zarr" / "zarr.json", "w"))
        with pytest.raises(NotImplementedError):
            zarr_io.ZarrArray.open(tmp_path / "bad.zarr")


def test_ngff_05_round_trip_without_gpu(tmp_path):
    a = np.random.default_rng(2).integers(0, 4000, (1, 1, 20, 90, 80)).astype(np.uint16)
    sim = si.to_spatial_image(a, dims=["c", "t", "z", "y", "x"], scale={"z": 2.0, "y": 1.0, "x": 1.0}, translation={"z": 5.0, "y": 7.0, "x": -2.0})
    url = str(tmp_path / "s5.zarr")
    back = ngff_utils.write_sim_to_ome_zarr(sim, url, ngff_version="0.5", zarr_array_creation_kwargs={"compressor": {"id": "zstd", "level": 1}})
    grp = json.load(open(os.path.join(url, "zarr.json")))
    assert grp["zarr_format"] == 3 and grp["node_type"] == "group" and list(grp["attributes"]) == ["ome"]
    ome = grp["attributes"]["ome"]
    assert ome["version"] == "0.5" and sorted(ome["multiscales"][0]) == ["axes", "datasets", "name"]       # version sits in "ome" only
    arr = json.load(open(os.path.join(url, "0", "zarr.json")))
    assert arr["dimension_names"] == ["c", "t", "z", "y", "x"] and [c["name"] for c in arr["codecs"]] == ["bytes", "zstd"]
    assert not os.path.exists(os.path.join(url, ".zattrs")) and not os.path.exists(os.path.join(url, "0", ".zarray"))
    assert zarr_io.is_zarr_backed(back.data) and si.get_origin_from_sim(back) == {"z": 5.0, "y": 7.0, "x": -2.0}
    np.testing.assert_array_equal(np.asarray(back.data), a)
    lazy = ngff_utils.read_sim_from_ome_zarr(url, 0)
    np.testing.assert_array_equal(np.asarray(lazy.isel({"c": 0, "t": 0}).sel({"z": slice(9.0, 21.0)}).data), a[0, 0, 2:9])
