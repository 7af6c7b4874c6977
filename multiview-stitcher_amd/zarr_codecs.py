"""Chunk codecs of Zarr v2 stores beyond the standard library's zlib / gzip: zstd, lz4 and the Blosc
container, which is what OME-Zarr data in the wild is written with (reference: ngff_utils.py:1564-1973 through
zarr-python / numcodecs, neither of which exists in this image).

The entropy coders themselves are third-party code: pyarrow's bundled zstd, lz4 (raw block) and snappy, zlib from the
standard library.  What is restated here is only the framing around them:

* ``numcodecs.Zstd``  -- a plain zstd frame;
* ``numcodecs.LZ4``   -- 4-byte little-endian uncompressed size + one LZ4 block;
* ``numcodecs.Blosc`` -- the Blosc-1 container (c-blosc README_HEADER.rst / blosc.c): 16-byte header (version,
  versionlz, flags, typesize, nbytes, blocksize, cbytes), block start table, per block ``nsplits`` streams each prefixed
  by its compressed size (a stream as long as its raw size is stored uncompressed), then the byte- (or bit-) shuffle is
  undone per block.  Inner compressors lz4 / lz4hc, zstd, zlib, snappy; ``blosclz`` (Blosc's own LZ77 variant, no
  third-party decoder here) raises.  No Blosc implementation exists in this image to produce reference frames: the
  container logic is pinned by round trips through this module's own writer and by hand-assembled frames
  (tests/test_zarr_io.py) -- parity with c-blosc is by construction, stated as such.
"""

from __future__ import annotations

import struct
import zlib

import numpy as np

_BLOSC_MAX_SPLITS = 16
_BLOSC_MIN_BUFFERSIZE = 128
_BLOSC_FORMATS = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}
_BLOSC_CNAME_FORMAT = {"blosclz": 0, "lz4": 1, "lz4hc": 1, "snappy": 2, "zlib": 3, "zstd": 4}


def _pa():
    try:
        import pyarrow as pa
    except ImportError as e:   # pragma: no cover
        raise NotImplementedError("zstd / lz4 / blosc chunk codecs need pyarrow's bundled codecs") from e
    return pa


def zstd_decode(buf, nbytes):
    return _pa().decompress(bytes(buf), decompressed_size=int(nbytes), codec="zstd", asbytes=True)


def zstd_encode(buf, level=1):
    pa = _pa()
    return pa.Codec("zstd", compression_level=int(level)).compress(bytes(buf), asbytes=True)


def lz4_decode(buf, nbytes=None):
    buf = bytes(buf)
    (n,) = struct.unpack("<I", buf[:4])
    return _pa().decompress(buf[4:], decompressed_size=n, codec="lz4_raw", asbytes=True)


def lz4_encode(buf):
    buf = bytes(buf)
    return struct.pack("<I", len(buf)) + _pa().Codec("lz4_raw").compress(buf, asbytes=True)


def _inner_decode(fmt, data, nbytes):
    if fmt == "lz4":
        return _pa().decompress(data, decompressed_size=nbytes, codec="lz4_raw", asbytes=True)
    if fmt == "zstd":
        return _pa().decompress(data, decompressed_size=nbytes, codec="zstd", asbytes=True)
    if fmt == "snappy":
        return _pa().decompress(data, decompressed_size=nbytes, codec="snappy", asbytes=True)
    if fmt == "zlib":
        return zlib.decompress(data)
    raise NotImplementedError(f"blosc inner compressor {fmt!r} has no decoder in this image")


def _inner_encode(fmt, data, level):
    pa = _pa()
    if fmt == "lz4":
        return pa.Codec("lz4_raw").compress(data, asbytes=True)
    if fmt == "zstd":
        return pa.Codec("zstd", compression_level=int(level)).compress(data, asbytes=True)
    if fmt == "snappy":
        return pa.Codec("snappy").compress(data, asbytes=True)
    if fmt == "zlib":
        return zlib.compress(data, int(level))
    raise NotImplementedError(f"blosc inner compressor {fmt!r} has no encoder in this image")


def _unshuffle(block, typesize):
    n = len(block) // typesize
    a = np.frombuffer(block, dtype=np.uint8)
    body = a[: n * typesize].reshape(typesize, n).T.reshape(-1)
    return body.tobytes() + block[n * typesize:]


def _shuffle(block, typesize):
    n = len(block) // typesize
    a = np.frombuffer(block, dtype=np.uint8)
    body = a[: n * typesize].reshape(n, typesize).T.reshape(-1)
    return body.tobytes() + block[n * typesize:]


def _bitunshuffle(block, typesize):
    """Inverse of bitshuffle's bit transpose over the first 8 * (n // 8) elements (bitshuffle-generic.c): the shuffled
    buffer holds, for every bit position of the element (typesize * 8 of them, least significant bit of byte 0 first), the
    bits of all elements packed 8 per byte (element i in bit i % 8)."""
    n = (len(block) // typesize) // 8 * 8
    if n == 0:
        return block
    a = np.frombuffer(block, dtype=np.uint8)
    bits = np.unpackbits(a[: n * typesize].reshape(typesize * 8, n // 8), axis=1, bitorder="little")   # (bitpos, element)
    out = np.packbits(bits.T.reshape(n, typesize, 8), axis=2, bitorder="little").reshape(-1)
    return out.tobytes() + block[n * typesize:]


def _bitshuffle(block, typesize):
    n = (len(block) // typesize) // 8 * 8
    if n == 0:
        return block
    a = np.frombuffer(block, dtype=np.uint8)
    bits = np.unpackbits(a[: n * typesize].reshape(n, typesize, 1), axis=2, bitorder="little").reshape(n, typesize * 8)
    out = np.packbits(bits.T, axis=1, bitorder="little").reshape(-1)
    return out.tobytes() + block[n * typesize:]


def _nsplits(flags, typesize, bsize, leftover):
    dont_split = (flags >> 4) & 1
    if not dont_split and typesize <= _BLOSC_MAX_SPLITS and bsize // typesize >= _BLOSC_MIN_BUFFERSIZE and not leftover:
        return typesize
    return 1


def blosc_decode(buf, nbytes_expected=None):
    buf = bytes(buf)
    if len(buf) < 16:
        raise ValueError("blosc frame shorter than its header")
    version, versionlz, flags, typesize = buf[0], buf[1], buf[2], buf[3]
    nbytes, blocksize, cbytes = struct.unpack("<III", buf[4:16])
    if nbytes_expected is not None and nbytes != nbytes_expected:
        raise ValueError(f"blosc frame holds {nbytes} bytes, the chunk needs {nbytes_expected}")
    if nbytes == 0:
        return b""
    if flags & 0x2:                     # memcpyed: the payload follows the header verbatim
        return buf[16:16 + nbytes]
    fmt = _BLOSC_FORMATS.get(flags >> 5)
    if fmt is None:
        raise ValueError(f"unknown blosc compressor format {flags >> 5}")
    nblocks = -(-nbytes // blocksize)
    bstarts = struct.unpack(f"<{nblocks}i", buf[16:16 + 4 * nblocks])
    out = []
    for b in range(nblocks):
        leftover = (b == nblocks - 1) and (nbytes % blocksize != 0)
        bsize = nbytes - b * blocksize if b == nblocks - 1 else blocksize
        ns = _nsplits(flags, typesize, bsize if not leftover else blocksize, leftover)
        neblock = bsize // ns
        pos = bstarts[b]
        parts = []
        for _ in range(ns):
            (c,) = struct.unpack("<i", buf[pos:pos + 4])
            pos += 4
            data = buf[pos:pos + c]
            pos += c
            parts.append(data if c == neblock else _inner_decode(fmt, data, neblock))
        block = b"".join(parts)
        if len(block) != bsize:
            raise ValueError("blosc block decoded to the wrong size")
        if (flags & 0x1) and typesize > 1:
            block = _unshuffle(block, typesize)
        elif flags & 0x4:
            block = _bitunshuffle(block, typesize)
        out.append(block)
    return b"".join(out)


def blosc_encode(buf, typesize, cname="lz4", clevel=5, shuffle=1, blocksize=0):
    """A valid Blosc-1 frame (shuffle: 0 none, 1 byte, 2 bit).  Streams that do not shrink are stored raw."""
    buf = bytes(buf)
    nbytes = len(buf)
    fmt = _BLOSC_FORMATS[_BLOSC_CNAME_FORMAT[cname]]
    if fmt == "blosclz":
        raise NotImplementedError("blosclz has no encoder in this image")
    typesize = int(typesize) if 0 < int(typesize) <= 255 else 1
    if blocksize <= 0:
        blocksize = max(typesize * _BLOSC_MIN_BUFFERSIZE, 1 << 18)
        blocksize -= blocksize % typesize
    blocksize = max(min(blocksize, nbytes), 1) if nbytes else 1
    flags = (_BLOSC_CNAME_FORMAT[cname] << 5) | (0x1 if shuffle == 1 and typesize > 1 else 0) | (0x4 if shuffle == 2 else 0)
    if nbytes == 0:
        return struct.pack("<BBBBIII", 2, 1, flags, typesize, 0, blocksize, 16)
    nblocks = -(-nbytes // blocksize)
    body, bstarts = [], []
    cur = 16 + 4 * nblocks
    for b in range(nblocks):
        block = buf[b * blocksize:(b + 1) * blocksize]
        leftover = (b == nblocks - 1) and (nbytes % blocksize != 0)
        if flags & 0x1:
            block = _shuffle(block, typesize)
        elif flags & 0x4:
            block = _bitshuffle(block, typesize)
        ns = _nsplits(flags, typesize, blocksize if leftover else len(block), leftover)
        neblock = len(block) // ns
        bstarts.append(cur)
        for k in range(ns):
            raw = block[k * neblock:(k + 1) * neblock]
            comp = _inner_encode(fmt, raw, clevel)
            if len(comp) >= neblock:
                comp = raw
            body.append(struct.pack("<i", len(comp)) + comp)
            cur += 4 + len(comp)
    payload = b"".join(body)
    header = struct.pack("<BBBBIII", 2, 1, flags, typesize, nbytes, blocksize, cur)
    return header + struct.pack(f"<{nblocks}i", *bstarts) + payload
