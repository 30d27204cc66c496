"""On-disk formats after the path: the float32 frame as PFM and as OpenEXR
(uncompressed scanline, FLOAT B/G/R - what the reference's imageio call writes, application.py:1117),
both written without imageio.  Round trips are bit-exact; the EXR layout is checked against the
format's published structure (magic, version, attribute table, offset table, per-scanline B|G|R planes)."""
import struct

import numpy as np

from raytracer_amd.scripts.application import read_exr, read_pfm, write_exr, write_pfm


def _img(h=9, w=13, seed=3):
    r = np.random.default_rng(seed)
    img = r.normal(0, 4, (h, w, 3)).astype(np.float32)
    img[0, 0] = [np.inf, -0.0, 1e-40]                   # inf, signed zero, subnormal survive
    return img


def test_pfm_round_trip(tmp_path):
    img = _img()
    write_pfm(tmp_path / "a.pfm", img)
    assert np.array_equal(read_pfm(tmp_path / "a.pfm").view(np.uint32), img.view(np.uint32))


def test_exr_round_trip_and_layout(tmp_path):
    img = _img()
    h, w, _ = img.shape
    write_exr(tmp_path / "a.exr", img)
    assert np.array_equal(read_exr(tmp_path / "a.exr").view(np.uint32), img.view(np.uint32))
    raw = (tmp_path / "a.exr").read_bytes()
    assert raw[:4] == bytes([0x76, 0x2F, 0x31, 0x01]) and struct.unpack_from("<i", raw, 4)[0] == 2   # magic, version 2, no flags
    assert raw[8:24] == b"channels\0chlist\0"
    # three 16-byte channel records after each one-letter name: FLOAT (2), pLinear 0, sampling 1 x 1
    p = 24 + 4
    for name in (b"B", b"G", b"R"):
        assert raw[p:p + 2] == name + b"\0"
        assert struct.unpack_from("<iB3xii", raw, p + 2) == (2, 0, 1, 1)
        p += 18
    # header ends with a null byte, then h offsets, then the scanlines in increasing y
    table = raw.index(b"screenWindowWidth\0float\0") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1
    offs = struct.unpack_from(f"<{h}Q", raw, table)
    assert offs[0] == table + 8 * h and all(b - a == 8 + 12 * w for a, b in zip(offs, offs[1:]))
    assert len(raw) == offs[-1] + 8 + 12 * w
    y, size = struct.unpack_from("<ii", raw, offs[2])
    assert (y, size) == (2, 12 * w)
    planes = np.frombuffer(raw, dtype="<f4", count=3 * w, offset=offs[2] + 8).reshape(3, w)
    assert np.array_equal(planes[0].view(np.uint32), img[2, :, 2].view(np.uint32))     # B plane first
    assert np.array_equal(planes[2].view(np.uint32), img[2, :, 0].view(np.uint32))     # R plane last
