"""Test infrastructure: the file formats at the reference program's boundary -- a minimal
baseline TIFF writer (uncompressed, one strip, 8/16-bit grey; what TIFFReadScanline needs, GMA.c:246-316) and the
.GMA container (int32 rows, int32 cols, row-major payload; GMA.c:168-244, :319-424)."""
import struct

import numpy as np

_GMA_TYPES = {"x": np.float64, "y": np.float64, "vx": np.float32, "vy": np.float32, "ex": np.float32, "ey": np.float32,
              "qual": np.float32, "flagcp": np.uint8}


def write_tiff(path, img):
    img = np.ascontiguousarray(img)
    assert img.dtype in (np.uint8, np.uint16) and img.ndim == 2
    h, w = img.shape
    bits = img.dtype.itemsize * 8
    data = img.astype("<u%d" % img.dtype.itemsize).tobytes()
    tags = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, 8),
            (277, 3, 1, 1), (278, 4, 1, h), (279, 4, 1, len(data)), (284, 3, 1, 1)]
    ifd_off = 8 + len(data) + (len(data) & 1)
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, ifd_off))
        f.write(data)
        if len(data) & 1:
            f.write(b"\0")
        f.write(struct.pack("<H", len(tags)))
        for tag, typ, cnt, val in tags:
            f.write(struct.pack("<HHI", tag, typ, cnt))
            f.write(struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
        f.write(struct.pack("<I", 0))


def write_gma(path, a):
    a = np.ascontiguousarray(a)
    assert a.ndim == 2
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", a.shape[0], a.shape[1]))
        f.write(a.tobytes())


def read_gma(path, dtype):
    with open(path, "rb") as f:
        r, c = struct.unpack("<ii", f.read(8))
        return np.frombuffer(f.read(), dtype=dtype, count=r * c).reshape(r, c).copy()


def read_vmap(outdir, t0, t1):
    """the eight .GMA outputs + meta.txt of one run (MIMC_main.c:100-109, :428-447)"""
    out = {k: read_gma(f"{outdir}/vmap_{t0}_{t1}_{k}.GMA", dt) for k, dt in _GMA_TYPES.items()}
    meta = {}
    with open(f"{outdir}/vmap_{t0}_{t1}_meta.txt") as f:
        for line in f:
            k, _, v = line.rstrip("\n").partition("=")
            meta[k] = v
    out["meta"] = meta
    return out
