"""TEST INFRASTRUCTURE: convert the raw GLD1 dumps written by oracle/ref_dump.cpp
(reference outputs) into tests/golden/<case>.npz.  Usage:
    python oracle/pack_fixtures.py /tmp/golden_raw tests/golden
"""
import struct
import sys
from pathlib import Path

import numpy as np


def read_gld(path):
    raw = Path(path).read_bytes()
    assert raw[:4] == b"GLD1", path
    pos, out = 4, {}
    while pos < len(raw):
        (nl,) = struct.unpack_from("<i", raw, pos); pos += 4
        name = raw[pos:pos + nl].decode(); pos += nl
        ty = chr(raw[pos]); pos += 1
        (n,) = struct.unpack_from("<q", raw, pos); pos += 8
        dt = {"d": np.float64, "i": np.int32}[ty]
        out[name] = np.frombuffer(raw, dtype=dt, count=n, offset=pos).copy()
        pos += n * np.dtype(dt).itemsize
    return out


def main(src, dst):
    dst = Path(dst); dst.mkdir(parents=True, exist_ok=True)
    total = 0
    for f in sorted(Path(src).glob("*.gld")):
        arrs = read_gld(f)
        o = dst / (f.stem + ".npz")
        np.savez_compressed(o, **arrs)
        total += o.stat().st_size
    print("packed", len(list(dst.glob('*.npz'))), "cases,", total / 1e6, "MB")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
