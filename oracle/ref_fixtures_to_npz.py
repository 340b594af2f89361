#!/usr/bin/env python3
"""Test infrastructure: oracle/_ref/ref_fixtures.bin (written by oracle/_ref/ref_fixture_driver, which runs the reference's
tests/random_matrix.hpp / random_vector.hpp and the host loop of tests/spmv.cpp:28-32) -> tests/golden/ref_fixtures.npz.
usage: python oracle/ref_fixtures_to_npz.py [in.bin [out.npz]]"""
import os
import struct
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "_ref", "ref_fixtures.bin")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(here), "tests", "golden", "ref_fixtures.npz")
out = {}
with open(src, "rb") as f:
    while True:
        head = f.read(48)
        if len(head) < 48:
            break
        name = head[:32].split(b"\0")[0].decode()
        ty = head[32:40].split(b"\0")[0].decode()
        (n,) = struct.unpack("<Q", head[40:48])
        out[name] = np.frombuffer(f.read(8 * n), dtype={"i8": np.int64, "f8": np.float64}[ty]).copy()
np.savez_compressed(dst, **out)
print("%s: %d arrays, %d bytes" % (dst, len(out), os.path.getsize(dst)))
