#!/usr/bin/env python3
"""Write tests/golden/monks_forest_s_rle.hdr: the scanlines of tests/golden/monks_forest_s.hdr (flat RGBE, a byte copy of the reference's
bin/monks_forest_s.hdr) re-encoded with Radiance's "new" run-length scheme -- per scanline the marker 2 2 hi(w) lo(w), then the four
component planes, each as runs (count | 128, value) and literals (count, bytes...).  Pixels are unchanged, so a decoder that handles both
encodings must return identical images for the two files (tests/test_io_formats.py).  Pure data generation; reads nothing outside the repo.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "tests", "golden")


def encode_plane(row):
    """one component plane of a scanline -> RLE bytes.  Runs of >= 3 equal bytes become runs (max 127), the rest literals (max 128)."""
    out = bytearray()
    n, i = len(row), 0
    while i < n:
        j = i
        while j < n and j - i < 127 and row[j] == row[i]:
            j += 1
        if j - i >= 3:
            out += bytes([128 + (j - i), row[i]])
            i = j
            continue
        k = i  # literal: up to the next run of >= 3
        while k < n and k - i < 128:
            if k + 2 < n and row[k] == row[k + 1] == row[k + 2]:
                break
            k += 1
        out += bytes([k - i]) + bytes(row[i:k])
        i = k
    return bytes(out)


def main():
    src = open(os.path.join(GOLDEN, "monks_forest_s.hdr"), "rb").read()
    pos, blank = 0, False
    while True:  # header lines, blank line, resolution line
        e = src.index(b"\n", pos)
        line = src[pos:e]
        pos = e + 1
        if not blank:
            blank = line == b""
            continue
        tok = line.split()
        assert tok[0] == b"-Y" and tok[2] == b"+X"
        h, w = int(tok[1]), int(tok[3])
        break
    header, body = src[:pos], src[pos:]
    assert len(body) == w * h * 4 and 8 <= w < 32768, "source is expected to be flat"
    out = bytearray(header)
    for y in range(h):
        scan = body[y * w * 4:(y + 1) * w * 4]
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            out += encode_plane(scan[c::4])
    dst = os.path.join(GOLDEN, "monks_forest_s_rle.hdr")
    open(dst, "wb").write(bytes(out))
    print(dst, len(out), "bytes (flat:", len(src), ")")


if __name__ == "__main__":
    sys.exit(main())
