"""lps_bgzf_walk_fd is host code (no GPU, no context): the block table of a BGZF file, walked in eight pieces for files of 64 MiB and more (pieces
start where four headers follow one another; a piece must end where the next begins) and serially otherwise.  Against a walk in Python."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "longphase-s_amd", "csrc", "liblps_hip.so")


class Block(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("out_off", C.c_uint64), ("in_len", C.c_uint32), ("out_len", C.c_uint32)]


def make_bgzf(n_blocks, seed, extra_field=False):
    rng = np.random.default_rng(seed); out = bytearray(); want = []; uoff = 0
    for k in range(n_blocks):
        raw = rng.integers(0, 200, int(rng.integers(1, 65_000)), dtype=np.uint8).tobytes() if k % 97 else b""
        c = zlib.compressobj(1, zlib.DEFLATED, -15); comp = c.compress(raw) + c.flush()
        extra = b"XY\x03\x00abc" if (extra_field and k % 5 == 0) else b""          # a second extra subfield before BC: XLEN grows
        xlen = 6 + len(extra); bsize = 12 + xlen + len(comp) + 8
        out += struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, xlen) + extra + struct.pack("<BBHH", 66, 67, 2, bsize - 1)
        want.append((len(out), uoff, len(comp), len(raw)))
        out += comp + struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw)); uoff += len(raw)
    return bytes(out), want, uoff


def walk(lib, path, offset, n):
    fd = os.open(path, os.O_RDONLY)
    try:
        blocks = C.POINTER(Block)(); nb = C.c_int64(0); inflated = C.c_int64(0)
        rc = lib.lps_bgzf_walk_fd(fd, offset, n, C.byref(blocks), C.byref(nb), C.byref(inflated))
        got = [(blocks[i].in_off, blocks[i].out_off, blocks[i].in_len, blocks[i].out_len) for i in range(nb.value)] if rc == 0 else None
        if rc == 0:
            lib.lps_bgzf_blocks_free(blocks)
        return rc, got, inflated.value
    finally:
        os.close(fd)


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(os.path.abspath(LIB))
    L.lps_bgzf_walk_fd.argtypes = [C.c_int, C.c_int64, C.c_int64, C.POINTER(C.POINTER(Block)), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.lps_bgzf_blocks_free.argtypes = [C.POINTER(Block)]
    return L


def test_small_file_serial_walk_and_subrange(lib, tmp_path):
    z, want, utot = make_bgzf(300, 3, extra_field=True)
    p = str(tmp_path / "a.bgzf"); open(p, "wb").write(z)
    rc, got, inflated = walk(lib, p, 0, len(z))
    assert rc == 0 and got == want and inflated == utot
    lo = want[40][0] - 12 - (6 + (7 if 40 % 5 == 0 else 0)); hi = want[200][0] - 12 - (6 + (7 if 200 % 5 == 0 else 0))      # block starts
    rc, got, inflated = walk(lib, p, lo, hi - lo)
    assert rc == 0 and len(got) == 160
    assert got == [(a - lo, b - want[40][1], c, d) for a, b, c, d in want[40:200]] and inflated == want[200][1] - want[40][1]
    bad = bytearray(z); bad[want[150][0] - 12 - 6 - (7 if 150 % 5 == 0 else 0)] ^= 0x55        # the first magic byte of block 150
    open(p, "wb").write(bytes(bad))
    assert walk(lib, p, 0, len(bad))[0] != 0
    open(p, "wb").write(z[:-3])
    assert walk(lib, p, 0, len(z) - 3)[0] != 0                                                # truncated


def test_file_of_64_mib_and_more_is_walked_in_pieces(lib, tmp_path):
    z, want, utot = make_bgzf(2400, 5)
    assert len(z) > (64 << 20)
    p = str(tmp_path / "b.bgzf"); open(p, "wb").write(z)
    rc, got, inflated = walk(lib, p, 0, len(z))
    assert rc == 0 and inflated == utot and got == want
