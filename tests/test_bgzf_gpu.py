"""GPU BGZF inflate (lps_bgzf_load) against zlib: every deflate block type (stored, fixed, dynamic),
several levels/strategies, empty and short blocks, the EOF block, and corrupt input."""
import gzip
import struct
import zlib

import numpy as np
import pytest

import util
from lps import abi, hip

pytestmark = pytest.mark.gpu
DATA = util.os.path.join(util.os.path.dirname(util.os.path.abspath(__file__)), "golden", "data")


def bgzf(payload, block, level, strategy=zlib.Z_DEFAULT_STRATEGY, eof=True):
    out = bytearray()
    def put(chunk):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        comp = c.compress(bytes(chunk)) + c.flush()
        out.extend(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25))
        out.extend(comp + struct.pack("<II", zlib.crc32(bytes(chunk)) & 0xFFFFFFFF, len(chunk)))
    for a in range(0, len(payload), block):
        put(payload[a:a + block])
    if eof:
        put(b"")
    return bytes(out)


def payloads():
    rng = np.random.default_rng(5)
    _, _, rec = util.bam_sections(_bam())
    yield "bam_records", bytes(rec)
    yield "random", rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes()
    yield "zeros", bytes(200_000)
    yield "text", (b"ACGTTGCA" * 7 + b"the quick brown fox\n") * 3000
    yield "long_matches", bytes(rng.integers(0, 4, 70_000, dtype=np.uint8)) * 3


_cache = {}


def _bam():
    if "p" not in _cache:
        import tempfile
        d = tempfile.mkdtemp()
        util.write_bam(util.os.path.join(DATA, "tiny_indel.sam.gz"), d + "/t.bam")
        _cache["p"] = d + "/t.bam"
    return _cache["p"]


@pytest.mark.parametrize("level,strategy,block", [(6, zlib.Z_DEFAULT_STRATEGY, 0xff00), (1, zlib.Z_DEFAULT_STRATEGY, 0xff00), (9, zlib.Z_DEFAULT_STRATEGY, 65280),
                                                    (0, zlib.Z_DEFAULT_STRATEGY, 40000), (6, zlib.Z_FIXED, 30000), (6, zlib.Z_RLE, 0xff00), (6, zlib.Z_HUFFMAN_ONLY, 0xff00),
                                                    (6, zlib.Z_DEFAULT_STRATEGY, 7), (6, zlib.Z_DEFAULT_STRATEGY, 1000)])
def test_inflate_matches_zlib(level, strategy, block):
    with hip.Context(0, abi.default_params()) as ctx:
        for name, data in payloads():
            if block < 100:
                data = data[:5000]
            z = bgzf(data, block, level, strategy)
            assert gzip.decompress(z) == data
            n = ctx.bgzf_load(z)
            assert n == len(data), name
            got = ctx.bgzf_read(0, n).tobytes()
            assert got == data, (name, level, strategy, block)


def test_corrupt_streams_are_rejected_and_ctx_survives():
    data = bytes(np.random.default_rng(1).integers(0, 64, 150_000, dtype=np.uint8))
    z = bytearray(bgzf(data, 0xff00, 6))
    with hip.Context(0, abi.default_params()) as ctx:
        bad = bytearray(z); bad[40] ^= 0x5a; bad[41] ^= 0xa5; bad[300] ^= 0xff
        with pytest.raises(hip.LpsError):
            ctx.bgzf_load(bytes(bad))
        isz = bytearray(z); isz[-28 - 4] ^= 1                       # ISIZE of the last data block off by one
        with pytest.raises(hip.LpsError):
            ctx.bgzf_load(bytes(isz))
        with pytest.raises(hip.LpsError):
            ctx.bgzf_load(bytes(z[:-5]))                            # truncated file
        with pytest.raises(hip.LpsError):
            ctx.bgzf_load(b"not a bgzf file, just some text that is long enough")
        assert ctx.bgzf_load(bytes(z)) == len(data)                 # still usable
        assert ctx.bgzf_read(1000, 5000).tobytes() == data[1000:6000]
        print("timings", ctx.bgzf_timings())


def _bam_header_end(ctx, n_inflated):
    head = ctx.bgzf_read(0, min(n_inflated, 1 << 20)).tobytes()
    assert head[:4] == b"BAM\1"
    lt = struct.unpack_from("<i", head, 4)[0]
    p = 8 + lt
    n_ref = struct.unpack_from("<i", head, p)[0]; p += 4
    names = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", head, p)[0]; names.append(head[p + 4:p + 4 + ln - 1].decode()); p += 4 + ln + 4
    return p, names


@pytest.mark.parametrize("name", ["snp_ont", "indels", "supp_overlap"])
def test_phase_from_gpu_inflated_bam(name, tmp_path):
    """whole input side on the GPU: BGZF inflate -> record discovery -> core decode -> phase; equals the host-decoded path and the reference golden"""
    import fixtures
    kw, cli, over = fixtures.PHASE_FIXTURES[name]
    s, V, R = util.make_case(kw)
    d = str(tmp_path)
    s.write_sam(d + "/r.sam")
    util.add_stale_tags(d + "/r.sam", d + "/t.sam")                  # optional fields after the qualities
    n_rec = util.write_bam(d + "/t.sam", d + "/t.bam", block=0xff00)
    assert n_rec == R.n_reads
    raw = np.fromfile(d + "/t.bam", dtype=np.uint8)
    with hip.Context(0, abi.default_params(**over)) as ctx:
        want = ctx.phase(V, s.ref, R)
        n = ctx.bgzf_load(raw)
        first, refs = _bam_header_end(ctx, n)
        tid = ctx.bam_scan(first, len(refs))
        assert tid.size == n_rec and (tid == 0).all()
        names = ctx.bam_names(0, n_rec)
        text = [l.split("\t")[0].encode() for l in open(d + "/t.sam") if not l.startswith("@")]
        assert names == text
        rank = {nm: i for i, nm in enumerate(sorted(set(names)))}
        ctx.load_resident(V, s.ref, 0, n_rec, [rank[x] for x in names])
        got = ctx.run_phase()
        assert np.array_equal(got.phase_set, want.phase_set) and np.array_equal(got.gt, want.gt)
        gpos, gps, ggt = util.load_golden_phase(name)
        util.assert_phase_equal(got.phase_set, got.gt, gps, ggt, name + " GPU-inflated BAM vs reference golden")
        # a sub-range push (second half of the records) also works and mixing with another push kind is refused
        ctx.load_resident(V, s.ref, n_rec // 2, n_rec - n_rec // 2, [rank[x] for x in names[n_rec // 2:]])
        ctx.run_phase()
        with pytest.raises(hip.LpsError, match="mixed"):
            ctx._check(ctx.L.lps_push_reads(ctx.h, hip.C.byref(R.c)), "lps_push_reads")
    s.close()


def _check_blocks(z):
    """structural checks of a BGZF byte string: member headers, BSIZE chain, ISIZE <= 0xff00"""
    p, n = 0, 0
    while p < len(z):
        assert z[p:p + 4] == b"\x1f\x8b\x08\x04" and z[p + 12:p + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", z, p + 16)[0] + 1
        isize = struct.unpack_from("<I", z, p + bsize - 4)[0]
        assert isize <= 0xff00 and bsize <= 65536
        p += bsize; n += 1
    assert p == len(z)
    return n


@pytest.mark.parametrize("size", [1, 2, 300, 0xff00 - 1, 0xff00, 0xff00 + 1, 200_000])
def test_gpu_deflate_round_trips(size):
    """lps_bgzf_deflate: GPU-written BGZF blocks inflate (zlib, incl. its CRC32/ISIZE checks) to the resident bytes; ratio close to zlib's Huffman-only"""
    with hip.Context(0, abi.default_params()) as ctx:
        for name, data in payloads():
            data = (data * (size // len(data) + 1))[:size]
            assert ctx.bgzf_load(bgzf(data, 0xff00, 1)) == len(data)
            z, ms = ctx.bgzf_deflate(0, len(data))
            assert _check_blocks(z) == (len(data) + 0xff00 - 1) // 0xff00
            assert gzip.decompress(z) == data, (name, size)            # gzip verifies CRC32 and ISIZE of every member
            if size >= 0xff00:
                want = len(bgzf(data, 0xff00, 6, zlib.Z_HUFFMAN_ONLY, eof=False))
                assert len(z) <= want * 1.03 + 64, (name, len(z), want)
                print(name, size, "gpu", len(z), "zlib huffman-only", want, "zlib level 6", len(bgzf(data, 0xff00, 6, eof=False)), "kernel ms", round(ms, 3))
            # a sub-range of the stream, and the GPU inflate reads what the GPU deflate wrote
            if size > 5000:
                z2, _ = ctx.bgzf_deflate(1234, size - 2345)
                assert gzip.decompress(z2) == data[1234:size - 1111]
                assert ctx.bgzf_load(z) == len(data) and ctx.bgzf_read(0, len(data)).tobytes() == data


def test_fuzzed_files_never_hang_or_pass_silently():
    """random byte flips / garbage inside a valid BGZF file: the loader must return (error or - if the flip hit nothing checked - identical data),
    never hang, crash or hand back bytes that differ from what zlib makes of the same file"""
    rng = np.random.default_rng(77)
    _, _, rec = util.bam_sections(_bam())
    base = bgzf(bytes(rec[:400_000]), 0xff00, 6)
    with hip.Context(0, abi.default_params()) as ctx:
        n_err = 0
        for trial in range(60):
            z = bytearray(base)
            kind = trial % 3
            if kind == 0:                                           # a few random byte flips
                for _ in range(int(rng.integers(1, 6))):
                    z[int(rng.integers(0, len(z) - 28))] ^= int(rng.integers(1, 256))
            elif kind == 1:                                         # a run of zeros / ones inside a block body
                p = int(rng.integers(30, len(z) - 4000)); z[p:p + int(rng.integers(8, 2000))] = bytes([0x00 if trial % 2 else 0xff]) * 1
            else:                                                   # endless-empty-block pattern: BFINAL=0 BTYPE=01 EOB repeated
                p = int(rng.integers(30, len(z) - 6000)); z[p:p + 4000] = bytes([0x02, 0x08, 0x20, 0x80, 0x00]) * 800
            try:
                want = gzip.decompress(bytes(z))
            except Exception:
                want = None
            try:
                n = ctx.bgzf_load(bytes(z))
                got = ctx.bgzf_read(0, n).tobytes()
            except hip.LpsError:
                n_err += 1
                continue
            assert want is not None and got == want, trial
        assert n_err >= 40
        assert ctx.bgzf_load(base) == 400_000                       # the context survived all of it


def test_deflate_of_host_bytes_fetched_in_pieces():
    """lps_bgzf_deflate_host (bytes a host-side splice produced) + lps_bgzf_deflate_fetch_range into page-locked memory from lps_host_alloc, in
    pieces: the concatenated members inflate (zlib) to the input; ranges outside the result are refused."""
    import ctypes as C
    rng = np.random.default_rng(9)
    _, _, rec = util.bam_sections(_bam())
    data = bytes(rec[:300_000]) + rng.integers(0, 256, 100_001, dtype=np.uint8).tobytes()
    with hip.Context(0, abi.default_params()) as ctx:
        L = ctx.L
        nb = C.c_int64(0)
        src = np.frombuffer(data, np.uint8)
        assert L.lps_bgzf_deflate_host(ctx.h, src.ctypes.data, src.size, C.byref(nb)) == 0 and nb.value > 0
        piece = 70_000
        pin = L.lps_host_alloc(piece)
        assert pin
        out = bytearray()
        for off in range(0, nb.value, piece):
            n = min(piece, nb.value - off)
            assert L.lps_bgzf_deflate_fetch_range(ctx.h, off, n, pin) == 0
            out += C.string_at(pin, n)
        assert L.lps_bgzf_deflate_fetch_range(ctx.h, nb.value - 10, 11, pin) != 0
        L.lps_host_free(pin)
        assert gzip.decompress(bytes(out)) == data
        assert L.lps_bgzf_deflate_host(ctx.h, None, 0, C.byref(nb)) == 0 and nb.value == 0


def test_file_of_several_upload_pieces_read_with_pread(tmp_path):
    """lps_bgzf_load_fd on a 160 MB file: the header walk runs in eight pieces (seeds found where four headers follow one another), the upload goes in
    64-MiB pieces beside the inflate kernel, whose wavefronts wait for their own bytes (the watermark); then a sub-range of the same file (what a .bai
    gives for a contig group), the memory entry point on the same bytes, and a file whose chain breaks in the middle (serial walk, the error names it)."""
    import os
    rng = np.random.default_rng(11)
    parts, blocks, zs, at = [], [], [], 0
    for k in range(2700):                                             # ~64 KB of packed-base-like bytes per block: barely compressible, 63 KB per block
        raw = rng.integers(0, 200, int(rng.integers(60_000, 65_280)), dtype=np.uint8).tobytes()
        z = bgzf(raw, 1 << 20, 1, eof=False)
        parts.append(raw); blocks.append((at, len(z))); at += len(z); zs.append(z)
    z = b"".join(zs) + bgzf(b"", 1, 1)[:28]
    data = b"".join(parts)
    assert len(z) > 150_000_000                                       # three upload pieces of 64 MiB
    path = str(tmp_path / "big.bgzf"); open(path, "wb").write(z)
    fd = os.open(path, os.O_RDONLY)
    try:
        with hip.Context(0, abi.default_params()) as ctx:
            assert ctx.bgzf_load_fd(fd, 0, len(z)) == len(data)
            for a in (0, 70_000_000, len(data) - 9_000):
                assert ctx.bgzf_read(a, 9_000).tobytes() == data[a:a + 9_000]
            lo, hi = blocks[400][0], blocks[2100][0]                   # a run of whole blocks from the middle of the file
            ulo = sum(len(x) for x in parts[:400]); uhi = sum(len(x) for x in parts[:2100])
            assert ctx.bgzf_load_fd(fd, lo, hi - lo) == uhi - ulo
            assert ctx.bgzf_read(0, 5_000).tobytes() == data[ulo:ulo + 5_000] and ctx.bgzf_read(uhi - ulo - 5_000, 5_000).tobytes() == data[uhi - 5_000:uhi]
            assert ctx.bgzf_load(z) == len(data)                       # from memory: same walk, same kernel
            assert ctx.bgzf_read(123_456_789, 4_096).tobytes() == data[123_456_789:123_456_789 + 4_096]
            bad = bytearray(z); bad[blocks[1500][0] + 1] ^= 0xff         # the second magic byte of a block in the middle
            bpath = str(tmp_path / "bad.bgzf"); open(bpath, "wb").write(bytes(bad))
            bfd = os.open(bpath, os.O_RDONLY)
            try:
                with pytest.raises(hip.LpsError, match="not a BGZF block header"):
                    ctx.bgzf_load_fd(bfd, 0, len(bad))
            finally:
                os.close(bfd)
            assert ctx.bgzf_load_fd(fd, 0, len(z)) == len(data)         # the context is still usable
    finally:
        os.close(fd)


def test_slow_source_makes_the_inflate_kernel_time_out_and_the_load_still_succeeds(tmp_path, monkeypatch):
    """The inflate kernel is launched beside the upload and its wavefronts wait a bounded time for their bytes.  A source slower than that bound
    (network storage, a file that fell out of the page cache) used to fail the load; now the members are inflated again after the upload.  The hooks
    make it happen on a 160 MB file: the launch waits for 64 members only, every 64-MiB piece of the upload is held back by 300 ms, a wavefront
    gives up after 20 ms."""
    import os
    rng = np.random.default_rng(12)
    parts, zs = [], []
    for k in range(2700):
        raw = rng.integers(0, 200, int(rng.integers(60_000, 65_280)), dtype=np.uint8).tobytes()
        parts.append(raw); zs.append(bgzf(raw, 1 << 20, 1, eof=False))
    z = b"".join(zs) + bgzf(b"", 1, 1)[:28]
    data = b"".join(parts)
    assert len(z) > 150_000_000
    path = str(tmp_path / "slow.bgzf"); open(path, "wb").write(z)
    fd = os.open(path, os.O_RDONLY)
    try:
        with hip.Context(0, abi.default_params()) as ctx:
            monkeypatch.setenv("LPS_BGZF_TEST_THROTTLE_MS", "300"); monkeypatch.setenv("LPS_BGZF_TEST_FIRST_ROUND", "64"); monkeypatch.setenv("LPS_BGZF_TEST_TIMEOUT_MS", "20")
            assert ctx.bgzf_load_fd(fd, 0, len(z)) == len(data)
            assert ctx.L.lps_bgzf_retried(ctx.h) == 1                    # the path under test did run
            for a in (0, 70_000_000, 140_000_000, len(data) - 9_000):
                assert ctx.bgzf_read(a, 9_000).tobytes() == data[a:a + 9_000]
            monkeypatch.delenv("LPS_BGZF_TEST_THROTTLE_MS"); monkeypatch.delenv("LPS_BGZF_TEST_FIRST_ROUND"); monkeypatch.delenv("LPS_BGZF_TEST_TIMEOUT_MS")
            assert ctx.bgzf_load_fd(fd, 0, len(z)) == len(data) and ctx.L.lps_bgzf_retried(ctx.h) == 0
            assert ctx.bgzf_read(123_456_789, 4_096).tobytes() == data[123_456_789:123_456_789 + 4_096]
    finally:
        os.close(fd)
