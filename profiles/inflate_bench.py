#!/usr/bin/env python3
"""Micro-benchmark of the GPU BGZF inflate (csrc/lps_inflate.hip) on BAM-like payload: records of generated 30x reads (core fields, name, CIGAR,
4-bit packed bases, qualities) cut into 0xff00-byte BGZF members, each deflated by zlib level 6 as htslib does.  Prints output GB/s of the kernel.
    python profiles/inflate_bench.py [contig_len]      (on the GPU box)"""
import os
import struct
import sys
import time
import zlib
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "longphase-s_amd"))
from lps import abi, hip  # noqa: E402
from lps.synth_gpu import SynthGpu  # noqa: E402


def member(chunk):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    z = co.compress(chunk) + co.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(z) + 25) + z + struct.pack("<II", zlib.crc32(chunk), len(chunk)))


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    g = SynthGpu(0, seed=5, contig_len=L, n_snp=L // 1000, coverage=30.0)
    h = g.to_host(); g.close()
    n = h.n_reads
    t0 = time.time()
    parts = []
    co, so, qo = h.cigar_off.astype(np.int64), h.seq_off.astype(np.int64), h.qual_off.astype(np.int64)
    for i in range(n):
        lq = int(h.l_qseq[i]); nc = int(co[i + 1] - co[i]); name = b"read%08d\0" % int(h.name_id[i])
        core = struct.pack("<iiBBHHHiiii", 0, int(h.ref_start[i]), len(name), int(h.mapq[i]), 4681, nc & 0xffff, int(h.flag[i]), lq, -1, -1, 0)
        body = core + name + h.cigar[co[i]:co[i + 1]].tobytes() + h.seq[so[i]:so[i] + (lq + 1) // 2].tobytes() + h.qual[qo[i]:qo[i] + lq].tobytes()
        parts.append(struct.pack("<i", len(body)) + body)
    raw = b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chrS\0" + struct.pack("<i", L) + b"".join(parts)
    chunks = [raw[i:i + 0xff00] for i in range(0, len(raw), 0xff00)]
    with Pool(min(16, os.cpu_count() or 8)) as p:
        members = p.map(member, chunks, chunksize=64)
    z = b"".join(members)
    print(f"payload {len(raw) / 1e6:.1f} MB in {len(chunks)} members, compressed {len(z) / 1e6:.1f} MB ({len(raw) / len(z):.2f}x), built in {time.time() - t0:.1f} s", flush=True)
    data = np.frombuffer(z, dtype=np.uint8)
    with hip.Context(0, abi.default_params()) as ctx:
        best = 1e9
        for rep in range(5):
            nb = ctx.bgzf_load(data)
            assert nb == len(raw)
            tm = ctx.bgzf_timings()
            best = min(best, tm["inflate_ms"])
            print(f"  run {rep}: h2d {tm['h2d_ms']:.1f} ms, inflate {tm['inflate_ms']:.2f} ms = {len(raw) / tm['inflate_ms'] / 1e6:.1f} GB/s of output", flush=True)
        got = ctx.bgzf_read(len(raw) - 4096, 4096).tobytes()
        assert got == raw[-4096:]
        print(f"inflate best {best:.2f} ms = {len(raw) / best / 1e6:.1f} GB/s of output ({len(z) / best / 1e6:.1f} GB/s of input)")


if __name__ == "__main__":
    main()
