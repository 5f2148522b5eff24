"""Shared helpers of the parity tests."""
import gzip
import json
import os

import numpy as np

import fixtures
from lps import abi
from lps.synth import Synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INDEX = json.load(open(os.path.join(GOLDEN, "index.json")))


def make_case(kw):
    s = Synth(**kw)
    V = abi.Variants(s.var_pos, s.var_ref, s.var_alt)
    R = abi.Reads.from_synth(s)
    return s, V, R


def make_extra_case(name):
    """A `phase --sv-file --mod-file` fixture: generator output, tables, lps_extra_variants (MOD read lists as stored with the golden results),
    params and the golden arrays."""
    kw, mod_kw, use_sv, cli, over, xover = fixtures.EXTRA_FIXTURES[name]
    s, V, R = make_case(kw)
    assert fixtures.input_digest(s) == INDEX["extra:" + name]["digest"], "generator drift: golden inputs differ"
    g = np.load(os.path.join(GOLDEN, f"phase_extra_{name}.npz"))
    assert np.array_equal(g["var_pos"], V.pos)
    if use_sv:
        assert np.array_equal(g["sv_pos"], s.sv_pos) and np.array_equal(g["sv_len"], s.sv_len)
    off = g["mod_off"].astype(np.int64)
    rows = [[(int(n), bool(f & 1), bool(f & 2)) for n, f in zip(g["mod_name"][off[i]:off[i + 1]], g["mod_flag"][off[i]:off[i + 1]])] for i in range(len(g["mod_pos"]))]
    X = abi.ExtraVariants(g["sv_pos"], g["sv_len"], g["mod_pos"], rows, **xover)
    return s, V, X, R, abi.default_params(**over), g


def load_golden_phase(name):
    z = np.load(os.path.join(GOLDEN, f"phase_{name}.npz"))
    return z["var_pos"], z["phase_set"], z["gt"]


def assert_phase_equal(ps_a, gt_a, ps_b, gt_b, what=""):
    """Bit-exact comparison of phased genotypes and block ids (gt only where phased)."""
    ps_a = np.asarray(ps_a); ps_b = np.asarray(ps_b)
    bad = np.nonzero(ps_a != ps_b)[0]
    assert bad.size == 0, f"{what}: {bad.size} PS mismatches, first at variant {bad[:5]}: {ps_a[bad[:5]]} vs {ps_b[bad[:5]]}"
    m = ps_a != 0
    badg = np.nonzero(np.asarray(gt_a)[m] != np.asarray(gt_b)[m])[0]
    assert badg.size == 0, f"{what}: {badg.size} GT mismatches"


def assert_stages_equal(ctx, d, what=""):
    """Every stage dump of the last lps_phase_chromosome against the oracle's dumps `d` (bit-exact, the fp32 edge matrix as uint32)."""
    cnt, var, al, q = ctx.dump_observations()
    n = d.c.n_obs
    assert np.array_equal(cnt, d.obs_count), what + ": per-read observation counts differ"
    assert np.array_equal(var, d.obs_var[:n]) and np.array_equal(al, d.obs_allele[:n]), what + ": observations differ"
    assert np.array_equal(q.astype(np.int32), d.obs_quality[:n].astype(np.int32)), what + ": observation qualities differ"
    cp, cf = ctx.dump_clips()
    o = np.lexsort((d.clip_fb[:d.c.n_clips], d.clip_pos[:d.c.n_clips]))
    assert np.array_equal(cp, d.clip_pos[:d.c.n_clips][o]) and np.array_equal(cf, d.clip_fb[:d.c.n_clips][o]), what + ": clips differ"
    cs, ce, dele = ctx.dump_cnv()
    assert list(cs) == list(d.cnv_start()) and list(ce) == list(d.cnv_end()), what + ": CNV intervals differ"
    assert np.array_equal(dele, d.aln_deleted), what + ": overlap-filter deletions differ"
    nodes, edge = ctx.dump_graph()
    N = d.c.n_nodes
    assert np.array_equal(nodes, d.node_var[:N]), what + ": graph nodes differ"
    assert np.array_equal(edge.view(np.uint32), d.edge[:N].view(np.uint32)), what + ": edge matrix differs bitwise"
    hp, blk = ctx.dump_votes()
    assert np.array_equal(hp, d.node_hp[:N]) and np.array_equal(blk, d.node_block[:N]), what + ": vote scan differs"


_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
_OPS = {c: i for i, c in enumerate("MIDNSHP=XB")}


def parse_sam(path):
    """SAM text -> abi.Reads (+ names).  name_id = rank of the name under byte-wise ordering."""
    import re
    op = gzip.open if path.endswith(".gz") else open
    rs, fl, mq, lq, names, cig, seq, qual = [], [], [], [], [], [], [], []
    cig_off, seq_off, qual_off = [0], [0], [0]
    for line in op(path, "rt"):
        if line.startswith("@"):
            continue
        f = line.rstrip("\n").split("\t")
        names.append(f[0].encode()); fl.append(int(f[1])); rs.append(int(f[3]) - 1); mq.append(int(f[4]))
        for ln, o in re.findall(r"(\d+)([MIDNSHP=XB])", f[5]):
            cig.append(int(ln) << 4 | _OPS[o])
        cig_off.append(len(cig))
        s = "" if f[9] == "*" else f[9]
        lq.append(len(s))
        packed = bytearray((len(s) + 1) // 2)
        for j, c in enumerate(s):
            packed[j >> 1] |= _NT16.get(c, 15) << (4 if (j & 1) == 0 else 0)
        seq.append(bytes(packed)); seq_off.append(seq_off[-1] + len(packed))
        q = bytes((ord(c) - 33) for c in f[10]) if f[10] != "*" else bytes([255]) * len(s)
        qual.append(q); qual_off.append(qual_off[-1] + len(q))
    uniq = sorted(set(names))
    rank = {n: i for i, n in enumerate(uniq)}
    R = abi.Reads(ref_start=rs, flag=fl, mapq=mq, l_qseq=lq, name_id=[rank[n] for n in names],
                  cigar_off=cig_off, cigar=np.array(cig, dtype=np.uint32), seq_off=seq_off,
                  seq=np.frombuffer(b"".join(seq), dtype=np.uint8), qual_off=qual_off,
                  qual=np.frombuffer(b"".join(qual), dtype=np.uint8))
    return R, names


def parse_vcf_variants(path, indels=False):
    """SnpParser::SnpParser (src/phase/ParsingBam.cpp:222-359) row selection: het, bi-allelic, SNP (+indels)."""
    pos, ref, alt = [], [], []
    for line in open(path):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        gt = f[9].split(":")[f[8].split(":").index("GT")]
        if gt not in ("0/1", "1/0", "0|1", "1|0") or "," in f[4] or f[4].startswith("<"):
            continue
        is_snp = len(f[3]) == 1 and len(f[4]) == 1
        if not is_snp and not indels:
            continue
        pos.append(int(f[1]) - 1); ref.append(f[3]); alt.append(f[4])
    return abi.Variants(pos, ref, alt)


def parse_fasta(path):
    seq = []
    for line in open(path):
        if not line.startswith(">"):
            seq.append(line.strip())
    return np.frombuffer("".join(seq).encode(), dtype=np.uint8)


def load_golden_haplotag(name):
    z = np.load(os.path.join(GOLDEN, f"haplotag_{name}.npz"))
    V = abi.Variants(z["pos"], [str(x) for x in z["ref"]], [str(x) for x in z["alt"]], hp1_is_alt=z["hp1_is_alt"],
                     phase_set=z["phase_set"])
    return V, z["hp"], z["ps"], z["pq"]


def assert_tags_equal(out, hp, ps, pq, what=""):
    """HP/PS/PQ as written to the BAM: PS and PQ only exist on tagged reads."""
    bad = np.nonzero(out.hp != hp)[0]
    assert bad.size == 0, f"{what}: {bad.size} HP mismatches, first reads {bad[:5]}: {out.hp[bad[:5]]} vs {hp[bad[:5]]}"
    m = hp != 0
    assert np.array_equal(out.ps[m], ps[m]), f"{what}: PS mismatch"
    assert np.array_equal(out.pq[m], pq[m]), f"{what}: PQ mismatch"


def load_golden_somatic(name):
    z = np.load(os.path.join(GOLDEN, f"somatic_tag_{name}.npz"))
    V = abi.Variants(z["pos"], [str(x) for x in z["ref"]], [str(x) for x in z["alt"]], hp1_is_alt=z["hp1_is_alt"],
                     phase_set=z["phase_set"], somatic_role=z["somatic_role"], derive_hp=z["derive_hp"], tumor_kind=z["tumor_kind"])
    V.log_pos, V.log_val = z["log_pos"], z["log_val"]
    V.dense_thr, V.dense_pos, V.dense_cnt, V.rd_name, V.rd_hp = z["dense_thr"], z["dense_pos"], z["dense_cnt"], z["rd_name"], z["rd_hp"]
    return V, z["hp"], z["ps"], z["pq"]


def make_normal_reads(name):
    genome, nkw, tkw, cli, over = fixtures.SOMATIC_FIXTURES[name]
    N = Synth(**dict(genome, **nkw))
    return N, abi.Reads.from_synth(N)


# 1-based field numbers of the reference's _somatic_var.out rows (SomaticVarCaller.cpp:1852-1917)
LOG = dict(tumAltCount=6, readCount=7, norAltCount=8, norVAF=18, tumVAF=19, norMpqVAF=20, norDepth=24, tumDepth=25, norDel=27, tumDel=28,
           norMpqReadRatio=31, H1=35, H2=36, H1_1=37, H2_1=38, H3=39, norH1=45, norH2=46, norNonDelAF=63)


def make_tumor_reads(name):
    genome, nkw, tkw, cli, over = fixtures.SOMATIC_FIXTURES[name]
    T = Synth(**dict(genome, **tkw))
    return T, abi.Reads.from_synth(T)


def assert_somatic_tags_equal(out, hp, ps, pq, what=""):
    """HP:Z code, PS (absent = -1) and PQ as written to the tagged tumor BAM."""
    bad = np.nonzero(out.hp != hp)[0]
    assert bad.size == 0, f"{what}: {bad.size} HP mismatches, first reads {bad[:5]}: {out.hp[bad[:5]]} vs {hp[bad[:5]]}"
    m = hp != 0
    assert np.array_equal(out.ps[m], ps[m]), f"{what}: PS mismatch"
    assert np.array_equal(out.pq[m], pq[m]), f"{what}: PQ mismatch"


def _reg2bin(beg, end):
    """SAM spec 5.3 (UCSC binning, min_shift 14, depth 5)."""
    end -= 1
    for sh, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return off + (beg >> sh)
    return 0


def _aux_bytes(tok):
    """One SAM optional field -> BAM bytes, integers in the smallest type (what htslib's SAM parser picks)."""
    import struct
    tag, typ, val = tok.split(":", 2)
    t = tag.encode()
    if typ == "i":
        v = int(val)
        if v < 0:
            code, fmt = ("c", "<b") if v >= -128 else ("s", "<h") if v >= -32768 else ("i", "<i")
        else:
            code, fmt = ("C", "<B") if v <= 255 else ("S", "<H") if v <= 65535 else ("I", "<I")
        return t + code.encode() + struct.pack(fmt, v)
    if typ == "A":
        return t + b"A" + val.encode()
    if typ == "f":
        return t + b"f" + struct.pack("<f", float(val))
    if typ in "ZH":
        return t + typ.encode() + val.encode() + b"\0"
    if typ == "B":
        sub, *xs = val.split(",")
        fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
        conv = float if sub == "f" else int
        return t + b"B" + sub.encode() + struct.pack("<i", len(xs)) + struct.pack("<%d%s" % (len(xs), fmt), *[conv(x) for x in xs])
    raise ValueError(tok)


def add_stale_tags(sam_in, sam_out):
    """Copy a SAM, giving the records optional fields (other tags, and HP/PS/PQ left over from an earlier run) in a fixed
    pattern, so that the tag strip/append rules of HaplotagProcess.cpp:337-361 are exercised."""
    op = gzip.open if sam_in.endswith(".gz") else open
    extra = [
        ["NM:i:17", "HP:i:2", "PS:i:123456", "PQ:i:7"],
        ["XA:Z:keep me", "PQ:i:300", "ZB:B:s,1,-2,3", "HP:i:1"],
        [],
        ["PS:i:70000", "tp:A:P", "de:f:0.0125", "PS:i:5"],
        ["HP:i:-3", "s1:i:-40000", "HP:i:1"],
    ]
    with op(sam_in, "rt") as fi, open(sam_out, "w") as fo:
        i = 0
        for line in fi:
            if line.startswith("@"):
                fo.write(line); continue
            f = line.rstrip("\n").split("\t")[:11] + extra[i % len(extra)]
            fo.write("\t".join(f) + "\n"); i += 1
    return i


def write_bam(sam_path, bam_path, block=60000):
    """SAM(.gz) text -> BAM (BGZF, SAM spec §4): test-side writer so the CLI's own BGZF/BAM reader is exercised
    on files that never passed through htslib.  No index.  Bin and optional-field encodings follow htslib's SAM parser, so the
    inflated byte stream equals what `test_view -b` makes from the same text."""
    import re, struct, zlib
    op = gzip.open if sam_path.endswith(".gz") else open
    header, recs, refs = [], [], []
    for line in op(sam_path, "rt"):
        if line.startswith("@"):
            header.append(line)
            if line.startswith("@SQ"):
                d = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t")[1:])
                refs.append((d["SN"], int(d["LN"])))
            continue
        recs.append(line.rstrip("\n").split("\t"))
    tid = {n: i for i, (n, _) in enumerate(refs)}
    text = "".join(header).encode()
    out = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs)))
    for n, ln in refs:
        out += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", ln)
    for f in recs:
        name = f[0].encode() + b"\0"
        ops = re.findall(r"(\d+)([MIDNSHP=XB])", f[5])
        cig = [(int(ln) << 4) | _OPS[o] for ln, o in ops]
        s = "" if f[9] == "*" else f[9]
        packed = bytearray((len(s) + 1) // 2)
        for j, c in enumerate(s):
            packed[j >> 1] |= _NT16.get(c, 15) << (4 if (j & 1) == 0 else 0)
        q = bytes((ord(c) - 33) for c in f[10]) if f[10] != "*" else bytes([255]) * len(s)
        pos = int(f[3]) - 1
        rlen = sum(int(ln) for ln, o in ops if o in "MDN=X")
        end = pos + rlen if (rlen and not int(f[1]) & 4) else pos + 1
        body = struct.pack("<iiBBHHHiiii", tid.get(f[2], -1), pos, len(name), int(f[4]), _reg2bin(pos, end), len(cig), int(f[1]), len(s),
                           tid.get(f[6] if f[6] != "=" else f[2], -1), int(f[7]) - 1, int(f[8]))
        body += name + struct.pack("<%dI" % len(cig), *cig) + bytes(packed) + q + b"".join(_aux_bytes(t) for t in f[11:])
        out += struct.pack("<i", len(body)) + body
    with open(bam_path, "wb") as fo:
        def put(chunk):
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = c.compress(bytes(chunk)) + c.flush()
            fo.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25))
            fo.write(comp + struct.pack("<II", zlib.crc32(bytes(chunk)) & 0xFFFFFFFF, len(chunk)))
        for a in range(0, len(out), block):
            put(out[a:a + block])
        put(b"")
    return len(recs)


def bam_sections(path):
    """Inflate a BAM -> (header text, reference table bytes, record stream bytes)."""
    import struct
    d = gzip.open(path, "rb").read()
    assert d[:4] == b"BAM\1"
    lt = struct.unpack_from("<i", d, 4)[0]
    text = d[8:8 + lt].rstrip(b"\0").decode()
    p = 8 + lt
    r0 = p
    n_ref = struct.unpack_from("<i", d, p)[0]; p += 4
    for _ in range(n_ref):
        p += 4 + struct.unpack_from("<i", d, p)[0] + 4
    return text, d[r0:p], d[p:]


def bam_record_tags(records):
    """Record stream -> [(qname, flag, pos, {tag: value})] for integer tags HP/PS/PQ (diagnostics of the CLI haplotag test)."""
    import struct
    out, p = [], 0
    size = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4, "d": 8}
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}
    while p < len(records):
        bs = struct.unpack_from("<i", records, p)[0]
        r = records[p + 4:p + 4 + bs]; p += 4 + bs
        _, pos, l_name, _, _, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", r, 0)
        a = 32 + l_name + 4 * n_cig + (l_seq + 1) // 2 + l_seq
        tags = []
        while a < len(r):
            tag, t = r[a:a + 2].decode(), chr(r[a + 2]); a += 3
            if t in "ZH":
                a0 = a; e = r.index(b"\0", a); a = e + 1
            elif t == "B":
                sub = chr(r[a]); cnt = struct.unpack_from("<i", r, a + 1)[0]; a += 5 + cnt * size[sub]
            else:
                if t in fmt and tag in ("HP", "PS", "PQ"):
                    tags.append((tag, struct.unpack_from(fmt[t], r, a)[0]))
                a += size[t]
                continue
            if t == "Z" and tag == "HP":
                tags.append((tag, r[a0:a - 1].decode()))
        out.append((r[32:32 + l_name - 1].decode(), flag, pos, tags))
    return out


def write_table_vcf(path, V, chrom, length):
    """Phased table (abi.Variants with hp1_is_alt / phase_set) -> minimal phased VCF, the haplotag CLI's -s input."""
    with open(path, "w") as fo:
        fo.write("##fileformat=VCFv4.2\n##contig=<ID=%s,length=%d>\n" % (chrom, length))
        fo.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n##FORMAT=<ID=PS,Number=1,Type=Integer,Description="Phase set identifier">\n')
        fo.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n")
        for i in range(V.n):
            gt = "1|0" if V.hp1_is_alt[i] else "0|1"
            r, a = (x.decode() if isinstance(x, bytes) else str(x) for x in (V.ref_str[i], V.alt_str[i]))
            fo.write("%s\t%d\t.\t%s\t%s\t30\tPASS\t.\tGT:PS\t%s:%d\n" % (chrom, int(V.pos[i]) + 1, r, a, gt, int(V.phase_set[i])))


def make_multi_contig(d, specs, unmapped=3):
    """Several generated contigs in ONE set of files (d/multi.fa, multi.vcf, multi.sam): specs = [(contig name, synth kwargs, in_vcf)].
    A contig with in_vcf False is present in BAM/FASTA (and listed as ##contig) but has no VCF records; `unmapped` unplaced reads (refID -1)
    end the SAM.  Read names get the contig as a prefix so they stay unique across contigs.  -> list of Synth digests"""
    digests = []
    sq, recs, vhead, vrecs = [], [], None, []
    with open(os.path.join(d, "multi.fa"), "w") as fa:
        for name, kw, in_vcf in specs:
            s = Synth(**kw)
            digests.append(fixtures.input_digest(s))
            s.write_fasta(os.path.join(d, "one.fa"), name); fa.write(open(os.path.join(d, "one.fa")).read())
            s.write_sam(os.path.join(d, "one.sam"), name)
            for line in open(os.path.join(d, "one.sam")):
                if line.startswith("@SQ"):
                    sq.append(line)
                elif not line.startswith("@"):
                    recs.append(name + "_" + line)
            s.write_vcf(os.path.join(d, "one.vcf"), name)
            for line in open(os.path.join(d, "one.vcf")):
                if line.startswith("##contig"):
                    sq_v = line
                    vhead = (vhead or []) + [sq_v]
                elif not line.startswith("#") and in_vcf:
                    vrecs.append(line)
            s.close()
    with open(os.path.join(d, "multi.sam"), "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n" + "".join(sq))
        f.writelines(recs)
        for k in range(unmapped):
            f.write("unplaced%d\t4\t*\t0\t0\t*\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n" % k)
    with open(os.path.join(d, "multi.vcf"), "w") as f:
        f.write("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n" + "".join(vhead))
        f.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">\n')
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n")
        f.writelines(vrecs)
    for fn in ("one.fa", "one.sam", "one.vcf"):
        os.remove(os.path.join(d, fn))
    return digests


def write_bai(bam_path, bai_path=None):
    """Minimal BAI (SAM spec 5.2) for a coordinate-sorted BAM: per reference only the metadata pseudo-bin 37450 with the virtual-offset range of
    the reference's records (what htslib writes as its first chunk) and the mapped/unmapped counts; no binning/linear index."""
    import struct, zlib
    raw = open(bam_path, "rb").read()
    blocks, p, u = [], 0, 0                     # (compressed offset, inflated offset, inflated size)
    data = bytearray()
    while p < len(raw):
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        chunk = zlib.decompress(raw[p + 12 + xlen:p + bsize - 8], -15)
        blocks.append((p, u, len(chunk))); data += chunk; u += len(chunk); p += bsize
    starts = [b[1] for b in blocks]

    def voff(off, end=False):
        import bisect
        k = bisect.bisect_right(starts, off) - 1
        while end and k > 0 and off == blocks[k][1] and False:
            k -= 1
        while blocks[k][2] == 0 and k + 1 < len(blocks) and not end:
            k += 1
        return (blocks[k][0] << 16) | (off - blocks[k][1])
    lt = struct.unpack_from("<i", data, 4)[0]
    q = 8 + lt
    n_ref = struct.unpack_from("<i", data, q)[0]; q += 4
    for _ in range(n_ref):
        q += 4 + struct.unpack_from("<i", data, q)[0] + 4
    span = {}
    while q < len(data):
        bs, tid = struct.unpack_from("<ii", data, q)
        flag = struct.unpack_from("<H", data, q + 18)[0]
        s = span.setdefault(tid, [q, 0, 0, 0])
        s[1] = q + 4 + bs; s[2 if not flag & 4 else 3] += 1
        q += 4 + bs
    out = bytearray(b"BAI\1" + struct.pack("<i", n_ref))
    for t in range(n_ref):
        if t in span:
            b, e, nm, nu = span[t]
            out += struct.pack("<i", 1) + struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", voff(b), voff(e, True), nm, nu) + struct.pack("<i", 0)
        else:
            out += struct.pack("<ii", 0, 0)
    open(bai_path or bam_path + ".bai", "wb").write(bytes(out))


def make_somatic_inputs(d, name, chrom="chrS"):
    """files of a tumor/normal fixture: ref.fa, normal_in.vcf, normal.sam, tumor.sam, tumor.vcf -> (normal digest, tumor digest)"""
    genome, nkw, tkw, cli, over = fixtures.ALL_SOMATIC[name]
    N = Synth(**dict(genome, **nkw)); T = Synth(**dict(genome, **tkw))
    N.write_fasta(os.path.join(d, "ref.fa"), chrom); N.write_vcf(os.path.join(d, "normal_in.vcf"), chrom); N.write_sam(os.path.join(d, "normal.sam"), chrom)
    T.write_sam(os.path.join(d, "tumor_plain.sam"), chrom); T.write_vcf_tumor(os.path.join(d, "tumor.vcf"), chrom, with_germline=True)
    add_stale_tags(os.path.join(d, "tumor_plain.sam"), os.path.join(d, "tumor.sam"))
    dg = (fixtures.input_digest(N), fixtures.input_digest(T)); indel = N.params["indel_var_frac"] > 0
    N.close(); T.close()
    return dg, indel
