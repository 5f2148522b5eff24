"""ctypes binding of tools/liblps_synth.so — seeded synthetic contig / variants / alignments.

Test + bench infrastructure only (SURVEY.md §8d "Concrete synthetic inputs").  The arrays returned are
views on memory owned by the generator handle; keep the `Synth` object alive while they are in use.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "..", "tools", "liblps_synth.so")


class SynthParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("contig_len", C.c_int64), ("n_snp", C.c_int32), ("coverage", C.c_double),
        ("len_median", C.c_double), ("len_sigma", C.c_double), ("len_min", C.c_int32), ("len_max", C.c_int32),
        ("sub_rate", C.c_double), ("ins_rate", C.c_double), ("del_rate", C.c_double),
        ("indel_var_frac", C.c_double), ("lowq_frac", C.c_double), ("mapq0_frac", C.c_double),
        ("secondary_frac", C.c_double), ("dup_frac", C.c_double), ("clip_every", C.c_int32),
        ("supp_frac", C.c_double), ("supp_overlap_frac", C.c_double), ("hpoly_every", C.c_double),
        ("snp_in_hpoly_frac", C.c_double), ("snp_pair_frac", C.c_double), ("tandem_frac", C.c_double),
        ("n_threads", C.c_int32), ("clip_pileups", C.c_int32), ("gap_start", C.c_int64), ("gap_len", C.c_int64), ("read_seed", C.c_uint64), ("somatic_every", C.c_double), ("tumor_purity", C.c_double),
        ("sv_every", C.c_double),
    ]


DEFAULTS = dict(
    seed=1, contig_len=5_000_000, n_snp=5000, coverage=10.0, len_median=15000.0, len_sigma=0.7585,
    len_min=1000, len_max=200000, sub_rate=0.01, ins_rate=0.01, del_rate=0.01, indel_var_frac=0.0,
    lowq_frac=0.10, mapq0_frac=0.01, secondary_frac=0.003, dup_frac=0.002, clip_every=7, supp_frac=0.02,
    supp_overlap_frac=0.5, hpoly_every=2000.0, snp_in_hpoly_frac=0.05, snp_pair_frac=0.01, tandem_frac=0.3,
    n_threads=8, clip_pileups=0, gap_start=0, gap_len=0, read_seed=0, somatic_every=0.0, tumor_purity=0.6, sv_every=0.0,
)

_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise RuntimeError(f"{_LIB} missing - run __graft_entry__.build()")
        L = C.CDLL(_LIB)
        L.synth_create.restype = C.c_void_p
        L.synth_create.argtypes = [C.POINTER(SynthParams)]
        L.synth_destroy.argtypes = [C.c_void_p]
        for n in ("synth_n_reads", "synth_n_variants", "synth_n_somatic", "synth_n_sv"):
            getattr(L, n).restype = C.c_int64
            getattr(L, n).argtypes = [C.c_void_p]
        for n in ("synth_ref", "synth_var_pos", "synth_var_hap", "synth_ref_start", "synth_l_qseq", "synth_flag",
                  "synth_mapq", "synth_name_id", "synth_read_hap", "synth_cigar_off", "synth_seq_off",
                  "synth_qual_off", "synth_cigar", "synth_seq", "synth_qual", "synth_som_pos", "synth_som_ref", "synth_som_alt", "synth_som_hap",
                  "synth_sv_pos", "synth_sv_len", "synth_sv_hap"):
            getattr(L, n).restype = C.c_void_p
            getattr(L, n).argtypes = [C.c_void_p]
        for n in ("synth_var_ref", "synth_var_alt"):
            getattr(L, n).restype = C.c_char_p
            getattr(L, n).argtypes = [C.c_void_p, C.c_int64]
        L.synth_write_fasta.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.synth_write_sam.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.synth_write_vcf.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        L.synth_write_vcf_tumor.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        _lib = L
    return _lib


def _view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class Synth:
    def __init__(self, **kw):
        L = _load()
        p = dict(DEFAULTS)
        p.update(kw)
        self.params = p
        sp = SynthParams(**p)
        self._h = L.synth_create(C.byref(sp))
        h = self._h
        n = self.n_reads = L.synth_n_reads(h)
        nv = self.n_variants = L.synth_n_variants(h)
        self.contig_len = p["contig_len"]
        self.ref = _view(L.synth_ref(h), self.contig_len, np.uint8)
        self.var_pos = _view(L.synth_var_pos(h), nv, np.int32)
        self.var_hap = _view(L.synth_var_hap(h), nv, np.uint8)
        self.var_ref = [L.synth_var_ref(h, i) for i in range(nv)]
        self.var_alt = [L.synth_var_alt(h, i) for i in range(nv)]
        ns = self.n_somatic = L.synth_n_somatic(h)
        self.som_pos = _view(L.synth_som_pos(h), ns, np.int32)
        self.som_ref = _view(L.synth_som_ref(h), ns, np.uint8)
        self.som_alt = _view(L.synth_som_alt(h), ns, np.uint8)
        self.som_hap = _view(L.synth_som_hap(h), ns, np.uint8)
        nx = self.n_sv = L.synth_n_sv(h)
        self.sv_pos = _view(L.synth_sv_pos(h), nx, np.int32)
        self.sv_len = _view(L.synth_sv_len(h), nx, np.int32)
        self.sv_hap = _view(L.synth_sv_hap(h), nx, np.uint8)
        self.ref_start = _view(L.synth_ref_start(h), n, np.int32)
        self.l_qseq = _view(L.synth_l_qseq(h), n, np.int32)
        self.flag = _view(L.synth_flag(h), n, np.uint16)
        self.mapq = _view(L.synth_mapq(h), n, np.uint8)
        self.name_id = _view(L.synth_name_id(h), n, np.uint32)
        self.read_hap = _view(L.synth_read_hap(h), n, np.uint8)
        self.cigar_off = _view(L.synth_cigar_off(h), n + 1, np.uint64)
        self.seq_off = _view(L.synth_seq_off(h), n + 1, np.uint64)
        self.qual_off = _view(L.synth_qual_off(h), n + 1, np.uint64)
        self.cigar = _view(L.synth_cigar(h), int(self.cigar_off[-1]) if n else 0, np.uint32)
        self.seq = _view(L.synth_seq(h), int(self.seq_off[-1]) if n else 0, np.uint8)
        self.qual = _view(L.synth_qual(h), int(self.qual_off[-1]) if n else 0, np.uint8)

    def write_fasta(self, path, chrom="chrS"):
        assert _load().synth_write_fasta(self._h, path.encode(), chrom.encode()) == 0

    def write_sam(self, path, chrom="chrS"):
        assert _load().synth_write_sam(self._h, path.encode(), chrom.encode()) == 0

    def write_vcf(self, path, chrom="chrS", phased=False):
        assert _load().synth_write_vcf(self._h, path.encode(), chrom.encode(), int(phased)) == 0

    def write_vcf_tumor(self, path, chrom="chrS", with_germline=False):
        assert _load().synth_write_vcf_tumor(self._h, path.encode(), chrom.encode(), int(with_germline)) == 0

    def close(self):
        if self._h:
            _load().synth_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- SV / MOD inputs of `phase` (test infrastructure): what a SV caller and `modcall` would have written for the generated reads
def read_ref_end(s):
    """Reference position after the last CIGAR operation of every alignment."""
    op = s.cigar & 15
    ln = (s.cigar >> 4).astype(np.int64)
    cons = np.where((op == 0) | (op == 2) | (op == 3) | (op == 7) | (op == 8), ln, 0)
    cs = np.concatenate([[0], np.cumsum(cons)])
    off = s.cigar_off.astype(np.int64)
    return s.ref_start.astype(np.int64) + cs[off[1:]] - cs[off[:-1]]


def make_mod_lines(s, mod_every=2500.0, seed=0, pair_frac=0.5, noise=0.1, wrong_strand=0.05, listed=0.9, taken=()):
    """-> list of (pos0, reverse, [(name_id, modified)]) = the records of a modcall VCF: one strand per record, consecutive positions are
    merged by the reader under the first one (METHParser, src/phase/ParsingBam.cpp:1707-1711).  Positions avoid `taken` and the SNP / SV rows."""
    g = np.random.default_rng(seed)
    avoid = set(int(p) for p in s.var_pos) | set(int(p) for p in s.sv_pos) | set(int(p) + 1 for p in s.sv_pos) | set(int(p) for p in taken)
    end = read_ref_end(s)
    order = np.argsort(s.ref_start, kind="stable")
    lines = []
    x = 500.0 + g.exponential(mod_every)
    hap_of_site = {}
    while x < s.contig_len - 500:
        p = int(x)
        x += 30 + g.exponential(mod_every)
        positions = [(p, False), (p + 1, True)] if g.random() < pair_frac else [(p, bool(g.random() < 0.5))]
        if any(q in avoid or q - 1 in avoid or q + 1 in avoid for q, _ in positions):
            continue
        mod_hap = int(g.integers(2))
        for q, rev in positions:
            cover = np.nonzero((s.ref_start <= q) & (end > q))[0]
            reads = []
            for r in cover:
                r_rev = bool(s.flag[r] & 0x10)
                if g.random() > listed:
                    continue
                if r_rev != rev and g.random() > wrong_strand:
                    continue
                modified = (int(s.read_hap[r]) == mod_hap) != (g.random() < noise)
                reads.append((int(s.name_id[r]), bool(modified)))
            if reads:
                lines.append((q, rev, reads))
    return lines


def merge_mod_lines(lines):
    """The representative rows METHParser builds from well-formed heterozygous records: (positions, per row [(name_id, modified, reverse)])."""
    rows = {}
    rep = None
    prev = None
    for q, rev, reads in lines:
        if prev is None or prev + 1 != q:
            rep = q
        d = rows.setdefault(rep, {})
        for name, modified in sorted(reads, key=lambda e: not e[1]):     # the reader takes the MR= list first, then NR= (:1746-1780)
            d[name] = (modified, rev)          # a name listed twice under one representative: the later entry wins (std::map assignment)
        prev = q
    pos = sorted(rows)
    return pos, [[(n, m, r) for n, (m, r) in sorted(rows[p].items())] for p in pos]


def sv_read_names(s, prefix=""):
    """Per generated SV the names of the alignments that span it and come from the haplotype carrying it (what a SV caller lists under RNAMES=)."""
    end = read_ref_end(s)
    out = []
    for p, h in zip(s.sv_pos, s.sv_hap):
        idx = np.nonzero((s.ref_start <= p) & (end > p + 1) & (s.read_hap == h))[0]
        out.append(sorted(set("%sr%09d" % (prefix, int(s.name_id[i])) for i in idx)))
    return out


def write_sv_vcf(path, chrom, sv_pos, sv_len, contig_len, gt=None, rnames=None):
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n##contig=<ID=%s,length=%d>\n##INFO=<ID=SVTYPE,Number=1,Type=String,Description=\"\">\n" % (chrom, contig_len))
        f.write("##INFO=<ID=SVLEN,Number=1,Type=Integer,Description=\"\">\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"\">\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n")
        for i, (p, l) in enumerate(zip(sv_pos, sv_len)):
            g = "0/1" if gt is None else gt[i]
            rn = "" if rnames is None else ";RNAMES=" + ",".join(rnames[i])
            f.write("%s\t%d\tsv%d\tN\t<%s>\t60\tPASS\tSVTYPE=%s;SVLEN=%d;END=%d%s\tGT:DR:DV\t%s:10:10\n"
                    % (chrom, int(p) + 1, i, "INS" if l > 0 else "DEL", "INS" if l > 0 else "DEL", int(l), int(p) + 1 + (0 if l > 0 else -int(l)), rn, g))


def write_mod_vcf(path, chrom, lines, contig_len, gt=None):
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n##contig=<ID=%s,length=%d>\n##INFO=<ID=RS,Number=1,Type=String,Description=\"\">\n" % (chrom, contig_len))
        f.write("##INFO=<ID=MR,Number=.,Type=String,Description=\"\">\n##INFO=<ID=NR,Number=.,Type=String,Description=\"\">\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"\">\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n")
        for i, (q, rev, reads) in enumerate(lines):
            mr = ",".join("r%09d" % n for n, m in reads if m)
            nr = ",".join("r%09d" % n for n, m in reads if not m)
            g = "0/1" if gt is None else gt[i]
            f.write("%s\t%d\t.\t%s\t<MOD>\t.\tPASS\tRS=%s;MR=%s;NR=%s;\tGT:MD:UD\t%s:%d:%d\n"
                    % (chrom, q + 1, "G" if rev else "C", "N" if rev else "P", mr, nr, g, sum(1 for _, m in reads if m), sum(1 for _, m in reads if not m)))


def make_extras_fast(h, mod_every=2000.0, sv_every=150000.0, seed=0, noise=0.1, listed=0.9):
    """SV and MOD rows for a large generated contig (host arrays of Synth or SynthGpu.to_host()): -> (sv_pos, sv_len, mod_pos, mod_off, mod_name,
    mod_flag) ready for lps_extra_variants.  One record per MOD row, reads of the record's strand, modified on one haplotype (with noise); the SV
    rows are not carried by any read (every spanning read is called REF).  Positions keep 3 bp away from the SNP rows and from each other."""
    g = np.random.default_rng(seed)
    end = read_ref_end(h)
    start = h.ref_start.astype(np.int64)
    rev = (h.flag & 0x10) != 0
    taken = np.asarray(h.var_pos, np.int64)

    def free(p):
        i = np.searchsorted(taken, p - 3)
        return i >= taken.size or taken[i] > p + 3

    sv_pos, sv_len, x = [], [], 5000.0
    while x < h.contig_len - 5000:
        p = int(x); x += 2000 + g.exponential(sv_every)
        if free(p):
            sv_pos.append(p); sv_len.append(int(g.integers(50, 800)) * (1 if g.random() < 0.5 else -1))
    sv_set = set(sv_pos)
    reach = int((end - start).max()) + 1
    mod_pos, off, names, flags = [], [0], [], []
    x = 500.0 + g.exponential(mod_every)
    while x < h.contig_len - 500:
        q = int(x); x += 10 + g.exponential(mod_every)
        if not free(q) or any((q + d) in sv_set for d in (-2, -1, 0, 1, 2)):
            continue
        lo, hi = np.searchsorted(start, q - reach, side="left"), np.searchsorted(start, q, side="right")
        idx = lo + np.nonzero(end[lo:hi] > q)[0]
        strand = bool(g.random() < 0.5)
        idx = idx[(rev[idx] == strand) & (g.random(idx.size) < listed)]
        if idx.size == 0:
            continue
        hap = int(g.integers(2))
        modified = (h.read_hap[idx] == hap) != (g.random(idx.size) < noise)
        nid, first = np.unique(h.name_id[idx], return_index=True)     # a name once per row
        mod_pos.append(q)
        names.append(nid.astype(np.uint32)); flags.append((modified[first].astype(np.uint8)) | (2 if strand else 0))
        off.append(off[-1] + nid.size)
    cat = lambda a, t: np.concatenate(a).astype(t) if a else np.zeros(0, t)
    return (np.array(sv_pos, np.int32), np.array(sv_len, np.int32), np.array(mod_pos, np.int32), np.array(off, np.uint64), cat(names, np.uint32), cat(flags, np.uint8))
