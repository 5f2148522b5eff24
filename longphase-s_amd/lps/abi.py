"""ctypes mirror of include/lps_abi.h (structs + array packing helpers).

Used by the test harness, bench.py and __graft_entry__ to drive liblps_hip.so through its C-ABI, and by the
tests to drive the CPU oracle (oracle/liblps_oracle.so) with the very same structs.
"""
import ctypes as C
import numpy as np

LPS_MAX_STAGES = 24


class Params(C.Structure):
    _fields_ = [
        ("is_ont", C.c_int32), ("phase_indel", C.c_int32), ("distance", C.c_int32),
        ("connect_adjacent", C.c_int32), ("mapping_quality", C.c_int32), ("base_quality", C.c_int32),
        ("edge_weight", C.c_double), ("snp_confidence", C.c_double), ("read_confidence", C.c_double),
        ("edge_threshold", C.c_double), ("overlap_threshold", C.c_double),
        ("percentage_threshold", C.c_double), ("tag_supplementary", C.c_int32), ("reserved", C.c_int32),
    ]


def default_params(**kw):
    """Reference defaults: src/phase/Phasing.cpp:88-116, src/haplotag/Haplotag.cpp:60-72."""
    p = Params(is_ont=1, phase_indel=0, distance=300000, connect_adjacent=35, mapping_quality=1,
               base_quality=12, edge_weight=0.1, snp_confidence=0.75, read_confidence=0.65,
               edge_threshold=0.7, overlap_threshold=0.2, percentage_threshold=0.6, tag_supplementary=0,
               reserved=0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class VariantTable(C.Structure):
    _fields_ = [("n", C.c_int64), ("pos", C.c_void_p), ("ref0", C.c_void_p), ("alt0", C.c_void_p),
                ("ref_len", C.c_void_p), ("alt_len", C.c_void_p), ("hp1_is_alt", C.c_void_p),
                ("phase_set", C.c_void_p), ("somatic_role", C.c_void_p), ("derive_hp", C.c_void_p), ("tumor_kind", C.c_void_p)]


class ReadBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("ref_start", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p),
                ("l_qseq", C.c_void_p), ("name_id", C.c_void_p), ("cigar_off", C.c_void_p),
                ("cigar", C.c_void_p), ("seq_off", C.c_void_p), ("seq", C.c_void_p), ("qual_off", C.c_void_p),
                ("qual", C.c_void_p)]


class PhaseResult(C.Structure):
    _fields_ = [("n", C.c_int64), ("phase_set", C.c_void_p), ("gt", C.c_void_p)]


class HaplotagResult(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("status", C.c_void_p), ("hp1", C.c_void_p), ("hp2", C.c_void_p),
                ("n_ps", C.c_void_p), ("ps_min", C.c_void_p), ("hp", C.c_void_p), ("pq", C.c_void_p),
                ("ps", C.c_void_p)]


class SomaticTagResult(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("status", C.c_void_p), ("hp1", C.c_void_p), ("hp2", C.c_void_p), ("hp3", C.c_void_p),
                ("derive_h1", C.c_void_p), ("derive_h2", C.c_void_p), ("n_ps", C.c_void_p), ("ps_min", C.c_void_p),
                ("hp", C.c_void_p), ("pq", C.c_void_p), ("ps", C.c_void_p)]


LPS_SITE_COUNTERS = 18
SC = dict(ALT=0, A=1, C=2, G=3, T=4, UNKNOWN=5, DEPTH=6, DEL=7, MPQ_ALT=8, MPQ_A=9, MPQ_C=10, MPQ_G=11, MPQ_T=12, MPQ_UNKNOWN=13,
          MPQ_DEPTH=14, READHP_UNTAG=15, READHP_H1=16, READHP_H2=17)


class SiteCounters(C.Structure):
    _fields_ = [("n", C.c_int64), ("counters", C.c_void_p), ("n_reads", C.c_int64), ("read_hp", C.c_void_p)]


LPS_TSITE_COUNTERS = 41


class TumorExtractResult(C.Structure):
    _fields_ = [("n", C.c_int64), ("site", C.c_void_p), ("n_reads", C.c_int64), ("status", C.c_void_p), ("hp1", C.c_void_p),
                ("hp2", C.c_void_p), ("hp3", C.c_void_p), ("hp", C.c_void_p), ("n_ps", C.c_void_p), ("ps_min", C.c_void_p),
                ("end_pos", C.c_void_p), ("read_len", C.c_void_p), ("has_site", C.c_void_p),
                ("pair_capacity", C.c_int64), ("n_pairs", C.c_int64), ("pair_site", C.c_void_p), ("pair_read", C.c_void_p),
                ("pair_base_hp", C.c_void_p), ("win_capacity", C.c_int64), ("n_windows", C.c_int64), ("win_site", C.c_void_p),
                ("win_allele", C.c_void_p), ("win_offset", C.c_void_p), ("win_base", C.c_void_p)]


READ_HP_STR = [".", "1", "2", "3", "4", "1-1", "1-2", "2-1", "2-2"]   # ReadHapUtil::readHapIntToString (HaplotagType.h:327-342)


class Timings(C.Structure):
    _fields_ = [("n_stages", C.c_int32), ("ms_kernel", C.c_float * LPS_MAX_STAGES), ("ms_total", C.c_float),
                ("n_obs", C.c_int64), ("n_nodes", C.c_int64), ("n_pairs", C.c_int64), ("n_reads_used", C.c_int64),
                ("algorithmic_bytes", C.c_int64 * LPS_MAX_STAGES), ("n_scan_segments", C.c_int64), ("n_scan_replayed", C.c_int64)]


def _ptr(a):
    return None if a is None else a.ctypes.data


class Variants:
    """Host-side variant table (keeps the numpy arrays alive next to the ctypes struct)."""

    def __init__(self, pos, ref, alt, hp1_is_alt=None, phase_set=None, somatic_role=None, derive_hp=None, tumor_kind=None):
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        n = self.pos.size
        self.ref_str = [r if isinstance(r, bytes) else r.encode() for r in ref]
        self.alt_str = [a if isinstance(a, bytes) else a.encode() for a in alt]
        self.ref0 = np.frombuffer(b"".join(r[:1] for r in self.ref_str), dtype=np.uint8).copy() if n else np.zeros(0, np.uint8)
        self.alt0 = np.frombuffer(b"".join(a[:1] for a in self.alt_str), dtype=np.uint8).copy() if n else np.zeros(0, np.uint8)
        self.ref_len = np.array([len(r) for r in self.ref_str], dtype=np.uint16)
        self.alt_len = np.array([len(a) for a in self.alt_str], dtype=np.uint16)
        self.hp1_is_alt = None if hp1_is_alt is None else np.ascontiguousarray(hp1_is_alt, dtype=np.uint8)
        self.phase_set = None if phase_set is None else np.ascontiguousarray(phase_set, dtype=np.int32)
        self.somatic_role = None if somatic_role is None else np.ascontiguousarray(somatic_role, dtype=np.uint8)
        self.derive_hp = None if derive_hp is None else np.ascontiguousarray(derive_hp, dtype=np.uint8)
        self.tumor_kind = None if tumor_kind is None else np.ascontiguousarray(tumor_kind, dtype=np.uint8)
        self.n = n
        self.c = VariantTable(n, _ptr(self.pos), _ptr(self.ref0), _ptr(self.alt0), _ptr(self.ref_len),
                              _ptr(self.alt_len), _ptr(self.hp1_is_alt), _ptr(self.phase_set), _ptr(self.somatic_role),
                              _ptr(self.derive_hp), _ptr(self.tumor_kind))


    @classmethod
    def from_snps(cls, pos, ref0, alt0, hp1_is_alt=None, phase_set=None):
        """SNP-only table straight from arrays (no per-row Python objects: whole-genome tables have millions of rows)."""
        self = cls.__new__(cls)
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        n = self.n = int(self.pos.size)
        self.ref0 = np.ascontiguousarray(ref0, dtype=np.uint8); self.alt0 = np.ascontiguousarray(alt0, dtype=np.uint8)
        self.ref_len = np.ones(n, np.uint16); self.alt_len = np.ones(n, np.uint16)
        self.ref_str = self.alt_str = None
        self.hp1_is_alt = None if hp1_is_alt is None else np.ascontiguousarray(hp1_is_alt, dtype=np.uint8)
        self.phase_set = None if phase_set is None else np.ascontiguousarray(phase_set, dtype=np.int32)
        self.somatic_role = self.derive_hp = self.tumor_kind = None
        self.c = VariantTable(n, _ptr(self.pos), _ptr(self.ref0), _ptr(self.alt0), _ptr(self.ref_len), _ptr(self.alt_len),
                              _ptr(self.hp1_is_alt), _ptr(self.phase_set), None, None, None)
        return self


class ExtraVariantTable(C.Structure):
    _fields_ = [("n_sv", C.c_int64), ("sv_pos", C.c_void_p), ("sv_len", C.c_void_p), ("n_mod", C.c_int64), ("mod_pos", C.c_void_p),
                ("mod_off", C.c_void_p), ("mod_name", C.c_void_p), ("mod_flag", C.c_void_p), ("sv_window", C.c_int32), ("reserved", C.c_int32),
                ("sv_threshold", C.c_double)]


class ExtraVariants:
    """SV and MOD rows co-phased with the SNPs (lps_extra_variants).  mod_rows: per MOD row a list of (name_id, modified, reverse)."""

    def __init__(self, sv_pos=(), sv_len=(), mod_pos=(), mod_rows=(), sv_window=20, sv_threshold=0.1):
        self.sv_pos = np.ascontiguousarray(sv_pos, dtype=np.int32); self.sv_len = np.ascontiguousarray(sv_len, dtype=np.int32)
        self.mod_pos = np.ascontiguousarray(mod_pos, dtype=np.int32)
        assert self.sv_pos.size == self.sv_len.size and self.mod_pos.size == len(mod_rows)
        off = [0]; names = []; flags = []
        for row in mod_rows:
            row = sorted(row)
            names += [int(r[0]) for r in row]; flags += [(1 if r[1] else 0) | (2 if r[2] else 0) for r in row]
            off.append(len(names))
        self.mod_off = np.array(off, np.uint64); self.mod_name = np.array(names, np.uint32); self.mod_flag = np.array(flags, np.uint8)
        self.n_sv = int(self.sv_pos.size); self.n_mod = int(self.mod_pos.size)
        self.c = ExtraVariantTable(self.n_sv, _ptr(self.sv_pos), _ptr(self.sv_len), self.n_mod, _ptr(self.mod_pos), _ptr(self.mod_off),
                                   _ptr(self.mod_name), _ptr(self.mod_flag), sv_window, 0, sv_threshold)


def extra_from_arrays(sv_pos, sv_len, mod_pos, mod_off, mod_name, mod_flag, sv_window=20, sv_threshold=0.1):
    """ExtraVariants from flat arrays (large tables: no per-row Python lists)."""
    x = ExtraVariants.__new__(ExtraVariants)
    x.sv_pos = np.ascontiguousarray(sv_pos, np.int32); x.sv_len = np.ascontiguousarray(sv_len, np.int32); x.mod_pos = np.ascontiguousarray(mod_pos, np.int32)
    x.mod_off = np.ascontiguousarray(mod_off, np.uint64); x.mod_name = np.ascontiguousarray(mod_name, np.uint32); x.mod_flag = np.ascontiguousarray(mod_flag, np.uint8)
    x.n_sv = int(x.sv_pos.size); x.n_mod = int(x.mod_pos.size)
    x.c = ExtraVariantTable(x.n_sv, _ptr(x.sv_pos), _ptr(x.sv_len), x.n_mod, _ptr(x.mod_pos), _ptr(x.mod_off), _ptr(x.mod_name), _ptr(x.mod_flag), sv_window, 0, sv_threshold)
    return x


class Reads:
    """Host-side SoA read batch."""

    FIELDS = (("ref_start", np.int32), ("flag", np.uint16), ("mapq", np.uint8), ("l_qseq", np.int32),
              ("name_id", np.uint32), ("cigar_off", np.uint64), ("cigar", np.uint32), ("seq_off", np.uint64),
              ("seq", np.uint8), ("qual_off", np.uint64), ("qual", np.uint8))

    def __init__(self, **arrays):
        for name, dt in self.FIELDS:
            setattr(self, name, np.ascontiguousarray(arrays[name], dtype=dt))
        self.n_reads = int(self.ref_start.size)
        assert self.cigar_off.size == self.n_reads + 1
        self.c = ReadBatch(self.n_reads, *[_ptr(getattr(self, n)) for n, _ in self.FIELDS])

    @classmethod
    def from_synth(cls, s):
        return cls(**{n: getattr(s, n) for n, _ in cls.FIELDS})

    def subset(self, idx):
        """New batch holding reads idx (in that order)."""
        idx = np.asarray(idx, dtype=np.int64)
        out = {}
        for n in ("ref_start", "flag", "mapq", "l_qseq", "name_id"):
            out[n] = getattr(self, n)[idx]
        for base, off in (("cigar", "cigar_off"), ("seq", "seq_off"), ("qual", "qual_off")):
            o = getattr(self, off)
            lens = (o[idx + 1] - o[idx]).astype(np.int64)
            new_off = np.zeros(idx.size + 1, dtype=np.uint64)
            np.cumsum(lens, out=new_off[1:])
            src = getattr(self, base)
            if idx.size:
                gather = np.concatenate([np.arange(int(o[i]), int(o[i + 1]), dtype=np.int64) for i in idx]) if lens.sum() else np.zeros(0, np.int64)
                out[base] = src[gather]
            else:
                out[base] = src[:0]
            out[off] = new_off
        return Reads(**out)


class PhaseOut:
    def __init__(self, n):
        self.phase_set = np.zeros(n, dtype=np.int32)
        self.gt = np.zeros(n, dtype=np.uint8)
        self.c = PhaseResult(n, _ptr(self.phase_set), _ptr(self.gt))


class HaplotagOut:
    def __init__(self, n):
        self.status = np.zeros(n, np.uint8)
        self.hp1 = np.zeros(n, np.int32)
        self.hp2 = np.zeros(n, np.int32)
        self.n_ps = np.zeros(n, np.uint8)
        self.ps_min = np.zeros(n, np.int32)
        self.hp = np.zeros(n, np.uint8)
        self.pq = np.zeros(n, np.int32)
        self.ps = np.zeros(n, np.int32)
        self.c = HaplotagResult(n, *[_ptr(getattr(self, k)) for k in
                                     ("status", "hp1", "hp2", "n_ps", "ps_min", "hp", "pq", "ps")])


class SomaticTagOut:
    I32 = ("hp1", "hp2", "hp3", "derive_h1", "derive_h2", "ps_min", "pq", "ps")
    U8 = ("status", "n_ps", "hp")

    def __init__(self, n):
        for k in self.I32:
            setattr(self, k, np.zeros(n, np.int32))
        for k in self.U8:
            setattr(self, k, np.zeros(n, np.uint8))
        self.c = SomaticTagResult(n, *[_ptr(getattr(self, k)) for k in
                                       ("status", "hp1", "hp2", "hp3", "derive_h1", "derive_h2", "n_ps", "ps_min", "hp", "pq", "ps")])


class SiteCountersOut:
    def __init__(self, n_var, n_reads):
        self.counters = np.zeros((n_var, LPS_SITE_COUNTERS), np.int32)
        self.read_hp = np.zeros(n_reads, np.uint8)
        self.c = SiteCounters(n_var, _ptr(self.counters), n_reads, _ptr(self.read_hp))


class TumorExtractOut:
    def __init__(self, n_var, n_reads, pair_cap, win_cap):
        self.site = np.zeros((n_var, LPS_TSITE_COUNTERS), np.int32)
        for k in ("hp1", "hp2", "hp3", "ps_min", "end_pos", "read_len"):
            setattr(self, k, np.zeros(n_reads, np.int32))
        for k in ("status", "hp", "n_ps", "has_site"):
            setattr(self, k, np.zeros(n_reads, np.uint8))
        self.pair_site = np.zeros(pair_cap, np.int32); self.pair_read = np.zeros(pair_cap, np.int32); self.pair_base_hp = np.zeros(pair_cap, np.uint8)
        self.win_site = np.zeros(win_cap, np.int32); self.win_allele = np.zeros(win_cap, np.uint8)
        self.win_offset = np.zeros(win_cap, np.int16); self.win_base = np.zeros(win_cap, np.uint8)
        self.c = TumorExtractResult(n_var, _ptr(self.site), n_reads, _ptr(self.status), _ptr(self.hp1), _ptr(self.hp2), _ptr(self.hp3),
                                    _ptr(self.hp), _ptr(self.n_ps), _ptr(self.ps_min), _ptr(self.end_pos), _ptr(self.read_len), _ptr(self.has_site),
                                    pair_cap, 0, _ptr(self.pair_site), _ptr(self.pair_read), _ptr(self.pair_base_hp),
                                    win_cap, 0, _ptr(self.win_site), _ptr(self.win_allele), _ptr(self.win_offset), _ptr(self.win_base))

    def pairs(self):
        n = self.c.n_pairs
        o = np.lexsort((self.pair_read[:n], self.pair_site[:n]))
        return self.pair_site[:n][o], self.pair_read[:n][o], self.pair_base_hp[:n][o]

    def windows(self):
        n = self.c.n_windows
        o = np.lexsort((self.win_base[:n], self.win_offset[:n], self.win_allele[:n], self.win_site[:n]))
        return self.win_site[:n][o], self.win_allele[:n][o], self.win_offset[:n][o], self.win_base[:n][o]


class BamRecords:
    """Raw (inflated) BAM records of one contig for lps_push_bam_records: byte blob + offset of each record's refID field."""

    def __init__(self, blob, rec_off, name_id):
        self.blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self.rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        self.name_id = np.ascontiguousarray(name_id, dtype=np.uint32)
        self.n_reads = int(self.rec_off.size)

    @classmethod
    def from_reads(cls, R, names=None, seed=0, lead=0):
        """Serialise an abi.Reads batch as BAM records (SAM spec 4.2) with variable-length names and a random aux tail,
        so that CIGAR/seq/qual land on every byte alignment.  `lead` = junk bytes before the first record."""
        import struct
        rng = np.random.default_rng(seed)
        out = bytearray(rng.integers(0, 256, lead, dtype=np.uint8).tobytes())
        off = []
        for i in range(R.n_reads):
            nm = (names[i] if names is not None else b"r%09d" % int(R.name_id[i])) + b"x" * int(rng.integers(0, 4)) + b"\0"
            cg = R.cigar[int(R.cigar_off[i]):int(R.cigar_off[i + 1])]
            lq = int(R.l_qseq[i])
            sq = R.seq[int(R.seq_off[i]):int(R.seq_off[i]) + (lq + 1) // 2]
            ql = R.qual[int(R.qual_off[i]):int(R.qual_off[i]) + lq]
            aux = rng.integers(0, 256, int(rng.integers(0, 9)), dtype=np.uint8).tobytes()
            cgw = cg
            if cg.size > 65535:
                # more operations than the 16-bit n_cigar_op holds: placeholder <l_seq>S<ref_len>N + the real CIGAR in a CG:B,I field (SAM spec 4.2.2),
                # here between two ordinary fields
                rlen = int(sum(int(w) >> 4 for w in cg if (int(w) & 15) in (0, 2, 3, 7, 8)))
                cgw = np.array([(lq << 4) | 4, (rlen << 4) | 3], dtype=np.uint32)
                aux = b"NMi" + struct.pack("<i", 5) + b"CGBI" + struct.pack("<I", cg.size) + cg.astype("<u4").tobytes() + b"XSZabc\0"
            body = struct.pack("<iiBBHHHiiii", 0, int(R.ref_start[i]), len(nm), int(R.mapq[i]), 4680, cgw.size, int(R.flag[i]), lq, -1, -1, 0)
            body += nm + cgw.astype("<u4").tobytes() + sq.tobytes() + ql.tobytes() + aux
            out += struct.pack("<i", len(body))
            off.append(len(out))
            out += body
        return cls(np.frombuffer(bytes(out), dtype=np.uint8), off, R.name_id)
