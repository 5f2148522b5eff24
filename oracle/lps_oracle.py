"""TEST INFRASTRUCTURE ONLY - ctypes binding of oracle/liblps_oracle.so (the CPU restatement, the checker).

Import only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "..", "longphase-s_amd"))
from lps import abi  # noqa: E402


class Dumps(C.Structure):
    _fields_ = [
        ("obs_capacity", C.c_int64), ("obs_count", C.c_void_p), ("obs_var", C.c_void_p),
        ("obs_allele", C.c_void_p), ("obs_quality", C.c_void_p), ("n_obs", C.c_int64),
        ("clip_capacity", C.c_int64), ("clip_pos", C.c_void_p), ("clip_fb", C.c_void_p), ("n_clips", C.c_int64),
        ("node_capacity", C.c_int64), ("node_var", C.c_void_p), ("edge", C.c_void_p), ("node_hp", C.c_void_p),
        ("node_block", C.c_void_p), ("n_nodes", C.c_int64), ("aln_deleted", C.c_void_p),
        ("n_cnv", C.c_int32), ("cnv_capacity", C.c_int32), ("cnv_start", C.c_void_p), ("cnv_end", C.c_void_p),
        ("ub_hazard", C.c_int32), ("n_pairs", C.c_int64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liblps_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liblps_oracle.so missing - run `make -C oracle` (or __graft_entry__.build())")
        _lib = C.CDLL(path)
        _lib.oracle_phase.restype = C.c_int
        _lib.oracle_phase.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.VariantTable), C.c_void_p, C.c_int64,
                                      C.POINTER(abi.ReadBatch), C.POINTER(abi.PhaseResult), C.POINTER(Dumps)]
        _lib.oracle_phase_x.restype = C.c_int
        _lib.oracle_phase_x.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.VariantTable), C.POINTER(abi.ExtraVariantTable), C.c_void_p, C.c_int64,
                                        C.POINTER(abi.ReadBatch), C.POINTER(abi.PhaseResult), C.POINTER(abi.PhaseResult), C.POINTER(abi.PhaseResult), C.POINTER(Dumps)]
        _lib.oracle_haplotag.restype = C.c_int
        _lib.oracle_haplotag.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.VariantTable), C.c_void_p, C.c_int64,
                                         C.POINTER(abi.ReadBatch), C.POINTER(abi.HaplotagResult)]
        _lib.oracle_haplotag_v.restype = C.c_int
        _lib.oracle_haplotag_v.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.VariantTable), C.c_void_p, C.c_int64,
                                           C.POINTER(abi.ReadBatch), C.c_void_p, C.c_void_p, C.POINTER(abi.HaplotagResult)]
        _lib.oracle_somatic_tag.restype = C.c_int
        _lib.oracle_somatic_tag.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.VariantTable), C.POINTER(abi.ReadBatch), C.POINTER(abi.SomaticTagResult)]
        _lib.oracle_somatic_extract_normal.restype = C.c_int
        _lib.oracle_somatic_extract_normal.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.VariantTable), C.c_void_p, C.c_int64, C.POINTER(abi.ReadBatch), C.POINTER(abi.SiteCounters)]
        _lib.oracle_somatic_extract_tumor.restype = C.c_int
        _lib.oracle_somatic_extract_tumor.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.VariantTable), C.c_void_p, C.c_int64, C.POINTER(abi.ReadBatch), C.POINTER(abi.TumorExtractResult)]
    return _lib


class PhaseDump:
    def __init__(self, n_reads, n_var, adj, obs_cap=None, with_edges=True):
        obs_cap = obs_cap or max(1, n_reads) * 64 + 1024
        self.obs_count = np.zeros(n_reads, np.int32)
        self.obs_var = np.zeros(obs_cap, np.int32)
        self.obs_allele = np.zeros(obs_cap, np.int8)
        self.obs_quality = np.zeros(obs_cap, np.int8)
        ccap = 2 * n_reads + 16
        self.clip_pos = np.zeros(ccap, np.int32)
        self.clip_fb = np.zeros(ccap, np.uint8)
        self.node_var = np.zeros(n_var, np.int32)
        self.edge = np.zeros((n_var, adj, 4), np.float32) if with_edges else None
        self.node_hp = np.zeros(n_var, np.int8)
        self.node_block = np.zeros(n_var, np.int32)
        self.aln_deleted = np.zeros(n_reads, np.uint8)
        self._cnv_start = np.zeros(4096, np.int32)
        self._cnv_end = np.zeros(4096, np.int32)
        p = lambda a: None if a is None else a.ctypes.data
        self.c = Dumps(obs_cap, p(self.obs_count), p(self.obs_var), p(self.obs_allele), p(self.obs_quality), 0,
                       ccap, p(self.clip_pos), p(self.clip_fb), 0,
                       n_var, p(self.node_var), p(self.edge), p(self.node_hp), p(self.node_block), 0,
                       p(self.aln_deleted), 0, 4096, p(self._cnv_start), p(self._cnv_end))

    def cnv_start(self):
        return self._cnv_start[:self.c.n_cnv]

    def cnv_end(self):
        return self._cnv_end[:self.c.n_cnv]


def phase(params, variants, ref, reads, dump=False, with_edges=True):
    """Run the CPU restatement.  ref: numpy uint8 array of the contig.  Returns (PhaseOut, PhaseDump|None)."""
    out = abi.PhaseOut(variants.n)
    d = PhaseDump(reads.n_reads, variants.n, params.connect_adjacent, with_edges=with_edges) if dump else None
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    rc = lib().oracle_phase(C.byref(params), C.byref(variants.c), ref.ctypes.data, ref.size, C.byref(reads.c),
                            C.byref(out.c), C.byref(d.c) if d else None)
    if rc != 0:
        raise RuntimeError(f"oracle_phase rc={rc}")
    if d is not None and d.c.n_obs > d.c.obs_capacity:      # observation dump did not fit: rerun with the exact size
        d = PhaseDump(reads.n_reads, variants.n, params.connect_adjacent, obs_cap=int(d.c.n_obs) + 16, with_edges=with_edges)
        rc = lib().oracle_phase(C.byref(params), C.byref(variants.c), ref.ctypes.data, ref.size, C.byref(reads.c),
                                C.byref(out.c), C.byref(d.c))
        assert rc == 0
    return out, d


def phase_x(params, variants, extra, ref, reads, dump=False, with_edges=True):
    """The CPU restatement with SV / MOD rows co-phased.  Returns (snp PhaseOut, sv PhaseOut, mod PhaseOut, PhaseDump|None); the dump holds
    indices into the position-sorted union of the three tables."""
    out, osv, omod = abi.PhaseOut(variants.n), abi.PhaseOut(extra.n_sv), abi.PhaseOut(extra.n_mod)
    n_u = variants.n + extra.n_sv + extra.n_mod
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    d = PhaseDump(reads.n_reads, n_u, params.connect_adjacent, with_edges=with_edges) if dump else None
    for _ in range(2):
        rc = lib().oracle_phase_x(C.byref(params), C.byref(variants.c), C.byref(extra.c), ref.ctypes.data, ref.size, C.byref(reads.c),
                                  C.byref(out.c), C.byref(osv.c), C.byref(omod.c), C.byref(d.c) if d else None)
        if rc != 0:
            raise RuntimeError(f"oracle_phase_x rc={rc}")
        if d is None or d.c.n_obs <= d.c.obs_capacity:
            break
        d = PhaseDump(reads.n_reads, n_u, params.connect_adjacent, obs_cap=int(d.c.n_obs) + 16, with_edges=with_edges)
    return out, osv, omod, d


def haplotag(params, variants, ref, reads, votes=None):
    """CPU restatement of the germline haplotag per-read scoring loop.  votes = (h1, h2) int32 arrays per alignment (judgeSVHap).  Returns abi.HaplotagOut."""
    out = abi.HaplotagOut(reads.n_reads)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    v1 = v2 = None
    if votes is not None:
        v1 = np.ascontiguousarray(votes[0], np.int32); v2 = np.ascontiguousarray(votes[1], np.int32)
    rc = lib().oracle_haplotag_v(C.byref(params), C.byref(variants.c), ref.ctypes.data, ref.size, C.byref(reads.c),
                                 None if v1 is None else v1.ctypes.data, None if v2 is None else v2.ctypes.data, C.byref(out.c))
    if rc != 0:
        raise RuntimeError(f"oracle_haplotag rc={rc}")
    return out


def somatic_tag(params, variants, reads):
    """CPU restatement of the somatic_haplotag tagging pass (merged normal+tumor table).  Returns abi.SomaticTagOut."""
    out = abi.SomaticTagOut(reads.n_reads)
    rc = lib().oracle_somatic_tag(C.byref(params), C.byref(variants.c), C.byref(reads.c), C.byref(out.c))
    if rc != 0:
        raise RuntimeError(f"oracle_somatic_tag rc={rc}")
    return out


def somatic_extract_normal(params, variants, ref, reads):
    """CPU restatement of the normal-BAM extraction pass of somatic_haplotag.  Returns abi.SiteCountersOut."""
    out = abi.SiteCountersOut(variants.n, reads.n_reads)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    rc = lib().oracle_somatic_extract_normal(C.byref(params), C.byref(variants.c), ref.ctypes.data, ref.size, C.byref(reads.c), C.byref(out.c))
    if rc != 0:
        raise RuntimeError(f"oracle_somatic_extract_normal rc={rc}")
    return out


def somatic_extract_tumor(params, variants, ref, reads, pair_cap=None, win_cap=None):
    """CPU restatement of the tumor-BAM extraction pass of somatic_haplotag.  Returns abi.TumorExtractOut."""
    pair_cap = pair_cap or 64 * reads.n_reads + 1024
    win_cap = win_cap or 1024 * reads.n_reads + 1024
    out = abi.TumorExtractOut(variants.n, reads.n_reads, pair_cap, win_cap)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    rc = lib().oracle_somatic_extract_tumor(C.byref(params), C.byref(variants.c), ref.ctypes.data, ref.size, C.byref(reads.c), C.byref(out.c))
    if rc == -9:
        return somatic_extract_tumor(params, variants, ref, reads, int(out.c.n_pairs) + 16, int(out.c.n_windows) + 16)
    if rc != 0:
        raise RuntimeError(f"oracle_somatic_extract_tumor rc={rc}")
    return out
