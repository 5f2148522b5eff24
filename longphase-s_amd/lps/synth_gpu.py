"""ctypes binding of tools/liblps_synth_gpu.so — the seeded synthetic contig generated ON THE GPU.

Test + bench infrastructure only (never imported by the product path).  `SynthGpu` owns DEVICE arrays laid out as
include/lps_abi.h's lps_read_batch; `device_batch()` hands them to lps_push_reads_device, `to_host()` copies them
back (numpy) for the oracle, `write_*` produce FASTA / VCF / SAM for the reference binary.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "..", "tools", "liblps_synth_gpu.so")


class SgParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("contig_len", C.c_int64), ("n_snp", C.c_int32), ("clip_every", C.c_int32), ("coverage", C.c_double),
        ("len_median", C.c_double), ("len_sigma", C.c_double), ("len_min", C.c_int32), ("len_max", C.c_int32),
        ("sub_rate", C.c_double), ("ins_rate", C.c_double), ("del_rate", C.c_double), ("lowq_frac", C.c_double),
        ("mapq0_frac", C.c_double), ("secondary_frac", C.c_double), ("dup_frac", C.c_double), ("supp_frac", C.c_double),
        ("supp_overlap_frac", C.c_double), ("hpoly_every", C.c_int32), ("clip_pileups", C.c_int32),
        ("snp_in_hpoly_frac", C.c_double), ("snp_pair_frac", C.c_double), ("read_seed", C.c_uint64),
    ]


# array ids of sg_dev_ptr / sg_copy_to_host (enum in lps_synth_gpu.hip)
ARRAYS = dict(ref=(0, np.uint8), var_pos=(1, np.int32), var_ref0=(2, np.uint8), var_alt0=(3, np.uint8), var_hap=(4, np.uint8),
              ref_start=(5, np.int32), l_qseq=(6, np.int32), flag=(7, np.uint16), mapq=(8, np.uint8), name_id=(9, np.uint32),
              read_hap=(10, np.uint8), cigar_off=(11, np.uint64), seq_off=(12, np.uint64), qual_off=(13, np.uint64),
              cigar=(14, np.uint32), seq=(15, np.uint8), qual=(16, np.uint8))

_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise RuntimeError(f"{_LIB} missing - run __graft_entry__.build()")
        L = C.CDLL(_LIB)
        L.sg_default_params.argtypes = [C.POINTER(SgParams)]
        L.sg_create.restype = C.c_void_p
        L.sg_create.argtypes = [C.c_int, C.POINTER(SgParams)]
        L.sg_destroy.argtypes = [C.c_void_p]
        L.sg_release_reads.argtypes = [C.c_void_p]
        for n in ("sg_n_reads", "sg_n_variants", "sg_n_cigar", "sg_n_seq", "sg_n_qual"):
            getattr(L, n).restype = C.c_int64
            getattr(L, n).argtypes = [C.c_void_p]
        L.sg_gen_ms.restype = C.c_double
        L.sg_gen_ms.argtypes = [C.c_void_p]
        L.sg_array_bytes.restype = C.c_int64
        L.sg_array_bytes.argtypes = [C.c_void_p, C.c_int]
        L.sg_dev_ptr.restype = C.c_void_p
        L.sg_dev_ptr.argtypes = [C.c_void_p, C.c_int]
        L.sg_copy_to_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.sg_write_fasta.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.sg_write_vcf.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        L.sg_write_sam.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        _lib = L
    return _lib


class HostCopy:
    """numpy copy of a generated contig; attribute names follow lps.synth.Synth so abi.Reads.from_synth works."""


class SynthGpu:
    def __init__(self, device=0, **kw):
        L = _load()
        p = SgParams()
        L.sg_default_params(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        self.params = p
        self.device = device
        self._h = L.sg_create(device, C.byref(p))
        if not self._h:
            raise RuntimeError("sg_create failed (see stderr)")
        self.n_reads = int(L.sg_n_reads(self._h))
        self.n_variants = int(L.sg_n_variants(self._h))
        self.n_cigar = int(L.sg_n_cigar(self._h))
        self.n_bases = int(L.sg_n_qual(self._h))        # incl. the <8 padding bytes per alignment
        self.contig_len = int(p.contig_len)
        self.gen_ms = float(L.sg_gen_ms(self._h))
        self._keep = None

    def dev_ptr(self, name):
        return _load().sg_dev_ptr(self._h, ARRAYS[name][0])

    def host(self, name):
        wid, dt = ARRAYS[name]
        nb = int(_load().sg_array_bytes(self._h, wid))
        out = np.empty(nb // np.dtype(dt).itemsize, dtype=dt)
        if nb:
            assert _load().sg_copy_to_host(self._h, wid, out.ctypes.data) == 0, name
        return out

    def device_batch(self):
        """lps_read_batch whose pointers are DEVICE pointers (for lps_push_reads_device)."""
        return abi.ReadBatch(self.n_reads, self.dev_ptr("ref_start"), self.dev_ptr("flag"), self.dev_ptr("mapq"), self.dev_ptr("l_qseq"),
                             self.dev_ptr("name_id"), self.dev_ptr("cigar_off"), self.dev_ptr("cigar"), self.dev_ptr("seq_off"),
                             self.dev_ptr("seq"), self.dev_ptr("qual_off"), self.dev_ptr("qual"))

    def variants(self, **kw):
        """abi.Variants of the het SNP table (host)."""
        return abi.Variants.from_snps(self.host("var_pos"), self.host("var_ref0"), self.host("var_alt0"), **kw)

    def to_host(self, reads=True):
        h = HostCopy()
        h.ref = self.host("ref")
        h.var_pos = self.host("var_pos"); h.var_ref0 = self.host("var_ref0"); h.var_alt0 = self.host("var_alt0"); h.var_hap = self.host("var_hap")
        h.n_variants = self.n_variants; h.n_reads = self.n_reads; h.contig_len = self.contig_len
        if reads:
            for n, _ in abi.Reads.FIELDS:
                setattr(h, n, self.host(n))
            h.read_hap = self.host("read_hap")
        return h

    def release_reads(self):
        """Free the per-base device arrays (after the library has taken its copy)."""
        _load().sg_release_reads(self._h)

    def write_fasta(self, path, chrom="chrS"):
        assert _load().sg_write_fasta(self._h, path.encode(), chrom.encode()) == 0

    def write_vcf(self, path, chrom="chrS", phased=False):
        assert _load().sg_write_vcf(self._h, path.encode(), chrom.encode(), int(phased)) == 0

    def write_sam(self, path, chrom="chrS", threads=8):
        assert _load().sg_write_sam(self._h, path.encode(), chrom.encode(), int(threads)) == 0

    def close(self):
        if self._h:
            _load().sg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
