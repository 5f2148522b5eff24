"""ctypes binding of csrc/liblps_hip.so — the product path (hand-written HIP kernels behind include/lps_abi.h).

There is NO CPU fallback: load() raises when the shared library (or any declared symbol) is missing, and
Context() raises when no GPU is visible.
"""
import ctypes as C
import os
import re

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# LPS_HIP_LIB: another BUILD of the same library (profiles/ab.sh compares two builds on one box without touching the in-tree file)
LIB_PATH = os.path.abspath(os.environ.get("LPS_HIP_LIB") or os.path.join(_HERE, "..", "csrc", "liblps_hip.so"))
HEADER = os.path.abspath(os.path.join(_HERE, "..", "..", "include", "lps_abi.h"))

_lib = None


def declared_symbols():
    """Every function include/lps_abi.h declares."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lps_[a-z_0-9]+)\s*\(", txt)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                           "The product path has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    older = set(os.environ.get("LPS_AB_OLDER_BUILD", "").split(","))       # A/B runs only (profiles/ab*.sh): entries a build from an earlier commit lacks
    for sym in declared_symbols():
        if not hasattr(L, sym) and sym not in older:
            raise RuntimeError(f"liblps_hip.so does not export {sym} declared in include/lps_abi.h")
    L.lps_create.restype = C.c_void_p
    L.lps_create.argtypes = [C.c_int, C.POINTER(abi.Params)]
    L.lps_destroy.argtypes = [C.c_void_p]
    L.lps_last_error.restype = C.c_char_p
    L.lps_last_error.argtypes = [C.c_void_p]
    L.lps_begin_chromosome.argtypes = [C.c_void_p]
    L.lps_set_variants.argtypes = [C.c_void_p, C.POINTER(abi.VariantTable)]
    L.lps_set_reference.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.lps_set_extra_variants.argtypes = [C.c_void_p, C.POINTER(abi.ExtraVariantTable)]
    L.lps_bgzf_deflate_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.lps_bgzf_deflate_fetch_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    L.lps_host_alloc.restype = C.c_void_p; L.lps_host_alloc.argtypes = [C.c_size_t]
    L.lps_host_free.restype = None; L.lps_host_free.argtypes = [C.c_void_p]
    L.lps_set_read_votes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.lps_get_extra_result.argtypes = [C.c_void_p, C.POINTER(abi.PhaseResult), C.POINTER(abi.PhaseResult)]
    L.lps_push_reads.argtypes = [C.c_void_p, C.POINTER(abi.ReadBatch)]
    L.lps_push_reads_device.argtypes = [C.c_void_p, C.POINTER(abi.ReadBatch)]
    L.lps_push_bam_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    L.lps_debug_std_sort.restype = None
    L.lps_debug_std_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.lps_debug_std_sort_gpu.restype = C.c_int
    L.lps_debug_std_sort_gpu.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.lps_bgzf_load.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.lps_bgzf_load_fd.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    L.lps_bgzf_read.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    L.lps_bgzf_deflate.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    L.lps_bgzf_deflate_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_double)]
    L.lps_haplotag_write_bgzf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.lps_somatic_write_bgzf.argtypes = L.lps_haplotag_write_bgzf.argtypes
    L.lps_bgzf_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.lps_bgzf_retried.argtypes = [C.c_void_p]
    if hasattr(L, "lps_alloc_ms"):                                   # (A/B runs load builds from before ABI 21 out of csrc/ab/)
        L.lps_alloc_ms.argtypes = [C.c_void_p]; L.lps_alloc_ms.restype = C.c_double
    L.lps_bam_scan.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]
    L.lps_bam_scan_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]
    L.lps_bam_record_tids.argtypes = [C.c_void_p, C.c_void_p]
    L.lps_bam_record_offsets.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    L.lps_bam_names.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.lps_push_bam_resident.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    L.lps_phase_chromosome.argtypes = [C.c_void_p, C.POINTER(abi.PhaseResult)]
    L.lps_phase_chromosome_steps.argtypes = [C.c_void_p, C.POINTER(abi.PhaseResult), C.c_int, C.POINTER(C.c_double)]
    L.lps_haplotag_chromosome_steps.argtypes = [C.c_void_p, C.POINTER(abi.HaplotagResult), C.c_int, C.POINTER(C.c_double)]
    L.lps_haplotag_chromosome.argtypes = [C.c_void_p, C.POINTER(abi.HaplotagResult)]
    L.lps_somatic_tag_chromosome.argtypes = [C.c_void_p, C.POINTER(abi.SomaticTagResult)]
    L.lps_somatic_extract_normal.argtypes = [C.c_void_p, C.POINTER(abi.SiteCounters)]
    L.lps_somatic_extract_tumor.argtypes = [C.c_void_p, C.POINTER(abi.TumorExtractResult)]
    L.lps_get_timings.argtypes = [C.c_void_p, C.POINTER(abi.Timings)]
    L.lps_set_stage_timing.argtypes = [C.c_void_p, C.c_int]
    L.lps_device_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.lps_comm_unique_id.argtypes = [C.c_void_p]
    L.lps_comm_create.restype = C.c_void_p
    L.lps_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.lps_comm_create_all.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.lps_comm_size.argtypes = [C.c_void_p]
    L.lps_comm_rank.argtypes = [C.c_void_p]
    L.lps_comm_destroy.argtypes = [C.c_void_p]
    L.lps_comm_destroy.restype = None
    L.lps_comm_bcast.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double)]
    L.lps_comm_bcast_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double)]
    L.lps_comm_bcast_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_double)]
    L.lps_set_variants_device.argtypes = [C.c_void_p, C.POINTER(abi.VariantTable)]
    L.lps_comm_last_error.restype = C.c_char_p
    L.lps_debug_set_obs_capacity.argtypes = [C.c_void_p, C.c_int64]
    L.lps_stage_name.restype = C.c_char_p
    L.lps_stage_name.argtypes = [C.c_int]
    L.lps_stream.restype = C.c_void_p
    L.lps_stream.argtypes = [C.c_void_p]
    L.lps_dump_observations.restype = C.c_int64
    L.lps_dump_observations.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int64]
    L.lps_dump_graph.restype = C.c_int64
    L.lps_dump_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.lps_dump_votes.restype = C.c_int64
    L.lps_dump_votes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.lps_dump_clips.restype = C.c_int64
    L.lps_dump_clips.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.lps_dump_cnv.restype = C.c_int64
    L.lps_dump_cnv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    _lib = L
    return L


class LpsError(RuntimeError):
    pass


class Context:
    """One lps_ctx = one GPU, one chromosome at a time."""

    def __init__(self, device=0, params=None):
        self.L = load()
        self.params = params or abi.default_params()
        self.h = self.L.lps_create(device, C.byref(self.params))
        if not self.h:
            raise LpsError("lps_create failed (no GPU visible?) - the product path has no CPU fallback")
        self.n_var = 0
        self.n_reads = 0

    def _check(self, rc, what):
        if rc != 0:
            raise LpsError(f"{what} rc={rc}: {self.L.lps_last_error(self.h).decode()}")

    def load_chromosome(self, variants, ref, reads_list):
        """begin_chromosome + set_variants + set_reference + push_reads (H2D, untimed part of the benchmark)."""
        self._check(self.L.lps_begin_chromosome(self.h), "lps_begin_chromosome")
        self._check(self.L.lps_set_variants(self.h, C.byref(variants.c)), "lps_set_variants")
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        self._check(self.L.lps_set_reference(self.h, ref.ctypes.data, ref.size), "lps_set_reference")
        self.n_reads = 0
        for r in (reads_list if isinstance(reads_list, (list, tuple)) else [reads_list]):
            if isinstance(r, abi.BamRecords):
                self._check(self.L.lps_push_bam_records(self.h, r.blob.ctypes.data, r.blob.size, r.rec_off.ctypes.data, r.n_reads,
                                                        r.name_id.ctypes.data), "lps_push_bam_records")
            else:
                self._check(self.L.lps_push_reads(self.h, C.byref(r.c)), "lps_push_reads")
            self.n_reads += r.n_reads
        self.n_var = variants.n

    def load_chromosome_device(self, variants, ref, batch, n_reads, table_dev=None):
        """As load_chromosome for a batch whose arrays already sit on this GPU (abi.ReadBatch of DEVICE pointers).  table_dev = (pos, ref0, alt0)
        device addresses of the contig's `variants.n` rows (e.g. inside the buffer lps_comm_bcast_to_device filled): the table is then taken from
        there (lps_set_variants_device) and the host arrays of `variants` are not uploaded."""
        self._check(self.L.lps_begin_chromosome(self.h), "lps_begin_chromosome")
        if table_dev is None:
            self._check(self.L.lps_set_variants(self.h, C.byref(variants.c)), "lps_set_variants")
        else:
            t = abi.VariantTable()
            t.n = variants.n
            t.pos, t.ref0, t.alt0 = int(table_dev[0]), int(table_dev[1]), int(table_dev[2])
            self._check(self.L.lps_set_variants_device(self.h, C.byref(t)), "lps_set_variants_device")
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        self._check(self.L.lps_set_reference(self.h, ref.ctypes.data, ref.size), "lps_set_reference")
        self._check(self.L.lps_push_reads_device(self.h, C.byref(batch)), "lps_push_reads_device")
        self.n_reads = n_reads
        self.n_var = variants.n

    def set_extra(self, extra):
        """SV / MOD rows co-phased with the SNPs (after load_chromosome / set_table; None = none)."""
        self._check(self.L.lps_set_extra_variants(self.h, C.byref(extra.c) if extra is not None else None), "lps_set_extra_variants")
        self._extra = extra

    def extra_result(self):
        """(sv PhaseOut, mod PhaseOut) of the last run_phase."""
        sv, mod = abi.PhaseOut(self._extra.n_sv), abi.PhaseOut(self._extra.n_mod)
        self._check(self.L.lps_get_extra_result(self.h, C.byref(sv.c), C.byref(mod.c)), "lps_get_extra_result")
        return sv, mod

    def set_read_votes(self, h1, h2):
        """haplotag: per-alignment votes from the phased SV / MOD files (judgeSVHap); None, None = none."""
        if h1 is None:
            self._check(self.L.lps_set_read_votes(self.h, None, None, 0), "lps_set_read_votes"); return
        a = np.ascontiguousarray(h1, np.int32); b = np.ascontiguousarray(h2, np.int32)
        self._check(self.L.lps_set_read_votes(self.h, a.ctypes.data, b.ctypes.data, a.size), "lps_set_read_votes")

    def set_table(self, variants, ref):
        """Replace the variant table (e.g. by the phased one) while the pushed reads stay resident."""
        self._check(self.L.lps_set_variants(self.h, C.byref(variants.c)), "lps_set_variants")
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        self._check(self.L.lps_set_reference(self.h, ref.ctypes.data, ref.size), "lps_set_reference")
        self.n_var = variants.n

    def bgzf_load(self, data):
        """Upload a whole .bam file (bytes / uint8 array) and inflate it on the GPU; returns the inflated size."""
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        n = C.c_int64(0)
        self._check(self.L.lps_bgzf_load(self.h, a.ctypes.data, a.size, C.byref(n)), "lps_bgzf_load")
        return int(n.value)

    def bgzf_load_fd(self, fd, offset, n_bytes):
        """The same for bytes [offset, offset + n_bytes) of an open file (pread into the upload pieces, no mapping); returns the inflated size."""
        n = C.c_int64(0)
        self._check(self.L.lps_bgzf_load_fd(self.h, int(fd), int(offset), int(n_bytes), C.byref(n)), "lps_bgzf_load_fd")
        return int(n.value)

    def bam_scan(self, first_record_offset, n_ref):
        n = C.c_int64(0)
        self._check(self.L.lps_bam_scan(self.h, first_record_offset, n_ref, C.byref(n)), "lps_bam_scan")
        tid = np.empty(n.value, dtype=np.int32)
        self._check(self.L.lps_bam_record_tids(self.h, tid.ctypes.data), "lps_bam_record_tids")
        return tid

    def bam_names(self, first, count):
        nb = C.c_int64(0)
        self._check(self.L.lps_bam_names(self.h, first, count, None, None, 0, C.byref(nb)), "lps_bam_names")
        off = np.empty(count + 1, dtype=np.uint32); buf = np.empty(max(1, nb.value), dtype=np.uint8)
        self._check(self.L.lps_bam_names(self.h, first, count, off.ctypes.data, buf.ctypes.data, buf.size, C.byref(nb)), "lps_bam_names")
        raw = buf.tobytes()
        return [raw[off[i]:off[i + 1] - 1] for i in range(count)]

    def load_resident(self, variants, ref, first, count, name_id):
        """begin_chromosome + set_variants + set_reference + lps_push_bam_resident (records of a BAM inflated on the GPU)."""
        self._check(self.L.lps_begin_chromosome(self.h), "lps_begin_chromosome")
        self._check(self.L.lps_set_variants(self.h, C.byref(variants.c)), "lps_set_variants")
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        self._check(self.L.lps_set_reference(self.h, ref.ctypes.data, ref.size), "lps_set_reference")
        nid = np.ascontiguousarray(name_id, dtype=np.uint32)
        self._check(self.L.lps_push_bam_resident(self.h, first, count, nid.ctypes.data), "lps_push_bam_resident")
        self.n_reads = count; self.n_var = variants.n

    def bgzf_read(self, offset, n):
        out = np.empty(n, dtype=np.uint8)
        self._check(self.L.lps_bgzf_read(self.h, offset, n, out.ctypes.data), "lps_bgzf_read")
        return out

    def bgzf_deflate(self, offset, n_bytes):
        """Deflate a piece of the resident stream into BGZF blocks on the GPU -> (bytes, kernel ms)."""
        nb = C.c_int64(0)
        self._check(self.L.lps_bgzf_deflate(self.h, offset, n_bytes, C.byref(nb)), "lps_bgzf_deflate")
        out = np.empty(max(1, nb.value), dtype=np.uint8); ms = C.c_double(0)
        self._check(self.L.lps_bgzf_deflate_fetch(self.h, out.ctypes.data, out.size, C.byref(ms)), "lps_bgzf_deflate_fetch")
        return out[:nb.value].tobytes(), ms.value

    def bgzf_timings(self):
        a, b = C.c_double(0), C.c_double(0)
        self.L.lps_bgzf_timings(self.h, C.byref(a), C.byref(b))
        return dict(h2d_ms=a.value, inflate_ms=b.value)

    def run_phase(self, out=None):
        out = out or abi.PhaseOut(self.n_var)
        self._check(self.L.lps_phase_chromosome(self.h, C.byref(out.c)), "lps_phase_chromosome")
        return out

    def run_phase_steps(self, out, k):
        """k consecutive lps_phase_chromosome calls without returning to Python in between; -> the calls' wall times in ms"""
        ms = (C.c_double * k)()
        self._check(self.L.lps_phase_chromosome_steps(self.h, C.byref(out.c), k, ms), "lps_phase_chromosome_steps")
        return [float(x) for x in ms]

    def run_haplotag_steps(self, out, k):
        ms = (C.c_double * k)()
        self._check(self.L.lps_haplotag_chromosome_steps(self.h, C.byref(out.c), k, ms), "lps_haplotag_chromosome_steps")
        return [float(x) for x in ms]

    def phase(self, variants, ref, reads):
        self.load_chromosome(variants, ref, reads)
        return self.run_phase()

    def run_haplotag(self, out=None):
        out = out or abi.HaplotagOut(self.n_reads)
        self._check(self.L.lps_haplotag_chromosome(self.h, C.byref(out.c)), "lps_haplotag_chromosome")
        return out

    def haplotag(self, variants, ref, reads):
        self.load_chromosome(variants, ref, reads)
        return self.run_haplotag()

    def run_somatic_tag(self, out=None):
        out = out or abi.SomaticTagOut(self.n_reads)
        self._check(self.L.lps_somatic_tag_chromosome(self.h, C.byref(out.c)), "lps_somatic_tag_chromosome")
        return out

    def somatic_tag(self, variants, ref, reads):
        self.load_chromosome(variants, ref, reads)
        return self.run_somatic_tag()

    def somatic_extract_normal(self, variants, ref, reads):
        self.load_chromosome(variants, ref, reads)
        out = abi.SiteCountersOut(self.n_var, self.n_reads)
        self._check(self.L.lps_somatic_extract_normal(self.h, C.byref(out.c)), "lps_somatic_extract_normal")
        return out

    def somatic_extract_tumor(self, variants, ref, reads, pair_cap=None, win_cap=None):
        self.load_chromosome(variants, ref, reads)
        pair_cap = pair_cap or 64 * self.n_reads + 1024
        win_cap = win_cap or 256 * self.n_reads + 1024
        for _ in range(2):
            out = abi.TumorExtractOut(self.n_var, self.n_reads, pair_cap, win_cap)
            rc = self.L.lps_somatic_extract_tumor(self.h, C.byref(out.c))
            if rc == -9:      # lists did not fit: retry with the sizes the library reported
                pair_cap, win_cap = int(out.c.n_pairs) + 16, int(out.c.n_windows) + 16
                continue
            self._check(rc, "lps_somatic_extract_tumor")
            return out
        self._check(rc, "lps_somatic_extract_tumor")

    def set_stage_timing(self, level):
        self._check(self.L.lps_set_stage_timing(self.h, int(level)), "lps_set_stage_timing")

    def timings(self):
        t = abi.Timings()
        self._check(self.L.lps_get_timings(self.h, C.byref(t)), "lps_get_timings")
        names = [self.L.lps_stage_name(i).decode() for i in range(t.n_stages)]
        return dict(stages={n: t.ms_kernel[i] for i, n in enumerate(names)}, ms_total=t.ms_total, n_obs=t.n_obs,
                    n_nodes=t.n_nodes, n_pairs=t.n_pairs, n_reads_used=t.n_reads_used,
                    n_scan_segments=t.n_scan_segments, n_scan_replayed=t.n_scan_replayed,
                    algorithmic_bytes={n: t.algorithmic_bytes[i] for i, n in enumerate(names)})

    # ---- stage dumps (parity tests)
    def dump_observations(self):
        n = self.L.lps_dump_observations(self.h, None, None, None, None, 0)
        if n < 0:
            raise LpsError("lps_dump_observations failed")
        cnt = np.zeros(self.n_reads, np.int32)
        var = np.zeros(max(n, 1), np.int32)
        al = np.zeros(max(n, 1), np.int8)
        q = np.zeros(max(n, 1), np.int16)
        self.L.lps_dump_observations(self.h, cnt.ctypes.data, var.ctypes.data, al.ctypes.data, q.ctypes.data, n)
        return cnt, var[:n], al[:n], q[:n]

    def dump_graph(self, with_edges=True):
        n = self.L.lps_dump_graph(self.h, None, None, 0)
        nodes = np.zeros(max(n, 1), np.int32)
        A = self.params.connect_adjacent
        edge = np.zeros((max(n, 1), A, 4), np.float32) if with_edges else None
        self.L.lps_dump_graph(self.h, nodes.ctypes.data, edge.ctypes.data if with_edges else None, n)
        return nodes[:n], (edge[:n] if with_edges else None)

    def dump_votes(self):
        n = self.L.lps_dump_votes(self.h, None, None, 0)
        hp = np.zeros(max(n, 1), np.int8)
        blk = np.zeros(max(n, 1), np.int32)
        self.L.lps_dump_votes(self.h, hp.ctypes.data, blk.ctypes.data, n)
        return hp[:n], blk[:n]

    def dump_clips(self):
        n = self.L.lps_dump_clips(self.h, None, None, 0)
        pos = np.zeros(max(n, 1), np.int32)
        fb = np.zeros(max(n, 1), np.uint8)
        self.L.lps_dump_clips(self.h, pos.ctypes.data, fb.ctypes.data, n)
        return pos[:n], fb[:n]

    def dump_cnv(self):
        n = int(self.L.lps_dump_cnv(self.h, None, None, 0, None))
        s = np.zeros(max(n, 1), np.int32)
        e = np.zeros(max(n, 1), np.int32)
        d = np.zeros(max(self.n_reads, 1), np.uint8)
        self.L.lps_dump_cnv(self.h, s.ctypes.data, e.ctypes.data, n, d.ctypes.data)
        return s[:n], e[:n], d[:self.n_reads]

    def close(self):
        if self.h:
            self.L.lps_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
