#!/usr/bin/env python3
"""bench.py — throughput of the `phase` hot path on MI355X (BASELINE.json metric: het SNPs phased / s, 50x ONT whole genome).

Default workload `wgs_50x` = BASELINE.json's headline configuration (configs[2]/[3]): germline `phase` over 24 contigs of GRCh38 length,
50x synthetic ONT reads, ~4 M het SNPs, ~155 Gbases.  The genome is streamed contig by contig, as the reference does (one chromosome per
worker, src/phase/PhasingProcess.cpp:113-173): a contig's decoded alignments are generated in HBM (tools/lps_synth_gpu.hip), handed to the
library (lps_push_reads_device), and then

    a STEP = one pass of the hot path over the whole genome = one lps_phase_chromosome() per contig,

everything from allele extraction to phased genotypes recomputed from the decoded reads as the push left them in HBM, results back in host
memory.  K steps are run as K consecutive calls per contig while that contig is resident (inputs in HBM when its timed region starts); `value` =
phased SNPs of all contigs x K / sum of the per-contig timed regions.  ONE-PASS CLOCK: the library keeps nothing between a push and a call or
from one call to the next (since ABI 20: no lps_prepare_reads, no re-laid copy of bases, qualities or CIGAR words; `prepare_ms` on the line is 0 by
construction), so every timed call does what the first call on a freshly loaded chromosome does - the reference phases each chromosome once
(src/phase/PhasingProcess.cpp:113-173).  What a first call additionally pays is the growth of the context's device buffers (hipMalloc); the
line carries it as `first_call_pass_ms` (the un-warmed first call of every contig, contexts side by side like the timed ones).

  python bench.py --gpus N --steps K --warmup W
N > 1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling - the 24 contigs are dealt longest-first onto the ranks
(SURVEY.md §8e), rank 0 broadcasts the packed SNP table to the other GPUs with RCCL (lps_comm_bcast_device -> ncclBroadcast over xGMI),
then no data-path collective; time = max over ranks of their summed timed regions, value = all phased SNPs x K / that.
torch.distributed (gloo) is the control plane only (unique id, barrier, max/sum of scalars).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel at the largest contig), `cpu_baseline` (the reference binary on one whole 50x
contig) and `parity_checked` (step-0 output of every benched contig compared with the oracle, outside the timed regions).
"""
import argparse
import concurrent.futures as cf
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "longphase-s_amd"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec

GRCH38 = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259), ("chr6", 170805979),
          ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
          ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285),
          ("chr19", 58617616), ("chr20", 64444167), ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415)]
WGS_SNPS = 4_000_000


def wgs_contigs(seed, coverage=50.0):
    total = sum(n for _, n in GRCH38)
    return [dict(name=c, contig_len=n, n_snp=int(round(WGS_SNPS * n / total)), coverage=coverage, seed=seed + i) for i, (c, n) in enumerate(GRCH38)]


def workload_contigs(name, seed):
    if name == "wgs_50x":          # BASELINE.json configs[2]/[3] (the configuration the metric is quoted on)
        return wgs_contigs(seed)
    if name == "chr1_50x":         # the largest contig of it alone (profiling runs)
        return wgs_contigs(seed)[:1]
    if name == "chr20_30x":        # BASELINE.json configs[1]
        return [dict(name="chr20", contig_len=64_444_167, n_snp=60_000, coverage=30.0, seed=seed)]
    if name == "chr20_30x_pileups":   # configs[1] with simulated CNV break points: clip pile-ups, so that the CNV interval stage and filter run
        return [dict(name="chr20", contig_len=64_444_167, n_snp=60_000, coverage=30.0, seed=seed, clip_pileups=50)]
    if name == "mini_wgs":         # eight small contigs: rehearsal of the multi-rank path (tests/test_bench_ranks_gpu.py), not a measurement
        return [dict(name=f"ctg{i}", contig_len=(6 - i % 3) * 1_000_000, n_snp=(6 - i % 3) * 1_000, coverage=12.0, seed=seed + i) for i in range(8)]
    if name == "5mb_10x":          # BASELINE.json configs[0] (plumbing size)
        return [dict(name="ctg5mb", contig_len=5_000_000, n_snp=5_000, coverage=10.0, seed=seed)]
    raise SystemExit(f"unknown workload {name}")


def lpt(contigs, world):
    """Longest-processing-time-first deal of the contigs onto `world` ranks (lps/shard.py does the same for the CLI)."""
    from lps import shard
    return shard.lpt_schedule([c["contig_len"] for c in contigs], world)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class ParityPool:
    """Oracle (CPU restatement) runs on host threads beside the GPU work; bounded by the host bytes in flight."""

    def __init__(self, workers, max_bytes=96 << 30):
        self.ex = cf.ThreadPoolExecutor(max_workers=workers)
        self.futs = []
        self.bytes = 0
        self.max_bytes = max_bytes
        self.cv = threading.Condition()
        # Held by a checker thread while it hands a contig's host copy (up to 22 GB) back to the OS, and by the main thread for every timed region:
        # an munmap of that size stops every thread of the process that needs the address-space lock (the HIP runtime's included) for ~0.5 s
        self.heavy = threading.Lock()

    def submit(self, name, P, V, host, out_ps, out_gt, hap=None):
        """hap = (VT, hp, pq, ps): the haplotag table of the benched run and the tags the GPU gave every alignment, checked against lps_oracle.haplotag"""
        from lps import abi
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import lps_oracle
        import numpy as np
        nb = int(host.qual.nbytes + host.seq.nbytes + host.cigar.nbytes)
        with self.cv:
            while self.bytes and self.bytes + nb > self.max_bytes:
                self.cv.wait()
            self.bytes += nb

        box = [host]; del host                                           # the checker thread holds the only reference once the caller dropped its own

        def job():
            R = None
            try:
                R = abi.Reads.from_synth(box[0])
                t0 = time.time()
                want, _ = lps_oracle.phase(P, V, box[0].ref, R)
                dt = time.time() - t0
                same_ps = bool(np.array_equal(want.phase_set, out_ps))
                m = want.phase_set != 0
                same_gt = bool(np.array_equal(want.gt[m], out_gt[m]))
                res = dict(contig=name, identical=same_ps and same_gt, oracle_s=dt, n_phased=int(m.sum()))
                if hap is not None:                                       # secondary metric: HP / PQ / PS of every alignment (HaplotagStrategy.cpp:243-300)
                    t0 = time.time()
                    hw = lps_oracle.haplotag(P, hap[0], box[0].ref, R)
                    res.update(haplotag_identical=bool(np.array_equal(hw.hp, hap[1]) and np.array_equal(hw.pq, hap[2]) and np.array_equal(hw.ps, hap[3])),
                               haplotag_oracle_s=time.time() - t0, n_tagged=int((hw.hp != 0).sum()))
                return res
            finally:
                with self.heavy:
                    R = None; box.clear()                                # the arrays are freed HERE, never while a timed region runs
                with self.cv:
                    self.bytes -= nb
                    self.cv.notify_all()
        self.futs.append(self.ex.submit(job))

    def results(self):
        return [f.result() for f in self.futs]


def pin(arr):
    """hipHostRegister a numpy array (P clock: decoded batches in pinned host memory)."""
    hiprt = C.CDLL("libamdhip64.so")
    hiprt.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    return hiprt.hipHostRegister(arr.ctypes.data, arr.nbytes, 0) == 0


def unpin(arr):
    """hipHostUnregister: a registered range must be released before its memory goes back to the allocator (a later array that reuses part of it
    would make every copy from it fail)."""
    hiprt = C.CDLL("libamdhip64.so")
    hiprt.hipHostUnregister.argtypes = [C.c_void_p]
    return hiprt.hipHostUnregister(arr.ctypes.data) == 0


def cpu_baseline(g, spec, threads, port, gpu_result=None):
    """The reference binary (oracle/_ref/longphase-s-ref: BAM + VCF + FASTA in, phased VCF out, incl. BGZF/BAM decode) on ONE WHOLE contig
    of the benched workload; kind 'port' (the oracle restatement on decoded arrays, 1 thread) when the binary is not there."""
    sample = f"whole contig {spec['name']} of the benched workload ({spec['contig_len']} bp, {spec['coverage']:.0f}x, {g.n_reads} alignments, {g.n_variants} het SNPs)"
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "longphase-s-ref")
    tv = os.path.join(ROOT, "oracle", "_ref", "test_view")
    out = dict(port) if port else {}
    out.update(kind="port", cores=1, sample=sample + "; oracle restatement on decoded arrays in memory (no BAM/BGZF decode), one thread")
    if not (os.path.exists(ref_bin) and os.path.exists(tv)):
        return out
    try:
        with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
            t0 = time.time()
            g.write_fasta(d + "/ref.fa", spec["name"]); g.write_vcf(d + "/in.vcf", spec["name"]); g.write_sam(d + "/reads.sam", spec["name"], threads)
            t1 = time.time()
            subprocess.check_call([tv, "-@", str(threads), "-b", "-x", "reads.bam.bai", "-p", "reads.bam", "reads.sam"], cwd=d, stdout=subprocess.DEVNULL)
            os.remove(d + "/reads.sam")
            log(f"cpu_baseline: files written in {t1-t0:.1f}s, BAM + index in {time.time()-t1:.1f}s ({os.path.getsize(d + '/reads.bam')/1e9:.2f} GB)")
            cmd = [ref_bin, "phase", "-s", "in.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(threads), "-o", "out", "--ont"]
            ts = []
            for _ in range(2):                       # first run also warms the page cache
                t0 = time.time(); r = subprocess.run(cmd, cwd=d, capture_output=True); ts.append(time.time() - t0)
                assert r.returncode == 0, r.stderr[-500:]
            n_ph = 0; same = True; vi = 0
            for ln in open(d + "/out.vcf"):                      # the reference's phased VCF against the GPU result of the same contig, row by row
                if ln.startswith("#"):
                    continue
                f = ln.rstrip("\n").split("\t"); smp = f[9].split(":")
                ps = 0 if smp[-1] == "." else int(smp[-1])
                n_ph += ps != 0
                if gpu_result is not None:
                    same = same and int(f[1]) - 1 == int(gpu_result[0][vi]) and ps == int(gpu_result[1][vi]) and (ps == 0 or smp[0] == ("1|0" if gpu_result[2][vi] else "0|1"))
                vi += 1
            same = same and (gpu_result is None or vi == len(gpu_result[0]))
            stages = [ln.strip() for ln in r.stderr.decode(errors="replace").splitlines() if "total" in ln.lower() or "parsing" in ln.lower()][-4:]
        return dict(value=float(n_ph / min(ts)), unit="SNPs/s", cores=threads, kind="reference",
                    sample=sample + f"; best of 2 runs of `longphase-s phase -t {threads} --ont` end to end (one contig => one compute thread, the others feed BGZF)",
                    wall_s=round(min(ts), 2), n_phased=n_ph, identical_to_gpu_result=(bool(same) if gpu_result is not None else None), reference_stage_lines=stages,
                    port_value=port.get("value") if port else None, port_note="oracle restatement, decoded arrays in memory, one thread")
    except Exception as e:  # noqa: BLE001
        log("cpu_baseline: reference run failed, reporting the port:", repr(e)[:300])
        return out


def bam_record_digest(path):
    """sha256 of the inflated record stream of a BAM (everything behind the header and the reference table)"""
    import gzip, hashlib
    h = hashlib.sha256()
    with gzip.open(path, "rb") as f:
        head = f.read(8); f.read(int.from_bytes(head[4:8], "little")); nref = int.from_bytes(f.read(4), "little")
        for _ in range(nref):
            ln = int.from_bytes(f.read(4), "little"); f.read(ln + 4)
        for b in iter(lambda: f.read(1 << 24), b""):
            h.update(b)
    return h.hexdigest()


def e2e_haplotag(d, ref_bin, cli, threads, dev):
    import shutil
    need = 2.5 * os.path.getsize(d + "/reads.bam")
    if shutil.disk_usage(d).free < need:
        return dict(skipped="less than %.0f GB free in %s" % (need / 1e9, d))
    rcmd = [ref_bin, "haplotag", "-s", "out.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(threads), "-o", "tag_ref"]
    t0 = time.time(); r = subprocess.run(rcmd, cwd=d, capture_output=True); t_ref = time.time() - t0
    assert r.returncode == 0, r.stderr[-500:]
    want = bam_record_digest(d + "/tag_ref.bam"); os.remove(d + "/tag_ref.bam")
    gcmd = [cli, "haplotag", "-s", "out.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(threads), "-o", "tag_gpu", "--gpu", str(dev)]
    gs = []
    for _ in range(2):
        t0 = time.time(); g = subprocess.run(gcmd, cwd=d, capture_output=True); gs.append(time.time() - t0)
        assert g.returncode == 0, g.stderr[-500:]
    same = want == bam_record_digest(d + "/tag_gpu.bam"); os.remove(d + "/tag_gpu.bam")
    return dict(reference_wall_s=round(t_ref, 2), wall_s=round(min(gs), 3), runs_s=[round(x, 3) for x in gs], over_cpu=round(t_ref / min(gs), 2), identical_record_stream=same,
                stage_line=g.stderr.decode(errors="replace").strip().splitlines()[-1][:700],
                note="`haplotag -t %d` of the reference against `longphase_amd haplotag` on the same BAM + phased VCF: file -> tagged BAM" % threads)


def whole_node_baseline(dev, P, threads, seed, n_contigs=16, contig_mb=10):
    """The reference's only parallelism is its chromosome loop (src/phase/PhasingProcess.cpp:106,113), so a single contig keeps ONE of its threads
    computing.  Here every thread gets a contig: `n_contigs` contigs of `contig_mb` Mb at 50x with the genome's SNP density in ONE BAM, phased by
    `longphase-s phase -t threads` (best of 2 runs, the first warms the page cache) - the whole-node CPU rate - and by the GPU from the same
    decoded alignments in PINNED host memory (clock P: H2D + layout + phase, the load of one contig overlapping the phase of another: two
    contexts, two host threads), every contig's result compared with the reference's VCF."""
    import numpy as np
    from lps import abi, hip
    from lps.synth_gpu import SynthGpu
    n_contigs = max(2, min(24, n_contigs))
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "longphase-s-ref"); tv = os.path.join(ROOT, "oracle", "_ref", "test_view")
    if not (os.path.exists(ref_bin) and os.path.exists(tv)):
        return None
    total = sum(n for _, n in GRCH38)
    L = contig_mb * 1_000_000; n_snp = int(round(WGS_SNPS * L / total))
    t_prep = time.time()
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        names = ["ctg%02d" % (k + 1) for k in range(n_contigs)]
        header = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{nm}\tLN:{L}\n" for nm in names)
        bam = subprocess.Popen([tv, "-@", str(threads), "-b", "-x", "reads.bam.bai", "-p", "reads.bam", "-"], cwd=d, stdin=subprocess.PIPE, stdout=subprocess.DEVNULL)
        bam.stdin.write(header.encode()); bam.stdin.flush()
        vhead, contigs = [], []
        with open(d + "/ref.fa", "wb") as fa, open(d + "/in.vcf.body", "wb") as vb:
            for k, nm in enumerate(names):
                g = SynthGpu(dev, contig_len=L, n_snp=n_snp, coverage=50.0, seed=seed + 1000 + k)
                g.write_fasta(d + "/one.fa", nm); fa.write(open(d + "/one.fa", "rb").read())
                g.write_vcf(d + "/one.vcf", nm)
                for line in open(d + "/one.vcf", "rb"):
                    if line.startswith(b"##contig"):
                        vhead.append(line)
                    elif not line.startswith(b"#"):
                        vb.write(line)
                g.write_sam(d + "/one.sam", nm, threads)
                subprocess.check_call(["grep", "-v", "^@", "one.sam"], cwd=d, stdout=bam.stdin)      # the records of this contig behind the common header
                os.remove(d + "/one.sam")
                contigs.append(dict(name=nm, V=g.variants(), ref=g.host("ref"), host=g.to_host(), n_reads=g.n_reads, n_bases=g.n_bases))
                g.close()
        bam.stdin.close()
        assert bam.wait() == 0, "test_view failed"
        with open(d + "/in.vcf", "wb") as f:
            f.write(b"##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n" + b"".join(vhead) +
                    b'##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n')
            f.write(open(d + "/in.vcf.body", "rb").read())
        prep_s = time.time() - t_prep
        log(f"whole-node sample: {n_contigs} x {contig_mb} Mb at 50x, BAM {os.path.getsize(d + '/reads.bam') / 1e9:.2f} GB, prepared in {prep_s:.0f}s")
        t_sync = time.time(); os.sync()          # the 8 GB just written are flushed before anything is timed (the files stay in the page cache): both
        log(f"whole-node sample: sync {time.time() - t_sync:.1f}s")     # sides then run without the write-back going on beside them
        cmd = [ref_bin, "phase", "-s", "in.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(threads), "-o", "out", "--ont"]
        ts = []
        for _ in range(2):
            t0 = time.time(); r = subprocess.run(cmd, cwd=d, capture_output=True); ts.append(time.time() - t0)
            assert r.returncode == 0, r.stderr[-500:]
        want = {}                                                         # contig -> {pos0: (ps, "a|b")}
        for ln in open(d + "/out.vcf"):
            if ln.startswith("#"):
                continue
            f = ln.rstrip("\n").split("\t"); smp = f[9].split(":")
            want.setdefault(f[0], {})[int(f[1]) - 1] = (0 if smp[-1] == "." else int(smp[-1]), smp[0])
        # ---- clock E: this repository's command line on the SAME files (BGZF blocks uploaded, inflated, scanned, decoded and phased on the GPU,
        #      the VCF written by the host), against the reference's wall time above; the output must be the reference's, line for line
        e_clock = None
        cli = os.path.join(ROOT, "longphase-s_amd", "cli", "longphase_amd")
        if os.path.exists(cli):
            try:
                gcmd = [cli, "phase", "-s", "in.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(threads), "-o", "outg", "--ont", "--gpu", str(dev)]
                es = []; best_err = b""
                for _ in range(6):      # (a second each; on some boxes every other run behind the reference's threads and the BAM's write-back takes twice as long: best of six)
                    t0 = time.time(); r = subprocess.run(gcmd, cwd=d, capture_output=True); es.append(time.time() - t0)
                    assert r.returncode == 0, r.stderr[-500:]
                    if es[-1] == min(es):
                        best_err = r.stderr                           # (the stage line reported is the best run's)
                body = lambda p: [ln for ln in open(p) if not ln.startswith("##commandline=") and not ln.startswith("##longphaseVersion=")]  # noqa: E731
                sweep = []                                               # (profiles/e2e_whole_node.py: other settings of the command line on the same files)
                for extra in json.loads(os.environ.get("LPS_E2E_SWEEP", "[]")):
                    xs = []; xenv = dict(os.environ); xargs = []          # (an item NAME=value sets the environment of the run instead of an argument)
                    for x in extra:
                        if "=" in x and not x.startswith("-"):
                            xenv[x.split("=", 1)[0]] = x.split("=", 1)[1]
                        else:
                            xargs.append(x)
                    for _ in range(3):
                        t0 = time.time(); rx = subprocess.run(gcmd + xargs, cwd=d, capture_output=True, env=xenv); xs.append(time.time() - t0); spawn = (round(t0, 3), round(time.time(), 3))
                    err_lines = rx.stderr.decode(errors="replace").strip().splitlines()
                    sweep.append(dict(args=extra, wall_s=round(min(xs), 3), rc=rx.returncode, last_run_spawned_and_reaped_at=spawn, stage_line=([ln for ln in err_lines if "| total " in ln] or err_lines[-1:])[-1][:400], debug=[ln[ln.find("["):][:400] for ln in err_lines if "[lps_" in ln or "[cli]" in ln]))
                e_clock = dict(sweep=sweep, wall_s=round(min(es), 3), runs_s=[round(x, 3) for x in es], vcf_identical_to_reference=body(d + "/out.vcf") == body(d + "/outg.vcf"),
                               over_cpu=round(min(ts) / min(es), 2), stage_line=([ln for ln in best_err.decode(errors="replace").strip().splitlines() if "| total " in ln] or [""])[-1][:600],
                               note="`longphase_amd phase` end to end on the same BAM + VCF + FASTA (page cache warm, like the reference's best run): file -> phased VCF, GPU start-up included")
            except Exception as e:  # noqa: BLE001
                log("whole-node sample: the command line run failed:", repr(e)[:300])
            # ---- the secondary path end to end on the same files (profiles/e2e_whole_node.py only: the reference takes a minute): `haplotag` of the
            #      8 GB BAM against the VCF the reference just phased, tagged BAM out; the record streams must be identical
            if e_clock is not None and os.environ.get("LPS_E2E_HAPLOTAG"):
                try:
                    e_clock["haplotag"] = e2e_haplotag(d, ref_bin, cli, threads, dev)
                except Exception as e:  # noqa: BLE001
                    log("whole-node sample: the haplotag run failed:", repr(e)[:300])
    # ---- the GPU on the same decoded alignments, from pinned host memory (clock P), loads and phases of different contigs overlapping
    for c in contigs:
        c["R"] = abi.Reads.from_synth(c["host"])
        c["pinned"] = all(pin(x) for x in (c["host"].qual, c["host"].seq, c["host"].cigar))
        c["out"] = abi.PhaseOut(c["V"].n)
    ctxs = [hip.Context(dev, P) for _ in range(2)]
    for cx, c in zip(ctxs, contigs):                                      # warm-up: buffers of both contexts at their working size
        cx.load_chromosome(c["V"], c["ref"], c["R"]); cx.run_phase(c["out"])
    def lane(i):
        for c in contigs[i::2]:
            ctxs[i].load_chromosome(c["V"], c["ref"], c["R"]); ctxs[i].run_phase(c["out"])
    th = [threading.Thread(target=lane, args=(i,)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    p_wall = time.perf_counter() - t0
    for cx in ctxs:
        cx.close()
    for c in contigs:
        for x in (c["host"].qual, c["host"].seq, c["host"].cigar):
            unpin(x)
    same = True; n_ph = 0; n_ref = 0
    for c in contigs:
        w = want.get(c["name"], {})
        pos = c["V"].pos; ps = c["out"].phase_set; gt = c["out"].gt
        n_ph += int((ps != 0).sum()); n_ref += sum(1 for v in w.values() if v[0])
        ok = len(w) == len(pos) and all(w[int(pos[i])][0] == int(ps[i]) and (ps[i] == 0 or w[int(pos[i])][1] == ("1|0" if gt[i] else "0|1")) for i in range(len(pos)))
        same = same and ok
    cpu = n_ref / min(ts); gpu_p = n_ph / p_wall
    return dict(value=float(cpu), unit="SNPs/s", cores=threads, kind="reference",
                sample=f"{n_contigs} contigs x {contig_mb} Mb at 50x ({sum(c['n_reads'] for c in contigs)} alignments, {sum(c['n_bases'] for c in contigs) / 1e9:.1f} Gbases, {n_ref} phased SNPs) in one BAM: "
                       f"`longphase-s phase -t {threads} --ont` end to end, every thread with a contig of its own; best of 2 runs",
                wall_s=round(min(ts), 2), runs_s=[round(x, 2) for x in ts], prepare_s=round(prep_s, 1), identical_to_gpu_result_all_contigs=bool(same),
                p_clock=dict(value=float(gpu_p), unit="SNPs/s", wall_s=round(p_wall, 3), pinned_host_memory=all(c["pinned"] for c in contigs),
                             note="decoded alignments in pinned host memory -> results in host memory: lps_set_variants + lps_set_reference + lps_push_reads (H2D, CIGAR words lane-chunked as they arrive) + "
                                  "lps_phase_chromosome per contig, two contexts on one GPU so that one contig's load overlaps another's phase; PCIe-inclusive, never `value`"),
                e_clock=e_clock,
                vs_baseline=dict(p_clock_over_cpu=round(gpu_p / cpu, 2), note="clock P (what SURVEY.md 8d judges the >= 20x target by) over the whole-node reference rate on the same sample; "
                                 "the top-level vs_baseline stays null: BASELINE.md holds no published number"))


def somatic_leg(dev, mb, steps, threads, seed=301, check=True):
    """BASELINE.json configs[4] on one contig of `mb` Mb: a tumor / normal pair (50x / 25x, SNP + indel VCFs, 60 % simulated purity).  The normal sample
    is phased on the GPU (prelude), then the THREE per-read passes of somatic_haplotag run from the decoded alignments resident in HBM - normal
    extraction (a20, SomaticVarCaller.cpp:123-293), tumor extraction (a21, :334-759) and tagging (a22, SomaticHaplotagProcess.cpp:310-527) - each
    timed as K consecutive calls that recompute everything (one-pass clock: nothing is kept between calls) and, with `check`, compared with the
    oracle: every per-site counter, per-read count, list entry and tag.  Rate = tumor alignments / (the three passes' times)."""
    import numpy as np
    from lps import abi, hip
    from lps.synth import Synth
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lps_oracle
    total = sum(n for _, n in GRCH38)
    L = int(mb * 1_000_000)
    genome = dict(seed=seed, contig_len=L, n_snp=int(round(WGS_SNPS * L / total)), indel_var_frac=0.10, somatic_every=20000.0, n_threads=threads)
    t0 = time.time()
    N = Synth(**dict(genome, coverage=25.0, read_seed=seed * 10 + 1, tumor_purity=0.0))
    T = Synth(**dict(genome, coverage=50.0, read_seed=seed * 10 + 2, tumor_purity=0.6))
    RN, RT = abi.Reads.from_synth(N), abi.Reads.from_synth(T)
    gen_s = time.time() - t0
    P = abi.default_params(phase_indel=1)
    V0 = abi.Variants(N.var_pos, N.var_ref, N.var_alt)
    res = dict(contig_mb=mb, tumor_alignments=int(RT.n_reads), normal_alignments=int(RN.n_reads), germline_rows=int(V0.n), somatic_rows=int(N.n_somatic), generation_s=round(gen_s, 1))
    with hip.Context(dev, P) as ctx:
        ph = ctx.phase(V0, N.ref, RN)
        keep = np.nonzero(ph.phase_set != 0)[0]
        # the merged table as the command line builds it (cli/longphase_amd.cpp, MultiGenomeVar map of HaplotagType.h:146-162): the normal sample's phased
        # het rows (role 0, no tumor row there) + the tumor VCF's rows, which hold the somatic sites only - role 2 with their tumor kind
        # for the two extraction passes; for the tagging pass the rows the caller flagged (here: all of them) become role 1 with the haplotype they derive from
        rows = [(int(N.var_pos[i]), N.var_ref[i], N.var_alt[i], int(ph.gt[i]), int(ph.phase_set[i]), 0, 0, 0) for i in keep]
        kind_of = lambda r, a: 1 if len(r) == 1 and len(a) == 1 else (2 if len(r) == 1 else 3)  # noqa: E731
        rows += [(int(p), bytes([r]), bytes([a]), 0, 0, 2, int(h) + 1, kind_of(bytes([r]), bytes([a]))) for p, r, a, h in zip(N.som_pos, N.som_ref, N.som_alt, N.som_hap)]
        rows.sort()
        rows = [r for k, r in enumerate(rows) if k == 0 or r[0] != rows[k - 1][0]]
        cols = lambda tag: dict(hp1_is_alt=[r[3] for r in rows], phase_set=[r[4] for r in rows], somatic_role=[(1 if tag and r[5] == 2 else r[5]) for r in rows],  # noqa: E731
                                derive_hp=[(r[6] if tag and r[5] == 2 else 0) for r in rows], tumor_kind=[r[7] for r in rows])
        V = abi.Variants([r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows], **cols(False))
        VT = abi.Variants([r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows], **cols(True))
        res["merged_table_rows"] = int(V.n)
        Lh = ctx.L
        def timed(fn, k):
            fn()                                                         # warm-up: buffers at their working size
            t0 = time.perf_counter()
            for _ in range(k):
                fn()
            return (time.perf_counter() - t0) / k * 1e3
        # ---- pass 1: normal BAM
        pinned_arrays = []
        def pin_all(o, names):                                           # results land in page-locked host memory (hipHostRegister), as a caller that cares would hold them
            ok = True
            for k in names:
                a = getattr(o, k)
                if pin(a):
                    pinned_arrays.append(a)
                else:
                    ok = False
            return ok
        ctx.load_chromosome(V, N.ref, RN)
        o1 = abi.SiteCountersOut(V.n, RN.n_reads)
        pinned = pin_all(o1, ("counters", "read_hp"))
        def p1():
            ctx._check(Lh.lps_somatic_extract_normal(ctx.h, C.byref(o1.c)), "lps_somatic_extract_normal")
        ms1 = timed(p1, steps); k1 = ctx.timings()["stages"]["extract"]
        # ---- pass 2: tumor BAM
        ctx.load_chromosome(V, T.ref, RT)
        pair_cap, win_cap = 64 * RT.n_reads + 1024, 256 * RT.n_reads + 1024
        o2 = abi.TumorExtractOut(V.n, RT.n_reads, pair_cap, win_cap)
        pinned = pin_all(o2, ("site", "hp1", "hp2", "hp3", "ps_min", "end_pos", "read_len", "status", "hp", "n_ps", "has_site", "pair_site", "pair_read", "pair_base_hp",
                              "win_site", "win_allele", "win_offset", "win_base")) and pinned
        def p2():
            ctx._check(Lh.lps_somatic_extract_tumor(ctx.h, C.byref(o2.c)), "lps_somatic_extract_tumor")
        ms2 = timed(p2, steps); k2 = ctx.timings()["stages"]["extract"]
        # ---- pass 3: tagging, tumor reads still resident; the table with the flagged rows
        ctx.set_table(VT, T.ref)
        o3 = abi.SomaticTagOut(RT.n_reads)
        pinned = pin_all(o3, abi.SomaticTagOut.I32 + abi.SomaticTagOut.U8) and pinned
        def p3():
            ctx._check(Lh.lps_somatic_tag_chromosome(ctx.h, C.byref(o3.c)), "lps_somatic_tag_chromosome")
        ms3 = timed(p3, steps); k3 = ctx.timings()["stages"]["extract"]
        for a in pinned_arrays:
            unpin(a)
    tot = ms1 + ms2 + ms3
    n_cig_t = int(RT.cigar.size); n_cig_n = int(RN.cigar.size)
    alg = {"normal_extract": 36 * RN.n_reads + 4 * n_cig_n, "tumor_extract": 36 * RT.n_reads + 4 * n_cig_t + 8 * int(o2.c.n_pairs) + 8 * int(o2.c.n_windows), "tag": 36 * RT.n_reads + 4 * n_cig_t + 16 * RT.n_reads}
    kms = {"normal_extract": k1, "tumor_extract": k2, "tag": k3}
    dom = max(kms, key=lambda k: kms[k])
    res.update(metric="tumor reads through the three somatic_haplotag passes / s", value=RT.n_reads / (tot * 1e-3), unit="reads/s", steps=steps,
               pass_ms=dict(normal_extract=round(ms1, 3), tumor_extract=round(ms2, 3), tag=round(ms3, 3)), kernel_ms={k: round(v, 4) for k, v in kms.items()},
               pairs=int(o2.c.n_pairs), windows=int(o2.c.n_windows), results_in_pinned_host_memory=bool(pinned),
               roofline=dict(bound="hbm", kernel=dom, achieved=alg[dom] / (kms[dom] * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=alg[dom] / (kms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             algorithmic_bytes=int(alg[dom]), note="SURVEY.md 8(d) closed form of the per-read passes: 36 B + 4 B x CIGAR words per alignment + what the pass writes per hit; the sites' bases are gathered in place"),
               clock="one-pass: every call recomputes from the decoded alignments as the push left them; call = launch to results in host memory")
    if check:
        t0 = time.time()
        ok = {}
        w1 = lps_oracle.somatic_extract_normal(P, V, N.ref, RN)
        ok["normal_extract"] = bool(np.array_equal(o1.read_hp, w1.read_hp) and np.array_equal(o1.counters, w1.counters))
        w2 = lps_oracle.somatic_extract_tumor(P, V, T.ref, RT)
        same = all(np.array_equal(getattr(o2, k), getattr(w2, k)) for k in ("status", "hp1", "hp2", "hp3", "hp", "ps_min", "end_pos", "read_len", "has_site"))
        same = same and np.array_equal(o2.site, w2.site) and o2.c.n_pairs == w2.c.n_pairs and o2.c.n_windows == w2.c.n_windows
        same = same and all(np.array_equal(a, b) for a, b in zip(o2.pairs(), w2.pairs())) and all(np.array_equal(a, b) for a, b in zip(o2.windows(), w2.windows()))
        ok["tumor_extract"] = bool(same)
        w3 = lps_oracle.somatic_tag(P, VT, RT)
        ok["tag"] = bool(all(np.array_equal(getattr(o3, k), getattr(w3, k)) for k in ("status", "hp1", "hp2", "hp3", "derive_h1", "derive_h2", "ps_min", "hp", "pq", "ps")))
        res.update(parity=ok, parity_checked=all(ok.values()), oracle_s=round(time.time() - t0, 1), tagged_somatic_reads=int((o3.hp >= 5).sum()), tagged_germline_reads=int(((o3.hp == 1) | (o3.hp == 2)).sum()))
    return res


def gpu_nodes():
    """GPUs of this machine from the KFD topology in sysfs - no HIP call (the parent of spawn_ranks must not touch the GPU).  None = unknown."""
    import glob
    n = 0; seen = False
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(ln.split()[:2] for ln in open(f) if len(ln.split()) >= 2)
        except OSError:
            continue
        seen = True
        n += int(props.get("simd_count", "0")) > 0
    return n if seen else None


def spawn_ranks(a):
    """`--gpus N` launched plainly (no torchrun, WORLD_SIZE unset): start the N ranks as CHILD processes - this process has made no GPU call and
    makes none; a process that has touched the GPU must never exec another program - hand them the rendezvous through the environment torchrun
    would set, relay rank 0's JSON line and exit with the worst of their codes."""
    import socket
    n_dev = gpu_nodes()
    if n_dev and (a.gpus + n_dev - 1) // n_dev > 6:
        raise SystemExit(f"--gpus {a.gpus} on {n_dev} GPU(s): more than 6 ranks would share a GPU")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's line is read on a helper thread while the children are SUPERVISED: when one of them dies (a port taken between the probe above and
    # the rendezvous, a rank that fails before init_process_group) the others are stopped instead of waiting for it in the rendezvous
    out0 = []
    rd = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True); rd.start()
    bad = 0
    while any(p.poll() is None for p in procs):
        rcs = [p.poll() for p in procs]
        failed = [rc for rc in rcs if rc not in (None, 0)]
        if failed:
            bad = failed[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    rd.join(5)
    sys.stdout.write(b"".join(out0).decode(errors="replace")); sys.stdout.flush()
    bad = bad or next((rc for rc in rcs if rc != 0), 0)
    raise SystemExit(bad)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="wgs_50x", choices=["wgs_50x", "chr1_50x", "chr20_30x", "chr20_30x_pileups", "5mb_10x", "mini_wgs", "somatic_tn"])
    ap.add_argument("--somatic-mb", type=float, default=0.0, help="contig length (Mb) of the tumor / normal leg: 160 for --workload somatic_tn, 16 for the sample inside the default run; 0 = those defaults")
    ap.add_argument("--no-somatic", action="store_true", help="skip the tumor / normal sample of the default run (about half a minute)")
    ap.add_argument("--seed", type=int, default=201)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity", default="all", help="all | none | comma-separated contig names whose step-0 output is compared with the oracle")
    ap.add_argument("--cpu-contig", default="", help="contig the reference binary is timed on (default: the smallest of the workload)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-whole-node", action="store_true", help="skip the 16-contig whole-node CPU baseline / P clock (about 1.5 min)")
    ap.add_argument("--ctx-per-gpu", type=int, default=4, help="contigs phased concurrently on one GPU, one context (stream, host thread) each")
    a = ap.parse_args()

    if a.workload == "somatic_tn":          # BASELINE.json configs[4] on one GPU: its own line (a step = the three passes over the pair)
        sys.path.insert(0, os.path.join(ROOT, "longphase-s_amd"))
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
        r = somatic_leg(int(os.environ.get("LOCAL_RANK", "0")), a.somatic_mb or 160.0, a.steps, min(16, ncpu), check=a.parity != "none")
        line = {"metric": r.pop("metric"), "value": r.pop("value"), "unit": r.pop("unit"), "n_gpus": 1, "steps": a.steps, "warmup": 1, "ms_per_step": sum(r["pass_ms"].values()),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i32", "data": "synthetic", "parity_checked": r.get("parity_checked"),
                "config": {"workload": f"somatic_haplotag passes, tumor 50x / normal 25x at 60 % purity, one contig of {r['contig_mb']:.0f} Mb, SNP + indel VCFs, decoded alignments resident in HBM"},
                "roofline": r.pop("roofline"), "somatic_tn": r}
        print(json.dumps(line), flush=True)
        if line["parity_checked"] is False:
            sys.exit(4)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)                       # never returns: `python3 bench.py --gpus N` without a launcher starts its N ranks itself
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = max(1, int(os.environ.get("WORLD_SIZE", "1")))
    if world != a.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} != --gpus {a.gpus}; using WORLD_SIZE")

    # product library first (binds the ROCm runtime it was built against); torch only as the control plane of a multi-rank run
    import numpy as np
    from lps import abi, hip
    from lps.synth_gpu import SynthGpu
    L = hip.load()
    dist = None
    if world > 1:
        _uid = (C.c_uint8 * 128)()
        L.lps_comm_unique_id(_uid)      # opens the system librccl now, before torch brings its own copy of that soname into the process
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))

    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    cpu_share = max(2, min(16, ncpu // world))
    n_dev = max(1, int(L.lps_device_count()))
    dev = local_rank % n_dev            # one rank per GPU on the driver's node; a rehearsal with fewer GPUs than ranks shares them
    P = abi.default_params()
    contigs = workload_contigs(a.workload, a.seed)
    mine = [contigs[i] for i in lpt(contigs, world)[rank]]
    parity_set = set(c["name"] for c in contigs) if a.parity == "all" else (set() if a.parity == "none" else set(a.parity.split(",")))
    pool = ParityPool(workers=max(2, cpu_share // 2), max_bytes=(24 << 30) + (72 << 30) // world) if parity_set else None

    # ---- SNP table: with several ranks, rank 0's packed table reaches the other GPUs by one RCCL broadcast (north_star)
    bcast = None; rccl_info = None; comm = None; invalid = None
    if world > 1:
        try:
            bcast, rccl_info, comm = snp_table_broadcast(L, dist, dev, rank, world, contigs)
        except Invalid as e:
            # a collective hung on some rank (all ranks know: the verdict was all-reduced): no timings are taken beside a thread stuck in RCCL
            invalid = str(e)
            if rank == 0:
                print(json.dumps({"metric": "het SNPs phased/sec", "value": None, "unit": "SNPs/s", "n_gpus": world, "valid": False,
                                  "rccl": {"collective": "ncclBroadcast", "ranks": world, "hung": invalid}}), flush=True)
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(3)                       # the stuck helper thread would keep a normal exit waiting

    # ---- C contexts on this rank's GPU, each with one contig resident, phase their contigs CONCURRENTLY (one host thread per context; the library
    #      call releases the GIL).  A single stream of ~70 dependent launches per call leaves much of the chip idle - launch gaps, tails of small
    #      kernels, the latency-bound scan - and a second contig fills it, as a second chromosome fills a second CPU core in the reference's OpenMP loop
    #      (PhasingProcess.cpp:113); the CLI's --gpus N does the same when N exceeds the devices.  Inputs of all C contigs are resident when the group's
    #      timed region starts; no stage events are recorded inside it.
    C_CTX = max(1, a.ctx_per_gpu)
    ctxs = [hip.Context(dev, P) for _ in range(C_CTX)]
    per_contig = []
    elapsed = 0.0; hap_elapsed = 0.0; total_phased = 0; total_reads = 0; total_tagged = 0; total_bases = 0
    largest = None; cpu_pick = None; p_clock = None; port = None
    cpu_name = a.cpu_contig or min(contigs, key=lambda c: c["contig_len"])["name"]
    gen_s = 0.0; push_s = 0.0; d2h_s = 0.0; first_call_s = 0.0; first_alloc_ms = 0.0; somatic_bad = False

    def barrier():
        if dist is not None:
            dist.barrier()

    import contextlib
    quiet = (lambda: pool.heavy) if pool is not None else contextlib.nullcontext   # `with quiet():` = no checker thread frees memory meanwhile

    def concurrent(fns):
        """start the callables together, one thread each; -> wall time until the last one is done"""
        if len(fns) == 1:
            t0 = time.perf_counter(); fns[0](); return time.perf_counter() - t0
        bar = threading.Barrier(len(fns) + 1); ends = [0.0] * len(fns); errs = []

        def work(i):
            bar.wait()
            try:
                fns[i]()
            except Exception as e:  # noqa: BLE001
                errs.append(e)
            ends[i] = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(fns))]
        for t in th:
            t.start()
        bar.wait(); t0 = time.perf_counter()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        return max(ends) - t0

    barrier()
    order = sorted(mine, key=lambda c: -c["contig_len"])                 # neighbours in size share a group: both are busy for the whole timed region
    for g0 in range(0, len(order), C_CTX):
        group = order[g0:g0 + C_CTX]
        slots = []
        for ctx, spec in zip(ctxs, group):
            kw = {k: v for k, v in spec.items() if k != "name"}
            t0 = time.time()
            g = SynthGpu(dev, **kw)
            gen_s += time.time() - t0
            V = g.variants()                   # host copy: what the oracle is given (and the haplotag table is built from)
            tdev = None
            if bcast is not None:              # the rows THIS context phases come from the broadcast buffer on the GPU, not from V
                tdev = bcast[spec["name"]]
                assert tdev[3] == V.n, "broadcast SNP table: row count differs from the contig's own"
            ref = g.host("ref")
            t0 = time.time()
            ctx.load_chromosome_device(V, ref, g.device_batch(), g.n_reads, table_dev=tdev[:3] if tdev else None)
            push_s += time.time() - t0
            host = None
            if spec["name"] in parity_set or (rank == 0 and spec["name"] == cpu_name):
                t0 = time.time(); host = g.to_host(); d2h_s += time.time() - t0
            if rank == 0 and world == 1 and spec["name"] == cpu_name and not a.no_cpu_baseline:      # (the CPU baseline is timed at N = 1 only)
                cpu_pick = [g, spec, None]      # keeps its device arrays for the SAM writer; everything else is released below
            else:
                g.release_reads()
            slots.append(dict(ctx=ctx, spec=spec, g=g, V=V, ref=ref, host=host, out=abi.PhaseOut(V.n)))
        # the FIRST call on the freshly pushed contigs (un-warmed: device buffers may still grow), timed on its own; then the remaining warm-up calls
        if a.warmup >= 1:
            first_call_s += concurrent([(lambda s=s: s["ctx"].run_phase(s["out"])) for s in slots])
            first_alloc_ms += max((float(s["ctx"].L.lps_alloc_ms(s["ctx"].h)) if hasattr(s["ctx"].L, "lps_alloc_ms") else 0.0) for s in slots)     # (the contexts run side by side: the longest of the group)
        for _ in range(max(0, a.warmup - 1)):
            concurrent([(lambda s=s: s["ctx"].run_phase(s["out"])) for s in slots])
        # ---- timed region of the group: K steps per contig, the contexts running side by side
        for s in slots:
            s["ctx"].set_stage_timing(0)

        def k_steps(s):
            s["step_ms"] = []
            # the K calls behind one entry of the library (lps_phase_chromosome_steps): each is synchronous and returns with its results in host
            # memory; between them the thread does not come back to the interpreter, where it would wait for the other contexts' threads' lock
            s["step_ms"] = [round(x, 2) for x in s["ctx"].run_phase_steps(s["out"], a.steps)]
        with quiet():
            dt = concurrent([(lambda s=s: k_steps(s)) for s in slots])
        elapsed += dt
        for s in slots:
            spec, g, V, out, ctx = s["spec"], s["g"], s["V"], s["out"], s["ctx"]
            tm = ctx.timings()
            n_ph = int((out.phase_set != 0).sum())
            if cpu_pick is not None and cpu_pick[0] is g:
                cpu_pick[2] = (V.pos.copy(), out.phase_set.copy(), out.gt.copy())
            total_phased += n_ph; total_reads += g.n_reads; total_bases += g.n_bases
            s["n_ph"] = n_ph
            s["rec"] = dict(contig=spec["name"], group=[x["spec"]["name"] for x in slots], alignments=g.n_reads, snps=V.n, phased=n_ph, gbases=round(g.n_bases / 1e9, 2),
                            group_ms_per_step=dt / a.steps * 1e3, obs=tm["n_obs"], pairs=tm["n_pairs"], nodes=tm["n_nodes"], alg=tm["algorithmic_bytes"],
                            scan_segments=tm["n_scan_segments"], scan_replayed=tm["n_scan_replayed"], gen_ms=g.gen_ms)
        s = slots[0]
        if largest is None or s["spec"]["contig_len"] > largest[0]["contig_len"]:
            # the largest contig ALONE on the GPU: call time, the dominant kernel's duration (hipEvents on the library's stream around it, live) and the
            # per-stage table (every stage event recorded, a few microseconds each)
            ctx, out = s["ctx"], s["out"]
            n_prof = max(1, min(5, a.steps))
            ctx.set_stage_timing(1); ex = 0.0
            t0 = time.perf_counter()
            for _ in range(n_prof):
                ctx.run_phase(out); ex += ctx.timings()["stages"]["extract"]
            solo = (time.perf_counter() - t0) / n_prof * 1e3
            ctx.set_stage_timing(2); st = {}
            for _ in range(n_prof):
                ctx.run_phase(out)
                for k, v in ctx.timings()["stages"].items():
                    st[k] = st.get(k, 0.0) + v / n_prof
            s["rec"].update(solo_ms_per_step=solo, extract_ms=ex / n_prof)
            largest = (s["spec"], s["rec"], st)
        # ---- secondary metric: reads haplotagged / s on the same resident alignments, table = each contig's phased SNPs
        for s in slots:
            V, out = s["V"], s["out"]
            idx = np.nonzero(out.phase_set != 0)[0]
            VT = abi.Variants.from_snps(V.pos[idx], V.ref0[idx], V.alt0[idx], hp1_is_alt=out.gt[idx], phase_set=out.phase_set[idx])
            s["ctx"].set_table(VT, s["ref"]); s["VT"] = VT
            s["hout"] = abi.HaplotagOut(s["g"].n_reads)
        concurrent([(lambda s=s: s["ctx"].run_haplotag(s["hout"])) for s in slots])

        def k_tags(s):
            s["ctx"].run_haplotag_steps(s["hout"], a.steps)
        with quiet():
            hdt = concurrent([(lambda s=s: k_tags(s)) for s in slots])
        hap_elapsed += hdt
        for s in slots:
            s["rec"].update(group_haplotag_ms_per_step=hdt / a.steps * 1e3, haplotag_kernel_ms=s["ctx"].timings()["stages"]["extract"], tagged=int((s["hout"].hp != 0).sum()))
            total_tagged += s["rec"]["tagged"]
            if s["host"] is not None and s["spec"]["name"] in parity_set:      # oracle on host threads, outside every timed region: phase result + tags
                ho = s["hout"]
                pool.submit(s["spec"]["name"], P, s["V"], s["host"], s["out"].phase_set.copy(), s["out"].gt.copy(), hap=(s["VT"], ho.hp.copy(), ho.pq.copy(), ho.ps.copy()))
                if not (rank == 0 and s["spec"]["name"] == cpu_name):
                    s["host"] = None                                       # (the P clock below still needs the copy of the contig it runs on)
        # ---- P clock: decoded batch in pinned host memory -> results in host memory, H2D included (never `value`); alone on the GPU
        for s in slots:
            if rank == 0 and s["host"] is not None and s["spec"]["name"] == cpu_name:
                R = abi.Reads.from_synth(s["host"])
                pinned = all(pin(x) for x in (s["host"].qual, s["host"].seq, s["host"].cigar))
                t0 = time.perf_counter(); s["ctx"].load_chromosome(s["V"], s["ref"], R); h2d = time.perf_counter() - t0
                t0 = time.perf_counter(); s["ctx"].run_phase(s["out"]); one = time.perf_counter() - t0
                p_clock = dict(contig=cpu_name, h2d_s=round(h2d, 3), step_s=round(one, 4), value=s["n_ph"] / (h2d + one), unit="SNPs/s", pinned_host_memory=bool(pinned),
                               note="lps_set_variants + lps_set_reference + lps_push_reads (H2D of the decoded batch) + one lps_phase_chromosome; PCIe-inclusive, never `value`")
                for x in (s["host"].qual, s["host"].seq, s["host"].cigar):
                    unpin(x)
        for s in slots:
            per_contig.append(s["rec"])
            if cpu_pick is None or cpu_pick[0] is not s["g"]:
                s["g"].close()
        log(f"[rank {rank}] " + " + ".join(f"{s['spec']['name']} ({s['g'].n_reads} alignments, {s['V'].n} SNPs)" for s in slots) +
            f": phase {dt / a.steps * 1e3:.3f} ms/step | haplotag {hdt / a.steps * 1e3:.3f} ms/step | phased {sum(s['n_ph'] for s in slots)}"
            f" | calls (ms) {[s['step_ms'] for s in slots]}")
    barrier()
    ctx = ctxs[0]

    my_elapsed = elapsed
    if dist is not None:
        import torch
        t = torch.tensor([elapsed, hap_elapsed], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([total_phased, total_reads, total_tagged, total_bases], dtype=torch.float64); dist.all_reduce(c, op=dist.ReduceOp.SUM)
        loads = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(loads, torch.tensor([my_elapsed, float(sum(s["contig_len"] for s in mine))], dtype=torch.float64))
        elapsed, hap_elapsed = float(t[0]), float(t[1])
        total_phased, total_reads, total_tagged, total_bases = (float(x) for x in c)
        rank_loads = [dict(rank=i, timed_s=round(float(x[0]), 4), bases_of_contigs=int(x[1])) for i, x in enumerate(loads)]
    else:
        rank_loads = [dict(rank=0, timed_s=round(my_elapsed, 4), bases_of_contigs=int(sum(s["contig_len"] for s in mine)))]

    parity = None
    if pool is not None:
        t0 = time.time(); res = pool.results()
        log(f"[rank {rank}] oracle comparison of {len(res)} contigs done ({time.time()-t0:.1f}s after the GPU work)")
        ok = all(r["identical"] and r.get("haplotag_identical", True) for r in res) and len(res) > 0
        if cpu_name in [r["contig"] for r in res]:
            r = [r for r in res if r["contig"] == cpu_name][0]
            port = dict(value=r["n_phased"] / r["oracle_s"], unit="SNPs/s")
        if dist is not None:
            import torch
            f = torch.tensor([1.0 if ok else 0.0, float(len(res))], dtype=torch.float64)
            allf = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]; dist.all_gather(allf, f)
            ok = all(float(x[0]) == 1.0 for x in allf if float(x[1]) > 0); n_checked = int(sum(float(x[1]) for x in allf))
        else:
            n_checked = len(res)
        parity = dict(checked=ok, contigs_checked=n_checked, contigs_total=len(contigs), oracle_seconds_rank0=round(sum(r["oracle_s"] for r in res), 1),
                      mismatching=[r["contig"] for r in res if not r["identical"]],
                      haplotag_mismatching=[r["contig"] for r in res if not r.get("haplotag_identical", True)],
                      haplotag_oracle_seconds_rank0=round(sum(r.get("haplotag_oracle_s", 0.0) for r in res), 1),
                      what="phase: phase_set of every variant and gt of every phased variant == oracle/lps_oracle.phase on the same arrays (copied back from HBM); "
                           "haplotag (secondary metric): hp, pq, ps of every alignment == lps_oracle.haplotag on the table the benched run produced")

    if rank == 0:
        assert largest is not None, "rank 0 holds the largest contig (longest-processing-time-first deal)"
        spec, rec, stage_avg = largest
        stage_avg = dict(stage_avg); stage_avg["extract"] = rec["extract_ms"]          # the two events around the kernel only (hipEvents on the library's stream), contig alone on the GPU
        dom = max((k for k in stage_avg if k != "d2h"), key=lambda k: stage_avg[k])
        alg = rec["alg"]
        achieved = alg.get(dom, 0) / (stage_avg[dom] * 1e-3) / 1e9 if stage_avg[dom] > 0 else 0.0
        traffic = None; traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(f"{spec['name']}_{spec['coverage']:.0f}x", {}).get(dom)
                if traffic is not None:
                    traffic_source = f"profiles/traffic.json ({tj.get('_collected', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of profiles/collect.sh')}): a committed counter run of this workload, NOT measured by this invocation"
            except Exception:  # noqa: BLE001
                traffic = None
        stages = {k: dict(ms=round(v, 4), alg_bytes=int(alg.get(k, 0)), gbs=round(alg.get(k, 0) / (v * 1e-3) / 1e9, 1) if v > 0 and alg.get(k, 0) else None) for k, v in stage_avg.items()}
        n_all = sum(c["n_snp"] for c in contigs)
        res = {
            "metric": "het SNPs phased/sec", "value": total_phased * a.steps / elapsed, "unit": "SNPs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if a.workload == "wgs_50x" else "weak", "vs_baseline": None, "dtype": "i32/f32", "data": "synthetic",
            # ---- scalars first (a reader that truncates the line still sees them); the long per-contig arrays come last
            "clock": "one-pass: every timed call starts from the decoded reads as the push left them; nothing prepared or cached (since ABI 20)",
            "prepare_ms": 0.0,
            "first_call_pass_ms": round(first_call_s * 1e3, 3) if a.warmup >= 1 else None,
            "first_call_alloc_ms": round(first_alloc_ms, 3) if a.warmup >= 1 else None,     # of which: host time in hipMalloc / hipFree (the stage buffers are sized in the first call; varies 5 - 450 ms with the host)
            "parity_checked": bool(parity and parity["checked"]),
            "whole_step_gbs": round(sum(sum(r["alg"].values()) for r in per_contig) * a.steps / my_elapsed / 1e9, 1),
            "secondary_value": total_reads * a.steps / hap_elapsed, "secondary_unit": "reads haplotagged/s", "secondary_ms_per_step": hap_elapsed / a.steps * 1e3,
            "host_nproc": ncpu,
            "config": {"workload": f"germline phase, {a.workload}: {len(contigs)} contig(s), {int(total_bases)/1e9:.1f} Gbases of synthetic ONT reads, {int(total_reads)} alignments, "
                                   f"~{n_all} het SNP strata; streamed per contig, decoded reads resident in HBM during a contig's timed region; a step = one pass over all contigs; "
                                   "one-pass clock: a call recomputes everything from the pushed arrays (no layout pass outside the timed region: the library has none)",
                       "seed": a.seed, "phased_per_step": int(total_phased), "contigs": len(contigs), "parallelism": f"contigs dealt longest-first onto {world} rank(s); {C_CTX} context(s) per GPU, each with one contig resident, run side by side", "contexts_per_gpu": C_CTX,
                       "generation_s": round(gen_s, 2), "device_push_s": round(push_s, 2), "d2h_for_oracle_s": round(d2h_s, 2)},
            "roofline": {"bound": "hbm", "kernel": dom, "at": f"{spec['name']} {spec['coverage']:.0f}x ({rec['alignments']} alignments, {rec['snps']} SNPs, {rec['obs']} observations)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes": int(alg.get(dom, 0)), "kernel_ms": stage_avg[dom], "solo_call_ms": rec.get("solo_ms_per_step"),
                         "note": "dominant stage by time at the largest contig, measured with that contig ALONE on the GPU: durations from hipEvents on the library's stream (extract: the two events around the kernel, live; the others: pass with every stage event recorded)"},
            "secondary": {"metric": "reads haplotagged/sec", "value": total_reads * a.steps / hap_elapsed, "unit": "reads/s", "ms_per_step": hap_elapsed / a.steps * 1e3,
                          "reads_tagged_per_step": int(total_tagged), "config": "germline haplotag, same resident 50x alignments (BASELINE.json configs[2]), table = this run's phased SNPs"},
            "p_clock": p_clock,
            "parity": parity,
            "stages_at_largest_contig": stages,
            "rank_loads": rank_loads,
        }
        if rccl_info is not None:
            res["rccl"] = rccl_info
        if a.workload == "wgs_50x" and not a.no_somatic and world == 1:
            # config 5 inside the default line: a bounded sample of the tumor / normal leg (`--workload somatic_tn` runs it at 160 Mb)
            try:
                t0 = time.time()
                sm = somatic_leg(dev, a.somatic_mb or 16.0, max(2, min(a.steps, 5)), min(16, ncpu), check=a.parity != "none")
                res["somatic_tn_value"] = sm["value"]; res["somatic_tn_unit"] = sm["unit"]; res["somatic_tn_parity_checked"] = sm.get("parity_checked")
                res["somatic_tn"] = sm
                if sm.get("parity_checked") is False:
                    somatic_bad = True
                log(f"somatic tumor / normal sample took {time.time()-t0:.1f}s")
            except Exception as e:  # noqa: BLE001
                log("somatic sample failed:", repr(e)[:300]); res["somatic_tn"] = dict(failed=repr(e)[:200])
        cpu_threads = a.cpu_threads or min(24, ncpu)          # the reference's compute parallelism is its chromosome loop: at most 24 threads compute on a genome (PhasingProcess.cpp:106,113)
        if not a.no_cpu_baseline and cpu_pick is not None:
            t0 = time.time()
            one = cpu_baseline(cpu_pick[0], cpu_pick[1], cpu_threads, port, cpu_pick[2])
            one["nproc"] = ncpu
            cpu_pick[0].close()
            log(f"cpu baseline (one contig) took {time.time()-t0:.1f}s")
            res["cpu_baseline"] = one
            if a.workload == "wgs_50x" and not a.no_whole_node:
                t0 = time.time()
                for cx in ctxs:                                            # (their buffers are not needed any more: room for the sample's two contexts)
                    cx.close()
                try:
                    wn = whole_node_baseline(dev, P, cpu_threads, a.seed, n_contigs=cpu_threads)
                except Exception as e:  # noqa: BLE001
                    log("whole-node baseline failed:", repr(e)[:300]); wn = None
                if wn is not None:
                    # the number that matters - every host thread with a contig of its own - is cpu_baseline.value; the one-contig run (one compute
                    # thread, the others feed BGZF) stays beside it
                    wn["k_clock_over_cpu"] = round(res["value"] / wn["value"], 1)
                    wn["nproc"] = ncpu
                    wn["one_contig"] = one
                    res["cpu_baseline"] = wn
                log(f"whole-node baseline took {time.time()-t0:.1f}s")
        elif port:
            res["cpu_baseline"] = dict(port, kind="port", cores=1, nproc=ncpu, sample=f"contig {cpu_name}, oracle restatement on decoded arrays, one thread")
        res["per_contig_rank0"] = [{k: v for k, v in r.items() if k != "alg"} for r in per_contig]
        print(json.dumps(res), flush=True)
    for cx in ctxs:
        cx.close()
    if comm:
        L.lps_comm_destroy(comm)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    rc = 0
    if somatic_bad:
        log(f"[rank {rank}] PARITY FAILED in the tumor / normal sample"); rc = 4
    if parity is not None and not parity["checked"]:
        log(f"[rank {rank}] PARITY FAILED: {parity}")
        rc = 4                                     # the JSON line above carries parity_checked false; the exit code says it too
    if dist is not None:
        # a multi-rank process holds two ROCm stacks (the system's, which liblps_hip.so and librccl were built against, and the one torch ships for
        # its gloo control plane): everything of ours is closed and flushed above, the interpreter's teardown of both at once is skipped
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(rc)
    if rc:
        sys.exit(rc)


class Invalid(Exception):
    """the run cannot publish a measurement (a collective hung on some rank): every rank prints what it knows and exits non-zero"""


def snp_table_broadcast(L, dist, dev, rank, world, contigs):
    """Rank 0 packs the SNP table of the whole genome (pos i32 | ref0 u8 | alt0 u8 per row, contig after contig); ONE ncclBroadcast (RCCL over xGMI,
    behind lps_comm_bcast_to_device) puts it into a device buffer on every rank's GPU, where it STAYS: each contig's rows are handed to the library
    as device pointers (lps_set_variants_device) - no copy back to the host.  Returns ({contig: (dev_pos, dev_ref0, dev_alt0)} or None, info, comm).

    Every decision that only some ranks could take alone - a device without a bus id, a failed ncclGetUniqueId, a communicator that does not come up -
    is taken TOGETHER: the ranks all-reduce an ok flag over the control plane (gloo) before the next step, so nobody waits in a collective its peers
    skipped.  A step that HANGS (watchdog, 120 s) leaves a thread stuck inside RCCL on this GPU: nothing measured beside it is published - Invalid."""
    import numpy as np
    import torch
    from lps.synth_gpu import SynthGpu
    info = dict(collective="ncclBroadcast", ranks=world)

    def agree(ok):                          # 1 fine, 0 failed here, -1 hung here -> the worst over all ranks
        t = torch.tensor([int(ok)], dtype=torch.int64); dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t[0])

    def guarded(what, fn):                  # run fn under the watchdog; -> (state, value)
        box = {}
        def run():
            try:
                box["v"] = fn()
            except Exception as e:  # noqa: BLE001
                box["e"] = e
        th = threading.Thread(target=run, daemon=True); th.start(); th.join(120.0)
        if th.is_alive():
            log(f"[rank {rank}] {what} not done after 120 s")
            return -1, None
        if "e" in box:
            log(f"[rank {rank}] {what} failed: {box['e']!r}")
            return 0, None
        return 1, box["v"]

    def settle(step, state):
        worst = agree(state)
        if worst < 0:
            info.update(hung=step); raise Invalid(step)
        if worst == 0:
            info.update(skipped=f"{step} failed on some rank: every rank derives its own contigs' table")
        return worst == 1

    # RCCL refuses two ranks on one GPU - and a refused ncclCommInitRank leaves the process without a usable HIP runtime - so the ranks first compare
    # the PCI bus ids of their devices; a rehearsal with fewer GPUs than ranks has no communicator
    bus = C.create_string_buffer(64)
    my_id = bus.value.decode() if L.lps_device_bus_id(dev, bus, 64) == 0 else None
    ids = [None] * world
    dist.all_gather_object(ids, my_id)
    if any(i is None for i in ids):
        info.update(skipped="lps_device_bus_id failed on some rank"); return None, info, None
    if len(set(ids)) != world:
        log(f"[rank {rank}] ranks share a GPU ({ids}): no RCCL communicator; every rank derives its own contigs' table")
        info.update(skipped=f"{world} ranks on {len(set(ids))} GPU(s): RCCL refuses ranks that share a device"); return None, info, None
    uid = (C.c_uint8 * 128)()
    ok = L.lps_comm_unique_id(uid) == 0 if rank == 0 else True
    if not settle("ncclGetUniqueId", 1 if ok else 0):
        return None, info, None
    t = torch.tensor(list(uid), dtype=torch.uint8); dist.broadcast(t, src=0)
    uid = (C.c_uint8 * 128)(*t.tolist())
    state, comm = guarded("lps_comm_create", lambda: L.lps_comm_create(dev, world, rank, uid))
    if state == 1 and not comm:
        log(f"[rank {rank}] lps_comm_create: {L.lps_comm_last_error().decode()}"); state = 0
    if not settle("ncclCommInitRank", state):
        if comm:
            L.lps_comm_destroy(comm)
        return None, info, None
    counts = torch.zeros(len(contigs), dtype=torch.int64)
    buf = None
    if rank == 0:
        parts = []
        for i, spec in enumerate(contigs):
            kw = {k: v for k, v in spec.items() if k != "name"}; kw["coverage"] = 0.0          # reference + variants only
            g = SynthGpu(dev, **kw)
            parts.append((g.host("var_pos"), g.host("var_ref0"), g.host("var_alt0"))); counts[i] = g.n_variants
            g.close()
    dist.broadcast(counts, src=0)
    n = int(counts.sum())
    if rank == 0:
        buf = np.zeros(n * 6, dtype=np.uint8)
        buf[:4 * n] = np.concatenate([p[0] for p in parts]).view(np.uint8)
        buf[4 * n:5 * n] = np.concatenate([p[1] for p in parts]); buf[5 * n:] = np.concatenate([p[2] for p in parts])
    ms = C.c_double(0); dptr = C.c_void_p(0)
    t0 = time.perf_counter()
    state, rc = guarded("lps_comm_bcast_to_device", lambda: L.lps_comm_bcast_to_device(comm, buf.ctypes.data if rank == 0 else None, 6 * n, 0, C.byref(dptr), C.byref(ms)))
    if state == 1 and rc != 0:
        log(f"[rank {rank}] lps_comm_bcast_to_device rc={rc}: {L.lps_comm_last_error().decode()}"); state = 0
    if not settle("ncclBroadcast", state):
        return None, info, comm
    info.update(bytes=6 * n, wall_ms=round((time.perf_counter() - t0) * 1e3, 2), device_ms=round(ms.value, 3), n_ranks_in_communicator=int(L.lps_comm_size(comm)),
                consumed="every rank loads its contigs' rows from the broadcast buffer on its GPU (lps_set_variants_device)")
    out = {}; o = 0; base = int(dptr.value)
    for i, spec in enumerate(contigs):
        k = int(counts[i]); out[spec["name"]] = (base + 4 * o, base + 4 * n + o, base + 5 * n + o, k); o += k
    return out, info, comm


if __name__ == "__main__":
    main()
