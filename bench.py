#!/usr/bin/env python3
"""bench.py — throughput of the `phase` hot path on MI355X (BASELINE.json metric: het SNPs phased / s).

A step = one lps_phase_chromosome() over the resident decoded reads of one synthetic chr20-sized contig at 30x
(BASELINE.json configs[1]): everything from allele extraction to phased genotypes is recomputed from the raw
reads in HBM and the result is copied back to host memory.  Inputs are uploaded before the timed region.

  python bench.py --gpus N --steps K --warmup W
N>1 (launched by torch.distributed.run, one rank per GPU): every rank owns an independent contig shard (weak
scaling, no data-path collective - phasing never crosses contigs); torch.distributed only provides
the barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "longphase-s_amd"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec

WORKLOADS = {
    # BASELINE.json configs[1]: germline phase chr20, 30x ONT synthetic, ~60k het SNPs
    "chr20_30x": dict(contig_len=64_444_167, n_snp=60_000, coverage=30.0),
    # BASELINE.json configs[0] (plumbing size)
    "5mb_10x": dict(contig_len=5_000_000, n_snp=5_000, coverage=10.0),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(seed, threads, sample_mb=8):
    """Reference CPU path on a bounded sample of the same workload (same generator, same density/coverage).
    kind 'reference' = the real LongPhase-S binary (oracle/_ref, BAM+VCF+FASTA in, incl. BGZF/BAM decode);
    falls back to kind 'port' (the oracle restatement on decoded arrays, 1 thread) when the binary is absent."""
    from lps.synth import Synth
    from lps import abi
    kw = dict(WORKLOADS["chr20_30x"])
    frac = sample_mb * 1e6 / kw["contig_len"]
    kw.update(contig_len=int(sample_mb * 1e6), n_snp=int(kw["n_snp"] * frac), seed=seed + 7000, n_threads=threads)
    s = Synth(**kw)
    sample = f"{sample_mb} Mb contig at 30x from the same generator ({s.n_reads} alignments, {s.n_variants} het SNPs)"
    out = {}
    # port: oracle restatement on decoded SoA, single thread
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lps_oracle
    V = abi.Variants(s.var_pos, s.var_ref, s.var_alt)
    R = abi.Reads.from_synth(s)
    t0 = time.time()
    o, _ = lps_oracle.phase(abi.default_params(), V, s.ref, R)
    tp = time.time() - t0
    port = dict(value=float((o.phase_set != 0).sum() / tp), unit="SNPs/s", cores=1, kind="port", sample=sample,
                note="decoded arrays in memory, no BAM/BGZF decode")
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "longphase-s-ref")
    tv = os.path.join(ROOT, "oracle", "_ref", "test_view")
    if os.path.exists(ref_bin) and os.path.exists(tv):
        try:
            with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
                s.write_fasta(d + "/ref.fa"); s.write_vcf(d + "/in.vcf"); s.write_sam(d + "/reads.sam")
                subprocess.check_call([tv, "-b", "-x", "reads.bam.bai", "-p", "reads.bam", "reads.sam"], cwd=d, stdout=subprocess.DEVNULL)
                os.remove(d + "/reads.sam")
                cmd = [ref_bin, "phase", "-s", "in.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(threads), "-o", "out", "--ont"]
                subprocess.run(cmd, cwd=d, capture_output=True)                     # warm the page cache
                ts = []
                for _ in range(3):
                    t0 = time.time(); r = subprocess.run(cmd, cwd=d, capture_output=True); ts.append(time.time() - t0)
                    assert r.returncode == 0, r.stderr[-500:]
                ts.sort()
                n_ph = sum(1 for ln in open(d + "/out.vcf") if not ln.startswith("#") and not ln.rstrip().endswith(":."))
                e2e = cli_e2e(d, threads, ts[1], n_ph)
            out = dict(value=float(n_ph / ts[1]), unit="SNPs/s", cores=threads, kind="reference",
                       sample=sample + f"; median of 3 runs of `longphase-s phase -t {threads}` end to end (one contig => one compute thread, the rest feed BGZF)",
                       port_value=port["value"], port_note=port["note"], e2e=e2e)
        except Exception as e:  # noqa: BLE001
            log("cpu_baseline: reference run failed, using the port:", repr(e)[:300])
    if not out:
        out = port
    s.close()
    return out


def cli_e2e(d, threads, ref_wall, n_ph):
    """Clock E of SURVEY.md §8d: the drop-in CLI (longphase-s_amd/cli/longphase_amd: BGZF inflate + BAM decode + GPU path +
    VCF rewrite, process start to exit) on the very files the reference binary was just timed on, and a byte
    comparison of the two output VCFs (minus the version / command-line header lines)."""
    cli = os.path.join(ROOT, "longphase-s_amd", "cli", "longphase_amd")
    if not os.path.exists(cli):
        return None
    cmd = [cli, "phase", "-s", "in.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(threads), "-o", "gpu", "--ont"]
    ts = []
    for _ in range(3):
        t0 = time.time(); r = subprocess.run(cmd, cwd=d, capture_output=True); ts.append(time.time() - t0)
        if r.returncode != 0:
            return {"error": r.stderr.decode()[-300:]}
    ts.sort()
    body = lambda p: [l for l in open(p) if not l.startswith("##commandline=") and not l.startswith("##longphaseVersion=")]  # noqa: E731
    stages = r.stderr.decode().strip().splitlines()[-1] if r.stderr else ""
    th = []
    for _ in range(2):                                              # same CLI with zlib on the host threads instead of the GPU inflate
        t0 = time.time(); rh = subprocess.run(cmd + ["--host-inflate"], cwd=d, capture_output=True); th.append(time.time() - t0)
    host_stages = rh.stderr.decode().strip().splitlines()[-1] if rh.stderr else ""
    tag = None
    try:                                                            # same comparison for `haplotag` (reads tagged / s, end to end)
        import gzip
        import hashlib
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "longphase-s-ref")
        rcmd = [ref_bin, "haplotag", "-s", "out.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(threads), "-o", "ref_tagged"]
        ccmd = [cli, "haplotag", "-s", "out.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(threads), "-o", "gpu_tagged"]
        tr, tc = [], []
        for _ in range(2):
            t0 = time.time(); r1 = subprocess.run(rcmd, cwd=d, capture_output=True); tr.append(time.time() - t0)
            t0 = time.time(); r2 = subprocess.run(ccmd, cwd=d, capture_output=True); tc.append(time.time() - t0)
            assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-300:], r2.stderr[-300:])

        def records_digest(path):
            h = hashlib.sha256(); n = 0
            with gzip.open(path, "rb") as f:
                head = f.read(8); lt = int.from_bytes(head[4:8], "little"); f.read(lt)
                nref = int.from_bytes(f.read(4), "little")
                for _ in range(nref):
                    ln = int.from_bytes(f.read(4), "little"); f.read(ln + 4)
                while True:
                    b = f.read(1 << 24)
                    if not b:
                        break
                    h.update(b); n += len(b)
            return h.hexdigest(), n
        a, b = records_digest(d + "/ref_tagged.bam"), records_digest(d + "/gpu_tagged.bam")
        n_aln = int([l for l in r2.stderr.decode().splitlines() if l.startswith("total alignment")][0].split()[2])
        tag = {"cli_wall_s": round(min(tc), 3), "reference_wall_s": round(min(tr), 3), "speedup": round(min(tr) / min(tc), 2),
               "cli_reads_per_s": n_aln / min(tc), "reference_reads_per_s": n_aln / min(tr), "identical_record_stream": a == b, "record_bytes": b[1],
               "cli_stages": r2.stderr.decode().strip().splitlines()[-1], "output_bytes": {"cli": os.path.getsize(d + "/gpu_tagged.bam"), "reference": os.path.getsize(d + "/ref_tagged.bam")},
               "note": "best of 2; both write every record; the CLI inflates, scores, re-tags and deflates on the GPU (per-4-KiB Huffman codes, no LZ77), "
                       "the reference uses htslib/zlib level 6 on its thread pool - see output_bytes"}
    except Exception as e:  # noqa: BLE001
        tag = {"error": repr(e)[:300]}
    gz = None
    try:                                                            # GPU BGZF inflate of the same (htslib-written) BAM, checked against zlib
        import gzip
        import numpy as np
        from lps import abi, hip
        raw = np.fromfile(d + "/reads.bam", dtype=np.uint8)
        with hip.Context(int(os.environ.get("LOCAL_RANK", "0")), abi.default_params()) as c2:
            c2.bgzf_load(raw)
            t0 = time.time(); n_inf = c2.bgzf_load(raw); wall = time.time() - t0
            tm = c2.bgzf_timings()
            t0 = time.time(); want = gzip.decompress(raw.tobytes()); zt = time.time() - t0
            same = len(want) == n_inf
            for a in range(0, n_inf, 64 << 20):
                k = min(64 << 20, n_inf - a)
                same = same and c2.bgzf_read(a, k).tobytes() == want[a:a + k]
        gz = {"compressed_bytes": int(raw.size), "inflated_bytes": n_inf, "h2d_ms": round(tm["h2d_ms"], 2), "inflate_kernel_ms": round(tm["inflate_ms"], 2),
              "inflate_GBps_out": round(n_inf / tm["inflate_ms"] / 1e6, 1), "call_wall_s": round(wall, 3), "identical_to_zlib": bool(same),
              "zlib_1thread_s": round(zt, 2)}
    except Exception as e:  # noqa: BLE001
        gz = {"error": repr(e)[:300]}
    som = None
    try:                                                            # BASELINE.json configs[4] in miniature: tumor/normal pair, somatic_haplotag end to end
        import hashlib
        from lps.synth import Synth
        genome = dict(contig_len=4_000_000, n_snp=3700, n_threads=threads, somatic_every=8000.0, seed=5101)
        N = Synth(**dict(genome, coverage=25.0, read_seed=5111, tumor_purity=0.0)); T = Synth(**dict(genome, coverage=50.0, read_seed=5112, tumor_purity=0.6))
        N.write_fasta(d + "/tn_ref.fa"); N.write_vcf(d + "/tn_normal_in.vcf"); N.write_sam(d + "/tn_normal.sam"); T.write_sam(d + "/tn_tumor.sam"); T.write_vcf_tumor(d + "/tn_tumor.vcf", "chrS", with_germline=True)
        n_t = int(T.n_reads); N.close(); T.close()
        tv = os.path.join(ROOT, "oracle", "_ref", "test_view"); ref_bin = os.path.join(ROOT, "oracle", "_ref", "longphase-s-ref")
        for smp in ("tn_normal", "tn_tumor"):
            subprocess.check_call([tv, "-b", "-x", smp + ".bam.bai", "-p", smp + ".bam", smp + ".sam"], cwd=d, stdout=subprocess.DEVNULL); os.remove(d + "/" + smp + ".sam")
        r0 = subprocess.run([ref_bin, "phase", "-s", "tn_normal_in.vcf", "-b", "tn_normal.bam", "-r", "tn_ref.fa", "-t", str(threads), "-o", "tn_normal_phased", "--ont"], cwd=d, capture_output=True)
        assert r0.returncode == 0, r0.stderr[-300:]
        common = ["somatic_haplotag", "-s", "tn_normal_phased.vcf", "-b", "tn_normal.bam", "--tumor-snv-file", "tn_tumor.vcf", "--tumor-bam-file", "tn_tumor.bam", "-r", "tn_ref.fa", "-t", str(threads)]
        tr, tc = [], []
        for _ in range(2):
            t0 = time.time(); r1 = subprocess.run([ref_bin] + common + ["-o", "tn_ref_out"], cwd=d, capture_output=True); tr.append(time.time() - t0)
            t0 = time.time(); r2 = subprocess.run([cli] + common + ["-o", "tn_gpu_out"], cwd=d, capture_output=True); tc.append(time.time() - t0)
            assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-300:], r2.stderr[-300:])

        def digest(path):
            import gzip
            h = hashlib.sha256()
            with gzip.open(path, "rb") as f:
                head = f.read(8); f.read(int.from_bytes(head[4:8], "little")); nref = int.from_bytes(f.read(4), "little")
                for _ in range(nref):
                    ln = int.from_bytes(f.read(4), "little"); f.read(ln + 4)
                for b in iter(lambda: f.read(1 << 24), b""):
                    h.update(b)
            return h.hexdigest()
        som = {"cli_wall_s": round(min(tc), 3), "reference_wall_s": round(min(tr), 3), "speedup": round(min(tr) / min(tc), 2), "tumor_reads_per_s_cli": n_t / min(tc), "tumor_reads_per_s_reference": n_t / min(tr),
               "identical_record_stream": digest(d + "/tn_ref_out.bam") == digest(d + "/tn_gpu_out.bam"), "identical_purity_report": open(d + "/tn_ref_out_purity.out").read() == open(d + "/tn_gpu_out_purity.out").read(),
               "sample": "4 Mb contig, normal 25x + tumor 50x at 60 % purity, automatic purity estimation, best of 2", "cli_stages": r2.stderr.decode().strip().splitlines()[-1]}
    except Exception as e:  # noqa: BLE001
        som = {"error": repr(e)[:300]}
    return {"cli_wall_s": round(ts[1], 3), "cli_stages": stages, "cli_host_inflate_wall_s": round(min(th), 3), "cli_host_inflate_stages": host_stages,
            "haplotag": tag, "somatic_haplotag": som, "gpu_bgzf": gz, "reference_wall_s": round(ref_wall, 3), "speedup": round(ref_wall / ts[1], 2),
            "cli_snps_per_s": float(n_ph / ts[1]), "identical_vcf": body(d + "/gpu.vcf") == body(d + "/out.vcf"),
            "note": "same BAM/VCF/FASTA files, process start to exit, median of 3; CLI wall includes HIP runtime start-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="chr20_30x", choices=sorted(WORKLOADS))
    ap.add_argument("--seed", type=int, default=101)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gen-threads", type=int, default=0)
    ap.add_argument("--cpu-sample-mb", type=int, default=8, help="contig length of the bounded cpu_baseline / CLI end-to-end sample")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} != --gpus {a.gpus}; using WORLD_SIZE")
    n_gpus = max(world, 1)

    # product library first (binds the ROCm runtime it was built against); torch only for the multi-rank barrier
    from lps import abi, hip
    from lps.synth import Synth
    hip.load()
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)   # control plane only (barrier / max)

    ncpu = os.cpu_count() or 8
    threads = a.gen_threads or max(2, min(16, ncpu // max(1, min(world, 8))))
    kw = dict(WORKLOADS[a.workload]); kw.update(seed=a.seed + rank, n_threads=threads)
    t0 = time.time()
    s = Synth(**kw)
    V = abi.Variants(s.var_pos, s.var_ref, s.var_alt)
    R = abi.Reads.from_synth(s)
    if rank == 0:
        log(f"generated {a.workload}: {s.n_reads} alignments, {s.n_variants} het SNPs, {s.qual.size/1e9:.2f} Gbases in {time.time()-t0:.1f}s")
    P = abi.default_params()
    n_dev = max(1, int(hip.load().lps_device_count()))
    ctx = hip.Context(local_rank % n_dev, P)        # one rank per GPU on the node the driver uses; a rehearsal with fewer GPUs than ranks shares them
    t0 = time.time()
    ctx.load_chromosome(V, s.ref, R)
    h2d_s = time.time() - t0
    out = abi.PhaseOut(V.n)
    for _ in range(a.warmup):
        ctx.run_phase(out)

    def barrier():
        if dist is not None:
            dist.barrier()

    # Timed region: events only around the extraction kernel (the dominant one) - every event recorded on the stream idles the GPU for a few
    # microseconds, the full per-stage set costs ~4 % of a step.  The per-stage table comes from a separate, untimed pass below.
    ctx.set_stage_timing(1)
    ctx.run_phase(out)
    extract_ms = 0.0
    barrier()
    t_start = time.perf_counter()
    for _ in range(a.steps):
        ctx.run_phase(out)                      # synchronous: returns with results in host memory (stream drained)
        extract_ms += ctx.timings()["stages"]["extract"]
    elapsed = time.perf_counter() - t_start
    barrier()
    n_phased = int((out.phase_set != 0).sum())
    stage_ms = {}
    ctx.set_stage_timing(2)
    n_prof = max(1, min(10, a.steps))
    for _ in range(n_prof):
        ctx.run_phase(out)
        for k, v in ctx.timings()["stages"].items():
            stage_ms[k] = stage_ms.get(k, 0.0) + v
    tm = ctx.timings()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([n_phased], dtype=torch.float64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_phased = float(c.item())
    else:
        total_phased = float(n_phased)

    # secondary metric of BASELINE.json: reads haplotagged / s (same resident reads, table = this run's phased SNPs)
    import numpy as np
    idx = np.nonzero(out.phase_set != 0)[0]
    VT = abi.Variants(V.pos[idx], [V.ref_str[i] for i in idx], [V.alt_str[i] for i in idx], hp1_is_alt=out.gt[idx], phase_set=out.phase_set[idx])
    import ctypes as _C
    ctx._check(ctx.L.lps_set_variants(ctx.h, _C.byref(VT.c)), "lps_set_variants")
    refa = np.ascontiguousarray(s.ref, dtype=np.uint8)
    ctx._check(ctx.L.lps_set_reference(ctx.h, refa.ctypes.data, refa.size), "lps_set_reference")
    hout = abi.HaplotagOut(R.n_reads)
    ctx.run_haplotag(hout)
    t_h = time.perf_counter()
    for _ in range(a.steps):
        ctx.run_haplotag(hout)
    hap_elapsed = time.perf_counter() - t_h
    hap_tm = ctx.timings()
    n_scored = int((hout.status == 0).sum())

    if rank == 0:
        stage_avg = {k: v / n_prof for k, v in stage_ms.items()}
        dom = max((k for k in stage_avg if k != "d2h"), key=lambda k: stage_avg[k])
        if dom == "extract":
            stage_avg["extract"] = extract_ms / a.steps     # measured live in the timed region (hipEvents on the library's stream)
        alg = tm["algorithmic_bytes"]
        # roofline of the dominant kernel (by time), algorithmic bytes / measured duration (hipEvents on the lib's stream)
        dom_bytes = alg.get(dom, 0)
        achieved = dom_bytes / (stage_avg[dom] * 1e-3) / 1e9 if stage_avg[dom] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(a.workload, {}).get(dom)
            except Exception:  # noqa: BLE001
                traffic = None
        stages = {}
        for k, v in stage_avg.items():
            b = alg.get(k, 0)
            stages[k] = dict(ms=round(v, 4), alg_bytes=int(b), gbs=round(b / (v * 1e-3) / 1e9, 2) if v > 0 and b else None)
        res = {
            "metric": "het SNPs phased/sec", "value": total_phased * a.steps / elapsed, "unit": "SNPs/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i32/f32", "data": "synthetic",
            "config": {"workload": f"germline phase, {a.workload} synthetic ONT ({s.n_reads} alignments, {s.n_variants} het SNPs, "
                                   f"{s.qual.size/1e9:.2f} Gbases) per GPU; decoded reads resident in HBM", "seed": a.seed,
                       "phased_per_step_per_gpu": n_phased, "obs": tm["n_obs"], "pairs": tm["n_pairs"], "nodes": tm["n_nodes"],
                       "h2d_seconds_untimed": round(h2d_s, 2)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "dominant stage by time, its duration from hipEvents inside the timed region; per-stage algorithmic GB/s in `stages`"},
            "stages": stages,
            "stages_note": f"all stages but the dominant one: separate untimed pass of {n_prof} steps with every stage event recorded",
            "scan": {"segments": tm["n_scan_segments"], "replayed_serially": tm["n_scan_replayed"]},
            "secondary": {"metric": "reads haplotagged/sec", "value": R.n_reads * a.steps / hap_elapsed * n_gpus, "unit": "reads/s",
                          "ms_per_step": hap_elapsed / a.steps * 1e3, "kernel_ms": hap_tm["stages"]["extract"],
                          "reads_scored_per_step": n_scored, "reads_tagged_per_step": int((hout.hp != 0).sum()),
                          "note": "same resident alignments, phased table = this run's phase output; per-GPU rate x n_gpus"},
        }
        if not a.no_cpu_baseline:
            t0 = time.time()
            res["cpu_baseline"] = cpu_baseline(a.seed, min(16, ncpu), a.cpu_sample_mb)
            log(f"cpu baseline took {time.time()-t0:.1f}s")
        print(json.dumps(res), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
