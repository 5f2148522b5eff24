#!/usr/bin/env python3
"""profiles/e2e_cli.py — clock E of SURVEY.md §8d: the drop-in CLI (longphase-s_amd/cli/longphase_amd) against the reference binary
(oracle/_ref/longphase-s-ref) on the same BAM/VCF/FASTA files, process start to exit, for `phase`, `haplotag`, `somatic_haplotag`, plus the GPU
BGZF inflate against zlib.  Moved out of bench.py (round 2): the driver's bench budget belongs to the headline workload.

  python profiles/e2e_cli.py [--contig-mb 64] [--threads 16] > profiles/rNN_e2e_cli.json      (on the GPU box)
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "longphase-s_amd"))


def log(*a):
    print(*a, file=sys.stderr, flush=True)


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
    ap.add_argument("--contig-mb", type=int, default=64, help="contig length (Mb) of the 30x sample (64 = chr20-sized)")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--seed", type=int, default=101)
    a = ap.parse_args()
    from lps.synth import Synth
    frac = a.contig_mb * 1e6 / 64_444_167
    s = Synth(seed=a.seed + 7000, contig_len=int(a.contig_mb * 1e6), n_snp=int(60_000 * frac), coverage=30.0, n_threads=a.threads)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "longphase-s-ref"); tv = os.path.join(ROOT, "oracle", "_ref", "test_view")
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        s.write_fasta(d + "/ref.fa"); s.write_vcf(d + "/in.vcf"); s.write_sam(d + "/reads.sam")
        subprocess.check_call([tv, "-b", "-x", "reads.bam.bai", "-p", "reads.bam", "reads.sam"], cwd=d, stdout=subprocess.DEVNULL)
        os.remove(d + "/reads.sam")
        cmd = [ref_bin, "phase", "-s", "in.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(a.threads), "-o", "out", "--ont"]
        subprocess.run(cmd, cwd=d, capture_output=True)
        ts = []
        for _ in range(3):
            t0 = time.time(); r = subprocess.run(cmd, cwd=d, capture_output=True); ts.append(time.time() - t0)
            assert r.returncode == 0, r.stderr[-500:]
        ts.sort()
        n_ph = sum(1 for ln in open(d + "/out.vcf") if not ln.startswith("#") and not ln.rstrip().endswith(":."))
        res = cli_e2e(d, a.threads, ts[1], n_ph)
    res["sample"] = f"{a.contig_mb} Mb contig at 30x ({s.n_reads} alignments, {s.n_variants} het SNPs), {a.threads} threads"
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
