#!/usr/bin/env python3
"""profiles/e2e_somatic.py — BASELINE.json configs[4] at one contig, end to end on files: a tumor / normal pair (50x / 25x ONT-like reads, SNP + indel
VCFs, 60 % purity) through `somatic_haplotag` of the reference binary and of longphase_amd, process start to exit; the tagged tumor BAM's record
stream and the purity report must be identical.

  python profiles/e2e_somatic.py [--contig-mb 24] [--threads 16] > profiles/rNN_e2e_somatic.json      (on the GPU box)
"""
import argparse
import gzip
import hashlib
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


def digest(path):
    h = hashlib.sha256()
    with gzip.open(path, "rb") as f:
        head = f.read(8); f.read(int.from_bytes(head[4:8], "little")); nref = int.from_bytes(f.read(4), "little")
        for _ in range(nref):
            ln = int.from_bytes(f.read(4), "little"); f.read(ln + 4)
        for b in iter(lambda: f.read(1 << 24), b""):
            h.update(b)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contig-mb", type=int, default=24)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    from lps.synth import Synth
    cli = os.path.join(ROOT, "longphase-s_amd", "cli", "longphase_amd")
    tv = os.path.join(ROOT, "oracle", "_ref", "test_view"); ref_bin = os.path.join(ROOT, "oracle", "_ref", "longphase-s-ref")
    L = a.contig_mb * 1_000_000
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        t0 = time.time()
        genome = dict(contig_len=L, n_snp=L // 1000, n_threads=a.threads, somatic_every=6000.0, indel_var_frac=0.15, seed=5201)
        N = Synth(**dict(genome, coverage=25.0, read_seed=5211, tumor_purity=0.0)); T = Synth(**dict(genome, coverage=50.0, read_seed=5212, tumor_purity=0.6))
        N.write_fasta(d + "/ref.fa"); N.write_vcf(d + "/normal_in.vcf"); N.write_sam(d + "/normal.sam"); T.write_sam(d + "/tumor.sam"); T.write_vcf_tumor(d + "/tumor.vcf", "chrS", with_germline=True)
        n_t, n_n, n_som = int(T.n_reads), int(N.n_reads), int(N.n_somatic); N.close(); T.close()
        for smp in ("normal", "tumor"):
            subprocess.check_call([tv, "-@", str(a.threads), "-b", "-x", smp + ".bam.bai", "-p", smp + ".bam", smp + ".sam"], cwd=d, stdout=subprocess.DEVNULL); os.remove(d + "/" + smp + ".sam")
        log(f"inputs: {n_n} normal + {n_t} tumor alignments, {n_som} somatic SNVs, BAMs {os.path.getsize(d + '/normal.bam') / 1e9:.2f} + {os.path.getsize(d + '/tumor.bam') / 1e9:.2f} GB, built in {time.time() - t0:.0f} s")
        r0 = subprocess.run([ref_bin, "phase", "-s", "normal_in.vcf", "-b", "normal.bam", "-r", "ref.fa", "-t", str(a.threads), "-o", "normal_phased", "--ont", "--indels"], cwd=d, capture_output=True)
        assert r0.returncode == 0, r0.stderr[-300:]
        common = ["somatic_haplotag", "-s", "normal_phased.vcf", "-b", "normal.bam", "--tumor-snv-file", "tumor.vcf", "--tumor-bam-file", "tumor.bam", "-r", "ref.fa", "-t", str(a.threads)]
        tr, tc = [], []
        for _ in range(2):
            t0 = time.time(); r1 = subprocess.run([ref_bin] + common + ["-o", "ref_out"], cwd=d, capture_output=True); tr.append(time.time() - t0)
            t0 = time.time(); r2 = subprocess.run([cli] + common + ["-o", "gpu_out"], cwd=d, capture_output=True); tc.append(time.time() - t0)
            assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-300:], r2.stderr[-300:])
            log(f"reference {tr[-1]:.2f} s, longphase_amd {tc[-1]:.2f} s")
        out = {"sample": f"{a.contig_mb} Mb contig, normal 25x ({n_n} alignments) + tumor 50x ({n_t}) at 60 % purity, SNP + indel VCFs, {n_som} somatic SNVs, automatic purity estimation, -t {a.threads}, best of 2",
               "cli_wall_s": round(min(tc), 3), "reference_wall_s": round(min(tr), 3), "speedup": round(min(tr) / min(tc), 2),
               "tumor_reads_per_s_cli": n_t / min(tc), "tumor_reads_per_s_reference": n_t / min(tr),
               "identical_record_stream": digest(d + "/ref_out.bam") == digest(d + "/gpu_out.bam"),
               "identical_purity_report": open(d + "/ref_out_purity.out").read() == open(d + "/gpu_out_purity.out").read(),
               "cli_stages": r2.stderr.decode().strip().splitlines()[-1], "reference_tail": r1.stderr.decode().strip().splitlines()[-3:]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
