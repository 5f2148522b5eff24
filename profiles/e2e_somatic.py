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
    ap.add_argument("--contigs", type=int, default=1, help="contigs of --contig-mb each in the pair (one BAM per sample, indexed): more than one lets --group-bytes cut the pair into groups")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--group-bytes", type=int, default=0, help="an extra run of longphase_amd with this --group-bytes (e.g. 1 = one contig per group) and one with --no-index; outputs must not change")
    a = ap.parse_args()
    from lps.synth import Synth
    cli = os.path.join(ROOT, "longphase-s_amd", "cli", "longphase_amd")
    tv = os.path.join(ROOT, "oracle", "_ref", "test_view"); ref_bin = os.path.join(ROOT, "oracle", "_ref", "longphase-s-ref")
    L = a.contig_mb * 1_000_000
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        t0 = time.time()
        n_t = n_n = n_som = 0
        names = ["chrS%d" % (k + 1) for k in range(a.contigs)] if a.contigs > 1 else ["chrS"]
        heads = {"normal.sam": [], "tumor.sam": [], "normal_in.vcf": [], "tumor.vcf": []}; bodies = {k: open(os.path.join(d, k + ".body"), "w") for k in heads}
        with open(d + "/ref.fa", "w") as fa:
            for k, nm in enumerate(names):
                genome = dict(contig_len=L, n_snp=L // 1000, n_threads=a.threads, somatic_every=6000.0, indel_var_frac=0.15, seed=5201 + 10 * k)
                N = Synth(**dict(genome, coverage=25.0, read_seed=5211 + 10 * k, tumor_purity=0.0)); T = Synth(**dict(genome, coverage=50.0, read_seed=5212 + 10 * k, tumor_purity=0.6))
                N.write_fasta(d + "/one.fa", nm); fa.write(open(d + "/one.fa").read())
                N.write_vcf(d + "/one_normal_in.vcf", nm); N.write_sam(d + "/one_normal.sam", nm); T.write_sam(d + "/one_tumor.sam", nm); T.write_vcf_tumor(d + "/one_tumor.vcf", nm, with_germline=True)
                n_t += int(T.n_reads); n_n += int(N.n_reads); n_som += int(N.n_somatic); N.close(); T.close()
                for key, one in (("normal.sam", "one_normal.sam"), ("tumor.sam", "one_tumor.sam"), ("normal_in.vcf", "one_normal_in.vcf"), ("tumor.vcf", "one_tumor.vcf")):
                    is_head = (lambda l: l.startswith("@")) if key.endswith(".sam") else (lambda l: l.startswith("#"))
                    for ln in open(os.path.join(d, one)):
                        if is_head(ln):
                            if ln not in heads[key]:
                                heads[key].append(ln)
                        else:
                            bodies[key].write(ln)
                    os.remove(os.path.join(d, one))
        for key in heads:
            bodies[key].close()
            h = heads[key]
            if key.endswith(".sam"):
                h = [x for x in h if x.startswith("@HD")][:1] + [x for x in h if x.startswith("@SQ")] + [x for x in h if not x.startswith(("@HD", "@SQ"))]
            else:
                h = [x for x in h if x.startswith("##")] + [x for x in h if x.startswith("#CHROM")][:1]
            with open(os.path.join(d, key), "w") as f:
                f.write("".join(h))
                with open(os.path.join(d, key + ".body")) as b:
                    for chunk in iter(lambda: b.read(1 << 24), ""):
                        f.write(chunk)
            os.remove(os.path.join(d, key + ".body"))
        for smp in ("normal", "tumor"):
            subprocess.check_call([tv, "-@", str(a.threads), "-b", "-x", smp + ".bam.bai", "-p", smp + ".bam", smp + ".sam"], cwd=d, stdout=subprocess.DEVNULL); os.remove(d + "/" + smp + ".sam")
        log(f"inputs: {a.contigs} contig(s) x {a.contig_mb} Mb, {n_n} normal + {n_t} tumor alignments, {n_som} somatic SNVs, BAMs {os.path.getsize(d + '/normal.bam') / 1e9:.2f} + {os.path.getsize(d + '/tumor.bam') / 1e9:.2f} GB, built in {time.time() - t0:.0f} s")
        r0 = subprocess.run([ref_bin, "phase", "-s", "normal_in.vcf", "-b", "normal.bam", "-r", "ref.fa", "-t", str(a.threads), "-o", "normal_phased", "--ont", "--indels"], cwd=d, capture_output=True)
        assert r0.returncode == 0, r0.stderr[-300:]
        common = ["somatic_haplotag", "-s", "normal_phased.vcf", "-b", "normal.bam", "--tumor-snv-file", "tumor.vcf", "--tumor-bam-file", "tumor.bam", "-r", "ref.fa", "-t", str(a.threads)]
        tr, tc = [], []
        for k in range(2):
            if k == 0 or a.contigs * a.contig_mb <= 48:                  # (the reference takes a minute at 160 Mb: once)
                t0 = time.time(); r1 = subprocess.run([ref_bin] + common + ["-o", "ref_out"], cwd=d, capture_output=True); tr.append(time.time() - t0)
                assert r1.returncode == 0, r1.stderr[-300:]
            t0 = time.time(); r2 = subprocess.run([cli] + common + ["-o", "gpu_out"], cwd=d, capture_output=True); tc.append(time.time() - t0)
            assert r2.returncode == 0, r2.stderr[-300:]
            log(f"reference {tr[-1]:.2f} s, longphase_amd {tc[-1]:.2f} s")
        want_bam, want_pur = digest(d + "/ref_out.bam"), open(d + "/ref_out_purity.out").read()
        extra = []
        for tag, args in ((("groups", ["--group-bytes", str(a.group_bytes)]), ("whole_files", ["--no-index"])) if a.group_bytes else ()):
            t0 = time.time(); rx = subprocess.run([cli] + common + ["-o", "x_" + tag] + args, cwd=d, capture_output=True); wall = time.time() - t0
            assert rx.returncode == 0, rx.stderr[-300:]
            err = rx.stderr.decode().strip().splitlines()
            extra.append(dict(run=tag, args=args, wall_s=round(wall, 3), identical_record_stream=digest(d + "/x_" + tag + ".bam") == want_bam,
                              identical_purity_report=open(d + "/x_" + tag + "_purity.out").read() == want_pur,
                              group_line=[ln for ln in err if ln.startswith("contig groups")][:1], stages=err[-1][:600]))
            os.remove(d + "/x_" + tag + ".bam")
        err2 = r2.stderr.decode().strip().splitlines()
        out = {"sample": f"{a.contigs} contig(s) x {a.contig_mb} Mb, normal 25x ({n_n} alignments) + tumor 50x ({n_t}) at 60 % purity, SNP + indel VCFs, {n_som} somatic SNVs, automatic purity estimation, -t {a.threads}; both BAMs indexed",
               "cli_wall_s": round(min(tc), 3), "reference_wall_s": round(min(tr), 3), "speedup": round(min(tr) / min(tc), 2),
               "tumor_reads_per_s_cli": n_t / min(tc), "tumor_reads_per_s_reference": n_t / min(tr),
               "identical_record_stream": digest(d + "/gpu_out.bam") == want_bam,
               "identical_purity_report": open(d + "/gpu_out_purity.out").read() == want_pur,
               "cli_group_line": [ln for ln in err2 if ln.startswith("contig groups")][:1], "cli_stages": err2[-1], "other_group_settings": extra,
               "reference_tail": r1.stderr.decode().strip().splitlines()[-3:]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
