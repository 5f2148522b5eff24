#!/usr/bin/env python3
"""One-off end-to-end comparison on a MULTI-contig input (run on the GPU box from the repo root; needs oracle/_ref):
N contigs of L Mb at 30x in one BAM - the shape of a whole genome in miniature, where the reference can use one compute thread per contig.
    python3 profiles/e2e_multi_contig.py [N=8] [L_Mb=8] > gpurun_out/e2e_multi_contig.json"""
import hashlib, gzip, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "longphase-s_amd"))
from lps.synth import Synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = min(16, os.cpu_count() or 8)
REF = os.path.join(ROOT, "oracle", "_ref", "longphase-s-ref"); TV = os.path.join(ROOT, "oracle", "_ref", "test_view"); CLI = os.path.join(ROOT, "longphase-s_amd", "cli", "longphase_amd")


def body(p):
    return [l for l in open(p) if not l.startswith("##commandline=") and not l.startswith("##longphaseVersion=")]


def digest(path):
    h = hashlib.sha256()
    with gzip.open(path, "rb") as f:
        head = f.read(8); f.read(int.from_bytes(head[4:8], "little")); nref = int.from_bytes(f.read(4), "little")
        for _ in range(nref):
            ln = int.from_bytes(f.read(4), "little"); f.read(ln + 4)
        for b in iter(lambda: f.read(1 << 24), b""):
            h.update(b)
    return h.hexdigest()


with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
    sq, vhead, n_snp, n_reads = [], [], 0, 0
    with open(d + "/ref.fa", "w") as fa, open(d + "/reads.sam.body", "w") as sb, open(d + "/in.vcf.body", "w") as vb:
        for k in range(N):
            name = "chr%02d" % (k + 1)
            s = Synth(contig_len=L * 1_000_000, n_snp=int(930 * L), coverage=30.0, seed=900 + k, n_threads=T)
            s.write_fasta(d + "/one.fa", name); fa.write(open(d + "/one.fa").read())
            s.write_sam(d + "/one.sam", name); s.write_vcf(d + "/one.vcf", name)
            for line in open(d + "/one.sam"):
                if line.startswith("@SQ"): sq.append(line)
                elif not line.startswith("@"): sb.write(name + "_" + line)
            for line in open(d + "/one.vcf"):
                if line.startswith("##contig"): vhead.append(line)
                elif not line.startswith("#"): vb.write(line)
            n_snp += s.n_variants; n_reads += s.n_reads; s.close()
    with open(d + "/reads.sam", "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n" + "".join(sq)); f.write(open(d + "/reads.sam.body").read())
    with open(d + "/in.vcf", "w") as f:
        f.write("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n" + "".join(vhead) +
                '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n')
        f.write(open(d + "/in.vcf.body").read())
    subprocess.check_call([TV, "-b", "-x", "reads.bam.bai", "-p", "reads.bam", "reads.sam"], cwd=d, stdout=subprocess.DEVNULL)
    for fn in ("reads.sam", "reads.sam.body", "in.vcf.body", "one.sam", "one.fa", "one.vcf"):
        os.remove(d + "/" + fn)
    out = {"contigs": N, "contig_mb": L, "het_snps": n_snp, "alignments": n_reads, "bam_bytes": os.path.getsize(d + "/reads.bam"), "threads": T}

    def timed(cmd, n=3):
        ts = []
        for _ in range(n):
            t0 = time.time(); r = subprocess.run(cmd, cwd=d, capture_output=True, env=dict(os.environ, LPS_DEBUG="1")); ts.append(time.time() - t0)
            assert r.returncode == 0, r.stderr[-500:]
        err = r.stderr.decode().strip().splitlines()
        return sorted(ts)[len(ts) // 2], " || ".join([l for l in err if l.startswith("[lps_")] + err[-1:])
    subprocess.run([REF, "phase", "-s", "in.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(T), "-o", "ref", "--ont"], cwd=d, capture_output=True)   # warm the page cache
    tr, _ = timed([REF, "phase", "-s", "in.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(T), "-o", "ref", "--ont"])
    tc, st = timed([CLI, "phase", "-s", "in.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(T), "-o", "gpu", "--ont"])
    out["phase"] = {"reference_wall_s": round(tr, 3), "cli_wall_s": round(tc, 3), "speedup": round(tr / tc, 2), "identical_vcf": body(d + "/ref.vcf") == body(d + "/gpu.vcf"), "cli_stages": st}
    tr, _ = timed([REF, "haplotag", "-s", "ref.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(T), "-o", "ref_tag"], 2)
    tc, st = timed([CLI, "haplotag", "-s", "ref.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(T), "-o", "gpu_tag"], 2)
    out["haplotag"] = {"reference_wall_s": round(tr, 3), "cli_wall_s": round(tc, 3), "speedup": round(tr / tc, 2), "identical_record_stream": digest(d + "/ref_tag.bam") == digest(d + "/gpu_tag.bam"),
                       "output_bytes": {"cli": os.path.getsize(d + "/gpu_tag.bam"), "reference": os.path.getsize(d + "/ref_tag.bam")}, "cli_stages": st}
    prof = os.environ.get("LPS_ROCPROF_DIR")                          # optional: kernel trace of the two CLI runs (the CLI must leave through exit handlers for the report)
    if prof:
        env = dict(os.environ, LPS_CLI_NO_FAST_EXIT="1", TMPDIR="/tmp")
        for what, cmd in (("phase", [CLI, "phase", "-s", "in.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(T), "-o", "gpu", "--ont"]),
                          ("haplotag", [CLI, "haplotag", "-s", "ref.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", str(T), "-o", "gpu_tag"])):
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(prof, what), "-o", what, "--"] + cmd, cwd=d, env=env, capture_output=True)
            out[what]["rocprof_rc"] = r.returncode
    print(json.dumps(out, indent=1))
