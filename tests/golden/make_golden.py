#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ by running the REAL reference binary
(oracle/_ref/longphase-s-ref, built by oracle/build_ref.sh from /root/reference) on generated inputs.

Run in the build container only (the reference cannot travel to the GPU box):
    python tests/golden/make_golden.py
Outputs (committed):
    tests/golden/phase_<name>.npz    variant positions + reference (PS, GT) per variant + input digest
    tests/golden/data/<name>.*       full inputs (FASTA, VCF, SAM.gz) of the tiny data fixtures + reference VCF
"""
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "longphase-s_amd"))
sys.path.insert(0, HERE)
from lps.synth import Synth  # noqa: E402
import fixtures  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "longphase-s-ref")
TEST_VIEW = os.path.join(ROOT, "oracle", "_ref", "test_view")


def parse_phased_vcf(path, var_pos):
    """-> (phase_set int32 [0 = '.'], gt uint8 [0 '0|1', 1 '1|0', 2 unphased]) aligned to var_pos."""
    by_pos = {}
    for line in open(path):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        fmt = f[8].split(":")
        smp = f[9].split(":")
        by_pos[int(f[1]) - 1] = (smp[fmt.index("GT")], smp[fmt.index("PS")])
    ps = np.zeros(len(var_pos), np.int32)
    gt = np.full(len(var_pos), 2, np.uint8)
    for i, p in enumerate(var_pos):
        g, s = by_pos[int(p)]
        if s != ".":
            ps[i] = int(s)
            assert g in ("0|1", "1|0"), g
            gt[i] = 0 if g == "0|1" else 1
        else:
            assert "|" not in g, (p, g)
    return ps, gt


def run_reference_phase(s, cli, workdir, chrom="chrS"):
    s.write_fasta(os.path.join(workdir, "ref.fa"), chrom)
    s.write_vcf(os.path.join(workdir, "in.vcf"), chrom)
    s.write_sam(os.path.join(workdir, "reads.sam"), chrom)
    subprocess.check_call([TEST_VIEW, "-b", "-x", "reads.bam.bai", "-p", "reads.bam", "reads.sam"], cwd=workdir,
                          stdout=subprocess.DEVNULL)
    cmd = [REF_BIN, "phase", "-s", "in.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", "1", "-o", "out"] + cli
    r = subprocess.run(cmd, cwd=workdir, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"reference failed rc={r.returncode}: {r.stderr[-2000:]}")
    return parse_phased_vcf(os.path.join(workdir, "out.vcf"), s.var_pos)


def main():
    assert os.path.exists(REF_BIN), "build the reference first: oracle/build_ref.sh"
    index = {}
    for name, (kw, cli, over) in fixtures.PHASE_FIXTURES.items():
        s = Synth(**kw)
        with tempfile.TemporaryDirectory() as d:
            ps, gt = run_reference_phase(s, cli, d)
        np.savez_compressed(os.path.join(HERE, f"phase_{name}.npz"), var_pos=np.array(s.var_pos), phase_set=ps, gt=gt)
        index[name] = dict(digest=fixtures.input_digest(s), n_var=int(s.n_variants), n_reads=int(s.n_reads),
                           n_phased=int((ps != 0).sum()), n_blocks=int(len(set(ps[ps != 0].tolist()))), cli=cli)
        print(name, index[name])
        s.close()
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    for name, (kw, cli, over) in fixtures.DATA_FIXTURES.items():
        s = Synth(**kw)
        with tempfile.TemporaryDirectory() as d:
            ps, gt = run_reference_phase(s, cli, d)
            for fn, dst in (("ref.fa", f"{name}.fa"), ("in.vcf", f"{name}.vcf"), ("out.vcf", f"{name}.ref_phased.vcf")):
                shutil.copy(os.path.join(d, fn), os.path.join(HERE, "data", dst))
            with open(os.path.join(d, "reads.sam"), "rb") as fi, gzip.GzipFile(os.path.join(HERE, "data", f"{name}.sam.gz"), "wb", mtime=0) as fo:
                shutil.copyfileobj(fi, fo)
        index["data:" + name] = dict(digest=fixtures.input_digest(s), n_var=int(s.n_variants), n_reads=int(s.n_reads),
                                     n_phased=int((ps != 0).sum()), cli=cli)
        print(name, index["data:" + name])
        s.close()
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
