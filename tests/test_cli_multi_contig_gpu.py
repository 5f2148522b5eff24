"""Multi-contig end-to-end check of the CLI (both input paths): four contigs in one BAM (one without VCF records), unplaced reads at the end,
supplementary alignments, indels.  `phase --indels` must write the reference's VCF byte for byte; `haplotag` on that VCF must write the reference's
record stream (contigs in VCF-header order, unplaced reads dropped, records of the variant-free contig copied untouched)."""
import hashlib
import json
import os
import subprocess

import pytest

import fixtures
import util

HERE = os.path.dirname(os.path.abspath(__file__))
CLI = os.path.join(HERE, "..", "longphase-s_amd", "cli", "longphase_amd")
pytestmark = pytest.mark.gpu


def _body(path):
    return [l for l in open(path).read().split("\n") if not l.startswith("##commandline=") and not l.startswith("##longphaseVersion=")]


@pytest.mark.parametrize("inflate", ["gpu", "host", "gpu_indexed", "gpu_indexed_3workers", "gpu_hostdeflate", "gpu_indexed_1pergroup"])
def test_multi_contig_phase_then_haplotag(inflate, tmp_path):
    gold = json.load(open(os.path.join(HERE, "golden", "cli_multi_contig.json")))
    d = str(tmp_path)
    assert util.make_multi_contig(d, fixtures.MULTI_CONTIG_FIXTURE) == gold["digests"], "generator drift"
    util.add_stale_tags(d + "/multi.sam", d + "/tagged_in.sam")
    util.write_bam(d + "/tagged_in.sam", d + "/reads.bam", block=30000)
    extra = ["--host-inflate"] if inflate == "host" else ["--gpu-inflate"]
    tag_extra = ["--host-deflate"] if inflate == "gpu_hostdeflate" else []
    if inflate.startswith("gpu_indexed"):
        util.write_bai(d + "/reads.bam")                              # only one contig's blocks are resident at a time
    workers = ["--gpus", "3"] if inflate.endswith("3workers") else []
    if inflate.endswith("1pergroup"):
        extra = extra + ["--group-bytes", "1"]                        # every contig is uploaded and inflated on its own   # three contexts (on the one GPU of the test box), contigs dealt longest-first
    r = subprocess.run([CLI, "phase", "-s", "multi.vcf", "-b", "reads.bam", "-r", "multi.fa", "-t", "3", "-o", "phased", "--ont", "--indels"] + extra + workers,
                       cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert ("(indexed)" in r.stderr) == inflate.startswith("gpu_indexed")
    ref_vcf = os.path.join(HERE, "golden", "data", "multi_contig.ref_phased.vcf")
    assert _body(d + "/phased.vcf") == _body(ref_vcf)
    r = subprocess.run([CLI, "haplotag", "-s", ref_vcf, "-b", "reads.bam", "-r", "multi.fa", "-t", "3", "-o", "tagged"] + extra + tag_extra + workers,
                       cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    if workers:
        assert "3 workers" in r.stderr                                 # haplotag --gpus: contigs dealt onto three contexts, written in VCF-header order
    else:
        assert ("(indexed)" in r.stderr) == inflate.startswith("gpu_indexed")
    text, refs, recs = util.bam_sections(d + "/tagged.bam")
    assert [l for l in text.split("\n") if l and not l.startswith("@PG")] == gold["header_without_pg"]
    got = util.bam_record_tags(recs)
    want = [(q, f, p, [tuple(t) for t in tg]) for q, f, p, tg in gold["tags"]]
    assert len(got) == gold["n_records"]
    for g, w in zip(got, want):
        assert g == w
    assert hashlib.sha256(recs).hexdigest() == gold["records_sha256"]
