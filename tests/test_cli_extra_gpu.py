"""End-to-end drop-in check of `phase --sv-file --mod-file`: three contigs in one BAM (the last without SNP records), SV and modcall VCFs that
hold every kind of record the reference's readers drop or mangle (tests/golden/make_golden.py make_cli_extra lists them).  `longphase_amd phase`
must write the reference's three VCFs - <prefix>.vcf, <prefix>_SV.vcf, <prefix>_mod.vcf - byte for byte (but for the version / command lines)."""
import gzip
import json
import os
import subprocess

import pytest

import fixtures
import util

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")
CLI = os.path.join(HERE, "..", "longphase-s_amd", "cli", "longphase_amd")
pytestmark = pytest.mark.gpu


def _body(text):
    return [l for l in text.split("\n") if not l.startswith("##commandline=") and not l.startswith("##longphaseVersion=")]


def _gz(name):
    return gzip.open(os.path.join(DATA, "cli_extra." + name + ".gz"), "rt").read()


@pytest.mark.parametrize("mode", ["gpu", "host", "gpu_indexed_3workers", "gz_inputs"])
def test_cli_phase_with_sv_and_mod_files(mode, tmp_path):
    gold = json.load(open(os.path.join(HERE, "golden", "cli_extra.json")))
    d = str(tmp_path)
    assert util.make_multi_contig(d, fixtures.CLI_EXTRA_FIXTURE, unmapped=0) == gold["digests"], "generator drift"
    util.write_bam(d + "/multi.sam", d + "/reads.bam", block=30000)
    sv, mod = "sv.vcf", "mod.vcf"
    if mode == "gz_inputs":                                             # SVParser / METHParser read .gz files too (ParsingBam.cpp:921-928, 1657-1664)
        sv, mod = "sv.vcf.gz", "mod.vcf.gz"
        for fn in (sv, mod):
            with gzip.open(os.path.join(d, fn), "wt") as f:
                f.write(_gz(fn[:-3]))
    else:
        for fn in (sv, mod):
            open(os.path.join(d, fn), "w").write(_gz(fn))
    flags = ["--host-inflate"] if mode == "host" else ["--gpu-inflate"]
    if mode.startswith("gpu_indexed"):
        util.write_bai(d + "/reads.bam")
        flags += ["--gpus", "3"]
    r = subprocess.run([CLI, "phase", "-s", "multi.vcf", "-b", "reads.bam", "-r", "multi.fa", "-t", "3", "-o", "out", "--ont", "--sv-file", sv, "--mod-file", mod] + flags,
                       cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for fn in ("out.vcf", "out_SV.vcf", "out_mod.vcf"):
        got, want = _body(open(os.path.join(d, fn)).read()), _body(_gz(fn))
        assert got == want, fn
    assert gold["phased_records"]["out_SV.vcf"] > 10 and gold["phased_records"]["out_mod.vcf"] > 100


def test_cli_refuses_rows_the_reference_would_spin_on(tmp_path):
    """A modcall record on the 0-based position of an SV row: the reference's extraction loop never ends on the first read that spans it; the CLI
    stops with the library's message."""
    d = str(tmp_path)
    util.make_multi_contig(d, fixtures.CLI_EXTRA_FIXTURE, unmapped=0)
    util.write_bam(d + "/multi.sam", d + "/reads.bam", block=30000)
    sv_lines = _gz("sv.vcf").split("\n")
    rec = next(l for l in sv_lines if l.startswith("chrA") and "\tsv0\t" in l).split("\t")
    mod = [l for l in _gz("mod.vcf").split("\n") if l.startswith("#")]
    mod.append("chrA\t%d\t.\tC\t<MOD>\t.\tPASS\tRS=P;MR=chrA_r000000001;NR=;\tGT:MD:UD\t0/1:1:0" % int(rec[1]))      # VCF POS of the SV = its 0-based row + 1 = this record's row
    open(d + "/sv.vcf", "w").write("\n".join(sv_lines)); open(d + "/mod.vcf", "w").write("\n".join(mod) + "\n")
    r = subprocess.run([CLI, "phase", "-s", "multi.vcf", "-b", "reads.bam", "-r", "multi.fa", "-o", "out", "--ont", "--sv-file", "sv.vcf", "--mod-file", "mod.vcf"],
                       cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "more than one of the SNP / SV / MOD tables" in r.stderr
