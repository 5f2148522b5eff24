"""End-to-end drop-in check: `longphase_amd phase` on the committed FASTA/VCF/reads of tests/golden/data
must write the same VCF as the reference binary did (tests/golden/data/*.ref_phased.vcf), except the two header lines
that carry the program version and command line (src/phase/ParsingBam.cpp:485-486)."""
import os
import subprocess

import pytest

from fixtures import DATA_FIXTURES
from util import write_bam

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")
CLI = os.path.join(HERE, "..", "longphase-s_amd", "cli", "longphase_amd")


def _body(path):
    return [l for l in open(path).read().split("\n") if not l.startswith("##commandline=") and not l.startswith("##longphaseVersion=")]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(DATA_FIXTURES))
def test_cli_phase_matches_reference_vcf(name, tmp_path):
    assert os.path.exists(CLI), "build the CLI first: make -C longphase-s_amd cli"
    bam = str(tmp_path / (name + ".bam"))
    assert write_bam(os.path.join(DATA, name + ".sam.gz"), bam) > 0
    flags = DATA_FIXTURES[name][1]
    prefix = str(tmp_path / "out")
    r = subprocess.run([CLI, "phase", "-s", os.path.join(DATA, name + ".vcf"), "-b", bam, "-r", os.path.join(DATA, name + ".fa"),
                        "-o", prefix, "-t", "4"] + flags, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got, want = _body(prefix + ".vcf"), _body(os.path.join(DATA, name + ".ref_phased.vcf"))
    assert got == want


@pytest.mark.gpu
def test_cli_rewrites_previously_phased_vcf(tmp_path):
    """Feeding the reference's own phased output back in (old PS keys, phased GTs) must reproduce it: exercises the
    PS strip / GT un-phase rules of SnpParser::writeLine (src/phase/ParsingBam.cpp:505-571)."""
    name = "tiny_snp"
    bam = str(tmp_path / "r.bam")
    write_bam(os.path.join(DATA, name + ".sam.gz"), bam)
    prefix = str(tmp_path / "again")
    r = subprocess.run([CLI, "phase", "-s", os.path.join(DATA, name + ".ref_phased.vcf"), "-b", bam, "-r", os.path.join(DATA, name + ".fa"),
                        "-o", prefix, "--ont"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = [l for l in _body(prefix + ".vcf")]
    want = _body(os.path.join(DATA, name + ".ref_phased.vcf"))
    assert got == want
