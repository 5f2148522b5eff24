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
@pytest.mark.parametrize("inflate", ["gpu", "host"])
@pytest.mark.parametrize("name", sorted(DATA_FIXTURES))
def test_cli_phase_matches_reference_vcf(name, inflate, tmp_path):
    assert os.path.exists(CLI), "build the CLI first: make -C longphase-s_amd cli"
    bam = str(tmp_path / (name + ".bam"))
    assert write_bam(os.path.join(DATA, name + ".sam.gz"), bam) > 0
    flags = DATA_FIXTURES[name][1] + (["--host-inflate"] if inflate == "host" else ["--gpu-inflate"])
    prefix = str(tmp_path / "out")
    r = subprocess.run([CLI, "phase", "-s", os.path.join(DATA, name + ".vcf"), "-b", bam, "-r", os.path.join(DATA, name + ".fa"),
                        "-o", prefix, "-t", "4"] + flags, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got, want = _body(prefix + ".vcf"), _body(os.path.join(DATA, name + ".ref_phased.vcf"))
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(DATA_FIXTURES))
def test_cli_phase_dot_matches_reference(name, tmp_path):
    """--dot: <chr>.dot in the working directory, byte for byte the file the reference wrote (PhasingGraph.cpp:402-409, 1031-1047): every connected
    pair of edgeConnectResult in the order it visits them, with its direction."""
    import gzip
    d = str(tmp_path)
    assert write_bam(os.path.join(DATA, name + ".sam.gz"), d + "/r.bam") > 0
    r = subprocess.run([CLI, "phase", "-s", os.path.join(DATA, name + ".vcf"), "-b", "r.bam", "-r", os.path.join(DATA, name + ".fa"), "-o", "out", "-t", "4", "--dot"]
                       + DATA_FIXTURES[name][1], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want = gzip.open(os.path.join(DATA, name + ".ref.dot.gz"), "rt").read()
    assert want.count("->") > 1000
    assert open(d + "/chrS.dot").read() == want
    assert _body(d + "/out.vcf") == _body(os.path.join(DATA, name + ".ref_phased.vcf"))


@pytest.mark.gpu
def test_cli_phase_table_through_the_collective(tmp_path):
    """The --gpus path of the table on the one-GPU box: rank 0's packed SNP table goes through lps_comm_bcast_to_device (a one-rank RCCL
    communicator) and reaches the context as device pointers (lps_set_variants_device) - same VCF as the reference's."""
    name = "tiny_indel"
    bam = str(tmp_path / (name + ".bam"))
    assert write_bam(os.path.join(DATA, name + ".sam.gz"), bam) > 0
    prefix = str(tmp_path / "out")
    r = subprocess.run([CLI, "phase", "-s", os.path.join(DATA, name + ".vcf"), "-b", bam, "-r", os.path.join(DATA, name + ".fa"), "-o", prefix, "-t", "4"] + DATA_FIXTURES[name][1],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, LPS_CLI_BCAST_ALWAYS="1"))
    assert r.returncode == 0, r.stderr
    assert "SNP table broadcast" in r.stderr
    assert _body(prefix + ".vcf") == _body(os.path.join(DATA, name + ".ref_phased.vcf"))


@pytest.mark.gpu
def test_cli_phase_deepsomatic_output(tmp_path):
    """--deepsomatic_output: <prefix>_preprocessed.vcf and the phased VCF written from it equal the reference's."""
    import gzip
    d = str(tmp_path)
    open(d + "/ds.vcf", "w").write(gzip.open(os.path.join(DATA, "cli_deepsomatic.ds.vcf.gz"), "rt").read())
    write_bam(os.path.join(DATA, "tiny_snp.sam.gz"), d + "/r.bam")
    r = subprocess.run([CLI, "phase", "-s", "ds.vcf", "-b", "r.bam", "-r", os.path.join(DATA, "tiny_snp.fa"), "-o", "o", "--ont", "--deepsomatic_output"],
                       cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(d + "/o_preprocessed.vcf").read() == gzip.open(os.path.join(DATA, "cli_deepsomatic.out_preprocessed.vcf.gz"), "rt").read()
    strip = lambda t: [l for l in t.split("\n") if not l.startswith("##commandline=") and not l.startswith("##longphaseVersion=")]
    assert strip(open(d + "/o.vcf").read()) == strip(gzip.open(os.path.join(DATA, "cli_deepsomatic.out.vcf.gz"), "rt").read())


@pytest.mark.gpu
def test_cli_phase_indel_quality(tmp_path):
    """--indels --indelQuality 25: het indel records below the threshold are dropped (src/phase/ParsingBam.cpp:325-340), listed in
    <prefix>_removed_indels.log and marked INDEL_QUAL_FILTERED in the output VCF (:467-473, 578-623); SNP records are not touched."""
    import gzip
    d = str(tmp_path)
    gz = lambda n: gzip.open(os.path.join(DATA, "cli_indelq." + n + ".gz"), "rt").read()
    open(d + "/iq.vcf", "w").write(gz("iq.vcf"))
    write_bam(os.path.join(DATA, "tiny_indel.sam.gz"), d + "/r.bam")
    r = subprocess.run([CLI, "phase", "-s", "iq.vcf", "-b", "r.bam", "-r", os.path.join(DATA, "tiny_indel.fa"), "-o", "o", "--ont", "--indels", "--indelQuality", "25"],
                       cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(d + "/o_removed_indels.log").read() == gz("out_removed_indels.log")
    strip = lambda t: [l for l in t.split("\n") if not l.startswith("##commandline=") and not l.startswith("##longphaseVersion=")]
    assert strip(open(d + "/o.vcf").read()) == strip(gz("out.vcf"))
    assert "INDEL_QUAL_FILTERED" in open(d + "/o.vcf").read()


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


@pytest.mark.gpu
def test_cli_phase_two_bam_files(tmp_path):
    """-b given twice (reads split over two files, names ranked across both): same VCF as with one file (ParsingBam.cpp:1252)."""
    import gzip
    name = "tiny_snp"
    lines = gzip.open(os.path.join(DATA, name + ".sam.gz"), "rt").read().splitlines(True)
    head = [l for l in lines if l.startswith("@")]; recs = [l for l in lines if not l.startswith("@")]
    for k in (0, 1):
        with open(tmp_path / f"p{k}.sam", "w") as f:
            f.writelines(head + recs[k::2])
        write_bam(str(tmp_path / f"p{k}.sam"), str(tmp_path / f"p{k}.bam"))
    with open(tmp_path / "all.sam", "w") as f:
        f.writelines(head + recs[0::2] + recs[1::2])
    prefix = str(tmp_path / "two")
    r = subprocess.run([CLI, "phase", "-s", os.path.join(DATA, name + ".vcf"), "-b", str(tmp_path / "p0.bam"), "-b", str(tmp_path / "p1.bam"), "-r",
                        os.path.join(DATA, name + ".fa"), "-o", prefix, "--ont"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    # oracle for the two-file order: the library on the concatenated alignments (p0 then p1), names ranked together
    import numpy as np
    import lps_oracle
    import util
    from lps import abi
    R, names = util.parse_sam(str(tmp_path / "all.sam"))
    V = util.parse_vcf_variants(os.path.join(DATA, name + ".vcf"))
    # the two files are each coordinate-sorted; the reference processes file 0 completely, then file 1
    ref = util.parse_fasta(os.path.join(DATA, name + ".fa"))
    want, _ = lps_oracle.phase(abi.default_params(), V, ref, R)
    got = {int(l.split("\t")[1]) - 1: l.rstrip("\n").split("\t")[9] for l in open(prefix + ".vcf") if not l.startswith("#")}
    for i in range(V.n):
        smp = got[int(V.pos[i])]
        if want.phase_set[i]:
            assert smp.endswith(":%d" % want.phase_set[i]) and smp.startswith("1|0" if want.gt[i] else "0|1")
        else:
            assert smp.endswith(":.")
