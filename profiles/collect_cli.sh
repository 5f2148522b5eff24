#!/bin/bash
# kernel stats of the CLI (phase + haplotag) on a chr20-30x BAM, from HEAD
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/cliprof; rm -rf $D; mkdir -p $D; cd $D
python3 - <<PY
import sys, os, subprocess
sys.path.insert(0, os.path.join("$ROOT", "longphase-s_amd"))
from lps.synth import Synth
s = Synth(seed=7101, contig_len=64_444_167, n_snp=60_000, coverage=30.0, n_threads=16)
s.write_fasta("ref.fa"); s.write_vcf("in.vcf"); s.write_sam("reads.sam")
subprocess.check_call([os.path.join("$ROOT", "oracle/_ref/test_view"), "-@", "16", "-b", "-x", "reads.bam.bai", "-p", "reads.bam", "reads.sam"], stdout=subprocess.DEVNULL)
os.remove("reads.sam")
PY
export TMPDIR=/tmp LPS_CLI_NO_FAST_EXIT=1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_cli_phase -o stats -- $ROOT/longphase-s_amd/cli/longphase_amd phase -s in.vcf -b reads.bam -r ref.fa -t 16 -o gpu --ont > /dev/null 2> $ROOT/gpurun_out/prof_cli_phase.err
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_cli_haplotag -o stats -- $ROOT/longphase-s_amd/cli/longphase_amd haplotag -s gpu.vcf -b reads.bam -r ref.fa -t 16 -o tagged > /dev/null 2> $ROOT/gpurun_out/prof_cli_haplotag.err
echo cli profiles done
