#!/usr/bin/env python3
"""The whole-node sample of bench.py on its own: 24 contigs x 10 Mb at 50x in one BAM (12.4 GB) - the reference at -t 24 (every thread with a contig),
clock P (decoded alignments in pinned memory -> results) and clock E (this repository's command line on the same files).  Run on the GPU box:
    python3 profiles/e2e_whole_node.py [n_contigs=24] [contig_mb=10] > gpurun_out/e2e_whole_node.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "longphase-s_amd"))
import bench  # noqa: E402
from lps import abi  # noqa: E402

os.environ.setdefault("LPS_E2E_HAPLOTAG", "1")        # also `haplotag` end to end on the same files (the reference takes about a minute)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
out = bench.whole_node_baseline(0, abi.default_params(), min(24, ncpu), 201, n_contigs=n, contig_mb=mb)
out["nproc"] = ncpu
print(json.dumps(out, indent=1))
