"""`python3 bench.py --gpus N` launched PLAINLY (no torchrun, WORLD_SIZE unset) must run N ranks: it starts them itself as child processes before
making any GPU call.  The test box has one GPU, so the ranks share it (RCCL refuses that: the collective is skipped and said so in the line);
what is rehearsed is the launch path, the gloo control plane, the per-rank contig deal and the aggregation - and that every rank's result is
oracle-checked (phase and haplotag) with a non-zero exit code when it is not."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("n", [2, 4])
def test_plain_launch_runs_n_ranks(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "mini_wgs", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--ctx-per-gpu", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == n
    assert d["parity_checked"] is True and d["parity"]["contigs_checked"] == 8 and not d["parity"]["haplotag_mismatching"]
    assert len(d["rank_loads"]) == n and all(x["bases_of_contigs"] > 0 for x in d["rank_loads"])
    assert "skipped" in d["rccl"] or d["rccl"].get("n_ranks_in_communicator") == n
