"""N>1 path on CPU: world_size-2 gloo run of the contig-sharding layer (lps/shard.py).  The per-contig compute is the
CPU oracle here (tests may use it); on the GPU box the same layer drives hip.Context.  The sharded result must equal
the single-process result byte for byte (shards are independent)."""
import os
import subprocess
import sys

import numpy as np

from lps import shard

WORKER = r'''
import os, sys, pickle
import numpy as np
root = sys.argv[1]
for p in ("longphase-s_amd", "oracle", "tests", "tests/golden"):
    sys.path.insert(0, os.path.join(root, p))
import torch.distributed as dist
from lps import abi, shard
from lps.synth import Synth
import lps_oracle
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
CONTIGS = [dict(seed=40 + i, contig_len=120_000 + 40_000 * i, n_snp=150 + 50 * i, coverage=10.0, n_threads=1) for i in range(5)]
def compute(i):
    s = Synth(**CONTIGS[i]); V = abi.Variants(s.var_pos, s.var_ref, s.var_alt); R = abi.Reads.from_synth(s)
    out, _ = lps_oracle.phase(abi.default_params(), V, s.ref, R)
    return (out.phase_set.tobytes(), out.gt.tobytes())
res = shard.run_sharded(len(CONTIGS), [c["n_snp"] for c in CONTIGS], compute, rank, world, dist)
dist.barrier()
if rank == 0:
    pickle.dump(res, open(sys.argv[2], "wb"))
dist.destroy_process_group()
'''


def test_lpt_schedule_is_balanced_and_complete():
    w = [248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 58, 64, 46, 50, 156, 57]
    sched = shard.lpt_schedule(w, 8)
    assert sorted(i for r in sched for i in r) == list(range(24))
    loads = [sum(w[i] for i in r) for r in sched]
    assert max(loads) <= 1.15 * sum(w) / 8
    assert shard.lpt_schedule(w, 8) == sched


def test_world2_gloo_equals_single_process(tmp_path):
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    outs = {}
    for world in (1, 2):
        out = tmp_path / f"res{world}.pkl"
        procs = []
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29611 + world))
            procs.append(subprocess.Popen([sys.executable, str(script), root, str(out)], env=env))
        for p in procs:
            assert p.wait(timeout=600) == 0
        import pickle
        outs[world] = pickle.load(open(out, "rb"))
    assert outs[1] == outs[2]
    assert len(outs[2]) == 5 and all(len(a) > 0 for a, _ in outs[2])


GPU_WORKER = r'''
import os, sys, pickle
import numpy as np
root = sys.argv[1]
for p in ("longphase-s_amd", "tests", "tests/golden"):
    sys.path.insert(0, os.path.join(root, p))
from lps import abi, hip, shard
from lps.synth import Synth
hip.load()                                     # product library before torch (binds the ROCm runtime it was built against)
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
CONTIGS = [dict(seed=40 + i, contig_len=120_000 + 40_000 * i, n_snp=150 + 50 * i, coverage=10.0, n_threads=1) for i in range(5)]
ctx = hip.Context(0, abi.default_params())     # both ranks share the one GPU of the test box: two contexts, two streams
def compute(i):
    s = Synth(**CONTIGS[i]); V = abi.Variants(s.var_pos, s.var_ref, s.var_alt); R = abi.Reads.from_synth(s)
    out = ctx.phase(V, s.ref, R)
    return (out.phase_set.tobytes(), out.gt.tobytes())
res = shard.run_sharded(len(CONTIGS), [c["n_snp"] for c in CONTIGS], compute, rank, world, dist)
dist.barrier()
if rank == 0:
    pickle.dump(res, open(sys.argv[2], "wb"))
ctx.close()
dist.destroy_process_group()
'''


import pytest  # noqa: E402


@pytest.mark.gpu
def test_world2_gloo_with_hip_contexts_equals_oracle(tmp_path):
    """The same sharding layer with hip.Context as the per-contig compute: two ranks (two contexts on the one GPU of the box), contigs dealt
    longest-first, results gathered in contig order - must equal the single-rank run and the CPU oracle byte for byte."""
    import pickle
    import lps_oracle
    from lps import abi
    from lps.synth import Synth
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    script = tmp_path / "gpu_worker.py"
    script.write_text(GPU_WORKER)
    outs = {}
    for world in (1, 2):
        out = tmp_path / f"gres{world}.pkl"
        procs = []
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29631 + world))
            procs.append(subprocess.Popen([sys.executable, str(script), root, str(out)], env=env))
        for p in procs:
            assert p.wait(timeout=600) == 0
        outs[world] = pickle.load(open(out, "rb"))
    assert outs[1] == outs[2]
    for i in range(5):
        s = Synth(seed=40 + i, contig_len=120_000 + 40_000 * i, n_snp=150 + 50 * i, coverage=10.0, n_threads=1)
        V = abi.Variants(s.var_pos, s.var_ref, s.var_alt); R = abi.Reads.from_synth(s)
        want, _ = lps_oracle.phase(abi.default_params(), V, s.ref, R)
        assert outs[2][i] == (want.phase_set.tobytes(), want.gt.tobytes()), f"contig {i} differs from the oracle"
