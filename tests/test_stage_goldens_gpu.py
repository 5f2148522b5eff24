"""The reference's own intermediate-stage outputs (`phase --dot`, `haplotag --log`; see test_stage_goldens.py) against the GPU's stage dumps:
connected pairs + direction from lps_dump_graph / lps_dump_votes, per-read votes / HP / PS / PQ from lps_haplotag_chromosome."""
import numpy as np
import pytest

from lps import abi, hip
from stage_util import connected_pairs
from test_stage_goldens import STAGE_HAPLOTAG, STAGE_PHASE, check_rows, load_haplotag, load_phase, tag_rows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", STAGE_PHASE)
def test_gpu_connected_pairs_equal_reference_dot(name):
    s, P, want = load_phase(name)
    V = abi.Variants(s.var_pos, s.var_ref, s.var_alt); R = abi.Reads.from_synth(s)
    with hip.Context(0, P) as ctx:
        ctx.phase(V, s.ref, R)
        nodes, edge = ctx.dump_graph()
        hp, _ = ctx.dump_votes()
    got = connected_pairs(np.asarray(s.var_pos), nodes, edge, hp, P.edge_threshold)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("name", STAGE_HAPLOTAG)
def test_gpu_votes_equal_reference_tag_log(name):
    s, P, VT, g = load_haplotag(name)
    with hip.Context(0, P) as ctx:
        out = ctx.haplotag(VT, s.ref, abi.Reads.from_synth(s))
    check_rows(tag_rows(s, P, out), g)
