"""The two intermediate stages the REFERENCE BINARY exposes, as golden vectors (tests/golden/stage_*.npz, made by `make_golden.py --stages`
with oracle/_ref/longphase-s-ref): `phase --dot` = every connected pair of edgeConnectResult with its direction (PhasingGraph.cpp:402-409,
1031-1047) and `haplotag --log` = the votes, PQ and PS judgeHaplotype arrived at for every read (HaplotagProcess.cpp:177-237).  Here the ORACLE
is held against them (a12 / a13 and a17 / a18 pinned to the reference itself, not only through final PS / GT and tags); test_stage_goldens_gpu.py
does the same with the GPU's stage dumps."""
import os

import numpy as np
import pytest

import fixtures
import lps_oracle
from lps import abi
from lps.synth import Synth
from stage_util import connected_pairs

GOLD = os.path.join(os.path.dirname(__file__), "golden")
STAGE_PHASE = ["snp_ont", "indels", "two_blocks", "supp_light_dups", "params_a"]
STAGE_HAPLOTAG = ["snp_ont", "indels", "supp_tagged", "strict"]


def load_phase(name):
    kw, cli, over = fixtures.PHASE_FIXTURES[name]
    s = Synth(**kw)
    g = np.load(os.path.join(GOLD, f"stage_dot_{name}.npz"))
    assert str(g["digest"]) == fixtures.input_digest(s), "generator drift"
    return s, abi.default_params(**over), g["edges"]


@pytest.mark.parametrize("name", STAGE_PHASE)
def test_oracle_connected_pairs_equal_reference_dot(name):
    s, P, want = load_phase(name)
    V = abi.Variants(s.var_pos, s.var_ref, s.var_alt); R = abi.Reads.from_synth(s)
    out, d = lps_oracle.phase(P, V, s.ref, R, dump=True)
    n = int(d.c.n_nodes)
    got = connected_pairs(np.asarray(s.var_pos), d.node_var[:n], d.edge[:n], d.node_hp[:n], P.edge_threshold)
    assert got.shape == want.shape and np.array_equal(got, want)


def tag_rows(s, P, out):
    """rows of the reference's tag log from a haplotag result: one per alignment that reached judgeHaplotype (status 0), in BAM order"""
    m = out.status == 0
    return dict(read_start=np.asarray(s.ref_start)[m], hp=out.hp[m].astype(np.int8), ps=out.ps[m], h1=out.hp1[m], h2=out.hp2[m], pq=out.pq[m])


def load_haplotag(name):
    src, tag_cli, over = fixtures.HAPLOTAG_FIXTURES[name]
    kw, phase_cli, _ = fixtures.PHASE_FIXTURES[src]
    s = Synth(**kw)
    g = np.load(os.path.join(GOLD, f"stage_taglog_{name}.npz"))
    assert str(g["digest"]) == fixtures.input_digest(s), "generator drift"
    t = np.load(os.path.join(GOLD, f"haplotag_{name}.npz"))
    VT = abi.Variants(t["pos"], [x.encode() for x in t["ref"]], [x.encode() for x in t["alt"]], hp1_is_alt=t["hp1_is_alt"], phase_set=t["phase_set"])
    return s, abi.default_params(**over), VT, g


def check_rows(got, g):
    assert got["hp"].size == g["hp"].size, (got["hp"].size, g["hp"].size)
    for k in ("read_start", "h1", "h2", "hp", "ps"):
        assert np.array_equal(got[k], g[k]), k
    tagged = g["hp"] != 0                                               # the log prints the PQ of tagged reads (the reference leaves it unset otherwise)
    assert np.array_equal(got["pq"][tagged], g["pq"][tagged])


@pytest.mark.parametrize("name", STAGE_HAPLOTAG)
def test_oracle_votes_equal_reference_tag_log(name):
    s, P, VT, g = load_haplotag(name)
    out = lps_oracle.haplotag(P, VT, s.ref, abi.Reads.from_synth(s))
    check_rows(tag_rows(s, P, out), g)
