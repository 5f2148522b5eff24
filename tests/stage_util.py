"""What the reference's intermediate-stage outputs (`phase --dot`, `haplotag --log`) look like when derived from OUR stage dumps."""
import numpy as np


def connected_pairs(var_pos, node_var, edge, node_hp, edge_threshold):
    """The rows of `phase --dot` (tests/golden/make_golden.py parse_dot) from a dump of the graph: edgeConnectResult (PhasingGraph.cpp:286-418)
    visits the nodes in position order, skips those it gives no haplotype (node_hp == 0: a gap beyond `distance`, or a tie behind the last
    connection) and, for every other node but the last, asks findBestEdgePair (:166-228) about each of the next A nodes: a pair is CONNECTED
    (and written to the .dot file with its direction) when its four cells do not tie and their similarity ratio does not exceed the threshold.
    SNP / indel graphs only (no MOD rows: their 0.3 threshold is not restated here)."""
    N = len(node_var); A = edge.shape[1]
    rows = []
    for i in range(N - 1):
        if node_hp[i] == 0:
            continue
        for k in range(A):
            j = i + 1 + k
            if j >= N:
                break
            rr, ra, ar, aa = (np.float32(x) for x in edge[i, k])
            para = np.float32(rr + aa); cross = np.float32(ra + ar)
            if para == cross:
                continue
            with np.errstate(divide="ignore", invalid="ignore"):
                esr = float(min(para, cross)) / float(max(para, cross))
            if esr > edge_threshold:
                continue
            rows.append((var_pos[node_var[i]], var_pos[node_var[j]], 1 if para > cross else 2))
    return np.array(rows, np.int32).reshape(-1, 3)
