"""Contig sharding across ranks (one process per GPU).

Phasing never crosses chromosomes (src/phase/PhasingProcess.cpp:113-173 handles each independently and merges by map
insert, src/shared/Util.cpp:7-12) and haplotag scoring is per read, so the path shards by contig with NO data-path
collective: contigs are scheduled longest-processing-time-first onto ranks (the reference uses
`#pragma omp parallel for schedule(dynamic)` over chromosomes), every rank processes its contigs on its own GPU, and
the per-contig results are gathered to rank 0 in contig order.  torch.distributed is only the control plane
(gather of small result objects, barrier); backend "gloo" on CPU, "nccl" (= RCCL) when tensors live on the GPU.
"""
from typing import Callable, List, Sequence


def lpt_schedule(weights: Sequence[float], n_ranks: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of contigs (by weight, e.g. SNP count) to ranks.
    Deterministic: ties broken by contig index; returns per-rank lists in processing order."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    load = [0.0] * n_ranks
    out: List[List[int]] = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += weights[i]
    return out


def run_sharded(n_contigs: int, weights: Sequence[float], compute: Callable[[int], object], rank: int, world: int, dist=None):
    """Every rank calls compute(contig) for its contigs; rank 0 returns the list of results in contig order
    (other ranks return None).  `dist` is an initialised torch.distributed module (or None when world == 1)."""
    mine = lpt_schedule(weights, world)[rank]
    local = {i: compute(i) for i in mine}
    if world == 1 or dist is None:
        return [local[i] for i in range(n_contigs)]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in gathered:
        merged.update(part)
    assert len(merged) == n_contigs, "a contig was not processed by any rank"
    return [merged[i] for i in range(n_contigs)]
