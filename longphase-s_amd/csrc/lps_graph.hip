// lps_graph.hip — haplotype-graph kernels of `phase` (gfx950, wave64).
//
// Replaces (reference file:line, relative to /root/reference/):
//   Clip::getCNVInterval (x2)            src/phase/PhasingGraph.cpp:1103-1227  -> k_clip_keys + sort; the state machine itself is replayed on the host (lps_abi.hip replay_cnv)
//   VairiantGraph::addEdge overlap filter src/phase/PhasingGraph.cpp:707-781   -> k_name_keys + sort + k_group_* + k_overlap_filter
//   addEdge type tagging / node set       :793-846                             -> k_mark_nodes + scan + k_graph_obs
//   addEdge pair loop + addSubEdge        :848-888, :25-70                     -> k_merge_plan/k_merge_multi + k_node_count/scatter + k_edges (orders the node lists)
//   findBestEdgePair                      :166-228                             -> epilogue of k_edges (edge-info byte)
//   edgeConnectResult + Onelongcase       :286-474, :251-283                   -> k_vote_scan
//   readCorrection + exportResult         :891-1029, :1049-1077                -> k_block_size + k_read_correction + k_final
//
// Order-exactness: edge cells are fp32 sums of +1.0f / (float)(double(x)+w) whose value depends on
// the order of the contributing reads (lexicographic read-name order in the reference).  k_edges therefore owns
// each cell in ONE register of ONE lane and replays the contributions of the node's reads in (name rank, index)
// order taken from a radix-sorted node-major list: no atomics, every cell is written exactly once.
#include <cstring>
#include <algorithm>
#include <type_traits>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "lps_graph.h"
#include "lps_stdsort.h"
#include <cstdlib>

// ================================================================================================ clips, read names, overlap filter
// One launch after the extraction:
//   * workgroups [0, nb_reads): thread per alignment.  The alignments of one read NAME are linked into a list - name_head[name] holds the last one
//     that arrived (+1), name_link[alignment] the one before it (+1, 0 = none) - with one atomicExch per alignment that has observations: no sort
//     of 600 k keys, no head flags, no scan (the reference's mergeReadMap, PhasingGraph.cpp:697, groups by name too; only ~2 % of the names hold
//     more than one alignment).  `name` is dense: the caller's name ids when they are (CLI and bench hand over ranks), else ranks made by
//     launch_dense_names.  Also: alignments with observations, the longest row.
//   * workgroup nb_reads: observation slots reserved by the extraction, over the arenas.
//   * the workgroups after it: clip events of ops before the op at which get_snp returned early (:1453-1455,1559-1561), compacted (one atomic per
//     workgroup) into sort keys (pos << 1 | front/back).
#define NAMES_B 1024      // alignments (and clip slots per pass) of a workgroup: a quarter of the atomics on the counters' single words that 256 would make
__global__ __launch_bounds__(NAMES_B) void k_name_link(int n_reads, const uint32_t *name, const RowDesc *rows, uint32_t *name_head, uint32_t *name_link,
                                                   LpsCounters *cnt, const unsigned long long *arena_ctr, unsigned long long arena_size,
                                                   ClipView C, unsigned long long *keys, int nb_reads, uint32_t *clip_tab) {
    __shared__ unsigned s_wcnt[NAMES_B / 64], s_wmax[NAMES_B / 64], s_base;
    const int w = threadIdx.x >> 6;
    if ((int)blockIdx.x < nb_reads) {
        const int r = blockIdx.x * blockDim.x + threadIdx.x;
        const int n = r < n_reads ? rows[r].cnt : 0;
        const bool kept = n > 0;
        if (r < n_reads) name_link[r] = kept ? atomicExch(&name_head[name[r]], (uint32_t)r + 1u) : 0u;
        // one atomic per workgroup: atomics on ONE word are served one after the other (~11 ns each)
        const unsigned long long m = __ballot(kept);
        const int mx = wave_max(n);
        if (lane_id() == 0) { s_wcnt[w] = (unsigned)__popcll(m); s_wmax[w] = (unsigned)max(mx, 0); }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tot = 0, big = 0;
            for (int q = 0; q < NAMES_B / 64; ++q) { tot += s_wcnt[q]; big = max(big, s_wmax[q]); }
            if (tot) atomicAdd(&cnt->n_kept, tot);
            if (big) atomicMax(&cnt->max_row, big);
        }
        return;
    }
    if ((int)blockIdx.x == nb_reads) {
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            unsigned long long v = l < LPS_ARENAS ? arena_ctr[l * 8] : 0ull;
            if (v > arena_size) atomicOr(&cnt->err, (unsigned)LPS_ERR_OBS_OVERFLOW);
            unsigned long long mx = v;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
            v = wave_sum(v);
            if (l == 0) { cnt->obs_total = v; cnt->arena_max = mx; }
        }
        return;
    }
    const unsigned n_ev = C.fixed + min(*C.n_ev, C.capacity - C.fixed);    // the jobs' own slots (unused ones hold read = -1), then what the general walker appended
    const unsigned cb = blockIdx.x - nb_reads - 1, n_cb = gridDim.x - nb_reads - 1;
    // a workgroup takes ONE contiguous share of the slots: counts what it keeps, reserves once (atomics on one word are served one after the other:
    // one per workgroup, not one per 256 slots), then goes over its share again (from the caches) and writes the keys
    const unsigned per = (n_ev + n_cb - 1) / n_cb, lo = min(n_ev, cb * per), hi = min(n_ev, lo + per);
    auto key_of = [&](unsigned e, unsigned long long &key) -> bool {
        const ClipEv ev = C.ev[e];
        key = ((unsigned long long)(unsigned)ev.pos << 1) | (unsigned)(ev.opidx_fb & 1);
        return ev.read >= 0 && (ev.opidx_fb >> 1) < rows[max(ev.read, 0)].fail;
    };
    unsigned mine = 0; unsigned long long key;
    for (unsigned e = lo + threadIdx.x; e < hi; e += blockDim.x) mine += key_of(e, key) ? 1u : 0u;
    const unsigned incl = (unsigned)wave_incl_scan_dpp((int)mine);
    if (lane_id() == 63) s_wcnt[w] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned tot = 0; for (int q = 0; q < NAMES_B / 64; ++q) tot += s_wcnt[q]; s_base = tot ? atomicAdd(&cnt->n_clips, tot) : 0u; }
    __syncthreads();
    unsigned off = s_base + incl - mine; for (int q = 0; q < w; ++q) off += s_wcnt[q];
    // Clip::getCNVInterval can only emit an interval when some position holds five or more front clips or five or more back clips (replay_cnv,
    // lps_abi.hip) - nearly never.  Every key is counted in two hashed tables; the smaller of its two counts bounds the key's multiplicity from
    // above (count-min), the largest such bound goes to the counters: below 5 the host skips the key sort, the copy and the replay altogether.
    unsigned worst = 0;
    for (unsigned e = lo + threadIdx.x; e < hi; e += blockDim.x) if (key_of(e, key)) {
        keys[off++] = key;
        const unsigned h1 = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> (64 - LPS_CLIP_TAB_BITS)), h2 = (unsigned)((key * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull) >> (64 - LPS_CLIP_TAB_BITS));
        const unsigned a = atomicAdd(&clip_tab[h1], 1u) + 1u, b = atomicAdd(&clip_tab[(1u << LPS_CLIP_TAB_BITS) + h2], 1u) + 1u;
        worst = max(worst, min(a, b));
    }
    worst = (unsigned)wave_max((int)worst);
    if (lane_id() == 0 && worst > 1u) atomicMax(&cnt->clip_mult, worst);     // (1 is the floor whenever there is a clip: nothing to tell)
    if (cb == 0 && threadIdx.x == 0 && *C.n_ev > C.capacity - C.fixed) atomicOr(&cnt->err, (unsigned)LPS_ERR_CLIP_OVERFLOW);
}

// Thread per alignment; the thread of the LAST alignment linked under a name that holds several (a few per workgroup) takes the group: its members
// in BAM order (the list is in arrival order: every member is found by another walk - groups have two or three members) go to mm_r, where the
// merged-row kernels find them, and the reference's overlap filter of several alignments of one read (:707-781) is replayed on them one after the
// other.  A deleted alignment's observations leave the per-variant counts the extraction took (var_del).
// 1 024 alignments per workgroup: its two reservations are RETURNING atomics on single words, which the L2 serves one after the other at ~11 ns each
//  - with 256 alignments per workgroup they alone were 55 us of this kernel at chr1-50x
#define GROUPS_B 1024
__global__ __launch_bounds__(GROUPS_B) void k_groups(int n_reads, const uint32_t *name, const RowDesc *rows, const ObsRec *obs, const int32_t *vpos, double overlap_threshold,
                                                const uint32_t *name_head, const uint32_t *name_link, LpsCounters *cnt,
                                                uint32_t *mm_r, uint32_t *stack, uint32_t *mg_start, uint32_t *mg_cnt, uint32_t *mg_name, uint8_t *deleted, uint32_t *var_del) {
    __shared__ unsigned s_k[GROUPS_B / 64], s_n[GROUPS_B / 64], s_mx[GROUPS_B / 64], s_base_k, s_base_n;
    const int r = blockIdx.x * blockDim.x + threadIdx.x, w = threadIdx.x >> 6;
    bool owner = false; int k = 0; uint32_t id = 0;
    if (r < n_reads && rows[r].cnt > 0) {
        id = name[r];
        if (name_head[id] == (uint32_t)r + 1u && name_link[r] != 0u) { owner = true; for (uint32_t x = (uint32_t)r + 1u; x; x = name_link[x - 1]) ++k; }
    }
    const int incl = wave_incl_scan_dpp(k);
    const unsigned long long om = __ballot(owner);
    const int kmx = wave_max(k);
    if (lane_id() == 63) s_k[w] = (unsigned)incl;
    if (lane_id() == 0) { s_n[w] = (unsigned)__popcll(om); s_mx[w] = (unsigned)kmx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned K = 0, N = 0, big = 0;
        for (int q = 0; q < GROUPS_B / 64; ++q) { K += s_k[q]; N += s_n[q]; big = max(big, s_mx[q]); }
        s_base_k = K ? atomicAdd(&cnt->mm_total, K) : 0u; s_base_n = N ? atomicAdd(&cnt->n_multi, N) : 0u;
        if (big) atomicMax(&cnt->max_group, big);
    }
    __syncthreads();
    unsigned base = s_base_k + (unsigned)(incl - k), qi = s_base_n + (unsigned)__popcll(om & lanemask_lt());
    for (int q = 0; q < w; ++q) { base += s_k[q]; qi += s_n[q]; }
    if (owner) {
    {
        uint32_t prev = 0;                                      // members in ascending alignment index (+1): the smallest above the last one taken
        for (int t = 0; t < k; ++t) {
            uint32_t best = 0xffffffffu;
            for (uint32_t x = (uint32_t)r + 1u; x; x = name_link[x - 1]) if (x > prev && x < best) best = x;
            mm_r[base + t] = best - 1u; prev = best;
        }
    }
    mg_start[qi] = base; mg_cnt[qi] = (uint32_t)k; mg_name[qi] = id;
    // ---- overlap filter (:707-781): the reference's sequential rule over the group's alignments; `stack` is scratch aligned with mm_r
    uint32_t *kept = stack + base; int nk = 0; int second = 0;
    auto fpos = [&](uint32_t a) { return vpos[obs[rows[a].off].var]; };
    auto lpos = [&](uint32_t a) { return vpos[obs[rows[a].off + rows[a].cnt - 1].var]; };
    for (int t = 0; t < k; ++t) {
        const uint32_t a = mm_r[base + t];
        const int fp = fpos(a), lp = lpos(a);
        bool del = false;
        while (0 <= fp && fp <= second) {                    // alignRange starts as {0,0} (:712-716)
            if (lp < second) { del = true; break; }
            const int pre = nk - 1;
            if (pre < 0) break;
            const uint32_t pr = kept[pre];
            const int ps = fpos(pr), pe = lpos(pr);
            const double ovS = max(ps, fp), ovE = min(pe, lp);
            if (ovS > ovE) break;
            const double ovLen = ovE - ovS + 1;
            const double alS = max(pe, lp), alE = min(ps, fp);
            const double span = alS - alE + 1;
            const double ratio = ovLen / span;
            if (ratio >= overlap_threshold) {
                const int len1 = pe - ps + 1, len2 = lp - fp + 1;
                if (len2 <= len1) { del = true; break; }
                deleted[pr] = 1; --nk;
                second = (pre > 0) ? lpos(kept[pre - 1]) : fp;
            } else break;
        }
        second = lp;
        if (del) deleted[a] = 1; else kept[nk++] = a;
    }
    }
    // ---- the observations of the deleted alignments leave their variants' counts.  One lane per group decided; the counts are taken back by the
    //      WHOLE wave, one deleted alignment after the other (a lane on its own walked its alignments' 20 - 60 observations one dependent load at a
    //      time: most of this kernel's 0.13 ms at chr1-50x)
    if (!var_del) return;
    const int l = lane_id();
    for (int t_next = 0;;) {
        uint32_t a = 0xffffffffu;
        if (owner) { while (t_next < k) { const uint32_t x = mm_r[base + t_next++]; if (deleted[x]) { a = x; break; } } }
        unsigned long long pending = __ballot(a != 0xffffffffu);
        if (!pending) break;
        while (pending) {
            const int src = (int)__builtin_ctzll(pending); pending &= pending - 1ull;
            const uint32_t aa = (uint32_t)__builtin_amdgcn_readlane((int)a, src);
            const uint32_t off = rows[aa].off; const int n = rows[aa].cnt;
            for (int j = l; j < n; j += 64) atomicAdd(&var_del[obs[off + j].var], 1u);
        }
    }
}

// ---- name ids that are not dense (the ABI allows any order-preserving integer): ranks by one sort.  keys = (name id << 32 | alignment).
__global__ void k_dense_keys(int n_reads, const uint32_t *name_id, unsigned long long *keys) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) keys[r] = (unsigned long long)name_id[r] << 32 | (unsigned)r;
}
__global__ void k_dense_heads(const unsigned long long *skeys, int n_reads, uint32_t *head) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_reads) head[s] = (s == 0 || (skeys[s] >> 32) != (skeys[s - 1] >> 32)) ? 1u : 0u;
}
__global__ void k_dense_ids(const unsigned long long *skeys, const uint32_t *head, const uint32_t *gidx, int n_reads, uint32_t *dense) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_reads) dense[(uint32_t)skeys[s]] = gidx[s] + head[s] - 1;
}

// ---- rows by sub-wave groups.  A row holds ~20-30 observations, so a 64-lane wave per row runs with a third of its lanes; the per-row passes below
// give every row a group of ROW_G lanes instead (two rows per wave): the same instruction stream serves two rows.
#ifndef ROW_G
#define ROW_G 32
#endif
#define ROWS_PER_WAVE (64 / ROW_G)
#define ROWS_PER_BLOCK (4 * ROWS_PER_WAVE)
__device__ __forceinline__ unsigned long long group_ballot(bool pred, int grp) { return (__ballot(pred) >> (grp * ROW_G)) & ((ROW_G == 64) ? ~0ull : ((1ull << ROW_G) - 1ull)); }

// ================================================================================================ CNV filter
// The four CNV mismatch-rate passes (PhasingGraph.cpp:520-692).  The reference carries ONE interval cursor from read to read (and, in the last
// pass, from observation to observation) over an interval list that holds every interval twice (getCNVInterval runs twice): cs = [s_0..s_(K-1),
// s_0..s_(K-1)], each half ascending and its intervals disjoint.  Which intervals a read "visits" therefore depends on all reads before it.
// What it depends on is ONE BIT: the half h the carried cursor sits in.  Inside a half the reference first walks the cursor back to the last
// interval that starts at or before the read (or to index 0 of the whole list when the read starts before s_0), and intervals that end before
// the read can neither be hit nor change where the forward scan ends - so for any offset inside the half the visited intervals that can hold an
// observation, the erasures and the cursor handed on are the same (checked exhaustively against the reference's loops on random interval sets
// when this was written; tests/test_phase_gpu.py cnv_* fixtures pin it end to end).  A read is therefore a function {0,1} -> {0,1} on the half;
// functions compose associatively, one device-wide scan gives every read its entry half, and counting / erasing are then embarrassingly
// parallel: each read replays the reference's own loops from the representative cursor of its half.  No bound on the number of intervals
// (the reference's cnvVec is an unbounded std::vector).
__device__ __forceinline__ int cnv_ub(const int32_t *cs, int K, int p) {   // number of interval starts <= p (first half)
    int lo = 0, hi = K;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (cs[m] <= p) lo = m + 1; else hi = m; }
    return lo;
}
// the cursor the reference's back-up loop (`while (ci > 0 && cs[ci] > rs) --ci`) reaches from anywhere in half h, taken at the representative
__device__ __forceinline__ int cnv_entry(const int32_t *cs, int K, int h, int rs) {
    const int A = cnv_ub(cs, K, rs);
    return A == 0 ? 0 : h * K + A - 1;
}

// where the reference's forward scan (`i = ci; while (i < nc && cs[i] <= re) ++i;`) ends, given U = number of first-half starts <= re: inside the
// first half it stops at the first start beyond re; when every start is <= re it runs on through the whole second half
__device__ __forceinline__ int cnv_walk_end(int ci, int K, int U) {
    if (ci < K) return U < K ? max(ci, U) : 2 * K;
    return max(ci, K + U);
}

// kept alignments in BAM order -> dense list (reads with observations that survived the overlap filter)
__global__ void k_cnv_list(const LpsCounters *cnt, int n_reads, const RowDesc *rows, const uint8_t *deleted, uint32_t *flag) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    flag[r] = (cnt->n_cnv != 0 && rows[r].cnt > 0 && !deleted[r]) ? 1u : 0u;
}
__global__ void k_cnv_compact(const LpsCounters *cnt, int n_reads, const uint32_t *flag, const uint32_t *idx, uint32_t *list, uint32_t *n_list) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    if (flag[r]) list[idx[r]] = r;
    if (r == n_reads - 1) *n_list = idx[r] + flag[r];
}

// thread per kept alignment: its transfer function on the cursor's half, bit h = half of the cursor it hands on when entered in half h.
// PASS4 = false: calculateCnvMismatchRate / aggregateCnvReadMismatchRate (same cursor walk); PASS4 = true: filterHighMismatchVariants.
// thread per kept alignment: transfer function of passes 1/2 (calculateCnvMismatchRate / aggregateCnvReadMismatchRate walk the cursor alike)
__global__ void k_cnv_fn12(const LpsCounters *cnt, const uint32_t *list, const uint32_t *n_list, const RowDesc *rows,
                           const ObsRec *obs, const int32_t *vpos, const int32_t *cs, uint8_t *fn) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *n_list) return;
    const int nc = (int)cnt->n_cnv, K = nc / 2;
    const uint32_t r = list[k]; const uint32_t off = rows[r].off; const int n = rows[r].cnt;
    const int rs = vpos[obs[off].var], re = vpos[obs[off + n - 1].var];
    const int A0 = cnv_ub(cs, K, rs), U0 = cnv_ub(cs, K, re);
    unsigned f = 0;
    for (int h = 0; h < 2; ++h) {
        const int ci = A0 == 0 ? 0 : h * K + A0 - 1;         // cnv_entry
        const int i = cnv_walk_end(ci, K, U0);
        f |= (unsigned)((i > 0 ? i - 1 : 0) >= K) << h;
    }
    fn[k] = (uint8_t)f;
}

// One observation of filterHighMismatchVariants' scan (`i = ci; while (i < nc && cs[i] <= p) { if (p in [cs[i], ce[i]] && rate >= 0.7) break; ++i; }
// ci = i > 0 ? i - 1 : 0`) without touching the interval list: U = number of first-half starts <= p, j = index of the interval that holds p
// (-1: none), hot = rate >= 0.7.  cs[i] <= p holds exactly for the indices i < U of the first half and K <= i < K + U of the second.
__device__ __forceinline__ int cnv_obs_step(int ci, int K, int U, int j, bool hot, bool *erased) {
    const bool can = hot && j >= 0;
    int i; bool brk = false;
    if (ci < K) {
        if (ci >= U) i = ci;
        else if (can && j >= ci) { i = j; brk = true; }
        else if (U < K) i = U;
        else if (can) { i = K + j; brk = true; }             // every start is <= p: the scan runs on into the second half
        else i = 2 * K;
    } else {
        const int a = ci - K;
        if (a >= U) i = ci;
        else if (can && j >= a) { i = K + j; brk = true; }
        else i = K + U;
    }
    *erased = brk;
    return i > 0 ? i - 1 : 0;
}

// filterHighMismatchVariants, a lane group per kept alignment.  ERASE = false: the alignment's transfer function on the cursor's half (both halves
// simulated); ERASE = true: the erasures, from the known entry half.  The group's lanes load the row's positions and mismatch rates side by side
// (a lane walking its row alone is one chain of dependent misses, ~3 per observation); rows without an observation that can stop the scan
// (rate >= 0.7) - nearly all - move the cursor exactly like passes 1/2 and erase nothing; the others replay the reference's loop on the
// loaded values.
template <bool ERASE>
__global__ __launch_bounds__(256) void k_cnv_rows(const LpsCounters *cnt, const uint32_t *list, const uint32_t *n_list, const uint8_t *pre, const RowDesc *rows,
        ObsRec *obs, const int32_t *vpos, const int32_t *cs, const int32_t *ce,
                                                  const double *miss, uint8_t *fn, uint32_t *var_del2) {
    const int l = lane_id(), grp = l / ROW_G, sl = l % ROW_G;
    const unsigned k = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE + grp;
    if (k >= *n_list) return;
    const int nc = (int)cnt->n_cnv, K = nc / 2;
    const uint32_t r = list[k]; const uint32_t off = rows[r].off; const int n = rows[r].cnt;
    const int rs = vpos[obs[off].var], re = vpos[obs[off + n - 1].var];
    const int A0 = cnv_ub(cs, K, rs), U0 = cnv_ub(cs, K, re);
    const bool touch = U0 > A0 || (A0 > 0 && ce[A0 - 1] >= rs);          // an interval intersects the row
    int c0 = A0 == 0 ? 0 : A0 - 1, c1 = A0 == 0 ? 0 : K + A0 - 1;        // cnv_entry of half 0 / half 1
    if (ERASE) { c0 = (pre[k] & 1) ? c1 : c0; }
    bool any_hot = false;
    if (touch) {
        for (int q0 = 0; q0 < n; q0 += ROW_G) {
            const int q = q0 + sl;
            int v = -1, p = 0; bool hot = false;
            if (q < n) { v = obs[off + q].var; p = vpos[v]; hot = miss[v] >= 0.7; }
            const unsigned long long hm = group_ballot(hot, grp);
            if (hm == 0ull && !any_hot) continue;                        // nothing can stop the scan yet: the cursor is brought up when it first matters (below)
            if (!any_hot && q0 > 0) {                                    // first hot chunk: bring the cursor(s) up to the end of the chunk before it
                const int pp = vpos[obs[off + q0 - 1].var]; const int Up = cnv_ub(cs, K, pp);
                int i = cnv_walk_end(c0, K, Up); c0 = i > 0 ? i - 1 : 0;
                if (!ERASE) { i = cnv_walk_end(c1, K, Up); c1 = i > 0 ? i - 1 : 0; }
            }
            any_hot = true;
            // per observation, side by side: how many intervals start at or before it, and which one holds it
            int Uq = 0, jq = -1;
            if (q < n) { Uq = cnv_ub(cs, K, p); if (Uq > 0 && p <= ce[Uq - 1]) jq = Uq - 1; }
            const int m = min(ROW_G, n - q0);
            bool mine = false;                                           // this lane's observation is erased
            for (int t = 0; t < m; ++t) {                                // the reference's scan, one observation after the other, on the loaded values
                const int src = grp * ROW_G + t;
                const int Ut = __shfl(Uq, src), jt = __shfl(jq, src); const bool ht = (hm >> t) & 1ull;
                bool e0, e1;
                c0 = cnv_obs_step(c0, K, Ut, jt, ht, &e0);
                if (!ERASE) c1 = cnv_obs_step(c1, K, Ut, jt, ht, &e1);
                if (ERASE && e0 && sl == t) mine = true;
            }
            if (ERASE && mine) { obs[off + q].var = -1 - v; atomicAdd(&var_del2[v], 1u); }
        }
    }
    if (ERASE) return;
    if (!any_hot) { int i = cnv_walk_end(c0, K, U0); c0 = i > 0 ? i - 1 : 0; i = cnv_walk_end(c1, K, U0); c1 = i > 0 ? i - 1 : 0; }   // (from the first hot chunk on every observation was replayed)
    if (sl == 0) fn[k] = (uint8_t)((unsigned)(c0 >= K) | ((unsigned)(c1 >= K) << 1));
}

struct CnvCompose {   // (a then b): bit h of the result = b(a(h)); identity 0b10
    __host__ __device__ uint8_t operator()(uint8_t a, uint8_t b) const { return (uint8_t)(((b >> (a & 1)) & 1) | (((b >> ((a >> 1) & 1)) & 1) << 1)); }
};

// thread per kept alignment: calculateCnvMismatchRate + aggregateCnvReadMismatchRate with the known entry half.  mm[read][interval] counts the
// ALT observations inside the interval once per visit of the interval (an interval is visited in both halves when the scan runs across the
// middle of the doubled list), and every visit then appends that count to the per-(position, allele) lists.
__global__ void k_cnv_count(const LpsCounters *cnt, const uint32_t *list, const uint32_t *n_list, const uint8_t *pre, const RowDesc *rows, const ObsRec *obs, const int32_t *vpos, const int32_t *cs,
                            const int32_t *ce, unsigned long long *agg_sum, int32_t *agg_cnt) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *n_list) return;
    const int nc = (int)cnt->n_cnv, K = nc / 2;
    const uint32_t r = list[k]; const uint32_t off = rows[r].off; const int n = rows[r].cnt;
    const int rs = vpos[obs[off].var], re = vpos[obs[off + n - 1].var];
    const int A0 = cnv_ub(cs, K, rs), U0 = cnv_ub(cs, K, re);
    const int ci = A0 == 0 ? 0 : (pre[k] & 1) * K + A0 - 1;            // cnv_entry
    const int i_end = cnv_walk_end(ci, K, U0);
    // intervals that can hold an observation of this read: those that start at or before re and do not end before rs
    for (int j = max(0, A0 - 1); j < U0; ++j) {
        const int s0 = cs[j], e0 = ce[j];
        if (e0 < rs) continue;
        const int visits = (j >= ci && j < min(i_end, K) ? 1 : 0) + (K + j >= max(ci, K) && K + j < i_end ? 1 : 0);
        if (!visits) continue;
        // observations inside [s0, e0]: a range of the position-sorted row, found by two searches; the loops over it have no data-dependent exit,
        // so their loads overlap (a row walked with an early exit is one chain of dependent misses)
        int qa = 0, qb = n;
        { int lo = 0, hi = n; while (lo < hi) { const int m = (lo + hi) >> 1; if (vpos[obs[off + m].var] < s0) lo = m + 1; else hi = m; } qa = lo; }
        { int lo = qa, hi = n; while (lo < hi) { const int m = (lo + hi) >> 1; if (vpos[obs[off + m].var] <= e0) lo = m + 1; else hi = m; } qb = lo; }
        int alt = 0;
        for (int q = qa; q < qb; ++q) alt += aq_allele((uint16_t)obs[off + q].aq);
        if (!alt) continue;
        const unsigned long long mm = (unsigned long long)alt * (unsigned long long)visits;
        for (int q = qa; q < qb; ++q) {
            const int v = obs[off + q].var; const int al = aq_allele((uint16_t)obs[off + q].aq);
            atomicAdd(&agg_sum[(size_t)v * 2 + al], mm * (unsigned long long)visits); atomicAdd(&agg_cnt[(size_t)v * 2 + al], visits);
        }
    }
}

// thread per variant: calculateAverageMismatchRate.  The reference scans the list from index 0 for every position (it never moves the cursor in
// this pass); both halves hold the same disjoint intervals, so "is p inside one of the visited intervals" is one binary search.
__global__ void k_cnv_miss(const LpsCounters *cnt, int n_var, const int32_t *vpos, const int32_t *cs, const int32_t *ce,
                           const unsigned long long *agg_sum, const int32_t *agg_cnt, double *miss) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_var) return;
    const int nc = (int)cnt->n_cnv, K = nc / 2;
    double m = -1.0;
    if (nc && agg_cnt[(size_t)v * 2] > 0 && agg_cnt[(size_t)v * 2 + 1] > 0) {
        const int p = vpos[v]; const int u = cnv_ub(cs, K, p);
        if (u > 0 && p <= ce[u - 1]) {
            const double a = (double)agg_sum[(size_t)v * 2] / (double)agg_cnt[(size_t)v * 2];
            const double c = (double)agg_sum[(size_t)v * 2 + 1] / (double)agg_cnt[(size_t)v * 2 + 1];
            if (a != 0 && c != 0) m = c / (a + c);
        }
    }
    miss[v] = m;
}

// ================================================================================================ nodes
// Workgroups are dealt round-robin over the 8 XCDs (one L2 each): workgroup b of a grid of 8k takes unit (b % 8) * k + b / 8, so that an XCD walks ONE
// contiguous eighth of the units and what neighbouring units share (list entries, rows of packed words) meets in one L2.

// Every observation was counted where it was made: the extraction's atomicAdd on var_cnt[variant] RETURNED the observation's rank inside the
// variant's list (kept in bits 10..31 of ObsRec.aq), observations that were dropped afterwards - alignments deleted by the overlap filter, entries
// erased by the CNV filter - were counted again in var_del / var_del2.  A variant is a graph node when observations are left (the reference's node
// set, PhasingGraph.cpp:793-846); its list of (read, slot) entries has room for every rank handed out, the dropped ones stay behind as holes.
// Two small launches instead of a device-wide scan library call each for the node numbers and the list offsets: workgroup sums, then every
// workgroup adds up the sums before it (a few hundred numbers) and scans its own 1 024 variants.
#define VSCAN_B 1024
struct VarSum { uint32_t nodes, cap, valid; };
__device__ __forceinline__ VarSum var_sum_of(int v, int n_var, const uint32_t *var_cnt, const uint32_t *var_del, const uint32_t *var_del2) {
    VarSum x{0u, 0u, 0u};
    if (v < n_var) { x.cap = var_cnt[v]; const uint32_t d = var_del[v] + var_del2[v]; x.valid = x.cap > d ? x.cap - d : 0u; x.nodes = x.valid ? 1u : 0u; }
    return x;
}
__global__ __launch_bounds__(VSCAN_B) void k_var_sums(int n_var, const uint32_t *var_cnt, const uint32_t *var_del, const uint32_t *var_del2, uint32_t *bsum) {
    __shared__ uint32_t s_a[16], s_b[16], s_c[16];
    const int v = blockIdx.x * VSCAN_B + threadIdx.x, w = threadIdx.x >> 6;
    const VarSum x = var_sum_of(v, n_var, var_cnt, var_del, var_del2);
    const uint32_t a = wave_sum(x.nodes), b = wave_sum(x.cap), c = wave_sum(x.valid);
    if (lane_id() == 0) { s_a[w] = a; s_b[w] = b; s_c[w] = c; }
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t A = 0, B = 0,
            Cc = 0; for (int q = 0; q < VSCAN_B / 64; ++q) { A += s_a[q]; B += s_b[q]; Cc += s_c[q]; } bsum[3 * blockIdx.x] = A; bsum[3 * blockIdx.x + 1] = B; bsum[3 * blockIdx.x + 2] = Cc; }
}
__global__ __launch_bounds__(VSCAN_B) void k_var_scan(int n_var, const uint32_t *var_cnt, const uint32_t *var_del, const uint32_t *var_del2, const uint32_t *bsum,
                                                      uint32_t *node_of, uint32_t *var_off, int32_t *nodes, uint32_t *node_off, uint32_t *node_cap, uint32_t *node_end, LpsCounters *cnt) {
    __shared__ uint32_t s_a[16], s_b[16], s_c[16], s_base[3];
    const int l = lane_id(), w = threadIdx.x >> 6;
    {   // sums of the workgroups before this one
        uint32_t a = 0, b = 0, c = 0;
        for (int q = threadIdx.x; q < (int)blockIdx.x; q += VSCAN_B) { a += bsum[3 * q]; b += bsum[3 * q + 1]; c += bsum[3 * q + 2]; }
        a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
        if (l == 0) { s_a[w] = a; s_b[w] = b; s_c[w] = c; }
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t A = 0, B = 0, Cc = 0; for (int q = 0; q < VSCAN_B / 64; ++q) { A += s_a[q]; B += s_b[q]; Cc += s_c[q]; } s_base[0] = A; s_base[1] = B; s_base[2] = Cc; }
        __syncthreads();
    }
    const int v = blockIdx.x * VSCAN_B + threadIdx.x;
    const VarSum x = var_sum_of(v, n_var, var_cnt, var_del, var_del2);
    const uint32_t ia = (uint32_t)wave_incl_scan_dpp((int)x.nodes), ib = (uint32_t)wave_incl_scan_dpp((int)x.cap), ic = (uint32_t)wave_incl_scan_dpp((int)x.valid);
    __syncthreads();
    if (l == 63) { s_a[w] = ia; s_b[w] = ib; s_c[w] = ic; }
    __syncthreads();
    uint32_t pa = s_base[0], pb = s_base[1], pc = s_base[2];
    for (int q = 0; q < w; ++q) { pa += s_a[q]; pb += s_b[q]; pc += s_c[q]; }
    const uint32_t nd = pa + ia - x.nodes, off = pb + ib - x.cap;
    if (v < n_var) {
        node_of[v] = nd; var_off[v] = off;
        if (x.nodes) { nodes[nd] = v; node_off[nd] = off; node_cap[nd] = x.cap; node_end[nd] = x.valid; }
    }
    if (v == n_var - 1) { cnt->n_nodes = nd + x.nodes; cnt->n_obs_final = (unsigned long long)(pc + ic); }
}

// ---- wave per extraction job (four alignments): the graph view of their observations and - for the alignments that are their read's only one,
// nearly all - the entries of the node-major lists, in ONE pass (the counting pass, the device-wide scans and the scatter pass of the first two
// rounds are gone: ranks come from the extraction, offsets from k_var_scan).
//   * a row's valid observations (alignment not deleted, entry not erased by the CNV filter) are compacted to the front of the row's slots:
//     g_pack[slot] = node << 2 | flags (bit 0 allele, bit 1 quality class): what k_edges reads per pair and k_read_correction per observation;
//   * entry of the node's list at var_off[variant] + rank: key = (read name, index in row) - the order the reference adds a node's reads in
//     (PhasingGraph.cpp:848-888: name order) - and the slot; a dropped observation leaves the all-ones key in its place (a hole k_edges sorts last);
//   * the type of the LAST alignment (BAM order) that saw an indel / SV / MOD row wins (:803-832: (*variantType)[pos] = ... is last-writer-wins);
//   * rows of reads with several alignments only get their packed words and ranks here: their merged row is built by k_merge_multi, which also
//     places its entries.
// The four rows need not be neighbours in the arena.  XCD-aware unit mapping: the entries of neighbouring reads are neighbours in the node lists.
template <bool KEY64>
__global__ __launch_bounds__(256) void k_graph_rows(int n_reads, const RowDesc *rows, const uint8_t *deleted, const ObsRec *obs, const uint32_t *name,
                                                    const uint32_t *name_head, const uint32_t *name_link, const uint32_t *node_of, const uint32_t *var_off,
                                                    int base_quality, int a_bits, uint32_t *g_pack, uint32_t *g_rank, int32_t *g_cnt,
                                                    uint32_t *mrow_off, int32_t *mrow_cnt, void *ukeys_v, uint32_t *uvals, uint32_t *vtype_key, int nb_reads
                                                    ) {
    const int l = lane_id();
    const int r0 = (xcd_unit((int)blockIdx.x, nb_reads) * 4 + (threadIdx.x >> 6)) * 4;
    if (r0 >= n_reads) return;
    unsigned long long *ukeys64 = (unsigned long long *)ukeys_v;
    int h_n = 0; uint32_t h_off = 0, h_name = 0; bool h_dead = true, h_single = false;
    if (l < 4 && r0 + l < n_reads) {
        const int r = r0 + l; const RowDesc d = rows[r]; h_n = max(d.cnt, 0); h_off = d.off;
        if (h_n > 0) { h_dead = deleted[r] != 0; h_name = name[r]; h_single = name_head[h_name] == (uint32_t)r + 1u && name_link[r] == 0u; }
    }
    int n[4]; uint32_t off[4], nm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { n[j] = __builtin_amdgcn_readlane(h_n, j); off[j] = (uint32_t)__builtin_amdgcn_readlane((int)h_off, j); nm[j] = (uint32_t)__builtin_amdgcn_readlane((int)h_name, j); }
    const unsigned dead = (unsigned)__ballot(h_dead) & 15u, single = (unsigned)__ballot(h_single) & 15u;
    const int c1 = n[0], c2 = c1 + n[1], c3 = c2 + n[2], total = c3 + n[3];
    int w0 = 0, w1 = 0, w2 = 0, w3 = 0;                                 // valid entries placed so far in each row (wave-uniform)
    const unsigned long long lt = lanemask_lt();
    for (int s0 = 0; s0 < total; s0 += 64) {
        const int s = s0 + l;
        const bool in = s < total;
        const int j = (s >= c1) + (s >= c2) + (s >= c3);
        const int cj = j == 0 ? 0 : (j == 1 ? c1 : (j == 2 ? c2 : c3));
        const uint32_t oj = j == 0 ? off[0] : (j == 1 ? off[1] : (j == 2 ? off[2] : off[3]));
        ObsRec o{-1, 0};
        if (in) o = obs[oj + (uint32_t)(s - cj)];
        const bool row_dead = (dead >> j) & 1u;
        const bool valid = in && !row_dead && o.var >= 0;
        const int v = o.var >= 0 ? o.var : -1 - o.var;                   // (an entry the CNV filter erased keeps its variant as -1 - index)
        const uint32_t rank = o.aq >> 10;
        // place among the valid entries of its row: the row's earlier rounds (w) + this round's valid lanes of the row below this one
        auto lanes = [&](int lo, int hi) __attribute__((always_inline)) -> unsigned long long {
            const int a = min(max(lo - s0, 0), 64), b = min(max(hi - s0, 0), 64);
            return ((b >= 64) ? ~0ull : ((1ull << b) - 1ull)) & ~((a >= 64) ? ~0ull : ((1ull << a) - 1ull));
        };
        const unsigned long long m = __ballot(valid);
        const unsigned long long r0m = lanes(0, c1), r1m = lanes(c1, c2), r2m = lanes(c2, c3), r3m = lanes(c3, total);
        const unsigned long long mine = j == 0 ? r0m : (j == 1 ? r1m : (j == 2 ? r2m : r3m));
        const int wj = j == 0 ? w0 : (j == 1 ? w1 : (j == 2 ? w2 : w3));
        const int a = wj + __popcll(m & mine & lt);
        w0 += __popcll(m & r0m); w1 += __popcll(m & r1m); w2 += __popcll(m & r2m); w3 += __popcll(m & r3m);
        if (in) {
            const uint32_t e = var_off[v] + rank;
            if (valid) {
                const uint16_t aq = (uint16_t)o.aq;
                const int q0 = aq_quality(aq);
                const int q = q0 < 0 ? ((q0 == -1 && !aq_allele(aq)) ? 30 : 60) : q0;   // sentinels -> quality 60; a SV row the read does not carry: 30 (:803-828)
                const unsigned fl = (unsigned)aq_allele(aq) | ((q >= base_quality) ? 2u : 0u);
                const uint32_t slot = oj + (uint32_t)a;
                g_pack[slot] = (node_of[v] << 2) | fl;
                if (q0 < 0) {                                             // :803-832 (-2 / -3: MOD on the forward / reverse strand)
                    const unsigned ty = q0 == -4 ? 3u : (q0 == -5 ? 4u : (q0 == -1 ? 1u : 2u));
                    atomicMax(&vtype_key[v], ((unsigned)(r0 + j) << 3) | ty);
                }
                if ((single >> j) & 1u) {
                    const uint32_t nmj = j == 0 ? nm[0] : (j == 1 ? nm[1] : (j == 2 ? nm[2] : nm[3]));
                    // (32-bit keys: key and slot are ONE 8-byte entry - one scattered store per observation instead of two)
                    if (KEY64) { ukeys64[e] = ((unsigned long long)nmj << a_bits) | (unsigned)a; uvals[e] = slot; }
                    else reinterpret_cast<uint2 *>(ukeys_v)[e] = make_uint2((nmj << a_bits) | (unsigned)a, slot);
                } else g_rank[slot] = rank;
            } else if (n[0] + n[1] + n[2] + n[3] > 0) {                   // a dropped observation: its place in the list stays a hole
                if (KEY64) ukeys64[e] = ~0ull; else reinterpret_cast<uint2 *>(ukeys_v)[e] = make_uint2(0xffffffffu, 0u);
            }
        }
    }
    if (l < 4 && r0 + l < n_reads) {
        const int wl = l == 0 ? w0 : (l == 1 ? w1 : (l == 2 ? w2 : w3));
        const int kept = h_dead ? 0 : wl;
        g_cnt[r0 + l] = kept;
        if (h_single) { mrow_off[h_name] = h_off; mrow_cnt[h_name] = kept; }
    }
}

// SV / MOD co-phasing re-indexes every observation after the extraction (k_extra_merge): the ranks are taken here instead, one pass over the rows.
// Alignments the overlap filter deleted count like the others and are taken off again in var_del, as k_groups does on the plain path.
__global__ __launch_bounds__(256) void k_count_ranks(int n_reads, const RowDesc *rows, const uint8_t *deleted, ObsRec *obs, uint32_t *var_cnt, uint32_t *var_del, LpsCounters *cnt) {
    const int l = lane_id();
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_reads) return;
    const RowDesc d = rows[r];
    const bool del = deleted[r] != 0;
    for (int k = l; k < d.cnt; k += 64) {
        ObsRec o = obs[d.off + k];
        const unsigned rk = atomicAdd(&var_cnt[o.var], 1u);
        if (rk > 0x3fffffu) atomicOr(&cnt->err, (unsigned)LPS_ERR_KEY_RANGE);
        o.aq = (o.aq & 0x3ffu) | (rk << 10);
        obs[d.off + k] = o;
        if (del) atomicAdd(&var_del[o.var], 1u);
    }
}

// ================================================================================================ merged rows
// Reads with ONE surviving alignment (the vast majority) use that alignment's row as their merged row.  Reads with several
// (supplementary alignments kept by the overlap filter) get a freshly reserved tail row that holds the concatenation of their
// rows in BAM order sorted by position (== node index): k_merge_plan (thread per name group) reserves it and queues the group,
// k_merge_multi (wave per queued group) places every element by rank: own index + elements of the other rows that sort before it
// (ties: earlier alignment first).  That is the stable order == libstdc++ std::sort for n <= 16 and differs from it only in the
// relative order of equal positions beyond that.
__global__ __launch_bounds__(256) void k_merge_plan(LpsCounters *cnt, const uint32_t *mg_start, const uint32_t *mg_cnt, const uint32_t *mg_name, const uint32_t *mm_r,
                             const RowDesc *rows, const int32_t *g_cnt, unsigned long long tail_lo, unsigned long long tail_size,
                             uint32_t *mrow_off, int32_t *mrow_cnt, uint32_t *mg_plan) {
    __shared__ unsigned s_tot[4], s_n[4]; __shared__ unsigned long long s_base_t;
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = threadIdx.x >> 6;
    bool multi = false; int total = 0; uint32_t id = 0;
    if (q < cnt->n_multi) {
        const uint32_t s0 = mg_start[q], k = mg_cnt[q]; id = mg_name[q];
        int alive = 0; uint32_t one = 0;
        for (uint32_t t = 0; t < k; ++t) { const uint32_t r = mm_r[s0 + t]; if (g_cnt[r] > 0) { ++alive; total += g_cnt[r]; one = r; } }
        if (alive == 0) { mrow_off[id] = 0; mrow_cnt[id] = 0; mg_plan[q] = 0u; }
        else if (alive == 1) { mrow_off[id] = rows[one].off; mrow_cnt[id] = total; mg_plan[q] = 1u | (one << 2); }   // the one row IS the merged row: only its entries are placed
        else multi = true;
    }
    // tail slots and the count of merged rows: ONE atomic each per workgroup (same-word atomics are served one after the other)
    const int mine = multi ? total : 0;
    const int incl = wave_incl_scan_dpp(mine);
    const unsigned long long mm = __ballot(multi);
    if (lane_id() == 63) s_tot[w] = (unsigned)incl;
    if (lane_id() == 0) s_n[w] = (unsigned)__popcll(mm);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long T = (unsigned long long)s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3]; const unsigned N = s_n[0] + s_n[1] + s_n[2] + s_n[3];
        s_base_t = T ? atomicAdd(&cnt->tail_total, T) : 0ull; if (N) atomicAdd(&cnt->n_merged, N);
    }
    __syncthreads();
    if (multi) {
        id = mg_name[q];                                                  // (read again: the value carried across the barriers was lost by the compiler - the stores below went to another name's slot)
        unsigned long long toff = s_base_t + (unsigned long long)(incl - mine);
        for (int qq = 0; qq < w; ++qq) toff += s_tot[qq];
        if (toff + (unsigned long long)total > tail_size) { atomicOr(&cnt->err, (unsigned)LPS_ERR_OBS_OVERFLOW); mrow_off[id] = 0; mrow_cnt[id] = 0; mg_plan[q] = 0u; }   // the host grows the buffers and reruns
        else { mrow_off[id] = (uint32_t)(tail_lo + toff); mrow_cnt[id] = total; mg_plan[q] = 2u; }
    }
}

// ---- std::sort of one row by one wavefront (rows that hold a position twice; see k_merge_multi).  The row sits in LDS in its unsorted order.
// __introsort_loop is walked as libstdc++ does (explicit stack for the right-hand parts), but each __unguarded_partition is done by all 64 lanes:
// with pivot v, A = positions holding >= v in ascending order and B = positions holding <= v in descending order, the serial loop swaps A_j with
// B_j for j = 1..m where m = #{j : A_j < B_j} (the pointers never look at a swapped position again, so both lists can be taken from the array
// as it was before the first swap), and returns cut = min(A_(m+1), B_m).  __final_insertion_sort is a stable sort, i.e. a rank count.
// Heapsort (depth limit hit) stays serial on lane 0.  k/p: n <= STDSORT_LDS elements in LDS; pa/pb: STDSORT_LDS uint16 each; stk: 192 ints.
constexpr int STDSORT_LDS = 1024;
__device__ inline void wave_std_sort(int32_t *k, uint8_t *p, int n, int *stk, uint16_t *pa, uint16_t *pb, int32_t *out_k, uint8_t *out_p, int l) {
    const unsigned long long lt = (1ull << l) - 1ull;
    if (n > 16) {
        int lg = 0; for (int m = n; m > 1; m >>= 1) ++lg;
        int *sf = stk, *sl = stk + 64, *sd = stk + 128; int sp = 1;
        if (l == 0) { sf[0] = 0; sl[0] = n; sd[0] = 2 * lg; }
        __threadfence_block();
        while (sp) {
            --sp; const int first = sf[sp]; int last = sl[sp], depth = sd[sp];
            while (last - first > 16) {
                if (depth == 0) { if (l == 0) { StdSortArrays a{k, p}; stdsort_heapsort(a, first, last); } __threadfence_block(); break; }
                --depth;
                if (l == 0) { StdSortArrays a{k, p}; stdsort_move_median_to_first(a, first, first + 1, first + (last - first) / 2, last - 1); }
                __threadfence_block();
                const int32_t pv = k[first];
                int nA = 0, nB = 0;
                for (int b0 = first + 1; b0 < last; b0 += 64) {
                    const int i = b0 + l; const bool f = i < last && k[i] >= pv; const unsigned long long m = __ballot(f);
                    if (f) pa[nA + __popcll(m & lt)] = (uint16_t)i;
                    nA += __popcll(m);
                }
                for (int b0 = last - 1; b0 > first; b0 -= 64) {
                    const int i = b0 - l; const bool f = i > first && k[i] <= pv; const unsigned long long m = __ballot(f);
                    if (f) pb[nB + __popcll(m & lt)] = (uint16_t)i;
                    nB += __popcll(m);
                }
                __threadfence_block();
                const int mn = nA < nB ? nA : nB; int m = 0;
                for (int j0 = 0; j0 < mn; j0 += 64) {
                    const int j = j0 + l; const int c = __popcll(__ballot(j < mn && pa[j] < pb[j])); m += c;
                    if (c < 64) break;
                }
                for (int j = l; j < m; j += 64) {
                    const int a = pa[j], b = pb[j]; const int32_t ka = k[a], kb = k[b]; const uint8_t qa = p[a], qb = p[b];
                    k[a] = kb; p[a] = qb; k[b] = ka; p[b] = qa;
                }
                int cut = 0x7fffffff; if (m < nA) cut = pa[m]; if (m > 0 && (int)pb[m - 1] < cut) cut = pb[m - 1];
                __threadfence_block();
                if (l == 0 && sp < 64) { sf[sp] = cut; sl[sp] = last; sd[sp] = depth; }
                if (sp < 64) ++sp;
                __threadfence_block();
                last = cut;
            }
        }
    }
    // what the loop leaves: runs of <= 16 elements, each run's keys between those of its neighbours - an element's final place is decided inside
    // the 16 positions either side of it
    for (int i = l; i < n; i += 64) {
        const int32_t ki = k[i]; const int lo = i > 16 ? i - 16 : 0, hi = i + 17 < n ? i + 17 : n; int rank = lo;
        for (int j = lo; j < hi; ++j) { const int32_t kj = k[j]; rank += (kj < ki) || (kj == ki && j < i); }
        out_k[rank] = ki; out_p[rank] = p[i];
    }
}

// test hook (lps_debug_std_sort_gpu): rows [row_start[r], row_start[r+1]) of (keys, payload) sorted in place, a wave per row, same path as k_merge_multi
__global__ __launch_bounds__(256) void k_debug_std_sort(int32_t *keys, uint8_t *payload, const long long *row_start, int n_rows) {
    __shared__ int s_stk[4][192]; __shared__ int32_t s_k[4][STDSORT_LDS]; __shared__ uint8_t s_p[4][STDSORT_LDS]; __shared__ uint16_t s_a[4][STDSORT_LDS], s_b[4][STDSORT_LDS];
    const int l = lane_id(), w = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + w; r < n_rows; r += gridDim.x * 4) {
        const long long o = row_start[r]; const int n = (int)(row_start[r + 1] - o);
        if (n <= STDSORT_LDS) {
            for (int i = l; i < n; i += 64) { s_k[w][i] = keys[o + i]; s_p[w][i] = payload[o + i]; }
            __threadfence_block();
            wave_std_sort(s_k[w], s_p[w], n, s_stk[w], s_a[w], s_b[w], keys + o, payload + o, l);
            __threadfence_block();
        } else if (l == 0) { StdSortArrays a{keys + o, payload + o}; stdsort_run(a, n, s_stk[w]); }
    }
}
void launch_debug_std_sort(int32_t *keys, uint8_t *payload, const long long *row_start, int n_rows, hipStream_t s) {
    hipLaunchKernelGGL(k_debug_std_sort, dim3(256), dim3(256), 0, s, keys, payload, row_start, n_rows);
}

template <bool KEY64>
__global__ __launch_bounds__(256) void k_merge_multi(const LpsCounters *cnt, const uint32_t *mg_start, const uint32_t *mg_cnt, const uint32_t *mg_name, const uint32_t *mg_plan, const uint32_t *mm_r,
                                                     const RowDesc *rows, const int32_t *g_cnt, uint32_t *g_pack, const uint32_t *g_rank, int32_t *t_node,
                                                             uint8_t *t_flag, uint32_t *t_src, uint32_t tail_lo,
                                                     const uint32_t *mrow_off, const int32_t *nodes, const uint32_t *var_off, int a_bits, void *ukeys_v, uint32_t *uvals
                                                     ) {
    __shared__ int s_stk[4][192]; __shared__ int32_t s_k[4][STDSORT_LDS]; __shared__ uint8_t s_p[4][STDSORT_LDS]; __shared__ uint16_t s_a[4][STDSORT_LDS], s_b[4][STDSORT_LDS];
    const int l = lane_id();
    const unsigned n_waves = gridDim.x * 4;
    unsigned long long *ukeys64 = (unsigned long long *)ukeys_v;
    for (unsigned q = blockIdx.x * 4 + (threadIdx.x >> 6); q < cnt->n_multi; q += n_waves) {
        const uint32_t plan = mg_plan[q];
        if ((plan & 3u) == 0u) continue;                                  // no observation left, or its tail reservation failed (the host grows the buffers and reruns)
        const uint32_t id = mg_name[q];
        // entry of the node's list for element a of the read's merged row: (read name, a) at the rank of the observation it came from
        auto place = [&](int a, uint32_t slot, int nd, uint32_t src) __attribute__((always_inline)) {
            const uint32_t e = var_off[nodes[nd]] + g_rank[src];
            if (KEY64) { ukeys64[e] = ((unsigned long long)id << a_bits) | (unsigned)a; uvals[e] = slot; }
            else reinterpret_cast<uint2 *>(ukeys_v)[e] = make_uint2((id << a_bits) | (unsigned)a, slot);
        };
        if ((plan & 3u) == 1u) {                                          // one alignment left: its row is the merged row
            const uint32_t r = plan >> 2, off = rows[r].off; const int n = g_cnt[r];
            for (int a = l; a < n; a += 64) place(a, off + (uint32_t)a, (int)(g_pack[off + a] >> 2), off + (uint32_t)a);
            continue;
        }
        const uint32_t s0 = mg_start[q], s1 = s0 + mg_cnt[q];
        const uint32_t base = mrow_off[id];
        int32_t *m_node = t_node + (base - tail_lo); uint8_t *m_flag = t_flag + (base - tail_lo); uint32_t *m_src = t_src + (base - tail_lo);
        const int w = threadIdx.x >> 6;
        int total = 0; for (uint32_t sa = s0; sa < s1; ++sa) total += g_cnt[mm_r[sa]];
        const bool in_lds = total <= STDSORT_LDS;
        // the alignments' rows concatenated in BAM order, in LDS when they fit (they do for real read lengths): the rank searches below are chains
        // of dependent probes - tens of cycles each in LDS, a microsecond each in HBM - and the std::sort path wants this copy anyway
        if (in_lds) {
            int at = 0;
            for (uint32_t sa = s0; sa < s1; ++sa) { const uint32_t ra = mm_r[sa]; const int na = g_cnt[ra]; const uint32_t oa = rows[ra].off;
                for (int k = l; k < na; k += 64) { const uint32_t wd = g_pack[oa + k]; s_k[w][at + k] = (int32_t)(wd >> 2); s_p[w][at + k] = (uint8_t)(wd & 3u); } at += na; }
            wave_sync();
        }
        bool dup = false;
        int aa = 0;                                                      // offset of alignment sa's row inside the concatenation
        for (uint32_t sa = s0; sa < s1; ++sa) {                      // source alignment (BAM order inside the group)
            const uint32_t ra = mm_r[sa]; const int na = g_cnt[ra]; const uint32_t oa = rows[ra].off;
            for (int k = l; k < na; k += 64) {
                const uint32_t wd = in_lds ? 0u : g_pack[oa + k];
                const int nd = in_lds ? s_k[w][aa + k] : (int)(wd >> 2); const uint8_t fl = in_lds ? s_p[w][aa + k] : (uint8_t)(wd & 3u);
                int rank = k, ab = 0;
                for (uint32_t sb = s0; sb < s1; ++sb) {
                    const uint32_t rb = mm_r[sb]; const int nb = g_cnt[rb];
                    if (sb != sa) {
                        const uint32_t ob = rows[rb].off;
                        auto at_b = [&](int m) __attribute__((always_inline)) -> int { return in_lds ? s_k[w][ab + m] : (int)(g_pack[ob + m] >> 2); };
                        // earlier alignment: its equal positions go first (count <= nd); later alignment: only smaller ones
                        int lo = 0, hi = nb;
                        if (sb < sa) { while (lo < hi) { const int m = (lo + hi) >> 1; if (at_b(m) <= nd) lo = m + 1; else hi = m; } }
                        else { while (lo < hi) { const int m = (lo + hi) >> 1; if (at_b(m) < nd) lo = m + 1; else hi = m; }
                               dup |= lo < nb && at_b(lo) == nd; }                                   // the same position in a later alignment
                        rank += lo;
                    }
                    ab += nb;
                }
                m_node[rank] = nd; m_flag[rank] = fl;
                // the observation this element came from: its rank in the node's list places the element.  The std::sort path below only
                // permutes elements of EQUAL node among themselves, so the pairing stays a bijection per node
                m_src[rank] = oa + k;
            }
            aa += na;
        }
        // The placement above is the STABLE order.  libstdc++'s std::sort leaves equal positions in another order once a row has more than 16
        // elements, and the reference's float sums see that order: when the row holds a position twice, rebuild it as the reference does -
        // alignments concatenated in BAM order, then std::sort restated step by step (lps_stdsort.h).  Rare (overlapping supplementary alignments).
#ifndef LPS_NO_STDSORT_FIX
        if (total > 16) {
            if (__ballot(dup)) {
                // sorted by the whole wave (wave_std_sort) from the LDS copy for rows of up to STDSORT_LDS elements, else in place in HBM by lane 0
                // (a chain of dependent accesses at ~1 us each; such rows do not occur with real read lengths)
                if (in_lds) { wave_sync(); wave_std_sort(s_k[w], s_p[w], total, s_stk[w], s_a[w], s_b[w], m_node, m_flag, l); }
                else {
                    int at = 0;
                    for (uint32_t sa = s0; sa < s1; ++sa) { const uint32_t ra = mm_r[sa]; const int na = g_cnt[ra]; const uint32_t oa = rows[ra].off;
                        for (int k = l; k < na; k += 64) { const uint32_t wd = g_pack[oa + k]; m_node[at + k] = (int32_t)(wd >> 2); m_flag[at + k] = (uint8_t)(wd & 3u); } at += na; }
                    __threadfence_block();
                    if (l == 0) { StdSortArrays a{m_node, m_flag}; stdsort_run(a, total, s_stk[w]); }
                }
                __threadfence_block();
            }
        }
#endif
        __threadfence_block(); wave_sync();
        for (int k = l; k < total; k += 64) {                             // the merged row as k_edges reads it, and its entries in the node lists
            const int nd = m_node[k];
            g_pack[base + k] = ((uint32_t)nd << 2) | (uint32_t)m_flag[k];
            place(k, base + (uint32_t)k, nd, m_src[k]);
        }
        wave_sync();                                                     // the LDS copy is reused by the wave's next group
    }
}

// ================================================================================================ edges
__device__ __forceinline__ float edge_upd(float x, bool hi, double w) {
    return hi ? x + 1.0f : (float)((double)x + w);           // SubEdge::addSubEdge (:40-43,62-65)
}

// One read's contribution to the two cells its source allele selects (x0: target REF, x1: target ALT); bit 0 of the packed word is the target's
// allele, bit 1 its quality class.  hi_mask (wave-uniform): 2 when the source observation is of high quality, else 0 (the pair never is).
__device__ __forceinline__ void cell_upd(float &x0, float &x1, uint32_t word, uint32_t hi_mask, double w) {
    const bool alt = word & 1u;
    const float x = alt ? x1 : x0;
    const float nx = edge_upd(x, (word & hi_mask) != 0u, w);
    x0 = alt ? x0 : nx; x1 = alt ? nx : x1;
}

// wave per source node i.  Lane k (< A) owns the four cells (rr,ra,ar,aa) towards node i+1+k in registers.
// For each read observing node i (in name-rank order) lane t loads the read's t-th following observation;
// its node distance d selects the owning lane, the (allele pair, quality class) travels there by ds_permute.
// SOFF: every slot of g_pack is below 2^30 words, so a window's byte offset fits the scalar offset of the buffer load
template <bool KEY64, bool SOFF>
__global__ __launch_bounds__(256) void k_edges(const LpsCounters *cnt, const uint32_t *node_off, const uint32_t *node_cap, const uint32_t *node_end,
                                               const void *ukeys_v, const uint32_t *uvals,
                                               void *skeys_v, uint32_t *svals,
                                               const uint32_t *mrow_off, const int32_t *mrow_cnt, int m_bits, int a_bits,
                                               const uint32_t *g_pack, uint32_t tail_lo, int A, double edge_weight,
                                               double edge_threshold, const int32_t *nodes, const uint32_t *vtype_key, float *edge, uint8_t *erec, uint32_t *node_pairs) {
    typedef typename std::conditional<KEY64, unsigned long long, uint32_t>::type key_t;
    const key_t *ukeys = (const key_t *)ukeys_v; key_t *skeys = (key_t *)skeys_v;
    const uint2 *uent = (const uint2 *)ukeys_v;                          // !KEY64: the unsorted lists hold {key, slot} entries (k_graph_rows)
    auto ukey = [&](uint32_t e) __attribute__((always_inline)) -> key_t { if constexpr (KEY64) return ukeys[e]; else return (key_t)uent[e].x; };
    const key_t HOLE = (key_t)~(key_t)0;                                // a dropped observation's place in the list (k_graph_rows)
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2, and a row of packed words is wanted by the ~35 source nodes before it:
    // workgroup b takes node block (b % 8) * (blocks / 8) + b / 8, so that an XCD walks ONE contiguous eighth of the nodes and a row is fetched into
    // one L2, not into all eight (the grid is a multiple of 8)
    const int i = xcd_unit((int)blockIdx.x, (int)gridDim.x) * 4 + (threadIdx.x >> 6), l = lane_id();
    const int n_nodes = (int)cnt->n_nodes;
    if (i >= n_nodes) return;
    // the node's list: room for every observation counted at the extraction (cap), of which n_valid are left after the filters; the others are holes
    // (wave-uniform by construction - one node per wave -, but only readfirstlane tells the compiler: the loops over the list then run on the scalar unit
    // instead of under an exec mask, with the cells copied around every pass)
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)node_off[i]); const int cap = __builtin_amdgcn_readfirstlane((int)node_cap[i]),
            n_valid = __builtin_amdgcn_readfirstlane((int)node_end[i]);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    unsigned long long pairs = 0;
    const uint32_t m_mask = (uint32_t)((1ull << m_bits) - 1ull);
    const uint32_t first4 = (uint32_t)__builtin_amdgcn_readfirstlane((i + 1) << 2);   // packed word of node i+1 with flag 0
    const bool short_list = cap <= 64;                                  // unsorted entries (ukeys/uvals): up to 64 are ordered in registers below
    if (!short_list) {                                                  // coverage above 64: rank sort through memory (rank = number of smaller keys; keys are unique, holes sort last)
        for (int a = l; a < cap; a += 64) {
            const key_t k = ukey(off + a);
            int rank = 0;
            for (int t = 0; t < cap; ++t) rank += ukey(off + t) < k;   // wave-uniform address: one broadcast load per step
            if (k != HOLE) { skeys[off + rank] = k; svals[off + rank] = KEY64 ? uvals[off + a] : uent[off + a].y; }
        }
        __threadfence_block(); wave_sync();
    }
    const uint32_t end = off + (uint32_t)(short_list ? cap : n_valid);
    for (uint32_t e0 = off; e0 < end; e0 += 64) {
        int nb = __builtin_amdgcn_readfirstlane((int)min(64u, end - e0));      // wave-uniform: the loops over it run on the scalar unit
        uint32_t my_val = 0, my_end = 0;
        if (short_list) {
            key_t key = HOLE; uint32_t v0 = 0, x0 = 0, vslot = 0;
            if (l < nb) { if constexpr (KEY64) { key = ukeys[e0 + l]; } else { const uint2 en = uent[e0 + l]; key = (key_t)en.x; vslot = en.y; } }
            if (key != HOLE) {
                const uint32_t m = (uint32_t)(key >> a_bits) & m_mask;
                v0 = KEY64 ? uvals[e0 + l] : vslot; x0 = mrow_off[m] + (uint32_t)mrow_cnt[m];
            }
            // rank = number of smaller keys (keys are unique): the read order of the reference (name rank, index in the merged read); holes rank last
            int rank = 0;
            if (!KEY64) {                                               // the whole key in one word (the usual case): one lane read per step
                const int klo = (int)(uint32_t)key;
                for (int t = 0; t < nb; ++t) rank += (unsigned)__builtin_amdgcn_readlane(klo, t) < (unsigned)klo;
            } else {
                const int klo = (int)(unsigned)(unsigned long long)key, khi = (int)(unsigned)((unsigned long long)key >> 32);
                for (int t = 0; t < nb; ++t) {
                    const unsigned long long o = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(khi, t) << 32) | (unsigned)__builtin_amdgcn_readlane(klo, t);
                    rank += o < (unsigned long long)key;
                }
            }
            const int nbv = __popcll(__ballot(key != HOLE));
            const int dst = ((l < nb && key != HOLE) ? rank : max(l, nbv)) << 2;   // holes and lanes past the list stay out of the ranks 0..nbv-1 of the valid entries
            my_val = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)v0);
            my_end = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)x0);
            nb = __builtin_amdgcn_readfirstlane(nbv);          // (a popcount of a ballot: uniform, but only this tells the compiler - the loop below then runs on the scalar unit)
        } else if (l < nb) {
            const key_t key = skeys[e0 + l];
            const uint32_t m = (uint32_t)(key >> a_bits) & m_mask;
            my_val = svals[e0 + l]; my_end = mrow_off[m] + (uint32_t)mrow_cnt[m];
        }
        // every lane's own observation (flag of the source side) and its share of the pair count, loaded side by side
        int my_sf = l < nb ? (int)(g_pack[my_val] & 3u) : 0;
        const int my_lim = l < nb ? (int)min((uint32_t)A, my_end - my_val - 1u) : 0;    // observations of the read inside the window (A <= 63)
        pairs += (unsigned long long)my_lim;
        // one lane read per step hands out the source flag (bits 0-1: bit 1 = the quality mask as the receiver needs it), the window length in BYTES
        // (bits 2-7: the descriptor's size word, no shift) and whether the read's row is a merged row of several alignments (bit 10; tail arena) -
        // the only rows that can hold a node twice inside the window
        int my_meta = my_sf | (my_lim << 2) | ((l < nb && my_val >= tail_lo) ? 1 << 10 : 0);
        // The reads of the block are taken in TWO runs: first those whose source observation shows REF, then those that show ALT, each run in rank
        // order.  A cell only ever sees the reads of one source allele (rr, ra: REF; ar, aa: ALT), so every cell still receives its updates in the
        // reference's order - and inside a run the pair of cells is FIXED: no branch on the source allele per read, and the compiler keeps the
        // cells in place (with the branch inside one loop it copied all four cells twice per read: 9 moves beside 19 useful instructions).
        const unsigned long long alt_mask = __ballot(l < nb && (my_sf & 1));
        const int n_ref = nb - __popcll(alt_mask);
        {
            const unsigned long long below = (1ull << l) - 1ull, valid = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
            const int pos = l >= nb ? l : ((my_sf & 1) ? n_ref + __popcll(alt_mask & below) : __popcll(~alt_mask & valid & below));
            my_val = (uint32_t)__builtin_amdgcn_ds_permute(pos << 2, (int)my_val);
            my_meta = __builtin_amdgcn_ds_permute(pos << 2, my_meta);
        }
        const unsigned long long slow_mask = __ballot((my_meta >> 10) & 1);
        // The t-th read's following observations are requested one read ahead, so that the loads of read t+1 are in flight while read t is applied
        // (two ahead were measured slower, 1.05 vs 1.02 ms: the rows come from L2 and the other waves of the SIMD cover the rest).  The request is a BUFFER
        // load whose descriptor is the read's window itself - base = the word after the source observation, size = the observations inside the
        // window, both wave-uniform and put together on the scalar unit: the lane adds its 4*l, and a lane beyond the window gets 0 from the bounds
        // check (no compare, no vector address arithmetic, no default).  A word of a following observation is never 0 (its node is >= 1).
        // The request is UNCONDITIONAL - past the list it reads a lane whose window is empty (lanes >= nb hold 0: no memory access, all lanes get 0):
        // a load under a branch makes the compiler wait for ALL outstanding loads where the paths join (s_waitcnt vmcnt(0)), i.e. for the request
        // just made, and the one-ahead scheme hides nothing
        auto request = [&](int t, uint32_t &w, int &meta) {
            const int tt = min(t, 63);
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)my_val, tt);
            meta = __builtin_amdgcn_readlane(my_meta, tt);
            if (SOFF) {     // the descriptor's base stays put and the window's place goes into the scalar offset; a raw buffer is out of range from
                            // num_records - scalar offset on, so the size word is the window's END: one addition instead of a 64-bit base
                const uint32_t so = v << 2;
                const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc((void *)(g_pack + 1), 0, (int)(so + ((uint32_t)meta & 0xfcu)), 0x00020000);
                w = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(win, 4 * l, (int)so, 0);
            } else {
                const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc((void *)(g_pack + (size_t)v + 1), 0, meta & 0xfc, 0x00020000);
                w = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(win, 4 * l, 0, 0);
            }
        };
        // No node twice in the row (all but the merged rows, ~2 % of the reads): the packed word itself travels to the lane that owns its target
        // and the receiver picks between the TWO cells of the run.  The permute address is 4*(d-1) + flag as unsigned: an empty slot (0), a node
        // before i+1 or beyond the window all land on lane 63 or on a lane >= A, whose cells are never stored (ds_permute takes
        // lane = address / 4 mod 64: the flag bits below do not matter)
        auto plain_read = [&](uint32_t w, int meta, float &x0, float &x1) {
            const uint32_t recv = (uint32_t)__builtin_amdgcn_ds_permute((int)min(w - first4, 255u), (int)w);
            // no branch around the update (some lane always receives): a lane that received nothing keeps its cells through the select masks
            const bool alt = recv & 1u, got = recv != 0u;
            const float nx = edge_upd(alt ? x1 : x0, (recv & (uint32_t)meta & 2u) != 0u, edge_weight);    // quality mask: wave-uniform
            x0 = (got && !alt) ? nx : x0; x1 = alt ? nx : x1;
        };
        // the same node twice inside the window (overlapping alignments of one read, neighbours in the position-sorted row): the second
        // occurrence is applied in a second round, after the first - window order, as the reference's pair loop goes
        auto merged_read = [&](uint32_t w, int meta) {
            const int sf = meta & 3;
            const int n2 = (int)(w >> 2), f2 = (int)(w & 3u);
            const int d = n2 - i;
            const bool ok = w != 0u && d >= 1 && d <= A;
            const int cell = ((sf & 1) << 1) | (f2 & 1);
            const int hi = (sf & f2 & 2) << 2;                          // both observations of high quality -> bit 3
            const int payload = 1 | (cell << 1) | hi;
            int round_of = 0, rounds = 1;
            for (int s = 1; s < 64; ++s) {
                const int dprev = __shfl_up(d, s); const bool okprev = __shfl_up((int)ok, s) != 0;
                const bool same = ok && l >= s && okprev && dprev == d && round_of == s - 1;
                if (!__ballot(same)) break;
                if (same) round_of = s;
                rounds = s + 1;
            }
            for (int rd = 0; rd < rounds; ++rd) {
                // lanes without a contribution in this round push to lane 63, which owns no target (A <= 63)
                const bool mine = ok && round_of == rd;
                const int recv = __builtin_amdgcn_ds_permute((mine ? (d - 1) : 63) << 2, mine ? payload : 0);
                if (recv & 1) {
                    const int c = (recv >> 1) & 3; const bool h = (recv >> 3) & 1;
                    const float x = c == 0 ? a0 : (c == 1 ? a1 : (c == 2 ? a2 : a3));
                    const float nx = edge_upd(x, h, edge_weight);
                    a0 = c == 0 ? nx : a0; a1 = c == 1 ? nx : a1; a2 = c == 2 ? nx : a2; a3 = c == 3 ? nx : a3;
                }
            }
        };
        uint32_t w0; int m0;
        request(0, w0, m0);
        int t = 0;
        // one run: [t, run_end) in stretches of plain reads (an inner loop that holds nothing but them) with the merged rows between the stretches
        auto run = [&](const int run_end, float &x0, float &x1) {
            while (t < run_end) {
                const unsigned long long rest = slow_mask >> t;
                const int stop = min(run_end, rest ? t + (int)__builtin_ctzll(rest) : 64);
                // two reads per trip, the request buffers taking turns: no copy of the word just requested into the register of the one just applied
                for (; t + 1 < stop; t += 2) {
                    uint32_t w1; int m1;
                    request(t + 1, w1, m1);
                    plain_read(w0, m0, x0, x1);
                    request(t + 2, w0, m0);
                    plain_read(w1, m1, x0, x1);
                }
                if (t < stop) {
                    uint32_t w1; int m1;
                    request(t + 1, w1, m1);
                    plain_read(w0, m0, x0, x1);
                    w0 = w1; m0 = m1; ++t;
                }
                if (t < run_end) {
                    uint32_t w1; int m1;
                    request(t + 1, w1, m1);
                    merged_read(w0, m0);
                    w0 = w1; m0 = m1; ++t;
                }
            }
        };
        run(n_ref, a0, a1);
        run(nb, a2, a3);
    }
    pairs = wave_sum(pairs);
    if (l == 0) node_pairs[i] = (uint32_t)pairs;   // summed later (no single-address atomics in the hot kernel)
    if (l < A) {
        reinterpret_cast<float4 *>(edge)[(size_t)i * A + l] = make_float4(a0, a1, a2, a3);
        // findBestEdgePair (:166-228) + the weight rules of edgeConnectResult (:216,:367) and Onelongcase (:261-265):
        // everything that does not depend on the scan state is folded into one vote BYTE per (i,k) (a 64-node tile of the scan is 2.2 KB of LDS):
        //   bits 3-4 = weight code (0 not connected / no such node, 1: 0.1, 2: 1, 3: 20), bit0 different haplotype,
        //   bit1 single-read vote (para+cross <= 1), bit2 counts towards the Onelongcase sums
        const float rr = a0, ra = a1, ar = a2, aa = a3;
        const float para = rr + aa, cross = ra + ar;
        const double esr = (double)fminf(para, cross) / (double)fmaxf(para, cross);
        int dir = 0;
        if (para > cross) dir = 1; else if (para < cross) dir = 2;
        // an edge between a SNP and a MOD row: threshold 0.3, and nothing connects when the four cells sum to less than one read (:197-202)
        const int typ = (int)(vtype_key[nodes[i]] & 7u), typ_t = (i + 1 + l < n_nodes) ? (int)(vtype_key[nodes[i + 1 + l]] & 7u) : 0;
        double thr = edge_threshold;
        if ((typ == 0 && typ_t == 2) || (typ == 2 && typ_t == 0)) thr = ((rr + ra + ar + aa) < 1) ? -1.0 : 0.3;
        if (esr > thr) dir = 0;
        const bool w20 = (esr <= 0.1 && (rr + aa + ra + ar) >= 1) || (para < 1 && cross >= 1) || (para >= 1 && cross < 1);
        const bool single = (para + cross) <= 1;
        const bool lowesr = esr < 0.2;
        float w = (typ == 4) ? 0.1f : (w20 ? 20.f : 1.f);
        const bool osum = !single && lowesr && w >= 1.f && typ != 3;
        if (dir == 0 || i + 1 + l >= n_nodes) w = 0.f;
        const unsigned fl = (dir == 2 ? 1u : 0u) | (single ? 2u : 0u) | (osum ? 4u : 0u);
        const unsigned wc = w == 0.f ? 0u : (typ == 4 ? 1u : (w20 ? 3u : 2u));
        erec[(size_t)i * A + l] = (uint8_t)(wc ? ((wc << 3) | fl) : 0u);
    }
}

// ================================================================================================ vote scan
// The reference's edgeConnectResult is a serial dependence chain: a node's haplotype depends on the votes of the
// <=A nodes before it.  A single wavefront is instruction-issue bound (~6 cycles per instruction), so walking a
// whole chromosome in one wave costs ~0.3 us per node.  Instead the chain is cut into segments of SCAN_SEG nodes
// that are walked CONCURRENTLY, one wavefront each, from a speculative start SCAN_WARM nodes before the segment:
//   * the speculative walk starts with the fresh state (no votes, no block) - after >=A consistently voted nodes
//     the vote accumulators no longer depend on anything before the start, only the haplotype LABELS may be
//     globally swapped; each wave therefore walks two variants (first block labelled 1 / labelled 2) interleaved
//     in one instruction stream (two independent chains also hide each other's issue stalls);
//   * k_scan_stitch then goes over the segments in order and accepts a variant only if its complete state at the
//     segment boundary (five accumulators of all 64 pending nodes, pending-connection marker) is BIT-IDENTICAL to
//     the true state handed over by the previous segment; block ids of the block that is open at the boundary are
//     remapped.  If neither variant matches, that segment is replayed serially from the true state.
// The result is therefore exactly the reference's, whatever the data; speculation only decides the speed.
// Per node: lane (n & 63) owns the five vote accumulators of node n; the owner's decision is one v_readlane; the
// A lanes of the following nodes add their votes with branch-free selects on the 8-byte vote records.
#define SCAN_TILE 64
#ifndef SCAN_SEG
#define SCAN_SEG 64
#endif
#ifndef SCAN_WARM
#define SCAN_WARM 64
#endif

struct Chain {
    float h1, h2, o1, o2; int vc;      // per lane: accumulators of the node this lane owns
    int lc, bs, force2;                // wave-uniform: lastConnectPos (node index), blockStart, pending label override
    int my_hp, my_blk;                 // per lane: result of the node this lane owns in the current tile
};

__device__ __forceinline__ void chain_init(Chain &c, int force2) {
    c.h1 = c.h2 = c.o1 = c.o2 = 0.f; c.vc = 0; c.lc = -1; c.bs = -1; c.force2 = force2; c.my_hp = 0; c.my_blk = -1;
}

// one node of edgeConnectResult (:306-418) for one chain.  s = owner lane of node i, rec = this lane's vote record.
__device__ __forceinline__ float vote_weight(unsigned rec) { const unsigned wc = (rec >> 3) & 3u; return wc == 3u ? 20.f : (wc == 2u ? 1.f : (wc == 1u ? 0.1f : 0.f)); }

__device__ __forceinline__ void chain_step(Chain &c, int i, int s, int l, bool gap, unsigned rec) {
    // Onelongcase override (:276), tie -> new block (:338), else argmax (:349)
    const bool use_sp = (c.vc > 3) && !(c.o1 == 0.f && c.o2 == 0.f);
    const float c1 = use_sp ? c.o1 : c.h1, c2 = use_sp ? c.o2 : c.h2;
    const int code = (c1 == c2) ? 0 : (c1 > c2 ? 1 : 2);
    const int code_s = __builtin_amdgcn_readlane(code, s);
    const bool skip = gap || (code_s == 0 && i < c.lc);                                     // :318, :340
    int hp_i = code_s;
    if (skip) hp_i = 0;
    else if (code_s == 0) { c.bs = i; hp_i = c.force2 ? 2 : 1; c.force2 = 0; }
    if (l == s) { c.my_hp = hp_i; c.my_blk = skip ? -1 : c.bs; c.h1 = c.h2 = c.o1 = c.o2 = 0.f; c.vc = 0; }
    if (!skip) {
        const float w = vote_weight(rec);
        const unsigned fl = rec & 7u;
        const bool to2 = ((fl & 1u) != 0) != (hp_i == 2);                                   // target haplotype 2
        const float wo = (fl & 4u) ? w : 0.f;
        c.h1 += to2 ? 0.f : w; c.h2 += to2 ? w : 0.f;
        c.o1 += to2 ? 0.f : wo; c.o2 += to2 ? wo : 0.f;
        c.vc += (fl >> 1) & 1u;
        const unsigned long long cm = __ballot(w != 0.f);
        if (cm) {                                                                           // lastConnectPos = last connected target (:411)
            const int sh = (s + 1) & 63;
            const unsigned long long rot = sh ? ((cm >> sh) | (cm << (64 - sh))) : cm;
            c.lc = i + 1 + (63 - __clzll(rot));
        }
    }
}

// chain_step without branches: the speculative walk steps TWO chains per node, and only straight-line code lets their instructions interleave
// (a lone wave is bound by the latency of each dependent instruction, not by issue slots).  Same arithmetic: a skipped node adds +0.f / 0,
// which leaves sums of non-negative weights bit-identical.
__device__ __forceinline__ void chain_step_flat(Chain &c, int i, int s, int l, bool gap, unsigned rec) {
    const bool use_sp = (c.vc > 3) && !(c.o1 == 0.f && c.o2 == 0.f);
    const float c1 = use_sp ? c.o1 : c.h1, c2 = use_sp ? c.o2 : c.h2;
    const int code = (c1 == c2) ? 0 : (c1 > c2 ? 1 : 2);
    const int code_s = __builtin_amdgcn_readlane(code, s);
    const bool tie_skip = code_s == 0 && i < c.lc;
    const bool skip = gap || tie_skip;                                                      // :318, :340
    const bool newblk = !skip && code_s == 0;                                               // tie -> new block (:338)
    c.bs = newblk ? i : c.bs;
    const int hp_i = skip ? 0 : (code_s == 0 ? (c.force2 ? 2 : 1) : code_s);
    c.force2 = newblk ? 0 : c.force2;
    const bool own = l == s;
    c.my_hp = own ? hp_i : c.my_hp; c.my_blk = own ? (skip ? -1 : c.bs) : c.my_blk;
    c.h1 = own ? 0.f : c.h1; c.h2 = own ? 0.f : c.h2; c.o1 = own ? 0.f : c.o1; c.o2 = own ? 0.f : c.o2; c.vc = own ? 0 : c.vc;
    const float w = skip ? 0.f : vote_weight(rec);
    const unsigned fl = skip ? 0u : (rec & 7u);
    const bool to2 = ((fl & 1u) != 0) != (hp_i == 2);                                       // target haplotype 2
    const float wo = (fl & 4u) ? w : 0.f;
    c.h1 += to2 ? 0.f : w; c.h2 += to2 ? w : 0.f;
    c.o1 += to2 ? 0.f : wo; c.o2 += to2 ? wo : 0.f;
    c.vc += (fl >> 1) & 1u;
    const unsigned long long cm = __ballot(w != 0.f);
    const int sh = (s + 1) & 63;
    const unsigned long long rot = (cm >> sh) | (cm << ((64 - sh) & 63));
    const int lc_new = i + 1 + (63 - __clzll(rot | 1ull));                                  // lastConnectPos = last connected target (:411)
    c.lc = cm ? lc_new : c.lc;
}

// state hand-over record of one chain at a node boundary (before processing node `at`)
struct ScanState { float h1[64], h2[64], o1[64], o2[64]; int vc[64]; int lc, bs, force2, pad; };

__device__ __forceinline__ void state_save(const Chain &c, ScanState *st, int l, int at) {
    st->h1[l] = c.h1; st->h2[l] = c.h2; st->o1[l] = c.o1; st->o2[l] = c.o2; st->vc[l] = c.vc;
    if (l == 0) { st->lc = max(c.lc, at); st->bs = c.bs; st->force2 = c.force2; }          // lc <= at never compares true again
}
__device__ __forceinline__ void state_load(Chain &c, const ScanState *st, int l) {
    c.h1 = st->h1[l]; c.h2 = st->h2[l]; c.o1 = st->o1[l]; c.o2 = st->o2[l]; c.vc = st->vc[l];
    c.lc = st->lc; c.bs = st->bs; c.force2 = st->force2; c.my_hp = 0; c.my_blk = -1;
}
__device__ __forceinline__ bool state_equal(const Chain &c, int at, const ScanState *st, int l) {
    const bool same = __float_as_uint(c.h1) == __float_as_uint(st->h1[l]) && __float_as_uint(c.h2) == __float_as_uint(st->h2[l]) &&
                      __float_as_uint(c.o1) == __float_as_uint(st->o1[l]) && __float_as_uint(c.o2) == __float_as_uint(st->o2[l]) && c.vc == st->vc[l];
    return __ballot(!same) == 0 && max(c.lc, at) == st->lc && c.force2 == 0 && st->force2 == 0;
}

// gap flags (:318) of the 64 nodes of a tile as a scalar mask (bit j: node t0+j must be skipped)
__device__ __forceinline__ unsigned long long tile_gapmask(const int32_t *nodes, const int32_t *vpos, int t0, int N, int distance, int l) {
    const int n_me = t0 + l;
    int my_pos = 0, my_next = 0;
    if (n_me < N) my_pos = vpos[nodes[n_me]];
    if (n_me + 1 < N) my_next = vpos[nodes[n_me + 1]];
    return __ballot((n_me + 1 < N) ? (abs(my_next - my_pos) > distance) : true);
}

// speculative walk of segment blockIdx.x by ONE wave, two label variants interleaved.  The vote bytes of the whole walk range (warm-up + segment,
// 2 x 2.2 KB at A = 35) are staged in LDS up front: with one byte per record every segment of a chromosome is resident at once (32 per CU).
__global__ __launch_bounds__(64) void k_scan_spec(const LpsCounters *cnt, const int32_t *nodes, const int32_t *vpos,
                                                  const uint8_t *erec, int A, int distance,
                                                  int8_t *hp_v /*[2][N]*/, int32_t *blk_v /*[2][N]*/, size_t vstride,
                                                  ScanState *st_b /*[seg][2]*/, ScanState *st_e /*[seg][2]*/, int warm /*multiple of 64*/) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_rec_dyn[];   // [(warm + SCAN_SEG) * A]
    const int l = lane_id();
    const int N = (int)cnt->n_nodes;
    const int seg = blockIdx.x;
    const int b = seg * SCAN_SEG;
    if (b >= N) return;
    const int a = max(0, b - warm), e = b + SCAN_SEG;               // walk [a, e); results for [b, e)
    const int last = min(e, N - 1);                                 // the last node of a contig is never processed (:308-311)
    {
        // a and e are multiples of 64, so the byte range [a * A, e * A) starts and ends on a 4-byte boundary
        const long long base = (long long)a * A, lim = (long long)N * A;
        const int n_words = (e - a) * A / 4;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(erec + base); uint32_t *dst = reinterpret_cast<uint32_t *>(s_rec_dyn);
        for (int q = l; q < n_words; q += 64) dst[q] = (base + 4ll * q < lim) ? src[q] : 0u;          // (bytes past N * A lie inside the allocation and belong to no node)
    }
    wave_sync();
    Chain c0, c1; chain_init(c0, 0); chain_init(c1, 1);
    for (int t0 = a; t0 < e; t0 += SCAN_TILE) {
        if (t0 == b) { state_save(c0, &st_b[seg * 2 + 0], l, b); state_save(c1, &st_b[seg * 2 + 1], l, b); }
        const unsigned long long gapmask = tile_gapmask(nodes, vpos, t0, N, distance, l);
        const uint8_t *rec = s_rec_dyn + (size_t)(t0 - a) * A;
        const int tend = min(SCAN_TILE, last - t0);
        int s = 0, k = (l - 1) & 63;                             // tiles are 64-aligned: owner lane of node t0+j is j
        unsigned cur = (tend > 0 && k < A) ? rec[k] : 0u;
        c0.my_hp = c1.my_hp = 0; c0.my_blk = c1.my_blk = -1;
        for (int j = 0; j < tend; ++j) {
            const int kn = (k - 1) & 63;                         // prefetch the record of node i+1
            const unsigned nxt = (j + 1 < tend && kn < A) ? rec[(j + 1) * A + kn] : 0u;
            const bool gap = (gapmask >> j) & 1ull;
            chain_step_flat(c0, t0 + j, s, l, gap, cur);
            chain_step_flat(c1, t0 + j, s, l, gap, cur);
            cur = nxt; k = kn; s = (s + 1) & 63;
        }
        if (t0 >= b && t0 + l < N) {                             // coalesced result store of the tile (both variants)
            hp_v[t0 + l] = (int8_t)c0.my_hp; blk_v[t0 + l] = c0.my_blk;
            hp_v[vstride + t0 + l] = (int8_t)c1.my_hp; blk_v[vstride + t0 + l] = c1.my_blk;
        }
    }
    state_save(c0, &st_e[seg * 2 + 0], l, e); state_save(c1, &st_e[seg * 2 + 1], l, e);
}

// wave per segment boundary: is the state a variant of segment seg-1 ends with bit-identical to the state a variant of
// segment seg starts its segment proper with?  bit (v_prev*2 + v) of match[seg].
// Also: the open-block ids of the four states of a segment side by side (bs[seg] = {start v0, start v1, end v0, end v1}) - k_scan_stitch's lone wave
// reads them for every segment, and out of the 1.3-KB state records every read was a page of its own.
__global__ __launch_bounds__(256) void k_scan_match(const LpsCounters *cnt, const ScanState *st_b, const ScanState *st_e, int32_t *match, int4 *bs) {
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), l = lane_id();
    const int N = (int)cnt->n_nodes;
    const int n_seg = (N + SCAN_SEG - 1) / SCAN_SEG;
    if (seg >= n_seg) return;
    if (l == 0) bs[seg] = make_int4(st_b[seg * 2].bs, st_b[seg * 2 + 1].bs, st_e[seg * 2].bs, st_e[seg * 2 + 1].bs);
    if (seg < 1) return;
    int bits = 0;
#pragma unroll
    for (int vp = 0; vp < 2; ++vp)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const ScanState *x = &st_e[(seg - 1) * 2 + vp], *y = &st_b[seg * 2 + v];
            const bool same = __float_as_uint(x->h1[l]) == __float_as_uint(y->h1[l]) && __float_as_uint(x->h2[l]) == __float_as_uint(y->h2[l]) &&
                              __float_as_uint(x->o1[l]) == __float_as_uint(y->o1[l]) && __float_as_uint(x->o2[l]) == __float_as_uint(y->o2[l]) && x->vc[l] == y->vc[l];
            if (__ballot(!same) == 0 && x->lc == y->lc && x->force2 == 0 && y->force2 == 0) bits |= 1 << (vp * 2 + v);
        }
    if (l == 0) match[seg] = bits;
}

// One wave decides, for every segment, which of its two variants is the true walk and how the id of the block that is open at its start has to
// be renamed.  With the boundary matches precomputed both are compositions of tiny functions - variant: {0,1} -> {0,1,none matches}, open
// block: "keep" or "becomes b" - so each lane composes the functions of its own run of segments, the 64 lane results are chained with
// v_readlane, and each lane replays its run with the true inputs: a few microseconds whatever the number of segments.  Where a segment has no
// matching variant on the true path the composition stops there: the wave replays that segment serially from the true state (records straight
// from global memory), compares states live until the walk is back on a stored state, and composes again from there - a break costs one
// segment's walk plus one more composition round, not a serial pass over all segments (one break in 5 000 segments used to cost 2.5 ms).
__global__ __launch_bounds__(64) void k_scan_stitch(const LpsCounters *cnt, const int32_t *nodes, const int32_t *vpos,
                                                    const uint8_t *erec, int A, int distance,
                                                    int8_t *hp_v, int32_t *blk_v, size_t vstride,
                                                    const ScanState *st_b, const ScanState *st_e, const int32_t *match, const int4 *bs,
                                                    int32_t *chosen /*[seg]*/, int32_t *remap_from, int32_t *remap_to, unsigned *n_replayed) {
    constexpr int SC_CH = 96;                                           // segments per lane whose states fit the LDS copy (6 144 segments = 393 k nodes; beyond: re-read)
    __shared__ int s_open[SC_CH * 64], s_end[SC_CH * 64]; __shared__ uint8_t s_mv[SC_CH * 64];
    const int l = lane_id();
    const int N = (int)cnt->n_nodes;
    const int n_seg = (N + SCAN_SEG - 1) / SCAN_SEG;
    if (n_seg == 0) return;
    // variant of segment seg that continues variant p of segment seg-1 (2: neither)
    auto next_variant = [](int m, int p) { return ((m >> (p * 2)) & 1) ? 0 : (((m >> (p * 2 + 1)) & 1) ? 1 : 2); };
    if (l == 0) { chosen[0] = 0; remap_from[0] = -2; remap_to[0] = -2; }     // segment 0 starts at node 0: its variant 0 IS the true walk
    int seg_lo = 1, v_in = 0, cur_bs = st_e[0].bs;                           // first undecided segment, variant chosen before it, block open at its start
    Chain t; chain_init(t, 0);
    while (seg_lo < n_seg) {
        // ---- composition over [seg_lo, n_seg): lane l owns the run [s0, s1)
        const int chunk = (n_seg - seg_lo + 63) / 64, s0 = min(n_seg, seg_lo + l * chunk), s1 = min(n_seg, s0 + chunk);
        int f0 = 0, f1 = 1;                                          // this lane's run as a function of the incoming variant
        // (the loads of a run do not depend on what the run computes: sixteen are requested together - one at a time, as the plain loop compiles, a
        // run of 80 segments costs 80 memory latencies, and this single wave is on every contig's critical path)
        for (int seg = s0; seg < s1; seg += 16) {
            int m[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) m[q] = match[min(seg + q, s1 - 1)];
#pragma unroll
            for (int q = 0; q < 16; ++q) if (seg + q < s1) { f0 = f0 == 2 ? 2 : next_variant(m[q], f0); f1 = f1 == 2 ? 2 : next_variant(m[q], f1); }
        }
        int vin = 0, v = v_in, J = 64;                               // J: first lane whose run holds a boundary nothing matches
        for (int j = 0; j < 64; ++j) {
            if (l == j) vin = v;
            const int a0 = __builtin_amdgcn_readlane(f0, j), a1 = __builtin_amdgcn_readlane(f1, j);
            const int nv = v == 2 ? 2 : (v ? a1 : a0);
            if (nv == 2 && v != 2) J = j;
            v = nv;
        }
        // how far this lane's run is on stored states: all of it before lane J, up to the break inside lane J, nothing behind
        int lim = l < J ? s1 : s0, pv_end = vin;
        if (l == J) { int pv = vin; lim = s1; for (int seg = s0; seg < s1; ++seg) { const int mv = next_variant(match[seg], pv); if (mv == 2) { lim = seg; break; } pv = mv; } pv_end = pv; }
        int has_set = 0, last_val = 0;                               // open block after the run: kept, or becomes last_val
        const bool cached = chunk <= SC_CH;                           // the run's (variant, open block, block at the end) stay in LDS for the pass below
        {
            int pv = vin;
            for (int seg = s0; seg < lim; seg += 8) {                 // both variants' states of eight segments requested together, then chosen
                int m[8], ob[8][2], eb[8][2];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int sg = min(seg + q, lim - 1);
                    const int4 x = bs[sg];
                    m[q] = match[sg]; ob[q][0] = x.x; ob[q][1] = x.y; eb[q][0] = x.z; eb[q][1] = x.w;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (seg + q >= lim) break;
                    const int mv = next_variant(m[q], pv), open_spec = mv ? ob[q][1] : ob[q][0], end_bs = mv ? eb[q][1] : eb[q][0];
                    if (end_bs != open_spec) { has_set = 1; last_val = end_bs; }
                    if (cached) { const int at = (seg + q - s0) * 64 + l; s_mv[at] = (uint8_t)mv; s_open[at] = open_spec; s_end[at] = end_bs; }
                    pv = mv;
                }
            }
        }
        int cin = 0, cb = cur_bs;
        for (int j = 0; j < 64; ++j) {
            if (l == j) cin = cb;
            const int hs = __builtin_amdgcn_readlane(has_set, j), lv = __builtin_amdgcn_readlane(last_val, j);
            cb = hs ? lv : cb;                                       // (lanes behind J contribute nothing: their runs are empty here)
        }
        {
            int pv = vin, cur = cin;
            for (int seg = s0; seg < lim; ++seg) {
                int mv, open_spec, end_bs;
                if (cached) { const int at = (seg - s0) * 64 + l; mv = s_mv[at]; open_spec = s_open[at]; end_bs = s_end[at]; }
                else { mv = next_variant(match[seg], pv); const int4 x = bs[seg]; open_spec = mv ? x.y : x.x; end_bs = mv ? x.w : x.z; }
                chosen[seg] = mv; remap_from[seg] = open_spec >= 0 ? open_spec : -2; remap_to[seg] = cur;
                cur = (end_bs == open_spec) ? cur : end_bs;          // the block open at b is still open at e
                pv = mv;
            }
        }
        if (J == 64) break;                                          // every boundary on the true path had a matching variant
        // ---- the boundary before segment `seg` matches nothing: walk on from the true state until it is a stored state again
        int seg = __builtin_amdgcn_readlane(lim, J);
        const int pv_star = __builtin_amdgcn_readlane(pv_end, J);
        cur_bs = cb;
        state_load(t, &st_e[(seg - 1) * 2 + pv_star], l); t.bs = cur_bs;
        bool resynced = false;
        while (seg < n_seg && !resynced) {
            const int b = seg * SCAN_SEG, e = b + SCAN_SEG;
            if (l == 0) { chosen[seg] = 0; remap_from[seg] = -2; remap_to[seg] = -2; atomicAdd(n_replayed, 1u); }
            const int last = min(e, N - 1);
            for (int t0 = b; t0 < e && t0 < N; t0 += SCAN_TILE) {
                const unsigned long long gapmask = tile_gapmask(nodes, vpos, t0, N, distance, l);
                const int tend = min(SCAN_TILE, last - t0);
                t.my_hp = 0; t.my_blk = -1;
                for (int j = 0; j < tend; ++j) {
                    const int i = t0 + j, k = (l - j - 1) & 63;
                    const unsigned rec = (k < A) ? erec[(size_t)i * A + k] : 0u;
                    chain_step(t, i, j, l, (gapmask >> j) & 1ull, rec);
                }
                if (t0 + l < N) { hp_v[t0 + l] = (int8_t)t.my_hp; blk_v[t0 + l] = t.my_blk; }
            }
            cur_bs = t.bs;
            ++seg;
            if (seg >= n_seg) break;
            int mv = -1;
            if (state_equal(t, seg * SCAN_SEG, &st_b[seg * 2 + 0], l)) mv = 0; else if (state_equal(t, seg * SCAN_SEG, &st_b[seg * 2 + 1], l)) mv = 1;
            if (mv >= 0) {                                           // back on a stored state: this segment is its variant mv
                const int open_spec = st_b[seg * 2 + mv].bs, end_bs = st_e[seg * 2 + mv].bs;
                if (l == 0) { chosen[seg] = mv; remap_from[seg] = open_spec >= 0 ? open_spec : -2; remap_to[seg] = cur_bs; }
                cur_bs = (end_bs == open_spec) ? cur_bs : end_bs;
                v_in = mv; seg_lo = seg + 1; resynced = true;
            }
        }
        if (!resynced) break;                                        // walked to the end
    }
}

// + which blocks carry a PS: a block's id is the index of its first node, so a block has a second member (PhasingGraph.cpp:423: size > 1) exactly when
// some OTHER node names it - a plain store of 1, no counting
__global__ void k_scan_finalize(const LpsCounters *cnt, const int8_t *hp_v, const int32_t *blk_v, size_t vstride,
                                const int32_t *chosen, const int32_t *remap_from, const int32_t *remap_to,
                                int8_t *hp_out, int32_t *block_out, uint8_t *bmulti) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = (int)cnt->n_nodes;
    if (i >= N) return;
    if (i == N - 1) { hp_out[i] = 0; block_out[i] = -1; return; }
    const int seg = i / SCAN_SEG, v = chosen[seg];
    int blk = blk_v[(size_t)v * vstride + i];
    if (blk >= 0 && blk == remap_from[seg]) blk = remap_to[seg];
    hp_out[i] = hp_v[(size_t)v * vstride + i]; block_out[i] = blk;
    if (blk >= 0 && blk != i) bmulti[blk] = 1;
}

// ================================================================================================ read correction
// per-node byte for the read-correction kernel: bit0 refhap (hp == 2), bits 1-3 type, bit 4 the node carries a PS (block of size > 1); + the pair count
__global__ void k_node_state(LpsCounters *cnt, const int32_t *nodes, const int32_t *block, const uint8_t *bmulti, const int8_t *hp, const uint32_t *vtype_key,
                             const uint32_t *node_pairs, uint8_t *nstate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < (int)cnt->n_nodes;
    unsigned long long pr = in ? node_pairs[i] : 0;
    pr = wave_sum(pr);
    __shared__ unsigned long long s_pr[16];
    if (lane_id() == 0) s_pr[threadIdx.x >> 6] = pr;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long tot = 0; for (unsigned q = 0; q < (blockDim.x + 63) / 64; ++q) tot += s_pr[q]; if (tot) atomicAdd(&cnt->n_pairs, tot); }
    if (!in) return;
    const int b = block[i];
    nstate[i] = (uint8_t)((hp[i] == 1 ? 0 : 1) | ((vtype_key[nodes[i]] & 7u) << 1) | ((b >= 0 && bmulti[b]) ? 16u : 0u));
}

// wave per alignment: readCorrection's per-read vote (:904-959).  SNP sites contribute integers (order-free, ballot +
// popcount); as soon as the read touches an indel site the 0.1 contributions are summed by one lane in the reference's
// order (doubles), so the result is bit-identical either way.
__global__ __launch_bounds__(256) void k_read_correction(int n_reads, const RowDesc *rows, const int32_t *g_cnt, const uint32_t *g_pack,
                                                         const uint8_t *nstate, double read_confidence, uint32_t *cnt4) {
    const int l = lane_id(), grp = l / ROW_G, sl = l % ROW_G;
    const int r = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE + grp;
    if (r >= n_reads) return;
    const int n = g_cnt[r];
    if (n <= 0) return;
    const uint32_t off = rows[r].off;
    int irc = 0, iac = 0; bool indel = false;
    for (int k0 = 0; k0 < n; k0 += ROW_G) {
        const int k = k0 + sl;
        int st = 0, al = 0;
        if (k < n) { const uint32_t wd = g_pack[off + k]; st = nstate[wd >> 2]; al = (int)(wd & 1u); }
        const bool live = (st & 16) != 0;
        const int ty = (st >> 1) & 7;
        const int h = al ^ (st & 1);                          // 0: haplotype 1, 1: haplotype 2
        irc += __popcll(group_ballot(live && ty <= 1 && h == 0, grp));
        iac += __popcll(group_ballot(live && ty <= 1 && h == 1, grp));
        indel |= group_ballot(live && ty >= 3, grp) != 0;
    }
    double rc = irc, ac = iac;
    if (indel) {                                              // exact order of the reference's double sums
        rc = 0; ac = 0;
        for (int k = 0; k < n; ++k) {
            const uint32_t wd = g_pack[off + k]; const int st = nstate[wd >> 2]; const int al = (int)(wd & 1u);
            if (!(st & 16)) continue;
            const int ty = (st >> 1) & 7, h = al ^ (st & 1);
            if (ty <= 1) { if (h == 0) rc++; else ac++; }
            else if (ty == 3 || ty == 4) { if (h == 0) rc += 0.1; else ac += 0.1; }
        }
    }
    if (fmax(rc, ac) / (rc + ac) > read_confidence && (rc + ac) > 1) {
        const int bh = (rc > ac) ? 0 : 1;
        for (int k = sl; k < n; k += ROW_G) { const uint32_t wd = g_pack[off + k]; atomicAdd(&cnt4[(size_t)(wd >> 2) * 4 + bh * 2 + (wd & 1u)], 1u); }
    }
}

__global__ void k_final(const LpsCounters *cnt, const int32_t *nodes, const int32_t *vpos, const int32_t *block,
                        const uint8_t *bmulti, const uint32_t *cnt4, double snp_confidence, int32_t *out_ps, uint8_t *out_gt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int)cnt->n_nodes) return;
    const int b = block[i];
    if (b < 0 || !bmulti[b]) return;
    const uint32_t *c = cnt4 + (size_t)i * 4;
    const double r1 = (double)c[0] + (double)c[3], r2 = (double)c[2] + (double)c[1];
    const double conf = fmax(r1, r2) / (r1 + r2);
    int g = -1;
    if (conf > snp_confidence) { if (r1 > r2) g = 0; else if (r1 < r2) g = 1; }
    if (g != -1) { out_ps[nodes[i]] = vpos[nodes[b]] + 1; out_gt[nodes[i]] = (uint8_t)g; }
}

// ================================================================================================ host side
size_t GraphTemp::need(size_t n_sort) {
    size_t a = 0, b = 0, c = 0;
    unsigned long long *k = nullptr; uint32_t *v = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, a, k, k, n_sort, 0, 64, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, b, k, k, v, v, n_sort, 0, 64, nullptr);
    (void)rocprim::exclusive_scan(nullptr, c, v, v, 0u, n_sort, rocprim::plus<uint32_t>(), nullptr);
    size_t d = 0; uint8_t *u = nullptr;
    (void)rocprim::exclusive_scan(nullptr, d, u, u, (uint8_t)2, n_sort, CnvCompose(), nullptr);
    return std::max(std::max(a, d), std::max(b, c)) + 256;
}

void sort_keys64(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, size_t n, int bits,
                 hipStream_t s) {
    if (n == 0) return;
    HIP_TRY(rocprim::radix_sort_keys(temp, temp_bytes, in, out, n, 0, bits, s));
}
void sort_keys64_range(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, size_t n, int begin_bit, int end_bit,
                       hipStream_t s) {
    if (n == 0) return;
    HIP_TRY(rocprim::radix_sort_keys(temp, temp_bytes, in, out, n, (unsigned)begin_bit, (unsigned)std::min(end_bit, 64), s));
}
void sort_pairs64(void *temp, size_t temp_bytes, const unsigned long long *kin, unsigned long long *kout, const uint32_t *vin,
                  uint32_t *vout, size_t n, int bits, hipStream_t s) {
    if (n == 0) return;
    HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, 0, bits, s));
}
void exscan_u32(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t s) {
    if (n == 0) return;
    HIP_TRY(rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s));
}

#define GRID(n, b) dim3((unsigned)(((n) + (b) - 1) / (b))), dim3(b)

void launch_cnv_filter(const LpsCounters *cnt, int n_reads, int n_var, const RowDesc *rows,
                       const uint8_t *deleted, ObsRec *obs, const int32_t *vpos,
                       const int32_t *cnv_start, const int32_t *cnv_end, long long *agg_sum, int32_t *agg_cnt, double *miss,
                       CnvScratch &W, uint32_t *var_del2, void *temp, size_t temp_bytes, hipStream_t s) {
    HIP_TRY(hipMemsetAsync(agg_sum, 0, (size_t)n_var * 2 * sizeof(long long), s));
    HIP_TRY(hipMemsetAsync(agg_cnt, 0, (size_t)n_var * 2 * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_cnv_list, GRID(n_reads, 256), 0, s, cnt, n_reads, rows, deleted, W.flag);
    exscan_u32(temp, temp_bytes, W.flag, W.idx, n_reads, s);
    hipLaunchKernelGGL(k_cnv_compact, GRID(n_reads, 256), 0, s, cnt, n_reads, W.flag, W.idx, W.list, W.n_list);
    // entry half of every kept alignment = (composition of the transfer functions of the alignments before it)(half 0); at most n_reads entries are
    // scanned - the tail beyond n_list holds the identity
    const unsigned row_grid = (unsigned)((n_reads + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    for (int pass4 = 0; pass4 < 2; ++pass4) {
        HIP_TRY(hipMemsetAsync(W.fn, 2, (size_t)n_reads, s));
        if (pass4) hipLaunchKernelGGL(k_cnv_rows<false>, dim3(row_grid), dim3(256), 0, s, cnt, W.list, W.n_list, W.pre, rows, obs, vpos, cnv_start, cnv_end, miss, W.fn, var_del2);
        else hipLaunchKernelGGL(k_cnv_fn12, GRID(n_reads, 128), 0, s, cnt, W.list, W.n_list, rows, obs, vpos, cnv_start, W.fn);
        size_t need = temp_bytes;
        HIP_TRY(rocprim::exclusive_scan(temp, need, W.fn, W.pre, (uint8_t)2, (size_t)n_reads, CnvCompose(), s));
        if (!pass4) {
            hipLaunchKernelGGL(k_cnv_count, GRID(n_reads, 128), 0, s, cnt, W.list, W.n_list, W.pre, rows, obs, vpos, cnv_start, cnv_end, (unsigned long long *)agg_sum, agg_cnt);
            hipLaunchKernelGGL(k_cnv_miss, GRID(n_var, 256), 0, s, cnt, n_var, vpos, cnv_start, cnv_end, (const unsigned long long *)agg_sum, agg_cnt, miss);
        } else {
            hipLaunchKernelGGL(k_cnv_rows<true>, dim3(row_grid), dim3(256), 0, s, cnt, W.list, W.n_list, W.pre, rows, obs, vpos, cnv_start, cnv_end, miss, W.fn, var_del2);
        }
    }
}

void launch_clip_sort(unsigned n_clips, unsigned long long *keys, unsigned long long *keys_sorted, void *temp, size_t temp_bytes, hipStream_t s) {
    if (n_clips) sort_keys64(temp, temp_bytes, keys, keys_sorted, n_clips, 33, s);          // key = pos << 1 | front/back: 33 bits, 5 digit passes instead of 8
}

// name ids -> dense ranks (only when the caller's ids are not dense): sort of (id, alignment), heads, scan
void launch_dense_names(int n_reads, const uint32_t *name_id, uint32_t name_max, unsigned long long *keys, unsigned long long *keys_s, uint32_t *head, uint32_t *gidx,
                        uint32_t *dense, void *temp, size_t temp_bytes, hipStream_t s) {
    if (!n_reads) return;
    int bits = 1; while ((1ull << bits) < (unsigned long long)name_max + 2) ++bits;
    hipLaunchKernelGGL(k_dense_keys, GRID(n_reads, 256), 0, s, n_reads, name_id, keys);
    sort_keys64_range(temp, temp_bytes, keys, keys_s, n_reads, 32, 32 + bits, s);   // stable by the name digits alone: equal names stay in alignment order
    hipLaunchKernelGGL(k_dense_heads, GRID(n_reads, 256), 0, s, keys_s, n_reads, head);
    exscan_u32(temp, temp_bytes, head, gidx, n_reads, s);
    hipLaunchKernelGGL(k_dense_ids, GRID(n_reads, 256), 0, s, keys_s, head, gidx, n_reads, dense);
}

void launch_names(const GraphView &G, const ClipView &C, unsigned long long *clip_keys, const unsigned long long *arena_ctr, unsigned long long arena_size, uint32_t *clip_tab, hipStream_t s) {
    if (!G.n_reads) return;
    const int nb_reads = (G.n_reads + NAMES_B - 1) / NAMES_B;
    hipLaunchKernelGGL(k_name_link, dim3(nb_reads + 1 + std::min(512, nb_reads)), dim3(NAMES_B), 0, s, G.n_reads, G.name, G.rows, G.name_head, G.name_link, G.cnt,
            arena_ctr, arena_size, C, clip_keys, nb_reads, clip_tab);
}

void launch_groups(const GraphView &G, double overlap_threshold, bool counted, hipStream_t s) {
    if (!G.n_reads) return;
    hipLaunchKernelGGL(k_groups, GRID(G.n_reads, GROUPS_B), 0, s, G.n_reads, G.name, G.rows, G.obs, G.vpos, overlap_threshold, G.name_head, G.name_link, G.cnt,
                       G.mm_r, G.stack, G.mg_start, G.mg_cnt, G.mg_name, G.deleted, counted ? G.var_del : nullptr);
}

void launch_count_ranks(const GraphView &G, hipStream_t s) {
    if (G.n_reads) hipLaunchKernelGGL(k_count_ranks, dim3((G.n_reads + 3) / 4), dim3(256), 0, s, G.n_reads, G.rows, G.deleted, G.obs, G.var_cnt, G.var_del, G.cnt);
}

// node numbers + list offsets, the graph view of the rows with the entries of single-alignment reads, the merged rows of the others with theirs
void launch_var_scan(const GraphView &G, hipStream_t s) {
    const int nvb = (G.n_var + VSCAN_B - 1) / VSCAN_B;
    hipLaunchKernelGGL(k_var_sums, dim3(nvb), dim3(VSCAN_B), 0, s, G.n_var, G.var_cnt, G.var_del, G.var_del2, G.bsum);
    hipLaunchKernelGGL(k_var_scan, dim3(nvb), dim3(VSCAN_B), 0, s, G.n_var, G.var_cnt, G.var_del, G.var_del2, G.bsum, G.node_of, G.var_off, G.nodes, G.node_off, G.node_cap, G.node_end, G.cnt);
}
void launch_graph_rows(const GraphView &G, int base_quality, int a_bits, bool key64, unsigned n_multi, hipStream_t s) {
    const int nb_reads = round_up8((G.n_reads + 15) / 16);                // workgroups of four extraction jobs (16 rows)
    if (key64) hipLaunchKernelGGL(k_graph_rows<true>, dim3(nb_reads), dim3(256), 0, s, G.n_reads, G.rows, G.deleted, G.obs, G.name, G.name_head, G.name_link, G.node_of,
            G.var_off, base_quality, a_bits, G.g_pack, G.g_rank, G.g_cnt, G.mrow_off, G.mrow_cnt, G.ukeys, G.uvals, G.vtype_key, nb_reads);
    else hipLaunchKernelGGL(k_graph_rows<false>, dim3(nb_reads), dim3(256), 0, s, G.n_reads, G.rows, G.deleted, G.obs, G.name, G.name_head, G.name_link, G.node_of,
            G.var_off, base_quality, a_bits, G.g_pack, G.g_rank, G.g_cnt, G.mrow_off, G.mrow_cnt, G.ukeys, G.uvals, G.vtype_key, nb_reads);
    if (!n_multi) return;                                                 // (known on the host since the overlap filter)
    hipLaunchKernelGGL(k_merge_plan, GRID(n_multi, 256), 0, s, G.cnt, G.mg_start, G.mg_cnt, G.mg_name, G.mm_r, G.rows, G.g_cnt, G.tail_lo, G.tail_size, G.mrow_off, G.mrow_cnt, G.mg_plan);
    const unsigned mb = std::min(1024u, (n_multi + 3) / 4);
    if (key64) hipLaunchKernelGGL(k_merge_multi<true>, dim3(mb), dim3(256), 0, s, G.cnt, G.mg_start, G.mg_cnt, G.mg_name, G.mg_plan, G.mm_r, G.rows, G.g_cnt, G.g_pack,
            G.g_rank, G.t_node, G.t_flag, G.t_src, (uint32_t)G.tail_lo, G.mrow_off, G.nodes, G.var_off, a_bits, G.ukeys, G.uvals);
    else hipLaunchKernelGGL(k_merge_multi<false>, dim3(mb), dim3(256), 0, s, G.cnt, G.mg_start, G.mg_cnt, G.mg_name, G.mg_plan, G.mm_r, G.rows, G.g_cnt, G.g_pack,
            G.g_rank, G.t_node, G.t_flag, G.t_src, (uint32_t)G.tail_lo, G.mrow_off, G.nodes, G.var_off, a_bits, G.ukeys, G.uvals);
}

void launch_edges(const GraphView &G, int m_bits, int a_bits, bool key64, double edge_weight, double edge_threshold, hipStream_t s) {
    const dim3 grid((((G.n_var + 3) / 4 + 7) / 8) * 8);
    const bool soff = G.tail_lo + G.tail_size < (1ull << 30);
#define LPS_EDGES(K, S) hipLaunchKernelGGL((k_edges<K, S>), grid, dim3(256), 0, s, G.cnt, G.node_off, G.node_cap, G.node_end, G.ukeys, G.uvals, G.skeys, G.svals, G.mrow_off, G.mrow_cnt, m_bits, a_bits, G.g_pack, (uint32_t)G.tail_lo, G.A, edge_weight, edge_threshold, G.nodes, G.vtype_key, G.edge, G.erec, G.node_pairs)
    if (key64) { if (soff) LPS_EDGES(true, true); else LPS_EDGES(true, false); }
    else { if (soff) LPS_EDGES(false, true); else LPS_EDGES(false, false); }
#undef LPS_EDGES
}

size_t scan_state_bytes(int n_var) { return (size_t)((n_var + SCAN_SEG - 1) / SCAN_SEG + 1) * 2 * sizeof(ScanState); }
int scan_segments(int n_var) { return (n_var + SCAN_SEG - 1) / SCAN_SEG + 1; }

// test hook (LPS_SCAN_FORCE_REPLAY=k): pretend that every k-th boundary has no matching variant, so that the serial replay path of k_scan_stitch runs
__global__ void k_scan_break_matches(int32_t *match, int segs, int every) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 0 && i < segs && i % every == 1 % every) match[i] = 0;
}

void launch_vote_scan(const LpsCounters *cnt, int n_var, const int32_t *nodes, const int32_t *vpos, const uint8_t *erec,
                      int A, int distance, int8_t *hp_v, int32_t *blk_v, void *st_b, void *st_e, int32_t *seg_i32 /*8*segs + 16*/,
                      unsigned *n_replayed, int8_t *hp, int32_t *block, uint8_t *bmulti, int warm_tiles, hipStream_t s) {
    const int segs = scan_segments(n_var);
    // warm-up of the speculative walks: SCAN_WARM nodes as a rule.  With SV / MOD rows in the graph (sparser votes across the SNP<->MOD threshold,
    // more ties) twice that: 41 of 1 454 boundaries of a chr20-sized graph missed with 64 nodes, none with 128, and a miss costs a 37-us replay
    const int warm = SCAN_WARM * std::max(1, warm_tiles);
    const size_t vstride = (size_t)n_var + 64;
    hipLaunchKernelGGL(k_scan_spec, dim3(segs), dim3(64), (size_t)(warm + SCAN_SEG) * A, s, cnt, nodes, vpos, erec, A, distance, hp_v, blk_v, vstride, (ScanState *)st_b, (ScanState *)st_e, warm);
    int4 *seg_bs = reinterpret_cast<int4 *>(seg_i32 + 4 * (((size_t)segs + 3) / 4 * 4));     // behind the four int arrays, 16-byte aligned
    hipLaunchKernelGGL(k_scan_match, dim3((segs + 3) / 4), dim3(256), 0, s, cnt, (const ScanState *)st_b, (const ScanState *)st_e, seg_i32 + 3 * segs, seg_bs);
    if (const char *e = getenv("LPS_SCAN_FORCE_REPLAY")) { const int every = atoi(e); if (every > 0) hipLaunchKernelGGL(k_scan_break_matches, GRID(segs, 256), 0, s, seg_i32 + 3 * segs, segs, every); }
    hipLaunchKernelGGL(k_scan_stitch, dim3(1), dim3(64), 0, s, cnt, nodes, vpos, erec, A, distance, hp_v, blk_v, vstride, (const ScanState *)st_b,
            (const ScanState *)st_e, seg_i32 + 3 * segs, seg_bs, seg_i32, seg_i32 + segs, seg_i32 + 2 * segs, n_replayed);
    hipLaunchKernelGGL(k_scan_finalize, GRID(n_var, 256), 0, s, cnt, hp_v, blk_v, vstride, seg_i32, seg_i32 + segs, seg_i32 + 2 * segs, hp, block, bmulti);
}

void launch_correction(const GraphView &G, const int32_t *block, const uint8_t *bmulti, const int8_t *hp, uint8_t *nstate, double read_conf, double snp_conf,
                       uint32_t *cnt4, int32_t *out_ps, uint8_t *out_gt, hipStream_t s) {
    hipLaunchKernelGGL(k_node_state, GRID(G.n_var, 1024), 0, s, G.cnt, G.nodes, block, bmulti, hp, G.vtype_key, G.node_pairs, nstate);
    hipLaunchKernelGGL(k_read_correction, dim3((G.n_reads + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, G.n_reads, G.rows, G.g_cnt, G.g_pack, nstate, read_conf, cnt4);
    hipLaunchKernelGGL(k_final, GRID(G.n_var, 256), 0, s, G.cnt, G.nodes, G.vpos, block, bmulti, cnt4, snp_conf, out_ps, out_gt);
}
