// lps_graph.hip — haplotype-graph kernels of `phase` (gfx950, wave64).
//
// Replaces (reference file:line, relative to /root/reference/):
//   Clip::getCNVInterval (x2)            src/phase/PhasingGraph.cpp:1103-1227  -> k_clip_keys + sort + k_cnv_state
//   VairiantGraph::addEdge overlap filter src/phase/PhasingGraph.cpp:707-781   -> k_name_keys + sort + k_group_* + k_overlap_filter
//   addEdge type tagging / node set       :793-846                             -> k_mark_nodes + scan + k_graph_obs
//   addEdge pair loop + addSubEdge        :848-888, :25-70                     -> k_merge_rows + k_node_keys + sort + k_edges
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
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "lps_graph.h"

// ================================================================================================ clips / CNV
__global__ void k_clip_keys(ClipView C, const int32_t *row_fail, unsigned n_clips, unsigned long long *keys) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_clips) return;
    const int opidx = C.opidx_fb[i] >> 1, fb = C.opidx_fb[i] & 1;
    // get_snp returning early (:1453-1455,1559-1561) keeps only clips of ops before the failing op
    const bool ok = opidx < row_fail[C.read[i]];
    keys[i] = ok ? (((unsigned long long)(unsigned)C.pos[i] << 1) | (unsigned)fb) : ~0ull;
}

struct CnvState {
    bool push, slowUp, slowDown; int curr, reject, pullDown, slowDownCount, candStart, candEnd;
    __device__ void reset() { push = slowUp = slowDown = false; curr = reject = pullDown = slowDownCount = 0; candStart = candEnd = -1; }
    __device__ void threshold(int up) {
        reject = up;
        if (up >= 20) { pullDown = up / 2; slowDownCount = 5; }
        else if (up >= 10) { pullDown = up / 2; slowDownCount = up / 4; }
        else { pullDown = 5; slowDownCount = 2; }
    }
};

// One thread: run-length the sorted clip keys into (pos, up, down) and replay the CNV state machine, twice
// (Clip ctor + PhasingProcess.cpp:148), appending [start,end] pairs.  Sequential by nature, O(#clipped reads).
__global__ void k_cnv_state(const unsigned long long *keys, unsigned n_clips, int32_t *cnv_start, int32_t *cnv_end,
                            LpsCounters *cnt) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned n = 0; while (n < n_clips && keys[n] != ~0ull) ++n;    // invalid keys sort last
    int n_cnv = 0;
    if (n == 0) { cnt->n_cnv = 0; cnt->ub_hazard += 1; return; }   // reference: UB on empty ClipCount
    const int Area = 30000;
    for (int rep = 0; rep < 2; ++rep) {
        CnvState s; s.reset();
        unsigned i = 0; bool sentinel_done = false; int last_up = 0, last_down = 0, last_pos = 0;
        while (true) {
            int pos, up = 0, down = 0;
            if (i < n) {
                pos = (int)(keys[i] >> 1);
                while (i < n && (int)(keys[i] >> 1) == pos) { if (keys[i] & 1) ++down; else ++up; ++i; }
                last_up = up; last_down = down; last_pos = pos;
            } else if (!sentinel_done) { pos = last_pos + Area; up = last_up; down = last_down; sentinel_done = true; }   // :1134
            else break;
            if (!s.push && !s.slowDown && !s.slowUp) {
                if (up >= 5 && s.curr == 0) { s.push = true; s.slowUp = false; s.slowDown = true; s.curr = up - down; s.candStart = pos; s.candEnd = pos + Area; s.threshold(up); }
                else if (up > down && s.curr == 0) { s.push = false; s.slowUp = true; s.slowDown = false; s.curr = up - down; s.candStart = pos; s.candEnd = pos + Area; }
            } else if (s.push && s.slowDown) {
                if (up > s.reject) { s.threshold(up); s.candStart = pos; s.candEnd = pos + Area; }
                s.curr = s.curr + up - down;
                if (s.curr > 30) s.candEnd = pos + Area;
                bool emitted = false;
                if (down >= s.pullDown) emitted = true;
                else if (s.curr <= s.slowDownCount && pos <= s.candEnd) emitted = true;
                if (emitted) { if (n_cnv < LPS_MAX_CNV) { cnv_start[n_cnv] = s.candStart; cnv_end[n_cnv] = pos; } ++n_cnv; s.reset(); }
                if (pos > s.candEnd || s.curr <= 0 || pos - s.candStart >= 200000) s.reset();
            } else if (s.slowUp) {
                if (s.curr > 20 ? down >= s.curr / 4 : down >= 5) { if (n_cnv < LPS_MAX_CNV) { cnv_start[n_cnv] = s.candStart; cnv_end[n_cnv] = pos; } ++n_cnv; s.reset(); }
                else if (up >= 5) { s.push = true; s.slowUp = false; s.slowDown = true; s.curr = up - down; s.candStart = pos; s.candEnd = pos + Area; s.threshold(up); }
                else {
                    s.curr = s.curr + up - down;
                    if (s.curr > 30) s.candEnd = pos + Area;
                    if (pos > s.candEnd || s.curr <= 0 || pos - s.candStart >= 200000) s.reset();
                }
            }
        }
    }
    if (n_cnv > LPS_MAX_CNV) { atomicOr(&cnt->err, (unsigned)LPS_ERR_CNV_CAP); n_cnv = LPS_MAX_CNV; }
    cnt->n_cnv = n_cnv;
}

// ================================================================================================ name groups
__global__ void k_name_keys(int n_reads, const uint32_t *name_id, const int32_t *row_cnt, unsigned long long *keys,
                            LpsCounters *cnt) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const bool kept = row_cnt[r] > 0;
    keys[r] = kept ? ((unsigned long long)name_id[r] << 32 | (unsigned)r) : ~0ull;
    if (kept) atomicAdd(&cnt->n_kept, 1u);
}

__global__ void k_group_heads(const unsigned long long *skeys, int n_reads, const LpsCounters *cnt, uint32_t *head) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_reads) return;
    const unsigned nk = cnt->n_kept;
    head[s] = ((unsigned)s < nk && (s == 0 || (skeys[s] >> 32) != (skeys[s - 1] >> 32))) ? 1u : 0u;
}

// gidx = exclusive scan of head; group of sorted slot s is gidx[s] + head[s] - 1
__global__ void k_group_starts(const unsigned long long *skeys, const uint32_t *head, const uint32_t *gidx, int n_reads,
                               LpsCounters *cnt, uint32_t *gstart, uint32_t *read_group) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_reads) return;
    const unsigned nk = cnt->n_kept;
    if ((unsigned)s >= nk) return;
    const uint32_t g = gidx[s] + head[s] - 1;
    if (head[s]) gstart[g] = s;
    read_group[(uint32_t)skeys[s]] = g;
    if ((unsigned)s == nk - 1) { gstart[g + 1] = nk; cnt->n_groups = g + 1; }
}

// Overlap filter of several alignments of one read name: one thread replays the reference's sequential rule for
// its group (groups with one alignment have nothing to do).  `stack` is scratch aligned with the sorted slots.
__global__ void k_overlap_filter(const unsigned long long *skeys, const uint32_t *gstart, const LpsCounters *cnt,
                                 const uint32_t *row_off, const int32_t *row_cnt, const int32_t *obs_var,
                                 const int32_t *vpos, double overlap_threshold, uint32_t *stack, uint8_t *deleted) {
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= cnt->n_groups) return;
    const uint32_t s0 = gstart[g], s1 = gstart[g + 1];
    if (s1 - s0 < 2) return;
    uint32_t *kept = stack + s0; int nk = 0; int second = 0;
    auto fpos = [&](uint32_t r) { return vpos[obs_var[row_off[r]]]; };
    auto lpos = [&](uint32_t r) { return vpos[obs_var[row_off[r] + row_cnt[r] - 1]]; };
    for (uint32_t s = s0; s < s1; ++s) {
        const uint32_t r = (uint32_t)skeys[s];
        const int fp = fpos(r), lp = lpos(r);
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
        if (del) deleted[r] = 1; else kept[nk++] = r;
    }
}

// ================================================================================================ CNV filter
// The four CNV mismatch-rate passes (PhasingGraph.cpp:520-692).  The reference carries ONE interval cursor from
// read to read (and, in the last pass, from variant to variant) over the interval list that holds every interval
// twice (getCNVInterval runs twice), so which intervals a read "visits" depends on all reads before it.
// Round-1 implementation: a single thread replays the cursor exactly (only launched when intervals exist);
// the counting itself is order-free integer arithmetic.  TODO(next round): function-composition scan of the cursor.
__global__ void k_cnv_filter_serial(const LpsCounters *cnt, int n_reads, const uint32_t *row_off, const int32_t *row_cnt,
                                    const uint8_t *deleted, int32_t *obs_var, const uint16_t *obs_aq, const int32_t *vpos,
                                    const int32_t *cnv_start, const int32_t *cnv_end, long long *agg_sum /*[nV][2]*/,
                                    int32_t *agg_cnt /*[nV][2]*/, double *miss /*[nV], <0 = undefined*/, int n_var) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int nc = (int)cnt->n_cnv;
    if (nc == 0) return;
    const int K = nc / 2;                                    // unique intervals; list = L ++ L
    int mm[LPS_MAX_CNV]; bool has[LPS_MAX_CNV];
    auto inr = [](int p, int s, int e) { return p >= s && p <= e; };
    // passes 1+2 share the cursor evolution, so they are fused per read
    int ci = 0;
    for (int r = 0; r < n_reads; ++r) {
        const int n = row_cnt[r];
        if (n <= 0 || deleted[r]) continue;
        const uint32_t off = row_off[r];
        const int rs = vpos[obs_var[off]], re = vpos[obs_var[off + n - 1]];
        while (ci > 0 && cnv_start[ci] > rs) --ci;
        for (int j = 0; j < K; ++j) { mm[j] = 0; has[j] = false; }
        int i = ci;
        while (i < nc && cnv_start[i] <= re) {                               // calculateCnvMismatchRate
            for (int k = 0; k < n; ++k) {
                const int p = vpos[obs_var[off + k]];
                if (p > cnv_end[i]) break;
                if (inr(p, cnv_start[i], cnv_end[i]) && aq_allele(obs_aq[off + k]) == 1) { mm[i % K]++; has[i % K] = true; }
            }
            ++i;
        }
        i = ci;
        while (i < nc && cnv_start[i] <= re) {                               // aggregateCnvReadMismatchRate
            for (int k = 0; k < n; ++k) {
                const int v = obs_var[off + k]; const int p = vpos[v];
                if (p > cnv_end[i]) break;
                if (inr(p, cnv_start[i], cnv_end[i]) && has[i % K]) {
                    const int al = aq_allele(obs_aq[off + k]);
                    agg_sum[(size_t)v * 2 + al] += mm[i % K]; agg_cnt[(size_t)v * 2 + al] += 1;
                }
            }
            ++i;
        }
        ci = i > 0 ? i - 1 : 0;
    }
    // calculateAverageMismatchRate: cursor never moves in the reference (no write-back), scan from 0
    bool any = false;
    for (int v = 0; v < n_var; ++v) {
        miss[v] = -1.0;
        if (agg_cnt[(size_t)v * 2] == 0 && agg_cnt[(size_t)v * 2 + 1] == 0) continue;
        const int p = vpos[v];
        for (int i = 0; i < nc; ++i) {
            if (cnv_start[i] > p) break;
            if (inr(p, cnv_start[i], cnv_end[i]) && agg_cnt[(size_t)v * 2] > 0 && agg_cnt[(size_t)v * 2 + 1] > 0) {
                const double a = (double)agg_sum[(size_t)v * 2] / (double)agg_cnt[(size_t)v * 2];
                const double c = (double)agg_sum[(size_t)v * 2 + 1] / (double)agg_cnt[(size_t)v * 2 + 1];
                if (a != 0 && c != 0) { miss[v] = c / (a + c); any = true; }
            }
        }
    }
    if (!any) return;
    ci = 0;                                                                  // filterHighMismatchVariants
    for (int r = 0; r < n_reads; ++r) {
        const int n = row_cnt[r];
        if (n <= 0 || deleted[r]) continue;
        const uint32_t off = row_off[r];
        const int rs = vpos[obs_var[off]];
        while (ci > 0 && cnv_start[ci] > rs) --ci;
        for (int k = 0; k < n; ++k) {
            const int v = obs_var[off + k]; const int p = vpos[v];
            int i = ci;
            while (i < nc && cnv_start[i] <= p) {
                if (inr(p, cnv_start[i], cnv_end[i]) && miss[v] >= 0.7) { obs_var[off + k] = -1 - v; break; }
                ++i;
            }
            ci = i > 0 ? i - 1 : 0;
        }
    }
}

// ================================================================================================ nodes
// wave per alignment: mark observed variants as graph nodes, record the type written by the LAST alignment
// (BAM order) that observes the position - the reference's (*variantType)[pos] = ... is last-writer-wins.
__global__ __launch_bounds__(256) void k_mark_nodes(int n_reads, const uint32_t *row_off, const int32_t *row_cnt,
                                                    const uint8_t *deleted, const int32_t *obs_var, const uint16_t *obs_aq,
                                                    uint32_t *is_node, uint32_t *vtype_key) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), l = lane_id();
    if (r >= n_reads) return;
    const int n = row_cnt[r];
    if (n <= 0 || deleted[r]) return;
    const uint32_t off = row_off[r];
    for (int k = l; k < n; k += 64) {
        const int v = obs_var[off + k]; const int q = aq_quality(obs_aq[off + k]);
        if (v < 0) continue;                                  // erased by the CNV filter
        const unsigned ty = (q == -4) ? 3u : (q == -5 ? 4u : 0u);
        is_node[v] = 1u;
        atomicMax(&vtype_key[v], ((unsigned)r << 3) | ty);
    }
}

__global__ void k_node_list(int n_var, const uint32_t *is_node, const uint32_t *node_of, const uint32_t *vtype_key,
                            int32_t *nodes, uint8_t *ntype, LpsCounters *cnt) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_var) return;
    if (is_node[v]) { nodes[node_of[v]] = v; ntype[node_of[v]] = (uint8_t)(vtype_key[v] & 7u); }
    if (v == n_var - 1) cnt->n_nodes = node_of[v] + is_node[v];
}

// wave per alignment: graph view of the observations (node index, allele, hi-quality flag) in the same slots
__global__ __launch_bounds__(256) void k_graph_obs(int n_reads, const uint32_t *row_off, const int32_t *row_cnt,
                                                   const uint8_t *deleted, const int32_t *obs_var, const uint16_t *obs_aq,
                                                   const uint32_t *node_of, int base_quality, int32_t *g_node, uint8_t *g_flag,
                                                   int32_t *g_cnt, LpsCounters *cnt) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), l = lane_id();
    if (r >= n_reads) return;
    const int n = row_cnt[r];
    if (n <= 0 || deleted[r]) { if (l == 0) g_cnt[r] = 0; return; }
    const uint32_t off = row_off[r];
    // compact in place (CNV-erased entries have var = -1); rows are wave-private so a running count suffices
    int w = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + l;
        int v = -1; uint16_t aq = 0;
        if (k < n) { v = obs_var[off + k]; aq = obs_aq[off + k]; }
        const bool ok = v >= 0;
        const unsigned long long m = __ballot(ok);
        if (ok) {
            int q = aq_quality(aq); if (q < 0) q = 60;          // indel sentinels -> quality 60 (:820-828)
            const uint32_t slot = off + w + __popcll(m & lanemask_lt());
            g_node[slot] = (int32_t)node_of[v];
            g_flag[slot] = (uint8_t)(aq_allele(aq) | ((q >= base_quality) ? 2 : 0));
        }
        w += __popcll(m);
    }
    if (l == 0) { g_cnt[r] = w; atomicAdd(&cnt->n_obs_final, (unsigned long long)w); }
}

// ================================================================================================ merged rows
// thread per name group: one surviving alignment -> its own row; several -> concatenate in BAM order and sort by
// position (== node index) into a freshly reserved tail row.  Insertion sort == libstdc++ std::sort for n<=16 and
// differs from it only in the relative order of equal positions beyond that.
__global__ void k_merge_rows(const unsigned long long *skeys, const uint32_t *gstart, LpsCounters *cnt,
                             const uint32_t *row_off, const int32_t *g_cnt, int32_t *g_node, uint8_t *g_flag,
                             unsigned long long capacity, uint32_t *mrow_off, int32_t *mrow_cnt) {
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= cnt->n_groups) return;
    const uint32_t s0 = gstart[g], s1 = gstart[g + 1];
    int alive = 0, total = 0; uint32_t one = 0;
    for (uint32_t s = s0; s < s1; ++s) { const uint32_t r = (uint32_t)skeys[s]; if (g_cnt[r] > 0) { ++alive; total += g_cnt[r]; one = r; } }
    if (alive == 0) { mrow_off[g] = 0; mrow_cnt[g] = 0; return; }
    if (alive == 1) { mrow_off[g] = row_off[one]; mrow_cnt[g] = total; return; }
    const unsigned long long base = atomicAdd(&cnt->obs_total, (unsigned long long)total);
    if (base + total > capacity) { atomicOr(&cnt->err, (unsigned)LPS_ERR_OBS_OVERFLOW); mrow_off[g] = 0; mrow_cnt[g] = 0; return; }
    atomicAdd(&cnt->n_multi, 1u);
    int w = 0;
    for (uint32_t s = s0; s < s1; ++s) {
        const uint32_t r = (uint32_t)skeys[s];
        for (int k = 0; k < g_cnt[r]; ++k) {
            const int nd = g_node[row_off[r] + k]; const uint8_t fl = g_flag[row_off[r] + k];
            int j = w;                                       // stable insertion
            while (j > 0 && g_node[base + j - 1] > nd) { g_node[base + j] = g_node[base + j - 1]; g_flag[base + j] = g_flag[base + j - 1]; --j; }
            g_node[base + j] = nd; g_flag[base + j] = fl; ++w;
        }
    }
    mrow_off[g] = (uint32_t)base; mrow_cnt[g] = total;
}

// wave per merged row: sort keys (node | name rank | index in row) -> value = slot of the entry
__global__ __launch_bounds__(256) void k_node_keys(const LpsCounters *cnt, const uint32_t *mrow_off, const int32_t *mrow_cnt,
                                                   const uint32_t *koff, const int32_t *g_node, int m_bits, int a_bits,
                                                   unsigned long long *keys, uint32_t *vals, unsigned long long n_keys, LpsCounters *cntw) {
    const unsigned g = blockIdx.x * 4 + (threadIdx.x >> 6); const int l = lane_id();
    if (g >= cnt->n_groups) return;
    const int n = mrow_cnt[g];
    if (n > (1 << a_bits)) { if (l == 0) atomicOr(&cntw->err, (unsigned)LPS_ERR_KEY_RANGE); return; }
    const uint32_t off = mrow_off[g], ko = koff[g];
    for (int a = l; a < n; a += 64) {
        if ((unsigned long long)ko + a >= n_keys) continue;
        keys[ko + a] = ((unsigned long long)(unsigned)g_node[off + a] << (m_bits + a_bits)) | ((unsigned long long)g << a_bits) | (unsigned)a;
        vals[ko + a] = off + a;
    }
}

__global__ void k_node_offsets(const unsigned long long *skeys, unsigned long long n_keys, int shift, uint32_t *node_off,
                               uint32_t *node_end) {
    const unsigned long long s = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_keys) return;
    const unsigned long long k = skeys[s];
    if (k == ~0ull) return;
    const uint32_t nd = (uint32_t)(k >> shift);
    if (s == 0 || (uint32_t)(skeys[s - 1] >> shift) != nd) node_off[nd] = (uint32_t)s;
    if (s + 1 == n_keys || skeys[s + 1] == ~0ull || (uint32_t)(skeys[s + 1] >> shift) != nd) node_end[nd] = (uint32_t)s + 1;
}

// ================================================================================================ edges
__device__ __forceinline__ float edge_upd(float x, bool hi, double w) {
    return hi ? x + 1.0f : (float)((double)x + w);           // SubEdge::addSubEdge (:40-43,62-65)
}

// wave per source node i.  Lane k (< A) owns the four cells (rr,ra,ar,aa) towards node i+1+k in registers.
// For each read observing node i (in name-rank order) lane t loads the read's t-th following observation;
// its node distance d selects the owning lane, the (allele pair, quality class) travels there by ds_permute.
__global__ __launch_bounds__(256) void k_edges(const LpsCounters *cnt, const uint32_t *node_off, const uint32_t *node_end,
                                               const unsigned long long *skeys, const uint32_t *svals,
                                               const uint32_t *mrow_off, const int32_t *mrow_cnt, int m_bits, int a_bits,
                                               const int32_t *g_node, const uint8_t *g_flag, int A, double edge_weight,
                                               double edge_threshold, float *edge, uint8_t *einfo, LpsCounters *cntw) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), l = lane_id();
    if (i >= (int)cnt->n_nodes) return;
    const uint32_t off = node_off[i], end = node_end[i];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    unsigned long long pairs = 0;
    const unsigned long long m_mask = (1ull << m_bits) - 1ull;
    for (uint32_t e0 = off; e0 < end; e0 += 64) {
        const int nb = (int)min(64u, end - e0);
        uint32_t my_val = 0, my_end = 0;
        if (l < nb) {
            const unsigned long long key = skeys[e0 + l];
            const uint32_t m = (uint32_t)((key >> a_bits) & m_mask);
            my_val = svals[e0 + l]; my_end = mrow_off[m] + (uint32_t)mrow_cnt[m];
        }
        for (int t = 0; t < nb; ++t) {
            const uint32_t idx = __shfl(my_val, t), rend = __shfl(my_end, t);
            const int sf = g_flag[idx];                       // wave-uniform address
            const uint32_t e2 = idx + 1 + l;
            const bool in_row = l < A && e2 < rend;
            const int n2 = in_row ? g_node[e2] : -1;
            const int f2 = in_row ? g_flag[e2] : 0;
            const int d = n2 - i;
            const bool ok = in_row && d >= 1 && d <= A;
            pairs += __popcll(__ballot(in_row));
            const int dprev = __shfl_up(d, 1);
            const bool okprev = __shfl_up((int)ok, 1) != 0;
            const bool dup = ok && l > 0 && okprev && dprev == d;
            const int cell = ((sf & 1) << 1) | (f2 & 1);
            const bool hi = (sf & 2) && (f2 & 2);
            const int payload = ok ? (1 | (cell << 1) | ((int)hi << 3)) : 0;
            if (__ballot(dup) == 0) {
                // lanes without a contribution push to lane 63, which owns no target (A <= 63)
                const int recv = __builtin_amdgcn_ds_permute((ok ? (d - 1) : 63) << 2, payload);
                if (l < A && (recv & 1)) {
                    const int c = (recv >> 1) & 3; const bool h = (recv >> 3) & 1;
                    const float x = c == 0 ? a0 : (c == 1 ? a1 : (c == 2 ? a2 : a3));
                    const float nx = edge_upd(x, h, edge_weight);
                    a0 = c == 0 ? nx : a0; a1 = c == 1 ? nx : a1; a2 = c == 2 ? nx : a2; a3 = c == 3 ? nx : a3;
                }
            } else {
                // the same node twice inside the window (overlapping alignments of one read): apply in window order
                for (int tt = 0; tt < A; ++tt) {
                    const int pd = __shfl(d, tt), pp = __shfl(payload, tt);
                    if ((pp & 1) && l == pd - 1) {
                        const int c = (pp >> 1) & 3; const bool h = (pp >> 3) & 1;
                        const float x = c == 0 ? a0 : (c == 1 ? a1 : (c == 2 ? a2 : a3));
                        const float nx = edge_upd(x, h, edge_weight);
                        a0 = c == 0 ? nx : a0; a1 = c == 1 ? nx : a1; a2 = c == 2 ? nx : a2; a3 = c == 3 ? nx : a3;
                    }
                }
            }
        }
    }
    if (l == 0 && pairs) atomicAdd(&cntw->n_pairs, pairs);
    if (l < A) {
        reinterpret_cast<float4 *>(edge)[(size_t)i * A + l] = make_float4(a0, a1, a2, a3);
        // findBestEdgePair (:166-228): everything that does not depend on the scan state
        const float rr = a0, ra = a1, ar = a2, aa = a3;
        const float para = rr + aa, cross = ra + ar;
        const double esr = (double)fminf(para, cross) / (double)fmaxf(para, cross);
        int dir = 0;
        if (para > cross) dir = 1; else if (para < cross) dir = 2;
        if (esr > edge_threshold) dir = 0;
        const bool w20 = (esr <= 0.1 && (rr + aa + ra + ar) >= 1) || (para < 1 && cross >= 1) || (para >= 1 && cross < 1);
        const bool single = (para + cross) <= 1;
        const bool lowesr = esr < 0.2;
        einfo[(size_t)i * A + l] = (uint8_t)(dir | (w20 ? EI_W20 : 0) | (single ? EI_SINGLE : 0) | (lowesr ? EI_LOWESR : 0));
    }
}

// ================================================================================================ vote scan
// ONE wavefront walks the nodes in position order (the reference's loop is a genuine serial dependence chain:
// a node's haplotype depends on the votes of the <=A nodes before it).  Lane (n & 63) owns the vote accumulators
// of node n; per node the owner lane decides, the decision is broadcast with v_readlane, and the A lanes of the
// next A nodes add their votes.  Edge info (1 byte per pair) streams through LDS tiles of 64 nodes.
#define SCAN_TILE 64
__global__ __launch_bounds__(64) void k_vote_scan(const LpsCounters *cnt, const int32_t *nodes, const int32_t *vpos,
                                                  const uint8_t *ntype, const uint8_t *einfo, int A, int distance,
                                                  int8_t *hp_out, int32_t *block_out) {
    __shared__ uint8_t s_info[2][SCAN_TILE * LPS_MAX_ADJACENT];
    const int l = lane_id();
    const int N = (int)cnt->n_nodes;
    float h1 = 0.f, h2 = 0.f, o1 = 0.f, o2 = 0.f; int vc = 0;
    int block_start = -1, last_connect = -1;
    const int tile_bytes = SCAN_TILE * A;
    constexpr int PRE = (SCAN_TILE * LPS_MAX_ADJACENT + 255) / 256;   // u32 words per lane covering one tile
    uint32_t pre[PRE];
    // global -> registers (issued at the start of a tile, consumed at its end: the latency hides under the tile)
    auto fetch_tile = [&](int t0) {
        const long long base = (long long)t0 * A, lim = (long long)N * A;
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
            const int b = q * 256 + l * 4;
            uint32_t w = 0;
            if (b < tile_bytes) {
                if (base + b + 3 < lim) w = *reinterpret_cast<const uint32_t *>(einfo + base + b);   // 4-byte aligned: 64*A*t0
                else for (int k = 0; k < 4; ++k) if (base + b + k < lim) w |= (uint32_t)einfo[base + b + k] << (8 * k);
            }
            pre[q] = w;
        }
    };
    auto stash_tile = [&](uint8_t *dst) {
#pragma unroll
        for (int q = 0; q < PRE; ++q) { const int b = q * 256 + l * 4; if (b < tile_bytes) *reinterpret_cast<uint32_t *>(dst + b) = pre[q]; }
    };
    if (l == 0 && N > 0) { hp_out[N - 1] = 0; block_out[N - 1] = -1; }   // the last node is never processed (:308-311)
    if (N > 0) { fetch_tile(0); stash_tile(s_info[0]); }
    wave_sync();
    for (int t0 = 0, buf = 0; t0 < N; t0 += SCAN_TILE, buf ^= 1) {
        const bool more = t0 + SCAN_TILE < N;
        if (more) fetch_tile(t0 + SCAN_TILE);
        // per-tile node attributes, one node per lane
        const int n_me = t0 + l;
        int my_pos = 0, my_next = 0, my_type = 0;
        if (n_me < N) { my_pos = vpos[nodes[n_me]]; my_type = ntype[n_me]; }
        if (n_me + 1 < N) my_next = vpos[nodes[n_me + 1]];
        const int my_gap = (n_me + 1 < N) ? (abs(my_next - my_pos) > distance) : 1;
        const unsigned long long gapmask = __ballot(my_gap != 0);
        const uint8_t *info = s_info[buf];
        const int tend = min(SCAN_TILE, N - 1 - t0);
        for (int j = 0; j < tend; ++j) {
            const int i = t0 + j, s = i & 63;
            const bool use_sp = (vc > 3) && !(o1 == 0.f && o2 == 0.f);                    // Onelongcase (:276)
            const float c1 = use_sp ? o1 : h1, c2 = use_sp ? o2 : h2;
            const int code = (c1 == c2) ? 0 : (c1 > c2 ? 1 : 2);
            const int code_s = __builtin_amdgcn_readlane(code, s);
            const int typ = __builtin_amdgcn_readlane(my_type, j);
            const bool gap = (gapmask >> j) & 1ull;
            const bool skip = gap || (code_s == 0 && i < last_connect);                    // :318,:340
            if (!skip && code_s == 0) block_start = i;
            const int hp_i = skip ? 0 : (code_s == 0 ? 1 : code_s);
            if (l == s) { hp_out[i] = (int8_t)hp_i; block_out[i] = skip ? -1 : block_start; h1 = h2 = o1 = o2 = 0.f; vc = 0; }
            const int k = (l - s - 1) & 63;
            const bool act = !skip && k < A && (i + 1 + k) < N;
            const int inf = act ? info[j * A + k] : 0;
            const int dir = inf & 3;
            const bool conn = act && dir != 0;
            const bool th1 = (hp_i == 1) == (dir == 1);
            const float w = (typ == 4) ? 0.1f : ((inf & EI_W20) ? 20.f : 1.f);           // :216,:367
            if (conn) {
                if (th1) h1 += w; else h2 += w;
                if (inf & EI_SINGLE) vc++;
                else if ((inf & EI_LOWESR) && w >= 1.f && typ != 3) { if (th1) o1 += w; else o2 += w; }
            }
            const unsigned long long cm = __ballot(conn);
            if (cm) {
                const int sh = (s + 1) & 63;
                const unsigned long long rot = sh ? ((cm >> sh) | (cm << (64 - sh))) : cm;
                last_connect = max(last_connect, i + 1 + (63 - __clzll(rot)));
            }
        }
        if (more) stash_tile(s_info[buf ^ 1]);
        wave_sync();
    }
}

// ================================================================================================ read correction
__global__ void k_block_size(const LpsCounters *cnt, const int32_t *block, uint32_t *bsize) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int)cnt->n_nodes) return;
    if (block[i] >= 0) atomicAdd(&bsize[block[i]], 1u);
}

// thread per alignment: readCorrection's per-read vote (:904-959).  Sequential in the read so that the 0.1
// contributions of indel sites are summed in the reference's order (doubles).
__global__ void k_read_correction(int n_reads, const uint32_t *row_off, const int32_t *g_cnt, const int32_t *g_node,
                                  const uint8_t *g_flag, const int32_t *block, const uint32_t *bsize, const int8_t *hp,
                                  const uint8_t *ntype, double read_confidence, uint32_t *cnt4) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int n = g_cnt[r];
    if (n <= 0) return;
    const uint32_t off = row_off[r];
    double rc = 0, ac = 0;
    for (int k = 0; k < n; ++k) {
        const int nd = g_node[off + k]; const int al = g_flag[off + k] & 1;
        const int b = block[nd];
        if (b >= 0 && bsize[b] > 1) {
            const int refhap = (hp[nd] == 1) ? 0 : 1;
            const int h = al == 0 ? refhap : 1 - refhap;
            const int ty = ntype[nd];
            if (ty == 0 || ty == 1) { if (h == 0) rc++; else ac++; }
            else if (ty == 3 || ty == 4) { if (h == 0) rc += 0.1; else ac += 0.1; }
        }
    }
    if (fmax(rc, ac) / (rc + ac) > read_confidence && (rc + ac) > 1) {
        const int bh = (rc > ac) ? 0 : 1;
        for (int k = 0; k < n; ++k) atomicAdd(&cnt4[(size_t)g_node[off + k] * 4 + bh * 2 + (g_flag[off + k] & 1)], 1u);
    }
}

__global__ void k_final(const LpsCounters *cnt, const int32_t *nodes, const int32_t *vpos, const int32_t *block,
                        const uint32_t *bsize, const uint32_t *cnt4, double snp_confidence, int32_t *out_ps, uint8_t *out_gt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int)cnt->n_nodes) return;
    const int b = block[i];
    if (b < 0 || bsize[b] <= 1) return;
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
    return std::max(a, std::max(b, c)) + 256;
}

void sort_keys64(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, size_t n, int bits,
                 hipStream_t s) {
    if (n == 0) return;
    HIP_TRY(rocprim::radix_sort_keys(temp, temp_bytes, in, out, n, 0, bits, s));
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

void launch_cnv_filter(const LpsCounters *cnt, int n_reads, int n_var, const uint32_t *row_off, const int32_t *row_cnt,
                       const uint8_t *deleted, int32_t *obs_var, const uint16_t *obs_aq, const int32_t *vpos,
                       const int32_t *cnv_start, const int32_t *cnv_end, long long *agg_sum, int32_t *agg_cnt, double *miss,
                       hipStream_t s) {
    HIP_TRY(hipMemsetAsync(agg_sum, 0, (size_t)n_var * 2 * sizeof(long long), s));
    HIP_TRY(hipMemsetAsync(agg_cnt, 0, (size_t)n_var * 2 * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_cnv_filter_serial, dim3(1), dim3(64), 0, s, cnt, n_reads, row_off, row_cnt, deleted, obs_var, obs_aq, vpos, cnv_start, cnv_end, agg_sum, agg_cnt, miss, n_var);
}

void launch_clip_cnv(const ClipView &C, const int32_t *row_fail, unsigned n_clips, unsigned long long *keys,
                     unsigned long long *keys_sorted, void *temp, size_t temp_bytes, int32_t *cnv_start, int32_t *cnv_end,
                     LpsCounters *cnt, hipStream_t s) {
    if (n_clips) {
        hipLaunchKernelGGL(k_clip_keys, GRID(n_clips, 256), 0, s, C, row_fail, n_clips, keys);
        sort_keys64(temp, temp_bytes, keys, keys_sorted, n_clips, 64, s);
    }
    hipLaunchKernelGGL(k_cnv_state, dim3(1), dim3(64), 0, s, keys_sorted, n_clips, cnv_start, cnv_end, cnt);
}

void launch_name_keys(int n_reads, const uint32_t *name_id, const int32_t *row_cnt, unsigned long long *keys,
                      LpsCounters *cnt, hipStream_t s) {
    hipLaunchKernelGGL(k_name_keys, GRID(n_reads, 256), 0, s, n_reads, name_id, row_cnt, keys, cnt);
}

void launch_groups(const unsigned long long *skeys, int n_reads, LpsCounters *cnt, uint32_t *head, uint32_t *gidx,
                   uint32_t *gstart, uint32_t *read_group, void *temp, size_t temp_bytes, hipStream_t s) {
    hipLaunchKernelGGL(k_group_heads, GRID(n_reads, 256), 0, s, skeys, n_reads, cnt, head);
    exscan_u32(temp, temp_bytes, head, gidx, n_reads, s);
    hipLaunchKernelGGL(k_group_starts, GRID(n_reads, 256), 0, s, skeys, head, gidx, n_reads, cnt, gstart, read_group);
}

void launch_overlap_filter(const unsigned long long *skeys, const uint32_t *gstart, const LpsCounters *cnt, int n_reads,
                           const uint32_t *row_off, const int32_t *row_cnt, const int32_t *obs_var, const int32_t *vpos,
                           double thr, uint32_t *stack, uint8_t *deleted, hipStream_t s) {
    hipLaunchKernelGGL(k_overlap_filter, GRID(n_reads, 128), 0, s, skeys, gstart, cnt, row_off, row_cnt, obs_var, vpos, thr, stack, deleted);
}

void launch_nodes(int n_reads, int n_var, const uint32_t *row_off, const int32_t *row_cnt, const uint8_t *deleted,
                  const int32_t *obs_var, const uint16_t *obs_aq, uint32_t *is_node, uint32_t *vtype_key, uint32_t *node_of,
                  int32_t *nodes, uint8_t *ntype, int base_quality, int32_t *g_node, uint8_t *g_flag, int32_t *g_cnt,
                  LpsCounters *cnt, void *temp, size_t temp_bytes, hipStream_t s) {
    hipLaunchKernelGGL(k_mark_nodes, dim3((n_reads + 3) / 4), dim3(256), 0, s, n_reads, row_off, row_cnt, deleted, obs_var, obs_aq, is_node, vtype_key);
    exscan_u32(temp, temp_bytes, is_node, node_of, n_var, s);
    hipLaunchKernelGGL(k_node_list, GRID(n_var, 256), 0, s, n_var, is_node, node_of, vtype_key, nodes, ntype, cnt);
    hipLaunchKernelGGL(k_graph_obs, dim3((n_reads + 3) / 4), dim3(256), 0, s, n_reads, row_off, row_cnt, deleted, obs_var, obs_aq, node_of, base_quality, g_node, g_flag, g_cnt, cnt);
}

void launch_merge_rows(const unsigned long long *skeys, const uint32_t *gstart, LpsCounters *cnt, int n_reads,
                       const uint32_t *row_off, const int32_t *g_cnt, int32_t *g_node, uint8_t *g_flag,
                       unsigned long long capacity, uint32_t *mrow_off, int32_t *mrow_cnt, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_rows, GRID(n_reads, 128), 0, s, skeys, gstart, cnt, row_off, g_cnt, g_node, g_flag, capacity, mrow_off, mrow_cnt);
}

void launch_node_lists(LpsCounters *cnt, int n_reads, int n_var, const uint32_t *mrow_off, const int32_t *mrow_cnt, uint32_t *koff,
                       const int32_t *g_node, int m_bits, int a_bits, int n_bits, unsigned long long *keys,
                       unsigned long long *keys_sorted, uint32_t *vals, uint32_t *vals_sorted, unsigned long long n_keys,
                       uint32_t *node_off, uint32_t *node_end, void *temp, size_t temp_bytes, hipStream_t s) {
    // koff = exclusive scan of mrow_cnt over groups (unused groups have mrow_cnt = 0 by memset)
    exscan_u32(temp, temp_bytes, reinterpret_cast<const uint32_t *>(mrow_cnt), koff, n_reads, s);
    HIP_TRY(hipMemsetAsync(keys, 0xff, n_keys * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_node_keys, dim3((n_reads + 3) / 4), dim3(256), 0, s, cnt, mrow_off, mrow_cnt, koff, g_node, m_bits, a_bits, keys, vals, n_keys, cnt);
    sort_pairs64(temp, temp_bytes, keys, keys_sorted, vals, vals_sorted, n_keys, n_bits + m_bits + a_bits, s);
    if (n_keys) hipLaunchKernelGGL(k_node_offsets, GRID(n_keys, 256), 0, s, keys_sorted, n_keys, m_bits + a_bits, node_off, node_end);
    (void)n_var;
}

void launch_edges(LpsCounters *cnt, int n_var, const uint32_t *node_off, const uint32_t *node_end,
                  const unsigned long long *skeys, const uint32_t *svals, const uint32_t *mrow_off, const int32_t *mrow_cnt,
                  int m_bits, int a_bits, const int32_t *g_node, const uint8_t *g_flag, int A, double edge_weight,
                  double edge_threshold, float *edge, uint8_t *einfo, hipStream_t s) {
    hipLaunchKernelGGL(k_edges, dim3((n_var + 3) / 4), dim3(256), 0, s, cnt, node_off, node_end, skeys, svals, mrow_off, mrow_cnt, m_bits, a_bits, g_node, g_flag, A, edge_weight, edge_threshold, edge, einfo, cnt);
}

void launch_vote_scan(const LpsCounters *cnt, const int32_t *nodes, const int32_t *vpos, const uint8_t *ntype,
                      const uint8_t *einfo, int A, int distance, int8_t *hp, int32_t *block, hipStream_t s) {
    hipLaunchKernelGGL(k_vote_scan, dim3(1), dim3(64), 0, s, cnt, nodes, vpos, ntype, einfo, A, distance, hp, block);
}

void launch_correction(const LpsCounters *cnt, int n_reads, int n_var, const uint32_t *row_off, const int32_t *g_cnt,
                       const int32_t *g_node, const uint8_t *g_flag, const int32_t *nodes, const int32_t *vpos,
                       const int32_t *block, uint32_t *bsize, const int8_t *hp, const uint8_t *ntype, double read_conf,
                       double snp_conf, uint32_t *cnt4, int32_t *out_ps, uint8_t *out_gt, hipStream_t s) {
    hipLaunchKernelGGL(k_block_size, GRID(n_var, 256), 0, s, cnt, block, bsize);
    hipLaunchKernelGGL(k_read_correction, GRID(n_reads, 128), 0, s, n_reads, row_off, g_cnt, g_node, g_flag, block, bsize, hp, ntype, read_conf, cnt4);
    hipLaunchKernelGGL(k_final, GRID(n_var, 256), 0, s, cnt, nodes, vpos, block, bsize, cnt4, snp_conf, out_ps, out_gt);
}
