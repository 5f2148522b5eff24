// lps_extra.hip — SV and MOD rows of `phase --sv-file / --mod-file` (gfx950).
//
//   BamParser::get_snp, SV branch   src/phase/ParsingBam.cpp:1397-1434
//   BamParser::get_snp, MOD branch  src/phase/ParsingBam.cpp:1373-1395
//
// The reference walks three cursors (SNP map, SV vector, MOD map) through every CIGAR operation and always serves the smallest position; a
// pending SNP at an operation that is not a match `break`s the inner loop.  What that does to a SV / MOD row at position p, in closed form:
//   * SNP rows are served exactly as without the other two tables (k_extract_phase is untouched);
//   * row p is served by the FIRST operation j with   E_j > p   and   (j is a match  or  no SNP lies in [ref_pos_j, p]),
//     E_j = ref_pos_j + oplen_j for EVERY operation code - insertions, clips and pads reach forward over positions they do not consume;
//     rows before the alignment start are skipped by the reference's "first" cursors (SV: start - 1 still counts, its cursor compares 1-based);
//   * a MOD row is recorded when the alignment's name is listed with the alignment's strand - and, a reference quirk kept here, only if the SNP
//     cursor is not at its end at that moment (the comparison `modPos < variantPos` then reads past the map: libstdc++ finds the entry count);
//   * a SV row is ALT when an I / D of about its length lies within svWindow operations of j.
// tests/test_extra_gpu.py holds this against the oracle's literal three-cursor walk, which in turn is pinned to the reference binary.
//
// k_extra_merge, one wavefront per alignment: stages the CIGAR like the extraction (512 operations per round), resolves the rows of the
// alignment's reach against it, and - only if any row was recorded - moves the alignment's row of observations to fresh arena slots with the
// new records merged in by position (merge path, 64 outputs per round).  Every observation leaves with its index in the position-sorted UNION
// of the three tables: the graph stages never ask which file a row came from (the reference's maps are keyed by position).
#include "lps_kernels.h"

#define XM_CAP 512      // recorded SV / MOD rows of one alignment kept in LDS; more (a 100-kb read over methylation calls every 200 bp) -> second walk

__device__ __forceinline__ int wave_incl_max(int v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if (l >= d) v = max(v, o); }
    return v;
}

// position of the last SNP row before p (INT_MIN: none); one lane
__device__ __forceinline__ int last_snp_before(const VarView &V, int p) {
    int lo = 0, hi = V.n;
    if (p >= 0) {
        const int b = p >> LPS_BUCKET_SHIFT;
        if (b < V.n_bucket) { lo = V.bucket[b]; hi = V.bucket[b + 1]; } else lo = V.bucket[V.n_bucket];
    } else hi = 0;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (V.pos[mid] < p) lo = mid + 1; else hi = mid; }
    return lo > 0 ? V.pos[lo - 1] : (int)0x80000000;
}

#define XM_WPB 1        // waves per workgroup (they share nothing)
// (since round 3 the general walker behind k_extra_find: it takes the alignments that kernel queued - more chunks than its table holds, an op of 2^24
// bases, a second reservation that did not fit - and finds their rows of observations already in union indices, as the extraction writes them)
__device__ void extra_merge_one(const int r, const int arena, const VarView &V, const ReadView &R, const ObsView &O, const ExtraView &X, int mapping_quality, LpsCounters *cnt,
                                int *sref, int *sqry, uint32_t *scig, int *spm, ObsRec *sex) {
    const int l = lane_id();
    const RowDesc rd = O.rows[r];
    const int start = R.ref_start[r], flag = R.flag[r];
    const bool live = !(R.mapq[r] < mapping_quality || (flag & 0x4) || (flag & 0x100) || (flag & 0x400) || start >= V.last_pos) && rd.fail == 0x7fffffff;
    if (!live) return;                                                  // filtered (:1282-1291) or get_snp returned early: the alignment has no row
    const int n_cig = R.cp_n[r];
    const uint32_t *cig = R.cig(r);
    const uint32_t name = R.name_id[r];
    const bool rev = (flag & 0x10) != 0;

    // first row of the alignment's cursors: SV rows from start - 1 on (the SV cursor compares the 1-based VCF position), MOD rows from start on
    int xs = wave_lower_bound(X.pos, 0, X.n, start - 1);
    if (xs < X.n && X.pos[xs] == start - 1 && X.kind[xs] == 2) ++xs;

    // how far the alignment can reach at most: E_j <= start + (reference bases consumed by all operations) + (longest operation).  One pass over
    // the CIGAR words with no staging; with sparse rows (SVs) most alignments have none below that bound and are done after it.  When the next
    // row lies within a quarter of the read length the alignment almost surely reaches it: no bound is taken, the walk below ends by itself.
    int xe = xs < X.n ? X.n : xs;
    if (xs < X.n && X.pos[xs] - start >= R.l_qseq[r] / 4) {
        long long cons = 0; int longest = 0;
        for (int i0 = 0; i0 < n_cig; i0 += LPS_SEG) {
            uint32_t wv[8];
            load_ops8(cig + i0, 8 * l, min(LPS_SEG, n_cig - i0), wv);                 // 6u (no length) past the end
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int len = (int)(wv[k] >> 4); cons += len & bit_mask(op_consume_bits(wv[k] & 15u), 0); longest = max(longest, len); }
        }
        cons = wave_sum(cons); longest = wave_max(longest);
        const long long bound = (long long)start + cons + longest;
        if (bound < 0x7fffffffll) xe = wave_lower_bound(X.pos, xs, X.n, (int)bound);
    }

    int n_emit = 0;
    uint32_t new_off = 0;
    if (xs < xe) {
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            int xp = xs, ref_pos = start, q_pos = 0, idx = 0;
#pragma unroll 1
            for (int i0 = 0; i0 < n_cig && xp < xe; i0 += LPS_SEG) {
                const int nseg = min(LPS_SEG, n_cig - i0);
                uint32_t wv[8]; int my_ref;
                load_ops8(cig + i0, 8 * l, nseg, wv);
                wave_sync();                                            // the previous round's readers are done with the LDS arrays
                (void)stage_ops8(wv, l, ref_pos, q_pos, sref, sqry, scig, my_ref);
                // prefix maximum of E over the segment's operations (padding past the CIGAR: E = 0x80000000, never a candidate)
                int e[8], rr = my_ref, mx = (int)0x80000000;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned op = wv[k] & 15u; const int len = (int)(wv[k] >> 4);
                    const int ek = (8 * l + k < nseg) ? rr + len : (int)0x80000000;
                    mx = max(mx, ek); e[k] = mx;
                    rr += len & bit_mask(op_consume_bits(op), 0);
                }
                const int inc = wave_incl_max(mx);
                int before = __shfl_up(inc, 1); if (l == 0) before = (int)0x80000000;
#pragma unroll
                for (int k = 0; k < 8; ++k) spm[8 * l + k] = max(before, e[k]);
                wave_sync();
                const int segmax = __shfl(inc, 63);
                // rows this segment can serve, 64 at a time
#pragma unroll 1
                while (xp < xe) {
                    const int row = xp + l;
                    const int p = row < xe ? X.pos[row] : 0x7fffffff;
                    const bool cand = p < segmax;
                    const int nb = __popcll(__ballot(cand));            // positions are sorted: the candidates are the first nb lanes
                    if (nb == 0) break;
                    bool found = false; int j = 0, rp = 0;
                    if (cand) {
                        int lo = 0, hi = nseg - 1;                      // spm[nseg-1] = segmax > p: the search ends inside the segment
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if (spm[mid] > p) hi = mid; else lo = mid + 1; }
                        j = lo;
                        int q = 0; bool have_q = false;
                        for (; j < nseg; ++j) {
                            const uint32_t c = scig[j]; const int op = c & 15u, len = (int)(c >> 4); rp = sref[j];
                            if (rp + len <= p) continue;
                            if (op_is_match(op)) { found = true; break; }
                            if (!have_q) { q = last_snp_before(V, p + 1); have_q = true; }      // SNP positions never equal p (lps_set_extra_variants)
                            if (rp > q) { found = true; break; }
                        }
                    }
                    const int nres = __popcll(__ballot(found));         // the served rows are a prefix too (the serving operation is monotone in p)
                    // ---- what the served rows record
                    bool emit = false; ObsRec rec{0, 0};
                    if (found) {
                        const int kind = X.kind[row], info = X.info[row];
                        if (kind == 1) {                                // :1403-1429
                            const int ja = i0 + j;
                            const double region = (double)(abs(info) + 1);
                            int allele = 0;
                            const int a = max(ja - X.sv_window, 0), b = min(ja + X.sv_window, n_cig);
                            for (int t = a; t < b; ++t) {
                                const uint32_t c = cig[t]; const int op = c & 15u; const double len = (double)(int)(c >> 4);
                                if ((op == 1 || op == 2) && fabs(region - len) / fabs(region) < X.sv_threshold) { allele = 1; break; }
                            }
                            emit = true; rec = ObsRec{X.u[row], (uint32_t)pack_aq(allele, -1)};
                        } else {                                        // :1377-1392
                            uint32_t lo = X.mod_off[info], hi = X.mod_off[info + 1]; const uint32_t end = hi;
                            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (X.mod_name[mid] < name) lo = mid + 1; else hi = mid; }
                            if (lo < end && X.mod_name[lo] == name) {
                                const unsigned f = X.mod_flag[lo];
                                // the reference compares modPos with *currentVariantIter even when that is end(): the entry count of the SNP map
                                const bool cursor_ok = V.last_pos >= max(rp, p + 1) || p < V.n;
                                if ((((f >> 1) & 1u) != 0) == rev && cursor_ok) { emit = true; rec = ObsRec{X.u[row], (uint32_t)pack_aq((f & 1u) ? 0 : 1, rev ? -3 : -2)}; }
                            }
                        }
                    }
                    const unsigned long long em = __ballot(emit);
                    if (emit) {
                        const int at = idx + __popcll(em & lanemask_lt());
                        if (pass == 0) { if (at < XM_CAP) sex[at] = rec; }
                        else O.rec[(size_t)new_off + rd.cnt + at] = rec;
                    }
                    idx += __popcll(em);
                    xp += nres;
                    if (nres < 64) break;                               // the rest waits for a later operation (or lies beyond this segment's reach)
                }
            }
            if (pass == 0) {
                n_emit = idx;
                if (n_emit == 0) break;
                const unsigned long long need = (unsigned long long)rd.cnt + (unsigned long long)n_emit;
                unsigned long long local = 0;
                if (l == 0) local = atomicAdd(&O.arena_ctr[arena * 8], need);
                local = __shfl(local, 0);
                if (local + need > O.arena_size) { if (l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_OBS_OVERFLOW); return; }   // the host grows the arenas and runs again
                new_off = (uint32_t)((unsigned long long)arena * O.arena_size + local);
                if (n_emit <= XM_CAP) {
                    wave_sync();
                    for (int k = l; k < n_emit; k += 64) O.rec[(size_t)new_off + rd.cnt + k] = sex[k];
                    break;
                }
            }
        }
    }
    if (n_emit == 0) return;                                            // nothing recorded: the row stays where it is (in union indices already)
    // ---- merge path: A = the alignment's SNP / indel observations (old row), B = the recorded rows (tail of the new row), both ascending and
    //      without common keys.  Output o comes from A[i] or B[o - i]; unread B entries always lie at or behind the outputs being written.
    __threadfence_block();
    const ObsRec *A = O.rec + rd.off; ObsRec *out = O.rec + new_off; const ObsRec *B = out + rd.cnt;
    const int nA = rd.cnt, nB = n_emit, total = nA + nB;
#pragma unroll 1
    for (int o0 = 0; o0 < total; o0 += 64) {
        const int o = o0 + l;
        ObsRec v{0, 0};
        if (o < total) {
            int lo = max(0, o - nB), hi = min(o, nA);
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (A[mid].var < B[o - mid - 1].var) lo = mid + 1; else hi = mid; }
            const int i = lo, j = o - lo;
            ObsRec a{0x7fffffff, 0}, b{0x7fffffff, 0};
            if (i < nA) a = A[i];
            if (j < nB) b = B[j];
            v = (a.var < b.var) ? a : b;
        }
        wave_sync();                                                    // every lane holds its value before anything of this round is written
        if (o < total) out[o] = v;
        wave_sync();
    }
    if (l == 0) { RowDesc d = rd; d.off = new_off; d.cnt = total; d.flags = 0; O.rows[r] = d; }
}
__global__ __launch_bounds__(64) void k_extra_merge(VarView V, ReadView R, ObsView O, ExtraView X, const uint32_t *list, const unsigned *n_list, int mapping_quality, LpsCounters *cnt) {
    __shared__ __attribute__((aligned(16))) int s_ref[LPS_SEG];
    __shared__ __attribute__((aligned(16))) int s_qry[LPS_SEG];
    __shared__ __attribute__((aligned(16))) uint32_t s_cig[LPS_SEG + 4];
    __shared__ int s_pm[LPS_SEG];
    __shared__ ObsRec s_ex[XM_CAP];
    const unsigned n = *n_list;
#pragma unroll 1
    for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
        extra_merge_one((int)list[i], (int)(i % (unsigned)O.n_arenas), V, R, O, X, mapping_quality, cnt, s_ref, s_qry, s_cig, s_pm, s_ex);
        wave_sync();                                                    // the LDS arrays are reused by the wave's next alignment
    }
}


// ================================================================================================ the stream walk for SV / MOD rows
// k_extra_merge above takes an alignment per wave and stages every op; with dense MOD rows (a row every 2 kb: every alignment holds ten) that is a
// second, slow walk over every CIGAR.  The same rows are found by the extraction's design: a wave takes a JOB of four alignments, walks their
// lane-chunks as one stream (lps_reads.hip) and keeps per chunk (reference coordinate at its start, running maximum of E = ref_pos + length over
// the alignment's ops through the chunk) - the rows of the four alignments' reaches are then one flattened list, a lane each: binary search of the
// chunk whose running maximum first exceeds the row, its 8 words, the first op with E > p, the scan forward to the op that serves the row (header
// of this file; chunks whose own ops all end at or before the row are passed over by a third table entry), the record.  The records of an alignment
// leave compacted in row order into the wave's arena and are merged by position into the alignment's row right there: ONE more reservation for the
// rows of the job that got records, the outputs of its four merges flattened over the lanes, the keys of rows and records in LDS.  What this walk
// cannot take (an alignment of more chunks than the table holds, an op of 2^24 bases, stream coordinates beyond 2^30) is queued for k_extra_merge.
#ifndef XF_TAB
#define XF_TAB 1024    // lane-chunks of a group's stream (8 192 words); 12.3 KB of LDS per wave (768: more alignments of the test genome left to the general walker, 70 us)
#endif

__global__ void k_read_x0(ExtraView X, const int32_t *ref_start, int n, int32_t *x0) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int key = ref_start[r] - 1;                                   // SV rows from start - 1 on (the SV cursor compares the 1-based VCF position), MOD rows from start on
    int lo = 0, hi = X.n;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (X.pos[m] < key) lo = m + 1; else hi = m; }
    if (lo < X.n && X.pos[lo] == key && X.kind[lo] == 2) ++lo;
    x0[r] = lo;
}

__global__ __launch_bounds__(64, 4) void k_extra_find(VarView V, ReadView R, ObsView O, ExtraView X, const int32_t *x0, uint32_t *redo, unsigned *n_redo, int mapping_quality, LpsCounters *cnt) {
    __shared__ __attribute__((aligned(16))) int2 s_tab[XF_TAB];          // (stream reference coordinate at the chunk's start, running maximum of E through the chunk)
    __shared__ int s_lm[XF_TAB];                                         // maximum of E over the chunk's own ops: chunks that cannot serve a row are passed over without loading them
    __shared__ int s_bm[XF_TAB / 16];                                    // ... and the maximum over 16 chunks: sixteen are passed over at a time
    __shared__ ExtHdr s_hdr[4];
    const int l = lane_id();
    const int job = blockIdx.x, r0 = job * 4;
    if (r0 >= R.n) return;
    const int nq = min(4, R.n - r0);
    const int arena = blockIdx.x % O.n_arenas;
    const unsigned long long arena_lo = (unsigned long long)arena * O.arena_size;
    // ---- plan: alignment q in lane q; the filters of the extraction + "get_snp returned early: the alignment has no row"
    int h_start = 0, h_n = 0, h_x0 = 0, h_flag = 0, h_rcnt = 0; unsigned h_cp = 0, h_name = 0, h_roff = 0; bool h_live = false;
    if (l <= nq) h_cp = R.cp_off[r0 + l];
    if (l < nq) {
        const int r = r0 + l; h_start = R.ref_start[r]; h_n = R.cp_n[r]; h_x0 = x0[r]; h_flag = R.flag[r]; h_name = R.name_id[r];
        const RowDesc rd = O.rows[r]; h_roff = rd.off; h_rcnt = rd.cnt;
        h_live = !(R.mapq[r] < mapping_quality || (h_flag & 0x4) || (h_flag & 0x100) || (h_flag & 0x400) || h_start >= V.last_pos) && rd.fail == 0x7fffffff && h_x0 < X.n;
    }
    const int h_nch = (int)(__shfl_down(h_cp, 1) - h_cp);
    bool general = l < nq && h_live && h_nch > XF_TAB;                  // more chunks than the table holds: left to k_extra_merge
    unsigned todo = (unsigned)__ballot(h_live && !general) & 15u;
#pragma unroll 1
    while (todo) {
        const int qa = __builtin_ctz(todo);
        const unsigned c_lo = __shfl(h_cp, qa);
        int qb = qa; unsigned gm = 1u << qa;
        for (int q = qa + 1; q < nq; ++q) {
            if (__shfl(h_cp, q + 1) - c_lo > (unsigned)XF_TAB) break;     // (alignments in between that are not walked pass by as words)
            if ((todo >> q) & 1u) { gm |= 1u << q; qb = q; }
        }
        todo &= ~gm;
        const bool h_in = l < 4 && ((gm >> l) & 1u);
        const bool h_walk = h_in && h_n > 0;
        const int h_c0 = (l <= nq) ? (int)(h_cp - c_lo) : 0;
        const uint32_t *cg = R.cigp + 8ull * c_lo;
        const int TC = __builtin_amdgcn_readlane(h_c0 + h_nch, qb);
        auto request = [&](int cid, uint32_t (&w)[8]) __attribute__((always_inline)) {
            const uint32_t *p = cg + 8 * min(cid, max(TC - 1, 0));
            const uint4 a = *reinterpret_cast<const uint4 *>(p), b = *reinterpret_cast<const uint4 *>(p + 4);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        };
        // first chunk of an alignment of the job (the running maximum starts again there): chunk index -> bit
        int hc[4]; 
#pragma unroll
        for (int q = 0; q < 4; ++q) hc[q] = q < nq ? __builtin_amdgcn_readlane(h_c0, q) : 0x7fffffff;   // (alignments in front of the stream: negative)
        int xq[4], pp[4]; bool walkq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { xq[q] = __builtin_amdgcn_readlane(h_x0, q); walkq[q] = (__ballot(h_walk) >> q) & 1ull; pp[q] = X.pos[min(xq[q] + l, X.n - 1)]; }
        if (l < 4) {
            ExtHdr &h = s_hdr[l];
            h.crel = 8 * h_c0; h.ncig = h_walk ? h_n : 0; h.c0 = h_c0; h.nch = h_walk ? h_nch : 0;
            h.lq = h_flag; h.soff = h_name;
        }
        // ---- walk: coordinates like stream_round (lps_kernels.h) + the running maximum of E, a segmented inclusive max-scan (heads = first chunks)
        int carry_r = 0, carry_e = (int)0x80000000; uint32_t big = 0; bool absurd = false;
#pragma unroll 1
        for (int R0 = 0; R0 < TC; R0 += 256) {                            // four rounds requested together, like k_extract_phase
            uint32_t wt[4][8];
#pragma unroll
            for (int t = 0; t < 4; ++t) request(R0 + 64 * t + l, wt[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cid = R0 + 64 * t + l; const bool live = cid < TC;
                unsigned rt = 0; int emax = (int)0x80000000; uint32_t bg = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t x = wt[t][k]; const unsigned len = x >> 4;
                    emax = max(emax, (int)(rt + len));                // E of word k relative to the chunk's start: reference consumed before it + its length, whatever its op
                    rt = __umul24(len, op_bit(LPS_RMASK2, x)) + rt; bg |= x;
                }
                rt = live ? rt : 0u; big |= live ? bg : 0u;
                const int ir = wave_incl_scan_dpp((int)rt);
                const int my_s = carry_r + ir - (int)rt;
                // E of alignment q of the job is kept as E + (q << 28): every value of an alignment lies above every value of the one before it, so
                // ONE running maximum over the stream (a DPP scan) is the running maximum of each alignment - no segmented scan
                const int aq = (cid >= hc[1]) + (cid >= hc[2]) + (cid >= hc[3]);       // the alignment of the job this chunk belongs to
                int v = live ? my_s + emax + (aq << 28) : (int)0x80000000;
                if (live) s_lm[cid] = v;
                v = max(wave_incl_max_dpp(v), carry_e);
                if (live) s_tab[cid] = make_int2(my_s, v);
                carry_r += __builtin_amdgcn_readlane(ir, 63); carry_e = __builtin_amdgcn_readlane(v, 63);
            }
            absurd |= (unsigned)carry_r > 0x07ffffffu;                     // (E + (q << 28) must stay a positive int)
            if (absurd) break;
        }
        if (absurd || __ballot(big >= 0x10000000u)) { general |= h_in; continue; }   // outside this walk's arithmetic: k_extra_merge takes the group's alignments
        wave_sync();
        static_assert(XF_TAB / 16 <= 64, "a lane per block of 16 chunks");
        if (l < XF_TAB / 16) { int m = (int)0x80000000; for (int t = 0; t < 16; ++t) { const int c = 16 * l + t; if (c < TC) m = max(m, s_lm[c]); } s_bm[l] = m; }
        wave_sync();
        // ---- reach of each alignment, its rows: [x0, first row at or beyond the reach)
        int b_sat = 0, b_reach = h_start;
        if (h_walk) { const int2 ts = s_tab[h_c0], te = s_tab[h_c0 + h_nch - 1]; b_sat = ts.x; b_reach = h_start + (te.y - (l << 28)) - ts.x; }
        int nrow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int reach = __builtin_amdgcn_readlane(b_reach, q);
            int n = __popcll(__ballot(walkq[q] && xq[q] + l < X.n && pp[q] < reach));
            if (n == 64) { for (;;) { int p2 = 0x7fffffff; if (xq[q] + n + l < X.n) p2 = X.pos[xq[q] + n + l]; const int m = __popcll(__ballot(p2 < reach)); n += m; if (m < 64) break; } }
            nrow[q] = n;
        }
        int cum[5]; cum[0] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) cum[q + 1] = cum[q] + nrow[q];
        const int T = cum[4];
        int xadj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xadj[q] = xq[q] - cum[q];
        if (l < 4) { ExtHdr &h = s_hdr[l]; h.vadj = SEL4(l, xadj); h.ds = b_sat - h_start; }
        int maxnch = l < 4 ? s_hdr[l].nch : 0;
        maxnch = max(max(__builtin_amdgcn_readlane(maxnch, 0), __builtin_amdgcn_readlane(maxnch, 1)), max(__builtin_amdgcn_readlane(maxnch, 2), __builtin_amdgcn_readlane(maxnch, 3)));
        // ONE reservation for the group: a slot per row of the reaches (most rows are recorded), and behind them room for the merged rows of its
        // alignments (their observations + as many records at most).  Its answer is waited for where the first record is stored
        int sumA = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) sumA += walkq[q] ? __builtin_amdgcn_readlane(h_rcnt, q) : 0;
        const unsigned long long want = (unsigned long long)(2 * T + sumA);
        unsigned long long off = 0;
        if (T > 0 && l == 0) off = atomicAdd(&O.arena_ctr[arena * 8], want);
        wave_sync();
        ObsRec *dst = nullptr;
        const int step0 = maxnch > 1 ? 1 << (31 - __builtin_clz(maxnch - 1)) : 0;
        int n_x[4] = {0, 0, 0, 0}; unsigned blocked = 0;                  // records of alignment k so far; k has a row that no op serves: the cursor stays there
#pragma unroll 1
        for (int i0 = 0; i0 < T; i0 += 64) {
            const int i = i0 + l;
            const bool in = i < T;
            bool found = false, emit = false; ObsRec rec{0, 0};
            const int q = (i >= cum[1]) + (i >= cum[2]) + (i >= cum[3]);
            if (in) {
                const int4 ha = *reinterpret_cast<const int4 *>(&s_hdr[q].crel), hb = *reinterpret_cast<const int4 *>(&s_hdr[q].vadj);
                const int hncig = ha.y, hc0 = ha.z, hnch = ha.w, hflag = hb.y, hds = hb.z;
                const int row = hb.x + i;
                const int4 xr = X.rec[row];                             // {pos, info, union index | kind << 30, last SNP position before the row}
                const int mrow = ((unsigned)xr.z >> 30) == 2u ? xr.y : 0;      // MOD row: where its listed reads are, requested beside the chunk's words
                const uint32_t m_lo = X.mod_off[mrow], m_hi = X.mod_off[mrow + 1];
                const int p = xr.x, ps = p + hds, pse = ps + (q << 28);       // pse: against the table's E entries (alignment q's values lie at q << 28)
                // first chunk of the alignment whose running maximum exceeds the row: the op that first reaches beyond it lies there
                int co = -1;
                for (int step = step0; step >= 1; step >>= 1) { const int t = co + step; const int ev = s_tab[hc0 + min(t, hnch - 1)].y; co = (t < hnch && ev <= pse) ? t : co; }
                ++co;                                                     // (co < hnch: ps < reach = the last chunk's running maximum)
                if (step0 == 0) co = 0;
                const int q_snp = xr.w; int j = 0, rp = 0;                // (SNP positions never equal p, lps_set_extra_variants: the last SNP before p + 1 is the last one before p)
                // (a row in the reach of a long clip or insertion that does not serve it - a SNP lies in between - is served by a much later op: the
                //  chunks in between whose own ops all end at or before the row are passed over by their table entry, not loaded)
                for (int cc = co; cc < hnch && !found; ++cc) {
                    if (cc > co) {
                        const int ac = hc0 + cc;
                        if ((ac & 15) == 0 && cc + 16 <= hnch && s_bm[ac >> 4] <= pse) { cc += 15; continue; }
                        if (s_lm[ac] <= pse) continue;
                    }
                    const uint32_t *cw = cg + 8 * (hc0 + cc);
                    const uint4 a = *reinterpret_cast<const uint4 *>(cw), b = *reinterpret_cast<const uint4 *>(cw + 4);
                    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                    int rr = s_tab[hc0 + cc].x;
                    for (int k = 0; k < 8 && !found; ++k) {
                        const int opi = 8 * cc + k;
                        if (opi >= hncig) break;
                        const int op = w[k] & 15u, len = (int)(w[k] >> 4);
                        if (rr + len > ps) {                            // (from the first such op on: every later op reaches beyond the row as well or does not matter - the test is the reference's)
                            if (op_is_match(op)) { found = true; j = opi; rp = rr; }
                            else if (rr - hds > q_snp) { found = true; j = opi; rp = rr; }      // no SNP lies in [ref_pos, p]
                        }
                        rr += len & -(int)op_bit(LPS_RMASK2, w[k]);
                    }
                }
                if (found) {
                    const int kind = (int)((unsigned)xr.z >> 30), info = xr.y, xu = xr.z & 0x3fffffff;
                    const int rp_true = rp - hds;
                    if (kind == 1) {                                    // :1403-1429
                        const double region = (double)(abs(info) + 1);
                        int allele = 0;
                        const int a = max(j - X.sv_window, 0), b = min(j + X.sv_window, hncig);
                        const uint32_t *cig = cg + 8 * hc0;
                        for (int c8 = a >> 3; c8 <= (b - 1) >> 3 && !allele; ++c8) {        // the window's words a lane-chunk at a time (two 16-byte loads a trip instead of one word)
                            const uint4 wa = *reinterpret_cast<const uint4 *>(cig + 8 * c8), wb = *reinterpret_cast<const uint4 *>(cig + 8 * c8 + 4);
                            const uint32_t ww[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const int t = 8 * c8 + k; const int op = ww[k] & 15u; const double len = (double)(int)(ww[k] >> 4);
                                if (t >= a && t < b && (op == 1 || op == 2) && fabs(region - len) / fabs(region) < X.sv_threshold) allele = 1;
                            }
                        }
                        emit = true; rec = ObsRec{xu, (uint32_t)pack_aq(allele, -1)};
                    } else {                                            // :1377-1392
                        const uint32_t name = (uint32_t)s_hdr[q].soff; const bool rev = (hflag & 0x10) != 0;
                        uint32_t lo = m_lo, hi = m_hi; const uint32_t end = hi;
                        // first listed read with name >= this one: a 4-ary search over name << 2 | flags (three probes a trip: a site lists a read per fold of coverage)
                        const uint32_t key = name << 2;
                        while (hi - lo > 7) {                             // 8-ary: seven probes a trip (a site lists a read per fold of coverage: two trips)
                            const uint32_t n8 = (hi - lo) >> 3;
                            uint32_t v[7];
#pragma unroll
                            for (int t = 0; t < 7; ++t) v[t] = X.mod_pack[lo + n8 * (t + 1)];
                            int g = 0;                                      // probes below the key: the answer lies behind the last of them
#pragma unroll
                            for (int t = 0; t < 7; ++t) g += v[t] < key ? 1 : 0;
                            const uint32_t nlo = g ? lo + n8 * g + 1 : lo, nhi = g < 7 ? lo + n8 * (g + 1) : hi;
                            lo = nlo; hi = nhi;
                        }
                        while (hi - lo > 3) {
                            const uint32_t n4 = (hi - lo) >> 2, m1 = lo + n4, m2 = m1 + n4, m3 = m2 + n4;
                            const uint32_t v1 = X.mod_pack[m1], v2 = X.mod_pack[m2], v3 = X.mod_pack[m3];
                            if (v1 >= key) hi = m1; else if (v2 >= key) { lo = m1 + 1; hi = m2; } else if (v3 >= key) { lo = m2 + 1; hi = m3; } else lo = m3 + 1;
                        }
                        bool have = false; uint32_t hit = 0u;                      // the answer is one of lo, lo + 1, lo + 2 (below hi) or hi itself: four probes, one trip
                        {
                            const bool i0 = lo < hi, i1 = lo + 1 < hi, i2 = lo + 2 < hi, ih = hi < end;
                            const uint32_t v0 = i0 ? X.mod_pack[lo] : 0u, v1 = i1 ? X.mod_pack[lo + 1] : 0u, v2 = i2 ? X.mod_pack[lo + 2] : 0u, vh = ih ? X.mod_pack[hi] : 0u;
                            if (i0 && v0 >= key) { have = true; hit = v0; } else if (i1 && v1 >= key) { have = true; hit = v1; } else if (i2 &&
                                    v2 >= key) { have = true; hit = v2; } else if (ih) { have = true; hit = vh; }
                        }
                        if (have && (hit >> 2) == name) {
                            const unsigned f = hit & 3u;
                            // the reference compares modPos with *currentVariantIter even when that is end(): the entry count of the SNP map
                            const bool cursor_ok = V.last_pos >= max(rp_true, p + 1) || p < V.n;
                            if ((((f >> 1) & 1u) != 0) == rev && cursor_ok) { emit = true; rec = ObsRec{xu, (uint32_t)pack_aq((f & 1u) ? 0 : 1, rev ? -3 : -2)}; }
                        }
                    }
                }
            }
            // per alignment: the served rows are a prefix (a row no op serves keeps the reference's cursor: nothing behind it is served); records in row order
            const unsigned long long fm = __ballot(in && !found), em0 = __ballot(emit);
            unsigned long long keep = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int a = max(cum[k] - i0, 0), b = min(cum[k + 1] - i0, 64);
                if (b > a) {
                    const unsigned long long rm = ((b >= 64) ? ~0ull : ((1ull << b) - 1ull)) & ~((1ull << a) - 1ull);
                    unsigned long long ok = rm;
                    if ((blocked >> k) & 1u) ok = 0;
                    else if (fm & rm) { const int first_bad = __builtin_ctzll(fm & rm); ok = rm & ((1ull << first_bad) - 1ull); blocked |= 1u << k; }
                    keep |= ok;
                }
            }
            const unsigned long long em = em0 & keep;
            if (i0 == 0) {                                                // the reservation has had the searches of the first round to arrive
                off = __shfl(off, 0);
                if (off + want > O.arena_size) { if (l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_OBS_OVERFLOW); return; }   // the host grows the arenas and runs again
                dst = O.rec + arena_lo + off;
            }
            if ((em >> l) & 1ull) {
                const int cq = SEL4(q, cum), a = max(cq - i0, 0);
                const unsigned long long before = em & lanemask_lt() & ~((1ull << a) - 1ull);
                dst[cq + SEL4(q, n_x) + __popcll(before)] = rec;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int a = max(cum[k] - i0, 0), b = min(cum[k + 1] - i0, 64);
                if (b > a) { const unsigned long long rm = ((b >= 64) ? ~0ull : ((1ull << b) - 1ull)) & ~((1ull << a) - 1ull); n_x[k] += __popcll(em & rm); }
            }
        }
        // ---- the records merged by position into the rows of the group's alignments (union indices on both sides, no common keys): the rows that
        //      got records move to fresh slots, ONE reservation for all of them, the outputs of the four merges flattened over the lanes; the keys of
        //      rows and records meet in LDS (the walk's tables are done with)
        {
            int nA[4], nB[4], cm[5], ca[5], cb[5]; cm[0] = ca[0] = cb[0] = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                nB[q] = n_x[q]; nA[q] = nB[q] ? __builtin_amdgcn_readlane(h_rcnt, q) : 0;
                cm[q + 1] = cm[q] + nA[q] + nB[q]; ca[q + 1] = ca[q] + nA[q]; cb[q + 1] = cb[q] + nB[q];
            }
            const int TM = cm[4];
            if (TM > 0) {
                unsigned roffq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) roffq[q] = (unsigned)__builtin_amdgcn_readlane((int)h_roff, q);
                const unsigned long long off2 = off + (unsigned long long)T;       // (TM <= sumA + T: reserved with the records)
                ObsRec *out = O.rec + arena_lo + off2;
                __threadfence_block();                                  // the records this wave wrote above are read back below
                int *s_ka = s_lm, *s_kb = reinterpret_cast<int *>(s_tab);   // keys of the rows (<= XF_TAB) / of the records (<= 2 * XF_TAB)
                const bool in_lds = ca[4] <= XF_TAB && cb[4] <= 2 * XF_TAB;
                wave_sync();
                if (in_lds) {
                    for (int i = l; i < ca[4]; i += 64) { const int q = (i >= ca[1]) + (i >= ca[2]) + (i >= ca[3]); s_ka[i] = O.rec[SEL4(q, roffq) + (unsigned)(i - SEL4(q, ca))].var; }
                    for (int i = l; i < cb[4]; i += 64) { const int q = (i >= cb[1]) + (i >= cb[2]) + (i >= cb[3]); s_kb[i] = dst[SEL4(q, cum) + (i - SEL4(q, cb))].var; }
                    wave_sync();
                }
#pragma unroll 1
                for (int o0 = 0; o0 < TM; o0 += 64) {
                    const int og = o0 + l;
                    if (og < TM) {
                        const int q = (og >= cm[1]) + (og >= cm[2]) + (og >= cm[3]);
                        const int o = og - SEL4(q, cm), na = SEL4(q, nA), nb = SEL4(q, nB), a0 = SEL4(q, ca), b0 = SEL4(q, cb);
                        const ObsRec *A = O.rec + SEL4(q, roffq), *B = dst + SEL4(q, cum);
                        int lo = max(0, o - nb), hi = min(o, na);
                        if (in_lds) { while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_ka[a0 + mid] < s_kb[b0 + o - mid - 1]) lo = mid + 1; else hi = mid; } }
                        else { while (lo < hi) { const int mid = (lo + hi) >> 1; if (A[mid].var < B[o - mid - 1].var) lo = mid + 1; else hi = mid; } }
                        const int i = lo, j = o - lo;
                        const int ka = i < na ? (in_lds ? s_ka[a0 + i] : A[i].var) : 0x7fffffff, kb = j < nb ? (in_lds ? s_kb[b0 + j] : B[j].var) : 0x7fffffff;
                        out[og] = (ka < kb) ? A[i] : B[j];
                    }
                }
                if (l < 4 && SEL4(l, nB) > 0) { RowDesc d = O.rows[r0 + l]; d.off = (uint32_t)(arena_lo + off2 + (unsigned)SEL4(l, cm)); d.cnt = SEL4(l, nA) + SEL4(l,
                        nB); d.flags = 0; O.rows[r0 + l] = d; }
            }
        }
        wave_sync();
    }
    if (l < nq && general) redo[atomicAdd(n_redo, 1u)] = (uint32_t)(r0 + l);   // (rare: one atomic per alignment that is left to the general walker)
}

void launch_extra_merge(const VarView &V, const ReadView &R, const ObsView &O, const ExtraView &X, int32_t *x0, uint32_t *redo, unsigned *n_redo, int mapping_quality,
        LpsCounters *cnt, hipStream_t s) {
    if (R.n <= 0) return;
    (void)hipMemsetAsync(n_redo, 0, sizeof(unsigned), s);
    hipLaunchKernelGGL(k_read_x0, dim3((R.n + 255) / 256), dim3(256), 0, s, X, R.ref_start, R.n, x0);
    hipLaunchKernelGGL(k_extra_find, dim3((R.n + 3) / 4), dim3(64), 0, s, V, R, O, X, x0, redo, n_redo, mapping_quality, cnt);
    hipLaunchKernelGGL(k_extra_merge, dim3(std::min(256, R.n)), dim3(64), 0, s, V, R, O, X, redo, n_redo, mapping_quality, cnt);   // what the stream walk queued (nothing with ordinary reads)
}
