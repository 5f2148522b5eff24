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
__global__ __launch_bounds__(64 * XM_WPB) void k_extra_merge(VarView V, ReadView R, ObsView O, ExtraView X, int mapping_quality, LpsCounters *cnt) {
    __shared__ __attribute__((aligned(16))) int s_ref[XM_WPB][LPS_SEG];
    __shared__ __attribute__((aligned(16))) int s_qry[XM_WPB][LPS_SEG];
    __shared__ __attribute__((aligned(16))) uint32_t s_cig[XM_WPB][LPS_SEG + 4];
    __shared__ int s_pm[XM_WPB][LPS_SEG];
    __shared__ ObsRec s_ex[XM_WPB][XM_CAP];
    const int w = threadIdx.x >> 6, l = lane_id();
    const int r = blockIdx.x * XM_WPB + w;
    if (r >= R.n) return;
    int *sref = s_ref[w], *sqry = s_qry[w], *spm = s_pm[w]; uint32_t *scig = s_cig[w]; ObsRec *sex = s_ex[w];

    const RowDesc rd = O.rows[r];
    const int start = R.ref_start[r], flag = R.flag[r];
    const bool live = !(R.mapq[r] < mapping_quality || (flag & 0x4) || (flag & 0x100) || (flag & 0x400) || start >= V.last_pos) && rd.fail == 0x7fffffff;
    if (!live) return;                                                  // filtered (:1282-1291) or get_snp returned early: the alignment has no row
    const int n_cig = (int)(R.cigar_off[r + 1] - R.cigar_off[r]);
    const uint32_t *cig = R.cigar + R.cigar_off[r];
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
                const int arena = blockIdx.x % O.n_arenas;
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
    if (n_emit == 0) {                                                  // nothing recorded: the row stays where it is, in union indices
        for (int k = l; k < rd.cnt; k += 64) { ObsRec *o = O.rec + rd.off + k; o->var = X.snp_u[o->var]; }
        return;
    }
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
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (X.snp_u[A[mid].var] < B[o - mid - 1].var) lo = mid + 1; else hi = mid; }
            const int i = lo, j = o - lo;
            ObsRec a{0x7fffffff, 0}, b{0x7fffffff, 0};
            if (i < nA) { a = A[i]; a.var = X.snp_u[a.var]; }
            if (j < nB) b = B[j];
            v = (a.var < b.var) ? a : b;
        }
        wave_sync();                                                    // every lane holds its value before anything of this round is written
        if (o < total) out[o] = v;
        wave_sync();
    }
    if (l == 0) { RowDesc d = rd; d.off = new_off; d.cnt = total; d.flags = 0; O.rows[r] = d; }
}

void launch_extra_merge(const VarView &V, const ReadView &R, const ObsView &O, const ExtraView &X, int mapping_quality, LpsCounters *cnt, hipStream_t s) {
    if (R.n <= 0) return;
    hipLaunchKernelGGL(k_extra_merge, dim3((R.n + XM_WPB - 1) / XM_WPB), dim3(64 * XM_WPB), 0, s, V, R, O, X, mapping_quality, cnt);
}
