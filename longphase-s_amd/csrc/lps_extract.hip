// lps_extract.hip — variant-table preparation and the read x variant allele extraction kernel (gfx950).
//
// Replaces (reference file:line, relative to /root/reference/):
//   SnpParser::getVariants_markindel   src/phase/ParsingBam.cpp:378-417     -> k_variant_prep (danger flag)
//   homopolymerLength                  src/shared/Util.cpp:21-54            -> k_variant_prep (hpoly)
//   SnpParser::filterSNP               src/phase/ParsingBam.cpp:837-912     -> k_filter_snp (erased flag)
//   BamParser::direct_detect_alleles   src/phase/ParsingBam.cpp:1243-1301   -> k_extract_phase (filters)
//   BamParser::get_snp / getClip       src/phase/ParsingBam.cpp:1303-1645   -> k_extract_phase
//
// Design (MI355X-first, not a translation of the reference's cursor walk):
//   * one 64-lane wavefront per alignment; CIGAR words are read with coalesced 256-B wave loads and turned into
//     (ref_pos, query_pos) prefix arrays by a wave scan, staged per 1024-op segment in LDS (8 KB per wave);
//   * instead of walking ops and advancing a variant cursor, the VARIANTS search the ops: the candidate variants
//     of the read (a contiguous slice of the position-sorted table found by a 64-ary wave search) are mapped one
//     per lane and each binary-searches the LDS prefix array for the op that contains it - ~26 variants x 10 LDS
//     probes instead of ~800 ops x table probes;
//   * seq/qual are touched only at variant sites (sparse 1-byte gathers), output rows are reserved with one
//     atomic per read and written compacted with ballot/popcount ranks, so every observation is written once.
#include "lps_kernels.h"

// ------------------------------------------------------------------------------------------------ variants
__global__ void k_variant_prep(VarView V) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V.n) return;
    const long long L = V.ref_len_eff;
    const long long p0 = V.pos[v];
    auto at = [&](long long i) -> char { return (i >= 0 && i < L) ? V.ref[i] : '\0'; };
    uint8_t danger = 0;
    if (V.ref_len[v] > 1 || V.alt_len[v] > 1) {
        long long p = p0; const char a = at(p + 1), b = at(p + 2); int i = 0;
        while (i < 5) { if (a != at(p + 1) || b != at(p + 2)) break; p += 2; ++i; }
        danger = (i == 5);
    }
    V.danger[v] = danger;
    V.hpoly[v] = (uint8_t)homopolymer_length(V.ref, L, p0);
    V.erased[v] = 0;
}

// filterSNP: the reference's erase-while-iterating pair scan only ever relates SNPs <= 2 bp apart, so the table
// splits into independent chains at every gap > 2 bp; the head of each chain replays the scan for its chain.
__global__ void k_filter_snp(VarView V) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V.n) return;
    if (v != 0 && V.pos[v] - V.pos[v - 1] <= 2) return;   // not a chain head
    int cur = v, nxt = v + 1;
    while (nxt < V.n && V.pos[nxt] - V.pos[nxt - 1] <= 2) {
        if (V.hpoly[cur] >= 3 && V.hpoly[nxt] >= 3 && V.pos[nxt] - V.pos[cur] <= 2) { V.erased[nxt] = 1; ++nxt; }
        else { cur = nxt; ++nxt; }
    }
}

void launch_variant_prep(const VarView &V, int is_ont, hipStream_t s) {
    if (V.n == 0) return;
    const int b = 256, g = (V.n + b - 1) / b;
    hipLaunchKernelGGL(k_variant_prep, dim3(g), dim3(b), 0, s, V);
    if (is_ont) hipLaunchKernelGGL(k_filter_snp, dim3(g), dim3(b), 0, s, V);
}

// ------------------------------------------------------------------------------------------------ extraction
__device__ __forceinline__ bool op_consumes_ref(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
__device__ __forceinline__ bool op_consumes_query(int op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }
__device__ __forceinline__ bool op_is_match(int op) { return op == 0 || op == 7 || op == 8; }

#define EXT_RPW 4   // alignments per wave: one output reservation (atomic) per workgroup covers 4 waves x 4 reads

struct ReadPlan { int v0, v1; bool live; };

// pass 1 of one alignment: filters, reference span (one coalesced sweep of the CIGAR, wave reduction) and the slice
// [v0,v1) of candidate variants.  Wave-uniform result.
__device__ __forceinline__ ReadPlan plan_read(const VarView &V, const ReadView &R, int r, int mapping_quality, LpsCounters *cnt) {
    ReadPlan p{0, 0, false};
    if (r >= R.n) return p;
    const int l = lane_id();
    const int start = R.ref_start[r];
    const int flag = R.flag[r];
    // direct_detect_alleles filters (:1282-1291) + region "chr:1-<lastSNPPos>" (:1273)
    if (R.mapq[r] < mapping_quality || (flag & 0x4) || (flag & 0x100) || (flag & 0x400) || start >= V.last_pos) return p;
    p.live = true;
    const uint64_t coff = R.cigar_off[r];
    const int n_cig = (int)(R.cigar_off[r + 1] - coff);
    const uint32_t *cig = R.cigar + coff;
    long long span = 0; bool bad = false;
    for (int c = l; c < n_cig; c += 64) {
        const uint32_t wd = cig[c]; const int op = wd & 15;
        if (op_consumes_ref(op)) span += wd >> 4;
        if (op > 8) bad = true;
    }
    span = wave_sum(span);
    if (__ballot(bad)) { if (l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_BAD_CIGAR); }
    long long endll = (long long)start + span; if (endll > 0x7fffffff) endll = 0x7fffffff;
    p.v0 = wave_lower_bound(V.pos, 0, V.n, start);
    p.v1 = wave_lower_bound(V.pos, p.v0, V.n, (int)endll);
    return p;
}

__global__ __launch_bounds__(256) void k_extract_phase(VarView V, ReadView R, ObsView O, ClipView C, int mapping_quality,
                                                       LpsCounters *cnt) {
    __shared__ int s_ref[4][LPS_SEG];
    __shared__ int s_qry[4][LPS_SEG];
    __shared__ unsigned long long s_cand[4][EXT_RPW];
    __shared__ unsigned long long s_base;
    const int w = threadIdx.x >> 6, l = lane_id();
    int *sref = s_ref[w], *sqry = s_qry[w];
    const int r0 = (blockIdx.x * 4 + w) * EXT_RPW;

    // ---- pass 1 for the wave's EXT_RPW alignments, then ONE reservation per workgroup
    ReadPlan plan[EXT_RPW];
#pragma unroll
    for (int q = 0; q < EXT_RPW; ++q) {
        plan[q] = plan_read(V, R, r0 + q, mapping_quality, cnt);
        if (l == 0) s_cand[w][q] = (unsigned long long)(plan[q].v1 - plan[q].v0);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < 4 * EXT_RPW; ++i) tot += (&s_cand[0][0])[i];
        s_base = tot ? atomicAdd(&cnt->obs_total, tot) : 0ull;
    }
    __syncthreads();
    unsigned long long base = s_base;
    for (int i = 0; i < w * EXT_RPW; ++i) base += (&s_cand[0][0])[i];

#pragma unroll 1
    for (int q = 0; q < EXT_RPW; ++q) {
        const int r = r0 + q;
        if (r >= R.n) break;
        const int v0 = plan[q].v0, v1 = plan[q].v1, cand = v1 - v0;
        const unsigned long long my_base = base; base += (unsigned long long)cand;
        if (!plan[q].live) { if (l == 0) { O.row_off[r] = 0; O.row_cnt[r] = 0; O.row_fail[r] = 0x7fffffff; O.row_flags[r] = 0; } continue; }
        if (my_base + (unsigned long long)cand > O.capacity) {
            if (l == 0) { atomicOr(&cnt->err, (unsigned)LPS_ERR_OBS_OVERFLOW); O.row_off[r] = 0; O.row_cnt[r] = 0; O.row_fail[r] = 0x7fffffff; O.row_flags[r] = 0; }
            continue;
        }
        const int start = R.ref_start[r];
        const uint64_t coff = R.cigar_off[r];
        const int n_cig = (int)(R.cigar_off[r + 1] - coff);
        const uint32_t *cig = R.cigar + coff;
        const uint8_t *seq = R.seq + R.seq_off[r];
        const uint8_t *qual = R.qual + R.qual_off[r];
        const int lq = R.l_qseq[r];

        // ---- pass 2: segments of LPS_SEG ops -> LDS prefix arrays; candidate variants search them
        int ref_pos = start, q_pos = 0, n_emit = 0, fail_op = 0x7fffffff, vcur = v0;
        bool had_any = false;
        for (int seg0 = 0; seg0 < n_cig; seg0 += LPS_SEG) {
            const int nseg = min(LPS_SEG, n_cig - seg0);
            for (int c0 = 0; c0 < nseg; c0 += 64) {
                const int idx = c0 + l;
                const uint32_t wd = idx < nseg ? cig[seg0 + idx] : 0u;
                const int op = idx < nseg ? (int)(wd & 15) : 6, len = (int)(wd >> 4);
                const int radv = op_consumes_ref(op) ? len : 0, qadv = op_consumes_query(op) ? len : 0;
                const int ir = wave_incl_scan(radv), iq = wave_incl_scan(qadv);
                const int my_ref = ref_pos + ir - radv, my_q = q_pos + iq - qadv;
                if (idx < nseg) { sref[idx] = my_ref; sqry[idx] = my_q; }
                // getClip (:1613-1620,1636-1645): soft/hard clips longer than 5; FRONT iff CIGAR index 0
                const bool clip = (op == 4 || op == 5) && len > 5;
                const unsigned long long cm = __ballot(clip);
                if (cm) {
                    unsigned cb = 0;
                    if (l == 0) cb = atomicAdd(&cnt->n_clips, (unsigned)__popcll(cm));
                    cb = __shfl(cb, 0);
                    if (clip) {
                        const unsigned slot = cb + __popcll(cm & lanemask_lt());
                        if (slot < C.capacity) { C.pos[slot] = my_ref; C.read[slot] = r; C.opidx_fb[slot] = ((seg0 + idx) << 1) | ((seg0 + idx) != 0); }
                        else atomicOr(&cnt->err, (unsigned)LPS_ERR_CLIP_OVERFLOW);
                    }
                }
                ref_pos += __shfl(ir, 63); q_pos += __shfl(iq, 63);
            }
            wave_sync();
            // variants whose position falls inside this segment's reference interval
            const int vend = (seg0 + nseg >= n_cig) ? v1 : wave_lower_bound(V.pos, vcur, v1, ref_pos);
            for (int vb = vcur; vb < vend; vb += 64) {
                const int v = vb + l;
                bool emit = false, fail = false; int allele = -1, qv = 0, opi = 0;
                if (v < vend) {
                    const int p = V.pos[v];
                    int lo = 0, hi = nseg;                       // first j with sref[j] > p
                    while (lo < hi) { const int m = (lo + hi) >> 1; if (sref[m] > p) hi = m; else lo = m + 1; }
                    const int j = lo - 1;
                    if (j >= 0) {
                        const uint32_t wd = cig[seg0 + j];
                        const int op = wd & 15, len = (int)(wd >> 4);
                        const int rs = sref[j], qs = sqry[j];
                        opi = seg0 + j;
                        if (p < rs + len) {
                            const int rl = V.ref_len[v], al = V.alt_len[v];
                            if (op_is_match(op)) {                                            // :1445-1520
                                const int off = p - rs;
                                if (qs + off + 1 > lq) fail = true;                           // :1453-1455
                                else {
                                    if (rl == 1 && al == 1) {
                                        const int qi = qs + off;
                                        const char base_c = nt16_char(seq[qi >> 1] >> ((~qi & 1) << 2));
                                        if (base_c == (char)V.ref0[v]) allele = 0; else if (base_c == (char)V.alt0[v]) allele = 1;
                                        qv = qual[qi];
                                    }
                                    const bool has_next = opi + 1 < n_cig;
                                    if (rl == 1 && al != 1 && has_next) {                     // insertion variant :1470-1491
                                        allele = (rs + len - 1 == p && (cig[seg0 + j + 1] & 15) == 1) ? 1 : 0;
                                        qv = V.danger[v] ? -5 : -4;
                                    }
                                    if (rl != 1 && al == 1 && has_next) {                     // deletion variant :1495-1510
                                        allele = (rs + len - 1 == p && (cig[seg0 + j + 1] & 15) == 2) ? 1 : 0;
                                        qv = V.danger[v] ? -5 : -4;
                                    }
                                    emit = allele != -1;
                                }
                            } else if (op == 2) {                                             // :1539-1607
                                // only the first variant at/after the deletion start is examined by the reference
                                const bool first_in = (v == 0) || V.pos[v - 1] < rs;
                                if (first_in && V.hpoly[v] >= 3) {
                                    if (qs + 1 > lq) fail = true;                             // :1559-1561
                                    else if (rl == 1 && al == 1) {
                                        const char base_c = nt16_char(seq[qs >> 1] >> ((~qs & 1) << 2));
                                        if (base_c == (char)V.ref0[v]) allele = 0; else if (base_c == (char)V.alt0[v]) allele = 1;
                                        qv = qual[qs];
                                        emit = allele != -1;
                                    } else if (rl != 1 && al == 1) { allele = 1; qv = -4; emit = true; }
                                }
                            }
                        }
                    }
                }
                if (fail) fail_op = min(fail_op, opi);
                had_any |= emit;
                if (emit && V.erased[v]) emit = false;                                         // filterSNP (:895-911)
                const unsigned long long em = __ballot(emit);
                if (emit) {
                    const unsigned long long slot = my_base + n_emit + __popcll(em & lanemask_lt());
                    O.var[slot] = v; O.aq[slot] = pack_aq(allele, qv);
                }
                n_emit += __popcll(em);
            }
            vcur = vend;
            wave_sync();
        }
        fail_op = wave_min(fail_op);
        const bool any = __ballot(had_any) != 0;
        if (l == 0) {
            const bool dropped = fail_op != 0x7fffffff;
            O.row_off[r] = (uint32_t)my_base;
            O.row_cnt[r] = dropped ? 0 : n_emit;
            O.row_fail[r] = fail_op;
            O.row_flags[r] = (!dropped && any && n_emit == 0) ? 1 : 0;
        }
    }
}

void launch_extract_phase(const VarView &V, const ReadView &R, const ObsView &O, const ClipView &C,
                          int mapping_quality, LpsCounters *cnt, hipStream_t s) {
    if (R.n == 0) return;
    hipLaunchKernelGGL(k_extract_phase, dim3((R.n + 4 * EXT_RPW - 1) / (4 * EXT_RPW)), dim3(256), 0, s, V, R, O, C, mapping_quality, cnt);
}
