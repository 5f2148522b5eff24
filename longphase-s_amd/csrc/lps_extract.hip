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

// coarse position index over the variant table (thread per bucket, plain binary search)
__global__ void k_bucket_index(VarView V, int32_t *bucket) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > V.n_bucket) return;
    const long long key = (long long)b << LPS_BUCKET_SHIFT;
    int lo = 0, hi = V.n;
    while (lo < hi) { const int m = (lo + hi) >> 1; if ((long long)V.pos[m] < key) lo = m + 1; else hi = m; }
    bucket[b] = lo;
}

// one 8-byte record per variant so that a candidate costs ONE gather in the extraction kernel
__global__ void k_variant_pack(VarView V, uint2 *rec) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V.n) return;
    const int rl = V.ref_len[v], al = V.alt_len[v];
    const unsigned kind = (rl == 1 && al == 1) ? 0u : ((rl == 1 && al != 1) ? 1u : ((rl != 1 && al == 1) ? 2u : 3u));
    const unsigned attr = (unsigned)V.ref0[v] | ((unsigned)V.alt0[v] << 8) | (kind << 16) | (V.danger[v] ? VREC_DANGER : 0u) |
                          (V.erased[v] ? VREC_ERASED : 0u) | (V.hpoly[v] >= 3 ? VREC_HPOLY3 : 0u) |
                          ((V.hp1_is_alt && V.hp1_is_alt[v]) ? VREC_HP1ALT : 0u) |
                          (V.somatic_role ? ((unsigned)(V.somatic_role[v] & 3) << 22) | ((unsigned)((V.derive_hp ? V.derive_hp[v] : 0) & 3) << 24) : 0u) |
                          (V.tumor_kind ? ((unsigned)(V.tumor_kind[v] & 7) << 26) : 0u);
    rec[v] = make_uint2((unsigned)V.pos[v], attr);
}

void launch_variant_prep(const VarView &V, int is_ont, int32_t *bucket, uint2 *rec, hipStream_t s) {
    if (V.n == 0) return;
    const int b = 256, g = (V.n + b - 1) / b;
    hipLaunchKernelGGL(k_variant_prep, dim3(g), dim3(b), 0, s, V);
    if (is_ont) hipLaunchKernelGGL(k_filter_snp, dim3(g), dim3(b), 0, s, V);
    hipLaunchKernelGGL(k_variant_pack, dim3(g), dim3(b), 0, s, V, rec);
    hipLaunchKernelGGL(k_bucket_index, dim3((V.n_bucket + 1 + b) / b), dim3(b), 0, s, V, bucket);
}

// ------------------------------------------------------------------------------------------------ extraction
#define EXT_RPW 4   // alignments per wave (processed one after the other)

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
    for (int c0 = 0; c0 < n_cig; c0 += 512) {               // 8 independent 256-B loads in flight per trip
        uint32_t wd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int c = c0 + u * 64 + l; wd[u] = c < n_cig ? cig[c] : 6u; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int op = wd[u] & 15;
            if (op_consumes_ref(op)) span += wd[u] >> 4;
            if (op > 8) bad = true;
        }
    }
    span = wave_sum(span);
    if (__ballot(bad)) { if (l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_BAD_CIGAR); }
    long long endll = (long long)start + span; if (endll > 0x7fffffff) endll = 0x7fffffff;
    p.v0 = var_lower_bound(V, start);
    p.v1 = var_lower_bound(V, (int)endll);
    return p;
}

__global__ __launch_bounds__(256, 6) void k_extract_phase(VarView V, ReadView R, ObsView O, ClipView C, int mapping_quality,
                                                       LpsCounters *cnt) {
    __shared__ int s_ref[4][LPS_SEG];
    __shared__ int s_qry[4][LPS_SEG];
    __shared__ uint32_t s_cig[4][LPS_SEG + 1];
    const int w = threadIdx.x >> 6, l = lane_id();
    int *sref = s_ref[w], *sqry = s_qry[w]; uint32_t *scig = s_cig[w];
    // Output rows are reserved per alignment on one of LPS_ARENAS counters (own cache line each).  Workgroups are
    // dealt round-robin over the 8 XCDs, so arena = blockIdx % 64 keeps each counter inside ONE XCD's L2, and pass 2
    // follows pass 1 of the same alignment immediately: its CIGAR re-read is an L2 hit instead of a second HBM sweep.
    const int arena = blockIdx.x % O.n_arenas;
    const unsigned long long arena_lo = (unsigned long long)arena * O.arena_size;
#pragma unroll 1
    for (int q = 0; q < EXT_RPW; ++q) {
        const int r = (blockIdx.x * 4 + w) * EXT_RPW + q;
        if (r >= R.n) break;
        const ReadPlan pl = plan_read(V, R, r, mapping_quality, cnt);
        const int v0 = pl.v0, v1 = pl.v1, cand = v1 - v0;
        const bool live_q = pl.live;
        unsigned long long my_base = 0; bool overflow = false;
        if (live_q && cand > 0) {
            unsigned long long off = 0;
            if (l == 0) off = atomicAdd(&O.arena_ctr[arena * 8], (unsigned long long)cand);
            off = __shfl(off, 0);
            overflow = off + (unsigned long long)cand > O.arena_size;
            my_base = arena_lo + off;
        }
        if (l < LPS_CLIP_SLOTS && (!live_q || overflow)) C.opidx_fb[(size_t)r * LPS_CLIP_SLOTS + l] = -1;
        if (!live_q) { if (l == 0) { O.row_off[r] = 0; O.row_cnt[r] = 0; O.row_fail[r] = 0x7fffffff; O.row_flags[r] = 0; } continue; }
        if (overflow) {
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

        // ---- pass 2: segments of LPS_SEG ops -> LDS (ref prefix, query prefix, raw op word); the candidate variants
        //      (one packed record per lane, loaded BEFORE the prefix build so both latencies overlap) search them
        int ref_pos = start, q_pos = 0, n_emit = 0, fail_op = 0x7fffffff, vcur = v0, n_clip = 0;
        bool had_any = false;
        for (int seg0 = 0; seg0 < n_cig; seg0 += LPS_SEG) {
            const int nseg = min(LPS_SEG, n_cig - seg0);
            uint2 vr = make_uint2(0x7fffffffu, 0u);
            if (vcur + l < v1) vr = V.rec[vcur + l];
            const uint32_t nextw = (seg0 + nseg < n_cig) ? cig[seg0 + nseg] : 0xfu;      // op after the segment (0xf = none)
            uint32_t wds[LPS_SEG / 64];                          // the whole segment's CIGAR words: LPS_SEG/64 loads in flight
#pragma unroll
            for (int u = 0; u < LPS_SEG / 64; ++u) { const int idx = u * 64 + l; wds[u] = idx < nseg ? cig[seg0 + idx] : 6u; }
#pragma unroll
            for (int u = 0; u < LPS_SEG / 64; ++u) {
                const int c0 = u * 64;
                if (c0 >= nseg) break;
                const int idx = c0 + l;
                const uint32_t wd = wds[u];
                const int op = idx < nseg ? (int)(wd & 15) : 6, len = (int)(wd >> 4);
                const int radv = op_consumes_ref(op) ? len : 0, qadv = op_consumes_query(op) ? len : 0;
                const int ir = wave_incl_scan_dpp(radv), iq = wave_incl_scan_dpp(qadv);
                const int my_ref = ref_pos + ir - radv, my_q = q_pos + iq - qadv;
                if (idx < nseg) { sref[idx] = my_ref; sqry[idx] = my_q; scig[idx] = wd; }
                // getClip (:1613-1620,1636-1645): soft/hard clips longer than 5; FRONT iff CIGAR index 0.
                // Events go to the read's own LPS_CLIP_SLOTS slots (no atomics here); compaction happens later.
                const bool clip = (op == 4 || op == 5) && len > 5;
                const unsigned long long cm = __ballot(clip);
                if (cm) {
                    if (clip) {
                        const int slot = n_clip + __popcll(cm & lanemask_lt());
                        if (slot < LPS_CLIP_SLOTS) { C.pos[(size_t)r * LPS_CLIP_SLOTS + slot] = my_ref; C.opidx_fb[(size_t)r * LPS_CLIP_SLOTS + slot] = ((seg0 + idx) << 1) | ((seg0 + idx) != 0); }
                        else atomicOr(&cnt->err, (unsigned)LPS_ERR_CLIP_OVERFLOW);
                    }
                    n_clip += __popcll(cm);
                }
                ref_pos += __shfl(ir, 63); q_pos += __shfl(iq, 63);
            }
            if (l == 0) scig[nseg] = nextw;
            wave_sync();
            // candidates are position-sorted: those inside this segment's reference interval form a prefix of the chunk
            const bool last_seg = seg0 + nseg >= n_cig;
            while (true) {
                const int v = vcur + l;
                const int p = (int)vr.x;
                const bool mine = v < v1 && (last_seg || p < ref_pos);
                const int n_in = __popcll(__ballot(mine));
                int pprev = __shfl_up(p, 1);                     // position of the previous variant (all lanes take part)
                if (l == 0) pprev = (v > 0 && v < v1) ? V.pos[v - 1] : -1;
                bool emit = false, fail = false; int allele = -1, qv = 0, opi = 0;
                if (mine) {
                    const unsigned at = vr.y;
                    int lo = 0, hi = nseg;                       // first j with sref[j] > p
                    while (lo < hi) { const int m = (lo + hi) >> 1; if (sref[m] > p) hi = m; else lo = m + 1; }
                    const int j = lo - 1;
                    if (j >= 0) {
                        const uint32_t wd = scig[j];
                        const int op = wd & 15, len = (int)(wd >> 4);
                        const int rs = sref[j], qs = sqry[j];
                        opi = seg0 + j;
                        if (p < rs + len) {
                            const unsigned kind = VREC_KIND(at);
                            const char ref_c = (char)(at & 0xff), alt_c = (char)((at >> 8) & 0xff);
                            if (op_is_match(op)) {                                            // :1445-1520
                                const int off = p - rs;
                                if (qs + off + 1 > lq) fail = true;                           // :1453-1455
                                else if (kind == 0) {
                                    const int qi = qs + off;
                                    const char base_c = nt16_char(seq[qi >> 1] >> ((~qi & 1) << 2));
                                    if (base_c == ref_c) allele = 0; else if (base_c == alt_c) allele = 1;
                                    qv = qual[qi];
                                    emit = allele != -1;
                                } else if ((kind == 1 || kind == 2) && opi + 1 < n_cig) {     // indel variant :1470-1510
                                    const int want = (kind == 1) ? 1 : 2;                      // next op must be I resp. D
                                    allele = (rs + len - 1 == p && (int)(scig[j + 1] & 15) == want) ? 1 : 0;
                                    qv = (at & VREC_DANGER) ? -5 : -4;
                                    emit = true;
                                }
                            } else if (op == 2) {                                             // :1539-1607
                                // only the first variant at/after the deletion start is examined by the reference
                                const bool first_in = (v == 0) || pprev < rs;
                                if (first_in && (at & VREC_HPOLY3)) {
                                    if (qs + 1 > lq) fail = true;                             // :1559-1561
                                    else if (kind == 0) {
                                        const char base_c = nt16_char(seq[qs >> 1] >> ((~qs & 1) << 2));
                                        if (base_c == ref_c) allele = 0; else if (base_c == alt_c) allele = 1;
                                        qv = qual[qs];
                                        emit = allele != -1;
                                    } else if (kind == 2) { allele = 1; qv = -4; emit = true; }
                                }
                            }
                        }
                    }
                    if (fail) fail_op = min(fail_op, opi);
                    had_any |= emit;
                    if (emit && (at & VREC_ERASED)) emit = false;                              // filterSNP (:895-911)
                }
                const unsigned long long em = __ballot(emit);
                if (emit) {
                    const unsigned long long slot = my_base + n_emit + __popcll(em & lanemask_lt());
                    O.var[slot] = v; O.aq[slot] = pack_aq(allele, qv);
                }
                n_emit += __popcll(em);
                vcur += n_in;
                if (n_in < 64 || vcur >= v1) break;
                vr = make_uint2(0x7fffffffu, 0u);
                if (vcur + l < v1) vr = V.rec[vcur + l];
            }
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
        if (l >= n_clip && l < LPS_CLIP_SLOTS) C.opidx_fb[(size_t)r * LPS_CLIP_SLOTS + l] = -1;   // unused slots
    }
}

void launch_extract_phase(const VarView &V, const ReadView &R, const ObsView &O, const ClipView &C,
                          int mapping_quality, LpsCounters *cnt, hipStream_t s) {
    if (R.n == 0) return;
    hipLaunchKernelGGL(k_extract_phase, dim3((R.n + 4 * EXT_RPW - 1) / (4 * EXT_RPW)), dim3(256), 0, s, V, R, O, C, mapping_quality, cnt);
}
