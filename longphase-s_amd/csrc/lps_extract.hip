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
//   * a 64-lane wavefront takes FOUR consecutive alignments (planned together: one chain of dependent loads for the four); CIGAR words are
//     staged 8 per lane, 512 per segment, turned into (reference start, query start) per op by one pair of DPP wave scans and kept in LDS;
//   * instead of walking ops and advancing a variant cursor, the VARIANTS search the ops: the candidate variants of the segment (a contiguous
//     slice of the position-sorted table, one packed 8-byte record per lane) find their op with a fixed, branch-free sequence of LDS probes;
//   * what a candidate needs from the read is decided in two steps.  The search leaves a HIT (variant, query index or the finished call) in an
//     LDS list - nothing else happens per segment, where only ~13 of 64 lanes hold a candidate.  When the wave's four alignments are through,
//     the hits (~100) are resolved 64 at a time with every lane busy: base + quality gathered (seq/qual are touched only at variant sites),
//     allele called, filterSNP's erasures applied, the survivors compacted in place;
//   * the rows of all four alignments are reserved with ONE atomicAdd of their exact size and leave LDS as coalesced 8-byte records
//     {variant, allele|quality}; the four 16-byte row descriptors are one 64-byte line; clip events go to a list (one reservation per wave
//     that has any).  A wave whose hits do not fit the list (long reads over very dense variants) queues itself for k_extract_redo, which
//     walks such alignments with the direct-to-memory path.
#include <algorithm>

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

// one 8-byte record per variant so that a candidate costs ONE gather in the extraction kernel
// + (the workgroups after the variants') the coarse position index over the table: thread per bucket, plain binary search
__global__ void k_variant_pack(VarView V, uint2 *rec, int32_t *bucket, int nb_pack) {
    if ((int)blockIdx.x >= nb_pack) {
        const int b = ((int)blockIdx.x - nb_pack) * blockDim.x + threadIdx.x;
        if (b > V.n_bucket) return;
        const long long key = (long long)b << LPS_BUCKET_SHIFT;
        int lo = 0, hi = V.n;
        while (lo < hi) { const int m = (lo + hi) >> 1; if ((long long)V.pos[m] < key) lo = m + 1; else hi = m; }
        bucket[b] = lo;
        return;
    }
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
    hipLaunchKernelGGL(k_variant_pack, dim3(g + (V.n_bucket + 1 + b) / b), dim3(b), 0, s, V, rec, bucket, g);
}

// ------------------------------------------------------------------------------------------------ extraction
#define EXT_RPW 4       // alignments per wave
#define EXT_CAP 384     // hits buffered per wave in LDS (8 bytes each): with 4 workgroups per CU the 160 KB of LDS hold 40 KB each
#define EXT_OVF 2048    // ... and in one chunk of global memory a wave takes when the LDS list is full (four reads of 200 kb hold ~1 000 hits)
#define EXT_CLIPS 16    // clip events buffered per wave

// first variant with pos >= key, searched by ONE lane (the planning step runs four of these side by side); == var_lower_bound
__device__ __forceinline__ int lane_var_lower_bound(const VarView &V, int key) {
    if (key < 0) return 0;
    const int b = key >> LPS_BUCKET_SHIFT;
    int lo, hi;
    if (b >= V.n_bucket) { lo = V.bucket[V.n_bucket]; hi = V.n; } else { lo = V.bucket[b]; hi = V.bucket[b + 1]; }
    while (lo < hi) { const int m = (lo + hi) >> 1; if (V.pos[m] < key) lo = m + 1; else hi = m; }
    return lo;
}

// hit word 0: bits 0-21 variant index, 22 finished call (else: gather base and quality at query index = word 1), 23 allele of a finished call,
//             24 quality sentinel -5 (else -4) of a finished call, 25 erased by filterSNP
#define HIT_FINAL (1u << 22)
#define HIT_ALLELE (1u << 23)
#define HIT_Q5 (1u << 24)
#define HIT_ERASED (1u << 25)

// waves per workgroup of k_extract_phase.  The waves of a workgroup share nothing; one wave per workgroup gives its 9.6 KB of LDS back the moment
// that wave is done instead of when the slowest of four is (alignments differ tenfold in length)
#ifndef EXT_WPB
#define EXT_WPB 1
#endif
__global__ __launch_bounds__(64 * EXT_WPB, 4) void k_extract_phase(VarView V, ReadView R, ObsView O, ClipView C, int mapping_quality,
                                                       LpsCounters *cnt, uint32_t *redo_list, unsigned *n_redo, uint2 *ovf, unsigned *ovf_ctr, unsigned ovf_chunks) {
    __shared__ __attribute__((aligned(16))) int s_ref[EXT_WPB][LPS_SEG];
    __shared__ __attribute__((aligned(16))) int s_qry[EXT_WPB][LPS_SEG];
    __shared__ __attribute__((aligned(16))) uint32_t s_cig[EXT_WPB][LPS_SEG + 4];
    __shared__ __attribute__((aligned(16))) uint2 s_hit[EXT_WPB][EXT_CAP];
    __shared__ ClipEv s_clip[EXT_WPB][EXT_CLIPS];
    enum { H_START, H_LQ, H_REL, H_V0, H_SOFF, H_QOFF = H_SOFF + 2, H_HIT0 = H_QOFF + 2, H_FAIL, H_WORDS };
    __shared__ int s_hdr[EXT_WPB][EXT_RPW + 1][H_WORDS];
    const int w = threadIdx.x >> 6, l = lane_id();
    int *sref = s_ref[w], *sqry = s_qry[w]; uint32_t *scig = s_cig[w];
    uint2 *hit = s_hit[w]; ClipEv *clipb = s_clip[w];
    int *hdr = s_hdr[w][0];
    // Output rows are reserved on one of LPS_ARENAS counters (own cache line each).  Workgroups are dealt round-robin over the 8 XCDs, so
    // arena = blockIdx % 64 keeps each counter inside ONE XCD's L2.
    const int arena = blockIdx.x % O.n_arenas;
    const unsigned long long arena_lo = (unsigned long long)arena * O.arena_size;
    const int job = blockIdx.x * EXT_WPB + w;
    const int r0 = job * EXT_RPW;
    if (r0 >= R.n) return;
    const int nq = min(EXT_RPW, R.n - r0);
    static_assert(EXT_RPW == 4 && LPS_SEG == 512, "lane layout of the planning step");

    // ---- plan: headers, alignment q in lane q.  direct_detect_alleles filters (:1282-1291) + region "chr:1-<lastSNPPos>" (:1273)
    int h_start = 0, h_lq = 0, h_rel = 0; bool h_live = false; unsigned long long h_coff = 0, h_soff = 0, h_qoff = 0;
    if (l <= nq) h_coff = R.cigar_off[r0 + l];
    if (l < nq) {
        const int r = r0 + l; h_start = R.ref_start[r]; h_lq = R.l_qseq[r]; h_soff = R.seq_off[r]; h_qoff = R.qual_off[r];
        const int flag = R.flag[r];
        h_live = !(R.mapq[r] < mapping_quality || (flag & 0x4) || (flag & 0x100) || (flag & 0x400) || h_start >= V.last_pos);
    }
    const unsigned live_mask = (unsigned)__ballot(h_live) & 15u;
    const unsigned long long c_lo = __shfl(h_coff, 0);
    if (l <= nq) h_rel = (int)(h_coff - c_lo);                         // op index of alignment q's first op inside the wave's CIGAR range
    const uint32_t *cg = R.cigar + c_lo;

    // what was requested ahead for segment (pf_q, pf_seg): CIGAR words, op after the segment, candidate records, predecessor position
    uint32_t pw[8]; uint32_t pnext = 0xfu; uint2 pvr = make_uint2(0x7fffffffu, 0u); int ppv = -1;
    int pf_q = -1, pf_seg = 0; bool pf_vr_ok = false;
#pragma unroll
    for (int u = 0; u < 8; ++u) pw[u] = 6u;
    int q = live_mask ? __builtin_ctz(live_mask) : nq;
    if (q < nq) {                                                      // first segment of the first alignment: on its way while the bounds are searched
        const int crel = __shfl(h_rel, q), ncq = __shfl(h_rel, q + 1) - crel;
        if (ncq > 0) {
            const int nsegn = min(LPS_SEG, ncq);
            pnext = (nsegn < ncq) ? cg[crel + nsegn] : 0xfu;
            request_ops8(cg + crel, 8 * l, nsegn, pw);
            pf_q = q; pf_seg = 0;
        }
    }
    // first candidate of each alignment: four lanes search the position-sorted table side by side
    int h_v0 = 0;
    if (l < 4 && h_live) h_v0 = lane_var_lower_bound(V, h_start);
    if (l <= 4) {
        int *h = s_hdr[w][l];
        h[H_REL] = h_rel;
        if (l < 4) {
            h[H_START] = h_start; h[H_LQ] = h_lq; h[H_V0] = h_v0;
            h[H_SOFF] = (int)(unsigned)h_soff; h[H_SOFF + 1] = (int)(unsigned)(h_soff >> 32);
            h[H_QOFF] = (int)(unsigned)h_qoff; h[H_QOFF + 1] = (int)(unsigned)(h_qoff >> 32);
            h[H_HIT0] = 0; h[H_FAIL] = 0x7fffffff;
        }
    }
    wave_sync();

    // ---- walk: segments of LPS_SEG ops -> LDS (ref prefix, query prefix, raw op word); the candidate variants (one packed record per lane)
    //      search them and leave hits
    int n_hit = 0, n_clip = 0;
    bool overflow = false;
    uint2 *ovf_mine = nullptr;                                          // this wave's chunk of the global hit list, taken when the LDS list is full
#pragma unroll 1
    while (q < nq && !overflow) {
        const int r = r0 + q;
        int *h = hdr + q * H_WORDS;
#define HU(i) __builtin_amdgcn_readfirstlane(h[i])
        const int start = HU(H_START), lq = HU(H_LQ), crel = HU(H_REL), n_cig = HU(H_WORDS + H_REL) - crel;
        const uint32_t *cig = cg + crel;
        int vcur = HU(H_V0);
#undef HU
        const unsigned rest = live_mask >> (q + 1);
        const int qn = rest ? q + 1 + __builtin_ctz(rest) : nq;                 // next alignment to walk
        const int *hn = hdr + qn * H_WORDS;                                     // (row nq exists: only its H_REL is meaningful)
        const int n_cig_n = qn < nq ? hn[H_WORDS + H_REL] - hn[H_REL] : 0;

        int ref_pos = start, q_pos = 0, fail_op = 0x7fffffff;
        if (l == 0) h[H_HIT0] = n_hit;
        for (int seg0 = 0; seg0 < n_cig && !overflow; seg0 += LPS_SEG) {
            const int nseg = min(LPS_SEG, n_cig - seg0);
            uint32_t wds[8]; uint32_t nextw; uint2 vr; int pv0;
            const bool have = pf_q == q && pf_seg == seg0;
            if (have) {
#pragma unroll
                for (int u = 0; u < 8; ++u) wds[u] = pw[u];
                finish_ops8(8 * l, nseg, wds);
                nextw = pnext;
            } else {
                nextw = (seg0 + nseg < n_cig) ? cig[seg0 + nseg] : 0xfu;      // op after the segment (0xf = none)
                load_ops8(cig + seg0, 8 * l, nseg, wds);
            }
            if (have && pf_vr_ok) { vr = pvr; pv0 = ppv; }
            else {
                vr = make_uint2(0x7fffffffu, 0u);
                if (vcur + l < V.n) vr = V.rec[vcur + l];
                pv0 = (vcur > 0 && vcur < V.n) ? V.pos[vcur - 1] : -1;
            }
            int my_ref;
            const unsigned seen = stage_ops8(wds, l, ref_pos, q_pos, sref, sqry, scig, my_ref);
            if (__ballot((seen & LPS_OPS_BAD) != 0u) && l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_BAD_CIGAR);   // the reference exits (:1625-1628)
            (void)my_ref;
            // getClip (:1613-1620,1636-1645): soft/hard clips longer than 5; FRONT iff CIGAR index 0.  Events wait in LDS for the wave's one reservation.
            if (__ballot((seen & LPS_OPS_CLIP) != 0u)) {                            // rare (first / last segment of a clipped alignment): words re-read from LDS
                int mine_n = 0;
#pragma unroll 1
                for (int k = 0; k < 8; ++k) { const uint32_t wd = scig[8 * l + k]; const unsigned op = wd & 15u; mine_n += ((op == 4u || op == 5u) && (wd >> 4) > 5u) ? 1 : 0; }
                const int incl = wave_incl_scan_dpp(mine_n);
                int slot = n_clip + incl - mine_n;
                if (mine_n) {
#pragma unroll 1
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t wd = scig[8 * l + k]; const unsigned op = wd & 15u;
                        if ((op == 4u || op == 5u) && (wd >> 4) > 5u) {
                            const int oi = seg0 + 8 * l + k;
                            if (slot < EXT_CLIPS) clipb[slot] = ClipEv{sref[8 * l + k], (oi << 1) | (oi != 0), r};
                            ++slot;
                        }
                    }
                }
                n_clip += __shfl(incl, 63);
                if (n_clip > EXT_CLIPS) overflow = true;                             // more clip ops than the buffer holds: the redo kernel takes the wave
            }
            if (l == 0) scig[nseg] = nextw;
            wave_sync();
            // ---- request the next segment's CIGAR words: same alignment, or the first segment of the next one
            const bool same = seg0 + LPS_SEG < n_cig;
            const bool has_next = same || (qn < nq && n_cig_n > 0);
            if (has_next) {
                const uint32_t *cign = same ? cig : cg + hn[H_REL];
                const int segn = same ? seg0 + LPS_SEG : 0, ncn = same ? n_cig : n_cig_n, nsegn = min(LPS_SEG, ncn - segn);
                pnext = (segn + nsegn < ncn) ? cign[segn + nsegn] : 0xfu;
                request_ops8(cign + segn, 8 * l, nsegn, pw);
                pf_q = same ? q : qn; pf_seg = segn;
            } else pf_q = -1;
            pf_vr_ok = false;
            // candidates are position-sorted: those before the end of this segment's reference interval form a prefix of the chunk
            bool first_round = true;
            while (true) {
                const int v = vcur + l;
                const int p = (int)vr.x;
                const bool mine = v < V.n && p < ref_pos;
                const int n_in = __popcll(__ballot(mine));
                const bool more = n_in == 64;
                if (first_round && has_next && !more) {          // ... and its candidate records + predecessor position
                    const int nv = same ? vcur + n_in : hn[H_V0];
                    pvr = make_uint2(0x7fffffffu, 0u);
                    if (nv + l < V.n) pvr = V.rec[nv + l];
                    ppv = (nv > 0 && nv < V.n) ? V.pos[nv - 1] : -1;
                    pf_vr_ok = true;
                }
                int pprev = __shfl_up(p, 1);                     // position of the previous variant (all lanes take part)
                if (l == 0) pprev = first_round ? pv0 : ((v > 0 && v < V.n) ? V.pos[v - 1] : -1);
                first_round = false;
                bool is_hit = false, fail = false; unsigned h0 = 0; int h1 = 0, opi = 0;
                if (mine) {
                    const unsigned at = vr.y;
                    // number of staged ops that start at or before p, by a fixed-trip search without branches: all LPS_SEG entries are valid numbers
                    // (the entries past the segment's ops hold its end position, which is beyond every candidate)
                    int lo = 0;
#pragma unroll
                    for (int step = LPS_SEG / 2; step >= 1; step >>= 1) lo += (sref[lo + step - 1] <= p) ? step : 0;
                    lo += (sref[lo] <= p) ? 1 : 0;               // lo <= LPS_SEG - 1 before this probe
                    const int j = lo - 1;
                    if (j >= 0) {
                        const uint32_t wd = scig[j];
                        const int op = wd & 15, len = (int)(wd >> 4);
                        const int rs = sref[j], qs = sqry[j];
                        opi = seg0 + j;
                        if (p < rs + len) {
                            const unsigned kind = VREC_KIND(at);
                            if (op_is_match(op)) {                                            // :1445-1520
                                const int off = p - rs;
                                if (qs + off + 1 > lq) fail = true;                           // :1453-1455
                                else if (kind == 0) { is_hit = true; h1 = qs + off; }         // base and quality are fetched when the hits are resolved
                                else if ((kind == 1 || kind == 2) && opi + 1 < n_cig) {       // indel variant :1470-1510
                                    const int want = (kind == 1) ? 1 : 2;                      // next op must be I resp. D
                                    const bool alt = rs + len - 1 == p && (int)(scig[j + 1] & 15) == want;
                                    is_hit = true; h0 = HIT_FINAL | (alt ? HIT_ALLELE : 0u) | ((at & VREC_DANGER) ? HIT_Q5 : 0u);
                                }
                            } else if (op == 2) {                                             // :1539-1607
                                // only the first variant at/after the deletion start is examined by the reference
                                const bool first_in = (v == 0) || pprev < rs;
                                if (first_in && (at & VREC_HPOLY3)) {
                                    if (qs + 1 > lq) fail = true;                             // :1559-1561
                                    else if (kind == 0) { is_hit = true; h1 = qs; }
                                    else if (kind == 2) { is_hit = true; h0 = HIT_FINAL | HIT_ALLELE; }
                                }
                            }
                        }
                    }
                    if (fail) fail_op = min(fail_op, opi);
                    if (is_hit) h0 |= (unsigned)v | ((at & VREC_ERASED) ? HIT_ERASED : 0u);
                }
                const unsigned long long hm = __ballot(is_hit);
                const int n_h = __popcll(hm);
                if (n_hit + n_h > EXT_CAP) {
                    if (n_hit + n_h > EXT_CAP + EXT_OVF) { overflow = true; break; }
                    if (!ovf_mine) {
                        unsigned ch = 0; if (l == 0) ch = atomicAdd(ovf_ctr, 1u);
                        ch = __shfl((int)ch, 0);
                        if (ch >= ovf_chunks) { overflow = true; break; }
                        ovf_mine = ovf + (size_t)ch * EXT_OVF;
                    }
                }
                if (is_hit) { const int at = n_hit + __popcll(hm & lanemask_lt()); const uint2 hv = make_uint2(h0, (unsigned)h1); if (at < EXT_CAP) hit[at] = hv; else ovf_mine[at - EXT_CAP] = hv; }
                n_hit += n_h;
                vcur += n_in;
                if (!more) break;
                vr = make_uint2(0x7fffffffu, 0u);
                if (vcur + l < V.n) vr = V.rec[vcur + l];
            }
            wave_sync();
        }
        fail_op = wave_min(fail_op);
        if (l == 0) h[H_FAIL] = fail_op;
        q = qn;
    }
    if (overflow) {
        // the hits (or clip events) of these four alignments do not fit the wave's LDS lists: k_extract_redo walks them with the direct-to-memory path
        if (l == 0) redo_list[atomicAdd(n_redo, 1u)] = (uint32_t)job;
        return;
    }
    wave_sync();

    // ---- resolve the hits, 64 at a time with every lane busy: gather base + quality, call the allele, drop what filterSNP erased, compact in place
    int hs[5]; int rfail[4];                                            // first hit of every row (rows that were not walked are empty), early-return op
    {
        int nxt = n_hit;
#pragma unroll
        for (int k = 3; k >= 0; --k) { const bool walked = k < nq && ((live_mask >> k) & 1u); hs[k] = walked ? hdr[k * H_WORDS + H_HIT0] : nxt; nxt = hs[k]; rfail[k] = walked ? hdr[k * H_WORDS + H_FAIL] : 0x7fffffff; }
        hs[4] = n_hit;
    }
    unsigned long long sb[4], qb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int *hk = hdr + k * H_WORDS;
        sb[k] = (unsigned long long)(unsigned)hk[H_SOFF] | ((unsigned long long)(unsigned)hk[H_SOFF + 1] << 32);
        qb[k] = (unsigned long long)(unsigned)hk[H_QOFF] | ((unsigned long long)(unsigned)hk[H_QOFF + 1] << 32);
    }
    // ---- ONE reservation for the rows of the wave: as many slots as there are hits (the few hits that turn out not to be observations - a base that is
    //      neither allele, a variant filterSNP erased - leave their slots unused), so that the resolved records go straight to their place
    unsigned long long off = 0; bool arena_full = false;
    if (n_hit > 0) {
        if (l == 0) off = atomicAdd(&O.arena_ctr[arena * 8], (unsigned long long)n_hit);
        off = __shfl(off, 0);
        if (off + (unsigned long long)n_hit > O.arena_size) arena_full = true;
    }
    const unsigned long long g0 = arena_lo + off;
    ObsRec *dst = O.rec + g0;
    int n_out = 0, n_emit[4] = {0, 0, 0, 0}; unsigned any_pre = 0;
    for (int i0 = 0; i0 < n_hit; i0 += 64) {
        const int i = i0 + l;
        const bool in = i < n_hit;
        uint2 hv = make_uint2(0u, 0u);
        if (in) hv = i < EXT_CAP ? hit[i] : ovf_mine[i - EXT_CAP];
        const int rq = (i >= hs[1]) + (i >= hs[2]) + (i >= hs[3]);       // row of the hit
        const unsigned long long so = rq == 0 ? sb[0] : (rq == 1 ? sb[1] : (rq == 2 ? sb[2] : sb[3]));
        const unsigned long long qo = rq == 0 ? qb[0] : (rq == 1 ? qb[1] : (rq == 2 ? qb[2] : qb[3]));
        const int rf = rq == 0 ? rfail[0] : (rq == 1 ? rfail[1] : (rq == 2 ? rfail[2] : rfail[3]));
        const int v = (int)(hv.x & 0x3fffffu);
        int allele = -1, qv = 0;
        if (in) {
            if (hv.x & HIT_FINAL) { allele = (hv.x & HIT_ALLELE) ? 1 : 0; qv = (hv.x & HIT_Q5) ? -5 : -4; }
            else {
                const int qi = (int)hv.y;
                const unsigned at = V.rec[v].y;
                const char ref_c = (char)(at & 0xff), alt_c = (char)((at >> 8) & 0xff);
                const char base_c = nt16_char(R.seq[so + (unsigned)(qi >> 1)] >> ((~qi & 1) << 2));
                if (base_c == ref_c) allele = 0; else if (base_c == alt_c) allele = 1;
                qv = R.qual[qo + (unsigned)qi];
            }
        }
        const bool pre = in && allele != -1 && rf == 0x7fffffff;          // an observation before filterSNP (rows that returned early hold none)
        const bool ok = pre && !(hv.x & HIT_ERASED);
        const unsigned long long pm = __ballot(pre), om = __ballot(ok);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = max(hs[k] - i0, 0), b = min(hs[k + 1] - i0, 64);   // lanes of row k in this round
            if (b > a) {
                const unsigned long long rm = ((b >= 64) ? ~0ull : ((1ull << b) - 1ull)) & ~((1ull << a) - 1ull);
                n_emit[k] += __popcll(om & rm); if (pm & rm) any_pre |= 1u << k;
            }
        }
        if (ok && !arena_full) dst[n_out + __popcll(om & lanemask_lt())] = ObsRec{(int32_t)v, (uint32_t)pack_aq(allele, qv)};
        n_out += __popcll(om);
    }
    if (arena_full && l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_OBS_OVERFLOW);          // the host grows the arenas and reruns
    if (l < nq) {
        int before = 0, mine = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (k < l) before += n_emit[k]; if (k == l) mine = n_emit[k]; }
        const int rf = l == 0 ? rfail[0] : (l == 1 ? rfail[1] : (l == 2 ? rfail[2] : rfail[3]));
        const bool walked = (live_mask >> l) & 1u;
        const bool ok = walked && !arena_full;
        RowDesc d;
        d.off = ok ? (uint32_t)(g0 + (unsigned)before) : 0u;
        d.cnt = ok ? mine : 0;
        d.fail = ok ? rf : 0x7fffffff;
        d.flags = (ok && rf == 0x7fffffff && ((any_pre >> l) & 1u) && mine == 0) ? 1u : 0u;
        O.rows[r0 + l] = d;
    }
    if (n_clip > 0 && !arena_full) {
        unsigned cb = 0;
        if (l == 0) cb = atomicAdd(C.n_ev, (unsigned)n_clip);
        cb = __shfl(cb, 0);
        if (l < n_clip && cb + (unsigned)l < C.capacity) C.ev[cb + l] = clipb[l];
    }
}

#define REDO_CAP 512    // observations buffered per wave of the redo kernel
// The alignments of waves whose hits did not fit k_extract_phase's LDS list (long reads over very dense variants), one job (four alignments) per
// wave: observations are called as their segment is searched and collected in an LDS buffer; when that fills up the wave reserves an upper bound
// for the row it is in - remaining reference span -> remaining candidates -, empties the buffer and writes the rest of that row directly.
__global__ __launch_bounds__(256, 4) void k_extract_redo(VarView V, ReadView R, ObsView O, ClipView C, int mapping_quality,
                                                      LpsCounters *cnt, const uint32_t *redo_list, const unsigned *n_redo) {
    __shared__ __attribute__((aligned(16))) int s_ref[4][LPS_SEG];
    __shared__ __attribute__((aligned(16))) int s_qry[4][LPS_SEG];
    __shared__ __attribute__((aligned(16))) uint32_t s_cig[4][LPS_SEG + 4];
    __shared__ int s_bvar[4][REDO_CAP];
    __shared__ uint16_t s_baq[4][REDO_CAP];
    enum { H_START, H_LQ, H_REL, H_V0, H_SOFF, H_QOFF = H_SOFF + 2, H_KIND = H_QOFF + 2, H_ROFF, H_RCNT, H_RFAIL, H_RFLAGS, H_WORDS };
    enum { ROW_DEAD = 0, ROW_BUFFERED = 1, ROW_GLOBAL = 2 };
    __shared__ int s_hdr[4][EXT_RPW + 1][H_WORDS];
    const int w = threadIdx.x >> 6, l = lane_id();
    int *sref = s_ref[w], *sqry = s_qry[w]; uint32_t *scig = s_cig[w];
    int *bvar = s_bvar[w]; uint16_t *baq = s_baq[w];
    int *hdr = s_hdr[w][0];
    // Output rows are reserved on one of LPS_ARENAS counters (own cache line each).  Workgroups are dealt round-robin over the 8 XCDs, so
    // arena = blockIdx % 64 keeps each counter inside ONE XCD's L2.
    const unsigned n_jobs = *n_redo;
#pragma unroll 1
    for (unsigned idx = blockIdx.x * 4 + w; idx < n_jobs; idx += gridDim.x * 4) {
    const int job = (int)redo_list[idx];
    const int arena = (job / 4) % O.n_arenas;
    const unsigned long long arena_lo = (unsigned long long)arena * O.arena_size;
    const int r0 = job * EXT_RPW;
    if (r0 >= R.n) continue;
    const int nq = min(EXT_RPW, R.n - r0);

    // ---- plan: headers, alignment q in lane q.  direct_detect_alleles filters (:1282-1291) + region "chr:1-<lastSNPPos>" (:1273)
    int h_start = 0, h_lq = 0, h_rel = 0; bool h_live = false; unsigned long long h_coff = 0, h_soff = 0, h_qoff = 0;
    if (l <= nq) h_coff = R.cigar_off[r0 + l];
    if (l < nq) {
        const int r = r0 + l; h_start = R.ref_start[r]; h_lq = R.l_qseq[r]; h_soff = R.seq_off[r]; h_qoff = R.qual_off[r];
        const int flag = R.flag[r];
        h_live = !(R.mapq[r] < mapping_quality || (flag & 0x4) || (flag & 0x100) || (flag & 0x400) || h_start >= V.last_pos);
    }
    const unsigned live_mask = (unsigned)__ballot(h_live) & 15u;
    const unsigned long long c_lo = __shfl(h_coff, 0);
    if (l <= nq) h_rel = (int)(h_coff - c_lo);                         // op index of alignment q's first op inside the wave's CIGAR range
    const uint32_t *cg = R.cigar + c_lo;

    // what was requested ahead for segment (pf_q, pf_seg): CIGAR words, op after the segment, candidate records, predecessor position
    uint32_t pw[8]; uint32_t pnext = 0xfu; uint2 pvr = make_uint2(0x7fffffffu, 0u); int ppv = -1;
    int pf_q = -1, pf_seg = 0; bool pf_vr_ok = false;
#pragma unroll
    for (int u = 0; u < 8; ++u) pw[u] = 6u;
    int q = live_mask ? __builtin_ctz(live_mask) : nq;
    if (q < nq) {                                                      // first segment of the first alignment: on its way while the bounds are searched
        const int crel = __shfl(h_rel, q), ncq = __shfl(h_rel, q + 1) - crel;
        if (ncq > 0) {
            const int nsegn = min(LPS_SEG, ncq);
            pnext = (nsegn < ncq) ? cg[crel + nsegn] : 0xfu;
            request_ops8(cg + crel, 8 * l, nsegn, pw);
            pf_q = q; pf_seg = 0;
        }
    }
    // first candidate of each alignment: four lanes search the position-sorted table side by side
    int h_v0 = 0;
    if (l < 4 && h_live) h_v0 = lane_var_lower_bound(V, h_start);
    if (l <= 4) {
        int *h = s_hdr[w][l];
        h[H_REL] = h_rel;
        if (l < 4) {
            h[H_START] = h_start; h[H_LQ] = h_lq; h[H_V0] = h_v0;
            h[H_SOFF] = (int)(unsigned)h_soff; h[H_SOFF + 1] = (int)(unsigned)(h_soff >> 32);
            h[H_QOFF] = (int)(unsigned)h_qoff; h[H_QOFF + 1] = (int)(unsigned)(h_qoff >> 32);
            h[H_KIND] = ROW_DEAD; h[H_ROFF] = 0; h[H_RCNT] = 0; h[H_RFAIL] = 0x7fffffff; h[H_RFLAGS] = 0;
        }
    }
    wave_sync();

    // ---- walk: segments of LPS_SEG ops -> LDS (ref prefix, query prefix, raw op word); the candidate variants (one packed record per lane)
    //      search them
    int n_buf = 0;                     // observations in the LDS buffer (rows of this wave not yet in HBM)
    bool arena_full = false;
#pragma unroll 1
    while (q < nq) {
        const int r = r0 + q;
        int *h = hdr + q * H_WORDS;
#define HU(i) __builtin_amdgcn_readfirstlane(h[i])
        const int start = HU(H_START), lq = HU(H_LQ), crel = HU(H_REL), n_cig = HU(H_WORDS + H_REL) - crel;
        const uint32_t *cig = cg + crel;
        const uint8_t *seq = R.seq + ((unsigned long long)(unsigned)HU(H_SOFF) | ((unsigned long long)(unsigned)HU(H_SOFF + 1) << 32));
        const uint8_t *qual = R.qual + ((unsigned long long)(unsigned)HU(H_QOFF) | ((unsigned long long)(unsigned)HU(H_QOFF + 1) << 32));
        int vcur = HU(H_V0);
#undef HU
        const unsigned rest = live_mask >> (q + 1);
        const int qn = rest ? q + 1 + __builtin_ctz(rest) : nq;                 // next alignment to walk
        const int *hn = hdr + qn * H_WORDS;                                     // (row nq exists: only its H_REL is meaningful)
        const int n_cig_n = qn < nq ? hn[H_WORDS + H_REL] - hn[H_REL] : 0;

        int ref_pos = start, q_pos = 0, n_emit = 0, fail_op = 0x7fffffff;
        bool had_any = false, direct = false;
        const int row_start = n_buf;                                            // of this row inside the buffer
        unsigned long long direct_base = 0;
        for (int seg0 = 0; seg0 < n_cig; seg0 += LPS_SEG) {
            const int nseg = min(LPS_SEG, n_cig - seg0);
            uint32_t wds[8]; uint32_t nextw; uint2 vr; int pv0;
            const bool have = pf_q == q && pf_seg == seg0;
            if (have) {
#pragma unroll
                for (int u = 0; u < 8; ++u) wds[u] = pw[u];
                finish_ops8(8 * l, nseg, wds);
                nextw = pnext;
            } else {
                nextw = (seg0 + nseg < n_cig) ? cig[seg0 + nseg] : 0xfu;      // op after the segment (0xf = none)
                load_ops8(cig + seg0, 8 * l, nseg, wds);
            }
            if (have && pf_vr_ok) { vr = pvr; pv0 = ppv; }
            else {
                vr = make_uint2(0x7fffffffu, 0u);
                if (vcur + l < V.n) vr = V.rec[vcur + l];
                pv0 = (vcur > 0 && vcur < V.n) ? V.pos[vcur - 1] : -1;
            }
            int my_ref;
            const unsigned seen = stage_ops8(wds, l, ref_pos, q_pos, sref, sqry, scig, my_ref);
            if (__ballot((seen & LPS_OPS_BAD) != 0u) && l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_BAD_CIGAR);   // the reference exits (:1625-1628)
            (void)my_ref;
            // getClip (:1613-1620,1636-1645): soft/hard clips longer than 5; FRONT iff CIGAR index 0.
            if (__ballot((seen & LPS_OPS_CLIP) != 0u)) {                            // rare (first / last segment of a clipped alignment): words re-read from LDS
                int mine_n = 0;
#pragma unroll 1
                for (int k = 0; k < 8; ++k) { const uint32_t wd = scig[8 * l + k]; const unsigned op = wd & 15u; mine_n += ((op == 4u || op == 5u) && (wd >> 4) > 5u) ? 1 : 0; }
                if (mine_n) {
#pragma unroll 1
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t wd = scig[8 * l + k]; const unsigned op = wd & 15u;
                        if ((op == 4u || op == 5u) && (wd >> 4) > 5u) {
                            const int oi = seg0 + 8 * l + k;
                            const unsigned e = atomicAdd(C.n_ev, 1u);               // (rare path: one atomic per event)
                            if (e < C.capacity) C.ev[e] = ClipEv{sref[8 * l + k], (oi << 1) | (oi != 0), r};
                        }
                    }
                }
            }
            if (l == 0) scig[nseg] = nextw;
            wave_sync();
            // ---- request the next segment's CIGAR words: same alignment, or the first segment of the next one
            const bool same = seg0 + LPS_SEG < n_cig;
            const bool has_next = same || (qn < nq && n_cig_n > 0);
            if (has_next) {
                const uint32_t *cign = same ? cig : cg + hn[H_REL];
                const int segn = same ? seg0 + LPS_SEG : 0, ncn = same ? n_cig : n_cig_n, nsegn = min(LPS_SEG, ncn - segn);
                pnext = (segn + nsegn < ncn) ? cign[segn + nsegn] : 0xfu;
                request_ops8(cign + segn, 8 * l, nsegn, pw);
                pf_q = same ? q : qn; pf_seg = segn;
            } else pf_q = -1;
            pf_vr_ok = false;
            // candidates are position-sorted: those before the end of this segment's reference interval form a prefix of the chunk
            bool first_round = true;
            while (true) {
                const int v = vcur + l;
                const int p = (int)vr.x;
                const bool mine = v < V.n && p < ref_pos;
                const int n_in = __popcll(__ballot(mine));
                const bool more = n_in == 64;
                if (first_round && has_next && !more) {          // ... and its candidate records + predecessor position
                    const int nv = same ? vcur + n_in : hn[H_V0];
                    pvr = make_uint2(0x7fffffffu, 0u);
                    if (nv + l < V.n) pvr = V.rec[nv + l];
                    ppv = (nv > 0 && nv < V.n) ? V.pos[nv - 1] : -1;
                    pf_vr_ok = true;
                }
                int pprev = __shfl_up(p, 1);                     // position of the previous variant (all lanes take part)
                if (l == 0) pprev = first_round ? pv0 : ((v > 0 && v < V.n) ? V.pos[v - 1] : -1);
                first_round = false;
                bool emit = false, fail = false; int allele = -1, qv = 0, opi = 0;
                if (mine) {
                    const unsigned at = vr.y;
                    // number of staged ops that start at or before p, by a fixed-trip search without branches: all LPS_SEG entries are valid numbers
                    // (the entries past the segment's ops hold its end position, which is beyond every candidate)
                    int lo = 0;
#pragma unroll
                    for (int step = LPS_SEG / 2; step >= 1; step >>= 1) lo += (sref[lo + step - 1] <= p) ? step : 0;
                    lo += (sref[lo] <= p) ? 1 : 0;               // lo <= LPS_SEG - 1 before this probe
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
                const int n_em = __popcll(em);
                if (!direct && n_buf + n_em > REDO_CAP) {
                    // ---- buffer full: reserve what is buffered + an upper bound for the rest of this row (every candidate up to the reference end
                    //      of the alignment emits at most once), move the buffer out, write the rest of the row directly
                    long long rem = 0;
                    for (int c = seg0 + nseg + l; c < n_cig; c += 64) { const uint32_t wd = cig[c]; if (op_consumes_ref(wd & 15)) rem += wd >> 4; }
                    rem = wave_sum(rem);
                    long long endll = (long long)ref_pos + rem; if (endll > 0x7fffffff) endll = 0x7fffffff;
                    const int v1 = var_lower_bound(V, (int)endll);
                    const unsigned long long need = (unsigned long long)n_buf + (unsigned long long)max(0, v1 - vcur);
                    unsigned long long off = 0;
                    if (l == 0) off = atomicAdd(&O.arena_ctr[arena * 8], need);
                    off = __shfl(off, 0);
                    if (off + need > O.arena_size) arena_full = true;
                    else {
                        const unsigned long long g0 = arena_lo + off;
                        for (int i = l; i < n_buf; i += 64) O.rec[g0 + i] = ObsRec{bvar[i], (uint32_t)baq[i]};
                        if (l < q) { int *hp = hdr + l * H_WORDS; if (hp[H_KIND] == ROW_BUFFERED) { hp[H_KIND] = ROW_GLOBAL; hp[H_ROFF] = (int)(uint32_t)(g0 + (unsigned)hp[H_ROFF]); } }
                        direct_base = g0 + row_start;
                    }
                    wave_sync();
                    direct = true; n_buf = 0;
                }
                if (emit) {
                    const int rank = n_emit + __popcll(em & lanemask_lt());
                    if (direct) { if (!arena_full) O.rec[direct_base + rank] = ObsRec{v, (uint32_t)pack_aq(allele, qv)}; }
                    else { bvar[row_start + rank] = v; baq[row_start + rank] = pack_aq(allele, qv); }
                }
                n_emit += n_em;
                if (!direct) n_buf += n_em;
                vcur += n_in;
                if (!more) break;
                vr = make_uint2(0x7fffffffu, 0u);
                if (vcur + l < V.n) vr = V.rec[vcur + l];
            }
            wave_sync();
        }
        fail_op = wave_min(fail_op);
        const bool any = __ballot(had_any) != 0;
        if (l == 0) {
            const bool dropped = fail_op != 0x7fffffff;
            h[H_KIND] = direct ? ROW_GLOBAL : ROW_BUFFERED;
            h[H_ROFF] = direct ? (int)(uint32_t)direct_base : row_start;
            h[H_RCNT] = dropped ? 0 : n_emit;
            h[H_RFAIL] = fail_op;
            h[H_RFLAGS] = (!dropped && any && n_emit == 0) ? 1 : 0;
        }
        q = qn;
    }
    wave_sync();
    // ---- one reservation for the buffered rows of the wave, coalesced copy-out, row descriptors
    unsigned long long off = 0;
    if (n_buf > 0) {
        if (l == 0) off = atomicAdd(&O.arena_ctr[arena * 8], (unsigned long long)n_buf);
        off = __shfl(off, 0);
        if (off + (unsigned long long)n_buf > O.arena_size) arena_full = true;
    }
    const unsigned long long g0 = arena_lo + off;
    if (!arena_full) for (int i = l; i < n_buf; i += 64) O.rec[g0 + i] = ObsRec{bvar[i], (uint32_t)baq[i]};
    if (arena_full && l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_OBS_OVERFLOW);          // the host grows the arenas and reruns
    if (l < nq) {
        const int *hp = hdr + l * H_WORDS; const int r = r0 + l; const int kind = hp[H_KIND];
        const bool ok = kind != ROW_DEAD && !arena_full;
        RowDesc d;
        d.off = !ok ? 0u : (kind == ROW_BUFFERED ? (uint32_t)(g0 + (unsigned)hp[H_ROFF]) : (uint32_t)hp[H_ROFF]);
        d.cnt = ok ? hp[H_RCNT] : 0;
        d.fail = ok ? hp[H_RFAIL] : 0x7fffffff;
        d.flags = ok ? (uint32_t)hp[H_RFLAGS] : 0u;
        O.rows[r] = d;
    }
    wave_sync();                                                         // the LDS buffers are reused by the wave's next job
    }
}

void launch_extract_phase(const VarView &V, const ReadView &R, const ObsView &O, const ClipView &C,
                          int mapping_quality, LpsCounters *cnt, uint32_t *redo_list, unsigned *n_redo, uint2 *ovf, unsigned *ovf_ctr, unsigned ovf_chunks, hipStream_t s) {
    if (R.n == 0) return;
    const int n_jobs = (R.n + EXT_RPW - 1) / EXT_RPW;
    hipLaunchKernelGGL(k_extract_phase, dim3((n_jobs + EXT_WPB - 1) / EXT_WPB), dim3(64 * EXT_WPB), 0, s, V, R, O, C, mapping_quality, cnt, redo_list, n_redo, ovf, ovf_ctr, ovf_chunks);
    // waves whose hits did not fit their LDS list queued themselves (none with ordinary read lengths and variant densities): a small grid drains the queue
    hipLaunchKernelGGL(k_extract_redo, dim3(std::min(256, (n_jobs + 3) / 4)), dim3(256), 0, s, V, R, O, C, mapping_quality, cnt, redo_list, n_redo);
}
