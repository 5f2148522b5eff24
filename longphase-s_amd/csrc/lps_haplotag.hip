// lps_haplotag.hip — per-read haplotype scoring of `haplotag` (germline), gfx950.
//
// Replaces (reference file:line, relative to /root/reference/):
//   ChromosomeProcessor::processSingleChrom filter cascade   src/haplotag/HaplotagParsingBam.cpp:453-486
//   CigarParser::parsingCigar + IsAltIndel                    src/haplotag/HaplotagParsingBam.cpp:541-670
//   GermlineHaplotagStrategy::judgeSnpHap / judgeDeletionHap  src/haplotag/HaplotagStrategy.cpp:20-209
// The read-level decision (judgeReadHap :243-300, PQ = int(-10*log10(..))) is taken on the host from the integer
// counts this kernel returns, because it needs the host libm's log10.
//
// Same wave-per-alignment design as k_extract_phase (LDS-staged CIGAR prefixes, variants search the ops, one packed
// record per candidate), but there is no output reservation: a read reduces to two vote counts and its PS range.
#include "lps_kernels.h"

// SOMATIC = false: germline `haplotag`.  SOMATIC = true: tagging pass of `somatic_haplotag` over the merged normal+tumor table
// (SomaticHaplotagCigarParser, src/somatic_haplotag/SomaticHaplotagProcess.cpp:557-579; judgeSomaticSnpHap / judgeNormalSnpHap /
// SomaticHaplotagStrategy::judgeTumorOnlySnpHap, src/haplotag/HaplotagStrategy.cpp:315-435,653-668): deletions cast no vote,
// indel rows vote with the read's own allele, somatic calls count H3 bases and which germline haplotype they derive from.
// MODE 2/3: normal-BAM extraction pass of somatic_haplotag (ExtractNorDataCigarParser, src/somatic_haplotag/SomaticVarCaller.cpp:227-293):
//   2 = germline votes gated by MAPQ (low-MAPQ reads are NOT skipped in this pass) + per-site base counters by atomics + the read's
//       haplotype; 3 = re-walk that adds the read's haplotype to ReadHpCount of every tumor site it touched (:171-173).
#ifndef HAP_WPB
#define HAP_WPB 1      // waves per workgroup: they share nothing, one wave per workgroup frees its LDS as soon as that wave is done
#endif
template <int MODE>
__global__ __launch_bounds__(64 * HAP_WPB, 6) void k_haplotag_score(VarView V, ReadView R, HapOut H, int mapping_quality, int tag_supplementary,
                                                        LpsCounters *cnt) {
    __shared__ __attribute__((aligned(16))) int s_ref[HAP_WPB][LPS_SEG];
    __shared__ __attribute__((aligned(16))) int s_qry[HAP_WPB][LPS_SEG];
    __shared__ __attribute__((aligned(16))) uint32_t s_cig[HAP_WPB][LPS_SEG + 4];
    const int w = threadIdx.x >> 6, l = lane_id();
    // XCD-aware: neighbouring reads are scored inside one XCD, so that their 16-byte result records meet in ONE L2 and leave as whole lines (with the
    // plain mapping every 128-byte line of results was written back as eight partials: 62 bytes written per byte of payload)
    const int r = xcd_unit((int)blockIdx.x, (int)gridDim.x) * HAP_WPB + w;
    if (r >= R.n) return;
    int *sref = s_ref[w], *sqry = s_qry[w]; uint32_t *scig = s_cig[w];
    const int start = R.ref_start[r];
    const int flag = R.flag[r];
    // every field of the read's header in one round trip (the filter cascade below needs three of them; the others used to be a second trip)
    const uint64_t coff = R.cigar_off[r], coff_end = R.cigar_off[r + 1], soff = R.seq_off[r];
    const int lq = R.l_qseq[r];
    constexpr bool SOMATIC = MODE == 1;
    constexpr bool EXTRACT = MODE == 2 || MODE == 3;
    const bool mq_ok = R.mapq[r] >= mapping_quality;
    int status = 0;                                                   // filter cascade (:453-486)
    if (!EXTRACT && !mq_ok) status = 1;                               // extraction passes run with mappingQualityFilter == false
    else if (flag & 0x4) status = 2;
    else if (flag & 0x100) status = 3;
    else if ((flag & 0x800) && !tag_supplementary) status = 4;
    else if (V.n == 0) status = 5;
    else if (!(start <= V.last_pos)) status = 6;
    int h1 = 0, h2 = 0, h3 = 0, d1 = 0, d2 = 0, ps_lo = 0x7fffffff, ps_hi = (int)0x80000000;
    if (status == 0) {
        const int n_cig = (int)(coff_end - coff);
        const uint32_t *cig = R.cigar + coff;
        const uint8_t *seq = R.seq + soff;
        uint32_t pw[8];                                              // the NEXT segment's words, requested while the current one is searched
        request_ops8(cig, 8 * l, min(LPS_SEG, n_cig), pw);            // (the first segment's: on their way while the first candidate is searched)
        int vcur = var_lower_bound(V, start);
        int ref_pos = start, q_pos = 0;
        int judged_op = -1;                                              // EXTRACT: last D op whose germline vote has been cast
        const int my_hp = (MODE == 3) ? (int)H.read_hp[r] : 0;
        for (int seg0 = 0; seg0 < n_cig && vcur < V.n; seg0 += LPS_SEG) {
            const int nseg = min(LPS_SEG, n_cig - seg0);
            uint2 vr = make_uint2(0x7fffffffu, 0u);
            if (vcur + l < V.n) vr = V.rec[vcur + l];
            const uint32_t nextw = (seg0 + nseg < n_cig) ? cig[seg0 + nseg] : 0xfu;
            uint32_t wds[8];                                         // 8 consecutive ops per lane (lps_kernels.h)
#pragma unroll
            for (int k = 0; k < 8; ++k) wds[k] = pw[k];
            finish_ops8(8 * l, nseg, wds);
            if (seg0 + LPS_SEG < n_cig) request_ops8(cig + seg0 + LPS_SEG, 8 * l, min(LPS_SEG, n_cig - seg0 - LPS_SEG), pw);
            int my_ref;
            const bool bad = (stage_ops8(wds, l, ref_pos, q_pos, sref, sqry, scig, my_ref) & LPS_OPS_BAD) != 0u;
            if (__ballot(bad) && l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_BAD_CIGAR);
            if (l == 0) scig[nseg] = nextw;
            wave_sync();
            while (true) {
                const int v = vcur + l;
                const int p = (int)vr.x;
                const bool mine = v < V.n && p < ref_pos;               // inside the reference interval walked so far
                const int n_in = __popcll(__ballot(mine));
                int pprev = __shfl_up(p, 1);
                if (l == 0) pprev = (v > 0 && v < V.n) ? V.pos[v - 1] : -1;
                int del_key = -1;                                        // EXTRACT: D op index of a NORMAL row waiting for its once-per-op vote
                if (mine) {
                    const unsigned at = vr.y;
                    // ops starting at or before p: fixed-trip search without branches (entries past the segment's ops hold its end position)
                    int lo = 0;
#pragma unroll
                    for (int step = LPS_SEG / 2; step >= 1; step >>= 1) lo += (sref[lo + step - 1] <= p) ? step : 0;
                    lo += (sref[lo] <= p) ? 1 : 0;
                    const int j = lo - 1;
                    if (j >= 0) {
                        const uint32_t wd = scig[j];
                        const int op = wd & 15, len = (int)(wd >> 4);
                        const int rs = sref[j], qs = sqry[j];
                        if (p < rs + len) {
                            const unsigned kind = VREC_KIND(at);
                            const char ref_c = (char)(at & 0xff), alt_c = (char)((at >> 8) & 0xff);
                            const bool hp1alt = (at & VREC_HP1ALT) != 0;
                            int vote = -1;                              // 0: haplotype carrying REF, 1: haplotype carrying ALT
                            bool count_ps = false;
                            if (EXTRACT) {
                                const unsigned tk = VREC_TKIND(at);
                                int32_t *sc = H.site + (size_t)v * LPS_SITE_COUNTERS;
                                if (op_is_match(op)) {
                                    const int qi = qs + (p - rs);
                                    const char base_c = qi < lq ? nt16_char(seq[qi >> 1] >> ((~qi & 1) << 2)) : 'N';
                                    bool is_alt = false;
                                    const bool has_next = seg0 + j + 1 < n_cig;
                                    if (kind == 0) is_alt = base_c == alt_c;
                                    else if ((kind == 1 || kind == 2) && has_next)
                                        is_alt = (rs + len - 1 == p) && (int)(scig[j + 1] & 15) == ((kind == 1) ? 1 : 2);
                                    if (tk >= 1 && tk <= 3) {                                 // countBaseNucleotide (HaplotagParsingBam.cpp:682-719)
                                        if (MODE == 3) atomicAdd(&sc[LPS_SC_READHP_UNTAG + my_hp], 1);
                                        else {
                                            const int bi = base_c == 'A' ? LPS_SC_A : base_c == 'C' ? LPS_SC_C : base_c == 'G' ? LPS_SC_G : base_c == 'T' ? LPS_SC_T : LPS_SC_UNKNOWN;
                                            if (mq_ok) { atomicAdd(&sc[bi + (LPS_SC_MPQ_A - LPS_SC_A)], 1); if (is_alt) atomicAdd(&sc[LPS_SC_MPQ_ALT], 1); atomicAdd(&sc[LPS_SC_MPQ_DEPTH], 1); }
                                            atomicAdd(&sc[bi], 1);
                                            if (is_alt) { if (tk == 3) atomicAdd(&sc[LPS_SC_DEL], 1); atomicAdd(&sc[LPS_SC_ALT], 1); }
                                            atomicAdd(&sc[LPS_SC_DEPTH], 1);
                                        }
                                    }
                                    if (MODE == 2 && mq_ok && VREC_ROLE(at) == 0) {           // germline judgeSnpHap on the NORMAL row
                                        if (kind == 0) { if (base_c == ref_c) vote = 0; else if (base_c == alt_c) vote = 1; count_ps = vote >= 0; }
                                        else if ((kind == 1 || kind == 2) && has_next) { vote = (kind == 1) ? (is_alt ? 1 : 0) : (is_alt ? 0 : 1); count_ps = true; }
                                    }
                                } else if (op == 2) {
                                    if (tk != 0) {                                            // processDeletionOperation (:265-282)
                                        if (MODE == 3) atomicAdd(&sc[LPS_SC_READHP_UNTAG + my_hp], 1);
                                        else if (tk == 1) { atomicAdd(&sc[LPS_SC_DEL], 1); atomicAdd(&sc[LPS_SC_DEPTH], 1); }
                                        else if (tk == 3) { atomicAdd(&sc[LPS_SC_ALT], 1); atomicAdd(&sc[LPS_SC_DEL], 1); atomicAdd(&sc[LPS_SC_DEPTH], 1); }
                                    }
                                    if (MODE == 2 && mq_ok && VREC_ROLE(at) == 0) del_key = seg0 + j;   // vote resolved below, once per D op
                                }
                            } else
                            if (SOMATIC) {
                                if (op_is_match(op)) {
                                    const int qi = qs + (p - rs);
                                    const char base_c = qi < lq ? nt16_char(seq[qi >> 1] >> ((~qi & 1) << 2)) : 'N';
                                    bool is_alt = false;                                      // IsAltIndel (HaplotagParsingBam.cpp:650-670)
                                    if (kind == 0) is_alt = base_c == alt_c;
                                    else if ((kind == 1 || kind == 2) && seg0 + j + 1 < n_cig)
                                        is_alt = (rs + len - 1 == p) && (int)(scig[j + 1] & 15) == ((kind == 1) ? 1 : 2);
                                    const unsigned role = VREC_ROLE(at);
                                    if (role == 0) {                                          // judgeNormalSnpHap (:403-435)
                                        if (kind == 0) { if (base_c == ref_c || base_c == alt_c) { vote = is_alt; count_ps = true; } }
                                        else if (kind == 1 || kind == 2) { vote = is_alt; count_ps = true; }
                                    } else if (role == 1) {                                   // somatic call: H3 when the read shows ALT (:653-668)
                                        if ((kind == 0 || kind == 1 || kind == 2) && is_alt) { ++h3; const unsigned dv = VREC_DERIVE(at); if (dv == 1) ++d1; else if (dv == 2) ++d2; }
                                    }
                                }
                            } else
                            if (op_is_match(op)) {                                            // judgeSnpHap (:20-130)
                                if (kind == 0) {
                                    const int qi = qs + (p - rs);
                                    const char base_c = qi < lq ? nt16_char(seq[qi >> 1] >> ((~qi & 1) << 2)) : 'N';
                                    if (base_c == ref_c) vote = 0; else if (base_c == alt_c) vote = 1;
                                    count_ps = vote >= 0;
                                } else if ((kind == 1 || kind == 2) && seg0 + j + 1 < n_cig) {
                                    const int want = (kind == 1) ? 1 : 2;
                                    const bool has = (rs + len - 1 == p) && (int)(scig[j + 1] & 15) == want;
                                    // insertion: the read's allele; deletion: the reference votes for the LONG allele (:98-129)
                                    vote = (kind == 1) ? (has ? 1 : 0) : (has ? 0 : 1);
                                    count_ps = true;
                                }
                            } else if (op == 2) {                                             // judgeDeletionHap (:147-209), once per D op
                                const bool first_in = (v == 0) || pprev < rs;
                                if (first_in && (at & VREC_HPOLY3)) {
                                    if (kind == 0) {
                                        const char base_c = qs < lq ? nt16_char(seq[qs >> 1] >> ((~qs & 1) << 2)) : 'N';
                                        if (base_c == ref_c) vote = 0; else if (base_c == alt_c) vote = 1;
                                        count_ps = true;
                                    } else if (kind == 2) { vote = 0; count_ps = true; }
                                }
                            }
                            if (vote >= 0) { if ((vote == 1) == hp1alt) ++h1; else ++h2; }
                            if (count_ps) { const int ps = V.phase_set[v]; ps_lo = min(ps_lo, ps); ps_hi = max(ps_hi, ps); }
                        }
                    }
                }
                if (MODE == 2) {
                    // once per D op: the FIRST NORMAL row inside the deletion casts the judgeDeletionHap vote (:285-291)
                    unsigned long long todo = __ballot(del_key >= 0);
                    while (todo) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const int k0 = __shfl(del_key, leader);
                        const unsigned long long same = __ballot(del_key == k0);
                        if (l == leader && k0 != judged_op) {
                            const unsigned at = vr.y;
                            if (at & VREC_HPOLY3) {
                                const unsigned kind = VREC_KIND(at);
                                const int qs = sqry[k0 - seg0];
                                const bool hp1alt = (at & VREC_HP1ALT) != 0;
                                int vote = -1; bool cps = false;
                                if (kind == 0) {
                                    const char base_c = qs < lq ? nt16_char(seq[qs >> 1] >> ((~qs & 1) << 2)) : 'N';
                                    if (base_c == (char)(at & 0xff)) vote = 0; else if (base_c == (char)((at >> 8) & 0xff)) vote = 1;
                                    cps = true;
                                } else if (kind == 2) { vote = 0; cps = true; }
                                if (vote >= 0) { if ((vote == 1) == hp1alt) ++h1; else ++h2; }
                                if (cps) { const int ps = V.phase_set[v]; ps_lo = min(ps_lo, ps); ps_hi = max(ps_hi, ps); }
                            }
                        }
                        judged_op = k0;
                        todo &= ~same;
                    }
                }
                vcur += n_in;
                if (n_in < 64 || vcur >= V.n) break;
                vr = make_uint2(0x7fffffffu, 0u);
                if (vcur + l < V.n) vr = V.rec[vcur + l];
            }
            wave_sync();
        }
        h1 = wave_sum(h1); h2 = wave_sum(h2); ps_lo = wave_min(ps_lo); ps_hi = wave_max(ps_hi);
        if (SOMATIC) { h3 = wave_sum(h3); d1 = wave_sum(d1); d2 = wave_sum(d2); }
    }
    if (MODE == 3) return;
    if (MODE == 2) {
        if (l == 0) {                                                 // judgeReadHap (HaplotagStrategy.cpp:243-300) without the PQ
            double mn, mx; int hp = 0;
            if (h1 > h2) { mn = h2; mx = h1; } else { mn = h1; mx = h2; }
            if (!(mx / (mx + mn) < H.pct_thr)) { if (h1 > h2) hp = 1; if (h1 < h2) hp = 2; }
            if (ps_lo <= ps_hi && ps_lo != ps_hi) hp = 0;
            H.read_hp[r] = (uint8_t)(status == 0 ? hp : 255);        // 255: read not processed by the pass
        }
        return;
    }
    if (MODE == 0 && H.rec) {
        // judgeReadHap (HaplotagStrategy.cpp:243-300) here: the threshold is an IEEE double division like the host's; PQ = int(-10 log10(min / (max + min)))
        // comes from a table the host built with ITS libm for votes below 64, beyond that the host computes it (PQ 255)
        if (l == 0) {
            const bool any = ps_lo <= ps_hi; const unsigned nps = any ? (ps_lo == ps_hi ? 1u : 2u) : 0u;
            int a = h1, b = h2; unsigned hp = 0, pq = 0;
            if (status == 0) {
                if (H.votes1) { a += H.votes1[r]; b += H.votes2[r]; }                 // judgeSVHap (:220-226): after the CIGAR walk, before the decision
                double mn, mx;
                if (a > b) { mn = b; mx = a; } else { mn = a; mx = b; }
                if (!(mx / (mx + mn) < H.pct_thr)) { if (a > b) hp = 1; if (a < b) hp = 2; }
                if (mx == 0) pq = 0; else if (mn == 0) pq = 40;
                else if (a >= 0 && b >= 0 && a < 64 && b < 64) pq = (unsigned)H.pq_tab[(a < b ? a : b) * 64 + (a < b ? b : a)];
                else pq = 255;
                if (nps > 1) hp = 0;
            }
            H.rec[r] = make_uint4((unsigned)status | (nps << 8) | (hp << 16) | (pq << 24), (unsigned)a, (unsigned)b, (unsigned)(any ? ps_lo : 0));
        }
        return;
    }
    if (l == 0) {
        H.status[r] = (uint8_t)status; H.hp1[r] = h1; H.hp2[r] = h2;
        const bool any = ps_lo <= ps_hi;
        H.n_ps[r] = any ? (ps_lo == ps_hi ? 1 : 2) : 0;               // only "more than one" matters to judgeReadHap
        H.ps_min[r] = any ? ps_lo : 0;
        if (SOMATIC) { H.hp3[r] = h3; H.d1[r] = d1; H.d2[r] = d2; }
    }
}

void launch_haplotag(const VarView &V, const ReadView &R, const HapOut &H, int mapping_quality, int tag_supplementary,
                     int mode, LpsCounters *cnt, hipStream_t s) {
    if (R.n == 0) return;
    const dim3 g(round_up8((R.n + HAP_WPB - 1) / HAP_WPB)), b(64 * HAP_WPB);   // a multiple of 8: the XCD-aware unit mapping
    if (mode == 1) hipLaunchKernelGGL(k_haplotag_score<1>, g, b, 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
    else if (mode == 2) hipLaunchKernelGGL(k_haplotag_score<2>, g, b, 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
    else if (mode == 3) hipLaunchKernelGGL(k_haplotag_score<3>, g, b, 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
    else hipLaunchKernelGGL(k_haplotag_score<0>, g, b, 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
}
