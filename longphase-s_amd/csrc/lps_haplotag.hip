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
    const uint64_t soff = R.seq_off[r]; const unsigned cp0 = R.cp_off[r]; const int n_words = R.cp_n[r];
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
        const int n_cig = n_words;
        const uint32_t *cig = R.cigp + 8ull * cp0;
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

// ---- germline haplotag as a STREAM walk (the extraction's design, lps_extract.hip): a wave takes FOUR consecutive alignments, their CIGAR words
// (in lane-chunks, lps_reads.hip) are one stream taken 512 words per round (8 per lane), one pair of DPP scans gives every lane-chunk its stream coordinates (8 bytes to LDS), and
// when the words are through the phased variants under the four alignments are taken 64 at a time as one flattened list, every lane busy:
// chunk search, the chunk's words, an 8-step walk to the op that covers the variant, judgeSnpHap / judgeDeletionHap (HaplotagStrategy.cpp:20-209),
// votes and phase sets reduced per alignment with ballots.  judgeReadHap (:243-300) follows right there and the four 16-byte records of the job
// leave as ONE 64-byte line.  A third of the vector instructions of k_haplotag_score<0>, which it replaces for the germline pass.
#ifndef HTG_WAVES
#define HTG_WAVES 4
#endif
#ifndef HTG_TAB
#define HTG_TAB 1024
#endif
// MODE 0: germline haplotag; 1: the tagging pass of somatic_haplotag over the merged normal + tumor table; 2 / 3: the normal-BAM extraction pass of
// somatic_haplotag (votes gated by MAPQ + per-site base counters + the read's haplotype / ReadHpCount of the touched sites) - the rules of
// k_haplotag_score<MODE>, which stays the general walker for records this walk's arithmetic cannot take
template <int MODE>
__global__ __launch_bounds__(64, HTG_WAVES) void k_haplotag_stream(VarView V, ReadView R, HapOut H, int mapping_quality, int tag_supplementary, LpsCounters *cnt) {
    constexpr bool SOM = MODE == 1, EXT = MODE == 2 || MODE == 3;
    __shared__ __attribute__((aligned(16))) int2 s_tab[HTG_TAB + 1];
    __shared__ ExtHdr s_hdr[4];
    const int l = lane_id();
    const int r0 = xcd_unit((int)blockIdx.x, (int)gridDim.x) * 4;        // XCD-aware: neighbouring jobs write neighbouring result lines into one L2
    if (r0 >= R.n) return;
    const int nq = min(4, R.n - r0);
    // ---- plan: headers, alignment q in lane q; the filter cascade of processSingleChrom (HaplotagParsingBam.cpp:453-486)
    int h_start = 0, h_lq = 0, h_status = 0, h_v0 = 0, h_n = 0; unsigned h_cp = 0; unsigned long long h_soff = 0; bool h_mq = false; int h_myhp = 0;
    if (l <= nq) h_cp = R.cp_off[r0 + l];
    if (l < nq) {
        const int r = r0 + l; h_start = R.ref_start[r]; h_lq = R.l_qseq[r]; h_soff = R.seq_off[r]; h_v0 = V.n ? R.v0[r] : 0; h_n = R.cp_n[r];
        const int flag = R.flag[r];
        h_mq = R.mapq[r] >= mapping_quality;
        if (MODE == 3) h_myhp = (int)H.read_hp[r];
        if (!EXT && !h_mq) h_status = 1;                                  // (the extraction passes run with mappingQualityFilter == false)
        else if (flag & 0x4) h_status = 2;
        else if (flag & 0x100) h_status = 3;
        else if ((flag & 0x800) && !tag_supplementary) h_status = 4;
        else if (V.n == 0) h_status = 5;
        else if (!(h_start <= V.last_pos)) h_status = 6;
    }
    const bool h_live = l < nq && h_status == 0;
    bool bad_cigar = false;
    const unsigned live_mask = (unsigned)__ballot(h_live) & 15u;
    const unsigned mq_mask = (unsigned)__ballot(h_mq) & 15u;              // EXT: alignments whose votes count (MAPQ)
    int myhp[4] = {0, 0, 0, 0}, judged[4] = {-1, -1, -1, -1};            // EXT: the read's haplotype of pass 2 (MODE 3); last D op whose once-per-op vote has been cast (MODE 2)
    if (MODE == 3) {
#pragma unroll
        for (int q = 0; q < 4; ++q) myhp[q] = __builtin_amdgcn_readlane(h_myhp, q);
    }
    const int h_nch = (int)(__shfl_down(h_cp, 1) - h_cp);                 // chunks of alignment q (lanes < nq)
    int vh1[4] = {0, 0, 0, 0}, vh2[4] = {0, 0, 0, 0}, plo[4], phi[4];      // per alignment: votes, smallest / largest phase set seen (wave-uniform)
    int vh3[4] = {0, 0, 0, 0}, vd1[4] = {0, 0, 0, 0}, vd2[4] = {0, 0, 0, 0};   // SOM: H3 bases at somatic calls and which germline haplotype they derive from
#pragma unroll
    for (int q = 0; q < 4; ++q) { plo[q] = 0x7fffffff; phi[q] = (int)0x80000000; }
    unsigned todo = live_mask;
#pragma unroll 1
    while (todo) {
        // ---- the job's alignments in groups whose chunks fit the table together (nearly always one group of four); an alignment that alone does
        //      not fit is walked with one table entry per 1 << shift chunks (see k_extract_phase)
        const int qa = __builtin_ctz(todo);
        const unsigned c_lo = __shfl(h_cp, qa);
        int qb = qa; unsigned gm = 1u << qa;
        for (int q = qa + 1; q < nq; ++q) {
            if (!((todo >> q) & 1u)) continue;
            if (__shfl(h_cp, q + 1) - c_lo > (unsigned)HTG_TAB) break;
            gm |= 1u << q; qb = q;
        }
        todo &= ~gm;
        int shift = 0;
        { const unsigned n1 = __shfl(h_cp, qa + 1) - c_lo; while (((n1 + (1u << shift) - 1u) >> shift) > (unsigned)HTG_TAB) ++shift; }
        const bool fast = shift == 0;
        const bool h_in = l < 4 && ((gm >> l) & 1u);
        const bool h_walk = h_in && h_n > 0;
        const int h_c0 = (l <= nq) ? (int)(h_cp - c_lo) : 0;              // first chunk of alignment q inside the stream
        const uint32_t *cg = R.cigp + 8ull * c_lo;
        const int TC = __builtin_amdgcn_readlane(h_c0 + h_nch, qb);       // chunks of the stream
        auto request = [&](int cid, uint32_t (&w)[8]) __attribute__((always_inline)) {          // unconditional, clamped (see k_extract_phase)
            const uint32_t *p = cg + 8 * min(cid, max(TC - 1, 0));
            const uint4 a = *reinterpret_cast<const uint4 *>(p), b = *reinterpret_cast<const uint4 *>(p + 4);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        };
        // the first 64 candidate positions of each alignment: requested ahead of the walk, looked at after it
        int v0q[4], pp[4]; bool walkq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0q[q] = __builtin_amdgcn_readlane(h_v0, q); walkq[q] = (__ballot(h_walk) >> q) & 1ull; pp[q] = V.n ? V.pos[min(v0q[q] + l, V.n - 1)] : 0x7fffffff; }
        if (l < 4) {
            ExtHdr &h = s_hdr[l];
            h.crel = fast ? 8 * h_c0 : 0; h.ncig = h_walk ? h_n : 0; h.c0 = fast ? h_c0 : 0; h.nch = h_walk ? (int)(((unsigned)h_nch + (1u << shift) - 1u) >> shift) : 0;
            h.lq = h_lq; h.soff = h_soff;
        }
        // ---- walk: four rounds per trip, all four requested at its head (see k_extract_phase); it counts the words whose op the reference rejects
        int carry_r = 0, carry_q = 0; unsigned special = 0; uint32_t big = 0; bool absurd = false;
#pragma unroll 1
        for (int R0 = 0; R0 < TC; R0 += 256) {
            uint32_t w0[8], w1[8], w2[8], w3[8];
            request(R0 + l, w0); request(R0 + 64 + l, w1); request(R0 + 128 + l, w2); request(R0 + 192 + l, w3);
            stream_round<LPS_BADMASK2>(w0, R0 + l, TC, shift, s_tab, carry_r, carry_q, special, big);
            stream_round<LPS_BADMASK2>(w1, R0 + 64 + l, TC, shift, s_tab, carry_r, carry_q, special, big);
            stream_round<LPS_BADMASK2>(w2, R0 + 128 + l, TC, shift, s_tab, carry_r, carry_q, special, big);
            stream_round<LPS_BADMASK2>(w3, R0 + 192 + l, TC, shift, s_tab, carry_r, carry_q, special, big);
            absurd |= (unsigned)carry_r > 0x3fffffffu || (unsigned)carry_q > 0x3fffffffu;
            if (absurd) break;
        }
        if (fast && l == 0) s_tab[TC] = make_int2(carry_r, carry_q);      // where the stream ends: the end of its last alignment
        // stream coordinates beyond 2^30 or one op of 2^24 bases and more (the sums are 24-bit multiplies): reference spans no aligner produces
        if (absurd || __ballot(big >= 0x10000000u)) { if (l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_KEY_RANGE); break; }
        if (__ballot(special != 0u)) {                                    // an op code the reference rejects: in an alignment that is walked? (rare path: the stream again)
            bool bad = false;
            for (int cid = l; cid < TC; cid += 64) {
                bool inq = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) inq |= s_hdr[q].ncig > 0 && (!fast || (cid >= s_hdr[q].c0 && 8 * (cid - s_hdr[q].c0) < s_hdr[q].ncig));
                for (int k = 0; k < 8; ++k) bad |= inq && op_bit(LPS_BADMASK2, cg[8 * cid + k]) != 0u;
            }
            bad_cigar |= __ballot(bad) != 0ull;
        }
        wave_sync();
        // ---- alignment bounds in stream coordinates, candidates of each: phased variants [v0, first variant at or beyond its reference end)
        int b_sat = 0, b_qat = 0, b_rend = h_start;
        if (h_walk) {
            if (fast) { const int2 ts = s_tab[h_c0], te = s_tab[h_c0 + h_nch]; b_sat = ts.x; b_qat = ts.y; b_rend = h_start + te.x - ts.x; }
            else b_rend = h_start + carry_r;
        }
        int ncand[4], rend[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rend[q] = __builtin_amdgcn_readlane(b_rend, q);
            int n = __popcll(__ballot(walkq[q] && v0q[q] + l < V.n && pp[q] < rend[q]));
            if (n == 64) {
                for (;;) { int p2 = 0x7fffffff; if (v0q[q] + n + l < V.n) p2 = V.pos[v0q[q] + n + l]; const int m = __popcll(__ballot(p2 < rend[q])); n += m; if (m < 64) break; }
            }
            ncand[q] = n;
        }
        int cum[5]; cum[0] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) cum[q + 1] = cum[q] + ncand[q];
        const int T = cum[4];
        int vadj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) vadj[q] = v0q[q] - cum[q];
        if (l < 4) { ExtHdr &h = s_hdr[l]; h.vadj = SEL4(l, vadj); h.ds = b_sat - h_start; h.dq = b_qat; }
        int maxnch = l < 4 ? s_hdr[l].nch : 0;
        maxnch = max(max(__builtin_amdgcn_readlane(maxnch, 0), __builtin_amdgcn_readlane(maxnch, 1)), max(__builtin_amdgcn_readlane(maxnch, 2), __builtin_amdgcn_readlane(maxnch, 3)));
        wave_sync();
        const int step0 = maxnch > 1 ? 1 << (31 - __builtin_clz(maxnch - 1)) : 0;
        uint2 pvr = V.n ? V.rec[min(SELC(l, cum, vadj) + l, V.n - 1)] : make_uint2(0u, 0u);
#pragma unroll 1
        for (int i0 = 0; i0 < T; i0 += 64) {
            const int i = i0 + l;
            const bool in = i < T;
            int vote = -1, ps_v = 0; bool count_ps = false, hp1alt = false; int h3v = 0;      // h3v (SOM): 1 H3 base, 2 / 3 deriving from haplotype 1 / 2 as well
            int del_key = -1, del_qs = 0; unsigned del_at = 0u;             // MODE 2: (alignment, D op) of a NORMAL row waiting for the op's one vote
            bool pairf = false; int pair_v = 0, pair_r = 0;                 // MODE 2: a tumor row this alignment touches (its ReadHpCount is added from the pair list)
            const uint2 vr = pvr;
            pvr = V.rec[min(SELC(i + 64, cum, vadj) + i + 64, V.n - 1)];
            if (in) {
                const int q = (i >= cum[1]) + (i >= cum[2]) + (i >= cum[3]);
                const int4 ha = *reinterpret_cast<const int4 *>(&s_hdr[q].crel), hb = *reinterpret_cast<const int4 *>(&s_hdr[q].vadj);
                const int hcrel = ha.x, hncig = ha.y, hc0 = ha.z, hnch = ha.w, hlq = hb.y;
                const int v = hb.x + i;
                const int p = (int)vr.x; const unsigned at = vr.y;
                hp1alt = (at & VREC_HP1ALT) != 0;
                const int ps = p + hb.z;
                int co = 0;
                for (int step = step0; step >= 1; step >>= 1) { const int t = co + step; const int sv = s_tab[hc0 + min(t, hnch - 1)].x; co = (t < hnch && sv <= ps) ? t : co; }
                const int2 base = s_tab[hc0 + co];
                const int x0 = (8 * (hc0 + co)) << shift;
                int rr = base.x, qq = base.y, jx = x0, rs = base.x, qs = base.y; uint32_t wj = 6u, wn = 6u;
                auto walk8 = [&](const uint32_t (&w)[9], int xb) __attribute__((always_inline)) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bool le = rr <= ps;
                        jx = le ? xb + k : jx; rs = le ? rr : rs; qs = le ? qq : qs; wj = le ? w[k] : wj; wn = le ? w[k + 1] : wn;
                        const unsigned len = w[k] >> 4;                  // (below 2^24, see the walk)
                        rr += (int)__umul24(len, op_bit(LPS_RMASK2, w[k])); qq += (int)__umul24(len, op_bit(LPS_QMASK2, w[k]));
                    }
                };
                for (int u = 0; u < (1 << shift); ++u) {                  // (one trip unless the alignment is walked in LONG mode)
                    const uint32_t *cw = cg + x0 + 8 * u;
                    uint32_t w[9];
                    const uint4 a = *reinterpret_cast<const uint4 *>(cw), b = *reinterpret_cast<const uint4 *>(cw + 4);
                    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w; w[8] = cw[8];
                    walk8(w, x0 + 8 * u);
                    if (rr > ps || x0 + 8 * u + 8 >= hcrel + hncig) break;
                }
                const int op = wj & 15, len = (int)(wj >> 4);
                const int opi = jx - hcrel;
                qs -= hb.w;
                if (ps < rs + len) {
                    const unsigned kind = VREC_KIND(at);
                    const char ref_c = (char)(at & 0xff), alt_c = (char)((at >> 8) & 0xff);
                    const uint8_t *seq = R.seq + s_hdr[q].soff;
                    if (EXT) {
                        // ExtractNorDataCigarParser (SomaticVarCaller.cpp:227-293): countBaseNucleotide at the tumor rows, the germline vote at the
                        // normal sample's rows (MAPQ-gated), ReadHpCount of the touched sites in the second pass
                        const unsigned tk = VREC_TKIND(at);
                        const bool mq_ok = (mq_mask >> q) & 1u;
                        int32_t *sc = H.site + (size_t)v * LPS_SITE_COUNTERS;
                        if (op_is_match(op)) {
                            const int qi = qs + (ps - rs);
                            const char base_c = qi < hlq ? nt16_char(__builtin_nontemporal_load(seq + (qi >> 1)) >> ((~qi & 1) << 2)) : 'N';
                            bool is_alt = false;
                            const bool has_next = opi + 1 < hncig;
                            if (kind == 0) is_alt = base_c == alt_c;
                            else if ((kind == 1 || kind == 2) && has_next) is_alt = (rs + len - 1 == ps) && (int)(wn & 15u) == ((kind == 1) ? 1 : 2);
                            if (tk >= 1 && tk <= 3) {                                 // countBaseNucleotide (HaplotagParsingBam.cpp:682-719)
                                if (MODE == 2) { pairf = true; pair_v = v; pair_r = r0 + q; }
                                if (MODE == 3) atomicAdd(&sc[LPS_SC_READHP_UNTAG + SEL4(q, myhp)], 1);
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
                                if (MODE == 2) { pairf = true; pair_v = v; pair_r = r0 + q; }
                                if (MODE == 3) atomicAdd(&sc[LPS_SC_READHP_UNTAG + SEL4(q, myhp)], 1);
                                else if (tk == 1) { atomicAdd(&sc[LPS_SC_DEL], 1); atomicAdd(&sc[LPS_SC_DEPTH], 1); }
                                else if (tk == 3) { atomicAdd(&sc[LPS_SC_ALT], 1); atomicAdd(&sc[LPS_SC_DEL], 1); atomicAdd(&sc[LPS_SC_DEPTH], 1); }
                            }
                            if (MODE == 2 && mq_ok && VREC_ROLE(at) == 0) { del_key = (q << 28) | opi; del_qs = qs; del_at = at; }   // the vote is cast below, once per D op
                        }
                    } else
                    if (SOM) {
                        // SomaticHaplotagCigarParser (SomaticHaplotagProcess.cpp:557-579): deletions cast no vote; judgeNormalSnpHap (HaplotagStrategy.cpp:403-435)
                        // at the normal sample's rows, the tumor ALT at a somatic call is an H3 base (:653-668)
                        if (op_is_match(op)) {
                            const int qi = qs + (ps - rs);
                            const char base_c = qi < hlq ? nt16_char(__builtin_nontemporal_load(seq + (qi >> 1)) >> ((~qi & 1) << 2)) : 'N';
                            bool is_alt = false;                                      // IsAltIndel (HaplotagParsingBam.cpp:650-670)
                            if (kind == 0) is_alt = base_c == alt_c;
                            else if ((kind == 1 || kind == 2) && opi + 1 < hncig) is_alt = (rs + len - 1 == ps) && (int)(wn & 15u) == ((kind == 1) ? 1 : 2);
                            const unsigned role = VREC_ROLE(at);
                            if (role == 0) {
                                if (kind == 0) { if (base_c == ref_c || base_c == alt_c) { vote = is_alt; count_ps = true; } }
                                else if (kind == 1 || kind == 2) { vote = is_alt; count_ps = true; }
                            } else if (role == 1) {
                                if ((kind == 0 || kind == 1 || kind == 2) && is_alt) { const unsigned dv = VREC_DERIVE(at); h3v = dv == 1 ? 2 : (dv == 2 ? 3 : 1); }
                            }
                        }
                    } else
                    if (op_is_match(op)) {                                        // judgeSnpHap (:20-130)
                        if (kind == 0) {
                            const int qi = qs + (ps - rs);
                            const char base_c = qi < hlq ? nt16_char(__builtin_nontemporal_load(seq + (qi >> 1)) >> ((~qi & 1) << 2)) : 'N';   // (a line touched once per launch: non-temporal)
                            if (base_c == ref_c) vote = 0; else if (base_c == alt_c) vote = 1;
                            count_ps = vote >= 0;
                        } else if ((kind == 1 || kind == 2) && opi + 1 < hncig) {
                            const int want = (kind == 1) ? 1 : 2;
                            const bool has = (rs + len - 1 == ps) && (int)(wn & 15u) == want;
                            vote = (kind == 1) ? (has ? 1 : 0) : (has ? 0 : 1);   // insertion: the read's allele; deletion: the reference votes for the LONG allele (:98-129)
                            count_ps = true;
                        }
                    } else if (op == 2) {                                         // judgeDeletionHap (:147-209), once per D op
                        const bool first_in = (v == 0) || V.pos[v - 1] + hb.z < rs;
                        if (first_in && (at & VREC_HPOLY3)) {
                            if (kind == 0) {
                                const char base_c = qs < hlq ? nt16_char(__builtin_nontemporal_load(seq + (qs >> 1)) >> ((~qs & 1) << 2)) : 'N';
                                if (base_c == ref_c) vote = 0; else if (base_c == alt_c) vote = 1;
                                count_ps = true;
                            } else if (kind == 2) { vote = 0; count_ps = true; }
                        }
                    }
                    if (count_ps) ps_v = V.phase_set[v];
                }
            }
            // ---- per alignment: votes by ballot; the phase-set range by two masked reductions, taken only when the alignment's variants do not all
            //      carry ONE phase set (they nearly always do: blocks are long)
            if (MODE == 2 && H.pair_ctr) {                                // the touched tumor rows of this round: one reservation in the job's arena
                const unsigned long long pm = __ballot(pairf);
                if (pm) {
                    const int arena = (int)(blockIdx.x % LPS_TARENAS);
                    unsigned long long pb = 0;
                    if (l == 0) pb = atomicAdd(&H.pair_ctr[arena * 16], (unsigned long long)__popcll(pm));
                    pb = __shfl(pb, 0);
                    if (pairf) {
                        const long long idx = (long long)pb + __popcll(pm & lanemask_lt());
                        if (idx < H.pair_arena) { const long long slot = (long long)arena * H.pair_arena + idx; H.apair_site[slot] = pair_v; H.apair_read[slot] = pair_r; }
                    }
                }
            }
            if (MODE == 2) {
                // once per D op: the FIRST normal row inside the deletion (the lowest lane: candidates are in position order) casts the judgeDeletionHap
                // vote (:285-291), unless the op's vote was cast in an earlier round of this alignment
                unsigned long long dtodo = __ballot(del_key >= 0);
                while (dtodo) {
                    const int leader = __builtin_ctzll(dtodo);
                    const int k0 = __builtin_amdgcn_readlane(del_key, leader);
                    const unsigned long long same = __ballot(del_key == k0);
                    const int q0 = k0 >> 28;
                    if (l == leader && k0 != SEL4(q0, judged) && (del_at & VREC_HPOLY3)) {
                        const unsigned kind = VREC_KIND(del_at);
                        if (kind == 0) {
                            const uint8_t *seq = R.seq + s_hdr[q0].soff; const int hlq = s_hdr[q0].lq;
                            const char base_c = del_qs < hlq ? nt16_char(__builtin_nontemporal_load(seq + (del_qs >> 1)) >> ((~del_qs & 1) << 2)) : 'N';
                            if (base_c == (char)(del_at & 0xff)) vote = 0; else if (base_c == (char)((del_at >> 8) & 0xff)) vote = 1;
                            count_ps = true;
                        } else if (kind == 2) { vote = 0; count_ps = true; }
                        if (count_ps) ps_v = V.phase_set[SELC(i, cum, vadj) + i];
                    }
                    judged[0] = q0 == 0 ? k0 : judged[0]; judged[1] = q0 == 1 ? k0 : judged[1]; judged[2] = q0 == 2 ? k0 : judged[2]; judged[3] = q0 == 3 ? k0 : judged[3];
                    dtodo &= ~same;
                }
            }
            const bool to1 = vote >= 0 && ((vote == 1) == hp1alt), to2 = vote >= 0 && !((vote == 1) == hp1alt);
            const unsigned long long m1 = __ballot(to1), m2 = __ballot(to2), mp = __ballot(count_ps);
            const unsigned long long m3 = SOM ? __ballot(h3v != 0) : 0ull, md1 = SOM ? __ballot(h3v == 2) : 0ull, md2 = SOM ? __ballot(h3v == 3) : 0ull;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int a = max(cum[k] - i0, 0), b = min(cum[k + 1] - i0, 64);
                if (b > a) {
                    const unsigned long long rm = ((b >= 64) ? ~0ull : ((1ull << b) - 1ull)) & ~((1ull << a) - 1ull);
                    vh1[k] += __popcll(m1 & rm); vh2[k] += __popcll(m2 & rm);
                    if (SOM) { vh3[k] += __popcll(m3 & rm); vd1[k] += __popcll(md1 & rm); vd2[k] += __popcll(md2 & rm); }
                    const unsigned long long pk = mp & rm;
                    if (pk) {
                        const int first = __builtin_amdgcn_readlane(ps_v, __builtin_ctzll(pk));
                        if (__ballot(count_ps && ps_v != first) & rm) {
                            const bool mine = (pk >> l) & 1ull;
                            plo[k] = min(plo[k], wave_min(mine ? ps_v : 0x7fffffff)); phi[k] = max(phi[k], wave_max(mine ? ps_v : (int)0x80000000));
                        } else { plo[k] = min(plo[k], first); phi[k] = max(phi[k], first); }
                    }
                }
            }
        }
        wave_sync();                                                      // the table and the headers are reused by the next group
    }
    if (bad_cigar && l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_BAD_CIGAR);
    if (MODE == 3) return;
    if (MODE == 2) {                                                      // judgeReadHap (HaplotagStrategy.cpp:243-300) without the PQ: the read's haplotype for the second pass
        if (l < nq) {
            const int h1 = SEL4(l, vh1), h2 = SEL4(l, vh2), lo = SEL4(l, plo), hi = SEL4(l, phi);
            double mn, mx; int hp = 0;
            if (h1 > h2) { mn = h2; mx = h1; } else { mn = h1; mx = h2; }
            if (!(mx / (mx + mn) < H.pct_thr)) { if (h1 > h2) hp = 1; if (h1 < h2) hp = 2; }
            if (lo <= hi && lo != hi) hp = 0;
            H.read_hp[r0 + l] = (uint8_t)(h_status == 0 ? hp : 255);     // 255: read not processed by the pass
        }
        return;
    }
    if (SOM) {                                                            // the counts of the tagging pass: the caller applies judgeSomaticReadHap / inheritHaplotype
        if (l < nq) {
            const int r = r0 + l; const int lo = SEL4(l, plo), hi = SEL4(l, phi); const bool any = lo <= hi;
            H.status[r] = (uint8_t)h_status; H.hp1[r] = SEL4(l, vh1); H.hp2[r] = SEL4(l, vh2); H.n_ps[r] = any ? (lo == hi ? 1 : 2) : 0; H.ps_min[r] = any ? lo : 0;
            H.hp3[r] = SEL4(l, vh3); H.d1[r] = SEL4(l, vd1); H.d2[r] = SEL4(l, vd2);
        }
        return;
    }
    // ---- judgeReadHap (:243-300) for the four alignments, one lane each; ONE 64-byte line of results per job
    if (l < nq) {
        const int r = r0 + l;
        const int h1 = SEL4(l, vh1), h2 = SEL4(l, vh2), lo = SEL4(l, plo), hi = SEL4(l, phi);
        const bool any = lo <= hi; const unsigned nps = any ? (lo == hi ? 1u : 2u) : 0u;
        int a = h1, b = h2; unsigned hp = 0, pq = 0;
        if (h_status == 0) {
            if (H.votes1) { a += H.votes1[r]; b += H.votes2[r]; }                 // judgeSVHap (:220-226): after the CIGAR walk, before the decision
            double mn, mx;
            if (a > b) { mn = b; mx = a; } else { mn = a; mx = b; }
            if (!(mx / (mx + mn) < H.pct_thr)) { if (a > b) hp = 1; if (a < b) hp = 2; }
            if (mx == 0) pq = 0; else if (mn == 0) pq = 40;
            else if (a >= 0 && b >= 0 && a < 64 && b < 64) pq = (unsigned)H.pq_tab[(a < b ? a : b) * 64 + (a < b ? b : a)];
            else pq = 255;
            if (nps > 1) hp = 0;
        }
        H.rec[r] = make_uint4((unsigned)h_status | (nps << 8) | (hp << 16) | (pq << 24), (unsigned)a, (unsigned)b, (unsigned)(any ? lo : 0));
    }
}

// ReadHpCount of the tumor rows the normal sample's alignments touch (ExtractNorDataCigarParser's second pass, SomaticVarCaller.cpp:227-293): from the pair
// list of k_haplotag_stream<2>, one thread per pair, with the read haplotype that walk decided
__global__ __launch_bounds__(256) void k_normal_pair_sites(HapOut H) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x, n_slots = (long long)LPS_TARENAS * H.pair_arena;
    if (t >= n_slots) return;
    const long long arena = t / H.pair_arena, idx = t - arena * H.pair_arena;
    if ((unsigned long long)idx >= H.pair_ctr[arena * 16]) return;
    atomicAdd(&H.site[(size_t)H.apair_site[t] * LPS_SITE_COUNTERS + LPS_SC_READHP_UNTAG + (int)H.read_hp[H.apair_read[t]]], 1);
}
void launch_normal_pair_sites(const HapOut &H, hipStream_t s) {
    hipLaunchKernelGGL(k_normal_pair_sites, dim3((unsigned)(((long long)LPS_TARENAS * H.pair_arena + 255) / 256)), dim3(256), 0, s, H);
}
void launch_haplotag(const VarView &V, const ReadView &R, const HapOut &H, int mapping_quality, int tag_supplementary,
                     int mode, LpsCounters *cnt, hipStream_t s, bool general) {
    if (R.n == 0) return;
    if (mode == 0 && H.rec && !general) {                                 // germline haplotag: the stream walk, four alignments per wave
        hipLaunchKernelGGL(k_haplotag_stream<0>, dim3(round_up8((R.n + 3) / 4)), dim3(64), 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
        return;
    }
    if (mode >= 1 && mode <= 3 && !general) {                             // the somatic passes on the same walk (a record it cannot take sets LPS_ERR_KEY_RANGE: the caller runs the general walker)
        const dim3 gs(round_up8((R.n + 3) / 4)), bs(64);
        if (mode == 1) hipLaunchKernelGGL(k_haplotag_stream<1>, gs, bs, 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
        else if (mode == 2) hipLaunchKernelGGL(k_haplotag_stream<2>, gs, bs, 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
        else hipLaunchKernelGGL(k_haplotag_stream<3>, gs, bs, 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
        return;
    }
    const dim3 g(round_up8((R.n + HAP_WPB - 1) / HAP_WPB)), b(64 * HAP_WPB);   // a multiple of 8: the XCD-aware unit mapping
    if (mode == 1) hipLaunchKernelGGL(k_haplotag_score<1>, g, b, 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
    else if (mode == 2) hipLaunchKernelGGL(k_haplotag_score<2>, g, b, 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
    else if (mode == 3) hipLaunchKernelGGL(k_haplotag_score<3>, g, b, 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
    else hipLaunchKernelGGL(k_haplotag_score<0>, g, b, 0, s, V, R, H, mapping_quality, tag_supplementary, cnt);
}
