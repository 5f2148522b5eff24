// lps_somatic.hip — tumor-BAM extraction pass of `somatic_haplotag` (row a21), gfx950.
//
// Replaces (reference file:line, relative to /root/reference/):
//   ExtractTumDataChrProcessor::processRead / classifyReadsByCase    src/somatic_haplotag/SomaticVarCaller.cpp:334-518
//   ExtractTumDataCigarParser::processMatchOperation / ...Deletion   :712-759
//   getWindowsDiffRef / getOrderWindowsDiffRef / processCigarOperation :627-710
//   SomaticJudgeHapStrategy::judgeSomaticSnpHap / judgeNormalSnpHap  src/haplotag/HaplotagStrategy.cpp:315-435
//   ExtractSomaticDataStragtegy::judgeTumorOnlySnpHap                :617-638
//   SomaticJudgeHapStrategy::judgeSomaticReadHap (without PQ)        :452-602
//
// Two passes of the same wave-per-alignment walker (LDS-staged CIGAR prefixes, packed variant records):
//   PASS 0  votes (H1/H2 at NORMAL rows, H3 at tumor-only rows), per-site base counters + alleleCount by atomics, the list of hits whose +-100 bp
//           difference window has to be taken (k_tumor_windows: one thread per hit and direction), the read's haplotype and its per-read record;
//   PASS 1  everything that needs the read's haplotype: base.ReadHpCount, classifyReadsByCase counters, somaticReadHpCount and the
//           (site, read, base HP) pairs of tumorPosReadCorrBaseHP.
// All per-site quantities are order-free integer counts, so atomics reproduce the reference exactly.
#include "lps_kernels.h"
#include "lps_graph.h"


// processCigarOperation (:627-652).  The reference's own enum has CIGAR_N == 6 (HaplotagType.h:29).
__device__ __forceinline__ bool win_next_op(const uint32_t *cig, int &idx, int end, int dir, int &remaining, int &readPos, int &refPos, int &op) {
    idx += dir;
    while (idx < end && idx >= 0) {
        op = cig[idx] & 15; const int len = (int)(cig[idx] >> 4);
        // (additions of a possibly-zero amount: an `x += ...` on one path and a `y += ...` on another become one addition through a selected
        // pointer, and the variables then live in scratch memory)
        const bool aligned = op == 0 || op == 3 || op == 6 || op == 7 || op == 8;
        remaining += aligned ? len : 0; readPos += op == 1 ? len * dir : 0; refPos += op == 2 ? len * dir : 0;
        if (aligned) return true;
        if (op != 1 && op != 2) return false;
        idx += dir;
    }
    return false;
}

// getOrderWindowsDiffRef (:654-685): WRITE=false counts the differences, WRITE=true stores them from slot `base`.
// One THREAD walks a window (k_tumor_windows): its 64 neighbours in the wave walk 64 other reads, so every byte load of a step is 64 different lines
// and the working set of a CU's waves (two lines per lane) is far beyond its L1 - every step went to L2 for a line to use ONE byte of it (1.1 ms per
// launch at 160 Mb).  The walk therefore keeps the aligned 8 bytes around the last read base (16 bases) and reference base in registers and loads
// again only when it leaves them: a sixteenth / an eighth of the loads.  (Both arrays are DevBuf allocations: 256-byte aligned, 64 bytes of slack.)
struct ByteWindow {
    const uint8_t *p; uintptr_t at = ~(uintptr_t)0; unsigned long long w = 0;
    __device__ __forceinline__ explicit ByteWindow(const void *q) : p((const uint8_t *)q) {}
    __device__ __forceinline__ unsigned operator[](long long i) {
        const uintptr_t a = (uintptr_t)(p + i), b = a & ~(uintptr_t)7;
        if (b != at) { w = *reinterpret_cast<const unsigned long long *>(b); at = b; }
        return (unsigned)(w >> (8u * (unsigned)(a & 7))) & 0xffu;
    }
};
// The counting walk also leaves a MEMO of what it found - which of the 100 steps differ (a bit each) and the read's base code at the first eight of them -
// so that the writing pass stores a window of up to eight differences (nearly all of them) from 24 bytes instead of walking it again.
struct WinMemo { unsigned long long lo, hi; };                             // bits 0-63 / 64-99: step i differs (bit i - 1); hi bit 63: the memo cannot stand for the walk
#define WIN_MEMO_MAX 8
template <bool WRITE>
__device__ __forceinline__ int win_dir(const uint32_t *cig, int idx, int n_cig, const uint8_t *seq, int readLen, const char *ref, int refLen,
                                       int readPos, int remaining, int refPos, int dir, const TumOut &T, long long base, int site, int allele,
                                       WinMemo *memo = nullptr, uint32_t *codes = nullptr) {
    int op = cig[idx] & 15, n = 0;
    ByteWindow sq(seq), rf(ref);
    unsigned long long mlo = 0, mhi = 0; uint32_t cd = 0; bool plain = true;
    auto leave = [&]() __attribute__((always_inline)) { if (!WRITE) { memo->lo = mlo; memo->hi = mhi | ((plain && n <= WIN_MEMO_MAX) ? 0ull : (1ull << 63)); *codes = cd; } return n; };
    for (int i = 1; i <= 100; ++i) {
        remaining--;
        if (remaining == 0 || remaining == -1) { if (!win_next_op(cig, idx, n_cig, dir, remaining, readPos, refPos, op)) return leave(); }
        if (op == 2 || op == 1 || op == 3 || op == 6 || op == 8) continue;
        readPos += dir; refPos += dir;
        if (readPos > readLen || refPos > refLen || readPos < 0 || refPos < 0) return leave();
        const int code = readPos < readLen ? (int)((sq[readPos >> 1] >> ((~readPos & 1) << 2)) & 15u) : -1;
        const char rb = code >= 0 ? nt16_char(code) : '\0';
        const char fb = refPos < refLen ? (char)rf[refPos] : '\0';
        if (rb != fb) {
            if (WRITE && base + n < T.win_cap) { T.win_site[base + n] = site; T.win_allele[base + n] = (uint8_t)allele; T.win_offset[base + n] = (int16_t)(i * dir); T.win_base[base + n] = (uint8_t)rb; }
            if (!WRITE) { if (i <= 64) mlo |= 1ull << (i - 1); else mhi |= 1ull << (i - 65); plain = plain && code >= 0; if (n < WIN_MEMO_MAX) cd |= (uint32_t)(code & 15) << (4 * n); }
            ++n;
        }
    }
    return leave();
}

// judgeSomaticReadHap (HaplotagStrategy.cpp:452-602) restricted to the haplotype decision (hpCount[4] is never incremented here)
__device__ __forceinline__ int somatic_read_hp(int h1, int h2, int h3, bool multi_ps, double thr) {
    const int h4 = 0;
    double tMin, tMax, nMin, nMax; int maxT, maxN;
    if (h3 > h4) { tMin = h4; tMax = h3; maxT = 3; } else { tMin = h3; tMax = h4; maxT = 4; }
    if (h1 > h2) { nMin = h2; nMax = h1; maxN = 1; } else { nMin = h1; nMax = h2; maxN = 2; }
    const double tumSim = (tMax == 0) ? 0.0 : tMax / (tMax + tMin), norSim = (nMax == 0) ? 0.0 : nMax / (nMax + nMin);
    int hp = 0;
    if (tMax != 0) { if (tumSim >= thr) { if (norSim >= thr) hp = (maxT == 3) ? (maxN == 1 ? 5 : 7) : (maxN == 1 ? 6 : 8); else hp = (maxT == 3) ? 3 : 4; } }
    else if (nMax != 0) { if (norSim >= thr) hp = maxN; }
    if (multi_ps) hp = 0;
    return hp;
}

template <int PASS>
__global__ __launch_bounds__(64) void k_tumor_extract(VarView V, ReadView R, TumOut T, int mapping_quality, int tag_supplementary, LpsCounters *cnt) {
    __shared__ __attribute__((aligned(16))) int s_ref[1][LPS_SEG];
    __shared__ __attribute__((aligned(16))) int s_qry[1][LPS_SEG];
    __shared__ __attribute__((aligned(16))) uint32_t s_cig[1][LPS_SEG + 4];
    const int w = threadIdx.x >> 6, l = lane_id();
    const int r = blockIdx.x + w;                                     // one wave per workgroup: the waves share nothing, LDS is freed per wave
    if (r >= R.n) return;
    int *sref = s_ref[w], *sqry = s_qry[w]; uint32_t *scig = s_cig[w];
    const int start = R.ref_start[r];
    const int flag = R.flag[r];
    const bool mq_ok = R.mapq[r] >= mapping_quality;
    // every field of the read's header in one round trip (as in k_haplotag_score)
    const uint64_t soff = R.seq_off[r]; const unsigned cp0 = R.cp_off[r]; const int n_words = R.cp_n[r];
    const int lq = R.l_qseq[r];
    int status = 0;                                                    // mappingQualityFilter == false in the extraction passes
    if (flag & 0x4) status = 2;
    else if (flag & 0x100) status = 3;
    else if ((flag & 0x800) && !tag_supplementary) status = 4;
    else if (V.n == 0) status = 5;
    else if (!(start <= V.last_pos)) status = 6;
    int h1 = 0, h2 = 0, h3 = 0, ps_lo = 0x7fffffff, ps_hi = (int)0x80000000, n_site = 0;
    int ref_pos = start, q_pos = 0;
    bool walked = false;
    if (status == 0) {
        const int n_cig = n_words;
        const uint32_t *cig = R.cigp + 8ull * cp0;
        const uint8_t *seq = R.seq + soff;
        int vcur = var_lower_bound(V, start);
        walked = vcur < V.n;                                           // parsingCigar returns at once when no variant is left (:555-557)
        // read-level facts of PASS 0 that PASS 1 needs
        int r_hp = 0; bool r_record = true, r_clean = false; int r_h1 = 0, r_h2 = 0;
        if (PASS == 1) { r_hp = T.hp[r]; r_h1 = T.hp1[r]; r_h2 = T.hp2[r]; r_record = T.n_ps[r] <= 1; r_clean = (r_h1 == 0 || r_h2 == 0) && T.hp3[r] != 0; }
        for (int seg0 = 0; seg0 < n_cig && walked; seg0 += LPS_SEG) {
            const int nseg = min(LPS_SEG, n_cig - seg0);
            uint2 vr = make_uint2(0x7fffffffu, 0u);
            if (vcur + l < V.n) vr = V.rec[vcur + l];
            const uint32_t nextw = (seg0 + nseg < n_cig) ? cig[seg0 + nseg] : 0xfu;
            uint32_t wds[8];                                         // 8 consecutive ops per lane (lps_kernels.h)
            load_ops8(cig + seg0, 8 * l, nseg, wds);
            int my_ref;
            const bool bad = (stage_ops8(wds, l, ref_pos, q_pos, sref, sqry, scig, my_ref) & LPS_OPS_BAD) != 0u;
            if (__ballot(bad) && l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_BAD_CIGAR);
            if (l == 0) scig[nseg] = nextw;
            wave_sync();
            while (vcur < V.n) {
                const int v = vcur + l;
                const int p = (int)vr.x;
                const bool mine = v < V.n && p < ref_pos;
                const int n_in = __popcll(__ballot(mine));
                bool want_win = false, pair = false; int win_allele = 0, opj = 0, win_off = 0, base_hp = 0;
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
                            const unsigned kind = VREC_KIND(at), tk = VREC_TKIND(at), role = VREC_ROLE(at);
                            const char ref_c = (char)(at & 0xff), alt_c = (char)((at >> 8) & 0xff);
                            const bool hp1alt = (at & VREC_HP1ALT) != 0;
                            int32_t *sc = T.site + (size_t)v * LPS_TSITE_COUNTERS;
                            if (op_is_match(op)) {
                                const int qi = qs + (p - rs);
                                const char base_c = qi < lq ? nt16_char(seq[qi >> 1] >> ((~qi & 1) << 2)) : 'N';
                                bool is_alt = false;
                                if (kind == 0) is_alt = base_c == alt_c;
                                else if ((kind == 1 || kind == 2) && seg0 + j + 1 < n_cig)
                                    is_alt = (rs + len - 1 == p) && (int)(scig[j + 1] & 15) == ((kind == 1) ? 1 : 2);
                                if (mq_ok) {                                                  // judgeSomaticSnpHap (:315-389)
                                    if (role == 0) {
                                        bool counted = false;
                                        if (kind == 0) counted = base_c == ref_c || base_c == alt_c;
                                        else if (kind == 1 || kind == 2) counted = true;         // base := isAlt ? Alt : Ref
                                        if (counted) {
                                            // (0/1 amounts, not `++h1` on one path and `++h2` on the other: the compiler turns that into one increment through a
                                            // selected pointer and keeps the counters in scratch memory)
                                            const bool to1 = hp1alt == is_alt; h1 += to1 ? 1 : 0; h2 += to1 ? 0 : 1; base_hp = to1 ? 1 : 2;
                                            const int ps = V.phase_set[v]; ps_lo = min(ps_lo, ps); ps_hi = max(ps_hi, ps);
                                        }
                                    } else if (tk != 0) {                                     // tumor-only row: H3 when the read shows the tumor ALT
                                        const bool to3 = (kind == 0 || kind == 1 || kind == 2) && is_alt; h3 += to3 ? 1 : 0; base_hp = to3 ? 3 : base_hp;
                                    }
                                    pair = pair || tk != 0; n_site += tk != 0 ? 1 : 0;          // tumorSnpPosVec (:722-724)
                                }
                                if (PASS == 0 && tk >= 1 && tk <= 3) {                        // :728-741
                                    if (tk != 1 || base_c == ref_c || base_c == alt_c) {
                                        atomicAdd(&sc[39 + (is_alt ? 1 : 0)], 1);
                                        want_win = true; win_allele = is_alt; opj = seg0 + j; win_off = p - rs;
                                    }
                                    const int bi = base_c == 'A' ? LPS_SC_A : base_c == 'C' ? LPS_SC_C : base_c == 'G' ? LPS_SC_G : base_c == 'T' ? LPS_SC_T : LPS_SC_UNKNOWN;
                                    if (mq_ok) { atomicAdd(&sc[bi + (LPS_SC_MPQ_A - LPS_SC_A)], 1); if (is_alt) atomicAdd(&sc[LPS_SC_MPQ_ALT], 1); atomicAdd(&sc[LPS_SC_MPQ_DEPTH], 1); }
                                    atomicAdd(&sc[bi], 1);
                                    if (is_alt) { if (tk == 3) atomicAdd(&sc[LPS_SC_DEL], 1); atomicAdd(&sc[LPS_SC_ALT], 1); }
                                    atomicAdd(&sc[LPS_SC_DEPTH], 1);
                                }
                                if (PASS == 1 && pair) {
                                    atomicAdd(&sc[15 + r_hp], 1);                             // base.ReadHpCount[hpResult] (:457)
                                    if (base_hp == 3) {                                       // classifyReadsByCase (:462-518) + somaticReadHpCount (:386-404)
                                        if (!r_record) atomicAdd(&sc[24], 1);
                                        else if (r_clean) {
                                            atomicAdd(&sc[25], 1);
                                            if (r_h1 == 0 && r_h2 == 0) atomicAdd(&sc[28], 1); else if (r_h1 != 0 && r_h2 == 0) atomicAdd(&sc[26],
                                                    1); else if (r_h1 == 0 && r_h2 != 0) atomicAdd(&sc[27], 1);
                                        } else atomicAdd(&sc[29], 1);
                                        atomicAdd(&sc[30 + r_hp], 1);
                                    }
                                }
                            } else if (op == 2 && PASS == 0) {                                // processDeletionOperation (:743-759)
                                if (tk == 1) { atomicAdd(&sc[LPS_SC_DEL], 1); atomicAdd(&sc[LPS_SC_DEPTH], 1); }
                                else if (tk == 3) { atomicAdd(&sc[LPS_SC_ALT], 1); atomicAdd(&sc[LPS_SC_DEL], 1); atomicAdd(&sc[LPS_SC_DEPTH], 1); }
                            }
                        }
                    }
                }
                if (PASS == 0) {
                    // getWindowsDiffRef (:687-710) walks up to 100 bases to either side of the site, one dependent pair of loads (read base, reference
                    // base) per step: inside this kernel ONE lane did that while 63 waited - 21.7 of the pass's 22 ms at 160 Mb.  The hit is listed
                    // instead (one reservation per round of the wave) and k_win_count / k_win_write take the walks one THREAD per (hit, direction).
                    const unsigned long long wm = __ballot(want_win);
                    if (wm) {
                        const int arena = (int)(blockIdx.x % LPS_TARENAS);
                        unsigned long long hb = 0;
                        if (l == 0) hb = atomicAdd(&T.hit_ctr[arena * 16], (unsigned long long)__popcll(wm));
                        hb = __shfl(hb, 0);
                        if (want_win) {
                            const long long idx = (long long)hb + __popcll(wm & lanemask_lt());
                            if (idx < T.hit_arena) { const long long slot = (long long)arena * T.hit_arena + idx;
                                T.hits[slot] = make_int4(v, r, opj, win_off | (win_allele << 30)); T.hit_rp[slot] = sqry[opj - seg0] + win_off; }
                        }
                    }
                } else {
                    const unsigned long long pm = __ballot(pair);
                    if (pm) {
                        const int arena = (int)(blockIdx.x % LPS_TARENAS);
                        unsigned long long pb = 0;
                        if (l == 0) pb = atomicAdd(&T.pair_ctr[arena * 16], (unsigned long long)__popcll(pm));
                        pb = __shfl(pb, 0);
                        if (pair) {
                            const long long idx = (long long)pb + __popcll(pm & lanemask_lt());
                            if (idx < T.pair_arena) { const long long slot = (long long)arena * T.pair_arena + idx;
                                T.apair_site[slot] = v; T.apair_read[slot] = r; T.apair_hp[slot] = (uint8_t)base_hp; }
                        }
                    }
                }
                vcur += n_in;
                if (n_in < 64 || vcur >= V.n) break;
                vr = make_uint2(0x7fffffffu, 0u);
                if (vcur + l < V.n) vr = V.rec[vcur + l];
            }
            wave_sync();
        }
        if (PASS == 0) { h1 = wave_sum(h1); h2 = wave_sum(h2); h3 = wave_sum(h3); ps_lo = wave_min(ps_lo); ps_hi = wave_max(ps_hi); n_site = wave_sum(n_site); }
    }
    if (PASS == 0 && l == 0) {
        const bool any = ps_lo <= ps_hi;
        T.status[r] = (uint8_t)status; T.hp1[r] = h1; T.hp2[r] = h2; T.hp3[r] = h3;
        T.n_ps[r] = any ? (ps_lo == ps_hi ? 1 : 2) : 0; T.ps_min[r] = any ? ps_lo : 0;
        T.hp[r] = (uint8_t)(status == 0 ? somatic_read_hp(h1, h2, h3, any && ps_lo != ps_hi, T.pct_thr) : 0);
        T.end_pos[r] = (status == 0 && walked) ? ref_pos : (status == 0 ? start : 0);
        T.read_len[r] = (status == 0 && walked) ? q_pos : 0;
        T.has_site[r] = n_site > 0;
    }
}

// ---- the two tumor passes as a STREAM walk (the design of k_extract_phase / k_haplotag_stream): a wave takes FOUR consecutive alignments, their CIGAR
// words are one stream (lane-chunks, lps_reads.hip), one pair of DPP scans per round gives every chunk its stream coordinates, and the rows of the merged
// table under the four alignments are taken 64 at a time as one flattened list, every lane busy: chunk search, the chunk's words, the 8-step walk to the
// op that covers the row, then exactly the rules of k_tumor_extract<0> above (which, with <1>, stays the general walker for records this walk's arithmetic
// cannot take: LPS_ERR_KEY_RANGE).  Votes, phase sets and the "met a tumor row" flag are reduced per alignment with ballots; one list reservation per
// ROUND of 64 candidates of four alignments instead of one per alignment and round.  ONE walk: it lists the (tumor row, alignment) pairs as well, and
// what the reference's second loop over the alignments adds at a row (k_tumor_extract<1>) is done from that list by k_tumor_pair_sites.
#ifndef TUM_TAB
#define TUM_TAB 1024
#endif
__global__ __launch_bounds__(64, 4) void k_tumor_stream(VarView V, ReadView R, TumOut T, int mapping_quality, int tag_supplementary, LpsCounters *cnt) {
    __shared__ __attribute__((aligned(16))) int2 s_tab[TUM_TAB + 1];
    __shared__ ExtHdr s_hdr[4];
    const int l = lane_id();
    const int r0 = (int)blockIdx.x * 4;
    if (r0 >= R.n) return;
    const int nq = min(4, R.n - r0);
    const int arena = (int)(blockIdx.x % LPS_TARENAS);
    // ---- plan: headers, alignment q in lane q (mappingQualityFilter == false in the extraction passes: MAPQ only gates the votes)
    int h_start = 0, h_lq = 0, h_status = 0, h_v0 = 0, h_n = 0; unsigned h_cp = 0; unsigned long long h_soff = 0; bool h_mq = false;
    if (l <= nq) h_cp = R.cp_off[r0 + l];
    if (l < nq) {
        const int r = r0 + l; h_start = R.ref_start[r]; h_lq = R.l_qseq[r]; h_soff = R.seq_off[r]; h_v0 = V.n ? R.v0[r] : 0; h_n = R.cp_n[r];
        const int flag = R.flag[r];
        h_mq = R.mapq[r] >= mapping_quality;
        if (flag & 0x4) h_status = 2;
        else if (flag & 0x100) h_status = 3;
        else if ((flag & 0x800) && !tag_supplementary) h_status = 4;
        else if (V.n == 0) h_status = 5;
        else if (!(h_start <= V.last_pos)) h_status = 6;
    }
    const bool h_walked = l < nq && h_status == 0 && h_v0 < V.n;         // parsingCigar returns at once when no variant is left (:555-557)
    bool bad_cigar = false;
    const unsigned live_mask = (unsigned)__ballot(h_walked) & 15u;
    const unsigned mq_mask = (unsigned)__ballot(h_mq) & 15u;
    const int h_nch = (int)(__shfl_down(h_cp, 1) - h_cp);
    int vh1[4] = {0, 0, 0, 0}, vh2[4] = {0, 0, 0, 0}, vh3[4] = {0, 0, 0, 0}, nsite[4] = {0, 0, 0, 0}, plo[4], phi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { plo[q] = 0x7fffffff; phi[q] = (int)0x80000000; }
    int e_end = h_start, e_len = 0;                                       // lane q: where alignment q ends on the reference, bases of the read its CIGAR consumes
    unsigned todo = live_mask;
#pragma unroll 1
    while (todo) {
        const int qa = __builtin_ctz(todo);
        const unsigned c_lo = __shfl(h_cp, qa);
        int qb = qa; unsigned gm = 1u << qa;
        for (int q = qa + 1; q < nq; ++q) {
            if (!((todo >> q) & 1u)) continue;
            if (__shfl(h_cp, q + 1) - c_lo > (unsigned)TUM_TAB) break;
            gm |= 1u << q; qb = q;
        }
        todo &= ~gm;
        int shift = 0;
        { const unsigned n1 = __shfl(h_cp, qa + 1) - c_lo; while (((n1 + (1u << shift) - 1u) >> shift) > (unsigned)TUM_TAB) ++shift; }
        const bool fast = shift == 0;
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
        int v0q[4], pp[4]; bool walkq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0q[q] = __builtin_amdgcn_readlane(h_v0, q); walkq[q] = (__ballot(h_walk) >> q) & 1ull; pp[q] = V.pos[min(v0q[q] + l, V.n - 1)]; }
        if (l < 4) {
            ExtHdr &h = s_hdr[l];
            h.crel = fast ? 8 * h_c0 : 0; h.ncig = h_walk ? h_n : 0; h.c0 = fast ? h_c0 : 0; h.nch = h_walk ? (int)(((unsigned)h_nch + (1u << shift) - 1u) >> shift) : 0;
            h.lq = h_lq; h.soff = h_soff;
        }
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
        if (fast && l == 0) s_tab[TC] = make_int2(carry_r, carry_q);
        if (absurd || __ballot(big >= 0x10000000u)) { if (l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_KEY_RANGE); break; }   // the host runs the pass again on k_tumor_extract
        if (__ballot(special != 0u)) {                                    // an op code the reference rejects: in an alignment that is walked?
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
        int b_sat = 0, b_qat = 0, b_rend = h_start;
        if (h_walk) {
            if (fast) { const int2 ts = s_tab[h_c0], te = s_tab[h_c0 + h_nch]; b_sat = ts.x; b_qat = ts.y; b_rend = h_start + te.x - ts.x; e_len = te.y - ts.y; }
            else { b_rend = h_start + carry_r; e_len = carry_q; }
            e_end = b_rend;
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
        const int TT = cum[4];
        int vadj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) vadj[q] = v0q[q] - cum[q];
        if (l < 4) { ExtHdr &h = s_hdr[l]; h.vadj = SEL4(l, vadj); h.ds = b_sat - h_start; h.dq = b_qat; }
        int maxnch = l < 4 ? s_hdr[l].nch : 0;
        maxnch = max(max(__builtin_amdgcn_readlane(maxnch, 0), __builtin_amdgcn_readlane(maxnch, 1)), max(__builtin_amdgcn_readlane(maxnch, 2), __builtin_amdgcn_readlane(maxnch, 3)));
        wave_sync();
        const int step0 = maxnch > 1 ? 1 << (31 - __builtin_clz(maxnch - 1)) : 0;
        uint2 pvr = V.rec[min(SELC(l, cum, vadj) + l, V.n - 1)];
#pragma unroll 1
        for (int i0 = 0; i0 < TT; i0 += 64) {
            const int i = i0 + l;
            const bool in = i < TT;
            bool to1 = false, to2 = false, to3 = false, count_ps = false, pairf = false, want_win = false;
            int ps_v = 0, base_hp = 0, win_allele = 0, win_off = 0, opi = 0, hit_q = 0, v = 0, q = 0;
            const uint2 vr = pvr;
            pvr = V.rec[min(SELC(i + 64, cum, vadj) + i + 64, V.n - 1)];
            if (in) {
                q = (i >= cum[1]) + (i >= cum[2]) + (i >= cum[3]);
                const int4 ha = *reinterpret_cast<const int4 *>(&s_hdr[q].crel), hb = *reinterpret_cast<const int4 *>(&s_hdr[q].vadj);
                const int hcrel = ha.x, hncig = ha.y, hc0 = ha.z, hnch = ha.w, hlq = hb.y;
                v = hb.x + i;
                const int p = (int)vr.x; const unsigned at = vr.y;
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
                        const unsigned len = w[k] >> 4;
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
                opi = jx - hcrel;
                qs -= hb.w;
                if (ps < rs + len) {
                    const unsigned kind = VREC_KIND(at), tk = VREC_TKIND(at), role = VREC_ROLE(at);
                    const char ref_c = (char)(at & 0xff), alt_c = (char)((at >> 8) & 0xff);
                    const bool hp1alt = (at & VREC_HP1ALT) != 0, mq_ok = (mq_mask >> q) & 1u;
                    int32_t *sc = T.site + (size_t)v * LPS_TSITE_COUNTERS;
                    if (op_is_match(op)) {
                        const int qi = qs + (ps - rs);
                        const uint8_t *seq = R.seq + s_hdr[q].soff;
                        const char base_c = qi < hlq ? nt16_char(__builtin_nontemporal_load(seq + (qi >> 1)) >> ((~qi & 1) << 2)) : 'N';
                        bool is_alt = false;
                        if (kind == 0) is_alt = base_c == alt_c;
                        else if ((kind == 1 || kind == 2) && opi + 1 < hncig) is_alt = (rs + len - 1 == ps) && (int)(wn & 15u) == ((kind == 1) ? 1 : 2);
                        if (mq_ok) {                                                  // judgeSomaticSnpHap (:315-389)
                            if (role == 0) {
                                bool counted = false;
                                if (kind == 0) counted = base_c == ref_c || base_c == alt_c;
                                else if (kind == 1 || kind == 2) counted = true;         // base := isAlt ? Alt : Ref
                                if (counted) { const bool h1v = hp1alt == is_alt; to1 = h1v; to2 = !h1v; base_hp = h1v ? 1 : 2; count_ps = true; ps_v = V.phase_set[v]; }
                            } else if (tk != 0) {                                     // tumor-only row: H3 when the read shows the tumor ALT
                                to3 = (kind == 0 || kind == 1 || kind == 2) && is_alt; base_hp = to3 ? 3 : base_hp;
                            }
                            pairf = tk != 0;                                          // tumorSnpPosVec (:722-724)
                        }
                        if (tk >= 1 && tk <= 3) {                                     // :728-741
                            if (tk != 1 || base_c == ref_c || base_c == alt_c) {
                                atomicAdd(&sc[39 + (is_alt ? 1 : 0)], 1);
                                want_win = true; win_allele = is_alt; win_off = ps - rs; hit_q = qs + win_off;
                            }
                            const int bi = base_c == 'A' ? LPS_SC_A : base_c == 'C' ? LPS_SC_C : base_c == 'G' ? LPS_SC_G : base_c == 'T' ? LPS_SC_T : LPS_SC_UNKNOWN;
                            if (mq_ok) { atomicAdd(&sc[bi + (LPS_SC_MPQ_A - LPS_SC_A)], 1); if (is_alt) atomicAdd(&sc[LPS_SC_MPQ_ALT], 1); atomicAdd(&sc[LPS_SC_MPQ_DEPTH], 1); }
                            atomicAdd(&sc[bi], 1);
                            if (is_alt) { if (tk == 3) atomicAdd(&sc[LPS_SC_DEL], 1); atomicAdd(&sc[LPS_SC_ALT], 1); }
                            atomicAdd(&sc[LPS_SC_DEPTH], 1);
                        }
                    } else if (op == 2) {                                             // processDeletionOperation (:743-759)
                        if (tk == 1) { atomicAdd(&sc[LPS_SC_DEL], 1); atomicAdd(&sc[LPS_SC_DEPTH], 1); }
                        else if (tk == 3) { atomicAdd(&sc[LPS_SC_ALT], 1); atomicAdd(&sc[LPS_SC_DEL], 1); atomicAdd(&sc[LPS_SC_DEPTH], 1); }
                    }
                }
            }
            {                                                             // the hit is listed; k_tumor_windows walks the +-100 bases one thread per (hit, direction)
                const unsigned long long wm = __ballot(want_win);
                if (wm) {
                    unsigned long long hb = 0;
                    if (l == 0) hb = atomicAdd(&T.hit_ctr[arena * 16], (unsigned long long)__popcll(wm));
                    hb = __shfl(hb, 0);
                    if (want_win) {
                        const long long idx = (long long)hb + __popcll(wm & lanemask_lt());
                        if (idx < T.hit_arena) { const long long slot = (long long)arena * T.hit_arena + idx;
                            T.hits[slot] = make_int4(v, r0 + q, opi, win_off | (win_allele << 30)); T.hit_rp[slot] = hit_q; }
                    }
                }
                // ... and so is the (tumor row, alignment, base haplotype) pair: what the reference's second loop over the alignments does at the row
                // needs the read's FINAL haplotype and nothing else of the walk - k_tumor_pair_sites takes it from this list, no second walk
                const unsigned long long pm = __ballot(pairf);
                if (pm) {
                    unsigned long long pb = 0;
                    if (l == 0) pb = atomicAdd(&T.pair_ctr[arena * 16], (unsigned long long)__popcll(pm));
                    pb = __shfl(pb, 0);
                    if (pairf) {
                        const long long idx = (long long)pb + __popcll(pm & lanemask_lt());
                        if (idx < T.pair_arena) { const long long slot = (long long)arena * T.pair_arena + idx;
                            T.apair_site[slot] = v; T.apair_read[slot] = r0 + q; T.apair_hp[slot] = (uint8_t)base_hp; }
                    }
                }
            }
            {
                const unsigned long long m1 = __ballot(to1), m2 = __ballot(to2), m3 = __ballot(to3), mp = __ballot(count_ps), ms = __ballot(pairf);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int a = max(cum[k] - i0, 0), b = min(cum[k + 1] - i0, 64);
                    if (b > a) {
                        const unsigned long long rm = ((b >= 64) ? ~0ull : ((1ull << b) - 1ull)) & ~((1ull << a) - 1ull);
                        vh1[k] += __popcll(m1 & rm); vh2[k] += __popcll(m2 & rm); vh3[k] += __popcll(m3 & rm); nsite[k] += __popcll(ms & rm);
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
        }
        wave_sync();                                                      // the table and the headers are reused by the next group
    }
    if (bad_cigar && l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_BAD_CIGAR);
    if (l < nq) {
        const int r = r0 + l;
        const int h1 = SEL4(l, vh1), h2 = SEL4(l, vh2), h3 = SEL4(l, vh3), lo = SEL4(l, plo), hi = SEL4(l, phi);
        const bool any = lo <= hi;
        T.status[r] = (uint8_t)h_status; T.hp1[r] = h1; T.hp2[r] = h2; T.hp3[r] = h3;
        T.n_ps[r] = any ? (lo == hi ? 1 : 2) : 0; T.ps_min[r] = any ? lo : 0;
        T.hp[r] = (uint8_t)(h_status == 0 ? somatic_read_hp(h1, h2, h3, any && lo != hi, T.pct_thr) : 0);
        T.end_pos[r] = h_walked ? e_end : (h_status == 0 ? h_start : 0);
        T.read_len[r] = h_walked ? e_len : 0;
        T.has_site[r] = SEL4(l, nsite) > 0;
    }
}

// the second loop of the reference over the alignments (SomaticVarCaller.cpp:440-518), from the pair list of the stream walk: one thread per listed
// (tumor row, alignment) - ReadHpCount of the row by the read's final haplotype, and for an H3 base the read's case (classifyReadsByCase :462-518,
// somaticReadHpCount :386-404)
__global__ __launch_bounds__(256) void k_tumor_pair_sites(TumOut T) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x, n_slots = (long long)LPS_TARENAS * T.pair_arena;
    if (t >= n_slots) return;
    const long long arena = t / T.pair_arena, idx = t - arena * T.pair_arena;
    if ((unsigned long long)idx >= T.pair_ctr[arena * 16]) return;
    const int v = T.apair_site[t], r = T.apair_read[t], base_hp = T.apair_hp[t];
    const int r_hp = T.hp[r], r_h1 = T.hp1[r], r_h2 = T.hp2[r]; const bool r_record = T.n_ps[r] <= 1, r_clean = (r_h1 == 0 || r_h2 == 0) && T.hp3[r] != 0;
    int32_t *sc = T.site + (size_t)v * LPS_TSITE_COUNTERS;
    atomicAdd(&sc[15 + r_hp], 1);                                         // base.ReadHpCount[hpResult] (:457)
    if (base_hp == 3) {
        if (!r_record) atomicAdd(&sc[24], 1);
        else if (r_clean) {
            atomicAdd(&sc[25], 1);
            if (r_h1 == 0 && r_h2 == 0) atomicAdd(&sc[28], 1); else if (r_h1 != 0 && r_h2 == 0) atomicAdd(&sc[26], 1); else if (r_h1 == 0 && r_h2 != 0) atomicAdd(&sc[27], 1);
        } else atomicAdd(&sc[29], 1);
        atomicAdd(&sc[30 + r_hp], 1);
    }
}

// thread t = (hit slot t >> 1, direction t & 1: 0 towards the read's start, 1 towards its end); a slot holds a hit when its index inside its arena is
// below the arena's counter.  ONE launch: a thread walks its window once (the reference's walk, win_dir above, untouched) and keeps what it found in
// registers (the memo); the wave sums its threads' counts, ONE atomic on the list's counter reserves the wave's stretch of the window list, and
// every thread stores its differences from the memo - a window of more than eight differences (or one that runs past the read's last base) is walked a
// second time, writing.  The list's order is that of the waves' reservations: it is a multiset (the reference keeps a map per site, the callers count).
__global__ __launch_bounds__(256) void k_tumor_windows(VarView V, ReadView R, TumOut T) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x, n_slots = (long long)LPS_TARENAS * T.hit_arena;
    const long long hs = t >> 1, arena = min(hs, n_slots - 1) / T.hit_arena, idx = hs - arena * T.hit_arena;
    const bool live = hs < n_slots && (unsigned long long)idx < T.hit_ctr[arena * 16];
    int n = 0, v = 0, r = 0, opj = 0, allele = 0, rp = 0, remaining = 0, dir = 1; WinMemo memo{0, 0}; uint32_t codes = 0;
    if (live) {
        const int4 h = T.hits[hs]; dir = (t & 1) ? +1 : -1;
        v = h.x; r = h.y; opj = h.z; const int win_off = h.w & 0x3fffffff; allele = (h.w >> 30) & 1; rp = T.hit_rp[hs];
        const uint32_t *cig = R.cig(r);
        const int len = (int)(cig[opj] >> 4);
        remaining = dir > 0 ? ((len - win_off > 0) ? len - win_off : 0) : (win_off > 0 ? win_off : 0);
        n = win_dir<false>(cig, opj, R.cp_n[r], R.seq + R.seq_off[r], R.l_qseq[r], V.ref, (int)V.ref_len_eff, rp, remaining, V.pos[v], dir, T, 0, v, allele, &memo, &codes);
    }
    const int incl = wave_incl_scan_dpp(n), total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) return;
    unsigned long long base = 0;
    if (lane_id() == 0) base = atomicAdd(T.win_total, (unsigned long long)total);
    base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if (n == 0) return;
    long long at = (long long)base + incl - n;
    if (!(memo.hi >> 63)) {                                               // from the memo
        int k = 0; unsigned long long lo = memo.lo, hi = memo.hi;
        while (lo | hi) {
            int i;
            if (lo) { i = __builtin_ctzll(lo) + 1; lo &= lo - 1; } else { i = __builtin_ctzll(hi) + 65; hi &= hi - 1; }
            if (at < T.win_cap) { T.win_site[at] = v; T.win_allele[at] = (uint8_t)allele; T.win_offset[at] = (int16_t)(i * dir); T.win_base[at] = (uint8_t)nt16_char((int)((codes >> (4 * k)) & 15u)); }
            ++at; ++k;
        }
        return;
    }
    (void)win_dir<true>(R.cig(r), opj, R.cp_n[r], R.seq + R.seq_off[r], R.l_qseq[r], V.ref, (int)V.ref_len_eff, rp, remaining, V.pos[v], dir, T, at, v, allele);
}
void launch_tumor_windows(const VarView &V, const ReadView &R, const TumOut &T, hipStream_t s) {
    const size_t n = (size_t)(2 * LPS_TARENAS * T.hit_arena);
    hipLaunchKernelGGL(k_tumor_windows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, V, R, T);
}

// totals of both lists (one wave), then the pairs out of their arenas into the caller's contiguous list: arena a's entries go behind those of the arenas before it
__global__ void k_tumor_totals(TumOut T) {
    const int a = threadIdx.x;                                            // LPS_TARENAS == 64 == one wave
    const unsigned long long p = T.pair_ctr[a * 16], h = T.hit_ctr[a * 16];
    unsigned long long ps = p, pm = p, hs = h, hm = h;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ps += __shfl_xor(ps, d); hs += __shfl_xor(hs, d); const unsigned long long x = __shfl_xor(pm, d), y = __shfl_xor(hm, d); pm = x > pm ? x : pm; hm = y > hm ? y : hm; }
    if (a == 0) { T.tot[0] = ps; T.tot[1] = pm; T.tot[2] = hs; T.tot[3] = hm; }
}
__global__ __launch_bounds__(256) void k_tumor_pairs_out(TumOut T) {
    static_assert(LPS_TARENAS == 64, "one wave reduces the arena counters");
    __shared__ unsigned long long s_base;
    const int a = blockIdx.y; const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long n_a = min(T.pair_ctr[a * 16], (unsigned long long)T.pair_arena);
    if ((unsigned long long)blockIdx.x * 256ull >= n_a) return;
    if (threadIdx.x < 64) {
        unsigned long long b = threadIdx.x < a ? min(T.pair_ctr[threadIdx.x * 16], (unsigned long long)T.pair_arena) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) b += __shfl_xor(b, d);
        if (threadIdx.x == 0) s_base = b;
    }
    __syncthreads();
    if ((unsigned long long)idx >= n_a) return;
    const long long src = (long long)a * T.pair_arena + idx, dst = (long long)s_base + idx;
    if (dst < T.pair_cap) { T.pair_site[dst] = T.apair_site[src]; T.pair_read[dst] = T.apair_read[src]; T.pair_hp[dst] = T.apair_hp[src]; }
}
void launch_tumor_pairs_out(const TumOut &T, hipStream_t s) {
    hipLaunchKernelGGL(k_tumor_totals, dim3(1), dim3(64), 0, s, T);
    hipLaunchKernelGGL(k_tumor_pairs_out, dim3((unsigned)((T.pair_arena + 255) / 256), LPS_TARENAS), dim3(256), 0, s, T);
}

// general = false: the stream walk (R.v0 must hold every alignment's first row); true: the per-op-prefix walker, which takes any BAM record
void launch_tumor_extract(const VarView &V, const ReadView &R, const TumOut &T, int mapping_quality, int tag_supplementary, int pass,
                          LpsCounters *cnt, hipStream_t s, bool general) {
    if (R.n == 0) return;
    if (!general) {
        if (pass == 0) hipLaunchKernelGGL(k_tumor_stream, dim3((R.n + 3) / 4), dim3(64), 0, s, V, R, T, mapping_quality, tag_supplementary, cnt);
        else hipLaunchKernelGGL(k_tumor_pair_sites, dim3((unsigned)(((long long)LPS_TARENAS * T.pair_arena + 255) / 256)), dim3(256), 0, s, T);   // (the walk listed the pairs: no second one)
        return;
    }
    const dim3 g(R.n), b(64);
    if (pass == 0) hipLaunchKernelGGL(k_tumor_extract<0>, g, b, 0, s, V, R, T, mapping_quality, tag_supplementary, cnt);
    else hipLaunchKernelGGL(k_tumor_extract<1>, g, b, 0, s, V, R, T, mapping_quality, tag_supplementary, cnt);
}
