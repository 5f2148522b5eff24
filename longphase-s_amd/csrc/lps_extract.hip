// lps_extract.hip — variant-table preparation and the read x variant allele extraction kernel (gfx950).
//
// Replaces (reference file:line, relative to /root/reference/):
//   SnpParser::getVariants_markindel   src/phase/ParsingBam.cpp:378-417     -> k_variant_table (danger flag)
//   homopolymerLength                  src/shared/Util.cpp:21-54            -> k_variant_table (hpoly)
//   SnpParser::filterSNP               src/phase/ParsingBam.cpp:837-912     -> k_variant_table (erased flag)
//   BamParser::direct_detect_alleles   src/phase/ParsingBam.cpp:1243-1301   -> k_extract_phase (filters)
//   BamParser::get_snp / getClip       src/phase/ParsingBam.cpp:1303-1645   -> k_extract_phase
//
// Design (MI355X-first, not a translation of the reference's cursor walk; the kernel's own comment below has the details):
//   * a 64-lane wavefront (one per workgroup) takes a JOB of four consecutive alignments.  Their CIGAR words are resident in lane-chunks of 8, every
//     alignment padded to a whole number of chunks (lps_reads.hip), so the job's words are ONE stream: 512 words per round, 8 per lane, four rounds
//     requested together; one pair of DPP wave scans per round turns the lanes' advances into stream coordinates, 8 bytes per chunk kept in LDS;
//   * clips (getClip) are read off the first two and last two words of each alignment; the walk only counts clip / rejected ops and sends the job to
//     the general walker (k_extract_redo) when the count disagrees, as it does for absurd lengths;
//   * instead of walking ops and advancing a variant cursor, the VARIANTS search the ops: the candidate variants of the four alignments (contiguous
//     slices of the position-sorted table) are counted, ONE atomicAdd reserves that many observation slots, and the candidates are taken 64 at a time
//     as one flattened list with every lane busy - binary search of the chunk table in LDS, the chunk's 8 words from the caches, an 8-step walk in
//     registers, the reference's rule for the op, base and quality gathered IN PLACE from the BAM record's own encodings (two random lines of HBM
//     per site: what bounds the kernel), allele called, filterSNP's erasure applied, the record written straight to its compacted place and counted
//     in its variant's list (the rank the counting atomic returns travels with the record);
//   * the four 16-byte row descriptors of a job are one 64-byte line; its clip events go to 16 slots the job owns (no list, no counter);
//   * a job whose chunks do not fit the LDS table is walked in groups of alignments, an alignment that alone does not fit with a coarser table; a job in
//     which get_snp's early return fires queues itself for k_extract_redo, the general per-op-prefix walker that takes any BAM record.
#include <algorithm>

#include "lps_kernels.h"

// ------------------------------------------------------------------------------------------------ variants
// ONE launch makes everything the walkers read about the table (round 4: four dependent launches before - prep, filterSNP, pack + buckets, first
// candidates - each a few microseconds of work behind a kernel boundary):
//   * workgroups [0, nb_var): thread per variant - danger flag (getVariants_markindel), homopolymer length, filterSNP's erasure, the packed 8-byte
//     record {pos, attr} so that a candidate costs ONE load in the walkers.  filterSNP (:837-912) relates only SNPs at most 2 bp apart: the table
//     splits into independent chains at every larger gap, and a variant that is not its chain's head replays the reference's erase-while-iterating
//     scan from the head down to itself (chains hold two or three rows; the homopolymer lengths on the way are recomputed, a few bytes of reference each);
//   * the next workgroups: the coarse position index (bucket b = first variant at or beyond b << LPS_BUCKET_SHIFT), plain binary search;
//   * the last workgroups (when asked for): first candidate row of every alignment, a binary search of the positions per alignment.
__device__ __forceinline__ int var_hpoly(const VarView &V, int v) { return homopolymer_length(V.ref, V.ref_len_eff, (long long)V.pos[v]); }
__global__ void k_variant_table(VarView V, int is_ont, uint2 *rec, int32_t *bucket, int nb_var, int nb_bucket, const int32_t *ref_start, int n_reads, int32_t *v0) {
    const int blk = (int)blockIdx.x;
    if (blk >= nb_var + nb_bucket) {                                       // ---- first candidate of every alignment (== lane_var_lower_bound)
        const int r = (blk - nb_var - nb_bucket) * blockDim.x + threadIdx.x;
        if (r >= n_reads) return;
        const int key = ref_start[r];
        int lo = 0, hi = V.n;
        if (key >= 0) while (lo < hi) { const int m = (lo + hi) >> 1; if (V.pos[m] < key) lo = m + 1; else hi = m; } else lo = 0;
        v0[r] = lo;
        return;
    }
    if (blk >= nb_var) {                                                   // ---- bucket index
        const int b = (blk - nb_var) * blockDim.x + threadIdx.x;
        if (b > V.n_bucket) return;
        const long long key = (long long)b << LPS_BUCKET_SHIFT;
        int lo = 0, hi = V.n;
        while (lo < hi) { const int m = (lo + hi) >> 1; if ((long long)V.pos[m] < key) lo = m + 1; else hi = m; }
        bucket[b] = lo;
        return;
    }
    const int v = blk * blockDim.x + threadIdx.x;
    if (v >= V.n) return;
    const long long L = V.ref_len_eff;
    const long long p0 = V.pos[v];
    auto at = [&](long long i) -> char { return (i >= 0 && i < L) ? V.ref[i] : '\0'; };
    const int rl = V.ref_len[v], al = V.alt_len[v];
    uint8_t danger = 0;
    if (rl > 1 || al > 1) {                                               // getVariants_markindel (:391-406): the 2-mer behind the site repeats five times
        long long p = p0; const char a = at(p + 1), b = at(p + 2); int i = 0;
        while (i < 5) { if (a != at(p + 1) || b != at(p + 2)) break; p += 2; ++i; }
        danger = (i == 5);
    }
    const int hp = var_hpoly(V, v);
    uint8_t erased = 0;
    if (is_ont && v != 0 && V.pos[v] - V.pos[v - 1] <= 2) {                // not a chain head: replay the chain's scan up to this row
        int h = v - 1;
        while (h > 0 && V.pos[h] - V.pos[h - 1] <= 2) --h;
        int cur = h, hp_cur = var_hpoly(V, h);
        for (int nxt = h + 1; nxt <= v; ++nxt) {
            const int hp_nxt = nxt == v ? hp : var_hpoly(V, nxt);
            const bool er = hp_cur >= 3 && hp_nxt >= 3 && V.pos[nxt] - V.pos[cur] <= 2;
            if (nxt == v) erased = er;
            if (!er) { cur = nxt; hp_cur = hp_nxt; }
        }
    }
    V.danger[v] = danger; V.hpoly[v] = (uint8_t)hp; V.erased[v] = erased;
    const unsigned kind = (rl == 1 && al == 1) ? 0u : ((rl == 1 && al != 1) ? 1u : ((rl != 1 && al == 1) ? 2u : 3u));
    const unsigned attr = (unsigned)V.ref0[v] | ((unsigned)V.alt0[v] << 8) | (kind << 16) | (danger ? VREC_DANGER : 0u) |
                          (erased ? VREC_ERASED : 0u) | (hp >= 3 ? VREC_HPOLY3 : 0u) |
                          ((V.hp1_is_alt && V.hp1_is_alt[v]) ? VREC_HP1ALT : 0u) |
                          (V.somatic_role ? ((unsigned)(V.somatic_role[v] & 3) << 22) | ((unsigned)((V.derive_hp ? V.derive_hp[v] : 0) & 3) << 24) : 0u) |
                          (V.tumor_kind ? ((unsigned)(V.tumor_kind[v] & 7) << 26) : 0u);
    rec[v] = make_uint2((unsigned)V.pos[v], attr);
}

void launch_variant_prep(const VarView &V, int is_ont, int32_t *bucket, uint2 *rec, hipStream_t s, const int32_t *ref_start, int n_reads, int32_t *v0) {
    if (V.n == 0) return;
    const int b = 256, nb_var = (V.n + b - 1) / b, nb_bucket = (V.n_bucket + 1 + b) / b, nb_reads = (v0 && n_reads > 0) ? (n_reads + b - 1) / b : 0;
    hipLaunchKernelGGL(k_variant_table, dim3(nb_var + nb_bucket + nb_reads), dim3(b), 0, s, V, is_ont, rec, bucket, nb_var, nb_bucket, ref_start, v0 ? n_reads : 0, v0);
}

// ------------------------------------------------------------------------------------------------ extraction
#define EXT_RPW 4       // alignments per wave (a "job")
#ifndef EXT_TAB
#define EXT_TAB 1024    // lane-chunks (8 CIGAR words each) a wave keeps in LDS - words (16 KB) and the chunks' coordinates (4 KB): 4 096 words, ~100 kb of
#endif                  // ONT read; a job that holds more is walked in groups, an alignment that alone holds more with a coarser table
#ifndef EXT_WAVES
#define EXT_WAVES 4     // waves per SIMD the register budget is sized for: 128 VGPRs, no spills; 5 / 6 / 8 waves (spills, smaller tables) were measured slower
#endif
#ifndef EXT_TRIP
#define EXT_TRIP 4      // rounds of the walk requested together (8 VGPRs each)
#endif
#ifndef EXT_GROUP_MAX
#define EXT_GROUP_MAX 4 // alignments walked together (their candidates are resolved together once their words are through)
#endif


// One wavefront = one job of four consecutive alignments; their CIGAR words lie back to back in lane-chunks of 8, each alignment padded to a whole
// number of chunks (lps_reads.hip).
//   * WALK.  The words of the job are ONE stream, taken 512 words per round, 8 consecutive words per lane, every lane busy whatever the alignments'
//     lengths.  A lane sums the reference / query advance of its 8 words (six vector instructions per word) and one pair of DPP scans over the wave
//     turns the sums into STREAM coordinates: reference and query bases consumed since the job's first word, running on across alignments.  The
//     lane's pair goes to LDS (8 bytes per chunk) - no per-op prefixes, no per-alignment bookkeeping, no branch and no store to memory inside the
//     loop; four rounds are requested together, 8 KB of the stream in flight per wave.  An alignment begins and ends on a chunk: where it starts
//     in the stream and where it ends on the reference are table entries.
//   * CLIPS (getClip :1613-1645) are looked for where the SAM format puts them: lane q examines the first two and the last two words of alignment
//     q, loaded ahead of the walk; their positions follow from the alignment's start and end.  The walk only COUNTS the words whose op is a clip
//     or one the reference rejects; a count that differs from what the alignments' ends hold - a clip in the middle of a CIGAR, an unknown op -
//     sends the job to the general walker (k_extract_redo), like a length of 2^24 and more.
//   * CANDIDATES.  With the job's table in LDS, the candidate variants of the four alignments (position-sorted slices of the variant table:
//     [first variant at or after the alignment's start, first variant at or beyond its reference end)) are counted, ONE atomicAdd reserves that
//     many observation slots, and the candidates are taken 64 at a time as one flattened list, every lane busy: binary search of the alignment's
//     chunks for the last chunk that starts at or before the variant, the chunk's 8 words (+ the one after) from the caches, an 8-step walk in
//     registers to the op that covers the variant, the reference's rules for that op (ParsingBam.cpp:1445-1607), base and quality gathered
//     right there (in place: the BAM record's 4-bit bases and its qualities, two lines of HBM), allele called, filterSNP's erasures applied, the observation counted (its rank
//     in the variant's list) and the record written to its final, compacted place.  Every CIGAR word is fetched from memory ONCE; a base /
//     quality pair costs two random lines.
// A job whose chunks do not fit the table (EXT_TAB) is walked in groups of alignments; an alignment that alone does not fit is walked with one
// table entry per 1 << shift chunks and re-reads the words it needs (LONG mode: read lengths beyond ~100 kb).  A job in which get_snp's early
// return fires (:1453-1455, :1559-1561: a record whose SEQ is shorter than its CIGAR) queues itself for k_extract_redo before it has written
// anything a later stage looks at.
__global__ __launch_bounds__(64, EXT_WAVES) void k_extract_phase(VarView V, ReadView R, ObsView O, ClipView C, int mapping_quality,
                                                      LpsCounters *cnt, uint32_t *redo_list, unsigned *n_redo, uint32_t *var_cnt, uint32_t *var_del) {
    __shared__ __attribute__((aligned(16))) int2 s_tab[EXT_TAB + 1];
    __shared__ ExtHdr s_hdr[EXT_RPW];
    __shared__ ClipEv s_clip[EXT_CLIPS];
    static_assert(EXT_CLIPS >= 4 * EXT_RPW, "four end words per alignment");
    const int l = lane_id();
    // Output rows are reserved on one of LPS_ARENAS counters (own cache line each).  Workgroups are dealt round-robin over the 8 XCDs, so
    // arena = blockIdx % 64 keeps each counter inside ONE XCD's L2.
    const int arena = blockIdx.x % O.n_arenas;
    const unsigned long long arena_lo = (unsigned long long)arena * O.arena_size;
    const int job = blockIdx.x;
    const int r0 = job * EXT_RPW;
    if (r0 >= R.n) return;
    const int nq = min(EXT_RPW, R.n - r0);
    static_assert(EXT_RPW == 4, "lane layout of the planning step");
    auto no_clips = [&]() __attribute__((always_inline)) { if (l < EXT_CLIPS) C.ev[(size_t)EXT_CLIPS * job + l] = ClipEv{0, 0, -1}; };   // the job's slots of the clip list hold nothing
    auto to_redo = [&]() __attribute__((always_inline)) { no_clips(); if (l == 0) redo_list[atomicAdd(n_redo, 1u)] = (uint32_t)job; };

    // ---- plan: headers, alignment q in lane q.  direct_detect_alleles filters (:1282-1291) + region "chr:1-<lastSNPPos>" (:1273)
    int h_start = 0, h_lq = 0, h_v0 = 0, h_n = 0; bool h_live = false; unsigned h_cp = 0; unsigned long long h_soff = 0, h_qoff = 0;
    if (l <= nq) h_cp = R.cp_off[r0 + l];
    if (l < nq) {
        const int r = r0 + l; h_start = R.ref_start[r]; h_lq = R.l_qseq[r]; h_soff = R.seq_off[r]; h_qoff = R.qual_off[r]; h_v0 = V.n ? R.v0[r] : 0; h_n = R.cp_n[r];
        const int flag = R.flag[r];
        h_live = !(R.mapq[r] < mapping_quality || (flag & 0x4) || (flag & 0x100) || (flag & 0x400) || h_start >= V.last_pos);
    }
    const unsigned live_mask = (unsigned)__ballot(h_live) & 15u;
    if (!live_mask) {                                                  // nothing to walk: four empty rows
        if (l < nq) O.rows[r0 + l] = RowDesc{0u, 0, 0x7fffffff, 0u};
        no_clips();
        return;
    }
    const int h_nch = (int)(__shfl_down(h_cp, 1) - h_cp);             // chunks of alignment q (lanes < nq)
    // what lane q collects for row q; clip events of the job
    unsigned row_off = 0; int row_cnt = 0; unsigned row_flags = 0;
    int n_clip = 0; bool fail = false, arena_full = false;

    // ---- the job's alignments in GROUPS whose chunks fit the table together: nearly always one group of four
    unsigned todo = live_mask;
#pragma unroll 1
    while (todo) {
        const int qa = __builtin_ctz(todo);
        const unsigned c_lo = __shfl(h_cp, qa);
        int qb = qa; unsigned gm = 1u << qa;
        for (int q = qa + 1; q < nq; ++q) {
            if (!((todo >> q) & 1u)) continue;
            if (__shfl(h_cp, q + 1) - c_lo > (unsigned)EXT_TAB || __popc(gm) >= EXT_GROUP_MAX) break;
            gm |= 1u << q; qb = q;
        }
        todo &= ~gm;
        // LONG mode: the group's first alignment alone holds more chunks than the table takes (then it is the whole group): one entry per
        // 1 << shift chunks, the words themselves are read again where a candidate needs them
        int shift = 0;
        { const unsigned n1 = __shfl(h_cp, qa + 1) - c_lo; while (((n1 + (1u << shift) - 1u) >> shift) > (unsigned)EXT_TAB) ++shift; }
        const bool fast = shift == 0;
        // the stream: from the first chunk of the group's first alignment to the last chunk of its last one (alignments in between that are not
        // walked - low MAPQ, secondary - pass by as words that only move the coordinates on)
        const bool h_in = l < 4 && ((gm >> l) & 1u);
        const bool h_strm = l < nq && l >= qa && l <= qb && h_n > 0;      // alignment q has words in the stream
        const bool h_walk = h_in && h_n > 0;
        const int h_c0 = (l <= nq) ? (int)(h_cp - c_lo) : 0;              // first chunk of alignment q inside the stream
        const uint32_t *cg = R.cigp + 8ull * c_lo;
        const int TC = __builtin_amdgcn_readlane(h_c0 + h_nch, qb);       // chunks of the stream
        // UNCONDITIONAL loads, clamped to the stream (a load under a branch makes the compiler wait for every outstanding load where the paths join)
        auto request = [&](int cid, uint32_t (&w)[8]) __attribute__((always_inline)) {
            const uint32_t *p = cg + 8 * min(cid, max(TC - 1, 0));
            const uint4 a = *reinterpret_cast<const uint4 *>(p), b = *reinterpret_cast<const uint4 *>(p + 4);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        };
        // the first two and the last two words of alignment q (lane q: where its clips are), the positions of each alignment's first 64 candidate
        // variants (counted against its reference end after the walk): requested ahead of the walk, looked at after it
        uint32_t e_f0, e_f1, e_b1, e_b0;
        {
            const uint32_t *e = cg + 8 * (h_strm ? h_c0 : 0); const int n = h_strm ? h_n : 1;
            e_f0 = e[0]; e_f1 = e[min(1, n - 1)]; e_b1 = e[max(n - 2, 0)]; e_b0 = e[n - 1];
        }
        int v0q[4], pp[4]; bool walkq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0q[q] = __builtin_amdgcn_readlane(h_v0, q); walkq[q] = (__ballot(h_walk) >> q) & 1ull; pp[q] = V.pos[min(v0q[q] + l, V.n - 1)]; }
        if (l < 4) {                                                      // (first half of the header)
            ExtHdr &h = s_hdr[l];
            h.crel = fast ? 8 * h_c0 : 0; h.ncig = h_walk ? h_n : 0; h.c0 = fast ? h_c0 : 0; h.nch = h_walk ? (int)(((unsigned)h_nch + (1u << shift) - 1u) >> shift) : 0;
            h.lq = h_lq; h.soff = h_soff; h.qoff = h_qoff;
        }

        // ---- walk.  FOUR rounds per trip, all four requested at its head: 8 KB of the stream in flight per wave, one exposed memory latency per
        //      trip (an ordinary job is two trips).  Nothing is carried from trip to trip in a buffer: a buffer whose load crosses the loop's back
        //      edge is copied at the end of the trip that issued the load (the compiler gives the in-loop load other registers than the one before
        //      the loop), and the copy waits for the load there - a prefetch that overlaps nothing (seen in the ISA with one and with two buffers)
        int carry_r = 0, carry_q = 0; unsigned special = 0; uint32_t big = 0; bool absurd = false;
#pragma unroll 1
        for (int R0 = 0; R0 < TC; R0 += 64 * EXT_TRIP) {
            uint32_t wt[EXT_TRIP][8];
#pragma unroll
            for (int t = 0; t < EXT_TRIP; ++t) request(R0 + 64 * t + l, wt[t]);
#pragma unroll
            for (int t = 0; t < EXT_TRIP; ++t) stream_round<LPS_CLIPMASK2 | LPS_BADMASK2>(wt[t], R0 + 64 * t + l, TC, shift, s_tab, carry_r, carry_q, special, big);
            absurd |= (unsigned)carry_r > 0x3fffffffu || (unsigned)carry_q > 0x3fffffffu;    // stream coordinates are 32-bit: absurd spans go to the general walker
            if (absurd) break;
        }
        if (fast && l == 0) s_tab[TC] = make_int2(carry_r, carry_q);      // where the stream ends: the end of its last alignment
        // what the alignments' ends hold against what the walk counted
        const unsigned c_f0 = op_bit(LPS_CLIPMASK2, e_f0), c_f1 = h_n >= 2 ? op_bit(LPS_CLIPMASK2, e_f1) : 0u,
                       c_b1 = h_n >= 4 ? op_bit(LPS_CLIPMASK2, e_b1) : 0u, c_b0 = h_n >= 3 ? op_bit(LPS_CLIPMASK2, e_b0) : 0u;
        const int expected = wave_sum(h_strm ? (int)(c_f0 + c_f1 + c_b1 + c_b0) : 0), counted = wave_sum((int)special);
        if (absurd || expected != counted || __ballot(big >= 0x10000000u)) {
            // (groups done before this one were counted: taken off as in the early-return case below)
            if (var_cnt) {
                unsigned gone = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)row_off, q); const int n = __builtin_amdgcn_readlane(row_cnt, q);
                    for (int i = l; i < n; i += 64) atomicAdd(&var_del[O.rec[o + i].var], 1u);
                    gone += (unsigned)n;
                }
                if (l == 0 && gone) atomicAdd(&cnt->n_abandoned, gone);
            }
            to_redo(); return;
        }
        wave_sync();

        // ---- where each alignment begins in the stream and ends on the reference; its clips; its candidates: variants [v0, first variant at or beyond its end)
        int b_sat = 0, b_qat = 0, b_rend = h_start;
        if (h_walk) {
            if (fast) { const int2 ts = s_tab[h_c0], te = s_tab[h_c0 + h_nch]; b_sat = ts.x; b_qat = ts.y; b_rend = h_start + te.x - ts.x; }
            else b_rend = h_start + carry_r;                                // LONG mode: the alignment is the whole stream
        }
        {   // getClip (:1613-1620,1636-1645): soft / hard clips longer than 5; FRONT iff CIGAR index 0.  Word i sits at the reference position the
            // walk has reached before it: the start (+ what word 0 consumes), the end (- what the last words consume)
            const bool k_f0 = h_walk && c_f0 && (e_f0 >> 4) > 5u, k_f1 = h_walk && c_f1 && (e_f1 >> 4) > 5u,
                       k_b1 = h_walk && c_b1 && (e_b1 >> 4) > 5u, k_b0 = h_walk && c_b0 && (e_b0 >> 4) > 5u;
            const int ne = (int)k_f0 + (int)k_f1 + (int)k_b1 + (int)k_b0;
            const int ne0 = __builtin_amdgcn_readlane(ne, 0), ne1 = __builtin_amdgcn_readlane(ne, 1), ne2 = __builtin_amdgcn_readlane(ne, 2), ne3 = __builtin_amdgcn_readlane(ne, 3);
            int slot = n_clip + (l > 0 ? ne0 : 0) + (l > 1 ? ne1 : 0) + (l > 2 ? ne2 : 0);
            if (l < 4 && ne) {
                const int p_b0 = b_rend - ref_len_of(e_b0), p_b1 = p_b0 - ref_len_of(e_b1), rd = r0 + l;
                if (k_f0) s_clip[slot++] = ClipEv{h_start, 0, rd};
                if (k_f1) s_clip[slot++] = ClipEv{h_start + ref_len_of(e_f0), (1 << 1) | 1, rd};
                if (k_b1) s_clip[slot++] = ClipEv{p_b1, ((h_n - 2) << 1) | 1, rd};
                if (k_b0) s_clip[slot++] = ClipEv{p_b0, ((h_n - 1) << 1) | 1, rd};
            }
            n_clip += ne0 + ne1 + ne2 + ne3;
        }
        int ncand[4], rend[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rend[q] = __builtin_amdgcn_readlane(b_rend, q);
            int n = __popcll(__ballot(walkq[q] && v0q[q] + l < V.n && pp[q] < rend[q]));
            if (n == 64) {                                                // more than a wave's worth (dense tables): count on
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
        // ---- ONE reservation for the rows of the group: a slot per candidate (the few candidates that turn out not to be observations - a base that
        //      is neither allele, a variant inside a deletion, a variant filterSNP erased - leave slots unused at the end), so that records go straight
        //      to their compacted place
        unsigned long long off = 0;
        if (T > 0 && l == 0) off = atomicAdd(&O.arena_ctr[arena * 8], (unsigned long long)T);
        wave_sync();
        const int step0 = maxnch > 1 ? 1 << (31 - __builtin_clz(maxnch - 1)) : 0;    // the steps step0, step0/2, .. 1 sum to >= maxnch - 1
        unsigned long long g0 = 0; ObsRec *dst = nullptr;
        int n_out = 0, n_emit[4] = {0, 0, 0, 0}; unsigned any_pre = 0;
        uint2 pvr = V.rec[min(SELC(l, cum, vadj) + l, V.n - 1)];           // records are requested one round ahead
        // an observation is counted where it is made: what the counting atomic returns is its rank inside the variant's list of observations, kept
        // beside allele and quality - the node-major lists are filled later without a counting pass and without a second atomic.  The atomic's
        // answer is not waited for: the record leaves at once, the rank follows a round later (or after the last round), when it has long arrived
        bool pend = false; uint32_t pend_slot = 0, pend_aq = 0; unsigned pend_rk = 0;
#pragma unroll 1
        for (int i0 = 0; i0 < T; i0 += 64) {
            const int i = i0 + l;
            const bool in = i < T;
            int allele = -1, qv = 0, v = 0; bool erased = false;
            const uint2 vr = pvr;
            pvr = V.rec[min(SELC(i + 64, cum, vadj) + i + 64, V.n - 1)];
            if (in) {
                const int q = (i >= cum[1]) + (i >= cum[2]) + (i >= cum[3]);
                const int4 ha = *reinterpret_cast<const int4 *>(&s_hdr[q].crel), hb = *reinterpret_cast<const int4 *>(&s_hdr[q].vadj);
                const int hcrel = ha.x, hncig = ha.y, hc0 = ha.z, hnch = ha.w, hlq = hb.y;
                v = hb.x + i;
                const int p = (int)vr.x; const unsigned at = vr.y;
                erased = (at & VREC_ERASED) != 0u;
                const int ps = p + hb.z;                                  // the variant in stream coordinates
                // last chunk of the alignment that starts at or before it (the alignment's first chunk does: the variant lies at or after its start)
                int co = 0;
                for (int step = step0; step >= 1; step >>= 1) { const int t = co + step; const int sv = s_tab[hc0 + min(t, hnch - 1)].x; co = (t < hnch && sv <= ps) ? t : co; }
                const int2 base = s_tab[hc0 + co];
                const int x0 = (8 * (hc0 + co)) << shift;                  // stream index of the chunk's first word
                // the op that covers the variant: the last one that starts at or before it.  Starts never decrease, so "starts at or before" holds
                // for a prefix of the words; the padding past the alignment's last word starts at its end, beyond every candidate
                int rr = base.x, qq = base.y, jx = x0, rs = base.x, qs = base.y; uint32_t wj = 6u, wn = 6u;
                auto walk8 = [&](const uint32_t (&w)[9], int xb) __attribute__((always_inline)) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bool le = rr <= ps;
                        jx = le ? xb + k : jx; rs = le ? rr : rs; qs = le ? qq : qs; wj = le ? w[k] : wj; wn = le ? w[k + 1] : wn;
                        const unsigned len = w[k] >> 4;                  // (below 2^24: the walk sent everything else to the general walker)
                        rr += (int)__umul24(len, op_bit(LPS_RMASK2, w[k])); qq += (int)__umul24(len, op_bit(LPS_QMASK2, w[k]));
                    }
                };
                for (int u = 0; u < (1 << shift); ++u) {                      // the entry's 8 << shift words, 8 at a time, until the walk is past the variant (one trip unless LONG)
                    const uint32_t *cw = cg + x0 + 8 * u;
                    uint32_t w[9];
                    const uint4 a = *reinterpret_cast<const uint4 *>(cw), b = *reinterpret_cast<const uint4 *>(cw + 4);
                    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w; w[8] = cw[8];
                    walk8(w, x0 + 8 * u);
                    if (rr > ps || x0 + 8 * u + 8 >= hcrel + hncig) break;
                }
                const int j = jx - x0;
                const int op = wj & 15, len = (int)(wj >> 4);
                const int opi = x0 + j - hcrel;                           // op index inside the alignment
                qs -= hb.w;                                               // query position inside the alignment
                if (ps < rs + len) {
                    const unsigned kind = VREC_KIND(at);
                    int qi = -1;
                    if (op_is_match(op)) {                                        // :1445-1520
                        const int o = ps - rs;
                        if (qs + o + 1 > hlq) fail = true;                        // :1453-1455
                        else if (kind == 0) qi = qs + o;
                        else if ((kind == 1 || kind == 2) && opi + 1 < hncig) {   // indel variant :1470-1510
                            const int want = (kind == 1) ? 1 : 2;                  // next op must be I resp. D
                            allele = (rs + len - 1 == ps && (int)(wn & 15u) == want) ? 1 : 0;
                            qv = (at & VREC_DANGER) ? -5 : -4;
                        }
                    } else if (op == 2) {                                         // :1539-1607
                        // only the first variant at/after the deletion start is examined by the reference
                        const bool first_in = (v == 0) || V.pos[v - 1] + hb.z < rs;
                        if (first_in && (at & VREC_HPOLY3)) {
                            if (qs + 1 > hlq) fail = true;                        // :1559-1561
                            else if (kind == 0) qi = qs;
                            else if (kind == 2) { allele = 1; qv = -4; }
                        }
                    }
                    if (qi >= 0) {                                                // base and quality at the variant site, in place (the BAM record's own encodings)
                        const char ref_c = (char)(at & 0xff), alt_c = (char)((at >> 8) & 0xff);
                        int code;
                        const ulonglong2 sqo = *reinterpret_cast<const ulonglong2 *>(&s_hdr[q].soff);
                        sq_fetch(R.seq, R.qual, sqo.x, sqo.y, qi, code, qv);
                        const char base_c = nt16_char(code);
                        if (base_c == ref_c) allele = 0; else if (base_c == alt_c) allele = 1;
                    }
                }
            }
            const bool pre = in && allele != -1;                          // an observation before filterSNP
            const bool ok = pre && !erased;
            const unsigned long long pm = __ballot(pre), om = __ballot(ok);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int a = max(cum[k] - i0, 0), b = min(cum[k + 1] - i0, 64);   // lanes of row k in this round
                if (b > a) {
                    const unsigned long long rm = ((b >= 64) ? ~0ull : ((1ull << b) - 1ull)) & ~((1ull << a) - 1ull);
                    n_emit[k] += __popcll(om & rm); if (pm & rm) any_pre |= 1u << k;
                }
            }
            if (i0 == 0) {                                                // the reservation has had the first round's searches to arrive
                off = __shfl(off, 0);
                if (off + (unsigned long long)T > O.arena_size) arena_full = true;
                g0 = arena_lo + off; dst = O.rec + g0;
            }
            const bool put = ok && !arena_full;
            const uint32_t slot = (uint32_t)(n_out + __popcll(om & lanemask_lt())), aqw = (uint32_t)pack_aq(allele, qv);
            unsigned rk = 0;
            if (put) { dst[slot] = ObsRec{O.snp_u ? O.snp_u[v] : (int32_t)v, aqw}; if (var_cnt) rk = atomicAdd(&var_cnt[v], 1u); }   // (with SV / MOD rows: the index in the union of the tables)
            if (pend) {
                if (pend_rk > 0x3fffffu) atomicOr(&cnt->err, (unsigned)LPS_ERR_KEY_RANGE); dst[pend_slot].aq = pend_aq | (pend_rk << 10); }
            pend = put && var_cnt != nullptr; pend_slot = slot; pend_aq = aqw; pend_rk = rk;
            n_out += __popcll(om);
        }
        if (pend) { if (pend_rk > 0x3fffffu) atomicOr(&cnt->err, (unsigned)LPS_ERR_KEY_RANGE); dst[pend_slot].aq = pend_aq | (pend_rk << 10); }
        if (__ballot(fail)) {
            // get_snp returned early somewhere in these alignments (SEQ shorter than the CIGAR says: the read is dropped but clips of earlier ops
            // stay): the general walker replays the whole job.  Nothing a later stage looks at has been written, but the observations made so far
            // were COUNTED: they are taken off again (var_del), their places in the variants' lists stay holes (the host fills the lists with the
            // hole key when n_abandoned is not zero); the reserved slots stay unused
            if (var_cnt) {
                __threadfence();
                unsigned gone = (unsigned)n_out;
                for (int i = l; i < n_out; i += 64) atomicAdd(&var_del[dst[i].var], 1u);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)row_off, q); const int n = __builtin_amdgcn_readlane(row_cnt, q);
                    for (int i = l; i < n; i += 64) atomicAdd(&var_del[O.rec[o + i].var], 1u);
                    gone += (unsigned)n;
                }
                if (l == 0 && gone) atomicAdd(&cnt->n_abandoned, gone);
            }
            to_redo();
            return;
        }
        // rows of the group's alignments: back to back in the group's reservation
        if (h_in) {
            int before = 0, mine = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { if (k < l) before += n_emit[k]; if (k == l) mine = n_emit[k]; }
            row_off = (uint32_t)(g0 + (unsigned)before); row_cnt = mine; row_flags = (((any_pre >> l) & 1u) && mine == 0) ? 1u : 0u;
        }
        wave_sync();                                                      // the table and the headers are reused by the next group
    }
    if (arena_full && l == 0) atomicOr(&cnt->err, (unsigned)LPS_ERR_OBS_OVERFLOW);          // the host grows the arenas and reruns
    if (l < nq) {
        const bool ok = h_live && !arena_full;
        O.rows[r0 + l] = RowDesc{ok ? row_off : 0u, ok ? row_cnt : 0, 0x7fffffff, ok ? row_flags : 0u};
    }
    if (l < EXT_CLIPS) C.ev[(size_t)EXT_CLIPS * job + l] = (l < n_clip && !arena_full) ? s_clip[l] : ClipEv{0, 0, -1};   // the job's own slots: no counter (ClipView)
}

#define REDO_CAP 512    // observations buffered per wave of the redo kernel
// The general walker: jobs k_extract_phase queued (a clip in the middle of a CIGAR, an op the reference rejects, an op of 2^24 bases and more, a record
// whose SEQ is shorter than its CIGAR says), one job (four alignments) per wave, per-op prefixes staged in LDS 512 ops at a time: observations are called as their segment is searched and collected in an LDS buffer; when that fills up the wave reserves an upper bound
// for the row it is in - remaining reference span -> remaining candidates -, empties the buffer and writes the rest of that row directly.
__global__ __launch_bounds__(256, 4) void k_extract_redo(VarView V, ReadView R, ObsView O, ClipView C, int mapping_quality,
                                                      LpsCounters *cnt, const uint32_t *redo_list, const unsigned *n_redo, uint32_t *var_cnt, uint32_t *var_del) {
    __shared__ __attribute__((aligned(16))) int s_ref[4][LPS_SEG];
    __shared__ __attribute__((aligned(16))) int s_qry[4][LPS_SEG];
    __shared__ __attribute__((aligned(16))) uint32_t s_cig[4][LPS_SEG + 4];
    __shared__ int s_bvar[4][REDO_CAP];
    __shared__ uint32_t s_baq[4][REDO_CAP];
    enum { H_START, H_LQ, H_CP, H_NCIG, H_V0, H_SOFF, H_QOFF = H_SOFF + 2, H_KIND = H_QOFF + 2, H_ROFF, H_RCNT, H_RFAIL, H_RFLAGS, H_WORDS };
    enum { ROW_DEAD = 0, ROW_BUFFERED = 1, ROW_GLOBAL = 2 };
    __shared__ int s_hdr[4][EXT_RPW + 1][H_WORDS];
    const int w = threadIdx.x >> 6, l = lane_id();
    int *sref = s_ref[w], *sqry = s_qry[w]; uint32_t *scig = s_cig[w];
    int *bvar = s_bvar[w]; uint32_t *baq = s_baq[w];
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
    int h_start = 0, h_lq = 0, h_n = 0; bool h_live = false; unsigned h_cp = 0; unsigned long long h_soff = 0, h_qoff = 0;
    if (l < nq) {
        const int r = r0 + l; h_start = R.ref_start[r]; h_lq = R.l_qseq[r]; h_soff = R.seq_off[r]; h_qoff = R.qual_off[r]; h_cp = R.cp_off[r]; h_n = R.cp_n[r];
        const int flag = R.flag[r];
        h_live = !(R.mapq[r] < mapping_quality || (flag & 0x4) || (flag & 0x100) || (flag & 0x400) || h_start >= V.last_pos);
    }
    const unsigned live_mask = (unsigned)__ballot(h_live) & 15u;
    const uint32_t *cg = R.cigp;                                       // alignment q's words: cg + 8 * (its first lane-chunk), cp_n of them

    // what was requested ahead for segment (pf_q, pf_seg): CIGAR words, op after the segment, candidate records, predecessor position
    uint32_t pw[8]; uint32_t pnext = 0xfu; uint2 pvr = make_uint2(0x7fffffffu, 0u); int ppv = -1;
    int pf_q = -1, pf_seg = 0; bool pf_vr_ok = false;
#pragma unroll
    for (int u = 0; u < 8; ++u) pw[u] = 6u;
    int q = live_mask ? __builtin_ctz(live_mask) : nq;
    if (q < nq) {                                                      // first segment of the first alignment: on its way while the bounds are searched
        const unsigned long long crel = 8ull * (unsigned)__shfl((int)h_cp, q); const int ncq = __shfl(h_n, q);
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
        h[H_CP] = (int)h_cp; h[H_NCIG] = h_n;
        if (l < 4) {
            h[H_START] = h_start; h[H_LQ] = h_lq; h[H_V0] = h_v0;
            h[H_SOFF] = (int)(unsigned)h_soff; h[H_SOFF + 1] = (int)(unsigned)(h_soff >> 32); h[H_QOFF] = (int)(unsigned)h_qoff; h[H_QOFF + 1] = (int)(unsigned)(h_qoff >> 32);
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
        const int start = HU(H_START), lq = HU(H_LQ), n_cig = HU(H_NCIG);
        const uint32_t *cig = cg + 8ull * (unsigned)HU(H_CP);
        const unsigned long long soff = (unsigned)HU(H_SOFF) | ((unsigned long long)(unsigned)HU(H_SOFF + 1) << 32), qoff = (unsigned)HU(H_QOFF) | ((unsigned long long)(unsigned)HU(H_QOFF + 1) << 32);
        int vcur = HU(H_V0);
#undef HU
        const unsigned rest = live_mask >> (q + 1);
        const int qn = rest ? q + 1 + __builtin_ctz(rest) : nq;                 // next alignment to walk
        const int *hn = hdr + qn * H_WORDS;                                     // (row nq exists; nothing of it is used)
        const int n_cig_n = qn < nq ? hn[H_NCIG] : 0;

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
                            const unsigned e = C.fixed + atomicAdd(C.n_ev, 1u);     // (rare path: one atomic per event, behind the jobs' own slots)
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
                const uint32_t *cign = same ? cig : cg + 8ull * (unsigned)hn[H_CP];
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
                                    int code; sq_fetch(R.seq, R.qual, soff, qoff, qi, code, qv);
                                    const char base_c = nt16_char(code);
                                    if (base_c == ref_c) allele = 0; else if (base_c == alt_c) allele = 1;
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
                                        int code; sq_fetch(R.seq, R.qual, soff, qoff, qs, code, qv);
                                        const char base_c = nt16_char(code);
                                        if (base_c == ref_c) allele = 0; else if (base_c == alt_c) allele = 1;
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
                        for (int i = l; i < n_buf; i += 64) O.rec[g0 + i] = ObsRec{O.snp_u ? O.snp_u[bvar[i]] : bvar[i], baq[i]};
                        if (l < q) { int *hp = hdr + l * H_WORDS; if (hp[H_KIND] == ROW_BUFFERED) { hp[H_KIND] = ROW_GLOBAL; hp[H_ROFF] = (int)(uint32_t)(g0 + (unsigned)hp[H_ROFF]); } }
                        direct_base = g0 + row_start;
                    }
                    wave_sync();
                    direct = true; n_buf = 0;
                }
                if (emit) {
                    const int rank = n_emit + __popcll(em & lanemask_lt());
                    unsigned rk = 0;                                        // rank inside the variant's list of observations (see k_extract_phase)
                    if (var_cnt) { rk = atomicAdd(&var_cnt[v], 1u); if (rk > 0x3fffffu) atomicOr(&cnt->err, (unsigned)LPS_ERR_KEY_RANGE); }
                    const uint32_t aqw = (uint32_t)pack_aq(allele, qv) | (rk << 10);
                    if (direct) { if (!arena_full) O.rec[direct_base + rank] = ObsRec{O.snp_u ? O.snp_u[v] : v, aqw}; }
                    else { bvar[row_start + rank] = v; baq[row_start + rank] = aqw; }
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
        if (var_cnt && fail_op != 0x7fffffff && n_emit > 0) {                    // the read is dropped (:1453-1455) after its observations were counted: taken off again
            __threadfence();
            for (int i = l; i < n_emit; i += 64) atomicAdd(&var_del[direct ? O.rec[direct_base + i].var : bvar[row_start + i]], 1u);
            if (l == 0) atomicAdd(&cnt->n_abandoned, (unsigned)n_emit);
        }
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
    if (!arena_full) for (int i = l; i < n_buf; i += 64) O.rec[g0 + i] = ObsRec{O.snp_u ? O.snp_u[bvar[i]] : bvar[i], baq[i]};
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
                          int mapping_quality, LpsCounters *cnt, uint32_t *redo_list, unsigned *n_redo, uint32_t *var_cnt, uint32_t *var_del, hipStream_t s) {
    if (R.n == 0) return;
    const int n_jobs = (R.n + EXT_RPW - 1) / EXT_RPW;
    hipLaunchKernelGGL(k_extract_phase, dim3(n_jobs), dim3(64), 0, s, V, R, O, C, mapping_quality, cnt, redo_list, n_redo, var_cnt, var_del);
    // jobs the lane-chunk table could not hold queued themselves (an alignment of more than ~200 kb of CIGAR; none with ordinary read lengths): a small grid drains the queue
    hipLaunchKernelGGL(k_extract_redo, dim3(std::min(256, (n_jobs + 3) / 4)), dim3(256), 0, s, V, R, O, C, mapping_quality, cnt, redo_list, n_redo, var_cnt, var_del);
}
