// lps_kernels.h — launch wrappers of the HIP kernels (definitions in lps_extract.hip / lps_graph.hip).
#pragma once
#include "lps_common.h"

// Device views -----------------------------------------------------------------------------------------
struct VarView {           // variant table, position-sorted (Appendix B of SURVEY.md)
    int n;
    const int32_t *pos;
    const uint8_t *ref0, *alt0;
    const uint16_t *ref_len, *alt_len;
    uint8_t *danger;       // getVariants_markindel
    uint8_t *hpoly;        // homopolymerLength at the site
    uint8_t *erased;       // filterSNP
    const uint8_t *hp1_is_alt;   // haplotag
    const int32_t *phase_set;    // haplotag
    const uint8_t *somatic_role, *derive_hp;   // somatic tagging
    const uint8_t *tumor_kind;                 // somatic extraction
    const uint2 *rec;      // packed per-variant record {pos, attr} used by the extraction kernels (see VREC_*)
    const int32_t *bucket; // coarse index: bucket[b] = first variant with pos >= (b << LPS_BUCKET_SHIFT); n_bucket+1 entries
    int n_bucket;
    const char *ref;       // reference bases
    long long ref_len_eff; // FastaParser truncation [0,last+5]
    int last_pos;
};

struct ReadView {
    int n;
    const int32_t *ref_start, *l_qseq;
    const uint16_t *flag;
    const uint8_t *mapq;
    const uint32_t *name_id;
    const uint64_t *seq_off, *qual_off;   // byte offsets of a read's 4-bit bases / its qualities inside seq / qual (for BAM-record pushes both point into the record blob: used in place)
    const uint8_t *seq, *qual;
    // The CIGAR words, resident in lane-chunks of 8 (lps_reads.hip): every alignment starts on a multiple of 8 words and is padded to one with op P,
    // length 0.  Read r = chunks [cp_off[r], cp_off[r + 1]), cp_n[r] real words.  Written in this layout by whatever makes the alignments resident
    // (lps_push_reads*, the BAM record decoder): there is no second copy and no pass between a push and the kernels.
    const uint32_t *cigp;
    const uint32_t *cp_off;
    const int32_t *cp_n;
    const int32_t *v0;         // first variant at or after the alignment's start (k_variant_table's last workgroups: one thread per alignment, before the wave-per-job kernels)
#ifdef __HIPCC__
    __device__ __forceinline__ const uint32_t *cig(int r) const { return cigp + 8ull * cp_off[r]; }
#endif
};
// lps_reads.hip: a pushed batch's CIGAR words (dense, cigar_off[i]..cigar_off[i+1]) into the resident lane-chunk layout
void launch_cp_count(int n, const uint64_t *cigar_off, uint32_t *nch, int32_t *ncig, unsigned *too_long, hipStream_t s);
void launch_cp_pack(int n, const uint64_t *cigar_off, const uint32_t *cigar, const uint32_t *rel_off, uint32_t chunk_base, uint32_t *cp_off, uint32_t *cigp, hipStream_t s);
void launch_sum_i32(const int32_t *v, int n, unsigned long long *out, hipStream_t s);
#ifdef __HIPCC__
// base code (4 bits) and quality of query index qi of a read whose bases start at seq + soff and whose qualities start at qual + qoff (the BAM
// record's own encodings, read in place: two lines of HBM per site)
__device__ __forceinline__ void sq_fetch(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual, unsigned long long soff, unsigned long long qoff, int qi, int &code, int &qv) {
    // each of these lines is touched once per launch: a non-temporal load keeps it out of the caches' way (profiles/micro/gather_bench.hip: 53 against
    // 47 G random loads / s; k_extract_phase at chr1-50x 1.27 against 1.30 ms, same box)
    code = (__builtin_nontemporal_load(seq + soff + (unsigned)(qi >> 1)) >> ((~qi & 1) << 2)) & 15; qv = __builtin_nontemporal_load(qual + qoff + (unsigned)qi);
}
#endif

// one alignment's row of observations: 16 bytes, written by one lane of the extraction wave (four rows = one 64-byte line per wave)
struct __attribute__((aligned(16))) RowDesc {
    uint32_t off;          // first slot of the row in the observation arena
    int32_t cnt;           // observations (0: none, filtered out, or get_snp returned early)
    int32_t fail;          // CIGAR op index at which get_snp returned early (INT_MAX = none): clips of later ops do not count
    uint32_t flags;        // bit0: had observations before filterSNP erased them
};
// one observation: 8 bytes {variant index (or -1 - index once the CNV filter erased it), pack_aq(allele, quality)}
struct __attribute__((aligned(8))) ObsRec { int32_t var; uint32_t aq; };
// a clip event of getClip: position, op index << 1 | (op index != 0), alignment
struct ClipEv { int32_t pos; int32_t opidx_fb; int32_t read; };

struct ObsView {           // rows placed by atomic reservation: row r = [rows[r].off, rows[r].off + rows[r].cnt)
    RowDesc *rows;
    ObsRec *rec;
    unsigned long long arena_size;      // slots per arena; arena a covers [a*arena_size, (a+1)*arena_size)
    unsigned long long *arena_ctr;      // LPS_ARENAS counters, 8 u64 apart (one cache line each)
    int n_arenas;                       // arenas in use: min(LPS_ARENAS, workgroups)
    const int32_t *snp_u;               // SV / MOD rows co-phased: index of SNP row v in the union of the three tables - what the extraction then writes as `var` (else nullptr)
};

// Clip events of the kept alignments (filtered by RowDesc.fail afterwards).  k_extract_phase writes the events of job j into ITS OWN slots
// [EXT_CLIPS * j, EXT_CLIPS * (j + 1)) - a job holds at most that many - and marks the unused ones (read = -1): no counter.  (An appended list cost
// one returning atomic per job on ONE word: ~11 ns each, served one after the other - 0.7 ms of a 1.2 ms kernel at chr1-50x, whatever the rest of
// the kernel did.)  Only the general walker (k_extract_redo, rare) appends, behind the fixed part: slot `fixed` + atomicAdd(n_ev).
struct ClipView {
    ClipEv *ev; unsigned *n_ev; unsigned capacity; unsigned fixed;
};
#define EXT_CLIPS 16    // clip events a job of four alignments can hold: its alignments' first two and last two CIGAR words

// SV / MOD rows of `phase --sv-file / --mod-file` (lps_extra.hip): both tables merged by position; u / snp_u = index of a row / of SNP row i in
// the position-sorted union of all three tables, the index space of every stage after the extraction when such rows are present
struct ExtraView {
    int n;
    const int32_t *pos;        // ascending, distinct, none equal to a SNP position
    const int32_t *info;       // SV: SVLEN as in the VCF; MOD: row number (mod_off)
    const uint8_t *kind;       // 1 SV, 2 MOD
    const int32_t *u;
    const int32_t *snp_u;      // [V.n]
    const uint32_t *mod_off;   // reads listed at MOD row m: [mod_off[m], mod_off[m+1]) of mod_name (ascending name ids) / mod_flag (bit0 modified, bit1 reverse)
    const uint32_t *mod_name;
    const uint8_t *mod_flag;
    int sv_window; double sv_threshold;
    const int4 *rec;           // per row {pos, info, u | kind << 30, position of the last SNP row before it}: one load for k_extra_find
    const uint32_t *mod_pack;  // mod_name << 2 | mod_flag: the search finds the flags with the name
};
void launch_extra_merge(const VarView &V, const ReadView &R, const ObsView &O, const ExtraView &X, int32_t *x0, uint32_t *redo, unsigned *n_redo, int mapping_quality,
        LpsCounters *cnt, hipStream_t s);

// one launch: derived columns + packed records, the bucket index and (v0 != NULL) the first candidate row of every alignment
void launch_variant_prep(const VarView &V, int is_ont, int32_t *bucket, uint2 *rec, hipStream_t s, const int32_t *ref_start = nullptr, int n_reads = 0, int32_t *v0 = nullptr);

// attr word of a packed variant record: bits 0-7 REF[0], 8-15 ALT[0], 16-17 kind (0 SNP, 1 insertion, 2 deletion,
// 3 other), 18 danger, 19 erased by filterSNP, 20 homopolymerLength >= 3
#define VREC_KIND(a) (((a) >> 16) & 3u)
#define VREC_DANGER (1u << 18)
#define VREC_ERASED (1u << 19)
#define VREC_HPOLY3 (1u << 20)
#define VREC_HP1ALT (1u << 21)   /* haplotag: haplotype 1 carries ALT */
#define VREC_ROLE(a) (((a) >> 22) & 3u)   /* somatic tagging: 0 normal phased-het row, 1 somatic call, 2 inert tumor row */
#define VREC_DERIVE(a) (((a) >> 24) & 3u) /* somaticReadDeriveByHP of role-1 rows */
#define VREC_TKIND(a) (((a) >> 26) & 7u)  /* somatic extraction: TUMOR row kind at this position (0 none, 1 SNP, 2 INS, 3 DEL, 4 other) */
void launch_extract_phase(const VarView &V, const ReadView &R, const ObsView &O, const ClipView &C,
                          int mapping_quality, LpsCounters *cnt, uint32_t *redo_list, unsigned *n_redo, uint32_t *var_cnt /* NULL: observations are counted later */,
                                  uint32_t *var_del, hipStream_t s);

// ---- device helpers shared by the extraction (phase) and scoring (haplotag) kernels
#ifdef __HIPCC__
__device__ __forceinline__ bool op_consumes_ref(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
__device__ __forceinline__ bool op_consumes_query(int op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }
__device__ __forceinline__ bool op_is_match(int op) { return op == 0 || op == 7 || op == 8; }


// ---- CIGAR staging, 8 consecutive ops per lane (a 512-op segment per wave): per-op work is a handful of VALU instructions and the wave-wide
//      prefix scan runs once per segment instead of once per 64 ops.
struct __attribute__((packed, aligned(4))) LpsU4 { uint32_t x, y, z, w; };
// ops [i0, i0+8) of a CIGAR of n ops; 6u (op P, length 0: consumes nothing) beyond its end.  May read up to 7 words past the end of the
// CIGAR array: DevBuf allocations carry 64 B of slack.  In two halves, so that a segment requested AHEAD is not waited for where it is requested:
// request_ops8 only issues the loads; finish_ops8 (same i0, n) blanks the words past the end where the words are consumed.
__device__ __forceinline__ void request_ops8(const uint32_t *cig, int i0, int n, uint32_t (&w)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = 6u;
    if (i0 < n) {
        const LpsU4 a = *reinterpret_cast<const LpsU4 *>(cig + i0), b = *reinterpret_cast<const LpsU4 *>(cig + i0 + 4);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    }
}
__device__ __forceinline__ void finish_ops8(int i0, int n, uint32_t (&w)[8]) {
    if (i0 < n && i0 + 8 > n) {
#pragma unroll
        for (int k = 1; k < 8; ++k) if (i0 + k >= n) w[k] = 6u;
    }
}
__device__ __forceinline__ void load_ops8(const uint32_t *cig, int i0, int n, uint32_t (&w)[8]) { request_ops8(cig, i0, n, w); finish_ops8(i0, n, w); }
// bit 0: the op consumes the reference (M D N = X: 0x18D), bit 16: it consumes the query (M I S = X: 0x193); ops 9..15 consume nothing
__device__ __forceinline__ unsigned op_consume_bits(unsigned op) { return ((0x193u << 16) | 0x18Du) >> op; }
__device__ __forceinline__ int bit_mask(unsigned x, int bit) { return (int)(x << (31 - bit)) >> 31; }      // 0 or -1 (v_bfe_i32)

// ---- the stream walk of k_extract_phase / k_haplotag_stream over CIGAR words in lane-chunks (lps_reads.hip)
// Properties of an op as ONE v_bfe_u32 on the raw word: the instruction takes bits [4:0] of the word as bit index - the op code plus 16 x (bit 0 of the
// length) - into a mask that repeats the 16 per-op bits in both halves.  No `& 15`, no shifted table.
#define LPS_RMASK2 0x018D018Du      // consumes the reference: M D N = X
#define LPS_QMASK2 0x01930193u      // consumes the query:     M I S = X
#define LPS_CLIPMASK2 0x00300030u   // S H
#define LPS_BADMASK2 0xfe00fe00u    // op codes the reference rejects (ParsingBam.cpp:1625-1628)
__device__ __forceinline__ unsigned op_bit(unsigned mask2, uint32_t word) { return __builtin_amdgcn_ubfe(mask2, word, 1u); }
__device__ __forceinline__ int ref_len_of(uint32_t w) { return (int)(w >> 4) & -(int)op_bit(LPS_RMASK2, w); }
// One round: the lane's 8 words (lane-chunk cid of a stream of TC chunks) -> its reference / query advance, scanned over the wave into the stream
// coordinates of the chunk's first word (table entry cid >> shift, for chunks that are multiples of 1 << shift), the carries moved on.  `special`
// counts the words whose op is in SPECIAL2, `big` ORs the words: lengths of 2^24 and more (bit 28 and up) are outside the 24-bit multiply below
// and send the job to the general walker.  Lanes past the stream's end hold a copy of its last chunk and count for nothing.
template <unsigned SPECIAL2>
__device__ __forceinline__ void stream_round(const uint32_t (&w)[8], const int cid, const int TC, const int shift, int2 *s_tab, int &carry_r, int &carry_q,
                                             unsigned &special, uint32_t &big) {
    unsigned rt = 0, qt = 0, sp = 0; uint32_t bg = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t x = w[k]; const unsigned len = x >> 4;
        rt = __umul24(len, op_bit(LPS_RMASK2, x)) + rt; qt = __umul24(len, op_bit(LPS_QMASK2, x)) + qt;
        sp += op_bit(SPECIAL2, x); bg |= x;
    }
    const bool live = cid < TC;
    rt = live ? rt : 0u; qt = live ? qt : 0u; special += live ? sp : 0u; big |= live ? bg : 0u;
    const int ir = wave_incl_scan_dpp((int)rt), iq = wave_incl_scan_dpp((int)qt);
    if (live && (cid & ((1 << shift) - 1)) == 0) s_tab[cid >> shift] = make_int2(carry_r + ir - (int)rt, carry_q + iq - (int)qt);
    carry_r += __builtin_amdgcn_readlane(ir, 63); carry_q += __builtin_amdgcn_readlane(iq, 63);
}

// Stages the segment whose words are in w (lane l owns ops 8l..8l+7 of it) in LDS: sref/sqry = reference / query position at which the op
// starts, scig = the raw word.  ref_pos / q_pos (wave-uniform) advance over the segment.  Returns, per lane, the set of op codes its 8 words
// hold (bit op; the padding past the segment sets bit 6): LPS_OPS_CLIP = a soft / hard clip is among them (the caller then looks closer),
// LPS_OPS_BAD = an op code the reference rejects (ParsingBam.cpp:1625-1628).  my_ref = reference position of the lane's first op.
#define LPS_OPS_CLIP 0x30u
#define LPS_OPS_BAD 0xfe00u
__device__ __forceinline__ unsigned stage_ops8(const uint32_t (&w)[8], int l, int &ref_pos, int &q_pos, int *sref, int *sqry, uint32_t *scig, int &my_ref) {
    // lane totals first, positions in a second sweep over the same 8 words: holding 16 prefix values across the wave scan would cost the
    // kernels a wave of occupancy, recomputing them costs 40 instructions per segment
    int rt = 0, qt = 0; unsigned seen = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned op = w[k] & 15u;
        const unsigned t = op_consume_bits(op); const int len = (int)(w[k] >> 4);
        rt += len & bit_mask(t, 0); qt += len & bit_mask(t, 16);
        seen |= 1u << op;
    }
    const int ir = wave_incl_scan_dpp(rt), iq = wave_incl_scan_dpp(qt);
    my_ref = ref_pos + ir - rt; int rr = my_ref, qq = q_pos + iq - qt;
    int4 *dr = reinterpret_cast<int4 *>(sref + 8 * l), *dq = reinterpret_cast<int4 *>(sqry + 8 * l); uint4 *dc = reinterpret_cast<uint4 *>(scig + 8 * l);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int pr[4], pq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned t = op_consume_bits(w[4 * h + k] & 15u); const int len = (int)(w[4 * h + k] >> 4);
            pr[k] = rr; pq[k] = qq; rr += len & bit_mask(t, 0); qq += len & bit_mask(t, 16);
        }
        dr[h] = make_int4(pr[0], pr[1], pr[2], pr[3]); dq[h] = make_int4(pq[0], pq[1], pq[2], pq[3]);
        dc[h] = make_uint4(w[4 * h], w[4 * h + 1], w[4 * h + 2], w[4 * h + 3]);
    }
    ref_pos += __shfl(ir, 63); q_pos += __shfl(iq, 63);
    return seen;
}

// first variant with pos >= key: one bucket lookup narrows the range to the variants of a 1-kb window, then one
// 64-wide probe round (falls back to the 64-ary search for very dense windows).  Wave-uniform.
__device__ __forceinline__ int var_lower_bound(const VarView &V, int key) {
    if (key < 0) return 0;
    const int b = key >> LPS_BUCKET_SHIFT;
    if (b >= V.n_bucket) return wave_lower_bound(V.pos, V.bucket[V.n_bucket], V.n, key);
    return wave_lower_bound(V.pos, V.bucket[b], V.bucket[b + 1], key);
}

// first variant with pos >= key, searched by ONE lane (the planning step runs four of these side by side); == var_lower_bound
__device__ __forceinline__ int lane_var_lower_bound(const VarView &V, int key) {
    if (key < 0) return 0;
    const int b = key >> LPS_BUCKET_SHIFT;
    int lo, hi;
    if (b >= V.n_bucket) { lo = V.bucket[V.n_bucket]; hi = V.n; } else { lo = V.bucket[b]; hi = V.bucket[b + 1]; }
    while (lo < hi) { const int m = (lo + hi) >> 1; if (V.pos[m] < key) lo = m + 1; else hi = m; }
    return lo;
}

// what a candidate lane needs to know about its alignment: three 16-byte LDS reads
struct __attribute__((aligned(16))) ExtHdr {
    int crel, ncig, c0, nch;               // first CIGAR word (relative to the job's first), CIGAR words, first chunk in the table, chunks it touches
    int vadj, lq, ds, dq;                  // variant of flattened candidate i = vadj + i; l_qseq; stream - true reference coordinate; stream query coordinate of the read's first base
    unsigned long long soff, qoff;         // where the read's bases / qualities start in ReadView::seq / qual
};

// waits for the loads into w[] (an empty asm that reads the registers): see the walk loop of k_extract_phase
__device__ __forceinline__ void drain8(const uint32_t (&w)[8]) {
    asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
}

// the c-th of four wave-uniform scalars, c = index of the first flattened candidate of alignments 1..3 (per-lane compare against thresholds)
#define SELC(c, t, a) ((c) >= (t)[3] ? (a)[3] : ((c) >= (t)[2] ? (a)[2] : ((c) >= (t)[1] ? (a)[1] : (a)[0])))
#define SEL4(q, a) ((q) >= 3 ? (a)[3] : ((q) >= 2 ? (a)[2] : ((q) >= 1 ? (a)[1] : (a)[0])))
#endif

struct HapOut { uint8_t *status; int32_t *hp1, *hp2; uint8_t *n_ps; int32_t *ps_min; int32_t *hp3, *d1, *d2;
                int32_t *site; uint8_t *read_hp; double pct_thr;     // site counters [nV][LPS_SITE_COUNTERS], per-read hp of the pass
                // germline haplotag (mode 0): ONE 16-byte record per read instead of five arrays, the read-level decision taken on the GPU:
                //   word 0 = status | n_ps << 8 | HP << 16 | PQ << 24 (PQ 255: votes of 64 or more, the host computes it), hp1, hp2 (votes included), ps_min
                uint4 *rec; const int *pq_tab /* [64][64]: PQ of (min, max) votes, built by the host's libm */; const int32_t *votes1, *votes2;
                // normal-BAM extraction on the stream walker (mode 2): the (tumor row, alignment) pairs it touches, appended in LPS_TARENAS arenas like
                // TumOut's lists - the rows' ReadHpCount needs the read's FINAL haplotype and is added from this list (launch_normal_pair_sites), no second walk
                unsigned long long *pair_ctr; long long pair_arena; int32_t *apair_site, *apair_read; };
void launch_normal_pair_sites(const HapOut &H, hipStream_t s);
void launch_haplotag(const VarView &V, const ReadView &R, const HapOut &H, int mapping_quality, int tag_supplementary,
                     int mode, LpsCounters *cnt, hipStream_t s, bool general = false);   // general: the per-op-prefix walker also for the germline pass (what the stream walk cannot take);   // mode 0 haplotag, 1 somatic tag, 2 normal extraction, 3 its read-HP pass

// tumor-BAM extraction (lps_somatic.hip)
struct TumOut {
    int32_t *site;                 // [nV][LPS_TSITE_COUNTERS]
    uint8_t *status; int32_t *hp1, *hp2, *hp3; uint8_t *hp; uint8_t *n_ps; int32_t *ps_min; int32_t *end_pos, *read_len; uint8_t *has_site;
    // Both lists of the passes - (site, read, base HP) pairs and window hits - are appended in LPS_TARENAS ARENAS, arena = workgroup % LPS_TARENAS,
    // every arena with its own counter on its own 128-byte line: a returning atomic on ONE word is served one after the other, ~11 ns each
    // (two lists x 370 k appending waves = 7 ms of the passes' 9 at 160 Mb), on 64 words they are served side by side.  tot[0..3] = pairs, the
    // fullest pair arena, hits, the fullest hit arena (k_tumor_totals).
    unsigned long long *pair_ctr, *hit_ctr, *tot;
    long long pair_arena, hit_arena;              // slots per arena
    int32_t *apair_site, *apair_read; uint8_t *apair_hp;   // the pairs in their arenas; compacted into pair_site / pair_read / pair_hp (k_tumor_pairs_out)
    long long pair_cap, win_cap;
    int32_t *pair_site, *pair_read; uint8_t *pair_hp;
    int32_t *win_site; uint8_t *win_allele; int16_t *win_offset; uint8_t *win_base;
    double pct_thr;
    int4 *hits; int *hit_rp;                      // the hits of pass 0: {row, alignment, CIGAR word index, offset inside the op | allele << 30} + the query position there
    unsigned long long *win_total;                // entries of the window list so far (k_tumor_windows reserves a wave's stretch with one atomic)
};
// the +-100 bp difference windows of the hits pass 0 listed (getWindowsDiffRef, SomaticVarCaller.cpp:654-710): ONE THREAD per (hit, direction)
void launch_tumor_windows(const VarView &V, const ReadView &R, const TumOut &T, hipStream_t s);
void launch_tumor_pairs_out(const TumOut &T, hipStream_t s);     // after pass 1: totals + the pairs out of their arenas into the caller's list
#define LPS_TARENAS 64
void launch_tumor_extract(const VarView &V, const ReadView &R, const TumOut &T, int mapping_quality, int tag_supplementary, int pass,
                          LpsCounters *cnt, hipStream_t s, bool general = false);
