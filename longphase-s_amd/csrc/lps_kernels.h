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
    const uint64_t *cigar_off, *seq_off, *qual_off;
    const uint32_t *cigar;
    const uint8_t *seq, *qual;
};

struct ObsView {           // CSR rows placed by atomic reservation: row r = [row_off[r], row_off[r]+row_cnt[r])
    uint32_t *row_off;
    int32_t *row_cnt;
    int32_t *row_fail;     // CIGAR op index at which get_snp returned early (INT_MAX = none)
    uint8_t *row_flags;    // bit0: had observations before filterSNP erased them
    int32_t *var;          // variant index
    uint16_t *aq;          // pack_aq(allele, quality)
    unsigned long long arena_size;      // slots per arena; arena a covers [a*arena_size, (a+1)*arena_size)
    unsigned long long *arena_ctr;      // LPS_ARENAS counters, 8 u64 apart (one cache line each)
    int n_arenas;                       // arenas in use: min(LPS_ARENAS, workgroups)
};

struct ClipView {          // clip events in LPS_CLIP_SLOTS fixed slots per alignment; filtered by row_fail afterwards
    int32_t *pos; int32_t *opidx_fb;   // opidx<<1 | (opidx!=0), -1 = unused slot
};

void launch_variant_prep(const VarView &V, int is_ont, int32_t *bucket, uint2 *rec, hipStream_t s);

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
                          int mapping_quality, LpsCounters *cnt, hipStream_t s);

// ---- device helpers shared by the extraction (phase) and scoring (haplotag) kernels
#ifdef __HIPCC__
__device__ __forceinline__ bool op_consumes_ref(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
__device__ __forceinline__ bool op_consumes_query(int op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }
__device__ __forceinline__ bool op_is_match(int op) { return op == 0 || op == 7 || op == 8; }


// first variant with pos >= key: one bucket lookup narrows the range to the variants of a 1-kb window, then one
// 64-wide probe round (falls back to the 64-ary search for very dense windows).  Wave-uniform.
__device__ __forceinline__ int var_lower_bound(const VarView &V, int key) {
    if (key < 0) return 0;
    const int b = key >> LPS_BUCKET_SHIFT;
    if (b >= V.n_bucket) return wave_lower_bound(V.pos, V.bucket[V.n_bucket], V.n, key);
    return wave_lower_bound(V.pos, V.bucket[b], V.bucket[b + 1], key);
}

#endif

struct HapOut { uint8_t *status; int32_t *hp1, *hp2; uint8_t *n_ps; int32_t *ps_min; int32_t *hp3, *d1, *d2;
                int32_t *site; uint8_t *read_hp; double pct_thr; };   // site counters [nV][LPS_SITE_COUNTERS], per-read hp of the pass
void launch_haplotag(const VarView &V, const ReadView &R, const HapOut &H, int mapping_quality, int tag_supplementary,
                     int mode, LpsCounters *cnt, hipStream_t s);   // mode 0 haplotag, 1 somatic tag, 2 normal extraction, 3 its read-HP pass

// tumor-BAM extraction (lps_somatic.hip)
struct TumOut {
    int32_t *site;                 // [nV][LPS_TSITE_COUNTERS]
    uint8_t *status; int32_t *hp1, *hp2, *hp3; uint8_t *hp; uint8_t *n_ps; int32_t *ps_min; int32_t *end_pos, *read_len; uint8_t *has_site;
    unsigned long long *list_ctr;  // [0] pairs, [1] windows
    long long pair_cap, win_cap;
    int32_t *pair_site, *pair_read; uint8_t *pair_hp;
    int32_t *win_site; uint8_t *win_allele; int16_t *win_offset; uint8_t *win_base;
    double pct_thr;
};
void launch_tumor_extract(const VarView &V, const ReadView &R, const TumOut &T, int mapping_quality, int tag_supplementary, int pass,
                          LpsCounters *cnt, hipStream_t s);
