/* lps_abi.h — C-ABI of liblps_hip.so: the MI355X-native (gfx950) implementation of LongPhase-S's
 * read->variant allele scoring + haplotype-graph phasing hot path.
 *
 * The reference has no FFI layer: the path is reached by direct C++ calls from its per-chromosome OpenMP
 * loops.  This header is the seam a maintainer binds instead (INTEGRATION.md shows the call-site patch):
 *
 *   phase     src/phase/PhasingProcess.cpp:128-158   BamParser::direct_detect_alleles -> SnpParser::filterSNP
 *                                                    -> Clip -> VairiantGraph::addEdge/phasingProcess/exportResult
 *   haplotag  src/haplotag/HaplotagParsingBam.cpp:482 ChromosomeProcessor::processRead
 *                                                    -> GermlineHaplotagChrProcessor::judgeHaplotype
 *                                                       (src/haplotag/HaplotagProcess.cpp:363)
 *
 * Conventions: plain pointers + sizes, no C++/torch types.  All input pointers are HOST pointers unless a
 * function name ends in _device.  The caller owns every buffer it passes for the duration of the call; the
 * library owns device memory inside lps_ctx.  Functions return 0 on success, <0 on error
 * (lps_last_error(ctx) gives the message; the reference prints to stderr and exit(1)s in the same places).
 * A ctx is single-threaded and bound to one GPU; one ctx processes one chromosome at a time.
 */
#ifndef LPS_ABI_H
#define LPS_ABI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPS_ABI_VERSION 21
#define LPS_MAX_ADJACENT 63 /* upper bound for lps_params.connect_adjacent (reference default 35) */

typedef struct lps_ctx lps_ctx;

/* PhasingParameters (src/phase/PhasingProcess.h:7-41) + haplotag thresholds (src/haplotag/Haplotag.cpp:60-72).
 * Defaults in comments are the reference's (src/phase/Phasing.cpp:88-116). */
typedef struct lps_params {
    int32_t is_ont;            /* --ont=1 / --pb=0 : ONT enables SnpParser::filterSNP               */
    int32_t phase_indel;       /* --indels (informational: indel rows simply appear in the table)   */
    int32_t distance;          /* -d 300000                                                          */
    int32_t connect_adjacent;  /* -a 35                                                              */
    int32_t mapping_quality;   /* -q 1                                                               */
    int32_t base_quality;      /* -p 12                                                              */
    double edge_weight;        /* -e 0.1                                                             */
    double snp_confidence;     /* -n 0.75                                                            */
    double read_confidence;    /* -m 0.65                                                            */
    double edge_threshold;     /* -1 0.7                                                             */
    double overlap_threshold;  /* -L 0.2                                                             */
    /* haplotag */
    double percentage_threshold; /* -p 0.6                                                           */
    int32_t tag_supplementary;   /* --tagSupplementary                                               */
    int32_t reserved;
} lps_params;

void lps_default_params(lps_params *p);

/* Per-chromosome variant table = std::map<int,RefAlt> (src/phase/ParsingBam.h:16-23,58), position-sorted.
 * ref0/alt0 are the first characters of REF/ALT exactly as they stand in the VCF; ref_len/alt_len the allele
 * string lengths (1/1 = SNP, 1/>1 = insertion, >1/1 = deletion).
 * haplotag only (src/haplotag/HaplotagType.h:110-144 VarData): hp1_is_alt = 1 when HP1 carries ALT ("1|0"),
 * phase_set = PS value.  Pass NULL for both in `phase`. */
typedef struct lps_variant_table {
    int64_t n;
    const int32_t *pos;      /* 0-based, strictly increasing */
    const uint8_t *ref0;
    const uint8_t *alt0;
    const uint16_t *ref_len;
    const uint16_t *alt_len;
    const uint8_t *hp1_is_alt; /* haplotag */
    const int32_t *phase_set;  /* haplotag */
    /* somatic tagging only (merged normal+tumor map, src/haplotag/HaplotagType.h:146-162 MultiGenomeVar), else NULL:
     *  somatic_role[i]: 0 = row of the NORMAL phased-het VCF (alleles/hp1_is_alt/phase_set are the normal ones; a tumor row
     *                       at the same position is shadowed exactly as in judgeSomaticSnpHap, HaplotagStrategy.cpp:315-389)
     *                   1 = tumor-only row flagged isSomaticVariant by the caller (SomaticVarCaller::getSomaticFlag :2397-2412)
     *                   2 = tumor-only row that is not a somatic call (kept: it still defines the last variant position)
     *  derive_hp[i]   : MultiGenomeVar::somaticReadDeriveByHP of role-1 rows (0 none, 1 H1, 2 H2) */
    const uint8_t *somatic_role;
    const uint8_t *derive_hp;
    /* somatic extraction (a20/a21): tumor_kind[i] = 0 no TUMOR row at this position, 1 SNP, 2 insertion, 3 deletion, 4 other.
     * Rows with somatic_role 0 additionally carry a NORMAL phased-het row (main alleles); tumor-only rows use the tumor
     * alleles as main alleles. */
    const uint8_t *tumor_kind;
} lps_variant_table;

/* Decoded alignments of ONE chromosome in BAM (coordinate) order = what sam_itr_multi_next hands to
 * BamParser::get_snp (src/phase/ParsingBam.cpp:1279-1293).  SoA; variable-length parts are the BAM record's
 * own encodings so packing is a memcpy of bam_get_cigar/bam_get_seq/bam_get_qual:
 *   cigar  uint32 words, oplen<<4|op            (cigar_off[i]..cigar_off[i+1])
 *   seq    4-bit packed, (l_qseq+1)/2 bytes/read (seq_off[i].. byte offsets)
 *   qual   l_qseq bytes/read                     (qual_off[i]..)
 * name_id: rank of the read name among all names of the batch set under byte-wise std::string ordering
 * (equal names <=> equal id).  The reference iterates merged reads in std::map<std::string,...> order
 * (src/phase/PhasingGraph.cpp:697,848) and float edge sums depend on that order.
 * Any order-preserving integer works (ids need not be dense). */
typedef struct lps_read_batch {
    int64_t n_reads;
    const int32_t *ref_start;  /* core.pos */
    const uint16_t *flag;      /* core.flag */
    const uint8_t *mapq;       /* core.qual */
    const int32_t *l_qseq;     /* core.l_qseq */
    const uint32_t *name_id;
    const uint64_t *cigar_off; /* n_reads+1 */
    const uint32_t *cigar;
    const uint64_t *seq_off;   /* n_reads+1 */
    const uint8_t *seq;
    const uint64_t *qual_off;  /* n_reads+1 */
    const uint8_t *qual;
} lps_read_batch;

/* SV and MOD rows co-phased with the SNPs (`phase --sv-file / --mod-file`): what BamParser holds next to the SNP map
 * (src/phase/ParsingBam.cpp:1207-1235) - SV_map[chr] = (start, SVLEN) pairs of SVParser (:915-1017), currentMod = per position the reads
 * METHParser listed (:1685-1786).  Both become graph nodes at their own positions; get_snp's SV branch (:1397-1434) calls a read ALT when an
 * insertion / deletion of about the SV's length lies within sv_window CIGAR operations of the operation that reaches the row, its MOD branch
 * (:1373-1395) takes the allele from the read lists.
 *   sv_pos   0-based = VCF POS - 1 (:1354), strictly increasing; sv_len = SVLEN as written in the VCF (sign kept)
 *   mod_pos  0-based representative position (:1709-1711), strictly increasing
 *   mod_off  n_mod+1 offsets into mod_name / mod_flag; mod_name = name_id of the listed reads (same id space as lps_read_batch.name_id;
 *            names that occur in no alignment can be left out), strictly increasing inside a row; mod_flag bit0 = listed under MR= (modified),
 *            bit1 = RS=N (reverse strand)
 * No position may occur in two of the three tables (SNP, SV, MOD): the reference's three-cursor loop does not terminate on such input
 * (none of its branches takes a row that ties with another cursor), the library refuses it. */
typedef struct lps_extra_variants {
    int64_t n_sv;
    const int32_t *sv_pos;
    const int32_t *sv_len;
    int64_t n_mod;
    const int32_t *mod_pos;
    const uint64_t *mod_off;
    const uint32_t *mod_name;
    const uint8_t *mod_flag;
    int32_t sv_window;      /* --svWindow 20    (src/phase/Phasing.cpp:113) */
    int32_t reserved;
    double sv_threshold;    /* --svThreshold 0.1 (src/phase/Phasing.cpp:114) */
} lps_extra_variants;

/* Result of one chromosome = PhasingResult entries (src/shared/Util.h:18-24) indexed like the variant
 * table: phase_set[i] = PS (block start position + 1) or 0 when variant i is not phased;
 * gt[i] = 0 for "0|1", 1 for "1|0" (only meaningful where phase_set[i] != 0).  Caller allocates n entries. */
typedef struct lps_phase_result {
    int64_t n;
    int32_t *phase_set;
    uint8_t *gt;
} lps_phase_result;

/* Per-read haplotag scores (inputs of GermlineHaplotagStrategy::judgeReadHap,
 * src/haplotag/HaplotagStrategy.cpp:243-300).  Caller allocates n_reads entries of each.
 *  status: 0 scored; 1 low MAPQ; 2 unmapped; 3 secondary; 4 supplementary (untagged); 5 empty table;
 *          6 start beyond last variant   (the filter cascade of HaplotagParsingBam.cpp:453-486)
 *  hp1/hp2: hpCount[GERMLINE_H1/H2]; n_ps: number of distinct PS seen (saturating at 255); ps_min: smallest.
 *  hp/pq/ps: the decision of judgeReadHap (hp 0 = untagged, 1, 2).  pq uses the host's libm log10
 *   and is filled by the host side of the library from the integer counts. */
typedef struct lps_haplotag_result {
    int64_t n_reads;
    uint8_t *status;
    int32_t *hp1;
    int32_t *hp2;
    uint8_t *n_ps;
    int32_t *ps_min;
    uint8_t *hp;
    int32_t *pq;
    int32_t *ps;
} lps_haplotag_result;

/* Per-read result of the somatic tagging pass (SomaticHaplotagChrProcessor::judgeHaplotype,
 * src/somatic_haplotag/SomaticHaplotagProcess.cpp:310-459 + inheritHaplotype :461-527).  Caller allocates n_reads entries.
 *  status as in lps_haplotag_result; hp1/hp2/hp3 = hpCount[1..3]; derive_h1/h2 = H3 bases whose variant derives from H1/H2;
 *  hp = ReadHP code (0 untagged, 1 H1, 2 H2, 3 H3, 4 H4, 5 "1-1", 6 "1-2", 7 "2-1", 8 "2-2", HaplotagType.h:97-108);
 *  ps = PS value or -1 when no PS tag is written (:416-434); pq as written to the PQ tag. */
typedef struct lps_somatic_tag_result {
    int64_t n_reads;
    uint8_t *status;
    int32_t *hp1;
    int32_t *hp2;
    int32_t *hp3;
    int32_t *derive_h1;
    int32_t *derive_h2;
    uint8_t *n_ps;
    int32_t *ps_min;
    uint8_t *hp;
    int32_t *pq;
    int32_t *ps;
} lps_somatic_tag_result;

/* Per-site counters of the normal-BAM extraction pass (ExtractNorDataChrProcessor + ExtractNorDataCigarParser,
 * src/somatic_haplotag/SomaticVarCaller.cpp:123-293; PosBase, src/haplotag/HaplotagType.h:165-224), one row of
 * LPS_SITE_COUNTERS int32 per table row (zero where tumor_kind == 0).  The derived ratios of calculateBaseCommonInfo
 * (:13-40) are plain host arithmetic on these integers. */
#define LPS_SITE_COUNTERS 18
enum { LPS_SC_ALT = 0, LPS_SC_A, LPS_SC_C, LPS_SC_G, LPS_SC_T, LPS_SC_UNKNOWN, LPS_SC_DEPTH, LPS_SC_DEL,
       LPS_SC_MPQ_ALT, LPS_SC_MPQ_A, LPS_SC_MPQ_C, LPS_SC_MPQ_G, LPS_SC_MPQ_T, LPS_SC_MPQ_UNKNOWN, LPS_SC_MPQ_DEPTH,
       LPS_SC_READHP_UNTAG, LPS_SC_READHP_H1, LPS_SC_READHP_H2 };
typedef struct lps_site_counters {
    int64_t n;          /* = variant table size */
    int32_t *counters;  /* [n][LPS_SITE_COUNTERS] */
    int64_t n_reads;    /* optional per-read germline haplotype of the pass (0 untag, 1, 2); read_hp may be NULL */
    uint8_t *read_hp;
} lps_site_counters;

/* Tumor-BAM extraction pass (ExtractTumDataChrProcessor + ExtractTumDataCigarParser, src/somatic_haplotag/SomaticVarCaller.cpp:334-518,
 * 627-759; SomaticData, src/haplotag/HaplotagType.h:226-294).  Per table row LPS_TSITE_COUNTERS int32:
 *   [0..14]  PosBase integers in LPS_SC_* order          [15..23] base.ReadHpCount[ReadHP 0..8]
 *   [24] unTag [25] totalCleanHP3Read [26] pure_H1_1_read [27] pure_H2_1_read [28] pure_H3_read [29] Mixed_HP_read   (classifyReadsByCase)
 *   [30..38] somaticReadHpCount[ReadHP 0..8]              [39] alleleCount[REF] [40] alleleCount[ALT]
 * Per read (n_reads entries each): status (as lps_haplotag_result, never 1: low-MAPQ reads are not skipped by this pass), hp1..hp3 =
 * hpCount[1..3], hp = judgeSomaticReadHap code, n_ps/ps_min of the NORMAL phase sets, end_pos = reference position after the last
 * CIGAR op, read_len = query bases walked, has_site = the read is in readHpResultSet (covers >=1 TUMOR row with MAPQ >= q).
 * Variable-length lists (caller-allocated capacities; the call fails with -9 and sets n_* to the needed size when they do not fit):
 *   pairs   (site row, read index, base HP) = tumorPosReadCorrBaseHP (:448-456)
 *   windows (site row, allele 0/1, offset -100..100, read base char) = PosSomaticOffsetBase (:654-710, 729-737) */
#define LPS_TSITE_COUNTERS 41
typedef struct lps_tumor_extract_result {
    int64_t n;               /* variant table size */
    int32_t *site;           /* [n][LPS_TSITE_COUNTERS] */
    int64_t n_reads;
    uint8_t *status; int32_t *hp1; int32_t *hp2; int32_t *hp3; uint8_t *hp; uint8_t *n_ps; int32_t *ps_min;
    int32_t *end_pos; int32_t *read_len; uint8_t *has_site;
    int64_t pair_capacity, n_pairs;
    int32_t *pair_site; int32_t *pair_read; uint8_t *pair_base_hp;
    int64_t win_capacity, n_windows;
    int32_t *win_site; uint8_t *win_allele; int16_t *win_offset; uint8_t *win_base;
} lps_tumor_extract_result;

/* Stage timings of the last lps_phase_chromosome / lps_haplotag call, measured with hipEvents on the
 * library's stream.  ms_kernel[i] pairs with lps_stage_name(i). */
#define LPS_MAX_STAGES 24
typedef struct lps_timings {
    int32_t n_stages;
    float ms_kernel[LPS_MAX_STAGES];
    float ms_total;            /* first launch -> last result byte in host memory */
    int64_t n_obs;             /* read x variant observations after filters */
    int64_t n_nodes;           /* graph nodes */
    int64_t n_pairs;           /* edge increments applied */
    int64_t n_reads_used;      /* alignments with >=1 observation */
    int64_t algorithmic_bytes[LPS_MAX_STAGES]; /* SURVEY.md §8d closed forms evaluated on this input */
    int64_t n_scan_segments;   /* vote scan: speculative segments of the last phase call ...          */
    int64_t n_scan_replayed;   /* ... and how many had to be replayed serially (exactness fallback)   */
} lps_timings;

int lps_abi_version(void);
/* test hook: the library's step-by-step restatement of libstdc++'s std::sort (by key only) applied on the HOST to (keys, payload) pairs -
 * the routine the GPU runs on merged reads that hold a position twice (src/phase/PhasingGraph.cpp:854; csrc/lps_stdsort.h) */
void lps_debug_std_sort(int32_t *keys, uint8_t *payload, int64_t n);
/* test hook: the same sort as the GPU runs it (a wavefront per row, csrc/lps_graph.hip wave_std_sort) on rows [row_start[r], row_start[r+1]) of
 * host arrays (keys, payload), in place.  Returns 0, or -1 when the device call fails. */
int lps_debug_std_sort_gpu(int device, int32_t *keys, uint8_t *payload, const int64_t *row_start, int64_t n_rows);
/* test hook: size of the observation arenas (slots) of the next phase run; 0 = the library's own estimate.  A value that is too small makes
 * the extraction overflow, which the library answers by growing the arenas and running again (tests/test_scale_gpu.py). */
int lps_debug_set_obs_capacity(lps_ctx *ctx, int64_t slots);
/* sizeof() of the ABI structs as compiled into the library: 0 lps_params, 1 lps_variant_table, 2 lps_read_batch,
 * 3 lps_phase_result, 4 lps_haplotag_result, 5 lps_timings, 6 lps_somatic_tag_result, 7 lps_site_counters, 8 lps_tumor_extract_result,
 * 9 lps_extra_variants (binding self-check). */
int lps_struct_size(int which);
int lps_device_count(void);
/* PCI bus id ("0000:c1:00.0") of a device: ranks that share a GPU must not form an RCCL communicator (lps_comm_create) */
int lps_device_bus_id(int device, char *buf, int len);

lps_ctx *lps_create(int device, const lps_params *params);
void lps_destroy(lps_ctx *ctx);
const char *lps_last_error(lps_ctx *ctx);

/* Start a new chromosome: drops reads/variants/reference held by the ctx (device buffers are reused). */
int lps_begin_chromosome(lps_ctx *ctx);
/* getVariants_markindel + getLastSNP (src/phase/ParsingBam.cpp:378-417,426-441). */
int lps_set_variants(lps_ctx *ctx, const lps_variant_table *table);
/* lps_set_variants for a table that is already on the ctx's GPU: pos / ref0 / alt0 (and ref_len / alt_len, NULL = every row is a SNP) are
 * DEVICE pointers, the haplotag / somatic columns must be NULL.  The rows are copied device-to-device (the caller's buffer may go away when the
 * call returns); order and range of the positions are checked by a kernel.  With lps_comm_bcast_to_device the table of a multi-GPU run travels
 * ncclBroadcast -> context with no host hop (reference analogue: one parsed SnpParser shared by the chromosome loop, PhasingProcess.cpp:113-173). */
int lps_set_variants_device(lps_ctx *ctx, const lps_variant_table *table);
/* SV / MOD rows of the chromosome (phase only).  Call after lps_set_variants; NULL or an empty table = none.  The tables are copied. */
int lps_set_extra_variants(lps_ctx *ctx, const lps_extra_variants *extra);
/* Results of the SV / MOD rows of the last lps_phase_chromosome (sv->n = n_sv, mod->n = n_mod; either may be NULL). */
int lps_get_extra_result(lps_ctx *ctx, lps_phase_result *sv, lps_phase_result *mod);
/* Reference bases of the chromosome; the library applies FastaParser's truncation to [0,lastVariant+5]
 * (src/phase/ParsingBam.cpp:47) itself.  May be shorter than the contig as long as it covers that prefix. */
int lps_set_reference(lps_ctx *ctx, const char *seq, int64_t len);
/* Append decoded alignments (H2D copy).  May be called repeatedly (batches / several BAM files).  Every push leaves the alignments in the layout the
 * kernels read - the CIGAR words go into lane-chunks of 8 as they are copied into the context (csrc/lps_reads.hip), bases and qualities stay in the BAM
 * record's own encodings and are gathered in place: lps_phase_chromosome / lps_haplotag_chromosome start from the pushed arrays, nothing is prepared
 * or cached between a push and a call (ABI 20 removed lps_prepare_reads). */
int lps_push_reads(lps_ctx *ctx, const lps_read_batch *batch);
/* Same as lps_push_reads for a batch that is already resident on the ctx's GPU: every pointer of `batch` is a DEVICE pointer
 * (e.g. the output of a GPU-side BAM decoder or generator).  The arrays are copied device-to-device (the CIGAR words into lane-chunks as they are
 * copied); the caller may free them when the call returns.  The operand checks of lps_push_reads (offsets, seq/qual lengths, coordinate order) run as a kernel. */
int lps_push_reads_device(lps_ctx *ctx, const lps_read_batch *batch);
/* Append alignments as RAW (inflated) BAM records - what `sam_itr_multi_next` fills into bam1_t in the loop of
 * direct_detect_alleles (src/phase/ParsingBam.cpp:1279) / processSingleChrom (src/haplotag/HaplotagParsingBam.cpp:453),
 * before any field is decoded.  `records` = n_bytes of the uncompressed BAM stream covering the records (anything
 * between records is ignored); rec_off[i] = byte offset, inside `records`, of record i's refID field (its 4-byte
 * block_size sits just before).  The record core, the CIGAR re-alignment and the seq/qual addressing are decoded on the GPU;
 * seq and qual are used in place.  name_id as in lps_read_batch.  Records must be coordinate-sorted and of one contig.
 * Cannot be mixed with lps_push_reads inside one chromosome.  A CIGAR of more than 65535 operations is taken from the record's CG:B,I field, as
 * htslib's bam_tag2cigar does behind sam_itr_multi_next. */
int lps_push_bam_records(lps_ctx *ctx, const uint8_t *records, int64_t n_bytes, const uint64_t *rec_off, int64_t n_records, const uint32_t *name_id);

/* --- whole BAM file on the GPU (replaces htslib's BGZF layer behind sam_itr_multi_next, src/phase/ParsingBam.cpp:1279).
 * lps_bgzf_load: `bgzf` = the complete .bam file bytes (e.g. an mmap).  Block headers are walked on the host, every block is
 * inflated on the GPU (lps_inflate.hip), the inflated stream stays resident in the ctx until the next lps_bgzf_load / lps_destroy
 * (lps_begin_chromosome does not drop it).  Fails on a corrupt block, an ISIZE mismatch or a CRC32 mismatch (checked on the GPU).
 * lps_bgzf_read: copy a piece of the inflated stream to the host (BAM header text and reference table). */
int lps_bgzf_load(lps_ctx *ctx, const uint8_t *bgzf, int64_t n_bytes, int64_t *inflated_bytes);
/* the same for bytes [offset, offset + n_bytes) of an open FILE: read with pread straight into the page-locked upload pieces, so that a multi-GB file
 * is never mapped (an 8 GB mapping costs 0.14 s to tear down when the process ends, and as many page-table entries to set up while it is copied) */
int lps_bgzf_load_fd(lps_ctx *ctx, int fd, int64_t offset, int64_t n_bytes, int64_t *inflated_bytes);
/* The header walk on its own, HOST ONLY (no GPU, no ctx - e.g. while the HIP runtime is still coming up): the table of the BGZF blocks in bytes
 * [offset, offset + n_bytes) of the file, in_off relative to `offset`.  lps_bgzf_load_fd_blocks then takes the table instead of walking again. */
typedef struct { uint64_t in_off, out_off; uint32_t in_len, out_len; } lps_bgzf_block;
int lps_bgzf_walk_fd(int fd, int64_t offset, int64_t n_bytes, lps_bgzf_block **blocks, int64_t *n_blocks, int64_t *inflated_bytes);
void lps_bgzf_blocks_free(lps_bgzf_block *blocks);
int lps_bgzf_load_fd_blocks(lps_ctx *ctx, int fd, int64_t offset, int64_t n_bytes, const lps_bgzf_block *blocks, int64_t n_blocks, int64_t *inflated_bytes);
int lps_bgzf_read(lps_ctx *ctx, int64_t offset, int64_t n, uint8_t *dst);
int lps_bgzf_timings(lps_ctx *ctx, double *h2d_ms, double *inflate_ms);
/* 1 when the last lps_bgzf_load* had to inflate a second time: the kernel launched beside the upload outran a slow source (its wavefronts wait a
 * bounded time for their bytes) and the members were inflated again once the upload was complete.  The load itself succeeded either way. */
int lps_bgzf_retried(lps_ctx *ctx);
/* GPU BGZF writer (replaces bgzf_write/deflate behind sam_write1, src/haplotag/HaplotagParsingBam.cpp:124-134): bytes [offset, offset+n_bytes)
 * of the resident stream are cut into 0xff00-byte blocks, each deflated with a per-block dynamic Huffman code (no LZ77) and wrapped as a BGZF
 * member with CRC32/ISIZE; lps_bgzf_deflate leaves the blocks on the device and returns their total size, lps_bgzf_deflate_fetch copies them
 * to the host.  No EOF block is appended. */
int lps_bgzf_deflate(lps_ctx *ctx, int64_t offset, int64_t n_bytes, int64_t *out_bytes);
/* the same writer for bytes that sit in HOST memory (e.g. records a host-side splice produced): uploaded through the ctx's pinned ring, cut into
 * BGZF blocks and deflated on the GPU; the result is fetched like that of lps_bgzf_deflate.  The resident stream of the ctx is not touched. */
int lps_bgzf_deflate_host(lps_ctx *ctx, const uint8_t *bytes, int64_t n_bytes, int64_t *out_bytes);
int lps_bgzf_deflate_fetch(lps_ctx *ctx, uint8_t *dst, int64_t cap, double *kernel_ms);
/* the same in pieces: bytes [offset, offset + n) of the deflated result to `dst`.  With `dst` from lps_host_alloc (pinned memory) the copy runs at
 * PCIe speed and a writer thread can put piece k on disk while piece k+1 arrives (a 1.9 GB result into pageable memory costs 0.3 s). */
int lps_bgzf_deflate_fetch_range(lps_ctx *ctx, int64_t offset, int64_t n, uint8_t *dst);
void *lps_host_alloc(size_t bytes);   /* page-locked host memory (hipHostMalloc); NULL on failure */
void lps_host_free(void *p);
/* haplotag output on the GPU (tag rules of src/haplotag/HaplotagProcess.cpp:337-361 + the BGZF writer above): the records of the ONE
 * lps_push_bam_resident of this chromosome are re-emitted in order - a record with status 0 loses its first HP, PS and PQ optional field and,
 * when hp != 0, gains HP:i PS:i PQ:i; every other record is copied untouched - behind `prefix` (e.g. the BAM header for the first contig),
 * cut into BGZF blocks and deflated.  status/hp/ps/pq: the arrays lps_haplotag_chromosome filled (cur count entries).  Result stays on the
 * device (lps_bgzf_deflate_fetch copies it out); no EOF block is appended. */
int lps_haplotag_write_bgzf(lps_ctx *ctx, const uint8_t *status, const uint8_t *hp, const int32_t *ps, const int32_t *pq, const uint8_t *prefix, int64_t prefix_bytes,
                            int64_t *out_bytes);
/* the same writer with somatic_haplotag's tags (addAuxiliaryTags, src/somatic_haplotag/SomaticHaplotagProcess.cpp:529-536): a record with
 * status 0 loses its first HP, PS and PQ field and, when hp != 0, gains HP:Z with the haplotype's name (1 2 3 4 1-1 1-2 2-1 2-2 for hp 1..8),
 * PS:i unless ps == -1, and PQ:i.  status/hp/ps/pq: what the caller decided from lps_somatic_tag_chromosome's counts. */
int lps_somatic_write_bgzf(lps_ctx *ctx, const uint8_t *status, const uint8_t *hp, const int32_t *ps, const int32_t *pq, const uint8_t *prefix, int64_t prefix_bytes,
                           int64_t *out_bytes);
/* Find every BAM record of the resident stream on the GPU, starting at first_record_offset (= the byte after the BAM header's reference
 * table; the caller parses the header with lps_bgzf_read).  Every byte position is tested against the necessary conditions of a record
 * start and the candidate list is verified to be exactly the record chain (serial fallback otherwise), so the result is exact.
 * Records are numbered in file order.  lps_bam_record_tids: refID of every record (n_records entries) - a coordinate-sorted BAM holds each
 * contig as one contiguous range.  lps_bam_names: read names of records [first, first+count) packed back to back INCLUDING their NUL,
 * name_off[count+1] relative offsets; call with names == NULL to get the byte count.
 * lps_push_bam_resident: like lps_push_bam_records for records [first, first+count) of the resident stream - nothing is uploaded. */
int lps_bam_scan(lps_ctx *ctx, int64_t first_record_offset, int32_t n_ref, int64_t *n_records);
/* same for a piece of a BAM: the resident stream holds some consecutive BGZF blocks (e.g. the virtual-offset range a .bai index gives for
 * one contig) and the record chain occupies [first_record_offset, end_offset) of their inflated bytes. */
int lps_bam_scan_range(lps_ctx *ctx, int64_t first_record_offset, int64_t end_offset, int32_t n_ref, int64_t *n_records);
int lps_bam_record_tids(lps_ctx *ctx, int32_t *tid);
/* offsets (inside the inflated stream) of the refID field of records [first, first+count) - for hosts that also need the bytes (lps_bgzf_read) */
int lps_bam_record_offsets(lps_ctx *ctx, int64_t first, int64_t count, uint64_t *rec_off);
int lps_bam_names(lps_ctx *ctx, int64_t first, int64_t count, uint32_t *name_off, char *names, int64_t names_cap, int64_t *names_bytes);
int lps_push_bam_resident(lps_ctx *ctx, int64_t first, int64_t count, const uint32_t *name_id);

/* How much of a phase run is timed with events on the stream (lps_get_timings): 2 = every stage, 1 = only the extraction kernel (default;
 * ms_kernel of the other stages reads 0), 0 = nothing but ms_total.  Each recorded event idles the GPU for ~3.5 us, 14 of them ~4 % of a
 * chr20 step: the timed region of bench.py uses 1, its per-stage table comes from a separate pass at 2, the CLI uses 0. */
int lps_set_stage_timing(lps_ctx *ctx, int level);

/* phase: everything between direct_detect_alleles and exportResult for the reads pushed so far.
 * Recomputes from the resident raw reads on every call (nothing is cached between calls). */
int lps_phase_chromosome(lps_ctx *ctx, lps_phase_result *out);
/* haplotag: per-read scoring of the reads pushed so far against the phased table. */
int lps_haplotag_chromosome(lps_ctx *ctx, lps_haplotag_result *out);
/* host milliseconds the last lps_phase_chromosome / lps_haplotag_chromosome call spent growing device buffers (hipMalloc / hipFree): the FIRST call
 * on a chromosome sizes the stage buffers, later calls find them in place - and on some hosts an allocation of a few GB takes hundreds of ms. */
double lps_alloc_ms(lps_ctx *ctx);
/* k consecutive calls of the two entries above behind one call (every call does the full work; ms_each, k doubles or NULL, receives each call's wall
 * time): for callers whose own loop would put an interpreter between the calls (bench.py's timed region) */
int lps_phase_chromosome_steps(lps_ctx *ctx, lps_phase_result *out, int k, double *ms_each);
int lps_haplotag_chromosome_steps(lps_ctx *ctx, lps_haplotag_result *out, int k, double *ms_each);
/* judgeSVHap (src/haplotag/HaplotagStrategy.cpp:220-226), `haplotag --sv-file --mod-file`: the votes a read brings along from the phased SV / MOD
 * files - one per record that lists the read's name under RNAMES= / MR= (src/haplotag/HaplotagVcfParser.cpp:403-468), for the haplotype the
 * record's GT puts ALT on - are added to hpCount[H1] / hpCount[H2] of every scored alignment before judgeReadHap.  h1 / h2: one entry per pushed
 * alignment, in push order (NULL, NULL = none).  Call after the reads are pushed; the arrays are copied and dropped by lps_begin_chromosome. */
int lps_set_read_votes(lps_ctx *ctx, const int32_t *h1, const int32_t *h2, int64_t n_reads);
/* somatic_haplotag tagging pass over the (tumor) reads pushed so far against the merged normal+tumor table. */
int lps_somatic_tag_chromosome(lps_ctx *ctx, lps_somatic_tag_result *out);
/* somatic_haplotag pass 1: the NORMAL sample's reads (pushed so far) counted at the tumor-VCF positions. */
int lps_somatic_extract_normal(lps_ctx *ctx, lps_site_counters *out);
/* somatic_haplotag pass 2: the TUMOR sample's reads (pushed so far) at the merged table. */
int lps_somatic_extract_tumor(lps_ctx *ctx, lps_tumor_extract_result *out);

int lps_get_timings(lps_ctx *ctx, lps_timings *t);
const char *lps_stage_name(int stage);
/* The hipStream_t (as void*) all kernels of this ctx are launched on. */
void *lps_stream(lps_ctx *ctx);

/* ---- multi-GPU (one node): the ONE collective of the path.  Rank 0 parses the VCF; its packed variant table (and, if wanted, reference
 * slices) reaches the other GPUs by ncclBroadcast - RCCL over xGMI - and from then on every GPU works on its own contigs with no exchange.
 * Replaces nothing in the reference (its workers share one parsed SnpParser in one address space, src/phase/PhasingProcess.cpp:113-173).
 *   multi-process (one rank per GPU): rank 0 calls lps_comm_unique_id, hands the 128 bytes to the others over any control channel,
 *                                     every rank calls lps_comm_create(device, n_ranks, rank, id);
 *   single process, N GPUs (CLI --gpus N): lps_comm_create_all; each worker thread then uses its own lps_comm.
 * lps_comm_bcast: `host_buf` is the source on `root` and the destination elsewhere (staged through the communicator's device buffer);
 * lps_comm_bcast_device: the same on a DEVICE buffer, in place.  Both are collective: every rank of the communicator must call them with the
 * same n_bytes and root.  ms (optional): duration of the broadcast on the communicator's stream.  Errors: <0, lps_comm_last_error(). */
typedef struct lps_comm lps_comm;
int lps_comm_unique_id(uint8_t id[128]);
lps_comm *lps_comm_create(int device, int n_ranks, int rank, const uint8_t id[128]);
int lps_comm_create_all(int n_devices, const int *devices, lps_comm **comms);
int lps_comm_size(lps_comm *comm);
int lps_comm_rank(lps_comm *comm);
void lps_comm_destroy(lps_comm *comm);
int lps_comm_bcast(lps_comm *comm, void *host_buf, int64_t n_bytes, int root, double *ms);
int lps_comm_bcast_device(lps_comm *comm, void *dev_buf, int64_t n_bytes, int root, double *ms);
/* root's `host_src` (ignored elsewhere) -> a device buffer owned by the communicator on EVERY rank; *dev_out points at it until the next
 * broadcast on this communicator or lps_comm_destroy.  ms = -1 when the timing events failed. */
int lps_comm_bcast_to_device(lps_comm *comm, const void *host_src, int64_t n_bytes, int root, void **dev_out, double *ms);
const char *lps_comm_last_error(void);

/* ---- stage dumps for parity tests (valid after lps_phase_chromosome; host buffers, caller-allocated) ---- */
/* Observations in canonical order (alignment index, then position): per kept alignment
 * obs_count[read] entries.  Returns total or <0.  Pass NULL arrays to query the total only. */
int64_t lps_dump_observations(lps_ctx *ctx, int32_t *obs_count /*n_reads*/, int32_t *var_index, int8_t *allele,
                              int16_t *quality, int64_t capacity);
/* Graph nodes (variant indices, ascending) and the dense edge matrix [n_nodes][connect_adjacent][4]
 * (cell order rr, ra, ar, aa = (allele_i<<1)|allele_j).  Returns n_nodes. */
int64_t lps_dump_graph(lps_ctx *ctx, int32_t *node_var_index, float *edge, int64_t node_capacity);
/* Vote scan output per node: hp (0 none,1,2) and block start node (-1 none). */
int64_t lps_dump_votes(lps_ctx *ctx, int8_t *hp, int32_t *block_node, int64_t node_capacity);
/* Clip events counted by getClip (src/phase/ParsingBam.cpp:1636-1645): (ref_pos, 0 FRONT / 1 BACK). */
int64_t lps_dump_clips(lps_ctx *ctx, int32_t *pos, uint8_t *front_back, int64_t capacity);
/* CNV intervals of Clip::getCNVInterval (each interval appears twice, as in the reference; their number is unbounded) and, optionally,
 * the per-alignment "removed by the overlap filter" flags (n_reads bytes).  Returns the number of entries (pass NULL / 0 to query it). */
int64_t lps_dump_cnv(lps_ctx *ctx, int32_t *start, int32_t *end, int64_t capacity, uint8_t *aln_deleted);

#ifdef __cplusplus
}
#endif
#endif
