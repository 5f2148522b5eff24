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
    unsigned long long capacity;
};

struct ClipView {          // raw clip events; filtered by row_fail afterwards
    int32_t *pos; int32_t *read; int32_t *opidx_fb;   // opidx<<1 | (opidx!=0)
    unsigned int capacity;
};

void launch_variant_prep(const VarView &V, int is_ont, hipStream_t s);
void launch_extract_phase(const VarView &V, const ReadView &R, const ObsView &O, const ClipView &C,
                          int mapping_quality, LpsCounters *cnt, hipStream_t s);
