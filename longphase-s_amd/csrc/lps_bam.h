// Device-side BAM record decode (lps_bam.hip): views + launchers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lps_common.h"

enum { LPS_BAM_ERR_BOUNDS = 1, LPS_BAM_ERR_UNSORTED = 2 };

struct BamView {
    const uint8_t *blob;        // device copy of the inflated BAM bytes pushed so far (+16 bytes of slack)
    uint64_t push_base;         // where this push's bytes start inside blob
    uint64_t push_bytes;        // bytes of this push
    const uint64_t *rec_off;    // per record of this push: offset of its refID field (block_size sits 4 bytes before)
};

void launch_bam_core(const BamView &B, int n, int at, int32_t *ref_start, int32_t *l_qseq, uint16_t *flag, uint8_t *mapq, uint64_t *seq_off,
                     uint64_t *qual_off, unsigned long long *cig_cnt /* [n+1] lane-chunks */, int32_t *cig_n /* [at + i] words */, uint64_t *cig_src, unsigned *err, hipStream_t s);
// words of record i -> chunks [chunk_off[i], chunk_off[i+1]) of cigp (padded with 6u), cp_off[i] = (uint32_t)chunk_off[i] for i = 0..n
void launch_bam_cigar(const BamView &B, int n, const uint64_t *chunk_off, const int32_t *cig_n, const uint64_t *cig_src, uint32_t *cigp, uint32_t *cp_off, hipStream_t s);
void bam_cigar_offsets(DevBuf<char> &temp, size_t &temp_bytes, const unsigned long long *cig_cnt, uint64_t *cigar_off, int n, uint64_t init, hipStream_t s);
// record discovery in a resident (GPU-inflated) BAM stream; returns 0, or <0: -2 stream too large, -3 no record found, -4 broken record chain
int bam_scan_records(const uint8_t *d, uint64_t first_rec, uint64_t total, int32_t n_ref, DevBuf<uint64_t> &cand, DevBuf<uint32_t> &wg_cnt, DevBuf<uint32_t> &wg_off,
                     DevBuf<char> &temp, size_t &temp_bytes, unsigned *flag, uint32_t *n_out_d, uint64_t *n_records, hipStream_t s);
void launch_bam_tid_lname(const uint8_t *d, const uint64_t *cand, uint32_t n, int32_t *tid, uint32_t *l_name, hipStream_t s);
void launch_bam_names(const uint8_t *d, const uint64_t *cand, uint32_t n, const uint32_t *name_off, uint8_t *names, hipStream_t s);
// haplotag writer: prefix + re-tagged records -> `stream` (the caller puts the prefix bytes at stream[0, prefix_bytes) BEFORE the call); -1 = malformed optional field
int64_t bam_tag_stream(const uint8_t *d, const uint64_t *rec, uint32_t n, const uint8_t *status, const uint8_t *hp, const int32_t *ps, const int32_t *pq, int somatic_tags, uint64_t prefix_bytes,
                       DevBuf<unsigned long long> &new_len, DevBuf<unsigned long long> &out_off, DevBuf<uint2> &spans, DevBuf<uint8_t> &stream, DevBuf<char> &temp, size_t &temp_bytes,
                       unsigned *err, hipStream_t s);
