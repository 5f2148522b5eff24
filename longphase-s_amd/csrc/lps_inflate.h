// GPU BGZF inflate (lps_inflate.hip) and BAM record discovery (lps_bam.hip): shared types + launchers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

enum { LPS_INF_ERR_DATA = 1, LPS_INF_ERR_OVERRUN = 2, LPS_INF_ERR_SIZE = 4, LPS_INF_ERR_CRC = 8, LPS_INF_ERR_TIMEOUT = 16 };

struct InflateBlock {       // one BGZF block: raw deflate bytes [in_off, in_off+in_len) -> [out_off, out_off+out_len)
    uint64_t in_off, out_off;
    uint32_t in_len, out_len;
};

size_t bgzf_inflate_scratch_bytes(int n_blk);      // per-lane scratch columns for the code lengths of a header
// `uploaded` (may be NULL): a word in page-locked host memory that the uploading host thread raises while the compressed bytes arrive - the number of bytes of `in` that are in place
// (monotonic, chunk ends on 128-byte boundaries, total_in when everything is there).  A wavefront starts on its 32 members when their bytes are there;
// the launch may therefore be made BEFORE the upload has finished.  A wave that has waited for timeout_ms sets LPS_INF_ERR_TIMEOUT and
// leaves (every wave reaches an exit): the caller then runs the launch again once the upload is complete (lps_bgzf_load does).
void launch_bgzf_inflate(const uint8_t *in, const InflateBlock *blk, int n_blk, uint8_t *out, unsigned *err, uint8_t *scratch, hipStream_t s,
                         const unsigned long long *uploaded = nullptr, unsigned long long total_in = 0, double timeout_ms = 5000.0);
void launch_bgzf_crc(const uint8_t *in, const InflateBlock *blk, int n_blk, const uint8_t *out, unsigned *err, hipStream_t s);
