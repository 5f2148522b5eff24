// GPU BGZF inflate (lps_inflate.hip) and BAM record discovery (lps_bam.hip): shared types + launchers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

enum { LPS_INF_ERR_DATA = 1, LPS_INF_ERR_OVERRUN = 2, LPS_INF_ERR_SIZE = 4, LPS_INF_ERR_CRC = 8 };

struct InflateBlock {       // one BGZF block: raw deflate bytes [in_off, in_off+in_len) -> [out_off, out_off+out_len)
    uint64_t in_off, out_off;
    uint32_t in_len, out_len;
};

size_t bgzf_inflate_scratch_bytes(int n_blk);      // per-lane scratch columns for the code lengths of a header
void launch_bgzf_inflate(const uint8_t *in, const InflateBlock *blk, int n_blk, uint8_t *out, unsigned *err, uint8_t *scratch, hipStream_t s);
void launch_bgzf_crc(const uint8_t *in, const InflateBlock *blk, int n_blk, const uint8_t *out, unsigned *err, hipStream_t s);
