// GPU BGZF writer (lps_deflate.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lps_common.h"

#define LPS_BGZF_BLOCK 0xff00u     /* input bytes per BGZF block (htslib BGZF_BLOCK_SIZE) */
#define LPS_BGZF_SLOT 65536u       /* a finished block is at most 64 KiB */

uint64_t bgzf_deflate_device(const uint8_t *src, uint64_t n_bytes, DevBuf<uint8_t> &slots, DevBuf<uint32_t> &slot_bytes, DevBuf<unsigned long long> &tmp64, DevBuf<uint64_t> &slot_off,
                             DevBuf<uint8_t> &packed, DevBuf<char> &temp, size_t &temp_bytes, hipStream_t s);
