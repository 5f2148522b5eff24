// lps_common.h — shared device/host helpers of liblps_hip.so (gfx950 only, wave64).
#pragma once
#include <chrono>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/lps_abi.h"

#define LPS_WAVE 64

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            char _b[512];                                                                               \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            throw std::string(_b);                                                                      \
        }                                                                                               \
    } while (0)

// ---------------------------------------------------------------- device memory
// Growable device buffer.  grow() keeps contents only when keep=true (append buffers).
// host time the calling thread spent growing device buffers (hipMalloc / hipFree and the synchronisation before a free): an entry point of the
// library zeroes it when it starts and reports it with its timings - on some hosts an allocation of a few GB takes hundreds of milliseconds
inline thread_local double g_lps_alloc_ms = 0.0;
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    bool carved = false;       // p points into another allocation (see ZeroPool in lps_abi.hip): not owned
    ~DevBuf() { if (p && !carved) (void)hipFree(p); }
    void carve(void *q, size_t n) { if (p && !carved) (void)hipFree(p); p = (T *)q; cap = n; carved = true; }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    void reserve(size_t n, hipStream_t s = nullptr, bool keep = false, size_t used = 0) {
        if (n <= cap) return;
        if (carved) { p = nullptr; cap = 0; carved = false; }
        size_t nc = cap ? cap : 256;
        while (nc < n) nc = nc + nc / 2 + 256;
        T *q = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        HIP_TRY(hipMalloc((void **)&q, nc * sizeof(T) + 64));      // 64 B of slack: kernels read whole 16-B vectors at the tail (CIGAR words)
        if (keep && p && used) HIP_TRY(hipMemcpyAsync(q, p, used * sizeof(T), hipMemcpyDeviceToDevice, s));
        if (p) { HIP_TRY(hipStreamSynchronize(s)); HIP_TRY(hipFree(p)); }
        g_lps_alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        p = q; cap = nc;
    }
};

// ---------------------------------------------------------------- error / status words written by kernels
enum {
    LPS_ERR_BAD_CIGAR = 1,      // unsupported CIGAR op (reference: exit(1), ParsingBam.cpp:1625-1628)
    LPS_ERR_OBS_OVERFLOW = 2,   // observation buffer too small -> host grows and reruns
    LPS_ERR_KEY_RANGE = 4,      // more than 2^22 observations of one variant (the rank inside its list is a 22-bit field)
    LPS_ERR_CLIP_OVERFLOW = 16,
};

// device-side counters block (one per ctx), zeroed at the start of every run
struct LpsCounters {
    unsigned long long obs_total;   // reserved observation slots summed over the arenas (k_name_link's extra workgroup)
    unsigned int n_clips;
    unsigned int err;
    unsigned int n_kept;            // alignments with >=1 observation
    unsigned int n_multi;           // read names with >= 2 such alignments (k_groups)
    unsigned int mm_total;          // ... and how many alignments they hold together
    unsigned int max_row;           // longest row (observations of one alignment)
    unsigned int max_group;         // most alignments under one read name
    unsigned int n_cnv;
    unsigned int ub_hazard;
    unsigned int n_nodes;
    unsigned int n_abandoned;       // observations counted at the extraction whose job then went to the general walker (their list places are holes)
    unsigned int clip_mult;         // upper bound of the largest number of clip events with one key (position, front / back): count-min over two hashed tables (k_name_link)
    // ---- accumulated by the stages after the overlap filter (cleared when they run again with the CNV filter)
    unsigned long long n_pairs;
    unsigned long long n_obs_final; // observations of kept alignments after all filters
    unsigned long long tail_total;  // slots reserved in the tail arena (merged rows of multi-alignment reads)
    unsigned int n_merged;          // merged rows built from >=2 alignments
    unsigned int pad;
    // ----
    unsigned long long arena_max;   // largest per-arena reservation (capacity planning on overflow)
};

#define LPS_ARENAS 64              // observation arenas, each with its own reservation counter on its own 64-B line

// ---------------------------------------------------------------- wave-level primitives (wave64)
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <class T>
__device__ __forceinline__ T wave_incl_scan(T v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d);
        if (lane_id() >= d) v += o;
    }
    return v;
}

// Inclusive wave64 scan of a 32-bit int with DPP row shifts / row broadcasts (gfx9 family: row_shr:n = 0x110+n,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143).  6 VALU instructions, no LDS crossbar traffic (vs 6 ds_bpermute).
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// the same for the running maximum (lanes a shift does not reach keep INT_MIN)
__device__ __forceinline__ int wave_incl_max_dpp(int v) {
    const int none = (int)0x80000000;
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x143, 0xc, 0xf, false));
    return v;
}

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
    return v;
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}

__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

// Wave-cooperative lower_bound over a sorted global int array: first index in [lo,hi) with a[idx] >= key.
// 64-ary search: ~3 rounds for 3e5 elements instead of 18 dependent loads.  All lanes must call with equal args.
__device__ __forceinline__ int wave_lower_bound(const int32_t *__restrict__ a, int lo, int hi, int key) {
    const int l = lane_id();
    while (hi - lo > 64) {
        const int step = (hi - lo + 63) >> 6;            // probes at lo + (l+1)*step - 1
        long long pi = (long long)lo + (long long)(l + 1) * step - 1;
        bool lt = (pi < hi) ? (a[pi] < key) : false;
        const int c = __popcll(__ballot(lt));            // number of probe points < key (prefix property)
        const int nlo = lo + c * step;                   // everything before is < key
        const int nhi = min(hi, lo + (c + 1) * step);    // probe c is >= key (or out of range)
        lo = nlo; hi = nhi;
        if (lo >= hi) return lo;
    }
    bool lt = (lo + l < hi) ? (a[lo + l] < key) : false;
    return lo + __popcll(__ballot(lt));
}

// homopolymerLength (src/shared/Util.cpp:21-54) on the truncated reference prefix
__device__ __forceinline__ int homopolymer_length(const char *__restrict__ ref, long long ref_len, long long p) {
    int len = 1;
    if (p + 1 >= ref_len) return len;
    const char e = ref[p];
    long long q = p - 1;
    while (q >= 0 && ref[q] == e) { --q; ++len; if (len >= 10 || q < 0) break; }
    q = p + 1;
    if (q < ref_len) {
        while (ref[q] == e) { ++q; ++len; if (q >= ref_len) break; if (len >= 10) break; }
    }
    return len;
}

__device__ __forceinline__ char nt16_char(int code) {
    // htslib seq_nt16_str "=ACMGRSVTWYHKDBN" packed little-endian into two 64-bit words
    const unsigned long long lo = 0x565352474d43413dull, hi = 0x4e42444b48595754ull;
    code &= 15;
    return (char)(((code < 8 ? lo : hi) >> ((code & 7) * 8)) & 0xff);
}

// observation quality/allele packing: bits 0..8 = quality+8 (sentinels -4/-5 .. 255), bit 9 = allele
__host__ __device__ __forceinline__ uint16_t pack_aq(int allele, int quality) { return (uint16_t)((allele << 9) | ((quality + 8) & 0x1ff)); }
__host__ __device__ __forceinline__ int aq_allele(uint16_t aq) { return (aq >> 9) & 1; }
__host__ __device__ __forceinline__ int aq_quality(uint16_t aq) { return (int)(aq & 0x1ff) - 8; }

// Workgroups are dealt round-robin over the 8 XCDs (one L2 each): workgroup b of a grid of 8k takes unit (b % 8) * k + b / 8, so that an XCD walks ONE
// contiguous eighth of the units and what neighbouring units share or write side by side meets in one L2.
__device__ __forceinline__ int xcd_unit(int b, int n_blocks8) { return (b & 7) * (n_blocks8 >> 3) + (b >> 3); }
__host__ __device__ inline int round_up8(int x) { return (x + 7) / 8 * 8; }

#define LPS_SEG 512           // CIGAR ops staged in LDS per wave and segment (4 KB/wave)
#define LPS_BUCKET_SHIFT 10    // coarse position index: bucket b = first variant with pos >= b << shift
