// lps_deflate.hip — BGZF writer on the GPU: a resident byte stream is cut into 0xff00-byte blocks (htslib's
// BGZF_BLOCK_SIZE), each block is DEFLATEd as 16 sub-blocks with their own dynamic Huffman code over the literals (no LZ77: packed bases and
// qualities hold few matches; what pays on BAM records is following the change of statistics from field to field), wrapped as a gzip member
// with the BC extra field, CRC32 and ISIZE.
//
// What it replaces: bgzf_write/deflate behind sam_write1 (src/haplotag/HaplotagParsingBam.cpp:124-134), the dominant cost of the
// reference's `haplotag`.  One WAVE per block:
//   1. histogram of the 65280 bytes (LDS atomics), per-lane CRC32 of its 1020-byte slice (slices are combined with x^(8n) mod P)
//   2. Huffman code lengths (<= 15 bits): rank sort of the used symbols across the wave, two-queue tree build and zlib-style overflow
//      repair on lane 0, canonical codes
//   3. header: HLIT/HDIST/HCLEN with a flat 4-bit code-length code (no run-length symbols), then every lane encodes its slice at the bit
//      offset given by a wave scan of the slice sizes; slice boundaries meet inside a dword, so edge dwords are OR-ed atomically
//   4. a block that would not shrink is emitted as a stored block.
// Blocks land in fixed 64 KiB slots; k_bgzf_pack then copies them back to back (prefix sum of the sizes).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "lps_common.h"
#include "lps_deflate.h"

#define CRC_POLY 0xedb88320u

__device__ __forceinline__ uint32_t multmodp(uint32_t a, uint32_t b) {     // a(x) * b(x) mod P in the reflected representation
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
__device__ __forceinline__ uint32_t x8nmodp(uint32_t n_bytes, const uint32_t *x2n) {   // x^(8 n) mod P; x2n[k] = x^(2^k) mod P
    uint32_t p = 1u << 31; uint32_t n = n_bytes; int k = 3;
    while (n) { if (n & 1) p = multmodp(x2n[k & 31], p); n >>= 1; ++k; }
    return p;
}

struct DefShared {
    uint32_t hist[288];
    uint32_t crc_tab[256];
    uint32_t x2n[32];
    uint16_t code[288];
    uint8_t len[288];
    uint16_t order[288];          // symbols sorted by (frequency, symbol), used ones first
    int32_t parent[2 * 288];      // tree scratch (lane 0)
    uint32_t weight[2 * 288];
    uint8_t depth[2 * 288];
    uint32_t data_bits, header_bits, header_kind;
};

// Each BGZF block is written as LPS_SUB deflate blocks of its own Huffman code: the bytes of a BAM record change character every few KiB (names,
// CIGAR words, 4-bit bases, qualities), and a code per 4 KiB follows that - on ONT-like records this beats zlib level 6 in size.
#define LPS_SUB_BYTES 4096u

__global__ void __launch_bounds__(64) k_bgzf_deflate(const uint8_t *src, uint64_t n_bytes, uint8_t *slots, uint32_t *slot_bytes) {
    __shared__ DefShared S;
    const int lane = threadIdx.x; const uint64_t blk = blockIdx.x;
    const uint64_t b0 = blk * LPS_BGZF_BLOCK; const uint32_t len = (uint32_t)((n_bytes - b0 < LPS_BGZF_BLOCK) ? n_bytes - b0 : LPS_BGZF_BLOCK);
    const uint8_t *in = src + b0; uint8_t *out = slots + blk * LPS_BGZF_SLOT;
    for (int k = lane; k < 256; k += 64) { uint32_t c = (uint32_t)k; for (int j = 0; j < 8; ++j) c = (c & 1) ? (c >> 1) ^ CRC_POLY : c >> 1; S.crc_tab[k] = c; }
    if (lane < 32) { uint32_t p = 1u << 30; for (int k = 0; k < lane; ++k) p = multmodp(p, p); S.x2n[lane] = p; }   // x^(2^lane) mod P
    uint32_t *w = reinterpret_cast<uint32_t *>(out + 16);                 // the deflate body starts at out + 18: 16 bits into this dword view
    for (uint32_t k = lane; k < (LPS_BGZF_SLOT - 16) / 4; k += 64) w[k] = 0;
    __syncthreads();
    // ---- CRC32 of the block: per-lane slices, combined as state_i * x^(8 * bytes after slice i)
    uint32_t crc;
    {
        const uint32_t per = (len + 63) / 64, s0 = min(len, per * lane), s1 = min(len, s0 + per);
        uint32_t c = lane == 0 ? 0xffffffffu : 0u;
        for (uint32_t k = s0; k < s1; ++k) c = S.crc_tab[(c ^ in[k]) & 255u] ^ (c >> 8);
        c = multmodp(x8nmodp(len - s1, S.x2n), c);
        for (int o = 32; o; o >>= 1) c ^= __shfl_xor(c, o);
        crc = c ^ 0xffffffffu;
    }
    uint32_t bitpos = 16;                                                 // wave-uniform write position in `w`
    auto emit = [&](uint64_t &acc, int &nacc, uint32_t &wpos, uint32_t v, int nb) {   // every dword is OR-ed: neighbours share edge dwords
        acc |= (uint64_t)v << nacc; nacc += nb;
        while (nacc >= 32) { atomicOr(&w[wpos], (uint32_t)acc); ++wpos; acc >>= 32; nacc -= 32; }
    };
    const uint32_t n_sub = len ? (len + LPS_SUB_BYTES - 1) / LPS_SUB_BYTES : 0; bool too_big = false;
    for (uint32_t sb = 0; sb < n_sub; ++sb) {
        const uint32_t q0 = sb * LPS_SUB_BYTES, qn = min(LPS_SUB_BYTES, len - q0); const uint8_t *qi = in + q0;
        for (int k = lane; k < 288; k += 64) { S.hist[k] = 0; S.len[k] = 0; }
        __syncthreads();
        const uint32_t per = (qn + 63) / 64, s0 = min(qn, per * lane), s1 = min(qn, s0 + per);
        for (uint32_t k = s0; k < s1; ++k) atomicAdd(&S.hist[qi[k]], 1u);
        if (lane == 0) S.hist[256] = 1;                                   // end-of-block symbol
        __syncthreads();
        // ---- code lengths.  rank sort (ascending frequency; unused symbols last)
        uint32_t used_here = 0;
        for (int s = lane; s < 257; s += 64) {
            const uint32_t f = S.hist[s]; if (f) ++used_here;
            const uint64_t key = f ? ((uint64_t)f << 16 | (uint32_t)s) : (0xffffffffull << 16 | (uint32_t)s);
            int rank = 0;
            for (int t = 0; t < 257; ++t) { const uint32_t g = S.hist[t]; const uint64_t kt = g ? ((uint64_t)g << 16 | (uint32_t)t) : (0xffffffffull << 16 | (uint32_t)t); rank += kt < key; }
            S.order[rank] = (uint16_t)s;
        }
        for (int o = 32; o; o >>= 1) used_here += __shfl_xor(used_here, o);
        __syncthreads();
        if (lane == 0) {
            const int n = (int)used_here;                                 // >= 2: at least one literal and the end-of-block symbol
            // two-queue Huffman: leaves 0..n-1 (sorted), internal nodes n..2n-2 are created in non-decreasing weight order
            for (int i = 0; i < n; ++i) S.weight[i] = S.hist[S.order[i]];
            int leaf = 0, inode = n, next = n;
            // (both cursors advance by a 0/1 amount: `return leaf++` / `return inode++` on two paths became ONE increment through a selected pointer,
            // which kept both cursors in scratch memory - three scratch accesses with their waits per pick)
            auto pick = [&]() -> int {
                const bool from_leaves = leaf < n && (inode >= next || S.weight[leaf] <= S.weight[inode]);
                const int r = from_leaves ? leaf : inode;
                leaf += from_leaves ? 1 : 0; inode += from_leaves ? 0 : 1;
                return r;
            };
            for (int m = 0; m < n - 1; ++m) { const int a = pick(), b = pick(); S.weight[next] = S.weight[a] + S.weight[b]; S.parent[a] = next; S.parent[b] = next; ++next; }
            const int root = next - 1; S.depth[root] = 0;
            for (int i = root - 1; i >= 0; --i) { const int d = S.depth[S.parent[i]] + 1; S.depth[i] = (uint8_t)(d > 255 ? 255 : d); }   // parents have larger indices
            // length limit 15 (zlib gen_bitlen's repair): count lengths, move overflowing leaves up, reassign in frequency order
            int bl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int overflow = 0;
            for (int i = 0; i < n; ++i) { int d = S.depth[i]; if (d > 15) { d = 15; ++overflow; } ++bl[d]; }
            while (overflow > 0) { int bits = 14; while (bl[bits] == 0) --bits; --bl[bits]; bl[bits + 1] += 2; --bl[15]; overflow -= 2; }
            int i = 0;                                                    // least frequent symbols get the longest codes
            for (int bits = 15; bits >= 1; --bits) for (int c = bl[bits]; c > 0; --c) S.len[S.order[i++]] = (uint8_t)bits;
            // canonical codes (RFC 1951 3.2.2), stored bit-reversed so that they can be emitted LSB first
            int blc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; for (int s = 0; s < 257; ++s) ++blc[S.len[s]];
            blc[0] = 0; int nextc[16]; int code = 0; for (int b = 1; b <= 15; ++b) { code = (code + blc[b - 1]) << 1; nextc[b] = code; }
            uint32_t bits_total = 0;
            for (int s = 0; s < 257; ++s) { const int l = S.len[s]; if (l) { S.code[s] = (uint16_t)(__brev((uint32_t)nextc[l]++) >> (32 - l)); bits_total += (uint32_t)l * S.hist[s]; } }
            S.data_bits = bits_total;
            // two ways to send the 258 code lengths, both with a FIXED code-length code so that no third Huffman code has to be built:
            //   kind 0: symbols 0..15 at 4 bits each (dense alphabets);  kind 1: 0..15 at 5 bits, zero runs 17 (3..10) / 18 (11..138) at 2 bits (sparse ones)
            uint32_t hb = 0;
            for (int s = 0; s < 258;) {
                if (s < 257 && S.len[s]) { hb += 5; ++s; continue; }
                int r = 0; while (s + r < 258 && (s + r == 257 || S.len[s + r] == 0)) ++r;
                int left = r; while (left >= 11) { const int t = left > 138 ? 138 : left; hb += 2 + 7; left -= t; } if (left >= 3) { hb += 2 + 3; left = 0; } hb += 5 * left;
                s += r;
            }
            S.header_kind = hb < 258 * 4 ? 1u : 0u;
            S.header_bits = 3 + 5 + 5 + 4 + 19 * 3 + (S.header_kind ? hb : 258u * 4u);
        }
        __syncthreads();
        const uint32_t header_bits = S.header_bits;
        if ((bitpos + header_bits + S.data_bits + 7) / 8 > len + 5 + 16) { too_big = true; break; }   // (uniform) will not beat a stored block, and must not leave the slot
        if (lane == 0) {
            uint64_t acc = 0; int nacc = (int)(bitpos & 31); uint32_t wpos = bitpos >> 5;
            emit(acc, nacc, wpos, sb + 1 == n_sub ? 1u : 0u, 1); emit(acc, nacc, wpos, 2, 2);   // BFINAL, dynamic
            emit(acc, nacc, wpos, 0, 5); emit(acc, nacc, wpos, 0, 5); emit(acc, nacc, wpos, 19 - 4, 4);   // HLIT = 257 codes, HDIST = 1 code, HCLEN = 19 lengths
            // code-length code lengths in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
            if (S.header_kind == 0) { for (int k = 0; k < 19; ++k) emit(acc, nacc, wpos, k < 3 ? 0u : 4u, 3); }
            else { emit(acc, nacc, wpos, 0, 3); emit(acc, nacc, wpos, 2, 3); emit(acc, nacc, wpos, 2, 3); for (int k = 3; k < 19; ++k) emit(acc, nacc, wpos, 5, 3); }
            if (S.header_kind == 0) { for (int s = 0; s < 258; ++s) { const uint32_t l = s < 257 ? S.len[s] : 0; emit(acc, nacc, wpos, __brev(l) >> 28, 4); } }   // flat code: symbol l = code l, MSB first
            else {                                                         // canonical: 17 -> 00, 18 -> 01 (2 bits), symbol l -> 16 + l (5 bits); all sent MSB first
                for (int s = 0; s < 258;) {
                    if (s < 257 && S.len[s]) { emit(acc, nacc, wpos, __brev(16u + S.len[s]) >> 27, 5); ++s; continue; }
                    int r = 0; while (s + r < 258 && (s + r == 257 || S.len[s + r] == 0)) ++r;
                    int left = r;
                    while (left >= 11) { const int t = left > 138 ? 138 : left; emit(acc, nacc, wpos, 2, 2); emit(acc, nacc, wpos, (uint32_t)(t - 11), 7); left -= t; }   // 18: code 01 -> bits 1,0 LSB-first value 2
                    if (left >= 3) { emit(acc, nacc, wpos, 0, 2); emit(acc, nacc, wpos, (uint32_t)(left - 3), 3); left = 0; }
                    while (left-- > 0) emit(acc, nacc, wpos, __brev(16u) >> 27, 5);
                    s += r;
                }
            }
            if (nacc) atomicOr(&w[wpos], (uint32_t)acc);
        }
        // ---- the lanes' slices at the bit offsets of a wave scan
        uint32_t my_bits = 0;
        for (uint32_t k = s0; k < s1; ++k) my_bits += S.len[qi[k]];
        if (lane == 63) my_bits += S.len[256];
        uint32_t off = my_bits;                                           // inclusive scan
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(off, o); if (lane >= o) off += v; }
        const uint32_t bit0 = bitpos + header_bits + off - my_bits;
        {
            uint64_t acc = 0; int nacc = (int)(bit0 & 31); uint32_t wpos = bit0 >> 5; bool first = true;
            for (uint32_t k = s0; k < s1; ++k) {
                const uint8_t v = qi[k]; acc |= (uint64_t)S.code[v] << nacc; nacc += S.len[v];
                if (nacc >= 32) { if (first) { atomicOr(&w[wpos], (uint32_t)acc); first = false; } else w[wpos] = (uint32_t)acc; ++wpos; acc >>= 32; nacc -= 32; }
            }
            if (lane == 63) { acc |= (uint64_t)S.code[256] << nacc; nacc += S.len[256]; if (nacc >= 32) { if (first) { atomicOr(&w[wpos],
                    (uint32_t)acc); first = false; } else w[wpos] = (uint32_t)acc; ++wpos; acc >>= 32; nacc -= 32; } }
            if (nacc) atomicOr(&w[wpos], (uint32_t)acc);
        }
        bitpos += header_bits + S.data_bits;                              // uniform: S.* were written before the barrier above
        __syncthreads();
    }
    uint8_t *body = out + 18; uint32_t body_bytes = (bitpos - 16 + 7) / 8;
    if (len == 0) {                                                       // empty input: the canonical empty deflate stream 03 00
        if (lane == 0) { body[0] = 3; body[1] = 0; }
        body_bytes = 2;
    } else if (too_big || body_bytes >= len + 5) {                        // did not shrink: one stored block instead
        __syncthreads();
        for (uint32_t k = lane; k < (LPS_BGZF_SLOT - 16) / 4; k += 64) w[k] = 0;
        __syncthreads();
        if (lane == 0) { body[0] = 1; body[1] = (uint8_t)len; body[2] = (uint8_t)(len >> 8); body[3] = (uint8_t)~len; body[4] = (uint8_t)(~len >> 8); }
        for (uint32_t k = lane; k < len; k += 64) body[5 + k] = in[k];
        body_bytes = 5 + len;
    }
    __syncthreads();
    if (lane == 0) {                                                      // gzip member header (RFC 1952 + BC subfield) and trailer
        const uint32_t bsize = 18 + body_bytes + 8;
        const uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)((bsize - 1) & 255), (uint8_t)((bsize - 1) >> 8)};
        for (int k = 0; k < 18; ++k) out[k] = hdr[k];
        uint8_t *t = out + 18 + body_bytes;
        for (int k = 0; k < 4; ++k) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)(len >> (8 * k)); }
        slot_bytes[blk] = bsize;
    }
}

// copy the slots back to back: wave per block
__global__ void __launch_bounds__(256) k_bgzf_pack(const uint8_t *slots, const uint32_t *slot_bytes, const uint64_t *slot_off, uint32_t n_blk, uint8_t *dst) {
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= n_blk) return;
    const uint8_t *s = slots + (uint64_t)b * LPS_BGZF_SLOT; uint8_t *d = dst + slot_off[b]; const uint32_t n = slot_bytes[b];
    for (uint32_t k = lane; k < n; k += 64) d[k] = s[k];
}
__global__ void k_widen(const uint32_t *in, unsigned long long *out, uint32_t n) { const uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i <= n) out[i] = i < n ? in[i] : 0; }

// src[0, n_bytes) (device) -> BGZF blocks back to back in `packed` (device); returns the packed size
uint64_t bgzf_deflate_device(const uint8_t *src, uint64_t n_bytes, DevBuf<uint8_t> &slots, DevBuf<uint32_t> &slot_bytes, DevBuf<unsigned long long> &tmp64, DevBuf<uint64_t> &slot_off,
                             DevBuf<uint8_t> &packed, DevBuf<char> &temp, size_t &temp_bytes, hipStream_t s) {
    const uint64_t n_blk64 = (n_bytes + LPS_BGZF_BLOCK - 1) / LPS_BGZF_BLOCK;
    if (n_blk64 == 0) return 0;
    const uint32_t n_blk = (uint32_t)n_blk64;
    slots.reserve((size_t)n_blk * LPS_BGZF_SLOT + 64, s); slot_bytes.reserve(n_blk + 1, s); tmp64.reserve(n_blk + 1, s); slot_off.reserve(n_blk + 1, s);
    hipLaunchKernelGGL(k_bgzf_deflate, dim3(n_blk), dim3(64), 0, s, src, n_bytes, slots.p, slot_bytes.p);
    hipLaunchKernelGGL(k_widen, dim3((n_blk + 256) / 256), dim3(256), 0, s, (const uint32_t *)slot_bytes.p, tmp64.p, n_blk);
    size_t need = 0;
    unsigned long long *off = reinterpret_cast<unsigned long long *>(slot_off.p);
    HIP_TRY(rocprim::exclusive_scan(nullptr, need, tmp64.p, off, 0ull, (size_t)n_blk + 1, rocprim::plus<unsigned long long>(), s));
    if (need + 256 > temp_bytes) { temp.reserve(need + 256, s); temp_bytes = need + 256; }
    HIP_TRY(rocprim::exclusive_scan(temp.p, need, tmp64.p, off, 0ull, (size_t)n_blk + 1, rocprim::plus<unsigned long long>(), s));
    uint64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, slot_off.p + n_blk, sizeof total, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    packed.reserve((size_t)total + 64, s);
    hipLaunchKernelGGL(k_bgzf_pack, dim3((n_blk + 3) / 4), dim3(256), 0, s, (const uint8_t *)slots.p, (const uint32_t *)slot_bytes.p, (const uint64_t *)slot_off.p, n_blk, packed.p);
    return total;
}
