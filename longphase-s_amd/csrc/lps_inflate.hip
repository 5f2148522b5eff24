// lps_inflate.hip — BGZF (RFC 1951 DEFLATE in 64 KiB gzip members) decoded on the GPU, SURVEY.md §8f rank 1.
//
// What it replaces: htslib's bgzf.c block inflate behind sam_itr_multi_next (src/phase/ParsingBam.cpp:1279,
// src/haplotag/HaplotagParsingBam.cpp:453), the largest share of the reference's wall clock.
//
// One LANE per BGZF block (blocks are independent, ≤64 KiB each; a chr20-30x BAM has ~50 k of them): the bit stream is
// inherently serial inside a block, so the parallelism is across blocks.  Per lane the Huffman tables live in LDS,
// laid out [entry][lane] so that the 64 lanes of a wave always hit 64 different banks/words whatever entry each one needs:
//   fast  u16[256]  litlen codes of <=8 bits: (symbol << 4) | length, 0 = take the canonical path
//   lsym  u16[288]  litlen symbols sorted by (length, symbol); llim/lbase u16[8]: limit + list offset of lengths 9..15 (branch-free canonical decode)
//   dsym  u8[32], dlim/dbase u16[16]   same for distances (decoded only after a length symbol, so no fast table)
//   ring  64 output bytes, written to HBM as aligned 32-byte segments
// 80 KiB per 64-lane workgroup => 2 workgroups per CU.  The input is read as aligned dwords, two dwords ahead of use.
// Errors (corrupt stream, output overrun) set a flag; nothing is written outside [out_off, out_off + out_len).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lps_inflate.h"

#define L(a, i) a[(i) * 64 + lane]
// header scratch inside the lane's OWN column of the fast table (lanes sit in different states, so a scratch laid out any other way would
// trample a neighbour's live table): code length i = byte (i & 1) of cell i >> 1 (cells 0..159), cnt[16] = cells 160.., offs[16] = cells 176..
#define LENS(i) lens[((((i) >> 1) * 64 + lane) << 1) + ((i) & 1)]

struct BitIn {
    const uint32_t *w2;     // address of the dword that a2 holds / is being loaded into a2
    uint32_t a0, a1, a2;    // a0, a1: dwords ready to enter the bit buffer; a2: in flight (requested at the end of the previous iteration)
    uint64_t buf;
    int cnt;
    bool shift;             // a0 was consumed this iteration: a0 <- a1 <- a2 once the wave has waited for its loads
};
// main-path refill: at most ONE per iteration (no memory access here - the reload happens at the end of the iteration)
__device__ __forceinline__ void refill(BitIn &b) {
    if (b.cnt <= 32) { b.buf |= (uint64_t)b.a0 << b.cnt; b.cnt += 32; b.shift = true; }
}
// header-path refill: may run many times inside one iteration, so it loads (and waits) on the spot; rare
__device__ __forceinline__ void refill_now(BitIn &b, const uint32_t *w_end) {
    if (b.cnt <= 32) { b.buf |= (uint64_t)b.a0 << b.cnt; b.cnt += 32; b.a0 = b.a1; b.a1 = b.a2; if (b.w2 < w_end) ++b.w2; b.a2 = *b.w2; }
}
__device__ __forceinline__ uint32_t take(BitIn &b, int n) { const uint32_t v = (uint32_t)b.buf & ((1u << n) - 1u); b.buf >>= n; b.cnt -= n; return v; }
__device__ __forceinline__ uint32_t peek15(const BitIn &b) { return __brev((uint32_t)b.buf) >> 17; }   // next 15 stream bits, first bit = MSB

// Canonical decode by limits: lim[len] = (first_code[len] + count[len]) << (15 - len) is non-decreasing in len, the code's length is the
// smallest len with peek < lim[len]; its symbol sits at base[len] + (peek >> (15 - len)) in the (length, symbol)-sorted list.
// Branch-free over the lengths LO..15, so the lanes of a wave never serialise on it.  Returns the length (0 = invalid code).
template <int LO>
__device__ __forceinline__ int decode_limit(uint32_t peek, const uint16_t *lim, const uint16_t *base, int lane, uint32_t &index) {
    int sel = 0;
#pragma unroll
    for (int len = 15; len >= LO; --len) sel = peek < (uint32_t)L(lim, len - LO) ? len : sel;
    const int s = sel ? sel : LO;
    index = (uint16_t)(L(base, s - LO) + (peek >> (15 - s)));
    return sel;
}

// lengths (bytes, [i][lane] in `lens`) -> (length, symbol)-sorted symbols + limit/base per length LO..15.  cnt/offs: 16-entry scratch.
// cnt8[k] receives the number of codes of length k (k = 1..8) for the fast table.  false when over-subscribed.
template <int LO, class SymT>
__device__ __forceinline__ bool build_canon(const uint8_t *lens, int first, int n, SymT *sym, uint16_t *lim, uint16_t *base, uint16_t *cnt, uint16_t *offs,
                                            int lane, int *cnt8) {
    for (int l = 0; l < 16; ++l) L(cnt, l) = 0;
    for (int s = 0; s < n; ++s) { const int l = LENS(first + s); L(cnt, l) = L(cnt, l) + 1; }
    int left = 1;
    for (int l = 1; l < 16; ++l) { left <<= 1; left -= L(cnt, l); if (left < 0) return false; }
    L(offs, 1) = 0;
    for (int l = 1; l < 15; ++l) L(offs, l + 1) = L(offs, l) + L(cnt, l);
    for (int s = 0; s < n; ++s) { const int l = LENS(first + s); if (l) { const int o = L(offs, l); L(sym, o) = (SymT)s; L(offs, l) = o + 1; } }
    int code = 0, index = 0;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        const int c = L(cnt, l);
        if (l >= LO) { L(lim, l - LO) = (uint16_t)((code + c) << (15 - l)); L(base, l - LO) = (uint16_t)(index - code); }
        if (cnt8 && l <= 8) cnt8[l] = c;
        index += c; code = (code + c) << 1;
    }
    return true;                                                       // incomplete sets are legal while unused; a bad code decodes to "invalid"
}

// fast table (codes of length <= 8) from the sorted symbol list and the per-length counts
__device__ __forceinline__ void build_fast(uint16_t *fast, const uint16_t *sym, const int *cnt8, int lane) {
    for (int k = 0; k < 256; ++k) L(fast, k) = 0;
    int code = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= 8; ++len) {
        const int c = cnt8[len];
        for (int j = 0; j < c; ++j) {
            const uint32_t rev = __brev((uint32_t)(code + j)) >> (32 - len);
            const uint16_t e = (uint16_t)((L(sym, index + j) << 4) | len);
            for (uint32_t k = rev; k < 256; k += 1u << len) L(fast, k) = e;
        }
        index += c; code = (code + c) << 1;
    }
}

// The loop below is ONE state machine per lane - header / stored byte / match byte / symbol - iterated in lockstep by the wave, so a lane in
// a long match or at a block boundary never makes the other 63 wait for more than one iteration's worth of its branch.
// A match copies up to 4 bytes per iteration; its source bytes are REQUESTED one iteration before they are used, so the HBM/L2 round trip hides
// behind a whole iteration of the other lanes' work.
// Output bytes go to a 64-byte LDS ring per lane and reach HBM as aligned 32-byte segments (2 x dwordx4), flushed at a wave-uniform
// cadence: few stores, so the in-order vmcnt queue does not stall the input prefetch behind them.
__global__ void __launch_bounds__(64) k_bgzf_inflate(const uint8_t *__restrict__ in, const InflateBlock *__restrict__ blk, int n_blk, uint8_t *out, unsigned *err) {
    __shared__ uint16_t s_fast[256 * 64];                                  // 32 KiB; a lane's column doubles as its scratch while it parses a header
    __shared__ uint16_t s_lsym[288 * 64];
    __shared__ uint16_t s_llim[8 * 64], s_lbase[8 * 64];                   // litlen lengths 9..15
    __shared__ uint8_t s_dsym[32 * 64];
    __shared__ uint16_t s_dlim[16 * 64], s_dbase[16 * 64];                 // distance (and code-length code) lengths 1..15
    __shared__ uint32_t s_ring[16 * 64];                                   // 64 output bytes per lane, slot = global address & 63
    const int lane = threadIdx.x, bi = blockIdx.x * 64 + lane;
    const bool active = bi < n_blk;
    const InflateBlock B = active ? blk[bi] : InflateBlock{0, 0, 0, 0};
    const uint64_t gbase = B.out_off;                                       // global byte offset of this block's output
    uint8_t *o = out + gbase; uint32_t op = 0, fl = 0; const uint32_t on = B.out_len;
    const uint8_t *ip = in + B.in_off;
    BitIn b;
    {   // aligned dword stream (pointer arithmetic on the kernel argument keeps these GLOBAL loads: a flat load would drag lgkmcnt into every
        // wait); `in` is 256-byte aligned.  Bits past the block's end are never consumed by a valid stream (checked at the end).
        const int sh = (int)(B.in_off & 3);
        const uint32_t *w = reinterpret_cast<const uint32_t *>(in) + (B.in_off >> 2);
        const uint32_t w0 = w[0]; b.a0 = w[1]; b.a1 = w[2]; b.a2 = w[3]; b.w2 = w + 3;
        b.buf = (uint64_t)(w0 >> (8 * sh)); b.cnt = 32 - 8 * sh; b.shift = false;
    }
    uint8_t *lens = reinterpret_cast<uint8_t *>(s_fast);
    uint16_t *t_cnt = s_fast + 160 * 64, *t_offs = s_fast + 176 * 64;
    uint8_t *ring = reinterpret_cast<uint8_t *>(s_ring);
    const uint32_t gb6 = (uint32_t)gbase & 63u;
    auto ring_at = [&](uint32_t pos) -> uint8_t & { const uint32_t slot = (gb6 + pos) & 63u; return ring[((slot >> 2) * 64 + lane) * 4 + (slot & 3)]; };
    auto hist = [&](uint32_t pos) -> uint8_t { return pos >= fl ? ring_at(pos) : __hip_atomic_load(o + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto flush_segment = [&]() {                                            // write [fl, next 32-byte boundary) if complete
        const uint32_t nb = (uint32_t)((((gbase + fl) | 31ull) + 1ull) - gbase);
        if (op < nb) return;
        if (nb - fl == 32) {
            const uint32_t d0 = ((uint32_t)(gbase + fl) & 63u) >> 2;
            uint4 x, y;
            x.x = s_ring[(d0 + 0) * 64 + lane]; x.y = s_ring[(d0 + 1) * 64 + lane]; x.z = s_ring[(d0 + 2) * 64 + lane]; x.w = s_ring[(d0 + 3) * 64 + lane];
            y.x = s_ring[(d0 + 4) * 64 + lane]; y.y = s_ring[(d0 + 5) * 64 + lane]; y.z = s_ring[(d0 + 6) * 64 + lane]; y.w = s_ring[(d0 + 7) * 64 + lane];
            uint4 *dst = reinterpret_cast<uint4 *>(o + fl); dst[0] = x; dst[1] = y;
        } else { for (uint32_t k = fl; k < nb; ++k) o[k] = ring_at(k); }
        fl = nb;
    };
    enum { ST_HDR = 0, ST_SYM = 1, ST_STORED = 2, ST_DONE = 3, ST_DIST = 4 };
    int state = (active && on) ? ST_HDR : ST_DONE; bool last = false; unsigned e = 0;
    uint32_t mlen = 0, msrc = 0, slen = 0, iter = 0, mpend = 0;
    uint32_t p_lo = 0, p_hi = 0, p_sh = 0; bool have_pend = false;
    const uint32_t *out32 = reinterpret_cast<const uint32_t *>(out);        // `out` is 256-byte aligned
    // One iteration = (1) lanes without a pending match decode one symbol / header / stored byte, (2) lanes with a pending match copy up to 4 bytes
    // whose source words were REQUESTED at the end of the previous iteration, (3) every lane requests what it needs next: the following input dword
    // and, for a pending match, the next source words.  All loads of (3) are consumed only after phase (1) of the next iteration, so the one wait per
    // iteration finds them complete; the lockstep keeps a lane in a long match or a header from stalling the other 63 for more than its own branch.
    // a valid stream never looks further than its own bytes (+ the 3 dwords fetched ahead); a crafted one (e.g. an endless run of empty
    // blocks) must not walk off the buffer
    const uint32_t *w_end = reinterpret_cast<const uint32_t *>(in) + ((B.in_off + B.in_len + 3) >> 2) + 4;
    // ... and must not spin: every iteration of a valid stream consumes input bits or produces output, an empty stored / fixed block costs 3 iterations
    // for at least 10 bits - anything beyond this bound is a crafted stream (the header path clamps its reads at w_end, so the guard above alone
    // would never trip on an endless run of empty blocks)
    const uint32_t iter_cap = 16u * (B.in_len * 8u + on) + 4096u;
    while (state != ST_DONE) {
        if (b.w2 > w_end || iter > iter_cap) { e = LPS_INF_ERR_DATA; break; }
        const bool copying = mlen != 0;
        if (!copying) {
            if (state == ST_SYM) {
                refill(b);
                int sym; const uint32_t fe = L(s_fast, (uint32_t)b.buf & 255u);
                if (fe) { const int l = fe & 15; sym = (int)(fe >> 4); b.buf >>= l; b.cnt -= l; }
                else { uint32_t idx; const int l = decode_limit<9>(peek15(b), s_llim, s_lbase, lane, idx); if (!l || idx >= 288) { e = LPS_INF_ERR_DATA; sym = 256; last = true; } else { sym = L(s_lsym, idx); b.buf >>= l; b.cnt -= l; } }
                if (sym < 256) {
                    if (op >= on) { e = LPS_INF_ERR_OVERRUN; state = ST_DONE; }
                    else {
                        ring_at(op) = (uint8_t)sym; ++op;
                        // a second literal in the same iteration when the next code is a short one (<= 8 bits; at least 18 bits are still buffered)
                        const uint32_t f2 = L(s_fast, (uint32_t)b.buf & 255u);
                        if (f2 && (f2 >> 4) < 256u && op < on) { const int l2 = f2 & 15; b.buf >>= l2; b.cnt -= l2; ring_at(op) = (uint8_t)(f2 >> 4); ++op; }
                    }
                }
                else if (sym == 256) { state = last ? ST_DONE : ST_HDR; }
                else {
                    sym -= 257;
                    if (sym >= 29) { e = LPS_INF_ERR_DATA; state = ST_DONE; }
                    else {
                        if (sym < 8) mpend = 3 + sym; else if (sym == 28) mpend = 258; else { const int x = (sym >> 2) - 1; mpend = ((4u + (sym & 3)) << x) + 3 + take(b, x); }
                        state = ST_DIST;                                   // the distance code is read next iteration (one refill per iteration)
                    }
                }
            } else if (state == ST_DIST) {
                refill(b);
                uint32_t idx; const int l = decode_limit<1>(peek15(b), s_dlim, s_dbase, lane, idx);
                const int ds = (l && idx < 30) ? (int)L(s_dsym, idx) : 30;
                if (ds >= 30) { e = LPS_INF_ERR_DATA; state = ST_DONE; }
                else {
                    b.buf >>= l; b.cnt -= l;
                    uint32_t dist;
                    if (ds < 4) dist = 1 + ds; else { const int x = (ds >> 1) - 1; dist = ((2u + (ds & 1)) << x) + 1 + take(b, x); }
                    if (dist > op || op + mpend > on) { e = dist > op ? LPS_INF_ERR_DATA : LPS_INF_ERR_OVERRUN; state = ST_DONE; }
                    else { mlen = mpend; msrc = op - dist; state = ST_SYM; }
                }
            } else if (state == ST_STORED) {
                if (slen == 0) { state = last ? ST_DONE : ST_HDR; }
                else { refill(b); ring_at(op) = (uint8_t)take(b, 8); ++op; --slen; }
            } else {                                                       // ST_HDR: block header (+ Huffman tables)
                refill_now(b, w_end);
                last = take(b, 1); const uint32_t type = take(b, 2);
                if (type == 0) {
                    take(b, b.cnt & 7); refill_now(b, w_end);
                    const uint32_t len = take(b, 16); refill_now(b, w_end); const uint32_t nlen = take(b, 16);
                    if ((len ^ 0xffffu) != nlen || op + len > on) { e = LPS_INF_ERR_DATA; state = ST_DONE; }
                    else { slen = len; state = ST_STORED; }
                } else if (type == 3) { e = LPS_INF_ERR_DATA; state = ST_DONE; }
                else {
                    int nlen = 288, ndist = 30; bool bad = false;
                    if (type == 1) {                                       // fixed codes (RFC 1951 3.2.6)
                        for (int s = 0; s < 144; ++s) LENS(s) = 8;
                        for (int s = 144; s < 256; ++s) LENS(s) = 9;
                        for (int s = 256; s < 280; ++s) LENS(s) = 7;
                        for (int s = 280; s < 288; ++s) LENS(s) = 8;
                        for (int s = 288; s < 318; ++s) LENS(s) = 5;
                    } else {                                               // dynamic: code-length code, then the two length vectors
                        nlen = (int)take(b, 5) + 257; ndist = (int)take(b, 5) + 1; const int ncode = (int)take(b, 4) + 4;
                        if (nlen > 286 || ndist > 30) bad = true;
                        for (int s = 0; s < 19; ++s) LENS(s) = 0;
                        for (int k = 0; k < ncode && !bad; ++k) {
                            refill_now(b, w_end);
                            const int pos = k < 3 ? 16 + k : k == 3 ? 0 : (k & 1) ? 8 - ((k - 3) >> 1) : 7 + ((k - 2) >> 1);   // 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                            LENS(pos) = (uint8_t)take(b, 3);
                        }
                        if (!bad && !build_canon<1, uint8_t>(lens, 0, 19, s_dsym, s_dlim, s_dbase, t_cnt, t_offs, lane, nullptr)) bad = true;
                        int idx = 0;
                        while (!bad && idx < nlen + ndist) {
                            refill_now(b, w_end);
                            uint32_t si; const int l = decode_limit<1>(peek15(b), s_dlim, s_dbase, lane, si);
                            if (!l || l > 7 || si >= 19) { bad = true; break; }
                            const int sym = L(s_dsym, si); b.buf >>= l; b.cnt -= l;
                            if (sym < 16) { LENS(idx) = (uint8_t)sym; ++idx; continue; }
                            int rep, val = 0;
                            if (sym == 16) { if (idx == 0) { bad = true; break; } val = LENS(idx - 1); rep = 3 + (int)take(b, 2); }
                            else if (sym == 17) rep = 3 + (int)take(b, 3);
                            else rep = 11 + (int)take(b, 7);
                            if (idx + rep > nlen + ndist) { bad = true; break; }
                            while (rep--) { LENS(idx) = (uint8_t)val; ++idx; }
                        }
                        if (!bad && LENS(256) == 0) bad = true;            // no end-of-block code
                    }
                    int cnt8[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                    if (!bad && !build_canon<9, uint16_t>(lens, 0, nlen, s_lsym, s_llim, s_lbase, t_cnt, t_offs, lane, cnt8)) bad = true;
                    if (!bad && !build_canon<1, uint8_t>(lens, nlen, ndist, s_dsym, s_dlim, s_dbase, t_cnt, t_offs, lane, nullptr)) bad = true;
                    if (bad) { e = LPS_INF_ERR_DATA; state = ST_DONE; }
                    else { build_fast(s_fast, s_lsym, cnt8, lane); state = ST_SYM; }   // overwrites the scratch
                }
            }
        }
        // ---- (2) after the decode work: the loads requested last iteration are consumed here
        if (copying) {
            if (have_pend) {
                const uint32_t word = __builtin_amdgcn_alignbyte(p_hi, p_lo, p_sh);
                const uint32_t k = mlen < 4u ? mlen : 4u;
                ring_at(op) = (uint8_t)word;
                if (k > 1) ring_at(op + 1) = (uint8_t)(word >> 8);
                if (k > 2) ring_at(op + 2) = (uint8_t)(word >> 16);
                if (k > 3) ring_at(op + 3) = (uint8_t)(word >> 24);
                op += k; msrc += k; mlen -= k;
            } else { const uint8_t v = hist(msrc); ring_at(op) = v; ++op; ++msrc; --mlen; }   // source in the ring or straddling the flushed boundary
        }
        if (b.shift) { b.a0 = b.a1; b.a1 = b.a2; ++b.w2; b.shift = false; }
        // ---- (3) requests for the next iteration (unconditional loads: lanes with nothing to fetch re-read their input dword, an L1 hit)
        have_pend = mlen != 0 && msrc + 4 <= fl && state != ST_DONE;
        {
            const uint64_t a = gbase + msrc;
            const uint32_t *q = have_pend ? out32 + (a >> 2) : b.w2;
            p_sh = (uint32_t)a & 3u;
            p_lo = q[0]; p_hi = q[1];
            b.a2 = *b.w2;
        }
        if ((++iter & 7u) == 0) flush_segment();
    }
    if (active && on) {
        if (!e) {
            if (op != on || mlen) e = LPS_INF_ERR_SIZE;
            // bytes consumed: everything before a0 (= w2 - 2 dwords) minus the whole bytes left in the bit buffer
            const uint64_t consumed = (uint64_t)((const uint8_t *)b.w2 - ip) - 8 - (uint64_t)(b.cnt >> 3);
            if (consumed > B.in_len) e = LPS_INF_ERR_DATA;
        }
        while (fl < op) { const uint32_t before = fl; flush_segment(); if (fl == before) { for (uint32_t k = fl; k < op; ++k) o[k] = ring_at(k); fl = op; } }
        if (e) atomicOr(err, e);
    }
}

// CRC32 of every inflated block against the value stored in its gzip trailer (htslib verifies it on every read): wave per block, per-lane
// slices combined as state_i * x^(8 * bytes after slice i) mod P.
__device__ __forceinline__ uint32_t crc_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) { if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; } m >>= 1; b = (b & 1) ? (b >> 1) ^ 0xedb88320u : b >> 1; }
    return p;
}
// slicing-by-4: four tables, one dependent step per 4 bytes; two independent halves of the lane's slice are interleaved so that the LDS latency of one
// overlaps the other.  Slices are 16-byte multiples from the block start; the words are fetched with two aligned loads + v_alignbyte.
__global__ void __launch_bounds__(256) k_bgzf_crc(const uint8_t *in, const InflateBlock *blk, int n_blk, const uint8_t *out, unsigned *err) {
    __shared__ uint32_t tab[4][256], x2n[32];
    for (int k = threadIdx.x; k < 256; k += 256) { uint32_t c = (uint32_t)k; for (int j = 0; j < 8; ++j) c = (c & 1) ? (c >> 1) ^ 0xedb88320u : c >> 1; tab[0][k] = c; }
    __syncthreads();
    for (int k = threadIdx.x; k < 256; k += 256) { uint32_t c = tab[0][k]; for (int t = 1; t < 4; ++t) { c = tab[0][c & 255u] ^ (c >> 8); tab[t][k] = c; } }
    if (threadIdx.x < 32) { uint32_t p = 1u << 30; for (int k = 0; k < (int)threadIdx.x; ++k) p = crc_multmodp(p, p); x2n[threadIdx.x] = p; }
    __syncthreads();
    const int bi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (bi >= n_blk) return;
    const InflateBlock B = blk[bi]; const uint32_t len = B.out_len;
    const uint32_t per = (((len + 63) / 64) + 31) & ~31u, s0 = min(len, per * lane), s1 = min(len, s0 + per), half = per / 2;   // per: multiple of 32 -> two 16-byte-multiple halves
    const uint32_t a0 = s0, a1 = min(s1, s0 + half), b0 = a1, b1 = s1;
    const uint32_t *o32 = reinterpret_cast<const uint32_t *>(out); const uint64_t base = B.out_off;
    auto word = [&](uint32_t pos) -> uint32_t { const uint64_t a = base + pos; const uint32_t lo = o32[a >> 2], hi = o32[(a >> 2) + 1]; return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(a & 3)); };
    auto step4 = [&](uint32_t c, uint32_t w) -> uint32_t { c ^= w; return tab[3][c & 255u] ^ tab[2][(c >> 8) & 255u] ^ tab[1][(c >> 16) & 255u] ^ tab[0][c >> 24]; };
    uint32_t ca = lane == 0 ? 0xffffffffu : 0u, cb = 0u; uint32_t pa = a0, pb = b0;
    while (pa + 4 <= a1 && pb + 4 <= b1) { const uint32_t wa = word(pa), wb = word(pb); ca = step4(ca, wa); cb = step4(cb, wb); pa += 4; pb += 4; }
    while (pa + 4 <= a1) { ca = step4(ca, word(pa)); pa += 4; }
    while (pb + 4 <= b1) { cb = step4(cb, word(pb)); pb += 4; }
    const uint8_t *o = out + base;
    for (; pa < a1; ++pa) ca = tab[0][(ca ^ o[pa]) & 255u] ^ (ca >> 8);
    for (; pb < b1; ++pb) cb = tab[0][(cb ^ o[pb]) & 255u] ^ (cb >> 8);
    auto shift = [&](uint32_t c, uint32_t n) { uint32_t p = 1u << 31; int k = 3; while (n) { if (n & 1) p = crc_multmodp(x2n[k & 31], p); n >>= 1; ++k; } return crc_multmodp(p, c); };
    uint32_t c = shift(ca, len - a1) ^ shift(cb, len - b1);                // state_i * x^(8 * bytes after the half)
    for (int s = 32; s; s >>= 1) c ^= __shfl_xor(c, s);
    c ^= 0xffffffffu;
    if (lane == 0) {
        const uint8_t *t = in + B.in_off + B.in_len;                     // trailer: CRC32, ISIZE
        const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (want != c) atomicOr(err, LPS_INF_ERR_CRC);
    }
}

void launch_bgzf_crc(const uint8_t *in, const InflateBlock *blk, int n_blk, const uint8_t *out, unsigned *err, hipStream_t s) {
    if (n_blk > 0) hipLaunchKernelGGL(k_bgzf_crc, dim3((n_blk + 3) / 4), dim3(256), 0, s, in, blk, n_blk, out, err);
}

void launch_bgzf_inflate(const uint8_t *in, const InflateBlock *blk, int n_blk, uint8_t *out, unsigned *err, hipStream_t s) {
    if (n_blk > 0) hipLaunchKernelGGL(k_bgzf_inflate, dim3((n_blk + 63) / 64), dim3(64), 0, s, in, blk, n_blk, out, err);
}
