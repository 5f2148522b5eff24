// lps_inflate.hip — BGZF (RFC 1951 DEFLATE in 64 KiB gzip members) decoded on the GPU, SURVEY.md §8f rank 1.
//
// What it replaces: htslib's bgzf.c block inflate behind sam_itr_multi_next (src/phase/ParsingBam.cpp:1279,
// src/haplotag/HaplotagParsingBam.cpp:453), the largest share of the reference's wall clock.
//
// One LANE per BGZF block (blocks are independent, <=64 KiB each; a chr20-30x BAM has ~50 k of them): the bit stream is
// inherently serial inside a block, so the parallelism is across blocks.  What limits the kernel is how many such lanes a CU holds: the
// Huffman tables of a lane live in LDS, and a lone wavefront per SIMD issues a dependent instruction only every ~6 cycles
// (profiles/r02_inflate_counters.md).  Round 2 put the tables on a diet - 548 bytes per lane instead of 1 280, FOUR wavefronts per CU, one on
// every SIMD, instead of two - laid out [entry][lane] so that lanes never share a word:
//   lsym8 u8[288] + lhi 36 B   litlen symbols sorted by (length, symbol), 9 bits each: low byte + a bit array
//   llim / lbase u16[16]        limit + list offset per code length 1..15 (branch-free canonical decode; no first-level table any more:
//                               the 8-bit table cost 512 B per lane, and with 64 lanes in lockstep some lane missed it every iteration anyway)
//   dsym u8[32], dlim / dbase u16[16]   same for distances; they also hold the code-length code while a dynamic header is read
//   ring  64 output bytes, written to HBM as aligned 32-byte segments
// A header's code lengths (<= 316) go to a per-lane scratch column in GLOBAL memory (written once, read once while the sorted lists are built:
// a few headers per block), the per-length counters and list cursors sit packed in registers.
// The input is read as aligned dwords, two dwords ahead of use.
// Errors (corrupt stream, output overrun) set a flag; nothing is written outside [out_off, out_off + out_len).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lps_inflate.h"

// members per workgroup = active lanes of its one wavefront (the tables of a lane are columns of stride INF_LANES).  32, not 64: the kernel is bound by
// the latency of its per-lane dependence chains, not by lanes - half-filled wavefronts need 17.5 KB of LDS each, so TWO fit a SIMD (228 VGPRs allow
// it) and one runs while the other waits: 29.3 -> 33.2 GB/s sustained; 16 lanes (four per SIMD) make it instruction-bound: 18.6 GB/s
#ifndef INF_LANES
#define INF_LANES 32
#endif
#define L(a, i) a[(i) * INF_LANES + lane]

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
// Branch-free over the lengths 1..15, so the lanes of a wave never serialise on it.  Returns the length (0 = invalid code).
__device__ __forceinline__ int decode_limit(uint32_t peek, const uint16_t *lim, const uint16_t *base, int lane, uint32_t &index) {
    int sel = 0;
#pragma unroll
    for (int len = 15; len >= 1; --len) sel = peek < (uint32_t)L(lim, len - 1) ? len : sel;
    const int s = sel ? sel : 1;
    index = (uint16_t)(L(base, s - 1) + (peek >> (15 - s)));
    return sel;
}

// sixteen 16-bit counters in four registers (per-length symbol counts / list cursors of a header): a lane's own, indexed at run time by selects
struct Pack16 { uint64_t a, b, c, d; };
__device__ __forceinline__ unsigned p16_get(const Pack16 &p, int k) {
    const uint64_t w = (k & 8) ? ((k & 4) ? p.d : p.c) : ((k & 4) ? p.b : p.a);
    return (unsigned)(w >> ((k & 3) * 16)) & 0xffffu;
}
__device__ __forceinline__ void p16_add(Pack16 &p, int k, unsigned v) {
    const uint64_t inc = (uint64_t)v << ((k & 3) * 16); const int q = k >> 2;
    p.a += q == 0 ? inc : 0ull; p.b += q == 1 ? inc : 0ull; p.c += q == 2 ? inc : 0ull; p.d += q == 3 ? inc : 0ull;
}
// counts per length -> limit/base per length 1..15 and the first list slot of every length.  false when over-subscribed; incomplete sets are
// legal while unused (a code that falls into the gap decodes to "invalid").
__device__ __forceinline__ bool canon_tables(const Pack16 &cnt, uint16_t *lim, uint16_t *base, int lane, Pack16 &offs) {
    bool ok = true; int left = 1;
#pragma unroll
    for (int l = 1; l < 16; ++l) { left <<= 1; left -= (int)p16_get(cnt, l); if (left < 0) ok = false; }
    int code = 0, index = 0; offs = Pack16{0, 0, 0, 0};
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        const int c = (int)p16_get(cnt, l);
        L(lim, l - 1) = (uint16_t)((code + c) << (15 - l)); L(base, l - 1) = (uint16_t)(index - code);
        p16_add(offs, l, (unsigned)index);
        index += c; code = (code + c) << 1;
    }
    return ok;
}

// The loop below is ONE state machine per lane - header / stored byte / match byte / symbol - iterated in lockstep by the wave, so a lane in
// a long match or at a block boundary never makes the other 63 wait for more than one iteration's worth of its branch.
// A match copies up to 4 bytes per iteration; its source bytes are REQUESTED one iteration before they are used, so the HBM/L2 round trip hides
// behind a whole iteration of the other lanes' work.
// Output bytes go to a 64-byte LDS ring per lane and reach HBM as aligned 32-byte segments (2 x dwordx4), flushed at a wave-uniform
// cadence: few stores, so the in-order vmcnt queue does not stall the input prefetch behind them.
#define INF_SCRATCH 320      // code lengths of one header per lane (<= 286 + 30), bytes; column layout [i][lane] per wavefront
__global__ void __launch_bounds__(INF_LANES) k_bgzf_inflate(const uint8_t *__restrict__ in, const InflateBlock *__restrict__ blk, int n_blk, uint8_t *out, unsigned *err,
                                                      uint8_t *scratch, const unsigned long long *uploaded, unsigned long long total_in, long long timeout_ticks) {
    __shared__ uint8_t s_lsym8[288 * INF_LANES];                                  // low byte of the (length, symbol)-sorted litlen symbols
    __shared__ uint8_t s_lhi[36 * INF_LANES];                                     // their ninth bit, 8 per byte
    __shared__ uint16_t s_llim[16 * INF_LANES], s_lbase[16 * INF_LANES];                 // litlen lengths 1..15
    __shared__ uint8_t s_dsym[32 * INF_LANES];
    __shared__ uint16_t s_dlim[16 * INF_LANES], s_dbase[16 * INF_LANES];                 // distance (and code-length code) lengths 1..15
    __shared__ uint32_t s_ring[16 * INF_LANES];                                   // 64 output bytes per lane, slot = global address & 63
    const int lane = threadIdx.x, bi = blockIdx.x * INF_LANES + lane;
    const bool active = bi < n_blk;
    if (uploaded) {
        // The upload may still be running (lps_bgzf_load launches this kernel beside it): wait until the bytes of this wavefront's members - and the
        // dwords the bit reader fetches ahead - are in place.  Workgroups are dispatched in member order and the upload runs in file order, faster than
        // the members are consumed, so the wait is short or none.  Chunk ends are multiples of 128 bytes: no cache line is ever read half-uploaded.
        // Every wave leaves this loop: the counter reaches total_in, or the wall clock (100 MHz) runs out and the launch reports a timeout.
        const InflateBlock last_member = blk[min(n_blk - 1, (int)blockIdx.x * INF_LANES + INF_LANES - 1)];
        unsigned long long need = last_member.in_off + last_member.in_len + 64ull;
        if (need > total_in) need = total_in;
        // the word lives in host memory: ONE lane reads it (one PCIe read per poll), and a wave that is far ahead of the upload sleeps for about half
        // of the time its bytes need at 40 GB/s before it looks again - a few polls per wave, not a stream of reads against the upload's direction
        const long long t0 = wall_clock64(); bool timed_out = false;
        for (;;) {
            unsigned long long have = 0;
            if (lane == 0) have = __hip_atomic_load(uploaded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            have = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(have >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)have);
            if (have >= need) break;
            const unsigned long long naps = (need - have) >> 18;            // 3.4 us each (s_sleep 127 = 8 128 cycles): 256 KiB of upload take 6.5 us
            for (unsigned long long k = 0, n_k = naps < 1 ? 1 : naps > 4096 ? 4096 : naps; k < n_k; ++k) __builtin_amdgcn_s_sleep(127);
            if (wall_clock64() - t0 > timeout_ticks) { timed_out = true; break; }
        }
        if (timed_out) { if (lane == 0) atomicOr(err, (unsigned)LPS_INF_ERR_TIMEOUT); return; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    const InflateBlock B = active ? blk[bi] : InflateBlock{0, 0, 0, 0};
    const uint64_t gbase = B.out_off;                                       // global byte offset of this block's output
    uint8_t *o = out + gbase; uint32_t op = 0, fl = 0; const uint32_t on = B.out_len;
    const uint8_t *ip = in + B.in_off;
    uint8_t *gl = scratch + (size_t)blockIdx.x * INF_SCRATCH * INF_LANES + lane;   // this lane's scratch column: entry i at gl[i * INF_LANES]
    BitIn b;
    {   // aligned dword stream (pointer arithmetic on the kernel argument keeps these GLOBAL loads: a flat load would drag lgkmcnt into every
        // wait); `in` is 256-byte aligned.  Bits past the block's end are never consumed by a valid stream (checked at the end).
        const int sh = (int)(B.in_off & 3);
        const uint32_t *w = reinterpret_cast<const uint32_t *>(in) + (B.in_off >> 2);
        const uint32_t w0 = w[0]; b.a0 = w[1]; b.a1 = w[2]; b.a2 = w[3]; b.w2 = w + 3;
        b.buf = (uint64_t)(w0 >> (8 * sh)); b.cnt = 32 - 8 * sh; b.shift = false;
    }
    uint8_t *ring = reinterpret_cast<uint8_t *>(s_ring);
    const uint32_t gb6 = (uint32_t)gbase & 63u;
    auto ring_at = [&](uint32_t pos) __attribute__((always_inline)) -> uint8_t & { const uint32_t slot = (gb6 + pos) & 63u; return ring[((slot >> 2) * INF_LANES + lane) * 4 + (slot & 3)]; };
    auto hist = [&](uint32_t pos) __attribute__((always_inline)) -> uint8_t { return pos >= fl ? ring_at(pos) : __hip_atomic_load(o + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto flush_segment = [&]() __attribute__((always_inline)) {                                            // write [fl, next 32-byte boundary) if complete
        const uint32_t nb = (uint32_t)((((gbase + fl) | 31ull) + 1ull) - gbase);
        if (op < nb) return;
        if (nb - fl == 32) {
            const uint32_t d0 = ((uint32_t)(gbase + fl) & 63u) >> 2;
            uint4 x, y;
            x.x = s_ring[(d0 + 0) * INF_LANES + lane]; x.y = s_ring[(d0 + 1) * INF_LANES + lane]; x.z = s_ring[(d0 + 2) * INF_LANES + lane]; x.w = s_ring[(d0 + 3) * INF_LANES + lane];
            y.x = s_ring[(d0 + 4) * INF_LANES + lane]; y.y = s_ring[(d0 + 5) * INF_LANES + lane]; y.z = s_ring[(d0 + 6) * INF_LANES + lane]; y.w = s_ring[(d0 + 7) * INF_LANES + lane];
            uint4 *dst = reinterpret_cast<uint4 *>(o + fl); dst[0] = x; dst[1] = y;
        } else { for (uint32_t k = fl; k < nb; ++k) o[k] = ring_at(k); }
        fl = nb;
    };
    // litlen symbol of sorted slot idx: low byte + ninth bit
    auto litlen_at = [&](uint32_t idx) __attribute__((always_inline)) -> int { return (int)L(s_lsym8, idx) | ((((int)s_lhi[(idx >> 3) * INF_LANES + lane] >> (idx & 7u)) & 1) << 8); };
    enum { ST_HDR = 0, ST_SYM = 1, ST_STORED = 2, ST_DONE = 3, ST_DIST = 4 };
    int state = (active && on) ? ST_HDR : ST_DONE; bool last = false;
    uint16_t e = 0;   // (16 bits on purpose: as an i32 its stores were merged with those of `op` through a pointer phi, which kept BOTH in scratch memory)
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
    auto decode_distance = [&]() __attribute__((always_inline)) {                                          // distance code + extra bits of the match whose length is in mpend
        uint32_t idx; const int l = decode_limit(peek15(b), s_dlim, s_dbase, lane, idx);
        const int ds = (l && idx < 30) ? (int)L(s_dsym, idx) : 30;
        if (ds >= 30) { e = LPS_INF_ERR_DATA; state = ST_DONE; return; }
        b.buf >>= l; b.cnt -= l;
        uint32_t dist;
        if (ds < 4) dist = 1 + ds; else { const int x = (ds >> 1) - 1; dist = ((2u + (ds & 1)) << x) + 1 + take(b, x); }
        if (dist > op || op + mpend > on) { e = dist > op ? LPS_INF_ERR_DATA : LPS_INF_ERR_OVERRUN; state = ST_DONE; }
        else { mlen = mpend; msrc = op - dist; state = ST_SYM; }
    };
    while (state != ST_DONE) {
        if (b.w2 > w_end || iter > iter_cap) { e = LPS_INF_ERR_DATA; break; }
        const bool copying = mlen != 0;
        if (!copying) {
            if (state == ST_SYM) {
                refill(b);
                int sym; uint32_t idx; const int l = decode_limit(peek15(b), s_llim, s_lbase, lane, idx);
                if (!l || idx >= 288) { e = LPS_INF_ERR_DATA; sym = 256; last = true; } else { sym = litlen_at(idx); b.buf >>= l; b.cnt -= l; }
                if (sym < 256) {
                    if (op >= on) { e = LPS_INF_ERR_OVERRUN; state = ST_DONE; }
                    else {
                        ring_at(op) = (uint8_t)sym; ++op;
                        // a second literal in the same iteration when the next code is one (at least 18 bits are still buffered: a code has 15 at most)
                        uint32_t i2; const int l2 = decode_limit(peek15(b), s_llim, s_lbase, lane, i2);
                        if (l2 && i2 < 288 && op < on) { const int s2 = litlen_at(i2); if (s2 < 256) { b.buf >>= l2; b.cnt -= l2; ring_at(op) = (uint8_t)s2; ++op; } }
                    }
                }
                else if (sym == 256) { state = last ? ST_DONE : ST_HDR; }
                else {
                    sym -= 257;
                    if (sym >= 29) { e = LPS_INF_ERR_DATA; state = ST_DONE; }
                    else {
                        if (sym < 8) mpend = 3 + sym; else if (sym == 28) mpend = 258; else { const int x = (sym >> 2) - 1; mpend = ((4u + (sym & 3)) << x) + 3 + take(b, x); }
                        // the distance code right away when its 28 bits at most (15 + 13 extra) are still buffered - a match then costs one iteration
                        // less; else next iteration, after its refill (one refill per iteration)
                        if (b.cnt >= 28) decode_distance(); else state = ST_DIST;
                    }
                }
            } else if (state == ST_DIST) {
                refill(b);
                decode_distance();
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
                    Pack16 cl{0, 0, 0, 0}, cd{0, 0, 0, 0}, offs;          // symbols per code length: litlen, distance
                    if (type == 1) {                                       // fixed codes (RFC 1951 3.2.6)
                        // (one rolled loop: unrolled, the 318 store addresses were hoisted out of the main loop and held in registers for the
                        // whole kernel - 338 registers, the hot path reading its state back from AGPRs)
#pragma unroll 1
                        for (int s = 0; s < 318; ++s) gl[s * INF_LANES] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
                        p16_add(cl, 7, 24); p16_add(cl, 8, 152); p16_add(cl, 9, 112); p16_add(cd, 5, 30);
                    } else {                                               // dynamic: code-length code, then the two length vectors
                        nlen = (int)take(b, 5) + 257; ndist = (int)take(b, 5) + 1; const int ncode = (int)take(b, 4) + 4;
                        if (nlen > 286 || ndist > 30) bad = true;
                        uint64_t clv = 0;                                  // the 19 lengths of the code-length code, 3 bits each
                        for (int k = 0; k < ncode && !bad; ++k) {
                            refill_now(b, w_end);
                            const int pos = k < 3 ? 16 + k : k == 3 ? 0 : (k & 1) ? 8 - ((k - 3) >> 1) : 7 + ((k - 2) >> 1);   // 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                            clv |= (uint64_t)take(b, 3) << (3 * pos);
                        }
                        Pack16 cc{0, 0, 0, 0};
                        for (int s = 0; s < 19; ++s) p16_add(cc, (int)((clv >> (3 * s)) & 7u), 1);
                        if (!bad && !canon_tables(cc, s_dlim, s_dbase, lane, offs)) bad = true;
                        for (int s = 0; s < 19 && !bad; ++s) { const int l = (int)((clv >> (3 * s)) & 7u); if (l) { L(s_dsym, p16_get(offs, l)) = (uint8_t)s; p16_add(offs, l, 1); } }
                        int idx = 0, prev = 0; bool eob = false;
                        auto put_len = [&](int v) __attribute__((always_inline)) {
                            gl[idx * INF_LANES] = (uint8_t)v;
                            if (idx < nlen) p16_add(cl, v, 1); else p16_add(cd, v, 1);
                            if (idx == 256 && v) eob = true;
                            prev = v; ++idx;
                        };
                        while (!bad && idx < nlen + ndist) {
                            refill_now(b, w_end);
                            uint32_t si; const int l = decode_limit(peek15(b), s_dlim, s_dbase, lane, si);
                            if (!l || l > 7 || si >= 19) { bad = true; break; }
                            const int sym = L(s_dsym, si); b.buf >>= l; b.cnt -= l;
                            if (sym < 16) { put_len(sym); continue; }
                            int rep, val = 0;
                            if (sym == 16) { if (idx == 0) { bad = true; break; } val = prev; rep = 3 + (int)take(b, 2); }
                            else if (sym == 17) rep = 3 + (int)take(b, 3);
                            else rep = 11 + (int)take(b, 7);
                            if (idx + rep > nlen + ndist) { bad = true; break; }
                            while (rep--) put_len(val);
                        }
                        if (!bad && !eob) bad = true;                      // no end-of-block code
                    }
                    // litlen list: symbols placed by (length, symbol); the lengths come back from the scratch column
                    if (!bad && !canon_tables(cl, s_llim, s_lbase, lane, offs)) bad = true;
                    if (!bad) {
                        for (int k = 0; k < 36; ++k) L(s_lhi, k) = 0;
                        for (int s = 0; s < nlen; ++s) {
                            const int l = gl[s * INF_LANES];
                            if (l) { const unsigned at = p16_get(offs, l); L(s_lsym8,
                                    at) = (uint8_t)s; if (s >> 8) s_lhi[(at >> 3) * INF_LANES + lane] |= (uint8_t)(1u << (at & 7u)); p16_add(offs, l, 1); }
                        }
                    }
                    if (!bad && !canon_tables(cd, s_dlim, s_dbase, lane, offs)) bad = true;
                    if (!bad) for (int s = 0; s < ndist; ++s) { const int l = gl[(nlen + s) * INF_LANES]; if (l) { L(s_dsym, p16_get(offs, l)) = (uint8_t)s; p16_add(offs, l, 1); } }
                    if (bad) { e = LPS_INF_ERR_DATA; state = ST_DONE; } else state = ST_SYM;
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
    auto word = [&](uint32_t pos) -> uint32_t { const uint64_t a = base + pos; const uint32_t lo = o32[a >> 2],
            hi = o32[(a >> 2) + 1]; return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(a & 3)); };
    auto step4 = [&](uint32_t c, uint32_t w) -> uint32_t { c ^= w; return tab[3][c & 255u] ^ tab[2][(c >> 8) & 255u] ^ tab[1][(c >> 16) & 255u] ^ tab[0][c >> 24]; };
    uint32_t ca = lane == 0 ? 0xffffffffu : 0u, cb = 0u; uint32_t pa = a0, pb = b0;
    // 16 bytes of each half per trip: five dwords per half (one 16-byte load + one dword; `word` alone costs two loads per FOUR bytes, and every lane's
    // loads go to lines of their own - the load instructions, not the table lookups, were what the kernel waited for)
    struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
    while (pa + 16 <= a1 && pb + 16 <= b1) {
        const uint64_t ga = base + pa, gb = base + pb;
        const uint32_t *qa = o32 + (ga >> 2), *qb = o32 + (gb >> 2);
        const U4 xa = *reinterpret_cast<const U4 *>(qa), xb = *reinterpret_cast<const U4 *>(qb); const uint32_t ya = qa[4], yb = qb[4];
        const uint32_t sa = (uint32_t)(ga & 3), sb = (uint32_t)(gb & 3);
        ca = step4(ca, __builtin_amdgcn_alignbyte(xa.y, xa.x, sa)); cb = step4(cb, __builtin_amdgcn_alignbyte(xb.y, xb.x, sb));
        ca = step4(ca, __builtin_amdgcn_alignbyte(xa.z, xa.y, sa)); cb = step4(cb, __builtin_amdgcn_alignbyte(xb.z, xb.y, sb));
        ca = step4(ca, __builtin_amdgcn_alignbyte(xa.w, xa.z, sa)); cb = step4(cb, __builtin_amdgcn_alignbyte(xb.w, xb.z, sb));
        ca = step4(ca, __builtin_amdgcn_alignbyte(ya, xa.w, sa));   cb = step4(cb, __builtin_amdgcn_alignbyte(yb, xb.w, sb));
        pa += 16; pb += 16;
    }
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

size_t bgzf_inflate_scratch_bytes(int n_blk) { return (size_t)((n_blk + INF_LANES - 1) / INF_LANES) * INF_SCRATCH * INF_LANES; }
void launch_bgzf_inflate(const uint8_t *in, const InflateBlock *blk, int n_blk, uint8_t *out, unsigned *err, uint8_t *scratch, hipStream_t s,
                         const unsigned long long *uploaded, unsigned long long total_in, double timeout_ms) {
    const long long ticks = (long long)(timeout_ms * 1e5);               // wall_clock64 counts at 100 MHz
    if (n_blk > 0) hipLaunchKernelGGL(k_bgzf_inflate, dim3((n_blk + INF_LANES - 1) / INF_LANES), dim3(INF_LANES), 0, s, in, blk, n_blk, out, err, scratch, uploaded, total_in, ticks);
}
