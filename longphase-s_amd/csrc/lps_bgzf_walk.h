// lps_bgzf_walk.h — the BGZF header walk (host only, header only): the table of the blocks of a byte range, each {where its deflate bytes start, where
// its output starts, both lengths}.  Needs neither the GPU nor the library: lps_abi.hip makes its tables with it (lps_bgzf_walk_fd, lps_bgzf_load*), and
// the command line includes it to walk a file on helper threads from its first instruction on, before the library is even loaded (cli/cli_bam.h).
// Blk: any aggregate {uint64 in_off, out_off; uint32 in_len, out_len} (InflateBlock of lps_inflate.h, lps_bgzf_block of include/lps_abi.h).
#pragma once
#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <unistd.h>

// bytes of a BGZF stream: memory, or a file descriptor read with pread from `base` on
struct ZSource {
    const uint8_t *mem = nullptr; int fd = -1; uint64_t base = 0;
    bool read(uint64_t pos, size_t len, uint8_t *dst) const {
        if (mem) { memcpy(dst, mem + pos, len); return true; }
        while (len) {
            const ssize_t r = pread(fd, dst, len, (off_t)(base + pos));
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) return false;
            dst += r; pos += (uint64_t)r; len -= (size_t)r;
        }
        return true;
    }
};

// One BGZF block header (18 bytes + extra subfields; RFC 1952 member with the BC subfield, SAM spec 4.1) in h[0, avail) = the file's bytes from p on
// -> BSIZE; 0 when it is none; ~0 when the extra field reaches beyond `avail` (the caller reads more)
static inline uint64_t bgzf_parse_header(const uint8_t *h, size_t avail, uint64_t p, uint64_t n, unsigned &xlen) {
    if (avail < 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return 0;
    xlen = h[10] | (h[11] << 8);
    if (12ull + xlen > avail) return p + 12ull + xlen > n ? 0 : ~0ull;
    uint64_t q = 12, bsize = 0;
    while (q + 4 <= 12ull + xlen) {
        const unsigned slen = h[q + 2] | (h[q + 3] << 8);
        if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2 && q + 6 <= 12ull + xlen) bsize = (uint64_t)(h[q + 4] | (h[q + 5] << 8)) + 1;
        q += 4 + slen;
    }
    if (!bsize || bsize < 12ull + xlen + 8 || p + bsize > n) return 0;
    return bsize;
}
// The block table of [from, to): false when a header is bad, a block is larger than 64 KiB, the source cannot be read or the chain does not land on
// `to` exactly.  out_off is relative to the piece's first block (utot = the piece's inflated size).  One read per block: the ISIZE at a block's end
// and the header behind it come together.
template <class Blk>
static bool bgzf_walk_piece(const ZSource &z, uint64_t n, uint64_t from, uint64_t to, std::vector<Blk> &blks, uint64_t &utot, uint64_t *bad_at = nullptr) {
    uint64_t p = from; utot = 0; uint8_t h[64], t[68]; size_t have = 0; std::vector<uint8_t> wide;
    while (p < to) {
        if (bad_at) *bad_at = p;
        const size_t want = (size_t)std::min<uint64_t>(sizeof h, n - p);
        if (have < want) { if (!z.read(p + have, want - have, h + have)) return false; have = want; }
        unsigned xlen = 0; uint64_t bsize = bgzf_parse_header(h, have, p, n, xlen);
        if (bsize == ~0ull) { wide.resize(12 + (size_t)xlen); if (!z.read(p, wide.size(), wide.data())) return false; bsize = bgzf_parse_header(wide.data(), wide.size(), p, n, xlen); }
        if (!bsize || bsize == ~0ull) return false;
        const size_t tl = (size_t)std::min<uint64_t>(sizeof t, n - (p + bsize - 4));
        if (!z.read(p + bsize - 4, tl, t)) return false;
        const uint64_t isize = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
        if (isize > 65536) return false;
        blks.push_back(Blk{p + 12 + xlen, utot, (uint32_t)(bsize - 12 - xlen - 8), (uint32_t)isize});
        utot += isize; p += bsize;
        have = tl - 4; memcpy(h, t + 4, have);
    }
    return p == to;
}
// The header walk is latency, not bytes (one small read per 20 - 30 KB block; 0.2 s for 8 GB on one thread): T threads take a piece each.  A piece
// starts at the first position behind k * n / T where FOUR block headers follow one another; the piece before it must end exactly there, otherwise
// (and whenever anything else looks wrong) the caller walks the file serially.
template <class Blk>
static bool bgzf_walk_parallel(const ZSource &z, uint64_t n, std::vector<Blk> &blks, uint64_t &utot) {
    // pieces walked side by side: the walk is one small read per 20 - 30 KB block - latency, not bytes - and it runs beside the GPU start-up, which it
    // must not outlast (8 pieces took 0.2 s for 12 GB: longer than the rest of the start-up)
    const int T = (int)std::max(8u, std::min(32u, std::thread::hardware_concurrency() / 2u));
    if (n < (64ull << 20)) return false;
    std::vector<uint64_t> seed((size_t)T + 1, 0); seed[(size_t)T] = n;
    std::vector<uint8_t> win((1u << 20) + 64);
    for (int k = 1; k < T; ++k) {
        const uint64_t w0 = n * (uint64_t)k / T; const size_t wl = (size_t)std::min<uint64_t>(win.size(), n - w0);
        if (!z.read(w0, wl, win.data())) return false;
        uint64_t found = 0;
        for (size_t i = 0; i + 18 <= wl && i < (1u << 20) && !found; ++i) {
            if (win[i] != 31 || win[i + 1] != 139 || win[i + 2] != 8 || !(win[i + 3] & 4)) continue;
            uint64_t q = w0 + i; int chain = 0; uint8_t h[64];
            for (; chain < 4 && q < n; ++chain) {
                const size_t hl = (size_t)std::min<uint64_t>(sizeof h, n - q); unsigned xl = 0;
                if (!z.read(q, hl, h)) return false;
                const uint64_t b = bgzf_parse_header(h, hl, q, n, xl);
                if (!b || b == ~0ull) break;
                q += b;
            }
            if (chain == 4 || (chain > 0 && q == n)) found = w0 + i;
        }
        if (!found) return false;
        seed[(size_t)k] = found;
    }
    std::vector<std::vector<Blk>> part((size_t)T); std::vector<uint64_t> ut((size_t)T, 0); std::vector<char> ok((size_t)T, 0); std::vector<std::thread> th;
    for (int k = 0; k < T; ++k) th.emplace_back([&,
            k] { part[(size_t)k].reserve((size_t)((seed[(size_t)k + 1] - seed[(size_t)k]) / 16384 + 16)); ok[(size_t)k] = bgzf_walk_piece(z, n, seed[(size_t)k],
            seed[(size_t)k + 1], part[(size_t)k], ut[(size_t)k]); });
    for (auto &t : th) t.join();
    size_t total = 0;
    for (int k = 0; k < T; ++k) { if (!ok[(size_t)k]) return false; total += part[(size_t)k].size(); }
    blks.clear(); blks.reserve(total); utot = 0;
    for (int k = 0; k < T; ++k) { for (Blk b : part[(size_t)k]) { b.out_off += utot; blks.push_back(b); } utot += ut[(size_t)k]; }
    return true;
}
