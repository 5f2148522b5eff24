// cli_bam.h — BGZF / BAM input (host inflate, GPU-resident file with .bai ranges) and the BGZF writer of longphase_amd.
#pragma once
#include "cli_common.h"
#include "../csrc/lps_bgzf_walk.h"

// ------------------------------------------------------------------------------------------------ BGZF / BAM
struct Bgzf {
    // whole-file reader: mmap the file, locate the BGZF blocks (18-byte headers), inflate them with a thread pool into one
    // contiguous byte stream (not zero-initialised: every byte is written by exactly one inflate call)
    uint8_t *data = nullptr; size_t size = 0;
    ~Bgzf() { free(data); }
    void load(const std::string &path, int threads) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) die("ERROR: Cannot open bam file " + path);
        struct stat st; if (fstat(fd, &st) != 0) die("ERROR: Cannot stat " + path);
        const size_t fsz = (size_t)st.st_size;
        const uint8_t *raw = fsz ? (const uint8_t *)mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
        if (fsz && raw == (const uint8_t *)MAP_FAILED) die("ERROR: Cannot map " + path);
        if (fsz) madvise((void *)raw, fsz, MADV_SEQUENTIAL | MADV_WILLNEED);
        struct Blk { size_t off, clen, uoff, ulen; };
        std::vector<Blk> blks; size_t p = 0, utot = 0;
        while (p + 18 <= fsz) {
            if (raw[p] != 31 || raw[p + 1] != 139) die("ERROR: " + path + " is not a BGZF/BAM file");
            const unsigned xlen = raw[p + 10] | (raw[p + 11] << 8);
            size_t q = p + 12, bsize = 0;
            while (q + 4 <= p + 12 + xlen && q + 4 <= fsz) {                // BC subfield carries BSIZE
                const unsigned slen = raw[q + 2] | (raw[q + 3] << 8);
                if (raw[q] == 'B' && raw[q + 1] == 'C' && slen == 2) bsize = (raw[q + 4] | (raw[q + 5] << 8)) + 1;
                q += 4 + slen;
            }
            if (!bsize || bsize < 12 + xlen + 8 || p + bsize > fsz) die("ERROR: truncated BGZF block in " + path);
            const size_t isize = raw[p + bsize - 4] | (raw[p + bsize - 3] << 8) | (raw[p + bsize - 2] << 16) | ((size_t)raw[p + bsize - 1] << 24);
            blks.push_back({p + 12 + xlen, bsize - 12 - xlen - 8, utot, isize});
            utot += isize; p += bsize;
        }
        if (p != fsz || blks.empty()) die("ERROR: " + path + " is not a BGZF/BAM file");
        const size_t huge = 2u << 20, cap = (utot + 64 + huge - 1) / huge * huge;
        data = (uint8_t *)aligned_alloc(huge, cap); size = utot;
        if (!data) die("ERROR: out of memory inflating " + path);
        madvise(data, cap, MADV_HUGEPAGE);                              // 2 MiB pages: fewer faults while 16 threads fill it
        const int nt = std::max(1, threads);
        std::vector<std::thread> th; std::vector<int> bad(nt, 0); std::atomic<size_t> next{0};
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] {
            z_stream zs{}; if (inflateInit2(&zs, -15) != Z_OK) { bad[t] = 1; return; }
            for (;;) {
                const size_t b0 = next.fetch_add(16); if (b0 >= blks.size()) break;
                for (size_t b = b0; b < std::min(blks.size(), b0 + 16); ++b) {
                    if (!blks[b].ulen) continue;
                    inflateReset(&zs);
                    zs.next_in = const_cast<uint8_t *>(raw) + blks[b].off; zs.avail_in = (uInt)blks[b].clen;
                    zs.next_out = data + blks[b].uoff; zs.avail_out = (uInt)blks[b].ulen;
                    if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) bad[t] = 1;
                }
            }
            inflateEnd(&zs);
        });
        for (auto &x : th) x.join();
        if (fsz) munmap((void *)raw, fsz);
        close(fd);
        for (int x : bad) if (x) die("ERROR: inflate failed in " + path);
    }
};

static uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

// One BAM file, inflated, plus where every record of every wanted contig sits in it.  Nothing is decoded on the host except
// refID (to route the record) and the read name (to rank it); the rest is lps_push_bam_records' job on the GPU.
struct ContigRecords { std::vector<uint64_t> rec_off; uint64_t lo = 0, hi = 0; };   // offsets relative to `lo`
struct BamFile {
    Bgzf z; std::vector<std::string> ref_names; std::map<std::string, ContigRecords> contigs;
    void load(const std::string &path, int threads, const std::map<std::string, int> &want) {
        z.load(path, threads);
        const uint8_t *d = z.data; const size_t n = z.size;
        if (n < 12 || memcmp(d, "BAM\1", 4)) die("ERROR: " + path + " is not a BAM file");
        size_t p = 4; const uint32_t l_text = rd32(d + p); p += 4 + (size_t)l_text;
        if (p + 4 > n) die("ERROR: truncated BAM header in " + path);
        const uint32_t n_ref = rd32(d + p); p += 4;
        ref_names.resize(n_ref);
        for (uint32_t i = 0; i < n_ref; ++i) { if (p + 4 > n) die("ERROR: truncated BAM header in " + path);
            const uint32_t l = rd32(d + p);
            p += 4;
            if (!l || p + l + 4 > n) die("ERROR: truncated BAM header in " + path);
            ref_names[i] = std::string((const char *)d + p, l - 1);
            p += l + 4;
            }
        std::vector<ContigRecords *> dst(n_ref, nullptr);
        for (uint32_t i = 0; i < n_ref; ++i) if (want.count(ref_names[i])) dst[i] = &contigs[ref_names[i]];
        while (p + 4 <= n) {
            const uint32_t bs = rd32(d + p);
            if (bs < 32 || p + 4 + bs > n) die("ERROR: truncated BAM record in " + path);
            const int32_t tid = (int32_t)rd32(d + p + 4);
            if (tid >= 0 && tid < (int32_t)n_ref && dst[tid]) {
                ContigRecords &c = *dst[tid];
                if (c.rec_off.empty()) c.lo = p;
                c.rec_off.push_back(p + 4 - c.lo); c.hi = p + 4 + bs;
            }
            p += 4 + (size_t)bs;
        }
    }
    const char *name_of(const ContigRecords &c, size_t i, size_t &len) const { const uint8_t *r = z.data + c.lo + c.rec_off[i];
        len = r[8] ? r[8] - 1u : 0u;
        return (const char *)r + 32;
        }
};

// equal names <=> equal id, order = std::string operator< (the std::map<std::string,...> order of PhasingGraph.cpp:833,848)
static void rank_names(const std::vector<std::pair<const char *, size_t>> &names, std::vector<uint32_t> &id) {
    std::vector<uint32_t> idx(names.size()); std::iota(idx.begin(), idx.end(), 0u);
    auto less = [&](uint32_t a, uint32_t b) { const size_t m = std::min(names[a].second, names[b].second);
        const int c = memcmp(names[a].first, names[b].first, m);
        return c ? c < 0 : names[a].second < names[b].second;
        };
    std::sort(idx.begin(), idx.end(), less);
    id.resize(names.size()); uint32_t cur = 0;
    for (size_t k = 0; k < idx.size(); ++k) { if (k && (less(idx[k - 1], idx[k]) || less(idx[k], idx[k - 1]))) ++cur; id[idx[k]] = cur; }
}

// One BAM file inflated and indexed ON THE GPU (lps_bgzf_load + lps_bam_scan): the host only maps the compressed file, parses the BAM header
// and ranks the read names of each contig.
struct GpuBam {
    std::vector<std::string> ref_names;
    std::map<std::string, std::pair<int64_t, int64_t>> range;
    // whole-file mode: contig -> (first record, count)
    std::vector<uint8_t> header;                                      // inflated bytes "BAM\1" .. end of the reference table
    std::vector<std::pair<uint64_t, uint64_t>> voff; bool indexed = false;   // .bai: virtual-offset range of every contig's records
    const uint8_t *raw = nullptr; size_t fsz = 0; int fd = -1; std::string path;
    double t_map = 0, t_inflate = 0, t_scan = 0; int64_t total = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // map the file, inflate the BAM header on the host (a few blocks), read <bam>.bai when there is one
    void open_file(const std::string &p, bool use_index) {
        path = p; const double t0 = now();
        fd = open(p.c_str(), O_RDONLY);
        if (fd < 0) die("ERROR: Cannot open bam file " + p);
        struct stat st; if (fstat(fd, &st) != 0 || st.st_size < 28) die("ERROR: " + p + " is not a BGZF/BAM file");
        fsz = (size_t)st.st_size;
        raw = (const uint8_t *)mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0);
        if (raw == (const uint8_t *)MAP_FAILED) die("ERROR: Cannot map " + p);
        size_t q = 0; size_t need = 12;                                 // grows as l_text / names become known
        auto parsed = [&]() -> bool {
            if (header.size() < 12) return false;
            if (memcmp(header.data(), "BAM\1", 4)) die("ERROR: " + p + " is not a BAM file");
            size_t h = 8 + (size_t)rd32(header.data() + 4); if (h + 4 > header.size()) { need = h + 4; return false; }
            const uint32_t n_ref = rd32(header.data() + h); h += 4; ref_names.assign(n_ref, std::string());
            for (uint32_t i = 0; i < n_ref; ++i) {
                if (h + 4 > header.size()) { need = h + 4; return false; }
                const uint32_t l = rd32(header.data() + h); if (!l) die("ERROR: truncated BAM header in " + p);
                if (h + 4 + l + 4 > header.size()) { need = h + 4 + l + 4; return false; }
                ref_names[i] = std::string((const char *)header.data() + h + 4, l - 1); h += 4 + (size_t)l + 4;
            }
            header.resize(h); return true;
        };
        while (!parsed()) {
            if (q + 18 > fsz) die("ERROR: truncated BAM header in " + p);
            const unsigned xlen = raw[q + 10] | (raw[q + 11] << 8); const size_t bsize = (size_t)(raw[q + 16] | (raw[q + 17] << 8)) + 1;
            if (raw[q] != 31 || raw[q + 1] != 139 || q + bsize > fsz) die("ERROR: " + p + " is not a BGZF/BAM file");
            const size_t isize = rd32(raw + q + bsize - 4), at = header.size(); header.resize(at + isize);
            z_stream zs{};
            zs.next_in = const_cast<uint8_t *>(raw) + q + 12 + xlen;
            zs.avail_in = (uInt)(bsize - 12 - xlen - 8);
            zs.next_out = header.data() + at;
            zs.avail_out = (uInt)isize;
            if (inflateInit2(&zs, -15) != Z_OK || (isize && inflate(&zs, Z_FINISH) != Z_STREAM_END)) die("ERROR: inflate failed in " + p);
            inflateEnd(&zs); q += bsize; (void)need;
        }
        if (use_index) read_bai();
        t_map = now() - t0;
    }
    // BAI (SAM spec 5.2): per reference the bins with their chunk lists; the pseudo-bin 37450 holds (first, last) virtual offset of the reference's records
    void read_bai() {
        std::string cand[2] = {path + ".bai", path.size() > 4 ? path.substr(0, path.size() - 4) + ".bai" : std::string()};
        std::vector<uint8_t> b;
        for (const std::string &c : cand) { if (c.empty()) continue;
            std::ifstream f(c, std::ios::binary);
            if (!f) continue;
            b.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
            break;
            }
        if (b.size() < 8 || memcmp(b.data(), "BAI\1", 4)) return;
        auto r32 = [&](size_t &p) -> uint32_t { if (p + 4 > b.size()) die("ERROR: truncated index for " + path);
            const uint32_t v = rd32(b.data() + p);
            p += 4;
            return v;
            };
        auto r64 = [&](size_t &p) -> uint64_t { const uint64_t lo = r32(p), hi = r32(p); return lo | (hi << 32); };
        size_t p = 4; const uint32_t n_ref = r32(p);
        if (n_ref != ref_names.size()) die("ERROR: index and header of " + path + " disagree on the number of references");
        voff.assign(n_ref, {0, 0});
        for (uint32_t i = 0; i < n_ref; ++i) {
            uint64_t lo = ~0ull, hi = 0, mlo = 0, mhi = 0; bool meta = false;
            const uint32_t n_bin = r32(p);
            for (uint32_t k = 0; k < n_bin; ++k) {
                const uint32_t bin = r32(p), n_chunk = r32(p);
                for (uint32_t c = 0; c < n_chunk; ++c) { const uint64_t beg = r64(p), end = r64(p);
                    if (bin == 37450) { if (c == 0) { mlo = beg;
                            mhi = end;
                            meta = true;
                            } } else { lo = std::min(lo, beg);
                        hi = std::max(hi, end);
                        } }
            }
            const uint32_t n_intv = r32(p); p += 8ull * n_intv;
            if (meta) voff[i] = {mlo, mhi}; else if (hi) voff[i] = {lo, hi};
        }
        indexed = true;
    }
    void close_file() { if (whole.th.joinable()) whole.th.join(); if (ahead.th.joinable()) ahead.th.join(); if (raw) munmap((void *)raw, fsz); if (fd >= 0) close(fd); raw = nullptr; fd = -1; }

    // The header walk of the next load made AHEAD, on a helper thread: it needs the file and the library's host code, not the GPU - the first one runs
    // while the HIP runtime is still coming up (0.07 s of an 8 GB file's load), the next group's while the contigs of this one are phased.
    ~GpuBam() { if (whole.th.joinable()) whole.th.join(); if (ahead.th.joinable()) ahead.th.join(); }            // (a walk nobody took: e.g. --gpus N, where every worker opens the file itself)
    struct WalkAhead { uint64_t beg = 0, len = 0; lps_bgzf_block *blocks = nullptr; int64_t n = 0, inflated = 0; int rc = -1; std::thread th; bool pending = false; } ahead;
    // The WHOLE file's table, walked from the moment the file is open - host-only code of csrc/lps_bgzf_walk.h compiled into this program, so it needs
    // neither the GPU nor the library and runs beside the VCF parse and the HIP start-up; every later load takes its table as a slice of it.  For
    // files that are one contig group anyway (the caller decides): a walk touches one page per block, which a load of the same bytes reads anyway.
    struct WholeWalk { std::vector<lps_bgzf_block> blks; std::vector<uint64_t> ends; bool ok = false, started = false; std::thread th; } whole;
    void walk_whole_file() {
        if (whole.started || fd < 0 || getenv("LPS_CLI_NO_WALK_AHEAD")) return;
        whole.started = true;
        whole.th = std::thread([this] {
            const ZSource z{nullptr, fd, 0}; uint64_t ut = 0;
            bool ok = bgzf_walk_parallel(z, (uint64_t)fsz, whole.blks, ut);
            if (!ok) { whole.blks.clear(); ut = 0; ok = bgzf_walk_piece(z, (uint64_t)fsz, 0, (uint64_t)fsz, whole.blks, ut) && !whole.blks.empty(); }
            if (ok) { whole.ends.resize(whole.blks.size()); for (size_t k = 0; k < whole.blks.size(); ++k) whole.ends[k] = whole.blks[k].in_off + whole.blks[k].in_len + 8; }
            whole.ok = ok;
        });
    }
    // the blocks of bytes [beg, beg + len) out of the whole file's table, offsets relative to the span; false: the span does not begin and end on blocks of it
    bool slice_whole(uint64_t beg, uint64_t len, std::vector<lps_bgzf_block> &out) {
        if (!whole.started) return false;
        if (whole.th.joinable()) whole.th.join();
        if (!whole.ok || !len) return false;
        size_t k0 = 0;
        if (beg) { auto it = std::lower_bound(whole.ends.begin(), whole.ends.end(), beg); if (it == whole.ends.end() || *it != beg) return false; k0 = (size_t)(it - whole.ends.begin()) + 1; }
        auto it9 = std::lower_bound(whole.ends.begin(), whole.ends.end(), beg + len);
        if (it9 == whole.ends.end() || *it9 != beg + len) return false;
        const size_t k9 = (size_t)(it9 - whole.ends.begin());
        if (k9 < k0) return false;
        out.assign(whole.blks.begin() + (ptrdiff_t)k0, whole.blks.begin() + (ptrdiff_t)k9 + 1);
        const uint64_t o0 = out.front().out_off;
        for (lps_bgzf_block &b : out) { b.in_off -= beg; b.out_off -= o0; }
        return true;
    }
    void walk_ahead(Lps &L, uint64_t beg, uint64_t len) {
        drop_walk(L);
        if (whole.started) return;                                      // (the whole file's table is being made: load_span slices it)
        if (getenv("LPS_CLI_NO_WALK_AHEAD")) return;                    // (A/B switch)
        ahead.beg = beg; ahead.len = len; ahead.pending = true; ahead.rc = -1;
        WalkAhead *a = &ahead; Lps *lib = &L; const int f = fd;
        ahead.th = std::thread([a, lib, f] {
            for (int spin = 0; !lib->ready.load(std::memory_order_acquire) && spin < 20000; ++spin) usleep(500);   // (the library is being loaded by the thread that creates the context)
            if (lib->ready.load(std::memory_order_acquire)) a->rc = lib->bgzf_walk_fd(f, (int64_t)a->beg, (int64_t)a->len, &a->blocks, &a->n, &a->inflated);
        });
    }
    void drop_walk(Lps &L) { if (ahead.th.joinable()) ahead.th.join(); if (ahead.blocks) L.bgzf_blocks_free(ahead.blocks); ahead.blocks = nullptr; ahead.pending = false; }
    // bytes [beg, beg + len) of the file onto the GPU, with the table walked ahead when it is the one for these bytes
    void load_span(Lps &L, lps_ctx *ctx, uint64_t beg, uint64_t len) {
        const double tj = now();
        std::vector<lps_bgzf_block> part;
        if (slice_whole(beg, len, part)) {
            if (getenv("LPS_CLI_DEBUG")) fprintf(stderr, "[cli] waited %.3f s for the whole file's header walk; %zu of %zu blocks in this span\n", now() - tj, part.size(), whole.blks.size());
            if (L.bgzf_load_fd_blocks(ctx, fd, (int64_t)beg, (int64_t)len, part.data(), (int64_t)part.size(), &total)) die(std::string("ERROR: ") + path + ": " + L.last_error(ctx));
            drop_walk(L);
            return;
        }
        if (ahead.pending && ahead.th.joinable()) ahead.th.join();
        if (getenv("LPS_CLI_DEBUG")) fprintf(stderr, "[cli] waited %.3f s for the header walk made ahead\n", now() - tj);
        const bool have = ahead.pending && ahead.rc == 0 && ahead.beg == beg && ahead.len == len;
        const int rc = have ? L.bgzf_load_fd_blocks(ctx, fd, (int64_t)beg, (int64_t)len, ahead.blocks, ahead.n, &total) : L.bgzf_load_fd(ctx, fd, (int64_t)beg, (int64_t)len, &total);
        drop_walk(L);
        if (rc) die(std::string("ERROR: ") + path + ": " + L.last_error(ctx));
    }
    // the compressed span of a group of consecutive contigs (first block of the first, last block of the last) as the index gives it
    void group_span(const std::vector<std::string> &chrs, uint64_t &cbeg, uint64_t &stop, uint64_t &ubeg, uint64_t &uend, uint64_t &last_isize) const {
        const size_t t0 = (size_t)tid_of(chrs.front()), t9 = (size_t)tid_of(chrs.back());
        cbeg = voff[t0].first >> 16; ubeg = voff[t0].first & 0xffff; const uint64_t cend = voff[t9].second >> 16; uend = voff[t9].second & 0xffff;
        stop = cend; last_isize = 0;
        if (uend) { if (cend + 18 > fsz) die("ERROR: index of " + path + " points past the end of the file");
            const uint64_t bsize = (uint64_t)(raw[cend + 16] | (raw[cend + 17] << 8)) + 1;
            stop = cend + bsize;
            if (stop > fsz) die("ERROR: truncated BGZF block in " + path);
            last_isize = rd32(raw + stop - 4);
            }
        if (cbeg >= stop || stop > fsz) die("ERROR: index of " + path + " is inconsistent");
    }
    void walk_group_ahead(Lps &L, const std::vector<std::string> &chrs) { if (chrs.empty()) return; uint64_t cbeg, stop, ubeg, uend, li; group_span(chrs, cbeg, stop, ubeg, uend, li); walk_ahead(L, cbeg, stop - cbeg); }
    // whole-file mode: everything resident at once, one contiguous record range per contig
    void load_all(Lps &L, lps_ctx *ctx) {
        const double t1 = now();
        posix_fadvise(fd, 0, (off_t)fsz, POSIX_FADV_WILLNEED);
        load_span(L, ctx, 0, fsz);                                     // (pread into the upload pieces: the mapping stays a few header pages)
        const double t2 = now(); t_inflate += t2 - t1;
        int64_t n = 0;
        if (L.bam_scan(ctx, (int64_t)header.size(), (int32_t)ref_names.size(), &n)) die(std::string("ERROR: ") + path + ": " + L.last_error(ctx));
        std::vector<int32_t> tid((size_t)n);
        if (n && L.bam_record_tids(ctx, tid.data())) die(std::string("ERROR: ") + L.last_error(ctx));
        for (int64_t i = 0; i < n;) {                                  // a coordinate-sorted BAM holds every contig as ONE run of records
            int64_t j = i; while (j < n && tid[(size_t)j] == tid[(size_t)i]) ++j;
            if (tid[(size_t)i] >= 0) { const std::string &nm = ref_names[(size_t)tid[(size_t)i]];
                if (range.count(nm)) die("ERROR: " + path + " is not coordinate-sorted");
                range[nm] = {i, j - i};
                }
            i = j;
        }
        t_scan += now() - t2;
    }
    // indexed mode.  Consecutive contigs are taken in GROUPS of up to `budget` compressed bytes: one upload + one inflate launch per group (a launch
    // over a single small contig cannot fill the GPU: the inflate kernel's latency is that of one 64 KiB block however few blocks there are).
    int tid_of(const std::string &chr) const { for (size_t t = 0; t < ref_names.size(); ++t) if (ref_names[t] == chr) return (int)t; return -1; }
    std::vector<std::vector<std::string>> plan_groups(const std::vector<std::string> &chrs, uint64_t budget) const {
        std::vector<std::vector<std::string>> groups; int last_tid = -2; uint64_t bytes = 0;
        for (const std::string &c : chrs) {
            const int t = tid_of(c); if (t < 0 || voff[(size_t)t].second <= voff[(size_t)t].first) continue;       // not in this BAM / no records
            const uint64_t sz = (voff[(size_t)t].second >> 16) - (voff[(size_t)t].first >> 16) + 65536;
            bool gap_free = t > last_tid && !groups.empty();
            if (gap_free) for (int k = last_tid + 1; k < t; ++k) if (voff[(size_t)k].second > voff[(size_t)k].first) gap_free = false;
            // a contig in between is not wanted: keep groups tight
            if (!gap_free || bytes + sz > budget) { groups.emplace_back(); bytes = 0; }
            groups.back().push_back(c); bytes += sz; last_tid = t;
        }
        return groups;
    }
    // upload + inflate + scan the records of a group of consecutive contigs; fills `range` for its members
    void load_group(Lps &L, lps_ctx *ctx, const std::vector<std::string> &chrs) {
        range.clear();
        if (chrs.empty()) return;
        const size_t t0 = (size_t)tid_of(chrs.front()), t9 = (size_t)tid_of(chrs.back());
        const double t1 = now();
        uint64_t cbeg, stop, ubeg, uend, last_isize;
        group_span(chrs, cbeg, stop, ubeg, uend, last_isize);
        load_span(L, ctx, cbeg, stop - cbeg);
        const int64_t end = uend ? total - (int64_t)last_isize + (int64_t)uend : total;
        const double t2 = now(); t_inflate += t2 - t1;
        int64_t n = 0;
        if (L.bam_scan_range(ctx, (int64_t)ubeg, end, (int32_t)ref_names.size(), &n)) die(std::string("ERROR: ") + path + ": " + L.last_error(ctx));
        std::vector<int32_t> tid((size_t)n);
        if (n && L.bam_record_tids(ctx, tid.data())) die(std::string("ERROR: ") + L.last_error(ctx));
        for (int64_t i = 0; i < n;) {
            int64_t j = i; while (j < n && tid[(size_t)j] == tid[(size_t)i]) ++j;
            if (tid[(size_t)i] < (int32_t)t0 || tid[(size_t)i] > (int32_t)t9) die("ERROR: index of " + path + " does not match its records");
            const std::string &nm = ref_names[(size_t)tid[(size_t)i]];
            if (range.count(nm)) die("ERROR: " + path + " is not coordinate-sorted");
            range[nm] = {i, j - i};
            i = j;
        }
        t_scan += now() - t2;
    }
    // names of records [first, first+count) -> (pointer, length) pairs into `store`
    void names(Lps &L, lps_ctx *ctx, int64_t first, int64_t count, std::vector<char> &store, std::vector<uint32_t> &off, std::vector<std::pair<const char *, size_t>> &out) {
        int64_t nb = 0;
        if (L.bam_names(ctx, first, count, nullptr, nullptr, 0, &nb)) die(std::string("ERROR: ") + L.last_error(ctx));
        store.resize((size_t)nb + 1); off.resize((size_t)count + 1);
        if (L.bam_names(ctx, first, count, off.data(), store.data(), (int64_t)store.size(), &nb)) die(std::string("ERROR: ") + L.last_error(ctx));
        for (int64_t i = 0; i < count; ++i) out.emplace_back(store.data() + off[(size_t)i], (size_t)(off[(size_t)i + 1] - off[(size_t)i]) - 1);
    }
};

// A whole BAM inflated and scanned on the GPU, then copied back ONCE: `out` as BamFile::load leaves it (the records of the wanted contigs; offsets
// relative to the contig's first record) for callers that splice records on the host.  The stream stays resident in `ctx` until its next lps_bgzf_load.
static void gpu_load_to_host(Lps &L, lps_ctx *ctx, const std::string &path, const std::map<std::string, int> &want, int threads, BamFile &out, double *t_inflate = nullptr) {
    GpuBam gb; gb.open_file(path, false); gb.load_all(L, ctx);
    const size_t huge = 2u << 20, cap = ((size_t)gb.total + 64 + huge - 1) / huge * huge;
    free(out.z.data);
    out.z.data = (uint8_t *)aligned_alloc(huge, cap); out.z.size = (size_t)gb.total;
    if (!out.z.data) die("ERROR: out of memory inflating " + path);
    madvise(out.z.data, cap, MADV_HUGEPAGE);
    {   // touched by a few threads first, so that the device-to-host copy does not fault page by page
        std::vector<std::thread> th; const int nt = std::max(1, std::min(threads, 8)); const size_t slice = (cap + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { const size_t a = std::min(cap, slice * t), e = std::min(cap, a + slice); for (size_t q = a; q < e; q += 4096) out.z.data[q] = 0; });
        for (auto &x : th) x.join();
    }
    if (gb.total && L.bgzf_read(ctx, 0, gb.total, out.z.data)) die(std::string("ERROR: ") + path + ": " + L.last_error(ctx));
    out.ref_names = gb.ref_names;
    for (auto &kv : gb.range) {
        if (!want.count(kv.first)) continue;
        ContigRecords &c = out.contigs[kv.first]; c.rec_off.resize((size_t)kv.second.second);
        if (kv.second.second && L.bam_record_offsets(ctx, kv.second.first, kv.second.second, c.rec_off.data())) die(std::string("ERROR: ") + path + ": " + L.last_error(ctx));
        if (c.rec_off.empty()) continue;
        c.lo = c.rec_off.front() - 4; c.hi = c.rec_off.back() + rd32(out.z.data + c.rec_off.back() - 4);      // a contig's records are one run of the stream
        if (c.hi > (uint64_t)gb.total) die("ERROR: truncated BAM record in " + path);
        for (uint64_t &o : c.rec_off) o -= c.lo;
    }
    if (t_inflate) *t_inflate += gb.t_inflate + gb.t_scan;
    gb.close_file();
}


struct BgzfWriter {
    FILE *f = nullptr;
    int threads = 1, level = 6, strategy = Z_RLE;
    unsigned long long bytes_out = 0;
    std::vector<uint8_t> pend;
    // pend: < one block of bytes not yet written
    static constexpr size_t B = 0xff00;
    void open(const std::string &path, int t, int lvl, int strat) { f = fopen(path.c_str(), "wb");
        if (!f) die("Fail to open write file: " + path);
        threads = std::max(1, t);
        level = lvl;
        strategy = strat;
        }
    // deflate the blocks of B bytes (the last one may be shorter) starting at p and write them in order; batches of 1024 blocks, the finished
    // batch is written by a helper thread while the pool deflates the next one
    std::thread writer; std::vector<std::vector<uint8_t>> inflight; std::atomic<int> write_bad{0};
    void wait_writer() { if (writer.joinable()) writer.join(); if (write_bad) die("ERROR: write output bam file failed"); }
    void emit(const uint8_t *p, size_t n) {
        const size_t total_blk = (n + B - 1) / B, batch = 1024;
        for (size_t b0 = 0; b0 < total_blk; b0 += batch) {
            const size_t nblk = std::min(batch, total_blk - b0); const uint8_t *q = p + b0 * B; const size_t qn = std::min(n - b0 * B, nblk * B);
            std::vector<std::vector<uint8_t>> out(nblk); std::atomic<size_t> next{0}; std::vector<std::thread> th; std::atomic<int> bad{0};
            auto work = [&] {
                z_stream zs{}; if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) { bad = 1; return; }
                for (;;) { const size_t b = next.fetch_add(1); if (b >= nblk) break;
                    const size_t off = b * B, len = std::min(B, qn - off);
                    std::vector<uint8_t> &o = out[b]; o.resize(18 + deflateBound(&zs, (uLong)len) + 8);
                    deflateReset(&zs);
                    zs.next_in = const_cast<uint8_t *>(q) + off;
                    zs.avail_in = (uInt)len;
                    zs.next_out = o.data() + 18;
                    zs.avail_out = (uInt)(o.size() - 26);
                    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { bad = 1; break; }
                    const size_t clen = zs.total_out, bsize = 18 + clen + 8;
                    if (bsize > 65536) { bad = 1; break; }
                    const uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)((bsize - 1) & 255), (uint8_t)((bsize - 1) >> 8)};
                    memcpy(o.data(), hdr, 18);
                    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), q + off, (uInt)len), isz = (uint32_t)len;
                    for (int k = 0; k < 4; ++k) { o[18 + clen + k] = (uint8_t)(crc >> (8 * k)); o[22 + clen + k] = (uint8_t)(isz >> (8 * k)); }
                    o.resize(bsize);
                }
                deflateEnd(&zs);
            };
            const int nt = (int)std::min<size_t>(threads, nblk);
            for (int t = 1; t < nt; ++t) th.emplace_back(work);
            work();
            for (auto &x : th) x.join();
            if (bad) die("ERROR: deflate failed");
            wait_writer();
            inflight.swap(out);
            for (auto &o : inflight) bytes_out += o.size();
            writer = std::thread([this] { for (auto &o : inflight) if (fwrite(o.data(), 1, o.size(), f) != o.size()) { write_bad = 1; break; } });
        }
    }
    void append(const uint8_t *p, size_t n) {
        if (!pend.empty()) {                                            // top up the open block first
            const size_t k = std::min(n, B - pend.size()); pend.insert(pend.end(), p, p + k); p += k; n -= k;
            if (pend.size() < B) return;
            emit(pend.data(), B); pend.clear();
        }
        const size_t whole = n / B * B;
        emit(p, whole);
        pend.assign(p + whole, p + n);
    }
    void flush_partial() { emit(pend.data(), pend.size()); pend.clear(); wait_writer(); }
    void write_raw(const uint8_t *p, size_t n) { wait_writer();
        if (n && fwrite(p, 1, n, f) != n) die("ERROR: write output bam file failed");
        bytes_out += n;
        }
    void finish() {
        emit(pend.data(), pend.size()); pend.clear(); wait_writer();
        static const uint8_t eof[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (fwrite(eof, 1, 28, f) != 28 || fclose(f) != 0) die("ERROR: write output bam file failed");
        f = nullptr;
    }
};

// byte length of one aux field starting at p (tag[2] type value), 0 when malformed
static size_t aux_field_len(const uint8_t *p, const uint8_t *end) {
    if (p + 3 > end) return 0;
    const uint8_t t = p[2]; size_t v = 0;
    switch (t) {
        case 'A': case 'c': case 'C': v = 1;
        break;
        case 's': case 'S': v = 2;
        break;
        case 'i': case 'I': case 'f': v = 4;
        break;
        case 'd': v = 8;
        break;
        case 'Z': case 'H': { const uint8_t *q = p + 3; while (q < end && *q) ++q; if (q >= end) return 0; v = (size_t)(q - (p + 3)) + 1; break; }
        case 'B': { if (p + 8 > end) return 0;
            const uint8_t st = p[3];
            const size_t cnt = rd32(p + 4);
            size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
            if (!es) return 0;
            v = 5 + cnt * es;
            break;
            }
        default: return 0;
    }
    return p + 3 + v <= end ? 3 + v : 0;
}
