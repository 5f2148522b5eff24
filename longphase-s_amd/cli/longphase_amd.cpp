// longphase_amd — host CLI over liblps_hip.so: `longphase_amd phase ...` with the reference's flags and output format.
//
// SURVEY.md §8(f) widening (rows f-1 and f-4): BGZF/BAM decoding and the phased-VCF rewriter, written from scratch on zlib
// (htslib is not available on the GPU box).  It restates, citing the reference (relative to /root/reference/):
//   option surface of `phase`                 src/phase/Phasing.cpp:9-116
//   SnpParser row selection                   src/phase/ParsingBam.cpp:222-359
//   per-chromosome driver                     src/phase/PhasingProcess.cpp:113-173   (the hot path is one lps_phase_chromosome call)
//   SnpParser::writeLine (VCF rewrite rules)  src/phase/ParsingBam.cpp:460-635
// Not supported (the reference path must be used): --sv-file, --mod-file, --dot, --deepsomatic_output, CRAM, CIGARs in CG tags.
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <map>
#include <numeric>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lps_abi.h"

static const char *kVersion = "1.0.0-mi355x";

[[noreturn]] static void die(const std::string &m) { std::cerr << m << "\n"; exit(1); }

// ------------------------------------------------------------------------------------------------ BGZF / BAM
struct Bgzf {
    // whole-file reader: locate the BGZF blocks, inflate them with a thread pool, expose one contiguous byte stream
    std::vector<uint8_t> data;
    void load(const std::string &path, int threads) {
        std::ifstream f(path, std::ios::binary);
        if (!f) die("ERROR: Cannot open bam file " + path);
        std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        struct Blk { size_t off, clen, uoff, ulen; };
        std::vector<Blk> blks; size_t p = 0, utot = 0;
        while (p + 18 <= raw.size()) {
            if (raw[p] != 31 || raw[p + 1] != 139) die("ERROR: " + path + " is not a BGZF/BAM file");
            const unsigned xlen = raw[p + 10] | (raw[p + 11] << 8);
            size_t q = p + 12, bsize = 0;
            while (q + 4 <= p + 12 + xlen) {                             // BC subfield carries BSIZE
                const unsigned slen = raw[q + 2] | (raw[q + 3] << 8);
                if (raw[q] == 'B' && raw[q + 1] == 'C' && slen == 2) bsize = (raw[q + 4] | (raw[q + 5] << 8)) + 1;
                q += 4 + slen;
            }
            if (!bsize || p + bsize > raw.size()) die("ERROR: truncated BGZF block in " + path);
            const size_t isize = raw[p + bsize - 4] | (raw[p + bsize - 3] << 8) | (raw[p + bsize - 2] << 16) | ((size_t)raw[p + bsize - 1] << 24);
            blks.push_back({p + 12 + xlen, bsize - 12 - xlen - 8, utot, isize});
            utot += isize; p += bsize;
        }
        data.resize(utot);
        const int nt = std::max(1, threads);
        std::vector<std::thread> th; std::vector<int> bad(nt, 0);
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] {
            for (size_t b = t; b < blks.size(); b += nt) {
                if (!blks[b].ulen) continue;
                z_stream zs{}; zs.next_in = raw.data() + blks[b].off; zs.avail_in = (uInt)blks[b].clen;
                zs.next_out = data.data() + blks[b].uoff; zs.avail_out = (uInt)blks[b].ulen;
                if (inflateInit2(&zs, -15) != Z_OK || inflate(&zs, Z_FINISH) != Z_STREAM_END) bad[t] = 1;
                inflateEnd(&zs);
            }
        });
        for (auto &x : th) x.join();
        for (int x : bad) if (x) die("ERROR: inflate failed in " + path);
    }
};

struct ReadPack {   // SoA of one chromosome, laid out as lps_read_batch wants it
    std::vector<int32_t> ref_start, l_qseq; std::vector<uint16_t> flag; std::vector<uint8_t> mapq; std::vector<uint32_t> name_id;
    std::vector<uint64_t> cigar_off{0}, seq_off{0}, qual_off{0}; std::vector<uint32_t> cigar; std::vector<uint8_t> seq, qual;
    std::vector<std::string> names;
    void assign_name_ids() {                                         // equal names <=> equal id, order = std::string operator<
        std::vector<uint32_t> idx(names.size()); std::iota(idx.begin(), idx.end(), 0u);
        std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return names[a] < names[b]; });
        name_id.resize(names.size()); uint32_t id = 0;
        for (size_t k = 0; k < idx.size(); ++k) { if (k && names[idx[k]] != names[idx[k - 1]]) ++id; name_id[idx[k]] = id; }
    }
    lps_read_batch view() const {
        return lps_read_batch{(int64_t)ref_start.size(), ref_start.data(), flag.data(), mapq.data(), l_qseq.data(), name_id.data(),
                              cigar_off.data(), cigar.data(), seq_off.data(), seq.data(), qual_off.data(), qual.data()};
    }
};

static uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

// decode every record of a BAM into per-contig packs (only contigs in `want`)
static void read_bam(const std::string &path, int threads, const std::map<std::string, int> &want, std::map<std::string, ReadPack> &packs) {
    Bgzf z; z.load(path, threads);
    const uint8_t *d = z.data.data(); const size_t n = z.data.size();
    if (n < 12 || memcmp(d, "BAM\1", 4)) die("ERROR: " + path + " is not a BAM file");
    size_t p = 4; const uint32_t l_text = rd32(d + p); p += 4 + l_text;
    const uint32_t n_ref = rd32(d + p); p += 4;
    std::vector<std::string> ref_names(n_ref);
    for (uint32_t i = 0; i < n_ref; ++i) { const uint32_t l = rd32(d + p); p += 4; ref_names[i] = std::string((const char *)d + p, l - 1); p += l + 4; }
    std::vector<ReadPack *> dst(n_ref, nullptr);
    for (uint32_t i = 0; i < n_ref; ++i) if (want.count(ref_names[i])) dst[i] = &packs[ref_names[i]];
    while (p + 4 <= n) {
        const uint32_t bs = rd32(d + p); const uint8_t *r = d + p + 4; p += 4 + bs;
        if (p > n) die("ERROR: truncated BAM record in " + path);
        const int32_t tid = (int32_t)rd32(r), pos = (int32_t)rd32(r + 4);
        const uint32_t l_name = r[8], mq = r[9], n_cig = r[12] | (r[13] << 8), fl = r[14] | (r[15] << 8), l_seq = rd32(r + 16);
        if (tid < 0 || tid >= (int32_t)n_ref || !dst[tid]) continue;
        ReadPack &k = *dst[tid];
        const uint8_t *q = r + 32;
        k.names.emplace_back((const char *)q, l_name ? l_name - 1 : 0); q += l_name;
        k.ref_start.push_back(pos); k.flag.push_back((uint16_t)fl); k.mapq.push_back((uint8_t)mq); k.l_qseq.push_back((int32_t)l_seq);
        const size_t c0 = k.cigar.size(); k.cigar.resize(c0 + n_cig); memcpy(k.cigar.data() + c0, q, 4ull * n_cig); q += 4ull * n_cig; k.cigar_off.push_back(k.cigar.size());
        k.seq.insert(k.seq.end(), q, q + (l_seq + 1) / 2); q += (l_seq + 1) / 2; k.seq_off.push_back(k.seq.size());
        k.qual.insert(k.qual.end(), q, q + l_seq); k.qual_off.push_back(k.qual.size());
    }
}

// ------------------------------------------------------------------------------------------------ text inputs
static bool read_lines(const std::string &path, std::vector<std::string> &lines) {   // plain or gzip text
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    std::string cur; char buf[1 << 16]; int k;
    while ((k = gzread(f, buf, sizeof buf)) > 0) {
        for (int i = 0; i < k; ++i) { if (buf[i] == '\n') { lines.push_back(cur); cur.clear(); } else cur.push_back(buf[i]); }
    }
    if (!cur.empty()) lines.push_back(cur);
    gzclose(f);
    return true;
}

struct ChrVariants { std::map<int32_t, std::pair<std::string, std::string>> rows; std::vector<int32_t> pos; std::vector<std::string> ref, alt; };

static std::vector<std::string> split_tab(const std::string &s) {
    std::vector<std::string> f; size_t a = 0;
    while (true) { size_t b = s.find('\t', a); if (b == std::string::npos) { f.push_back(s.substr(a)); break; } f.push_back(s.substr(a, b - a)); a = b + 1; }
    return f;
}

// SnpParser::SnpParser (ParsingBam.cpp:222-359): het bi-allelic SNPs (bcf_is_snp: every allele one base), with --indels every other
// het bi-allelic record.  GT of the first sample must be 0/1, 1/0, 0|1 or 1|0.
static void parse_vcf(const std::vector<std::string> &lines, bool indels, std::vector<std::string> &chr_order, std::map<std::string, ChrVariants> &out) {
    for (const std::string &ln : lines) {
        if (ln.empty()) continue;
        if (ln[0] == '#') {
            if (ln.compare(0, 13, "##contig=<ID=") == 0) { size_t e = ln.find_first_of(",>", 13); std::string c = ln.substr(13, e - 13); if (!out.count(c)) { out[c]; chr_order.push_back(c); } }
            continue;
        }
        std::vector<std::string> f = split_tab(ln);
        if (f.size() < 10) continue;
        const std::string &ref = f[3], &alt = f[4];
        if (alt.find(',') != std::string::npos || alt.empty() || alt[0] == '<' || alt == "." || alt == "*") continue;
        const bool is_snp = ref.size() == 1 && alt.size() == 1;
        if (!is_snp && !indels) continue;
        // GT position inside FORMAT
        std::vector<std::string> fmt, smp; { std::stringstream a(f[8]), b(f[9]); std::string x; while (std::getline(a, x, ':')) fmt.push_back(x); while (std::getline(b, x, ':')) smp.push_back(x); }
        size_t gi = std::find(fmt.begin(), fmt.end(), "GT") - fmt.begin();
        if (gi >= fmt.size() || gi >= smp.size()) die("pos " + f[1] + " missing GT value");
        const std::string &gt = smp[gi];
        if (!(gt == "0/1" || gt == "1/0" || gt == "0|1" || gt == "1|0")) continue;
        if (!out.count(f[0])) { out[f[0]]; chr_order.push_back(f[0]); }
        out[f[0]].rows[std::stoi(f[1]) - 1] = {ref, alt};           // map semantics: the later record at one position wins
    }
    for (auto &kv : out) for (auto &r : kv.second.rows) { kv.second.pos.push_back(r.first); kv.second.ref.push_back(r.second.first); kv.second.alt.push_back(r.second.second); }
}

static void read_fasta(const std::string &path, const std::map<std::string, ChrVariants> &want, std::map<std::string, std::string> &seqs) {
    std::ifstream f(path); if (!f) die("ERROR: Cannot open reference " + path);
    std::string ln, cur; std::string *dst = nullptr;
    while (std::getline(f, ln)) {
        if (!ln.empty() && ln[0] == '>') { std::string name = ln.substr(1, ln.find_first_of(" \t", 1) - 1); dst = want.count(name) ? &seqs[name] : nullptr; continue; }
        if (dst) { if (!ln.empty() && ln.back() == '\r') ln.pop_back(); dst->append(ln); }
    }
}

// ------------------------------------------------------------------------------------------------ VCF rewriter
struct Phased { int32_t ps; char a, b; };
// SnpParser::writeLine (ParsingBam.cpp:460-635) restated
static void write_vcf(const std::vector<std::string> &lines, const std::string &out_path, const std::map<std::string, std::map<int32_t, Phased>> &res,
                      const std::map<std::string, ChrVariants> &vars, const std::string &command) {
    std::ofstream o(out_path); if (!o) die("Fail to open write file: " + out_path);
    bool ps_def = false, cmd_done = false;
    for (const std::string &in : lines) {
        if (in.compare(0, 2, "##") == 0) { if (in.compare(0, 16, "##FORMAT=<ID=PS,") == 0) ps_def = true; o << in << "\n"; continue; }
        if (in.compare(0, 6, "#CHROM") == 0 || in.compare(0, 6, "#chrom") == 0) {
            if (!cmd_done) {
                if (!ps_def) { o << "##FORMAT=<ID=PS,Number=1,Type=Integer,Description=\"Phase set identifier\">\n"; ps_def = true; }
                o << "##longphaseVersion=" << kVersion << "\n" << "##commandline=\"" << command << "\"\n"; cmd_done = true;
            }
            o << in << "\n"; continue;
        }
        std::istringstream iss(in);
        std::vector<std::string> f((std::istream_iterator<std::string>(iss)), std::istream_iterator<std::string>());
        if (f.empty()) continue;
        if (f.size() < 10) { o << in << "\n"; continue; }
        const int32_t pidx = std::stoi(f[1]) - 1;
        auto colon_index = [](const std::string &fmt, size_t upto) { int c = 0; for (size_t i = 0; i < upto; ++i) if (fmt[i] == ':') ++c; return c; };
        auto value_start = [](const std::string &v, int colons) { int cur = 0; size_t st = 0; for (size_t i = 0; i < v.size(); ++i) { if (cur >= colons) break; if (v[i] == ':') ++cur; ++st; } return st; };
        if (f[8].find("PS") != std::string::npos) {                  // strip an existing PS key and value
            const size_t pp = f[8].find("PS"); const int cp = colon_index(f[8], pp);
            if (f[8].find(":", pp + 1) != std::string::npos) f[8].erase(pp, 3); else f[8].erase(pp - 1, 3);
            const size_t st = value_start(f[9], cp);
            if (f[9].find(":", st + 1) != std::string::npos) { const size_t e = f[9].find(":", st + 1); f[9].erase(st, e - st + 1); }
            else f[9].erase(st - 1, f[9].length() - st + 1);
        }
        if (f[8].find("GT") != std::string::npos) {                  // un-phase an existing phased GT
            const size_t gp = f[8].find("GT"); const size_t st = value_start(f[9], colon_index(f[8], gp));
            if (st + 2 < f[9].size() + 1 && f[9][st + 1] == '|') {
                if (f[9][st] > f[9][st + 2]) { f[9][st + 1] = f[9][st]; f[9][st] = f[9][st + 2]; f[9][st + 2] = f[9][st + 1]; }
                f[9][st + 1] = '/';
            }
        }
        const Phased *ph = nullptr;
        auto rc = res.find(f[0]);
        if (rc != res.end()) { auto it = rc->second.find(pidx); if (it != rc->second.end()) ph = &it->second; }
        bool extracted = false;
        auto vc = vars.find(f[0]);
        if (vc != vars.end()) extracted = std::binary_search(vc->second.pos.begin(), vc->second.pos.end(), pidx);
        if (ph && extracted) {
            f[8] += ":PS"; f[9] += ":" + std::to_string(ph->ps);
            const size_t gp = f[8].find("GT"); const size_t st = value_start(f[9], colon_index(f[8], gp));
            f[9][st] = ph->a; f[9][st + 1] = '|'; f[9][st + 2] = ph->b;
        } else { f[8] += ":PS"; f[9] += ":."; }
        for (size_t i = 0; i < f.size(); ++i) { if (i) o << "\t"; o << f[i]; }
        o << "\n";
    }
}

// ------------------------------------------------------------------------------------------------ phase
static const char *kUsage =
    "Usage: longphase_amd phase [OPTION] ... READSFILE\n"
    "   -s, --snp-file=NAME   -b, --bam-file=NAME (repeatable)   -r, --reference=NAME   -o, --out-prefix=NAME (result)   -t, --threads=Num (1)\n"
    "   --ont | --pb   --indels   -q MAPQ(1)  -p baseQuality(12)  -e edgeWeight(0.1)  -a connectAdjacent(35)  -d distance(300000)\n"
    "   -1 edgeThreshold(0.7)  -L overlapThreshold(0.2)  -m readConfidence(0.65)  -n snpConfidence(0.75)  --gpu=ID (0)\n";

static int phase_main(int argc, char **argv, const std::string &command) {
    lps_params P; lps_default_params(&P);
    std::string snp, ref, prefix = "result"; std::vector<std::string> bams; int threads = 1, gpu = 0; bool ont = false, pb = false;
    auto need = [&](int &i) -> std::string { if (i + 1 >= argc) { std::cerr << kUsage; exit(1); } return argv[++i]; };
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i], v; size_t eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); }
        auto val = [&]() { return v.empty() ? need(i) : v; };
        if (a == "-s" || a == "--snp-file") snp = val();
        else if (a == "-b" || a == "--bam-file") bams.push_back(val());
        else if (a == "-r" || a == "--reference") ref = val();
        else if (a == "-o" || a == "--out-prefix") prefix = val();
        else if (a == "-t" || a == "--threads") threads = std::stoi(val());
        else if (a == "--ont") ont = true; else if (a == "--pb") pb = true;
        else if (a == "--indels") P.phase_indel = 1;
        else if (a == "-q" || a == "--mappingQuality") P.mapping_quality = std::stoi(val());
        else if (a == "-p" || a == "--baseQuality") P.base_quality = std::stoi(val());
        else if (a == "-e" || a == "--edgeWeight") P.edge_weight = std::stod(val());
        else if (a == "-a" || a == "--connectAdjacent") P.connect_adjacent = std::stoi(val());
        else if (a == "-d" || a == "--distance") P.distance = std::stoi(val());
        else if (a == "-1" || a == "--edgeThreshold") P.edge_threshold = std::stod(val());
        else if (a == "-L" || a == "--overlapThreshold") P.overlap_threshold = std::stod(val());
        else if (a == "-m" || a == "--readConfidence") P.read_confidence = std::stod(val());
        else if (a == "-n" || a == "--snpConfidence") P.snp_confidence = std::stod(val());
        else if (a == "-x" || a == "--mismatchRate") (void)val();
        else if (a == "--gpu") gpu = std::stoi(val());
        else if (a == "--help") { std::cout << kUsage; return 0; }
        else if (a == "--sv-file" || a == "--mod-file" || a == "--dot" || a == "--deepsomatic_output" || a == "--indelQuality") die("longphase_amd: " + a + " is not supported by the GPU path; use the reference binary");
        else { std::cerr << "longphase_amd: unknown option " << a << "\n" << kUsage; return 1; }
    }
    if (snp.empty() || bams.empty() || ref.empty()) { std::cerr << "longphase_amd phase: missing arguments\n" << kUsage; return 1; }
    if (ont == pb) { std::cerr << "longphase_amd phase: missing arguments. --ont or --pb\n" << kUsage; return 1; }   // Phasing.cpp:175-183
    P.is_ont = ont;

    std::vector<std::string> vcf_lines;
    if (!read_lines(snp, vcf_lines)) die("ERROR: Cannot open vcf file " + snp);
    std::vector<std::string> chr_order; std::map<std::string, ChrVariants> vars;
    parse_vcf(vcf_lines, P.phase_indel != 0, chr_order, vars);
    std::map<std::string, int> want; for (auto &kv : vars) if (!kv.second.pos.empty()) want[kv.first] = 1;
    std::map<std::string, std::string> seqs; read_fasta(ref, vars, seqs);
    std::map<std::string, ReadPack> packs;
    for (const std::string &b : bams) read_bam(b, threads, want, packs);

    lps_ctx *ctx = lps_create(gpu, &P);
    if (!ctx) die("longphase_amd: cannot create a GPU context (no CPU fallback)");
    std::map<std::string, std::map<int32_t, Phased>> res;
    for (const std::string &chr : chr_order) {                       // PhasingProcess.cpp:113-173
        ChrVariants &cv = vars[chr];
        if (cv.pos.empty() || !packs.count(chr) || !seqs.count(chr)) continue;
        ReadPack &pk = packs[chr];
        if (pk.ref_start.empty()) continue;
        pk.assign_name_ids();
        std::vector<uint8_t> r0(cv.pos.size()), a0(cv.pos.size()); std::vector<uint16_t> rl(cv.pos.size()), al(cv.pos.size());
        for (size_t i = 0; i < cv.pos.size(); ++i) { r0[i] = (uint8_t)cv.ref[i][0]; a0[i] = (uint8_t)cv.alt[i][0]; rl[i] = (uint16_t)cv.ref[i].size(); al[i] = (uint16_t)cv.alt[i].size(); }
        lps_variant_table vt{}; vt.n = (int64_t)cv.pos.size(); vt.pos = cv.pos.data(); vt.ref0 = r0.data(); vt.alt0 = a0.data(); vt.ref_len = rl.data(); vt.alt_len = al.data();
        const std::string &sq = seqs[chr];
        lps_read_batch rb = pk.view();
        std::vector<int32_t> ps(cv.pos.size()); std::vector<uint8_t> gt(cv.pos.size());
        lps_phase_result pr{(int64_t)cv.pos.size(), ps.data(), gt.data()};
        if (lps_begin_chromosome(ctx) || lps_set_variants(ctx, &vt) || lps_set_reference(ctx, sq.data(), (int64_t)sq.size()) || lps_push_reads(ctx, &rb) ||
            lps_phase_chromosome(ctx, &pr)) die(std::string("longphase_amd: ") + lps_last_error(ctx));
        auto &rc = res[chr];
        for (size_t i = 0; i < cv.pos.size(); ++i) if (ps[i]) rc[cv.pos[i]] = Phased{ps[i], gt[i] ? '1' : '0', gt[i] ? '0' : '1'};
        std::cerr << "(" << chr << ")";
    }
    std::cerr << "\n";
    lps_destroy(ctx);
    write_vcf(vcf_lines, prefix + ".vcf", res, vars, command);
    return 0;
}

// `longphase_amd view BAM CONTIG` — decoded records as SAM columns 1-11 (CPU-only check of the BGZF/BAM reader)
static int view_main(int argc, char **argv) {
    if (argc < 4) die("Usage: longphase_amd view <in.bam> <contig> [threads]");
    std::map<std::string, int> want{{argv[3], 1}}; std::map<std::string, ReadPack> packs;
    read_bam(argv[2], argc > 4 ? atoi(argv[4]) : 1, want, packs);
    const ReadPack &k = packs[argv[3]];
    std::string line;
    for (size_t i = 0; i < k.ref_start.size(); ++i) {
        line = k.names[i] + "\t" + std::to_string(k.flag[i]) + "\t" + argv[3] + "\t" + std::to_string(k.ref_start[i] + 1) + "\t" + std::to_string(k.mapq[i]) + "\t";
        for (uint64_t c = k.cigar_off[i]; c < k.cigar_off[i + 1]; ++c) line += std::to_string(k.cigar[c] >> 4) + "MIDNSHP=XB"[k.cigar[c] & 15];
        if (k.cigar_off[i] == k.cigar_off[i + 1]) line += "*";
        line += "\t*\t0\t0\t";
        for (int32_t j = 0; j < k.l_qseq[i]; ++j) line += "=ACMGRSVTWYHKDBN"[(k.seq[k.seq_off[i] + (j >> 1)] >> ((j & 1) ? 0 : 4)) & 15];
        line += "\t";
        for (int32_t j = 0; j < k.l_qseq[i]; ++j) line += (char)(k.qual[k.qual_off[i] + j] + 33);
        std::cout << line << "\n";
    }
    return 0;
}

int main(int argc, char **argv) {
    std::string command; for (int i = 0; i < argc; ++i) { if (i) command += " "; command += argv[i]; }
    if (argc < 2) { std::cout << "Version: " << kVersion << "\nUsage: longphase_amd <command> [options]\n    phase    run phasing algorithm on the GPU.\n"; return 0; }
    const std::string cmd = argv[1];
    if (cmd == "phase") return phase_main(argc, argv, command);
    if (cmd == "view") return view_main(argc, argv);
    if (cmd == "haplotag" || cmd == "somatic_haplotag") die("longphase_amd: the " + cmd + " scoring passes are available through the C-ABI (include/lps_abi.h); the BAM writer is not built yet");
    std::cerr << "Unrecognized command: " << cmd << "\n"; return 1;
}
