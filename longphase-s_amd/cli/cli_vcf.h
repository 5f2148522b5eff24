// cli_vcf.h — text inputs of longphase_amd: VCF row selection (SnpParser / VcfParser restatements), FASTA, and the phased-VCF rewriter.
#pragma once
#include "cli_common.h"
#include <limits>

// ------------------------------------------------------------------------------------------------ text inputs
static bool read_lines(const std::string &path, std::vector<std::string> &lines) {   // plain or gzip text
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    gzbuffer(f, 1u << 20);
    std::string all; size_t have = 0; int k;                           // the whole text first: the lines are then cut with memchr, their count known before the first is made
    for (;;) { if (all.size() - have < (1u << 22)) all.resize(std::max<size_t>(all.size() * 2, 1u << 24)); k = gzread(f, &all[have], (unsigned)std::min<size_t>(all.size() - have, 1u << 30)); if (k <= 0) break; have += (size_t)k; }
    gzclose(f);
    const char *b = all.data(), *e = b + have; size_t n = 0;
    for (const char *q = b; q < e;) { const char *nl = (const char *)memchr(q, '\n', (size_t)(e - q)); ++n; if (!nl) break; q = nl + 1; }
    lines.reserve(lines.size() + n);
    for (const char *q = b; q < e;) { const char *nl = (const char *)memchr(q, '\n', (size_t)(e - q)); lines.emplace_back(q, nl ? (size_t)(nl - q) : (size_t)(e - q)); if (!nl) break; q = nl + 1; }
    return true;
}

// rows of one contig, position-sorted, one per position (has(): is there a row at this 0-based position)
struct ChrVariants { std::vector<int32_t> pos; std::vector<std::string> ref, alt; bool has(int32_t p) const { return std::binary_search(pos.begin(), pos.end(), p); } };

static std::vector<std::string> split_tab(const std::string &s) {
    std::vector<std::string> f; size_t a = 0;
    while (true) { size_t b = s.find('\t', a);
        if (b == std::string::npos) { f.push_back(s.substr(a));
            break;
            } f.push_back(s.substr(a, b - a));
        a = b + 1;
        }
    return f;
}

// SnpParser::SnpParser (ParsingBam.cpp:222-359): het bi-allelic SNPs (bcf_is_snp: every allele one base), with --indels every other
// het bi-allelic record.  GT of the first sample must be 0/1, 1/0, 0|1 or 1|0.
// --indelQuality N (with --indels; :229-235, 325-340): a het record that is not a SNP and whose QUAL is below N is dropped before anything else is
// looked at, logged to <prefix>_removed_indels.log, and its FILTER reads INDEL_QUAL_FILTERED in the output VCF (IndelQual below).
// A whole-genome VCF holds millions of lines and the GPU is idle until the table exists: the lines are parsed in slices on `threads` host threads
// (no allocation per skipped line: fields are found in place), what the slices kept is applied in line order - first mention of a contig,
// "the later record at one position wins", the log and the first error are what one thread walking the file would produce.
struct IndelQual { int threshold = 0; std::ofstream log; std::map<std::string, std::set<int32_t>> filtered; };
struct VcfKept { int kind = 0; std::string chr, ref, alt, text; int32_t pos = 0; };   // 1 ##contig line, 2 row, 3 row dropped by --indelQuality (text: its log line), 4 error (text)
static void parse_vcf_line(const std::string &ln, bool indels, int iq_threshold, std::vector<VcfKept> &out) {
    if (ln.empty()) return;
    if (ln[0] == '#') {
        if (ln.compare(0, 13, "##contig=<ID=") == 0) { const size_t e = ln.find_first_of(",>", 13); VcfKept k; k.kind = 1; k.chr = ln.substr(13, e - 13); out.push_back(std::move(k)); }
        return;
    }
    // the first ten fields in place
    const char *b = ln.data(), *e = b + ln.size(); const char *fs[10]; size_t fl[10]; int nf = 0;
    for (const char *q = b; nf < 10;) { const char *t = (const char *)memchr(q, '\t', (size_t)(e - q)); fs[nf] = q; fl[nf] = (size_t)((t ? t : e) - q); ++nf; if (!t) break; q = t + 1; }
    if (nf < 10) return;
    auto has_comma = [](const char *p, size_t n) { return memchr(p, ',', n) != nullptr; };
    const char *ref = fs[3], *alt = fs[4]; const size_t rl = fl[3], al = fl[4];
    const bool iq_on = iq_threshold > 0 && indels;
    const bool unusable = has_comma(alt, al) || al == 0 || alt[0] == '<' || (al == 1 && (alt[0] == '.' || alt[0] == '*'));
    const bool is_snp = rl == 1 && al == 1;
    if (!iq_on && (unusable || (!is_snp && !indels))) return;
    if (unusable && has_comma(alt, al)) {                                  // all-single-base alleles: htslib calls it a SNP, the multi-allele check drops it
        bool all1 = rl == 1; size_t a = 0;
        while (all1) { const char *c = (const char *)memchr(alt + a, ',', al - a); const size_t bnd = c ? (size_t)(c - alt) : al; if (bnd - a != 1) all1 = false; if (!c) break; a = bnd + 1; }
        if (all1) return;
    }
    if (!is_snp && !indels) return;
    // GT position inside FORMAT; pieces as std::getline cuts them (a trailing empty piece does not exist)
    auto piece = [](const char *p, size_t n, size_t want, const char *&q, size_t &ql) -> bool {      // piece number `want` of p[0, n) split at ':'
        size_t idx = 0, a = 0;
        for (;;) { const char *c = (const char *)memchr(p + a, ':', n - a); const size_t bnd = c ? (size_t)(c - p) : n;
            if (!c && bnd == a) return false;                                 // (the empty piece behind a trailing ':' or of an empty string)
            if (idx == want) { q = p + a; ql = bnd - a; return true; }
            if (!c) return false;
            ++idx; a = bnd + 1; }
    };
    size_t gi = 0; bool have_gt = false;
    for (;; ++gi) { const char *q; size_t ql; if (!piece(fs[8], fl[8], gi, q, ql)) break; if (ql == 2 && q[0] == 'G' && q[1] == 'T') { have_gt = true; break; } }
    const char *gt = nullptr; size_t gl = 0;
    if (!have_gt || !piece(fs[9], fl[9], gi, gt, gl)) { VcfKept k; k.kind = 4; k.text = "pos " + std::string(fs[1], fl[1]) + " missing GT value"; out.push_back(std::move(k)); return; }
    if (!(gl == 3 && (gt[1] == '/' || gt[1] == '|') && ((gt[0] == '0' && gt[2] == '1') || (gt[0] == '1' && gt[2] == '0')))) return;
    VcfKept k; k.chr.assign(fs[0], fl[0]);
    if (!is_snp && iq_threshold > 0) {                                     // :325-340, before the multi-allele check
        const std::string qs(fs[5], fl[5]);
        float qual = 0.0f; bool missing = qs == ".";
        if (!missing) { try { qual = std::stof(qs); } catch (...) { qual = 0.0f; missing = true; } }
        if (qual < (float)iq_threshold) {
            const char *c = (const char *)memchr(alt, ',', al);
            k.kind = 3; k.pos = std::stoi(std::string(fs[1], fl[1])) - 1;
            k.text = k.chr + "\t" + std::string(fs[1], fl[1]) + "\t" + std::string(ref, rl) + "\t" + std::string(alt, c ? (size_t)(c - alt) : al) + "\t" + (missing ? std::string(".") : std::to_string(qual)) + "\n";
            out.push_back(std::move(k));
            return;
        }
    }
    if (unusable) return;
    k.kind = 2; k.pos = std::stoi(std::string(fs[1], fl[1])) - 1; k.ref.assign(ref, rl); k.alt.assign(alt, al);
    out.push_back(std::move(k));
}
static void parse_vcf(const std::vector<std::string> &lines, bool indels, std::vector<std::string> &chr_order, std::map<std::string, ChrVariants> &out, IndelQual *iq = nullptr, int threads = 1) {
    const int iq_threshold = iq ? iq->threshold : 0;
    const size_t n = lines.size();
    const int T = (int)std::max<size_t>(1, std::min<size_t>({(size_t)std::max(1, threads), (size_t)16, n / 20000 + 1}));
    std::vector<std::vector<VcfKept>> kept((size_t)T);
    auto slice = [&](int t) { const size_t a = n * (size_t)t / (size_t)T, b = n * (size_t)(t + 1) / (size_t)T; kept[(size_t)t].reserve((b - a) / 2 + 16);
        for (size_t i = a; i < b; ++i) parse_vcf_line(lines[i], indels, iq_threshold, kept[(size_t)t]); };
    { std::vector<std::thread> th; for (int t = 1; t < T; ++t) th.emplace_back(slice, t); slice(0); for (auto &x : th) x.join(); }
    // in line order: contigs in order of first mention, the rows of a contig with their line number (the later record at one position wins)
    struct Row { int32_t pos; uint32_t seq; VcfKept *k; };
    std::map<std::string, std::vector<Row>> rows; uint32_t seq = 0;
    std::string last_chr; std::vector<Row> *last = nullptr;
    for (auto &part : kept) for (VcfKept &k : part) {
        if (k.kind == 4) die(k.text);
        if (k.kind == 3) { if (iq->log.is_open()) iq->log << k.text; iq->filtered[k.chr].insert(k.pos); continue; }
        if (!last || k.chr != last_chr) { if (!out.count(k.chr)) { out[k.chr]; chr_order.push_back(k.chr); } last = &rows[k.chr]; last_chr = k.chr; }
        if (k.kind == 2) last->push_back(Row{k.pos, seq++, &k});
    }
    for (auto &kv : rows) {
        std::vector<Row> &r = kv.second; ChrVariants &cv = out[kv.first];
        if (!std::is_sorted(r.begin(), r.end(), [](const Row &a, const Row &b) { return a.pos < b.pos; }))
            std::sort(r.begin(), r.end(), [](const Row &a, const Row &b) { return a.pos != b.pos ? a.pos < b.pos : a.seq < b.seq; });
        cv.pos.reserve(r.size()); cv.ref.reserve(r.size()); cv.alt.reserve(r.size());
        for (size_t i = 0; i < r.size(); ++i) {
            if (i + 1 < r.size() && r[i + 1].pos == r[i].pos) continue;    // a later record at the same position replaces this one
            cv.pos.push_back(r[i].pos); cv.ref.push_back(std::move(r[i].k->ref)); cv.alt.push_back(std::move(r[i].k->alt));
        }
    }
}

static void read_fasta(const std::string &path, const std::map<std::string, ChrVariants> &want, std::map<std::string, std::string> &seqs) {
    std::ifstream f(path); if (!f) die("ERROR: Cannot open reference " + path);
    std::string ln, cur; std::string *dst = nullptr;
    while (std::getline(f, ln)) {
        if (!ln.empty() && ln[0] == '>') { std::string name = ln.substr(1, ln.find_first_of(" \t", 1) - 1);
            dst = want.count(name) ? &seqs[name] : nullptr;
            continue;
            }
        if (dst) { if (!ln.empty() && ln.back() == '\r') ln.pop_back(); dst->append(ln); }
    }
}

// SnpParser::preprocessDeepsomaticVCF (ParsingBam.cpp:651-835), `phase --deepsomatic_output`: keeps the records whose FILTER holds GERMLINE and
// rewrites their GT to the unphased diploid genotype whose expected allele fractions (1 / 0.5+0.5) lie closest, in squared error, to the observed
// ones - AD when it has one count per allele and a positive sum, else 1 - sum(VAF) and the VAF values.  Header lines pass through.
static void preprocess_deepsomatic(const std::vector<std::string> &in, std::vector<std::string> &out) {
    for (const std::string &line : in) {
        if (line.compare(0, 1, "#") == 0) { out.push_back(line); continue; }
        std::istringstream iss(line);
        std::vector<std::string> f((std::istream_iterator<std::string>(iss)), std::istream_iterator<std::string>());
        if (f.size() < 10) continue;
        if (f[6].find("GERMLINE") == std::string::npos) continue;
        auto split = [](const std::string &s, char d) { std::vector<std::string> v; std::istringstream ss(s); std::string x; while (std::getline(ss, x, d)) v.push_back(x); return v; };
        const std::vector<std::string> fmt = split(f[8], ':'); std::vector<std::string> smp = split(f[9], ':');
        int vaf_i = -1, gt_i = -1, ad_i = -1;
        for (int i = 0; i < (int)fmt.size(); ++i) { if (fmt[i] == "VAF") vaf_i = i; if (fmt[i] == "GT") gt_i = i; if (fmt[i] == "AD") ad_i = i; }
        if (gt_i >= 0 && gt_i < (int)smp.size()) {
            int alt_count = 0;
            if (!f[4].empty() && f[4] != ".") for (const std::string &t : split(f[4], ',')) if (!t.empty()) ++alt_count;
            const int n_allele = alt_count + 1;
            std::vector<double> obs; bool have = false;
            if (ad_i >= 0 && ad_i < (int)smp.size()) {
                std::vector<long long> ad; long long sum = 0;
                for (const std::string &t : split(smp[ad_i], ',')) { long long v = 0; if (!(t == "." || t.empty())) { try { v = std::stoll(t); } catch (...) { v = 0; } } ad.push_back(v); }
                for (long long v : ad) sum += v;
                if (sum > 0 && (int)ad.size() == n_allele) { for (long long v : ad) obs.push_back((double)v / (double)sum); have = true; }
            }
            if (!have && vaf_i >= 0 && vaf_i < (int)smp.size()) {
                std::vector<double> vafs;
                for (const std::string &t : split(smp[vaf_i], ',')) { if (t == "." || t.empty()) continue; try { vafs.push_back(std::stod(t)); } catch (...) {} }
                if (alt_count == (int)vafs.size() && alt_count >= 1) {
                    double sum = 0.0; for (double v : vafs) sum += v;
                    obs.clear(); obs.push_back(std::max(0.0, 1.0 - sum)); for (double v : vafs) obs.push_back(v);
                    have = true;
                }
            }
            if (have && n_allele >= 1) {
                int best_a = 0, best_b = 0; double best = std::numeric_limits<double>::infinity();
                for (int a = 0; a < n_allele; ++a) for (int b = a; b < n_allele; ++b) {
                    double cost = 0.0;
                    for (int i = 0; i < n_allele; ++i) { const double e = (a == b) ? (i == a ? 1.0 : 0.0) : ((i == a || i == b) ? 0.5 : 0.0); const double d = obs[(size_t)i] - e; cost += d * d; }
                    if (cost < best) { best = cost; best_a = a; best_b = b; }
                }
                smp[(size_t)gt_i] = std::to_string(best_a) + "/" + std::to_string(best_b);
                f[9].clear(); for (size_t i = 0; i < smp.size(); ++i) { if (i) f[9] += ":"; f[9] += smp[i]; }
            }
        }
        std::string o; for (size_t i = 0; i < f.size(); ++i) { if (i) o += "\t"; o += f[i]; }
        out.push_back(o);
    }
}

// ------------------------------------------------------------------------------------------------ VCF rewriter
struct Phased { int32_t ps; char a, b; };
// SnpParser::writeLine (ParsingBam.cpp:460-635) restated; SVParser::writeLine (:1042-1193) and METHParser::writeLine (:1788-1942) differ from it only
// in how a record finds its result - `lookup(chromosome, 1-based POS)` returns it, or nullptr when the record is not phased or was not a row of the table
// (records are independent of each other: the header goes first, then the records in slices of the line vector, one host thread each, written in order)
static int g_text_threads = 8;
template <class Lookup>
static void rewrite_vcf(const std::vector<std::string> &lines, const std::string &out_path, const std::string &command, Lookup lookup, const IndelQual *iq = nullptr) {
    std::ofstream o(out_path); if (!o) die("Fail to open write file: " + out_path);
    bool ps_def = false, cmd_done = false;
    auto colon_index = [](const std::string &fmt, size_t upto) { int c = 0;
        for (size_t i = 0; i < upto; ++i) if (fmt[i] == ':') ++c;
        return c;
        };
    auto value_start = [](const std::string &v, int colons) { int cur = 0;
        size_t st = 0;
        for (size_t i = 0; i < v.size(); ++i) { if (cur >= colons) break;
            if (v[i] == ':') ++cur;
            ++st;
            } return st;
        };
    // one record -> its output line(s), appended to `out`
    auto record = [&](const std::string &in, std::string &out) {
        std::istringstream iss(in);
        std::vector<std::string> f((std::istream_iterator<std::string>(iss)), std::istream_iterator<std::string>());
        if (f.empty()) return;
        if (f.size() < 10) { out += in; out += '\n'; return; }
        const int32_t pidx = std::stoi(f[1]) - 1;
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
        const Phased *ph = lookup(f[0], pidx + 1);
        if (ph) {
            f[8] += ":PS"; f[9] += ":" + std::to_string(ph->ps);
            const size_t gp = f[8].find("GT"); const size_t st = value_start(f[9], colon_index(f[8], gp));
            f[9][st] = ph->a; f[9][st + 1] = '|'; f[9][st + 2] = ph->b;
        } else { f[8] += ":PS"; f[9] += ":."; }
        if (iq && iq->threshold > 0) { auto fc = iq->filtered.find(f[0]); if (fc != iq->filtered.end() && fc->second.count(pidx)) f[6] = "INDEL_QUAL_FILTERED"; }   // :619-623
        for (size_t i = 0; i < f.size(); ++i) { if (i) out += '\t'; out += f[i]; }
        out += '\n';
    };
    // header lines (and whatever else starts with '#'): in order, they carry state
    auto header = [&](const std::string &in) {
        if (in.compare(0, 2, "##") == 0) { if (in.compare(0, 16, "##FORMAT=<ID=PS,") == 0) ps_def = true; o << in << "\n";
            if (iq && iq->threshold > 0 && in.compare(0, 17, "##FILTER=<ID=PASS") == 0)          // :467-473
                o << "##FILTER=<ID=INDEL_QUAL_FILTERED,Description=\"Indel filtered due to QUAL below threshold (" << iq->threshold << ")\">\n";
            return; }
        if (!cmd_done) {
            if (!ps_def) { o << "##FORMAT=<ID=PS,Number=1,Type=Integer,Description=\"Phase set identifier\">\n"; ps_def = true; }
            o << "##longphaseVersion=" << kVersion << "\n" << "##commandline=\"" << command << "\"\n"; cmd_done = true;
        }
        o << in << "\n";
    };
    auto is_header = [](const std::string &in) { return in.compare(0, 2, "##") == 0 || in.compare(0, 6, "#CHROM") == 0 || in.compare(0, 6, "#chrom") == 0; };
    size_t i = 0;
    while (i < lines.size()) {
        if (is_header(lines[i])) { header(lines[i]); ++i; continue; }
        size_t j = i; while (j < lines.size() && !is_header(lines[j])) ++j;      // a run of records
        const size_t n = j - i; const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, g_text_threads), n / 4096 + 1));
        std::vector<std::string> part((size_t)nt); std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { const size_t a = i + n * (size_t)t / (size_t)nt,
                b = i + n * (size_t)(t + 1) / (size_t)nt; part[(size_t)t].reserve((b - a) * 96); for (size_t k = a; k < b; ++k) record(lines[k], part[(size_t)t]); });
        for (auto &x : th) x.join();
        for (const std::string &x : part) o.write(x.data(), (std::streamsize)x.size());
        i = j;
    }
}

static void write_vcf(const std::vector<std::string> &lines, const std::string &out_path, const std::map<std::string, std::map<int32_t, Phased>> &res,
                      const std::map<std::string, ChrVariants> &vars, const std::string &command, const IndelQual *iq = nullptr) {
    rewrite_vcf(lines, out_path, command, [&](const std::string &chr, int32_t pos1) -> const Phased * {
        auto rc = res.find(chr); auto vc = vars.find(chr);
        if (rc == res.end() || vc == vars.end()) return nullptr;
        auto it = rc->second.find(pos1 - 1);
        if (it == rc->second.end() || !std::binary_search(vc->second.pos.begin(), vc->second.pos.end(), pos1 - 1)) return nullptr;
        return &it->second;
    }, iq);
}

// ------------------------------------------------------------------------------------------------ haplotag
// Phased-het rows of the SNP VCF = the haplotag table: VcfParser::parserProcess (src/haplotag/HaplotagVcfParser.cpp:234-400).
struct PhasedRow { std::string ref, alt; int32_t ps; uint8_t hp1_is_alt; };
static void parse_phased_vcf(const std::vector<std::string> &lines, std::vector<std::string> &chr_vec, std::map<std::string, int> &chr_len,
                             std::map<std::string, std::map<int32_t, PhasedRow>> &rows) {
    for (const std::string &in : lines) {
        if (in.compare(0, 2, "##") == 0) {
            if (in.find("contig=") != std::string::npos) {                                   // :236-248 (needs ",length=")
                const size_t a = in.find("ID=") + 3, b = in.find(",length="), e = in.find(">");
                if (b == std::string::npos) die("[ERROR] contig header line without length: " + in);
                const std::string chr = in.substr(a, b - a);
                chr_vec.push_back(chr); chr_len[chr] = std::stoi(in.substr(b + 8, e - b - 8));
            }
            if (in.compare(0, 16, "##FORMAT=<ID=PS,") == 0 && in.find("Type=Integer") == std::string::npos) die("longphase_amd: only an Integer PS field is supported");
            continue;
        }
        if (in.empty() || in[0] == '#') continue;
        std::istringstream iss(in);
        std::vector<std::string> f((std::istream_iterator<std::string>(iss)), std::istream_iterator<std::string>());
        if (f.empty()) continue;
        if (f.size() < 10) die("[ERROR](VcfParser::parserProcess) => VCF file format not supported: " + in);
        auto start_of = [&](const char *key) { const size_t kp = f[8].find(key);
            int colons = 0;
            for (size_t i = 0; i < kp && i < f[8].size(); ++i) if (f[8][i] == ':') ++colons;
            int cur = 0;
            size_t st = 0;
            for (size_t i = 0; i < f[9].size(); ++i) { if (cur >= colons) break;
                if (f[9][i] == ':') ++cur;
                ++st;
                } return st;
            };
        const size_t g = start_of("GT");
        if (g + 2 >= f[9].size() + 0 && g + 2 > f[9].size() - 1) continue;
        if (!(f[9][g] != f[9][g + 2] && f[9][g + 1] == '|')) continue;                      // phased hetero GT only (:296)
        const size_t ps0 = start_of("PS");
        const size_t pe = f[9].find(':', ps0 + 1);
        const std::string psv = pe != std::string::npos ? f[9].substr(ps0, pe - ps0) : f[9].substr(ps0);
        PhasedRow r; r.ref = f[3];
        if (f[4].find(',') != std::string::npos) { if (f[9].find('2') != std::string::npos) continue;
            r.alt = f[4].substr(0, f[4].find(','));
            }   // :333-347
        else r.alt = f[4];
        try { r.ps = std::stoi(psv); } catch (...) { die("longphase_amd: phased record without an integer PS value: " + in); }
        if (f[9][g] == '0' && f[9][g + 2] == '1') r.hp1_is_alt = 0;
        else if (f[9][g] == '1' && f[9][g + 2] == '0') r.hp1_is_alt = 1;
        else die("longphase_amd: phased genotype other than 0|1 / 1|0 is not supported: " + in);
        rows[f[0]][std::stoi(f[1]) - 1] = r;
    }
}

// BGZF writer: the byte stream is cut into 0xff00-byte blocks (htslib's BGZF_BLOCK_SIZE) that are deflated by a thread pool and
// written in order; ends with the 28-byte EOF block.

struct TumorRow { std::string ref, alt; int kind; };                    // kind: 1 SNP, 2 insertion, 3 deletion, 4 MNP (VarData::setVariantType)
static void parse_tumor_vcf(const std::vector<std::string> &lines, std::vector<std::string> &chr_vec, std::map<std::string, int> &chr_len,
                            std::map<std::string, std::map<int32_t, TumorRow>> &rows) {
    for (const std::string &in : lines) {
        if (in.compare(0, 2, "##") == 0) {
            if (in.find("contig=") != std::string::npos) {
                const size_t a = in.find("ID=") + 3, b = in.find(",length="), e = in.find(">");
                if (b == std::string::npos) die("[ERROR] contig header line without length: " + in);
                const std::string chr = in.substr(a, b - a); chr_vec.push_back(chr); chr_len[chr] = std::stoi(in.substr(b + 8, e - b - 8));
            }
            continue;
        }
        if (in.empty() || in[0] == '#') continue;
        std::istringstream iss(in);
        std::vector<std::string> f((std::istream_iterator<std::string>(iss)), std::istream_iterator<std::string>());
        if (f.empty()) continue;
        if (f.size() < 10) die("[ERROR](VcfParser::parserProcess) => VCF file format not supported: " + in);
        const size_t kp = f[8].find("GT"); int colons = 0; for (size_t i = 0; i < kp && i < f[8].size(); ++i) if (f[8][i] == ':') ++colons;
        int cur = 0; size_t g = 0; for (size_t i = 0; i < f[9].size(); ++i) { if (cur >= colons) break; if (f[9][i] == ':') ++cur; ++g; }
        if (g + 2 >= f[9].size() + 1) continue;
        const char a = f[9][g], m = f[9][g + 1], b = g + 2 < f[9].size() ? f[9][g + 2] : '\0';
        if (a != b && m == '|') die("longphase_amd: phased records in the tumor VCF are not supported: " + in);
        // HaplotagVcfParser.cpp:296-400 would need PS handling
        if (!((a == '1' && m == '/' && b == '1') || (a == '0' && m == '/' && b == '1'))) continue;               // :470-520: 1/1 and 0/1 only
        TumorRow r; r.ref = f[3]; r.alt = f[4].find(',') != std::string::npos ? f[4].substr(0, f[4].find(',')) : f[4];
        if (r.ref.size() == 1 && r.alt.size() == 1) r.kind = 1;
        else if (r.ref.size() == 1 && r.alt.size() > 1) r.kind = 2;
        else if (r.ref.size() > 1 && r.alt.size() == 1) r.kind = 3;
        else if (r.ref.size() > 1 && r.ref.size() == r.alt.size()) r.kind = 4; else die("(loadVariantType)Invalid allele: " + r.ref + " " + r.alt);
        if ((r.kind == 2 || r.kind == 3) && std::abs((int)r.alt.size() - (int)r.ref.size()) > 100) continue;
        // tumor INDELs longer than 100 bp are skipped
        rows[f[0]][std::stoi(f[1]) - 1] = r;
    }
}

// TumorPurityEstimator (src/somatic_haplotag/TumorPurityEstimator.cpp) restated: LCVF filters :92-150, histogram of the normal germline read
// counts smoothed with a sigma-0.5 Gaussian :443-600, peak / valley analysis for the dynamic count threshold :649-1060, box-plot statistics with
// linear-interpolation percentiles :281-344, one outlier-removal round, and the quadratic model in (median, IQR) :66.  Writes <prefix>_purity.out.
