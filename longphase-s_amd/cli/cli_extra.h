// cli_extra.h — the SV and MOD inputs of `phase` (--sv-file, --mod-file): the tables SVParser (src/phase/ParsingBam.cpp:915-1027) and METHParser
// (:1647-1786, 1944-1952) build from their VCFs, row for row including what they drop, and the two extra output files (<prefix>_SV.vcf,
// <prefix>_mod.vcf; :1029-1193, 1667-1678, 1788-1942).  The rows reach the GPU as lps_extra_variants (include/lps_abi.h).
#pragma once
#include "cli_common.h"
#include "cli_vcf.h"
#include <array>

static std::vector<std::string> split_space(const std::string &s) {      // the reference tokenises on any white space (istream_iterator)
    std::istringstream iss(s);
    return std::vector<std::string>((std::istream_iterator<std::string>(iss)), std::istream_iterator<std::string>());
}

// index of the first character of the GT value inside the sample column (:964-980)
static size_t gt_value_start(const std::string &fmt, const std::string &smp) {
    int colons = 0; const int gt = (int)fmt.find("GT");
    for (int i = 0; i < gt; ++i) if (fmt[(size_t)i] == ':') ++colons;
    int cur = 0; size_t st = 0;
    for (size_t i = 0; i < smp.size(); ++i) { if (cur >= colons) break; if (smp[i] == ':') ++cur; ++st; }
    return st;
}
static bool gt_is_hom(const std::string &smp, size_t st) { return st + 2 <= smp.size() && smp.c_str()[st] == smp.c_str()[st + 2]; }

struct SvTable {
    // chromosome -> VCF POS (1-based, as the reference keys it, :1006,1014) -> SVLEN values
    std::map<std::string, std::map<int, std::map<int, bool>>> chr;
    bool find(const std::string &c, int key) const { auto it = chr.find(c); return it != chr.end() && it->second.count(key); }   // SVParser::findSV (:1194-1206)

    void parse(const std::vector<std::string> &lines, const std::map<std::string, ChrVariants> &snps) {
        std::map<std::string, std::map<int, bool>> dup;                 // posDuplicate, keyed by the 0-based position
        for (const std::string &in : lines) {
            if (in.empty() || in[0] == '#') continue;
            const std::vector<std::string> f = split_space(in);
            if (f.size() < 10) continue;
            const int pos = std::stoi(f[1]) - 1; const std::string &c = f[0];
            bool filter = gt_is_hom(f[9], gt_value_start(f[8], f[9]));                                                   // :984-986
            { auto sc = snps.find(c); if (sc != snps.end() && sc->second.has(pos)) filter = true; }               // :988-990 the row sits on a SNP
            auto &d = dup[c]; auto di = d.find(pos);
            if (di == d.end()) d[pos] = false; else { di->second = true; filter = true; }                                // :992-999 second record at a position
            if (filter) continue;
            const size_t at = f[7].find("SVLEN=");
            if (at == std::string::npos) continue;
            const size_t semi = f[7].find(';', at + 6);
            chr[c][std::stoi(f[1])][std::stoi(f[7].substr(at + 6, semi - (at + 6)))] = true;
        }
        // :930-940 positions seen twice are erased - by their 0-based value from a map keyed 1-based, i.e. the row one base to the LEFT goes
        for (auto &c : dup) for (auto &p : c.second) if (p.second) chr[c.first].erase(p.first);
    }
};

struct ModEntry { bool reverse, modified; };
struct ModTable {
    std::map<std::string, std::map<int, std::map<std::string, ModEntry>>> chr;     // chromosome -> representative position (0-based) -> read name
    std::map<int, int> representative;                                              // position -> representative position, ONE map for all chromosomes (:1783)

    void parse(const std::vector<std::string> &lines, const std::map<std::string, ChrVariants> &snps, const SvTable &sv) {
        int rep = -1, up = -1;                                          // representativePos, upMethPos: carried across chromosomes as in the reference
        for (const std::string &in : lines) {
            if (in.empty() || in[0] == '#') continue;
            const std::vector<std::string> f = split_space(in);
            if (f.size() < 10) continue;
            const int pos = std::stoi(f[1]) - 1; const std::string &c = f[0];
            if (up + 1 != pos) rep = pos;                               // :1709-1711 a run of consecutive positions is one row, at its first position
            if (gt_is_hom(f[9], gt_value_start(f[8], f[9]))) continue;  // :1725-1727
            { auto sc = snps.find(c); if ((sc != snps.end() && sc->second.has(pos)) || sv.find(c, pos)) continue; }   // :1730-1732 (findSV with the 0-based value)
            bool reverse;
            if (f[7].find("RS=P") != std::string::npos) reverse = false; else if (f[7].find("RS=N") != std::string::npos) reverse = true; else continue;
            auto list = [&](const char *key, bool modified) {           // :1748-1780
                int rp = (int)f[7].find(key); rp = (int)f[7].find("=", (size_t)rp); rp++;
                const int nx = (int)f[7].find(";", (size_t)rp);
                std::stringstream ss(f[7].substr((size_t)rp, (size_t)(nx - rp))); std::string read;
                while (std::getline(ss, read, ',')) chr[c][rep][read] = ModEntry{reverse, modified};
            };
            list("MR=", true); list("NR=", false);
            representative[pos] = rep; up = pos;
        }
    }
};

// lps_extra_variants of one contig: names are mapped onto the contig's name ranks; names no alignment of the contig carries are left out
struct ExtraRows {
    std::vector<int32_t> sv_pos, sv_len, mod_pos; std::vector<uint64_t> mod_off; std::vector<uint32_t> mod_name; std::vector<uint8_t> mod_flag;
    lps_extra_variants x{};
    bool any() const { return !sv_pos.empty() || !mod_pos.empty(); }
    void build(const SvTable &sv, const ModTable &mod, const std::string &c, const std::vector<std::pair<const char *, size_t>> &names, const std::vector<uint32_t> &name_id,
               int sv_window, double sv_threshold) {
        auto si = sv.chr.find(c);
        if (si != sv.chr.end()) for (auto &st : si->second) for (auto &ln : st.second) { sv_pos.push_back(st.first - 1); sv_len.push_back(ln.first); }   // SV_map (:1224-1230)
        auto mi = mod.chr.find(c);
        mod_off.push_back(0);
        if (mi != mod.chr.end() && !mi->second.empty()) {
            std::unordered_map<std::string, uint32_t> id; id.reserve(names.size() * 2);
            for (size_t i = 0; i < names.size(); ++i) id.emplace(std::string(names[i].first, names[i].second), name_id[i]);
            std::vector<std::pair<uint32_t, uint8_t>> row;
            for (auto &r : mi->second) {
                row.clear();
                for (auto &e : r.second) { auto it = id.find(e.first); if (it != id.end()) row.emplace_back(it->second, (uint8_t)((e.second.modified ? 1 : 0) | (e.second.reverse ? 2 : 0))); }
                std::sort(row.begin(), row.end());
                mod_pos.push_back(r.first);
                for (auto &e : row) { mod_name.push_back(e.first); mod_flag.push_back(e.second); }
                mod_off.push_back(mod_name.size());
            }
        }
        x.n_sv = (int64_t)sv_pos.size(); x.sv_pos = sv_pos.data(); x.sv_len = sv_len.data();
        x.n_mod = (int64_t)mod_pos.size(); x.mod_pos = mod_pos.data(); x.mod_off = mod_off.data(); x.mod_name = mod_name.data(); x.mod_flag = mod_flag.data();
        x.sv_window = sv_window; x.sv_threshold = sv_threshold;
    }
};

// <prefix>_SV.vcf: a record carries a result when its position (0-based) is phased AND a kept SV row starts there (:1148-1152)
static void write_sv_vcf(const std::vector<std::string> &lines, const std::string &path, const std::map<std::string, std::map<int32_t, Phased>> &res, const SvTable &sv,
                         const std::string &command) {
    rewrite_vcf(lines, path, command, [&](const std::string &c, int32_t pos1) -> const Phased * {
        auto rc = res.find(c); if (rc == res.end() || !sv.find(c, pos1)) return nullptr;
        auto it = rc->second.find(pos1 - 1); return it == rc->second.end() ? nullptr : &it->second;
    });
}
// <prefix>_mod.vcf: every record of a run reports the result of the run's representative position (:1819-1825, 1897-1900); positions the reader
// never stored map to 0 (operator[] on the map)
static void write_mod_vcf(const std::vector<std::string> &lines, const std::string &path, const std::map<std::string, std::map<int32_t, Phased>> &res, const ModTable &mod,
                          const std::string &command) {
    rewrite_vcf(lines, path, command, [&](const std::string &c, int32_t pos1) -> const Phased * {
        auto ri = mod.representative.find(pos1 - 1); const int rep = ri == mod.representative.end() ? 0 : ri->second;
        auto rc = res.find(c); auto mc = mod.chr.find(c);
        if (rc == res.end() || mc == mod.chr.end() || !mc->second.count(rep)) return nullptr;
        auto it = rc->second.find(rep); return it == rc->second.end() ? nullptr : &it->second;
    });
}

// `haplotag --sv-file / --mod-file` (src/haplotag/HaplotagVcfParser.cpp:269-300, 403-468): every phased heterozygous record of the file gives one
// vote to each read it lists - under RNAMES= in a SV file, under MR= in a modcall file - for the haplotype its GT puts ALT on (0|1: haplotype 2,
// 1|0: haplotype 1).  One table for all chromosomes, keyed by read name (VCF_Info::readSVHapCount); judgeSVHap adds it to the read's counts.
typedef std::unordered_map<std::string, std::array<int32_t, 2>> ReadVotes;
static void parse_read_votes(const std::vector<std::string> &lines, const char *key, ReadVotes &votes) {
    int hap = -1;                                                       // the reference leaves it unset for a GT other than 0|1 / 1|0: what the previous record left
    for (const std::string &in : lines) {
        if (in.empty() || in[0] == '#') continue;
        const std::vector<std::string> f = split_space(in);
        if (f.empty()) continue;
        if (f.size() < 10) die("[ERROR](VcfParser::parserProcess) => VCF file format not supported: " + in);
        const size_t st = gt_value_start(f[8], f[9]);
        const char *g = f[9].c_str();
        if (st + 2 > f[9].size() || !(g[st] != g[st + 2] && g[st + 1] == '|')) continue;           // only phased heterozygous records
        int rp = (int)f[7].find(key); rp = (int)f[7].find("=", (size_t)rp); rp++;
        const int nx = (int)f[7].find(";", (size_t)rp);
        std::stringstream ss(f[7].substr((size_t)rp, (size_t)(nx - rp)));
        if (g[st] == '0' && g[st + 2] == '1') hap = 1; else if (g[st] == '1' && g[st + 2] == '0') hap = 0;
        std::string read;
        while (std::getline(ss, read, ',')) { auto it = votes.find(read); if (it == votes.end()) it = votes.emplace(read, std::array<int32_t, 2>{0, 0}).first; if (hap >= 0) it->second[(size_t)hap]++; }
    }
}
