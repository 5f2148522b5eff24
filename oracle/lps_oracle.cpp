// lps_oracle — TEST INFRASTRUCTURE ONLY.  CPU restatement of LongPhase-S's hot path on the
// SoA inputs of include/lps_abi.h.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library, and only as the checker; the product (liblps_hip.so) never links or calls it.
//
// Pinning: tests/test_oracle_golden.py, tests/test_oracle_haplotag_golden.py and tests/test_oracle_somatic_*_golden.py check this
// restatement against the golden vectors committed under tests/golden/, which tests/golden/make_golden.py made by running the real
// reference binary (oracle/_ref/longphase-s-ref, built by oracle/build_ref.sh from /root/reference) on generated inputs.  That build
// is zlib-only htslib with generated headers written by the script and without jemalloc (DESIGN.md §5 lists the stand-ins), so by the
// rule for reference builds the parity status reads "parity unpinned" although none of the stand-ins touches scoring arithmetic.
//
// Every function cites the reference lines (relative to /root/reference/) whose behaviour it restates.
// It is a restatement on index space (variant index instead of std::map<int,...> position keys), not a copy.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../include/lps_abi.h"

namespace {

struct Obs { int32_t var; int32_t allele; int32_t quality; };  // quality: base quality or sentinel -4/-5
struct Aln { int64_t read; std::vector<Obs> obs; bool emptied_by_filter = false; };

struct Table {
    const lps_variant_table *t;
    std::vector<uint8_t> danger;
    int64_t ref_len;          // FastaParser truncation: [0, lastVariant+5]
    const char *ref;
};

// src/shared/Util.cpp:21-54 homopolymerLength
int homopolymer_length(int64_t p, const char *ref, int64_t ref_len) {
    int len = 1;
    if (p + 1 >= ref_len) return len;
    char e = ref[p];
    int64_t q = p - 1;                       // reference .at(p-1) throws for p==0; generator avoids p==0
    while (q >= 0 && ref[q] == e) { --q; ++len; if (len >= 10 || q < 0) break; }
    q = p + 1;
    if (q < ref_len) {
        while (ref[q] == e) { ++q; ++len; if (q >= ref_len) break; if (len >= 10) break; }
    }
    return len;
}

// src/phase/ParsingBam.cpp:378-417 getVariants_markindel: an indel is "danger" when the 2-mer right after the
// site repeats five times (positions p+1..p+10).
void mark_danger(Table &T) {
    const lps_variant_table &t = *T.t;
    T.danger.assign(t.n, 0);
    auto at = [&](int64_t i) -> char { return (i >= 0 && i < T.ref_len) ? T.ref[i] : '\0'; };
    for (int64_t v = 0; v < t.n; ++v) {
        if (t.ref_len[v] <= 1 && t.alt_len[v] <= 1) continue;
        int64_t p = t.pos[v]; char a = at(p + 1), b = at(p + 2); int i = 0;
        while (i < 5) { if (a != at(p + 1) || b != at(p + 2)) break; p += 2; ++i; }
        T.danger[v] = (i == 5);
    }
}

inline char seq_base(const uint8_t *seq, int64_t i) {
    static const char nt[] = "=ACMGRSVTWYHKDBN";     // htslib seq_nt16_str
    return nt[(seq[i >> 1] >> ((~i & 1) << 2)) & 15];
}

struct ClipEvent { int32_t pos; uint8_t fb; };

// src/phase/ParsingBam.cpp:1303-1634 get_snp (SNP/indel rows; SV and MOD inputs are out of scope)
// returns false when the reference would have hit an unsupported CIGAR op (exit(1)).
bool extract_read(const Table &T, const lps_read_batch &b, int64_t r, std::vector<Obs> &out,
                  std::vector<ClipEvent> &clips) {
    const lps_variant_table &t = *T.t;
    out.clear();
    const uint32_t *cig = b.cigar + b.cigar_off[r];
    const int n_cig = (int)(b.cigar_off[r + 1] - b.cigar_off[r]);
    const uint8_t *seq = b.seq + b.seq_off[r];
    const uint8_t *qual = b.qual + b.qual_off[r];
    const int64_t l_qseq = b.l_qseq[r];
    int64_t ref_pos = b.ref_start[r], query_pos = 0;
    // :1318 monotone cursor == lower_bound because alignments arrive coordinate-sorted
    int64_t cur = std::lower_bound(t.pos, t.pos + t.n, (int32_t)ref_pos) - t.pos;
    for (int i = 0; i < n_cig; ++i) {
        const int op = cig[i] & 15; const int64_t len = cig[i] >> 4;
        while (cur < t.n && t.pos[cur] < ref_pos) ++cur;                                  // :1361-1364
        while (cur < t.n && t.pos[cur] < ref_pos + len) {                                 // :1368-1370
            if (!(op == 0 || op == 7 || op == 8)) break;                                  // :1521
            const int64_t vp = t.pos[cur], off = vp - ref_pos;
            if (query_pos + off + 1 > l_qseq) { out.clear(); return true; }               // :1453-1455 drop read
            int allele = -1, q = 0;
            const int rl = t.ref_len[cur], al = t.alt_len[cur];
            if (rl == 1 && al == 1) {                                                     // :1458-1466
                char base = seq_base(seq, query_pos + off);
                if (base == (char)t.ref0[cur]) allele = 0; else if (base == (char)t.alt0[cur]) allele = 1;
                q = qual[query_pos + off];
            }
            if (rl == 1 && al != 1 && i + 1 < n_cig) {                                    // :1470-1491 insertion
                allele = (ref_pos + len - 1 == vp && (cig[i + 1] & 15) == 1) ? 1 : 0;
                q = T.danger[cur] ? -5 : -4;
            }
            if (rl != 1 && al == 1 && i + 1 < n_cig) {                                    // :1495-1510 deletion
                allele = (ref_pos + len - 1 == vp && (cig[i + 1] & 15) == 2) ? 1 : 0;
                q = T.danger[cur] ? -5 : -4;
            }
            if (allele != -1) out.push_back({(int32_t)cur, allele, q});
            ++cur;
        }
        if (op == 0 || op == 7 || op == 8) { query_pos += len; ref_pos += len; }
        else if (op == 1) query_pos += len;
        else if (op == 2) {                                                               // :1539-1607
            // cur at end(): reference dereferences end() (UB) -> defined here as "no variant"
            if (cur < t.n && !(ref_pos + len + 1 == t.pos[cur]) && t.pos[cur] >= ref_pos && t.pos[cur] < ref_pos + len) {
                if (homopolymer_length(t.pos[cur], T.ref, T.ref_len) >= 3) {
                    if (query_pos + 1 > l_qseq) { out.clear(); return true; }             // :1559-1561
                    const int rl = t.ref_len[cur], al = t.alt_len[cur];
                    int allele = -1, q = 0;
                    if (rl == 1 && al == 1) {
                        char base = seq_base(seq, query_pos);
                        if (base == (char)t.ref0[cur]) allele = 0; else if (base == (char)t.alt0[cur]) allele = 1;
                        q = qual[query_pos];
                    } else if (rl != 1 && al == 1) { allele = 1; q = -4; }
                    if (allele != -1) { out.push_back({(int32_t)cur, allele, q}); ++cur; }
                }
            }
            ref_pos += len;
        }
        else if (op == 3) ref_pos += len;
        else if (op == 4) { query_pos += len; if (len > 5) clips.push_back({(int32_t)ref_pos, (uint8_t)(i == 0 ? 0 : 1)}); }  // :1613-1616,1636-1645
        else if (op == 5) { if (len > 5) clips.push_back({(int32_t)ref_pos, (uint8_t)(i == 0 ? 0 : 1)}); }
        else if (op == 6) {}
        else return false;                                                               // :1625-1628
    }
    return true;
}

// SV / MOD rows next to the SNP table, as BamParser holds them (src/phase/ParsingBam.cpp:1207-1235): SV_map[chr] = (start, svlen) pairs in
// start order, currentMod = map<pos, map<read name, RefAlt>>.  Index space: sv[k] / mod[k] are rows of lps_extra_variants; u_* give the row's
// index in the position-sorted union of all three tables (what the graph stages run on).
struct Extras {
    const lps_extra_variants *x = nullptr;
    std::vector<int32_t> u_of_snp, u_of_sv, u_of_mod, snp_of_u, upos;
};

// src/phase/ParsingBam.cpp:1303-1634 get_snp WITH the SV (:1397-1434) and MOD (:1373-1395) branches: the reference's three-cursor walk, cursor
// for cursor.  Emits unified indices; quality -1 = SV, -2 / -3 = MOD on the forward / reverse strand.
// Returns 0 ok, 1 unsupported CIGAR op (exit(1) in the reference), 2 the reference's loop would never end (two of the three cursors stand on
// the same position: none of its three branches takes the row).
int extract_read_x(const Table &T, const Extras &E, const lps_read_batch &b, int64_t r, std::vector<Obs> &out,
                   std::vector<ClipEvent> &clips, int *ub_hazard) {
    const lps_variant_table &t = *T.t; const lps_extra_variants &x = *E.x;
    out.clear();
    const uint32_t *cig = b.cigar + b.cigar_off[r];
    const int n_cig = (int)(b.cigar_off[r + 1] - b.cigar_off[r]);
    const uint8_t *seq = b.seq + b.seq_off[r];
    const uint8_t *qual = b.qual + b.qual_off[r];
    const int64_t l_qseq = b.l_qseq[r];
    const bool rev = (b.flag[r] & 0x10) != 0;
    int64_t ref_pos = b.ref_start[r], query_pos = 0;
    const int64_t nV = t.n, nS = x.n_sv, nM = x.n_mod;
    // :1318-1327 the three "first" cursors: alignments arrive coordinate-sorted, so each is a lower bound.  The SV cursor compares the 1-based
    // VCF start (SV_map holds `start`, :1014,1228) with the 0-based alignment start.
    int64_t cur = std::lower_bound(t.pos, t.pos + nV, (int32_t)ref_pos) - t.pos;
    int64_t cs = 0; while (cs < nS && (int64_t)x.sv_pos[cs] + 1 < ref_pos) ++cs;
    int64_t cm = std::lower_bound(x.mod_pos, x.mod_pos + nM, (int32_t)ref_pos) - x.mod_pos;
    // what `(*iter).first` reads at end(): for the two std::map cursors libstdc++ keeps _M_node_count right behind the header node, i.e. the
    // number of entries (UB, but stable with the reference's toolchain); every use of the SV value at end() is guarded.
    auto vpos = [&](int64_t i) -> int64_t { return i < nV ? t.pos[i] : nV; };
    auto mpos = [&](int64_t i) -> int64_t { return i < nM ? x.mod_pos[i] : nM; };
    for (int i = 0; i < n_cig; ++i) {
        const int op = cig[i] & 15; const int64_t len = cig[i] >> 4;
        int64_t modPos = mpos(cm), svPos = cs < nS ? x.sv_pos[cs] : 0, variantPos = vpos(cur);                  // :1351-1358
        while (cur < nV && variantPos < ref_pos) { ++cur; variantPos = vpos(cur); }                             // :1361-1364
        while ((cm < nM && modPos < ref_pos + len) || (cs < nS && svPos < ref_pos + len) || (cur < nV && variantPos < ref_pos + len)) {
            if ((cur == nV || modPos < variantPos) && (cs == nS || modPos < svPos) && cm < nM) {                // :1373-1395 MOD row
                const uint32_t *names = x.mod_name + x.mod_off[cm], *ne = x.mod_name + x.mod_off[cm + 1];
                const uint32_t *it = std::lower_bound(names, ne, b.name_id[r]);
                if (cur == nV && ub_hazard) (*ub_hazard)++;                                                      // compares with end()'s garbage
                if (it != ne && *it == b.name_id[r] && modPos < variantPos) {
                    const uint8_t f = x.mod_flag[x.mod_off[cm] + (it - names)];
                    if (((f >> 1) & 1) == (rev ? 1 : 0)) out.push_back({E.u_of_mod[cm], (f & 1) ? 0 : 1, rev ? -3 : -2});
                }
                ++cm; modPos = mpos(cm);
            } else if ((cur == nV || svPos < variantPos) && (cm == nM || svPos < modPos) && cs < nS) {          // :1397-1434 SV row
                int allele = 0;
                const int64_t sv_start = (int64_t)x.sv_pos[cs] + 1, sv_end = sv_start + std::abs((int64_t)x.sv_len[cs]);
                const double sv_region = (double)(sv_end - sv_start + 1);
                for (int j = std::max(i - x.sv_window, 0); j < std::min(i + x.sv_window, n_cig); ++j) {
                    const int o2 = cig[j] & 15; const int l2 = (int)(cig[j] >> 4);
                    if ((o2 == 1 || o2 == 2) && std::abs(sv_region - l2) / std::abs(sv_region) < x.sv_threshold) { allele = 1; break; }
                }
                out.push_back({E.u_of_sv[cs], allele, -1});
                ++cs; svPos = cs < nS ? x.sv_pos[cs] : 0;
            } else if ((cs == nS || variantPos < svPos) && (cm == nM || variantPos < modPos) && cur < nV) {     // :1437-1522 SNP / indel row
                if (!(op == 0 || op == 7 || op == 8)) break;
                const int64_t vp = variantPos, off = vp - ref_pos;
                if (query_pos + off + 1 > l_qseq) { out.clear(); return 0; }
                int allele = -1, q = 0;
                const int rl = t.ref_len[cur], al = t.alt_len[cur];
                if (rl == 1 && al == 1) {
                    char base = seq_base(seq, query_pos + off);
                    if (base == (char)t.ref0[cur]) allele = 0; else if (base == (char)t.alt0[cur]) allele = 1;
                    q = qual[query_pos + off];
                }
                if (rl == 1 && al != 1 && i + 1 < n_cig) { allele = (ref_pos + len - 1 == vp && (cig[i + 1] & 15) == 1) ? 1 : 0; q = T.danger[cur] ? -5 : -4; }
                if (rl != 1 && al == 1 && i + 1 < n_cig) { allele = (ref_pos + len - 1 == vp && (cig[i + 1] & 15) == 2) ? 1 : 0; q = T.danger[cur] ? -5 : -4; }
                if (allele != -1) out.push_back({E.u_of_snp[cur], allele, q});
                ++cur; variantPos = vpos(cur);
            } else return 2;
        }
        if (op == 0 || op == 7 || op == 8) { query_pos += len; ref_pos += len; }
        else if (op == 1) query_pos += len;
        else if (op == 2) {                                                                                       // :1539-1607, as in extract_read
            if (cur < nV && !(ref_pos + len + 1 == t.pos[cur]) && t.pos[cur] >= ref_pos && t.pos[cur] < ref_pos + len) {
                if (homopolymer_length(t.pos[cur], T.ref, T.ref_len) >= 3) {
                    if (query_pos + 1 > l_qseq) { out.clear(); return 0; }
                    const int rl = t.ref_len[cur], al = t.alt_len[cur];
                    int allele = -1, q = 0;
                    if (rl == 1 && al == 1) {
                        char base = seq_base(seq, query_pos);
                        if (base == (char)t.ref0[cur]) allele = 0; else if (base == (char)t.alt0[cur]) allele = 1;
                        q = qual[query_pos];
                    } else if (rl != 1 && al == 1) { allele = 1; q = -4; }
                    if (allele != -1) { out.push_back({E.u_of_snp[cur], allele, q}); ++cur; }
                }
            }
            ref_pos += len;
        }
        else if (op == 3) ref_pos += len;
        else if (op == 4) { query_pos += len; if (len > 5) clips.push_back({(int32_t)ref_pos, (uint8_t)(i == 0 ? 0 : 1)}); }
        else if (op == 5) { if (len > 5) clips.push_back({(int32_t)ref_pos, (uint8_t)(i == 0 ? 0 : 1)}); }
        else if (op == 6) {}
        else return 1;
    }
    return 0;
}

// src/phase/ParsingBam.cpp:837-912 filterSNP (variant-table half): which variants are erased
void filter_snp(const Table &T, std::vector<uint8_t> &erased) {
    const lps_variant_table &t = *T.t;
    erased.assign(t.n, 0);
    std::vector<int> hp(t.n);
    for (int64_t v = 0; v < t.n; ++v) hp[v] = homopolymer_length(t.pos[v], T.ref, T.ref_len);
    int64_t cur = 0, nxt = 1;
    while (cur < t.n && nxt < t.n) {
        if (hp[cur] >= 3 && hp[nxt] >= 3 && std::abs(t.pos[cur] - t.pos[nxt]) <= 2) { erased[nxt] = 1; ++nxt; continue; }
        // advance both to the next surviving entries
        ++cur; while (cur < t.n && erased[cur]) ++cur;
        nxt = cur + 1; while (nxt < t.n && erased[nxt]) ++nxt;
    }
}

// src/phase/PhasingGraph.cpp:1103-1227 Clip::Clip + getCNVInterval + updateThreshold.  The reference runs
// getCNVInterval twice (ctor + PhasingProcess.cpp:148) so every interval is appended twice.
struct CnvState {
    bool push = false, slowUp = false, slowDown = false;
    int curr = 0, reject = 0, pullDown = 0, slowDownCount = 0, candStart = -1, candEnd = -1;
    void reset() { *this = CnvState(); }
    void threshold(int up) {
        reject = up;
        if (up >= 20) { pullDown = up / 2; slowDownCount = 5; }
        else if (up >= 10) { pullDown = up / 2; slowDownCount = up / 4; }
        else { pullDown = 5; slowDownCount = 2; }
    }
};

void cnv_pass(std::map<int, std::pair<int, int>> clip /*copy: sentinel is appended then removed*/,
              std::vector<std::pair<int, int>> &cnv) {
    const int Area = 30000;
    if (clip.empty()) return;   // reference: UB/segfault
    { auto last = *clip.rbegin(); clip[last.first + Area] = last.second; }
    CnvState s;
    for (auto &kv : clip) {
        const int pos = kv.first, up = kv.second.first, down = kv.second.second;
        if (!s.push && !s.slowDown && !s.slowUp) {
            if (up >= 5 && s.curr == 0) { s.push = true; s.slowUp = false; s.slowDown = true; s.curr = up - down; s.candStart = pos; s.candEnd = pos + Area; s.threshold(up); }
            else if (up > down && s.curr == 0) { s.push = false; s.slowUp = true; s.slowDown = false; s.curr = up - down; s.candStart = pos; s.candEnd = pos + Area; }
        } else if (s.push && s.slowDown) {
            if (up > s.reject) { s.threshold(up); s.candStart = pos; s.candEnd = pos + Area; }
            s.curr = s.curr + up - down;
            if (s.curr > 30) s.candEnd = pos + Area;
            if (down >= s.pullDown) { cnv.emplace_back(s.candStart, pos); s.reset(); }
            else if (s.curr <= s.slowDownCount && pos <= s.candEnd) { cnv.emplace_back(s.candStart, pos); s.reset(); }
            if (pos > s.candEnd || s.curr <= 0 || pos - s.candStart >= 200000) s.reset();
        } else if (s.slowUp) {
            if (s.curr > 20 ? down >= s.curr / 4 : down >= 5) { cnv.emplace_back(s.candStart, pos); s.reset(); }
            else if (up >= 5) { s.push = true; s.slowUp = false; s.slowDown = true; s.curr = up - down; s.candStart = pos; s.candEnd = pos + Area; s.threshold(up); }
            else {
                s.curr = s.curr + up - down;
                if (s.curr > 30) s.candEnd = pos + Area;
                if (pos > s.candEnd || s.curr <= 0 || pos - s.candStart >= 200000) s.reset();
            }
        }
    }
}

// src/phase/PhasingGraph.cpp:707-781 overlap filter of several alignments of one read name
void overlap_filter(const lps_params &P, const lps_read_batch &b, const lps_variant_table &t,
                    std::vector<Aln> &alns, int *ub_hazard) {
    struct NameState { int second = 0; std::vector<int> kept; };
    std::map<uint32_t, NameState> st;
    std::vector<char> del(alns.size(), 0);
    auto first_pos = [&](const Aln &a) { return t.pos[a.obs.front().var]; };
    auto last_pos = [&](const Aln &a) { return t.pos[a.obs.back().var]; };
    std::map<uint32_t, int> name_count;
    for (auto &a : alns) name_count[b.name_id[a.read]]++;
    for (int ri = 0; ri < (int)alns.size(); ++ri) {
        Aln &a = alns[ri];
        if (a.obs.empty()) {
            // reference reads variantVec.front()/back() of an EMPTY vector (UB); only observable when the name
            // has other alignments.  Defined here: the alignment does not take part.
            if (name_count[b.name_id[a.read]] > 1 && ub_hazard) (*ub_hazard)++;
            continue;
        }
        NameState &s = st[b.name_id[a.read]];
        const int fp = first_pos(a), lp = last_pos(a);
        bool to_del = false;
        // alignRange[readName] is default-constructed {0,0} before the find() (:712,716): first stays 0
        while (0 <= fp && fp <= s.second) {
            if (lp < s.second) { to_del = true; del[ri] = 1; break; }
            int pre = (int)s.kept.size() - 1;
            if (pre < 0) break;
            const Aln &pa = alns[s.kept[pre]];
            const int ps = first_pos(pa), pe = last_pos(pa);
            double ovS = std::max(ps, fp), ovE = std::min(pe, lp);
            if (ovS > ovE) break;
            double ovLen = ovE - ovS + 1;
            double alS = std::max(pe, lp), alE = std::min(ps, fp);
            double span = alS - alE + 1;
            double ratio = ovLen / span;
            if (ratio >= P.overlap_threshold) {
                int len1 = pe - ps + 1, len2 = lp - fp + 1;
                if (len2 <= len1) { to_del = true; del[ri] = 1; break; }
                del[s.kept[pre]] = 1; s.kept.pop_back();
                s.second = (pre > 0) ? last_pos(alns[s.kept[pre - 1]]) : fp;
            } else break;
        }
        s.second = lp;
        if (!to_del) s.kept.push_back(ri);
    }
    std::vector<Aln> keep;
    for (size_t i = 0; i < alns.size(); ++i) if (!del[i]) keep.push_back(std::move(alns[i]));
    alns.swap(keep);
}

// src/phase/PhasingGraph.cpp:520-692 the four CNV mismatch-rate passes (called :785-791)
void cnv_filter(const lps_variant_table &t, const std::vector<std::pair<int, int>> &cnv, std::vector<Aln> &alns) {
    if (alns.empty() || cnv.empty()) return;
    auto in = [](int p, int s, int e) { return p >= s && p <= e; };
    std::vector<std::map<int, int>> mm(alns.size());
    size_t ci = 0;
    for (size_t r = 0; r < alns.size(); ++r) {                                            // calculateCnvMismatchRate
        auto &o = alns[r].obs; if (o.empty()) continue;
        int rs = t.pos[o.front().var], re = t.pos[o.back().var];
        while (ci > 0 && cnv[ci].first > rs) --ci;
        size_t i = ci;
        while (i < cnv.size() && cnv[i].first <= re) {
            for (auto &v : o) { int p = t.pos[v.var]; if (p > cnv[i].second) break; if (in(p, cnv[i].first, cnv[i].second) && v.allele == 1) mm[r][cnv[i].first]++; }
            ++i;
        }
        ci = i > 0 ? i - 1 : 0;
    }
    std::map<int, std::map<int, std::vector<int>>> agg;                                    // aggregateCnvReadMismatchRate
    ci = 0;
    for (size_t r = 0; r < alns.size(); ++r) {
        auto &o = alns[r].obs; if (o.empty()) continue;
        int rs = t.pos[o.front().var], re = t.pos[o.back().var];
        while (ci > 0 && cnv[ci].first > rs) --ci;
        size_t i = ci;
        while (i < cnv.size() && cnv[i].first <= re) {
            for (auto &v : o) {
                int p = t.pos[v.var]; if (p > cnv[i].second) break;
                auto it = mm[r].find(cnv[i].first);
                if (in(p, cnv[i].first, cnv[i].second) && it != mm[r].end()) agg[p][v.allele].push_back(it->second);
            }
            ++i;
        }
        ci = i > 0 ? i - 1 : 0;
    }
    std::map<int, double> miss;                                                            // calculateAverageMismatchRate
    if (!agg.empty()) {
        ci = 0;
        auto mean = [](const std::vector<int> &d) { double s = 0; for (int x : d) s += x; return d.empty() ? 0.0 : s / d.size(); };
        for (auto &kv : agg) {
            while (ci > 0 && cnv[ci].first > kv.first) --ci;
            size_t i = ci;
            while (i < cnv.size()) {
                if (cnv[i].first > kv.first) break;
                if (in(kv.first, cnv[i].first, cnv[i].second)) {
                    auto r0 = kv.second.find(0), r1 = kv.second.find(1);
                    if (r0 != kv.second.end() && r1 != kv.second.end()) {
                        double a = mean(r0->second), c = mean(r1->second);
                        if (a != 0 && c != 0) miss[kv.first] = c / (a + c);
                    }
                }
                ++i;
            }
            // reference leaves cnvIndex untouched in this pass (no write-back)
        }
    }
    if (miss.empty()) return;                                                             // filterHighMismatchVariants
    ci = 0;
    for (auto &a : alns) {
        auto &o = a.obs; if (o.empty()) continue;
        int rs = t.pos[o.front().var];
        while (ci > 0 && cnv[ci].first > rs) --ci;
        size_t k = 0;
        while (k < o.size()) {
            bool erase = false; int p = t.pos[o[k].var];
            size_t i = ci;
            while (i < cnv.size() && cnv[i].first <= p) {
                if (in(p, cnv[i].first, cnv[i].second)) {
                    auto m = miss.find(p);
                    if (m != miss.end() && m->second >= 0.7) { erase = true; o.erase(o.begin() + k); break; }
                }
                ++i;
            }
            if (!erase) ++k;
            ci = i > 0 ? i - 1 : 0;
        }
    }
}

struct SortKey { int pos; int var; int allele; int quality; };

}  // namespace

extern "C" {

// Dumps are optional (NULL) stage outputs used by the parity tests.
typedef struct oracle_dumps {
    int64_t obs_capacity;
    int32_t *obs_count;      /* n_reads, 0 for dropped alignments; BEFORE the overlap/CNV filters, AFTER filterSNP */
    int32_t *obs_var; int8_t *obs_allele; int8_t *obs_quality;
    int64_t n_obs;
    int64_t clip_capacity; int32_t *clip_pos; uint8_t *clip_fb; int64_t n_clips;
    int64_t node_capacity; int32_t *node_var; float *edge; int8_t *node_hp; int32_t *node_block; int64_t n_nodes;
    uint8_t *aln_deleted;    /* n_reads: 1 when removed by the overlap filter */
    int32_t n_cnv; int32_t cnv_capacity; int32_t *cnv_start; int32_t *cnv_end;   /* every interval twice, as in the reference; unbounded there (std::vector) */
    int32_t ub_hazard;       /* >0: input touched behaviour that is UB in the reference */
    int64_t n_pairs;
} oracle_dumps;

// xp != NULL: SV / MOD rows are co-phased (`--sv-file` / `--mod-file`).  From the extraction on everything runs on the position-sorted UNION of
// the three tables (the reference's maps are keyed by position and never ask which file a row came from); the dumps then hold union indices and
// out_sv / out_mod receive the rows' results.  Returns -2 bad CIGAR op, -3 the reference's extraction loop would not terminate.
int oracle_phase_x(const lps_params *Pp, const lps_variant_table *tp, const lps_extra_variants *xp, const char *ref, int64_t ref_len_in,
                   const lps_read_batch *bp, lps_phase_result *out, lps_phase_result *out_sv, lps_phase_result *out_mod, oracle_dumps *D) {
    const lps_params &P = *Pp; const lps_variant_table &t0 = *tp; const lps_read_batch &b = *bp;
    for (int64_t i = 0; i < out->n; ++i) { out->phase_set[i] = 0; out->gt[i] = 0; }
    if (out_sv) for (int64_t i = 0; i < out_sv->n; ++i) { out_sv->phase_set[i] = 0; out_sv->gt[i] = 0; }
    if (out_mod) for (int64_t i = 0; i < out_mod->n; ++i) { out_mod->phase_set[i] = 0; out_mod->gt[i] = 0; }
    if (D) { D->n_obs = 0; D->n_clips = 0; D->n_nodes = 0; D->n_cnv = 0; D->ub_hazard = 0; D->n_pairs = 0; }
    if (t0.n == 0) return 0;                                        // BamParser exits on an empty SNP map (:1218-1221); PhasingProcess skips the contig (:121)
    Table T; T.t = &t0; T.ref = ref;
    const int32_t last_pos = t0.pos[t0.n - 1];
    Extras E;
    if (xp && (xp->n_sv > 0 || xp->n_mod > 0)) {
        E.x = xp;
        const int64_t nU = t0.n + xp->n_sv + xp->n_mod;
        E.u_of_snp.resize(t0.n); E.u_of_sv.resize(xp->n_sv); E.u_of_mod.resize(xp->n_mod); E.upos.reserve(nU); E.snp_of_u.reserve(nU);
        int64_t a = 0, s2 = 0, m = 0;
        while (a < t0.n || s2 < xp->n_sv || m < xp->n_mod) {
            const int64_t pa = a < t0.n ? t0.pos[a] : INT64_MAX, ps = s2 < xp->n_sv ? xp->sv_pos[s2] : INT64_MAX, pm = m < xp->n_mod ? xp->mod_pos[m] : INT64_MAX;
            const int32_t u = (int32_t)E.upos.size();
            if (pa <= ps && pa <= pm) { E.u_of_snp[a] = u; E.snp_of_u.push_back((int32_t)a); E.upos.push_back((int32_t)pa); ++a; }
            else if (ps <= pm) { E.u_of_sv[s2] = u; E.snp_of_u.push_back(-1); E.upos.push_back((int32_t)ps); ++s2; }
            else { E.u_of_mod[m] = u; E.snp_of_u.push_back(-1); E.upos.push_back((int32_t)pm); ++m; }
        }
    }
    // the table the graph stages see: positions of the union (the other columns are not read after the extraction)
    lps_variant_table tu = t0;
    if (E.x) { tu.n = (int64_t)E.upos.size(); tu.pos = E.upos.data(); }
    const lps_variant_table &t = tu;
    T.ref_len = std::min<int64_t>(ref_len_in, (int64_t)last_pos + 6);   // ParsingBam.cpp:47 faidx_fetch_seq(0,last+5)
    mark_danger(T);
    const int A = P.connect_adjacent;

    // ---- a1/a2/a3: direct_detect_alleles + get_snp + getClip  (ParsingBam.cpp:1243-1645)
    std::vector<Aln> alns; std::vector<ClipEvent> clips; std::vector<Obs> tmp;
    for (int64_t r = 0; r < b.n_reads; ++r) {
        if (b.ref_start[r] >= last_pos) continue;                  // region "chr:1-<lastSNPPos>" (:1273)
        const int fl = b.flag[r];
        if (b.mapq[r] < P.mapping_quality || (fl & 0x4) || (fl & 0x100) || (fl & 0x400)) continue;   // :1282-1291
        if (E.x) { const int rc = extract_read_x(T, E, b, r, tmp, clips, D ? &D->ub_hazard : nullptr); if (rc) return rc == 1 ? -2 : -3; }
        else if (!extract_read(T, b, r, tmp, clips)) return -2;
        if (!tmp.empty()) { Aln a; a.read = r; a.obs = tmp; alns.push_back(std::move(a)); }
    }
    // ---- a5: filterSNP (ONT only)  (PhasingProcess.cpp:138-140)
    std::vector<uint8_t> erased(t0.n, 0);
    if (P.is_ont) {
        filter_snp(T, erased);
        for (auto &a : alns) {
            size_t k = 0; for (auto &o : a.obs) { const int sv = E.x ? E.snp_of_u[o.var] : o.var; if (sv < 0 || !erased[sv]) a.obs[k++] = o; }
            if (k == 0) a.emptied_by_filter = true;
            a.obs.resize(k);
        }
    }
    if (D) {
        if (D->obs_count) for (int64_t r = 0; r < b.n_reads; ++r) D->obs_count[r] = 0;
        int64_t n = 0;
        for (auto &a : alns) {
            if (D->obs_count) D->obs_count[a.read] = (int32_t)a.obs.size();
            for (auto &o : a.obs) { if (D->obs_var && n < D->obs_capacity) { D->obs_var[n] = o.var; D->obs_allele[n] = (int8_t)o.allele; D->obs_quality[n] = (int8_t)o.quality; } ++n; }
        }
        D->n_obs = n;
        int64_t c = 0;
        for (auto &e : clips) { if (D->clip_pos && c < D->clip_capacity) { D->clip_pos[c] = e.pos; D->clip_fb[c] = e.fb; } ++c; }
        D->n_clips = c;
        if (D->aln_deleted) for (int64_t r = 0; r < b.n_reads; ++r) D->aln_deleted[r] = 0;
    }
    if (alns.empty()) return 0;                                     // PhasingProcess.cpp:143-145
    // ---- a7: Clip / CNV intervals (twice)
    std::map<int, std::pair<int, int>> clipCount;
    for (auto &e : clips) { auto &c = clipCount[e.pos]; if (e.fb == 0) c.first++; else c.second++; }
    std::vector<std::pair<int, int>> cnv;
    if (clipCount.empty()) { if (D) D->ub_hazard++; }
    else { cnv_pass(clipCount, cnv); cnv_pass(clipCount, cnv); }
    if (D) { D->n_cnv = (int32_t)cnv.size(); for (size_t i = 0; i < cnv.size() && (int64_t)i < D->cnv_capacity && D->cnv_start; ++i) { D->cnv_start[i] = cnv[i].first; D->cnv_end[i] = cnv[i].second; } }
    // ---- a8: overlap filter; a9: CNV mismatch filter
    {
        std::vector<int64_t> before; for (auto &a : alns) before.push_back(a.read);
        overlap_filter(P, b, t, alns, D ? &D->ub_hazard : nullptr);
        if (D && D->aln_deleted) { size_t k = 0; for (int64_t r : before) { if (k < alns.size() && alns[k].read == r) ++k; else D->aln_deleted[r] = 1; } }
    }
    cnv_filter(t, cnv, alns);
    // ---- a10: type tagging + node set  (PhasingGraph.cpp:793-846)
    std::vector<int8_t> vtype(t.n, -1);                            // 0 SNP, 1 SV, 2 MOD, 3 indel, 4 danger indel (last writer wins)
    std::vector<uint8_t> is_node(t.n, 0);
    std::map<uint32_t, std::vector<SortKey>> merged;               // keyed by name order
    for (auto &a : alns) for (auto o : a.obs) {
        if (o.quality == -2 || o.quality == -3) { vtype[o.var] = 2; o.quality = 60; }                       // :803-806
        else if (o.quality == -1) { vtype[o.var] = 1; o.quality = (o.allele == 1) ? 60 : 30; }                // :808-817
        else if (o.quality == -4) { vtype[o.var] = 3; o.quality = 60; }
        else if (o.quality == -5) { vtype[o.var] = 4; o.quality = 60; }
        else vtype[o.var] = 0;
        merged[b.name_id[a.read]].push_back({t.pos[o.var], o.var, o.allele, o.quality});
        is_node[o.var] = 1;
    }
    std::vector<int32_t> node_of(t.n, -1), nodes;
    for (int64_t v = 0; v < t.n; ++v) if (is_node[v]) { node_of[v] = (int32_t)nodes.size(); nodes.push_back((int32_t)v); }
    const int64_t N = (int64_t)nodes.size();
    // ---- a11: pair loop + addSubEdge  (PhasingGraph.cpp:848-888, :25-70) -> dense [N][A][4]
    std::vector<float> edge((size_t)N * A * 4, 0.0f);
    int64_t n_pairs = 0;
    for (auto &kv : merged) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end(), [](const SortKey &x, const SortKey &y) { return x.pos < y.pos; });   // Util.cpp:3-5
        for (size_t i = 0; i + 1 < v.size(); ++i) {
            for (size_t j = i + 1; j < v.size() && j <= i + (size_t)A; ++j) {
                ++n_pairs;
                const int d = node_of[v[j].var] - node_of[v[i].var];
                if (d < 1 || d > A) continue;                      // never queried by edgeConnectResult
                float &cell = edge[((size_t)node_of[v[i].var] * A + (d - 1)) * 4 + (v[i].allele << 1 | v[j].allele)];
                if (v[i].quality >= P.base_quality && v[j].quality >= P.base_quality) cell++;
                else cell = cell + P.edge_weight;                  // float = float + double -> rounded once
            }
        }
    }
    if (D) D->n_pairs = n_pairs;
    // ---- a12/a13: edgeConnectResult + findBestEdgePair + Onelongcase  (PhasingGraph.cpp:166-228,251-283,286-474)
    struct Vote { int src; float para, cross, weight; int hap; double esr; };
    std::vector<float> h1(N, 0.f), h2(N, 0.f);
    std::vector<std::vector<Vote>> votes(N);
    std::vector<int8_t> hp(N, 0); std::vector<int32_t> block(N, -1);
    int blockStart = -1; int64_t lastConnect = -1;
    std::map<int, std::vector<int>> blocks;
    for (int64_t i = 0; i + 1 < N; ++i) {                           // last node never processed (:308-311)
        const int cp = t.pos[nodes[i]], np = t.pos[nodes[i + 1]];
        if (std::abs(np - cp) > P.distance) continue;
        float a1 = h1[i], a2 = h2[i];
        {   // Onelongcase
            int counter = 0; float s1 = 0, s2 = 0;
            for (auto &vt : votes[i]) {
                if ((vt.para + vt.cross) <= 1) counter++;
                else if (vt.esr < 0.2 && vt.weight >= 1 && vtype[nodes[vt.src]] != 3) { if (vt.hap == 1) s1 += vt.weight; else if (vt.hap == 2) s2 += vt.weight; }
            }
            if (!(counter <= 3 || (s1 == 0 && s2 == 0))) { a1 = s1; a2 = s2; }
        }
        if (a1 == a2) {
            if (i < lastConnect) continue;
            blockStart = (int)i; blocks[blockStart].push_back((int)i); hp[i] = 1;
        } else { hp[i] = (a1 > a2) ? 1 : 2; blocks[blockStart].push_back((int)i); }
        block[i] = blockStart;
        for (int k = 0; k < A && i + 1 + k < N; ++k) {
            const int64_t j = i + 1 + k;
            const float *c = &edge[((size_t)i * A + k) * 4];
            const float rr = c[0], ra = c[1], ar = c[2], aa = c[3];
            Vote vt; vt.src = (int)i; vt.weight = 1; vt.hap = 0;
            int dir = -1;
            double esr = (double)std::min(rr + aa, ar + ra) / (double)std::max(rr + aa, ar + ra);
            if (rr + aa > ra + ar) dir = 1; else if (rr + aa < ra + ar) dir = 2;
            double thr = P.edge_threshold;
            {   // :197-202 an edge between a SNP and a MOD row: 0.3, and nothing passes when the four cells sum to less than one read
                const int ti = vtype[nodes[i]], tj = vtype[nodes[j]];
                if ((ti == 0 && tj == 2) || (ti == 2 && tj == 0)) { thr = 0.3; if ((rr + ra + ar + aa) < 1) thr = -1; }
            }
            if (esr > thr) dir = -1;
            // the reference's `else if` hangs off `if(debug)` (:210-217, debug is always false): independent of esr>thr
            if ((esr <= 0.1 && (rr + aa + ra + ar) >= 1) || ((rr + aa) < 1 && (ra + ar) >= 1) || ((rr + aa) >= 1 && (ra + ar) < 1)) vt.weight = 20;
            vt.para = rr + aa; vt.cross = ra + ar; vt.esr = esr;
            if (vtype[nodes[i]] == 4) vt.weight = 0.1;
            if (dir != -1) {
                int th = (hp[i] == 1) ? dir : (dir == 1 ? 2 : 1);
                if (th == 1) h1[j] += vt.weight; else h2[j] += vt.weight;
                vt.hap = th; votes[j].push_back(vt);
                lastConnect = j;
            }
        }
    }
    std::vector<int32_t> ps(N, 0); std::vector<int8_t> refhap(N, -1);
    for (auto &kv : blocks) {
        if (kv.second.size() <= 1) continue;
        for (int m : kv.second) { ps[m] = t.pos[nodes[kv.first]] + 1; refhap[m] = (hp[m] == hp[kv.second.front()]) ? 0 : 1; }
    }
    if (D) {
        D->n_nodes = N;
        for (int64_t i = 0; i < N && i < D->node_capacity; ++i) {
            if (D->node_var) D->node_var[i] = nodes[i];
            if (D->node_hp) D->node_hp[i] = hp[i];
            if (D->node_block) D->node_block[i] = block[i];
        }
        if (D->edge && N <= D->node_capacity) std::memcpy(D->edge, edge.data(), edge.size() * sizeof(float));
    }
    // ---- a14: readCorrection  (PhasingGraph.cpp:891-1029)
    std::vector<double> cnt((size_t)N * 4, 0.0);                   // [node][hp][allele]
    for (auto &a : alns) {
        double rc = 0, ac = 0;
        for (auto &o : a.obs) {
            int nd = node_of[o.var];
            if (ps[nd] != 0) {
                int h = (o.allele == 0) ? refhap[nd] : 1 - refhap[nd];
                int ty = vtype[o.var];
                if (ty == 0 || ty == 1) { if (h == 0) rc++; else ac++; }
                else if (ty == 3 || ty == 4) { if (h == 0) rc += 0.1; else ac += 0.1; }
            }
        }
        if (std::max(rc, ac) / (rc + ac) > P.read_confidence && (rc + ac) > 1) {
            int bh = (rc > ac) ? 0 : 1;
            for (auto &o : a.obs) cnt[(size_t)node_of[o.var] * 4 + bh * 2 + o.allele]++;
        }
    }
    // ---- a15: final genotype + exportResult  (PhasingGraph.cpp:983-1026,1049-1077)
    for (int64_t i = 0; i < N; ++i) {
        const double *c = &cnt[(size_t)i * 4];
        double r1 = c[0] + c[3], r2 = c[2] + c[1];
        double conf = std::max(r1, r2) / (r1 + r2);
        int g = -1;
        if (conf > P.snp_confidence) { if (r1 > r2) g = 0; else if (r1 < r2) g = 1; }
        if (g != -1 && ps[i] != 0) {
            const int32_t u = nodes[i];
            if (!E.x) { out->phase_set[u] = ps[i]; out->gt[u] = (uint8_t)g; }
            else if (E.snp_of_u[u] >= 0) { out->phase_set[E.snp_of_u[u]] = ps[i]; out->gt[E.snp_of_u[u]] = (uint8_t)g; }
            else {
                const auto sv = std::lower_bound(E.u_of_sv.begin(), E.u_of_sv.end(), u);
                if (sv != E.u_of_sv.end() && *sv == u) { if (out_sv) { out_sv->phase_set[sv - E.u_of_sv.begin()] = ps[i]; out_sv->gt[sv - E.u_of_sv.begin()] = (uint8_t)g; } }
                else { const auto md = std::lower_bound(E.u_of_mod.begin(), E.u_of_mod.end(), u); if (out_mod) { out_mod->phase_set[md - E.u_of_mod.begin()] = ps[i]; out_mod->gt[md - E.u_of_mod.begin()] = (uint8_t)g; } }
            }
        }
    }
    return 0;
}

int oracle_phase(const lps_params *Pp, const lps_variant_table *tp, const char *ref, int64_t ref_len_in,
                 const lps_read_batch *bp, lps_phase_result *out, oracle_dumps *D) {
    return oracle_phase_x(Pp, tp, nullptr, ref, ref_len_in, bp, out, nullptr, nullptr, D);
}


// ------------------------------------------------------------------------------------------------ haplotag
// Per-read scoring loop of `haplotag` (germline): filter cascade of ChromosomeProcessor::processSingleChrom
// (src/haplotag/HaplotagParsingBam.cpp:453-486), CigarParser::parsingCigar (:541-647), judgeSnpHap /
// judgeDeletionHap (src/haplotag/HaplotagStrategy.cpp:20-209) and judgeReadHap (:243-300).
// The table holds the phased-het rows of the normal VCF (HaplotagVcfParser.cpp:304-400): HP1 = ALT when hp1_is_alt.
// votes_h1 / votes_h2 (may be NULL): per alignment, what judgeSVHap (src/haplotag/HaplotagStrategy.cpp:220-226) adds from the phased SV / MOD files
int oracle_haplotag_v(const lps_params *Pp, const lps_variant_table *tp, const char *ref, int64_t ref_len_in,
                      const lps_read_batch *bp, const int32_t *votes_h1, const int32_t *votes_h2, lps_haplotag_result *out) {
    const lps_params &P = *Pp; const lps_variant_table &t = *tp; const lps_read_batch &b = *bp;
    Table T; T.t = &t; T.ref = ref;
    const int32_t last_pos = t.n ? t.pos[t.n - 1] : -1;
    T.ref_len = std::min<int64_t>(ref_len_in, (int64_t)last_pos + 6);
    for (int64_t r = 0; r < b.n_reads; ++r) {
        out->status[r] = 0; out->hp1[r] = 0; out->hp2[r] = 0; out->n_ps[r] = 0; out->ps_min[r] = 0; out->hp[r] = 0; out->pq[r] = 0; out->ps[r] = 0;
        const int fl = b.flag[r];
        if (b.mapq[r] < P.mapping_quality) { out->status[r] = 1; continue; }
        if (fl & 0x4) { out->status[r] = 2; continue; }
        if (fl & 0x100) { out->status[r] = 3; continue; }
        if ((fl & 0x800) && !P.tag_supplementary) { out->status[r] = 4; continue; }
        if (t.n == 0) { out->status[r] = 5; continue; }
        if (!(b.ref_start[r] <= last_pos)) { out->status[r] = 6; continue; }
        // ---- parsingCigar
        const uint32_t *cig = b.cigar + b.cigar_off[r];
        const int n_cig = (int)(b.cigar_off[r + 1] - b.cigar_off[r]);
        const uint8_t *seq = b.seq + b.seq_off[r];
        const int64_t lq = b.l_qseq[r];
        int64_t ref_pos = b.ref_start[r], query_pos = 0;
        int64_t cur = std::lower_bound(t.pos, t.pos + t.n, (int32_t)ref_pos) - t.pos;
        int h1 = 0, h2 = 0; std::map<int, int> countPS;
        auto vote_allele = [&](int64_t v, bool alt) { if ((t.hp1_is_alt[v] != 0) == alt) h1++; else h2++; };
        if (cur < t.n) for (int i = 0; i < n_cig; ++i) {
            const int op = cig[i] & 15; const int64_t len = cig[i] >> 4;
            while (cur < t.n && t.pos[cur] < ref_pos) ++cur;
            if (op == 0 || op == 7 || op == 8) {
                while (cur < t.n && t.pos[cur] < ref_pos + len) {
                    const int64_t off = t.pos[cur] - ref_pos, qi = query_pos + off;
                    const int rl = t.ref_len[cur], al = t.alt_len[cur];
                    if (rl == 1 && al == 1) {                                          // judgeSnpHap SNP (:36-65)
                        const char base = qi < lq ? seq_base(seq, qi) : 'N';           // reference reads out of bounds here
                        if (base == (char)t.ref0[cur] || base == (char)t.alt0[cur]) { vote_allele(cur, base == (char)t.alt0[cur]); countPS[t.phase_set[cur]]++; }
                    } else if (rl == 1 && al > 1 && i + 1 < n_cig) {                   // insertion (:68-96)
                        const bool has = (ref_pos + len - 1 == t.pos[cur]) && (cig[i + 1] & 15) == 1;
                        vote_allele(cur, has); countPS[t.phase_set[cur]]++;
                    } else if (rl > 1 && al == 1 && i + 1 < n_cig) {                   // deletion (:98-129): votes for the LONG allele
                        const bool has = (ref_pos + len - 1 == t.pos[cur]) && (cig[i + 1] & 15) == 2;
                        vote_allele(cur, !has); countPS[t.phase_set[cur]]++;
                    }
                    ++cur;
                }
                query_pos += len; ref_pos += len;
            } else if (op == 1) query_pos += len;
            else if (op == 2) {
                bool judged = false;
                while (cur < t.n && t.pos[cur] < ref_pos + len) {
                    if (!judged) {                                                     // processDeletionOperation: once per D op
                        judged = true;
                        const int64_t p = t.pos[cur];
                        if (!(ref_pos + len + 1 == p) && p >= ref_pos && p < ref_pos + len && homopolymer_length(p, T.ref, T.ref_len) >= 3) {
                            const int rl = t.ref_len[cur], al = t.alt_len[cur];
                            if (rl == 1 && al == 1) {                                  // judgeDeletionHap SNP (:175-189)
                                const char base = query_pos < lq ? seq_base(seq, query_pos) : 'N';
                                if (base == (char)t.ref0[cur] || base == (char)t.alt0[cur]) vote_allele(cur, base == (char)t.alt0[cur]);
                                countPS[t.phase_set[cur]]++;
                            } else if (rl > 1 && al == 1) { vote_allele(cur, false); countPS[t.phase_set[cur]]++; }   // (:192-206)
                        }
                    }
                    ++cur;
                }
                ref_pos += len;
            } else if (op == 3) ref_pos += len;
            else if (op == 4) query_pos += len;
            else if (op == 5 || op == 6) {}
            else return -2;
        }
        if (votes_h1 && votes_h2) { h1 += votes_h1[r]; h2 += votes_h2[r]; }                  // judgeSVHap, HaplotagProcess.cpp:401
        out->hp1[r] = h1; out->hp2[r] = h2;
        out->n_ps[r] = (uint8_t)std::min<size_t>(countPS.size(), 255); out->ps_min[r] = countPS.empty() ? 0 : countPS.begin()->first;
        // ---- judgeReadHap
        double mn, mx; int hp = 0, pq = 0;
        if (h1 > h2) { mn = h2; mx = h1; } else { mn = h1; mx = h2; }
        if (mx / (mx + mn) < P.percentage_threshold) pq = 0;
        else { if (h1 > h2) hp = 1; if (h1 < h2) hp = 2; }
        if (mx == 0) pq = 0; else if (mx == mx + mn) pq = 40; else pq = -10 * (std::log10((double)mn / double(mx + mn)));
        if (countPS.size() > 1) hp = 0;
        // (a read tagged by SV / MOD votes alone has no PS: the reference reads begin() of the empty map - the node count, 0, with libstdc++)
        out->hp[r] = (uint8_t)hp; out->pq[r] = pq; out->ps[r] = (hp && !countPS.empty()) ? countPS.begin()->first : 0;
    }
    return 0;
}

int oracle_haplotag(const lps_params *Pp, const lps_variant_table *tp, const char *ref, int64_t ref_len_in,
                    const lps_read_batch *bp, lps_haplotag_result *out) {
    return oracle_haplotag_v(Pp, tp, ref, ref_len_in, bp, nullptr, nullptr, out);
}


// ------------------------------------------------------------------------------------------------ somatic tagging (a22)
// SomaticHaplotagChrProcessor::judgeHaplotype (src/somatic_haplotag/SomaticHaplotagProcess.cpp:310-459),
// SomaticHaplotagCigarParser (:557-579), SomaticJudgeHapStrategy::judgeSomaticSnpHap / judgeNormalSnpHap
// (src/haplotag/HaplotagStrategy.cpp:315-435), SomaticHaplotagStrategy::judgeTumorOnlySnpHap (:653-668),
// judgeSomaticReadHap (:452-602), inheritHaplotype (SomaticHaplotagProcess.cpp:461-527).
int oracle_somatic_tag(const lps_params *Pp, const lps_variant_table *tp, const lps_read_batch *bp, lps_somatic_tag_result *out) {
    const lps_params &P = *Pp; const lps_variant_table &t = *tp; const lps_read_batch &b = *bp;
    const int32_t last_pos = t.n ? t.pos[t.n - 1] : -1;
    for (int64_t r = 0; r < b.n_reads; ++r) {
        out->status[r] = 0; out->hp1[r] = out->hp2[r] = out->hp3[r] = 0; out->derive_h1[r] = out->derive_h2[r] = 0;
        out->n_ps[r] = 0; out->ps_min[r] = 0; out->hp[r] = 0; out->pq[r] = 0; out->ps[r] = -1;
        const int fl = b.flag[r];
        if (b.mapq[r] < P.mapping_quality) { out->status[r] = 1; continue; }
        if (fl & 0x4) { out->status[r] = 2; continue; }
        if (fl & 0x100) { out->status[r] = 3; continue; }
        if ((fl & 0x800) && !P.tag_supplementary) { out->status[r] = 4; continue; }
        if (t.n == 0) { out->status[r] = 5; continue; }
        if (!(b.ref_start[r] <= last_pos)) { out->status[r] = 6; continue; }
        const uint32_t *cig = b.cigar + b.cigar_off[r];
        const int n_cig = (int)(b.cigar_off[r + 1] - b.cigar_off[r]);
        const uint8_t *seq = b.seq + b.seq_off[r];
        const int64_t lq = b.l_qseq[r];
        int64_t ref_pos = b.ref_start[r], query_pos = 0;
        int64_t cur = std::lower_bound(t.pos, t.pos + t.n, (int32_t)ref_pos) - t.pos;
        int h1 = 0, h2 = 0, h3 = 0, d1 = 0, d2 = 0; std::map<int, int> norPS;
        if (cur < t.n) for (int i = 0; i < n_cig; ++i) {
            const int op = cig[i] & 15; const int64_t len = cig[i] >> 4;
            while (cur < t.n && t.pos[cur] < ref_pos) ++cur;
            if (op == 0 || op == 7 || op == 8) {
                while (cur < t.n && t.pos[cur] < ref_pos + len) {
                    const int64_t qi = query_pos + (t.pos[cur] - ref_pos);
                    const char base = qi < lq ? seq_base(seq, qi) : 'N';
                    const int rl = t.ref_len[cur], al = t.alt_len[cur];
                    const bool snp = rl == 1 && al == 1, ins = rl == 1 && al > 1, del = rl > 1 && al == 1;
                    bool isAlt = false;                                               // IsAltIndel (HaplotagParsingBam.cpp:650-670)
                    if (snp) isAlt = base == (char)t.alt0[cur];
                    else if (ins && i + 1 < n_cig) isAlt = (ref_pos + len - 1 == t.pos[cur]) && (cig[i + 1] & 15) == 1;
                    else if (del && i + 1 < n_cig) isAlt = (ref_pos + len - 1 == t.pos[cur]) && (cig[i + 1] & 15) == 2;
                    const int role = t.somatic_role[cur];
                    if (role == 0) {                                                  // judgeNormalSnpHap
                        bool counted = false, alt = false;
                        if (snp) { if (base == (char)t.ref0[cur] || base == (char)t.alt0[cur]) { counted = true; alt = base == (char)t.alt0[cur]; } }
                        else if (ins || del) { counted = true; alt = isAlt; }         // base := isAlt ? Alt : Ref (:330-337)
                        if (counted) { if ((t.hp1_is_alt[cur] != 0) == alt) h1++; else h2++; norPS[t.phase_set[cur]]++; }
                    } else if (role == 1) {                                           // SomaticHaplotagStrategy::judgeTumorOnlySnpHap
                        bool h3v = false;
                        if (snp) h3v = base == (char)t.alt0[cur];                     // guarded by Ref==base||Alt==base (:360-361)
                        else if (ins || del) h3v = isAlt;
                        if (h3v) { h3++; if (t.derive_hp[cur] == 1) d1++; else if (t.derive_hp[cur] == 2) d2++; }
                    }
                    ++cur;
                }
                query_pos += len; ref_pos += len;
            } else if (op == 1) query_pos += len;
            else if (op == 2) { while (cur < t.n && t.pos[cur] < ref_pos + len) ++cur; ref_pos += len; }   // only statistics in the reference
            else if (op == 3) ref_pos += len;
            else if (op == 4) query_pos += len;
            else if (op == 5 || op == 6) {}
            else return -2;
        }
        out->hp1[r] = h1; out->hp2[r] = h2; out->hp3[r] = h3; out->derive_h1[r] = d1; out->derive_h2[r] = d2;
        out->n_ps[r] = (uint8_t)std::min<size_t>(norPS.size(), 255); out->ps_min[r] = norPS.empty() ? 0 : norPS.begin()->first;
        // ---- judgeSomaticReadHap (hpCount[4] is never incremented by the tagging pass)
        double tMin, tMax, nMin, nMax; int maxT, maxN;
        const int h4 = 0;
        if (h3 > h4) { tMin = h4; tMax = h3; maxT = 3; } else { tMin = h3; tMax = h4; maxT = 4; }
        if (h1 > h2) { nMin = h2; nMax = h1; maxN = 1; } else { nMin = h1; nMax = h2; maxN = 2; }
        const double tumSim = (tMax == 0) ? 0.0 : tMax / (tMax + tMin);
        const double norSim = (nMax == 0) ? 0.0 : nMax / (nMax + nMin);
        int hp = 0, pq = 0;
        const double thr = P.percentage_threshold;
        if (tMax != 0) {
            if (tumSim >= thr) {
                if (norSim >= thr) hp = (maxT == 3) ? (maxN == 1 ? 5 : 7) : (maxN == 1 ? 6 : 8);
                else hp = (maxT == 3) ? 3 : 4;
            } else pq = 0;
        } else if (nMax != 0) { if (norSim >= thr) hp = maxN; else pq = 0; }
        if (norPS.size() > 1) hp = 0;
        if (nMax == 0 && tMax == 0) pq = 0;
        else if (tMax != 0) { if (tMax == tMax + tMin) pq = 40; else pq = -10 * (std::log10((double)tMin / double(tMax + tMin))); }
        else if (nMax != 0) { if (nMax == nMax + nMin) pq = 40; else pq = -10 * (std::log10((double)nMin / double(nMax + nMin))); }
        // ---- inheritHaplotype
        if (hp == 3) {
            int mx, mn, mh;
            if (d1 > d2) { mx = d1; mn = d2; mh = 1; } else { mx = d2; mn = d1; mh = 2; }
            const float sim = (mx == 0) ? 0.0f : ((float)mx / ((float)mx + (float)mn));
            if (sim >= P.percentage_threshold) hp = (mh == 1) ? 5 : 7;
        }
        int ps = -1;
        if (hp != 0) {
            if (hp != 1 && hp != 2) { if (!norPS.empty()) ps = norPS.begin()->first; }
            else ps = norPS.begin()->first;
        }
        out->hp[r] = (uint8_t)hp; out->pq[r] = pq; out->ps[r] = ps;
    }
    return 0;
}


// ------------------------------------------------------------------------------------------------ somatic extraction, normal BAM (a20)
// ExtractNorDataChrProcessor::processRead (src/somatic_haplotag/SomaticVarCaller.cpp:123-174), ExtractNorDataCigarParser
// (:227-293), CigarParser::countBaseNucleotide / countDeletionBase (src/haplotag/HaplotagParsingBam.cpp:682-729), germline votes by
// GermlineHaplotagStrategy (HaplotagStrategy.cpp:20-209) gated by MAPQ, judgeReadHap (:243-300).  The extraction passes run
// with ParsingBamControl::mappingQualityFilter == false (HaplotagParsingBam.h:57): low-MAPQ reads are NOT skipped.
int oracle_somatic_extract_normal(const lps_params *Pp, const lps_variant_table *tp, const char *ref, int64_t ref_len_in,
                                  const lps_read_batch *bp, lps_site_counters *out) {
    const lps_params &P = *Pp; const lps_variant_table &t = *tp; const lps_read_batch &b = *bp;
    Table T; T.t = &t; T.ref = ref;
    const int32_t last_pos = t.n ? t.pos[t.n - 1] : -1;
    T.ref_len = std::min<int64_t>(ref_len_in, (int64_t)last_pos + 6);
    std::memset(out->counters, 0, (size_t)t.n * LPS_SITE_COUNTERS * sizeof(int32_t));
    auto C = [&](int64_t v, int k) -> int32_t & { return out->counters[v * LPS_SITE_COUNTERS + k]; };
    for (int64_t r = 0; r < b.n_reads; ++r) {
        if (out->read_hp) out->read_hp[r] = 0;
        const int fl = b.flag[r];
        if ((fl & 0x4) || (fl & 0x100) || ((fl & 0x800) && !P.tag_supplementary) || t.n == 0 || !(b.ref_start[r] <= last_pos)) continue;
        const bool mq_ok = b.mapq[r] >= P.mapping_quality;
        const uint32_t *cig = b.cigar + b.cigar_off[r];
        const int n_cig = (int)(b.cigar_off[r + 1] - b.cigar_off[r]);
        const uint8_t *seq = b.seq + b.seq_off[r];
        const int64_t lq = b.l_qseq[r];
        int64_t ref_pos = b.ref_start[r], query_pos = 0;
        int64_t cur = std::lower_bound(t.pos, t.pos + t.n, (int32_t)ref_pos) - t.pos;
        int h1 = 0, h2 = 0; std::map<int, int> countPS; std::vector<int64_t> touched;
        auto vote_allele = [&](int64_t v, bool alt) { if ((t.hp1_is_alt[v] != 0) == alt) h1++; else h2++; };
        if (cur < t.n) for (int i = 0; i < n_cig; ++i) {
            const int op = cig[i] & 15; const int64_t len = cig[i] >> 4;
            while (cur < t.n && t.pos[cur] < ref_pos) ++cur;
            if (op == 0 || op == 7 || op == 8) {
                while (cur < t.n && t.pos[cur] < ref_pos + len) {
                    const int64_t qi = query_pos + (t.pos[cur] - ref_pos);
                    const char base = qi < lq ? seq_base(seq, qi) : 'N';
                    const int rl = t.ref_len[cur], al = t.alt_len[cur];
                    const bool snp = rl == 1 && al == 1, ins = rl == 1 && al > 1, del = rl > 1 && al == 1;
                    bool isAlt = false;
                    if (snp) isAlt = base == (char)t.alt0[cur];
                    else if (ins && i + 1 < n_cig) isAlt = (ref_pos + len - 1 == t.pos[cur]) && (cig[i + 1] & 15) == 1;
                    else if (del && i + 1 < n_cig) isAlt = (ref_pos + len - 1 == t.pos[cur]) && (cig[i + 1] & 15) == 2;
                    const int tk = t.tumor_kind[cur];
                    if (tk >= 1 && tk <= 3) {                                          // processMatchOperation :232-243 + countBaseNucleotide
                        touched.push_back(cur);
                        const int bi = base == 'A' ? LPS_SC_A : base == 'C' ? LPS_SC_C : base == 'G' ? LPS_SC_G : base == 'T' ? LPS_SC_T : LPS_SC_UNKNOWN;
                        if (mq_ok) { C(cur, bi + (LPS_SC_MPQ_A - LPS_SC_A))++; if (isAlt) C(cur, LPS_SC_MPQ_ALT)++; C(cur, LPS_SC_MPQ_DEPTH)++; }
                        C(cur, bi)++;
                        if (isAlt) { if (tk == 3) C(cur, LPS_SC_DEL)++; C(cur, LPS_SC_ALT)++; }
                        C(cur, LPS_SC_DEPTH)++;
                    }
                    if (mq_ok && t.somatic_role[cur] == 0) {                           // germline judgeSnpHap on the NORMAL row (:255-261)
                        if (snp) { if (base == (char)t.ref0[cur] || base == (char)t.alt0[cur]) { vote_allele(cur, base == (char)t.alt0[cur]); countPS[t.phase_set[cur]]++; } }
                        else if (ins && i + 1 < n_cig) { vote_allele(cur, isAlt); countPS[t.phase_set[cur]]++; }
                        else if (del && i + 1 < n_cig) { vote_allele(cur, !isAlt); countPS[t.phase_set[cur]]++; }
                    }
                    ++cur;
                }
                query_pos += len; ref_pos += len;
            } else if (op == 1) query_pos += len;
            else if (op == 2) {
                bool judged = false;
                while (cur < t.n && t.pos[cur] < ref_pos + len) {
                    const int tk = t.tumor_kind[cur];
                    if (tk != 0) {                                                     // processDeletionOperation :265-282
                        touched.push_back(cur);
                        if (tk == 1) { C(cur, LPS_SC_DEL)++; C(cur, LPS_SC_DEPTH)++; }
                        else if (tk == 3) { C(cur, LPS_SC_ALT)++; C(cur, LPS_SC_DEL)++; C(cur, LPS_SC_DEPTH)++; }
                    }
                    if (mq_ok && t.somatic_role[cur] == 0 && !judged) {                // :285-291, then judgeDeletionHap
                        judged = true;
                        const int64_t p = t.pos[cur];
                        if (homopolymer_length(p, T.ref, T.ref_len) >= 3) {
                            const int rl = t.ref_len[cur], al = t.alt_len[cur];
                            if (rl == 1 && al == 1) {
                                const char base = query_pos < lq ? seq_base(seq, query_pos) : 'N';
                                if (base == (char)t.ref0[cur] || base == (char)t.alt0[cur]) vote_allele(cur, base == (char)t.alt0[cur]);
                                countPS[t.phase_set[cur]]++;
                            } else if (rl > 1 && al == 1) { vote_allele(cur, false); countPS[t.phase_set[cur]]++; }
                        }
                    }
                    ++cur;
                }
                ref_pos += len;
            } else if (op == 3) ref_pos += len;
            else if (op == 4) query_pos += len;
            else if (op == 5 || op == 6) {}
            else return -2;
        }
        double mn, mx; int hp = 0;
        if (h1 > h2) { mn = h2; mx = h1; } else { mn = h1; mx = h2; }
        if (!(mx / (mx + mn) < P.percentage_threshold)) { if (h1 > h2) hp = 1; if (h1 < h2) hp = 2; }
        if (countPS.size() > 1) hp = 0;
        if (out->read_hp) out->read_hp[r] = (uint8_t)hp;
        for (int64_t v : touched) C(v, LPS_SC_READHP_UNTAG + hp)++;
    }
    return 0;
}


// ------------------------------------------------------------------------------------------------ somatic extraction, tumor BAM (a21)
namespace {
// processCigarOperation (src/somatic_haplotag/SomaticVarCaller.cpp:627-652).  NB the reference's enum has CIGAR_N == 6 (HaplotagType.h:29).
bool win_next_op(const uint32_t *cig, int &idx, int end, int dir, int &remaining, int &readPos, int &refPos, int &op) {
    idx += dir;
    while (idx < end && idx >= 0) {
        op = cig[idx] & 15; const int len = cig[idx] >> 4;
        if (op == 0 || op == 3 || op == 6 || op == 7 || op == 8) { remaining += len; return true; }
        else if (op == 1) readPos += len * dir;
        else if (op == 2) refPos += len * dir;
        else return false;
        idx += dir;
    }
    return false;
}
// getOrderWindowsDiffRef (:654-685)
void win_dir(const uint32_t *cig, int idx, int n_cig, const uint8_t *seq, int readLen, const char *ref, int refLen, int readPos, int remaining,
             int refPos, int dir, std::vector<std::pair<int, char>> &out) {
    int op = cig[idx] & 15;
    for (int i = 1; i <= 100; ++i) {
        remaining--;
        if (remaining == 0 || remaining == -1) { if (!win_next_op(cig, idx, n_cig, dir, remaining, readPos, refPos, op)) return; }
        if (op == 2 || op == 1 || op == 3 || op == 6 || op == 8) continue;
        readPos += dir; refPos += dir;
        if (readPos > readLen || refPos > refLen || readPos < 0 || refPos < 0) return;
        const char rb = readPos < readLen ? seq_base(seq, readPos) : '\0';     // reference reads one past the end here
        const char fb = refPos < refLen ? ref[refPos] : '\0';
        if (rb != fb) out.emplace_back(i * dir, rb);
    }
}
}  // namespace

int oracle_somatic_extract_tumor(const lps_params *Pp, const lps_variant_table *tp, const char *ref, int64_t ref_len_in,
                                 const lps_read_batch *bp, lps_tumor_extract_result *out) {
    const lps_params &P = *Pp; const lps_variant_table &t = *tp; const lps_read_batch &b = *bp;
    const int32_t last_pos = t.n ? t.pos[t.n - 1] : -1;
    const int refLen = (int)std::min<int64_t>(ref_len_in, (int64_t)last_pos + 6);
    std::memset(out->site, 0, (size_t)t.n * LPS_TSITE_COUNTERS * sizeof(int32_t));
    auto C = [&](int64_t v, int k) -> int32_t & { return out->site[v * LPS_TSITE_COUNTERS + k]; };
    int64_t n_pairs = 0, n_win = 0;
    for (int64_t r = 0; r < b.n_reads; ++r) {
        out->status[r] = 0; out->hp1[r] = out->hp2[r] = out->hp3[r] = 0; out->hp[r] = 0; out->n_ps[r] = 0; out->ps_min[r] = 0;
        out->end_pos[r] = 0; out->read_len[r] = 0; out->has_site[r] = 0;
        const int fl = b.flag[r];
        if (fl & 0x4) { out->status[r] = 2; continue; }
        if (fl & 0x100) { out->status[r] = 3; continue; }
        if ((fl & 0x800) && !P.tag_supplementary) { out->status[r] = 4; continue; }
        if (t.n == 0) { out->status[r] = 5; continue; }
        if (!(b.ref_start[r] <= last_pos)) { out->status[r] = 6; continue; }
        const bool mq_ok = b.mapq[r] >= P.mapping_quality;
        const uint32_t *cig = b.cigar + b.cigar_off[r];
        const int n_cig = (int)(b.cigar_off[r + 1] - b.cigar_off[r]);
        const uint8_t *seq = b.seq + b.seq_off[r];
        const int lq = b.l_qseq[r];
        int ref_pos = b.ref_start[r], query_pos = 0;
        int64_t cur = std::lower_bound(t.pos, t.pos + t.n, (int32_t)ref_pos) - t.pos;
        int h1 = 0, h2 = 0, h3 = 0; std::map<int, int> norPS;
        std::vector<int64_t> h3_sites; std::vector<std::pair<int64_t, int>> tum_sites;   // (row, baseHP)
        std::vector<std::pair<int, char>> win;
        bool walked = cur < t.n;
        if (walked) for (int i = 0; i < n_cig; ++i) {
            const int op = cig[i] & 15; const int len = cig[i] >> 4;
            while (cur < t.n && t.pos[cur] < ref_pos) ++cur;
            if (op == 0 || op == 7 || op == 8) {
                while (cur < t.n && t.pos[cur] < ref_pos + len) {
                    const int off = t.pos[cur] - ref_pos, qi = query_pos + off;
                    const char base = qi < lq ? seq_base(seq, qi) : 'N';
                    const int rl = t.ref_len[cur], al = t.alt_len[cur];
                    const bool snp = rl == 1 && al == 1, ins = rl == 1 && al > 1, del = rl > 1 && al == 1;
                    bool isAlt = false;
                    if (snp) isAlt = base == (char)t.alt0[cur];
                    else if (ins && i + 1 < n_cig) isAlt = (ref_pos + len - 1 == t.pos[cur]) && (cig[i + 1] & 15) == 1;
                    else if (del && i + 1 < n_cig) isAlt = (ref_pos + len - 1 == t.pos[cur]) && (cig[i + 1] & 15) == 2;
                    const int tk = t.tumor_kind[cur]; const int role = t.somatic_role[cur];
                    // getWindowsDiffRef (:687-710) - computed for every variant, used only below
                    win.clear();
                    {
                        const int fwd = (len - off > 0) ? len - off : 0, rev = off > 0 ? off : 0;
                        win_dir(cig, i, n_cig, seq, lq, ref, refLen, query_pos + off, rev, t.pos[cur], -1, win);
                        win_dir(cig, i, n_cig, seq, lq, ref, refLen, query_pos + off, fwd, t.pos[cur], +1, win);
                    }
                    int baseHP = 0;
                    if (mq_ok) {                                                        // judgeSomaticSnpHap (HaplotagStrategy.cpp:315-389)
                        if (role == 0) {
                            bool counted = false, alt = false;
                            if (snp) { if (base == (char)t.ref0[cur] || base == (char)t.alt0[cur]) { counted = true; alt = base == (char)t.alt0[cur]; } }
                            else if (ins || del) { counted = true; alt = isAlt; }
                            if (counted) { if ((t.hp1_is_alt[cur] != 0) == alt) { h1++; baseHP = 1; } else { h2++; baseHP = 2; } norPS[t.phase_set[cur]]++; }
                        } else if (tk != 0) {                                           // tumor-only row: H3 when the read shows the tumor ALT (:617-638)
                            bool h3v = false;
                            if (snp) h3v = base == (char)t.alt0[cur]; else if (ins || del) h3v = isAlt;
                            if (h3v) { h3++; baseHP = 3; h3_sites.push_back(cur); }
                        }
                        if (tk != 0) tum_sites.emplace_back(cur, baseHP);               // tumorSnpPosVec (:722-724)
                    }
                    if (tk >= 1 && tk <= 3) {                                           // :728-741
                        if (tk != 1 || base == (char)t.ref0[cur] || base == (char)t.alt0[cur]) {
                            C(cur, 39 + (isAlt ? 1 : 0))++;
                            for (auto &w : win) {
                                if (n_win < out->win_capacity) { out->win_site[n_win] = (int32_t)cur; out->win_allele[n_win] = isAlt; out->win_offset[n_win] = (int16_t)w.first; out->win_base[n_win] = (uint8_t)w.second; }
                                ++n_win;
                            }
                        }
                        const int bi = base == 'A' ? LPS_SC_A : base == 'C' ? LPS_SC_C : base == 'G' ? LPS_SC_G : base == 'T' ? LPS_SC_T : LPS_SC_UNKNOWN;
                        if (mq_ok) { C(cur, bi + (LPS_SC_MPQ_A - LPS_SC_A))++; if (isAlt) C(cur, LPS_SC_MPQ_ALT)++; C(cur, LPS_SC_MPQ_DEPTH)++; }
                        C(cur, bi)++;
                        if (isAlt) { if (tk == 3) C(cur, LPS_SC_DEL)++; C(cur, LPS_SC_ALT)++; }
                        C(cur, LPS_SC_DEPTH)++;
                    }
                    ++cur;
                }
                query_pos += len; ref_pos += len;
            } else if (op == 1) query_pos += len;
            else if (op == 2) {
                while (cur < t.n && t.pos[cur] < ref_pos + len) {                       // processDeletionOperation :743-759
                    const int tk = t.tumor_kind[cur];
                    if (tk == 1) { C(cur, LPS_SC_DEL)++; C(cur, LPS_SC_DEPTH)++; }
                    else if (tk == 3) { C(cur, LPS_SC_ALT)++; C(cur, LPS_SC_DEL)++; C(cur, LPS_SC_DEPTH)++; }
                    ++cur;
                }
                ref_pos += len;
            } else if (op == 3) ref_pos += len;
            else if (op == 4) query_pos += len;
            else if (op == 5 || op == 6) {}
            else return -2;
        }
        // judgeSomaticReadHap without PQ (the extraction pass does not use it)
        double tMin, tMax, nMin, nMax; int maxT, maxN;
        const int h4 = 0;
        if (h3 > h4) { tMin = h4; tMax = h3; maxT = 3; } else { tMin = h3; tMax = h4; maxT = 4; }
        if (h1 > h2) { nMin = h2; nMax = h1; maxN = 1; } else { nMin = h1; nMax = h2; maxN = 2; }
        const double tumSim = (tMax == 0) ? 0.0 : tMax / (tMax + tMin), norSim = (nMax == 0) ? 0.0 : nMax / (nMax + nMin);
        int hp = 0; const double thr = P.percentage_threshold;
        if (tMax != 0) { if (tumSim >= thr) { if (norSim >= thr) hp = (maxT == 3) ? (maxN == 1 ? 5 : 7) : (maxN == 1 ? 6 : 8); else hp = (maxT == 3) ? 3 : 4; } }
        else if (nMax != 0) { if (norSim >= thr) hp = maxN; }
        if (norPS.size() > 1) hp = 0;
        out->hp1[r] = h1; out->hp2[r] = h2; out->hp3[r] = h3; out->hp[r] = (uint8_t)hp;
        out->n_ps[r] = (uint8_t)std::min<size_t>(norPS.size(), 255); out->ps_min[r] = norPS.empty() ? 0 : norPS.begin()->first;
        out->end_pos[r] = walked ? ref_pos : b.ref_start[r]; out->read_len[r] = walked ? query_pos : 0;
        if (!h3_sites.empty()) {                                                        // classifyReadsByCase (:462-518) + somaticReadHpCount (:386-404)
            const bool record = norPS.size() <= 1;
            const bool clean = (h1 == 0 || h2 == 0) && h3 != 0;
            for (int64_t v : h3_sites) {
                if (!record) C(v, 24)++;
                else if (clean) { C(v, 25)++; if (h1 == 0 && h2 == 0) C(v, 28)++; else if (h1 != 0 && h2 == 0) C(v, 26)++; else if (h1 == 0 && h2 != 0) C(v, 27)++; }
                else C(v, 29)++;
                C(v, 30 + hp)++;
            }
        }
        if (!tum_sites.empty()) {                                                       // :407-458
            out->has_site[r] = 1;
            for (auto &sv : tum_sites) {
                if (n_pairs < out->pair_capacity) { out->pair_site[n_pairs] = (int32_t)sv.first; out->pair_read[n_pairs] = (int32_t)r; out->pair_base_hp[n_pairs] = (uint8_t)sv.second; }
                ++n_pairs;
                C(sv.first, 15 + hp)++;
            }
        }
    }
    out->n_pairs = n_pairs; out->n_windows = n_win;
    return (n_pairs > out->pair_capacity || n_win > out->win_capacity) ? -9 : 0;
}

}  // extern "C"

// test helper: the REAL libstdc++ std::sort on (key, payload) pairs compared by key only - what the reference does to a merged read's variants
// (src/phase/PhasingGraph.cpp:854 with src/shared/Util.cpp:3-5).  Pins the library's restatement (csrc/lps_stdsort.h).
extern "C" void oracle_std_sort(int32_t *keys, uint8_t *payload, int64_t n) {
    struct E { int32_t k; uint8_t p; };
    std::vector<E> v((size_t)n);
    for (int64_t i = 0; i < n; ++i) v[(size_t)i] = E{keys[i], payload[i]};
    std::sort(v.begin(), v.end(), [](const E &a, const E &b) { return a.k < b.k; });
    for (int64_t i = 0; i < n; ++i) { keys[i] = v[(size_t)i].k; payload[i] = v[(size_t)i].p; }
}
