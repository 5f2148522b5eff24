// lps_synth — seeded synthetic ONT-like data generator (test + bench infrastructure, host C++).
//
// Produces, for ONE contig:  a reference sequence, a het-variant table (SNPs, optional indels) with a
// hidden haplotype assignment, and coordinate-sorted alignments as decoded SoA arrays laid out exactly as
// include/lps_abi.h's lps_read_batch expects (BAM-encoded CIGAR words, 4-bit packed SEQ, raw QUAL).
// The same alignments can be written as SAM text (+FASTA, +VCF) so the reference binary built by
// oracle/build_ref.sh can be run on identical inputs (SURVEY.md §8d "Concrete synthetic inputs").
//
// Properties required by the reference's quirks:
//   * every clip_every-th read carries a >=20 bp soft clip (the reference crashes on contigs without clips)
//   * no variant at position 0; homopolymer runs are injected so the homopolymer rules fire
//   * split (primary+supplementary) reads share their query bases in any overlap, so duplicate
//     observations of one read are identical and std::sort instability cannot matter.
//
// C ABI (ctypes): synth_create / synth_get_* / synth_write_* / synth_destroy.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (auto &v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    double normal() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    int geometric(double p) {  // >=1
        int k = 1;
        while (uni() > p && k < 64) ++k;
        return k;
    }
};

}  // namespace

extern "C" {

struct synth_params {
    uint64_t seed;
    int64_t contig_len;
    int32_t n_snp;            // target number of het SNP sites
    double coverage;
    double len_median;        // 15000
    double len_sigma;         // 0.7585 -> mean ~ 20 kb
    int32_t len_min, len_max; // 1000, 200000
    double sub_rate, ins_rate, del_rate;  // 0.01 each
    double indel_var_frac;    // fraction of extra het indel variants relative to n_snp (0 = SNP only)
    double lowq_frac;         // 0.10 of bases with quality in [2,11]
    double mapq0_frac;        // 0.01
    double secondary_frac;    // 0.003 (flag 0x100)
    double dup_frac;          // 0.002 (flag 0x400)
    int32_t clip_every;       // 7
    double supp_frac;         // 0.02 of reads split into primary + supplementary
    double supp_overlap_frac; // of those, fraction whose halves overlap on the reference
    double hpoly_every;       // mean spacing of injected homopolymer runs (2000)
    double snp_in_hpoly_frac; // fraction of SNPs snapped into a homopolymer run (0.05)
    double snp_pair_frac;     // fraction of SNPs followed by another SNP 1-2 bp away (0.01)
    double tandem_frac;       // fraction of indel variants placed before an injected 2-mer tandem repeat
    int32_t n_threads;
    int32_t clip_pileups;     // number of simulated CNV break-point clip pile-ups (0 = none)
    int64_t gap_start;        // no variants inside [gap_start, gap_start+gap_len)  (two-block fixtures)
    int64_t gap_len;
    uint64_t read_seed;       // 0 = derive reads from `seed`; otherwise same genome/variants, different reads (tumor/normal pairs)
    double somatic_every;     // mean spacing of somatic SNVs (0 = none); they exist only in tumor-clone molecules
    double tumor_purity;      // fraction of molecules drawn from the tumor clone (carry the somatic SNVs of their haplotype)
    double sv_every;          // mean spacing of heterozygous structural variants (0 = none): insertions / deletions of 50..800 bp on one haplotype
};

struct Synth {
    synth_params p;
    std::string ref;
    // variants
    std::vector<int32_t> vpos;
    std::vector<std::string> vref, valt;
    std::vector<uint8_t> vhap;   // haplotype (0/1) that carries ALT
    // somatic SNVs (tumor clone only)
    std::vector<int32_t> spos; std::vector<char> sref, salt; std::vector<uint8_t> shap;
    // structural variants: anchor base (0-based; VCF POS = anchor + 1), signed length (+ insertion, - deletion), haplotype that carries it
    std::vector<int32_t> xpos, xlen; std::vector<uint8_t> xhap;
    // reads SoA
    std::vector<int32_t> ref_start, l_qseq;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq;
    std::vector<uint32_t> name_id;   // generation index; name = r%09u
    std::vector<uint8_t> hap;        // truth haplotype of the molecule
    std::vector<uint64_t> cigar_off, seq_off, qual_off;
    std::vector<uint32_t> cigar;
    std::vector<uint8_t> seq, qual;
};

static const char *kBases = "ACGT";
static inline uint8_t nt16(char c) {
    switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; default: return 15; }
}

struct OneAln {
    int32_t pos; uint16_t flag; uint8_t mapq; uint32_t name; uint8_t hap;
    std::vector<uint32_t> cig; std::string q; std::vector<uint8_t> ql;
};

static void push_op(std::vector<uint32_t> &c, uint32_t op, uint32_t len) {
    if (!len) return;
    if (!c.empty() && (c.back() & 15u) == op) c.back() += len << 4; else c.push_back(len << 4 | op);
}

// simulate one molecule from haplotype `hap` over reference [start, start+span)
static void simulate(const Synth &S, Rng &g, int hap, bool clone, int64_t start, int64_t span,
                     std::vector<uint32_t> &cig, std::string &q, std::vector<uint8_t> &ql,
                     std::vector<int64_t> &q2r /* ref coordinate of every M base, -1 for I */) {
    const synth_params &P = S.p;
    size_t vi = std::lower_bound(S.vpos.begin(), S.vpos.end(), (int32_t)start) - S.vpos.begin();
    size_t si = std::lower_bound(S.spos.begin(), S.spos.end(), (int32_t)start) - S.spos.begin();
    size_t xi = std::lower_bound(S.xpos.begin(), S.xpos.end(), (int32_t)start) - S.xpos.begin();
    int64_t end = std::min<int64_t>(start + span, P.contig_len);
    auto put = [&](char b, int64_t r) {
        q.push_back(b);
        uint8_t qq;
        if (g.uni() < P.lowq_frac) qq = 2 + g.below(10);
        else { double v = 25.0 + 5.0 * g.normal(); qq = (uint8_t)std::min(50.0, std::max(2.0, v)); }
        ql.push_back(qq);
        q2r.push_back(r);
    };
    int64_t p = start;
    while (p < end) {
        while (vi < S.vpos.size() && S.vpos[vi] < p) ++vi;
        bool edge = (p - start < 6) || (end - p < 8);
        char b = S.ref[p];
        int skip = 0, insn = 0; std::string insb;
        if (vi < S.vpos.size() && S.vpos[vi] == p) {
            const std::string &r = S.vref[vi], &a = S.valt[vi];
            if (S.vhap[vi] == hap) {
                if (r.size() == 1 && a.size() == 1) b = a[0];
                else if (r.size() == 1) { insb = a.substr(1); insn = (int)insb.size(); }
                else skip = (int)r.size() - 1;
            }
        }
        while (si < S.spos.size() && S.spos[si] < p) ++si;
        if (clone && si < S.spos.size() && S.spos[si] == p && S.shap[si] == hap) b = S.salt[si];
        if (!edge && g.uni() < P.sub_rate) { char nb; do nb = kBases[g.below(4)]; while (nb == b); b = nb; }
        push_op(cig, 0, 1); put(b, p); ++p;
        if (insn) { push_op(cig, 1, insn); for (char c : insb) put(c, -1); }
        if (skip) { int k = (int)std::min<int64_t>(skip, end - p); push_op(cig, 2, k); p += k; }
        while (xi < S.xpos.size() && S.xpos[xi] < p - 1) ++xi;
        if (xi < S.xpos.size() && S.xpos[xi] == p - 1 && S.xhap[xi] == hap && !edge) {
            // the molecule carries the SV: usually at its length (a caller's SVLEN is a consensus, reads scatter by a few percent), sometimes far
            // enough off that get_snp's |region - oplen| / region < threshold test fails, sometimes not at all
            const double u = g.uni(); const int L0 = std::abs(S.xlen[xi]);
            int k = u < 0.8 ? (int)(L0 * (0.98 + 0.04 * g.uni())) : (u < 0.9 ? (int)(L0 * (g.uni() < 0.5 ? 0.8 : 1.25)) : 0);
            if (k > 0 && S.xlen[xi] > 0) { push_op(cig, 1, k); for (int i = 0; i < k; ++i) put(kBases[g.below(4)], -1); }
            else if (k > 0) { k = (int)std::min<int64_t>(k, end - 8 - p); if (k > 0) { push_op(cig, 2, k); p += k; } }
        }
        if (!edge && p < end - 8) {
            double u = g.uni();
            if (u < P.ins_rate) { int k = g.geometric(0.6); push_op(cig, 1, k); for (int i = 0; i < k; ++i) put(kBases[g.below(4)], -1); }
            else if (u < P.ins_rate + P.del_rate) { int k = (int)std::min<int64_t>(g.geometric(0.6), end - 8 - p); if (k > 0) { push_op(cig, 2, k); p += k; } }
        }
    }
    // must not end on D/I
    while (!cig.empty() && (cig.back() & 15u) != 0) {
        uint32_t op = cig.back() & 15u, len = cig.back() >> 4; cig.pop_back();
        if (op == 1) { q.resize(q.size() - len); ql.resize(ql.size() - len); q2r.resize(q2r.size() - len); }
    }
}

// cut [qa,qb) x ref-consuming ops out of a full alignment: returns cigar for the kept part with clips
static void slice_alignment(const std::vector<uint32_t> &cig, int64_t start, int64_t rfrom, int64_t rto,
                            bool hard, std::vector<uint32_t> &out, int64_t &newpos, int64_t &q_from, int64_t &q_to) {
    // keep M bases whose ref coordinate lies in [rfrom, rto); everything before/after becomes a clip
    int64_t r = start, q = 0; out.clear(); newpos = -1; q_from = -1; q_to = -1;
    std::vector<uint32_t> body;
    for (uint32_t w : cig) {
        uint32_t op = w & 15u; int64_t len = w >> 4;
        if (op == 0) {
            int64_t a = std::max(r, rfrom), b = std::min(r + len, rto);
            if (a < b) {
                if (newpos < 0) { newpos = a; q_from = q + (a - r); }
                push_op(body, 0, (uint32_t)(b - a)); q_to = q + (b - r);
            }
            r += len; q += len;
        } else if (op == 1) {
            if (newpos >= 0 && r > rfrom && r < rto) { push_op(body, 1, (uint32_t)len); q_to = q + len; }
            q += len;
        } else if (op == 2) {
            if (newpos >= 0 && r >= rfrom && r + len < rto) push_op(body, 2, (uint32_t)len);
            r += len;
        }
    }
    while (!body.empty() && (body.back() & 15u) != 0) {
        if ((body.back() & 15u) == 1) q_to -= body.back() >> 4;
        body.pop_back();
    }
    int64_t qtot = q;
    uint32_t cop = hard ? 5u : 4u;
    push_op(out, cop, (uint32_t)q_from);
    for (uint32_t w : body) out.push_back(w);
    push_op(out, cop, (uint32_t)(qtot - q_to));
}

Synth *synth_create(const synth_params *pp) {
    Synth *S = new Synth(); S->p = *pp; const synth_params &P = S->p;
    Rng g(P.seed * 0x9E3779B97F4A7C15ull + 12345);
    const int64_t L = P.contig_len;
    // ---- reference
    S->ref.resize(L);
    {
        int nt = std::max(1, P.n_threads);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] {
            int64_t a = L * t / nt, b = L * (t + 1) / nt; Rng r(P.seed ^ (0xABCDEF12345ull + t * 7919ull));
            for (int64_t i = a; i < b; ) { uint64_t x = r.next(); for (int k = 0; k < 32 && i < b; ++k, ++i) { S->ref[i] = kBases[x & 3]; x >>= 2; } }
        });
        for (auto &x : th) x.join();
    }
    std::vector<int64_t> hp_pos; std::vector<int> hp_len;
    if (P.hpoly_every > 0) {
        for (double x = 50 + g.uni() * P.hpoly_every; x < L - 50; x += 20 + (-std::log(1 - g.uni())) * P.hpoly_every) {
            int64_t s = (int64_t)x; int len = 3 + g.below(6); char b = kBases[g.below(4)];
            for (int i = 0; i < len; ++i) S->ref[s + i] = b;
            // make the flanks differ so the run length is what we injected
            if (S->ref[s - 1] == b) S->ref[s - 1] = kBases[(strchr(kBases, b) - kBases + 1) & 3];
            if (S->ref[s + len] == b) S->ref[s + len] = kBases[(strchr(kBases, b) - kBases + 2) & 3];
            hp_pos.push_back(s); hp_len.push_back(len);
        }
    }
    // ---- variants
    {
        std::vector<int64_t> pos;
        double mean = (double)L / std::max(1, P.n_snp);
        for (double x = 100 + (-std::log(1 - g.uni())) * mean; x < L - 100; x += 1 + (-std::log(1 - g.uni())) * mean) {
            int64_t p = (int64_t)x;
            if (!hp_pos.empty() && g.uni() < P.snp_in_hpoly_frac) {
                size_t k = std::lower_bound(hp_pos.begin(), hp_pos.end(), p) - hp_pos.begin();
                if (k >= hp_pos.size()) k = hp_pos.size() - 1;
                p = hp_pos[k] + g.below(hp_len[k]);
            }
            pos.push_back(p);
            if (g.uni() < P.snp_pair_frac) pos.push_back(p + 1 + g.below(2));
        }
        std::sort(pos.begin(), pos.end()); pos.erase(std::unique(pos.begin(), pos.end()), pos.end());
        int64_t last_end = 0;
        for (int64_t p : pos) {
            if (p < 1 || p >= L - 60 || p < last_end) continue;
            if (P.gap_len > 0 && p >= P.gap_start && p < P.gap_start + P.gap_len) continue;
            bool indel = g.uni() < P.indel_var_frac / (1.0 + P.indel_var_frac);
            std::string r(1, S->ref[p]), a;
            if (!indel) { char nb; do nb = kBases[g.below(4)]; while (nb == r[0]); a = std::string(1, nb); last_end = p + 1; }
            else {
                int k = 1 + g.below(5);
                if (g.uni() < P.tandem_frac) {   // plant a 2-mer x6 tandem repeat right after the site
                    char u = kBases[g.below(4)], v = kBases[(strchr(kBases, u) - kBases + 1 + g.below(3)) & 3];
                    for (int i = 0; i < 12; ++i) S->ref[p + 1 + i] = (i & 1) ? v : u;
                    k = 2;
                }
                if (g.uni() < 0.5) { a = r; for (int i = 0; i < k; ++i) a.push_back(kBases[g.below(4)]); last_end = p + 14; }
                else { r = S->ref.substr(p, 1 + k); a = std::string(1, S->ref[p]); last_end = p + 14 + k; }
            }
            S->vpos.push_back((int32_t)p); S->vref.push_back(r); S->valt.push_back(a); S->vhap.push_back((uint8_t)g.below(2));
        }
    }
    // ---- somatic SNVs: away from germline variants, one haplotype each
    if (P.somatic_every > 0) {
        Rng gs(P.seed * 77 + 5);
        for (double x = 500 + (-std::log(1 - gs.uni())) * P.somatic_every; x < L - 500; x += 50 + (-std::log(1 - gs.uni())) * P.somatic_every) {
            int64_t p = (int64_t)x;
            auto it = std::lower_bound(S->vpos.begin(), S->vpos.end(), (int32_t)p - 20);
            if (it != S->vpos.end() && *it <= p + 20) continue;
            char r = S->ref[p], a; do a = kBases[gs.below(4)]; while (a == r);
            S->spos.push_back((int32_t)p); S->sref.push_back(r); S->salt.push_back(a); S->shap.push_back((uint8_t)gs.below(2));
        }
    }
    // ---- structural variants: well away from the small variants and from each other
    if (P.sv_every > 0) {
        Rng gx(P.seed * 131 + 17);
        int64_t last = 0;
        for (double x = 2000 + (-std::log(1 - gx.uni())) * P.sv_every; x < L - 3000; x += 1500 + (-std::log(1 - gx.uni())) * P.sv_every) {
            int64_t p = (int64_t)x; if (p < last + 1500) continue;
            auto it = std::lower_bound(S->vpos.begin(), S->vpos.end(), (int32_t)p - 30);
            if (it != S->vpos.end() && *it <= p + 30) continue;
            const int len = 50 + (int)gx.below(750);
            S->xpos.push_back((int32_t)p); S->xlen.push_back(gx.uni() < 0.5 ? len : -len); S->xhap.push_back((uint8_t)gx.below(2));
            last = p + len;
        }
    }
    // ---- reads (own seed when read_seed != 0: tumor / normal samples of one genome)
    if (P.read_seed) g = Rng(P.read_seed * 0x9E3779B97F4A7C15ull + 999);
    const uint64_t rseed = P.read_seed ? P.read_seed : P.seed;
    const double mean_len = P.len_median * std::exp(P.len_sigma * P.len_sigma / 2);
    int64_t n_mol = (int64_t)(P.coverage * (double)L / mean_len);
    struct Mol { int64_t start, span; uint32_t id; };
    std::vector<Mol> mols(n_mol);
    for (int64_t i = 0; i < n_mol; ++i) {
        double len = P.len_median * std::exp(P.len_sigma * g.normal());
        len = std::min<double>(P.len_max, std::max<double>(P.len_min, len));
        int64_t st = (int64_t)(g.uni() * (double)(L - P.len_min));
        mols[i] = {st, std::min<int64_t>((int64_t)len, L - st), (uint32_t)i};
    }
    // simulated CNV break points: pile-ups of front clips at one coordinate and back clips further on
    std::vector<std::pair<int64_t, int64_t>> pile;
    for (int k = 0; k < P.clip_pileups; ++k) {
        int64_t a = (int64_t)((k + 0.3) * (double)L / P.clip_pileups), b = a + 60000 + g.below(40000);
        if (b < L - 1000) pile.push_back({a, b});
    }
    std::vector<std::vector<OneAln>> out(n_mol);
    {
        int nt = std::max(1, P.n_threads);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] {
            std::vector<int64_t> q2r;
            for (int64_t i = t; i < n_mol; i += nt) {
                const Mol &m = mols[i]; Rng r(rseed * 1000003ull + 0x51ED270B1ull * (m.id + 1));
                int hap = (int)r.below(2);
                const bool clone = P.somatic_every > 0 && r.uni() < P.tumor_purity;
                int64_t mstart = m.start, mspan = m.span;
                int force_front = -1, force_back = -1;
                for (size_t k = 0; k < pile.size(); ++k) {   // snap molecules crossing a break point
                    if (mstart < pile[k].first && mstart + mspan > pile[k].first + 2000 && r.uni() < 0.5) { mspan -= pile[k].first - mstart; mstart = pile[k].first; force_front = 1; }
                    else if (mstart < pile[k].second - 2000 && mstart + mspan > pile[k].second && r.uni() < 0.5) { mspan = pile[k].second - mstart; force_back = 1; }
                }
                std::vector<uint32_t> cig; std::string q; std::vector<uint8_t> ql; q2r.clear();
                simulate(*S, r, hap, clone, mstart, mspan, cig, q, ql, q2r);
                if (cig.empty()) continue;
                uint16_t fl = r.below(2) ? 16 : 0;
                double u = r.uni();
                if (u < P.secondary_frac) fl |= 0x100; else if (u < P.secondary_frac + P.dup_frac) fl |= 0x400;
                uint8_t mq = r.uni() < P.mapq0_frac ? 0 : 60;
                bool clip = force_front > 0 || force_back > 0 || (P.clip_every > 0 && (m.id % P.clip_every) == 0);
                bool split = !clip && r.uni() < P.supp_frac && mspan > 6000;
                if (!split) {
                    OneAln a; a.pos = (int32_t)mstart; a.flag = fl; a.mapq = mq; a.name = m.id; a.hap = (uint8_t)hap;
                    if (clip) {
                        int k = 20 + r.below(31); bool front = force_front > 0 ? true : (force_back > 0 ? false : (r.below(2) == 0));
                        std::string cs; std::vector<uint8_t> cq;
                        for (int j = 0; j < k; ++j) { cs.push_back(kBases[r.below(4)]); cq.push_back(2 + r.below(30)); }
                        if (front) { a.cig.push_back((uint32_t)k << 4 | 4u); for (auto w : cig) a.cig.push_back(w); a.q = cs + q; a.ql = cq; a.ql.insert(a.ql.end(), ql.begin(), ql.end()); }
                        else { a.cig = cig; a.cig.push_back((uint32_t)k << 4 | 4u); a.q = q + cs; a.ql = ql; a.ql.insert(a.ql.end(), cq.begin(), cq.end()); }
                    } else { a.cig = std::move(cig); a.q = std::move(q); a.ql = std::move(ql); }
                    out[i].push_back(std::move(a));
                } else {
                    int64_t end = mstart; for (uint32_t w : cig) if ((w & 15u) == 0 || (w & 15u) == 2) end += w >> 4;
                    int64_t mid = mstart + (end - mstart) / 2, m1 = mid, m2 = mid;
                    if (r.uni() < P.supp_overlap_frac) { int64_t ov = (int64_t)((end - mstart) * (0.05 + 0.4 * r.uni())); m1 = mid - ov / 2; m2 = mid + ov / 2; }
                    for (int h = 0; h < 2; ++h) {
                        OneAln a; int64_t np, qf, qt;
                        slice_alignment(cig, mstart, h == 0 ? mstart : m1, h == 0 ? m2 : end, h == 1, a.cig, np, qf, qt);
                        if (np < 0) continue;
                        a.pos = (int32_t)np; a.flag = fl | (h ? 0x800 : 0); a.mapq = mq; a.name = m.id; a.hap = (uint8_t)hap;
                        if (h == 0) { a.q = q; a.ql = ql; } else { a.q = q.substr(qf, qt - qf); a.ql.assign(ql.begin() + qf, ql.begin() + qt); }
                        out[i].push_back(std::move(a));
                    }
                }
            }
        });
        for (auto &x : th) x.join();
    }
    std::vector<OneAln *> all;
    for (auto &v : out) for (auto &a : v) all.push_back(&a);
    std::stable_sort(all.begin(), all.end(), [](const OneAln *a, const OneAln *b) { return a->pos < b->pos; });
    size_t n = all.size();
    S->ref_start.resize(n); S->l_qseq.resize(n); S->flag.resize(n); S->mapq.resize(n); S->name_id.resize(n); S->hap.resize(n);
    S->cigar_off.resize(n + 1); S->seq_off.resize(n + 1); S->qual_off.resize(n + 1);
    uint64_t co = 0, so = 0, qo = 0;
    for (size_t i = 0; i < n; ++i) {
        const OneAln &a = *all[i];
        S->ref_start[i] = a.pos; S->l_qseq[i] = (int32_t)a.q.size(); S->flag[i] = a.flag; S->mapq[i] = a.mapq; S->name_id[i] = a.name; S->hap[i] = a.hap;
        S->cigar_off[i] = co; S->seq_off[i] = so; S->qual_off[i] = qo;
        co += a.cig.size(); so += (a.q.size() + 1) / 2; qo += a.q.size();
    }
    S->cigar_off[n] = co; S->seq_off[n] = so; S->qual_off[n] = qo;
    S->cigar.resize(co); S->seq.assign(so, 0); S->qual.resize(qo);
    {
        int nt = std::max(1, P.n_threads); std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] {
            for (size_t i = t; i < n; i += nt) {
                const OneAln &a = *all[i];
                std::copy(a.cig.begin(), a.cig.end(), S->cigar.begin() + S->cigar_off[i]);
                std::copy(a.ql.begin(), a.ql.end(), S->qual.begin() + S->qual_off[i]);
                uint8_t *sp = S->seq.data() + S->seq_off[i];
                for (size_t j = 0; j < a.q.size(); ++j) sp[j >> 1] |= nt16(a.q[j]) << ((~j & 1) << 2);
            }
        });
        for (auto &x : th) x.join();
    }
    return S;
}

void synth_destroy(Synth *S) { delete S; }
int64_t synth_n_reads(Synth *S) { return (int64_t)S->ref_start.size(); }
int64_t synth_n_variants(Synth *S) { return (int64_t)S->vpos.size(); }
int64_t synth_n_somatic(Synth *S) { return (int64_t)S->spos.size(); }
int64_t synth_n_sv(Synth *S) { return (int64_t)S->xpos.size(); }
const int32_t *synth_sv_pos(Synth *S) { return S->xpos.data(); }
const int32_t *synth_sv_len(Synth *S) { return S->xlen.data(); }
const uint8_t *synth_sv_hap(Synth *S) { return S->xhap.data(); }
const int32_t *synth_som_pos(Synth *S) { return S->spos.data(); }
const char *synth_som_ref(Synth *S) { return S->sref.data(); }
const char *synth_som_alt(Synth *S) { return S->salt.data(); }
const uint8_t *synth_som_hap(Synth *S) { return S->shap.data(); }
const char *synth_ref(Synth *S) { return S->ref.data(); }
const int32_t *synth_var_pos(Synth *S) { return S->vpos.data(); }
const uint8_t *synth_var_hap(Synth *S) { return S->vhap.data(); }
const char *synth_var_ref(Synth *S, int64_t i) { return S->vref[i].c_str(); }
const char *synth_var_alt(Synth *S, int64_t i) { return S->valt[i].c_str(); }
const int32_t *synth_ref_start(Synth *S) { return S->ref_start.data(); }
const int32_t *synth_l_qseq(Synth *S) { return S->l_qseq.data(); }
const uint16_t *synth_flag(Synth *S) { return S->flag.data(); }
const uint8_t *synth_mapq(Synth *S) { return S->mapq.data(); }
const uint32_t *synth_name_id(Synth *S) { return S->name_id.data(); }
const uint8_t *synth_read_hap(Synth *S) { return S->hap.data(); }
const uint64_t *synth_cigar_off(Synth *S) { return S->cigar_off.data(); }
const uint64_t *synth_seq_off(Synth *S) { return S->seq_off.data(); }
const uint64_t *synth_qual_off(Synth *S) { return S->qual_off.data(); }
const uint32_t *synth_cigar(Synth *S) { return S->cigar.data(); }
const uint8_t *synth_seq(Synth *S) { return S->seq.data(); }
const uint8_t *synth_qual(Synth *S) { return S->qual.data(); }

int synth_write_fasta(Synth *S, const char *path, const char *chr) {
    FILE *f = fopen(path, "w"); if (!f) return -1;
    fprintf(f, ">%s\n", chr);
    const int64_t L = S->p.contig_len;
    for (int64_t i = 0; i < L; i += 60) { fwrite(S->ref.data() + i, 1, (size_t)std::min<int64_t>(60, L - i), f); fputc('\n', f); }
    fclose(f);
    std::string fai = std::string(path) + ".fai"; f = fopen(fai.c_str(), "w"); if (!f) return -1;
    fprintf(f, "%s\t%lld\t%zu\t60\t61\n", chr, (long long)L, strlen(chr) + 2); fclose(f);
    return 0;
}

// phased==0: GT 0/1 (input to `phase`); phased==1: truth-phased a|b with one PS per contig (input to `haplotag`)
int synth_write_vcf(Synth *S, const char *path, const char *chr, int phased) {
    FILE *f = fopen(path, "w"); if (!f) return -1;
    fprintf(f, "##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n##contig=<ID=%s,length=%lld>\n", chr, (long long)S->p.contig_len);
    fprintf(f, "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">\n");
    if (phased) fprintf(f, "##FORMAT=<ID=PS,Number=1,Type=Integer,Description=\"Phase set identifier\">\n");
    fprintf(f, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n");
    for (size_t i = 0; i < S->vpos.size(); ++i) {
        if (!phased) fprintf(f, "%s\t%d\t.\t%s\t%s\t30\tPASS\t.\tGT:GQ\t0/1:30\n", chr, S->vpos[i] + 1, S->vref[i].c_str(), S->valt[i].c_str());
        else fprintf(f, "%s\t%d\t.\t%s\t%s\t30\tPASS\t.\tGT:GQ:PS\t%s:30:%d\n", chr, S->vpos[i] + 1, S->vref[i].c_str(), S->valt[i].c_str(), S->vhap[i] ? "0|1" : "1|0", S->vpos[0] + 1);
    }
    fclose(f); return 0;
}

// tumor VCF for somatic_haplotag: somatic SNVs as 0/1 (plus, optionally, the germline hets as a caller would report them)
int synth_write_vcf_tumor(Synth *S, const char *path, const char *chr, int with_germline) {
    FILE *f = fopen(path, "w"); if (!f) return -1;
    fprintf(f, "##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n##contig=<ID=%s,length=%lld>\n", chr, (long long)S->p.contig_len);
    fprintf(f, "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">\n");
    fprintf(f, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n");
    size_t i = 0, j = 0;
    while (i < S->spos.size() || (with_germline && j < S->vpos.size())) {
        bool som = i < S->spos.size() && (!with_germline || j >= S->vpos.size() || S->spos[i] < S->vpos[j]);
        if (som) { fprintf(f, "%s\t%d\t.\t%c\t%c\t30\tPASS\t.\tGT:GQ\t0/1:30\n", chr, S->spos[i] + 1, S->sref[i], S->salt[i]); ++i; }
        else { fprintf(f, "%s\t%d\t.\t%s\t%s\t30\tPASS\t.\tGT:GQ\t0/1:30\n", chr, S->vpos[j] + 1, S->vref[j].c_str(), S->valt[j].c_str()); ++j; }
    }
    fclose(f); return 0;
}

int synth_write_sam(Synth *S, const char *path, const char *chr) {
    FILE *f = fopen(path, "w"); if (!f) return -1;
    fprintf(f, "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:%s\tLN:%lld\n", chr, (long long)S->p.contig_len);
    static const char *nt = "=ACMGRSVTWYHKDBN"; static const char *ops = "MIDNSHP=XB";
    std::string line;
    for (size_t i = 0; i < S->ref_start.size(); ++i) {
        char buf[128];
        snprintf(buf, sizeof buf, "r%09u\t%u\t%s\t%d\t%u\t", S->name_id[i], S->flag[i], chr, S->ref_start[i] + 1, S->mapq[i]);
        line = buf;
        for (uint64_t c = S->cigar_off[i]; c < S->cigar_off[i + 1]; ++c) { snprintf(buf, sizeof buf, "%u%c", S->cigar[c] >> 4, ops[S->cigar[c] & 15u]); line += buf; }
        line += "\t*\t0\t0\t";
        int l = S->l_qseq[i]; const uint8_t *sp = S->seq.data() + S->seq_off[i], *qp = S->qual.data() + S->qual_off[i];
        for (int j = 0; j < l; ++j) line.push_back(nt[(sp[j >> 1] >> ((~j & 1) << 2)) & 15]);
        line.push_back('\t');
        for (int j = 0; j < l; ++j) line.push_back((char)(33 + qp[j]));
        line.push_back('\n');
        fwrite(line.data(), 1, line.size(), f);
    }
    fclose(f); return 0;
}

}  // extern "C"
