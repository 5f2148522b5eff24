// lps_synth_gpu — seeded synthetic ONT-like contig generated ON THE GPU (test + bench infrastructure, never linked by the product).
//
// Why a second generator: tools/lps_synth.cpp builds every read base by base on host threads (~0.1 Gbase/s); the workload the
// metric is quoted on (50x whole genome, 155 Gbases) needs tens of Gbases per second.  Here every quantity is a pure
// function of (seed, index) - a counter-based hash instead of a sequential RNG - so reference, variants, molecules, CIGARs and
// bases are filled by independent threads straight into device memory in the layout of include/lps_abi.h's lps_read_batch
// (BAM-encoded CIGAR words, 4-bit SEQ, raw QUAL), ready for lps_push_reads_device.  The same arrays can be copied to the host for
// the oracle, or written as FASTA / VCF / SAM for the reference binary (oracle/_ref).
//
// Model (same knobs as lps_synth.cpp, SURVEY.md §8d "Concrete synthetic inputs"):
//   reference  i.i.d. ACGT + one homopolymer run (3-8) per hpoly_every window
//   variants   het SNPs, one per stratum of contig_len / n_snp bases (+ snp_pair_frac close pairs, snp_in_hpoly_frac snapped into runs)
//   molecules  log-normal lengths, uniform starts, fair haplotype coin; errors per 32-base BLOCK of the molecule: at most one
//              insertion or deletion (geometric length <= 8) placed inside the block, substitutions per base - blocks are independent,
//              so a base's reference position follows from a per-block prefix of query lengths
//   alignments every clip_every-th molecule soft-clipped (the reference crashes on contigs without clips),
//              supp_frac split into primary + supplementary (0x800) that share their blocks where they overlap, optional clip pile-ups
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define SG_TRY(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess) {                                                                                   \
            char _b[512];                                                                                         \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
            throw std::string(_b);                                                                                \
        }                                                                                                         \
    } while (0)

extern "C" {
typedef struct sg_params {
    uint64_t seed;
    int64_t contig_len;
    int32_t n_snp;
    int32_t clip_every;        // 7
    double coverage;
    double len_median;         // 15000
    double len_sigma;          // 0.7585 -> mean ~ 20 kb
    int32_t len_min, len_max;  // 1000, 200000
    double sub_rate, ins_rate, del_rate;   // 0.01 each
    double lowq_frac;          // 0.10 of bases with quality in [2,11]
    double mapq0_frac;         // 0.01
    double secondary_frac;     // 0.003 (flag 0x100)
    double dup_frac;           // 0.002 (flag 0x400)
    double supp_frac;          // 0.02 of molecules split into primary + supplementary
    double supp_overlap_frac;  // of those, fraction whose halves overlap on the reference
    int32_t hpoly_every;       // window that holds one injected homopolymer run (2000)
    int32_t clip_pileups;      // simulated CNV break points (0 = none)
    double snp_in_hpoly_frac;  // 0.05
    double snp_pair_frac;      // 0.01
    uint64_t read_seed;        // 0 = derive reads from `seed`
} sg_params;
}

namespace {

__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t hsh(uint64_t seed, uint64_t a) { return mix64(seed + 0x9E3779B97F4A7C15ull * (a + 1)); }
__host__ __device__ inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

constexpr int BLK = 32;     // bases of a molecule per error block

struct Aln {                // one alignment (32 bytes)
    int32_t ref_start;      // = mstart + BLK * b0
    uint32_t mol;           // molecule id = name id
    int32_t mstart;         // molecule start on the reference (after pile-up snapping)
    int32_t b0, nb;         // blocks [b0, b0 + nb) of the molecule
    int32_t front_clip, back_clip;
    uint16_t flag;
    uint8_t mapq;
    uint8_t kind;           // bit0 front clip is hard, bit1 back clip is hard, bit2 molecule haplotype
};

struct DevParams {
    uint64_t s_ref, s_hp, s_var, s_mol, s_err, s_sub, s_base, s_qual;
    long long L;
    int n_strata; double stratum;
    int W;                  // homopolymer window
    uint32_t p_ins, p_del;  // per block, scaled to 2^32
    uint32_t p_sub;         // per base, scaled to 2^32
    uint32_t p_lowq;
    double snp_in_hpoly_frac, snp_pair_frac;
    double len_median, len_sigma; int len_min, len_max;
    double mapq0_frac, secondary_frac, dup_frac, supp_frac, supp_overlap_frac;
    int clip_every, n_pile;
    long long n_mol;
};

// ---- reference
__global__ void k_ref(DevParams P, char *ref) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // 32 bases per thread
    const long long p0 = i * 32;
    if (p0 >= P.L) return;
    uint64_t x = hsh(P.s_ref, (uint64_t)i);
    for (int k = 0; k < 32 && p0 + k < P.L; ++k, x >>= 2) ref[p0 + k] = "ACGT"[x & 3];
}
__host__ __device__ inline void hpoly_run(const DevParams &P, long long w, long long &s, int &len, int &base) {
    const uint64_t h = hsh(P.s_hp, (uint64_t)w);
    s = w * P.W + 40 + (long long)(h % (uint64_t)(P.W - 80)); len = 3 + (int)((h >> 32) % 6); base = (int)((h >> 48) & 3);
}
__global__ void k_hpoly(DevParams P, char *ref) {
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if ((w + 1) * P.W > P.L) return;
    long long s; int len, b; hpoly_run(P, w, s, len, b);
    const char c = "ACGT"[b];
    for (int i = 0; i < len; ++i) ref[s + i] = c;
    if (ref[s - 1] == c) ref[s - 1] = "ACGT"[(b + 1) & 3];          // flanks differ: the run has the injected length
    if (ref[s + len] == c) ref[s + len] = "ACGT"[(b + 2) & 3];
}

// ---- variants: stratum i -> 1 or 2 SNPs
__device__ inline int stratum_variants(const DevParams &P, int i, int32_t *pos /*2*/) {
    const long long lo = 100 + (long long)((double)i * P.stratum), hi = 100 + (long long)((double)(i + 1) * P.stratum);
    if (hi <= lo || hi > P.L - 100) return 0;
    const uint64_t h = hsh(P.s_var, (uint64_t)i);
    long long p = lo + (long long)(h % (uint64_t)(hi - lo));
    if (u01(hsh(P.s_var ^ 0x1111, (uint64_t)i)) < P.snp_in_hpoly_frac) {     // snap into the homopolymer run of p's window when it lies in this stratum
        const long long w = p / P.W;
        if ((w + 1) * P.W <= P.L) { long long s; int len, b; hpoly_run(P, w, s, len, b); if (s >= lo && s + len <= hi) p = s + (long long)((h >> 40) % (uint64_t)len); }
    }
    pos[0] = (int32_t)p;
    int n = 1;
    if (u01(hsh(P.s_var ^ 0x2222, (uint64_t)i)) < P.snp_pair_frac) { const long long q = p + 1 + (long long)((h >> 50) & 1); if (q < hi) { pos[1] = (int32_t)q; n = 2; } }
    return n;
}
__global__ void k_var_count(DevParams P, uint32_t *cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_strata) return;
    int32_t pos[2]; cnt[i] = (uint32_t)stratum_variants(P, i, pos);
}
// hapcode[p]: bits 0-1 reference base, 2-3 haplotype-0 base, 4-5 haplotype-1 base
__global__ void k_hapcode(DevParams P, const char *ref, uint8_t *code) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.L) return;
    const char c = ref[p]; const int b = c == 'A' ? 0 : (c == 'C' ? 1 : (c == 'G' ? 2 : 3));
    code[p] = (uint8_t)(b | (b << 2) | (b << 4));
}
__global__ void k_var_fill(DevParams P, const uint32_t *off, const char *ref, uint8_t *code, int32_t *vpos, uint8_t *vref, uint8_t *valt, uint8_t *vhap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_strata) return;
    int32_t pos[2]; const int n = stratum_variants(P, i, pos);
    for (int k = 0; k < n; ++k) {
        const uint32_t v = off[i] + k; const int32_t p = pos[k];
        const uint64_t h = hsh(P.s_var ^ 0x3333, (uint64_t)p);
        const int rb = code[p] & 3, ab = (rb + 1 + (int)(h % 3)) & 3, hap = (int)((h >> 20) & 1);
        vpos[v] = p; vref[v] = (uint8_t)"ACGT"[rb]; valt[v] = (uint8_t)"ACGT"[ab]; vhap[v] = (uint8_t)hap;
        code[p] = (uint8_t)(rb | ((hap == 0 ? ab : rb) << 2) | ((hap == 1 ? ab : rb) << 4));
    }
}

// ---- molecules -> alignments
__device__ inline int molecule_alignments(const DevParams &P, long long i, Aln *out /*2*/) {
    const uint64_t h0 = hsh(P.s_mol, (uint64_t)i * 8 + 0), h1 = hsh(P.s_mol, (uint64_t)i * 8 + 1), h2 = hsh(P.s_mol, (uint64_t)i * 8 + 2),
                   h3 = hsh(P.s_mol, (uint64_t)i * 8 + 3), h4 = hsh(P.s_mol, (uint64_t)i * 8 + 4);
    double ua = u01(h0); if (ua < 1e-300) ua = 1e-300;
    const double z = sqrt(-2.0 * log(ua)) * cos(6.283185307179586 * u01(h1));
    double len = P.len_median * exp(P.len_sigma * z);
    len = fmin((double)P.len_max, fmax((double)P.len_min, len));
    long long start = (long long)(u01(h2) * (double)(P.L - P.len_min));
    long long span = (long long)len; if (span > P.L - start) span = P.L - start;
    const int hap = (int)(h3 & 1);
    int force_front = 0, force_back = 0;
    for (int k = 0; k < P.n_pile; ++k) {                                     // snap molecules that cross a simulated break point: front clips pile up at a, back clips at b
        const long long a = (long long)(((double)k + 0.3) * (double)P.L / (double)P.n_pile), b = a + 8000 + (long long)(hsh(P.s_mol ^ 0x77, (uint64_t)k) % 12000);
        if (b >= P.L - 1000) continue;
        const double u = u01(hsh(P.s_mol ^ 0x99, (uint64_t)i * 64 + k));
        if (start < a && start + span > a + 2000 && u < 0.5) { span -= a - start; start = a; force_front = 1; }
        else if (start < b - 2000 && start + span > b && u < 0.5) { start += (b - start) % BLK; span = b - start; force_back = 1; }   // whole blocks up to b: the clip sits exactly there
    }
    const int nb = (int)(span / BLK);
    if (nb < 4) return 0;
    uint16_t fl = (h3 & 2) ? 16 : 0;
    const double uf = u01(h4);
    if (uf < P.secondary_frac) fl |= 0x100; else if (uf < P.secondary_frac + P.dup_frac) fl |= 0x400;
    const uint8_t mq = u01(hsh(P.s_mol, (uint64_t)i * 8 + 5)) < P.mapq0_frac ? 0 : 60;
    const bool clip = force_front || force_back || (P.clip_every > 0 && (i % P.clip_every) == 0);
    const uint64_t h6 = hsh(P.s_mol, (uint64_t)i * 8 + 6);
    const bool split = !clip && u01(h6) < P.supp_frac && (long long)nb * BLK > 6000;
    Aln a{}; a.mol = (uint32_t)i; a.mstart = (int32_t)start; a.flag = fl; a.mapq = mq; a.kind = (uint8_t)(hap << 2);
    if (!split) {
        a.ref_start = (int32_t)start; a.b0 = 0; a.nb = nb;
        if (clip) { const int k = 20 + (int)((h6 >> 20) % 31); const bool front = force_front ? true : (force_back ? false : ((h6 >> 40) & 1) == 0); if (front) a.front_clip = k; else a.back_clip = k; }
        out[0] = a; return 1;
    }
    const uint64_t h7 = hsh(P.s_mol, (uint64_t)i * 8 + 7);
    int mid = nb / 2, b1 = mid, b2 = mid;
    if (u01(h7) < P.supp_overlap_frac) { const int ov = (int)((double)nb * (0.05 + 0.4 * u01(h7 * 0x9E3779B97F4A7C15ull))); b1 = mid - ov / 2; b2 = mid + ov / 2; }
    if (b1 < 1) b1 = 1; if (b2 > nb - 1) b2 = nb - 1; if (b2 < 1) b2 = 1;
    Aln p = a; p.ref_start = (int32_t)start; p.b0 = 0; p.nb = b2; p.back_clip = (nb - b2) * BLK;                       // primary: soft clip for the rest of the read
    Aln s = a; s.ref_start = (int32_t)(start + (long long)b1 * BLK); s.b0 = b1; s.nb = nb - b1; s.front_clip = b1 * BLK; s.kind |= 1; s.flag |= 0x800;   // supplementary: hard clip
    out[0] = p; out[1] = s; return 2;
}
__global__ void k_mol_count(DevParams P, uint32_t *cnt) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_mol) return;
    Aln a[2]; cnt[i] = (uint32_t)molecule_alignments(P, i, a);
}
__global__ void k_mol_fill(DevParams P, const uint32_t *off, Aln *aln, uint32_t *key, uint32_t *idx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_mol) return;
    Aln a[2]; const int n = molecule_alignments(P, i, a);
    for (int k = 0; k < n; ++k) { const uint32_t j = off[i] + k; aln[j] = a[k]; key[j] = (uint32_t)a[k].ref_start; idx[j] = j; }
}
__global__ void k_gather(const Aln *in, const uint32_t *idx, long long n, Aln *out, unsigned long long *n_blk) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = in[idx[i]]; n_blk[i] = (unsigned long long)out[i].nb;
}

// error event of block `blk` of molecule `mol`: type 0 none, 1 insertion of k bases after block offset o, 2 deletion of the k bases after offset o
__device__ inline void block_event(const DevParams &P, uint32_t mol, uint32_t blk, int &type, int &o, int &k) {
    const uint64_t h = hsh(P.s_err, ((uint64_t)mol << 24) ^ (uint64_t)blk);
    const uint32_t u = (uint32_t)h;
    type = u < P.p_ins ? 1 : (u - P.p_ins < P.p_del ? 2 : 0);
    o = 1 + (int)((h >> 32) % 20);
    const uint32_t g = (uint32_t)(h >> 44) & 0xffffu;                         // geometric(0.6), at most 8
    k = 1 + (g < 26214u) + (g < 10486u) + (g < 4194u) + (g < 1678u) + (g < 671u) + (g < 268u) + (g < 107u);
}
__device__ inline int block_qlen(int type, int k) { return BLK + (type == 1 ? k : (type == 2 ? -k : 0)); }

// thread per alignment: query offset of every block (relative to the first aligned base), CIGAR op count, l_qseq
__global__ void k_aln_sizes(DevParams P, const Aln *aln, long long n, const unsigned long long *blk_off, uint32_t *qoff,
                            unsigned long long *n_cig, unsigned long long *n_seq, unsigned long long *n_qual, int32_t *l_qseq) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Aln a = aln[i];
    uint32_t q = 0; int ev = 0;
    uint32_t *qo = qoff + blk_off[i];
    for (int b = 0; b < a.nb; ++b) {
        int t, o, k; block_event(P, a.mol, (uint32_t)(a.b0 + b), t, o, k);
        qo[b] = q; q += (uint32_t)block_qlen(t, k); ev += t != 0;
    }
    const int fs = (a.kind & 1) ? 0 : a.front_clip, bs = (a.kind & 2) ? 0 : a.back_clip;
    const int lq = fs + (int)q + bs;
    l_qseq[i] = lq;
    n_cig[i] = (unsigned long long)(2 * ev + 1 + (a.front_clip > 0) + (a.back_clip > 0));
    n_seq[i] = (unsigned long long)((lq + 7) / 8) * 4;                         // rows padded to whole 8-base groups: aligned stores in k_fill
    n_qual[i] = (unsigned long long)((lq + 7) / 8) * 8;
}
__global__ void k_headers(const Aln *aln, long long n, int32_t *ref_start, uint16_t *flag, uint8_t *mapq, uint32_t *name_id, uint8_t *read_hap) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Aln a = aln[i];
    ref_start[i] = a.ref_start; flag[i] = a.flag; mapq[i] = a.mapq; name_id[i] = a.mol; read_hap[i] = (uint8_t)((a.kind >> 2) & 1);
}
// thread per alignment: CIGAR words (M runs merged across blocks)
__global__ void k_cigar(DevParams P, const Aln *aln, long long n, const unsigned long long *cig_off, uint32_t *cigar) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Aln a = aln[i];
    uint32_t *c = cigar + cig_off[i];
    if (a.front_clip > 0) *c++ = ((uint32_t)a.front_clip << 4) | ((a.kind & 1) ? 5u : 4u);
    uint32_t run = 0;
    for (int b = 0; b < a.nb; ++b) {
        int t, o, k; block_event(P, a.mol, (uint32_t)(a.b0 + b), t, o, k);
        if (t == 0) { run += BLK; continue; }
        run += (uint32_t)(o + 1);
        *c++ = (run << 4) | 0u;
        *c++ = ((uint32_t)k << 4) | (t == 1 ? 1u : 2u);
        run = (uint32_t)(BLK - 1 - o - (t == 2 ? k : 0));
    }
    *c++ = (run << 4) | 0u;
    if (a.back_clip > 0) *c++ = ((uint32_t)a.back_clip << 4) | ((a.kind & 2) ? 5u : 4u);
}
// workgroup per alignment, thread per group of 8 query bases: 4 bytes of SEQ + 8 bytes of QUAL
__global__ __launch_bounds__(256) void k_fill(DevParams P, const Aln *aln, const unsigned long long *blk_off, const uint32_t *qoff, const int32_t *l_qseq,
                                              const unsigned long long *seq_off, const unsigned long long *qual_off, const uint8_t *code,
                                              uint8_t *seq, uint8_t *qual) {
    const long long i = blockIdx.x;
    const Aln a = aln[i];
    const int lq = l_qseq[i];
    const int fs = (a.kind & 1) ? 0 : a.front_clip;
    const uint32_t *qo = qoff + blk_off[i];
    const int hapshift = 2 + 2 * ((a.kind >> 2) & 1);
    int tl, ol, kl; block_event(P, a.mol, (uint32_t)(a.b0 + a.nb - 1), tl, ol, kl);
    const int body_len = (int)qo[a.nb - 1] + block_qlen(tl, kl);
    uint32_t *seq4 = reinterpret_cast<uint32_t *>(seq + seq_off[i]); uint2 *qual8 = reinterpret_cast<uint2 *>(qual + qual_off[i]);
    const int n_groups = (lq + 7) / 8;
    for (int g = threadIdx.x; g < n_groups; g += blockDim.x) {
        uint32_t sw = 0; unsigned long long qw = 0;
        int b = -1, bq0 = 0, bql = 0, bt = 0, bo = 0, bk = 0;                  // current block of the body
        for (int j = 0; j < 8; ++j) {
            const int q = g * 8 + j;
            int base = 0, qv = 0;
            if (q < lq) {
                const int qb = q - fs;
                uint64_t hq;
                if (qb < 0 || qb >= body_len) { hq = hsh(P.s_base, ((uint64_t)a.mol << 24) ^ (uint64_t)q); base = (int)(hq >> 60) & 3; }   // soft-clipped bases: random
                else {
                    if (b < 0 || qb >= bq0 + bql) {
                        if (b >= 0 && b + 1 < a.nb && qb < (int)(b + 2 < a.nb ? qo[b + 2] : 0x7fffffff)) ++b;        // the next block (groups walk forwards)
                        else { int lo = 0, hi = a.nb - 1; while (lo < hi) { const int m = (lo + hi + 1) >> 1; if ((int)qo[m] <= qb) lo = m; else hi = m - 1; } b = lo; }
                        block_event(P, a.mol, (uint32_t)(a.b0 + b), bt, bo, bk);
                        bq0 = (int)qo[b]; bql = block_qlen(bt, bk);
                    }
                    const int jb = qb - bq0;
                    // keyed by (molecule, block, offset in the block): the two alignments of a split molecule hold the same bases AND qualities where they overlap
                    hq = hsh(P.s_qual, ((uint64_t)a.mol << 24) ^ (uint64_t)((a.b0 + b) * 64 + jb));
                    int roff = jb; bool inserted = false;
                    if (bt == 1) { if (jb > bo + bk) roff = jb - bk; else if (jb > bo) inserted = true; }
                    else if (bt == 2) { if (jb > bo) roff = jb + bk; }
                    if (inserted) base = (int)(hq >> 60) & 3;
                    else {
                        const long long rp = (long long)a.mstart + (long long)(a.b0 + b) * BLK + roff;
                        base = (code[rp] >> hapshift) & 3;
                        const uint64_t hs = hsh(P.s_sub, ((uint64_t)a.mol << 24) ^ (uint64_t)((a.b0 + b) * BLK + roff));
                        if ((uint32_t)hs < P.p_sub) base = (base + 1 + (int)((hs >> 32) % 3)) & 3;
                    }
                }
                if ((uint32_t)hq < P.p_lowq) qv = 2 + (int)((hq >> 32) % 10);
                else {                                                         // ~N(25,5): Irwin-Hall of four bytes
                    const int s4 = (int)((hq >> 32) & 255) + (int)((hq >> 40) & 255) + (int)((hq >> 48) & 255) + (int)((hq >> 56) & 15) * 17;
                    int v = 25 + (int)lrintf((float)(s4 - 510) * (5.0f / 147.8f));
                    qv = v < 2 ? 2 : (v > 50 ? 50 : v);
                }
                sw |= (uint32_t)(1u << base) << (((j >> 1) << 3) + ((~j & 1) << 2));   // byte j/2, high nibble first
                qw |= (unsigned long long)(unsigned)qv << (8 * j);
            }
        }
        seq4[g] = sw; qual8[g] = make_uint2((uint32_t)qw, (uint32_t)(qw >> 32));
    }
}

template <class T> struct Dev {
    T *p = nullptr; size_t n = 0;
    ~Dev() { if (p) (void)hipFree(p); }
    void alloc(size_t k) { if (p) { (void)hipFree(p); p = nullptr; } n = k; SG_TRY(hipMalloc((void **)&p, (k + 16) * sizeof(T))); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; } n = 0; }
};

void exscan64(unsigned long long *in, unsigned long long *out, size_t n, Dev<char> &tmp) {
    size_t need = 0;
    SG_TRY(rocprim::exclusive_scan(nullptr, need, in, out, 0ull, n, rocprim::plus<unsigned long long>(), nullptr));
    if (need > tmp.n) tmp.alloc(need + 256);
    SG_TRY(rocprim::exclusive_scan(tmp.p, need, in, out, 0ull, n, rocprim::plus<unsigned long long>(), nullptr));
}
void exscan32(uint32_t *in, uint32_t *out, size_t n, Dev<char> &tmp) {
    size_t need = 0;
    SG_TRY(rocprim::exclusive_scan(nullptr, need, in, out, 0u, n, rocprim::plus<uint32_t>(), nullptr));
    if (need > tmp.n) tmp.alloc(need + 256);
    SG_TRY(rocprim::exclusive_scan(tmp.p, need, in, out, 0u, n, rocprim::plus<uint32_t>(), nullptr));
}

}  // namespace

struct sg_handle {
    int device = 0; sg_params p{}; std::string err;
    long long n_var = 0, n_aln = 0; unsigned long long n_cig = 0, n_seq = 0, n_qual = 0;
    Dev<char> ref; Dev<uint8_t> code; Dev<int32_t> vpos; Dev<uint8_t> vref, valt, vhap;
    Dev<int32_t> ref_start, l_qseq; Dev<uint16_t> flag; Dev<uint8_t> mapq, read_hap; Dev<uint32_t> name_id;
    Dev<unsigned long long> cig_off, seq_off, qual_off; Dev<uint32_t> cigar; Dev<uint8_t> seq, qual;
    double gen_ms = 0;
};

enum { SG_REF = 0, SG_VAR_POS, SG_VAR_REF, SG_VAR_ALT, SG_VAR_HAP, SG_REF_START, SG_L_QSEQ, SG_FLAG, SG_MAPQ, SG_NAME_ID, SG_READ_HAP,
       SG_CIGAR_OFF, SG_SEQ_OFF, SG_QUAL_OFF, SG_CIGAR, SG_SEQ, SG_QUAL, SG_N_ARRAYS };

static bool sg_array(sg_handle *h, int which, const void **p, size_t *bytes) {
    const size_t n = (size_t)h->n_aln, v = (size_t)h->n_var;
    switch (which) {
        case SG_REF: *p = h->ref.p; *bytes = (size_t)h->p.contig_len; return true;
        case SG_VAR_POS: *p = h->vpos.p; *bytes = v * 4; return true;
        case SG_VAR_REF: *p = h->vref.p; *bytes = v; return true;
        case SG_VAR_ALT: *p = h->valt.p; *bytes = v; return true;
        case SG_VAR_HAP: *p = h->vhap.p; *bytes = v; return true;
        case SG_REF_START: *p = h->ref_start.p; *bytes = n * 4; return true;
        case SG_L_QSEQ: *p = h->l_qseq.p; *bytes = n * 4; return true;
        case SG_FLAG: *p = h->flag.p; *bytes = n * 2; return true;
        case SG_MAPQ: *p = h->mapq.p; *bytes = n; return true;
        case SG_NAME_ID: *p = h->name_id.p; *bytes = n * 4; return true;
        case SG_READ_HAP: *p = h->read_hap.p; *bytes = n; return true;
        case SG_CIGAR_OFF: *p = h->cig_off.p; *bytes = (n + 1) * 8; return true;
        case SG_SEQ_OFF: *p = h->seq_off.p; *bytes = (n + 1) * 8; return true;
        case SG_QUAL_OFF: *p = h->qual_off.p; *bytes = (n + 1) * 8; return true;
        case SG_CIGAR: *p = h->cigar.p; *bytes = (size_t)h->n_cig * 4; return true;
        case SG_SEQ: *p = h->seq.p; *bytes = (size_t)h->n_seq; return true;
        case SG_QUAL: *p = h->qual.p; *bytes = (size_t)h->n_qual; return true;
    }
    return false;
}

extern "C" {

void sg_default_params(sg_params *p) {
    memset(p, 0, sizeof *p);
    p->seed = 1; p->contig_len = 5000000; p->n_snp = 5000; p->clip_every = 7; p->coverage = 10.0; p->len_median = 15000.0; p->len_sigma = 0.7585;
    p->len_min = 1000; p->len_max = 200000; p->sub_rate = p->ins_rate = p->del_rate = 0.01; p->lowq_frac = 0.10; p->mapq0_frac = 0.01;
    p->secondary_frac = 0.003; p->dup_frac = 0.002; p->supp_frac = 0.02; p->supp_overlap_frac = 0.5; p->hpoly_every = 2000; p->clip_pileups = 0;
    p->snp_in_hpoly_frac = 0.05; p->snp_pair_frac = 0.01; p->read_seed = 0;
}

const char *sg_last_error(sg_handle *h) { return h ? h->err.c_str() : "null handle"; }

sg_handle *sg_create(int device, const sg_params *pp) {
    sg_handle *h = new sg_handle(); h->device = device; h->p = *pp;
    const sg_params &p = h->p;
    try {
        if (p.contig_len < 100000 || p.contig_len > 0x7ffffff0ll) throw std::string("contig_len out of range");
        if (p.n_snp < 1 || p.hpoly_every < 200 || p.len_min < 4 * BLK) throw std::string("bad parameters");
        SG_TRY(hipSetDevice(device));
        hipEvent_t e0, e1; SG_TRY(hipEventCreate(&e0)); SG_TRY(hipEventCreate(&e1)); SG_TRY(hipEventRecord(e0, nullptr));
        DevParams P{};
        const uint64_t s = mix64(p.seed * 0x9E3779B97F4A7C15ull + 0x1234567ull), rs = p.read_seed ? mix64(p.read_seed * 0x9E3779B97F4A7C15ull + 0x7654321ull) : s;
        P.s_ref = mix64(s ^ 1); P.s_hp = mix64(s ^ 2); P.s_var = mix64(s ^ 3); P.s_mol = mix64(rs ^ 4); P.s_err = mix64(rs ^ 5); P.s_sub = mix64(rs ^ 6); P.s_base = mix64(rs ^ 7); P.s_qual = mix64(rs ^ 8);
        P.L = p.contig_len; P.n_strata = p.n_snp; P.stratum = (double)(p.contig_len - 200) / (double)p.n_snp; P.W = p.hpoly_every;
        if (P.stratum < 4.0) throw std::string("too many SNPs for this contig");
        auto prob32 = [](double x) { x = std::min(1.0, std::max(0.0, x)); return (uint32_t)std::min(4294967295.0, x * 4294967296.0); };
        P.p_ins = prob32(p.ins_rate * BLK); P.p_del = prob32(std::min(p.del_rate * BLK, 1.0 - std::min(1.0, p.ins_rate * BLK)));
        P.p_sub = prob32(p.sub_rate); P.p_lowq = prob32(p.lowq_frac);
        P.snp_in_hpoly_frac = p.snp_in_hpoly_frac; P.snp_pair_frac = p.snp_pair_frac; P.len_median = p.len_median; P.len_sigma = p.len_sigma; P.len_min = p.len_min; P.len_max = p.len_max;
        P.mapq0_frac = p.mapq0_frac; P.secondary_frac = p.secondary_frac; P.dup_frac = p.dup_frac; P.supp_frac = p.supp_frac; P.supp_overlap_frac = p.supp_overlap_frac;
        P.clip_every = p.clip_every; P.n_pile = std::min(p.clip_pileups, 64);
        const double mean_len = p.len_median * std::exp(p.len_sigma * p.len_sigma / 2);
        P.n_mol = (long long)(p.coverage * (double)p.contig_len / mean_len);
        if (p.coverage > 0 && (P.n_mol < 1 || P.n_mol > 0x3fffffffll)) throw std::string("molecule count out of range");
        const long long L = P.L;
        Dev<char> tmp;
        // reference + variants
        h->ref.alloc((size_t)L); h->code.alloc((size_t)L);
        hipLaunchKernelGGL(k_ref, dim3((unsigned)((L / 32 + 256) / 256)), dim3(256), 0, nullptr, P, h->ref.p);
        hipLaunchKernelGGL(k_hpoly, dim3((unsigned)((L / P.W + 256) / 256)), dim3(256), 0, nullptr, P, h->ref.p);
        hipLaunchKernelGGL(k_hapcode, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, nullptr, P, h->ref.p, h->code.p);
        Dev<uint32_t> vcnt, voff; vcnt.alloc((size_t)P.n_strata + 1); voff.alloc((size_t)P.n_strata + 1);
        SG_TRY(hipMemsetAsync(vcnt.p, 0, ((size_t)P.n_strata + 1) * 4, nullptr));
        hipLaunchKernelGGL(k_var_count, dim3((P.n_strata + 255) / 256), dim3(256), 0, nullptr, P, vcnt.p);
        exscan32(vcnt.p, voff.p, (size_t)P.n_strata + 1, tmp);
        uint32_t nv = 0; SG_TRY(hipMemcpy(&nv, voff.p + P.n_strata, 4, hipMemcpyDeviceToHost));
        h->n_var = nv;
        h->vpos.alloc(nv); h->vref.alloc(nv); h->valt.alloc(nv); h->vhap.alloc(nv);
        hipLaunchKernelGGL(k_var_fill, dim3((P.n_strata + 255) / 256), dim3(256), 0, nullptr, P, voff.p, h->ref.p, h->code.p, h->vpos.p, h->vref.p, h->valt.p, h->vhap.p);
        if (p.coverage <= 0) {                                                  // reference + variant table only (the SNP-table broadcast of a multi-rank run)
            SG_TRY(hipDeviceSynchronize()); h->code.release();
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            return h;
        }
        // molecules -> alignments sorted by start
        Dev<uint32_t> mcnt, moff; mcnt.alloc((size_t)P.n_mol + 1); moff.alloc((size_t)P.n_mol + 1);
        SG_TRY(hipMemsetAsync(mcnt.p, 0, ((size_t)P.n_mol + 1) * 4, nullptr));
        hipLaunchKernelGGL(k_mol_count, dim3((unsigned)((P.n_mol + 255) / 256)), dim3(256), 0, nullptr, P, mcnt.p);
        exscan32(mcnt.p, moff.p, (size_t)P.n_mol + 1, tmp);
        uint32_t na = 0; SG_TRY(hipMemcpy(&na, moff.p + P.n_mol, 4, hipMemcpyDeviceToHost));
        if (na == 0) throw std::string("no alignments generated");
        h->n_aln = na;
        Dev<Aln> a0, a1; a0.alloc(na); a1.alloc(na);
        Dev<uint32_t> key, key_s, idx, idx_s; key.alloc(na); key_s.alloc(na); idx.alloc(na); idx_s.alloc(na);
        hipLaunchKernelGGL(k_mol_fill, dim3((unsigned)((P.n_mol + 255) / 256)), dim3(256), 0, nullptr, P, moff.p, a0.p, key.p, idx.p);
        { size_t need = 0; SG_TRY(rocprim::radix_sort_pairs(nullptr, need, key.p, key_s.p, idx.p, idx_s.p, (size_t)na, 0, 32, nullptr));
          if (need > tmp.n) tmp.alloc(need + 256);
          SG_TRY(rocprim::radix_sort_pairs(tmp.p, need, key.p, key_s.p, idx.p, idx_s.p, (size_t)na, 0, 32, nullptr)); }
        Dev<unsigned long long> nblk, blk_off; nblk.alloc((size_t)na + 1); blk_off.alloc((size_t)na + 1);
        SG_TRY(hipMemsetAsync(nblk.p, 0, ((size_t)na + 1) * 8, nullptr));
        hipLaunchKernelGGL(k_gather, dim3((na + 255) / 256), dim3(256), 0, nullptr, a0.p, idx_s.p, (long long)na, a1.p, nblk.p);
        exscan64(nblk.p, blk_off.p, (size_t)na + 1, tmp);
        unsigned long long tot_blk = 0; SG_TRY(hipMemcpy(&tot_blk, blk_off.p + na, 8, hipMemcpyDeviceToHost));
        a0.release(); key.release(); key_s.release(); idx.release(); idx_s.release(); mcnt.release(); moff.release();
        Dev<uint32_t> qoff; qoff.alloc((size_t)tot_blk);
        Dev<unsigned long long> ncig, nseq, nqual; ncig.alloc((size_t)na + 1); nseq.alloc((size_t)na + 1); nqual.alloc((size_t)na + 1);
        SG_TRY(hipMemsetAsync(ncig.p, 0, ((size_t)na + 1) * 8, nullptr)); SG_TRY(hipMemsetAsync(nseq.p, 0, ((size_t)na + 1) * 8, nullptr)); SG_TRY(hipMemsetAsync(nqual.p, 0, ((size_t)na + 1) * 8, nullptr));
        h->l_qseq.alloc(na); h->ref_start.alloc(na); h->flag.alloc(na); h->mapq.alloc(na); h->name_id.alloc(na); h->read_hap.alloc(na);
        h->cig_off.alloc((size_t)na + 1); h->seq_off.alloc((size_t)na + 1); h->qual_off.alloc((size_t)na + 1);
        hipLaunchKernelGGL(k_aln_sizes, dim3((na + 127) / 128), dim3(128), 0, nullptr, P, a1.p, (long long)na, blk_off.p, qoff.p, ncig.p, nseq.p, nqual.p, h->l_qseq.p);
        hipLaunchKernelGGL(k_headers, dim3((na + 255) / 256), dim3(256), 0, nullptr, a1.p, (long long)na, h->ref_start.p, h->flag.p, h->mapq.p, h->name_id.p, h->read_hap.p);
        exscan64(ncig.p, h->cig_off.p, (size_t)na + 1, tmp); exscan64(nseq.p, h->seq_off.p, (size_t)na + 1, tmp); exscan64(nqual.p, h->qual_off.p, (size_t)na + 1, tmp);
        SG_TRY(hipMemcpy(&h->n_cig, h->cig_off.p + na, 8, hipMemcpyDeviceToHost)); SG_TRY(hipMemcpy(&h->n_seq, h->seq_off.p + na, 8, hipMemcpyDeviceToHost));
        SG_TRY(hipMemcpy(&h->n_qual, h->qual_off.p + na, 8, hipMemcpyDeviceToHost));
        ncig.release(); nseq.release(); nqual.release();
        h->cigar.alloc((size_t)h->n_cig); h->seq.alloc((size_t)h->n_seq); h->qual.alloc((size_t)h->n_qual);
        hipLaunchKernelGGL(k_cigar, dim3((na + 127) / 128), dim3(128), 0, nullptr, P, a1.p, (long long)na, h->cig_off.p, h->cigar.p);
        hipLaunchKernelGGL(k_fill, dim3(na), dim3(256), 0, nullptr, P, a1.p, blk_off.p, qoff.p, h->l_qseq.p, h->seq_off.p, h->qual_off.p, h->code.p, h->seq.p, h->qual.p);
        SG_TRY(hipEventRecord(e1, nullptr)); SG_TRY(hipEventSynchronize(e1));
        float ms = 0; SG_TRY(hipEventElapsedTime(&ms, e0, e1)); h->gen_ms = ms;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        SG_TRY(hipGetLastError());
        h->code.release();
    } catch (std::string &e) { h->err = e; fprintf(stderr, "sg_create: %s\n", e.c_str()); delete h; return nullptr; }
    return h;
}

void sg_destroy(sg_handle *h) { if (h) { (void)hipSetDevice(h->device); delete h; } }
int64_t sg_n_reads(sg_handle *h) { return h->n_aln; }
int64_t sg_n_variants(sg_handle *h) { return h->n_var; }
int64_t sg_n_cigar(sg_handle *h) { return (int64_t)h->n_cig; }
int64_t sg_n_seq(sg_handle *h) { return (int64_t)h->n_seq; }
int64_t sg_n_qual(sg_handle *h) { return (int64_t)h->n_qual; }
double sg_gen_ms(sg_handle *h) { return h->gen_ms; }
int64_t sg_array_bytes(sg_handle *h, int which) { const void *p; size_t b; return sg_array(h, which, &p, &b) ? (int64_t)b : -1; }
const void *sg_dev_ptr(sg_handle *h, int which) { const void *p; size_t b; return sg_array(h, which, &p, &b) ? p : nullptr; }
// Drop the big per-base arrays (after lps_push_reads_device has taken its copy): keeps reference + variants + headers.
void sg_release_reads(sg_handle *h) { (void)hipSetDevice(h->device); h->cigar.release(); h->seq.release(); h->qual.release(); }
int sg_copy_to_host(sg_handle *h, int which, void *dst) {
    const void *p; size_t b;
    if (!sg_array(h, which, &p, &b) || (!p && b)) return -1;
    if (hipSetDevice(h->device) != hipSuccess) return -1;
    if (b && hipMemcpy(dst, p, b, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}

// ---- text files for the reference binary (host side; arrays copied back first)
int sg_write_fasta(sg_handle *h, const char *path, const char *chr) {
    const int64_t L = h->p.contig_len; std::vector<char> ref((size_t)L);
    if (sg_copy_to_host(h, SG_REF, ref.data())) return -1;
    FILE *f = fopen(path, "w"); if (!f) return -1;
    fprintf(f, ">%s\n", chr);
    std::vector<char> out; out.reserve((size_t)L + (size_t)L / 60 + 2);
    for (int64_t i = 0; i < L; i += 60) { out.insert(out.end(), ref.begin() + i, ref.begin() + std::min<int64_t>(L, i + 60)); out.push_back('\n'); }
    fwrite(out.data(), 1, out.size(), f); fclose(f);
    std::string fai = std::string(path) + ".fai"; f = fopen(fai.c_str(), "w"); if (!f) return -1;
    fprintf(f, "%s\t%lld\t%zu\t60\t61\n", chr, (long long)L, strlen(chr) + 2); fclose(f);
    return 0;
}
// phased==0: GT 0/1 (input of `phase`); phased==1: truth a|b with one PS (input of `haplotag`)
int sg_write_vcf(sg_handle *h, const char *path, const char *chr, int phased) {
    const size_t nv = (size_t)h->n_var; std::vector<int32_t> pos(nv); std::vector<uint8_t> r(nv), a(nv), hp(nv);
    if (sg_copy_to_host(h, SG_VAR_POS, pos.data()) || sg_copy_to_host(h, SG_VAR_REF, r.data()) || sg_copy_to_host(h, SG_VAR_ALT, a.data()) || sg_copy_to_host(h, SG_VAR_HAP, hp.data())) return -1;
    FILE *f = fopen(path, "w"); if (!f) return -1;
    fprintf(f, "##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n##contig=<ID=%s,length=%lld>\n", chr, (long long)h->p.contig_len);
    fprintf(f, "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">\n");
    if (phased) fprintf(f, "##FORMAT=<ID=PS,Number=1,Type=Integer,Description=\"Phase set identifier\">\n");
    fprintf(f, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n");
    for (size_t i = 0; i < nv; ++i) {
        if (!phased) fprintf(f, "%s\t%d\t.\t%c\t%c\t30\tPASS\t.\tGT:GQ\t0/1:30\n", chr, pos[i] + 1, r[i], a[i]);
        else fprintf(f, "%s\t%d\t.\t%c\t%c\t30\tPASS\t.\tGT:GQ:PS\t%s:30:%d\n", chr, pos[i] + 1, r[i], a[i], hp[i] ? "0|1" : "1|0", pos[0] + 1);
    }
    fclose(f); return 0;
}
int sg_write_sam(sg_handle *h, const char *path, const char *chr, int n_threads) {
    const size_t n = (size_t)h->n_aln;
    std::vector<int32_t> rs(n), lq(n); std::vector<uint16_t> fl(n); std::vector<uint8_t> mq(n); std::vector<uint32_t> nm(n);
    std::vector<uint64_t> co(n + 1), so(n + 1), qo(n + 1); std::vector<uint32_t> cg((size_t)h->n_cig); std::vector<uint8_t> sq((size_t)h->n_seq), ql((size_t)h->n_qual);
    if (sg_copy_to_host(h, SG_REF_START, rs.data()) || sg_copy_to_host(h, SG_L_QSEQ, lq.data()) || sg_copy_to_host(h, SG_FLAG, fl.data()) || sg_copy_to_host(h, SG_MAPQ, mq.data()) ||
        sg_copy_to_host(h, SG_NAME_ID, nm.data()) || sg_copy_to_host(h, SG_CIGAR_OFF, co.data()) || sg_copy_to_host(h, SG_SEQ_OFF, so.data()) || sg_copy_to_host(h, SG_QUAL_OFF, qo.data()) ||
        sg_copy_to_host(h, SG_CIGAR, cg.data()) || sg_copy_to_host(h, SG_SEQ, sq.data()) || sg_copy_to_host(h, SG_QUAL, ql.data())) return -1;
    FILE *f = fopen(path, "w"); if (!f) return -1;
    fprintf(f, "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:%s\tLN:%lld\n", chr, (long long)h->p.contig_len);
    static const char *nt = "=ACMGRSVTWYHKDBN"; static const char *ops = "MIDNSHP=XB";
    const size_t CH = 2048; const int nt_ = std::max(1, n_threads);
    for (size_t c0 = 0; c0 < n; c0 += CH * nt_) {
        std::vector<std::string> part((size_t)nt_); std::vector<std::thread> th;
        for (int t = 0; t < nt_; ++t) th.emplace_back([&, t] {
            std::string &s = part[(size_t)t]; char buf[160];
            const size_t a = std::min(n, c0 + CH * t), b = std::min(n, a + CH);
            for (size_t i = a; i < b; ++i) {
                s.append(buf, (size_t)snprintf(buf, sizeof buf, "r%09u\t%u\t%s\t%d\t%u\t", nm[i], fl[i], chr, rs[i] + 1, mq[i]));
                for (uint64_t c = co[i]; c < co[i + 1]; ++c) s.append(buf, (size_t)snprintf(buf, sizeof buf, "%u%c", cg[c] >> 4, ops[cg[c] & 15u]));
                s += "\t*\t0\t0\t";
                const int l = lq[i]; const uint8_t *sp = sq.data() + so[i], *qp = ql.data() + qo[i];
                const size_t at = s.size(); s.resize(at + 2 * (size_t)l + 2);
                char *d = &s[at];
                for (int j = 0; j < l; ++j) d[j] = nt[(sp[j >> 1] >> ((~j & 1) << 2)) & 15];
                d[l] = '\t';
                for (int j = 0; j < l; ++j) d[l + 1 + j] = (char)(33 + qp[j]);
                d[2 * l + 1] = '\n';
            }
        });
        for (auto &x : th) x.join();
        for (auto &s : part) if (!s.empty()) fwrite(s.data(), 1, s.size(), f);
    }
    fclose(f); return 0;
}

}  // extern "C"
