// lps_reads.hip — the resident layout of the per-base data the phase extraction gathers from (gfx950).
//
// get_snp (src/phase/ParsingBam.cpp:1445-1520) needs, per (alignment, variant) observation, ONE base and ONE base quality of the read - two single
// bytes somewhere in 19 GB of SEQ / QUAL at 50x chr1.  A random byte costs a whole 128-byte line of HBM traffic whatever is asked for
// (profiles/micro/gather_bench.hip: 47 G random lines/s chip-wide = the HBM rate at 128 B each; one- and sixteen-byte loads cost the same), and with
// the BAM record's own layout - all 4-bit bases of a read, then all its qualities - the base and its quality always sit in two different lines: 31 M
// line fetches per chr1-50x launch, twice the bytes of the CIGAR stream.  Resident reads therefore keep both in ONE line: a read's bases and
// qualities are cut into blocks of 84 bases, 128 bytes each:
//     bytes 0..83   quality of bases 84 b .. 84 b + 83
//     bytes 84..125 the same bases, 4 bits each as in the BAM record (high nibble = even index)
//     bytes 126,127 unused
// (1.524 bytes per base instead of 1.5).  k_extract_phase / k_extract_redo read the base and then the quality from the same block: one line from
// HBM, the second load hits it.  Built once per chromosome after the pushes (lps_prepare_reads), part of loading like the copies themselves.
#include "lps_kernels.h"

__global__ void k_sq_count(int n, const int32_t *l_qseq, uint32_t *nblk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) nblk[i] = i < n ? (uint32_t)((max(l_qseq[i], 0) + LPS_SQ_BASES - 1) / LPS_SQ_BASES) : 0u;
}

// wave per read: lane i of a round writes dword (i & 31) of block (i >> 5); neighbouring lanes write neighbouring dwords and read neighbouring bytes
__global__ __launch_bounds__(256) void k_sq_pack(int n, const int32_t *l_qseq, const uint64_t *seq_off, const uint64_t *qual_off, const uint8_t *seq, const uint8_t *qual,
                                                 const uint32_t *blk, uint8_t *sq) {
    const int l = lane_id();
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += gridDim.x * 4) {
        const int lq = l_qseq[r];
        if (lq <= 0) continue;
        const uint8_t *s = seq + seq_off[r], *q = qual + qual_off[r];
        const uint32_t b0 = blk[r], nb = blk[r + 1] - b0;
        uint32_t *dst = reinterpret_cast<uint32_t *>(sq + (size_t)b0 * 128);
        const int n_sb = (lq + 1) >> 1;
        for (uint32_t i = l; i < nb * 32u; i += 64u) {
            const int b = (int)(i >> 5), w = (int)(i & 31u), base = b * LPS_SQ_BASES;
            uint32_t val = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = 4 * w + k; uint32_t byte = 0;
                if (t < LPS_SQ_BASES) { const int qi = base + t; if (qi < lq) byte = q[qi]; }
                else if (t < LPS_SQ_BASES + LPS_SQ_BASES / 2) { const int sb = (base >> 1) + (t - LPS_SQ_BASES); if (sb < n_sb) byte = s[sb]; }
                val |= byte << (8 * k);
            }
            dst[i] = val;
        }
    }
}

void launch_sq_count(int n, const int32_t *l_qseq, uint32_t *nblk, hipStream_t s) {
    hipLaunchKernelGGL(k_sq_count, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, s, n, l_qseq, nblk);
}
void launch_sq_pack(const ReadView &R, const uint32_t *blk, uint8_t *sq, hipStream_t s) {
    if (R.n == 0) return;
    hipLaunchKernelGGL(k_sq_pack, dim3((unsigned)std::min(65536, (R.n + 3) / 4)), dim3(256), 0, s, R.n, R.l_qseq, R.seq_off, R.qual_off, R.seq, R.qual, blk, sq);
}

// ---- CIGAR words in lane-chunks: every alignment's words start on a multiple of 8 words and are padded to one with op P, length 0 (6u: consumes
// nothing, neither a clip nor an op the reference rejects).  The stream walk of k_extract_phase / k_haplotag_stream takes 8 words per lane: with
// this layout an alignment begins and ends ON a lane-chunk - its stream coordinates are table entries, no partial sums, no blanking of a tail, no
// separate loads for its first and last chunk.  +0.5 % of words at ONT read lengths; built once per chromosome like the blocks above.
__global__ void k_cp_count(int n, const uint64_t *cigar_off, uint32_t *nch, int32_t *ncig, unsigned *too_long) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { nch[i] = 0u; return; }
    const uint64_t w = cigar_off[i + 1] - cigar_off[i];
    if (w > 0x7fffffffull) { atomicOr(too_long, 1u); nch[i] = 0u; ncig[i] = 0; return; }
    nch[i] = (uint32_t)((w + 7) >> 3); ncig[i] = (int32_t)w;
}
__global__ __launch_bounds__(256) void k_cp_pack(int n, const uint64_t *cigar_off, const uint32_t *cigar, const uint32_t *cp_off, uint32_t *cigp) {
    const int l = lane_id();
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += gridDim.x * 4) {
        const uint32_t c0 = cp_off[r], nw = (cp_off[r + 1] - c0) * 8u;
        const uint64_t o = cigar_off[r]; const uint32_t nreal = (uint32_t)(cigar_off[r + 1] - o);
        uint32_t *dst = cigp + (size_t)c0 * 8;
        for (uint32_t i = l; i < nw; i += 64u) dst[i] = i < nreal ? cigar[o + i] : 6u;
    }
}
void launch_cp_count(int n, const uint64_t *cigar_off, uint32_t *nch, int32_t *ncig, unsigned *too_long, hipStream_t s) {
    hipLaunchKernelGGL(k_cp_count, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, s, n, cigar_off, nch, ncig, too_long);
}
void launch_cp_pack(int n, const uint64_t *cigar_off, const uint32_t *cigar, const uint32_t *cp_off, uint32_t *cigp, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_cp_pack, dim3((unsigned)std::min(65536, (n + 3) / 4)), dim3(256), 0, s, n, cigar_off, cigar, cp_off, cigp);
}
