// lps_reads.hip — the resident layout of the alignments' CIGAR words (gfx950).
//
// The stream walks of k_extract_phase / k_haplotag_stream / k_extra_find take a job's CIGAR words 8 per lane.  Resident words therefore lie in
// LANE-CHUNKS: every alignment's words start on a multiple of 8 words and are padded to one with op P, length 0 (6u: consumes nothing, neither a
// clip nor an op the reference rejects).  An alignment then begins and ends ON a lane-chunk - its stream coordinates are table entries, no partial
// sums, no blanking of a tail, no separate loads for its first and last chunk.  +0.5 % of words at ONT read lengths.
//
// This is the ONLY copy of the words in the context and it is written by whatever makes the alignments resident: a push copies a batch's words into
// the context anyway (lps_push_reads: after the upload; lps_push_reads_device: instead of the device-to-device copy it used to be; the BAM record
// decoder k_bam_cigar, lps_bam.hip: it re-aligns every word as it is), so the padding costs no pass of its own and lps_phase_chromosome /
// lps_haplotag_chromosome start from the resident arrays with nothing to prepare.  Bases and qualities are NOT re-laid: the kernels gather the
// base and its quality in place from the BAM record's own encodings (ReadView::seq / qual; two lines of HBM per site).  Round 3 kept an
// interleaved copy of both (one line per site, 0.13 ms less in k_extract_phase at chr1-50x) that took 21 ms per chromosome to build - a loss for
// every caller that phases a chromosome once, which is every caller (src/phase/PhasingProcess.cpp:113-173).
#include <algorithm>

#include "lps_kernels.h"

// chunks and words of every alignment of a batch (entry n: 0 chunks, so that an exclusive scan over n + 1 entries ends with the total)
__global__ void k_cp_count(int n, const uint64_t *cigar_off, uint32_t *nch, int32_t *ncig, unsigned *too_long) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { nch[i] = 0u; return; }
    const uint64_t w = cigar_off[i + 1] - cigar_off[i];
    if (w > 0x7fffffffull) { atomicOr(too_long, 1u); nch[i] = 0u; ncig[i] = 0; return; }
    nch[i] = (uint32_t)((w + 7) >> 3); ncig[i] = (int32_t)w;
}
// wave per alignment: words [cigar_off[r], cigar_off[r + 1]) of `cigar` -> chunks [chunk_base + rel_off[r], chunk_base + rel_off[r + 1]) of `cigp`,
// coalesced both ways; cp_off[r] (and cp_off[n]) = the absolute chunk offsets
__global__ __launch_bounds__(256) void k_cp_pack(int n, const uint64_t *cigar_off, const uint32_t *cigar, const uint32_t *rel_off, uint32_t chunk_base,
                                                 uint32_t *cp_off, uint32_t *cigp) {
    const int l = lane_id();
    if (blockIdx.x == 0 && threadIdx.x == 0) cp_off[n] = chunk_base + rel_off[n];
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += gridDim.x * 4) {
        const uint32_t c0 = chunk_base + rel_off[r], nw = (rel_off[r + 1] - rel_off[r]) * 8u;
        const uint64_t o = cigar_off[r]; const uint32_t nreal = (uint32_t)(cigar_off[r + 1] - o);
        if (l == 0) cp_off[r] = c0;
        uint32_t *dst = cigp + (size_t)c0 * 8;
        for (uint32_t i = l; i < nw; i += 64u) dst[i] = i < nreal ? cigar[o + i] : 6u;
    }
}
void launch_cp_count(int n, const uint64_t *cigar_off, uint32_t *nch, int32_t *ncig, unsigned *too_long, hipStream_t s) {
    hipLaunchKernelGGL(k_cp_count, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, s, n, cigar_off, nch, ncig, too_long);
}
void launch_cp_pack(int n, const uint64_t *cigar_off, const uint32_t *cigar, const uint32_t *rel_off, uint32_t chunk_base, uint32_t *cp_off, uint32_t *cigp, hipStream_t s) {
    hipLaunchKernelGGL(k_cp_pack, dim3((unsigned)std::max(1, std::min(65536, (n + 3) / 4))), dim3(256), 0, s, n, cigar_off, cigar, rel_off, chunk_base, cp_off, cigp);
}

// sum of n non-negative ints added to *out (the real CIGAR words of a BAM-record push: what the algorithmic-byte count is taken from)
__global__ void k_sum_i32(const int32_t *v, int n, unsigned long long *out) {
    unsigned long long t = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) t += (unsigned long long)max(v[i], 0);
    t = wave_sum(t);
    if (lane_id() == 0 && t) atomicAdd(out, t);
}
void launch_sum_i32(const int32_t *v, int n, unsigned long long *out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_sum_i32, dim3((unsigned)std::min(1024, (n + 255) / 256)), dim3(256), 0, s, v, n, out);
}
