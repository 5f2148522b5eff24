// lps_bam.hip — device-side BAM record decode: the host hands over the INFLATED BAM byte
// stream of one contig plus the byte offset of every record; the GPU turns it into the read SoA the scoring kernels use.
//   k_bam_core   thread per record: fixed 32-byte core (SAM spec §4.2; htslib bam1_core_t as consumed by
//                src/phase/ParsingBam.cpp:1282-1299,1303-1316) -> ref_start/flag/mapq/l_qseq, CIGAR op count,
//                seq/qual byte offsets INTO the blob (4-bit seq and qual are used in place, no copy)
//   k_bam_cigar  wave per record: the CIGAR words are at arbitrary byte alignment inside a record; they are re-packed
//                into an aligned u32 array with two aligned dword loads + v_alignbyte per word
// Records are validated against the blob bounds before any kernel dereferences the offsets they imply.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "lps_bam.h"
#include "lps_common.h"

__device__ __forceinline__ uint32_t ld_u32_unaligned(const uint8_t *p) {
    const uintptr_t a = (uintptr_t)p; const unsigned sh = (unsigned)(a & 3u);
    const uint32_t *q = (const uint32_t *)(a & ~(uintptr_t)3);
    const uint32_t lo = q[0], hi = sh ? q[1] : 0u;
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}
__device__ __forceinline__ uint32_t ld_u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

__global__ void __launch_bounds__(256) k_bam_core(BamView B, int n, int at, int32_t *ref_start, int32_t *l_qseq, uint16_t *flag, uint8_t *mapq,
                                                  uint64_t *seq_off, uint64_t *qual_off, unsigned long long *cig_cnt, unsigned *err) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    if (i == n) { cig_cnt[n] = 0; return; }
    const uint64_t ro = B.rec_off[i];
    unsigned e = 0;
    if (ro < 4 || ro + 32 > B.push_bytes) { atomicOr(err, LPS_BAM_ERR_BOUNDS); cig_cnt[i] = 0; ref_start[at + i] = 0; l_qseq[at + i] = 0; flag[at + i] = 4; mapq[at + i] = 0; seq_off[at + i] = qual_off[at + i] = B.push_base; return; }
    const uint8_t *r = B.blob + B.push_base + ro;
    const uint32_t block_size = ld_u32_unaligned(r - 4);
    const int32_t pos = (int32_t)ld_u32_unaligned(r + 4);
    const uint32_t l_name = r[8], mq = r[9], n_cig = ld_u16(r + 12), fl = ld_u16(r + 14), l_seq = ld_u32_unaligned(r + 16);
    const uint64_t need = 32ull + l_name + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq;
    if (block_size < need || ro + block_size > B.push_bytes || (int32_t)l_seq < 0) e |= LPS_BAM_ERR_BOUNDS;
    if (i > 0) {                                                      // coordinate-sorted input (sam_itr order)
        const uint64_t rp = B.rec_off[i - 1];
        if (rp >= 4 && rp + 32 <= B.push_bytes && (int32_t)ld_u32_unaligned(B.blob + B.push_base + rp + 4) > pos) e |= LPS_BAM_ERR_UNSORTED;
        if (rp >= ro) e |= LPS_BAM_ERR_BOUNDS;
    }
    if (!e && n_cig == 2) {                                           // htslib's placeholder for >65535 ops: <l_seq>S<ref_len>N + CG:B,I tag
        const uint32_t c0 = ld_u32_unaligned(r + 32 + l_name), c1 = ld_u32_unaligned(r + 36 + l_name);
        if ((c0 & 15u) == 4u && (c0 >> 4) == l_seq && (c1 & 15u) == 3u) e |= LPS_BAM_ERR_CG_TAG;
    }
    if (e) atomicOr(err, e);
    const bool ok = (e & LPS_BAM_ERR_BOUNDS) == 0;
    ref_start[at + i] = pos; l_qseq[at + i] = ok ? (int32_t)l_seq : 0; flag[at + i] = (uint16_t)fl; mapq[at + i] = (uint8_t)mq;
    cig_cnt[i] = ok ? n_cig : 0;
    const uint64_t so = B.push_base + ro + 32 + l_name + 4ull * n_cig;
    seq_off[at + i] = ok ? so : B.push_base; qual_off[at + i] = ok ? so + (l_seq + 1ull) / 2 : B.push_base;
}

__global__ void __launch_bounds__(256) k_bam_cigar(BamView B, int n, const uint64_t *cigar_off /* [n+1], absolute */, uint32_t *cigar) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const uint64_t c0 = cigar_off[i]; const int n_cig = (int)(cigar_off[i + 1] - c0);
    if (n_cig == 0) return;
    const uint8_t *r = B.blob + B.push_base + B.rec_off[i];
    const uint8_t *src = r + 32 + r[8];
    for (int j = lane; j < n_cig; j += 64) cigar[c0 + j] = ld_u32_unaligned(src + 4ull * j);
}

void launch_bam_core(const BamView &B, int n, int at, int32_t *ref_start, int32_t *l_qseq, uint16_t *flag, uint8_t *mapq, uint64_t *seq_off,
                     uint64_t *qual_off, unsigned long long *cig_cnt, unsigned *err, hipStream_t s) {
    hipLaunchKernelGGL(k_bam_core, dim3((n + 1 + 255) / 256), dim3(256), 0, s, B, n, at, ref_start, l_qseq, flag, mapq, seq_off, qual_off, cig_cnt, err);
}
void launch_bam_cigar(const BamView &B, int n, const uint64_t *cigar_off, uint32_t *cigar, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_bam_cigar, dim3((n + 3) / 4), dim3(256), 0, s, B, n, cigar_off, cigar);
}

// cigar_off[at + i] = init + sum_{k<i} cig_cnt[k], i = 0..n  (cig_cnt[n] == 0)
void bam_cigar_offsets(DevBuf<char> &temp, size_t &temp_bytes, const unsigned long long *cig_cnt, uint64_t *cigar_off, int n, uint64_t init, hipStream_t s) {
    size_t need = 0;
    unsigned long long *out = reinterpret_cast<unsigned long long *>(cigar_off);
    HIP_TRY(rocprim::exclusive_scan(nullptr, need, cig_cnt, out, (unsigned long long)init, (size_t)n + 1, rocprim::plus<unsigned long long>(), s));
    if (need + 256 > temp_bytes) { temp.reserve(need + 256, s); temp_bytes = need + 256; }
    HIP_TRY(rocprim::exclusive_scan(temp.p, need, cig_cnt, out, (unsigned long long)init, (size_t)n + 1, rocprim::plus<unsigned long long>(), s));
}
