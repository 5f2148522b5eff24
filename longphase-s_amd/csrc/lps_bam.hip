// lps_bam.hip — device-side BAM record decode: the host hands over the INFLATED BAM byte
// stream of one contig plus the byte offset of every record; the GPU turns it into the read SoA the scoring kernels use.
//   k_bam_core   thread per record: fixed 32-byte core (SAM spec §4.2; htslib bam1_core_t as consumed by
//                src/phase/ParsingBam.cpp:1282-1299,1303-1316) -> ref_start/flag/mapq/l_qseq, CIGAR op count,
//                seq/qual byte offsets INTO the blob (4-bit seq and qual are used in place, no copy)
//   k_bam_cigar  wave per record: the CIGAR words are at arbitrary byte alignment inside a record; they are re-packed
//                with two aligned dword loads + v_alignbyte per word, straight into the resident lane-chunk layout
//                (every alignment padded to a multiple of 8 words, lps_reads.hip)
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

__device__ __forceinline__ uint32_t aux_len_dev(const uint8_t *p, const uint8_t *end) {       // bytes of one optional field, 0 = malformed
    if (p + 3 > end) return 0;
    const uint8_t t = p[2]; uint64_t v;
    if (t == 'A' || t == 'c' || t == 'C') v = 1; else if (t == 's' || t == 'S') v = 2; else if (t == 'i' || t == 'I' || t == 'f') v = 4; else if (t == 'd') v = 8;
    else if (t == 'Z' || t == 'H') { const uint8_t *q = p + 3; while (q < end && *q) ++q; if (q >= end) return 0; v = (uint64_t)(q - (p + 3)) + 1; }
    else if (t == 'B') { if (p + 8 > end) return 0; const uint8_t st = p[3]; const uint64_t cnt = ld_u32_unaligned(p + 4);
        const uint64_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0; if (!es) return 0; v = 5 + cnt * es; }
    else return 0;
    return (p + 3 + v <= end) ? (uint32_t)(3 + v) : 0u;
}

// CIGARs of more than 65535 operations (ultra-long ONT reads) do not fit the 16-bit n_cigar_op: the record then carries the placeholder
// <l_seq>S<ref_len>N and the real CIGAR in an optional field CG:B,I (SAM spec 4.2.2).  htslib moves it back when it reads a record (bam_tag2cigar,
// behind sam_itr_multi_next, src/phase/ParsingBam.cpp:1279) under exactly these conditions; -> offset of the field from the refID field (0: none)
__device__ __forceinline__ uint32_t find_cg_field(const uint8_t *r, uint32_t block_size, uint32_t l_name, uint32_t n_cig, uint32_t l_seq, int32_t tid, int32_t pos, uint32_t *count) {
    if (n_cig == 0 || tid < 0 || pos < 0) return 0;
    const uint32_t c0 = ld_u32_unaligned(r + 32 + l_name);
    if ((c0 & 15u) != 4u || (c0 >> 4) != l_seq) return 0;
    const uint8_t *p = r + 32 + l_name + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq, *end = r + block_size;
    while (p < end) {
        const uint32_t l = aux_len_dev(p, end); if (!l) return 0;
        if (p[0] == 'C' && p[1] == 'G') {                                // bam_aux_get: the first field of that name
            if (p[2] != 'B' || !(p[3] == 'I' || p[3] == 'i')) return 0;
            const uint32_t cnt = ld_u32_unaligned(p + 4);
            if (cnt < n_cig || cnt >= (1u << 29)) return 0;
            *count = cnt; return (uint32_t)(p - r);
        }
        p += l;
    }
    return 0;
}


__global__ void __launch_bounds__(256) k_bam_core(BamView B, int n, int at, int32_t *ref_start, int32_t *l_qseq, uint16_t *flag, uint8_t *mapq,
                                                  uint64_t *seq_off, uint64_t *qual_off, unsigned long long *cig_cnt /* lane-chunks of 8 words */, int32_t *cig_n /* words */,
                                                  uint64_t *cig_src, unsigned *err) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    if (i == n) { cig_cnt[n] = 0; return; }
    const uint64_t ro = B.rec_off[i];
    unsigned e = 0;
    if (ro < 4 || ro + 32 > B.push_bytes) { atomicOr(err,
            LPS_BAM_ERR_BOUNDS); cig_cnt[i] = 0; cig_n[at + i] = 0; cig_src[i] = B.push_base; ref_start[at + i] = 0; l_qseq[at + i] = 0; flag[at + i] = 4; mapq[at + i] = 0; seq_off[at + i] = qual_off[at + i] = B.push_base; return; }
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
    uint32_t n_real = n_cig; uint64_t src = B.push_base + ro + 32 + l_name;
    if (!e) {                                                         // the real CIGAR of a record with more than 65535 operations sits in its CG:B,I field
        uint32_t cnt = 0; const uint32_t cg = find_cg_field(r, block_size, l_name, n_cig, l_seq, (int32_t)ld_u32_unaligned(r), pos, &cnt);
        if (cg) { n_real = cnt; src = B.push_base + ro + cg + 8; }
    }
    if (e) atomicOr(err, e);
    const bool ok = (e & LPS_BAM_ERR_BOUNDS) == 0;
    ref_start[at + i] = pos; l_qseq[at + i] = ok ? (int32_t)l_seq : 0; flag[at + i] = (uint16_t)fl; mapq[at + i] = (uint8_t)mq;
    cig_cnt[i] = ok ? (n_real + 7u) >> 3 : 0; cig_n[at + i] = ok ? (int32_t)n_real : 0; cig_src[i] = src;
    const uint64_t so = B.push_base + ro + 32 + l_name + 4ull * n_cig;
    seq_off[at + i] = ok ? so : B.push_base; qual_off[at + i] = ok ? so + (l_seq + 1ull) / 2 : B.push_base;
}

__global__ void __launch_bounds__(256) k_bam_cigar(BamView B, int n, const uint64_t *chunk_off /* [n+1], absolute */, const int32_t *cig_n, const uint64_t *cig_src,
                                                   uint32_t *cigp, uint32_t *cp_off) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i > n) return;
    const uint64_t c0 = chunk_off[i];
    if (lane == 0) cp_off[i] = (uint32_t)c0;
    if (i == n) return;
    const int n_cig = cig_n[i], n_pad = (int)(chunk_off[i + 1] - c0) * 8;
    const uint8_t *src = B.blob + cig_src[i];                            // behind the read name, or inside the CG field
    uint32_t *dst = cigp + 8ull * c0;
    for (int j = lane; j < n_pad; j += 64) dst[j] = j < n_cig ? ld_u32_unaligned(src + 4ull * j) : 6u;   // (6u: op P, length 0)
}

void launch_bam_core(const BamView &B, int n, int at, int32_t *ref_start, int32_t *l_qseq, uint16_t *flag, uint8_t *mapq, uint64_t *seq_off,
                     uint64_t *qual_off, unsigned long long *cig_cnt, int32_t *cig_n, uint64_t *cig_src, unsigned *err, hipStream_t s) {
    hipLaunchKernelGGL(k_bam_core, dim3((n + 1 + 255) / 256), dim3(256), 0, s, B, n, at, ref_start, l_qseq, flag, mapq, seq_off, qual_off, cig_cnt, cig_n, cig_src, err);
}
void launch_bam_cigar(const BamView &B, int n, const uint64_t *chunk_off, const int32_t *cig_n, const uint64_t *cig_src, uint32_t *cigp, uint32_t *cp_off, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_bam_cigar, dim3((n + 1 + 3) / 4), dim3(256), 0, s, B, n, chunk_off, cig_n, cig_src, cigp, cp_off);
}

// cigar_off[i] = init + sum_{k<i} cig_cnt[k], i = 0..n  (cig_cnt[n] == 0); counts and offsets in lane-chunks
void bam_cigar_offsets(DevBuf<char> &temp, size_t &temp_bytes, const unsigned long long *cig_cnt, uint64_t *cigar_off, int n, uint64_t init, hipStream_t s) {
    size_t need = 0;
    unsigned long long *out = reinterpret_cast<unsigned long long *>(cigar_off);
    HIP_TRY(rocprim::exclusive_scan(nullptr, need, cig_cnt, out, (unsigned long long)init, (size_t)n + 1, rocprim::plus<unsigned long long>(), s));
    if (need + 256 > temp_bytes) { temp.reserve(need + 256, s); temp_bytes = need + 256; }
    HIP_TRY(rocprim::exclusive_scan(temp.p, need, cig_cnt, out, (unsigned long long)init, (size_t)n + 1, rocprim::plus<unsigned long long>(), s));
}

// ================================================================================================ record discovery in a resident BAM stream
// A BAM stream is a chain of records (block_size + body) that can only be walked serially.  On the GPU every byte position is tested against
// the NECESSARY conditions of a record start (fields in range, sizes consistent, name NUL-terminated), which true starts always pass and random
// payload bytes pass with negligible probability; the ordered candidate list is then VERIFIED to be exactly the chain (candidate i ends where
// candidate i+1 begins, the first is the known first record, the last ends at the end of the stream).  If the verification fails the chain is
// walked serially over the candidate list (k_bam_chain_serial), so the result is exact for any input.
__device__ __forceinline__ bool bam_candidate(const uint8_t *d, uint64_t p, uint64_t total, int32_t n_ref) {
    if (p + 36 > total) return false;
    const int32_t tid = (int32_t)ld_u32_unaligned(d + p + 4);
    if (tid < -1 || tid >= n_ref) return false;
    const uint32_t bs = ld_u32_unaligned(d + p);
    if (bs < 34 || bs > (1u << 29) || p + 4 + bs > total) return false;
    const int32_t mtid = (int32_t)ld_u32_unaligned(d + p + 24);
    if (mtid < -1 || mtid >= n_ref) return false;
    if ((int32_t)ld_u32_unaligned(d + p + 8) < -1) return false;
    const uint32_t l_name = d[p + 12], n_cig = ld_u16(d + p + 16), l_seq = ld_u32_unaligned(d + p + 20);
    if (l_name < 1 || (int32_t)l_seq < 0) return false;
    if (32ull + l_name + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq > bs) return false;
    return d[p + 36 + l_name - 1] == 0;
}

// positions k of 16 consecutive ones whose refID field (4 bytes at +4) is a contig index or -1; w = the six dwords from the aligned address at or
// before the first position's refID, SH = that refID's byte offset inside w[0]
template <int SH>
__device__ __forceinline__ uint32_t refid_mask(const uint32_t (&w)[6], int32_t n_ref) {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int32_t tid = (int32_t)__builtin_amdgcn_alignbyte(w[((SH + k) >> 2) + 1], w[(SH + k) >> 2], (SH + k) & 3);
        if (tid >= -1 && tid < n_ref) m |= 1u << k;
    }
    return m;
}

// 256 threads x 16 consecutive positions; PASS 0 counts per workgroup, PASS 1 writes the ordered candidates (value = offset of the refID field)
template <int PASS>
__global__ void __launch_bounds__(256) k_bam_candidates(const uint8_t *d, uint64_t begin, uint64_t total, int32_t n_ref, uint32_t *wg_count,
                                                        const uint32_t *wg_off, uint64_t *cand) {
    __shared__ uint32_t s_cnt[256];
    const uint64_t p0 = begin + ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    uint32_t mask = 0;
    if (p0 + 64 <= total) {
        // the refID test alone rejects nearly every position: the 19 bytes it looks at for the thread's 16 positions come in as six aligned dwords
        // (neighbouring threads: neighbouring 16 bytes), the 16 candidates' refIDs are cut out of them in registers; only survivors see the full test
        const uint64_t a = (p0 + 4) & ~3ull; const int sh = (int)((p0 + 4) & 3ull);        // sh is the same for every thread of the launch
        const uint32_t *q = reinterpret_cast<const uint32_t *>(d + a);
        const uint32_t w[6] = {q[0], q[1], q[2], q[3], q[4], q[5]};
        uint32_t pre = sh == 0 ? refid_mask<0>(w, n_ref) : sh == 1 ? refid_mask<1>(w, n_ref) : sh == 2 ? refid_mask<2>(w, n_ref) : refid_mask<3>(w, n_ref);
        while (pre) { const int k = __ffs(pre) - 1; pre &= pre - 1; if (bam_candidate(d, p0 + k, total, n_ref)) mask |= 1u << k; }
    } else {
        for (int k = 0; k < 16; ++k) if (p0 + k < total && bam_candidate(d, p0 + k, total, n_ref)) mask |= 1u << k;
    }
    s_cnt[threadIdx.x] = __popc(mask);
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) { const uint32_t v = threadIdx.x >= (unsigned)s ? s_cnt[threadIdx.x - s] : 0; __syncthreads(); s_cnt[threadIdx.x] += v; __syncthreads(); }
    if (PASS == 0) { if (threadIdx.x == 255) wg_count[blockIdx.x] = s_cnt[255]; return; }
    uint32_t at = wg_off[blockIdx.x] + s_cnt[threadIdx.x] - __popc(mask);
    for (int k = 0; k < 16; ++k) if (mask >> k & 1) cand[at++] = p0 + k + 4;
}

// flag[0] |= 1 when the candidates are not exactly the record chain
__global__ void __launch_bounds__(256) k_bam_chain_check(const uint8_t *d, const uint64_t *cand, uint32_t n, uint64_t first_ref_off, uint64_t total, unsigned *flag) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = cand[i]; const uint64_t next = r + ld_u32_unaligned(d + r - 4) + 4;   // refID offset of the following record
    bool ok = (i + 1 < n) ? cand[i + 1] == next : next - 4 == total;
    if (i == 0 && r != first_ref_off) ok = false;
    if (!ok) atomicOr(flag, 1u);
}

// exact fallback: walk the chain serially, keeping the candidates that are on it (binary search per hop); flag |= 2 when the chain leaves the list
__global__ void k_bam_chain_serial(const uint8_t *d, uint64_t *cand, uint32_t n, uint64_t first_ref_off, uint64_t total, uint32_t *n_out, unsigned *flag) {
    if (blockIdx.x || threadIdx.x) return;
    uint64_t r = first_ref_off; uint32_t out = 0, lo = 0;
    while (r - 4 < total) {
        uint32_t a = lo, b = n;
        while (a < b) { const uint32_t m = (a + b) >> 1; if (cand[m] < r) a = m + 1; else b = m; }
        if (a >= n || cand[a] != r) { atomicOr(flag, 2u); break; }
        cand[out++] = r; lo = a + 1;                                    // out <= a: in place
        r = r + ld_u32_unaligned(d + r - 4) + 4;
    }
    if (r - 4 != total) atomicOr(flag, 2u);
    *n_out = out;
}

// per record: contig id and name length (for the host's per-contig ranges and name ranking)
__global__ void __launch_bounds__(256) k_bam_tid_lname(const uint8_t *d, const uint64_t *cand, uint32_t n, int32_t *tid, uint32_t *l_name) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = cand[i];
    tid[i] = (int32_t)ld_u32_unaligned(d + r); l_name[i] = d[r + 8];
}
__global__ void __launch_bounds__(256) k_bam_names(const uint8_t *d, const uint64_t *cand, uint32_t n, const uint32_t *name_off, uint8_t *names) {
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const uint64_t r = cand[i]; const uint32_t l = d[r + 8], o = name_off[i];
    for (uint32_t k = lane; k < l; k += 64) names[o + k] = d[r + 32 + k];
}

int bam_scan_records(const uint8_t *d, uint64_t first_rec, uint64_t total /* end of the record range */, int32_t n_ref, DevBuf<uint64_t> &cand, DevBuf<uint32_t> &wg_cnt, DevBuf<uint32_t> &wg_off,
                     DevBuf<char> &temp, size_t &temp_bytes, unsigned *flag, uint32_t *n_out_d, uint64_t *n_records, hipStream_t s) {
    *n_records = 0;
    if (first_rec >= total) return 0;
    const uint64_t span = total - first_rec; const uint64_t n_wg64 = (span + 4095) / 4096;
    if (n_wg64 > 0x7fffffffull) return -2;
    const uint32_t n_wg = (uint32_t)n_wg64;
    wg_cnt.reserve(n_wg + 1, s); wg_off.reserve(n_wg + 1, s);
    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(unsigned), s));
    HIP_TRY(hipMemsetAsync(wg_cnt.p + n_wg, 0, sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_bam_candidates<0>, dim3(n_wg), dim3(256), 0, s, d, first_rec, total, n_ref, wg_cnt.p, (const uint32_t *)nullptr, (uint64_t *)nullptr);
    size_t need = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, need, wg_cnt.p, wg_off.p, 0u, (size_t)n_wg + 1, rocprim::plus<uint32_t>(), s));
    if (need + 256 > temp_bytes) { temp.reserve(need + 256, s); temp_bytes = need + 256; }
    HIP_TRY(rocprim::exclusive_scan(temp.p, need, wg_cnt.p, wg_off.p, 0u, (size_t)n_wg + 1, rocprim::plus<uint32_t>(), s));
    uint32_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, wg_off.p + n_wg, sizeof n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (n == 0) return -3;
    cand.reserve((size_t)n + 1, s);
    hipLaunchKernelGGL(k_bam_candidates<1>, dim3(n_wg), dim3(256), 0, s, d, first_rec, total, n_ref, wg_cnt.p, (const uint32_t *)wg_off.p, cand.p);
    hipLaunchKernelGGL(k_bam_chain_check, dim3((n + 255) / 256), dim3(256), 0, s, d, (const uint64_t *)cand.p, n, first_rec + 4, total, flag);
    unsigned f = 0;
    HIP_TRY(hipMemcpyAsync(&f, flag, sizeof f, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (f) {                                                            // some candidate is not a record: exact serial walk
        HIP_TRY(hipMemsetAsync(flag, 0, sizeof(unsigned), s));
        hipLaunchKernelGGL(k_bam_chain_serial, dim3(1), dim3(64), 0, s, d, cand.p, n, first_rec + 4, total, n_out_d, flag);
        HIP_TRY(hipMemcpyAsync(&f, flag, sizeof f, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&n, n_out_d, sizeof n, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (f) return -4;
    }
    *n_records = n;
    return 0;
}
void launch_bam_tid_lname(const uint8_t *d, const uint64_t *cand, uint32_t n, int32_t *tid, uint32_t *l_name, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_bam_tid_lname, dim3((n + 255) / 256), dim3(256), 0, s, d, cand, n, tid, l_name);
}
void launch_bam_names(const uint8_t *d, const uint64_t *cand, uint32_t n, const uint32_t *name_off, uint8_t *names, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_bam_names, dim3((n + 3) / 4), dim3(256), 0, s, d, cand, n, name_off, names);
}

// ================================================================================================ tagged-record stream (haplotag writer)
// HaplotagProcess.cpp:337-361 on the device: a scored record loses its first HP, PS and PQ optional field and, when tagged, gains HP:i PS:i PQ:i;
// every other record is copied untouched.  k_tag_sizes walks the optional fields (thread per record), k_tag_write copies bytes (wave per record).
// somatic_haplotag's tags (src/somatic_haplotag/SomaticHaplotagProcess.cpp:529-536: HP:Z with the haplotype's name - 1 2 3 4 1-1 1-2 2-1 2-2 -, PS:i only for a read with a phase set, PQ:i)
__device__ __forceinline__ uint32_t som_tag_len(uint32_t hp, int32_t ps) { return 3u + ((hp >= 5u && hp <= 8u) ? 3u : 1u) + 1u + (ps != -1 ? 7u : 0u) + 7u; }
__device__ __forceinline__ uint8_t som_tag_byte(uint32_t b, uint32_t hp, int32_t ps, int32_t pq) {
    const uint32_t sl = (hp >= 5u && hp <= 8u) ? 3u : 1u;
    if (b < 3u) return b == 0 ? (uint8_t)'H' : b == 1 ? (uint8_t)'P' : (uint8_t)'Z';
    b -= 3u;
    if (b < sl) { if (sl == 1u) return (hp >= 1u && hp <= 4u) ? (uint8_t)('0' + hp) : (uint8_t)'.'; const uint32_t k = hp - 5u; return b == 0 ? (uint8_t)('1' + (k >> 1)) : b == 1 ? (uint8_t)'-' : (uint8_t)('1' + (k & 1u)); }
    if (b == sl) return 0;
    b -= sl + 1u;
    if (ps != -1) { if (b < 7u) return b == 0 ? (uint8_t)'P' : b == 1 ? (uint8_t)'S' : b == 2 ? (uint8_t)'i' : (uint8_t)((uint32_t)ps >> (8 * (b - 3u))); b -= 7u; }
    return b == 0 ? (uint8_t)'P' : b == 1 ? (uint8_t)'Q' : b == 2 ? (uint8_t)'i' : (uint8_t)((uint32_t)pq >> (8 * (b - 3u)));
}
__global__ void __launch_bounds__(256) k_tag_sizes(const uint8_t *d, const uint64_t *rec, uint32_t n, const uint8_t *status, const uint8_t *hp, const int32_t *ps,
        int som, unsigned long long *new_len,
                                                   uint2 *spans /* [n][4] (offset from refID, length), sorted by offset, length 0 = none; [3] = the CG field,
                                                           re-appended at the end */, unsigned *err) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    if (i == n) { new_len[n] = 0; return; }
    const uint64_t r = rec[i]; const uint32_t bs = ld_u32_unaligned(d + r - 4);
    uint2 sp[3] = {{0, 0}, {0, 0}, {0, 0}}; uint2 cgs = {0, 0}; uint32_t removed = 0;
    const uint32_t l_name = d[r + 8], n_cig = ld_u16(d + r + 12), l_seq = ld_u32_unaligned(d + r + 16);
    {   // a record htslib would have read through bam_tag2cigar is written with its CG field LAST (bam_write1 re-creates placeholder + field), scored or not
        uint32_t cnt = 0; const uint32_t cg = find_cg_field(d + r, bs, l_name, n_cig, l_seq, (int32_t)ld_u32_unaligned(d + r), (int32_t)ld_u32_unaligned(d + r + 4), &cnt);
        if (cg) { cgs.x = cg; cgs.y = 8u + 4u * cnt; }
    }
    if (status[i] == 0) {
        const uint8_t *p = d + r + 32 + l_name + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq, *end = d + r + bs; int k = 0; bool seen[3] = {false, false, false};
        while (p < end) {
            const uint32_t l = aux_len_dev(p, end); if (!l) { atomicOr(err, 1u); break; }
            const int which = (p[0] == 'H' && p[1] == 'P') ? 0 : (p[0] == 'P' && p[1] == 'S') ? 1 : (p[0] == 'P' && p[1] == 'Q') ? 2 : -1;
            if (which >= 0 && !seen[which]) { seen[which] = true; sp[k].x = (uint32_t)(p - (d + r)); sp[k].y = l; ++k; removed += l; }   // encountered in offset order
            p += l;
        }
    }
    spans[(size_t)i * 4 + 0] = sp[0]; spans[(size_t)i * 4 + 1] = sp[1]; spans[(size_t)i * 4 + 2] = sp[2]; spans[(size_t)i * 4 + 3] = cgs;
    new_len[i] = 4ull + bs - removed + ((status[i] == 0 && hp[i]) ? (som ? som_tag_len(hp[i], ps[i]) : 21u) : 0u);
}

__global__ void __launch_bounds__(256) k_tag_write(const uint8_t *d, const uint64_t *rec, uint32_t n, const uint8_t *status, const uint8_t *hp, const int32_t *ps, const int32_t *pq,
                                                   int som, const unsigned long long *out_off, const uint2 *spans, uint8_t *out) {
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const uint64_t r = rec[i]; const uint32_t bs = ld_u32_unaligned(d + r - 4);
    const uint2 s0 = spans[(size_t)i * 4], s1 = spans[(size_t)i * 4 + 1], s2 = spans[(size_t)i * 4 + 2], cg = spans[(size_t)i * 4 + 3];
    const uint32_t kept = bs - s0.y - s1.y - s2.y - cg.y; const bool tag = status[i] == 0 && hp[i] != 0;
    const uint32_t tlen = !tag ? 0u : som ? som_tag_len(hp[i], ps[i]) : 21u, nbs = kept + tlen + cg.y;
    uint8_t *o = out + out_off[i];
    if (lane < 4) o[lane] = (uint8_t)(nbs >> (8 * lane));
    for (uint32_t j = lane; j < kept; j += 64) {
        // the dropped spans in offset order (the CG field can sit anywhere among them)
        uint32_t src = j;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            uint2 lo = {0xffffffffu, 0};                                  // the pass-th smallest offset among the spans in use
            const uint2 all[4] = {s0, s1, s2, cg};
            int taken = 0;
            for (int a = 0; a < 4; ++a) {
                if (!all[a].y) continue;
                int before = 0;
                for (int b2 = 0; b2 < 4; ++b2) if (all[b2].y && all[b2].x < all[a].x) ++before;
                if (before == pass) { lo = all[a]; taken = 1; }
            }
            if (taken && src >= lo.x) src += lo.y;
        }
        o[4 + j] = d[r + src];
    }
    if (tag && som) { if (lane < tlen) o[4 + kept + lane] = som_tag_byte(lane, hp[i], ps[i], pq[i]); }
    else if (tag && lane < 21) {                                          // addAuxiliaryTags: HP:i PS:i PQ:i, 7 bytes each
        const uint32_t f = lane / 7, b = lane % 7; const uint32_t val = f == 0 ? (uint32_t)hp[i] : f == 1 ? (uint32_t)ps[i] : (uint32_t)pq[i];
        const char *nm = f == 0 ? "HP" : f == 1 ? "PS" : "PQ";
        o[4 + kept + lane] = b < 2 ? (uint8_t)nm[b] : b == 2 ? (uint8_t)'i' : (uint8_t)(val >> (8 * (b - 3)));
    }
    for (uint32_t j = lane; j < cg.y; j += 64) o[4 + kept + tlen + j] = d[r + cg.x + j];   // ... and the CG field behind everything else
}

// -> total bytes of prefix + re-tagged records in `stream`; -1 when an optional field is malformed
int64_t bam_tag_stream(const uint8_t *d, const uint64_t *rec, uint32_t n, const uint8_t *status, const uint8_t *hp, const int32_t *ps, const int32_t *pq, int somatic_tags, uint64_t prefix_bytes,
                       DevBuf<unsigned long long> &new_len, DevBuf<unsigned long long> &out_off, DevBuf<uint2> &spans, DevBuf<uint8_t> &stream, DevBuf<char> &temp, size_t &temp_bytes,
                       unsigned *err, hipStream_t s) {
    new_len.reserve((size_t)n + 1, s); out_off.reserve((size_t)n + 1, s); spans.reserve((size_t)n * 4 + 4, s);
    HIP_TRY(hipMemsetAsync(err, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_tag_sizes, dim3((n + 256) / 256), dim3(256), 0, s, d, rec, n, status, hp, ps, somatic_tags, new_len.p, spans.p, err);
    size_t need = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, need, new_len.p, out_off.p, (unsigned long long)prefix_bytes, (size_t)n + 1, rocprim::plus<unsigned long long>(), s));
    if (need + 256 > temp_bytes) { temp.reserve(need + 256, s); temp_bytes = need + 256; }
    HIP_TRY(rocprim::exclusive_scan(temp.p, need, new_len.p, out_off.p, (unsigned long long)prefix_bytes, (size_t)n + 1, rocprim::plus<unsigned long long>(), s));
    unsigned long long total = 0; unsigned e = 0;
    HIP_TRY(hipMemcpyAsync(&total, out_off.p + n, sizeof total, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&e, err, sizeof e, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (e) return -1;
    stream.reserve((size_t)total + 64, s, true, prefix_bytes);
    if (n) hipLaunchKernelGGL(k_tag_write, dim3((n + 3) / 4), dim3(256), 0, s, d, rec, n, status, hp, ps, pq, somatic_tags, (const unsigned long long *)out_off.p, (const uint2 *)spans.p, stream.p);
    return (int64_t)total;
}
