// stream_bench.hip - how fast can one-wave workgroups stream "jobs" (a few KB each, back to back in memory) on MI355X?  Calibration for the walk of
// k_extract_phase / k_haplotag_stream: 158 k jobs of ~12.6 KB of CIGAR words at chr1-50x, one wave per job, 512 words (2 KB) per round.
// hipcc --offload-arch=gfx950 -O3 -o stream_bench stream_bench.hip ; ./stream_bench
// LAYOUT 0: a lane takes 32 contiguous bytes per round (two 16-byte loads, 32 bytes apart from its neighbour's) - the kernels' layout
// LAYOUT 1: two loads per round, each contiguous across the wave (lane l: bytes 16 l and 1024 + 16 l of the round's 2 KB)
// DEPTH: rounds in flight ahead of the one being summed (1 = the kernels' prefetch); DEPTH 0: every round of the job requested up front (jobs of 6 rounds)
// LDS: bytes of shared memory per workgroup (what caps the resident waves: 160 KB per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
template <int LAYOUT> __device__ __forceinline__ void req(const uint32_t *base, int round, int l, U4 &a, U4 &b) {
    const uint32_t *p = base + 512 * round;
    if (LAYOUT == 0) { a = *reinterpret_cast<const U4 *>(p + 8 * l); b = *reinterpret_cast<const U4 *>(p + 8 * l + 4); }
    else { a = *reinterpret_cast<const U4 *>(p + 4 * l); b = *reinterpret_cast<const U4 *>(p + 256 + 4 * l); }
}
template <int LAYOUT, int DEPTH, int JOBS_PER_WAVE>
__global__ __launch_bounds__(64) void k_stream(const uint32_t *data, int rounds, long long n_jobs, unsigned *sink, int lds_words) {
    extern __shared__ int s_tab[];
    const int l = threadIdx.x;
    unsigned acc = 0;
    for (int jj = 0; jj < JOBS_PER_WAVE; ++jj) {
    const long long job = (long long)blockIdx.x * JOBS_PER_WAVE + jj;
    if (job >= n_jobs) break;
    const uint32_t *base = data + job * 512ll * rounds + (job & 3);     // (word-aligned only, like CIGAR arrays)
    int carry = 0;
    if (DEPTH == 0) {
        U4 a[6], b[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) req<LAYOUT>(base, r, l, a[r], b[r]);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            int s = (a[r].x >> 4) + (a[r].y >> 4) + (a[r].z >> 4) + (a[r].w >> 4) + (b[r].x >> 4) + (b[r].y >> 4) + (b[r].z >> 4) + (b[r].w >> 4);
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(s, d); if (l >= d) s += t; }
            s_tab[(r * 64 + l) % lds_words] = carry + s;
            carry += __shfl(s, 63);
        }
    } else {
        constexpr int DD = DEPTH ? DEPTH : 1;
        U4 pa[DD], pb[DD];
#pragma unroll
        for (int d = 0; d < DD; ++d) req<LAYOUT>(base, min(d, rounds - 1), l, pa[d], pb[d]);
        for (int r = 0; r < rounds; ++r) {
            const U4 a = pa[0], b = pb[0];
#pragma unroll
            for (int d = 0; d + 1 < DD; ++d) { pa[d] = pa[d + 1]; pb[d] = pb[d + 1]; }
            req<LAYOUT>(base, min(r + DEPTH, rounds - 1), l, pa[DD - 1], pb[DD - 1]);
            int s = (a.x >> 4) + (a.y >> 4) + (a.z >> 4) + (a.w >> 4) + (b.x >> 4) + (b.y >> 4) + (b.z >> 4) + (b.w >> 4);
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(s, d); if (l >= d) s += t; }
            s_tab[(r * 64 + l) % lds_words] = carry + s;
            carry += __shfl(s, 63);
        }
    }
    acc += (unsigned)carry + (unsigned)s_tab[l % lds_words];
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <int LAYOUT, int DEPTH, int JPW> static void run(const char *name, const uint32_t *d, uint64_t bytes, int rounds, int lds, unsigned *sink) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long n_jobs = (long long)(bytes / (2048ull * rounds)) - 1;
    const unsigned grid = (unsigned)((n_jobs + JPW - 1) / JPW);
    CK(hipFuncSetAttribute((const void *)k_stream<LAYOUT, DEPTH, JPW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((k_stream<LAYOUT, DEPTH, JPW>), dim3(grid), dim3(64), lds, 0, d, rounds, n_jobs, sink, lds / 4);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_stream<LAYOUT, DEPTH, JPW>), dim3(grid), dim3(64), lds, 0, d, rounds, n_jobs, sink, lds / 4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s lds %6d  %8.3f ms  %6.2f TB/s\n", name, lds, ms, (double)n_jobs * rounds * 2048 / (ms * 1e-3) / 1e12);
}
int main(int argc, char **argv) {
    const uint64_t bytes = 2ull << 30;
    uint32_t *d; unsigned *sink;
    CK(hipMalloc(&d, bytes + 4096)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(d, 0x11, bytes + 4096)); CK(hipDeviceSynchronize());
    printf("2 GiB of jobs, one wave per job, 6 rounds of 2 KB per job unless said otherwise\n");
    const int ldss[] = {8704, 4608, 2304, 1024};                        // ~4, 8 (the cap of waves per SIMD is 8) ...
    for (int lds : ldss) {
        run<0, 1, 1>("32 B per lane, 1 round ahead", d, bytes, 6, lds, sink);
        run<0, 2, 1>("32 B per lane, 2 rounds ahead", d, bytes, 6, lds, sink);
        run<0, 0, 1>("32 B per lane, all 6 rounds up front", d, bytes, 6, lds, sink);
        run<1, 1, 1>("2 x 16 B coalesced, 1 round ahead", d, bytes, 6, lds, sink);
        run<1, 2, 1>("2 x 16 B coalesced, 2 rounds ahead", d, bytes, 6, lds, sink);
        run<1, 0, 1>("2 x 16 B coalesced, all 6 rounds up front", d, bytes, 6, lds, sink);
    }
    run<0, 1, 8>("32 B per lane, 1 ahead, 8 jobs per wave (persistent-ish)", d, bytes, 6, 8704, sink);
    run<0, 2, 8>("32 B per lane, 2 ahead, 8 jobs per wave", d, bytes, 6, 8704, sink);
    run<0, 1, 1>("32 B per lane, 1 ahead, jobs of 48 rounds", d, bytes, 48, 8704, sink);
    run<0, 2, 1>("32 B per lane, 2 ahead, jobs of 48 rounds", d, bytes, 48, 8704, sink);
    run<0, 1, 1>("32 B per lane, 1 ahead, jobs of 2 rounds", d, bytes, 2, 8704, sink);
    return 0;
}
