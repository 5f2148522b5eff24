// gather_bench.hip - what does a random single-byte gather cost on MI355X?  (calibration for k_extract_phase's seq / qual gathers: 31 M per chr1-50x
// launch over a 19 GB footprint).  hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip ; ./gather_bench
// Variants: (1) 1-byte loads, one random address per lane; (2) the same with the non-temporal hint; (3) two independent 1-byte loads per lane (two
// arrays: seq + qual); (4) ONE 2-byte load per lane (base + quality side by side); (5) 16-byte loads; (6) coalesced 16-byte stream for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <int MODE>
__global__ void k_gather(const uint8_t *a, const uint8_t *b, uint64_t bytes, uint64_t n, unsigned *sink) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (uint64_t k = i; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix(k * 2 + 1);
        const uint64_t off = h % bytes;
        if (MODE == 1) acc += a[off];
        else if (MODE == 2) acc += __builtin_nontemporal_load(a + off);
        else if (MODE == 3) { acc += a[off]; acc += b[(off >> 1)]; }
        else if (MODE == 4) acc += *reinterpret_cast<const uint16_t *>(a + (off & ~1ull));
        else if (MODE == 5) { const uint4 v = *reinterpret_cast<const uint4 *>(a + (off & ~15ull)); acc += v.x ^ v.y ^ v.z ^ v.w; }
        else if (MODE == 6) { const uint4 v = *reinterpret_cast<const uint4 *>(a + ((k * 16) % bytes)); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <int MODE> static void run(const char *name, const uint8_t *a, const uint8_t *b, uint64_t bytes, uint64_t n, unsigned *sink) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 16, threads = 256;
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(threads), 0, 0, a, b, bytes, n / 8, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(threads), 0, 0, a, b, bytes, n, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.3f ms  %7.2f G lane-loads/s\n", name, ms, n / (ms * 1e-3) / 1e9);
}
int main(int argc, char **argv) {
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 16ull) << 30;
    const uint64_t n = 64ull << 20;
    uint8_t *a, *b; unsigned *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes / 2 + 64)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 1, bytes / 2)); CK(hipDeviceSynchronize());
    printf("footprint %llu GiB, %llu M loads per launch\n", (unsigned long long)(bytes >> 30), (unsigned long long)(n >> 20));
    run<1>("1 B random", a, b, bytes, n, sink);
    run<2>("1 B random, non-temporal", a, b, bytes, n, sink);
    run<3>("1 B + 1 B random (two arrays)", a, b, bytes, n, sink);
    run<4>("2 B random (one array)", a, b, bytes, n, sink);
    run<5>("16 B random", a, b, bytes, n, sink);
    run<6>("16 B coalesced stream (1 GiB)", a, b, bytes, n, sink);
    return 0;
}
