// Standalone reproducer attempt for the lost writes of DESIGN.md section 5 (round 2: a kernel's stores to staging memory
// from hipMallocAsync came back as zeros for whole workgroups).  No library code: allocate from the stream-ordered pool,
// one kernel writes 224-byte records after some work, a second kernel reads them, free, synchronise -- all on ONE stream.
//   hipcc --offload-arch=gfx950 -O2 -o tools/malloc_async_repro tools/malloc_async_repro.hip
//   tools/malloc_async_repro        default pool (every synchronisation returns freed blocks to the OS: fresh mappings)
//   tools/malloc_async_repro k      hipMemPoolAttrReleaseThreshold = max (blocks stay in the pool)
//   tools/malloc_async_repro m      hipMemsetAsync of the fresh block before the first kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
struct Rec { uint4 v[14]; };  // 224 bytes: the size of the records the library staged (an XYZZ point of G1)
__global__ void k_write(Rec* tmp, unsigned n, int spin) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = i + 1;
    for (int s = 0; s < spin; s++) x = x * 1664525u + 1013904223u;  // the real kernel computes for ~a millisecond before it stores
    Rec r;
    for (int j = 0; j < 14; j++) r.v[j] = make_uint4(i + 1, j, x, ~i);
    tmp[i] = r;
}
__global__ void k_read(const Rec* tmp, unsigned n, unsigned* out) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = tmp[i].v[13].x;
}
int main(int argc, char** argv) {
    const bool keep = argc > 1 && argv[1][0] == 'k', clear = argc > 1 && argv[1][0] == 'm';
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (keep) {
        hipMemPool_t pool;
        uint64_t all = ~0ull;
        CK(hipDeviceGetDefaultMemPool(&pool, 0));
        CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &all));
    }
    unsigned* out;
    CK(hipMalloc(&out, 4 << 20));
    long bad = 0, total = 0;
    for (int rep = 0; rep < 30; rep++)
        for (unsigned n : {300u, 257u, 1000u, 5000u, 70000u}) {
            Rec* tmp;
            CK(hipMallocAsync((void**)&tmp, (size_t)n * sizeof(Rec) + (size_t)n * 56, st));
            if (clear) CK(hipMemsetAsync(tmp, 0, (size_t)n * sizeof(Rec), st));
            k_write<<<(n + 255) / 256, 256, 0, st>>>(tmp, n, 20000);
            k_read<<<(n + 255) / 256, 256, 0, st>>>(tmp, n, out);
            CK(hipFreeAsync(tmp, st));
            std::vector<unsigned> h(n);
            CK(hipMemcpyAsync(h.data(), out, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            for (unsigned i = 0; i < n; i++) { total++; bad += h[i] != i + 1; }
        }
    printf("mode %s: %ld of %ld records lost\n", keep ? "keep-in-pool" : clear ? "memset-first" : "default", bad, total);
    return bad ? 1 : 0;
}
