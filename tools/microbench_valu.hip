// Integer-VALU issue-rate microbenchmark for gfx950 (MI355X).
// The MSM is integer-ALU bound; the guides give no v_mad_u64_u32 rate, so it is
// measured here.  For each instruction: 8 independent dependency chains per lane, WAVES waves
// per SIMD on every CU, result in wave-instructions per cycle per SIMD at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;   // loop trips
constexpr int UNROLL = 8;     // instructions per chain per trip
constexpr int CHAINS = 8;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(256) bench(uint64_t* out, uint32_t seed, uint64_t* clk) {
    uint32_t a = seed * (threadIdx.x + 1) | 1, b = seed ^ (threadIdx.x * 2654435761u) | 1;
    uint64_t acc[CHAINS];
    uint32_t lo[CHAINS], hi[CHAINS];
    double d[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { acc[c] = a + c; lo[c] = a + c; hi[c] = b + c; d[c] = 1.0 + c; }
    double da = 1.0000001, db = 0.9999999;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
                if (KIND == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[c]) : "v"(a));
                if (KIND == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo[c]) : "v"(a));
                if (KIND == 3) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(lo[c]) : "v"(a) : "vcc");
                if (KIND == 4) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo[c]) : "v"(a) : "vcc");
                if (KIND == 5) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
                if (KIND == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(lo[c]) : "v"(hi[c]));
                if (KIND == 7) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(da), "v"(db));
                if (KIND == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo[c]) : "v"(a), "v"(b));
                if (KIND == 9) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo[c]) : "v"(a));
                if (KIND == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[c]) : "v"(a));
                if (KIND == 11) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(acc[c]), "+v"(hi[c]) : "v"(a), "v"(b) : "vcc");
                if (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo[c]) : "v"(a) : "vcc");
                if (KIND == 13) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(lo[c]) : "v"(a), "v"(b));
                if (KIND == 14) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(lo[c]) : "v"(a));
                if (KIND == 15) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[c]) : "v"(da));
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += acc[c] + lo[c] + hi[c] + (uint64_t)d[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(uint64_t*, uint32_t, uint64_t*);
struct Entry { const char* name; kern_t k; int instr_per_slot; };

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    Entry entries[] = {
        {"v_mad_u64_u32", bench<0>, 1}, {"v_mul_lo_u32", bench<1>, 1}, {"v_mul_hi_u32", bench<2>, 1},
        {"v_add_co_u32", bench<3>, 1}, {"v_addc_co_u32", bench<4>, 1}, {"v_lshl_add_u64", bench<5>, 1},
        {"v_mov_b32", bench<6>, 1}, {"v_fma_f64", bench<7>, 1}, {"v_mad_u32_u24", bench<8>, 1},
        {"v_mul_hi_u32_u24", bench<9>, 1}, {"v_add_u32", bench<10>, 1}, {"mad_u64+addc pair", bench<11>, 2},
        {"v_cndmask_b32", bench<12>, 1}, {"v_mad_i32_i24", bench<13>, 1}, {"v_alignbit_b32", bench<14>, 1},
        {"v_mul_f64", bench<15>, 1},
    };
    uint64_t *out, *clk;
    const int max_blocks = cus * 8;
    CHECK(hipMalloc(&out, sizeof(uint64_t) * max_blocks * 256));
    CHECK(hipMalloc(&clk, sizeof(uint64_t) * max_blocks));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    uint64_t* hclk = (uint64_t*)malloc(sizeof(uint64_t) * max_blocks);
    printf("%-20s %6s %12s %14s %16s %14s\n", "instr", "w/SIMD", "ms", "Ginstr/s(lane)", "cyc/waveinstr", "memtime/instr");
    for (auto& e : entries) {
        for (int wps : {1, 2, 4}) {            // waves per SIMD: blocks of 256 threads = 4 waves = 1 wave/SIMD
            int blocks = cus * wps;
            e.k<<<blocks, 256>>>(out, 12345u, clk);  // warmup
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            e.k<<<blocks, 256>>>(out, 12345u, clk);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(hclk, clk, sizeof(uint64_t) * blocks, hipMemcpyDeviceToHost));
            double avgclk = 0;
            for (int i = 0; i < blocks; i++) avgclk += (double)hclk[i];
            avgclk /= blocks;
            double winstr_per_wave = (double)ITERS * UNROLL * CHAINS * e.instr_per_slot;
            double lane_ops = winstr_per_wave * 64.0 * 4.0 * blocks;
            // s_memtime ticks at a constant 100 MHz; wave-instr issue cycles per SIMD estimated at 2.4 GHz
            double secs = ms * 1e-3;
            double cyc_per_winstr = secs * 2.4e9 / (winstr_per_wave * wps);
            printf("%-20s %6d %12.4f %14.2f %16.3f %14.4f\n", e.name, wps, ms, lane_ops / secs * 1e-9, cyc_per_winstr,
                   avgclk / (winstr_per_wave));
        }
    }
    return 0;
}
