// Radix-2 number-theoretic transforms over the BLS12-381 scalar field Fr (2-adicity 32) and the
// batched polynomial products built on them.  The reference has no FFT: its Poly.Mul is
// schoolbook (algebra.go:92-105).  Here the NTT is only a *multiplication engine* -- the QAP
// lives on the reference's integer domain {1..n} (qap.go:42-55), see quotient.cuh.
//
// Forward = Gentleman-Sande DIF (natural in, bit-reversed out); inverse = Cooley-Tukey DIT
// (bit-reversed in, natural out).  Point-wise products happen in the bit-reversed domain, so no
// permutation pass exists.  A transform of 2^p points is cut into ceil(p/8) passes; each pass
// stages a tile of 2^k rows x COLS columns (<= 2048 Fr = 64 KB) in LDS, runs k butterfly
// stages there, and touches HBM exactly once for reading and once for writing.  Batched
// transforms (many blocks of 2^p points back to back) use the same kernel: butterflies never
// cross a 2^p boundary.
#pragma once
#include "field.cuh"

namespace ps {

PS_INL Fr fr_add(const Fr& a, const Fr& b) { return fe_add<FrParams>(a, b); }
PS_INL Fr fr_sub(const Fr& a, const Fr& b) { return fe_sub<FrParams>(a, b); }
PS_INL Fr fr_mul(const Fr& a, const Fr& b) { return fe_mul<FrParams>(a, b); }
PS_INL Fr fr_zero() { return fe_zero<FrParams>(); }
PS_INL Fr fr_one() { return fe_one<FrParams>(); }

__device__ __constant__ u32 c_fr_roots[33][8] = PS_FR_ROOTS;
__device__ __constant__ u32 c_fr_roots_inv[33][8] = PS_FR_ROOTS_INV;
__device__ __constant__ u32 c_fr_inv2pow[33][8] = PS_FR_INV2POW;

// tw[i] = w^i, w the primitive 2^log_tab-th root (or its inverse), i < 2^(log_tab-1)
__global__ void __launch_bounds__(256) k_twiddle_table(Fr* __restrict__ tw, int log_tab, int inverse) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << (log_tab - 1))) return;
    Fr acc = fr_one();
    for (int k = 0; k < log_tab - 1; k++) {
        if ((i >> k) & 1) {
            Fr r;
            const u32* src = inverse ? c_fr_roots_inv[log_tab - k] : c_fr_roots[log_tab - k];
#pragma unroll
            for (int j = 0; j < 8; j++) r.l[j] = src[j];
            acc = fr_mul(acc, r);
        }
    }
    tw[i] = acc;
}

constexpr int NTT_MAX_K = 8;        // butterfly stages per pass
constexpr int NTT_TILE_LOG = 11;    // 2048 Fr = 64 KB of LDS per workgroup

// One pass: k stages with half-distances D*2^m, m = 0..k-1 (descending m for the DIF forward,
// ascending for the DIT inverse).  Column q = (hi, lo) with lo = q mod D; element (t, q) lives
// at hi*D*2^k + t*D + lo.
template <bool INV>
__global__ void __launch_bounds__(256) k_ntt_pass(Fr* __restrict__ data, int logD, int k, int logCols,
                                                  const Fr* __restrict__ tw, int log_tab, int scale_log) {
    extern __shared__ __align__(16) unsigned char ntt_smem[];
    Fr* tile = reinterpret_cast<Fr*>(ntt_smem);
    const u32 COLS = 1u << logCols, rows = 1u << k;
    const u32 tile_elems = rows << logCols;
    const u64 q0 = (u64)blockIdx.x << logCols;
    const u64 Dm1 = (1ull << logD) - 1;
    for (u32 e = threadIdx.x; e < tile_elems; e += blockDim.x) {
        u32 t, col;
        if (logD >= logCols) { t = e >> logCols; col = e & (COLS - 1); }
        else { u32 lo = e & (u32)Dm1; t = (e >> logD) & (rows - 1); u32 hl = e >> (logD + k); col = (hl << logD) | lo; }
        u64 q = q0 + col;
        u64 addr = ((q >> logD) << (logD + k)) + ((u64)t << logD) + (q & Dm1);
        tile[t * COLS + col] = data[addr];
    }
    __syncthreads();
    const u32 nbf = (rows >> 1) << logCols;
    for (int st = 0; st < k; st++) {
        const int m = INV ? st : (k - 1 - st);
        const u32 mmask = (1u << m) - 1;
        for (u32 bf = threadIdx.x; bf < nbf; bf += blockDim.x) {
            u32 col = bf & (COLS - 1), r = bf >> logCols;
            u32 t0 = ((r >> m) << (m + 1)) | (r & mmask), t1 = t0 | (1u << m);
            u64 lo = (q0 + col) & Dm1;
            u64 j = ((u64)(t0 & mmask) << logD) + lo;
            Fr w = tw[j << (log_tab - 1 - (logD + m))];
            Fr a = tile[t0 * COLS + col], b = tile[t1 * COLS + col];
            if (!INV) {
                tile[t0 * COLS + col] = fr_add(a, b);
                tile[t1 * COLS + col] = fr_mul(fr_sub(a, b), w);
            } else {
                Fr bw = fr_mul(b, w);
                tile[t0 * COLS + col] = fr_add(a, bw);
                tile[t1 * COLS + col] = fr_sub(a, bw);
            }
        }
        __syncthreads();
    }
    Fr sc;
    if (scale_log >= 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) sc.l[j] = c_fr_inv2pow[scale_log][j];
    }
    for (u32 e = threadIdx.x; e < tile_elems; e += blockDim.x) {
        u32 t, col;
        if (logD >= logCols) { t = e >> logCols; col = e & (COLS - 1); }
        else { u32 lo = e & (u32)Dm1; t = (e >> logD) & (rows - 1); u32 hl = e >> (logD + k); col = (hl << logD) | lo; }
        u64 q = q0 + col;
        u64 addr = ((q >> logD) << (logD + k)) + ((u64)t << logD) + (q & Dm1);
        Fr v = tile[t * COLS + col];
        if (scale_log >= 0) v = fr_mul(v, sc);
        data[addr] = v;
    }
}

struct NttTables {
    Fr* fwd = nullptr;
    Fr* inv = nullptr;
    int log_tab = 0;
};

static inline int ilog2_ceil(u64 v) {
    int l = 0;
    while ((1ull << l) < v) l++;
    return l;
}

// (re)build the twiddle tables for transforms up to 2^log_size points
static inline hipError_t ntt_tables_ensure(NttTables& t, int log_size, hipStream_t st) {
    if (log_size < 1) log_size = 1;
    if (t.log_tab >= log_size) return hipSuccess;
    if (log_size > 32) return hipErrorInvalidValue;
    hipError_t e;
    if (t.fwd) { (void)hipStreamSynchronize(st); (void)hipFree(t.fwd); (void)hipFree(t.inv); t.fwd = t.inv = nullptr; t.log_tab = 0; }
    size_t entries = (size_t)1 << (log_size - 1);
    if ((e = hipMalloc((void**)&t.fwd, entries * sizeof(Fr))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&t.inv, entries * sizeof(Fr))) != hipSuccess) return e;
    unsigned blocks = (unsigned)((entries + 255) / 256);
    hipLaunchKernelGGL(k_twiddle_table, dim3(blocks), dim3(256), 0, st, t.fwd, log_size, 0);
    hipLaunchKernelGGL(k_twiddle_table, dim3(blocks), dim3(256), 0, st, t.inv, log_size, 1);
    t.log_tab = log_size;
    return hipGetLastError();
}

// `total` = batch * 2^p elements, contiguous.  Forward: natural -> bit-reversed (per block).
template <bool INV>
static inline hipError_t ntt_run(const NttTables& tb, hipStream_t st, Fr* data, u64 total, int p) {
    if (p == 0) return hipSuccess;
    if (p > tb.log_tab) return hipErrorInvalidValue;
    const int log_total = ilog2_ceil(total);
    int npass = (p + NTT_MAX_K - 1) / NTT_MAX_K;
    // stage groups of nearly equal size; DIF walks them from the top, DIT from the bottom
    int done = 0;
    for (int ps_i = 0; ps_i < npass; ps_i++) {
        int k = (p - done + (npass - ps_i) - 1) / (npass - ps_i);
        int logD = INV ? done : (p - done - k);
        int logCols = NTT_TILE_LOG - k;
        if (logCols > log_total - k) logCols = log_total - k;
        if (logCols < 0) logCols = 0;
        u64 cols_total = total >> k;
        unsigned grid = (unsigned)(cols_total >> logCols);
        size_t smem = ((size_t)sizeof(Fr) << k) << logCols;
        int scale_log = (INV && ps_i == npass - 1) ? p : -1;
        hipLaunchKernelGGL(k_ntt_pass<INV>, dim3(grid), dim3(256), smem, st, data, logD, k, logCols,
                           INV ? tb.inv : tb.fwd, tb.log_tab, scale_log);
        done += k;
    }
    return hipGetLastError();
}

// ---- element-wise helpers ----
// out[i] = a[i] * b[i]
__global__ void __launch_bounds__(256) k_fr_pointwise_mul(Fr* __restrict__ out, const Fr* __restrict__ a,
                                                          const Fr* __restrict__ b, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = fr_mul(a[i], b[i]);
}
// dst[i] = i < n_src ? src[i] : 0, for i < n_dst
__global__ void __launch_bounds__(256) k_fr_copy_pad(Fr* __restrict__ dst, const Fr* __restrict__ src, u64 n_src, u64 n_dst) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_dst) return;
    dst[i] = i < n_src ? src[i] : fr_zero();
}
// plain little-endian limbs <-> Montgomery form
__global__ void __launch_bounds__(256) k_fr_to_mont(Fr* __restrict__ dst, const u32* __restrict__ src, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr a;
#pragma unroll
    for (int j = 0; j < 8; j++) a.l[j] = src[8 * i + j];
    dst[i] = fe_to_mont<FrParams>(a);
}
__global__ void __launch_bounds__(256) k_fr_from_mont(u32* __restrict__ dst, const Fr* __restrict__ src, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr a = fe_from_mont<FrParams>(src[i]);
#pragma unroll
    for (int j = 0; j < 8; j++) dst[8 * i + j] = a.l[j];
}

static inline unsigned nblk(u64 n) { return (unsigned)((n + 255) / 256); }

// out[0..n_out) = (a * b)[0..n_out); a, b Montgomery coefficient vectors on the device.
// ta, tb: scratch of 2^ceil(log2(na+nb-1)) elements each.
static inline hipError_t poly_mul_dev(const NttTables& tabs, hipStream_t st, const Fr* a, u64 na, const Fr* b, u64 nb,
                                      Fr* out, u64 n_out, Fr* ta, Fr* tb) {
    if (na == 0 || nb == 0) {
        if (n_out) hipLaunchKernelGGL(k_fr_copy_pad, dim3(nblk(n_out)), dim3(256), 0, st, out, a, (u64)0, n_out);
        return hipGetLastError();
    }
    int p = ilog2_ceil(na + nb - 1);
    u64 S = 1ull << p;
    hipLaunchKernelGGL(k_fr_copy_pad, dim3(nblk(S)), dim3(256), 0, st, ta, a, na, S);
    hipLaunchKernelGGL(k_fr_copy_pad, dim3(nblk(S)), dim3(256), 0, st, tb, b, nb, S);
    hipError_t e;
    if ((e = ntt_run<false>(tabs, st, ta, S, p)) != hipSuccess) return e;
    if ((e = ntt_run<false>(tabs, st, tb, S, p)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_fr_pointwise_mul, dim3(nblk(S)), dim3(256), 0, st, ta, ta, tb, S);
    if ((e = ntt_run<true>(tabs, st, ta, S, p)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_fr_copy_pad, dim3(nblk(n_out)), dim3(256), 0, st, out, ta, S < n_out ? S : n_out, n_out);
    return hipGetLastError();
}

}  // namespace ps
