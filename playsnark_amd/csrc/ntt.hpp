// Radix-2 number-theoretic transforms over the BLS12-381 scalar field Fr (2-adicity 32) and the
// batched polynomial products built on them.  The reference has no FFT: its Poly.Mul is
// schoolbook (algebra.go:92-105).  Here the NTT is only a *multiplication engine* -- the QAP
// lives on the reference's integer domain {1..n} (qap.go:42-55), see quotient.hpp.
//
// Forward = Cooley-Tukey butterflies (a + w b, a - w b) on natural-order input, bit-reversed
// output; inverse = Gentleman-Sande butterflies (a + b, (a - b) w^-1) on bit-reversed input,
// natural output.  Point-wise products happen in the bit-reversed domain, so no permutation pass
// exists.  Two properties matter on this machine:
//   * all butterflies of a block share ONE twiddle (w = omega^bitrev(block)), so twiddle traffic
//     is a broadcast, not a gather;
//   * the forward transform only ever *adds* a product to a value, so with the lazy Fr
//     representation (field.hpp) values grow by ~r per stage and need no reduction at all; the
//     inverse doubles per stage along the all-sums path; the factor is divided out by the last pass
//     (and by an earlier one only if it would pass 2^16).
// A transform of 2^p points is cut into ceil(p/10) passes; each pass stages a tile of 2^k rows x
// COLS columns (2048 Fr = 80 KB) in LDS, runs k butterfly stages there, and touches HBM exactly
// once for reading and once for writing.  Batched transforms (many blocks of 2^p points back to
// back) use the same kernel: butterflies never cross a 2^p boundary.
#pragma once
#include <algorithm>

#include "field.hpp"

namespace ps {

__device__ __constant__ i32 c_fr_roots[33][10] = PS_FR28_ROOTS;
__device__ __constant__ i32 c_fr_roots_inv[33][10] = PS_FR28_ROOTS_INV;
__device__ __constant__ i32 c_fr_inv2pow[33][10] = PS_FR28_INV2POW;

// tw[i] = w^i, w the primitive 2^log_tab-th root (or its inverse), i < 2^(log_tab-1)
__global__ void __launch_bounds__(256) k_twiddle_table(Fr* __restrict__ tw, int log_tab, int inverse) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << (log_tab - 1))) return;
    Fr acc = fr_one();
    for (int k = 0; k < log_tab - 1; k++) {
        if ((i >> k) & 1) {
            Fr r;
            const i32* src = inverse ? c_fr_roots_inv[log_tab - k] : c_fr_roots[log_tab - k];
#pragma unroll
            for (int j = 0; j < FR_L; j++) r.l[j] = src[j];
            acc = fr_mul(acc, r);
        }
    }
    tw[i] = acc;
}

// Tile size, measured on the quotient at n = 2^20 (Groth16 route / values route): 2048 elements (80 KB, two 512-thread
// workgroups per CU, transforms of 2^20 in two passes) 11.3 / 2.9 ms; 1024 elements (40 KB, four 256-thread workgroups per
// CU, three passes) 10.6-10.9 / 2.65-2.7 ms; 512 elements 10.9 / 2.7 ms.  The passes are bound by their instruction count,
// not by HBM, so the extra pass costs less than the finer interleaving of load, butterfly and store phases gains.
// Waves per SIMD the pass kernels are compiled for.  Until late in round 4: four (128 registers per lane, four 40 KB tiles per
// CU), at the price of 10-14 spilled registers in k_ntt_mid and the inverse pass.  Three (the compiler then takes 137-139
// registers, nothing spilled): the quotient at 2^20 gates 8.96 -> 8.70 ms, 3-7 % at every size from 2^10 up (A/B on one box).
// With the field product as a single chain (field.hpp, fr_mul_chain) the kernels need 102-121 registers, so FOUR waves per SIMD
// are resident again -- without spills; the bound below only keeps the compiler from trading registers for them.
#ifndef PS_NTT_WAVES
#define PS_NTT_WAVES 3
#endif
constexpr int NTT_MAX_K = 9;        // butterfly stages per pass (a tile then keeps >= 2 contiguous columns = 80 B runs)
constexpr int NTT_TILE_LOG = 10;    // 1024 Fr = 40 KB of LDS per workgroup

// One pass over transforms of 2^p points: k stages with half-distances D*2^m, m = k-1..0 for the
// forward transform and 0..k-1 for the inverse.  Column q = (hi, lo) with lo = q mod D; element
// (t, q) lives at hi*D*2^k + t*D + lo.
// Element-wise work folded into the first pass's load and the last pass's store of a transform, so the
// Newton -> monomial levels (quotient.hpp) need no separate prepare / multiply / combine kernels:
//   load  NTT_LD_UPPER_HALF   x[node*s + i] = i < s/2 ? src[node*s + s/2 + i] : 0      (s = 2^logs)
//   store NTT_ST_MUL          out = value * aux[index]
//   store NTT_ST_COMBINE      dst[node*s + i] = (i < s/2 ? dst[node*s + i] : 0) + value
//   load  NTT_LD_SCALE_PAD    x[i] = i < cnt ? src[i] * aux[i] : 0
//   load  NTT_LD_PAD          x[i] = i < cnt ? src[i] : 0
//   load  NTT_LD_REV_PAD      x[i] = i < cnt ? src[top - i] : 0
//   store NTT_ST_TAKE         dst[i] = value for i < cnt (nothing else is written)
//   store NTT_ST_REV_TAKE     dst[top - i] = value for i < cnt
enum { NTT_LD_PLAIN = 0, NTT_LD_UPPER_HALF = 1, NTT_LD_SCALE_PAD = 2, NTT_LD_PAD = 3, NTT_LD_REV_PAD = 4 };
enum { NTT_ST_PLAIN = 0, NTT_ST_MUL = 1, NTT_ST_COMBINE = 2, NTT_ST_TAKE = 3, NTT_ST_REV_TAKE = 4 };
struct NttFuse {
    int ld = NTT_LD_PLAIN;
    const Fr* ld_src = nullptr;   // UPPER_HALF / SCALE_PAD: the array read instead of `data`
    const Fr* ld_aux = nullptr;   // SCALE_PAD: the per-element factors
    int st = NTT_ST_PLAIN;
    const Fr* st_aux = nullptr;   // MUL: the per-element factors
    u64 aux_mask = ~0ull;         // MUL: factor index = element index & aux_mask (a batch of transforms sharing one factor array)
    Fr* st_dst = nullptr;         // COMBINE / TAKE: the array written instead of `data`
    int logs = 0;                 // UPPER_HALF / COMBINE: node size
    u64 cnt = 0;                  // *_PAD / *TAKE: element count
    u64 top = 0;                  // REV_PAD / REV_TAKE: index that maps to 0
    // A batch of two transforms in one buffer (the two interpolations of Groth16's route, quotient.hpp): element index =
    // member << batch_log | i.  SCALE_PAD reads member 1 from ld_src2 (cnt and the factors apply per member); TAKE writes
    // member m's first cnt values to st_dst[m << st_member_log | i].  The defaults describe a single transform.
    int batch_log = 63;
    const Fr* ld_src2 = nullptr;
    int st_member_log = 0;
};

// __launch_bounds__(512, PS_NTT_WAVES): hipcc's second argument is waves per SIMD, not blocks per CU.  Four 256-thread workgroups
// per CU (two of 512 with the 80 KB tile) need 4 waves per SIMD, i.e. at most 128 VGPRs; with "2" the inverse pass took 139 and ran one workgroup per CU
// (measured: no difference in the quotient's time either way -- the passes are bound by their instruction count).
#if defined(PS_NTT_TUNE)
__device__ unsigned long long* ntt_trace = nullptr;  // 4 timestamps (s_memtime, 100 MHz) per workgroup of the traced launch
#define PS_NTT_STAMP(i) do { if (ntt_trace && threadIdx.x == 0) ntt_trace[4 * (size_t)blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PS_NTT_STAMP(i) do { } while (0)
#endif
// ---- pieces of a pass, shared by k_ntt_pass and k_ntt_mid ----
struct NttTile {  // geometry of one workgroup's tile: 2^k rows x 2^logCols columns, first column q0
    int p, logD, k, logCols;
    u64 q0;
};
// HBM index of tile element e (the order that keeps consecutive e consecutive in HBM whichever way the pass strides), and
// its slot t * COLS + col in the LDS tile
PS_INL void ntt_tile_map(const NttTile& g, u32 e, u64& addr, u32& slot) {
    const u32 COLS = 1u << g.logCols, rows = 1u << g.k;
    const u64 Dm1 = (1ull << g.logD) - 1;
    u32 t, col;
    if (g.logD >= g.logCols) { t = e >> g.logCols; col = e & (COLS - 1); }
    else { u32 lo = e & (u32)Dm1; t = (e >> g.logD) & (rows - 1); u32 hl = e >> (g.logD + g.k); col = (hl << g.logD) | lo; }
    const u64 q = g.q0 + col;
    addr = ((q >> g.logD) << (g.logD + g.k)) + ((u64)t << g.logD) + (q & Dm1);
    slot = t * COLS + col;
}
PS_INL Fr ntt_load_fused(const Fr* __restrict__ data, u64 addr, const NttFuse& fz) {
    if (fz.ld == NTT_LD_PLAIN) return data[addr];
    if (fz.ld == NTT_LD_UPPER_HALF) {
        const u64 half = 1ull << (fz.logs - 1);
        return (addr & (2 * half - 1)) < half ? fz.ld_src[addr + half] : fr_zero();
    }
    if (fz.ld == NTT_LD_SCALE_PAD) {
        const u64 i = addr & ((1ull << fz.batch_log) - 1);
        const Fr* src = (addr >> fz.batch_log) ? fz.ld_src2 : fz.ld_src;
        return i < fz.cnt ? fr_mul(src[i], fz.ld_aux[i]) : fr_zero();
    }
    if (fz.ld == NTT_LD_PAD) return addr < fz.cnt ? fz.ld_src[addr] : fr_zero();
    return addr < fz.cnt ? fz.ld_src[fz.top - addr] : fr_zero();
}
PS_INL void ntt_store_fused(Fr* __restrict__ data, u64 addr, const Fr& v, const NttFuse& fz) {
    if (fz.st == NTT_ST_PLAIN) {
        data[addr] = v;
    } else if (fz.st == NTT_ST_MUL) {
        data[addr] = fr_mul(v, fz.st_aux[addr & fz.aux_mask]);
    } else if (fz.st == NTT_ST_COMBINE) {
        const u64 half = 1ull << (fz.logs - 1);
        fz.st_dst[addr] = (addr & (2 * half - 1)) < half ? fr_norm(fr_add(fz.st_dst[addr], v)) : v;
    } else if (fz.st == NTT_ST_TAKE) {
        const u64 i = addr & ((1ull << fz.batch_log) - 1);
        if (i < fz.cnt) fz.st_dst[((addr >> fz.batch_log) << fz.st_member_log) + i] = v;
    } else if (addr < fz.cnt) {
        fz.st_dst[fz.top - addr] = v;
    }
}

// The k butterfly stages of a pass on the LDS tile (ends with a barrier).
// Two butterfly stages per LDS round trip where possible: a thread takes the four rows that differ
// in row bits m and m+1, so loads, stores, index arithmetic and barriers are halved.  Forward
// (Cooley-Tukey) walks m = k-1 .. 0, inverse (Gentleman-Sande) m = 0 .. k-1.
#if defined(PS_NTT_X_NONORM)  // timing experiment only (wrong results): what the carry-save steps behind the forward butterflies cost
#define PS_NTT_FNORM(x) (x)
#else
#define PS_NTT_FNORM(x) fr_norm(x)
#endif
// LDS slot of tile element i.  An element is 10 dwords, so butterfly partners 4, 8, 16, 32 elements apart fall into the same
// banks 2-, 4-, 8-, 16-way (SQ_LDS_BANK_CONFLICT: 70 % of the LDS-active cycles of k_ntt_mid); XOR-ing the low five bits of
// the index with the next five spreads every such stride over different banks and keeps runs of consecutive elements intact.
#if defined(PS_NTT_NO_SWIZZLE)
#define PS_NTT_SW(i) (i)
#else
#define PS_NTT_SW(i) ((i) ^ (((i) >> 5) & 31u))
#endif
template <bool INV>
__device__ inline __attribute__((always_inline)) void ntt_tile_stages(Fr* tile, const NttTile& g, const Fr* __restrict__ tw, int log_tab) {
    const int p = g.p, logD = g.logD, k = g.k, logCols = g.logCols;
    const u32 COLS = 1u << logCols, rows = 1u << k;
    const u64 q0 = g.q0;
    const u64 Dm1 = (1ull << logD) - 1;
    const u64 smask = (1ull << p) - 1;
    // global index of element (row t, column col) of this tile, and the twiddle of the butterfly block
    // that holds global index i at a stage with half-distance 2^logh (returns false when w = 1)
    auto gidx = [&](u32 t, u32 col) -> u64 {
        const u64 q = q0 + col;
        return ((q >> logD) << (logD + k)) + ((u64)t << logD) + (q & Dm1);
    };
    auto twiddle = [&](u64 i, int logh, Fr& w) -> bool {
        const int M = p - 1 - logh;  // this stage has 2^M blocks per transform
        if (M <= 0) return false;    // the single block of the outermost stage: w = 1
        const u32 blk = (u32)((i & smask) >> (logh + 1));
#if defined(PS_NTT_X_NEARTW)  // (timing experiment only: every twiddle from the same two cache lines -- wrong results)
        w = tw[blk & 1u];
#else
        w = tw[((size_t)1 << M) + blk];  // the compact table (NttTables::cfwd / cinv)
#endif
        return true;
    };
    const u32 nbf = (rows >> 1) << logCols, nq = (rows >> 2) << logCols;
    int done_st = 0;
    while (done_st < k) {
        const int left = k - done_st;
        // an odd stage count runs its single stage first (forward) / last (inverse)
        const bool single = (left == 1) || (!INV && (left & 1));
        if (single) {
            const int m = INV ? done_st : (k - 1 - done_st);
            const u32 mmask = (1u << m) - 1;
            const int logh = logD + m;
            for (u32 bf = threadIdx.x; bf < nbf; bf += blockDim.x) {
                u32 col = bf & (COLS - 1), r = bf >> logCols;
                u32 t0 = ((r >> m) << (m + 1)) | (r & mmask), t1 = t0 | (1u << m);
                Fr a = tile[PS_NTT_SW(t0 * COLS + col)], b = tile[PS_NTT_SW(t1 * COLS + col)], w;
                const bool has_w = twiddle(gidx(t0, col), logh, w);
                if (!INV) {
                    Fr wb = has_w ? fr_mul(b, w) : b;
                    tile[PS_NTT_SW(t0 * COLS + col)] = fr_norm(fr_add(a, wb));
                    tile[PS_NTT_SW(t1 * COLS + col)] = fr_norm(fr_sub(a, wb));
                } else {
                    tile[PS_NTT_SW(t0 * COLS + col)] = fr_norm(fr_add(a, b));
                    Fr d = fr_norm(fr_sub(a, b));
                    tile[PS_NTT_SW(t1 * COLS + col)] = has_w ? fr_mul(d, w) : d;
                }
            }
            done_st += 1;
        } else {
            const int m_lo = INV ? done_st : (k - 2 - done_st), m_hi = m_lo + 1;
            const u32 lowmask = (1u << m_lo) - 1;
            for (u32 qd = threadIdx.x; qd < nq; qd += blockDim.x) {
                u32 col = qd & (COLS - 1), r = qd >> logCols;
                u32 t00 = ((r >> m_lo) << (m_lo + 2)) | (r & lowmask);
                u32 t01 = t00 | (1u << m_lo), t10 = t00 | (1u << m_hi), t11 = t10 | (1u << m_lo);
                Fr a = tile[PS_NTT_SW(t00 * COLS + col)], b = tile[PS_NTT_SW(t01 * COLS + col)], c = tile[PS_NTT_SW(t10 * COLS + col)], d = tile[PS_NTT_SW(t11 * COLS + col)];
                const u64 i00 = gidx(t00, col), i10 = gidx(t10, col);
                Fr w;
                if (!INV) {
                    // stage m_hi: (a, c), (b, d) share one block; stage m_lo: (a', b') and (c', d')
                    if (twiddle(i00, logD + m_hi, w)) { c = fr_mul(c, w); d = fr_mul(d, w); }
                    Fr a1 = fr_add(a, c), c1 = fr_sub(a, c), b1 = fr_add(b, d), d1 = fr_sub(b, d);  // class 2, fresh sums
                    twiddle(i00, logD + m_lo, w);  // m_lo < k-1: never the outermost stage
                    Fr wb = fr_mul(b1, w);
                    tile[PS_NTT_SW(t00 * COLS + col)] = PS_NTT_FNORM(fr_add(a1, wb));
                    tile[PS_NTT_SW(t01 * COLS + col)] = PS_NTT_FNORM(fr_sub(a1, wb));
                    twiddle(i10, logD + m_lo, w);
                    Fr wd = fr_mul(d1, w);
                    tile[PS_NTT_SW(t10 * COLS + col)] = PS_NTT_FNORM(fr_add(c1, wd));
                    tile[PS_NTT_SW(t11 * COLS + col)] = PS_NTT_FNORM(fr_sub(c1, wd));
                } else {
                    // stage m_lo: (a, b) and (c, d) in neighbouring blocks; stage m_hi: (a', c'), (b', d')
                    twiddle(i00, logD + m_lo, w);  // m_lo < the outermost stage: always a real twiddle
                    Fr a1 = fr_add(a, b), b1 = fr_mul(fr_norm(fr_sub(a, b)), w);
                    twiddle(i10, logD + m_lo, w);
                    Fr c1 = fr_add(c, d), d1 = fr_mul(fr_norm(fr_sub(c, d)), w);
                    const bool has_w = twiddle(i00, logD + m_hi, w);
                    tile[PS_NTT_SW(t00 * COLS + col)] = fr_norm(fr_add(a1, c1));
                    tile[PS_NTT_SW(t01 * COLS + col)] = fr_norm(fr_add(b1, d1));
                    Fr e = fr_norm(fr_sub(a1, c1)), f = fr_norm(fr_sub(b1, d1));
                    tile[PS_NTT_SW(t10 * COLS + col)] = has_w ? fr_mul(e, w) : e;
                    tile[PS_NTT_SW(t11 * COLS + col)] = has_w ? fr_mul(f, w) : f;
                }
            }
            done_st += 2;
        }
        __syncthreads();
    }
}

template <bool INV>
__global__ void __launch_bounds__(512, PS_NTT_WAVES) k_ntt_pass(Fr* __restrict__ data, int p, int logD, int k, int logCols,
                                                  const Fr* __restrict__ tw, int log_tab, NttFuse fz, int scale_log) {
    extern __shared__ __align__(16) unsigned char ntt_smem[];
    Fr* tile = reinterpret_cast<Fr*>(ntt_smem);
    PS_NTT_STAMP(0);
    const NttTile g{p, logD, k, logCols, (u64)blockIdx.x << logCols};
    const u32 tile_elems = 1u << (k + logCols);
    for (u32 e = threadIdx.x; e < tile_elems; e += blockDim.x) {
        u64 addr;
        u32 slot;
        ntt_tile_map(g, e, addr, slot);
        tile[PS_NTT_SW(slot)] = ntt_load_fused(data, addr, fz);
    }
    __syncthreads();
    PS_NTT_STAMP(1);
    ntt_tile_stages<INV>(tile, g, tw, log_tab);
    // The inverse doubles along the all-sums path (a product pulls a value back under 2r, a sum does not);
    // 2^16 r still fits the lazy limbs with room to spare (top limb < 2^20), so ntt_run asks for the
    // scaling 2^-scale_log only in the last pass of a transform, or earlier for very long ones.
    PS_NTT_STAMP(2);
    Fr sc;
    if (INV && scale_log) {
#pragma unroll
        for (int j = 0; j < FR_L; j++) sc.l[j] = c_fr_inv2pow[scale_log][j];
    }
    for (u32 e = threadIdx.x; e < tile_elems; e += blockDim.x) {
        u64 addr;
        u32 slot;
        ntt_tile_map(g, e, addr, slot);
        Fr v = tile[PS_NTT_SW(slot)];
#if !defined(PS_NTT_X_NOSCALE)  // (timing experiment only: the inverse's scaling product left out -- wrong results)
        if (INV && scale_log) v = fr_mul(v, sc);
#endif
        ntt_store_fused(data, addr, v, fz);
    }
    PS_NTT_STAMP(3);
}

// A cyclic convolution with a stored transform is forward transform -> point-wise product -> inverse transform, and the
// forward's LAST pass and the inverse's FIRST pass both run on the same contiguous tiles (stride 1, the lowest k stages).
// k_ntt_mid does both on one visit of the tile: load, k forward stages, times aux, k inverse stages, store -- one HBM round
// trip and one load / store phase instead of two (a pass costs ~12 us + ~18 us of memory phase whatever it computes:
// DESIGN.md section 6; the quotient of the reference's key form is ~150 such passes at 2^20).  first / last: the load
// fusion of the forward's first pass and the store fusion (and scaling) of the inverse's last pass apply here too when the
// transform fits one tile.
__global__ void __launch_bounds__(512, PS_NTT_WAVES) k_ntt_mid(Fr* __restrict__ data, int p, int k, int logCols, const Fr* __restrict__ tw_fwd,
                                                 const Fr* __restrict__ tw_inv, int log_tab, NttFuse fl, NttFuse fs, const Fr* __restrict__ aux,
                                                 u64 aux_mask, int scale_log) {
    extern __shared__ __align__(16) unsigned char ntt_smem[];
    Fr* tile = reinterpret_cast<Fr*>(ntt_smem);
    const NttTile g{p, 0, k, logCols, (u64)blockIdx.x << logCols};
    const u32 tile_elems = 1u << (k + logCols);
    for (u32 e = threadIdx.x; e < tile_elems; e += blockDim.x) {
        u64 addr;
        u32 slot;
        ntt_tile_map(g, e, addr, slot);
        tile[PS_NTT_SW(slot)] = ntt_load_fused(data, addr, fl);
    }
    __syncthreads();
    ntt_tile_stages<false>(tile, g, tw_fwd, log_tab);
    for (u32 e = threadIdx.x; e < tile_elems; e += blockDim.x) {
        u64 addr;
        u32 slot;
        ntt_tile_map(g, e, addr, slot);
        tile[PS_NTT_SW(slot)] = fr_mul(tile[PS_NTT_SW(slot)], aux[addr & aux_mask]);
    }
    __syncthreads();
    ntt_tile_stages<true>(tile, g, tw_inv, log_tab);
    Fr sc;
    if (scale_log) {
#pragma unroll
        for (int j = 0; j < FR_L; j++) sc.l[j] = c_fr_inv2pow[scale_log][j];
    }
    for (u32 e = threadIdx.x; e < tile_elems; e += blockDim.x) {
        u64 addr;
        u32 slot;
        ntt_tile_map(g, e, addr, slot);
        Fr v = tile[PS_NTT_SW(slot)];
        if (scale_log) v = fr_mul(v, sc);
        ntt_store_fused(data, addr, v, fs);
    }
}

#if defined(PS_NTT_PASS8)
// ---- MEASURED AND NOT SHIPPED (round 3): the same pass with the data in REGISTERS, -DPS_NTT_PASS8 ----
// Eight elements per thread, three butterfly stages per LDS round trip: 320 instructions per butterfly instead of 450 --
// and 6 % SLOWER on the quotient (A/B on one box, tools/ntt_ab.sh: 11.6-12.0 ms against 10.9-11.1).  A pass is not bound by
// its instruction count alone: with one round of workgroups the chip loads, computes and stores in lock-step, so a
// 2^20-element pass costs ~12 us + ~18 us of memory phase + 3.5 us per stage, added up (without its butterflies the pass
// still takes 32-46 us, tools/ntt_tune.sh with -DPS_NTT8_SKIP_BF), and what hides one phase under another is waves per
// SIMD: this kernel holds 172 / 218 VGPRs (two waves per SIMD -- which its 40 KB tile per 128 threads would enforce anyway: 4 096
// resident elements per CU are 512 threads of eight), k_ntt_pass 128 (four).  Prefetching the next tile into
// registers from a persistent loop made it worse (in-order vmcnt: the first twiddle load waits for the prefetch).  Kept
// behind the switch as the record of the experiment; DESIGN.md section 6.
// k_ntt_pass above spends 450 instructions per butterfly for the 250 of its field product: a round trip through LDS and
// the index arithmetic of a four-row group every two stages, a carry-save step behind every butterfly, one twiddle load
// per butterfly pair.  Here a thread holds the eight elements that differ in three consecutive stage bits (a "window"
// of the tile's element index), runs the three stages on them in registers -- 12 butterflies, 7 twiddles, one carry-save
// step per element and window (the forward transform only adds products: limb class 1 + 3 <= 7; the inverse, whose sums
// double, normalises once more inside the window) -- and meets the other threads in LDS only between windows.  The first
// window is loaded straight from HBM and the last one stored straight to it, so a pass of k stages makes ceil(k/3) - 1
// LDS round trips instead of k/2, and a 2^21-point transform is 9 + 6 + 6 stages with four of them in all.
// Tile element index: bits [logL, logL + k) are the row (stage) bits, logL = min(logD, logCols), so that consecutive
// elements are consecutive in HBM whichever way the pass strides.
PS_INL Fr ntt_ld(const Fr* __restrict__ data, u64 addr, const NttFuse& fz) {
    if (fz.ld == NTT_LD_PLAIN) return data[addr];
    if (fz.ld == NTT_LD_UPPER_HALF) {
        const u64 half = 1ull << (fz.logs - 1);
        return (addr & (2 * half - 1)) < half ? fz.ld_src[addr + half] : fr_zero();
    }
    if (fz.ld == NTT_LD_SCALE_PAD) return addr < fz.cnt ? fr_mul(fz.ld_src[addr], fz.ld_aux[addr]) : fr_zero();
    if (fz.ld == NTT_LD_PAD) return addr < fz.cnt ? fz.ld_src[addr] : fr_zero();
    return addr < fz.cnt ? fz.ld_src[fz.top - addr] : fr_zero();
}
PS_INL void ntt_st(Fr* __restrict__ data, u64 addr, const Fr& v, const NttFuse& fz) {
    if (fz.st == NTT_ST_PLAIN) {
        data[addr] = v;
    } else if (fz.st == NTT_ST_MUL) {
        data[addr] = fr_mul(v, fz.st_aux[addr & fz.aux_mask]);
    } else if (fz.st == NTT_ST_COMBINE) {
        const u64 half = 1ull << (fz.logs - 1);
        fz.st_dst[addr] = (addr & (2 * half - 1)) < half ? fr_norm(fr_add(fz.st_dst[addr], v)) : v;
    } else if (addr < fz.cnt) {
        fz.st_dst[fz.st == NTT_ST_TAKE ? addr : fz.top - addr] = v;
    }
}

// one butterfly, Cooley-Tukey (a + w b, a - w b) forward, Gentleman-Sande (a + b, (a - b) w) inverse
template <bool INV>
__device__ inline __attribute__((always_inline)) void ntt8_bf(Fr& a, Fr& b, const Fr& w, bool has_w) {
    if (!INV) {
        const Fr wb = has_w ? fr_mul(b, w) : b;
        b = fr_sub(a, wb);
        a = fr_add(a, wb);
    } else {
        const Fr d = fr_sub(a, b);
        a = fr_add(a, b);
        b = has_w ? fr_mul(d, w) : d;
    }
}

template <bool INV>
__global__ void __launch_bounds__(256, 2) k_ntt_pass8(Fr* __restrict__ data, int p, int logD, int k, int logCols,
                                                      const Fr* __restrict__ tw, int log_tab, NttFuse fz, int scale_log) {
    extern __shared__ __align__(16) unsigned char ntt_smem[];
    Fr* tile = reinterpret_cast<Fr*>(ntt_smem);
    const u32 tid = threadIdx.x;
    const int logL = logD < logCols ? logD : logCols;
    const u64 q0 = (u64)blockIdx.x << logCols;
    const u64 Dm1 = (1ull << logD) - 1, smask = (1ull << p) - 1;
    const u32 rmask = (1u << k) - 1u, lmask = (1u << logL) - 1u;
    auto gaddr_at = [&](u32 e, u64 qbase) -> u64 {  // HBM index of element e of the tile whose first column is qbase
        const u32 t = (e >> logL) & rmask, col = ((e >> (logL + k)) << logL) | (e & lmask);
        const u64 q = qbase + col;
        return ((q >> logD) << (logD + k)) + ((u64)t << logD) + (q & Dm1);
    };
    auto gaddr = [&](u32 e) -> u64 { return gaddr_at(e, q0); };
    // twiddle of the butterfly block that holds HBM index `a` at the stage of row bit m (false: the single block of a
    // transform's outermost stage, w = 1)
    auto twiddle = [&](u64 a, int m, Fr& w) -> bool {
        const int logh = logD + m, M = p - 1 - logh;
        if (M <= 0) return false;
        const u32 blk = (u32)((a & smask) >> (logh + 1));
        w = tw[(size_t)(__brev(blk) >> (32 - M)) << (log_tab - 1 - M)];
        return true;
    };
    PS_NTT_STAMP(0);
    const int ngroups = k >= 3 ? (k + 2) / 3 : 1;
    // eight named values, not an array: every index below is a literal, so they stay in registers (an Fr x[8] walked by
    // loops inside the window loop went to scratch: 336 bytes, a hundred scratch stores per pass)
    Fr x0, x1, x2, x3, x4, x5, x6, x7;
#define PS_NTT8_EACH(OP) OP(x0, 0) OP(x1, 1) OP(x2, 2) OP(x3, 3) OP(x4, 4) OP(x5, 5) OP(x6, 6) OP(x7, 7)
    for (int gi = 0; gi < ngroups; gi++) {
        // window [mwin, mwin + 3) of row bits and the stages [m_lo, m_hi] it runs (forward: downwards)
        int mwin, m_lo, m_hi;
        if (k < 3) { mwin = 0; m_lo = 0; m_hi = k - 1; }
        else if (!INV) { m_hi = k - 1 - 3 * gi; m_lo = m_hi - 2 < 0 ? 0 : m_hi - 2; mwin = m_lo; }
        else { m_lo = 3 * gi; m_hi = m_lo + 2 > k - 1 ? k - 1 : m_lo + 2; mwin = m_lo + 3 <= k ? m_lo : k - 3; }
        const int pos = logL + mwin;
        const u32 base = ((tid >> pos) << (pos + 3)) | (tid & ((1u << pos) - 1u));
        if (gi == 0) {
#define PS_NTT8_LDG(X, J) X = ntt_ld(data, gaddr(base | ((u32)(J) << pos)), fz);
            PS_NTT8_EACH(PS_NTT8_LDG)
#undef PS_NTT8_LDG
        } else {
#define PS_NTT8_LDS(X, J) X = tile[base | ((u32)(J) << pos)];
            PS_NTT8_EACH(PS_NTT8_LDS)
#undef PS_NTT8_LDS
        }
        // the (up to) three stages of the window; the stage of window bit s pairs the elements whose numbers differ in bit s,
        // and the elements of a pair share their block's twiddle with every pair that agrees with them above bit s
        Fr w;
        bool hw;
        auto tw_of = [&](int j0, int m) -> bool { return twiddle(gaddr(base | ((u32)j0 << pos)), m, w); };
        const bool do0 = mwin >= m_lo && mwin <= m_hi, do1 = mwin + 1 >= m_lo && mwin + 1 <= m_hi, do2 = mwin + 2 >= m_lo && mwin + 2 <= m_hi;
#define PS_NTT8_S2 { hw = tw_of(0, mwin + 2); ntt8_bf<INV>(x0, x4, w, hw); ntt8_bf<INV>(x1, x5, w, hw); ntt8_bf<INV>(x2, x6, w, hw); ntt8_bf<INV>(x3, x7, w, hw); }
#define PS_NTT8_S1 { hw = tw_of(0, mwin + 1); ntt8_bf<INV>(x0, x2, w, hw); ntt8_bf<INV>(x1, x3, w, hw); \
                     hw = tw_of(4, mwin + 1); ntt8_bf<INV>(x4, x6, w, hw); ntt8_bf<INV>(x5, x7, w, hw); }
#define PS_NTT8_S0 { hw = tw_of(0, mwin); ntt8_bf<INV>(x0, x1, w, hw); hw = tw_of(2, mwin); ntt8_bf<INV>(x2, x3, w, hw); \
                     hw = tw_of(4, mwin); ntt8_bf<INV>(x4, x5, w, hw); hw = tw_of(6, mwin); ntt8_bf<INV>(x6, x7, w, hw); }
#if defined(PS_NTT8_SKIP_BF)  // measurement build: the pass without its butterflies (what the memory phases alone cost)
        (void)do0; (void)do1; (void)do2; (void)hw; (void)tw_of;
        if (false)
#endif
        if (!INV) {
            if (do2) PS_NTT8_S2
            if (do1) PS_NTT8_S1
            if (do0) PS_NTT8_S0
        } else {
            if (do0) PS_NTT8_S0
            if (do1) PS_NTT8_S1
            if (do2) {
                // two stages of sums may lie behind us: limb class 4, the next sum would not fit an i32 limb
#define PS_NTT8_NORM(X, J) X = fr_norm(X);
                PS_NTT8_EACH(PS_NTT8_NORM)
                PS_NTT8_S2
            }
        }
#undef PS_NTT8_S0
#undef PS_NTT8_S1
#undef PS_NTT8_S2
        if (gi + 1 < ngroups) {
            // a thread writes the slots it read itself (windows partition the tile): one barrier, before the NEXT window's reads
#define PS_NTT8_STS(X, J) tile[base | ((u32)(J) << pos)] = fr_norm(X);
            PS_NTT8_EACH(PS_NTT8_STS)
#undef PS_NTT8_STS
            __syncthreads();
            if (gi == 0) PS_NTT_STAMP(1);
        } else {
            PS_NTT_STAMP(2);
            Fr sc;
            if (INV && scale_log) {
#pragma unroll
                for (int j = 0; j < FR_L; j++) sc.l[j] = c_fr_inv2pow[scale_log][j];
            }
#define PS_NTT8_STG(X, J) { Fr v = fr_norm(X); if (INV && scale_log) v = fr_mul(v, sc); ntt_st(data, gaddr(base | ((u32)(J) << pos)), v, fz); }
            PS_NTT8_EACH(PS_NTT8_STG)
#undef PS_NTT8_STG
        }
    }
    PS_NTT_STAMP(3);
#undef PS_NTT8_NORM
#undef PS_NTT8_EACH
}

#endif  // PS_NTT_PASS8

#if defined(PS_NTT_TUNE)
static unsigned long long* g_ntt_trace_buf = nullptr;
static int g_ntt_launch_no = 0;
static int g_ntt_trace_meta[6] = {0, 0, 0, 0, 0, 0};  // grid, k, logD, inverse, p, threads of the traced launch
#endif
struct NttTables {
    Fr* fwd = nullptr;   // w^i, i < 2^(log_tab-1)
    Fr* inv = nullptr;
    // The same values in the order the passes read them: level M (2^M blocks per transform at a stage) at [2^M, 2^(M+1)), block
    // order inside the level -- cfwd[2^M + blk] = w_{2^(M+1)}^bitrev_M(blk).  With the power table every block's twiddle sits
    // in a cache line of its own at a bit-reversed place; here the 64 twiddles a wave needs at a low stage are 2.5 KB in a row.
    Fr* cfwd = nullptr;
    Fr* cinv = nullptr;
    int log_tab = 0;
};
// c[idx] for idx in [2, 2^log_tab): level M = floor(log2 idx), block idx - 2^M   (c[0], c[1] = 1: never read)
__global__ void __launch_bounds__(256) k_twiddle_compact(Fr* __restrict__ c, const Fr* __restrict__ tw, int log_tab) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (1u << log_tab)) return;
    if (idx < 2) { c[idx] = fr_one(); return; }
    const int M = 31 - __clz(idx);
    const u32 blk = idx - (1u << M);
    c[idx] = tw[(size_t)(__brev(blk) >> (32 - M)) << (log_tab - 1 - M)];
}

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
    if (t.fwd) {
        (void)hipStreamSynchronize(st);
        for (Fr** p : {&t.fwd, &t.inv, &t.cfwd, &t.cinv}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        t.log_tab = 0;
    }
    size_t entries = (size_t)1 << (log_size - 1);
    if ((e = hipMalloc((void**)&t.fwd, entries * sizeof(Fr))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&t.inv, entries * sizeof(Fr))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&t.cfwd, 2 * entries * sizeof(Fr))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&t.cinv, 2 * entries * sizeof(Fr))) != hipSuccess) return e;
    unsigned blocks = (unsigned)((entries + 255) / 256);
    hipLaunchKernelGGL(k_twiddle_table, dim3(blocks), dim3(256), 0, st, t.fwd, log_size, 0);
    hipLaunchKernelGGL(k_twiddle_table, dim3(blocks), dim3(256), 0, st, t.inv, log_size, 1);
    hipLaunchKernelGGL(k_twiddle_compact, dim3(2 * blocks), dim3(256), 0, st, t.cfwd, (const Fr*)t.fwd, log_size);
    hipLaunchKernelGGL(k_twiddle_compact, dim3(2 * blocks), dim3(256), 0, st, t.cinv, (const Fr*)t.inv, log_size);
    t.log_tab = log_size;
    return hipGetLastError();
}

#if defined(PS_NTT_TUNE)
// tiles beyond the default 64 KB of dynamic LDS (PS_NTT_TILE=11: 80 KB)
static inline hipError_t ntt_tune_raise_lds() {
    static bool raised = false;
    if (raised) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_pass<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_pass<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_mid), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    raised = e == hipSuccess;
    return e;
}
#endif

// `total` = batch * 2^p elements, contiguous.  Forward: natural -> bit-reversed (per block).
template <bool INV>
static inline hipError_t ntt_run(const NttTables& tb, hipStream_t st, Fr* data, u64 total, int p, const NttFuse& fuse = NttFuse()) {
    if (p == 0) return (fuse.ld || fuse.st) ? hipErrorInvalidValue : hipSuccess;
    if (p > tb.log_tab) return hipErrorInvalidValue;
    // a transform that fits one tile runs all its stages in one pass (contiguous in HBM: no stride to respect);
    // longer ones are cut into passes of at most NTT_MAX_K stages (with the 80 KB tile of round 1, measured at 2^20:
    // 8 stages 12.25 ms, 9: 11.9, 10: 11.9; see the note at NTT_TILE_LOG for the tile size)
    int max_k = NTT_MAX_K, tile_log = NTT_TILE_LOG;
#if defined(PS_NTT_TUNE)  // measurement builds: pass shapes from the environment (tools/ntt_tune.sh)
    if (const char* e = getenv("PS_NTT_MAXK")) max_k = atoi(e);
    if (const char* e = getenv("PS_NTT_TILE")) tile_log = atoi(e);
    if (ntt_tune_raise_lds() != hipSuccess) return hipErrorInvalidValue;
#endif
    int npass = p <= tile_log ? 1 : (p + max_k - 1) / max_k;
    // stage groups of nearly equal size; the forward walks them from the top, the inverse from the bottom
    int done = 0, unscaled = 0;  // unscaled: inverse stages whose factor 2 per stage has not been divided out yet
    // stages per pass: nearly equal, except that k_ntt_pass8 works in windows of three stages -- passes of six stages behind
    // a first one of up to nine need the fewest windows (2^21 points: 9 + 6 + 6 = seven windows, 7 + 7 + 7 = nine)
    int ks[8];
    {
        int left = p;
        for (int i = 0; i < npass; i++) { ks[i] = (left + (npass - i) - 1) / (npass - i); left -= ks[i]; }
#if defined(PS_NTT_TUNE)
        if (getenv("PS_NTT_SPLIT")) {  // the odd stage goes to the pass on contiguous data (the forward's last, the inverse's first)
            left = p;
            for (int i = 0; i < npass; i++) { const int j = INV ? npass - 1 - i : i; ks[j] = left / (npass - i); left -= ks[j]; }
        }
#endif
#if defined(PS_NTT_PASS8)
        const int k0 = p - 6 * (npass - 1);
        // the long pass is the one that runs on contiguous data (logD = 0: the inverse's first, the forward's last); the strided
        // passes then keep 16 columns = 640-byte runs
        if (npass >= 2 && k0 >= 4 && k0 <= max_k && max_k >= 6) {
            for (int i = 0; i < npass; i++) ks[i] = 6;
            ks[INV ? 0 : npass - 1] = k0;
        }
#endif
    }
    for (int ps_i = 0; ps_i < npass; ps_i++) {
        int k = ks[ps_i];
        int logD = INV ? done : (p - done - k);
        int logCols = tile_log - k;
        { const int free_cols = __builtin_ctzll(total >> k); if (logCols > free_cols) logCols = free_cols; }  // (total: any multiple of 2^p -- a batch of three)
        if (logCols < 0) logCols = 0;
        u64 cols_total = total >> k;
        unsigned grid = (unsigned)(cols_total >> logCols);
        size_t smem = ((size_t)sizeof(Fr) << k) << logCols;
        // one four-row group per thread and double stage; four 256-thread workgroups share a CU (4 x 40 KB
        // of LDS), so 4 waves per SIMD hide the LDS and multiplier latency
        unsigned threads = (unsigned)std::min<u64>(512, std::max<u64>(64, ((u64)1 << (k + logCols)) >> 2));
        NttFuse fz;  // the load belongs to the first pass, the store to the last
        if (ps_i == 0) { fz.ld = fuse.ld; fz.ld_src = fuse.ld_src; fz.ld_aux = fuse.ld_aux; }
        if (ps_i == npass - 1) { fz.st = fuse.st; fz.st_aux = fuse.st_aux; fz.st_dst = fuse.st_dst; fz.aux_mask = fuse.aux_mask; }
        fz.logs = fuse.logs;
        fz.cnt = fuse.cnt;
        fz.top = fuse.top;
        // inverse: divide the doubling out in the last pass, and earlier only where the next pass would push
        // the growth past 2^16 (the lazy top limb holds 2^16 r with room to spare)
        int scale_log = 0;
        if (INV) {
            unscaled += k;
            const int next_k = ps_i + 1 < npass ? ks[ps_i + 1] : 0;
            if (ps_i == npass - 1 || unscaled + next_k > 16) { scale_log = unscaled; unscaled = 0; }
        }
#if defined(PS_NTT_PASS8)
        if (smem > 64 * 1024) {  // tiles beyond the default dynamic-LDS limit (gfx950 has 160 KB per CU)
            static bool raised = false;
            if (!raised) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_pass8<INV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
                raised = true;
            }
        }
        if (k + logCols >= 6) {  // eight elements per thread (tiles of fewer than 64 elements keep the small kernel)
#if defined(PS_NTT_TUNE)
            static const int traced8 = getenv("PS_NTT_TRACE_LAUNCH") ? atoi(getenv("PS_NTT_TRACE_LAUNCH")) : -1;
            const bool trace8 = g_ntt_launch_no++ == traced8 && grid <= 8192;
            if (trace8) {
                if (!g_ntt_trace_buf && hipMalloc((void**)&g_ntt_trace_buf, 4 * 8192 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
                (void)hipStreamSynchronize(st);
                (void)hipMemset(g_ntt_trace_buf, 0, 4 * 8192 * sizeof(unsigned long long));
                (void)hipMemcpyToSymbol(HIP_SYMBOL(ntt_trace), &g_ntt_trace_buf, sizeof(void*));
                g_ntt_trace_meta[0] = (int)grid; g_ntt_trace_meta[1] = k; g_ntt_trace_meta[2] = logD; g_ntt_trace_meta[3] = INV ? 1 : 0;
                g_ntt_trace_meta[4] = p; g_ntt_trace_meta[5] = 1 << (k + logCols - 3);
            }
#endif
            hipLaunchKernelGGL(k_ntt_pass8<INV>, dim3(grid), dim3(1u << (k + logCols - 3)), smem, st, data, p, logD, k, logCols,
                               INV ? tb.inv : tb.fwd, tb.log_tab, fz, scale_log);
#if defined(PS_NTT_TUNE)
            if (trace8) {
                (void)hipStreamSynchronize(st);
                unsigned long long* none = nullptr;
                (void)hipMemcpyToSymbol(HIP_SYMBOL(ntt_trace), &none, sizeof(void*));
            }
#endif
            done += k;
            continue;
        }
#endif
#if defined(PS_NTT_TUNE)
        // PS_NTT_TRACE_LAUNCH=i: the i-th pass launched by this process leaves its workgroups' phase stamps in a buffer that
        // ps_debug_ntt_trace copies out (tools/ntt_phases.py)
        static const int traced = getenv("PS_NTT_TRACE_LAUNCH") ? atoi(getenv("PS_NTT_TRACE_LAUNCH")) : -1;
        const bool trace_this = g_ntt_launch_no++ == traced && grid <= 8192;
        if (trace_this) {
            if (!g_ntt_trace_buf && hipMalloc((void**)&g_ntt_trace_buf, 4 * 8192 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
            (void)hipStreamSynchronize(st);
            (void)hipMemset(g_ntt_trace_buf, 0, 4 * 8192 * sizeof(unsigned long long));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(ntt_trace), &g_ntt_trace_buf, sizeof(void*));
            g_ntt_trace_meta[0] = (int)grid; g_ntt_trace_meta[1] = k; g_ntt_trace_meta[2] = logD; g_ntt_trace_meta[3] = INV ? 1 : 0;
            g_ntt_trace_meta[4] = p; g_ntt_trace_meta[5] = (int)threads;
        }
#endif
        hipLaunchKernelGGL(k_ntt_pass<INV>, dim3(grid), dim3(threads), smem, st, data, p, logD, k, logCols,
                           INV ? tb.cinv : tb.cfwd, tb.log_tab, fz, scale_log);
#if defined(PS_NTT_TUNE)
        if (trace_this) {
            (void)hipStreamSynchronize(st);
            unsigned long long* none = nullptr;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(ntt_trace), &none, sizeof(void*));
        }
#endif
        done += k;
    }
    return hipGetLastError();
}

// Cyclic convolution of every 2^p-block of `data` with the stored transform `aux` (bit-reversed order, as ntt_run<false>
// leaves it; index = element index & aux_mask):  data <- INTT(NTT(load(data)) * aux), with the load fusion of `ld_fuse` on the
// way in and the store fusion of `st_fuse` on the way out.  Same arithmetic, butterfly for butterfly, as
//     ntt_run<false>(.., {fuse.ld.., st = NTT_ST_MUL, st_aux = aux});  ntt_run<true>(.., {fuse.st..});
// but the forward's last pass and the inverse's first pass are ONE kernel (k_ntt_mid): 2 npass - 1 passes instead of 2 npass.
static inline hipError_t ntt_conv(const NttTables& tb, hipStream_t st, Fr* data, u64 total, int p, const NttFuse& ld_fuse, const NttFuse& st_fuse,
                                  const Fr* aux, u64 aux_mask = ~0ull) {
    if (p == 0 || p > tb.log_tab || st_fuse.st == NTT_ST_MUL || ld_fuse.st != NTT_ST_PLAIN || st_fuse.ld != NTT_LD_PLAIN) return hipErrorInvalidValue;
    int max_k = NTT_MAX_K, tile_log = NTT_TILE_LOG;
#if defined(PS_NTT_TUNE)  // measurement builds: pass shapes from the environment, as in ntt_run
    if (const char* e = getenv("PS_NTT_MAXK")) max_k = atoi(e);
    if (const char* e = getenv("PS_NTT_TILE")) tile_log = atoi(e);
    if (ntt_tune_raise_lds() != hipSuccess) return hipErrorInvalidValue;
#endif
    const int npass = p <= tile_log ? 1 : (p + max_k - 1) / max_k;
    int ks[8];  // forward order (from the top); the inverse walks the same groups back
    {
        int left = p;
        for (int i = 0; i < npass; i++) { ks[i] = (left + (npass - i) - 1) / (npass - i); left -= ks[i]; }
#if defined(PS_NTT_TUNE)
        if (getenv("PS_NTT_SPLIT")) {
            left = p;
            for (int i = 0; i < npass; i++) { ks[i] = left / (npass - i); left -= ks[i]; }
        }
#endif
    }
    auto shape = [&](int k, int& logCols, unsigned& grid, size_t& smem, unsigned& threads) {
        logCols = tile_log - k;
        { const int free_cols = __builtin_ctzll(total >> k); if (logCols > free_cols) logCols = free_cols; }  // (total: any multiple of 2^p -- a batch of three)
        if (logCols < 0) logCols = 0;
        grid = (unsigned)((total >> k) >> logCols);
        smem = ((size_t)sizeof(Fr) << k) << logCols;
        threads = (unsigned)std::min<u64>(512, std::max<u64>(64, ((u64)1 << (k + logCols)) >> 2));
    };
    int done = 0;
    for (int i = 0; i + 1 < npass; i++) {  // forward passes above the tile-resident stages
        const int k = ks[i], logD = p - done - k;
        int logCols; unsigned grid, threads; size_t smem;
        shape(k, logCols, grid, smem, threads);
        const NttFuse fz = i == 0 ? ld_fuse : NttFuse();
        hipLaunchKernelGGL(k_ntt_pass<false>, dim3(grid), dim3(threads), smem, st, data, p, logD, k, logCols, tb.cfwd, tb.log_tab, fz, 0);
        done += k;
    }
    const int km = ks[npass - 1];
    int unscaled = km;
    {
        int logCols; unsigned grid, threads; size_t smem;
        shape(km, logCols, grid, smem, threads);
        const NttFuse fl = npass == 1 ? ld_fuse : NttFuse(), fs = npass == 1 ? st_fuse : NttFuse();
        int scale_log = 0;
        const int next_k = npass > 1 ? ks[npass - 2] : 0;
        if (npass == 1 || unscaled + next_k > 16) { scale_log = unscaled; unscaled = 0; }
        hipLaunchKernelGGL(k_ntt_mid, dim3(grid), dim3(threads), smem, st, data, p, km, logCols, (const Fr*)tb.cfwd, (const Fr*)tb.cinv, tb.log_tab, fl, fs,
                           aux, aux_mask, scale_log);
    }
    done = km;
    for (int i = npass - 2; i >= 0; i--) {  // inverse passes, upwards
        const int k = ks[i], logD = done;
        int logCols; unsigned grid, threads; size_t smem;
        shape(k, logCols, grid, smem, threads);
        const NttFuse fz = i == 0 ? st_fuse : NttFuse();
        unscaled += k;
        int scale_log = 0;
        const int next_k = i > 0 ? ks[i - 1] : 0;
        if (i == 0 || unscaled + next_k > 16) { scale_log = unscaled; unscaled = 0; }
        hipLaunchKernelGGL(k_ntt_pass<true>, dim3(grid), dim3(threads), smem, st, data, p, logD, k, logCols, tb.cinv, tb.log_tab, fz, scale_log);
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
// plain little-endian words (8 x u32, canonical) <-> Montgomery Fr
__global__ void __launch_bounds__(256) k_fr_to_mont(Fr* __restrict__ dst, const u32* __restrict__ src, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = src[8 * i + j];
    dst[i] = fr_to_mont(fr_from_words8(w));
}
__global__ void __launch_bounds__(256) k_fr_from_mont(u32* __restrict__ dst, const Fr* __restrict__ src, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
    fr_to_words8(w, fr_from_mont(src[i]));
#pragma unroll
    for (int j = 0; j < 8; j++) dst[8 * i + j] = w[j];
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
