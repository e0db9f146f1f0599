// Pippenger bucket MSM for BLS12-381 G1 / G2 on gfx950 -- the replacement for the reference's
// serial "blind evaluation" loop  acc = acc.Add(acc, tmp.Mul(p[i], blindedPoint[i]))
// (algebra.go:355-357; also groth16.go:176-178, pinochio.go:225-227).
//
// Pipeline (all on one HIP stream, no host round trip until the W window sums come back):
//   1 k_digits_grouped  signed c-bit digits, one window x 16K scalars per workgroup, entries grouped by
//                   counter cache line in LDS; a returning global atomic on the bucket counter gives each
//                   entry its rank inside its bucket
//   2 k_scan_*      exclusive prefix sum of the W*2^(c-1) bucket counters (wave shuffles + LDS)
//   3 k_scatter     counting-sort scatter: sorted[offs[bucket] + rank] = point index | sign
//   4 k_accumulate  the hot kernel.  Each thread owns a fixed slice of M consecutive sorted
//                   entries -- perfectly balanced whatever the bucket-size distribution (skewed
//                   witness vectors, repeated scalars) -- and walks it with an XYZZ accumulator,
//                   flushing whenever the bucket id changes.  Buckets that lie inside one slice
//                   are final; runs cut by a slice boundary go to a head/tail partial slot.
//   5 k_fixup       per bucket: add up the partial slots of the slices it spans
//   6 k_reduce_l1/pyr/sum/fin  per bucket set  sum_b (b+1)*B[b]: 8-bucket running sums, then sums by the bits of the
//     segment index; their weights 2^k are applied on the host
//   host            the weights of the reduction's partial results, Horner over the W window sums (c doublings each)
//                   and affine normalisation
//
// Data layout in HBM: points AoS affine Montgomery, 14 x 28-bit limbs per coordinate (112 B G1 /
// 224 B G2, 16-B aligned, read with dwordx4 loads); scalars 8 x u32 little-endian plain; keys/ranks/sorted as [W][n] u32 so the
// digit kernel's stores are coalesced across scalars; buckets and partials AoS XYZZ.
#pragma once
#include "curve.hpp"

namespace ps {

// Wave priority of the sort and tail kernels (s_setprio; 0 = leave it alone).  With sums in flight their waves share SIMDs with two
// accumulation waves each, and the oldest sum's tail is what the host waits for.
#ifndef PS_TAIL_PRIO
#define PS_TAIL_PRIO 0
#endif
#if PS_TAIL_PRIO > 0
#define PS_TAIL_PRIO_HERE __builtin_amdgcn_s_setprio(PS_TAIL_PRIO)
#else
#define PS_TAIL_PRIO_HERE do { } while (0)
#endif

struct MsmPlan {
    int c;       // window bits
    int W;       // windows (digits per scalar)
    u32 NB;      // buckets per bucket set = 2^(c-1)
    u64 G;       // buckets in all: sets * NB
    int M;       // sorted entries per accumulation thread
    int SEG;     // buckets per reduction thread
    int sets;    // bucket sets: W, or 1 when the points come with their window table 2^(c w) P (all windows share one set)
    bool table;  // the point pass reads the window table (a plain plan with a single window also has sets == 1)
    // trees of lane-cooperative additions (qtail.hpp) instead of chains, decided separately for the two halves of the tail:
    bool qtail;   // REDUCTION by rows / columns / bits of the buckets themselves (k_qreduce_*): bucket sets of <= 2^16 buckets in all
    bool hybrid;  // ... else the 8-bucket running sums (k_reduce_l1) and quads behind them; false: the chains of round 2 throughout
    bool shortsum;  // fewer than PS_QTAIL_MAX_ENTRIES digits: short slices, cut buckets summed by quads (k_qfixup), one stream when alone
    int lpb;      // quads per bucket in k_qfixup (0: the one-thread-per-bucket k_fixup)
    int rc_s;     // column bits of the row / column split of a bucket set (0: bit sums straight from the buckets)
    bool busy;    // planned while other sums were pending on the context: lane-time counts, not depth (capi.hip, msm_plan_tail)
};
// Entries of the sorted list: point index | window << ENTRY_W_SHIFT | sign << 31.  The window field is used only
// with a window table (then the index must fit ENTRY_W_SHIFT bits); without one the index may use all 31 bits.
constexpr int ENTRY_W_SHIFT = 26;

// cost model in field multiplications: N*W mixed adds (10) + G * (fix-up + 2 reduction adds) (42)
static inline MsmPlan msm_plan(size_t n, int max_bits, int forced_c) {
    MsmPlan best{};
    double best_cost = 1e300;
    // c >= 4 keeps W <= 64.  Above c = 16 the bucket array (W * 2^(c-1) XYZZ points of 224 / 448 B) no
    // longer fits the 256 MB infinity cache and the sort scatters over too wide a range: measured at
    // 2^22 and 2^24 points, c = 16 beats 17..20 although the model below prefers them.
    for (int c = 4; c <= 20; c++) {
        if (forced_c ? c != forced_c : c > 16) continue;
        int W = max_bits / c + 1;
        double nb = (double)(1u << (c - 1));
        double cost = (double)n * W * 10.0 + W * nb * 42.0;
        if (cost < best_cost) {
            best_cost = cost;
            best.c = c;
            best.W = W;
        }
    }
    best.NB = 1u << (best.c - 1);
    best.sets = best.W;
    best.G = (u64)best.W * best.NB;
    // entries per accumulation thread: 32 keeps ~2^19 threads busy at 2^20 points; larger inputs get
    // longer slices (same thread count), so a bucket of n / 2^(c-1) entries still spans only a few
    // slices and the fix-up chains stay short
    best.M = 32;
    const u64 total = (u64)n * best.W;
    while (best.M < 1024 && total / (2 * (u64)best.M) >= (1u << 19)) best.M *= 2;
    while (best.M < 1024 && (u64)best.M * 5 * best.G < total) best.M *= 2;  // ... and an average bucket spans <= 5 slices (below)
    while (best.M > 8 && total / (u64)best.M < (1u << 17) && (u64)(best.M / 2) * 5 * best.G >= total) best.M /= 2;  // short sums (below)
    best.SEG = 8;
    return best;
}
// Plan over a window table built for c-bit windows: one bucket set whatever the number of windows.
static inline MsmPlan msm_plan_table(size_t n, int max_bits, int c) {
    MsmPlan pl{};
    pl.c = c;
    pl.W = max_bits / c + 1;
    pl.NB = 1u << (c - 1);
    pl.sets = 1;
    pl.table = true;
    pl.G = pl.NB;
    pl.M = 32;
    const u64 total = (u64)n * pl.W;
    while (pl.M < 1024 && total / (2 * (u64)pl.M) >= (1u << 19)) pl.M *= 2;
    // An average bucket must not span more than ~5 slices: from 8 on a bucket takes the heavy path, which is built for a
    // few enormous buckets (a workgroup per job), not for ALL of them -- 2^19 points over a 16-bit table (256 entries per
    // bucket, M = 32) took 9.9 ms per sum instead of 1.6.  No longer than needed: a slice is a serial chain, and 2^18
    // points at M = 64 leave one wave per SIMD (Groth16 at 2^18: 7.0 -> 7.7 ms).
    while (pl.M < 1024 && (u64)pl.M * 5 * pl.G < total) pl.M *= 2;
    // Short sums are latency-bound and a slice is a serial chain of M additions (10 us each): while there are fewer
    // than 2^17 threads (two waves per SIMD), halve it -- down to 8, and never past the five-slice rule above.
    while (pl.M > 8 && total / (u64)pl.M < (1u << 17) && (u64)(pl.M / 2) * 5 * pl.G >= total) pl.M /= 2;
    pl.SEG = 8;
    return pl;
}

// ---------------------------------------------------------------------------------------
// 1. digits + bucket ranks
// ---------------------------------------------------------------------------------------
// Signed digits are made independent of each other by adding the constant
// C = sum_{w < W-1} 2^(c*w + c-1) once: with k' = k + C, d_w = ((k' >> cw) & mask) - 2^(c-1) for
// w < W-1 and d_{W-1} = k' >> c(W-1) (non-negative, <= 2^(c-1)); sum d_w 2^(cw) = k.  So a
// workgroup can compute the digits of ONE window for a chunk of scalars.
//
// Scattered returning atomics run at only ~2e10/s chip-wide (each lane = one 64-byte request at the
// memory side), which used to make this the second most expensive kernel.  Here a workgroup takes
// one window and DIGITS_CHUNK scalars, groups its entries by counter cache line in LDS (counting
// sort on bucket >> 4), and issues the atomics in that order: the 64 lanes of a wave then touch a
// few lines instead of 64.  Output is in block-grouped order: (key, value = index | sign, rank).
PS_INL u32 limb_sel8(const u32* k, int i) {  // k[i] (0 beyond the end) without dynamic register indexing
    u32 v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) v = (i == j) ? k[j] : v;
    return v;
}
constexpr int DIGITS_THREADS = 1024;
#ifndef PS_DIGITS_PER_THREAD
#define PS_DIGITS_PER_THREAD 8  // 8192 scalars per workgroup: two partition workgroups per CU (64 KB of LDS each).  With 16 (one
                               // workgroup per CU, runs of 16 entries per bin) k_sort_partition took 108 instead of 93 us at 2^20
#endif
constexpr int DIGITS_PER_THREAD = PS_DIGITS_PER_THREAD;
constexpr int DIGITS_CHUNK = DIGITS_THREADS * DIGITS_PER_THREAD;  // 8192 scalars per workgroup
constexpr int DIGITS_BINS = 2048;

__device__ inline u32 wave_incl_scan(u32 v);
struct DigitConst { u32 w[8]; };  // the constant C above

__global__ void __launch_bounds__(DIGITS_THREADS) k_digits_grouped(const u32* __restrict__ scalars, u32 n, int c, int W, u32 NB,
                                                                   DigitConst cadd, int bin_shift, int fold_neg, int single_set,
                                                                   u32* __restrict__ counts, u32* __restrict__ ent_key,
                                                                   u32* __restrict__ ent_val, u32* __restrict__ ent_rank) {
    __shared__ u32 hist[DIGITS_BINS];
    __shared__ u32 wtot[DIGITS_THREADS / 64];
    extern __shared__ __align__(16) unsigned char dg_smem[];
    u32* lkey = reinterpret_cast<u32*>(dg_smem);
    u32* lval = lkey + DIGITS_CHUNK;
    const int w = blockIdx.y;
    const u32 chunk_base = blockIdx.x * DIGITS_CHUNK;
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i < DIGITS_BINS; i += DIGITS_THREADS) hist[i] = 0;
    __syncthreads();
    const u32 mask = (1u << c) - 1u;
    const int bit = w * c, li = bit >> 5, sh = bit & 31;
    const bool top = w == W - 1;
    u32 dig[DIGITS_PER_THREAD];  // (magnitude-1) | sign << 31, or 0xffffffff for a zero digit
#pragma unroll
    for (int j = 0; j < DIGITS_PER_THREAD; j++) {
        const u32 i = chunk_base + j * DIGITS_THREADS + tid;
        u32 code = 0xffffffffu;
        if (i < n) {
            u32 k[8];
            const uint4* sp = reinterpret_cast<const uint4*>(scalars) + 2 * (size_t)i;
            uint4 a = sp[0], b = sp[1];
            k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
            // Witness vectors (Value.ToFieldElement, curve.go:17-19): a negative value v is stored as r - |v|.
            // (r - |v|) P = -(|v| P) for a point of order r, so the short scalar |v| is used with the point
            // negated and the vector keeps its few windows.  Such an element is recognised by its top word.
            u32 flip = 0;
            if (fold_neg && k[7] != 0) {
                u32 borrow = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    u64 t = (u64)FrParams::mod(q) - k[q] - borrow;
                    k[q] = (u32)t;
                    borrow = (u32)(t >> 63);
                }
                flip = 0x80000000u;
            }
            u32 carry = 0;  // k' = k + C
#pragma unroll
            for (int q = 0; q < 8; q++) {
                u64 t = (u64)k[q] + cadd.w[q] + carry;
                k[q] = (u32)t;
                carry = (u32)(t >> 32);
            }
            u64 two = (u64)limb_sel8(k, li) | ((u64)limb_sel8(k, li + 1) << 32);
            if (li == 7) two |= (u64)carry << 32;  // bit 256 of k' (only the top window can see it)
            u32 raw = top ? (u32)(two >> sh) : ((u32)(two >> sh) & mask);
            int d = top ? (int)raw : (int)raw - (int)NB;
            if (d != 0) {
                u32 mag = d < 0 ? (u32)(-d) : (u32)d;
                code = (mag - 1u) | ((d < 0 ? 0x80000000u : 0u) ^ flip);
                atomicAdd(&hist[(mag - 1u) >> bin_shift], 1u);
            }
        }
        dig[j] = code;
    }
    __syncthreads();
    // exclusive scan of the 2048 bins: 2 per thread
    u32 h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
    u32 inc = wave_incl_scan(h0 + h1);
    if ((tid & 63) == 63) wtot[tid >> 6] = inc;
    __syncthreads();
    u32 base = 0, total = 0;
    for (int q = 0; q < DIGITS_THREADS / 64; q++) {
        u32 t = wtot[q];
        if (q < (int)(tid >> 6)) base += t;
        total += t;
    }
    u32 ex = base + inc - (h0 + h1);
    __syncthreads();
    hist[2 * tid] = ex;
    hist[2 * tid + 1] = ex + h0;
    __syncthreads();
    const u32 wtag = single_set ? (u32)w << ENTRY_W_SHIFT : 0u;  // with a window table the entry names its window
    const u32 key_base = single_set ? 0u : (u32)w * NB;          // ... and all windows share one bucket set
#pragma unroll
    for (int j = 0; j < DIGITS_PER_THREAD; j++) {
        u32 code = dig[j];
        if (code != 0xffffffffu) {
            u32 m1 = code & 0x7fffffffu;
            u32 pos = atomicAdd(&hist[m1 >> bin_shift], 1u);
            lkey[pos] = m1;
            lval[pos] = (chunk_base + j * DIGITS_THREADS + tid) | (code & 0x80000000u) | wtag;
        }
    }
    __syncthreads();
    const size_t out = (size_t)w * n + chunk_base;
    const u32 limit = n - chunk_base < (u32)DIGITS_CHUNK ? n - chunk_base : (u32)DIGITS_CHUNK;
    for (u32 p0 = 0; p0 < limit; p0 += DIGITS_THREADS) {
        const u32 p = p0 + tid;
        u32 key = 0xffffffffu, val = 0, rank = 0;
        const bool act = p < total;
        if (act) { key = key_base + lkey[p]; val = lval[p]; }
        // lanes that share a bucket are served by one atomic of their count: up to 8 rounds, each
        // serving the group of the first unserved lane (narrow top windows and skewed witnesses put a
        // whole wave into a handful of buckets); what is left goes one atomic per lane
        bool done = !act;
        for (int round = 0; round < 8; round++) {
            const unsigned long long pending = __ballot(!done);
            if (!pending) break;
            const int leader = __ffsll((long long)pending) - 1;
            const u32 lk = __shfl(key, leader, 64);
            const bool mine = !done && key == lk;
            const unsigned long long same = __ballot(mine);
            if (__popcll(same) == 1) break;  // nothing to share
            u32 b0 = 0;
            if ((int)(tid & 63) == leader) b0 = atomicAdd(&counts[lk], (u32)__popcll(same));
            b0 = __shfl(b0, leader, 64);
            if (mine) {
                rank = b0 + (u32)__popcll(same & ((1ull << (tid & 63)) - 1ull));
                done = true;
            }
        }
        if (!done) rank = atomicAdd(&counts[key], 1u);
        if (p < limit) {
            ent_key[out + p] = key;
            ent_val[out + p] = val;
            ent_rank[out + p] = rank;
        }
    }
}

// ---------------------------------------------------------------------------------------
// 2. exclusive scan of u32 counters (2048 items per 256-thread block)
// ---------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;

__device__ inline u32 wave_incl_scan(u32 v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, total in *total
__device__ inline u32 block_excl_scan(u32 v, u32* total) {
    __shared__ u32 wsum[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    u32 inc = wave_incl_scan(v);
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
    for (int w = 0; w < nw; w++) {
        u32 s = wsum[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_blocks(const u32* __restrict__ in, u32* __restrict__ out,
                                                            u32* __restrict__ bsum, u64 n) {
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS];
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        v[j] = (base + j < n) ? in[base + j] : 0u;
        s += v[j];
    }
    u32 total;
    u32 ex = block_excl_scan(s, &total);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        if (base + j < n) out[base + j] = ex;
        ex += v[j];
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// single block: exclusive scan of the per-tile sums in place; grand total appended at out_total
__global__ void __launch_bounds__(1024) k_scan_top(u32* __restrict__ bsum, u32 nb, u32* __restrict__ out_total) {
    __shared__ u32 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u32 start = 0; start < nb; start += 1024) {
        u32 idx = start + threadIdx.x;
        u32 v = idx < nb ? bsum[idx] : 0u;
        u32 total;
        u32 ex = block_excl_scan(v, &total);
        u32 carry = carry_s;
        if (idx < nb) bsum[idx] = ex + carry;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_total = carry_s;
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_add(u32* __restrict__ out, const u32* __restrict__ bsum, u64 n) {
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u32 add = bsum[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++)
        if (base + j < n) out[base + j] += add;
}

// ---------------------------------------------------------------------------------------
// 3. scatter
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scatter(const u32* __restrict__ ent_key, const u32* __restrict__ ent_val,
                                                 const u32* __restrict__ ent_rank, const u32* __restrict__ offs, u64 total,
                                                 u32* __restrict__ sorted) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    u32 key = ent_key[idx];
    if (key == 0xffffffffu) return;
    sorted[offs[key] + ent_rank[idx]] = ent_val[idx];
}

// ---------------------------------------------------------------------------------------
// 1'-3'. The sort without per-entry global atomics (used whenever the bucket count G <= 2^20; the kernels
// above remain for larger forced windows).  A returning atomic per entry on W * 2^(c-1) scattered counters
// costs ~2e10/s chip-wide and collapses when all windows share ONE bucket set (window tables: every workgroup
// hammers the same lines).  Two-level counting sort instead, most significant part first:
//   k_sort_count      digits -> codes[w][i] (kept for the next kernel), LDS histogram of the coarse bin
//                     key >> 10 per workgroup, one global add per (workgroup, non-empty bin)
//   k_sort_scan       exclusive scan of the <= 1024 coarse bins
//   k_sort_partition  entries grouped by coarse bin in LDS, one returning global add per (workgroup, bin)
//                     reserves the run, coalesced writes of (fine key, value) into the bin's region
//   k_sort_fine       one workgroup per coarse bin: LDS histogram of its 2^fb fine keys, scan -> offs[],
//                     second walk places the values: sorted[] and offs[] exactly as the old pipeline left them
// fb (5..10 fine bits) is chosen so that there are several hundred coarse bins whatever the bucket count: a short-scalar
// plan has 20 K buckets, and twenty workgroups walking 130 K entries each would leave the chip idle.
// ---------------------------------------------------------------------------------------
constexpr int SORT_FINE_BITS = 10;
constexpr u32 SORT_FINE = 1u << SORT_FINE_BITS;
constexpr u32 SORT_MAX_COARSE = 1024;
constexpr u64 SORT_MAX_BUCKETS = (u64)SORT_FINE * SORT_MAX_COARSE;
static_assert(DIGITS_THREADS == (int)SORT_MAX_COARSE && SORT_FINE == SORT_MAX_COARSE, "one thread per histogram bin");

// k' = k + C of scalar i (DigitConst: the signed digits of k are the windows of k' minus NB, independently of each other);
// a negative witness value r - |v| is folded to |v| with `flip` = the sign bit its digits' codes get
__device__ inline void scalar_plus_c(const u32* __restrict__ scalars, u32 i, const DigitConst& cadd, int fold_neg, u32 (&k)[8], u32& carry,
                                     u32& flip) {
    const uint4* sp = reinterpret_cast<const uint4*>(scalars) + 2 * (size_t)i;
    uint4 a = sp[0], b = sp[1];
    k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
    flip = 0;
    if (fold_neg && k[7] != 0) {  // r - |v| of a negative witness value: |v| with the point negated
        u32 borrow = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            u64 t = (u64)FrParams::mod(q) - k[q] - borrow;
            k[q] = (u32)t;
            borrow = (u32)(t >> 63);
        }
        flip = 0x80000000u;
    }
    carry = 0;  // k' = k + C
#pragma unroll
    for (int q = 0; q < 8; q++) {
        u64 t = (u64)k[q] + cadd.w[q] + carry;
        k[q] = (u32)t;
        carry = (u32)(t >> 32);
    }
}
// digit of window w from k': (magnitude - 1) | sign << 31, or 0xffffffff for a zero digit (see k_digits_grouped)
__device__ inline u32 digit_from(const u32 (&k)[8], u32 carry, int c, int W, int w, u32 NB, u32 flip) {
    const u32 mask = (1u << c) - 1u;
    const int bit = w * c, li = bit >> 5, sh = bit & 31;
    const bool top = w == W - 1;
    u64 two = (u64)limb_sel8(k, li) | ((u64)limb_sel8(k, li + 1) << 32);
    if (li == 7) two |= (u64)carry << 32;
    u32 raw = top ? (u32)(two >> sh) : ((u32)(two >> sh) & mask);
    int d = top ? (int)raw : (int)raw - (int)NB;
    if (d == 0) return 0xffffffffu;
    u32 mag = d < 0 ? (u32)(-d) : (u32)d;
    return (mag - 1u) | ((d < 0 ? 0x80000000u : 0u) ^ flip);
}
__device__ inline u32 digit_code(const u32* __restrict__ scalars, u32 i, int c, int W, int w, u32 NB, const DigitConst& cadd,
                                 int fold_neg) {
    u32 k[8], carry, flip;
    scalar_plus_c(scalars, i, cadd, fold_neg, k, carry, flip);
    return digit_from(k, carry, c, W, w, NB, flip);
}

// LDS counter updates with one shortcut: the lanes that share the counter of the wave's first active lane are served by
// ONE add when there are at least eight of them (a hot bucket -- the ones of a witness, a nearly empty top window -- puts
// most of a wave onto one counter, and a 60-way same-address LDS atomic costs ~600 cycles); the rest go lane by lane.
// Must be called wave-uniformly.
__device__ inline void lds_count(u32* ctr, u32 bin, bool active) {
    const unsigned long long act = __ballot(active);
    if (!act) return;
    const int leader = __ffsll((long long)act) - 1;
    const u32 lb = __shfl(bin, leader, 64);
    const bool mine = active && bin == lb;
    const unsigned long long same = __ballot(mine);
    if (__popcll(same) >= 8) {
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&ctr[lb], (u32)__popcll(same));
        if (active && !mine) atomicAdd(&ctr[bin], 1u);
    } else if (active) {
        atomicAdd(&ctr[bin], 1u);
    }
}
__device__ inline u32 lds_rank(u32* ctr, u32 bin, bool active) {
    const unsigned long long act = __ballot(active);
    if (!act) return 0;
    const int lane = (int)(threadIdx.x & 63);
    const int leader = __ffsll((long long)act) - 1;
    const u32 lb = __shfl(bin, leader, 64);
    const bool mine = active && bin == lb;
    const unsigned long long same = __ballot(mine);
    u32 pos = 0;
    if (__popcll(same) >= 8) {
        u32 b0 = 0;
        if (lane == leader) b0 = atomicAdd(&ctr[lb], (u32)__popcll(same));
        b0 = __shfl(b0, leader, 64);
        if (mine) pos = b0 + (u32)__popcll(same & ((1ull << lane) - 1ull));
        else if (active) pos = atomicAdd(&ctr[bin], 1u);
    } else if (active) {
        pos = atomicAdd(&ctr[bin], 1u);
    }
    return pos;
}

// One pass over the scalars for ALL windows: a thread reads its scalar once and walks its W digits.  (Until round 3 the grid
// had a row of workgroups per window, each reading the scalars again -- 13 x 32 MB through the Infinity Cache for 2^20
// full-width scalars: 62 us for a kernel whose own traffic is 86 MB.)
constexpr int COUNT_PER_THREAD = 2;  // scalars per thread: 2048 per workgroup, 512 workgroups at 2^20
__global__ void __launch_bounds__(DIGITS_THREADS) k_sort_count(const u32* __restrict__ scalars, u32 n, int c, int W, u32 NB,
                                                               DigitConst cadd, int fold_neg, int single_set, u32 ncoarse, int fb,
                                                               u32* __restrict__ codes, u32* __restrict__ coarse_cnt) {
    PS_TAIL_PRIO_HERE;
    __shared__ u32 hist[SORT_MAX_COARSE];
    const u32 tid = threadIdx.x;
    hist[tid] = 0;  // DIGITS_THREADS == SORT_MAX_COARSE
    __syncthreads();
    for (int j = 0; j < COUNT_PER_THREAD; j++) {
        const u32 i = (blockIdx.x * COUNT_PER_THREAD + (u32)j) * DIGITS_THREADS + tid;
        const bool in = i < n;
        u32 k[8], carry = 0, flip = 0;
        if (in) scalar_plus_c(scalars, i, cadd, fold_neg, k, carry, flip);
        for (int w = 0; w < W; w++) {  // uniform trip count: lds_count is called by whole waves
            u32 code = 0xffffffffu;
            if (in) {
                code = digit_from(k, carry, c, W, w, NB, flip);
                codes[(size_t)w * n + i] = code;
            }
            const u32 key_base = single_set ? 0u : (u32)w * NB;
            lds_count(hist, (key_base + (code & 0x7fffffffu)) >> fb, code != 0xffffffffu);
        }
    }
    __syncthreads();
    if (tid < ncoarse && hist[tid]) atomicAdd(&coarse_cnt[tid], hist[tid]);
}

// A coarse bin that holds more than SORT_BIG entries (hot buckets: all-equal scalars, the wire values of a boolean
// circuit, a carry-only window) is not walked by its one workgroup -- 2^19 entries took it 1.5 ms -- but cut into tiles of
// SORT_TILE entries that any workgroup takes: k_sort_big_count (LDS histogram per tile, one global add per (tile, bucket)),
// k_sort_big_scan (per big bin: scan of its 2^fb counters), k_sort_big_scatter (per tile: one returning add per (tile,
// bucket) reserves the run, LDS ranks inside it).
constexpr u32 SORT_BIG = 1u << 16;
constexpr u32 SORT_TILE = 1u << 14;

// one workgroup: coarse_off = exclusive scan of coarse_cnt, coarse_cur = a copy (the partition kernel's cursors),
// grand total -> coarse_off[ncoarse] and offs[G]; tile_base = exclusive scan of the big bins' tile counts
__global__ void __launch_bounds__(SORT_MAX_COARSE) k_sort_scan(const u32* __restrict__ coarse_cnt, u32 ncoarse, u32* __restrict__ coarse_off,
                                                               u32* __restrict__ coarse_cur, u32* __restrict__ offs_total,
                                                               u32* __restrict__ tile_base) {
    PS_TAIL_PRIO_HERE;
    const u32 tid = threadIdx.x;
    const u32 cnt = tid < ncoarse ? coarse_cnt[tid] : 0u;
    u32 total;
    const u32 ex = block_excl_scan(cnt, &total);
    if (tid < ncoarse) { coarse_off[tid] = ex; coarse_cur[tid] = ex; }
    if (tid == 0) { coarse_off[ncoarse] = total; *offs_total = total; }
    __syncthreads();
    const u32 tiles = cnt > SORT_BIG ? (cnt + SORT_TILE - 1) / SORT_TILE : 0u;
    u32 ttotal;
    const u32 tex = block_excl_scan(tiles, &ttotal);
    if (tid < ncoarse) tile_base[tid] = tex;
    if (tid == 0) tile_base[ncoarse] = ttotal;
}

// (16 waves of at most 64 registers: the block fits beside ONE wave per SIMD of another sum's accumulation, 240 + 256 <= 512)
__global__ void __launch_bounds__(DIGITS_THREADS, 8) k_sort_partition(const u32* __restrict__ codes, u32 n, u32 NB, int single_set, int fb,
                                                                   u32* __restrict__ coarse_cur, unsigned short* __restrict__ part_key,
                                                                   u32* __restrict__ part_val) {
    PS_TAIL_PRIO_HERE;
    __shared__ u32 hist[SORT_MAX_COARSE], binstart[SORT_MAX_COARSE], gbase[SORT_MAX_COARSE];
    __shared__ u32 wtot[DIGITS_THREADS / 64];
    extern __shared__ __align__(16) unsigned char dg_smem[];
    u32* lkey = reinterpret_cast<u32*>(dg_smem);
    u32* lval = lkey + DIGITS_CHUNK;
    const int w = blockIdx.y;
    const u32 chunk_base = blockIdx.x * DIGITS_CHUNK;
    const u32 tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    const u32 key_base = single_set ? 0u : (u32)w * NB;
    const u32 wtag = single_set ? (u32)w << ENTRY_W_SHIFT : 0u;
    u32 dig[DIGITS_PER_THREAD];
#pragma unroll
    for (int j = 0; j < DIGITS_PER_THREAD; j++) {
        const u32 i = chunk_base + j * DIGITS_THREADS + tid;
        const u32 code = i < n ? codes[(size_t)w * n + i] : 0xffffffffu;
        dig[j] = code;
        lds_count(hist, (key_base + (code & 0x7fffffffu)) >> fb, code != 0xffffffffu);
    }
    __syncthreads();
    // exclusive scan of the 1024 bins, one per thread; reserve each non-empty bin's run in its global region
    const u32 cnt = hist[tid];
    const u32 inc = wave_incl_scan(cnt);
    if ((tid & 63) == 63) wtot[tid >> 6] = inc;
    __syncthreads();
    u32 before = 0, total = 0;
    for (int q = 0; q < DIGITS_THREADS / 64; q++) {
        const u32 t = wtot[q];
        if (q < (int)(tid >> 6)) before += t;
        total += t;
    }
    const u32 ex = before + inc - cnt;
    __syncthreads();
    binstart[tid] = ex;
    hist[tid] = ex;  // now the bin's fill cursor
    gbase[tid] = cnt ? atomicAdd(&coarse_cur[tid], cnt) : 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DIGITS_PER_THREAD; j++) {
        const u32 code = dig[j];
        const bool act = code != 0xffffffffu;
        const u32 key = key_base + (code & 0x7fffffffu);
        const u32 pos = lds_rank(hist, key >> fb, act);
        if (act) {
            lkey[pos] = key;
            lval[pos] = (chunk_base + j * DIGITS_THREADS + tid) | (code & 0x80000000u) | wtag;
        }
    }
    __syncthreads();
    for (u32 p = tid; p < total; p += DIGITS_THREADS) {
        const u32 key = lkey[p], b = key >> fb;
        const u32 dst = gbase[b] + (p - binstart[b]);
        part_key[dst] = (unsigned short)(key & ((1u << fb) - 1u));
        part_val[dst] = lval[p];
    }
}

__global__ void __launch_bounds__(SORT_FINE) k_sort_fine(const unsigned short* __restrict__ part_key, const u32* __restrict__ part_val,
                                                         const u32* __restrict__ coarse_off, u32 G, int fb, u32* __restrict__ offs,
                                                         u32* __restrict__ sorted) {
    PS_TAIL_PRIO_HERE;
    __shared__ u32 hist[SORT_FINE];
    __shared__ u32 wtot[SORT_FINE / 64];
    const u32 p = blockIdx.x, tid = threadIdx.x;
    const u32 lo = coarse_off[p], hi = coarse_off[p + 1];
    if (hi - lo > SORT_BIG) return;  // a big bin: the tile kernels below
    hist[tid] = 0;
    __syncthreads();
    // Four strides per trip, loads first: a partition that holds a hot bucket is walked by ONE workgroup (10^5 entries of
    // a witness's ones: 100 trips), and with one dependent global load per trip that walk was bound by memory latency
    // (0.55 ms, as long as the accumulation of the whole sum).
    constexpr u32 UNR = 4;
    for (u32 i0 = lo; i0 < hi; i0 += UNR * SORT_FINE) {
        u32 k[UNR];
        bool act[UNR];
#pragma unroll
        for (u32 u = 0; u < UNR; u++) {
            const u32 i = i0 + u * SORT_FINE + tid;
            act[u] = i < hi;
            k[u] = act[u] ? (u32)part_key[i] : 0u;
        }
#pragma unroll
        for (u32 u = 0; u < UNR; u++) lds_count(hist, k[u], act[u]);
    }
    __syncthreads();
    const u32 cnt = hist[tid];
    const u32 inc = wave_incl_scan(cnt);
    if ((tid & 63) == 63) wtot[tid >> 6] = inc;
    __syncthreads();
    u32 before = 0;
    for (int q = 0; q < (int)(tid >> 6); q++) before += wtot[q];
    const u32 start = lo + before + inc - cnt;
    const u32 g = (p << fb) + tid;  // threads beyond the partition's 2^fb buckets own no bin (their count is 0)
    if (tid < (1u << fb) && g < G) offs[g] = start;
    __syncthreads();
    hist[tid] = start;  // cursor
    __syncthreads();
    for (u32 i0 = lo; i0 < hi; i0 += UNR * SORT_FINE) {
        u32 k[UNR], v[UNR];
        bool act[UNR];
#pragma unroll
        for (u32 u = 0; u < UNR; u++) {
            const u32 i = i0 + u * SORT_FINE + tid;
            act[u] = i < hi;
            k[u] = act[u] ? (u32)part_key[i] : 0u;
            v[u] = act[u] ? part_val[i] : 0u;
        }
#pragma unroll
        for (u32 u = 0; u < UNR; u++) {
            const u32 pos = lds_rank(hist, k[u], act[u]);
            if (act[u]) sorted[pos] = v[u];
        }
    }
}

// tile t of the big bins: its bin p (tile_base[p] <= t < tile_base[p+1]) and its entry range
__device__ inline bool sort_big_tile(u32 t, const u32* __restrict__ tile_base, const u32* __restrict__ coarse_off, u32 ncoarse,
                                     u32* p_out, u32* lo_out, u32* hi_out) {
    if (t >= tile_base[ncoarse]) return false;
    u32 lo = 0, hi = ncoarse;  // tile_base[lo] <= t < tile_base[hi]
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (tile_base[mid] <= t) lo = mid; else hi = mid;
    }
    // bins without tiles repeat their successor's base: step to the bin that owns tile t
    while (tile_base[lo + 1] <= t) lo++;
    const u32 first = coarse_off[lo] + (t - tile_base[lo]) * SORT_TILE, end = coarse_off[lo + 1];
    *p_out = lo;
    *lo_out = first;
    *hi_out = end - first > SORT_TILE ? first + SORT_TILE : end;
    return true;
}
constexpr u32 SORT_TILE_PER_THREAD = SORT_TILE / SORT_FINE;  // 16

__global__ void __launch_bounds__(SORT_FINE) k_sort_big_count(const unsigned short* __restrict__ part_key, const u32* __restrict__ coarse_off,
                                                              const u32* __restrict__ tile_base, u32 ncoarse, int fb,
                                                              u32* __restrict__ gcnt) {
    PS_TAIL_PRIO_HERE;
    __shared__ u32 hist[SORT_FINE];
    const u32 tid = threadIdx.x;
    for (u32 t = blockIdx.x;; t += gridDim.x) {
        u32 p, lo, hi;
        if (!sort_big_tile(t, tile_base, coarse_off, ncoarse, &p, &lo, &hi)) return;
        hist[tid] = 0;
        __syncthreads();
#pragma unroll 4
        for (u32 j = 0; j < SORT_TILE_PER_THREAD; j++) {
            const u32 i = lo + j * SORT_FINE + tid;
            const bool act = i < hi;
            lds_count(hist, act ? (u32)part_key[i] : 0u, act);
        }
        __syncthreads();
        if (tid < (1u << fb) && hist[tid]) atomicAdd(&gcnt[((size_t)p << fb) + tid], hist[tid]);
        __syncthreads();
    }
}
// per big bin: offs[] of its buckets, and the same values left in gcnt[] as the scatter's cursors
__global__ void __launch_bounds__(SORT_FINE) k_sort_big_scan(const u32* __restrict__ coarse_off, u32 G, int fb, u32* __restrict__ gcnt,
                                                             u32* __restrict__ offs) {
    PS_TAIL_PRIO_HERE;
    __shared__ u32 wtot[SORT_FINE / 64];
    const u32 p = blockIdx.x, tid = threadIdx.x;
    const u32 lo = coarse_off[p], hi = coarse_off[p + 1];
    if (hi - lo <= SORT_BIG) return;
    const u32 g = (p << fb) + tid;
    const bool own = tid < (1u << fb) && g < G;
    const u32 cnt = own ? gcnt[g] : 0u;
    const u32 inc = wave_incl_scan(cnt);
    if ((tid & 63) == 63) wtot[tid >> 6] = inc;
    __syncthreads();
    u32 before = 0;
    for (int q = 0; q < (int)(tid >> 6); q++) before += wtot[q];
    const u32 start = lo + before + inc - cnt;
    if (own) { offs[g] = start; gcnt[g] = start; }
}
__global__ void __launch_bounds__(SORT_FINE) k_sort_big_scatter(const unsigned short* __restrict__ part_key, const u32* __restrict__ part_val,
                                                                const u32* __restrict__ coarse_off, const u32* __restrict__ tile_base,
                                                                u32 ncoarse, int fb, u32* __restrict__ gcur, u32* __restrict__ sorted) {
    PS_TAIL_PRIO_HERE;
    __shared__ u32 hist[SORT_FINE];
    const u32 tid = threadIdx.x;
    for (u32 t = blockIdx.x;; t += gridDim.x) {
        u32 p, lo, hi;
        if (!sort_big_tile(t, tile_base, coarse_off, ncoarse, &p, &lo, &hi)) return;
        hist[tid] = 0;
        __syncthreads();
        u32 k[SORT_TILE_PER_THREAD];
#pragma unroll
        for (u32 j = 0; j < SORT_TILE_PER_THREAD; j++) {
            const u32 i = lo + j * SORT_FINE + tid;
            const bool act = i < hi;
            k[j] = act ? (u32)part_key[i] : 0xffffffffu;
            lds_count(hist, act ? k[j] : 0u, act);
        }
        __syncthreads();
        // this tile's run inside every bucket it touches: one returning global add per (tile, bucket)
        const u32 cnt = hist[tid];
        __syncthreads();
        hist[tid] = (tid < (1u << fb) && cnt) ? atomicAdd(&gcur[((size_t)p << fb) + tid], cnt) : 0u;
        __syncthreads();
#pragma unroll
        for (u32 j = 0; j < SORT_TILE_PER_THREAD; j++) {
            const bool act = k[j] != 0xffffffffu;
            const u32 pos = lds_rank(hist, act ? k[j] : 0u, act);
            if (act) sorted[pos] = part_val[lo + j * SORT_FINE + tid];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// element access: the bucket kernels are written once over a *kernel field* KF and a storage
// field.  Fp -> Fp (one lane per logical thread).  Fp2 storage -> Fp2s (two lanes per logical
// thread, each lane loads / stores its own component of the same Fp2 records in HBM or LDS).
// ---------------------------------------------------------------------------------------
template <class KF> struct FieldTraits { typedef KF Store; static constexpr int LANES = 1; };
template <> struct FieldTraits<Fp2s> { typedef Fp2 Store; static constexpr int LANES = 2; };
template <class S> struct KernelField { typedef S type; };
template <> struct KernelField<Fp2> { typedef Fp2s type; };

__device__ inline Fp ld_f(const Fp* p, const Fp*) { return *p; }
__device__ inline void st_f(Fp* p, const Fp& v) { *p = v; }
__device__ inline Fp2s ld_f(const Fp2* p, const Fp2s*) { return Fp2s{pair_lane() ? p->c1 : p->c0}; }
__device__ inline void st_f(Fp2* p, const Fp2s& v) { if (pair_lane()) p->c1 = v.v; else p->c0 = v.v; }

template <class KF>
__device__ inline Affine<KF> ld_affine(const Affine<typename FieldTraits<KF>::Store>* p) {
    Affine<KF> r;
    r.x = ld_f(&p->x, (const KF*)0);
    r.y = ld_f(&p->y, (const KF*)0);
    return r;
}
template <class KF>
__device__ inline Xyzz<KF> ld_xyzz(const Xyzz<typename FieldTraits<KF>::Store>* p) {
    Xyzz<KF> r;
    r.x = ld_f(&p->x, (const KF*)0);
    r.y = ld_f(&p->y, (const KF*)0);
    r.zz = ld_f(&p->zz, (const KF*)0);
    r.zzz = ld_f(&p->zzz, (const KF*)0);
    return r;
}
template <class KF>
__device__ inline void st_xyzz(Xyzz<typename FieldTraits<KF>::Store>* p, const Xyzz<KF>& v) {
    st_f(&p->x, v.x);
    st_f(&p->y, v.y);
    st_f(&p->zz, v.zz);
    st_f(&p->zzz, v.zzz);
}
template <class KF> __device__ inline u32 logical_tid() { return (blockIdx.x * blockDim.x + threadIdx.x) / FieldTraits<KF>::LANES; }
template <class KF> __device__ inline u32 logical_local() { return threadIdx.x / FieldTraits<KF>::LANES; }
template <class KF> __device__ inline u32 logical_block() { return blockDim.x / FieldTraits<KF>::LANES; }
template <class KF> __device__ inline bool pair_leader() { return FieldTraits<KF>::LANES == 1 || pair_lane() == 0; }

// LDS tree sum over the logical threads of a block; result in sm[0]
template <class KF>
__device__ inline void block_tree_sum(Xyzz<typename FieldTraits<KF>::Store>* sm, const Xyzz<KF>& mine) {
    const u32 lt = logical_local<KF>();
    st_xyzz<KF>(&sm[lt], mine);
    __syncthreads();
    for (u32 stride = logical_block<KF>() >> 1; stride > 0; stride >>= 1) {
        if (lt < stride) {
            Xyzz<KF> a = ld_xyzz<KF>(&sm[lt]), b = ld_xyzz<KF>(&sm[lt + stride]);
            xyzz_add_inl<KF>(a, b);
            st_xyzz<KF>(&sm[lt], a);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// 4. bucket accumulation over fixed slices of the sorted entry list
// ---------------------------------------------------------------------------------------
// Slice length a sum actually uses.  The plan is made before the sort has run, for W digits per scalar; a witness of bits or
// small values leaves one digit per scalar, and 2^19 entries in slices of 32 are 256 waves on a chip that holds 2 048 (the
// accumulation of Groth16's A over 2^20 booleanity gates: 0.46-0.60 ms for 0.06 ms of work).  So every kernel that walks the
// slices derives their length from the length of the sorted list E = offs[G], the planned slice count T and the planned M:
// M again as soon as the list is a quarter of the plan, shorter below, never under 4 (or a shorter planned M).
__device__ inline int eff_slice(u32 E, u32 T, int M) {
#if defined(PS_NO_EFF_SLICE)  // measurement builds: the planned length throughout
    return M;
#endif
    const u64 m = (4ull * E + T - 1) / (T ? T : 1u);
    const u64 lo = M < 4 ? (u64)M : 4ull;
    return m >= (u64)M ? M : (int)(m < lo ? lo : m);
}

template <class F>
PS_INL bool affine_is_identity(const Affine<F>& p) { return fp_all_zero(p.x) && fp_all_zero(p.y); }  // stored points are canonical

template <class KF>
__device__ inline void flush_run(const Xyzz<KF>& acc, u32 g, u32 rs, u32 re, u32 slice_start, u32 t,
                                 const u32* __restrict__ offs, Xyzz<typename FieldTraits<KF>::Store>* __restrict__ buckets,
                                 Xyzz<typename FieldTraits<KF>::Store>* __restrict__ parts) {
    bool whole = (rs == offs[g]) && (re == offs[g + 1]);
    if (whole) st_xyzz<KF>(&buckets[g], acc);
    else if (rs == slice_start) st_xyzz<KF>(&parts[2 * (size_t)t], acc);
    else st_xyzz<KF>(&parts[2 * (size_t)t + 1], acc);
}

// Point of a sorted entry.  A plain array is indexed by the entry's low 31 bits (w_stride = 0, rows of
// sizeof(Affine) bytes); a window table [w][w_stride] by index + window * w_stride with the window in the entry's
// bits ENTRY_W_SHIFT..30 and rows padded to `pstride` = 128 / 256 bytes, so that a point gathered at random from a
// table far larger than the Infinity Cache costs one HBM line, not the two a 112-byte record usually straddles.
template <class KF>
__device__ inline Affine<KF> ld_entry_point(const char* __restrict__ points, u32 e, u32 idx_mask, u64 w_stride, u32 pstride) {
    const size_t row = (size_t)(e & idx_mask) + (size_t)((e >> ENTRY_W_SHIFT) & 31u) * w_stride;
    return ld_affine<KF>(reinterpret_cast<const Affine<typename FieldTraits<KF>::Store>*>(points + row * pstride));
}

// Bucket that holds sorted position p (> the current bucket's end): usually the next one; behind a run of EMPTY buckets --
// skewed scalars leave most of 2^19 buckets empty, and walking them one dependent load at a time cost 40 ms on the wire
// values of a boolean circuit -- a binary search over offs[].
__device__ inline void next_bucket(const u32* __restrict__ offs, u32 G, u32 p, u32& g, u32& bend) {
    g++;
    bend = offs[g + 1];
    if (bend > p) return;
    u32 lo = g + 1, hi = G;  // offs[lo] <= p < offs[hi]
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (offs[mid] <= p) lo = mid; else hi = mid;
    }
    g = lo;
    bend = offs[g + 1];
}

// Waves per SIMD of the G2 (lane-pair) accumulation.  2: 256 VGPRs, 46 of the kernel's ~300 live registers spilled to
// scratch, no prefetch; 1: 256 VGPRs + 69 AGPRs, nothing spilled, next point prefetched.  A/B on one box (tools/ab_libs.sh):
// a lone G2 sum 8.9 vs 8.8 ms, three in flight 7.97 vs 8.17, PHGR13Prove 29.1 vs 30.7 -- a 300-register wave leaves no room
// on its SIMD for the waves of the other sums' tails, so wherever the kernel shares the chip the spilling variant wins.
#ifndef PS_G2_ACC_WAVES
#define PS_G2_ACC_WAVES 2
#endif
#define PS_ACC_WAVES(KF) (FieldTraits<KF>::LANES == 2 ? PS_G2_ACC_WAVES : 2)
// PREFETCH: the next entry's point is requested before the current addition starts (28 more VGPRs), so that
// the ~2 us of an HBM gather hide under the ~7 us of the addition even when both waves of a SIMD miss together.
template <class KF, bool PREFETCH>
__global__ void __launch_bounds__(256, PS_ACC_WAVES(KF)) k_accumulate(const char* __restrict__ points,
                                                       const u32* __restrict__ sorted, const u32* __restrict__ offs,
                                                       u32 G, int Mplan, u32 T, u32 idx_mask, u64 w_stride, u32 pstride,
                                                       Xyzz<typename FieldTraits<KF>::Store>* __restrict__ buckets,
                                                       Xyzz<typename FieldTraits<KF>::Store>* __restrict__ parts,
                                                       u32* __restrict__ heavy_count) {
    const u32 E = offs[G];
    const int M = eff_slice(E, T, Mplan);
    const u32 t = logical_tid<KF>();
    if (blockIdx.x == 0 && threadIdx.x == 0) *heavy_count = 0;  // the fix-up's list of heavy buckets starts empty (no memset launch)
    const u64 start64 = (u64)t * (u64)M;
    if (start64 >= E) return;
    const u32 start = (u32)start64;
    const u32 end = (E - start < (u32)M) ? E : start + (u32)M;
    // bucket containing `start`: largest g with offs[g] <= start (and offs[g+1] > start)
    u32 lo = 0, hi = G;  // invariant: offs[lo] <= start < offs[hi]
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (offs[mid] <= start) lo = mid; else hi = mid;
    }
    u32 g = lo;
    u32 bend = offs[g + 1];
    u32 run_start = start;
    Xyzz<KF> acc = xyzz_identity<KF>();
    if (PREFETCH) {
        u32 e = sorted[start];
        u32 e1 = start + 1 < end ? sorted[start + 1] : e;  // entries run two ahead of the additions, points one ahead
        Affine<KF> pt = ld_entry_point<KF>(points, e, idx_mask, w_stride, pstride);
        for (u32 p = start; p < end; p++) {
            Affine<KF> nxt = pt;
            u32 e2 = e1;
            if (p + 1 < end) nxt = ld_entry_point<KF>(points, e1, idx_mask, w_stride, pstride);
            if (p + 2 < end) e2 = sorted[p + 2];
            if (p >= bend) {
                flush_run<KF>(acc, g, run_start, p, start, t, offs, buckets, parts);
                acc = xyzz_identity<KF>();
                run_start = p;
                next_bucket(offs, G, p, g, bend);
            }
            if (!affine_is_identity<KF>(pt)) {
                if (e >> 31) pt.y = f_neg(pt.y);
                xyzz_madd_inl<KF>(acc, pt.x, pt.y);
            }
            pt = nxt; e = e1; e1 = e2;
        }
    } else {
        for (u32 p = start; p < end; p++) {
            if (p >= bend) {
                flush_run<KF>(acc, g, run_start, p, start, t, offs, buckets, parts);
                acc = xyzz_identity<KF>();
                run_start = p;
                next_bucket(offs, G, p, g, bend);
            }
            const u32 e = sorted[p];
            Affine<KF> pt = ld_entry_point<KF>(points, e, idx_mask, w_stride, pstride);
            if (!affine_is_identity<KF>(pt)) {
                if (e >> 31) pt.y = f_neg(pt.y);
                xyzz_madd_inl<KF>(acc, pt.x, pt.y);
            }
        }
    }
    flush_run<KF>(acc, g, run_start, end, start, t, offs, buckets, parts);
}

// ---------------------------------------------------------------------------------------
// 5. fix-up of buckets cut by slice boundaries
// ---------------------------------------------------------------------------------------
constexpr u32 HEAVY_SPAN = 8;  // buckets cut into more slices than this go to the heavy-bucket kernels

// Waves per SIMD the tail kernels are compiled for.  2 = at most 256 registers per wave.  With "1" the compiler took 306
// (k_reduce_l1<Fp>: 256 VGPRs + 50 AGPRs) and 357 (G2): such a wave cannot share a SIMD's 512 registers with ONE wave of
// another sum's accumulation (234 / 256), so with sums in flight the running sums waited for the accumulation to drain --
// 2.4 ms instead of 0.26 in a kernel trace of 2^20-point sums three in flight -- and the pipeline stalled behind them
// (2.78 ms per sum for 2.30 ms of accumulation).
#ifndef PS_TAIL_WAVES
#define PS_TAIL_WAVES 2
#endif

template <class KF>
__global__ void __launch_bounds__(256, PS_TAIL_WAVES) k_fixup(const u32* __restrict__ offs, u32 G, int Mplan, u32 T,
                                                  const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ parts,
                                                  Xyzz<typename FieldTraits<KF>::Store>* __restrict__ buckets,
                                                  u32* __restrict__ heavy_count, u32* __restrict__ heavy_list) {
    PS_TAIL_PRIO_HERE;
    u32 g = logical_tid<KF>();
    if (g >= G) return;
    const int M = eff_slice(offs[G], T, Mplan);
    u32 lo = offs[g], hi = offs[g + 1];
    if (lo == hi) {  // empty bucket: the identity (no memset of the bucket array: every bucket is written by exactly one of the
                     // accumulation's flush, this kernel and k_qfixup_heavy)
        st_xyzz<KF>(&buckets[g], xyzz_identity<KF>());
        return;
    }
    u32 t0 = lo / (u32)M, t1 = (hi - 1) / (u32)M;
    if (t0 == t1) return;  // lay inside one slice: already final
    if (t1 - t0 >= HEAVY_SPAN) {  // long chain: one workgroup per bucket instead of one thread
        if (pair_leader<KF>()) heavy_list[atomicAdd(heavy_count, 1u)] = g;
        return;
    }
    Xyzz<KF> acc = xyzz_identity<KF>();
    for (u32 t = t0; t <= t1; t++) {
        u32 slice_start = t * (u32)M;
        u32 rs = lo > slice_start ? lo : slice_start;
        Xyzz<KF> part = ld_xyzz<KF>(&parts[2 * (size_t)t + (rs == slice_start ? 0 : 1)]);
        xyzz_add_inl<KF>(acc, part);
    }
    st_xyzz<KF>(&buckets[g], acc);
}

// Heavy buckets (skewed scalars: a witness that is half ones puts n/2 entries into one bucket).  The
// partial slots of the slices t0..t1 of such a bucket are summed in two levels: the slices are cut into
// jobs, any workgroup takes any job (strided sums, then a tree), and a last kernel adds up the job results
// of each bucket -- k_qfixup_heavy_part / k_qfixup_heavy in qtail.hpp, on lane quads.  The dependency chain
// is ~2 (s + 6) quad additions whatever the skew (heavy_chunk_of), and a single enormous bucket still uses the chip.
constexpr u32 HEAVY_CHUNK = 1024;  // (job size of the one-lane kernels of rounds 1-2; k_heavy_jobs keeps it for npb = 0)
// Slices per job of a heavy bucket that spans `span` slices, for the quad-tree kernels (qtail.hpp; npb = partial sums per block:
// 64 G1, 32 G2): npb * s with s = ceil(sqrt(span) / npb), so that both levels -- a job's s strided additions and a tree of
// log2(npb), then the same over the bucket's ~sqrt(span) jobs -- are equally deep whatever the size of the bucket.
__host__ __device__ inline u32 heavy_chunk_of(u32 span, u32 npb) {
    u32 s = 1;
    while ((u64)s * s * npb * npb < span) s++;
    return npb * s;
}

// job_base[h] = number of jobs of the heavy buckets before h; job_base[nheavy] = total.  One workgroup.
__global__ void __launch_bounds__(256) k_heavy_jobs(const u32* __restrict__ offs, u32 G, int Mplan, u32 T, const u32* __restrict__ heavy_count,
                                                    const u32* __restrict__ heavy_list, u32* __restrict__ job_base, u32 npb) {
    PS_TAIL_PRIO_HERE;
    const int M = eff_slice(offs[G], T, Mplan);
    __shared__ u32 wsum[4];
    const u32 nheavy = *heavy_count;
    const u32 tid = threadIdx.x;
    u32 running = 0;
    for (u32 base = 0; base < nheavy; base += 256) {
        const u32 h = base + tid;
        u32 cnt = 0;
        if (h < nheavy) {
            const u32 g = heavy_list[h];
            const u32 t0 = offs[g] / (u32)M, t1 = (offs[g + 1] - 1) / (u32)M;
            const u32 chunk = npb ? heavy_chunk_of(t1 - t0 + 1, npb) : HEAVY_CHUNK;
            cnt = (t1 - t0 + chunk) / chunk;  // ceil((t1 - t0 + 1) / chunk)
        }
        u32 inc = wave_incl_scan(cnt);
        if ((tid & 63) == 63) wsum[tid >> 6] = inc;
        __syncthreads();
        u32 before = 0, tile = 0;
        for (u32 q = 0; q < 4; q++) {
            if (q < (tid >> 6)) before += wsum[q];
            tile += wsum[q];
        }
        if (h < nheavy) job_base[h] = running + before + inc - cnt;
        running += tile;
        __syncthreads();
    }
    if (tid == 0) job_base[nheavy] = running;
}

// (The one-lane kernels that summed a heavy bucket's jobs in rounds 1-2 -- strided chains and an LDS tree per job of 1024
// slices -- are gone: every plan uses the two levels of quad trees in qtail.hpp, k_qfixup_heavy_part / k_qfixup_heavy.)

// ---------------------------------------------------------------------------------------
// 6. bucket reduction: window sum = sum_{b=0}^{NB-1} (b+1) * B[b]
// ---------------------------------------------------------------------------------------
// A point addition is ~14 multiplications (about 16 us for a wave, alone on its SIMD or not), and a sum that is
// pipelined behind others pays for every wave-step of this phase: the levels are chosen for little work first, short
// dependency chains second.  With segs = NB / 8 segments per bucket set and m = max(1, segs / 8):
//   L1   one thread per segment of 8 buckets: running sums  acc_s = sum (b-8s+1) B[b], run_s = sum B[b]
//        (14 additions deep, NB/8 threads; window sum = sum_s acc_s + 8 * sum_s s * run_s)
//   PYR  the weight s = 8 j + i of run_s is split into its low three bits and j: five sums over each group of 8
//        segments, one thread each -- A1_j = sum_i acc, Q0_j / Q1_j / Q2_j = sum of the run_s whose i has bit 0 / 1 / 2,
//        R1_j = sum_i run (3 or 7 additions deep, 5 m threads)
//   SUM  per bucket set 4 + log2(m) jobs: the plain sums of A1, Q0, Q1, Q2 over j and, for every bit k of j, T_k = the
//        sum of the R1_j whose j has bit k.  512 terms per workgroup of 64 (8 in a row, then an LDS tree), the pieces of
//        a job summed by FIN.  (Round 1 summed all 2^(c-4) run_s once per bit of s: 8.5 additions per segment where
//        this takes 23/8 + 7.5/8.)
//   host the weights: sum = A + sum_q 2^(q+3) Q_q + sum_k 2^(k+6) T_k, a Horner chain of c-1 doublings per bucket set
//        that takes the CPU a few tens of microseconds (msm_fold_host) and was 19 dependent doublings, 0.3 ms, on a GPU
//        lane.
#ifndef PS_RED_SEG_LOG
#define PS_RED_SEG_LOG 3
#endif
constexpr int RED_SEG_LOG = PS_RED_SEG_LOG;
constexpr int RED_SEG = 1 << RED_SEG_LOG;
constexpr int RED_DIRECT_JOBS = 4;   // A, Q0, Q1, Q2
constexpr u32 RED_SUM_LANES = 64;    // logical threads of a SUM / FIN workgroup
#ifndef PS_RED_SUM_TERMS
#define PS_RED_SUM_TERMS 512
#endif
constexpr u32 RED_SUM_TERMS = PS_RED_SUM_TERMS;   // terms of a direct job per SUM workgroup

struct ReducePlan {
    u32 segs;   // segments (of RED_SEG buckets) per bucket set
    u32 m;      // groups of 8 segments per bucket set
    int kb;     // log2(m): bit jobs
    u32 njobs;  // RED_DIRECT_JOBS + kb results per bucket set, in the order A, Q0, Q1, Q2, T_0 .. T_{kb-1}
    u32 nblk;   // SUM workgroups per job
    bool small; // NB <= RED_SMALL_NB: one kernel, njobs = c results V_0 .. V_{c-1} (k_reduce_small)
};
constexpr u32 RED_SMALL_NB = 1024;
static inline ReducePlan reduce_plan(u32 NB) {
    ReducePlan r;
    r.small = NB <= RED_SMALL_NB;
    if (r.small) {
        r.segs = r.m = 0;
        r.kb = 0;
        r.njobs = 1;
        while ((1u << (r.njobs - 1)) < NB) r.njobs++;  // NB = 2^(c-1): c jobs
        r.nblk = 1;
        return r;
    }
    r.segs = NB >> RED_SEG_LOG;
    r.m = r.segs >= 8 ? r.segs / 8 : 1;
    r.kb = 0;
    while ((1u << r.kb) < r.m) r.kb++;
    r.njobs = RED_DIRECT_JOBS + (u32)r.kb;
    r.nblk = r.m / RED_SUM_TERMS;
    if (r.nblk < 1) r.nblk = 1;
    if (r.nblk > RED_SUM_LANES) r.nblk = RED_SUM_LANES;
    return r;
}

template <class KF>
__global__ void __launch_bounds__(256, PS_TAIL_WAVES) k_reduce_l1(const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ buckets,
                                                      u32 nseg_total, Xyzz<typename FieldTraits<KF>::Store>* __restrict__ accs,
                                                      Xyzz<typename FieldTraits<KF>::Store>* __restrict__ runs) {
    PS_TAIL_PRIO_HERE;
    u32 idx = logical_tid<KF>();
    if (idx >= nseg_total) return;
    const Xyzz<typename FieldTraits<KF>::Store>* B = buckets + (size_t)idx * RED_SEG;
    // run += B[b]; acc += run  -- written as ONE inlined addition per step (x += y with the roles
    // swapped every other step) so the kernel holds a single copy of the 14-multiplication body
    Xyzz<KF> x = xyzz_identity<KF>(), y = ld_xyzz<KF>(&B[RED_SEG - 1]);  // x = run, y = bucket
    Xyzz<KF> acc = xyzz_identity<KF>();
    for (int step = 0; step < 2 * RED_SEG; step++) {
        xyzz_add_inl<KF>(x, y);
        if ((step & 1) == 0) {  // x was run (now updated); next: acc += run
            Xyzz<KF> t = x; x = acc; y = t;
        } else {                // x was acc; next: run += next bucket
            acc = x; x = y;
            int b = RED_SEG - 2 - (step >> 1);
            if (b >= 0) y = ld_xyzz<KF>(&B[b]);
        }
    }
    st_xyzz<KF>(&accs[idx], acc);
    st_xyzz<KF>(&runs[idx], x);
}

// lvl: five arrays of sets * m points (A1, Q0, Q1, Q2, R1).  Logical thread = (role, set, j); the roles are laid out
// one after the other so that a wave runs one role.
template <class KF>
__global__ void __launch_bounds__(256, PS_TAIL_WAVES) k_reduce_pyr(const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ accs,
                                                       const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ runs,
                                                       u32 segs, u32 m, u32 sets,
                                                       Xyzz<typename FieldTraits<KF>::Store>* __restrict__ lvl) {
    const u32 per_role = sets * m;
    const u32 L = logical_tid<KF>();
    if (L >= 5u * per_role) return;
    const u32 role = L / per_role, id = L % per_role;
    const u32 set = id / m, j = id % m;
    const u32 mask = role == 1 ? 0xAAu : role == 2 ? 0xCCu : role == 3 ? 0xF0u : 0xFFu;
    const Xyzz<typename FieldTraits<KF>::Store>* src = (role == 0 ? accs : runs) + (size_t)set * segs + (size_t)j * 8;
    const u32 valid = segs < 8 ? segs : 8;
    Xyzz<KF> acc = xyzz_identity<KF>();
    for (u32 i = 0; i < valid; i++) {
        if (!((mask >> i) & 1u)) continue;
        Xyzz<KF> v = ld_xyzz<KF>(&src[i]);
        xyzz_add_inl<KF>(acc, v);
    }
    st_xyzz<KF>(&lvl[(size_t)role * per_role + id], acc);
}

// grid = sets * njobs * nblk workgroups of RED_SUM_LANES logical threads; out[(set * njobs + job) * nblk + blk]
template <class KF>
__global__ void __launch_bounds__(128) k_reduce_sum(const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ lvl, u32 m, u32 sets,
                                                    u32 njobs, u32 nblk, Xyzz<typename FieldTraits<KF>::Store>* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    typedef typename FieldTraits<KF>::Store S;
    Xyzz<S>* sm = reinterpret_cast<Xyzz<S>*>(smem_raw);
    const u32 blk = blockIdx.x % nblk, job = (blockIdx.x / nblk) % njobs, set = blockIdx.x / (nblk * njobs);
    const u32 lt = logical_local<KF>();
    const bool direct = job < (u32)RED_DIRECT_JOBS;
    const u32 arr = direct ? job : (u32)RED_DIRECT_JOBS;  // bit jobs read R1
    const Xyzz<S>* src = lvl + (size_t)arr * sets * m + (size_t)set * m;
    const u32 cnt = direct ? m : m >> 1;
    const u32 chunk = (cnt + nblk - 1) / nblk;
    const u32 k = direct ? 0u : job - (u32)RED_DIRECT_JOBS;
    const u32 low = (1u << k) - 1u;
    u32 t_end = (blk + 1) * chunk;
    if (t_end > cnt) t_end = cnt;
    Xyzz<KF> acc = xyzz_identity<KF>();
    for (u32 t = blk * chunk + lt; t < t_end; t += RED_SUM_LANES) {
        const u32 j = direct ? t : (((t & ~low) << 1) | (1u << k) | (t & low));  // the t-th j with bit k set
        Xyzz<KF> v = ld_xyzz<KF>(&src[j]);
        xyzz_add_inl<KF>(acc, v);
    }
    block_tree_sum<KF>(sm, acc);
    if (lt == 0) st_xyzz<KF>(&out[blockIdx.x], ld_xyzz<KF>(&sm[0]));
}

// one workgroup per (set, job): tree sum of the job's nblk <= RED_SUM_LANES pieces
template <class KF>
__global__ void __launch_bounds__(128) k_reduce_fin(const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ pieces, u32 nblk,
                                                    Xyzz<typename FieldTraits<KF>::Store>* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    typedef typename FieldTraits<KF>::Store S;
    Xyzz<S>* sm = reinterpret_cast<Xyzz<S>*>(smem_raw);
    const u32 lt = logical_local<KF>();
    Xyzz<KF> v = lt < nblk ? ld_xyzz<KF>(&pieces[(size_t)blockIdx.x * nblk + lt]) : xyzz_identity<KF>();
    block_tree_sum<KF>(sm, v);
    if (lt == 0) st_xyzz<KF>(&out[blockIdx.x], ld_xyzz<KF>(&sm[0]));
}

// Small bucket sets (NB <= 1024: short sums, where the chain of kernels above is pure latency -- 28 dependent additions for
// 512 buckets): sum_b (b+1) B[b] straight from the bits of v = b + 1, V_k = the sum of the buckets whose v has bit k, one
// single-wave workgroup per (set, k): NB / 128 additions in a row, then the LDS tree.  sum = sum_k 2^k V_k.
template <class KF>
__global__ void __launch_bounds__(128) k_reduce_small(const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ buckets, u32 NB, u32 c,
                                                      Xyzz<typename FieldTraits<KF>::Store>* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    typedef typename FieldTraits<KF>::Store S;
    Xyzz<S>* sm = reinterpret_cast<Xyzz<S>*>(smem_raw);
    const u32 set = blockIdx.x / c, k = blockIdx.x % c, lt = logical_local<KF>();
    const Xyzz<S>* B = buckets + (size_t)set * NB;
    Xyzz<KF> acc = xyzz_identity<KF>();
    if (k + 1 < c) {  // v in [1, NB) with bit k set: insert a 1 at position k of t
        const u32 low = (1u << k) - 1u;
        for (u32 t = lt; t < (NB >> 1); t += RED_SUM_LANES) {
            const u32 v = ((t & ~low) << 1) | (1u << k) | (t & low);
            Xyzz<KF> b = ld_xyzz<KF>(&B[v - 1]);
            xyzz_add_inl<KF>(acc, b);
        }
    } else if (lt == 0) {  // v = NB, the only value with bit c - 1
        acc = ld_xyzz<KF>(&B[NB - 1]);
    }
    block_tree_sum<KF>(sm, acc);
    if (lt == 0) st_xyzz<KF>(&out[blockIdx.x], ld_xyzz<KF>(&sm[0]));
}

// Several bucket sets (the plain plan: one per window): the weights 2^(j+2) of a set's partial results are applied here,
// one workgroup per set -- every lane doubles its own result, then an LDS tree -- because W Horner chains of c - 1
// doublings each would cost the host's scalar code more (0.4 ms at W = 16) than the 0.15 ms this takes.  With ONE set
// (window tables) the chain is 19 doublings, 50 us on the host, and the kernel is skipped.
template <class KF>
__global__ void __launch_bounds__(128) k_reduce_weights(const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ res, u32 njobs, int small,
                                                        int seg_log, Xyzz<typename FieldTraits<KF>::Store>* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    typedef typename FieldTraits<KF>::Store S;
    Xyzz<S>* sm = reinterpret_cast<Xyzz<S>*>(smem_raw);
    const u32 lt = logical_local<KF>();
    Xyzz<KF> v = lt < njobs ? ld_xyzz<KF>(&res[(size_t)blockIdx.x * njobs + lt]) : xyzz_identity<KF>();
    // results of k_reduce_small: V_j weighs 2^j; of the pyramid: A weighs 1, job j >= 1 weighs 2^seg_log * 2^(j-1)
    // (seg_log = RED_SEG_LOG; 0 for the [S, W_0, ..] results of k_qreduce_bits)
    const int shift = lt >= njobs ? 0 : small ? (int)lt : (lt == 0 ? 0 : (int)lt - 1 + seg_log);
    for (int i = 0; i < shift; i++) v = xyzz_dbl_inl<KF>(v);
    block_tree_sum<KF>(sm, v);
    if (lt == 0) st_xyzz<KF>(&out[blockIdx.x], ld_xyzz<KF>(&sm[0]));
}

// ---------------------------------------------------------------------------------------
// auxiliary kernels: format conversion, fixed-base multiplication
// ---------------------------------------------------------------------------------------
// big-endian canonical bytes -> plain little-endian limbs, reduced mod r (scalars)
__global__ void __launch_bounds__(256) k_scalars_from_be32(const uint8_t* __restrict__ in, u32 n, u32* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32* src = reinterpret_cast<const u32*>(in) + 8 * (size_t)i;
    u32 k[8];
#pragma unroll
    for (int j = 0; j < 8; j++) k[j] = __builtin_bswap32(src[7 - j]);
    // reduce: 2^256 / r < 2.3 so at most two subtractions
#pragma unroll
    for (int rep = 0; rep < 2; rep++) fe_reduce_once<FrParams>(k, 0);
#pragma unroll
    for (int j = 0; j < 8; j++) out[8 * (size_t)i + j] = k[j];
}

__global__ void __launch_bounds__(256) k_scalars_to_be32(const u32* __restrict__ in, u32 n, uint8_t* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32* dst = reinterpret_cast<u32*>(out) + 8 * (size_t)i;
#pragma unroll
    for (int j = 0; j < 8; j++) dst[7 - j] = __builtin_bswap32(in[8 * (size_t)i + j]);
}

// SetInt64 (curve.go:17-19)
__global__ void __launch_bounds__(256) k_scalars_from_i64(const int64_t* __restrict__ in, u32 n, u32* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t v = in[i];
    u64 mag = v < 0 ? (u64)0 - (u64)v : (u64)v;
    FrSat a = fe_zero<FrParams>();
    a.l[0] = (u32)mag;
    a.l[1] = (u32)(mag >> 32);
    if (v < 0) a = fe_neg<FrParams>(a);
#pragma unroll
    for (int j = 0; j < 8; j++) out[8 * (size_t)i + j] = a.l[j];
}

PS_INL Fp curve_b(const Fp*) { Fp b; constexpr i32 v[FP_L] = PS_FP28_FOUR;
#pragma unroll
    for (int i = 0; i < FP_L; i++) b.l[i] = v[i];
    return b; }
PS_INL Fp2 curve_b(const Fp2*) { Fp b = curve_b((const Fp*)0); return Fp2{b, b}; }

template <class F>
PS_HD inline bool affine_on_curve(const Affine<F>& p) {
    F lhs = f_sqr(p.y);
    F rhs = f_add(f_mul(f_sqr(p.x), p.x), curve_b((const F*)0));
    return f_eq(lhs, rhs);
}

// ZCash uncompressed bytes -> device layout; bad points counted in *nbad
__global__ void __launch_bounds__(256) k_points_from_bytes_g1(const uint8_t* __restrict__ in, u32 n,
                                                              Affine<Fp>* __restrict__ out, u32* __restrict__ nbad) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = in + 96 * (size_t)i;
    Affine<Fp> a;
    if (p[0] & 0x40) {  // the identity is 0x40 followed by zeros, nothing else
        a.x = fp_zero(); a.y = fp_zero();
        bool zero = p[0] == 0x40;
        for (int j = 1; j < 96; j++) zero = zero && p[j] == 0;
        if (!zero) atomicAdd(nbad, 1u);
    } else {
        bool ok = (p[0] & 0xE0) == 0;
        a.x = fp_from_be48(p, ok);
        a.y = fp_from_be48(p + 48, ok);
        ok = ok && affine_on_curve<Fp>(a);
        if (!ok) atomicAdd(nbad, 1u);
    }
    out[i] = a;
}
__global__ void __launch_bounds__(256) k_points_from_bytes_g2(const uint8_t* __restrict__ in, u32 n,
                                                              Affine<Fp2>* __restrict__ out, u32* __restrict__ nbad) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = in + 192 * (size_t)i;
    Affine<Fp2> a;
    if (p[0] & 0x40) {
        a.x = f_zero((const Fp2*)0); a.y = f_zero((const Fp2*)0);
        bool zero = p[0] == 0x40;
        for (int j = 1; j < 192; j++) zero = zero && p[j] == 0;
        if (!zero) atomicAdd(nbad, 1u);
    } else {
        bool ok = (p[0] & 0xE0) == 0;
        a.x.c1 = fp_from_be48(p, ok);
        a.x.c0 = fp_from_be48(p + 48, ok);
        a.y.c1 = fp_from_be48(p + 96, ok);
        a.y.c0 = fp_from_be48(p + 144, ok);
        ok = ok && affine_on_curve<Fp2>(a);
        if (!ok) atomicAdd(nbad, 1u);
    }
    out[i] = a;
}
__global__ void __launch_bounds__(256) k_points_to_bytes_g1(const Affine<Fp>* __restrict__ in, u32 n,
                                                            uint8_t* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t* p = out + 96 * (size_t)i;
    Affine<Fp> a = in[i];
    fp_to_be48(p, a.x);
    fp_to_be48(p + 48, a.y);
    if (affine_is_identity<Fp>(a)) p[0] = 0x40;
}
__global__ void __launch_bounds__(256) k_points_to_bytes_g2(const Affine<Fp2>* __restrict__ in, u32 n,
                                                            uint8_t* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t* p = out + 192 * (size_t)i;
    Affine<Fp2> a = in[i];
    fp_to_be48(p, a.x.c1);
    fp_to_be48(p + 48, a.x.c0);
    fp_to_be48(p + 96, a.y.c1);
    fp_to_be48(p + 144, a.y.c0);
    if (affine_is_identity<Fp2>(a)) p[0] = 0x40;
}

// ---- ZCash compressed encodings (what kyber's MarshalBinary emits [upstream], pinochio.go:258-272) ----
// 48 B (G1) / 96 B (G2): x big-endian (G2: c1 then c0), top bits 0x80 compressed, 0x40 infinity,
// 0x20 "y is lexicographically larger".  Returns false for a malformed encoding.
PS_HD inline bool decompress_point(Affine<Fp>& a, const uint8_t* p) {
    if (!(p[0] & 0x80)) return false;
    if (p[0] & 0x40) {
        bool zero = (p[0] & 0x3f) == 0;
        for (int i = 1; i < 48; i++) zero = zero && p[i] == 0;
        a.x = fp_zero(); a.y = fp_zero();
        return zero;
    }
    uint8_t xb[48];
    for (int i = 0; i < 48; i++) xb[i] = p[i];
    xb[0] &= 0x1f;
    bool ok = true;
    a.x = fp_from_be48(xb, ok);
    Fp rhs = f_add(f_mul(f_sqr(a.x), a.x), curve_b((const Fp*)0));
    Fp y = fp_sqrt(f_norm(rhs), ok);
    if (!ok) return false;
    y = fp_canon(y);
    if (fp_lex_larger(y) != ((p[0] & 0x20) != 0)) y = fp_canon(f_neg(y));
    a.y = y;
    return true;
}
PS_HD inline bool decompress_point(Affine<Fp2>& a, const uint8_t* p) {
    if (!(p[0] & 0x80)) return false;
    if (p[0] & 0x40) {
        bool zero = (p[0] & 0x3f) == 0;
        for (int i = 1; i < 96; i++) zero = zero && p[i] == 0;
        a.x = f_zero((const Fp2*)0); a.y = f_zero((const Fp2*)0);
        return zero;
    }
    uint8_t xb[48];
    for (int i = 0; i < 48; i++) xb[i] = p[i];
    xb[0] &= 0x1f;
    bool ok = true;
    a.x.c1 = fp_from_be48(xb, ok);
    a.x.c0 = fp_from_be48(p + 48, ok);
    Fp2 rhs = f_norm(f_add(f_mul(f_sqr(a.x), a.x), curve_b((const Fp2*)0)));
    Fp2 y = fp_sqrt(rhs, ok);
    if (!ok) return false;
    y = fp_canon(y);
    if (fp_lex_larger(y) != ((p[0] & 0x20) != 0)) y = fp_canon(f_neg(y));
    a.y = y;
    return true;
}
// affine (canonical Montgomery coordinates) -> compressed bytes
PS_HD inline void compress_point(uint8_t* out, const Affine<Fp>& a) {
    if (fp_all_zero(a.x) && fp_all_zero(a.y)) { for (int i = 0; i < 48; i++) out[i] = 0; out[0] = 0xC0; return; }
    fp_to_be48(out, a.x);
    out[0] |= 0x80 | (fp_lex_larger(a.y) ? 0x20 : 0);
}
PS_HD inline void compress_point(uint8_t* out, const Affine<Fp2>& a) {
    if (fp_all_zero(a.x) && fp_all_zero(a.y)) { for (int i = 0; i < 96; i++) out[i] = 0; out[0] = 0xC0; return; }
    fp_to_be48(out, a.x.c1);
    fp_to_be48(out + 48, a.x.c0);
    out[0] |= 0x80 | (fp_lex_larger(a.y) ? 0x20 : 0);
}

// out[stride * i ..] = ZCash compressed form of in[i] (what kyber's MarshalBinary emits)
template <class F>
__global__ void __launch_bounds__(256) k_points_compress(const Affine<F>* __restrict__ in, u32 n, u32 stride,
                                                         uint8_t* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    compress_point(out + (size_t)stride * i, in[i]);
}

template <class F>
__global__ void __launch_bounds__(256) k_points_decompress(const uint8_t* __restrict__ in, u32 n, u32 stride,
                                                           Affine<F>* __restrict__ out, u32* __restrict__ nbad) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> a;
    if (!decompress_point(a, in + (size_t)stride * i)) {
        atomicAdd(nbad, 1u);
        a.x = f_zero((const F*)0); a.y = f_zero((const F*)0);
    }
    out[i] = a;
}

// [r]P == O ?  On-curve points of E(Fp) / E'(Fp2) outside the order-r subgroup exist (cofactors
// 0x396c8c005555e1568c00aaab0000aaab and a 509-bit one); kyber's UnmarshalBinary rejects them [upstream],
// and the signed-digit folding (r - |v|)P = -(|v|P) and the Miller loop both assume order r.
template <class F>
PS_HD inline bool affine_in_subgroup(const Affine<F>& a) {
    if (fp_all_zero(a.x) && fp_all_zero(a.y)) return true;
    u32 r[8];
    for (int i = 0; i < 8; i++) r[i] = FrParams::mod(i);
    return xyzz_is_identity(xyzz_mul_scalar<F>(xyzz_from_affine<F>(a.x, a.y), r));
}
template <class F>
__global__ void __launch_bounds__(256) k_points_subgroup(const Affine<F>* __restrict__ in, u32 n, u32* __restrict__ nbad) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!affine_in_subgroup<F>(in[i])) atomicAdd(nbad, 1u);
}

PS_INL Affine<Fp> generator(const Fp*) {
    Affine<Fp> g;
    constexpr i32 x[FP_L] = PS_G1_GEN28_X; constexpr i32 y[FP_L] = PS_G1_GEN28_Y;
#pragma unroll
    for (int i = 0; i < FP_L; i++) { g.x.l[i] = x[i]; g.y.l[i] = y[i]; }
    return g;
}
PS_INL Affine<Fp2> generator(const Fp2*) {
    Affine<Fp2> g;
    constexpr i32 x0[FP_L] = PS_G2_GEN28_X0; constexpr i32 x1[FP_L] = PS_G2_GEN28_X1;
    constexpr i32 y0[FP_L] = PS_G2_GEN28_Y0; constexpr i32 y1[FP_L] = PS_G2_GEN28_Y1;
#pragma unroll
    for (int i = 0; i < FP_L; i++) { g.x.c0.l[i] = x0[i]; g.x.c1.l[i] = x1[i]; g.y.c0.l[i] = y0[i]; g.y.c1.l[i] = y1[i]; }
    return g;
}

// Fixed-base table: T[j][d] = d * 2^(8j) * G for j = 0..31, d = 0..255 (d = 0 unused).
// out[i] = sum_j T[j][byte_j(k_i)]: 32 mixed additions per scalar instead of 255 doublings +
// ~128 additions.  This is the device form of Point.Mul(s, nil) (curve.go:25-31, algebra.go:373).
template <class KF>
__global__ void __launch_bounds__(256, 1) k_fixed_base_table(Xyzz<typename FieldTraits<KF>::Store>* __restrict__ table) {
    // one logical thread per (j, d); d*2^(8j)*G by double-and-add on the generator, XYZZ out (k_batch_to_affine follows)
    typedef typename FieldTraits<KF>::Store S;
    const u32 idx = logical_tid<KF>();
    if (idx >= 32 * 256) return;
    const u32 j = idx >> 8, d = idx & 255;
    const Affine<S> gs = generator((const S*)0);
    const Affine<KF> g = ld_affine<KF>(&gs);
    Xyzz<KF> base = xyzz_from_affine<KF>(g.x, g.y);
#pragma unroll 1
    for (u32 i = 0; i < 8 * j; i++) base = xyzz_dbl_inl<KF>(base);
    Xyzz<KF> acc = xyzz_identity<KF>();
#pragma unroll 1
    for (int bit = 7; bit >= 0; bit--) {
        acc = xyzz_dbl_inl<KF>(acc);
        if ((d >> bit) & 1) xyzz_add_inl<KF>(acc, base);
    }
    st_xyzz<KF>(&table[idx], acc);
}

// Window table of a resident CRS array: T[w][i] = 2^(c w) P_i lets every window of a sum share ONE bucket set, which is
// what allows c = 20 (13 digits per 255-bit scalar instead of 16).  Row 0 is the array itself in padded rows; row w is
// c doublings of row w - 1 (XYZZ, lane pairs for G2) followed by k_batch_to_affine.
template <class F>
__global__ void __launch_bounds__(256) k_table_row0(const Affine<F>* __restrict__ pts, char* __restrict__ row, u32 row_stride, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    *reinterpret_cast<Affine<F>*>(row + (size_t)i * row_stride) = pts[i];
}
template <class KF>
__global__ void __launch_bounds__(256, 1) k_table_next(const char* __restrict__ prev, u32 prev_stride, u32 n, int c,
                                                       Xyzz<typename FieldTraits<KF>::Store>* __restrict__ out) {
    typedef typename FieldTraits<KF>::Store S;
    const u32 i = logical_tid<KF>();
    if (i >= n) return;
    Affine<KF> a = ld_affine<KF>(reinterpret_cast<const Affine<S>*>(prev + (size_t)i * prev_stride));
    Xyzz<KF> t = xyzz_identity<KF>();
    if (!affine_is_identity<KF>(a)) {
        t = xyzz_from_affine<KF>(a.x, a.y);
#pragma unroll 1
        for (int k = 0; k < c; k++) t = xyzz_dbl_inl<KF>(t);
    }
    st_xyzz<KF>(&out[i], t);
}

// ---- XYZZ -> affine for whole arrays: ONE field inversion per thread (Montgomery's trick) ----
// An inversion is a 381-bit exponentiation, ~570 products -- more than the 32 mixed additions of a fixed-base
// multiplication (320) or the 20 doublings of a table row (180) it used to follow in every thread.  Here thread t owns
// the points t, t + T, t + 2T, ...: a forward pass leaves the running product of the keys before each point in pre[],
// one inversion, a backward pass peels the inverses off: 9 products per point plus 570 / (n / T).
// The key of a point is ZZ * ZZZ for G1 and the norm d0^2 + d1^2 of d = ZZ * ZZZ for G2 (1 / d = conj(d) / norm), so
// that the chain and the inversion stay in Fp for both groups.  The identity (ZZ = 0) is skipped and stored as (0, 0).
PS_INL Fp inv_key(const Fp& d) { return d; }
PS_INL Fp inv_key(const Fp2& d) {
    const Fp d0 = f_norm(d.c0), d1 = f_norm(d.c1);
    return f_norm(f_add(fp_mul_call(d0, d0), fp_mul_call(d1, d1)));
}
PS_INL Fp inv_from_key(const Fp&, const Fp& kinv) { return kinv; }
PS_INL Fp2 inv_from_key(const Fp2& d, const Fp& kinv) {
    return Fp2{fp_mul_call(f_norm(d.c0), kinv), f_neg(fp_mul_call(f_norm(d.c1), kinv))};
}
template <class F>
__global__ void __launch_bounds__(256, 1) k_batch_to_affine(const Xyzz<F>* __restrict__ in, u32 n, u32 T, Fp* __restrict__ pre,
                                                            char* __restrict__ out, u32 out_stride) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || t >= n) return;
    Fp run = fp_one();
    u32 last = t;
#pragma unroll 1
    for (u32 i = t; i < n; i += T) {
        pre[i] = run;
        const Fp key = inv_key(f_mul(in[i].zz, in[i].zzz));
        if (!f_is_zero(key)) run = fp_mul_call(run, key);
        last = i;
    }
    Fp inv = f_inv(run);
#pragma unroll 1
    for (u32 i = last;; i -= T) {
        const Xyzz<F> p = in[i];
        Affine<F> r;
        r.x = f_zero((const F*)0);
        r.y = f_zero((const F*)0);
        const F d = f_mul(p.zz, p.zzz);
        const Fp key = inv_key(d);
        if (!f_is_zero(key)) {
            const Fp kinv = fp_mul_call(inv, pre[i]);
            inv = fp_mul_call(inv, key);
            const F dinv = inv_from_key(d, kinv);
            r.x = fp_canon(f_mul(p.x, f_mul(dinv, p.zzz)));  // stored affine coordinates are always canonical
            r.y = fp_canon(f_mul(p.y, f_mul(dinv, p.zz)));
        }
        *reinterpret_cast<Affine<F>*>(out + (size_t)i * out_stride) = r;
        if (i == t) break;
    }
}

// out[i] = a[i] + b[i] + c[i] (affine in, affine out; (0,0) is the identity)
template <class F>
__global__ void __launch_bounds__(256, 2) k_points_add3(const Affine<F>* __restrict__ a, const Affine<F>* __restrict__ b,
                                                        const Affine<F>* __restrict__ c3, u32 n, Affine<F>* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xyzz<F> acc = xyzz_identity<F>();
    const Affine<F>* src[3] = {a, b, c3};
#pragma unroll 1
    for (int k = 0; k < 3; k++) {
        Affine<F> t = src[k][i];
        if (!(f_is_zero(t.x) && f_is_zero(t.y))) xyzz_madd<F>(acc, t.x, t.y);
    }
    Affine<F> r;
    if (!xyzz_to_affine<F>(acc, r.x, r.y)) { r.x = f_zero((const F*)0); r.y = f_zero((const F*)0); }
    out[i] = r;
}

// out[i] = k_i * G in XYZZ form (k_batch_to_affine follows); lane pairs for G2
template <class KF>
__global__ void __launch_bounds__(256, 1) k_fixed_base_mul(const Affine<typename FieldTraits<KF>::Store>* __restrict__ table,
                                                           const u32* __restrict__ scalars, u32 n,
                                                           Xyzz<typename FieldTraits<KF>::Store>* __restrict__ out) {
    const u32 i = logical_tid<KF>();
    if (i >= n) return;
    Xyzz<KF> acc = xyzz_identity<KF>();
#pragma unroll 1
    for (int j = 0; j < 32; j++) {
        u32 limb = scalars[8 * (size_t)i + (j >> 2)];
        u32 d = (limb >> (8 * (j & 3))) & 255u;
        if (d) {
            Affine<KF> t = ld_affine<KF>(&table[j * 256 + d]);
            xyzz_madd<KF>(acc, t.x, t.y);
        }
    }
    st_xyzz<KF>(&out[i], acc);
}

}  // namespace ps

#include "qtail.hpp"
