// The tail of a SHORT sum (fix-up of cut buckets, bucket reduction) as shallow trees of lane-cooperative point additions.
//
// Why: below ~2^21 digits the chip is mostly idle after the accumulation and a sum's time is the DEPTH of its chain of
// dependent point additions -- 13 us each for one lane (14 field products, 5 500 multiply-adds, one wave alone on its
// SIMD).  Round 2's tail at 2^16 points: fix-up <= 7 deep, 8-bucket running sums 16, pyramid 7, sums 14: 44 additions,
// 0.66 ms of a 1.04 ms sum (profiles/r03_timeline_before_*.txt).  Two changes:
//   * ONE point addition is spread over four "coordinate lanes" (a quad; eight lanes for G2, whose Fp2 elements already
//     live on lane pairs): lane j holds coordinate j of [X, ZZ, Y, ZZZ], and add-2008-s becomes four levels of ONE field
//     product per lane instead of fourteen products in a row:
//         L1  m = [X1 ZZ2, ZZ1 X2, Y1 ZZZ2, ZZZ1 Y2] = [u1, u2, s1, s2]     d = m - swap(m) = [-p, p, -r, r]
//         L2  e = [d d, ZZ1 ZZ2, d d, ZZZ1 ZZZ2]      = [pp, zz12, rr, zzz12]
//         L3  f = [u1 pp, zz12 pp, p pp, p pp]        = [qq, ZZ3, ppp, ppp]     x3 = rr - ppp - 2 qq  (lane 2)
//         L4  g = [-, -, (-r)(x3 - qq) - s1 ppp, zzz12 ppp] = [-, -, Y3, ZZZ3]
//     1 764 multiply-adds deep instead of 5 488 (4.5 us), operands moved inside the quad (DPP for G1).
//   * the chains become trees: cut buckets are summed by LPB quads each (k_qfixup), and  sum_b (b+1) B[b]  with
//     b = hi 2^s + lo  is  sum_hi (2^s hi + 1) R_hi + sum_lo lo C_lo  over the row sums R_hi and column sums C_lo of the
//     buckets laid out as a 2^(cb-s) x 2^s matrix (k_qreduce_rowcol: a tree per row and per column), and the two short
//     weighted sums are split by the bits of hi and lo (k_qreduce_bits: a tree per bit).  The host applies the weights
//     2^k (a Horner chain of cb doublings, ~10 us).  Depth at 2^16 points (32 768 buckets): 8 + 8 quad additions.
// Same group elements as the chains (the group law is exact), hence the same bytes.
#pragma once

namespace ps {

template <class KF> struct QTraits {
    static constexpr int S = FieldTraits<KF>::LANES;  // lanes per coordinate (2: an Fp2 component pair)
    static constexpr int GL = 4 * S;                  // lanes per point
};
PS_INL Fp& fp_of(Fp& a) { return a; }
PS_INL const Fp& fp_of(const Fp& a) { return a; }
PS_INL Fp& fp_of(Fp2s& a) { return a.v; }
PS_INL const Fp& fp_of(const Fp2s& a) { return a.v; }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline i32 lane_read(i32 v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
#else
PS_HD inline i32 lane_read(i32 v, int) { return v; }
#endif

template <class KF> __device__ inline int q_coord() { return (int)((threadIdx.x / QTraits<KF>::S) & 3u); }

// my copy of the value held by coordinate lane `src` (0..3, may differ from lane to lane) of my quad
template <class KF>
__device__ inline KF q_get(const KF& v, int src) {
    constexpr int S = QTraits<KF>::S;
    const int lane = (int)(threadIdx.x & 63u);
    const int from = (lane & ~(4 * S - 1)) | (src * S) | (lane & (S - 1));
    KF r;
#pragma unroll
    for (int i = 0; i < FP_L; i++) fp_of(r).l[i] = lane_read(fp_of(v).l[i], from);
    return r;
}
// G1: the quad is a DPP quad, every move is a quad_perm (one full-rate v_mov_dpp per limb, no LDS crossbar)
template <int CTRL>
__device__ inline Fp q_perm(const Fp& v) {
    Fp r;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < FP_L; i++) r.l[i] = __builtin_amdgcn_mov_dpp(v.l[i], CTRL, 0xF, 0xF, false);
#else
    r = v;
#endif
    return r;
}
// the partner coordinate of the pairs (X, ZZ), (Y, ZZZ): [1, 0, 3, 2]
__device__ inline Fp q_swap(const Fp& v) { return q_perm<0xB1>(v); }
__device__ inline Fp2s q_swap(const Fp2s& v) { return Fp2s{q_perm<0x4E>(v.v)}; }  // lane pairs: quad_perm [2,3,0,1] swaps the pairs of a 4-lane half
// broadcast of coordinate lane K
template <int K> __device__ inline Fp q_bcast(const Fp& v) { return q_perm<K * 0x55>(v); }
template <int K> __device__ inline Fp2s q_bcast(const Fp2s& v) { return q_get<Fp2s>(v, K); }

// flag of coordinate lane `src`, the same on every lane of the quad
template <class KF>
__device__ inline bool q_flag(bool v, int src) {
    constexpr int S = QTraits<KF>::S;
    const unsigned long long b = __ballot(v);
    const int lane = (int)(threadIdx.x & 63u);
    return (b >> ((lane & ~(4 * S - 1)) | (src * S))) & 1ull;
}

template <class KF>
__device__ inline KF q_load(const Xyzz<typename FieldTraits<KF>::Store>* p) {  // my coordinate of *p
    typedef typename FieldTraits<KF>::Store St;
    const int j = q_coord<KF>();
    const int mi = j == 0 ? 0 : j == 1 ? 2 : j == 2 ? 1 : 3;  // struct order x, y, zz, zzz
    return ld_f(reinterpret_cast<const St*>(p) + mi, (const KF*)0);
}
template <class KF>
__device__ inline void q_store(Xyzz<typename FieldTraits<KF>::Store>* p, const KF& v) {
    typedef typename FieldTraits<KF>::Store St;
    const int j = q_coord<KF>();
    const int mi = j == 0 ? 0 : j == 1 ? 2 : j == 2 ? 1 : 3;
    st_f(reinterpret_cast<St*>(p) + mi, v);
}
template <class KF>
__device__ inline KF q_select(const KF& c0, const KF& c1, const KF& c2, const KF& c3, int j) {
    KF r;
#pragma unroll
    for (int i = 0; i < FP_L; i++)
        fp_of(r).l[i] = j == 0 ? fp_of(c0).l[i] : j == 1 ? fp_of(c1).l[i] : j == 2 ? fp_of(c2).l[i] : fp_of(c3).l[i];
    return r;
}

// P + P for a point in quad form (the rare branch of q_add): gathered into every lane, doubled out of line
template <class KF>
__device__ __attribute__((noinline)) void q_dbl_cold(KF& a) {
    Xyzz<KF> A;
    A.x = q_get<KF>(a, 0); A.zz = q_get<KF>(a, 1); A.y = q_get<KF>(a, 2); A.zzz = q_get<KF>(a, 3);
    A = xyzz_dbl<KF>(A);
    a = q_select<KF>(A.x, A.zz, A.y, A.zzz, q_coord<KF>());
}

// acc += q, both in quad form (a, b: my coordinate).  Control flow is uniform inside a quad.  Limb classes: stored
// coordinates are class ~1 (products, or f_norm of a sum of <= 4 fresh products); d is a difference of two products
// (|limb| < 2^28, |V| < 1.25 p); x3 - qq is class 2, |V| < 5.7 p: every product sees class(a) class(b) <= 4.
template <class KF>
__device__ inline __attribute__((always_inline)) void q_add(KF& a, const KF& b) {
    const int j = q_coord<KF>();
    const bool bz = q_flag<KF>(f_is_zero(b), 1), az = q_flag<KF>(f_is_zero(a), 1);  // ZZ == 0: the identity
    if (bz) return;
    if (az) { a = b; return; }
    const KF m = f_mul_ilp(a, q_swap(b));              // [u1, u2, s1, s2]
    const KF d = f_sub(m, q_swap(m));              // [-p, p, -r, r]
    const bool dz = f_is_zero(d);
    if (q_flag<KF>(dz, 1)) {                       // same x: P + P or P + (-P)
        if (q_flag<KF>(dz, 3)) q_dbl_cold<KF>(a);
        else a = f_zero((const KF*)0);
        return;
    }
    const bool odd = (j & 1) != 0;
    KF oa, ob;
#pragma unroll
    for (int i = 0; i < FP_L; i++) {
        fp_of(oa).l[i] = odd ? fp_of(a).l[i] : fp_of(d).l[i];
        fp_of(ob).l[i] = odd ? fp_of(b).l[i] : fp_of(d).l[i];
    }
    const KF e = f_mul_ilp(oa, ob);                    // [pp, zz12, rr, zzz12]
    const KF PP = q_bcast<0>(e), Pd = q_bcast<1>(d);
#pragma unroll
    for (int i = 0; i < FP_L; i++) {
        fp_of(oa).l[i] = j == 0 ? fp_of(m).l[i] : j == 1 ? fp_of(e).l[i] : fp_of(Pd).l[i];
        fp_of(ob).l[i] = j == 0 ? fp_of(e).l[i] : fp_of(PP).l[i];
    }
    const KF f = f_mul_ilp(oa, ob);                    // [qq, ZZ3, ppp, ppp]
    const KF QQ = q_bcast<0>(f);
    const KF x3 = f_norm(f_sub(f_sub(f_sub(e, f), QQ), QQ));  // lane 2: rr - ppp - 2 qq
    const KF x3q = f_sub(x3, QQ);
    KF oc, od;
#pragma unroll
    for (int i = 0; i < FP_L; i++) {
        fp_of(oa).l[i] = j == 2 ? fp_of(d).l[i] : fp_of(e).l[i];
        fp_of(ob).l[i] = j == 2 ? fp_of(x3q).l[i] : fp_of(f).l[i];
        fp_of(oc).l[i] = j == 2 ? fp_of(m).l[i] : 0;
        fp_of(od).l[i] = j == 2 ? fp_of(f).l[i] : 0;
    }
    const KF g = f_mul2sub_ilp(oa, ob, oc, od);        // lane 2: (-r)(x3 - qq) - s1 ppp = Y3; lane 3: zzz12 ppp = ZZZ3
    const KF X3 = q_bcast<2>(x3);
    a = q_select<KF>(X3, f, f_norm(g), g, j);
}

// Tree sum over groups of `grp` consecutive points of the block (grp a power of two); result on the group's first
// point.  sm: one Fp per thread.  Every thread of the block must call it.
template <class KF>
__device__ inline void q_block_tree(Fp* sm, KF& mine, u32 pt, u32 grp) {
    constexpr u32 GL = QTraits<KF>::GL;
    const u32 sub = pt & (grp - 1);
    for (u32 stride = grp >> 1; stride > 0; stride >>= 1) {
        if (sub >= stride && sub < 2 * stride) sm[threadIdx.x] = fp_of(mine);
        __syncthreads();
        if (sub < stride) {
            KF o;
            fp_of(o) = sm[threadIdx.x + stride * GL];
            q_add<KF>(mine, o);
        }
    }
}

// ---- fix-up of the buckets cut by slice boundaries: LPB quads per bucket ----
// Quad `sub` of bucket g sums the partial slots of the slices t0 + sub, t0 + sub + LPB, ..; a tree over the LPB quads
// follows.  Buckets that span more than QFIX_HEAVY x LPB slices go to the heavy-bucket kernels (msm.hpp section 5).
constexpr u32 QFIX_HEAVY = 16;  // (4: buckets a little above the average went heavy by the thousand -- Groth16 on 2^14 constraints 1.7 -> 2.1 ms)
template <class KF>
__global__ void __launch_bounds__(256) k_qfixup(const u32* __restrict__ offs, u32 G, int Mplan, u32 T, u32 LPB,
                                                const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ parts,
                                                Xyzz<typename FieldTraits<KF>::Store>* __restrict__ buckets,
                                                u32* __restrict__ heavy_count, u32* __restrict__ heavy_list) {
    PS_TAIL_PRIO_HERE;
    constexpr u32 GL = QTraits<KF>::GL;
    __shared__ Fp sm[256];
    const int M = eff_slice(offs[G], T, Mplan);
    const u32 pt = threadIdx.x / GL;
    const u64 gp = (u64)blockIdx.x * (256 / GL) + pt;
    const u32 g = (u32)(gp / LPB), sub = (u32)(gp % LPB);
    const bool valid = g < G;
    KF acc = f_zero((const KF*)0);
    bool store = false;
    if (valid) {
        const u32 lo = offs[g], hi = offs[g + 1];
        if (lo == hi) {
            store = true;  // an empty bucket gets its identity here (the bucket array is never cleared)
        } else {
            const u32 t0 = lo / (u32)M, t1 = (hi - 1) / (u32)M;
            if (t0 != t1) {
                if (t1 - t0 >= QFIX_HEAVY * LPB) {
                    if (sub == 0 && (threadIdx.x & (GL - 1)) == 0) heavy_list[atomicAdd(heavy_count, 1u)] = g;
                } else {
                    store = true;
                    for (u32 t = t0 + sub; t <= t1; t += LPB) {
                        const u32 slice_start = t * (u32)M;
                        const u32 rs = lo > slice_start ? lo : slice_start;
                        const KF p = q_load<KF>(&parts[2 * (size_t)t + (rs == slice_start ? 0 : 1)]);
                        q_add<KF>(acc, p);
                    }
                }
            }
        }
    }
    q_block_tree<KF>(sm, acc, pt, LPB);
    if (store && sub == 0) q_store<KF>(&buckets[g], acc);
}

// ---- heavy buckets (a witness that is half ones puts n/2 digits into one bucket) ----
// Two levels of quad trees: jobs of a block's worth of slices (64 G1 / 32 G2 quads; s partial sums per quad for the largest
// buckets, heavy_chunk_of), a tree as deep as the job is long, then per bucket the same over its jobs' results.  512 partial sums of a G2 bucket (Groth16's B on 2^12
// constraints of the tiled toy circuit): 5 + 4 quad additions deep, 0.10 ms, where the one-lane kernels of the long sums
// (k_fixup_heavy_part: strided chains, then a 7-level tree of 20 us lane-pair additions) took 0.31 ms.
template <class KF>
__device__ inline u32 q_tree_size(u32 count) {  // the power of two >= count, at most the block's quads
    constexpr u32 NPB = 256 / QTraits<KF>::GL;
    u32 g = 1;
    while (g < count && g < NPB) g <<= 1;
    return g;
}
template <class KF>
__global__ void __launch_bounds__(256) k_qfixup_heavy_part(const u32* __restrict__ offs, u32 G, int Mplan, u32 T,
                                                            const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ parts,
                                                            const u32* __restrict__ heavy_count, const u32* __restrict__ heavy_list,
                                                            const u32* __restrict__ job_base,
                                                            Xyzz<typename FieldTraits<KF>::Store>* __restrict__ hparts) {
    PS_TAIL_PRIO_HERE;
    constexpr u32 GL = QTraits<KF>::GL, NPB = 256 / GL;
    __shared__ Fp sm[256];
    const u32 nheavy = *heavy_count;
    if (nheavy == 0) return;
    const int M = eff_slice(offs[G], T, Mplan);
    const u32 njobs = job_base[nheavy];
    const u32 pt = threadIdx.x / GL;
    for (u32 j = blockIdx.x; j < njobs; j += gridDim.x) {
        u32 lo_h = 0, hi_h = nheavy;  // job_base[lo_h] <= j < job_base[hi_h]
        while (hi_h - lo_h > 1) {
            const u32 mid = (lo_h + hi_h) >> 1;
            if (job_base[mid] <= j) lo_h = mid; else hi_h = mid;
        }
        const u32 g = heavy_list[lo_h];
        const u32 lo = offs[g], hi = offs[g + 1];
        const u32 t0 = lo / (u32)M, t1 = (hi - 1) / (u32)M;
        const u32 chunk = heavy_chunk_of(t1 - t0 + 1, NPB);  // as k_heavy_jobs cut the bucket
        const u32 ts = t0 + (j - job_base[lo_h]) * chunk;
        const u32 te = (t1 - ts >= chunk) ? ts + chunk - 1 : t1;
        KF acc = f_zero((const KF*)0);
        for (u32 t = ts + pt; t <= te; t += NPB) {
            const u32 slice_start = t * (u32)M;
            const u32 rs = lo > slice_start ? lo : slice_start;
            const KF p = q_load<KF>(&parts[2 * (size_t)t + (rs == slice_start ? 0 : 1)]);
            q_add<KF>(acc, p);
        }
        q_block_tree<KF>(sm, acc, pt, q_tree_size<KF>(te - ts + 1));
        if (pt == 0) q_store<KF>(&hparts[j], acc);
        __syncthreads();
    }
}
template <class KF>
__global__ void __launch_bounds__(256) k_qfixup_heavy(const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ hparts,
                                                       Xyzz<typename FieldTraits<KF>::Store>* __restrict__ buckets,
                                                       const u32* __restrict__ heavy_count, const u32* __restrict__ heavy_list,
                                                       const u32* __restrict__ job_base) {
    PS_TAIL_PRIO_HERE;
    constexpr u32 GL = QTraits<KF>::GL, NPB = 256 / GL;
    __shared__ Fp sm[256];
    const u32 nheavy = *heavy_count;
    const u32 pt = threadIdx.x / GL;
    for (u32 h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const u32 j0 = job_base[h], j1 = job_base[h + 1];
        KF acc = f_zero((const KF*)0);
        for (u32 j = j0 + pt; j < j1; j += NPB) { const KF p = q_load<KF>(&hparts[j]); q_add<KF>(acc, p); }
        q_block_tree<KF>(sm, acc, pt, q_tree_size<KF>(j1 - j0));
        if (pt == 0) q_store<KF>(&buckets[heavy_list[h]], acc);
        __syncthreads();
    }
}

// ---- reduction, first level: row sums R[set][hi] and column sums C[set][lo] of the 2^(cb-s) x 2^s bucket matrix ----
// One block per row / column; each quad takes every NP-th term, then the tree.
template <class KF>
// `extra` (optional): a second array of the same shape whose ROW sums go to R2 (jobs rows + cols .. 2 rows + cols - 1): the
// long sums' hybrid tail reduces the segments' `run` sums by rows and columns and needs the plain total of their `acc` sums.
__global__ void __launch_bounds__(512) k_qreduce_rowcol(const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ buckets, int cb, int s,
                                                         Xyzz<typename FieldTraits<KF>::Store>* __restrict__ R,
                                                         Xyzz<typename FieldTraits<KF>::Store>* __restrict__ C,
                                                         const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ extra,
                                                         Xyzz<typename FieldTraits<KF>::Store>* __restrict__ R2) {
    PS_TAIL_PRIO_HERE;
    constexpr u32 GL = QTraits<KF>::GL;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Fp* sm = reinterpret_cast<Fp*>(smem_raw);
    const u32 rows = 1u << (cb - s), cols = 1u << s, jobs = rows + cols + (extra ? rows : 0u);
    const u32 set = blockIdx.x / jobs, job = blockIdx.x % jobs;
    const bool second = job >= rows + cols;
    const Xyzz<typename FieldTraits<KF>::Store>* B = (second ? extra : buckets) + ((size_t)set << cb);
    const u32 pt = threadIdx.x / GL, NP = blockDim.x / GL;
    KF acc = f_zero((const KF*)0);
    if (job < rows || second) {
        const u32 row = second ? job - rows - cols : job;
        for (u32 i = pt; i < cols; i += NP) { const KF p = q_load<KF>(&B[(size_t)row * cols + i]); q_add<KF>(acc, p); }
    } else {
        const u32 lo = job - rows;
        for (u32 i = pt; i < rows; i += NP) { const KF p = q_load<KF>(&B[(size_t)i * cols + lo]); q_add<KF>(acc, p); }
    }
    q_block_tree<KF>(sm, acc, pt, NP);
    if (pt == 0)
        q_store<KF>(second ? &R2[(size_t)set * rows + (job - rows - cols)] : job < rows ? &R[(size_t)set * rows + job] : &C[(size_t)set * cols + (job - rows)], acc);
}

// ---- reduction, second level: per set cb + 1 results  [S, W_0 .. W_{cb-1}],  set sum = S + sum_k 2^k W_k ----
//   S   = sum_hi R_hi                                        (the "+1" of b + 1)
//   W_k = sum of the C_lo whose lo has bit k          k <  s
//       = sum of the R_hi whose hi has bit k - s      k >= s
// With s = 0 (R = the buckets themselves, no C) this is the whole reduction of a small bucket set in one kernel.
template <class KF>
__global__ void __launch_bounds__(512) k_qreduce_bits(const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ R,
                                                       const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ C, int cb, int s,
                                                       Xyzz<typename FieldTraits<KF>::Store>* __restrict__ out,
                                                       const u32* __restrict__ entries_src, u32* __restrict__ entries_dst,
                                                       const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ R0) {
    PS_TAIL_PRIO_HERE;
    constexpr u32 GL = QTraits<KF>::GL;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Fp* sm = reinterpret_cast<Fp*>(smem_raw);
    const u32 rows = 1u << (cb - s), cols = 1u << s, jobs = (u32)cb + 1u;
    const u32 set = blockIdx.x / jobs, job = blockIdx.x % jobs;
    const u32 pt = threadIdx.x / GL, NP = blockDim.x / GL;
    if (blockIdx.x == 0 && threadIdx.x == 0 && entries_dst) *entries_dst = *entries_src;  // the entry count rides along
    KF acc = f_zero((const KF*)0);
    if (job == 0) {  // R0 (optional): the row sums whose total is job 0's result instead of R's (k_qreduce_rowcol's `extra`)
        const Xyzz<typename FieldTraits<KF>::Store>* src0 = (R0 ? R0 : R) + (size_t)set * rows;
        for (u32 i = pt; i < rows; i += NP) { const KF p = q_load<KF>(&src0[i]); q_add<KF>(acc, p); }
    } else {
        const u32 k = job - 1;
        const bool fromC = k < (u32)s;
        const u32 kk = fromC ? k : k - (u32)s, cnt = (fromC ? cols : rows) >> 1, low = (1u << kk) - 1u;
        const Xyzz<typename FieldTraits<KF>::Store>* src = fromC ? C + (size_t)set * cols : R + (size_t)set * rows;
        for (u32 t = pt; t < cnt; t += NP) {
            const u32 idx = ((t & ~low) << 1) | (1u << kk) | (t & low);  // the t-th index with bit kk set
            const KF p = q_load<KF>(&src[idx]);
            q_add<KF>(acc, p);
        }
    }
    q_block_tree<KF>(sm, acc, pt, NP);
    if (pt == 0) q_store<KF>(&out[blockIdx.x], acc);
}

}  // namespace ps
