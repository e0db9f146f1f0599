// C ABI of libplaysnark_hip.so (include/playsnark_hip.h).  Product code: no CPU fallback --
// every compute entry point runs the HIP kernels or fails with PS_ERR_NO_DEVICE / PS_ERR_HIP.
#include "../../include/playsnark_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <future>
#include <string>
#include <vector>

#include "msm.hpp"
#include "quotient.hpp"
#include "lagrange.hpp"

namespace ps {
#include "hostfield.inc"
}
using namespace ps;

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(PS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
    } while (0)

extern "C" const char* ps_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* ps_version(void) { return "playsnark_hip 0.4 (gfx950), ABI 4"; }
extern "C" int ps_abi_version(void) { return PS_ABI_VERSION; }
extern "C" int ps_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return PS_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(PS_ERR_HIP, std::string("hipMalloc workspace: ") + hipGetErrorString(e));
        cap = want;
        return PS_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

constexpr size_t PS_PINNED_SLOT = 64 * sizeof(Xyzz<Fp2>) + 16;  // what the host folds: window sums (W <= 64) or one set's partial results (<= 21), then the entry count of a short sum

struct ps_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // The latency-bound tail of a sum (fix-up, reduction, copy of the window sums) runs on its own
    // high-priority stream: when it shares the chip with another sum's accumulation, its few
    // workgroups are dispatched ahead of the thousands the accumulation has queued.
    hipStream_t tail = nullptr;
    hipEvent_t ev_acc_local = nullptr, ev_tail_done = nullptr, ev_sort_local = nullptr;
    bool tail_used = false;
    // MSM workspace
    DevBuf counts, offs, bsum, keys, ranks, vals, sorted, buckets, parts, segs, wins, heavy, hparts, coarse;
    DevBuf affine_tmp;  // XYZZ points + chain products of k_batch_to_affine (fixed-base multiplications, window tables)
    DevBuf staging;                  // byte staging for uploads / downloads
    // ps_msm_be32 / ps_msm_i64 (seam S1: one upload per BlindEval call): the converted scalars of the call live in a vector
    // the context keeps, so a call does not pay a hipMalloc and a hipFree (which synchronises the device) of 32 bytes per
    // scalar on top of its copy over PCIe
    struct ps_scalars* up_scalars = nullptr;
    // the provers' own scalar vectors (Groth16: SA, SB, SC; PHGR13: h), kept between proofs like up_scalars: a hipMalloc and a
    // hipFree of 32 MB each per proof were ~1 ms of host time at 2^20 constraints, the free synchronising the device
    struct ps_scalars* pv_scalars[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t pv_cap[4] = {0, 0, 0, 0};
    size_t up_cap = 0;
    DevBuf fb_table[2];              // fixed-base tables (G1, G2)
    bool fb_ready[2] = {false, false};
    u32* d_flag = nullptr;           // small device scratch word (bad-point counter etc.)
    void* h_pinned = nullptr;        // pinned host scratch for window sums
    size_t h_pinned_cap = 0;
    // asynchronous sums: a FIFO of at most PS_MSM_QUEUE (ps_msm_launch, ..., then ps_msm_finish pops
    // the oldest).  Each pending sum has its own workspace (the context itself, then `pipe`, `pipe2`:
    // own streams + buffers); accumulations are chained in launch order, so a later sum's sort and
    // accumulation run beside the earlier ones' latency-bound fix-up / reduction / host fold.
    struct PendingMsm { int group; MsmPlan plan; ps_ctx* wc; };
    PendingMsm q[PS_MSM_QUEUE];
    int q_head = 0, q_len = 0;
    bool pending = false;            // q_len > 0
    ps_ctx *pipe = nullptr, *pipe2 = nullptr, *pipe3 = nullptr, *pipe4 = nullptr, *pipe5 = nullptr;
    hipEvent_t ev_fork = nullptr;
    ps_ctx* last_chain = nullptr;    // workspace of the sum launched last (its ev_acc_local = accumulation done)
    ps_ctx* last_timed = nullptr;
    float phase_ms[PS_PROVE_PHASES] = {0, 0, 0, 0};  // host wall clock of the last prover call
    bool use_tables = true;          // the provers build window tables for their CRS arrays (ps_ctx_set_tables)
    long long table_budget = -1;     // bytes a prover may spend on ONE window table; negative: what hipMemGetInfo leaves (ps_ctx_set_table_budget)
    void* d_small = nullptr;         // 256 B of device scratch for the provers' few loose scalars (r, s, rs, 1, 0)
    hipEvent_t ev_q = nullptr;       // ordering between the context stream and the quotient's high-priority stream
    ps_msm_info last_info{};
    int forced_c = 0;
    int forced_slice = 0;
    int forced_tail = 0;             // ps_msm_set_tail: 0 automatic, 1 chains, 2 trees of lane-cooperative additions
    QuotientCache* qcache = nullptr;
    // Groth16 fused driver: secondary context (own stream + workspace) for the concurrent G2 MSM,
    // and the concatenated CRS arrays of the last proving key
    ps_ctx* aux = nullptr;
    unsigned long long g16_key[4] = {0, 0, 0, 0};
    uint8_t g16_fixed[96 * 3 + 192 * 2] = {0};
    ps_points *g16_pa = nullptr, *g16_pb = nullptr, *g16_pc = nullptr;
    bool g16_tabs = false;           // whether window tables were wanted when the arrays above were made
    bool g16_split = false;          // ... and whether PC was made for the split form of C (prove.inc, groth16_prove_impl)
    size_t g16_b1_min_n = (size_t)1 << 19;  // Lagrange-form keys of this many constraints or more: B in G1 as a sum of its own (PS_G16_B1_MIN_N)
    hipEvent_t g16_ready = nullptr;
    // PHGR13 driver: vbs + wbs + ybs summed pointwise once per evaluation key (gz, pinochio.go:239-242)
    unsigned long long phgr_key[3] = {0, 0, 0};
    ps_points* phgr_bsum = nullptr;
    // optional per-stage timing (HIP events on `stream`, the stream the kernels run on)
    bool timing = false;
    hipEvent_t ev[PS_MSM_STAGES + 1] = {};
    bool ev_valid = false;
    hipEvent_t ev_multi[PS_MSM_MULTI_MAX] = {};  // one per result of ps_msm_multi (created on first use)
    hipEvent_t ev_acc[PS_MSM_MULTI_MAX] = {};    // ... and one per accumulation kernel
    hipEvent_t ev_sorted = nullptr;
};

struct Storage {  // shared device allocation behind slices
    void* p = nullptr;
    std::atomic<int> refs{1};  // handles (slices) on it may be made and freed by several host threads
    // Recorded by a producer that returns before its kernel has run (ps_points_from_scalars,
    // ps_scalars_from_device_be32); every stream that consumes the array waits for it first.
    hipEvent_t ready = nullptr;
    // ps_points_precompute: window table T[w][i] = 2^(table_c * w) * P[i], w < table_W, rows of table_stride points.
    // A key shared by several contexts (one per host thread) gets its table from whichever prover comes first: built
    // under `mu`, complete in memory before `table` is published (release), so a reader that sees the pointer (acquire)
    // sees the geometry and may gather from it without an event.  Releasing / rebuilding with another window size is
    // not synchronised with sums in flight on OTHER contexts: callers do that while no other thread uses the array.
    std::mutex mu;
    std::atomic<void*> table{nullptr};
    int table_c = 0, table_W = 0;
    size_t table_stride = 0;
    // A prover found no room for the table (points_ensure_table).  The mark is not for ever: it records what the asking context
    // had to offer -- its budget and the free device memory -- and a later request that has more of either asks again (a
    // key is shared between contexts: one with a small ps_ctx_set_table_budget must not decide for all).
    bool table_declined = false;
    long long declined_budget = 0;  // negative: automatic
    size_t declined_free = 0;
    size_t count = 0;  // elements in the allocation (point arrays)
    // Window tables of index ranges of the allocation (ViewTable below): an index-range shard of a key whose table was built
    // for the WHOLE array would reduce that table's 2^(c-1) buckets for an eighth of the points; the first sum over such
    // a view builds a table of the view's own (window chosen for ITS length) under `mu`.  Slots are published with release
    // and never change afterwards (until ps_points_precompute(p, -1) / the last handle goes), readers scan them without the lock.
    static constexpr int VIEW_SLOTS = 64;
    std::atomic<struct ViewTable*> vtab[VIEW_SLOTS] = {};
};
struct ViewTable {
    size_t first, n;  // the index range [first, first + n) of the allocation
    int c, W;         // window bits, rows
    std::atomic<void*> table;  // W rows of n points (row stride n); nullptr: no room when it was asked for (declined_* as for the whole
                               // table) -- a declined record becomes the table when a later request has room: published last, with release
    long long declined_budget;
    size_t declined_free;
};
static void storage_free_view_tables(Storage* s) {
    for (auto& slot : s->vtab) {
        ViewTable* v = slot.exchange(nullptr);
        if (v) { if (void* t = v->table.load()) (void)hipFree(t); delete v; }
    }
}
static void storage_unref(Storage* s) {
    if (s && s->refs.fetch_sub(1) == 1) {
        if (s->ready) (void)hipEventDestroy(s->ready);
        if (s->table.load()) (void)hipFree(s->table.load());
        storage_free_view_tables(s);
        if (s->p) (void)hipFree(s->p);
        delete s;
    }
}
static int storage_mark_ready(Storage* s, hipStream_t producer) {
    if (!s->ready) {
        hipError_t e = hipEventCreateWithFlags(&s->ready, hipEventDisableTiming);
        if (e != hipSuccess) { s->ready = nullptr; return PS_ERR_HIP; }
    }
    return hipEventRecord(s->ready, producer) == hipSuccess ? PS_OK : PS_ERR_HIP;
}
static int storage_wait_ready(const Storage* s, hipStream_t consumer) {
    if (!s->ready) return PS_OK;
    return hipStreamWaitEvent(consumer, s->ready, 0) == hipSuccess ? PS_OK : PS_ERR_HIP;
}

static std::atomic<unsigned long long> g_points_uid{0};
struct ps_points {
    int group;
    size_t n;
    Storage* st;
    size_t first;  // element offset into the storage
    int device;
    unsigned long long uid = ++g_points_uid;  // identity for caches of derived arrays
};
struct ps_scalars {
    size_t n;
    Storage* st;
    size_t first;
    int max_bits;  // upper bound on the bit length of every element (of |v| when neg_small)
    int device;
    bool neg_small = false;  // int64 witness vector with negative values: elements are v or r - |v|, |v| < 2^64
};

static inline size_t point_bytes(int group) { return group == PS_G1 ? sizeof(Affine<Fp>) : sizeof(Affine<Fp2>); }
static inline size_t wire_bytes(int group) { return group == PS_G1 ? 96 : 192; }
static inline const void* points_ptr(const ps_points* p) {
    return (const char*)p->st->p + p->first * point_bytes(p->group);
}
static inline const u32* scalars_ptr(const ps_scalars* s) { return (const u32*)s->st->p + 8 * s->first; }

static inline unsigned nblocks(size_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
extern "C" int ps_ctx_create(int device, ps_ctx** out) {
    if (!out) return fail(PS_ERR_ARG, "ps_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(PS_ERR_NO_DEVICE, "no HIP device visible: playsnark_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(PS_ERR_ARG, "ps_ctx_create: bad device index");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(PS_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", library is built for gfx950 only");
    ps_ctx* c = new ps_ctx();
    if (const char* e = getenv("PS_G16_B1_MIN_N")) c->g16_b1_min_n = (size_t)strtoull(e, nullptr, 10);  // tests / measurements
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&c->tail, hipStreamNonBlocking, greatest));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_acc_local, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_tail_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_sort_local, hipEventDisableTiming));
    }
    HIP_TRY(hipMalloc((void**)&c->d_flag, 64));
    c->h_pinned_cap = PS_MSM_MULTI_MAX * PS_PINNED_SLOT + 64;
    HIP_TRY(hipHostMalloc(&c->h_pinned, c->h_pinned_cap));
    for (auto& e : c->ev) HIP_TRY(hipEventCreate(&e));
    *out = c;
    return PS_OK;
}

extern "C" void ps_ctx_destroy(ps_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->tail) (void)hipStreamSynchronize(c->tail);
    for (DevBuf* b : {&c->counts, &c->offs, &c->bsum, &c->keys, &c->ranks, &c->sorted, &c->buckets, &c->parts,
                      &c->segs, &c->wins, &c->heavy, &c->hparts, &c->vals, &c->coarse, &c->staging, &c->affine_tmp, &c->fb_table[0], &c->fb_table[1]})
        b->release();
    quotient_cache_free(c->qcache);
    if (c->up_scalars) ps_scalars_free(c->up_scalars);
    for (ps_scalars* v : c->pv_scalars)
        if (v) ps_scalars_free(v);
    if (c->g16_pa) ps_points_free(c->g16_pa);
    if (c->g16_pb) ps_points_free(c->g16_pb);
    if (c->g16_pc) ps_points_free(c->g16_pc);
    if (c->g16_ready) (void)hipEventDestroy(c->g16_ready);
    if (c->phgr_bsum) ps_points_free(c->phgr_bsum);
    if (c->aux) ps_ctx_destroy(c->aux);
    if (c->pipe) ps_ctx_destroy(c->pipe);
    if (c->pipe2) ps_ctx_destroy(c->pipe2);
    if (c->pipe3) ps_ctx_destroy(c->pipe3);
    if (c->pipe4) ps_ctx_destroy(c->pipe4);
    if (c->pipe5) ps_ctx_destroy(c->pipe5);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_multi) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_acc) if (e) (void)hipEventDestroy(e);
    if (c->ev_sorted) (void)hipEventDestroy(c->ev_sorted);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->ev_q) (void)hipEventDestroy(c->ev_q);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->ev_acc_local) (void)hipEventDestroy(c->ev_acc_local);
    if (c->ev_tail_done) (void)hipEventDestroy(c->ev_tail_done);
    if (c->ev_sort_local) (void)hipEventDestroy(c->ev_sort_local);
    if (c->tail) (void)hipStreamDestroy(c->tail);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int ps_ctx_sync(ps_ctx* c) {
    if (!c) return fail(PS_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->tail));
    return PS_OK;
}
extern "C" void* ps_ctx_stream(ps_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ---------------------------------------------------------------------------------------
// scalars
// ---------------------------------------------------------------------------------------
static int scalars_alloc(ps_ctx* c, size_t n, ps_scalars** out) {
    Storage* st = new Storage();
    hipError_t e = hipMalloc(&st->p, std::max<size_t>(32 * n, 32));
    if (e != hipSuccess) {
        delete st;
        return fail(PS_ERR_HIP, std::string("hipMalloc scalars: ") + hipGetErrorString(e));
    }
    *out = new ps_scalars{n, st, 0, 255, c->device};
    return PS_OK;
}

extern "C" int ps_scalars_from_device_be32(ps_ctx* c, const void* d_be32, size_t n, ps_scalars** out) {
    if (!c || !out || (n && !d_be32)) return fail(PS_ERR_ARG, "ps_scalars_from_device_be32: NULL argument");
    if (n >= (1ull << 31)) return fail(PS_ERR_ARG, "vector too long");
    HIP_TRY(hipSetDevice(c->device));
    int rc = scalars_alloc(c, n, out);
    if (rc) return rc;
    if (n) hipLaunchKernelGGL(k_scalars_from_be32, dim3(nblocks(n)), dim3(256), 0, c->stream, (const uint8_t*)d_be32,
                              (u32)n, (u32*)(*out)->st->p);
    HIP_TRY(hipGetLastError());
    if (storage_mark_ready((*out)->st, c->stream)) return fail(PS_ERR_HIP, "ps_scalars_from_device_be32: event record failed");
    return PS_OK;
}

extern "C" int ps_scalars_upload(ps_ctx* c, const uint8_t* be32, size_t n, ps_scalars** out) {
    if (!c || !out || (n && !be32)) return fail(PS_ERR_ARG, "ps_scalars_upload: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = c->staging.ensure(32 * n + 32);
    if (rc) return rc;
    // staging is reused: make sure earlier consumers on the stream are done with it
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n) HIP_TRY(hipMemcpyAsync(c->staging.p, be32, 32 * n, hipMemcpyHostToDevice, c->stream));
    rc = ps_scalars_from_device_be32(c, c->staging.p, n, out);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));  // caller's buffer is not retained past return
    return PS_OK;
}

// The context's own vector for the scalars of ONE ps_msm_be32 / ps_msm_i64 call: grown when a call needs more, never shrunk,
// resized in place (the handle is the context's alone and no sum is pending on it between calls).
static int upload_vector(ps_ctx* c, size_t n, ps_scalars** out) {
    if (!c->up_scalars || c->up_cap < n) {
        if (c->up_scalars) { ps_scalars_free(c->up_scalars); c->up_scalars = nullptr; c->up_cap = 0; }
        const size_t cap = n + n / 8 + 64;
        int rc = scalars_alloc(c, cap, &c->up_scalars);
        if (rc) return rc;
        c->up_cap = cap;
    }
    ps_scalars* s = c->up_scalars;
    s->n = n;
    s->first = 0;
    s->max_bits = 255;
    s->neg_small = false;
    *out = s;
    return PS_OK;
}

// The same for the provers' vectors (slot 0..3): handles owned by the context -- the caller must NOT free them; views made of them
// (ps_scalars_slice) are freed as usual.  No sum is pending on a context between proofs, so the memory is free to reuse.
static int prover_vector(ps_ctx* c, int slot, size_t n, ps_scalars** out) {
    if (!c->pv_scalars[slot] || c->pv_cap[slot] < n) {
        if (c->pv_scalars[slot]) { ps_scalars_free(c->pv_scalars[slot]); c->pv_scalars[slot] = nullptr; c->pv_cap[slot] = 0; }
        const size_t cap = n + n / 8 + 64;
        int rc = scalars_alloc(c, cap, &c->pv_scalars[slot]);
        if (rc) return rc;
        c->pv_cap[slot] = cap;
    }
    ps_scalars* s = c->pv_scalars[slot];
    s->n = n;
    s->first = 0;
    s->max_bits = 255;
    s->neg_small = false;
    *out = s;
    return PS_OK;
}

extern "C" int ps_scalars_upload_i64(ps_ctx* c, const int64_t* v, size_t n, ps_scalars** out) {
    if (!c || !out || (n && !v)) return fail(PS_ERR_ARG, "ps_scalars_upload_i64: NULL argument");
    if (n >= (1ull << 31)) return fail(PS_ERR_ARG, "vector too long");
    HIP_TRY(hipSetDevice(c->device));
    int rc = c->staging.ensure(8 * n + 32);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n) HIP_TRY(hipMemcpyAsync(c->staging.p, v, 8 * n, hipMemcpyHostToDevice, c->stream));
    rc = scalars_alloc(c, n, out);
    if (rc) return rc;
    if (n) hipLaunchKernelGGL(k_scalars_from_i64, dim3(nblocks(n)), dim3(256), 0, c->stream, (const int64_t*)c->staging.p,
                              (u32)n, (u32*)(*out)->st->p);
    HIP_TRY(hipGetLastError());
    bool any_neg = false;
    for (size_t i = 0; i < n; i++) any_neg |= v[i] < 0;
    (*out)->max_bits = 64;  // witnesses need only ceil(64/c) windows; negatives are folded onto -P (k_digits_grouped)
    (*out)->neg_small = any_neg;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PS_OK;
}

extern "C" int ps_scalars_download(ps_ctx* c, const ps_scalars* s, size_t first, size_t n, uint8_t* out) {
    if (!c || !s || (n && !out)) return fail(PS_ERR_ARG, "ps_scalars_download: NULL argument");
    if (first + n > s->n) return fail(PS_ERR_LENGTH, "ps_scalars_download: range out of bounds");
    if (!n) return PS_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (storage_wait_ready(s->st, c->stream)) return fail(PS_ERR_HIP, "ps_scalars_download: event wait failed");
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = c->staging.ensure(32 * n);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scalars_to_be32, dim3(nblocks(n)), dim3(256), 0, c->stream, scalars_ptr(s) + 8 * first, (u32)n,
                       (uint8_t*)c->staging.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->staging.p, 32 * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PS_OK;
}
extern "C" size_t ps_scalars_len(const ps_scalars* s) { return s ? s->n : 0; }
extern "C" int ps_scalars_slice(const ps_scalars* s, size_t first, size_t n, ps_scalars** out) {
    if (!s || !out) return fail(PS_ERR_ARG, "ps_scalars_slice: NULL argument");
    if (first + n > s->n) return fail(PS_ERR_LENGTH, "ps_scalars_slice: range out of bounds");
    s->st->refs.fetch_add(1);
    *out = new ps_scalars{n, s->st, s->first + first, s->max_bits, s->device, s->neg_small};
    return PS_OK;
}
extern "C" void ps_scalars_free(ps_scalars* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    storage_unref(s->st);
    delete s;
}

// internal: wrap a fresh device allocation of n Fr values (plain limbs) produced by the library
static ps_scalars* scalars_adopt(ps_ctx* c, void* dptr, size_t n) {
    Storage* st = new Storage();
    st->p = dptr;
    return new ps_scalars{n, st, 0, 255, c->device};
}

// ---------------------------------------------------------------------------------------
// points
// ---------------------------------------------------------------------------------------
static int points_alloc(ps_ctx* c, int group, size_t n, ps_points** out) {
    Storage* st = new Storage();
    hipError_t e = hipMalloc(&st->p, std::max<size_t>(point_bytes(group) * n, 256));
    if (e != hipSuccess) {
        delete st;
        return fail(PS_ERR_HIP, std::string("hipMalloc points: ") + hipGetErrorString(e));
    }
    st->count = n;
    *out = new ps_points{group, n, st, 0, c->device};
    return PS_OK;
}

extern "C" int ps_points_upload(ps_ctx* c, int group, const uint8_t* pts, size_t n, int fmt, ps_points** out) {
    if (!c || !out || (n && !pts)) return fail(PS_ERR_ARG, "ps_points_upload: NULL argument");
    if (group != PS_G1 && group != PS_G2) return fail(PS_ERR_ARG, "ps_points_upload: bad group");
    if (fmt != PS_FMT_AFFINE && fmt != PS_FMT_COMPRESSED) return fail(PS_ERR_ARG, "ps_points_upload: bad format");
    if (n >= (1ull << 31)) return fail(PS_ERR_ARG, "vector too long");
    HIP_TRY(hipSetDevice(c->device));
    const size_t wb = fmt == PS_FMT_AFFINE ? wire_bytes(group) : wire_bytes(group) / 2;
    int rc = c->staging.ensure(wb * n + 32);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n) HIP_TRY(hipMemcpyAsync(c->staging.p, pts, wb * n, hipMemcpyHostToDevice, c->stream));
    rc = points_alloc(c, group, n, out);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->d_flag, 0, 4, c->stream));
    if (n && fmt == PS_FMT_COMPRESSED) {  // GPU batch decompression: one Fp / Fp2 square root per point
        if (group == PS_G1)
            hipLaunchKernelGGL(k_points_decompress<Fp>, dim3(nblocks(n)), dim3(256), 0, c->stream,
                               (const uint8_t*)c->staging.p, (u32)n, (u32)wb, (Affine<Fp>*)(*out)->st->p, c->d_flag);
        else
            hipLaunchKernelGGL(k_points_decompress<Fp2>, dim3(nblocks(n)), dim3(256), 0, c->stream,
                               (const uint8_t*)c->staging.p, (u32)n, (u32)wb, (Affine<Fp2>*)(*out)->st->p, c->d_flag);
    } else if (n) {
        if (group == PS_G1)
            hipLaunchKernelGGL(k_points_from_bytes_g1, dim3(nblocks(n)), dim3(256), 0, c->stream,
                               (const uint8_t*)c->staging.p, (u32)n, (Affine<Fp>*)(*out)->st->p, c->d_flag);
        else
            hipLaunchKernelGGL(k_points_from_bytes_g2, dim3(nblocks(n)), dim3(256), 0, c->stream,
                               (const uint8_t*)c->staging.p, (u32)n, (Affine<Fp2>*)(*out)->st->p, c->d_flag);
    }
    HIP_TRY(hipGetLastError());
    u32 nbad = 0;
    HIP_TRY(hipMemcpyAsync(&nbad, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (nbad) {
        ps_points_free(*out);
        *out = nullptr;
        return fail(PS_ERR_ENCODING, std::to_string(nbad) + " point(s) not canonical / not on the curve");
    }
    return PS_OK;
}

// XYZZ -> affine for n points on c->stream (k_batch_to_affine): `tmp` holds the XYZZ points followed by the chain products.
// A context buffer, not hipMallocAsync.  Round 2 saw points 256.. of a 300-point array come back as the identity with
// staging memory from the stream-ordered pool; round 3 reproduced and isolated it (tools/malloc_async_probe.py against a
// -DPS_AFFINE_TMP_ASYNC build, profiles/r03_malloc_async_probe.txt): allocation, both kernels and the free are on ONE
// stream in the right order; a hipStreamSynchronize BETWEEN the kernels changes nothing; a device-to-host copy right after
// the first kernel already shows whole workgroups' output as zeros; a hipMemsetAsync of the block before the first
// kernel, or hipMemPoolAttrReleaseThreshold = max (freed blocks stay in the pool), makes every result right.  So: with
// the default pool every synchronisation returns the freed block to the OS, each hipMallocAsync hands out FRESHLY MAPPED
// memory, and on this runtime (ROCm 7.2, gfx950) the first kernel's writes to such memory are partly wiped -- consistent
// with the clearing of new pages not being ordered before the stream's next kernel.  Not an ordering bug of this library;
// memory that persists (this buffer) is not exposed to it.
// Threads: at least 16 points per inversion, at most 2^16 chains.
static size_t batch_affine_tmp_bytes(size_t n, size_t xyzz_bytes) { return n * (xyzz_bytes + sizeof(Fp)); }
template <class F>
static void batch_to_affine(ps_ctx* c, char* tmp, size_t n, char* out, u32 out_stride) {
    const u32 T = (u32)std::min<size_t>(65536, std::max<size_t>(1, n / 16));
    hipLaunchKernelGGL(k_batch_to_affine<F>, dim3(nblocks(T)), dim3(256), 0, c->stream, (const Xyzz<F>*)tmp, (u32)n, T,
                       (Fp*)(tmp + n * sizeof(Xyzz<F>)), out, out_stride);
}

#define PS_ASYNC_PROBE_SETUP \
        const char* probe = getenv("PS_ASYNC_PROBE"); /* experiments of tools/malloc_async_probe.py */ \
        if (probe && strchr(probe, 'r')) { /* keep freed blocks in the pool: from the second call on the memory is not fresh */ \
            static bool once = false; \
            if (!once) { \
                hipMemPool_t pool; \
                uint64_t keep = ~0ull; \
                HIP_TRY(hipDeviceGetDefaultMemPool(&pool, c->device)); \
                HIP_TRY(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep)); \
                once = true; \
            } \
        }
template <class F>
static int fixed_base(ps_ctx* c, int gi, const ps_scalars* k, ps_points* out) {
    typedef typename KernelField<F>::type KF;
    constexpr unsigned LN = FieldTraits<KF>::LANES;
    if (!c->fb_ready[gi]) {
        int rc = c->fb_table[gi].ensure(sizeof(Affine<F>) * 32 * 256);
        if (rc) return rc;
        if ((rc = c->affine_tmp.ensure(batch_affine_tmp_bytes(32 * 256, sizeof(Xyzz<F>))))) return rc;
        hipLaunchKernelGGL(k_fixed_base_table<KF>, dim3(32 * LN), dim3(256), 0, c->stream, (Xyzz<F>*)c->affine_tmp.p);
        batch_to_affine<F>(c, (char*)c->affine_tmp.p, 32 * 256, (char*)c->fb_table[gi].p, (u32)sizeof(Affine<F>));
        HIP_TRY(hipGetLastError());
        c->fb_ready[gi] = true;
    }
    if (k->n) {
#if defined(PS_AFFINE_TMP_ASYNC)  // the variant round 2 dropped (stream-ordered staging memory), kept for tools/malloc_async_probe.py
        char* tmp = nullptr;
        PS_ASYNC_PROBE_SETUP
        HIP_TRY(hipMallocAsync((void**)&tmp, batch_affine_tmp_bytes(k->n, sizeof(Xyzz<F>)), c->stream));
#else
        int rc = c->affine_tmp.ensure(batch_affine_tmp_bytes(k->n, sizeof(Xyzz<F>)));
        if (rc) return rc;
        char* tmp = (char*)c->affine_tmp.p;
#endif
#if defined(PS_AFFINE_TMP_ASYNC)
        if (probe && strchr(probe, 'm')) HIP_TRY(hipMemsetAsync(tmp, 0, batch_affine_tmp_bytes(k->n, sizeof(Xyzz<F>)), c->stream));
#endif
        hipLaunchKernelGGL(k_fixed_base_mul<KF>, dim3(nblocks(k->n * LN)), dim3(256), 0, c->stream,
                           (const Affine<F>*)c->fb_table[gi].p, scalars_ptr(k), (u32)k->n, (Xyzz<F>*)tmp);
#if defined(PS_AFFINE_TMP_ASYNC)
        if (probe && strchr(probe, 's')) HIP_TRY(hipStreamSynchronize(c->stream));
        if (probe && strchr(probe, 'd')) {  // what did the first kernel leave in memory?  (copy engine's view)
            std::vector<Xyzz<F>> host(k->n);
            HIP_TRY(hipMemcpyAsync(host.data(), tmp, sizeof(Xyzz<F>) * k->n, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            size_t zero = 0, first = k->n;
            for (size_t i = 0; i < k->n; i++)
                if (fp_all_zero(host[i].zz)) { zero++; if (first == k->n) first = i; }
            fprintf(stderr, "[probe] after k_fixed_base_mul: %zu of %zu points have ZZ == 0 in memory (first %zu), tmp = %p\n", zero, (size_t)k->n, first, (void*)tmp);
        }
#endif
        batch_to_affine<F>(c, tmp, k->n, (char*)out->st->p, (u32)sizeof(Affine<F>));
        HIP_TRY(hipGetLastError());
#if defined(PS_AFFINE_TMP_ASYNC)
        HIP_TRY(hipFreeAsync(tmp, c->stream));
#endif
    }
    return PS_OK;
}

extern "C" int ps_points_from_scalars(ps_ctx* c, int group, const ps_scalars* k, ps_points** out) {
    if (!c || !k || !out) return fail(PS_ERR_ARG, "ps_points_from_scalars: NULL argument");
    if (group != PS_G1 && group != PS_G2) return fail(PS_ERR_ARG, "bad group");
    HIP_TRY(hipSetDevice(c->device));
    int rc = points_alloc(c, group, k->n, out);
    if (rc) return rc;
    if (storage_wait_ready(k->st, c->stream)) return fail(PS_ERR_HIP, "ps_points_from_scalars: event wait failed");
    rc = group == PS_G1 ? fixed_base<Fp>(c, 0, k, *out) : fixed_base<Fp2>(c, 1, k, *out);
    if (!rc && storage_mark_ready((*out)->st, c->stream)) rc = fail(PS_ERR_HIP, "ps_points_from_scalars: event record failed");
    if (rc) {
        ps_points_free(*out);
        *out = nullptr;
    }
    return rc;
}

extern "C" int ps_points_download(ps_ctx* c, const ps_points* p, size_t first, size_t n, uint8_t* out) {
    if (!c || !p || (n && !out)) return fail(PS_ERR_ARG, "ps_points_download: NULL argument");
    if (first + n > p->n) return fail(PS_ERR_LENGTH, "ps_points_download: range out of bounds");
    if (!n) return PS_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (storage_wait_ready(p->st, c->stream)) return fail(PS_ERR_HIP, "ps_points_download: event wait failed");
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t wb = wire_bytes(p->group);
    int rc = c->staging.ensure(wb * n);
    if (rc) return rc;
    const char* src = (const char*)points_ptr(p) + first * point_bytes(p->group);
    if (p->group == PS_G1)
        hipLaunchKernelGGL(k_points_to_bytes_g1, dim3(nblocks(n)), dim3(256), 0, c->stream, (const Affine<Fp>*)src, (u32)n,
                           (uint8_t*)c->staging.p);
    else
        hipLaunchKernelGGL(k_points_to_bytes_g2, dim3(nblocks(n)), dim3(256), 0, c->stream, (const Affine<Fp2>*)src, (u32)n,
                           (uint8_t*)c->staging.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->staging.p, wb * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PS_OK;
}
extern "C" int ps_points_download_fmt(ps_ctx* c, const ps_points* p, size_t first, size_t n, int fmt, uint8_t* out) {
    if (fmt == PS_FMT_AFFINE) return ps_points_download(c, p, first, n, out);
    if (fmt != PS_FMT_COMPRESSED) return fail(PS_ERR_ARG, "ps_points_download_fmt: bad format");
    if (!c || !p || (n && !out)) return fail(PS_ERR_ARG, "ps_points_download_fmt: NULL argument");
    if (first + n > p->n) return fail(PS_ERR_LENGTH, "ps_points_download_fmt: range out of bounds");
    if (!n) return PS_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (storage_wait_ready(p->st, c->stream)) return fail(PS_ERR_HIP, "ps_points_download_fmt: event wait failed");
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t wb = wire_bytes(p->group) / 2;
    int rc = c->staging.ensure(wb * n);
    if (rc) return rc;
    const char* src = (const char*)points_ptr(p) + first * point_bytes(p->group);
    if (p->group == PS_G1)
        hipLaunchKernelGGL(k_points_compress<Fp>, dim3(nblocks(n)), dim3(256), 0, c->stream, (const Affine<Fp>*)src, (u32)n, (u32)wb,
                           (uint8_t*)c->staging.p);
    else
        hipLaunchKernelGGL(k_points_compress<Fp2>, dim3(nblocks(n)), dim3(256), 0, c->stream, (const Affine<Fp2>*)src, (u32)n, (u32)wb,
                           (uint8_t*)c->staging.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->staging.p, wb * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PS_OK;
}
extern "C" size_t ps_points_len(const ps_points* p) { return p ? p->n : 0; }
extern "C" int ps_points_group(const ps_points* p) { return p ? p->group : 0; }
extern "C" int ps_points_slice(const ps_points* p, size_t first, size_t n, ps_points** out) {
    if (!p || !out) return fail(PS_ERR_ARG, "ps_points_slice: NULL argument");
    if (first + n > p->n) return fail(PS_ERR_LENGTH, "ps_points_slice: range out of bounds");
    p->st->refs.fetch_add(1);
    *out = new ps_points{p->group, n, p->st, p->first + first, p->device};
    return PS_OK;
}
extern "C" void ps_points_free(ps_points* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    storage_unref(p->st);
    delete p;
}

// Window table of a resident array (see k_table_next).  Built for the whole allocation behind `p`, so slices
// of it (index-range shards) use it too.
static inline size_t table_row_bytes(int group) { return group == PS_G1 ? 128 : 256; }  // one / two HBM lines per point
static int table_window_for(size_t n) {
    int best = 8;
    double best_cost = 1e300;
    for (int c = 8; c <= 22; c++) {
        const int W = 255 / c + 1;
        // All windows share one bucket set, so a top window that covers only a few bits would pile its n digits onto a
        // few buckets (c = 19: 8 bits, a million entries on 116 buckets): only window sizes whose top window spans at
        // least 2^-6 of the buckets are candidates (8, 9, 10, 13, 16, 20).
        if (255 - c * (W - 1) < c - 6) continue;
        const double cost = (double)n * W * 10.0 + (double)(1u << (c - 1)) * 42.0;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}
template <class F>
static int build_table(ps_ctx* c, const void* points, size_t n, void* table, int group, int wbits, int W) {
    typedef typename KernelField<F>::type KF;
    constexpr unsigned LN = FieldTraits<KF>::LANES;
    const u32 rb = (u32)table_row_bytes(group);
    char* tab = (char*)table;
    hipLaunchKernelGGL(k_table_row0<F>, dim3(nblocks(n)), dim3(256), 0, c->stream, (const Affine<F>*)points, tab, rb, (u32)n);
    if (W > 1) {
        int rc = c->affine_tmp.ensure(batch_affine_tmp_bytes(n, sizeof(Xyzz<F>)));
        if (rc) return rc;
        char* tmp = (char*)c->affine_tmp.p;
        for (int w = 1; w < W; w++) {
            hipLaunchKernelGGL(k_table_next<KF>, dim3(nblocks(n * LN)), dim3(256), 0, c->stream, (const char*)(tab + (size_t)(w - 1) * n * rb),
                               rb, (u32)n, wbits, (Xyzz<F>*)tmp);
            batch_to_affine<F>(c, tmp, n, tab + (size_t)w * n * rb, rb);
        }
    }
    HIP_TRY(hipGetLastError());
    return PS_OK;
}
// Room for a table of `bytes`?  Within the context's budget (ps_ctx_set_table_budget; negative = automatic) and leaving a
// sixteenth of the device, at least 2 GiB, for the workspaces of the sums themselves (buckets, sort buffers).
static bool table_fits(const ps_ctx* c, size_t bytes) {
    if (c->table_budget >= 0 && bytes > (size_t)c->table_budget) return false;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    const size_t reserve = std::max<size_t>(total_b / 16, (size_t)2 << 30);
    return bytes + reserve <= free_b;
}
// What a context had to offer when a table was declined, and whether a later request has more: a larger budget, or a
// quarter more free device memory (hipMemGetInfo is only asked on this path -- a key whose table was declined).
static void decline_record(const ps_ctx* c, long long* budget, size_t* free_bytes) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    *budget = c->table_budget;
    *free_bytes = free_b;
}
static bool decline_outdated(const ps_ctx* c, long long budget, size_t free_bytes) {
    if (budget >= 0 && (c->table_budget < 0 || c->table_budget > budget)) return true;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    return free_b > free_bytes + free_bytes / 4;
}

// The table of the allocation behind `p` for `window_bits` (0 = automatic), built once whoever asks first.
// optional: a prover's request -- no room (budget, free memory, or hipMalloc failing) is not an error: the array is marked
// and the sums over it take the plain plan.  Returns with the table complete in memory (the context stream is waited for).
static int points_ensure_table(ps_ctx* c, const ps_points* p, int window_bits, bool optional) {
    Storage* st = p->st;
    const size_t n = st->count;
    if (n == 0) return PS_OK;
    if (n >= (1ull << ENTRY_W_SHIFT)) {
        if (optional) return PS_OK;
        return fail(PS_ERR_ARG, "ps_points_precompute: at most 2^26 - 1 points per array (split it with ps_points_slice before uploading)");
    }
    const int wbits = window_bits ? window_bits : table_window_for(n);
    if (st->table.load(std::memory_order_acquire) && st->table_c == wbits) return PS_OK;
    std::lock_guard<std::mutex> lock(st->mu);
    if (st->table.load(std::memory_order_acquire) && st->table_c == wbits) return PS_OK;  // another context built it meanwhile
    if (optional && st->table.load()) return PS_OK;  // the caller's own table stays
    if (optional && st->table_declined && !decline_outdated(c, st->declined_budget, st->declined_free)) return PS_OK;  // asked before, nothing has changed
    const int W = 255 / wbits + 1;
    const size_t bytes = table_row_bytes(p->group) * n * (size_t)W;
    const size_t work = batch_affine_tmp_bytes(n, p->group == PS_G1 ? sizeof(Xyzz<Fp>) : sizeof(Xyzz<Fp2>));
    if (optional && !table_fits(c, bytes + (c->affine_tmp.cap >= work ? 0 : work))) {
        decline_record(c, &st->declined_budget, &st->declined_free);
        st->table_declined = true;
        return PS_OK;
    }
    st->table_declined = false;
    if (storage_wait_ready(st, c->stream)) return fail(PS_ERR_HIP, "ps_points_precompute: event wait failed");
    if (void* old = st->table.load()) {  // another window size: sums of THIS process that may still read the old table are waited for
        HIP_TRY(hipDeviceSynchronize());
        st->table.store(nullptr, std::memory_order_release);
        (void)hipFree(old);
    }
    void* tab = nullptr;
    hipError_t e = hipMalloc(&tab, bytes);
    int rc = PS_OK;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rc = fail(PS_ERR_HIP, std::string("hipMalloc window table: ") + hipGetErrorString(e));
    } else {
        rc = p->group == PS_G1 ? build_table<Fp>(c, st->p, n, tab, PS_G1, wbits, W) : build_table<Fp2>(c, st->p, n, tab, PS_G2, wbits, W);
        if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail(PS_ERR_HIP, "ps_points_precompute: building the table failed");
        if (rc) (void)hipFree(tab);
    }
    if (rc) {
        if (!optional) return rc;
        decline_record(c, &st->declined_budget, &st->declined_free);
        st->table_declined = true;  // no room after all (e.g. the staging buffer): the plain plan needs no extra memory
        g_last_error.clear();
        return PS_OK;
    }
    st->table_c = wbits;
    st->table_W = W;
    st->table_stride = n;
    st->table.store(tab, std::memory_order_release);
    return PS_OK;
}
// ---- window tables of index ranges (ViewTable) ----
static const ViewTable* view_table_find(const Storage* st, size_t first, size_t n, int wbits /* 0: any */) {
    for (const auto& slot : st->vtab) {
        const ViewTable* v = slot.load(std::memory_order_acquire);
        if (!v) break;  // slots fill in order
        if (v->first == first && v->n == n && (wbits == 0 || v->c == wbits)) return v;
    }
    return nullptr;
}
// Table of the index range behind the view `p` (p->first, p->n) for its own length, built on first use.  Always optional:
// no room (budget, free memory, no free slot) leaves the view on whatever plan it had.  Returns the table or nullptr.
static const ViewTable* view_table_ensure(ps_ctx* c, const ps_points* p) {
    Storage* st = p->st;
    const size_t n = p->n;
    const int wbits = table_window_for(n);
    const ViewTable* have = view_table_find(st, p->first, n, 0);
    if (have && (have->table.load(std::memory_order_acquire) || !decline_outdated(c, have->declined_budget, have->declined_free)))
        return have->table.load(std::memory_order_acquire) ? have : nullptr;
    std::lock_guard<std::mutex> lock(st->mu);
    have = view_table_find(st, p->first, n, 0);
    int slot = -1;
    for (int i = 0; i < Storage::VIEW_SLOTS; i++) {
        ViewTable* v = st->vtab[i].load(std::memory_order_acquire);
        if (!v) { slot = i; break; }
        if (v == have) {
            if (v->table.load(std::memory_order_acquire)) return v;  // another context built it meanwhile
            if (!decline_outdated(c, v->declined_budget, v->declined_free)) return nullptr;
        }
    }
    const int W = 255 / wbits + 1;
    const size_t bytes = table_row_bytes(p->group) * n * (size_t)W;
    const size_t work = batch_affine_tmp_bytes(n, p->group == PS_G1 ? sizeof(Xyzz<Fp>) : sizeof(Xyzz<Fp2>));
    ViewTable* decl = const_cast<ViewTable*>(have);  // a declined record is refreshed in place (only its decline fields, under mu)
    auto declined = [&]() -> const ViewTable* {
        if (decl) { decline_record(c, &decl->declined_budget, &decl->declined_free); return nullptr; }
        if (slot < 0) return nullptr;
        ViewTable* v = new ViewTable{p->first, n, wbits, W, {nullptr}, 0, 0};
        decline_record(c, &v->declined_budget, &v->declined_free);
        st->vtab[slot].store(v, std::memory_order_release);
        return nullptr;
    };
    if (slot < 0 && !decl) return nullptr;  // every slot taken by other ranges
    if (!table_fits(c, bytes + (c->affine_tmp.cap >= work ? 0 : work))) return declined();
    if (storage_wait_ready(st, c->stream)) return nullptr;
    void* tab = nullptr;
    if (hipMalloc(&tab, bytes) != hipSuccess) { (void)hipGetLastError(); return declined(); }
    const void* base = points_ptr(p);
    int rc = p->group == PS_G1 ? build_table<Fp>(c, base, n, tab, PS_G1, wbits, W) : build_table<Fp2>(c, base, n, tab, PS_G2, wbits, W);
    if (hipStreamSynchronize(c->stream) != hipSuccess || rc) {
        (void)hipFree(tab);
        g_last_error.clear();
        return declined();
    }
    if (decl) {
        // the declined record becomes the table: readers that saw table == nullptr took another plan; publish the pointer last
        decl->c = wbits; decl->W = W;
        decl->table.store(tab, std::memory_order_release);
        return decl;
    }
    ViewTable* v = new ViewTable{p->first, n, wbits, W, {tab}, 0, 0};
    st->vtab[slot].store(v, std::memory_order_release);
    return v;
}

extern "C" int ps_points_precompute(ps_ctx* c, ps_points* p, int window_bits) {
    if (!c || !p) return fail(PS_ERR_ARG, "ps_points_precompute: NULL argument");
    if (window_bits == -1) {  // release
        Storage* st0 = p->st;
        std::lock_guard<std::mutex> lock(st0->mu);
        void* old = st0->table.load();
        st0->table_declined = false;
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipDeviceSynchronize());
        storage_free_view_tables(st0);
        if (!old) return PS_OK;
        st0->table.store(nullptr, std::memory_order_release);
        (void)hipFree(old);
        st0->table_c = st0->table_W = 0;
        return PS_OK;
    }
    if (window_bits != 0 && (window_bits < 8 || window_bits > 22)) return fail(PS_ERR_ARG, "ps_points_precompute: window bits must be 0 (automatic), -1 (release) or 8..22");
    HIP_TRY(hipSetDevice(c->device));
    return points_ensure_table(c, p, window_bits, false);
}
extern "C" int ps_points_table_window(const ps_points* p) { return p && p->st->table.load(std::memory_order_acquire) ? p->st->table_c : 0; }

// Every point of the array in the order-r subgroup?  ([r]P on the device: ~400 group operations per point;
// an opt-in check for arrays that arrive from outside -- ps_points_upload itself only tests the curve equation.)
extern "C" int ps_points_check_subgroup(ps_ctx* c, const ps_points* p, int* ok) {
    if (!c || !p || !ok) return fail(PS_ERR_ARG, "ps_points_check_subgroup: NULL argument");
    *ok = 0;
    HIP_TRY(hipSetDevice(c->device));
    if (storage_wait_ready(p->st, c->stream)) return fail(PS_ERR_HIP, "ps_points_check_subgroup: event wait failed");
    HIP_TRY(hipMemsetAsync(c->d_flag, 0, 4, c->stream));
    if (p->n) {
        if (p->group == PS_G1)
            hipLaunchKernelGGL(k_points_subgroup<Fp>, dim3(nblocks(p->n)), dim3(256), 0, c->stream, (const Affine<Fp>*)points_ptr(p),
                               (u32)p->n, c->d_flag);
        else
            hipLaunchKernelGGL(k_points_subgroup<Fp2>, dim3(nblocks(p->n)), dim3(256), 0, c->stream, (const Affine<Fp2>*)points_ptr(p),
                               (u32)p->n, c->d_flag);
    }
    HIP_TRY(hipGetLastError());
    u32 nbad = 0;
    HIP_TRY(hipMemcpyAsync(&nbad, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *ok = nbad == 0;
    return PS_OK;
}

// ---------------------------------------------------------------------------------------
// host-side folding of window sums / partial sums (same field code, compiled for the host)
// ---------------------------------------------------------------------------------------
static void fp_to_be48_host(uint8_t* p, const Fp& a) { fp_to_be48(p, a); }
static bool fp_from_be48_host(Fp& out, const uint8_t* p) {
    bool ok = true;
    out = fp_from_be48(p, ok);
    return ok;
}
static void write_affine(uint8_t* out, const Xyzz<Fp>& acc) {
    Fp x, y;
    if (!xyzz_to_affine<Fp>(acc, x, y)) {
        memset(out, 0, 96);
        out[0] = 0x40;
        return;
    }
    fp_to_be48_host(out, x);
    fp_to_be48_host(out + 48, y);
}
static void write_affine(uint8_t* out, const Xyzz<Fp2>& acc) {
    Fp2 x, y;
    if (!xyzz_to_affine<Fp2>(acc, x, y)) {
        memset(out, 0, 192);
        out[0] = 0x40;
        return;
    }
    fp_to_be48_host(out, x.c1);
    fp_to_be48_host(out + 48, x.c0);
    fp_to_be48_host(out + 96, y.c1);
    fp_to_be48_host(out + 144, y.c0);
}
// the same for sums formed in the host representation (hostfield.inc)
static void write_affine(uint8_t* out, const Xyzz<Fq>& acc) {
    Fq x, y;
    if (!xyzz_to_affine<Fq>(acc, x, y)) { memset(out, 0, 96); out[0] = 0x40; return; }
    fq_to_be48(out, x);
    fq_to_be48(out + 48, y);
}
static void write_affine(uint8_t* out, const Xyzz<Fq2>& acc) {
    Fq2 x, y;
    if (!xyzz_to_affine<Fq2>(acc, x, y)) { memset(out, 0, 192); out[0] = 0x40; return; }
    fq_to_be48(out, x.c1);
    fq_to_be48(out + 48, x.c0);
    fq_to_be48(out + 96, y.c1);
    fq_to_be48(out + 144, y.c0);
}
template <class F>
static Xyzz<typename HostField<F>::type> xyzz_to_host(const Xyzz<F>& p) {
    Xyzz<typename HostField<F>::type> r;
    r.x = to_host(p.x); r.y = to_host(p.y); r.zz = to_host(p.zz); r.zzz = to_host(p.zzz);
    return r;
}
template <class F>
static Affine<typename HostField<F>::type> affine_to_host(const Affine<F>& a) {
    Affine<typename HostField<F>::type> r;
    r.x = to_host(a.x); r.y = to_host(a.y);
    return r;
}
static bool is_identity_encoding(const uint8_t* p, size_t wb) {  // 0x40 followed by zeros, nothing else
    if (p[0] != 0x40) return false;
    for (size_t i = 1; i < wb; i++)
        if (p[i]) return false;
    return true;
}
static bool read_affine(Affine<Fp>& a, const uint8_t* p) {
    if (p[0] & 0x40) { a.x = fp_zero(); a.y = fp_zero(); return is_identity_encoding(p, 96); }
    if (p[0] & 0xE0) return false;
    if (!fp_from_be48_host(a.x, p) || !fp_from_be48_host(a.y, p + 48)) return false;
    return affine_on_curve<Fp>(a);
}
static bool read_affine(Affine<Fp2>& a, const uint8_t* p) {
    if (p[0] & 0x40) { a.x = f_zero((const Fp2*)0); a.y = f_zero((const Fp2*)0); return is_identity_encoding(p, 192); }
    if (p[0] & 0xE0) return false;
    if (!fp_from_be48_host(a.x.c1, p) || !fp_from_be48_host(a.x.c0, p + 48) || !fp_from_be48_host(a.y.c1, p + 96) ||
        !fp_from_be48_host(a.y.c0, p + 144))
        return false;
    return affine_on_curve<Fp2>(a);
}

template <class F>
static int points_sum_t(const uint8_t* pts, size_t k, uint8_t* out, size_t wb) {
    typedef typename HostField<F>::type H;
    Xyzz<H> acc = xyzz_identity<H>();
    for (size_t i = 0; i < k; i++) {
        Affine<F> a;
        if (!read_affine(a, pts + wb * i)) return fail(PS_ERR_ENCODING, "ps_points_sum: bad point encoding");
        if (!affine_is_identity<F>(a)) { Affine<H> h = affine_to_host<F>(a); xyzz_madd<H>(acc, h.x, h.y); }
    }
    write_affine(out, acc);
    return PS_OK;
}
template <class F>
static int point_convert_t(int in_fmt, int out_fmt, const uint8_t* in, uint8_t* out, size_t wb) {
    Affine<F> a;
    bool ok = in_fmt == PS_FMT_AFFINE ? read_affine(a, in) : decompress_point(a, in);
    if (!ok) return fail(PS_ERR_ENCODING, "ps_point_convert: bad point encoding");
    if (out_fmt == PS_FMT_COMPRESSED) {
        compress_point(out, a);
        return PS_OK;
    }
    if (affine_is_identity<F>(a)) { memset(out, 0, wb); out[0] = 0x40; return PS_OK; }
    write_affine(out, xyzz_from_affine<F>(a.x, a.y));
    return PS_OK;
}
extern "C" int ps_point_convert(int group, int in_fmt, int out_fmt, const uint8_t* in, uint8_t* out) {
    if (!in || !out) return fail(PS_ERR_ARG, "ps_point_convert: NULL argument");
    if ((in_fmt != PS_FMT_AFFINE && in_fmt != PS_FMT_COMPRESSED) || (out_fmt != PS_FMT_AFFINE && out_fmt != PS_FMT_COMPRESSED))
        return fail(PS_ERR_ARG, "ps_point_convert: bad format");
    if (group == PS_G1) return point_convert_t<Fp>(in_fmt, out_fmt, in, out, 96);
    if (group == PS_G2) return point_convert_t<Fp2>(in_fmt, out_fmt, in, out, 192);
    return fail(PS_ERR_ARG, "bad group");
}
extern "C" int ps_points_sum(int group, const uint8_t* pts, size_t k, uint8_t* out) {
    if ((k && !pts) || !out) return fail(PS_ERR_ARG, "ps_points_sum: NULL argument");
    if (group == PS_G1) return points_sum_t<Fp>(pts, k, out, 96);
    if (group == PS_G2) return points_sum_t<Fp2>(pts, k, out, 192);
    return fail(PS_ERR_ARG, "bad group");
}

// ---------------------------------------------------------------------------------------
// MSM driver
// ---------------------------------------------------------------------------------------
// The pipeline is split in two: the sort (digits, scan, scatter) depends on the scalars only, the
// point pass (accumulate, fixup, reduce) on one point array.  ps_msm_multi runs one sort for several
// point arrays that share a scalar vector (computeSolCommit x9, pinochio.go:231-241).
#define PS_STAGE_MARK() do { if (timed) HIP_TRY(hipEventRecord(c->ev[evi++], st)); } while (0)
// `st`: the stream the sort runs on.  With sums in flight that is the workspace's HIGH-PRIORITY stream: the sort of sum i+1
// shares the chip with the accumulation of sum i, whose waves hold 468 of a SIMD's 512 registers -- on the normal-priority
// stream its workgroups were dispatched behind the accumulation's own (k_sort_partition: 2.0-2.6 ms instead of 0.09 in a
// kernel trace of 2^20-point sums), the next accumulation waited for it, and the step was 2.78 ms for 2.30 ms of accumulation.
static int msm_sort(ps_ctx* c, const ps_scalars* sc, const MsmPlan& pl, bool timed, hipStream_t st) {
    const size_t n = sc->n;
    const u64 total = (u64)pl.W * n;  // upper bound on entries
    const u64 G = pl.G;
    const u32 scan_tiles = (u32)((G + SCAN_TILE - 1) / SCAN_TILE);
    int rc;
    if ((rc = c->counts.ensure(4 * G))) return rc;
    if ((rc = c->offs.ensure(4 * (G + 1)))) return rc;
    if ((rc = c->bsum.ensure(4 * (size_t)scan_tiles + 4))) return rc;
    if ((rc = c->keys.ensure(4 * total))) return rc;
    if ((rc = c->ranks.ensure(4 * total))) return rc;
    if ((rc = c->vals.ensure(4 * total))) return rc;
    if ((rc = c->sorted.ensure(4 * total + 4))) return rc;
    if (c->tail_used) HIP_TRY(hipStreamWaitEvent(st, c->ev_tail_done, 0));  // the last tail still reads offs
    if (storage_wait_ready(sc->st, st)) return fail(PS_ERR_HIP, "msm: event wait failed");  // asynchronously produced scalars
    int evi = 0;
    PS_STAGE_MARK();  // 0: start
    DigitConst cadd{};  // C = sum_{w < W-1} 2^(c*w + c-1)
    for (int w = 0; w + 1 < pl.W; w++) {
        int bit = w * pl.c + pl.c - 1;
        if (bit < 256) cadd.w[bit >> 5] |= 1u << (bit & 31);
    }
    const dim3 dgrid((unsigned)((n + DIGITS_CHUNK - 1) / DIGITS_CHUNK), (unsigned)pl.W);
    const int fold_neg = sc->neg_small ? 1 : 0, single = pl.sets == 1 ? 1 : 0;
    if (G <= SORT_MAX_BUCKETS) {  // two-level counting sort, no per-entry global atomics (msm.hpp 1'-3')
        int fb = 5;  // fine bits: as few as keep the coarse bins within SORT_MAX_COARSE
        while (((G + (1ull << fb) - 1) >> fb) > SORT_MAX_COARSE) fb++;
#if defined(PS_TAIL_TUNE)
        if (const char* e = getenv("PS_T_FB")) fb = std::max(fb, std::min(atoi(e), 10));
#endif
        const u32 ncoarse = (u32)((G + (1ull << fb) - 1) >> fb);
        if ((rc = c->coarse.ensure(4 * (4 * (size_t)SORT_MAX_COARSE + 8)))) return rc;
        u32* coarse_cnt = (u32*)c->coarse.p;
        u32* coarse_off = coarse_cnt + SORT_MAX_COARSE + 1;
        u32* coarse_cur = coarse_off + SORT_MAX_COARSE + 1;
        u32* tile_base = coarse_cur + SORT_MAX_COARSE + 1;
        HIP_TRY(hipMemsetAsync(coarse_cnt, 0, 4 * SORT_MAX_COARSE, st));
        const bool may_have_big_bins = total > SORT_BIG;  // a bin of more than SORT_BIG entries needs that many digits
        if (may_have_big_bins) HIP_TRY(hipMemsetAsync(c->counts.p, 0, 4 * G, st));  // per-bucket counters of the big bins (k_sort_big_*)
        const dim3 cgrid((unsigned)((n + (size_t)COUNT_PER_THREAD * DIGITS_THREADS - 1) / ((size_t)COUNT_PER_THREAD * DIGITS_THREADS)));
        hipLaunchKernelGGL(k_sort_count, cgrid, dim3(DIGITS_THREADS), 0, st, scalars_ptr(sc), (u32)n, pl.c, pl.W, pl.NB, cadd, fold_neg,
                           single, ncoarse, fb, (u32*)c->ranks.p, coarse_cnt);
        PS_STAGE_MARK();  // 1: after digits + coarse histogram
        hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(SORT_MAX_COARSE), 0, st, (const u32*)coarse_cnt, ncoarse, coarse_off, coarse_cur,
                           (u32*)c->offs.p + G, tile_base);
        hipLaunchKernelGGL(k_sort_partition, dgrid, dim3(DIGITS_THREADS), 2 * DIGITS_CHUNK * sizeof(u32), st, (const u32*)c->ranks.p,
                           (u32)n, pl.NB, single, fb, coarse_cur, (unsigned short*)c->keys.p, (u32*)c->vals.p);
        PS_STAGE_MARK();  // 2: after scan + partition
        hipLaunchKernelGGL(k_sort_fine, dim3(ncoarse), dim3(SORT_FINE), 0, st, (const unsigned short*)c->keys.p, (const u32*)c->vals.p,
                           (const u32*)coarse_off, (u32)G, fb, (u32*)c->offs.p, (u32*)c->sorted.p);
        // bins with more than SORT_BIG entries, tile by tile (no tiles: the three kernels return at once; short sums, which
        // are bound by the number of launches, skip them)
        const unsigned big_grid = (unsigned)std::min<u64>(2048, total / SORT_TILE + 1);
        if (may_have_big_bins) {
        hipLaunchKernelGGL(k_sort_big_count, dim3(big_grid), dim3(SORT_FINE), 0, st, (const unsigned short*)c->keys.p, (const u32*)coarse_off,
                           (const u32*)tile_base, ncoarse, fb, (u32*)c->counts.p);
        hipLaunchKernelGGL(k_sort_big_scan, dim3(ncoarse), dim3(SORT_FINE), 0, st, (const u32*)coarse_off, (u32)G, fb, (u32*)c->counts.p,
                           (u32*)c->offs.p);
        hipLaunchKernelGGL(k_sort_big_scatter, dim3(big_grid), dim3(SORT_FINE), 0, st, (const unsigned short*)c->keys.p,
                           (const u32*)c->vals.p, (const u32*)coarse_off, (const u32*)tile_base, ncoarse, fb, (u32*)c->counts.p,
                           (u32*)c->sorted.p);
        }
    } else {
        HIP_TRY(hipMemsetAsync(c->counts.p, 0, 4 * G, st));
        {
            int bin_shift = 4;  // 16 counters = one 64-byte line
            while ((pl.NB >> bin_shift) > (u32)DIGITS_BINS) bin_shift++;
            hipLaunchKernelGGL(k_digits_grouped, dgrid, dim3(DIGITS_THREADS), 2 * DIGITS_CHUNK * sizeof(u32), st, scalars_ptr(sc),
                               (u32)n, pl.c, pl.W, pl.NB, cadd, bin_shift, fold_neg, single, (u32*)c->counts.p, (u32*)c->keys.p,
                               (u32*)c->vals.p, (u32*)c->ranks.p);
        }
        PS_STAGE_MARK();  // 1: after memset + digits
        hipLaunchKernelGGL(k_scan_blocks, dim3(scan_tiles), dim3(SCAN_BLOCK), 0, st, (const u32*)c->counts.p, (u32*)c->offs.p,
                           (u32*)c->bsum.p, (u64)G);
        hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, (u32*)c->bsum.p, scan_tiles, (u32*)c->offs.p + G);
        hipLaunchKernelGGL(k_scan_add, dim3(scan_tiles), dim3(SCAN_BLOCK), 0, st, (u32*)c->offs.p, (const u32*)c->bsum.p, (u64)G);
        PS_STAGE_MARK();  // 2: after scan
        hipLaunchKernelGGL(k_scatter, dim3(nblocks(total)), dim3(256), 0, st, (const u32*)c->keys.p, (const u32*)c->vals.p,
                           (const u32*)c->ranks.p, (const u32*)c->offs.p, total, (u32*)c->sorted.p);
    }
    PS_STAGE_MARK();  // 3: after scatter
    HIP_TRY(hipGetLastError());
    // entry count for introspection (read back with the window sums; a short sum's last kernel appends it to its results)
    if (!pl.qtail) HIP_TRY(hipMemcpyAsync((char*)c->h_pinned + c->h_pinned_cap - 8, (u32*)c->offs.p + G, 4, hipMemcpyDeviceToHost, st));
    return PS_OK;
}

// The window table a sum over (pts, n) with c-bit windows and W digits reads: the table of the whole allocation (rows of
// table_stride points, the view starts at its own offset) or the table of exactly this index range (ViewTable).
struct TableRef { const char* base; u64 stride; };
static bool table_ref(const ps_points* pts, size_t n, int wbits, int W, TableRef* out) {
    const Storage* st = pts->st;
    const size_t rb = table_row_bytes(pts->group);
    if (const void* t = st->table.load(std::memory_order_acquire)) {
        if (st->table_c == wbits && W <= st->table_W) {
            *out = TableRef{(const char*)t + pts->first * rb, (u64)st->table_stride};
            return true;
        }
    }
    if (const ViewTable* v = view_table_find(st, pts->first, n, wbits)) {
        const void* t = v->table.load(std::memory_order_acquire);
        if (t && W <= v->W) {
            *out = TableRef{(const char*)t, (u64)v->n};
            return true;
        }
    }
    return false;
}

// Window sums of one point array over the sorted entries of `c`, copied to pinned slot `slot` of `c`.
// The work buffers and the stream are those of `wc` (c itself, or c->aux when ps_msm_multi
// alternates two workspaces so that the latency-bound tail of one sum -- fix-up and reduction, ~1.3 ms
// of short dependency chains on a mostly idle chip -- runs under the next sum's accumulation).
// wait_acc: event the accumulation waits for; acc_done: recorded right after it.
template <class F>
static int msm_points_t(ps_ctx* c, ps_ctx* wc, const ps_points* pts, size_t n, const MsmPlan& pl, bool timed, int slot,
                        hipEvent_t wait_acc, hipEvent_t acc_done, bool inline_tail = false) {
    typedef typename KernelField<F>::type KF;      // Fp -> Fp, Fp2 -> lane-split Fp2s
    constexpr unsigned LN = FieldTraits<KF>::LANES;  // lanes per logical thread
    const u64 total = (u64)pl.W * n;
    const u64 G = pl.G;
    const u32 nthreads_acc = (u32)((total + pl.M - 1) / pl.M);
    ReducePlan rp = reduce_plan(pl.NB);
    if (pl.qtail) {  // row sums and column sums in `segs`, c results per set: [S, W_0 .. W_{c-2}] (qtail.hpp)
        rp.small = false;
        rp.njobs = (u32)pl.c;
        rp.nblk = 1;
        rp.m = 0;
        rp.segs = ((1u << (pl.c - 1 - pl.rc_s)) + (1u << pl.rc_s) + 1) / 2;  // two arrays of `segs` points hold R and C
    }
    const u32 nseg_total = rp.segs * (u32)pl.sets;
    const u32 per_role = rp.m * (u32)pl.sets, nres = rp.njobs * (u32)pl.sets;
    const bool tab = pl.table;
    const u32 pstride = tab ? (u32)table_row_bytes(pts->group) : (u32)sizeof(Affine<F>);
    TableRef tref{};
    if (tab && !table_ref(pts, n, pl.c, pl.W, &tref)) return fail(PS_ERR_ARG, "msm: internal: the plan names a window table the array does not carry");
    const char* src = tab ? tref.base : (const char*)points_ptr(pts);
    const u32 idx_mask = tab ? (1u << ENTRY_W_SHIFT) - 1u : 0x7fffffffu;
    const u64 w_stride = tab ? tref.stride : 0ull;
    int rc;
    if ((rc = wc->buckets.ensure(sizeof(Xyzz<F>) * G))) return rc;
    if ((rc = wc->parts.ensure(sizeof(Xyzz<F>) * 2 * (size_t)nthreads_acc))) return rc;
    if ((rc = wc->segs.ensure(sizeof(Xyzz<F>) * (2 * (size_t)nseg_total + 5 * (size_t)per_role + (size_t)nres * rp.nblk)))) return rc;
    if ((rc = wc->wins.ensure(sizeof(Xyzz<F>) * ((size_t)nres + (size_t)pl.sets) + 16))) return rc;
    // a heavy bucket spans >= HEAVY_SPAN slices for the one-lane fix-up, >= QFIX_HEAVY x (quads per bucket) for the quad fix-up
    const bool qfix = pl.shortsum && pl.lpb >= 1;
    const u32 qlpb = qfix ? std::min<u32>((u32)pl.lpb, 256 / QTraits<KF>::GL) : 0;
    const u64 heavy_slices = qfix ? std::min<u64>(HEAVY_SPAN, (u64)QFIX_HEAVY * qlpb - 1) : HEAVY_SPAN;
    const size_t max_heavy = (size_t)(total / (heavy_slices * (u64)pl.M)) + 2;
    // heavy: [count][bucket list: max_heavy][job_base: max_heavy + 1]; hparts: one point per job
    if ((rc = wc->heavy.ensure(4 * (2 * max_heavy + 2)))) return rc;
    const u32 heavy_npb = 256 / QTraits<KF>::GL;  // a job is at least one block's worth of slices (heavy_chunk_of)
    const size_t max_jobs = (size_t)nthreads_acc / heavy_npb + max_heavy + 1;
    if ((rc = wc->hparts.ensure(sizeof(Xyzz<F>) * max_jobs))) return rc;
    if (sizeof(Xyzz<F>) * (pl.sets > 1 ? (size_t)pl.sets : (size_t)nres) + 16 > PS_PINNED_SLOT) return fail(PS_ERR_ARG, "too many windows");
    hipStream_t st = wc->stream;
    int evi = 4;  // ev[3] = after the scatter (msm_sort); ev[4] = the accumulation may start
    if (wc->tail_used) HIP_TRY(hipStreamWaitEvent(st, wc->ev_tail_done, 0));  // buffers of the previous sum
    if (storage_wait_ready(pts->st, st)) return fail(PS_ERR_HIP, "msm: event wait failed");  // asynchronously produced points
    if (wait_acc) HIP_TRY(hipStreamWaitEvent(st, wait_acc, 0));
    PS_STAGE_MARK();  // 4: buffers cleared and the previous sum's accumulation done ("queue")
    constexpr bool PF = LN == 1 || PS_G2_ACC_WAVES == 1;  // next point prefetched (the lane-pair G2 kernel at two waves per SIMD has no registers to spare)
    hipLaunchKernelGGL((k_accumulate<KF, PF>), dim3(nblocks((size_t)nthreads_acc * LN)), dim3(256), 0, st, src,
                       (const u32*)c->sorted.p, (const u32*)c->offs.p, (u32)G, pl.M, nthreads_acc, idx_mask, w_stride, pstride,
                       (Xyzz<F>*)wc->buckets.p, (Xyzz<F>*)wc->parts.p, (u32*)wc->heavy.p);
    if (acc_done) HIP_TRY(hipEventRecord(acc_done, st));
    PS_STAGE_MARK();  // 5: after accumulate
    HIP_TRY(hipEventRecord(wc->ev_acc_local, st));
    if (!inline_tail) {  // a lone short sum keeps one stream: the hop to the tail stream costs it ~10 us and buys nothing
        st = wc->tail;   // ---- the rest runs on the high-priority tail stream ----
        HIP_TRY(hipStreamWaitEvent(st, wc->ev_acc_local, 0));
    }
    if (pl.shortsum && pl.lpb >= 1) {
        // the plan of a multi-sum is shared by arrays of both groups (PHGR13: six G1 sums and a G2 one over one sort): the
        // quads of a bucket must fit THIS group's 256-thread block (64 G1 points, 32 G2 points)
        const u32 lpb = qlpb;
        hipLaunchKernelGGL(k_qfixup<KF>, dim3(nblocks(G * (u64)lpb * QTraits<KF>::GL)), dim3(256), 0, st, (const u32*)c->offs.p, (u32)G, pl.M,
                           nthreads_acc, lpb, (const Xyzz<F>*)wc->parts.p, (Xyzz<F>*)wc->buckets.p, (u32*)wc->heavy.p, (u32*)wc->heavy.p + 1);
    } else
    hipLaunchKernelGGL(k_fixup<KF>, dim3(nblocks(G * LN)), dim3(256), 0, st, (const u32*)c->offs.p, (u32)G, pl.M, nthreads_acc,
                       (const Xyzz<F>*)wc->parts.p, (Xyzz<F>*)wc->buckets.p, (u32*)wc->heavy.p, (u32*)wc->heavy.p + 1);
    {
        const u32* hcount = (const u32*)wc->heavy.p;
        const u32* hlist = hcount + 1;
        u32* job_base = (u32*)wc->heavy.p + 1 + max_heavy;
        hipLaunchKernelGGL(k_heavy_jobs, dim3(1), dim3(256), 0, st, (const u32*)c->offs.p, (u32)G, pl.M, nthreads_acc, hcount, hlist, job_base, heavy_npb);
        // two levels of quad trees for every plan (qtail.hpp): the one-lane kernels of rounds 1-2 (jobs of 1024 slices, strided
        // chains, an 8-level LDS tree of 14-36 us additions) took 0.43 ms for the 2^19 ones of a boolean witness at 2^20 points
        hipLaunchKernelGGL(k_qfixup_heavy_part<KF>, dim3(1024), dim3(256), 0, st, (const u32*)c->offs.p, (u32)G, pl.M, nthreads_acc,
                           (const Xyzz<F>*)wc->parts.p,
                           hcount, hlist, (const u32*)job_base, (Xyzz<F>*)wc->hparts.p);
        hipLaunchKernelGGL(k_qfixup_heavy<KF>, dim3(256), dim3(256), 0, st, (const Xyzz<F>*)wc->hparts.p, (Xyzz<F>*)wc->buckets.p, hcount,
                           hlist, (const u32*)job_base);
    }
    PS_STAGE_MARK();  // 6: after fixup
    {
        Xyzz<F>* accs = (Xyzz<F>*)wc->segs.p;
        Xyzz<F>* runs = accs + nseg_total;
        Xyzz<F>* lvl = runs + nseg_total;
        Xyzz<F>* pieces = lvl + 5 * (size_t)per_role;
        if (pl.qtail) {
            // row / column sums, then the bit sums of both (qtail.hpp); small sets: the bit sums straight from the buckets
            constexpr u32 QGL = QTraits<KF>::GL, QNP = 512 / QGL;
            const int cb = pl.c - 1, s = pl.rc_s;
            const u32 rows = 1u << (cb - s), cols = 1u << s;
            Xyzz<F>* R = accs;
            Xyzz<F>* C = accs + (size_t)pl.sets * rows;
            if (s > 0) {
                // quads per row / column job: all jobs resident at once (a quad kernel holds ~250 VGPRs: 8 waves per CU,
                // 2048 on the chip -- a second round of blocks would double the kernel's time), at most one term per quad
                u32 np = std::min<u32>(QNP, std::max(rows, cols));
                while (np > 64 / QGL && (u64)np * QGL * (rows + cols) * (u32)pl.sets > 2048ull * 64) np >>= 1;
#if defined(PS_TAIL_TUNE)
                if (pl.busy) if (const char* e = getenv("PS_T_NP_BUSY")) np = std::max<u32>(64 / QGL, std::min<u32>(np, (u32)atoi(e)));
#endif
                hipLaunchKernelGGL(k_qreduce_rowcol<KF>, dim3((rows + cols) * (u32)pl.sets), dim3(np * QGL), np * QGL * sizeof(Fp), st,
                                   (const Xyzz<F>*)wc->buckets.p, cb, s, R, C, (const Xyzz<F>*)nullptr, (Xyzz<F>*)nullptr);
            }
            const u32 terms = std::max(rows, cols / 2);
            const u32 np = std::min<u32>(QNP, std::max<u32>(terms, 64 / QGL));
            hipLaunchKernelGGL(k_qreduce_bits<KF>, dim3(nres), dim3(np * QGL), np * QGL * sizeof(Fp), st,
                               s > 0 ? (const Xyzz<F>*)R : (const Xyzz<F>*)wc->buckets.p, (const Xyzz<F>*)C, cb, s, (Xyzz<F>*)wc->wins.p,
                               (const u32*)c->offs.p + G, (u32*)((Xyzz<F>*)wc->wins.p + (pl.sets > 1 ? nres + (size_t)pl.sets : (size_t)nres)),
                               (const Xyzz<F>*)nullptr);
        } else if (rp.small) {
            hipLaunchKernelGGL(k_reduce_small<KF>, dim3(nres), dim3(RED_SUM_LANES * LN), RED_SUM_LANES * sizeof(Xyzz<F>), st,
                               (const Xyzz<F>*)wc->buckets.p, pl.NB, rp.njobs, (Xyzz<F>*)wc->wins.p);
        } else {
            hipLaunchKernelGGL(k_reduce_l1<KF>, dim3(nblocks((size_t)nseg_total * LN)), dim3(256), 0, st, (const Xyzz<F>*)wc->buckets.p,
                               nseg_total, accs, runs);
            if (pl.hybrid && rp.segs >= 64) {
                // Behind the 8-bucket running sums the parallelism is gone (2^16 segments at 2^20 points) and what is left is
                // depth: sum = sum_s acc_s + 8 sum_s s run_s, the second sum by rows, columns and bits of s on quads (qtail.hpp)
                // -- the same 4 + kb results [A, W_0, W_1, ..] the pyramid leaves ([A, Q0, Q1, Q2, T_0, ..]), 9 + 8 quad additions
                // deep instead of 7 + 14 + 6 one-lane ones: 0.45 -> 0.13 ms of a lone 2^20-point sum.
                constexpr u32 QGL = QTraits<KF>::GL, QNP = 256 / QGL;  // 256-thread blocks: one fits where ONE accumulation block retired
                int cb2 = 0;
                while ((1u << cb2) < rp.segs) cb2++;
                const int s2 = (cb2 + 1) / 2;
                const u32 rows2 = 1u << (cb2 - s2), cols2 = 1u << s2;
                Xyzz<F>* R_run = lvl;
                Xyzz<F>* C_run = R_run + (size_t)pl.sets * rows2;
                Xyzz<F>* R_acc = C_run + (size_t)pl.sets * cols2;
                u32 np = std::min<u32>(QNP, std::max(rows2, cols2));
                while (np > 64 / QGL && (u64)np * QGL * (2 * rows2 + cols2) * (u32)pl.sets > 2048ull * 64) np >>= 1;
                hipLaunchKernelGGL(k_qreduce_rowcol<KF>, dim3((2 * rows2 + cols2) * (u32)pl.sets), dim3(np * QGL), np * QGL * sizeof(Fp), st,
                                   (const Xyzz<F>*)runs, cb2, s2, R_run, C_run, (const Xyzz<F>*)accs, R_acc);
                const u32 np2 = std::min<u32>(QNP, std::max<u32>(std::max(rows2, cols2 / 2), 64 / QGL));
                hipLaunchKernelGGL(k_qreduce_bits<KF>, dim3(nres), dim3(np2 * QGL), np2 * QGL * sizeof(Fp), st, (const Xyzz<F>*)R_run,
                                   (const Xyzz<F>*)C_run, cb2, s2, (Xyzz<F>*)wc->wins.p, (const u32*)nullptr, (u32*)nullptr, (const Xyzz<F>*)R_acc);
            } else {
            hipLaunchKernelGGL(k_reduce_pyr<KF>, dim3(nblocks(5 * (size_t)per_role * LN)), dim3(256), 0, st, (const Xyzz<F>*)accs,
                               (const Xyzz<F>*)runs, rp.segs, rp.m, (u32)pl.sets, lvl);
            hipLaunchKernelGGL(k_reduce_sum<KF>, dim3(nres * rp.nblk), dim3(RED_SUM_LANES * LN), RED_SUM_LANES * sizeof(Xyzz<F>), st,
                               (const Xyzz<F>*)lvl, rp.m, (u32)pl.sets, rp.njobs, rp.nblk, rp.nblk > 1 ? pieces : (Xyzz<F>*)wc->wins.p);
            if (rp.nblk > 1)
                hipLaunchKernelGGL(k_reduce_fin<KF>, dim3(nres), dim3(RED_SUM_LANES * LN), RED_SUM_LANES * sizeof(Xyzz<F>), st,
                                   (const Xyzz<F>*)pieces, rp.nblk, (Xyzz<F>*)wc->wins.p);
            }
        }
        if (pl.sets > 1)  // per-set weights on the device; the set sums follow the partial results in `wins`
            hipLaunchKernelGGL(k_reduce_weights<KF>, dim3((unsigned)pl.sets), dim3(RED_SUM_LANES * LN), RED_SUM_LANES * sizeof(Xyzz<F>), st,
                               (const Xyzz<F>*)wc->wins.p, rp.njobs, (rp.small && !pl.qtail) ? 1 : 0, pl.qtail ? 0 : RED_SEG_LOG,
                               (Xyzz<F>*)wc->wins.p + nres);
    }
    PS_STAGE_MARK();  // 7: after reduction
    HIP_TRY(hipGetLastError());
    // one set: its partial results (the host applies the weights); several: the set sums
    HIP_TRY(hipMemcpyAsync((char*)c->h_pinned + (size_t)slot * PS_PINNED_SLOT, (const Xyzz<F>*)wc->wins.p + (pl.sets > 1 ? nres : 0),
                           sizeof(Xyzz<F>) * (pl.sets > 1 ? (size_t)pl.sets : (size_t)nres) + (pl.qtail ? 4 : 0), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(wc->ev_tail_done, st));
    wc->tail_used = true;
    return PS_OK;
}
#undef PS_STAGE_MARK

// Plan of a sum on context `c`.  Entry offsets, ranks and totals are 32-bit (k_scan_*, k_scatter, the sorted
// list): a sum whose W * n digits would not fit is refused rather than wrapped (W = 16 windows: n < 2^28).
static bool table_usable(const ps_ctx* c, const ps_points* pts, size_t n, int max_bits) {
    const Storage* st = pts->st;
    // a forced window size means "the plain path with this c" (A/B runs); entries keep 26 bits for the index
    return st->table.load(std::memory_order_acquire) && !c->forced_c && n < (1ull << ENTRY_W_SHIFT) && max_bits / st->table_c + 1 <= st->table_W;
}
// Behind the accumulation the chip is mostly idle and a sum's time is the depth of its dependency chain, so the tail runs as
// trees of lane-cooperative additions (qtail.hpp) wherever that costs little lane-time.  Short sums (fewer than
// PS_QTAIL_MAX_ENTRIES digits) also get slices as short as keeps one wave per SIMD busy and a tree fix-up.
// forced: 0 automatic, 1 the chains of round 2 throughout, 2 the trees wherever they apply.
#ifndef PS_QTAIL_MAX_ENTRIES
#define PS_QTAIL_MAX_ENTRIES (1ull << 21)
#endif
static void msm_plan_lpb(MsmPlan& pl, size_t n, int group) {
    // quads per cut bucket: enough for the average span, within one 256-thread block and 2^17 lanes in all
    const u32 gl = group == PS_G1 ? 4 : 8;
    const u64 total = (u64)pl.W * n;
    const u64 span = total / (pl.G * (u64)pl.M) + 1;  // slices an average bucket touches, rounded up
    if (pl.G * (u64)gl > (1ull << 17)) { pl.lpb = 0; return; }  // too many buckets for a quad each: one thread per bucket (k_fixup)
#if defined(PS_TAIL_TUNE)
    if (pl.busy) if (const char* e = getenv("PS_T_LPB_BUSY")) if (atoi(e) == 0 && pl.G >= 4096) { pl.lpb = 0; return; }
#endif
    u32 lpb = 1;
    while (lpb < span && lpb * 2 * gl <= 256 && pl.G * (u64)(lpb * 2) * gl <= (1ull << 17)) lpb *= 2;  // one round of blocks (2048 waves)
    pl.lpb = (int)lpb;
}
static void msm_plan_tail(MsmPlan& pl, size_t n, int group, int forced, bool busy) {
    const u64 total = (u64)pl.W * n;
    // the reduction: quads over the buckets themselves while all sets together hold <= 2^16 buckets (2 quad additions per
    // bucket: a few per cent of the accumulation's lane-time even with sums in flight); beyond, running sums first
    pl.qtail = forced ? forced == 2 : pl.G <= (1ull << 16);
    pl.hybrid = forced != 1;
    pl.shortsum = forced ? (forced == 2 && total < PS_QTAIL_MAX_ENTRIES) : total < PS_QTAIL_MAX_ENTRIES;
    pl.lpb = 0;
    pl.rc_s = 0;
    pl.busy = busy;
    if (pl.qtail) {
        const u32 np = 512 / (group == PS_G1 ? 4 : 8);  // quads of a reduction block
        const int cb = pl.c - 1;
        pl.rc_s = (pl.NB / 2 <= 2 * np) ? 0 : (cb + 1) / 2;  // small sets: bit sums straight from the buckets
    }
    if (pl.shortsum) {
        // one wave per SIMD when the sum is alone (the shortest chain that still fills the chip); two when other sums are in
        // flight (2^16 points, A/B on one box: slices of 16 / 8 entries 0.504 / 0.519 ms alone, 0.348 / 0.328 in flight)
        u64 threads = (group == PS_G1 ? (1ull << 16) : (1ull << 15)) << (busy ? 1 : 0);
#if defined(PS_TAIL_TUNE)
        if (busy) if (const char* e = getenv("PS_T_MSHIFT")) threads = (group == PS_G1 ? (1ull << 16) : (1ull << 15)) << atoi(e);
#endif
        int M = 2;
        while (M < 32 && total / (u64)M > threads) M *= 2;
        pl.M = M;
    }
}

#ifndef PS_TABLE_COST_MARGIN
#define PS_TABLE_COST_MARGIN 0.9  // the plain plan must be this much cheaper in the model to replace a table that exists
#endif
#ifndef PS_VIEW_TABLE_MIN_POINTS
#define PS_VIEW_TABLE_MIN_POINTS 1024  // index-range views shorter than this keep the plan of the whole array (latency-bound anyway)
#endif
static inline double table_plan_cost(size_t n, int max_bits, int wbits) {
    return (double)n * (max_bits / wbits + 1) * 10.0 + (double)(1u << (wbits - 1)) * 42.0;
}
static int msm_plan_checked(ps_ctx* c, const ps_points* const* pts, size_t k, size_t n, int max_bits, MsmPlan* out) {
    // The window-table plan needs every array of the call to carry a table for the same window size
    bool tab = k > 0;
    for (size_t i = 0; i < k && tab; i++)
        tab = table_usable(c, pts[i], n, max_bits) && pts[i]->st->table_c == pts[0]->st->table_c;
    int tab_c = tab ? pts[0]->st->table_c : 0;
    // Index-range views of arrays that carry a table for the WHOLE allocation (the shares of a sharded prover: an eighth of
    // a 2^20-point key over its 20-bit table reduces 2^19 buckets for 2^17 points -- 9.2 ms per share where the work is
    // ~2 ms, tools/g16_shares.py): a table of the view's own, window chosen for ITS length, built on first use under the
    // table budget (view_table_ensure).  Only where the model says it pays, and only for arrays whose owner asked for tables.
    if (k > 0 && !c->forced_c && c->use_tables && n >= PS_VIEW_TABLE_MIN_POINTS && n < (1ull << ENTRY_W_SHIFT)) {
        const int vc = table_window_for(n);
        bool views = true;
        for (size_t i = 0; i < k && views; i++) {
            const Storage* st = pts[i]->st;
            views = n < st->count && (st->table.load(std::memory_order_acquire) || view_table_find(st, pts[i]->first, n, 0));
        }
        if (views && (!tab || table_plan_cost(n, max_bits, vc) < PS_TABLE_COST_MARGIN * table_plan_cost(n, max_bits, tab_c))) {
            bool all = true;
            for (size_t i = 0; i < k && all; i++) {
                const ViewTable* v = view_table_ensure(c, pts[i]);
                all = v && v->c == vc;
            }
            if (all) { tab = true; tab_c = vc; }
        }
    }
    MsmPlan pl = tab ? msm_plan_table(n, max_bits, tab_c) : msm_plan(n, max_bits, c->forced_c);
    if (tab && k == 1) {
        // Short scalars (an int64 witness over a table built for 20-bit windows: 4 windows, but 2^19 buckets to reduce) are
        // cheaper on the plain plan with its small windows -- same cost model as msm_plan: 10 field products per bucket
        // addition, 42 per bucket of the reduction (measured at 2^20 points, 40-bit values: 1.28 ms over the table, 1.04 plain).
        // One array only: for the seven sums of PHGR13 over ONE sort the tables stay (2^20 booleanity gates: 10.3 ms against
        // 12.0 with the plain plan -- five bucket sets per array, a fifth more digits to sort).
        const MsmPlan plain = msm_plan(n, max_bits, 0);
        const double cost_tab = (double)n * pl.W * 10.0 + (double)pl.G * 42.0;
        const double cost_plain = (double)n * plain.W * 10.0 + (double)plain.G * 42.0;
        if (cost_plain < PS_TABLE_COST_MARGIN * cost_tab) pl = plain;
    }
    int plan_group = PS_G1;  // the group the tail is sized for: G2 as soon as one array of the call is (its points take twice the lanes)
    for (size_t i = 0; i < k; i++)
        if (pts[i]->group == PS_G2) plan_group = PS_G2;
    msm_plan_tail(pl, n, plan_group, c->forced_tail, c->q_len > 0);
    if (c->forced_slice) pl.M = c->forced_slice;
    if (pl.shortsum) msm_plan_lpb(pl, n, plan_group);
    if ((u64)pl.W * (u64)n >= (1ull << 32))
        return fail(PS_ERR_ARG, "MSM too long: windows x length = " + std::to_string((u64)pl.W * (u64)n) +
                                    " digits do not fit the 32-bit sort offsets (split the sum, e.g. ps_points_slice)");
    *out = pl;
    return PS_OK;
}

#ifndef PS_MULTI_CHAIN_MIN_ENTRIES
#define PS_MULTI_CHAIN_MIN_ENTRIES (1ull << 40)  // the point passes of ONE multi-sum: never chained (msm_multi_points)
#endif
#ifndef PS_CHAIN_MIN_ENTRIES
#define PS_CHAIN_MIN_ENTRIES (1ull << 21)  // sums with fewer digits than this are not chained behind the previous accumulation
#endif
static int msm_launch_any(ps_ctx* wc, const ps_points* pts, const ps_scalars* sc, const MsmPlan& pl, hipEvent_t wait_acc,
                          hipEvent_t acc_done, bool inline_tail) {
    hipStream_t ss = inline_tail ? wc->stream : wc->tail;  // a lone short sum keeps one stream
    int rc = msm_sort(wc, sc, pl, wc->timing, ss);
    if (rc) return rc;
    if (ss != wc->stream) {
        HIP_TRY(hipEventRecord(wc->ev_sort_local, ss));
        HIP_TRY(hipStreamWaitEvent(wc->stream, wc->ev_sort_local, 0));
    }
    rc = pts->group == PS_G1 ? msm_points_t<Fp>(wc, wc, pts, sc->n, pl, wc->timing, 0, wait_acc, acc_done, inline_tail)
                             : msm_points_t<Fp2>(wc, wc, pts, sc->n, pl, wc->timing, 0, wait_acc, acc_done, inline_tail);
    wc->ev_valid = wc->timing && !rc;
    return rc;
}

template <class F>
static void msm_fold_host(ps_ctx* c, const MsmPlan& pl, int slot, uint8_t* out) {
    typedef typename HostField<F>::type H;  // six 64-bit words: the chain below costs the CPU 0.05 ms instead of 0.35
    const Xyzz<F>* res = (const Xyzz<F>*)((const char*)c->h_pinned + (size_t)slot * PS_PINNED_SLOT);
    Xyzz<H> acc = xyzz_identity<H>();
    if (pl.sets == 1) {
        // the reduction leaves A, Q0, Q1, Q2, T_0 .. T_{kb-1} (msm.hpp section 6): sum = A + 8 (Q0 + 2 Q1 + 4 Q2 + 8 sum_k 2^k T_k)
        const ReducePlan rp = reduce_plan(pl.NB);
        if (pl.qtail) {  // [S, W_0 .. W_{c-2}] of k_qreduce_bits: sum = S + sum_k 2^k W_k
            for (int j = pl.c - 1; j >= 1; j--) {
                acc = xyzz_dbl<H>(acc);
                xyzz_add<H>(acc, xyzz_to_host<F>(res[j]));
            }
            xyzz_add<H>(acc, xyzz_to_host<F>(res[0]));
        } else if (rp.small) {  // V_0 .. V_{c-1} of k_reduce_small: sum = sum_k 2^k V_k
            for (int j = (int)rp.njobs - 1; j >= 0; j--) {
                acc = xyzz_dbl<H>(acc);
                xyzz_add<H>(acc, xyzz_to_host<F>(res[j]));
            }
        } else {
            for (int j = (int)rp.njobs - 1; j >= 1; j--) {
                acc = xyzz_dbl<H>(acc);
                xyzz_add<H>(acc, xyzz_to_host<F>(res[j]));
            }
            for (int i = 0; i < RED_SEG_LOG; i++) acc = xyzz_dbl<H>(acc);
            xyzz_add<H>(acc, xyzz_to_host<F>(res[0]));
        }
    } else {
        for (int w = pl.sets - 1; w >= 0; w--) {  // Horner over the window sums (k_reduce_weights made them)
            for (int i = 0; i < pl.c; i++) acc = xyzz_dbl<H>(acc);
            xyzz_add<H>(acc, xyzz_to_host<F>(res[w]));
        }
    }
    write_affine(out, acc);
}

// digits that were not zero (introspection): behind the results of slot 0 for a short sum, else in the pinned buffer's tail
static u32 msm_entries(const ps_ctx* c, const MsmPlan& pl, int group) {
    if (!pl.qtail) return *(const u32*)((const char*)c->h_pinned + c->h_pinned_cap - 8);
    const size_t pt = group == PS_G1 ? sizeof(Xyzz<Fp>) : sizeof(Xyzz<Fp2>);
    return *(const u32*)((const char*)c->h_pinned + pt * (pl.sets > 1 ? (size_t)pl.sets : (size_t)pl.c));
}

static void write_identity(int group, uint8_t* out) {  // zero.Clone(), algebra.go:353
    memset(out, 0, wire_bytes(group));
    out[0] = 0x40;
}

// k sums over one scalar vector.  The sort runs on ring[0]'s stream; the point passes rotate over the workspaces of
// `ring` (entries may coincide when k is small), accumulation i+1 chained behind accumulation i so that it runs beside
// the fix-up and reduction of sum i.  Four workspaces: a tail that shares the chip with the following accumulations is
// starved of CUs and may outlast two of them (measured: 8 ms for a G2 tail under three G1 accumulations), and the sum
// that reuses its buffers must not wait for it.  Events live on `c`.
constexpr int PS_MULTI_RING = 4;
static int msm_multi_sort(ps_ctx* c, ps_ctx* const* ring, const ps_scalars* sc, const MsmPlan& pl, bool fork) {
    int rc;
    ps_ctx* w0 = ring[0];
    if (fork) {  // inputs prepared on the context stream are visible to the stream the sort runs on
        if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(w0->tail, c->ev_fork, 0));
    }
    if ((rc = msm_sort(w0, sc, pl, false, w0->tail))) return rc;  // high priority: see msm_sort
    w0->ev_valid = false;
    if (!c->ev_sorted) HIP_TRY(hipEventCreateWithFlags(&c->ev_sorted, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->ev_sorted, w0->tail));
    for (int j = 0; j < PS_MULTI_RING; j++) {
        bool seen = false;
        for (int i = 0; i < j; i++) seen = seen || ring[i] == ring[j];
        if (!seen) HIP_TRY(hipStreamWaitEvent(ring[j]->stream, c->ev_sorted, 0));
    }
    return PS_OK;
}
// first_wait: a point pass launched before this call that the first accumulation is chained behind
static int msm_multi_points(ps_ctx* c, ps_ctx* const* ring, const ps_points* const* pts, size_t k, const ps_scalars* sc,
                            const MsmPlan& pl, hipEvent_t first_wait) {
    ps_ctx* w0 = ring[0];
    for (size_t i = 0; i < k; i++) {
        if (!c->ev_multi[i]) HIP_TRY(hipEventCreateWithFlags(&c->ev_multi[i], hipEventDisableTiming));
        if (!c->ev_acc[i]) HIP_TRY(hipEventCreateWithFlags(&c->ev_acc[i], hipEventDisableTiming));
        ps_ctx* wc = ring[i % PS_MULTI_RING];
        // The accumulations of a multi-sum are NOT chained behind each other (round 4; PS_MULTI_CHAIN_MIN_ENTRIES): all of them
        // are enqueued at once, on rotating workspaces, and when they share the chip the workgroups of the next one fill the
        // SIMDs the previous one's last, ragged round leaves idle.  A/B on one box, PHGR13Prove at 2^20 constraints (seven
        // sums over one sort): 30.5-31.8 ms chained, 27.7-28.7 unchained.  (Independent sums in flight, msm_launch_impl, keep
        // the chain: 2^20-point G1 sums 2.97 ms per step chained, 3.03-3.08 unchained.)
        const bool chained = (u64)pl.W * sc->n >= PS_MULTI_CHAIN_MIN_ENTRIES;
        hipEvent_t wait = !chained ? nullptr : i ? c->ev_acc[i - 1] : first_wait;
        int rc = pts[i]->group == PS_G1 ? msm_points_t<Fp>(w0, wc, pts[i], sc->n, pl, false, (int)i, wait, c->ev_acc[i])
                                        : msm_points_t<Fp2>(w0, wc, pts[i], sc->n, pl, false, (int)i, wait, c->ev_acc[i]);
        if (rc) {
            for (int j = 0; j < PS_MULTI_RING; j++) (void)ps_ctx_sync(ring[j]);
            return rc;
        }
        HIP_TRY(hipEventRecord(c->ev_multi[i], wc->tail));
    }
    return PS_OK;
}
// the workspaces of a multi-sum on context c: `self` puts the context itself first (ps_msm_multi), otherwise only workers
static int msm_multi_ring(ps_ctx* c, bool self, size_t k, ps_ctx** ring) {
    ps_ctx** slots[4] = {&c->aux, &c->pipe, &c->pipe2, &c->pipe3};
    int have = 0;
    if (self) ring[have++] = c;
    for (int j = 0; have < PS_MULTI_RING && (size_t)have < std::max<size_t>(k, 1); j++) {
        if (!*slots[j]) {
            int rc = ps_ctx_create(c->device, slots[j]);
            if (rc) return rc;
        }
        ring[have++] = *slots[j];
    }
    for (int j = have; j < PS_MULTI_RING; j++) ring[j] = ring[j % have];
    return PS_OK;
}

// fold result i on the host while the GPU works on i+1...; out == nullptr just drains
static int msm_multi_finish(ps_ctx* c, ps_ctx* w0, const ps_points* const* pts, size_t k, const MsmPlan& pl, uint8_t* const* out) {
    for (size_t i = 0; i < k; i++) {
        HIP_TRY(hipEventSynchronize(c->ev_multi[i]));
        if (!out) continue;
        if (pts[i]->group == PS_G1) msm_fold_host<Fp>(w0, pl, (int)i, out[i]);
        else msm_fold_host<Fp2>(w0, pl, (int)i, out[i]);
    }
    c->last_info = ps_msm_info{pl.c, pl.W, msm_entries(w0, pl, pts[0]->group), pl.G, pl.M, pl.table ? 1 : 0};
    return PS_OK;
}

static int msm_multi_check(const char* who, const ps_points* const* pts, size_t k, const ps_scalars* sc) {
    if (k > PS_MSM_MULTI_MAX) return fail(PS_ERR_ARG, std::string(who) + ": more than PS_MSM_MULTI_MAX point arrays");
    for (size_t i = 0; i < k; i++) {
        if (!pts[i]) return fail(PS_ERR_ARG, std::string(who) + ": NULL argument");
        if (pts[i]->n != sc->n)  // algebra.go:350-352
            return fail(PS_ERR_LENGTH, "mismatch of length between poly " + std::to_string(sc->n) + " and blinded eval points " +
                                           std::to_string(pts[i]->n));
    }
    return PS_OK;
}

extern "C" int ps_msm_multi(ps_ctx* c, const ps_points* const* pts, size_t k, const ps_scalars* sc, uint8_t* const* out) {
    if (!c || !pts || !sc || !out) return fail(PS_ERR_ARG, "ps_msm_multi: NULL argument");
    if (k == 0) return PS_OK;
    int rc = msm_multi_check("ps_msm_multi", pts, k, sc);
    if (rc) return rc;
    for (size_t i = 0; i < k; i++)
        if (!out[i]) return fail(PS_ERR_ARG, "ps_msm_multi: NULL argument");
    if (c->pending) return fail(PS_ERR_ARG, "ps_msm_multi: an MSM is already pending on this context");
    if (sc->n == 0) {
        for (size_t i = 0; i < k; i++) write_identity(pts[i]->group, out[i]);
        return PS_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    MsmPlan pl;
    if ((rc = msm_plan_checked(c, pts, k, sc->n, sc->max_bits, &pl))) return rc;
    if (c->aux && c->aux->pending) return fail(PS_ERR_ARG, "ps_msm_multi: the auxiliary context is busy");
    ps_ctx* ring[PS_MULTI_RING];
    if ((rc = msm_multi_ring(c, true, k, ring))) return rc;
    if ((rc = msm_multi_sort(c, ring, sc, pl, true))) return rc;
    if ((rc = msm_multi_points(c, ring, pts, k, sc, pl, nullptr))) return rc;
    return msm_multi_finish(c, c, pts, k, pl, out);
}

// allow_self = false keeps the context's own stream out of the rotation (the provers run the quotient on it while
// the sums run on the three worker contexts pipe, pipe2, aux)
static int msm_launch_impl(ps_ctx* c, const ps_points* pts, const ps_scalars* sc, bool allow_self) {
    if (!c || !pts || !sc) return fail(PS_ERR_ARG, "ps_msm: NULL argument");
    if (pts->n != sc->n)  // algebra.go:350-352
        return fail(PS_ERR_LENGTH, "mismatch of length between poly " + std::to_string(sc->n) + " and blinded eval points " +
                                       std::to_string(pts->n));
    if (c->q_len == PS_MSM_QUEUE) return fail(PS_ERR_ARG, "ps_msm_launch: PS_MSM_QUEUE sums are already pending on this context");
    HIP_TRY(hipSetDevice(c->device));
    // a workspace no pending sum is using: the context itself first
    static_assert(PS_MSM_QUEUE >= 1 && PS_MSM_QUEUE <= 6, "workspaces exist for at most six pending sums");
    ps_ctx** slots[7] = {nullptr, &c->pipe, &c->pipe2, &c->aux, &c->pipe3, &c->pipe4, &c->pipe5};
    ps_ctx* wc = nullptr;
    for (int w = allow_self ? 0 : 1; w < (allow_self ? PS_MSM_QUEUE : PS_MSM_QUEUE + 1) && !wc; w++) {
        ps_ctx* cand = w == 0 ? c : *slots[w];
        bool busy = false;
        for (int j = 0; j < c->q_len; j++) busy = busy || (cand && c->q[(c->q_head + j) % PS_MSM_QUEUE].wc == cand);
        if (busy) continue;
        if (!cand) {
            int rc = ps_ctx_create(c->device, slots[w]);
            if (rc) return rc;
            cand = *slots[w];
        }
        wc = cand;
    }
    if (!wc) return fail(PS_ERR_ARG, "ps_msm_launch: no free workspace");
    if (wc != c) {
        wc->timing = c->timing;
        wc->forced_c = c->forced_c;
        wc->forced_slice = c->forced_slice;
        wc->forced_tail = c->forced_tail;
    }
    // Inputs enqueued on the context stream before the FIRST launch of a burst are visible to the worker
    // streams: the fork event is recorded while the queue is empty, ahead of that launch's own kernels
    // (recording it later would order a worker's sort behind the running accumulation).  Arrays that a
    // producer left in flight on some stream (ps_points_from_scalars, ps_scalars_from_device_be32) carry
    // their own ready event, which msm_sort / msm_points_t make the worker stream wait for -- so inputs
    // produced between two launches of a burst are ordered as well.
    if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    if (c->q_len == 0) HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    ps_ctx::PendingMsm& e = c->q[(c->q_head + c->q_len) % PS_MSM_QUEUE];
    if (sc->n == 0) {
        e = {pts->group, MsmPlan{}, wc};
    } else {
        MsmPlan pl;
        int rc0 = msm_plan_checked(c, &pts, 1, sc->n, sc->max_bits, &pl);
        if (rc0) return rc0;
        if (wc != c) HIP_TRY(hipStreamWaitEvent(wc->stream, c->ev_fork, 0));
        const bool lone = pl.shortsum && c->q_len == 0;
        if (!lone) HIP_TRY(hipStreamWaitEvent(wc->tail, c->ev_fork, 0));  // the sort runs on the workspace's high-priority stream
        // chain the accumulations: this one starts when the previously launched one is done
        // ... unless the sum is short: accumulations of a few hundred thousand entries do not fill the chip, and chained
        // they are three launch latencies in a row (A/B on one box: Groth16 on 2^10 constraints 1.62 -> 1.54 ms, PHGR13 on 2^14 4.25 -> 3.9 ms;
        // below that the host's ~25 launches per sum are the bound)
        const bool chained = (u64)pl.W * sc->n >= PS_CHAIN_MIN_ENTRIES;
        hipEvent_t wait = (chained && c->last_chain && c->last_chain != wc) ? c->last_chain->ev_acc_local : nullptr;
        int rc = msm_launch_any(wc, pts, sc, pl, wait, nullptr, lone);
        if (rc) return rc;
        c->last_chain = wc;
        e = {pts->group, pl, wc};
    }
    c->q_len++;
    c->pending = true;
    return PS_OK;
}
extern "C" int ps_msm_launch(ps_ctx* c, const ps_points* pts, const ps_scalars* sc) { return msm_launch_impl(c, pts, sc, true); }

extern "C" int ps_msm_finish(ps_ctx* c, uint8_t* out) {
    if (!c || !out) return fail(PS_ERR_ARG, "ps_msm_finish: NULL argument");
    if (c->q_len == 0) return fail(PS_ERR_ARG, "ps_msm_finish: nothing pending");
    const ps_ctx::PendingMsm e = c->q[c->q_head];
    c->q_head = (c->q_head + 1) % PS_MSM_QUEUE;
    c->q_len--;
    c->pending = c->q_len > 0;
    const MsmPlan& pl = e.plan;
    if (pl.W == 0) {  // empty sum: the identity
        write_identity(e.group, out);
        return PS_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(e.wc->ev_tail_done));  // the copy of the results: the last thing the sum enqueued (tail stream, or its own stream for a lone short sum)
    if (e.group == PS_G1) msm_fold_host<Fp>(e.wc, pl, 0, out);
    else msm_fold_host<Fp2>(e.wc, pl, 0, out);
    c->last_info = ps_msm_info{pl.c, pl.W, msm_entries(e.wc, pl, e.group), pl.G, pl.M, pl.table ? 1 : 0};
    c->last_timed = e.wc;
    return PS_OK;
}

extern "C" int ps_msm(ps_ctx* c, const ps_points* pts, const ps_scalars* sc, uint8_t* out) {
    if (c && c->q_len) return fail(PS_ERR_ARG, "ps_msm: sums are pending on this context (ps_msm_finish them first)");
    int rc = ps_msm_launch(c, pts, sc);
    if (rc) return rc;
    return ps_msm_finish(c, out);
}
// Seam S1 (Poly.BlindEval, algebra.go:348-359, as the shim calls it): the scalars cross PCIe inside the call -- 32 bytes each
// from pageable memory, ~0.6 ms per 2^20 at the link's rate, and nothing of the sum can start before its digits are sorted, so
// that time does not hide.  What the call does NOT pay any more (round 4): a hipMalloc and a hipFree of the converted vector
// (the free synchronises the device), and a second synchronisation -- ps_msm's own wait for the result covers the copy.
extern "C" int ps_msm_be32(ps_ctx* c, const ps_points* pts, const uint8_t* be32, size_t n, uint8_t* out) {
    if (!c || !pts || !out || (n && !be32)) return fail(PS_ERR_ARG, "ps_msm_be32: NULL argument");
    if (c->q_len) return fail(PS_ERR_ARG, "ps_msm: sums are pending on this context (ps_msm_finish them first)");
    if (n >= (1ull << 31)) return fail(PS_ERR_ARG, "vector too long");
    HIP_TRY(hipSetDevice(c->device));
    int rc = c->staging.ensure(32 * n + 32);
    if (rc) return rc;
    ps_scalars* s = nullptr;
    if ((rc = upload_vector(c, n, &s))) return rc;
    // (no sum is pending and every earlier call returned after its result: the staging buffer and the vector are free)
    if (n) {
        HIP_TRY(hipMemcpyAsync(c->staging.p, be32, 32 * n, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_scalars_from_be32, dim3(nblocks(n)), dim3(256), 0, c->stream, (const uint8_t*)c->staging.p, (u32)n, (u32*)s->st->p);
        HIP_TRY(hipGetLastError());
    }
    if (storage_mark_ready(s->st, c->stream)) return fail(PS_ERR_HIP, "ps_msm_be32: event record failed");
    rc = ps_msm(c, pts, s, out);
    if (rc) (void)hipStreamSynchronize(c->stream);  // the caller's buffer is not read past return, error or not
    return rc;
}
extern "C" int ps_msm_i64(ps_ctx* c, const ps_points* pts, const int64_t* v, size_t n, uint8_t* out) {
    if (!c || !pts || !out || (n && !v)) return fail(PS_ERR_ARG, "ps_msm_i64: NULL argument");
    if (c->q_len) return fail(PS_ERR_ARG, "ps_msm: sums are pending on this context (ps_msm_finish them first)");
    if (n >= (1ull << 31)) return fail(PS_ERR_ARG, "vector too long");
    HIP_TRY(hipSetDevice(c->device));
    int rc = c->staging.ensure(8 * n + 32);
    if (rc) return rc;
    ps_scalars* s = nullptr;
    if ((rc = upload_vector(c, n, &s))) return rc;
    if (n) {
        HIP_TRY(hipMemcpyAsync(c->staging.p, v, 8 * n, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_scalars_from_i64, dim3(nblocks(n)), dim3(256), 0, c->stream, (const int64_t*)c->staging.p, (u32)n, (u32*)s->st->p);
        HIP_TRY(hipGetLastError());
    }
    bool any_neg = false;
    for (size_t i = 0; i < n; i++) any_neg |= v[i] < 0;
    s->max_bits = 64;
    s->neg_small = any_neg;
    if (storage_mark_ready(s->st, c->stream)) return fail(PS_ERR_HIP, "ps_msm_i64: event record failed");
    rc = ps_msm(c, pts, s, out);
    if (rc) (void)hipStreamSynchronize(c->stream);
    return rc;
}
extern "C" int ps_msm_last_info(ps_ctx* c, ps_msm_info* out) {
    if (!c || !out) return fail(PS_ERR_ARG, "NULL argument");
    *out = c->last_info;
    return PS_OK;
}
extern "C" int ps_msm_set_slice(ps_ctx* c, int entries) {
    if (!c || entries < 0 || entries > 4096) return fail(PS_ERR_ARG, "slice must be 0 (automatic) or 1..4096");
    c->forced_slice = entries;
    return PS_OK;
}
extern "C" int ps_msm_set_tail(ps_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 2) return fail(PS_ERR_ARG, "tail mode must be 0 (automatic), 1 (chains) or 2 (trees)");
    c->forced_tail = mode;
    return PS_OK;
}
extern "C" int ps_ctx_set_timing(ps_ctx* c, int enable) {
    if (!c) return fail(PS_ERR_ARG, "ctx is NULL");
    c->timing = enable != 0;
    c->ev_valid = false;
    for (ps_ctx* w : {c->pipe, c->pipe2, c->aux, c->pipe3, c->pipe4, c->pipe5})
        if (w) { w->timing = c->timing; w->ev_valid = false; }
    return PS_OK;
}
extern "C" int ps_msm_last_stage_ms(ps_ctx* c, float* ms) {
    if (!c || !ms) return fail(PS_ERR_ARG, "NULL argument");
    const ps_ctx* t = c->last_timed ? c->last_timed : c;  // the workspace of the sum finished last
    if (!t->ev_valid) return fail(PS_ERR_ARG, "no timed MSM on this context (ps_ctx_set_timing)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(t->ev[PS_MSM_STAGES]));
    for (int i = 0; i < PS_MSM_STAGES; i++) HIP_TRY(hipEventElapsedTime(&ms[i], t->ev[i], t->ev[i + 1]));
    return PS_OK;
}
extern "C" int ps_prove_last_phase_ms(ps_ctx* c, float* ms) {
    if (!c || !ms) return fail(PS_ERR_ARG, "NULL argument");
    for (int i = 0; i < PS_PROVE_PHASES; i++) ms[i] = c->phase_ms[i];
    return PS_OK;
}
extern "C" int ps_ctx_set_tables(ps_ctx* c, int enable) {
    if (!c) return fail(PS_ERR_ARG, "ctx is NULL");
    c->use_tables = enable != 0;
    return PS_OK;
}
// Issue rate of the multiplier the bucket additions are bound by: v_mad_u64_u32, eight independent chains per lane, two
// waves per SIMD on every CU (the recipe of tools/microbench_valu.hip), timed with HIP events on the context stream.
// ~1 ms; bench.py calls it so that the roofline's integer peak is measured in the same run, on the same chip and clocks.
__global__ void __launch_bounds__(256) k_microbench_mad(unsigned long long* out, u32 seed, int iters) {
    const u32 a = (seed * (threadIdx.x + 1)) | 1u, b = (seed ^ (threadIdx.x * 2654435761u)) | 1u;
    unsigned long long acc[8];
#pragma unroll
    for (int c = 0; c < 8; c++) acc[c] = a + c;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < 8; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
        }
    }
    unsigned long long s = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
extern "C" int ps_microbench_mad(ps_ctx* c, double* lane_mads_per_s) {
    if (!c || !lane_mads_per_s) return fail(PS_ERR_ARG, "ps_microbench_mad: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    const int blocks = prop.multiProcessorCount * 2, iters = 2048;  // 256 threads = one wave per SIMD: two waves per SIMD
    int rc = c->staging.ensure(sizeof(unsigned long long) * 256 * (size_t)blocks);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {  // the first repetition also warms the clocks up
        HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        hipLaunchKernelGGL(k_microbench_mad, dim3(blocks), dim3(256), 0, c->stream, (unsigned long long*)c->staging.p, 12345u + rep, iters);
        HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        HIP_TRY(hipEventSynchronize(c->ev[1]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        const double ops = (double)iters * 64.0 * 256.0 * blocks;
        if (ms > 0) best = std::max(best, ops / (ms * 1e-3));
    }
    c->ev_valid = false;
    *lane_mads_per_s = best;
    return PS_OK;
}
// ---- measurement only (tools/pair_add_probe.py; not in the public header): batched-AFFINE bucket additions, one round ----
// VERDICT r3 asked for the go/no-go on batched-affine accumulation to be settled by a measurement: the first round of a
// pairwise reduction of the sorted entries -- thread t adds the 16 adjacent pairs of its 32 sorted entries in affine
// coordinates (2M + 1S + the inversion's share per addition instead of 8M + 2S) with ONE field inversion per thread shared by
// Montgomery's trick: pass A gathers the points and leaves the running products of the denominators x2 - x1 in HBM, the
// inversion, pass B gathers the points AGAIN (32 points are 3.5 KB: no lane can hold them) and peels the inverses off.
// real_inv = 0 replaces the inversion by a copy: the bound a free inversion would give (trick + additions + traffic alone).
// Bucket boundaries and P +- P are ignored (a real kernel would do more work, not less); the sums are checked on a sample.
template <bool REAL_INV>
__global__ void __launch_bounds__(256, 2) k_pair_add_probe(const char* __restrict__ points, const u32* __restrict__ sorted,
                                                        const u32* __restrict__ offs, u32 G, u32 idx_mask, u64 w_stride, u32 pstride,
                                                        u32 T, Fp* __restrict__ prefix, Affine<Fp>* __restrict__ out) {
    const u32 E = offs[G];
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || (u64)32 * t + 32 > E) return;
    const u32 base = 32u * t;
    Fp run = fp_one();
#pragma unroll 1
    for (u32 i = 0; i < 16; i++) {
        const u32 e1 = sorted[base + 2 * i], e2 = sorted[base + 2 * i + 1];
        const Affine<Fp> a = ld_entry_point<Fp>(points, e1, idx_mask, w_stride, pstride), b = ld_entry_point<Fp>(points, e2, idx_mask, w_stride, pstride);
        Fp d = f_sub(b.x, a.x);
        if (f_is_zero(d)) d = fp_one();
        prefix[(size_t)i * T + t] = run;
        run = f_mul(run, d);
    }
    Fp inv = REAL_INV ? f_inv(run) : run;
#pragma unroll 1
    for (int i = 15; i >= 0; i--) {
        const u32 e1 = sorted[base + 2 * i], e2 = sorted[base + 2 * i + 1];
        Affine<Fp> a = ld_entry_point<Fp>(points, e1, idx_mask, w_stride, pstride), b = ld_entry_point<Fp>(points, e2, idx_mask, w_stride, pstride);
        if (e1 >> 31) a.y = f_neg(a.y);
        if (e2 >> 31) b.y = f_neg(b.y);
        Fp d = f_sub(b.x, a.x);
        if (f_is_zero(d)) d = fp_one();
        const Fp dinv = f_mul(inv, prefix[(size_t)i * T + t]);
        inv = f_mul(inv, d);
        const Fp lam = f_mul(f_sub(b.y, a.y), dinv);
        Affine<Fp> r;
        r.x = f_norm(f_sub(f_sub(f_sqr(lam), a.x), b.x));
        r.y = f_norm(f_sub(f_mul(lam, f_sub(a.x, r.x)), a.y));
        out[(size_t)i * T + t] = r;
    }
}
// ms[0]: average kernel time over 3 launches; sample: the first 32 sorted entries and the 16 sums of thread 0 (96-byte affine)
extern "C" int ps_debug_pair_add_probe(ps_ctx* c, const ps_points* pts, const ps_scalars* sc, int real_inv, float* ms, uint32_t* sample_entries,
                                       uint8_t* sample_sums, uint32_t* info) {
    if (!c || !pts || !sc || !ms || !sample_entries || !sample_sums || !info) return fail(PS_ERR_ARG, "ps_debug_pair_add_probe: NULL argument");
    if (pts->group != PS_G1 || pts->n != sc->n || c->q_len) return fail(PS_ERR_ARG, "ps_debug_pair_add_probe: one G1 array, no sums pending");
    HIP_TRY(hipSetDevice(c->device));
    MsmPlan pl;
    int rc = msm_plan_checked(c, &pts, 1, sc->n, sc->max_bits, &pl);
    if (rc) return rc;
    if (!pl.table) return fail(PS_ERR_ARG, "ps_debug_pair_add_probe: the array carries no window table (ps_points_precompute first)");
    if ((rc = msm_sort(c, sc, pl, false, c->stream))) return rc;
    TableRef tref{};
    if (!table_ref(pts, sc->n, pl.c, pl.W, &tref)) return fail(PS_ERR_ARG, "ps_debug_pair_add_probe: no table");
    HIP_TRY(hipStreamSynchronize(c->stream));
    u32 E = 0;
    HIP_TRY(hipMemcpy(&E, (const u32*)c->offs.p + pl.G, 4, hipMemcpyDeviceToHost));
    const u32 T = E / 32;
    if (!T) return fail(PS_ERR_ARG, "ps_debug_pair_add_probe: fewer than 32 entries");
    Fp* prefix = nullptr;
    Affine<Fp>* out = nullptr;
    HIP_TRY(hipMalloc((void**)&prefix, sizeof(Fp) * 16 * (size_t)T));
    if (hipMalloc((void**)&out, sizeof(Affine<Fp>) * 16 * (size_t)T) != hipSuccess) { (void)hipFree(prefix); return fail(PS_ERR_HIP, "hipMalloc"); }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const u32 idx_mask = (1u << ENTRY_W_SHIFT) - 1u, pstride = (u32)table_row_bytes(PS_G1);
    auto launch = [&]() {
        if (real_inv) hipLaunchKernelGGL(k_pair_add_probe<true>, dim3(nblocks(T)), dim3(256), 0, c->stream, tref.base, (const u32*)c->sorted.p,
                                         (const u32*)c->offs.p, (u32)pl.G, idx_mask, tref.stride, pstride, T, prefix, out);
        else hipLaunchKernelGGL(k_pair_add_probe<false>, dim3(nblocks(T)), dim3(256), 0, c->stream, tref.base, (const u32*)c->sorted.p,
                                (const u32*)c->offs.p, (u32)pl.G, idx_mask, tref.stride, pstride, T, prefix, out);
    };
    launch();  // warm-up
    HIP_TRY(hipEventRecord(e0, c->stream));
    for (int rep = 0; rep < 3; rep++) launch();
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float total = 0.f;
    HIP_TRY(hipEventElapsedTime(&total, e0, e1));
    ms[0] = total / 3.f;
    info[0] = E; info[1] = T; info[2] = (u32)pl.c; info[3] = (u32)pl.W;
    HIP_TRY(hipMemcpy(sample_entries, c->sorted.p, 4 * 32, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; i++) {
        Affine<Fp> a;
        HIP_TRY(hipMemcpy(&a, out + (size_t)i * T, sizeof a, hipMemcpyDeviceToHost));
        write_affine(sample_sums + 96 * i, xyzz_from_affine<Fp>(a.x, a.y));
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(prefix); (void)hipFree(out);
    return PS_OK;
}

extern "C" int ps_ctx_set_table_budget(ps_ctx* c, long long bytes) {
    if (!c) return fail(PS_ERR_ARG, "ctx is NULL");
    c->table_budget = bytes;
    return PS_OK;
}
extern "C" int ps_msm_set_window(ps_ctx* c, int bits) {
    if (!c || bits < 0 || bits > 20 || (bits > 0 && bits < 4)) return fail(PS_ERR_ARG, "window bits must be 0 or 4..20");
    c->forced_c = bits;
    return PS_OK;
}

#if defined(PS_NTT_TUNE)  // measurement builds only (tools/ntt_phases.py): the phase stamps of the traced NTT pass
extern "C" int ps_debug_ntt_trace(unsigned long long* out, int* meta) {
    if (!g_ntt_trace_buf) return PS_ERR_ARG;
    for (int i = 0; i < 6; i++) meta[i] = g_ntt_trace_meta[i];
    HIP_TRY(hipMemcpy(out, g_ntt_trace_buf, 4 * 8192 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PS_OK;
}
#endif
#include "prove.inc"
#include "lagrange.inc"
#include "pairing.inc"
