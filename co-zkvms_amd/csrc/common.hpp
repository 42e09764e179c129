// Shared host-side plumbing for libcozk: context, error handling, device buffers.
#pragma once
#ifdef COZK_COUNT_COPIES
#include <hip/hip_runtime.h>
#include <map>
#include <string>
#include <cstdio>
#include <mutex>
struct CozkCopyCounter {
    std::map<std::string, unsigned long> m;
    std::mutex mu;
    ~CozkCopyCounter() {
        for (auto& kv : m) fprintf(stderr, "[copies] %8lu  %s\n", kv.second, kv.first.c_str());
    }
};
inline CozkCopyCounter& cozk_copy_counter() { static CozkCopyCounter c; return c; }
inline void cozk_count_copy(const char* what, const char* f, int l) {
    auto& c = cozk_copy_counter();
    std::lock_guard<std::mutex> g(c.mu);
    const char* b = f;
    for (const char* p = f; *p; p++) if (*p == '/') b = p + 1;
    c.m[std::string(what) + " " + b + ":" + std::to_string(l)]++;
}
#define hipMemcpyAsync(...) (cozk_count_copy("memcpy", __FILE__, __LINE__), hipMemcpyAsync(__VA_ARGS__))
#define hipStreamSynchronize(...) (cozk_count_copy("sync", __FILE__, __LINE__), hipStreamSynchronize(__VA_ARGS__))
#define hipMemsetAsync(...) (cozk_count_copy("memset", __FILE__, __LINE__), hipMemsetAsync(__VA_ARGS__))
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <stdexcept>

#include "../../include/cozk.h"
#include "ec.hip.hpp"

struct CozkError : std::runtime_error {
    int code;
    CozkError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            char buf_[512];                                                                    \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                      \
            throw CozkError(e_ == hipErrorOutOfMemory ? COZK_ERR_OOM : COZK_ERR_HIP, buf_);    \
        }                                                                                      \
    } while (0)

#define COZK_REQUIRE(cond, msg)                                   \
    do {                                                          \
        if (!(cond)) throw CozkError(COZK_ERR_INVALID_ARG, msg);  \
    } while (0)

// A growable device scratch buffer (never shrinks; freed with the context).
// gives the parked blocks of the calling thread's context back to the driver (defined after cozk_ctx)
static inline void trim_current_pool();
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) {  // the caching allocator may be sitting on the memory we need
            (void)hipGetLastError();
            p = nullptr;
            trim_current_pool();
            HIP_TRY(hipMalloc(&p, bytes));
        }
        cap = bytes;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() { return reinterpret_cast<T*>(p); }
};

// what the sort phase of one MSM launch set produces (two of them: the sort of set k+1 overlaps the accumulation of set k)
struct MsmSortWs {
    DevBuf hist, off0, refs, offA, ptrs;
    uint32_t* max_pinned = nullptr;   // [0] fullest bucket (read back behind max_event); + 64: polynomial descriptors (H2D staging)
    hipEvent_t max_event = nullptr, done = nullptr, consumed = nullptr;
    void release() {
        for (DevBuf* b : {&hist, &off0, &refs, &offA, &ptrs}) b->release();
        if (max_pinned) (void)hipHostFree(max_pinned);
        max_pinned = nullptr;
        for (hipEvent_t* e : {&max_event, &done, &consumed}) {
            if (*e) (void)hipEventDestroy(*e);
            *e = nullptr;
        }
    }
};
struct MsmWorkspace {
    MsmSortWs sort[2];
    DevBuf offB, offC, partA, partB, bsum, chunk, grp, out, ptrs, exc, blk, perm, offP;
    hipEvent_t fork = nullptr;
    void release() {
        for (MsmSortWs& s : sort) s.release();
        for (DevBuf* b : {&offB, &offC, &partA, &partB, &bsum, &chunk, &grp, &out, &ptrs, &exc, &blk, &perm, &offP}) b->release();
        if (fork) (void)hipEventDestroy(fork);
        fork = nullptr;
    }
};

// Caching device allocator, one per context.  hipFree synchronises the whole device (~130 us measured) and the
// round loops create and drop hundreds of short-lived buffers per proof (eq tables, bound layers, fold outputs):
// freed blocks are parked per size class and handed out again.  Reuse is stream-ordered by construction: every
// user of a block enqueues on the context's one stream, so work on the old contents precedes work on the new.
struct DevPool {
    std::unordered_map<size_t, std::vector<void*>> parked;
    std::unordered_map<void*, size_t> live;  // block -> size class
    size_t parked_bytes = 0;
    static size_t size_class(size_t bytes) {  // four classes per octave: <= 25 % slack
        if (bytes <= 256) return 256;
        size_t top = (size_t)1 << (63 - __builtin_clzll((unsigned long long)bytes));
        size_t q = top >> 2;
        return (bytes + q - 1) / q * q;
    }
    void trim() {
        for (auto& kv : parked)
            for (void* p : kv.second) (void)hipFree(p);
        parked.clear();
        parked_bytes = 0;
    }
    void* alloc(size_t bytes) {
        size_t c = size_class(bytes);
        auto it = parked.find(c);
        if (it != parked.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            parked_bytes -= c;
            live[p] = c;
            return p;
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, c);
        if (e != hipSuccess) {  // give the parked blocks back and try once more
            (void)hipGetLastError();
            trim();
            HIP_TRY(hipMalloc(&p, c));
        }
        live[p] = c;
        return p;
    }
    // false: not one of ours
    bool release(void* p) {
        auto it = live.find(p);
        if (it == live.end()) return false;
        parked[it->second].push_back(p);
        parked_bytes += it->second;
        live.erase(it);
        return true;
    }
    void destroy() {
        trim();
        for (auto& kv : live) (void)hipFree(kv.first);
        live.clear();
    }
};

// HIP-event timing of named kernels for bench.py's per-kernel roofline entries (cozk_prof_read_kernel): a slot
// accumulates launches, elapsed ms on the context's stream and the ALGORITHMIC bytes of those launches
enum {
    COZK_PROF_EVAL_CHI = 0,   // k_poly_eval_chi
    COZK_PROF_LINCOMB,        // k_poly_lincomb
    COZK_PROF_BIND_CUBIC,     // k_layer_bind_cubic
    COZK_PROF_MSM_SCATTER,    // k_msm_scatter_lds
    COZK_PROF_LAYER_OUTPUT,   // k_layer_output (mul_vec local half)
    COZK_PROF_SLOTS
};
struct ProfSlot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double ms = 0.0;
    uint64_t launches = 0, alg_bytes = 0;
    void reset() {
        for (auto& pr : events) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        events.clear();
        ms = 0.0;
        launches = alg_bytes = 0;
    }
};

struct cozk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // side stream: MSM sort phase of the next launch set
    DevPool pool;
    std::string last_error;
    MsmWorkspace msm_ws;
    DevBuf scratch;       // small reductions (round evaluations, block partials)
    DevBuf scratch2;
    void* pinned = nullptr;  // pinned host staging for small D2H results
    size_t pinned_cap = 0;
    int resident_rounds = -1;        // cozk_ctx_set_resident_rounds: 1 on, 0 off, -1 automatic (ctx_resident_rounds_enabled)
    void* mailbox = nullptr;         // fine-grained pinned host memory shared with the resident round kernel
    uint32_t* round_flag = nullptr;  // pinned word a stream write bumps behind each round's finishing kernel
    uint32_t round_seq = 0;
    uint32_t armed_seq = 0;           // != 0: a finishing kernel will publish this sequence number itself (fetch_fe just waits for it)
    unsigned* finish_ticket = nullptr;  // device counter of the finishing kernel's workgroups (zero between launches)
    // timing of the dominant kernel (bench roofline): accumulated HIP-event time of the
    // bucket-accumulation launches on this stream
    bool prof_enabled = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    double prof_ms = 0.0;
    uint64_t prof_launches = 0;
    uint64_t prof_units = 0;  // point additions issued by those launches
    uint64_t prof_alg_bytes = 0;  // algorithmic bytes of those launches: n * (64 B base + scalar bytes) per MSM
    ProfSlot prof_slots[COZK_PROF_SLOTS];
    // native Rep3 ring (ring.hip): an ncclComm_t of ring_n ranks, this context being rank ring_rank
    void* ring_comm = nullptr;
    int ring_rank = 0, ring_n = 0;
    uint64_t ring_bytes = 0;
};

// brackets ONE kernel launch with a pair of events on `st` while profiling is enabled (no-op otherwise)
struct ProfScope {
    cozk_ctx* ctx;
    hipStream_t st;
    int slot;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(cozk_ctx* c, int slot_, uint64_t alg_bytes, hipStream_t stream = nullptr) : ctx(c), st(stream ? stream : c->stream), slot(slot_) {
        if (!ctx->prof_enabled) return;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            e0 = e1 = nullptr;
            return;
        }
        (void)hipEventRecord(e0, st);
        ctx->prof_slots[slot].launches += 1;
        ctx->prof_slots[slot].alg_bytes += alg_bytes;
    }
    ~ProfScope() {
        if (!e0) return;
        (void)hipEventRecord(e1, st);
        ctx->prof_slots[slot].events.push_back({e0, e1});
    }
};

struct cozk_bases {
    cozk_ctx* ctx;
    size_t n;           // number of SRS points
    int nwin;           // 16 if a window table was precomputed, else 1
    g1_affine* table;   // [nwin][n]: table[w][i] = 2^(16 w) * G_i   (table[0] = the points themselves)
    bool has_inf = true;  // some table entry is the point at infinity (set by build_window_table; the gather kernel then tests for it)
};

struct cozk_vec {       // device array of field elements / small scalars
    cozk_ctx* ctx;
    size_t n;           // element count
    int kind;           // COZK_SCALAR_*
    void* d;            // device pointer
    size_t bytes;
    bool owned;
};

static inline size_t scalar_kind_bytes(int kind) {
    switch (kind) {
        case COZK_SCALAR_FR: return 32;
        case COZK_SCALAR_U8: return 1;
        case COZK_SCALAR_U16: return 2;
        case COZK_SCALAR_U32: return 4;
        case COZK_SCALAR_U64: return 8;
        case COZK_SCALAR_I64: return 8;
        default: return 0;
    }
}

// registry of live contexts (an object may outlive its context in a host language with a GC: its free then falls
// back to hipFree) and the context of the ABI call running on this thread (for allocations made deep inside it)
inline std::mutex g_ctx_mu;
inline std::unordered_set<cozk_ctx*> g_live_ctx;
inline thread_local cozk_ctx* t_cur_ctx = nullptr;
static inline void trim_current_pool() {
    if (t_cur_ctx) t_cur_ctx->pool.trim();
}
static inline bool ctx_is_live(cozk_ctx* ctx) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    return g_live_ctx.count(ctx) != 0;
}
// Resident round kernels (cozk_layer_prove_rounds) are safe only for a context whose progress does not depend on GPU
// work of another context that the driver may have mapped to the same hardware queue.  libcozk cannot see such
// dependencies, so unless the host said otherwise (cozk_ctx_set_resident_rounds) the kernel is used only while this is
// the ONE live context on its device in this process -- the reference's deployment, one party per process.
static inline bool ctx_resident_rounds_enabled(cozk_ctx* ctx) {
    if (ctx->resident_rounds >= 0) return ctx->resident_rounds != 0;
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    int same_device = 0;
    for (cozk_ctx* c : g_live_ctx) same_device += c->device == ctx->device;
    return same_device <= 1;
}
static inline void* ctx_dev_alloc(cozk_ctx* ctx, size_t bytes) {
    static const bool no_pool = getenv("COZK_NO_POOL") != nullptr;  // diagnostic: plain hipMalloc / hipFree
    if (!ctx || no_pool) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
        return p;
    }
    return ctx->pool.alloc(bytes ? bytes : 16);
}
static inline void ctx_dev_free(cozk_ctx* ctx, void* p) {
    if (!p) return;
    if (ctx && ctx_is_live(ctx) && ctx->pool.release(p)) return;
    (void)hipFree(p);
}

template <class F>
static int cozk_guard(cozk_ctx* ctx, F&& f) {
    struct Cur {
        cozk_ctx* prev;
        explicit Cur(cozk_ctx* c) : prev(t_cur_ctx) { t_cur_ctx = c; }
        ~Cur() { t_cur_ctx = prev; }
    } cur(ctx);
    try {
        if (ctx) HIP_TRY(hipSetDevice(ctx->device));
        // every entry point starts from a clean per-thread error state: a failed HIP call of the host's own (or of a
        // library it uses -- RCCL probes capabilities with calls that fail benignly) must not surface in our launch checks
        (void)hipGetLastError();
        f();
        return COZK_OK;
    } catch (const CozkError& e) {
        if (ctx) ctx->last_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        if (ctx) ctx->last_error = e.what();
        return COZK_ERR_INTERNAL;
    }
}

static inline void* ctx_pinned(cozk_ctx* ctx, size_t bytes) {
    if (bytes > ctx->pinned_cap) {
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        size_t cap = bytes < 65536 ? 65536 : bytes;
        HIP_TRY(hipHostMalloc(&ctx->pinned, cap, hipHostMallocDefault));
        ctx->pinned_cap = cap;
    }
    return ctx->pinned;
}
