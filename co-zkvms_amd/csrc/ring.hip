// Native Rep3 ring over RCCL: the exchange step of rep3::arithmetic::{mul, mul_vec} and of
// reshare_additive(_many) (mpc-core/src/protocols/rep3/arithmetic.rs:144-164) -- every party sends its new additive
// share c.a to the NEXT party and receives c.b from the PREVIOUS one (mpc-types/src/protocols/rep3/id.rs:31-47) -- as one
// ncclSend / ncclRecv pair per call, enqueued on the context's stream.  One party per GPU (one process per GPU, the
// reference's deployment): the payload goes GPU to GPU over xGMI and nothing on the host waits for it -- the kernels
// that produce the send buffer and the kernels that consume the receive buffer are ordered by the stream.
//
// xGMI is point to point (7 links x ~153 GB/s per GPU), and a 3-party ring uses exactly one outgoing and one incoming
// link per GPU, so a reshare is bound by ONE link: ~4.4 ms for the 2^25-element first layer of the config-3 grand
// product (1 GiB), which is why the ring stays asynchronous behind the stream instead of being waited for.
//
// librccl (570 MB) is loaded lazily on the first cozk_ring_* call, reusing a copy the process already holds (a torch
// host has one), so hosts that never open a ring (the plain prover, worker sub-nets) do not pay for it and libcozk
// loads where RCCL is absent.
#include <dlfcn.h>
#include <string.h>

#include <rccl/rccl.h>

#include "common.hpp"
#include "poly.hip.hpp"
#include "prf.hip.hpp"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)  // a copy the process already mapped wins (never two RCCLs in one process)
            if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!api.handle)
            for (const char* n : names)
                if ((api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!api.handle) {
            api.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char* s) {
            void* p = dlsym(api.handle, s);
            if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + s;
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    return api;
}

RcclApi& rccl_or_throw() {
    RcclApi& a = rccl();
    if (!a.error.empty() || !a.handle) throw CozkError(COZK_ERR_INTERNAL, "ring: " + (a.error.empty() ? std::string("librccl unavailable") : a.error));
    return a;
}

#define RCCL_TRY(api, expr)                                                                                   \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) throw CozkError(COZK_ERR_INTERNAL, std::string(#expr " failed: ") + (api).GetErrorString(r_)); \
    } while (0)

// send n bytes to the next party, receive n bytes from the previous one, on the context's stream
void ring_exchange(cozk_ctx* ctx, const void* dev_send, void* dev_recv, size_t nbytes) {
    COZK_REQUIRE(ctx->ring_comm, "ring: cozk_ring_init has not been called on this context");
    if (nbytes == 0) return;
    RcclApi& a = rccl_or_throw();
    ncclComm_t comm = (ncclComm_t)ctx->ring_comm;
    const int next = (ctx->ring_rank + 1) % ctx->ring_n, prev = (ctx->ring_rank + ctx->ring_n - 1) % ctx->ring_n;
    RCCL_TRY(a, a.GroupStart());
    ncclResult_t rs = a.Send(dev_send, nbytes, ncclUint8, next, comm, ctx->stream);
    ncclResult_t rr = rs == ncclSuccess ? a.Recv(dev_recv, nbytes, ncclUint8, prev, comm, ctx->stream) : rs;
    ncclResult_t rg = a.GroupEnd();
    if (rs != ncclSuccess || rr != ncclSuccess) throw CozkError(COZK_ERR_INTERNAL, std::string("ring: ncclSend/ncclRecv failed: ") + a.GetErrorString(rs != ncclSuccess ? rs : rr));
    if (rg != ncclSuccess) throw CozkError(COZK_ERR_INTERNAL, std::string("ring: ncclGroupEnd failed: ") + a.GetErrorString(rg));
    (void)hipGetLastError();  // RCCL probes capabilities with HIP calls that may fail benignly: never leave their error for our next launch check
    ctx->ring_bytes += nbytes;
}

int native_reshare_cb(void* user, const void* dev_send, void* dev_recv, size_t nbytes) {
    cozk_ctx* ctx = static_cast<cozk_ctx*>(user);
    return cozk_guard(ctx, [&] { ring_exchange(ctx, dev_send, dev_recv, nbytes); }) == COZK_OK ? 0 : 1;
}

// c.a = x (x) y + PRF(key_self, ctr + j) - PRF(key_prev, ctr + j): the local half of mul_vec
// (mpc-types/src/protocols/rep3/arithmetic/ops.rs:71-78) -- the same arithmetic as k_mul_vec_local in poly.hip
__global__ void __launch_bounds__(256) k_ring_mul_local(const fe* __restrict__ xa, const fe* __restrict__ xb, const fe* __restrict__ ya,
                                                        const fe* __restrict__ yb, size_t n, fe* __restrict__ out, prf_key key_self, prf_key key_prev,
                                                        uint64_t ctr) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    fe v = sh_local_mul<2>(sh_load<2>(xa, xb, j), sh_load<2>(ya, yb, j));
    v = Fr::add(v, Fr::sub(prf_fr(key_self, ctr + j), prf_fr(key_prev, ctr + j)));
    fe_store(out + j, v);
}

}  // namespace

extern "C" {

int cozk_ring_unique_id(uint8_t out[COZK_RING_ID_BYTES]) {
    if (!out) return COZK_ERR_INVALID_ARG;
    static_assert(COZK_RING_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ring id size");
    try {
        RcclApi& a = rccl_or_throw();
        ncclUniqueId id;
        RCCL_TRY(a, a.GetUniqueId(&id));
        memcpy(out, id.internal, COZK_RING_ID_BYTES);
        return COZK_OK;
    } catch (const std::exception&) {
        return COZK_ERR_INTERNAL;
    }
}

int cozk_ring_init(cozk_ctx* ctx, const uint8_t id_bytes[COZK_RING_ID_BYTES], int rank, int nranks) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && id_bytes && nranks >= 1 && rank >= 0 && rank < nranks, "ring_init: bad argument");
        COZK_REQUIRE(!ctx->ring_comm, "ring_init: this context already has a ring");
        RcclApi& a = rccl_or_throw();
        ncclUniqueId id;
        memcpy(id.internal, id_bytes, COZK_RING_ID_BYTES);
        ncclComm_t comm = nullptr;
        RCCL_TRY(a, a.CommInitRank(&comm, nranks, id, rank));  // blocks until every rank of the ring has called it
        (void)hipGetLastError();  // (see ring_exchange)
        HIP_TRY(hipSetDevice(ctx->device));
        ctx->ring_comm = comm;
        ctx->ring_rank = rank;
        ctx->ring_n = nranks;
        ctx->ring_bytes = 0;
    });
}

int cozk_ring_destroy(cozk_ctx* ctx) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx, "ring_destroy: null context");
        if (!ctx->ring_comm) return;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        RcclApi& a = rccl_or_throw();
        ncclComm_t comm = (ncclComm_t)ctx->ring_comm;
        ctx->ring_comm = nullptr;
        ncclResult_t r = a.CommDestroy(comm);
        (void)hipGetLastError();
        if (r != ncclSuccess) throw CozkError(COZK_ERR_INTERNAL, std::string("ncclCommDestroy failed: ") + a.GetErrorString(r));
    });
}

int cozk_ring_info(cozk_ctx* ctx, int* rank, int* nranks, uint64_t* bytes_sent) {
    if (!ctx || !ctx->ring_comm) return COZK_ERR_INVALID_ARG;
    if (rank) *rank = ctx->ring_rank;
    if (nranks) *nranks = ctx->ring_n;
    if (bytes_sent) *bytes_sent = ctx->ring_bytes;
    return COZK_OK;
}

int cozk_reshare(cozk_ctx* ctx, const cozk_vec* send, cozk_vec* recv) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && send && recv && send->kind == COZK_SCALAR_FR && recv->kind == COZK_SCALAR_FR && send->n == recv->n && send->d != recv->d,
                     "reshare: send / recv must be distinct FR vectors of one length");
        ring_exchange(ctx, send->d, recv->d, send->n * sizeof(fe));
    });
}

int cozk_rep3_mul_vec(cozk_ctx* ctx, const cozk_vec* xa, const cozk_vec* xb, const cozk_vec* ya, const cozk_vec* yb, const uint8_t* key_self,
                      const uint8_t* key_prev, uint64_t counter, cozk_vec** out_a, cozk_vec** out_b) {
    if (!out_a || !out_b) return COZK_ERR_INVALID_ARG;
    *out_a = *out_b = nullptr;
    int rc = cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && xa && xb && ya && yb && key_self && key_prev && xa->n == xb->n && xa->n == ya->n && xa->n == yb->n &&
                         xa->kind == COZK_SCALAR_FR && xb->kind == COZK_SCALAR_FR && ya->kind == COZK_SCALAR_FR && yb->kind == COZK_SCALAR_FR,
                     "rep3_mul_vec: four FR vectors of one length and two PRF keys are required");
        COZK_REQUIRE(ctx->ring_comm, "rep3_mul_vec: cozk_ring_init has not been called on this context");
    });
    if (rc != COZK_OK) return rc;
    rc = cozk_vec_alloc(ctx, xa->n, COZK_SCALAR_FR, out_a);
    if (rc == COZK_OK) rc = cozk_vec_alloc(ctx, xa->n, COZK_SCALAR_FR, out_b);
    if (rc == COZK_OK)
        rc = cozk_guard(ctx, [&] {
            size_t n = xa->n;
            if (n == 0) return;
            k_ring_mul_local<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>((const fe*)xa->d, (const fe*)xb->d, (const fe*)ya->d, (const fe*)yb->d, n,
                                                                                   (fe*)(*out_a)->d, prf_key_from_bytes(key_self), prf_key_from_bytes(key_prev),
                                                                                   counter);
            HIP_TRY(hipGetLastError());
            ring_exchange(ctx, (*out_a)->d, (*out_b)->d, n * sizeof(fe));
        });
    if (rc != COZK_OK) {
        cozk_vec_free(*out_a);
        cozk_vec_free(*out_b);
        *out_a = *out_b = nullptr;
    }
    return rc;
}

int cozk_ring_net_native(cozk_ctx* ctx, cozk_ring_net* out) {
    if (!ctx || !out || !ctx->ring_comm) return COZK_ERR_INVALID_ARG;
    out->user = ctx;
    out->reshare = native_reshare_cb;
    out->stream_ordered = 1;
    return COZK_OK;
}

}  // extern "C"
