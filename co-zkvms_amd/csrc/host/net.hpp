// Network seam: the star (worker <-> coordinator) and ring (party -> next party)
// interfaces the worker-side drivers are written against, plus two implementations:
//   * in-process channels (threads + condition variables): parties and the coordinator of one
//     proof inside one process, one GPU stream per party -- the stand-in for the reference's
//     localhost QUIC demo (co-jolt/examples/run_3_party_jolt.sh);
//   * C callbacks, so that a host (the Rust prover, or bench.py over torch.distributed / RCCL)
//     plugs its own transport in.
// Traits mirrored: MpcStarNetWorker::{send_response, receive_request},
// MpcStarNetCoordinator::{receive_responses, broadcast_request, send_request, receive_response}
// (mpc-net/src/mpc_star.rs:5-66) and Rep3Network::{reshare_many} (used
// mpc-core/src/protocols/rep3/arithmetic.rs:144-164).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "wire.hpp"

namespace cozk {

struct StarNetWorker {
    virtual ~StarNetWorker() {}
    virtual void send_response(const Bytes& b) = 0;
    virtual Bytes receive_request() = 0;
    uint64_t bytes_up = 0, bytes_down = 0, n_msgs = 0;
};

struct StarNetCoordinator {
    virtual ~StarNetCoordinator() {}
    virtual int n_workers() const = 0;
    // responses ordered by global_worker_id = worker*3 + party (mpc-net/src/rep3/mod.rs:29-32)
    virtual std::vector<Bytes> receive_responses() = 0;
    virtual Bytes receive_response(int party) = 0;
    virtual void broadcast_request(const Bytes& b) = 0;
    virtual void send_request(int party, const Bytes& b) = 0;
};

struct RingNet {
    virtual ~RingNet() {}
    // send n field elements (device) to the next party, receive n from the previous one (device)
    virtual void reshare(cozk_ctx* ctx, const fe* dev_send, fe* dev_recv, size_t n) = 0;
    uint64_t bytes_sent = 0;
};

// ---------------------------------------------------------------- in-process transport
struct Abort {
    std::atomic<bool> flag{false};
};

// One producer, one consumer (every channel here is: worker -> coordinator, coordinator -> worker, party -> next party): a ring of
// slots with two counters, no lock.  A sumcheck round is a ~20 us ping-pong between a worker and the coordinator and a chained proof
// has ~1660 of them; the mutex + deque + condition variable this replaces measured 2.5-7 us per round trip (the consumer saw the
// count rise inside the producer's critical section and then waited for its mutex).
template <class T>
struct Chan {
    static constexpr size_t CAP = 1024;  // messages in flight per direction are a handful (request / response protocols)
    std::vector<T> slots = std::vector<T>(CAP);
    alignas(64) std::atomic<size_t> head{0};  // next slot to pop (written by the consumer)
    alignas(64) std::atomic<size_t> tail{0};  // next slot to fill (written by the producer)
    Abort* abort = nullptr;
    void wait_step(int& spin) {
        // busy-poll first (a sched_yield per probe costs microseconds on a loaded host, a sleep tens), then yield, then sleep
        if ((++spin & 1023) == 0 && abort && abort->flag.load()) throw CozkError(COZK_ERR_INTERNAL, "aborted: a peer failed");
        if (spin < 40000) __builtin_ia32_pause();
        else if (spin < 60000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    void push(T v) {
        const size_t t = tail.load(std::memory_order_relaxed);
        int spin = 0;
        while (t - head.load(std::memory_order_acquire) >= CAP) wait_step(spin);
        slots[t % CAP] = std::move(v);
        tail.store(t + 1, std::memory_order_release);
    }
    T pop() {
        const size_t h = head.load(std::memory_order_relaxed);
        int spin = 0;
        while (tail.load(std::memory_order_acquire) == h) wait_step(spin);
        T v = std::move(slots[h % CAP]);
        head.store(h + 1, std::memory_order_release);
        return v;
    }
};

struct InProcStar {
    int n;
    Abort abort;
    std::vector<Chan<Bytes>> up, down;
    explicit InProcStar(int nparties) : n(nparties), up(nparties), down(nparties) {
        for (auto& c : up) c.abort = &abort;
        for (auto& c : down) c.abort = &abort;
    }
};

struct InProcStarWorker : StarNetWorker {
    InProcStar* s;
    int id;
    InProcStarWorker(InProcStar* s_, int id_) : s(s_), id(id_) {}
    void send_response(const Bytes& b) override {
        bytes_up += b.size();
        n_msgs++;
        s->up[id].push(b);
    }
    Bytes receive_request() override {
        Bytes b = s->down[id].pop();
        bytes_down += b.size();
        return b;
    }
};

struct InProcStarCoordinator : StarNetCoordinator {
    InProcStar* s;
    explicit InProcStarCoordinator(InProcStar* s_) : s(s_) {}
    int n_workers() const override { return s->n; }
    std::vector<Bytes> receive_responses() override {
        std::vector<Bytes> r;
        for (int p = 0; p < s->n; p++) r.push_back(s->up[p].pop());
        return r;
    }
    Bytes receive_response(int party) override { return s->up[party].pop(); }
    void broadcast_request(const Bytes& b) override {
        for (int p = 0; p < s->n; p++) s->down[p].push(b);
    }
    void send_request(int party, const Bytes& b) override { s->down[party].push(b); }
};

struct RingMsg {
    const fe* ptr;
    size_t n;
};

struct InProcRing {
    Abort* abort;
    Chan<RingMsg> data[3];
    Chan<int> ack[3];
    explicit InProcRing(Abort* a) : abort(a) {
        for (auto& c : data) c.abort = a;
        for (auto& c : ack) c.abort = a;
    }
};

struct InProcRingNet : RingNet {
    InProcRing* r;
    int id;
    InProcRingNet(InProcRing* r_, int id_) : r(r_), id(id_) {}
    void reshare(cozk_ctx* ctx, const fe* dev_send, fe* dev_recv, size_t n) override {
        int next = (id + 1) % 3, prev = (id + 2) % 3;
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // dev_send is complete before the peer reads it
        r->data[next].push(RingMsg{dev_send, n});
        RingMsg m = r->data[id].pop();
        if (m.n != n) throw CozkError(COZK_ERR_INTERNAL, "ring reshare: length mismatch between parties");
        // same-device or peer copy; UVA resolves the source device
        HIP_TRY(hipMemcpyAsync(dev_recv, m.ptr, n * sizeof(fe), hipMemcpyDefault, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        r->ack[prev].push(1);  // prev may now reuse its send buffer
        (void)r->ack[id].pop();
        bytes_sent += n * sizeof(fe);
    }
};

// ---------------------------------------------------------------- C-callback transport
struct CallbackStarWorker : StarNetWorker {
    cozk_star_net cb;
    explicit CallbackStarWorker(const cozk_star_net& c) : cb(c) {}
    void send_response(const Bytes& b) override {
        bytes_up += b.size();
        n_msgs++;
        if (cb.send_response(cb.user, b.data(), b.size()) != 0) throw CozkError(COZK_ERR_INTERNAL, "star send_response callback failed");
    }
    Bytes receive_request() override {
        // requests are small (a point of <= 64 challenges + a claim); 1 MiB is a generous cap
        Bytes b(1 << 20);
        size_t len = 0;
        if (cb.receive_request(cb.user, b.data(), b.size(), &len) != 0 || len > b.size())
            throw CozkError(COZK_ERR_INTERNAL, "star receive_request callback failed");
        b.resize(len);
        bytes_down += len;
        return b;
    }
};

struct CallbackRingNet : RingNet {
    cozk_ring_net cb;
    explicit CallbackRingNet(const cozk_ring_net& c) : cb(c) {}
    void reshare(cozk_ctx* ctx, const fe* dev_send, fe* dev_recv, size_t n) override {
        // a foreign transport reads dev_send from the host's side: the stream must have produced it; a stream-ordered
        // one (the native RCCL ring) enqueues behind the producing kernels and nothing waits
        if (!cb.stream_ordered) HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (cb.reshare(cb.user, (const void*)dev_send, (void*)dev_recv, n * sizeof(fe)) != 0)
            throw CozkError(COZK_ERR_INTERNAL, cb.stream_ordered ? std::string("ring reshare failed: ") + ctx->last_error : std::string("ring reshare callback failed"));
        bytes_sent += n * sizeof(fe);
    }
};

}  // namespace cozk
