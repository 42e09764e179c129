// Single-node star transport for the worker sub-net / party axis: a byte all-gather through one POSIX
// shared-memory segment, one cache-line-aligned mailbox pair per participant.
//
// Why it exists: a co-jolt proof is ~300 sumcheck rounds deep and every round ends in one star exchange of
// a few hundred bytes (reference: mpc-net/src/mpc_star.rs:29-66 send_response / receive_responses /
// broadcast_request).  With one process per GPU on ONE MI355X node the exchange partner is a process on
// the same host, so the round trip should cost a cache-line hand-off (~1 us), not a socket round trip
// through a collective library (~100 us x 300 rounds = a sixth of the whole 2^20-cycle proof).  The bulk
// data path (Rep3 ring reshare) stays on RCCL/xGMI; only these latency-bound control messages go here.
//
// Protocol (lock-free, double-buffered): round r >= 1, participant i writes its payload into
// box[i][r & 1], then publishes seq = r with release order; every participant then waits for
// box[p][r & 1].seq == r (acquire) for all p and copies the payloads out.  A participant can only start
// writing round r + 2 (the next use of the same buffer) after it has completed round r + 1, which needs
// every other participant's round r + 1 message, which they publish only after they have finished reading
// round r -- so two buffers suffice and no reader can observe a torn payload.
// Failure behaviour: a wait that exceeds `timeout_ms`, or a peer that raised the shared abort flag, returns
// non-zero (the engine turns that into COZK_ERR_INTERNAL on every rank); no wait is unbounded.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>

#include "../../include/cozk.h"

namespace {
struct alignas(64) ShmHeader {
    std::atomic<uint32_t> magic;
    std::atomic<uint32_t> abort;
    uint32_t n;
    uint32_t pad;
    uint64_t slot;
};
struct alignas(64) ShmBox {
    std::atomic<uint64_t> seq;
    uint64_t len;
    // payload follows, `slot` bytes, 64-byte aligned
};
constexpr uint32_t kMagic = 0x636f7a6bu;  // "cozk"

inline uint64_t now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
}  // namespace

struct cozk_shm_hub {
    std::string name;
    int n = 0, me = 0;
    size_t slot = 0, box_stride = 0, map_len = 0;
    uint8_t* base = nullptr;
    uint64_t round = 0;
    uint64_t timeout_ns = 120ull * 1000000000ull;
    bool creator = false;
    ShmHeader* hdr() const { return reinterpret_cast<ShmHeader*>(base); }
    ShmBox* box(int p, int parity) const {
        return reinterpret_cast<ShmBox*>(base + sizeof(ShmHeader) + ((size_t)p * 2 + parity) * box_stride);
    }
    uint8_t* payload(int p, int parity) const { return reinterpret_cast<uint8_t*>(box(p, parity)) + sizeof(ShmBox); }
};

static int shm_all_gather(void* user, const void* send, size_t len, void* recv, size_t cap, size_t* lens) {
    cozk_shm_hub* h = static_cast<cozk_shm_hub*>(user);
    if (!h || !h->base || len > h->slot || (len && !send) || !recv || !lens) return 1;
    const uint64_t r = ++h->round;
    const int par = (int)(r & 1);
    ShmBox* mine = h->box(h->me, par);
    if (len) memcpy(h->payload(h->me, par), send, len);
    mine->len = len;
    mine->seq.store(r, std::memory_order_release);
    uint64_t t0 = 0;
    for (int p = 0; p < h->n; p++) {
        ShmBox* b = h->box(p, par);
        uint32_t spins = 0;
        while (b->seq.load(std::memory_order_acquire) != r) {
            if (h->hdr()->abort.load(std::memory_order_relaxed)) return 3;
            if (++spins < 4096) {
                __builtin_ia32_pause();
                continue;
            }
            // past ~20 us of spinning: a peer is inside a long GPU phase; stop burning its host core
            if (!t0) t0 = now_ns();
            if ((spins & 63) == 0 && now_ns() - t0 > h->timeout_ns) {
                h->hdr()->abort.store(1, std::memory_order_relaxed);
                return 4;
            }
            if (spins < 4096 + 2000) sched_yield();
            else {
                timespec ts{0, 20000};
                nanosleep(&ts, nullptr);
            }
        }
        const size_t l = b->len;
        if (l > cap) {
            h->hdr()->abort.store(1, std::memory_order_relaxed);
            return 2;
        }
        if (l) memcpy(static_cast<uint8_t*>(recv) + (size_t)p * cap, h->payload(p, par), l);
        lens[p] = l;
    }
    return 0;
}

extern "C" {

// create == 1: make the segment (exactly one participant; fails if the name exists), zero-filled by the
// kernel; create == 0: attach to it (after the creator returned -- order the two with the launcher's own
// barrier).  slot_bytes = largest single message (the harness sends <= 256 KiB).
int cozk_shm_hub_open(const char* name, int create, int n_participants, int my_index, size_t slot_bytes,
                      cozk_shm_hub** out) {
    if (!name || !out || n_participants < 1 || my_index < 0 || my_index >= n_participants || slot_bytes == 0)
        return COZK_ERR_INVALID_ARG;
    *out = nullptr;
    cozk_shm_hub* h = new (std::nothrow) cozk_shm_hub();
    if (!h) return COZK_ERR_OOM;
    h->name = name;
    h->n = n_participants;
    h->me = my_index;
    h->slot = (slot_bytes + 63) & ~(size_t)63;
    h->box_stride = sizeof(ShmBox) + h->slot;
    h->map_len = sizeof(ShmHeader) + (size_t)n_participants * 2 * h->box_stride;
    h->creator = create != 0;
    int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) {
        delete h;
        return COZK_ERR_INTERNAL;
    }
    if (create) {
        if (ftruncate(fd, (off_t)h->map_len) != 0) {
            close(fd);
            shm_unlink(name);
            delete h;
            return COZK_ERR_OOM;
        }
    } else {
        struct stat st;
        if (fstat(fd, &st) != 0 || (size_t)st.st_size != h->map_len) {  // attached with different geometry
            close(fd);
            delete h;
            return COZK_ERR_INVALID_ARG;
        }
    }
    void* m = mmap(nullptr, h->map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        if (create) shm_unlink(name);
        delete h;
        return COZK_ERR_OOM;
    }
    h->base = static_cast<uint8_t*>(m);
    if (create) {
        h->hdr()->n = (uint32_t)n_participants;
        h->hdr()->slot = h->slot;
        h->hdr()->magic.store(kMagic, std::memory_order_release);
    } else if (h->hdr()->magic.load(std::memory_order_acquire) != kMagic || h->hdr()->n != (uint32_t)n_participants ||
               h->hdr()->slot != h->slot) {
        munmap(m, h->map_len);
        delete h;
        return COZK_ERR_INVALID_ARG;
    }
    *out = h;
    return COZK_OK;
}

// the cozk_hub_net view of the segment (valid until cozk_shm_hub_close)
int cozk_shm_hub_net(cozk_shm_hub* h, cozk_hub_net* out) {
    if (!h || !out) return COZK_ERR_INVALID_ARG;
    out->user = h;
    out->n_participants = h->n;
    out->my_index = h->me;
    out->all_gather = shm_all_gather;
    return COZK_OK;
}

// remove the name (creator, once everyone has attached); the mapping stays valid until close
int cozk_shm_hub_unlink(cozk_shm_hub* h) {
    if (!h) return COZK_ERR_INVALID_ARG;
    if (h->creator) shm_unlink(h->name.c_str());
    return COZK_OK;
}

int cozk_shm_hub_set_timeout_ms(cozk_shm_hub* h, uint64_t ms) {
    if (!h || ms == 0) return COZK_ERR_INVALID_ARG;
    h->timeout_ns = ms * 1000000ull;
    return COZK_OK;
}

// raise the shared abort flag: every participant blocked in (or entering) all_gather returns an error
void cozk_shm_hub_abort(cozk_shm_hub* h) {
    if (h && h->base) h->hdr()->abort.store(1, std::memory_order_relaxed);
}

void cozk_shm_hub_close(cozk_shm_hub* h) {
    if (!h) return;
    if (h->base) munmap(h->base, h->map_len);
    delete h;
}

}  // extern "C"
