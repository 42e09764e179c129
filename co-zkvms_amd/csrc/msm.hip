// Pippenger MSM over BN254 G1 for gfx950 -- the kernel set behind the MSM seam
// (`VariableBaseMSM::{msm_field_elements, batch_msm}`; call sites
// co-jolt/src/poly/commitment/pst13.rs:286-294,319-323,461-469,
// co-noir-spartan/co-spartan/src/worker.rs:585,801-804).
//
// MI355X-first design (not a translation of jolt-core's rayon-over-windows CPU code):
//   * 16-bit signed digits -> 2^15 buckets per group.  With a precomputed window table
//     T[w][i] = 2^(16w) G_i (16 x n x 64 B, cheap in 288 GB of HBM) every window of every scalar
//     lands in ONE bucket group per polynomial, so the per-window running-sum reductions and the
//     Horner doublings disappear; without the table the 16 windows are 16 groups.
//   * counting sort by bucket in HBM (histogram -> scan -> scatter of 4-byte point references);
//     bucket order is arbitrary, which is fine because group addition commutes and all special
//     cases (doubling / cancellation) are handled.
//   * bucket accumulation = segmented reduction: one lane owns a segment of <= L consecutive
//     sorted references and adds the gathered affine points into an XYZZ accumulator
//     (8M+2S, no inversion).  Segments, not buckets, are the unit of parallelism, so skewed
//     scalars (0/1 flags, u8/u16 witness columns: one bucket holding n/2 points) cost the same as
//     uniform ones; further levels fold the per-segment partial sums until one value per bucket.
//   * a batch of P polynomials is just P x more groups in the same launches (amortises the
//     latency-bound bucket-reduction tail).
// Results are returned as affine points, so the output is bit-exact to any correct MSM.
#include "common.hpp"
#include "fq9.hip.hpp"

static constexpr uint32_t NB = 1u << 15;  // buckets per group
static constexpr int CH = 16;             // buckets per reduction chunk
static constexpr int TPB = 256;

// ------------------------------------------------------------------ scalar -> signed digits
template <int KIND>
struct ScalarKind;
template <>
struct ScalarKind<COZK_SCALAR_FR> {
    static constexpr int NWIN = 16;
    static __device__ __forceinline__ bool load(const void* p, size_t i, uint32_t w[8]) {
        fe s = Fr::from_mont(fe_load(reinterpret_cast<const fe*>(p) + i));
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = s.l[k];
        return false;
    }
};
template <>
struct ScalarKind<COZK_SCALAR_U8> {
    static constexpr int NWIN = 1;
    static __device__ __forceinline__ bool load(const void* p, size_t i, uint32_t w[8]) {
        w[0] = reinterpret_cast<const uint8_t*>(p)[i];
        return false;
    }
};
template <>
struct ScalarKind<COZK_SCALAR_U16> {
    static constexpr int NWIN = 2;
    static __device__ __forceinline__ bool load(const void* p, size_t i, uint32_t w[8]) {
        w[0] = reinterpret_cast<const uint16_t*>(p)[i];
        return false;
    }
};
template <>
struct ScalarKind<COZK_SCALAR_U32> {
    static constexpr int NWIN = 3;
    static __device__ __forceinline__ bool load(const void* p, size_t i, uint32_t w[8]) {
        w[0] = reinterpret_cast<const uint32_t*>(p)[i];
        w[1] = 0;
        return false;
    }
};
template <>
struct ScalarKind<COZK_SCALAR_U64> {
    static constexpr int NWIN = 5;
    static __device__ __forceinline__ bool load(const void* p, size_t i, uint32_t w[8]) {
        uint64_t v = reinterpret_cast<const uint64_t*>(p)[i];
        w[0] = (uint32_t)v;
        w[1] = (uint32_t)(v >> 32);
        w[2] = 0;
        return false;
    }
};
template <>
struct ScalarKind<COZK_SCALAR_I64> {
    static constexpr int NWIN = 5;
    static __device__ __forceinline__ bool load(const void* p, size_t i, uint32_t w[8]) {
        int64_t v = reinterpret_cast<const int64_t*>(p)[i];
        bool neg = v < 0;
        uint64_t m = neg ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
        w[0] = (uint32_t)m;
        w[1] = (uint32_t)(m >> 32);
        w[2] = 0;
        return neg;
    }
};

static inline int kind_nwin(int kind) {
    switch (kind) {
        case COZK_SCALAR_FR: return 16;
        case COZK_SCALAR_U8: return 1;
        case COZK_SCALAR_U16: return 2;
        case COZK_SCALAR_U32: return 3;
        default: return 5;
    }
}

// Enumerate the non-zero signed digits of scalar i: calls f(window, bucket, negative).
template <int KIND, class F>
__device__ __forceinline__ void for_each_digit(const void* scalars, size_t i, F&& f) {
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool neg = ScalarKind<KIND>::load(scalars, i, w);
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < ScalarKind<KIND>::NWIN; k++) {
        uint32_t v = ((w[k >> 1] >> (16 * (k & 1))) & 0xffffu) + carry;
        bool sub = v > 32768u;
        carry = sub ? 1u : 0u;
        uint32_t mag = sub ? 65536u - v : v;
        f(k, mag, sub != neg);
    }
}

// counter[key] += 1 for every active lane; lanes that share the first active lane's key are
// merged into one atomic (kills the single-hot-bucket contention of 0/1 and tiny-valued columns).
static __device__ __forceinline__ uint32_t agg_atomic_add(uint32_t* counters, uint32_t key, bool active) {
    uint64_t act = __ballot(active);
    uint32_t pos = 0;
    if (act == 0) return 0;
    int lane = threadIdx.x & 63;
    int first = __ffsll((unsigned long long)act) - 1;
    uint32_t k0 = __shfl(key, first);
    uint64_t same = __ballot(active && key == k0);
    int cnt = __popcll(same);
    bool merged = false;
    if (cnt >= 4) {
        uint32_t base = 0;
        if (lane == first) base = atomicAdd(&counters[k0], (uint32_t)cnt);
        base = __shfl(base, first);
        if (active && key == k0) {
            pos = base + __popcll(same & ((1ull << lane) - 1ull));
            merged = true;
        }
    }
    if (active && !merged) pos = atomicAdd(&counters[key], 1u);
    return pos;
}

template <int KIND>
__global__ void __launch_bounds__(TPB) k_msm_hist(const void* scalars, size_t n, uint32_t* hist, int grouped) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    bool inr = i < n;
    size_t ii = inr ? i : 0;
    for_each_digit<KIND>(scalars, ii, [&](int k, uint32_t mag, bool) {
        bool act = inr && mag != 0;
        uint32_t key = (grouped ? (uint32_t)k * NB : 0u) + (mag - 1u);
        agg_atomic_add(hist, act ? key : 0u, act);
    });
}

template <int KIND>
__global__ void __launch_bounds__(TPB) k_msm_scatter(const void* scalars, size_t n, uint32_t* cursor, uint32_t* refs,
                                                  int grouped, uint32_t table_n, uint32_t base_off) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    bool inr = i < n;
    size_t ii = inr ? i : 0;
    for_each_digit<KIND>(scalars, ii, [&](int k, uint32_t mag, bool negative) {
        bool act = inr && mag != 0;
        uint32_t key = (grouped ? (uint32_t)k * NB : 0u) + (mag - 1u);
        uint32_t pos = agg_atomic_add(cursor, act ? key : 0u, act);
        if (act) {
            uint32_t ref = (grouped ? 0u : (uint32_t)k * table_n) + base_off + (uint32_t)ii;
            refs[pos] = ref | (negative ? 0x80000000u : 0u);
        }
    });
}

// ------------------------------------------------------------------ LDS-privatised counting sort
// With the window table every digit of a polynomial lands in ONE group of 2^15 buckets, and a
// 2^15 x u32 histogram is exactly 128 KiB -- it fits the 160 KiB LDS of a CDNA4 CU.  One 1024-thread
// workgroup per CU keeps the whole histogram in LDS: digits are counted with LDS atomics (no HBM
// traffic, no global contention even for 0/1 columns where every lane hits bucket 0), then flushed
// with one global atomic per non-empty (workgroup, bucket).  The scatter pass recounts, claims a
// contiguous range per (workgroup, bucket) with one returning global atomic, and hands out slots
// with returning LDS atomics.  Global atomics drop from 2 x 16 n to <= 2 x 2^15 x workgroups.
struct MsmPolyDesc {
    const void* scalars;
    uint32_t n;         // scalars in this polynomial's slice
    uint32_t base_off;  // first SRS index of the slice
    uint32_t group;     // bucket group (= polynomial index inside the launch set)
    uint32_t pad;
};
static constexpr int LTPB = 1024;  // launch bound of the LDS-histogram kernels; the launch width is chosen in msm_sort

template <int KIND>
__global__ void __launch_bounds__(LTPB) k_msm_hist_lds(const MsmPolyDesc* __restrict__ descs, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[NB];
    const MsmPolyDesc d = descs[blockIdx.y];
    for (uint32_t b = threadIdx.x; b < NB; b += blockDim.x) lh[b] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (size_t)gridDim.x * blockDim.x) {
        for_each_digit<KIND>(d.scalars, i, [&](int, uint32_t mag, bool) {
            if (mag != 0) atomicAdd(&lh[mag - 1u], 1u);
        });
    }
    __syncthreads();
    uint32_t* h = hist + (size_t)d.group * NB;
    for (uint32_t b = threadIdx.x; b < NB; b += blockDim.x) {
        uint32_t c = lh[b];
        if (c) atomicAdd(&h[b], c);
    }
}

template <int KIND>
__global__ void __launch_bounds__(LTPB) k_msm_scatter_lds(const MsmPolyDesc* __restrict__ descs, uint32_t* __restrict__ cursor,
                                                       uint32_t* __restrict__ refs, uint32_t table_n) {
    __shared__ uint32_t lh[NB];
    const MsmPolyDesc d = descs[blockIdx.y];
    for (uint32_t b = threadIdx.x; b < NB; b += blockDim.x) lh[b] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (size_t)gridDim.x * blockDim.x) {
        for_each_digit<KIND>(d.scalars, i, [&](int, uint32_t mag, bool) {
            if (mag != 0) atomicAdd(&lh[mag - 1u], 1u);
        });
    }
    __syncthreads();
    uint32_t* cur = cursor + (size_t)d.group * NB;
    for (uint32_t b = threadIdx.x; b < NB; b += blockDim.x) {
        uint32_t c = lh[b];
        lh[b] = c ? atomicAdd(&cur[b], c) : 0u;  // this workgroup's slot range inside bucket b
    }
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (size_t)gridDim.x * blockDim.x) {
        for_each_digit<KIND>(d.scalars, i, [&](int k, uint32_t mag, bool negative) {
            if (mag != 0) {
                uint32_t pos = atomicAdd(&lh[mag - 1u], 1u);
                uint32_t ref = (uint32_t)k * table_n + d.base_off + (uint32_t)i;
                refs[pos] = ref | (negative ? 0x80000000u : 0u);
            }
        });
    }
}

// ------------------------------------------------------------------ scans
// exclusive prefix over nb counters in three small launches: per-block sums, one-block scan of the
// block sums, per-block rescan + offset.  1024 counters per block (256 threads x uint4).
// FROM_OFFSETS: the counter of bucket b is ceil((in[b+1] - in[b]) / L) (segments of the next level).
static constexpr uint32_t SCAN_PER_BLOCK = 1024;

template <bool FROM_OFFSETS>
static __device__ __forceinline__ void scan_load4(const uint32_t* __restrict__ in, uint32_t nb, uint32_t L, uint32_t b0, uint32_t c[4],
                                                  const uint32_t* __restrict__ perm) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t b = b0 + k;
        if (b >= nb) c[k] = 0;
        else if (FROM_OFFSETS) {
            uint32_t q = perm ? perm[b] : b;  // with a permutation: the items of position b are those of entry perm[b]
            c[k] = (in[q + 1] - in[q] + L - 1u) / L;
        } else c[k] = in[b];
    }
}

static __device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* sh4, uint32_t* total) {
    // inclusive wave scan
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(x, off);
        if ((int)(threadIdx.x & 63) >= off) x += y;
    }
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) sh4[w] = x;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < w; i++) base += sh4[i];
    if (total) *total = sh4[0] + sh4[1] + sh4[2] + sh4[3];
    __syncthreads();
    return base + x - v;
}

template <bool FROM_OFFSETS>
__global__ void __launch_bounds__(256) k_scan_sums(const uint32_t* __restrict__ in, uint32_t nb, uint32_t L, uint32_t* __restrict__ bsums,
                                                 const uint32_t* __restrict__ perm) {
    __shared__ uint32_t sh4[4];
    uint32_t c[4];
    scan_load4<FROM_OFFSETS>(in, nb, L, blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * 4, c, perm);
    uint32_t tot;
    (void)block_exclusive_scan_256(c[0] + c[1] + c[2] + c[3], sh4, &tot);
    if (threadIdx.x == 0) bsums[blockIdx.x] = tot;
}

// exclusive scan of the block sums in place (nblk <= 16384); writes the grand total to *total_out
__global__ void __launch_bounds__(1024) k_scan_top(uint32_t* bsums, uint32_t nblk, uint32_t* total_out) {
    __shared__ uint32_t sh[1024];
    uint32_t tid = threadIdx.x;
    uint32_t per = (nblk + 1023u) / 1024u;
    uint32_t b0 = tid * per, b1 = min(nblk, b0 + per);
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; b++) sum += bsums[b];
    sh[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = tid >= d ? sh[tid - d] : 0;
        __syncthreads();
        sh[tid] += v;
        __syncthreads();
    }
    uint32_t run = sh[tid] - sum;
    for (uint32_t b = b0; b < b1; b++) {
        uint32_t c = bsums[b];
        bsums[b] = run;
        run += c;
    }
    if (tid == 1023) *total_out = sh[1023];
}

template <bool FROM_OFFSETS>
__global__ void __launch_bounds__(256) k_scan_final(const uint32_t* in, uint32_t nb, uint32_t L, const uint32_t* __restrict__ bsums,
                                                  uint32_t* off, uint32_t* cursor, const uint32_t* __restrict__ perm) {
    __shared__ uint32_t sh4[4];
    uint32_t c[4];
    uint32_t b0 = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * 4;
    scan_load4<FROM_OFFSETS>(in, nb, L, b0, c, perm);
    uint32_t run = bsums[blockIdx.x] + block_exclusive_scan_256(c[0] + c[1] + c[2] + c[3], sh4, nullptr);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t b = b0 + k;
        if (b < nb) {
            off[b] = run;
            if (cursor) cursor[b] = run;
        }
        run += c[k];
    }
}

// legacy single-block scan (kept for tiny inputs and as the reference the multi-block one is tested against)
// off[b] = exclusive prefix of cnt, off[nb] = total; cursor (optional) = copy of off[0..nb)
template <bool FROM_OFFSETS>
__global__ void __launch_bounds__(1024) k_scan(const uint32_t* in, uint32_t nb, uint32_t L, uint32_t* off,
                                             uint32_t* cursor) {
    __shared__ uint32_t sh[1024];
    uint32_t tid = threadIdx.x;
    uint32_t per = (nb + 1023u) / 1024u;
    uint32_t b0 = tid * per, b1 = min(nb, b0 + per);
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; b++) {
        uint32_t c = FROM_OFFSETS ? (in[b + 1] - in[b] + L - 1u) / L : in[b];
        sum += c;
    }
    sh[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = tid >= d ? sh[tid - d] : 0;
        __syncthreads();
        sh[tid] += v;
        __syncthreads();
    }
    uint32_t run = sh[tid] - sum;
    for (uint32_t b = b0; b < b1; b++) {
        uint32_t c = FROM_OFFSETS ? (in[b + 1] - in[b] + L - 1u) / L : in[b];
        off[b] = run;
        if (cursor) cursor[b] = run;
        run += c;
    }
    if (tid == 1023) off[nb] = sh[1023];
}

// *out = max(*out, max_i v[i]): the fullest bucket of a launch set decides how many fold levels it needs
__global__ void __launch_bounds__(256) k_max_u32(const uint32_t* __restrict__ v, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = max(m, v[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// largest b with off[b] <= t (buckets without segments are skipped automatically)
static __device__ __forceinline__ uint32_t find_bucket(const uint32_t* off, uint32_t nb, uint32_t t) {
    uint32_t lo = 0, hi = nb;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// largest b in [lo, hi) with off[b] <= t
static __device__ __forceinline__ uint32_t find_bucket_in(const uint32_t* off, uint32_t lo, uint32_t hi, uint32_t t) {
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}
// blk[j] = bucket of segment j * TPB (the first lane of workgroup j of the accumulate kernels).  A lane's own search
// then runs between blk[j] and blk[j + 1] -- a handful of probes into lines its neighbours touch too -- instead of
// ~19 dependent probes across the whole offset array (measured: ~10 us at the head of every ~300 us segment).
__global__ void __launch_bounds__(256) k_block_buckets(const uint32_t* __restrict__ off, uint32_t nb, uint32_t nblk, uint32_t* __restrict__ blk) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j > nblk) return;
    uint64_t t = (uint64_t)j * TPB;
    blk[j] = (j < nblk && t < off[nb]) ? find_bucket(off, nb, (uint32_t)t) : nb - 1;
}
static __device__ __forceinline__ uint32_t find_bucket_blk(const uint32_t* __restrict__ off, uint32_t nb, const uint32_t* __restrict__ blk,
                                                           uint32_t t) {
    uint32_t lo = blk[blockIdx.x], hi = min(nb, blk[blockIdx.x + 1] + 1u);
    return find_bucket_in(off, lo, hi, t);
}

// ------------------------------------------------------------------ segments in order of their length
// A wave runs as long as its longest lane.  Segments follow bucket order, and bucket fill is Poisson: on uniform Fr
// digits (512 per bucket, 8-9 segments of 57..64) a wave idles 4.6 % of its lane-iterations, on the 32- and
// 64-reference buckets of u16 / u32 columns 30 % (computed from the distributions; confirmed by the timings).
// So the gather kernel walks the buckets in order of their segment length `per` (a counting sort over <= 65 keys;
// order inside a key is irrelevant): the 64 lanes of a wave then run the same number of additions.  Outputs keep
// their bucket-order slots, so the fold levels are unchanged.
static constexpr uint32_t SEGKEYS = 128;
static __device__ __forceinline__ uint32_t seg_key(const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ off_seg, uint32_t b) {
    uint32_t nseg = off_seg[b + 1] - off_seg[b];
    if (nseg == 0) return 0;
    uint32_t cnt = off_in[b + 1] - off_in[b];
    uint32_t per = (cnt + nseg - 1) / nseg;
    return per < SEGKEYS ? per : SEGKEYS - 1;
}
__global__ void __launch_bounds__(1024) k_seg_hist(const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ off_seg, uint32_t nb,
                                                uint32_t* __restrict__ bins) {
    __shared__ uint32_t lh[SEGKEYS];
    if (threadIdx.x < SEGKEYS) lh[threadIdx.x] = 0;
    __syncthreads();
    uint32_t b = blockIdx.x * 1024 + threadIdx.x;
    if (b < nb) atomicAdd(&lh[seg_key(off_in, off_seg, b)], 1u);
    __syncthreads();
    if (threadIdx.x < SEGKEYS && lh[threadIdx.x]) atomicAdd(&bins[threadIdx.x], lh[threadIdx.x]);
}
// bins[k] <- first position of key k; longest segments first, so the tail of the launch is made of short lanes
__global__ void k_seg_bins(uint32_t* bins) {
    uint32_t run = 0;
    for (int k = (int)SEGKEYS - 1; k >= 0; k--) {
        uint32_t c = bins[k];
        bins[k] = run;
        run += c;
    }
}
__global__ void __launch_bounds__(1024) k_seg_perm(const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ off_seg, uint32_t nb,
                                                uint32_t* __restrict__ bins, uint32_t* __restrict__ perm) {
    __shared__ uint32_t lh[SEGKEYS], base[SEGKEYS];
    if (threadIdx.x < SEGKEYS) lh[threadIdx.x] = 0;
    __syncthreads();
    uint32_t b = blockIdx.x * 1024 + threadIdx.x;
    uint32_t key = 0, rank = 0;
    if (b < nb) {
        key = seg_key(off_in, off_seg, b);
        rank = atomicAdd(&lh[key], 1u);
    }
    __syncthreads();
    if (threadIdx.x < SEGKEYS && lh[threadIdx.x]) base[threadIdx.x] = atomicAdd(&bins[threadIdx.x], lh[threadIdx.x]);
    __syncthreads();
    if (b < nb) perm[base[key] + rank] = b;
}

// ------------------------------------------------------------------ bucket accumulation
// level 0: gather affine SRS/table points by reference -- THE dominant kernel of the whole path.
// Segment t of the balanced split: the bucket's cnt items go to its nseg = ceil(cnt / L) segments in equal shares
// (ceil(cnt / nseg) each) instead of nseg-1 full segments and a short one, so the lanes of a wave run the same
// number of iterations (the short remainders idled ~6 % of the lanes).
// permuted walk: lane t is segment (t - offp[pos]) of bucket perm[pos], pos = the position whose range of offp holds t;
// `slot` = where the segment's partial sum goes (bucket order: off_out[b] + s)
static __device__ __forceinline__ void segment_range_perm(const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ off_out,
                                                          const uint32_t* __restrict__ offp, const uint32_t* __restrict__ perm, uint32_t nb, uint32_t t,
                                                          const uint32_t* __restrict__ blk, uint32_t& begin, uint32_t& end, uint32_t& slot) {
    uint32_t pos = blk ? find_bucket_blk(offp, nb, blk, t) : find_bucket(offp, nb, t);
    uint32_t b = perm[pos];
    uint32_t s = t - offp[pos];
    uint32_t cnt = off_in[b + 1] - off_in[b];
    uint32_t nseg = off_out[b + 1] - off_out[b];
    uint32_t per = (cnt + nseg - 1) / nseg;
    begin = off_in[b] + min(s * per, cnt);
    end = off_in[b] + min((s + 1) * per, cnt);
    slot = off_out[b] + s;
}
static __device__ __forceinline__ void segment_range(const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ off_out, uint32_t nb,
                                                     uint32_t t, uint32_t& begin, uint32_t& end, const uint32_t* __restrict__ blk = nullptr) {
    uint32_t b = blk ? find_bucket_blk(off_out, nb, blk, t) : find_bucket(off_out, nb, t);
    uint32_t s = t - off_out[b];
    uint32_t cnt = off_in[b + 1] - off_in[b];
    uint32_t nseg = off_out[b + 1] - off_out[b];
    uint32_t per = (cnt + nseg - 1) / nseg;
    begin = off_in[b] + min(s * per, cnt);
    end = off_in[b] + min((s + 1) * per, cnt);
}
// saturated-limb (8 x 32) gather: every special case handled (P == +-Q, infinity)
static __device__ __forceinline__ g1_xyzz gather_segment(const g1_affine* __restrict__ table, const uint32_t* __restrict__ refs, uint32_t begin,
                                                         uint32_t end) {
    g1_xyzz acc = G1::identity();
    // (measured: software-pipelining the gather costs 16 more VGPRs -> 3 waves/SIMD and gains nothing; with
    // 4 waves per SIMD the 64-byte HBM gathers already hide behind the ~3600-instruction mixed additions)
    for (uint32_t e = begin; e < end; e++) {
        uint32_t ref = refs[e];
        g1_affine p = affine_load(table + (ref & 0x7fffffffu));
        if (ref >> 31) p.y = Fq::neg(p.y);
        acc = G1::add_mixed(acc, p);
    }
    return acc;
}
__global__ void __launch_bounds__(TPB) k_msm_accum0(const g1_affine* __restrict__ table, const uint32_t* __restrict__ refs,
                                                 const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ off_out,
                                                 uint32_t nb, uint32_t L, g1_xyzz* __restrict__ out) {
    uint32_t t = blockIdx.x * TPB + threadIdx.x;
    uint32_t total = off_out[nb];
    if (t >= total) return;
    (void)L;
    uint32_t begin, end;
    segment_range(off_in, off_out, nb, t, begin, end);
    xyzz_store(out + t, gather_segment(table, refs, begin, end));
}
// The same gather in 9 x 29-bit unsaturated limbs (fq9.hip.hpp): no carry instructions in the products, no
// conditional subtractions anywhere.  A segment that meets P == +-Q (repeated SRS points, cancelling digits) is
// queued in `exc` ([0] = count, [1..] = segment ids) and redone by k_msm_accum0_fix with the saturated formulas.
template <bool CHECK_INF, bool PF = false>
__global__ void __launch_bounds__(TPB) k_msm_accum0_f9(const g1_affine* __restrict__ table, const uint32_t* __restrict__ refs,
                                                    const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ off_out,
                                                    uint32_t nb, g1_xyzz* __restrict__ out, uint32_t* __restrict__ exc,
                                                    const uint32_t* __restrict__ blk, const uint32_t* __restrict__ offp,
                                                    const uint32_t* __restrict__ perm) {
    uint32_t t = blockIdx.x * TPB + threadIdx.x;
    uint32_t total = off_out[nb];
    if (t >= total) return;
    uint32_t begin, end, slot;
    segment_range_perm(off_in, off_out, offp, perm, nb, t, blk, begin, end, slot);
    xyzz9 acc;
    bool have = false;
    uint32_t ref_next = PF && begin < end ? refs[begin] : 0u;
    for (uint32_t e = begin; e < end; e++) {
        uint32_t ref = PF ? ref_next : refs[e];
        if (PF && e + 1 < end) ref_next = refs[e + 1];  // one iteration ahead: the point gather no longer waits for its index
        g1_affine p = affine_load(table + (ref & 0x7fffffffu));
        if (CHECK_INF && G1::is_inf(p)) continue;  // tables without a point at infinity (the usual SRS) skip the test
        f9 qx = f9_from_fe(p.x), qy = f9_from_fe(p.y);
        if (ref >> 31) {  // negative digit: -y as 2p - y, limb-wise (limbs < 2^30, value < 2p: fine as a product operand)
#pragma unroll
            for (int i = 0; i < 9; i++) qy.l[i] = F9_C2[i] - qy.l[i];
        }
        if (!have) {
            acc = xyzz9_from_affine(qx, qy);
            have = true;
        } else if (!madd9(acc, qx, qy)) {
            exc[1 + atomicAdd(exc, 1u)] = t;
            return;
        }
    }
    xyzz_store(out + slot, have ? xyzz9_to_xyzz(acc) : G1::identity());
}
__global__ void __launch_bounds__(TPB) k_msm_accum0_fix(const g1_affine* __restrict__ table, const uint32_t* __restrict__ refs,
                                                     const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ off_out,
                                                     uint32_t nb, g1_xyzz* __restrict__ out, const uint32_t* __restrict__ exc,
                                                     const uint32_t* __restrict__ offp, const uint32_t* __restrict__ perm) {
    uint32_t n = exc[0];
    for (uint32_t q = blockIdx.x * TPB + threadIdx.x; q < n; q += gridDim.x * TPB) {  // every lane reaches the end of the queue
        uint32_t t = exc[1 + q];
        uint32_t begin, end, slot;
        segment_range_perm(off_in, off_out, offp, perm, nb, t, nullptr, begin, end, slot);
        xyzz_store(out + slot, gather_segment(table, refs, begin, end));
    }
}

// level >= 1: fold partial sums
__global__ void __launch_bounds__(TPB) k_msm_accumN(const g1_xyzz* __restrict__ in, const uint32_t* __restrict__ off_in,
                                                 const uint32_t* __restrict__ off_out, uint32_t nb, uint32_t L,
                                                 g1_xyzz* __restrict__ out, const uint32_t* __restrict__ blk) {
    uint32_t t = blockIdx.x * TPB + threadIdx.x;
    uint32_t total = off_out[nb];
    if (t >= total) return;
    uint32_t b = find_bucket_blk(off_out, nb, blk, t);
    uint32_t s = t - off_out[b];
    // balanced split: the bucket's cnt items go to its nseg = ceil(cnt / L) segments in equal shares
    // (ceil(cnt / nseg) each) instead of nseg-1 full segments and a short one, so the lanes of a wave run
    // the same number of iterations (the short remainders idled ~6 % of the lanes)
    uint32_t cnt = off_in[b + 1] - off_in[b];
    uint32_t nseg = off_out[b + 1] - off_out[b];
    uint32_t per = (cnt + nseg - 1) / nseg;
    (void)L;
    uint32_t begin = off_in[b] + min(s * per, cnt);
    uint32_t end = off_in[b] + min((s + 1) * per, cnt);
    g1_xyzz acc = G1::identity();
    if (begin < end) acc = xyzz_load(in + begin);
    for (uint32_t e = begin + 1; e < end; e++) acc = G1::add(acc, xyzz_load(in + e));
    xyzz_store(out + t, acc);
}

// ------------------------------------------------------------------ bucket reduction: sum_k (k+1) B_k
static __device__ __forceinline__ g1_xyzz bucket_value(const g1_xyzz* items, const uint32_t* off, uint32_t b) {
    uint32_t o = off[b];
    if (off[b + 1] > o) return xyzz_load(items + o);
    return G1::identity();
}

// dense copy of the per-bucket sums of one launch set into the batch-wide [group][bucket] array
__global__ void __launch_bounds__(TPB) k_msm_gather_buckets(const g1_xyzz* __restrict__ items, const uint32_t* __restrict__ off,
                                                         uint32_t nb, g1_xyzz* __restrict__ dense) {
    uint32_t b = blockIdx.x * TPB + threadIdx.x;
    if (b >= nb) return;
    xyzz_store(dense + b, bucket_value(items, off, b));
}

__global__ void __launch_bounds__(TPB) k_msm_reduce_chunks(const g1_xyzz* __restrict__ dense, uint32_t ngroups,
                                                        g1_xyzz* __restrict__ chunk_out) {
    uint32_t t = blockIdx.x * TPB + threadIdx.x;
    const uint32_t cpg = NB / CH;
    if (t >= ngroups * cpg) return;
    uint32_t g = t / cpg, c = t % cpg;
    uint32_t k0 = c * CH;
    g1_xyzz run = G1::identity(), acc = G1::identity();
    for (int j = CH - 1; j >= 0; j--) {
        run = G1::add(run, xyzz_load(dense + (size_t)g * NB + k0 + (uint32_t)j));
        acc = G1::add(acc, run);
    }
    // acc = sum_j (j+1) B_{k0+j};  add k0 * run with k0 = c << 4
    if (c != 0 && !G1::is_identity(run)) {
        g1_xyzz m = G1::identity();
        for (int bit = 10; bit >= 0; bit--) {
            m = G1::dbl(m);
            if ((c >> bit) & 1u) m = G1::add(m, run);
        }
        for (int d = 0; d < 4; d++) m = G1::dbl(m);
        acc = G1::add(acc, m);
    }
    xyzz_store(chunk_out + t, acc);
}

__global__ void __launch_bounds__(TPB) k_msm_reduce_groups(const g1_xyzz* __restrict__ chunk, g1_xyzz* __restrict__ grp) {
    __shared__ g1_xyzz sh[TPB];
    const uint32_t cpg = NB / CH;
    uint32_t g = blockIdx.x, tid = threadIdx.x;
    g1_xyzz acc = G1::identity();
    for (uint32_t c = tid; c < cpg; c += TPB) acc = G1::add(acc, xyzz_load(chunk + (size_t)g * cpg + c));
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t s = TPB / 2; s > 0; s >>= 1) {
        if (tid < s) {
            acc = G1::add(acc, sh[tid + s]);
            sh[tid] = acc;
        }
        __syncthreads();
    }
    if (tid == 0) xyzz_store(grp + g, acc);
}

// Horner over the G window groups of each polynomial (G = 1 with a window table) + to_affine
__global__ void k_msm_finalize(const g1_xyzz* __restrict__ grp, uint32_t G, uint32_t npoly, g1_affine* __restrict__ out) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npoly) return;
    g1_xyzz acc = xyzz_load(grp + (size_t)p * G + (G - 1));
    for (int w = (int)G - 2; w >= 0; w--) {
        for (int d = 0; d < 16; d++) acc = G1::dbl(acc);
        acc = G1::add(acc, xyzz_load(grp + (size_t)p * G + w));
    }
    affine_store(out + p, G1::to_affine(acc));
}

// ------------------------------------------------------------------ SRS kernels
// table[w][i] = 2^16 * table[w-1][i]
__global__ void __launch_bounds__(TPB) k_precompute_window(const g1_affine* __restrict__ prev, g1_affine* __restrict__ next, size_t n) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    g1_xyzz acc = G1::dbl_affine(affine_load(prev + i));
    for (int d = 1; d < 16; d++) acc = G1::dbl(acc);
    affine_store(next + i, G1::to_affine(acc));
}

// out[i] = s_i * g  (fixed-base, double-and-add)
__global__ void __launch_bounds__(TPB) k_fixed_base_mul(const fe* __restrict__ scalars, g1_affine g, g1_affine* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    fe s = Fr::from_mont(fe_load(scalars + i));
    g1_xyzz acc = G1::identity();
    for (int k = 7; k >= 0; k--) {
        uint32_t w = s.l[k];
        for (int b = 31; b >= 0; b--) {
            acc = G1::dbl(acc);
            if ((w >> b) & 1u) acc = G1::add_mixed(acc, g);
        }
    }
    affine_store(out + i, G1::to_affine(acc));
}

__global__ void __launch_bounds__(TPB) k_pair_sums(const g1_affine* __restrict__ in, g1_affine* __restrict__ out, size_t nout) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= nout) return;
    g1_xyzz acc = G1::from_affine(affine_load(in + 2 * i));
    acc = G1::add_mixed(acc, affine_load(in + 2 * i + 1));
    affine_store(out + i, G1::to_affine(acc));
}

// ------------------------------------------------------------------ host orchestration
static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// *flag = 1 if any of the n points is the point at infinity, (0, 0)
__global__ void __launch_bounds__(TPB) k_any_infinity(const g1_affine* __restrict__ pts, size_t n, uint32_t* __restrict__ flag) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n && G1::is_inf(affine_load(pts + i))) atomicOr(flag, 1u);
}
static void build_window_table(cozk_ctx* ctx, cozk_bases* b) {
    for (int w = 1; w < b->nwin; w++) {
        k_precompute_window<<<cdiv(b->n, TPB), TPB, 0, ctx->stream>>>(b->table + (size_t)(w - 1) * b->n,
                                                                     b->table + (size_t)w * b->n, b->n);
    }
    HIP_TRY(hipGetLastError());
    // 2^(16 w) P is the point at infinity only if P is (prime-order group): checking the points themselves covers the table
    ctx->scratch.reserve(64);
    uint32_t* flag = ctx->scratch.as<uint32_t>();
    HIP_TRY(hipMemsetAsync(flag, 0, 4, ctx->stream));
    k_any_infinity<<<cdiv(b->n, TPB), TPB, 0, ctx->stream>>>(b->table, b->n, flag);
    uint32_t h = 1;
    HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    b->has_inf = h != 0;
}

#define KIND_DISPATCH(kind, CALL)                                              \
    switch (kind) {                                                            \
        case COZK_SCALAR_FR: { constexpr int K = COZK_SCALAR_FR; CALL; } break;   \
        case COZK_SCALAR_U8: { constexpr int K = COZK_SCALAR_U8; CALL; } break;   \
        case COZK_SCALAR_U16: { constexpr int K = COZK_SCALAR_U16; CALL; } break; \
        case COZK_SCALAR_U32: { constexpr int K = COZK_SCALAR_U32; CALL; } break; \
        case COZK_SCALAR_U64: { constexpr int K = COZK_SCALAR_U64; CALL; } break; \
        case COZK_SCALAR_I64: { constexpr int K = COZK_SCALAR_I64; CALL; } break; \
        default: throw CozkError(COZK_ERR_INVALID_ARG, "unknown scalar kind");    \
    }

// One launch set = P MSMs (polynomial p runs over bases[offsets[p] .. offsets[p]+ns[p]); scalars[p] = device
// pointer of kinds[p]) in two phases that msm_batch pipelines on two streams:
//   msm_sort        digits -> histogram -> offsets -> sorted point references (LDS atomics, HBM writes)
//   msm_accumulate  gather + fold levels -> one XYZZ sum per bucket in `dense_out` (integer ALU)
// The phases stress different units, so the sort of set k+1 hides behind the accumulation of set k.
struct MsmSetPlan {
    uint32_t P = 0, G = 1, nb = 0, L0 = 8;
    uint64_t M = 0, bound = 0, maxseg0 = 0;
    bool pre = false;
};

static MsmSetPlan msm_plan(const cozk_bases* bases, const size_t* offsets, const size_t* ns, const int* kinds, size_t P) {
    COZK_REQUIRE(P >= 1, "msm: empty batch");
    COZK_REQUIRE((uint64_t)bases->n * (uint64_t)bases->nwin < (1ull << 31), "msm: table too large for 31-bit refs");
    MsmSetPlan pl;
    pl.P = (uint32_t)P;
    pl.pre = bases->nwin == 16;
    pl.G = pl.pre ? 1u : 16u;
    pl.nb = (uint32_t)P * pl.G * NB;
    for (size_t p = 0; p < P; p++) {
        COZK_REQUIRE(offsets[p] + ns[p] <= bases->n, "msm: base slice out of range");
        uint64_t m = (uint64_t)ns[p] * kind_nwin(kinds[p]);
        pl.M += m;
        if (m > pl.bound) pl.bound = m;
    }
    COZK_REQUIRE(pl.M < (1ull << 32), "msm: batch too large (reference count exceeds 32 bits)");
    // segment length of level 0: aim at >= ~2^18 lanes, clamp to [8, 127] (measured: 256-long segments leave
    // too few waves per SIMD and run the gather kernel 14 % slower)
    uint32_t L0 = (uint32_t)(pl.M >> 18);
    if (L0 < 8) L0 = 8;
    if (L0 > 127) L0 = 127;  // < SEGKEYS; measured with length-ordered segments: 127 beats 64 by 2 ms per proof (fewer partial sums)
    pl.L0 = L0;
    pl.maxseg0 = pl.M / L0 + pl.nb;
    return pl;
}

static void msm_scan(hipStream_t st, uint32_t* bsums, uint32_t nb, bool from_offsets, const uint32_t* in, uint32_t L, uint32_t* off, uint32_t* cursor,
                     const uint32_t* perm = nullptr) {
    const uint32_t nscan_blocks = (nb + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    if (from_offsets) {
        k_scan_sums<true><<<nscan_blocks, 256, 0, st>>>(in, nb, L, bsums, perm);
        k_scan_top<<<1, 1024, 0, st>>>(bsums, nscan_blocks, off + nb);
        k_scan_final<true><<<nscan_blocks, 256, 0, st>>>(in, nb, L, bsums, off, cursor, perm);
    } else {
        k_scan_sums<false><<<nscan_blocks, 256, 0, st>>>(in, nb, L, bsums, nullptr);
        k_scan_top<<<1, 1024, 0, st>>>(bsums, nscan_blocks, off + nb);
        k_scan_final<false><<<nscan_blocks, 256, 0, st>>>(in, nb, L, bsums, off, cursor, nullptr);
    }
}

// sort phase on stream `st` into the sort workspace `sw`
static void msm_sort(cozk_ctx* ctx, hipStream_t st, MsmSortWs& sw, const MsmSetPlan& pl, const cozk_bases* bases, const size_t* offsets,
                     const size_t* ns, const void* const* scalars, const int* kinds) {
    const uint32_t nb = pl.nb, G = pl.G;
    const size_t P = pl.P;
    sw.hist.reserve((size_t)(nb + 2) * 4);
    sw.off0.reserve((size_t)(nb + 1) * 4);
    sw.offA.reserve((size_t)(nb + 1) * 4);
    sw.refs.reserve((size_t)pl.M * 4);
    uint32_t* hist = sw.hist.as<uint32_t>();
    uint32_t* off0 = sw.off0.as<uint32_t>();
    uint32_t* refs = sw.refs.as<uint32_t>();
    const uint32_t nscan_blocks = (nb + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    sw.ptrs.reserve((size_t)nscan_blocks * 4 + P * sizeof(MsmPolyDesc) + 64);
    uint32_t* bsums = sw.ptrs.as<uint32_t>();
    MsmPolyDesc* d_descs = reinterpret_cast<MsmPolyDesc*>(((uintptr_t)(bsums + nscan_blocks) + 15) & ~(uintptr_t)15);
    HIP_TRY(hipMemsetAsync(hist, 0, (size_t)(nb + 2) * 4, st));
    uint32_t* d_maxcnt = hist + nb + 1;  // not touched by the scans (they own hist[0..nb])
    if (!sw.max_pinned) HIP_TRY(hipHostMalloc((void**)&sw.max_pinned, 64 + 64 * sizeof(MsmPolyDesc), hipHostMallocDefault));
    if (!sw.max_event) HIP_TRY(hipEventCreateWithFlags(&sw.max_event, hipEventDisableTiming));
    // after the histogram: fullest bucket -> pinned host word, behind an event the host waits on only when it
    // sizes the fold levels (by then the GPU is inside the long gather pass, so the queue never drains)
    auto read_back_max = [&] {
        k_max_u32<<<std::min<uint32_t>(cdiv(nb, 256), 1024u), 256, 0, st>>>(hist, nb, d_maxcnt);
        HIP_TRY(hipMemcpyAsync(sw.max_pinned, d_maxcnt, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(sw.max_event, st));
    };
    if (pl.pre) {
        // LDS-privatised counting sort: one launch per scalar kind, grid = (workgroups per polynomial, polynomials).
        // The descriptors travel through pinned memory owned by the workspace (no host sync here).
        MsmPolyDesc* h_descs = reinterpret_cast<MsmPolyDesc*>(reinterpret_cast<char*>(sw.max_pinned) + 64);
        COZK_REQUIRE(P <= 64, "msm: too many polynomials in one launch set");
        uint32_t nd = 0;
        std::vector<std::pair<int, std::pair<uint32_t, uint32_t>>> runs;  // kind -> (first desc, count)
        size_t max_n = 0;
        for (int kind = 0; kind <= COZK_SCALAR_I64; kind++) {
            uint32_t first = nd;
            for (size_t p = 0; p < P; p++)
                if (kinds[p] == kind && ns[p]) {
                    h_descs[nd++] = MsmPolyDesc{scalars[p], (uint32_t)ns[p], (uint32_t)offsets[p], (uint32_t)p, 0};
                    if (ns[p] > max_n) max_n = ns[p];
                }
            if (nd > first) runs.push_back({kind, {first, nd - first}});
        }
        HIP_TRY(hipMemcpyAsync(d_descs, h_descs, nd * sizeof(MsmPolyDesc), hipMemcpyHostToDevice, st));
        // 256-thread workgroups (one wave per SIMD, 128 KiB of LDS each): narrow enough to slip into the CU slots that free
        // up under the register-heavy gather kernel of the previous launch set, so the sort really runs beside it (wider
        // workgroups wait for the gather's grid to drain: measured 127.8 -> 123.5 ms per proof at 2^20, same box)
        static const int ltpb = getenv("COZK_MSM_LTPB") ? atoi(getenv("COZK_MSM_LTPB")) : 256;
        static const uint32_t wgs_max = getenv("COZK_MSM_WGS") ? (uint32_t)atoi(getenv("COZK_MSM_WGS")) : 64u;
        COZK_REQUIRE(ltpb >= 64 && ltpb <= LTPB && ltpb % 64 == 0, "COZK_MSM_LTPB out of range");
        COZK_REQUIRE(wgs_max >= 1 && wgs_max <= 1024, "COZK_MSM_WGS out of range (1..1024)");
        uint32_t wgs = (uint32_t)((max_n + 4095) / 4096);  // >= 4096 scalars per workgroup (measured: 64 beats 16 workgroups 2x)
        if (wgs < 1) wgs = 1;
        if (wgs > wgs_max) wgs = wgs_max;
        for (auto& r : runs) {
            dim3 grid(wgs, r.second.second);
            KIND_DISPATCH(r.first, (k_msm_hist_lds<K><<<grid, ltpb, 0, st>>>(d_descs + r.second.first, hist)));
        }
        read_back_max();
        msm_scan(st, bsums, nb, false, hist, 1, off0, hist);  // hist doubles as the scatter cursor after the scan
        for (auto& r : runs) {
            dim3 grid(wgs, r.second.second);
            // algorithmic bytes of the placement: every scalar read once + one 4-byte reference written per 16-bit window
            uint64_t alg = 0;
            for (uint32_t q = 0; q < r.second.second; q++) {
                const MsmPolyDesc& d = h_descs[r.second.first + q];
                const uint64_t sb = scalar_kind_bytes(r.first);
                alg += (uint64_t)d.n * (sb + 4ull * ((sb * 8 + 15) / 16));
            }
            ProfScope prof(ctx, COZK_PROF_MSM_SCATTER, alg, st);
            KIND_DISPATCH(r.first, (k_msm_scatter_lds<K><<<grid, ltpb, 0, st>>>(d_descs + r.second.first, hist, refs, (uint32_t)bases->n)));
        }
    } else {
        for (size_t p = 0; p < P; p++) {
            uint32_t* h = hist + (size_t)p * G * NB;
            if (ns[p]) KIND_DISPATCH(kinds[p], (k_msm_hist<K><<<cdiv(ns[p], TPB), TPB, 0, st>>>(scalars[p], ns[p], h, 1)));
        }
        read_back_max();
        msm_scan(st, bsums, nb, false, hist, 1, off0, hist);
        for (size_t p = 0; p < P; p++) {
            uint32_t* cur = hist + (size_t)p * G * NB;
            if (ns[p])
                KIND_DISPATCH(kinds[p], (k_msm_scatter<K><<<cdiv(ns[p], TPB), TPB, 0, st>>>(scalars[p], ns[p], cur, refs, 1, (uint32_t)bases->n,
                                                                                         (uint32_t)offsets[p])));
        }
    }
    msm_scan(st, bsums, nb, true, off0, pl.L0, sw.offA.as<uint32_t>(), nullptr);  // level-0 segment offsets
    HIP_TRY(hipGetLastError());
}

// accumulate phase on the context's stream; reads the sort workspace `sw` (its producer must have been waited for)
static void msm_accumulate(cozk_ctx* ctx, MsmSortWs& sw, const MsmSetPlan& pl, const cozk_bases* bases, const size_t* ns, const int* kinds,
                           g1_xyzz* dense_out) {
    MsmWorkspace& ws = ctx->msm_ws;
    hipStream_t st = ctx->stream;
    const uint32_t nb = pl.nb, L0 = pl.L0;
    // upper levels fold <= L1 partial sums per lane.  The chain is serial (a full XYZZ addition per step, ~25 us
    // when a wave runs alone), so it is kept short: 64-long chains made the fold of the one heavy bucket of a
    // u16 / u32 / flag column (the carry digit: n/2 references) cost as much as its whole gather pass.
    const uint32_t L1 = 8;
    const uint64_t maxseg0 = pl.maxseg0;
    // segment counts obey x_{k+1} = x_k / L1 + nb <= max(x_0, 2 nb): size both ping-pong buffers for that
    const uint64_t maxpart = maxseg0 > 2ull * nb ? maxseg0 : 2ull * nb;
    ws.partA.reserve((size_t)maxpart * sizeof(g1_xyzz));
    ws.partB.reserve((size_t)maxpart * sizeof(g1_xyzz));
    ws.offB.reserve((size_t)(nb + 1) * 4);
    ws.offC.reserve((size_t)(nb + 1) * 4);
    ws.ptrs.reserve((size_t)((nb + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK) * 4 + 64);
    uint32_t* refs = sw.refs.as<uint32_t>();
    uint32_t* off0 = sw.off0.as<uint32_t>();
    uint32_t* offA = sw.offA.as<uint32_t>();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->prof_enabled) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, st));
    }
    ws.blk.reserve((size_t)(cdiv(maxpart, TPB) + 2) * 4);
    uint32_t* blk = ws.blk.as<uint32_t>();
    if (pl.pre) {
        ws.exc.reserve((size_t)(maxseg0 + 2) * 4);
        uint32_t* exc = ws.exc.as<uint32_t>();
        HIP_TRY(hipMemsetAsync(exc, 0, 4, st));
        // walk the buckets in order of their segment length (see "segments in order of their length")
        ws.perm.reserve((size_t)nb * 4);
        ws.offP.reserve((size_t)(nb + 1) * 4 + SEGKEYS * 4);
        uint32_t* perm = ws.perm.as<uint32_t>();
        uint32_t* offP = ws.offP.as<uint32_t>();
        uint32_t* bins = offP + nb + 1;
        HIP_TRY(hipMemsetAsync(bins, 0, SEGKEYS * 4, st));
        k_seg_hist<<<cdiv(nb, 1024), 1024, 0, st>>>(off0, offA, nb, bins);
        k_seg_bins<<<1, 1, 0, st>>>(bins);
        k_seg_perm<<<cdiv(nb, 1024), 1024, 0, st>>>(off0, offA, nb, bins, perm);
        msm_scan(st, ws.ptrs.as<uint32_t>(), nb, true, offA, 1, offP, nullptr, perm);  // segment offsets in walking order
        const uint32_t nblk0 = cdiv(maxseg0, TPB);
        k_block_buckets<<<cdiv(nblk0 + 1, 256), 256, 0, st>>>(offP, nb, nblk0, blk);
        if (bases->has_inf)
            k_msm_accum0_f9<true><<<nblk0, TPB, 0, st>>>(bases->table, refs, off0, offA, nb, ws.partA.as<g1_xyzz>(), exc, blk, offP, perm);
        else  // index prefetch on (same-box A/B: -0.4 ms per proof, one more VGPR)
            k_msm_accum0_f9<false, true><<<nblk0, TPB, 0, st>>>(bases->table, refs, off0, offA, nb, ws.partA.as<g1_xyzz>(), exc, blk, offP, perm);
        k_msm_accum0_fix<<<256, TPB, 0, st>>>(bases->table, refs, off0, offA, nb, ws.partA.as<g1_xyzz>(), exc, offP, perm);
    } else {
        k_msm_accum0<<<cdiv(maxseg0, TPB), TPB, 0, st>>>(bases->table, refs, off0, offA, nb, L0, ws.partA.as<g1_xyzz>());
    }
    if (ctx->prof_enabled) {
        HIP_TRY(hipEventRecord(e1, st));
        ctx->prof_events.push_back({e0, e1});
        ctx->prof_launches += 1;
        ctx->prof_units += pl.M;
        for (size_t p = 0; p < pl.P; p++) ctx->prof_alg_bytes += (uint64_t)ns[p] * (64 + scalar_kind_bytes(kinds[p]));
    }
    // further levels until the fullest bucket is down to one value
    HIP_TRY(hipEventSynchronize(sw.max_event));
    uint64_t fullest = *(volatile uint32_t*)sw.max_pinned;
    COZK_REQUIRE(fullest <= pl.bound, "msm: histogram larger than the reference count");
    uint64_t cnt = (fullest + L0 - 1) / L0;
    uint64_t maxseg = maxseg0;
    g1_xyzz* cur_items = ws.partA.as<g1_xyzz>();
    g1_xyzz* nxt_items = ws.partB.as<g1_xyzz>();
    // level offsets ping-pong between two buffers of the accumulate workspace (offA belongs to the sort phase)
    uint32_t* cur_off = offA;
    uint32_t* nxt_off = ws.offB.as<uint32_t>();
    uint32_t* spare_off = ws.offC.as<uint32_t>();
    while (cnt > 1) {
        uint64_t nseg = maxseg / L1 + nb;
        msm_scan(st, ws.ptrs.as<uint32_t>(), nb, true, cur_off, L1, nxt_off, nullptr);
        const uint32_t nblkN = cdiv(nseg, TPB);
        k_block_buckets<<<cdiv(nblkN + 1, 256), 256, 0, st>>>(nxt_off, nb, nblkN, blk);
        k_msm_accumN<<<nblkN, TPB, 0, st>>>(cur_items, cur_off, nxt_off, nb, L1, nxt_items, blk);
        std::swap(cur_items, nxt_items);
        uint32_t* done = cur_off;
        cur_off = nxt_off;
        nxt_off = (done == offA) ? spare_off : done;
        maxseg = nseg;
        cnt = (cnt + L1 - 1) / L1;
    }
    k_msm_gather_buckets<<<cdiv(nb, TPB), TPB, 0, st>>>(cur_items, cur_off, nb, dense_out);
    HIP_TRY(hipGetLastError());
}

static void affine_to_abi(const g1_affine& a, uint64_t xy[8], int* inf) {
    bool is_inf = G1::is_inf(a);
    for (int i = 0; i < 4; i++) {
        xy[i] = (uint64_t)a.x.l[2 * i] | ((uint64_t)a.x.l[2 * i + 1] << 32);
        xy[4 + i] = (uint64_t)a.y.l[2 * i] | ((uint64_t)a.y.l[2 * i + 1] << 32);
    }
    if (inf) *inf = is_inf ? 1 : 0;
}
static g1_affine abi_to_affine(const uint64_t xy[8], int inf) {
    g1_affine a;
    for (int i = 0; i < 4; i++) {
        a.x.l[2 * i] = (uint32_t)xy[i];
        a.x.l[2 * i + 1] = (uint32_t)(xy[i] >> 32);
        a.y.l[2 * i] = (uint32_t)xy[4 + i];
        a.y.l[2 * i + 1] = (uint32_t)(xy[4 + i] >> 32);
    }
    if (inf) {
        a.x = Fq::zero();
        a.y = Fq::zero();
    }
    return a;
}

// k MSMs with per-polynomial base slices; launch sets are cut so that one set holds at most 2^28 .. 2^30 point
// references (1 GiB of refs) and at most 64 (window table) / 4 (16 window groups) polynomials
void msm_batch(cozk_ctx* ctx, const cozk_bases* bases, const size_t* offsets, const size_t* ns, const void* const* scalars,
               const int* kinds, size_t k, uint64_t* out_xy, int* out_inf) {
    MsmWorkspace& ws = ctx->msm_ws;
    hipStream_t st = ctx->stream;
    const uint32_t G = bases->nwin == 16 ? 1u : 16u;
    const size_t maxP = bases->nwin == 16 ? 64 : 4;  // measured: 64 small-scalar polynomials per set beat 16 by 1.7 ms per proof
    // the bucket-reduction tail (running sums, Horner, to-affine) is latency-bound, so it runs ONCE over
    // the dense bucket sums of up to `tail_cap` polynomials instead of once per launch set
    const size_t tail_cap = 256 / G;
    ws.out.reserve(k * sizeof(g1_affine));
    g1_affine* d_out = ws.out.as<g1_affine>();
    for (size_t t0 = 0; t0 < k; t0 += tail_cap) {
        size_t kt = k - t0 < tail_cap ? k - t0 : tail_cap;
        const uint32_t ngroups = (uint32_t)kt * G;
        ws.bsum.reserve((size_t)ngroups * NB * sizeof(g1_xyzz));
        ws.chunk.reserve((size_t)ngroups * (NB / CH) * sizeof(g1_xyzz));
        ws.grp.reserve((size_t)ngroups * sizeof(g1_xyzz));
        g1_xyzz* dense = ws.bsum.as<g1_xyzz>();
        // cut the launch sets, then pipeline: sort(k + 1) on the side stream while accumulate(k) runs on the main one
        struct SetRange { size_t s, P; };
        std::vector<SetRange> sets;
        // References per launch set: a quarter of the batch, so that the sort of set k + 1 still hides behind the accumulation of set
        // k, within [2^28, 2^30] (the two sort workspaces hold 4 bytes per reference).  Measured: at 2^22 coefficients 2^30 beats 2^28
        // by 7 % of the commit (4 instead of 16 polynomials per set paid the per-set scans and the host's wait for the fullest bucket
        // 4 x as often); at 2^20 one 2^30 set would hold all 64 field-element polynomials and lose the overlap (+4 %).
        // COZK_MSM_SET_REFS_LOG2 (24..31) pins the cap for A/B runs.
        uint64_t set_refs_cap;
        {
            static const int pinned = getenv("COZK_MSM_SET_REFS_LOG2") ? atoi(getenv("COZK_MSM_SET_REFS_LOG2")) : 0;
            if (pinned) set_refs_cap = 1ull << (pinned < 24 ? 24 : (pinned > 31 ? 31 : pinned));
            else {
                uint64_t total = 0;
                for (size_t p = 0; p < kt; p++) total += (uint64_t)ns[t0 + p] * kind_nwin(kinds[t0 + p]);
                set_refs_cap = total / 4;
                if (set_refs_cap < (1ull << 28)) set_refs_cap = 1ull << 28;
                if (set_refs_cap > (1ull << 30)) set_refs_cap = 1ull << 30;
            }
        }
        {
            size_t s = 0;
            while (s < kt) {
                size_t P = 0;
                uint64_t M = 0;
                while (s + P < kt && P < maxP) {
                    uint64_t m = (uint64_t)ns[t0 + s + P] * kind_nwin(kinds[t0 + s + P]);
                    if (P > 0 && M + m > set_refs_cap) break;
                    M += m;
                    P++;
                }
                sets.push_back({s, P});
                s += P;
            }
        }
        // the first set's sort has nothing to hide behind: start with the cheapest one (fewest references)
        if (sets.size() > 2) {
            size_t best = 0;
            uint64_t best_m = ~0ull;
            for (size_t i = 0; i < sets.size(); i++) {
                uint64_t m = 0;
                for (size_t p = 0; p < sets[i].P; p++) m += (uint64_t)ns[t0 + sets[i].s + p] * kind_nwin(kinds[t0 + sets[i].s + p]);
                if (m < best_m) {
                    best_m = m;
                    best = i;
                }
            }
            std::rotate(sets.begin(), sets.begin() + best, sets.begin() + best + 1);
        }
        static const bool pipeline = getenv("COZK_MSM_SERIAL") == nullptr;
        hipStream_t side = st;
        if (pipeline && sets.size() > 1) {
            if (!ctx->stream2) {
                // the sort stream outranks the main stream: its memory-bound workgroups take the CU slots that free up
                // under the ALU-bound gather kernel instead of queueing behind that kernel's own next workgroups
                static const bool prio = getenv("COZK_MSM_NO_PRIORITY") == nullptr;
                int lo = 0, hi = 0;
                HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
                if (prio) HIP_TRY(hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, hi));
                else HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
            }
            side = ctx->stream2;
        }
        for (int b = 0; b < 2; b++) {
            if (!ws.sort[b].done) HIP_TRY(hipEventCreateWithFlags(&ws.sort[b].done, hipEventDisableTiming));
            if (!ws.sort[b].consumed) HIP_TRY(hipEventCreateWithFlags(&ws.sort[b].consumed, hipEventDisableTiming));
        }
        if (side != st) {
            // the side stream must see everything the caller enqueued on the main stream (scalars written by kernels,
            // the previous batch's reads of the sort workspaces)
            if (!ws.fork) HIP_TRY(hipEventCreateWithFlags(&ws.fork, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ws.fork, st));
            HIP_TRY(hipStreamWaitEvent(side, ws.fork, 0));
        }
        std::vector<MsmSetPlan> plans(sets.size());
        auto launch_sort = [&](size_t i) {
            const SetRange& r = sets[i];
            plans[i] = msm_plan(bases, offsets + t0 + r.s, ns + t0 + r.s, kinds + t0 + r.s, r.P);
            if (plans[i].M == 0) return;
            MsmSortWs& sw = ws.sort[i & 1];
            if (i >= 2 && side != st) HIP_TRY(hipStreamWaitEvent(side, sw.consumed, 0));  // its previous user has finished reading it
            msm_sort(ctx, side, sw, plans[i], bases, offsets + t0 + r.s, ns + t0 + r.s, scalars + t0 + r.s, kinds + t0 + r.s);
            if (side != st) HIP_TRY(hipEventRecord(sw.done, side));
        };
        if (!sets.empty()) launch_sort(0);
        for (size_t i = 0; i < sets.size(); i++) {
            if (side != st && i + 1 < sets.size()) launch_sort(i + 1);
            const SetRange& r = sets[i];
            g1_xyzz* out_i = dense + (size_t)r.s * G * NB;
            if (plans[i].M != 0) {
                MsmSortWs& sw = ws.sort[i & 1];
                if (side != st) HIP_TRY(hipStreamWaitEvent(st, sw.done, 0));
                msm_accumulate(ctx, sw, plans[i], bases, ns + t0 + r.s, kinds + t0 + r.s, out_i);
                if (side != st) HIP_TRY(hipEventRecord(sw.consumed, st));
            } else {
                HIP_TRY(hipMemsetAsync(out_i, 0, (size_t)plans[i].nb * sizeof(g1_xyzz), st));  // all buckets = identity (zz = 0)
            }
            if (side == st && i + 1 < sets.size()) launch_sort(i + 1);
        }
        k_msm_reduce_chunks<<<cdiv((uint64_t)ngroups * (NB / CH), TPB), TPB, 0, st>>>(dense, ngroups, ws.chunk.as<g1_xyzz>());
        k_msm_reduce_groups<<<ngroups, TPB, 0, st>>>(ws.chunk.as<g1_xyzz>(), ws.grp.as<g1_xyzz>());
        k_msm_finalize<<<cdiv(kt, 64), 64, 0, st>>>(ws.grp.as<g1_xyzz>(), G, (uint32_t)kt, d_out + t0);
        HIP_TRY(hipGetLastError());
    }
    g1_affine* h = reinterpret_cast<g1_affine*>(ctx_pinned(ctx, k * sizeof(g1_affine)));
    HIP_TRY(hipMemcpyAsync(h, d_out, k * sizeof(g1_affine), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t i = 0; i < k; i++) affine_to_abi(h[i], out_xy + 8 * i, out_inf ? out_inf + i : nullptr);
}

// ------------------------------------------------------------------ C ABI
extern "C" {

int cozk_bases_upload(cozk_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, size_t n, int precompute,
                      cozk_bases** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && xy && out && n > 0, "bases_upload: bad argument");
        cozk_bases* b = new cozk_bases{ctx, n, precompute ? 16 : 1, nullptr};
        try {
            HIP_TRY(hipMalloc((void**)&b->table, (size_t)b->nwin * n * sizeof(g1_affine)));
            std::vector<g1_affine> h(n);
            for (size_t i = 0; i < n; i++) h[i] = abi_to_affine(xy + 8 * i, infinity ? infinity[i] : 0);
            HIP_TRY(hipMemcpyAsync(b->table, h.data(), n * sizeof(g1_affine), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            build_window_table(ctx, b);
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        } catch (...) {
            if (b->table) (void)hipFree(b->table);
            delete b;
            throw;
        }
        *out = b;
    });
}

int cozk_bases_from_scalars(cozk_ctx* ctx, const cozk_vec* s, const uint64_t* g_xy, int precompute, cozk_bases** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && s && g_xy && out && s->kind == COZK_SCALAR_FR && s->n > 0, "bases_from_scalars: bad argument");
        size_t n = s->n;
        cozk_bases* b = new cozk_bases{ctx, n, precompute ? 16 : 1, nullptr};
        try {
            HIP_TRY(hipMalloc((void**)&b->table, (size_t)b->nwin * n * sizeof(g1_affine)));
            g1_affine g = abi_to_affine(g_xy, 0);
            k_fixed_base_mul<<<cdiv(n, TPB), TPB, 0, ctx->stream>>>(reinterpret_cast<const fe*>(s->d), g, b->table, n);
            HIP_TRY(hipGetLastError());
            build_window_table(ctx, b);
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        } catch (...) {
            if (b->table) (void)hipFree(b->table);
            delete b;
            throw;
        }
        *out = b;
    });
}

int cozk_bases_pair_sums(cozk_ctx* ctx, const cozk_bases* src, int precompute, cozk_bases** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && src && out && src->n >= 2 && src->n % 2 == 0, "bases_pair_sums: bad argument");
        size_t n = src->n / 2;
        cozk_bases* b = new cozk_bases{ctx, n, precompute ? 16 : 1, nullptr};
        try {
            HIP_TRY(hipMalloc((void**)&b->table, (size_t)b->nwin * n * sizeof(g1_affine)));
            k_pair_sums<<<cdiv(n, TPB), TPB, 0, ctx->stream>>>(src->table, b->table, n);
            HIP_TRY(hipGetLastError());
            build_window_table(ctx, b);
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        } catch (...) {
            if (b->table) (void)hipFree(b->table);
            delete b;
            throw;
        }
        *out = b;
    });
}

int cozk_bases_download(cozk_ctx* ctx, const cozk_bases* b, size_t offset, size_t n, uint64_t* xy, uint8_t* infinity) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && b && xy && offset + n <= b->n, "bases_download: bad argument");
        std::vector<g1_affine> h(n);
        HIP_TRY(hipMemcpyAsync(h.data(), b->table + offset, n * sizeof(g1_affine), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < n; i++) {
            int inf;
            affine_to_abi(h[i], xy + 8 * i, &inf);
            if (infinity) infinity[i] = (uint8_t)inf;
        }
    });
}

int cozk_bases_free(cozk_bases* b) {
    if (!b) return COZK_OK;
    if (b->table) (void)hipFree(b->table);
    delete b;
    return COZK_OK;
}

size_t cozk_bases_len(const cozk_bases* b) { return b ? b->n : 0; }

int cozk_msm_vec(cozk_ctx* ctx, const cozk_bases* bases, size_t offset, const cozk_vec* scalars, uint64_t out_xy[8],
                 int* out_infinity) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && bases && scalars && out_xy, "msm_vec: bad argument");
        const void* p = scalars->d;
        int kind = scalars->kind;
        size_t n = scalars->n;
        msm_batch(ctx, bases, &offset, &n, &p, &kind, 1, out_xy, out_infinity);
    });
}

int cozk_batch_msm_vec(cozk_ctx* ctx, const cozk_bases* bases, size_t offset, const cozk_vec* const* scalars, size_t k,
                       uint64_t* out_xy, int* out_infinity) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && bases && scalars && out_xy && k > 0, "batch_msm_vec: bad argument");
        std::vector<const void*> ptrs(k);
        std::vector<int> kinds(k);
        std::vector<size_t> offs(k, offset), ns(k);
        size_t n = scalars[0]->n;
        for (size_t i = 0; i < k; i++) {
            // "batch commit requires all batches to have the same length" (pst13.rs:310-313)
            COZK_REQUIRE(scalars[i] && scalars[i]->n == n, "batch_msm_vec: polynomials must have equal length");
            ptrs[i] = scalars[i]->d;
            kinds[i] = scalars[i]->kind;
            ns[i] = n;
        }
        msm_batch(ctx, bases, offs.data(), ns.data(), ptrs.data(), kinds.data(), k, out_xy, out_infinity);
    });
}

int cozk_batch_msm_slices(cozk_ctx* ctx, const cozk_bases* bases, const size_t* offsets, const cozk_vec* const* scalars,
                          const size_t* lens, size_t k, uint64_t* out_xy, int* out_infinity) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && bases && offsets && scalars && out_xy && k > 0, "batch_msm_slices: bad argument");
        std::vector<const void*> ptrs(k);
        std::vector<int> kinds(k);
        std::vector<size_t> ns(k);
        for (size_t i = 0; i < k; i++) {
            COZK_REQUIRE(scalars[i], "batch_msm_slices: null scalar vector");
            ns[i] = lens ? lens[i] : scalars[i]->n;
            COZK_REQUIRE(ns[i] <= scalars[i]->n, "batch_msm_slices: length exceeds the scalar vector");
            ptrs[i] = scalars[i]->d;
            kinds[i] = scalars[i]->kind;
        }
        msm_batch(ctx, bases, offsets, ns.data(), ptrs.data(), kinds.data(), k, out_xy, out_infinity);
    });
}

int cozk_msm(cozk_ctx* ctx, const cozk_bases* bases, size_t offset, const void* host_scalars, int kind, size_t n,
             uint64_t out_xy[8], int* out_infinity) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && bases && (host_scalars || n == 0) && out_xy && scalar_kind_bytes(kind), "msm: bad argument");
        size_t bytes = n * scalar_kind_bytes(kind);
        ctx->scratch2.reserve(bytes ? bytes : 16);
        if (bytes) HIP_TRY(hipMemcpyAsync(ctx->scratch2.p, host_scalars, bytes, hipMemcpyHostToDevice, ctx->stream));
        const void* p = ctx->scratch2.p;
        msm_batch(ctx, bases, &offset, &n, &p, &kind, 1, out_xy, out_infinity);
    });
}

int cozk_g1_sum(cozk_ctx* ctx, const uint64_t* xy, const int* infinity, size_t k, uint64_t out_xy[8], int* out_infinity) {
    return cozk_guard(nullptr, [&] {
        (void)ctx;
        g1_xyzz acc = G1::identity();
        for (size_t i = 0; i < k; i++) acc = G1::add_mixed(acc, abi_to_affine(xy + 8 * i, infinity ? infinity[i] : 0));
        affine_to_abi(G1::to_affine(acc), out_xy, out_infinity);
    });
}

int cozk_g1_mul(cozk_ctx* ctx, const uint64_t xy[8], int infinity, const uint64_t s[4], uint64_t out_xy[8], int* out_infinity) {
    return cozk_guard(nullptr, [&] {
        (void)ctx;
        fe sm;
        for (int i = 0; i < 4; i++) {
            sm.l[2 * i] = (uint32_t)s[i];
            sm.l[2 * i + 1] = (uint32_t)(s[i] >> 32);
        }
        fe sc = Fr::from_mont(sm);
        g1_affine p = abi_to_affine(xy, infinity);
        g1_xyzz acc = G1::identity();
        for (int k = 7; k >= 0; k--)
            for (int b = 31; b >= 0; b--) {
                acc = G1::dbl(acc);
                if ((sc.l[k] >> b) & 1u) acc = G1::add_mixed(acc, p);
            }
        affine_to_abi(G1::to_affine(acc), out_xy, out_infinity);
    });
}

int cozk_prof_enable(cozk_ctx* ctx, int on) {
    return cozk_guard(ctx, [&] {
        ctx->prof_enabled = on != 0;
        for (auto& pr : ctx->prof_events) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        ctx->prof_events.clear();
        ctx->prof_ms = 0;
        ctx->prof_launches = 0;
        ctx->prof_units = 0;
        ctx->prof_alg_bytes = 0;
        for (ProfSlot& sl : ctx->prof_slots) sl.reset();
    });
}

static const char* const PROF_KERNEL_NAMES[COZK_PROF_SLOTS] = {"k_poly_eval_chi", "k_poly_lincomb", "k_layer_bind_cubic", "k_msm_scatter_lds",
                                                               "k_layer_output"};
const char* cozk_prof_kernel_name(int slot) { return slot >= 0 && slot < COZK_PROF_SLOTS ? PROF_KERNEL_NAMES[slot] : nullptr; }
int cozk_prof_read_kernel(cozk_ctx* ctx, int slot, uint64_t* launches, double* total_ms, uint64_t* alg_bytes) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && slot >= 0 && slot < COZK_PROF_SLOTS, "prof_read_kernel: bad slot");
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->stream2) HIP_TRY(hipStreamSynchronize(ctx->stream2));
        ProfSlot& sl = ctx->prof_slots[slot];
        for (auto& pr : sl.events) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
            sl.ms += ms;
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        sl.events.clear();
        if (launches) *launches = sl.launches;
        if (total_ms) *total_ms = sl.ms;
        if (alg_bytes) *alg_bytes = sl.alg_bytes;
    });
}

int cozk_prof_read(cozk_ctx* ctx, uint64_t* launches, double* total_ms, uint64_t* point_adds, uint64_t* alg_bytes) {
    return cozk_guard(ctx, [&] {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (auto& pr : ctx->prof_events) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
            ctx->prof_ms += ms;
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        ctx->prof_events.clear();
        if (launches) *launches = ctx->prof_launches;
        if (total_ms) *total_ms = ctx->prof_ms;
        if (point_adds) *point_adds = ctx->prof_units;
        if (alg_bytes) *alg_bytes = ctx->prof_alg_bytes;
    });
}

}  // extern "C"
