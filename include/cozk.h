/*
 * cozk.h -- C ABI of libcozk, the MI355X-native engine for the sumcheck + polynomial-commitment
 * hot path of ChainSafe/co-zkvms (co-jolt / co-noir-spartan workers).
 *
 * This is the drop-in boundary.  Every entry point names the reference interface
 * it replaces (paths relative to the reference repository root).  Conventions:
 *   - plain pointers and sizes only; no C++/torch types; every function returns an int status
 *     (COZK_OK = 0, negative = error; never unwinds across the boundary);
 *     cozk_last_error(ctx) returns the message of the last failure on that context.
 *   - field elements: BN254 Fr / Fq as 4 x u64 little-endian limbs in MONTGOMERY form (R = 2^256),
 *     i.e. the in-memory layout of arkworks `Fp256<MontBackend<_,4>>` (ark-ff 0.5).
 *   - G1 points cross the boundary affine: x[4], y[4] (Fq Montgomery, 64 B) + infinity flag.
 *   - one cozk_ctx per (party, GPU); a ctx owns one HIP stream and is NOT thread-safe -- mirror of
 *     the single-owner IoContext forks (mpc-core/src/protocols/rep3/network.rs:108-119).
 *   - handles (cozk_vec / cozk_bases / cozk_poly / cozk_layer / cozk_spliteq) are device resident;
 *     only round messages (3-8 field elements), commitments and opening proofs cross PCIe.
 */
#ifndef COZK_H
#define COZK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COZK_OK 0
#define COZK_ERR_INVALID_ARG (-1)
#define COZK_ERR_HIP (-2)
#define COZK_ERR_OOM (-3)
#define COZK_ERR_INTERNAL (-4)
#define COZK_ERR_NO_DEVICE (-5)

/* scalar kinds: the variants of jolt-core `MultilinearPolynomial` that
 * `VariableBaseMSM::batch_msm` dispatches on (call site co-jolt/src/poly/commitment/pst13.rs:319-323) */
#define COZK_SCALAR_FR 0  /* LargeScalars: Fr Montgomery, 32 B */
#define COZK_SCALAR_U8 1  /* U8Scalars  (also 0/1 flags) */
#define COZK_SCALAR_U16 2 /* U16Scalars */
#define COZK_SCALAR_U32 3 /* U32Scalars */
#define COZK_SCALAR_U64 4 /* U64Scalars */
#define COZK_SCALAR_I64 5 /* I64Scalars */

/* BindingOrder (jolt-core poly::multilinear_polynomial::BindingOrder; used
 * co-jolt/src/poly/dense_mlpoly.rs:310-459) */
#define COZK_LOW_TO_HIGH 0
#define COZK_HIGH_TO_LOW 1

/* share mode of a polynomial handle */
#define COZK_MODE_PLAIN 1 /* one field element per entry (plain prover / public polynomial) */
#define COZK_MODE_REP3 2  /* Rep3PrimeFieldShare {a, b}: mpc-types/src/protocols/rep3/arithmetic/types.rs:22-29 */

typedef struct cozk_ctx cozk_ctx;
typedef struct cozk_bases cozk_bases;
typedef struct cozk_vec cozk_vec;

/* ---------------------------------------------------------------- context ----------------- */
/* replaces `icicle_init()` (co-jolt/examples/rep3_jolt.rs:195) + IoContext creation */
int cozk_ctx_create(int device, cozk_ctx** out);
int cozk_ctx_destroy(cozk_ctx* ctx);
const char* cozk_last_error(cozk_ctx* ctx);
/* cozk_layer_prove_rounds keeps one single-workgroup kernel resident for the tail of a layer's sumcheck; while it
 * waits for the host's challenge, kernels of other streams that the driver mapped to the same hardware queue
 * cannot start.  That is harmless for independent provers, but provers that need EACH OTHER's round messages to
 * make progress (several parties of one protocol run driven from one process on one GPU) would then wait for each
 * other.  libcozk cannot see such dependencies, so the DEFAULT is automatic and safe: the resident kernel is used
 * only while the context is the one live context on its device in this process (one party per process -- the
 * reference's deployment); as soon as a second context exists, rounds are one launch each.  enable > 0 forces it on
 * (the host vouches that its contexts are independent), 0 off, < 0 restores the automatic default.  If the kernel's
 * watchdog fires anyway (a round callback slower than COZK_RESIDENT_TIMEOUT_S, default 10 s -- e.g. a peer that a
 * transport with a longer timeout is still waiting for), the remaining rounds of that call fall back to
 * per-round launches; the proof is unaffected. */
int cozk_ctx_set_resident_rounds(cozk_ctx* ctx, int enable);
int cozk_ctx_synchronize(cozk_ctx* ctx);
/* raw hipStream_t of the context (so a host can order its own work / events against it) */
int cozk_ctx_stream(cozk_ctx* ctx, void** out_stream);
int cozk_device_count(int* out);

/* ---------------------------------------------------------------- device vectors ---------- */
/* upload a scalar vector (kind = COZK_SCALAR_*); `host` holds n elements of that kind */
int cozk_vec_upload(cozk_ctx* ctx, const void* host, size_t n, int kind, cozk_vec** out);
/* An FR vector whose canonical values fit `kind` (COZK_SCALAR_U32 / COZK_SCALAR_U64) as a vector of that kind -- what
 * VariableBaseMSM::msm_field_elements does before it dispatches on the scalars' bit length (jolt-core msm; call site
 * co-jolt/src/poly/commitment/pst13.rs:286-294): the MSM then sorts 3 / 5 windows of a 4- / 8-byte scalar instead of 16 of a
 * 32-byte one.  COZK_ERR_INVALID_ARG (and no vector) if any value does not fit. */
int cozk_vec_narrow(cozk_ctx* ctx, const cozk_vec* fr, int kind, cozk_vec** out);
int cozk_vec_alloc(cozk_ctx* ctx, size_t n, int kind, cozk_vec** out);
int cozk_vec_download(cozk_ctx* ctx, const cozk_vec* v, void* host);
int cozk_vec_free(cozk_vec* v);
size_t cozk_vec_len(const cozk_vec* v);
/* raw device pointer (for zero-copy interop with the host's own device buffers, e.g. RCCL staging) */
void* cozk_vec_device_ptr(const cozk_vec* v);
/* SYNTHETIC TEST DATA ONLY (benchmarks, fixtures): element i draws from SplitMix64(seed + i * 0xD1342543DE82EF95):
 * FR = canonical value rejection-sampled below r, stored in Montgomery form; small kinds = low bits.  max_bits > 0
 * masks the value to that many bits (e.g. 1 for 0/1 flags).  SplitMix64 is not a PRF: nothing secret is ever drawn
 * from it -- shares and masks come from the keyed ChaCha12 PRF below. */
int cozk_vec_fill_random(cozk_ctx* ctx, cozk_vec* v, uint64_t seed, int max_bits);

/* Keyed PRF of the engine: PRF(key, j) = element j of the ChaCha12 stream keyed with the 32-byte `key` (one block per
 * element: counter = j, rejection-sampled below r; csrc/prf.hip.hpp).  Keys are what the reference's parties exchange as
 * 32-byte ChaCha seeds (mpc-types/src/protocols/rep3.rs:29,177; mpc-core/src/protocols/rep3/network.rs:190-211) and
 * come from the host's CryptoRng; (key, counter range) pairs must never be reused for different data. */
#define COZK_PRF_KEY_BYTES 32
/* out[i] = PRF(key, counter + i), Montgomery form (FR vector) */
int cozk_vec_fill_prf(cozk_ctx* ctx, cozk_vec* v, const uint8_t key[COZK_PRF_KEY_BYTES], uint64_t counter);
/* Rep3 sharing of a secret vector on the device -- the witness scatter (rep3::share_field_element,
 * mpc-core/src/protocols/rep3/arithmetic.rs:21-33; jolt/vm/../witness.rs generate_poly_shares_rep3):
 * t0[i] = PRF(key0, counter + i), t1[i] = PRF(key1, counter + i), t2 = v - t0 - t1; returns `party`'s (a, b) =
 * (t0, t2) / (t1, t0) / (t2, t1).  The dealer calls it once per party. */
int cozk_rep3_share_vec(cozk_ctx* ctx, const cozk_vec* v, const uint8_t key0[COZK_PRF_KEY_BYTES],
                        const uint8_t key1[COZK_PRF_KEY_BYTES], uint64_t counter, int party,
                        cozk_vec** out_a, cozk_vec** out_b);
/* The witness scatter device to device (jolt/vm/jolt/coordinator.rs:72-91; receive_witness_share in jolt/vm/.../witness.rs):
 * as cozk_rep3_share_vec, but the secret lives on the DEALER's context and the outputs are vectors of `party_ctx` (another
 * GPU, or the same one): generated on the dealer's device, moved by a peer copy over xGMI when the devices differ.
 * Returns after the copy has completed.  The outputs are blocks of party_ctx's allocator: the call drains party_ctx's
 * stream before the dealer's stream writes them, and -- like every call that takes a context -- must come from the
 * thread that owns party_ctx (a context and its allocator are single-owner). */
int cozk_rep3_scatter(cozk_ctx* dealer, const cozk_vec* v, const uint8_t key0[COZK_PRF_KEY_BYTES],
                      const uint8_t key1[COZK_PRF_KEY_BYTES], uint64_t counter, cozk_ctx* party_ctx, int party,
                      cozk_vec** out_a, cozk_vec** out_b);
/* element-wise out[i] = a[i] (op) b[i] on 32-byte field elements: the local arithmetic of
 * mpc-types/src/protocols/additive/ops.rs (AdditivePrimeFieldShare is repr(transparent) over F).
 * base_field = 0: Fr (scalar field, what shares live in); 1: Fq (G1 coordinate field). */
#define COZK_OP_ADD 0
#define COZK_OP_SUB 1
#define COZK_OP_MUL 2
int cozk_vec_binop(cozk_ctx* ctx, int op, int base_field, const cozk_vec* a, const cozk_vec* b,
                   cozk_vec* out);
/* v[i] *= s (Fr) in place */
int cozk_vec_scale(cozk_ctx* ctx, cozk_vec* v, const uint64_t s[4]);

/* ---------------------------------------------------------------- MSM seam ---------------- */
/* Upload SRS points (`ck.powers_of_g[i]`, co-jolt/src/poly/commitment/pst13.rs:286-287,461-462) once;
 * replaces the ICICLE `gpu_bases: Option<&[GpuBaseType]>` argument (pst13.rs:52-59,288,320).
 * xy = n x 8 u64 (x[4], y[4]); infinity = n bytes or NULL.  precompute != 0 additionally builds the
 * window table 2^(16w) * G_i (16 x n x 64 B of HBM) that merges all Pippenger windows into one
 * bucket set. */
int cozk_bases_upload(cozk_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, size_t n,
                      int precompute, cozk_bases** out);
/* bases[i] = scalars[i] * g on device -- `MultilinearPC::setup` building `powers_of_g`
 * (ark-poly-commit, invoked by PST13::setup co-jolt/src/poly/commitment/pst13.rs:49-62).
 * g_xy = affine generator (8 u64). */
int cozk_bases_from_scalars(cozk_ctx* ctx, const cozk_vec* scalars_fr, const uint64_t* g_xy,
                            int precompute, cozk_bases** out);
int cozk_bases_download(cozk_ctx* ctx, const cozk_bases* b, size_t offset, size_t n, uint64_t* xy,
                        uint8_t* infinity);
int cozk_bases_free(cozk_bases* b);
size_t cozk_bases_len(const cozk_bases* b);
/* G'[b] = G[2b] + G[2b+1]: folds the duplicated scalars `q[k][x >> 1]` of PST13 `open`
 * (pst13.rs:459) into a half-size MSM */
int cozk_bases_pair_sums(cozk_ctx* ctx, const cozk_bases* b, int precompute, cozk_bases** out);

/* `VariableBaseMSM::msm_field_elements(bases[offset..offset+n], _, scalars, _, _)`
 * (call sites pst13.rs:286-294,461-469; co-noir-spartan/co-spartan/src/worker.rs:801-804).
 * Host-scalar form (PCIe inclusive) and device-resident form. */
int cozk_msm(cozk_ctx* ctx, const cozk_bases* bases, size_t offset, const void* host_scalars,
             int kind, size_t n, uint64_t out_xy[8], int* out_infinity);
int cozk_msm_vec(cozk_ctx* ctx, const cozk_bases* bases, size_t offset, const cozk_vec* scalars,
                 uint64_t out_xy[8], int* out_infinity);
/* `VariableBaseMSM::batch_msm(bases[..n], _, polys)` (pst13.rs:319-323): k MSMs over one base
 * slice; out_xy = k x 8 u64, out_infinity = k ints. */
int cozk_batch_msm_vec(cozk_ctx* ctx, const cozk_bases* bases, size_t offset,
                       const cozk_vec* const* scalars, size_t k, uint64_t* out_xy,
                       int* out_infinity);

/* k MSMs with a different base slice per polynomial: poly i runs over bases[offsets[i] .. +lens[i])
 * (lens = NULL: each vector's full length).  One launch set for all `nv` MSMs of PST13 `open`
 * (pst13.rs:445-471), whose levels live concatenated in one bases handle. */
int cozk_batch_msm_slices(cozk_ctx* ctx, const cozk_bases* bases, const size_t* offsets,
                          const cozk_vec* const* scalars, const size_t* lens, size_t k,
                          uint64_t* out_xy, int* out_infinity);

/* G1 helpers used by the coordinator-side combine (`combine_commitment_shares`, pst13.rs:72-108;
 * `coordinate_prove`, :110-122): out = sum of k affine points */
int cozk_g1_sum(cozk_ctx* ctx, const uint64_t* xy, const int* infinity, size_t k,
                uint64_t out_xy[8], int* out_infinity);
/* out = s * P (host helper for `combine_commitments`, pst13.rs:333-348, and trapdoor checks) */
int cozk_g1_mul(cozk_ctx* ctx, const uint64_t xy[8], int infinity, const uint64_t s[4],
                uint64_t out_xy[8], int* out_infinity);

/* ---------------------------------------------------------------- polynomial seam ---------- */
typedef struct cozk_poly cozk_poly;       /* Rep3DensePolynomial (co-jolt/src/poly/dense_mlpoly.rs:23-32); PLAIN = DensePolynomial */
typedef struct cozk_layer cozk_layer;     /* Rep3DenseInterleavedPolynomial (co-jolt/src/poly/dense_interleaved_poly.rs:35-48) */
typedef struct cozk_spliteq cozk_spliteq; /* SplitEqPolynomial (jolt-core; used dense_interleaved_poly.rs:218-303) */

/* Rep3DensePolynomial::from_vec_shares(a, b) (dense_mlpoly.rs:77-84); b = NULL for MODE_PLAIN. Copies. */
int cozk_poly_create(cozk_ctx* ctx, int mode, const cozk_vec* a, const cozk_vec* b, cozk_poly** out);
/* split_poly: zero-copy chunk over the high variables (dense_mlpoly.rs:275-301) */
int cozk_poly_chunk(cozk_ctx* ctx, const cozk_poly* src, size_t offset, size_t len, cozk_poly** out);
int cozk_poly_free(cozk_poly* p);
size_t cozk_poly_len(const cozk_poly* p);
int cozk_poly_mode(const cozk_poly* p);
/* current (bound) coefficients -> host; a, b = len x 4 u64 */
int cozk_poly_download(cozk_ctx* ctx, const cozk_poly* p, uint64_t* a, uint64_t* b);
/* copy_share_a (dense_mlpoly.rs:103-110) as a zero-copy view: component 0 = a, 1 = b */
int cozk_poly_share_view(cozk_ctx* ctx, const cozk_poly* p, int component, cozk_vec** out);
/* PolynomialBinding::bind / bind_parallel (dense_mlpoly.rs:310-459) */
int cozk_poly_bind(cozk_ctx* ctx, cozk_poly* p, const uint64_t r[4], int order);
/* get_bound_coeff / final_sumcheck_claim (dense_mlpoly.rs:251-257,461-465) */
int cozk_poly_get_coeff(cozk_ctx* ctx, const cozk_poly* p, size_t index, uint64_t a[4], uint64_t b[4]);
/* EqPolynomial::evals(r) on device, big-endian (use: dense_mlpoly.rs:149-153,183-185) */
int cozk_eq_evals(cozk_ctx* ctx, const uint64_t* r, int nv, cozk_vec** out);
/* batch_evaluate / evaluate_at_chi (dense_mlpoly.rs:160-192): out[k] = additive share of poly_k . chi */
int cozk_poly_batch_evaluate_at_chi(cozk_ctx* ctx, const cozk_poly* const* polys, size_t k,
                                    const cozk_vec* chi, uint64_t* out);
/* dot_product_with_public (dense_mlpoly.rs:228-234) -> share (a, b) */
int cozk_poly_dot_product_with_public(cozk_ctx* ctx, const cozk_poly* p, const cozk_vec* pub,
                                      uint64_t a[4], uint64_t b[4]);
/* linear_combination (dense_mlpoly.rs:195-226; multilinear_polynomial.rs:196-296); PLAIN inputs in a
 * REP3 combination are public polynomials added via add_public for `party_id` */
int cozk_poly_linear_combination(cozk_ctx* ctx, const cozk_poly* const* polys, const uint64_t* coeffs,
                                 size_t k, int out_mode, int party_id, cozk_poly** out);
/* compute_leaves of the memory-checking instances (K11; co-jolt/src/jolt/vm/bytecode/worker.rs:57-100,
 * read_write_memory/worker.rs:207-300): leaf[i] = sum_k col_coeffs[k] * cols[k][i] (compact public columns: U8 /
 * U16 / U32 / U64 vectors, CompactPolynomial::field_mul) + sum_j poly_coeffs[j] * polys[j][i] (shared or public
 * Fr polynomials, mul_public) + constant (e.g. -tau).  In MODE_REP3 the public part enters through add_public
 * (party 0's a, party 1's b).  Writes n leaves at out_a / out_b[offset ..] (out_b NULL for MODE_PLAIN), so the leaves
 * of a batch land in one buffer that cozk_layer_create adopts. */
int cozk_fingerprint_leaves(cozk_ctx* ctx, const cozk_vec* const* cols, const uint64_t* col_coeffs, size_t n_cols,
                            const cozk_poly* const* polys, const uint64_t* poly_coeffs, size_t n_polys,
                            const uint64_t constant[4], int mode, int party_id, cozk_vec* out_a, cozk_vec* out_b,
                            size_t offset, size_t n);
/* compute_quadratic inner sums for the live openings of one round (opening_proof.rs:374-414):
 * out[2i] = eval_0, out[2i+1] = eval_2 (additive) */
int cozk_open_quadratic_evals(cozk_ctx* ctx, const cozk_poly* const* polys,
                              const cozk_poly* const* eqs, size_t k, uint64_t* out);
/* one round of prove_arbitrary_worker's evaluation loop (co-jolt/src/subprotocols/sumcheck.rs:189-215) for
 * the comb_funcs the reference uses: a product of m <= 4 polynomials, at most one of them REP3 (Spartan
 * inner / shift sumchecks r1cs/spartan/worker.rs:162-235; output check read_write_memory/worker.rs:149-164).
 * out[e] = additive evaluation at x = 0, 2, .., degree; HighToLow (sumcheck_evals, dense_mlpoly.rs:113-147) */
int cozk_prod_sumcheck_evals(cozk_ctx* ctx, const cozk_poly* const* polys, size_t m, int degree,
                             uint64_t* out);
/* co-noir-spartan: Rep3Sumcheck::first_sumcheck_prove_round evaluations at X = 0..3 of
 * sum_b eq * (za x zb) - into_additive(zc * eq), before the additive zero-mask
 * (co-noir-spartan/co-spartan/src/sumcheck.rs:171-280); fix_variables = cozk_poly_bind(.., LOW_TO_HIGH)
 * on the SoA components (mpc-core/src/protocols/rep3/poly.rs:56-62) */
int cozk_spartan_first_round(cozk_ctx* ctx, const cozk_poly* za, const cozk_poly* zb,
                             const cozk_poly* zc, const cozk_poly* pub, uint64_t out[16]);
/* Rep3Sumcheck::second_sumcheck_prove_round: Rep3 evaluations at X = 0..2 of
 * sum_b z * (alpha A + beta B + gamma C), before the Rep3 mask (sumcheck.rs:282-395) */
int cozk_spartan_second_round(cozk_ctx* ctx, const cozk_poly* z, const cozk_poly* a, const cozk_poly* b,
                              const cozk_poly* c, const uint64_t coef[12], uint64_t out_a[12],
                              uint64_t out_b[12]);
/* SpartanProverWorker::zero_round (co-spartan/src/worker.rs:153-182): (za, zb, zc) = (A, B, C) z on shares;
 * CSR rows: row_ptr (U32, nrows+1), col (U32, nnz), val_a/b/c (FR, nnz) */
int cozk_sparse_matvec3(cozk_ctx* ctx, const cozk_vec* row_ptr, const cozk_vec* col,
                        const cozk_vec* val_a, const cozk_vec* val_b, const cozk_vec* val_c,
                        const cozk_poly* z, cozk_poly** out_za, cozk_poly** out_zb, cozk_poly** out_zc);
/* one fold of PST13 `open` (pst13.rs:445-459): q[b] = r[2b+1]-r[2b]; r'[b] = r[2b](1-p) + r[2b+1]p */
int cozk_pst_fold(cozk_ctx* ctx, const cozk_vec* r, const uint64_t p[4], cozk_vec* q, cozk_vec* r_next);

/* Rep3DenseInterleavedPolynomial::new (dense_interleaved_poly.rs:61-75); take_ownership != 0 adopts
 * the vectors' device buffers instead of copying (the vectors become empty views) */
int cozk_layer_create(cozk_ctx* ctx, int mode, const cozk_vec* a, const cozk_vec* b,
                      int take_ownership, cozk_layer** out);
int cozk_layer_free(cozk_layer* l);
size_t cozk_layer_len(const cozk_layer* l);
int cozk_layer_download(cozk_ctx* ctx, const cozk_layer* l, uint64_t* a, uint64_t* b);
int cozk_layer_clone(cozk_ctx* ctx, const cozk_layer* src, cozk_layer** out);
/* Rep3Bindable::bind (dense_interleaved_poly.rs:155-195) */
int cozk_layer_bind(cozk_ctx* ctx, cozk_layer* l, const uint64_t r[4]);
/* Rep3BatchedCubicSumcheckWorker::compute_cubic (dense_interleaved_poly.rs:210-365): 4 additive
 * coefficient shares (low -> high) of the round polynomial */
int cozk_layer_compute_cubic(cozk_ctx* ctx, const cozk_layer* l, const cozk_spliteq* eq,
                             const uint64_t prev_claim[4], uint64_t out_coeffs[16]);
/* one round of Rep3BatchedCubicSumcheckWorker::prove_sumcheck (co-jolt/src/subprotocols/sumcheck.rs:107-122) in
 * one call: Rep3Bindable::bind + SplitEqPolynomial::bind with the previous round's challenge r (NULL in the first
 * round), then compute_cubic; small layers run as a single launch */
int cozk_layer_round(cozk_ctx* ctx, cozk_layer* l, cozk_spliteq* eq, const uint64_t* r,
                     const uint64_t prev_claim[4], uint64_t out_coeffs[16]);
/* the whole round loop of prove_sumcheck (sumcheck.rs:96-131) for one layer, the host's transport as a callback:
 * per round cb(user, round, coeffs[4x4], r_out[4], next_claim_out[4]) sends the round polynomial's coefficient
 * shares and returns the challenge and this party's additive share of the next claim (0 = ok).  Large rounds
 * are one launch each; the tail of the layer (<= 2048 elements) runs in one resident kernel that trades sums and
 * challenges with the host through pinned memory.  out_r = num_rounds x 4; final_claims = L.a, L.b, R.a, R.b */
typedef int (*cozk_round_cb)(void* user, int round, const uint64_t coeffs[16], uint64_t r_out[4],
                             uint64_t next_claim_out[4]);
int cozk_layer_prove_rounds(cozk_ctx* ctx, cozk_layer* l, cozk_spliteq* eq, const uint64_t claim[4],
                            int num_rounds, cozk_round_cb cb, void* user, uint64_t* out_r,
                            uint64_t final_claims[16]);
/* the same loop for a worker sub-net (jolt/vm/instruction_lookups/worker.rs:593-597): the callback gets the raw sums
 * g(0), g(2), g(3) of its chunk -- the coordinator inserts g(1) = claim - g(0) -- and returns the challenge */
typedef int (*cozk_round_evals_cb)(void* user, int round, const uint64_t evals[12], uint64_t r_out[4]);
int cozk_layer_prove_rounds_evals(cozk_ctx* ctx, cozk_layer* l, cozk_spliteq* eq, int num_rounds,
                                  cozk_round_evals_cb cb, void* user, uint64_t* out_r,
                                  uint64_t final_claims[16]);
/* raw sums g(0), g(2), g(3) of compute_cubic (12 u64) for worker sub-nets: the coordinator inserts
 * claim - g(0) itself, as for the reference's primary sumcheck (instruction_lookups/worker.rs:593-597) */
int cozk_layer_compute_cubic_evals(cozk_ctx* ctx, const cozk_layer* l, const cozk_spliteq* eq,
                                   uint64_t out_evals[12]);
/* final_claims (dense_interleaved_poly.rs:367-372): out = L.a, L.b, R.a, R.b (b = 0 for PLAIN) */
int cozk_layer_final_claims(cozk_ctx* ctx, const cozk_layer* l, uint64_t out[16]);
/* local half of layer_output -> mul_vec (dense_interleaved_poly.rs:122-141; local product
 * mpc-types/src/protocols/rep3/arithmetic/ops.rs:71-78): out[j] = L[j] x R[j] + mask_j, where
 * mask_j = PRF(key_self, counter + j) - PRF(key_prev, counter + j) when masked != 0 (key_self is shared with the
 * next party, key_prev with the previous one: the three masks sum to zero; keys may be NULL when masked == 0) */
int cozk_layer_output_local(cozk_ctx* ctx, const cozk_layer* l, int masked, const uint8_t* key_self,
                            const uint8_t* key_prev, uint64_t counter, cozk_vec** out);
/* rep3::arithmetic::mul_vec, local half, on SoA share vectors */
int cozk_rep3_mul_vec_local(cozk_ctx* ctx, int mode, const cozk_vec* xa, const cozk_vec* xb,
                            const cozk_vec* ya, const cozk_vec* yb, int masked, const uint8_t* key_self,
                            const uint8_t* key_prev, uint64_t counter, cozk_vec** out);
/* claimed_outputs (grand_product.rs:266-272): out = (len/2) x 4 u64 additive products */
int cozk_layer_claimed_outputs(cozk_ctx* ctx, const cozk_layer* l, uint64_t* out);

/* ---- toggled / sparse batched grand product (co-jolt/src/subprotocols/sparse_grand_product.rs; Lasso's read / write
 * memory checking of the instruction lookups, jolt/vm/instruction_lookups/worker.rs:763-859).
 * cozk_toggle = Rep3BatchedGrandProductToggleLayer (sparse_grand_product.rs:30-70): public 0/1 flags (one U8 column of N
 * entries per PAIR of circuits -- `flag_indices[batch_index / 2]`, :84) and shared fingerprints (2 * n_pairs circuits x N,
 * circuit-major).  The Rep3SparseInterleavedPolynomial layers above it (co-jolt/src/poly/sparse_interleaved_poly.rs) are
 * kept DENSE on the device -- a missing entry is a stored share of one -- i.e. they are ordinary cozk_layer objects:
 * cozk_toggle_layer_output gives the first of them, cozk_layer_output_local + the ring reshare the rest, and
 * cozk_layer_prove_rounds proves them; every party's round messages equal the reference's sparse computation. */
typedef struct cozk_toggle cozk_toggle;
/* Rep3BatchedGrandProductToggleLayer::new (:57-70); take_ownership != 0 adopts the fingerprint buffers */
int cozk_toggle_create(cozk_ctx* ctx, int mode, const cozk_vec* const* flags, size_t n_pairs, cozk_vec* fp_a,
                       cozk_vec* fp_b, int take_ownership, cozk_toggle** out);
int cozk_toggle_free(cozk_toggle* t);
size_t cozk_toggle_batch(const cozk_toggle* t); /* circuits = 2 * n_pairs */
size_t cozk_toggle_len(const cozk_toggle* t);   /* fingerprints per circuit */
/* layer_output (:76-97): entry b * N + i = flag ? fingerprint : promote_to_trivial_share(party_id, one) */
int cozk_toggle_layer_output(cozk_ctx* ctx, const cozk_toggle* t, int party_id, cozk_layer** out);
/* Rep3Bindable::bind (:153-290), incl. the coalesce step (:104-134) when one entry per circuit is left */
int cozk_toggle_bind(cozk_ctx* ctx, cozk_toggle* t, const uint64_t r[4]);
/* one round of prove_sumcheck over the toggle layer (compute_cubic, :311-823): bind layer + split-eq tables with the
 * previous challenge r (NULL in the first round), then this party's additive g(0), g(2), g(3) of
 * sum_x eq(x) (flag(x) fingerprint(x) + 1 - flag(x)) */
int cozk_toggle_round(cozk_ctx* ctx, cozk_toggle* t, cozk_spliteq* eq, const uint64_t* r, int party_id,
                      uint64_t out_evals[12]);
/* final_claims (:825-835): the bound flag (public) and the bound fingerprint share */
int cozk_toggle_final_claims(cozk_ctx* ctx, const cozk_toggle* t, uint64_t flag[4], uint64_t fp_a[4], uint64_t fp_b[4]);
/* current (bound) flags and fingerprints -> host; any output pointer may be NULL */
int cozk_toggle_download(cozk_ctx* ctx, const cozk_toggle* t, uint64_t* flags, uint64_t* fp_a, uint64_t* fp_b,
                         size_t* n_flags, size_t* n_fp);

/* ---- Lasso's primary sumcheck of the instruction lookups (co-jolt/src/jolt/vm/instruction_lookups/worker.rs:180-720):
 *   sum_x eq(r, x) ( sum_i flag_i(x) g_i(E_1(x), .., E_alpha(x)) - lookup_output(x) ) = 0.
 * cozk_primary holds eq (public), the instruction flags (public 0/1 U8 columns), the E polynomials and lookup_outputs
 * (shared) and binds them LowToHigh once per round.  The collations g_i (combine_lookups_rep3_batched,
 * co-jolt/src/jolt/instruction/, one .rs each) are given as a table of forms over memory indices:
 *   COZK_G_CONCAT  (and.rs:89-101, utils/instruction_utils.rs:26-47): sum_j 2^(bits (n-1-j)) E_mems[j]        -- local
 *                  ADD SUB AND OR XOR SLL MUL MULU MULHU VIRTUAL_ADVICE VIRTUAL_MOVE; bits = 0: the plain sum of SRA / SRL
 *                  (sra.rs:122-132); the same memory listed twice: MOVSIGN (virtual_movsign.rs:126-139)
 *   COZK_G_PRODUCT (beq.rs:106-130 -> product_many, mpc-core rep3/arithmetic.rs:86-102): prod_j E_mems[j]
 *   COZK_G_LTU     (sltu.rs:139-170): mems = C LTU memories then C - 1 EQ memories: sum_i ltu_i prod_{j<i} eq_j
 *   COZK_G_NOT_PRODUCT (bne.rs:108-140), COZK_G_NOT_LTU (bgeu.rs:113-132), COZK_G_NOT_SLT (bge.rs:121-140): 1 - the form
 *   COZK_G_SLT     (slt.rs:184-302): mems = left_msb, right_msb, C - 1 LTU, C - 2 EQ, lt_abs, eq_abs (2C + 1):
 *                  l (1 - r) + (l r + (1 - l)(1 - r)) (lt_abs + sum_i ltu_i eq_abs prod_{j<i} eq_j)
 *   COZK_G_LTE     (virtual_assert_lte.rs:144-209): mems = C LTU then C EQ: sum_i ltu_i prod_{j<i} eq_j + prod_j eq_j
 *   COZK_G_NOT_FIRST (virtual_assert_halfword_alignment.rs:111-125): 1 - E_mems[0]                               -- local
 *   COZK_G_DIV0    (virtual_assert_valid_div0.rs:36-42, the plain formula): mems = C left_is_zero then C div_by_zero:
 *                  1 - prod left_is_zero + prod div_by_zero   (the Rep3 body at :159-225 computes 1 - (.. + ..): a sign
 *                  slip in the reference that its own plain verifier would reject; not reproduced)
 *   COZK_G_UNSIGNED_REM (virtual_assert_valid_unsigned_remainder.rs:154-249): mems = C LTU, C - 1 EQ, C right_is_zero:
 *                  LTU form + prod right_is_zero
 *   COZK_G_SIGNED_REM (virtual_assert_valid_signed_remainder.rs:40-67; its Rep3 body, :265-273, is todo!() in the reference, the
 *                  multiplication schedule here is ours): mems = left_msb, right_msb, C - 1 EQ, C - 1 LTU, eq_abs, lt_abs,
 *                  C left_is_zero, C right_is_zero (4C + 2)
 *   COZK_G_ZERO    (virtual_pow2.rs:70-78, virtual_right_shift_padding.rs:74-82): 0                                   -- local
 * The multiplicative forms multiply shared values: each level is ONE batched mul_vec / reshare_additive_many over all
 * active (index, instruction, point) items -- cozk_primary_level does the local half and names the device buffers of the
 * ring exchange, which the host runs (cozk_reshare / its own transport) before the next call.  The last multiplication of
 * every form stays additive (into_additive follows it in worker.rs:553): same totals, one ring round less. */
#define COZK_G_CONCAT 0
#define COZK_G_PRODUCT 1
#define COZK_G_LTU 2
#define COZK_G_NOT_PRODUCT 3
#define COZK_G_NOT_LTU 4
#define COZK_G_SLT 5
#define COZK_G_NOT_SLT 6
#define COZK_G_LTE 7
#define COZK_G_NOT_FIRST 8
#define COZK_G_DIV0 9
#define COZK_G_UNSIGNED_REM 10
#define COZK_G_SIGNED_REM 11
#define COZK_G_ZERO 12
#define COZK_PRIMARY_MAX_MEMS 20
typedef struct cozk_primary cozk_primary;
typedef struct cozk_primary_instr {
    int form;                         /* COZK_G_* */
    int n_mems;                       /* 1..COZK_PRIMARY_MAX_MEMS (the chunk count C follows from form and n_mems) */
    int mems[COZK_PRIMARY_MAX_MEMS];  /* indices into the E polynomials, in the order the form lists them */
    int bits;                         /* CONCAT: operand bits per chunk */
} cozk_primary_instr;
/* flags: U8 0/1 columns, or FR vectors (flags that are already bound: the remaining rounds after a worker sub-net split) */
int cozk_primary_create(cozk_ctx* ctx, int mode, int party_id, const cozk_primary_instr* instrs, size_t n_instr,
                        const cozk_vec* const* flags, const cozk_poly* const* E, size_t n_mem,
                        const cozk_poly* lookup_outputs, const cozk_vec* eq, cozk_primary** out);
int cozk_primary_free(cozk_primary* p);
int cozk_primary_degree(const cozk_primary* p); /* sumcheck_poly_degree (worker.rs:701-708): max g degree + 2 */
size_t cozk_primary_len(const cozk_primary* p);
/* one round = primary_sumcheck_prover_message (worker.rs:454-598), in three steps:
 * round_begin: bind with the previous challenge r (NULL in the first round), the pass over the linear instructions, the
 *   item list of the multiplicative ones (*n_items; *n_levels mul_vec levels follow, 0 if there are no items);
 * level (1 .. n_levels): local products + zero-sharing masks PRF(key_self, counter + j) - PRF(key_prev, counter + j) of
 *   n_elems elements (every item's reshared values of this level x degree; 0: nothing to exchange at this level); Rep3: exchange *send -> next party, previous party's -> *recv, then go on;
 * round_finish: out_evals = degree x 4 u64, this party's additive evaluations at X = 0, 2, 3, .., degree. */
int cozk_primary_round_begin(cozk_ctx* ctx, cozk_primary* p, const uint64_t* r, size_t* n_items, int* n_levels);
int cozk_primary_level(cozk_ctx* ctx, cozk_primary* p, int level, const uint8_t* key_self, const uint8_t* key_prev,
                       uint64_t counter, const void** send, void** recv, size_t* n_elems);
int cozk_primary_round_finish(cozk_ctx* ctx, cozk_primary* p, uint64_t* out_evals);
/* after the last round: bind with the last challenge; E_evals = n_mem x (a[4], b[4]), flag_evals = n_instr x 4 (public),
 * out_eval = (a[4], b[4]), eq_eval[4] (may be NULL)  (worker.rs:427-452) */
int cozk_primary_final_evals(cozk_ctx* ctx, cozk_primary* p, const uint64_t r[4], uint64_t* E_evals,
                             uint64_t* flag_evals, uint64_t out_eval[8], uint64_t eq_eval[4]);

/* ---- co-jolt's Spartan outer sumcheck over Az / Bz / Cz (co-jolt/src/poly/spartan_interleaved_poly.rs,
 * co-jolt/src/r1cs/spartan/worker.rs:63-120,277-300):  sum_x eq(tau, x) (Az(x) Bz(x) - Cz(x)) = 0.
 * The constraint system is an input (the concrete Jolt constraints live in jolt-core): linear combinations over the
 * witness columns (`flattened_polynomials`, one entry per step each), as jolt-core's r1cs builder holds them:
 *   uniform constraint i (Constraint {a, b, c}):           row i of a step:  Az = a.z, Bz = b.z, Cz = c.z
 *   cross-step constraint j (OffsetEqConstraint {cond, a, b}): Az = a.z - b.z, Bz = cond.z, Cz = 0, where an LC with
 *   offset != 0 reads the NEXT step (its constant term only at the last step; spartan_interleaved_poly.rs:666-684).
 * Az, Bz, Cz live as dense share arrays over rows = step * padded_num_constraints + constraint (a public column enters as
 * its trivial share), built on the device straight from the columns. */
typedef struct cozk_lc {
    int first_term; /* terms [first_term, first_term + n_terms) of the system's term arrays */
    int n_terms;
    int offset;
} cozk_lc;
typedef struct cozk_r1cs {
    const int* term_var;       /* column index, or -1 for the constant */
    const int64_t* term_coeff; /* small integer coefficients (F::from_i64) */
    size_t n_terms;
    const cozk_lc* uniform;    /* 3 per constraint: a, b, c */
    size_t n_uniform;
    const cozk_lc* cross;      /* 3 per constraint: a, b, cond */
    size_t n_cross;
    size_t padded_num_constraints; /* rows per step: a power of two >= n_uniform + n_cross */
} cozk_r1cs;
typedef struct cozk_outer cozk_outer;
/* compute_spartan_Az_Bz_Cz + GruenSplitEqPolynomial::new(tau); vars: REP3 polynomials are shared columns, PLAIN ones
 * public; tau = log2(steps * padded_num_constraints) challenges */
int cozk_outer_create(cozk_ctx* ctx, int mode, int party_id, const cozk_r1cs* sys, const cozk_poly* const* vars,
                      size_t n_vars, const uint64_t* tau, size_t n_tau, cozk_outer** out);
int cozk_outer_free(cozk_outer* st);
size_t cozk_outer_len(const cozk_outer* st);
int cozk_outer_download(cozk_ctx* ctx, const cozk_outer* st, uint64_t* az_a, uint64_t* az_b, uint64_t* bz_a,
                        uint64_t* bz_b, uint64_t* cz_a, uint64_t* cz_b);
/* first_sumcheck_round / subsequent_sumcheck_round (spartan_interleaved_poly.rs:189-612) without the network leg: bind with
 * the previous challenge r (NULL in the first round), then the cubic round polynomial of
 * process_eq_sumcheck_round_worker (subprotocols/sumcheck_spartan.rs:44-79) as 4 additive coefficient shares */
int cozk_outer_round(cozk_ctx* ctx, cozk_outer* st, const uint64_t* r, const uint64_t claim[4], uint64_t out_coeffs[16]);
/* final_sumcheck_evals (:648-664) after binding with the last challenge: additive Az(r), Bz(r), Cz(r) */
int cozk_outer_final_evals(cozk_ctx* ctx, cozk_outer* st, const uint64_t r[4], uint64_t out[12]);
/* ---- Spartan inner / shift sumchecks (co-jolt/src/r1cs/spartan/worker.rs:100-275).
 * EqPlusOnePolynomial::evals(r, None).1 (jolt-core, used worker.rs:116): out[y] = eq_plus_one(r, y), big-endian, 2^nv entries;
 * eq_plus_one(x, y) = 1 iff y = x + 1 and x < 2^nv - 1 (no wrap-around).  The eq half of the pair is cozk_eq_evals. */
int cozk_eq_plus_one_evals(cozk_ctx* ctx, const uint64_t* r, int nv, cozk_vec** out);
/* bind_z / bind_shift_z (worker.rs:139-152): dot_product_with_public (dense_mlpoly.rs:228-234) of k polynomials of one length
 * with n_pub = 1 or 2 public vectors in ONE pass over the polynomials.  out[(p * n_pub + q) * 8 ..] = share (a[4], b[4]);
 * b = 0 for a PLAIN polynomial (the dot product of a public polynomial is a public value) */
int cozk_poly_batch_dot_public(cozk_ctx* ctx, const cozk_poly* const* polys, size_t k, const cozk_vec* const* pubs,
                               size_t n_pub, uint64_t* out);

/* ---- co-noir-spartan's public lookup round (co-noir-spartan/co-spartan/src/worker.rs:400-575,694-724,836-846;
 * co-noir-spartan/spartan/src/logup.rs:31-80; co-spartan/src/sumcheck.rs:434-500): plain Fr data, no shares. */
/* hash_tuple (worker.rs:836-846): out[j] = idx[j] + v_msg * eq[idx[j]] for the (pre-filtered) indices, the tail up to n_out
 * (a power of two) repeats entry 0 */
int cozk_hash_tuple(cozk_ctx* ctx, const cozk_vec* idx_u32, const cozk_vec* eq, const uint64_t v_msg[4], size_t n_out,
                    cozk_vec** out);
/* eq_tilde_{rx,ry}(_chunk) of third_round (worker.rs:296-343,376-391): out[j] = src[idx[j]] (0xffffffff = usize::MAX and the
 * padding up to n_out: 0) */
int cozk_vec_gather(cozk_ctx* ctx, const cozk_vec* idx_u32, const cozk_vec* src, size_t n_out, cozk_vec** out);
/* LogLookupProof::prove's field work (logup.rs:45-70): phi = x + values, h = m / phi (m = NULL: 1 / phi) */
int cozk_logup_h(cozk_ctx* ctx, const cozk_vec* values, const cozk_vec* m, const uint64_t x[4], cozk_vec** out_phi,
                 cozk_vec** out_h);
/* boost_degree (spartan/src/utils.rs:11-27): scale by 2^-(new_num_vars - num_vars) and repeat up to 2^new_num_vars */
int cozk_vec_boost_degree(cozk_ctx* ctx, const cozk_vec* v, int new_num_vars, cozk_vec** out);
/* distributed_sumcheck_worker's prover (worker.rs:694-724) = ark-linear-sumcheck IPForMLSumcheck::{prover_init,
 * prove_round} over a ListOfProductsOfPolynomials: product q = coefs[q] * prod of counts[q] polynomials
 * (factor_idx lists them product after product; <= 48 polynomials, <= 32 products, <= 4 factors) */
typedef struct cozk_prodlist cozk_prodlist;
int cozk_prodlist_create(cozk_ctx* ctx, const cozk_vec* const* polys, size_t n_polys, const uint64_t* coefs,
                         const int* counts, const int* factor_idx, size_t n_terms, cozk_prodlist** out);
int cozk_prodlist_free(cozk_prodlist* pl);
int cozk_prodlist_degree(const cozk_prodlist* pl); /* max_multiplicands */
/* prove_round: fix_variables with the previous randomness r (NULL in the first round), then out_evals =
 * (degree + 1) x 4 u64: the evaluations at t = 0 .. degree */
int cozk_prodlist_round(cozk_ctx* ctx, cozk_prodlist* pl, const uint64_t* r, uint64_t* out_evals);
/* the last fix_variables + obtain_distrbuted_sumcheck_prover_state (sumcheck.rs:434-452): n_polys x 4 u64 */
int cozk_prodlist_final(cozk_ctx* ctx, cozk_prodlist* pl, const uint64_t r[4], uint64_t* out_vals);

/* SplitEqPolynomial::{new, bind} */
int cozk_spliteq_new(cozk_ctx* ctx, const uint64_t* w, int nv, cozk_spliteq** out);
int cozk_spliteq_free(cozk_spliteq* e);
int cozk_spliteq_lens(const cozk_spliteq* e, size_t* e1_len, size_t* e2_len);
int cozk_spliteq_bind(cozk_ctx* ctx, cozk_spliteq* e, const uint64_t r[4]);

/* the layer's current coefficients as a borrowed dense polynomial (for evaluating a layer's MLE) */
int cozk_layer_as_poly(cozk_ctx* ctx, const cozk_layer* l, cozk_poly** out);

/* ---------------------------------------------------------------- network seam ------------- */
/* Host-supplied transports for the worker drivers (libcozk's C++ host layer, csrc/host/prover.hpp).
 * Star = MpcStarNetWorker::{send_response, receive_request} (mpc-net/src/mpc_star.rs:5-66); payloads
 * are ark-serialize uncompressed bytes.  Ring = Rep3Network reshare (send to next, receive from prev;
 * mpc-core/src/protocols/rep3/arithmetic.rs:144-164) on DEVICE buffers, e.g. an RCCL
 * ncclSend/ncclRecv pair.  Callbacks return 0 on success. */
typedef struct cozk_star_net {
    void* user;
    int (*send_response)(void* user, const void* bytes, size_t len);
    int (*receive_request)(void* user, void* buf, size_t cap, size_t* out_len);
} cozk_star_net;
typedef struct cozk_ring_net {
    void* user;
    int (*reshare)(void* user, const void* dev_send, void* dev_recv, size_t nbytes);
    int stream_ordered; /* 0: libcozk synchronises the context's stream before the call and the callback returns when
                           dev_recv is complete; 1: the callback only ENQUEUES the exchange on the context's stream
                           (cozk_ctx_stream) -- no host synchronisation on either side (cozk_ring_net_native) */
} cozk_ring_net;

/* Native ring: the same exchange carried by RCCL inside libcozk -- one ncclSend (to the next party) / ncclRecv (from the
 * previous one) pair per call on the context's stream, GPU to GPU over xGMI, asynchronous to the host.  One party per
 * GPU / process.  Set-up mirrors ncclCommInitRank: ONE participant draws an id (cozk_ring_unique_id), the host
 * distributes those 128 bytes through its own channel (the mpc-net connection it already has), every participant calls
 * cozk_ring_init(ctx, id, rank, nranks) -- rank = PartyID, nranks = 3 for Rep3; the call blocks until all have joined.
 * librccl is loaded on first use.  Replaces Rep3Network::{reshare, send_next, recv_prev} as used by
 * mpc-core/src/protocols/rep3/arithmetic.rs:144-164. */
#define COZK_RING_ID_BYTES 128
int cozk_ring_unique_id(uint8_t out[COZK_RING_ID_BYTES]);
int cozk_ring_init(cozk_ctx* ctx, const uint8_t id[COZK_RING_ID_BYTES], int rank, int nranks);
int cozk_ring_destroy(cozk_ctx* ctx);
int cozk_ring_info(cozk_ctx* ctx, int* rank, int* nranks, uint64_t* bytes_sent);
/* reshare_additive_many (arithmetic.rs:152-164): send `send` to the next party, receive `recv` from the previous one */
int cozk_reshare(cozk_ctx* ctx, const cozk_vec* send, cozk_vec* recv);
/* rep3::arithmetic::mul_vec, whole: out_a = x (x) y + PRF(key_self, counter + j) - PRF(key_prev, counter + j) (local
 * product, mpc-types/src/protocols/rep3/arithmetic/ops.rs:71-78, + zero-sharing mask), then the ring exchange:
 * out_b = the previous party's out_a.  x, y as SoA component vectors; both outputs are new vectors. */
int cozk_rep3_mul_vec(cozk_ctx* ctx, const cozk_vec* xa, const cozk_vec* xb, const cozk_vec* ya, const cozk_vec* yb,
                      const uint8_t* key_self, const uint8_t* key_prev, uint64_t counter, cozk_vec** out_a,
                      cozk_vec** out_b);
/* a cozk_ring_net backed by the context's native ring, for the worker drivers and cozk_harness_prove_distributed */
int cozk_ring_net_native(cozk_ctx* ctx, cozk_ring_net* out);

/* ---------------------------------------------------------------- wire format -------------- */
/* ark-serialize *uncompressed* G1Affine, the encoding of every point the workers send and of the proof structs
 * PST13Commitment{nv, g_product} / Proof{proofs} (mpc-net/src/rep3/quic/worker.rs:187-219; co-jolt/src/poly/commitment/
 * pst13.rs:397-401): x || y as 32-byte LE canonical integers, SWFlags in the top bits of the last byte (bit 6 = infinity,
 * bit 7 = y > -y).  Host-only (no device needed).  decode validates like arkworks' Validate::Yes -- canonical
 * coordinates, consistent flags, on the curve -- and returns COZK_ERR_INVALID_ARG for bytes arkworks would reject. */
int cozk_wire_g1_encode(const uint64_t xy[8], int infinity, uint8_t out[64]);
int cozk_wire_g1_decode(const uint8_t in[64], uint64_t xy[8], int* infinity);

/* ---------------------------------------------------------------- worker drivers ----------- */
/* The C++ round loops (csrc/host/prover.hpp) run against host-supplied nets: the calls a Rust host makes
 * when it wants the whole loop rather than the per-round kernels. */
typedef struct cozk_worker_params {
    int mode;               /* COZK_MODE_PLAIN / COZK_MODE_REP3 */
    int party;              /* PartyID 0..2 */
    uint8_t key_self[COZK_PRF_KEY_BYTES]; /* zero-sharing PRF key shared with the next party */
    uint8_t key_prev[COZK_PRF_KEY_BYTES]; /* ... with the previous party */
    uint64_t mask_counter;                /* starting counter of the zero-sharing stream */
} cozk_worker_params;
/* construct + prove_grand_product_worker (co-jolt/src/subprotocols/grand_product.rs:111-130,239-255) on a
 * clone of `leaves`; ring may be NULL for the plain prover.  out_r = final point (r_cap x 4 u64). */
int cozk_worker_prove_grand_product(cozk_ctx* ctx, const cozk_worker_params* wp, const cozk_star_net* star,
                                    const cozk_ring_net* ring, cozk_layer* leaves, size_t batch_size,
                                    uint64_t* out_r, size_t r_cap, size_t* out_r_len);
/* prove_arbitrary_worker (co-jolt/src/subprotocols/sumcheck.rs:168-246), comb_func = product of the m
 * polynomials; binds them HighToLow in place; out_r = num_rounds x 4, out_final_evals = m x 4 (additive) */
int cozk_worker_prove_arbitrary(cozk_ctx* ctx, const cozk_worker_params* wp, const cozk_star_net* star,
                                cozk_poly* const* polys, size_t m, int combined_degree,
                                const uint64_t claim[4], int num_rounds, uint64_t* out_r,
                                uint64_t* out_final_evals);
/* rep3_first_sumcheck_worker / second (co-noir-spartan/co-spartan/src/worker.rs:593-639, sumcheck.rs:171-395) */
int cozk_worker_spartan_first_sumcheck(cozk_ctx* ctx, const cozk_worker_params* wp, const cozk_star_net* star,
                                       cozk_poly* za, cozk_poly* zb, cozk_poly* zc, cozk_poly* eq,
                                       uint64_t* out_point, uint64_t out_finals[16]);
int cozk_worker_spartan_second_sumcheck(cozk_ctx* ctx, const cozk_worker_params* wp, const cozk_star_net* star,
                                        cozk_poly* z, cozk_poly* a, cozk_poly* b, cozk_poly* c,
                                        const uint64_t coef[12], uint64_t* out_point, uint64_t out_finals[16]);

/* ---------------------------------------------------------------- in-process harness ------- */
/* Counterpart of the reference's runner (co-jolt/examples/rep3_jolt.rs:118-317, run_3_party_jolt.sh):
 * synthesises one trace's witness, runs every party on its own thread / ctx and the
 * coordinator on the caller's thread, and verifies the assembled proof. */
typedef struct cozk_harness cozk_harness;
typedef struct cozk_harness_config {
    int mode;          /* COZK_MODE_PLAIN: one party (plain prover); COZK_MODE_REP3: three parties */
    int log_n;         /* padded trace length N = 2^log_n ("cycles") */
    int n_fr;          /* committed polynomials with uniform Fr scalars (shared in REP3) */
    int n_u16;         /* public u16-valued polynomials */
    int n_u32;         /* public u32-valued polynomials */
    int n_flags;       /* public 0/1 flag polynomials */
    int n_small;       /* shared polynomials of length N/16 (own batch_commit + own opening) */
    int gp_batch;      /* circuits in the dense grand product */
    int gp_log_leaves; /* log2(interleaved leaves per circuit) */
    int precompute;    /* build the 16-window SRS table */
    int devices[3];    /* HIP device per party */
    uint64_t seed;
    int log_workers;   /* worker sub-nets per party: 2^log_workers workers, each holding one high-variable chunk
                          of every polynomial and gp_batch / 2^log_workers circuits (reference: split_poly,
                          dense_mlpoly.rs:275-301; co-jolt/README.md:44).  0 = the single-worker path. */
    int worker_devices[8]; /* HIP device per worker index (in-process form) */
    int leaf_fingerprints; /* 0: the grand-product leaves are seeded random shares (cloned per prove);
                              1: compute_leaves (K11): after the commitments the coordinator sends (gamma, tau) and every
                              circuit's leaves are fingerprints of committed columns -- read leaves gamma u16 + gamma^2 u32 +
                              gamma^3 flag + gamma^k shared - tau over the N cycles, then the write leaves (+ gamma^(k+1)), as
                              the bytecode instance lays them out (jolt/vm/bytecode/worker.rs:57-100).  Needs
                              gp_log_leaves == log_n + 1, n_fr >= 1 and log_workers == 0. */
} cozk_harness_config;
typedef struct cozk_harness_result {
    int verified; /* 1 accepted, 0 rejected, -1 verifier not run */
    double wall_ms;
    /* worker-side phase times, max over parties */
    double t_commit_ms, t_gp_construct_ms, t_gp_prove_ms, t_eval_ms, t_open_ms, t_worker_ms;
    uint64_t bytes_star_up, bytes_star_down, bytes_ring, star_messages;
    uint64_t proof_len;
    uint8_t proof_digest[32]; /* SHA-256 of the serialized proof */
    double t_hub_wait_ms;     /* cozk_harness_prove_distributed: time this participant's coordinator copy spent inside the hub's
                               * all_gather (waiting for the slowest participant of every star exchange); 0 in-process */
    uint64_t hub_exchanges;
} cozk_harness_result;
int cozk_harness_create(const cozk_harness_config* cfg, cozk_harness** out);
const char* cozk_harness_error(const cozk_harness* h);
int cozk_harness_destroy(cozk_harness* h);
int cozk_harness_prove(cozk_harness* h, int verify, cozk_harness_result* res);
int cozk_harness_proof_bytes(const cozk_harness* h, uint8_t* out, size_t cap);

/* Distributed form (BASELINE config 3, "one MI355X per party"): one party per process / GPU.  Every process
 * runs its own copy of the deterministic coordinator; a star gather becomes an all-gather of the parties'
 * messages through the host transport (torch.distributed / RCCL), the ring reshare goes through
 * cozk_ring_net on device pointers.  all_gather: participant i contributes `len` bytes; recv holds
 * n_participants slots of `cap` bytes, lens[i] the received lengths.  Returns 0 on success. */
typedef struct cozk_hub_net {
    void* user;
    int n_participants;
    int my_index;
    int (*all_gather)(void* user, const void* send, size_t len, void* recv, size_t cap, size_t* lens);
} cozk_hub_net;
int cozk_harness_create_party(const cozk_harness_config* cfg, int local_party, cozk_harness** out);
/* one (party, worker) participant of the worker sub-net form (cfg.log_workers > 0): the hub then has
 * nparties * 2^log_workers participants, index = worker * nparties + party; ring may be NULL for MODE_PLAIN */
int cozk_harness_create_participant(const cozk_harness_config* cfg, int local_party, int local_worker,
                                    cozk_harness** out);
int cozk_harness_prove_distributed(cozk_harness* h, const cozk_hub_net* hub, const cozk_ring_net* ring,
                                   int verify, cozk_harness_result* res);
/* Single-node hub: byte all-gather through one POSIX shared-memory segment (lock-free, double-buffered
 * mailboxes; ~1 us per exchange) for one-process-per-GPU runs on ONE node -- the per-round star messages
 * (mpc-net/src/mpc_star.rs:29-66) are a few hundred bytes and latency-bound.  Exactly one participant opens
 * with create = 1 (fails if `name` exists), the others attach with create = 0 afterwards (order the two
 * with the launcher's barrier); cozk_shm_hub_net fills a cozk_hub_net whose callbacks stay valid until
 * close.  Waits are bounded (default 120 s, set_timeout_ms) and any participant can abort all of them. */
typedef struct cozk_shm_hub cozk_shm_hub;
int cozk_shm_hub_open(const char* name, int create, int n_participants, int my_index, size_t slot_bytes,
                      cozk_shm_hub** out);
int cozk_shm_hub_net(cozk_shm_hub* hub, cozk_hub_net* out);
int cozk_shm_hub_unlink(cozk_shm_hub* hub);
int cozk_shm_hub_set_timeout_ms(cozk_shm_hub* hub, uint64_t ms);
void cozk_shm_hub_abort(cozk_shm_hub* hub);
void cozk_shm_hub_close(cozk_shm_hub* hub);
/* synchronous raw copy between any two pointers (device or host) on the context's stream: lets a host
 * transport stage ring payloads in buffers of its own (e.g. torch tensors) */
int cozk_copy(cozk_ctx* ctx, void* dst, const void* src, size_t nbytes);
cozk_ctx* cozk_harness_ctx(cozk_harness* h, int party);

/* ---------------------------------------------------------------- co-noir-spartan harness ---- */
/* BASELINE config 4 restated: a satisfied synthetic R1CS with 2^log_n constraints and
 * variables, 3 entries per row shared by A, B, C; worker side of SpartanProverWorker::prove
 * (co-noir-spartan/co-spartan/src/worker.rs:119-300): zero_round, PST commit of z, first (degree-3) and
 * second (degree-2) sumcheck, z(ry), distributed_open; the calling thread is coordinator + verifier.
 * MODE_REP3 runs three parties (devices[p]); MODE_PLAIN one. */
typedef struct cozk_spartan cozk_spartan;
typedef struct cozk_spartan_config {
    int mode;
    int log_n;
    int precompute; /* window table for the SRS (as cozk_harness_config) */
    int devices[3];
    uint64_t seed;
    int lookup_round; /* 1: also the PUBLIC part of the protocol with one public worker (party 0's GPU):
                       * third_round's tail (worker.rs:296-343: val_a, val_b, val_c, commitments of eq_tilde_rx / ry) and
                       * fourth_round (worker.rs:398-575: two logup lookups, distributed sumcheck, batch opening of 15
                       * polynomials under ck_index); its proof part is appended and verified (spartan/src/logup.rs:117-190) */
    int log_pub_workers; /* 0..3 (needs lookup_round): 2^k public workers, each on chunk j of the index as setup.rs's split_ipk
                          * deals it (rows / cols / val / freq chunks, the SRS slice of ck_index with its generator scaled by
                          * eq(t_high, j)); the coordinator sums val_a, val_b, val_c and the commitments (coordinator.rs:425-475),
                          * sums the sumcheck messages of the first qv - k rounds, proves the last k rounds itself on the
                          * gathered finals (distributed_sumcheck_coordinator, coordinator.rs:748-811), and finishes the batched
                          * opening's last k folds.  The proof bytes equal the one-worker proof. */
} cozk_spartan_config;
typedef struct cozk_spartan_result {
    int verified; /* 1 ok, 0 rejected, -1 not run */
    double wall_ms;
    double t_zero_round_ms, t_commit_ms, t_sumcheck1_ms, t_matrix_build_ms, t_sumcheck2_ms, t_open_ms, t_worker_ms;
    uint64_t bytes_star_up, bytes_star_down, star_messages;
    uint64_t proof_len;
    uint8_t proof_digest[32]; /* SHA-256 of the serialized proof */
    double t_lookup_ms;       /* cfg.lookup_round: the public worker's third-round tail + fourth round */
    int pub_workers;          /* public workers that proved the lookup round (1 << log_pub_workers) */
    uint64_t pub_star_messages, pub_bytes_up, pub_bytes_down; /* their star (log_pub_workers > 0) */
} cozk_spartan_result;
int cozk_spartan_create(const cozk_spartan_config* cfg, cozk_spartan** out);
const char* cozk_spartan_error(const cozk_spartan* h);
int cozk_spartan_destroy(cozk_spartan* h);
int cozk_spartan_prove(cozk_spartan* h, int verify, cozk_spartan_result* res);
int cozk_spartan_proof_bytes(const cozk_spartan* h, uint8_t* out, size_t cap);

/* ---------------------------------------------------------------- instruction-lookups harness ---- */
/* SURVEY.md 8(f)1 restated synthetically: the toggled / sparse batched grand product of Lasso's read / write memory
 * checking (jolt/vm/instruction_lookups/worker.rs:763-859 + subprotocols/sparse_grand_product.rs): n_pairs memories, each
 * with one public 0/1 flag column over the 2^log_n cycles (density_pct % set) shared by its read and write circuit, and a
 * shared fingerprint vector per circuit.  Workers on the GPU(s), coordinator + plain verifier on the calling thread. */
typedef struct cozk_lookups cozk_lookups;
typedef struct cozk_lookups_config {
    int mode;        /* COZK_MODE_PLAIN / COZK_MODE_REP3 */
    int log_n;       /* cycles N = 2^log_n */
    int n_pairs;     /* memories: 2 * n_pairs circuits */
    int density_pct; /* share of the flags that are set, 0..100 */
    int devices[3];
    uint64_t seed;
    int log_workers; /* worker sub-nets of the primary sumcheck (jolt/vm/instruction_lookups/worker.rs:194-360, coordinator.rs:
                        97-150): 2^log_workers workers per party, each proving the first log_n - log_workers rounds on its
                        high-variable chunk of every polynomial; worker 0 finishes the rest on the gathered finals.  The
                        workers of a party are time-sliced on its context here (one GPU each in a real deployment) */
    int primary; /* 1: run Lasso's primary sumcheck first (n_pairs E polynomials, a five-instruction synthetic table of the
                    three collation forms, lookup_outputs = sum_i flag_i g_i(E)); the proof then starts with its part */
    int mix;     /* instruction mix of the synthetic trace: 0 = uniform over the 27 RV32I instructions (37 % of the cycles run a
                    multiplicative collation); 1 = trace-shaped, the proportions of a sha2-chain guest (ADD / XOR / AND / OR / SLL /
                    SRL dominant, ~6 % multiplicative; csrc/host/lookups_harness.hpp LOOKUPS_SHA2_MIX) */
} cozk_lookups_config;
typedef struct cozk_lookups_result {
    int verified; /* 1 ok, 0 rejected, -1 not run */
    double wall_ms, t_primary_ms, t_construct_ms, t_prove_ms, t_worker_ms;
    uint64_t bytes_star_up, bytes_star_down, bytes_ring, star_messages;
    uint64_t proof_len;
    uint8_t proof_digest[32];
} cozk_lookups_result;
int cozk_lookups_create(const cozk_lookups_config* cfg, cozk_lookups** out);
const char* cozk_lookups_error(const cozk_lookups* h);
int cozk_lookups_destroy(cozk_lookups* h);
int cozk_lookups_prove(cozk_lookups* h, int verify, cozk_lookups_result* res);
int cozk_lookups_proof_bytes(const cozk_lookups* h, uint8_t* out, size_t cap);

/* ---------------------------------------------------------------- Spartan outer-sumcheck harness ---- */
/* SURVEY.md 8(f)2 restated synthetically: Az / Bz / Cz of a satisfied constraint system over 14 witness columns (shared and
 * public; 5 uniform + 2 cross-step constraints, 8 rows per step; csrc/host/outer_harness.hpp) and co-jolt's outer cubic
 * sumcheck with the Gruen split-eq (r1cs/spartan/worker.rs:63-100,277-300); coordinator + plain verifier on the caller. */
typedef struct cozk_outer_harness cozk_outer_harness;
typedef struct cozk_outer_config {
    int mode;
    int log_steps; /* steps (cycles) = 2^log_steps; rows = 8 (toy system) or 128 (Jolt constraint set) per step */
    int devices[3];
    uint64_t seed;
    int system; /* 0: the 7-constraint toy system; 1: the reference's own constraint SET (co-jolt/src/r1cs/constraints.rs:39-257,
                 * 70 uniform + 2 cross-step constraints over the 78 inputs of r1cs/inputs.rs) on a synthetic satisfying trace */
    int full;   /* 1: the whole Rep3UniformSpartanProver::prove (r1cs/spartan/worker.rs:63-273: outer + inner + shift sumchecks,
                 * two batch_evaluate + opening appends); 0: the outer sumcheck alone */
} cozk_outer_config;
typedef struct cozk_outer_result {
    int verified;
    double wall_ms, t_build_ms, t_prove_ms, t_worker_ms;
    double t_outer_ms, t_inner_ms, t_shift_ms, t_openings_ms; /* cfg.full: the parts of t_prove_ms */
    uint64_t bytes_star_up, bytes_star_down, star_messages;
    uint64_t proof_len;
    uint8_t proof_digest[32];
} cozk_outer_result;
int cozk_outer_harness_create(const cozk_outer_config* cfg, cozk_outer_harness** out);
const char* cozk_outer_harness_error(const cozk_outer_harness* h);
int cozk_outer_harness_destroy(cozk_outer_harness* h);
int cozk_outer_harness_prove(cozk_outer_harness* h, int verify, cozk_outer_result* res);
int cozk_outer_harness_proof_bytes(const cozk_outer_harness* h, uint8_t* out, size_t cap);

/* ---- ONE chained co-jolt worker flow (JoltRep3Prover::prove, co-jolt/src/jolt/vm/jolt/worker.rs:175-266, against its coordinator
 * jolt/vm/jolt/coordinator.rs:118-222): commit-all -> bytecode memory checking -> instruction lookups (primary sumcheck, toggled
 * read / write + dense init / final grand products) -> read-write memory checking + output check -> Spartan (outer + inner + shift) ->
 * ONE reduce_and_prove over all accumulated openings; one transcript, one opening accumulator, all leaves K11 fingerprints of
 * committed columns.  Synthetic Jolt-shaped witness (csrc/host/flow_harness.hpp); oracle/pyflow.py restates it. */
typedef struct cozk_flow cozk_flow;
typedef struct cozk_flow_config {
    int mode;
    int log_n;        /* trace length 2^log_n */
    int log_m;        /* subtable / memory size M of the instruction lookups (Jolt: 16) */
    int log_b;        /* bytecode size */
    int log_mem;      /* read-write memory size */
    int n_mem;        /* NUM_MEMORIES of the instruction lookups (Jolt RV32I: 54) */
    int n_subtables;  /* Subtables::COUNT */
    int devices[3];
    uint64_t seed;
    int precompute;   /* the SRS window table (cozk_bases_upload) */
    int small_witness; /* 0: read / final counters, E polynomials and memory values are uniform field elements -- what a Rep3 party
                        * commits to (its share of every value is uniform) and an upper bound for a plain prover; 1: they are as wide
                        * as a real trace makes them (counters < 2^log_n, subtable entries and memory words 32 bits), so a PLAIN
                        * prover's commitments fill 2 of the 16 windows; Rep3 shares stay uniform either way */
} cozk_flow_config;
typedef struct cozk_flow_result {
    int verified;
    double wall_ms, t_commit_ms, t_bytecode_ms, t_primary_ms, t_lookups_gp_ms, t_rw_ms, t_spartan_ms, t_open_ms, t_worker_ms;
    double t_spartan_build_ms;
    uint64_t bytes_star_up, bytes_star_down, bytes_ring, star_messages;
    uint64_t n_polys, n_openings;
    uint64_t proof_len;
    uint8_t proof_digest[32];
} cozk_flow_result;
int cozk_flow_create(const cozk_flow_config* cfg, cozk_flow** out);
const char* cozk_flow_error(const cozk_flow* h);
int cozk_flow_destroy(cozk_flow* h);
size_t cozk_flow_num_polys(const cozk_flow* h);
cozk_ctx* cozk_flow_ctx(cozk_flow* h, int party);
int cozk_flow_prove(cozk_flow* h, int verify, cozk_flow_result* res);
int cozk_flow_proof_bytes(const cozk_flow* h, uint8_t* out, size_t cap);

/* ---------------------------------------------------------------- profiling ---------------- */
/* HIP-event timing of the dominant kernel (MSM bucket accumulation, k_msm_accum0) on the ctx stream,
 * for bench.py's roofline object: launches, total ms, point additions issued, and the algorithmic
 * bytes of those launches (n x (64 B base + scalar bytes) per MSM). */
int cozk_prof_enable(cozk_ctx* ctx, int on);
int cozk_prof_read(cozk_ctx* ctx, uint64_t* launches, double* total_ms, uint64_t* point_adds,
                   uint64_t* alg_bytes);
/* the same for the HBM-bound kernels of the polynomial seam: slot 0 k_poly_eval_chi, 1 k_poly_lincomb,
 * 2 k_layer_bind_cubic, 3 k_msm_scatter_lds, 4 k_layer_output (cozk_prof_kernel_name; NULL past the last slot);
 * alg_bytes = the algorithmic bytes of SURVEY.md 8d for those launches (stated per kernel in DESIGN.md 4) */
const char* cozk_prof_kernel_name(int slot);
int cozk_prof_read_kernel(cozk_ctx* ctx, int slot, uint64_t* launches, double* total_ms, uint64_t* alg_bytes);
/* Fq Montgomery-multiply micro-benchmark: `iters` dependent products per lane on `lanes` lanes;
 * returns elapsed ms (HIP events) -- the measured integer-ALU peak. */
int cozk_bench_montmul(cozk_ctx* ctx, size_t lanes, int iters, int variant, double* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* COZK_H */
