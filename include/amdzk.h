/* amdzk — C ABI of the MI355X (gfx950) Halo2/KZG prover backend.
 *
 * This is the drop-in boundary. The reference (anon-aadhaar-halo2) consumes the
 * prover only through the Rust API of
 *     halo2_proofs 0.2.0 @ PSE v2023_01_20   (/root/reference/Cargo.lock:469-471)
 *     halo2curves  0.3.1                      (/root/reference/Cargo.lock:484-486)
 * re-exported by halo2-base (/root/reference/src/lib.rs:15-18, src/signal.rs:1-5,
 * src/conditional_secrets.rs:1-5, src/timestamp.rs:1-4, src/chip.rs:6). A fork of halo2_proofs
 * patched in with Cargo [patch] redirects the functions named next to each entry point below to
 * this library (INTEGRATION.md shows the `extern "C"` block and the call sites).
 *
 * Number format = halo2curves' in-memory format, so Rust slices cross the boundary untouched:
 *   Fr / Fq      : 4 x u64 little-endian limbs, Montgomery form, R = 2^256            (32 B)
 *   G1Affine     : {x: Fq, y: Fq}; (0,0) is the identity                              (64 B)
 *   G1 (Jacobian): {x, y, z: Fq}; z = 0 is the identity                               (96 B)
 * Points returned by this library are always normalised: z = 1 (Montgomery one) or the identity
 * (0, 1, 0) — equal as group elements to what the CPU prover computes; its own (x,y,z) depend on
 * rayon's thread count, and every caller normalises before writing to the transcript.
 *
 * Ownership: the caller owns every host buffer; the library never keeps a host pointer after a call
 * returns. Device objects are opaque handles freed by the caller.
 * Errors: every function returns 0 on success and a negative amdzk_status on failure; the message is
 * available from amdzk_last_error(). Nothing throws or aborts across this boundary.
 * Threading: one amdzk_ctx per (GPU, host thread). Calls on one ctx are ordered on one HIP stream.
 * There is NO CPU fallback: without a usable gfx950 device amdzk_init fails with AMDZK_E_NO_DEVICE.
 */
#ifndef AMDZK_H
#define AMDZK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct amdzk_ctx amdzk_ctx;
typedef struct amdzk_srs amdzk_srs;

typedef enum {
  AMDZK_OK = 0,
  AMDZK_E_NO_DEVICE = -1,
  AMDZK_E_INVALID = -2,   /* bad argument (null pointer, size mismatch, k out of range) */
  AMDZK_E_HIP = -3,       /* a HIP runtime call failed */
  AMDZK_E_NOMEM = -4,
  AMDZK_E_UNSUPPORTED = -5
} amdzk_status;

/* basis selector for MSM: ParamsKZG.g (monomial) or ParamsKZG.g_lagrange */
#define AMDZK_BASIS_G 0
#define AMDZK_BASIS_G_LAGRANGE 1

/* flags for amdzk_ntt_fr* */
#define AMDZK_NTT_SCALE_NINV 1u /* multiply the result by 1/2^log_n (EvaluationDomain::ifft's divisor) */

/* ---- context ------------------------------------------------------------------------------- */
int amdzk_init(int device_id, amdzk_ctx** out);
void amdzk_destroy(amdzk_ctx* ctx);
const char* amdzk_last_error(const amdzk_ctx* ctx);
/* ABI version of this header: major*1000 + minor. */
int amdzk_version(void);
/* "amdzk <abi> src=<16 hex digits> arch=gfx950": the hash is over the comment-stripped kernel sources and build flags the
 * binary was made from (csrc/Makefile stamps it; the same function as bench.kernel_src_hash()). A static string. */
const char* amdzk_build_info(void);
/* Run subsequent work of this ctx on an existing hipStream_t (e.g. torch's current stream);
 * NULL restores the ctx's own stream. */
int amdzk_set_stream(amdzk_ctx* ctx, void* hip_stream);
/* HIP's current device is per host thread; a ctx belongs to ONE device. Every entry point below pins the calling
 * thread to the ctx's device for the duration of the call and restores the caller's device afterwards, so a ctx
 * may be created on one thread and used from another (one call at a time). amdzk_ctx_device returns that device;
 * the *_check_affinity calls verify that the ctx's stream, workspaces and tables / a pointer / a proving key's
 * buffers really live there (AMDZK_E_INVALID with a message otherwise) — a cheap assertion for multi-GPU hosts. */
int amdzk_ctx_device(const amdzk_ctx* ctx);
int amdzk_ctx_check_affinity(amdzk_ctx* ctx);
int amdzk_ptr_check_affinity(amdzk_ctx* ctx, const void* dptr);
/* How this context's calls wait for the device: AMDZK_WAIT_SPIN (default; hipStreamSynchronize — lowest latency, the
 * waiting thread keeps a core busy) or AMDZK_WAIT_BLOCK: the thread POLLS a completion event — hipEventQuery, 20 us of
 * yielding, then 50-us sleeps — and leaves its core to the threads that have kernels to launch. Every polled wait ends up
 * to one sleep quantum (about 50 us) late and a proof has 8-10 of them, so a lone proof is about 0.3-0.5 ms slower; it is
 * for hosts that keep more proofs in flight than they have cores to spare — one GPU's share of an 8-GPU host is 2 cores on
 * the reference box, where spinning costs 17 % of the rate, DESIGN.md §5. (The runtime's own blocking waits are not used:
 * hipEventBlockingSync events still spin on this ROCm.) Default from AMDZK_HOST_WAIT=spin|block when the ctx is made.
 * A host process that runs the device with hipDeviceScheduleBlockingSync (hipSetDeviceFlags) gets polling waits on EVERY
 * host wait of this library whatever this setting says: amdzk_init reads hipGetDeviceFlags and the library then never
 * enters the runtime's blocking waits (first screen). */
#define AMDZK_WAIT_SPIN 0
#define AMDZK_WAIT_BLOCK 1
int amdzk_set_host_wait(amdzk_ctx* ctx, int mode);
int amdzk_sync(amdzk_ctx* ctx);

/* ---- device memory (plain hipMalloc'd bytes; any device pointer of the same GPU is accepted by
 *      the *_dev entry points, including torch tensors' data_ptr()) ------------------------- */
int amdzk_dev_alloc(amdzk_ctx* ctx, size_t bytes, void** dptr);
int amdzk_dev_free(amdzk_ctx* ctx, void* dptr);
int amdzk_dev_upload(amdzk_ctx* ctx, void* dptr, const void* host, size_t bytes);
int amdzk_dev_download(amdzk_ctx* ctx, void* host, const void* dptr, size_t bytes);
int amdzk_dev_memset(amdzk_ctx* ctx, void* dptr, int byte, size_t bytes);
/* Streaming one witness per proof (what a caller of create_proof after Circuit::synthesize,
 * /root/reference/src/lib.rs:328-397, does): synthesize into pinned host memory (amdzk_host_alloc), start the
 * upload of the NEXT proof's witness with amdzk_dev_upload_async — it runs on a second HIP stream of the ctx,
 * beside the kernels of the proof in progress — and call amdzk_upload_fence before the create_proof that reads
 * it: work submitted to the ctx after the fence runs after every upload issued before it. The fence never blocks
 * the host: uploads that have already finished (the pipelined case) cost one event query and put no dependency on
 * the ctx's stream; one still in flight is waited for on the device. The source of an asynchronous upload must stay untouched until a fence + amdzk_sync, or the
 * create_proof that follows the fence, has returned. */
int amdzk_host_alloc(amdzk_ctx* ctx, size_t bytes, void** hptr);
int amdzk_host_free(amdzk_ctx* ctx, void* hptr);
int amdzk_dev_upload_async(amdzk_ctx* ctx, void* dptr, const void* host, size_t bytes);
int amdzk_upload_fence(amdzk_ctx* ctx);

/* ---- SRS: replaces the resident part of poly::kzg::commitment::ParamsKZG {g, g_lagrange} [UP]
 * g, g_lagrange: n = 2^k G1Affine each (either may be NULL if that basis is never used).
 * The bases are uploaded once and expanded on the device into per-window multiples
 * (2^(c*w) * g[i]) so that every MSM is bucket-accumulation only (MSM). */
int amdzk_srs_upload(amdzk_ctx* ctx, const uint64_t* g, const uint64_t* g_lagrange, uint32_t k,
                     amdzk_srs** out);
/* ParamsKZG::setup(k, rng) [UP] with the trapdoor s supplied by the caller (as unsafe as upstream's
 * setup: tests and benchmarks only): g[i] = s^i G, g_lagrange[i] = L_i(s) G, built on the device.
 * g_out / g_lagrange_out (2^k G1Affine each, host) may be NULL. */
int amdzk_srs_setup(amdzk_ctx* ctx, uint32_t k, const uint64_t s[4], amdzk_srs** out, uint64_t* g_out,
                    uint64_t* g_lagrange_out);
/* ParamsKZG::{write, read} [UP]: k (u32 LE) | n x g | n x g_lagrange (halo2curves 32-byte compressed
 * G1: x little-endian, bit 7 of byte 31 = parity of y, all-zero = identity) | g2 | s_g2 (64 bytes each,
 * opaque to the prover and passed through). Point (de)compression runs on the device; read rejects
 * encodings that are not on the curve. */
size_t amdzk_srs_serialized_size(uint32_t k);
int amdzk_srs_write(amdzk_ctx* ctx, const amdzk_srs* srs, const uint8_t g2[64], const uint8_t s_g2[64],
                    uint8_t* out, size_t cap);
int amdzk_srs_read(amdzk_ctx* ctx, const uint8_t* data, size_t len, amdzk_srs** out,
                   uint8_t g2_out[64], uint8_t s_g2_out[64]);
/* ParamsKZG::downsize(new_k) [UP]: params for a smaller domain from the same trapdoor: g[..2^new_k]
 * is kept and g_lagrange recomputed with arithmetic::g_to_lagrange (below). Upstream shrinks in place;
 * here a new handle is returned and `srs` stays valid (free both). Fails if new_k > k. */
int amdzk_srs_downsize(amdzk_ctx* ctx, const amdzk_srs* srs, uint32_t new_k, amdzk_srs** out);
/* ParamsKZG::get_g() [UP] (basis 0) / the g_lagrange vector (basis 1): 2^k G1Affine to the host. */
int amdzk_srs_get(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, uint64_t* out);
void amdzk_srs_free(amdzk_ctx* ctx, amdzk_srs* srs);
/* arithmetic::g_to_lagrange(g_projective, k) [UP]: g_lagrange = (1/n) FFT_{omega^-1}(g) over G1 — radix-2
 * butterflies whose twiddle products are scalar multiplications — normalised to affine.
 * g, g_lagrange_out: 2^k G1Affine on the host. */
int amdzk_g_to_lagrange(amdzk_ctx* ctx, const uint64_t* g, uint32_t k, uint64_t* g_lagrange_out);

/* ---- MSM: replaces arithmetic::best_multiexp(coeffs, bases) as called from
 * ParamsKZG::commit / commit_lagrange [UP].
 * scalars: len x Fr (len <= 2^k; bases[0..len) are used, as `&g[..len]` in the original).
 * out: one normalised Jacobian point (12 x u64). */
int amdzk_msm_g1(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, const uint64_t* scalars,
                 size_t len, uint64_t out_jacobian[12]);
/* ncols independent MSMs over the same bases in one submission (one per committed column). */
int amdzk_msm_g1_batch(amdzk_ctx* ctx, const amdzk_srs* srs, int basis,
                       const uint64_t* const* scalars, size_t ncols, size_t len,
                       uint64_t* out_jacobian /* ncols x 12 */);
/* Same, scalars already resident: column c starts at d_scalars + c*col_stride (in Fr elements). */
int amdzk_msm_g1_dev(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, const void* d_scalars,
                     size_t ncols, size_t len, size_t col_stride,
                     uint64_t* out_jacobian /* host, ncols x 12 */);

/* ---- NTT: replaces arithmetic::best_fft(a, omega, log_n) [UP] (rows a3, a4).
 * In place, natural order in, natural order out: a[j] <- sum_i a[i] * omega^(i*j). */
int amdzk_ntt_fr(amdzk_ctx* ctx, uint64_t* a, uint32_t log_n, const uint64_t omega[4],
                 uint32_t flags);
/* ncols resident columns, column c at d_a + c*col_stride (Fr elements), each of 2^log_n. */
int amdzk_ntt_fr_dev(amdzk_ctx* ctx, void* d_a, uint32_t log_n, const uint64_t omega[4],
                     uint32_t flags, size_t ncols, size_t col_stride);

/* ncols host vectors of 2^log_n in one submission. */
int amdzk_ntt_fr_batch(amdzk_ctx* ctx, uint64_t* const* cols, size_t ncols, uint32_t log_n,
                       const uint64_t omega[4], uint32_t flags);

/* ---- representation changes on resident data: Fr::from_raw (canonical 4 x u64 -> Montgomery) and
 * Fr::to_repr (Montgomery -> canonical), n elements in place. Values must be < r. */
int amdzk_fr_from_raw_dev(amdzk_ctx* ctx, void* d_a, size_t n);
int amdzk_fr_to_repr_dev(amdzk_ctx* ctx, void* d_a, size_t n);

/* ---- EvaluationDomain: replaces poly::domain::EvaluationDomain [UP] (rows a4-a6).
 * amdzk_domain_new(j, k) = EvaluationDomain::new(j, k): j = cs.degree(), n = 2^k,
 * extended_k = smallest e with 2^e >= n*(j-1). Columns are device-resident; `col_stride` and the
 * in/out strides are in Fr elements. */
typedef struct amdzk_domain amdzk_domain;
int amdzk_domain_new(amdzk_ctx* ctx, uint32_t j, uint32_t k, amdzk_domain** out);
void amdzk_domain_free(amdzk_ctx* ctx, amdzk_domain* dom);
uint32_t amdzk_domain_k(const amdzk_domain* dom);
uint32_t amdzk_domain_extended_k(const amdzk_domain* dom);
/* what: 0 omega, 1 omega_inv, 2 extended_omega, 3 extended_omega_inv, 4 g_coset (= Fr::ZETA),
 * 5 g_coset_inv, 6 ifft_divisor, 7 extended_ifft_divisor */
int amdzk_domain_constant(const amdzk_domain* dom, int what, uint64_t out[4]);
/* lagrange_to_coeff / coeff_to_lagrange: in place on ncols columns of 2^k. */
int amdzk_lagrange_to_coeff_dev(amdzk_ctx* ctx, const amdzk_domain* dom, void* d_cols, size_t ncols,
                                size_t col_stride);
int amdzk_coeff_to_lagrange_dev(amdzk_ctx* ctx, const amdzk_domain* dom, void* d_cols, size_t ncols,
                                size_t col_stride);
/* coeff_to_extended: 2^k coefficients in -> 2^extended_k evaluations on the zeta coset out
 * (distribute_powers_zeta, zero padding and the NTT in one pass structure). d_ext != d_coeff. */
int amdzk_coeff_to_extended_dev(amdzk_ctx* ctx, const amdzk_domain* dom, const void* d_coeff,
                                size_t in_stride, void* d_ext, size_t out_stride, size_t ncols);
/* extended_to_coeff: in place on 2^extended_k values; the first n*(j-1) entries are the result. */
int amdzk_extended_to_coeff_dev(amdzk_ctx* ctx, const amdzk_domain* dom, void* d_ext, size_t ncols,
                                size_t col_stride);
/* divide_by_vanishing_poly: a[i] *= 1/((zeta*extended_omega^i)^n - 1), in place. */
int amdzk_divide_by_vanishing_dev(amdzk_ctx* ctx, const amdzk_domain* dom, void* d_ext, size_t ncols,
                                  size_t col_stride);

/* ---- create_proof: replaces plonk::{keygen_pk, create_proof} [UP] for
 * KZGCommitmentScheme<Bn256> + ProverSHPLONK (or ProverGWC, AMDZK_MULTIOPEN_GWC) + Blake2bWrite/Challenge255
 * (or the EVM transcript), one circuit instance (amdzk_create_proof_multi: several),
 * phase-0 advice. The circuit arrives as plain data: what ConstraintSystem holds after
 * Circuit::configure (/root/reference/src/lib.rs:295-326 etc.) and selector compression.
 *
 * Expressions are postfix u32 words, op << 24 | payload:
 *   1 CONST  payload = index into `constants`      2 FIXED / 3 ADVICE / 4 INSTANCE
 *   5 NEG   6 ADD   7 MUL                              payload = column << 8 | (rotation + 128)
 *   8 SCALE  payload = index into `constants` (multiply the top of stack by that constant)
 * `expr_offsets[e] .. expr_offsets[e+1]` delimits expression e inside `expr_words`. The first
 * num_gates expressions are the gate polynomials in order; then, per lookup l,
 * lookup_shape[2l] input expressions followed by lookup_shape[2l+1] table expressions. */
typedef struct amdzk_circuit {
  uint32_t k, num_fixed, num_advice, num_instance, blinding_factors, cs_degree;
  uint32_t num_advice_queries;   const int32_t* advice_queries;   /* (column, rotation) pairs, query order */
  uint32_t num_fixed_queries;    const int32_t* fixed_queries;
  uint32_t num_instance_queries; const int32_t* instance_queries;
  uint32_t num_gates, num_lookups, num_exprs;
  const uint32_t* lookup_shape;  /* 2 x num_lookups */
  const uint32_t* expr_offsets;  /* num_exprs + 1 */
  const uint32_t* expr_words;
  uint32_t num_constants;        const uint64_t* constants;       /* Fr, Montgomery */
  uint32_t num_perm_columns;     const uint32_t* perm_columns;    /* (kind: 0 advice 1 fixed 2 instance, index) */
} amdzk_circuit;
typedef struct amdzk_pk amdzk_pk;
/* fixed_values: num_fixed x 2^k Fr (Lagrange, row-major per column). perm_mapping: for permutation
 * column i and row j the pair (i', j') of permutation::keygen::Assembly::mapping, as
 * perm_mapping[2*(i*n + j) + {0,1}]. transcript_repr: VerifyingKey::transcript_repr (upstream derives
 * it from the Debug string of the pinned VK; the caller supplies it). The proving key owns the
 * per-proof workspace: one proof at a time per key. */
int amdzk_keygen(amdzk_ctx* ctx, const amdzk_srs* srs, const amdzk_circuit* circuit,
                 const uint64_t* fixed_values, const uint32_t* perm_mapping,
                 const uint64_t transcript_repr[4], amdzk_pk** out);
/* The same with the key's modes as explicit flags (amdzk_keygen takes them from the environment: AMDZK_FULL_COSETS=1,
 * AMDZK_SERIAL=1), so that two keys of one process can differ:
 *   AMDZK_KEYGEN_FULL_COSETS  evaluate the quotient numerator on all 2^(extended_k - k) cosets of upstream's extended
 *                             domain instead of cs_degree - 1 of them: upstream's own computation, byte-identical
 *                             also for witnesses that do NOT satisfy the circuit (satisfying ones are identical either way)
 *   AMDZK_KEYGEN_SERIAL       one proof's kernels on the caller's stream only, strictly one after another (no lanes:
 *                             see amdzk_create_proof) */
#define AMDZK_KEYGEN_FULL_COSETS 1u
#define AMDZK_KEYGEN_SERIAL 2u
int amdzk_keygen_ex(amdzk_ctx* ctx, const amdzk_srs* srs, const amdzk_circuit* circuit,
                    const uint64_t* fixed_values, const uint32_t* perm_mapping,
                    const uint64_t transcript_repr[4], uint32_t flags, amdzk_pk** out);
void amdzk_pk_free(amdzk_ctx* ctx, amdzk_pk* pk);
int amdzk_pk_check_affinity(amdzk_ctx* ctx, const amdzk_pk* pk);
/* VerifyingKey commitments: fixed columns (num_fixed x G1Affine), permutation (num_perm_columns x G1Affine). */
int amdzk_pk_commitments(const amdzk_pk* pk, uint64_t* fixed_out, uint64_t* perm_out);
/* instances[c]: instance_lens[c] public inputs of instance column c (host). d_advice: num_advice
 * witness columns of 2^k rows, device-resident, column c at d_advice + c*advice_stride (untouched:
 * blinding happens on an internal copy). rng_seed: ChaCha20Rng::seed_from_u64(seed) drives every
 * Fr::random draw in upstream order. proof_out may be NULL to query *proof_len. */
int amdzk_create_proof(amdzk_ctx* ctx, amdzk_pk* pk, const uint64_t* const* instances,
                       const size_t* instance_lens, const void* d_advice, size_t advice_stride,
                       uint64_t rng_seed, uint8_t* proof_out, size_t proof_cap, size_t* proof_len);
/* Same with a choice of TranscriptWrite: Blake2bWrite/Challenge255 (32-byte compressed points,
 * little-endian scalars) or halo2-solidity-verifier's Keccak256Transcript/ChallengeEvm — the wire
 * format /root/reference/solidity_verifier_contract/contract.sol reads (64-byte uncompressed points,
 * 32-byte big-endian scalars, contract.sol:77-112). */
#define AMDZK_TRANSCRIPT_BLAKE2B 0
#define AMDZK_TRANSCRIPT_KECCAK256_EVM 1
/* OR into `transcript_kind` to open with poly::kzg::multiopen::ProverGWC instead of ProverSHPLONK
 * (SURVEY.md §8(a) row a12): one witness commitment W_z = (sum_j v^j p_j - sum_j v^j p_j(z)) / (X - z) per
 * distinct evaluation point z, points in first-seen query order. */
#define AMDZK_MULTIOPEN_GWC 0x100
int amdzk_create_proof_ex(amdzk_ctx* ctx, amdzk_pk* pk, const uint64_t* const* instances,
                          const size_t* instance_lens, const void* d_advice, size_t advice_stride,
                          uint64_t rng_seed, int transcript_kind, uint8_t* proof_out,
                          size_t proof_cap, size_t* proof_len);
/* ---- several circuit instances in ONE proof: upstream's create_proof(params, pk, &[circuit; N], &[instances; N], rng,
 * transcript) [UP] takes slices; the reference passes one circuit (/root/reference/src/lib.rs:299-300 uses
 * phase 0 only; later phases stay out of scope). Per-circuit steps run circuit after circuit in upstream's order — instance
 * values absorbed, advice blinded and committed, lookups permuted, permutation products, lookup products, then the advice /
 * permutation / lookup evaluations and openings — while the challenges, the random polynomial and h(X) exist once (h folds the
 * instances' terms in one chain with y).
 * A key owns ONE instance's per-proof workspace: amdzk_pk_clone_workspace makes a handle that shares `pk`'s key material
 * (fixed / permutation columns and cosets, programs, domain) and owns one more workspace. It is a complete key for
 * amdzk_create_proof* too (the cheap way to several proofs of one circuit in flight). Free clones (amdzk_pk_free) BEFORE
 * the key they were made from. */
int amdzk_pk_clone_workspace(amdzk_ctx* ctx, const amdzk_pk* pk, amdzk_pk** out);
/* pks[c]: instance c's workspace — the key or clones of it, pairwise different, all on this ctx's device.
 * instances[c][col] / instance_lens[c][col] and d_advice[c] (+ advice_stride) as for amdzk_create_proof_ex; one RNG stream
 * and one transcript for the whole proof. n_circuits = 1 is amdzk_create_proof_ex. More than one instance runs on the
 * caller's stream only (no lanes). */
int amdzk_create_proof_multi(amdzk_ctx* ctx, amdzk_pk* const* pks, size_t n_circuits,
                             const uint64_t* const* const* instances, const size_t* const* instance_lens,
                             const void* const* d_advice, size_t advice_stride, uint64_t rng_seed,
                             int transcript_kind, uint8_t* proof_out, size_t proof_cap, size_t* proof_len);
size_t amdzk_proof_size_multi(const amdzk_pk* pk, size_t n_circuits, int transcript_kind);
/* Exact byte length of the proof for this key, transcript and multiopen scheme (transcript_kind as for
 * amdzk_create_proof_ex, including AMDZK_MULTIOPEN_GWC); a function of the circuit shape only. */
size_t amdzk_proof_size(const amdzk_pk* pk, int transcript_kind);
/* For callers whose `R: RngCore` is not ChaCha20Rng::seed_from_u64: draw
 * amdzk_proof_random_count(pk) scalars with Fr::random(&mut rng) and pass them (Montgomery Fr, in
 * draw order); the prover consumes them exactly where upstream's create_proof draws. */
size_t amdzk_proof_random_count(const amdzk_pk* pk);
int amdzk_create_proof_scalars(amdzk_ctx* ctx, amdzk_pk* pk, const uint64_t* const* instances,
                               const size_t* instance_lens, const void* d_advice,
                               size_t advice_stride, const uint64_t* scalars, size_t scalar_count,
                               int transcript_kind, uint8_t* proof_out, size_t proof_cap,
                               size_t* proof_len);

/* ---- the PLONK layer function by function: the kernels create_proof runs,
 * callable on their own on device-resident columns — for a fork that replaces upstream one function at a time and
 * for the isolated parity tests (tests/test_gpu_plonk_ops.py). Small operands come from the host. */
/* ff::BatchInvert [UP] in place: non-zero elements are inverted, zeros stay zero. */
int amdzk_batch_invert_dev(amdzk_ctx* ctx, void* d_a, size_t n);
/* poly::batch_invert_assigned [UP] (what create_proof does to the synthesized Assigned<F> cells before it commits to them,
 * SURVEY.md Appendix A step 3): out[i] = num[i] * den[i]^-1, zero where den[i] = 0. A Trivial cell is (x, 1), Zero is (0, 1);
 * d_den = NULL means all cells are trivial. In place (d_out = d_num) is allowed. */
int amdzk_batch_invert_assigned_dev(amdzk_ctx* ctx, const void* d_num, const void* d_den, size_t n, void* d_out);
/* The running product of permutation::prover::Argument::commit and lookup::prover::commit_product [UP]: column c
 * (n elements at d_cols + c*col_stride) is replaced by z with z[0] = 1, z[i] = z[i-1] * f[i-1]; with chain != 0 the
 * permutation argument's last_z is threaded through the columns: z_c[0] = z_{c-1}[chain_row]. */
int amdzk_grand_product_dev(amdzk_ctx* ctx, void* d_cols, size_t ncols, size_t n, size_t col_stride,
                            int chain, size_t chain_row);
/* arithmetic::eval_polynomial [UP] for nq (polynomial, point) pairs: out[q] = d_polys[q](points[q]). d_polys: host
 * array of device pointers to n coefficients; points and out on the host. */
int amdzk_eval_poly_dev(amdzk_ctx* ctx, const void* const* d_polys, const uint64_t* points, size_t nq,
                        uint32_t n, uint64_t* out);
/* d_out = (accumulate ? d_out : 0) + sum_j coefs[j] * d_polys[j] over n coefficients (multiopen's linear
 * combinations). d_out must not alias an input. */
int amdzk_poly_axpy_dev(amdzk_ctx* ctx, const void* const* d_polys, const uint64_t* coefs, size_t m,
                        void* d_out, size_t n, int accumulate);
/* arithmetic::kate_division [UP] in place: a(X) -> (a(X) - a(root)) / (X - root); n coefficients, the top one 0. */
int amdzk_kate_div_dev(amdzk_ctx* ctx, void* const* d_polys, const uint64_t* roots, size_t npolys, uint32_t n);
/* lookup::prover::permute_expression_pair [UP] for nlookups pairs of n rows (column l at + l*n, Montgomery form):
 * d_inputs is sorted in place into A', d_permuted_tables_out receives S' (first occurrences aligned, leftovers in
 * ascending order assigned from the last repeated row backwards); rows >= usable come back zero for the caller to
 * blind. AMDZK_E_INVALID "not in table" when an input value is missing from its table. */
int amdzk_permute_expression_pair_dev(amdzk_ctx* ctx, void* d_inputs, const void* d_tables,
                                      void* d_permuted_tables_out, size_t nlookups, uint32_t n, uint32_t usable);
/* evaluation::Evaluator::evaluate_h + divide_by_vanishing_poly + extended_to_coeff + the split into pieces [UP]:
 * d_polys = the key's NP = A + I + 2L + nsets + L committed polynomials in coefficient form, n each, at
 * d_polys + i*poly_stride, in the order advice | instance | A' | S' | permutation products | lookup products;
 * challenges as Montgomery Fr. Writes the cs_degree - 1 pieces of h(X) (n coefficients each, consecutive) to
 * d_pieces_out. Overwrites the key's per-proof workspace. */
int amdzk_quotient_eval_dev(amdzk_ctx* ctx, amdzk_pk* pk, const void* d_polys, size_t poly_stride,
                            const uint64_t theta[4], const uint64_t beta[4], const uint64_t gamma[4],
                            const uint64_t y[4], void* d_pieces_out);
/* What the last create_proof on this key left in its workspace (test hook): what = 0 the NP committed polynomials in
 * coefficient form [NP][n]; 1 the challenges theta, beta, gamma, y; 2 the pieces of h(X) [cs_degree - 1][n].
 * out may be NULL to query *count (in Fr elements). */
int amdzk_pk_inspect(amdzk_ctx* ctx, const amdzk_pk* pk, int what, uint64_t* out, size_t cap, size_t* count);

/* Test hooks for the host pass that prepares quotient-domain programs for the limb-resident interpreter (bounds in
 * units of p, placement of the weak reductions, fusion of `t*x` with the accumulate): the pass on caller-supplied words
 * `op << 24 | arg` (opcodes: csrc/plonk_kernels.hpp ExprOp) — pure host code, callable without a device — and the
 * finalised h(X) program of a key. out may be NULL to query *out_n. */
int amdzk_debug_limb_program(const uint32_t* words, size_t n, uint32_t* out, size_t cap, size_t* out_n, uint32_t* depth);
int amdzk_pk_h_program(const amdzk_pk* pk, uint32_t* out, size_t cap, size_t* out_n);

/* ---- timing / profiling hooks used by bench.py (HIP events on this ctx's stream) ------------ */
int amdzk_timer_start(amdzk_ctx* ctx);
int amdzk_timer_stop(amdzk_ctx* ctx, float* ms); /* synchronises on the stop event */
/* When enabled, every kernel launch of this ctx is bracketed by HIP events; query by kernel name. */
int amdzk_prof_enable(amdzk_ctx* ctx, int on);
int amdzk_prof_reset(amdzk_ctx* ctx);
int amdzk_prof_get(amdzk_ctx* ctx, const char* kernel_name, uint64_t* launches, double* total_ms);
/* Writes a '\n'-separated list "name launches total_ms" into buf. Returns bytes needed. */
size_t amdzk_prof_dump(amdzk_ctx* ctx, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* AMDZK_H */
