// Prover driver: keygen_pk and create_proof for KZG + SHPLONK + Blake2b, one circuit instance,
// phase 0 — halo2_proofs 0.2.0 @ PSE v2023_01_20 [UP] (/root/reference/Cargo.lock:469-471):
//   plonk::keygen::{keygen_vk, keygen_pk}, plonk::prover::create_proof,
//   plonk::{permutation,lookup,vanishing}::prover, poly::kzg::multiopen::shplonk::ProverSHPLONK.
// This is the function the reference's circuits are handed to; the circuit itself
// arrives as a plain-data description of its ConstraintSystem (amdzk_circuit) plus its fixed
// columns, copy-constraint mapping and witness columns. The order of transcript operations and RNG
// draws follows SURVEY.md Appendix A.
//
// Control flow, Fiat-Shamir and the O(columns) bookkeeping stay on the host; every O(n) step is a
// kernel on resident columns: all committed columns of a phase go through ONE batched MSM, all
// polynomials through ONE batched iNTT and ONE batched coset NTT, and the whole h(X) numerator is
// ONE interpreter launch. Host<->device traffic per proof: the blinding scalars and the random
// polynomial up (n*32 B), commitments and evaluations down. The lookup permutation
// (permute_expression_pair) also runs on the device (bitonic sort of the
// canonical values + multiset alignment).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <map>

#include "hostcrypto.hpp"
#include "plonk_kernels.hpp"

using namespace bn254;
using zkhost::Blake2bWrite;
using zkhost::ChaCha20Rng;

// from the other translation units
struct amdzk_srs;
struct amdzk_domain;
int zk_msm_dev_xyzz(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, const Fr* d_scalars, size_t ncols, size_t len, size_t col_stride,
                    G1X** d_out);
int zk_msm_finish(amdzk_ctx* ctx, const G1X* d_res, size_t ncols, uint64_t* out_jac);
int zk_lagrange_to_coeff(amdzk_ctx* ctx, const amdzk_domain* d, const Fr* d_in, size_t in_stride, Fr* d_out, size_t out_stride, size_t ncols);
extern "C" {
int amdzk_domain_new(amdzk_ctx* ctx, uint32_t j, uint32_t k, amdzk_domain** out);
void amdzk_domain_free(amdzk_ctx* ctx, amdzk_domain* d);
uint32_t amdzk_domain_extended_k(const amdzk_domain* d);
int amdzk_domain_constant(const amdzk_domain* d, int what, uint64_t out[4]);
int amdzk_lagrange_to_coeff_dev(amdzk_ctx* ctx, const amdzk_domain* d, void* d_cols, size_t ncols, size_t col_stride);
int amdzk_coeff_to_extended_dev(amdzk_ctx* ctx, const amdzk_domain* d, const void* d_coeff, size_t in_stride, void* d_ext, size_t out_stride,
                                size_t ncols);
int amdzk_extended_to_coeff_dev(amdzk_ctx* ctx, const amdzk_domain* d, void* d_ext, size_t ncols, size_t col_stride);
int amdzk_divide_by_vanishing_dev(amdzk_ctx* ctx, const amdzk_domain* d, void* d_ext, size_t ncols, size_t col_stride);
// prover-private: the quotient domain = nc cosets of the size-n subgroup, data in Montgomery radix 2^261 (poly.hip) —
// what the h(X) program multiplies in
int zk_quotient_plan(amdzk_ctx* ctx, amdzk_domain* d, uint32_t nc);
Fr zk_quotient_coset_g(const amdzk_domain* d, uint32_t c);
int zk_coeff_to_cosets_r261(amdzk_ctx* ctx, const amdzk_domain* d, const Fr* d_coeff, size_t in_stride, Fr* d_out, size_t out_stride,
                            size_t ncols);
int zk_cosets_to_pieces(amdzk_ctx* ctx, const amdzk_domain* d, Fr* d_h, Fr* d_pieces, uint32_t npieces);
int amdzk_fr_to_repr_dev(amdzk_ctx* ctx, void* d_a, size_t n);
int amdzk_fr_from_raw_dev(amdzk_ctx* ctx, void* d_a, size_t n);
}

namespace {

// host-format expression words (include/amdzk.h)
enum : uint32_t { XOP_CONST = 1, XOP_FIXED = 2, XOP_ADVICE = 3, XOP_INSTANCE = 4, XOP_NEG = 5, XOP_ADD = 6, XOP_MUL = 7, XOP_SCALE = 8 };

Fr fr_delta() {  // Fr::DELTA = 7^(2^28)  (contract.sol:440)
  Fr r;
  uint64_t v[4] = {0x870e56bbe533e9a2ULL, 0x5b5f898e5e963f25ULL, 0x64ec26aad4c86e71ULL, 0x09226b6e22c6f0caULL};
  memcpy(r.l, v, 32);
  return to_mont(r);
}

struct Program {
  std::vector<uint32_t> words;
  uint32_t depth = 0, cur = 0;
  bool uses_hot = false;
  ExprInstr* d_instr = nullptr;  // resolved instructions (device)
  // Lagrange-domain programs: instruction indices at which an independent piece starts (the stack is empty there):
  // run_program cuts the program there into up to EXPR_MAX_PARTS parts that run side by side (ExprArgs::nparts)
  std::vector<uint32_t> piece_starts;
  void piece() { piece_starts.push_back((uint32_t)words.size()); }
  // h(X) programs: term j (closed by the j-th OP_ACC) carries the factor beta^term_beta[j] in its power of y
  std::vector<uint32_t> term_beta;
  uint32_t next_beta = 0;
  void op(uint32_t o, uint32_t arg = 0) {
    words.push_back((o << 24) | (arg & 0xffffffu));
    if (o == OP_ACC) {
      term_beta.push_back(next_beta);
      next_beta = 0;
    }
  }
  void push() {
    cur++;
    if (cur > depth) depth = cur;
  }
  void pop() { cur--; }
};

struct RotTable {
  std::vector<int32_t> rots;
  uint32_t index(int32_t r) {
    for (size_t i = 0; i < rots.size(); i++)
      if (rots[i] == r) return (uint32_t)i;
    rots.push_back(r);
    return (uint32_t)rots.size() - 1;
  }
};

bool fr_less_canon(const std::array<uint64_t, 4>& a, const std::array<uint64_t, 4>& b) {
  for (int i = 3; i >= 0; i--)
    if (a[i] != b[i]) return a[i] < b[i];
  return false;
}
std::array<uint64_t, 4> canon(const Fr& a) {
  Fr c = from_mont(a);
  std::array<uint64_t, 4> o;
  memcpy(o.data(), c.l, 32);
  return o;
}

}  // namespace

// Source of the prover's Fr::random draws, in upstream's draw order: either ChaCha20Rng::seed_from_u64
// replayed here, or scalars the caller drew from its own RngCore (amdzk_create_proof_scalars).
struct RandomSource {
  zkhost::ChaCha20Rng* rng = nullptr;
  const uint64_t* scalars = nullptr;
  size_t count = 0, used = 0;
  bool exhausted = false;
  Fr fr() {
    if (rng) return rng->fr();
    if (used >= count) {
      exhausted = true;
      return Fr::zero();
    }
    Fr r;
    memcpy(r.l, scalars + 4 * used, 32);
    used++;
    return r;
  }
};

struct amdzk_pk {
  uint32_t k = 0, ek = 0, bf = 0, degree = 0, F = 0, A = 0, I = 0, S = 0, L = 0, nsets = 0, chunk = 0, qdeg = 0;
  uint32_t nc = 0;       // cosets of the quotient domain (poly.hip zk_quotient_plan): qdeg of the 2^(ek-k) upstream uses
  size_t n = 0, ext = 0;  // ext = nc * n rows: every "extended" column holds [coset][row]
  std::vector<std::pair<int, int>> advice_queries, fixed_queries, instance_queries;
  std::vector<std::pair<int, int>> perm_cols;  // (kind, index)
  std::vector<std::vector<uint32_t>> exprs;
  uint32_t num_gates = 0;
  std::vector<std::pair<uint32_t, uint32_t>> lookup_shape;  // (#inputs, #tables); expressions follow the gates in order
  std::vector<Fr> consts;                                   // circuit constants, then the dynamic ones
  uint32_t c_one = 0, c_theta = 0, c_beta = 0, c_gamma = 0, c_y = 0, c_betainv = 0;
  amdzk_domain* dom = nullptr;
  const amdzk_srs* srs = nullptr;
  Fr transcript_repr, omega, omega_inv;
  std::vector<G1Affine> fixed_commitments, perm_commitments;

  // device: key material
  Fr *fixed_lag = nullptr, *fixed_poly = nullptr, *fixed_coset = nullptr;
  Fr *sigma_lag = nullptr, *sigma_poly = nullptr, *sigma_coset = nullptr;
  Fr *l0_c = nullptr, *llast_c = nullptr, *lactive_c = nullptr, *x_coset = nullptr, *omega_pow = nullptr;
  // delta^j * omega^i ([S][n], Lagrange) and delta^j * X on the quotient cosets ([S][ext], radix 2^261): the identity
  // permutation's columns. With them v + beta delta^j X + gamma = beta (delta^j X + w), w = (v + gamma) / beta — the SAME w
  // that serves v + beta sigma + gamma = beta (sigma + w): three products per permutation column instead of four, in
  // the Lagrange-domain fractions and in h(X) (the beta^m of a set cancels in a fraction and rides on the term's power of y).
  Fr *dxw_lag = nullptr, *dx_coset = nullptr;
  std::vector<uint32_t> h_term_beta_pow;  // per term of the h(X) program: the power of beta its power of y is multiplied by
  // device: per-proof workspace. poly arena order: adv | inst | la | ls | zp | zl
  size_t NP = 0;
  // P: the committed columns' Lagrange values [NP][n] (what commit_lagrange and the Lagrange-domain programs read);
  // PQ: their coefficients [NP][n] (evaluations, multiopen); PC: their values on the quotient domain [NP][ext].
  // Out of place, so that a phase's transforms run on a lane while its commitments and the next phase's programs
  // still read the Lagrange values.
  Fr *P = nullptr, *PQ = nullptr, *PC = nullptr;
  Fr *ci = nullptr, *ct = nullptr;  // [L][n] compressed lookup input / table
  Fr *rnd = nullptr, *hq = nullptr, *hpieces = nullptr, *hpoly = nullptr, *frac = nullptr, *scratch = nullptr, *scan_tmp = nullptr;
  Fr *frac2 = nullptr, *scratch2 = nullptr, *scan_tmp2 = nullptr;  // the lookup products' own scratch: they run beside the permutation products
  Fr *sets_L = nullptr, *sets_N = nullptr, *sets_Q = nullptr, *hx = nullptr;  // SHPLONK buffers
  size_t sets_Q_pairs = 0;  // (set, point) pairs sets_Q holds n coefficients for
  // Lanes (common.hpp): 0 = the caller's ctx, 1 and 2 = its auxiliary streams. AMDZK_KEYGEN_SERIAL / AMDZK_SERIAL=1
  // keeps everything on the caller's stream (one proof's kernels strictly one after another, as in rounds 1-2).
  bool use_lanes = true;
  uint32_t max_sets = 16, max_set_points = 0;
  // What the multiopen argument derives from the key alone, built by the first proof (the polynomials live at fixed
  // addresses in this key's workspace): the evaluation list, the query list and SHPLONK's rotation sets in terms of
  // rotations. Only the ORDER of a set's points (upstream keeps them in a BTreeSet of field elements) depends on x.
  struct Multiopen {
    bool built = false;
    std::vector<std::pair<const Fr*, int>> ev;      // (polynomial, rotation) in the order the evaluations are written
    size_t n_written = 0;                           // ... of which the first n_written go to the transcript
    std::vector<int> rots;                          // distinct rotations, first seen first
    std::vector<uint32_t> ev_rot;                   // per evaluation: index into rots
    std::vector<const Fr*> q_poly;                  // the queries, upstream order
    std::vector<uint32_t> q_rot, q_ev;              // per query: index into rots / into ev
    struct Set {
      std::vector<uint32_t> rot_ids;                // the set's rotations (ascending index into rots)
      std::vector<const Fr*> polys;                 // its polynomials, first seen first
      std::vector<std::vector<uint32_t>> ev_idx;    // [poly][k]: evaluation of polys[poly] at rots[rot_ids[k]]
    };
    std::vector<Set> sets;                          // first seen first
  } mo;
  Fr *lk_ts = nullptr, *lk_left = nullptr;  // lookup permutation: sorted tables, leftovers [L][n]
  // The first lk_const lookups have ONE table expression over fixed columns and constants only: their compressed table
  // does not depend on theta or on the witness, so its sorted canonical form is made once at keygen ([lk_const][n]).
  uint32_t lk_const = 0;
  Fr* lk_ts_const = nullptr;
  // (Permuting the lookups among them that also have ONE input expression before theta exists, on lane C beside the advice
  // commitment, was measured and dropped: a proof alone took 19.1-19.3 ms with it against 18.7-19.0 without, 21
  // proofs x 3 alternating runs, profiles/r03q_constant_tables_and_early_lookups.txt — the small sort kernels stretch the
  // chip-filling commitment by more than they save behind theta; started behind its level-1 kernel instead they stretch
  // its bucket reduction and lane B's transforms: 18.0-18.2 ms against 17.6-17.95.)
  uint32_t* lk_flags = nullptr;              // [4][L][n+8]
  int* d_err = nullptr;
  int* h_err = nullptr;                      // pinned: where create_proof reads d_err (the first word of `pin`'s tail block)
  // misc small device buffers (blinding uploads, points, evals, coefs) and pointer-table scratch, one slice per lane:
  // a slice is reused in stream order by the lane that owns it
  Fr* small_l[3] = {nullptr, nullptr, nullptr};
  void* ptrs_l[3] = {nullptr, nullptr, nullptr};
  Fr* small = nullptr;   // = small_l[0]
  void* ptrs = nullptr;  // = ptrs_l[0]
  size_t small_cap = 0, ptrs_cap = 0;
  // programs
  Program prog_compress, prog_pfrac, prog_lfrac, prog_h;
  RotTable rots;
  Fr* d_consts = nullptr;
  Fr* d_consts261 = nullptr;  // the same table times 32 (= radix 2^261): constants of programs run on the extended domain
  uint32_t h_terms = 0;       // terms of the h(X) program = powers of y its OP_WACC ops index
  Fr* d_ypow = nullptr;       // [h_terms]: y^(h_terms-1-j) in radix 2^261, refreshed per proof
  const Fr** d_cols_lag = nullptr;
  const Fr** d_cols_ext = nullptr;
  Fr** d_outs_compress = nullptr;
  Fr** d_outs_pfrac = nullptr;
  Fr** d_outs_lfrac = nullptr;
  std::vector<void*> allocs;
  // A workspace clone (amdzk_pk_clone_workspace) shares the key material above — columns, cosets, compiled programs,
  // domain, constant tables — with the key it was made from and owns one more circuit instance's per-proof workspace and
  // pointer tables: `allocs` holds only what the clone itself allocated.
  const amdzk_pk* clone_of = nullptr;
  // create_proof over several circuit instances (amdzk_create_proof_multi): the evaluation / query lists over all of
  // them, built by the first such proof on this key for a given list of instance keys
  Multiopen mo_multi;
  std::vector<const amdzk_pk*> mo_multi_keys;
  std::vector<const Fr*> h_cols_lag, h_cols_ext;  // host copies of the slot tables (program resolution)
  // pinned host staging (bump allocator, reset whenever the stream is known to be idle)
  char* pin = nullptr;
  size_t pin_cap = 0, pin_off = 0;

  Fr* adv() { return P; }
  Fr* inst() { return P + (size_t)A * n; }
  Fr* la() { return P + (size_t)(A + I) * n; }
  Fr* ls() { return P + (size_t)(A + I + L) * n; }
  Fr* zp() { return P + (size_t)(A + I + 2 * L) * n; }
  Fr* zl() { return P + (size_t)(A + I + 2 * L + nsets) * n; }
  Fr* q_adv() { return PQ; }
  Fr* q_la() { return PQ + (size_t)(A + I) * n; }
  Fr* q_ls() { return PQ + (size_t)(A + I + L) * n; }
  Fr* q_zp() { return PQ + (size_t)(A + I + 2 * L) * n; }
  Fr* q_zl() { return PQ + (size_t)(A + I + 2 * L + nsets) * n; }
  // slots, Lagrange table
  uint32_t sl_fixed(uint32_t c) { return c; }
  uint32_t sl_adv(uint32_t c) { return F + c; }
  uint32_t sl_inst(uint32_t c) { return F + A + c; }
  uint32_t sl_sigma(uint32_t c) { return F + A + I + c; }
  uint32_t sl_ci(uint32_t l) { return F + A + I + S + l; }
  uint32_t sl_ct(uint32_t l) { return F + A + I + S + L + l; }
  uint32_t sl_la(uint32_t l) { return F + A + I + S + 2 * L + l; }
  uint32_t sl_ls(uint32_t l) { return F + A + I + S + 3 * L + l; }
  uint32_t sl_omega() { return F + A + I + S + 4 * L; }
  uint32_t sl_dxw(uint32_t c) { return F + A + I + S + 4 * L + 1 + c; }
  uint32_t nslots_lag() { return F + A + I + 2 * S + 4 * L + 1; }
  // slots, extended table
  uint32_t se_sigma(uint32_t c) { return F + A + I + c; }
  uint32_t se_zp(uint32_t s) { return F + A + I + S + s; }
  uint32_t se_zl(uint32_t l) { return F + A + I + S + nsets + l; }
  uint32_t se_la(uint32_t l) { return F + A + I + S + nsets + L + l; }
  uint32_t se_ls(uint32_t l) { return F + A + I + S + nsets + 2 * L + l; }
  uint32_t se_l0() { return F + A + I + S + nsets + 3 * L; }
  uint32_t se_llast() { return se_l0() + 1; }
  uint32_t se_lactive() { return se_l0() + 2; }
  uint32_t se_x() { return se_l0() + 3; }
  uint32_t se_dx(uint32_t c) { return se_l0() + 4 + c; }
  uint32_t nslots_ext() { return se_l0() + 4 + S; }
};

namespace {

template <class T>
int dalloc(amdzk_ctx* ctx, amdzk_pk* pk, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
  if (e != hipSuccess) ZK_FAIL(ctx, AMDZK_E_NOMEM, "prover: hipMalloc(%zu) failed: %s", count * sizeof(T), hipGetErrorString(e));
  pk->allocs.push_back(q);
  *p = (T*)q;
  return AMDZK_OK;
}

int h2d(amdzk_ctx* ctx, void* d, const void* h, size_t bytes) {
  if (bytes) ZK_HIP(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
  return AMDZK_OK;
}
// host -> device through the key's pinned staging area: the source may be a temporary, and the copy
// is truly asynchronous (no pageable-memory staging inside the runtime).
int h2d_staged(amdzk_ctx* ctx, amdzk_pk* pk, void* d, const void* h, size_t bytes) {
  if (!bytes) return AMDZK_OK;
  if (!pk->pin || bytes > pk->pin_cap) return h2d(ctx, d, h, bytes);
  size_t off = (pk->pin_off + 63) & ~(size_t)63;
  if (off + bytes > pk->pin_cap) {  // wrap: every stream that may still be reading the staging area must be done with it
    ZK_TRY(zk_sync_all(ctx));
    off = 0;
  }
  memcpy(pk->pin + off, h, bytes);
  pk->pin_off = off + bytes;
  ZK_HIP(ctx, hipMemcpyAsync(d, pk->pin + off, bytes, hipMemcpyHostToDevice, ctx->stream));
  return AMDZK_OK;
}
int d2h(amdzk_ctx* ctx, void* h, const void* d, size_t bytes) {
  if (bytes) {
    ZK_HIP(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  }
  return AMDZK_OK;
}
int d2d(amdzk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes) ZK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return AMDZK_OK;
}

// lookup::prover::permute_expression_pair for L lookups at once, on Montgomery-form columns of n rows each:
// A (compressed inputs, [L][n]) is sorted in place into A', S ([L][n]) receives the aligned table S'; rows >= usable
// of both come back zero (the caller blinds them). T: the compressed tables (read only). Ts / left: [L][n] scratch,
// flags: 4 x L x (n + 8) u32 scratch. Canonical keys (numeric order = upstream's Ord for Fr), rows >= usable padded
// with an all-ones sentinel (> any canonical value) so the power-of-two sort leaves the real rows in front.
// Ts_sorted: the first `presorted` tables as keygen left them (canonical, padded, sorted), or null. Enqueues the whole
// permutation on ctx's stream; d_err receives 1 + the index of a lookup whose input is not in its table (0: none) —
// zk_permute_check reads it.
int zk_permute_expression_pairs(amdzk_ctx* ctx, Fr* A, const Fr* T, Fr* Ts, Fr* S, Fr* left, uint32_t* flags, int* d_err, size_t L, uint32_t n,
                                uint32_t usable, const Fr* Ts_sorted = nullptr, size_t presorted = 0) {
  if (L == 0) return AMDZK_OK;
  if (presorted > L || (presorted && !Ts_sorted)) ZK_FAIL(ctx, AMDZK_E_INVALID, "permute_expression_pairs: bad presorted tables");
  const size_t rest = L - presorted;
  Fr* Tr = Ts + presorted * n;
  if (presorted) ZK_TRY(d2d(ctx, Ts, Ts_sorted, presorted * n * 32));
  ZK_TRY(amdzk_fr_to_repr_dev(ctx, A, L * n));
  ZK_HIP(ctx, hipMemset2DAsync(A + usable, (size_t)n * 32, 0xFF, (size_t)(n - usable) * 32, L, ctx->stream));
  if (rest) {
    ZK_TRY(d2d(ctx, Tr, T + presorted * n, rest * n * 32));
    ZK_TRY(amdzk_fr_to_repr_dev(ctx, Tr, rest * n));
    ZK_HIP(ctx, hipMemset2DAsync(Tr + usable, (size_t)n * 32, 0xFF, (size_t)(n - usable) * 32, rest, ctx->stream));
  }
  ZK_HIP(ctx, hipMemsetAsync(d_err, 0, sizeof(int), ctx->stream));
  ZK_TRY(zk_lookup_permute(ctx, A, Ts, S, left, L, n, usable, flags, (size_t)n + 8, d_err, presorted));
  ZK_HIP(ctx, hipMemset2DAsync(A + usable, (size_t)n * 32, 0, (size_t)(n - usable) * 32, L, ctx->stream));
  ZK_HIP(ctx, hipMemset2DAsync(S + usable, (size_t)n * 32, 0, (size_t)(n - usable) * 32, L, ctx->stream));
  ZK_TRY(amdzk_fr_from_raw_dev(ctx, A, L * n));
  ZK_TRY(amdzk_fr_from_raw_dev(ctx, S, L * n));
  return AMDZK_OK;
}

// the word zk_permute_expression_pairs left in d_err, once its work on ctx's stream is done
int zk_permute_check(amdzk_ctx* ctx, const int* d_err) {
  int herr = 0;
  ZK_TRY(d2h(ctx, &herr, d_err, sizeof(int)));
  if (herr) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: lookup %d input not in table (ConstraintSystemFailure)", herr - 1);
  return AMDZK_OK;
}

// MSM of ncols resident columns -> affine points on the host, in two halves: commit_launch enqueues the kernels on
// `ctx`'s stream (a lane or the caller's ctx) and returns; commit_finish copies the result down and waits for that
// stream only. Between the two the host enqueues work on other lanes. One commitment batch per lane at a time (the
// result sits in the lane's MSM workspace until it is finished).
struct PendingCommit {
  amdzk_ctx* ctx = nullptr;
  G1X* d_res = nullptr;
  size_t ncols = 0;
};
int commit_launch(amdzk_ctx* ctx, amdzk_pk* pk, int basis, const Fr* d_cols, size_t ncols, PendingCommit& pc) {
  pc.ctx = ctx;
  pc.ncols = ncols;
  pc.d_res = nullptr;
  if (ncols == 0) return AMDZK_OK;
  return zk_msm_dev_xyzz(ctx, pk->srs, basis, d_cols, ncols, pk->n, pk->n, &pc.d_res);
}
int commit_finish(PendingCommit& pc, std::vector<G1Affine>& out) {
  amdzk_ctx* ctx = pc.ctx;
  const size_t ncols = pc.ncols;
  out.resize(ncols);
  if (ncols == 0) return AMDZK_OK;
  std::vector<uint64_t> jac(12 * ncols);
  ZK_TRY(zk_msm_finish(ctx, pc.d_res, ncols, jac.data()));
  for (size_t i = 0; i < ncols; i++) {
    const G1Jac* j = reinterpret_cast<const G1Jac*>(&jac[12 * i]);
    if (j->z.is_zero()) {
      out[i].x = Fq::zero();
      out[i].y = Fq::zero();
    } else {
      out[i].x = j->x;
      out[i].y = j->y;
    }
  }
  return AMDZK_OK;
}
int commit_cols(amdzk_ctx* ctx, amdzk_pk* pk, int basis, const Fr* d_cols, size_t ncols, std::vector<G1Affine>& out) {
  PendingCommit pc;
  ZK_TRY(commit_launch(ctx, pk, basis, d_cols, ncols, pc));
  return commit_finish(pc, out);
}

// Translate a host-format postfix expression into device ops. The postfix words are first rebuilt
// into a tree so that a binary operation with a leaf operand (a column or a constant) becomes ONE
// fused instruction on the top of stack (MUL_COL / ADD_COL / SUB_COL / MUL_CONST / ADD_CONST) instead
// of push + pop through the LDS stack. Field addition and multiplication are exact and commutative,
// so the value is the one upstream's Expression::evaluate produces. Lagrange and extended programs
// share slot numbers for fixed/advice/instance columns.
struct ENode {
  uint32_t op, payload;
  int l, r;
};

int emit_tree(amdzk_ctx* ctx, amdzk_pk* pk, Program& pr, const std::vector<ENode>& t, int i) {
  const ENode& n = t[i];
  auto is_col = [&](int j) { return t[j].op == XOP_FIXED || t[j].op == XOP_ADVICE || t[j].op == XOP_INSTANCE; };
  auto col_arg = [&](int j) -> uint32_t {
    const ENode& c = t[j];
    uint32_t col = c.payload >> 8;
    int32_t rot = (int32_t)(c.payload & 0xff) - 128;
    uint32_t slot = c.op == XOP_FIXED ? pk->sl_fixed(col) : c.op == XOP_ADVICE ? pk->sl_adv(col) : pk->sl_inst(col);
    return (slot << 8) | pk->rots.index(rot);
  };
  switch (n.op) {
    case XOP_CONST:
      pr.op(OP_PUSH_CONST, n.payload);
      pr.push();
      return AMDZK_OK;
    case XOP_FIXED:
    case XOP_ADVICE:
    case XOP_INSTANCE:
      pr.op(OP_PUSH_COL, col_arg(i));
      pr.push();
      return AMDZK_OK;
    case XOP_NEG:
      ZK_TRY(emit_tree(ctx, pk, pr, t, n.l));
      pr.op(OP_NEG);
      return AMDZK_OK;
    case XOP_SCALE:
      ZK_TRY(emit_tree(ctx, pk, pr, t, n.l));
      pr.op(OP_MUL_CONST, n.payload);
      return AMDZK_OK;
    case XOP_ADD: {
      int a = n.l, b = n.r;
      if (t[b].op == XOP_NEG && is_col(t[b].l)) {  // a + (-col) -> a - col
        ZK_TRY(emit_tree(ctx, pk, pr, t, a));
        pr.op(OP_SUB_COL, col_arg(t[b].l));
        return AMDZK_OK;
      }
      if (!is_col(b) && t[b].op != XOP_CONST && (is_col(a) || t[a].op == XOP_CONST)) std::swap(a, b);
      if (is_col(b)) {
        ZK_TRY(emit_tree(ctx, pk, pr, t, a));
        pr.op(OP_ADD_COL, col_arg(b));
        return AMDZK_OK;
      }
      if (t[b].op == XOP_CONST) {
        ZK_TRY(emit_tree(ctx, pk, pr, t, a));
        pr.op(OP_ADD_CONST, t[b].payload);
        return AMDZK_OK;
      }
      ZK_TRY(emit_tree(ctx, pk, pr, t, a));
      ZK_TRY(emit_tree(ctx, pk, pr, t, b));
      pr.op(OP_ADD);
      pr.pop();
      return AMDZK_OK;
    }
    case XOP_MUL: {
      int a = n.l, b = n.r;
      if (!is_col(b) && t[b].op != XOP_CONST && (is_col(a) || t[a].op == XOP_CONST)) std::swap(a, b);
      if (is_col(b)) {
        ZK_TRY(emit_tree(ctx, pk, pr, t, a));
        pr.op(OP_MUL_COL, col_arg(b));
        return AMDZK_OK;
      }
      if (t[b].op == XOP_CONST) {
        ZK_TRY(emit_tree(ctx, pk, pr, t, a));
        pr.op(OP_MUL_CONST, t[b].payload);
        return AMDZK_OK;
      }
      ZK_TRY(emit_tree(ctx, pk, pr, t, a));
      ZK_TRY(emit_tree(ctx, pk, pr, t, b));
      pr.op(OP_MUL);
      pr.pop();
      return AMDZK_OK;
    }
    default:
      ZK_FAIL(ctx, AMDZK_E_INVALID, "circuit: bad expression node %u", n.op);
  }
}

int emit_expr(amdzk_ctx* ctx, amdzk_pk* pk, Program& pr, const std::vector<uint32_t>& words) {
  std::vector<ENode> t;
  std::vector<int> st;
  for (uint32_t w : words) {
    uint32_t op = w >> 24, pl = w & 0xffffffu;
    switch (op) {
      case XOP_CONST:
        if (pl >= pk->c_one) ZK_FAIL(ctx, AMDZK_E_INVALID, "circuit: constant index %u out of range", pl);
        t.push_back(ENode{op, pl, -1, -1});
        st.push_back((int)t.size() - 1);
        break;
      case XOP_FIXED:
      case XOP_ADVICE:
      case XOP_INSTANCE: {
        uint32_t col = pl >> 8;
        uint32_t lim = op == XOP_FIXED ? pk->F : op == XOP_ADVICE ? pk->A : pk->I;
        if (col >= lim) ZK_FAIL(ctx, AMDZK_E_INVALID, "circuit: column %u out of range", col);
        t.push_back(ENode{op, pl, -1, -1});
        st.push_back((int)t.size() - 1);
      } break;
      case XOP_NEG:
      case XOP_SCALE:
        if (st.empty()) ZK_FAIL(ctx, AMDZK_E_INVALID, "circuit: malformed expression");
        if (op == XOP_SCALE && pl >= pk->c_one) ZK_FAIL(ctx, AMDZK_E_INVALID, "circuit: constant index %u out of range", pl);
        t.push_back(ENode{op, pl, st.back(), -1});
        st.back() = (int)t.size() - 1;
        break;
      case XOP_ADD:
      case XOP_MUL: {
        if (st.size() < 2) ZK_FAIL(ctx, AMDZK_E_INVALID, "circuit: malformed expression");
        int r = st.back();
        st.pop_back();
        int l = st.back();
        t.push_back(ENode{op, 0, l, r});
        st.back() = (int)t.size() - 1;
      } break;
      default:
        ZK_FAIL(ctx, AMDZK_E_INVALID, "circuit: bad expression word %08x", w);
    }
  }
  if (st.size() != 1) ZK_FAIL(ctx, AMDZK_E_INVALID, "circuit: malformed expression");
  return emit_tree(ctx, pk, pr, t, st[0]);
}

// fold(acc * theta + expr) over a lookup's expressions (first term: 0*theta + e0 = e0)
int emit_compressed(amdzk_ctx* ctx, amdzk_pk* pk, Program& pr, uint32_t first, uint32_t count) {
  for (uint32_t i = 0; i < count; i++) {
    if (i > 0) pr.op(OP_MUL_CONST, pk->c_theta);
    ZK_TRY(emit_expr(ctx, pk, pr, pk->exprs[first + i]));
    if (i > 0) {
      pr.op(OP_ADD);
      pr.pop();
    }
  }
  return AMDZK_OK;
}

// Quotient-domain programs run on the limb-resident interpreter (plonk_kernels.hip expr_eval_limbs_kernel): values are
// 9 x 29-bit limbs, lazily reduced. This pass walks the (straight-line, wave-uniform) program once with the bound of
// every stack entry in units of p and
//   * inserts OP_REDUCE where a product would leave f29_mul's range (a*b < 169 p^2), where a subtrahend is too large
//     for the K*p constants (below 2p: K = 3, below 9p: K = 10), before a value sinks into the LDS stack with a bound
//     above 8, and before OP_STORE (packing needs a value below 2p);
//   * picks OP_SUB / OP_SUB_BIG and OP_NEG / OP_NEG_BIG by the subtrahend's bound;
//   * replaces the Horner fold h = h*y + term (OP_ACC) by a sum of products with one reduction per group: the program
//     is cut into its terms (the stack is empty at every OP_ACC), a term that ends in `* hot[k]` loses that factor
//     and joins group k, the others group 4; term j of the original order adds term_j * y^(K-1-j) to the wide
//     accumulator (OP_WACC j; every sixth term of a group also moves the columns' carries up), and each group ends
//     with OP_WFLUSH k: h (+)= reduce(wide) * hot[k], h canonical in its output row. A group is split when its sum of
//     bounds would leave the reduction's range.
// Bounds: a column, constant or hot value is below 1 (canonical); a product is below 2; a sum adds the bounds; a
// difference a - b adds K to a's; the weak reduction gives 1.0002; a flushed group sum(bounds) / 169.3 + 1.
// nparts > 1 cuts the finalised program into that many independent pieces of about equal length (Program::piece_starts):
// a piece is a run of terms of the group order, closed by its own flush, and its first flush overwrites ITS h (bit 4) —
// the interpreter runs piece p on the workgroups with blockIdx.y = p into h + p * rows, and the pieces' sums are added
// afterwards (h is linear in the terms). One proof alone fills the chip's wavefront slots only that way.
// Returns the number of terms (= the powers of y of amdzk_pk::d_ypow that the OP_WACC instructions point at).
constexpr uint32_t H_PARTS_MAX = 8;
uint32_t finalize_limb_program(Program& pr, uint32_t nparts = 1) {
  const double LIM = 160.0, RED = 1.01, GROUP_LIM = 169.0 * 30.0;  // a flushed group stays below ~31 p (+ h, canonical)
  struct Term {
    std::vector<uint32_t> words;
    double bound = 0;
    uint32_t index = 0, group = 4;
  };
  std::vector<Term> terms;
  std::vector<uint32_t> tail;  // programs without OP_ACC (OP_STORE only) keep their order
  std::vector<uint32_t> out;
  std::vector<double> st;  // bounds, st.back() = top of stack
  auto emit = [&](uint32_t op, uint32_t arg = 0) { out.push_back((op << 24) | (arg & 0xffffffu)); };
  auto reduce_tos = [&]() {
    emit(OP_REDUCE);
    st.back() = RED;
  };
  uint32_t depth = 0, nterms = 0;
  for (size_t wi = 0; wi < pr.words.size(); wi++) {
    const uint32_t w = pr.words[wi], op = w >> 24, arg = w & 0xffffffu;
    switch (op) {
      case OP_PUSH_COL:
      case OP_PUSH_CONST:
      case OP_PUSH_HOT:
        if (!st.empty() && st.back() > 8.0) reduce_tos();
        emit(op, arg);
        st.push_back(1.0);
        break;
      case OP_MUL_HOT:
        // the closing `* hot[k]` of a term is factored out of its group instead of being multiplied in
        if (st.size() == 1 && wi + 1 < pr.words.size() && (pr.words[wi + 1] >> 24) == OP_ACC) {
          Term t;
          t.group = arg;
          t.bound = st.back();
          t.index = nterms++;
          t.words.swap(out);
          terms.push_back(std::move(t));
          st.clear();
          wi++;  // the OP_ACC is consumed
          break;
        }
        [[fallthrough]];
      case OP_MUL_COL:
      case OP_MUL_CONST:
        if (st.back() >= LIM) reduce_tos();
        emit(op, arg);
        st.back() = 2.0;
        break;
      case OP_ADD_COL:
      case OP_ADD_CONST:
        if (st.back() + 1.0 > 40.0) reduce_tos();
        emit(op, arg);
        st.back() += 1.0;
        break;
      case OP_SUB_COL:
        if (st.back() + 3.0 > 40.0) reduce_tos();
        emit(op, arg);
        st.back() += 3.0;
        break;
      case OP_ADD: {
        if (st[st.size() - 2] + st.back() > 40.0) reduce_tos();
        const double b = st.back();
        st.pop_back();
        emit(op);
        st.back() += b;
      } break;
      case OP_SUB: {
        if (st.back() >= 9.0) reduce_tos();
        const double b = st.back();
        st.pop_back();
        emit(b < 2.0 ? OP_SUB : OP_SUB_BIG);
        st.back() += b < 2.0 ? 3.0 : 10.0;
      } break;
      case OP_MUL: {
        if (st[st.size() - 2] * st.back() >= LIM) reduce_tos();
        st.pop_back();
        emit(op);
        st.back() = 2.0;
      } break;
      case OP_NEG:
        if (st.back() >= 9.0) reduce_tos();
        emit(st.back() < 2.0 ? OP_NEG : OP_NEG_BIG);
        st.back() = st.back() < 2.0 ? 3.0 : 10.0;
        break;
      case OP_SQR:
        if (st.back() * st.back() >= LIM) reduce_tos();
        emit(op);
        st.back() = 2.0;
        break;
      case OP_ACC: {  // end of a term without a hot factor
        Term t;
        t.group = 4;
        t.bound = st.back();
        t.index = nterms++;
        t.words.swap(out);
        terms.push_back(std::move(t));
        st.clear();
      } break;
      case OP_STORE:
        if (st.back() >= 2.0) reduce_tos();
        emit(op, arg);
        st.pop_back();
        tail.insert(tail.end(), out.begin(), out.end());
        out.clear();
        break;
      case OP_PICK:  // a copy of the entry `arg` below the top; the top sinks into the LDS stack
        if (st.back() > 8.0) reduce_tos();
        emit(op, arg);
        st.push_back(st[st.size() - 1 - arg]);
        break;
      case OP_NIP:
        emit(op, arg);
        st.erase(st.end() - 1 - arg, st.end() - 1);
        break;
      default:
        emit(op, arg);
        break;
    }
    if (st.size() > depth) depth = (uint32_t)st.size();
  }
  tail.insert(tail.end(), out.begin(), out.end());
  std::vector<uint32_t> fin;
  bool first = true;
  auto flush = [&](uint32_t g) {
    fin.push_back((OP_WFLUSH << 24) | g | (first ? 16u : 0u));
    first = false;
  };
  size_t term_words = 0;
  for (const Term& t : terms) term_words += t.words.size() + 1;
  if (terms.empty() || !tail.empty()) nparts = 1;  // (programs that store columns are not cut)
  pr.piece_starts.clear();
  uint32_t part = 0;
  size_t part_begin = 0;
  if (nparts > 1) pr.piece_starts.push_back(0);
  for (uint32_t g = 0; g <= 4; g++) {
    double sum = 0;
    uint32_t since_carry = 0;
    bool open = false;
    for (const Term& t : terms) {
      if (t.group != g) continue;
      if (open && sum + t.bound > GROUP_LIM) {
        flush(g);
        sum = 0;
        since_carry = 0;
        open = false;
      }
      // the next piece starts where this one has its share of the instructions
      if (part + 1 < nparts && fin.size() - part_begin >= (term_words + nparts - 1) / nparts) {
        if (open) flush(g);
        sum = 0;
        since_carry = 0;
        open = false;
        part++;
        part_begin = fin.size();
        pr.piece_starts.push_back((uint32_t)fin.size());
        first = true;
      }
      fin.insert(fin.end(), t.words.begin(), t.words.end());
      const bool carry = ++since_carry == 6;  // a column holds six un-carried terms
      if (carry) since_carry = 0;
      fin.push_back((OP_WACC << 24) | (carry ? 1u << 23 : 0u) | t.index);
      sum += t.bound;
      open = true;
    }
    if (open) flush(g);
  }
  fin.insert(fin.end(), tail.begin(), tail.end());
  pr.words.swap(fin);
  pr.depth = depth + 1;
  return nterms;
}

// d_consts261[i] = 32 * consts[i] in the ordinary form, i.e. consts[i] in radix 2^261 (a few hundred values).
int upload_consts261(amdzk_ctx* ctx, amdzk_pk* pk) {
  Fr k32 = Fr::one();
  for (int i = 0; i < 5; i++) k32 = add(k32, k32);
  std::vector<Fr> c(pk->consts.size());
  for (size_t i = 0; i < c.size(); i++) c[i] = mul(pk->consts[i], k32);
  ZK_TRY(h2d_staged(ctx, pk, pk->d_consts261, c.data(), c.size() * 32));
  // without the pinned staging area the copy above reads `c` asynchronously: finish it before `c` goes away
  if (!pk->pin || c.size() * 32 > pk->pin_cap) ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  return AMDZK_OK;
}

// d_ypow[j] = y^(K-1-j) for the K terms of the h(X) program, radix 2^261 (upstream folds the constraint values with
// Horner, h = h*y + value, in the same order: term j carries y^(K-1-j)).
int upload_ypow(amdzk_ctx* ctx, amdzk_pk* pk) {
  const uint32_t K = pk->h_terms;
  if (!K) return AMDZK_OK;
  Fr k32 = Fr::one();
  for (int i = 0; i < 5; i++) k32 = add(k32, k32);
  std::vector<Fr> pw(K);
  Fr cur = k32;
  const Fr y = pk->consts[pk->c_y], beta = pk->consts[pk->c_beta];
  std::vector<Fr> bpow = {Fr::one()};  // beta^m for the terms whose factor beta^m was taken out (the permutation products)
  for (uint32_t j = K; j-- > 0;) {
    const uint32_t m = j < pk->h_term_beta_pow.size() ? pk->h_term_beta_pow[j] : 0;
    while (bpow.size() <= m) bpow.push_back(mul(bpow.back(), beta));
    pw[j] = m ? mul(cur, bpow[m]) : cur;
    cur = mul(cur, y);
  }
  ZK_TRY(h2d_staged(ctx, pk, pk->d_ypow, pw.data(), pw.size() * 32));
  if (!pk->pin || pw.size() * 32 > pk->pin_cap) ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  return AMDZK_OK;
}

// Resolve slots / rotation indices / constant indices into addresses and row offsets for one domain
// and upload the 16-byte instructions.
int upload_program(amdzk_ctx* ctx, amdzk_pk* pk, Program& pr, bool extended) {
  const std::vector<const Fr*>& cols = extended ? pk->h_cols_ext : pk->h_cols_lag;
  // The interpreters fetch every instruction's operand, and the instruction two ahead, unconditionally: an
  // instruction without an operand names the first constant, and two END instructions close the program.
  const Fr* dummy = extended ? pk->d_consts261 : pk->d_consts;
  std::vector<ExprInstr> ins(pr.words.size() + 2);
  for (size_t i = 0; i < ins.size(); i++) {
    const uint32_t w = i < pr.words.size() ? pr.words[i] : (uint32_t)OP_END << 24, op = w >> 24, arg = w & 0xffffffu;
    ins[i].op_arg = w;
    ins[i].rot = 0;
    ins[i].ptr = dummy;
    if (op == OP_PUSH_COL || op == OP_MUL_COL || op == OP_ADD_COL || op == OP_SUB_COL) {
      if ((arg >> 8) >= cols.size() || (arg & 0xff) >= pk->rots.rots.size()) ZK_FAIL(ctx, AMDZK_E_INVALID, "program: bad column operand");
      ins[i].ptr = cols[arg >> 8];
      ins[i].rot = pk->rots.rots[arg & 0xff];  // rows of one coset are consecutive: a rotation is a row offset in both domains
    } else if (op == OP_PUSH_CONST || op == OP_MUL_CONST || op == OP_ADD_CONST) {
      if (arg >= pk->consts.size()) ZK_FAIL(ctx, AMDZK_E_INVALID, "program: bad constant operand");
      ins[i].ptr = (extended ? pk->d_consts261 : pk->d_consts) + arg;
    } else if (op == OP_WACC) {
      if (!extended || (arg & 0x7fffffu) >= pk->h_terms) ZK_FAIL(ctx, AMDZK_E_INVALID, "program: bad power of y");
      ins[i].ptr = pk->d_ypow + (arg & 0x7fffffu);
    }
  }
  ZK_TRY(dalloc(ctx, pk, &pr.d_instr, ins.size()));
  ZK_TRY(h2d(ctx, pr.d_instr, ins.data(), ins.size() * sizeof(ExprInstr)));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));  // `ins` is a host temporary
  return AMDZK_OK;
}

int run_program(amdzk_ctx* ctx, amdzk_pk* pk, Program& pr, bool extended, Fr* const* d_outs, Fr* h_out, const char* name) {
  ExprArgs a;
  a.prog = pr.d_instr;
  a.prog_len = (uint32_t)pr.words.size();
  a.cols = extended ? pk->d_cols_ext : pk->d_cols_lag;
  a.outs = d_outs;
  a.h_out = h_out;
  a.nrows = extended ? pk->ext : pk->n;
  a.mask = pk->n - 1;
  // Quotient-domain programs (h(X), l_active) run in radix 2^261: their columns come from
  // zk_coeff_to_cosets_r261, their constants from d_consts261, and the result goes back through
  // zk_cosets_to_pieces. Lagrange-domain programs read the caller's radix-2^256 witness as is.
  a.radix261 = extended ? 1u : 0u;
  a.nparts = 0;
  if (!extended && pr.piece_starts.size() > 1) {  // balanced by instruction count, cut at piece boundaries only
    const uint32_t total = (uint32_t)pr.words.size(), want = std::min<uint32_t>(EXPR_MAX_PARTS, (uint32_t)pr.piece_starts.size());
    uint32_t begin = 0;
    for (size_t i = 1; i <= pr.piece_starts.size() && a.nparts < want; i++) {
      const uint32_t end = i < pr.piece_starts.size() ? pr.piece_starts[i] : total;
      const bool last_part = a.nparts + 1 == want;
      if ((!last_part && end >= (uint64_t)total * (a.nparts + 1) / want) || (last_part && end == total)) {
        a.part_start[a.nparts] = begin;
        a.part_len[a.nparts] = end - begin;
        a.nparts++;
        begin = end;
      }
    }
    if (begin != total) a.nparts = 0;  // (cannot happen: the last part runs to the end) — fall back to one part
  }
  if (extended && pr.piece_starts.size() > 1) {  // the pieces finalize_limb_program cut: one per blockIdx.y, h_out + p * rows each
    if (pr.piece_starts.size() > (size_t)EXPR_MAX_PARTS) ZK_FAIL(ctx, AMDZK_E_INVALID, "program: too many pieces");
    for (size_t i = 0; i < pr.piece_starts.size(); i++) {
      a.part_start[i] = pr.piece_starts[i];
      a.part_len[i] = (i + 1 < pr.piece_starts.size() ? pr.piece_starts[i + 1] : (uint32_t)pr.words.size()) - pr.piece_starts[i];
    }
    a.nparts = (uint32_t)pr.piece_starts.size();
  }
  for (int i = 0; i < EXPR_HOT; i++) a.hot[i] = EXPR_NO_SLOT;
  if (extended && pr.uses_hot) {
    a.hot[0] = pk->se_l0();
    a.hot[1] = pk->se_llast();
    a.hot[2] = pk->se_lactive();
    a.hot[3] = pk->se_x();
  }
  // LDS stack slots: the limb interpreter keeps the top of the stack in registers, so a program whose stack holds at most
  // pr.depth - 1 values (finalize_limb_program) needs pr.depth - 2 slots: pr.depth - 1 leaves one spare
  return extended ? zk_expr_eval_limbs(ctx, a, pr.depth > 1 ? pr.depth - 1 : 1, name) : zk_expr_eval(ctx, a, pr.depth + 1, name);
}

Fr rotate_omega(const amdzk_pk* pk, const Fr& x, int rot) {
  return rot >= 0 ? mul(x, pow_u64(pk->omega, (uint64_t)rot)) : mul(x, pow_u64(pk->omega_inv, (uint64_t)(-rot)));
}

Fr eval_small(const std::vector<Fr>& poly, const Fr& x) {
  Fr acc = Fr::zero();
  for (size_t i = poly.size(); i-- > 0;) acc = add(mul(acc, x), poly[i]);
  return acc;
}

void trace_fr(const char* label, const Fr& v) {
  if (!getenv("AMDZK_TRACE")) return;
  uint8_t b[32];
  zkhost::fr_to_repr(v, b);
  fprintf(stderr, "[amdzk] %s ", label);
  for (int i = 31; i >= 0; i--) fprintf(stderr, "%02x", b[i]);
  fprintf(stderr, "\n");
}
void trace_pt(const char* label, const G1Affine& p) {
  if (!getenv("AMDZK_TRACE")) return;
  uint8_t b[32];
  zkhost::fq_to_repr(p.x, b);
  fprintf(stderr, "[amdzk] %s x=", label);
  for (int i = 31; i >= 0; i--) fprintf(stderr, "%02x", b[i]);
  fprintf(stderr, "\n");
}

}  // namespace

// plonk::evaluation::Evaluator::evaluate_h + divide_by_vanishing_poly + extended_to_coeff + the split into pieces
// (SURVEY.md §8(a) rows a6, a7, a10): from the committed polynomials in coefficient form (pk->PQ, arena order) and the
// challenges in pk->consts to the degree-1 pieces of h(X) in pk->hpieces. The numerator is evaluated on nc cosets of
// the size-n subgroup (poly.hip, zk_quotient_plan), then divided by X^n - 1, interpolated per coset and recombined.
static int quotient_from_cosets(amdzk_ctx* ctx, amdzk_pk* pk) {
  ZK_TRY(upload_consts261(ctx, pk));
  ZK_TRY(upload_ypow(ctx, pk));
  // the program writes h where its first group of terms is flushed: a constraint system without a single term has none
  if (!pk->h_terms) ZK_HIP(ctx, hipMemsetAsync(pk->hq, 0, (size_t)pk->ext * 32, ctx->stream));
  ZK_TRY(run_program(ctx, pk, pk->prog_h, true, nullptr, pk->hq, "expr_evaluate_h"));
  ZK_TRY(zk_cosets_to_pieces(ctx, pk->dom, pk->hq, pk->hpieces, pk->qdeg));
  return AMDZK_OK;
}
static int quotient_pieces(amdzk_ctx* ctx, amdzk_pk* pk) {
  ZK_TRY(zk_coeff_to_cosets_r261(ctx, pk->dom, pk->PQ, pk->n, pk->PC, pk->ext, pk->NP));
  return quotient_from_cosets(ctx, pk);
}

// The per-proof workspace of ONE circuit instance (arenas, lookup / product scratch, multiopen buffers, small staging):
// what a key owns besides its key material, and all a workspace clone allocates.
static int alloc_proof_workspace(amdzk_ctx* ctx, amdzk_pk* pk) {
  const size_t n = pk->n, ext = pk->ext;
  const uint32_t A = pk->A, I = pk->I, L = pk->L, ns = pk->nsets;
#define KG_TRY(x) ZK_TRY(x)
  pk->NP = (size_t)A + I + 2 * L + ns + L;
  KG_TRY(dalloc(ctx, pk, &pk->P, pk->NP * n));
  KG_TRY(dalloc(ctx, pk, &pk->PQ, pk->NP * n));
  KG_TRY(dalloc(ctx, pk, &pk->PC, pk->NP * ext));
  KG_TRY(dalloc(ctx, pk, &pk->ci, (size_t)L * n));
  KG_TRY(dalloc(ctx, pk, &pk->ct, (size_t)L * n));
  KG_TRY(dalloc(ctx, pk, &pk->lk_ts, (size_t)L * n));
  KG_TRY(dalloc(ctx, pk, &pk->lk_left, (size_t)L * n));
  KG_TRY(dalloc(ctx, pk, &pk->lk_flags, (size_t)4 * L * (n + 8)));
  KG_TRY(dalloc(ctx, pk, &pk->d_err, 1));
  KG_TRY(dalloc(ctx, pk, &pk->rnd, n));
  KG_TRY(dalloc(ctx, pk, &pk->hq, (size_t)H_PARTS_MAX * ext));  // one h per piece of the cut h(X) program (finalize_limb_program)
  KG_TRY(dalloc(ctx, pk, &pk->hpieces, (size_t)pk->qdeg * n));
  KG_TRY(dalloc(ctx, pk, &pk->hpoly, n));
  const size_t nfrac = std::max<size_t>(std::max<size_t>(ns, L), 1);
  KG_TRY(dalloc(ctx, pk, &pk->frac, nfrac * n));
  KG_TRY(dalloc(ctx, pk, &pk->scratch, std::max(nfrac * n, ext)));
  KG_TRY(dalloc(ctx, pk, &pk->scan_tmp, zk_scan_totals_elems(n, nfrac) + 2 * nfrac + 8));
  KG_TRY(dalloc(ctx, pk, &pk->frac2, std::max<size_t>(L, 1) * n));
  KG_TRY(dalloc(ctx, pk, &pk->scratch2, std::max<size_t>(L, 1) * n));
  KG_TRY(dalloc(ctx, pk, &pk->scan_tmp2, zk_scan_totals_elems(n, std::max<size_t>(L, 1)) + 2 * std::max<size_t>(L, 1) + 8));
  const size_t max_rsets = pk->max_sets;
  KG_TRY(dalloc(ctx, pk, &pk->sets_L, max_rsets * n));
  KG_TRY(dalloc(ctx, pk, &pk->sets_N, max_rsets * n));
  KG_TRY(dalloc(ctx, pk, &pk->hx, n));
  pk->small_cap = std::max<size_t>((size_t)pk->NP * (pk->bf + 2) + 4096, 8192);
  for (int l = 0; l < 3; l++) KG_TRY(dalloc(ctx, pk, &pk->small_l[l], pk->small_cap));
  pk->small = pk->small_l[0];
  pk->pin_cap = std::max<size_t>((size_t)8 << 20, 2 * n * 32);
  if (hipHostMalloc((void**)&pk->pin, pk->pin_cap + 64, hipHostMallocDefault) != hipSuccess) {
    pk->pin = nullptr;
    pk->pin_cap = 0;
    ZK_FAIL(ctx, AMDZK_E_NOMEM, "prover: hipHostMalloc of the pinned staging area failed");
  }
  pk->h_err = (int*)(pk->pin + pk->pin_cap);  // behind the staging ring
  pk->ptrs_cap = 8192;
  for (int l = 0; l < 3; l++) {
    void** pp = nullptr;
    KG_TRY(dalloc(ctx, pk, &pp, pk->ptrs_cap));
    pk->ptrs_l[l] = pp;
  }
  pk->ptrs = pk->ptrs_l[0];

#undef KG_TRY
  return AMDZK_OK;
}

// The slot -> column pointer tables the interpreters read (Lagrange and quotient domain): key columns and this
// workspace's arenas.
static int build_column_tables(amdzk_ctx* ctx, amdzk_pk* pk) {
  const size_t n = pk->n, ext = pk->ext;
  const uint32_t F = pk->F, A = pk->A, I = pk->I, S = pk->S, L = pk->L, ns = pk->nsets;
#define KG_TRY(x) ZK_TRY(x)
  // ---- column pointer tables
  {
    std::vector<const Fr*> lag(pk->nslots_lag()), ex(pk->nslots_ext());
    for (uint32_t i = 0; i < F; i++) lag[pk->sl_fixed(i)] = pk->fixed_lag + (size_t)i * n, ex[i] = pk->fixed_coset + (size_t)i * ext;
    for (uint32_t i = 0; i < A; i++) lag[pk->sl_adv(i)] = pk->adv() + (size_t)i * n, ex[pk->sl_adv(i)] = pk->PC + (size_t)i * ext;
    for (uint32_t i = 0; i < I; i++) lag[pk->sl_inst(i)] = pk->inst() + (size_t)i * n, ex[pk->sl_inst(i)] = pk->PC + (size_t)(A + i) * ext;
    for (uint32_t i = 0; i < S; i++) lag[pk->sl_sigma(i)] = pk->sigma_lag + (size_t)i * n, ex[pk->se_sigma(i)] = pk->sigma_coset + (size_t)i * ext;
    for (uint32_t i = 0; i < S; i++) lag[pk->sl_dxw(i)] = pk->dxw_lag + (size_t)i * n, ex[pk->se_dx(i)] = pk->dx_coset + (size_t)i * ext;
    for (uint32_t l = 0; l < L; l++) {
      lag[pk->sl_ci(l)] = pk->ci + (size_t)l * n;
      lag[pk->sl_ct(l)] = pk->ct + (size_t)l * n;
      lag[pk->sl_la(l)] = pk->la() + (size_t)l * n;
      lag[pk->sl_ls(l)] = pk->ls() + (size_t)l * n;
      ex[pk->se_la(l)] = pk->PC + (size_t)(A + I + l) * ext;
      ex[pk->se_ls(l)] = pk->PC + (size_t)(A + I + L + l) * ext;
      ex[pk->se_zl(l)] = pk->PC + (size_t)(A + I + 2 * L + ns + l) * ext;
    }
    for (uint32_t s = 0; s < ns; s++) ex[pk->se_zp(s)] = pk->PC + (size_t)(A + I + 2 * L + s) * ext;
    lag[pk->sl_omega()] = pk->omega_pow;
    ex[pk->se_l0()] = pk->l0_c;
    ex[pk->se_llast()] = pk->llast_c;
    ex[pk->se_lactive()] = pk->lactive_c;
    ex[pk->se_x()] = pk->x_coset;
    pk->h_cols_lag = lag;
    pk->h_cols_ext = ex;
    KG_TRY(dalloc(ctx, pk, &pk->d_cols_lag, lag.size()));
    KG_TRY(dalloc(ctx, pk, &pk->d_cols_ext, ex.size()));
    KG_TRY(h2d(ctx, pk->d_cols_lag, lag.data(), lag.size() * sizeof(Fr*)));
    KG_TRY(h2d(ctx, pk->d_cols_ext, ex.data(), ex.size() * sizeof(Fr*)));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  }

#undef KG_TRY
  return AMDZK_OK;
}

// Where the Lagrange-domain programs store: compressed lookup inputs / tables, permutation fractions, lookup fractions.
static int build_output_tables(amdzk_ctx* ctx, amdzk_pk* pk) {
  const size_t n = pk->n;
  const uint32_t L = pk->L, ns = pk->nsets;
  {
    std::vector<Fr*> outs(2 * L);
    for (uint32_t l = 0; l < L; l++) outs[2 * l] = pk->ci + (size_t)l * n, outs[2 * l + 1] = pk->ct + (size_t)l * n;
    ZK_TRY(dalloc(ctx, pk, &pk->d_outs_compress, outs.size()));
    ZK_TRY(h2d(ctx, pk->d_outs_compress, outs.data(), outs.size() * sizeof(Fr*)));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  }
  {
    std::vector<Fr*> outs(2 * ns);
    for (uint32_t s = 0; s < ns; s++) outs[2 * s] = pk->frac + (size_t)s * n, outs[2 * s + 1] = pk->zp() + (size_t)s * n;
    ZK_TRY(dalloc(ctx, pk, &pk->d_outs_pfrac, outs.size()));
    ZK_TRY(h2d(ctx, pk->d_outs_pfrac, outs.data(), outs.size() * sizeof(Fr*)));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  }
  {
    std::vector<Fr*> outs(2 * L);
    for (uint32_t l = 0; l < L; l++) outs[2 * l] = pk->frac2 + (size_t)l * n, outs[2 * l + 1] = pk->zl() + (size_t)l * n;  // frac2: beside the permutation products
    ZK_TRY(dalloc(ctx, pk, &pk->d_outs_lfrac, outs.size()));
    ZK_TRY(h2d(ctx, pk->d_outs_lfrac, outs.data(), outs.size() * sizeof(Fr*)));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  }
  return AMDZK_OK;
}

extern "C" {

void amdzk_pk_free(amdzk_ctx* ctx, amdzk_pk* pk) {
  ZK_ENTER(ctx);
  if (!pk) return;
  if (ctx) zk_host_wait(ctx, ctx->stream);
  for (void* p : pk->allocs) hipFree(p);
  if (pk->pin) hipHostFree(pk->pin);
  if (pk->dom && !pk->clone_of) amdzk_domain_free(ctx, pk->dom);
  delete pk;
}

// Every device allocation of the key (columns, cosets, per-proof workspace) must live on ctx's device.
int amdzk_pk_check_affinity(amdzk_ctx* ctx, const amdzk_pk* pk) {
  ZK_ENTER(ctx);
  if (!ctx || !pk) return AMDZK_E_INVALID;
  for (void* p : pk->allocs) ZK_TRY(zk_ptr_on_device(ctx, p, "proving-key buffer"));
  return AMDZK_OK;
}

// keygen with the environment's defaults for the key's modes (AMDZK_FULL_COSETS, AMDZK_SERIAL); amdzk_keygen_ex takes
// them as explicit flags, so that two keys of one process can differ.
int amdzk_keygen(amdzk_ctx* ctx, const amdzk_srs* srs, const amdzk_circuit* c, const uint64_t* fixed_values, const uint32_t* perm_mapping,
                 const uint64_t transcript_repr[4], amdzk_pk** out) {
  uint32_t flags = 0;
  if (const char* e = getenv("AMDZK_FULL_COSETS")) flags |= atoi(e) != 0 || !*e ? AMDZK_KEYGEN_FULL_COSETS : 0u;
  if (const char* e = getenv("AMDZK_SERIAL")) flags |= atoi(e) != 0 ? AMDZK_KEYGEN_SERIAL : 0u;
  return amdzk_keygen_ex(ctx, srs, c, fixed_values, perm_mapping, transcript_repr, flags, out);
}

int amdzk_keygen_ex(amdzk_ctx* ctx, const amdzk_srs* srs, const amdzk_circuit* c, const uint64_t* fixed_values, const uint32_t* perm_mapping,
                    const uint64_t transcript_repr[4], uint32_t flags, amdzk_pk** out) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (flags & ~(uint32_t)(AMDZK_KEYGEN_FULL_COSETS | AMDZK_KEYGEN_SERIAL)) ZK_FAIL(ctx, AMDZK_E_INVALID, "keygen: unknown flags %#x", flags);
  if (!srs || !c || !out || !transcript_repr) ZK_FAIL(ctx, AMDZK_E_INVALID, "keygen: null argument");
  if (c->cs_degree < 3) ZK_FAIL(ctx, AMDZK_E_INVALID, "keygen: cs_degree %u < 3", c->cs_degree);
  amdzk_pk* pk = new amdzk_pk();
#define KG_TRY(x)                  \
  do {                             \
    int _r = (x);                  \
    if (_r != AMDZK_OK) {          \
      amdzk_pk_free(ctx, pk);      \
      return _r;                   \
    }                              \
  } while (0)
  pk->srs = srs;
  pk->k = c->k;
  pk->n = (size_t)1 << c->k;
  pk->bf = c->blinding_factors;
  pk->degree = c->cs_degree;
  pk->F = c->num_fixed;
  pk->A = c->num_advice;
  pk->I = c->num_instance;
  pk->S = c->num_perm_columns;
  pk->L = c->num_lookups;
  pk->chunk = pk->degree - 2;
  pk->nsets = (pk->S + pk->chunk - 1) / pk->chunk;
  pk->qdeg = pk->degree - 1;
  if (pk->n < pk->bf + 3) {
    amdzk_pk_free(ctx, pk);
    ZK_FAIL(ctx, AMDZK_E_INVALID, "keygen: not enough rows (n = %zu, blinding factors = %u)", pk->n, pk->bf);
  }
  memcpy(pk->transcript_repr.l, transcript_repr, 32);
  KG_TRY(amdzk_domain_new(ctx, pk->degree, pk->k, &pk->dom));
  pk->ek = amdzk_domain_extended_k(pk->dom);
  // h(X) has qdeg = degree - 1 pieces: that many cosets pin it down (AMDZK_FULL_COSETS=1: all 2^(ek-k), upstream's own
  // computation — identical output for satisfying witnesses, and the way to reproduce upstream's bytes for others)
  pk->nc = (flags & AMDZK_KEYGEN_FULL_COSETS) ? (1u << (pk->ek - pk->k)) : pk->qdeg;
  pk->use_lanes = !(flags & AMDZK_KEYGEN_SERIAL);
  KG_TRY(zk_quotient_plan(ctx, pk->dom, pk->nc));
  pk->ext = (size_t)pk->nc * pk->n;
  amdzk_domain_constant(pk->dom, 0, (uint64_t*)pk->omega.l);
  amdzk_domain_constant(pk->dom, 1, (uint64_t*)pk->omega_inv.l);
  for (uint32_t i = 0; i < c->num_advice_queries; i++) pk->advice_queries.push_back({c->advice_queries[2 * i], c->advice_queries[2 * i + 1]});
  for (uint32_t i = 0; i < c->num_fixed_queries; i++) pk->fixed_queries.push_back({c->fixed_queries[2 * i], c->fixed_queries[2 * i + 1]});
  for (uint32_t i = 0; i < c->num_instance_queries; i++)
    pk->instance_queries.push_back({c->instance_queries[2 * i], c->instance_queries[2 * i + 1]});
  for (uint32_t i = 0; i < pk->S; i++) pk->perm_cols.push_back({(int)c->perm_columns[2 * i], (int)c->perm_columns[2 * i + 1]});
  pk->num_gates = c->num_gates;
  uint32_t nexpr = c->num_gates;
  for (uint32_t l = 0; l < pk->L; l++) {
    pk->lookup_shape.push_back({c->lookup_shape[2 * l], c->lookup_shape[2 * l + 1]});
    nexpr += c->lookup_shape[2 * l] + c->lookup_shape[2 * l + 1];
  }
  if (nexpr != c->num_exprs) {
    amdzk_pk_free(ctx, pk);
    ZK_FAIL(ctx, AMDZK_E_INVALID, "keygen: expression count mismatch (%u vs %u)", nexpr, c->num_exprs);
  }
  for (uint32_t e = 0; e < c->num_exprs; e++)
    pk->exprs.emplace_back(c->expr_words + c->expr_offsets[e], c->expr_words + c->expr_offsets[e + 1]);
  // constants: circuit | one theta beta gamma y 1/beta
  pk->consts.resize(c->num_constants);
  if (c->num_constants) memcpy(pk->consts.data(), c->constants, (size_t)c->num_constants * 32);
  pk->c_one = c->num_constants;
  pk->c_theta = pk->c_one + 1;
  pk->c_beta = pk->c_one + 2;
  pk->c_gamma = pk->c_one + 3;
  pk->c_y = pk->c_one + 4;
  pk->c_betainv = pk->c_one + 5;
  pk->consts.resize(pk->c_betainv + 1, Fr::zero());
  pk->consts[pk->c_one] = Fr::one();

  const size_t n = pk->n, ext = pk->ext;
  const uint32_t F = pk->F, A = pk->A, I = pk->I, S = pk->S, L = pk->L, ns = pk->nsets;
  // ---- device allocations
  KG_TRY(dalloc(ctx, pk, &pk->fixed_lag, (size_t)F * n));
  KG_TRY(dalloc(ctx, pk, &pk->fixed_poly, (size_t)F * n));
  KG_TRY(dalloc(ctx, pk, &pk->fixed_coset, (size_t)F * ext));
  KG_TRY(dalloc(ctx, pk, &pk->sigma_lag, (size_t)S * n));
  KG_TRY(dalloc(ctx, pk, &pk->sigma_poly, (size_t)S * n));
  KG_TRY(dalloc(ctx, pk, &pk->sigma_coset, (size_t)S * ext));
  KG_TRY(dalloc(ctx, pk, &pk->l0_c, ext));
  KG_TRY(dalloc(ctx, pk, &pk->llast_c, ext));
  KG_TRY(dalloc(ctx, pk, &pk->lactive_c, ext));
  KG_TRY(dalloc(ctx, pk, &pk->x_coset, ext));
  KG_TRY(dalloc(ctx, pk, &pk->omega_pow, n));
  KG_TRY(dalloc(ctx, pk, &pk->dxw_lag, (size_t)S * n));
  KG_TRY(dalloc(ctx, pk, &pk->dx_coset, (size_t)S * ext));
  KG_TRY(alloc_proof_workspace(ctx, pk));

  // ---- host-side tables: omega powers, coset points, l0 / l_last / l_blind (Lagrange)
  {
    std::vector<Fr> op(n), l0(n, Fr::zero()), ll(n, Fr::zero()), lb(n, Fr::zero());
    Fr cur = Fr::one();
    for (size_t i = 0; i < n; i++) {
      op[i] = cur;
      cur = mul(cur, pk->omega);
    }
    KG_TRY(h2d(ctx, pk->omega_pow, op.data(), n * 32));
    l0[0] = Fr::one();
    ll[n - pk->bf - 1] = Fr::one();
    for (size_t i = n - pk->bf; i < n; i++) lb[i] = Fr::one();
    // to cosets via the same device path as every other polynomial: pack [l0 | l_last | l_blind] into hq-sized temp
    Fr* tmp = pk->scratch;  // >= ext >= 3n? ext >= 2n only when degree >= 4; use three separate passes
    Fr* dst[3] = {pk->l0_c, pk->llast_c, pk->lactive_c};
    std::vector<Fr>* src[3] = {&l0, &ll, &lb};
    for (int t = 0; t < 3; t++) {
      KG_TRY(h2d(ctx, tmp, src[t]->data(), n * 32));
      KG_TRY(amdzk_lagrange_to_coeff_dev(ctx, pk->dom, tmp, 1, n));
      KG_TRY(zk_coeff_to_cosets_r261(ctx, pk->dom, tmp, n, dst[t], ext, 1));
      ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
    }
    std::vector<Fr> xc(ext);
    for (uint32_t c = 0; c < pk->nc; c++) {
      cur = zk_quotient_coset_g(pk->dom, c);
      for (int i = 0; i < 5; i++) cur = add(cur, cur);  // 32 * g_c * omega^i: the points of coset c in radix 2^261
      for (size_t i = 0; i < n; i++) {
        xc[(size_t)c * n + i] = cur;
        cur = mul(cur, pk->omega);
      }
    }
    KG_TRY(h2d(ctx, pk->x_coset, xc.data(), ext * 32));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
    // the identity permutation: delta^j * omega^i (Lagrange) and delta^j * X on the cosets (same radix as x_coset)
    Fr dj = Fr::one();
    const Fr delta = fr_delta();
    for (uint32_t j = 0; j < S; j++) {
      KG_TRY(d2d(ctx, pk->dxw_lag + (size_t)j * n, pk->omega_pow, n * 32));
      KG_TRY(d2d(ctx, pk->dx_coset + (size_t)j * ext, pk->x_coset, ext * 32));
      if (j) {
        KG_TRY(zk_scale(ctx, pk->dxw_lag + (size_t)j * n, n, dj));
        KG_TRY(zk_scale(ctx, pk->dx_coset + (size_t)j * ext, ext, dj));
      }
      dj = mul(dj, delta);
    }
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  }
  // ---- fixed columns and permutation polynomials
  if (F) {
    if (!fixed_values) {
      amdzk_pk_free(ctx, pk);
      ZK_FAIL(ctx, AMDZK_E_INVALID, "keygen: fixed_values is null");
    }
    KG_TRY(h2d(ctx, pk->fixed_lag, fixed_values, (size_t)F * n * 32));
    KG_TRY(d2d(ctx, pk->fixed_poly, pk->fixed_lag, (size_t)F * n * 32));
    KG_TRY(amdzk_lagrange_to_coeff_dev(ctx, pk->dom, pk->fixed_poly, F, n));
    KG_TRY(zk_coeff_to_cosets_r261(ctx, pk->dom, pk->fixed_poly, n, pk->fixed_coset, ext, F));
    KG_TRY(commit_cols(ctx, pk, AMDZK_BASIS_G_LAGRANGE, pk->fixed_lag, F, pk->fixed_commitments));
  }
  if (S) {
    if (!perm_mapping) {
      amdzk_pk_free(ctx, pk);
      ZK_FAIL(ctx, AMDZK_E_INVALID, "keygen: perm_mapping is null");
    }
    // sigma_i(omega^j) = delta^(i') * omega^(j'), (i', j') = mapping[i][j]
    std::vector<Fr> dpow(S), op(n), sig((size_t)S * n);
    Fr delta = fr_delta(), cur = Fr::one();
    for (uint32_t i = 0; i < S; i++) {
      dpow[i] = cur;
      cur = mul(cur, delta);
    }
    cur = Fr::one();
    for (size_t i = 0; i < n; i++) {
      op[i] = cur;
      cur = mul(cur, pk->omega);
    }
    for (uint32_t i = 0; i < S; i++)
      for (size_t j = 0; j < n; j++) {
        uint32_t pi = perm_mapping[2 * ((size_t)i * n + j)], pj = perm_mapping[2 * ((size_t)i * n + j) + 1];
        if (pi >= S || pj >= n) {
          amdzk_pk_free(ctx, pk);
          ZK_FAIL(ctx, AMDZK_E_INVALID, "keygen: permutation mapping out of range");
        }
        sig[(size_t)i * n + j] = mul(dpow[pi], op[pj]);
      }
    KG_TRY(h2d(ctx, pk->sigma_lag, sig.data(), (size_t)S * n * 32));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
    KG_TRY(d2d(ctx, pk->sigma_poly, pk->sigma_lag, (size_t)S * n * 32));
    KG_TRY(amdzk_lagrange_to_coeff_dev(ctx, pk->dom, pk->sigma_poly, S, n));
    KG_TRY(zk_coeff_to_cosets_r261(ctx, pk->dom, pk->sigma_poly, n, pk->sigma_coset, ext, S));
    KG_TRY(commit_cols(ctx, pk, AMDZK_BASIS_G_LAGRANGE, pk->sigma_lag, S, pk->perm_commitments));
  }
  // l_active = 1 - (l_last + l_blind) on the coset: lactive_c currently holds l_blind's coset
  {
    // prog: one - l_last - l_blind, stored into lactive_c. Done with a tiny dedicated program below, after tables exist.
  }

  KG_TRY(build_column_tables(ctx, pk));

  // ---- programs
  auto colkind_slot_lag = [&](std::pair<int, int> kc) { return kc.first == 0 ? pk->sl_adv(kc.second) : kc.first == 1 ? pk->sl_fixed(kc.second) : pk->sl_inst(kc.second); };
  const uint32_t r0 = pk->rots.index(0), r1 = pk->rots.index(1), rm1 = pk->rots.index(-1), rlast = pk->rots.index(-(int32_t)(pk->bf + 1));
  auto COL = [&](uint32_t slot, uint32_t r) { return (slot << 8) | r; };
  // (1) lookup compression: ci[l], ct[l]. Leading lookups whose table is one expression over fixed columns and constants
  // (amdzk_pk::lk_const) get ct[l] once, below.
  {
    Program& pr = pk->prog_compress;
    uint32_t e = pk->num_gates;
    for (uint32_t l = 0; l < L && !getenv("AMDZK_NO_TABLE_CACHE"); l++) {
      const uint32_t ni = pk->lookup_shape[l].first, nt = pk->lookup_shape[l].second;
      bool constant = nt == 1;
      if (constant)
        for (uint32_t w : pk->exprs[e + ni]) constant = constant && (w >> 24) != XOP_ADVICE && (w >> 24) != XOP_INSTANCE;
      if (!constant) break;
      pk->lk_const++;
      e += ni + nt;
    }
    e = pk->num_gates;
    for (uint32_t l = 0; l < L; l++) {
      pr.piece();
      KG_TRY(emit_compressed(ctx, pk, pr, e, pk->lookup_shape[l].first));
      pr.op(OP_STORE, 2 * l);
      pr.pop();
      e += pk->lookup_shape[l].first;
      if (l >= pk->lk_const) {
        KG_TRY(emit_compressed(ctx, pk, pr, e, pk->lookup_shape[l].second));
        pr.op(OP_STORE, 2 * l + 1);
        pr.pop();
      }
      e += pk->lookup_shape[l].second;
    }
  }
  // (2) permutation fractions: den[s] -> frac column s (inverted later), num[s] -> zp column s. Per set: the columns'
  // w_j = (v_j + gamma) / beta stay on the stack and serve both products, prod_j (sigma_j + w_j) and
  // prod_j (delta^j omega^row + w_j); the common factor beta^m of numerator and denominator cancels in num / den.
  {
    Program& pr = pk->prog_pfrac;
    for (uint32_t s = 0; s < ns; s++) {
      const uint32_t lo = s * pk->chunk, hi = std::min(S, lo + pk->chunk), m = hi - lo;
      pr.piece();
      for (uint32_t j = lo; j < hi; j++) {
        pr.op(OP_PUSH_COL, COL(colkind_slot_lag(pk->perm_cols[j]), r0));
        pr.push();
        pr.op(OP_ADD_CONST, pk->c_gamma);
        pr.op(OP_MUL_CONST, pk->c_betainv);
      }
      for (int side = 0; side < 2; side++) {  // 0: denominator (sigma columns), 1: numerator (identity-permutation columns)
        for (uint32_t j = lo; j < hi; j++) {
          // the stack holds the m values w, then (from the second factor on) the running product
          pr.op(OP_PICK, j == lo ? m - 1 : m - (j - lo));
          pr.push();
          pr.op(OP_ADD_COL, COL(side == 0 ? pk->sl_sigma(j) : pk->sl_dxw(j), r0));
          if (j > lo) {
            pr.op(OP_MUL);
            pr.pop();
          }
        }
        if (side == 1) {
          pr.op(OP_NIP, m);
          pr.cur -= m;
        }
        pr.op(OP_STORE, 2 * s + side);
        pr.pop();
      }
    }
  }
  // (3) lookup fractions: den = (a'+beta)(s'+gamma) -> frac2[l]; num = (ci+beta)(ct+gamma) -> zl[l]
  {
    Program& pr = pk->prog_lfrac;
    for (uint32_t l = 0; l < L; l++) {
      pr.piece();
      pr.op(OP_PUSH_COL, COL(pk->sl_la(l), r0));
      pr.push();
      pr.op(OP_ADD_CONST, pk->c_beta);
      pr.op(OP_PUSH_COL, COL(pk->sl_ls(l), r0));
      pr.push();
      pr.op(OP_ADD_CONST, pk->c_gamma);
      pr.op(OP_MUL);
      pr.pop();
      pr.op(OP_STORE, 2 * l);
      pr.pop();
      pr.op(OP_PUSH_COL, COL(pk->sl_ci(l), r0));
      pr.push();
      pr.op(OP_ADD_CONST, pk->c_beta);
      pr.op(OP_PUSH_COL, COL(pk->sl_ct(l), r0));
      pr.push();
      pr.op(OP_ADD_CONST, pk->c_gamma);
      pr.op(OP_MUL);
      pr.pop();
      pr.op(OP_STORE, 2 * l + 1);
      pr.pop();
    }
  }
  // (4) the h(X) numerator: gates, permutation, lookups — evaluation.rs evaluate_h order
  {
    Program& pr = pk->prog_h;
    pr.uses_hot = true;
    auto colkind_slot_ext = colkind_slot_lag;  // fixed/advice/instance share slot numbers in both tables
    for (uint32_t g = 0; g < pk->num_gates; g++) {
      KG_TRY(emit_expr(ctx, pk, pr, pk->exprs[g]));
      pr.op(OP_ACC);
      pr.pop();
    }
    if (ns > 0) {
      // l_0 * (1 - z_0)
      pr.op(OP_PUSH_CONST, pk->c_one); pr.push();
      pr.op(OP_SUB_COL, COL(pk->se_zp(0), r0));
      pr.op(OP_MUL_HOT, 0);
      pr.op(OP_ACC); pr.pop();
      // l_last * (z_l^2 - z_l)
      pr.op(OP_PUSH_COL, COL(pk->se_zp(ns - 1), r0)); pr.push();
      pr.op(OP_SQR);
      pr.op(OP_SUB_COL, COL(pk->se_zp(ns - 1), r0));
      pr.op(OP_MUL_HOT, 1);
      pr.op(OP_ACC); pr.pop();
      // l_0 * (z_i - z_{i-1}(omega^last X))
      for (uint32_t s = 1; s < ns; s++) {
        pr.op(OP_PUSH_COL, COL(pk->se_zp(s), r0)); pr.push();
        pr.op(OP_SUB_COL, COL(pk->se_zp(s - 1), rlast));
        pr.op(OP_MUL_HOT, 0);
        pr.op(OP_ACC); pr.pop();
      }
      // l_active * (z_i(omega X) prod(v + beta sigma + gamma) - z_i(X) prod(v + beta delta^j X + gamma))
      //   = beta^m * l_active * (z_i(omega X) prod(sigma_j + w_j) - z_i(X) prod(delta^j X + w_j)),  w_j = (v_j + gamma) / beta:
      // the w_j stay on the stack for both products (one product per column instead of two), delta^j X is a key column,
      // and beta^m goes into the term's power of y (h_term_beta_pow, upload_ypow)
      for (uint32_t s = 0; s < ns; s++) {
        const uint32_t lo = s * pk->chunk, hi = std::min(S, lo + pk->chunk), m = hi - lo;
        for (uint32_t j = lo; j < hi; j++) {
          pr.op(OP_PUSH_COL, COL(colkind_slot_ext(pk->perm_cols[j]), r0)); pr.push();
          pr.op(OP_ADD_CONST, pk->c_gamma);
          pr.op(OP_MUL_CONST, pk->c_betainv);
        }
        pr.op(OP_PUSH_COL, COL(pk->se_zp(s), r1)); pr.push();
        for (uint32_t j = lo; j < hi; j++) {
          pr.op(OP_PICK, m - (j - lo)); pr.push();
          pr.op(OP_ADD_COL, COL(pk->se_sigma(j), r0));
          pr.op(OP_MUL); pr.pop();
        }
        pr.op(OP_PUSH_COL, COL(pk->se_zp(s), r0)); pr.push();
        for (uint32_t j = lo; j < hi; j++) {
          pr.op(OP_PICK, 1 + m - (j - lo)); pr.push();
          pr.op(OP_ADD_COL, COL(pk->se_dx(j), r0));
          pr.op(OP_MUL); pr.pop();
        }
        pr.op(OP_SUB); pr.pop();
        pr.op(OP_NIP, m); pr.cur -= m;
        pr.op(OP_MUL_HOT, 2);
        pr.next_beta = m;
        pr.op(OP_ACC); pr.pop();
      }
    }
    uint32_t e = pk->num_gates;
    for (uint32_t l = 0; l < L; l++) {
      const uint32_t ni = pk->lookup_shape[l].first, nt = pk->lookup_shape[l].second;
      // l_0 * (1 - z)
      pr.op(OP_PUSH_CONST, pk->c_one); pr.push();
      pr.op(OP_SUB_COL, COL(pk->se_zl(l), r0));
      pr.op(OP_MUL_HOT, 0);
      pr.op(OP_ACC); pr.pop();
      // l_last * (z^2 - z)
      pr.op(OP_PUSH_COL, COL(pk->se_zl(l), r0)); pr.push();
      pr.op(OP_SQR);
      pr.op(OP_SUB_COL, COL(pk->se_zl(l), r0));
      pr.op(OP_MUL_HOT, 1);
      pr.op(OP_ACC); pr.pop();
      // l_active * (z(wX)(a'+beta)(s'+gamma) - z(X)(ci+beta)(ct+gamma))
      pr.op(OP_PUSH_COL, COL(pk->se_zl(l), r1)); pr.push();
      pr.op(OP_PUSH_COL, COL(pk->se_la(l), r0)); pr.push();
      pr.op(OP_ADD_CONST, pk->c_beta);
      pr.op(OP_MUL); pr.pop();
      pr.op(OP_PUSH_COL, COL(pk->se_ls(l), r0)); pr.push();
      pr.op(OP_ADD_CONST, pk->c_gamma);
      pr.op(OP_MUL); pr.pop();
      pr.op(OP_PUSH_COL, COL(pk->se_zl(l), r0)); pr.push();
      KG_TRY(emit_compressed(ctx, pk, pr, e, ni));
      pr.op(OP_ADD_CONST, pk->c_beta);
      pr.op(OP_MUL); pr.pop();
      KG_TRY(emit_compressed(ctx, pk, pr, e + ni, nt));
      pr.op(OP_ADD_CONST, pk->c_gamma);
      pr.op(OP_MUL); pr.pop();
      pr.op(OP_SUB); pr.pop();
      pr.op(OP_MUL_HOT, 2);
      pr.op(OP_ACC); pr.pop();
      // l_0 * (a' - s')
      pr.op(OP_PUSH_COL, COL(pk->se_la(l), r0)); pr.push();
      pr.op(OP_SUB_COL, COL(pk->se_ls(l), r0));
      pr.op(OP_MUL_HOT, 0);
      pr.op(OP_ACC); pr.pop();
      // l_active * (a' - s')(a' - a'(w^-1 X))
      pr.op(OP_PUSH_COL, COL(pk->se_la(l), r0)); pr.push();
      pr.op(OP_SUB_COL, COL(pk->se_ls(l), r0));
      pr.op(OP_PUSH_COL, COL(pk->se_la(l), r0)); pr.push();
      pr.op(OP_SUB_COL, COL(pk->se_la(l), rm1));
      pr.op(OP_MUL); pr.pop();
      pr.op(OP_MUL_HOT, 2);
      pr.op(OP_ACC); pr.pop();
      e += ni + nt;
    }
  }
  if (pk->rots.rots.size() > 255) {
    amdzk_pk_free(ctx, pk);
    ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "keygen: more than 255 distinct rotations");
  }
  KG_TRY(build_output_tables(ctx, pk));
  KG_TRY(dalloc(ctx, pk, &pk->d_consts, pk->consts.size()));
  KG_TRY(h2d(ctx, pk->d_consts, pk->consts.data(), pk->consts.size() * 32));
  KG_TRY(dalloc(ctx, pk, &pk->d_consts261, pk->consts.size()));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  KG_TRY(upload_program(ctx, pk, pk->prog_compress, false));
  KG_TRY(upload_program(ctx, pk, pk->prog_pfrac, false));
  KG_TRY(upload_program(ctx, pk, pk->prog_lfrac, false));
  {
    // pieces of the h(X) program: 6 by default. One piece is 1.5 wavefronts per SIMD at k = 15 (3 cosets x 2^15 rows) and
    // the interpreter alone took 2.30 ms of a lone proof's critical path; 8 pieces 1.70 ms. Latency of one proof, median
    // of 15, two runs each on one box: 1 piece 18.34 / 18.39 ms, 4: 17.71 / 17.86, 6: 17.61 / 17.78, 8: 17.63 / 17.46;
    // 10 proofs in flight: 78.0 / 76.7, 77.8 / 78.3, 78.2 / 78.5, 78.0 / 77.6 proofs/s (no difference).
    // AMDZK_H_PARTS=1..8 for experiments.
    const char* e = getenv("AMDZK_H_PARTS");
    uint32_t parts = e ? (uint32_t)atoi(e) : 6u;
    parts = parts < 1 ? 1 : parts > H_PARTS_MAX ? H_PARTS_MAX : parts;
    pk->h_terms = finalize_limb_program(pk->prog_h, parts);
  }
  pk->h_term_beta_pow = pk->prog_h.term_beta;
  KG_TRY(dalloc(ctx, pk, &pk->d_ypow, (size_t)std::max<uint32_t>(pk->h_terms, 1)));
  KG_TRY(upload_program(ctx, pk, pk->prog_h, true));
  if (getenv("AMDZK_DUMP_PROG")) {  // debugging aid: what the compiled h(X) program is made of
    static const char* names[] = {"END", "PUSH_COL", "PUSH_CONST", "ADD", "SUB", "MUL", "NEG", "MUL_CONST", "ADD_CONST", "MUL_COL",
                                  "ADD_COL", "SUB_COL", "ACC", "STORE", "SQR", "PUSH_HOT", "MUL_HOT", "REDUCE", "SUB_BIG", "NEG_BIG",
                                  "WACC", "WFLUSH"};
    std::map<uint32_t, size_t> hist;
    std::map<std::pair<uint32_t, uint32_t>, size_t> pairs;
    const auto& w = pk->prog_h.words;
    for (size_t i = 0; i < w.size(); i++) {
      hist[w[i] >> 24]++;
      if (i + 1 < w.size()) pairs[{w[i] >> 24, w[i + 1] >> 24}]++;
    }
    fprintf(stderr, "[amdzk] prog_h: %zu instructions, stack depth %u\n", w.size(), pk->prog_h.depth);
    for (auto& kv : hist) fprintf(stderr, "[amdzk]   %-10s %zu\n", kv.first < 22 ? names[kv.first] : "?", kv.second);
    auto is_mul = [](uint32_t o) { return o == OP_MUL || o == OP_MUL_CONST || o == OP_MUL_COL || o == OP_MUL_HOT || o == OP_SQR || o == OP_WACC; };
    size_t mm = 0;
    for (auto& kv : pairs)
      if (is_mul(kv.first.first) && is_mul(kv.first.second)) {
        mm += kv.second;
        fprintf(stderr, "[amdzk]   product -> product: %s -> %s x %zu\n", names[kv.first.first], names[kv.first.second], kv.second);
      }
    fprintf(stderr, "[amdzk]   products directly followed by a product: %zu\n", mm);
  }
  // Constant tables (amdzk_pk::lk_const): ct[l] evaluated here, once, and its canonical, padded, sorted form kept
  if (pk->lk_const) {
    const uint32_t cnt = pk->lk_const, usable = (uint32_t)n - (pk->bf + 1);
    KG_TRY(dalloc(ctx, pk, &pk->lk_ts_const, (size_t)cnt * n));
    Program pr;
    uint32_t e = pk->num_gates;
    for (uint32_t l = 0; l < cnt; l++) {
      pr.piece();
      KG_TRY(emit_expr(ctx, pk, pr, pk->exprs[e + pk->lookup_shape[l].first]));
      pr.op(OP_STORE, 2 * l + 1);
      pr.pop();
      e += pk->lookup_shape[l].first + 1;
    }
    KG_TRY(upload_program(ctx, pk, pr, false));
    KG_TRY(run_program(ctx, pk, pr, false, pk->d_outs_compress, nullptr, "expr_const_tables"));
    KG_TRY(d2d(ctx, pk->lk_ts_const, pk->ct, (size_t)cnt * n * 32));
    KG_TRY(amdzk_fr_to_repr_dev(ctx, pk->lk_ts_const, (size_t)cnt * n));
    ZK_HIP(ctx, hipMemset2DAsync(pk->lk_ts_const + usable, (size_t)n * 32, 0xFF, (size_t)(n - usable) * 32, cnt, ctx->stream));
    KG_TRY(zk_sort_keys(ctx, pk->lk_ts_const, cnt, n, n));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  }
  // l_active coset = 1 - (l_last + l_blind): lactive_c holds l_blind's coset; tiny one-off program
  {
    KG_TRY(upload_consts261(ctx, pk));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
    Program pr;
    pr.op(OP_PUSH_CONST, pk->c_one); pr.push();
    pr.op(OP_SUB_COL, COL(pk->se_llast(), r0));
    pr.op(OP_SUB_COL, COL(pk->se_lactive(), r0));
    pr.op(OP_STORE, 0); pr.pop();
    (void)finalize_limb_program(pr);
    KG_TRY(upload_program(ctx, pk, pr, true));
    Fr** d_out = nullptr;
    KG_TRY(dalloc(ctx, pk, &d_out, 1));
    Fr* tgt = pk->lactive_c;
    KG_TRY(h2d(ctx, d_out, &tgt, sizeof(Fr*)));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
    KG_TRY(run_program(ctx, pk, pr, true, d_out, nullptr, "expr_l_active"));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  }
#undef KG_TRY
  *out = pk;
  return AMDZK_OK;
}

// VK material a verifier needs: commitments of the fixed columns and of the permutation polynomials.
int amdzk_pk_commitments(const amdzk_pk* pk, uint64_t* fixed_out /* F x 8 */, uint64_t* perm_out /* S x 8 */) {
  if (!pk) return AMDZK_E_INVALID;
  if (fixed_out && pk->F) memcpy(fixed_out, pk->fixed_commitments.data(), (size_t)pk->F * 64);
  if (perm_out && pk->S) memcpy(perm_out, pk->perm_commitments.data(), (size_t)pk->S * 64);
  return AMDZK_OK;
}

int amdzk_create_proof_ex(amdzk_ctx* ctx, amdzk_pk* pk, const uint64_t* const* instances, const size_t* instance_lens, const void* d_advice,
                          size_t advice_stride, uint64_t rng_seed, int transcript_kind, uint8_t* proof_out, size_t proof_cap,
                          size_t* proof_len);
static int create_proof_impl(amdzk_ctx* ctx, amdzk_pk* const* pks, size_t ncirc, const uint64_t* const* const* instances,
                             const size_t* const* instance_lens, const void* const* d_advice, size_t advice_stride, RandomSource& rng,
                             int transcript_kind, uint8_t* proof_out, size_t proof_cap, size_t* proof_len);
// one circuit instance
static int create_proof_impl(amdzk_ctx* ctx, amdzk_pk* pk, const uint64_t* const* instances, const size_t* instance_lens, const void* d_advice,
                             size_t advice_stride, RandomSource& rng, int transcript_kind, uint8_t* proof_out, size_t proof_cap,
                             size_t* proof_len) {
  amdzk_pk* const pks[1] = {pk};
  const uint64_t* const* const inst[1] = {instances};
  const size_t* const lens[1] = {instance_lens};
  const void* const adv[1] = {d_advice};
  return create_proof_impl(ctx, pks, 1, inst, lens, adv, advice_stride, rng, transcript_kind, proof_out, proof_cap, proof_len);
}

// Number of Fr::random draws one create_proof makes for this key:
// advice tails + advice blinds, per lookup 2 tails + 2 blinds, per permutation set tail + blind,
// per lookup product tail + blind, the random polynomial + blind, the h-piece blinds.
size_t amdzk_proof_random_count(const amdzk_pk* pk) {
  if (!pk) return 0;
  const size_t bf = pk->bf;
  return (size_t)pk->A * (bf + 1) + pk->A + (size_t)pk->L * (2 * (bf + 1) + 2) + (size_t)pk->nsets * (bf + 1) + (size_t)pk->L * (bf + 1) +
         pk->n + 1 + pk->qdeg;
}

// Length of the proof create_proof writes for this key: commitments — advice, 2 per lookup (A', S'), one per
// permutation set, one per lookup product, the random polynomial, the h pieces, SHPLONK's two — then the
// evaluations: advice and fixed queries, the random polynomial, sigma columns, 3 per permutation set but 2
// for the last, 5 per lookup. With AMDZK_MULTIOPEN_GWC the two SHPLONK points become one per opening point.
// Distinct evaluation points of the proof's queries = distinct rotations (omega^r x are pairwise different
// for the |r| << n that occur): what ProverGWC writes one witness commitment for.
static size_t opening_point_count(const amdzk_pk* pk) {
  std::vector<int> rots = {0};  // sigma columns, h(X), the random polynomial
  auto note = [&](int r) {
    if (std::find(rots.begin(), rots.end(), r) == rots.end()) rots.push_back(r);
  };
  for (auto& q : pk->advice_queries) note(q.second);
  for (auto& q : pk->fixed_queries) note(q.second);
  if (pk->nsets) note(1);
  if (pk->nsets > 1) note(-(int)(pk->bf + 1));
  if (pk->L) {
    note(1);
    note(-1);
  }
  return rots.size();
}

size_t amdzk_proof_size(const amdzk_pk* pk, int format) {
  if (!pk) return 0;
  const int transcript_kind = format & 0xff;
  const size_t openings = (format & AMDZK_MULTIOPEN_GWC) ? opening_point_count(pk) : 2;
  const size_t points = (size_t)pk->A + 2 * (size_t)pk->L + pk->nsets + pk->L + 1 + pk->qdeg + openings;
  const size_t scalars = pk->advice_queries.size() + pk->fixed_queries.size() + 1 + pk->S + (pk->nsets ? 3 * (size_t)pk->nsets - 1 : 0) +
                         5 * (size_t)pk->L;
  return points * (transcript_kind == AMDZK_TRANSCRIPT_KECCAK256_EVM ? 64 : 32) + scalars * 32;
}

// create_proof with the caller's randomness: `scalars` = amdzk_proof_random_count(pk) Fr elements
// (Montgomery), drawn by the caller with Fr::random(&mut rng) in order.
int amdzk_create_proof_scalars(amdzk_ctx* ctx, amdzk_pk* pk, const uint64_t* const* instances, const size_t* instance_lens,
                               const void* d_advice, size_t advice_stride, const uint64_t* scalars, size_t scalar_count, int transcript_kind,
                               uint8_t* proof_out, size_t proof_cap, size_t* proof_len) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!pk || !scalars) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof_scalars: null argument");
  if (scalar_count < amdzk_proof_random_count(pk))
    ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof_scalars: %zu scalars given, %zu needed", scalar_count, amdzk_proof_random_count(pk));
  RandomSource rs;
  rs.scalars = scalars;
  rs.count = scalar_count;
  return create_proof_impl(ctx, pk, instances, instance_lens, d_advice, advice_stride, rs, transcript_kind, proof_out, proof_cap, proof_len);
}

int amdzk_create_proof(amdzk_ctx* ctx, amdzk_pk* pk, const uint64_t* const* instances, const size_t* instance_lens, const void* d_advice,
                       size_t advice_stride, uint64_t rng_seed, uint8_t* proof_out, size_t proof_cap, size_t* proof_len) {
  ZK_ENTER(ctx);
  return amdzk_create_proof_ex(ctx, pk, instances, instance_lens, d_advice, advice_stride, rng_seed, AMDZK_TRANSCRIPT_BLAKE2B, proof_out,
                               proof_cap, proof_len);
}

int amdzk_create_proof_ex(amdzk_ctx* ctx, amdzk_pk* pk, const uint64_t* const* instances, const size_t* instance_lens, const void* d_advice,
                          size_t advice_stride, uint64_t rng_seed, int transcript_kind, uint8_t* proof_out, size_t proof_cap,
                          size_t* proof_len) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  ChaCha20Rng chacha(rng_seed);
  RandomSource rs;
  rs.rng = &chacha;
  return create_proof_impl(ctx, pk, instances, instance_lens, d_advice, advice_stride, rs, transcript_kind, proof_out, proof_cap, proof_len);
}

static int create_proof_body(amdzk_ctx* ctx, amdzk_pk* const* pks, size_t ncirc, const uint64_t* const* const* instances_all,
                             const size_t* const* instance_lens_all, const void* const* d_advice_all, size_t advice_stride, RandomSource& rng,
                             int transcript_kind, uint8_t* proof_out, size_t proof_cap, size_t* proof_len);
static int create_proof_impl(amdzk_ctx* ctx, amdzk_pk* const* pks, size_t ncirc, const uint64_t* const* const* instances,
                             const size_t* const* instance_lens, const void* const* d_advice, size_t advice_stride, RandomSource& rng,
                             int transcript_kind, uint8_t* proof_out, size_t proof_cap, size_t* proof_len) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!pks || ncirc == 0 || !pks[0]) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: no proving key");
  amdzk_pk* pk = pks[0];
  // AMDZK_MSM_PIPELINE=1 (experiment, off by default): lanes mode also cuts every commitment batch into column groups on
  // two streams with chained level-1 kernels (msm.hip zk_msm_dev_xyzz). Measured slower on both counts — 22.7-25.8 ms
  // against 21.5 for one proof, 66-74 against 74 proofs/s — because a group's small kernels wait for workgroup slots
  // behind the other group's level-1 kernel (profiles/r03b_stream_priorities_and_gating.txt). The caller's ctx gets
  // its own setting back.
  const bool keep_pipeline = ctx->msm_pipeline;
  if (pk->use_lanes && ncirc == 1 && getenv("AMDZK_MSM_PIPELINE") && atoi(getenv("AMDZK_MSM_PIPELINE")) != 0) {
    ctx->msm_pipeline = true;
    for (amdzk_ctx* l : ctx->lanes)
      if (l) l->msm_pipeline = true;
  }
  const int r = create_proof_body(ctx, pks, ncirc, instances, instance_lens, d_advice, advice_stride, rng, transcript_kind, proof_out, proof_cap, proof_len);
  ctx->msm_pipeline = keep_pipeline;
  if (r != AMDZK_OK) {  // a failed proof may have left work on the lanes: the key's workspace must be quiet before it is used again
    const std::string keep = ctx->err;
    (void)zk_sync_all(ctx);
    ctx->err = keep;
  }
  return r;
}
// One proof over ncirc instances of the circuit (upstream's `circuits: &[C]`, `instances: &[&[&[F]]]`): pks[c] holds
// instance c's workspace — the key itself for c = 0, workspace clones of it for the others (amdzk_pk_clone_workspace).
// `pk` below is pks[0]: what the proof has once (transcript representative, random polynomial, h(X), multiopen buffers,
// staging); every per-circuit step runs in a loop over the instances with `pk` shadowed by that instance's key.
// Upstream's order (plonk/prover.rs [UP]): instances, advice, lookup permutations, permutation products and lookup
// products are each written circuit after circuit; the challenges, the random polynomial and h(X) exist once; the
// evaluations are advice (per circuit), fixed, random, sigma, permutation products (per circuit), lookups (per circuit).
static int create_proof_body(amdzk_ctx* ctx, amdzk_pk* const* pks, size_t ncirc, const uint64_t* const* const* instances_all,
                             const size_t* const* instance_lens_all, const void* const* d_advice_all, size_t advice_stride, RandomSource& rng,
                             int transcript_kind, uint8_t* proof_out, size_t proof_cap, size_t* proof_len) {
  amdzk_pk* const pk = pks[0];
  const size_t NC = ncirc;
  if (!proof_len) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: null argument");
  for (size_t c = 0; c < NC; c++) {
    if (!pks[c] || (pk->A && (!d_advice_all || !d_advice_all[c]))) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: null argument (circuit %zu)", c);
    const amdzk_pk* root_c = pks[c]->clone_of ? pks[c]->clone_of : pks[c];
    const amdzk_pk* root_0 = pk->clone_of ? pk->clone_of : pk;
    if (root_c != root_0) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: circuit %zu's key is not the first key or a workspace clone of it", c);
    for (size_t d = 0; d < c; d++)
      if (pks[d] == pks[c]) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: circuits %zu and %zu share one workspace", d, c);
  }
  const size_t n = pk->n;
  const uint32_t F = pk->F, A = pk->A, I = pk->I, S = pk->S, L = pk->L, ns = pk->nsets, bf = pk->bf;
  const size_t usable = n - (bf + 1);
  if (advice_stride < n) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: advice stride < n");
  const bool use_gwc = (transcript_kind & AMDZK_MULTIOPEN_GWC) != 0;
  transcript_kind &= ~AMDZK_MULTIOPEN_GWC;
  zkhost::Blake2bWrite t_blake;
  zkhost::Keccak256Write t_keccak;
  if (transcript_kind != AMDZK_TRANSCRIPT_BLAKE2B && transcript_kind != AMDZK_TRANSCRIPT_KECCAK256_EVM)
    ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: unknown transcript kind %d", transcript_kind);
  zkhost::TranscriptWrite& T = transcript_kind == AMDZK_TRANSCRIPT_BLAKE2B ? (zkhost::TranscriptWrite&)t_blake : (zkhost::TranscriptWrite&)t_keccak;
  const bool ttrace = getenv("AMDZK_TRACE_TIME") != nullptr;
  auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tlast = tnow();
  auto tick = [&](const char* label) {
    if (!ttrace) return;
    zk_host_wait(ctx, ctx->stream);
    double t = tnow();
    fprintf(stderr, "[amdzk-time] %-28s %8.3f ms\n", label, t - tlast);
    tlast = t;
  };
  // Lanes: M = the caller's ctx (everything the transcript waits for), B and C = its auxiliary streams (common.hpp).
  // M carries the chain commitment -> challenge -> next phase; B takes each phase's columns to coefficient form and to
  // the quotient domain as soon as they are blinded (out of place: the commitments and the next phase's programs keep
  // reading the Lagrange values); C computes the lookup products beside the permutation products and commits the random
  // polynomial at the very start. Only the ORDER OF TRANSCRIPT WRITES is upstream's;
  // the arithmetic between two challenges is unordered there too. With AMDZK_KEYGEN_SERIAL, or while per-kernel
  // profiling is on, B = C = M and everything below degenerates to one stream.
  amdzk_ctx *M = ctx, *B = ctx, *C = ctx;
  if (pk->use_lanes && NC == 1) {  // several instances: one stream (each lane holds one commitment batch's result at a time)
    ZK_TRY(zk_lane(ctx, 0, &B));
    ZK_TRY(zk_lane(ctx, 1, &C));
    B->msm_pipeline = C->msm_pipeline = ctx->msm_pipeline;
  }
  const bool serial = B == M;
  // latency mode of the commitments (common.hpp): while this proof runs on lanes; the caller's setting comes back at the end
  struct LatencyMode {
    amdzk_ctx* c[3];
    bool keep[3];
    LatencyMode(amdzk_ctx* m, amdzk_ctx* b, amdzk_ctx* cc, bool on) : c{m, b, cc} {
      for (int i = 0; i < 3; i++) keep[i] = c[i]->msm_latency_mode, c[i]->msm_latency_mode = on || keep[i];
    }
    ~LatencyMode() {
      for (int i = 0; i < 3; i++) c[i]->msm_latency_mode = keep[i];
    }
  } latency_mode(M, B, C, !serial && !(getenv("AMDZK_LATENCY_MODE") && atoi(getenv("AMDZK_LATENCY_MODE")) == 0));
  auto lane_id = [&](amdzk_ctx* l) { return l == M ? 0 : l == B ? 1 : 2; };
// a failure on a lane is reported through the caller's ctx
#define LN_TRY(lane, expr)                                  \
  do {                                                      \
    int _lr = (expr);                                       \
    if (_lr != AMDZK_OK) {                                  \
      if ((lane) != ctx) ctx->err = (lane)->err;            \
      return _lr;                                           \
    }                                                       \
  } while (0)
  auto upload_small_on = [&](amdzk_ctx* l, const std::vector<Fr>& v, size_t off_elems) -> int {
    if (off_elems + v.size() > pk->small_cap) ZK_FAIL(ctx, AMDZK_E_NOMEM, "create_proof: small buffer overflow");
    LN_TRY(l, h2d_staged(l, pk, pk->small_l[lane_id(l)] + off_elems, v.data(), v.size() * 32));
    return AMDZK_OK;
  };
  auto upload_small = [&](const std::vector<Fr>& v, size_t off_elems) -> int { return upload_small_on(M, v, off_elems); };
  auto write_points = [&](const std::vector<G1Affine>& pts, const char* label) -> int {
    for (auto& p : pts) {
      trace_pt(label, p);
      if (!T.write_point(p)) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: %s commitment is the identity (cannot write points at infinity to the transcript)", label);
    }
    return AMDZK_OK;
  };
  // blinding tails: cnt scalars per column (column-major draw order) scattered into rows [row0, row0+cnt), on lane l
  auto blind_rows = [&](amdzk_ctx* l, Fr* d_cols, uint32_t ncols, size_t row0, uint32_t cnt, const std::vector<Fr>& vals) -> int {
    ZK_TRY(upload_small_on(l, vals, 0));  // staged through pinned memory: `vals` may die right after
    LN_TRY(l, zk_scatter_rows(l, d_cols, n, row0, pk->small_l[lane_id(l)], cnt, ncols));
    return AMDZK_OK;
  };
  // A commitment batch begun on a lane and collected when the transcript needs it. On one stream (serial) it is
  // collected at once: a context holds one batch's result at a time.
  struct Commit {
    PendingCommit pc;
    std::vector<G1Affine> pts;
    bool begun = false, done = false;
  };
  auto commit_end = [&](Commit& c) -> int {
    if (c.begun && !c.done) LN_TRY(c.pc.ctx, commit_finish(c.pc, c.pts));
    c.done = true;
    return AMDZK_OK;
  };
  auto commit_begin = [&](amdzk_ctx* l, int basis, const Fr* d_cols, size_t ncols, Commit& c) -> int {
    LN_TRY(l, commit_launch(l, pk, basis, d_cols, ncols, c.pc));
    c.begun = true;
    if (serial) ZK_TRY(commit_end(c));
    return AMDZK_OK;
  };
  // columns [first, first + count) of the arena, blinded on lane `after`: coefficients (PQ) and quotient-domain values (PC) on B
  // after_l1: the columns' commitment batch has already been launched on `after`; the transforms start behind its
  // level-1 kernel (both fill the chip: side by side they only stretch each other) and run beside its tail and beside
  // the next phase's latency-bound kernels instead
  static const bool gate_l1 = !(getenv("AMDZK_NTT_AFTER_L1") && atoi(getenv("AMDZK_NTT_AFTER_L1")) == 0);
  auto transforms_on_B = [&](amdzk_pk* pk, amdzk_ctx* after, size_t first, size_t count, bool after_l1 = false) -> int {
    if (!count) return AMDZK_OK;
    // (behind the WHOLE batch instead — AMDZK_NTT_AFTER_L1=2 in an experiment — measured 19.1-19.8 ms per proof against 18.7-19.1)
    if (after_l1 && gate_l1) ZK_TRY(zk_stream_after_l1(B, after));
    else ZK_TRY(zk_stream_after(B, after));
    LN_TRY(B, zk_lagrange_to_coeff(B, pk->dom, pk->P + first * n, n, pk->PQ + first * n, n, count));
    LN_TRY(B, zk_coeff_to_cosets_r261(B, pk->dom, pk->PQ + first * n, n, pk->PC + first * pk->ext, pk->ext, count));
    return AMDZK_OK;
  };

  // 0. vk, instances
  T.common_scalar(pk->transcript_repr);
  for (size_t ci = 0; ci < NC && I; ci++) {  // columns are zero beyond the caller's values: clear on the device, upload only what was given
    amdzk_pk* const pk = pks[ci];
    const uint64_t* const* instances = instances_all ? instances_all[ci] : nullptr;
    const size_t* instance_lens = instance_lens_all ? instance_lens_all[ci] : nullptr;
    ZK_HIP(ctx, hipMemsetAsync(pk->inst(), 0, (size_t)I * n * 32, ctx->stream));
    std::vector<Fr> iv;
    for (uint32_t c = 0; c < I; c++) {
      const size_t len = instance_lens ? instance_lens[c] : 0;
      if (len > usable) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: instance column %u too long (InstanceTooLarge)", c);
      if (!len) continue;
      if (!instances || !instances[c]) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: instance column %u is null", c);
      iv.resize(len);
      for (size_t i = 0; i < len; i++) {
        memcpy(iv[i].l, instances[c] + 4 * i, 32);
        T.common_scalar(iv[i]);
      }
      ZK_TRY(h2d_staged(ctx, pk, pk->inst() + (size_t)c * n, iv.data(), len * 32));
      // without room in the pinned staging area the copy reads `iv` asynchronously: finish it before the next column reuses it
      if (!pk->pin || len * 32 > pk->pin_cap) ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
    }
  }
  // The random polynomial of the vanishing argument (step 5 below) depends on nothing but the RNG: its n draws follow
  // all blinding draws, whose number is fixed by the key, so they are taken from that position of the stream now —
  // generated and committed on lane C while M commits the advice columns.
  const size_t draws_before_random = NC * ((size_t)A * (bf + 1) + A + (size_t)L * (2 * (bf + 1) + 2) + (size_t)ns * (bf + 1) + (size_t)L * (bf + 1));
  // With the seeded ChaCha20Rng every Fr::random is one key-stream block, draw j = block ctr0 + j, and WHICH draw blinds
  // which cell is fixed by the key: the blinding tails are generated on the device straight into their rows
  // (chacha20_blind_rows, like the random polynomial) and the host only moves its counter past them — it used to draw
  // ~1,800 scalars per proof at 0.3 us each, a third of them in front of the proof's first kernel. A caller's own
  // RngCore (pre-drawn scalars) keeps the host path. Draw positions, upstream's order:
  //   advice of instance c: A (bf + 1) tails, column-major, then A blinds;  then per instance and lookup: bf + 1 tails
  //   of A', bf + 1 of S', two blinds;  then per instance and permutation set bf tails + a blind;  then per instance and
  //   lookup product bf tails + a blind;  then the random polynomial.
  const uint64_t ctr0 = rng.rng ? rng.rng->block_counter() : 0;
  const size_t per_adv = (size_t)A * (bf + 1) + A, per_lk = (size_t)L * (2 * (bf + 1) + 2), per_pz = (size_t)ns * (bf + 1), per_lz = (size_t)L * (bf + 1);
  const size_t base_lk = NC * per_adv, base_pz = base_lk + NC * per_lk, base_lz = base_pz + NC * per_pz;
  auto blind_dev = [&](amdzk_ctx* l, Fr* d_cols, uint32_t ncols, size_t row0, uint32_t cnt, size_t first_draw, uint32_t draw_stride) -> int {
    LN_TRY(l, zk_chacha20_blind_rows(l, d_cols, n, row0, cnt, ncols, rng.rng->key(), ctr0 + first_draw, draw_stride, zkhost::fr_r3()));
    return AMDZK_OK;
  };
  Commit cm_rnd;
  std::vector<Commit> cm_zp_all(NC), cm_zl_all(NC);
  {
    ZK_TRY(zk_stream_after(C, M));  // the previous proof on this key may still be reading rnd on M's stream
    if (rng.rng) {
      // ChaCha20Rng: every draw is one key-stream block, so draw j of the stream is block j: one kernel
      if (!rng.rng->at_block_boundary()) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: RNG stream not on a block boundary");
      LN_TRY(C, zk_chacha20_fr_random(C, pk->rnd, n, rng.rng->key(), rng.rng->block_counter() + draws_before_random, zkhost::fr_r3()));
    } else {  // the caller's own RngCore: its pre-drawn scalars
      if (draws_before_random + n > rng.count) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: ran out of caller-supplied random scalars");
      LN_TRY(C, h2d(C, pk->rnd, rng.scalars + 4 * draws_before_random, n * 32));  // the caller's buffer outlives the call
    }
    ZK_TRY(commit_begin(C, AMDZK_BASIS_G, pk->rnd, 1, cm_rnd));
  }
  // 1. advice: copy in, blind the unusable rows of every column, draw the (unused) blinds, commit
  for (size_t ci = 0; ci < NC; ci++) {
    amdzk_pk* const pk = pks[ci];
    if (A) {
      ZK_HIP(ctx, hipMemcpy2DAsync(pk->adv(), n * 32, d_advice_all[ci], advice_stride * 32, n * 32, A, hipMemcpyDeviceToDevice, ctx->stream));
      if (rng.rng) {
        ZK_TRY(blind_dev(M, pk->adv(), A, usable, bf + 1, ci * per_adv, bf + 1));
        rng.rng->skip_blocks(per_adv);
      } else {
        std::vector<Fr> tail((size_t)A * (bf + 1));
        for (auto& v : tail) v = rng.fr();
        for (uint32_t c = 0; c < A; c++) (void)rng.fr();
        ZK_TRY(blind_rows(M, pk->adv(), A, usable, bf + 1, tail));
      }
    }
    Commit cm;
    if (A && !serial && gate_l1) ZK_TRY(commit_begin(M, AMDZK_BASIS_G_LAGRANGE, pk->adv(), A, cm));
    ZK_TRY(transforms_on_B(pk, M, 0, (size_t)A + I, cm.begun));
    if (A && !cm.begun) ZK_TRY(commit_begin(M, AMDZK_BASIS_G_LAGRANGE, pk->adv(), A, cm));
    ZK_TRY(commit_end(cm));
    ZK_TRY(write_points(cm.pts, "advice"));
  }
  ZK_TRY(commit_end(cm_rnd));  // long done; lane C's MSM workspace is free for the lookup products' commitment
  tick("advice");
  Fr theta = T.squeeze_challenge();
  trace_fr("theta", theta);
  for (size_t ci = 0; ci < NC; ci++) {
    amdzk_pk* const pk = pks[ci];
    pk->consts[pk->c_theta] = theta;
    ZK_TRY(h2d_staged(ctx, pk, pk->d_consts + pk->c_theta, &pk->consts[pk->c_theta], 32));
  }
  // 2. lookups: compress, permute (on the device), blind, commit
  amdzk_pk* const pk0 = pk;  // the small staging buffers the lambdas above write are the first key's
  for (size_t ci = 0; ci < NC && L; ci++) {
    amdzk_pk* const pk = pks[ci];
    ZK_TRY(run_program(ctx, pk, pk->prog_compress, false, pk->d_outs_compress, nullptr, "expr_lookup_compress"));
    ZK_TRY(d2d(ctx, pk->la(), pk->ci, (size_t)L * n * 32));
    ZK_TRY(zk_permute_expression_pairs(ctx, pk->la(), pk->ct, pk->lk_ts, pk->ls(), pk->lk_left, pk->lk_flags, pk->d_err, L, (uint32_t)n,
                                       (uint32_t)usable, pk->lk_ts_const, pk->lk_const));
    // the "input not in table" word comes down behind the permutation and is read once the host has waited for this
    // phase's commitment anyway (it used to be a host wait of its own in the middle of the phase: 0.1 ms of idle device)
    *pk->h_err = 0;
    ZK_HIP(ctx, hipMemcpyAsync(pk->h_err, pk->d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    tick("  lookup: device permute");
    // RNG order per lookup: a' tail, s' tail, blind(a'), blind(s')
    if (rng.rng) {
      const uint32_t per = 2 * (bf + 1) + 2;
      ZK_TRY(blind_dev(M, pk->la(), L, usable, bf + 1, base_lk + ci * per_lk, per));
      ZK_TRY(blind_dev(M, pk->ls(), L, usable, bf + 1, base_lk + ci * per_lk + (bf + 1), per));
      rng.rng->skip_blocks(per_lk);
    } else {
      std::vector<Fr> ta((size_t)L * (bf + 1)), ts((size_t)L * (bf + 1));
      for (uint32_t l = 0; l < L; l++) {
        for (uint32_t i = 0; i <= bf; i++) ta[(size_t)l * (bf + 1) + i] = rng.fr();
        for (uint32_t i = 0; i <= bf; i++) ts[(size_t)l * (bf + 1) + i] = rng.fr();
        (void)rng.fr();
        (void)rng.fr();
      }
      ZK_TRY(blind_rows(M, pk->la(), L, usable, bf + 1, ta));
      ZK_TRY(upload_small(ts, ta.size()));
      ZK_TRY(zk_scatter_rows(ctx, pk->ls(), n, usable, pk0->small + ta.size(), bf + 1, L));
    }
    Commit cmc;
    if (!serial && gate_l1) ZK_TRY(commit_begin(M, AMDZK_BASIS_G_LAGRANGE, pk->la(), 2 * L, cmc));
    ZK_TRY(transforms_on_B(pk, M, (size_t)A + I, 2 * (size_t)L, cmc.begun));
    if (!cmc.begun) ZK_TRY(commit_begin(M, AMDZK_BASIS_G_LAGRANGE, pk->la(), 2 * L, cmc));
    ZK_TRY(commit_end(cmc));
    if (*pk->h_err) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: lookup %d input not in table (ConstraintSystemFailure)", *pk->h_err - 1);
    const std::vector<G1Affine>& cm = cmc.pts;
    for (uint32_t l = 0; l < L; l++) {
      std::vector<G1Affine> two = {cm[l], cm[L + l]};
      ZK_TRY(write_points(two, "lookup_permuted"));
    }
  }
  tick("lookups_permuted");
  Fr beta = T.squeeze_challenge();
  Fr gamma = T.squeeze_challenge();
  trace_fr("beta", beta);
  trace_fr("gamma", gamma);
  // the permutation factors are evaluated as beta (sigma + w) and beta (delta^j X + w) with w = (v + gamma) / beta
  if (S && beta.is_zero()) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "create_proof: the challenge beta is zero (probability 2^-254): the factored permutation terms need 1 / beta");
  const Fr beta_inv = inv(beta);
  for (size_t ci = 0; ci < NC; ci++) {
    amdzk_pk* const pk = pks[ci];
    pk->consts[pk->c_beta] = beta;
    pk->consts[pk->c_gamma] = gamma;
    pk->consts[pk->c_betainv] = beta_inv;
    ZK_TRY(h2d_staged(ctx, pk, pk->d_consts + pk->c_beta, &pk->consts[pk->c_beta], (size_t)(pk->consts.size() - pk->c_beta) * 32));
  }
  tick("  perm: challenges+consts");
  // 3. + 4. permutation grand products on M, lookup grand products on C (they depend on beta and gamma only, not on
  // each other). RNG order: the permutation sets' tails and blinds, then the lookups'.
  // (Several instances: every instance's permutation products are committed before the first lookup product.)
  std::vector<std::vector<Fr>> tail_p_all(NC), tail_l_all(NC);
  if (rng.rng) {
    rng.rng->skip_blocks(NC * (per_pz + per_lz));  // generated on the device, below
  } else {
    for (size_t ci = 0; ci < NC; ci++) {
      tail_p_all[ci].resize((size_t)ns * bf);
      for (uint32_t s = 0; s < ns; s++) {
        for (uint32_t i = 0; i < bf; i++) tail_p_all[ci][(size_t)s * bf + i] = rng.fr();
        (void)rng.fr();
      }
    }
    for (size_t ci = 0; ci < NC; ci++) {
      tail_l_all[ci].resize((size_t)L * bf);
      for (uint32_t l = 0; l < L; l++) {
        for (uint32_t i = 0; i < bf; i++) tail_l_all[ci][(size_t)l * bf + i] = rng.fr();
        (void)rng.fr();
      }
    }
  }
  // 5. vanishing: the random polynomial's n draws and its blind (generated above from this position of the stream)
  if (rng.rng) rng.rng->skip_blocks(n);
  else rng.used += n;
  (void)rng.fr();
  auto lookup_products = [&](size_t ci, bool ordered_behind_m) -> int {
    amdzk_pk* const pk = pks[ci];
    const std::vector<Fr>& tail_l = tail_l_all[ci];
    Commit& cm_zl = cm_zl_all[ci];
    if (!ordered_behind_m) ZK_TRY(zk_stream_after(C, M));  // beta, gamma and the permuted columns are in place
    LN_TRY(C, run_program(C, pk, pk->prog_lfrac, false, pk->d_outs_lfrac, nullptr, "expr_lookup_fractions"));
    LN_TRY(C, zk_batch_invert(C, pk->frac2, pk->scratch2, (size_t)L * n));
    LN_TRY(C, zk_mul_elem(C, pk->zl(), pk->frac2, (size_t)L * n));
    LN_TRY(C, zk_running_product(C, pk->zl(), L, n, n, false, 0, pk->scan_tmp2));
    if (rng.rng) ZK_TRY(blind_dev(C, pk->zl(), L, n - bf, bf, base_lz + ci * per_lz, bf + 1));
    else ZK_TRY(blind_rows(C, pk->zl(), L, n - bf, bf, tail_l));
    // ... and their commitment, enqueued BEFORE the permutation chain: the lookup chain is the shorter one, so its
    // level-1 kernel runs while M is still in fractions, inversion and scans rather than beside M's own level-1 kernel.
    // (Measured: 18.8-19.3 ms per proof either way — what one lane gains the other loses; kept for the simpler order.)
    if (serial || !gate_l1) ZK_TRY(transforms_on_B(pk, C, (size_t)A + I + 2 * L + ns, L));
    ZK_TRY(commit_begin(C, AMDZK_BASIS_G_LAGRANGE, pk->zl(), L, cm_zl));
    if (!serial && gate_l1) ZK_TRY(transforms_on_B(pk, C, (size_t)A + I + 2 * L + ns, L, true));
    return AMDZK_OK;
  };
  auto perm_products = [&](size_t ci) -> int {  // fractions, inversion, running products, blinding: everything in front of the commitment
    amdzk_pk* const pk = pks[ci];
    const std::vector<Fr>& tail_p = tail_p_all[ci];
    ZK_TRY(run_program(ctx, pk, pk->prog_pfrac, false, pk->d_outs_pfrac, nullptr, "expr_perm_fractions"));
    tick("  perm: fractions program");
    ZK_TRY(zk_batch_invert(ctx, pk->frac, pk->scratch, (size_t)ns * n));
    tick("  perm: batch invert");
    ZK_TRY(zk_mul_elem(ctx, pk->zp(), pk->frac, (size_t)ns * n));
    ZK_TRY(zk_running_product(ctx, pk->zp(), ns, n, n, true, usable, pk->scan_tmp));
    tick("  perm: running product");
    if (rng.rng) ZK_TRY(blind_dev(M, pk->zp(), ns, n - bf, bf, base_pz + ci * per_pz, bf + 1));
    else ZK_TRY(blind_rows(M, pk->zp(), ns, n - bf, bf, tail_p));
    return AMDZK_OK;
  };
  auto perm_commit = [&](size_t ci) -> int {
    amdzk_pk* const pk = pks[ci];
    if (serial || !gate_l1) ZK_TRY(transforms_on_B(pk, M, (size_t)A + I + 2 * L, ns));
    ZK_TRY(commit_begin(M, AMDZK_BASIS_G_LAGRANGE, pk->zp(), ns, cm_zp_all[ci]));
    return AMDZK_OK;
  };
  if (!serial) {
    // Lanes (one instance): the HOST enqueues M's chain first — seven launches the transcript waits for — and lane C's
    // lookup products (a dozen launches and a commitment batch's fourteen) while M's fractions and inversion run: the
    // device used to sit 0.48 ms behind beta / gamma waiting for M's first kernel (profiles/r03zz_timeline_single_proof.txt).
    if (L) ZK_TRY(zk_stream_after(C, M));  // C starts behind beta, gamma and the permuted columns — NOT behind M's products below
    if (ns) ZK_TRY(perm_products(0));
    if (L) ZK_TRY(lookup_products(0, true));
    if (ns) ZK_TRY(perm_commit(0));
  } else {
    for (size_t ci = 0; ci < NC && L; ci++) ZK_TRY(lookup_products(ci, false));
    for (size_t ci = 0; ci < NC && ns; ci++) {
      ZK_TRY(perm_products(ci));
      ZK_TRY(perm_commit(ci));
    }
  }
  if (ns && !serial && gate_l1) ZK_TRY(transforms_on_B(pk, M, (size_t)A + I + 2 * L, ns, true));  // lanes: one instance
  for (size_t ci = 0; ci < NC; ci++) {
    ZK_TRY(commit_end(cm_zp_all[ci]));
    ZK_TRY(write_points(cm_zp_all[ci].pts, "perm_z"));
  }
  tick("perm_products");
  for (size_t ci = 0; ci < NC; ci++) {
    ZK_TRY(commit_end(cm_zl_all[ci]));
    ZK_TRY(write_points(cm_zl_all[ci].pts, "lookup_z"));
  }
  tick("lookup_products");
  ZK_TRY(write_points(cm_rnd.pts, "random_poly"));
  Fr y = T.squeeze_challenge();
  trace_fr("y", y);
  for (size_t ci = 0; ci < NC; ci++) {
    amdzk_pk* const pk = pks[ci];
    pk->consts[pk->c_y] = y;
    ZK_TRY(h2d_staged(ctx, pk, pk->d_consts + pk->c_y, &pk->consts[pk->c_y], 32));
  }
  // 6. h(X): every committed column is on the quotient domain once lane B has drained
  ZK_TRY(zk_stream_after(M, B));
  for (size_t ci = 0; ci < NC; ci++) ZK_TRY(quotient_from_cosets(ctx, pks[ci]));  // theta, beta, gamma, delta powers, y are all known by now
  if (NC > 1) {
    // evaluate_h folds the instances' terms in ONE Horner chain with y, instance after instance: with K terms per instance
    // the numerator is sum_c y^(K (NC - 1 - c)) * numerator_c, and the division by X^n - 1, the interpolation and the cut
    // into pieces are linear — so the pieces are the same combination of the instances' pieces.
    std::vector<const Fr*> pp(NC);
    std::vector<Fr> cf(NC);
    const Fr yK = pow_u64(y, pk->h_terms);
    Fr cur = Fr::one();
    for (size_t ci = NC; ci-- > 0;) {
      pp[ci] = pks[ci]->hpieces;
      cf[ci] = cur;
      cur = mul(cur, yK);
    }
    const size_t len = (size_t)pk->qdeg * n;
    ZK_TRY(h2d_staged(ctx, pk, pk->ptrs, pp.data(), pp.size() * sizeof(Fr*)));
    ZK_TRY(upload_small(cf, 0));
    ZK_TRY(zk_lincomb(ctx, (const Fr* const*)pk->ptrs, pk->small, (uint32_t)NC, pk->scratch, len, false));  // scratch holds >= ext >= qdeg * n
    ZK_TRY(d2d(ctx, pk->hpieces, pk->scratch, len * 32));
  }
  {
    for (uint32_t i = 0; i < pk->qdeg; i++) (void)rng.fr();  // h_blinds
    std::vector<G1Affine> cm;
    ZK_TRY(commit_cols(ctx, pk, AMDZK_BASIS_G, pk->hpieces, pk->qdeg, cm));  // consecutive n-blocks
    ZK_TRY(write_points(cm, "h_piece"));
  }
  tick("h_eval+commit");
  Fr x = T.squeeze_challenge();
  trace_fr("x", x);
  Fr xn = pow_u64(x, n);
  // h_poly = sum_i piece_i * xn^i
  {
    std::vector<const Fr*> pp(pk->qdeg);
    std::vector<Fr> cf(pk->qdeg);
    Fr cur = Fr::one();
    for (uint32_t i = 0; i < pk->qdeg; i++) {
      pp[i] = pk->hpieces + (size_t)i * n;
      cf[i] = cur;
      cur = mul(cur, xn);
    }
    ZK_TRY(h2d_staged(ctx, pk, pk->ptrs, pp.data(), pp.size() * sizeof(Fr*)));
    ZK_TRY(upload_small(cf, 0));
    ZK_TRY(zk_lincomb(ctx, (const Fr* const*)pk->ptrs, pk->small, pk->qdeg, pk->hpoly, n, false));
  }
  // 7. evaluations. One list of (polynomial, rotation) in proof order, then the two extra
  //    evaluations SHPLONK needs (h_poly at x; random at x is already in the list).
  amdzk_pk::Multiopen& mo = NC == 1 ? pk->mo : pk->mo_multi;
  if (NC > 1) {  // the cached lists name the polynomials of one particular list of instance keys
    std::vector<const amdzk_pk*> keys(pks, pks + NC);
    if (keys != pk->mo_multi_keys) {
      pk->mo_multi = amdzk_pk::Multiopen();
      pk->mo_multi_keys = keys;
    }
  }
  if (!mo.built) {
    auto rot_id = [&](int rot) -> uint32_t {
      for (size_t i = 0; i < mo.rots.size(); i++)
        if (mo.rots[i] == rot) return (uint32_t)i;
      mo.rots.push_back(rot);
      return (uint32_t)mo.rots.size() - 1;
    };
    auto addq = [&](const Fr* p, int rot) {
      mo.ev.push_back({p, rot});
      mo.ev_rot.push_back(rot_id(rot));
    };
    // written evaluations: advice (instance after instance), fixed, random, sigma, permutation products (instance after
    // instance), lookups (instance after instance)
    for (size_t ci = 0; ci < NC; ci++)
      for (auto& q : pk->advice_queries) addq(pks[ci]->q_adv() + (size_t)q.first * n, q.second);
    for (auto& q : pk->fixed_queries) addq(pk->fixed_poly + (size_t)q.first * n, q.second);
    addq(pk->rnd, 0);
    for (uint32_t i = 0; i < S; i++) addq(pk->sigma_poly + (size_t)i * n, 0);
    for (size_t ci = 0; ci < NC; ci++)
      for (uint32_t s = 0; s < ns; s++) {
        addq(pks[ci]->q_zp() + (size_t)s * n, 0);
        addq(pks[ci]->q_zp() + (size_t)s * n, 1);
        if (s + 1 < ns) addq(pks[ci]->q_zp() + (size_t)s * n, -(int)(bf + 1));
      }
    for (size_t ci = 0; ci < NC; ci++)
      for (uint32_t l = 0; l < L; l++) {
        addq(pks[ci]->q_zl() + (size_t)l * n, 0);
        addq(pks[ci]->q_zl() + (size_t)l * n, 1);
        addq(pks[ci]->q_la() + (size_t)l * n, 0);
        addq(pks[ci]->q_la() + (size_t)l * n, -1);
        addq(pks[ci]->q_ls() + (size_t)l * n, 0);
      }
    mo.n_written = mo.ev.size();
    addq(pk->hpoly, 0);
    // 8. multiopen queries in upstream order
    std::map<std::pair<const Fr*, int>, uint32_t> where;
    for (size_t i = 0; i < mo.ev.size(); i++) where.emplace(mo.ev[i], (uint32_t)i);
    bool missing = false;
    auto addpq = [&](const Fr* p, int rot) {
      auto it = where.find({p, rot});
      if (it == where.end()) {
        missing = true;
        return;
      }
      mo.q_poly.push_back(p);
      mo.q_rot.push_back(rot_id(rot));
      mo.q_ev.push_back(it->second);
    };
    // per instance: advice queries, the permutation argument's openings, the lookups' openings; then what exists once
    for (size_t ci = 0; ci < NC; ci++) {
      amdzk_pk* const pk = pks[ci];
      for (auto& q : pk->advice_queries) addpq(pk->q_adv() + (size_t)q.first * n, q.second);
      for (uint32_t s = 0; s < ns; s++) {
        addpq(pk->q_zp() + (size_t)s * n, 0);
        addpq(pk->q_zp() + (size_t)s * n, 1);
      }
      for (int s = (int)ns - 2; s >= 0; s--) addpq(pk->q_zp() + (size_t)s * n, -(int)(bf + 1));
      for (uint32_t l = 0; l < L; l++) {
        addpq(pk->q_zl() + (size_t)l * n, 0);
        addpq(pk->q_la() + (size_t)l * n, 0);
        addpq(pk->q_ls() + (size_t)l * n, 0);
        addpq(pk->q_la() + (size_t)l * n, -1);
        addpq(pk->q_zl() + (size_t)l * n, 1);
      }
    }
    for (auto& q : pk->fixed_queries) addpq(pk->fixed_poly + (size_t)q.first * n, q.second);
    for (uint32_t i = 0; i < S; i++) addpq(pk->sigma_poly + (size_t)i * n, 0);
    addpq(pk->hpoly, 0);
    addpq(pk->rnd, 0);
    if (missing) {
      mo = amdzk_pk::Multiopen();
      pk->mo_multi_keys.clear();
      ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: a multiopen query has no evaluation");
    }
    // shplonk construct_intermediate_sets, in terms of rotations: the polynomials with their sets of rotations
    // (first seen first), then the distinct sets with their polynomials (first seen first)
    std::vector<const Fr*> cr_poly;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> cr_rots;  // (rot id, ev index), ascending rot id
    std::map<const Fr*, uint32_t> cr_of;
    for (size_t i = 0; i < mo.q_poly.size(); i++) {
      auto it = cr_of.find(mo.q_poly[i]);
      if (it == cr_of.end()) {
        it = cr_of.emplace(mo.q_poly[i], (uint32_t)cr_poly.size()).first;
        cr_poly.push_back(mo.q_poly[i]);
        cr_rots.emplace_back();
      }
      auto& v = cr_rots[it->second];
      const std::pair<uint32_t, uint32_t> e{mo.q_rot[i], mo.q_ev[i]};
      auto pos = std::lower_bound(v.begin(), v.end(), e, [](const auto& x1, const auto& x2) { return x1.first < x2.first; });
      if (pos == v.end() || pos->first != e.first) v.insert(pos, e);
    }
    for (size_t c = 0; c < cr_poly.size(); c++) {
      std::vector<uint32_t> ids, evs;
      for (auto& e : cr_rots[c]) ids.push_back(e.first), evs.push_back(e.second);
      amdzk_pk::Multiopen::Set* hit = nullptr;
      for (auto& st : mo.sets)
        if (st.rot_ids == ids) hit = &st;
      if (!hit) {
        mo.sets.emplace_back();
        hit = &mo.sets.back();
        hit->rot_ids = ids;
      }
      hit->polys.push_back(cr_poly[c]);
      hit->ev_idx.push_back(evs);
    }
    size_t pairs = 0;
    for (auto& st : mo.sets) pairs += st.rot_ids.size();
    if (mo.sets.size() > pk->max_sets) {
      mo = amdzk_pk::Multiopen();
      ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "create_proof: more than %u rotation sets", pk->max_sets);
    }
    if (std::max<size_t>(pairs, 1) > pk->sets_Q_pairs) {
      ZK_TRY(dalloc(ctx, pk, &pk->sets_Q, std::max<size_t>(pairs, 1) * n));
      pk->sets_Q_pairs = std::max<size_t>(pairs, 1);
    }
    mo.built = true;
  }
  // the points x * omega^rot, once per distinct rotation
  const size_t nrot = mo.rots.size();
  std::vector<Fr> rot_pt(nrot);
  std::vector<std::array<uint64_t, 4>> rot_canon(nrot);
  for (size_t r = 0; r < nrot; r++) {
    rot_pt[r] = rotate_omega(pk, x, mo.rots[r]);
    rot_canon[r] = canon(rot_pt[r]);
  }
  std::vector<Fr> evals(mo.ev.size());
  {
    const size_t nq = mo.ev.size();
    if (nq > pk->ptrs_cap || 2 * nq > pk->small_cap) ZK_FAIL(ctx, AMDZK_E_NOMEM, "create_proof: too many queries (%zu)", nq);
    std::vector<const Fr*> pp(nq);
    std::vector<Fr> pts(nq);
    for (size_t i = 0; i < nq; i++) pp[i] = mo.ev[i].first, pts[i] = rot_pt[mo.ev_rot[i]];
    ZK_TRY(h2d_staged(ctx, pk, pk->ptrs, pp.data(), nq * sizeof(Fr*)));
    ZK_TRY(upload_small(pts, 0));
    ZK_TRY(zk_poly_eval(ctx, (const Fr* const*)pk->ptrs, pk->small, pk->small + nq, nq, (uint32_t)n));
    ZK_TRY(d2h(ctx, evals.data(), pk->small + nq, nq * 32));
    for (size_t i = 0; i < mo.n_written; i++) T.write_scalar(evals[i]);
  }
  tick("evals");

  // 9a. GWC (multiopen/gwc/prover.rs [UP]): v <- transcript; queries grouped by point in first-seen order;
  // per point z:  W_z = (sum_j v^j p_j - sum_j v^j p_j(z)) / (X - z), committed and written in that order.
  if (use_gwc) {
    struct PS {
      std::array<uint64_t, 4> key;
      Fr z;
      std::vector<const Fr*> polys;
      std::vector<Fr> evals;
    };
    std::vector<PS> psets;  // one per distinct point (= distinct rotation), first seen first
    std::vector<int> ps_of_rot(nrot, -1);
    for (size_t i = 0; i < mo.q_poly.size(); i++) {
      const uint32_t r = mo.q_rot[i];
      if (ps_of_rot[r] < 0) {
        ps_of_rot[r] = (int)psets.size();
        psets.push_back(PS{rot_canon[r], rot_pt[r], {}, {}});
      }
      PS& hit = psets[ps_of_rot[r]];
      hit.polys.push_back(mo.q_poly[i]);
      hit.evals.push_back(evals[mo.q_ev[i]]);
    }
    const size_t np = psets.size();
    if (np > 16) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "create_proof: more than 16 opening points");
    Fr v = T.squeeze_challenge();
    trace_fr("gwc_v", v);
    for (size_t i = 0; i < np; i++) {
      const size_t m = psets[i].polys.size();
      if (m > pk->ptrs_cap || m + 1 > pk->small_cap) ZK_FAIL(ctx, AMDZK_E_NOMEM, "create_proof: opening set too large");
      std::vector<Fr> cf(m);
      Fr cur = Fr::one(), eb = Fr::zero();
      for (size_t j = 0; j < m; j++) {
        cf[j] = cur;
        eb = add(eb, mul(cur, psets[i].evals[j]));
        cur = mul(cur, v);
      }
      std::vector<Fr> low = {eb};
      Fr* Wi = pk->sets_N + i * n;
      ZK_TRY(h2d_staged(ctx, pk, pk->ptrs, psets[i].polys.data(), m * sizeof(Fr*)));
      ZK_TRY(upload_small(cf, 0));
      ZK_TRY(upload_small(low, m));
      ZK_TRY(zk_lincomb(ctx, (const Fr* const*)pk->ptrs, pk->small, (uint32_t)m, Wi, n, false));
      ZK_TRY(zk_sub_low(ctx, Wi, pk->small + m, 1));
    }
    std::vector<Fr*> pp(np);
    std::vector<Fr> roots(np);
    for (size_t i = 0; i < np; i++) {
      pp[i] = pk->sets_N + i * n;
      roots[i] = psets[i].z;
    }
    ZK_TRY(h2d_staged(ctx, pk, pk->ptrs, pp.data(), np * sizeof(Fr*)));
    ZK_TRY(upload_small(roots, 0));
    ZK_TRY(zk_kate_div(ctx, (Fr* const*)pk->ptrs, pk->small, np, (uint32_t)n));
    std::vector<G1Affine> cm;
    ZK_TRY(commit_cols(ctx, pk, AMDZK_BASIS_G, pk->sets_N, np, cm));
    ZK_TRY(write_points(cm, "gwc_w"));
  } else
  // 9b. SHPLONK (multiopen/shplonk/prover.rs [UP])
  {
    // construct_intermediate_sets: the sets are the key's (mo.sets); their points, and the super point set, are kept
    // in ascending order of the canonical field elements as upstream's BTreeSets do
    const size_t nr = mo.sets.size();
    if (nr > 16) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "create_proof: more than 16 rotation sets");
    auto by_point = [&](uint32_t r1, uint32_t r2) { return fr_less_canon(rot_canon[r1], rot_canon[r2]); };
    std::vector<uint32_t> super(nrot);
    for (size_t r = 0; r < nrot; r++) super[r] = (uint32_t)r;
    std::sort(super.begin(), super.end(), by_point);
    Fr ys = T.squeeze_challenge();
    Fr v = T.squeeze_challenge();
    trace_fr("shplonk_y", ys);
    trace_fr("shplonk_v", v);
    // Per set i with points p_t (ascending): R_i(X) = sum_j y^j R_ij(X), R_ij the interpolation of polynomial j's
    // evaluations. Interpolation is linear, so R_i is the interpolation of E_i[t] = sum_j y^j eval_ij[t]:
    // R_i = sum_t E_i[t] c_it prod_{s != t} (X - p_s), c_it = 1 / prod_{s != t} (p_t - p_s) — one host product per
    // evaluation instead of one small interpolation per polynomial, and ONE field inversion (batched over all c_it)
    // instead of one per basis polynomial and point: the host used to spend 0.26 ms here with the GPU idle.
    std::vector<std::vector<Fr>> set_pts(nr);
    std::vector<std::vector<uint32_t>> set_order(nr);  // positions in rot_ids, by ascending point
    std::vector<std::vector<Fr>> set_c(nr);            // c_it
    {
      std::vector<Fr> dens;
      for (size_t i = 0; i < nr; i++) {
        const amdzk_pk::Multiopen::Set& st = mo.sets[i];
        const size_t m = st.rot_ids.size();
        std::vector<uint32_t>& order = set_order[i];
        order.resize(m);
        for (size_t t = 0; t < m; t++) order[t] = (uint32_t)t;
        std::sort(order.begin(), order.end(), [&](uint32_t t1, uint32_t t2) { return by_point(st.rot_ids[t1], st.rot_ids[t2]); });
        for (size_t t = 0; t < m; t++) set_pts[i].push_back(rot_pt[st.rot_ids[order[t]]]);
        for (size_t t = 0; t < m; t++) {
          Fr den = Fr::one();
          for (size_t s2 = 0; s2 < m; s2++)
            if (s2 != t) den = mul(den, sub(set_pts[i][t], set_pts[i][s2]));
          dens.push_back(den);  // non-zero: the points of a set are distinct
        }
      }
      // Montgomery's trick: prefix products, one inversion, walk back
      std::vector<Fr> pre(dens.size() + 1, Fr::one());
      for (size_t k2 = 0; k2 < dens.size(); k2++) pre[k2 + 1] = mul(pre[k2], dens[k2]);
      Fr acc = inv(pre[dens.size()]);
      std::vector<Fr> dinv(dens.size());
      for (size_t k2 = dens.size(); k2-- > 0;) {
        dinv[k2] = mul(acc, pre[k2]);
        acc = mul(acc, dens[k2]);
      }
      size_t at = 0;
      for (size_t i = 0; i < nr; i++)
        for (size_t t = 0; t < set_pts[i].size(); t++) set_c[i].push_back(dinv[at++]);
    }
    std::vector<std::vector<Fr>> lowsum(nr);  // R_i, coefficients
    for (size_t i = 0; i < nr; i++) {
      const amdzk_pk::Multiopen::Set& st = mo.sets[i];
      const size_t np = set_pts[i].size();
      std::vector<Fr> E(np, Fr::zero());
      Fr yp = Fr::one();
      for (size_t j = 0; j < st.polys.size(); j++) {
        for (size_t t = 0; t < np; t++) E[t] = add(E[t], mul(yp, evals[st.ev_idx[j][set_order[i][t]]]));
        yp = mul(yp, ys);
      }
      lowsum[i].assign(np, Fr::zero());
      for (size_t t = 0; t < np; t++) {
        std::vector<Fr> num(1, Fr::one());  // prod_{s != t} (X - p_s), ascending coefficients
        for (size_t s2 = 0; s2 < np; s2++) {
          if (s2 == t) continue;
          num.push_back(Fr::zero());
          for (size_t d = num.size() - 1; d > 0; d--) num[d] = sub(num[d - 1], mul(set_pts[i][s2], num[d]));
          num[0] = neg(mul(set_pts[i][s2], num[0]));
        }
        const Fr w = mul(E[t], set_c[i][t]);
        for (size_t d = 0; d < np; d++) lowsum[i][d] = add(lowsum[i][d], mul(w, num[d]));
      }
    }
    // L_i = sum_j y^j P_ij ; N_i = (L_i - R_i) / prod_t (X - p_t), R_i = sum_j y^j R_ij. The division runs once, not once
    // per point: 1 / prod_t (X - p_t) = sum_t c_t / (X - p_t) with c_t = 1 / prod_{s != t} (p_t - p_s) (the points of a
    // set are distinct), and L_i - R_i vanishes at every p_t, so N_i = sum_t c_t Q_it with Q_it = (L_i - R_i) / (X - p_t)
    // — all (set, point) quotients in ONE division launch, then h(X) = sum_i v^i N_i = sum_it (v^i c_it) Q_it in one
    // linear combination. Exact arithmetic, same polynomial. The L_i of different sets are independent: one lane each.
    size_t maxm = 0;
    for (size_t i = 0; i < nr; i++) maxm = std::max(maxm, set_pts[i].size());
    std::vector<const Fr*> q_src;
    std::vector<Fr*> q_dst;
    std::vector<Fr> q_root, q_low, q_coef;
    amdzk_ctx* lanes3[3] = {M, B, C};
    Fr vpow = Fr::one();
    for (size_t i = 0; i < nr; i++) {
      const size_t m = mo.sets[i].polys.size(), np = set_pts[i].size();
      if (m > pk->ptrs_cap || m > pk->small_cap / 2) ZK_FAIL(ctx, AMDZK_E_NOMEM, "create_proof: rotation set too large");
      std::vector<Fr> cf(m);
      Fr cur = Fr::one();
      for (size_t j = 0; j < m; j++) {
        cf[j] = cur;
        cur = mul(cur, ys);
      }
      amdzk_ctx* ln = lanes3[i % 3];
      const int li = lane_id(ln);
      if (ln != M && i < 3) ZK_TRY(zk_stream_after(ln, M));  // the evaluations above came off M; hpoly is in place
      LN_TRY(ln, h2d_staged(ln, pk, pk->ptrs_l[li], mo.sets[i].polys.data(), m * sizeof(Fr*)));
      ZK_TRY(upload_small_on(ln, cf, 0));
      Fr* Li = pk->sets_L + i * n;
      LN_TRY(ln, zk_lincomb(ln, (const Fr* const*)pk->ptrs_l[li], pk->small_l[li], (uint32_t)m, Li, n, false));
      for (size_t t = 0; t < np; t++) {
        q_src.push_back(Li);
        q_dst.push_back(pk->sets_Q + q_dst.size() * n);
        q_root.push_back(set_pts[i][t]);
        q_coef.push_back(mul(vpow, set_c[i][t]));
        for (size_t d = 0; d < maxm; d++) q_low.push_back(d < np ? lowsum[i][d] : Fr::zero());
      }
      vpow = mul(vpow, v);
    }
    ZK_TRY(zk_stream_after(M, B));
    ZK_TRY(zk_stream_after(M, C));
    {
      const size_t nq = q_dst.size();
      if (2 * nq > pk->ptrs_cap || nq * (maxm + 2) > pk->small_cap) ZK_FAIL(ctx, AMDZK_E_NOMEM, "create_proof: too many opening points");
      void** pt = (void**)pk->ptrs;
      ZK_TRY(h2d_staged(ctx, pk, pt, q_dst.data(), nq * sizeof(Fr*)));
      ZK_TRY(h2d_staged(ctx, pk, pt + nq, q_src.data(), nq * sizeof(Fr*)));
      ZK_TRY(upload_small(q_root, 0));
      ZK_TRY(upload_small(q_low, nq));
      ZK_TRY(upload_small(q_coef, nq + q_low.size()));
      ZK_TRY(zk_kate_div_from(ctx, (Fr* const*)pt, (const Fr* const*)(pt + nq), pk->small, pk->small + nq, (uint32_t)maxm, nq, (uint32_t)n));
      ZK_TRY(zk_lincomb(ctx, (const Fr* const*)pt, pk->small + nq + q_low.size(), (uint32_t)nq, pk->hx, n, false));
    }
    std::vector<G1Affine> cm;
    ZK_TRY(commit_cols(ctx, pk, AMDZK_BASIS_G, pk->hx, 1, cm));
    ZK_TRY(write_points(cm, "shplonk_h1"));
    Fr u = T.squeeze_challenge();
    trace_fr("u", u);
    // l(X) = sum_i v^i z_i (L_i - r_i) - zt(u) h(X);  then / (X - u) / z_0 — the factor 1 / z_0 rides in on the coefficients
    Fr zt = Fr::one();
    for (uint32_t r : super) zt = mul(zt, sub(u, rot_pt[r]));
    std::vector<const Fr*> pp(nr + 1);
    std::vector<Fr> cf(nr + 1);
    Fr cur = Fr::one(), z0 = Fr::one(), cterm = Fr::zero();
    for (size_t i = 0; i < nr; i++) {
      Fr zi = Fr::one();
      for (uint32_t r : super)
        if (!std::binary_search(mo.sets[i].rot_ids.begin(), mo.sets[i].rot_ids.end(), r)) zi = mul(zi, sub(u, rot_pt[r]));
      if (i == 0) z0 = zi;
      const Fr ri = eval_small(lowsum[i], u);  // R_i(u) = sum_j y^j R_ij(u)
      Fr w = mul(cur, zi);
      pp[i] = pk->sets_L + i * n;
      cf[i] = w;
      cterm = add(cterm, mul(w, ri));
      cur = mul(cur, v);
    }
    pp[nr] = pk->hx;
    cf[nr] = neg(zt);
    const Fr z0inv = inv(z0);
    for (auto& c : cf) c = mul(c, z0inv);
    cterm = mul(cterm, z0inv);
    Fr* lx = pk->sets_Q;  // reuse
    ZK_TRY(h2d_staged(ctx, pk, pk->ptrs, pp.data(), (nr + 1) * sizeof(Fr*)));
    ZK_TRY(upload_small(cf, 0));
    std::vector<Fr> tailv = {cterm, u};
    ZK_TRY(upload_small(tailv, nr + 1));
    ZK_TRY(zk_lincomb(ctx, (const Fr* const*)pk->ptrs, pk->small, (uint32_t)(nr + 1), lx, n, false));
    std::vector<Fr*> one_p = {lx};
    ZK_TRY(h2d_staged(ctx, pk, (void**)pk->ptrs + nr + 1, one_p.data(), sizeof(Fr*)));
    ZK_TRY(zk_kate_div_from(ctx, (Fr* const*)((void**)pk->ptrs + nr + 1), nullptr, pk->small + nr + 2, pk->small + nr + 1, 1, 1, (uint32_t)n));
    ZK_TRY(commit_cols(ctx, pk, AMDZK_BASIS_G, lx, 1, cm));
    ZK_TRY(write_points(cm, "shplonk_h2"));
  }
  tick("multiopen");
  if (rng.exhausted) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: ran out of caller-supplied random scalars");
  *proof_len = T.proof.size();
  if (proof_out) {
    if (proof_cap < T.proof.size()) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof: proof buffer too small (%zu < %zu)", proof_cap, T.proof.size());
    memcpy(proof_out, T.proof.data(), T.proof.size());
  }
  (void)F;
  return AMDZK_OK;
#undef LN_TRY
}

// One more circuit instance's workspace for `src`'s circuit: a key handle that shares src's key material (fixed and
// permutation columns in all three forms, the domain, the compiled programs' text, constant lookup tables — read-only
// during proofs) and owns its own per-proof workspace, pointer tables and uploaded programs (their instructions carry
// absolute column addresses). What amdzk_create_proof_multi takes for its second, third, ... instance — and, since a
// clone is a complete key for create_proof, the cheap way to keep several proofs of one circuit in flight: a
// clone costs the workspace (the arenas), not the key (354 + 354 MiB of permutation cosets at the metric's shape).
// Free it with amdzk_pk_free BEFORE the key it was made from.
int amdzk_pk_clone_workspace(amdzk_ctx* ctx, const amdzk_pk* src, amdzk_pk** out) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!src || !out) ZK_FAIL(ctx, AMDZK_E_INVALID, "pk_clone_workspace: null argument");
  amdzk_pk* pk = new amdzk_pk(*src);
  pk->clone_of = src->clone_of ? src->clone_of : src;
  pk->allocs.clear();
  pk->pin = nullptr;
  pk->pin_cap = pk->pin_off = 0;
  pk->mo = amdzk_pk::Multiopen();
  pk->mo_multi = amdzk_pk::Multiopen();
  pk->mo_multi_keys.clear();
  pk->sets_Q = nullptr;
  pk->sets_Q_pairs = 0;
  pk->prog_compress.d_instr = pk->prog_pfrac.d_instr = pk->prog_lfrac.d_instr = pk->prog_h.d_instr = nullptr;
#define CL_TRY(x)                \
  do {                           \
    int _r = (x);                \
    if (_r != AMDZK_OK) {        \
      amdzk_pk_free(ctx, pk);    \
      return _r;                 \
    }                            \
  } while (0)
  CL_TRY(alloc_proof_workspace(ctx, pk));
  CL_TRY(build_column_tables(ctx, pk));
  CL_TRY(build_output_tables(ctx, pk));
  CL_TRY(dalloc(ctx, pk, &pk->d_consts, pk->consts.size()));
  CL_TRY(h2d(ctx, pk->d_consts, pk->consts.data(), pk->consts.size() * 32));
  CL_TRY(dalloc(ctx, pk, &pk->d_consts261, pk->consts.size()));
  CL_TRY(dalloc(ctx, pk, &pk->d_ypow, (size_t)std::max<uint32_t>(pk->h_terms, 1)));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  CL_TRY(upload_program(ctx, pk, pk->prog_compress, false));
  CL_TRY(upload_program(ctx, pk, pk->prog_pfrac, false));
  CL_TRY(upload_program(ctx, pk, pk->prog_lfrac, false));
  CL_TRY(upload_program(ctx, pk, pk->prog_h, true));
  CL_TRY(upload_consts261(ctx, pk));
  // the compressed constant tables (amdzk_pk::lk_const) are written once, at keygen, into the key's ct columns — the
  // per-proof compression program skips them — so a new workspace starts with a copy
  if (pk->lk_const) CL_TRY(d2d(ctx, pk->ct, src->ct, (size_t)pk->lk_const * pk->n * 32));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
#undef CL_TRY
  *out = pk;
  return AMDZK_OK;
}

// plonk::create_proof(params, pk, &[circuit; N], &[instances; N], rng, transcript) [UP]: n_circuits instances of the
// key's circuit in ONE proof. pks[c]: instance c's workspace — pks[0] the key (or a clone), the others workspace clones of
// the same key, all different. instances[c][col] / instance_lens[c][col] and d_advice[c] as for amdzk_create_proof_ex.
int amdzk_create_proof_multi(amdzk_ctx* ctx, amdzk_pk* const* pks, size_t n_circuits, const uint64_t* const* const* instances,
                             const size_t* const* instance_lens, const void* const* d_advice, size_t advice_stride, uint64_t rng_seed,
                             int transcript_kind, uint8_t* proof_out, size_t proof_cap, size_t* proof_len) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!pks || n_circuits == 0) ZK_FAIL(ctx, AMDZK_E_INVALID, "create_proof_multi: no circuits");
  ChaCha20Rng chacha(rng_seed);
  RandomSource rs;
  rs.rng = &chacha;
  return create_proof_impl(ctx, pks, n_circuits, instances, instance_lens, d_advice, advice_stride, rs, transcript_kind, proof_out, proof_cap, proof_len);
}

// Byte length of the proof amdzk_create_proof_multi writes for n_circuits instances (amdzk_proof_size for one).
size_t amdzk_proof_size_multi(const amdzk_pk* pk, size_t n_circuits, int format) {
  if (!pk || n_circuits == 0) return 0;
  const int transcript_kind = format & 0xff;
  const size_t openings = (format & AMDZK_MULTIOPEN_GWC) ? opening_point_count(pk) : 2;
  const size_t points = n_circuits * ((size_t)pk->A + 2 * (size_t)pk->L + pk->nsets + pk->L) + 1 + pk->qdeg + openings;
  const size_t scalars = n_circuits * (pk->advice_queries.size() + (pk->nsets ? 3 * (size_t)pk->nsets - 1 : 0) + 5 * (size_t)pk->L) +
                         pk->fixed_queries.size() + 1 + pk->S;
  return points * (transcript_kind == AMDZK_TRANSCRIPT_KECCAK256_EVM ? 64 : 32) + scalars * 32;
}

// ---- function-by-function entry points of the PLONK layer: the same kernels create_proof runs,
// callable on their own — by the isolated parity tests and by a fork that replaces upstream one function at a time.

// evaluate_h and the quotient: d_polys = the NP = A + I + 2L + nsets + L committed polynomials in COEFFICIENT form, n
// each, in the key's arena order [advice | instance | permuted inputs A' | permuted tables S' | permutation products
// | lookup products]; the challenges as Montgomery Fr. Writes the cs_degree - 1 pieces of h(X) (n coefficients each)
// to d_pieces_out. Uses (and overwrites) the key's per-proof workspace.
int amdzk_quotient_eval_dev(amdzk_ctx* ctx, amdzk_pk* pk, const void* d_polys, size_t poly_stride, const uint64_t theta[4],
                            const uint64_t beta[4], const uint64_t gamma[4], const uint64_t y[4], void* d_pieces_out) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!pk || !d_polys || !theta || !beta || !gamma || !y || !d_pieces_out) ZK_FAIL(ctx, AMDZK_E_INVALID, "quotient_eval: null argument");
  if (poly_stride < pk->n) ZK_FAIL(ctx, AMDZK_E_INVALID, "quotient_eval: stride < n");
  ZK_HIP(ctx, hipMemcpy2DAsync(pk->PQ, pk->n * 32, d_polys, poly_stride * 32, pk->n * 32, pk->NP, hipMemcpyDeviceToDevice, ctx->stream));
  memcpy(pk->consts[pk->c_theta].l, theta, 32);
  memcpy(pk->consts[pk->c_beta].l, beta, 32);
  memcpy(pk->consts[pk->c_gamma].l, gamma, 32);
  memcpy(pk->consts[pk->c_y].l, y, 32);
  if (pk->S && pk->consts[pk->c_beta].is_zero()) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "quotient_eval: beta is zero");
  pk->consts[pk->c_betainv] = inv(pk->consts[pk->c_beta]);
  ZK_TRY(quotient_pieces(ctx, pk));
  ZK_TRY(d2d(ctx, d_pieces_out, pk->hpieces, (size_t)pk->qdeg * pk->n * 32));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  return AMDZK_OK;
}

// What the last create_proof on this key left in its workspace, for tests that check one stage at a time against the
// oracle: what = 0 the NP committed polynomials in coefficient form ([NP][n], arena order as above); 1 the
// challenges theta, beta, gamma, y; 2 the pieces of h(X) ([cs_degree - 1][n]). `out` holds cap Fr elements;
// *count = elements available.
int amdzk_pk_inspect(amdzk_ctx* ctx, const amdzk_pk* pk, int what, uint64_t* out, size_t cap, size_t* count) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!pk || !count) ZK_FAIL(ctx, AMDZK_E_INVALID, "pk_inspect: null argument");
  const Fr* src = nullptr;
  size_t cnt = 0;
  Fr ch[4];
  if (what == 0) {
    src = pk->PQ;
    cnt = pk->NP * pk->n;
  } else if (what == 2) {
    src = pk->hpieces;
    cnt = (size_t)pk->qdeg * pk->n;
  } else if (what == 1) {
    ch[0] = pk->consts[pk->c_theta];
    ch[1] = pk->consts[pk->c_beta];
    ch[2] = pk->consts[pk->c_gamma];
    ch[3] = pk->consts[pk->c_y];
    cnt = 4;
  } else {
    ZK_FAIL(ctx, AMDZK_E_INVALID, "pk_inspect: unknown selector %d", what);
  }
  *count = cnt;
  if (!out) return AMDZK_OK;
  if (cap < cnt) ZK_FAIL(ctx, AMDZK_E_INVALID, "pk_inspect: buffer holds %zu elements, %zu needed", cap, cnt);
  if (what == 1) memcpy(out, ch, sizeof(ch));
  else ZK_TRY(d2h(ctx, out, src, cnt * 32));
  return AMDZK_OK;
}

// Test hooks for the host pass that prepares quotient-domain programs for the limb-resident interpreter
// (finalize_limb_program): (a) the pass on caller-supplied program words — pure host code, no device — and (b) the
// finalised h(X) program of a key. Words are `op << 24 | arg` with the opcodes of csrc/plonk_kernels.hpp.
int amdzk_debug_limb_program(const uint32_t* words, size_t n, uint32_t* out, size_t cap, size_t* out_n, uint32_t* depth) {
  if ((!words && n) || !out_n) return AMDZK_E_INVALID;
  Program pr;
  pr.words.assign(words, words + n);
  (void)finalize_limb_program(pr);
  *out_n = pr.words.size();
  if (depth) *depth = pr.depth;
  if (out) {
    if (cap < pr.words.size()) return AMDZK_E_INVALID;
    memcpy(out, pr.words.data(), pr.words.size() * sizeof(uint32_t));
  }
  return AMDZK_OK;
}
int amdzk_pk_h_program(const amdzk_pk* pk, uint32_t* out, size_t cap, size_t* out_n) {
  if (!pk || !out_n) return AMDZK_E_INVALID;
  *out_n = pk->prog_h.words.size();
  if (out) {
    if (cap < pk->prog_h.words.size()) return AMDZK_E_INVALID;
    memcpy(out, pk->prog_h.words.data(), pk->prog_h.words.size() * sizeof(uint32_t));
  }
  return AMDZK_OK;
}

// lookup::prover::permute_expression_pair for nlookups (input, table) pairs of n rows each, Montgomery form,
// column l at + l * n: d_inputs is sorted in place into A', d_permuted_tables_out receives S'; rows >= usable of both
// are zero. Fails with "not in table" when an input value is missing from its table.
int amdzk_permute_expression_pair_dev(amdzk_ctx* ctx, void* d_inputs, const void* d_tables, void* d_permuted_tables_out, size_t nlookups,
                                      uint32_t n, uint32_t usable) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!d_inputs || !d_tables || !d_permuted_tables_out) ZK_FAIL(ctx, AMDZK_E_INVALID, "permute_expression_pair: null argument");
  if (n < 2 || (n & (n - 1)) || usable > n) ZK_FAIL(ctx, AMDZK_E_INVALID, "permute_expression_pair: n must be a power of two >= usable");
  char* ws = nullptr;  // Ts[L][n] | left[L][n] | flags 4 x L x (n + 8) | err
  const size_t L = nlookups, col = L * (size_t)n * 32, fl = 4 * L * ((size_t)n + 8) * 4;
  ZK_TRY(zk_ws_reserve(ctx, 5, 2 * col + fl + 256, (void**)&ws));
  if (L == 0) return AMDZK_OK;
  ZK_TRY(zk_permute_expression_pairs(ctx, (Fr*)d_inputs, (const Fr*)d_tables, (Fr*)ws, (Fr*)d_permuted_tables_out, (Fr*)(ws + col),
                                     (uint32_t*)(ws + 2 * col), (int*)(ws + 2 * col + fl), L, n, usable));
  return zk_permute_check(ctx, (const int*)(ws + 2 * col + fl));
}

}  // extern "C"
