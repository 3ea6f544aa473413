// Fr number-theoretic transform for gfx950.
//
// Replaces halo2_proofs::arithmetic::best_fft (and, through the coset/scale options, the NTT
// inside poly::domain::EvaluationDomain::{lagrange_to_coeff, coeff_to_lagrange, coeff_to_extended,
// extended_to_coeff}) — halo2_proofs 0.2.0 @ v2023_01_20 [UP], /root/reference/Cargo.lock:469-471;
// SURVEY.md §8(a) rows a3-a6.
//
// The CPU original is a bit-reversal followed by log_n radix-2 rounds over the whole array (log_n
// passes over memory). Here the transform is a 1-, 2- or 3-step Cooley-Tukey decomposition
//     n = n1*n2*n3,  input index i = (i1,i2,i3) row-major, output index j = j1 + n1*j2 + n1*n2*j3
// in which step p does all length-n_p sub-transforms of one digit inside LDS (radix-2 DIF rounds on a
// tile of n_p x C elements; C consecutive elements of the faster digits share the tile so that
// every global access is a run of C*32 contiguous bytes), multiplies by the inter-step twiddle
// omega^(g*i_{p+1}*J_p) on the way out, and the last step writes straight to natural order. The
// vector therefore crosses HBM once per step (2-3 times), not log_n times, and no bit-reversal pass
// exists. Arithmetic is exact, so the result is bit-identical to best_fft's.
//
// Data stay AoS (32-byte Fr, two 16-byte accesses per lane) — the host format — so Rust slices and
// device columns share one layout. Inside a tile the elements are 9 x 29-bit limbs (fp29.cuh): every
// multiplication is data x precomputed constant, the constants (twiddles, coset and output factors) are kept
// in Montgomery radix 2^261, and a butterfly is a lazy sum, a difference and one in-place product.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "fp29.cuh"

using namespace bn254;

namespace {

constexpr int NTT_THREADS = 256;

struct NttPassArgs {
  const Fr* in;
  Fr* out;
  size_t in_col_stride;   // elements between consecutive columns (batch)
  size_t out_col_stride;
  const Fr* tw;           // omega^i, each stored as omega^i * 2^261 (packed canonical): see fp29.cuh "mixed radix"; i < n (tw_full)
                          // or i < n/2 (the second half is the negation of the first)
  uint32_t tw_full;
  uint32_t log_n;
  uint32_t s;             // log2 of this step's sub-transform length n_p
  uint32_t log_c;         // log2 of C (tile = 2^s x C elements)
  uint32_t log_stride;    // log2 of S_p = product of later radices (0 in the last step)
  uint32_t log_prev;      // log2 of N_{p-1} = product of earlier radices
  uint32_t log_n1;        // log2 of n1
  uint32_t log_n2;        // log2 of n2 (3-step only)
  uint32_t log_next;      // log2 of n_{p+1} (non-last steps)
  uint32_t pass;          // 0-based step index
  uint32_t npass;
  uint32_t in_len;        // first step: elements at index >= in_len read as zero (zero padding)
  uint32_t flags;
  Fr in_c[2];             // first step, IN_COSET: element i is multiplied by in_c[i%3 - 1] when i%3 != 0   (radix 2^261)
  Fr out_c[3];            // last step, OUT_MUL: element j is multiplied by out_c[j%3]                       (radix 2^261)
  Fr in_c0;               // first step, IN_ALL: elements with i%3 == 0 are multiplied too (a global input factor)
  // Per-element multiplier tables (radix 2^261, packed canonical), one entry per element of a column: F_IN_TABLE
  // multiplies input element i by in_tab[i] in the first step, F_OUT_TABLE output element j by out_tab[j] in the last.
  // blockIdx.z selects one of several transforms of the SAME input column with different tables (the cosets of
  // the quotient domain, poly.hip): tables and outputs advance by *_z_stride per z, tables by tab_col_stride per column.
  const Fr* in_tab;
  const Fr* out_tab;
  size_t tab_col_stride, tab_z_stride;
  size_t in_z_stride, out_z_stride;
};

constexpr uint32_t F_IN_COSET = 1u;
constexpr uint32_t F_OUT_MUL = 2u;
constexpr uint32_t F_IN_ALL = 4u;
constexpr uint32_t F_IN_TABLE = 8u;
constexpr uint32_t F_OUT_TABLE = 16u;

__device__ __forceinline__ uint32_t brev(uint32_t x, uint32_t bits) {
  return bits == 0 ? 0u : (__brev(x) >> (32 - bits));
}

__device__ __forceinline__ Fr ld_fr(const Fr* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
__device__ __forceinline__ void st_fr(Fr* p, const Fr& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// omega^e for e in [0, n). Domains up to 2^18 keep the whole table (8 MiB at most: the inter-step multiplier is then one
// load); larger ones the first half, the second half being its negation (a 9-limb subtraction and a select per element).
constexpr uint32_t TW_FULL_MAX_LOG_N = 18;
__device__ __forceinline__ Fr tw_lookup(const Fr* tw, uint32_t e, uint32_t log_n, uint32_t full) {
  if (full) return ld_fr(tw + e);  // (wave-uniform: a kernel argument)
  uint32_t half = 1u << (log_n - 1);
  Fr w = ld_fr(tw + (e & (half - 1)));
  return (e & half) ? neg(w) : w;
}

// The tile lives in LDS as 9 x 29-bit limbs (36 B per element; the odd word stride is bank-conflict free),
// not as packed 32-byte values: a butterfly is then a lazy limb-wise sum, a difference plus 10p, and ONE
// in-place product — no unpack/pack and no carry-chain add around it. Bounds, in units of p: elements
// enter below 2 (canonical, or a coset product); a sum doubles the bound, a twiddle product resets it
// below 2; every third round the sums are brought back below 2 with f29_reduce_weak (~30 instructions), so
// no element exceeds 16.8 and every subtrahend stays below 9 (the range f29_sub10 covers). Elements are
// packed to canonical 32-byte values only when they leave the tile.
// Element e of the tile sits at slot e ^ g((e >> 5) & 3), g(t) = t replicated over the low five bits. The radix-4 blocks
// read and write, per element of a group, the elements whose index is the lane number with two zero bits inserted at
// bit p = (a - 1) + log C; for p < 5 the 32 lanes of a half-wavefront would otherwise reach only 8 of the 32 banks (the
// limb stride, 9 words, is odd, so banks follow the slot number). XOR-ing a copy of index bits 5-6 — exactly the lane bits
// the insertion pushed out of the low five — into every bit pair fills the two zero bits again, whatever p is, and is a
// bijection on the tile (identity for tiles below 128 elements); contiguous accesses (the load phase) stay conflict-free.
// Measured (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, tools/ntt_microbench.py): 34 % -> 26 % of the LDS-active cycles in
// the last step, an unchanged 45 % in the first (whose conflicts are the bit-reversed read of its store phase), and
// 0.6 % of the kernels' time: the transform does not wait for LDS (SQ_WAIT_INST_LDS is 4 % of its wave cycles).
__device__ __forceinline__ uint32_t tile_slot(uint32_t e) {
  const uint32_t t = (e >> 5) & 3u;
  return e ^ (t | (t << 2) | ((t & 1u) << 4));
}
__device__ __forceinline__ Fr29 lds_ld(const uint32_t* L, uint32_t e) {
  Fr29 r;
  const uint32_t s = tile_slot(e) * 9;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = L[s + i];
  return r;
}
__device__ __forceinline__ void lds_st(uint32_t* L, uint32_t e, const Fr29& v) {
  const uint32_t s = tile_slot(e) * 9;
#pragma unroll
  for (int i = 0; i < 9; i++) L[s + i] = v.l[i];
}
// the staged sub-transform twiddles: plain indexing
__device__ __forceinline__ Fr29 tw_ld(const uint32_t* T, uint32_t i) {
  Fr29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.l[k] = T[i * 9 + k];
  return r;
}
__device__ __forceinline__ void tw_st(uint32_t* T, uint32_t i, const Fr29& v) {
#pragma unroll
  for (int k = 0; k < 9; k++) T[i * 9 + k] = v.l[k];
}
__device__ __forceinline__ Fr pack_out(const Fr29& v_below_2p) { return f29_pack_canonical<FrP>(v_below_2p); }
// between two steps (the workspace only the next step reads): the value as it is, below 2p < 2^255 — the next step's
// bounds start from "below 2p" anyway, and the subtraction and select of the canonical form are ~45 instructions
__device__ __forceinline__ Fr pack_between(const Fr29& v_below_2p) { return f29_pack_raw<FrP>(v_below_2p); }

template <bool LAST>
__global__ __launch_bounds__(NTT_THREADS) void ntt_step_kernel(NttPassArgs a) {
  const uint32_t nthreads = blockDim.x;  // a quarter of the tile (one radix-4 group per thread and block), at least 64
  extern __shared__ uint4 lds_raw[];
  uint32_t* L = reinterpret_cast<uint32_t*>(lds_raw);
  const uint32_t tid = threadIdx.x;
  const uint32_t C = 1u << a.log_c;
  const uint32_t rows = 1u << a.s;
  const uint32_t tile = rows << a.log_c;
  uint32_t* TW = L + (size_t)tile * 9;  // rows/2 sub-transform twiddles omega_{n_p}^i (radix 2^261), as limbs
  const uint32_t tile_id = blockIdx.x;
  const Fr* in = a.in + (size_t)blockIdx.y * a.in_col_stride + (size_t)blockIdx.z * a.in_z_stride;
  Fr* out = a.out + (size_t)blockIdx.y * a.out_col_stride + (size_t)blockIdx.z * a.out_z_stride;
  const size_t tab_off = (size_t)blockIdx.y * a.tab_col_stride + (size_t)blockIdx.z * a.tab_z_stride;
  const bool in_table = a.pass == 0 && (a.flags & F_IN_TABLE);
  const uint32_t log_n = a.log_n;

  // ---- stage the sub-transform twiddles: omega_{n_p}^i = omega^(i * n/n_p)
  for (uint32_t i = tid; i < (rows >> 1); i += nthreads) tw_st(TW, i, fr29_unpack(ld_fr(a.tw + ((size_t)i << (log_n - a.s)))));

  // ---- load the tile (whole elements; consecutive lanes read consecutive elements of a contiguous run)
  size_t in_base, row_stride;
  uint32_t hi = 0, lo_base = 0, j2 = 0, j1_blk = 0;
  if (!LAST) {
    const uint32_t log_lo_blocks = a.log_stride - a.log_c;
    hi = tile_id >> log_lo_blocks;
    lo_base = (tile_id & ((1u << log_lo_blocks) - 1)) << a.log_c;
    in_base = ((size_t)hi << (a.s + a.log_stride)) + lo_base;
    row_stride = (size_t)1 << a.log_stride;
    // element (x, c) at in_base + x*row_stride + c, LDS index x*C + c
    for (uint32_t e = tid; e < tile; e += nthreads) {
      const uint32_t x = e >> a.log_c, c = e & (C - 1);
      const size_t gi = in_base + (size_t)x * row_stride + c;
      Fr v = Fr::zero();
      if (a.pass != 0 || gi < a.in_len) v = ld_fr(in + gi);
      Fr29 xv = fr29_unpack(v);
      if (in_table && gi < a.in_len) xv = f29_mul(xv, fr29_unpack(ld_fr(a.in_tab + tab_off + gi)));
      lds_st(L, e, xv);
    }
  } else {
    // rows of the tile are C consecutive values of j1 (the fastest output digit)
    if (a.npass == 3) {
      j2 = tile_id & ((1u << a.log_n2) - 1);
      j1_blk = tile_id >> a.log_n2;
    } else {
      j1_blk = tile_id;
    }
    row_stride = a.npass == 1 ? 0 : ((size_t)1 << (log_n - a.log_n1));
    in_base = (size_t)(j1_blk << a.log_c) * row_stride + (a.npass == 3 ? ((size_t)j2 << a.s) : 0);
    // element (rr, x) at in_base + rr*row_stride + x, LDS index x*C + rr
    for (uint32_t q = tid; q < tile; q += nthreads) {
      const uint32_t rr = q >> a.s, x = q & (rows - 1);
      const size_t gi = in_base + (size_t)rr * row_stride + x;
      Fr v = Fr::zero();
      if (a.pass != 0 || gi < a.in_len) v = ld_fr(in + gi);
      Fr29 xv = fr29_unpack(v);
      if (in_table && gi < a.in_len) xv = f29_mul(xv, fr29_unpack(ld_fr(a.in_tab + tab_off + gi)));
      lds_st(L, (x << a.log_c) + rr, xv);
    }
  }
  __syncthreads();

  // ---- first step only: move into the coset (distribute_powers_zeta) — whole elements
  if (a.pass == 0 && (a.flags & F_IN_COSET)) {
    const Fr29 c1 = fr29_unpack(a.in_c[0]), c2 = fr29_unpack(a.in_c[1]), c0 = fr29_unpack(a.in_c0);
    for (uint32_t e = tid; e < tile; e += nthreads) {
      uint32_t x = e >> a.log_c, c = e & (C - 1);
      size_t gi = LAST ? (in_base + (size_t)c * row_stride + x) : (in_base + (size_t)x * row_stride + c);
      uint32_t m = (uint32_t)(gi % 3);
      if (gi < a.in_len) {
        if (m != 0) lds_st(L, e, f29_mul(lds_ld(L, e), m == 1 ? c1 : c2));
        else if (a.flags & F_IN_ALL) lds_st(L, e, f29_mul(lds_ld(L, e), c0));
      }
    }
    __syncthreads();
  }

  // ---- DIF rounds over the row dimension, two at a time: a thread takes the four elements x0 + j * 2^(a-1), j < 4, of
  // rounds a and a - 1 (half-lengths 2^a and 2^(a-1)) through both rounds in registers — one LDS round trip and one
  // barrier per TWO rounds, three twiddle loads for four butterflies. Result row r holds output index brev(r).
  // An odd number of rounds starts with a single round. The last block (a = 1) multiplies by ONE constant only: its
  // round-1 twiddles are omega^0 (skipped) and omega_4, round 0 has none — 1 product where the round-by-round code
  // spent 4. Bounds in units of p: every element enters a block below 4 (2 at the start); sums of the first round are
  // below 8 (a valid subtrahend of f29_sub10*) and stay lazy, twiddle products below 2, so the block leaves below 1.0003
  // (x0: the sum of sums, weakly reduced), 2, 4, 2 — and the last block, which has differences without a product behind
  // them, below 18, 16, 24: all inside the range of the product / weak reduction every element leaves the tile through.
  // The arithmetic of a block is fp29.cuh's f29_dif4 / f29_dif4_last (checked on the host with the bounds as hard failures).
  int st = (int)a.s - 1;
  if (a.s & 1) {  // single round, half-length 2^(s-1): butterfly (x, x + h), twiddle omega_{n_p}^x
    const uint32_t h = 1u << st;
    for (uint32_t b = tid; b < (tile >> 1); b += nthreads) {
      const uint32_t c = b & (C - 1), x0 = b >> a.log_c;
      const uint32_t e0 = (x0 << a.log_c) + c, e1 = ((x0 + h) << a.log_c) + c;
      const Fr29 u = lds_ld(L, e0), v = lds_ld(L, e1);
      lds_st(L, e0, f29_add(u, v));
      lds_st(L, e1, st > 0 ? f29_mul(f29_sub10_lazy(u, v), tw_ld(TW, x0)) : f29_sub10(u, v));
    }
    __syncthreads();
    st--;
  }
  for (; st >= 1; st -= 2) {  // rounds st and st - 1
    const uint32_t hq = 1u << (st - 1);  // distance between the four elements of a group
    const uint32_t estep = hq << a.log_c;
    const bool last_block = st == 1;
    for (uint32_t q = tid; q < (tile >> 2); q += nthreads) {
      const uint32_t c = q & (C - 1), gq = q >> a.log_c;
      const uint32_t lo = gq & (hq - 1), hi = gq >> (st - 1);
      const uint32_t e0 = ((((hi << 2) << (st - 1)) | lo) << a.log_c) + c, e1 = e0 + estep, e2 = e1 + estep, e3 = e2 + estep;
      Fr29 x0 = lds_ld(L, e0), x1 = lds_ld(L, e1), x2 = lds_ld(L, e2), x3 = lds_ld(L, e3);
      if (last_block) f29_dif4_last(x0, x1, x2, x3, tw_ld(TW, 1u << (a.s - 2)));  // omega_4
      else f29_dif4(x0, x1, x2, x3, tw_ld(TW, lo << (a.s - 1 - st)), tw_ld(TW, (lo + hq) << (a.s - 1 - st)), tw_ld(TW, lo << (a.s - st)));
      lds_st(L, e0, x0);
      lds_st(L, e1, x1);
      lds_st(L, e2, x2);
      lds_st(L, e3, x3);
    }
    __syncthreads();
  }

  // ---- store: every element leaves through a product (below 2p) or a weak reduction, then is packed
  if (!LAST) {
    // twiddle omega^(g * i_next * (J_prev + N_prev*j)),  g = n / N_{p+1}
    const uint32_t J_prev = hi;  // step 0: hi = 0; step 1 of 3: hi = j1
    const uint32_t log_g = log_n - (a.log_prev + a.s + a.log_next);
    const uint32_t sh_next = a.log_stride - a.log_next;
    for (uint32_t e = tid; e < tile; e += nthreads) {
      uint32_t j = e >> a.log_c, c = e & (C - 1);
      Fr29 x = lds_ld(L, (brev(j, a.s) << a.log_c) + c);
      uint32_t i_next = (lo_base + c) >> sh_next;
      uint32_t Jp = J_prev + (j << a.log_prev);
      uint32_t ex = (i_next * Jp) << log_g;
      x = ex != 0 ? f29_mul(x, fr29_unpack(tw_lookup(a.tw, ex, log_n, a.tw_full))) : f29_reduce_weak(x);
      st_fr(out + in_base + (size_t)j * row_stride + c, pack_between(x));
    }
  } else {
    const size_t out_base = (size_t)(j1_blk << a.log_c) + (a.npass == 3 ? ((size_t)j2 << a.log_n1) : 0);
    const uint32_t log_ostride = log_n - a.s;  // N_{P-1}
    const Fr29 oc0 = fr29_unpack(a.out_c[0]), oc1 = fr29_unpack(a.out_c[1]), oc2 = fr29_unpack(a.out_c[2]);
    for (uint32_t e = tid; e < tile; e += nthreads) {
      uint32_t j = e >> a.log_c, rr = e & (C - 1);
      Fr29 x = lds_ld(L, (brev(j, a.s) << a.log_c) + rr);
      size_t oi = out_base + rr + ((size_t)j << log_ostride);
      if (a.flags & F_OUT_TABLE) {
        x = f29_mul(x, fr29_unpack(ld_fr(a.out_tab + tab_off + oi)));
      } else if (a.flags & F_OUT_MUL) {
        const uint32_t m = (uint32_t)(oi % 3);
        x = f29_mul(x, m == 0 ? oc0 : m == 1 ? oc1 : oc2);
      } else {
        x = f29_reduce_weak(x);
      }
      st_fr(out + oi, pack_out(x));
    }
  }
}

// tw[i] = omega^i for i < count, stored as the packed canonical integer omega^i * 2^261 (the radix the
// butterflies' constant operand is kept in; negation in tw_lookup is radix-independent). Each thread raises
// omega to its chunk start, then walks.
__global__ void twiddle_gen_kernel(Fr* tw, Fr omega, uint32_t count, uint32_t chunk) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t start = (uint64_t)t * chunk;
  if (start >= count) return;
  Fr cur = pow_u64(omega, start);
  uint32_t end = (uint32_t)((start + chunk < count) ? start + chunk : count);
  for (uint32_t i = (uint32_t)start; i < end; i++) {
    st_fr(tw + i, fr29_const_to_r261(cur));
    cur = mul(cur, omega);
  }
}

int get_twiddles(amdzk_ctx* ctx, uint32_t log_n, const uint64_t omega[4], Fr** out) {
  TwiddleKey key;
  key.log_n = log_n;
  for (int i = 0; i < 4; i++) key.w[i] = omega[i];
  auto it = ctx->twiddles.find(key);
  if (it != ctx->twiddles.end()) {
    *out = it->second;
    return AMDZK_OK;
  }
  uint32_t count = log_n == 0 ? 1 : log_n <= TW_FULL_MAX_LOG_N ? (1u << log_n) : (1u << (log_n - 1));
  Fr* d = nullptr;
  ZK_HIP(ctx, hipMalloc((void**)&d, (size_t)count * sizeof(Fr)));
  Fr w;
  memcpy(w.l, omega, 32);
  uint32_t chunk = 64;
  uint32_t threads = (count + chunk - 1) / chunk;
  dim3 grid((threads + 63) / 64), block(64);
  ZK_LAUNCH(ctx, "twiddle_gen", twiddle_gen_kernel, grid, block, 0, d, w, count, chunk);
  ctx->twiddles[key] = d;
  *out = d;
  return AMDZK_OK;
}

// Plan: number of steps and their radices.
struct Plan {
  uint32_t npass;
  uint32_t s[3];
};

uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* v = getenv(name);
  return v ? (uint32_t)atoi(v) : dflt;
}

Plan make_plan(uint32_t log_n, uint32_t tile_log) {
  // Keep at least 4 elements (128 contiguous bytes) per global run: s <= tile_log - 2
  // (AMDZK_NTT_SMAX_SLACK=1 allows 2-element runs when that saves a whole step).
  uint32_t smax = tile_log - 2 + env_u32("AMDZK_NTT_SMAX_SLACK", 1);
  Plan p;
  if (log_n <= tile_log) {  // whole column in one tile
    p.npass = 1;
    p.s[0] = log_n;
    p.s[1] = p.s[2] = 0;
    return p;
  }
  p.npass = log_n <= 2 * smax ? 2 : 3;
  uint32_t rem = log_n;
  for (uint32_t i = 0; i < p.npass; i++) {
    uint32_t left = p.npass - i;
    p.s[i] = (rem + left - 1) / left;
    rem -= p.s[i];
  }
  for (uint32_t i = p.npass; i < 3; i++) p.s[i] = 0;
  return p;
}

}  // namespace

// Generalised entry used by the C ABI and by the domain helpers. Out of place when d_in != d_out.
//   in_len   : number of valid input elements per column (rest read as zero); 0 means n.
//   in_coset : if non-null, 2 constants applied to input element i with i%3 = 1, 2
//   out_mul  : if non-null, 3 constants applied to output element j by j%3
//   tabs     : per-element multiplier tables and the z dimension (NttTables, common.hpp); null = none
int zk_ntt_ex(amdzk_ctx* ctx, const Fr* d_in, size_t in_stride, Fr* d_out, size_t out_stride, uint32_t log_n,
              const uint64_t omega[4], size_t ncols, uint32_t in_len, const Fr* in_coset, const Fr* out_mul, const Fr* in_first,
              const NttTables* tabs) {
  if (log_n > 27) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "ntt: log_n %u > 27", log_n);
  if (ncols == 0) return AMDZK_OK;
  const size_t n = (size_t)1 << log_n;
  if (in_len == 0 || in_len > n) in_len = (uint32_t)n;
  if (ncols > 1 && (out_stride < n || in_stride < in_len)) ZK_FAIL(ctx, AMDZK_E_INVALID, "ntt: column stride too small");
  if (ncols > 65535) ZK_FAIL(ctx, AMDZK_E_INVALID, "ntt: more than 65535 columns in one call");
  Fr* tw = nullptr;
  ZK_TRY(get_twiddles(ctx, log_n, omega, &tw));
  const uint32_t tile_log = env_u32("AMDZK_NTT_TILE_LOG", 10);
  Plan plan = make_plan(log_n, tile_log);

  const uint32_t nz = tabs && tabs->nz ? tabs->nz : 1;
  if (nz > 1 && (in_coset || out_mul)) ZK_FAIL(ctx, AMDZK_E_INVALID, "ntt: the z dimension goes with multiplier tables only");
  if (nz > 65535) ZK_FAIL(ctx, AMDZK_E_INVALID, "ntt: nz too large");
  Fr* ws = nullptr;
  if (plan.npass > 1) ZK_TRY(zk_ws_reserve(ctx, 0, ncols * nz * n * sizeof(Fr), (void**)&ws));

  NttPassArgs a;
  memset(&a, 0, sizeof(a));
  a.tw = tw;
  a.tw_full = log_n >= 1 && log_n <= TW_FULL_MAX_LOG_N;
  a.log_n = log_n;
  a.npass = plan.npass;
  a.log_n1 = plan.s[0];
  a.log_n2 = plan.s[1];
  a.in_len = in_len;
  // callers pass constants in halo2curves' radix-2^256 form; the kernels want c * 2^261 = (32 c) * 2^256
  Fr k32 = Fr::one();
  for (int i = 0; i < 5; i++) k32 = add(k32, k32);
  if (in_coset) {
    a.in_c[0] = mul(in_coset[0], k32);
    a.in_c[1] = mul(in_coset[1], k32);
    if (in_first) a.in_c0 = mul(*in_first, k32);
  }
  if (out_mul) {
    a.out_c[0] = mul(out_mul[0], k32);
    a.out_c[1] = mul(out_mul[1], k32);
    a.out_c[2] = mul(out_mul[2], k32);
  }
  uint32_t log_prev = 0;
  for (uint32_t p = 0; p < plan.npass; p++) {
    const bool last = (p + 1 == plan.npass);
    a.pass = p;
    a.s = plan.s[p];
    a.log_prev = log_prev;
    a.log_stride = log_n - log_prev - a.s;
    a.log_next = last ? 0 : plan.s[p + 1];
    uint32_t lc = a.s >= tile_log ? 0 : tile_log - a.s;
    if (!last && lc > a.log_stride) lc = a.log_stride;
    if (last) {
      if (plan.npass == 1) lc = 0;
      else if (lc > a.log_n1) lc = a.log_n1;
    }
    a.log_c = lc;
    a.flags = 0;
    if (p == 0 && in_coset) a.flags |= F_IN_COSET | (in_first ? F_IN_ALL : 0u);
    if (last && out_mul) a.flags |= F_OUT_MUL;
    // first step reads the caller's input, last step writes the caller's output, the workspace
    // carries the intermediate layout. A single-step transform covers a column with one tile (all
    // loads precede all stores), so it may run in place.
    // the workspace holds [column][z][n]; the caller's input is shared by all z, its output advances by out_z_stride
    a.in = (p == 0) ? d_in : ws;
    a.in_col_stride = (p == 0) ? in_stride : (size_t)nz * n;
    a.in_z_stride = (p == 0) ? 0 : n;
    a.out = last ? d_out : ws;
    a.out_col_stride = last ? out_stride : (size_t)nz * n;
    a.out_z_stride = last ? (tabs ? tabs->out_z_stride : 0) : n;
    if (tabs) {
      a.in_tab = tabs->in_tab;
      a.out_tab = tabs->out_tab;
      a.tab_col_stride = tabs->tab_col_stride;
      a.tab_z_stride = tabs->tab_z_stride;
      if (p == 0 && tabs->in_tab) a.flags |= F_IN_TABLE;
      if (last && tabs->out_tab) a.flags |= F_OUT_TABLE;
    }
    const uint32_t tile_elems_log = a.s + a.log_c;
    uint32_t threads = (uint32_t)1 << (tile_elems_log > 2 ? tile_elems_log - 2 : 0);
    threads = threads < 64 ? 64 : threads > (uint32_t)NTT_THREADS ? (uint32_t)NTT_THREADS : threads;
    dim3 grid((uint32_t)(n >> tile_elems_log), (uint32_t)ncols, nz), block(threads);
    const size_t tile_elems = (size_t)1 << tile_elems_log;
    size_t shmem = (tile_elems + ((size_t)1 << a.s) / 2 + 1) * 9 * sizeof(uint32_t);  // limbs, see ntt_step_kernel
    shmem += env_u32(last ? "AMDZK_NTT_LDS_PAD_LAST" : "AMDZK_NTT_LDS_PAD_FIRST", 0);  // occupancy experiments
    if (last) {
      if (shmem > 65536) ZK_HIP(ctx, hipFuncSetAttribute((const void*)ntt_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
      ZK_LAUNCH(ctx, "ntt_step_last", ntt_step_kernel<true>, grid, block, shmem, a);
    } else {
      if (shmem > 65536) ZK_HIP(ctx, hipFuncSetAttribute((const void*)ntt_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
      ZK_LAUNCH(ctx, "ntt_step", ntt_step_kernel<false>, grid, block, shmem, a);
    }
    log_prev += a.s;
  }
  return AMDZK_OK;
}

Fr zk_fr_inv_pow2(uint32_t log_n) {
  Fr two = add(Fr::one(), Fr::one());
  return inv(pow_u64(two, log_n));
}

int zk_ntt_dev(amdzk_ctx* ctx, Fr* d_a, uint32_t log_n, const uint64_t omega[4], uint32_t flags,
               size_t ncols, size_t col_stride) {
  if (flags & AMDZK_NTT_SCALE_NINV) {
    Fr ninv = zk_fr_inv_pow2(log_n);
    Fr oc[3] = {ninv, ninv, ninv};
    return zk_ntt_ex(ctx, d_a, col_stride, d_a, col_stride, log_n, omega, ncols, 0, nullptr, oc, nullptr, nullptr);
  }
  return zk_ntt_ex(ctx, d_a, col_stride, d_a, col_stride, log_n, omega, ncols, 0, nullptr, nullptr, nullptr, nullptr);
}
