// BN254 G1 multi-scalar multiplication for gfx950.
//
// Replaces halo2_proofs::arithmetic::best_multiexp as reached from
// poly::kzg::commitment::ParamsKZG::{commit, commit_lagrange}
// (halo2_proofs 0.2.0 @ v2023_01_20 [UP], /root/reference/Cargo.lock:469-471; SURVEY.md §8(a) a1,a2).
//
// The CPU original is Pippenger per thread-chunk: c = ceil(ln n) bit windows, 256/c+1 segments,
// 2^c-1 buckets per segment, a running-sum fold per segment and c doublings between segments.
// Same mathematics, different shape here, chosen for this chip:
//
//  * The bases of a ParamsKZG never change, and HBM is 288 GB: at upload time every base is expanded
//    into its W window multiples T[w][i] = 2^(c*w) * g[i] (affine, 64 B). A scalar's W signed digits
//    then all fall into ONE bucket set of 2^(c-1) buckets per column — no per-window bucket sets, no
//    doublings, and the bucket fold runs once per column instead of once per window.
//  * Bucket membership comes from a counting sort of (bucket, point-id) pairs (histogram -> scan ->
//    scatter), so accumulation is gather-only and needs no locks.
//  * Accumulation is cut into equal-sized tasks over the sorted list (not one thread per bucket):
//    skewed witness columns (mostly 0/1/small limbs) and uniform scalars load the chip equally.
//    Level 1 adds affine points into XYZZ accumulators (8M+2S per add), levels 2-3 fold the
//    per-task partial sums.
//  * All of this runs on fp29.cuh's 9 x 29-bit limbs in Montgomery radix 2^261 (in-place 64-bit column
//    accumulation, lazy reduction: 1.7x the 32-bit product on this chip). The window tables store that
//    radix, partial sums cross HBM as raw limbs, and only the one result per column is converted back.
//  * The weighted fold sum_b (b+1)*B_b is done without a serial running sum: with b = r + 64*g,
//    sum = sum_r r*C_r + 64*sum_g g*T_g + sum_g T_g where C_r / T_g are plain column / row sums
//    (lane-per-output serial sums: the prover is bound by instruction issue, and a shuffle tree spends six
//    wave-level additions where a lane loop spends one), then three short reductions per column.
//  * Many columns over the same bases (all advice columns of a phase) go through every kernel in
//    one launch: grid.y = column.
//
// No MFMA: this is 254-bit modular integer work (v_mad_u64_u32), and it is bound by VALU instruction issue, not by HBM.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "fp29.cuh"

using namespace bn254;

struct amdzk_srs {
  uint32_t k = 0;
  uint32_t c = 0;        // window bits
  uint32_t W = 0;        // windows = ceil(255 / c)
  size_t n = 0;
  // [basis] -> W x n affine points, window-major: T[w][i] = 2^(c*w) * bases[i]. Coordinates are packed
  // canonical integers in Montgomery radix 2^261 (fp29.cuh) — the form the level-1 accumulation kernel
  // multiplies in; (0, 0) is still the identity.
  G1Affine* table[2] = {nullptr, nullptr};
  // [basis] -> the n bases themselves in halo2curves' radix-2^256 form (ParamsKZG::get_g, write, downsize)
  G1Affine* base[2] = {nullptr, nullptr};
};

namespace {

constexpr int MSM_THREADS = 256;

__device__ __forceinline__ uint4 ldg4(const void* p) { return *reinterpret_cast<const uint4*>(p); }

__device__ __forceinline__ Fq ld_fq(const Fq* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fq r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
__device__ __forceinline__ void st_fq(Fq* p, const Fq& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
__device__ __forceinline__ G1Affine ld_aff(const G1Affine* p) {
  G1Affine r;
  r.x = ld_fq(&p->x);
  r.y = ld_fq(&p->y);
  return r;
}
__device__ __forceinline__ void st_aff(G1Affine* p, const G1Affine& v) {
  st_fq(&p->x, v.x);
  st_fq(&p->y, v.y);
}
__device__ __forceinline__ G1X ld_x(const G1X* p) {
  G1X r;
  r.x = ld_fq(&p->x);
  r.y = ld_fq(&p->y);
  r.zz = ld_fq(&p->zz);
  r.zzz = ld_fq(&p->zzz);
  return r;
}
__device__ __forceinline__ void st_x(G1X* p, const G1X& v) {
  st_fq(&p->x, v.x);
  st_fq(&p->y, v.y);
  st_fq(&p->zz, v.zz);
  st_fq(&p->zzz, v.zzz);
}

// ------------------------------------------------------------------ window tables
// next[i] = 2^c * prev[i], affine in, affine out (one Fermat inversion per point; upload-time only).
__global__ void table_next_kernel(const G1Affine* prev, G1Affine* next, size_t n, uint32_t c) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine p = ld_aff(prev + i);
  G1X x = x_dbl_affine(p);
  for (uint32_t k = 1; k < c; k++) x = x_dbl(x);
  st_aff(next + i, x_to_affine(x));
}

// In place: radix-2^256 Montgomery coordinates -> radix-2^261 (one product per coordinate; 0 stays 0).
__global__ void table_to_r261_kernel(G1Affine* t, size_t count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  G1Affine p = ld_aff(t + i);
  p.x = fq29_pack_canonical(fq29_from_r256(p.x));
  p.y = fq29_pack_canonical(fq29_from_r256(p.y));
  st_aff(t + i, p);
}

// ------------------------------------------------------------------ digits
// Signed base-2^C digits of the canonical scalar: d_w in [-(2^(C-1)-1), 2^(C-1)], sum d_w 2^(Cw) = s.
// f(w, bucket, negative) is called for every non-zero digit; bucket = |d|-1.
template <int C, class Fn>
__device__ __forceinline__ void for_each_digit(const Fr& canon, Fn f) {
  constexpr int W = (255 + C - 1) / C;
  uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < W; w++) {
    constexpr uint32_t mask = (1u << C) - 1u;
    const int o = C * w, limb = o >> 5, sh = o & 31;
    uint32_t v = 0;
    if (limb < 8) {
      v = canon.l[limb] >> sh;
      if (sh + C > 32 && limb + 1 < 8) v |= canon.l[limb + 1] << (32 - sh);
    }
    v = (v & mask) + carry;
    carry = v > (1u << (C - 1)) ? 1u : 0u;
    uint32_t mag = carry ? ((1u << C) - v) : v;
    if (mag != 0) f((uint32_t)w, mag - 1, carry);
  }
}

struct DigitArgs {
  const Fr* scalars;     // column 0
  size_t col_stride;     // elements
  uint32_t len;
  uint32_t nb;           // buckets per column = 2^(C-1)
  uint32_t chunk;        // scalars per workgroup
  uint32_t nblk;         // workgroups per column
  uint32_t* blk_hist;    // [ncols][nblk][nb]: per-workgroup counts, later per-workgroup start offsets
  const uint32_t* off0;  // [ncols][nb+1] bucket offsets (scatter only)
  uint32_t* entries;     // [ncols][ecap] (scatter only)
  size_t ecap;
  uint32_t table_n;      // row length of the window table (2^k)
};

// One workgroup owns `chunk` consecutive scalars of one column. Counting and cursor bumping happen in
// LDS (ds_add / ds_add_rtn), so global memory sees one coalesced histogram write per workgroup
// instead of one atomic per digit.
// BIG: 1024 threads per workgroup instead of 256 — for the few-columns / long-column case (k >= 19: one 2^22-point
// column is 64 workgroups of 256 threads otherwise, a quarter of the chip's SIMDs with one wavefront each, every thread a
// dependent chain of load -> Montgomery reduction -> 16 LDS atomics -> 16 scattered stores: 1.96 + 0.43 ms of the 9.9 ms
// of a 2^22-point MSM, profiles/r04a_*). The counters of a workgroup fill LDS (2^15 buckets = 128 KiB), so a compute unit
// holds one workgroup whatever its size: sixteen wavefronts hide that chain where four do not.
template <int C, bool SCATTER, bool BIG>
__global__ __launch_bounds__(BIG ? 1024 : MSM_THREADS) void msm_digit_kernel(DigitArgs a) {
  extern __shared__ uint32_t lds_cnt[];  // nb counters / cursors
  constexpr uint32_t THREADS = BIG ? 1024 : MSM_THREADS;
  const uint32_t col = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
  uint32_t* gh = a.blk_hist + ((size_t)col * a.nblk + blk) * a.nb;
  if (!SCATTER) {
    for (uint32_t b = t; b < a.nb; b += THREADS) lds_cnt[b] = 0;
  } else {
    const uint32_t* off = a.off0 + (size_t)col * (a.nb + 1);
    for (uint32_t b = t; b < a.nb; b += THREADS) lds_cnt[b] = off[b] + gh[b];
  }
  __syncthreads();
  const uint32_t lo_i = blk * a.chunk, hi_i = min(lo_i + a.chunk, a.len);
  uint32_t* ent = a.entries + (size_t)col * a.ecap;
  for (uint32_t i = lo_i + t; i < hi_i; i += THREADS) {
    const uint4* sp = reinterpret_cast<const uint4*>(a.scalars + (size_t)col * a.col_stride + i);
    uint4 lo = sp[0], hi = sp[1];
    if ((lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) == 0) continue;
    Fr s;
    s.l[0] = lo.x; s.l[1] = lo.y; s.l[2] = lo.z; s.l[3] = lo.w;
    s.l[4] = hi.x; s.l[5] = hi.y; s.l[6] = hi.z; s.l[7] = hi.w;
    Fr canon = fr29_from_mont(s);  // = to_repr() of the original
    if (!SCATTER) {
      for_each_digit<C>(canon, [&](uint32_t, uint32_t b, uint32_t) { atomicAdd(&lds_cnt[b], 1u); });
    } else {
      for_each_digit<C>(canon, [&](uint32_t w, uint32_t b, uint32_t negv) {
        uint32_t pos = atomicAdd(&lds_cnt[b], 1u);
        ent[pos] = (w * a.table_n + i) | (negv << 31);
      });
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (uint32_t b = t; b < a.nb; b += THREADS) gh[b] = lds_cnt[b];
  }
}

// Per bucket: turn the per-workgroup counts into per-workgroup start offsets (relative to the
// bucket's own start) and emit the bucket total.
__global__ __launch_bounds__(256) void msm_blk_offsets_kernel(uint32_t* blk_hist, uint32_t* cnt, uint32_t nb, uint32_t nblk) {
  const uint32_t col = blockIdx.y, b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  uint32_t* h = blk_hist + (size_t)col * nblk * nb + b;
  uint32_t run = 0;
  for (uint32_t k = 0; k < nblk; k++) {
    uint32_t v = h[(size_t)k * nb];
    h[(size_t)k * nb] = run;
    run += v;
  }
  cnt[(size_t)col * nb + b] = run;
}

// ------------------------------------------------------------------ scan
// off_out[b] = exclusive prefix sum over b of v[b]; off_out[nb] = total.
//   mode 1: v[b] = src[b]                       (src has nb entries per column)
//   mode 0: v[b] = ceil((src[b+1]-src[b]) / T)   (src has nb+1 entries per column)
//   mode 2: v[b] = number of T-aligned chunks of the whole list that intersect [src[b], src[b+1])
// The source row is first copied into LDS with coalesced loads (thread t takes words t, t + 1024, ...: all of a thread's
// loads in flight at once) — a thread walking its own `per` consecutive words straight from global memory waits for one
// dependent load after the other: 62 us per launch at 2^15 buckets, 15 us staged (profiles/r04a_*). Rows of up to
// SCAN_LDS_WORDS words (buckets + 1) are staged; longer ones are read in place.
constexpr uint32_t SCAN_LDS_WORDS = 32768 + 1;
__global__ __launch_bounds__(1024) void msm_scan_kernel(const uint32_t* src, uint32_t* off_out,
                                                         uint32_t nb, uint32_t T, int src_is_hist) {
  extern __shared__ uint32_t scan_lds[];  // [16] wavefront totals, then the staged row
  uint32_t* wtot = scan_lds;
  const uint32_t col = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const uint32_t words = src_is_hist == 1 ? nb : nb + 1;
  const uint32_t* s = src + (size_t)col * words;
  if (words <= SCAN_LDS_WORDS) {
    uint32_t* st = scan_lds + 16;
    for (uint32_t i = t; i < words; i += 1024) st[i] = s[i];
    __syncthreads();
    s = st;
  }
  uint32_t* o = off_out + (size_t)col * (nb + 1);
  const uint32_t per = (nb + 1023) / 1024;
  const uint32_t b0 = t * per, b1 = min(b0 + per, nb);
  auto val = [&](uint32_t b) -> uint32_t {
    if (src_is_hist == 1) return s[b];
    if (src_is_hist == 0) return (s[b + 1] - s[b] + T - 1) / T;
    return s[b + 1] > s[b] ? (s[b + 1] - 1) / T - s[b] / T + 1 : 0u;
  };
  uint32_t sum = 0;
  for (uint32_t b = b0; b < b1; b++) sum += val(b);
  // inclusive scan of the 1024 thread sums: shuffles inside a wavefront, the 16 wavefront totals through LDS (two
  // barriers; the Hillis-Steele scan over LDS this replaces took twenty, and the kernel is pure latency: four or five
  // launches of one workgroup per column on the chain of every commitment batch)
  uint32_t inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t v = __shfl_up(inc, d, 64);
    if ((int)lane >= d) inc += v;
  }
  if (lane == 63) wtot[wv] = inc;
  __syncthreads();
  if (wv == 0) {
    uint32_t w = lane < 16 ? wtot[lane] : 0u;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const uint32_t v = __shfl_up(w, d, 64);
      if ((int)lane >= d) w += v;
    }
    if (lane < 16) wtot[lane] = w;  // inclusive totals
  }
  __syncthreads();
  uint32_t run = inc - sum + (wv ? wtot[wv - 1] : 0u);
  for (uint32_t b = b0; b < b1; b++) {  // stores do not wait for each other
    o[b] = run;
    run += val(b);
  }
  if (t == 1023) o[nb] = wtot[15];
}

// ------------------------------------------------------------------ accumulation levels
// From the table gather to the last fold everything runs on fp29.cuh's 9 x 29-bit limbs in Montgomery radix
// 2^261. Partial sums cross HBM as raw G1X29 (36 words, 144 B): limbs normalised, values only lazily
// reduced (x < 9p, y < 5p, zz, zzz < 2p) — no conversion or reduction on the way out or in. Only the one
// result per column is converted to the packed radix-2^256 XYZZ that zk_msm_finish and the callers read.
__device__ __forceinline__ G1X29 ld_x29(const G1X29* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint32_t w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint4 v = q[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  G1X29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.x.l[i] = w[i];
    r.y.l[i] = w[9 + i];
    r.zz.l[i] = w[18 + i];
    r.zzz.l[i] = w[27 + i];
  }
  return r;
}
__device__ __forceinline__ void st_x29(G1X29* p, const G1X29& v) {
  uint32_t w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    w[i] = v.x.l[i];
    w[9 + i] = v.y.l[i];
    w[18 + i] = v.zz.l[i];
    w[27 + i] = v.zzz.l[i];
  }
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

struct AccArgs {
  const uint32_t* off_in;   // [ncols][nb+1] offsets of the input list per bucket
  const uint32_t* off_out;  // [ncols][nb+1] task offsets per bucket (ceil(cnt/T))
  uint32_t nb;
  uint32_t T;
  const uint32_t* entries;  // level 1 input
  size_t ecap;
  const G1Affine* table;
  const G1X29* in_list;     // level >= 2 input
  size_t in_cap;
  G1X29* out_list;
  size_t out_cap;
};

// Balanced accumulation level (FIRST: table points by id; otherwise partial sums of the previous
// level): thread t owns entries [t*T, (t+1)*T) of the column's sorted list, whatever
// buckets they belong to, and emits one partial sum per bucket segment it crosses (slot
// off_out[b] + (t - off_in[b]/T)). Every lane of a wavefront performs the same number of additions,
// so skewed witness columns (thousands of tiny buckets next to a few huge ones) no longer leave most
// lanes idle behind the longest task. FIRST is the hot kernel of the whole prover: a mixed addition (madd-2008-s on
// 29-bit limbs) is 6 products, 2 squares and one two-product reduction (y3 = r (q - x3) - y1 ppp, f29_mul2) — 1,539
// multiply-adds — on a window-table point (packed canonical, radix 2^261).
template <bool FIRST>
__global__ __launch_bounds__(MSM_THREADS) void msm_accum_seg_kernel(AccArgs a) {
  const uint32_t col = blockIdx.y;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t* off_in = a.off_in + (size_t)col * (a.nb + 1);
  const uint32_t* off_out = a.off_out + (size_t)col * (a.nb + 1);
  const uint32_t total = off_in[a.nb];
  const uint32_t start = t * a.T;
  if (start >= total) return;
  const uint32_t end = min(start + a.T, total);
  uint32_t lo = 0, hi = a.nb;  // largest b with off_in[b] <= start: the non-empty bucket holding `start`
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (off_in[mid] <= start) lo = mid; else hi = mid;
  }
  uint32_t b = lo, b_end = off_in[b + 1];
  const uint32_t* ent = FIRST ? a.entries + (size_t)col * a.ecap : nullptr;
  const G1X29* in = FIRST ? nullptr : a.in_list + (size_t)col * a.in_cap;
  G1X29* out = a.out_list + (size_t)col * a.out_cap;
  G1X29 acc = G1X29::inf();
  for (uint32_t e = start; e < end; e++) {
    if (e >= b_end) {  // crossed into the next non-empty bucket: flush
      st_x29(out + off_out[b] + (t - off_in[b] / a.T), acc);
      acc = G1X29::inf();
      do {
        b++;
        b_end = off_in[b + 1];
      } while (e >= b_end);
    }
    if (FIRST) {
      const uint32_t id = ent[e];
      G1Affine p = ld_aff(a.table + (id & 0x7fffffffu));
      const bool p_inf = p.is_inf();
      if (id >> 31) p.y = neg(p.y);  // negating a canonical value does not depend on the Montgomery radix
      acc = x29_add_affine(acc, fq29_unpack(p.x), fq29_unpack(p.y), p_inf);
    } else {
      acc = x29_add(acc, ld_x29(in + e));
    }
  }
  st_x29(out + off_out[b] + (t - off_in[b] / a.T), acc);
}

// Level 1 as a PERSISTENT grid (AMDZK_L1_LDS=9; round-4 experiment, profiles/r04b_*): a fixed number of workgroups, each
// fetching (column, block) items from an atomic counter — items are numbered block-major, so the blocks that hold entries
// come first whatever the column — instead of one workgroup per 256 * T entries of CAPACITY: the grid of the generic
// kernel is sized on len * W entries per column, and a witness column (5-6 non-zero digits per scalar of 20) leaves 70 %
// of its workgroups with nothing to do. Same arithmetic, same slots, same results.
__global__ __launch_bounds__(MSM_THREADS) void msm_accum_l1_persist_kernel(AccArgs a, uint32_t* counter, uint32_t blocks_x, uint32_t ncols) {
  __shared__ uint32_t s_item;
  for (;;) {
    if (threadIdx.x == 0) s_item = atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t item = s_item;
    __syncthreads();
    if (item >= blocks_x * ncols) return;
    const uint32_t col = item % ncols, bx = item / ncols;
    const uint32_t t = bx * blockDim.x + threadIdx.x;
    const uint32_t* off_in = a.off_in + (size_t)col * (a.nb + 1);
    const uint32_t* off_out = a.off_out + (size_t)col * (a.nb + 1);
    const uint32_t total = off_in[a.nb];
    const uint32_t start = t * a.T;
    if (start >= total) continue;
    const uint32_t end = min(start + a.T, total);
    uint32_t lo = 0, hi = a.nb;
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (off_in[mid] <= start) lo = mid; else hi = mid;
    }
    uint32_t b = lo, b_end = off_in[b + 1];
    const uint32_t* ent = a.entries + (size_t)col * a.ecap;
    G1X29* out = a.out_list + (size_t)col * a.out_cap;
    G1X29 acc = G1X29::inf();
    for (uint32_t e = start; e < end; e++) {
      if (e >= b_end) {
        st_x29(out + off_out[b] + (t - off_in[b] / a.T), acc);
        acc = G1X29::inf();
        do {
          b++;
          b_end = off_in[b + 1];
        } while (e >= b_end);
      }
      const uint32_t id = ent[e];
      G1Affine p = ld_aff(a.table + (id & 0x7fffffffu));
      const bool p_inf = p.is_inf();
      if (id >> 31) p.y = neg(p.y);
      acc = x29_add_affine(acc, fq29_unpack(p.x), fq29_unpack(p.y), p_inf);
    }
    st_x29(out + off_out[b] + (t - off_in[b] / a.T), acc);
  }
}

// Level 1 with the XYZZ accumulator in LDS (AMDZK_L1_LDS=1; round-4 experiment, profiles/r04b_*): the generic kernel above
// keeps 36 accumulator limbs + 18 limbs of the table point + the temporaries of the addition in registers — 152 VGPRs,
// three wavefronts per SIMD. Here the accumulator lives in LDS as [word][thread] (conflict-free 32-bit accesses, 36 KiB
// per 256-thread workgroup, four workgroups per compute unit) and each coordinate is read where the formula needs it
// and written back when its new value exists: 12 coordinate moves (108 LDS words) per 2,160-instruction addition.
constexpr int L1L_THREADS = 256;
struct LdsAcc {
  uint32_t* base;  // &lds[threadIdx.x]; word w of this thread's accumulator at base[w * L1L_THREADS]
  __device__ __forceinline__ Fq29 ld(int f) const {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = base[(f * 9 + i) * L1L_THREADS];
    return r;
  }
  __device__ __forceinline__ void st(int f, const Fq29& v) const {
#pragma unroll
    for (int i = 0; i < 9; i++) base[(f * 9 + i) * L1L_THREADS] = v.l[i];
  }
  __device__ __forceinline__ G1X29 all() const {
    G1X29 r;
    r.x = ld(0); r.y = ld(1); r.zz = ld(2); r.zzz = ld(3);
    return r;
  }
};
template <int WAVES>
__global__ __launch_bounds__(L1L_THREADS, WAVES) void msm_accum_l1_lds_kernel(AccArgs a) {
  __shared__ uint32_t lds[36 * L1L_THREADS];
  const uint32_t col = blockIdx.y;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t* off_in = a.off_in + (size_t)col * (a.nb + 1);
  const uint32_t* off_out = a.off_out + (size_t)col * (a.nb + 1);
  const uint32_t total = off_in[a.nb];
  const uint32_t start = t * a.T;
  if (start >= total) return;
  const uint32_t end = min(start + a.T, total);
  uint32_t lo = 0, hi = a.nb;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (off_in[mid] <= start) lo = mid; else hi = mid;
  }
  uint32_t b = lo, b_end = off_in[b + 1];
  const uint32_t* ent = a.entries + (size_t)col * a.ecap;
  G1X29* out = a.out_list + (size_t)col * a.out_cap;
  const LdsAcc A{lds + threadIdx.x};
  bool acc_inf = true;
  for (uint32_t e = start; e < end; e++) {
    if (e >= b_end) {
      st_x29(out + off_out[b] + (t - off_in[b] / a.T), acc_inf ? G1X29::inf() : A.all());
      acc_inf = true;
      do {
        b++;
        b_end = off_in[b + 1];
      } while (e >= b_end);
    }
    const uint32_t id = ent[e];
    G1Affine p = ld_aff(a.table + (id & 0x7fffffffu));
    if (p.is_inf()) continue;
    if (id >> 31) p.y = neg(p.y);
    const Fq29 qx = fq29_unpack(p.x), qy = fq29_unpack(p.y);
    if (acc_inf) {
      A.st(0, qx);
      A.st(1, qy);
      A.st(2, f29_one<Fq29P>());
      A.st(3, f29_one<Fq29P>());
      acc_inf = false;
      continue;
    }
    const Fq29 u2 = f29_mul(qx, A.ld(2));
    const Fq29 s2 = f29_mul(qy, A.ld(3));
    const Fq29 pd = f29_sub10(u2, A.ld(0));
    const Fq29 r = f29_sub6(s2, A.ld(1));
    const Fq29 pp = f29_sqr(pd);
    const Fq29 rr = f29_sqr(r);
    if (f29_is_zero_mod_p(pp)) {
      if (f29_is_zero_mod_p(rr)) {
        const G1X29 d = x29_dbl_affine(qx, qy);
        A.st(0, d.x); A.st(1, d.y); A.st(2, d.zz); A.st(3, d.zzz);
      } else {
        acc_inf = true;
      }
      continue;
    }
    const Fq29 ppp = f29_mul(pd, pp);
    const Fq29 q = f29_mul(A.ld(0), pp);
    const Fq29 s = f29_add(ppp, f29_add_lazy(q, q));
    const Fq29 ox = f29_sub7(rr, s);
    A.st(0, ox);
    const Fq29 tq = f29_sub10_lazy(q, ox);
    A.st(1, f29_mul2(r, tq, f29_neg6(A.ld(1)), ppp));
    A.st(2, f29_mul(A.ld(2), pp));
    A.st(3, f29_mul(A.ld(3), ppp));
  }
  st_x29(out + off_out[b] + (t - off_in[b] / a.T), acc_inf ? G1X29::inf() : A.all());
}

// ------------------------------------------------------------------ wavefront reductions
__device__ __forceinline__ G1X29 shfl_xor_x29(const G1X29& v, int m) {
  G1X29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.x.l[i] = __shfl_xor(v.x.l[i], m, 64);
    r.y.l[i] = __shfl_xor(v.y.l[i], m, 64);
    r.zz.l[i] = __shfl_xor(v.zz.l[i], m, 64);
    r.zzz.l[i] = __shfl_xor(v.zzz.l[i], m, 64);
  }
  return r;
}
// All 64 lanes end with the sum of the 64 inputs.
__device__ __forceinline__ G1X29 wave_sum29(G1X29 v) {
#pragma unroll 1
  for (int m = 32; m >= 1; m >>= 1) v = x29_add(v, shfl_xor_x29(v, m));
  return v;
}
__device__ __forceinline__ G1X29 shfl_down_x29(const G1X29& v, int d) {
  G1X29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.x.l[i] = __shfl_down(v.x.l[i], d, 64);
    r.y.l[i] = __shfl_down(v.y.l[i], d, 64);
    r.zz.l[i] = __shfl_down(v.zz.l[i], d, 64);
    r.zzz.l[i] = __shfl_down(v.zzz.l[i], d, 64);
  }
  return r;
}
// Lane l ends with the sum of the inputs of lanes l .. 63 (six shuffle rounds).
__device__ __forceinline__ G1X29 wave_suffix29(G1X29 v, uint32_t lane) {
#pragma unroll 1
  for (int d = 1; d < 64; d <<= 1) {
    const G1X29 o = shfl_down_x29(v, d);
    v = x29_add(v, lane + d < 64 ? o : G1X29::inf());
  }
  return v;
}
// The same over the first `n` lanes only (n a power of two <= 64; the other lanes must hold the identity).
__device__ __forceinline__ G1X29 wave_sum29_n(G1X29 v, int n) {
#pragma unroll 1
  for (int m = n >> 1; m >= 1; m >>= 1) v = x29_add(v, shfl_xor_x29(v, m));
  return v;
}
__device__ __forceinline__ G1X29 wave_suffix29_n(G1X29 v, uint32_t lane, int n) {
#pragma unroll 1
  for (int d = 1; d < n; d <<= 1) {
    const G1X29 o = shfl_down_x29(v, d);
    v = x29_add(v, (int)lane + d < n ? o : G1X29::inf());
  }
  return v;
}
// The 32-bit flavour, for the start-up-only group FFT below.
__device__ __forceinline__ Fq shfl_xor_fq(const Fq& v, int m) {
  Fq r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = __shfl_xor(v.l[i], m, 64);
  return r;
}

// Final level: one lane per bucket folds whatever is left and writes the dense bucket array. A bucket
// that still holds more than FINAL_SERIAL partial sums (a witness column where one value dominates)
// is finished by the whole wavefront: lanes take strided shares, then a shuffle-tree sum.
constexpr uint32_t FINAL_SERIAL = 6;
__global__ __launch_bounds__(64) void msm_accum_final_kernel(const uint32_t* off_in_all, uint32_t nb, const G1X29* in_list, size_t in_cap,
                                                             G1X29* dense) {
  const uint32_t col = blockIdx.y, lane = threadIdx.x;
  const uint32_t b = blockIdx.x * 64 + lane;  // nb is a multiple of 64
  const uint32_t* off_in = off_in_all + (size_t)col * (nb + 1);
  const G1X29* in = in_list + (size_t)col * in_cap;
  const uint32_t lo = off_in[b], hi = off_in[b + 1];
  const uint32_t serial_end = min(hi, lo + FINAL_SERIAL);
  G1X29 acc = G1X29::inf();
  for (uint32_t e = lo; e < serial_end; e++) acc = x29_add(acc, ld_x29(in + e));
  unsigned long long heavy = __ballot(hi > serial_end);
  while (heavy) {
    const int src = __ffsll((long long)heavy) - 1;
    heavy &= heavy - 1;
    const uint32_t h_lo = __shfl(serial_end, src, 64), h_hi = __shfl(hi, src, 64);
    G1X29 part = G1X29::inf();
    for (uint32_t e = h_lo + lane; e < h_hi; e += 64) part = x29_add(part, ld_x29(in + e));
    part = wave_sum29(part);
    if ((int)lane == src) acc = x29_add(acc, part);
  }
  st_x29(dense + (size_t)col * nb + b, acc);
}

// rows[col][g] = sum_r dense[col][64g + r],  cols[col][r] = sum_g dense[col][64g + r].
// Lane-parallel serial sums, no shuffles: the prover is bound by instruction issue, and a 64-lane shuffle tree
// spends 6 wave-level point additions to add 64 values that a lane-per-output loop adds in 63 for all 64
// outputs at once. One workgroup of ROWCOL_WAVES wavefronts per "unit": unit 0 computes the 64 column sums (lane = r,
// each wavefront 1 / ROWCOL_WAVES of the g range), unit u >= 1 the row sums of g in [64(u-1), 64u) (lane = g, each
// wavefront 64 / ROWCOL_WAVES of the 64 r); the partial results meet in LDS, pairwise. The kernel is a handful of
// workgroups per column and pure latency — a point addition is ~6.5 us of dependent products — and it sits on the
// transcript's chain at the end of every commitment batch, so what counts is its DEPTH: 8 + 3 additions with eight
// wavefronts (round 3) against 16 + 3 with four.
constexpr uint32_t ROWCOL_WAVES = 8;
// Units [0, U) with U = ceil(G / 64): the column sums over the row groups g in [64 u, 64 u + 64) (lane = r; cols[col][u][r] —
// msm_fold adds the U slices: one slice up to k = 18, eight at k = 22, where ONE unit summing all 512 row groups was a
// chain of 64 + 3 additions, 0.7 ms of a 2^22-point MSM); units [U, 2U): the row sums of g in [64 (unit - U), ...) (lane = g).
__global__ __launch_bounds__(64 * ROWCOL_WAVES) void msm_rowcol_kernel(const G1X29* dense, uint32_t nb, G1X29* rows, G1X29* cols) {
  __shared__ G1X29 part[ROWCOL_WAVES / 2][64];
  const uint32_t col = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t G = nb >> 6, U = (G + 63) >> 6;
  const bool col_unit = blockIdx.x < U;
  const uint32_t unit = col_unit ? blockIdx.x : blockIdx.x - U;
  const G1X29* d = dense + (size_t)col * nb;
  const uint32_t g_row = unit * 64 + lane;  // row units only
  constexpr uint32_t per = 64 / ROWCOL_WAVES;
  G1X29 acc = G1X29::inf();
  if (col_unit) {
    const uint32_t g0 = unit * 64 + per * wv, g1 = min(G, g0 + per);
    for (uint32_t g = g0; g < g1; g++) acc = x29_add(acc, ld_x29(d + 64 * g + lane));
  } else if (g_row < G) {
    for (uint32_t r = per * wv; r < per * wv + per; r++) acc = x29_add(acc, ld_x29(d + 64 * g_row + r));
  }
  for (uint32_t half = ROWCOL_WAVES / 2; half >= 1; half >>= 1) {  // waves [half, 2 half) hand their sums to waves [0, half)
    if (wv >= half && wv < 2 * half) part[wv - half][lane] = acc;
    __syncthreads();
    if (wv < half) acc = x29_add(acc, part[wv][lane]);
    __syncthreads();
  }
  if (wv == 0) {
    if (col_unit) st_x29(cols + ((size_t)col * U + unit) * 64 + lane, acc);
    else if (g_row < G) st_x29(rows + (size_t)col * G + g_row, acc);
  }
}

// The same sums as shuffle trees (latency mode): one wavefront per output GROUP — workgroup (u, j) of 64 lanes loads the 64
// buckets of row group g = 64 u' + j (row units) or, transposed, the 64 row groups' buckets of residue r = j (column
// units), and adds them in six shuffle rounds: depth 6 instead of 8 + 3, at 4.4 x the additions (128 wavefronts x 6 per
// 64 x 64 block instead of 16 x 11). Same slots, same values as group elements.
__global__ __launch_bounds__(64) void msm_rowcol_tree_kernel(const G1X29* dense, uint32_t nb, G1X29* rows, G1X29* cols) {
  const uint32_t col = blockIdx.z, lane = threadIdx.x, j = blockIdx.x;
  const uint32_t G = nb >> 6, U = (G + 63) >> 6;
  const bool col_unit = blockIdx.y < U;
  const uint32_t unit = col_unit ? blockIdx.y : blockIdx.y - U;
  const G1X29* d = dense + (size_t)col * nb;
  G1X29 v = G1X29::inf();
  if (col_unit) {  // residue r = j, row groups 64 unit + lane
    const uint32_t g = unit * 64 + lane;
    if (g < G) v = ld_x29(d + 64 * g + j);
  } else {  // row group g = 64 unit + j, residues r = lane
    const uint32_t g = unit * 64 + j;
    if (g < G) v = ld_x29(d + 64 * g + lane);
  }
  v = wave_sum29(v);
  if (lane == 0) {
    if (col_unit) st_x29(cols + ((size_t)col * U + unit) * 64 + j, v);
    else if (unit * 64 + j < G) st_x29(rows + (size_t)col * G + unit * 64 + j, v);
  }
}

// ------------------------------------------------------------------ quad-lane point additions (latency mode)
// One XYZZ addition spread over the FOUR lanes of a quad: every lane of the quad holds the same two points (replicated), takes
// one of the (up to) four independent products of each of the formula's four rounds — its operands picked by its position in
// the quad — and the results go round the quad with DPP quad_perm moves, so that all four lanes end with the same sum. 4
// products + selects + broadcasts per lane (~1,300 VALU) instead of 14 products (~3,100): the bucket reduction's chain of ~30
// dependent additions per commitment batch is what a lone proof waits for (a lone wavefront issues a dependent instruction
// every ~6 cycles whatever it is), and instruction-level parallelism inside ONE lane bought nothing (profiles/r04c_*).
// Costs 4 lanes per addition, so only where the chip is empty anyway: amdzk_ctx::msm_latency_mode. Bounds as x29_add
// (y leaves as the sum of two reduced products, below 4p).
template <int J> __device__ __forceinline__ Fq29 quad_bcast(const Fq29& v) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], J * 0x55, 0xf, 0xf, false);  // quad_perm:[J,J,J,J]
  return r;
}
__device__ __forceinline__ Fq29 quad_sel(uint32_t role, const Fq29& a0, const Fq29& a1, const Fq29& a2, const Fq29& a3) {
  Fq29 r;
  const bool hi = (role & 2u) != 0, odd = (role & 1u) != 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint32_t lo2 = odd ? a1.l[i] : a0.l[i], hi2 = odd ? a3.l[i] : a2.l[i];
    r.l[i] = hi ? hi2 : lo2;
  }
  return r;
}
__device__ __forceinline__ G1X29 x29_dbl_quad(const G1X29& p, uint32_t role) {
  if (p.is_inf()) return p;
  const Fq29 u = f29_add(p.y, p.y);                                                   // < 10
  Fq29 res = f29_mul(quad_sel(role, u, p.x, u, p.x), quad_sel(role, u, p.x, u, p.x));  // u^2 (100 p^2) | x^2 (81)
  const Fq29 v = quad_bcast<0>(res), xx = quad_bcast<1>(res);
  const Fq29 m = f29_add(f29_add_lazy(xx, xx), xx);                                   // < 6
  res = f29_mul(quad_sel(role, u, p.x, m, m), quad_sel(role, v, v, m, m));             // w = u v | s = x v | m^2 (36)
  const Fq29 w = quad_bcast<0>(res), sv = quad_bcast<1>(res), mm = quad_bcast<2>(res);
  G1X29 r;
  r.x = f29_sub5(mm, f29_add(sv, sv));                                                // < 7
  const Fq29 d = f29_sub8(sv, r.x);                                                   // < 10
  res = f29_mul(quad_sel(role, m, f29_neg3(w), v, w), quad_sel(role, d, p.y, p.zz, p.zzz));  // m d (60) | (3p - w) y (15) | v zz | w zzz
  r.y = f29_add(quad_bcast<0>(res), quad_bcast<1>(res));                              // < 4
  r.zz = quad_bcast<2>(res);
  r.zzz = quad_bcast<3>(res);
  return r;
}
__device__ __forceinline__ G1X29 x29_add_quad(const G1X29& a, const G1X29& b, uint32_t role) {
  if (b.is_inf()) return a;  // the four lanes hold the same points: every branch is uniform over the quad
  if (a.is_inf()) return b;
  Fq29 res = f29_mul(quad_sel(role, a.x, b.x, a.y, b.y), quad_sel(role, b.zz, a.zz, b.zzz, a.zzz));  // u1 (18 p^2) | u2 | s1 (10) | s2
  const Fq29 u1 = quad_bcast<0>(res), s1 = quad_bcast<2>(res);
  const Fq29 p = f29_sub3(quad_bcast<1>(res), u1);                                    // < 5
  const Fq29 r = f29_sub3(quad_bcast<3>(res), s1);                                    // < 5
  res = f29_mul(quad_sel(role, p, r, a.zz, a.zzz), quad_sel(role, p, r, b.zz, b.zzz));  // pp (25) | rr | zz1 zz2 | zzz1 zzz2
  const Fq29 pp = quad_bcast<0>(res), rr = quad_bcast<1>(res), zz12 = quad_bcast<2>(res), zzz12 = quad_bcast<3>(res);
  if (f29_is_zero_mod_p(pp)) {
    if (f29_is_zero_mod_p(rr)) return x29_dbl_quad(a, role);
    return G1X29::inf();
  }
  res = f29_mul(quad_sel(role, p, u1, zz12, zz12), pp);                               // ppp (10) | q | zz3
  const Fq29 ppp = quad_bcast<0>(res), q = quad_bcast<1>(res);
  G1X29 o;
  o.zz = quad_bcast<2>(res);
  const Fq29 sq = f29_add(ppp, f29_add_lazy(q, q));                                   // < 6
  o.x = f29_sub7(rr, sq);                                                             // < 9
  const Fq29 t = f29_sub10(q, o.x);                                                   // < 12
  res = f29_mul(quad_sel(role, r, f29_neg3(s1), zzz12, zzz12), quad_sel(role, t, ppp, ppp, ppp));  // r t (60) | (3p - s1) ppp (6) | zzz3
  o.y = f29_add(quad_bcast<0>(res), quad_bcast<1>(res));                              // < 4
  o.zzz = quad_bcast<2>(res);
  return o;
}

// A point in LDS as 36 consecutive words; all four lanes of a quad read the same words (a broadcast read).
__device__ __forceinline__ G1X29 lds_ld_x29(const uint32_t* p) {
  G1X29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.x.l[i] = p[i];
    r.y.l[i] = p[9 + i];
    r.zz.l[i] = p[18 + i];
    r.zzz.l[i] = p[27 + i];
  }
  return r;
}
__device__ __forceinline__ void lds_st_x29(uint32_t* p, const G1X29& v, uint32_t role) {  // each lane of the quad stores one coordinate
  const Fq29 c = quad_sel(role, v.x, v.y, v.zz, v.zzz);
#pragma unroll
  for (int i = 0; i < 9; i++) p[role * 9 + i] = c.l[i];
}

// A folding level (msm_accum_seg_kernel<false>) with quad additions: quad t owns partial sums [t * T, (t + 1) * T) of the column's
// list; the four lanes load the same sums, add them together, and lane 0 of the quad stores. Same slots, same values.
__global__ __launch_bounds__(MSM_THREADS) void msm_accum_fold_quad_kernel(AccArgs a) {
  const uint32_t col = blockIdx.y;
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, t = gtid >> 2, role = gtid & 3u;
  const uint32_t* off_in = a.off_in + (size_t)col * (a.nb + 1);
  const uint32_t* off_out = a.off_out + (size_t)col * (a.nb + 1);
  const uint32_t total = off_in[a.nb];
  const uint32_t start = t * a.T;
  if (start >= total) return;
  const uint32_t end = min(start + a.T, total);
  uint32_t lo = 0, hi = a.nb;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (off_in[mid] <= start) lo = mid; else hi = mid;
  }
  uint32_t b = lo, b_end = off_in[b + 1];
  const G1X29* in = a.in_list + (size_t)col * a.in_cap;
  G1X29* out = a.out_list + (size_t)col * a.out_cap;
  G1X29 acc = G1X29::inf();
  for (uint32_t e = start; e < end; e++) {
    if (e >= b_end) {
      if (role == 0) st_x29(out + off_out[b] + (t - off_in[b] / a.T), acc);
      acc = G1X29::inf();
      do {
        b++;
        b_end = off_in[b + 1];
      } while (e >= b_end);
    }
    acc = x29_add_quad(acc, ld_x29(in + e), role);
  }
  if (role == 0) st_x29(out + off_out[b] + (t - off_in[b] / a.T), acc);
}

// Row / column sums with quad additions: one workgroup of 64 quads per OUTPUT (a row group's 64 residues, or a residue's 64 row
// groups): the 64 points meet pairwise through LDS in six rounds. grid = (64 outputs, 2 units, columns); nb <= 4096 (G <= 64).
__global__ __launch_bounds__(256) void msm_rowcol_quad_kernel(const G1X29* dense, uint32_t nb, G1X29* rows, G1X29* cols) {
  __shared__ uint32_t pts[64 * 36];
  const uint32_t col = blockIdx.z, j = blockIdx.x, quad = threadIdx.x >> 2, role = threadIdx.x & 3u;
  const uint32_t G = nb >> 6;
  const bool col_unit = blockIdx.y == 0;
  const G1X29* d = dense + (size_t)col * nb;
  if (!col_unit && j >= G) return;
  // column unit: residue r = j, this quad's row group g = quad; row unit: row group g = j, this quad's residue r = quad
  const uint32_t g = col_unit ? quad : j, r_ = col_unit ? j : quad;
  G1X29 v = g < G ? ld_x29(d + 64 * g + r_) : G1X29::inf();
  for (uint32_t s = 32; s >= 1; s >>= 1) {
    if (quad >= s && quad < 2 * s) lds_st_x29(pts + (quad - s) * 36, v, role);
    __syncthreads();
    if (quad < s) v = x29_add_quad(v, lds_ld_x29(pts + quad * 36), role);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (col_unit) st_x29(cols + (size_t)col * 64 + j, v);
    else st_x29(rows + (size_t)col * G + j, v);
  }
}

// msm_fold for nb <= 4096 with quad additions: quads [0, 64) hold the row sums T_g, quads [64, 128) the column sums C_r; suffix
// sums by six shift-and-add rounds through LDS (two buffers per vector), their sum by six halving rounds, then quad 0:
// out = 64 * sum_g g T_g + sum_g T_g + sum_r r C_r.
__global__ __launch_bounds__(512) void msm_fold_quad_kernel(const G1X29* rows, const G1X29* cols, uint32_t nb, G1X* out) {
  __shared__ uint32_t buf[2][2][64 * 36];  // [vector][ping-pong][point]
  __shared__ uint32_t keep[3 * 36];        // sum_g T_g, sum_g g T_g, sum_r r C_r
  const uint32_t col = blockIdx.x, quad = threadIdx.x >> 2, role = threadIdx.x & 3u, vec = quad >> 6, l = quad & 63u;
  const uint32_t G = nb >> 6;
  G1X29 v = vec == 0 ? (l < G ? ld_x29(rows + (size_t)col * G + l) : G1X29::inf()) : ld_x29(cols + (size_t)col * 64 + l);
  int cur = 0;
  lds_st_x29(buf[vec][cur] + l * 36, v, role);
  __syncthreads();
  for (uint32_t dd = 1; dd < 64; dd <<= 1) {  // v_l <- v_l + v_{l + dd}: after six rounds the suffix sums S_l
    if (l + dd < 64) v = x29_add_quad(v, lds_ld_x29(buf[vec][cur] + (l + dd) * 36), role);
    cur ^= 1;
    lds_st_x29(buf[vec][cur] + l * 36, v, role);
    __syncthreads();
  }
  if (vec == 0 && l == 0) lds_st_x29(keep, v, role);  // S_0 of the rows = sum_g T_g
  if (l == 0) v = G1X29::inf();                        // the weighted sum is S_1 + ... + S_63
  for (uint32_t s = 32; s >= 1; s >>= 1) {
    __syncthreads();
    if (l >= s && l < 2 * s) lds_st_x29(buf[vec][cur] + (l - s) * 36, v, role);
    __syncthreads();
    if (l < s) v = x29_add_quad(v, lds_ld_x29(buf[vec][cur] + l * 36), role);
  }
  if (l == 0) lds_st_x29(keep + (1 + vec) * 36, v, role);
  __syncthreads();
  if (quad == 0) {
    G1X29 acc = lds_ld_x29(keep + 36);
#pragma unroll 1
    for (int i = 0; i < 6; i++) acc = x29_dbl_quad(acc, role);
    acc = x29_add_quad(acc, lds_ld_x29(keep), role);
    acc = x29_add_quad(acc, lds_ld_x29(keep + 72), role);
    if (role == 0) st_x(out + col, x29_to_r256(acc));
  }
}

__device__ __forceinline__ G1X29 x29_mul_small(const G1X29& p, uint32_t k, int nbits) {
  G1X29 acc = G1X29::inf();
#pragma unroll 1
  for (int b = nbits - 1; b >= 0; b--) {
    acc = x29_dbl(acc);
    if ((k >> b) & 1) acc = x29_add(acc, p);
  }
  return acc;
}
// 32-bit flavour (group FFT, start-up only)
__device__ __forceinline__ G1X x_mul_small(const G1X& p, uint32_t k, int nbits) {
  G1X acc = G1X::inf();
#pragma unroll 1
  for (int b = nbits - 1; b >= 0; b--) {
    acc = x_dbl(acc);
    if ((k >> b) & 1) acc = x_add(acc, p);
  }
  return acc;
}

// out[col] = sum_r r*cols[r] + sum_g (64g+1)*rows[g]
//          = sum_r r*cols[r] + 64 * (sum_g g*rows[g]) + sum_g rows[g].
// A weighted sum sum_l l * v_l over the 64 lanes of a wavefront is the sum of the suffix sums S_1 .. S_63 (S_l = v_l + ... +
// v_63): six shuffle rounds for the suffixes, six for their sum — 12 dependent point additions where the per-lane 6-bit
// double-and-add followed by a shuffle tree took 18 — and S_0, the plain sum, comes with it (round 3). One wavefront per 64
// row sums (W = ceil(G / 64) of them; g = 64 w + l gives sum_g g v_g = sum_w (A_w + 64 w B_w), A_w the wavefront's
// weighted sum and B_w its plain sum) and one for the 64 column sums; lane 0 of the first wavefront combines them by
// Horner (6 doublings per factor 64). The kernel is pure latency, one workgroup per column, at the end of every
// commitment batch. The result leaves in the packed radix-2^256 XYZZ form (x29_to_r256): the only radix conversion of
// the whole MSM.
// MAXW: the most row-sum wavefronts a workgroup may have (1 up to 2^12 buckets, i.e. every proof-sized MSM: two wavefronts,
// the whole register file each; 8 for the window widths of k >= 19).
template <int MAXW>
__global__ __launch_bounds__(64 * (MAXW + 1)) void msm_fold_kernel(const G1X29* rows, const G1X29* cols, uint32_t nb, G1X* out) {
  __shared__ G1X29 partA[8], partB[8], partC;
  const uint32_t col = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const uint32_t G = nb >> 6, W = (G + 63) >> 6;
  if (wv <= W) {
    G1X29 v;
    if (wv < W) {
      const uint32_t g = wv * 64 + lane;
      v = g < G ? ld_x29(rows + (size_t)col * G + g) : G1X29::inf();
    } else {
      v = ld_x29(cols + (size_t)col * W * 64 + lane);  // the W slices of the column sums (msm_rowcol)
      for (uint32_t u = 1; u < W; u++) v = x29_add(v, ld_x29(cols + ((size_t)col * W + u) * 64 + lane));
    }
    const G1X29 suf = wave_suffix29(v, lane);
    if (lane == 0 && wv < W) partB[wv] = suf;
    const G1X29 wsum = wave_sum29(lane == 0 ? G1X29::inf() : suf);
    if (lane == 0) {
      if (wv < W) partA[wv] = wsum;
      else partC = wsum;
    }
  }
  __syncthreads();
  if (wv == 0) {
    G1X29 acc;
    if (W > 1) {
      // wavefront 0, lanes w < W (<= 8): Y = sum_w w B_w = the sum of the suffix sums of B over w = 1 .. W - 1, sum_w B_w (the
      // suffix sum of lane 0, parked in partB[0]) and sum_w A_w — three shuffle rounds each instead of lane 0 walking the
      // W values three times. One point live at a time beside the running one: the kernel may hold 168 registers.
      G1X29 y;
      {
        const G1X29 sufb = wave_suffix29_n(lane < W ? partB[lane] : G1X29::inf(), lane, 8);
        if (lane == 0) partB[0] = sufb;  // lane 0 alone reads it back below
        y = wave_sum29_n(lane == 0 || lane >= W ? G1X29::inf() : sufb, 8);
      }
#pragma unroll 1
      for (int i = 0; i < 6; i++) y = x29_dbl(y);
      acc = x29_add(wave_sum29_n(lane < W ? partA[lane] : G1X29::inf(), 8), y);
    } else {
      acc = partA[0];
    }
    if (t == 0) {
#pragma unroll 1
      for (int i = 0; i < 6; i++) acc = x29_dbl(acc);
      acc = x29_add(acc, partB[0]);
      st_x(out + col, x29_to_r256(x29_add(acc, partC)));
    }
  }
}

uint32_t pick_window_bits(uint32_t k) {
  const char* e = getenv("AMDZK_MSM_C");
  if (e) {
    int v = atoi(e);
    if (v >= 8 && v <= 16) return (uint32_t)v;
  }
  if (k <= 10) return 8;
  if (k <= 12) return 10;
  if (k <= 13) return 11;
  if (k <= 14) return 12;
  // k = 17, 18: 13 and 14 prove at the same rate, 13 with a 3.5 ms shorter proof (fewer buckets to fold per column:
  // profiles/r02j_msm_window_sweep.txt)
  if (k <= 18) return 13;
  if (k <= 20) return 15;
  return 16;
}

template <bool SCATTER>
int launch_digits(amdzk_ctx* ctx, uint32_t c, const DigitArgs& a, dim3 grid, bool big) {
  const size_t shmem = (size_t)a.nb * sizeof(uint32_t);
  const char* nm = SCATTER ? "msm_scatter" : "msm_hist";
  switch (c) {
#define ZK_CASE_T(CC, BIG)                                         \
  {                                                                \
    auto kfn = msm_digit_kernel<CC, SCATTER, BIG>;                 \
    if (shmem > 65536) ZK_HIP(ctx, hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    ZK_LAUNCH(ctx, nm, kfn, grid, dim3(BIG ? 1024 : MSM_THREADS), shmem, a); \
  }
#define ZK_CASE(CC) \
  case CC:          \
    ZK_CASE_T(CC, false) break;
#define ZK_CASE2(CC)               \
  case CC:                         \
    if (big) ZK_CASE_T(CC, true)   \
    else ZK_CASE_T(CC, false)      \
    break;
    ZK_CASE(8) ZK_CASE(9) ZK_CASE(10) ZK_CASE(11) ZK_CASE(12) ZK_CASE(13) ZK_CASE2(14) ZK_CASE2(15) ZK_CASE2(16)
#undef ZK_CASE
#undef ZK_CASE2
#undef ZK_CASE_T
    default:
      ZK_FAIL(ctx, AMDZK_E_INVALID, "msm: window bits %u unsupported", c);
  }
  return AMDZK_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------- host side
static int srs_build(amdzk_ctx* ctx, const void* g, const void* g_lagrange, bool src_on_device, uint32_t k, amdzk_srs** out) {
  if (!out || (!g && !g_lagrange)) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs: null argument");
  if (k > 26) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "srs: k %u > 26", k);
  amdzk_srs* s = new amdzk_srs();
  s->k = k;
  s->n = (size_t)1 << k;
  s->c = pick_window_bits(k);
  s->W = (255 + s->c - 1) / s->c;
  const void* src[2] = {g, g_lagrange};
  for (int b = 0; b < 2; b++) {
    if (!src[b]) continue;
    size_t bytes = (size_t)s->W * s->n * sizeof(G1Affine);
    hipError_t e = hipMalloc((void**)&s->table[b], bytes);
    if (e != hipSuccess) {
      for (int j = 0; j < 2; j++)
        if (s->table[j]) hipFree(s->table[j]);
      delete s;
      ZK_FAIL(ctx, AMDZK_E_NOMEM, "srs: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    e = hipMalloc((void**)&s->base[b], s->n * sizeof(G1Affine));
    if (e != hipSuccess) {
      for (int j = 0; j < 2; j++) {
        if (s->table[j]) hipFree(s->table[j]);
        if (s->base[j]) hipFree(s->base[j]);
      }
      delete s;
      ZK_FAIL(ctx, AMDZK_E_NOMEM, "srs: hipMalloc(%zu) failed: %s", s->n * sizeof(G1Affine), hipGetErrorString(e));
    }
    ZK_HIP(ctx, hipMemcpyAsync(s->base[b], src[b], s->n * sizeof(G1Affine), src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                               ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(s->table[b], s->base[b], s->n * sizeof(G1Affine), hipMemcpyDeviceToDevice, ctx->stream));
    dim3 grid((unsigned)((s->n + 255) / 256)), block(256);
    for (uint32_t w = 1; w < s->W; w++)
      ZK_LAUNCH(ctx, "msm_table_next", table_next_kernel, grid, block, 0, s->table[b] + (size_t)(w - 1) * s->n,
                s->table[b] + (size_t)w * s->n, s->n, s->c);
    const size_t count = (size_t)s->W * s->n;  // all windows are built: switch the whole table to radix 2^261
    ZK_LAUNCH(ctx, "msm_table_to_r261", table_to_r261_kernel, dim3((unsigned)((count + 255) / 256)), block, 0, s->table[b], count);
  }
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));  // source buffers may be released by the caller
  *out = s;
  return AMDZK_OK;
}

int zk_srs_upload(amdzk_ctx* ctx, const uint64_t* g, const uint64_t* g_lagrange, uint32_t k, amdzk_srs** out) {
  return srs_build(ctx, g, g_lagrange, false, k, out);
}

// ---- ParamsKZG::setup(k, rng) [UP] with the secret s given explicitly (unsafe by construction, test
// and benchmark use only, exactly like upstream's setup): g[i] = s^i G, g_lagrange[i] = L_i(s) G with
// L_i(s) = omega^i (s^n - 1) / (n (s - omega^i)). Everything on the device.
namespace {
// out[i] = sum over set bits j of the canonical scalar of table[j] (table[j] = 2^j G), affine.
__global__ __launch_bounds__(256) void fixed_base_mul_kernel(const Fr* scalars, const G1Affine* table, G1Affine* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + i);
  uint4 lo = sp[0], hi = sp[1];
  Fr s;
  s.l[0] = lo.x; s.l[1] = lo.y; s.l[2] = lo.z; s.l[3] = lo.w;
  s.l[4] = hi.x; s.l[5] = hi.y; s.l[6] = hi.z; s.l[7] = hi.w;
  Fr c = from_mont(s);
  G1X acc = G1X::inf();
#pragma unroll 1
  for (int limb = 0; limb < 8; limb++) {
    uint32_t w = c.l[limb];
#pragma unroll 1
    for (int b = 0; b < 32; b++)
      if ((w >> b) & 1) acc = x_add_affine(acc, ld_aff(table + limb * 32 + b));
  }
  st_aff(out + i, x_to_affine(acc));
}
// p[i] = base^i
__global__ void powers_kernel(Fr* p, Fr base, size_t count, uint32_t chunk) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t start = t * chunk;
  if (start >= count) return;
  Fr cur = pow_u64(base, start);
  size_t end = start + chunk < count ? start + chunk : count;
  for (size_t i = start; i < end; i++) {
    uint4* q = reinterpret_cast<uint4*>(p + i);
    q[0] = make_uint4(cur.l[0], cur.l[1], cur.l[2], cur.l[3]);
    q[1] = make_uint4(cur.l[4], cur.l[5], cur.l[6], cur.l[7]);
    cur = mul(cur, base);
  }
}
// den[i] = s - w[i]
__global__ void sub_from_const_kernel(Fr* den, const Fr* w, Fr s, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* q = reinterpret_cast<const uint4*>(w + i);
  uint4 a = q[0], b = q[1];
  Fr v;
  v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
  v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
  Fr r = sub(s, v);
  uint4* o = reinterpret_cast<uint4*>(den + i);
  o[0] = make_uint4(r.l[0], r.l[1], r.l[2], r.l[3]);
  o[1] = make_uint4(r.l[4], r.l[5], r.l[6], r.l[7]);
}
// a[i] = a[i] * w[i] * c
__global__ void mul2_kernel(Fr* a, const Fr* w, Fr c, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* q = reinterpret_cast<const uint4*>(w + i);
  uint4 x0 = q[0], x1 = q[1];
  Fr v;
  v.l[0] = x0.x; v.l[1] = x0.y; v.l[2] = x0.z; v.l[3] = x0.w;
  v.l[4] = x1.x; v.l[5] = x1.y; v.l[6] = x1.z; v.l[7] = x1.w;
  const uint4* p = reinterpret_cast<const uint4*>(a + i);
  uint4 y0 = p[0], y1 = p[1];
  Fr u;
  u.l[0] = y0.x; u.l[1] = y0.y; u.l[2] = y0.z; u.l[3] = y0.w;
  u.l[4] = y1.x; u.l[5] = y1.y; u.l[6] = y1.z; u.l[7] = y1.w;
  Fr r = mul(mul(u, v), c);
  uint4* o = reinterpret_cast<uint4*>(a + i);
  o[0] = make_uint4(r.l[0], r.l[1], r.l[2], r.l[3]);
  o[1] = make_uint4(r.l[4], r.l[5], r.l[6], r.l[7]);
}
}  // namespace

int zk_batch_invert(amdzk_ctx* ctx, Fr* d_a, Fr* d_scratch, size_t total);

// ---- ParamsKZG::{write, read} [UP]: k as u32 LE, then the n points of g and
// of g_lagrange in halo2curves' 32-byte compressed form, then g2 and s_g2 (64 bytes each, passed
// through untouched: the prover never uses G2). Compression / decompression (one Fq square root per
// point, q = 3 mod 4) run on the device.
namespace {
__global__ __launch_bounds__(256) void g1_compress_kernel(const G1Affine* pts, uint8_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine p = ld_aff(pts + i);
  uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (!p.is_inf()) {
    Fq x = from_mont(p.x), y = from_mont(p.y);
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = x.l[j];
    w[7] |= (y.l[0] & 1u) << 31;  // byte 31 bit 7 = parity of y
  }
  uint4* o = reinterpret_cast<uint4*>(out + 32 * i);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

__global__ __launch_bounds__(256) void g1_decompress_kernel(const uint8_t* in, G1Affine* out, size_t n, int* err) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* q4 = reinterpret_cast<const uint4*>(in + 32 * i);
  uint4 a = q4[0], b = q4[1];
  Fq x;
  x.l[0] = a.x; x.l[1] = a.y; x.l[2] = a.z; x.l[3] = a.w;
  x.l[4] = b.x; x.l[5] = b.y; x.l[6] = b.z; x.l[7] = b.w;
  const uint32_t ysign = x.l[7] >> 31;
  x.l[7] &= 0x7fffffffu;
  G1Affine p;
  p.x = Fq::zero();
  p.y = Fq::zero();
  bool ok = true;
  // x must be canonical (< q)
  {
    uint32_t borrow = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint64_t t = (uint64_t)x.l[j] - FqP::p(j) - borrow;
      borrow = (uint32_t)(t >> 63);
    }
    ok = borrow == 1;
  }
  if (ok && !(x.is_zero() && ysign == 0)) {
    Fq xm = to_mont(x);
    Fq three = add(add(Fq::one(), Fq::one()), Fq::one());
    Fq rhs = add(mul(sqr(xm), xm), three);
    // (q+1)/4
    const uint32_t e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
    Fq y = pow_u256(rhs, e);
    if (sqr(y) != rhs) {
      ok = false;
    } else {
      Fq yc = from_mont(y);
      if ((yc.l[0] & 1u) != ysign) y = neg(y);
      p.x = xm;
      p.y = y;
    }
  }
  if (!ok) atomicExch(err, 1);
  st_aff(out + i, p);
}
}  // namespace

size_t zk_srs_serialized_size(uint32_t k) { return 4 + 2 * ((size_t)32 << k) + 128; }

int zk_srs_write(amdzk_ctx* ctx, const amdzk_srs* s, const uint8_t g2[64], const uint8_t s_g2[64], uint8_t* out, size_t cap) {
  if (!s || !s->base[0] || !s->base[1] || !out) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs_write: both bases must be resident");
  const size_t need = zk_srs_serialized_size(s->k);
  if (cap < need) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs_write: buffer too small (%zu < %zu)", cap, need);
  uint8_t* d = nullptr;
  ZK_TRY(zk_ws_reserve(ctx, 3, 2 * s->n * 32, (void**)&d));
  dim3 grid((unsigned)((s->n + 255) / 256)), block(256);
  ZK_LAUNCH(ctx, "g1_compress", g1_compress_kernel, grid, block, 0, s->base[0], d, s->n);
  ZK_LAUNCH(ctx, "g1_compress", g1_compress_kernel, grid, block, 0, s->base[1], d + s->n * 32, s->n);
  uint32_t k = s->k;
  memcpy(out, &k, 4);
  ZK_HIP(ctx, hipMemcpyAsync(out + 4, d, 2 * s->n * 32, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  if (g2) memcpy(out + 4 + 2 * s->n * 32, g2, 64); else memset(out + 4 + 2 * s->n * 32, 0, 64);
  if (s_g2) memcpy(out + 4 + 2 * s->n * 32 + 64, s_g2, 64); else memset(out + 4 + 2 * s->n * 32 + 64, 0, 64);
  return AMDZK_OK;
}

static int srs_build(amdzk_ctx* ctx, const void* g, const void* g_lagrange, bool src_on_device, uint32_t k, amdzk_srs** out);

int zk_srs_read(amdzk_ctx* ctx, const uint8_t* data, size_t len, amdzk_srs** out, uint8_t g2_out[64], uint8_t s_g2_out[64]) {
  if (!data || !out || len < 4) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs_read: null or truncated input");
  uint32_t k;
  memcpy(&k, data, 4);
  if (k > 26) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "srs_read: k %u > 26", k);
  const size_t n = (size_t)1 << k;
  if (len < zk_srs_serialized_size(k)) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs_read: %zu bytes given, %zu needed for k = %u", len, zk_srs_serialized_size(k), k);
  char* ws = nullptr;
  ZK_TRY(zk_ws_reserve(ctx, 3, 2 * n * 32 + 2 * n * sizeof(G1Affine) + 256, (void**)&ws));
  uint8_t* d_in = (uint8_t*)ws;
  G1Affine* pts = (G1Affine*)(ws + 2 * n * 32);
  int* d_err = (int*)(ws + 2 * n * 32 + 2 * n * sizeof(G1Affine));
  ZK_HIP(ctx, hipMemcpyAsync(d_in, data + 4, 2 * n * 32, hipMemcpyHostToDevice, ctx->stream));
  ZK_HIP(ctx, hipMemsetAsync(d_err, 0, sizeof(int), ctx->stream));
  ZK_LAUNCH(ctx, "g1_decompress", g1_decompress_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, d_in, pts, 2 * n, d_err);
  int herr = 0;
  ZK_HIP(ctx, hipMemcpyAsync(&herr, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  if (herr) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs_read: invalid point encoding (not on the curve or x >= q)");
  if (g2_out) memcpy(g2_out, data + 4 + 2 * n * 32, 64);
  if (s_g2_out) memcpy(s_g2_out, data + 4 + 2 * n * 32 + 64, 64);
  return srs_build(ctx, pts, pts + n, true, k, out);
}

int zk_srs_setup(amdzk_ctx* ctx, uint32_t k, const uint64_t s_mont[4], const uint64_t omega_mont[4], amdzk_srs** out,
                 uint64_t* g_out, uint64_t* g_lagrange_out) {
  if (k > 26) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "srs_setup: k %u > 26", k);
  const size_t n = (size_t)1 << k;
  Fr s, omega;
  memcpy(s.l, s_mont, 32);
  memcpy(omega.l, omega_mont, 32);
  // workspace (slot 3): gtable[256] | scal[n] | w[n] | scratch[n] | g[n] | gl[n]
  char* ws = nullptr;
  const size_t bytes = 256 * sizeof(G1Affine) + 3 * n * sizeof(Fr) + 2 * n * sizeof(G1Affine);
  ZK_TRY(zk_ws_reserve(ctx, 3, bytes, (void**)&ws));
  G1Affine* gtab = (G1Affine*)ws;
  Fr* scal = (Fr*)(ws + 256 * sizeof(G1Affine));
  Fr* w = scal + n;
  Fr* scratch = w + n;
  G1Affine* g = (G1Affine*)(scratch + n);
  G1Affine* gl = g + n;
  {  // 2^j G on the host (256 doublings)
    std::vector<G1Affine> t(256);
    G1Affine gen;
    gen.x = Fq::one();
    gen.y = add(Fq::one(), Fq::one());
    t[0] = gen;
    for (int j = 1; j < 256; j++) t[j] = x_to_affine(x_dbl_affine(t[j - 1]));
    ZK_HIP(ctx, hipMemcpyAsync(gtab, t.data(), 256 * sizeof(G1Affine), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  }
  const uint32_t chunk = 64;
  dim3 pg((unsigned)(((n + chunk - 1) / chunk + 63) / 64)), pb(64);
  dim3 eg((unsigned)((n + 255) / 256)), eb(256);
  ZK_LAUNCH(ctx, "srs_powers", powers_kernel, pg, pb, 0, scal, s, n, chunk);
  ZK_LAUNCH(ctx, "srs_fixed_base_mul", fixed_base_mul_kernel, eg, eb, 0, scal, gtab, g, n);
  // Lagrange scalars
  ZK_LAUNCH(ctx, "srs_powers", powers_kernel, pg, pb, 0, w, omega, n, chunk);
  ZK_LAUNCH(ctx, "srs_sub", sub_from_const_kernel, eg, eb, 0, scal, w, s, n);
  ZK_TRY(zk_batch_invert(ctx, scal, scratch, n));
  Fr two = add(Fr::one(), Fr::one());
  Fr mult = mul(sub(pow_u64(s, n), Fr::one()), inv(pow_u64(two, k)));  // (s^n - 1) / n
  ZK_LAUNCH(ctx, "srs_mul2", mul2_kernel, eg, eb, 0, scal, w, mult, n);
  ZK_LAUNCH(ctx, "srs_fixed_base_mul", fixed_base_mul_kernel, eg, eb, 0, scal, gtab, gl, n);
  if (g_out) ZK_HIP(ctx, hipMemcpyAsync(g_out, g, n * sizeof(G1Affine), hipMemcpyDeviceToHost, ctx->stream));
  if (g_lagrange_out) ZK_HIP(ctx, hipMemcpyAsync(g_lagrange_out, gl, n * sizeof(G1Affine), hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  return srs_build(ctx, g, gl, true, k, out);
}

// ---- arithmetic::g_to_lagrange(g, k) and ParamsKZG::downsize(k) [UP]:
// g_lagrange = (1/n) * FFT_{omega^-1}(g) over the group — the same radix-2 network as the scalar NTT with
// every twiddle product a G1 scalar multiplication. Decimation in frequency on XYZZ points in HBM
// (k launches, one butterfly per thread), then one pass that scales by 1/n, normalises to affine and
// undoes the bit reversal. Start-up work only: n/2 * log n + n scalar multiplications.
namespace {
// canon: canonical (non-Montgomery) scalar; MSB-first double-and-add.
__device__ G1X x_scalar_mul(const G1X& p, const Fr& canon) {
  G1X acc = G1X::inf();
  int top = -1;
  for (int limb = 7; limb >= 0 && top < 0; limb--)
    if (canon.l[limb]) top = limb * 32 + 31 - __clz(canon.l[limb]);
#pragma unroll 1
  for (int b = top; b >= 0; b--) {
    acc = x_dbl(acc);
    if ((canon.l[b >> 5] >> (b & 31)) & 1) acc = x_add(acc, p);
  }
  return acc;
}
__global__ __launch_bounds__(256) void ecfft_load_kernel(const G1Affine* g, G1X* a, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st_x(a + i, x_from_affine(ld_aff(g + i)));
}
// One DIF round: (u, v) at distance half -> (u + v, (u - v) * w^(pos << tw_shift)).
__global__ __launch_bounds__(256) void ecfft_round_kernel(G1X* a, size_t n, uint32_t half_log, const Fr* tw, uint32_t tw_shift) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n / 2) return;
  const size_t half = (size_t)1 << half_log, pos = j & (half - 1);
  const size_t i0 = ((j >> half_log) << (half_log + 1)) + pos, i1 = i0 + half;
  G1X u = ld_x(a + i0), v = ld_x(a + i1);
  st_x(a + i0, x_add(u, v));
  G1X d = x_add(u, x_neg(v));
  if (pos != 0) {
    const uint4* q = reinterpret_cast<const uint4*>(tw + (pos << tw_shift));
    uint4 lo = q[0], hi = q[1];
    Fr w;
    w.l[0] = lo.x; w.l[1] = lo.y; w.l[2] = lo.z; w.l[3] = lo.w;
    w.l[4] = hi.x; w.l[5] = hi.y; w.l[6] = hi.z; w.l[7] = hi.w;
    d = x_scalar_mul(d, from_mont(w));
  }
  st_x(a + i1, d);
}
__global__ __launch_bounds__(256) void ecfft_finish_kernel(const G1X* a, G1Affine* out, size_t n, uint32_t k, Fr ninv_canon) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  size_t r = k ? (size_t)(__brevll((unsigned long long)i) >> (64 - k)) : 0;
  st_aff(out + r, x_to_affine(x_scalar_mul(ld_x(a + i), ninv_canon)));
}

// d_g, d_out: n affine points on the device (may not alias); d_xyzz: n G1X; d_tw: max(n/2, 1) Fr.
int ecfft_to_lagrange(amdzk_ctx* ctx, const G1Affine* d_g, uint32_t k, const Fr& omega_inv, const Fr& n_inv, G1X* d_xyzz, Fr* d_tw,
                      G1Affine* d_out) {
  const size_t n = (size_t)1 << k;
  dim3 eg((unsigned)((n + 255) / 256)), hg((unsigned)((n / 2 + 255) / 256)), eb(256);
  ZK_LAUNCH(ctx, "ecfft_load", ecfft_load_kernel, eg, eb, 0, d_g, d_xyzz, n);
  if (k > 0) {
    const uint32_t chunk = 64;
    dim3 pg((unsigned)(((n / 2 + chunk - 1) / chunk + 63) / 64)), pb(64);
    ZK_LAUNCH(ctx, "srs_powers", powers_kernel, pg, pb, 0, d_tw, omega_inv, n / 2, chunk);
    for (uint32_t s = 0; s < k; s++)
      ZK_LAUNCH(ctx, "ecfft_round", ecfft_round_kernel, hg, eb, 0, d_xyzz, n, k - 1 - s, (const Fr*)d_tw, s);
  }
  ZK_LAUNCH(ctx, "ecfft_finish", ecfft_finish_kernel, eg, eb, 0, (const G1X*)d_xyzz, d_out, n, k, from_mont(n_inv));
  return AMDZK_OK;
}
}  // namespace

int zk_g_to_lagrange(amdzk_ctx* ctx, const uint64_t* g, uint32_t k, const uint64_t omega_inv[4], const uint64_t n_inv[4], uint64_t* out) {
  if (!g || !out) ZK_FAIL(ctx, AMDZK_E_INVALID, "g_to_lagrange: null argument");
  if (k > 26) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "g_to_lagrange: k %u > 26", k);
  const size_t n = (size_t)1 << k;
  Fr wi, ni;
  memcpy(wi.l, omega_inv, 32);
  memcpy(ni.l, n_inv, 32);
  char* ws = nullptr;  // g[n] | gl[n] | xyzz[n] | tw[n/2]
  ZK_TRY(zk_ws_reserve(ctx, 3, 2 * n * sizeof(G1Affine) + n * sizeof(G1X) + (n / 2 + 1) * sizeof(Fr), (void**)&ws));
  G1Affine* dg = (G1Affine*)ws;
  G1Affine* dl = dg + n;
  G1X* dx = (G1X*)(dl + n);
  Fr* tw = (Fr*)(dx + n);
  ZK_HIP(ctx, hipMemcpyAsync(dg, g, n * sizeof(G1Affine), hipMemcpyHostToDevice, ctx->stream));
  ZK_TRY(ecfft_to_lagrange(ctx, dg, k, wi, ni, dx, tw, dl));
  ZK_HIP(ctx, hipMemcpyAsync(out, dl, n * sizeof(G1Affine), hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  return AMDZK_OK;
}

int zk_srs_downsize(amdzk_ctx* ctx, const amdzk_srs* srs, uint32_t new_k, const uint64_t omega_inv[4], const uint64_t n_inv[4],
                    amdzk_srs** out) {
  if (!srs || !out) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs_downsize: null argument");
  if (new_k > srs->k) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs_downsize: k %u > params k %u", new_k, srs->k);
  if (!srs->base[0]) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs_downsize: params hold no monomial basis g");
  const size_t n = (size_t)1 << new_k;
  Fr wi, ni;
  memcpy(wi.l, omega_inv, 32);
  memcpy(ni.l, n_inv, 32);
  char* ws = nullptr;  // gl[n] | xyzz[n] | tw[n/2]
  ZK_TRY(zk_ws_reserve(ctx, 3, n * sizeof(G1Affine) + n * sizeof(G1X) + (n / 2 + 1) * sizeof(Fr), (void**)&ws));
  G1Affine* dl = (G1Affine*)ws;
  G1X* dx = (G1X*)(dl + n);
  Fr* tw = (Fr*)(dx + n);
  ZK_TRY(ecfft_to_lagrange(ctx, srs->base[0], new_k, wi, ni, dx, tw, dl));
  return srs_build(ctx, srs->base[0], dl, true, new_k, out);
}

// ParamsKZG::get_g() / g_lagrange: the affine bases back on the host.
int zk_srs_get(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, uint64_t* out) {
  if (!srs || !out || basis < 0 || basis > 1) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs_get: bad argument");
  if (!srs->base[basis]) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs_get: basis %d was not uploaded", basis);
  ZK_HIP(ctx, hipMemcpyAsync(out, srs->base[basis], srs->n * sizeof(G1Affine), hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  return AMDZK_OK;
}

void zk_srs_free(amdzk_ctx*, amdzk_srs* s) {
  if (!s) return;
  for (int b = 0; b < 2; b++) {
    if (s->table[b]) hipFree(s->table[b]);
    if (s->base[b]) hipFree(s->base[b]);
  }
  delete s;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// One group of columns of a batch: counting sort, level-1 accumulation, folds, bucket reduction — every launch on
// ctx->stream (the caller may have pointed it at the ctx's second MSM stream). The level-1 kernel fills the chip; what
// precedes it (the sort) and what follows it (folds, per-bucket sums, the row/column reduction: few wavefronts,
// latency-bound) does not. `l1_after` / `l1_done`: the level-1 launch waits for the previous group's and signals its
// own end, so that group g's tail and group g + 2's sort run UNDER group g + 1's level-1 kernel (zk_msm_dev_xyzz).
struct MsmGeom {
  uint32_t c, W, nb, T1, TL, chunk, nblk;
  int nlev;  // accumulation levels in front of the per-bucket final: 2 (level 1 + one fold) or 3
  bool latency;  // latency mode for this batch: amdzk_ctx::msm_latency_mode and a batch of a few columns
  bool big_digits;  // counting sort with 1024-thread workgroups, up to 256 of them per column
  size_t ecap, cap[4], G;
  size_t o_bh, o_cnt, o_off[4], o_ent, o_list[4], o_dense, o_rows, o_cols, o_ctr, bytes;
};
static constexpr int MSM_NLEV = 3;  // at most: level 1 + two folding levels, then the per-bucket final (MsmGeom::nlev of them are used)

static MsmGeom msm_geometry(const amdzk_srs* srs, size_t ncols, size_t len, size_t ncols_for_task_size, bool latency_mode) {
  MsmGeom g;
  // latency mode applies to batches of a few columns only (the h pieces, the multiopen argument's two, the random polynomial:
  // the chip is empty behind them); the bucket reduction of a 141-column batch is 2,000 wavefronts and throughput-bound —
  // spending 4 lanes per addition there made a lone proof 0.4 ms SLOWER (profiles/r04h_*)
  static const size_t latency_cols = getenv("AMDZK_LATENCY_COLS") && atoi(getenv("AMDZK_LATENCY_COLS")) > 0 ? (size_t)atoi(getenv("AMDZK_LATENCY_COLS")) : 8;
  g.latency = latency_mode && ncols_for_task_size <= latency_cols;
  g.c = srs->c;
  g.W = srs->W;
  g.nb = 1u << (g.c - 1);
  g.ecap = align_up(len * g.W ? len * g.W : 1, 4);
  // task sizes: level 1 adds T1 table points per thread, levels 2.. fold TL partial sums per thread;
  // all levels use the balanced segmented kernel. Every extra folding level costs a latency-bound launch, so two
  // folding levels; T1 = 12 for the large batches (profiles/r02j_msm_task_size.txt: T1 = 8 / 12 / 16 / 32 / 48 take
  // 5.8 / 5.7 / 6.5 / 7.1 / 7.8 ms per proof in level 1 against 2.4 / 2.1 / 1.9 / 1.6 / 1.6 in the folds — a task of 32
  // additions leaves the last of its few thousand wavefronts running alone — and 76.9 against 76.0 proofs/s for
  // 12 against 16 with ten proofs in flight). e_total is a capacity: zero digits never become entries.
  const size_t e_total = g.ecap * ncols_for_task_size;
  g.T1 = 4;
  if (e_total > (size_t)4 * 262144) g.T1 = 8;
  if (e_total > (size_t)8 * 262144) g.T1 = 12;
  // one long column (k >= 19): tens of millions of entries keep the chip full whatever the task size, and every partial
  // sum a task leaves behind is one more addition in the folds (profiles/r04a_*: 2^22 points, T1 = 12 / 24 / 32 / 48 / 64: 7.48 / 7.44 / 7.36 / 7.32 / 7.28 ms)
  if (g.ecap >= ((size_t)1 << 23)) g.T1 = g.ecap >= ((size_t)1 << 25) ? 64 : 32;  // 2^19 points: 1.53 ms with 32, 1.69 with 64; 2^22: 7.33 / 7.14
  if (const char* e = getenv("AMDZK_MSM_T1")) g.T1 = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : g.T1;
  g.TL = 6;  // 4 / 6 / 8 / 12 / 16 prove at the same rate (76.7-78.0 proofs/s); 6 has the shortest proof (profiles/r02j_msm_task_size.txt)
  {
    // ... unless the FULLEST buckets of a uniform column would then reach the per-bucket final with more than FINAL_SERIAL partial
    // sums. The top window of a 254-bit scalar has few digits — (r >> c (W - 1)) + 1 of them: 97 at c = 13 — so those buckets
    // hold len / 97 entries on top of the average len W / nb, they are neighbours (one wavefront of msm_accum_final), and each
    // takes that wavefront through the cooperative path in turn: at 2^18 points (3,983 entries in each of them) the final
    // kernel was 1.48 ms of a 2.33 ms MSM and 10.4 ms of the k = 18 proof (profiles/r04x_k18_top_window_buckets.txt). Two
    // folds of TL must bring the fullest bucket's T1-sums down to ~4: TL = 7 at 2^17, 9 at 2^18; 6 up to 2^16 and for c >= 15.
    const unsigned top_shift = g.c * (g.W - 1);
    const uint64_t r_top = 0x30644e72e131a029ULL;  // the scalar field's modulus, bits 192..255 (contract.sol:211)
    const double top_digits = top_shift >= 192 && top_shift < 256 ? (double)((r_top >> (top_shift - 192)) + 1) : (double)g.nb;
    const double fullest = (double)len * g.W / g.nb + (double)len / std::min<double>(top_digits, (double)g.nb);
    if (top_digits >= 8)  // two or four such buckets (c = 11, 12) cost 40 us each in the final: not worth wider folds
      while (g.TL < 16 && fullest / g.T1 / ((double)g.TL * g.TL) > 4.5) g.TL++;
  }
  if (const char* e = getenv("AMDZK_MSM_TL")) g.TL = (uint32_t)atoi(e) > 1 ? (uint32_t)atoi(e) : g.TL;
  // Folding levels: two (TL = 6) in front of the per-bucket final. ONE fold (AMDZK_MSM_NLEV=2, with TL = max(6, partial
  // sums per bucket / 4)) was measured in round 4 and is a trap: the AVERAGE bucket of a proof-sized MSM then reaches the
  // final kernel with ~5 partial sums, but witness columns are skewed — a few hundred buckets per column hold hundreds — and
  // every such bucket takes the whole wavefront of its 64 neighbours through the cooperative path: msm_accum_final
  // 0.76 -> 8.9 ms per proof, 79.6 -> 72 proofs/s (profiles/r04d_one_fold_level_ab.txt).
  g.nlev = MSM_NLEV;
  if (const char* e = getenv("AMDZK_MSM_NLEV")) {
    if (atoi(e) == 2) {
      const double per_bucket = (double)len * g.W / g.nb / g.T1;
      g.nlev = 2;
      if (!getenv("AMDZK_MSM_TL")) g.TL = std::max<uint32_t>(6, (uint32_t)((per_bucket + 3.0) / 4.0));
    }
  }
  g.cap[0] = g.ecap;
  g.cap[1] = g.ecap / g.T1 + g.nb + 1;
  for (int l = 2; l <= MSM_NLEV; l++) g.cap[l] = g.cap[l - 1] / g.TL + g.nb + 1;
  // counting-sort geometry: one workgroup per `chunk` scalars, at most 64 workgroups per column — 256 workgroups of 1024
  // threads when the batch is a few long columns (msm_digit_kernel<.., BIG>: window widths 14-16 only)
  g.big_digits = g.c >= 14 && len >= ((size_t)1 << 18) && ncols * ((len + 65535) / 65536) < 256;
  if (const char* e = getenv("AMDZK_MSM_BIG_DIGITS")) g.big_digits = g.c >= 14 && atoi(e) != 0;
  const size_t maxblk = g.big_digits ? 256 : 64;
  g.chunk = g.big_digits ? 4096 : 2048;
  while ((len + g.chunk - 1) / g.chunk > maxblk) g.chunk <<= 1;
  g.nblk = len ? (uint32_t)((len + g.chunk - 1) / g.chunk) : 1;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
  g.o_bh = take(ncols * (size_t)g.nblk * g.nb * 4);
  g.o_cnt = take(ncols * g.nb * 4);
  for (int l = 0; l <= MSM_NLEV; l++) g.o_off[l] = take(ncols * (g.nb + 1) * 4);
  g.o_ent = take(ncols * g.ecap * 4);
  g.o_list[0] = 0;
  for (int l = 1; l <= MSM_NLEV; l++) g.o_list[l] = take(ncols * g.cap[l] * sizeof(G1X29));
  g.o_dense = take(ncols * g.nb * sizeof(G1X29));
  g.G = g.nb >> 6;
  g.o_rows = take(ncols * g.G * sizeof(G1X29));
  g.o_cols = take(ncols * ((g.G + 63) / 64) * 64 * sizeof(G1X29));
  g.o_ctr = take(256);
  g.bytes = o;
  return g;
}

static int msm_group(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, const MsmGeom& g, char* ws, const Fr* d_scalars, size_t ncols, size_t len,
                     size_t col_stride, G1X* outp, hipEvent_t l1_after, hipEvent_t l1_done) {
  const uint32_t nb = g.nb;
  uint32_t* blk_hist = (uint32_t*)(ws + g.o_bh);
  uint32_t* cnt = (uint32_t*)(ws + g.o_cnt);
  uint32_t* off[MSM_NLEV + 1];
  G1X29* list[MSM_NLEV + 1];
  for (int l = 0; l <= MSM_NLEV; l++) off[l] = (uint32_t*)(ws + g.o_off[l]), list[l] = (G1X29*)(ws + g.o_list[l]);
  uint32_t* entries = (uint32_t*)(ws + g.o_ent);
  G1X29* dense = (G1X29*)(ws + g.o_dense);
  G1X29 *rows = (G1X29*)(ws + g.o_rows), *cols = (G1X29*)(ws + g.o_cols);

  DigitArgs da;
  da.scalars = d_scalars;
  da.col_stride = col_stride;
  da.len = (uint32_t)len;
  da.nb = nb;
  da.chunk = g.chunk;
  da.nblk = g.nblk;
  da.blk_hist = blk_hist;
  da.off0 = off[0];
  da.entries = entries;
  da.ecap = g.ecap;
  da.table_n = (uint32_t)srs->n;
  dim3 dgrid(g.nblk, (unsigned)ncols);
  // quad-lane point additions in the folds and the bucket reduction: in latency mode (-1), never (0), always (1: tests)
  static const int tail_quad = getenv("AMDZK_TAIL_QUAD") ? atoi(getenv("AMDZK_TAIL_QUAD")) : -1;
  const bool quad_on = (tail_quad < 0 ? g.latency : tail_quad != 0) && ncols <= 65535;
  const size_t scan_shmem = (16 + (nb + 1 <= SCAN_LDS_WORDS ? (size_t)nb + 1 : 0)) * sizeof(uint32_t);
  if (scan_shmem > 65536) {
    static bool scan_attr_set = false;  // per process: the attribute belongs to the function, not to the context
    if (!scan_attr_set) {
      ZK_HIP(ctx, hipFuncSetAttribute((const void*)msm_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((16 + SCAN_LDS_WORDS) * sizeof(uint32_t))));
      scan_attr_set = true;
    }
  }
  ZK_TRY(launch_digits<false>(ctx, g.c, da, dgrid, g.big_digits));
  ZK_LAUNCH(ctx, "msm_blk_offsets", msm_blk_offsets_kernel, dim3((nb + 255) / 256, (unsigned)ncols), dim3(256), 0, blk_hist, cnt, nb, g.nblk);
  ZK_LAUNCH(ctx, "msm_scan", msm_scan_kernel, dim3((unsigned)ncols), dim3(1024), scan_shmem, cnt, off[0], nb, 1u, 1);
  ZK_TRY(launch_digits<true>(ctx, g.c, da, dgrid, g.big_digits));
  for (int l = 1; l <= g.nlev; l++) {
    const uint32_t T = l == 1 ? g.T1 : g.TL;
    ZK_LAUNCH(ctx, "msm_scan", msm_scan_kernel, dim3((unsigned)ncols), dim3(1024), scan_shmem, off[l - 1], off[l], nb, T, 2);
    AccArgs a;
    a.off_in = off[l - 1];
    a.off_out = off[l];
    a.nb = nb;
    a.T = T;
    a.entries = entries;
    a.ecap = g.ecap;
    a.table = srs->table[basis];
    a.in_list = list[l - 1];
    a.in_cap = g.cap[l - 1];
    a.out_list = list[l];
    a.out_cap = g.cap[l];
    const size_t threads = (g.cap[l - 1] + T - 1) / T;
    dim3 grid((unsigned)((threads + MSM_THREADS - 1) / MSM_THREADS), (unsigned)ncols);
    if (l == 1) {
      if (l1_after) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, l1_after, 0));
      static const int l1_lds = getenv("AMDZK_L1_LDS") ? atoi(getenv("AMDZK_L1_LDS")) : 0;
      if (l1_lds == 9) {
        uint32_t* ctr = (uint32_t*)(ws + g.o_ctr);
        ZK_HIP(ctx, hipMemsetAsync(ctr, 0, sizeof(uint32_t), ctx->stream));
        const unsigned persist = (unsigned)std::min<size_t>((size_t)grid.x * ncols, (size_t)ctx->num_cu * 3);
        ZK_LAUNCH(ctx, "msm_accum_l1", msm_accum_l1_persist_kernel, dim3(persist), dim3(MSM_THREADS), 0, a, ctr, grid.x, (uint32_t)ncols);
      } else if (l1_lds == 4) ZK_LAUNCH(ctx, "msm_accum_l1", msm_accum_l1_lds_kernel<4>, grid, dim3(L1L_THREADS), 0, a);
      else if (l1_lds == 3) ZK_LAUNCH(ctx, "msm_accum_l1", msm_accum_l1_lds_kernel<3>, grid, dim3(L1L_THREADS), 0, a);
      else ZK_LAUNCH(ctx, "msm_accum_l1", msm_accum_seg_kernel<true>, grid, dim3(MSM_THREADS), 0, a);
      if (l1_done) ZK_HIP(ctx, hipEventRecord(l1_done, ctx->stream));
    } else if (quad_on && !(getenv("AMDZK_FOLD_QUAD") && atoi(getenv("AMDZK_FOLD_QUAD")) == 0)) {
      const dim3 qgrid((unsigned)((4 * threads + MSM_THREADS - 1) / MSM_THREADS), (unsigned)ncols);
      ZK_LAUNCH(ctx, "msm_accum_fold", msm_accum_fold_quad_kernel, qgrid, dim3(MSM_THREADS), 0, a);
    } else {
      ZK_LAUNCH(ctx, "msm_accum_fold", msm_accum_seg_kernel<false>, grid, dim3(MSM_THREADS), 0, a);
    }
  }
  const unsigned fold_w = (unsigned)((g.G + 63) / 64);  // 1, 2, 4 or 8 (c <= 16): at most 9 wavefronts per workgroup (launch bound 576)
  ZK_LAUNCH(ctx, "msm_accum_final", msm_accum_final_kernel, dim3(nb / 64, (unsigned)ncols), dim3(64), 0, off[g.nlev], nb, list[g.nlev],
            g.cap[g.nlev], dense);
  static const int tail_tree = getenv("AMDZK_TAIL_TREE") ? atoi(getenv("AMDZK_TAIL_TREE")) : -1;  // -1: in latency mode; 0 / 1: never / always
  const bool quad = quad_on && fold_w == 1;
  if (quad)
    ZK_LAUNCH(ctx, "msm_rowcol", msm_rowcol_quad_kernel, dim3(64, 2, (unsigned)ncols), dim3(256), 0, dense, nb, rows, cols);
  else if ((tail_tree < 0 ? g.latency : tail_tree != 0) && ncols <= 65535)
    ZK_LAUNCH(ctx, "msm_rowcol", msm_rowcol_tree_kernel, dim3(64, (unsigned)(2 * ((g.G + 63) / 64)), (unsigned)ncols), dim3(64), 0, dense, nb, rows, cols);
  else
    ZK_LAUNCH(ctx, "msm_rowcol", msm_rowcol_kernel, dim3((unsigned)(2 * ((g.G + 63) / 64)), (unsigned)ncols), dim3(64 * ROWCOL_WAVES), 0, dense, nb, rows, cols);
  if (quad) ZK_LAUNCH(ctx, "msm_fold", msm_fold_quad_kernel, dim3((unsigned)ncols), dim3(512), 0, rows, cols, nb, outp);
  else if (fold_w == 1) ZK_LAUNCH(ctx, "msm_fold", msm_fold_kernel<1>, dim3((unsigned)ncols), dim3(128), 0, rows, cols, nb, outp);
  else ZK_LAUNCH(ctx, "msm_fold", msm_fold_kernel<8>, dim3((unsigned)ncols), dim3(64 * (fold_w + 1)), 0, rows, cols, nb, outp);
  return AMDZK_OK;
}

// ncols MSMs of length len over srs->table[basis]; results (XYZZ) land in d_out[ncols].
// With ctx->msm_pipeline (an experiment: create_proof sets it under AMDZK_MSM_PIPELINE=1) a batch of many columns is cut
// into groups that alternate between the ctx's stream and a second stream of its own, level-1 kernels chained, so that
// a group's latency-bound tail and sort could hide under the next group's level-1 kernel (see msm_group). It does not
// pay on this chip — the small kernels then queue for workgroup slots behind the chip-filling one — and is off by
// default: one group, one stream.
int zk_msm_dev_xyzz(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, const Fr* d_scalars, size_t ncols,
                    size_t len, size_t col_stride, G1X** d_out) {
  if (!srs || basis < 0 || basis > 1 || !srs->table[basis]) ZK_FAIL(ctx, AMDZK_E_INVALID, "msm: basis %d not uploaded", basis);
  if (len > srs->n) ZK_FAIL(ctx, AMDZK_E_INVALID, "msm: len %zu > 2^k = %zu", len, srs->n);
  if (ncols == 0 || ncols > 65535) ZK_FAIL(ctx, AMDZK_E_INVALID, "msm: ncols %zu out of range", ncols);
  if ((uint64_t)srs->W * srs->n >= (1ull << 31)) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "msm: table too large for 31-bit ids");
  // groups: at least AMDZK_MSM_GROUP_COLS columns each (default 16), at most 6
  size_t ngroups = 1;
  if (ctx->msm_pipeline && !ctx->prof) {
    size_t per = 16;
    if (const char* e = getenv("AMDZK_MSM_GROUP_COLS")) per = atoi(e) > 0 ? (size_t)atoi(e) : per;
    ngroups = std::min<size_t>(std::max<size_t>(ncols / per, 1), 6);
  }
  const size_t gcols = (ncols + ngroups - 1) / ngroups;
  ngroups = (ncols + gcols - 1) / gcols;
  const MsmGeom g = msm_geometry(srs, gcols, len, ncols, ctx->msm_latency_mode);  // task sizes as for the whole batch
  const size_t o_out = align_up(ngroups * g.bytes, 256);
  char* ws = nullptr;
  ZK_TRY(zk_ws_reserve(ctx, 1, o_out + align_up(ncols * sizeof(G1X), 256), (void**)&ws));
  G1X* outp = (G1X*)(ws + o_out);
  if (!ctx->msm_l1_evt) ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->msm_l1_evt, hipEventDisableTiming));
  if (ngroups == 1) {
    ZK_TRY(msm_group(ctx, srs, basis, g, ws, d_scalars, ncols, len, col_stride, outp, nullptr, ctx->msm_l1_evt));
    ctx->msm_l1_fresh = true;
    *d_out = outp;
    return AMDZK_OK;
  }
  if (!ctx->msm_stream) ZK_HIP(ctx, zk_stream_create(&ctx->msm_stream, ctx->parent != nullptr));
  for (hipEvent_t& e : ctx->msm_evt)
    if (!e) ZK_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipStream_t main_stream = ctx->stream;
  // the scalars were produced on the ctx's stream: the second stream starts after them
  ZK_HIP(ctx, hipEventRecord(ctx->msm_evt[6], main_stream));
  ZK_HIP(ctx, hipStreamWaitEvent(ctx->msm_stream, ctx->msm_evt[6], 0));
  int rc = AMDZK_OK;
  for (size_t gi = 0; gi < ngroups && rc == AMDZK_OK; gi++) {
    const size_t first = gi * gcols, m = std::min(gcols, ncols - first);
    ctx->stream = (gi & 1) ? ctx->msm_stream : main_stream;
    rc = msm_group(ctx, srs, basis, g, ws + gi * g.bytes, d_scalars + first * col_stride, m, len, col_stride, outp + first,
                   gi ? ctx->msm_evt[gi - 1] : nullptr, gi + 1 < ngroups ? ctx->msm_evt[gi] : ctx->msm_l1_evt);
  }
  ctx->stream = main_stream;
  ZK_TRY(rc);
  ctx->msm_l1_fresh = true;
  ZK_HIP(ctx, hipEventRecord(ctx->msm_evt[7], ctx->msm_stream));
  ZK_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->msm_evt[7], 0));
  *d_out = outp;
  return AMDZK_OK;
}

// Host finish: XYZZ -> normalised Jacobian (z = 1) with one shared inversion (Montgomery's trick).
int zk_msm_finish(amdzk_ctx* ctx, const G1X* d_res, size_t ncols, uint64_t* out_jac) {
  G1X* h = nullptr;
  ZK_TRY(zk_pinned_reserve(ctx, ncols * sizeof(G1X), (void**)&h));
  ZK_HIP(ctx, hipMemcpyAsync(h, d_res, ncols * sizeof(G1X), hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  std::vector<Fq> pre(ncols);
  Fq acc = Fq::one();
  for (size_t i = 0; i < ncols; i++) {
    pre[i] = acc;
    if (!h[i].is_inf()) acc = mul(acc, h[i].zzz);
  }
  acc = inv(acc);
  for (size_t i = ncols; i-- > 0;) {
    G1Jac* o = reinterpret_cast<G1Jac*>(out_jac + 12 * i);
    if (h[i].is_inf()) {
      o->x = Fq::zero();
      o->y = Fq::one();
      o->z = Fq::zero();
      continue;
    }
    Fq izzz = mul(acc, pre[i]);
    acc = mul(acc, h[i].zzz);
    Fq iz = mul(h[i].zz, izzz);
    Fq izz = sqr(iz);
    o->x = mul(h[i].x, izz);
    o->y = mul(h[i].y, izzz);
    o->z = Fq::one();
  }
  return AMDZK_OK;
}
