// Device kernels of the PLONK-specific part of create_proof (halo2_proofs 0.2.0 @ v2023_01_20 [UP],
// /root/reference/Cargo.lock:469-471):
//
//   expr_eval_limbs_kernel — plonk::evaluation::Evaluator::evaluate_h (a7): the whole h(X) numerator as one
//                        straight-line stack program on the quotient cosets, executed by every row in lock step.
//   expr_eval_kernel   — every other "evaluate an expression on all rows" loop of the prover (lookup
//                        compression a9, the numerators/denominators of the permutation and lookup grand
//                        products a8), the same way on the Lagrange domain.
//   batch_invert_kernel, scan_* kernels — BatchInvert + the running products z(X) (a8, a9)
//   poly_eval_kernel   — arithmetic::eval_polynomial for all (polynomial, point) queries at once (a11)
//   lincomb_kernel, kate_div_kernel, small helpers — SHPLONK's polynomial algebra (a12)
//
// Upstream walks a per-row "calculation graph" on CPU threads; here the host compiles the whole h(X)
// numerator (custom gates, permutation and lookup terms, each with its power of y) into ONE postfix
// program whose instruction stream is wave-uniform (scalar loads / scalar branches) while the data
// path is one row per lane: column reads are 32-B-per-lane contiguous, the operand stack lives in LDS
// with the top of stack in registers.
#include <stdlib.h>
#include <string.h>

#include "plonk_kernels.hpp"
#include "fp29.cuh"

using namespace bn254;

namespace {

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

// ------------------------------------------------------------------------------ interpreter
// Shape of both interpreters (what tools/isa_walk.py — a control-flow walk of the compiled kernel — says matters):
//  * the instruction stream is read through the CONSTANT address space: wave-uniform 16-byte scalar loads straight into
//    SGPRs, two instructions ahead, no vector load + wait + readfirstlane in front of every instruction;
//  * the operand of the NEXT instruction (a column's row, or a constant every lane reads alike) is ALWAYS fetched
//    while the current one executes — instructions without an operand name a dummy constant (prover.hip
//    upload_program), and the program ends in two END instructions, so neither fetch is conditional;
//  * hipcc structurizes the (uniform) dispatch, and every loop-carried register is then copied twice per interpreted
//    instruction at the merge blocks: only the top of stack and the operand in flight are loop-carried registers.
__device__ __forceinline__ bool op_reads_col(uint32_t op) {
  return op == OP_PUSH_COL || op == OP_MUL_COL || op == OP_ADD_COL || op == OP_SUB_COL;
}
typedef const ExprInstr __attribute__((address_space(4))) * ExprProgPtr;
struct ExprWord {  // one instruction in scalar registers
  uint32_t op_arg;
  int32_t rot;
  const Fr* ptr;
};
__device__ __forceinline__ ExprWord expr_word(ExprProgPtr prog, uint32_t i) {
  ExprWord w;
  w.op_arg = prog[i].op_arg;
  w.rot = prog[i].rot;
  w.ptr = prog[i].ptr;
  return w;
}
// rows come in blocks of mask + 1 = n (one block in the Lagrange domain, one per coset in the quotient domain): a
// rotation wraps inside the row's own block; operands that are not columns are read at index 0 by every lane
__device__ __forceinline__ Fr expr_fetch(const ExprWord& in, size_t row, size_t blk_base, size_t mask) {
  const size_t sel = op_reads_col(in.op_arg >> 24) ? ~(size_t)0 : 0;
  const size_t idx = (blk_base + ((row + (size_t)(int64_t)in.rot) & mask)) & sel;
  return ld_fr(in.ptr + idx);
}

// Lagrange-domain programs (lookup compression, the fractions of the grand products): columns, constants and results in
// halo2curves' radix-2^256 form, packed values on the stack. (Quotient-domain programs: expr_eval_limbs_kernel below.)
__global__ __launch_bounds__(EXPR_THREADS) void expr_eval_kernel(ExprArgs a) {
  auto fmul = [](const Fr& x, const Fr& y) -> Fr { return fr29_mul_std(x, y); };
  extern __shared__ uint4 lds_raw[];
  Fr* stack = reinterpret_cast<Fr*>(lds_raw);  // [depth][EXPR_THREADS]
  const uint32_t tid = threadIdx.x;
  const size_t row = (size_t)blockIdx.x * EXPR_THREADS + tid;
  if (row >= a.nrows) return;  // domains smaller than one block (no barriers below, so an early exit is safe)
  Fr tos = Fr::zero();
  uint32_t sp = 0;  // elements on the stack, including tos
  const size_t blk_base = row & ~a.mask;
  const uint32_t first = a.nparts ? a.part_start[blockIdx.y] : 0u, prog_len = a.nparts ? a.part_len[blockIdx.y] : a.prog_len;
  const ExprProgPtr prog = (ExprProgPtr)a.prog + first;
  ExprWord cur = expr_word(prog, 0), nxt = expr_word(prog, 1);
  Fr pre = expr_fetch(cur, row, blk_base, a.mask);
  for (uint32_t pc = 0; pc < prog_len; pc++) {
    const uint32_t op = cur.op_arg >> 24, arg = cur.op_arg & 0xffffffu;
    const Fr v = pre;  // operand of this instruction (if it has one), fetched one instruction ago
    const ExprWord nn = expr_word(prog, pc + 2);
    pre = expr_fetch(nxt, row, blk_base, a.mask);
    switch (op) {
      case OP_PUSH_COL:
      case OP_PUSH_CONST:
        if (sp > 0) stack[(sp - 1) * EXPR_THREADS + tid] = tos;
        tos = v;
        sp++;
        break;
      case OP_MUL_COL:
      case OP_MUL_CONST:
        tos = fmul(tos, v);
        break;
      case OP_ADD_COL:
      case OP_ADD_CONST:
        tos = add(tos, v);
        break;
      case OP_SUB_COL:
        tos = sub(tos, v);
        break;
      case OP_ADD:
        tos = add(stack[(sp - 2) * EXPR_THREADS + tid], tos);
        sp--;
        break;
      case OP_SUB:
        tos = sub(stack[(sp - 2) * EXPR_THREADS + tid], tos);
        sp--;
        break;
      case OP_MUL:
        tos = fmul(stack[(sp - 2) * EXPR_THREADS + tid], tos);
        sp--;
        break;
      case OP_NEG:
        tos = neg(tos);
        break;
      case OP_SQR:
        tos = fmul(tos, tos);
        break;
      case OP_STORE:
        st_fr(a.outs[arg] + row, tos);
        sp--;
        if (sp > 0) tos = stack[(sp - 1) * EXPR_THREADS + tid];
        break;
      case OP_PICK:  // entry `arg` below the top; the old top sinks into LDS first (arg = 0 then reads it back: a dup)
        stack[(sp - 1) * EXPR_THREADS + tid] = tos;
        tos = stack[(sp - 1 - arg) * EXPR_THREADS + tid];
        sp++;
        break;
      case OP_NIP:
        sp -= arg;
        break;
      default:
        break;
    }
    cur = nxt;
    nxt = nn;
  }
}

// ------------------------------------------------------------------------------ interpreter, limb-resident
// The h(X) program (radix 2^261 data) spends most of its time in products whose operands and results used to be packed
// to canonical 32-byte values and unpacked again around every operation (81 of 310 instructions per product). Here the
// top of stack and the coset's X values live on 9 x 29-bit limbs (fp29.cuh), lazily reduced: sums
// and differences are limb-wise with a carry pass, products reset the bound to 2p, and the host — which knows the
// whole (wave-uniform) program — tracks every value's bound and inserts OP_REDUCE where a product or a difference
// would leave its range (prover.hip finalize_limb_program). The fold with y, h = sum_j y^(K-1-j) term_j, is a sum of
// products with ONE reduction per group of terms (fp29.cuh f29_wide_*): the host groups the terms by the hot column
// that multiplies them (l_0, l_last, l_active, or none), each term is added as term_j * y^(K-1-j) — 81 multiply-adds
// into 17 un-carried columns, the power a per-proof constant — and a group is reduced, multiplied by its hot column
// and added to h once: 4 reductions and 3 hot-column products per row instead of 332 and 239. The operand of every
// instruction is unpacked at the top of the loop; the operand stack below the top is in LDS as limbs (36 B per
// entry), and so are the 17 columns (8 B each); h is read-modified-written in its output row by the 4 flushes.
// No per-thread arrays, so nothing in scratch.
__device__ __forceinline__ Fr29 lds_ld29(const uint32_t* s, uint32_t slot, uint32_t tid) {
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = s[(slot * 9 + i) * EXPR_THREADS + tid];
  return r;
}
__device__ __forceinline__ void lds_st29(uint32_t* s, uint32_t slot, uint32_t tid, const Fr29& v) {
#pragma unroll
  for (int i = 0; i < 9; i++) s[(slot * 9 + i) * EXPR_THREADS + tid] = v.l[i];
}

__global__ __launch_bounds__(EXPR_THREADS) void expr_eval_limbs_kernel(ExprArgs a) {
  extern __shared__ uint4 lds_raw[];
  // State that only a few instructions touch does not live in registers: hipcc structurizes the (wave-uniform)
  // dispatch, and every loop-carried register is then copied twice per interpreted instruction at the merge blocks
  // (tools/isa_walk.py: 34 + 18 of ~130 overhead instructions were `wide` and h). The 17 un-carried columns of the
  // sum of products sit in LDS (one 64-bit slot per column and lane), h in its output row.
  uint64_t* wide_lds = reinterpret_cast<uint64_t*>(lds_raw);                                  // [17][EXPR_THREADS]
  uint32_t* stack = reinterpret_cast<uint32_t*>(wide_lds + (size_t)17 * EXPR_THREADS);        // [depth][9][EXPR_THREADS]
  const uint32_t tid = threadIdx.x;
  const size_t row = (size_t)blockIdx.x * EXPR_THREADS + tid;
  if (row >= a.nrows) return;  // no barriers below
  Fr29 tos;
#pragma unroll
  for (int i = 0; i < 9; i++) tos.l[i] = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) wide_lds[k * EXPR_THREADS + tid] = 0;
  // hot[3] (the coset's X values: an operand of every permutation factor) stays in registers; hot[0..2] (l_0, l_last,
  // l_active) multiply a whole group of terms once each and are read from memory when that happens — 27 registers
  // that keep three waves per SIMD resident.
  Fr29 hotx = tos;
  if (a.hot[3] != EXPR_NO_SLOT) hotx = fr29_unpack(ld_fr(a.cols[a.hot[3]] + row));
  auto hot = [&](uint32_t k) -> Fr29 { return fr29_unpack(ld_fr(a.cols[a.hot[k]] + row)); };
  uint32_t sp = 0;  // elements on the stack, including tos
  const size_t blk_base = row & ~a.mask;
  // piece blockIdx.y of the program (prover.hip finalize_limb_program) accumulates into its own h
  const uint32_t first = a.nparts ? a.part_start[blockIdx.y] : 0u, prog_len = a.nparts ? a.part_len[blockIdx.y] : a.prog_len;
  const ExprProgPtr prog = (ExprProgPtr)a.prog + first;
  Fr* const h_out = a.h_out + (size_t)blockIdx.y * a.nrows;
  ExprWord cur = expr_word(prog, 0), nxt = expr_word(prog, 1);
  Fr pre = expr_fetch(cur, row, blk_base, a.mask);
  for (uint32_t pc = 0; pc < prog_len; pc++) {
    const uint32_t op = cur.op_arg >> 24, arg = cur.op_arg & 0xffffffu;
    // the operand of this instruction (if it has one) as limbs: unpacking is also what frees `pre` for the next fetch
    const Fr29 x = fr29_unpack(pre);
    const ExprWord nn = expr_word(prog, pc + 2);
    pre = expr_fetch(nxt, row, blk_base, a.mask);
    switch (op) {
      case OP_PUSH_COL:
      case OP_PUSH_CONST:
        if (sp > 0) lds_st29(stack, sp - 1, tid, tos);
        tos = x;
        sp++;
        break;
      case OP_MUL_COL:
      case OP_MUL_CONST:
        tos = f29_mul(tos, x);
        break;
      case OP_ADD_COL:
      case OP_ADD_CONST:
        tos = f29_add(tos, x);
        break;
      case OP_SUB_COL:
        tos = f29_sub3(tos, x);
        break;
      case OP_PUSH_HOT:
        if (sp > 0) lds_st29(stack, sp - 1, tid, tos);
        if (arg == 3) tos = hotx;
        else tos = hot(arg);
        sp++;
        break;
      case OP_MUL_HOT:
        if (arg == 3) tos = f29_mul(tos, hotx);
        else tos = f29_mul(tos, hot(arg));
        break;
      case OP_ADD:
        tos = f29_add(lds_ld29(stack, sp - 2, tid), tos);
        sp--;
        break;
      case OP_SUB:
        tos = f29_sub3(lds_ld29(stack, sp - 2, tid), tos);
        sp--;
        break;
      case OP_SUB_BIG:
        tos = f29_sub10(lds_ld29(stack, sp - 2, tid), tos);
        sp--;
        break;
      case OP_MUL:
        tos = f29_mul(lds_ld29(stack, sp - 2, tid), tos);
        sp--;
        break;
      case OP_NEG:
        tos = f29_neg3(tos);
        break;
      case OP_NEG_BIG:
        tos = f29_neg10(tos);
        break;
      case OP_SQR:
        tos = f29_sqr(tos);
        break;
      case OP_REDUCE:
        tos = f29_reduce_weak(tos);
        break;
      case OP_WACC: {  // wide += tos * y^(K-1-j), the power (radix 2^261, canonical) fetched like a constant
        // column by column through LDS; a column takes six terms of 9 * 2^58 before its carries must move up: the
        // host marks every sixth term of a group (bit 23) and the carry pass rides along with that term (two copies of
        // the loop: as one loop with a flag, every term pays six selects per column)
        if ((arg >> 23) & 1) {
          uint64_t up = 0;
#pragma unroll
          for (int k = 0; k < 17; k++) {
            uint64_t c = wide_lds[k * EXPR_THREADS + tid] + up;
#pragma unroll
            for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) f29_mad_vv(c, tos.l[i], x.l[k - i]);
            if (k < 16) {
              up = c >> 29;
              c &= F29_MASK;
            }
            wide_lds[k * EXPR_THREADS + tid] = c;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 17; k++) {
            uint64_t c = wide_lds[k * EXPR_THREADS + tid];
#pragma unroll
            for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) f29_mad_vv(c, tos.l[i], x.l[k - i]);
            wide_lds[k * EXPR_THREADS + tid] = c;
          }
        }
        sp--;
        if (sp > 0) tos = lds_ld29(stack, sp - 1, tid);
      } break;
      case OP_WFLUSH: {  // h (+)= reduce(wide) * hot;  wide = 0. The group's sum is within the reduction's range (host)
        F29Wide wide;
#pragma unroll
        for (int k = 0; k < 17; k++) {
          wide.c[k] = wide_lds[k * EXPR_THREADS + tid];
          wide_lds[k * EXPR_THREADS + tid] = 0;
        }
        f29_wide_carry(wide);
        Fr29 g = f29_wide_redc<Fr29P>(wide);
        const uint32_t k = arg & 7;
        if (k == 3) g = f29_mul(g, hotx);
        else if (k < 3) g = f29_mul(g, hot(k));
        if (!(arg & 16)) g = f29_add(g, fr29_unpack(ld_fr(h_out + row)));  // bit 4: the first group of the program (piece)
        st_fr(h_out + row, f29_pack_canonical<FrP>(f29_reduce_weak(g)));
      } break;
      case OP_STORE:  // the host reduced tos below 2p
        st_fr(a.outs[arg] + row, f29_pack_canonical<FrP>(tos));
        sp--;
        if (sp > 0) tos = lds_ld29(stack, sp - 1, tid);
        break;
      case OP_PICK:
        lds_st29(stack, sp - 1, tid, tos);
        tos = lds_ld29(stack, sp - 1 - arg, tid);
        sp++;
        break;
      case OP_NIP:
        sp -= arg;
        break;
      default:
        break;
    }
    cur = nxt;
    nxt = nn;
  }
}

// ------------------------------------------------------------------------------ Fr::random on the device
// vanishing::prover::Argument::commit draws the n coefficients of its random polynomial one Fr::random at a time: 8 x
// next_u64 of ChaCha20Rng = one 64-byte ChaCha20 block each (block counter in words 12-13, nonce 0), reduced from
// 512 bits with halo2curves' from_u512: d0 * R^2 + d1 * R^3 in Montgomery arithmetic (d0, d1 the two 256-bit
// halves; bn254.cuh's product accepts a second operand up to 2^256). Thread i produces draw counter0 + i — the same
// values, in the same order, as the host loop this replaces (hostcrypto.hpp ChaCha20Rng::fr, 2.9 ms per proof).
struct ChaChaKey {
  uint32_t k[8];
};
// Draw i of the launch is key-stream block counter0 + (i / cnt) * draw_stride + i % cnt and lands at
// out[(i / cnt) * out_stride + i % cnt]: cnt = n, strides 0 = n consecutive draws into n consecutive elements (the random
// polynomial); cnt = blinding rows per column, draw_stride = draws per column in upstream's order (tails, then the unused
// commitment blinds), out_stride = the column stride = the blinding tails of a batch of columns written in place.
__device__ __forceinline__ void chacha20_fr_random_body(Fr* out, size_t n, const ChaChaKey& key, uint64_t counter0, const Fr& r3, uint32_t cnt,
                                                        uint32_t draw_stride, size_t out_stride) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t grp = i / cnt, r = i % cnt;
  const uint64_t ctr = counter0 + grp * draw_stride + r;
  out += grp * out_stride + r - i;  // out + i below = the element this draw belongs to
  uint32_t c[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3], key.k[4], key.k[5], key.k[6], key.k[7],
                    (uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t x[16];
#pragma unroll
  for (int j = 0; j < 16; j++) x[j] = c[j];
  auto rotl = [](uint32_t v, int s) { return (v << s) | (v >> (32 - s)); };
#define CQR(a, b, cc, d)                        \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);   \
  x[cc] += x[d]; x[b] = rotl(x[b] ^ x[cc], 12); \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);    \
  x[cc] += x[d]; x[b] = rotl(x[b] ^ x[cc], 7);
#pragma unroll 1
  for (int r = 0; r < 10; r++) {
    CQR(0, 4, 8, 12) CQR(1, 5, 9, 13) CQR(2, 6, 10, 14) CQR(3, 7, 11, 15)
    CQR(0, 5, 10, 15) CQR(1, 6, 11, 12) CQR(2, 7, 8, 13) CQR(3, 4, 9, 14)
  }
#undef CQR
  Fr d0, d1;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    d0.l[j] = x[j] + c[j];
    d1.l[j] = x[8 + j] + c[8 + j];
  }
  st_fr(out + i, add(mul(Fr::r2(), d0), mul(r3, d1)));
}
// Two kernels of one body: the random polynomial of the vanishing argument — ONE launch per proof, the first kernel of every
// proof, which is what tools/summarize_prof.py and tools/timeline_single_proof.py cut a trace into proofs by — and the
// blinding tails (several launches per proof).
__global__ __launch_bounds__(256) void chacha20_fr_random_kernel(Fr* out, size_t n, ChaChaKey key, uint64_t counter0, Fr r3) {
  chacha20_fr_random_body(out, n, key, counter0, r3, (uint32_t)(n < 0xffffffffu ? n : 0xffffffffu), 0u, (size_t)0);
}
__global__ __launch_bounds__(256) void chacha20_blind_rows_kernel(Fr* out, size_t n, ChaChaKey key, uint64_t counter0, Fr r3, uint32_t cnt,
                                                                  uint32_t draw_stride, size_t out_stride) {
  chacha20_fr_random_body(out, n, key, counter0, r3, cnt, draw_stride, out_stride);
}

// ------------------------------------------------------------------------------ batch inversion
// In place over a flat array; zeros stay zero (ff::BatchInvert). Montgomery's trick on two levels, one Fermat inversion
// per WORKGROUP: a thread owns BI_E elements (element j of thread t of workgroup w is w * 256 * BI_E + j * 256 + t:
// every access of a wavefront is one contiguous run) and leaves their running products in `scratch`; the 256 thread
// products are prefix- and suffix-scanned in LDS (8 rounds each), thread 0 inverts the workgroup's product — once
// per 256 * BI_E elements, by the binary Euclidean algorithm (a quarter of the Fermat chain's latency) — and every thread gets the inverse of its own product as
// total^-1 * (product of the threads before) * (product of the threads after), then walks its elements backwards.
// 3 products per element + (2 * 8 + 2) per thread + 380 per workgroup. Round 2 ran one Fermat chain per 128
// consecutive elements of one thread: 236 wavefronts for the 59 permutation columns, 0.53 ms of pure latency.
constexpr uint32_t BI_E = 8;
__global__ __launch_bounds__(256) void batch_invert_kernel(Fr* a, Fr* scratch, size_t total) {
  __shared__ Fr pre[256], suf[256];
  const uint32_t t = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 256 * BI_E + t;
  Fr acc = Fr::one();
#pragma unroll 1
  for (uint32_t j = 0; j < BI_E; j++) {
    const size_t i = base + (size_t)j * 256;
    if (i >= total) break;
    st_fr(scratch + i, acc);
    const Fr v = ld_fr(a + i);
    if (!v.is_zero()) acc = fr29_mul_std(acc, v);
  }
  pre[t] = acc;
  suf[t] = acc;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {  // inclusive prefix and suffix products of the thread products
    const Fr p = t >= d ? pre[t - d] : Fr::one();
    const Fr q = t + d < 256 ? suf[t + d] : Fr::one();
    __syncthreads();
    if (t >= d) pre[t] = fr29_mul_std(p, pre[t]);
    if (t + d < 256) suf[t] = fr29_mul_std(suf[t], q);
    __syncthreads();
  }
  __shared__ Fr tot_inv;
  if (t == 0) tot_inv = inv_gcd(pre[255]);  // never zero: zeros were left out of the products. One lane: binary Euclid (bn254.cuh)
  __syncthreads();
  acc = tot_inv;
  if (t > 0) acc = fr29_mul_std(acc, pre[t - 1]);
  if (t < 255) acc = fr29_mul_std(acc, suf[t + 1]);
  // acc = inverse of this thread's product
#pragma unroll 1
  for (uint32_t j = BI_E; j-- > 0;) {
    const size_t i = base + (size_t)j * 256;
    if (i >= total) continue;
    const Fr v = ld_fr(a + i);
    if (v.is_zero()) continue;
    st_fr(a + i, fr29_mul_std(acc, ld_fr(scratch + i)));
    acc = fr29_mul_std(acc, v);
  }
}

// a[i] = a[i] * b[i]
__global__ __launch_bounds__(256) void mul_elem_kernel(Fr* a, const Fr* b, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    st_fr(a + i, fr29_mul_std(ld_fr(a + i), ld_fr(b + i)));
}

// ------------------------------------------------------------------------------ running products
// Exclusive prefix product of each column, three steps. Block = 256 threads x SCAN_E elements.
constexpr int SCAN_E = 8;
constexpr int SCAN_BLOCK = 256 * SCAN_E;

__global__ __launch_bounds__(256) void scan_local_kernel(const Fr* in, Fr* out, Fr* totals, size_t n, size_t col_stride,
                                                         uint32_t nblk) {
  // No per-thread array here: the element values are read again from global memory in the second pass (in == out is
  // allowed: a thread reads each of its own elements before it overwrites it). Round 1 kept them in `Fr m[SCAN_E]`,
  // which hipcc placed in scratch memory (the loops are not fully unrolled, so the index is dynamic); with the 29-bit
  // product inlined, the dead carry-out of its v_mad_u64_u32 was allocated to the very SGPR the scratch addresses
  // were formed in, and the build returned wrong running products (tools/repro_valu_salu_sgpr_waw.hip).
  __shared__ Fr part[256];
  const uint32_t col = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
  const Fr* src = in + (size_t)col * col_stride;
  Fr* dst = out + (size_t)col * col_stride;
  const size_t base = (size_t)blk * SCAN_BLOCK + (size_t)t * SCAN_E;
  Fr tot = Fr::one();
#pragma unroll 1
  for (int i = 0; i < SCAN_E; i++)
    if (base + i < n) tot = i == 0 ? ld_fr(src + base) : fr29_mul_std(tot, ld_fr(src + base + i));
  part[t] = tot;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive product scan
    Fr v = t >= d ? part[t - d] : Fr::one();
    __syncthreads();
    if (t >= d) part[t] = fr29_mul_std(v, part[t]);
    __syncthreads();
  }
  Fr pre = t == 0 ? Fr::one() : part[t - 1];
#pragma unroll 1
  for (int i = 0; i < SCAN_E; i++) {
    if (base + i >= n) break;
    const Fr v = ld_fr(src + base + i);
    st_fr(dst + base + i, pre);
    pre = fr29_mul_std(pre, v);
  }
  if (t == 255) st_fr(totals + (size_t)col * nblk + blk, part[255]);
}

// Per column: totals[b] <- product of the blocks before b; rel[col] <- exclusive prefix at row u
// (relative to a start value of one), needs the local scan result at u.
__global__ void scan_blocks_kernel(Fr* totals, const Fr* local, Fr* rel, uint32_t ncols, uint32_t nblk, size_t col_stride,
                                   size_t u) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncols) return;
  Fr* t = totals + (size_t)col * nblk;
  Fr acc = Fr::one();
  for (uint32_t b = 0; b < nblk; b++) {
    Fr v = ld_fr(t + b);
    st_fr(t + b, acc);
    acc = fr29_mul_std(acc, v);
  }
  if (rel) st_fr(rel + col, fr29_mul_std(ld_fr(t + u / SCAN_BLOCK), ld_fr(local + (size_t)col * col_stride + u)));
}

// start[0] = 1, start[c] = start[c-1] * rel[c-1]  (permutation sets chain through last_z)
__global__ void scan_chain_kernel(const Fr* rel, Fr* start, uint32_t ncols) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Fr acc = Fr::one();
  for (uint32_t c = 0; c < ncols; c++) {
    st_fr(start + c, acc);
    acc = fr29_mul_std(acc, ld_fr(rel + c));
  }
}

// z[col][row] = start[col] * totals[col][blk(row)] * local[col][row]   (start may be null = one)
__global__ __launch_bounds__(256) void scan_apply_kernel(Fr* z, const Fr* totals, const Fr* start, size_t n, size_t col_stride,
                                                         uint32_t nblk) {
  const uint32_t col = blockIdx.y;
  Fr* c = z + (size_t)col * col_stride;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr f = ld_fr(totals + (size_t)col * nblk + i / SCAN_BLOCK);
    if (start) f = fr29_mul_std(f, ld_fr(start + col));
    st_fr(c + i, fr29_mul_std(f, ld_fr(c + i)));
  }
}

// ------------------------------------------------------------------------------ evaluation
// out[q] = polys[q](points[q]), one workgroup per query. Lane t owns coefficients t, t + 256, t + 512, ...:
//   p(x) = sum_t x^t * sum_j c[t + 256 j] * (x^256)^j.
// The inner sums are sums of products with ONE reduction (fp29.cuh f29_wide_*): the powers (x^256)^j, j < 256, are
// built once per workgroup in LDS (thread j raises x^256 to the j-th power), then every term is 81 multiply-adds
// into 17 un-carried columns — 109 instructions per coefficient against 256 for a Horner step (product, carry-
// normalised sum, unpack). Polynomials longer than 2^16 take an outer Horner step per 2^16 coefficients. x^t comes
// from two 16-entry tables (x^a, (x^16)^b) instead of a 16-product exponentiation per lane. Powers are kept as limbs in
// radix 2^261, coefficients enter in the ordinary form (mixed-radix product).
constexpr uint32_t PE_T = 256;  // lanes = stride of a lane's coefficients = entries of the power table
__device__ __forceinline__ Fr29 pe_pow(const Fr29& base, uint32_t e, int bits) {  // base^e, e < 2^bits; radix 2^261 in and out
  Fr29 r = f29_one<Fr29P>();
#pragma unroll 1
  for (int b = bits - 1; b >= 0; b--) {
    r = f29_sqr(r);
    if ((e >> b) & 1) r = f29_mul(r, base);
  }
  return r;
}
// gridDim.y > 1 (polynomials longer than 2^16): workgroup (q, seg) evaluates segment seg alone — v_seg = sum over its 2^16
// coefficients c[2^16 seg + i] x^i — into parts[q * gridDim.y + seg], and poly_eval_combine_kernel folds the segments with
// x^(2^16): a k = 18 proof's ~120 queries were 120 workgroups walking four segments each, 0.79 ms on an otherwise idle chip.
__global__ __launch_bounds__(PE_T) void poly_eval_kernel(const Fr* const* polys, const Fr* points, Fr* out, uint32_t n, Fr* parts) {
  __shared__ uint32_t XP[PE_T][9];   // (x^256)^j
  __shared__ uint32_t XA[16][9], XB[16][9];  // x^a, (x^16)^b
  __shared__ Fr red[PE_T];
  const uint32_t q = blockIdx.x, t = threadIdx.x;
  const Fr* p = polys[q];
  const Fr29 x = f29_mul(fr29_unpack(ld_fr(points + q)), f29_k_in<Fr29P>());  // x * 2^261, below 2p
  Fr29 x16 = x;
#pragma unroll 1
  for (int i = 0; i < 4; i++) x16 = f29_sqr(x16);
  Fr29 x256 = x16;
#pragma unroll 1
  for (int i = 0; i < 4; i++) x256 = f29_sqr(x256);
  const uint32_t J = (n + PE_T - 1) / PE_T;       // coefficients per lane (at most)
  const uint32_t ts = J < PE_T ? J : PE_T;          // power-table entries in use
  const uint32_t nseg = (J + PE_T - 1) / PE_T;
  {
    const Fr29 e = pe_pow(x256, t, 8);  // every lane: the loop is uniform, the products are masked
    if (t < ts) {
#pragma unroll
      for (int i = 0; i < 9; i++) XP[t][i] = e.l[i];
    }
    if (t < 32) {
      const Fr29 f = pe_pow(t < 16 ? x : x16, t & 15, 4);
#pragma unroll
      for (int i = 0; i < 9; i++) (t < 16 ? XA : XB)[t & 15][i] = f.l[i];
    }
  }
  Fr29 xbig = x256;  // (x^256)^256, the step between segments
  if (nseg > 1) {
#pragma unroll 1
    for (int i = 0; i < 8; i++) xbig = f29_sqr(xbig);
  }
  __syncthreads();
  Fr s = Fr::zero();
  if (t < n) {
    Fr29 acc;
#pragma unroll
    for (int i = 0; i < 9; i++) acc.l[i] = 0;
    const bool split = gridDim.y > 1;
#pragma unroll 1
    for (uint32_t seg = split ? blockIdx.y + 1 : nseg; seg-- > (split ? blockIdx.y : 0u);) {
      F29Wide w;
      f29_wide_zero(w);
      const uint32_t j0 = seg * PE_T, jn = (J - j0) < PE_T ? (J - j0) : PE_T;
#pragma unroll 1
      for (uint32_t j = 0; j < jn; j++) {
        const uint32_t idx = t + PE_T * (j0 + j);
        if (idx < n) f29_wide_madd(w, fr29_unpack(ld_fr(p + idx)), XP[j]);
        if (j % 6 == 5) f29_wide_carry(w);
      }
      f29_wide_carry(w);
      const Fr29 part = f29_wide_redc<Fr29P>(w);  // below 256 * 2 / 169.3 + 1 < 4.1 p
      // segments from the top down: acc = acc * (x^256)^256 + part; acc below 2 + 4.1
      acc = (split || seg + 1 == nseg) ? part : f29_add(f29_mul(acc, xbig), part);
    }
    Fr29 xa, xb;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      xa.l[i] = XA[t & 15][i];
      xb.l[i] = XB[t >> 4][i];
    }
    acc = f29_mul(f29_mul(acc, xa), xb);  // * x^t; 6.1 * 2 and 2 * 2 stay far inside the product's range
    s = f29_pack_canonical<FrP>(f29_reduce_weak(acc));
  }
  red[t] = s;
  __syncthreads();
  for (uint32_t d = PE_T / 2; d >= 1; d >>= 1) {
    if (t < d) red[t] = add(red[t], red[t + d]);
    __syncthreads();
  }
  if (t == 0) st_fr(gridDim.y > 1 ? parts + (size_t)q * gridDim.y + blockIdx.y : out + q, red[0]);
}
// out[q] = sum_seg parts[q][seg] * (x_q^65536)^seg, Horner from the top segment down; one lane per query (a handful of products).
__global__ __launch_bounds__(64) void poly_eval_combine_kernel(const Fr* parts, const Fr* points, Fr* out, uint32_t nq, uint32_t nseg) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  Fr xb = ld_fr(points + q);
#pragma unroll 1
  for (int i = 0; i < 16; i++) xb = mul(xb, xb);
  Fr acc = ld_fr(parts + (size_t)q * nseg + nseg - 1);
#pragma unroll 1
  for (uint32_t seg = nseg - 1; seg-- > 0;) acc = add(mul(acc, xb), ld_fr(parts + (size_t)q * nseg + seg));
  st_fr(out + q, acc);
}

// ------------------------------------------------------------------------------ linear combinations
// out[i] = (accumulate ? out[i] : 0) + sum_j coefs[j] * polys[j][i]
// out[i] (+)= sum_j coefs[j] * polys[j][i]. Data x constant: the coefficients are converted once per workgroup
// to radix 2^261 (fp29.cuh, "mixed radix") and staged in LDS, LC_CHUNK at a time.
constexpr uint32_t LC_CHUNK = 256;
// The terms of one output element are summed as UNREDUCED products — 17 un-carried 64-bit columns per lane, 81
// multiply-adds per term, a carry pass every six terms — and reduced once per chunk of coefficients (fp29.cuh
// f29_wide_*): sum_j (d_j * 2^256)(c_j * 2^261) * 2^-261 = (sum_j d_j c_j) * 2^256. 109 instructions per term against
// 308 for product + pack + modular add. T < 256 p^2 per chunk, so the reduction comes out below 2.6p.
// blockIdx.y = part: with gridDim.y > 1 the m terms are cut into gridDim.y runs of `per_part` and part p writes ITS sum to
// out + p * part_stride (lincomb_sum_parts_kernel adds the parts up): a combination of hundreds of polynomials of 2^15
// coefficients is otherwise 128 workgroups walking the whole list one term at a time.
__global__ __launch_bounds__(256) void lincomb_kernel(const Fr* const* polys, const Fr* coefs, uint32_t m, Fr* out, size_t n,
                                                      int accumulate, uint32_t per_part, size_t part_stride) {
  __shared__ uint32_t c261[LC_CHUNK][9];  // the chunk's coefficients in radix 2^261, as limbs (every lane reads the same entry)
  if (gridDim.y > 1) {
    const uint32_t first_term = blockIdx.y * per_part;
    polys += first_term;
    coefs += first_term;
    m = first_term < m ? min(per_part, m - first_term) : 0;
    out += (size_t)blockIdx.y * part_stride;
  }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // every thread of the block walks the same number of rounds, so the barriers below are uniform
  const size_t rounds = n > (size_t)blockIdx.x * blockDim.x ? (n - (size_t)blockIdx.x * blockDim.x + stride - 1) / stride : 0;
  for (uint32_t j0 = 0; j0 < m; j0 += LC_CHUNK) {
    const uint32_t jn = min(LC_CHUNK, m - j0);
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < jn; j += blockDim.x) {
      const Fr29 c = fr29_unpack(fr29_const_to_r261(ld_fr(coefs + j0 + j)));
#pragma unroll
      for (int i = 0; i < 9; i++) c261[j][i] = c.l[i];
    }
    __syncthreads();
    for (size_t r = 0; r < rounds; r++) {
      const size_t i = first + r * stride;
      if (i >= n) continue;
      F29Wide w;
      f29_wide_zero(w);
      for (uint32_t j = 0; j < jn; j++) {
        f29_wide_madd(w, fr29_unpack(ld_fr(polys[j0 + j] + i)), c261[j]);
        if (j % 6 == 5) f29_wide_carry(w);
      }
      f29_wide_carry(w);
      const Fr part = f29_pack_canonical<FrP>(f29_reduce_weak(f29_wide_redc<Fr29P>(w)));
      st_fr(out + i, (accumulate || j0 != 0) ? add(ld_fr(out + i), part) : part);
    }
  }
}
// out[i] (+)= sum_p parts[p * stride + i]
__global__ __launch_bounds__(256) void lincomb_sum_parts_kernel(const Fr* parts, size_t stride, uint32_t nparts, Fr* out, size_t n, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr acc = ld_fr(parts + i);
    for (uint32_t p = 1; p < nparts; p++) acc = add(acc, ld_fr(parts + (size_t)p * stride + i));
    st_fr(out + i, accumulate ? add(ld_fr(out + i), acc) : acc);
  }
}
__global__ __launch_bounds__(256) void scale_kernel(Fr* a, size_t n, Fr c) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    st_fr(a + i, fr29_mul_std(ld_fr(a + i), c));
}

// a[i] -= low[i] for i < m   (subtract a low-degree polynomial)
__global__ void sub_low_kernel(Fr* a, const Fr* low, uint32_t m) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) st_fr(a + i, sub(ld_fr(a + i), ld_fr(low + i)));
}

// ------------------------------------------------------------------------------ Kate division
// In place: a(X) of n coefficients -> q(X) = (a(X) - a(b)) / (X - b), n coefficients with q[n-1] = 0.
// q_i = sum_{j>i} a_j b^(j-i-1): a suffix recurrence. Each workgroup owns KD_BLOCK consecutive
// coefficients: chunk Horner per thread + LDS suffix scan give the block's value relative to its start
// (pass 1, `totals`); a per-polynomial serial pass turns the block values into the carry entering each
// block from above (`carries`); pass 2 repeats the scan and replays every chunk with its carry.
// grid = (blocks per polynomial, polynomials).
constexpr uint32_t KD_THREADS = 256, KD_E = 8, KD_BLOCK = KD_THREADS * KD_E;

// The dividend of polynomial `poly` is srcs[poly] (polys[poly] itself when srcs is null) minus the low-degree
// polynomial lows[poly * low_stride ..] (low_stride coefficients, zero padded; none when lows is null) — the copy and
// the subtraction SHPLONK needs before each of its divisions ride in on the loads.
template <bool WRITE>
__global__ __launch_bounds__(KD_THREADS) void kate_div_kernel(Fr* const* polys, const Fr* roots, uint32_t n, uint32_t nblk, Fr* totals,
                                                              const Fr* carries, const Fr* const* srcs, const Fr* lows, uint32_t low_stride) {
  __shared__ Fr S[KD_THREADS];
  const uint32_t t = threadIdx.x, blk = blockIdx.x, poly = blockIdx.y;
  Fr* a = polys[poly];
  const Fr* src = srcs ? srcs[poly] : a;
  const Fr* low = lows ? lows + (size_t)poly * low_stride : nullptr;
  auto ld_src = [&](uint32_t j) -> Fr {
    Fr v = ld_fr(src + j);
    if (low && j < low_stride) v = sub(v, ld_fr(low + j));
    return v;
  };
  const Fr b = ld_fr(roots + poly);
  const Fr br = fr29_const_to_r261(b);  // the root in radix 2^261: the chunk recurrences are data x constant
  const uint32_t s = blk * KD_BLOCK + t * KD_E, e = min(s + KD_E, n);
  Fr acc = Fr::zero();
  if (s < n) {
    acc = ld_src(e - 1);
    for (uint32_t j = e - 1; j-- > s;) acc = add(fr29_mul_const(acc, br), ld_src(j));
  }
  S[t] = acc;  // chunk value relative to its own start
  __syncthreads();
  Fr pw = pow_u64(b, KD_E);  // b^(E*d) for the current stride d
  for (uint32_t d = 1; d < KD_THREADS; d <<= 1) {
    Fr v = t + d < KD_THREADS ? S[t + d] : Fr::zero();
    __syncthreads();
    S[t] = add(S[t], fr29_mul_std(pw, v));
    __syncthreads();
    pw = sqr(pw);
  }
  if (!WRITE) {
    if (t == 0) st_fr(totals + (size_t)poly * nblk + blk, S[0]);
    return;
  }
  if (s < n) {
    // carry entering this chunk: later chunks of this block + the carry entering the block, moved down
    Fr prev = t + 1 < KD_THREADS ? S[t + 1] : Fr::zero();
    if (carries) prev = add(prev, mul(pow_u64(b, (uint64_t)KD_E * (KD_THREADS - 1 - t)), ld_fr(carries + (size_t)poly * nblk + blk)));
    for (uint32_t j = e; j-- > s;) {
      Fr aj = ld_src(j);
      st_fr(a + j, prev);
      prev = add(aj, fr29_mul_const(prev, br));
    }
  }
}

// carries[blk] = sum_{j >= end of blk} a_j b^(j - end of blk) = totals[blk+1] + b^KD_BLOCK * carries[blk+1]
__global__ void kate_carry_kernel(const Fr* totals, Fr* carries, const Fr* roots, uint32_t nblk, uint32_t npolys) {
  const uint32_t poly = blockIdx.x * blockDim.x + threadIdx.x;
  if (poly >= npolys) return;
  const Fr bl = pow_u64(ld_fr(roots + poly), KD_BLOCK);
  Fr c = Fr::zero();
  for (uint32_t blk = nblk; blk-- > 0;) {
    st_fr(carries + (size_t)poly * nblk + blk, c);
    c = add(ld_fr(totals + (size_t)poly * nblk + blk), fr29_mul_std(bl, c));
  }
}

// ------------------------------------------------------------------------------ lookup permutation
// plonk::lookup::prover::permute_expression_pair on the device: keys are
// canonical field elements (numeric order = upstream's Ord for Fr), sorted by a bitonic network —
// LDS-resident for strides inside a 2048-element tile, one global pass per larger stride.
constexpr uint32_t SORT_TILE = 2048;

__device__ __forceinline__ bool key_less(const Fr& a, const Fr& b) {
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    if (a.l[i] != b.l[i]) return a.l[i] < b.l[i];
  }
  return false;
}
__device__ __forceinline__ bool key_eq(const Fr& a, const Fr& b) { return a == b; }

// k_from..k_to (inclusive, powers of two): runs every (k, j) step with j < SORT_TILE inside LDS.
// full = 1: k from 2 (local sort of each tile); full = 0: only the j < SORT_TILE tail of stage k_to.
// Columns blockIdx.y < ncols_a are data + y * col_stride, the others data_b + (y - ncols_a) * col_stride: two batches of
// columns (the lookups' inputs and their tables) sort in one launch sequence — a workgroup's time is the latency of its
// own ~80 barrier-separated stages whatever the grid, so the second batch rides along.
__global__ __launch_bounds__(1024) void bitonic_lds_kernel(Fr* data, Fr* data_b, uint32_t ncols_a, size_t col_stride, uint32_t n, uint32_t k_from,
                                                           uint32_t k_to) {
  extern __shared__ uint4 lds_raw[];
  Fr* L = reinterpret_cast<Fr*>(lds_raw);
  const uint32_t tile = n < SORT_TILE ? n : SORT_TILE;
  const uint32_t base = blockIdx.x * tile, t = threadIdx.x;
  Fr* col = blockIdx.y < ncols_a ? data + (size_t)blockIdx.y * col_stride : data_b + (size_t)(blockIdx.y - ncols_a) * col_stride;
  for (uint32_t i = t; i < tile; i += blockDim.x) L[i] = ld_fr(col + base + i);
  // A stage with j <= 64 touches, from wavefront w, only elements [128 w', 128 w' + 128) for the two runs of 64 pairs it
  // owns (p and p + blockDim.x): consecutive stages of that kind hand data from a wavefront to itself, and LDS executes a
  // wavefront's accesses in order — the workgroup barrier is only needed next to a stage with j >= 128 (57 of the 78 stages
  // of a 4096-key tile run without one). Needs blockDim.x to be a multiple of 64 and pairs p, p + blockDim.x, ... per thread.
  // The elision below is an argument about 64-lane wavefronts: refuse to build for anything else, and trap a launch whose
  // workgroup is not whole wavefronts (zk_sort_keys2 launches 64, 128, ..., 1024 threads).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "bitonic_lds_kernel: the barrier elision between stages with stride <= 64 assumes 64-lane wavefronts (gfx9 family: gfx950)"
#endif
  if ((blockDim.x & 63u) || __builtin_amdgcn_wavefrontsize() != 64) __builtin_trap();
  bool prev_wide = true;  // the load phase above wrote across wavefronts
  for (uint32_t k = k_from; k <= k_to; k <<= 1) {
    uint32_t jstart = k >> 1;
    if (jstart >= tile) jstart = tile >> 1;
    for (uint32_t j = jstart; j >= 1; j >>= 1) {
      const bool wide = j >= 128;
      if (wide || prev_wide) __syncthreads();
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (uint32_t p = t; p < (tile >> 1); p += blockDim.x) {
        const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i + j;  // j is a power of two
        bool asc = (((base + i) & k) == 0);
        Fr a = L[i], b = L[l];
        bool sw = asc ? key_less(b, a) : key_less(a, b);
        if (sw) {
          L[i] = b;
          L[l] = a;
        }
      }
      prev_wide = wide;
    }
  }
  __syncthreads();
  for (uint32_t i = t; i < tile; i += blockDim.x) st_fr(col + base + i, L[i]);
}

__global__ __launch_bounds__(256) void bitonic_global_kernel(Fr* data, Fr* data_b, uint32_t ncols_a, size_t col_stride, uint32_t n, uint32_t k,
                                                             uint32_t j) {
  Fr* col = blockIdx.y < ncols_a ? data + (size_t)blockIdx.y * col_stride : data_b + (size_t)(blockIdx.y - ncols_a) * col_stride;
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (n >> 1)) return;
  const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i + j;  // j is a power of two
  bool asc = ((i & k) == 0);
  Fr a = ld_fr(col + i), b = ld_fr(col + l);
  bool sw = asc ? key_less(b, a) : key_less(a, b);
  if (sw) {
    st_fr(col + i, b);
    st_fr(col + l, a);
  }
}

// For each row r < u of the sorted input A: rep[r] = 1 if A[r] == A[r-1]; otherwise claim the first
// table position holding that value (used[p] = 1), or raise *err if the value is not in the table.
__global__ __launch_bounds__(256) void lookup_mark_kernel(const Fr* A, const Fr* Ts, size_t col_stride, uint32_t u, uint32_t* rep,
                                                          uint32_t* used, size_t flag_stride, int* err) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, lk = blockIdx.y;
  if (r >= u) return;
  const Fr* a = A + (size_t)lk * col_stride;
  const Fr* ts = Ts + (size_t)lk * col_stride;
  Fr v = ld_fr(a + r);
  bool first = r == 0 || !key_eq(ld_fr(a + r - 1), v);
  rep[(size_t)lk * flag_stride + r] = first ? 0u : 1u;
  if (!first) return;
  uint32_t lo = 0, hi = u;  // lowest p with ts[p] >= v
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (key_less(ld_fr(ts + mid), v)) lo = mid + 1; else hi = mid;
  }
  if (lo >= u || !key_eq(ld_fr(ts + lo), v)) {
    atomicExch(err, (int)lk + 1);
    return;
  }
  used[(size_t)lk * flag_stride + lo] = 1u;
}

// out[i] = number of set (invert = 0) / clear (invert = 1) flags before i, i < cnt; out[cnt] = total. One workgroup per
// column walks it in tiles of 4096 flags — four consecutive flags per lane, so a wavefront's loads cover 1 KiB of
// contiguous memory (a lane used to own cnt / 1024 consecutive flags: 64 cache lines per load, 0.37 ms per launch at
// 2^18 rows) — with a shuffle scan per wavefront, the 16 wavefront totals through LDS and a running carry between tiles.
constexpr int FS_PER = 4;  // 8 per lane measured slower (0.45 against 0.30 ms per k = 18 proof)
__global__ __launch_bounds__(1024) void flag_scan_kernel(const uint32_t* flags, uint32_t* out, uint32_t cnt, size_t stride, int invert) {
  __shared__ uint32_t wsum[2][16];
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  const uint32_t* f = flags + (size_t)blockIdx.x * stride;
  uint32_t* o = out + (size_t)blockIdx.x * stride;
  uint32_t carry = 0;
  int buf = 0;
  for (uint32_t base = 0; base < cnt; base += 1024 * FS_PER, buf ^= 1) {
    const uint32_t i0 = base + FS_PER * t;
    uint32_t v[FS_PER], s = 0;
#pragma unroll
    for (int k = 0; k < FS_PER; k++) {
      const uint32_t idx = i0 + k;
      const uint32_t bit = idx < cnt ? (f[idx] ? 1u : 0u) : 0u;
      v[k] = idx < cnt ? (invert ? 1u - bit : bit) : 0u;
      s += v[k];
    }
    uint32_t inc = s;  // inclusive scan over the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(inc, d, 64);
      if ((int)lane >= d) inc += y;
    }
    if (lane == 63) wsum[buf][wv] = inc;
    __syncthreads();  // one barrier per tile: the totals alternate between two buffers
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) {
      const uint32_t x = wsum[buf][w];
      total += x;
      if (w < wv) before += x;
    }
    uint32_t run = carry + before + inc - s;
#pragma unroll
    for (int k = 0; k < FS_PER; k++) {
      if (i0 + k < cnt) o[i0 + k] = run;
      run += v[k];
    }
    carry += total;
  }
  if (t == 0) o[cnt] = carry;
}

// left[rank] = Ts[p] for every unclaimed table position p (ascending).
__global__ __launch_bounds__(256) void lookup_compact_kernel(const Fr* Ts, size_t col_stride, uint32_t u, const uint32_t* used,
                                                             const uint32_t* rank_left, size_t flag_stride, Fr* left) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, lk = blockIdx.y;
  if (p >= u) return;
  if (used[(size_t)lk * flag_stride + p]) return;
  st_fr(left + (size_t)lk * col_stride + rank_left[(size_t)lk * flag_stride + p], ld_fr(Ts + (size_t)lk * col_stride + p));
}

// S[r] = A[r] on first occurrences; repeated rows take the leftovers, ascending leftovers to
// descending rows (upstream pops repeated rows from the back while walking the BTreeMap upwards).
__global__ __launch_bounds__(256) void lookup_assign_kernel(const Fr* A, const Fr* left, Fr* S, size_t col_stride, uint32_t u,
                                                            const uint32_t* rep, const uint32_t* rank_rep, const uint32_t* rank_left,
                                                            size_t flag_stride, int* err) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, lk = blockIdx.y;
  if (r >= u) return;
  const size_t fo = (size_t)lk * flag_stride, co = (size_t)lk * col_stride;
  const uint32_t R = rank_rep[fo + u];
  if (r == 0 && R != rank_left[fo + u]) atomicExch(err, (int)lk + 1);  // multiset sizes must agree
  if (!rep[fo + r]) {
    st_fr(S + co + r, ld_fr(A + co + r));
  } else {
    uint32_t k = R - 1 - rank_rep[fo + r];
    st_fr(S + co + r, ld_fr(left + co + k));
  }
}

// dst[col][row0 + i] = src[col][i], i < cnt   (blinding rows from a packed host upload)
__global__ void scatter_rows_kernel(Fr* dst, size_t col_stride, size_t row0, const Fr* src, uint32_t cnt, uint32_t ncols) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt * ncols) return;
  uint32_t col = i / cnt, r = i % cnt;
  st_fr(dst + (size_t)col * col_stride + row0 + r, ld_fr(src + i));
}

}  // namespace

// ------------------------------------------------------------------------------ launch wrappers
int zk_expr_eval(amdzk_ctx* ctx, const ExprArgs& a, uint32_t depth, const char* name) {
  size_t shmem = (size_t)(depth ? depth : 1) * EXPR_THREADS * sizeof(Fr);
  const dim3 grid((unsigned)((a.nrows + EXPR_THREADS - 1) / EXPR_THREADS), a.nparts ? a.nparts : 1u), block(EXPR_THREADS);
  if (a.radix261) ZK_FAIL(ctx, AMDZK_E_INVALID, "expr_eval: radix-2^261 programs run on the limb-resident interpreter");
  if (a.nparts > (uint32_t)EXPR_MAX_PARTS) ZK_FAIL(ctx, AMDZK_E_INVALID, "expr_eval: too many program parts");
  if (shmem > 65536) ZK_HIP(ctx, hipFuncSetAttribute((const void*)expr_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  ZK_LAUNCH(ctx, name, expr_eval_kernel, grid, block, shmem, a);
  return AMDZK_OK;
}

// h[row] += h[p * rows + row], p = 1 .. nparts - 1 (canonical values): the pieces of a cut h(X) program
__global__ void sum_parts_kernel(Fr* h, size_t rows, uint32_t nparts) {
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  Fr acc = ld_fr(h + row);
  for (uint32_t p = 1; p < nparts; p++) acc = add(acc, ld_fr(h + (size_t)p * rows + row));
  st_fr(h + row, acc);
}

int zk_expr_eval_limbs(amdzk_ctx* ctx, const ExprArgs& a, uint32_t depth, const char* name) {
  size_t shmem = (size_t)(depth ? depth : 1) * EXPR_THREADS * 9 * sizeof(uint32_t) + (size_t)17 * EXPR_THREADS * sizeof(uint64_t);
  if (a.nparts > (uint32_t)EXPR_MAX_PARTS) ZK_FAIL(ctx, AMDZK_E_INVALID, "expr_eval_limbs: too many program parts");
  if (a.nparts > 1 && !a.h_out) ZK_FAIL(ctx, AMDZK_E_INVALID, "expr_eval_limbs: a cut program needs h_out (nparts x nrows)");
  const dim3 grid((unsigned)((a.nrows + EXPR_THREADS - 1) / EXPR_THREADS), a.nparts ? a.nparts : 1u), block(EXPR_THREADS);
  if (shmem > 65536) ZK_HIP(ctx, hipFuncSetAttribute((const void*)expr_eval_limbs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  ZK_LAUNCH(ctx, name, expr_eval_limbs_kernel, grid, block, shmem, a);
  if (a.nparts > 1)
    ZK_LAUNCH(ctx, "sum_parts", sum_parts_kernel, dim3((unsigned)((a.nrows + 255) / 256)), dim3(256), 0, a.h_out, a.nrows, a.nparts);
  return AMDZK_OK;
}

int zk_chacha20_fr_random(amdzk_ctx* ctx, Fr* d_out, size_t n, const uint32_t key[8], uint64_t counter0, const Fr& r3) {
  ChaChaKey k;
  memcpy(k.k, key, sizeof(k.k));
  if (n) ZK_LAUNCH(ctx, "chacha20_fr_random", chacha20_fr_random_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_out, n, k, counter0, r3);
  return AMDZK_OK;
}

// The blinding tails of ncols columns drawn where they belong: column c (at d_cols + c * col_stride) gets draws
// counter0 + c * draw_stride + [0, cnt) of the ChaCha20 stream into rows [row0, row0 + cnt).
int zk_chacha20_blind_rows(amdzk_ctx* ctx, Fr* d_cols, size_t col_stride, size_t row0, uint32_t cnt, uint32_t ncols, const uint32_t key[8],
                           uint64_t counter0, uint32_t draw_stride, const Fr& r3) {
  ChaChaKey k;
  memcpy(k.k, key, sizeof(k.k));
  const size_t total = (size_t)cnt * ncols;
  if (total) ZK_LAUNCH(ctx, "chacha20_blind_rows", chacha20_blind_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, d_cols + row0, total, k,
                       counter0, r3, cnt, draw_stride, col_stride);
  return AMDZK_OK;
}

int zk_batch_invert(amdzk_ctx* ctx, Fr* d_a, Fr* d_scratch, size_t total) {
  const size_t per_wg = (size_t)256 * BI_E;
  if (total) ZK_LAUNCH(ctx, "batch_invert", batch_invert_kernel, dim3((unsigned)((total + per_wg - 1) / per_wg)), dim3(256), 0, d_a, d_scratch, total);
  return AMDZK_OK;
}

int zk_mul_elem(amdzk_ctx* ctx, Fr* d_a, const Fr* d_b, size_t total) {
  unsigned gx = (unsigned)((total + 255) / 256);
  if (gx > 4096) gx = 4096;
  if (total) ZK_LAUNCH(ctx, "mul_elem", mul_elem_kernel, dim3(gx), dim3(256), 0, d_a, d_b, total);
  return AMDZK_OK;
}

size_t zk_scan_totals_elems(size_t n, size_t ncols) { return ((n + SCAN_BLOCK - 1) / SCAN_BLOCK) * ncols; }

// z[col] = exclusive prefix product of frac[col] (in place), times the chained start value when
// `chain` (permutation sets: start[c] = z[c-1][u]). d_tmp: totals | rel | start.
int zk_running_product(amdzk_ctx* ctx, Fr* d_cols, size_t ncols, size_t n, size_t col_stride, bool chain, size_t u, Fr* d_tmp) {
  if (ncols == 0) return AMDZK_OK;
  const uint32_t nblk = (uint32_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  Fr* totals = d_tmp;
  Fr* rel = totals + (size_t)nblk * ncols;
  Fr* start = rel + ncols;
  ZK_LAUNCH(ctx, "scan_local", scan_local_kernel, dim3(nblk, (unsigned)ncols), dim3(256), 0, d_cols, d_cols, totals, n, col_stride, nblk);
  ZK_LAUNCH(ctx, "scan_blocks", scan_blocks_kernel, dim3((unsigned)((ncols + 63) / 64)), dim3(64), 0, totals, d_cols,
            chain ? rel : (Fr*)nullptr, (uint32_t)ncols, nblk, col_stride, u);
  if (chain) ZK_LAUNCH(ctx, "scan_chain", scan_chain_kernel, dim3(1), dim3(64), 0, rel, start, (uint32_t)ncols);
  unsigned gx = (unsigned)((n + 255) / 256);
  if (gx > 1024) gx = 1024;
  ZK_LAUNCH(ctx, "scan_apply", scan_apply_kernel, dim3(gx, (unsigned)ncols), dim3(256), 0, d_cols, totals, chain ? start : (const Fr*)nullptr, n,
            col_stride, nblk);
  return AMDZK_OK;
}

int zk_poly_eval(amdzk_ctx* ctx, const Fr* const* d_polys, const Fr* d_points, Fr* d_out, size_t nq, uint32_t n) {
  if (!nq) return AMDZK_OK;
  const uint32_t nseg = (n + 65535u) / 65536u;  // PE_T lanes x PE_T power-table entries per segment
  if (nseg <= 1 || nq > 65535) {
    ZK_LAUNCH(ctx, "poly_eval", poly_eval_kernel, dim3((unsigned)nq), dim3(256), 0, d_polys, d_points, d_out, n, (Fr*)nullptr);
    return AMDZK_OK;
  }
  Fr* parts = nullptr;
  ZK_TRY(zk_ws_reserve(ctx, 4, nq * nseg * sizeof(Fr), (void**)&parts));  // slot 4 is also kate_div's and lincomb's: stream order keeps them apart
  ZK_LAUNCH(ctx, "poly_eval", poly_eval_kernel, dim3((unsigned)nq, nseg), dim3(256), 0, d_polys, d_points, d_out, n, parts);
  ZK_LAUNCH(ctx, "poly_eval_combine", poly_eval_combine_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, (const Fr*)parts, d_points, d_out, (uint32_t)nq,
            nseg);
  return AMDZK_OK;
}

int zk_lincomb(amdzk_ctx* ctx, const Fr* const* d_polys, const Fr* d_coefs, uint32_t m, Fr* d_out, size_t n, bool accumulate) {
  if (m == 0) {  // the empty sum
    if (!accumulate && n) ZK_HIP(ctx, hipMemsetAsync(d_out, 0, n * sizeof(Fr), ctx->stream));
    return AMDZK_OK;
  }
  unsigned gx = (unsigned)((n + 255) / 256);
  if (gx > 2048) gx = 2048;
  // few rows, many terms: parts of >= 32 terms until the launch has ~2048 workgroups
  uint32_t nparts = 1;
  while (nparts < 16 && (size_t)gx * nparts * 2 <= 2048 && m / (nparts * 2) >= 32) nparts *= 2;
  if (nparts == 1) {
    ZK_LAUNCH(ctx, "lincomb", lincomb_kernel, dim3(gx), dim3(256), 0, d_polys, d_coefs, m, d_out, n, accumulate ? 1 : 0, m, (size_t)0);
    return AMDZK_OK;
  }
  Fr* parts = nullptr;
  ZK_TRY(zk_ws_reserve(ctx, 4, (size_t)nparts * n * sizeof(Fr), (void**)&parts));  // slot 4 is also kate_div's: stream order keeps them apart
  const uint32_t per_part = (m + nparts - 1) / nparts;
  ZK_LAUNCH(ctx, "lincomb", lincomb_kernel, dim3(gx, nparts), dim3(256), 0, d_polys, d_coefs, m, parts, n, 0, per_part, n);
  ZK_LAUNCH(ctx, "lincomb_sum_parts", lincomb_sum_parts_kernel, dim3(gx), dim3(256), 0, (const Fr*)parts, n, nparts, d_out, n, accumulate ? 1 : 0);
  return AMDZK_OK;
}

int zk_scale(amdzk_ctx* ctx, Fr* d_a, size_t n, const Fr& c) {
  unsigned gx = (unsigned)((n + 255) / 256);
  if (gx > 2048) gx = 2048;
  if (n) ZK_LAUNCH(ctx, "scale", scale_kernel, dim3(gx), dim3(256), 0, d_a, n, c);
  return AMDZK_OK;
}

int zk_sub_low(amdzk_ctx* ctx, Fr* d_a, const Fr* d_low, uint32_t m) {
  if (m) ZK_LAUNCH(ctx, "sub_low", sub_low_kernel, dim3((m + 63) / 64), dim3(64), 0, d_a, d_low, m);
  return AMDZK_OK;
}

int zk_kate_div(amdzk_ctx* ctx, Fr* const* d_polys, const Fr* d_roots, size_t npolys, uint32_t n) {
  return zk_kate_div_from(ctx, d_polys, nullptr, d_roots, nullptr, 0, npolys, n);
}

// d_polys[i] = (d_srcs[i] - low_i) / (X - root_i), low_i = d_lows[i * low_stride ..] (low_stride coefficients each; null: none).
// d_srcs null: in place. A source may feed several quotients; a source must not be one of the OTHER destinations.
int zk_kate_div_from(amdzk_ctx* ctx, Fr* const* d_polys, const Fr* const* d_srcs, const Fr* d_roots, const Fr* d_lows, uint32_t low_stride,
                     size_t npolys, uint32_t n) {
  if (npolys == 0 || n == 0) return AMDZK_OK;
  const uint32_t nblk = (n + KD_BLOCK - 1) / KD_BLOCK;
  dim3 grid(nblk, (unsigned)npolys), block(KD_THREADS);
  if (nblk == 1) {
    ZK_LAUNCH(ctx, "kate_div", kate_div_kernel<true>, grid, block, 0, d_polys, d_roots, n, nblk, (Fr*)nullptr, (const Fr*)nullptr, d_srcs, d_lows,
              low_stride);
    return AMDZK_OK;
  }
  Fr* tmp = nullptr;  // totals | carries
  ZK_TRY(zk_ws_reserve(ctx, 4, 2 * npolys * nblk * sizeof(Fr), (void**)&tmp));
  Fr* totals = tmp;
  Fr* carries = tmp + npolys * nblk;
  ZK_LAUNCH(ctx, "kate_div_totals", kate_div_kernel<false>, grid, block, 0, d_polys, d_roots, n, nblk, totals, (const Fr*)nullptr, d_srcs, d_lows,
            low_stride);
  ZK_LAUNCH(ctx, "kate_div_carry", kate_carry_kernel, dim3((unsigned)((npolys + 63) / 64)), dim3(64), 0, totals, carries, d_roots, nblk,
            (uint32_t)npolys);
  ZK_LAUNCH(ctx, "kate_div", kate_div_kernel<true>, grid, block, 0, d_polys, d_roots, n, nblk, (Fr*)nullptr, (const Fr*)carries, d_srcs, d_lows,
            low_stride);
  return AMDZK_OK;
}

int zk_scatter_rows(amdzk_ctx* ctx, Fr* d_dst, size_t col_stride, size_t row0, const Fr* d_src, uint32_t cnt, uint32_t ncols) {
  uint32_t total = cnt * ncols;
  if (total) ZK_LAUNCH(ctx, "scatter_rows", scatter_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, d_dst, col_stride, row0, d_src, cnt, ncols);
  return AMDZK_OK;
}

// Sort ncols_a columns at d_cols and ncols_b columns at d_cols_b (n keys each, n a power of two, canonical, ascending, in
// place; both batches with the same column stride) in one sequence of launches.
int zk_sort_keys2(amdzk_ctx* ctx, Fr* d_cols, size_t ncols_a, Fr* d_cols_b, size_t ncols_b, uint32_t n, size_t col_stride) {
  const size_t ncols = ncols_a + ncols_b;
  if (ncols == 0 || n < 2) return AMDZK_OK;
  if (ncols > 65535) ZK_FAIL(ctx, AMDZK_E_INVALID, "sort_keys: more than 65535 columns");
  const uint32_t tile = n < SORT_TILE ? n : SORT_TILE;
  const size_t shmem = (size_t)tile * sizeof(Fr);
  const unsigned threads = tile / 2 >= 1024 ? 1024 : (tile / 2 >= 64 ? tile / 2 : 64);
  const uint32_t na = (uint32_t)ncols_a;
  if (shmem > 65536) ZK_HIP(ctx, hipFuncSetAttribute((const void*)bitonic_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  ZK_LAUNCH(ctx, "sort_bitonic_lds", bitonic_lds_kernel, dim3(n / tile, (unsigned)ncols), dim3(threads), shmem, d_cols, d_cols_b, na, col_stride, n, 2u, tile);
  for (uint32_t k = tile << 1; k <= n && k != 0; k <<= 1) {
    for (uint32_t j = k >> 1; j >= tile; j >>= 1)
      ZK_LAUNCH(ctx, "sort_bitonic_global", bitonic_global_kernel, dim3((n / 2 + 255) / 256, (unsigned)ncols), dim3(256), 0, d_cols, d_cols_b, na, col_stride, n,
                k, j);
    ZK_LAUNCH(ctx, "sort_bitonic_lds", bitonic_lds_kernel, dim3(n / tile, (unsigned)ncols), dim3(threads), shmem, d_cols, d_cols_b, na, col_stride, n, k, k);
  }
  return AMDZK_OK;
}
// Sort ncols columns of n (power of two) canonical keys ascending, in place.
int zk_sort_keys(amdzk_ctx* ctx, Fr* d_cols, size_t ncols, uint32_t n, size_t col_stride) {
  return zk_sort_keys2(ctx, d_cols, ncols, nullptr, 0, n, col_stride);
}

// permute_expression_pair for L lookups: A (sorted inputs, in place), Ts (sorted tables), S out.
// flags: 4 u32 arrays of flag_stride per lookup: rep | used | rank_rep | rank_left (flag_stride >= u+1).
int zk_lookup_permute(amdzk_ctx* ctx, Fr* A, Fr* Ts, Fr* S, Fr* left, size_t L, uint32_t n, uint32_t u, uint32_t* flags, size_t flag_stride,
                      int* d_err, size_t tables_presorted) {
  if (L == 0) return AMDZK_OK;
  if (tables_presorted > L) ZK_FAIL(ctx, AMDZK_E_INVALID, "lookup_permute: more presorted tables than lookups");
  uint32_t* rep = flags;
  uint32_t* used = flags + L * flag_stride;
  uint32_t* rank_rep = flags + 2 * L * flag_stride;
  uint32_t* rank_left = flags + 3 * L * flag_stride;
  ZK_TRY(zk_sort_keys2(ctx, A, L, Ts + tables_presorted * n, L - tables_presorted, n, n));  // inputs and unsorted tables together
  ZK_HIP(ctx, hipMemsetAsync(flags, 0, 2 * L * flag_stride * sizeof(uint32_t), ctx->stream));
  dim3 grid((u + 255) / 256, (unsigned)L), block(256);
  ZK_LAUNCH(ctx, "lookup_mark", lookup_mark_kernel, grid, block, 0, A, Ts, (size_t)n, u, rep, used, flag_stride, d_err);
  ZK_LAUNCH(ctx, "lookup_flag_scan", flag_scan_kernel, dim3((unsigned)L), dim3(1024), 0, rep, rank_rep, u, flag_stride, 0);
  ZK_LAUNCH(ctx, "lookup_flag_scan", flag_scan_kernel, dim3((unsigned)L), dim3(1024), 0, used, rank_left, u, flag_stride, 1);
  ZK_LAUNCH(ctx, "lookup_compact", lookup_compact_kernel, grid, block, 0, Ts, (size_t)n, u, used, rank_left, flag_stride, left);
  ZK_LAUNCH(ctx, "lookup_assign", lookup_assign_kernel, grid, block, 0, A, left, S, (size_t)n, u, rep, rank_rep, rank_left, flag_stride, d_err);
  return AMDZK_OK;
}

// ------------------------------------------------------------------------------ C ABI, function by function
// The kernels above under the names SURVEY.md §8(b) lists, on device-resident columns (Montgomery Fr, halo2curves'
// in-memory form). Small operands (points, coefficients, roots, pointer lists) come from the host.
namespace {
int upload_ptrs_and_frs(amdzk_ctx* ctx, const void* const* ptrs, size_t nptrs, const uint64_t* frs, size_t nfrs, size_t extra_frs, void*** d_ptrs,
                        Fr** d_frs) {
  char* ws = nullptr;
  const size_t pbytes = (nptrs * sizeof(void*) + 255) / 256 * 256;
  ZK_TRY(zk_ws_reserve(ctx, 6, pbytes + (nfrs + extra_frs) * sizeof(Fr) + 256, (void**)&ws));
  if (nptrs) ZK_HIP(ctx, hipMemcpyAsync(ws, ptrs, nptrs * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
  if (nfrs) ZK_HIP(ctx, hipMemcpyAsync(ws + pbytes, frs, nfrs * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));  // the host arrays belong to the caller
  *d_ptrs = (void**)ws;
  *d_frs = (Fr*)(ws + pbytes);
  return AMDZK_OK;
}
}  // namespace

extern "C" {

// ff::BatchInvert over n elements in place: every non-zero element is replaced by its inverse, zeros stay zero.
int amdzk_batch_invert_dev(amdzk_ctx* ctx, void* d_a, size_t n) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!d_a && n) ZK_FAIL(ctx, AMDZK_E_INVALID, "batch_invert: null pointer");
  Fr* scratch = nullptr;
  ZK_TRY(zk_ws_reserve(ctx, 7, (n ? n : 1) * sizeof(Fr), (void**)&scratch));
  return zk_batch_invert(ctx, (Fr*)d_a, scratch, n);
}

// poly::batch_invert_assigned [UP] on the device — the step between Circuit::synthesize and the advice commitments
// (SURVEY.md Appendix A step 3): a cell is Assigned::Rational(numerator, denominator) (Trivial(x) = (x, 1), Zero = (0, 1)) and
// evaluates to numerator * denominator^-1, with a zero denominator giving zero (BatchInvert leaves zeros, as upstream's
// `invert().unwrap_or(zero)`). d_den == NULL: every cell is trivial. d_out may be d_num (in place); it must not overlap d_den.
int amdzk_batch_invert_assigned_dev(amdzk_ctx* ctx, const void* d_num, const void* d_den, size_t n, void* d_out) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (n && (!d_num || !d_out)) ZK_FAIL(ctx, AMDZK_E_INVALID, "batch_invert_assigned: null pointer");
  if (!n) return AMDZK_OK;
  if (d_out != d_num) ZK_HIP(ctx, hipMemcpyAsync(d_out, d_num, n * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
  if (!d_den) return AMDZK_OK;
  Fr* ws = nullptr;
  ZK_TRY(zk_ws_reserve(ctx, 7, 2 * n * sizeof(Fr), (void**)&ws));
  ZK_HIP(ctx, hipMemcpyAsync(ws, d_den, n * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
  ZK_TRY(zk_batch_invert(ctx, ws, ws + n, n));
  return zk_mul_elem(ctx, (Fr*)d_out, ws, n);
}

// The running product of permutation::prover::Argument::commit / lookup::prover::commit_product: column c of
// d_cols (n elements at + c * col_stride) is replaced by z with z[0] = 1, z[i] = z[i-1] * f[i-1]. chain != 0 threads the
// permutation argument's last_z through the columns: z_c[0] = z_{c-1}[chain_row] instead of 1.
int amdzk_grand_product_dev(amdzk_ctx* ctx, void* d_cols, size_t ncols, size_t n, size_t col_stride, int chain, size_t chain_row) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!d_cols && ncols) ZK_FAIL(ctx, AMDZK_E_INVALID, "grand_product: null pointer");
  if (ncols > 65535 || col_stride < n || (chain && chain_row >= n)) ZK_FAIL(ctx, AMDZK_E_INVALID, "grand_product: bad shape");
  Fr* tmp = nullptr;
  ZK_TRY(zk_ws_reserve(ctx, 7, (zk_scan_totals_elems(n, ncols) + 2 * ncols + 8) * sizeof(Fr), (void**)&tmp));
  return zk_running_product(ctx, (Fr*)d_cols, ncols, n, col_stride, chain != 0, chain_row, tmp);
}

// arithmetic::eval_polynomial for nq (polynomial, point) pairs: out[q] = d_polys[q](points[q]); polynomials are n
// coefficients on the device (d_polys: host array of device pointers), points and results on the host.
int amdzk_eval_poly_dev(amdzk_ctx* ctx, const void* const* d_polys, const uint64_t* points, size_t nq, uint32_t n, uint64_t* out) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (nq == 0) return AMDZK_OK;
  if (!d_polys || !points || !out) ZK_FAIL(ctx, AMDZK_E_INVALID, "eval_poly: null pointer");
  void** dp = nullptr;
  Fr* df = nullptr;
  ZK_TRY(upload_ptrs_and_frs(ctx, d_polys, nq, points, nq, nq, &dp, &df));
  ZK_TRY(zk_poly_eval(ctx, (const Fr* const*)dp, df, df + nq, nq, n));
  ZK_HIP(ctx, hipMemcpyAsync(out, df + nq, nq * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  return AMDZK_OK;
}

// d_out[i] = (accumulate ? d_out[i] : 0) + sum_j coefs[j] * d_polys[j][i], i < n: the linear combinations of
// multiopen (sum_j y^j P_j, the h(X) fold by x^n, ...). d_out must not alias an input.
int amdzk_poly_axpy_dev(amdzk_ctx* ctx, const void* const* d_polys, const uint64_t* coefs, size_t m, void* d_out, size_t n, int accumulate) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if ((m && (!d_polys || !coefs)) || (!d_out && n)) ZK_FAIL(ctx, AMDZK_E_INVALID, "poly_axpy: null pointer");
  void** dp = nullptr;
  Fr* df = nullptr;
  ZK_TRY(upload_ptrs_and_frs(ctx, d_polys, m, coefs, m, 0, &dp, &df));
  return zk_lincomb(ctx, (const Fr* const*)dp, df, (uint32_t)m, (Fr*)d_out, n, accumulate != 0);
}

// arithmetic::kate_division in place for npolys polynomials of n coefficients: a(X) -> (a(X) - a(root)) / (X - root),
// n coefficients with the top one zero (upstream returns n - 1).
int amdzk_kate_div_dev(amdzk_ctx* ctx, void* const* d_polys, const uint64_t* roots, size_t npolys, uint32_t n) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (npolys == 0) return AMDZK_OK;
  if (!d_polys || !roots) ZK_FAIL(ctx, AMDZK_E_INVALID, "kate_div: null pointer");
  void** dp = nullptr;
  Fr* df = nullptr;
  ZK_TRY(upload_ptrs_and_frs(ctx, (const void* const*)d_polys, npolys, roots, npolys, 0, &dp, &df));
  return zk_kate_div(ctx, (Fr* const*)dp, df, npolys, n);
}

}  // extern "C"
