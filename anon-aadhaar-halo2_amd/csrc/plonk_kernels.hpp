// Interfaces of plonk_kernels.hip used by the prover driver (prover.hip).
#pragma once
#include "common.hpp"

constexpr int EXPR_THREADS = 128;

// Device program words: op << 24 | arg. Column ops: arg = slot << 8 | rotation-table index.
enum ExprOp : uint32_t {
  OP_END = 0,
  OP_PUSH_COL = 1,
  OP_PUSH_CONST = 2,
  OP_ADD = 3,
  OP_SUB = 4,
  OP_MUL = 5,
  OP_NEG = 6,
  OP_MUL_CONST = 7,
  OP_ADD_CONST = 8,
  OP_MUL_COL = 9,
  OP_ADD_COL = 10,
  OP_SUB_COL = 11,
  OP_ACC = 12,    // h = h*y + pop(): host-side programs only (prover.hip finalize_limb_program turns the fold into
                  // OP_WACC / OP_WFLUSH); neither interpreter executes it
  OP_STORE = 13,  // outs[arg][row] = pop()
  OP_SQR = 14,
  OP_PUSH_HOT = 15,  // push hot[arg]
  OP_MUL_HOT = 16,   // tos *= hot[arg]
  // ---- only in programs finalised for the limb-resident interpreter (expr_eval_limbs_kernel, radix 2^261): values
  // stay on 9 x 29-bit limbs, lazily reduced; the host tracks each value's bound and places the reductions
  OP_REDUCE = 17,       // tos = weak reduction of tos (below 1.0002 p)
  OP_SUB_BIG = 18,      // as OP_SUB with a subtrahend (tos) between 2p and 9p
  OP_NEG_BIG = 19,      // as OP_NEG with tos between 2p and 9p
  // h(X) = sum_j y^(K-1-j) term_j is not folded by Horner there: the terms are grouped by the hot column that
  // multiplies them, every term is ADDED, UNREDUCED, into 17 un-carried columns as term_j * y^(K-1-j) (one constant
  // per term, refreshed per proof), and a group is reduced once, multiplied by its hot column once and added to h
  OP_WACC = 20,         // wide += pop() * y^(K-1-j), j = arg & 0x7fffff (ExprInstr::ptr: that power, radix 2^261);
                        // arg bit 23: move the columns' carries up with this term (every sixth term of a group)
  OP_WFLUSH = 21,       // h_out[row] (+)= reduce(wide) * hot[arg & 7]  (4: no factor);  wide = 0;  arg & 16: first group (=, not +=)
  // ---- both interpreters: values shared by several factors stay on the stack and are copied from there (the permutation
  // argument's w_j = (v_j + gamma) / beta serves sigma_j + w_j AND delta^j X + w_j: one product per column instead of two)
  OP_PICK = 22,         // push a copy of the entry `arg` below the top (0: the top itself)
  OP_NIP = 23,          // discard the `arg` entries below the top; the top stays
};

constexpr int EXPR_HOT = 4;
constexpr int EXPR_MAX_PARTS = 8;
constexpr uint32_t EXPR_NO_SLOT = 0xffffffffu;

// One resolved instruction (16 bytes, fetched with a single scalar load): op << 24 | arg, the row
// offset of the operand's rotation (already scaled for the domain), and the operand's base address —
// a column for *_COL ops, the constant itself for *_CONST ops and OP_WACC, a dummy constant for every
// other op (the interpreters fetch the operand unconditionally). A program ends in two OP_END words.
struct ExprInstr {
  uint32_t op_arg;
  int32_t rot;
  const bn254::Fr* ptr;
};

struct ExprArgs {
  const ExprInstr* prog;
  uint32_t prog_len;
  const bn254::Fr* const* cols;  // slot -> column base (device array of device pointers); used for the hot slots
  bn254::Fr* const* outs;        // OP_STORE targets
  bn254::Fr* h_out;              // OP_WFLUSH: h per row, canonical (may be null for programs without it)
  size_t mask;                   // n - 1: rotations wrap inside blocks of n rows (nrows = n, or nc * n in the quotient domain)
  size_t nrows;
  uint32_t hot[EXPR_HOT];        // column slots the hot ops read at rotation 0, or EXPR_NO_SLOT (limb interpreter: slot 3 is
                                 // held in registers for the whole row, 0..2 are read where a group is flushed)
  // 0: columns, constants and results are in halo2curves' radix-2^256 Montgomery form (bn254.cuh product).
  // 1: everything the program touches is in radix 2^261 (32 x the value in the ordinary form): products use
  //    fp29.cuh's in-place 29-bit product. The prover runs the h(X) program this way (extended domain only).
  uint32_t radix261;
  // The program as `nparts` independent pieces (each leaves the stack empty), piece p = instructions
  // [part_start[p], part_start[p] + part_len[p]), run by the workgroups with blockIdx.y = p. The limb interpreter's
  // pieces (the h(X) program cut by the host) each accumulate into h_out + p * nrows, summed into h_out afterwards. n = 2^15 rows are 512 wavefronts — half a wavefront per SIMD walking a long program one dependent
  // product after the other; cut at set / lookup boundaries the same work is 8 x as many wavefronts. 0: the whole program.
  uint32_t nparts;
  uint32_t part_start[EXPR_MAX_PARTS], part_len[EXPR_MAX_PARTS];
};

int zk_expr_eval(amdzk_ctx* ctx, const ExprArgs& a, uint32_t depth, const char* name);
// the same for a radix-2^261 program finalised by the host for the limb-resident interpreter (prover.hip finalize_limb_program)
int zk_expr_eval_limbs(amdzk_ctx* ctx, const ExprArgs& a, uint32_t depth, const char* name);
// d_out[i] = Fr::random of ChaCha20 block counter0 + i under `key` (rand_chacha's ChaCha20Rng, halo2curves' from_u512)
int zk_chacha20_fr_random(amdzk_ctx* ctx, bn254::Fr* d_out, size_t n, const uint32_t key[8], uint64_t counter0, const bn254::Fr& r3);
int zk_chacha20_blind_rows(amdzk_ctx* ctx, bn254::Fr* d_cols, size_t col_stride, size_t row0, uint32_t cnt, uint32_t ncols, const uint32_t key[8],
                           uint64_t counter0, uint32_t draw_stride, const bn254::Fr& r3);
int zk_batch_invert(amdzk_ctx* ctx, bn254::Fr* d_a, bn254::Fr* d_scratch, size_t total);
int zk_mul_elem(amdzk_ctx* ctx, bn254::Fr* d_a, const bn254::Fr* d_b, size_t total);
size_t zk_scan_totals_elems(size_t n, size_t ncols);
int zk_running_product(amdzk_ctx* ctx, bn254::Fr* d_cols, size_t ncols, size_t n, size_t col_stride, bool chain, size_t u,
                       bn254::Fr* d_tmp);
int zk_poly_eval(amdzk_ctx* ctx, const bn254::Fr* const* d_polys, const bn254::Fr* d_points, bn254::Fr* d_out, size_t nq, uint32_t n);
int zk_lincomb(amdzk_ctx* ctx, const bn254::Fr* const* d_polys, const bn254::Fr* d_coefs, uint32_t m, bn254::Fr* d_out, size_t n,
               bool accumulate);
int zk_scale(amdzk_ctx* ctx, bn254::Fr* d_a, size_t n, const bn254::Fr& c);
int zk_sub_low(amdzk_ctx* ctx, bn254::Fr* d_a, const bn254::Fr* d_low, uint32_t m);
int zk_kate_div(amdzk_ctx* ctx, bn254::Fr* const* d_polys, const bn254::Fr* d_roots, size_t npolys, uint32_t n);
int zk_kate_div_from(amdzk_ctx* ctx, bn254::Fr* const* d_polys, const bn254::Fr* const* d_srcs, const bn254::Fr* d_roots, const bn254::Fr* d_lows,
                     uint32_t low_stride, size_t npolys, uint32_t n);
int zk_scatter_rows(amdzk_ctx* ctx, bn254::Fr* d_dst, size_t col_stride, size_t row0, const bn254::Fr* d_src, uint32_t cnt,
                    uint32_t ncols);
int zk_sort_keys(amdzk_ctx* ctx, bn254::Fr* d_cols, size_t ncols, uint32_t n, size_t col_stride);
// two batches of columns (same n and stride) in one sequence of launches
int zk_sort_keys2(amdzk_ctx* ctx, bn254::Fr* d_cols, size_t ncols_a, bn254::Fr* d_cols_b, size_t ncols_b, uint32_t n, size_t col_stride);
int zk_lookup_permute(amdzk_ctx* ctx, bn254::Fr* A, bn254::Fr* Ts, bn254::Fr* S, bn254::Fr* left, size_t L, uint32_t n, uint32_t u,
                      uint32_t* flags, size_t flag_stride, int* d_err, size_t tables_presorted = 0);  // the first tables_presorted
                                                                                                    // columns of Ts are sorted already
