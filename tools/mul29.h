// Experiment (tools/, not part of the product): a BN254 base-field Montgomery product on 9 limbs of 29 bits.
//
// Why: one product of the shipped 8 x 32-bit CIOS (csrc/bn254.cuh) compiles to 128 v_mad_u64_u32 +
// 123 v_lshl_add_u64 + 249 v_mov_b32 — gfx950 wants even-aligned VGPR pairs for 64-bit operands, so the
// 64-bit addend of every multiply-add is rebuilt with two moves. With 29-bit limbs a whole column of
// products (<= 9 of a*b at < 2^60 each plus <= 9 of m*p at < 2^58) fits one 64-bit accumulator, so every
// multiply-add is `acc = x*y + acc` IN PLACE: no carries, no moves inside a column, one 64-bit shift per
// column. R = 2^261 > 4p, so inputs below 4p give an output below 2p and no final conditional subtraction
// is needed (the value stays lazily reduced; limbs come out normalised to 29 bits, the top limb smaller).
//
// Contract of mul29(a, b): limbs of a and b below 2^30 (one lazy limb-wise addition of normalised values is
// allowed), values a*b < 100 p^2; returns a*b*2^-261 mod p as a value in [0, 2p) with limbs < 2^29.
#ifndef AMDZK_TOOLS_MUL29_H
#define AMDZK_TOOLS_MUL29_H
#include <stdint.h>

#if defined(__HIPCC__)
#define MUL29_HD __host__ __device__ __forceinline__
#else
#define MUL29_HD inline
#endif

struct F29 {
  uint32_t l[9];
};

// p = 21888242871839275222246405745257275088696311157297823662689037894645226208583 (contract.sol:210)
#define MUL29_P(i)                                                                                                  \
  ((i) == 0 ? 0x187cfd47u : (i) == 1 ? 0x010460b6u : (i) == 2 ? 0x1c72a34fu : (i) == 3 ? 0x02d522d0u : (i) == 4 ? 0x1585d978u \
   : (i) == 5 ? 0x02db40c0u : (i) == 6 ? 0x00a6e141u : (i) == 7 ? 0x0e5c2634u : 0x0030644eu)
#define MUL29_PINV 0x04866389u  // -p^-1 mod 2^29
#define MUL29_MASK 0x1fffffffu

MUL29_HD F29 mul29(const F29& a, const F29& b) {
  uint32_t m[9];
  F29 r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {  // columns whose Montgomery digit m[k] is still to be determined
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * MUL29_P(k - i);
    m[k] = ((uint32_t)acc * MUL29_PINV) & MUL29_MASK;
    acc += (uint64_t)m[k] * MUL29_P(0);  // low 29 bits are now zero
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {  // columns that produce result limbs
#pragma unroll
    for (int i = k - 8; i <= 8; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * MUL29_P(k - i);
    r.l[k - 9] = (uint32_t)acc & MUL29_MASK;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}

// Packed little-endian 8 x u32 (value < 2^256) <-> 9 x 29-bit limbs.
MUL29_HD F29 unpack29(const uint32_t w[8]) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, lo = bit >> 5, sh = bit & 31;
    uint64_t v = w[lo];
    if (lo + 1 < 8) v |= (uint64_t)w[lo + 1] << 32;
    r.l[i] = (uint32_t)(v >> sh) & MUL29_MASK;
  }
  return r;
}
#endif
