// BN254 field and G1 arithmetic for gfx950 (and the host side of the product path).
//
// Replaces, on the device, what the reference reaches through
//   halo2curves 0.3.1 bn256::{Fr,Fq,G1,G1Affine}   (Cargo.lock:484-486).
// Number format is halo2curves' in-memory format: little-endian limbs, Montgomery form with
// R = 2^256, so a Rust `&[Fr]` / `&[G1Affine]` slice can be handed over byte-for-byte.
// A 4 x u64 little-endian value is the same 32 bytes as the 8 x u32 used here.
//
// Layout choice for CDNA4: 8 x 32-bit limbs. The only wide multiplier the VALU has is
// v_mad_u64_u32 (32x32+64 -> 64); every product below is written so that hipcc emits it.
// Moduli are < 2^254, so the CIOS Montgomery loop never needs the 10th carry word.
//
// All functions are __host__ __device__: the host uses them for the O(1)/O(columns) glue
// (normalising a handful of points, domain constants), never for O(n) work.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__
#else
#define ZK_HD inline
#define ZK_D inline
#endif

namespace bn254 {

struct FqP {
  static ZK_HD constexpr uint32_t p(int i) {
    constexpr uint32_t v[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                               0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return v[i];
  }
  static ZK_HD constexpr uint32_t r1(int i) {  // R mod p  (Montgomery one)
    constexpr uint32_t v[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                               0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return v[i];
  }
  static ZK_HD constexpr uint32_t r2(int i) {  // R^2 mod p
    constexpr uint32_t v[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                               0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    return v[i];
  }
  static constexpr uint32_t inv = 0xe4866389u;  // -p^{-1} mod 2^32
};

struct FrP {
  static ZK_HD constexpr uint32_t p(int i) {
    constexpr uint32_t v[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                               0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return v[i];
  }
  static ZK_HD constexpr uint32_t r1(int i) {
    constexpr uint32_t v[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                               0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return v[i];
  }
  static ZK_HD constexpr uint32_t r2(int i) {
    constexpr uint32_t v[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                               0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    return v[i];
  }
  static constexpr uint32_t inv = 0xefffffffu;
};

template <class P>
struct alignas(16) Fp {
  uint32_t l[8];

  static ZK_HD Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
  }
  static ZK_HD Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = P::r1(i);
    return r;
  }
  static ZK_HD Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = P::r2(i);
    return r;
  }
  ZK_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= l[i];
    return o == 0;
  }
  ZK_HD bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i];
    return o == 0;
  }
  ZK_HD bool operator!=(const Fp& b) const { return !(*this == b); }
};

// r = a - p if a >= p else a   (a < 2p)
template <class P>
ZK_HD Fp<P> reduce_once(const Fp<P>& a) {
  Fp<P> d;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)a.l[i] - P::p(i) - borrow;
    d.l[i] = (uint32_t)t;
    borrow = (uint32_t)(t >> 63);
  }
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
  return r;
}

template <class P>
ZK_HD Fp<P> add(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> s;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)a.l[i] + b.l[i] + carry;
    s.l[i] = (uint32_t)t;
    carry = (uint32_t)(t >> 32);
  }
  // p < 2^254 so a+b < 2^255: no carry out of limb 7.
  return reduce_once(s);
}

template <class P>
ZK_HD Fp<P> sub(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> d;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
    d.l[i] = (uint32_t)t;
    borrow = (uint32_t)(t >> 63);
  }
  uint32_t mask = 0u - borrow;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)d.l[i] + (P::p(i) & mask) + carry;
    d.l[i] = (uint32_t)t;
    carry = (uint32_t)(t >> 32);
  }
  return d;
}

template <class P>
ZK_HD Fp<P> neg(const Fp<P>& a) {
  return sub(Fp<P>::zero(), a);
}

template <class P>
ZK_HD Fp<P> dbl(const Fp<P>& a) {
  return add(a, a);
}

// Montgomery product a*b*R^-1 mod p, CIOS. Device: 32-bit limbs (v_mad_u64_u32). Host: the same
// algorithm over 4 x 64-bit limbs with unsigned __int128 (3-4x faster on x86-64; same canonical result).
// Invariant: after every outer iteration t < 2p < 2^255, so t fits the limbs + a zero top word.
// Only the FIRST operand must be < p; the second may be any 256-bit value.
template <class P>
ZK_HD Fp<P> mul(const Fp<P>& a, const Fp<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t bi = b.l[i];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c = (uint64_t)a.l[j] * bi + t[j] + c;
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    uint32_t t8 = (uint32_t)c;  // 9th limb (the 9th limb of t itself is always 0 here)
    uint32_t m = t[0] * P::inv;
    c = ((uint64_t)m * P::p(0) + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      c = (uint64_t)m * P::p(j) + t[j] + c;
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    t[7] = (uint32_t)c + t8;  // < 2^32 because the shifted sum is < 2p < 2^255
  }
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = t[i];
  return reduce_once(r);
#else
  typedef unsigned __int128 u128;
  uint64_t A[4], B[4], M[4], t[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    A[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
    B[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
    M[i] = (uint64_t)P::p(2 * i) | ((uint64_t)P::p(2 * i + 1) << 32);
  }
  // -p^-1 mod 2^64 from the 32-bit constant by one Newton step: x <- x*(2 + p*x)  (x = -p^-1)
  uint64_t inv64 = (uint64_t)P::inv;
  inv64 = inv64 * (2 + M[0] * inv64);
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)A[j] * B[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    uint64_t t4 = (uint64_t)c;
    uint64_t m = t[0] * inv64;
    c = ((u128)m * M[0] + t[0]) >> 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * M[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    t[3] = (uint64_t)c + t4;
  }
  Fp<P> r;
  for (int i = 0; i < 4; i++) {
    r.l[2 * i] = (uint32_t)t[i];
    r.l[2 * i + 1] = (uint32_t)(t[i] >> 32);
  }
  return reduce_once(r);
#endif
}

template <class P>
ZK_HD Fp<P> sqr(const Fp<P>& a) {
  return mul(a, a);
}

template <class P>
ZK_HD Fp<P> to_mont(const Fp<P>& a) {
  return mul(a, Fp<P>::r2());
}
template <class P>
ZK_HD Fp<P> from_mont(const Fp<P>& a) {
  Fp<P> o = Fp<P>::zero();
  o.l[0] = 1;
  return mul(a, o);
}

// a^e, e given as 8 x u32 little endian (plain integer). Variable time.
template <class P>
ZK_HD Fp<P> pow_u256(const Fp<P>& a, const uint32_t e[8]) {
  Fp<P> r = Fp<P>::one();
  bool started = false;
  for (int i = 7; i >= 0; i--) {
    for (int b = 31; b >= 0; b--) {
      if (started) r = sqr(r);
      if ((e[i] >> b) & 1) {
        r = started ? mul(r, a) : a;
        started = true;
      }
    }
  }
  return r;
}

template <class P>
ZK_HD Fp<P> pow_u64(const Fp<P>& a, uint64_t e) {
  uint32_t ee[8] = {(uint32_t)e, (uint32_t)(e >> 32), 0, 0, 0, 0, 0, 0};
  return pow_u256(a, ee);
}

// a^-1; inv(0) = 0. On the device: Fermat, a^(p-2) (uniform control flow). On the host: the binary extended Euclidean
// algorithm on the canonical integer (about 2 us against 25 us for the 380-product chain — the prover's host side
// inverts a few dozen values per proof between two kernel launches: Lagrange bases and partial-fraction weights of the
// opening sets, the shared denominator of every commitment batch).
template <class P>
ZK_HD Fp<P> inv(const Fp<P>& a) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t e[8];
  uint32_t borrow = 2;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)P::p(i) - borrow;
    e[i] = (uint32_t)t;
    borrow = (uint32_t)(t >> 63);
  }
  return pow_u256(a, e);
#else
  if (a.is_zero()) return a;
  struct U {
    uint64_t w[4];
  };
  auto ld = [](const Fp<P>& f) {
    U u;
    for (int i = 0; i < 4; i++) u.w[i] = (uint64_t)f.l[2 * i] | ((uint64_t)f.l[2 * i + 1] << 32);
    return u;
  };
  auto add = [](U& x, const U& y) {  // x += y (no overflow past 2^256 for the values used here)
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (unsigned __int128)x.w[i] + y.w[i];
      x.w[i] = (uint64_t)c;
      c >>= 64;
    }
  };
  auto sub = [](U& x, const U& y) {  // x -= y, x >= y
    unsigned __int128 b = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 t = (unsigned __int128)x.w[i] - y.w[i] - (uint64_t)b;
      x.w[i] = (uint64_t)t;
      b = (t >> 64) & 1;
    }
  };
  auto ge = [](const U& x, const U& y) {
    for (int i = 3; i >= 0; i--)
      if (x.w[i] != y.w[i]) return x.w[i] > y.w[i];
    return true;
  };
  auto shr1 = [](U& x) {
    for (int i = 0; i < 4; i++) x.w[i] = (x.w[i] >> 1) | (i < 3 ? x.w[i + 1] << 63 : 0);
  };
  auto is_one = [](const U& x) { return x.w[0] == 1 && !(x.w[1] | x.w[2] | x.w[3]); };
  U p;
  for (int i = 0; i < 4; i++) p.w[i] = (uint64_t)P::p(2 * i) | ((uint64_t)P::p(2 * i + 1) << 32);
  U u = ld(a), v = p, x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};  // invariants: x1 * a = u, x2 * a = v (mod p)
  auto halve = [&](U& x) {  // x / 2 mod p
    if (x.w[0] & 1) add(x, p);
    shr1(x);
  };
  while (!is_one(u) && !is_one(v)) {
    while (!(u.w[0] & 1)) {
      shr1(u);
      halve(x1);
    }
    while (!(v.w[0] & 1)) {
      shr1(v);
      halve(x2);
    }
    if (ge(u, v)) {
      sub(u, v);
      if (!ge(x1, x2)) add(x1, p);
      sub(x1, x2);
    } else {
      sub(v, u);
      if (!ge(x2, x1)) add(x2, p);
      sub(x2, x1);
    }
  }
  const U& r = is_one(u) ? x1 : x2;  // (a_mont)^-1 as an integer = a^-1 R^-1; two products by R^2 bring it to a^-1 R
  Fp<P> o;
  for (int i = 0; i < 4; i++) {
    o.l[2 * i] = (uint32_t)r.w[i];
    o.l[2 * i + 1] = (uint32_t)(r.w[i] >> 32);
  }
  return mul(mul(o, Fp<P>::r2()), Fp<P>::r2());
#endif
}

// a^-1 by the binary extended Euclidean algorithm on 8 x u32 limbs, host or device; inv_gcd(0) = 0. Data-dependent
// control flow: on the device this is for ONE lane working while its workgroup waits (batch inversion: one inversion per
// workgroup) — ~750 shift/subtract steps of ~40 dependent instructions against the Fermat chain's 380 products of ~206.
// inv_gcd_plain: the inverse of the 256-bit INTEGER a modulo p, as an integer (no Montgomery factor in or out).
template <class P>
ZK_HD Fp<P> inv_gcd_plain(const Fp<P>& a) {
  if (a.is_zero()) return a;
  uint32_t u[8], v[8], x1[8], x2[8], p[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u[i] = a.l[i];
    p[i] = v[i] = P::p(i);
    x1[i] = x2[i] = 0;
  }
  x1[0] = 1;
  auto add_p = [&](uint32_t* x) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      c += (uint64_t)x[i] + p[i];
      x[i] = (uint32_t)c;
      c >>= 32;
    }
  };
  auto sub = [](uint32_t* x, const uint32_t* y) {  // x -= y, returns the borrow
    uint64_t b = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint64_t t = (uint64_t)x[i] - y[i] - b;
      x[i] = (uint32_t)t;
      b = (t >> 32) & 1;
    }
    return (uint32_t)b;
  };
  auto shr1 = [](uint32_t* x) {
#pragma unroll
    for (int i = 0; i < 7; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
    x[7] >>= 1;
  };
  auto ge = [](const uint32_t* x, const uint32_t* y) {  // x >= y: no borrow out of x - y (fully unrolled: no indexed array)
    uint64_t b = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) b = (((uint64_t)x[i] - y[i] - b) >> 32) & 1;
    return b == 0;
  };
  auto is_one = [](const uint32_t* x) { return x[0] == 1 && !(x[1] | x[2] | x[3] | x[4] | x[5] | x[6] | x[7]); };
  // invariants: x1 * a = u, x2 * a = v (mod p); x1, x2 in [0, p)
  while (!is_one(u) && !is_one(v)) {
    while (!(u[0] & 1)) {
      shr1(u);
      if (x1[0] & 1) add_p(x1);
      shr1(x1);
    }
    while (!(v[0] & 1)) {
      shr1(v);
      if (x2[0] & 1) add_p(x2);
      shr1(x2);
    }
    if (ge(u, v)) {
      sub(u, v);
      if (sub(x1, x2)) add_p(x1);
    } else {
      sub(v, u);
      if (sub(x2, x1)) add_p(x2);
    }
  }
  Fp<P> o;
  const bool first = is_one(u);
#pragma unroll
  for (int i = 0; i < 8; i++) o.l[i] = first ? x1[i] : x2[i];
  return o;
}
// the inverse of the Montgomery representative a R is a^-1 R^-1 as an integer: two products by R^2 make it a^-1 R
template <class P>
ZK_HD Fp<P> inv_gcd(const Fp<P>& a) {
  return mul(mul(inv_gcd_plain(a), Fp<P>::r2()), Fp<P>::r2());
}

using Fr = Fp<FrP>;
using Fq = Fp<FqP>;

// ---------------------------------------------------------------- G1: y^2 = x^3 + 3
// Affine point; (0,0) is the identity (halo2curves G1Affine convention).
struct alignas(16) G1Affine {
  Fq x, y;
  ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
};

// Jacobian point as halo2curves G1 {x,y,z}; z = 0 is the identity.
struct alignas(16) G1Jac {
  Fq x, y, z;
};

// Extended Jacobian ("XYZZ"): x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2. ZZ = 0 is the identity.
// Used for bucket accumulators: mixed add costs 8M+2S and needs no inversion.
struct alignas(16) G1X {
  Fq x, y, zz, zzz;
  static ZK_HD G1X inf() {
    G1X r;
    r.x = Fq::zero();
    r.y = Fq::zero();
    r.zz = Fq::zero();
    r.zzz = Fq::zero();
    return r;
  }
  ZK_HD bool is_inf() const { return zz.is_zero(); }
};

ZK_HD G1X x_from_affine(const G1Affine& p) {
  G1X r;
  if (p.is_inf()) return G1X::inf();
  r.x = p.x;
  r.y = p.y;
  r.zz = Fq::one();
  r.zzz = Fq::one();
  return r;
}

// 2*(affine) -> XYZZ  (mdbl-2008-s-1, a = 0)
ZK_HD G1X x_dbl_affine(const G1Affine& p) {
  if (p.is_inf()) return G1X::inf();
  Fq u = dbl(p.y);
  Fq v = sqr(u);
  Fq w = mul(u, v);
  Fq s = mul(p.x, v);
  Fq xx = sqr(p.x);
  Fq m = add(dbl(xx), xx);
  G1X r;
  r.x = sub(sqr(m), dbl(s));
  r.y = sub(mul(m, sub(s, r.x)), mul(w, p.y));
  r.zz = v;
  r.zzz = w;
  return r;
}

// 2*P in XYZZ (dbl-2008-s-1, a = 0)
ZK_HD G1X x_dbl(const G1X& p) {
  if (p.is_inf()) return p;
  Fq u = dbl(p.y);
  Fq v = sqr(u);
  Fq w = mul(u, v);
  Fq s = mul(p.x, v);
  Fq xx = sqr(p.x);
  Fq m = add(dbl(xx), xx);
  G1X r;
  r.x = sub(sqr(m), dbl(s));
  r.y = sub(mul(m, sub(s, r.x)), mul(w, p.y));
  r.zz = mul(v, p.zz);
  r.zzz = mul(w, p.zzz);
  return r;
}

// acc + affine q   (madd-2008-s), complete: handles acc = inf, q = inf, acc = q, acc = -q.
ZK_HD G1X x_add_affine(const G1X& a, const G1Affine& q) {
  if (q.is_inf()) return a;
  if (a.is_inf()) return x_from_affine(q);
  Fq u2 = mul(q.x, a.zz);
  Fq s2 = mul(q.y, a.zzz);
  Fq p = sub(u2, a.x);
  Fq r = sub(s2, a.y);
  if (p.is_zero()) {
    if (r.is_zero()) return x_dbl_affine(q);
    return G1X::inf();
  }
  Fq pp = sqr(p);
  Fq ppp = mul(p, pp);
  Fq qq = mul(a.x, pp);
  G1X o;
  o.x = sub(sub(sqr(r), ppp), dbl(qq));
  o.y = sub(mul(r, sub(qq, o.x)), mul(a.y, ppp));
  o.zz = mul(a.zz, pp);
  o.zzz = mul(a.zzz, ppp);
  return o;
}

// a + b, both XYZZ (add-2008-s), complete.
ZK_HD G1X x_add(const G1X& a, const G1X& b) {
  if (b.is_inf()) return a;
  if (a.is_inf()) return b;
  Fq u1 = mul(a.x, b.zz);
  Fq u2 = mul(b.x, a.zz);
  Fq s1 = mul(a.y, b.zzz);
  Fq s2 = mul(b.y, a.zzz);
  Fq p = sub(u2, u1);
  Fq r = sub(s2, s1);
  if (p.is_zero()) {
    if (r.is_zero()) return x_dbl(a);
    return G1X::inf();
  }
  Fq pp = sqr(p);
  Fq ppp = mul(p, pp);
  Fq qq = mul(u1, pp);
  G1X o;
  o.x = sub(sub(sqr(r), ppp), dbl(qq));
  o.y = sub(mul(r, sub(qq, o.x)), mul(s1, ppp));
  o.zz = mul(mul(a.zz, b.zz), pp);
  o.zzz = mul(mul(a.zzz, b.zzz), ppp);
  return o;
}

ZK_HD G1Affine a_neg(const G1Affine& p) {
  G1Affine r;
  r.x = p.x;
  r.y = p.y.is_zero() ? p.y : neg(p.y);
  return r;
}

ZK_HD G1X x_neg(const G1X& p) {
  G1X r = p;
  r.y = p.y.is_zero() ? p.y : neg(p.y);
  return r;
}

// XYZZ -> Jacobian {x,y,z} with the same affine value: z = ZZZ/ZZ, so z^2 = ZZ, z^3 = ZZZ.
// Needs one inversion; used only on the host for O(columns) results.
ZK_HD G1Affine x_to_affine(const G1X& p) {
  G1Affine r;
  if (p.is_inf()) {
    r.x = Fq::zero();
    r.y = Fq::zero();
    return r;
  }
  Fq izzz = inv(p.zzz);
  Fq iz = mul(p.zz, izzz);   // 1/z  where z = zzz/zz
  Fq izz = sqr(iz);          // 1/zz
  r.x = mul(p.x, izz);
  r.y = mul(p.y, izzz);
  return r;
}

}  // namespace bn254
