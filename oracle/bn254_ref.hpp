// ORACLE — TEST INFRASTRUCTURE ONLY. Nothing in the product path may include, link or call this.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use oracle/.
//
// PARITY UNPINNED against the reference: anon-aadhaar-halo2 holds no golden vector, known-answer
// test or fixture for this path (every reference test is MockProver), and the
// algorithms live in un-vendored crates that cannot be built here (no cargo/rustc, no network):
//   halo2curves 0.3.1  git tag 0.3.1      #9b67e19b  (/root/reference/Cargo.lock:484-486)
//   halo2_proofs 0.2.0 git PSE v2023_01_20 #c7e42e41 (/root/reference/Cargo.lock:469-471)
// This file restates halo2curves' bn256 Fr/Fq/G1 arithmetic (4 x u64 Montgomery, R = 2^256) from the
// published algorithm. It is pinned by (a) the constants the reference does hold
// (solidity_verifier_contract/contract.sol:210-211 moduli, :82 curve b=3, :440 delta = 7^(2^28)),
// and (b) pure-Python big-integer golden vectors (tests/golden/, generator committed).
#pragma once
#include <stdint.h>
#include <string.h>

namespace oref {

typedef unsigned __int128 u128;

struct FqParams {
  static constexpr uint64_t P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL,
                                    0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr uint64_t R1[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL,
                                     0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
  static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL,
                                     0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
  static constexpr uint64_t INV = 0x87d20782e4866389ULL;
};
struct FrParams {
  static constexpr uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL,
                                    0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr uint64_t R1[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL,
                                     0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
  static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL,
                                     0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
  static constexpr uint64_t INV = 0xc2e1f593efffffffULL;
};

static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t& carry) {
  u128 t = (u128)a + b + carry;
  carry = (uint64_t)(t >> 64);
  return (uint64_t)t;
}
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t& borrow) {
  u128 t = (u128)a - b - borrow;
  borrow = (uint64_t)(t >> 127);
  return (uint64_t)t;
}
static inline uint64_t mac(uint64_t a, uint64_t b, uint64_t c, uint64_t& carry) {
  u128 t = (u128)a + (u128)b * c + carry;
  carry = (uint64_t)(t >> 64);
  return (uint64_t)t;
}

// Field element, Montgomery form, same memory layout as halo2curves `Fr([u64;4])`.
template <class PP>
struct F {
  uint64_t v[4];

  static F zero() { return F{{0, 0, 0, 0}}; }
  static F one() { return F{{PP::R1[0], PP::R1[1], PP::R1[2], PP::R1[3]}}; }
  static F from_raw(const uint64_t a[4]) {  // canonical integer -> Montgomery
    F x{{a[0], a[1], a[2], a[3]}};
    F r2{{PP::R2[0], PP::R2[1], PP::R2[2], PP::R2[3]}};
    return x * r2;
  }
  static F from_u64(uint64_t a) {
    uint64_t t[4] = {a, 0, 0, 0};
    return from_raw(t);
  }
  void to_raw(uint64_t out[4]) const {  // Montgomery -> canonical integer (halo2curves to_repr)
    F o{{1, 0, 0, 0}};
    F r = (*this) * o;
    memcpy(out, r.v, 32);
  }
  bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
  bool operator==(const F& b) const {
    return v[0] == b.v[0] && v[1] == b.v[1] && v[2] == b.v[2] && v[3] == b.v[3];
  }
  bool operator!=(const F& b) const { return !(*this == b); }

  F operator+(const F& b) const {
    uint64_t c = 0;
    F s;
    for (int i = 0; i < 4; i++) s.v[i] = adc(v[i], b.v[i], c);
    return s.sub_p_if_ge();
  }
  F operator-(const F& b) const {
    uint64_t bo = 0;
    F d;
    for (int i = 0; i < 4; i++) d.v[i] = sbb(v[i], b.v[i], bo);
    uint64_t mask = 0 - bo, c = 0;
    for (int i = 0; i < 4; i++) d.v[i] = adc(d.v[i], PP::P[i] & mask, c);
    return d;
  }
  F neg() const { return zero() - *this; }
  F dbl() const { return *this + *this; }

  // halo2curves montgomery_reduce after a schoolbook 4x4 product.
  F operator*(const F& b) const {
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
      uint64_t c = 0;
      for (int j = 0; j < 4; j++) t[i + j] = mac(t[i + j], v[i], b.v[j], c);
      t[i + 4] = c;
    }
    uint64_t carry2 = 0;
    for (int i = 0; i < 4; i++) {
      uint64_t k = t[i] * PP::INV;
      uint64_t c = 0;
      (void)mac(t[i], k, PP::P[0], c);
      for (int j = 1; j < 4; j++) t[i + j] = mac(t[i + j], k, PP::P[j], c);
      t[i + 4] = adc(t[i + 4], carry2, c);
      carry2 = c;
    }
    F r{{t[4], t[5], t[6], t[7]}};
    return r.sub_p_if_ge();
  }
  F square() const { return (*this) * (*this); }

  F pow(const uint64_t e[4]) const {
    F r = one();
    for (int i = 3; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        r = r.square();
        if ((e[i] >> b) & 1) r = r * (*this);
      }
    return r;
  }
  F pow_u64(uint64_t e) const {
    uint64_t ee[4] = {e, 0, 0, 0};
    return pow(ee);
  }
  F invert() const {  // a^(p-2); 0 -> 0
    uint64_t e[4], bo = 0;
    e[0] = sbb(PP::P[0], 2, bo);
    for (int i = 1; i < 4; i++) e[i] = sbb(PP::P[i], 0, bo);
    return pow(e);
  }

 private:
  F sub_p_if_ge() const {
    uint64_t bo = 0;
    F d;
    for (int i = 0; i < 4; i++) d.v[i] = sbb(v[i], PP::P[i], bo);
    return bo ? *this : d;
  }
};

typedef F<FrParams> Fr;
typedef F<FqParams> Fq;

// Fr constants of halo2curves bn256 [UP]; checked in tests against 7^((r-1)/2^28) etc.
static inline Fr fr_from_hex4(uint64_t a3, uint64_t a2, uint64_t a1, uint64_t a0) {
  uint64_t t[4] = {a0, a1, a2, a3};
  return Fr::from_raw(t);
}
static const int FR_S = 28;
static inline Fr fr_root_of_unity() {  // 7^((r-1)/2^28), order 2^28
  return fr_from_hex4(0x03ddb9f5166d18b7ULL, 0x98865ea93dd31f74ULL, 0x3215cf6dd39329c8ULL,
                      0xd34f1ed960c37c9cULL);
}
static inline Fr fr_zeta() {  // halo2curves Fr::ZETA [UP], a primitive cube root of unity
  return fr_from_hex4(0x30644e72e131a029ULL, 0x048b6e193fd84104ULL, 0xcc37a73fec2bc5e9ULL,
                      0xb8ca0b2d36636f23ULL);
}
static inline Fr fr_delta() {  // 7^(2^28); contract.sol:440
  return fr_from_hex4(0x09226b6e22c6f0caULL, 0x64ec26aad4c86e71ULL, 0x5b5f898e5e963f25ULL,
                      0x870e56bbe533e9a2ULL);
}

// ------------------------------------------------------------------ G1: y^2 = x^3 + 3
struct G1Affine {  // (0,0) = identity, as halo2curves
  Fq x, y;
  bool is_identity() const { return x.is_zero() && y.is_zero(); }
};

struct G1 {  // Jacobian, z = 0 identity, as halo2curves `G1 {x,y,z}`
  Fq x, y, z;
  static G1 identity() { return G1{Fq::zero(), Fq::one(), Fq::zero()}; }
  bool is_identity() const { return z.is_zero(); }
  static G1 from_affine(const G1Affine& p) {
    if (p.is_identity()) return identity();
    return G1{p.x, p.y, Fq::one()};
  }
  // dbl-2009-l
  G1 dbl() const {
    if (is_identity()) return *this;
    Fq a = x.square();
    Fq b = y.square();
    Fq c = b.square();
    Fq d = ((x + b).square() - a - c).dbl();
    Fq e = a + a + a;
    Fq f = e.square();
    Fq z3 = (z * y).dbl();
    Fq x3 = f - d.dbl();
    Fq y3 = e * (d - x3) - c.dbl().dbl().dbl();
    return G1{x3, y3, z3};
  }
  // add-2007-bl, complete by case split
  G1 add(const G1& o) const {
    if (is_identity()) return o;
    if (o.is_identity()) return *this;
    Fq z1z1 = z.square();
    Fq z2z2 = o.z.square();
    Fq u1 = x * z2z2;
    Fq u2 = o.x * z1z1;
    Fq s1 = y * z2z2 * o.z;
    Fq s2 = o.y * z1z1 * z;
    if (u1 == u2) {
      if (s1 == s2) return dbl();
      return identity();
    }
    Fq h = u2 - u1;
    Fq i = h.dbl().square();
    Fq j = h * i;
    Fq r = (s2 - s1).dbl();
    Fq v = u1 * i;
    Fq x3 = r.square() - j - v.dbl();
    Fq y3 = r * (v - x3) - (s1 * j).dbl();
    Fq z3 = ((z + o.z).square() - z1z1 - z2z2) * h;
    return G1{x3, y3, z3};
  }
  // madd-2007-bl
  G1 add_mixed(const G1Affine& o) const {
    if (o.is_identity()) return *this;
    if (is_identity()) return from_affine(o);
    Fq z1z1 = z.square();
    Fq u2 = o.x * z1z1;
    Fq s2 = o.y * z1z1 * z;
    if (x == u2) {
      if (y == s2) return dbl();
      return identity();
    }
    Fq h = u2 - x;
    Fq hh = h.square();
    Fq i = hh.dbl().dbl();
    Fq j = h * i;
    Fq r = (s2 - y).dbl();
    Fq v = x * i;
    Fq x3 = r.square() - j - v.dbl();
    Fq y3 = r * (v - x3) - (y * j).dbl();
    Fq z3 = (z + h).square() - z1z1 - hh;
    return G1{x3, y3, z3};
  }
  G1 neg() const { return G1{x, y.neg(), z}; }
  G1Affine to_affine() const {
    if (is_identity()) return G1Affine{Fq::zero(), Fq::zero()};
    Fq zi = z.invert();
    Fq zi2 = zi.square();
    return G1Affine{x * zi2, y * zi2 * zi};
  }
  // double-and-add, scalar as canonical 4 x u64
  G1 mul_raw(const uint64_t e[4]) const {
    G1 acc = identity();
    for (int i = 3; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        acc = acc.dbl();
        if ((e[i] >> b) & 1) acc = acc.add(*this);
      }
    return acc;
  }
};

static inline G1Affine g1_generator() { return G1Affine{Fq::one(), Fq::from_u64(2)}; }
static inline bool g1_on_curve(const G1Affine& p) {
  if (p.is_identity()) return true;
  return p.y.square() == p.x.square() * p.x + Fq::from_u64(3);
}

}  // namespace oref
