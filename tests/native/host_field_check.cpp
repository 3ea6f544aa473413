// CPU check of the product's host+device field/curve header (bn254.cuh, 8 x u32 limbs) against the
// oracle (4 x u64 limbs). Compiled with g++ by tests/test_host_field.py. Exit code 0 = all equal.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../anon-aadhaar-halo2_amd/csrc/bn254.cuh"
#include "../../oracle/bn254_ref.hpp"

static uint64_t sm_state = 12345;
static uint64_t sm() {
  uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
template <class OF> static OF rnd() {
  uint64_t t[4] = {sm(), sm(), sm(), sm() >> 3};
  return OF::from_raw(t);  // from_raw reduces through the Montgomery product; any 253-bit input is fine
}
template <class PF, class OF> static PF conv(const OF& a) { PF r; memcpy(r.l, a.v, 32); return r; }
template <class PF, class OF> static bool same(const PF& a, const OF& b) { return memcmp(a.l, b.v, 32) == 0; }

#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

template <class PF, class OF> static int field_checks(const char* name) {
  OF edge[6] = {OF::zero(), OF::one(), OF::zero() - OF::one(), OF::one() + OF::one(), rnd<OF>(), rnd<OF>()};
  for (int it = 0; it < 20000; it++) {
    OF a = it < 36 ? edge[it % 6] : rnd<OF>();
    OF b = it < 36 ? edge[it / 6] : rnd<OF>();
    PF pa = conv<PF>(a), pb = conv<PF>(b);
    CHECK(same(bn254::mul(pa, pb), a * b));
    CHECK(same(bn254::add(pa, pb), a + b));
    CHECK(same(bn254::sub(pa, pb), a - b));
    CHECK(same(bn254::neg(pa), a.neg()));
    CHECK(same(bn254::sqr(pa), a.square()));
    CHECK(same(bn254::dbl(pa), a.dbl()));
  }
  for (int it = 0; it < 50; it++) {
    OF a = rnd<OF>();
    PF pa = conv<PF>(a);
    CHECK(same(bn254::inv(pa), a.invert()));
    CHECK(same(bn254::inv_gcd(pa), a.invert()));
    uint64_t raw[4]; a.to_raw(raw);
    PF fm = bn254::from_mont(pa);
    CHECK(memcmp(fm.l, raw, 32) == 0);
    CHECK(same(bn254::to_mont(fm), a));
  }
  CHECK(same(bn254::inv(PF::zero()), OF::zero()));
  CHECK(same(bn254::inv_gcd(PF::zero()), OF::zero()));
  CHECK(same(bn254::inv_gcd(PF::one()), OF::one()));
  CHECK(same(bn254::inv_gcd(conv<PF>(OF::zero() - OF::one())), OF::zero() - OF::one()));  // (p-1)^-1 = p-1
  CHECK(same(PF::one(), OF::one()));
  printf("%s ok\n", name);
  return 0;
}

static bool same_pt(const bn254::G1X& x, const oref::G1& j) {
  bn254::G1Affine a = bn254::x_to_affine(x);
  oref::G1Affine b = j.to_affine();
  return memcmp(a.x.l, b.x.v, 32) == 0 && memcmp(a.y.l, b.y.v, 32) == 0;
}

int main() {
  if (field_checks<bn254::Fr, oref::Fr>("Fr")) return 1;
  if (field_checks<bn254::Fq, oref::Fq>("Fq")) return 1;
  // curve ops
  oref::G1Affine g = oref::g1_generator();
  oref::G1 P = oref::G1::from_affine(g), Q = oref::G1::identity();
  bn254::G1Affine pg; memcpy(&pg, &g, 64);
  bn254::G1X X = bn254::x_from_affine(pg), Y = bn254::G1X::inf();
  CHECK(same_pt(bn254::x_add(X, Y), P));
  CHECK(same_pt(bn254::x_add(Y, X), P));
  CHECK(same_pt(bn254::x_add_affine(Y, pg), P));
  CHECK(same_pt(bn254::x_add_affine(X, pg), P.dbl()));        // P + P through the doubling branch
  CHECK(same_pt(bn254::x_add(X, X), P.dbl()));
  CHECK(same_pt(bn254::x_add_affine(X, bn254::a_neg(pg)), oref::G1::identity()));
  CHECK(same_pt(bn254::x_add(X, bn254::x_neg(X)), oref::G1::identity()));
  CHECK(same_pt(bn254::x_dbl_affine(pg), P.dbl()));
  for (int it = 0; it < 300; it++) {
    uint64_t e[4] = {sm(), sm() & 0xffff, 0, 0};
    oref::G1 R = P.mul_raw(e);
    oref::G1Affine ra = R.to_affine();
    bn254::G1Affine pra; memcpy(&pra, &ra, 64);
    Q = Q.add(R);
    Y = (it & 1) ? bn254::x_add_affine(Y, pra) : bn254::x_add(Y, bn254::x_from_affine(pra));
    CHECK(same_pt(Y, Q));
    if (it % 7 == 0) { Q = Q.dbl(); Y = bn254::x_dbl(Y); CHECK(same_pt(Y, Q)); }
  }
  printf("G1 ok\n");
  return 0;
}
