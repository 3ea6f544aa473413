// Host build of the product's 29-bit field / XYZZ code (anon-aadhaar-halo2_amd/csrc/fp29.cuh) against the
// product's 32-bit code (bn254.cuh, itself checked against the oracle by host_field_check.cpp), with the
// documented bounds compiled in as hard failures (FP29_CHECK_BOUNDS).
#define FP29_CHECK_BOUNDS 1
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../anon-aadhaar-halo2_amd/csrc/fp29.cuh"

using namespace bn254;

static unsigned long long S = 0x9E3779B97F4A7C15ULL;
static unsigned long long rnd() {
  S ^= S << 13;
  S ^= S >> 7;
  S ^= S << 17;
  return S;
}
static Fq rand_fq() {  // uniform-ish canonical value, as Montgomery form of something
  Fq a;
  for (int i = 0; i < 8; i++) a.l[i] = (uint32_t)rnd();
  a.l[7] &= 0x0fffffffu;  // < 2^252 < p
  return a;
}
static bool eq(const Fq& a, const Fq& b) { return memcmp(a.l, b.l, 32) == 0; }
static bool eq_aff(const G1Affine& a, const G1Affine& b) { return eq(a.x, b.x) && eq(a.y, b.y); }

int main() {
  int fails = 0;
  // --- field: radix changes and products
  for (int t = 0; t < 20000; t++) {
    Fq x = rand_fq(), y = rand_fq();
    if (t == 0) x = Fq::zero();
    if (t == 1) { x = Fq::one(); y = Fq::one(); }
    if (t == 2) for (int i = 0; i < 8; i++) x.l[i] = FqP::p(i) - (i == 0);  // p - 1
    Fq29 a = fq29_from_r256(x), b = fq29_from_r256(y);
    if (!eq(fq29_to_r256(a), x)) { fails++; printf("radix round trip\n"); }
    if (!eq(fq29_to_r256(f29_mul(a, b)), mul(x, y))) { fails++; printf("mul\n"); }
    if (!eq(fq29_to_r256(f29_add(a, b)), add(x, y))) { fails++; printf("add\n"); }
    if (!eq(fq29_to_r256(f29_sub3(a, b)), sub(x, y))) { fails++; printf("sub3\n"); }
    if (!eq(fq29_to_r256(f29_sub10(a, f29_add(f29_add(b, b), b))), sub(x, add(add(y, y), y)))) { fails++; printf("sub10\n"); }
    if (!eq(fq29_pack_canonical(fq29_unpack(x)), x)) { fails++; printf("pack\n"); }
    // dedicated squaring and the two-products-one-reduction form, on canonical and on lazily reduced operands
    if (!eq(fq29_to_r256(f29_sqr(a)), sqr(x))) { fails++; printf("sqr\n"); }
    {
      Fq z = rand_fq(), w = rand_fq();
      Fq29 c = fq29_from_r256(z), d = fq29_from_r256(w);
      if (!eq(fq29_to_r256(f29_mul2(a, b, c, d)), add(mul(x, y), mul(z, w)))) { fails++; printf("mul2\n"); }
      if (!eq(fq29_to_r256(f29_mul2(a, b, f29_neg3(c), d)), sub(mul(x, y), mul(z, w)))) { fails++; printf("mul2 with neg3\n"); }
      // the shapes the point formulas use: r < 8p, t < 12p, (6p - y) with y < 5p, ppp < 2p
      Fq29 r8 = f29_sub6(a, f29_add(f29_add(b, b), f29_add(b, b))), t12 = f29_sub10(c, f29_add(f29_add(d, d), f29_add(d, d)));
      Fq29 y5 = f29_add(f29_add(c, c), f29_add(c, d));
      Fq r8v = sub(x, add(add(y, y), add(y, y))), t12v = sub(z, add(add(w, w), add(w, w))), y5v = add(add(z, z), add(z, w));
      if (!eq(fq29_to_r256(f29_mul2(r8, t12, f29_neg6(y5), b)), sub(mul(r8v, t12v), mul(y5v, y)))) { fails++; printf("mul2 lazy\n"); }
      if (!eq(fq29_to_r256(f29_sqr(t12)), sqr(t12v))) { fails++; printf("sqr lazy\n"); }
      // the un-carried difference as one operand of a product / of the two-product reduction
      Fq29 tl = f29_sub10_lazy(c, f29_add(f29_add(d, d), f29_add(d, d)));
      if (!eq(fq29_to_r256(f29_mul(tl, b)), mul(t12v, y))) { fails++; printf("mul with lazy difference\n"); }
      if (!eq(fq29_to_r256(f29_mul2(r8, tl, f29_neg6(y5), b)), sub(mul(r8v, t12v), mul(y5v, y)))) { fails++; printf("mul2 with lazy difference\n"); }
    }
    if (fails > 5) return 1;
  }
  {  // sums of products with one reduction (f29_wide_*): sum_j a_j b_j, carried every six terms, against mul/add
    for (int t = 0; t < 300; t++) {
      const int m = 1 + (int)(rnd() % 40);
      F29Wide w;
      f29_wide_zero(w);
      Fq want = Fq::zero();
      for (int j = 0; j < m; j++) {
        Fq x = rand_fq(), y = rand_fq();
        Fq29 a = fq29_from_r256(x), b = fq29_from_r256(y);
        a = fq29_unpack(fq29_pack_canonical(a));  // canonical operands (below p), as the kernels feed them
        b = fq29_unpack(fq29_pack_canonical(b));
        f29_wide_madd(w, a, b.l);
        if (j % 6 == 5) f29_wide_carry(w);
        want = add(want, mul(x, y));
      }
      f29_wide_carry(w);
      Fq29 r = f29_reduce_weak(f29_wide_redc<Fq29P>(w));
      if (!eq(fq29_to_r256(r), want)) { fails++; printf("wide sum of %d products\n", m); if (fails > 5) return 1; }
    }
  }
  printf("Fq29 field ok\n");
  // --- scalar field, mixed radix as ntt.hip uses it: data in radix 2^256 times a constant kept in radix 2^261
  for (int t = 0; t < 20000; t++) {
    Fr x, c;
    for (int i = 0; i < 8; i++) {
      x.l[i] = (uint32_t)rnd();
      c.l[i] = (uint32_t)rnd();
    }
    x.l[7] &= 0x0fffffffu;
    c.l[7] &= 0x0fffffffu;
    if (t == 0) x = Fr::zero();
    if (t == 1) c = Fr::one();
    if (t == 2) for (int i = 0; i < 8; i++) x.l[i] = FrP::p(i) - (i == 0);  // r - 1
    if (memcmp(fr29_mul_std(x, c).l, mul(x, c).l, 32) != 0) { fails++; printf("fr29_mul_std\n"); if (fails > 5) return 1; }
    Fr c261 = fr29_const_to_r261(c);
    Fr want = mul(x, c);
    Fr got = fr29_mul_const(x, c261);
    if (memcmp(got.l, want.l, 32) != 0) { fails++; printf("Fr mixed-radix product\n"); if (fails > 5) return 1; }
    // negating the packed constant commutes with the radix change (tw_lookup negates table entries)
    if (!c.is_zero()) {
      Fr got_n = fr29_mul_const(x, neg(c261)), want_n = mul(x, neg(c));
      if (memcmp(got_n.l, want_n.l, 32) != 0) { fails++; printf("Fr negated constant\n"); if (fails > 5) return 1; }
    }
  }
  for (int t = 0; t < 300; t++) {  // inversion chain on the 29-bit product = bn254.cuh's inv
    Fr x;
    for (int i = 0; i < 8; i++) x.l[i] = (uint32_t)rnd();
    x.l[7] &= 0x0fffffffu;
    if (t == 0) x = Fr::one();
    if (t == 1) for (int i = 0; i < 8; i++) x.l[i] = FrP::p(i) - (i == 0);  // r - 1
    if (memcmp(fr29_from_mont(x).l, from_mont(x).l, 32) != 0) { fails++; printf("fr29_from_mont\n"); if (fails > 5) return 1; }
    Fr got = fr29_inv(x), want = inv(x);
    if (memcmp(got.l, want.l, 32) != 0) { fails++; printf("fr29_inv\n"); if (fails > 5) return 1; }
  }
  printf("Fr29 mixed radix ok\n");
  // --- weak reduction, both fields: any normalised 9-limb value -> same residue, below 2p, normalised
  {
    auto check = [&](auto tag, const char* name) {
      using P = decltype(tag);
      for (int t = 0; t < 40000; t++) {
        Fp29<P> v;
        for (int i = 0; i < 9; i++) v.l[i] = (uint32_t)rnd() & F29_MASK;
        if (t % 4 == 1) v.l[8] = (uint32_t)(rnd() % 170) * (P::p(8) + 1) + (uint32_t)(rnd() % 5) - 2;  // quotient boundaries
        if (t % 4 == 2) v.l[8] = (uint32_t)rnd() & 0x3ffffff;                                            // the range the NTT produces
        if (t == 3) for (int i = 0; i < 9; i++) v.l[i] = F29_MASK;
        if (t == 7) for (int i = 0; i < 9; i++) v.l[i] = 0;
        v.l[8] &= F29_MASK;
        Fp29<P> r = f29_reduce_weak(v);
        for (int i = 0; i < 8; i++)
          if (r.l[i] >> 29) { fails++; printf("%s reduce_weak: limb not normalised\n", name); }
        // below 2p: r - 2p must borrow
        int64_t acc = 0;
        for (int i = 0; i < 9; i++) { acc += (int64_t)r.l[i] - 2 * (int64_t)P::p(i); acc >>= 29; }
        if (acc >= 0) { fails++; printf("%s reduce_weak: result >= 2p\n", name); }
        // same residue: v * 1 and r * 1 through the product (both operands in range: v < 169p, one < p)
        Fp29<P> a = f29_mul(v, f29_one<P>()), b = f29_mul(r, f29_one<P>());
        bool same = true;
        {
          // compare canonical forms
          Fp29<P> ca = f29_reduce_weak(a), cb = f29_reduce_weak(b);
          auto canon = [](Fp29<P> x) {
            int64_t bw = 0;
            Fp29<P> t2;
            for (int i = 0; i < 9; i++) { int64_t d = (int64_t)x.l[i] - (int64_t)P::p(i) + bw; t2.l[i] = i < 8 ? ((uint32_t)d & F29_MASK) : (uint32_t)d; bw = d >> 29; }
            return bw < 0 ? x : t2;
          };
          ca = canon(ca);
          cb = canon(cb);
          for (int i = 0; i < 9; i++) same = same && ca.l[i] == cb.l[i];
        }
        if (!same) { fails++; printf("%s reduce_weak: residue changed\n", name); }
        if (fails > 5) return;
      }
    };
    check(Fr29P{}, "Fr");
    check(Fq29P{}, "Fq");
    if (fails) return 1;
  }
  printf("weak reduction ok\n");
  // --- ntt.hip's radix-4 blocks (f29_dif4 / f29_dif4_last): the lazy first-round sums against the same block with every
  // intermediate normalised, on random inputs and on the largest operands the kernel's bounds allow (elements below 4p
  // with all-ones low limbs, canonical twiddles with all-ones low limbs); the bound checks are hard failures here
  {
    auto canon = [](Fr29 x) {
      x = f29_reduce_weak(f29_norm(x));
      int64_t bw = 0;
      Fr29 t2;
      for (int i = 0; i < 9; i++) { int64_t d = (int64_t)x.l[i] - (int64_t)Fr29P::p(i) + bw; t2.l[i] = i < 8 ? ((uint32_t)d & F29_MASK) : (uint32_t)d; bw = d >> 29; }
      return bw < 0 ? x : t2;
    };
    auto same = [&](const Fr29& a, const Fr29& b) {
      Fr29 ca = canon(a), cb = canon(b);
      for (int i = 0; i < 9; i++) if (ca.l[i] != cb.l[i]) return false;
      return true;
    };
    auto rnd_elem = [&](int mode, uint32_t top_units) {  // normalised, below top_units * p
      Fr29 v;
      for (int i = 0; i < 8; i++) v.l[i] = mode == 1 ? F29_MASK : mode == 2 ? 0u : ((uint32_t)rnd() & F29_MASK);
      const uint32_t lim = top_units * Fr29P::p(8) - 1;  // top limb strictly below top_units * p8: value below top_units * p
      v.l[8] = mode == 1 ? lim : mode == 2 ? 0u : (uint32_t)(rnd() % (lim + 1));
      return v;
    };
    for (int t = 0; t < 30000; t++) {
      const int mx = t < 81 ? t : -1;  // the first 81 cases walk the extreme / zero patterns of the four elements
      Fr29 x[4], tw[3];
      for (int j = 0; j < 4; j++) x[j] = rnd_elem(mx < 0 ? 0 : (mx / (j == 0 ? 1 : j == 1 ? 3 : j == 2 ? 9 : 27)) % 3, 4);
      for (int j = 0; j < 3; j++) tw[j] = rnd_elem(t % 5 == 0 ? 1 : 0, 1);
      // reference: the block with normalised intermediates
      const Fr29 s0 = f29_add(x[0], x[2]), s1 = f29_add(x[1], x[3]);
      const Fr29 d0 = f29_mul(f29_sub10_lazy(x[0], x[2]), tw[0]), d1 = f29_mul(f29_sub10_lazy(x[1], x[3]), tw[1]);
      const Fr29 r0 = f29_reduce_weak(f29_add(s0, s1)), r1 = f29_mul(f29_sub10_lazy(s0, s1), tw[2]), r2 = f29_add(d0, d1),
                 r3 = f29_mul(f29_sub10_lazy(d0, d1), tw[2]);
      Fr29 y0 = x[0], y1 = x[1], y2 = x[2], y3 = x[3];
      f29_dif4(y0, y1, y2, y3, tw[0], tw[1], tw[2]);
      bool ok = same(y0, r0) && same(y1, r1) && same(y2, r2) && same(y3, r3);
      for (int i = 0; i < 8; i++) ok = ok && !((y0.l[i] | y1.l[i] | y2.l[i] | y3.l[i]) >> 29);
      ok = ok && y0.l[8] <= Fr29P::p(8) + 176 && y1.l[8] < 2 * Fr29P::p(8) + 2 && y2.l[8] < 4 * Fr29P::p(8) + 4 && y3.l[8] < 2 * Fr29P::p(8) + 2;
      // the last block
      const Fr29 e0 = f29_sub10(x[0], x[2]), e1 = f29_mul(f29_sub10_lazy(x[1], x[3]), tw[0]);
      const Fr29 q1 = f29_sub10(s0, s1), q2 = f29_add(e0, e1), q3 = f29_sub10(e0, e1);
      Fr29 z0 = x[0], z1 = x[1], z2 = x[2], z3 = x[3];
      f29_dif4_last(z0, z1, z2, z3, tw[0]);
      ok = ok && same(z0, r0) && same(z1, q1) && same(z2, q2) && same(z3, q3);
      for (int i = 0; i < 8; i++) ok = ok && !((z0.l[i] | z1.l[i] | z2.l[i] | z3.l[i]) >> 29);
      ok = ok && z1.l[8] < 18 * Fr29P::p(8) + 18 && z2.l[8] < 16 * Fr29P::p(8) + 16 && z3.l[8] < 24 * Fr29P::p(8) + 24;
      // and each result still goes through the product / weak reduction it leaves the tile by
      (void)f29_mul(z1, tw[1]); (void)f29_mul(z2, tw[1]); (void)f29_mul(z3, tw[1]);
      (void)f29_reduce_weak(z1); (void)f29_reduce_weak(z2); (void)f29_reduce_weak(z3);
      if (!ok) { fails++; printf("radix-4 block %d\n", t); if (fails > 5) return 1; }
    }
    if (fails) return 1;
  }
  printf("radix-4 blocks ok\n");
  // --- points: k*G for small k with the 32-bit code
  G1Affine g;
  g.x = Fq::one();
  g.y = add(Fq::one(), Fq::one());
  std::vector<G1Affine> pts;
  {
    G1X acc = x_from_affine(g);
    for (int k = 1; k <= 64; k++) {
      pts.push_back(x_to_affine(acc));
      acc = x_add_affine(acc, g);
    }
  }
  auto conv = [](const G1Affine& p, Fq29& x, Fq29& y) {  // canonical radix-2^261 table entry, as msm.hip stores it
    x = fq29_unpack(fq29_pack_canonical(fq29_from_r256(p.x)));
    y = fq29_unpack(fq29_pack_canonical(fq29_from_r256(p.y)));
  };
  for (int t = 0; t < 3000; t++) {
    G1X a32 = G1X::inf();
    G1X29 a29 = G1X29::inf();
    const int len = 1 + (int)(rnd() % 40);
    for (int i = 0; i < len; i++) {
      G1Affine q = pts[rnd() % pts.size()];
      const unsigned mode = (unsigned)(rnd() % 16);
      if (mode == 0) q = G1Affine{Fq::zero(), Fq::zero()};          // identity
      if (mode == 1 && !a32.is_inf()) q = x_to_affine(a32);         // acc + acc -> doubling
      if (mode == 2 && !a32.is_inf()) {                              // acc + (-acc) -> identity
        q = x_to_affine(a32);
        q.y = neg(q.y);
      }
      const bool negate = (rnd() & 1) != 0 && !q.is_inf();
      if (negate) q.y = neg(q.y);
      Fq29 qx, qy;
      conv(q, qx, qy);
      a32 = x_add_affine(a32, q);
      a29 = x29_add_affine(a29, qx, qy, q.is_inf());
      for (int j = 0; j < 8; j++)
        if (a29.x.l[j] >> 29 || a29.y.l[j] >> 29 || a29.zz.l[j] >> 29 || a29.zzz.l[j] >> 29) { fails++; printf("limb not normalised\n"); }
    }
    G1X back = x29_to_r256(a29);
    if (a32.is_inf() != back.is_inf() || (!a32.is_inf() && !eq_aff(x_to_affine(a32), x_to_affine(back)))) {
      fails++;
      printf("point chain mismatch at test %d\n", t);
      if (fails > 5) return 1;
    }
  }
  // --- XYZZ + XYZZ and doubling: random trees of partial sums, as the fold levels build them
  auto lift = [&](const G1Affine& q) {
    Fq29 qx, qy;
    conv(q, qx, qy);
    return x29_add_affine(G1X29::inf(), qx, qy, q.is_inf());
  };
  for (int t = 0; t < 3000; t++) {
    std::vector<G1X> v32;
    std::vector<G1X29> v29;
    const int m = 2 + (int)(rnd() % 12);
    for (int i = 0; i < m; i++) {  // leaves: short chains, so ZZ != 1 and the lazy bounds are exercised
      G1X a = G1X::inf();
      G1X29 b = G1X29::inf();
      const int len = (int)(rnd() % 4);
      for (int j = 0; j < len; j++) {
        G1Affine q = pts[rnd() % pts.size()];
        if (rnd() & 1) q.y = neg(q.y);
        Fq29 qx, qy;
        conv(q, qx, qy);
        a = x_add_affine(a, q);
        b = x29_add_affine(b, qx, qy, false);
      }
      v32.push_back(a);
      v29.push_back(b);
    }
    while (v32.size() > 1) {
      size_t i = rnd() % v32.size(), j = rnd() % v32.size();
      const unsigned mode = (unsigned)(rnd() % 8);
      G1X r32;
      G1X29 r29;
      if (mode == 0) {  // doubling
        r32 = x_dbl(v32[i]);
        r29 = x29_dbl(v29[i]);
        j = i;
      } else if (mode == 1) {  // a + a through the addition formula
        r32 = x_add(v32[i], v32[i]);
        r29 = x29_add(v29[i], v29[i]);
        j = i;
      } else if (mode == 2 && !v32[i].is_inf()) {  // a + (-a) with a different representative of -a
        G1Affine q = x_to_affine(v32[i]);
        q.y = neg(q.y);
        r32 = x_add(v32[i], x_from_affine(q));
        r29 = x29_add(v29[i], lift(q));
        j = i;
      } else {
        if (i == j) continue;
        r32 = x_add(v32[i], v32[j]);
        r29 = x29_add(v29[i], v29[j]);
      }
      for (int q = 0; q < 8; q++)
        if (r29.x.l[q] >> 29 || r29.y.l[q] >> 29 || r29.zz.l[q] >> 29 || r29.zzz.l[q] >> 29) { fails++; printf("limb not normalised (tree)\n"); }
      G1X back = x29_to_r256(r29);
      if (r32.is_inf() != back.is_inf() || (!r32.is_inf() && !eq_aff(x_to_affine(r32), x_to_affine(back)))) {
        fails++;
        printf("x29_add/x29_dbl mismatch at test %d mode %u\n", t, mode);
        if (fails > 5) return 1;
      }
      v32[i] = r32;
      v29[i] = r29;
      if (j != i) {
        v32.erase(v32.begin() + j);
        v29.erase(v29.begin() + j);
      }
      if (mode <= 2 && v32.size() > 1 && (rnd() & 3) == 0) {
        v32.pop_back();
        v29.pop_back();
      }
    }
  }
  printf(fails ? "FAILED\n" : "G1X29 ok\n");
  return fails != 0;
}
