// ORACLE — TEST INFRASTRUCTURE ONLY (see bn254_ref.hpp header: "parity unpinned" vs the reference;
// pinned by contract.sol constants + pure-Python golden vectors).
//
// CPU restatement of the arithmetic layer of
//   halo2_proofs 0.2.0 @ PSE v2023_01_20 (#c7e42e41, /root/reference/Cargo.lock:469-471)
//     src/arithmetic.rs : best_multiexp, multiexp_serial, best_fft, eval_polynomial,
//                         kate_division, parallelize                          [UP]
// which the reference's circuits reach through `create_proof` (SURVEY.md §3.2, §8(a) rows a1,a3,a11,a12).
// [UP] = upstream source is not in /root/reference; restated from the published algorithm.
//
// C ABI (ctypes-friendly). All field elements: 4 x u64 little endian, Montgomery (R=2^256).
#include <math.h>
#include <omp.h>
#include <stdlib.h>

#include <vector>

#include "bn254_ref.hpp"

using namespace oref;

extern "C" {

// ---------------------------------------------------------------- scalar helpers (tests)
void oracle_fr_mul(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  const Fr* A = (const Fr*)a;
  const Fr* B = (const Fr*)b;
  Fr* O = (Fr*)out;
  for (size_t i = 0; i < n; i++) O[i] = A[i] * B[i];
}
void oracle_fr_add(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  const Fr* A = (const Fr*)a;
  const Fr* B = (const Fr*)b;
  Fr* O = (Fr*)out;
  for (size_t i = 0; i < n; i++) O[i] = A[i] + B[i];
}
void oracle_fr_sub(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  const Fr* A = (const Fr*)a;
  const Fr* B = (const Fr*)b;
  Fr* O = (Fr*)out;
  for (size_t i = 0; i < n; i++) O[i] = A[i] - B[i];
}
void oracle_fr_inv(const uint64_t* a, uint64_t* out, size_t n) {
  const Fr* A = (const Fr*)a;
  Fr* O = (Fr*)out;
  for (size_t i = 0; i < n; i++) O[i] = A[i].invert();
}
void oracle_fr_from_raw(const uint64_t* a, uint64_t* out, size_t n) {
  Fr* O = (Fr*)out;
  for (size_t i = 0; i < n; i++) O[i] = Fr::from_raw(a + 4 * i);
}
void oracle_fr_to_raw(const uint64_t* a, uint64_t* out, size_t n) {
  const Fr* A = (const Fr*)a;
  for (size_t i = 0; i < n; i++) A[i].to_raw(out + 4 * i);
}
void oracle_fq_mul(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  const Fq* A = (const Fq*)a;
  const Fq* B = (const Fq*)b;
  Fq* O = (Fq*)out;
  for (size_t i = 0; i < n; i++) O[i] = A[i] * B[i];
}
void oracle_fq_from_raw(const uint64_t* a, uint64_t* out, size_t n) {
  Fq* O = (Fq*)out;
  for (size_t i = 0; i < n; i++) O[i] = Fq::from_raw(a + 4 * i);
}
void oracle_fq_to_raw(const uint64_t* a, uint64_t* out, size_t n) {
  const Fq* A = (const Fq*)a;
  for (size_t i = 0; i < n; i++) A[i].to_raw(out + 4 * i);
}
void oracle_fr_constants(uint64_t* root_of_unity, uint64_t* zeta, uint64_t* delta) {
  *(Fr*)root_of_unity = fr_root_of_unity();
  *(Fr*)zeta = fr_zeta();
  *(Fr*)delta = fr_delta();
}
// omega for a domain of size 2^k: ROOT_OF_UNITY^(2^(S-k))   (EvaluationDomain::new [UP])
void oracle_fr_omega(uint32_t k, uint64_t* out) {
  Fr w = fr_root_of_unity();
  for (uint32_t i = k; i < (uint32_t)FR_S; i++) w = w.square();
  *(Fr*)out = w;
}

// ---------------------------------------------------------------- G1 helpers
void oracle_g1_generator(uint64_t* out) { *(G1Affine*)out = g1_generator(); }
int oracle_g1_on_curve(const uint64_t* p) { return g1_on_curve(*(const G1Affine*)p) ? 1 : 0; }
// out[i] = scalars[i] * base  (affine out).  scalars Montgomery Fr.
void oracle_g1_mul_many(const uint64_t* base, const uint64_t* scalars, uint64_t* out, size_t n) {
  const G1Affine B = *(const G1Affine*)base;
  const Fr* S = (const Fr*)scalars;
  G1Affine* O = (G1Affine*)out;
#pragma omp parallel for schedule(dynamic, 16)
  for (size_t i = 0; i < n; i++) {
    uint64_t e[4];
    S[i].to_raw(e);
    O[i] = G1::from_affine(B).mul_raw(e).to_affine();
  }
}
// Test SRS bases g[i] = tau^i * G computed incrementally in the exponent, then one fixed-base
// multiplication each (what ParamsKZG::setup does with a window table [UP]; same values).
void oracle_srs_powers(const uint64_t* tau, uint64_t* out_g, size_t n) {
  std::vector<Fr> pw(n);
  Fr t = *(const Fr*)tau, cur = Fr::one();
  for (size_t i = 0; i < n; i++) {
    pw[i] = cur;
    cur = cur * t;
  }
  uint64_t gen[8];
  oracle_g1_generator(gen);
  oracle_g1_mul_many(gen, (const uint64_t*)pw.data(), out_g, n);
}
void oracle_g1_add_affine(const uint64_t* a, const uint64_t* b, uint64_t* out) {
  *(G1Affine*)out = G1::from_affine(*(const G1Affine*)a).add_mixed(*(const G1Affine*)b).to_affine();
}
void oracle_g1_jac_to_affine(const uint64_t* jac, uint64_t* out, size_t n) {
  const G1* J = (const G1*)jac;
  G1Affine* O = (G1Affine*)out;
  for (size_t i = 0; i < n; i++) O[i] = J[i].to_affine();
}

// ---------------------------------------------------------------- MSM
// Naive sum of double-and-add products: the independent cross-check for Pippenger.
void oracle_msm_naive(const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t* out_affine) {
  const Fr* S = (const Fr*)scalars;
  const G1Affine* B = (const G1Affine*)bases;
  G1 acc = G1::identity();
  for (size_t i = 0; i < n; i++) {
    uint64_t e[4];
    S[i].to_raw(e);
    acc = acc.add(G1::from_affine(B[i]).mul_raw(e));
  }
  *(G1Affine*)out_affine = acc.to_affine();
}

// arithmetic.rs multiexp_serial [UP]: c = 1 (<4), 3 (<32), else ceil(ln n); segments = 256/c + 1;
// per segment, high to low: c doublings of acc, bucket fill (mixed adds), running-sum fold.
static size_t get_at(size_t segment, size_t c, const uint8_t bytes[32]) {
  size_t skip_bits = segment * c;
  size_t skip_bytes = skip_bits / 8;
  if (skip_bytes >= 32) return 0;
  uint8_t v[8] = {0};
  for (size_t i = 0; i < 8 && skip_bytes + i < 32; i++) v[i] = bytes[skip_bytes + i];
  uint64_t tmp;
  memcpy(&tmp, v, 8);
  tmp >>= skip_bits - skip_bytes * 8;
  tmp = tmp % ((uint64_t)1 << c);
  return (size_t)tmp;
}

static void multiexp_serial(const Fr* coeffs, const G1Affine* bases, size_t n, G1& acc) {
  std::vector<uint64_t> repr(4 * n);
  for (size_t i = 0; i < n; i++) coeffs[i].to_raw(&repr[4 * i]);
  size_t c;
  if (n < 4)
    c = 1;
  else if (n < 32)
    c = 3;
  else
    c = (size_t)ceil(log((double)(uint32_t)n));
  size_t segments = 256 / c + 1;
  std::vector<G1> buckets((1u << c) - 1);
  std::vector<uint8_t> used((1u << c) - 1);
  for (size_t seg = segments; seg-- > 0;) {
    for (size_t k = 0; k < c; k++) acc = acc.dbl();
    std::fill(used.begin(), used.end(), 0);
    for (size_t i = 0; i < n; i++) {
      size_t d = get_at(seg, c, (const uint8_t*)&repr[4 * i]);
      if (d != 0) {
        // Bucket::{None,Affine,Projective} of the original: first point stored as-is, then mixed adds.
        if (!used[d - 1]) {
          buckets[d - 1] = G1::from_affine(bases[i]);
          used[d - 1] = 1;
        } else {
          buckets[d - 1] = buckets[d - 1].add_mixed(bases[i]);
        }
      }
    }
    G1 running = G1::identity();
    for (size_t b = buckets.size(); b-- > 0;) {
      if (used[b]) running = running.add(buckets[b]);
      acc = acc.add(running);
    }
  }
}

// arithmetic.rs best_multiexp [UP]: split into num_threads chunks of len n/num_threads (plus a
// remainder chunk), multiexp_serial each, fold the partial results in order.
// `threads` plays the role of rayon's current_num_threads(). Output: affine (x,y); the reference
// returns a projective point whose coordinates depend on the thread count, so parity is defined on
// the affine value (what `commit` callers write to the transcript after batch_normalize).
void oracle_best_multiexp(const uint64_t* scalars, const uint64_t* bases, size_t n, int threads,
                          uint64_t* out_affine) {
  const Fr* S = (const Fr*)scalars;
  const G1Affine* B = (const G1Affine*)bases;
  G1 total = G1::identity();
  if (threads < 1) threads = 1;
  if (n > (size_t)threads) {
    size_t chunk = n / threads;
    size_t num_chunks = (n + chunk - 1) / chunk;
    std::vector<G1> results(num_chunks, G1::identity());
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (size_t ci = 0; ci < num_chunks; ci++) {
      size_t lo = ci * chunk, hi = lo + chunk > n ? n : lo + chunk;
      multiexp_serial(S + lo, B + lo, hi - lo, results[ci]);
    }
    for (size_t ci = 0; ci < num_chunks; ci++) total = total.add(results[ci]);
  } else {
    multiexp_serial(S, B, n, total);
  }
  *(G1Affine*)out_affine = total.to_affine();
}

// ---------------------------------------------------------------- NTT
// Naive O(n^2) DFT: out[j] = sum_i a[i] * omega^(i*j). Cross-check for best_fft.
void oracle_dft_naive(const uint64_t* a, size_t n, const uint64_t* omega, uint64_t* out) {
  const Fr* A = (const Fr*)a;
  Fr w = *(const Fr*)omega;
  Fr* O = (Fr*)out;
  Fr wj = Fr::one();
  for (size_t j = 0; j < n; j++) {
    Fr acc = Fr::zero(), x = Fr::one();
    for (size_t i = 0; i < n; i++) {
      acc = acc + A[i] * x;
      x = x * wj;
    }
    O[j] = acc;
    wj = wj * w;
  }
}

static inline size_t bitreverse(size_t n, uint32_t l) {
  size_t r = 0;
  for (uint32_t i = 0; i < l; i++) {
    r = (r << 1) | (n & 1);
    n >>= 1;
  }
  return r;
}

// arithmetic.rs best_fft [UP]: bit-reversal swap, twiddle table of n/2 powers, log_n radix-2 DIT
// rounds (chunk doubles, twiddle stride halves). The original recurses across rayon threads for
// large n; butterflies and their order within a round are the same, and field arithmetic is exact,
// so the iterative form is value-identical. Rounds are parallelised over independent butterflies.
void oracle_best_fft(uint64_t* a, const uint64_t* omega, uint32_t log_n, int threads) {
  Fr* A = (Fr*)a;
  size_t n = (size_t)1 << log_n;
  Fr w = *(const Fr*)omega;
  if (threads < 1) threads = 1;
  for (size_t k = 0; k < n; k++) {
    size_t rk = bitreverse(k, log_n);
    if (k < rk) {
      Fr t = A[k];
      A[k] = A[rk];
      A[rk] = t;
    }
  }
  std::vector<Fr> tw(n / 2 ? n / 2 : 1);
  {
    Fr cur = Fr::one();
    for (size_t i = 0; i < n / 2; i++) {
      tw[i] = cur;
      cur = cur * w;
    }
  }
  size_t chunk = 2, twiddle_chunk = n / 2;
  for (uint32_t r = 0; r < log_n; r++) {
    size_t half = chunk / 2;
#pragma omp parallel for schedule(static) num_threads(threads) if (n >= 4096)
    for (size_t b = 0; b < n / 2; b++) {
      size_t blk = b / half, i = b % half;
      Fr* lo = A + blk * chunk + i;
      Fr* hi = lo + half;
      Fr t = *hi;
      if (i != 0) t = t * tw[i * twiddle_chunk];
      *hi = *lo - t;
      *lo = *lo + t;
    }
    chunk *= 2;
    twiddle_chunk /= 2;
  }
}

// arithmetic.rs eval_polynomial [UP]: Horner from the top coefficient.
void oracle_eval_polynomial(const uint64_t* poly, size_t n, const uint64_t* point, uint64_t* out) {
  const Fr* P = (const Fr*)poly;
  Fr x = *(const Fr*)point, acc = Fr::zero();
  for (size_t i = n; i-- > 0;) acc = acc * x + P[i];
  *(Fr*)out = acc;
}

// arithmetic.rs kate_division [UP]: q(X) = (a(X) - a(b)) / (X - b), len n-1.
void oracle_kate_division(const uint64_t* a, size_t n, const uint64_t* b, uint64_t* q) {
  const Fr* A = (const Fr*)a;
  Fr* Q = (Fr*)q;
  Fr nb = ((const Fr*)b)->neg();
  Fr tmp = Fr::zero();
  for (size_t i = n - 1; i-- > 0;) {
    Fr lead = A[i + 1] - tmp;
    Q[i] = lead;
    tmp = lead * nb;
  }
}

// Batch inversion (ff::BatchInvert semantics: zeros stay zero) [UP].
void oracle_batch_invert(uint64_t* a, size_t n) {
  Fr* A = (Fr*)a;
  std::vector<Fr> pre(n);
  Fr acc = Fr::one();
  for (size_t i = 0; i < n; i++) {
    pre[i] = acc;
    if (!A[i].is_zero()) acc = acc * A[i];
  }
  acc = acc.invert();
  for (size_t i = n; i-- > 0;) {
    if (A[i].is_zero()) continue;
    Fr t = A[i];
    A[i] = acc * pre[i];
    acc = acc * t;
  }
}

int oracle_max_threads() { return omp_get_max_threads(); }

}  // extern "C"
