// ORACLE — TEST INFRASTRUCTURE ONLY ("parity unpinned" vs the reference; see bn254_ref.hpp).
//
// CPU restatement of halo2_proofs 0.2.0 @ v2023_01_20 src/poly/domain.rs EvaluationDomain [UP]
// (/root/reference/Cargo.lock:469-471): new, lagrange_to_coeff, coeff_to_lagrange,
// coeff_to_extended, extended_to_coeff, divide_by_vanishing_poly, distribute_powers_zeta.
// SURVEY.md §8(a) rows a4-a6. Every method is built from oracle_best_fft as the original is.
#include <vector>

#include "bn254_ref.hpp"

using namespace oref;

extern "C" void oracle_best_fft(uint64_t* a, const uint64_t* omega, uint32_t log_n, int threads);

struct OracleDomain {
  uint32_t k, extended_k;
  uint64_t n, quotient_poly_degree;
  Fr omega, omega_inv, extended_omega, extended_omega_inv, g_coset, g_coset_inv, ifft_divisor, extended_ifft_divisor;
  std::vector<Fr> t_evaluations;
};

static void distribute_powers_zeta(const OracleDomain* d, Fr* a, size_t len, bool into_coset) {
  Fr cp[2] = {into_coset ? d->g_coset : d->g_coset_inv, into_coset ? d->g_coset_inv : d->g_coset};
  for (size_t idx = 0; idx < len; idx++) {
    size_t i = idx % 3;
    if (i != 0) a[idx] = a[idx] * cp[i - 1];
  }
}

extern "C" {

OracleDomain* oracle_domain_new(uint32_t j, uint32_t k) {
  OracleDomain* d = new OracleDomain();
  d->k = k;
  d->quotient_poly_degree = j - 1;
  d->n = 1ull << k;
  uint32_t ek = k;
  while ((1ull << ek) < d->n * d->quotient_poly_degree) ek++;
  d->extended_k = ek;
  Fr w = fr_root_of_unity();
  for (uint32_t i = ek; i < (uint32_t)FR_S; i++) w = w.square();
  d->extended_omega = w;
  d->extended_omega_inv = w.invert();
  for (uint32_t i = k; i < ek; i++) w = w.square();
  d->omega = w;
  d->omega_inv = w.invert();
  d->g_coset = fr_zeta();
  d->g_coset_inv = d->g_coset.square();
  Fr orig = d->g_coset.pow_u64(d->n), step = d->extended_omega.pow_u64(d->n), cur = orig;
  do {
    d->t_evaluations.push_back((cur - Fr::one()).invert());
    cur = cur * step;
  } while (cur != orig);
  d->ifft_divisor = Fr::from_u64(1ull << k).invert();
  d->extended_ifft_divisor = Fr::from_u64(1ull << ek).invert();
  return d;
}
void oracle_domain_free(OracleDomain* d) { delete d; }
uint32_t oracle_domain_extended_k(const OracleDomain* d) { return d->extended_k; }
uint32_t oracle_domain_t_len(const OracleDomain* d) { return (uint32_t)d->t_evaluations.size(); }
void oracle_domain_t_evaluations(const OracleDomain* d, uint64_t* out) { memcpy(out, d->t_evaluations.data(), d->t_evaluations.size() * 32); }
void oracle_domain_constant(const OracleDomain* d, int what, uint64_t* out) {
  const Fr* src[8] = {&d->omega, &d->omega_inv, &d->extended_omega, &d->extended_omega_inv,
                      &d->g_coset, &d->g_coset_inv, &d->ifft_divisor, &d->extended_ifft_divisor};
  memcpy(out, src[what]->v, 32);
}

void oracle_lagrange_to_coeff(const OracleDomain* d, uint64_t* a, int threads) {
  oracle_best_fft(a, d->omega_inv.v, d->k, threads);
  Fr* A = (Fr*)a;
  for (uint64_t i = 0; i < d->n; i++) A[i] = A[i] * d->ifft_divisor;
}
void oracle_coeff_to_lagrange(const OracleDomain* d, uint64_t* a, int threads) {
  oracle_best_fft(a, d->omega.v, d->k, threads);
}
// in: n coefficients; out: 2^extended_k values
void oracle_coeff_to_extended(const OracleDomain* d, const uint64_t* in, uint64_t* out, int threads) {
  size_t ext = (size_t)1 << d->extended_k;
  memcpy(out, in, d->n * 32);
  memset(out + 4 * d->n, 0, (ext - d->n) * 32);
  distribute_powers_zeta(d, (Fr*)out, d->n, true);
  oracle_best_fft(out, d->extended_omega.v, d->extended_k, threads);
}
// in place on 2^extended_k values; caller truncates to n*(j-1)
void oracle_extended_to_coeff(const OracleDomain* d, uint64_t* a, int threads) {
  size_t ext = (size_t)1 << d->extended_k;
  oracle_best_fft(a, d->extended_omega_inv.v, d->extended_k, threads);
  Fr* A = (Fr*)a;
  for (size_t i = 0; i < ext; i++) A[i] = A[i] * d->extended_ifft_divisor;
  distribute_powers_zeta(d, A, ext, false);
}
void oracle_divide_by_vanishing_poly(const OracleDomain* d, uint64_t* a) {
  size_t ext = (size_t)1 << d->extended_k, m = d->t_evaluations.size();
  Fr* A = (Fr*)a;
  for (size_t i = 0; i < ext; i++) A[i] = A[i] * d->t_evaluations[i % m];
}

}  // extern "C"
