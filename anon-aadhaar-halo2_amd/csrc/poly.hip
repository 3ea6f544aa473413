// EvaluationDomain on the device + pointwise polynomial kernels.
//
// Replaces halo2_proofs::poly::domain::EvaluationDomain::{new, lagrange_to_coeff, coeff_to_lagrange,
// coeff_to_extended, extended_to_coeff, divide_by_vanishing_poly} (halo2_proofs 0.2.0 @ v2023_01_20
// [UP], /root/reference/Cargo.lock:469-471). The zeta-coset scaling
// (distribute_powers_zeta), the zero padding to the extended domain, the 1/n divisor and the
// coset un-scaling are all folded into the first / last NTT step instead of being separate passes
// over memory.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "fp29.cuh"

using namespace bn254;

int zk_ntt_ex(amdzk_ctx* ctx, const Fr* d_in, size_t in_stride, Fr* d_out, size_t out_stride, uint32_t log_n,
              const uint64_t omega[4], size_t ncols, uint32_t in_len, const Fr* in_coset, const Fr* out_mul, const Fr* in_first);
Fr zk_fr_inv_pow2(uint32_t log_n);

struct amdzk_domain {
  uint32_t k = 0, extended_k = 0, j = 0;
  uint64_t quotient_poly_degree = 0;
  Fr omega, omega_inv, extended_omega, extended_omega_inv;
  Fr g_coset, g_coset_inv;            // ZETA, ZETA^2
  Fr ifft_divisor, extended_ifft_divisor;
  std::vector<Fr> t_evaluations;      // already inverted, length 2^(extended_k - k)
  Fr* d_t_evaluations = nullptr;
};

namespace {

Fr fr_from_canonical(uint64_t a3, uint64_t a2, uint64_t a1, uint64_t a0) {
  Fr r;
  uint64_t v[4] = {a0, a1, a2, a3};
  memcpy(r.l, v, 32);
  return to_mont(r);
}
// halo2curves bn256 Fr::ROOT_OF_UNITY = 7^((r-1)/2^28) and Fr::ZETA [UP]; S = 28.
Fr fr_root_of_unity() {
  return fr_from_canonical(0x03ddb9f5166d18b7ULL, 0x98865ea93dd31f74ULL, 0x3215cf6dd39329c8ULL, 0xd34f1ed960c37c9cULL);
}
Fr fr_zeta() {
  return fr_from_canonical(0x30644e72e131a029ULL, 0x048b6e193fd84104ULL, 0xcc37a73fec2bc5e9ULL, 0xb8ca0b2d36636f23ULL);
}
constexpr uint32_t FR_S = 28;

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

// a[i] *= table[i & mask]   (divide_by_vanishing_poly: table = inverted t_evaluations)
__global__ __launch_bounds__(256) void mul_periodic_kernel(Fr* a, size_t col_stride, size_t n, const Fr* table,
                                                           uint32_t mask) {
  Fr* col = a + (size_t)blockIdx.y * col_stride;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    st_fr(col + i, fr29_mul_std(ld_fr(col + i), ld_fr(table + (i & mask))));
}

// a[i] = a[i] * c  (c = R^2: canonical -> Montgomery, Fr::from_raw; c = 1: Montgomery -> canonical, to_repr)
__global__ __launch_bounds__(256) void mul_const_kernel(Fr* a, size_t n, Fr c) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    st_fr(a + i, fr29_mul_std(ld_fr(a + i), c));
}

}  // namespace

extern "C" {

int amdzk_fr_from_raw_dev(amdzk_ctx* ctx, void* d_a, size_t n) {
  ZK_ENTER(ctx);
  if (!ctx || (!d_a && n)) return AMDZK_E_INVALID;
  unsigned gx = (unsigned)((n + 255) / 256);
  if (gx > 4096) gx = 4096;
  if (n) ZK_LAUNCH(ctx, "fr_mul_const", mul_const_kernel, dim3(gx), dim3(256), 0, (Fr*)d_a, n, Fr::r2());
  return AMDZK_OK;
}
int amdzk_fr_to_repr_dev(amdzk_ctx* ctx, void* d_a, size_t n) {
  ZK_ENTER(ctx);
  if (!ctx || (!d_a && n)) return AMDZK_E_INVALID;
  unsigned gx = (unsigned)((n + 255) / 256);
  if (gx > 4096) gx = 4096;
  Fr o = Fr::zero();
  o.l[0] = 1;
  if (n) ZK_LAUNCH(ctx, "fr_mul_const", mul_const_kernel, dim3(gx), dim3(256), 0, (Fr*)d_a, n, o);
  return AMDZK_OK;
}

int amdzk_domain_new(amdzk_ctx* ctx, uint32_t j, uint32_t k, amdzk_domain** out) {
  ZK_ENTER(ctx);
  if (!ctx || !out) return AMDZK_E_INVALID;
  if (j < 2) ZK_FAIL(ctx, AMDZK_E_INVALID, "domain_new: degree j = %u < 2", j);
  amdzk_domain* d = new amdzk_domain();
  d->k = k;
  d->j = j;
  d->quotient_poly_degree = j - 1;
  const uint64_t n = 1ull << k;
  uint32_t ek = k;
  while ((1ull << ek) < n * d->quotient_poly_degree) ek++;
  if (ek > FR_S || ek > 27) {
    delete d;
    ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "domain_new: extended_k %u too large", ek);
  }
  d->extended_k = ek;
  Fr w = fr_root_of_unity();
  for (uint32_t i = ek; i < FR_S; i++) w = sqr(w);
  d->extended_omega = w;
  d->extended_omega_inv = inv(w);
  for (uint32_t i = k; i < ek; i++) w = sqr(w);
  d->omega = w;
  d->omega_inv = inv(w);
  d->g_coset = fr_zeta();
  d->g_coset_inv = sqr(d->g_coset);
  d->ifft_divisor = zk_fr_inv_pow2(k);
  d->extended_ifft_divisor = zk_fr_inv_pow2(ek);
  // t(X) = X^n - 1 on the coset zeta * <extended_omega>: period 2^(ek-k); stored inverted.
  Fr orig = pow_u64(d->g_coset, n), step = pow_u64(d->extended_omega, n), cur = orig;
  do {
    d->t_evaluations.push_back(inv(sub(cur, Fr::one())));
    cur = mul(cur, step);
  } while (cur != orig);
  if (d->t_evaluations.size() != (size_t)1 << (ek - k)) {
    delete d;
    ZK_FAIL(ctx, AMDZK_E_INVALID, "domain_new: t_evaluations period mismatch");
  }
  size_t bytes = d->t_evaluations.size() * sizeof(Fr);
  hipError_t e = hipMalloc((void**)&d->d_t_evaluations, bytes);
  if (e != hipSuccess) {
    delete d;
    ZK_FAIL(ctx, AMDZK_E_NOMEM, "domain_new: hipMalloc failed");
  }
  ZK_HIP(ctx, hipMemcpyAsync(d->d_t_evaluations, d->t_evaluations.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *out = d;
  return AMDZK_OK;
}

void amdzk_domain_free(amdzk_ctx* ctx, amdzk_domain* d) {
  ZK_ENTER(ctx);
  if (!d) return;
  if (ctx) hipStreamSynchronize(ctx->stream);
  if (d->d_t_evaluations) hipFree(d->d_t_evaluations);
  delete d;
}

// what: 0 omega, 1 omega_inv, 2 extended_omega, 3 extended_omega_inv, 4 g_coset, 5 g_coset_inv,
//       6 ifft_divisor, 7 extended_ifft_divisor
int amdzk_domain_constant(const amdzk_domain* d, int what, uint64_t out[4]) {
  if (!d || !out) return AMDZK_E_INVALID;
  const Fr* src[8] = {&d->omega, &d->omega_inv, &d->extended_omega, &d->extended_omega_inv,
                      &d->g_coset, &d->g_coset_inv, &d->ifft_divisor, &d->extended_ifft_divisor};
  if (what < 0 || what > 7) return AMDZK_E_INVALID;
  memcpy(out, src[what]->l, 32);
  return AMDZK_OK;
}
uint32_t amdzk_domain_k(const amdzk_domain* d) { return d ? d->k : 0; }
uint32_t amdzk_domain_extended_k(const amdzk_domain* d) { return d ? d->extended_k : 0; }

int amdzk_lagrange_to_coeff_dev(amdzk_ctx* ctx, const amdzk_domain* d, void* d_cols, size_t ncols, size_t col_stride) {
  ZK_ENTER(ctx);
  if (!ctx || !d || !d_cols) return AMDZK_E_INVALID;
  Fr oc[3] = {d->ifft_divisor, d->ifft_divisor, d->ifft_divisor};
  return zk_ntt_ex(ctx, (Fr*)d_cols, col_stride, (Fr*)d_cols, col_stride, d->k, (const uint64_t*)d->omega_inv.l, ncols, 0, nullptr, oc, nullptr);
}

int amdzk_coeff_to_lagrange_dev(amdzk_ctx* ctx, const amdzk_domain* d, void* d_cols, size_t ncols, size_t col_stride) {
  ZK_ENTER(ctx);
  if (!ctx || !d || !d_cols) return AMDZK_E_INVALID;
  return zk_ntt_ex(ctx, (Fr*)d_cols, col_stride, (Fr*)d_cols, col_stride, d->k, (const uint64_t*)d->omega.l, ncols, 0, nullptr, nullptr, nullptr);
}

int amdzk_coeff_to_extended_dev(amdzk_ctx* ctx, const amdzk_domain* d, const void* d_coeff, size_t in_stride,
                                void* d_ext, size_t out_stride, size_t ncols) {
  ZK_ENTER(ctx);
  if (!ctx || !d || !d_coeff || !d_ext) return AMDZK_E_INVALID;
  Fr ic[2] = {d->g_coset, d->g_coset_inv};
  return zk_ntt_ex(ctx, (const Fr*)d_coeff, in_stride, (Fr*)d_ext, out_stride, d->extended_k,
                   (const uint64_t*)d->extended_omega.l, ncols, 1u << d->k, ic, nullptr, nullptr);
}

// The prover's private flavour of the two conversions. The h(X) interpreter multiplies data by data with
// fp29.cuh's in-place product, which is closed only on radix-2^261 Montgomery values; x*2^261 is the same
// 32 bytes as (32 x)*2^256, so "extended-domain data in radix 2^261" is simply 32 times the polynomial in
// the ordinary form — a factor the (linear) transform picks up from its input constants for free, and
// drops again through its output constants on the way back. The public amdzk_* entry points above are
// unchanged.
static Fr fr_k32() {
  Fr k = Fr::one();
  for (int i = 0; i < 5; i++) k = add(k, k);
  return k;
}
int zk_coeff_to_extended_r261(amdzk_ctx* ctx, const amdzk_domain* d, const Fr* d_coeff, size_t in_stride, Fr* d_ext, size_t out_stride,
                              size_t ncols) {
  const Fr k32 = fr_k32();
  Fr ic[2] = {mul(d->g_coset, k32), mul(d->g_coset_inv, k32)};
  return zk_ntt_ex(ctx, d_coeff, in_stride, d_ext, out_stride, d->extended_k, (const uint64_t*)d->extended_omega.l, ncols, 1u << d->k, ic,
                   nullptr, &k32);
}
int zk_extended_to_coeff_from_r261(amdzk_ctx* ctx, const amdzk_domain* d, Fr* d_ext, size_t ncols, size_t col_stride) {
  const Fr div = mul(d->extended_ifft_divisor, inv(fr_k32()));
  Fr oc[3] = {div, mul(div, d->g_coset_inv), mul(div, d->g_coset)};
  return zk_ntt_ex(ctx, d_ext, col_stride, d_ext, col_stride, d->extended_k, (const uint64_t*)d->extended_omega_inv.l, ncols, 0, nullptr, oc,
                   nullptr);
}

int amdzk_extended_to_coeff_dev(amdzk_ctx* ctx, const amdzk_domain* d, void* d_ext, size_t ncols, size_t col_stride) {
  ZK_ENTER(ctx);
  if (!ctx || !d || !d_ext) return AMDZK_E_INVALID;
  // ifft divisor and the inverse coset powers [1, zeta^-1 = zeta^2, zeta^-2 = zeta] in one multiplier
  Fr oc[3] = {d->extended_ifft_divisor, mul(d->extended_ifft_divisor, d->g_coset_inv), mul(d->extended_ifft_divisor, d->g_coset)};
  return zk_ntt_ex(ctx, (Fr*)d_ext, col_stride, (Fr*)d_ext, col_stride, d->extended_k,
                   (const uint64_t*)d->extended_omega_inv.l, ncols, 0, nullptr, oc, nullptr);
}

int amdzk_divide_by_vanishing_dev(amdzk_ctx* ctx, const amdzk_domain* d, void* d_ext, size_t ncols, size_t col_stride) {
  ZK_ENTER(ctx);
  if (!ctx || !d || !d_ext) return AMDZK_E_INVALID;
  const size_t n = (size_t)1 << d->extended_k;
  unsigned gx = (unsigned)((n + 255) / 256);
  if (gx > 2048) gx = 2048;
  ZK_LAUNCH(ctx, "mul_periodic", mul_periodic_kernel, dim3(gx, (unsigned)ncols), dim3(256), 0, (Fr*)d_ext, col_stride, n,
            d->d_t_evaluations, (uint32_t)(d->t_evaluations.size() - 1));
  return AMDZK_OK;
}

}  // extern "C"
