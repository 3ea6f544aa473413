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
              const uint64_t omega[4], size_t ncols, uint32_t in_len, const Fr* in_coset, const Fr* out_mul, const Fr* in_first,
              const NttTables* tabs);
Fr zk_fr_inv_pow2(uint32_t log_n);

struct amdzk_domain {
  uint32_t k = 0, extended_k = 0, j = 0;
  uint64_t quotient_poly_degree = 0;
  Fr omega, omega_inv, extended_omega, extended_omega_inv;
  Fr g_coset, g_coset_inv;            // ZETA, ZETA^2
  Fr ifft_divisor, extended_ifft_divisor;
  std::vector<Fr> t_evaluations;      // already inverted, length 2^(extended_k - k)
  Fr* d_t_evaluations = nullptr;
  // Quotient plan (prover-private, zk_quotient_plan): the h(X) numerator is evaluated on nc cosets g_c * H of the
  // size-n subgroup H, g_c = zeta * extended_omega^c — the first nc of the 2^(extended_k - k) cosets that make up
  // upstream's extended domain (extended index j = c + 2^(ek-k) * i  <->  coset c, row i).
  uint32_t nc = 0;
  Fr* d_coset_in = nullptr;    // [nc][n]  32 * g_c^m                              (radix 2^261 constants)
  Fr* d_coset_out = nullptr;   // [nc][n]  g_c^-m / (32 * n * (g_c^n - 1))         (radix 2^261 constants)
  std::vector<Fr> coset_g;     // g_c
  std::vector<Fr> vinv;        // [nc][nc] inverse of V[c][j] = (g_c^n)^j, row-major [j][c]
  Fr* d_vinv261 = nullptr;     // the same as radix-2^261 constants, on the device (coset_combine's weights)
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

// tab[m] = scale * g^m as a radix-2^261 constant (packed canonical), m < count. Each thread raises g to its
// chunk start, then walks.
__global__ void coset_table_kernel(Fr* tab, Fr g, Fr scale, uint32_t count, uint32_t chunk) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t start = (uint64_t)t * chunk;
  if (start >= count) return;
  Fr cur = mul(scale, pow_u64(g, start));
  const uint32_t end = (uint32_t)((start + chunk < count) ? start + chunk : count);
  for (uint32_t i = (uint32_t)start; i < end; i++) {
    st_fr(tab + i, fr29_const_to_r261(cur));
    cur = mul(cur, g);
  }
}

// out[j][m] = sum_c w[j][c] * d[c][m], j < nout <= nc: the pieces of h(X) from its per-coset interpolants.
// w: radix-2^261 constants, row-major [j][c], in device memory. Templated on nc (1..8: the loops unroll completely, so
// the per-coset values live in registers — no dynamically indexed per-thread array next to inlined products, the shape
// DESIGN.md §6 records as miscompiled at -O3); more cosets (constraint degree above 9) take the generic kernel, which
// re-reads d[c][m] per output instead of keeping an array.
template <int NC>
__global__ __launch_bounds__(256) void coset_combine_kernel(const Fr* d, Fr* out, uint32_t n, uint32_t nout, const Fr* w) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  Fr v[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) v[c] = ld_fr(d + (size_t)c * n + m);
  for (uint32_t j = 0; j < nout; j++) {
    Fr acc = fr29_mul_const(v[0], ld_fr(w + (size_t)j * NC));
#pragma unroll
    for (int c = 1; c < NC; c++) acc = add(acc, fr29_mul_const(v[c], ld_fr(w + (size_t)j * NC + c)));
    st_fr(out + (size_t)j * n + m, acc);
  }
}
__global__ __launch_bounds__(256) void coset_combine_generic_kernel(const Fr* d, Fr* out, uint32_t n, uint32_t nc, uint32_t nout, const Fr* w) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  for (uint32_t j = 0; j < nout; j++) {
    Fr acc = fr29_mul_const(ld_fr(d + m), ld_fr(w + (size_t)j * nc));
    for (uint32_t c = 1; c < nc; c++) acc = add(acc, fr29_mul_const(ld_fr(d + (size_t)c * n + m), ld_fr(w + (size_t)j * nc + c)));
    st_fr(out + (size_t)j * n + m, acc);
  }
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
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  *out = d;
  return AMDZK_OK;
}

void amdzk_domain_free(amdzk_ctx* ctx, amdzk_domain* d) {
  ZK_ENTER(ctx);
  if (!d) return;
  if (ctx) zk_host_wait(ctx, ctx->stream);
  if (d->d_t_evaluations) hipFree(d->d_t_evaluations);
  if (d->d_coset_in) hipFree(d->d_coset_in);
  if (d->d_coset_out) hipFree(d->d_coset_out);
  if (d->d_vinv261) hipFree(d->d_vinv261);
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
  return zk_ntt_ex(ctx, (Fr*)d_cols, col_stride, (Fr*)d_cols, col_stride, d->k, (const uint64_t*)d->omega_inv.l, ncols, 0, nullptr, oc, nullptr, nullptr);
}

}  // extern "C"
// lagrange_to_coeff out of place (prover-private): the Lagrange values stay where they are
int zk_lagrange_to_coeff(amdzk_ctx* ctx, const amdzk_domain* d, const Fr* d_in, size_t in_stride, Fr* d_out, size_t out_stride, size_t ncols) {
  Fr oc[3] = {d->ifft_divisor, d->ifft_divisor, d->ifft_divisor};
  return zk_ntt_ex(ctx, d_in, in_stride, d_out, out_stride, d->k, (const uint64_t*)d->omega_inv.l, ncols, 0, nullptr, oc, nullptr, nullptr);
}
extern "C" {

int amdzk_coeff_to_lagrange_dev(amdzk_ctx* ctx, const amdzk_domain* d, void* d_cols, size_t ncols, size_t col_stride) {
  ZK_ENTER(ctx);
  if (!ctx || !d || !d_cols) return AMDZK_E_INVALID;
  return zk_ntt_ex(ctx, (Fr*)d_cols, col_stride, (Fr*)d_cols, col_stride, d->k, (const uint64_t*)d->omega.l, ncols, 0, nullptr, nullptr, nullptr, nullptr);
}

int amdzk_coeff_to_extended_dev(amdzk_ctx* ctx, const amdzk_domain* d, const void* d_coeff, size_t in_stride,
                                void* d_ext, size_t out_stride, size_t ncols) {
  ZK_ENTER(ctx);
  if (!ctx || !d || !d_coeff || !d_ext) return AMDZK_E_INVALID;
  Fr ic[2] = {d->g_coset, d->g_coset_inv};
  return zk_ntt_ex(ctx, (const Fr*)d_coeff, in_stride, (Fr*)d_ext, out_stride, d->extended_k,
                   (const uint64_t*)d->extended_omega.l, ncols, 1u << d->k, ic, nullptr, nullptr, nullptr);
}

// ---- the quotient on nc cosets instead of the whole extended domain (prover-private).
// h(X) = numerator / (X^n - 1) has degree below (j-1) n, so its j-1 pieces h_t (h = sum_t X^(t n) h_t) are pinned
// down by the numerator on ANY j-1 cosets g_c H on which X^n - 1 does not vanish — upstream evaluates on all
// 2^(ek-k) >= j-1 of them only because its domain is a power of two. On g_c H: X^n = g_c^n =: gamma_c is a
// constant, so E_c(X) := h(X) on that coset is interpolated by h_0 + gamma_c h_1 + ... (degree below n): per coset
// one size-n inverse transform, then one nc x nc inverse Vandermonde per coefficient. Same polynomial, hence the
// same pieces and commitments, for 3/4 (degree 4) of the extended transforms and of the h(X) evaluation.
// nc = 2^(ek-k) reproduces upstream's own computation exactly, also for witnesses that do not satisfy the circuit.
int zk_quotient_plan(amdzk_ctx* ctx, amdzk_domain* d, uint32_t nc) {
  const uint32_t ncosets = 1u << (d->extended_k - d->k);
  if (nc == 0 || nc > ncosets) ZK_FAIL(ctx, AMDZK_E_INVALID, "quotient plan: %u cosets requested, the extended domain has %u", nc, ncosets);
  if (nc > 64) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "quotient plan: more than 64 cosets (constraint degree above 65)");
  if (d->nc == nc) return AMDZK_OK;
  const size_t n = (size_t)1 << d->k;
  if (d->d_coset_in) ZK_HIP(ctx, hipFree(d->d_coset_in));
  if (d->d_coset_out) ZK_HIP(ctx, hipFree(d->d_coset_out));
  if (d->d_vinv261) ZK_HIP(ctx, hipFree(d->d_vinv261));
  d->d_coset_in = d->d_coset_out = d->d_vinv261 = nullptr;
  d->nc = 0;
  ZK_HIP(ctx, hipMalloc((void**)&d->d_coset_in, (size_t)nc * n * sizeof(Fr)));
  ZK_HIP(ctx, hipMalloc((void**)&d->d_coset_out, (size_t)nc * n * sizeof(Fr)));
  Fr k32 = Fr::one();
  for (int i = 0; i < 5; i++) k32 = add(k32, k32);
  const Fr base = inv(mul(k32, pow_u64(add(Fr::one(), Fr::one()), d->k)));  // 1 / (32 n)
  d->coset_g.assign(nc, Fr::zero());
  std::vector<Fr> gamma(nc);
  const uint32_t chunk = 64;
  dim3 grid((unsigned)(((n + chunk - 1) / chunk + 63) / 64)), block(64);
  for (uint32_t c = 0; c < nc; c++) {
    const Fr g = mul(d->g_coset, pow_u64(d->extended_omega, c));
    d->coset_g[c] = g;
    gamma[c] = pow_u64(g, n);
    const Fr den = sub(gamma[c], Fr::one());
    if (den.is_zero()) ZK_FAIL(ctx, AMDZK_E_INVALID, "quotient plan: X^n - 1 vanishes on coset %u", c);
    ZK_LAUNCH(ctx, "coset_table", coset_table_kernel, grid, block, 0, d->d_coset_in + (size_t)c * n, g, k32, (uint32_t)n, chunk);
    ZK_LAUNCH(ctx, "coset_table", coset_table_kernel, grid, block, 0, d->d_coset_out + (size_t)c * n, inv(g), mul(base, inv(den)), (uint32_t)n, chunk);
  }
  // vinv = V^-1 with V[c][t] = gamma_c^t (Gauss-Jordan on the host; nc is small)
  std::vector<Fr> a((size_t)nc * 2 * nc, Fr::zero());
  for (uint32_t c = 0; c < nc; c++) {
    Fr p = Fr::one();
    for (uint32_t t = 0; t < nc; t++) {
      a[(size_t)c * 2 * nc + t] = p;
      p = mul(p, gamma[c]);
    }
    a[(size_t)c * 2 * nc + nc + c] = Fr::one();
  }
  for (uint32_t col = 0; col < nc; col++) {
    uint32_t piv = col;
    while (piv < nc && a[(size_t)piv * 2 * nc + col].is_zero()) piv++;
    if (piv == nc) ZK_FAIL(ctx, AMDZK_E_INVALID, "quotient plan: singular Vandermonde");
    if (piv != col)
      for (uint32_t t = 0; t < 2 * nc; t++) std::swap(a[(size_t)piv * 2 * nc + t], a[(size_t)col * 2 * nc + t]);
    const Fr pi = inv(a[(size_t)col * 2 * nc + col]);
    for (uint32_t t = 0; t < 2 * nc; t++) a[(size_t)col * 2 * nc + t] = mul(a[(size_t)col * 2 * nc + t], pi);
    for (uint32_t r = 0; r < nc; r++) {
      if (r == col) continue;
      const Fr f = a[(size_t)r * 2 * nc + col];
      if (f.is_zero()) continue;
      for (uint32_t t = 0; t < 2 * nc; t++) a[(size_t)r * 2 * nc + t] = sub(a[(size_t)r * 2 * nc + t], mul(f, a[(size_t)col * 2 * nc + t]));
    }
  }
  d->vinv.assign((size_t)nc * nc, Fr::zero());
  for (uint32_t t = 0; t < nc; t++)
    for (uint32_t c = 0; c < nc; c++) d->vinv[(size_t)t * nc + c] = a[(size_t)t * 2 * nc + nc + c];
  {
    std::vector<Fr> w(d->vinv.size());
    for (size_t i = 0; i < w.size(); i++) w[i] = fr29_const_to_r261(d->vinv[i]);
    ZK_HIP(ctx, hipMalloc((void**)&d->d_vinv261, w.size() * sizeof(Fr)));
    ZK_HIP(ctx, hipMemcpyAsync(d->d_vinv261, w.data(), w.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));  // `w` is a host temporary
  }
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  d->nc = nc;
  return AMDZK_OK;
}
uint32_t zk_quotient_cosets(const amdzk_domain* d) { return d->nc; }
Fr zk_quotient_coset_g(const amdzk_domain* d, uint32_t c) { return d->coset_g[c]; }

// n coefficients per column -> nc x n evaluations, coset c of column y at d_out + y*out_stride + c*n, values in
// radix 2^261 (what the h(X) program multiplies in): nc size-n transforms of a_m * (32 g_c^m).
int zk_coeff_to_cosets_r261(amdzk_ctx* ctx, const amdzk_domain* d, const Fr* d_coeff, size_t in_stride, Fr* d_out, size_t out_stride,
                            size_t ncols) {
  if (!d->nc) ZK_FAIL(ctx, AMDZK_E_INVALID, "coeff_to_cosets: no quotient plan");
  const size_t n = (size_t)1 << d->k;
  if (out_stride < (size_t)d->nc * n) ZK_FAIL(ctx, AMDZK_E_INVALID, "coeff_to_cosets: output stride too small");
  NttTables t;
  t.in_tab = d->d_coset_in;
  t.tab_z_stride = n;
  t.nz = d->nc;
  t.out_z_stride = n;
  return zk_ntt_ex(ctx, d_coeff, in_stride, d_out, out_stride, d->k, (const uint64_t*)d->omega.l, ncols, 0, nullptr, nullptr, nullptr, &t);
}

// d_h: the h(X) numerator on the nc cosets ([nc][n], radix 2^261; overwritten) -> d_pieces: `npieces` <= nc pieces of
// n coefficients each in the ordinary form (the division by X^n - 1 rides in the output table).
int zk_cosets_to_pieces(amdzk_ctx* ctx, const amdzk_domain* d, Fr* d_h, Fr* d_pieces, uint32_t npieces) {
  if (!d->nc) ZK_FAIL(ctx, AMDZK_E_INVALID, "cosets_to_pieces: no quotient plan");
  if (npieces > d->nc) ZK_FAIL(ctx, AMDZK_E_INVALID, "cosets_to_pieces: %u pieces from %u cosets", npieces, d->nc);
  const size_t n = (size_t)1 << d->k;
  NttTables t;
  t.out_tab = d->d_coset_out;
  t.tab_col_stride = n;
  ZK_TRY(zk_ntt_ex(ctx, d_h, n, d_h, n, d->k, (const uint64_t*)d->omega_inv.l, d->nc, 0, nullptr, nullptr, nullptr, &t));
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define COMBINE(NC)                                                                                                                     \
  case NC:                                                                                                                              \
    ZK_LAUNCH(ctx, "coset_combine", coset_combine_kernel<NC>, grid, block, 0, (const Fr*)d_h, d_pieces, (uint32_t)n, npieces, (const Fr*)d->d_vinv261); \
    break;
  switch (d->nc) {
    COMBINE(1) COMBINE(2) COMBINE(3) COMBINE(4) COMBINE(5) COMBINE(6) COMBINE(7) COMBINE(8)
    default:
      ZK_LAUNCH(ctx, "coset_combine", coset_combine_generic_kernel, grid, block, 0, (const Fr*)d_h, d_pieces, (uint32_t)n, d->nc, npieces,
                (const Fr*)d->d_vinv261);
  }
#undef COMBINE
  return AMDZK_OK;
}

int amdzk_extended_to_coeff_dev(amdzk_ctx* ctx, const amdzk_domain* d, void* d_ext, size_t ncols, size_t col_stride) {
  ZK_ENTER(ctx);
  if (!ctx || !d || !d_ext) return AMDZK_E_INVALID;
  // ifft divisor and the inverse coset powers [1, zeta^-1 = zeta^2, zeta^-2 = zeta] in one multiplier
  Fr oc[3] = {d->extended_ifft_divisor, mul(d->extended_ifft_divisor, d->g_coset_inv), mul(d->extended_ifft_divisor, d->g_coset)};
  return zk_ntt_ex(ctx, (Fr*)d_ext, col_stride, (Fr*)d_ext, col_stride, d->extended_k,
                   (const uint64_t*)d->extended_omega_inv.l, ncols, 0, nullptr, oc, nullptr, nullptr);
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
