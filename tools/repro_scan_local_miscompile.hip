// Reproducer for the one wrong-result build of round 1: scan_local_kernel (plonk_kernels.hip at 12ec628) with the
// 29-bit product fr29_mul_std inlined. The kernel text below is that revision's, with the product as a template
// switch:  V = 0 bn254.cuh's 32-bit mul (shipped then, correct)      V = 1 fr29_mul_std inlined (wrong on MI355X)
//          V = 2 fr29_mul_std behind a noinline wrapper (correct then)
//          V = 3 as 1, but the thread's eight elements are re-read from global memory instead of kept in
//                `Fr m[8]` (the array hipcc places in scratch because its loops are not fully unrolled)
// Each variant scans the same random columns; the host computes the reference with the same field code.
// Build (the failing build used the C product): hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/repro_scan_local_miscompile.hip -o tools/repro_scan_local_miscompile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../anon-aadhaar-halo2_amd/csrc/fp29.cuh"

using namespace bn254;

constexpr int SCAN_E = 8;
constexpr int SCAN_BLOCK = 256 * SCAN_E;

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
__device__ __noinline__ Fr mul_std_noinline(const Fr& a, const Fr& b) { return fr29_mul_std(a, b); }

template <int V> __device__ __forceinline__ Fr MUL(const Fr& a, const Fr& b) {
  if (V == 0) return mul(a, b);
  if (V == 2) return mul_std_noinline(a, b);
  return fr29_mul_std(a, b);
}

template <int V>
__global__ __launch_bounds__(256) void scan_local_kernel(const Fr* in, Fr* out, Fr* totals, size_t n, size_t col_stride, uint32_t nblk, Fr* dbg_tot) {
  __shared__ Fr part[256];
  const uint32_t col = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
  const Fr* src = in + (size_t)col * col_stride;
  Fr* dst = out + (size_t)col * col_stride;
  const size_t base = (size_t)blk * SCAN_BLOCK + (size_t)t * SCAN_E;
  Fr m[SCAN_E];
  Fr tot;
  if (V != 3) {
#pragma unroll
    for (int i = 0; i < SCAN_E; i++) m[i] = base + i < n ? ld_fr(src + base + i) : Fr::one();
    tot = m[0];
#pragma unroll
    for (int i = 1; i < SCAN_E; i++) tot = MUL<V>(tot, m[i]);
  } else {
    tot = base < n ? ld_fr(src + base) : Fr::one();
#pragma unroll 1
    for (int i = 1; i < SCAN_E; i++) tot = MUL<1>(tot, base + i < n ? ld_fr(src + base + i) : Fr::one());
  }
  st_fr(dbg_tot + ((size_t)col * nblk + blk) * 256 + t, tot);  // stage 1 on its own: the product of the thread's 8 elements
  part[t] = tot;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive product scan
    Fr v = t >= d ? part[t - d] : Fr::one();
    __syncthreads();
    if (t >= d) part[t] = MUL<V == 3 ? 1 : V>(v, part[t]);
    __syncthreads();
  }
  Fr pre = t == 0 ? Fr::one() : part[t - 1];
  if (V != 3) {
#pragma unroll
    for (int i = 0; i < SCAN_E; i++) {
      if (base + i < n) st_fr(dst + base + i, pre);
      pre = MUL<V>(pre, m[i]);
    }
  } else {
#pragma unroll 1
    for (int i = 0; i < SCAN_E; i++) {
      if (base + i >= n) break;
      const Fr v = ld_fr(src + base + i);
      st_fr(dst + base + i, pre);
      pre = MUL<1>(pre, v);
    }
  }
  if (t == 255) st_fr(totals + (size_t)col * nblk + blk, part[255]);
}

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

int main() {
  const size_t n = 1 << 13, ncols = 4;
  const uint32_t nblk = (uint32_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  std::vector<Fr> h(ncols * n), want(ncols * n), want_tot(ncols * nblk);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (auto& v : h) {
    for (int j = 0; j < 8; j++) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      v.l[j] = (uint32_t)s;
    }
    v.l[7] &= 0x0fffffffu;  // below r
  }
  for (size_t c = 0; c < ncols; c++)
    for (uint32_t b = 0; b < nblk; b++) {
      Fr acc = Fr::one();
      for (size_t i = (size_t)b * SCAN_BLOCK; i < n && i < (size_t)(b + 1) * SCAN_BLOCK; i++) {
        want[c * n + i] = acc;
        acc = mul(acc, h[c * n + i]);
      }
      want_tot[c * nblk + b] = acc;
    }
  Fr *d_in, *d_out, *d_tot;
  CK(hipMalloc(&d_in, h.size() * sizeof(Fr)));
  CK(hipMalloc(&d_out, h.size() * sizeof(Fr)));
  CK(hipMalloc(&d_tot, want_tot.size() * sizeof(Fr)));
  Fr* d_dbg;
  CK(hipMalloc(&d_dbg, ncols * nblk * 256 * sizeof(Fr)));
  std::vector<Fr> got_dbg(ncols * nblk * 256);
  CK(hipMemcpy(d_in, h.data(), h.size() * sizeof(Fr), hipMemcpyHostToDevice));
  std::vector<Fr> got(h.size()), got_tot(want_tot.size());
  int rc = 0;
  for (int v = 0; v < 4; v++) {
    CK(hipMemset(d_out, 0, h.size() * sizeof(Fr)));
    dim3 grid(nblk, (unsigned)ncols), block(256);
    if (v == 0) hipLaunchKernelGGL(scan_local_kernel<0>, grid, block, 0, 0, d_in, d_out, d_tot, n, n, nblk, d_dbg);
    if (v == 1) hipLaunchKernelGGL(scan_local_kernel<1>, grid, block, 0, 0, d_in, d_out, d_tot, n, n, nblk, d_dbg);
    if (v == 2) hipLaunchKernelGGL(scan_local_kernel<2>, grid, block, 0, 0, d_in, d_out, d_tot, n, n, nblk, d_dbg);
    if (v == 3) hipLaunchKernelGGL(scan_local_kernel<3>, grid, block, 0, 0, d_in, d_out, d_tot, n, n, nblk, d_dbg);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), d_out, h.size() * sizeof(Fr), hipMemcpyDeviceToHost));
    CK(hipMemcpy(got_tot.data(), d_tot, want_tot.size() * sizeof(Fr), hipMemcpyDeviceToHost));
    size_t bad = 0, first = (size_t)-1;
    for (size_t i = 0; i < h.size(); i++)
      if (memcmp(got[i].l, want[i].l, 32) != 0) {
        bad++;
        if (first == (size_t)-1) first = i;
      }
    CK(hipMemcpy(got_dbg.data(), d_dbg, got_dbg.size() * sizeof(Fr), hipMemcpyDeviceToHost));
    size_t bad_stage1 = 0;
    for (size_t c = 0; c < ncols; c++)
      for (size_t t8 = 0; t8 < n / SCAN_E; t8++) {
        Fr acc = h[c * n + t8 * SCAN_E];
        for (int i = 1; i < SCAN_E; i++) acc = mul(acc, h[c * n + t8 * SCAN_E + i]);
        bad_stage1 += memcmp(acc.l, got_dbg[c * (n / SCAN_E) + t8].l, 32) != 0;
      }
    size_t bad_tot = 0;
    for (size_t i = 0; i < want_tot.size(); i++) bad_tot += memcmp(got_tot[i].l, want_tot[i].l, 32) != 0;
    printf("variant %d: %zu of %zu running products wrong (first at element %zd: row %zd of its block, lane %zd, slot %zd), %zu of %zu block totals wrong; stage 1 (per-thread products of 8): %zu of %zu wrong\n", v, bad,
           h.size(), (ssize_t)first, first == (size_t)-1 ? -1 : (ssize_t)(first % SCAN_BLOCK), first == (size_t)-1 ? -1 : (ssize_t)((first % SCAN_BLOCK) / SCAN_E),
           first == (size_t)-1 ? -1 : (ssize_t)(first % SCAN_E), bad_tot, want_tot.size(), bad_stage1, ncols * n / SCAN_E);
    if (bad || bad_tot) rc = 1;
  }
  return rc;
}
