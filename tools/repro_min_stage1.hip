// Minimised form of tools/repro_scan_local_miscompile.hip: only the first stage of round 1's scan_local_kernel —
// eight field elements into a per-thread array, then their product with fr29_mul_std inlined — in three spellings:
//   A  the original: `#pragma unroll` on both loops (hipcc leaves the product loop partially unrolled, so m[] is
//      indexed dynamically and lives in scratch)
//   B  `#pragma unroll 1` on the product loop (rolled, m[] still in scratch)
//   C  no array: each element is loaded right before it is multiplied in
// and compares each with the host's result. Build with -O3 / -O2 / -O1 to see which pipeline is affected:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/repro_min_stage1.hip -o tools/repro_min_stage1_O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../anon-aadhaar-halo2_amd/csrc/fp29.cuh"

using namespace bn254;
constexpr int E = 8;

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

template <int V> __global__ __launch_bounds__(256) void stage1(const Fr* in, Fr* out, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, base = t * E;
  Fr tot;
  if (V == 2) {
    tot = base < n ? ld_fr(in + base) : Fr::one();
#pragma unroll 1
    for (int i = 1; i < E; i++) tot = fr29_mul_std(tot, base + i < n ? ld_fr(in + base + i) : Fr::one());
  } else {
    Fr m[E];
#pragma unroll
    for (int i = 0; i < E; i++) m[i] = base + i < n ? ld_fr(in + base + i) : Fr::one();
    tot = m[0];
    if (V == 0) {
#pragma unroll
      for (int i = 1; i < E; i++) tot = fr29_mul_std(tot, m[i]);
    } else {
#pragma unroll 1
      for (int i = 1; i < E; i++) tot = fr29_mul_std(tot, m[i]);
    }
  }
  st_fr(out + t, tot);
}

int main() {
  const size_t n = 1 << 13, nt = n / E;
  std::vector<Fr> h(n), want(nt), got(nt);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (auto& v : h) {
    for (int j = 0; j < 8; j++) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      v.l[j] = (uint32_t)s;
    }
    v.l[7] &= 0x0fffffffu;
  }
  for (size_t t = 0; t < nt; t++) {
    Fr acc = h[t * E];
    for (int i = 1; i < E; i++) acc = mul(acc, h[t * E + i]);
    want[t] = acc;
  }
  Fr *d_in, *d_out;
  hipMalloc(&d_in, n * sizeof(Fr));
  hipMalloc(&d_out, nt * sizeof(Fr));
  hipMemcpy(d_in, h.data(), n * sizeof(Fr), hipMemcpyHostToDevice);
  int rc = 0;
  for (int v = 0; v < 3; v++) {
    hipMemset(d_out, 0, nt * sizeof(Fr));
    if (v == 0) hipLaunchKernelGGL(stage1<0>, dim3(nt / 256), dim3(256), 0, 0, d_in, d_out, n);
    if (v == 1) hipLaunchKernelGGL(stage1<1>, dim3(nt / 256), dim3(256), 0, 0, d_in, d_out, n);
    if (v == 2) hipLaunchKernelGGL(stage1<2>, dim3(nt / 256), dim3(256), 0, 0, d_in, d_out, n);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipMemcpy(got.data(), d_out, nt * sizeof(Fr), hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t t = 0; t < nt; t++) bad += memcmp(got[t].l, want[t].l, 32) != 0;
    printf("spelling %c: %zu of %zu products wrong\n", 'A' + v, bad, nt);
    rc |= bad != 0;
  }
  return rc;
}
