// Microbenchmark (tools/, not part of the product): the compiler's 29-bit Montgomery product (fp29.cuh, C) against
// the one-asm-block product (f29_mul_asm / f29_sqr_asm), in a dependent chain and in a mixed-addition-shaped
// pattern (several independent products per step), at 1, 2, 3, 4 waves per SIMD; outputs must be identical.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mb_mulasm.hip -o tools/mb_mulasm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../anon-aadhaar-halo2_amd/csrc/fp29.cuh"

using namespace bn254;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

template <bool ASM> __device__ __forceinline__ Fq29 M(const Fq29& a, const Fq29& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (ASM) return f29_mul_asm(a, b);
#endif
  return f29_mul(a, b);
}
template <bool ASM> __device__ __forceinline__ Fq29 S(const Fq29& a) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (ASM) return f29_sqr_asm(a);
#endif
  return f29_sqr(a);
}

template <bool ASM> __global__ __launch_bounds__(256) void chain_kernel(Fq29* out, const Fq* in, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 x = fq29_unpack(in[tid & 1023]), y = fq29_unpack(in[(tid + 7) & 1023]);
  for (int i = 0; i < iters; i++) {
    x = M<ASM>(x, y);
    y = M<ASM>(y, x);
  }
  Fq29 r;
  for (int i = 0; i < 9; i++) r.l[i] = x.l[i] ^ y.l[i];
  out[tid] = r;
}
template <bool ASM> __global__ __launch_bounds__(256) void mix_kernel(Fq29* out, const Fq* in, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 a = fq29_unpack(in[tid & 1023]), b = fq29_unpack(in[(tid + 7) & 1023]), c = fq29_unpack(in[(tid + 13) & 1023]),
       d = fq29_unpack(in[(tid + 29) & 1023]);
  for (int i = 0; i < iters; i++) {
    Fq29 u = M<ASM>(a, c), s = M<ASM>(b, d), p = M<ASM>(a, d), q = M<ASM>(b, c);
    Fq29 pp = S<ASM>(u), rr = S<ASM>(s);
    a = M<ASM>(pp, p);
    b = M<ASM>(rr, q);
    c = M<ASM>(u, q);
    d = M<ASM>(s, p);
  }
  Fq29 r;
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] ^ b.l[i] ^ c.l[i] ^ d.l[i];
  out[tid] = r;
}

template <class K> float time_it(K launch, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; r++) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d}\n", p.gcnArchName, cus);
  const size_t maxthreads = (size_t)cus * 4 * 256;
  Fq29 *b0, *b1;
  CK(hipMalloc(&b0, maxthreads * sizeof(Fq29)));
  CK(hipMalloc(&b1, maxthreads * sizeof(Fq29)));
  Fq* in;
  CK(hipMalloc(&in, 1024 * sizeof(Fq)));
  {
    static Fq h[1024];
    uint32_t s = 12345;
    for (int i = 0; i < 1024; i++) {
      for (int j = 0; j < 8; j++) {
        s = s * 1664525u + 1013904223u;
        h[i].l[j] = s;
      }
      h[i].l[7] &= 0x0fffffffu;  // below p
    }
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  }
  Fq29* h0 = (Fq29*)malloc(maxthreads * sizeof(Fq29));
  Fq29* h1 = (Fq29*)malloc(maxthreads * sizeof(Fq29));
  for (int wps = 1; wps <= 4; wps++) {
    const int blocks = cus * wps;
    const size_t nthr = (size_t)blocks * 256;
    {
      const int iters = 300;
      const double n = (double)nthr * iters * 2;
      float c = time_it([&]() { chain_kernel<false><<<blocks, 256>>>(b0, in, iters); }, 3);
      float a = time_it([&]() { chain_kernel<true><<<blocks, 256>>>(b1, in, iters); }, 3);
      CK(hipMemcpy(h0, b0, nthr * sizeof(Fq29), hipMemcpyDeviceToHost));
      CK(hipMemcpy(h1, b1, nthr * sizeof(Fq29), hipMemcpyDeviceToHost));
      printf("{\"bench\": \"dependent_mul_chain\", \"waves_per_simd\": %d, \"c_Gmul_per_s\": %.2f, \"asm_Gmul_per_s\": %.2f, \"speedup\": %.3f, \"equal\": %s}\n", wps,
             n / c * 1e-6, n / a * 1e-6, c / a, memcmp(h0, h1, nthr * sizeof(Fq29)) == 0 ? "true" : "false");
    }
    {
      const int iters = 80;
      const double n = (double)nthr * iters * 10;
      float c = time_it([&]() { mix_kernel<false><<<blocks, 256>>>(b0, in, iters); }, 3);
      float a = time_it([&]() { mix_kernel<true><<<blocks, 256>>>(b1, in, iters); }, 3);
      CK(hipMemcpy(h0, b0, nthr * sizeof(Fq29), hipMemcpyDeviceToHost));
      CK(hipMemcpy(h1, b1, nthr * sizeof(Fq29), hipMemcpyDeviceToHost));
      printf("{\"bench\": \"madd_shaped_mix\", \"waves_per_simd\": %d, \"c_Gmul_per_s\": %.2f, \"asm_Gmul_per_s\": %.2f, \"speedup\": %.3f, \"equal\": %s}\n", wps,
             n / c * 1e-6, n / a * 1e-6, c / a, memcmp(h0, h1, nthr * sizeof(Fq29)) == 0 ? "true" : "false");
    }
  }
  return 0;
}
