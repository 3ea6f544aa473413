// Microbenchmark (tools/, not part of the product): dependent chains of the shipped 8 x 32-bit Montgomery
// product (csrc/bn254.cuh) against the 9 x 29-bit in-place-accumulation product of tools/mul29.h, at 1, 2,
// 4 and 8 waves per SIMD. Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mb_mul29.hip -o tools/mb_mul29
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../anon-aadhaar-halo2_amd/csrc/bn254.cuh"
#include "mul29.h"

using namespace bn254;

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

__global__ void mul32_kernel(Fq* out, const Fq* in, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x = in[tid & 1023], y = in[(tid + 7) & 1023];
  for (int i = 0; i < iters; i++) {
    x = mul(x, y);
    y = mul(y, x);
  }
  out[tid] = add(x, y);
}

__global__ void mul29_kernel(F29* out, const Fq* in, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  F29 x = unpack29(in[tid & 1023].l), y = unpack29(in[(tid + 7) & 1023].l);
  for (int i = 0; i < iters; i++) {
    x = mul29(x, y);
    y = mul29(y, x);
  }
  F29 r;
  for (int i = 0; i < 9; i++) r.l[i] = x.l[i] ^ y.l[i];
  out[tid] = r;
}

// A mixed-addition-shaped dependency pattern (8 products + 2 squares, 3-4 independent products at a time)
// so the comparison is not only a single serial chain.
__global__ void mix32_kernel(Fq* out, const Fq* in, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = in[tid & 1023], b = in[(tid + 7) & 1023], c = in[(tid + 13) & 1023], d = in[(tid + 29) & 1023];
  for (int i = 0; i < iters; i++) {
    Fq u = mul(a, c), s = mul(b, d), p = mul(a, d), q = mul(b, c);
    Fq pp = mul(u, u), rr = mul(s, s);
    a = mul(pp, p);
    b = mul(rr, q);
    c = mul(u, q);
    d = mul(s, p);
  }
  out[tid] = add(add(a, b), add(c, d));
}
__global__ void mix29_kernel(F29* out, const Fq* in, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  F29 a = unpack29(in[tid & 1023].l), b = unpack29(in[(tid + 7) & 1023].l), c = unpack29(in[(tid + 13) & 1023].l),
      d = unpack29(in[(tid + 29) & 1023].l);
  for (int i = 0; i < iters; i++) {
    F29 u = mul29(a, c), s = mul29(b, d), p = mul29(a, d), q = mul29(b, c);
    F29 pp = mul29(u, u), rr = mul29(s, s);
    a = mul29(pp, p);
    b = mul29(rr, q);
    c = mul29(u, q);
    d = mul29(s, p);
  }
  F29 r;
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] ^ b.l[i] ^ c.l[i] ^ d.l[i];
  out[tid] = r;
}

template <class K>
float time_it(K launch, int reps) {
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
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.gcnArchName, cus, p.clockRate / 1000);
  void* buf;
  CK(hipMalloc(&buf, (size_t)cus * 8 * 256 * 64));
  Fq* in;
  CK(hipMalloc(&in, 1024 * sizeof(Fq)));
  {  // arbitrary operands below p: limbs from an LCG, top limb kept small
    Fq h[1024];
    uint32_t s = 12345;
    for (int i = 0; i < 1024; i++) {
      for (int j = 0; j < 8; j++) {
        s = s * 1664525u + 1013904223u;
        h[i].l[j] = s;
      }
      h[i].l[7] &= 0x1fffffffu;
    }
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  }
  for (int wps = 1; wps <= 8; wps *= 2) {
    const int blocks = cus * wps, iters = 400;
    const double n = (double)blocks * 256 * iters * 2;
    float a = time_it([&]() { mul32_kernel<<<blocks, 256>>>((Fq*)buf, in, iters); }, 3);
    float b = time_it([&]() { mul29_kernel<<<blocks, 256>>>((F29*)buf, in, iters); }, 3);
    printf("{\"bench\": \"dependent_mul_chain\", \"waves_per_simd\": %d, \"mul32_Gmul_per_s\": %.2f, \"mul29_Gmul_per_s\": %.2f, \"speedup\": %.3f}\n", wps,
           n / a * 1e-6, n / b * 1e-6, a / b);
  }
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = cus * wps, iters = 100;
    const double n = (double)blocks * 256 * iters * 10;
    float a = time_it([&]() { mix32_kernel<<<blocks, 256>>>((Fq*)buf, in, iters); }, 3);
    float b = time_it([&]() { mix29_kernel<<<blocks, 256>>>((F29*)buf, in, iters); }, 3);
    printf("{\"bench\": \"madd_shaped_mix\", \"waves_per_simd\": %d, \"mul32_Gmul_per_s\": %.2f, \"mul29_Gmul_per_s\": %.2f, \"speedup\": %.3f}\n", wps,
           n / a * 1e-6, n / b * 1e-6, a / b);
  }
  return 0;
}
