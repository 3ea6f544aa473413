// Micro-benchmarks that size the integer ALU side of the prover on gfx950: raw VALU instruction
// rates (v_mad_u64_u32, v_mul_lo/hi_u32, v_add, v_fma_f64) and Fq mul / G1 mixed-add throughput
// of bn254.cuh at several occupancies. Build: hipcc -O3 --offload-arch=gfx950 -o microbench microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../anon-aadhaar-halo2_amd/csrc/bn254.cuh"
using namespace bn254;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void raw_kernel(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = b + 77, d = a * 3 + 1;
  uint64_t acc0 = a, acc1 = b, acc2 = c, acc3 = d;
  double f0 = a, f1 = b, f2 = c, f3 = d;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (MODE == 0) {  // v_mad_u64_u32, 4 independent chains
        acc0 = (uint64_t)(uint32_t)acc0 * a + acc0; acc1 = (uint64_t)(uint32_t)acc1 * b + acc1;
        acc2 = (uint64_t)(uint32_t)acc2 * c + acc2; acc3 = (uint64_t)(uint32_t)acc3 * d + acc3;
      } else if (MODE == 1) {  // v_mul_lo_u32
        a = a * b + 1; b = b * c + 1; c = c * d + 1; d = d * a + 1;
      } else if (MODE == 2) {  // v_mul_hi_u32
        a = __umulhi(a, b) | 1; b = __umulhi(b, c) | 1; c = __umulhi(c, d) | 1; d = __umulhi(d, a) | 1;
      } else if (MODE == 3) {  // v_add_u32 / xor
        a = a + b; b = b ^ c; c = c + d; d = d ^ a;
      } else if (MODE == 4) {  // v_fma_f64
        f0 = fma(f0, 1.0000001, f1); f1 = fma(f1, 0.9999999, f2); f2 = fma(f2, 1.0000002, f3); f3 = fma(f3, 0.9999998, f0);
      } else if (MODE == 5) {  // v_mad_u32_u24
        a = __umul24(a, b) + c; b = __umul24(b, c) + d; c = __umul24(c, d) + a; d = __umul24(d, a) + b;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)(acc0 ^ acc1 ^ acc2 ^ acc3) ^ (uint32_t)(f0 + f1 + f2 + f3);
}

__global__ void fqmul_kernel(Fq* out, int iters) {
  Fq x = Fq::one(), y = Fq::r2();
  x.l[0] += threadIdx.x; y.l[1] ^= blockIdx.x;
  for (int i = 0; i < iters; i++) { x = mul(x, y); y = mul(y, x); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = add(x, y);
}
__global__ void fqsqr_add_kernel(Fq* out, int iters) {
  Fq x = Fq::one(), y = Fq::r2();
  x.l[0] += threadIdx.x; y.l[1] ^= blockIdx.x;
  for (int i = 0; i < iters; i++) { x = add(x, y); y = sub(y, x); x = add(x, x); y = sub(x, y); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = add(x, y);
}
// per-lane distinct data (a uniform input gets scalarised by the compiler and measures the SALU)
__global__ void madd_kernel(G1X* out, const G1Affine* pts, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  G1Affine g = pts[tid & 1023];
  G1X acc = x_dbl_affine(pts[(tid + 7) & 1023]);
  for (int i = 0; i < iters; i++) acc = x_add_affine(acc, g);
  out[tid] = acc;
}
// same work with the field product kept out of line (one copy of the 3 KB multiply in the I-cache)
__device__ __noinline__ Fq mul_nl(const Fq& a, const Fq& b) { return mul(a, b); }
__device__ __forceinline__ G1X x_add_affine_nl(const G1X& a, const G1Affine& q) {
  Fq u2 = mul_nl(q.x, a.zz);
  Fq s2 = mul_nl(q.y, a.zzz);
  Fq p = sub(u2, a.x);
  Fq r = sub(s2, a.y);
  Fq pp = mul_nl(p, p);
  Fq ppp = mul_nl(p, pp);
  Fq qq = mul_nl(a.x, pp);
  G1X o;
  o.x = sub(sub(mul_nl(r, r), ppp), dbl(qq));
  o.y = sub(mul_nl(r, sub(qq, o.x)), mul_nl(a.y, ppp));
  o.zz = mul_nl(a.zz, pp);
  o.zzz = mul_nl(a.zzz, ppp);
  return o;
}
__global__ void madd_nl_kernel(G1X* out, const G1Affine* pts, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  G1Affine g = pts[tid & 1023];
  G1X acc = x_dbl_affine(pts[(tid + 7) & 1023]);
  for (int i = 0; i < iters; i++) acc = x_add_affine_nl(acc, g);
  out[tid] = acc;
}
__global__ void fqmul_var_kernel(Fq* out, const G1Affine* pts, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x = pts[tid & 1023].x, y = pts[(tid + 3) & 1023].y;
  for (int i = 0; i < iters; i++) { x = mul(x, y); y = mul(y, x); }
  out[tid] = add(x, y);
}
__global__ void gen_points_kernel(G1Affine* pts) {  // pts[i] = (i+2) G, 1024 points
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  G1Affine g; g.x = Fq::one(); g.y = add(Fq::one(), Fq::one());
  G1X acc = x_dbl_affine(g);
  for (int k = 0; k < i; k++) acc = x_add_affine(acc, g);
  pts[i] = x_to_affine(acc);
}

template <class K>
float time_it(K launch, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < reps; r++) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.gcnArchName, cus, p.clockRate / 1000);
  void* buf; CK(hipMalloc(&buf, (size_t)cus * 32 * 256 * sizeof(G1X)));
  const char* names[6] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add/xor_u32", "v_fma_f64", "v_mad_u32_u24"};
  for (int mode = 0; mode < 6; mode++) {
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
      int blocks = cus * wps, iters = 2000;  // 256 threads = 4 waves = 1 per SIMD
      float ms = 0;
      auto L = [&]() {
        switch (mode) {
          case 0: raw_kernel<0><<<blocks, 256>>>((uint32_t*)buf, iters, 1); break;
          case 1: raw_kernel<1><<<blocks, 256>>>((uint32_t*)buf, iters, 1); break;
          case 2: raw_kernel<2><<<blocks, 256>>>((uint32_t*)buf, iters, 1); break;
          case 3: raw_kernel<3><<<blocks, 256>>>((uint32_t*)buf, iters, 1); break;
          case 4: raw_kernel<4><<<blocks, 256>>>((uint32_t*)buf, iters, 1); break;
          case 5: raw_kernel<5><<<blocks, 256>>>((uint32_t*)buf, iters, 1); break;
        }
      };
      ms = time_it(L, 3);
      double ops = (double)blocks * 256 * iters * 64;  // 16 unrolled x 4 chains
      double per_cu_clk = ops / (ms * 1e-3) / cus / (p.clockRate * 1e3);
      printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"Gops\": %.1f, \"lane_ops_per_cu_per_clk\": %.2f}\n",
             names[mode], wps, ms, ops / ms * 1e-6, per_cu_clk);
    }
  }
  for (int wps = 1; wps <= 8; wps *= 2) {
    int blocks = cus * wps, iters = 500;
    float ms = time_it([&]() { fqmul_kernel<<<blocks, 256>>>((Fq*)buf, iters); }, 3);
    double n = (double)blocks * 256 * iters * 2;
    printf("{\"bench\": \"fq_mul\", \"waves_per_simd\": %d, \"ms\": %.3f, \"Gmul_per_s\": %.2f, \"us_per_dependent_mul\": %.3f}\n", wps, ms, n / ms * 1e-6, ms * 1e3 / (iters * 2));
  }
  for (int wps = 1; wps <= 8; wps *= 2) {
    int blocks = cus * wps, iters = 2000;
    float ms = time_it([&]() { fqsqr_add_kernel<<<blocks, 256>>>((Fq*)buf, iters); }, 3);
    double n = (double)blocks * 256 * iters * 4;
    printf("{\"bench\": \"fq_addsub\", \"waves_per_simd\": %d, \"ms\": %.3f, \"Gop_per_s\": %.2f}\n", wps, ms, n / ms * 1e-6);
  }
  G1Affine* pts; CK(hipMalloc(&pts, 1024 * sizeof(G1Affine)));
  gen_points_kernel<<<4, 256>>>(pts); CK(hipDeviceSynchronize());
  for (int wps = 1; wps <= 8; wps *= 2) {
    int blocks = cus * wps, iters = 200;
    float ms = time_it([&]() { fqmul_var_kernel<<<blocks, 256>>>((Fq*)buf, pts, iters); }, 3);
    double n = (double)blocks * 256 * iters * 2;
    printf("{\"bench\": \"fq_mul_varying_data\", \"waves_per_simd\": %d, \"ms\": %.3f, \"Gmul_per_s\": %.2f}\n", wps, ms, n / ms * 1e-6);
  }
  for (int wps = 1; wps <= 4; wps *= 2) {
    int blocks = cus * wps, iters = 100;
    float ms = time_it([&]() { madd_kernel<<<blocks, 256>>>((G1X*)buf, pts, iters); }, 3);
    double n = (double)blocks * 256 * iters;
    printf("{\"bench\": \"g1_mixed_add_xyzz\", \"waves_per_simd\": %d, \"ms\": %.3f, \"Madd_per_s\": %.1f, \"us_per_dependent_add\": %.3f}\n", wps, ms, n / ms * 1e-3, ms * 1e3 / iters);
    ms = time_it([&]() { madd_nl_kernel<<<blocks, 256>>>((G1X*)buf, pts, iters); }, 3);
    printf("{\"bench\": \"g1_mixed_add_xyzz_noinline_mul\", \"waves_per_simd\": %d, \"ms\": %.3f, \"Madd_per_s\": %.1f, \"us_per_dependent_add\": %.3f}\n", wps, ms, n / ms * 1e-3, ms * 1e3 / iters);
  }
  return 0;
}
