// Device-side check (debugging aid): bn254.cuh's mul against fp29.cuh's ordinary-radix drop-in on random
// canonical operands over the full range, evaluated ON THE GPU. Prints the number of mismatches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../anon-aadhaar-halo2_amd/csrc/fp29.cuh"
using namespace bn254;

__device__ bool lt_p(const Fr& a) {
  for (int i = 7; i >= 0; i--)
    if (a.l[i] != FrP::p(i)) return a.l[i] < FrP::p(i);
  return false;
}
__global__ void k(unsigned* bad, unsigned* firstbad, int iters) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long s = 0x9E3779B97F4A7C15ULL * (tid + 1);
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int it = 0; it < iters; it++) {
    Fr x, c;
    do { for (int i = 0; i < 8; i++) x.l[i] = (uint32_t)rnd(); x.l[7] &= 0x3fffffffu; } while (!lt_p(x));
    do { for (int i = 0; i < 8; i++) c.l[i] = (uint32_t)rnd(); c.l[7] &= 0x3fffffffu; } while (!lt_p(c));
    if (it % 5 == 0) for (int i = 0; i < 8; i++) c.l[i] = FrP::p(i) - (i == 0 ? 1 + (uint32_t)(rnd() % 1000) : 0);
    if (it % 7 == 0) x = Fr::one();
    Fr w = mul(x, c), g = fr29_mul_std(x, c);
    bool ok = true;
    for (int i = 0; i < 8; i++) ok = ok && w.l[i] == g.l[i];
    Fr g2 = fr29_mul_const(x, fr29_const_to_r261(c));
    for (int i = 0; i < 8; i++) ok = ok && w.l[i] == g2.l[i];
    if (!ok) { atomicAdd(bad, 1u); atomicMin(firstbad, tid * 1000u + (unsigned)it); }
  }
}
int main() {
  unsigned *d, h[2] = {0, 0xffffffffu};
  if (hipMalloc(&d, 8) != hipSuccess) return 2;
  hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
  k<<<256, 256>>>(d, d + 1, 50);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 3; }
  hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("device mul vs fr29_mul_std / fr29_mul_const: mismatches %u of %u (first %u)\n", h[0], 256u * 256u * 50u, h[1]);
  return h[0] != 0;
}
