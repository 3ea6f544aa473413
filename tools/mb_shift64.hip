// Microbenchmark (tools/, not part of the product): what the 64-bit right shift between two columns of the 29-bit product costs
// (acc >>= 29: one v_lshrrev_b64 per column, 152 of the 2,262 instructions of the level-1 kernel's mixed addition), against the
// same shift written as two 32-bit instructions (v_alignbit_b32 for the low word, v_lshrrev_b32 for the high word), alone and
// interleaved 1:1 with multiply-adds. cycles per wave-instruction and SIMD at 1, 2, 4 wavefronts per SIMD.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mb_shift64.hip -o tools/mb_shift64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void k(uint64_t* out, int iters) {
  uint32_t x = threadIdx.x * 2654435761u + 12345u, y = x ^ 0x9e3779b9u;
  uint64_t a0 = x, a1 = y, a2 = x + 7, a3 = y + 9;
  uint64_t b0 = ((uint64_t)x << 32) | y, b1 = b0 + 13, b2 = b0 + 17, b3 = b0 + 19;
  uint32_t c0 = x, c1 = y, c2 = x + 3, c3 = y + 5, d0 = y + 1, d1 = x + 2, d2 = y + 3, d3 = x + 4;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (MODE == 0) {  // 4 x v_lshrrev_b64 (independent)
        asm volatile("v_lshrrev_b64 %0, 29, %0\n\tv_lshrrev_b64 %1, 29, %1\n\tv_lshrrev_b64 %2, 29, %2\n\tv_lshrrev_b64 %3, 29, %3" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
      } else if (MODE == 1) {  // the same shift as alignbit (low word) + lshr (high word): 8 instructions
        asm volatile("v_alignbit_b32 %0, %4, %0, 29\n\tv_lshrrev_b32 %4, 29, %4\n\tv_alignbit_b32 %1, %5, %1, 29\n\tv_lshrrev_b32 %5, 29, %5\n\t"
                     "v_alignbit_b32 %2, %6, %2, 29\n\tv_lshrrev_b32 %6, 29, %6\n\tv_alignbit_b32 %3, %7, %3, 29\n\tv_lshrrev_b32 %7, 29, %7"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
      } else if (MODE == 2) {  // 4 mads + 4 v_lshrrev_b64, interleaved: 8 instructions
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_lshrrev_b64 %4, 29, %4\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_lshrrev_b64 %5, 29, %5\n\t"
                     "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_lshrrev_b64 %6, 29, %6\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_lshrrev_b64 %7, 29, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(x), "v"(y) : "vcc");
      } else if (MODE == 3) {  // 4 mads + 4 (alignbit + lshr): 12 instructions
        asm volatile("v_mad_u64_u32 %0, vcc, %12, %13, %0\n\tv_alignbit_b32 %4, %8, %4, 29\n\tv_lshrrev_b32 %8, 29, %8\n\t"
                     "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\tv_alignbit_b32 %5, %9, %5, 29\n\tv_lshrrev_b32 %9, 29, %9\n\t"
                     "v_mad_u64_u32 %2, vcc, %12, %13, %2\n\tv_alignbit_b32 %6, %10, %6, 29\n\tv_lshrrev_b32 %10, 29, %10\n\t"
                     "v_mad_u64_u32 %3, vcc, %12, %13, %3\n\tv_alignbit_b32 %7, %11, %7, 29\n\tv_lshrrev_b32 %11, 29, %11"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
                     : "v"(x), "v"(y) : "vcc");
      } else if (MODE == 4) {  // 4 x v_and_b32
        asm volatile("v_and_b32 %0, 0x1fffffff, %1\n\tv_and_b32 %1, 0x1fffffff, %2\n\tv_and_b32 %2, 0x1fffffff, %3\n\tv_and_b32 %3, 0x1fffffff, %0" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
      } else if (MODE == 5) {  // 4 x v_alignbit_b32
        asm volatile("v_alignbit_b32 %0, %1, %0, 29\n\tv_alignbit_b32 %1, %2, %1, 29\n\tv_alignbit_b32 %2, %3, %2, 29\n\tv_alignbit_b32 %3, %0, %3, 29" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3 ^ c0 ^ c1 ^ c2 ^ c3 ^ d0 ^ d1 ^ d2 ^ d3;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  void* buf; CK(hipMalloc(&buf, (size_t)cus * 8 * 256 * 8));
  const char* names[6] = {"v_lshrrev_b64 x4", "(v_alignbit_b32 + v_lshrrev_b32) x4 = the same shifts", "4 mad + 4 v_lshrrev_b64", "4 mad + 4 (alignbit + lshr)", "v_and_b32 x4",
                          "v_alignbit_b32 x4"};
  const int per_group[6] = {4, 8, 8, 12, 4, 4};
  for (int mode = 0; mode < 6; mode++)
    for (int wps = 1; wps <= 4; wps *= 2) {
      int blocks = cus * wps, iters = 1000;
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      auto L = [&]() {
        switch (mode) {
          case 0: k<0><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 1: k<1><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 2: k<2><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 3: k<3><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 4: k<4><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 5: k<5><<<blocks, 256>>>((uint64_t*)buf, iters); break;
        }
      };
      L(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); L(); L(); L(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
      double groups = (double)blocks * 4 * iters * 16;  // wavefront-level groups executed
      double cycles_per_group = ms * 1e-3 * (p.clockRate * 1e3) * (cus * 4.0) / groups;  // per SIMD
      printf("{\"group\": \"%s\", \"instructions_per_group\": %d, \"waves_per_simd\": %d, \"cycles_per_group_per_simd\": %.2f, \"cycles_per_instruction\": %.2f}\n", names[mode],
             per_group[mode], wps, cycles_per_group, cycles_per_group / per_group[mode]);
    }
  return 0;
}
