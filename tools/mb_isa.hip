// Instruction-rate microbenchmark with inline asm (the compiler cannot fold these): lane-ops per CU per clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void k(uint64_t* out, int iters) {
  uint32_t x = threadIdx.x * 2654435761u + 12345u, y = x ^ 0x9e3779b9u;
  uint64_t a0 = x, a1 = y, a2 = x + 7, a3 = y + 9;
  uint32_t c0 = x, c1 = y, c2 = x + 3, c3 = y + 5;
  uint64_t b0 = x + 11, b1 = y + 13, b2 = x + 17, b3 = y + 19;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (MODE == 0) {
        asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");
      } else if (MODE == 1) {
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %1, %1, 0, %2\n\tv_lshl_add_u64 %2, %2, 0, %3\n\tv_lshl_add_u64 %3, %3, 0, %0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      } else if (MODE == 2) {
        asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
      } else if (MODE == 3) {
        asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc\n\tv_add_co_u32 %2, vcc, %2, %3\n\tv_addc_co_u32 %3, vcc, %3, %0, vcc"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "vcc");
      } else if (MODE == 4) {
        asm volatile("v_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %1, %1, %2\n\tv_mul_lo_u32 %2, %2, %3\n\tv_mul_lo_u32 %3, %3, %0" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
      } else if (MODE == 9) {  // 8 independent chains, every carry-out in its own SGPR pair (as compiled code has them; vcc for all serialises)
        asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n\tv_mad_u64_u32 %1, s[22:23], %8, %9, %1\n\tv_mad_u64_u32 %2, s[24:25], %8, %9, %2\n\t"
                     "v_mad_u64_u32 %3, s[26:27], %8, %9, %3\n\tv_mad_u64_u32 %4, s[28:29], %8, %9, %4\n\tv_mad_u64_u32 %5, s[30:31], %8, %9, %5\n\t"
                     "v_mad_u64_u32 %6, s[32:33], %8, %9, %6\n\tv_mad_u64_u32 %7, s[34:35], %8, %9, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(x), "v"(y)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
      } else if (MODE == 10) {
        asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n\tv_mad_u64_u32 %1, s[20:21], %8, %9, %1\n\tv_mad_u64_u32 %2, s[20:21], %8, %9, %2\n\tv_mad_u64_u32 %3, s[20:21], %8, %9, %3\n\tv_mad_u64_u32 %4, s[20:21], %8, %9, %4\n\tv_mad_u64_u32 %5, s[20:21], %8, %9, %5\n\tv_mad_u64_u32 %6, s[20:21], %8, %9, %6\n\tv_mad_u64_u32 %7, s[20:21], %8, %9, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(x), "v"(y)
                     : "s20", "s21");
      } else if (MODE == 11) {
        asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n\tv_mad_u64_u32 %1, s[22:23], %8, %9, %1\n\tv_mad_u64_u32 %2, s[20:21], %8, %9, %2\n\tv_mad_u64_u32 %3, s[22:23], %8, %9, %3\n\tv_mad_u64_u32 %4, s[20:21], %8, %9, %4\n\tv_mad_u64_u32 %5, s[22:23], %8, %9, %5\n\tv_mad_u64_u32 %6, s[20:21], %8, %9, %6\n\tv_mad_u64_u32 %7, s[22:23], %8, %9, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(x), "v"(y)
                     : "s20", "s21", "s22", "s23");
      } else if (MODE == 12) {
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(x), "v"(y)
                     : "vcc");
      } else if (MODE == 13) {
        asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n\tv_mad_u64_u32 %1, s[22:23], %8, %9, %1\n\tv_mad_u64_u32 %2, s[24:25], %8, %9, %2\n\tv_mad_u64_u32 %3, s[26:27], %8, %9, %3\n\tv_mad_u64_u32 %4, s[20:21], %8, %9, %4\n\tv_mad_u64_u32 %5, s[22:23], %8, %9, %5\n\tv_mad_u64_u32 %6, s[24:25], %8, %9, %6\n\tv_mad_u64_u32 %7, s[26:27], %8, %9, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(x), "v"(y)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
      } else if (MODE == 6) {  // 4 mads + 4 moves, interleaved: do the cheap instructions issue in the multiplier's shadow?
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mov_b32 %4, %5\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mov_b32 %5, %6\n\t"
                     "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mov_b32 %6, %7\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mov_b32 %7, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y) : "vcc");
      } else if (MODE == 7) {  // 4 mads + 4 and/shift (the limb-splitting instructions around a product)
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_and_b32 %4, 0x1fffffff, %5\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_lshrrev_b32 %5, 29, %6\n\t"
                     "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_and_b32 %6, 0x1fffffff, %7\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_lshrrev_b32 %7, 29, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y) : "vcc");
      } else if (MODE == 8) {  // 4 mads + 4 LDS-free 64-bit adds
        asm volatile("v_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_lshl_add_u64 %4, %4, 0, %5\n\tv_mad_u64_u32 %1, vcc, %6, %7, %1\n\tv_lshl_add_u64 %5, %5, 0, %4\n\t"
                     "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\tv_lshl_add_u64 %4, %4, 0, %5\n\tv_mad_u64_u32 %3, vcc, %6, %7, %3\n\tv_lshl_add_u64 %5, %5, 0, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1) : "v"(x), "v"(y) : "vcc");
      } else if (MODE == 5) {  // mad with carry-out consumed by addc (the column-accumulate pattern)
        asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1) : "v"(x), "v"(y) : "vcc");
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ c0 ^ c1 ^ c2 ^ c3 ^ b0 ^ b1 ^ b2 ^ b3;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  void* buf; CK(hipMalloc(&buf, (size_t)cus * 8 * 256 * 8));
  const char* names[14] = {"v_mad_u64_u32", "v_lshl_add_u64", "v_mov_b32", "v_add_co/addc_u32", "v_mul_lo_u32", "mad+addc pair (2 instr)",
                          "4 mad + 4 v_mov (8 instr)", "4 mad + 4 and/shift (8 instr)", "4 mad + 4 v_lshl_add_u64 (8 instr)",
                          "v_mad_u64_u32, 8 chains, carry-outs in 8 SGPR pairs (8 instr)",
                          "8 chains, every carry-out in s[20:21] (8 instr)", "8 chains, carry-outs alternate 2 SGPR pairs (8 instr)",
                          "8 chains, every carry-out in vcc (8 instr)", "8 chains, carry-outs rotate over 4 SGPR pairs (8 instr)"};
  for (int mode = 0; mode < 14; mode++)
    for (int wps = 1; wps <= 8; wps *= 2) {
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
          case 6: k<6><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 7: k<7><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 8: k<8><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 9: k<9><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 10: k<10><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 11: k<11><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 12: k<12><<<blocks, 256>>>((uint64_t*)buf, iters); break;
          case 13: k<13><<<blocks, 256>>>((uint64_t*)buf, iters); break;
        }
      };
      L(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); L(); L(); L(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
      double instrs = (double)blocks * 256 * iters * (mode >= 6 ? 128 : 64);  // 16 x 4 (or 8) instructions per lane
      double per_cu_clk = instrs / (ms * 1e-3) / cus / (p.clockRate * 1e3);
      printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"lane_instr_per_cu_per_clk\": %.1f, \"cycles_per_wave_instr_per_simd\": %.2f}\n",
             names[mode], wps, ms, per_cu_clk, 256.0 / per_cu_clk);
    }
  return 0;
}
