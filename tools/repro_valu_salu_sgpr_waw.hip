// Minimal reproducer for the one wrong-result build of round 1 (scan_local_kernel with the 29-bit product inlined,
// plonk_kernels.hip): what differed in that build's ISA (hipcc 7.2 -O3 -save-temps, good vs bad variant) is register
// allocation, not arithmetic — the dead carry-out of 1296 v_mad_u64_u32 was allocated to s[8:9], the SGPR the
// kernel's scratch (private array m[8], dynamically indexed) addresses are formed in:
//
//     v_mad_u64_u32 v[6:7], s[8:9], v43, s25, v[6:7]     ; VALU writes s[8:9] (carry-out, never read)
//     s_add_i32     s8, s35, 0x50                        ; SALU writes s8 = scratch offset of m[i]
//     ... 3 VALU ..., s_nop 0
//     scratch_load_dwordx4 v[6:9], off, s8               ; VMEM reads s8
//
// hipcc counts the 5 wait states "VALU writes SGPR -> VMEM reads that SGPR" from the v_mad and is satisfied; the
// question this program answers on the hardware is whether the LATE VALU write of s8 can land after the SALU write
// (a write-after-write on an SGPR between the vector and the scalar pipe), so that the load uses the carry mask
// instead of the offset. Each lane fills a private array with 1000 + index, then runs the sequence above with 0..6
// independent VALU instructions between the s_add and the load, and counts loads that did not return 1016.
// Control: the same sequence with the carry-out in a different SGPR pair. (The SALU write is an s_mov_b32 here: the
// kernel's s_add_i32 also sets SCC, which an asm statement inside a compiled loop must not do behind the compiler's
// back — the first two versions of this file looped forever for that reason, not because of the hazard.)
// Build + run: make -C tools repro && tools/repro_valu_salu_sgpr_waw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define SEQ(SDST, FILL)                                                                       \
  asm volatile("v_mad_u64_u32 %[acc], " SDST ", %[x], %[y], %[acc]\n\t"                       \
               "s_mov_b32 s20, %[base]\n\t" FILL "s_nop 0\n\t"                            \
               "scratch_load_dword %[out], off, s20\n\t"                                      \
               "s_waitcnt vmcnt(0)"                                                           \
               : [acc] "+v"(acc), [out] "=&v"(got)                                            \
               : [x] "v"(x), [y] "v"(y), [base] "s"(base + 64u)                               \
               : "s20", "s21", "s22", "s23", "scc", "memory")
// Second pattern, also taken from that ISA: the load is issued FIRST and the v_mad that overwrites its address SGPR
// with a carry-out comes right behind it (write-after-read), FILL instructions later.
#define SEQ_WAR(SDST, FILL)                                                                   \
  asm volatile("s_mov_b32 s20, %[base]\n\t"                                                   \
               "s_nop 4\n\t"                                                                  \
               "scratch_load_dword %[out], off, s20\n\t" FILL                                 \
               "v_mad_u64_u32 %[acc], " SDST ", %[x], %[y], %[acc]\n\t"                       \
               "s_waitcnt vmcnt(0)"                                                           \
               : [acc] "+v"(acc), [out] "=&v"(got)                                            \
               : [x] "v"(x), [y] "v"(y), [base] "s"(base + 64u)                               \
               : "s20", "s21", "s22", "s23", "scc", "memory")
#define F1 "v_xor_b32 %[x], %[x], %[y]\n\t"

__device__ __forceinline__ bool base_unused_guard(volatile uint32_t* p) { return p[63] != 1063u; }

template <int SAME, int WAR>
__global__ void probe(uint32_t* bad, uint32_t iters) {
  volatile uint32_t priv[64];
  for (int i = 0; i < 64; i++) priv[i] = 1000u + (uint32_t)i;
  if (base_unused_guard(priv)) return;
  const uint32_t base = (uint32_t)(uintptr_t)&priv[0];
  // Small factors: the 64-bit sum never overflows, so the carry-out the v_mad writes is always 0. If that late write
  // wins, the load reads scratch offset 0 — inside this lane's own scratch, harmless — instead of offset base + 64.
  // (A first version used full-range factors: its carry masks are arbitrary 32-bit offsets and the run had to be
  // killed after 60 s without output; never probe this with addresses that can leave the allocation.)
  uint64_t acc = threadIdx.x;
  uint32_t x = threadIdx.x + 1, y = 3u, got = 0;
  uint32_t wrong[7] = {0, 0, 0, 0, 0, 0, 0};
  for (uint32_t it = 0; it < iters; it++) {
#define RUN(N, FILL)                                          \
  if (WAR) { if (SAME) SEQ_WAR("s[20:21]", FILL); else SEQ_WAR("s[22:23]", FILL); } \
  else { if (SAME) SEQ("s[20:21]", FILL); else SEQ("s[22:23]", FILL); } \
  wrong[N] += got != 1016u;
    RUN(0, "")
    RUN(1, F1)
    RUN(2, F1 F1)
    RUN(3, F1 F1 F1)
    RUN(4, F1 F1 F1 F1)
    RUN(5, F1 F1 F1 F1 F1)
    RUN(6, F1 F1 F1 F1 F1 F1)
  }
  for (int n = 0; n < 7; n++) atomicAdd(&bad[n], wrong[n]);
  if (acc == 0x1234567812345678ull) bad[7] = (uint32_t)priv[3];  // keep acc and priv alive
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 8 * sizeof(uint32_t));
  for (int war = 0; war < 2; war++)
  for (int same = 1; same >= 0; same--) {
    hipMemset(d, 0, 8 * sizeof(uint32_t));
    printf("%s ", war ? "load, then v_mad writes its address SGPR (WAR):" : "v_mad, s_mov, load (WAW):                       ");
    if (war) {
      if (same) hipLaunchKernelGGL((probe<1, 1>), dim3(64), dim3(256), 0, 0, d, 100u);
      else hipLaunchKernelGGL((probe<0, 1>), dim3(64), dim3(256), 0, 0, d, 100u);
    } else if (same) hipLaunchKernelGGL((probe<1, 0>), dim3(64), dim3(256), 0, 0, d, 100u);
    else hipLaunchKernelGGL((probe<0, 0>), dim3(64), dim3(256), 0, 0, d, 100u);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    uint32_t h[8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s: wrong loads out of %u per filler count 0..6:", same ? "carry-out in the address SGPR (s[20:21])" : "control, carry-out elsewhere (s[22:23])  ", 64u * 256u * 100u);
    for (int n = 0; n < 7; n++) printf(" %u", h[n]);
    printf("\n");
  }
  return 0;
}
