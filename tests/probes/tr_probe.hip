// Probe: pins the lane <-> element map of ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4bf16) on gfx950.
// Expected (cdna_hip_programming.md T10): within each 16-lane group, lane 4q+p supplies the address of block row q,
// columns 4p..4p+3; lane i receives column i of the 4 rows, row q in element q.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__global__ void k(const __bf16* in, float* out) {
  __shared__ __attribute__((aligned(16))) __bf16 T[16 * 16];
  for (int i = threadIdx.x; i < 256; i += 64) T[i] = in[i];
  __syncthreads();
  int l = threadIdx.x;
  int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  const __bf16* addr = &T[(4 * g + q) * 16 + 4 * p];
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)addr);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}
int main() {
  __bf16 h[256];
  for (int i = 0; i < 256; ++i) h[i] = (__bf16)(float)i;
  __bf16* d; float* o; float ho[256];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
    int g = l >> 4, i = l & 15;
    float expect = (4 * g + j) * 16 + i;
    if (ho[l * 4 + j] != expect) { if (bad < 8) printf("lane %d elem %d: got %g expected %g\n", l, j, ho[l*4+j], expect); ++bad; }
  }
  printf("lane0: %g %g %g %g | lane1: %g %g %g %g | lane17: %g %g %g %g\n", ho[0],ho[1],ho[2],ho[3],ho[4],ho[5],ho[6],ho[7],ho[68],ho[69],ho[70],ho[71]);
  printf(bad ? "TR PROBE MISMATCH (%d)\n" : "TR PROBE OK\n", bad);
  return bad != 0;
}
