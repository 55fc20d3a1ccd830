// stream_probe.hip — how fast can ONE CU pull an L2-resident stream into LDS?  (build: hipcc --offload-arch=gfx950 -O3)
// Every workgroup streams the same `table` (L2 / MALL resident after the first pass) through an LDS ring, no arithmetic:
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), `depth` chunks of 16 KiB in flight, counted vmcnt
//   mode 1: global_load_dwordx4 -> registers -> ds_write_b128 (8 loads in flight per lane)
//   mode 2: global_load_dwordx4 -> registers only (accumulate, no LDS)
// Reports GB/s per CU and chip-wide for 256 / 512 / 1024-thread workgroups (one per CU: grid = 256).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NT, int MODE>
__global__ __launch_bounds__(NT) void stream_kernel(const uint4* __restrict__ table, size_t table_vec, int iters, uint4* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * 16384];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = NT / 64;
  constexpr int PIECES = 16 / NW;  // 1-KiB pieces per wave per 16-KiB chunk
  const size_t start = ((size_t)blockIdx.x * 977) % (table_vec / 1024) * 1024;  // different phase per workgroup, same table
  uint4 acc = {0, 0, 0, 0};
  if constexpr (MODE == 0) {
    auto issue = [&](int c) {
      const size_t base = (start + (size_t)c * 1024) % table_vec;
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int p = wave * PIECES + i;
        const uint4* src = table + base + p * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + (c & 3) * 16384 + p * 1024), 16, 0, 0);
      }
    };
    issue(0); issue(1); issue(2);
    for (int c = 0; c < iters; ++c) {
      if constexpr (PIECES == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (PIECES == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      issue(c + 3);
      acc.x += *reinterpret_cast<const unsigned*>(smem + (c & 3) * 16384 + tid * 4);  // touch the landed chunk
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    for (int c = 0; c < iters; c += 1) {
      const size_t base = (start + (size_t)c * 1024) % table_vec;
      uint4 v[1024 / NT];
#pragma unroll
      for (int i = 0; i < 1024 / NT; ++i) v[i] = table[base + i * NT + tid];
#pragma unroll
      for (int i = 0; i < 1024 / NT; ++i) {
        if constexpr (MODE == 1) *reinterpret_cast<uint4*>(smem + (c & 3) * 16384 + (i * NT + tid) * 16) = v[i];
        else acc.x ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
      }
      if constexpr (MODE == 1) {
        __syncthreads();
        acc.x += *reinterpret_cast<const unsigned*>(smem + (c & 3) * 16384 + tid * 4);
      }
    }
  }
  if (acc.x == 0x12345678u) out[tid] = acc;
}

template <int NT, int MODE>
void run(const uint4* table, size_t table_vec, uint4* out, const char* name) {
  const int iters = 2048;  // 32 MiB per workgroup
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {256, 512}) {
    hipLaunchKernelGGL((stream_kernel<NT, MODE>), dim3(grid), dim3(NT), 0, 0, table, table_vec, iters, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream_kernel<NT, MODE>), dim3(grid), dim3(NT), 0, 0, table, table_vec, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * iters * 16384.0;
    printf("%-34s NT=%4d grid=%3d: %7.1f us  %6.1f GB/s per CU (256 CUs)  %5.2f TB/s chip\n", name, NT, grid, ms * 1e3, bytes / ms / 1e6 / 256, bytes / ms / 1e9);
  }
}

int main(int argc, char** argv) {
  const size_t table_bytes = (argc > 1 ? atol(argv[1]) : 1) << 20;  // MiB; 1 MiB fits every L2
  const size_t table_vec = table_bytes / 16;
  uint4 *table, *out;
  CK(hipMalloc(&table, table_bytes + (64 << 20)));
  CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(table, 1, table_bytes + (64 << 20)));
  printf("table %zu MiB\n", table_bytes >> 20);
  run<256, 0>(table, table_vec, out, "LDS-DMA ring (3 chunks in flight)");
  run<512, 0>(table, table_vec, out, "LDS-DMA ring (3 chunks in flight)");
  run<1024, 0>(table, table_vec, out, "LDS-DMA ring (3 chunks in flight)");
  run<256, 1>(table, table_vec, out, "global_load -> ds_write");
  run<512, 1>(table, table_vec, out, "global_load -> ds_write");
  run<1024, 1>(table, table_vec, out, "global_load -> ds_write");
  run<256, 2>(table, table_vec, out, "global_load -> registers");
  run<512, 2>(table, table_vec, out, "global_load -> registers");
  run<1024, 2>(table, table_vec, out, "global_load -> registers");
  return 0;
}
