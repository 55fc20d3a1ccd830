// gemm_args.h — kernel-argument block and epilogue helpers shared by gemm.hip (generic register-staged kernel) and
// gemm_glds.hip (LDS-DMA pipelined kernel for the large bf16 shapes).
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Grouped mode (weight-gradient GEMMs of identical shape deferred and launched together): operand pointers of problem
// z = blockIdx.z, passed by value so that the launch needs no host-to-device copy and can be captured into a hipGraph.
constexpr int D2R_GEMM_GROUP_MAX = 16;
struct GemmGroup {
  const void* A[D2R_GEMM_GROUP_MAX];
  const void* B[D2R_GEMM_GROUP_MAX];
  void* C[D2R_GEMM_GROUP_MAX];
  float* dbias[D2R_GEMM_GROUP_MAX];
};

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  const void* R;
  void* P;
  float* ws;  // split-K partial slabs [splits][M][N] (fp32), then [splits][M] bias-gradient partials; or null
  float* dbias;  // TN only: dbias[m] += sum_k A[k,m]
  const void* G;  // optional [M,N] (layout of C): the result is multiplied by act_grad(gact, G[m,n])
  int gact;
  int grouped;  // != 0: per-problem pointers come from the GemmGroup kernel argument
  int M, N, K, nh, splits, tiles_per_split;
  int64_t lda, ldb, ldc, ldr;
  int64_t sAb, sAh, sBb, sBh, sCb, sCh, sRb, sRh, sBiasB;
  float alpha, beta;
  int act, c_dtype, vecA, vecB, vecC, xcd;
};

// Loads VEC consecutive elements [c0, c0+VEC) of a row; zero outside [0, climit) or when !row_ok.
template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_guard(const T* rowp, int c0, int climit, bool row_ok, bool vec_ok) {
  Pack<T, VEC> r;
  if (row_ok && vec_ok && c0 + VEC <= climit) {
    r = ld_pack<T, VEC>(rowp + c0);
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) r.v[j] = (row_ok && c0 + j < climit) ? rowp[c0 + j] : from_f<T>(0.f);
  }
  return r;
}

// XCD-aware tile order (cdna_hip_programming.md T1): consecutive workgroup ids are dealt round-robin to the 8 XCDs,
// each with a private L2.  Remap the linear id so that every XCD owns a CONTIGUOUS run of tiles (n fastest): the
// N-tiles that share an A row-panel then hit the same L2 instead of re-fetching it over the fabric 8 times.
// Bijective for any grid size.  Speed only; placement is not guaranteed and nothing depends on it for correctness.
__device__ __forceinline__ void xcd_tile(int enable, int& tile_m, int& tile_n) {
  const int gx = gridDim.x, nwg = gridDim.x * gridDim.y;
  int id = blockIdx.y * gx + blockIdx.x;
  if (enable && gridDim.z == 1 && nwg >= 16) {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  tile_m = id / gx;
  tile_n = id - tile_m * gx;
}

// The same remap over a 3-D grid (z = problem of a grouped launch): every XCD gets a contiguous run of (z, tile_m,
// tile_n), i.e. whole problems, so a problem's operand panels are fetched into ONE L2 instead of eight.
__device__ __forceinline__ void xcd_tile_3d(int enable, int& tile_m, int& tile_n, int& z) {
  const int gx = gridDim.x, gxy = gridDim.x * gridDim.y, nwg = gxy * gridDim.z;
  int id = (blockIdx.z * gridDim.y + blockIdx.y) * gx + blockIdx.x;
  if (enable && nwg >= 16) {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  z = id / gxy;
  const int rem = id - z * gxy;
  tile_m = rem / gx;
  tile_n = rem - tile_m * gx;
}

__device__ __forceinline__ void store_c(void* C, int c_dtype, int64_t idx, float v) {
  if (c_dtype == D2R_BF16) reinterpret_cast<bf16_t*>(C)[idx] = (bf16_t)v;
  else reinterpret_cast<float*>(C)[idx] = v;
}
__device__ __forceinline__ float load_c(const void* C, int c_dtype, int64_t idx) {
  return c_dtype == D2R_BF16 ? (float)reinterpret_cast<const bf16_t*>(C)[idx] : reinterpret_cast<const float*>(C)[idx];
}

