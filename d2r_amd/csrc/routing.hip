// routing.hip — K1 router pooling and K8 route_aggregate (path normalisation + threshold gate + aggregation).
//
// HBM-bound streaming kernels: every activation element is read once as a 16-byte pack, all per-sample
// routing coefficients live in LDS/registers, reductions are fixed-order (wave shuffles -> LDS -> stage-2
// kernel), no float atomics.
//
// Reference: models/Router.py:23 (mean over tokens); models/DynamicInteraction.py:50-67 (=:119-132,
// :170-187, :239-252) for P=6 and :104-117 (=:224-237) for the final layer P=1; RIC relu models/Cells.py:38.
#include "common.h"

struct Ptrs8 {
  const void* p[8];
};
struct MPtrs8 {
  void* p[8];
};

#define TH_GATE 1e-4f          /* self.threshold (models/DynamicInteraction.py:24) */
// final layer: self.threshold / self.num_cell (models/DynamicInteraction.py:109), evaluated in double like Python, compared in fp32
__device__ __forceinline__ float th_gate_final(int nc) { return (float)(1e-4 / (double)nc); }
#define EPS_NORM 1e-8f

// =====================================================================================================
// K1: mean over tokens.  grid = (D / (16*VEC), B, nsrc); 256 threads = 16 column packs x 16 row groups
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void meanpool_fwd_kernel(Ptrs8 srcs, int B, int L, int D, float* __restrict__ pooled) {
  constexpr int VEC = PackOf<T>::N;
  __shared__ float sh[16][16 * VEC + 1];
  const int cp = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int b = blockIdx.y, s = blockIdx.z;
  const int col0 = (blockIdx.x * 16 + cp) * VEC;
  const T* x = reinterpret_cast<const T*>(srcs.p[s]) + (int64_t)b * L * D;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (col0 < D) {
    for (int l = rg; l < L; l += 16) {
      Pack<T, VEC> p = ld_pack<T, VEC>(x + (int64_t)l * D + col0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += to_f<T>(p.v[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) sh[rg][cp * VEC + j] = acc[j];
  __syncthreads();
  const int c = threadIdx.x;  // one thread per column of the block's 16*VEC columns
  if (c < 16 * VEC) {
    const int col = blockIdx.x * 16 * VEC + c;
    if (col < D) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += sh[r][c];
      pooled[((int64_t)s * B + b) * D + col] = t / (float)L;
    }
  }
}

extern "C" int d2r_meanpool_fwd(int dtype, const void* const* h_srcs, int nsrc, int B, int L, int D, float* pooled,
                                void* stream) {
  D2R_REQUIRE(h_srcs && pooled, "d2r_meanpool_fwd: null pointer");
  D2R_REQUIRE(nsrc >= 1 && nsrc <= 8, "d2r_meanpool_fwd: nsrc=%d outside [1,8]", nsrc);
  D2R_REQUIRE(B >= 1 && L >= 1 && D >= 1, "d2r_meanpool_fwd: bad shape");
  const int VEC = dtype != D2R_F32 ? 8 : 4;
  D2R_REQUIRE(D % VEC == 0, "d2r_meanpool_fwd: D=%d must be a multiple of %d", D, VEC);
  Ptrs8 ps;
  for (int i = 0; i < 8; ++i) ps.p[i] = i < nsrc ? h_srcs[i] : nullptr;
  for (int i = 0; i < nsrc; ++i) D2R_REQUIRE(ps.p[i] && d2r_aligned16(ps.p[i]), "d2r_meanpool_fwd: source %d null or unaligned", i);
  dim3 grid(d2r_cdiv(D, 16 * VEC), B, nsrc), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) hipLaunchKernelGGL((meanpool_fwd_kernel<bf16_t>), grid, block, 0, st, ps, B, L, D, pooled);
  else if (dtype == D2R_F16) hipLaunchKernelGGL((meanpool_fwd_kernel<f16_t>), grid, block, 0, st, ps, B, L, D, pooled);
  else if (dtype == D2R_F32) hipLaunchKernelGGL((meanpool_fwd_kernel<float>), grid, block, 0, st, ps, B, L, D, pooled);
  else return d2r_fail(D2R_ERR_INVALID, "d2r_meanpool_fwd: bad dtype %d", dtype);
  return d2r_check_launch("d2r_meanpool_fwd");
}

template <typename T>
__global__ __launch_bounds__(256) void meanpool_bwd_kernel(const float* __restrict__ dpooled, int B, int L, int D,
                                                           T* __restrict__ dX, int accumulate) {
  constexpr int VEC = PackOf<T>::N;
  const int npk = D / VEC;
  const int64_t total = (int64_t)B * L * npk;
  const float invL = 1.f / (float)L;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int pk = (int)(idx % npk);
    const int64_t bl = idx / npk;
    const int b = (int)(bl / L);
    const float* g = dpooled + (int64_t)b * D + pk * VEC;
    T* dst = dX + bl * D + pk * VEC;
    Pack<T, VEC> o;
    if (accumulate) {
      Pack<T, VEC> old = ld_pack<T, VEC>(dst);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(to_f<T>(old.v[j]) + g[j] * invL);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(g[j] * invL);
    }
    st_pack<T, VEC>(dst, o);
  }
}

extern "C" int d2r_meanpool_bwd(int dtype, const float* dpooled, int B, int L, int D, void* dX, int accumulate,
                                void* stream) {
  D2R_REQUIRE(dpooled && dX && d2r_aligned16(dX), "d2r_meanpool_bwd: null or unaligned pointer");
  const int VEC = dtype != D2R_F32 ? 8 : 4;
  D2R_REQUIRE(B >= 1 && L >= 1 && D % VEC == 0, "d2r_meanpool_bwd: bad shape");
  int blocks = d2r_cdiv((int64_t)B * L * (D / VEC), 256);
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) hipLaunchKernelGGL((meanpool_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, dpooled, B, L, D, (bf16_t*)dX, accumulate);
  else if (dtype == D2R_F16) hipLaunchKernelGGL((meanpool_bwd_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, dpooled, B, L, D, (f16_t*)dX, accumulate);
  else if (dtype == D2R_F32) hipLaunchKernelGGL((meanpool_bwd_kernel<float>), dim3(blocks), dim3(256), 0, st, dpooled, B, L, D, (float*)dX, accumulate);
  else return d2r_fail(D2R_ERR_INVALID, "d2r_meanpool_bwd: bad dtype %d", dtype);
  return d2r_check_launch("d2r_meanpool_bwd");
}

// The same broadcast for up to 8 pooled gradients [n, B, D] into n different [B, L, D] tensors in ONE launch (blockIdx.y = source;
// bit j of acc_mask: accumulate into dX[j]): the routers of a middle / final routing layer pool six different inputs.
template <typename T>
__global__ __launch_bounds__(256) void meanpool_bwd_multi_kernel(const float* __restrict__ dpooled, int B, int L, int D, MPtrs8 dX,
                                                                 unsigned acc_mask) {
  constexpr int VEC = PackOf<T>::N;
  const int npk = D / VEC, src = blockIdx.y;
  const int64_t total = (int64_t)B * L * npk;
  const float invL = 1.f / (float)L;
  const float* gp = dpooled + (int64_t)src * B * D;
  T* out = reinterpret_cast<T*>(dX.p[src]);
  const bool accumulate = (acc_mask >> src) & 1u;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int pk = (int)(idx % npk);
    const int64_t bl = idx / npk;
    const int b = (int)(bl / L);
    const float* g = gp + (int64_t)b * D + pk * VEC;
    T* dst = out + bl * D + pk * VEC;
    Pack<T, VEC> o;
    if (accumulate) {
      Pack<T, VEC> old = ld_pack<T, VEC>(dst);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(to_f<T>(old.v[j]) + g[j] * invL);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(g[j] * invL);
    }
    st_pack<T, VEC>(dst, o);
  }
}

extern "C" int d2r_meanpool_bwd_multi(int dtype, const float* dpooled, int n, int B, int L, int D, void* const* dX, unsigned acc_mask,
                                      void* stream) {
  D2R_REQUIRE(dpooled && dX && n >= 1 && n <= 8, "d2r_meanpool_bwd_multi: bad argument");
  D2R_REQUIRE(d2r_is16(dtype) && B >= 1 && L >= 1 && D % 8 == 0, "d2r_meanpool_bwd_multi: 16-bit dtypes, D a multiple of 8");
  MPtrs8 t = {};
  for (int j = 0; j < n; ++j) {
    D2R_REQUIRE(dX[j] && d2r_aligned16(dX[j]), "d2r_meanpool_bwd_multi: null or unaligned pointer %d", j);
    for (int k = 0; k < j; ++k) D2R_REQUIRE(dX[j] != dX[k], "d2r_meanpool_bwd_multi: outputs %d and %d alias", k, j);
    t.p[j] = dX[j];
  }
  int blocks = d2r_cdiv((int64_t)B * L * (D / 8), 256);
  if (blocks > 1024) blocks = 1024;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) hipLaunchKernelGGL((meanpool_bwd_multi_kernel<bf16_t>), dim3(blocks, n), dim3(256), 0, st, dpooled, B, L, D, t, acc_mask);
  else hipLaunchKernelGGL((meanpool_bwd_multi_kernel<f16_t>), dim3(blocks, n), dim3(256), 0, st, dpooled, B, L, D, t, acc_mask);
  return d2r_check_launch("d2r_meanpool_bwd_multi");
}

// =====================================================================================================
// K8 forward.  gates: fp32 [B, nc, P] (sample-major: the layout the routers' grouped GEMM writes).  Cells 1 (GLAC) and 5 (GESC) are [B,D] broadcasts.
// nc = number of cells of the layer: the first nc of [RIC, GLAC, IMRC, CMRC, CRCMC, GESC] (6 in the reference, which
// hard-indexes them, models/DynamicInteraction.py:41-48; 2..5 = the declared-subset extension of SURVEY.md section 8c);
// P = nc outputs (first / middle layers) or 1 (final layer).  probs: [B, P, nc].
// =====================================================================================================
template <typename T, int VEC>
__device__ __forceinline__ void ld_f(const T* p, float (&o)[VEC]) {
  Pack<T, VEC> k = ld_pack<T, VEC>(p);
#pragma unroll
  for (int j = 0; j < VEC; ++j) o[j] = to_f<T>(k.v[j]);
}
template <typename T, int VEC>
__device__ __forceinline__ void st_f(T* p, const float (&o)[VEC]) {
  Pack<T, VEC> k;
#pragma unroll
  for (int j = 0; j < VEC; ++j) k.v[j] = from_f<T>(o[j]);
  st_pack<T, VEC>(p, k);
}

__device__ __forceinline__ float uniform_f(float v) {  // block-uniform value kept in a scalar register
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

template <typename T>
__global__ __launch_bounds__(256) void agg_fwd6_kernel(Ptrs8 embs, const float* __restrict__ gates, int B, int L, int D,
                                                       int nc, MPtrs8 outs, float* __restrict__ probs, int64_t ldp) {
  constexpr int VEC = PackOf<T>::N;
  __shared__ float csh[6][6];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid < 6) {
    const int i = tid;
    float g[6], S = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      g[j] = (j < nc && i < nc) ? gates[((int64_t)b * nc + j) * nc + i] : 0.f;
      S += g[j];
    }
    const float skip = S < TH_GATE ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float ph = g[j] / (S + EPS_NORM);
      csh[i][j] = ph + (j == 0 ? skip : 0.f);
      if (blockIdx.x == 0 && i < nc && j < nc) probs[(int64_t)b * ldp + i * nc + j] = ph;
    }
  }
  __syncthreads();
  float c[6][6];  // the sample's 36 coefficients, in scalar registers
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int k = 0; k < 6; ++k) c[i][k] = uniform_f(csh[i][k]);
  const int npk = D / VEC;
  const int total = L * npk;
  for (int idx = blockIdx.x * 256 + tid; idx < total; idx += gridDim.x * 256) {
    const int l = idx / npk, pk = idx - l * npk;
    const int64_t off = ((int64_t)b * L + l) * D + pk * VEC, boff = (int64_t)b * D + pk * VEC;
    float e[6][VEC];
    // six loads issued together (absent cells of a declared subset: the host passes a stand-in pointer, the value is
    // zeroed by a select; a branch around a load would put a full wait after each)
#pragma unroll
    for (int k = 0; k < 6; ++k)
      ld_f<T, VEC>(reinterpret_cast<const T*>(embs.p[k]) + ((k == 1 || k == 5) ? boff : off), e[k]);
#pragma unroll
    for (int j = 0; j < VEC; ++j) e[0][j] = fmaxf(e[0][j], 0.f);
#pragma unroll
    for (int k = 1; k < 6; ++k)
#pragma unroll
      for (int j = 0; j < VEC; ++j) e[k][j] = k < nc ? e[k][j] : 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i >= nc) break;
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float t = c[i][0] * e[0][j];
#pragma unroll
        for (int k = 1; k < 6; ++k) t += c[i][k] * e[k][j];
        o[j] = t;
      }
      st_f<T, VEC>(reinterpret_cast<T*>(outs.p[i]) + off, o);
    }
  }
}

// final layer (P = 1): out = sum_j (g_j emb_j + s_j ref_j) / (sum s + sum g)
template <typename T>
__global__ __launch_bounds__(256) void agg_fwd1_kernel(Ptrs8 embs, Ptrs8 refs, const float* __restrict__ gates, int B,
                                                       int L, int D, int nc, T* __restrict__ out, float* __restrict__ probs, int64_t ldp) {
  constexpr int VEC = PackOf<T>::N;
  __shared__ float cg[6], cs[6];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid == 0) {
    float g[6], s[6], sg = 0.f, ss = 0.f;
    const float thf = th_gate_final(nc);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      g[j] = j < nc ? gates[(int64_t)b * nc + j] : 0.f;
      s[j] = (j < nc && g[j] < thf) ? 1.f : 0.f;
      sg += g[j];
      ss += s[j];
      if (blockIdx.x == 0 && j < nc) probs[(int64_t)b * ldp + j] = g[j];
    }
    const float inv = 1.f / (ss + sg);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      cg[j] = g[j] * inv;
      cs[j] = s[j] * inv;
    }
  }
  __syncthreads();
  const int npk = D / VEC;
  const int total = L * npk;
  for (int idx = blockIdx.x * 256 + tid; idx < total; idx += gridDim.x * 256) {
    const int l = idx / npk, pk = idx - l * npk;
    const int64_t off = ((int64_t)b * L + l) * D + pk * VEC, boff = (int64_t)b * D + pk * VEC;
    float o[VEC], e[VEC];
    ld_f<T, VEC>(reinterpret_cast<const T*>(embs.p[0]) + off, e);  // x0 == ref_0
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = cg[0] * fmaxf(e[j], 0.f) + cs[0] * e[j];
#pragma unroll
    for (int k = 1; k < 6; ++k) {
      if (k >= nc) break;
      const bool bc = (k == 1 || k == 5);
      ld_f<T, VEC>(reinterpret_cast<const T*>(embs.p[k]) + (bc ? boff : off), e);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] += cg[k] * e[j];
      if (cs[k] != 0.f) {  // block-uniform: the skip term is read only for closed paths
        ld_f<T, VEC>(reinterpret_cast<const T*>(refs.p[k]) + off, e);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] += cs[k] * e[j];
      }
    }
    st_f<T, VEC>(out + off, o);
  }
}

static int agg_chunks(int L, int D, int VEC) {
  int64_t packs = (int64_t)L * (D / VEC);
  int c = (int)((packs + 1023) / 1024);  // ~4 packs per thread
  return c < 1 ? 1 : c;
}

extern "C" int d2r_route_aggregate_fwd(int dtype, const void* const* h_embs, const void* const* h_refs,
                                       const float* gates, int B, int L, int D, int ncell, int P, void* const* h_outs,
                                       float* probs, int64_t ld_probs, void* stream) {
  D2R_REQUIRE(h_embs && gates && h_outs && probs, "d2r_route_aggregate_fwd: null pointer");
  D2R_REQUIRE(ncell >= 2 && ncell <= 6, "d2r_route_aggregate_fwd: ncell=%d (2..6)", ncell);
  D2R_REQUIRE(P == ncell || P == 1, "d2r_route_aggregate_fwd: P=%d (must be ncell=%d or 1)", P, ncell);
  D2R_REQUIRE(P != 1 || h_refs, "d2r_route_aggregate_fwd: the final layer needs refs");
  D2R_REQUIRE(ld_probs >= (int64_t)P * ncell, "d2r_route_aggregate_fwd: ld_probs %lld < P*ncell", (long long)ld_probs);
  D2R_REQUIRE(dtype == D2R_F32 || dtype == D2R_BF16 || dtype == D2R_F16, "d2r_route_aggregate_fwd: bad dtype %d", dtype);
  const int VEC = dtype != D2R_F32 ? 8 : 4;
  D2R_REQUIRE(B >= 1 && B <= 65535 && L >= 1 && D % VEC == 0, "d2r_route_aggregate_fwd: bad shape B=%d L=%d D=%d", B, L, D);
  Ptrs8 e{}, r{};
  MPtrs8 o{};
  for (int j = 0; j < ncell; ++j) {
    e.p[j] = h_embs[j];
    D2R_REQUIRE(e.p[j] && d2r_aligned16(e.p[j]), "d2r_route_aggregate_fwd: emb %d null or unaligned", j);
    if (P == 1) {
      r.p[j] = h_refs[j];
      D2R_REQUIRE(r.p[j] && d2r_aligned16(r.p[j]), "d2r_route_aggregate_fwd: ref %d null or unaligned", j);
    }
  }
  for (int i = 0; i < P; ++i) {
    o.p[i] = h_outs[i];
    D2R_REQUIRE(o.p[i] && d2r_aligned16(o.p[i]), "d2r_route_aggregate_fwd: out %d null or unaligned", i);
  }
  dim3 grid(agg_chunks(L, D, VEC), B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (P != 1) {
    for (int k = ncell; k < 6; ++k) e.p[k] = (k == 1 || k == 5) ? e.p[1] : e.p[0];  // stand-ins for absent cells (read, then zeroed)
    if (dtype == D2R_BF16) hipLaunchKernelGGL((agg_fwd6_kernel<bf16_t>), grid, block, 0, st, e, gates, B, L, D, ncell, o, probs, ld_probs);
    else if (dtype == D2R_F16) hipLaunchKernelGGL((agg_fwd6_kernel<f16_t>), grid, block, 0, st, e, gates, B, L, D, ncell, o, probs, ld_probs);
    else hipLaunchKernelGGL((agg_fwd6_kernel<float>), grid, block, 0, st, e, gates, B, L, D, ncell, o, probs, ld_probs);
  } else {
    if (dtype == D2R_BF16) hipLaunchKernelGGL((agg_fwd1_kernel<bf16_t>), grid, block, 0, st, e, r, gates, B, L, D, ncell, (bf16_t*)o.p[0], probs, ld_probs);
    else if (dtype == D2R_F16) hipLaunchKernelGGL((agg_fwd1_kernel<f16_t>), grid, block, 0, st, e, r, gates, B, L, D, ncell, (f16_t*)o.p[0], probs, ld_probs);
    else hipLaunchKernelGGL((agg_fwd1_kernel<float>), grid, block, 0, st, e, r, gates, B, L, D, ncell, (float*)o.p[0], probs, ld_probs);
  }
  return d2r_check_launch("d2r_route_aggregate_fwd");
}

// =====================================================================================================
// K8 backward.  Block = (token chunk, sample); a thread owns one column pack and walks the chunk's rows.
// Per block: NDOT dot-product partials + NBC broadcast-gradient partial rows -> workspace; stage 2 finishes.
//   workspace layout: dots [B][nchunk][NDOT] | bcast [B][nchunk][NBC][D]
// =====================================================================================================
#define AGG_LC 8  /* token rows per block */

template <int N>
__device__ __forceinline__ void block_reduce_store(float (&acc)[N], float* sh /*[waves][N]*/, float* dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float t = wave_sum(acc[k]);
    if (lane == 0) sh[wave * N + k] = t;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += sh[w * N + threadIdx.x];
    dst[threadIdx.x] = t;
  }
  __syncthreads();
}

// One block = AGG_LC token rows of one sample; blockDim = RG * (D / VEC) threads (RG row groups of D/VEC column packs).
// VEC = 4 elements per thread (8-byte loads for the 16-bit types): the 36 dot accumulators, the six incoming gradients
// and the four embeddings of a row then fit 128 registers (four waves per SIMD; the 8-wide version needed all 512 and
// ran one), and the 36 path coefficients of the sample sit in scalar registers.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void agg_bwd6_kernel(Ptrs8 embs, const float* __restrict__ gates, Ptrs8 douts, int B,
                                                       int L, int D, int nc, MPtrs8 dembs, float* __restrict__ ws_dots,
                                                       float* __restrict__ ws_bc) {
  extern __shared__ float dsh[];  // [RG][2][D] broadcast partials, then reused for the dot reduction
  __shared__ float csh[6][6];
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x, tid = threadIdx.x;
  if (tid < 6) {
    const int i = tid;
    float g[6], S = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      g[j] = (j < nc && i < nc) ? gates[((int64_t)b * nc + j) * nc + i] : 0.f;
      S += g[j];
    }
    const float skip = (i < nc && S < TH_GATE) ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      csh[i][j] = g[j] / (S + EPS_NORM) + (j == 0 ? skip : 0.f);  // (absent output i >= nc: all zero)
    }
  }
  __syncthreads();
  float c[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int k = 0; k < 6; ++k) c[i][k] = uniform_f(csh[i][k]);
  const int npk = D / VEC;
  const int RG = blockDim.x / npk;  // row groups per block (>= 1, checked on the host)
  const int pk = tid % npk, rg = tid / npk;
  const bool active = rg < RG;
  float dots[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) dots[k] = 0.f;
  float b1[VEC], b5[VEC], e1[VEC], e5[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) b1[j] = b5[j] = e1[j] = e5[j] = 0.f;
  const int l0 = chunk * AGG_LC, l1 = min(L, l0 + AGG_LC);
  if (active) {
    const int64_t boff = (int64_t)b * D + pk * VEC;
    ld_f<T, VEC>(reinterpret_cast<const T*>(embs.p[1]) + boff, e1);
    if (nc > 5) ld_f<T, VEC>(reinterpret_cast<const T*>(embs.p[5]) + boff, e5);
    for (int l = l0 + rg; l < l1; l += RG) {
      const int64_t off = ((int64_t)b * L + l) * D + pk * VEC;
      // all ten loads of the token row are issued before the first use (one memory latency per row).  Holding the next
      // row's packs in flight as well costs 30 registers (two waves per SIMD instead of three) and measured slower.
      float dv[6][VEC], x0[VEC], ek[3][VEC], o[VEC];
      // (absent cells: the host passes a valid stand-in pointer, the values are zeroed by a select - a branch around a
      // load would put a full wait after each of them)
#pragma unroll
      for (int i = 0; i < 6; ++i) ld_f<T, VEC>(reinterpret_cast<const T*>(douts.p[i]) + off, dv[i]);
      ld_f<T, VEC>(reinterpret_cast<const T*>(embs.p[0]) + off, x0);
#pragma unroll
      for (int k = 2; k <= 4; ++k) ld_f<T, VEC>(reinterpret_cast<const T*>(embs.p[k]) + off, ek[k - 2]);
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) dv[i][j] = i < nc ? dv[i][j] : 0.f;
#pragma unroll
      for (int k = 2; k <= 4; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) ek[k - 2][j] = k < nc ? ek[k - 2][j] : 0.f;
      // cell 0 (RIC): emb = relu(x0)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float r = fmaxf(x0[j], 0.f);
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          dots[i * 6 + 0] += dv[i][j] * r;
          t += c[i][0] * dv[i][j];
        }
        o[j] = x0[j] > 0.f ? t : 0.f;
      }
      st_f<T, VEC>(reinterpret_cast<T*>(dembs.p[0]) + off, o);
      // full cells 2,3,4
#pragma unroll
      for (int k = 2; k <= 4; ++k) {
        if (k >= nc) break;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            dots[i * 6 + k] += dv[i][j] * ek[k - 2][j];
            t += c[i][k] * dv[i][j];
          }
          o[j] = t;
        }
        st_f<T, VEC>(reinterpret_cast<T*>(dembs.p[k]) + off, o);
      }
      // broadcast cells 1,5
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float t1 = 0.f, t5 = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          dots[i * 6 + 1] += dv[i][j] * e1[j];
          dots[i * 6 + 5] += dv[i][j] * e5[j];
          t1 += c[i][1] * dv[i][j];
          t5 += c[i][5] * dv[i][j];
        }
        b1[j] += t1;
        b5[j] += t5;
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      dsh[(rg * 2 + 0) * D + pk * VEC + j] = b1[j];
      dsh[(rg * 2 + 1) * D + pk * VEC + j] = b5[j];
    }
  }
  __syncthreads();
  float* wb = ws_bc + ((int64_t)b * nchunk + chunk) * 2 * D;
  for (int cidx = tid; cidx < 2 * D; cidx += blockDim.x) {
    float t = 0.f;
    for (int r = 0; r < RG; ++r) t += dsh[r * 2 * D + cidx];
    wb[cidx] = t;
  }
  __syncthreads();
  block_reduce_store<36>(dots, dsh, ws_dots + ((int64_t)b * nchunk + chunk) * 36);
}

// stage 2 (P=6): grid = B.  d_gates[j][b][i] from dp_hat (+ dprobs), broadcast grads summed over chunks.
template <typename T>
__global__ __launch_bounds__(256) void agg_bwd6_finish_kernel(const float* __restrict__ gates,
                                                              const float* __restrict__ dprobs,
                                                              const float* __restrict__ ws_dots,
                                                              const float* __restrict__ ws_bc, int B, int D, int nchunk,
                                                              int nc, int64_t ldp, T* __restrict__ demb1, T* __restrict__ demb5,
                                                              float* __restrict__ d_gates) {
  __shared__ float dph[36];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (blockIdx.y == 0) {  // block-uniform: only the first column block finishes the 36 dot products
  if (tid < 36) {
    const int i = tid / 6, j = tid - i * 6;
    float t = (dprobs && i < nc && j < nc) ? dprobs[(int64_t)b * ldp + i * nc + j] : 0.f;
    // eight chunk partials per trip, all loaded before the first add (one memory latency per trip instead of one per chunk;
    // fixed order of additions)
    const float* wd = ws_dots + (int64_t)b * nchunk * 36 + tid;
    for (int c0 = 0; c0 < nchunk; c0 += 8) {
      float p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = c0 + u < nchunk ? wd[(int64_t)(c0 + u) * 36] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) t += p[u];
    }
    dph[tid] = t;
  }
  __syncthreads();
  if (tid < nc) {
    const int i = tid;
    float g[6], S = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      g[j] = j < nc ? gates[((int64_t)b * nc + j) * nc + i] : 0.f;
      S += g[j];
    }
    const float inv = 1.f / (S + EPS_NORM);
    float dotp = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) dotp += dph[i * 6 + j] * (g[j] * inv);
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j < nc) d_gates[((int64_t)b * nc + j) * nc + i] = (dph[i * 6 + j] - dotp) * inv;
  }
  }
  const int cidx = blockIdx.y * 256 + tid;  // one column of [demb1 | demb5] per thread, grid.y column blocks
  if (cidx < 2 * D) {
    const float* w = ws_bc + (int64_t)b * nchunk * 2 * D + cidx;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int k = 0;
    for (; k + 3 < nchunk; k += 4) {
      t0 += w[(int64_t)k * 2 * D];
      t1 += w[(int64_t)(k + 1) * 2 * D];
      t2 += w[(int64_t)(k + 2) * 2 * D];
      t3 += w[(int64_t)(k + 3) * 2 * D];
    }
    for (; k < nchunk; ++k) t0 += w[(int64_t)k * 2 * D];
    const float t = (t0 + t1) + (t2 + t3);
    if (cidx < D) demb1[(int64_t)b * D + cidx] = from_f<T>(t);
    else if (demb5) demb5[(int64_t)b * D + cidx - D] = from_f<T>(t);
  }
}

// final layer backward (P = 1)
template <typename T>
__global__ __launch_bounds__(256) void agg_bwd1_kernel(Ptrs8 embs, Ptrs8 refs, const float* __restrict__ gates,
                                                       const T* __restrict__ dout, const T* __restrict__ out, int B,
                                                       int L, int D, int nc, MPtrs8 dembs, MPtrs8 drefs,
                                                       float* __restrict__ ws_dots, float* __restrict__ ws_bc) {
  constexpr int VEC = PackOf<T>::N;
  extern __shared__ float dsh[];  // [RG][D]
  __shared__ float cg[6], cs[6];
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x, tid = threadIdx.x;
  if (tid == 0) {
    float g[6], s[6], sg = 0.f, ss = 0.f;
    const float thf = th_gate_final(nc);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      g[j] = j < nc ? gates[(int64_t)b * nc + j] : 0.f;
      s[j] = (j < nc && g[j] < thf) ? 1.f : 0.f;
      sg += g[j];
      ss += s[j];
    }
    const float inv = 1.f / (ss + sg);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      cg[j] = g[j] * inv;
      cs[j] = s[j] * inv;
    }
  }
  __syncthreads();
  const int npk = D / VEC;
  const int RG = 256 / npk;
  const int pk = tid % npk, rg = tid / npk;
  const bool active = rg < RG;
  float dots[8];  // 0..5: <dout, emb_j>, 6: <dout, out>, 7: unused
#pragma unroll
  for (int k = 0; k < 8; ++k) dots[k] = 0.f;
  float bs[VEC], e1[VEC], e5[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) bs[j] = e1[j] = e5[j] = 0.f;
  const int l0 = chunk * AGG_LC, l1 = min(L, l0 + AGG_LC);
  const float cg0 = uniform_f(cg[0]);
  float cgk[3], csk[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) cgk[k] = uniform_f(cg[k + 2]);
#pragma unroll
  for (int k = 0; k < 6; ++k) csk[k] = uniform_f(cs[k]);
  if (active) {
    const int64_t boff = (int64_t)b * D + pk * VEC;
    ld_f<T, VEC>(reinterpret_cast<const T*>(embs.p[1]) + boff, e1);
    if (nc > 5) ld_f<T, VEC>(reinterpret_cast<const T*>(embs.p[5]) + boff, e5);
    for (int l = l0 + rg; l < l1; l += RG) {
      const int64_t off = ((int64_t)b * L + l) * D + pk * VEC;
      // the six loads of the token row are issued before the first store (absent cells: stand-in pointer, zeroed by a select)
      float dv[VEC], eo[VEC], x0[VEC], ek[3][VEC], o[VEC];
      ld_f<T, VEC>(dout + off, dv);
      ld_f<T, VEC>(out + off, eo);
      ld_f<T, VEC>(reinterpret_cast<const T*>(embs.p[0]) + off, x0);  // x0 = ref_0
#pragma unroll
      for (int k = 2; k <= 4; ++k) ld_f<T, VEC>(reinterpret_cast<const T*>(embs.p[k]) + off, ek[k - 2]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        dots[6] += dv[j] * eo[j];
        dots[1] += dv[j] * e1[j];
        dots[5] += dv[j] * e5[j];
        bs[j] += dv[j];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        dots[0] += dv[j] * fmaxf(x0[j], 0.f);
        o[j] = x0[j] > 0.f ? cg0 * dv[j] : 0.f;
      }
      st_f<T, VEC>(reinterpret_cast<T*>(dembs.p[0]) + off, o);
#pragma unroll
      for (int k = 2; k <= 4; ++k) {
        if (k >= nc) break;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          dots[k] += dv[j] * ek[k - 2][j];
          o[j] = cgk[k - 2] * dv[j];
        }
        st_f<T, VEC>(reinterpret_cast<T*>(dembs.p[k]) + off, o);
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        if (k >= nc) break;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = csk[k] * dv[j];
        st_f<T, VEC>(reinterpret_cast<T*>(drefs.p[k]) + off, o);
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) dsh[rg * D + pk * VEC + j] = bs[j];
  }
  __syncthreads();
  float* wb = ws_bc + ((int64_t)b * nchunk + chunk) * D;
  for (int cidx = tid; cidx < D; cidx += 256) {
    float t = 0.f;
    for (int r = 0; r < RG; ++r) t += dsh[r * D + cidx];
    wb[cidx] = t;
  }
  __syncthreads();
  block_reduce_store<8>(dots, dsh, ws_dots + ((int64_t)b * nchunk + chunk) * 8);
}

template <typename T>
__global__ __launch_bounds__(256) void agg_bwd1_finish_kernel(const float* __restrict__ gates,
                                                              const float* __restrict__ dprobs,
                                                              const float* __restrict__ ws_dots,
                                                              const float* __restrict__ ws_bc, int B, int D, int nchunk,
                                                              int nc, int64_t ldp, T* __restrict__ demb1, T* __restrict__ demb5,
                                                              float* __restrict__ d_gates) {
  __shared__ float dt[8], cgs[6];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < 8) {
    float t = 0.f;
    const float* wd = ws_dots + (int64_t)b * nchunk * 8 + tid;
    for (int c0 = 0; c0 < nchunk; c0 += 8) {  // (eight partials per trip in flight together, fixed order of additions)
      float p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = c0 + u < nchunk ? wd[(int64_t)(c0 + u) * 8] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) t += p[u];
    }
    dt[tid] = t;
  }
  __syncthreads();
  if (tid == 0) {  // every column block recomputes the coefficients; only block y==0 stores d_gates
    float g[6], sg = 0.f, ss = 0.f;
    const float thf = th_gate_final(nc);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      g[j] = j < nc ? gates[(int64_t)b * nc + j] : 0.f;
      ss += (j < nc && g[j] < thf) ? 1.f : 0.f;
      sg += g[j];
    }
    const float inv = 1.f / (ss + sg);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      cgs[j] = g[j] * inv;
      if (blockIdx.y == 0 && j < nc) d_gates[(int64_t)b * nc + j] = (dt[j] - dt[6]) * inv + (dprobs ? dprobs[(int64_t)b * ldp + j] : 0.f);
    }
  }
  __syncthreads();
  const int cidx = blockIdx.y * 256 + tid;
  if (cidx < D) {
    const float* w = ws_bc + (int64_t)b * nchunk * D + cidx;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int k = 0;
    for (; k + 3 < nchunk; k += 4) {
      t0 += w[(int64_t)k * D];
      t1 += w[(int64_t)(k + 1) * D];
      t2 += w[(int64_t)(k + 2) * D];
      t3 += w[(int64_t)(k + 3) * D];
    }
    for (; k < nchunk; ++k) t0 += w[(int64_t)k * D];
    const float t = (t0 + t1) + (t2 + t3);
    demb1[(int64_t)b * D + cidx] = from_f<T>(cgs[1] * t);
    if (demb5) demb5[(int64_t)b * D + cidx] = from_f<T>(cgs[5] * t);
  }
}

extern "C" size_t d2r_route_aggregate_bwd_workspace(int B, int L, int D, int P) {
  const size_t nchunk = (size_t)((L + AGG_LC - 1) / AGG_LC);
  return (size_t)B * nchunk * ((P != 1 ? 36 : 8) + (size_t)(P != 1 ? 2 : 1) * D) * sizeof(float);
}

// d_probs: gradient flowing into the returned path probabilities (sim_paths -> JS loss), may be NULL;
// h_outs: the forward outputs (only outs[0] of the final layer is read).
extern "C" int d2r_route_aggregate_bwd(int dtype, const void* const* h_embs, const void* const* h_refs,
                                          const float* gates, const void* const* h_douts, const void* const* h_outs,
                                          const float* d_probs, int64_t ld_dprobs, int B, int L, int D, int ncell, int P,
                                          void* const* h_dembs, void* const* h_drefs, float* d_gates, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  D2R_REQUIRE(h_embs && gates && h_douts && h_dembs && d_gates, "d2r_route_aggregate_bwd: null pointer");
  D2R_REQUIRE(ncell >= 2 && ncell <= 6, "d2r_route_aggregate_bwd: ncell=%d (2..6)", ncell);
  D2R_REQUIRE(P == ncell || P == 1, "d2r_route_aggregate_bwd: P=%d (must be ncell=%d or 1)", P, ncell);
  D2R_REQUIRE(P != 1 || (h_refs && h_drefs && h_outs), "d2r_route_aggregate_bwd: the final layer needs refs, d_refs and outs");
  D2R_REQUIRE(dtype == D2R_F32 || dtype == D2R_BF16 || dtype == D2R_F16, "d2r_route_aggregate_bwd: bad dtype %d", dtype);
  const int VEC = dtype != D2R_F32 ? 8 : 4;
  D2R_REQUIRE(B >= 1 && B <= 65535 && L >= 1 && D % VEC == 0 && D / VEC <= 256, "d2r_route_aggregate_bwd: bad shape B=%d L=%d D=%d", B, L, D);
  if (!workspace || workspace_bytes < d2r_route_aggregate_bwd_workspace(B, L, D, P))
    return d2r_fail(D2R_ERR_WORKSPACE, "d2r_route_aggregate_bwd: workspace %zu < %zu", workspace_bytes, d2r_route_aggregate_bwd_workspace(B, L, D, P));
  Ptrs8 e{}, r{}, dv{};
  MPtrs8 de{}, dr{};
  for (int j = 0; j < ncell; ++j) {
    e.p[j] = h_embs[j];
    de.p[j] = h_dembs[j];
    D2R_REQUIRE(e.p[j] && de.p[j] && d2r_aligned16(e.p[j]) && d2r_aligned16(de.p[j]), "d2r_route_aggregate_bwd: emb/d_emb %d null or unaligned", j);
    if (P == 1) {
      r.p[j] = h_refs[j];
      dr.p[j] = h_drefs[j];
      D2R_REQUIRE(r.p[j] && dr.p[j] && d2r_aligned16(r.p[j]) && d2r_aligned16(dr.p[j]), "d2r_route_aggregate_bwd: ref/d_ref %d null or unaligned", j);
    }
  }
  for (int i = 0; i < P; ++i) {
    dv.p[i] = h_douts[i];
    D2R_REQUIRE(dv.p[i] && d2r_aligned16(dv.p[i]), "d2r_route_aggregate_bwd: dout %d null or unaligned", i);
  }
  // stand-ins for absent cells (read, then ignored by the kernels)
  for (int i = P; i < 6 && P != 1; ++i) dv.p[i] = dv.p[0];
  for (int k = ncell; k <= 4; ++k) e.p[k] = e.p[0];
  const int nchunk = (L + AGG_LC - 1) / AGG_LC;
  const int RG = 256 / (D / VEC);
  constexpr int V6 = 4;  // elements per thread of the P = ncell kernel (see agg_bwd6_kernel)
  const int npk6 = D / V6, RG6 = npk6 <= 256 ? 256 / npk6 : 0;
  float* ws_dots = (float*)workspace;
  float* ws_bc = ws_dots + (size_t)B * nchunk * (P != 1 ? 36 : 8);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nchunk, B), block(256);
  if (P != 1) {
    D2R_REQUIRE(D % V6 == 0 && RG6 >= 1, "d2r_route_aggregate_bwd: D=%d (multiple of 4, at most 1024)", D);
    const dim3 block6((RG6 * npk6 + 63) / 64 * 64);
    size_t shmem = (size_t)RG6 * 2 * D * sizeof(float);
    if (shmem < 4 * 36 * sizeof(float)) shmem = 4 * 36 * sizeof(float);
    if (dtype == D2R_BF16) {
      hipLaunchKernelGGL((agg_bwd6_kernel<bf16_t, V6>), grid, block6, shmem, st, e, gates, dv, B, L, D, ncell, de, ws_dots, ws_bc);
      hipLaunchKernelGGL((agg_bwd6_finish_kernel<bf16_t>), dim3(B, d2r_cdiv(2 * D, 256)), block, 0, st, gates, d_probs, ws_dots, ws_bc, B, D, nchunk, ncell, ld_dprobs, (bf16_t*)de.p[1], (bf16_t*)de.p[5], d_gates);
    } else if (dtype == D2R_F16) {
      hipLaunchKernelGGL((agg_bwd6_kernel<f16_t, V6>), grid, block6, shmem, st, e, gates, dv, B, L, D, ncell, de, ws_dots, ws_bc);
      hipLaunchKernelGGL((agg_bwd6_finish_kernel<f16_t>), dim3(B, d2r_cdiv(2 * D, 256)), block, 0, st, gates, d_probs, ws_dots, ws_bc, B, D, nchunk, ncell, ld_dprobs, (f16_t*)de.p[1], (f16_t*)de.p[5], d_gates);
    } else {
      hipLaunchKernelGGL((agg_bwd6_kernel<float, V6>), grid, block6, shmem, st, e, gates, dv, B, L, D, ncell, de, ws_dots, ws_bc);
      hipLaunchKernelGGL((agg_bwd6_finish_kernel<float>), dim3(B, d2r_cdiv(2 * D, 256)), block, 0, st, gates, d_probs, ws_dots, ws_bc, B, D, nchunk, ncell, ld_dprobs, (float*)de.p[1], (float*)de.p[5], d_gates);
    }
  } else {
    const void* outp = h_outs[0];
    D2R_REQUIRE(outp && d2r_aligned16(outp), "d2r_route_aggregate_bwd: out null or unaligned");
    size_t shmem = (size_t)RG * D * sizeof(float);
    if (shmem < 4 * 8 * sizeof(float)) shmem = 4 * 8 * sizeof(float);
    if (dtype == D2R_BF16) {
      hipLaunchKernelGGL((agg_bwd1_kernel<bf16_t>), grid, block, shmem, st, e, r, gates, (const bf16_t*)dv.p[0], (const bf16_t*)outp, B, L, D, ncell, de, dr, ws_dots, ws_bc);
      hipLaunchKernelGGL((agg_bwd1_finish_kernel<bf16_t>), dim3(B, d2r_cdiv(D, 256)), block, 0, st, gates, d_probs, ws_dots, ws_bc, B, D, nchunk, ncell, ld_dprobs, (bf16_t*)de.p[1], (bf16_t*)de.p[5], d_gates);
    } else if (dtype == D2R_F16) {
      hipLaunchKernelGGL((agg_bwd1_kernel<f16_t>), grid, block, shmem, st, e, r, gates, (const f16_t*)dv.p[0], (const f16_t*)outp, B, L, D, ncell, de, dr, ws_dots, ws_bc);
      hipLaunchKernelGGL((agg_bwd1_finish_kernel<f16_t>), dim3(B, d2r_cdiv(D, 256)), block, 0, st, gates, d_probs, ws_dots, ws_bc, B, D, nchunk, ncell, ld_dprobs, (f16_t*)de.p[1], (f16_t*)de.p[5], d_gates);
    } else {
      hipLaunchKernelGGL((agg_bwd1_kernel<float>), grid, block, shmem, st, e, r, gates, (const float*)dv.p[0], (const float*)outp, B, L, D, ncell, de, dr, ws_dots, ws_bc);
      hipLaunchKernelGGL((agg_bwd1_finish_kernel<float>), dim3(B, d2r_cdiv(D, 256)), block, 0, st, gates, d_probs, ws_dots, ws_bc, B, D, nchunk, ncell, ld_dprobs, (float*)de.p[1], (float*)de.p[5], d_gates);
    }
  }
  return d2r_check_launch("d2r_route_aggregate_bwd");
}
