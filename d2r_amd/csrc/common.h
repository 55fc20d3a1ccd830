// common.h — shared device/host helpers for libd2r_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/d2r_hip.h"
#include "../../include/d2r_hip_probes.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

#define D2R_WAVE 64

// ---- error plumbing (thread-local message, no allocation) -------------------------------------
extern thread_local char d2r_err_buf[512];
int d2r_fail(int code, const char* fmt, ...);
int d2r_check_launch(const char* what);
int d2r_event_record(void* stream, void** ev);  // core.hip: the two halves of d2r_stream_fork (events from a library-owned ring)
int d2r_stream_wait(void* stream, void* ev);
int d2r_stream_fork(void* from, void* to);  // core.hip: `to` waits for what is enqueued on `from` (no-op when they are the same stream)

#define D2R_REQUIRE(cond, ...)                                  \
  do {                                                          \
    if (!(cond)) return d2r_fail(D2R_ERR_INVALID, __VA_ARGS__); \
  } while (0)

static inline bool d2r_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t d2r_esize(int dtype) { return dtype == D2R_F32 ? 4 : 2; }
static inline bool d2r_is16(int dtype) { return dtype == D2R_BF16 || dtype == D2R_F16; }

// ---- scalar conversions -------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return (float)v; }
template <> __device__ __forceinline__ float to_f<f16_t>(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return (bf16_t)v; }  // RNE, NaN-preserving
template <> __device__ __forceinline__ f16_t from_f<f16_t>(float v) { return (f16_t)v; }     // RNE; |v| > 65504 -> inf

// ---- 16-byte packs --------------------------------------------------------------------------------
template <typename T> struct PackOf { static constexpr int N = 16 / sizeof(T); };
template <typename T, int N> struct alignas(sizeof(T) * N) Pack { T v[N]; };

template <typename T, int N> __device__ __forceinline__ Pack<T, N> ld_pack(const T* p) {
  return *reinterpret_cast<const Pack<T, N>*>(p);
}
template <typename T, int N> __device__ __forceinline__ void st_pack(T* p, const Pack<T, N>& v) {
  *reinterpret_cast<Pack<T, N>*>(p) = v;
}

// ---- wave / block reductions (64-wide) -------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); `sh` must hold >= 16 floats; result broadcast.
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = sh[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, sh[i]);
  return t;
}

// ---- activations ------------------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case D2R_ACT_RELU: return fmaxf(x, 0.f);
    case D2R_ACT_TANH: return tanhf(x);
    case D2R_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    case D2R_ACT_QUICK_GELU: return x / (1.f + expf(-1.702f * x));
    case D2R_ACT_TANH_RELU: return fmaxf(tanhf(x), 0.f);
    case D2R_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    default: return x;
  }
}
// derivative given `r` = activation OUTPUT (relu/tanh/tanh_relu/sigmoid) or PRE-activation (gelu/quick_gelu)
__device__ __forceinline__ float act_grad(int act, float r) {
  switch (act) {
    case D2R_ACT_RELU: return r > 0.f ? 1.f : 0.f;
    case D2R_ACT_TANH: return 1.f - r * r;
    case D2R_ACT_GELU: {
      const float cdf = 0.5f * (1.f + erff(r * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * expf(-0.5f * r * r);
      return cdf + r * pdf;
    }
    case D2R_ACT_QUICK_GELU: {
      const float s = 1.f / (1.f + expf(-1.702f * r));
      return s + 1.702f * r * s * (1.f - s);
    }
    case D2R_ACT_TANH_RELU: return r > 0.f ? 1.f - r * r : 0.f;  // r = relu(tanh(x)) > 0  <=>  tanh(x) > 0
    case D2R_ACT_SIGMOID: return r * (1.f - r);
    default: return 1.f;
  }
}

// ---- counter-based dropout mask (shared by d2r_dropout and the fused attention cores) ---------------------------------
// 24-bit uniform from (seed, element index): splitmix64 finaliser.  keep(i) <=> d2r_rand24(seed, i) >= p * 2^24.
__device__ __forceinline__ uint32_t d2r_rand24(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 40);
}
static inline uint32_t d2r_drop_threshold(float p) { return (uint32_t)((double)p * 16777216.0); }

static inline int d2r_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
