// elementwise.hip — 16-byte vectorised elementwise kernels (activation fwd/bwd, squared difference, FiLM
// mul-add, gate lerp, axpby, casts, AdamW).  Grid-stride over 16-B packs with a scalar tail; when a pointer is
// not 16-B aligned the scalar path is used for everything.
#include <stdlib.h>
#include "common.h"

template <int NIN, int NOUT>
struct EwPtrs {
  const void* in[NIN > 0 ? NIN : 1];
  void* out[NOUT];
};

// F::apply(const float (&in)[NIN], float (&out)[NOUT])
template <typename T, int NIN, int NOUT, typename F>
__global__ __launch_bounds__(256) void ew_kernel(EwPtrs<NIN, NOUT> p, int64_t n, int vec_ok, F f) {
  constexpr int VEC = PackOf<T>::N;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t npk = vec_ok ? n / VEC : 0;
  for (int64_t k = tid; k < npk; k += nthreads) {
    Pack<T, VEC> pin[NIN > 0 ? NIN : 1], pout[NOUT];
#pragma unroll
    for (int a = 0; a < NIN; ++a) pin[a] = ld_pack<T, VEC>(reinterpret_cast<const T*>(p.in[a]) + k * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float x[NIN > 0 ? NIN : 1], y[NOUT];
#pragma unroll
      for (int a = 0; a < NIN; ++a) x[a] = to_f<T>(pin[a].v[j]);
      f.apply(x, y);
#pragma unroll
      for (int o = 0; o < NOUT; ++o) pout[o].v[j] = from_f<T>(y[o]);
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) st_pack<T, VEC>(reinterpret_cast<T*>(p.out[o]) + k * VEC, pout[o]);
  }
  for (int64_t e = npk * VEC + tid; e < n; e += nthreads) {
    float x[NIN > 0 ? NIN : 1], y[NOUT];
#pragma unroll
    for (int a = 0; a < NIN; ++a) x[a] = to_f<T>(reinterpret_cast<const T*>(p.in[a])[e]);
    f.apply(x, y);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) reinterpret_cast<T*>(p.out[o])[e] = from_f<T>(y[o]);
  }
}

template <int NIN, int NOUT, typename F>
static int ew_launch(const char* name, int dtype, const EwPtrs<NIN, NOUT>& p, int64_t n, F f, void* stream) {
  if (n < 0) return d2r_fail(D2R_ERR_INVALID, "%s: negative size", name);
  if (n == 0) return D2R_OK;
  int vec_ok = 1;
  for (int a = 0; a < NIN; ++a) {
    if (!p.in[a]) return d2r_fail(D2R_ERR_INVALID, "%s: null input %d", name, a);
    vec_ok &= d2r_aligned16(p.in[a]);
  }
  for (int o = 0; o < NOUT; ++o) {
    if (!p.out[o]) return d2r_fail(D2R_ERR_INVALID, "%s: null output %d", name, o);
    vec_ok &= d2r_aligned16(p.out[o]);
  }
  const int64_t work = n / (dtype != D2R_F32 ? 8 : 4) + 1;
  int blocks = (int)((work + 255) / 256);
  if (blocks > 2048) blocks = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) hipLaunchKernelGGL((ew_kernel<bf16_t, NIN, NOUT, F>), dim3(blocks), dim3(256), 0, st, p, n, vec_ok, f);
  else if (dtype == D2R_F16) hipLaunchKernelGGL((ew_kernel<f16_t, NIN, NOUT, F>), dim3(blocks), dim3(256), 0, st, p, n, vec_ok, f);
  else if (dtype == D2R_F32) hipLaunchKernelGGL((ew_kernel<float, NIN, NOUT, F>), dim3(blocks), dim3(256), 0, st, p, n, vec_ok, f);
  else return d2r_fail(D2R_ERR_INVALID, "%s: bad dtype %d", name, dtype);
  return d2r_check_launch(name);
}

// ---- functors -----------------------------------------------------------------------------------------
struct ActFwdF {
  int act;
  __device__ void apply(const float (&x)[1], float (&y)[1]) const { y[0] = act_apply(act, x[0]); }
};
struct ActBwdF {  // in: dY, ref
  int act;
  __device__ void apply(const float (&x)[2], float (&y)[1]) const { y[0] = x[0] * act_grad(act, x[1]); }
};
struct SqDiffFwdF {
  __device__ void apply(const float (&x)[2], float (&y)[1]) const {
    const float d = x[0] - x[1];
    y[0] = d * d;
  }
};
struct SqDiffBwdF {  // in: a, b, dout -> da, db
  __device__ void apply(const float (&x)[3], float (&y)[2]) const {
    const float g = 2.f * (x[0] - x[1]) * x[2];
    y[0] = g;
    y[1] = -g;
  }
};
struct MulAddFwdF {  // a, s, h -> a*s + h
  __device__ void apply(const float (&x)[3], float (&y)[1]) const { y[0] = x[0] * x[1] + x[2]; }
};
struct MulAddBwdF {  // a, s, dout -> da, ds
  __device__ void apply(const float (&x)[3], float (&y)[2]) const {
    y[0] = x[2] * x[1];
    y[1] = x[2] * x[0];
  }
};
struct LerpFwdF {  // g, a, b -> g*a + (1-g)*b
  __device__ void apply(const float (&x)[3], float (&y)[1]) const { y[0] = x[0] * x[1] + (1.f - x[0]) * x[2]; }
};
struct LerpBwdF {  // g, a, b, dout -> dg, da, db
  __device__ void apply(const float (&x)[4], float (&y)[3]) const {
    y[0] = x[3] * (x[1] - x[2]);
    y[1] = x[3] * x[0];
    y[2] = x[3] * (1.f - x[0]);
  }
};
struct AddF {
  __device__ void apply(const float (&x)[2], float (&y)[1]) const { y[0] = x[0] + x[1]; }
};
struct AxpbyF {  // x, y_old -> alpha*x + beta*y_old
  float alpha, beta;
  __device__ void apply(const float (&x)[2], float (&y)[1]) const { y[0] = alpha * x[0] + (beta != 0.f ? beta * x[1] : 0.f); }
};

extern "C" int d2r_act_fwd(int dtype, int act, const void* X, void* Y, int64_t n, void* stream) {
  EwPtrs<1, 1> p{{X}, {Y}};
  return ew_launch("d2r_act_fwd", dtype, p, n, ActFwdF{act}, stream);
}
extern "C" int d2r_act_bwd(int dtype, int act, const void* dY, const void* ref, void* dX, int64_t n, void* stream) {
  EwPtrs<2, 1> p{{dY, ref}, {dX}};
  return ew_launch("d2r_act_bwd", dtype, p, n, ActBwdF{act}, stream);
}
extern "C" int d2r_sqdiff_fwd(int dtype, const void* a, const void* b, void* out, int64_t n, void* stream) {
  EwPtrs<2, 1> p{{a, b}, {out}};
  return ew_launch("d2r_sqdiff_fwd", dtype, p, n, SqDiffFwdF{}, stream);
}
extern "C" int d2r_sqdiff_bwd(int dtype, const void* a, const void* b, const void* dout, void* da, void* db,
                              int64_t n, void* stream) {
  EwPtrs<3, 2> p{{a, b, dout}, {da, db}};
  return ew_launch("d2r_sqdiff_bwd", dtype, p, n, SqDiffBwdF{}, stream);
}
extern "C" int d2r_muladd_fwd(int dtype, const void* a, const void* s, const void* h, void* out, int64_t n,
                              void* stream) {
  EwPtrs<3, 1> p{{a, s, h}, {out}};
  return ew_launch("d2r_muladd_fwd", dtype, p, n, MulAddFwdF{}, stream);
}
extern "C" int d2r_muladd_bwd(int dtype, const void* a, const void* s, const void* dout, void* da, void* ds,
                              int64_t n, void* stream) {
  EwPtrs<3, 2> p{{a, s, dout}, {da, ds}};
  return ew_launch("d2r_muladd_bwd", dtype, p, n, MulAddBwdF{}, stream);
}
extern "C" int d2r_lerp_fwd(int dtype, const void* g, const void* a, const void* b, void* out, int64_t n,
                            void* stream) {
  EwPtrs<3, 1> p{{g, a, b}, {out}};
  return ew_launch("d2r_lerp_fwd", dtype, p, n, LerpFwdF{}, stream);
}
extern "C" int d2r_lerp_bwd(int dtype, const void* g, const void* a, const void* b, const void* dout, void* dg,
                            void* da, void* db, int64_t n, void* stream) {
  EwPtrs<4, 3> p{{g, a, b, dout}, {dg, da, db}};
  return ew_launch("d2r_lerp_bwd", dtype, p, n, LerpBwdF{}, stream);
}
extern "C" int d2r_axpby(int dtype, float alpha, const void* x, float beta, void* y, int64_t n, void* stream) {
  EwPtrs<2, 1> p{{x, y}, {y}};
  return ew_launch("d2r_axpby", dtype, p, n, AxpbyF{alpha, beta}, stream);
}

// Two independent problems of one size in one launch (the per-sample chains of the routing cells come in text / image or a / b pairs of
// 32 x 768 elements: a launch each is 4-5 us of latency for microseconds of nothing): element for element the same arithmetic as two calls.
struct ActBwd2F {  // in: dY1, ref1, dY2, ref2
  int act;
  __device__ void apply(const float (&x)[4], float (&y)[2]) const {
    y[0] = x[0] * act_grad(act, x[1]);
    y[1] = x[2] * act_grad(act, x[3]);
  }
};
struct Add2F {  // in: a1, b1, a2, b2
  __device__ void apply(const float (&x)[4], float (&y)[2]) const {
    y[0] = x[0] + x[1];
    y[1] = x[2] + x[3];
  }
};
extern "C" int d2r_act_bwd2(int dtype, int act, const void* dY1, const void* ref1, void* dX1, const void* dY2, const void* ref2, void* dX2,
                            int64_t n, void* stream) {
  EwPtrs<4, 2> p{{dY1, ref1, dY2, ref2}, {dX1, dX2}};
  return ew_launch("d2r_act_bwd2", dtype, p, n, ActBwd2F{act}, stream);
}
extern "C" int d2r_add2(int dtype, const void* a1, const void* b1, void* out1, const void* a2, const void* b2, void* out2, int64_t n,
                        void* stream) {
  EwPtrs<4, 2> p{{a1, b1, a2, b2}, {out1, out2}};
  return ew_launch("d2r_add2", dtype, p, n, Add2F{}, stream);
}
extern "C" int d2r_add(int dtype, const void* a, const void* b, void* out, int64_t n, void* stream) {
  EwPtrs<2, 1> p{{a, b}, {out}};
  return ew_launch("d2r_add", dtype, p, n, AddF{}, stream);
}

// ---- dropout ------------------------------------------------------------------------------------------------
// nn.Dropout of the BERT path (models/modeling_unimo.py:330,388,413,468): y = keep(i) ? x / (1 - p) : 0 (+ add).
// keep(i) comes from a counter-based generator (splitmix64 finaliser of seed and element index), so the backward
// pass regenerates the mask from (seed, index) instead of storing it: dx = d2r_dropout(dy) with the same seed.
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, const T* __restrict__ add, T* __restrict__ y,
                                                      int64_t n, uint32_t thresh, float scale, uint64_t seed, int vec_ok) {
  constexpr int VEC = PackOf<T>::N;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t npk = vec_ok ? n / VEC : 0;
  for (int64_t k = tid; k < npk; k += nthreads) {
    const Pack<T, VEC> px = ld_pack<T, VEC>(x + k * VEC);
    Pack<T, VEC> pa, po;
    if (add) pa = ld_pack<T, VEC>(add + k * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float v = d2r_rand24(seed, (uint64_t)(k * VEC + j)) >= thresh ? to_f<T>(px.v[j]) * scale : 0.f;
      if (add) v += to_f<T>(pa.v[j]);
      po.v[j] = from_f<T>(v);
    }
    st_pack<T, VEC>(y + k * VEC, po);
  }
  for (int64_t e = npk * VEC + tid; e < n; e += nthreads) {
    float v = d2r_rand24(seed, (uint64_t)e) >= thresh ? to_f<T>(x[e]) * scale : 0.f;
    if (add) v += to_f<T>(add[e]);
    y[e] = from_f<T>(v);
  }
}

extern "C" int d2r_dropout(int dtype, const void* x, const void* add, void* y, int64_t n, float p, uint64_t seed,
                           void* stream) {
  D2R_REQUIRE(x && y && n >= 0 && p >= 0.f && p < 1.f, "d2r_dropout: bad arguments (0 <= p < 1)");
  if (n == 0) return D2R_OK;
  const uint32_t thresh = d2r_drop_threshold(p);  // drop when the 24-bit uniform is below p * 2^24
  const float scale = 1.f / (1.f - p);
  const int vec_ok = d2r_aligned16(x) && d2r_aligned16(y) && d2r_aligned16(add);
  const int64_t work = n / (dtype != D2R_F32 ? 8 : 4) + 1;
  int blocks = (int)((work + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)add, (bf16_t*)y, n, thresh, scale, seed, vec_ok);
  else if (dtype == D2R_F16) hipLaunchKernelGGL((dropout_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, (const f16_t*)x, (const f16_t*)add, (f16_t*)y, n, thresh, scale, seed, vec_ok);
  else if (dtype == D2R_F32) hipLaunchKernelGGL((dropout_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)x, (const float*)add, (float*)y, n, thresh, scale, seed, vec_ok);
  else return d2r_fail(D2R_ERR_INVALID, "d2r_dropout: bad dtype %d", dtype);
  return d2r_check_launch("d2r_dropout");
}

// out[0] = sum_k coef[k] * x_k[0]  (scalar loss combination: CE + js terms, models/unimo_model.py:160,
// models/modeling_unimo.py:849)
struct LinCombArgs {
  const float* x[8];
  float c[8];
};
__global__ void lincomb_kernel(LinCombArgs a, int n, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < n; ++k) t += a.c[k] * a.x[k][0];
    out[0] = t;
  }
}
extern "C" int d2r_lincomb(const float* const* h_x, const float* h_coef, int n, float* out, void* stream) {
  D2R_REQUIRE(h_x && h_coef && out && n >= 1 && n <= 8, "d2r_lincomb: bad arguments");
  LinCombArgs a;
  for (int k = 0; k < 8; ++k) {
    a.x[k] = k < n ? h_x[k] : nullptr;
    a.c[k] = k < n ? h_coef[k] : 0.f;
    D2R_REQUIRE(k >= n || a.x[k], "d2r_lincomb: null input %d", k);
  }
  hipLaunchKernelGGL(lincomb_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, n, out);
  return d2r_check_launch("d2r_lincomb");
}

// ---- casts ---------------------------------------------------------------------------------------------
template <typename S, typename Dt>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ src, Dt* __restrict__ dst, int64_t n) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n / 4;
  for (int64_t k = tid; k < n4; k += nthreads) {
    Pack<S, 4> a = ld_pack<S, 4>(src + k * 4);
    Pack<Dt, 4> b;
#pragma unroll
    for (int j = 0; j < 4; ++j) b.v[j] = from_f<Dt>(to_f<S>(a.v[j]));
    st_pack<Dt, 4>(dst + k * 4, b);
  }
  for (int64_t e = n4 * 4 + tid; e < n; e += nthreads) dst[e] = from_f<Dt>(to_f<S>(src[e]));
}

extern "C" int d2r_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream) {
  D2R_REQUIRE(src && dst && n >= 0, "d2r_cast: bad arguments");
  D2R_REQUIRE(d2r_aligned16(src) && d2r_aligned16(dst), "d2r_cast: pointers must be 16-byte aligned");
  if (n == 0) return D2R_OK;
  int blocks = (int)((n / 4 + 256) / 256);
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == D2R_F32 && dst_dtype == D2R_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == D2R_F32 && dst_dtype == D2R_F16) hipLaunchKernelGGL((cast_kernel<float, f16_t>), dim3(blocks), dim3(256), 0, st, (const float*)src, (f16_t*)dst, n);
  else if (src_dtype == D2R_BF16 && dst_dtype == D2R_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == D2R_F16 && dst_dtype == D2R_F32) hipLaunchKernelGGL((cast_kernel<f16_t, float>), dim3(blocks), dim3(256), 0, st, (const f16_t*)src, (float*)dst, n);
  else if (src_dtype == D2R_F32 && dst_dtype == D2R_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)src, (float*)dst, n);
  else if (src_dtype == D2R_BF16 && dst_dtype == D2R_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
  else if (src_dtype == D2R_F16 && dst_dtype == D2R_F16) hipLaunchKernelGGL((cast_kernel<f16_t, f16_t>), dim3(blocks), dim3(256), 0, st, (const f16_t*)src, (f16_t*)dst, n);
  else return d2r_fail(D2R_ERR_INVALID, "d2r_cast: bad dtypes %d -> %d", src_dtype, dst_dtype);
  return d2r_check_launch("d2r_cast");
}

// ---- strided row copy: dst[r][0..width) = src[r][0..width) for r < rows (byte pitches) -------------------------------------
// (hipMemcpy2DAsync device-to-device is issued by the runtime as one blit kernel PER ROW for these shapes: 148 launches of
//  3 us per training step for the three gathers / scatters of a routing layer; this is one launch each.)
__global__ __launch_bounds__(256) void copy_rows_kernel(unsigned char* __restrict__ dst, int64_t dpitch, const unsigned char* __restrict__ src,
                                                        int64_t spitch, int64_t width, int64_t rows, int vec) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    const int64_t per_row = width / 16, total = per_row * rows;
    for (int64_t i = tid; i < total; i += nthreads) {
      const int64_t r = i / per_row, c = i - r * per_row;
      *reinterpret_cast<uint4*>(dst + r * dpitch + c * 16) = *reinterpret_cast<const uint4*>(src + r * spitch + c * 16);
    }
  } else {
    const int64_t total = width * rows;
    for (int64_t i = tid; i < total; i += nthreads) {
      const int64_t r = i / width, c = i - r * width;
      dst[r * dpitch + c] = src[r * spitch + c];
    }
  }
}
extern "C" int d2r_copy_rows(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width, int64_t rows, void* stream) {
  D2R_REQUIRE(dst && src && width >= 0 && rows >= 0 && dst_pitch >= width && src_pitch >= width, "d2r_copy_rows: bad arguments");
  if (width == 0 || rows == 0) return D2R_OK;
  const int vec = d2r_aligned16(dst) && d2r_aligned16(src) && dst_pitch % 16 == 0 && src_pitch % 16 == 0 && width % 16 == 0;
  const int64_t work = (vec ? width / 16 : width) * rows;
  int blocks = (int)((work + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char*)dst, dst_pitch, (const unsigned char*)src,
                     src_pitch, width, rows, vec);
  return d2r_check_launch("d2r_copy_rows");
}

// ---- K14 AdamW over a flat fp32 range (modules/train.py:287-322; torch.optim.AdamW semantics) -----------
typedef float f32x4nt __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ Pack<float, 4> ld4(const float* p) {
  if constexpr (NT) {
    const f32x4nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(p));
    Pack<float, 4> r;
    r.v[0] = t.x, r.v[1] = t.y, r.v[2] = t.z, r.v[3] = t.w;
    return r;
  } else {
    return ld_pack<float, 4>(p);
  }
}
template <bool NT>
__device__ __forceinline__ void st4(float* p, const Pack<float, 4>& v) {
  if constexpr (NT) {
    f32x4nt t;
    t.x = v.v[0], t.y = v.v[1], t.z = v.v[2], t.w = v.v[3];
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4nt*>(p));
  } else {
    st_pack<float, 4>(p, v);
  }
}

template <typename H, bool NT>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    H* __restrict__ w16, int64_t n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt, float gscale,
                                                    const float* __restrict__ d_hyper, const int* __restrict__ d_skip) {
  if (d_skip && *d_skip) return;  // overflowed loss-scaled gradients: this step is dropped
  if (d_hyper) {  // hipGraph-safe variant: per-step scalars live in device memory, refreshed before each replay
    lr = d_hyper[0];
    bc1 = d_hyper[1];
    bc2_sqrt = d_hyper[2];
    gscale = d_hyper[3];
  }
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n / 4;
  auto upd = [&](float& wi, float gi, float& mi, float& vi) {
    gi *= gscale;
    wi *= (1.f - lr * wd);                       // decoupled weight decay
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    wi -= (lr / bc1) * (mi / denom);
  };
  for (int64_t k = tid; k < n4; k += nthreads) {
    Pack<float, 4> pw = ld4<NT>(w + k * 4), pg = ld4<NT>(g + k * 4);
    Pack<float, 4> pm = ld4<NT>(m + k * 4), pv = ld4<NT>(v + k * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) upd(pw.v[j], pg.v[j], pm.v[j], pv.v[j]);
    st4<NT>(w + k * 4, pw);
    st4<NT>(m + k * 4, pm);
    st4<NT>(v + k * 4, pv);
    if (w16) {
      Pack<H, 4> ph;
#pragma unroll
      for (int j = 0; j < 4; ++j) ph.v[j] = (H)pw.v[j];
      st_pack<H, 4>(w16 + k * 4, ph);
    }
  }
  for (int64_t e = n4 * 4 + tid; e < n; e += nthreads) {
    float wi = w[e], mi = m[e], vi = v[e];
    upd(wi, g[e], mi, vi);
    w[e] = wi; m[e] = mi; v[e] = vi;
    if (w16) w16[e] = (H)wi;
  }
}

static int g_adamw_nt = 0, g_adamw_blocks = 0;  // A/B (include/d2r_hip_probes.h): non-temporal loads / stores, grid cap
extern "C" void d2r_adamw_probe_mode(int nt, int blocks) { g_adamw_nt = nt, g_adamw_blocks = blocks; }

static int adamw_launch(const char* name, float* w, const float* g, float* m, float* v, void* w16, int w16_dtype, int64_t n, float lr,
                        float b1, float b2, float eps, float wd, float bc1, float bc2s, float gscale, const float* d_hyper,
                        const int* d_skip, void* stream) {
  D2R_REQUIRE(d2r_aligned16(w) && d2r_aligned16(g) && d2r_aligned16(m) && d2r_aligned16(v), "%s: pointers must be 16-byte aligned", name);
  D2R_REQUIRE(!w16 || ((reinterpret_cast<uintptr_t>(w16) & 7u) == 0 && d2r_is16(w16_dtype)),
              "%s: the 16-bit shadow must be 8-byte aligned and D2R_BF16 or D2R_F16 (got dtype %d)", name, w16_dtype);
  if (n == 0) return D2R_OK;
  int blocks = (int)((n / 4 + 256) / 256);
  const int cap = g_adamw_blocks > 0 ? g_adamw_blocks : 2048;
  if (blocks > cap) blocks = cap;
#define D2R_ADAMW_LAUNCH(H, NT) \
  hipLaunchKernelGGL((adamw_kernel<H, NT>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, g, m, v, (H*)w16, n, lr, b1, b2, eps, wd, bc1, bc2s, \
                     gscale, d_hyper, d_skip)
  if (w16 && w16_dtype == D2R_F16) {
    if (g_adamw_nt) D2R_ADAMW_LAUNCH(f16_t, true);
    else D2R_ADAMW_LAUNCH(f16_t, false);
  } else {
    if (g_adamw_nt) D2R_ADAMW_LAUNCH(bf16_t, true);
    else D2R_ADAMW_LAUNCH(bf16_t, false);
  }
#undef D2R_ADAMW_LAUNCH
  return d2r_check_launch(name);
}

extern "C" int d2r_adamw_step(float* w, const float* g, float* m, float* v, void* w16, int w16_dtype, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int64_t step,
                              float grad_scale, const int* d_skip, void* stream) {
  D2R_REQUIRE(w && g && m && v && n >= 0 && step >= 1, "d2r_adamw_step: bad arguments");
  // double on the host, rounded once: FusedAdamW.stage_hyper (the hipGraph path) computes the very same values
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  return adamw_launch("d2r_adamw_step", w, g, m, v, w16, w16_dtype, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale,
                      nullptr, d_skip, stream);
}

// hipGraph-capturable form: d_hyper = device float[4] {lr, 1-beta1^t, sqrt(1-beta2^t), grad_scale}
extern "C" int d2r_adamw_step_dev(float* w, const float* g, float* m, float* v, void* w16, int w16_dtype, int64_t n,
                                  const float* d_hyper, float beta1, float beta2, float eps, float weight_decay,
                                  const int* d_skip, void* stream) {
  D2R_REQUIRE(w && g && m && v && d_hyper && n >= 0, "d2r_adamw_step_dev: bad arguments");
  return adamw_launch("d2r_adamw_step_dev", w, g, m, v, w16, w16_dtype, n, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, 1.f, d_hyper,
                      d_skip, stream);
}

// ---- overflow check of loss-scaled gradients (fp16 compute dtype): one streaming pass, flag |= any(!isfinite(g)) -----
__global__ __launch_bounds__(256) void nonfinite_kernel(const float* __restrict__ g, int64_t n, int* __restrict__ flag) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n / 4;
  bool bad = false;
  for (int64_t k = tid; k < n4; k += nthreads) {
    const Pack<float, 4> p = ld_pack<float, 4>(g + k * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) bad |= !(fabsf(p.v[j]) <= 3.4028234e38f);  // false for inf and for NaN
  }
  for (int64_t e = n4 * 4 + tid; e < n; e += nthreads) bad |= !(fabsf(g[e]) <= 3.4028234e38f);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
extern "C" int d2r_grad_nonfinite(const float* g, int64_t n, int* d_flag, void* stream) {
  D2R_REQUIRE(g && d_flag && n >= 0 && d2r_aligned16(g), "d2r_grad_nonfinite: bad arguments");
  if (n == 0) return D2R_OK;
  int blocks = (int)((n / 4 + 256) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(nonfinite_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, d_flag);
  return d2r_check_launch("d2r_grad_nonfinite");
}
