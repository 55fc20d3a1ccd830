// misc.hip — latency kernels of the path: K6 SAF gate (BatchNorm1d(1)+sigmoid+l1norm), K9 js_div, K13 cross
// entropy, K10 Block merge, K12 embeddings.  All statistics fp32, fixed reduction order.
#include "common.h"

// =====================================================================================================
// K6 SAF gate (models/XModules.py:380-381): one 1024-thread workgroup; a[B,n] fp32 is tiny (B*(Lq+1))
// =====================================================================================================
// Sum and sum of squares of the scores over this rank's samples (fp64): the global-batch-exact mode under data parallelism
// all-reduces them and hands the totals to the gate (gstats), which then normalises with the statistics of the WHOLE batch as the
// reference's BatchNorm1d(1) does on one GPU (models/XModules.py:376,381).
__global__ __launch_bounds__(1024) void saf_gate_stats_kernel(const float* __restrict__ a, int N, double* __restrict__ sums) {
  __shared__ double shd[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  double s = 0.0, q = 0.0;
  for (int i = tid; i < N; i += blockDim.x) {
    const double v = (double)a[i];
    s += v;
    q += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (lane == 0) shd[0][wave] = s, shd[1][wave] = q;
  __syncthreads();
  if (tid == 0) {
    double ts = 0.0, tq = 0.0;
    for (int i = 0; i < nw; ++i) ts += shd[0][i], tq += shd[1][i];
    sums[0] = ts, sums[1] = tq;
  }
}

// 16-bit copy of an fp32 value (same rounding as d2r_cast): the consumers of the gate read it as a GEMM operand
__device__ __forceinline__ void store16(void* p, int dtype, int64_t i, float v) {
  if (dtype == D2R_BF16) reinterpret_cast<bf16_t*>(p)[i] = (bf16_t)v;
  else reinterpret_cast<f16_t*>(p)[i] = (f16_t)v;
}

__global__ __launch_bounds__(1024) void saf_gate_fwd_kernel(const float* __restrict__ a, int B, int n,
                                                            const float* __restrict__ bn_w,
                                                            const float* __restrict__ bn_b, float* running_mean,
                                                            float* running_var, int train, float* __restrict__ w,
                                                            float* __restrict__ saved, const double* __restrict__ gstats,
                                                            double ntotal, void* __restrict__ w16, int w16_dtype) {
  __shared__ float sh[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int N = B * n;
  float mean, var;
  if (train && gstats) {  // statistics of the global batch (all ranks): E[a], E[a^2] - E[a]^2 in fp64
    const double m = gstats[0] / ntotal, v = gstats[1] / ntotal - m * m;
    mean = (float)m, var = (float)(v > 0.0 ? v : 0.0);
    if (tid == 0) {
      running_mean[0] = 0.9f * running_mean[0] + 0.1f * mean;
      running_var[0] = 0.9f * running_var[0] + 0.1f * var * (float)(ntotal / (ntotal > 1.0 ? ntotal - 1.0 : 1.0));
    }
  } else if (train) {
    float s = 0.f;
    for (int i = tid; i < N; i += blockDim.x) s += a[i];
    mean = block_sum(s, sh) / (float)N;
    float q = 0.f;
    for (int i = tid; i < N; i += blockDim.x) {
      const float d = a[i] - mean;
      q += d * d;
    }
    var = block_sum(q, sh) / (float)N;
    if (tid == 0) {  // momentum 0.1, unbiased running variance (torch.nn.BatchNorm1d)
      running_mean[0] = 0.9f * running_mean[0] + 0.1f * mean;
      running_var[0] = 0.9f * running_var[0] + 0.1f * var * ((float)N / (float)(N > 1 ? N - 1 : 1));
    }
  } else {
    mean = running_mean[0];
    var = running_var[0];
  }
  const float rstd = 1.f / sqrtf(var + 1e-5f);
  if (tid == 0) {
    saved[0] = mean;
    saved[1] = rstd;
  }
  const float gw = bn_w[0], gb = bn_b[0];
  for (int b = wave; b < B; b += nw) {
    float s = 0.f;
    for (int k = lane; k < n; k += 64) {
      const float y = (a[b * n + k] - mean) * rstd * gw + gb;
      s += 1.f / (1.f + expf(-y));
    }
    s = wave_sum(s);
    const float inv = 1.f / (s + 1e-8f);
    for (int k = lane; k < n; k += 64) {
      const float y = (a[b * n + k] - mean) * rstd * gw + gb;
      const float wv = inv / (1.f + expf(-y));
      w[b * n + k] = wv;
      if (w16) store16(w16, w16_dtype, b * n + k, wv);
    }
  }
}

__global__ __launch_bounds__(1024) void saf_gate_bwd_kernel(const float* __restrict__ a, const float* __restrict__ dw,
                                                            int B, int n, const float* __restrict__ bn_w,
                                                            const float* __restrict__ bn_b,
                                                            const float* __restrict__ saved, int train,
                                                            float* __restrict__ da, float* __restrict__ d_bn_w,
                                                            float* __restrict__ d_bn_b, int phase, double* __restrict__ gsums,
                                                            double ntotal, void* __restrict__ da16, int da16_dtype, int accumulate) {
  // phase 0: the whole backward with this rank's own sums (local batch statistics).  Global-batch-exact mode: phase 1 stops behind
  // stage 1 (d y in `da`, the LOCAL sums = this rank's share of the BatchNorm parameter gradients, and the fp64 sums for the
  // all-reduce in gsums); phase 2 finishes with the all-reduced sums over the ntotal scores of the global batch.
  __shared__ float sh[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int N = B * n;
  const float mean = saved[0], rstd = saved[1], gw = bn_w[0], gb = bn_b[0];
  if (phase == 2) {
    const float m1 = (float)(gsums[0] / ntotal), m2 = (float)(gsums[1] / ntotal);
    for (int i = tid; i < N; i += blockDim.x) {
      const float xh = (a[i] - mean) * rstd;
      const float v = gw * rstd * (da[i] - m1 - xh * m2);
      da[i] = v;
      if (da16) store16(da16, da16_dtype, i, v);
    }
    return;
  }
  float sdy = 0.f, sdyx = 0.f;
  for (int b = wave; b < B; b += nw) {
    float s = 0.f;
    for (int k = lane; k < n; k += 64) {
      const float y = (a[b * n + k] - mean) * rstd * gw + gb;
      s += 1.f / (1.f + expf(-y));
    }
    s = wave_sum(s);
    const float inv = 1.f / (s + 1e-8f);
    float dot = 0.f;  // sum_k dw * w
    for (int k = lane; k < n; k += 64) {
      const float y = (a[b * n + k] - mean) * rstd * gw + gb;
      dot += dw[b * n + k] * inv / (1.f + expf(-y));
    }
    dot = wave_sum(dot);
    for (int k = lane; k < n; k += 64) {
      const float xh = (a[b * n + k] - mean) * rstd;
      const float sg = 1.f / (1.f + expf(-(xh * gw + gb)));
      const float dsg = (dw[b * n + k] - dot) * inv;
      const float dy = dsg * sg * (1.f - sg);
      da[b * n + k] = dy;  // stage 1: holds dy
      sdy += dy;
      sdyx += dy * xh;
    }
  }
  const float tdy = block_sum(sdy, sh);
  const float tdyx = block_sum(sdyx, sh);
  if (tid == 0) {  // (accumulate: straight into the caller's fp32 gradient sinks)
    d_bn_w[0] = accumulate ? d_bn_w[0] + tdyx : tdyx;
    d_bn_b[0] = accumulate ? d_bn_b[0] + tdy : tdy;
    if (phase == 1) gsums[0] = (double)tdy, gsums[1] = (double)tdyx;
  }
  if (phase == 1) return;
  __syncthreads();
  const float m1 = tdy / (float)N, m2 = tdyx / (float)N;
  for (int i = tid; i < N; i += blockDim.x) {
    const float dy = da[i];
    float v;
    if (train) {
      const float xh = (a[i] - mean) * rstd;
      v = gw * rstd * (dy - m1 - xh * m2);
    } else {
      v = gw * rstd * dy;
    }
    da[i] = v;
    if (da16) store16(da16, da16_dtype, i, v);
  }
}

extern "C" int d2r_saf_gate_fwd(const float* a, int B, int n, const float* bn_weight, const float* bn_bias,
                                float* running_mean, float* running_var, int train, float* w, float* saved,
                                void* stream) {
  return d2r_saf_gate_fwd_ex(a, B, n, bn_weight, bn_bias, running_mean, running_var, train, w, saved, nullptr, 0.0, nullptr, 0, stream);
}

extern "C" int d2r_saf_gate_stats(const float* a, int B, int n, double* sums, void* stream) {
  D2R_REQUIRE(a && sums && B >= 1 && n >= 1, "d2r_saf_gate_stats: bad argument");
  hipLaunchKernelGGL(saf_gate_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, B * n, sums);
  return d2r_check_launch("d2r_saf_gate_stats");
}

extern "C" int d2r_saf_gate_fwd_ex(const float* a, int B, int n, const float* bn_weight, const float* bn_bias, float* running_mean,
                                   float* running_var, int train, float* w, float* saved, const double* gstats, double ntotal,
                                   void* w16, int w16_dtype, void* stream) {
  D2R_REQUIRE(!w16 || d2r_is16(w16_dtype), "d2r_saf_gate_fwd_ex: the copy of w is bf16 or fp16");
  D2R_REQUIRE(a && bn_weight && bn_bias && running_mean && running_var && w && saved, "d2r_saf_gate_fwd: null pointer");
  D2R_REQUIRE(B >= 1 && n >= 1, "d2r_saf_gate_fwd: bad shape");
  D2R_REQUIRE(!gstats || ntotal >= (double)B * n, "d2r_saf_gate_fwd: the global element count is smaller than this rank's");
  hipLaunchKernelGGL(saf_gate_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, B, n, bn_weight, bn_bias,
                     running_mean, running_var, train, w, saved, gstats, ntotal, w16, w16_dtype);
  return d2r_check_launch("d2r_saf_gate_fwd");
}

extern "C" int d2r_saf_gate_bwd(const float* a, const float* dw, int B, int n, const float* bn_weight,
                                const float* bn_bias, const float* saved, int train, float* da, float* d_bn_weight,
                                float* d_bn_bias, void* stream) {
  return d2r_saf_gate_bwd_ex(a, dw, B, n, bn_weight, bn_bias, saved, train, da, d_bn_weight, d_bn_bias, 0, nullptr, 0.0, nullptr, 0, 0, stream);
}

extern "C" int d2r_saf_gate_bwd_ex(const float* a, const float* dw, int B, int n, const float* bn_weight, const float* bn_bias,
                                   const float* saved, int train, float* da, float* d_bn_weight, float* d_bn_bias, int phase,
                                   double* gsums, double ntotal, void* da16, int da16_dtype, int accumulate, void* stream) {
  D2R_REQUIRE(!da16 || d2r_is16(da16_dtype), "d2r_saf_gate_bwd_ex: the copy of da is bf16 or fp16");
  D2R_REQUIRE(a && bn_weight && bn_bias && saved && da && (phase == 2 || (dw && d_bn_weight && d_bn_bias)), "d2r_saf_gate_bwd: null pointer");
  D2R_REQUIRE(B >= 1 && n >= 1, "d2r_saf_gate_bwd: bad shape");
  D2R_REQUIRE(phase == 0 || (phase >= 1 && phase <= 2 && gsums && train && ntotal >= (double)B * n), "d2r_saf_gate_bwd: bad phase / global sums");
  hipLaunchKernelGGL(saf_gate_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, dw, B, n, bn_weight, bn_bias,
                     saved, train, da, d_bn_weight, d_bn_bias, phase, gsums, ntotal, da16, da16_dtype, accumulate);
  return d2r_check_launch("d2r_saf_gate_bwd");
}

// ---- the two rank-one products around the gate in the backward pass of the SAF-weighted sum wsum[b] = w[b] @ S[b] (S [B,n,E]) ----
// As GEMMs (a [1,E] x [E,n] product per sample; an outer product with a reduction of ONE) they took 19-47 us each on the tiled kernel.
//   d2r_saf_dweights: dw[b,i] = <dwsum[b,:], S[b,i,:]>                      fp32 out, a wave per row, fixed reduction order
//   d2r_saf_dscores:  dS[b,i,:] = w[b,i] * dwsum[b,:] + da[b,i] * w_saf[:]  one pass, rounded once
template <typename T>
__global__ __launch_bounds__(256) void saf_dweights_kernel(const T* __restrict__ dwsum, const T* __restrict__ S, int B, int n, int E,
                                                           float* __restrict__ dw) {
  constexpr int VEC = PackOf<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * n) return;
  const int b = (int)(row / n);
  const T* s = S + row * E;
  const T* d = dwsum + (int64_t)b * E;
  float acc = 0.f;
  for (int pk = lane; pk * VEC < E; pk += 64) {
    const Pack<T, VEC> sv = ld_pack<T, VEC>(s + pk * VEC), dv = ld_pack<T, VEC>(d + pk * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc += to_f<T>(sv.v[j]) * to_f<T>(dv.v[j]);
  }
  acc = wave_sum(acc);
  if (lane == 0) dw[row] = acc;
}
template <typename T>
__global__ __launch_bounds__(256) void saf_dscores_kernel(const T* __restrict__ w, const T* __restrict__ dwsum, const float* __restrict__ da,
                                                          const T* __restrict__ w_saf, int B, int n, int E, T* __restrict__ dS) {
  constexpr int VEC = PackOf<T>::N;
  const int npk = E / VEC;
  const int64_t total = (int64_t)B * n * npk;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t row = e / npk;
    const int pk = (int)(e - row * npk), b = (int)(row / n);
    const float wi = to_f<T>(w[row]), dai = da[row];
    const Pack<T, VEC> dv = ld_pack<T, VEC>(dwsum + (int64_t)b * E + pk * VEC), sv = ld_pack<T, VEC>(w_saf + pk * VEC);
    Pack<T, VEC> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(wi * to_f<T>(dv.v[j]) + dai * to_f<T>(sv.v[j]));
    st_pack<T, VEC>(dS + row * E + pk * VEC, o);
  }
}

extern "C" int d2r_saf_dweights(int dtype, const void* dwsum, const void* S, int B, int n, int E, float* dw, void* stream) {
  D2R_REQUIRE(dwsum && S && dw && B >= 1 && n >= 1 && E >= 8 && E % 8 == 0, "d2r_saf_dweights: bad argument");
  D2R_REQUIRE(d2r_is16(dtype) && d2r_aligned16(dwsum) && d2r_aligned16(S), "d2r_saf_dweights: 16-bit, 16-byte aligned operands");
  const dim3 grid((unsigned)(((int64_t)B * n + 3) / 4));
  if (dtype == D2R_BF16) hipLaunchKernelGGL(saf_dweights_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dwsum, (const bf16_t*)S, B, n, E, dw);
  else hipLaunchKernelGGL(saf_dweights_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const f16_t*)dwsum, (const f16_t*)S, B, n, E, dw);
  return d2r_check_launch("d2r_saf_dweights");
}
extern "C" int d2r_saf_dscores(int dtype, const void* w, const void* dwsum, const float* da, const void* w_saf, int B, int n, int E, void* dS,
                               void* stream) {
  D2R_REQUIRE(w && dwsum && da && w_saf && dS && B >= 1 && n >= 1 && E >= 8 && E % 8 == 0, "d2r_saf_dscores: bad argument");
  D2R_REQUIRE(d2r_is16(dtype) && d2r_aligned16(dwsum) && d2r_aligned16(w_saf) && d2r_aligned16(dS), "d2r_saf_dscores: 16-bit, 16-byte aligned operands");
  const int64_t total = (int64_t)B * n * (E / 8);
  const dim3 grid((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256));
  if (dtype == D2R_BF16)
    hipLaunchKernelGGL(saf_dscores_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (const bf16_t*)dwsum, da, (const bf16_t*)w_saf, B, n, E, (bf16_t*)dS);
  else
    hipLaunchKernelGGL(saf_dscores_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const f16_t*)w, (const f16_t*)dwsum, da, (const f16_t*)w_saf, B, n, E, (f16_t*)dS);
  return d2r_check_launch("d2r_saf_dscores");
}

// =====================================================================================================
// K9 js_div (models/XModules.py:32-41) on [B,B] logits; one workgroup, wave per row
// =====================================================================================================
__device__ __forceinline__ void row_logsoftmax_stats(const float* x, int n, int lane, float& mx, float& lse) {
  float m = -INFINITY;
  for (int k = lane; k < n; k += 64) m = fmaxf(m, x[k]);
  m = wave_max(m);
  float s = 0.f;
  for (int k = lane; k < n; k += 64) s += expf(x[k] - m);
  s = wave_sum(s);
  mx = m;
  lse = logf(s);
}

__global__ __launch_bounds__(1024) void jsdiv_fwd_kernel(const float* __restrict__ P, const float* __restrict__ Q, int B,
                                                         float* __restrict__ out) {
  __shared__ float sh[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float acc = 0.f;
  for (int r = wave; r < B; r += nw) {
    const float* p = P + (int64_t)r * B;
    const float* q = Q + (int64_t)r * B;
    float mp, lp, mq, lq;
    row_logsoftmax_stats(p, B, lane, mp, lp);
    row_logsoftmax_stats(q, B, lane, mq, lq);
    for (int k = lane; k < B; k += 64) {
      const float logp = p[k] - mp - lp, logq = q[k] - mq - lq;
      const float pp = expf(logp), qq = expf(logq);
      const float logm = logf(0.5f * (pp + qq));
      acc += (pp > 0.f ? pp * (logp - logm) : 0.f) + (qq > 0.f ? qq * (logq - logm) : 0.f);
    }
  }
  const float t = block_sum(acc, sh);
  if (threadIdx.x == 0) out[0] = 0.5f * t / (float)B;
}

// dP_ij = dout/(2B) * p_ij * (gp_ij - sum_k gp_ik p_ik),  gp = log p - log m   (same for Q)
__global__ __launch_bounds__(1024) void jsdiv_bwd_kernel(const float* __restrict__ P, const float* __restrict__ Q, int B,
                                                         const float* __restrict__ dout, float* __restrict__ dP,
                                                         float* __restrict__ dQ) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float sc = dout[0] * 0.5f / (float)B;
  for (int r = wave; r < B; r += nw) {
    const float* p = P + (int64_t)r * B;
    const float* q = Q + (int64_t)r * B;
    float mp, lp, mq, lq;
    row_logsoftmax_stats(p, B, lane, mp, lp);
    row_logsoftmax_stats(q, B, lane, mq, lq);
    float dp = 0.f, dq = 0.f;
    for (int k = lane; k < B; k += 64) {
      const float logp = p[k] - mp - lp, logq = q[k] - mq - lq;
      const float pp = expf(logp), qq = expf(logq);
      const float logm = logf(0.5f * (pp + qq));
      dp += pp > 0.f ? pp * (logp - logm) : 0.f;
      dq += qq > 0.f ? qq * (logq - logm) : 0.f;
    }
    dp = wave_sum(dp);
    dq = wave_sum(dq);
    for (int k = lane; k < B; k += 64) {
      const float logp = p[k] - mp - lp, logq = q[k] - mq - lq;
      const float pp = expf(logp), qq = expf(logq);
      const float logm = logf(0.5f * (pp + qq));
      dP[(int64_t)r * B + k] = pp > 0.f ? sc * pp * ((logp - logm) - dp) : 0.f;
      dQ[(int64_t)r * B + k] = qq > 0.f ? sc * qq * ((logq - logm) - dq) : 0.f;
    }
  }
}

extern "C" int d2r_jsdiv_fwd(const float* p_logits, const float* q_logits, int B, float* out, void* stream) {
  D2R_REQUIRE(p_logits && q_logits && out && B >= 1, "d2r_jsdiv_fwd: bad arguments");
  hipLaunchKernelGGL(jsdiv_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p_logits, q_logits, B, out);
  return d2r_check_launch("d2r_jsdiv_fwd");
}
extern "C" int d2r_jsdiv_bwd(const float* p_logits, const float* q_logits, int B, const float* dout, float* dp,
                             float* dq, void* stream) {
  D2R_REQUIRE(p_logits && q_logits && dout && dp && dq && B >= 1, "d2r_jsdiv_bwd: bad arguments");
  hipLaunchKernelGGL(jsdiv_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p_logits, q_logits, B, dout, dp, dq);
  return d2r_check_launch("d2r_jsdiv_bwd");
}

// =====================================================================================================
// K13 cross entropy, mean over the batch (models/unimo_model.py:147,160)
// =====================================================================================================
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     int B, int C, float* __restrict__ loss) {
  __shared__ float sh[16];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float* x = logits + (int64_t)b * C;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, x[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(x[c] - m);
    acc += m + logf(s) - x[labels[b]];
  }
  const float t = block_sum(acc, sh);
  if (threadIdx.x == 0) loss[0] = t / (float)B;
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     int B, int C, const float* __restrict__ dloss,
                                                     float* __restrict__ dlogits) {
  const float sc = dloss[0] / (float)B;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    const float* x = logits + (int64_t)b * C;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, x[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(x[c] - m);
    const float inv = 1.f / s;
    const int64_t y = labels[b];
    for (int c = 0; c < C; ++c) dlogits[(int64_t)b * C + c] = sc * (expf(x[c] - m) * inv - (c == y ? 1.f : 0.f));
  }
}
extern "C" int d2r_ce_fwd(const float* logits, const int64_t* labels, int B, int C, float* loss, void* stream) {
  D2R_REQUIRE(logits && labels && loss && B >= 1 && C >= 1, "d2r_ce_fwd: bad arguments");
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, B, C, loss);
  return d2r_check_launch("d2r_ce_fwd");
}
extern "C" int d2r_ce_bwd(const float* logits, const int64_t* labels, int B, int C, const float* dloss,
                          float* dlogits, void* stream) {
  D2R_REQUIRE(logits && labels && dloss && dlogits && B >= 1 && C >= 1, "d2r_ce_bwd: bad arguments");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(d2r_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, logits, labels, B, C, dloss, dlogits);
  return d2r_check_launch("d2r_ce_bwd");
}

// =====================================================================================================
// K10 Block merge (models/XModules.py:541-549): wave per (sample, chunk)
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void block_merge_fwd_kernel(const T* __restrict__ m0, const T* __restrict__ m1, int B,
                                                              int C, int R, int S, T* __restrict__ out,
                                                              float* __restrict__ zraw) {
  const int lane = threadIdx.x & 63;
  const int64_t bc = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bc >= (int64_t)B * C) return;
  const T* a = m0 + bc * R * S;
  const T* b = m1 + bc * R * S;
  float q = 0.f;
  for (int s = lane; s < S; s += 64) {
    float z = 0.f;
    for (int r = 0; r < R; ++r) z += to_f<T>(a[r * S + s]) * to_f<T>(b[r * S + s]);
    zraw[bc * S + s] = z;
    q += fabsf(z);  // (sign(z) sqrt|z|)^2 = |z|
  }
  const float nrm = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
  for (int s = lane; s < S; s += 64) {
    const float z = zraw[bc * S + s];
    const float y = z > 0.f ? sqrtf(z) : (z < 0.f ? -sqrtf(-z) : 0.f);
    out[bc * S + s] = from_f<T>(y / nrm);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void block_merge_bwd_kernel(const T* __restrict__ m0, const T* __restrict__ m1,
                                                              const float* __restrict__ zraw,
                                                              const T* __restrict__ dout, int B, int C, int R, int S,
                                                              T* __restrict__ dm0, T* __restrict__ dm1) {
  const int lane = threadIdx.x & 63;
  const int64_t bc = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bc >= (int64_t)B * C) return;
  float q = 0.f;
  for (int s = lane; s < S; s += 64) q += fabsf(zraw[bc * S + s]);
  const float n2 = sqrtf(wave_sum(q));
  const float nrm = fmaxf(n2, 1e-12f);
  float dot = 0.f;  // sum_s yhat * dout
  for (int s = lane; s < S; s += 64) {
    const float z = zraw[bc * S + s];
    const float y = z > 0.f ? sqrtf(z) : (z < 0.f ? -sqrtf(-z) : 0.f);
    dot += (y / nrm) * to_f<T>(dout[bc * S + s]);
  }
  dot = wave_sum(dot);
  const T* a = m0 + bc * R * S;
  const T* b = m1 + bc * R * S;
  for (int s = lane; s < S; s += 64) {
    const float z = zraw[bc * S + s];
    const float y = z > 0.f ? sqrtf(z) : (z < 0.f ? -sqrtf(-z) : 0.f);
    const float d = to_f<T>(dout[bc * S + s]);
    // F.normalize: y / max(||y||, eps); the clamp branch (||y|| < eps) has no norm-gradient term
    const float dy = n2 > 1e-12f ? (d - (y / nrm) * dot) / nrm : d / nrm;
    const float dz = z != 0.f ? dy * 0.5f / sqrtf(fabsf(z)) : 0.f;
    for (int r = 0; r < R; ++r) {
      dm0[bc * R * S + r * S + s] = from_f<T>(dz * to_f<T>(b[r * S + s]));
      dm1[bc * R * S + r * S + s] = from_f<T>(dz * to_f<T>(a[r * S + s]));
    }
  }
}

extern "C" int d2r_block_merge_fwd(int dtype, const void* m0, const void* m1, int B, int C, int R, int S, void* out,
                                   float* zraw, void* stream) {
  D2R_REQUIRE(m0 && m1 && out && zraw && B >= 1 && C >= 1 && R >= 1 && S >= 1, "d2r_block_merge_fwd: bad arguments");
  dim3 grid(d2r_cdiv((int64_t)B * C, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) hipLaunchKernelGGL((block_merge_fwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)m0, (const bf16_t*)m1, B, C, R, S, (bf16_t*)out, zraw);
  else if (dtype == D2R_F16) hipLaunchKernelGGL((block_merge_fwd_kernel<f16_t>), grid, block, 0, st, (const f16_t*)m0, (const f16_t*)m1, B, C, R, S, (f16_t*)out, zraw);
  else if (dtype == D2R_F32) hipLaunchKernelGGL((block_merge_fwd_kernel<float>), grid, block, 0, st, (const float*)m0, (const float*)m1, B, C, R, S, (float*)out, zraw);
  else return d2r_fail(D2R_ERR_INVALID, "d2r_block_merge_fwd: bad dtype %d", dtype);
  return d2r_check_launch("d2r_block_merge_fwd");
}
extern "C" int d2r_block_merge_bwd(int dtype, const void* m0, const void* m1, const float* zraw, const void* dout,
                                   int B, int C, int R, int S, void* dm0, void* dm1, void* stream) {
  D2R_REQUIRE(m0 && m1 && zraw && dout && dm0 && dm1 && B >= 1 && C >= 1 && R >= 1 && S >= 1, "d2r_block_merge_bwd: bad arguments");
  dim3 grid(d2r_cdiv((int64_t)B * C, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) hipLaunchKernelGGL((block_merge_bwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)m0, (const bf16_t*)m1, zraw, (const bf16_t*)dout, B, C, R, S, (bf16_t*)dm0, (bf16_t*)dm1);
  else if (dtype == D2R_F16) hipLaunchKernelGGL((block_merge_bwd_kernel<f16_t>), grid, block, 0, st, (const f16_t*)m0, (const f16_t*)m1, zraw, (const f16_t*)dout, B, C, R, S, (f16_t*)dm0, (f16_t*)dm1);
  else if (dtype == D2R_F32) hipLaunchKernelGGL((block_merge_bwd_kernel<float>), grid, block, 0, st, (const float*)m0, (const float*)m1, zraw, (const float*)dout, B, C, R, S, (float*)dm0, (float*)dm1);
  else return d2r_fail(D2R_ERR_INVALID, "d2r_block_merge_bwd: bad dtype %d", dtype);
  return d2r_check_launch("d2r_block_merge_bwd");
}

// =====================================================================================================
// K12 embeddings
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ tt,
                                                             const float* __restrict__ word, const float* __restrict__ pos,
                                                             const float* __restrict__ type, int B, int L, int D,
                                                             T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= (int64_t)B * L) return;
  const int l = (int)(tok % L);
  const float* w = word + ids[tok] * D;
  const float* t = type + tt[tok] * D;
  const float* p = pos + (int64_t)l * D;
  for (int c = lane * 4; c < D; c += 256) {
    const Pack<float, 4> a = ld_pack<float, 4>(w + c), b = ld_pack<float, 4>(t + c), d = ld_pack<float, 4>(p + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[tok * D + c + j] = from_f<T>((a.v[j] + b.v[j]) + d.v[j]);
  }
}

// Deterministic gradients of the three tables (no float atomics: run-to-run bit-identical, so data-parallel replicas
// and hipGraph replays agree exactly).
// word table: the FIRST token carrying an id (its "leader") sums the dY rows of every token with that id in token
// order and adds the result to the table row; all other tokens with that id do nothing.
template <typename T, bool IDS_IN_LDS>
__global__ __launch_bounds__(256) void bert_embed_bwd_word_kernel(const T* __restrict__ dY, const int64_t* __restrict__ ids,
                                                                  int64_t ntok, int D, int64_t pad_id,
                                                                  float* __restrict__ dword) {
  // the scans below touch every id: keep them in LDS (as int32) when they fit, one coalesced pass per workgroup
  constexpr int LDS_IDS = IDS_IN_LDS ? 8192 : 1;
  __shared__ int sid[LDS_IDS];
  if constexpr (IDS_IN_LDS) {
    for (int64_t j = threadIdx.x; j < ntok; j += 256) sid[j] = (int)ids[j];
    __syncthreads();
  }
  auto id_at = [&](int64_t j) -> int64_t { return IDS_IN_LDS ? (int64_t)sid[j] : ids[j]; };
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= ntok) return;
  const int64_t id = id_at(tok);
  if (id == pad_id) return;
  for (int64_t j0 = 0; j0 < tok; j0 += 64) {  // leader test (wave-uniform exit)
    const int64_t j = j0 + lane;
    if (__ballot(j < tok && id_at(j) == id)) return;
  }
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
  for (int64_t j0 = tok & ~(int64_t)63; j0 < ntok; j0 += 64) {
    const int64_t j = j0 + lane;
    uint64_t hits = __ballot(j >= tok && j < ntok && id_at(j) == id);
    while (hits) {
      const int b = __builtin_ctzll(hits);
      hits &= hits - 1;
      const T* row = dY + (j0 + b) * D;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < D) {
          const Pack<T, 4> v = ld_pack<T, 4>(row + c);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[i][k] += to_f<T>(v.v[k]);
        }
      }
    }
  }
  float* dst = dword + id * D;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < D) {
      Pack<float, 4> o = ld_pack<float, 4>(dst + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) o.v[k] += acc[i][k];
      st_pack<float, 4>(dst + c, o);
    }
  }
}
// type table: grid (cdiv(D,64), ntype); wave w of 16 sums tokens w, w+16, ... of its type in order, the 16 partials
// are combined in wave order.  Loads are unconditional (masked after the fact) so that 8 of them are in flight.
template <typename T>
__global__ __launch_bounds__(1024) void bert_embed_bwd_type_kernel(const T* __restrict__ dY, const int64_t* __restrict__ tt,
                                                                   int64_t ntok, int D, float* __restrict__ dtype_tab) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t ty = blockIdx.y;
  float s = 0.f;
  if (c < D) {
    int64_t t = w;
    for (; t + 16 * 7 < ntok; t += 16 * 8) {
      float v[8];
      bool m[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = to_f<T>(dY[(t + 16 * u) * D + c]);
        m[u] = tt[t + 16 * u] == ty;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += m[u] ? v[u] : 0.f;
    }
    for (; t < ntok; t += 16)
      if (tt[t] == ty) s += to_f<T>(dY[t * D + c]);
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < D) {
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += part[i][lane];
    dtype_tab[ty * D + c] += r;
  }
}
template <typename T, bool ACC>
__global__ __launch_bounds__(256) void batch_sum_rows_kernel(const T* __restrict__ dY, int B, int L, int D,
                                                             float* __restrict__ dst) {
  // dst[l, c] (+)= sum_b dY[b, l, c]
  const int l = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += to_f<T>(dY[((int64_t)b * L + l) * D + c]);
    dst[(int64_t)l * D + c] = ACC ? dst[(int64_t)l * D + c] + s : s;
  }
}

extern "C" int d2r_bert_embed_fwd(int dtype, const int64_t* ids, const int64_t* tt, const float* word, const float* pos,
                                  const float* type, int B, int L, int D, int vocab, int ntype, void* out, void* stream) {
  D2R_REQUIRE(ids && tt && word && pos && type && out, "d2r_bert_embed_fwd: null pointer");
  D2R_REQUIRE(B >= 1 && L >= 1 && D % 4 == 0, "d2r_bert_embed_fwd: bad shape");
  D2R_REQUIRE(d2r_aligned16(word) && d2r_aligned16(pos) && d2r_aligned16(type), "d2r_bert_embed_fwd: tables must be 16-byte aligned");
  (void)vocab; (void)ntype;
  dim3 grid(d2r_cdiv((int64_t)B * L, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) hipLaunchKernelGGL((bert_embed_fwd_kernel<bf16_t>), grid, block, 0, st, ids, tt, word, pos, type, B, L, D, (bf16_t*)out);
  else if (dtype == D2R_F16) hipLaunchKernelGGL((bert_embed_fwd_kernel<f16_t>), grid, block, 0, st, ids, tt, word, pos, type, B, L, D, (f16_t*)out);
  else if (dtype == D2R_F32) hipLaunchKernelGGL((bert_embed_fwd_kernel<float>), grid, block, 0, st, ids, tt, word, pos, type, B, L, D, (float*)out);
  else return d2r_fail(D2R_ERR_INVALID, "d2r_bert_embed_fwd: bad dtype %d", dtype);
  return d2r_check_launch("d2r_bert_embed_fwd");
}

template <typename T>
static void launch_bert_embed_bwd(const void* dY, const int64_t* ids, const int64_t* tt, int B, int L, int D, int ntype,
                                  int64_t pad_id, float* dword, float* dpos, float* dtype_tab, hipStream_t st) {
  const int64_t ntok = (int64_t)B * L;
  if (ntok <= 8192) hipLaunchKernelGGL((bert_embed_bwd_word_kernel<T, true>), dim3(d2r_cdiv(ntok, 4)), dim3(256), 0, st, (const T*)dY, ids, ntok, D, pad_id, dword);
  else hipLaunchKernelGGL((bert_embed_bwd_word_kernel<T, false>), dim3(d2r_cdiv(ntok, 4)), dim3(256), 0, st, (const T*)dY, ids, ntok, D, pad_id, dword);
  hipLaunchKernelGGL((bert_embed_bwd_type_kernel<T>), dim3(d2r_cdiv(D, 64), ntype), dim3(1024), 0, st, (const T*)dY, tt, ntok, D, dtype_tab);
  hipLaunchKernelGGL((batch_sum_rows_kernel<T, true>), dim3(L), dim3(256), 0, st, (const T*)dY, B, L, D, dpos);
}

extern "C" int d2r_bert_embed_bwd(int dtype, const void* dY, const int64_t* ids, const int64_t* tt, int B, int L, int D,
                                  int ntype, int64_t pad_id, float* dword, float* dpos, float* dtype_tab, void* stream) {
  D2R_REQUIRE(dY && ids && tt && dword && dpos && dtype_tab, "d2r_bert_embed_bwd: null pointer");
  D2R_REQUIRE(B >= 1 && L >= 1 && D >= 4 && D % 4 == 0 && D <= 1024 && ntype >= 1, "d2r_bert_embed_bwd: bad shape (D % 4 == 0, D <= 1024)");
  D2R_REQUIRE(d2r_aligned16(dword) && (reinterpret_cast<uintptr_t>(dY) & 7u) == 0, "d2r_bert_embed_bwd: dword must be 16-byte, dY 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) launch_bert_embed_bwd<bf16_t>(dY, ids, tt, B, L, D, ntype, pad_id, dword, dpos, dtype_tab, st);
  else if (dtype == D2R_F16) launch_bert_embed_bwd<f16_t>(dY, ids, tt, B, L, D, ntype, pad_id, dword, dpos, dtype_tab, st);
  else if (dtype == D2R_F32) launch_bert_embed_bwd<float>(dY, ids, tt, B, L, D, ntype, pad_id, dword, dpos, dtype_tab, st);
  else return d2r_fail(D2R_ERR_INVALID, "d2r_bert_embed_bwd: bad dtype %d", dtype);
  return d2r_check_launch("d2r_bert_embed_bwd");
}

// patches[(b*gh+py)*gw+px, (c*p+i)*p+j] = pixels[b, c, py*p+i, px*p+j]  (matches conv weight [E,3,p,p].view(E,-1))
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ px, int B, int H, int W, int p,
                                                       T* __restrict__ out) {
  const int gh = H / p, gw = W / p, K = 3 * p * p;
  const int64_t total = (int64_t)B * gh * gw * K;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int k = (int)(idx % K);
    const int64_t t = idx / K;
    const int pxi = (int)(t % gw), pyi = (int)((t / gw) % gh), b = (int)(t / ((int64_t)gw * gh));
    const int j = k % p, i = (k / p) % p, c = k / (p * p);
    out[idx] = from_f<T>(px[(((int64_t)b * 3 + c) * H + pyi * p + i) * W + pxi * p + j]);
  }
}
extern "C" int d2r_patchify(int dtype, const float* pixels, int B, int H, int W, int p, void* patches, void* stream) {
  D2R_REQUIRE(pixels && patches && B >= 1 && p >= 1 && H % p == 0 && W % p == 0, "d2r_patchify: bad arguments");
  const int64_t total = (int64_t)B * (H / p) * (W / p) * 3 * p * p;
  int blocks = d2r_cdiv(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) hipLaunchKernelGGL((patchify_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, pixels, B, H, W, p, (bf16_t*)patches);
  else if (dtype == D2R_F16) hipLaunchKernelGGL((patchify_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, pixels, B, H, W, p, (f16_t*)patches);
  else if (dtype == D2R_F32) hipLaunchKernelGGL((patchify_kernel<float>), dim3(blocks), dim3(256), 0, st, pixels, B, H, W, p, (float*)patches);
  else return d2r_fail(D2R_ERR_INVALID, "d2r_patchify: bad dtype %d", dtype);
  return d2r_check_launch("d2r_patchify");
}

template <typename T>
__global__ __launch_bounds__(256) void clip_embed_finish_kernel(T* __restrict__ x, const float* __restrict__ cls,
                                                                const float* __restrict__ pos, int B, int ntok, int D) {
  const int64_t total = (int64_t)B * ntok * D;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int d = (int)(idx % D);
    const int t = (int)((idx / D) % ntok);
    const float base = t == 0 ? cls[d] : to_f<T>(x[idx]);
    x[idx] = from_f<T>(base + pos[(int64_t)t * D + d]);
  }
}
extern "C" int d2r_clip_embed_finish(int dtype, void* x, const float* cls, const float* pos, int B, int ntok, int D,
                                     void* stream) {
  D2R_REQUIRE(x && cls && pos && B >= 1 && ntok >= 1 && D >= 1, "d2r_clip_embed_finish: bad arguments");
  int blocks = d2r_cdiv((int64_t)B * ntok * D, 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) hipLaunchKernelGGL((clip_embed_finish_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (bf16_t*)x, cls, pos, B, ntok, D);
  else if (dtype == D2R_F16) hipLaunchKernelGGL((clip_embed_finish_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, (f16_t*)x, cls, pos, B, ntok, D);
  else if (dtype == D2R_F32) hipLaunchKernelGGL((clip_embed_finish_kernel<float>), dim3(blocks), dim3(256), 0, st, (float*)x, cls, pos, B, ntok, D);
  else return d2r_fail(D2R_ERR_INVALID, "d2r_clip_embed_finish: bad dtype %d", dtype);
  return d2r_check_launch("d2r_clip_embed_finish");
}

__global__ __launch_bounds__(256) void copy_row_kernel(const float* __restrict__ src, float* __restrict__ dst, int D) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < D; c += gridDim.x * 256) dst[c] = src[c];
}
extern "C" int d2r_clip_embed_bwd(int dtype, const void* dX, int B, int ntok, int D, float* dcls, float* dpos,
                                  void* stream) {
  D2R_REQUIRE(dX && dcls && dpos && B >= 1 && ntok >= 1 && D >= 1, "d2r_clip_embed_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == D2R_BF16) hipLaunchKernelGGL((batch_sum_rows_kernel<bf16_t, false>), dim3(ntok), dim3(256), 0, st, (const bf16_t*)dX, B, ntok, D, dpos);
  else if (dtype == D2R_F16) hipLaunchKernelGGL((batch_sum_rows_kernel<f16_t, false>), dim3(ntok), dim3(256), 0, st, (const f16_t*)dX, B, ntok, D, dpos);
  else if (dtype == D2R_F32) hipLaunchKernelGGL((batch_sum_rows_kernel<float, false>), dim3(ntok), dim3(256), 0, st, (const float*)dX, B, ntok, D, dpos);
  else return d2r_fail(D2R_ERR_INVALID, "d2r_clip_embed_bwd: bad dtype %d", dtype);
  hipLaunchKernelGGL(copy_row_kernel, dim3(d2r_cdiv(D, 256)), dim3(256), 0, st, (const float*)dpos, dcls, D);  // dcls = dpos[0]
  return d2r_check_launch("d2r_clip_embed_bwd");
}
