// rowops.hip — row-wise kernels: softmax fwd/bwd, LayerNorm fwd/bwd, l2norm fwd/bwd, column sums.
// One 64-lane wave owns one row; statistics are fp32 and reduced with wave shuffles (no LDS, no atomics).
#include "common.h"

// =====================================================================================================
// softmax over rows of length `cols` (<= 64*MAXPL), wave per row, values held in registers.
// Input and output dtypes are independent: in bf16 mode the logits stay fp32 (softmax(100*s/sqrt(768)) is
// far too sharp for bf16 logits) and only the probabilities are rounded to bf16.
// =====================================================================================================
template <typename TX, typename TY, int MAXPL>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const TX* __restrict__ X, TY* __restrict__ Y, int64_t ld,
                                                          int64_t rows, int cols, float scale,
                                                          const float* __restrict__ mask, int64_t rows_per_mask) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TX* x = X + row * ld;
  const float* mk = mask ? mask + (row / rows_per_mask) * cols : nullptr;
  float v[MAXPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = lane + i * 64;
    if (c < cols) {
      float t = scale * to_f<TX>(x[c]);
      if (mk) t += mk[c];
      v[i] = t;
      mx = fmaxf(mx, t);
    } else {
      v[i] = -INFINITY;
    }
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = lane + i * 64;
    if (c < cols) {
      v[i] = expf(v[i] - mx);
      s += v[i];
    }
  }
  s = wave_sum(s);
  const float inv = 1.f / s;
  TY* y = Y + row * ld;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = lane + i * 64;
    if (c < cols) y[c] = from_f<TY>(v[i] * inv);
  }
}

template <typename TP, typename TD, int MAXPL>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const TP* __restrict__ P, const TD* __restrict__ dP,
                                                          TP* __restrict__ dS, int64_t ld, int64_t rows, int cols,
                                                          float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TP* p = P + row * ld;
  const TD* dp = dP + row * ld;
  float pv[MAXPL], dv[MAXPL];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = lane + i * 64;
    if (c < cols) {
      pv[i] = to_f<TP>(p[c]);
      dv[i] = to_f<TD>(dp[c]);
      dot += pv[i] * dv[i];
    }
  }
  dot = wave_sum(dot);
  TP* ds = dS + row * ld;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = lane + i * 64;
    if (c < cols) ds[c] = from_f<TP>(scale * pv[i] * (dv[i] - dot));
  }
}

template <typename TX, typename TY>
static int softmax_fwd_launch(const void* X, void* Y, int64_t ld, int64_t rows, int cols, float scale,
                              const float* mask, int64_t rpm, hipStream_t st) {
  dim3 grid(d2r_cdiv(rows, 4)), block(256);
  const TX* x = (const TX*)X;
  TY* y = (TY*)Y;
  if (cols <= 256) hipLaunchKernelGGL((softmax_fwd_kernel<TX, TY, 4>), grid, block, 0, st, x, y, ld, rows, cols, scale, mask, rpm);
  else if (cols <= 640) hipLaunchKernelGGL((softmax_fwd_kernel<TX, TY, 10>), grid, block, 0, st, x, y, ld, rows, cols, scale, mask, rpm);
  else hipLaunchKernelGGL((softmax_fwd_kernel<TX, TY, 32>), grid, block, 0, st, x, y, ld, rows, cols, scale, mask, rpm);
  return d2r_check_launch("d2r_softmax_fwd");
}

extern "C" int d2r_softmax_fwd(int x_dtype, int y_dtype, const void* X, void* Y, int64_t ld, int64_t rows, int cols,
                               float scale, const float* mask, int64_t rows_per_mask, void* stream) {
  D2R_REQUIRE(X && Y, "d2r_softmax_fwd: null pointer");
  D2R_REQUIRE(cols >= 1 && cols <= 2048, "d2r_softmax_fwd: cols=%d outside [1,2048]", cols);
  D2R_REQUIRE(ld >= cols && rows >= 0, "d2r_softmax_fwd: bad ld/rows");
  D2R_REQUIRE(!mask || rows_per_mask >= 1, "d2r_softmax_fwd: rows_per_mask must be >= 1 with a mask");
  D2R_REQUIRE(X != Y || x_dtype == y_dtype, "d2r_softmax_fwd: in-place needs equal dtypes");
  if (rows == 0) return D2R_OK;
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == D2R_F32 && y_dtype == D2R_F32) return softmax_fwd_launch<float, float>(X, Y, ld, rows, cols, scale, mask, rows_per_mask, st);
  if (x_dtype == D2R_F32 && y_dtype == D2R_BF16) return softmax_fwd_launch<float, bf16_t>(X, Y, ld, rows, cols, scale, mask, rows_per_mask, st);
  else if (x_dtype == D2R_F32 && y_dtype == D2R_F16) return softmax_fwd_launch<float, f16_t>(X, Y, ld, rows, cols, scale, mask, rows_per_mask, st);
  if (x_dtype == D2R_BF16 && y_dtype == D2R_BF16) return softmax_fwd_launch<bf16_t, bf16_t>(X, Y, ld, rows, cols, scale, mask, rows_per_mask, st);
  else if (x_dtype == D2R_F16 && y_dtype == D2R_F16) return softmax_fwd_launch<f16_t, f16_t>(X, Y, ld, rows, cols, scale, mask, rows_per_mask, st);
  return d2r_fail(D2R_ERR_INVALID, "d2r_softmax_fwd: unsupported dtypes %d -> %d", x_dtype, y_dtype);
}

template <typename TP, typename TD>
static int softmax_bwd_launch(const void* P, const void* dP, void* dS, int64_t ld, int64_t rows, int cols, float scale,
                              hipStream_t st) {
  dim3 grid(d2r_cdiv(rows, 4)), block(256);
  const TP* p = (const TP*)P;
  const TD* dp = (const TD*)dP;
  TP* ds = (TP*)dS;
  if (cols <= 256) hipLaunchKernelGGL((softmax_bwd_kernel<TP, TD, 4>), grid, block, 0, st, p, dp, ds, ld, rows, cols, scale);
  else if (cols <= 640) hipLaunchKernelGGL((softmax_bwd_kernel<TP, TD, 10>), grid, block, 0, st, p, dp, ds, ld, rows, cols, scale);
  else hipLaunchKernelGGL((softmax_bwd_kernel<TP, TD, 32>), grid, block, 0, st, p, dp, ds, ld, rows, cols, scale);
  return d2r_check_launch("d2r_softmax_bwd");
}

extern "C" int d2r_softmax_bwd(int p_dtype, int dp_dtype, const void* P, const void* dP, void* dS, int64_t ld,
                               int64_t rows, int cols, float scale, void* stream) {
  D2R_REQUIRE(P && dP && dS, "d2r_softmax_bwd: null pointer");
  D2R_REQUIRE(cols >= 1 && cols <= 2048, "d2r_softmax_bwd: cols=%d outside [1,2048]", cols);
  D2R_REQUIRE(ld >= cols && rows >= 0, "d2r_softmax_bwd: bad ld/rows");
  D2R_REQUIRE(dP != dS || p_dtype == dp_dtype, "d2r_softmax_bwd: in-place needs equal dtypes");
  if (rows == 0) return D2R_OK;
  hipStream_t st = (hipStream_t)stream;
  if (p_dtype == D2R_F32 && dp_dtype == D2R_F32) return softmax_bwd_launch<float, float>(P, dP, dS, ld, rows, cols, scale, st);
  if (p_dtype == D2R_BF16 && dp_dtype == D2R_F32) return softmax_bwd_launch<bf16_t, float>(P, dP, dS, ld, rows, cols, scale, st);
  else if (p_dtype == D2R_F16 && dp_dtype == D2R_F32) return softmax_bwd_launch<f16_t, float>(P, dP, dS, ld, rows, cols, scale, st);
  if (p_dtype == D2R_BF16 && dp_dtype == D2R_BF16) return softmax_bwd_launch<bf16_t, bf16_t>(P, dP, dS, ld, rows, cols, scale, st);
  else if (p_dtype == D2R_F16 && dp_dtype == D2R_F16) return softmax_bwd_launch<f16_t, f16_t>(P, dP, dS, ld, rows, cols, scale, st);
  return d2r_fail(D2R_ERR_INVALID, "d2r_softmax_bwd: unsupported dtypes %d / %d", p_dtype, dp_dtype);
}

// =====================================================================================================
// generic stage-2 reduction: out[c] = sum_p ws[p*ncols + c]
// =====================================================================================================
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ ws, int nparts, int stride,
                                                           int ncols, float* __restrict__ out) {
  // 64 columns x 4 part-groups per block; 4 independent accumulators per thread keep loads in flight;
  // the order of additions is fixed (deterministic)
  __shared__ float sh[4][64];
  const int l = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < ncols) {
    int p = grp;
    for (; p + 12 < nparts; p += 16) {
      a0 += ws[(int64_t)p * stride + c];
      a1 += ws[(int64_t)(p + 4) * stride + c];
      a2 += ws[(int64_t)(p + 8) * stride + c];
      a3 += ws[(int64_t)(p + 12) * stride + c];
    }
    for (; p < nparts; p += 4) a0 += ws[(int64_t)p * stride + c];
  }
  sh[grp][l] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && c < ncols) out[c] = (sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l]);
}

// stage 2 of LayerNorm backward: column c < D -> dgamma, else dbeta; ws rows are [2*D]; optional accumulation.
// A block sums 16 columns: thread (g = tid / 16, l = tid % 16) walks the partial rows g, g + 16, ... (64-byte segments of the
// freshly written, L2-resident partials), then the 16 groups are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void ln_sum_partials_kernel(const float* __restrict__ ws, int nparts, int D,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int accumulate) {
  __shared__ float sh[16][17];
  const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + l, stride = 2 * D;
  float a0 = 0.f, a1 = 0.f;
  if (c < stride) {
    // eight partial rows per trip, loaded before the first add (rows past the end read row `grp` again, times zero):
    // the walk is a latency chain of L2 hits, not a bandwidth problem
    for (int p = grp; p < nparts; p += 128) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = p + u * 16;
        t[u] = ws[(int64_t)(q < nparts ? q : grp) * stride + c] * (q < nparts ? 1.f : 0.f);
      }
      a0 += (t[0] + t[2]) + (t[4] + t[6]);
      a1 += (t[1] + t[3]) + (t[5] + t[7]);
    }
  }
  sh[grp][l] = a0 + a1;
  __syncthreads();
  if (grp == 0 && c < stride) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += sh[g][l];
    float* dst = c < D ? dgamma + c : dbeta + (c - D);
    *dst = accumulate ? *dst + t : t;
  }
}

// The same sum for up to 32 LayerNorms of one shape in ONE launch (blockIdx.y = problem): the encoder layers' backward calls leave
// their partials in scratch that outlives the call and the caller sums them with the grouped weight gradients (round 3: 52 launches
// of a 7 us latency chain per training step -> 4).  Same walk, same order of additions per problem: bit-identical to the single sum.
struct LnSumGroup {
  const float* ws[32];
  float* dg[32];
  float* db[32];
};
__global__ __launch_bounds__(256) void ln_sum_partials_grouped_kernel(LnSumGroup grp, int nparts, int D, int accumulate) {
  __shared__ float sh[16][17];
  const float* __restrict__ ws = grp.ws[blockIdx.y];
  const int l = threadIdx.x & 15, g16 = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + l, stride = 2 * D;
  float a0 = 0.f, a1 = 0.f;
  if (c < stride) {
    for (int p = g16; p < nparts; p += 128) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = p + u * 16;
        t[u] = ws[(int64_t)(q < nparts ? q : g16) * stride + c] * (q < nparts ? 1.f : 0.f);
      }
      a0 += (t[0] + t[2]) + (t[4] + t[6]);
      a1 += (t[1] + t[3]) + (t[5] + t[7]);
    }
  }
  sh[g16][l] = a0 + a1;
  __syncthreads();
  if (g16 == 0 && c < stride) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += sh[g][l];
    float* dst = c < D ? grp.dg[blockIdx.y] + c : grp.db[blockIdx.y] + (c - D);
    *dst = accumulate ? *dst + t : t;
  }
}

// =====================================================================================================
// LayerNorm: wave per row, row kept in registers as 16-byte packs (D % VEC == 0, D <= 64*VEC*MAXP)
// =====================================================================================================
template <typename T, int MAXP>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ X, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int64_t rows,
                                                            int D, T* __restrict__ Y, float* __restrict__ mean,
                                                            float* __restrict__ rstd) {
  constexpr int VEC = PackOf<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int npk = D / VEC;
  const T* x = X + row * D;
  float v[MAXP][VEC], gm[MAXP][VEC], bt[MAXP][VEC];
  float s = 0.f;
  // every load of the row (x, gamma, beta) is issued before the first reduction: lanes past the row read pack 0 and are
  // masked by a select (a branch around a load costs a full memory wait per load; the parameter loads used to start after
  // the two wave reductions)
  Pack<T, VEC> px[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pk = lane + i * 64, pc = pk < npk ? pk : 0;
    px[i] = ld_pack<T, VEC>(x + pc * VEC);
#pragma unroll
    for (int j = 0; j < VEC; j += 4) {
      const float4 g4 = *reinterpret_cast<const float4*>(gamma + pc * VEC + j);
      const float4 b4 = *reinterpret_cast<const float4*>(beta + pc * VEC + j);
      gm[i][j] = g4.x, gm[i][j + 1] = g4.y, gm[i][j + 2] = g4.z, gm[i][j + 3] = g4.w;
      bt[i][j] = b4.x, bt[i][j + 1] = b4.y, bt[i][j + 2] = b4.z, bt[i][j + 3] = b4.w;
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // all loads of the row are issued before anything is converted
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const float mk = lane + i * 64 < npk ? 1.f : 0.f;  // (a multiply, not a select: a select lets the compiler sink the load under a branch)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      v[i][j] = to_f<T>(px[i].v[j]);
      s += v[i][j] * mk;
    }
  }
  const float mu = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pk = lane + i * 64;
    if (pk < npk) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float d = v[i][j] - mu;
        q += d * d;
      }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / D + eps);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
  T* y = Y + row * D;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pk = lane + i * 64;
    if (pk < npk) {
      Pack<T, VEC> o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>((v[i][j] - mu) * rs * gm[i][j] + bt[i][j]);
      st_pack<T, VEC>(y + pk * VEC, o);
    }
  }
}

// Backward: each wave walks rows (stride = 4*gridDim.x) keeping per-column dgamma/dbeta partials in registers; the loads of the
// NEXT row are issued before the current row is reduced (one row per wave in flight left the kernel latency-bound at ~2 TB/s:
// the chip needs ~13 MB outstanding, 1024 waves x 6 KB gave half of that).  The block's 4 waves are combined through LDS and
// written to ws[block][2][D].
template <typename T, int MAXP>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dY, const T* __restrict__ X,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int64_t rows, int D,
                                                            T* __restrict__ dX, float* __restrict__ ws,
                                                            const T* __restrict__ dres) {
  constexpr int VEC = PackOf<T>::N;
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [4][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int npk = D / VEC;
  float dg[MAXP][VEC], db[MAXP][VEC], gm[MAXP][VEC];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pk = lane + i * 64;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      dg[i][j] = 0.f;
      db[i][j] = 0.f;
      gm[i][j] = (pk < npk) ? gamma[pk * VEC + j] : 0.f;
    }
  }
  struct RowRegs {
    Pack<T, VEC> x[MAXP], dy[MAXP], r[MAXP];
    float mu, rs;
  };
  auto fetch = [&](int64_t row, RowRegs& q) {
    q.mu = mean[row];
    q.rs = rstd[row];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int pk = lane + i * 64;
      if (pk < npk) {
        q.x[i] = ld_pack<T, VEC>(X + row * D + pk * VEC);
        q.dy[i] = ld_pack<T, VEC>(dY + row * D + pk * VEC);
        if (dres) q.r[i] = ld_pack<T, VEC>(dres + row * D + pk * VEC);
      }
    }
  };
  const int64_t step = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  RowRegs cur, nxt;
  if (row < rows) fetch(row, cur);
  for (; row < rows; row += step) {
    const bool more = row + step < rows;
    if (more) fetch(row + step, nxt);  // in flight while this row is reduced and stored
    const float mu = cur.mu, rs = cur.rs;
    float xh[MAXP][VEC], g[MAXP][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int pk = lane + i * 64;
      if (pk < npk) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float d = to_f<T>(cur.dy[i].v[j]);
          xh[i][j] = (to_f<T>(cur.x[i].v[j]) - mu) * rs;
          g[i][j] = d * gm[i][j];
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
          dg[i][j] += d * xh[i][j];
          db[i][j] += d;
        }
      }
    }
    s1 = wave_sum(s1) / D;
    s2 = wave_sum(s2) / D;
    T* dx = dX + row * D;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int pk = lane + i * 64;
      if (pk < npk) {
        Pack<T, VEC> o;
        if (dres) {  // gradient arriving through the skip connection around this LayerNorm's block
#pragma unroll
          for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(rs * (g[i][j] - s1 - xh[i][j] * s2) + to_f<T>(cur.r[i].v[j]));
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(rs * (g[i][j] - s1 - xh[i][j] * s2));
        }
        st_pack<T, VEC>(dx + pk * VEC, o);
      }
    }
    if (more) cur = nxt;
  }
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pk = lane + i * 64;
    if (pk < npk) {
      // 16-byte LDS stores (a lane's VEC consecutive columns): written one float at a time the lanes of a wave are 32 bytes apart and hit
      // eight banks - SQ_LDS_BANK_CONFLICT was 59 % of this kernel's LDS cycles (tests/probes/pmc_lds.sh, round 4)
#pragma unroll
      for (int j = 0; j < VEC; j += 4) {
        Pack<float, 4> pg, pb;
#pragma unroll
        for (int u = 0; u < 4; ++u) pg.v[u] = dg[i][j + u], pb.v[u] = db[i][j + u];
        st_pack<float, 4>(&sh[(wave * 2 + 0) * D + pk * VEC + j], pg);
        st_pack<float, 4>(&sh[(wave * 2 + 1) * D + pk * VEC + j], pb);
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256) {
    const float t = sh[0 * 2 * D + c] + sh[1 * 2 * D + c] + sh[2 * 2 * D + c] + sh[3 * 2 * D + c];
    ws[(int64_t)blockIdx.x * 2 * D + c] = t;
  }
}

static int ln_blocks(int64_t rows) {
  int64_t b = (rows + 3) / 4;
  return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));  // two 4-wave blocks per CU, two rows per wave in flight
}

extern "C" size_t d2r_layernorm_bwd_workspace(int64_t rows, int D) {
  return (size_t)ln_blocks(rows) * 2 * D * sizeof(float);
}

template <typename T>
static int ln_check(const char* fn, int D) {
  constexpr int VEC = PackOf<T>::N;
  if (D % VEC != 0 || D < VEC || D > 64 * VEC * 4) return d2r_fail(D2R_ERR_INVALID, "%s: D=%d unsupported (need D %% %d == 0 and D <= %d)", fn, D, VEC, 64 * VEC * 4);
  return D2R_OK;
}

extern "C" int d2r_layernorm_fwd(int dtype, const void* X, const float* gamma, const float* beta, float eps,
                                 int64_t rows, int D, void* Y, float* mean, float* rstd, void* stream) {
  D2R_REQUIRE(X && Y && gamma && beta && mean && rstd, "d2r_layernorm_fwd: null pointer");
  D2R_REQUIRE(d2r_aligned16(X) && d2r_aligned16(Y), "d2r_layernorm_fwd: X/Y must be 16-byte aligned");
  if (rows == 0) return D2R_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(d2r_cdiv(rows, 4)), block(256);
  if (dtype == D2R_BF16) {
    if (int rc = ln_check<bf16_t>("d2r_layernorm_fwd", D)) return rc;
    if (D <= 64 * 8 * 2) hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t, 2>), grid, block, 0, st, (const bf16_t*)X, gamma, beta, eps, rows, D, (bf16_t*)Y, mean, rstd);
    else hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t, 4>), grid, block, 0, st, (const bf16_t*)X, gamma, beta, eps, rows, D, (bf16_t*)Y, mean, rstd);
  } else if (dtype == D2R_F16) {
    if (int rc = ln_check<f16_t>("d2r_layernorm_fwd", D)) return rc;
    if (D <= 64 * 8 * 2) hipLaunchKernelGGL((layernorm_fwd_kernel<f16_t, 2>), grid, block, 0, st, (const f16_t*)X, gamma, beta, eps, rows, D, (f16_t*)Y, mean, rstd);
    else hipLaunchKernelGGL((layernorm_fwd_kernel<f16_t, 4>), grid, block, 0, st, (const f16_t*)X, gamma, beta, eps, rows, D, (f16_t*)Y, mean, rstd);
  } else if (dtype == D2R_F32) {
    if (int rc = ln_check<float>("d2r_layernorm_fwd", D)) return rc;
    hipLaunchKernelGGL((layernorm_fwd_kernel<float, 4>), grid, block, 0, st, (const float*)X, gamma, beta, eps, rows, D, (float*)Y, mean, rstd);
  } else {
    return d2r_fail(D2R_ERR_INVALID, "d2r_layernorm_fwd: bad dtype %d", dtype);
  }
  return d2r_check_launch("d2r_layernorm_fwd");
}

// dX = LN'(dY) (+ dres); dgamma/dbeta overwritten, or accumulated when `accumulate`
extern "C" int d2r_layernorm_bwd_ex(int dtype, const void* dY, const void* X, const float* gamma, const float* mean,
                                    const float* rstd, int64_t rows, int D, void* dX, const void* dres, float* dgamma,
                                    float* dbeta, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  D2R_REQUIRE(dY && X && gamma && mean && rstd && dX && ((dgamma != nullptr) == (dbeta != nullptr)), "d2r_layernorm_bwd: null pointer");
  D2R_REQUIRE(d2r_aligned16(X) && d2r_aligned16(dY) && d2r_aligned16(dX) && d2r_aligned16(dres), "d2r_layernorm_bwd: tensors must be 16-byte aligned");
  if (workspace_bytes < d2r_layernorm_bwd_workspace(rows, D) || !workspace)
    return d2r_fail(D2R_ERR_WORKSPACE, "d2r_layernorm_bwd: workspace %zu < %zu", workspace_bytes, d2r_layernorm_bwd_workspace(rows, D));
  hipStream_t st = (hipStream_t)stream;
  const int nb = ln_blocks(rows);
  float* ws = (float*)workspace;
  const size_t shmem = (size_t)4 * 2 * D * sizeof(float);
  if (dtype == D2R_BF16) {
    if (int rc = ln_check<bf16_t>("d2r_layernorm_bwd", D)) return rc;
    // register arrays are sized by MAXP = ceil(D / (64 lanes * 8)): D=768 needs 2, not 4 (occupancy)
    if (D <= 64 * 8 * 2) hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, 2>), dim3(nb), dim3(256), shmem, st, (const bf16_t*)dY, (const bf16_t*)X, gamma, mean, rstd, rows, D, (bf16_t*)dX, ws, (const bf16_t*)dres);
    else hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, 4>), dim3(nb), dim3(256), shmem, st, (const bf16_t*)dY, (const bf16_t*)X, gamma, mean, rstd, rows, D, (bf16_t*)dX, ws, (const bf16_t*)dres);
  } else if (dtype == D2R_F16) {
    if (int rc = ln_check<f16_t>("d2r_layernorm_bwd", D)) return rc;
    // register arrays are sized by MAXP = ceil(D / (64 lanes * 8)): D=768 needs 2, not 4 (occupancy)
    if (D <= 64 * 8 * 2) hipLaunchKernelGGL((layernorm_bwd_kernel<f16_t, 2>), dim3(nb), dim3(256), shmem, st, (const f16_t*)dY, (const f16_t*)X, gamma, mean, rstd, rows, D, (f16_t*)dX, ws, (const f16_t*)dres);
    else hipLaunchKernelGGL((layernorm_bwd_kernel<f16_t, 4>), dim3(nb), dim3(256), shmem, st, (const f16_t*)dY, (const f16_t*)X, gamma, mean, rstd, rows, D, (f16_t*)dX, ws, (const f16_t*)dres);
  } else if (dtype == D2R_F32) {
    if (int rc = ln_check<float>("d2r_layernorm_bwd", D)) return rc;
    if (D <= 64 * 4 * 3) hipLaunchKernelGGL((layernorm_bwd_kernel<float, 3>), dim3(nb), dim3(256), shmem, st, (const float*)dY, (const float*)X, gamma, mean, rstd, rows, D, (float*)dX, ws, (const float*)dres);
    else hipLaunchKernelGGL((layernorm_bwd_kernel<float, 4>), dim3(nb), dim3(256), shmem, st, (const float*)dY, (const float*)X, gamma, mean, rstd, rows, D, (float*)dX, ws, (const float*)dres);
  } else {
    return d2r_fail(D2R_ERR_INVALID, "d2r_layernorm_bwd: bad dtype %d", dtype);
  }
  if (int rc = d2r_check_launch("d2r_layernorm_bwd")) return rc;
  if (!dgamma) return D2R_OK;  // deferred: the partials stay in `workspace` for d2r_layernorm_bwd_sum_grouped
  hipLaunchKernelGGL(ln_sum_partials_kernel, dim3(d2r_cdiv(2 * D, 16)), dim3(256), 0, st, ws, nb, D, dgamma, dbeta, accumulate);
  return d2r_check_launch("d2r_layernorm_bwd(sum)");
}

extern "C" int d2r_layernorm_bwd_sum_grouped(const void* const* partials, float* const* dgamma, float* const* dbeta, int n, int64_t rows,
                                             int D, int accumulate, void* stream) {
  D2R_REQUIRE(partials && dgamma && dbeta && n >= 0 && rows >= 1 && D >= 1, "d2r_layernorm_bwd_sum_grouped: bad argument");
  const int nb = ln_blocks(rows);
  for (int i0 = 0; i0 < n; i0 += 32) {
    const int m = n - i0 < 32 ? n - i0 : 32;
    LnSumGroup grp;
    for (int i = 0; i < m; ++i) {
      D2R_REQUIRE(partials[i0 + i] && dgamma[i0 + i] && dbeta[i0 + i], "d2r_layernorm_bwd_sum_grouped: null pointer in problem %d", i0 + i);
      grp.ws[i] = (const float*)partials[i0 + i], grp.dg[i] = dgamma[i0 + i], grp.db[i] = dbeta[i0 + i];
    }
    hipLaunchKernelGGL(ln_sum_partials_grouped_kernel, dim3(d2r_cdiv(2 * D, 16), m), dim3(256), 0, (hipStream_t)stream, grp, nb, D, accumulate);
    if (int rc = d2r_check_launch("d2r_layernorm_bwd_sum_grouped")) return rc;
  }
  return D2R_OK;
}

extern "C" int d2r_layernorm_bwd(int dtype, const void* dY, const void* X, const float* gamma, const float* mean,
                                 const float* rstd, int64_t rows, int D, void* dX, float* dgamma, float* dbeta,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  return d2r_layernorm_bwd_ex(dtype, dY, X, gamma, mean, rstd, rows, D, dX, nullptr, dgamma, dbeta, 0, workspace,
                              workspace_bytes, stream);
}

// =====================================================================================================
// l2norm rows: y = x / (||x|| + 1e-8)
// =====================================================================================================
template <typename T, int MAXP>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ X, T* __restrict__ Y,
                                                         float* __restrict__ norm, int64_t rows, int D) {
  constexpr int VEC = PackOf<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int npk = D / VEC;
  const T* x = X + row * D;
  float v[MAXP][VEC];
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pk = lane + i * 64;
    if (pk < npk) {
      Pack<T, VEC> p = ld_pack<T, VEC>(x + pk * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        v[i][j] = to_f<T>(p.v[j]);
        q += v[i][j] * v[i][j];
      }
    }
  }
  const float n = sqrtf(wave_sum(q));
  if (lane == 0) norm[row] = n;
  const float inv = 1.f / (n + 1e-8f);
  T* y = Y + row * D;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pk = lane + i * 64;
    if (pk < npk) {
      Pack<T, VEC> o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(v[i][j] * inv);
      st_pack<T, VEC>(y + pk * VEC, o);
    }
  }
}

// dx = dy/(n+eps) - x * (sum dy*x) / (n (n+eps)^2)
template <typename T, int MAXP>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ dY, const T* __restrict__ X,
                                                         const float* __restrict__ norm, T* __restrict__ dX,
                                                         int64_t rows, int D) {
  constexpr int VEC = PackOf<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int npk = D / VEC;
  const T* x = X + row * D;
  const T* dy = dY + row * D;
  float xv[MAXP][VEC], dv[MAXP][VEC];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pk = lane + i * 64;
    if (pk < npk) {
      Pack<T, VEC> px = ld_pack<T, VEC>(x + pk * VEC);
      Pack<T, VEC> pd = ld_pack<T, VEC>(dy + pk * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        xv[i][j] = to_f<T>(px.v[j]);
        dv[i][j] = to_f<T>(pd.v[j]);
        dot += xv[i][j] * dv[i][j];
      }
    }
  }
  dot = wave_sum(dot);
  const float n = norm[row];
  const float a = 1.f / (n + 1e-8f);
  const float b = n > 0.f ? dot * a * a / n : 0.f;
  T* dx = dX + row * D;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pk = lane + i * 64;
    if (pk < npk) {
      Pack<T, VEC> o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(dv[i][j] * a - xv[i][j] * b);
      st_pack<T, VEC>(dx + pk * VEC, o);
    }
  }
}

extern "C" int d2r_l2norm_fwd(int dtype, const void* X, void* Y, float* norm, int64_t rows, int D, void* stream) {
  D2R_REQUIRE(X && Y && norm, "d2r_l2norm_fwd: null pointer");
  D2R_REQUIRE(d2r_aligned16(X) && d2r_aligned16(Y), "d2r_l2norm_fwd: X/Y must be 16-byte aligned");
  if (rows == 0) return D2R_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(d2r_cdiv(rows, 4)), block(256);
  if (dtype == D2R_BF16) {
    if (int rc = ln_check<bf16_t>("d2r_l2norm_fwd", D)) return rc;
    hipLaunchKernelGGL((l2norm_fwd_kernel<bf16_t, 4>), grid, block, 0, st, (const bf16_t*)X, (bf16_t*)Y, norm, rows, D);
  } else if (dtype == D2R_F16) {
    if (int rc = ln_check<f16_t>("d2r_l2norm_fwd", D)) return rc;
    hipLaunchKernelGGL((l2norm_fwd_kernel<f16_t, 4>), grid, block, 0, st, (const f16_t*)X, (f16_t*)Y, norm, rows, D);
  } else if (dtype == D2R_F32) {
    if (int rc = ln_check<float>("d2r_l2norm_fwd", D)) return rc;
    hipLaunchKernelGGL((l2norm_fwd_kernel<float, 4>), grid, block, 0, st, (const float*)X, (float*)Y, norm, rows, D);
  } else {
    return d2r_fail(D2R_ERR_INVALID, "d2r_l2norm_fwd: bad dtype %d", dtype);
  }
  return d2r_check_launch("d2r_l2norm_fwd");
}

extern "C" int d2r_l2norm_bwd(int dtype, const void* dY, const void* X, const float* norm, void* dX, int64_t rows,
                              int D, void* stream) {
  D2R_REQUIRE(dY && X && norm && dX, "d2r_l2norm_bwd: null pointer");
  D2R_REQUIRE(d2r_aligned16(X) && d2r_aligned16(dY) && d2r_aligned16(dX), "d2r_l2norm_bwd: tensors must be 16-byte aligned");
  if (rows == 0) return D2R_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(d2r_cdiv(rows, 4)), block(256);
  if (dtype == D2R_BF16) {
    if (int rc = ln_check<bf16_t>("d2r_l2norm_bwd", D)) return rc;
    hipLaunchKernelGGL((l2norm_bwd_kernel<bf16_t, 4>), grid, block, 0, st, (const bf16_t*)dY, (const bf16_t*)X, norm, (bf16_t*)dX, rows, D);
  } else if (dtype == D2R_F16) {
    if (int rc = ln_check<f16_t>("d2r_l2norm_bwd", D)) return rc;
    hipLaunchKernelGGL((l2norm_bwd_kernel<f16_t, 4>), grid, block, 0, st, (const f16_t*)dY, (const f16_t*)X, norm, (f16_t*)dX, rows, D);
  } else if (dtype == D2R_F32) {
    if (int rc = ln_check<float>("d2r_l2norm_bwd", D)) return rc;
    hipLaunchKernelGGL((l2norm_bwd_kernel<float, 4>), grid, block, 0, st, (const float*)dY, (const float*)X, norm, (float*)dX, rows, D);
  } else {
    return d2r_fail(D2R_ERR_INVALID, "d2r_l2norm_bwd: bad dtype %d", dtype);
  }
  return d2r_check_launch("d2r_l2norm_bwd");
}

// =====================================================================================================
// column sums (bias gradients): out[n] = sum_m X[m,n]; stage 1 over row slices, stage 2 sum_partials
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int64_t ld, int64_t M, int N,
                                                     float* __restrict__ ws, float* __restrict__ direct, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float s = 0.f;
  for (int64_t m = blockIdx.y; m < M; m += gridDim.y) s += to_f<T>(X[m * ld + c]);
  if (direct) direct[c] = accumulate ? direct[c] + s : s;  // (one slice: the partial IS the sum)
  else ws[(int64_t)blockIdx.y * N + c] = s;
}

// vectorised variant: 256 threads = 64 column packs (1 KiB contiguous per row and wave) x 4 row groups
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ X, int64_t ld, int64_t M, int N,
                                                         float* __restrict__ ws, float* __restrict__ direct, int accumulate) {
  constexpr int VEC = PackOf<T>::N;
  __shared__ float sh[4][64 * VEC];
  const int pk = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col0 = (blockIdx.x * 64 + pk) * VEC;
  const int64_t rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = blockIdx.y * rows_per, r1 = r0 + rows_per < M ? r0 + rows_per : M;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (col0 < N) {
    for (int64_t m = r0 + rg; m < r1; m += 4) {
      const Pack<T, VEC> p = ld_pack<T, VEC>(X + m * ld + col0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += to_f<T>(p.v[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) sh[rg][pk * VEC + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < 64 * VEC; c += 256) {
    const int col = blockIdx.x * 64 * VEC + c;
    if (col >= N) continue;
    const float t = sh[0][c] + sh[1][c] + sh[2][c] + sh[3][c];
    if (direct) direct[col] = accumulate ? direct[col] + t : t;
    else ws[(int64_t)blockIdx.y * N + col] = t;
  }
}

static int colsum_slices(int64_t M) {
  int64_t s = (M + 31) / 32;
  return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

extern "C" size_t d2r_colsum_workspace(int64_t M, int N) { return (size_t)colsum_slices(M) * N * sizeof(float); }

static int colsum_run(const char* fn, int dtype, const void* X, int64_t ld, int64_t M, int N, float* out, int accumulate, void* workspace,
                      size_t workspace_bytes, void* stream) {
  D2R_REQUIRE(X && out, "%s: null pointer", fn);
  D2R_REQUIRE(ld >= N && M >= 0 && N >= 1, "%s: bad shape", fn);
  if (!workspace || workspace_bytes < d2r_colsum_workspace(M, N))
    return d2r_fail(D2R_ERR_WORKSPACE, "%s: workspace %zu < %zu", fn, workspace_bytes, d2r_colsum_workspace(M, N));
  hipStream_t st = (hipStream_t)stream;
  const int S = colsum_slices(M);
  // one slice of rows (M <= 32: the bias gradients of the per-sample linears, routers and head): its partial sums ARE the result - they
  // go straight to `out` (added to it on request) instead of through the workspace and a second launch
  float* direct = S == 1 ? out : nullptr;
  if (S > 1 && accumulate) return d2r_fail(D2R_ERR_INVALID, "%s: accumulation needs M <= 32 (one row slice)", fn);
  dim3 grid(d2r_cdiv(N, 256), S), block(256);
  float* ws = (float*)workspace;
  if (dtype != D2R_BF16 && dtype != D2R_F16 && dtype != D2R_F32) return d2r_fail(D2R_ERR_INVALID, "%s: bad dtype %d", fn, dtype);
  const int VEC = dtype != D2R_F32 ? 8 : 4;
  const bool vec = d2r_aligned16(X) && (ld * (int64_t)d2r_esize(dtype)) % 16 == 0 && N % VEC == 0;
  if (vec) {
    dim3 gv(d2r_cdiv(N, 64 * VEC), S);
    if (dtype == D2R_BF16) hipLaunchKernelGGL((colsum_vec_kernel<bf16_t>), gv, block, 0, st, (const bf16_t*)X, ld, M, N, ws, direct, accumulate);
    else if (dtype == D2R_F16) hipLaunchKernelGGL((colsum_vec_kernel<f16_t>), gv, block, 0, st, (const f16_t*)X, ld, M, N, ws, direct, accumulate);
    else hipLaunchKernelGGL((colsum_vec_kernel<float>), gv, block, 0, st, (const float*)X, ld, M, N, ws, direct, accumulate);
  } else if (dtype == D2R_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)X, ld, M, N, ws, direct, accumulate);
  else if (dtype == D2R_F16) hipLaunchKernelGGL((colsum_kernel<f16_t>), grid, block, 0, st, (const f16_t*)X, ld, M, N, ws, direct, accumulate);
  else hipLaunchKernelGGL((colsum_kernel<float>), grid, block, 0, st, (const float*)X, ld, M, N, ws, direct, accumulate);
  if (int rc = d2r_check_launch(fn)) return rc;
  if (direct) return D2R_OK;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(d2r_cdiv(N, 64)), dim3(256), 0, st, ws, S, N, N, out);
  return d2r_check_launch("d2r_colsum(sum)");
}
extern "C" int d2r_colsum(int dtype, const void* X, int64_t ld, int64_t M, int N, float* out, void* workspace,
                          size_t workspace_bytes, void* stream) {
  return colsum_run("d2r_colsum", dtype, X, ld, M, N, out, 0, workspace, workspace_bytes, stream);
}
// out[n] += sum_m X[m, n] for M <= 32 rows: a bias gradient added straight into its fp32 sink (one launch instead of three)
extern "C" int d2r_colsum_add(int dtype, const void* X, int64_t ld, int64_t M, int N, float* out, void* workspace,
                              size_t workspace_bytes, void* stream) {
  return colsum_run("d2r_colsum_add", dtype, X, ld, M, N, out, 1, workspace, workspace_bytes, stream);
}
