// gemm_args.h — kernel-argument block and epilogue helpers shared by gemm.hip (generic register-staged kernel) and
// gemm_glds.hip (LDS-DMA pipelined kernel for the large bf16 shapes).
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// The two 16-bit element types share every kernel: same tiles, same LDS images, same transposing reads (ds_read_b64_tr_b16 moves
// 16-bit lanes whatever they hold); only the MFMA opcode differs.  H16<T> names the vector types and the opcodes for T.
template <typename T> struct H16;
template <> struct H16<bf16_t> {
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  static constexpr int DT = D2R_BF16;
  static __device__ __forceinline__ f32x4 mfma32(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x4 mfma16(v4 a, v4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ v4 tr_read(const bf16_t* lds) {  // compiler-tracked form (kernels without LDS-DMA in flight)
    typedef __attribute__((address_space(3))) v4 lds_v4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)lds);
  }
};
template <> struct H16<f16_t> {
  typedef f16x8 v8;
  typedef f16x4 v4;
  static constexpr int DT = D2R_F16;
  static __device__ __forceinline__ f32x4 mfma32(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x4 mfma16(v4 a, v4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ v4 tr_read(const f16_t* lds) {
    typedef __attribute__((ext_vector_type(4))) short i16x4;
    typedef __attribute__((address_space(3))) i16x4 lds_v4;
    return __builtin_bit_cast(v4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)lds));
  }
};
template <> struct H16<float> {  // placeholder so that `typename H16<T>::v8` parses in the discarded 16-bit branches of fp32 kernels
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  static constexpr int DT = D2R_F32;
};
// 16-bit OUTPUT type of a kernel instantiated for inputs T: T itself, or bf16 for the fp32-input kernels
template <typename T> struct Out16 { typedef T type; };
template <> struct Out16<float> { typedef bf16_t type; };

// Which kernel the last d2r_gemm / d2r_gemm_tn_grouped of this thread launched (read by the optional launch timer of gemm.hip):
// 0 register-staged generic kernel, 1 LDS-DMA 128x64, 2 LDS-DMA 128x128 on four waves, 3 LDS-DMA 128x128 on eight waves
// (+10: software-pipelined K-loop), 20 grouped LDS-DMA weight-gradient kernel, 21 grouped generic 64x64, 30 skinny fp32.
extern thread_local int d2r_gemm_variant_tl;

// Grouped mode (weight-gradient GEMMs of identical shape deferred and launched together): operand pointers of problem
// z = blockIdx.z, passed by value so that the launch needs no host-to-device copy and can be captured into a hipGraph.
constexpr int D2R_GEMM_GROUP_MAX = 32;
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
  int dtype;  // element type of A and B
  int band;   // xcd_tile: column-band width in n-tiles (0: n fastest over the whole output)
  int gbatch;   // batched grouped mode of the LDS-DMA kernel: problems per group (blockIdx.z = group * gbatch + batch index)
  int m_store;  // same mode: rows stored (<= M; M itself is the multiple of 8 the operand loads are clamped to)
  int dbg;  // timing experiments (D2R_GEMM_DBG): 1 = no MFMA, 2 = no DMA issue, 3 = no epilogue stores
  unsigned long long* ts;  // timing experiments (d2r_gemm_debug_stamps): s_memtime stamps of workgroup (0,0) of the LDS-DMA kernel, [waves][8]
  unsigned* kflags;  // in-kernel split-K of the LDS-DMA kernel (gemm_glds_splitk_kernel): arrival counters per output tile, zero between launches
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
// `band` > 0 (wide outputs): inside an XCD's run the tiles are walked in column BANDS of `band` n-tiles, m fastest across
// bands - the B panels of one band (band x BN x K elements) stay resident in the XCD's 4 MB L2 while the row panels stream past
// once.  With n fastest over a whole 3072-wide output every row of tiles sweeps a 4.7 MB B matrix that does not fit, and the
// matrix is re-fetched past L2 for every row panel on every XCD (measured: 4.1x the algorithmic bytes for the 128x128 dX kernel).
__device__ __forceinline__ void xcd_tile(int enable, int& tile_m, int& tile_n, int band = 0) {
  const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy;
  int id = blockIdx.y * gx + blockIdx.x;
  if (enable && gridDim.z == 1 && nwg >= 16) {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  if (band > 0 && band < gx) {
    const int per_band = gy * band;
    const int b = id / per_band;
    const int first = b * band;                       // first n-tile of the band
    const int w = min(band, gx - first);              // the last band may be narrower
    const int rem = id - b * per_band;                // (for the last band: rem < gy * w, because the bands tile the grid exactly)
    tile_m = rem / w;
    tile_n = first + rem - tile_m * w;
    return;
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
  else if (c_dtype == D2R_F16) reinterpret_cast<f16_t*>(C)[idx] = (f16_t)v;
  else reinterpret_cast<float*>(C)[idx] = v;
}
__device__ __forceinline__ float load_c(const void* C, int c_dtype, int64_t idx) {
  return c_dtype == D2R_BF16  ? (float)reinterpret_cast<const bf16_t*>(C)[idx]
         : c_dtype == D2R_F16 ? (float)reinterpret_cast<const f16_t*>(C)[idx]
                              : reinterpret_cast<const float*>(C)[idx];
}



// Hardware-transposed LDS read (ds_read_b64_tr_b16) as inline asm.  Through the builtin, hipcc cannot tell which LDS bytes the
// read touches and drains EVERY in-flight LDS-DMA (s_waitcnt vmcnt(0)) in front of it — which serialises a DMA pipeline that
// keeps tiles in flight across the reads.  The asm form is invisible to that pass: the caller orders DMA arrival itself
// (counted vmcnt + barrier) and must put `lds_reads_done()` between the reads and the first MFMA that consumes them
// (cdna_hip_programming.md 5.7 form (iii), rule 18).
template <typename V4>
__device__ __forceinline__ V4 lds_tr_read(const void* p) {
  V4 r;
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
__device__ __forceinline__ void lds_reads_done() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}


// ---- compact epilogue arithmetic ---------------------------------------------------------------------------------------
// The activation of a GEMM epilogue is a RUN-TIME choice (one kernel serves every nn.Linear).  Calling act_apply() per
// element from fully unrolled loops expands the seven-way switch (with erff / tanhf / expf bodies) at every call site:
// the 128x64 kernel grew to 30,000 instructions whose instruction-cache misses cost a third of a K = 768 GEMM.  These
// helpers take the switch ONCE per 8-wide pack; the callers keep the pack loop rolled.
// D2R_FAST_ACT=1 (build-time, `D2R_FAST_ACT=1 python -m d2r_amd.build`; default 0): the 16-bit epilogues evaluate exp and the sigmoid's
// division with the hardware transcendentals (v_exp_f32, v_rcp_f32: about one ulp) and the normal distribution function through
// erfc(z) = t exp(-z^2 + P(t)), t = 1 / (1 + z / 2), P the degree-9 fit published in Numerical Recipes (erfcc): fractional error below
// 1.2e-7 in exact arithmetic, 6e-6 as evaluated in fp32 - the result is rounded to an 8- or 11-bit significand next.  libm's erff / expf
// / IEEE division cost 30-60 VALU instructions per element: with 128 outputs per lane of a 256 x 256 tile more issue time than a K = 768
// main loop (in the step: 49.7 / 57.8 us for the GELU forward / backward products of the text branch against 44.4 / 47.9; every GEMM of a
// step 17.75 -> 17.28 ms one-stream, the step -0.25 ms).  NOT the default: logits and loss are unchanged (1.9e-5 against 2.0e-5 from the
// oracle), but 1.2 % of the fp16 GELU outputs round to the neighbouring value, and the gradient-cosine gates of the 16-bit paths
// (tests/test_gpu_model.py default-init, tests/test_gpu_bench_shapes.py C2 shape) are single draws from a heavy-tailed distribution - the
// derivative of Block's signed square root is unbounded at zero - that such a re-draw moves by more than their margin:
// tests/probes/grad_cos_seeds.py over eight (init, batch) seeds gives 0.29-0.994 (median 0.973) with libm and 0.18-0.991 (median 0.944)
// with this arithmetic, the committed seed 0.994 / 0.944 (profiles/grad_cos_seeds_r04.log).  The gates stay as they are, so does libm.
#ifndef D2R_FAST_ACT
#define D2R_FAST_ACT 0
#endif
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_erfc_abs(float az) {  // erfc(az), az >= 0
  const float t = fast_rcp(fmaf(0.5f, az, 1.f));
  float p = 0.17087277f;
  p = fmaf(p, t, -0.82215223f);
  p = fmaf(p, t, 1.48851587f);
  p = fmaf(p, t, -1.13520398f);
  p = fmaf(p, t, 0.27886807f);
  p = fmaf(p, t, -0.18628806f);
  p = fmaf(p, t, 0.09678418f);
  p = fmaf(p, t, 0.37409196f);
  p = fmaf(p, t, 1.00002368f);
  p = fmaf(p, t, -1.26551223f);
  return t * fast_exp(fmaf(-az, az, p));
}
// Phi(x) = 0.5 (1 + erf(x / sqrt 2))
__device__ __forceinline__ float fast_phi(float x) {
  const float h = 0.5f * fast_erfc_abs(fabsf(x) * 0.70710678118654752f);
  return x >= 0.f ? 1.f - h : h;
}
__device__ __forceinline__ float gelu_fwd(float x) { return D2R_FAST_ACT ? x * fast_phi(x) : 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float r) {
  if (D2R_FAST_ACT) return fast_phi(r) + r * 0.3989422804014327f * fast_exp(-0.5f * r * r);
  const float cdf = 0.5f * (1.f + erff(r * 0.70710678118654752f));
  return cdf + r * 0.3989422804014327f * expf(-0.5f * r * r);
}
__device__ __forceinline__ float sigmoid1702(float x) { return D2R_FAST_ACT ? fast_rcp(1.f + fast_exp(-1.702f * x)) : 1.f / (1.f + expf(-1.702f * x)); }

template <int N>
__device__ __forceinline__ void act_apply_vec(int act, float (&v)[N]) {
  switch (act) {
    case D2R_ACT_RELU:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = fmaxf(v[j], 0.f);
      break;
    case D2R_ACT_TANH:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = tanhf(v[j]);
      break;
    case D2R_ACT_GELU:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = gelu_fwd(v[j]);
      break;
    case D2R_ACT_QUICK_GELU:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = D2R_FAST_ACT ? v[j] * sigmoid1702(v[j]) : v[j] / (1.f + expf(-1.702f * v[j]));
      break;
    case D2R_ACT_TANH_RELU:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = fmaxf(tanhf(v[j]), 0.f);
      break;
    case D2R_ACT_SIGMOID:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = 1.f / (1.f + expf(-v[j]));
      break;
    default: break;
  }
}
// v[j] *= d act / d x evaluated at r[j] (activation output, or pre-activation for gelu / quick_gelu: see act_grad)
template <int N>
__device__ __forceinline__ void act_grad_mul_vec(int act, const float (&r)[N], float (&v)[N]) {
  switch (act) {
    case D2R_ACT_RELU:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = r[j] > 0.f ? v[j] : 0.f;
      break;
    case D2R_ACT_TANH:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] *= 1.f - r[j] * r[j];
      break;
    case D2R_ACT_GELU:
#pragma unroll
      for (int j = 0; j < N; ++j) {
        v[j] *= gelu_grad(r[j]);
      }
      break;
    case D2R_ACT_QUICK_GELU:
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float sg = sigmoid1702(r[j]);
        v[j] *= sg + 1.702f * r[j] * sg * (1.f - sg);
      }
      break;
    case D2R_ACT_TANH_RELU:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = r[j] > 0.f ? v[j] * (1.f - r[j] * r[j]) : 0.f;
      break;
    case D2R_ACT_SIGMOID:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] *= r[j] * (1.f - r[j]);
      break;
    default: break;
  }
}
// scalar forms for the rarely taken element-wise paths: ONE out-of-line copy instead of one expansion per call site
static __device__ __attribute__((noinline)) float act_apply_cold(int act, float x) { return act_apply(act, x); }
static __device__ __attribute__((noinline)) float act_grad_cold(int act, float r) { return act_grad(act, r); }

// One 8-column pack of the LDS-staged 16-bit epilogue, shared by the MFMA GEMM kernels:
//   out = act(pv) [* act'(G)] [+ R] [+ beta * C_old],  preact <- pv;   `n_ok` = valid columns of the pack (8 = whole pack)
template <typename H>
__device__ __forceinline__ void epilogue_pack8(const GemmArgs& g, const Pack<H, 8>& pv, H* Cg, H* Pg, const H* Rg, const H* Gg, int64_t ci,
                                               int64_t ri, int n_ok) {
  float v[8], t[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = (float)pv.v[u];
  if (n_ok >= 8) {
    if (Pg) st_pack<H, 8>(Pg + ci, pv);
    act_apply_vec<8>(g.act, v);
    if (Gg) {
      const Pack<H, 8> gv = ld_pack<H, 8>(Gg + ci);
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = (float)gv.v[u];
      act_grad_mul_vec<8>(g.gact, t, v);
    }
    if (Rg) {
      const Pack<H, 8> rv = ld_pack<H, 8>(Rg + ri);
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] += (float)rv.v[u];
    }
    if (g.beta != 0.f) {
      const Pack<H, 8> cv = ld_pack<H, 8>(Cg + ci);
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] += g.beta * (float)cv.v[u];
    }
    Pack<H, 8> ov;
#pragma unroll
    for (int u = 0; u < 8; ++u) ov.v[u] = (H)v[u];
    st_pack<H, 8>(Cg + ci, ov);
    return;
  }
  // ragged right edge: element by element (static indices only: a runtime index into the packs would go to scratch)
  act_apply_vec<8>(g.act, v);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (u < n_ok) {
      if (Pg) Pg[ci + u] = pv.v[u];
      float x = v[u];
      if (Gg) x *= act_grad_cold(g.gact, (float)Gg[ci + u]);
      if (Rg) x += (float)Rg[ri + u];
      if (g.beta != 0.f) x += g.beta * (float)Cg[ci + u];
      Cg[ci + u] = (H)x;
    }
  }
}
