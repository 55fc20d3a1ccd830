// gemm.hip — K11 gemm_bias_act: LDS-tiled MFMA GEMM family for gfx950 (wave64).
//
//   C[z] = epilogue(alpha * A[z] x B[z]),   z = batch index (two-level strides: b*s?b + h*s?h)
//
// bf16 inputs use v_mfma_f32_16x16x32_bf16 (each lane feeds 8 consecutive k), fp32 inputs the exact-f32
// v_mfma_f32_16x16x4_f32.  Operand tiles are staged global -> registers -> LDS with 16-byte accesses, the loads
// of tile t+1 being issued before the MFMAs of tile t.  An operand whose global layout is k-contiguous
// (A of NT/NN, B of NT) sits in LDS as [row][k] and a fragment is one ds_read_b128; an operand that is
// k-STRIDED in memory (B of NN, A and B of TN) is kept as loaded, [k][row], and its bf16 fragments are fetched
// with two ds_read_b64_tr_b16 (hardware transpose read; lane map pinned by tests/probes/tr_probe.hip), fp32
// fragments with plain ds_read_b32 — no transposing stores anywhere.
//
// Weight-gradient GEMMs (TN, 768x768 outputs reduced over thousands of tokens) have few output tiles: they are
// split along K over gridDim.z into fp32 partial slabs in a caller workspace and combined by a fixed-order
// reduce kernel (deterministic; no float atomics).
//
// Replaces (reference file:line): every F.linear / torch.bmm / torch.matmul of the hot path — see
// include/d2r_hip.h, section K11.
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "gemm_args.h"
#include "ktimer.h"

template <typename T, int LAYOUT, int BM, int BN, int WAVES_M, int WAVES_N, int NBUF, bool GROUPED = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g, GemmGroup grp) {
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int BK = IS_BF16 ? 64 : 16;
  constexpr int VEC = 16 / sizeof(T);
  constexpr int NT = 256;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // per-wave output tile
  constexpr int TM = WM / 16, TN = WN / 16;            // 16x16 MFMA tiles per wave
  static_assert(WM % 16 == 0 && WN % 16 == 0, "wave tile must be a multiple of 16");
  constexpr bool A_KCONT = (LAYOUT != D2R_GEMM_TN);  // A stored [M,K]
  constexpr bool B_KCONT = (LAYOUT == D2R_GEMM_NT);  // B stored [N,K]
  // LDS row strides (elements): k-contiguous image [rows][BK + pad]; k-strided image [BK][rows + pad]
  constexpr int LDK = BK + (IS_BF16 ? 8 : 1);        // bf16: 144 B rows -> conflict-free ds_read_b128
  constexpr int LDA_T = BM + (IS_BF16 ? 8 : 16);
  constexpr int LDB_T = BN + (IS_BF16 ? 8 : 16);
  constexpr int SZ_A = A_KCONT ? BM * LDK : BK * LDA_T;
  constexpr int SZ_B = B_KCONT ? BN * LDK : BK * LDB_T;
  constexpr int CH_A = BM * BK / VEC, CH_B = BN * BK / VEC;  // 16-B chunks per tile
  constexpr int IT_A = (CH_A + NT - 1) / NT, IT_B = (CH_B + NT - 1) / NT;

  // one LDS array: two operand-tile buffers (double buffering); reused by the vectorised bf16 epilogue
  constexpr int SZ_BUF = SZ_A + SZ_B;
  constexpr int SZ_EPI = 4 * WM * (WN + 8);  // bf16 C staging, per wave [WM][WN+8]
  constexpr int SZ_ALL = (NBUF * SZ_BUF * (int)sizeof(T) >= SZ_EPI * 2) ? NBUF * SZ_BUF : (SZ_EPI * 2 + (int)sizeof(T) - 1) / (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) T smem[SZ_ALL];
  T* As = smem;
  T* Bs = smem + SZ_A;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  int tile_m, tile_n;
  int z = blockIdx.z, split = 0;
  if constexpr (GROUPED) xcd_tile_3d(g.xcd, tile_m, tile_n, z);
  else xcd_tile(g.xcd, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (g.splits > 1) {
    split = z;
    z = 0;
  }
  const int zb = z / g.nh, zh = z % g.nh;
  const T* A = reinterpret_cast<const T*>(g.A) + zb * g.sAb + zh * g.sAh;
  const T* B = reinterpret_cast<const T*>(g.B) + zb * g.sBb + zh * g.sBh;
  if constexpr (GROUPED) {  // problem z has its own operands (same shape and leading dimensions)
    A = reinterpret_cast<const T*>(grp.A[z]);
    B = reinterpret_cast<const T*>(grp.B[z]);
    g.C = grp.C[z];
    g.dbias = grp.dbias[z];
  }
  const bool vecA = g.vecA != 0, vecB = g.vecB != 0;

  Pack<T, VEC> ra[IT_A], rb[IT_B];

  auto prefetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < IT_A; ++it) {
      const int c = tid + it * NT;
      if (CH_A % NT == 0 || c < CH_A) {
        if constexpr (A_KCONT) {
          const int r = c / (BK / VEC), kc = (c % (BK / VEC)) * VEC;
          const int row = m0 + r;
          ra[it] = load_guard<T, VEC>(A + (int64_t)row * g.lda, k0 + kc, g.K, row < g.M, vecA);
        } else {  // stored [K][M]
          const int k = c / (BM / VEC), rc = (c % (BM / VEC)) * VEC;
          const int row = k0 + k;
          ra[it] = load_guard<T, VEC>(A + (int64_t)row * g.lda, m0 + rc, g.M, row < g.K, vecA);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < IT_B; ++it) {
      const int c = tid + it * NT;
      if (CH_B % NT == 0 || c < CH_B) {
        if constexpr (B_KCONT) {
          const int r = c / (BK / VEC), kc = (c % (BK / VEC)) * VEC;
          const int row = n0 + r;
          rb[it] = load_guard<T, VEC>(B + (int64_t)row * g.ldb, k0 + kc, g.K, row < g.N, vecB);
        } else {  // stored [K][N]
          const int k = c / (BN / VEC), rc = (c % (BN / VEC)) * VEC;
          const int row = k0 + k;
          rb[it] = load_guard<T, VEC>(B + (int64_t)row * g.ldb, n0 + rc, g.N, row < g.K, vecB);
        }
      }
    }
  };

  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < IT_A; ++it) {
      const int c = tid + it * NT;
      if (CH_A % NT == 0 || c < CH_A) {
        if constexpr (A_KCONT) {
          const int r = c / (BK / VEC), kc = (c % (BK / VEC)) * VEC;
          if constexpr (IS_BF16) {
            st_pack<T, VEC>(&As[r * LDK + kc], ra[it]);
          } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) As[r * LDK + kc + j] = ra[it].v[j];
          }
        } else {
          const int k = c / (BM / VEC), rc = (c % (BM / VEC)) * VEC;
          st_pack<T, VEC>(&As[k * LDA_T + rc], ra[it]);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < IT_B; ++it) {
      const int c = tid + it * NT;
      if (CH_B % NT == 0 || c < CH_B) {
        if constexpr (B_KCONT) {
          const int r = c / (BK / VEC), kc = (c % (BK / VEC)) * VEC;
          if constexpr (IS_BF16) {
            st_pack<T, VEC>(&Bs[r * LDK + kc], rb[it]);
          } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) Bs[r * LDK + kc + j] = rb[it].v[j];
          }
        } else {
          const int k = c / (BN / VEC), rc = (c % (BN / VEC)) * VEC;
          st_pack<T, VEC>(&Bs[k * LDB_T + rc], rb[it]);
        }
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias gradient of a linear layer inside its weight-gradient GEMM (TN): the workgroups of the first tile column
  // also multiply their A fragments with an all-ones B fragment -> acc_b[i][r] = sum_k A[k, row] in every column
  const bool do_bias = (LAYOUT == D2R_GEMM_TN) && g.dbias != nullptr && tile_n == 0;
  f32x4 acc_b[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) acc_b[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int tq = (lane & 15) >> 2, tp = lane & 3;  // transpose-read address roles inside a 16-lane group
  const int nk_all = (g.K + BK - 1) / BK;
  int t_begin = 0, t_end = nk_all;
  if (g.splits > 1) {
    t_begin = split * g.tiles_per_split;
    t_end = min(nk_all, t_begin + g.tiles_per_split);
  }
  // pipeline: global loads of tile t+1 are issued before the MFMAs of tile t and written to the OTHER LDS buffer
  // after them; one barrier per K-tile
  int cur = 0;
  if (t_begin < t_end) {
    prefetch(t_begin * BK);
    if constexpr (NBUF == 2) stage();
  }
  if constexpr (NBUF == 2) __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const bool more = t + 1 < t_end;
    if constexpr (NBUF == 1) {  // single buffer: write, barrier, then issue the next tile's loads
      stage();
      __syncthreads();
    }
    if (more) prefetch((t + 1) * BK);
    if constexpr (NBUF == 2) {
      As = smem + cur * SZ_BUF;
      Bs = As + SZ_A;
    }
    if constexpr (IS_BF16) {
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        typedef typename H16<T>::v8 h8;
        typedef typename H16<T>::v4 h4;
        h8 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if constexpr (A_KCONT) {
            af[i] = *reinterpret_cast<const h8*>(&As[(wm0 + i * 16 + fr) * LDK + kk * 32 + fq * 8]);
          } else {
            const T* p0 = &As[(kk * 32 + fq * 8 + tq) * LDA_T + wm0 + i * 16 + tp * 4];
            const h4 lo = H16<T>::tr_read(p0), hi = H16<T>::tr_read(p0 + 4 * LDA_T);
            af[i] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (B_KCONT) {
            bfr[j] = *reinterpret_cast<const h8*>(&Bs[(wn0 + j * 16 + fr) * LDK + kk * 32 + fq * 8]);
          } else {
            const T* p0 = &Bs[(kk * 32 + fq * 8 + tq) * LDB_T + wn0 + j * 16 + tp * 4];
            const h4 lo = H16<T>::tr_read(p0), hi = H16<T>::tr_read(p0 + 4 * LDB_T);
            bfr[j] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = H16<T>::mfma32(af[i], bfr[j], acc[i][j]);
        if constexpr (LAYOUT == D2R_GEMM_TN) {
          if (do_bias) {
            const T one = (T)1.f;
            const h8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
            for (int i = 0; i < TM; ++i) acc_b[i] = H16<T>::mfma32(af[i], ones, acc_b[i]);
          }
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        float af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = A_KCONT ? As[(wm0 + i * 16 + fr) * LDK + ks * 4 + fq] : As[(ks * 4 + fq) * LDA_T + wm0 + i * 16 + fr];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[j] = B_KCONT ? Bs[(wn0 + j * 16 + fr) * LDK + ks * 4 + fq] : Bs[(ks * 4 + fq) * LDB_T + wn0 + j * 16 + fr];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
        if constexpr (LAYOUT == D2R_GEMM_TN) {
          if (do_bias) {
#pragma unroll
            for (int i = 0; i < TM; ++i) acc_b[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], 1.f, acc_b[i], 0, 0, 0);
          }
        }
      }
    }
    if constexpr (NBUF == 2) {
      if (more) {
        As = smem + (cur ^ 1) * SZ_BUF;
        Bs = As + SZ_A;
        stage();
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg -------------
  if constexpr (LAYOUT == D2R_GEMM_TN) {
    if (do_bias && (wave % WAVES_N) == 0 && fr == 0) {  // one wave per row band, one lane per 4 rows
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm0 + i * 16 + fq * 4 + r;
          if (row >= g.M) continue;
          if (g.splits > 1) g.ws[(int64_t)g.splits * g.M * g.N + (int64_t)split * g.M + row] = acc_b[i][r];
          else g.dbias[row] += acc_b[i][r];  // this workgroup is the only writer of the row
        }
    }
  }
  if (g.splits > 1) {  // raw fp32 partial slab; the reduce kernel applies alpha/beta
    float* slab = g.ws + (int64_t)split * g.M * g.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 16 + fr;
        if (col >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm0 + i * 16 + fq * 4 + r;
          if (row < g.M) slab[(int64_t)row * g.N + col] = acc[i][j][r];
        }
      }
    return;
  }
  const int64_t cz = zb * g.sCb + zh * g.sCh;
  const int64_t rz = zb * g.sRb + zh * g.sRh;
  typedef typename Out16<T>::type O16;
  if (g.c_dtype == H16<O16>::DT && g.vecC) {
    // 16-bit output: the MFMA layout gives each lane one column of four rows (2-byte stores, 32 B segments).
    // Stage v = alpha*acc + bias per wave in LDS as bf16 [WM][WN+8], then each lane handles 8 consecutive columns
    // of one row: 16-byte loads of residual / old C, 16-byte stores of preact and C (8 lanes = one 128 B row).
    constexpr int LDE = WN + 8;
    O16* Cs = reinterpret_cast<O16*>(smem) + wave * WM * LDE;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 16 + fr;
        const float bv = (g.bias && col < g.N) ? g.bias[zb * g.sBiasB + col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(i * 16 + fq * 4 + r) * LDE + j * 16 + fr] = (O16)(g.alpha * acc[i][j][r] + bv);
      }
    // same-wave hand-off through LDS: wait for the wave's own ds_writes (no block barrier: regions are private)
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    constexpr int CPR = WN / 8;  // 16-byte chunks per row
    O16* Cg = reinterpret_cast<O16*>(g.C);
    O16* Pg = reinterpret_cast<O16*>(g.P);
    const O16* Rg = reinterpret_cast<const O16*>(g.R);
    const O16* Gg = reinterpret_cast<const O16*>(g.G);
#pragma unroll 1
    for (int it = 0; it < WM * CPR / 64; ++it) {  // (rolled on purpose: one copy of the epilogue arithmetic)
      const int e = it * 64 + lane;
      const int rl = e / CPR, ch = e % CPR;
      const int row = m0 + wm0 + rl, col = n0 + wn0 + ch * 8;
      if (row >= g.M || col >= g.N) continue;
      const Pack<O16, 8> pv = ld_pack<O16, 8>(Cs + rl * LDE + ch * 8);
      const int64_t ci = cz + (int64_t)row * g.ldc + col;
      const int64_t ri = rz + (int64_t)row * g.ldr + col;
      epilogue_pack8(g, pv, Cg, Pg, Rg, Gg, ci, ri, g.N - col);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn0 + j * 16 + fr;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[zb * g.sBiasB + col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm0 + i * 16 + fq * 4 + r;
        if (row >= g.M) continue;
        float v = g.alpha * acc[i][j][r] + bv;
        const int64_t ci = cz + (int64_t)row * g.ldc + col;
        if (g.P) store_c(g.P, g.c_dtype, ci, v);
        v = act_apply_cold(g.act, v);
        if (g.R) v += load_c(g.R, g.c_dtype, rz + (int64_t)row * g.ldr + col);
        if (g.beta != 0.f) v += g.beta * load_c(g.C, g.c_dtype, ci);
        store_c(g.C, g.c_dtype, ci, v);
      }
    }
  }
}

// C[m,n] = epilogue(alpha * sum_s ws[s][m][n])   (fixed order over s; same epilogue as the main kernel)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g) {
  const int64_t total = (int64_t)g.M * g.N;
  if (g.dbias) {
    const float* wb = g.ws + (int64_t)g.splits * total;
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < g.M; m += (int64_t)gridDim.x * 256) {
      float s = 0.f;
      for (int k = 0; k < g.splits; ++k) s += wb[(int64_t)k * g.M + m];
      g.dbias[m] += s;
    }
  }
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    // eight slabs per trip, loaded before the first add (slabs past the last are slab 0 times zero; same order of additions):
    // a load per trip with a wait behind it made the walk a chain of up to sixteen memory latencies
    for (int k0 = 0; k0 < g.splits; k0 += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u;
        t[u] = g.ws[(int64_t)(k < g.splits ? k : 0) * total + idx] * (k < g.splits ? 1.f : 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    const int m = (int)(idx / g.N), n = (int)(idx - (int64_t)m * g.N);
    const int64_t ci = (int64_t)m * g.ldc + n;
    float v = g.alpha * s + (g.bias ? g.bias[n] : 0.f);
    if (g.P) store_c(g.P, g.c_dtype, ci, v);
    v = act_apply_cold(g.act, v);
    if (g.G) v *= act_grad_cold(g.gact, load_c(g.G, g.c_dtype, ci));
    if (g.R) v += load_c(g.R, g.c_dtype, (int64_t)m * g.ldr + n);
    if (g.beta != 0.f) v += g.beta * load_c(g.C, g.c_dtype, ci);
    store_c(g.C, g.c_dtype, ci, v);
  }
}

// tuning state (A/B measurements through d2r_gemm_tuning, include/d2r_hip_probes.h); the defaults are the measured winners and the
// library reads no environment variable
static int g_nbuf = 1;
static int g_vepi = 1;
static int g_tile = -1;
static int g_xcd = 1;
static int g_wgrad_glds = 1;
static int g_dbg = 0;  // ablation switches of the measurement build (-DD2R_GEMM_PROBES=1): tile code 2000 + mode
// 256-wide tiles for forward / dX products (tile 110 / 111: off / on), taken when the product fills at least 70 % of the slots of its
// rounds of workgroups with at least g_gemm8_min tiles (gemm8_pays).  Measured (tests/probes/gemm8_probe.py, profiles/gemm8_probe_r04.log):
// the deep-pipelined loop reaches 1360-1480 TFLOP/s at K >= 4096, but the K = 768 / 3072 products of the path have 48-300 such tiles
// (one partial round of workgroups) and twelve K-tiles of loop between a 2.5 k-cycle prologue and the store burst of the whole grid:
// alone on the GPU 530-860 TFLOP/s against 580-940 on the 128-wide kernels; in the training step, where the other branch stream
// fills the CUs a partial round leaves free, the rule is worth 0.5-0.9 % (21.06 against 21.21 ms on one box).
static int g_gemm8 = 1;
static int g_group = 1;  // d2r_gemm_group: grouped launches of independent forward / dX products (d2r_gemm_tuning tile 120 / 121: off / on)
static int g_gemm8_wgrad = 1;  // ... for the grouped weight gradients (tile 102 / 103)
static int g_splitk = 1;  // in-kernel split-K of the 128-wide kernel for deep reductions over a partial round of tiles (tile 130 / 131: off / on)
static int g_gemm8_min = 150;  // fewest 256 x 256 tiles of a forward / dX product that takes them (tile 1000 + n sets it)

// ---- optional per-launch timing of the GEMM entry points (bench.py's roofline leg) -----------------------------------------
// The whole-layer / whole-module C calls (encoder_layer.hip, interaction.hip) launch their GEMMs from inside the library, where
// a Python-side event bracket cannot see them.  When enabled, d2r_gemm and d2r_gemm_tn_grouped bracket each launch with a pair
// of HIP events on the launching stream and remember (family, flops, algorithmic bytes); d2r_gemm_timer_read resolves them.
// Measurement aid: off by default, one mutex acquisition per GEMM when off is avoided by the plain flag test.
thread_local int d2r_gemm_variant_tl = 0;

struct GemmTimerRec {
  int family;  // dtype * 8 + layout * 2 + grouped + 100 * kernel variant (d2r_gemm_variant_tl)
  double flops, bytes;
  hipEvent_t e0, e1;
};
static std::vector<GemmTimerRec> g_timer_recs;
static std::mutex g_timer_mu;
static volatile int g_timer_on = 0;

// An op that is itself timed (d2r_xattn_bwd_multi) may reach GEMM entry points with scopes of their own (its second-generation fallback):
// only the OUTERMOST scope of a thread records, or the same GPU time would appear under two families.
static thread_local int g_timer_depth = 0;
D2RTimerScope::D2RTimerScope(hipStream_t stream, int fam, double fl, double by) : family(fam), flops(fl), bytes(by), st(stream) {
  if (fam < 10000) d2r_gemm_variant_tl = 0;
  if (g_timer_depth++ > 0) return;
  if (!g_timer_on) return;
  if (hipEventCreate(&e0) != hipSuccess) return;
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return; }
  (void)hipEventRecord(e0, st);
  armed = true;
}
D2RTimerScope::~D2RTimerScope() {
  --g_timer_depth;
  if (!armed) return;
  GemmTimerRec rec;
  rec.family = family < 10000 ? family + 100 * d2r_gemm_variant_tl : family;
  rec.flops = flops, rec.bytes = bytes, rec.e0 = e0, rec.e1 = e1;
  (void)hipEventRecord(e1, st);
  std::lock_guard<std::mutex> lk(g_timer_mu);
  g_timer_recs.push_back(rec);
}
typedef D2RTimerScope GemmTimerScope;

extern "C" int d2r_gemm_timer(int on) {
  std::lock_guard<std::mutex> lk(g_timer_mu);
  if (on) {
    for (auto& r : g_timer_recs) (void)hipEventDestroy(r.e0), (void)hipEventDestroy(r.e1);
    g_timer_recs.clear();
  }
  g_timer_on = on ? 1 : 0;
  return D2R_OK;
}

extern "C" int d2r_gemm_timer_read(int* family, double* flops, double* bytes, float* ms, int capacity) {
  std::lock_guard<std::mutex> lk(g_timer_mu);
  int n = 0;
  for (auto& r : g_timer_recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess && n < capacity && family) {
      family[n] = r.family, flops[n] = r.flops, bytes[n] = r.bytes, ms[n] = t;
      ++n;
    }
    (void)hipEventDestroy(r.e0), (void)hipEventDestroy(r.e1);
  }
  const int total = (int)g_timer_recs.size();
  g_timer_recs.clear();
  return family ? n : total;
}
extern "C" void d2r_gemm_tuning(int nbuf, int vepi, int tile) {
  g_nbuf = nbuf & 0xff;
  g_xcd = (nbuf >> 8) & 1 ? 0 : 1;  // bit 8 of the first argument disables the XCD-aware tile order (A/B runs)
  g_vepi = vepi;
  if (tile == 100 || tile == 101) g_wgrad_glds = tile - 100;  // A/B switch of the grouped weight-gradient kernel (0: 64x64 generic)
  else if (tile == 102 || tile == 103) g_gemm8_wgrad = tile - 102;  // 256 x 256 deep-pipelined grouped weight gradients off / on
  else if (tile >= 110 && tile <= 113) g_gemm8 = tile - 110;        // 256 x 256 forward / dX products off / on / (A/B: 2 = not with an activation in the epilogue, 3 = only without any second operand)
  else if (tile == 130 || tile == 131) g_splitk = tile - 130;        // in-kernel split-K of the 128-wide kernel off / on
  else if (tile == 120 || tile == 121) g_group = tile - 120;        // grouped launches of d2r_gemm_group off / on
  else if (tile >= 2000) g_dbg = tile - 2000;
  else if (tile >= 1000) g_gemm8_min = tile - 1000;
  else g_tile = tile;
}

template <typename T, int LAYOUT, int BM, int BN, int WM_, int WN_>
static void launch_tile(const GemmArgs& a, int gz, hipStream_t st) {
  dim3 grid(d2r_cdiv(a.N, BN), d2r_cdiv(a.M, BM), gz);
  static const GemmGroup no_group = {};
  if (g_nbuf == 2) hipLaunchKernelGGL((gemm_kernel<T, LAYOUT, BM, BN, WM_, WN_, 2>), grid, dim3(256), 0, st, a, no_group);
  else hipLaunchKernelGGL((gemm_kernel<T, LAYOUT, BM, BN, WM_, WN_, 1>), grid, dim3(256), 0, st, a, no_group);
}

int d2r_gemm_glds_try(const GemmArgs& a, int layout, int batch, int bn, hipStream_t st);  // gemm_glds.hip
int d2r_gemm8_fwd_ok(const GemmArgs& a, int layout, int batch);                           // gemm8.hip
int d2r_gemm8_fwd_launch(const GemmArgs* probs, int n, int layout, hipStream_t st);
int d2r_gemm8_wgrad_ok(int dtype, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, const void* A, const void* B, const void* C);
int d2r_gemm8_wgrad_launch(int dtype, int count, const int* M, const int* N, const int* K, const int64_t* lda, const int64_t* ldb, const int64_t* ldc,
                           const void* const* A, const void* const* B, float* const* C, float* const* dbias, float beta, hipStream_t st);
// One workgroup per CU: a launch of t tiles runs in ceil(t / 256) rounds.  The wide tiles pay when the last round is well filled.
int d2r_gemm_glds_splitk_plan(const GemmArgs& a, int layout, int batch, size_t ws_bytes, size_t* slab_bytes);
int d2r_gemm_glds_splitk_launch(const GemmArgs& a, int layout, hipStream_t st);
static bool gemm8_pays(const GemmArgs& a) {
  if (g_gemm8 == 2 && (a.act != 0 || a.G != nullptr)) return false;
  if (g_gemm8 == 3 && (a.act != 0 || a.G != nullptr || a.R != nullptr || a.P != nullptr)) return false;
  const int64_t t = (int64_t)d2r_cdiv(a.M, 256) * d2r_cdiv(a.N, 256);
  if (t < g_gemm8_min) return false;
  const int64_t rounds = (t + 255) / 256;
  return t * 100 >= rounds * 256 * 70;  // at least 70 % of the slots of its rounds
}
static int g_glds = 1;

// ---- skinny fp32 GEMM: M <= 32 rows (router MLPs, poolers, Block head: per-sample vectors, batch-size rows) ------------------
// C[M,N] = act(alpha * A[M,K] op(B) + bias) (+ R) (+ beta * C), fp32 in and out, NT (B [N,K]) or NN (B [K,N]), batched.
// These products are latency chains, not throughput problems (2.4 MB of weights, 38 MFLOP): the tiled kernel walks 48 k-tiles
// of 16 with a barrier each and needs split-K slabs + a reduce launch to occupy the chip (22 + 6 us).  Here a workgroup owns
// 16 output columns; its four waves split K, every wave issues ALL its loads before the first MFMA (one memory latency),
// both 16-row tiles share the B fragments, and the four partial tiles are summed through LDS in a fixed order.
// k-slot mapping of v_mfma_f32_16x16x4_f32: lane group g = lane / 16 feeds k-slot g; a lane loads 4 consecutive k (one float4 per
// operand row) and uses element j in the j-th of four MFMAs, i.e. MFMA j covers k0 + {j, 4+j, 8+j, 12+j}: a permutation of
// the 16 k of the step, the same for both operands.
template <int LAYOUT>
__global__ __launch_bounds__(256) void gemm_skinny_f32_kernel(GemmArgs g) {
  constexpr int MAXS = 12;  // steps of 64 k per pass (4 waves x 16): K = 768 in one pass
  __shared__ float red[4][2][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int z = blockIdx.z, zb = z / g.nh, zh = z - zb * g.nh;
  const float* A = reinterpret_cast<const float*>(g.A) + zb * g.sAb + zh * g.sAh;
  const float* B = reinterpret_cast<const float*>(g.B) + zb * g.sBb + zh * g.sBh;
  const int n0 = blockIdx.x * 16;
  const int col = min(n0 + fr, g.N - 1);               // clamped: the matching outputs are not stored
  const int r0 = min(fr, g.M - 1), r1 = min(16 + fr, g.M - 1);
  const bool two = g.M > 16;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  // the K / 16 steps of 16 are dealt to the four waves in contiguous runs (round 3: K a multiple of 16, not of 64: Block's 80- and
  // 1200-deep chunk products took the tiled kernel, 12 and 52 us each on the serial stretch between forward and backward)
  const int k16 = g.K / 16, per = (k16 + 3) / 4;
  const int kbeg = min(wave * per, k16), kend = min(kbeg + per, k16);
  for (int s0 = kbeg; s0 < kbeg + per; s0 += MAXS) {
    const int ns = max(0, min(MAXS, kend - s0));
    f32x4 a0[MAXS], a1[MAXS], b[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      if (s < ns) {
        const int k = (s0 + s) * 16 + fq * 4;
        a0[s] = *reinterpret_cast<const f32x4*>(A + (int64_t)r0 * g.lda + k);
        if (two) a1[s] = *reinterpret_cast<const f32x4*>(A + (int64_t)r1 * g.lda + k);
        if constexpr (LAYOUT == D2R_GEMM_NT) {
          b[s] = *reinterpret_cast<const f32x4*>(B + (int64_t)col * g.ldb + k);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[s][j] = B[(int64_t)(k + j) * g.ldb + col];
        }
      }
    }
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      if (s < ns) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s][j], b[s][j], acc0, 0, 0, 0);
          if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s][j], b[s][j], acc1, 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[wave][0][fq * 4 + r][fr] = acc0[r];
    red[wave][1][fq * 4 + r][fr] = acc1[r];
  }
  __syncthreads();
  const int64_t cz = zb * g.sCb + zh * g.sCh, rz = zb * g.sRb + zh * g.sRh;
  float* C = reinterpret_cast<float*>(g.C);
  float* P = reinterpret_cast<float*>(g.P);
  const float* R = reinterpret_cast<const float*>(g.R);
  for (int e = tid; e < 512; e += 256) {
    const int mt = e >> 8, row = (e >> 4) & 15, c = e & 15;
    const int m = mt * 16 + row, n = n0 + c;
    if (m >= g.M || n >= g.N) continue;
    float v = (red[0][mt][row][c] + red[1][mt][row][c]) + (red[2][mt][row][c] + red[3][mt][row][c]);
    v = g.alpha * v + (g.bias ? g.bias[zb * g.sBiasB + n] : 0.f);
    const int64_t ci = cz + (int64_t)m * g.ldc + n;
    if (P) P[ci] = v;
    v = act_apply_cold(g.act, v);
    if (R) v += R[rz + (int64_t)m * g.ldr + n];
    if (g.beta != 0.f) v += g.beta * C[ci];
    C[ci] = v;
  }
}

static int g_skinny = 1;
template <int LAYOUT>
static bool skinny_f32_try(const GemmArgs& a, int batch, hipStream_t st) {
  if (!g_skinny || a.M > 32 || a.M < 1 || a.K < 64 || a.K % 16 != 0 || a.c_dtype != D2R_F32 || a.G || a.dbias || !a.vecA) return false;
  if (LAYOUT == D2R_GEMM_NT && !a.vecB) return false;
  hipLaunchKernelGGL((gemm_skinny_f32_kernel<LAYOUT>), dim3(d2r_cdiv(a.N, 16), 1, batch), dim3(256), 0, st, a);
  d2r_gemm_variant_tl = 30;
  return true;
}

// ---- short-reduction fp32 TN product: C[M,N] = alpha * A[K,M]^T B[K,N] (+ beta * C), dbias[m] += sum_k A[k,m], K <= 64 ---------------
// The weight gradients of the linears that see one row per SAMPLE (Block fusion head, routers, poolers: K = batch size) are
// rank-K updates of matrices with up to 1600 x 768 elements: an output-bandwidth problem (read-modify-write of C), which the tiled
// MFMA kernel served with 64 x 64 tiles of a K-loop that is two steps long (20 us per product, five of them on the serial stretch
// between forward and backward).  Here a workgroup owns a 64 x 64 tile of C, stages the K x 64 slices of both operands in LDS
// (coalesced 256-byte rows) and every thread accumulates a 4 x 4 block with k ascending - plain FMAs, a fixed order - then
// read-modify-writes its four float4.  Batched like the other kernels (blockIdx.z).
// T: operand type (fp32, or a 16-bit type converted on the way into LDS); GROUPED: blockIdx.z selects one of the problems of `grp`
// (d2r_gemm_tn_grouped: the deferred rank-B updates of the routing cells' per-sample linears), else a batch index.
template <typename T, bool GROUPED>
__global__ __launch_bounds__(256) void gemm_rank_tn_kernel(GemmArgs g, GemmGroup grp) {
  constexpr int KMAX = 64;
  __shared__ float As[KMAX][64 + 4], Bs[KMAX][64 + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int z = blockIdx.z;
  const T *A, *B;
  float* C;
  float* dbias = g.dbias;
  if constexpr (GROUPED) {
    A = reinterpret_cast<const T*>(grp.A[z]), B = reinterpret_cast<const T*>(grp.B[z]), C = reinterpret_cast<float*>(grp.C[z]);
    dbias = g.dbias ? grp.dbias[z] : nullptr;
  } else {
    const int zb = z / g.nh, zh = z - zb * g.nh;
    A = reinterpret_cast<const T*>(g.A) + zb * g.sAb + zh * g.sAh;
    B = reinterpret_cast<const T*>(g.B) + zb * g.sBb + zh * g.sBh;
    C = reinterpret_cast<float*>(g.C) + zb * g.sCb + zh * g.sCh;
  }
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  for (int e = tid; e < g.K * 64; e += 256) {
    const int k = e >> 6, c = e & 63;
    As[k][c] = (m0 + c < g.M) ? (float)A[(int64_t)k * g.lda + m0 + c] : 0.f;
    Bs[k][c] = (n0 + c < g.N) ? (float)B[(int64_t)k * g.ldb + n0 + c] : 0.f;
  }
  __syncthreads();
  float acc[4][4] = {};
  for (int k = 0; k < g.K; ++k) {
    const f32x4 av = *reinterpret_cast<const f32x4*>(&As[k][ty * 4]);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(&Bs[k][tx * 4]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
  }
  const bool vec = (g.ldc & 3) == 0 && n0 + tx * 4 + 4 <= g.N && (reinterpret_cast<uintptr_t>(C) & 15u) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= g.M) continue;
    float* p = C + (int64_t)m * g.ldc + n0 + tx * 4;
    if (vec) {
      f32x4 o = {g.alpha * acc[i][0], g.alpha * acc[i][1], g.alpha * acc[i][2], g.alpha * acc[i][3]};
      if (g.beta != 0.f) {
        const f32x4 old = *reinterpret_cast<const f32x4*>(p);
        o = f32x4{o[0] + g.beta * old[0], o[1] + g.beta * old[1], o[2] + g.beta * old[2], o[3] + g.beta * old[3]};
      }
      *reinterpret_cast<f32x4*>(p) = o;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + tx * 4 + j < g.N) p[j] = g.alpha * acc[i][j] + (g.beta != 0.f ? g.beta * p[j] : 0.f);
    }
  }
  if (dbias && blockIdx.x == 0 && tid < 64 && m0 + tid < g.M) {  // (ungrouped: batch 1 only, checked by the dispatcher)
    float sum = 0.f;
    for (int k = 0; k < g.K; ++k) sum += As[k][tid];
    dbias[m0 + tid] += sum;
  }
}

static bool rank_tn_f32_try(const GemmArgs& a, int batch, hipStream_t st) {
  if (a.K < 1 || a.K > 64 || a.c_dtype != D2R_F32 || a.dtype != D2R_F32 || a.G || a.R || a.P || a.bias || a.act != D2R_ACT_NONE) return false;
  if (a.dbias && batch != 1) return false;
  if ((int64_t)a.M * a.N < 4096) return false;  // (tiny outputs: the tiled kernel's 32 x 64 tiles do as well)
  static const GemmGroup no_group = {};
  hipLaunchKernelGGL((gemm_rank_tn_kernel<float, false>), dim3(d2r_cdiv(a.N, 64), d2r_cdiv(a.M, 64), batch), dim3(256), 0, st, a, no_group);
  d2r_gemm_variant_tl = 32;
  return true;
}

// ---- skinny 16-bit GEMM: M <= 32 rows of 16-bit operands (the per-sample vectors of the routing cells: GLAC's global branch,
// GESC, the cells' cls poolers and their dX) -------------------------------------------------------------------------------------
// Same decomposition as the fp32 kernel above - a workgroup owns 16 output columns, its four waves split K, all loads of a pass in
// flight before the first MFMA, partials summed through LDS in a fixed order - on v_mfma_f32_16x16x32 (a lane loads 8 consecutive
// k = one 16-byte pack per operand row).  NT reads B [N,K] rows directly; NN (B [K,N], the dX direction) stages each wave's k-rows
// of the 16 columns in LDS and takes the fragments with the transposing read.  Epilogue = the split-K reduce kernel's (bias,
// saved pre-activation, activation, activation gradient of a reference, residual, beta), 16-bit or fp32 output.  Replaces, per
// call, a 32x64-tile launch over split-K slabs plus its reduce launch (about 110 launches per training step).
template <typename E, int LAYOUT>
__device__ __forceinline__ void gemm_skinny_h16_body(const GemmArgs& g, int n0) {
  typedef typename H16<E>::v8 E8;
  typedef typename H16<E>::v4 E4;
  constexpr int MAXS = 8;  // k-steps of 32 per wave and pass: K = 1024 per pass
  constexpr bool NN = LAYOUT == D2R_GEMM_NN;
  __shared__ float red[4][2][16][17];
  __shared__ __attribute__((aligned(16))) E bs[NN ? 4 * MAXS * 32 * 16 : 8];  // NN: per wave [k of the pass][16 columns]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  const E* A = reinterpret_cast<const E*>(g.A);
  const E* B = reinterpret_cast<const E*>(g.B);
  const int col = min(n0 + fr, g.N - 1);  // clamped: the matching outputs are not stored
  const int r0 = min(fr, g.M - 1), r1 = min(16 + fr, g.M - 1);
  const bool two = g.M > 16;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int ksteps = g.K / 32, per = (ksteps + 3) / 4;
  const int kbeg = min(wave * per, ksteps), kend = min(kbeg + per, ksteps);
  E* mine = bs + (NN ? wave * (MAXS * 32 * 16) : 0);
  for (int s0 = kbeg; s0 < kbeg + per; s0 += MAXS) {  // (the same trip count in every wave: the NN staging synchronises per wave only)
    const int ns = max(0, min(MAXS, kend - s0));
    E8 a0[MAXS], a1[MAXS], b[MAXS];
    Pack<E, 8> xr[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      if (s < ns) {
        const int k = (s0 + s) * 32 + fq * 8;
        a0[s] = *reinterpret_cast<const E8*>(A + (int64_t)r0 * g.lda + k);
        if (two) a1[s] = *reinterpret_cast<const E8*>(A + (int64_t)r1 * g.lda + k);
        if constexpr (!NN) b[s] = *reinterpret_cast<const E8*>(B + (int64_t)col * g.ldb + k);
        else xr[s] = ld_pack<E, 8>(B + (int64_t)((s0 + s) * 32 + (lane >> 1)) * g.ldb + n0 + (lane & 1) * 8);  // 32 k-rows x 32 bytes
      }
    }
    if constexpr (NN) {
#pragma unroll
      for (int s = 0; s < MAXS; ++s)
        if (s < ns) st_pack<E, 8>(mine + (s * 32 + (lane >> 1)) * 16 + (lane & 1) * 8, xr[s]);
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's stores have landed (the slab is private to the wave)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {
        if (s < ns) {
          const E* p0 = mine + (s * 32 + fq * 8 + tq) * 16 + tp * 4;
          const E4 lo = H16<E>::tr_read(p0), hi = H16<E>::tr_read(p0 + 4 * 16);
          b[s] = E8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
    }
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      if (s < ns) {
        acc0 = H16<E>::mfma32(a0[s], b[s], acc0);
        if (two) acc1 = H16<E>::mfma32(a1[s], b[s], acc1);
      }
    }
    if constexpr (NN) {
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();  // the fragments are in registers before the next pass overwrites the slab
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[wave][0][fq * 4 + r][fr] = acc0[r];
    red[wave][1][fq * 4 + r][fr] = acc1[r];
  }
  __syncthreads();
  for (int e = tid; e < 512; e += 256) {
    const int mt = e >> 8, row = (e >> 4) & 15, c = e & 15;
    const int m = mt * 16 + row, n = n0 + c;
    if (m >= g.M || n >= g.N) continue;
    float v = (red[0][mt][row][c] + red[1][mt][row][c]) + (red[2][mt][row][c] + red[3][mt][row][c]);
    v = g.alpha * v + (g.bias ? g.bias[n] : 0.f);
    const int64_t ci = (int64_t)m * g.ldc + n;
    if (g.P) store_c(g.P, g.c_dtype, ci, v);
    v = act_apply_cold(g.act, v);
    if (g.G) v *= act_grad_cold(g.gact, load_c(g.G, g.c_dtype, ci));
    if (g.R) v += load_c(g.R, g.c_dtype, (int64_t)m * g.ldr + n);
    if (g.beta != 0.f) v += g.beta * load_c(g.C, g.c_dtype, ci);
    store_c(g.C, g.c_dtype, ci, v);
  }
}
template <typename E, int LAYOUT>
__global__ __launch_bounds__(256) void gemm_skinny_h16_kernel(GemmArgs g) {
  gemm_skinny_h16_body<E, LAYOUT>(g, blockIdx.x * 16);
}
// up to four INDEPENDENT skinny products in one launch (grid y = problem): the per-sample linears of different routing cells that are
// ready at the same point of a layer (text pool | image pool of GLAC's global branch and of GESC, ...) - 7 us of latency each for
// microseconds of nothing.  Every workgroup runs the body of the single launch: bit-identical results.
constexpr int D2R_SKINNY_GROUP_MAX = 4;
struct SkinnyGroup {
  GemmArgs a[D2R_SKINNY_GROUP_MAX];
};
template <typename E, int LAYOUT>
__global__ __launch_bounds__(256) void gemm_skinny_h16_group_kernel(SkinnyGroup grp) {
  const GemmArgs& g = grp.a[blockIdx.y];
  const int n0 = blockIdx.x * 16;
  if (n0 >= g.N) return;  // (the grid is as wide as the widest problem)
  gemm_skinny_h16_body<E, LAYOUT>(g, n0);
}

template <int LAYOUT>
static bool skinny_h16_ok(const GemmArgs& a, int batch) {
  if (!g_skinny || batch != 1 || a.M > 32 || a.M < 1 || a.K < 64 || a.K % 32 != 0 || a.dbias || !a.vecA || !a.vecB) return false;
  if (LAYOUT == D2R_GEMM_NN && a.N % 16 != 0) return false;  // (a 16-column slab per workgroup is loaded unguarded)
  return true;
}
template <typename E, int LAYOUT>
static bool skinny_h16_try(const GemmArgs& a, int batch, hipStream_t st) {
  if (!skinny_h16_ok<LAYOUT>(a, batch)) return false;
  hipLaunchKernelGGL((gemm_skinny_h16_kernel<E, LAYOUT>), dim3(d2r_cdiv(a.N, 16)), dim3(256), 0, st, a);
  d2r_gemm_variant_tl = 31;
  return true;
}
template <typename E, int LAYOUT>
static void skinny_h16_group_launch(const GemmArgs* probs, int n, hipStream_t st) {
  SkinnyGroup grp = {};
  int gx = 1;
  for (int i = 0; i < n; ++i) {
    grp.a[i] = probs[i];
    gx = std::max(gx, d2r_cdiv(probs[i].N, 16));
  }
  hipLaunchKernelGGL((gemm_skinny_h16_group_kernel<E, LAYOUT>), dim3(gx, n), dim3(256), 0, st, grp);
  d2r_gemm_variant_tl = 31;
}

// ---- 16-bit matrix-vector product: C[M,1] = act(alpha * A[M,K] b[K] + bias) (N = 1, e.g. the SAF scores a = S w of 4-6 thousand rows):
// a wave per row, 16-byte loads, fixed reduction order; 16-bit or fp32 output.  (18 us on the tiled kernel, whose 64-column tiles
// compute 63 columns of nothing.)
template <typename E>
__global__ __launch_bounds__(256) void gemv_h16_kernel(GemmArgs g) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= g.M) return;
  const E* a = reinterpret_cast<const E*>(g.A) + row * g.lda;
  const E* b = reinterpret_cast<const E*>(g.B);
  float acc = 0.f;
  for (int pk = lane; pk * 8 < g.K; pk += 64) {
    const Pack<E, 8> av = ld_pack<E, 8>(a + pk * 8), bv = ld_pack<E, 8>(b + pk * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += (float)av.v[j] * (float)bv.v[j];
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    float v = g.alpha * acc + (g.bias ? g.bias[0] : 0.f);
    v = act_apply_cold(g.act, v);
    store_c(g.C, g.c_dtype, row * g.ldc, v);
  }
}
template <typename E, int LAYOUT>
static bool gemv_h16_try(const GemmArgs& a, int batch, hipStream_t st) {
  if (LAYOUT != D2R_GEMM_NT || a.N != 1 || batch != 1 || a.M < 64 || a.K % 8 != 0 || !a.vecA || !d2r_aligned16(a.B)) return false;
  if (a.R || a.P || a.G || a.dbias || a.beta != 0.f) return false;
  hipLaunchKernelGGL((gemv_h16_kernel<E>), dim3((unsigned)((a.M + 3) / 4)), dim3(256), 0, st, a);
  d2r_gemm_variant_tl = 34;
  return true;
}

template <typename T, int LAYOUT>
static int launch_layout(GemmArgs a, int batch, hipStream_t st, void* ws, size_t ws_bytes) {
  if constexpr (sizeof(T) == 4 && LAYOUT != D2R_GEMM_TN) {
    if (skinny_f32_try<LAYOUT>(a, batch, st)) return d2r_check_launch("d2r_gemm(skinny)");
  }
  if constexpr (sizeof(T) == 2 && LAYOUT != D2R_GEMM_TN) {
    if (skinny_h16_try<T, LAYOUT>(a, batch, st)) return d2r_check_launch("d2r_gemm(skinny 16-bit)");
    if (gemv_h16_try<T, LAYOUT>(a, batch, st)) return d2r_check_launch("d2r_gemm(matrix-vector)");
  }
  if constexpr (sizeof(T) == 4 && LAYOUT == D2R_GEMM_TN) {
    if (rank_tn_f32_try(a, batch, st)) return d2r_check_launch("d2r_gemm(rank-K TN)");
  }
  if constexpr (sizeof(T) == 2) {
    // large bf16 shapes: LDS-DMA pipelined 128xBN kernel (forced with tile 4 = 128x128, 5 = 128x64 for A/B runs)
    int bn = 0;
    if (g_tile == 4) bn = 128;
    else if (g_tile == 5) bn = 64;
    else if (g_tile == 6) bn = 129;  // 128x128 on eight waves (A/B runs)
    else if (g_tile == 7) bn = 1128;  // software-pipelined K-loop: 128x128 on four waves
    else if (g_tile == 8) bn = 1064;  //                             128x64
    else if (g_tile == 9) bn = 1129;  //                             128x128 on eight waves
    // measured (profiles/gemm_ab_r01_e.log): the 128x64 LDS-DMA kernel beats the register-staged 64x64 tiles on every
    // NT / NN shape of the workload (351 vs 275, 548 vs 422, 616 vs 343 TFLOP/s ...); weight-gradient GEMMs that carry
    // the bias-gradient side product stay on the generic kernel
    // wide outputs (N >= 1536: FFN up-projection, fused q|k|v, FFN dX): the 128x128 tile on eight waves fetches a third
    // less from L2 per flop (NT 6304x3072x768: 561 vs 526, NN: 515 vs 452 TFLOP/s); narrow outputs keep the 128x64 tile
    // (a deep reduction amortises the wider tile's longer prologue also for narrow outputs: NT 6304x768x3072 722 vs 671 TFLOP/s,
    //  tests/probes/gemm_variants.py; not so in the NN direction, 616 vs 654)
    // Round 3: the comparison above was made with every launch ALONE on the GPU.  In the training step two branch streams keep
    // two launches in flight most of the time, the chip is saturated, and what counts is bytes per flop: with the 128x128 tile for
    // every output of at least 128 columns the step is 2 % faster (1361-1366 against 1317-1342 samples/s on one box,
    // profiles/tile_balance_r03.log; alone on the GPU a 4096x768x768 product takes 13 us with either tile).
    else if (g_tile < 0 && g_glds && LAYOUT != D2R_GEMM_TN)
      bn = a.N >= 128 ? 129 : 64;
    if (a.dbias) bn = 0;
    if constexpr (LAYOUT != D2R_GEMM_TN) {
      // 256 x 256 tiles on the deep-pipelined kernel (gemm8.hip) when the product fills the chip with them (tile code 11 forces them
      // wherever they are eligible; d2r_gemm_tuning 110 / 111 switches the automatic choice off / on: A/B runs)
      if ((g_tile == 11 || (g_tile < 0 && g_gemm8 && gemm8_pays(a))) && d2r_gemm8_fwd_ok(a, LAYOUT, batch)) {
        d2r_gemm_variant_tl = 8;
        return d2r_gemm8_fwd_launch(&a, 1, LAYOUT, st);
      }
    }
    if (bn == 129 && g_splitk && ws) {  // deep reductions over a partial round of tiles: split K inside the launch (gemm_glds.hip)
      size_t slab = 0;
      const int want = d2r_gemm_glds_splitk_plan(a, LAYOUT, batch, ws_bytes, &slab);
      if (want > 1) {
        GemmArgs b = a;
        b.splits = want, b.tiles_per_split = 0, b.ws = (float*)ws;
        b.kflags = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + ((ws_bytes - 4096) & ~(size_t)15));  // (the last 4 KiB of a workspace: zero at first use, kept zero)
        d2r_gemm_variant_tl = 3;
        return d2r_gemm_glds_splitk_launch(b, LAYOUT, st);
      }
    }
    if (bn) {
      GemmArgs b = a;
      b.ws = nullptr; b.splits = 1; b.tiles_per_split = 0;
      if (d2r_gemm_glds_try(b, LAYOUT, batch, bn, st)) return d2r_check_launch("d2r_gemm(glds)");
    }
  }
  constexpr int BK = sizeof(T) == 2 ? 64 : 16;
  const int64_t t128 = (int64_t)d2r_cdiv(a.M, 128) * d2r_cdiv(a.N, 128) * batch;
  const int64_t t12864 = (int64_t)d2r_cdiv(a.M, 128) * d2r_cdiv(a.N, 64) * batch;
  const int64_t t64 = (int64_t)d2r_cdiv(a.M, 64) * d2r_cdiv(a.N, 64) * batch;
  // tile choice, measured on MI355X with tests/bench_gemm.py (profiles/gemm_ab_r01.log): with this register-staged
  // single-buffer loop, occupancy wins — 64x64 tiles (46 VGPRs, 8 waves/SIMD) beat 128x64 / 128x128 on every NT/NN
  // shape of the workload (288-474 vs 100-350 TFLOP/s); the reduction-heavy TN weight-gradient GEMMs prefer
  // 128x128 when the output is large and 128x64 + split-K when it is 768x768.
  int tile;  // 0: 32x64, 1: 64x64, 2: 128x64, 3: 128x128
  if (a.M <= 32) tile = 0;
  else if (LAYOUT == D2R_GEMM_TN && batch == 1) tile = ((int64_t)a.M * a.N >= 1500000) ? 3 : 1;  // gemm_ab_r01_c.log
  else tile = 1;
  if (g_tile >= 0 && g_tile <= 3 && a.M > 32) tile = g_tile;
  if (a.G) {  // the activation-gradient epilogue exists in the LDS-staged (vectorised, bf16) epilogue of the small tiles only
    if (!a.vecC || a.c_dtype != H16<typename Out16<T>::type>::DT)
      return d2r_fail(D2R_ERR_INVALID, "d2r_gemm: grad_ref needs a 16-bit output (of the input type) with 16-byte aligned rows");
    if (tile > 1) tile = 1;
  }
  if (tile > 1) a.vecC = 0;  // the LDS-staged epilogue only pays on the small tiles (register pressure on the large ones)
  // deterministic split-K for reduction-heavy GEMMs with few output tiles (weight gradients)
  a.splits = 1;
  a.tiles_per_split = 0;
  a.ws = nullptr;
  const int nk = d2r_cdiv(a.K, BK);
  if (batch == 1 && ws && nk >= 8) {
    const int64_t tiles = tile == 3 ? t128 : (tile == 2 ? t12864 : (tile == 1 ? t64 : (int64_t)d2r_cdiv(a.M, 32) * d2r_cdiv(a.N, 64)));
    // Split so that ALL workgroups of the launch are resident at once: a partial second round of workgroups doubles
    // the time of these short kernels.  128-row tiles run 2 workgroups per CU (VGPR-limited) -> 512 slots: 3072x768
    // outputs (144 tiles) split 3 ways, not 4 (324 -> 385 TFLOP/s), 2304x768 (108 tiles) 4 ways, not 5 (249 -> 306).
    // 64x64 tiles: the best measured split of the 768x768 weight gradients is 6 (864 workgroups); more splits only
    // add slab traffic.  profiles/gemm_ab_r01_d.log.
    constexpr int cap_small = 896, cap_large = 512;
    const int64_t capacity = tile <= 1 ? cap_small : cap_large;
    if (tiles < capacity) {
      int want = (int)(capacity / tiles);
      if (want > 16) want = 16;
      if (want > nk / 2) want = nk / 2;
      const size_t slab_bytes = ws_bytes > 4096 ? ws_bytes - 4096 : 0;  // (the last 4 KiB hold the tile counters of the in-kernel split-K)
      while (want > 1 && (size_t)want * ((size_t)a.M * a.N + (a.dbias ? a.M : 0)) * sizeof(float) > slab_bytes) --want;
      if (want > 1) {
        a.tiles_per_split = d2r_cdiv(nk, want);
        a.splits = d2r_cdiv(nk, a.tiles_per_split);
        a.ws = (float*)ws;
      }
    }
  }
  const int gz = a.splits > 1 ? a.splits : batch;
  switch (tile) {
    case 0: launch_tile<T, LAYOUT, 32, 64, 2, 2>(a, gz, st); break;
    case 1: launch_tile<T, LAYOUT, 64, 64, 2, 2>(a, gz, st); break;
    case 2: launch_tile<T, LAYOUT, 128, 64, 2, 2>(a, gz, st); break;
    default: launch_tile<T, LAYOUT, 128, 128, 2, 2>(a, gz, st); break;
  }
  if (int rc = d2r_check_launch("d2r_gemm")) return rc;
  if (a.splits > 1) {
    int blocks = d2r_cdiv((int64_t)a.M * a.N, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
    return d2r_check_launch("d2r_gemm(split-K reduce)");
  }
  return D2R_OK;
}

template <typename T>
static int launch_dtype(const GemmArgs& a, int layout, int batch, hipStream_t st, void* ws, size_t ws_bytes) {
  switch (layout) {
    case D2R_GEMM_NT: return launch_layout<T, D2R_GEMM_NT>(a, batch, st, ws, ws_bytes);
    case D2R_GEMM_NN: return launch_layout<T, D2R_GEMM_NN>(a, batch, st, ws, ws_bytes);
    case D2R_GEMM_TN: return launch_layout<T, D2R_GEMM_TN>(a, batch, st, ws, ws_bytes);
  }
  return d2r_fail(D2R_ERR_INVALID, "d2r_gemm: bad layout %d", layout);
}

// descriptor -> kernel argument block (validation included); `batch` = nb * nh
static int gemm_desc_to_args(const d2r_gemm_desc* d, GemmArgs& a, int& batch) {
  D2R_REQUIRE(d != nullptr, "d2r_gemm: null descriptor");
  D2R_REQUIRE(d->A && d->B && d->C, "d2r_gemm: null operand");
  D2R_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, "d2r_gemm: negative size");
  D2R_REQUIRE(d->dtype == D2R_F32 || d2r_is16(d->dtype), "d2r_gemm: bad dtype %d", d->dtype);
  D2R_REQUIRE(d->c_dtype == D2R_F32 || d2r_is16(d->c_dtype), "d2r_gemm: bad c_dtype %d", d->c_dtype);
  D2R_REQUIRE(d->c_dtype == D2R_F32 || d->dtype == D2R_F32 || d->c_dtype == d->dtype,
              "d2r_gemm: a 16-bit output of 16-bit inputs has the inputs' type (dtype %d, c_dtype %d)", d->dtype, d->c_dtype);
  D2R_REQUIRE(d->nb >= 1 && d->nh >= 1, "d2r_gemm: batch must be >= 1");
  D2R_REQUIRE((int64_t)d->nb * d->nh <= 65535, "d2r_gemm: batch %lld exceeds grid.z", (long long)d->nb * d->nh);
  const bool a_kcont = d->layout != D2R_GEMM_TN, b_kcont = d->layout == D2R_GEMM_NT;
  D2R_REQUIRE(d->layout == D2R_GEMM_NT || d->layout == D2R_GEMM_NN || d->layout == D2R_GEMM_TN, "d2r_gemm: bad layout %d", d->layout);
  D2R_REQUIRE(d->lda >= (a_kcont ? d->K : d->M), "d2r_gemm: lda %lld too small", (long long)d->lda);
  D2R_REQUIRE(d->ldb >= (b_kcont ? d->K : d->N), "d2r_gemm: ldb %lld too small", (long long)d->ldb);
  D2R_REQUIRE(d->ldc >= d->N, "d2r_gemm: ldc %lld < N", (long long)d->ldc);
  D2R_REQUIRE(!d->residual || d->ldr >= d->N, "d2r_gemm: ldr too small");
  D2R_REQUIRE(!d->workspace || d2r_aligned16(d->workspace), "d2r_gemm: workspace must be 16-byte aligned");
  a = GemmArgs{};
  a.A = d->A; a.B = d->B; a.C = d->C; a.bias = d->bias; a.R = d->residual; a.P = d->preact;
  a.M = d->M; a.N = d->N; a.K = d->K; a.nh = d->nh;
  a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.ldr = d->ldr;
  a.sAb = d->sAb; a.sAh = d->sAh; a.sBb = d->sBb; a.sBh = d->sBh;
  a.sCb = d->sCb; a.sCh = d->sCh; a.sRb = d->sRb; a.sRh = d->sRh; a.sBiasB = d->s_bias_b;
  a.alpha = d->alpha; a.beta = d->beta; a.act = d->act; a.c_dtype = d->c_dtype; a.dtype = d->dtype;
  a.ws = nullptr; a.splits = 1; a.tiles_per_split = 0; a.xcd = g_xcd; a.band = 0;
  a.dbg = g_dbg;
  a.ts = nullptr;
  D2R_REQUIRE(!d->dbias || (d->layout == D2R_GEMM_TN && d->nb * d->nh == 1), "d2r_gemm: dbias needs the TN layout and batch 1");
  a.dbias = d->dbias;
  D2R_REQUIRE(!d->grad_ref || d->nb * d->nh == 1, "d2r_gemm: grad_ref needs batch 1");
  a.G = d->grad_ref;
  a.gact = d->grad_act;
  const int64_t es = (int64_t)d2r_esize(d->dtype);
  auto vec_ok = [&](const void* p, int64_t ld, int64_t sb, int64_t sh) {
    return d2r_aligned16(p) && (ld * es) % 16 == 0 && (sb * es) % 16 == 0 && (sh * es) % 16 == 0;
  };
  a.vecA = vec_ok(d->A, d->lda, d->sAb, d->sAh) ? 1 : 0;
  a.vecB = vec_ok(d->B, d->ldb, d->sBb, d->sBh) ? 1 : 0;
  {
    const int64_t cs = (int64_t)d2r_esize(d->c_dtype);
    auto cvec = [&](const void* p, int64_t ld, int64_t sb, int64_t sh) {
      return !p || (d2r_aligned16(p) && (ld * cs) % 16 == 0 && (sb * cs) % 16 == 0 && (sh * cs) % 16 == 0);
    };
    a.vecC = (g_vepi && cvec(d->C, d->ldc, d->sCb, d->sCh) && cvec(d->preact, d->ldc, d->sCb, d->sCh) &&
              cvec(d->grad_ref, d->ldc, d->sCb, d->sCh) &&
              cvec(d->residual, d->ldr, d->sRb, d->sRh)) ? 1 : 0;
  }
  batch = d->nb * d->nh;
  return D2R_OK;
}
static double gemm_desc_flops(const d2r_gemm_desc* d) { return 2.0 * d->nb * d->nh * d->M * d->N * (double)d->K; }
static double gemm_desc_bytes(const d2r_gemm_desc* d) {
  const double es = (double)d2r_esize(d->dtype), cs = (double)d2r_esize(d->c_dtype);
  return (double)d->nb * d->nh * (((double)d->M * d->K + (double)d->N * d->K) * es +
                                  (double)d->M * d->N * cs * (1 + (d->beta != 0.f) + (d->residual != nullptr) + (d->grad_ref != nullptr) + (d->preact != nullptr)));
}

extern "C" int d2r_gemm(const d2r_gemm_desc* d, void* stream) {
  GemmArgs a;
  int batch = 1;
  if (int rc = gemm_desc_to_args(d, a, batch)) return rc;
  if (d->M == 0 || d->N == 0) return D2R_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  GemmTimerScope timed(st, d->dtype * 8 + d->layout * 2, gemm_desc_flops(d), gemm_desc_bytes(d));
  if (d->dtype == D2R_BF16) return launch_dtype<bf16_t>(a, d->layout, batch, st, d->workspace, d->workspace_bytes);
  if (d->dtype == D2R_F16) return launch_dtype<f16_t>(a, d->layout, batch, st, d->workspace, d->workspace_bytes);
  return launch_dtype<float>(a, d->layout, batch, st, d->workspace, d->workspace_bytes);
}

// `n` INDEPENDENT products (no output of one is an operand or an output of another) enqueued together: the 16-bit forward / dX
// products of one layout that fit the 128 x 128 LDS-DMA tiles leave as grouped launches of up to 16 problems (gemm_glds.hip:
// every tile is computed exactly as in a launch of its own - bit-identical to n d2r_gemm calls); everything else is launched one
// by one, in the order given.
int d2r_gemm_glds_group_ok(const GemmArgs& a, int layout, int batch);                           // gemm_glds.hip
int d2r_gemm_glds_group_launch(const GemmArgs* probs, int n, int layout, hipStream_t st);
extern "C" int d2r_gemm_group(const d2r_gemm_desc* descs, int n, void* stream) {
  D2R_REQUIRE(n >= 0 && (n == 0 || descs), "d2r_gemm_group: null descriptor array");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  std::vector<GemmArgs> args(n);
  std::vector<char> grouped(n, 0);
  for (int layout : {D2R_GEMM_NT, D2R_GEMM_NN}) {
    for (int dtype : {D2R_BF16, D2R_F16}) {
      std::vector<int> idx;
      for (int i = 0; i < n; ++i) {
        const d2r_gemm_desc& d = descs[i];
        if (!g_group || g_tile >= 0 || d.layout != layout || d.dtype != dtype || d.M == 0 || d.N == 0) continue;
        int batch = 1;
        if (int rc = gemm_desc_to_args(&d, args[i], batch)) return rc;
        if (d2r_gemm_glds_group_ok(args[i], layout, batch) && !(d.M <= 32)) idx.push_back(i);
      }
      if (idx.size() < 2) continue;  // (a single problem: its own launch, with its column-band tile order)
      for (size_t first = 0; first < idx.size(); first += 16) {
        const int m = (int)std::min<size_t>(16, idx.size() - first);
        std::vector<GemmArgs> probs(m);
        double fl = 0, by = 0;
        for (int k = 0; k < m; ++k) {
          const int i = idx[first + k];
          probs[k] = args[i], grouped[i] = 1;
          fl += gemm_desc_flops(&descs[i]), by += gemm_desc_bytes(&descs[i]);
        }
        GemmTimerScope timed(st, dtype * 8 + layout * 2, fl, by);
        if (int rc = d2r_gemm_glds_group_launch(probs.data(), m, layout, st)) return rc;
      }
    }
  }
  // the per-sample products (at most 32 rows): up to four of one layout and type per launch
  for (int layout : {D2R_GEMM_NT, D2R_GEMM_NN}) {
    for (int dtype : {D2R_BF16, D2R_F16}) {
      std::vector<int> idx;
      for (int i = 0; i < n; ++i) {
        const d2r_gemm_desc& d = descs[i];
        if (grouped[i] || !g_group || g_tile >= 0 || d.layout != layout || d.dtype != dtype || d.M == 0 || d.N == 0 || d.M > 32) continue;
        int batch = 1;
        if (int rc = gemm_desc_to_args(&d, args[i], batch)) return rc;
        if (layout == D2R_GEMM_NT ? skinny_h16_ok<D2R_GEMM_NT>(args[i], batch) : skinny_h16_ok<D2R_GEMM_NN>(args[i], batch)) idx.push_back(i);
      }
      if (idx.size() < 2) continue;
      for (size_t first = 0; first + 1 < idx.size(); first += D2R_SKINNY_GROUP_MAX) {  // (a last problem alone: its own launch below)
        const int m = (int)std::min<size_t>(D2R_SKINNY_GROUP_MAX, idx.size() - first);
        GemmArgs probs[D2R_SKINNY_GROUP_MAX];
        double fl = 0, by = 0;
        for (int k = 0; k < m; ++k) {
          const int i = idx[first + k];
          probs[k] = args[i], grouped[i] = 1;
          fl += gemm_desc_flops(&descs[i]), by += gemm_desc_bytes(&descs[i]);
        }
        GemmTimerScope timed(st, dtype * 8 + layout * 2, fl, by);
        if (dtype == D2R_F16) {
          if (layout == D2R_GEMM_NT) skinny_h16_group_launch<f16_t, D2R_GEMM_NT>(probs, m, st);
          else skinny_h16_group_launch<f16_t, D2R_GEMM_NN>(probs, m, st);
        } else {
          if (layout == D2R_GEMM_NT) skinny_h16_group_launch<bf16_t, D2R_GEMM_NT>(probs, m, st);
          else skinny_h16_group_launch<bf16_t, D2R_GEMM_NN>(probs, m, st);
        }
        if (int rc = d2r_check_launch("d2r_gemm_group(skinny)")) return rc;
      }
    }
  }
  for (int i = 0; i < n; ++i)
    if (!grouped[i])
      if (int rc = d2r_gemm(&descs[i], stream)) return rc;
  return D2R_OK;
}

// ---- grouped weight gradients -------------------------------------------------------------------------------
// `count` TN GEMMs of ONE shape, C_i (fp32) = beta * C_i + A_i^T B_i (+ dbias_i[m] += sum_k A_i[k,m]), in launches of up
// to 16 problems (blockIdx.z = problem).  A 768x768 weight gradient alone has 144 output tiles and needs split-K,
// fp32 slabs and a reduce launch to fill the chip (175 TFLOP/s); sixteen of them give 2304 workgroups that each
// run the whole reduction (356 TFLOP/s, profiles/gemm_grouped_r01.log).  Nothing in the backward pass reads a weight
// gradient, so the caller may defer these products and launch them together.
int d2r_gemm_glds_wgrad_try(const GemmArgs& a, const GemmGroup& grp, int n, hipStream_t st);  // gemm_glds.hip

template <typename T>
static int launch_grouped_tn(const GemmArgs& base, const void* const* A, const void* const* B, float* const* C,
                             float* const* dbias, int count, hipStream_t st) {
  for (int first = 0; first < count; first += D2R_GEMM_GROUP_MAX) {
    const int n = count - first < D2R_GEMM_GROUP_MAX ? count - first : D2R_GEMM_GROUP_MAX;
    GemmGroup grp = {};
    for (int i = 0; i < n; ++i) {
      grp.A[i] = A[first + i], grp.B[i] = B[first + i], grp.C[i] = C[first + i];
      grp.dbias[i] = dbias ? dbias[first + i] : nullptr;
    }
    GemmArgs a = base;
    a.dbias = dbias ? reinterpret_cast<float*>(1) : nullptr;  // per-problem pointer substituted in the kernel; non-null enables the path
    if (sizeof(T) == 2 && g_wgrad_glds) {  // 128x128 LDS-DMA tiles (half the L2 bytes per flop of the 64x64 tiles below)
      GemmArgs b = a;
      b.xcd = g_xcd;
      if (d2r_gemm_glds_wgrad_try(b, grp, n, st)) {
        if (int rc = d2r_check_launch("d2r_gemm_tn_grouped(glds)")) return rc;
        continue;
      }
    }
    dim3 grid(d2r_cdiv(a.N, 64), d2r_cdiv(a.M, 64), n);
    if (a.K >= 1 && a.K <= 64 && (int64_t)a.M * a.N >= 4096) {
      // reduction over one row per SAMPLE (pooled-vector linears of the routing cells): rank-K updates, see gemm_rank_tn_kernel
      d2r_gemm_variant_tl = 33;
      hipLaunchKernelGGL((gemm_rank_tn_kernel<T, true>), grid, dim3(256), 0, st, a, grp);
      if (int rc = d2r_check_launch("d2r_gemm_tn_grouped(rank-K)")) return rc;
      continue;
    }
    // Operand strips fetched past L2 per round of resident workgroups (profiles/gemm_grouped_pmc_r01.txt): with whole
    // problems per XCD a round covers (resident / grid.x) row strips + grid.x column strips - right for 12 tile columns
    // (1.5x the operand bytes fetched vs 4.5x), wrong for 48 (6.3x): there the plain round-robin order, which gives an
    // XCD every eighth tile column, fetches 2.5x.
    if (grid.x >= 24 && grid.x % 8 == 0) a.xcd = 0;
    d2r_gemm_variant_tl = 21;
    hipLaunchKernelGGL((gemm_kernel<T, D2R_GEMM_TN, 64, 64, 2, 2, 1, true>), grid, dim3(256), 0, st, a, grp);
    if (int rc = d2r_check_launch("d2r_gemm_tn_grouped")) return rc;
  }
  return D2R_OK;
}

static int tn_grouped_same(int dtype, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, const void* const* h_A, const void* const* h_B,
                           float* const* h_C, float* const* h_dbias, int count, float beta, void* stream, bool allow8) {
  D2R_REQUIRE(h_A && h_B && h_C && count >= 0, "d2r_gemm_tn_grouped: null pointer array");
  D2R_REQUIRE(dtype == D2R_F32 || d2r_is16(dtype), "d2r_gemm_tn_grouped: bad dtype %d", dtype);
  D2R_REQUIRE(M >= 1 && N >= 1 && K >= 0 && lda >= M && ldb >= N && ldc >= N, "d2r_gemm_tn_grouped: bad shape");
  if (count == 0) return D2R_OK;
  const int64_t es = (int64_t)d2r_esize(dtype);
  GemmArgs a = {};
  a.M = M, a.N = N, a.K = K, a.nh = 1, a.splits = 1, a.lda = lda, a.ldb = ldb, a.ldc = ldc;
  a.alpha = 1.f, a.beta = beta, a.act = D2R_ACT_NONE, a.c_dtype = D2R_F32, a.dtype = dtype, a.xcd = g_xcd, a.grouped = 1, a.dbg = 0, a.band = 0;
  a.vecA = (lda * es) % 16 == 0, a.vecB = (ldb * es) % 16 == 0, a.vecC = 0;
  for (int i = 0; i < count; ++i) {
    D2R_REQUIRE(h_A[i] && h_B[i] && h_C[i] && (!h_dbias || h_dbias[i]), "d2r_gemm_tn_grouped: null operand in problem %d", i);
    a.vecA &= d2r_aligned16(h_A[i]) ? 1 : 0;
    a.vecB &= d2r_aligned16(h_B[i]) ? 1 : 0;
    // the problems of one launch run in parallel and each workgroup does a non-atomic C = beta*C + A^T B (dbias += ...):
    // two problems sharing an output would race
    for (int j = 0; j < i; ++j) {
      D2R_REQUIRE(h_C[i] != h_C[j], "d2r_gemm_tn_grouped: problems %d and %d write the same C", j, i);
      D2R_REQUIRE(!h_dbias || h_dbias[i] != h_dbias[j], "d2r_gemm_tn_grouped: problems %d and %d write the same dbias", j, i);
    }
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // algorithmic bytes: both operands once, the fp32 sink read and written
  GemmTimerScope timed(st, dtype * 8 + D2R_GEMM_TN * 2 + 1, 2.0 * count * M * N * (double)K,
                       (double)count * (((double)K * M + (double)K * N) * es + 2.0 * M * N * 4.0));
  if (allow8 && g_gemm8_wgrad && d2r_is16(dtype) && (int64_t)count * d2r_cdiv(M, 256) * d2r_cdiv(N, 256) >= 96) {
    bool ok = true;
    for (int i = 0; i < count && ok; ++i) ok = d2r_gemm8_wgrad_ok(dtype, M, N, K, lda, ldb, ldc, h_A[i], h_B[i], h_C[i]) != 0;
    if (ok) {
      std::vector<int> Ms(count, M), Ns(count, N), Ks(count, K);
      std::vector<int64_t> la(count, lda), lb(count, ldb), lc(count, ldc);
      d2r_gemm_variant_tl = 28;
      return d2r_gemm8_wgrad_launch(dtype, count, Ms.data(), Ns.data(), Ks.data(), la.data(), lb.data(), lc.data(), h_A, h_B, h_C, h_dbias, beta, st);
    }
  }
  if (dtype == D2R_BF16) return launch_grouped_tn<bf16_t>(a, h_A, h_B, h_C, h_dbias, count, st);
  if (dtype == D2R_F16) return launch_grouped_tn<f16_t>(a, h_A, h_B, h_C, h_dbias, count, st);
  return launch_grouped_tn<float>(a, h_A, h_B, h_C, h_dbias, count, st);
}

extern "C" int d2r_gemm_tn_grouped(int dtype, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, const void* const* h_A,
                                   const void* const* h_B, float* const* h_C, float* const* h_dbias, int count, float beta, void* stream) {
  return tn_grouped_same(dtype, M, N, K, lda, ldb, ldc, h_A, h_B, h_C, h_dbias, count, beta, stream, true);
}


// Weight gradients of DIFFERENT shapes in one call: problem i is C_i[M_i,N_i] (fp32, ldc_i) = beta * C_i + A_i^T B_i (+ dbias_i).  The
// problems that fit the 256-wide deep-pipelined kernel (gemm8.hip) leave together, as launches of up to 40 problems whose tiles fill
// the chip whatever the single shapes are (a 768 x 768 gradient alone has 9 such tiles); the others go through d2r_gemm_tn_grouped
// shape class by shape class.  No two problems of a call may share an output.
extern "C" int d2r_gemm_tn_grouped_v(int dtype, int count, const int* M, const int* N, const int* K, const int64_t* lda, const int64_t* ldb,
                                     const int64_t* ldc, const void* const* h_A, const void* const* h_B, float* const* h_C,
                                     float* const* h_dbias, float beta, void* stream) {
  D2R_REQUIRE(count >= 0 && (count == 0 || (M && N && K && lda && ldb && ldc && h_A && h_B && h_C)), "d2r_gemm_tn_grouped_v: null array");
  D2R_REQUIRE(dtype == D2R_F32 || d2r_is16(dtype), "d2r_gemm_tn_grouped_v: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  std::vector<int> wide, rest;
  for (int i = 0; i < count; ++i) {
    D2R_REQUIRE(h_A[i] && h_B[i] && h_C[i] && (!h_dbias || h_dbias[i]), "d2r_gemm_tn_grouped_v: null operand in problem %d", i);
    D2R_REQUIRE(M[i] >= 1 && N[i] >= 1 && K[i] >= 0 && lda[i] >= M[i] && ldb[i] >= N[i] && ldc[i] >= N[i], "d2r_gemm_tn_grouped_v: bad shape in problem %d", i);
    for (int j = 0; j < i; ++j) {
      D2R_REQUIRE(h_C[i] != h_C[j], "d2r_gemm_tn_grouped_v: problems %d and %d write the same C", j, i);
      D2R_REQUIRE(!h_dbias || h_dbias[i] != h_dbias[j], "d2r_gemm_tn_grouped_v: problems %d and %d write the same dbias", j, i);
    }
    const bool w = g_gemm8_wgrad && d2r_gemm8_wgrad_ok(dtype, M[i], N[i], K[i], lda[i], ldb[i], ldc[i], h_A[i], h_B[i], h_C[i]);
    (w ? wide : rest).push_back(i);
  }
  int64_t wtiles = 0;
  for (int i : wide) wtiles += (int64_t)d2r_cdiv(M[i], 256) * d2r_cdiv(N[i], 256);
  if (wtiles < 96) {  // too few wide tiles to occupy the chip: the 128-wide grouped kernel, class by class
    rest.insert(rest.end(), wide.begin(), wide.end());
    wide.clear();
  }
  if (!wide.empty()) {
    const int n = (int)wide.size();
    std::vector<int> Ms(n), Ns(n), Ks(n);
    std::vector<int64_t> la(n), lb(n), lc(n);
    std::vector<const void*> As(n), Bs(n);
    std::vector<float*> Cs(n), Ds(n);
    double flops = 0, bytes = 0;
    for (int k = 0; k < n; ++k) {
      const int i = wide[k];
      Ms[k] = M[i], Ns[k] = N[i], Ks[k] = K[i], la[k] = lda[i], lb[k] = ldb[i], lc[k] = ldc[i], As[k] = h_A[i], Bs[k] = h_B[i], Cs[k] = h_C[i];
      Ds[k] = h_dbias ? h_dbias[i] : nullptr;
      flops += 2.0 * M[i] * N[i] * (double)K[i];
      bytes += ((double)K[i] * M[i] + (double)K[i] * N[i]) * 2.0 + (beta != 0.f ? 2.0 : 1.0) * M[i] * N[i] * 4.0;
    }
    GemmTimerScope timed(st, dtype * 8 + D2R_GEMM_TN * 2 + 1, flops, bytes);
    d2r_gemm_variant_tl = 28;
    if (int rc = d2r_gemm8_wgrad_launch(dtype, n, Ms.data(), Ns.data(), Ks.data(), la.data(), lb.data(), lc.data(), As.data(), Bs.data(), Cs.data(),
                                        h_dbias ? Ds.data() : nullptr, beta, st))
      return rc;
  }
  std::vector<char> done(count, 0);
  for (size_t a = 0; a < rest.size(); ++a) {
    const int i = rest[a];
    if (done[i]) continue;
    std::vector<const void*> As, Bs;
    std::vector<float*> Cs, Ds;
    for (size_t b = a; b < rest.size(); ++b) {
      const int j = rest[b];
      if (done[j] || M[j] != M[i] || N[j] != N[i] || K[j] != K[i] || lda[j] != lda[i] || ldb[j] != ldb[i] || ldc[j] != ldc[i]) continue;
      done[j] = 1;
      As.push_back(h_A[j]), Bs.push_back(h_B[j]), Cs.push_back(h_C[j]), Ds.push_back(h_dbias ? h_dbias[j] : nullptr);
    }
    if (int rc = tn_grouped_same(dtype, M[i], N[i], K[i], lda[i], ldb[i], ldc[i], As.data(), Bs.data(), Cs.data(), h_dbias ? Ds.data() : nullptr,
                                 (int)As.size(), beta, stream, false))  // (decided above: these did not qualify for the wide tiles)
      return rc;
  }
  return D2R_OK;
}


// ---- grouped + batched TN products with a 16-bit output (internal: attention.hip's cross-attention backward) -----------------
// C_g[b] (16-bit [m_store, N], ldc, batch stride sCb) = A_g[b]^T B_g[b],  A_g[b] [K, M] (lda, sAb), B_g[b] [K, N] (ldb, sBb) for
// g < ngroups, b < nb in ONE launch of the LDS-DMA kernel (mode 2).  M is the multiple of 8 the A rows are loaded up to
// (the padded key count of P / dS), m_store <= M the rows written.  K arbitrary.  Falls back to one batched d2r_gemm per group.
int d2r_gemm_glds_batched_tn_try(const GemmArgs& a, const GemmGroup& grp, int ngroups, hipStream_t st);  // gemm_glds.hip
int d2r_gemm_tn_batched16(int dtype, int M, int m_store, int N, int K, int64_t lda, int64_t sAb, int64_t ldb, int64_t sBb, int64_t ldc,
                          int64_t sCb, const void* const* A, const void* const* B, void* const* C, int ngroups, int nb, void* stream) {
  D2R_REQUIRE(d2r_is16(dtype) && A && B && C && ngroups >= 1 && ngroups <= D2R_GEMM_GROUP_MAX && nb >= 1, "d2r_gemm_tn_batched16: bad arguments");
  D2R_REQUIRE(m_store >= 1 && m_store <= M && N >= 1 && K >= 1 && lda >= M && ldb >= N && ldc >= N, "d2r_gemm_tn_batched16: bad shape");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  GemmArgs a = {};
  a.M = M, a.m_store = m_store, a.N = N, a.K = K, a.nh = 1, a.splits = 1, a.lda = lda, a.ldb = ldb, a.ldc = ldc, a.sAb = sAb, a.sBb = sBb, a.sCb = sCb;
  a.alpha = 1.f, a.beta = 0.f, a.act = D2R_ACT_NONE, a.c_dtype = dtype, a.dtype = dtype, a.xcd = g_xcd, a.grouped = 1, a.gbatch = nb;
  auto ok16 = [](const void* p, int64_t ld, int64_t sb) { return d2r_aligned16(p) && (ld * 2) % 16 == 0 && (sb * 2) % 16 == 0; };
  a.vecA = a.vecB = a.vecC = 1;
  GemmGroup grp = {};
  for (int i = 0; i < ngroups; ++i) {
    D2R_REQUIRE(A[i] && B[i] && C[i], "d2r_gemm_tn_batched16: null operand in group %d", i);
    a.vecA &= ok16(A[i], lda, sAb) ? 1 : 0, a.vecB &= ok16(B[i], ldb, sBb) ? 1 : 0, a.vecC &= ok16(C[i], ldc, sCb) ? 1 : 0;
    grp.A[i] = A[i], grp.B[i] = B[i], grp.C[i] = C[i];
  }
  {
    GemmTimerScope timed(st, dtype * 8 + D2R_GEMM_TN * 2 + 1, 2.0 * ngroups * nb * (double)m_store * N * K,
                         (double)ngroups * nb * (((double)K * m_store + (double)K * N) * 2.0 + (double)m_store * N * 2.0));
    if (g_glds && d2r_gemm_glds_batched_tn_try(a, grp, ngroups, st)) return d2r_check_launch("d2r_gemm_tn_batched16(glds)");
  }
  for (int i = 0; i < ngroups; ++i) {  // not eligible (alignment): the generic batched kernel, one launch per group
    d2r_gemm_desc d = {};
    d.dtype = d.c_dtype = dtype, d.layout = D2R_GEMM_TN, d.act = D2R_ACT_NONE, d.M = m_store, d.N = N, d.K = K, d.nb = nb, d.nh = 1, d.alpha = 1.f;
    d.A = A[i], d.lda = lda, d.sAb = sAb, d.B = B[i], d.ldb = ldb, d.sBb = sBb, d.C = C[i], d.ldc = ldc, d.sCb = sCb;
    if (int rc = d2r_gemm(&d, stream)) return rc;
  }
  return D2R_OK;
}
