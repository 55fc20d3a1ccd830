// gemm8.hip — deep-pipelined 256 x 256 x 64 MFMA GEMM for gfx950 (16-bit operands, fp32 accumulation), one workgroup of eight waves
// per CU (128 KiB of LDS).  Serves the products of the path that are large enough to fill the chip with 256-wide tiles:
//   MODE 1: grouped weight gradients dW_i (fp32) = beta * dW_i + dY_i^T X_i (+ bias gradients) of DIFFERENT shapes in one launch
//           (reference: autograd of every nn.Linear of models/modeling_unimo.py:334-470, models/Cells.py, models/Refinement.py,
//           models/SelfAttention.py, models/XModules.py: the deferred products of a whole branch leave as a few launches of a
//           thousand tiles each);
//   MODE 0: forward (NT) and dX (NN) products with the full 16-bit epilogue of gemm_args.h (bias, activation, saved pre-activation,
//           activation gradient of a reference, residual, beta), one problem or a group of independent problems per launch.
//
// Structure (cdna_hip_programming.md section 5, "The 256^2 8-phase template"; written from its description, not from its source):
//   * a K-tile (64 deep) of both operands is staged as FOUR 16 KiB units A0 | B0 | B1 | A1 by LDS-DMA (global_load_lds_dwordx4, linear
//     LDS image, XOR swizzle applied to the SOURCE address and to the read address); two K-tile buffers.  Unit A_h holds rows
//     {wr*128 + h*64 + 0..63} (wr = 0, 1), unit B_h columns {wc*64 + h*32 + 0..31} (wc = 0..3): every wave reads its own quarter of
//     every unit, and a unit is free again as soon as ONE phase has consumed it;
//   * a K-tile is four PHASES of 16 MFMAs per wave (one quadrant of the wave's 128 x 64 output over K = 64): phase 1 reads A0 and B0,
//     phase 2 B1, phase 3 A1, phase 4 nothing (the B0 fragments stay in registers); every phase issues the two DMA instructions of ONE
//     unit 1 1/4 to 1 3/4 K-tiles ahead, waits with a COUNTED vmcnt (four units stay in flight; never 0 in the steady state) and runs
//       reads + DMA issue + vmcnt | s_barrier | lgkmcnt(0) | 16 MFMAs at raised priority | s_barrier;
//   * waves 4-7 run ONE barrier behind waves 0-3 (the two waves of a SIMD are one of each group): while one group multiplies, the
//     other one reads LDS and issues DMA, so the matrix pipe of a SIMD alternates between its two waves instead of idling while both
//     load.  The hazards are placed by the barrier count, not by clean runs:
//       RAW  a unit is read one phase AFTER the phase whose vmcnt retired it (every wave waits for its own pieces before its first
//            barrier of that phase; the reader has passed a barrier that every issuer reached after its wait);
//       WAR  a unit is re-staged at least TWO phases after the phase that read it (the reads of phase q are complete behind the
//            lgkmcnt(0) after the first barrier of q; an issuer of phase q+2 has passed a barrier the reader reached after that).
// k-contiguous operands (A of NT / NN, B of NT) are read with ds_read_b128, k-strided ones (B of NN, both of TN) with the
// transposing ds_read_b64_tr_b16; all LDS reads are inline asm (hipcc drains every in-flight LDS-DMA in front of an LDS read it
// can see).  Rows / columns past the matrix edge are clamped on load and masked on store; a ragged reduction length (TN only:
// token rows) reads zeros for A from a zero page and clamped rows for B.
#include "gemm_args.h"

#ifndef D2R_G8_STAMPS  // 1: the kernels carry cycle stamps (tests/probes/gemm8_probe.py stamps; D2R_G8_STAMPS=1 python -m d2r_amd.build)
#define D2R_G8_STAMPS 0
#endif

__device__ __attribute__((aligned(256))) unsigned char d2r_g8_zero_page[256];

namespace {

constexpr int G8_UNIT = 16384;
constexpr int G8_A0 = 0, G8_B0 = G8_UNIT, G8_B1 = 2 * G8_UNIT, G8_A1 = 3 * G8_UNIT, G8_BUF = 4 * G8_UNIT;
constexpr int G8_SMEM = 2 * G8_BUF;

// XOR mask (16-byte chunks) of k-row k of a k-strided unit image [64 k][256 B]: whole chunk PAIRS move, by (bits 0-1, bit 3) of k
// (gemm_glds.hip kswz<16>: the eight k-rows of one transposing read land on eight different 8-bank groups)
__device__ __forceinline__ int g8_kswz(int k) { return ((k & 3) | ((k >> 1) & 4)) << 1; }

template <int OFF, typename V8>
__device__ __forceinline__ V8 g8_read128(unsigned addr) {
  V8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF, typename V4>
__device__ __forceinline__ V4 g8_read_tr(unsigned addr) {
  V4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

__device__ __forceinline__ void g8_barrier() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void g8_lgkm0() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// all but the `units` youngest staging units (two DMA instructions each) of this wave have landed
__device__ __forceinline__ void g8_wait_units(int units) {
  if (units >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (units == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (units == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (units == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// ---- main loop: acc[mi][nj] (mi = mh*4 + i, nj = nh*2 + j) of this wave's 128 x 64 block of the tile at (m0, n0) ------------------
// acc[mi][nj][r] = C[m0 + wr*128 + mh*64 + i*16 + fr][n0 + wc*64 + nh*32 + j*16 + fq*4 + r]   (fr = lane & 15, fq = lane >> 4)
// BIAS (TN): accb[e][*] = sum_k A[k, m] for m = m0 + wr*128 + (wc>>1)*64 + ((wc&1)*2 + e)*16 + fr, when do_bias
// The DMA instructions of a phase are issued in its LOAD half (between the fragment reads and the first barrier).  Issuing them inside
// the phase's MFMA cluster instead (behind the 3rd and the 9th MFMA, the counted waits then seeing one unit fewer in flight) was built
// and measured slower on every shape: an LDS-DMA instruction holds the wave's issue for ~50 cycles wherever it sits, and inside the
// cluster that lengthens the half both wave groups wait for (NT 4096^3 1299 against 1360 TFLOP/s, grouped weight gradients 883-928
// against 971-1001: profiles/gemm8_probe_r04.log, "v1").
// stamps: null, or (measurement builds) 64 s_memtime stamps per wave for workgroup 0: see G8_STAMP
template <typename E, int LAYOUT, bool BIAS>
__device__ __forceinline__ void g8_main(const E* __restrict__ A, const E* __restrict__ B, int M, int N, int K, int64_t lda, int64_t ldb,
                                        int m0, int n0, unsigned char* smem, f32x4 (&acc)[8][4], f32x4 (&accb)[2], bool do_bias,
                                        unsigned long long* stamps) {
  typedef typename H16<E>::v8 h8;
  typedef typename H16<E>::v4 h4;
  constexpr bool A_KC = LAYOUT != D2R_GEMM_TN, B_KC = LAYOUT == D2R_GEMM_NT;
  constexpr bool RAGGED = LAYOUT == D2R_GEMM_TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  const int nk = (K + 63) >> 6;
  const int krem = K - (nk - 1) * 64;  // valid k-rows of the last tile (64 unless ragged)

  // ---- per-lane source pointers of this wave's DMA pieces (unit h, piece i: instruction wave + 8*i of the unit's sixteen) -------------
  const E* pa[2][2];
  const E* pb[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ins = wave + 8 * i;
      if constexpr (A_KC) {  // 8 rows x 128 B per instruction
        const int lr = ins * 8 + (lane >> 3), c = (lane & 7) ^ (lr & 7);
        const int grow = min(m0 + (lr >> 6) * 128 + h * 64 + (lr & 63), M - 1);
        pa[h][i] = A + (int64_t)grow * lda + c * 8;
      } else {  // 4 k-rows x 256 B per instruction
        const int krow = ins * 4 + (lane >> 4), c = (lane & 15) ^ g8_kswz(krow);
        const int gcol = min(m0 + (c >> 3) * 128 + h * 64 + (c & 7) * 8, M - 8);
        pa[h][i] = A + (int64_t)krow * lda + gcol;
      }
      if constexpr (B_KC) {
        const int lr = ins * 8 + (lane >> 3), c = (lane & 7) ^ (lr & 7);
        const int gn = min(n0 + (lr >> 5) * 64 + h * 32 + (lr & 31), N - 1);
        pb[h][i] = B + (int64_t)gn * ldb + c * 8;
      } else {
        const int krow = ins * 4 + (lane >> 4), c = (lane & 15) ^ g8_kswz(krow);
        const int gcol = min(n0 + (c >> 2) * 64 + h * 32 + (c & 3) * 8, N - 8);
        pb[h][i] = B + (int64_t)krow * ldb + gcol;
      }
    }
  const int64_t strideA = A_KC ? 64 : 64 * lda, strideB = B_KC ? 64 : 64 * ldb;  // elements per K-tile

  // issue the two pieces of one unit of K-tile `tile` into buffer `buf`, advance the pointers
  auto issue = [&](const E* (&p)[2], int unit_off, int buf, int tile, bool isA, int64_t stride, int64_t ld) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const E* src = p[i];
      if constexpr (RAGGED) {
        if (tile == nk - 1 && krem < 64) {  // (uniform) rows past the reduction length: zeros for A, the last valid row for B
          const int krow = (wave + 8 * i) * 4 + (lane >> 4);
          if (krow >= krem) src = isA ? reinterpret_cast<const E*>(d2r_g8_zero_page) : src - (int64_t)(krow - (krem - 1)) * ld;
        }
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + buf * G8_BUF + unit_off + (wave + 8 * i) * 1024), 16, 0, 0);
      p[i] += stride;
    }
  };
  auto issueA = [&](int h, int buf, int tile) { issue(pa[h], h ? G8_A1 : G8_A0, buf, tile, true, strideA, lda); };
  auto issueB = [&](int h, int buf, int tile) { issue(pb[h], h ? G8_B1 : G8_B0, buf, tile, false, strideB, ldb); };

  // ---- per-lane LDS read addresses (buffer 0; the unit offsets and the fragment offsets are instruction immediates) -------------------
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
  unsigned ra[4], rb[2];  // k-contiguous: [0], [1] = the two 32-deep halves; k-strided: one per 16-wide fragment
  if constexpr (A_KC) {
    ra[0] = sbase + (wr * 64 + fr) * 128 + ((fq ^ (fr & 7)) << 4);
    ra[1] = ra[0] ^ 64;
    ra[2] = ra[3] = 0;
  } else {
    const int s = (tq | ((fq & 1) << 2)) << 1;  // g8_kswz of this lane's k-rows (the same for k and k + 4, for both halves of the tile)
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = sbase + (fq * 8 + tq) * 256 + (((wr * 8 + i * 2 + (tp >> 1)) ^ s) << 4) + (tp & 1) * 8;
  }
  if constexpr (B_KC) {
    rb[0] = sbase + (wc * 32 + fr) * 128 + ((fq ^ (fr & 7)) << 4);
    rb[1] = rb[0] ^ 64;
  } else {
    const int s = (tq | ((fq & 1) << 2)) << 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) rb[j] = sbase + (fq * 8 + tq) * 256 + (((wc * 4 + j * 2 + (tp >> 1)) ^ s) << 4) + (tp & 1) * 8;
  }

  h8 af[4][2], bf0[2][2], bf1[2][2];
  // fragment reads of one A unit (4 fragments x 2 halves) / one B unit (2 x 2) at unit offset UO of the buffer at `bo` bytes
#define G8_READ_A(UO)                                                                                    \
  do {                                                                                                   \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int i = 0; i < 4; ++i) {     \
      if constexpr (A_KC) {                                                                              \
        af[i][kk] = (i == 0)   ? g8_read128<(UO) + 0 * 2048, h8>(ra[kk] + bo)                             \
                    : (i == 1) ? g8_read128<(UO) + 1 * 2048, h8>(ra[kk] + bo)                             \
                    : (i == 2) ? g8_read128<(UO) + 2 * 2048, h8>(ra[kk] + bo)                             \
                               : g8_read128<(UO) + 3 * 2048, h8>(ra[kk] + bo);                            \
      } else {                                                                                           \
        const h4 lo = kk ? g8_read_tr<(UO) + 8192, h4>(ra[i] + bo) : g8_read_tr<(UO), h4>(ra[i] + bo);    \
        const h4 hi = kk ? g8_read_tr<(UO) + 8192 + 1024, h4>(ra[i] + bo) : g8_read_tr<(UO) + 1024, h4>(ra[i] + bo); \
        af[i][kk] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);                             \
      }                                                                                                  \
    }                                                                                                    \
  } while (0)
#define G8_READ_B(UO, BF)                                                                                \
  do {                                                                                                   \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int j = 0; j < 2; ++j) {     \
      if constexpr (B_KC) {                                                                              \
        BF[j][kk] = (j == 0) ? g8_read128<(UO), h8>(rb[kk] + bo) : g8_read128<(UO) + 2048, h8>(rb[kk] + bo); \
      } else {                                                                                           \
        const h4 lo = kk ? g8_read_tr<(UO) + 8192, h4>(rb[j] + bo) : g8_read_tr<(UO), h4>(rb[j] + bo);    \
        const h4 hi = kk ? g8_read_tr<(UO) + 8192 + 1024, h4>(rb[j] + bo) : g8_read_tr<(UO) + 1024, h4>(rb[j] + bo); \
        BF[j][kk] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);                             \
      }                                                                                                  \
    }                                                                                                    \
  } while (0)
  // the 16 MFMAs of quadrant (MH, NH); operands swapped: the accumulator tile is C^T, a lane owns ONE row and 4 consecutive columns
#define G8_MFMA(MH, NH, BF)                                                                              \
  do {                                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int i = 0; i < 4; ++i)       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
            acc[(MH) * 4 + i][(NH) * 2 + j] = H16<E>::mfma32(BF[j][kk], af[i][kk], acc[(MH) * 4 + i][(NH) * 2 + j]); \
    __builtin_amdgcn_s_setprio(0);                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  } while (0)
  // measurement builds: stamp number idx (wave-uniform, < 64) of this wave -> lane idx of a register pair, stored at the end
  unsigned st_lo = 0, st_hi = 0;
#define G8_STAMP(idx)                                                                                    \
  do {                                                                                                   \
    if (stamps && (idx) < 64) {                                                                          \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                        \
      st_lo = lane == (idx) ? (unsigned)t_ : st_lo;                                                      \
      st_hi = lane == (idx) ? (unsigned)(t_ >> 32) : st_hi;                                              \
    }                                                                                                    \
  } while (0)
  // bias gradient side product of the waves whose A unit is MH: four MFMAs against an all-ones fragment
#define G8_BIAS(MH)                                                                                      \
  do {                                                                                                   \
    if constexpr (BIAS) {                                                                                \
      if (do_bias && (wc >> 1) == (MH)) {                                                                \
        const E one = (E)1.f;                                                                            \
        const h8 ones = {one, one, one, one, one, one, one, one};                                        \
        if (wc & 1) {                                                                                    \
          _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                             \
            accb[0] = H16<E>::mfma32(ones, af[2][kk], accb[0]);                                          \
            accb[1] = H16<E>::mfma32(ones, af[3][kk], accb[1]);                                          \
          }                                                                                              \
        } else {                                                                                         \
          _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                             \
            accb[0] = H16<E>::mfma32(ones, af[0][kk], accb[0]);                                          \
            accb[1] = H16<E>::mfma32(ones, af[1][kk], accb[1]);                                          \
          }                                                                                              \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
      }                                                                                                  \
    }                                                                                                    \
  } while (0)

  // ---- prologue: the four units of tile 0 and the first two of tile 1; tile 0's A0 / B0 landed -----------------------------------------
  G8_STAMP(0);
  issueA(0, 0, 0);
  issueB(0, 0, 0);
  issueB(1, 0, 0);
  issueA(1, 0, 0);
  if (nk > 1) {
    issueA(0, 1, 1);
    issueB(0, 1, 1);
  }
  __builtin_amdgcn_sched_barrier(0);
  g8_wait_units(nk > 1 ? 4 : 2);
  g8_barrier();
  G8_STAMP(1);
  if (wr == 1) g8_barrier();  // waves 4-7 run one barrier behind waves 0-3

  constexpr int INF = 4;  // units still in flight behind the one a wait retires (this phase's unit is issued BEFORE the wait)
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    const unsigned bo = buf ? (unsigned)G8_BUF : 0u;
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
    const int sb = 2 + t * 12;  // stamps: 3 per phase (first barrier passed, MFMAs issued, second barrier passed)
    // ---- phase 1: A0, B0 of tile t; stage B1 of tile t+1; B1 of tile t retired --------------------------------------------------------
    G8_READ_B(G8_B0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    G8_READ_A(G8_A0);
    __builtin_amdgcn_sched_barrier(0);
    if (more1) issueB(1, buf ^ 1, t + 1);
    __builtin_amdgcn_sched_barrier(0);
    g8_wait_units(more1 ? INF : 1);
    g8_barrier();
    G8_STAMP(sb + 0);
    g8_lgkm0();
    G8_MFMA(0, 0, bf0);
    G8_BIAS(0);
    G8_STAMP(sb + 1);
    g8_barrier();
    G8_STAMP(sb + 2);
    // ---- phase 2: B1 of tile t; stage A1 of tile t+1; A1 of tile t retired --------------------------------------------------------------
    G8_READ_B(G8_B1, bf1);
    __builtin_amdgcn_sched_barrier(0);
    if (more1) issueA(1, buf ^ 1, t + 1);
    __builtin_amdgcn_sched_barrier(0);
    g8_wait_units(more1 ? INF : 0);
    g8_barrier();
    G8_STAMP(sb + 3);
    g8_lgkm0();
    G8_MFMA(0, 1, bf1);
    G8_STAMP(sb + 4);
    g8_barrier();
    G8_STAMP(sb + 5);
    // ---- phase 3: A1 of tile t; stage A0 of tile t+2 (its slot was last read in phase 1) ---------------------------------------------------
    G8_READ_A(G8_A1);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) issueA(0, buf, t + 2);
    __builtin_amdgcn_sched_barrier(0);
    g8_barrier();
    G8_STAMP(sb + 6);
    g8_lgkm0();
    G8_MFMA(1, 1, bf1);
    G8_BIAS(1);
    G8_STAMP(sb + 7);
    g8_barrier();
    G8_STAMP(sb + 8);
    // ---- phase 4: no reads (B0's fragments are still in registers); stage B0 of tile t+2; A0 / B0 of tile t+1 retired ---------------------
    if (more2) issueB(0, buf, t + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (more1) g8_wait_units(more2 ? INF : 2);
    g8_barrier();
    G8_STAMP(sb + 9);
    G8_MFMA(1, 0, bf0);
    G8_STAMP(sb + 10);
    g8_barrier();
    G8_STAMP(sb + 11);
  }
  if (wr == 0) g8_barrier();  // (balances the extra barrier of waves 4-7)
  G8_STAMP(63);
  if (stamps && blockIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamps[wave * 64 + lane] = ((unsigned long long)st_hi << 32) | st_lo;
  }
#undef G8_STAMP
#undef G8_READ_A
#undef G8_READ_B
#undef G8_MFMA
#undef G8_BIAS
}

// ---- grouped weight gradients (MODE 1) -------------------------------------------------------------------------------------------
constexpr int G8_WMAX = 40;
struct G8ProbW {
  const void* A;  // dY [K tokens, M] (lda)
  const void* B;  // X  [K tokens, N] (ldb)
  float* C;       // dW [M, N] fp32 (ldc)
  float* dbias;   // [M] fp32 or null
  int M, N, K, lda, ldb, ldc, tn, pad;
};
struct G8GroupW {
  int nprob, ntiles;
  float beta;
  int xcd;
  unsigned long long* stamps;  // measurement builds only (see g8_main), else null
  int tile_end[G8_WMAX];  // exclusive prefix sums of the problems' tile counts
  G8ProbW p[G8_WMAX];
};

// linear workgroup id -> linear tile id such that every XCD (ids congruent mod 8) walks a CONTIGUOUS run of tiles
__device__ __forceinline__ int g8_xcd_remap(int id, int nwg, int enable) {
  if (!enable || nwg < 16) return id;
  const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <typename E>
__global__ __launch_bounds__(512) void gemm8_wgrad_kernel(const G8GroupW g) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G8_SMEM];
  const int L = g8_xcd_remap(blockIdx.x, g.ntiles, g.xcd);
  int lo = 0, hi = g.nprob - 1;  // first problem whose tile_end exceeds L
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (L < g.tile_end[mid]) hi = mid;
    else lo = mid + 1;
  }
  const int z = __builtin_amdgcn_readfirstlane(lo);
  const G8ProbW& P = g.p[z];
  const int first = z ? g.tile_end[z - 1] : 0;
  const int rem = L - first, tile_m = rem / P.tn, tile_n = rem - tile_m * P.tn;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int M = P.M, N = P.N;
  f32x4 acc[8][4], accb[2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = P.dbias != nullptr && tile_n == 0;
  g8_main<E, D2R_GEMM_TN, true>(reinterpret_cast<const E*>(P.A), reinterpret_cast<const E*>(P.B), M, N, P.K, P.lda, P.ldb, m0, n0, smem, acc, accb,
                                     do_bias, D2R_G8_STAMPS ? g.stamps : nullptr);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  float* C = P.C;
  const int64_t ldc = P.ldc;
  const float beta = g.beta;
#pragma unroll
  for (int mh = 0; mh < 2; ++mh) {
    f32x4 old[4][4];
    if (beta != 0.f) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = min(m0 + wr * 128 + mh * 64 + i * 16 + fr, M - 1);
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
          const int col = min(n0 + wc * 64 + (nj >> 1) * 32 + (nj & 1) * 16 + fq * 4, N - 4);
          old[i][nj] = *reinterpret_cast<const f32x4*>(C + (int64_t)row * ldc + col);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + wr * 128 + mh * 64 + i * 16 + fr;
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) {
        const int col = n0 + wc * 64 + (nj >> 1) * 32 + (nj & 1) * 16 + fq * 4;
        f32x4 v = acc[mh * 4 + i][nj];
        if (beta != 0.f) v = f32x4{v[0] + beta * old[i][nj][0], v[1] + beta * old[i][nj][1], v[2] + beta * old[i][nj][2], v[3] + beta * old[i][nj][3]};
        if (row < M && col + 4 <= N) *reinterpret_cast<f32x4*>(C + (int64_t)row * ldc + col) = v;
      }
    }
  }
  if (do_bias && fq == 0) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int row = m0 + wr * 128 + (wc >> 1) * 64 + ((wc & 1) * 2 + e) * 16 + fr;
      if (row < M) P.dbias[row] += accb[e][0];  // every column of accb holds the sum; this lane is the row's only writer
    }
  }
}

// ---- forward / dX products (MODE 0): one problem, or a group of independent problems, 16-bit epilogue -------------------------------
constexpr int G8_FMAX = 16;
struct G8ProbF {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  const void* R;   // residual (ldr)
  void* P;         // saved pre-activation (ldc)
  const void* G;   // reference of the activation gradient (ldc)
  int M, N, K, lda, ldb, ldc, ldr, tn;
  float alpha, beta;
  int act, gact, band, pad;
};
struct G8GroupF {
  int nprob, ntiles, xcd, pad;
  unsigned long long* stamps;
  int tile_end[G8_FMAX];
  G8ProbF p[G8_FMAX];
};

template <typename E, int LAYOUT>
__global__ __launch_bounds__(512) void gemm8_fwd_kernel(const G8GroupF g) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G8_SMEM];
  const int L = g8_xcd_remap(blockIdx.x, g.ntiles, g.xcd);
  int lo = 0, hi = g.nprob - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (L < g.tile_end[mid]) hi = mid;
    else lo = mid + 1;
  }
  const int z = __builtin_amdgcn_readfirstlane(lo);
  const G8ProbF& P = g.p[z];
  const int first = z ? g.tile_end[z - 1] : 0;
  int rem = L - first, tile_m, tile_n;
  {
    // column bands (wide outputs): inside the problem the tiles are walked in bands of `band` n-tiles, m fastest across a band's rows,
    // so the B panels of a band stay in the XCD's L2 while the A row panels stream past once (gemm_args.h xcd_tile)
    const int gx = P.tn, gy = (P.M + 255) >> 8, band = P.band;
    if (band > 0 && band < gx) {
      const int per_band = gy * band, b = rem / per_band, firstn = b * band, w = min(band, gx - firstn);
      rem -= b * per_band;
      tile_m = rem / w;
      tile_n = firstn + rem - tile_m * w;
    } else {
      tile_m = rem / gx;
      tile_n = rem - tile_m * gx;
    }
  }
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int M = P.M, N = P.N;
  f32x4 acc[8][4], accb[2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  g8_main<E, LAYOUT, false>(reinterpret_cast<const E*>(P.A), reinterpret_cast<const E*>(P.B), M, N, P.K, P.lda, P.ldb, m0, n0, smem, acc, accb, false,
                                 D2R_G8_STAMPS ? g.stamps : nullptr);

  // ---- epilogue: per wave and 64-row half, the block goes through LDS (8-byte stores from the accumulator layout, 16-byte row
  //      packs back) and leaves through the shared pack epilogue; every DMA has landed (the tail waits drain to vmcnt(0)) and every
  //      wave is past its last LDS read (the closing barriers of the loop) -------------------------------------------------------------------
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  constexpr int LDE = 64 + 8;
  E* Cs = reinterpret_cast<E*>(smem) + wave * 64 * LDE;
  GemmArgs ga;
  ga.act = P.act, ga.gact = P.gact, ga.beta = P.beta;
  E* Cg = reinterpret_cast<E*>(P.C);
  E* Pg = reinterpret_cast<E*>(P.P);
  const E* Rg = reinterpret_cast<const E*>(P.R);
  const E* Gg = reinterpret_cast<const E*>(P.G);
  const float alpha = P.alpha;
  const float* bias = P.bias;
  g8_barrier();
#pragma unroll
  for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) {
        const int cl = (nj >> 1) * 32 + (nj & 1) * 16 + fq * 4;
        const int col = n0 + wc * 64 + cl;
        Pack<E, 4> pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float bv = (bias && col + r < N) ? bias[col + r] : 0.f;
          pk.v[r] = (E)(alpha * acc[mh * 4 + i][nj][r] + bv);
        }
        st_pack<E, 4>(Cs + (i * 16 + fr) * LDE + cl, pk);
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {  // (rolled on purpose: one copy of the epilogue arithmetic)
      const int e = it * 64 + lane;
      const int rl = e >> 3, ch = e & 7;
      const int row = m0 + wr * 128 + mh * 64 + rl, col = n0 + wc * 64 + ch * 8;
      if (row >= M || col >= N) continue;
      const Pack<E, 8> pv = ld_pack<E, 8>(Cs + rl * LDE + ch * 8);
      epilogue_pack8(ga, pv, Cg, Pg, Rg, Gg, (int64_t)row * P.ldc + col, (int64_t)row * P.ldr + col, N - col);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename E>
void launch_fwd(const G8GroupF& g, int layout, hipStream_t st) {
  if (layout == D2R_GEMM_NT) hipLaunchKernelGGL((gemm8_fwd_kernel<E, D2R_GEMM_NT>), dim3(g.ntiles), dim3(512), 0, st, g);
  else hipLaunchKernelGGL((gemm8_fwd_kernel<E, D2R_GEMM_NN>), dim3(g.ntiles), dim3(512), 0, st, g);
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------------
static unsigned long long* g8_stamps = nullptr;  // measurement builds (-DD2R_G8_STAMPS=1): destination of workgroup 0's cycle stamps
extern "C" void d2r_gemm8_debug_stamps(unsigned long long* dst) { g8_stamps = dst; }
// Eligibility of one forward / dX problem for the 256-wide tiles (checked by the dispatcher in gemm.hip; everything else keeps the
// 128-wide LDS-DMA kernels): 16-bit operands and output of one type, batch 1, K a multiple of 64 and at least 128, 16-byte aligned
// rows; k-strided operands need N a multiple of 8.
int d2r_gemm8_fwd_ok(const GemmArgs& a, int layout, int batch) {
  if (!d2r_is16(a.dtype) || a.c_dtype != a.dtype || batch != 1 || layout == D2R_GEMM_TN) return 0;
  if (a.K % 64 != 0 || a.K < 128 || a.M < 256 || a.N < 256 || a.N % 8 != 0) return 0;
  if (!a.vecA || !a.vecB || !a.vecC || a.dbias) return 0;
  return 1;
}

static void g8_fill_prob(G8ProbF& p, const GemmArgs& a) {
  p.A = a.A, p.B = a.B, p.C = a.C, p.bias = a.bias, p.R = a.R, p.P = a.P, p.G = a.G;
  p.M = a.M, p.N = a.N, p.K = a.K, p.lda = (int)a.lda, p.ldb = (int)a.ldb, p.ldc = (int)a.ldc, p.ldr = (int)a.ldr;
  p.tn = d2r_cdiv(a.N, 256);
  p.alpha = a.alpha, p.beta = a.beta, p.act = a.act, p.gact = a.gact, p.pad = 0;
  // column bands: the B panels of a band (band x 256 x K x 2 bytes) should take about a third of the 4 MB L2 of an XCD
  const int64_t panel = (int64_t)256 * a.K * 2;
  int band = (int)((int64_t)(1536 << 10) / (panel > 0 ? panel : 1));
  if (band < 2) band = 2;
  p.band = (p.tn > band && (int64_t)p.tn * panel > (3 << 20)) ? band : 0;
}

// `n` (<= G8_FMAX) independent problems of one layout and dtype in one launch
int d2r_gemm8_fwd_launch(const GemmArgs* probs, int n, int layout, hipStream_t st) {
  if (n < 1 || n > G8_FMAX) return d2r_fail(D2R_ERR_INVALID, "d2r_gemm8_fwd_launch: %d problems (1..%d)", n, G8_FMAX);
  G8GroupF g = {};
  g.nprob = n, g.xcd = 1;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    g8_fill_prob(g.p[i], probs[i]);
    tiles += d2r_cdiv(probs[i].M, 256) * g.p[i].tn;
    g.tile_end[i] = tiles;
  }
  for (int i = n; i < G8_FMAX; ++i) g.tile_end[i] = tiles;
  g.ntiles = tiles;
  g.stamps = g8_stamps;
  if (probs[0].dtype == D2R_F16) launch_fwd<f16_t>(g, layout, st);
  else launch_fwd<bf16_t>(g, layout, st);
  return d2r_check_launch("d2r_gemm8(fwd)");
}

// Grouped weight gradients of different shapes: problem i is C_i[M_i, N_i] (fp32, ldc_i) = beta * C_i + A_i^T B_i with A_i [K_i, M_i] (lda_i),
// B_i [K_i, N_i] (ldb_i) 16-bit, dbias_i[m] += sum_k A_i[k, m].  Every problem must pass d2r_gemm8_wgrad_ok.  Launches of up to G8_WMAX
// problems; no two problems of the call may share an output (checked by the caller).
int d2r_gemm8_wgrad_ok(int dtype, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, const void* A, const void* B, const void* C) {
  if (!d2r_is16(dtype) || M < 128 || N < 128 || K < 128 || M % 8 != 0 || N % 8 != 0) return 0;
  if ((lda * 2) % 16 != 0 || (ldb * 2) % 16 != 0 || (ldc * 4) % 16 != 0) return 0;
  if (!d2r_aligned16(A) || !d2r_aligned16(B) || !d2r_aligned16(C)) return 0;
  if (lda >= ((int64_t)1 << 31) || ldb >= ((int64_t)1 << 31) || ldc >= ((int64_t)1 << 31)) return 0;
  return 1;
}

int d2r_gemm8_wgrad_launch(int dtype, int count, const int* M, const int* N, const int* K, const int64_t* lda, const int64_t* ldb, const int64_t* ldc,
                           const void* const* A, const void* const* B, float* const* C, float* const* dbias, float beta, hipStream_t st) {
  for (int first = 0; first < count; first += G8_WMAX) {
    const int n = count - first < G8_WMAX ? count - first : G8_WMAX;
    G8GroupW g = {};
    g.nprob = n, g.beta = beta, g.xcd = 1;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
      const int s = first + i;
      G8ProbW& p = g.p[i];
      p.A = A[s], p.B = B[s], p.C = C[s], p.dbias = dbias ? dbias[s] : nullptr;
      p.M = M[s], p.N = N[s], p.K = K[s], p.lda = (int)lda[s], p.ldb = (int)ldb[s], p.ldc = (int)ldc[s];
      p.tn = d2r_cdiv(N[s], 256), p.pad = 0;
      tiles += d2r_cdiv(M[s], 256) * p.tn;
      g.tile_end[i] = tiles;
    }
    for (int i = n; i < G8_WMAX; ++i) g.tile_end[i] = tiles;
    g.ntiles = tiles;
    g.stamps = g8_stamps;
    if (dtype == D2R_F16) hipLaunchKernelGGL((gemm8_wgrad_kernel<f16_t>), dim3(tiles), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((gemm8_wgrad_kernel<bf16_t>), dim3(tiles), dim3(512), 0, st, g);
    if (int rc = d2r_check_launch("d2r_gemm8(wgrad)")) return rc;
  }
  return D2R_OK;
}

// ---- probe entry points (tests/probes/gemm8_probe.py; declared in include/d2r_hip_probes.h, not part of the drop-in surface) ---------
extern "C" int d2r_gemm8_probe_fwd(int dtype, int layout, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                                   int64_t ldc, const float* bias, int act, int nrep, void* stream) {
  GemmArgs a = {};
  a.A = A, a.B = B, a.C = C, a.bias = bias, a.M = M, a.N = N, a.K = K, a.lda = lda, a.ldb = ldb, a.ldc = ldc, a.ldr = ldc;
  a.alpha = 1.f, a.beta = 0.f, a.act = act, a.dtype = dtype, a.c_dtype = dtype, a.vecA = a.vecB = a.vecC = 1;
  if (!d2r_gemm8_fwd_ok(a, layout, 1)) return d2r_fail(D2R_ERR_INVALID, "d2r_gemm8_probe_fwd: shape not eligible");
  GemmArgs probs[G8_FMAX];
  if (nrep < 1 || nrep > G8_FMAX) return d2r_fail(D2R_ERR_INVALID, "d2r_gemm8_probe_fwd: nrep");
  for (int i = 0; i < nrep; ++i) probs[i] = a;  // (the same problem nrep times: identical results, a grouped launch of nrep x the tiles)
  return d2r_gemm8_fwd_launch(probs, nrep, layout, reinterpret_cast<hipStream_t>(stream));
}
