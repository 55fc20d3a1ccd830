// xattn2.hip — K2 / K4, second generation: single-head attention over the full 768-wide feature as a STREAMING kernel.
//
//   O[b] = softmax(scale * Q K^T + mask[b]) V (+ residual)         Q [B,Lq,768], K / V [B,Lk,768], Lk <= 640
//
// Replaces (reference file:line) the CrossModalAlignment core models/XModules.py:300-310 (= models/Refinement.py:105-115,
// scale 100/sqrt(768)) and the ContextRichCrossModalCell core models/Cells.py:244-246 (unscaled, residual Qs).
//
// The op moves B(2Lq+2Lk)*768*2 bytes for 4*B*Lq*Lk*768 flops (78 flop/B): HBM / L2 bound.  What limits a workgroup is the
// rate at which ONE CU can pull K and V (each re-read by every query tile of the sample) out of its XCD's L2, so the kernel
// is built around a continuous LDS-DMA stream:
//   * a 512-thread workgroup owns QT = 16 or 32 query rows of one sample; all query tiles of a sample run on one XCD
//     (blockIdx remap) so that K / V are fetched into ONE L2;
//   * K and then V flow through a 4-slot ring of 16-key chunks (24 KB each, global_load_lds_dwordx4, three chunks in
//     flight behind counted vmcnt waits), the LDS image XOR-swizzled through the per-lane SOURCE address;
//   * scores: the 768-wide contraction is split over the 8 waves (96 features each, Q fragments in registers), the 8
//     partial [16 keys x QT] tiles are summed through LDS in a fixed order -> S fp32 [QT][Lk] in LDS;
//   * softmax on the finished rows (wave per row, fp32 statistics, row log-sum-exp saved for the backward), P written
//     in place as bf16;
//   * O^T = V^T P^T with v_mfma_f32_16x16x16_bf16 per 16-key chunk: a wave owns 96 output columns, V^T fragments come
//     from the ring through ds_read_b64_tr_b16; the output tile is staged in LDS and stored as whole 1536-byte rows.
// Nothing of size [Lq, Lk] reaches HBM.  Deterministic (fixed summation order, no atomics).
#include <math.h>
#include <stdlib.h>

#include "gemm_args.h"

// the ablation switches (D2R_X2_DBG) are compiled in only with -DD2R_X3_PROBES=1 (run-time branches in the tile loops otherwise)
#ifndef D2R_X3_PROBES
#define D2R_X3_PROBES 0
#endif
#define X2_DBG(a) (D2R_X3_PROBES ? (a).dbg : 0)

namespace {

constexpr int X2_MAXCORE = 4;  // attention problems ("cores") of one launch: same shapes and strides, own tensors

template <typename E>
struct X2Args {
  const E *q[X2_MAXCORE], *k[X2_MAXCORE], *v[X2_MAXCORE], *res[X2_MAXCORE];
  E* o[X2_MAXCORE];
  float* lse[X2_MAXCORE];
  const float* mask;
  int64_t ldq, sqb, ldk, skb, ldv, svb, ldo, sob, ldr, srb;
  int B, Lq, Lk, ntile, ncore;
  float scale;
  int dbg;  // timing experiments only (D2R_X2_DBG): 1 = no DMA issue, 2 = no score / PV arithmetic, 3 = no swizzle
};

// wave-uniform pick of a per-core kernel argument (a select chain: a run-time index into the by-value argument arrays
// would be served from scratch)
template <typename P>
__device__ __forceinline__ P x2_pick(P const (&arr)[X2_MAXCORE], int core) {
  P r = arr[0];
#pragma unroll
  for (int c = 1; c < X2_MAXCORE; ++c) r = core == c ? arr[c] : r;
  return r;
}

constexpr int XE = 768, CH = 16, ROWB = XE * 2 /*1536*/, CB = CH * ROWB /*24576*/, NB = 4, NW = 8;

// position of 16-byte chunk c (0..95) of ring row r: XOR inside aligned groups of 16 chunks (256 B = one bank row)
__device__ __forceinline__ int swz(int c, int r) { return (c & ~15) | ((c & 15) ^ (r & 15)); }

template <int NQT, int LKMAX>
struct X2Lds {
  static constexpr int QT = NQT * 16;
  static constexpr int LSS = LKMAX + 4;                  // S row stride in floats (pad: conflict-free 16-byte reads of P rows)
  static constexpr int RING = 0;
  static constexpr int SPART = RING + NB * CB;            // [8 waves][QT][16 keys] fp32
  static constexpr int S = SPART + NW * QT * CH * 4;      // [QT][LSS] fp32; P (bf16) is written in place, same row stride
  static constexpr int MS = S + QT * LSS * 4;             // additive key mask fp32 [LKMAX]
  static constexpr int TOTAL = MS + LKMAX * 4;
  static_assert(TOTAL <= 160 * 1024, "LDS budget");
  static_assert(QT * (XE + 8) * 2 <= NB * CB, "the output tile is staged in the ring");
};

template <typename E, int NQT, int LKMAX>
__global__ __launch_bounds__(512) void xattn2_fwd_kernel(X2Args<E> a) {
  typedef typename H16<E>::v8 E8;
  typedef typename H16<E>::v4 E4;
  using L = X2Lds<NQT, LKMAX>;
  constexpr int QT = L::QT, LSS = L::LSS;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[L::TOTAL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  // all query tiles of a (core, sample) on one XCD (blocks b and b+8 share an XCD): id -> (core, sample, tile)
  const int id = blockIdx.x, xcd = id & 7, rr = id >> 3;
  const int tile = rr % a.ntile, unit = rr / a.ntile;
  const int core = unit % a.ncore, b = (unit / a.ncore) * 8 + xcd;
  if (b >= a.B) return;
  const int q0 = tile * QT;
  const E* Qg = x2_pick(a.q, core) + b * a.sqb;
  const E* Kg = x2_pick(a.k, core) + b * a.skb;
  const E* Vg = x2_pick(a.v, core) + b * a.svb;
  const E* Rg = x2_pick(a.res, core);
  E* Og = x2_pick(a.o, core) + b * a.sob;
  float* Lg = x2_pick(a.lse, core) + (int64_t)b * a.Lq;
  float* Sm = reinterpret_cast<float*>(smem + L::S);
  float* Sp = reinterpret_cast<float*>(smem + L::SPART);
  float* Ms = reinterpret_cast<float*>(smem + L::MS);
  const int nkc = (a.Lk + CH - 1) / CH, G = 2 * nkc;  // K chunks, then as many V chunks

  // key mask into LDS and the Q fragments of this wave's 96-feature slice into registers BEFORE the DMA stream starts
  // (an ordinary global load beside in-flight LDS-DMA makes the compiler drain the whole queue)
  for (int key = tid; key < LKMAX; key += 512) Ms[key] = key < a.Lk ? (a.mask ? a.mask[(int64_t)b * a.Lk + key] : 0.f) : -INFINITY;
  E8 qf[NQT][3];
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    const int qrow = min(q0 + t * 16 + fr, a.Lq - 1);
    const E* qp = Qg + (int64_t)qrow * a.ldq + wave * 96 + fq * 8;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) qf[t][kk] = *reinterpret_cast<const E8*>(qp + kk * 32);
  }
  // per-lane source of this wave's three 1-KiB pieces of a chunk image [16 rows][1536 B]
  int prow[3], pcol[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int o = (wave * 3 + i) * 1024 + lane * 16;
    const int row = o / ROWB, cp = (o - row * ROWB) >> 4;
    prow[i] = row;
    pcol[i] = (X2_DBG(a) == 3 ? cp : swz(cp, row)) * 8;  // the swizzle is an involution: image position cp of row `row` holds source chunk swz(cp,row)
  }
  auto issue = [&](int g) {
    const bool isv = g >= nkc;
    const E* src = isv ? Vg : Kg;
    const int64_t ld = isv ? a.ldv : a.ldk;
    const int key0 = (isv ? g - nkc : g) * CH;
    unsigned char* base = smem + L::RING + (g & (NB - 1)) * CB;
    if (X2_DBG(a) == 1) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int key = min(key0 + prow[i], a.Lk - 1);  // clamped: masked (-inf) scores / zero probabilities for keys >= Lk
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)key * ld + pcol[i]),
                                       (__attribute__((address_space(3))) void*)(base + (wave * 3 + i) * 1024), 16, 0, 0);
    }
  };
  auto wait_chunk = [&](int g) {  // chunk g landed (own pieces); chunks g+1, g+2 may stay in flight
    const int ahead = min(2, G - 1 - g);
    if (ahead == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // Q fragments and mask values have landed: the queue holds DMA pieces only from here
  __syncthreads();
  issue(0);
  if (G > 1) issue(1);
  if (G > 2) issue(2);

  // ---- phase K: S[q][key] = scale * sum_d Q[q,d] K[key,d] + mask[key] ------------------------------------------------
  for (int g = 0; g < nkc; ++g) {
    wait_chunk(g);
    if (g + 3 < G) issue(g + 3);  // slot (g+3)&3 = (g-1)&3: every wave finished chunk g-1 before the barrier above
    const unsigned char* slot = smem + L::RING + (g & (NB - 1)) * CB;
    if (X2_DBG(a) == 2) continue;
    f32x4 s[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      const int c = wave * 12 + kk * 4 + fq;
      const E8 kf = *reinterpret_cast<const E8*>(slot + fr * ROWB + swz(c, fr) * 16);
#pragma unroll
      for (int t = 0; t < NQT; ++t) s[t] = H16<E>::mfma32(kf, qf[t][kk], s[t]);
    }
    // s[t][r] = partial S[key = fq*4 + r][q = fr] over this wave's 96 features
#pragma unroll
    for (int t = 0; t < NQT; ++t) *reinterpret_cast<f32x4*>(Sp + ((wave * QT + t * 16 + fr) * CH + fq * 4)) = s[t];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < QT * CH) {  // fixed-order sum of the eight partials
      const int qq = tid >> 4, key = tid & 15;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += Sp[(w * QT + qq) * CH + key];
      Sm[qq * LSS + g * CH + key] = v * a.scale + Ms[g * CH + key];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // ---- softmax rows (wave per row), P in place as bf16 [q][key] (row stride LSS*4 bytes) ---------------------------------
  {
    constexpr int PER = LKMAX / 64;
    const int nkeys = nkc * CH;
    for (int row = wave; row < QT; row += NW) {
      float v[PER];
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int key = lane + 64 * i;
        v[i] = key < nkeys ? Sm[row * LSS + key] : -INFINITY;
        mx = fmaxf(mx, v[i]);
      }
      mx = wave_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        v[i] = __expf(v[i] - mx);
        sum += v[i];
      }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();  // every lane has its scores in registers before the row is overwritten
      E* Prow = reinterpret_cast<E*>(Sm + row * LSS);
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int key = lane + 64 * i;
        if (key < nkeys) Prow[key] = (E)(v[i] * inv);
      }
      if (lane == 0 && q0 + row < a.Lq) Lg[q0 + row] = mx + logf(sum);
    }
  }
  // (the lse store above is the only vector-memory op besides the DMA stream; it is older than nothing we wait for by count:
  //  the waits below are exact again because a store retires in order with the DMA pieces issued before it)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // re-prime the count: after the drain, chunks nkc..nkc+2 may already have been consumed from the queue; issue state is
  // tracked by `g`, and wait_chunk only relies on issue order, so nothing else to do.

  // ---- phase V: O^T[d][q] += V[key][d] P[q][key], 16 keys per step ----------------------------------------------------
  f32x4 o[6][NQT];
#pragma unroll
  for (int dt = 0; dt < 6; ++dt)
#pragma unroll
    for (int t = 0; t < NQT; ++t) o[dt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int g = nkc; g < G; ++g) {
    // chunks up to min(g+2, G-1) have been issued; everything issued before the drain above has landed
    // pieces issued since the drain: chunks nkc+3 .. g+2 (those before it have landed): the youngest min(2, ...) chunks may stay in flight
    const int ahead = min(min(2, G - 1 - g), g + 2 - (nkc + 2));
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (g + 3 < G) issue(g + 3);
    const unsigned char* slot = smem + L::RING + (g & (NB - 1)) * CB;
    const int key0 = (g - nkc) * CH;
    if (X2_DBG(a) == 2) continue;
    E4 pf[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) pf[t] = *reinterpret_cast<const E4*>(reinterpret_cast<const E*>(Sm + (t * 16 + fr) * LSS) + key0 + fq * 4);
    const int vrow = fq * 4 + tq;
    E4 vf[6];
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      const int c = wave * 12 + dt * 2 + (tp >> 1);
      vf[dt] = lds_tr_read<E4>(slot + vrow * ROWB + swz(c, vrow) * 16 + (tp & 1) * 8);
    }
    lds_reads_done();
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
#pragma unroll
      for (int t = 0; t < NQT; ++t) o[dt][t] = H16<E>::mfma16(vf[dt], pf[t], o[dt][t]);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the ring is dead: it now stages the output tile
  // ---- epilogue: o[dt][t][r] = O[q = t*16 + fr][d = wave*96 + dt*16 + fq*4 + r] -> LDS rows -> whole-row global stores ----
  constexpr int LDO = XE + 8;
  E* Os = reinterpret_cast<E*>(smem + L::RING);
#pragma unroll
  for (int t = 0; t < NQT; ++t)
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      Pack<E, 4> pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk.v[r] = (E)o[dt][t][r];
      st_pack<E, 4>(Os + (t * 16 + fr) * LDO + wave * 96 + dt * 16 + fq * 4, pk);
    }
  __syncthreads();
  for (int e = tid; e < QT * (XE / 8); e += 512) {
    const int row = e / (XE / 8), ch = e - row * (XE / 8);
    const int qrow = q0 + row;
    if (qrow >= a.Lq) continue;
    Pack<E, 8> v = ld_pack<E, 8>(Os + row * LDO + ch * 8);
    if (Rg) {
      const Pack<E, 8> rv = ld_pack<E, 8>(Rg + b * a.srb + (int64_t)qrow * a.ldr + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v.v[j] = (E)((float)v.v[j] + (float)rv.v[j]);
    }
    st_pack<E, 8>(Og + (int64_t)qrow * a.ldo + ch * 8, v);
  }
}

template <typename E>
static int x2_launch(int ncore, const void* const* q, int64_t ldq, int64_t sqb, const void* const* k, int64_t ldk, int64_t skb,
                     const void* const* v, int64_t ldv, int64_t svb, void* const* o, int64_t ldo, int64_t sob, const void* const* residual,
                     int64_t ldr, int64_t srb, const float* mask, float* const* lse, int B, int Lq, int Lk, float scale, hipStream_t st) {
  X2Args<E> a = {};
  for (int c = 0; c < ncore; ++c) {
    a.q[c] = (const E*)q[c], a.k[c] = (const E*)k[c], a.v[c] = (const E*)v[c], a.o[c] = (E*)o[c], a.lse[c] = lse[c];
    a.res[c] = residual ? (const E*)residual[c] : nullptr;
  }
  a.mask = mask, a.ncore = ncore;
  a.ldq = ldq, a.sqb = sqb, a.ldk = ldk, a.skb = skb, a.ldv = ldv, a.svb = svb, a.ldo = ldo, a.sob = sob, a.ldr = ldr, a.srb = srb;
  a.B = B, a.Lq = Lq, a.Lk = Lk, a.scale = scale;
  a.dbg = 0;
  const int bgrp = (B + 7) / 8 * 8;
  if (Lk <= 256) {
    // 32-query tiles halve the K / V re-reads; 16-query tiles double the workgroups: take 32 when that still fills the chip
    const int nt32 = (Lq + 31) / 32;
    if (nt32 * B * ncore >= 192) {
      a.ntile = nt32;
      hipLaunchKernelGGL((xattn2_fwd_kernel<E, 2, 256>), dim3(bgrp * a.ntile * ncore), dim3(512), 0, st, a);
    } else {
      a.ntile = (Lq + 15) / 16;
      hipLaunchKernelGGL((xattn2_fwd_kernel<E, 1, 256>), dim3(bgrp * a.ntile * ncore), dim3(512), 0, st, a);
    }
  } else {
    a.ntile = (Lq + 15) / 16;
    hipLaunchKernelGGL((xattn2_fwd_kernel<E, 1, 640>), dim3(bgrp * a.ntile * ncore), dim3(512), 0, st, a);
  }
  return 1;
}

}  // namespace

// Host entry used by d2r_xattn_fwd / d2r_xattn_fwd_multi (attention.hip): `ncore` (1..4) attention problems of identical shape and
// strides in ONE launch (h_* are host arrays of device pointers; h_res may be NULL).  Returns 1 when the launch was taken.
int d2r_xattn2_fwd_try(int dtype, int ncore, const void* const* q, int64_t ldq, int64_t sqb, const void* const* k, int64_t ldk, int64_t skb,
                       const void* const* v, int64_t ldv, int64_t svb, void* const* o, int64_t ldo, int64_t sob, const void* const* residual,
                       int64_t ldr, int64_t srb, const float* mask, float* const* lse, int B, int Lq, int Lk, float scale, hipStream_t st) {
  if (Lk > 640 || Lk < 1 || Lq < 1 || ncore < 1 || ncore > X2_MAXCORE) return 0;
  if (dtype == D2R_F16)
    return x2_launch<f16_t>(ncore, q, ldq, sqb, k, ldk, skb, v, ldv, svb, o, ldo, sob, residual, ldr, srb, mask, lse, B, Lq, Lk, scale, st);
  return x2_launch<bf16_t>(ncore, q, ldq, sqb, k, ldk, skb, v, ldv, svb, o, ldo, sob, residual, ldr, srb, mask, lse, B, Lq, Lk, scale, st);
}
