// attention.hip — K3 fused multi-head attention core for the short sequences of this path (L <= 256), bf16.
//
//   O = softmax(scale * Q K^T + mask) V (+ residual)      per (batch, head), head dim 64 or 48
//
// Replaces (reference file:line): BertSelfAttention scores/softmax/context (models/modeling_unimo.py:385-424),
// CLIPAttention (:150-215) and the 16-head self-attention of the intra-modal reasoning cell
// (models/SelfAttention.py:20-60) — there three launches (batched QK^T GEMM, softmax, batched PV GEMM) that write
// and re-read the [B,H,L,L] score and probability tensors; here one launch that keeps them in registers.
//
// Forward: one 512-thread workgroup per (b, h); K and V of the head sit in LDS ([key][64+8] bf16) and each wave
// takes 16 query rows per round.  A wave computes S^T = K Q^T for its 16 queries with v_mfma_f32_16x16x32_bf16 — in the MFMA C layout a lane then
// holds ONE query (col = lane&15) and 4 keys per tile, so the softmax is a per-lane loop plus two cross-lane steps
// (xor 16, 32), and the probabilities are already laid out as the B operand of O^T = V^T P^T (k-slot j of lane group
// g <-> key 32u + 4g + j, 32u + 16 + 4g + j-4); V^T fragments come from LDS through ds_read_b64_tr_b16 with the
// same key permutation.  No LDS round trip for P, no [L,L] tensor in HBM; the only extra output is the row
// log-sum-exp (fp32) the backward needs.
//
// Backward: one 512-thread workgroup per (b, h) with Q, K, V, dO resident in LDS.  Phase A (a wave owns 16
// queries): recompute P^T, dP^T = V dO^T, D = rowsum(P.dP), dS^T, dQ^T = K^T dS^T.  Phase B (a wave owns 16 keys):
// recompute P and dS in the other orientation (rows = queries), dV^T += dO^T P, dK^T += Q^T dS.  Fixed summation
// order, no atomics.
#include <math.h>
#include <stdlib.h>

#include "gemm_args.h"

namespace {

struct MhaArgs {
  const bf16_t *q, *k, *v, *res, *dO;
  bf16_t *o, *dq, *dk, *dv;
  const float* mask;  // additive [B, Lk] or null
  float* lse;         // [B, H, Lq]
  int64_t ldq, sqb, ldk, skb, ldv, svb, ldo, sob, ldr, srb, ldg, sgb, lddq, sdqb, lddk, sdkb, lddv, sdvb;
  int B, H, Lq, Lk;
  float scale;
};

constexpr int LDW = 72;  // LDS row stride in bf16: 64 columns + 8 pad (144 B: conflict-free b128 and tr16 reads)
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// rows [0, TOTAL) x 64 columns of a head slice into LDS; rows >= valid and columns >= DH are zero.  All global
// loads of the slice are issued before the first LDS store (one latency, not one per 16 bytes).
template <int DH, int TOTAL, int NTHREADS>
__device__ __forceinline__ void fill_rows(bf16_t* dst, const bf16_t* src, int64_t ld, int valid, int tid) {
  constexpr int CHUNKS = TOTAL * 8, IT = (CHUNKS + NTHREADS - 1) / NTHREADS;
  Pack<bf16_t, 8> v[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = tid + it * NTHREADS, row = c >> 3, col = (c & 7) * 8;
    if (c < CHUNKS && row < valid && col < DH) {
      v[it] = ld_pack<bf16_t, 8>(src + (int64_t)row * ld + col);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[it].v[j] = (bf16_t)0.f;
    }
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = tid + it * NTHREADS, row = c >> 3, col = (c & 7) * 8;
    if (CHUNKS % NTHREADS == 0 || c < CHUNKS) st_pack<bf16_t, 8>(dst + row * LDW + col, v[it]);
  }
}

__device__ __forceinline__ bf16x8 lds_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

// A fragment of M^T for a matrix kept [k][col] in LDS: rows k0 + {0..3} and k0 + 16 + {0..3} of this lane group,
// column c0 + (lane & 15)
__device__ __forceinline__ bf16x8 lds_frag_tr(const bf16_t* base, int k0, int c0, int tq, int tp) {
  const bf16_t* p0 = base + (k0 + tq) * LDW + c0 + tp * 4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p0);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 16 * LDW));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ bf16x8 pack_frag(const f32x4& a, const f32x4& b) {
  return bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
}

__device__ __forceinline__ float group4_max(float v) {  // over the 4 lane groups holding one MFMA column
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group4_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

template <int DH, int NK32>
__global__ __launch_bounds__(512) void mha_fwd_kernel(MhaArgs a) {
  constexpr int LKP = NK32 * 32, NKT = NK32 * 2, NDT = DH / 16, NW = 8;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[LKP * LDW];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[LKP * LDW];
  __shared__ __attribute__((aligned(16))) float Ms[LKP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  fill_rows<DH, LKP, 512>(Ks, a.k + b * a.skb + h * DH, a.ldk, a.Lk, tid);
  fill_rows<DH, LKP, 512>(Vs, a.v + b * a.svb + h * DH, a.ldv, a.Lk, tid);
  for (int key = tid; key < LKP; key += 512)
    Ms[key] = key < a.Lk ? (a.mask ? a.mask[(int64_t)b * a.Lk + key] : 0.f) : -INFINITY;
  __syncthreads();
  const int nqt = (a.Lq + 15) / 16;
  for (int qt = wave; qt < nqt; qt += NW) {  // 16 queries per wave and round; no barrier below
    asm volatile("" ::: "memory");  // keep the (round-invariant) K/V fragment reads inside the round: no hoisting
    const int qrow = qt * 16 + fr;
    const bool qok = qrow < a.Lq;
    const bf16_t* Qg = a.q + b * a.sqb + h * DH + (int64_t)qrow * a.ldq;
    bf16x8 qf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int col = kk * 32 + fq * 8;
      if (qok && col < DH) {
        qf[kk] = *reinterpret_cast<const bf16x8*>(Qg + col);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[kk][j] = (bf16_t)0.f;
      }
    }
    f32x4 s[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(&Ks[(t * 16 + fr) * LDW + kk * 32 + fq * 8]), qf[kk], acc, 0, 0, 0);
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(&Ms[t * 16 + fq * 4]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[r] = acc[r] * a.scale + m4[r];
        mx = fmaxf(mx, acc[r]);
      }
      s[t] = acc;
    }
    mx = group4_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] = __expf(s[t][r] - mx);
        sum += s[t][r];
      }
    sum = group4_sum(sum);
    const float inv = 1.f / sum;
    if (fq == 0 && qok) a.lse[((int64_t)b * a.H + h) * a.Lq + qrow] = mx + logf(sum);
    f32x4 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NK32; ++u) {
      const bf16x8 pf = pack_frag(s[2 * u] * inv, s[2 * u + 1] * inv);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_tr(Vs, u * 32 + fq * 4, dt * 16, tq, tp), pf, o[dt], 0, 0, 0);
    }
    if (!qok) continue;
    bf16_t* Og = a.o + b * a.sob + h * DH + (int64_t)qrow * a.ldo;
    const bf16_t* Rg = a.res ? a.res + b * a.srb + h * DH + (int64_t)qrow * a.ldr : nullptr;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int c = dt * 16 + fq * 4;
      Pack<bf16_t, 4> out;
      if (Rg) {
        const Pack<bf16_t, 4> rv = ld_pack<bf16_t, 4>(Rg + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) out.v[r] = (bf16_t)(o[dt][r] + (float)rv.v[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) out.v[r] = (bf16_t)o[dt][r];
      }
      st_pack<bf16_t, 4>(Og + c, out);
    }
  }
}

template <int DH, int NK32>
__global__ __launch_bounds__(512) void mha_bwd_kernel(MhaArgs a) {
  constexpr int LP = NK32 * 32, NT16 = NK32 * 2, NDT = DH / 16, NW = 8;
  __shared__ __attribute__((aligned(16))) bf16_t Qs[LP * LDW];
  __shared__ __attribute__((aligned(16))) bf16_t Ks[LP * LDW];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[LP * LDW];
  __shared__ __attribute__((aligned(16))) bf16_t Gs[LP * LDW];
  __shared__ __attribute__((aligned(16))) float Ms[LP];
  __shared__ __attribute__((aligned(16))) float Ls[LP];
  __shared__ __attribute__((aligned(16))) float Ds[LP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  fill_rows<DH, LP, 512>(Qs, a.q + b * a.sqb + h * DH, a.ldq, a.Lq, tid);
  fill_rows<DH, LP, 512>(Ks, a.k + b * a.skb + h * DH, a.ldk, a.Lk, tid);
  fill_rows<DH, LP, 512>(Vs, a.v + b * a.svb + h * DH, a.ldv, a.Lk, tid);
  fill_rows<DH, LP, 512>(Gs, a.dO + b * a.sgb + h * DH, a.ldg, a.Lq, tid);
  for (int i = tid; i < LP; i += 512) {
    Ms[i] = i < a.Lk ? (a.mask ? a.mask[(int64_t)b * a.Lk + i] : 0.f) : -INFINITY;
    Ls[i] = i < a.Lq ? a.lse[((int64_t)b * a.H + h) * a.Lq + i] : INFINITY;
    Ds[i] = 0.f;
  }
  __syncthreads();
  const int nqt = (a.Lq + 15) / 16, nkt = (a.Lk + 15) / 16;

  // ---- phase A: 16 queries per wave -> D, dQ ----------------------------------------------------------
  for (int qt = wave; qt < nqt; qt += NW) {
    const int q0 = qt * 16;
    bf16x8 qf[2], gf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qf[kk] = lds_frag(&Qs[(q0 + fr) * LDW + kk * 32 + fq * 8]);
      gf[kk] = lds_frag(&Gs[(q0 + fr) * LDW + kk * 32 + fq * 8]);
    }
    const float lse = Ls[q0 + fr];
    // P^T tile t (rows = keys t*16 + fq*4 + r, col = query fr) and dP^T = V dO^T in the same layout; computed twice
    // (once for D, once for dS) instead of keeping 2 x NT16 accumulator tiles live across the row reduction
    auto tile = [&](int t, f32x4& pv, f32x4& dpv) {
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(&Ks[(t * 16 + fr) * LDW + kk * 32 + fq * 8]), qf[kk], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(&Vs[(t * 16 + fr) * LDW + kk * 32 + fq * 8]), gf[kk], da, 0, 0, 0);
      }
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(&Ms[t * 16 + fq * 4]);
#pragma unroll
      for (int r = 0; r < 4; ++r) sa[r] = __expf(sa[r] * a.scale + m4[r] - lse);
      pv = sa;
      dpv = da;
    };
    float dsum = 0.f;
#pragma unroll 2
    for (int t = 0; t < NT16; ++t) {
      f32x4 pv, dpv;
      tile(t, pv, dpv);
#pragma unroll
      for (int r = 0; r < 4; ++r) dsum += pv[r] * dpv[r];
    }
    dsum = group4_sum(dsum);
    if (fq == 0) Ds[q0 + fr] = dsum;
    f32x4 dq[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int u = 0; u < NK32; ++u) {
      f32x4 p0, p1, d0, d1;
      tile(2 * u, p0, d0);
      tile(2 * u + 1, p1, d1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d0[r] = p0[r] * (d0[r] - dsum) * a.scale;
        d1[r] = p1[r] * (d1[r] - dsum) * a.scale;
      }
      const bf16x8 dsf = pack_frag(d0, d1);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_tr(Ks, u * 32 + fq * 4, dt * 16, tq, tp), dsf, dq[dt], 0, 0, 0);
    }
    if (q0 + fr < a.Lq) {
      bf16_t* dQg = a.dq + b * a.sdqb + h * DH + (int64_t)(q0 + fr) * a.lddq;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        Pack<bf16_t, 4> out;
#pragma unroll
        for (int r = 0; r < 4; ++r) out.v[r] = (bf16_t)dq[dt][r];
        st_pack<bf16_t, 4>(dQg + dt * 16 + fq * 4, out);
      }
    }
  }
  __syncthreads();

  // ---- phase B: 16 keys per wave -> dK, dV --------------------------------------------------------------
  for (int kt = wave; kt < nkt; kt += NW) {
    const int k0 = kt * 16;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      kf[kk] = lds_frag(&Ks[(k0 + fr) * LDW + kk * 32 + fq * 8]);
      vf[kk] = lds_frag(&Vs[(k0 + fr) * LDW + kk * 32 + fq * 8]);
    }
    const float mk = Ms[k0 + fr];
    f32x4 dk[NDT], dv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int u = 0; u < NK32; ++u) {
      f32x4 pt[2], dst[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int r0 = (u * 2 + half) * 16;
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(&Qs[(r0 + fr) * LDW + kk * 32 + fq * 8]), kf[kk], sa, 0, 0, 0);
          da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(&Gs[(r0 + fr) * LDW + kk * 32 + fq * 8]), vf[kk], da, 0, 0, 0);
        }
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(&Ls[r0 + fq * 4]);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(&Ds[r0 + fq * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __expf(sa[r] * a.scale + mk - l4[r]);
          pt[half][r] = pv;
          dst[half][r] = pv * (da[r] - d4[r]) * a.scale;
        }
      }
      const bf16x8 pf = pack_frag(pt[0], pt[1]), dsf = pack_frag(dst[0], dst[1]);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_tr(Gs, u * 32 + fq * 4, dt * 16, tq, tp), pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_tr(Qs, u * 32 + fq * 4, dt * 16, tq, tp), dsf, dk[dt], 0, 0, 0);
      }
    }
    if (k0 + fr < a.Lk) {
      bf16_t* dKg = a.dk + b * a.sdkb + h * DH + (int64_t)(k0 + fr) * a.lddk;
      bf16_t* dVg = a.dv + b * a.sdvb + h * DH + (int64_t)(k0 + fr) * a.lddv;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        Pack<bf16_t, 4> ok, ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ok.v[r] = (bf16_t)dk[dt][r];
          ov.v[r] = (bf16_t)dv[dt][r];
        }
        st_pack<bf16_t, 4>(dKg + dt * 16 + fq * 4, ok);
        st_pack<bf16_t, 4>(dVg + dt * 16 + fq * 4, ov);
      }
    }
  }
}

bool aligned_slice(const void* p, int64_t ld, int64_t sb, int dh) {
  return p && (reinterpret_cast<uintptr_t>(p) & 15u) == 0 && ld % 8 == 0 && sb % 8 == 0 && (dh * 2) % 16 == 0;
}

template <int DH>
void launch_fwd(const MhaArgs& a, int nk32, hipStream_t st) {
  const dim3 grid(a.B * a.H), block(512);
  switch (nk32) {
#define D2R_CASE(N) case N: hipLaunchKernelGGL((mha_fwd_kernel<DH, N>), grid, block, 0, st, a); break;
    D2R_CASE(1) D2R_CASE(2) D2R_CASE(3) D2R_CASE(4) D2R_CASE(5) D2R_CASE(6) D2R_CASE(7) D2R_CASE(8)
#undef D2R_CASE
  }
}
template <int DH>
void launch_bwd(const MhaArgs& a, int nk32, hipStream_t st) {
  const dim3 grid(a.B * a.H), block(512);
  switch (nk32) {
#define D2R_CASE(N) case N: hipLaunchKernelGGL((mha_bwd_kernel<DH, N>), grid, block, 0, st, a); break;
    D2R_CASE(1) D2R_CASE(2) D2R_CASE(3) D2R_CASE(4) D2R_CASE(5) D2R_CASE(6) D2R_CASE(7) D2R_CASE(8)
#undef D2R_CASE
  }
}

}  // namespace

extern "C" int d2r_mha_supported(int dtype, int Lq, int Lk, int head_dim) {
  return dtype == D2R_BF16 && (head_dim == 64 || head_dim == 48) && Lq >= 1 && Lk >= 1 && Lq <= 256 && Lk <= 256;
}

extern "C" int d2r_mha_fwd(int dtype, const void* q, int64_t ldq, int64_t sqb, const void* k, int64_t ldk, int64_t skb,
                           const void* v, int64_t ldv, int64_t svb, void* o, int64_t ldo, int64_t sob,
                           const void* residual, int64_t ldr, int64_t srb, const float* mask, float* lse, int B, int H,
                           int Lq, int Lk, int head_dim, float scale, void* stream) {
  D2R_REQUIRE(d2r_mha_supported(dtype, Lq, Lk, head_dim), "d2r_mha_fwd: unsupported (bf16, head_dim 64|48, L <= 256 only)");
  D2R_REQUIRE(B >= 1 && H >= 1 && lse, "d2r_mha_fwd: bad arguments");
  D2R_REQUIRE(aligned_slice(q, ldq, sqb, head_dim) && aligned_slice(k, ldk, skb, head_dim) && aligned_slice(v, ldv, svb, head_dim) &&
                  aligned_slice(o, ldo, sob, head_dim) && (!residual || aligned_slice(residual, ldr, srb, head_dim)),
              "d2r_mha_fwd: pointers must be 16-byte aligned, strides multiples of 8 elements");
  MhaArgs a = {};
  a.q = (const bf16_t*)q, a.k = (const bf16_t*)k, a.v = (const bf16_t*)v, a.res = (const bf16_t*)residual, a.o = (bf16_t*)o;
  a.mask = mask, a.lse = lse;
  a.ldq = ldq, a.sqb = sqb, a.ldk = ldk, a.skb = skb, a.ldv = ldv, a.svb = svb, a.ldo = ldo, a.sob = sob, a.ldr = ldr, a.srb = srb;
  a.B = B, a.H = H, a.Lq = Lq, a.Lk = Lk, a.scale = scale;
  const int nk32 = d2r_cdiv(Lk, 32);
  if (head_dim == 64) launch_fwd<64>(a, nk32, (hipStream_t)stream);
  else launch_fwd<48>(a, nk32, (hipStream_t)stream);
  return d2r_check_launch("d2r_mha_fwd");
}

extern "C" int d2r_mha_bwd(int dtype, const void* q, int64_t ldq, int64_t sqb, const void* k, int64_t ldk, int64_t skb,
                           const void* v, int64_t ldv, int64_t svb, const void* dO, int64_t ldg, int64_t sgb,
                           const float* mask, const float* lse, void* dq, int64_t lddq, int64_t sdqb, void* dk,
                           int64_t lddk, int64_t sdkb, void* dv, int64_t lddv, int64_t sdvb, int B, int H, int Lq, int Lk,
                           int head_dim, float scale, void* stream) {
  D2R_REQUIRE(d2r_mha_supported(dtype, Lq, Lk, head_dim), "d2r_mha_bwd: unsupported (bf16, head_dim 64|48, L <= 256 only)");
  D2R_REQUIRE(B >= 1 && H >= 1 && lse, "d2r_mha_bwd: bad arguments");
  D2R_REQUIRE(aligned_slice(q, ldq, sqb, head_dim) && aligned_slice(k, ldk, skb, head_dim) && aligned_slice(v, ldv, svb, head_dim) &&
                  aligned_slice(dO, ldg, sgb, head_dim) && aligned_slice(dq, lddq, sdqb, head_dim) &&
                  aligned_slice(dk, lddk, sdkb, head_dim) && aligned_slice(dv, lddv, sdvb, head_dim),
              "d2r_mha_bwd: pointers must be 16-byte aligned, strides multiples of 8 elements");
  MhaArgs a = {};
  a.q = (const bf16_t*)q, a.k = (const bf16_t*)k, a.v = (const bf16_t*)v, a.dO = (const bf16_t*)dO;
  a.dq = (bf16_t*)dq, a.dk = (bf16_t*)dk, a.dv = (bf16_t*)dv;
  a.mask = mask, a.lse = const_cast<float*>(lse);
  a.ldq = ldq, a.sqb = sqb, a.ldk = ldk, a.skb = skb, a.ldv = ldv, a.svb = svb, a.ldg = ldg, a.sgb = sgb;
  a.lddq = lddq, a.sdqb = sdqb, a.lddk = lddk, a.sdkb = sdkb, a.lddv = lddv, a.sdvb = sdvb;
  a.B = B, a.H = H, a.Lq = Lq, a.Lk = Lk, a.scale = scale;
  const int nk32 = d2r_cdiv(Lq > Lk ? Lq : Lk, 32);
  if (head_dim == 64) launch_bwd<64>(a, nk32, (hipStream_t)stream);
  else launch_bwd<48>(a, nk32, (hipStream_t)stream);
  return d2r_check_launch("d2r_mha_bwd");
}

// =====================================================================================================
// K2 / K4: single-head attention over the full 768-wide feature (CrossModalAlignment, models/XModules.py:300-310 and
// models/Refinement.py:105-115, logit scale 100/sqrt(768); ContextRichCrossModalCell core, models/Cells.py:244-246,
// unscaled, residual Qs).  Algorithmic traffic B*(2Lq+2Lk)*768*2 bytes: HBM-bound on paper.
//
// Forward: a 512-thread workgroup owns 32 query rows of one sample.
//   phase 1  S^T = K Q^T over the 768-wide contraction: wave w computes the key tiles {w, w+8} x both 16-query
//            tiles; K fragments stream straight from global memory (each lane 16 contiguous bytes of one key row),
//            Q fragments come from LDS (staged once).  Row max / sum are combined across waves through LDS; the
//            normalised probabilities go to LDS as bf16 [query][key].
//   phase 2  O^T = V^T P^T: wave w owns output columns [96w, 96w+96); it streams ITS [32 keys x 96] sub-block of V
//            through a private double-buffered LDS slab (registers -> LDS, hardware-transposed back with
//            ds_read_b64_tr_b16), so phase 2 needs no workgroup barrier at all.
// Nothing of size [Lq, Lk] reaches HBM; the extra output is the row log-sum-exp for the backward pass.
// =====================================================================================================
namespace {

struct XattnArgs {
  const bf16_t *q, *k, *v, *res, *dO;
  bf16_t *o, *dq, *p_out, *ds_out;
  const float* mask;
  float* lse;
  int64_t ldq, sqb, ldk, skb, ldv, svb, ldo, sob, ldr, srb, ldg, sgb, lddq, sdqb;
  int B, Lq, Lk, lkp;
  float scale;
};

constexpr int XD = 768, XQ = 32, XW = 8, XCOLS = XD / XW;  // 96 output columns per wave
constexpr int LDQ = XD + 8;                                  // Qs row stride (bf16)
constexpr int LDP = 256 + 8;                                 // Ps row stride (bf16), keys padded to <= 256
constexpr int LDV = XCOLS + 8;                               // V slab row stride (bf16)

// S^T / dP^T style product for this wave's (up to two) key tiles and both query tiles:
// acc[ti][qt] += sum_d X[key, d] * Y[q, d], X rows from global (clamped), Y rows from LDS
__device__ __forceinline__ void xattn_scores(const bf16_t* __restrict__ Xg, int64_t ldx, int Lk, const bf16_t* Ys, int wave,
                                             int nt16, int fr, int fq, f32x4 (&acc)[2][2]) {
  const bf16_t* xrow[2];
  bool act[2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int t = wave + ti * XW;
    act[ti] = t < nt16 && t * 16 < Lk;
    const int key = min(t * 16 + fr, Lk - 1);
    xrow[ti] = Xg + (int64_t)key * ldx + fq * 8;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) acc[ti][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (!act[0]) return;  // tiles are assigned in order: no first tile, no second
#pragma unroll 8
  for (int kk = 0; kk < XD / 32; ++kk) {
    const bf16x8 y0 = lds_frag(Ys + fr * LDQ + kk * 32 + fq * 8);
    const bf16x8 y1 = lds_frag(Ys + (16 + fr) * LDQ + kk * 32 + fq * 8);
    const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xrow[0] + kk * 32);
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, y0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, y1, acc[0][1], 0, 0, 0);
    if (act[1]) {
      const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(xrow[1] + kk * 32);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, y0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, y1, acc[1][1], 0, 0, 0);
    }
  }
}

// out^T[d, q] = sum_key M[key, d] * W[q, key] for this wave's 96 columns d and both query tiles; M rows stream from
// global through the wave's private LDS slab, W (bf16 [32][LDP]) is read from LDS.  acc[dt][qt].
__device__ __forceinline__ void xattn_apply(const bf16_t* __restrict__ Mg, int64_t ldm, int Lk, int nk32, const bf16_t* Ws,
                                            bf16_t* slab, int wave, int lane, f32x4 (&acc)[6][2]) {
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
#pragma unroll
  for (int dt = 0; dt < 6; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) acc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // 32 rows x 96 columns = 384 16-byte chunks per step: 6 per lane; two register sets keep the loads of steps
  // u+1 and u+2 in flight while step u is multiplied (the loop is latency-bound, not bandwidth-bound)
  Pack<bf16_t, 8> stage[2][6];
  auto load = [&](int u, Pack<bf16_t, 8> (&st)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = lane + i * 64, row = c / 12, ch = c % 12;
      const int key = min(u * 32 + row, Lk - 1);  // clamped: the matching probabilities are exactly zero
      st[i] = ld_pack<bf16_t, 8>(Mg + (int64_t)key * ldm + wave * XCOLS + ch * 8);
    }
  };
  auto store = [&](bf16_t* dst, const Pack<bf16_t, 8> (&st)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = lane + i * 64, row = c / 12, ch = c % 12;
      st_pack<bf16_t, 8>(dst + row * LDV + ch * 8, st[i]);
    }
  };
  load(0, stage[0]);
  if (nk32 > 1) load(1, stage[1]);
#pragma unroll 2
  for (int u = 0; u < nk32; ++u) {
    bf16_t* cur = slab + (u & 1) * 32 * LDV;
    if (u & 1) {
      store(cur, stage[1]);
      if (u + 2 < nk32) load(u + 2, stage[1]);
    } else {
      store(cur, stage[0]);
      if (u + 2 < nk32) load(u + 2, stage[0]);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's slab stores have landed (wave-private region)
    __builtin_amdgcn_wave_barrier();
    const bf16x8 w0 = lds_frag(Ws + fr * LDP + u * 32 + fq * 8);
    const bf16x8 w1 = lds_frag(Ws + (16 + fr) * LDP + u * 32 + fq * 8);
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      const bf16_t* p0 = cur + (fq * 8 + tq) * LDV + dt * 16 + tp * 4;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p0);
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * LDV));
      const bf16x8 m = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      acc[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m, w0, acc[dt][0], 0, 0, 0);
      acc[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m, w1, acc[dt][1], 0, 0, 0);
    }
  }
}

// 32 rows x 768 columns (rows clamped past `valid`) from global to LDS [32][LDQ]; all loads issued before the stores
__device__ __forceinline__ void xattn_stage_rows(bf16_t* dst, const bf16_t* __restrict__ src, int64_t ld, int row0, int valid, int tid) {
  Pack<bf16_t, 8> v[6];
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int c = tid + it * 512, row = c / (XD / 8), ch = c % (XD / 8);
    v[it] = ld_pack<bf16_t, 8>(src + (int64_t)min(row0 + row, valid - 1) * ld + ch * 8);
  }
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int c = tid + it * 512, row = c / (XD / 8), ch = c % (XD / 8);
    st_pack<bf16_t, 8>(dst + row * LDQ + ch * 8, v[it]);
  }
}

__global__ __launch_bounds__(512) void xattn_fwd_kernel(XattnArgs a) {
  // LDS: Qs [32][776] (phase 1) and the 8 private V slabs (phase 2) share one region; Ps, Ms, reductions beside it
  constexpr int SLAB = 2 * 32 * LDV;  // bf16 elements per wave
  constexpr int REGION = (XQ * LDQ > XW * SLAB) ? XQ * LDQ : XW * SLAB;
  __shared__ __attribute__((aligned(16))) bf16_t region[REGION];
  __shared__ __attribute__((aligned(16))) bf16_t Ps[XQ * LDP];
  __shared__ __attribute__((aligned(16))) float Ms[256];
  __shared__ float red[XW][XQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.y, q0 = blockIdx.x * XQ;
  const int nk32 = (a.Lk + 31) / 32, nt16 = nk32 * 2;
  bf16_t* Qs = region;
  {  // stage the 32 query rows (clamped past Lq) and the key mask
    xattn_stage_rows(Qs, a.q + b * a.sqb, a.ldq, q0, a.Lq, tid);
    for (int key = tid; key < 256; key += 512)
      Ms[key] = key < a.Lk ? (a.mask ? a.mask[(int64_t)b * a.Lk + key] : 0.f) : -INFINITY;
  }
  __syncthreads();
  f32x4 s[2][2];
  xattn_scores(a.k + b * a.skb, a.ldk, a.Lk, Qs, wave, nt16, fr, fq, s);
  // ---- softmax over keys for query columns (qt*16 + fr): in-lane, across the 4 lane groups, across the 8 waves --
  float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int t = wave + ti * XW;
    if (t >= nt16) continue;
    const f32x4 m4 = *reinterpret_cast<const f32x4*>(&Ms[t * 16 + fq * 4]);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[ti][qt][r] = s[ti][qt][r] * a.scale + m4[r];
        mx[qt] = fmaxf(mx[qt], s[ti][qt][r]);
      }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    mx[qt] = group4_max(mx[qt]);
    if (fq == 0) red[wave][qt * 16 + fr] = mx[qt];
  }
  __syncthreads();
  float sum[2] = {0.f, 0.f};
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float m = red[0][qt * 16 + fr];
#pragma unroll
    for (int w = 1; w < XW; ++w) m = fmaxf(m, red[w][qt * 16 + fr]);
    mx[qt] = m;
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    if (wave + ti * XW >= nt16) continue;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[ti][qt][r] = __expf(s[ti][qt][r] - mx[qt]);
        sum[qt] += s[ti][qt][r];
      }
  }
  __syncthreads();  // everyone has read the maxima: `red` is reused for the sums
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    sum[qt] = group4_sum(sum[qt]);
    if (fq == 0) red[wave][qt * 16 + fr] = sum[qt];
  }
  __syncthreads();
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float t = red[0][qt * 16 + fr];
#pragma unroll
    for (int w = 1; w < XW; ++w) t += red[w][qt * 16 + fr];
    sum[qt] = t;
    if (wave == 0 && fq == 0 && q0 + qt * 16 + fr < a.Lq) a.lse[(int64_t)b * a.Lq + q0 + qt * 16 + fr] = mx[qt] + logf(t);
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int t = wave + ti * XW;
    if (t >= nt16) continue;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const float inv = 1.f / sum[qt];
      Pack<bf16_t, 4> pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk.v[r] = (bf16_t)(s[ti][qt][r] * inv);
      st_pack<bf16_t, 4>(Ps + (qt * 16 + fr) * LDP + t * 16 + fq * 4, pk);
    }
  }
  __syncthreads();  // Ps complete; Qs is dead: its region now holds the V slabs
  f32x4 o[6][2];
  xattn_apply(a.v + b * a.svb, a.ldv, a.Lk, nk32, Ps, region + wave * SLAB, wave, lane, o);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qrow = q0 + qt * 16 + fr;
    if (qrow >= a.Lq) continue;
    bf16_t* Og = a.o + b * a.sob + (int64_t)qrow * a.ldo + wave * XCOLS;
    const bf16_t* Rg = a.res ? a.res + b * a.srb + (int64_t)qrow * a.ldr + wave * XCOLS : nullptr;
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      const int c = dt * 16 + fq * 4;
      Pack<bf16_t, 4> out;
      if (Rg) {
        const Pack<bf16_t, 4> rv = ld_pack<bf16_t, 4>(Rg + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) out.v[r] = (bf16_t)(o[dt][qt][r] + (float)rv.v[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) out.v[r] = (bf16_t)o[dt][qt][r];
      }
      st_pack<bf16_t, 4>(Og + c, out);
    }
  }
}

// Backward, first half: dS and P for 32 query rows (both to HBM as bf16 [B, Lq, lkp] for the two key-side products
// dK = dS^T Q and dV = P^T dO, which are plain batched GEMMs), and dQ = dS K in the same launch.
__global__ __launch_bounds__(512) void xattn_bwd_kernel(XattnArgs a) {
  constexpr int SLAB = 2 * 32 * LDV;
  constexpr int REGION = (2 * XQ * LDQ > XW * SLAB) ? 2 * XQ * LDQ : XW * SLAB;
  __shared__ __attribute__((aligned(16))) bf16_t region[REGION];
  __shared__ __attribute__((aligned(16))) bf16_t DSs[XQ * LDP];
  __shared__ __attribute__((aligned(16))) float Ms[256];
  __shared__ float red[XW][XQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.y, q0 = blockIdx.x * XQ;
  const int nk32 = (a.Lk + 31) / 32, nt16 = nk32 * 2;
  bf16_t *Qs = region, *Gs = region + XQ * LDQ;
  {
    xattn_stage_rows(Qs, a.q + b * a.sqb, a.ldq, q0, a.Lq, tid);
    xattn_stage_rows(Gs, a.dO + b * a.sgb, a.ldg, q0, a.Lq, tid);
    for (int key = tid; key < 256; key += 512)
      Ms[key] = key < a.Lk ? (a.mask ? a.mask[(int64_t)b * a.Lk + key] : 0.f) : -INFINITY;
  }
  __syncthreads();
  f32x4 s[2][2], dp[2][2];
  xattn_scores(a.k + b * a.skb, a.ldk, a.Lk, Qs, wave, nt16, fr, fq, s);
  xattn_scores(a.v + b * a.svb, a.ldv, a.Lk, Gs, wave, nt16, fr, fq, dp);
  float lse[2], dsum[2] = {0.f, 0.f};
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qrow = q0 + qt * 16 + fr;
    lse[qt] = qrow < a.Lq ? a.lse[(int64_t)b * a.Lq + qrow] : INFINITY;  // padded query rows: p = 0
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int t = wave + ti * XW;
    if (t >= nt16) continue;
    const f32x4 m4 = *reinterpret_cast<const f32x4*>(&Ms[t * 16 + fq * 4]);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[ti][qt][r] = __expf(s[ti][qt][r] * a.scale + m4[r] - lse[qt]);
        dsum[qt] += s[ti][qt][r] * dp[ti][qt][r];
      }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    dsum[qt] = group4_sum(dsum[qt]);
    if (fq == 0) red[wave][qt * 16 + fr] = dsum[qt];
  }
  __syncthreads();
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float t = red[0][qt * 16 + fr];
#pragma unroll
    for (int w = 1; w < XW; ++w) t += red[w][qt * 16 + fr];
    dsum[qt] = t;
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int t = wave + ti * XW;
    if (t >= nt16) continue;
    const int col = t * 16 + fq * 4;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      Pack<bf16_t, 4> pk, dk;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pk.v[r] = (bf16_t)s[ti][qt][r];
        dk.v[r] = (bf16_t)(s[ti][qt][r] * (dp[ti][qt][r] - dsum[qt]) * a.scale);
      }
      st_pack<bf16_t, 4>(DSs + (qt * 16 + fr) * LDP + col, dk);
      const int qrow = q0 + qt * 16 + fr;
      if (qrow < a.Lq && col < a.lkp) {
        const int64_t off = ((int64_t)b * a.Lq + qrow) * a.lkp + col;
        st_pack<bf16_t, 4>(a.p_out + off, pk);
        st_pack<bf16_t, 4>(a.ds_out + off, dk);
      }
    }
  }
  __syncthreads();  // DSs complete; Qs / Gs are dead: the region now holds the K slabs
  f32x4 dq[6][2];
  xattn_apply(a.k + b * a.skb, a.ldk, a.Lk, nk32, DSs, region + wave * SLAB, wave, lane, dq);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qrow = q0 + qt * 16 + fr;
    if (qrow >= a.Lq) continue;
    bf16_t* Dg = a.dq + b * a.sdqb + (int64_t)qrow * a.lddq + wave * XCOLS;
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      Pack<bf16_t, 4> out;
#pragma unroll
      for (int r = 0; r < 4; ++r) out.v[r] = (bf16_t)dq[dt][qt][r];
      st_pack<bf16_t, 4>(Dg + dt * 16 + fq * 4, out);
    }
  }
}

}  // namespace

int d2r_xattn2_fwd_try(const void* q, int64_t ldq, int64_t sqb, const void* k, int64_t ldk, int64_t skb, const void* v, int64_t ldv,
                       int64_t svb, void* o, int64_t ldo, int64_t sob, const void* residual, int64_t ldr, int64_t srb, const float* mask,
                       float* lse, int B, int Lq, int Lk, float scale, hipStream_t st);  // xattn2.hip

extern "C" int d2r_xattn_supported(int dtype, int Lq, int Lk, int D) {
  return dtype == D2R_BF16 && D == 768 && Lq >= 1 && Lk >= 1 && Lk <= 256;
}

extern "C" int d2r_xattn_fwd(int dtype, const void* q, int64_t ldq, int64_t sqb, const void* k, int64_t ldk, int64_t skb,
                             const void* v, int64_t ldv, int64_t svb, void* o, int64_t ldo, int64_t sob,
                             const void* residual, int64_t ldr, int64_t srb, const float* mask, float* lse, int B, int Lq,
                             int Lk, int D, float scale, void* stream) {
  D2R_REQUIRE(d2r_xattn_supported(dtype, Lq, Lk, D), "d2r_xattn_fwd: unsupported (bf16, D = 768, Lk <= 256 only)");
  D2R_REQUIRE(B >= 1 && lse, "d2r_xattn_fwd: bad arguments");
  D2R_REQUIRE(aligned_slice(q, ldq, sqb, 768) && aligned_slice(k, ldk, skb, 768) && aligned_slice(v, ldv, svb, 768) &&
                  aligned_slice(o, ldo, sob, 768) && (!residual || aligned_slice(residual, ldr, srb, 768)),
              "d2r_xattn_fwd: pointers must be 16-byte aligned, strides multiples of 8 elements");
  static const int use_v2 = getenv("D2R_XATTN2") ? atoi(getenv("D2R_XATTN2")) : 1;
  if (use_v2 && d2r_xattn2_fwd_try(q, ldq, sqb, k, ldk, skb, v, ldv, svb, o, ldo, sob, residual, ldr, srb, mask, lse, B, Lq, Lk, scale,
                                   (hipStream_t)stream))
    return d2r_check_launch("d2r_xattn_fwd(v2)");
  XattnArgs a = {};
  a.q = (const bf16_t*)q, a.k = (const bf16_t*)k, a.v = (const bf16_t*)v, a.res = (const bf16_t*)residual, a.o = (bf16_t*)o;
  a.mask = mask, a.lse = lse;
  a.ldq = ldq, a.sqb = sqb, a.ldk = ldk, a.skb = skb, a.ldv = ldv, a.svb = svb, a.ldo = ldo, a.sob = sob, a.ldr = ldr, a.srb = srb;
  a.B = B, a.Lq = Lq, a.Lk = Lk, a.scale = scale;
  hipLaunchKernelGGL(xattn_fwd_kernel, dim3(d2r_cdiv(Lq, XQ), B), dim3(512), 0, (hipStream_t)stream, a);
  return d2r_check_launch("d2r_xattn_fwd");
}

extern "C" int d2r_xattn_bwd(int dtype, const void* q, int64_t ldq, int64_t sqb, const void* k, int64_t ldk, int64_t skb,
                             const void* v, int64_t ldv, int64_t svb, const void* dO, int64_t ldg, int64_t sgb,
                             const float* mask, const float* lse, void* dq, int64_t lddq, int64_t sdqb, void* P, void* dS,
                             int lkp, int B, int Lq, int Lk, int D, float scale, void* stream) {
  D2R_REQUIRE(d2r_xattn_supported(dtype, Lq, Lk, D), "d2r_xattn_bwd: unsupported (bf16, D = 768, Lk <= 256 only)");
  D2R_REQUIRE(B >= 1 && lse && P && dS && lkp >= Lk && lkp % 8 == 0 && lkp <= 256, "d2r_xattn_bwd: bad arguments (lkp: Lk padded to a multiple of 8)");
  D2R_REQUIRE(aligned_slice(q, ldq, sqb, 768) && aligned_slice(k, ldk, skb, 768) && aligned_slice(v, ldv, svb, 768) &&
                  aligned_slice(dO, ldg, sgb, 768) && aligned_slice(dq, lddq, sdqb, 768) && d2r_aligned16(P) && d2r_aligned16(dS),
              "d2r_xattn_bwd: pointers must be 16-byte aligned, strides multiples of 8 elements");
  XattnArgs a = {};
  a.q = (const bf16_t*)q, a.k = (const bf16_t*)k, a.v = (const bf16_t*)v, a.dO = (const bf16_t*)dO, a.dq = (bf16_t*)dq;
  a.p_out = (bf16_t*)P, a.ds_out = (bf16_t*)dS, a.lkp = lkp;
  a.mask = mask, a.lse = const_cast<float*>(lse);
  a.ldq = ldq, a.sqb = sqb, a.ldk = ldk, a.skb = skb, a.ldv = ldv, a.svb = svb, a.ldg = ldg, a.sgb = sgb, a.lddq = lddq, a.sdqb = sdqb;
  a.B = B, a.Lq = Lq, a.Lk = Lk, a.scale = scale;
  hipLaunchKernelGGL(xattn_bwd_kernel, dim3(d2r_cdiv(Lq, XQ), B), dim3(512), 0, (hipStream_t)stream, a);
  return d2r_check_launch("d2r_xattn_bwd");
}
