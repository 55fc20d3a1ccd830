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
