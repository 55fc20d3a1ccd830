// xattn3.hip — K2 / K4, third generation: single-head attention over the full 768-wide feature with the QUERIES split over the
// waves (flash-attention shape at head dimension 768), forward and backward.
//
//   O[b] = softmax(scale * Q K^T + mask[b]) V (+ residual)         Q [B,Lq,768], K / V [B,Lk,768], Lk <= 256
//
// Replaces (reference file:line) the CrossModalAlignment core models/XModules.py:300-310 (= models/Refinement.py:105-115,
// scale 100/sqrt(768)) and the ContextRichCrossModalCell core models/Cells.py:244-246 (unscaled, residual Qs), and their
// autograd backward.
//
// Why a third structure.  The op moves B(2Lq+2Lk)*768*2 bytes for 4*B*Lq*Lk*768 flops (78 flop/B): HBM bound on paper.  The second
// generation (xattn2.hip) split the 768-deep contraction over eight waves and summed the partial score tiles through LDS: two
// workgroup barriers per 16-key chunk, S and P staged in LDS (50 KB) next to a 4-slot ring, 32 queries per workgroup (every
// sample's K and V streamed four to seven times out of L2) — and, found in round 3 in the ISA, hipcc puts `s_waitcnt vmcnt(0)`
// in front of the first ORDINARY ds_read that follows an LDS-DMA in flight (it cannot tell which LDS bytes the DMA writes), so
// the ring never actually ran ahead: every chunk paid the full memory latency.  Measured (rocprofv3, three problems per launch):
// 54-59 us for 96 MB of algorithmic traffic = 22 % of 8 TB/s.  Here:
//   * a 256-thread workgroup (one wave per SIMD, up to 512 registers per lane) owns 64 queries of one (problem, sample); a WAVE
//     owns 16 of them for the whole kernel: its Q (and dO) rows live in registers as MFMA B fragments (96 VGPRs each), its score
//     tiles S^T = K Q^T stay in registers (lane = one query: the softmax is an in-lane loop plus two shuffles), and its output
//     O^T = V^T P^T [768 x 16] accumulates in 192 registers: nothing of the softmax ever touches LDS and no wave waits for
//     another except on the ring;
//   * LDS holds ONLY the K / V ring: six slots of 16 keys x 1536 B filled by LDS-DMA (global_load_lds_dwordx4, the swizzle
//     applied to the per-lane SOURCE address), up to five chunks = 120 KB in flight per CU behind counted vmcnt waits, one barrier
//     per chunk; EVERY LDS read while the stream runs is inline asm (ds_read_b128 / ds_read_b64_tr_b16 with hand-counted
//     lgkmcnt waits), invisible to the compiler's wait insertion;
//   * every query tile of a (problem, sample) runs on one XCD: K and V cross the fabric once and are re-read out of that XCD's L2;
//   * backward, query side: the same ring carrying K_t, V_t pairs; S^T = K Q^T, dP^T = V dO^T, D = rowsum(dO o (O - residual)) from
//     the saved output, P and dS leave as 16-bit [B, Lq, lkp];
//   * backward, product side (one launch): dV = P^T dO, dK = dS^T Q and dQ = dS K for one (problem, sample) per workgroup: the
//     small matrix (P / dS, at most 256 x 256) resident in LDS — its fragments then in registers —, the wide one streamed in
//     64-column chunks, output tiles of 16 rows split over eight waves.
// Deterministic: fixed summation order, no atomics.
#include <math.h>
#include <stdlib.h>

#include "gemm_args.h"

#ifndef D2R_X3_PROBES
#define D2R_X3_PROBES 0
#endif
constexpr bool X3_PROBES = D2R_X3_PROBES != 0;

namespace {

constexpr int X3_MAXCORE = 4;
constexpr int XE = 768, CH = 16, ROWB = XE * 2 /*1536*/, CB = CH * ROWB /*24576*/, NS = 6, NW = 4, QW = 16, QT = NW * QW /*64*/;
constexpr int NKK = XE / 32 /*24 k-steps of the 768-deep contraction*/, NDT = XE / 16 /*48 output column tiles*/;

template <typename E>
struct X3Args {
  const E *q[X3_MAXCORE], *k[X3_MAXCORE], *v[X3_MAXCORE], *res[X3_MAXCORE], *dO[X3_MAXCORE];
  E *o[X3_MAXCORE], *p_out[X3_MAXCORE], *ds_out[X3_MAXCORE];
  float* lse[X3_MAXCORE];
  const float* mask;
  int64_t ldq, sqb, ldk, skb, ldv, svb, ldo, sob, ldr, srb, ldg, sgb;
  int B, Lq, Lk, lkp, ntile, ncore;
  float scale;
  int dbg;  // timing experiments only (D2R_X3_DBG): 1 = no DMA issue, 2 = no LDS fragment reads / MFMAs, 3 = DMA + waits only in phase V
  unsigned long long* ts;  // timing experiments only: cycle stamps of block 0's waves (forward kernel), [4 waves][64]
};

template <typename P>
__device__ __forceinline__ P x3_pick(P const (&arr)[X3_MAXCORE], int core) {  // wave-uniform select chain (no scratch)
  P r = arr[0];
#pragma unroll
  for (int c = 1; c < X3_MAXCORE; ++c) r = core == c ? arr[c] : r;
  return r;
}

// position of 16-byte chunk c (0..95) of ring row r: XOR inside aligned groups of 16 chunks (256 B = one bank row)
__device__ __forceinline__ int swz(int c, int r) { return (c & ~15) | ((c & 15) ^ (r & 15)); }
// ... of a chunk that is read with ds_read_b64_tr_b16 (forward kernel, V chunks): a 32-lane group of that read covers rows 0-7 or 8-15,
// 32 contiguous bytes (two chunks) of each: the XOR acts on chunk PAIRS, or rows r and r ^ 1 would land on the same banks (measured:
// SQ_LDS_BANK_CONFLICT = a third of the forward kernel's LDS cycles with the row swizzle above)
__device__ __forceinline__ int swz_tr(int c, int r) { return (c & ~15) | ((c & 15) ^ ((r & 7) << 1)); }

__device__ __forceinline__ float g4max(float v) {  // over the 4 lane groups holding one MFMA column
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float g4sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ---- LDS reads the compiler does not see (see the header): the caller counts lgkmcnt itself -----------------------------------------
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
template <typename V8>
__device__ __forceinline__ V8 lds_b128(unsigned addr) {
  V8 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
template <typename V4>
__device__ __forceinline__ V4 lds_tr(unsigned addr) {
  V4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {  // all but the N youngest LDS reads have returned (they return in order)
  if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
  else static_assert(N < 0, "unsupported count");
  __builtin_amdgcn_sched_barrier(0);  // (rule 18: a register-only MFMA must not be hoisted above the wait)
}
__device__ __forceinline__ void wait_vm_n(int n) {  // s_waitcnt vmcnt(n), n even in [0, 28]; odd values round DOWN (safe)
  switch (n >> 1) {
    case 14: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// Block -> (problem, sample, query tile) with every tile of a (problem, sample) on one XCD (blocks b and b + 8 share one)
__device__ __forceinline__ bool x3_decode(int ntile, int ncore, int B, int& core, int& b, int& tile) {
  const int id = blockIdx.x, xcd = id & 7, rr = id >> 3;
  tile = rr % ntile;
  const int unit = rr / ntile;
  core = unit % ncore;
  b = (unit / ncore) * 8 + xcd;
  return b < B;
}

// per-lane byte offsets inside a ring slot.  Row fragment of k-step kk (16 keys x 32 features; lane (fr, fq) reads 16 bytes of key
// row fr): row[kk & 3] + (kk >> 2) * 256.  Transposed fragment of column tile dt (lane (tq, tp) of group fq addresses row
// fq*4 + tq, 8 bytes): tr[dt & 7] + (dt >> 3) * 256.
struct RingOffs {
  unsigned row[4], tr[8];
};
__device__ __forceinline__ RingOffs ring_offs(int fr, int fq, int tq, int tp) {
  RingOffs r;
#pragma unroll
  for (int m = 0; m < 4; ++m) r.row[m] = fr * ROWB + (((m * 4 + fq) ^ (fr & 15)) << 4);
  const int vrow = fq * 4 + tq;
#pragma unroll
  for (int m = 0; m < 8; ++m) r.tr[m] = vrow * ROWB + (((m * 2 + (tp >> 1)) ^ ((vrow & 7) << 1)) << 4) + (tp & 1) * 8;
  return r;
}

// ================================================================================================================================
// forward
// ================================================================================================================================
template <typename E, int LKMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void xattn3_fwd_kernel(X3Args<E> a) {
  typedef typename H16<E>::v8 E8;
  typedef typename H16<E>::v4 E4;
  constexpr int NT = LKMAX / CH;  // key tiles held in registers
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * CB + LKMAX * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  int core, b, tile;
  if (!x3_decode(a.ntile, a.ncore, a.B, core, b, tile)) return;
  const int q0 = tile * QT + wave * QW;
  const E* Kg = x3_pick(a.k, core) + b * a.skb;
  const E* Vg = x3_pick(a.v, core) + b * a.svb;
  float* Ms = reinterpret_cast<float*>(smem + NS * CB);
  const int nkc = (a.Lk + CH - 1) / CH, G = 2 * nkc;
  // (the probes' hooks - cycle stamps, ablation switches - are compiled in only with -DD2R_X3_PROBES=1, D2R_X3_PROBES=1 python -m
  //  d2r_amd.build: as run-time branches they cost forty 64-bit registers and a basic-block boundary per tile in the production kernel)
  unsigned long long tsv[X3_PROBES ? 40 : 1];
  const int dbg = X3_PROBES ? a.dbg : 0;
  const bool stamping = X3_PROBES && a.ts != nullptr && blockIdx.x == 0;
#define X3_STAMP(i) do { if (X3_PROBES && stamping) tsv[X3_PROBES ? (i) : 0] = __builtin_amdgcn_s_memtime(); } while (0)
  X3_STAMP(0);

  for (int key = tid; key < LKMAX; key += 256) Ms[key] = key < a.Lk ? (a.mask ? a.mask[(int64_t)b * a.Lk + key] : 0.f) : -INFINITY;
  // this wave's 16 query rows as B fragments of the score product: lane (fr, fq) holds Q[q0 + fr][32 kk + 8 fq .. + 7]
  E8 qf[NKK];
  {
    const int qrow = min(q0 + fr, a.Lq - 1);
    const E* qp = x3_pick(a.q, core) + b * a.sqb + (int64_t)qrow * a.ldq + fq * 8;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = *reinterpret_cast<const E8*>(qp + kk * 32);
  }
  // per-lane source of this wave's six 1-KiB pieces of a chunk image [16 rows][1536 B]
  int prow[6], pcol[6], pcolv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int o = (wave * 6 + i) * 1024 + lane * 16;
    const int row = o / ROWB, cp = (o - row * ROWB) >> 4;
    prow[i] = row;
    pcol[i] = swz(cp, row) * 8;  // the swizzle is an involution: image position cp of row `row` holds source chunk swz(cp,row)
    pcolv[i] = swz_tr(cp, row) * 8;  // V chunks: the image the transposing reads want
  }
  // pieces [i0, i1) of chunk g  (issuing the six pieces of a chunk two at a time between the MFMA batches of a step measured SLOWER
  // than all six right behind the barrier: 48 vs 40 us for three problems of the text branch)
  auto issue_pieces = [&](int g, int i0, int i1) {
    if (dbg == 1 || g >= G) return;
    const bool isv = g >= nkc;
    const E* src = isv ? Vg : Kg;
    const int ld = (int)(isv ? a.ldv : a.ldk);  // (per-sample offsets fit 32 bits: checked on the host)
    const int key0 = (isv ? g - nkc : g) * CH;
    unsigned char* base = smem + (g % NS) * CB;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i < i0 || i >= i1) continue;
      const int key = min(key0 + prow[i], a.Lk - 1);  // clamped: masked (-inf) scores / zero probabilities for keys >= Lk
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (key * ld + (isv ? pcolv[i] : pcol[i]))),
                                       (__attribute__((address_space(3))) void*)(base + (wave * 6 + i) * 1024), 16, 0, 0);
    }
  };
  auto issue = [&](int g) { issue_pieces(g, 0, 6); };
  // chunk g has landed for THIS wave when at most `ahead` younger chunks (6 pieces each) are outstanding; the barrier makes it
  // visible to every wave and tells the issuing side that everybody has left chunk g - 1
  auto ring_wait = [&](int g) {
    wait_vm_n(6 * min(NS - 2, G - 1 - g));
    asm volatile("s_barrier" ::: "memory");
  };
  const RingOffs ro = ring_offs(fr, fq, tq, tp);
  const unsigned sbase = lds_addr(smem), mbase = lds_addr(Ms) + fq * 16;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // Q fragments and mask values have landed: only DMA pieces in the queue from here
  __syncthreads();
  X3_STAMP(1);
#pragma unroll
  for (int g = 0; g < NS - 1; ++g)
    if (g < G) issue(g);

  // ---- phase K: S^T tile t = K_t Q^T (rows = keys t*16 + fq*4 + r, column = query fr), all tiles kept in registers ----------------
  f32x4 s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    s[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (t < nkc) {
      ring_wait(t);
      const int gn = t + NS - 1;  // next chunk, into the slot of chunk t - 1: every wave left it before the barrier above
      const unsigned slot = sbase + (t % NS) * CB;
      if (dbg == 2) { issue(gn); s[t] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
      const f32x4 m4 = lds_b128<f32x4>(mbase + t * CH * 4);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      E8 kf[2][8];  // three batches of eight k-steps, the next batch in flight behind the MFMAs of the current one
      auto fetch = [&](int k0, E8 (&dst)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = lds_b128<E8>(slot + ro.row[(k0 + j) & 3] + ((k0 + j) >> 2) * 256);
      };
      issue(gn);
      fetch(0, kf[0]);
      fetch(8, kf[1]);
      wait_lgkm<8>();
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = H16<E>::mfma32(kf[0][j], qf[j], acc);
      __builtin_amdgcn_sched_barrier(0);
      fetch(16, kf[0]);
      wait_lgkm<8>();
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = H16<E>::mfma32(kf[1][j], qf[8 + j], acc);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm<0>();
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = H16<E>::mfma32(kf[0][j], qf[16 + j], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[t][r] = acc[r] * a.scale + m4[r];
      X3_STAMP(2 + t);
    }
  }
  // ---- softmax over the keys of query fr: in-lane over tiles and rows, then across the four lane groups ---------------------------
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
  mx = g4max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[t][r] = __expf(s[t][r] - mx);
      sum += s[t][r];
    }
  sum = g4sum(sum);
  const float inv = 1.f / sum;
  E4 pf[NT];  // P^T tiles as B fragments of v_mfma_f32_16x16x16: lane (fr, fq) holds P[query fr][keys t*16 + fq*4 .. + 3]
#pragma unroll
  for (int t = 0; t < NT; ++t) pf[t] = E4{(E)(s[t][0] * inv), (E)(s[t][1] * inv), (E)(s[t][2] * inv), (E)(s[t][3] * inv)};

  // ---- phase V: O^T[d][q] += V^T P^T, all 48 column tiles of the 768-wide output per wave, TWO key tiles per step ----------------
  // v_mfma_f32_16x16x32 contracts over 32 keys: k-slot (fq, j) = key tile 2u, row fq*4 + j for j < 4, key tile 2u+1, row fq*4 + j - 4
  // otherwise - the A operand is then the transposing read of chunk 2u followed by the SAME read of chunk 2u+1, the B operand the two
  // probability packs side by side (half the MFMA issue slots of the 16-key form; an odd last tile is paired with zeros).
  f32x4 o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  X3_STAMP(18);
  int issued = nkc + NS - 2;  // youngest chunk in the queue (phase K issued up to chunk g + NS - 1 at step g)
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
    const int t0 = 2 * u, t1 = 2 * u + 1;
    if (t0 < nkc) {
      const bool two = t1 < nkc;
      const int g0 = nkc + t0, g1 = two ? g0 + 1 : g0;
      wait_vm_n(6 * (min(issued, G - 1) - g1));  // chunk g1 (and every older one) has landed for this wave
      asm volatile("s_barrier" ::: "memory");    // ... for every wave, and everybody has left the chunks before g0
      for (int gn = issued + 1; gn <= g0 + NS - 1; ++gn) issue(gn);  // chunk x goes into the slot chunk x - NS has left
      issued = max(issued, g0 + NS - 1);
      if (dbg == 2 || dbg == 3) continue;
      const unsigned slot0 = sbase + (g0 % NS) * CB, slot1 = sbase + (g1 % NS) * CB;
      E8 pp;
#pragma unroll
      for (int j = 0; j < 4; ++j) pp[j] = pf[t0][j], pp[4 + j] = two ? pf[t1][j] : (E)0.f;
      constexpr int NB = 6;  // column tiles per batch: 12 reads, the next batch in flight behind the MFMAs (lgkmcnt counts to 15)
      E4 va[2][NB], vb[2][NB];
      auto fetch = [&](int d0, E4 (&da)[NB], E4 (&db)[NB]) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const unsigned off = ro.tr[(d0 + i) & 7] + ((d0 + i) >> 3) * 256;
          da[i] = lds_tr<E4>(slot0 + off);
          db[i] = lds_tr<E4>(slot1 + off);
        }
      };
      fetch(0, va[0], vb[0]);
#pragma unroll
      for (int bq = 0; bq < NDT / NB; ++bq) {
        if (bq + 1 < NDT / NB) {
          fetch((bq + 1) * NB, va[(bq + 1) & 1], vb[(bq + 1) & 1]);
          wait_lgkm<12>();
        } else {
          wait_lgkm<0>();
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          E8 vv;
#pragma unroll
          for (int j = 0; j < 4; ++j) vv[j] = va[bq & 1][i][j], vv[4 + j] = vb[bq & 1][i][j];
          o[bq * NB + i] = H16<E>::mfma32(vv, pp, o[bq * NB + i]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      X3_STAMP(19 + u);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave left the ring: it now stages the output tile
  X3_STAMP(35);
  // ---- epilogue: o[dt][r] = O[q = fr][d = dt*16 + fq*4 + r] -> this wave's 16 LDS rows -> whole-row global stores -------------------
  constexpr int LDO = XE + 8;
  E* Os = reinterpret_cast<E*>(smem) + wave * QW * LDO;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    Pack<E, 4> pk;
#pragma unroll
    for (int r = 0; r < 4; ++r) pk.v[r] = (E)o[dt][r];
    st_pack<E, 4>(Os + fr * LDO + dt * 16 + fq * 4, pk);
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();  // (the 16 rows are private to this wave)
  const E* Rg = x3_pick(a.res, core);
  E* Og = x3_pick(a.o, core) + b * a.sob;
  for (int e = lane; e < QW * (XE / 8); e += 64) {
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
  if (fq == 0 && q0 + fr < a.Lq) x3_pick(a.lse, core)[(int64_t)b * a.Lq + q0 + fr] = mx + logf(sum);
  if constexpr (X3_PROBES) if (stamping) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tsv[36] = __builtin_amdgcn_s_memtime();
    if (lane == 0)
      for (int i = 0; i < 37; ++i) a.ts[wave * 64 + i] = tsv[i];
  }
#undef X3_STAMP
}

// ================================================================================================================================
// backward, query side: dS and P (16-bit [B, Lq, lkp]) for the product kernel below
// ================================================================================================================================
// Key tile t needs K_t (scores) and V_t (dP) together: the ring carries K_0, V_0, K_1, V_1, ... as 24-KB chunks (chunk 2t = K_t,
// 2t+1 = V_t); two chunks are issued per step into the slots of tile t - 1.
template <typename E, int LKMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void xattn3_bwd_kernel(X3Args<E> a) {
  typedef typename H16<E>::v8 E8;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * CB + LKMAX * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  int core, b, tile;
  if (!x3_decode(a.ntile, a.ncore, a.B, core, b, tile)) return;
  const int q0 = tile * QT + wave * QW;
  const E* Kg = x3_pick(a.k, core) + b * a.skb;
  const E* Vg = x3_pick(a.v, core) + b * a.svb;
  float* Ms = reinterpret_cast<float*>(smem + NS * CB);
  const int nkc = (a.Lk + CH - 1) / CH, G = 2 * nkc;
  for (int key = tid; key < LKMAX; key += 256) Ms[key] = key < a.Lk ? (a.mask ? a.mask[(int64_t)b * a.Lk + key] : 0.f) : -INFINITY;
  // this wave's 16 rows of Q and dO as B fragments, and D = rowsum(dO o O) from the saved forward output (minus the residual the
  // forward added): lane (fr, fq) holds row q0 + fr, columns 32 kk + 8 fq .. + 7
  const int qrow = min(q0 + fr, a.Lq - 1);
  const bool qok = q0 + fr < a.Lq;
  E8 qf[NKK], gf[NKK];
  float dsum = 0.f;
  {
    const E* qp = x3_pick(a.q, core) + b * a.sqb + (int64_t)qrow * a.ldq + fq * 8;
    const E* gp = x3_pick(a.dO, core) + b * a.sgb + (int64_t)qrow * a.ldg + fq * 8;
    const E* op = x3_pick(a.o, core) + b * a.sob + (int64_t)qrow * a.ldo + fq * 8;
    const E* rbase = x3_pick(a.res, core);
    const E* rp = rbase ? rbase + b * a.srb + (int64_t)qrow * a.ldr + fq * 8 : op;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) gf[kk] = *reinterpret_cast<const E8*>(gp + kk * 32);
    // (in groups of six k-steps, fenced: hoisting all loads of O and the residual next to Q and dO would spill)
#pragma unroll
    for (int k6 = 0; k6 < NKK; k6 += 6) {
      E8 of[6], rf[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        of[j] = *reinterpret_cast<const E8*>(op + (k6 + j) * 32);
        rf[j] = *reinterpret_cast<const E8*>(rp + (k6 + j) * 32);  // (no residual: rp == op, the term is dropped by the select below)
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += (float)gf[k6 + j][e] * ((float)of[j][e] - (rbase ? (float)rf[j][e] : 0.f));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = *reinterpret_cast<const E8*>(qp + kk * 32);
  }
  dsum = g4sum(dsum);
  const float lse = qok ? x3_pick(a.lse, core)[(int64_t)b * a.Lq + qrow] : INFINITY;  // padded query rows: p = 0
  int prow[6], pcol[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int o = (wave * 6 + i) * 1024 + lane * 16;
    const int row = o / ROWB, cp = (o - row * ROWB) >> 4;
    prow[i] = row;
    pcol[i] = swz(cp, row) * 8;
  }
  auto issue = [&](int g) {  // chunk g: K (even) or V (odd) rows of key tile g / 2
    const bool isv = g & 1;
    const E* src = isv ? Vg : Kg;
    const int ld = (int)(isv ? a.ldv : a.ldk);
    const int key0 = (g >> 1) * CH;
    unsigned char* base = smem + (g % NS) * CB;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int key = min(key0 + prow[i], a.Lk - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (key * ld + pcol[i])),
                                       (__attribute__((address_space(3))) void*)(base + (wave * 6 + i) * 1024), 16, 0, 0);
    }
  };
  const RingOffs ro = ring_offs(fr, fq, tq, tp);
  const unsigned sbase = lds_addr(smem), mbase = lds_addr(Ms) + fq * 16;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // fragments, lse and mask values have landed
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 4; ++g)
    if (g < G) issue(g);

  E* Pout = x3_pick(a.p_out, core);
  E* DSout = x3_pick(a.ds_out, core);
  const int nst = q0 < a.Lq ? 2 : 0;  // store instructions this wave executes per step
  for (int t = 0; t < nkc; ++t) {
    // Queue of this wave, oldest first, when step t waits:  [chunks 2t, 2t+1] [stores of step t-2] [chunks 2t+2, 2t+3]
    // [stores of step t-1]  (chunks 2t+2, 2t+3 went out at step t-1 right behind its barrier, every step ends with two 8-byte
    // stores; vmcnt counts stores too).  Chunks 2t and 2t+1 have landed when only what is younger is outstanding.
    // A wave whose 16 queries all lie past Lq may execute no store at all: it counts none (a smaller count only waits longer).
    wait_vm_n(t + 1 < nkc ? 12 + min(t, 2) * nst : 0);  // (last tile: everything must land anyway)
    asm volatile("s_barrier" ::: "memory");
    if (2 * t + 4 < G) issue(2 * t + 4), issue(2 * t + 5);  // into the slots of tile t - 1: every wave left them before the barrier
    const unsigned sk = sbase + ((2 * t) % NS) * CB, sv = sbase + ((2 * t + 1) % NS) * CB;
    const f32x4 m4 = lds_b128<f32x4>(mbase + t * CH * 4);
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
    // K_t / V_t row fragments: six batches of four k-steps of both, the next batch in flight behind the MFMAs of the current one
    E8 kf[2][4], vf[2][4];
    auto fetch = [&](int k0, E8 (&dk)[4], E8 (&dv)[4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned off = ro.row[(k0 + j) & 3] + ((k0 + j) >> 2) * 256;
        dk[j] = lds_b128<E8>(sk + off);
        dv[j] = lds_b128<E8>(sv + off);
      }
    };
    fetch(0, kf[0], vf[0]);
#pragma unroll
    for (int bq = 0; bq < NKK / 4; ++bq) {
      if (bq + 1 < NKK / 4) {
        fetch((bq + 1) * 4, kf[(bq + 1) & 1], vf[(bq + 1) & 1]);
        wait_lgkm<8>();
      } else {
        wait_lgkm<0>();
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sa = H16<E>::mfma32(kf[bq & 1][j], qf[bq * 4 + j], sa);
        da = H16<E>::mfma32(vf[bq & 1][j], gf[bq * 4 + j], da);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    Pack<E, 4> pk, dk;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __expf(sa[r] * a.scale + m4[r] - lse);
      pk.v[r] = (E)p;
      dk.v[r] = (E)(p * (da[r] - dsum) * a.scale);
    }
    // (exactly two store instructions per step and wave, so that the wait above can count them; lanes outside the matrix are
    //  masked off, their address is clamped)
    const int col = t * CH + fq * 4;
    const bool st_ok = qok && col < a.lkp;
    const int64_t off = ((int64_t)b * a.Lq + qrow) * a.lkp + (col < a.lkp ? col : 0);
    if (st_ok) st_pack<E, 4>(Pout + off, pk);
    if (st_ok) st_pack<E, 4>(DSout + off, dk);
  }
}

// ================================================================================================================================
// backward, product side: out_g[b] (16-bit [Md, 768]) = sum_k Wt_g[b](k, m) X_g[b][k, :]
//   trans = 0: Wt(k, m) = W[k * ldw + m]   (dV = P^T dO, dK = dS^T Q: k = query, m = key)
//   trans = 1: Wt(k, m) = W[m * ldw + k]   (dQ = dS K:                 k = key,   m = query)
// ================================================================================================================================
constexpr int DKV_MAXG = 12;
template <typename E>
struct DkvArgs {
  const E *W[DKV_MAXG], *X[DKV_MAXG];
  E* out[DKV_MAXG];
  int64_t ldx[DKV_MAXG], sxb[DKV_MAXG], ldo[DKV_MAXG], sob[DKV_MAXG];
  int Kd[DKV_MAXG], Md[DKV_MAXG], trans[DKV_MAXG];
  int64_t swb;  // batch stride of W (elements): Lq * lkp
  int ldw, B, ngroup;
};
constexpr int DKV_KM = 256 /*rows of the staged image at most*/, DKV_LDW = 272 /*256 columns + 16: 8 consecutive rows tile the 64 banks*/,
              DKV_LDX = 80 /*64 columns + 16*/, DKV_XC = 64;

// A (or B) fragment of M^T for a matrix kept [k][col] in LDS with row stride LD, natural k order: lane (fr, fq) gets
// M[k0 + 8 fq + j][c0 + fr], j = 0..7
template <typename E, int LD>
__device__ __forceinline__ typename H16<E>::v8 x3_frag_tr(const E* base, int k0, int c0, int fq, int tq, int tp) {
  typedef typename H16<E>::v4 E4;
  typedef typename H16<E>::v8 E8;
  const E* p0 = base + (k0 + fq * 8 + tq) * LD + c0 + tp * 4;
  const E4 lo = H16<E>::tr_read(p0);
  const E4 hi = H16<E>::tr_read(p0 + 4 * LD);
  return E8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Row permutation of the LDS images of both product kernels (bit 2 and bit 3 of the row index swap places): the eight rows one
// 32-lane group of ds_read_b64_tr_b16 touches (k0 + {0..3, 8..11}, then + 4) become consecutive; with a row stride of an odd number of
// 32-byte units (272 and 80 elements here) their 32-byte segments fall on eight different bank groups.
__device__ __forceinline__ int dkv_prow(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// lane (fr, fq) gets M[k0 + 8 fq + j][c0 + fr], j = 0..7, from a row-permuted image with row stride LD (elements)
template <typename E>
__device__ __forceinline__ typename H16<E>::v8 dkv_frag(const E* base, int LD, int k0, int c0, int fq, int tq, int tp) {
  typedef typename H16<E>::v4 E4;
  typedef typename H16<E>::v8 E8;
  const E* p0 = base + (k0 + (fq >> 1) * 16 + (fq & 1) * 4 + tq) * LD + c0 + tp * 4;  // = dkv_prow(k0 + fq*8 + tq)
  const E4 lo = H16<E>::tr_read(p0);
  const E4 hi = H16<E>::tr_read(p0 + 8 * LD);                                        // = dkv_prow(k0 + fq*8 + tq + 4)
  return E8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__host__ __device__ __forceinline__ int dkv_ldw(int nmt) {  // row stride of the W image: columns + 16, an odd number of 16-element units
  const int u = nmt + 1;
  return (u | 1) * 16;
}

template <typename E>
__global__ __launch_bounds__(512) void xattn3_dkv_kernel(DkvArgs<E> a) {
  typedef typename H16<E>::v8 E8;
  constexpr int KSM = DKV_KM / 32;  // k-steps (32 rows of the reduction each) at most
  __shared__ __attribute__((aligned(16))) E sm[DKV_KM * DKV_LDW];  // the Wt image [k][m]; afterwards the X chunk image [k][DKV_LDX]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  // the groups of a sample next to each other on one XCD (they read the same P / dS / Q / dO / K out of its L2)
  const int id = blockIdx.x, xcd = id & 7, rr = id >> 3;
  const int g = rr % a.ngroup, b = (rr / a.ngroup) * 8 + xcd;
  if (b >= a.B) return;
  // (g comes from blockIdx: the by-value argument arrays are indexed through scalar loads of the kernarg segment, no scratch)
  const E* Wg = a.W[g] + b * a.swb;
  const int64_t ldx = a.ldx[g], ldo = a.ldo[g];
  const E* Xg = a.X[g] + b * a.sxb[g];
  E* Og = a.out[g] + b * a.sob[g];
  const int Kd = a.Kd[g], Md = a.Md[g], trans = a.trans[g], ldw = a.ldw;
  const int KP = (Kd + 31) / 32 * 32, KS = KP / 32, nmt = (Md + 15) / 16, MP8 = nmt * 2;  // 16-byte packs per staged row
  // ---- Wt (rows >= Kd and columns >= Md zero) -> LDS -> this wave's B fragments (output row tiles wave, wave + 8) -------------------
  if (!trans) {
    for (int idx = tid; idx < KP * MP8; idx += 512) {
      const int row = idx / MP8, ch = idx - row * MP8;
      const bool ok = row < Kd && ch * 8 < ldw;
      Pack<E, 8> v = ld_pack<E, 8>(Wg + (ok ? (int64_t)row * ldw + ch * 8 : 0));
#pragma unroll
      for (int j = 0; j < 8; ++j) v.v[j] = (ok && ch * 8 + j < Md) ? v.v[j] : (E)0.f;
      st_pack<E, 8>(sm + dkv_prow(row) * DKV_LDW + ch * 8, v);
    }
  } else {
    // W is [m][k] in memory: read 16-byte packs along k, store them transposed (eight 2-byte stores)
    const int KP8 = KP / 8, MPAD = nmt * 16;
    for (int idx = tid; idx < MPAD * KP8; idx += 512) {
      const int m = idx / KP8, ch = idx - m * KP8;
      const bool ok = m < Md && ch * 8 < ldw;
      const Pack<E, 8> v = ld_pack<E, 8>(Wg + (ok ? (int64_t)m * ldw + ch * 8 : 0));
#pragma unroll
      for (int j = 0; j < 8; ++j) sm[dkv_prow(ch * 8 + j) * DKV_LDW + m] = (ok && ch * 8 + j < Kd) ? v.v[j] : (E)0.f;
    }
  }
  __syncthreads();
  E8 wf[2][KSM];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) {
      const int n = min(wave + 8 * i, nmt - 1), k0 = min(ks, KS - 1) * 32;  // (clamped: unused fragments read valid LDS)
      wf[i][ks] = dkv_frag<E>(sm, DKV_LDW, k0, n * 16, fq, tq, tp);
    }
  __syncthreads();  // the Wt image is dead: the region now holds the X chunks
  // ---- X in 64-column chunks: registers -> LDS (one chunk ahead in registers), out^T tile = X_chunk^T Wt ---------------------------
  constexpr int NPK = DKV_KM * (DKV_XC / 8) / 512;  // 16-byte packs of a chunk per thread (4)
  Pack<E, 8> xr[NPK];
  auto xload = [&](int c) {
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
      const int idx = tid + i * 512, row = idx >> 3, ch = idx & 7;
      xr[i] = ld_pack<E, 8>(Xg + (int64_t)min(row, Kd - 1) * ldx + c * DKV_XC + ch * 8);  // rows past Kd: finite, times Wt = 0
    }
  };
  auto xstore = [&]() {
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
      const int idx = tid + i * 512, row = idx >> 3, ch = idx & 7;
      if (row < KP) st_pack<E, 8>(sm + dkv_prow(row) * DKV_LDX + ch * 8, xr[i]);
    }
  };
  xload(0);
  for (int c = 0; c < XE / DKV_XC; ++c) {
    xstore();
    __syncthreads();
    if (c + 1 < XE / DKV_XC) xload(c + 1);
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) {
      if (ks < KS) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const E8 af = dkv_frag<E>(sm, DKV_LDX, ks * 32, m * 16, fq, tq, tp);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][m] = H16<E>::mfma32(af, wf[i][ks], acc[i][m]);
        }
      }
    }
    // acc[i][m][r] = out[row = n*16 + fr][d = c*64 + m*16 + fq*4 + r]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave + 8 * i) * 16 + fr;
      if (wave + 8 * i < nmt && row < Md) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          Pack<E, 4> pk;
#pragma unroll
          for (int r = 0; r < 4; ++r) pk.v[r] = (E)acc[i][m][r];
          st_pack<E, 4>(Og + (int64_t)row * ldo + c * DKV_XC + m * 16 + fq * 4, pk);
        }
      }
    }
    __syncthreads();  // everyone is done reading the chunk before it is overwritten
  }
}

// ---- compact product kernel (the shapes of the path: Lq, Lk <= 256 with min(Lq, Lk) <= 128) --------------------------------------------
// Same product as above with (i) at most EIGHT B fragments per wave - two output row tiles x up to four k-steps, or one row tile x
// up to eight - so that the kernel fits 128 VGPRs, (ii) an LDS footprint sized by the launch (<= 64 KB: the W image [KP][LDW] or the
// X chunk image [KP][80]) - two workgroups per CU, so that the 9 x 32 = 288 workgroups of a three-problem launch are all resident
// instead of running as one and an eighth rounds, and (iii) ROW-PERMUTED LDS images: bit 2 and bit 3 of the row index swap places,
// which makes the eight rows one 32-lane group of ds_read_b64_tr_b16 touches (k0 + {0..3, 8..11}, then + 4) consecutive; with a row
// stride of an odd number of 32-byte units their 32-byte segments fall on eight different bank groups (the natural order put rows
// r and r + 8 on the same banks: SQ_LDS_BANK_CONFLICT was 55 % of the LDS cycles).
template <typename E>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void xattn3_dkv2_kernel(DkvArgs<E> a) {
  typedef typename H16<E>::v8 E8;
  extern __shared__ __attribute__((aligned(16))) unsigned char dkv_smem[];
  E* sm = reinterpret_cast<E*>(dkv_smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  const int id = blockIdx.x, xcd = id & 7, rr = id >> 3;
  const int g = rr % a.ngroup, b = (rr / a.ngroup) * 8 + xcd;
  if (b >= a.B) return;
  const E* Wg = a.W[g] + b * a.swb;
  const int64_t ldx = a.ldx[g], ldo = a.ldo[g];
  const E* Xg = a.X[g] + b * a.sxb[g];
  E* Og = a.out[g] + b * a.sob[g];
  const int Kd = a.Kd[g], Md = a.Md[g], trans = a.trans[g], ldw = a.ldw;
  const int KP = (Kd + 31) / 32 * 32, KS = KP / 32, nmt = (Md + 15) / 16, MPAD = nmt * 16, LDW = dkv_ldw(nmt);
  const bool two = nmt > 8;  // two output row tiles per wave (then KS <= 4: checked on the host)
  // ---- Wt (rows >= Kd and columns >= Md zero) -> LDS ------------------------------------------------------------------------------
  if (!trans) {
    const int MP8 = MPAD / 8;
    for (int idx = tid; idx < KP * MP8; idx += 512) {
      const int row = idx / MP8, ch = idx - row * MP8;
      const bool ok = row < Kd && ch * 8 < ldw;
      Pack<E, 8> v = ld_pack<E, 8>(Wg + (ok ? (int64_t)row * ldw + ch * 8 : 0));
#pragma unroll
      for (int j = 0; j < 8; ++j) v.v[j] = (ok && ch * 8 + j < Md) ? v.v[j] : (E)0.f;
      st_pack<E, 8>(sm + dkv_prow(row) * LDW + ch * 8, v);
    }
  } else {
    // W is [m][k] in memory: consecutive lanes take consecutive m (2-byte stores of a wave fall into one LDS row per k)
    const int KP8 = KP / 8;
    for (int idx = tid; idx < MPAD * KP8; idx += 512) {
      const int ch = idx / MPAD, m = idx - ch * MPAD;
      const bool ok = m < Md && ch * 8 < ldw;
      const Pack<E, 8> v = ld_pack<E, 8>(Wg + (ok ? (int64_t)m * ldw + ch * 8 : 0));
#pragma unroll
      for (int j = 0; j < 8; ++j) sm[dkv_prow(ch * 8 + j) * LDW + m] = (ok && ch * 8 + j < Kd) ? v.v[j] : (E)0.f;
    }
  }
  __syncthreads();
  E8 wf[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const int i = two ? (f >> 2) : 0, ks = two ? (f & 3) : f;
    const int n = min(wave + 8 * i, nmt - 1), k0 = min(ks, KS - 1) * 32;  // (clamped: unused fragments read valid LDS)
    wf[f] = dkv_frag<E>(sm, LDW, k0, n * 16, fq, tq, tp);
  }
  __syncthreads();  // the Wt image is dead: the region now holds the X chunks
  // ---- X in 64-column chunks: registers -> LDS (one chunk ahead in registers), out tile = Wt^T X_chunk ------------------------------
  constexpr int NPK = DKV_KM * (DKV_XC / 8) / 512;  // 16-byte packs of a chunk per thread (4)
  Pack<E, 8> xr[NPK];
  auto xload = [&](int c) {
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
      const int idx = tid + i * 512, row = idx >> 3, ch = idx & 7;
      if (row < KP) xr[i] = ld_pack<E, 8>(Xg + (int64_t)min(row, Kd - 1) * ldx + c * DKV_XC + ch * 8);  // rows past Kd: finite, times Wt = 0
    }
  };
  auto xstore = [&]() {
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
      const int idx = tid + i * 512, row = idx >> 3, ch = idx & 7;
      if (row < KP) st_pack<E, 8>(sm + dkv_prow(row) * DKV_LDX + ch * 8, xr[i]);
    }
  };
  xload(0);
  for (int c = 0; c < XE / DKV_XC; ++c) {
    xstore();
    __syncthreads();
    if (c + 1 < XE / DKV_XC) xload(c + 1);
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks < KS) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const E8 af = dkv_frag<E>(sm, DKV_LDX, ks * 32, m * 16, fq, tq, tp);
          acc[0][m] = H16<E>::mfma32(af, wf[ks], acc[0][m]);
          if (ks < 4 && two) acc[1][m] = H16<E>::mfma32(af, wf[4 + (ks & 3)], acc[1][m]);
        }
      }
    }
    // acc[i][m][r] = out[row = n*16 + fr][d = c*64 + m*16 + fq*4 + r]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave + 8 * i) * 16 + fr;
      if ((i == 0 || two) && wave + 8 * i < nmt && row < Md) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          Pack<E, 4> pk;
#pragma unroll
          for (int r = 0; r < 4; ++r) pk.v[r] = (E)acc[i][m][r];
          st_pack<E, 4>(Og + (int64_t)row * ldo + c * DKV_XC + m * 16 + fq * 4, pk);
        }
      }
    }
    __syncthreads();  // everyone is done reading the chunk before it is overwritten
  }
}

unsigned long long* g_x3_stamps = nullptr;  // tests/probes/xattn3_probe.py (d2r_xattn3_debug_stamps): cycle stamps of block 0
int g_x3_dbg = 0;                            // ablation mode of the measurement build (d2r_xattn3_debug_mode)

template <typename E>
void x3_fill(X3Args<E>& a, int ncore, const void* const* q, int64_t ldq, int64_t sqb, const void* const* k, int64_t ldk, int64_t skb,
             const void* const* v, int64_t ldv, int64_t svb, void* const* o, int64_t ldo, int64_t sob, const void* const* residual,
             int64_t ldr, int64_t srb, const float* mask, float* const* lse, int B, int Lq, int Lk, float scale) {
  for (int c = 0; c < ncore; ++c) {
    a.q[c] = (const E*)q[c], a.k[c] = (const E*)k[c], a.v[c] = (const E*)v[c], a.o[c] = (E*)o[c], a.lse[c] = lse[c];
    a.res[c] = residual ? (const E*)residual[c] : nullptr;
  }
  a.mask = mask, a.ncore = ncore;
  a.ldq = ldq, a.sqb = sqb, a.ldk = ldk, a.skb = skb, a.ldv = ldv, a.svb = svb, a.ldo = ldo, a.sob = sob, a.ldr = ldr, a.srb = srb;
  a.B = B, a.Lq = Lq, a.Lk = Lk, a.scale = scale;
  a.ntile = (Lq + QT - 1) / QT;
  a.dbg = g_x3_dbg;
  a.ts = g_x3_stamps;
}

template <typename E>
int x3_fwd_launch(int ncore, const void* const* q, int64_t ldq, int64_t sqb, const void* const* k, int64_t ldk, int64_t skb, const void* const* v,
                  int64_t ldv, int64_t svb, void* const* o, int64_t ldo, int64_t sob, const void* const* residual, int64_t ldr, int64_t srb,
                  const float* mask, float* const* lse, int B, int Lq, int Lk, float scale, hipStream_t st) {
  X3Args<E> a = {};
  x3_fill<E>(a, ncore, q, ldq, sqb, k, ldk, skb, v, ldv, svb, o, ldo, sob, residual, ldr, srb, mask, lse, B, Lq, Lk, scale);
  hipLaunchKernelGGL((xattn3_fwd_kernel<E, 256>), dim3((B + 7) / 8 * 8 * a.ntile * ncore), dim3(256), 0, st, a);
  return 1;
}

template <typename E>
int x3_bwd_launch(int ncore, const void* const* q, int64_t ldq, int64_t sqb, const void* const* k, int64_t ldk, int64_t skb, const void* const* v,
                  int64_t ldv, int64_t svb, const void* const* dO, int64_t ldg, int64_t sgb, const void* const* o, int64_t ldo, int64_t sob,
                  const void* const* residual, int64_t ldr, int64_t srb, const float* mask, const float* const* lse, void* const* P,
                  void* const* dS, int lkp, int B, int Lq, int Lk, float scale, hipStream_t st) {
  X3Args<E> a = {};
  x3_fill<E>(a, ncore, q, ldq, sqb, k, ldk, skb, v, ldv, svb, const_cast<void* const*>(o), ldo, sob, residual, ldr, srb, mask,
             const_cast<float* const*>(lse), B, Lq, Lk, scale);
  for (int c = 0; c < ncore; ++c) a.dO[c] = (const E*)dO[c], a.p_out[c] = (E*)P[c], a.ds_out[c] = (E*)dS[c];
  a.lkp = lkp, a.ldg = ldg, a.sgb = sgb;
  hipLaunchKernelGGL((xattn3_bwd_kernel<E, 256>), dim3((B + 7) / 8 * 8 * a.ntile * ncore), dim3(256), 0, st, a);
  return 1;
}

template <typename E>
int x3_dkv_launch(int ngroup, const void* const* W, const void* const* X, void* const* out, const int64_t* ldx, const int64_t* sxb,
                  const int64_t* ldo, const int64_t* sob, const int* trans, int lkp, int B, int Lq, int Lk, hipStream_t st) {
  DkvArgs<E> a = {};
  for (int g = 0; g < ngroup; ++g) {
    a.W[g] = (const E*)W[g], a.X[g] = (const E*)X[g], a.out[g] = (E*)out[g];
    a.ldx[g] = ldx[g], a.sxb[g] = sxb[g], a.ldo[g] = ldo[g], a.sob[g] = sob[g], a.trans[g] = trans[g];
    a.Kd[g] = trans[g] ? Lk : Lq, a.Md[g] = trans[g] ? Lq : Lk;
  }
  a.swb = (int64_t)Lq * lkp, a.ldw = lkp, a.B = B, a.ngroup = ngroup;
  // the compact kernel (two workgroups per CU) when every group fits its eight fragments per wave and 64 KB of LDS
  bool compact = true;
  size_t lds = 0;
  for (int g = 0; g < ngroup && compact; ++g) {
    const int KP = (a.Kd[g] + 31) / 32 * 32, KS = KP / 32, nmt = (a.Md[g] + 15) / 16;
    compact = nmt <= 8 ? KS <= 8 : (nmt <= 16 && KS <= 4);
    const size_t need = (size_t)KP * (size_t)(dkv_ldw(nmt) > DKV_LDX ? dkv_ldw(nmt) : DKV_LDX) * sizeof(E);
    lds = need > lds ? need : lds;
  }
  // (only when the first kernel's one workgroup per CU would need a second round: with at most 256 workgroups it is the faster one,
  //  54 against 59 us for the three products of one 128 x 128 problem)
  if (compact && lds <= 65536 && (B + 7) / 8 * 8 * ngroup > 256) {
    hipLaunchKernelGGL((xattn3_dkv2_kernel<E>), dim3((B + 7) / 8 * 8 * ngroup), dim3(512), lds, st, a);
    return 1;
  }
  hipLaunchKernelGGL((xattn3_dkv_kernel<E>), dim3((B + 7) / 8 * 8 * ngroup), dim3(512), 0, st, a);
  return 1;
}

}  // namespace

// Host entry used by d2r_xattn_fwd_multi (attention.hip).  Returns 1 when the launch was taken (Lk <= 256), else 0.
// measurement aid of tests/probes/xattn3_probe.py, not declared in include/d2r_hip.h: block 0 of the forward kernel leaves
// s_memtime stamps of its four waves in dst [4][64] (0 start, 1 loop entry, 2+t end of score tile t, 18 softmax done, 19+t end of
// value tile t, 35 ring released, 36 end); nullptr switches it off
extern "C" void d2r_xattn3_debug_stamps(unsigned long long* dst) { g_x3_stamps = dst; }
extern "C" void d2r_xattn3_debug_mode(int mode) { g_x3_dbg = mode; }  // acts on a -DD2R_X3_PROBES=1 build only

int d2r_xattn3_fwd_try(int dtype, int ncore, const void* const* q, int64_t ldq, int64_t sqb, const void* const* k, int64_t ldk, int64_t skb,
                       const void* const* v, int64_t ldv, int64_t svb, void* const* o, int64_t ldo, int64_t sob, const void* const* residual,
                       int64_t ldr, int64_t srb, const float* mask, float* const* lse, int B, int Lq, int Lk, float scale, hipStream_t st) {
  if (Lk > 256 || Lk < 1 || Lq < 1 || ncore < 1 || ncore > X3_MAXCORE) return 0;
  if (dtype == D2R_F16)
    return x3_fwd_launch<f16_t>(ncore, q, ldq, sqb, k, ldk, skb, v, ldv, svb, o, ldo, sob, residual, ldr, srb, mask, lse, B, Lq, Lk, scale, st);
  return x3_fwd_launch<bf16_t>(ncore, q, ldq, sqb, k, ldk, skb, v, ldv, svb, o, ldo, sob, residual, ldr, srb, mask, lse, B, Lq, Lk, scale, st);
}

// Query side of the backward pass (dS and P) for `ncore` problems; o / residual: the forward's outputs and the residuals it added
// (residual may be NULL).  Returns 1 when taken (Lk <= 256).
int d2r_xattn3_bwd_try(int dtype, int ncore, const void* const* q, int64_t ldq, int64_t sqb, const void* const* k, int64_t ldk, int64_t skb,
                       const void* const* v, int64_t ldv, int64_t svb, const void* const* dO, int64_t ldg, int64_t sgb, const void* const* o,
                       int64_t ldo, int64_t sob, const void* const* residual, int64_t ldr, int64_t srb, const float* mask,
                       const float* const* lse, void* const* P, void* const* dS, int lkp, int B, int Lq, int Lk, float scale, hipStream_t st) {
  if (Lk > 256 || Lk < 1 || Lq < 1 || ncore < 1 || ncore > X3_MAXCORE || !o) return 0;
  if (dtype == D2R_F16)
    return x3_bwd_launch<f16_t>(ncore, q, ldq, sqb, k, ldk, skb, v, ldv, svb, dO, ldg, sgb, o, ldo, sob, residual, ldr, srb, mask, lse, P, dS, lkp, B,
                                Lq, Lk, scale, st);
  return x3_bwd_launch<bf16_t>(ncore, q, ldq, sqb, k, ldk, skb, v, ldv, svb, dO, ldg, sgb, o, ldo, sob, residual, ldr, srb, mask, lse, P, dS, lkp, B,
                               Lq, Lk, scale, st);
}

// Product side: for `ngroup` (<= 12) operand triples and every sample in one launch (see the kernel).  W_g: 16-bit [B][Lq][lkp]
// (P or dS).  trans_g = 0: out_g[b] [Lk, 768] = W_g[b]^T X_g[b] with X_g[b] [Lq, 768] (dV, dK); trans_g = 1: out_g[b] [Lq, 768] =
// W_g[b] X_g[b] with X_g[b] [Lk, 768] (dQ).  Returns 1 when taken.
int d2r_xattn3_dkv_try(int dtype, int ngroup, const void* const* W, const void* const* X, void* const* out, const int64_t* ldx, const int64_t* sxb,
                       const int64_t* ldo, const int64_t* sob, const int* trans, int lkp, int B, int Lq, int Lk, hipStream_t st) {
  if (Lq > DKV_KM || Lk > DKV_KM || lkp % 8 != 0 || lkp > DKV_KM || lkp < Lk || ngroup < 1 || ngroup > DKV_MAXG) return 0;
  for (int g = 0; g < ngroup; ++g)
    if (!d2r_aligned16(W[g]) || !d2r_aligned16(X[g]) || (reinterpret_cast<uintptr_t>(out[g]) & 7u) || ldx[g] % 8 != 0 || sxb[g] % 8 != 0 ||
        ldo[g] % 4 != 0 || sob[g] % 4 != 0)
      return 0;
  if (dtype == D2R_F16) return x3_dkv_launch<f16_t>(ngroup, W, X, out, ldx, sxb, ldo, sob, trans, lkp, B, Lq, Lk, st);
  return x3_dkv_launch<bf16_t>(ngroup, W, X, out, ldx, sxb, ldo, sob, trans, lkp, B, Lq, Lk, st);
}
