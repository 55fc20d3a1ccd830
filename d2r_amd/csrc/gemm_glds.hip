// gemm_glds.hip — bf16 MFMA GEMM for the LARGE shapes of the path (fused q|k|v projections, FFN up/down, their
// dX / dW): 128 x BN x 64 tiles, operand tiles streamed global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR
// staging, no ds_write), two LDS buffers, counted vmcnt so the next tile's DMA stays in flight across the barriers.
//
// LDS-DMA writes wave-uniform base + lane*16 B, i.e. the LDS image is linear; bank conflicts are avoided by an XOR
// swizzle applied to the per-lane SOURCE address and, identically, to the read address (cdna_hip_programming.md
// rule 21):
//   k-contiguous operand (A of NT/NN, B of NT): image [rows][64 k] (128 B rows); 16-B chunk c of row r sits at
//       position c ^ (r & 7); fragments are ds_read_b128.
//   k-strided operand (B of NN, A/B of TN): image [64 k][R rows] (R*2 B rows); chunk c of k-row k sits at position
//       c ^ kswz(k) (whole chunk pairs move, see kswz below); fragments are two ds_read_b64_tr_b16 (hardware transpose).
// Rows / columns past the matrix edge are CLAMPED to a valid address instead of zero-filled: they only feed output
// elements that are never stored.  Requirements (checked by the host dispatcher, else the generic kernel runs):
// bf16, batch 1, K % 64 == 0, 16-B aligned operands, and M % 8 == N % 8 == 0 for k-strided operands.
#include "gemm_args.h"

#ifndef D2R_GEMM_PROBES
#define D2R_GEMM_PROBES 0
#endif


// XOR mask (in 16-byte chunks) of k-row k of a k-strided image with CH chunks per row.  One 32-lane group of ds_read_b64_tr_b16
// touches the eight k-rows k0 + {0..3, 8..11} (then + 4), 32 contiguous bytes = one chunk PAIR of each.  Rows of 256 or 512 bytes
// all start on bank 0: the mask moves whole pairs, by a different amount for each of the eight rows (bits 0-1 and bit 3 of k), so
// that the group covers all 64 banks once.  (Rounds 1-2 XORed single chunks with the low bits of k: rows k and k ^ 1 then share
// their chunk pair - a two-way conflict on every transposing read, SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles of the
// weight-gradient kernel, 27-29 % of the NN kernels'.)  Rows of 128 bytes alternate between the two halves of the banks by
// themselves; the mask then only has to separate the four rows of a half (bits 1 and 3 of k) over its four pairs.
template <int CH>
__device__ __forceinline__ int kswz(int k) {
  if constexpr (CH >= 16) return ((k & 3) | ((k >> 1) & 4)) << 1;
  else return (((k >> 1) & 1) | ((k >> 2) & 2)) << 1;
}

template <int CH>
__device__ __forceinline__ int kpos(int c, int k) { return c ^ kswz<CH>(k); }

// NWN waves along N (2: four waves, 4: eight waves per workgroup); a wave owns 64 rows x BN/NWN columns.
// PIPE = 1: software-pipelined K-step (all fragment reads of the step issued up front behind counted lgkmcnt waits, the
// closing barrier in the MIDDLE of the MFMA block, the DMA of tile t+2 issued right behind it: two tiles in flight).
// MODE 1 (TN only): grouped weight-gradient mode — blockIdx.z selects one of up to 32 same-shape problems (operand pointers in
// `grp`), the output is fp32 and ACCUMULATED (C += A^T B, 16-byte loads / stores), and the workgroups of the first tile column
// also produce the bias gradient dbias[m] += sum_k A[k,m] (their A fragments times an all-ones fragment).
// MODE 2 (TN only): grouped AND batched products with a 16-bit output through the ordinary epilogue — blockIdx.z = group * g.gbatch
// + batch index, operands at grp.A/B/C[group] + batch * g.sAb / sBb / sCb; the reduction length need not be a multiple of 64
// (as in mode 1); rows are LOADED up to g.M (a multiple of 8) and STORED up to g.m_store.  Serves the key-side products of the
// cross-attention backward (dV = P^T dO, dK = dS^T Q for every sample of up to four attention problems: one launch).
template <typename E, int LAYOUT, int BN, int NWN, int PIPE, int MODE, int NWM, bool SPLITK = false>
__device__ __forceinline__ void gemm_glds_body(GemmArgs g, const GemmGroup& grp, int tile_m, int tile_n, int z, int split = 0) {
  constexpr bool WGRAD = MODE == 1;        // fp32 accumulate epilogue + bias gradient
  constexpr bool RAGGED = MODE != 0;       // grouped launch (3-D tile order), ragged reduction length
  typedef typename H16<E>::v8 h8;  // E: E or f16_t (same tiles and LDS images; the MFMA opcode differs)
  typedef typename H16<E>::v4 h4;
  constexpr int BM = 64 * NWM, BK = 64, NW = NWM * NWN;  // NWM waves along M (a wave owns 64 rows): 128-row tiles, or 256 (weight gradients)
  constexpr bool A_KCONT = (LAYOUT != D2R_GEMM_TN);
  constexpr bool B_KCONT = (LAYOUT == D2R_GEMM_NT);
  constexpr int WM = BM / NWM, WN = BN / NWN, TM = WM / 16, TN = WN / 16;
  constexpr int ACH = BM / 8, AROWS = 64 / ACH;  // k-strided A image [64 k][BM m]: 16-byte chunks per k-row, k-rows per DMA instruction
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, BUF = A_BYTES + B_BYTES;
  // LDS-DMA instructions (1 KiB each) per tile and per wave
  constexpr int IA = A_BYTES / 1024 / NW, IB = B_BYTES / 1024 / NW;
  static_assert(IA * NW * 1024 == A_BYTES && IB * NW * 1024 == B_BYTES && IA >= 1 && IB >= 1, "tile does not split over the waves");
  constexpr int SZ_EPI = NW * WM * (WN + 8) * 2;
  constexpr int SZ_ALL = 2 * BUF > SZ_EPI ? 2 * BUF : SZ_EPI;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SZ_ALL];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / NWN) * WM, wn0 = (wave % NWN) * WN;
  // the probes' hooks (cycle stamps of workgroup (0,0), ablation switches g.dbg: tests/probes/gemm_stamps.py, gemm_ablation.py) exist only
  // in a library built with -DD2R_GEMM_PROBES=1 (D2R_GEMM_PROBES=1 python -m d2r_amd.build): as run-time branches they cut the K-loop
  // into a dozen basic blocks per step, which kept the compiler from moving the second half-step's fragment reads over the first
  // half-step's MFMAs
  constexpr bool PROBES = D2R_GEMM_PROBES != 0;
  const int dbg = PROBES ? g.dbg : 0;
  const bool stamping = PROBES && g.ts != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
#define GEMM_STAMP(i) do { if (stamping) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g.ts[wave * 8 + (i)] = t_; } } while (0)
  GEMM_STAMP(0);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const E* A = reinterpret_cast<const E*>(g.A);
  const E* B = reinterpret_cast<const E*>(g.B);
  if constexpr (WGRAD) {
    static_assert(LAYOUT == D2R_GEMM_TN, "the weight-gradient mode is a TN product");
    A = reinterpret_cast<const E*>(grp.A[z]);
    B = reinterpret_cast<const E*>(grp.B[z]);
    g.C = grp.C[z];
    g.dbias = grp.dbias[z];
  }
  if constexpr (MODE == 2) {
    static_assert(LAYOUT == D2R_GEMM_TN, "the batched mode is a TN product");
    const int zg = z / g.gbatch, zb = z - zg * g.gbatch;
    A = reinterpret_cast<const E*>(grp.A[zg]) + zb * g.sAb;
    B = reinterpret_cast<const E*>(grp.B[zg]) + zb * g.sBb;
    g.C = reinterpret_cast<E*>(grp.C[zg]) + zb * g.sCb;
  }
  const bool do_bias = WGRAD && g.dbias != nullptr && tile_n == 0 && (wave % NWN) == 0;
  f32x4 acc_b[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) acc_b[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane source offsets (elements) of this wave's DMA instructions, without the k0 term
  int64_t offA[IA], offB[IB];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int ins = wave + NW * i;
    if constexpr (A_KCONT) {  // 8 rows x 128 B per instruction
      const int row = ins * 8 + (lane >> 3), c = (lane & 7) ^ (row & 7);
      const int grow = min(m0 + row, g.M - 1);
      offA[i] = (int64_t)grow * g.lda + c * 8;
    } else {  // [64 k][BM m]: AROWS k-rows x BM*2 bytes per instruction (4 x 256 B for the 128-row tile, 2 x 512 B for 256 rows)
      const int krow = ins * AROWS + lane / ACH, c = (lane % ACH) ^ kswz<ACH>(krow);
      const int col = min(m0 + c * 8, g.M - 8);
      offA[i] = (int64_t)krow * g.lda + col;
    }
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int ins = wave + NW * i;
    if constexpr (B_KCONT) {
      const int row = ins * 8 + (lane >> 3), c = (lane & 7) ^ (row & 7);
      const int grow = min(n0 + row, g.N - 1);
      offB[i] = (int64_t)grow * g.ldb + c * 8;
    } else if constexpr (BN == 128) {
      const int krow = ins * 4 + (lane >> 4), c = (lane & 15) ^ kswz<16>(krow);
      const int col = min(n0 + c * 8, g.N - 8);
      offB[i] = (int64_t)krow * g.ldb + col;
    } else {  // [64 k][64 n]: 8 k-rows x 128 B per instruction
      const int krow = ins * 8 + (lane >> 3), c = (lane & 7) ^ kswz<8>(krow);
      const int col = min(n0 + c * 8, g.N - 8);
      offB[i] = (int64_t)krow * g.ldb + col;
    }
  }

  // weight-gradient mode: the reduction length (token rows) need not be a multiple of 64 — the k-rows of the last tile past
  // the end are CLAMPED to the last valid row here and the A rows are zeroed in LDS before they are read (see the K-loop)
  const int nk_w = (g.K + BK - 1) / BK, rem_w = g.K - (nk_w - 1) * BK;
  auto issue = [&](int t, int buf) {
    if (dbg == 2) return;
    const int k0 = t * BK;
    unsigned char* base = smem + buf * BUF;
    int64_t adjA = 0, adjB = 0;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      if constexpr (RAGGED) {
        const int krow = (wave + NW * i) * AROWS + lane / ACH;
        adjA = (t == nk_w - 1 && krow >= rem_w) ? (int64_t)(rem_w - 1 - krow) * g.lda : 0;
      }
      const E* src = A + offA[i] + adjA + (A_KCONT ? (int64_t)k0 : (int64_t)k0 * g.lda);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + (wave + NW * i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      if constexpr (RAGGED) {
        const int krow = (wave + NW * i) * 4 + (lane >> 4);
        adjB = (t == nk_w - 1 && krow >= rem_w) ? (int64_t)(rem_w - 1 - krow) * g.ldb : 0;
      }
      const E* src = B + offB[i] + adjB + (B_KCONT ? (int64_t)k0 : (int64_t)k0 * g.ldb);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave + NW * i) * 1024), 16, 0, 0);
    }
  };

  // one DMA instruction of tile t (idx < IA: A piece idx, else B piece idx - IA); called with compile-time indices from unrolled loops
  auto issue_one = [&](int t, int buf, int idx) {
    const int k0 = t * BK;
    unsigned char* base = smem + buf * BUF;
    if (idx < IA) {
      int64_t adjA = 0;
      if constexpr (RAGGED) {
        const int krow = (wave + NW * idx) * AROWS + lane / ACH;
        adjA = (t == nk_w - 1 && krow >= rem_w) ? (int64_t)(rem_w - 1 - krow) * g.lda : 0;
      }
      const E* src = A + offA[idx] + adjA + (A_KCONT ? (int64_t)k0 : (int64_t)k0 * g.lda);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + (wave + NW * idx) * 1024), 16, 0, 0);
    } else {
      const int ib = idx - IA;
      int64_t adjB = 0;
      if constexpr (RAGGED) {
        const int krow = (wave + NW * ib) * 4 + (lane >> 4);
        adjB = (t == nk_w - 1 && krow >= rem_w) ? (int64_t)(rem_w - 1 - krow) * g.ldb : 0;
      }
      const E* src = B + offB[ib] + adjB + (B_KCONT ? (int64_t)k0 : (int64_t)k0 * g.ldb);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave + NW * ib) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  int nk = RAGGED ? nk_w : g.K / BK;
  if constexpr (SPLITK) {  // this workgroup reduces K-tiles [kb, ke) of the product: the operand pointers move, the loop shortens
    static_assert(MODE == 0 && PIPE == 0, "split-K is built for the plain forward / dX loop");
    const int kb = nk * split / g.splits, ke = nk * (split + 1) / g.splits;
    A += A_KCONT ? (int64_t)kb * BK : (int64_t)kb * BK * g.lda;
    B += B_KCONT ? (int64_t)kb * BK : (int64_t)kb * BK * g.ldb;
    nk = ke - kb;
  }
  // fragments of K-substep kk (32 k) from the staged tile at bA / bB
  auto read_frags = [&](const unsigned char* bA, const unsigned char* bB, int kk, h8 (&af)[TM], h8 (&bfr)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (A_KCONT) {
        const int row = wm0 + i * 16 + fr, c = kk * 4 + fq;
        af[i] = *reinterpret_cast<const h8*>(bA + row * 128 + ((c ^ (row & 7)) << 4));
      } else {
        const int k = kk * 32 + fq * 8 + tq, col = wm0 + i * 16 + tp * 4;
        const int c = col >> 3, h = (col & 7) >> 2;
        const h4 lo = lds_tr_read<h4>(bA + k * (BM * 2) + ((c ^ kswz<ACH>(k)) << 4) + h * 8);
        const h4 hi = lds_tr_read<h4>(bA + (k + 4) * (BM * 2) + ((c ^ kswz<ACH>(k + 4)) << 4) + h * 8);
        af[i] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (B_KCONT) {
        const int row = wn0 + j * 16 + fr, c = kk * 4 + fq;
        bfr[j] = *reinterpret_cast<const h8*>(bB + row * 128 + ((c ^ (row & 7)) << 4));
      } else {
        constexpr int ROWB = BN * 2;
        const int k = kk * 32 + fq * 8 + tq, col = wn0 + j * 16 + tp * 4;
        const int c = col >> 3, h = (col & 7) >> 2;
        const h4 lo = lds_tr_read<h4>(bB + k * ROWB + (kpos<BN / 8>(c, k) << 4) + h * 8);
        const h4 hi = lds_tr_read<h4>(bB + (k + 4) * ROWB + (kpos<BN / 8>(c, k + 4) << 4) + h * 8);
        bfr[j] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
  };
  if constexpr (PIPE == 1) {
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    for (int t = 0; t < nk; ++t) {
      const int cur = t & 1;
      // tile t was issued at least one whole K-step ago; the newest tile (t+1) may stay in flight
      if (t + 1 < nk) {
        if constexpr (IA + IB == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (IA + IB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (IA + IB == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if constexpr (IA + IB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_barrier" ::: "memory");  // tile t is visible to every wave
      const unsigned char* bA = smem + cur * BUF;
      const unsigned char* bB = bA + A_BYTES;
      if constexpr (RAGGED) {
        if (t == nk - 1 && rem_w < BK) {  // rows past the reduction length: zero A (B holds clamped, finite rows) -> no contribution
          unsigned char* z0 = smem + cur * BUF + rem_w * (BM * 2);
          const int zbytes = (BK - rem_w) * (BM * 2);
          for (int off = tid * 16; off < zbytes; off += NW * 64 * 16) *reinterpret_cast<uint4*>(z0 + off) = uint4{0u, 0u, 0u, 0u};
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
      }
      h8 af0[TM], bf0[TN], af1[TM], bf1[TN];
      read_frags(bA, bB, 0, af0, bf0);
      read_frags(bA, bB, 1, af1, bf1);
      if constexpr (!A_KCONT || !B_KCONT) lds_reads_done();  // (asm transposed reads are not tracked by the compiler)
      __builtin_amdgcn_sched_barrier(0);  // every LDS read of the step is issued before its first MFMA
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = H16<E>::mfma32(bf0[j], af0[i], acc[i][j]);
      if constexpr (WGRAD) {
        if (do_bias) {
          const E one = (E)1.f;
          const h8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            acc_b[i] = H16<E>::mfma32(ones, af0[i], acc_b[i]);
            acc_b[i] = H16<E>::mfma32(ones, af1[i], acc_b[i]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // every wave has its fragments in registers: buffer `cur` is free -> refill it with tile t+2 while the second half runs
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      {
        // the DMA instructions of tile t+2 go out BETWEEN the MFMAs of the second half-step (one per STRIDE MFMAs): issued in one
        // run they queue up behind the texture addresser (1 KiB = 16 cycles each, eight waves at once) and the wave's MFMAs wait
        constexpr int NM = TM * TN, ND = IA + IB, STRIDE = NM / ND > 0 ? NM / ND : 1;
        const bool more = t + 2 < nk;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          const int i = m / TN, j = m % TN;
          acc[i][j] = H16<E>::mfma32(bf1[j], af1[i], acc[i][j]);
          if (m % STRIDE == STRIDE - 1 && m / STRIDE < ND) {
            __builtin_amdgcn_sched_barrier(0);
            if (more) issue_one(t + 2, cur, m / STRIDE);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if constexpr (ND > NM / STRIDE) {
#pragma unroll
          for (int r = NM / STRIDE; r < ND; ++r)
            if (more) issue_one(t + 2, cur, r);
        }
      }
    }
  } else {
  issue(0, 0);
  unsigned long long ph_issue = 0, ph_wait = 0, ph_math = 0, ph_close = 0, ph_t = 0;  // (stamping only) cycles per phase, summed over the K loop
#define GEMM_PHASE(acc_) do { if (stamping) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - ph_t; ph_t = n_; } } while (0)
  if (stamping) ph_t = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    if (t + 1 < nk) {
      issue(t + 1, cur ^ 1);  // buffer cur^1 was last read in iteration t-1; every wave has passed its closing barrier
      GEMM_PHASE(ph_issue);
      if constexpr (IA + IB == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (IA + IB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr (IA + IB == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if constexpr (IA + IB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");  // (asm + memory clobber: the compiler may not move LDS reads / DMA issues across it)
    GEMM_PHASE(ph_wait);
    if (t == 0) GEMM_STAMP(1);
    // tile t has landed for every wave (own counted vmcnt + barrier)
    const unsigned char* bA = smem + cur * BUF;
    const unsigned char* bB = bA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      h8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (A_KCONT) {
          const int row = wm0 + i * 16 + fr, c = kk * 4 + fq;
          af[i] = *reinterpret_cast<const h8*>(bA + row * 128 + ((c ^ (row & 7)) << 4));
        } else {
          const int k = kk * 32 + fq * 8 + tq, col = wm0 + i * 16 + tp * 4;
          const int c = col >> 3, h = (col & 7) >> 2;
          const h4 lo = lds_tr_read<h4>(bA + k * (BM * 2) + ((c ^ kswz<ACH>(k)) << 4) + h * 8);
          const h4 hi = lds_tr_read<h4>(bA + (k + 4) * (BM * 2) + ((c ^ kswz<ACH>(k + 4)) << 4) + h * 8);
          af[i] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (B_KCONT) {
          const int row = wn0 + j * 16 + fr, c = kk * 4 + fq;
          bfr[j] = *reinterpret_cast<const h8*>(bB + row * 128 + ((c ^ (row & 7)) << 4));
        } else {
          constexpr int ROWB = BN * 2;
          const int k = kk * 32 + fq * 8 + tq, col = wn0 + j * 16 + tp * 4;
          const int c = col >> 3, h = (col & 7) >> 2;
          const h4 lo = lds_tr_read<h4>(bB + k * ROWB + (kpos<BN / 8>(c, k) << 4) + h * 8);
          const h4 hi = lds_tr_read<h4>(bB + (k + 4) * ROWB + (kpos<BN / 8>(c, k + 4) << 4) + h * 8);
          bfr[j] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
      if constexpr (!A_KCONT || !B_KCONT) lds_reads_done();  // (asm transposed reads are not tracked by the compiler)
      if (dbg == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bfr[j]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // operands swapped: the accumulator tile is C^T, a lane owns ONE row and 4 consecutive columns
          acc[i][j] = H16<E>::mfma32(bfr[j], af[i], acc[i][j]);
    }
    GEMM_PHASE(ph_math);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone is done reading buffer `cur` before it is refilled at t+1
    GEMM_PHASE(ph_close);
  }
  if (stamping && lane == 0) g.ts[wave * 8 + 4] = ph_issue, g.ts[wave * 8 + 5] = ph_wait, g.ts[wave * 8 + 6] = ph_math, g.ts[wave * 8 + 7] = ph_close;
#undef GEMM_PHASE

  }
  GEMM_STAMP(2);

  if constexpr (SPLITK) {
    // The `splits` workgroups of an output tile (grid z = split index, dispatched in order: every producer is resident or done before its
    // finisher starts) meet here.  Producers 0 .. splits-2 park their accumulators in the workspace - accumulator layout: lane-contiguous
    // 16-byte packs, the finisher's lanes hold the same elements - release them and count themselves in; the finisher (the last split)
    // waits for the count, adds the parked partials in split order (fixed summation order: deterministic) and runs the epilogue.  The
    // counter is back at zero when the finisher is done: launches on one stream reuse the workspace without a clear in between.
    // Cache traffic: the workgroups of a tile sit on different XCDs, whose L2s are not coherent with each other.  The partials therefore
    // travel as agent-scope RELAXED atomic accesses (64-bit; sc1: write-through on the store side, no stale hit on the load side), and
    // the hand-over is "stores complete (vmcnt 0) -> barrier -> relaxed counter increment" / "relaxed counter poll -> barrier -> loads": no
    // release / acquire fence anywhere - on gfx950 those are whole-L2 write-backs and invalidates, and an acquire inside the poll loop
    // invalidated the XCD's L2 under the producers' feet (first version: 35 -> 90 us for the 4096 x 768 x 3072 product).
    const int tile_id = tile_m * gridDim.x + tile_n;
    constexpr int PER = TM * TN * NW * 64 * 2;  // 64-bit words per parked tile
    typedef unsigned long long u64;
    u64* park = reinterpret_cast<u64*>(g.ws) + (int64_t)tile_id * (g.splits - 1) * PER + tid;
    unsigned* cnt = g.kflags + tile_id;
    if (split + 1 < g.splits) {
      u64* dst = park + (int64_t)split * PER;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const u64 lo = (u64)__float_as_uint(acc[i][j][0]) | ((u64)__float_as_uint(acc[i][j][1]) << 32);
          const u64 hi = (u64)__float_as_uint(acc[i][j][2]) | ((u64)__float_as_uint(acc[i][j][3]) << 32);
          __hip_atomic_store(dst + ((i * TN + j) * 2 + 0) * (NW * 64), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(dst + ((i * TN + j) * 2 + 1) * (NW * 64), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid == 0) {
      // bounded wait (a producer is never behind its finisher in dispatch order; the bound only keeps a broken launch from hanging the GPU)
      for (int spin = 0; spin < (1 << 22); ++spin) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(g.splits - 1)) break;
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
    for (int sp = 0; sp + 1 < g.splits; ++sp) {
      const u64* src = park + (int64_t)sp * PER;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const u64 lo = __hip_atomic_load(src + ((i * TN + j) * 2 + 0) * (NW * 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const u64 hi = __hip_atomic_load(src + ((i * TN + j) * 2 + 1) * (NW * 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc[i][j] = f32x4{acc[i][j][0] + __uint_as_float((unsigned)lo), acc[i][j][1] + __uint_as_float((unsigned)(lo >> 32)),
                            acc[i][j][2] + __uint_as_float((unsigned)hi), acc[i][j][3] + __uint_as_float((unsigned)(hi >> 32))};
        }
    }
    __syncthreads();  // (every wave has its partials before the counter is handed back)
    if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  if constexpr (WGRAD) {
    // acc[i][j][r] = C[wm0 + i*16 + fr][wn0 + j*16 + fq*4 + r]: one 16-byte fp32 pack per lane and tile
    float* Cg = reinterpret_cast<float*>(g.C);
    const bool vec = (g.ldc & 3) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = m0 + wm0 + i * 16 + fr;
      if (row >= g.M) continue;
      if (do_bias && fq == 0) g.dbias[row] += acc_b[i][0];  // every column of acc_b holds sum_k A[k,row]; this lane is the row's only writer
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 16 + fq * 4;
        float* p = Cg + (int64_t)row * g.ldc + col;
        if (vec && col + 4 <= g.N) {
          f32x4 old = *reinterpret_cast<const f32x4*>(p);
          *reinterpret_cast<f32x4*>(p) = f32x4{g.beta * old[0] + acc[i][j][0], g.beta * old[1] + acc[i][j][1],
                                               g.beta * old[2] + acc[i][j][2], g.beta * old[3] + acc[i][j][3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (col + r < g.N) p[r] = g.beta * p[r] + acc[i][j][r];
        }
      }
    }
    return;
  }
  if (dbg == 3) {  // (every accumulator kept live: the MFMAs must not be eliminated with the epilogue)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j][0]), "v"(acc[i][j][1]), "v"(acc[i][j][2]), "v"(acc[i][j][3]));
    return;
  }
  // ---- epilogue (same semantics as the generic kernel) ------------------------------------------------
  if constexpr (MODE == 2) g.M = g.m_store;  // (operand loads are done: from here on M only guards the stores)
  if (g.c_dtype == H16<E>::DT && g.vecC) {
    constexpr int LDE = WN + 8;
    E* Cs = reinterpret_cast<E*>(smem) + wave * WM * LDE;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        // acc[i][j][r] = C[wm0 + i*16 + fr][wn0 + j*16 + fq*4 + r]: four consecutive columns -> one 8-byte LDS store
        const int col = n0 + wn0 + j * 16 + fq * 4;
        Pack<E, 4> pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float bv = (g.bias && dbg != 4 && col + r < g.N) ? g.bias[col + r] : 0.f;
          pk.v[r] = (E)(g.alpha * acc[i][j][r] + bv);
        }
        st_pack<E, 4>(Cs + (i * 16 + fr) * LDE + j * 16 + fq * 4, pk);
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    constexpr int CPR = WN / 8;
    E* Cg = reinterpret_cast<E*>(g.C);
    E* Pg = reinterpret_cast<E*>(g.P);
    const E* Rg = reinterpret_cast<const E*>(g.R);
    const E* Gg = reinterpret_cast<const E*>(g.G);
#pragma unroll 1
    for (int it = 0; it < WM * CPR / 64; ++it) {  // (rolled on purpose: one copy of the epilogue arithmetic)
      const int e = it * 64 + lane;
      const int rl = e / CPR, ch = e % CPR;
      const int row = m0 + wm0 + rl, col = n0 + wn0 + ch * 8;
      if (row >= g.M || col >= g.N) continue;
      const Pack<E, 8> pv = ld_pack<E, 8>(Cs + rl * LDE + ch * 8);
      const int64_t ci = (int64_t)row * g.ldc + col;
      const int64_t ri = (int64_t)row * g.ldr + col;
      if (dbg == 5) continue;
      epilogue_pack8(g, pv, Cg, Pg, Rg, Gg, ci, ri, g.N - col);
    }
    if (stamping) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      GEMM_STAMP(3);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = m0 + wm0 + i * 16 + fr;
    if (row >= g.M) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = n0 + wn0 + j * 16 + fq * 4 + r;
        if (col >= g.N) continue;
        float v = g.alpha * acc[i][j][r] + (g.bias ? g.bias[col] : 0.f);
        const int64_t ci = (int64_t)row * g.ldc + col;
        if (g.P) store_c(g.P, g.c_dtype, ci, v);
        v = act_apply_cold(g.act, v);
        if (g.R) v += load_c(g.R, g.c_dtype, (int64_t)row * g.ldr + col);
        if (g.beta != 0.f) v += g.beta * load_c(g.C, g.c_dtype, ci);
        store_c(g.C, g.c_dtype, ci, v);
      }
    }
  }
}

template <typename E, int LAYOUT, int BN, int NWN = 2, int PIPE = 0, int MODE = 0, int NWM = 2>
__global__ __launch_bounds__(NWM * NWN * 64) void gemm_glds_kernel(GemmArgs g, GemmGroup grp) {
  int tile_m, tile_n, z = 0;
  if constexpr (MODE != 0) xcd_tile_3d(g.xcd, tile_m, tile_n, z);
  else xcd_tile(g.xcd, tile_m, tile_n, g.band);
  gemm_glds_body<E, LAYOUT, BN, NWN, PIPE, MODE, NWM>(g, grp, tile_m, tile_n, z);
}

// ---- in-kernel split-K of the 128 x 128 eight-wave kernel (NT / NN, forward / dX products) ---------------------------------------------
// The N = 768 products of the path with a deep reduction (FFN down-projection and its mirror in the backward pass, the q|k|v dX, the
// d_other product of a routing module: K = 2304 ... 13,824) have 192 or 300 output tiles of 36-216 K-tiles each: three quarters of a
// round of workgroups, or one and a sixth.  Splitting K over 2-4 workgroups per tile fills the CUs (two workgroups per CU interleave
// their latencies); the partial sums meet inside the launch (gemm_glds_body, SPLITK) instead of in a reduce launch.  Measured alone on
// the GPU (tests/probes/splitk_probe.py, profiles/splitk_probe_r04.log): the meeting costs about 10 us per launch, so K = 13,824
// gains 17-23 % (134 -> 107 us, 190 -> 157 us) while K = 2304 / 3072 end within -10 ... +14 % of the unsplit launch: the plan below
// takes products with K >= 6144 only (the two d_other products of a step).
template <typename E, int LAYOUT>
__global__ __launch_bounds__(512) void gemm_glds_splitk_kernel(GemmArgs g) {
  // XCD-aware tile order inside a plane of the grid (z = split index; the planes are dispatched one after the other)
  const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy;
  int id = blockIdx.y * gx + blockIdx.x;
  if (g.xcd && nwg >= 16) {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int tile_m = id / gx, tile_n = id - tile_m * gx;
  static const GemmGroup no_group = {};
  gemm_glds_body<E, LAYOUT, 128, 4, 0, 0, 2, true>(g, no_group, tile_m, tile_n, 0, blockIdx.z);
}

// Launches the split-K variant; the caller (gemm.hip) has checked the shape and carved the workspace (d2r_gemm_glds_splitk_plan).
int d2r_gemm_glds_splitk_launch(const GemmArgs& a, int layout, hipStream_t st) {
  dim3 grid(d2r_cdiv(a.N, 128), d2r_cdiv(a.M, 128), a.splits);
  const bool f16 = a.dtype == D2R_F16;
  if (layout == D2R_GEMM_NT) {
    if (f16) hipLaunchKernelGGL((gemm_glds_splitk_kernel<f16_t, D2R_GEMM_NT>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((gemm_glds_splitk_kernel<bf16_t, D2R_GEMM_NT>), grid, dim3(512), 0, st, a);
  } else if (layout == D2R_GEMM_NN) {
    if (f16) hipLaunchKernelGGL((gemm_glds_splitk_kernel<f16_t, D2R_GEMM_NN>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((gemm_glds_splitk_kernel<bf16_t, D2R_GEMM_NN>), grid, dim3(512), 0, st, a);
  } else {
    return d2r_fail(D2R_ERR_INVALID, "d2r_gemm(split-K): layout");
  }
  return d2r_check_launch("d2r_gemm(glds split-K)");
}

// How many workgroups per output tile (1 = no split), from a model of the launch: rounds of up to 512 resident workgroups (two per CU),
// a round with more than 256 runs its K-tiles 1.37 x slower per workgroup (two workgroups share a CU: profiles/tile_balance_r03.log).
int d2r_gemm_glds_splitk_plan(const GemmArgs& a, int layout, int batch, size_t ws_bytes, size_t* slab_bytes) {
  *slab_bytes = 0;
  if (layout == D2R_GEMM_TN || batch != 1 || !d2r_is16(a.dtype) || a.c_dtype != a.dtype || !a.vecA || !a.vecB || !a.vecC) return 1;
  if (a.K % 64 != 0 || a.K < 6144 || a.M < 128 || a.N < 128 || a.dbias) return 1;  // (measured: below K = 6144 the meeting costs what the split gains)
  if (layout != D2R_GEMM_NT && a.N % 8 != 0) return 1;
  const int64_t tiles = (int64_t)d2r_cdiv(a.M, 128) * d2r_cdiv(a.N, 128);
  if (tiles > 448 || tiles > 1024 || ws_bytes < 8192) return 1;
  const int nk = a.K / 64;
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= 4; ++s) {
    if (nk / s < 8) break;
    if (s > 1 && (size_t)tiles * (s - 1) * (128 * 128 * 4) > ws_bytes - 4096) break;
    int64_t left = tiles * s;
    double cost = 0.0;
    while (left > 0) {
      const int64_t now = left > 512 ? 512 : left;
      cost += (double)nk / s * (now > 256 ? 1.37 : 1.0);
      left -= now;
    }
    if (s > 1) cost += 30.0 + 8.0 * (s - 2);  // the meeting, in K-tile times (0.3 us): park 64 KB, count in, poll, fetch - about 10 us per launch, measured
    if (cost < best_cost - 1e-9) best_cost = cost, best = s;
  }
  if (best > 1) *slab_bytes = (size_t)tiles * (best - 1) * (128 * 128 * 4);
  return best;
}

// ---- grouped forward / dX launch: up to 16 INDEPENDENT problems of one layout and type on the 128 x 128 eight-wave tiles ------------
// The cells of a routing layer (models/Cells.py:30-255) run chains of 768 x 768 products over 4-6 thousand rows: 192-300 tiles per
// launch, one partial round of workgroups, 13-15 us for work worth 5.  Products that do not depend on each other (the query projections
// of the three alignment cells and IMRC's q|k|v; the cells' second linears; ...) leave as ONE launch of a thousand tiles.  Every tile is
// computed exactly as in a launch of its own (same tile shape, same accumulation order): bit-identical results.
struct GemmFwdProb {
  const void *A, *B;
  void* C;
  const float* bias;
  const void* R;
  void* P;
  const void* G;
  int M, N, K, lda, ldb, ldc, ldr;
  float alpha, beta;
  int act, gact, tn, pad;
};
constexpr int D2R_GEMM_FWD_GROUP_MAX = 16;
struct GemmFwdGroup {
  int nprob, ntiles, xcd, pad;
  int tile_end[D2R_GEMM_FWD_GROUP_MAX];
  GemmFwdProb p[D2R_GEMM_FWD_GROUP_MAX];
};

template <typename E, int LAYOUT>
__global__ __launch_bounds__(512) void gemm_glds_group_kernel(GemmArgs g, GemmFwdGroup fg) {
  // linear workgroup id -> linear tile id: every XCD (ids congruent mod 8) walks a CONTIGUOUS run of tiles, i.e. whole problems or
  // whole row panels of one
  int L = blockIdx.x;
  const int nwg = fg.ntiles;
  if (fg.xcd && nwg >= 16) {
    const int q = nwg >> 3, r = nwg & 7, xcd = L & 7, k = L >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  int lo = 0, hi = fg.nprob - 1;  // first problem whose tile_end exceeds L
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (L < fg.tile_end[mid]) hi = mid;
    else lo = mid + 1;
  }
  const int zp = __builtin_amdgcn_readfirstlane(lo);
  const GemmFwdProb& P = fg.p[zp];
  const int rem = L - (zp ? fg.tile_end[zp - 1] : 0);
  const int tile_m = rem / P.tn, tile_n = rem - tile_m * P.tn;
  g.A = P.A, g.B = P.B, g.C = P.C, g.bias = P.bias, g.R = P.R, g.P = P.P, g.G = P.G;
  g.M = P.M, g.N = P.N, g.K = P.K, g.lda = P.lda, g.ldb = P.ldb, g.ldc = P.ldc, g.ldr = P.ldr;
  g.alpha = P.alpha, g.beta = P.beta, g.act = P.act, g.gact = P.gact;
  gemm_glds_body<E, LAYOUT, 128, 4, 0, 0, 2>(g, GemmGroup{}, tile_m, tile_n, 0);
}

// eligibility of one problem for the grouped launch (the conditions of d2r_gemm_glds_try for the 128 x 128 eight-wave tile, plus a
// 16-bit vectorised output: the shared pack epilogue)
int d2r_gemm_glds_group_ok(const GemmArgs& a, int layout, int batch) {
  if (!d2r_is16(a.dtype) || a.c_dtype != a.dtype || batch != 1 || layout == D2R_GEMM_TN) return 0;
  if (a.K % 64 != 0 || a.K < 128 || a.M < 128 || a.N < 128 || !a.vecA || !a.vecB || !a.vecC || a.dbias) return 0;
  if (layout == D2R_GEMM_NN && a.N % 8 != 0) return 0;
  if (a.lda >= ((int64_t)1 << 31) || a.ldb >= ((int64_t)1 << 31) || a.ldc >= ((int64_t)1 << 31) || a.ldr >= ((int64_t)1 << 31)) return 0;
  return 1;
}

int d2r_gemm_glds_group_launch(const GemmArgs* probs, int n, int layout, hipStream_t st) {
  if (n < 1 || n > D2R_GEMM_FWD_GROUP_MAX) return d2r_fail(D2R_ERR_INVALID, "d2r_gemm_glds_group_launch: %d problems (1..%d)", n, D2R_GEMM_FWD_GROUP_MAX);
  GemmFwdGroup fg = {};
  fg.nprob = n, fg.xcd = 1;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    const GemmArgs& a = probs[i];
    GemmFwdProb& p = fg.p[i];
    p.A = a.A, p.B = a.B, p.C = a.C, p.bias = a.bias, p.R = a.R, p.P = a.P, p.G = a.G;
    p.M = a.M, p.N = a.N, p.K = a.K, p.lda = (int)a.lda, p.ldb = (int)a.ldb, p.ldc = (int)a.ldc, p.ldr = (int)a.ldr;
    p.alpha = a.alpha, p.beta = a.beta, p.act = a.act, p.gact = a.gact, p.tn = d2r_cdiv(a.N, 128), p.pad = 0;
    tiles += d2r_cdiv(a.M, 128) * p.tn;
    fg.tile_end[i] = tiles;
  }
  for (int i = n; i < D2R_GEMM_FWD_GROUP_MAX; ++i) fg.tile_end[i] = tiles;
  fg.ntiles = tiles;
  GemmArgs base = probs[0];
  base.ts = nullptr, base.dbg = 0, base.band = 0, base.vecC = 1;
  const bool f16 = base.dtype == D2R_F16;
  d2r_gemm_variant_tl = 3;
  if (layout == D2R_GEMM_NT) {
    if (f16) hipLaunchKernelGGL((gemm_glds_group_kernel<f16_t, D2R_GEMM_NT>), dim3(tiles), dim3(512), 0, st, base, fg);
    else hipLaunchKernelGGL((gemm_glds_group_kernel<bf16_t, D2R_GEMM_NT>), dim3(tiles), dim3(512), 0, st, base, fg);
  } else {
    if (f16) hipLaunchKernelGGL((gemm_glds_group_kernel<f16_t, D2R_GEMM_NN>), dim3(tiles), dim3(512), 0, st, base, fg);
    else hipLaunchKernelGGL((gemm_glds_group_kernel<bf16_t, D2R_GEMM_NN>), dim3(tiles), dim3(512), 0, st, base, fg);
  }
  return d2r_check_launch("d2r_gemm_group(glds)");
}

template <typename E, int LAYOUT>
static void launch_glds(const GemmArgs& a, int bn, hipStream_t st) {
  static const GemmGroup no_group = {};
  const bool pipe = bn >= 1000;  // 1064 / 1128 / 1129: the software-pipelined K-loop
  if (pipe) bn -= 1000;
  if (bn == 129) {  // 128 x 128 tile on EIGHT waves (2 x 4): per wave as the 128x64 kernel, a third less L2 traffic per flop
    dim3 grid(d2r_cdiv(a.N, 128), d2r_cdiv(a.M, 128));
    if (pipe) hipLaunchKernelGGL((gemm_glds_kernel<E, LAYOUT, 128, 4, 1>), grid, dim3(512), 0, st, a, no_group);
    else hipLaunchKernelGGL((gemm_glds_kernel<E, LAYOUT, 128, 4>), grid, dim3(512), 0, st, a, no_group);
    return;
  }
  if (bn == 128) {
    dim3 grid(d2r_cdiv(a.N, 128), d2r_cdiv(a.M, 128));
    if (pipe) hipLaunchKernelGGL((gemm_glds_kernel<E, LAYOUT, 128, 2, 1>), grid, dim3(256), 0, st, a, no_group);
    else hipLaunchKernelGGL((gemm_glds_kernel<E, LAYOUT, 128>), grid, dim3(256), 0, st, a, no_group);
  } else {
    dim3 grid(d2r_cdiv(a.N, 64), d2r_cdiv(a.M, 128));
    if (pipe) hipLaunchKernelGGL((gemm_glds_kernel<E, LAYOUT, 64, 2, 1>), grid, dim3(256), 0, st, a, no_group);
    else hipLaunchKernelGGL((gemm_glds_kernel<E, LAYOUT, 64>), grid, dim3(256), 0, st, a, no_group);
  }
}

// measurement aid of tests/probes/gemm_stamps.py (not declared in include/d2r_hip.h): workgroup (0,0) of the next LDS-DMA launches
// leaves s_memtime stamps per wave in dst[wave*8 + i]: 0 entry, 1 first tile landed, 2 K-loop done, 3 stores drained
static unsigned long long* g_gemm_stamps = nullptr;
extern "C" void d2r_gemm_debug_stamps(unsigned long long* dst) { g_gemm_stamps = dst; }

// Returns 1 when the launch was taken by the LDS-DMA kernel, 0 when the shape is not eligible.
int d2r_gemm_glds_try(const GemmArgs& a, int layout, int batch, int bn, hipStream_t st) {
  if (!d2r_is16(a.dtype) || batch != 1 || a.K % 64 != 0 || a.K < 128 || a.M < 128 || a.N < 64) return 0;
  if ((bn % 1000) == 129 && a.N < 128) return 0;
  if (!a.vecA || !a.vecB) return 0;
  if (a.G && !(a.vecC && a.c_dtype == a.dtype)) return 0;  // the activation-gradient epilogue is in the vectorised path only
  const bool a_strided = layout == D2R_GEMM_TN, b_strided = layout != D2R_GEMM_NT;
  if ((a_strided && a.M % 8 != 0) || (b_strided && a.N % 8 != 0)) return 0;
  const bool f16 = a.dtype == D2R_F16;
  GemmArgs ab = a;
  ab.ts = g_gemm_stamps;
  {
    // column bands for wide outputs: the B panels of a band (band x BN x K x 2 bytes) should take about a third of the 4 MB L2
    const int bnw = (bn % 1000) == 64 ? 64 : 128;
    const int gx = d2r_cdiv(a.N, bnw);
    const int64_t panel = (int64_t)bnw * a.K * 2;
    int band = (int)((int64_t)(1536 << 10) / (panel > 0 ? panel : 1));
    if (band < 2) band = 2;
    // (not for deep reductions: at K = 3072 every band re-streams A row panels as large as the B band itself - measured 2-6 % slower
    //  alone on the GPU and +2 GB per step past L2 for the 128x64 dX kernel)
    ab.band = (a.K <= 1536 && gx > band && gx * panel > (3 << 20)) ? band : 0;
  }
  d2r_gemm_variant_tl = ((bn % 1000) == 64 ? 1 : (bn % 1000) == 128 ? 2 : 3) + (bn >= 1000 ? 10 : 0);
  switch (layout) {
    case D2R_GEMM_NT: f16 ? launch_glds<f16_t, D2R_GEMM_NT>(ab, bn, st) : launch_glds<bf16_t, D2R_GEMM_NT>(ab, bn, st); break;
    case D2R_GEMM_NN: f16 ? launch_glds<f16_t, D2R_GEMM_NN>(ab, bn, st) : launch_glds<bf16_t, D2R_GEMM_NN>(ab, bn, st); break;
    case D2R_GEMM_TN: f16 ? launch_glds<f16_t, D2R_GEMM_TN>(ab, bn, st) : launch_glds<bf16_t, D2R_GEMM_TN>(ab, bn, st); break;
    default: return 0;
  }
  return 1;
}


// Grouped weight gradients on the LDS-DMA kernel: `n` (<= 16) same-shape TN problems, 128 x 128 tiles, software-pipelined
// K-loop over the token rows, fp32 accumulate epilogue, bias gradients.  Returns 1 when taken, 0 when the shape is not
// eligible (then the generic 64 x 64 kernel runs).
int d2r_gemm_glds_wgrad_try(const GemmArgs& a, const GemmGroup& grp, int n, hipStream_t st) {
  if (!d2r_is16(a.dtype) || a.K < 128 || a.M < 128 || a.N < 128 || !a.vecA || !a.vecB || a.M % 8 != 0 || a.N % 8 != 0) return 0;
  d2r_gemm_variant_tl = 20;
  dim3 grid(d2r_cdiv(a.N, 128), d2r_cdiv(a.M, 128), n);
  if (a.dtype == D2R_F16) hipLaunchKernelGGL((gemm_glds_kernel<f16_t, D2R_GEMM_TN, 128, 2, 1, 1>), grid, dim3(256), 0, st, a, grp);
  else hipLaunchKernelGGL((gemm_glds_kernel<bf16_t, D2R_GEMM_TN, 128, 2, 1, 1>), grid, dim3(256), 0, st, a, grp);
  return 1;
}


// Grouped + batched TN products with a 16-bit output (mode 2 of the kernel): `ngroups` (<= 32) operand triples, `a.gbatch`
// problems each at the batch strides a.sAb / sBb / sCb.  Returns 1 when taken.
int d2r_gemm_glds_batched_tn_try(const GemmArgs& a, const GemmGroup& grp, int ngroups, hipStream_t st) {
  if (!d2r_is16(a.dtype) || a.c_dtype != a.dtype || !a.vecA || !a.vecB || !a.vecC || a.M % 8 != 0 || a.N % 8 != 0 || a.M < 8 || a.N < 8 || a.K < 1) return 0;
  if (ngroups < 1 || ngroups > D2R_GEMM_GROUP_MAX || a.gbatch < 1 || (int64_t)ngroups * a.gbatch > 65535) return 0;
  dim3 grid(d2r_cdiv(a.N, 128), d2r_cdiv(a.M, 128), ngroups * a.gbatch);
  d2r_gemm_variant_tl = 22;
  if (a.dtype == D2R_F16) hipLaunchKernelGGL((gemm_glds_kernel<f16_t, D2R_GEMM_TN, 128, 2, 1, 2>), grid, dim3(256), 0, st, a, grp);
  else hipLaunchKernelGGL((gemm_glds_kernel<bf16_t, D2R_GEMM_TN, 128, 2, 1, 2>), grid, dim3(256), 0, st, a, grp);
  return 1;
}
