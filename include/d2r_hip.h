/*
 * d2r_hip.h — C ABI of libd2r_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the D2R
 * dual-branch dynamic-routing forward/backward hot path.
 *
 * The reference (SorF520/D2R) has NO plugin / operator / FFI layer: its boundary is two Python classes
 * (`UnimoModelF.forward`, models/unimo_model.py:149-162; `MSDTrainer`, modules/train.py:53-328) and every
 * op is an ATen call.  This header is therefore the seam a maintainer of the reference would bind with
 * `ctypes`: each entry point replaces one ATen op *sequence* of the reference, cited
 * per function as reference file:line.
 *
 * Conventions:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless named `h_*`;
 *   - kernels are enqueued on the caller's `stream` (a hipStream_t passed as void*); the library never
 *     allocates, frees, synchronises or takes ownership — workspaces are caller-provided;
 *   - return 0 on success, a negative d2r_status on error; d2r_last_error() gives a thread-local message;
 *   - dtype of activations/weights `T` is D2R_F32, D2R_BF16 or D2R_F16 (IEEE half: BASELINE.json configs[4]); every
 *     entry point that says "16-bit" takes either of the two and runs the same kernel with the other MFMA operand type;
 *     accumulation, softmax statistics, router logits, biases, LayerNorm parameters and all reductions are fp32;
 *   - deterministic: fixed reduction order, no float atomics anywhere (the embedding-table gradient is a fixed-order gather-sum).
 */
#ifndef D2R_HIP_H
#define D2R_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { D2R_F32 = 0, D2R_BF16 = 1, D2R_F16 = 2 } d2r_dtype;

typedef enum {
  D2R_OK = 0,
  D2R_ERR_INVALID = -1,     /* bad argument (shape, alignment, null pointer, unsupported combination) */
  D2R_ERR_LAUNCH = -2,      /* hipGetLastError() after launch */
  D2R_ERR_WORKSPACE = -3    /* caller-provided workspace too small */
} d2r_status;

typedef enum {
  D2R_ACT_NONE = 0,
  D2R_ACT_RELU = 1,
  D2R_ACT_TANH = 2,
  D2R_ACT_GELU = 3,        /* erf GELU  (BertIntermediate, models/modeling_unimo.py:443-456) */
  D2R_ACT_QUICK_GELU = 4,  /* x*sigmoid(1.702x) (CLIPMLP, models/modeling_unimo.py:121-133) */
  D2R_ACT_TANH_RELU = 5,   /* relu(tanh(x)) = Router activateFunc (models/Router.py:6-8) */
  D2R_ACT_SIGMOID = 6
} d2r_act;

/* A-operand / B-operand storage of d2r_gemm: C[M,N] = sum_k A(m,k) * B(k,n) */
typedef enum {
  D2R_GEMM_NT = 0,  /* A stored [M,K] (k contiguous), B stored [N,K] (k contiguous): y = x W^T (nn.Linear fwd) */
  D2R_GEMM_NN = 1,  /* A stored [M,K], B stored [K,N] (n contiguous): dX = dY W ; O = P V             */
  D2R_GEMM_TN = 2   /* A stored [K,M] (m contiguous), B stored [K,N]: dW = dY^T X ; dV = P^T dO        */
} d2r_gemm_layout;

const char* d2r_version(void);
const char* d2r_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * K11  gemm_bias_act — every nn.Linear of the path, and the attention matmuls, as one MFMA kernel family.
 * Replaces: F.linear (+ReLU/tanh/GELU/quick_gelu, + residual add) e.g. models/SelfAttention.py:29-31,52-53,
 * models/Refinement.py:105-107,133-137, models/XModules.py:300-302, models/Cells.py:145-160,236-246,
 * models/modeling_unimo.py:173-176,217,353-355,411,451,466; torch.bmm/matmul at models/XModules.py:305,310,
 * models/SelfAttention.py:33,39, models/Cells.py:244-246; and their autograd backward (NN / TN layouts).
 *
 *   C[b,h] = epilogue( alpha * A[b,h] x B[b,h] )      batch index z = b*nh + h, pointer offset b*s?b + h*s?h
 *   epilogue(v) = act(v + bias[n]) + residual[m,n]     (then  C = v + beta*C_old  when beta != 0)
 * `preact` (optional) receives v + bias before the activation (needed by GELU backward).
 * All leading dimensions / strides are in ELEMENTS.  A and B have dtype `dtype`; C/residual/preact have
 * `c_dtype`.  bias is fp32 [N] (row b*s_bias_b for batch b) or NULL.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int dtype;      /* d2r_dtype of A and B */
  int c_dtype;    /* d2r_dtype of C, residual, preact */
  int layout;     /* d2r_gemm_layout */
  int act;        /* d2r_act */
  int M, N, K;
  int nb, nh;     /* batch = nb*nh (use 1,1 for a plain GEMM) */
  float alpha, beta;
  const void* A; int64_t lda, sAb, sAh;
  const void* B; int64_t ldb, sBb, sBh;
  void* C;       int64_t ldc, sCb, sCh;
  const float* bias;
  const void* residual; int64_t ldr, sRb, sRh;
  void* preact;  /* same ld/strides as C */
  /* optional scratch for deterministic split-K (GEMMs with few output tiles and a long reduction: weight
   * gradients, M<=32 router/pooler products; 16-bit NT / NN products with K >= 2048 over at most 448 tiles of 128 x 128, whose
   * partial sums meet inside the launch); used only when batch == 1.  NULL disables split-K.  The LAST 4 KiB of the workspace
   * hold the tile counters of the in-launch variant: they must be zero before the first use and are zero again when a launch has
   * finished; launches that share a workspace must be ordered (one stream). */
  void* workspace; size_t workspace_bytes;
  int64_t s_bias_b;  /* bias stride per outer batch index b (grouped linears: one bias row per group); 0 = shared */
  /* TN layout, batch == 1 only: dbias[m] += sum_k A[k,m] (fp32 [M], ACCUMULATED) — the bias gradient of a linear
   * layer computed inside its weight-gradient GEMM (dW = dY^T X reads dY anyway), replacing a separate column-sum
   * launch.  NULL disables. */
  float* dbias;
  /* optional, batch == 1: the result (after bias / act) is multiplied elementwise by d act(grad_act) evaluated at
   * grad_ref[m,n] (dtype and leading dimension of C; the pre-activation for gelu / quick_gelu, the activation
   * output for relu / tanh / sigmoid) — the activation backward of the PREVIOUS linear fused into this dX GEMM. */
  const void* grad_ref;
  int grad_act;
} d2r_gemm_desc;

int d2r_gemm(const d2r_gemm_desc* d, void* stream);
/* `n` INDEPENDENT products enqueued together (no output of one is an operand or an output of another; the caller guarantees it).
 * The 16-bit NT / NN products with at least 128 rows and columns leave as grouped launches of up to 16 problems on the 128 x 128
 * LDS-DMA tiles - the chains of 768 x 768 linears of the routing cells (models/Cells.py:30-255, models/Refinement.py:133-154) are
 * 192-300 tiles each, one partial round of workgroups; four to six of them together fill the chip - and every tile is computed
 * exactly as by d2r_gemm: bit-identical results.  Everything else is launched by d2r_gemm, one by one, in the order given. */
int d2r_gemm_group(const d2r_gemm_desc* descs, int n, void* stream);
/* `count` weight-gradient GEMMs of ONE shape in launches of up to 16 problems:
 *   C_i[M,N] (fp32, ldc) = beta * C_i + A_i^T B_i,   A_i [K,M] (lda), B_i [K,N] (ldb) of dtype;   dbias_i[m] += sum_k A_i[k,m]
 * h_A / h_B / h_C / h_dbias are HOST arrays of device pointers (h_dbias may be NULL).  Replaces `count` d2r_gemm TN
 * calls (dW = dY^T X of the 768x768 linears in models/Cells.py, Refinement.py, SelfAttention.py, XModules.py) that
 * each needed split-K slabs and a reduce launch to fill the chip; the caller defers them, nothing in the backward
 * pass reads a weight gradient.  Deterministic. */
int d2r_gemm_tn_grouped(int dtype, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, const void* const* h_A,
                        const void* const* h_B, float* const* h_C, float* const* h_dbias, int count, float beta,
                        void* stream);
/* Weight gradients of DIFFERENT shapes in one call: problem i is C_i[M_i,N_i] (fp32, ldc_i) = beta * C_i + A_i^T B_i with A_i [K_i,M_i]
 * (lda_i), B_i [K_i,N_i] (ldb_i) of dtype, dbias_i[m] += sum_k A_i[k,m] (h_dbias may be NULL).  All arrays are HOST arrays of `count`
 * entries.  The 16-bit problems with at least 128 rows, columns and reduction rows leave together on 256 x 256 tiles (launches of up
 * to 40 problems: the deferred weight gradients of a whole branch - models/modeling_unimo.py:334-470 and the cells' linears - fill
 * the chip whatever the single shapes are); the rest go through d2r_gemm_tn_grouped shape class by shape class.  No two problems may
 * share an output.  Deterministic. */
int d2r_gemm_tn_grouped_v(int dtype, int count, const int* M, const int* N, const int* K, const int64_t* lda, const int64_t* ldb,
                          const int64_t* ldc, const void* const* h_A, const void* const* h_B, float* const* h_C, float* const* h_dbias,
                          float beta, void* stream);
/* (measurement aids - per-launch timers, kernel-choice overrides, cycle stamps - are declared in d2r_hip_probes.h: not part of the
 * drop-in surface, process-global, not thread-safe; the library itself reads no environment variable) */
/* Data parallelism: there is deliberately NO d2r_comm_* entry point.  The gradient reduction of the path is a sum of the flat
 * fp32 gradient buffer over ranks; the host side (d2r_amd/dp.py) issues it as bucketed torch.distributed collectives - RCCL
 * all-reduce, or reduce-scatter + all-gather per bucket - on a communication stream ordered behind the compute streams.  The
 * data movement is RCCL's over xGMI; nothing of it is arithmetic of this library, so nothing of it sits behind this ABI.  A
 * binding from another host language calls its own RCCL on the same buffers (the 1 / world factor is d2r_adamw_step's grad_scale). */

/* ------------------------------------------------------------------------------------------------
 * Row kernels (fp32 statistics, wave-shuffle reductions)
 * ------------------------------------------------------------------------------------------------ */
/* softmax over the last dim of X[rows, cols] (row stride ld): Y = softmax(scale*X + mask[row / rows_per_mask]).
 * Replaces torch.softmax at models/XModules.py:309 (scale 100/sqrt(768)), models/SelfAttention.py:36,
 * models/Cells.py:245,204 and models/modeling_unimo.py:194,376-385 (additive -10000 key mask, fp32 [*,cols]). */
int d2r_softmax_fwd(int x_dtype, int y_dtype, const void* X, void* Y, int64_t ld, int64_t rows, int cols,
                    float scale, const float* mask, int64_t rows_per_mask, void* stream);
/* dS = scale * P o (dP - rowsum(dP o P)); dS has P's dtype, dP may be fp32 */
int d2r_softmax_bwd(int p_dtype, int dp_dtype, const void* P, const void* dP, void* dS, int64_t ld, int64_t rows,
                    int cols, float scale, void* stream);

/* LayerNorm over D (models/modeling_unimo.py:231,250,283,408,463,742). mean/rstd: fp32 [rows]. */
int d2r_layernorm_fwd(int dtype, const void* X, const float* gamma, const float* beta, float eps, int64_t rows,
                      int D, void* Y, float* mean, float* rstd, void* stream);
size_t d2r_layernorm_bwd_workspace(int64_t rows, int D);
/* dX always; dgamma/dbeta (fp32 [D]) are OVERWRITTEN. workspace: d2r_layernorm_bwd_workspace bytes. */
int d2r_layernorm_bwd(int dtype, const void* dY, const void* X, const float* gamma, const float* mean,
                      const float* rstd, int64_t rows, int D, void* dX, float* dgamma, float* dbeta,
                      void* workspace, size_t workspace_bytes, void* stream);

/* l2norm rows: y = x / (sqrt(sum x^2) + 1e-8)  (eps OUTSIDE the root; models/Cells.py:23-27).  norm: fp32 [rows] */
int d2r_l2norm_fwd(int dtype, const void* X, void* Y, float* norm, int64_t rows, int D, void* stream);
int d2r_l2norm_bwd(int dtype, const void* dY, const void* X, const float* norm, void* dX, int64_t rows, int D,
                   void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise kernels (16-byte vectorised)
 * ------------------------------------------------------------------------------------------------ */
/* dX = dY * act'(.)  — `ref` is the activation OUTPUT for relu/tanh/tanh_relu/sigmoid and the PRE-activation
 * for gelu/quick_gelu. */
int d2r_act_bwd(int dtype, int act, const void* dY, const void* ref, void* dX, int64_t n, void* stream);
int d2r_act_fwd(int dtype, int act, const void* X, void* Y, int64_t n, void* stream);
/* out = (a - b)^2 ; da = 2(a-b)dout, db = -da   (models/Cells.py:147,156) */
int d2r_sqdiff_fwd(int dtype, const void* a, const void* b, void* out, int64_t n, void* stream);
int d2r_sqdiff_bwd(int dtype, const void* a, const void* b, const void* dout, void* da, void* db, int64_t n,
                   void* stream);
/* out = a*s + h  (FiLM modulation, models/Refinement.py:136) ; backward gives da, ds (dh = dout) */
int d2r_muladd_fwd(int dtype, const void* a, const void* s, const void* h, void* out, int64_t n, void* stream);
int d2r_muladd_bwd(int dtype, const void* a, const void* s, const void* dout, void* da, void* ds, int64_t n,
                   void* stream);
/* out = g*a + (1-g)*b (GESC gate, models/Cells.py:205) ; backward gives dg, da, db */
int d2r_lerp_fwd(int dtype, const void* g, const void* a, const void* b, void* out, int64_t n, void* stream);
int d2r_lerp_bwd(int dtype, const void* g, const void* a, const void* b, const void* dout, void* dg, void* da,
                 void* db, int64_t n, void* stream);
/* nn.Dropout (models/modeling_unimo.py:330,388,413,468): y[i] = keep(i) ? x[i] / (1-p) : 0, plus add[i] when add != NULL
 * (the skip connection that follows the dropout in BertSelfOutput / BertOutput).  keep(i) is a pure function of
 * (seed, i): the backward pass calls the same entry point on dy with the same seed — no mask is stored. */
int d2r_dropout(int dtype, const void* x, const void* add, void* y, int64_t n, float p, uint64_t seed, void* stream);
int d2r_add(int dtype, const void* a, const void* b, void* out, int64_t n, void* stream);
/* two independent problems of one size in one launch (element for element the arithmetic of two d2r_add / d2r_act_bwd calls): the
 * text / image and a / b pairs of per-sample vectors in the routing cells' backward (models/Cells.py:179-218, :222-255) */
int d2r_add2(int dtype, const void* a1, const void* b1, void* out1, const void* a2, const void* b2, void* out2, int64_t n, void* stream);
int d2r_act_bwd2(int dtype, int act, const void* dY1, const void* ref1, void* dX1, const void* dY2, const void* ref2, void* dX2, int64_t n,
                 void* stream);
/* out[0] = sum_k h_coef[k] * x_k[0]  (n <= 8 fp32 device scalars): loss = CE - w1*JS1 - w2*JS2 */
int d2r_lincomb(const float* const* h_x, const float* h_coef, int n, float* out, void* stream);
/* y = alpha*x + beta*y (dtype T), used for gradient accumulation ; cast between dtypes */
int d2r_axpby(int dtype, float alpha, const void* x, float beta, void* y, int64_t n, void* stream);
int d2r_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream);
/* column sums: out[n] (fp32, OVERWRITTEN) = sum_m X[m, n]   (bias gradients). workspace via query. */
size_t d2r_colsum_workspace(int64_t M, int N);
int d2r_colsum(int dtype, const void* X, int64_t ld, int64_t M, int N, float* out, void* workspace,
               size_t workspace_bytes, void* stream);
/* out[n] += sum_m X[m, n] for M <= 32 rows (one row slice): the bias gradient of a per-sample linear added straight into its fp32 sink */
int d2r_colsum_add(int dtype, const void* X, int64_t ld, int64_t M, int N, float* out, void* workspace, size_t workspace_bytes,
                   void* stream);

/* ------------------------------------------------------------------------------------------------
 * K1  router pooling  (models/Router.py:23: x.mean(-2));  pooled: fp32 [B, D]
 * nsrc sources pooled in ONE launch (layer >= 1 pools the six aggregated tensors at once).
 * ------------------------------------------------------------------------------------------------ */
int d2r_meanpool_fwd(int dtype, const void* const* h_srcs, int nsrc, int B, int L, int D, float* pooled /*[nsrc,B,D]*/,
                     void* stream);
/* dX[b,l,:] (+)= dpooled[b,:]/L ; accumulate!=0 adds into dX */
int d2r_meanpool_bwd(int dtype, const float* dpooled, int B, int L, int D, void* dX, int accumulate, void* stream);
/* n <= 8 pooled gradients dpooled[n,B,D] broadcast into n distinct [B,L,D] tensors in one launch (16-bit dtypes); bit j of acc_mask:
 * accumulate into dX[j] instead of overwriting it. */
int d2r_meanpool_bwd_multi(int dtype, const float* dpooled, int n, int B, int L, int D, void* const* dX, unsigned acc_mask, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K8  route_aggregate — path normalisation, threshold gate and aggregation of the cell outputs of one routing layer.
 * Replaces models/DynamicInteraction.py:50-67 (=:119-132, :170-187, :239-252) and the final-layer rule
 * :104-117 (=:224-237).  Cell order [RIC, GLAC, IMRC, CMRC, CRCMC, GESC] (:41-48).
 *   ncell    : number of cells of the layer = the first ncell of that list.  The reference hard-indexes six
 *              (ncell = 6); 2..5 is the declared-subset extension of SURVEY.md section 8c (BASELINE configs[4]: 4 cells):
 *              path normalisation over the ncell cells, final-layer threshold 1e-4/ncell (self.threshold/self.num_cell).
 *   embs[j]  : ncell cell outputs; j=1 (GLAC) and j=5 (GESC) are per-sample [B,D] broadcasts, the rest [B,L,D];
 *              embs[0] is the RIC *input* x0 — relu (models/Cells.py:38) is applied in-kernel.
 *   gates    : fp32 [B, ncell, P] raw router outputs g_j (P = ncell, or 1 for the final layer)
 *   refs     : (final layer only) the ncell layer inputs ref_j [B,L,D] used by the skip term
 *   outs[i]  : P output tensors [B,L,D];  probs: fp32 [B, P, ncell] (normalised for P=ncell, raw for P=1), sample b at
 *              probs + b*ld_probs (ld_probs >= P*ncell: the layers of a module write slices of one [B, total_paths] row)
 * ------------------------------------------------------------------------------------------------ */
int d2r_route_aggregate_fwd(int dtype, const void* const* h_embs /*ncell*/, const void* const* h_refs /*ncell or NULL*/,
                            const float* gates, int B, int L, int D, int ncell, int P, void* const* h_outs /*P*/,
                            float* probs, int64_t ld_probs, void* stream);
size_t d2r_route_aggregate_bwd_workspace(int B, int L, int D, int P);
/* d_embs[j] are OVERWRITTEN (d_embs[0] is w.r.t. the RIC input x0, relu' applied; broadcast ones are [B,D]);
 * d_refs[j] (final layer: gradient of the skip term) OVERWRITTEN, d_gates fp32 [B,ncell,P] OVERWRITTEN.
 * d_probs: fp32 [B,P,ncell] (sample stride ld_dprobs) gradient flowing into the returned `probs` (sim_paths -> JS loss) or NULL;
 * h_outs: forward outputs (only outs[0] of the final layer is read; may be NULL for P=ncell). */
int d2r_route_aggregate_bwd(int dtype, const void* const* h_embs, const void* const* h_refs, const float* gates,
                            const void* const* h_douts /*P*/, const void* const* h_outs, const float* d_probs,
                            int64_t ld_dprobs, int B, int L, int D, int ncell, int P, void* const* h_dembs /*ncell*/,
                            void* const* h_drefs /*ncell or NULL*/, float* d_gates, void* workspace,
                            size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K6  SAF gate — BatchNorm1d(1) + sigmoid + l1norm over the Lq+1 alignment scores of every sample
 * (models/XModules.py:380-381).  a: fp32 [B, n] pre-BN scores.  train!=0: batch statistics over all B*n
 * scalars and running-stat update (momentum 0.1, unbiased running var); else running stats.
 * stats (fp32[4]) receives {mean, biased var, rstd, sum-of-sigmoid...}; w: fp32 [B, n] attention weights.
 * ------------------------------------------------------------------------------------------------ */
int d2r_saf_gate_fwd(const float* a, int B, int n, const float* bn_weight, const float* bn_bias,
                     float* running_mean, float* running_var, int train, float* w, float* saved /*[2]: mean,rstd*/,
                     void* stream);
int d2r_saf_gate_bwd(const float* a, const float* dw, int B, int n, const float* bn_weight, const float* bn_bias,
                     const float* saved, int train, float* da, float* d_bn_weight, float* d_bn_bias, void* stream);
/* Global-batch-exact BatchNorm under data parallelism ("optional"): the reference's BatchNorm1d(1) normalises over all
 * B*(Lq+1) scores of the WHOLE batch (models/XModules.py:376,381).  d2r_saf_gate_stats leaves this rank's fp64 sums {sum a, sum a^2}
 * in `sums`; the caller all-reduces them over the ranks and passes the totals (and the global element count) to d2r_saf_gate_fwd_ex.
 * Backward: phase 1 of d2r_saf_gate_bwd_ex stops behind the sigmoid / l1norm part (d y in `da`, this rank's share of the BatchNorm
 * parameter gradients in d_bn_weight / d_bn_bias, fp64 {sum dy, sum dy*xhat} in gsums); after the all-reduce phase 2 finishes `da`
 * with the global sums.  gstats = NULL / phase 0: local-batch statistics (= d2r_saf_gate_fwd / _bwd). */
int d2r_saf_gate_stats(const float* a, int B, int n, double* sums, void* stream);
/* w16 / da16 (optional): a bf16 / fp16 copy of w (forward) and of the finished da (backward: phase 0 or 2), rounded as d2r_cast would -
 * their consumers read them as GEMM operands.  accumulate != 0: the BatchNorm parameter gradients are ADDED to d_bn_weight / d_bn_bias
 * (the caller's fp32 gradient sinks) instead of overwriting them. */
int d2r_saf_gate_fwd_ex(const float* a, int B, int n, const float* bn_weight, const float* bn_bias, float* running_mean,
                        float* running_var, int train, float* w, float* saved, const double* gstats, double ntotal, void* w16,
                        int w16_dtype, void* stream);
int d2r_saf_gate_bwd_ex(const float* a, const float* dw, int B, int n, const float* bn_weight, const float* bn_bias,
                        const float* saved, int train, float* da, float* d_bn_weight, float* d_bn_bias, int phase, double* gsums,
                        double ntotal, void* da16, int da16_dtype, int accumulate, void* stream);

/* The two rank-one products around the gate in the BACKWARD pass of the SAF-weighted sum wsum[b] = w[b] @ S[b]
 * (models/XModules.py:382-384; S [B,n,E], 16-bit): dw[b,i] = <dwsum[b,:], S[b,i,:]> (fp32 [B,n]) and
 * dS[b,i,:] = w[b,i] * dwsum[b,:] + da[b,i] * w_saf[:] (w 16-bit [B,n], da fp32 [B,n], w_saf = attn_sim_w.weight [E], 16-bit). */
int d2r_saf_dweights(int dtype, const void* dwsum, const void* S, int B, int n, int E, float* dw, void* stream);
int d2r_saf_dscores(int dtype, const void* w, const void* dwsum, const float* da, const void* w_saf, int B, int n, int E, void* dS,
                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * K9  js_div on two [B,B] logit matrices (models/XModules.py:32-41) and K13 cross-entropy
 * (models/unimo_model.py:147,160).  All fp32; single-workgroup latency kernels.
 * ------------------------------------------------------------------------------------------------ */
int d2r_jsdiv_fwd(const float* p_logits, const float* q_logits, int B, float* out /*[1]*/, void* stream);
int d2r_jsdiv_bwd(const float* p_logits, const float* q_logits, int B, const float* dout /*[1]*/, float* dp,
                  float* dq, void* stream);
int d2r_ce_fwd(const float* logits, const int64_t* labels, int B, int C, float* loss /*[1]*/, void* stream);
int d2r_ce_bwd(const float* logits, const int64_t* labels, int B, int C, const float* dloss /*[1]*/,
               float* dlogits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K10 Block fusion core (models/XModules.py:541-549): z[b,c,s] = sum_r m0[b,c,r,s]*m1[b,c,r,s];
 * signed sqrt; L2-normalise over s.  m0,m1: [B, C, R*S] (dtype T), out: [B, C*S] (dtype T), zraw fp32 [B,C*S].
 * ------------------------------------------------------------------------------------------------ */
int d2r_block_merge_fwd(int dtype, const void* m0, const void* m1, int B, int C, int R, int S, void* out,
                        float* zraw, void* stream);
int d2r_block_merge_bwd(int dtype, const void* m0, const void* m1, const float* zraw, const void* dout, int B,
                        int C, int R, int S, void* dm0, void* dm1, void* stream);

/* As d2r_layernorm_bwd, plus: dX += dres when dres != NULL (the gradient arriving through the skip connection
 * around the block this LayerNorm belongs to), and dgamma/dbeta ACCUMULATED when accumulate != 0. */
int d2r_layernorm_bwd_ex(int dtype, const void* dY, const void* X, const float* gamma, const float* mean,
                         const float* rstd, int64_t rows, int D, void* dX, const void* dres, float* dgamma,
                         float* dbeta, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* dgamma == dbeta == NULL in d2r_layernorm_bwd_ex DEFERS the second stage: the per-block partial sums stay in `workspace` (which
 * then has to outlive the call) and this entry point sums them for n LayerNorms of one shape (rows, D) in one launch per 32
 * problems; dgamma[i] / dbeta[i] overwritten, or accumulated when accumulate != 0.  Bit-identical to the undeferred sum. */
int d2r_layernorm_bwd_sum_grouped(const void* const* partials, float* const* dgamma, float* const* dbeta, int n, int64_t rows,
                                  int D, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3 fused multi-head attention core (16-bit dtypes, head_dim 64 or 48, Lq, Lk <= 1024: up to 256 tokens with the whole
 * head resident in LDS, above that a loop over 128-row blocks with the online softmax — BASELINE configs[3] / [4])
 *   O[b,:,h] = softmax(scale * Q_h K_h^T + mask[b]) V_h (+ residual)
 * Replaces BertSelfAttention scores/softmax/context (models/modeling_unimo.py:385-424), CLIPAttention (:150-215)
 * and the 16-head attention of models/SelfAttention.py:20-60 — scores and probabilities never reach HBM.
 * q/k/v/o/residual/dO/dq/dk/dv: bf16 [B, L, *] views given as (pointer to column 0 of head 0, row stride,
 * batch stride) in elements; head h occupies columns [h*head_dim, (h+1)*head_dim).  mask: fp32 additive [B,Lk] or
 * NULL.  lse: fp32 [B,H,Lq] row log-sum-exp written by fwd and read by bwd.  dsum (bwd): fp32 [B,H,Lq] scratch the
 * long-sequence backward hands from its dQ kernel to its dK/dV kernel; may be NULL when Lq, Lk <= 256.  Pointers 16-byte
 * aligned, strides multiples of 8 elements.  Deterministic.
 * p_drop / seed: dropout on the probabilities (attention_probs_dropout_prob, models/modeling_unimo.py:388) inside the kernel:
 * probability (b, h, q, key) is kept iff the counter-based generator of d2r_dropout, at element index
 * ((b*H + h)*Lq + q)*lkp + key with lkp = Lk rounded up to 8, says so, and is then scaled by 1/(1-p); the backward
 * regenerates the mask from the same (p_drop, seed).  p_drop = 0: no dropout.
 * ------------------------------------------------------------------------------------------------ */
int d2r_mha_supported(int dtype, int Lq, int Lk, int head_dim);
int d2r_mha_fwd(int dtype, const void* q, int64_t ldq, int64_t sqb, const void* k, int64_t ldk, int64_t skb,
                const void* v, int64_t ldv, int64_t svb, void* o, int64_t ldo, int64_t sob, const void* residual,
                int64_t ldr, int64_t srb, const float* mask, float* lse, int B, int H, int Lq, int Lk, int head_dim,
                float scale, float p_drop, uint64_t seed, void* stream);
int d2r_mha_bwd(int dtype, const void* q, int64_t ldq, int64_t sqb, const void* k, int64_t ldk, int64_t skb,
                const void* v, int64_t ldv, int64_t svb, const void* dO, int64_t ldg, int64_t sgb, const float* mask,
                const float* lse, float* dsum, void* dq, int64_t lddq, int64_t sdqb, void* dk, int64_t lddk, int64_t sdkb,
                void* dv, int64_t lddv, int64_t sdvb, int B, int H, int Lq, int Lk, int head_dim, float scale, float p_drop,
                uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K2 / K4 fused single-head attention over the full 768-wide feature (16-bit dtypes, D = 768, Lk <= 640)
 *   O[b] = softmax(scale * Q K^T + mask[b]) V (+ residual)
 * Replaces the CrossModalAlignment core (models/XModules.py:300-310 = models/Refinement.py:105-115, scale
 * 100/sqrt(768)) and the ContextRichCrossModalCell core (models/Cells.py:244-246, scale 1, residual Qs): three
 * launches (QK^T GEMM, softmax, PV GEMM) and an fp32 [B,Lq,Lk] round trip become one launch.
 * q/k/v/o/residual/dO/dq: bf16 [B, L, 768] views as (pointer, row stride, batch stride) in elements.
 * lse: fp32 [B, Lq].  d2r_xattn_bwd writes dq plus P and dS (bf16 [B, Lq, lkp], lkp = Lk rounded up to 8) for
 * the key-side products dV = P^T dO and dK = dS^T Q, which the caller runs as batched d2r_gemm (TN) launches.
 * ------------------------------------------------------------------------------------------------ */
int d2r_xattn_supported(int dtype, int Lq, int Lk, int D);
int d2r_xattn_fwd(int dtype, const void* q, int64_t ldq, int64_t sqb, const void* k, int64_t ldk, int64_t skb,
                  const void* v, int64_t ldv, int64_t svb, void* o, int64_t ldo, int64_t sob, const void* residual,
                  int64_t ldr, int64_t srb, const float* mask, float* lse, int B, int Lq, int Lk, int D, float scale,
                  void* stream);
int d2r_xattn_bwd(int dtype, const void* q, int64_t ldq, int64_t sqb, const void* k, int64_t ldk, int64_t skb,
                  const void* v, int64_t ldv, int64_t svb, const void* dO, int64_t ldg, int64_t sgb, const float* mask,
                  const float* lse, void* dq, int64_t lddq, int64_t sdqb, void* P, void* dS, int lkp, int B, int Lq,
                  int Lk, int D, float scale, void* stream);

/* Several attention problems ("cores") of IDENTICAL shape and strides in one launch: the three cross-modal alignment cores of a
 * routing layer (GlobalLocalAlignmentCell, CrossModalRefinementCell, ContextRichCrossModalCell: models/Cells.py:147,85,238 all
 * call the same softmax(100 q k^T / sqrt(768)) v on their own projections of the same two token tensors) give three times
 * the workgroups of one - a single core leaves half of the 256 CUs idle at B = 32.  h_* are HOST arrays of `ncore` (1..4)
 * device pointers; h_residual may be NULL (or hold NULLs).  d2r_xattn_bwd_multi also runs the key-side products
 * dV = P^T dO and dK = dS^T Q of every sample and core (ONE grouped, batched launch of the LDS-DMA GEMM kernel when dk / dv
 * share their strides, e.g. the two halves of a packed k|v gradient), so it returns dq, dk and dv; h_P / h_dS are scratch
 * (16-bit [B, Lq, lkp] each, lkp = Lk rounded up to 8).  With h_o given (and Lk <= 256) the third-generation kernels run
 * (xattn3.hip: queries split over the waves, K / V streamed once through a six-slot LDS-DMA ring, D = rowsum(dO o (O - residual))
 * from the saved output); without it the second-generation query-side kernel, which recomputes D from P and dP. */
int d2r_xattn_fwd_multi(int dtype, int ncore, const void* const* h_q, int64_t ldq, int64_t sqb, const void* const* h_k, int64_t ldk,
                        int64_t skb, const void* const* h_v, int64_t ldv, int64_t svb, void* const* h_o, int64_t ldo, int64_t sob,
                        const void* const* h_residual, int64_t ldr, int64_t srb, const float* mask, float* const* h_lse, int B, int Lq,
                        int Lk, int D, float scale, void* stream);
int d2r_xattn_bwd_multi(int dtype, int ncore, const void* const* h_q, int64_t ldq, int64_t sqb, const void* const* h_k, int64_t ldk,
                        int64_t skb, const void* const* h_v, int64_t ldv, int64_t svb, const void* const* h_dO, int64_t ldg, int64_t sgb,
                        const void* const* h_o /* the forward outputs (residual included), or NULL */, int64_t ldo, int64_t sob,
                        const void* const* h_residual /* what the forward added, or NULL */, int64_t ldr, int64_t srb,
                        const float* mask, const float* const* h_lse, void* const* h_dq, int64_t lddq, int64_t sdqb, void* const* h_dk,
                        int64_t lddk, int64_t sdkb, void* const* h_dv, int64_t lddv, int64_t sdvb, void* const* h_P, void* const* h_dS,
                        int lkp, int B, int Lq, int Lk, int D, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K15 one transformer encoder layer per call (16-bit dtypes): BertLayer.forward (models/modeling_unimo.py:473-512,
 * post-LayerNorm, GELU) and CLIPEncoderLayer.forward (:222-268, pre-LayerNorm, quick_gelu), forward or backward.
 * Same kernels, same order as the single-op entry points; the 7 forward / ~16 backward launches are issued from
 * C++ in one call, skip-connection gradients ride in GEMM / LayerNorm epilogues, parameter gradients accumulate
 * into the caller's fp32 sinks.  All activation buffers are [B*L, *] row-major in `dtype`.
 *   post-LN: h1 = x + attn(x) ; n1 = LN1(h1) ; h2 = n1 + ffn(n1) ; y = LN2(h2)
 *   pre-LN : n1 = LN1(x) ; h1 = x + attn(n1) ; h2 = LN2(h1) ; y = h1 + ffn(h2)
 *   attn(a) = mha(a Wqkv^T + bqkv) Wo^T + bo ;  ffn(a) = act(a W1^T + b1) W2^T + b2
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int dtype;                  /* D2R_BF16 or D2R_F16 */
  int pre_ln;                 /* 0 post-LN (BERT), 1 pre-LN (CLIP ViT) */
  int act;                    /* D2R_ACT_GELU | D2R_ACT_QUICK_GELU */
  int B, L, E, H, F;          /* batch, tokens, hidden, heads, intermediate */
  float eps, scale;           /* LayerNorm eps; attention logit scale (head_dim^-0.5) */
  const float* mask;          /* additive key mask fp32 [B,L] or NULL */
  const void *w_qkv, *w_o, *w_1, *w_2;           /* [3E,E] (q|k|v rows), [E,E], [F,E], [E,F] in dtype */
  const float *b_qkv, *b_o, *b_1, *b_2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  float *gw_qkv, *gw_o, *gw_1, *gw_2, *gb_qkv, *gb_o, *gb_1, *gb_2, *gln1_g, *gln1_b, *gln2_g, *gln2_b; /* bwd: += */
  const void* x;              /* in  [B*L,E] */
  void* y;                    /* out [B*L,E] */
  void *qkv, *ctx, *h1, *n1, *f_pre, *f, *h2;    /* saved by fwd, read by bwd: [T,3E] [T,E] [T,E] [T,E] [T,F] [T,F] [T,E] */
  float *lse, *mean1, *rstd1, *mean2, *rstd2;    /* [B,H,L], [T] x4 */
  const void* dy;             /* bwd in  [B*L,E] */
  void* dx;                   /* bwd out [B*L,E] */
  void* scratch; size_t scratch_bytes;           /* bwd: >= d2r_encoder_layer_bwd_scratch() */
  void* splitk_ws; size_t splitk_bytes;          /* bwd: split-K scratch of the weight-gradient GEMMs (may be NULL) */
  /* bwd: optional second stream (hipStream_t) for the four weight-gradient GEMMs, forked from `stream` inside the
   * call; the caller joins it before anything reads the gradient sinks, and keeps x, the saved activations, dy and
   * scratch alive until it has drained.  NULL: everything on `stream`. */
  void* wgrad_stream;
  /* bwd: != 0 skips the four weight-gradient GEMMs (and bias gradients); the caller launches them later, grouped with
   * the same products of other layers (d2r_gemm_tn_grouped), from o_dy[] and the saved activations. */
  int defer_wgrad;
  /* bwd, written by the call: the output gradients of the qkv / out / fc1 / fc2 linears ([T,3E] [T,E] [T,F] [T,E], inside
   * `scratch` or dy itself) — dW = o_dy^T x with x = x|n1, ctx, n1|h2, f. */
  const void* o_dy[4];
  /* bwd, with defer_wgrad: != 0 also defers the second stage of the two LayerNorm backward passes (their gamma / beta gradients):
   * the call writes the per-block partial sums into `scratch` and reports them in o_lnws[0] (LayerNorm 1) / o_lnws[1] (LayerNorm 2);
   * the caller sums them later with d2r_layernorm_bwd_sum_grouped(rows = B*L, D = E, accumulate = 1) into gln{1,2}_{g,b}. */
  int defer_ln;
  const void* o_lnws[2];
  /* training-time dropout of the BERT layer (models/modeling_unimo.py:388 on the attention probabilities, :413 / :468 on
   * the two dense outputs in front of their residual adds); 0 disables.  The masks are functions of (seed, element index)
   * as in d2r_dropout: the probabilities use seed_attn inside the fused attention core, the dense outputs seed_hidden[0]
   * (attention output) and seed_hidden[1] (FFN output) with the element index in the [B*L, E] tensor.  The backward
   * call regenerates them from the same values. */
  float p_attn, p_hidden;
  uint64_t seed_attn, seed_hidden[2];
} d2r_encoder_layer_desc;
size_t d2r_encoder_layer_bwd_scratch(int B, int L, int E, int F);
int d2r_encoder_layer_fwd(const d2r_encoder_layer_desc* d, void* stream);
int d2r_encoder_layer_bwd(d2r_encoder_layer_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K16 one whole (Reversed_)InteractionModule per call (16-bit compute dtype): the DR_step routing layers of
 * models/InteractionModule.py:22-55 (=:75-108) — every router (models/Router.py:22-26), the up to six cells of each layer
 * (models/Cells.py:30-255 with SelfAttention.py / Refinement.py / XModules.py:277-394), the path normalisation, threshold
 * gates and aggregation (models/DynamicInteraction.py:37-69, 90-134) and the concatenated path probabilities
 * (InteractionModule.py:33-53) — forward, or backward, issued from C++ in ONE call: the same kernels as the single-op
 * entry points above, without one Python autograd node and one foreign call per launch.  In the backward pass every
 * multi-consumer gradient (the module's two inputs, a layer's aggregated outputs) is accumulated in GEMM epilogues
 * (beta = 1 / residual operand) instead of separate add launches, and the 768x768-class weight gradients are queued and
 * launched grouped (d2r_gemm_tn_grouped) at the end of the call.  Parameter gradients ACCUMULATE into the caller's fp32
 * sinks.  `own` is the branch's own modality (text for InteractionModule, image tokens for the reversed module).
 * Requires the fused attention cores for the shapes: d2r_xattn_supported(Lq,Lk), (Lq,Lq) and d2r_mha_supported(Lq,Lq,48).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const void* w;   /* [N,K] weight in the compute dtype (router linears: fp32 masters) */
  const float* b;  /* fp32 [N] */
  float* gw;       /* fp32 [N,K] gradient sink, accumulated */
  float* gb;       /* fp32 [N] gradient sink, accumulated */
} d2r_linear_params;
/* linears of one routing layer, in this order (same-input projections are ONE fused linear: rows back to back) */
enum {
  D2R_RL_R0 = 0,      /* the ncell routers' mlp.0, fused [ncell*hid, 768] fp32 */
  D2R_RL_R2,          /* the ncell routers' mlp.2, fused [ncell*P, hid] fp32 (one group per cell) */
  D2R_RL_IMRC_QKV,    /* sa.att_layer.linears.0|1|2  [2304,768] */
  D2R_RL_IMRC_FC1, D2R_RL_IMRC_FC2,
  D2R_RL_GLAC_Q, D2R_RL_GLAC_KV /* key|value [1536,768] */, D2R_RL_GLAC_LOC /* fc_sim_tranloc */, D2R_RL_GLAC_FC1,
  D2R_RL_GLAC_TPOOL /* text_cls_pool.dense (applied to OWN tokens) */, D2R_RL_GLAC_IPOOL /* image_cls_pool.dense (OTHER) */,
  D2R_RL_GLAC_GLO /* fc_sim_tranglo */, D2R_RL_GLAC_FC2, D2R_RL_GLAC_SAFW /* SAF_module.attn_sim_w [1,768] */,
  D2R_RL_CMRC_Q, D2R_RL_CMRC_KV, D2R_RL_CMRC_SCALE, D2R_RL_CMRC_SHIFT, D2R_RL_CMRC_FC1, D2R_RL_CMRC_FC2,
  D2R_RL_CRCMC_Q, D2R_RL_CRCMC_KV, D2R_RL_CRCMC_MLP1, D2R_RL_CRCMC_MLP2, D2R_RL_CRCMC_FC1, D2R_RL_CRCMC_FC2,
  D2R_RL_GESC_TPOOL, D2R_RL_GESC_IPOOL, D2R_RL_GESC_MLP0, D2R_RL_GESC_MLP2,
  D2R_RL_NLIN
};
typedef struct {
  d2r_linear_params lin[D2R_RL_NLIN];   /* entries of absent cells (declared subset) are ignored */
  const float *bn_weight, *bn_bias;     /* GLAC SAF BatchNorm1d(1) */
  float *bn_running_mean, *bn_running_var;
  float *g_bn_weight, *g_bn_bias;       /* fp32 [1] sinks, accumulated */
} d2r_routing_layer_params;
typedef struct {
  int dtype;                 /* D2R_BF16 */
  int B, Lq, Lk;             /* batch, own tokens, other tokens (hidden size is 768: models/Cells.py:140-143) */
  int ncell, nlayer;         /* cells per layer (2..6), routing layers = DR_step (>= 2): layer 0, middle layers, final layer */
  int hid_router, heads_imrc, hid_imrc;
  int train;                 /* BatchNorm: batch statistics + running-stat update (1) or running statistics (0) */
  const d2r_routing_layer_params* layers;  /* HOST array [nlayer] */
  const void* own;           /* [B,Lq,768] */
  const void* other;         /* [B,Lk,768] */
  void* out;                 /* [B,Lq,768]  fwd: written; bwd: read (final-layer aggregation backward) */
  float* paths;              /* fp32 [B, ncell*ncell*(nlayer-1) + ncell]  fwd: written */
  void* arena; size_t arena_bytes;        /* saved activations: fwd writes, bwd reads; >= d2r_interaction_arena_bytes() */
  void* splitk_ws; size_t splitk_bytes;   /* split-K scratch of small-M / weight-gradient GEMMs (may be NULL) */
  /* backward only */
  const void* d_out;         /* [B,Lq,768] or NULL (= zero) */
  const float* d_paths;      /* fp32 [B, total_paths] or NULL */
  void* d_own; void* d_other;             /* [B,Lq,768], [B,Lk,768]  OVERWRITTEN */
  void* scratch; size_t scratch_bytes;    /* >= d2r_interaction_bwd_scratch() */
  /* The key | value projections of `other` of EVERY alignment cell (GLAC, CMRC, CRCMC) of EVERY layer as one fused linear: rows
   * [k | v] of (layer 0: GLAC, CMRC, CRCMC), (layer 1: ...), ... i.e. [nkv * 1536, 768] with nkv = cells-with-alignment * nlayer
   * (`other` is the same tensor in every layer, models/DynamicInteraction.py:95-102): one GEMM with N = 13,824 at DR_step 3 in
   * the forward pass, one dX and one dW product in the backward pass.  The D2R_RL_*_KV entries of `layers` are then unused. */
  d2r_linear_params kv_all;
  /* Optional, data parallelism in the global-batch-exact mode: the BatchNorm1d(1) of every GLAC cell takes its statistics over the
   * samples of ALL ranks.  bn_sync(user, pair, stream) must SUM the two fp64 values at the device address `pair` over the ranks,
   * ordered on `stream` (e.g. an RCCL all-reduce), and return 0; it is called twice per routing layer and direction from inside the
   * call (the library itself has no communication).  bn_sync_buf: device scratch of 4 doubles per routing layer; bn_world: number
   * of ranks (the global score count is bn_world * B * (Lq + 1)).  bn_sync = NULL: local-batch statistics. */
  int (*bn_sync)(void* user, double* pair, void* stream);
  void* bn_sync_user;
  double* bn_sync_buf;
  int bn_world;
} d2r_interaction_desc;
int d2r_interaction_supported(int dtype, int Lq, int Lk, int ncell, int heads_imrc);
size_t d2r_interaction_arena_bytes(int B, int Lq, int Lk, int ncell, int nlayer, int hid_router, int hid_imrc);
size_t d2r_interaction_bwd_scratch(int B, int Lq, int Lk, int ncell, int nlayer, int hid_router, int hid_imrc);
int d2r_interaction_fwd(const d2r_interaction_desc* d, void* stream);
int d2r_interaction_bwd(const d2r_interaction_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K17 the classification head as ONE call each way, fp32: Block fusion of the two pooled vectors (models/XModules.py:478-555:
 * linear0 / linear1 [mm, E], `chunks` rank-`rank` merge projections per side, product, sum over the rank, signed square root,
 * per-chunk l2 normalisation, linear_out [E, mm]), fc [classes, E] (models/unimo_model.py:156-158), cross entropy and
 * loss = ce + js (models/unimo_model.py:160).  The same launches in the same order as the single-op entry points
 * (d2r_gemm, d2r_block_merge_*, d2r_ce_*, d2r_lincomb); parameter gradients ACCUMULATE into the caller's fp32 sinks.
 * merge0 / merge1: the `chunks` projections of a side back to back, [chunks * rank * (mm / chunks), mm / chunks].
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int B, E, mm, chunks, rank, classes;
  d2r_linear_params lin0, lin1, merge0, merge1, lin_out, fc;   /* fp32 weights */
  const float* x0; const float* x1;     /* fp32 [B, E]: pooled text / image vectors (Block's two inputs) */
  const int64_t* labels;                /* [B] */
  const float* js;                      /* fp32 [1]: the JS term of the loss */
  float* loss; float* logits; float* pooled;   /* [1], [B, classes], [B, E] (Block's output)  OVERWRITTEN */
  void* arena; size_t arena_bytes;      /* forward activations kept for the backward call: >= d2r_head_arena_bytes() */
  void* splitk_ws; size_t splitk_bytes; /* split-K scratch of the weight-gradient GEMMs (may be NULL) */
  /* backward only */
  const float* d_loss;                  /* fp32 [1] */
  float* d_x0; float* d_x1; float* d_js;       /* [B, E], [B, E], [1]  OVERWRITTEN */
  void* scratch; size_t scratch_bytes;  /* >= d2r_head_bwd_scratch() */
  /* backward, optional: gradients arriving at the OTHER outputs of the head (a second loss on the logits - distillation, label
   * smoothing outside the model - or on Block's output), added to the cross-entropy path: fp32 [B, classes] / [B, E] or NULL */
  const float* d_logits; const float* d_pooled;
} d2r_head_desc;
size_t d2r_head_arena_bytes(int B, int E, int mm, int chunks, int rank, int classes);
size_t d2r_head_bwd_scratch(int B, int E, int mm, int chunks, int rank, int classes);
int d2r_head_fwd(const d2r_head_desc* d, void* stream);
int d2r_head_bwd(const d2r_head_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K12 embeddings (models/modeling_unimo.py:87-118, 272-331)
 * ------------------------------------------------------------------------------------------------ */
/* out[b,l,:] = word[ids[b,l]] + pos[l] + type[tt[b,l]]   (tables fp32, out dtype T; LayerNorm separately) */
int d2r_bert_embed_fwd(int dtype, const int64_t* ids, const int64_t* tt, const float* word, const float* pos,
                       const float* type, int B, int L, int D, int vocab, int ntype, void* out, void* stream);
/* Gradients of the three tables, ACCUMULATED (+=) into dword [vocab,D], dpos [>=L,D], dtype_tab [ntype,D] (fp32, so
 * they may be the flat gradient buffer itself): dword[id] += sum of dY rows of the tokens carrying id, dtype_tab
 * likewise per token type, dpos[l] += sum_b dY[b,l].  Deterministic (fixed summation order, no float atomics).
 * Rows with ids == pad_id get no gradient (padding_idx, models/modeling_unimo.py:277).  D % 4 == 0, D <= 1024. */
int d2r_bert_embed_bwd(int dtype, const void* dY, const int64_t* ids, const int64_t* tt, int B, int L, int D,
                       int ntype, int64_t pad_id, float* dword, float* dpos, float* dtype_tab, void* stream);
/* im2col for the stride=kernel patch conv: pixels fp32 [B,3,H,W] -> patches T [B*(H/p)*(W/p), 3*p*p] */
int d2r_patchify(int dtype, const float* pixels, int B, int H, int W, int p, void* patches, void* stream);
/* x[b,0,:] = cls + pos[0]; x[b,1+i,:] = patch_emb[b,i,:] + pos[1+i]  (in place on x [B,1+np,D], rows 1.. already
 * hold the patch GEMM output) */
int d2r_clip_embed_finish(int dtype, void* x, const float* cls, const float* pos, int B, int ntok, int D,
                          void* stream);
/* dcls[d] = sum_b dX[b,0,d]; dpos[t,d] = sum_b dX[b,t,d]  (fp32, OVERWRITTEN) */
int d2r_clip_embed_bwd(int dtype, const void* dX, int B, int ntok, int D, float* dcls, float* dpos, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K14 fused AdamW over a flat fp32 parameter range (modules/train.py:287-322: torch.optim.AdamW defaults
 * betas=(0.9,0.999), eps=1e-8, weight_decay=1e-2, decoupled) + optional 16-bit shadow copy of the weights.
 * lr already includes the schedule factor; step is the 1-based step count (bias correction).
 * ------------------------------------------------------------------------------------------------ */
int d2r_adamw_step(float* w, const float* g, float* m, float* v, void* w16 /*or NULL*/, int w16_dtype, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                   const int* d_skip /*or NULL*/, void* stream);
/* hipGraph-capturable form: the per-step scalars {lr, 1-beta1^t, sqrt(1-beta2^t), grad_scale} are read from the
 * device array d_hyper[4], which the host refreshes before every graph replay.
 * w16 / w16_dtype: optional 16-bit shadow copy of the weights (D2R_BF16 or D2R_F16) rewritten by the same pass.
 * d_skip: optional device flag; when *d_skip != 0 the launch changes nothing (loss-scaled fp16 training: the step whose
 * gradients overflowed is dropped, as torch.cuda.amp.GradScaler.step does, without a host round trip). */
int d2r_adamw_step_dev(float* w, const float* g, float* m, float* v, void* w16 /*or NULL*/, int w16_dtype, int64_t n,
                       const float* d_hyper, float beta1, float beta2, float eps, float weight_decay,
                       const int* d_skip /*or NULL*/, void* stream);
/* dst[r][0..width) = src[r][0..width) for r < rows; pitches and width in BYTES (16-byte vector path when everything is 16-byte
 * aligned).  One launch - the runtime's device-to-device hipMemcpy2DAsync issues one blit kernel per row for these shapes. */
int d2r_copy_rows(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width, int64_t rows, void* stream);
/* *d_flag |= 1 when g[0..n) holds an inf or a NaN (d_flag is device memory the caller zeroes; one pass over g). */
int d2r_grad_nonfinite(const float* g, int64_t n, int* d_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D2R_HIP_H */
