// encoder_layer.hip — one transformer encoder layer (forward, or backward) per C call.
//
// Replaces (reference file:line): BertLayer.forward (models/modeling_unimo.py:473-512, post-LayerNorm, GELU) and
// CLIPEncoderLayer.forward (:222-268, pre-LayerNorm, quick_gelu) and their autograd backward.  The kernels are the
// ones the single-op entry points launch (d2r_gemm, d2r_mha_*, d2r_layernorm_*, d2r_act_bwd), in the same order;
// what this file removes is the host work between them: the 7 forward / ~16 backward launches of a layer are
// issued from C++ in one call instead of one Python autograd node and one ctypes call each, skip-connection
// gradients ride in GEMM / LayerNorm epilogues instead of separate add kernels, and every parameter gradient is
// accumulated straight into the caller's fp32 sinks.
#include <atomic>
#include <mutex>

#include "common.h"

namespace {

struct Gemm {
  d2r_gemm_desc d;
  Gemm(int dtype, int layout, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
       int64_t ldc, int c_dtype) {
    memset(&d, 0, sizeof(d));
    d.dtype = dtype, d.c_dtype = c_dtype, d.layout = layout, d.act = D2R_ACT_NONE;
    d.M = M, d.N = N, d.K = K, d.nb = 1, d.nh = 1, d.alpha = 1.f, d.beta = 0.f;
    d.A = A, d.lda = lda, d.B = B, d.ldb = ldb, d.C = C, d.ldc = ldc;
  }
};

// y = act(x W^T + b) (+ residual); W [N,K]
int linear_fwd(const d2r_encoder_layer_desc* L, int T, int N, int K, const void* x, const void* w, const float* b,
               void* y, int act, const void* residual, void* preact, void* stream) {
  Gemm g(L->dtype, D2R_GEMM_NT, T, N, K, x, K, w, K, y, N, L->dtype);
  g.d.bias = b, g.d.act = act, g.d.residual = residual, g.d.ldr = N, g.d.preact = preact;
  return d2r_gemm(&g.d, stream);
}
// Fork: `side` waits for everything enqueued on `stream` so far.  Events come from a small ring; a wait captures
// the event's state at the time of the call, so re-recording an event later does not disturb earlier waits.
int fork_stream(void* stream, void* side) {
  constexpr int RING = 64;
  static hipEvent_t ring[RING];
  static std::atomic<unsigned> next{0};
  static std::once_flag once;
  static hipError_t init_err = hipSuccess;
  std::call_once(once, [] {
    for (int i = 0; i < RING && init_err == hipSuccess; ++i) init_err = hipEventCreateWithFlags(&ring[i], hipEventDisableTiming);
  });
  if (init_err != hipSuccess) return d2r_fail(D2R_ERR_LAUNCH, "encoder layer: hipEventCreate failed: %s", hipGetErrorString(init_err));
  hipEvent_t ev = ring[next.fetch_add(1) % RING];
  hipError_t e = hipEventRecord(ev, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)side, ev, 0);
  if (e != hipSuccess) return d2r_fail(D2R_ERR_LAUNCH, "encoder layer: stream fork failed: %s", hipGetErrorString(e));
  return D2R_OK;
}

// dx = dy W (+ residual);  gW += dy^T x, gb += colsum(dy).  The weight-gradient GEMM feeds nothing downstream in
// the backward pass: with a side stream it is forked off BEFORE the dX GEMM and overlaps the rest of the chain.
int linear_bwd(d2r_encoder_layer_desc* L, int which, int T, int N, int K, const void* dy, const void* x, const void* w,
               void* dx, const void* dx_residual, float* gw, float* gb, void* stream, const void* grad_ref = nullptr,
               int grad_act = D2R_ACT_NONE) {
  L->o_dy[which] = dy;  // reported to the caller (deferred weight gradients need it)
  if (L->defer_wgrad) {  // the caller launches dW / db later, grouped with the same product of other layers
    Gemm gx(L->dtype, D2R_GEMM_NN, T, K, N, dy, N, w, K, dx, K, L->dtype);
    gx.d.residual = dx_residual, gx.d.ldr = K;
    gx.d.grad_ref = grad_ref, gx.d.grad_act = grad_act;
    return d2r_gemm(&gx.d, stream);
  }
  void* wstream = stream;
  if (L->wgrad_stream && L->wgrad_stream != stream) {
    if (int rc = fork_stream(stream, L->wgrad_stream)) return rc;
    wstream = L->wgrad_stream;
  }
  Gemm gw_(L->dtype, D2R_GEMM_TN, N, K, T, dy, N, x, K, gw, K, D2R_F32);
  gw_.d.beta = 1.f, gw_.d.dbias = gb, gw_.d.workspace = L->splitk_ws, gw_.d.workspace_bytes = L->splitk_bytes;
  if (int rc = d2r_gemm(&gw_.d, wstream)) return rc;
  Gemm gx(L->dtype, D2R_GEMM_NN, T, K, N, dy, N, w, K, dx, K, L->dtype);
  gx.d.residual = dx_residual, gx.d.ldr = K;
  gx.d.grad_ref = grad_ref, gx.d.grad_act = grad_act;  // dx = (dy W) * act'(grad_ref): the previous layer's act backward
  return d2r_gemm(&gx.d, stream);
}

size_t align256(size_t n) { return (n + 255) / 256 * 256; }

}  // namespace

#define D2R_TRY(expr)            \
  do {                           \
    if (int rc_ = (expr)) return rc_; \
  } while (0)

extern "C" size_t d2r_encoder_layer_bwd_scratch(int B, int L, int E, int F) {
  // every intermediate gradient has its own buffer (none is reused inside the layer): the weight-gradient GEMMs
  // that read them may still be running on the side stream when the main chain has moved on
  const size_t T = (size_t)B * L, es = 2;
  return 7 * align256(T * E * es) + align256(T * F * es) + align256(T * 3 * E * es) + 2 * align256(d2r_layernorm_bwd_workspace((int64_t)T, E)) +
         align256(T * (size_t)(E / 32) * sizeof(float));  // + D scratch of the long-sequence attention backward (<= E/32 heads)
}

static int check_desc(const d2r_encoder_layer_desc* L, const char* fn) {
  D2R_REQUIRE(L != nullptr, "%s: null descriptor", fn);
  D2R_REQUIRE(d2r_is16(L->dtype), "%s: 16-bit compute dtypes only (the fp32 path runs op by op)", fn);
  D2R_REQUIRE(L->B >= 1 && L->L >= 1 && L->E >= 8 && L->H >= 1 && L->F >= 8 && L->E % L->H == 0, "%s: bad shape", fn);
  D2R_REQUIRE(d2r_mha_supported(L->dtype, L->L, L->L, L->E / L->H), "%s: attention shape unsupported by the fused core", fn);
  D2R_REQUIRE(L->act == D2R_ACT_GELU || L->act == D2R_ACT_QUICK_GELU, "%s: activation must be gelu or quick_gelu", fn);
  D2R_REQUIRE(L->p_attn >= 0.f && L->p_attn < 1.f && L->p_hidden >= 0.f && L->p_hidden < 1.f, "%s: dropout probabilities must be in [0, 1)", fn);
  D2R_REQUIRE(L->w_qkv && L->w_o && L->w_1 && L->w_2 && L->b_qkv && L->b_o && L->b_1 && L->b_2 && L->ln1_g && L->ln1_b &&
                  L->ln2_g && L->ln2_b, "%s: null parameter", fn);
  D2R_REQUIRE(L->x && L->y && L->qkv && L->ctx && L->h1 && L->n1 && L->f_pre && L->f && L->h2 && L->lse && L->mean1 &&
                  L->rstd1 && L->mean2 && L->rstd2, "%s: null activation buffer", fn);
  return D2R_OK;
}

extern "C" int d2r_encoder_layer_fwd(const d2r_encoder_layer_desc* L, void* stream) {
  D2R_TRY(check_desc(L, "d2r_encoder_layer_fwd"));
  const int T = L->B * L->L, E = L->E, F = L->F, dh = E / L->H;
  const int64_t E3 = 3 * (int64_t)E;
  const char* qkv = (const char*)L->qkv;
  const void* attn_in = L->x;
  if (L->pre_ln) {
    D2R_TRY(d2r_layernorm_fwd(L->dtype, L->x, L->ln1_g, L->ln1_b, L->eps, T, E, L->n1, L->mean1, L->rstd1, stream));
    attn_in = L->n1;
  }
  D2R_TRY(linear_fwd(L, T, 3 * E, E, attn_in, L->w_qkv, L->b_qkv, L->qkv, D2R_ACT_NONE, nullptr, nullptr, stream));
  D2R_TRY(d2r_mha_fwd(L->dtype, qkv, E3, L->L * E3, qkv + 2 * E, E3, L->L * E3, qkv + 4 * E, E3, L->L * E3, L->ctx, E,
                      (int64_t)L->L * E, nullptr, 0, 0, L->mask, L->lse, L->B, L->H, L->L, L->L, dh, L->scale, L->p_attn, L->seed_attn,
                      stream));
  // dense output in front of its residual add: out = dropout(dense(.)) + residual.  Without dropout the residual rides in the
  // GEMM epilogue; with it the GEMM writes the dense output and one elementwise pass applies the mask and adds the residual
  // in place (the mask is a function of the element index, so the pass needs no extra buffer).
  const bool drop = L->p_hidden > 0.f;
  const int64_t TE = (int64_t)T * E;
  auto dense_res = [&](int N_, int K_, const void* in, const void* w, const float* b, void* out, const void* res, uint64_t seed) -> int {
    if (!drop) return linear_fwd(L, T, N_, K_, in, w, b, out, D2R_ACT_NONE, res, nullptr, stream);
    if (int rc = linear_fwd(L, T, N_, K_, in, w, b, out, D2R_ACT_NONE, nullptr, nullptr, stream)) return rc;
    return d2r_dropout(L->dtype, out, res, out, TE, L->p_hidden, seed, stream);
  };
  D2R_TRY(dense_res(E, E, L->ctx, L->w_o, L->b_o, L->h1, L->x, L->seed_hidden[0]));
  if (L->pre_ln) {
    D2R_TRY(d2r_layernorm_fwd(L->dtype, L->h1, L->ln2_g, L->ln2_b, L->eps, T, E, L->h2, L->mean2, L->rstd2, stream));
    D2R_TRY(linear_fwd(L, T, F, E, L->h2, L->w_1, L->b_1, L->f, L->act, nullptr, L->f_pre, stream));
    D2R_TRY(dense_res(E, F, L->f, L->w_2, L->b_2, L->y, L->h1, L->seed_hidden[1]));
  } else {
    D2R_TRY(d2r_layernorm_fwd(L->dtype, L->h1, L->ln1_g, L->ln1_b, L->eps, T, E, L->n1, L->mean1, L->rstd1, stream));
    D2R_TRY(linear_fwd(L, T, F, E, L->n1, L->w_1, L->b_1, L->f, L->act, nullptr, L->f_pre, stream));
    D2R_TRY(dense_res(E, F, L->f, L->w_2, L->b_2, L->h2, L->n1, L->seed_hidden[1]));
    D2R_TRY(d2r_layernorm_fwd(L->dtype, L->h2, L->ln2_g, L->ln2_b, L->eps, T, E, L->y, L->mean2, L->rstd2, stream));
  }
  return D2R_OK;
}

extern "C" int d2r_encoder_layer_bwd(d2r_encoder_layer_desc* L, void* stream) {
  D2R_TRY(check_desc(L, "d2r_encoder_layer_bwd"));
  D2R_REQUIRE(L->dy && L->dx && L->gw_qkv && L->gw_o && L->gw_1 && L->gw_2 && L->gb_qkv && L->gb_o && L->gb_1 && L->gb_2 &&
                  L->gln1_g && L->gln1_b && L->gln2_g && L->gln2_b, "d2r_encoder_layer_bwd: null gradient pointer");
  D2R_REQUIRE(L->scratch && L->scratch_bytes >= d2r_encoder_layer_bwd_scratch(L->B, L->L, L->E, L->F) && d2r_aligned16(L->scratch),
              "d2r_encoder_layer_bwd: scratch too small (need d2r_encoder_layer_bwd_scratch bytes, 16-byte aligned)");
  const int T = L->B * L->L, E = L->E, F = L->F, dh = E / L->H;
  const int64_t E3 = 3 * (int64_t)E;
  const size_t es = 2;
  char* p = (char*)L->scratch;
  auto take = [&](size_t elems) { void* r = p; p += align256(elems * es); return r; };
  void* a0 = take((size_t)T * E);  // five [T,E] gradients
  void* a1 = take((size_t)T * E);
  void* a2 = take((size_t)T * E);
  void* a3 = take((size_t)T * E);
  void* a4 = take((size_t)T * E);
  void* a5 = take((size_t)T * E);  // with dropout: the masked copies of the two dense-output gradients
  void* a6 = take((size_t)T * E);
  void* df = take((size_t)T * F);  // d f / d f_pre (in place)
  char* dqkv = (char*)take((size_t)T * 3 * E);
  void* lnws = p;
  const size_t lnws_bytes = d2r_layernorm_bwd_workspace(T, E);
  void* lnws1 = p + align256(lnws_bytes);  // LayerNorm 1's partial sums when their summation is deferred (else both use lnws)
  float* dsum = reinterpret_cast<float*>(p + 2 * align256(lnws_bytes));
  const bool dln = L->defer_wgrad && L->defer_ln;
  float *g1g = dln ? nullptr : L->gln1_g, *g1b = dln ? nullptr : L->gln1_b, *g2g = dln ? nullptr : L->gln2_g, *g2b = dln ? nullptr : L->gln2_b;
  void* ws1 = dln ? lnws1 : lnws;
  L->o_lnws[0] = dln ? lnws1 : nullptr, L->o_lnws[1] = dln ? lnws : nullptr;
  const char* qkv = (const char*)L->qkv;
  // gradient entering a dense layer whose output was dropped: mask(g) / (1-p); the unmasked g still feeds the skip connection
  const bool drop = L->p_hidden > 0.f;
  auto masked = [&](const void* g, void* buf, uint64_t seed, const void** out) -> int {
    *out = g;
    if (!drop) return D2R_OK;
    *out = buf;
    return d2r_dropout(L->dtype, g, nullptr, buf, (int64_t)T * E, L->p_hidden, seed, stream);
  };
  const void *g_ffn = nullptr, *g_att = nullptr;
  if (!L->pre_ln) {
    // y = LN2(h2), h2 = n1 + drop(ffn(n1)), n1 = LN1(h1), h1 = x + drop(attn(x))
    void *d_h2 = a0, *d_n1 = a1, *d_h1 = a2, *d_ctx = a3;
    (void)a4;
    D2R_TRY(d2r_layernorm_bwd_ex(L->dtype, L->dy, L->h2, L->ln2_g, L->mean2, L->rstd2, T, E, d_h2, nullptr, g2g, g2b, 1, lnws, lnws_bytes, stream));
    D2R_TRY(masked(d_h2, a5, L->seed_hidden[1], &g_ffn));
    D2R_TRY(linear_bwd(L, 3, T, E, F, g_ffn, L->f, L->w_2, df, nullptr, L->gw_2, L->gb_2, stream, L->f_pre, L->act));  // df = d f_pre
    D2R_TRY(linear_bwd(L, 2, T, F, E, df, L->n1, L->w_1, d_n1, d_h2, L->gw_1, L->gb_1, stream));  // ffn path + skip
    D2R_TRY(d2r_layernorm_bwd_ex(L->dtype, d_n1, L->h1, L->ln1_g, L->mean1, L->rstd1, T, E, d_h1, nullptr, g1g, g1b, 1, ws1, lnws_bytes, stream));
    D2R_TRY(masked(d_h1, a6, L->seed_hidden[0], &g_att));
    D2R_TRY(linear_bwd(L, 1, T, E, E, g_att, L->ctx, L->w_o, d_ctx, nullptr, L->gw_o, L->gb_o, stream));
    D2R_TRY(d2r_mha_bwd(L->dtype, qkv, E3, L->L * E3, qkv + 2 * E, E3, L->L * E3, qkv + 4 * E, E3, L->L * E3, d_ctx, E, (int64_t)L->L * E,
                        L->mask, L->lse, dsum, dqkv, E3, L->L * E3, dqkv + 2 * E, E3, L->L * E3, dqkv + 4 * E, E3, L->L * E3, L->B, L->H,
                        L->L, L->L, dh, L->scale, L->p_attn, L->seed_attn, stream));
    D2R_TRY(linear_bwd(L, 0, T, 3 * E, E, dqkv, L->x, L->w_qkv, L->dx, d_h1, L->gw_qkv, L->gb_qkv, stream));  // + skip
  } else {
    // y = h1 + ffn(h2), h2 = LN2(h1), h1 = x + attn(n1), n1 = LN1(x)
    void *d_h2 = a0, *d_h1 = a1, *d_ctx = a2, *d_n1 = a3;
    (void)a4;
    D2R_TRY(masked(L->dy, a5, L->seed_hidden[1], &g_ffn));
    D2R_TRY(linear_bwd(L, 3, T, E, F, g_ffn, L->f, L->w_2, df, nullptr, L->gw_2, L->gb_2, stream, L->f_pre, L->act));  // df = d f_pre
    D2R_TRY(linear_bwd(L, 2, T, F, E, df, L->h2, L->w_1, d_h2, nullptr, L->gw_1, L->gb_1, stream));
    D2R_TRY(d2r_layernorm_bwd_ex(L->dtype, d_h2, L->h1, L->ln2_g, L->mean2, L->rstd2, T, E, d_h1, L->dy, g2g, g2b, 1, lnws, lnws_bytes, stream));  // + skip
    D2R_TRY(masked(d_h1, a6, L->seed_hidden[0], &g_att));
    D2R_TRY(linear_bwd(L, 1, T, E, E, g_att, L->ctx, L->w_o, d_ctx, nullptr, L->gw_o, L->gb_o, stream));
    D2R_TRY(d2r_mha_bwd(L->dtype, qkv, E3, L->L * E3, qkv + 2 * E, E3, L->L * E3, qkv + 4 * E, E3, L->L * E3, d_ctx, E, (int64_t)L->L * E,
                        L->mask, L->lse, dsum, dqkv, E3, L->L * E3, dqkv + 2 * E, E3, L->L * E3, dqkv + 4 * E, E3, L->L * E3, L->B, L->H,
                        L->L, L->L, dh, L->scale, L->p_attn, L->seed_attn, stream));
    D2R_TRY(linear_bwd(L, 0, T, 3 * E, E, dqkv, L->n1, L->w_qkv, d_n1, nullptr, L->gw_qkv, L->gb_qkv, stream));
    D2R_TRY(d2r_layernorm_bwd_ex(L->dtype, d_n1, L->x, L->ln1_g, L->mean1, L->rstd1, T, E, L->dx, d_h1, g1g, g1b, 1, ws1, lnws_bytes, stream));  // + skip
  }
  return D2R_OK;
}
