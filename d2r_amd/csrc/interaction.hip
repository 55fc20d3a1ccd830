// interaction.hip — K16: one whole (Reversed_)InteractionModule per C call, forward or backward (host code only: it
// sequences the kernels of gemm*.hip / attention.hip / routing.hip / rowops.hip / elementwise.hip / misc.hip).
//
// Replaces (reference file:line): InteractionModule.forward models/InteractionModule.py:22-55 (=:75-108) with the routing
// layers models/DynamicInteraction.py:37-69, 90-134 (=:157-189, 210-254), the routers models/Router.py:22-26 and the six
// cells models/Cells.py:30-255 (+ SelfAttention.py:11-70, Refinement.py:86-154, XModules.py:277-394), and their autograd
// backward.  What this file removes is HOST work: ~45 forward / ~110 backward launches per layer are issued from C++
// instead of one Python autograd node and one foreign call each (the step was host-bound: 27 of 30 ms), every
// multi-consumer gradient is accumulated in a GEMM epilogue (beta = 1 / residual operand) instead of a separate add
// launch, and the 768x768-class weight gradients are queued and launched grouped at the end of the call.
//
// Notation: T = B*Lq own-modality rows, S = B*Lk other-modality rows, E = 768, n = Lq + 1 (SAF rows per sample).
#include <vector>

#include "common.h"

namespace {

constexpr int E = 768;
const float XSCALE = 100.0f / sqrtf(768.0f);  // softmax(100 q k^T / sqrt(768)), models/XModules.py:305-309

#define TRY(expr)                     \
  do {                                \
    if (int rc_ = (expr)) return rc_; \
  } while (0)

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct Arena {
  char* base;
  size_t off = 0;
  explicit Arena(void* b) : base((char*)b) {}
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += al(bytes);
    return p;
  }
};

struct Dims {
  int B, Lq, Lk, nc, nl, hid, hidi, heads;
  int T, S, n, lkp, lqp;
  int kvpl, nkv;   // alignment cells (GLAC, CMRC, CRCMC) per layer / in the module: each owns a 1536-column block of the k|v run
  int64_t ldkv;    // row stride of the module-wide k|v projection of `other`: nkv * 1536
  size_t es;
};

// column block of (layer l, alignment cell which: 0 GLAC, 1 CMRC, 2 CRCMC) in the k|v run (layer-major, cells in this order)
int kv_block(const Dims& d, int l, int which) { return l * d.kvpl + (which == 0 ? 0 : which == 1 ? (d.nc > 1) : (d.nc > 1) + (d.nc > 3)); }

Dims make_dims(int B, int Lq, int Lk, int nc, int nl, int hid, int hidi, int heads) {
  Dims d;
  d.B = B, d.Lq = Lq, d.Lk = Lk, d.nc = nc, d.nl = nl, d.hid = hid, d.hidi = hidi, d.heads = heads;
  d.T = B * Lq, d.S = B * Lk, d.n = Lq + 1, d.lkp = (Lk + 7) / 8 * 8, d.lqp = (Lq + 7) / 8 * 8;
  d.kvpl = (nc > 1) + (nc > 3) + (nc > 4), d.nkv = d.kvpl * nl, d.ldkv = (int64_t)d.nkv * 2 * E;
  d.es = 2;
  return d;
}

// ---- saved forward activations of one layer ---------------------------------------------------------------
struct LayerF {
  float *pooled, *h, *gates;
  void *qkv, *y, *f1, *e2;
  float* lse_i;
  void *g_q, *g_kv, *g_c, *g_sq, *g_loc, *g_l2, *g_sl, *g_pt, *g_pi, *g_dg, *g_glo, *g_l2g, *g_sg, *g_S, *g_w16, *g_wsum, *e1;
  float *g_lse, *g_nloc, *g_nglo, *g_a, *g_w, *g_saved, *g_ne1;
  void *c_q, *c_kv, *c_c, *c_s, *c_h, *c_mod, *c_f, *e3;
  float* c_lse;
  void *r_q, *r_kv, *r_c, *r_Qs, *r_Ks, *r_a, *r_b, *e4;
  float *r_lse, *r_lse2;
  void *s_a, *s_b, *s_ab, *s_z1, *s_z, *s_g, *e5;
  void* outs[6];
};

// kvall: the module-wide k|v projection [S, nkv*1536] (taken from the arena before the first layer); l: layer index
void plan_fwd(Arena& A, const Dims& d, int P, bool first, bool final, LayerF& L, char* kvall, int l) {
  const size_t TE = (size_t)d.T * E * d.es, BE = (size_t)d.B * E * d.es;
  memset(&L, 0, sizeof(L));
  L.pooled = (float*)A.take((size_t)(first ? 1 : d.nc) * d.B * E * 4);
  L.h = (float*)A.take((size_t)d.B * d.nc * d.hid * 4);
  L.gates = (float*)A.take((size_t)d.B * d.nc * P * 4);
  if (d.nc > 1) {  // GLAC
    L.g_q = A.take(TE), L.g_kv = kvall ? kvall + (size_t)kv_block(d, l, 0) * 2 * E * d.es : nullptr, L.g_c = A.take(TE), L.g_sq = A.take(TE), L.g_loc = A.take(TE);
    L.g_l2 = A.take(TE), L.g_sl = A.take(TE);
    L.g_pt = A.take(BE), L.g_pi = A.take(BE), L.g_dg = A.take(BE), L.g_glo = A.take(BE), L.g_l2g = A.take(BE), L.g_sg = A.take(BE);
    L.g_S = A.take((size_t)d.B * d.n * E * d.es);
    L.g_w16 = A.take((size_t)d.B * d.n * d.es + 16), L.g_wsum = A.take(BE), L.e1 = A.take(BE);
    L.g_lse = (float*)A.take((size_t)d.T * 4), L.g_nloc = (float*)A.take((size_t)d.T * 4), L.g_nglo = (float*)A.take((size_t)d.B * 4);
    L.g_a = (float*)A.take((size_t)d.B * d.n * 4), L.g_w = (float*)A.take((size_t)d.B * d.n * 4);
    L.g_saved = (float*)A.take(16), L.g_ne1 = (float*)A.take((size_t)d.B * 4);
  }
  if (d.nc > 2) {  // IMRC
    L.qkv = A.take(3 * TE), L.y = A.take(TE), L.f1 = A.take((size_t)d.T * d.hidi * d.es), L.e2 = A.take(TE);
    L.lse_i = (float*)A.take((size_t)d.B * 64 * d.Lq * 4);  // up to 64 heads (the size queries do not know the head count)
  }
  if (d.nc > 3) {  // CMRC
    L.c_q = A.take(TE), L.c_kv = kvall ? kvall + (size_t)kv_block(d, l, 1) * 2 * E * d.es : nullptr, L.c_c = A.take(TE), L.c_s = A.take(TE), L.c_h = A.take(TE);
    L.c_mod = A.take(TE), L.c_f = A.take(TE), L.e3 = A.take(TE);
    L.c_lse = (float*)A.take((size_t)d.T * 4);
  }
  if (d.nc > 4) {  // CRCMC
    L.r_q = A.take(TE), L.r_kv = kvall ? kvall + (size_t)kv_block(d, l, 2) * 2 * E * d.es : nullptr, L.r_c = A.take(TE), L.r_Qs = A.take(TE), L.r_Ks = A.take(TE);
    L.r_a = A.take(TE), L.r_b = A.take(TE), L.e4 = A.take(TE);
    L.r_lse = (float*)A.take((size_t)d.T * 4), L.r_lse2 = (float*)A.take((size_t)d.T * 4);
  }
  if (d.nc > 5) {  // GESC
    L.s_a = A.take(BE), L.s_b = A.take(BE), L.s_ab = A.take(BE), L.s_z1 = A.take(BE), L.s_z = A.take(BE), L.s_g = A.take(BE);
    L.e5 = A.take(BE);
  }
  if (!final)
    for (int i = 0; i < P; ++i) L.outs[i] = A.take(TE);
}

// ---- launch helpers ------------------------------------------------------------------------------------------
// global-batch-exact BatchNorm (d2r_interaction_desc.bn_sync): `buf` = this layer's four doubles, `ntotal` = scores of the global batch
struct BnSync {
  int (*fn)(void*, double*, void*) = nullptr;
  void* user = nullptr;
  double* buf = nullptr;
  double ntotal = 0.0;
};

struct Ctx {
  int dt;
  void* st;
  void* ws;
  size_t wsb;
  BnSync bn = {};
};

struct G {
  d2r_gemm_desc d;
  G(int dtype, int cdtype, int layout, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
    int64_t ldc) {
    memset(&d, 0, sizeof(d));
    d.dtype = dtype, d.c_dtype = cdtype, d.layout = layout, d.act = D2R_ACT_NONE;
    d.M = M, d.N = N, d.K = K, d.nb = 1, d.nh = 1, d.alpha = 1.f, d.beta = 0.f;
    d.A = A, d.lda = lda, d.B = B, d.ldb = ldb, d.C = C, d.ldc = ldc;
  }
  G& batch(int nb, int64_t sA, int64_t sB, int64_t sC) {
    d.nb = nb, d.sAb = sA, d.sBb = sB, d.sCb = sC;
    return *this;
  }
  G& ws(const Ctx& c) {
    d.workspace = c.ws, d.workspace_bytes = c.wsb;
    return *this;
  }
};

// y[M,N] = act(x[M,K] W^T + b) (+ residual); x rows at stride ldx; the split-K scratch goes with M <= 64 (pooled vectors)
int lin(const Ctx& c, int M, int N, int K, const void* x, int64_t ldx, const d2r_linear_params& p, void* y, int act = D2R_ACT_NONE,
        const void* res = nullptr, int dt = -1, int cdt = -1) {
  if (dt < 0) dt = c.dt;
  if (cdt < 0) cdt = dt;
  G g(dt, cdt, D2R_GEMM_NT, M, N, K, x, ldx, p.w, K, y, N);
  g.d.bias = p.b, g.d.act = act, g.d.residual = res, g.d.ldr = N;
  if (M <= 64 || K >= 6144) g.ws(c);  // (pooled vectors: split-K slabs; very deep reductions: split K inside the launch)
  return d2r_gemm(&g.d, c.st);
}

// dx[M,Kf] (ldc) = beta*dx + dy[M,Nf] (ldy) W[Nf,Kf] (+ residual [M,Kf]) (* act'(grad_ref))
int dxg(const Ctx& c, int M, int Kf, int Nf, const void* dy, int64_t ldy, const void* w, void* dx, int64_t ldc, float beta = 0.f,
        const void* res = nullptr, const void* gref = nullptr, int gact = D2R_ACT_NONE, int dt = -1) {
  if (dt < 0) dt = c.dt;
  G g(dt, dt, D2R_GEMM_NN, M, Kf, Nf, dy, ldy, w, Kf, dx, ldc);
  g.d.beta = beta, g.d.residual = res, g.d.ldr = Kf, g.d.grad_ref = gref, g.d.grad_act = gact;
  if (M <= 64 || Nf >= 6144) g.ws(c);  // (pooled vectors: split-K slabs; the d_other product, K = 13,824: split K inside the launch)
  return d2r_gemm(&g.d, c.st);
}

// weight gradient gw[Nf,Kf] += dy[M,Nf]^T x[M,Kf] (+ gb += colsum dy), launched on the spot (split-K inside)
int dwg(const Ctx& c, int M, int Nf, int Kf, const void* dy, int64_t ldy, const void* x, int64_t ldx, const d2r_linear_params& p,
        int dt = -1) {
  if (dt < 0) dt = c.dt;
  G g(dt, D2R_F32, D2R_GEMM_TN, Nf, Kf, M, dy, ldy, x, ldx, p.gw, Kf);
  g.d.beta = 1.f, g.d.dbias = p.gb;
  g.ws(c);
  return d2r_gemm(&g.d, c.st);
}

// ---- grouped launches of independent products (d2r_gemm_group) -----------------------------------------------------------------------
// The cells of a routing layer are independent between the layer's inputs and its aggregation, and their token-row linears are
// 768-wide products over 4-6 thousand rows: 192-300 tiles each, one partial round of workgroups (13-15 us for work worth 5).  The
// products of DIFFERENT cells that are ready at the same point of the layer leave as one grouped launch (a thousand tiles: the
// same kernel at its saturated rate); every tile is computed exactly as in a launch of its own, so the results are bit-identical to
// the op-by-op schedule.  A Group must only hold products that neither read nor write each other's outputs.
struct Group {
  std::vector<d2r_gemm_desc> v;
  void lin(const Ctx& c, int M, int N, int K, const void* x, int64_t ldx, const d2r_linear_params& p, void* y, int act = D2R_ACT_NONE,
           const void* res = nullptr) {
    G g(c.dt, c.dt, D2R_GEMM_NT, M, N, K, x, ldx, p.w, K, y, N);
    g.d.bias = p.b, g.d.act = act, g.d.residual = res, g.d.ldr = N;
    v.push_back(g.d);
  }
  void dxg(const Ctx& c, int M, int Kf, int Nf, const void* dy, int64_t ldy, const void* w, void* dx, int64_t ldc, float beta = 0.f,
           const void* res = nullptr, const void* gref = nullptr, int gact = D2R_ACT_NONE) {
    G g(c.dt, c.dt, D2R_GEMM_NN, M, Kf, Nf, dy, ldy, w, Kf, dx, ldc);
    g.d.beta = beta, g.d.residual = res, g.d.ldr = Kf, g.d.grad_ref = gref, g.d.grad_act = gact;
    v.push_back(g.d);
  }
  int flush(const Ctx& c) {
    const int rc = v.empty() ? D2R_OK : d2r_gemm_group(v.data(), (int)v.size(), c.st);
    v.clear();
    return rc;
  }
};

// queued weight gradients (768x768 / 1536x768 class): nothing in the backward pass reads them
struct WJob {
  int M, Nf, Kf;
  int64_t ldy, ldx;
  const void *dy, *x;
  float *gw, *gb;
};
typedef std::vector<WJob> Jobs;

void defer(Jobs& jobs, int M, int Nf, int Kf, const void* dy, int64_t ldy, const void* x, int64_t ldx, const d2r_linear_params& p) {
  jobs.push_back(WJob{M, Nf, Kf, ldy, ldx, dy, x, p.gw, p.gb});
}

// All queued products of the call in ONE d2r_gemm_tn_grouped_v (different shapes share launches on the 256-wide kernel); a sink that
// was queued twice (one parameter used at two sites) waits for the next round: the problems of a call run in parallel.
int flush_jobs(const Ctx& c, Jobs& jobs) {
  while (!jobs.empty()) {
    std::vector<int> M, N, K;
    std::vector<int64_t> lda, ldb, ldc;
    std::vector<const void*> A, Bp;
    std::vector<float*> Cp, Dp;
    Jobs rest;
    for (const WJob& b : jobs) {
      bool dup = false;
      for (size_t k = 0; k < Cp.size(); ++k) dup |= (Cp[k] == b.gw) || (b.gb && Dp[k] == b.gb);
      if (dup) {
        rest.push_back(b);
        continue;
      }
      M.push_back(b.Nf), N.push_back(b.Kf), K.push_back(b.M), lda.push_back(b.ldy), ldb.push_back(b.ldx), ldc.push_back(b.Kf);
      A.push_back(b.dy), Bp.push_back(b.x), Cp.push_back(b.gw), Dp.push_back(b.gb);
    }
    TRY(d2r_gemm_tn_grouped_v(c.dt, (int)A.size(), M.data(), N.data(), K.data(), lda.data(), ldb.data(), ldc.data(), A.data(), Bp.data(), Cp.data(),
                              Dp.data(), 1.f, c.st));
    jobs.swap(rest);
  }
  return D2R_OK;
}

int copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, void* st) {
  return d2r_copy_rows(dst, (int64_t)dpitch, src, (int64_t)spitch, (int64_t)width, (int64_t)height, st);
}

// single-head attention over the 768-wide feature: forward
int xat_fwd(const Ctx& c, const Dims& d, const void* q, const void* k, int64_t ldk, const void* v, int64_t ldv, int Lk, void* o,
            const void* res, float* lse, float scale) {
  return d2r_xattn_fwd(c.dt, q, E, (int64_t)d.Lq * E, k, ldk, (int64_t)Lk * ldk, v, ldv, (int64_t)Lk * ldv, o, E, (int64_t)d.Lq * E, res,
                       E, (int64_t)d.Lq * E, nullptr, lse, d.B, d.Lq, Lk, E, scale, c.st);
}
// backward: dq [T,E]; dk / dv written at (ptr, ld) with batch stride Lk*ld; P / dS: [B,Lq,lkp] scratch
int xat_bwd(const Ctx& c, const Dims& d, const void* q, const void* k, int64_t ldk, const void* v, int64_t ldv, int Lk, const void* dO,
            const void* o, const void* res, const float* lse, void* dq, void* dk, int64_t lddk, void* dv, int64_t lddv, void* P, void* dS, float scale) {
  const int lkp = (Lk + 7) / 8 * 8;
  const void *qa[1] = {q}, *ka[1] = {k}, *va[1] = {v}, *ga[1] = {dO}, *oa[1] = {o}, *ra[1] = {res};
  const float* la[1] = {lse};
  void *dqa[1] = {dq}, *dka[1] = {dk}, *dva[1] = {dv}, *pa[1] = {P}, *dsa[1] = {dS};
  return d2r_xattn_bwd_multi(c.dt, 1, qa, E, (int64_t)d.Lq * E, ka, ldk, (int64_t)Lk * ldk, va, ldv, (int64_t)Lk * ldv, ga, E, (int64_t)d.Lq * E,
                             oa, E, (int64_t)d.Lq * E, res ? ra : nullptr, E, (int64_t)d.Lq * E, nullptr, la, dqa, E, (int64_t)d.Lq * E, dka, lddk, (int64_t)Lk * lddk, dva, lddv, (int64_t)Lk * lddv, pa, dsa, lkp,
                             d.B, d.Lq, Lk, E, scale, c.st);
}

// ---- forward of one routing layer --------------------------------------------------------------------------
// Schedule (one stream): routers and the per-sample vector chains (GLAC's global branch, GESC) as before; the token-row products in
// four grouped launches A - D between the attention cores and the element-wise steps that separate them.
int layer_fwd(const Ctx& c, const Dims& d, const d2r_routing_layer_params& p, int P, bool first, bool final, int train,
              const void* const* refs, const void* other, LayerF& L, void* out_final, float* probs, int64_t ldp) {
  const int B = d.B, T = d.T, nc = d.nc, hid = d.hid, n = d.n;
  const int64_t TEe = (int64_t)d.Lq * E, SEe = (int64_t)d.Lk * E;
  const d2r_linear_params* lp = p.lin;
  Group grp;
  // --- group A: everything that reads only the layer's inputs: IMRC's q|k|v, the query projections of the three alignment cells,
  //     CRCMC's second input branch Ks = tanh(W x) ---------------------------------------------------------------------------------
  if (nc > 2) grp.lin(c, T, 3 * E, E, refs[2], E, lp[D2R_RL_IMRC_QKV], L.qkv);
  if (nc > 1) grp.lin(c, T, E, E, refs[1], E, lp[D2R_RL_GLAC_Q], L.g_q);
  if (nc > 3) grp.lin(c, T, E, E, refs[3], E, lp[D2R_RL_CMRC_Q], L.c_q);
  if (nc > 4) grp.lin(c, T, E, E, refs[4], E, lp[D2R_RL_CRCMC_Q], L.r_q);
  if (nc > 4) grp.lin(c, T, E, E, refs[4], E, lp[D2R_RL_CRCMC_MLP2], L.r_Ks, D2R_ACT_TANH);
  TRY(grp.flush(c));
  // --- routers (fp32 end to end: routing decisions are exact) ---------------------------------------------------------------------
  {
    if (first) {  // the cells of layer 0 read the same tensor: pool once, ONE plain GEMM with N = ncell*hid
      TRY(d2r_meanpool_fwd(c.dt, refs, 1, B, d.Lq, E, L.pooled, c.st));
      TRY(lin(c, B, nc * hid, E, L.pooled, E, lp[D2R_RL_R0], L.h, D2R_ACT_RELU, nullptr, D2R_F32));
    } else {
      TRY(d2r_meanpool_fwd(c.dt, refs, nc, B, d.Lq, E, L.pooled, c.st));
      G g(D2R_F32, D2R_F32, D2R_GEMM_NT, B, hid, E, L.pooled, E, lp[D2R_RL_R0].w, E, L.h, (int64_t)nc * hid);
      g.batch(nc, (int64_t)B * E, (int64_t)hid * E, hid);
      g.d.bias = lp[D2R_RL_R0].b, g.d.s_bias_b = hid, g.d.act = D2R_ACT_RELU;
      TRY(d2r_gemm(&g.d, c.st));
    }
    G g(D2R_F32, D2R_F32, D2R_GEMM_NT, B, P, hid, L.h, (int64_t)nc * hid, lp[D2R_RL_R2].w, hid, L.gates, (int64_t)nc * P);
    g.batch(nc, hid, (int64_t)P * hid, P);
    g.d.bias = lp[D2R_RL_R2].b, g.d.s_bias_b = P, g.d.act = D2R_ACT_TANH_RELU;
    TRY(d2r_gemm(&g.d, c.st));
  }
  // --- IMRC (cell 2): 16-head attention over its own q|k|v -------------------------------------------------------------------------
  if (nc > 2) {
    const int dh = E / d.heads;
    const char* qkv = (const char*)L.qkv;
    TRY(d2r_mha_fwd(c.dt, qkv, 3 * E, (int64_t)d.Lq * 3 * E, qkv + E * d.es, 3 * E, (int64_t)d.Lq * 3 * E, qkv + 2 * E * d.es, 3 * E,
                    (int64_t)d.Lq * 3 * E, L.y, E, TEe, refs[2], E, TEe, nullptr, L.lse_i, B, d.heads, d.Lq, d.Lq, dh, 1.0f / sqrtf((float)dh), 0.f, 0, c.st));
  }
  // --- the alignment cores of GLAC / CMRC / CRCMC: ONE attention launch for all of them (same shapes, same
  //     softmax(100 q k^T / sqrt(768)) v).  Keys and values are column blocks of the module-wide k|v projection of `other` ---------
  {
    const void *qa[3], *ka[3], *va[3];
    void* oa[3];
    float* la[3];
    int ncore = 0;
    auto core = [&](void* qbuf, const void* kv, void* o, float* lse) {
      qa[ncore] = qbuf, ka[ncore] = kv, va[ncore] = (const char*)kv + E * d.es, oa[ncore] = o, la[ncore] = lse;
      ++ncore;
    };
    if (nc > 1) core(L.g_q, L.g_kv, L.g_c, L.g_lse);
    if (nc > 3) core(L.c_q, L.c_kv, L.c_c, L.c_lse);
    if (nc > 4) core(L.r_q, L.r_kv, L.r_c, L.r_lse);
    if (ncore)
      TRY(d2r_xattn_fwd_multi(c.dt, ncore, qa, E, TEe, ka, d.ldkv, (int64_t)d.Lk * d.ldkv, va, d.ldkv, (int64_t)d.Lk * d.ldkv, oa, E, TEe, nullptr, E,
                              TEe, nullptr, la, B, d.Lq, d.Lk, E, XSCALE, c.st));
  }
  // --- the per-sample vector chains: GLAC's (cell 1) global branch and GESC (cell 5).  Both read only the layer's inputs and are independent
  //     of each other: their linears (32 rows: latency, not work) go in lock step, one grouped launch per stage (d2r_gemm_group) --------
  {
    Group pg;
    if (nc > 1) pg.lin(c, B, E, E, refs[1], TEe, lp[D2R_RL_GLAC_TPOOL], L.g_pt, D2R_ACT_TANH);
    if (nc > 1) pg.lin(c, B, E, E, other, SEe, lp[D2R_RL_GLAC_IPOOL], L.g_pi, D2R_ACT_TANH);
    if (nc > 5) pg.lin(c, B, E, E, refs[5], TEe, lp[D2R_RL_GESC_TPOOL], L.s_a, D2R_ACT_TANH);
    if (nc > 5) pg.lin(c, B, E, E, other, SEe, lp[D2R_RL_GESC_IPOOL], L.s_b, D2R_ACT_TANH);
    TRY(pg.flush(c));
    if (nc > 1) TRY(d2r_sqdiff_fwd(c.dt, L.g_pt, L.g_pi, L.g_dg, (int64_t)B * E, c.st));
    if (nc > 5) TRY(d2r_add(c.dt, L.s_a, L.s_b, L.s_ab, (int64_t)B * E, c.st));
    if (nc > 1) pg.lin(c, B, E, E, L.g_dg, E, lp[D2R_RL_GLAC_GLO], L.g_glo);
    if (nc > 5) pg.lin(c, B, E, E, L.s_ab, E, lp[D2R_RL_GESC_MLP0], L.s_z1, D2R_ACT_TANH);
    TRY(pg.flush(c));
    if (nc > 1) TRY(d2r_l2norm_fwd(c.dt, L.g_glo, L.g_l2g, L.g_nglo, B, E, c.st));
    if (nc > 1) pg.lin(c, B, E, E, L.g_l2g, E, lp[D2R_RL_GLAC_FC2], L.g_sg);
    if (nc > 5) pg.lin(c, B, E, E, L.s_z1, E, lp[D2R_RL_GESC_MLP2], L.s_z);
    TRY(pg.flush(c));
    if (nc > 5) {
      TRY(d2r_softmax_fwd(c.dt, c.dt, L.s_z, L.s_g, E, B, E, 1.0f, nullptr, 1, c.st));
      TRY(d2r_lerp_fwd(c.dt, L.s_g, L.s_a, L.s_b, L.e5, (int64_t)B * E, c.st));
    }
    if (nc > 1) TRY(d2r_sqdiff_fwd(c.dt, refs[1], L.g_c, L.g_sq, (int64_t)T * E, c.st));  // GLAC's local branch: (t - c)^2 behind the core
  }
  // --- group B: the first linears behind the cores and behind IMRC's attention ---------------------------------------------------------
  if (nc > 2) grp.lin(c, T, d.hidi, E, L.y, E, lp[D2R_RL_IMRC_FC1], L.f1, D2R_ACT_RELU);
  if (nc > 1) grp.lin(c, T, E, E, L.g_sq, E, lp[D2R_RL_GLAC_LOC], L.g_loc);
  if (nc > 3) grp.lin(c, T, E, E, L.c_c, E, lp[D2R_RL_CMRC_SCALE], L.c_s, D2R_ACT_TANH);
  if (nc > 3) grp.lin(c, T, E, E, L.c_c, E, lp[D2R_RL_CMRC_SHIFT], L.c_h);
  if (nc > 4) grp.lin(c, T, E, E, L.r_c, E, lp[D2R_RL_CRCMC_MLP1], L.r_Qs, D2R_ACT_TANH);
  if (nc > 4) grp.lin(c, T, E, E, L.r_Ks, E, lp[D2R_RL_CRCMC_FC2], L.r_b);
  TRY(grp.flush(c));
  if (nc > 1) TRY(d2r_l2norm_fwd(c.dt, L.g_loc, L.g_l2, L.g_nloc, T, E, c.st));
  if (nc > 3) TRY(d2r_muladd_fwd(c.dt, refs[3], L.c_s, L.c_h, L.c_mod, (int64_t)T * E, c.st));
  // --- group C: the second linears -------------------------------------------------------------------------------------------------------
  if (nc > 2) grp.lin(c, T, E, d.hidi, L.f1, d.hidi, lp[D2R_RL_IMRC_FC2], L.e2, D2R_ACT_NONE, L.y);
  if (nc > 1) grp.lin(c, T, E, E, L.g_l2, E, lp[D2R_RL_GLAC_FC1], L.g_sl);
  if (nc > 3) grp.lin(c, T, E, E, L.c_mod, E, lp[D2R_RL_CMRC_FC1], L.c_f, D2R_ACT_RELU);
  if (nc > 4) grp.lin(c, T, E, E, L.r_Qs, E, lp[D2R_RL_CRCMC_FC1], L.r_a);
  TRY(grp.flush(c));
  // --- CRCMC (cell 4): its second core, unscaled softmax(Q K^T) over its own tokens, + Qs ------------------------------------------------
  if (nc > 4) TRY(xat_fwd(c, d, L.r_a, L.r_b, E, L.r_Ks, E, d.Lq, L.e4, L.r_Qs, L.r_lse2, 1.0f));
  // --- CMRC (cell 3): last linear + skip ---------------------------------------------------------------------------------------------------
  if (nc > 3) TRY(lin(c, T, E, E, L.c_f, E, lp[D2R_RL_CMRC_FC2], L.e3, D2R_ACT_NONE, refs[3]));
  // --- GLAC: the SAF gate over [global | local] ----------------------------------------------------------------------------------------------
  if (nc > 1) {
    // S = cat([sg[:,None], sl], 1)  [B, n, E]
    const size_t row = (size_t)E * d.es;
    TRY(copy2d(L.g_S, (size_t)n * row, L.g_sg, row, row, B, c.st));
    TRY(copy2d((char*)L.g_S + row, (size_t)n * row, L.g_sl, (size_t)d.Lq * row, (size_t)d.Lq * row, B, c.st));
    TRY(lin(c, B * n, 1, E, L.g_S, E, lp[D2R_RL_GLAC_SAFW], L.g_a, D2R_ACT_NONE, nullptr, -1, D2R_F32));
    if (c.bn.fn && train) {  // statistics over the samples of every rank: local sums, the caller's all-reduce, then the gate
      TRY(d2r_saf_gate_stats(L.g_a, B, n, c.bn.buf, c.st));
      if (c.bn.fn(c.bn.user, c.bn.buf, c.st)) return d2r_fail(D2R_ERR_INVALID, "d2r_interaction_fwd: bn_sync failed");
      TRY(d2r_saf_gate_fwd_ex(L.g_a, B, n, p.bn_weight, p.bn_bias, p.bn_running_mean, p.bn_running_var, train, L.g_w, L.g_saved, c.bn.buf, c.bn.ntotal,
                              L.g_w16, c.dt, c.st));
    } else {  // (the gate also leaves the 16-bit copy of w that the weighted sum below reads as a GEMM operand)
      TRY(d2r_saf_gate_fwd_ex(L.g_a, B, n, p.bn_weight, p.bn_bias, p.bn_running_mean, p.bn_running_var, train, L.g_w, L.g_saved, nullptr, 0.0, L.g_w16,
                              c.dt, c.st));
    }
    G g(c.dt, c.dt, D2R_GEMM_NN, 1, E, n, L.g_w16, n, L.g_S, E, L.g_wsum, E);  // wsum[b] = w[b] @ S[b]
    g.batch(B, n, (int64_t)n * E, E);
    TRY(d2r_gemm(&g.d, c.st));
    TRY(d2r_l2norm_fwd(c.dt, L.g_wsum, L.e1, L.g_ne1, B, E, c.st));
  }
  // --- K8: path normalisation, gates, aggregation ------------------------------------------------------------------
  const void* embs[6] = {refs[0], L.e1, L.e2, L.e3, L.e4, L.e5};
  void* outs[6];
  for (int i = 0; i < 6; ++i) outs[i] = final ? (i == 0 ? out_final : nullptr) : L.outs[i];
  return d2r_route_aggregate_fwd(c.dt, embs, final ? refs : nullptr, L.gates, B, d.Lq, E, nc, P, outs, probs, ldp, c.st);
}

// ---- backward scratch of one layer -----------------------------------------------------------------------------
struct LayerB {
  float *dgates, *dG, *dh, *dhp, *dpooled, *cs0, *cs2, *bn2;
  void *cws;
  size_t cws_bytes;
  void *agg_ws;
  size_t agg_bytes;
  void* de[6];
  void* dx[6];
  void *i_df1, *i_dy, *i_dqkv;
  float* i_dsum;  // D = rowsum(P o dP) scratch of the long-sequence attention backward
  void *g_dwsum, *g_dw16, *g_dS, *g_dsl, *g_dl2, *g_dloc, *g_dsq, *g_da, *g_dc, *g_dq, *g_P, *g_dSa, *g_dkv, *g_dl2g, *g_dglo, *g_ddg,
      *g_dpt, *g_dpi, *g_dptp, *g_dpip, *g_da16;
  float *g_dwf, *g_daf;
  void *c_dfp, *c_dmod, *c_da, *c_ds, *c_dsp, *c_dc, *c_dq, *c_P, *c_dSa, *c_dkv, *c_tmp;
  void *r_da, *r_P2, *r_dS2, *r_dKv, *r_db, *r_dQs, *r_dKs, *r_dQsp, *r_dKsp, *r_dc, *r_dq, *r_P, *r_dSa, *r_dkv;
  void *s_dg, *s_da1, *s_db1, *s_dz, *s_dz1p, *s_dab, *s_dat, *s_dbt, *s_dap, *s_dbp;
};

// dkvall: gradient of the module-wide k|v projection [S, nkv*1536] (taken from the scratch before the first layer)
void plan_bwd(Arena& A, const Dims& d, int P, bool first, bool final, LayerB& K, char* dkvall, int l) {
  const size_t TE = (size_t)d.T * E * d.es, BE = (size_t)d.B * E * d.es;
  auto dkv = [&](int which) -> void* { return dkvall ? dkvall + (size_t)kv_block(d, l, which) * 2 * E * d.es : nullptr; };
  memset(&K, 0, sizeof(K));
  K.dgates = (float*)A.take((size_t)d.B * d.nc * P * 4), K.dG = (float*)A.take((size_t)d.B * d.nc * P * 4);
  K.dh = (float*)A.take((size_t)d.B * d.nc * d.hid * 4), K.dhp = (float*)A.take((size_t)d.B * d.nc * d.hid * 4);
  K.dpooled = (float*)A.take((size_t)(first ? 1 : d.nc) * d.B * E * 4);
  K.cs0 = (float*)A.take((size_t)d.nc * d.hid * 4), K.cs2 = (float*)A.take((size_t)d.nc * P * 4), K.bn2 = (float*)A.take(16);
  K.cws_bytes = d2r_colsum_workspace(d.B, d.nc * d.hid) + d2r_colsum_workspace(d.B, d.nc * P) + 256;
  K.cws = A.take(K.cws_bytes);
  K.agg_bytes = d2r_route_aggregate_bwd_workspace(d.B, d.Lq, E, P);
  K.agg_ws = A.take(K.agg_bytes);
  K.de[1] = A.take(BE), K.de[5] = A.take(BE);
  for (int j : {2, 3, 4}) K.de[j] = A.take(TE);
  if (first) K.de[0] = nullptr;  // written straight into d_own
  if (!first)
    for (int j = 0; j < d.nc; ++j) K.dx[j] = A.take(TE);  // gradients w.r.t. this layer's six inputs
  if (final) K.de[0] = A.take(TE);  // final layer: relu-path gradient of x0 (the skip-path gradient goes to dx[0])
  if (d.nc > 2) K.i_df1 = A.take((size_t)d.T * d.hidi * d.es), K.i_dy = A.take(TE), K.i_dqkv = A.take(3 * TE);
  if (d.nc > 2) K.i_dsum = (float*)A.take((size_t)d.B * 64 * d.Lq * 4);
  if (d.nc > 1) {
    K.g_dwsum = A.take(BE), K.g_dw16 = A.take((size_t)d.B * d.n * d.es + 16), K.g_dS = A.take((size_t)d.B * d.n * E * d.es);
    K.g_dsl = A.take(TE), K.g_dl2 = A.take(TE), K.g_dloc = A.take(TE), K.g_dsq = A.take(TE), K.g_da = A.take(TE), K.g_dc = A.take(TE);
    K.g_dq = A.take(TE), K.g_P = A.take((size_t)d.T * d.lkp * d.es), K.g_dSa = A.take((size_t)d.T * d.lkp * d.es), K.g_dkv = dkv(0);
    K.g_dl2g = A.take(BE), K.g_dglo = A.take(BE), K.g_ddg = A.take(BE), K.g_dpt = A.take(BE), K.g_dpi = A.take(BE);
    K.g_dptp = A.take(BE), K.g_dpip = A.take(BE), K.g_da16 = A.take((size_t)d.B * d.n * d.es + 16);
    K.g_dwf = (float*)A.take((size_t)d.B * d.n * 4), K.g_daf = (float*)A.take((size_t)d.B * d.n * 4);
  }
  if (d.nc > 3) {
    K.c_dfp = A.take(TE), K.c_dmod = A.take(TE), K.c_da = A.take(TE), K.c_ds = A.take(TE), K.c_dsp = A.take(TE), K.c_dc = A.take(TE);
    K.c_dq = A.take(TE), K.c_P = A.take((size_t)d.T * d.lkp * d.es), K.c_dSa = A.take((size_t)d.T * d.lkp * d.es), K.c_dkv = dkv(1);
    K.c_tmp = A.take(TE);
  }
  if (d.nc > 4) {
    K.r_da = A.take(TE), K.r_P2 = A.take((size_t)d.T * d.lqp * d.es), K.r_dS2 = A.take((size_t)d.T * d.lqp * d.es), K.r_dKv = A.take(TE);
    K.r_db = A.take(TE), K.r_dQs = A.take(TE), K.r_dKs = A.take(TE), K.r_dQsp = A.take(TE), K.r_dKsp = A.take(TE), K.r_dc = A.take(TE);
    K.r_dq = A.take(TE), K.r_P = A.take((size_t)d.T * d.lkp * d.es), K.r_dSa = A.take((size_t)d.T * d.lkp * d.es), K.r_dkv = dkv(2);
  }
  if (d.nc > 5) {
    K.s_dg = A.take(BE), K.s_da1 = A.take(BE), K.s_db1 = A.take(BE), K.s_dz = A.take(BE), K.s_dz1p = A.take(BE), K.s_dab = A.take(BE);
    K.s_dat = A.take(BE), K.s_dbt = A.take(BE), K.s_dap = A.take(BE), K.s_dbp = A.take(BE);
  }
}

// fp32 sink += fp32 temp
int acc32(const Ctx& c, const float* tmp, float* sink, int64_t nel) { return d2r_axpby(D2R_F32, 1.f, tmp, 1.f, sink, nel, c.st); }

// ---- backward of one routing layer --------------------------------------------------------------------------------
// douts[i]: gradient of output i (NULL = zero; only legal for the final layer's single output);  dx[j]: gradient w.r.t.
// input j, OVERWRITTEN (layer 0: all six alias d_own);  d_other: ACCUMULATED (zeroed by the caller).
// Schedule (one stream): aggregation, routers (their pooled gradient initialises every dx[j]), the per-sample vector chains, then
// the token-row products in grouped launches B1 - B5 between the attention backward kernels and the element-wise steps.  The
// products that ACCUMULATE into an input gradient dx[j] ("tails", beta = 1) are grouped only when the dx[j] are six different
// buffers; in layer 0, where all six alias d_own, they are launched one after the other in a fixed order.
int layer_bwd(const Ctx& c, const Dims& d, const d2r_routing_layer_params& p, int P, bool first, bool final, int train,
              const void* const* refs, const void* other, const LayerF& L, LayerB& K, const void* const* douts, const void* out_final,
              const float* dprobs, int64_t ldp, void* const* dx, void* d_other, Jobs& jobs) {
  const int B = d.B, T = d.T, nc = d.nc, hid = d.hid, n = d.n;
  const int64_t TEe = (int64_t)d.Lq * E, SEe = (int64_t)d.Lk * E, TEn = (int64_t)T * E, BEn = (int64_t)B * E;
  const d2r_linear_params* lp = p.lin;
  Group grp;
  // --- K8 backward ------------------------------------------------------------------------------------------------------------------
  {
    const void* embs[6] = {refs[0], L.e1, L.e2, L.e3, L.e4, L.e5};
    void* dembs[6];
    void* drefs[6];
    for (int j = 0; j < 6; ++j) dembs[j] = K.de[j], drefs[j] = nullptr;
    if (final) {
      for (int j = 0; j < nc; ++j) drefs[j] = dx[j];  // skip-path gradients initialise dx[j]; de[0] holds the relu path
    } else {
      dembs[0] = dx[0];  // the relu-path gradient of x0 initialises dx[0] (layer 0: d_own)
    }
    const void* outs[1] = {out_final};
    TRY(d2r_route_aggregate_bwd(c.dt, embs, final ? refs : nullptr, L.gates, douts, final ? outs : nullptr, dprobs, ldp, B, d.Lq, E, nc, P,
                                dembs, final ? drefs : nullptr, K.dgates, K.agg_ws, K.agg_bytes, c.st));
    if (final) TRY(d2r_axpby(c.dt, 1.f, K.de[0], 1.f, dx[0], TEn, c.st));  // x0 is ref_0: relu path + skip path
  }
  // --- routers ------------------------------------------------------------------------------------------------------------------------
  {
    TRY(d2r_act_bwd(D2R_F32, D2R_ACT_TANH_RELU, K.dgates, L.gates, K.dG, (int64_t)B * nc * P, c.st));
    G gx(D2R_F32, D2R_F32, D2R_GEMM_NN, B, hid, P, K.dG, (int64_t)nc * P, lp[D2R_RL_R2].w, hid, K.dh, (int64_t)nc * hid);
    gx.batch(nc, P, (int64_t)P * hid, hid);
    TRY(d2r_gemm(&gx.d, c.st));
    G gw(D2R_F32, D2R_F32, D2R_GEMM_TN, P, hid, B, K.dG, (int64_t)nc * P, L.h, (int64_t)nc * hid, lp[D2R_RL_R2].gw, hid);
    gw.batch(nc, P, hid, (int64_t)P * hid);
    gw.d.beta = 1.f;
    TRY(d2r_gemm(&gw.d, c.st));
    if (B <= 32) {  // (one row slice: the column sums go straight into the bias gradient)
      TRY(d2r_colsum_add(D2R_F32, K.dG, (int64_t)nc * P, B, nc * P, lp[D2R_RL_R2].gb, K.cws, K.cws_bytes, c.st));
    } else {
      TRY(d2r_colsum(D2R_F32, K.dG, (int64_t)nc * P, B, nc * P, K.cs2, K.cws, K.cws_bytes, c.st));
      TRY(acc32(c, K.cs2, lp[D2R_RL_R2].gb, (int64_t)nc * P));
    }
    TRY(d2r_act_bwd(D2R_F32, D2R_ACT_RELU, K.dh, L.h, K.dhp, (int64_t)B * nc * hid, c.st));
    if (first) {
      TRY(dxg(c, B, E, nc * hid, K.dhp, (int64_t)nc * hid, lp[D2R_RL_R0].w, K.dpooled, E, 0.f, nullptr, nullptr, D2R_ACT_NONE, D2R_F32));
      TRY(dwg(c, B, nc * hid, E, K.dhp, (int64_t)nc * hid, L.pooled, E, lp[D2R_RL_R0], D2R_F32));
      TRY(d2r_meanpool_bwd(c.dt, K.dpooled, B, d.Lq, E, dx[0], 1, c.st));
    } else {
      G g0(D2R_F32, D2R_F32, D2R_GEMM_NN, B, E, hid, K.dhp, (int64_t)nc * hid, lp[D2R_RL_R0].w, E, K.dpooled, E);
      g0.batch(nc, hid, (int64_t)hid * E, (int64_t)B * E);
      TRY(d2r_gemm(&g0.d, c.st));
      G gw0(D2R_F32, D2R_F32, D2R_GEMM_TN, hid, E, B, K.dhp, (int64_t)nc * hid, L.pooled, E, lp[D2R_RL_R0].gw, E);
      gw0.batch(nc, hid, (int64_t)B * E, (int64_t)hid * E);
      gw0.d.beta = 1.f;
      TRY(d2r_gemm(&gw0.d, c.st));
      if (B <= 32) {
        TRY(d2r_colsum_add(D2R_F32, K.dhp, (int64_t)nc * hid, B, nc * hid, lp[D2R_RL_R0].gb, K.cws, K.cws_bytes, c.st));
      } else {
        TRY(d2r_colsum(D2R_F32, K.dhp, (int64_t)nc * hid, B, nc * hid, K.cs0, K.cws, K.cws_bytes, c.st));
        TRY(acc32(c, K.cs0, lp[D2R_RL_R0].gb, (int64_t)nc * hid));
      }
      // dx[0] already holds the aggregation's gradient (and, in the final layer, every dx[j] the skip-path gradient)
      TRY(d2r_meanpool_bwd_multi(c.dt, K.dpooled, nc, B, d.Lq, E, dx, final ? 0xffu : 1u, c.st));  // (one launch for the nc inputs)
    }
  }
  // from here on every dx[j] is initialised: the cells ACCUMULATE into it (GEMM epilogues, beta = 1)
  // --- GLAC: the SAF gate's backward (per-sample chain), which yields the gradients of the global and the local score -------------------
  const int64_t ldsg = (int64_t)n * E;
  if (nc > 1) {
    TRY(d2r_l2norm_bwd(c.dt, K.de[1], L.g_wsum, L.g_ne1, K.g_dwsum, B, E, c.st));
    // d w[b] = d wsum[b] S[b]^T (fp32), the gate's backward, then d S[b] = w[b]^T d wsum[b] + d a w_saf in ONE pass (rank-one products)
    TRY(d2r_saf_dweights(c.dt, K.g_dwsum, L.g_S, B, n, E, K.g_dwf, c.st));
    if (c.bn.fn && train) {
      TRY(d2r_saf_gate_bwd_ex(L.g_a, K.g_dwf, B, n, p.bn_weight, p.bn_bias, L.g_saved, train, K.g_daf, p.g_bn_weight, p.g_bn_bias, 1, c.bn.buf + 2,
                              c.bn.ntotal, nullptr, 0, 1, c.st));
      if (c.bn.fn(c.bn.user, c.bn.buf + 2, c.st)) return d2r_fail(D2R_ERR_INVALID, "d2r_interaction_bwd: bn_sync failed");
      TRY(d2r_saf_gate_bwd_ex(L.g_a, nullptr, B, n, p.bn_weight, p.bn_bias, L.g_saved, train, K.g_daf, nullptr, nullptr, 2, c.bn.buf + 2, c.bn.ntotal,
                              K.g_da16, c.dt, 0, c.st));
    } else {  // BatchNorm parameter gradients straight into their sinks; da also as the 16-bit operand of attn_sim_w's weight gradient
      TRY(d2r_saf_gate_bwd_ex(L.g_a, K.g_dwf, B, n, p.bn_weight, p.bn_bias, L.g_saved, train, K.g_daf, p.g_bn_weight, p.g_bn_bias, 0, nullptr, 0.0,
                              K.g_da16, c.dt, 1, c.st));
    }
    TRY(d2r_saf_dscores(c.dt, L.g_w16, K.g_dwsum, K.g_daf, lp[D2R_RL_GLAC_SAFW].w, B, n, E, K.g_dS, c.st));
    TRY(dwg(c, B * n, 1, E, K.g_da16, 1, L.g_S, E, lp[D2R_RL_GLAC_SAFW]));
    // split dS: row 0 of every sample = d sg (used in place, row stride n*E), rows 1.. = d sl (made contiguous)
    const size_t row = (size_t)E * d.es;
    TRY(copy2d(K.g_dsl, (size_t)d.Lq * row, (char*)K.g_dS + row, (size_t)n * row, (size_t)d.Lq * row, B, c.st));
  }
  // --- CRCMC: backward of its second core (d e4 -> d a, d b, d Ks as its value) ---------------------------------------------------------
  if (nc > 4)
    TRY(xat_bwd(c, d, L.r_a, L.r_b, E, L.r_Ks, E, d.Lq, K.de[4], L.e4, L.r_Qs, L.r_lse2, K.r_da, K.r_db, E, K.r_dKv, E, K.r_P2, K.r_dS2, 1.0f));
  // --- group B1: the last linear of every cell, backwards ---------------------------------------------------------------------------------
  if (nc > 2) {
    grp.dxg(c, T, d.hidi, E, K.de[2], E, lp[D2R_RL_IMRC_FC2].w, K.i_df1, d.hidi, 0.f, nullptr, L.f1, D2R_ACT_RELU);  // d f1_pre
    defer(jobs, T, E, d.hidi, K.de[2], E, L.f1, d.hidi, lp[D2R_RL_IMRC_FC2]);
  }
  if (nc > 1) {
    grp.dxg(c, T, E, E, K.g_dsl, E, lp[D2R_RL_GLAC_FC1].w, K.g_dl2, E);
    defer(jobs, T, E, E, K.g_dsl, E, L.g_l2, E, lp[D2R_RL_GLAC_FC1]);
  }
  if (nc > 3) {
    grp.dxg(c, T, E, E, K.de[3], E, lp[D2R_RL_CMRC_FC2].w, K.c_dfp, E, 0.f, nullptr, L.c_f, D2R_ACT_RELU);
    defer(jobs, T, E, E, K.de[3], E, L.c_f, E, lp[D2R_RL_CMRC_FC2]);
  }
  if (nc > 4) {
    grp.dxg(c, T, E, E, K.r_da, E, lp[D2R_RL_CRCMC_FC1].w, K.r_dQs, E, 0.f, K.de[4]);  // + residual Qs -> e4
    defer(jobs, T, E, E, K.r_da, E, L.r_Qs, E, lp[D2R_RL_CRCMC_FC1]);
    grp.dxg(c, T, E, E, K.r_db, E, lp[D2R_RL_CRCMC_FC2].w, K.r_dKs, E, 0.f, K.r_dKv);  // + Ks as the attention's value
    defer(jobs, T, E, E, K.r_db, E, L.r_Ks, E, lp[D2R_RL_CRCMC_FC2]);
  }
  TRY(grp.flush(c));
  if (nc > 1) TRY(d2r_l2norm_bwd(c.dt, K.g_dl2, L.g_loc, L.g_nloc, K.g_dloc, T, E, c.st));
  if (nc > 4) {
    TRY(d2r_act_bwd2(c.dt, D2R_ACT_TANH, K.r_dQs, L.r_Qs, K.r_dQsp, K.r_dKs, L.r_Ks, K.r_dKsp, TEn, c.st));
  }
  // --- group B2: the first linears, backwards ------------------------------------------------------------------------------------------------
  if (nc > 2) {
    grp.dxg(c, T, E, d.hidi, K.i_df1, d.hidi, lp[D2R_RL_IMRC_FC1].w, K.i_dy, E, 0.f, K.de[2]);  // + skip y -> e2
    defer(jobs, T, d.hidi, E, K.i_df1, d.hidi, L.y, E, lp[D2R_RL_IMRC_FC1]);
  }
  if (nc > 1) {
    grp.dxg(c, T, E, E, K.g_dloc, E, lp[D2R_RL_GLAC_LOC].w, K.g_dsq, E);
    defer(jobs, T, E, E, K.g_dloc, E, L.g_sq, E, lp[D2R_RL_GLAC_LOC]);
  }
  if (nc > 3) {
    grp.dxg(c, T, E, E, K.c_dfp, E, lp[D2R_RL_CMRC_FC1].w, K.c_dmod, E);
    defer(jobs, T, E, E, K.c_dfp, E, L.c_mod, E, lp[D2R_RL_CMRC_FC1]);
  }
  if (nc > 4) {
    grp.dxg(c, T, E, E, K.r_dQsp, E, lp[D2R_RL_CRCMC_MLP1].w, K.r_dc, E);
    defer(jobs, T, E, E, K.r_dQsp, E, L.r_c, E, lp[D2R_RL_CRCMC_MLP1]);
  }
  TRY(grp.flush(c));
  // --- IMRC's attention backward; the element-wise steps of GLAC and CMRC --------------------------------------------------------------------
  if (nc > 2) {
    const int dh = E / d.heads;
    const char* qkv = (const char*)L.qkv;
    char* dqkv = (char*)K.i_dqkv;
    const int64_t E3 = 3 * E, sb3 = (int64_t)d.Lq * 3 * E;
    TRY(d2r_mha_bwd(c.dt, qkv, E3, sb3, qkv + E * d.es, E3, sb3, qkv + 2 * E * d.es, E3, sb3, K.i_dy, E, TEe, nullptr, L.lse_i, K.i_dsum, dqkv, E3, sb3,
                    dqkv + E * d.es, E3, sb3, dqkv + 2 * E * d.es, E3, sb3, B, d.heads, d.Lq, d.Lq, dh, 1.0f / sqrtf((float)dh), 0.f, 0, c.st));
    defer(jobs, T, 3 * E, E, dqkv, 3 * E, refs[2], E, lp[D2R_RL_IMRC_QKV]);  // (with the other layers' q|k|v gradients: one grouped launch per module)
  }
  if (nc > 1) TRY(d2r_sqdiff_bwd(c.dt, refs[1], L.g_c, K.g_dsq, K.g_da, K.g_dc, TEn, c.st));
  if (nc > 3) {
    TRY(d2r_muladd_bwd(c.dt, refs[3], L.c_s, K.c_dmod, K.c_da, K.c_ds, TEn, c.st));
    TRY(d2r_act_bwd(c.dt, D2R_ACT_TANH, K.c_ds, L.c_s, K.c_dsp, TEn, c.st));
  }
  // --- group B3: CMRC's d c through fc_scale (fc_shift accumulates into it next), and - where the dx[j] are separate buffers - the tails
  //     that need nothing further: IMRC's q|k|v (+ skip x -> y) and CRCMC's second input branch ---------------------------------------------
  if (nc > 3) {
    grp.dxg(c, T, E, E, K.c_dsp, E, lp[D2R_RL_CMRC_SCALE].w, K.c_dc, E);
    defer(jobs, T, E, E, K.c_dsp, E, L.c_c, E, lp[D2R_RL_CMRC_SCALE]);
  }
  if (!first) {
    if (nc > 2) grp.dxg(c, T, E, 3 * E, K.i_dqkv, 3 * E, lp[D2R_RL_IMRC_QKV].w, dx[2], E, 1.f, K.i_dy);
    if (nc > 4) grp.dxg(c, T, E, E, K.r_dKsp, E, lp[D2R_RL_CRCMC_MLP2].w, dx[4], E, 1.f);
  }
  TRY(grp.flush(c));
  if (nc > 4) defer(jobs, T, E, E, K.r_dKsp, E, refs[4], E, lp[D2R_RL_CRCMC_MLP2]);
  if (nc > 3) {
    TRY(dxg(c, T, E, E, K.c_dmod, E, lp[D2R_RL_CMRC_SHIFT].w, K.c_dc, E, 1.f));
    defer(jobs, T, E, E, K.c_dmod, E, L.c_c, E, lp[D2R_RL_CMRC_SHIFT]);
  }
  // --- ONE backward launch for the alignment cores: dS, P, dQ; then dV / dK of every sample and core (one grouped launch) straight
  //     into the column blocks of the module-wide k|v gradient -------------------------------------------------------------------------------
  {
    const void *qa[3], *ka[3], *va[3], *ga[3], *oa[3];
    const float* la[3];
    void *dqa[3], *dka[3], *dva[3], *pa[3], *dsa[3];
    int ncore = 0;
    auto core = [&](const void* q, const void* kv, const void* o, const void* dO, const float* lse, void* dq, void* dkv, void* Pm, void* dS) {
      qa[ncore] = q, ka[ncore] = kv, va[ncore] = (const char*)kv + E * d.es, ga[ncore] = dO, la[ncore] = lse, oa[ncore] = o;
      dqa[ncore] = dq, dka[ncore] = dkv, dva[ncore] = (char*)dkv + E * d.es, pa[ncore] = Pm, dsa[ncore] = dS;
      ++ncore;
    };
    if (nc > 1) core(L.g_q, L.g_kv, L.g_c, K.g_dc, L.g_lse, K.g_dq, K.g_dkv, K.g_P, K.g_dSa);
    if (nc > 3) core(L.c_q, L.c_kv, L.c_c, K.c_dc, L.c_lse, K.c_dq, K.c_dkv, K.c_P, K.c_dSa);
    if (nc > 4) core(L.r_q, L.r_kv, L.r_c, K.r_dc, L.r_lse, K.r_dq, K.r_dkv, K.r_P, K.r_dSa);
    const int64_t skv = (int64_t)d.Lk * d.ldkv;
    if (ncore)
      TRY(d2r_xattn_bwd_multi(c.dt, ncore, qa, E, TEe, ka, d.ldkv, skv, va, d.ldkv, skv, ga, E, TEe, oa, E, TEe, nullptr, E, TEe, nullptr, la, dqa, E, TEe, dka, d.ldkv, skv, dva,
                              d.ldkv, skv, pa, dsa, d.lkp, B, d.Lq, d.Lk, E, XSCALE, c.st));
  }
  // --- the per-sample vector chains: GLAC's global branch, GESC (both accumulate into the cls rows of dx[1] / dx[5] and d_other) ----------
  //     in lock step: the dX products of the two chains (32 rows each) leave as one grouped launch per stage; the accumulating tails are
  //     grouped where they write different buffers (layer 0: all dx[j] alias d_own; both image-pool tails add into d_other)
  {
    Group pg;
    if (nc > 5) {
      TRY(d2r_lerp_bwd(c.dt, L.s_g, L.s_a, L.s_b, K.de[5], K.s_dg, K.s_da1, K.s_db1, BEn, c.st));
      TRY(d2r_softmax_bwd(c.dt, c.dt, L.s_g, K.s_dg, K.s_dz, E, B, E, 1.0f, c.st));
    }
    if (nc > 1) {
      pg.dxg(c, B, E, E, K.g_dS, ldsg, lp[D2R_RL_GLAC_FC2].w, K.g_dl2g, E);
      defer(jobs, B, E, E, K.g_dS, ldsg, L.g_l2g, E, lp[D2R_RL_GLAC_FC2]);
    }
    if (nc > 5) {
      pg.dxg(c, B, E, E, K.s_dz, E, lp[D2R_RL_GESC_MLP2].w, K.s_dz1p, E, 0.f, nullptr, L.s_z1, D2R_ACT_TANH);
      defer(jobs, B, E, E, K.s_dz, E, L.s_z1, E, lp[D2R_RL_GESC_MLP2]);
    }
    TRY(pg.flush(c));
    if (nc > 1) {
      TRY(d2r_l2norm_bwd(c.dt, K.g_dl2g, L.g_glo, L.g_nglo, K.g_dglo, B, E, c.st));
      pg.dxg(c, B, E, E, K.g_dglo, E, lp[D2R_RL_GLAC_GLO].w, K.g_ddg, E);
      defer(jobs, B, E, E, K.g_dglo, E, L.g_dg, E, lp[D2R_RL_GLAC_GLO]);
    }
    if (nc > 5) {
      pg.dxg(c, B, E, E, K.s_dz1p, E, lp[D2R_RL_GESC_MLP0].w, K.s_dab, E);
      defer(jobs, B, E, E, K.s_dz1p, E, L.s_ab, E, lp[D2R_RL_GESC_MLP0]);
    }
    TRY(pg.flush(c));
    if (nc > 1) {
      TRY(d2r_sqdiff_bwd(c.dt, L.g_pt, L.g_pi, K.g_ddg, K.g_dpt, K.g_dpi, BEn, c.st));
      TRY(d2r_act_bwd2(c.dt, D2R_ACT_TANH, K.g_dpt, L.g_pt, K.g_dptp, K.g_dpi, L.g_pi, K.g_dpip, BEn, c.st));
    }
    if (nc > 5) {
      TRY(d2r_add2(c.dt, K.s_da1, K.s_dab, K.s_dat, K.s_db1, K.s_dab, K.s_dbt, BEn, c.st));
      TRY(d2r_act_bwd2(c.dt, D2R_ACT_TANH, K.s_dat, L.s_a, K.s_dap, K.s_dbt, L.s_b, K.s_dbp, BEn, c.st));
    }
    // tails (beta = 1).  Order of the additions into a buffer as in the sequential schedule: GLAC's before GESC's.
    if (nc > 1) {
      pg.dxg(c, B, E, E, K.g_dptp, E, lp[D2R_RL_GLAC_TPOOL].w, dx[1], TEe, 1.f);  // token 0 of every sample
      defer(jobs, B, E, E, K.g_dptp, E, refs[1], TEe, lp[D2R_RL_GLAC_TPOOL]);
      pg.dxg(c, B, E, E, K.g_dpip, E, lp[D2R_RL_GLAC_IPOOL].w, d_other, SEe, 1.f);
      defer(jobs, B, E, E, K.g_dpip, E, other, SEe, lp[D2R_RL_GLAC_IPOOL]);
    }
    if (nc > 5 && !first) {  // dx[5] is a buffer of its own: GESC's text-pool tail joins the launch
      pg.dxg(c, B, E, E, K.s_dap, E, lp[D2R_RL_GESC_TPOOL].w, dx[5], TEe, 1.f);
      defer(jobs, B, E, E, K.s_dap, E, refs[5], TEe, lp[D2R_RL_GESC_TPOOL]);
    }
    TRY(pg.flush(c));
    if (nc > 5) {
      if (first) {
        pg.dxg(c, B, E, E, K.s_dap, E, lp[D2R_RL_GESC_TPOOL].w, dx[5], TEe, 1.f);
        defer(jobs, B, E, E, K.s_dap, E, refs[5], TEe, lp[D2R_RL_GESC_TPOOL]);
      }
      pg.dxg(c, B, E, E, K.s_dbp, E, lp[D2R_RL_GESC_IPOOL].w, d_other, SEe, 1.f);
      defer(jobs, B, E, E, K.s_dbp, E, other, SEe, lp[D2R_RL_GESC_IPOOL]);
      TRY(pg.flush(c));
    }
  }
  // --- the query-side projections (the key / value side of every cell and layer is one product at the end of the module): group B4 where
  //     the dx[j] are separate buffers, one after the other in layer 0 -------------------------------------------------------------------------
  if (nc > 3) TRY(d2r_add(c.dt, K.c_da, K.de[3], K.c_tmp, TEn, c.st));  // FiLM path + skip x -> e3
  Group tails;
  Group& tg = first ? tails : grp;
  auto step = [&]() -> int { return first ? tails.flush(c) : D2R_OK; };  // layer 0: every product alone, in this order
  if (nc > 1) {
    tg.dxg(c, T, E, E, K.g_dq, E, lp[D2R_RL_GLAC_Q].w, dx[1], E, 1.f, K.g_da);  // += dq Wq + d(sqdiff)/dx
    defer(jobs, T, E, E, K.g_dq, E, refs[1], E, lp[D2R_RL_GLAC_Q]);
    TRY(step());
  }
  if (nc > 3) {
    tg.dxg(c, T, E, E, K.c_dq, E, lp[D2R_RL_CMRC_Q].w, dx[3], E, 1.f, K.c_tmp);
    defer(jobs, T, E, E, K.c_dq, E, refs[3], E, lp[D2R_RL_CMRC_Q]);
    TRY(step());
  }
  if (nc > 4) {
    tg.dxg(c, T, E, E, K.r_dq, E, lp[D2R_RL_CRCMC_Q].w, dx[4], E, 1.f);
    defer(jobs, T, E, E, K.r_dq, E, refs[4], E, lp[D2R_RL_CRCMC_Q]);
    TRY(step());
  }
  TRY(grp.flush(c));
  if (first) {
    if (nc > 4) TRY(dxg(c, T, E, E, K.r_dKsp, E, lp[D2R_RL_CRCMC_MLP2].w, dx[4], E, 1.f));
    if (nc > 2) TRY(dxg(c, T, E, 3 * E, K.i_dqkv, 3 * E, lp[D2R_RL_IMRC_QKV].w, dx[2], E, 1.f, K.i_dy));  // += dqkv Wqkv + skip x -> y
  }
  return D2R_OK;
}

int check(const d2r_interaction_desc* D, const char* fn, bool bwd, void* stream) {
  D2R_REQUIRE(D != nullptr, "%s: null descriptor", fn);
  D2R_REQUIRE(D->B >= 1 && D->Lq >= 1 && D->Lk >= 1 && D->nlayer >= 2 && D->hid_router >= 1 && D->hid_imrc >= 8 && D->heads_imrc >= 1,
              "%s: bad shape", fn);
  D2R_REQUIRE(d2r_interaction_supported(D->dtype, D->Lq, D->Lk, D->ncell, D->heads_imrc),
              "%s: unsupported (bf16, 2..6 cells, token counts within the fused attention cores' limits)", fn);
  D2R_REQUIRE(D->layers && D->own && D->other && D->out && D->paths && D->arena && d2r_aligned16(D->arena), "%s: null / unaligned pointer", fn);
  D2R_REQUIRE(!D->bn_sync || (D->bn_sync_buf && D->bn_world >= 1), "%s: bn_sync needs bn_sync_buf (4 doubles per layer) and bn_world >= 1", fn);
  D2R_REQUIRE(D->kv_all.w && D->kv_all.b && (!bwd || (D->kv_all.gw && D->kv_all.gb)), "%s: kv_all (the fused k|v projections of `other`) missing", fn);
  D2R_REQUIRE(D->arena_bytes >= d2r_interaction_arena_bytes(D->B, D->Lq, D->Lk, D->ncell, D->nlayer, D->hid_router, D->hid_imrc),
              "%s: arena too small", fn);
  if (bwd) {
    D2R_REQUIRE(D->d_own && D->d_other && D->scratch && d2r_aligned16(D->scratch), "%s: null / unaligned gradient pointer", fn);
    D2R_REQUIRE(D->scratch_bytes >= d2r_interaction_bwd_scratch(D->B, D->Lq, D->Lk, D->ncell, D->nlayer, D->hid_router, D->hid_imrc),
                "%s: scratch too small", fn);
  }
  return D2R_OK;
}

}  // namespace

extern "C" int d2r_interaction_supported(int dtype, int Lq, int Lk, int ncell, int heads_imrc) {
  if (!d2r_is16(dtype) || ncell < 2 || ncell > 6 || heads_imrc < 1 || heads_imrc > 64 || E % heads_imrc) return 0;
  if (!d2r_xattn_supported(dtype, Lq, Lk, E)) return 0;
  if (ncell > 4 && !d2r_xattn_supported(dtype, Lq, Lq, E)) return 0;
  if (ncell > 2 && !d2r_mha_supported(dtype, Lq, Lq, E / heads_imrc)) return 0;
  return 1;
}

extern "C" size_t d2r_interaction_arena_bytes(int B, int Lq, int Lk, int ncell, int nlayer, int hid_router, int hid_imrc) {
  const Dims d = make_dims(B, Lq, Lk, ncell, nlayer, hid_router, hid_imrc, 16);
  Arena A(nullptr);
  LayerF L;
  A.take((size_t)d.S * d.ldkv * d.es);  // the module-wide k|v projection of `other`
  for (int l = 0; l < nlayer; ++l) plan_fwd(A, d, l == nlayer - 1 ? 1 : ncell, l == 0, l == nlayer - 1, L, nullptr, l);
  return A.off + 256;
}

extern "C" size_t d2r_interaction_bwd_scratch(int B, int Lq, int Lk, int ncell, int nlayer, int hid_router, int hid_imrc) {
  const Dims d = make_dims(B, Lq, Lk, ncell, nlayer, hid_router, hid_imrc, 16);
  Arena A(nullptr);
  LayerB K;
  A.take((size_t)d.S * d.ldkv * d.es);  // gradient of the module-wide k|v projection
  for (int l = 0; l < nlayer; ++l) plan_bwd(A, d, l == nlayer - 1 ? 1 : ncell, l == 0, l == nlayer - 1, K, nullptr, l);
  A.take((size_t)d.T * E * d.es);  // zero gradient standing in for a missing d_out
  return A.off + 256;
}

extern "C" int d2r_interaction_fwd(const d2r_interaction_desc* D, void* stream) {
  TRY(check(D, "d2r_interaction_fwd", false, stream));
  const Dims d = make_dims(D->B, D->Lq, D->Lk, D->ncell, D->nlayer, D->hid_router, D->hid_imrc, D->heads_imrc);
  const Ctx c{D->dtype, stream, D->splitk_ws, D->splitk_bytes};
  Arena A(D->arena);
  const int nc = d.nc, total = nc * nc * (d.nl - 1) + nc;
  // keys | values of every alignment cell of every layer: `other` is the same tensor in all of them
  // (models/DynamicInteraction.py:95-102), so ONE projection with N = nkv * 1536 (13,824 at DR_step 3) serves the module
  char* kvall = (char*)A.take((size_t)d.S * d.ldkv * d.es);
  if (d.nkv) TRY(lin(c, d.S, (int)d.ldkv, E, D->other, E, D->kv_all, kvall));
  const void* refs[6];
  for (int j = 0; j < 6; ++j) refs[j] = D->own;
  LayerF L;
  for (int l = 0; l < d.nl; ++l) {
    const bool first = l == 0, final = l == d.nl - 1;
    const int P = final ? 1 : nc;
    plan_fwd(A, d, P, first, final, L, kvall, l);
    Ctx cl = c;
    if (D->bn_sync) cl.bn = BnSync{D->bn_sync, D->bn_sync_user, D->bn_sync_buf + 4 * l, (double)D->bn_world * d.B * d.n};
    TRY(layer_fwd(cl, d, D->layers[l], P, first, final, D->train, refs, D->other, L, D->out, D->paths + (size_t)l * nc * nc, total));
    for (int j = 0; j < nc && !final; ++j) refs[j] = L.outs[j];
  }
  return D2R_OK;
}

extern "C" int d2r_interaction_bwd(const d2r_interaction_desc* D, void* stream) {
  TRY(check(D, "d2r_interaction_bwd", true, stream));
  const Dims d = make_dims(D->B, D->Lq, D->Lk, D->ncell, D->nlayer, D->hid_router, D->hid_imrc, D->heads_imrc);
  const Ctx c{D->dtype, stream, D->splitk_ws, D->splitk_bytes};
  const int nc = d.nc, nl = d.nl, total = nc * nc * (nl - 1) + nc;
  const size_t TE = (size_t)d.T * E * d.es;
  std::vector<LayerF> F(nl);
  std::vector<LayerB> K(nl);
  Arena A(D->arena), Z(D->scratch);
  char* kvall = (char*)A.take((size_t)d.S * d.ldkv * d.es);
  char* dkvall = (char*)Z.take((size_t)d.S * d.ldkv * d.es);
  for (int l = 0; l < nl; ++l) plan_fwd(A, d, l == nl - 1 ? 1 : nc, l == 0, l == nl - 1, F[l], kvall, l);
  for (int l = 0; l < nl; ++l) plan_bwd(Z, d, l == nl - 1 ? 1 : nc, l == 0, l == nl - 1, K[l], dkvall, l);
  void* zero_out = Z.take(TE);
  hipError_t e = hipMemsetAsync(D->d_other, 0, (size_t)d.S * E * d.es, (hipStream_t)stream);
  if (e == hipSuccess && !D->d_out) e = hipMemsetAsync(zero_out, 0, TE, (hipStream_t)stream);
  if (e != hipSuccess) return d2r_fail(D2R_ERR_LAUNCH, "d2r_interaction_bwd: hipMemsetAsync failed: %s", hipGetErrorString(e));
  Jobs jobs;
  for (int l = nl - 1; l >= 0; --l) {
    const bool first = l == 0, final = l == nl - 1;
    const int P = final ? 1 : nc;
    const void* refs[6];
    void* dx[6];
    const void* douts[6];
    for (int j = 0; j < 6; ++j) {
      refs[j] = first ? D->own : (j < nc ? F[l - 1].outs[j] : nullptr);
      dx[j] = first ? D->d_own : K[l].dx[j];
      douts[j] = final ? (j == 0 ? (D->d_out ? D->d_out : zero_out) : nullptr) : K[l + 1].dx[j];
    }
    const float* dprobs = D->d_paths ? D->d_paths + (size_t)l * nc * nc : nullptr;
    Ctx cl = c;
    if (D->bn_sync) cl.bn = BnSync{D->bn_sync, D->bn_sync_user, D->bn_sync_buf + 4 * l, (double)D->bn_world * d.B * d.n};
    TRY(layer_bwd(cl, d, D->layers[l], P, first, final, D->train, refs, D->other, F[l], K[l], douts, D->out, dprobs, total, dx, D->d_other, jobs));
  }
  if (d.nkv) {
    // every cell and layer wrote its dK | dV block: the gradient w.r.t. `other` through ALL key / value projections is one product
    // with a 13,824-long reduction (fp32 accumulation inside the MFMAs instead of nine 16-bit read-modify-write passes over d_other),
    // and their weight gradients one [13824, 768] product over the S rows
    TRY(dxg(c, d.S, E, (int)d.ldkv, dkvall, d.ldkv, D->kv_all.w, D->d_other, E, 1.f));
    defer(jobs, d.S, (int)d.ldkv, E, dkvall, d.ldkv, D->other, E, D->kv_all);
  }
  return flush_jobs(c, jobs);
}

// ======================================================================================================================================
// K17: the classification head as one call each way (Block fusion -> fc -> cross entropy -> loss = ce + js), fp32 throughout.
// The same launches, in the same order, with the same descriptors as the op-by-op path of d2r_amd/modules.py (Block.forward,
// UnimoModelF.forward) - what goes away is the host time between them: this is the one stretch of a training step where both
// branch streams are idle (forward of the head, then its backward, ~35 short launches), and it was paced by the host.
// ======================================================================================================================================
namespace {

struct HeadDims {
  int B, E, mm, chunks, rank, size, RS, classes;
  size_t n_h, n_m;  // floats per [B, mm] and [B, chunks * rank * size] tensor
};
int head_dims(const d2r_head_desc* D, HeadDims& h) {
  D2R_REQUIRE(D && D->B >= 1 && D->E >= 16 && D->mm >= 16 && D->chunks >= 1 && D->rank >= 1 && D->classes >= 1, "d2r_head: bad sizes");
  D2R_REQUIRE(D->mm % D->chunks == 0, "d2r_head: mm must be a multiple of chunks");
  h.B = D->B, h.E = D->E, h.mm = D->mm, h.chunks = D->chunks, h.rank = D->rank, h.size = D->mm / D->chunks, h.classes = D->classes;
  h.RS = h.rank * h.size;
  h.n_h = (size_t)h.B * h.mm, h.n_m = (size_t)h.B * h.chunks * h.RS;
  return D2R_OK;
}
struct HeadActs {  // forward activations kept for the backward pass (arena)
  float *h0, *h1, *m0, *m1, *z, *zraw, *ce;
};
void head_plan(Arena& A, const HeadDims& h, HeadActs& a) {
  a.h0 = (float*)A.take(h.n_h * 4), a.h1 = (float*)A.take(h.n_h * 4), a.m0 = (float*)A.take(h.n_m * 4), a.m1 = (float*)A.take(h.n_m * 4);
  a.z = (float*)A.take(h.n_h * 4), a.zraw = (float*)A.take(h.n_h * 4), a.ce = (float*)A.take(16);
}
struct HeadGrads {  // backward scratch
  float *dlogits, *dpooled, *dz, *dm0, *dm1, *dh0, *dh1, *tmpb;
  void* csws;
  size_t cswsb;
};
void head_plan_bwd(Arena& Z, const HeadDims& h, HeadGrads& g) {
  g.dlogits = (float*)Z.take((size_t)h.B * h.classes * 4), g.dpooled = (float*)Z.take((size_t)h.B * h.E * 4), g.dz = (float*)Z.take(h.n_h * 4);
  g.dm0 = (float*)Z.take(h.n_m * 4), g.dm1 = (float*)Z.take(h.n_m * 4), g.dh0 = (float*)Z.take(h.n_h * 4), g.dh1 = (float*)Z.take(h.n_h * 4);
  g.tmpb = (float*)Z.take((size_t)h.chunks * h.RS * 4);
  g.cswsb = d2r_colsum_workspace(h.B, h.chunks * h.RS);
  g.csws = Z.take(g.cswsb);
}
// the `chunks` rank projections of one side as ONE batched product: y[:, c] = x[:, c] W_c^T + b_c  (x [B, mm], W_c [RS, size])
int head_merge_fwd(const Ctx& c, const HeadDims& h, const float* x, const d2r_linear_params& p, float* y) {
  G g(D2R_F32, D2R_F32, D2R_GEMM_NT, h.B, h.RS, h.size, x, h.mm, p.w, h.size, y, (int64_t)h.chunks * h.RS);
  g.batch(h.chunks, h.size, (int64_t)h.RS * h.size, h.RS);
  g.d.bias = p.b, g.d.s_bias_b = h.RS;
  return d2r_gemm(&g.d, c.st);
}
int head_merge_bwd(const Ctx& c, const HeadDims& h, const float* x, const d2r_linear_params& p, const float* dy, float* dx, const HeadGrads& s) {
  {
    G g(D2R_F32, D2R_F32, D2R_GEMM_NN, h.B, h.size, h.RS, dy, (int64_t)h.chunks * h.RS, p.w, h.size, dx, h.mm);
    g.batch(h.chunks, h.RS, (int64_t)h.RS * h.size, h.size);
    TRY(d2r_gemm(&g.d, c.st));
  }
  {
    G g(D2R_F32, D2R_F32, D2R_GEMM_TN, h.RS, h.size, h.B, dy, (int64_t)h.chunks * h.RS, x, h.mm, p.gw, h.size);
    g.batch(h.chunks, h.RS, h.size, (int64_t)h.RS * h.size);
    g.d.beta = 1.f;
    TRY(d2r_gemm(&g.d, c.st));
  }
  TRY(d2r_colsum(D2R_F32, dy, (int64_t)h.chunks * h.RS, h.B, h.chunks * h.RS, s.tmpb, s.csws, s.cswsb, c.st));
  return d2r_axpby(D2R_F32, 1.f, s.tmpb, 1.f, p.gb, (int64_t)h.chunks * h.RS, c.st);
}

}  // namespace

extern "C" size_t d2r_head_arena_bytes(int B, int E, int mm, int chunks, int rank, int classes) {
  d2r_head_desc D = {};
  D.B = B, D.E = E, D.mm = mm, D.chunks = chunks, D.rank = rank, D.classes = classes;
  HeadDims h;
  if (head_dims(&D, h) != D2R_OK) return 0;
  Arena A(nullptr);
  HeadActs a;
  head_plan(A, h, a);
  return A.off;
}
extern "C" size_t d2r_head_bwd_scratch(int B, int E, int mm, int chunks, int rank, int classes) {
  d2r_head_desc D = {};
  D.B = B, D.E = E, D.mm = mm, D.chunks = chunks, D.rank = rank, D.classes = classes;
  HeadDims h;
  if (head_dims(&D, h) != D2R_OK) return 0;
  Arena Z(nullptr);
  HeadGrads g;
  head_plan_bwd(Z, h, g);
  return Z.off;
}

extern "C" int d2r_head_fwd(const d2r_head_desc* D, void* stream) {
  HeadDims h;
  TRY(head_dims(D, h));
  D2R_REQUIRE(D->x0 && D->x1 && D->labels && D->js && D->loss && D->logits && D->pooled && D->arena, "d2r_head_fwd: null pointer");
  D2R_REQUIRE(D->arena_bytes >= d2r_head_arena_bytes(h.B, h.E, h.mm, h.chunks, h.rank, h.classes), "d2r_head_fwd: arena too small");
  const Ctx c{D2R_F32, stream, D->splitk_ws, D->splitk_bytes};
  Arena A(D->arena);
  HeadActs a;
  head_plan(A, h, a);
  TRY(lin(c, h.B, h.mm, h.E, D->x0, h.E, D->lin0, a.h0));
  TRY(lin(c, h.B, h.mm, h.E, D->x1, h.E, D->lin1, a.h1));
  TRY(head_merge_fwd(c, h, a.h0, D->merge0, a.m0));
  TRY(head_merge_fwd(c, h, a.h1, D->merge1, a.m1));
  TRY(d2r_block_merge_fwd(D2R_F32, a.m0, a.m1, h.B, h.chunks, h.rank, h.size, a.z, a.zraw, stream));
  TRY(lin(c, h.B, h.E, h.mm, a.z, h.mm, D->lin_out, D->pooled));
  TRY(lin(c, h.B, h.classes, h.E, D->pooled, h.E, D->fc, D->logits));
  TRY(d2r_ce_fwd(D->logits, D->labels, h.B, h.classes, a.ce, stream));
  const float* xs[2] = {a.ce, D->js};
  const float coef[2] = {1.f, 1.f};
  return d2r_lincomb(xs, coef, 2, D->loss, stream);
}

extern "C" int d2r_head_bwd(const d2r_head_desc* D, void* stream) {
  HeadDims h;
  TRY(head_dims(D, h));
  D2R_REQUIRE(D->x0 && D->x1 && D->labels && D->logits && D->pooled && D->arena && D->d_loss && D->d_x0 && D->d_x1 && D->d_js && D->scratch,
              "d2r_head_bwd: null pointer");
  D2R_REQUIRE(D->scratch_bytes >= d2r_head_bwd_scratch(h.B, h.E, h.mm, h.chunks, h.rank, h.classes), "d2r_head_bwd: scratch too small");
  for (const d2r_linear_params* p : {&D->lin0, &D->lin1, &D->merge0, &D->merge1, &D->lin_out, &D->fc})
    D2R_REQUIRE(p->gw && p->gb, "d2r_head_bwd: every linear needs its gradient sinks");
  const Ctx c{D2R_F32, stream, D->splitk_ws, D->splitk_bytes};
  Arena A(D->arena), Z(D->scratch);
  HeadActs a;
  HeadGrads g;
  head_plan(A, h, a);
  head_plan_bwd(Z, h, g);
  TRY(d2r_axpby(D2R_F32, 1.f, D->d_loss, 0.f, D->d_js, 1, stream));  // loss = ce + js
  TRY(d2r_ce_bwd(D->logits, D->labels, h.B, h.classes, D->d_loss, g.dlogits, stream));
  if (D->d_logits) TRY(d2r_axpby(D2R_F32, 1.f, D->d_logits, 1.f, g.dlogits, (int64_t)h.B * h.classes, stream));  // a second consumer of the logits
  TRY(dxg(c, h.B, h.E, h.classes, g.dlogits, h.classes, D->fc.w, g.dpooled, h.E));
  TRY(dwg(c, h.B, h.classes, h.E, g.dlogits, h.classes, D->pooled, h.E, D->fc));
  if (D->d_pooled) TRY(d2r_axpby(D2R_F32, 1.f, D->d_pooled, 1.f, g.dpooled, (int64_t)h.B * h.E, stream));  // ... of Block's output
  TRY(dxg(c, h.B, h.mm, h.E, g.dpooled, h.E, D->lin_out.w, g.dz, h.mm));
  TRY(dwg(c, h.B, h.E, h.mm, g.dpooled, h.E, a.z, h.mm, D->lin_out));
  TRY(d2r_block_merge_bwd(D2R_F32, a.m0, a.m1, a.zraw, g.dz, h.B, h.chunks, h.rank, h.size, g.dm0, g.dm1, stream));
  TRY(head_merge_bwd(c, h, a.h0, D->merge0, g.dm0, g.dh0, g));
  TRY(head_merge_bwd(c, h, a.h1, D->merge1, g.dm1, g.dh1, g));
  TRY(dxg(c, h.B, h.E, h.mm, g.dh0, h.mm, D->lin0.w, D->d_x0, h.E));
  TRY(dwg(c, h.B, h.mm, h.E, g.dh0, h.mm, D->x0, h.E, D->lin0));
  TRY(dxg(c, h.B, h.E, h.mm, g.dh1, h.mm, D->lin1.w, D->d_x1, h.E));
  return dwg(c, h.B, h.mm, h.E, g.dh1, h.mm, D->x1, h.E, D->lin1);
}

