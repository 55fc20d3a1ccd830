"""Host-side mirror of the reference's model classes for the D2R hot path, computing through libd2r_hip.so.

Class / attribute names reproduce the reference's ``state_dict`` keys exactly (1,210 keys at DR_step=3) so that
``best_model.pth`` checkpoints and the weight-ingest rename rule (modules/train.py:92-111) interchange.  All
arithmetic is done by the HIP kernels behind ``d2r_amd.functional``; these classes only own parameters and
sequence launches.

Reference map (file:line under the reference tree):
  Router                           models/Router.py:10-26
  *Cell                            models/Cells.py:30-40,42-60,76-87,131-175,179-218,222-255
  SelfAttention / Refinement       models/SelfAttention.py:11-70, models/Refinement.py:86-154
  CrossModalAlignment / SAF / Block  models/XModules.py:277-328,366-394,478-555
  DynamicInteraction layers        models/DynamicInteraction.py:20-254
  (Reversed_)InteractionModule     models/InteractionModule.py:9-108
  encoders, UnimoModel             models/modeling_unimo.py:87-527,649-894
  UnimoModelF                      models/unimo_model.py:138-162
"""
from __future__ import annotations

import contextlib
import math
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import functional as F
from ._lib import (ACT_GELU, ACT_NONE, ACT_QUICK_GELU, ACT_RELU, ACT_TANH, ACT_TANH_RELU)

E = 768  # the routing cells are hard-wired to 768 features (models/Cells.py:140-143)
CELL_ORDER = ("ric", "glac", "imrc", "cmrc", "crcmc", "gesc")  # index j of models/DynamicInteraction.py:41-48


class D2RModule(nn.Module):
    """Base: carries the compute dtype of the HIP path (fp32 or bf16 activations/weights)."""

    cdtype = torch.float32

    def set_compute_dtype(self, dtype: torch.dtype):
        assert dtype in (torch.float32, torch.bfloat16, torch.float16)
        for m in self.modules():
            if isinstance(m, D2RModule):
                m.cdtype = dtype
        return self

    def fusion_groups(self):
        """(owner module, [Linear, ...]) groups of same-input projections that ParamStore lays out back to back
        so that ONE GEMM serves them (q|k|v of every self-attention, k|v of every cross-modal alignment)."""
        for m in self.modules():
            members = getattr(m, "_fusion_members", None)
            if members is not None:
                got = members()
                if isinstance(got, dict):
                    for key, linears in got.items():
                        yield m, key, list(linears)
                else:
                    yield m, None, list(got)

    def _fused_linear(self, key=None):
        """The FusedLinear installed by ParamStore, or None on an un-prepared model (then the members run singly)."""
        d = getattr(self, "_fused", None)
        return None if d is None else d.get(key)


def _compute_weight(p: nn.Parameter, dtype: torch.dtype) -> torch.Tensor:
    if dtype == torch.float32:
        return p
    lp = getattr(p, "_d2r_lp", None)  # bf16 shadow maintained by d2r_amd.params.ParamStore / the fused AdamW
    if lp is not None:
        return lp
    return F.cast(p.detach(), dtype)  # un-prepared model: cast on the fly (still a HIP kernel)


class Linear(D2RModule):
    """nn.Linear parameters (fp32 masters) + fused GEMM epilogue."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        bound = 1.0 / math.sqrt(in_features)
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound)) if bias else None

    def forward(self, x, act=ACT_NONE, residual=None, fp32=False, out_dtype=None):
        """fp32=True keeps the whole product in fp32 (router / gate logits)."""
        if fp32:
            return F.linear(x, self.weight, self.bias, None, act, residual, out_dtype)
        return F.linear(x, self.weight, self.bias, _compute_weight(self.weight, self.cdtype), act, residual, out_dtype)


class LayerNorm(D2RModule):
    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        return F.layer_norm(x, self.weight, self.bias, self.eps)


class BatchNorm1dScalar(nn.Module):
    """BatchNorm1d(1) parameters/buffers of AttentionFiltration (models/XModules.py:376)."""

    def __init__(self, num_features=1):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Sequential(nn.Module):
    """Index-named children ('0', '2', ...) like nn.Sequential, without its forward."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            self.add_module(k, v)

    def __getitem__(self, i):
        return self._modules[str(i)]


# ------------------------------------------------------------------------------------------------------
# router + cells
# ------------------------------------------------------------------------------------------------------
class Router(D2RModule):
    """relu(tanh(W2 relu(W1 mean_tokens(x)))) — fp32 end to end so that routing decisions are exact."""

    def __init__(self, num_out_path, embed_size, hid):
        super().__init__()
        self.num_out_path = num_out_path
        self.mlp = _Sequential(**{"0": Linear(embed_size, hid), "2": Linear(hid, num_out_path)})
        with torch.no_grad():
            self.mlp[2].bias.fill_(1.5)  # models/Router.py:19-20

    def gate_from_pooled(self, pooled):  # pooled: fp32 [B, 768]
        h = self.mlp[0](pooled, act=ACT_RELU, fp32=True)
        return self.mlp[2](h, act=ACT_TANH_RELU, fp32=True)

    def forward(self, x):
        return self.gate_from_pooled(F.mean_pool([x])[0])


class BertPooler(D2RModule):
    def __init__(self, hidden=E):
        super().__init__()
        self.dense = Linear(hidden, hidden)

    def forward(self, hidden_states, fp32=False):
        if fp32:  # loss-side poolers: [B,768] only, kept fp32 (the Block signed-sqrt amplifies bf16 rounding)
            return self.dense(F.cast_ad(hidden_states[:, 0].contiguous(), torch.float32), act=ACT_TANH, fp32=True)
        return self.dense(hidden_states[:, 0], act=ACT_TANH)  # strided rows, no copy


class CrossModalAlignment(D2RModule):
    """softmax(100 (Wq t)(Wk i)^T / sqrt(768)) (Wv i): the live part of models/XModules.py:300-310 (identical in
    models/Refinement.py:105-115).  fc_1/fc_2 are kept as (dead) parameters; the discarded reverse-attention /
    contrastive branch (XModules.py:312-326) is not computed."""

    def __init__(self):
        super().__init__()
        self.query, self.key, self.value = Linear(E, E), Linear(E, E), Linear(E, E)
        self.fc_1, self.fc_2 = Linear(E, E), Linear(E, E)

    def _fusion_members(self):
        return [self.key, self.value]

    def forward(self, own, other, kv=None):
        """kv: this alignment's [B, Lk, 1536] block of the module-wide k|v projection of `other` (computed once per module
        forward by _InteractionBase), or None -> projected here."""
        if kv is not None:
            return F.attention_kv(self.query(own), kv, 1, 100.0 / math.sqrt(E))
        fz = self._fused_linear()
        if fz is not None:
            return F.attention_kv(self.query(own), fz(other, self.cdtype), 1, 100.0 / math.sqrt(E))
        return F.attention(self.query(own), self.key(other), self.value(other), 1, 100.0 / math.sqrt(E))


class RectifiedIdentityCell(D2RModule):
    """emb = relu(x): the relu is fused into the route_aggregate kernel, so the cell returns its input."""

    def __init__(self, args, num_out_path):
        super().__init__()
        self.router = Router(num_out_path, args.embed_size, args.hid_router)

    def forward(self, x, other=None, kv=None):
        return x


class AttentionLayer(D2RModule):
    def __init__(self, embed_size, h):
        super().__init__()
        self.h = h
        self.linears = nn.ModuleList([Linear(embed_size, embed_size) for _ in range(3)])

    def _fusion_members(self):
        return list(self.linears)


class FeedForward(D2RModule):
    def __init__(self, embed_size, hidden):
        super().__init__()
        self.fc1, self.fc2 = Linear(embed_size, hidden), Linear(hidden, embed_size)


class SelfAttention(D2RModule):
    """IMRC body (models/SelfAttention.py:56-70): y = x + MHA16(x) (no out-proj); y + W2 relu(W1 y)."""

    def __init__(self, embed_size, hid_size, h):
        super().__init__()
        self.h = h
        self.att_layer = AttentionLayer(embed_size, h)
        self.feed_forward_layer = FeedForward(embed_size, hid_size)

    def forward(self, x):
        fz = self.att_layer._fused_linear()
        scale = 1.0 / math.sqrt(x.shape[-1] // self.h)
        if fz is not None:
            y = F.attention_qkv(fz(x, self.cdtype), self.h, scale, residual=x)
        else:
            q, k, v = (lin(x) for lin in self.att_layer.linears)
            y = F.attention(q, k, v, self.h, scale, residual=x)
        return self.feed_forward_layer.fc2(self.feed_forward_layer.fc1(y, act=ACT_RELU), residual=y)


class IntraModelReasoningCell(D2RModule):
    def __init__(self, args, num_out_path):
        super().__init__()
        self.router = Router(num_out_path, args.embed_size, args.hid_router)
        self.sa = SelfAttention(args.embed_size, args.hid_IMRC, args.num_head_IMRC)

    def forward(self, x, other=None, kv=None):
        return self.sa(x)


class Refinement(D2RModule):
    """CMRC body (models/Refinement.py:133-154)."""

    def __init__(self, embed_size):
        super().__init__()
        self.fc_scale, self.fc_shift = Linear(embed_size, embed_size), Linear(embed_size, embed_size)
        self.fc_1, self.fc_2 = Linear(embed_size, embed_size), Linear(embed_size, embed_size)
        self.CrossModalAlignment = CrossModalAlignment()

    def forward(self, own, other, kv=None):
        c = self.CrossModalAlignment(own, other, kv)
        mod = F.muladd(own, self.fc_scale(c, act=ACT_TANH), self.fc_shift(c))
        return self.fc_2(self.fc_1(mod, act=ACT_RELU), residual=own)


class CrossModalRefinementCell(D2RModule):
    def __init__(self, args, num_out_path):
        super().__init__()
        self.refine = Refinement(args.embed_size)
        self.router = Router(num_out_path, args.embed_size, args.hid_router)

    def forward(self, own, other, kv=None):
        return self.refine(own, other, kv)


class AttentionFiltration(D2RModule):
    """SAF (models/XModules.py:366-394): l2norm(l1norm(sigmoid(BN1(w.S))) @ S)."""

    def __init__(self, sim_dim):
        super().__init__()
        self.attn_sim_w = Linear(sim_dim, 1)
        self.bn = BatchNorm1dScalar(1)
        r = math.sqrt(6.0) / math.sqrt(sim_dim + 1)
        with torch.no_grad():
            self.attn_sim_w.weight.uniform_(-r, r)
            self.attn_sim_w.bias.zero_()

    def forward(self, S):
        B, n, _ = S.shape
        a = self.attn_sim_w(S, out_dtype=torch.float32).view(B, n)
        w = F.saf_gate(a, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var, self.training)
        if self.training:
            self.bn.num_batches_tracked += 1
        return F.l2norm(F.weighted_row_sum(F.cast_ad(w, S.dtype), S))


class GlobalLocalAlignmentCell(D2RModule):
    def __init__(self, args, num_out_path):
        super().__init__()
        self.router = Router(num_out_path, args.embed_size, args.hid_router)
        self.CrossModalAlignment = CrossModalAlignment()
        self.SAF_module = AttentionFiltration(E)
        self.text_cls_pool, self.image_cls_pool = BertPooler(), BertPooler()
        self.fc_sim_tranloc, self.fc_sim_tranglo = Linear(E, E), Linear(E, E)
        self.fc_1, self.fc_2 = Linear(E, E), Linear(E, E)

    def forward(self, own, other, kv=None):
        c = self.CrossModalAlignment(own, other, kv)
        sl = self.fc_1(F.l2norm(self.fc_sim_tranloc(F.sqdiff(own, c))))  # [B,Lq,768]
        dg = F.sqdiff(self.text_cls_pool(own), self.image_cls_pool(other))
        sg = self.fc_2(F.l2norm(self.fc_sim_tranglo(dg)))  # [B,768]
        S = torch.cat([sg.unsqueeze(1), sl], dim=1)  # memory plumbing only
        return self.SAF_module(S)  # [B,768], broadcast over Lq by the aggregate kernel


class GlobalEnhancedSemanticCell(D2RModule):
    def __init__(self, args, num_out_path):
        super().__init__()
        self.router = Router(num_out_path, args.embed_size, args.hid_router)
        self.text_cls_pool, self.image_cls_pool = BertPooler(), BertPooler()
        self.fc_mlp = _Sequential(**{"0": Linear(E, E), "2": Linear(E, E)})

    def forward(self, own, other, kv=None):
        a, b = self.text_cls_pool(own), self.image_cls_pool(other)
        z = self.fc_mlp[2](self.fc_mlp[0](F.add(a, b), act=ACT_TANH))
        return F.lerp_gate(F.softmax_rows(z), a, b)  # [B,768]


class ContextRichCrossModalCell(D2RModule):
    def __init__(self, args, num_out_path):
        super().__init__()
        self.router = Router(num_out_path, args.embed_size, args.hid_router)
        self.CrossModalAlignment = CrossModalAlignment()
        self.fc_mlp_1 = _Sequential(**{"0": Linear(E, E)})
        self.fc_mlp_2 = _Sequential(**{"0": Linear(E, E)})
        self.fc_1, self.fc_2 = Linear(E, E), Linear(E, E)

    def forward(self, own, other, kv=None):
        c = self.CrossModalAlignment(own, other, kv)
        Qs = self.fc_mlp_1[0](c, act=ACT_TANH)
        Ks = self.fc_mlp_2[0](own, act=ACT_TANH)
        return F.attention(self.fc_1(Qs), self.fc_2(Ks), Ks, 1, 1.0, residual=Qs)  # unscaled softmax(QK^T)


# ------------------------------------------------------------------------------------------------------
# routing layers
# ------------------------------------------------------------------------------------------------------
class _RoutingLayer(D2RModule):
    """One DynamicInteraction layer; ``swap`` selects the reversed (image-branch) variant."""

    first_layer = False
    swap = False

    def __init__(self, args, num_cell, num_out_path):
        super().__init__()
        # The reference hard-indexes six cells (models/DynamicInteraction.py:39-48: any other num_cell crashes there).
        # num_cell in 2..5 is the declared-subset EXTENSION of SURVEY.md section 8c (BASELINE configs[4]: 4 cells): the
        # layer owns the first num_cell cells of CELL_ORDER (and only their parameters), normalises the path
        # probabilities over them, and the final layer gates at self.threshold / self.num_cell.
        if not 2 <= num_cell <= 6:
            raise ValueError(f"num_cell={num_cell}: a routing layer has 2..6 cells ({', '.join(CELL_ORDER)} in this order)")
        self.num_cell, self.num_out_path = num_cell, num_out_path
        self.threshold, self.eps = 1e-4, 1e-8
        self.cell_names = CELL_ORDER[:num_cell]
        classes = dict(ric=RectifiedIdentityCell, imrc=IntraModelReasoningCell, glac=GlobalLocalAlignmentCell,
                       cmrc=CrossModalRefinementCell, crcmc=ContextRichCrossModalCell, gesc=GlobalEnhancedSemanticCell)
        for name in ("ric", "imrc", "glac", "cmrc", "crcmc", "gesc"):  # registration order of the reference (:28-35)
            if name in self.cell_names:
                setattr(self, name, classes[name](args, num_out_path))

    def _cells(self):
        return [getattr(self, n) for n in self.cell_names]

    def _fusion_members(self):
        return {"r0": [c.router.mlp[0] for c in self._cells()], "r2": [c.router.mlp[2] for c in self._cells()]}

    def _route(self, refs: List[torch.Tensor], other: torch.Tensor, kv=None):
        """kv: {cell name: its [B, Lk, 1536] block of the module-wide k|v projection} or None."""
        cells = self._cells()
        r0, r2 = self._fused_linear("r0"), self._fused_linear("r2")
        if r0 is not None:  # the six routers as two GEMMs (fp32): hidden [B, 6*hid] then gates [B, 6*P]
            if self.first_layer:  # all six read the same pooled vector: one plain GEMM, N = 6*hid
                h = r0(F.mean_pool([refs[0]])[0], torch.float32, act=ACT_RELU)
            else:
                h = r0.grouped(F.mean_pool(refs), torch.float32, act=ACT_RELU, x_gm=True)
            gb = r2.grouped(h, torch.float32, act=ACT_TANH_RELU)  # [B, 6*P]
            G = gb.view(gb.shape[0], self.num_cell, self.num_out_path)  # [B,ncell,P]: the layout K8 reads (no copy)
        else:
            if self.first_layer:  # six routers read the same tensor: pool once
                pooled = F.mean_pool([refs[0]])[0]
                gates = [c.router.gate_from_pooled(pooled) for c in cells]
            else:
                pooled = F.mean_pool(refs)  # [6,B,768] in one launch
                gates = [c.router.gate_from_pooled(pooled[j]) for j, c in enumerate(cells)]
            G = torch.stack(gates, dim=1)  # fp32 [B,ncell,P]
        embs = [c(refs[j], other, None if kv is None else kv.get(self.cell_names[j])) for j, c in enumerate(cells)]
        if self.num_out_path == 1:
            probs, outs = F.route_aggregate(G, *embs, refs=refs[1:])
        else:
            probs, outs = F.route_aggregate(G, *embs)
        return outs, probs


class DynamicInteraction_Layer0(_RoutingLayer):
    first_layer = True

    def forward(self, text, image, kv=None):
        own, other = (image, text) if self.swap else (text, image)
        return self._route([own] * self.num_cell, other, kv)


class DynamicInteraction_Layer(_RoutingLayer):
    def forward(self, ref_wrd, text, image, kv=None):
        return self._route(list(ref_wrd), text if self.swap else image, kv)


class Reversed_DynamicInteraction_Layer0(DynamicInteraction_Layer0):
    swap = True


class Reversed_DynamicInteraction_Layer(DynamicInteraction_Layer):
    swap = True


class _InteractionBase(D2RModule):
    layer0_cls = DynamicInteraction_Layer0
    layer_cls = DynamicInteraction_Layer

    def __init__(self, args, num_layer_routing=3, num_cells=4, path_hid=128):
        super().__init__()
        if num_layer_routing < 2:
            raise ValueError("DR_step must be >= 2")
        self.num_cells = num_cells
        self.dynamic_itr_l0 = self.layer0_cls(args, num_cells, num_cells)
        self.dynamic_itr_l1 = nn.ModuleList(
            [self.layer_cls(args, num_cells, num_cells) for _ in range(num_layer_routing - 2)])
        self.dynamic_itr_l2 = self.layer_cls(args, num_cells, 1)
        total_paths = num_cells ** 2 * (num_layer_routing - 1) + num_cells
        self.path_mapping = Linear(total_paths, path_hid)  # dead parameter (models/InteractionModule.py:19)
        self.bn = nn.BatchNorm1d(args.embed_size)  # dead module (:20)

    KV_CELLS = ("glac", "cmrc", "crcmc")  # the cells that project `other` to keys and values (models/Cells.py:147,85,238)

    def _kv_alignments(self):
        """Every CrossModalAlignment of the module, layer by layer: they all project the SAME tensor (`other` is the raw
        encoder output in every layer, models/DynamicInteraction.py:95-102)."""
        out = []
        for layer in [self.dynamic_itr_l0, *self.dynamic_itr_l1, self.dynamic_itr_l2]:
            for name in self.KV_CELLS:
                if name in layer.cell_names:
                    cell = getattr(layer, name)
                    out.append(cell.refine.CrossModalAlignment if name == "cmrc" else cell.CrossModalAlignment)
        return out

    def _fusion_members(self):
        # one [n * 1536, 768] run of every key | value projection of `other` (all cells, all layers): ONE GEMM per branch
        # (N = 13,824 at DR_step 3) instead of nine with N = 1,536, and one dX / dW product each in the backward pass
        members = [lin for cma in self._kv_alignments() for lin in (cma.key, cma.value)]
        return {"kv_all": members} if len(members) > 2 else {}

    def _bundle_spec(self):
        """{RL name: (weight leaf, bias leaf)} per routing layer for functional.InteractionBundle (None: not prepared)."""
        def wb(lin):
            return (lin.weight, lin.bias)

        def fused(owner, key=None):
            fz = owner._fused_linear(key)
            if fz is None:
                raise LookupError
            return (fz.weight, fz.bias)

        layers = []
        for layer in [self.dynamic_itr_l0, *self.dynamic_itr_l1, self.dynamic_itr_l2]:
            s = {"R0": fused(layer, "r0"), "R2": fused(layer, "r2")}
            names = layer.cell_names
            if "glac" in names:
                g = layer.glac
                s.update(GLAC_Q=wb(g.CrossModalAlignment.query), GLAC_KV=fused(g.CrossModalAlignment), GLAC_LOC=wb(g.fc_sim_tranloc),
                         GLAC_FC1=wb(g.fc_1), GLAC_TPOOL=wb(g.text_cls_pool.dense), GLAC_IPOOL=wb(g.image_cls_pool.dense),
                         GLAC_GLO=wb(g.fc_sim_tranglo), GLAC_FC2=wb(g.fc_2), GLAC_SAFW=wb(g.SAF_module.attn_sim_w))
                bn = g.SAF_module.bn
                s["bn"] = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
            if "imrc" in names:
                sa = layer.imrc.sa
                s.update(IMRC_QKV=fused(sa.att_layer), IMRC_FC1=wb(sa.feed_forward_layer.fc1), IMRC_FC2=wb(sa.feed_forward_layer.fc2))
            if "cmrc" in names:
                r = layer.cmrc.refine
                s.update(CMRC_Q=wb(r.CrossModalAlignment.query), CMRC_KV=fused(r.CrossModalAlignment), CMRC_SCALE=wb(r.fc_scale),
                         CMRC_SHIFT=wb(r.fc_shift), CMRC_FC1=wb(r.fc_1), CMRC_FC2=wb(r.fc_2))
            if "crcmc" in names:
                c = layer.crcmc
                s.update(CRCMC_Q=wb(c.CrossModalAlignment.query), CRCMC_KV=fused(c.CrossModalAlignment), CRCMC_MLP1=wb(c.fc_mlp_1[0]),
                         CRCMC_MLP2=wb(c.fc_mlp_2[0]), CRCMC_FC1=wb(c.fc_1), CRCMC_FC2=wb(c.fc_2))
            if "gesc" in names:
                e = layer.gesc
                s.update(GESC_TPOOL=wb(e.text_cls_pool.dense), GESC_IPOOL=wb(e.image_cls_pool.dense), GESC_MLP0=wb(e.fc_mlp[0]),
                         GESC_MLP2=wb(e.fc_mlp[2]))
            layers.append(s)
        return layers

    def _bundle(self, own, other):
        """The cached InteractionBundle when the one-call path applies (bf16 model prepared by ParamStore, token counts the
        fused attention cores support), else None -> the layers run op by op."""
        if not COMPOSITE_ROUTING or self.cdtype not in F.LOWP or own.dtype != self.cdtype or not own.is_cuda:
            return None
        b = getattr(self, "_bundle_cache", None)
        l0 = self.dynamic_itr_l0
        probe = l0._fused_linear("r0")
        if probe is None or getattr(l0.ric.router.mlp[0].weight, "_d2r_grad", None) is None:
            return None
        if b is None or b.key[:2] != (probe.weight.data_ptr(), probe.weight._d2r_grad.data_ptr()):
            try:
                spec = self._bundle_spec()
                imrc = getattr(l0, "imrc", None)
                kv = self._fused_linear("kv_all")
                if kv is None:  # a single alignment in the whole module (two cells, DR_step 2 ...): its own k | v pair is the run
                    cmas = self._kv_alignments()
                    kv = cmas[0]._fused_linear() if cmas else None
                b = F.InteractionBundle(spec, self.num_cells, l0.ric.router.mlp[0].out_features,
                                        imrc.sa.h if imrc is not None else 16,
                                        imrc.sa.feed_forward_layer.fc1.out_features if imrc is not None else E,
                                        kv_all=None if kv is None else (kv.weight, kv.bias))
            except (LookupError, F._lib.D2RError):
                return None
            self._bundle_cache = b
        return b if b.supports(own, other) else None

    def forward(self, text, image):
        own, other = (image, text) if self.dynamic_itr_l0.swap else (text, image)
        bundle = self._bundle(own, other)
        if bundle is not None:  # K16: the whole module as one C call per direction
            out, paths = F.interaction(own, other, bundle, self.training)
            if self.training:  # BatchNorm1d bookkeeping of the GLAC cells (the running statistics are updated in the call)
                counters = [layer.glac.SAF_module.bn.num_batches_tracked for layer in [self.dynamic_itr_l0, *self.dynamic_itr_l1, self.dynamic_itr_l2]
                            if "glac" in layer.cell_names]
                if counters:
                    torch._foreach_add_(counters, 1)  # one launch for the module's counters instead of one each
            paths = F.gather_batch(paths)  # (global-batch-exact data parallelism: the paths of every rank's samples)
            return [out], F.matmul_nt(paths, paths)
        B = text.shape[0]
        layers = [self.dynamic_itr_l0, *self.dynamic_itr_l1, self.dynamic_itr_l2]
        kvs = [None] * len(layers)
        fz = self._fused_linear("kv_all")
        if fz is not None:  # ONE k|v projection of `other` for every alignment cell of every layer (as the one-call path)
            blocks = iter(F.split_columns(fz(other, self.cdtype), len(self._kv_alignments())))
            kvs = [{name: next(blocks) for name in self.KV_CELLS if name in layer.cell_names} for layer in layers]
        refs, p0 = self.dynamic_itr_l0(text, image, kvs[0])
        plist = [p0.reshape(B, -1)]
        for layer, kv in zip(self.dynamic_itr_l1, kvs[1:-1]):
            refs, pm = layer(refs, text, image, kv)
            plist.append(pm.reshape(B, -1))
        out, pf = self.dynamic_itr_l2(refs, text, image, kvs[-1])
        plist.append(pf.reshape(B, -1))
        paths = F.gather_batch(torch.cat(plist, dim=-1))  # fp32 [B, 36(DR-1)+6]; DR_step=2 (extension): cat(l0, l2)
        return out, F.matmul_nt(paths, paths)


class InteractionModule(_InteractionBase):
    pass


class Reversed_InteractionModule(_InteractionBase):
    layer0_cls = Reversed_DynamicInteraction_Layer0
    layer_cls = Reversed_DynamicInteraction_Layer


# ------------------------------------------------------------------------------------------------------
# encoders
# ------------------------------------------------------------------------------------------------------
EARLY_SELF_LAYERS = True
INTERLEAVE_ENCODERS = True  # issue the two encoders layer by layer in alternation
COMPOSITE_LAYERS = True  # whole encoder layers as one C call (bf16 only)
COMPOSITE_HEAD = True  # Block fusion + fc + cross entropy + loss as one C call each way (fp32)
COMPOSITE_ROUTING = True  # whole interaction modules as one C call (bf16 only)


def _layer_bundle(layer, x):
    """The cached LayerBundle of a BertLayer / CLIPEncoderLayer when the one-call path applies (bf16 model prepared by
    ParamStore, fused q|k|v, attention shape supported by the fused core), else None -> the op-by-op path."""
    if not COMPOSITE_LAYERS or layer.cdtype not in F.LOWP or x.dtype != layer.cdtype or not x.is_cuda:
        return None
    att = layer.attention.self if hasattr(layer, "attention") else layer.self_attn
    fz = att._fused_linear()
    if fz is None or getattr(fz.weight, "_d2r_lp", None) is None:
        return None
    b = getattr(layer, "_bundle_cache", None)
    if b is None or b.key != (fz.weight._d2r_lp.data_ptr(), fz.weight._d2r_grad.data_ptr()):
        b = layer._bundle()
        layer._bundle_cache = b
    return b if b.supports(x) else None


class BertSelfAttention(D2RModule):
    def __init__(self, config):
        super().__init__()
        self.num_attention_heads = config.num_attention_heads
        self.query, self.key, self.value = (Linear(config.hidden_size, config.hidden_size) for _ in range(3))
        self.p_drop = config.attention_probs_dropout_prob

    def _fusion_members(self):
        return [self.query, self.key, self.value]


class BertSelfOutput(D2RModule):
    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertAttention(D2RModule):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)


class BertIntermediate(D2RModule):
    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.intermediate_size)
        self.fusion_dense = Linear(config.hidden_size, config.intermediate_size)  # dead (modeling_unimo.py:447)


class BertOutput(D2RModule):
    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertLayer(D2RModule):
    """Post-LN BERT layer (models/modeling_unimo.py:473-512)."""

    def __init__(self, config):
        super().__init__()
        if config.hidden_act != "gelu":
            raise NotImplementedError(f"BERT hidden_act={config.hidden_act!r} (only 'gelu' is implemented)")
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self.p_hidden = config.hidden_dropout_prob

    def _bundle(self):
        sa, so = self.attention.self, self.attention.output
        fz = sa._fused_linear()
        return F.LayerBundle(pre_ln=False, act=ACT_GELU, H=sa.num_attention_heads, eps=so.LayerNorm.eps,
                             qkv=(fz.weight, fz.bias), o=(so.dense.weight, so.dense.bias),
                             fc1=(self.intermediate.dense.weight, self.intermediate.dense.bias),
                             fc2=(self.output.dense.weight, self.output.dense.bias),
                             ln1=(so.LayerNorm.weight, so.LayerNorm.bias),
                             ln2=(self.output.LayerNorm.weight, self.output.LayerNorm.bias))

    def forward(self, x, key_mask=None):
        sa = self.attention.self
        # the whole layer is one C call, training-time dropout (models/modeling_unimo.py:388,413,468) included: the
        # probabilities are masked inside the fused attention core, the two dense outputs by one in-place pass each
        p_att = sa.p_drop if self.training else 0.0
        p_hid = self.p_hidden if self.training else 0.0
        bundle = _layer_bundle(self, x)
        if bundle is not None:
            return F.encoder_layer(x, bundle, key_mask, p_att, p_hid)
        H = sa.num_attention_heads
        scale = 1.0 / math.sqrt(x.shape[-1] // H)
        fz = sa._fused_linear()
        if fz is not None:
            ctx = F.attention_qkv(fz(x, self.cdtype), H, scale, mask=key_mask, p_drop=p_att)
        else:
            ctx = F.attention(sa.query(x), sa.key(x), sa.value(x), H, scale, mask=key_mask, p_drop=p_att)
        if p_hid > 0.0:
            a = self.attention.output.LayerNorm(F.dropout(self.attention.output.dense(ctx), p_hid, True, residual=x))
            h = self.intermediate.dense(a, act=ACT_GELU)
            return self.output.LayerNorm(F.dropout(self.output.dense(h), p_hid, True, residual=a))
        a = self.attention.output.LayerNorm(self.attention.output.dense(ctx, residual=x))
        h = self.intermediate.dense(a, act=ACT_GELU)
        return self.output.LayerNorm(self.output.dense(h, residual=a))


class CLIPAttention(D2RModule):
    def __init__(self, config):
        super().__init__()
        self.num_heads = config.num_attention_heads
        d = config.hidden_size
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = Linear(d, d), Linear(d, d), Linear(d, d), Linear(d, d)
        self.p_drop = config.attention_dropout

    def _fusion_members(self):
        return [self.q_proj, self.k_proj, self.v_proj]


class CLIPMLP(D2RModule):
    def __init__(self, config):
        super().__init__()
        if config.hidden_act != "quick_gelu":
            raise NotImplementedError(f"CLIP hidden_act={config.hidden_act!r} (only 'quick_gelu' is implemented)")
        self.fc1 = Linear(config.hidden_size, config.intermediate_size)
        self.fc2 = Linear(config.intermediate_size, config.hidden_size)


class CLIPEncoderLayer(D2RModule):
    """Pre-LN ViT layer with quick_gelu (models/modeling_unimo.py:222-268)."""

    def __init__(self, config):
        super().__init__()
        self.self_attn = CLIPAttention(config)
        self.layer_norm1 = LayerNorm(config.hidden_size, eps=1e-5)  # the reference builds nn.LayerNorm(dim): eps 1e-5
        self.mlp = CLIPMLP(config)
        self.layer_norm2 = LayerNorm(config.hidden_size, eps=1e-5)

    def _bundle(self):
        at = self.self_attn
        fz = at._fused_linear()
        return F.LayerBundle(pre_ln=True, act=ACT_QUICK_GELU, H=at.num_heads, eps=self.layer_norm1.eps,
                             qkv=(fz.weight, fz.bias), o=(at.out_proj.weight, at.out_proj.bias),
                             fc1=(self.mlp.fc1.weight, self.mlp.fc1.bias), fc2=(self.mlp.fc2.weight, self.mlp.fc2.bias),
                             ln1=(self.layer_norm1.weight, self.layer_norm1.bias),
                             ln2=(self.layer_norm2.weight, self.layer_norm2.bias))

    def forward(self, x):
        at = self.self_attn
        p_att = at.p_drop if self.training else 0.0  # attention_dropout (models/modeling_unimo.py:204), 0 by default
        bundle = _layer_bundle(self, x)
        if bundle is not None:
            return F.encoder_layer(x, bundle, None, p_att, 0.0)
        h = self.layer_norm1(x)
        d = x.shape[-1] // at.num_heads
        fz = at._fused_linear()
        if fz is not None:
            ctx = F.attention_qkv(fz(h, self.cdtype), at.num_heads, d ** -0.5, p_drop=p_att)
        else:
            ctx = F.attention(at.q_proj(h), at.k_proj(h), at.v_proj(h), at.num_heads, d ** -0.5, p_drop=p_att)
        x = at.out_proj(ctx, residual=x)
        h = self.mlp.fc1(self.layer_norm2(x), act=ACT_QUICK_GELU)
        return self.mlp.fc2(h, residual=x)


class _PatchEmbedding(nn.Module):
    def __init__(self, embed_dim, patch):
        super().__init__()
        bound = 1.0 / math.sqrt(3 * patch * patch)
        self.weight = nn.Parameter(torch.empty(embed_dim, 3, patch, patch).uniform_(-bound, bound))


class _Embedding(nn.Module):
    def __init__(self, n, dim, padding_idx=None):
        super().__init__()
        w = torch.randn(n, dim)
        if padding_idx is not None:
            w[padding_idx].zero_()
        self.weight = nn.Parameter(w)


class CLIPVisionEmbeddings(D2RModule):
    def __init__(self, config):
        super().__init__()
        self.embed_dim, self.image_size, self.patch_size = config.hidden_size, config.image_size, config.patch_size
        self.class_embedding = nn.Parameter(torch.randn(self.embed_dim))
        self.patch_embedding = _PatchEmbedding(self.embed_dim, self.patch_size)
        self.num_patches = (self.image_size // self.patch_size) ** 2
        self.num_positions = self.num_patches + 1
        self.position_embedding = _Embedding(self.num_positions, self.embed_dim)
        self.register_buffer("position_ids", torch.arange(self.num_positions).expand((1, -1)))

    def forward(self, pixel_values):
        w = self.patch_embedding.weight
        return F.clip_embed(pixel_values, w, _compute_weight(w, self.cdtype), self.class_embedding,
                            self.position_embedding.weight, self.patch_size)


class BertEmbeddings(D2RModule):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = _Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = _Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = _Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.p_drop = config.hidden_dropout_prob
        if getattr(config, "position_embedding_type", "absolute") != "absolute":
            raise NotImplementedError("only absolute position embeddings")
        if config.pad_token_id != 0:
            raise NotImplementedError("pad_token_id must be 0")
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))

    def forward(self, input_ids, token_type_ids):
        x = F.bert_embed(input_ids, token_type_ids, self.word_embeddings.weight, self.position_embeddings.weight,
                         self.token_type_embeddings.weight, self.cdtype)
        return F.dropout(self.LayerNorm(x), self.p_drop, self.training)  # models/modeling_unimo.py:329-330


class UnimoEncoder(D2RModule):
    def __init__(self, vision_config, text_config):
        super().__init__()
        assert vision_config.num_hidden_layers == text_config.num_hidden_layers  # models/modeling_unimo.py:670
        self.vision_layers = nn.ModuleList([CLIPEncoderLayer(vision_config) for _ in range(vision_config.num_hidden_layers)])
        self.text_layer = nn.ModuleList([BertLayer(text_config) for _ in range(text_config.num_hidden_layers)])

    def run_vision(self, v):
        for layer in self.vision_layers:
            v = layer(v)
        return v

    def run_text(self, t, key_mask):
        for layer in self.text_layer:
            t = layer(t, key_mask)
        return t

    def forward(self, vision_embeds, text_embeds, key_mask):
        """12 vision layers then 12 text layers, no cross-talk (models/modeling_unimo.py:682-712)."""
        return self.run_text(text_embeds, key_mask), self.run_vision(vision_embeds)


class Block(D2RModule):
    """Bilinear Block fusion (models/XModules.py:478-555), pos_norm='before_cat', no dropout."""

    def __init__(self, input_dims, output_dim, mm_dim=1600, chunks=20, rank=15):
        super().__init__()
        self.mm_dim, self.chunks, self.rank = mm_dim, chunks, rank
        assert mm_dim % chunks == 0, "uneven chunking is not used by the model"
        self.size = mm_dim // chunks
        self.linear0, self.linear1 = Linear(input_dims[0], mm_dim), Linear(input_dims[1], mm_dim)
        self.merge_linears0 = nn.ModuleList([Linear(self.size, self.size * rank) for _ in range(chunks)])
        self.merge_linears1 = nn.ModuleList([Linear(self.size, self.size * rank) for _ in range(chunks)])
        self.linear_out = Linear(mm_dim, output_dim)

    def forward(self, x):
        """x: two fp32 [B,768] pooled vectors.  The whole fusion runs in fp32 (per-sample vectors only): the signed
        square root has an unbounded derivative at 0, which would amplify bf16 rounding into every gradient."""
        x0, x1 = self.linear0(x[0], fp32=True), self.linear1(x[1], fp32=True)
        s = self.size
        f0, f1 = self._fused_linear("m0"), self._fused_linear("m1")
        if f0 is not None:  # 20 chunk projections = one batched GEMM each
            m0, m1 = f0.grouped(x0, torch.float32), f1.grouped(x1, torch.float32)
        else:
            m0 = torch.stack([self.merge_linears0[c](x0[:, c * s:(c + 1) * s], fp32=True) for c in range(self.chunks)], dim=1)
            m1 = torch.stack([self.merge_linears1[c](x1[:, c * s:(c + 1) * s], fp32=True) for c in range(self.chunks)], dim=1)
        return self.linear_out(F.block_merge(m0, m1, self.chunks, self.rank, s), fp32=True)

    def _fusion_members(self):
        return {"m0": list(self.merge_linears0), "m1": list(self.merge_linears1)}


class UnimoModel(D2RModule):
    def __init__(self, args, vision_config, text_config, add_pooling_layer=True, num_self_layer=1):
        super().__init__()
        self.args, self.vision_config, self.text_config = args, vision_config, text_config
        if vision_config.hidden_size != E or text_config.hidden_size != E:
            raise ValueError("the routing cells are hard-wired to hidden size 768")
        self.vision_embeddings = CLIPVisionEmbeddings(vision_config)
        self.vision_pre_layrnorm = LayerNorm(E, eps=1e-5)
        self.vision_post_layernorm = LayerNorm(E, eps=1e-5)  # dead
        self.text_embeddings = BertEmbeddings(text_config)
        self.encoder = UnimoEncoder(vision_config, text_config)
        self.self_text = nn.ModuleList([BertLayer(text_config) for _ in range(num_self_layer)])
        self.text_cls_pool = BertPooler()
        self.self_vision = nn.ModuleList([CLIPEncoderLayer(vision_config) for _ in range(num_self_layer)])
        self.vision_cls_pool = BertPooler()
        self.block_fusion = Block([E, E], E)
        self.text_pool, self.vision_pool = BertPooler(), BertPooler()
        num_cells = int(getattr(args, "num_cells", 6))  # 6 = the reference (models/modeling_unimo.py:781-782); 2..5: extension
        self.itr_module = InteractionModule(args, num_layer_routing=args.DR_step, num_cells=num_cells, path_hid=128)
        self.Reversed_itr_module = Reversed_InteractionModule(args, num_layer_routing=args.DR_step, num_cells=num_cells,
                                                              path_hid=128)
        self.text_pooler = BertPooler() if add_pooling_layer else None  # dead (ingest assert needs it)
        self.use_streams = os.environ.get("D2R_STREAMS", "1") != "0"
        self._streams = None

    def _head_bundle(self, fc):
        """The cached HeadBundle when the one-call head applies (fp32 parameters with gradient sinks, the 20 + 20 merge groups
        fused by ParamStore), else None."""
        if not COMPOSITE_HEAD:
            return None
        blk = self.block_fusion
        f0, f1 = blk._fused_linear("m0"), blk._fused_linear("m1")
        w0 = blk.linear0.weight
        if f0 is None or f1 is None or getattr(w0, "_d2r_grad", None) is None or not w0.is_cuda:
            return None
        b = getattr(self, "_head_cache", None)
        if b is None or b[0] is not fc or b[1].key != (w0.data_ptr(), w0._d2r_grad.data_ptr()):
            try:
                hb = F.HeadBundle([(blk.linear0.weight, blk.linear0.bias), (blk.linear1.weight, blk.linear1.bias), (f0.weight, f0.bias),
                                   (f1.weight, f1.bias), (blk.linear_out.weight, blk.linear_out.bias), (fc.weight, fc.bias)],
                                  blk.mm_dim, blk.chunks, blk.rank)
            except F._lib.D2RError:
                return None
            b = self._head_cache = (fc, hb)
        return b[1]

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, pixel_values=None, head=None):
        """-> (pooler_output [B,768], js_loss, aux) — models/modeling_unimo.py:786-894.
        head = (fc Linear, labels): UnimoModelF's classifier and loss are computed here, together with Block, as one C call
        (aux["loss"], aux["logits"]) when the one-call head applies."""
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if token_type_ids is None:
            raise ValueError("token_type_ids is None!")  # models/modeling_unimo.py:808-809
        # additive key mask (1-m)*-10000 (:58-59): tiny integer->float plumbing on [B,L]
        key_mask = ((1.0 - attention_mask.to(torch.float32)) * -10000.0).contiguous()
        # The vision and text halves are independent until the routing modules, and the two routing branches are
        # independent of each other: run them on two HIP streams so their (individually small, <= 600-workgroup)
        # kernels overlap and fill the 256 CUs.  Autograd replays each backward op on its forward stream.  With
        # use_streams off the same ops are issued in the same order on the launching stream (bit-identical results).
        two = self.use_streams and pixel_values.is_cuda
        if two:
            main = torch.cuda.current_stream()
            if self._streams is None:
                self._streams = (torch.cuda.Stream(), torch.cuda.Stream())
                for st in self._streams:
                    F.register_compute_stream(st)  # joined at the end of every backward pass
            sT, sV = self._streams
            on_t, on_v = (lambda: torch.cuda.stream(sT)), (lambda: torch.cuda.stream(sV))
            for x in (input_ids, token_type_ids, key_mask):
                x.record_stream(sT)
            pixel_values.record_stream(sV)
            sT.wait_stream(main)
            sV.wait_stream(main)
        else:
            on_t = on_v = contextlib.nullcontext
        with on_v():
            v_enc = self.vision_pre_layrnorm(self.vision_embeddings(pixel_values))
        with on_t():
            t_enc = self.text_embeddings(input_ids, token_type_ids)
        if two and INTERLEAVE_ENCODERS:
            # The host needs about as long to enqueue a step as the GPU needs to run it: a whole encoder issued before the other
            # leaves the second stream empty for the first milliseconds of every pass.  Layer i of both encoders is issued before
            # layer i + 1 of either (autograd replays the same alternation backwards).
            for i, (lv, lt) in enumerate(zip(self.encoder.vision_layers, self.encoder.text_layer)):
                with on_v():
                    v_enc = lv(v_enc)
                with on_t():
                    t_enc = lt(t_enc, key_mask)
        else:
            with on_v():
                v_enc = self.encoder.run_vision(v_enc)
            with on_t():
                t_enc = self.encoder.run_text(t_enc, key_mask)
        # The extra self layers and cls poolers read their OWN encoder's output only: issued in front of the barrier, the shorter
        # branch (text: 128 tokens against 197) runs them while the other encoder is still busy (EARLY_SELF_LAYERS = False: behind it).
        def self_layers():
            with on_t():
                t_out = t_enc
                for layer in self.self_text:
                    t_out = layer(t_out, key_mask)
                tc = self.text_cls_pool(t_out, fp32=True)
            with on_v():
                v_out = v_enc
                for layer in self.self_vision:
                    v_out = layer(v_out)
                vc = self.vision_cls_pool(v_out, fp32=True)
            return tc, vc

        if EARLY_SELF_LAYERS:
            t_cls, v_cls = self_layers()
        if two:
            # barrier through the launching stream, then fork again (a direct sT<->sV cross wait is legal HIP but
            # crashes hipStreamEndCapture on ROCm 7.2 when the step is being captured into a hipGraph)
            v_enc.record_stream(sT)
            t_enc.record_stream(sV)
            main.wait_stream(sT)
            main.wait_stream(sV)
        t_enc, v_enc = F.stream_join(t_enc, v_enc)
        if two:
            sT.wait_stream(main)
            sV.wait_stream(main)
        if not EARLY_SELF_LAYERS:
            t_cls, v_cls = self_layers()
        with on_t():
            (emb_t,), sim_paths = self.itr_module(t_enc, v_enc)
            t_cls = F.gather_batch(t_cls)  # (global-batch-exact data parallelism: the [B,B] matrices span the global batch)
            js1 = F.js_div(sim_paths, F.matmul_nt(t_cls, t_cls))
            tp = self.text_pool(emb_t, fp32=True)
        with on_v():
            (emb_v,), rev_sim_paths = self.Reversed_itr_module(t_enc, v_enc)
            v_cls = F.gather_batch(v_cls)
            js2 = F.js_div(rev_sim_paths, F.matmul_nt(v_cls, v_cls))
            vp_ = self.vision_pool(emb_v, fp32=True)
        if two:
            main.wait_stream(sT)
            main.wait_stream(sV)
            for x in (js1, js2, tp, vp_, emb_t, emb_v, sim_paths, rev_sim_paths, t_enc, v_enc):
                x.record_stream(main)
        js_loss = F.lincomb([-self.args.weight_js_1, -self.args.weight_js_2], [js1, js2])
        aux = dict(emb_text=emb_t, emb_image=emb_v, sim_paths=sim_paths, rev_sim_paths=rev_sim_paths,
                   text_encode_out=t_enc, vision_encode_out=v_enc, text_pooled=tp, vision_pooled=vp_)
        hb = self._head_bundle(head[0]) if head is not None and tp.dtype == torch.float32 and tp.is_cuda else None
        if hb is not None:
            # the one stretch of a step where both branch streams wait for the launching stream: ~35 short launches of the
            # head's forward and backward, paced by the host when issued op by op -> one call each way
            aux["loss"], aux["logits"], pooled = F.head(tp, vp_, js_loss, head[1], hb)
        else:
            pooled = self.block_fusion([tp, vp_])
        return pooled, js_loss, aux


class UnimoModelF(D2RModule):
    """Drop-in surface: forward(input_ids, attention_mask, token_type_ids, labels, images) -> (loss, logits)."""

    def __init__(self, args, vision_config, text_config, num_classes: int = 3):
        super().__init__()
        self.args, self.vision_config, self.text_config = args, vision_config, text_config
        self.model = UnimoModel(args, vision_config, text_config)
        self.fc = Linear(text_config.hidden_size, num_classes)
        self.last_aux = None

    def forward(self, input_ids, attention_mask, token_type_ids, labels, images):
        pooled, js_loss, aux = self.model(input_ids=input_ids, attention_mask=attention_mask,
                                          token_type_ids=token_type_ids, pixel_values=images, head=(self.fc, labels))
        if "loss" in aux:  # Block, fc, cross entropy and the sum were one call inside the model
            loss, logits = aux.pop("loss"), aux.pop("logits")
        else:
            logits = self.fc(pooled, fp32=True)
            loss = F.lincomb([1.0, 1.0], [F.cross_entropy(logits, labels), js_loss])
        aux["js_loss"] = js_loss
        self.last_aux = aux
        return loss, logits
