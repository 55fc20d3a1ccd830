"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A CPU, fp32, plain-PyTorch *functional restatement* of the D2R dual-branch dynamic-routing
forward (SorF520/D2R), written from the maths in SURVEY.md Appendix A.  Weights are taken from a
flat ``dict[str, Tensor]`` whose keys are exactly the reference's ``state_dict`` names, so the same
seeded weights can be loaded into (i) the reference itself (``oracle/ref_loader.py``, this container
only), (ii) this oracle and (iii) the HIP product (``d2r_amd``).  Backward is obtained with autograd on
the dict's leaf tensors.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this file.  ``d2r_amd`` never does.

Parity status: PINNED.  ``oracle/make_goldens.py`` runs the real reference (imported from
/root/reference in the build container) on seeded weights/inputs, checks this restatement against it
and commits the reference's outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks
the restatement against those fixtures everywhere (no reference needed).

Reference citations (file:line under /root/reference):
  router                  models/Router.py:6-8,22-26
  RIC/IMRC/CMRC/GLAC/GESC/CRCMC cells   models/Cells.py:30-40,42-60,76-87,131-175,179-218,222-255
  IMRC body               models/SelfAttention.py:27-42,52-53,64-70
  cross-modal alignment   models/XModules.py:300-310, models/Refinement.py:105-115
  CMRC refine             models/Refinement.py:133-154
  SAF                     models/XModules.py:380-384
  routing layers          models/DynamicInteraction.py:37-69,90-134,157-189,210-254
  interaction modules     models/InteractionModule.py:22-55,75-108
  encoders + glue         models/modeling_unimo.py:87-118,136-268,272-527,649-729,786-894
  js_div / Block          models/XModules.py:32-41,454-555
  head + loss             models/unimo_model.py:149-162
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
E = 768  # hard-wired embedding width of the routing cells (models/Cells.py:140-143)

CELL_ORDER = ("ric", "glac", "imrc", "cmrc", "crcmc", "gesc")  # models/DynamicInteraction.py:41-48


# --------------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    # text encoder (BertConfig defaults = bert-base)
    vocab_size: int = 30522
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    text_layers: int = 12
    text_heads: int = 12
    text_intermediate: int = 3072
    text_ln_eps: float = 1e-12
    # vision encoder (CLIPVisionConfig defaults = ViT-B/32 @224)
    image_size: int = 224
    patch_size: int = 32
    vision_layers: int = 12
    vision_heads: int = 12
    vision_intermediate: int = 3072
    vision_ln_eps: float = 1e-5
    # routing
    DR_step: int = 3
    num_head_IMRC: int = 16
    hid_IMRC: int = 768
    hid_router: int = 768
    weight_js_1: float = 0.1
    weight_js_2: float = 0.1
    num_classes: int = 3
    # number of cells per routing layer = the first num_cells of CELL_ORDER.  The reference hard-indexes six
    # (models/DynamicInteraction.py:39-48: any other value crashes); 2..5 is the declared-subset EXTENSION of SURVEY.md
    # section 8c (BASELINE configs[4]: 4 cells) whose only oracle is this restatement: path normalisation over the
    # existing cells, num_cells outputs per first / middle layer, final-layer threshold self.threshold / self.num_cell.
    num_cells: int = 6
    # Block fusion (models/XModules.py:478-522 defaults)
    mm_dim: int = 1600
    chunks: int = 20
    rank: int = 15

    @property
    def num_image_tokens(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + 1


# --------------------------------------------------------------------------------------------------
# parameter inventory: name -> shape, in any order (load_state_dict is by name)
# --------------------------------------------------------------------------------------------------
def _lin(spec, name, out_f, in_f):
    spec[name + ".weight"] = (out_f, in_f)
    spec[name + ".bias"] = (out_f,)


def _ln(spec, name, d=E):
    spec[name + ".weight"] = (d,)
    spec[name + ".bias"] = (d,)


def _bert_layer_spec(spec, p, cfg: OracleConfig):
    for n in ("query", "key", "value"):
        _lin(spec, f"{p}.attention.self.{n}", E, E)
    _lin(spec, f"{p}.attention.output.dense", E, E)
    _ln(spec, f"{p}.attention.output.LayerNorm")
    _lin(spec, f"{p}.intermediate.dense", cfg.text_intermediate, E)
    _lin(spec, f"{p}.intermediate.fusion_dense", cfg.text_intermediate, E)  # dead (modeling_unimo.py:447)
    _lin(spec, f"{p}.output.dense", E, cfg.text_intermediate)
    _ln(spec, f"{p}.output.LayerNorm")


def _clip_layer_spec(spec, p, cfg: OracleConfig):
    for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
        _lin(spec, f"{p}.self_attn.{n}", E, E)
    _ln(spec, f"{p}.layer_norm1")
    _lin(spec, f"{p}.mlp.fc1", cfg.vision_intermediate, E)
    _lin(spec, f"{p}.mlp.fc2", E, cfg.vision_intermediate)
    _ln(spec, f"{p}.layer_norm2")


def _router_spec(spec, p, n_out, cfg):
    _lin(spec, f"{p}.router.mlp.0", cfg.hid_router, E)
    _lin(spec, f"{p}.router.mlp.2", n_out, cfg.hid_router)


def _xalign_spec(spec, p):
    for n in ("query", "key", "value", "fc_1", "fc_2"):  # fc_1/fc_2 are dead parameters
        _lin(spec, f"{p}.{n}", E, E)


def _routing_layer_spec(spec, p, n_out, cfg: OracleConfig):
    full: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    _routing_layer_spec_all(full, p, n_out, cfg)
    cells = CELL_ORDER[:cfg.num_cells]
    for k, v in full.items():  # a declared subset owns only the parameters of its cells
        if k[len(p) + 1:].split(".", 1)[0] in cells:
            spec[k] = v


def _routing_layer_spec_all(spec, p, n_out, cfg: OracleConfig):
    for c in CELL_ORDER:
        _router_spec(spec, f"{p}.{c}", n_out, cfg)
    # IMRC
    for i in range(3):
        _lin(spec, f"{p}.imrc.sa.att_layer.linears.{i}", E, E)
    _lin(spec, f"{p}.imrc.sa.feed_forward_layer.fc1", cfg.hid_IMRC, E)
    _lin(spec, f"{p}.imrc.sa.feed_forward_layer.fc2", E, cfg.hid_IMRC)
    # GLAC
    _xalign_spec(spec, f"{p}.glac.CrossModalAlignment")
    _lin(spec, f"{p}.glac.SAF_module.attn_sim_w", 1, E)
    spec[f"{p}.glac.SAF_module.bn.weight"] = (1,)
    spec[f"{p}.glac.SAF_module.bn.bias"] = (1,)
    spec[f"{p}.glac.SAF_module.bn.running_mean"] = (1,)
    spec[f"{p}.glac.SAF_module.bn.running_var"] = (1,)
    spec[f"{p}.glac.SAF_module.bn.num_batches_tracked"] = ()
    for n in ("text_cls_pool.dense", "image_cls_pool.dense", "fc_sim_tranloc", "fc_sim_tranglo", "fc_1", "fc_2"):
        _lin(spec, f"{p}.glac.{n}", E, E)
    # CMRC
    for n in ("fc_scale", "fc_shift", "fc_1", "fc_2"):
        _lin(spec, f"{p}.cmrc.refine.{n}", E, E)
    _xalign_spec(spec, f"{p}.cmrc.refine.CrossModalAlignment")
    # CRCMC
    _xalign_spec(spec, f"{p}.crcmc.CrossModalAlignment")
    for n in ("fc_mlp_1.0", "fc_mlp_2.0", "fc_1", "fc_2"):
        _lin(spec, f"{p}.crcmc.{n}", E, E)
    # GESC
    for n in ("text_cls_pool.dense", "image_cls_pool.dense", "fc_mlp.0", "fc_mlp.2"):
        _lin(spec, f"{p}.gesc.{n}", E, E)


def _interaction_module_spec(spec, p, cfg: OracleConfig):
    nc = cfg.num_cells
    _routing_layer_spec(spec, f"{p}.dynamic_itr_l0", nc, cfg)
    for i in range(cfg.DR_step - 2):
        _routing_layer_spec(spec, f"{p}.dynamic_itr_l1.{i}", nc, cfg)
    _routing_layer_spec(spec, f"{p}.dynamic_itr_l2", 1, cfg)
    total_paths = nc * nc * (cfg.DR_step - 1) + nc  # models/InteractionModule.py:18
    _lin(spec, f"{p}.path_mapping", 128, total_paths)  # dead
    _ln(spec, f"{p}.bn")  # dead BatchNorm1d(768)
    spec[f"{p}.bn.running_mean"] = (E,)
    spec[f"{p}.bn.running_var"] = (E,)
    spec[f"{p}.bn.num_batches_tracked"] = ()


def block_sizes(cfg: OracleConfig) -> List[int]:
    """get_sizes_list (models/XModules.py:454-466) for the divisible case used by the model."""
    s = (cfg.mm_dim + cfg.chunks - 1) // cfg.chunks
    sizes = [s] * cfg.chunks
    sizes[-1] -= sum(sizes) - cfg.mm_dim
    assert sum(sizes) == cfg.mm_dim and min(sizes) > 0
    return sizes


def param_spec(cfg: OracleConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Every key of ``UnimoModelF.state_dict()`` with its shape (1210 keys at DR_step=3)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    m = "model"
    ntok = cfg.num_image_tokens
    s[f"{m}.vision_embeddings.class_embedding"] = (E,)
    s[f"{m}.vision_embeddings.position_ids"] = (1, ntok)
    s[f"{m}.vision_embeddings.patch_embedding.weight"] = (E, 3, cfg.patch_size, cfg.patch_size)
    s[f"{m}.vision_embeddings.position_embedding.weight"] = (ntok, E)
    _ln(s, f"{m}.vision_pre_layrnorm")
    _ln(s, f"{m}.vision_post_layernorm")  # dead
    s[f"{m}.text_embeddings.position_ids"] = (1, cfg.max_position_embeddings)
    s[f"{m}.text_embeddings.word_embeddings.weight"] = (cfg.vocab_size, E)
    s[f"{m}.text_embeddings.position_embeddings.weight"] = (cfg.max_position_embeddings, E)
    s[f"{m}.text_embeddings.token_type_embeddings.weight"] = (cfg.type_vocab_size, E)
    _ln(s, f"{m}.text_embeddings.LayerNorm")
    for i in range(cfg.vision_layers):
        _clip_layer_spec(s, f"{m}.encoder.vision_layers.{i}", cfg)
    for i in range(cfg.text_layers):
        _bert_layer_spec(s, f"{m}.encoder.text_layer.{i}", cfg)
    _bert_layer_spec(s, f"{m}.self_text.0", cfg)
    _lin(s, f"{m}.text_cls_pool.dense", E, E)
    _clip_layer_spec(s, f"{m}.self_vision.0", cfg)
    _lin(s, f"{m}.vision_cls_pool.dense", E, E)
    # Block
    _lin(s, f"{m}.block_fusion.linear0", cfg.mm_dim, E)
    _lin(s, f"{m}.block_fusion.linear1", cfg.mm_dim, E)
    for which in (0, 1):
        for c, sz in enumerate(block_sizes(cfg)):
            _lin(s, f"{m}.block_fusion.merge_linears{which}.{c}", sz * cfg.rank, sz)
    _lin(s, f"{m}.block_fusion.linear_out", E, cfg.mm_dim)
    _lin(s, f"{m}.text_pool.dense", E, E)
    _lin(s, f"{m}.vision_pool.dense", E, E)
    _interaction_module_spec(s, f"{m}.itr_module", cfg)
    _interaction_module_spec(s, f"{m}.Reversed_itr_module", cfg)
    _lin(s, f"{m}.text_pooler.dense", E, E)  # dead
    _lin(s, "fc", cfg.num_classes, E)
    return s


INT_KEYS = ("position_ids", "num_batches_tracked")


def is_dead_param(name: str) -> bool:
    """Parameters that never receive a gradient in the reference."""
    if "fusion_dense" in name or "vision_post_layernorm" in name or ".text_pooler." in name:
        return True
    if ".path_mapping." in name or name.endswith("itr_module.bn.weight") or name.endswith("itr_module.bn.bias"):
        return True
    if "CrossModalAlignment.fc_1" in name or "CrossModalAlignment.fc_2" in name:
        return True
    return False


# --------------------------------------------------------------------------------------------------
# small maths helpers
# --------------------------------------------------------------------------------------------------
def lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def lnorm(sd, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def l2n(x, dim=-1, eps=1e-8):  # eps outside the root (models/Cells.py:23-27)
    return x / (x.pow(2).sum(dim=dim, keepdim=True).sqrt() + eps)


def l1n(x, dim, eps=1e-8):  # models/Cells.py:16-20
    return x / (x.abs().sum(dim=dim, keepdim=True) + eps)


def cls_pool(sd, p, x):  # BertPooler: tanh(W x[:,0])  (models/Cells.py:90-102)
    return torch.tanh(lin(sd, p + ".dense", x[:, 0]))


def router_gate(sd, p, x):
    """models/Router.py:22-26 — relu(tanh(W2 relu(W1 mean_tokens(x))))."""
    h = F.relu(lin(sd, p + ".router.mlp.0", x.mean(dim=-2)))
    return F.relu(torch.tanh(lin(sd, p + ".router.mlp.2", h)))


def xalign(sd, p, own, other):
    """Live part of CrossModalAlignment (models/XModules.py:300-310): single head, head-dim 768."""
    q = lin(sd, p + ".query", own)
    k = lin(sd, p + ".key", other)
    v = lin(sd, p + ".value", other)
    s = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(E)
    return torch.bmm(torch.softmax(100.0 * s, dim=-1), v)


# --------------------------------------------------------------------------------------------------
# the six cells: each returns (emb, gate); emb is [B,Lq,768] or a per-sample [B,768] broadcast
# --------------------------------------------------------------------------------------------------
class BNState:
    """Collects BatchNorm1d(1) running-stat updates so the functional oracle stays side-effect free."""

    def __init__(self, train: bool):
        self.train = train
        self.updates: Dict[str, Tensor] = {}


def _saf(sd, p, S, st: BNState):
    """AttentionFiltration (models/XModules.py:380-384) on S[B,Lq+1,768] -> [B,768]."""
    a = lin(sd, p + ".attn_sim_w", S).squeeze(-1)  # [B, Lq+1]
    w, b = sd[p + ".bn.weight"], sd[p + ".bn.bias"]
    if st.train:
        mu = a.mean()
        var_b = a.var(unbiased=False)
        n = a.numel()
        with torch.no_grad():
            mom = 0.1
            st.updates[p + ".bn.running_mean"] = (1 - mom) * sd[p + ".bn.running_mean"] + mom * mu.reshape(1)
            st.updates[p + ".bn.running_var"] = (1 - mom) * sd[p + ".bn.running_var"] + mom * (
                var_b * n / max(n - 1, 1)).reshape(1)
            st.updates[p + ".bn.num_batches_tracked"] = sd[p + ".bn.num_batches_tracked"] + 1
    else:
        mu, var_b = sd[p + ".bn.running_mean"][0], sd[p + ".bn.running_var"][0]
    a = (a - mu) / torch.sqrt(var_b + 1e-5) * w + b
    att = l1n(torch.sigmoid(a), dim=-1)  # [B, Lq+1]
    return l2n(torch.bmm(att.unsqueeze(1), S).squeeze(1), dim=-1)


def cell_ric(sd, p, own, other, st):
    return F.relu(own), router_gate(sd, p, own)


def cell_imrc(sd, p, own, other, st, heads=16):
    g = router_gate(sd, p, own)
    B, L, _ = own.shape
    dk = E // heads
    q, k, v = (lin(sd, f"{p}.sa.att_layer.linears.{i}", own).view(B, L, heads, dk).transpose(1, 2) for i in range(3))
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dk), dim=-1) @ v
    y = own + a.transpose(1, 2).reshape(B, L, E)
    f = lin(sd, p + ".sa.feed_forward_layer.fc2", F.relu(lin(sd, p + ".sa.feed_forward_layer.fc1", y)))
    return y + f, g


def cell_cmrc(sd, p, own, other, st):
    g = router_gate(sd, p, own)
    c = xalign(sd, p + ".refine.CrossModalAlignment", own, other)
    mod = own * torch.tanh(lin(sd, p + ".refine.fc_scale", c)) + lin(sd, p + ".refine.fc_shift", c)
    return lin(sd, p + ".refine.fc_2", F.relu(lin(sd, p + ".refine.fc_1", mod))) + own, g


def cell_glac(sd, p, own, other, st):
    g = router_gate(sd, p, own)
    c = xalign(sd, p + ".CrossModalAlignment", own, other)
    sl = lin(sd, p + ".fc_1", l2n(lin(sd, p + ".fc_sim_tranloc", (own - c).pow(2))))
    dg = (cls_pool(sd, p + ".text_cls_pool", own) - cls_pool(sd, p + ".image_cls_pool", other)).pow(2)
    sg = lin(sd, p + ".fc_2", l2n(lin(sd, p + ".fc_sim_tranglo", dg)))
    S = torch.cat([sg.unsqueeze(1), sl], dim=1)
    return _saf(sd, p + ".SAF_module", S, st), g  # [B,768] broadcast over Lq


def cell_crcmc(sd, p, own, other, st):
    g = router_gate(sd, p, own)
    c = xalign(sd, p + ".CrossModalAlignment", own, other)
    Qs = torch.tanh(lin(sd, p + ".fc_mlp_1.0", c))
    Ks = torch.tanh(lin(sd, p + ".fc_mlp_2.0", own))
    P = torch.softmax(torch.bmm(lin(sd, p + ".fc_1", Qs), lin(sd, p + ".fc_2", Ks).transpose(1, 2)), dim=-1)
    return Qs + torch.bmm(P, Ks), g


def cell_gesc(sd, p, own, other, st):
    g = router_gate(sd, p, own)
    a = cls_pool(sd, p + ".text_cls_pool", own)
    b = cls_pool(sd, p + ".image_cls_pool", other)
    gate = torch.softmax(lin(sd, p + ".fc_mlp.2", torch.tanh(lin(sd, p + ".fc_mlp.0", a + b))), dim=-1)
    return gate * a + (1 - gate) * b, g  # [B,768] broadcast over Lq


CELLS = {"ric": cell_ric, "glac": cell_glac, "imrc": cell_imrc, "cmrc": cell_cmrc, "crcmc": cell_crcmc,
         "gesc": cell_gesc}


def _full(emb, L):
    return emb if emb.dim() == 3 else emb.unsqueeze(1).expand(-1, L, -1)


def routing_layer(sd, p, refs: List[Tensor], other: Tensor, n_out: int, st: BNState, trace: Optional[dict] = None):
    """One DynamicInteraction layer.  ``refs`` holds 6 own-modality inputs (layer 0: the same tensor six
    times).  Returns (list of n_out tensors, path probs [B, n_out, 6])."""
    L = refs[0].shape[1]
    nc = len(refs)  # number of cells (6 in the reference)
    embs, gates = [], []
    for j, c in enumerate(CELL_ORDER[:nc]):
        e, g = CELLS[c](sd, f"{p}.{c}", refs[j], other, st)
        embs.append(_full(e, L))
        gates.append(g)  # [B, n_out]
    G = torch.stack(gates, dim=2)  # [B, n_out, nc]
    if n_out == 1:  # final layer (models/DynamicInteraction.py:104-117)
        skip = (G < 1e-4 / nc).float()  # [B,1,nc]  (self.threshold / self.num_cell, :109)
        num = sum(G[:, 0, j, None, None] * embs[j] + skip[:, 0, j, None, None] * refs[j] for j in range(nc))
        den = (skip.sum(-1) + G.sum(-1))[:, :, None]  # [B,1,1]
        out, probs = [num / den], G
        if trace is not None:
            trace[p + ".gate_mask"] = skip
    else:  # models/DynamicInteraction.py:50-67
        skip = (G.sum(-1) < 1e-4).float()  # [B, n_out]
        probs = G / (G.sum(-1, keepdim=True) + 1e-8)
        out = [sum(probs[:, i, j, None, None] * embs[j] for j in range(nc)) + skip[:, i, None, None] * embs[0]
               for i in range(n_out)]
        if trace is not None:
            trace[p + ".gate_mask"] = skip
    if trace is not None:
        trace[p + ".probs"] = probs
        trace[p + ".raw_gates"] = G
    return out, probs


def interaction_module(sd, p, own, other, dr_step, st: BNState, trace: Optional[dict] = None, num_cells: int = 6):
    """InteractionModule.forward (models/InteractionModule.py:22-55); the reversed module is the same
    function called with the modalities swapped (models/DynamicInteraction.py:157-189,210-254)."""
    refs, p0 = routing_layer(sd, p + ".dynamic_itr_l0", [own] * num_cells, other, num_cells, st, trace)
    plist = [p0.reshape(own.shape[0], -1)]
    for i in range(dr_step - 2):
        refs, pm = routing_layer(sd, f"{p}.dynamic_itr_l1.{i}", refs, other, num_cells, st, trace)
        plist.append(pm.reshape(own.shape[0], -1))
    out, pf = routing_layer(sd, p + ".dynamic_itr_l2", refs, other, 1, st, trace)
    plist.append(pf.reshape(own.shape[0], -1))
    paths = torch.cat(plist, dim=-1)  # [B, 36(DR-1)+6]
    return out[0], paths @ paths.t()


# --------------------------------------------------------------------------------------------------
# encoders
# --------------------------------------------------------------------------------------------------
def bert_layer(sd, p, x, ext_mask, cfg: OracleConfig):
    B, L, _ = x.shape
    H, dk = cfg.text_heads, E // cfg.text_heads
    q, k, v = (lin(sd, f"{p}.attention.self.{n}", x).view(B, L, H, dk).transpose(1, 2) for n in ("query", "key", "value"))
    s = q @ k.transpose(-1, -2) / math.sqrt(dk) + ext_mask
    ctx = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, E)
    a = lnorm(sd, p + ".attention.output.LayerNorm", lin(sd, p + ".attention.output.dense", ctx) + x, cfg.text_ln_eps)
    h = F.gelu(lin(sd, p + ".intermediate.dense", a))
    return lnorm(sd, p + ".output.LayerNorm", lin(sd, p + ".output.dense", h) + a, cfg.text_ln_eps)


def clip_layer(sd, p, x, cfg: OracleConfig):
    B, L, _ = x.shape
    H, dk = cfg.vision_heads, E // cfg.vision_heads
    h = lnorm(sd, p + ".layer_norm1", x, cfg.vision_ln_eps)
    q = (lin(sd, p + ".self_attn.q_proj", h) * dk ** -0.5).view(B, L, H, dk).transpose(1, 2)
    k = lin(sd, p + ".self_attn.k_proj", h).view(B, L, H, dk).transpose(1, 2)
    v = lin(sd, p + ".self_attn.v_proj", h).view(B, L, H, dk).transpose(1, 2)
    ctx = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(B, L, E)
    x = x + lin(sd, p + ".self_attn.out_proj", ctx)
    h = lnorm(sd, p + ".layer_norm2", x, cfg.vision_ln_eps)
    h = lin(sd, p + ".mlp.fc1", h)
    h = h * torch.sigmoid(1.702 * h)  # quick_gelu
    return x + lin(sd, p + ".mlp.fc2", h)


def vision_embed(sd, pixel_values, cfg: OracleConfig):
    p = "model.vision_embeddings"
    B = pixel_values.shape[0]
    pe = F.conv2d(pixel_values, sd[p + ".patch_embedding.weight"], stride=cfg.patch_size).flatten(2).transpose(1, 2)
    x = torch.cat([sd[p + ".class_embedding"].expand(B, 1, -1), pe], dim=1)
    return x + sd[p + ".position_embedding.weight"][None]


def text_embed(sd, input_ids, token_type_ids, cfg: OracleConfig):
    p = "model.text_embeddings"
    L = input_ids.shape[1]
    # padding_idx=0: the pad row receives no gradient (models/modeling_unimo.py:277)
    x = F.embedding(input_ids, sd[p + ".word_embeddings.weight"], padding_idx=0) + \
        sd[p + ".token_type_embeddings.weight"][token_type_ids]
    x = x + sd[p + ".position_embeddings.weight"][:L][None]
    return lnorm(sd, p + ".LayerNorm", x, cfg.text_ln_eps)


def js_div(p_logits, q_logits):
    """models/XModules.py:32-41."""
    p = torch.softmax(p_logits, dim=-1)
    q = torch.softmax(q_logits, dim=-1)
    logm = ((p + q) / 2).log()
    B = p.shape[0]
    kl = lambda t: torch.xlogy(t, t).sum() / B - (t * logm).sum() / B
    return (kl(p) + kl(q)) / 2


def block_fusion(sd, p, x0, x1, cfg: OracleConfig):
    """Bilinear Block fusion (models/XModules.py:523-555, pos_norm='before_cat', no dropout)."""
    a = lin(sd, p + ".linear0", x0)
    b = lin(sd, p + ".linear1", x1)
    B = a.shape[0]
    zs, off = [], 0
    for c, sz in enumerate(block_sizes(cfg)):
        m = lin(sd, f"{p}.merge_linears0.{c}", a[:, off:off + sz]) * lin(sd, f"{p}.merge_linears1.{c}", b[:, off:off + sz])
        z = m.view(B, cfg.rank, sz).sum(1)
        z = torch.sqrt(F.relu(z)) - torch.sqrt(F.relu(-z))
        zs.append(F.normalize(z, p=2, dim=1))
        off += sz
    return lin(sd, p + ".linear_out", torch.cat(zs, dim=1))


# --------------------------------------------------------------------------------------------------
# full model
# --------------------------------------------------------------------------------------------------
def encode(sd, cfg: OracleConfig, input_ids, attention_mask, token_type_ids, images):
    """Embeddings + 12+12 encoder layers (models/modeling_unimo.py:798-828)."""
    if token_type_ids is None:
        raise ValueError("token_type_ids is None!")  # models/modeling_unimo.py:808-809
    v = lnorm(sd, "model.vision_pre_layrnorm", vision_embed(sd, images, cfg), cfg.vision_ln_eps)
    ext = (1.0 - attention_mask[:, None, None, :].to(torch.long)) * -10000.0  # :58-59
    t = text_embed(sd, input_ids, token_type_ids, cfg)
    for i in range(cfg.vision_layers):
        v = clip_layer(sd, f"model.encoder.vision_layers.{i}", v, cfg)
    for i in range(cfg.text_layers):
        t = bert_layer(sd, f"model.encoder.text_layer.{i}", t, ext, cfg)
    return t, v, ext


def forward(sd: Dict[str, Tensor], cfg: OracleConfig, input_ids, attention_mask, token_type_ids, labels, images,
            train: bool = True, trace: Optional[dict] = None):
    """UnimoModelF.forward (models/unimo_model.py:149-162) -> (loss, logits, aux)."""
    st = BNState(train)
    t_enc, v_enc, ext = encode(sd, cfg, input_ids, attention_mask, token_type_ids, images)
    t_cls = cls_pool(sd, "model.text_cls_pool", bert_layer(sd, "model.self_text.0", t_enc, ext, cfg))
    v_cls = cls_pool(sd, "model.vision_cls_pool", clip_layer(sd, "model.self_vision.0", v_enc, cfg))
    out_t, sim_p = interaction_module(sd, "model.itr_module", t_enc, v_enc, cfg.DR_step, st, trace, cfg.num_cells)
    out_v, sim_pr = interaction_module(sd, "model.Reversed_itr_module", v_enc, t_enc, cfg.DR_step, st, trace, cfg.num_cells)
    js = -cfg.weight_js_1 * js_div(sim_p, t_cls @ t_cls.t()) - cfg.weight_js_2 * js_div(sim_pr, v_cls @ v_cls.t())
    pooled = block_fusion(sd, "model.block_fusion", cls_pool(sd, "model.text_pool", out_t),
                          cls_pool(sd, "model.vision_pool", out_v), cfg)
    logits = lin(sd, "fc", pooled)
    loss = F.cross_entropy(logits, labels.long()) + js
    aux = dict(js_loss=js, sim_paths=sim_p, rev_sim_paths=sim_pr, emb_text=out_t, emb_image=out_v,
               text_encode_out=t_enc, vision_encode_out=v_enc, pooled=pooled, bn_updates=st.updates)
    return loss, logits, aux


# --------------------------------------------------------------------------------------------------
# seeded, name-keyed weights and synthetic batches (shared fixture protocol)
# --------------------------------------------------------------------------------------------------
def _crc(name: str) -> int:
    import zlib
    return zlib.crc32(name.encode())


def interaction_spec(cfg: OracleConfig, prefix: str = "") -> "OrderedDict[str, Tuple[int, ...]]":
    """Keys of one (Reversed_)InteractionModule alone; ``prefix`` is '' for a bare module."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    _interaction_module_spec(s, "X", cfg)
    return OrderedDict((prefix + k[2:], v) for k, v in s.items())


def seeded_state_dict(cfg: OracleConfig, seed: int = 0, router_bias: str = "init",
                      std: float = 0.02, spec=None, seed_prefix: str = "") -> "OrderedDict[str, Tensor]":
    """Deterministic weights keyed by parameter NAME (independent of construction order).

    router_bias: 'init' -> 1.5 (Router.py:20, all paths open); 'normal' -> N(0,1) (about half of the
    paths pruned); 'closed' -> -5 (everything pruned: skip path and +1e-8 division)."""
    sd: "OrderedDict[str, Tensor]" = OrderedDict()
    for name, shape in (param_spec(cfg) if spec is None else spec).items():
        g = torch.Generator().manual_seed((_crc(seed_prefix + name) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "position_ids":
            sd[name] = torch.arange(shape[1]).expand(1, -1).clone()
        elif leaf == "num_batches_tracked":
            sd[name] = torch.zeros((), dtype=torch.long)
        elif leaf == "running_mean":
            sd[name] = 0.05 * torch.randn(shape, generator=g)
        elif leaf == "running_var":
            sd[name] = 1.0 + 0.1 * torch.rand(shape, generator=g)
        elif "LayerNorm" in name or "layer_norm" in name or "layrnorm" in name or "layernorm" in name or ".bn." in name:
            sd[name] = (1.0 + 0.05 * torch.randn(shape, generator=g)) if leaf == "weight" else 0.02 * torch.randn(shape, generator=g)
        elif name.endswith("router.mlp.2.bias"):
            if router_bias == "init":
                sd[name] = torch.full(shape, 1.5)
            elif router_bias == "normal":
                sd[name] = torch.randn(shape, generator=g)
            elif router_bias == "closed":
                sd[name] = torch.full(shape, -5.0)
            else:
                raise ValueError(router_bias)
        elif leaf == "bias":
            sd[name] = 0.02 * torch.randn(shape, generator=g)
        elif name.endswith("SAF_module.attn_sim_w.weight"):
            r = math.sqrt(6.0) / math.sqrt(E + 1)
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * r
        else:
            sd[name] = std * torch.randn(shape, generator=g)
    return sd


def synthetic_batch(cfg: OracleConfig, B: int, L: int, seed: int = 0, ragged: bool = True):
    """(input_ids, attention_mask, token_type_ids, labels, images) per SURVEY.md §8d."""
    g = torch.Generator().manual_seed(1000 + seed)
    ids = torch.randint(1000, 30000, (B, L), generator=g)
    ids[:, 0] = 101
    mask = torch.ones(B, L, dtype=torch.long)
    if ragged:
        lens = torch.randint(max(L // 4, 1), L + 1, (B,), generator=g)
        for b in range(B):
            mask[b, lens[b]:] = 0
            ids[b, lens[b]:] = 0
    tt = torch.zeros(B, L, dtype=torch.long)
    images = torch.randn(B, 3, cfg.image_size, cfg.image_size, generator=g)
    labels = torch.randint(0, cfg.num_classes, (B,), generator=g)
    return ids, mask, tt, labels, images
