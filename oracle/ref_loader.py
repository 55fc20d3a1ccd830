"""ORACLE — TEST INFRASTRUCTURE ONLY (build container only; /root/reference does not exist on the GPU box).

Imports the real reference (SorF520/D2R, read-only at /root/reference) in-process so that
``oracle/make_goldens.py`` can (i) validate ``oracle/d2r_oracle.py`` against it and (ii) write the
reference's own outputs to ``tests/golden/``.  Nothing is copied from the reference: it is imported
where it lies.  Two accommodations, both outside the reference tree:

* ``models/modeling_unimo.py:8-10`` imports ``apply_chunking_to_forward`` from
  ``transformers.modeling_utils`` (4.30 layout); transformers 5.x keeps it in ``pytorch_utils`` — we set
  the attribute on the module before importing ``models.*``.
* the cells call ``BertConfig.from_pretrained(args.bert_name)`` / ``CLIPConfig.from_pretrained(...)`` in
  their constructors (``models/Cells.py:136-139,190-191,228``; ``models/Refinement.py:131``); we point
  them at local directories holding default ``config.json`` files (written under ``oracle/_cfg``).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"
_CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cfg")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models"))


def _prepare():
    sys.dont_write_bytecode = True
    os.environ.setdefault("HF_HUB_OFFLINE", "1")
    import transformers.modeling_utils as mu
    if not hasattr(mu, "apply_chunking_to_forward"):
        from transformers.pytorch_utils import apply_chunking_to_forward
        mu.apply_chunking_to_forward = apply_chunking_to_forward
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from transformers import BertConfig, CLIPConfig
    bert_dir, clip_dir = os.path.join(_CFG_DIR, "bert"), os.path.join(_CFG_DIR, "clip")
    if not os.path.exists(os.path.join(bert_dir, "config.json")):
        os.makedirs(bert_dir, exist_ok=True)
        BertConfig().save_pretrained(bert_dir)
    if not os.path.exists(os.path.join(clip_dir, "config.json")):
        os.makedirs(clip_dir, exist_ok=True)
        CLIPConfig().save_pretrained(clip_dir)
    return bert_dir, clip_dir


def make_args(cfg, device="cpu"):
    bert_dir, clip_dir = _prepare()
    return types.SimpleNamespace(
        bert_name=bert_dir, vit_name=clip_dir, device=device, DR_step=cfg.DR_step,
        weight_js_1=cfg.weight_js_1, weight_js_2=cfg.weight_js_2, embed_size=768,
        num_head_IMRC=cfg.num_head_IMRC, hid_IMRC=cfg.hid_IMRC, hid_router=cfg.hid_router, alpha=0, margin=0.1,
        raw_feature_norm_CMRC="clipped_l2norm", lambda_softmax_CMRC=4.0,
        lr=3e-5, warmup_ratio=0.01, num_epochs=1, batch_size=4, eval_begin_epoch=1, load_path=None,
        save_path=None, max_seq=128)


def hf_configs(cfg):
    """HF config objects equivalent to an OracleConfig (dropout forced to 0 — SURVEY.md §8c)."""
    _prepare()
    from transformers import BertConfig, CLIPVisionConfig
    tc = BertConfig(vocab_size=cfg.vocab_size, num_hidden_layers=cfg.text_layers,
                    num_attention_heads=cfg.text_heads, intermediate_size=cfg.text_intermediate,
                    max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
                    layer_norm_eps=cfg.text_ln_eps, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    vc = CLIPVisionConfig(image_size=cfg.image_size, patch_size=cfg.patch_size, num_hidden_layers=cfg.vision_layers,
                          num_attention_heads=cfg.vision_heads, intermediate_size=cfg.vision_intermediate,
                          layer_norm_eps=cfg.vision_ln_eps, attention_dropout=0.0)
    return vc, tc


def build_reference_model(cfg):
    """The reference's ``UnimoModelF`` (models/unimo_model.py:138-162) on CPU."""
    args = make_args(cfg)
    vc, tc = hf_configs(cfg)
    from models.unimo_model import UnimoModelF
    assert cfg.num_classes == 3, "the reference head is hard-wired to 3 classes (models/unimo_model.py:145)"
    return UnimoModelF(args, vc, tc), args


def build_reference_interaction(cfg, reversed_branch: bool):
    """The reference's (Reversed_)InteractionModule alone (models/InteractionModule.py:9-108)."""
    args = make_args(cfg)
    from models.InteractionModule import InteractionModule, Reversed_InteractionModule
    cls = Reversed_InteractionModule if reversed_branch else InteractionModule
    return cls(args, num_layer_routing=cfg.DR_step, num_cells=6, path_hid=128), args
