"""Light config objects with the attribute names the model reads from HF ``BertConfig`` / ``CLIPVisionConfig``
(run.py:142-143).  Real HF config objects work unchanged (duck typing); these avoid importing transformers in
benchmarks and tests.  ``default_args`` mirrors the argparse defaults of run.py:39-84 that the model/trainer read."""
from __future__ import annotations

import types
from dataclasses import dataclass

import torch


@dataclass
class TextConfig:  # BertConfig() defaults = bert-base-uncased architecture
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    hidden_act: str = "gelu"
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    pad_token_id: int = 0
    position_embedding_type: str = "absolute"


@dataclass
class VisionConfig:  # CLIPVisionConfig() defaults = ViT-B/32 @ 224
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    image_size: int = 224
    patch_size: int = 32
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    attention_dropout: float = 0.0


def default_args(**over):
    a = types.SimpleNamespace(
        bert_name="bert-base-uncased", vit_name="clip-vit-base-patch32", num_epochs=30, device="cuda", batch_size=32,
        lr=3e-5, warmup_ratio=0.01, eval_begin_epoch=1, seed=2023, load_path=None, save_path="./output/",
        max_seq=128, alpha=0.0, margin=0.1, DR_step=3, weight_js_1=0.1, weight_js_2=0.1, embed_size=768,
        num_head_IMRC=16, hid_IMRC=768, hid_router=768, compute_dtype=torch.float32, cleanup_output=False,
        dp_overlap=False)
    for k, v in over.items():
        setattr(a, k, v)
    return a
