"""ORACLE — TEST INFRASTRUCTURE ONLY.  The table of golden cases shared by ``make_goldens.py`` (writer,
build container) and ``tests/`` (readers, everywhere)."""
from __future__ import annotations

from dataclasses import dataclass

from .d2r_oracle import OracleConfig


def small_cfg(layers=2, image_size=64, patch=32, dr=3) -> OracleConfig:
    return OracleConfig(text_layers=layers, vision_layers=layers, image_size=image_size, patch_size=patch,
                        DR_step=dr)


@dataclass(frozen=True)
class RoutingCase:
    """(Reversed_)InteractionModule alone on random own/other token tensors."""
    name: str
    reversed_branch: bool
    B: int
    Lq: int   # own-modality tokens
    Lk: int   # other-modality tokens
    DR_step: int
    router_bias: str
    train: bool = True
    seed: int = 0
    in_scale: float = 1.0


@dataclass(frozen=True)
class ModelCase:
    """Full UnimoModelF forward/backward."""
    name: str
    layers: int
    image_size: int
    patch: int
    B: int
    L: int
    DR_step: int
    router_bias: str
    train: bool = True
    seed: int = 0
    # compact fixture (full-size cases): the images are regenerated from the seed (their checksums are stored) and only
    # token 0 of the two routed embeddings is stored, so that the committed file stays far below 1 MB
    compact: bool = False

    def cfg(self) -> OracleConfig:
        return small_cfg(self.layers, self.image_size, self.patch, self.DR_step)


ROUTING_CASES = [
    RoutingCase("rt_text_init", False, 3, 8, 5, 3, "init"),
    RoutingCase("rt_text_normal", False, 3, 8, 5, 3, "normal"),
    RoutingCase("rt_text_closed", False, 2, 8, 5, 3, "closed"),
    RoutingCase("rt_text_eval", False, 3, 8, 5, 3, "normal", train=False),
    RoutingCase("rt_text_dr4", False, 2, 6, 10, 4, "normal", seed=1),
    RoutingCase("rt_img_init", True, 3, 5, 8, 3, "init"),
    RoutingCase("rt_img_normal", True, 3, 10, 16, 3, "normal", seed=2),
    RoutingCase("rt_text_ragged", False, 4, 19, 7, 3, "normal", seed=3, in_scale=0.5),
    # DR_step 8 (BASELINE.json configs[4]): the reference builds DR_step - 2 = six middle layers (models/InteractionModule.py:16,27-29);
    # about half of the paths pruned, so that closed routers and skip paths occur in the middle of the chain
    RoutingCase("rt_text_dr8", False, 2, 6, 5, 8, "normal", seed=4),
    RoutingCase("rt_img_dr8", True, 2, 5, 6, 8, "normal", seed=5),
]

MODEL_CASES = [
    ModelCase("m_l2_init", 2, 64, 32, 3, 8, 3, "init"),
    ModelCase("m_l2_normal", 2, 64, 32, 3, 8, 3, "normal"),
    ModelCase("m_l2_eval", 2, 64, 32, 3, 8, 3, "normal", train=False),
    ModelCase("m_l2_dr4", 2, 96, 32, 2, 12, 4, "normal", seed=1),
    ModelCase("m_l12", 12, 96, 32, 2, 16, 3, "normal", seed=2),
    ModelCase("m_l2_dr8", 2, 64, 32, 2, 8, 8, "normal", seed=6),
    # BASELINE.json configs[0] ("C1") at FULL size: MVSA-Single as the reference's own run.py builds it - batch 4, max_seq 64,
    # 224x224 images at patch 32 (49 patches + CLS = 50 image tokens), 12+12 encoder layers, DR_step 3 (run.py:70), default
    # router initialisation (every path open), ragged text lengths
    ModelCase("m_c1", 12, 224, 32, 4, 64, 3, "init", seed=5, compact=True),
]
