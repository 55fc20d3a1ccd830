"""Synthetic counterpart of the reference's data layer (processor/dataset.py:50-102): a map-style dataset that emits
the same 6-tuple ``(input_ids, input_mask, segment_ids, img_mask, label, image)`` with the same dtypes/shapes, and
a pinned-memory prefetching loader.  The real MVSA/HFM JSON+JPEG pipeline needs the HF tokenizer / CLIP processor
files and the datasets, which are not available offline; the model/trainer are agnostic to
where the 6-tuple comes from."""
from __future__ import annotations

import torch
from torch.utils.data import DataLoader, Dataset


class SyntheticMSDDataset(Dataset):
    """Deterministic per-index samples: ids ~ U{1000..29999} with [CLS]=101 / [SEP]=102 / pad 0, ragged lengths,
    images ~ N(0,1) (what CLIPProcessor's normalisation produces), labels ~ U{0..C-1}; `learnable=True` plants a
    label-dependent offset in the image so that a short training run can show a falling loss."""

    def __init__(self, n: int, max_seq: int = 128, image_size: int = 224, num_classes: int = 3, seed: int = 0,
                 learnable: bool = True, num_image_tokens: int = 50):
        self.n, self.max_seq, self.image_size, self.num_classes = n, max_seq, image_size, num_classes
        self.seed, self.learnable, self.num_image_tokens = seed, learnable, num_image_tokens

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + idx)
        L = self.max_seq
        length = int(torch.randint(max(L // 4, 3), L + 1, (1,), generator=g))
        ids = torch.zeros(L, dtype=torch.long)
        ids[:length] = torch.randint(1000, 30000, (length,), generator=g)
        ids[0], ids[length - 1] = 101, 102
        mask = torch.zeros(L, dtype=torch.long)
        mask[:length] = 1
        seg = torch.zeros(L, dtype=torch.long)
        label = int(torch.randint(0, self.num_classes, (1,), generator=g))
        image = torch.randn(3, self.image_size, self.image_size, generator=g)
        if self.learnable:
            image = image + 0.5 * (label - (self.num_classes - 1) / 2.0)
            ids[1] = 2000 + label
        img_mask = torch.ones(self.num_image_tokens, dtype=torch.long)  # unused by the model (train.py:281-284)
        return ids, mask, seg, img_mask, torch.tensor(label), image


def make_loader(ds: Dataset, batch_size: int, shuffle: bool, num_workers: int = 0, drop_last: bool = False,
                sampler=None) -> DataLoader:
    """pin_memory + (optionally) worker processes, as run.py:131-140; H2D copies are issued non_blocking by the trainer."""
    return DataLoader(ds, batch_size=batch_size, shuffle=shuffle and sampler is None, num_workers=num_workers,
                      pin_memory=torch.cuda.is_available(), drop_last=drop_last, sampler=sampler,
                      persistent_workers=num_workers > 0)
