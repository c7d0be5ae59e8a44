"""Batch sources with the layout of the reference's collate output (updown-baseline/updown/data/datasets.py:173-202,
623-632): image_features (B,R,F) f32 zero-padded regions, caption_tokens (B,L) int64 0-padded without boundary
tokens, sentiment (B,1) f32 in {-1,0,1}.  The h5 / nltk readers themselves are out of scope; a
tensor file with those three arrays, or the synthetic generator of BASELINE.md §4, stands in."""
from typing import Dict, Iterator

import torch


class SyntheticCaptionData:
    """feats ~ N(0,1); caption lengths ~ U{8..L}, ids ~ U{2..V-1}, 0-padded; sentiment ~ U{-1,0,1}."""

    def __init__(self, num_images: int, R: int, F: int, L: int, V: int, seed: int = 1234):
        g = torch.Generator().manual_seed(seed)
        self.feats = torch.randn(num_images, R, F, generator=g)
        lens = torch.randint(min(8, L), L + 1, (num_images,), generator=g)
        ids = torch.randint(2, V, (num_images, L), generator=g)
        self.caps = torch.where(torch.arange(L).unsqueeze(0) < lens.unsqueeze(1), ids, torch.zeros_like(ids))
        self.senti = torch.randint(-1, 2, (num_images, 1), generator=g).float()
        self.image_id = torch.arange(num_images)

    def __len__(self):
        return self.feats.size(0)


class TensorFileData:
    """A .pt file holding {"image_features", "caption_tokens", "sentiment"[, "image_id"]} (loaded weights_only)."""

    def __init__(self, path: str):
        d = torch.load(path, map_location="cpu", weights_only=True)
        self.feats, self.caps = d["image_features"].float(), d["caption_tokens"].long()
        self.senti = d.get("sentiment", torch.zeros(self.feats.size(0), 1)).float().view(-1, 1)
        self.image_id = d.get("image_id", torch.arange(self.feats.size(0)))

    def __len__(self):
        return self.feats.size(0)


def cycle(data, batch_size: int, device, rank: int = 0, world: int = 1, seed: int = 0, shuffle: bool = True
          ) -> Iterator[Dict[str, torch.Tensor]]:
    """Endless iterator of device-resident batches (updown-baseline/updown/utils/common.py:7-27); each rank draws a
    disjoint row shard of every global batch of batch_size * world rows; uploads go through pinned memory."""
    g = torch.Generator().manual_seed(seed)
    n = len(data)
    gb = batch_size * world
    while True:
        perm = torch.randperm(n, generator=g) if shuffle else torch.arange(n)
        for i in range(0, n - gb + 1, gb):
            idx = perm[i + rank * batch_size: i + (rank + 1) * batch_size]
            out = {}
            for k, t in (("image_features", data.feats), ("caption_tokens", data.caps), ("sentiment", data.senti),
                         ("image_id", data.image_id)):
                out[k] = t[idx].pin_memory().to(device, non_blocking=True)
            yield out
