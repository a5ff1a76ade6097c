"""Latent cache: encode every clip once with the VAE, train from `cached_{i}.pt` files afterwards.

Mirror of the reference's `handle_cache_latents` (train.py:266-314) and `CachedDataset` (utils/dataset.py:589-603), SURVEY
§8(f) row 1 (config C5's "VAE-encode offload": removes the 4.4 TFLOP encode from every step).  File format kept exactly:
  `{output_dir}/cached_latents/cached_{i}.pt` = `torch.save(batch)` where `batch['pixel_values']` has been REPLACED by the
  scaled latents `(B,4,F,h,w)` and every value is stripped of its batch dimension (`v[0]`; lists of prompts -> first item);
  the key stays `'pixel_values'` (train.py:744 reads it back as latents when `cache_latents` is on).
Differences, both deliberate: the encode runs through the native VAE at its bf16 compute type (the reference casts the VAE to
fp16), and `CachedDataset` loads onto THIS process's device instead of the hard-coded `'cuda:0'` (utils/dataset.py:602), which
is what makes the cached path usable with one process per GPU.
"""
import os

import torch
from torch.utils.data import DataLoader, Dataset

from ..models.vae import tensor_to_vae_latent


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class CachedDataset(Dataset):
    def __init__(self, cache_dir: str = "", map_location=None):
        self.cache_dir = cache_dir
        self.map_location = map_location
        self.cached_data_list = self.get_files_list()

    def get_files_list(self):
        return sorted(f"{self.cache_dir}/{x}" for x in os.listdir(self.cache_dir) if x.endswith(".pt"))

    def __len__(self):
        return len(self.cached_data_list)

    def __getitem__(self, index):
        return torch.load(self.cached_data_list[index], map_location=self.map_location or _device())


def handle_cache_latents(should_cache, output_dir, train_dataloader, train_batch_size, vae, cached_latent_dir=None,
                         shuffle=False):
    """Returns a DataLoader over the cached batches (or None when caching is off), like the reference."""
    if not should_cache:
        return None
    cached_latent_dir = os.path.abspath(cached_latent_dir) if cached_latent_dir is not None else None
    if cached_latent_dir is None:
        cache_save_dir = f"{output_dir}/cached_latents"
        os.makedirs(cache_save_dir, exist_ok=True)
        dev = _device()
        for i, batch in enumerate(train_dataloader):
            with torch.no_grad():
                batch["pixel_values"] = tensor_to_vae_latent(batch["pixel_values"].to(dev), vae)
            for k, v in batch.items():
                batch[k] = v[0]
            torch.save(batch, f"{cache_save_dir}/cached_{i}.pt")
    else:
        cache_save_dir = cached_latent_dir
    return DataLoader(CachedDataset(cache_dir=cache_save_dir), batch_size=train_batch_size, shuffle=shuffle, num_workers=0)
