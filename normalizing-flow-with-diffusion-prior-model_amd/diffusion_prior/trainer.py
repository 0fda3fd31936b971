"""Bits/dim of the flow under the diffusion prior — mirrors `calculate_bpd_with_diff_prior`, reference
diffusion_prior/trainer.py:17-50.  The reference's training loop in the same file (aim logging, FID/KID evaluation,
optimiser plumbing) is outside the hot path and is not mirrored."""
import math

import torch

from normalizing_flow import preprocess_batch, initialize_with_zeros


@torch.no_grad()
def calculate_bpd_with_diff_prior(nf, diff_prior, *, data_loader, n_bits: int, n_bins: int, device: torch.device,
                                  reference_accumulation: bool = True):
    """For every batch: encode with the flow (dequantisation noise U[0,1)/n_bins), evaluate the variational bound of
    every latent part with its diffusion model, `bpd = mean_b((sum_parts nll - log_det) / ln 2)`.

    The reference overwrites `bpd` on every batch instead of summing and then divides by the number of batches
    (trainer.py:44-46); `reference_accumulation=True` (default) reproduces exactly that, False returns the mean over
    batches of the per-batch value."""
    nf.set_eval_mode()
    diff_prior.eval()
    bpd, acc, n_batches = 0.0, 0.0, 0
    for data in data_loader:
        batch = data[0].to(device) if isinstance(data, (list, tuple)) else data.to(device)
        batch = preprocess_batch(batch, n_bits, n_bins)
        log_likelihood = initialize_with_zeros(1, batch.size(0), device)
        latents, log_likelihood = nf.transform(batch + torch.rand_like(batch) / n_bins, log_likelihood)
        diff_nll = sum(diff_prior.evaluate_neg_log_likelihood(latents)) - log_likelihood
        bpd = (diff_nll / math.log(2.0)).mean(dim=0)
        acc = acc + bpd
        n_batches += 1
    result = bpd / len(data_loader) if reference_accumulation else acc / max(n_batches, 1)
    nf.set_train_mode()
    diff_prior.train()
    return result
