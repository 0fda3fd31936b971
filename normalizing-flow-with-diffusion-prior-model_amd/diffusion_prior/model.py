"""Diffusion prior composition — mirrors reference diffusion_prior/model.py: DiffusionModel :9-76, DiffusionPrior :79-159.

One (Unet + GaussianDiffusion) per latent part of the flow, `auto_normalize=False` (reference :37-38).  The
reference keeps its per-part models in a plain Python list, so its state_dict() is empty and .to()/.train() do not
propagate; here they live in an nn.ModuleList under the attribute `priors` — extra keys
that the reference never wrote, nothing it did write is renamed — and `parameters(only_trainable)` keeps the
reference's generator semantics.
"""
import numpy as np
import torch
import torch.nn as nn

from normalizing_flow.base import default_device
from .gaussian_diffusion import Unet, GaussianDiffusion


class DiffusionModel(nn.Module):
    """A single diffusion model over one latent part  (reference model.py:9-76)."""

    def __init__(self, unet_kwargs, diffusion_kwargs, image_size, channels):
        super().__init__()
        network = Unet(**unet_kwargs, channels=channels)
        self.device = default_device()
        self.__diffusion = GaussianDiffusion(network, **diffusion_kwargs, image_size=image_size,
                                             auto_normalize=False).to(self.device)
        self.loss_type = self.__diffusion.loss_type

    @property
    def diffusion(self) -> GaussianDiffusion:
        return self.__diffusion

    def forward(self, x):
        return self.__diffusion(x)

    def sample_latent(self, n_samples: int, return_all_timesteps: bool = False):
        return self.__diffusion.sample(batch_size=n_samples, return_all_timesteps=return_all_timesteps)

    def sample_latent_iter(self, n_samples: int):
        return self.__diffusion.sample_iter(batch_size=n_samples)

    def sample_latent_given_start_iter(self, start):
        """(reference model.py:57-70) as a generator: noise the start to T-1, then the reverse chain on the captured step."""
        d = self.__diffusion
        with torch.no_grad():
            timesteps = torch.ones(start.size(0), dtype=torch.int64, device=start.device) * (d.num_timesteps - 1)
            x_t = d.q_sample(start, timesteps, noise=None)
        return (yield from d.reverse_from_iter(x_t, d.num_timesteps - 1))

    def sample_latent_given_start(self, start):
        it = self.sample_latent_given_start_iter(start)
        try:
            while True:
                next(it)
        except StopIteration as done:
            return done.value

    def interpolate_latent(self, latent_1, latent_2, lam):
        return self.__diffusion.interpolate(latent_1, latent_2, t=None, lam=lam)

    def calc_neg_log_likelihood_loop(self, x):
        return self.__diffusion.calc_neg_log_likelihood_loop(x)

    def calc_neg_log_likelihood_iter(self, x):
        d = self.__diffusion
        if d.use_graph and x.is_cuda:
            return d.calc_neg_log_likelihood_iter(x)
        return d.calc_neg_log_likelihood_loop(x)


class DiffusionPrior(nn.Module):
    """Models the flow's prior with one diffusion model per latent part  (reference model.py:79-159)."""

    def __init__(self, *, latent_formater, unet_kwargs, diffusion_kwargs, image_sizes, latent_channels):
        super().__init__()
        self.device = default_device()
        self.latent_formater = latent_formater
        self.priors = nn.ModuleList([DiffusionModel(unet_kwargs, diffusion_kwargs, image_size=image_sizes[i],
                                                    channels=latent_channels[i])
                                     for i in range(self.latent_formater.get_num_latent_parts())])
        self.loss_type = self.priors[0].loss_type
        self.concurrent_parts = True          # one HIP stream per latent part in the sampling loops (see _sample_parts)
        self._part_streams = None

    def forward(self, latents: list) -> list:
        """List of per-part losses  (reference model.py:105-119)."""
        losses = []
        processed_latents = self.latent_formater.process_latents(latents)
        for i in range(0, len(processed_latents)):
            losses.append(self.priors[i](processed_latents[i]))
        return losses

    def _sample_parts(self, make) -> list:
        """Runs one sampling chain per latent part; make(prior, idx) returns the chain as a generator that yields between
        timesteps (or a plain result).  The chains are independent and each step is ~100 small kernels that leave most of the
        256 CUs idle, so on a GPU every part gets its own HIP stream (its own hardware queue) and the host enqueues the
        parts' timesteps in round-robin: the chains overlap (2x for the three MNIST/CIFAR parts).  Each chain draws its noise
        from a generator of its own, seeded from the default generator in part order, so the samples are those of the
        sequential loop (`concurrent_parts = False`) bit for bit."""
        def drain(chain):
            if not hasattr(chain, "__next__"):
                return chain
            try:
                while True:
                    next(chain)
            except StopIteration as done:
                return done.value

        dev = next(self.priors[0].parameters()).device
        if not (self.concurrent_parts and dev.type == "cuda" and len(self.priors) > 1) or torch.cuda.is_current_stream_capturing():
            return [drain(make(prior, i)) for i, prior in enumerate(self.priors)]
        cur = torch.cuda.current_stream(dev)
        if self._part_streams is None or len(self._part_streams) != len(self.priors):
            # (measured, round 5: the largest part's chain alone takes 1.13 s of the 1.33-1.41 s call; putting it on a high-priority HIP
            # stream made the call 1.9 s - profiles/r05_g_sampling_parts.txt - so all parts keep default-priority streams)
            self._part_streams = [torch.cuda.Stream(device=dev) for _ in self.priors]
        n = len(self.priors)
        chains, outs, live = [None] * n, [None] * n, []
        for i, prior in enumerate(self.priors):
            self._part_streams[i].wait_stream(cur)
            with torch.cuda.stream(self._part_streams[i]):
                chains[i] = make(prior, i)
                if hasattr(chains[i], "__next__"):
                    live.append(i)
                else:
                    outs[i] = chains[i]
        while live:
            for i in list(live):
                with torch.cuda.stream(self._part_streams[i]):
                    try:
                        next(chains[i])
                    except StopIteration as done:
                        outs[i] = done.value
                        live.remove(i)
        for side, out in zip(self._part_streams, outs):
            cur.wait_stream(side)
            if isinstance(out, torch.Tensor):
                out.record_stream(cur)
        return outs

    def sample_latents(self, n_samples: int, return_all_timesteps: bool = False) -> list:
        """(reference model.py:121-134)"""
        if not return_all_timesteps:
            return self.latent_formater.postprocess(self._sample_parts(lambda prior, i: prior.sample_latent_iter(n_samples)))
        return [prior.sample_latent(n_samples, True) for prior in self.priors]

    @torch.no_grad()
    def sample_latents_given_start(self, latents: list) -> list:
        return self._sample_parts(lambda prior, idx: prior.sample_latent_given_start_iter(latents[idx]))

    def evaluate_neg_log_likelihood(self, latents: list) -> list:
        processed_latents = self.latent_formater.process_latents(latents)
        bounds = self._sample_parts(lambda prior, i: prior.calc_neg_log_likelihood_iter(processed_latents[i]))
        return [bound / np.prod(latent.shape[1:]) for bound, latent in zip(bounds, processed_latents)]

    def interpolate_latents(self, latents1, latents2, lam):
        return [prior.interpolate_latent(latents1[idx], latents2[idx], lam) for idx, prior in enumerate(self.priors)]

    def parameters(self, only_trainable=False):
        """Generator over all per-part parameters  (reference model.py:152-159)."""
        for prior in self.priors:
            for param in prior.parameters():
                if only_trainable and not param.requires_grad:
                    continue
                yield param
