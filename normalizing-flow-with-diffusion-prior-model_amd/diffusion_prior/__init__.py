"""MI355X-native drop-in for the reference's `diffusion_prior` package (hot-path symbols)."""
from .gaussian_diffusion import Unet, GaussianDiffusion, ModelPrediction
from .latent_formaters import IdentityFormater, CatFormater, get_formater
from .model import DiffusionModel, DiffusionPrior
from .trainer import calculate_bpd_with_diff_prior
