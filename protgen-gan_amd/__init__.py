"""protgen-gan_amd — MI355X-native engine for the ProtGen-GAN hot path (ProtBERT
generator fill loop, critic + WGAN-GP, fused AdamW), behind the reference's Python API.

The directory name carries a hyphen, so import it through `pgload.py` at the repository
root (`import pgload; pg = pgload.load()`), which registers it as `protgen_gan_amd`."""
from . import _lib  # noqa: F401  (raises if libprotgen_hip.so is missing: no fallback)
from .bert import ProtBertConfig, ProtBertForMaskedLM
from .loss import compute_gradient_penalty, critic_loss, generator_loss
from .models import Critic, CriticHead, Generator
from .optim import AdamW
from .tokenizer import ProtTokenizer

__all__ = ["ProtBertConfig", "ProtBertForMaskedLM", "Generator", "Critic", "CriticHead", "critic_loss",
           "generator_loss", "compute_gradient_penalty", "AdamW", "ProtTokenizer"]
