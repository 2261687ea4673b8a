"""Bernoulli on {-1, +1} (reference: vgiwae/shared/bernoulli_between_m1_and_p1.py:4-26)."""
import torch
from torch.distributions.bernoulli import Bernoulli


class Bernoulli_m1_p1(Bernoulli):
    def sample(self, sample_shape=torch.Size()):
        return 2 * super().sample(sample_shape) - 1

    def log_prob(self, value):
        return super().log_prob((value + 1) / 2)

    @property
    def mean(self):
        raise NotImplementedError()

    @property
    def variance(self):
        raise NotImplementedError()
