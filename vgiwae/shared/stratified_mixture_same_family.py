"""Mixture whose sample() returns one draw per component (reference:
vgiwae/shared/stratified_mixture_same_family.py:4-33); used only by the torch.distributions view of the model."""
import torch
from torch.distributions.mixture_same_family import MixtureSameFamily


class StratifiedMixtureSameFamily(MixtureSameFamily):
    sample_stratified = True

    def set_sample_not_stratified(self):
        self.sample_stratified = False

    def rsample(self, sample_shape=torch.Size()):
        if not self.sample_stratified:
            return super().rsample(sample_shape)
        return self.component_distribution.rsample(sample_shape)

    def sample(self, sample_shape=torch.Size()):
        if not self.sample_stratified:
            return super().sample(sample_shape)
        return self.component_distribution.sample(sample_shape)

    def sample_not_stratified(self, sample_shape=torch.Size()):
        return super().sample(sample_shape)
