"""A MixtureSameFamily that, by default, draws one sample from every component instead of picking components at
random (reference behaviour: vgiwae/shared/stratified_mixture_same_family.py:4-33).  Only the torch.distributions
view of a model uses it; the kernels sample stratified mixtures themselves."""
import torch
from torch.distributions.mixture_same_family import MixtureSameFamily


class StratifiedMixtureSameFamily(MixtureSameFamily):
    sample_stratified = True          # switched off per instance by set_sample_not_stratified()

    def set_sample_not_stratified(self):
        self.sample_stratified = False

    def _draw(self, method: str, sample_shape):
        """stratified: the component distribution's own draw, shape sample_shape + batch + [components] + event;
        otherwise the mixture's ancestral draw"""
        if self.sample_stratified:
            return getattr(self.component_distribution, method)(sample_shape)
        return getattr(super(), method)(sample_shape)

    def sample(self, sample_shape=torch.Size()):
        return self._draw("sample", sample_shape)

    def rsample(self, sample_shape=torch.Size()):
        return self._draw("rsample", sample_shape)

    def sample_not_stratified(self, sample_shape=torch.Size()):
        return MixtureSameFamily.sample(self, sample_shape)
