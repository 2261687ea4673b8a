"""Total optimiser steps of a run, for the cosine schedule's T_max
(reference: vgiwae/utils/estimate_total_train_steps.py:7-36)."""
import math


class PLEstimateTotalTrainSteps:
    @property
    def estimated_num_training_steps(self):
        tr = self.trainer
        max_epochs, max_steps = getattr(tr, "max_epochs", None), getattr(tr, "max_steps", -1)
        no_epochs = max_epochs is None or max_epochs == -1
        if no_epochs and (max_steps is None or max_steps == -1):
            return float("inf")
        if no_epochs:
            return max_steps
        batches = getattr(tr, "num_training_batches", float("inf"))
        if batches == float("inf"):
            return max_steps
        accumulate = getattr(tr, "accumulate_grad_batches", 1) or 1
        steps = math.ceil(batches / accumulate) * max_epochs
        if max_steps is not None and max_steps != -1:
            steps = min(steps, max_steps)
        return steps
