"""Whole-step capture in a HIP graph for launch-bound (small-batch) training.

At the reference's batch sizes (64 .. 1024 datapoints) a training step is about sixty short kernels and the host
needs longer to issue them than the GPU to run them.  `GraphedStep` records one step - forward pipelines,
backward, clipping, optimiser - into a graph and replays it with a single launch.  The library's entry points
allocate nothing and take the stream as an argument, so they record as they are.  Noise drawn inside the library stays random across
replays through a device-resident Philox counter that the recording advances (demiss_vae_amd.rng.CaptureStream);
explicit noise tensors (`eps=`, keep-masks) must live outside the graph and be refilled in `refresh`, which runs
before every replay.  Optimisers need `capturable=True`."""
import torch

from . import rng


class GraphedStep:
    def __init__(self, body, refresh=None, nets=(), warmup=3):
        """body(): one training step on static tensors, returns a tensor (e.g. the bound) or None.
        refresh(): refills the static noise / batch tensors; called before every replay.
        nets: the PackedNets the step trains.  Their fragment images are re-packed inside the recording (so every
        replay packs the current weights) and marked stale after each replay for code running outside the graph -
        unless the step's optimiser keeps the images current itself (demiss_vae_amd.optim.Adam.attach_images): then
        nothing is packed inside the recording, and a replay only checks on the host that nothing else changed the
        parameters since."""
        self.refresh, self.nets = refresh, tuple(nets)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up on a side stream, as graph capture requires
            for _ in range(warmup):
                if refresh is not None:
                    refresh()
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.noise = rng.CaptureStream(torch.cuda.current_device())
        if refresh is not None:
            refresh()
        for n in self.nets:
            n.packed()          # allocate the image outside the recording ...
            if not n.image_kept_by_optimizer:
                n.invalidate()      # ... and make sure the packing kernel is part of it
        with torch.cuda.graph(self.graph):
            with self.noise:
                self.out = body()
        for n in self.nets:
            if not n.image_kept_by_optimizer:
                n.invalidate()

    def __call__(self):
        if self.refresh is not None:
            self.refresh()
        for n in self.nets:
            if n.image_kept_by_optimizer:
                n.packed()      # re-packs (eagerly, before the replay) only if something else changed the parameters
        self.graph.replay()
        for n in self.nets:
            if not n.image_kept_by_optimizer:
                n.invalidate()
        return self.out
