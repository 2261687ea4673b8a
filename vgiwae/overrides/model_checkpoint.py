"""Checkpoint policy of the entry scripts (reference: vgiwae/overrides/model_checkpoint.py:11-66, a Lightning callback:
best model by validation loss, last model, snapshots at chosen epochs / steps).  The Lightning-free runner implements
the same policy in demiss_vae_amd.run.Checkpointer; the name is kept importable from here."""
from demiss_vae_amd.run import Checkpointer as ModelCheckpoint  # noqa: F401
