"""Drop-in `vgiwae` package: the reference's import paths, class names and constructor signatures
(what its YAML configs name), with the training / evaluation arithmetic running on the MI355X kernels
of demiss_vae_amd.  Only the hot path is here; diagnostics outside it are named so configs load, and say
so when called."""
