"""Empty placeholder for `torchvision`, which is not installed in the build container.

Used ONLY by tools/golden_gen/make_golden.py so that the reference's `vgiwae.data` package can be imported (its
modules say `import torchvision` at the top; only the MNIST / Omniglot loaders ever touch an attribute of it, and the
golden vectors are generated from the toy data the reference ships).  Plumbing only: nothing is implemented here, so
any use of the package's functionality fails loudly with an AttributeError.
"""
