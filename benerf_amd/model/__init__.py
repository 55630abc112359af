"""Host-side mirror of the reference's `model` package (model/nerf.py, model/optimize.py,
model/embedder.py, model/component.py) dispatching to the HIP kernels."""
