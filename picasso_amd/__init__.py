"""picasso_amd — MI355X-native backend for Picasso's localization hot path.

identify -> ROI cut -> Gaussian MLE fit -> localization table, as hand-written
HIP kernels for gfx950 behind the C ABI of include/picasso_hip.h, with the
reference's Python surface (picasso.localize / picasso.gaussmle) on top.
"""
__version__ = "0.1.0"
