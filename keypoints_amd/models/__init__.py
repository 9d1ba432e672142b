"""Drop-in mirror of the reference's `keypoints.models` package surface
(functional, knn, vgg, transporter, keynet, autoencoder) backed by the MI355X HIP kernels."""
from . import functional, knn, vgg, transporter, keynet, autoencoder  # noqa: F401
from .knn import Container  # noqa: F401   (the reference exports it at package level: autoencoder.py:1)
