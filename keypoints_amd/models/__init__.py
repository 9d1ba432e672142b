"""Drop-in mirror of the reference's `keypoints.models` package surface
(functional, knn, vgg, transporter, keynet) backed by the MI355X HIP kernels."""
from . import functional, knn, vgg, transporter, keynet  # noqa: F401
