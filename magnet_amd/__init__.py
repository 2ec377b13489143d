"""magnet_amd — MI355X-native (gfx950) implementation of MaGNet's multi-view matching hot path.

Scope is SURVEY.md §8: Gaussian depth-candidate sampling, pose/depth warping of F-Net features,
consistency-weighted matching score, iterative Gaussian update — behind the reference's
`MAGNET.forward` / `homography.est_costvolume_CW` interface.  The compute lives in
`csrc/*.hip` behind the C ABI declared in `include/magnet_hip.h`; this package is the ctypes
host layer.  There is no CPU fallback: importing works anywhere, calling needs the built
library and a gfx950 device.
"""
__version__ = "0.1.0"
