"""Synthetic similarity-transformed views for the scale/rotation estimator tests (test helper)."""
import numpy as np
from scipy import ndimage

from mrs_optic_flow_amd import synth


def canvas(seed: int, res: int) -> np.ndarray:
    c = synth.canvas_np(seed, res + 96, res + 96, True).astype(np.float32)
    return ndimage.gaussian_filter(c, 1.5)


def view(base: np.ndarray, res: int, scale: float, rot_deg: float) -> np.ndarray:
    """res x res uint8 view of `base` scaled by `scale` and rotated by `rot_deg` about the image centre."""
    c = (base.shape[0] - 1) / 2
    yy, xx = np.mgrid[0:res, 0:res].astype(np.float64)
    yy -= res / 2
    xx -= res / 2
    th = np.deg2rad(rot_deg)
    xs = (np.cos(th) * xx - np.sin(th) * yy) / scale + c
    ys = (np.sin(th) * xx + np.cos(th) * yy) / scale + c
    return np.clip(np.rint(ndimage.map_coordinates(base, [ys, xs], order=3)), 0, 255).astype(np.uint8)
