"""CPU oracle of the operator-form LASSO of the reference's ``examples/cameraman.ipynb`` (TEST INFRASTRUCTURE:
as oracle/cpu_ref.py: the checker in tests/, tools/ and examples/, never the product).

Restates, in NumPy / SciPy, the callbacks of the notebook's cell 8 (``/root/reference/examples/cameraman.ipynb``
lines 219-272):

    f(x)     = array([ |correlate2d(idwt(x), K, "same", "symm") - b|^2 ])          (an array of ONE value: m = 1, :143)
    jac_f(x) = 2 dwt(correlate2d(correlate2d(idwt(x), K) - b, K)).reshape(1, -1)
    g(x)     = array([ l1_ratio |x|_1 ])
    prox     = where(|x| <= l1_ratio w, 0, x - l1_ratio w sign(x))
    L        = 2 max|dctn(K) / dctn(unit)|^2                                        (cell 8, last lines)

with two substitutions for packages this image does not have: ``pywt.dwt2 / idwt2(..., "haar")`` are written out
(one orthonormal Haar level, the [cA, cH, cV, cD] layout of the notebook's dwt_array) and ``skimage.filters.window
(("gaussian", 4), (9, 9))`` is the outer product of scipy.signal.windows.gaussian(9, 4) with itself, which is what
that call returns.  The image is synthetic (skimage's camera() is not available either).

Pinned: tests/golden/make_golden_r4.py runs the IMPORTED reference solver on these callbacks and stores its
outputs (fixture G13); oracle.cpu_ref on the same callbacks must reproduce them exactly."""
from __future__ import annotations

import numpy as np
from scipy.fftpack import dctn
from scipy.signal import correlate2d

L1_RATIO = 2e-5


def gaussian_kernel(size=9, std=4.0):
    k = np.exp(-0.5 * (np.arange(size) - (size - 1) / 2.0) ** 2 / std ** 2)
    return np.outer(k, k)


def synthetic_image(size, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size] / size
    img = 0.5 + 0.3 * np.sin(6 * xx) * np.cos(4 * yy)
    for _ in range(12):                                           # a few rectangles: edges for the wavelets
        a, b = rng.integers(0, size - size // 8, 2)
        img[a:a + size // 8, b:b + size // 8] += rng.uniform(-0.3, 0.3)
    return img


def dwt(img):
    """One orthonormal Haar level, flattened [cA, cH, cV, cD] (the notebook's dwt_array)."""
    a, b, c, d = img[0::2, 0::2], img[0::2, 1::2], img[1::2, 0::2], img[1::2, 1::2]
    return np.array([(a + b + c + d) / 2, (a + b - c - d) / 2, (a - b + c - d) / 2, (a - b - c + d) / 2]).flatten()


def idwt(vec, shape):
    h, w = shape[0] // 2, shape[1] // 2
    cA, cH, cV, cD = vec.reshape(4, h, w)
    img = np.empty((2 * h, 2 * w))
    img[0::2, 0::2] = (cA + cH + cV + cD) / 2
    img[0::2, 1::2] = (cA + cH - cV - cD) / 2
    img[1::2, 0::2] = (cA - cH + cV - cD) / 2
    img[1::2, 1::2] = (cA - cH - cV + cD) / 2
    return img


def lipschitz(kernel):
    unit = np.zeros(kernel.shape)
    unit[0, 0] = 1
    spectrum = dctn(kernel) / dctn(unit)
    return 2 * np.max(abs(spectrum)) ** 2


class BlurHaarL1Ref:
    """The four callbacks of the notebook for a kernel and an observed image."""

    def __init__(self, kernel, observed, l1_ratio=L1_RATIO):
        self.kernel = np.asarray(kernel, float)
        self.observed = np.asarray(observed, float)
        self.l1_ratio = float(l1_ratio)
        self.shape = self.observed.shape
        self.n_features = self.observed.size

    def blur(self, img):
        return correlate2d(img, self.kernel, mode="same", boundary="symm")

    def f(self, x):
        return np.array([np.linalg.norm(self.blur(idwt(x, self.shape)) - self.observed) ** 2])

    def jac_f(self, x):
        return 2 * dwt(self.blur(self.blur(idwt(x, self.shape)) - self.observed)).reshape(1, -1)

    def g(self, x):
        return np.array([self.l1_ratio * np.linalg.norm(x, ord=1)])

    def prox_wsum_g(self, weight, x):
        return np.where(np.abs(x) <= self.l1_ratio * weight, 0, x - self.l1_ratio * weight * np.sign(x))

    def callbacks(self):
        return self.f, self.g, self.jac_f, self.prox_wsum_g


def make_deblur(size=256, seed=1, noise=1e-3):
    """(kernel, observed image, x0 = dwt(observed), L) of the notebook's set-up on the synthetic image."""
    kernel = gaussian_kernel()
    rng = np.random.default_rng(seed)
    observed = correlate2d(synthetic_image(size), kernel, mode="same", boundary="symm") + rng.standard_normal((size, size)) * noise
    return kernel, observed, dwt(observed), lipschitz(kernel)
