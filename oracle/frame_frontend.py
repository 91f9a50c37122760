"""CPU restatement of the reference's Atari frame front-end -- TEST INFRASTRUCTURE ONLY
(importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from ga3c_amd/).

Reference path (ga3c/Environment.py):
    :52-54  _rgb2gray(rgb)   = np.dot(rgb[..., :3], [0.299, 0.587, 0.114])            -> f64 [H, W]
    :57-60  _preprocess      = misc.imresize(gray, [84, 84], 'bilinear') -> uint8;  .astype(f32) / 128 - 1
    :62-68  _get_current_state: FIFO of 4 planes, oldest first, transposed to [84, 84, 4]
    :70-74  _update_frame_q : drop the oldest plane when full, append the new one

`misc.imresize` is scipy.misc.imresize, a THIRD-PARTY function absent from this image (removed in SciPy 1.3;
the reference pins nothing -- README.md names no SciPy version; any SciPy <= 1.2 has it).  Its published
algorithm (scipy/misc/pilutil.py, SciPy 0.19 - 1.2) is restated here:
    imresize(arr, size, 'bilinear')  = fromimage(toimage(arr).resize((size[1], size[0]), resample=BILINEAR))
    toimage(2-D non-uint8 array)     = 8-bit 'L' image of bytescale(arr)          (mode=None, cmin/cmax=None)
    bytescale(data)                  = ((data - min) * (255 / (max - min))).clip(0, 255) + 0.5 -> uint8 (truncation);
                                       max == min -> scale 255 / 1
i.e. every frame is CONTRAST-STRETCHED to 0..255 by its own min / max before the resize -- a quirk of the
reference's input definition, kept.  `Image.resize(..., BILINEAR)` is Pillow's two-pass convolution resampler
(libImaging/Resample.c): triangle filter whose support is stretched by the downscale factor, horizontal pass then
vertical pass, 8-bit intermediate, coefficients in 22-bit fixed point.  It is restated in pil_bilinear_u8().

PINNING: Pillow IS importable here (PIL 12.2), so pil_bilinear_u8 is checked bit-for-bit against
Image.resize on random and structured images (tests/test_frontend_oracle.py) and the fixtures in
tests/golden/frontend.npz hold Pillow's own outputs.  bytescale is restated from SciPy's published source and has
no runnable reference here: that one step is "parity unpinned" (checked only against its documented examples).
The f64 gray product is checked against np.dot itself.
"""
import ctypes
import math

import numpy as np

GRAY = (0.299, 0.587, 0.114)                      # Environment.py:54
PRECISION_BITS = 32 - 8 - 2                       # Resample.c: 8 bits of pixel, 2 of headroom


_libm = ctypes.CDLL("libm.so.6")
_libm.fma.restype = ctypes.c_double
_libm.fma.argtypes = [ctypes.c_double] * 3
_fma = np.frompyfunc(_libm.fma, 3, 1)


def rgb2gray(rgb):
    """f64 fma(b, .114, fma(g, .587, r * .299)): np.dot(rgb[..., :3], GRAY) of a [H, W, 3+] uint8 frame evaluates
    each pixel in exactly this order with fused multiply-adds on this image's BLAS (x86-64 FMA kernels) --
    tests/test_frontend_oracle.py holds the two equal, bit for bit.  The last bit matters only where bytescale's
    truncation sits on an integer boundary."""
    rgb = np.asarray(rgb)
    r, g, b = (rgb[..., i].astype(np.float64) for i in range(3))
    return _fma(b, GRAY[2], _fma(g, GRAY[1], r * GRAY[0])).astype(np.float64)


def bytescale(data, low=0, high=255):
    """scipy.misc.bytescale with cmin = cmax = None (pilutil.py, SciPy <= 1.2)."""
    data = np.asarray(data)
    if data.dtype == np.uint8:
        return data
    cmin, cmax = data.min(), data.max()
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1
    scale = float(high - low) / cscale
    bytedata = (data - cmin) * scale + low
    return (bytedata.clip(low, high) + 0.5).astype(np.uint8)


def bilinear_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc of Pillow's Resample.c for the whole-image box and the triangle
    filter (support 1).  Returns (ksize, bounds int32[out,2] = (first input index, tap count), kk int32[out,ksize])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
    return ksize, bounds, kk


def _pass_rows(img, out_size, bounds, kk):
    """One resampling pass along axis 1 (ImagingResampleHorizontal_8bpc): int32 accumulate from 1 << 21, >> 22, clip8."""
    h = img.shape[0]
    out = np.empty((h, out_size), np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.full(h, 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc += src[:, xmin + x] * int(kk[xx, x])
        out[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_bilinear_u8(img, out_h, out_w):
    """Image.fromarray(img, 'L').resize((out_w, out_h), BILINEAR) for uint8 [H, W]: horizontal pass, then vertical,
    each skipped when that dimension does not change (Resample.c: ImagingResample)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    if w != out_w:
        _, b, k = bilinear_coeffs(w, out_w)
        img = _pass_rows(img, out_w, b, k)
    if h != out_h:
        _, b, k = bilinear_coeffs(h, out_h)
        img = _pass_rows(np.ascontiguousarray(img.T), out_h, b, k).T
    return np.ascontiguousarray(img)


def preprocess_u8(rgb, out_h=84, out_w=84):
    """Environment._preprocess up to (not including) the f32 conversion: the uint8 plane the frame queue holds."""
    return pil_bilinear_u8(bytescale(rgb2gray(rgb)), out_h, out_w)


def preprocess(rgb, out_h=84, out_w=84):
    """Environment._preprocess (Environment.py:57-60): f32 plane in {k/128 - 1}."""
    return preprocess_u8(rgb, out_h, out_w).astype(np.float32) / np.float32(128.0) - np.float32(1.0)


class FrameQueue:
    """frame_q + _get_current_state of Environment.py:62-74, holding uint8 planes."""

    def __init__(self, depth=4):
        self.depth, self.q = depth, []

    def clear(self):                                  # Environment.reset: frame_q.queue.clear()
        self.q = []

    def push(self, plane):
        if len(self.q) == self.depth:
            self.q.pop(0)
        self.q.append(np.asarray(plane, np.uint8))

    def state_u8(self):
        if len(self.q) < self.depth:
            return None
        return np.ascontiguousarray(np.transpose(np.array(self.q), [1, 2, 0]))
