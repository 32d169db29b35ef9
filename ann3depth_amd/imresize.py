"""scipy.misc.imresize / bytescale / imsave as the reference's preprocessor uses them (tools/data_preprocessor.py:195-208),
without scipy.misc (removed in SciPy 1.3) or PIL.

    imresize(arr, (rows, cols))   = fromimage(toimage(arr).resize((cols, rows), BILINEAR))     (scipy/misc/pilutil.py)

toimage() turns anything that is not uint8 into uint8 with bytescale() — per-image min-max scaling: this is where the
depth maps lose their metric scale — finds the channel axis of a 3-D array (the first axis of length 3) and hands PIL an
8-bit 'L' or 'RGB' image.  PIL's resize (Pillow >= 2.7: Resample.c, ImagingResample) is a separable convolution with the
triangle filter, widened by the scale factor when shrinking (antialiasing), in fixed point for 8-bit images: coefficients
rounded to 22 fractional bits, accumulator started at one half, result `>> 22` and clamped; horizontal pass first, each
pass through an 8-bit intermediate, a pass whose size does not change is skipped.

Pinned against Pillow itself where it is installed (tests/test_imresize_pillow.py: bit-identical to
PIL.Image.resize(..., BILINEAR) on random 8-bit images, up and down, grey and RGB, including the preprocessor's
480x640 -> 55x73); scipy.misc's part (bytescale / toimage) is restated from its source and checked on hand-computed
cases (tests/test_hdf5.py).
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bytescale(data, cmin=None, cmax=None, high=255, low=0):
    """scipy.misc.bytescale: uint8 passes through; everything else is scaled from [cmin, cmax] (default: its own min and
    max) to [low, high] in the array's arithmetic, then `+ 0.5` and truncated."""
    data = np.asarray(data)
    if data.dtype == np.uint8:
        return data
    if cmin is None:
        cmin = data.min()
    if cmax is None:
        cmax = data.max()
    cscale = cmax - cmin
    if cscale < 0:
        raise ValueError('`cmax` should be larger than `cmin`.')
    if cscale == 0:
        cscale = 1
    scale = float(high - low) / cscale
    bytedata = (data - cmin) * scale + low
    return (bytedata.clip(low, high) + 0.5).astype(np.uint8)


def _coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    bounds, ks = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)             # C casts: truncation towards zero
        xmax = min(int(center + support + 0.5), in_size)
        x = np.arange(xmin, xmax)
        w = np.clip(1.0 - np.abs((x - center + 0.5) / filterscale), 0.0, None)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        k = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
        bounds.append(xmin)
        ks.append(k)
    return bounds, ks


def _pass(img, out_size, axis):
    """One 8-bit resampling pass along `axis` of a uint8 array."""
    in_size = img.shape[axis]
    if in_size == out_size:
        return img
    bounds, ks = _coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        k = ks[xx].reshape((-1,) + (1,) * (src.ndim - 1))
        ss = (1 << (PRECISION_BITS - 1)) + (src[bounds[xx]:bounds[xx] + len(ks[xx])] * k).sum(0)
        out[xx] = np.clip(ss >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def toimage(arr):
    """The 8-bit image scipy.misc.toimage builds: (rows, cols) or (rows, cols, 3) uint8, channels last."""
    data = np.asarray(arr)
    if np.iscomplexobj(data):
        raise ValueError('Cannot convert a complex-valued array.')
    if data.ndim == 2:
        return bytescale(data)
    if data.ndim != 3 or not (3 in data.shape or 4 in data.shape):
        raise ValueError("'arr' does not have a suitable array shape for any mode.")
    want = 3 if 3 in data.shape else 4
    ca = int(np.flatnonzero(np.asarray(data.shape) == want)[0])
    return np.moveaxis(bytescale(data), ca, 2)


def imresize(arr, size):
    """scipy.misc.imresize(arr, (rows, cols), interp='bilinear') -> uint8 array of shape (rows, cols[, channels])."""
    img = toimage(arr)
    rows, cols = int(size[0]), int(size[1])
    img = _pass(img, cols, 1)          # horizontal first
    return _pass(img, rows, 0)


def imsave(path, arr):
    """scipy.misc.imsave: toimage(arr, channel_axis=2).save(path) — an 8-bit PNG, greyscale or RGB."""
    from . import png
    a = np.asarray(arr)
    img = bytescale(a) if a.ndim == 2 or a.shape[2] in (3, 4) else None
    if img is None:
        raise ValueError('imsave: (rows, cols) or (rows, cols, 3|4) arrays only')
    png.imsave(path, img)
