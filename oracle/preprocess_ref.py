"""ORACLE (test infrastructure) -- PARITY UNPINNED: cv2 is absent in the build container, so the reference's input pipeline
cannot be run; this restates it from the reference's own code plus OpenCV's published resize algorithm, and is held by
known-answer tests and by torch.nn.functional.interpolate (same half-pixel / clamp semantics, independent code).

aerialpeople_crop.__getitem__ (copenet/src/copenet/dsets/aerialpeople.py:125-141,174) and resize_with_pad
(copenet/src/copenet/utils/utils.py:214-235)."""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406])      # aerialpeople.py:67-68
STD = np.array([0.229, 0.224, 0.225])


def cv2_resize_linear(img, dw, dh):
    """cv2.resize(img, (dw, dh)) for a float64 HWC image: INTER_LINEAR as in OpenCV's resize.cpp -- inverse map
    fx = (float)((dx + 0.5) * (1 / (dw / w)) - 0.5), floor, clamp to the border, float32 coefficients, double data."""
    h, w = img.shape[:2]

    def taps(d, s):
        scale = 1.0 / (d / s)
        f = ((np.arange(d) + 0.5) * scale - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = (f - i).astype(np.float32)
        lo = i < 0
        f[lo], i[lo] = 0.0, 0
        hi = i >= s - 1
        f[hi], i[hi] = 0.0, s - 1
        return i, np.minimum(i + 1, s - 1), f.astype(np.float64)
    x0, x1, fx = taps(dw, w)
    y0, y1, fy = taps(dh, h)
    rows = img[:, x0] * (1 - fx)[None, :, None] + img[:, x1] * fx[None, :, None]
    return rows[y0] * (1 - fy)[:, None, None] + rows[y1] * fy[:, None, None]


def resize_with_pad(img, size=224):
    """utils.py:214-235."""
    bigger = img.shape[0] if img.shape[0] > img.shape[1] else img.shape[1]
    scale = size / bigger
    out = cv2_resize_linear(img, int(scale * img.shape[1]), int(scale * img.shape[0]))
    pad_top = (size - out.shape[0]) // 2
    pad_left = (size - out.shape[1]) // 2
    canvas = np.zeros((size, size, 3))
    canvas[pad_top:pad_top + out.shape[0], pad_left:pad_left + out.shape[1]] = out
    return canvas, scale, [pad_left, pad_top]


def preprocess(frame_bgr_u8, crop):
    """One view of __getitem__: frame[:, :, ::-1] / 255 (:125), crop (:127), resize_with_pad (:140), CHW float32,
    Normalize (:174).  crop = (y0, y1, x0, x1)."""
    img = frame_bgr_u8[:, :, ::-1] / 255.0
    y0, y1, x0, x1 = crop
    im, scale, pad = resize_with_pad(img[y0:y1, x0:x1, :])
    chw = im.transpose(2, 0, 1).astype(np.float32)
    return ((chw - MEAN.astype(np.float32)[:, None, None]) / STD.astype(np.float32)[:, None, None]).astype(np.float32), scale, pad
