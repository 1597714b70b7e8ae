"""The reference's output path: accumulators -> displayed pixels -> 24-bit BMP.

What the reference shows and saves after every image (``Alone/PathTracer_Dialog.cpp:161-185``) is produced by
``ConvertRGBAToBMPBuffer`` (``Alone/PathTracer_bitmap.cpp:237-286``) and ``SaveBMP`` (:146-205):

* pixel = ``(int) min(sum * 255.f / n, 255.f)`` per channel, ``min`` being the Windows macro ``a < b ? a : b``
  (so a NaN from 0/0 on a never-sampled pixel shows as 255); a negative red sum marks the pixel pure red
  (the reference tests ``.x`` three times, :262);
* bytes in B, G, R order, rows padded to a multiple of 4, image row 0 written first (a BMP stores rows bottom-up,
  so the picture is flipped relative to the accumulator, as in the reference);
* 54-byte header: 'BM', bfOffBits 0x36, 24 bits, BI_RGB, 0x0ec4 pixels per metre.
"""
import struct

import numpy as np


def to_display_rgb(image_color, image_ray_nb=None):
    """uint8[H,W,3] (R,G,B) exactly as the reference quantises them."""
    c = np.asarray(image_color, np.float32)
    n = np.ones(c.shape[:2], np.float32) if image_ray_nb is None else np.asarray(image_ray_nb, np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = c[..., :3] * np.float32(255.0) / n[..., None]
        v = np.where(v < np.float32(255.0), v, np.float32(255.0))  # min(a,b) = a<b ? a : b  (NaN -> 255)
        out = v.astype(np.int32).astype(np.uint8)
    neg = c[..., 0] < 0
    out[neg] = (255, 0, 0)
    return out


def to_bmp_buffer(image_color, image_ray_nb=None):
    """Padded B,G,R scanlines, image row 0 first (ConvertRGBAToBMPBuffer)."""
    rgb = to_display_rgb(image_color, image_ray_nb)
    h, w, _ = rgb.shape
    psw = (w * 3 + 3) & ~3
    buf = np.zeros((h, psw), np.uint8)
    buf[:, :w * 3] = rgb[..., ::-1].reshape(h, w * 3)
    return buf.tobytes(), psw


def save_bmp(path, image_color, image_ray_nb=None):
    """SaveBMP (:146-205)."""
    data, _ = to_bmp_buffer(image_color, image_ray_nb)
    h, w = np.asarray(image_color).shape[:2]
    with open(path, "wb") as f:
        f.write(struct.pack("<HIHHI", 0x4D42, 14 + 40 + len(data), 0, 0, 0x36))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, 0, 0x0EC4, 0x0EC4, 0, 0))
        f.write(data)
