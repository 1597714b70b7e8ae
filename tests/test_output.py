"""Display quantisation and BMP bytes as the reference produces them (PathTracer_bitmap.cpp:146-286)."""
import os
import struct

import numpy as np
import pytest

from opencl_pathtracer_amd import output


def test_quantisation_rules():
    c = np.zeros((2, 3, 4), np.float32)
    n = np.full((2, 3), 4.0, np.float32)
    c[0, 0] = (2.0, 1.0, 0.5, 0)      # 0.5, 0.25, 0.125 of full scale
    c[0, 1] = (8.0, 100.0, 4.0, 0)    # >= 1 -> clamps to 255
    c[0, 2] = (-1.0, 3.0, 3.0, 0)     # negative red sum -> pure red marker
    n[1, 0] = 0.0                     # never sampled: 0/0 = NaN -> min() macro yields 255
    rgb = output.to_display_rgb(c, n)
    assert tuple(rgb[0, 0]) == (127, 63, 31)
    assert tuple(rgb[0, 1]) == (255, 255, 255)
    assert tuple(rgb[0, 2]) == (255, 0, 0)
    assert tuple(rgb[1, 0]) == (255, 255, 255)
    assert tuple(rgb[1, 1]) == (0, 0, 0)


def test_bmp_file_layout(tmp_path):
    w, h = 5, 3  # 15 bytes per row -> padded to 16
    c = np.zeros((h, w, 4), np.float32)
    c[0, 0, :3] = (1.0, 0.5, 0.25)
    c[2, 4, :3] = (0.25, 0.5, 1.0)
    p = tmp_path / "img.bmp"
    output.save_bmp(str(p), c, np.ones((h, w), np.float32))
    raw = p.read_bytes()
    magic, size, _, _, off = struct.unpack_from("<HIHHI", raw, 0)
    assert magic == 0x4D42 and off == 0x36 and size == 54 + 16 * h == len(raw)
    hdr = struct.unpack_from("<IiiHHIIiiII", raw, 14)
    assert hdr[:5] == (40, w, h, 1, 24) and hdr[5] == 0 and hdr[7] == hdr[8] == 0x0EC4
    assert raw[54:57] == bytes([63, 127, 255])           # first pixel of accumulator row 0, as B,G,R
    last = 54 + 2 * 16 + 4 * 3
    assert raw[last:last + 3] == bytes([255, 127, 63]) and raw[54 + 15] == 0  # padding byte


@pytest.mark.gpu
def test_example_renderer_writes_the_bmp_the_host_path_would(tmp_path, scene_factory):
    """examples/render.py: device-side quantisation + header == output.save_bmp of the oracle's accumulators."""
    import subprocess
    import sys
    import oracle_ffi as O
    from opencl_pathtracer_amd import output
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "cornell.bmp"
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "render.py"), "--scene", "cornell", "--width", "50",
                        "--height", "38", "--spp", "6", "--depth", "4", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Msamples/s" in r.stdout
    sc = scene_factory("cornell", 50, 38)
    color, count, _, _ = O.oracle_render(sc, 50, 38, 4, 6, default_arithmetic=True)  # (the example renders the reference's own pixels)
    want = tmp_path / "want.bmp"
    output.save_bmp(str(want), color, count)
    assert out.read_bytes() == want.read_bytes()
