#!/usr/bin/env python3
"""Render one of the built-in scenes (or a Wavefront OBJ) on the MI355X and save what the reference's viewer would
save: a 24-bit BMP of min(255 * sum / n, 255) (Alone/PathTracer_Dialog.cpp:161-185).

    python examples/render.py --scene cornell --width 512 --height 512 --spp 64 --depth 4 -o cornell.bmp
    python examples/render.py --obj mesh.obj --spp 32 -o mesh.bmp

The pixels are quantised on the device (ptmi_read_display): 3 bytes per pixel cross the bus.
"""
import argparse
import os
import struct
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import opencl_pathtracer_amd as pt  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scene", default="cornell", help="cornell | mayalike | matmix | tris<N>[k|m] (opencl_pathtracer_amd.scenes.build)")
    ap.add_argument("--obj", help="a Wavefront OBJ file instead of a built-in scene (opencl_pathtracer_amd.obj_import)")
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--strict-arithmetic", action="store_true",
                    help="the strict arithmetic instead of the reference's own build's (both bit-exact modes, DESIGN.md 2)")
    ap.add_argument("-o", "--output", default="render.bmp")
    args = ap.parse_args()

    w, h = args.width, args.height
    if args.obj:
        from opencl_pathtracer_amd import obj_import
        scene = obj_import.scene_from_obj(args.obj, w, h)
    else:
        scene = pt.scenes.build(args.scene, w, h)
    scene = pt.bvh_create(scene)
    flags = 0 if args.strict_arithmetic else pt.backend.FLAG_DEFAULT_ARITHMETIC  # default: the reference kernel's own pixels
    be = pt.Backend().setup_context(w, h, args.depth, scene.lightsSize, pt.structs.JITTERED, device=args.device, flags=flags)
    be.initialize_memory(scene)
    t0 = time.time()
    be.render(0, args.spp)
    be.synchronize()
    dt = time.time() - t0
    rows = be.read_display()  # uint8[h, stride]: B,G,R scanlines, image row 0 first, padded to 4 bytes
    c = be.counters()
    be.release()
    with open(args.output, "wb") as f:  # SaveBMP, Alone/PathTracer_bitmap.cpp:146-205
        f.write(struct.pack("<HIHHI", 0x4D42, 14 + 40 + rows.size, 0, 0, 0x36))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, 0, 0x0EC4, 0x0EC4, 0, 0))
        f.write(rows.tobytes())
    print(f"{args.output}: {w}x{h}, {args.spp} spp, depth {args.depth}: {c['segments'] / dt / 1e6:.1f} Msamples/s "
          f"({c['paths'] / dt / 1e6:.1f} Mpaths/s) in {dt * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
