"""MI355X-native path-tracing integrator: drop-in device backend for adjerbetian/OpenCL_PathTracer scenes.

`backend`  - host-side mirror of the reference's backend interface over the C ABI (libptmi.so)
`scenes`   - scene construction conventions (Triangle_Create, lights, materials) + BASELINE workloads
`structs`  - numpy dtypes of the scene contract (include/ptmi_scene.h)
`scene_cache` - the reference's scene-cache files (sizes.pth / pointers.pth / textureData.pth)
"""
from . import structs, scenes, backend, scene_cache, output, obj_import  # noqa: F401
from .backend import Backend, PtmiError, bvh_create, render_scene, validate_scene  # noqa: F401

__all__ = ["structs", "scenes", "backend", "scene_cache", "output", "obj_import", "Backend", "PtmiError", "bvh_create", "render_scene", "validate_scene"]
