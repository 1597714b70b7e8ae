"""The product BVH builder (ptmi_bvh_create) against the reference's own BVH_Create.

* where oracle/_ref/libref_bvh.so exists (the reference's PathTracer_BVH.cpp compiled UNMODIFIED) every
  field the reference writes must be equal, and the triangle array must be reordered identically;
* everywhere: committed digests of trees that were checked against the reference when they were made
  (tests/golden/bvh_digests.json, generator: this file run as a script).
"""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import oracle_ffi as O
from opencl_pathtracer_amd import scenes, bvh_create, structs as S, PtmiError

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bvh_digests.json")
NODE_FIELDS = ["cutAxis", "triangleStartIndex", "nbTriangles", "son1Id", "son2Id", "isLeaf"]
CASES = [("cornell", 64, 48), ("matmix", 96, 96), ("tris20k", 96, 64), ("tris1m", 160, 90)]


def digest(sc):
    h = hashlib.sha256()
    for f in NODE_FIELDS:
        h.update(np.ascontiguousarray(sc.bvh[f]).tobytes())
    for bb in ("trianglesAABB", "centroidsAABB"):
        for f in ("pMin", "pMax", "centroid", "isEmpty"):
            h.update(np.ascontiguousarray(sc.bvh[bb][f]).tobytes())
    leaf = sc.bvh["isLeaf"] != 0
    h.update(np.ascontiguousarray(sc.bvh["comments"][leaf]).tobytes())
    h.update(np.ascontiguousarray(sc.triangulation["id"]).tobytes())
    return {"nodes": int(len(sc.bvh)), "max_depth": int(sc.bvhMaxDepth), "leaves": int(leaf.sum()),
            "sha256": h.hexdigest()}


def assert_same_tree(ref_nodes, ref_tris, ref_depth, sc):
    assert len(ref_nodes) == len(sc.bvh) and ref_depth == sc.bvhMaxDepth
    for f in NODE_FIELDS:
        assert np.array_equal(ref_nodes[f], sc.bvh[f]), f
    for bb in ("trianglesAABB", "centroidsAABB"):
        for f in ("pMin", "pMax", "centroid", "isEmpty"):
            assert np.array_equal(ref_nodes[bb][f], sc.bvh[bb][f]), (bb, f)
    leaf = ref_nodes["isLeaf"] != 0
    assert np.array_equal(ref_nodes["comments"][leaf], sc.bvh["comments"][leaf])
    bits = lambda x: np.ascontiguousarray(x).view(np.uint32) if x.dtype == np.float32 else x  # (a zero-area triangle's N is NaN)
    for f in S.Triangle.names:  # field-wise: the reference's member-wise swap does not move padding bytes
        a, b = ref_tris[f], sc.triangulation[f]
        if a.dtype.names:
            for g in a.dtype.names:
                assert np.array_equal(bits(a[g]), bits(b[g])), (f, g)
        else:
            assert np.array_equal(bits(a), bits(b)), f


@pytest.mark.parametrize("name,w,h", CASES)
def test_matches_reference_builder(name, w, h, built):
    if not O.have_ref_bvh():
        (O.missing_reference if os.path.isdir("/root/reference") else pytest.skip)("oracle/_ref/libref_bvh.so not present")
    if name == "tris1m" and os.environ.get("PTMI_SKIP_SLOW"):
        pytest.skip("slow")
    sc = scenes.build(name, w, h)
    ref_nodes, ref_tris, ref_depth = O.ref_bvh_create(sc.triangulation)
    bvh_create(sc)
    assert_same_tree(ref_nodes, ref_tris, ref_depth, sc)


@pytest.mark.parametrize("name,w,h", CASES)
def test_matches_committed_digest(name, w, h, built):
    golden = json.load(open(GOLDEN))
    sc = bvh_create(scenes.build(name, w, h))
    assert digest(sc) == golden[name]


def _random_soup(n, seed, spread=1.0):
    rs = np.random.RandomState(seed)
    c = rs.uniform(-spread, spread, (n, 1, 3))
    v = (c + rs.uniform(-0.2, 0.2, (n, 3, 3))).astype(np.float32)
    return scenes.triangle_create(v[:, 0], v[:, 1], v[:, 2])


@pytest.mark.parametrize("n", [1, 2, 4, 5, 9, 33, 257])
def test_small_and_ragged_sizes(n, built):
    tris = _random_soup(n, 100 + n)
    sc = scenes.cornell_box(8, 8)
    sc.triangulation = tris
    bvh_create(sc)
    assert sc.bvh["isLeaf"][0] == (1 if n <= 4 else 0) or n > 4
    leaves = sc.bvh[sc.bvh["isLeaf"] != 0]
    assert leaves["nbTriangles"].sum() == n  # every triangle in exactly one leaf
    covered = np.zeros(n, bool)
    for l in leaves:
        covered[l["triangleStartIndex"]:l["triangleStartIndex"] + l["nbTriangles"]] = True
    assert covered.all() and sorted(sc.triangulation["id"].tolist()) == list(range(n))
    if O.have_ref_bvh():
        ref_nodes, ref_tris, ref_depth = O.ref_bvh_create(tris)
        assert_same_tree(ref_nodes, ref_tris, ref_depth, sc)


def test_coincident_centroids_stop_on_min_diagonal(built):
    """All centroids (nearly) equal: the reference stops with NODE_LEAF_MIN_DIAG and a big leaf (BVH.cpp:142-149)."""
    rs = np.random.RandomState(7)
    base = rs.uniform(-0.2, 0.2, (1, 3, 3)).astype(np.float32)
    v = np.repeat(base, 40, axis=0)
    tris = scenes.triangle_create(v[:, 0], v[:, 1], v[:, 2])
    sc = scenes.cornell_box(8, 8)
    sc.triangulation = tris
    bvh_create(sc)
    assert len(sc.bvh) == 1 and sc.bvh["isLeaf"][0] and sc.bvh["nbTriangles"][0] == 40
    assert sc.bvh["comments"][0] == S.NODE_LEAF_MIN_DIAG
    if O.have_ref_bvh():
        assert_same_tree(*O.ref_bvh_create(tris), sc)


@pytest.mark.parametrize("seed", range(0, 60, 3))
def test_fuzzed_scenes_match_reference_builder(seed, built):
    """scenes.fuzz_scene (soup at mixed scales, coincident stacks of up to 40 triangles, flat boxes, clusters of thousands)
    with and without the hostile records (zero-area triangles, 1e6-unit coordinates, a 1e-6-unit triangle) and the corrupted
    ones (w components that are not 1 enter the boxes' fourth component): the same tree as the reference's own builder,
    field for field, and the same triangle order."""
    import warnings
    if not O.have_ref_bvh():
        (O.missing_reference if os.path.isdir("/root/reference") else pytest.skip)("oracle/_ref/libref_bvh.so not present")
    for suffix in ("", "h", "r", "hr"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sc = scenes.build(f"fuzz{seed}{suffix}_l1", 64, 64)
        ref_nodes, ref_tris, ref_depth = O.ref_bvh_create(sc.triangulation)
        bvh_create(sc)
        assert_same_tree(ref_nodes, ref_tris, ref_depth, sc)


@pytest.mark.parametrize("value", [np.nan, np.inf, -np.inf, 3e38])
def test_boxes_that_are_not_numbers_are_an_error_not_a_fault(value, built):
    """The reference's builder ASSERTs on them (BVH.cpp:199: a bin index out of range; without assertions it writes out of
    bounds).  The product's refuses non-finite boxes outright; with finite but overflowing ones (3e38) it either builds a tree
    (a usable one: every triangle in one leaf) or refuses - it never faults."""
    import warnings
    for seed in range(1, 7):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sc = scenes.build(f"fuzz{seed}_l1", 64, 64)
        rs = np.random.RandomState(seed)
        t = sc.triangulation
        for i in rs.choice(len(t), max(1, len(t) // 20), replace=False):
            t["AABB"][rs.choice(["pMin", "pMax", "centroid"])][i][int(rs.randint(0, 3))] = value
        if np.isfinite(value):
            try:
                bvh_create(sc)
            except PtmiError as e:
                assert e.code == -5 and "overflow" in str(e)
                continue
            leaves = sc.bvh[sc.bvh["isLeaf"] != 0]
            assert leaves["nbTriangles"].sum() == len(t) and sorted(sc.triangulation["id"].tolist()) == list(range(len(t)))
        else:
            with pytest.raises(PtmiError, match="not finite") as e:
                bvh_create(sc)
            assert e.value.code == -5


def test_builder_under_sanitizers(tmp_path):
    """csrc/bvh_build.cpp compiled with g++ -fsanitize=address,undefined and run on the fuzzed scenes (valid, hostile, corrupted
    records) and on boxes that are not numbers: no out-of-bounds access, no undefined behaviour.  (It found the node array
    overrun of a split that leaves one side empty - boxes whose extents overflow - which is an error code now.)"""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not installed")
    asan = subprocess.run([gxx, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run([gxx, "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(asan) and os.path.exists(asan) and os.path.isabs(ubsan) and os.path.exists(ubsan)):
        pytest.skip("libasan / libubsan not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = tmp_path / "stub.cpp"
    stub.write_text('#include <string>\nnamespace ptmi_internal { void set_global_error(const std::string&) {} }\n')
    so = tmp_path / "libbvh_asan.so"
    r = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                        "-fno-sanitize-recover=undefined", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "opencl_pathtracer_amd", "csrc"),
                        os.path.join(root, "opencl_pathtracer_amd", "csrc", "bvh_build.cpp"), str(stub), "-o", str(so)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = {**os.environ, "LD_PRELOAD": asan + ":" + ubsan, "ASAN_OPTIONS": "detect_leaks=0"}
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "sanitize", "bvh_builder_asan.py"), str(so)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "clean" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_empty_triangulation_is_an_error(built):
    sc = scenes.cornell_box(8, 8)
    sc.triangulation = np.zeros(0, S.Triangle)
    with pytest.raises(PtmiError):
        bvh_create(sc)


def test_tree_invariants_1m(built):
    """Properties the traversal relies on, at the BASELINE size: pre-order numbering (son1 = parent+1),
    children boxes inside the parent box, depth below the 30-entry stack."""
    sc = bvh_create(scenes.build("tris1m", 160, 90))
    b = sc.bvh
    inner = np.flatnonzero(b["isLeaf"] == 0)
    assert np.array_equal(b["son1Id"][inner], inner + 1)
    assert (b["son2Id"][inner] > b["son1Id"][inner]).all() and b["son2Id"].max() < len(b)
    assert sc.bvhMaxDepth < S.BVH_MAX_DEPTH and len(b) == 665533 and sc.bvhMaxDepth == 22
    for son in ("son1Id", "son2Id"):
        child = b["trianglesAABB"][b[son][inner]]
        parent = b["trianglesAABB"][inner]
        assert (child["pMin"][:, :3] >= parent["pMin"][:, :3]).all() and (child["pMax"][:, :3] <= parent["pMax"][:, :3]).all()
    leaves = b[b["isLeaf"] != 0]
    assert leaves["nbTriangles"].sum() == 1000000 and leaves["nbTriangles"].max() <= 4


if __name__ == "__main__":  # regenerate the digests (only meaningful where the reference builder agrees)
    out = {}
    for name, w, h in CASES:
        sc = scenes.build(name, w, h)
        if O.have_ref_bvh():
            ref = O.ref_bvh_create(sc.triangulation)
        bvh_create(sc)
        if O.have_ref_bvh():
            assert_same_tree(*ref, sc)
        out[name] = digest(sc)
        print(name, out[name])
    json.dump(out, open(GOLDEN, "w"), indent=1, sort_keys=True)
