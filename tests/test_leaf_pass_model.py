"""The equivalence the wavefront kernel's leaf passes rest on (csrc/kernel_wavefront.hip: leaf_pass), as an executable model.

The reference tests the triangles of a leaf one after the other and updates the limit between tests (FullKernel.cl:638-646;
a shadow query stops at the first hit, :724-727).  A leaf pass tests up to four triangles of each waiting ray side by side,
all against the limit at ENTRY, and resolves them with one minimum per ray over a 64-bit key: (distance bits, ~index) for a
closest-hit query, the index for a shadow query; a pass holds 64 items, so a leaf can be split over passes anywhere.
Property: for any distances (ties included), any pass / fail pattern of the other tests and any split, both give the same
accepted triangle, the same final limit and the same number of counted tests."""
import numpy as np
from hypothesis import given, settings, strategies as st


def sequential(limit, nsd, ok, shadow):
    """the reference's leaf loop: returns (accepted index or -1, limit afterwards, tests counted)"""
    hit = -1
    for i, (d, p) in enumerate(zip(nsd, ok)):
        if p and not d > limit:       # Triangle_Intersects rejects with `> limit` (cl:560): equal distances are accepted
            hit, limit = i, d
            if shadow:
                return hit, limit, i + 1
    return hit, limit, len(nsd)


def key_bits(d):
    return int(np.float32(d).view(np.uint32))   # non-negative floats order like their bit patterns


def by_passes(limit, nsd, ok, shadow, chunks):
    """leaf passes: `chunks` says how many of the ray's triangles each pass takes (1..4: owner's offer cut by the pass's room)"""
    hit, i, counted = -1, 0, 0
    for take in chunks:
        if i >= len(nsd):
            break
        idx = range(i, min(i + take, len(nsd)))
        none = (1 << 64) - 1
        key = none if shadow else (key_bits(limit) << 32) | 0xFFFFFFFF
        mine = {}
        for j in idx:                                        # the items of one pass: tested side by side
            if ok[j] and not nsd[j] > limit:                 # against the limit at entry
                mine[j] = (j << 32) if shadow else (key_bits(nsd[j]) << 32) | (~j & 0xFFFFFFFF)
                key = min(key, mine[j])                      # the LDS minimum
        if shadow:
            if key != none:
                return key >> 32, nsd[key >> 32], counted + (key >> 32) - i + 1
        else:
            for j, k in mine.items():                        # every candidate asks whether it is the one its owner keeps
                if k == key:
                    hit = j
            limit = float(np.uint32(key >> 32).view(np.float32))   # the owner reads its limit back from the key
        counted += len(idx)
        i += len(idx)
    return hit, limit, counted


# few distinct distances, so that ties and "equal to the limit" happen all the time
distance = st.sampled_from([np.float32(x) for x in (1e-5, 0.25, 0.5, 0.5000001, 1.0, 2.0, 7.5, 1e30)])


@settings(max_examples=400, deadline=None)
@given(st.lists(st.tuples(distance, st.booleans()), min_size=1, max_size=11), distance | st.just(np.float32(np.inf)),
       st.booleans(), st.lists(st.integers(1, 4), min_size=11, max_size=11))
def test_passes_equal_the_sequential_leaf_loop(tris, limit, shadow, chunks):
    nsd, ok = [t[0] for t in tris], [t[1] for t in tris]
    want = sequential(limit, nsd, ok, shadow)
    got = by_passes(limit, nsd, ok, shadow, chunks)
    assert got[0] == want[0] and got[2] == want[2]
    assert np.float32(got[1]) == np.float32(want[1])
