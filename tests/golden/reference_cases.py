"""The cases of tests/golden/reference/*.npz (pure data: imported by the generator make_reference_vectors.py and by
tests/test_reference_vectors.py).  name -> (scene, scene seed, W, H, max depth, rng seed, ReSTIR, script): make_golden.CASES' format.
W % 16 == 0 and H % 8 == 0: the reference's blocked field layout (pathtracer.py:74)."""
ALL_CASES = {
    "ref_s1_32x16_d4": ("s1", 0, 32, 16, 4, 0, False, [("accumulate", 2)]),
    "ref_sunlit_48x24_d5": ("sunlit", 0, 48, 24, 5, 11, False, [("accumulate", 2), ("end_frame",), ("still", 2), ("accumulate", 1)]),
    "ref_sunlit_restir_32x24_d4": ("sunlit", 0, 32, 24, 4, 7, True, [("accumulate", 2)]),
    "ref_sunlit_moving_32x16_d4": ("sunlit", 0, 32, 16, 4, 9, False, [("accumulate", 2), ("end_frame",), ("move", 0.45), ("accumulate", 1),
                                                                        ("end_frame",), ("still", 3), ("accumulate", 1)]),
    "ref_dense_32x16_d3": ("dense", 12345, 32, 16, 3, 3, False, [("accumulate", 1)]),
    # the dense case of tests/golden (make_golden.CASES["dense_48x40_d5"]) as the reference's source computes it: five bounces, three passes
    "ref_dense_48x40_d5": ("dense", 12345, 48, 40, 5, 3, False, [("accumulate", 3)]),
    # ReSTIR through a camera move: spatial reuse on the half-resolution pass, albedo demodulation (pathtracer.py:981-982), Catmull-Rom history
    "ref_sunlit_restir_moving_32x24_d4": ("sunlit", 0, 32, 24, 4, 17, True, [("accumulate", 2), ("end_frame",), ("move", 0.44), ("accumulate", 1),
                                                                               ("end_frame",), ("still", 3), ("accumulate", 2)]),
    # the example-6-style scene (eleven materials of the CSV, a small sun, no voxel edges) at six bounces, without its sky
    "ref_s6_plain_48x24_d6": ("s6", 0, 48, 24, 6, 29, False, [("accumulate", 3)]),
    # the physical sky's LOOKUP (atmos.py:94-131: jittered direction, wrapped bilinear fetch; NEE and shift() transmittance) on given tables
    "ref_s6_sky_lookup_32x16_d4": ("s6", 0, 32, 16, 4, 23, True, [("accumulate", 2)], ("given", 64, 5)),
    # the sky / cloud PRECOMPUTE (atmos.py: the 256x128 transmittance LUT, the cloud ambient, 3 cloud passes, 4 slices of the two skybox tables)
    # at 8 x 8 texels -- the table size is an instance attribute of Atmos, set from outside (3840 in the reference: hours of Python per column)
    "ref_s6_sky_precompute_16x8_d3": ("s6", 0, 16, 8, 3, 21, False, [("accumulate", 1)], (8, 3, 4)),
}

# the tests run the cases whose fixture is committed (the generator knows them all: a case takes minutes to an hour of Python)
import os as _os
_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "reference")
CASES = {k: v for k, v in ALL_CASES.items() if _os.path.exists(_os.path.join(_DIR, k + ".npz"))}
