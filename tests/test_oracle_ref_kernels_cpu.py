"""Kernel-level pinning of the CPU oracle against the reference's OWN native code.

tests/golden/ref_kernel_golden.npz holds the outputs of the reference's unmodified raymarching.cu / gridencoder.cu / shencoder.cu /
freqencoder.cu (compiled for gfx950 by oracle/build_ref.py, run on an MI355X by tests/golden/make_golden_ref_kernels.py) on the seeded
`small` case table of tests/ref_kernel_cases.py.  radnerf_oracle.c must reproduce them: bit for bit for Morton codes, bitfields, the slab
test, every marcher variant (cascades 1 and 2, noise, dt_gamma 0), the packed training march, fp32 grid features and dy_dx, the march and
frequency-encoder backward; within the stated ulp-scale tolerances where the kernels use fast intrinsics (__expf, __sinf), atomics
(table gradients) or half accumulation.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_kernel_cases as rkc  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kernel_golden.npz")
NAMES = ("_raymarching_face", "_gridencoder", "_shencoder", "_freqencoder")


@pytest.fixture(scope="module")
def golden_ref():
    return np.load(GOLDEN)


@pytest.fixture(scope="module")
def orc_mods(oracle_mod):
    from oracle import ref_backends
    saved = {n: sys.modules.get(n) for n in NAMES}
    ref_backends.install()
    mods = {n: sys.modules[n] for n in NAMES}
    for n, m in saved.items():                       # do not leak the stand-in modules into other tests
        if m is None:
            sys.modules.pop(n, None)
        else:
            sys.modules[n] = m
    return mods


_CASES = rkc.cases("small")


def test_fixture_covers_every_case(golden_ref):
    have = {k.split("/")[0] for k in golden_ref.files}
    assert have == {c.name for c in _CASES}
    # every pybind entry point of the four reference extensions is exercised
    fns = {c.fn for c in _CASES}
    assert fns == {"near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "morton3D_dilation", "march_rays", "composite_rays",
                   "march_rays_train", "march_rays_train_backward", "composite_rays_train_forward", "composite_rays_train_backward", "grid_encode_forward",
                   "grid_encode_backward", "grad_total_variation", "sh_encode_forward", "sh_encode_backward", "freq_encode_forward", "freq_encode_backward"}


@pytest.mark.parametrize("case", _CASES, ids=lambda c: c.name)
def test_oracle_matches_reference_kernels(case, golden_ref, orc_mods):
    got = rkc.run_case(case, orc_mods, "cpu", f32_only=True)
    ref = {}
    for k in golden_ref.files:
        name, key = k.split("/")
        if name == case.name:
            ref[int(key) if key.isdigit() else key] = golden_ref[k]
    assert set(ref) == set(got)
    rkc.compare(case, got, ref, "oracle vs reference kernels (fixture)")
