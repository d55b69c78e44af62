"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol the header declares,
the Python surface has the reference's names, and the product refuses to run without a HIP device."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gsrast_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes

    from based_on_point_cloud_partitions_amd import _lib

    names = _declared()
    assert len(names) >= 14
    lib = ctypes.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gsrast_mi355x.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names  # the ctypes binding covers the whole header
    assert _lib.load().gsr_abi_version() == _lib.ABI_VERSION


def test_size_queries():
    from based_on_point_cloud_partitions_amd import _lib

    lib = _lib.load()
    assert lib.gsr_geometry_bytes(500000) >= 500000 * (48 + 24 + 3 + 4 + 4)
    assert lib.gsr_binning_bytes(1000, 100, 64, 64) >= 1000 * 24
    assert lib.gsr_backward_scratch_bytes(1000, 64, 64) >= 36000  # nine floats per gradient row (one row per instance at most)
    assert lib.gsr_image_bytes(1920, 1080) >= 1920 * 1080 * 8 + 8160 * 8
    assert lib.gsr_knn_scratch_bytes(1000) > 0 and lib.gsr_sort_scratch_bytes(1000) > 0
    # the R-dependent part of the binning buffer: (depth, id) pairs + a second u64 array + point_list + the radix path's
    # second value array and block histograms = 24 B per instance + 4.25 B for the radix path, + the blend log's four 64-bit lane masks
    # per list position (32 B, round 5: render_forward.hip) -- nothing else
    P, W, H = 500_000, 1920, 1080
    per_instance = (lib.gsr_binning_bytes(40_000_000, P, W, H) - lib.gsr_binning_bytes(0, P, W, H)) / 40_000_000
    assert 56.0 <= per_instance <= 60.5, per_instance
    assert lib.gsr_backward_scratch_bytes(40_000_000, W, H) <= 36 * 40_000_000 + 256


def test_set_variant_reports_failure():
    from based_on_point_cloud_partitions_amd import _lib

    lib = _lib.load()
    assert lib.gsr_set_variant(b"ABI_TEST_SWITCH", 3) == 0 and lib.gsr_get_variant(b"ABI_TEST_SWITCH", 0) == 3
    assert lib.gsr_set_variant(b"A_NAME_OF_MORE_THAN_TWENTY_THREE_CHARACTERS", 1) == -1
    assert b"cannot be stored" in lib.gsr_last_error()
    with pytest.raises(_lib.GsrError):
        _lib.set_variant("A_NAME_OF_MORE_THAN_TWENTY_THREE_CHARACTERS", 1)


def test_operator_surface_matches_reference_names():
    import inspect
    import sys

    sys.path.insert(0, os.path.join(ROOT, "based_on_point_cloud_partitions_amd", "dropin"))
    import diff_gaussian_rasterization as dgr
    from simple_knn._C import distCUDA2  # noqa: F401

    # diff_gaussian_rasterization/__init__.py:157-169
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
        "campos", "prefiltered", "debug")
    sig = inspect.signature(dgr.GaussianRasterizer.forward)
    assert list(sig.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations",
                                    "cov3D_precomp"]
    assert hasattr(dgr.GaussianRasterizer, "markVisible") and hasattr(dgr, "rasterize_gaussians")
    for n in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"):  # ext.cpp:15-19
        assert hasattr(dgr._C, n)


def _settings():
    from based_on_point_cloud_partitions_amd import GaussianRasterizationSettings

    z = torch.zeros(3)
    return GaussianRasterizationSettings(16, 16, 1.0, 1.0, z, 1.0, torch.eye(4), torch.eye(4), 0, z, False, False)


def test_argument_validation_matches_reference():
    from based_on_point_cloud_partitions_amd import GaussianRasterizer

    r = GaussianRasterizer(_settings())
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(x, x, torch.zeros(4, 1), scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(x, x, torch.zeros(4, 1), shs=torch.zeros(4, 16, 3))


def test_no_cpu_fallback():
    from based_on_point_cloud_partitions_amd import GaussianRasterizer, GsrError, distCUDA2

    r = GaussianRasterizer(_settings())
    x = torch.zeros(4, 3)
    with pytest.raises(GsrError, match="no CPU fallback"):
        r(x, x, torch.zeros(4, 1), shs=torch.zeros(4, 16, 3), scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(GsrError, match="no CPU fallback"):
        distCUDA2(torch.zeros(10, 3))


def test_shipped_library_has_no_ablation_or_extra_flags():
    """GSR_ABLATE / GSR_BIN_ABLATE (timing experiments that give wrong results on purpose) are 0 in the built .so and it was built
    without make EXTRA=... (tools/ab_build.sh + tools/ab_run.sh builds such variants elsewhere)."""
    from based_on_point_cloud_partitions_amd import _lib

    assert not os.environ.get("GSR_LIBRARY")
    assert _lib.build_info() == f"abi={_lib.ABI_VERSION} ablate=0 bin_ablate=0 extra=[]"


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "based_on_point_cloud_partitions_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, os.path.join(dp, f)


def test_graft_entry_build_runs_and_checks_the_abi_version():
    """build() is the driver's "does it build" check; it must not pin a stale ABI number."""
    import __graft_entry__ as g

    g.build()


def test_rendered_estimate_follows_the_gaussian_count():
    """The wrapper sizes the binning buffer before the forward from the largest num_rendered seen for (device, P, W, H); after a
    densification step P is new, and the estimate is the latest entry of the same device and image size scaled by P."""
    from based_on_point_cloud_partitions_amd import rasterizer as rz

    saved = dict(rz._R_SEEN)
    try:
        rz._R_SEEN.clear()
        assert rz._guess_rendered((0, 1000, 640, 480)) == 0
        rz._remember_rendered((0, 1000, 640, 480), 8000)
        rz._remember_rendered((0, 1000, 640, 480), 6000)          # the maximum is kept
        assert rz._guess_rendered((0, 1000, 640, 480)) == 8000
        assert rz._guess_rendered((0, 1500, 640, 480)) == 12001   # unseen P: scaled
        assert rz._guess_rendered((0, 1500, 320, 240)) == 0       # another image size: no basis
        assert rz._guess_rendered((1, 1500, 640, 480)) == 0       # another device
        rz._remember_rendered((0, 2000, 640, 480), 30000)
        assert rz._guess_rendered((0, 4000, 640, 480)) == 60001   # the most recent entry is the basis
        assert rz._guess_rendered((0, 4001, 640, 480)) == 0       # more than twice any known P: another scene, no estimate
        assert rz._guess_rendered((0, 450, 640, 480)) == 0
        assert rz._guess_rendered((0, 500, 640, 480)) == 4001     # (from the 1000-Gaussian entry: the 2000 one is too far)
    finally:
        rz._R_SEEN.clear()
        rz._R_SEEN.update(saved)
