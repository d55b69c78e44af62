"""The experiment switches of DESIGN.md §4 (environment GSR_<NAME>, gsr_set_variant) select alternative kernels for A/B
measurements; none may change a result beyond the documented bars.  One mid-size scene through every alternative, against the
oracle: forward bit-exact (lists in the mode the switch combination builds), gradients at the element-wise bar of helpers.py."""
import numpy as np
import pytest

from helpers import assert_forward_bit_exact, hip_backward_raw, hip_forward_raw, oracle_forward, scene, two_stage_check
from oracle import oracle

pytestmark = pytest.mark.gpu

DEFAULTS = dict(QUAD=1, CULL=1, BIN=2, LPT=1, MAILBOX=1, BWD_FASTEXP=1, BWD_NQ=0, KEYS=0, WAVE8X8=1, EARLY=1, BANDS=8,
                GB_BLOCK=128, PRE_DIRECT=1, GB_DIRECT=1, GB_RK=1, BWD_PADLDS=0, FWD_PADLDS=0, GB_BALANCED=1, MID_HINT=1, EMIT_SCAN=1,
                BLEND_LOG=1, BLEND_LOG_LIST=0)
ALTERNATIVES = [
    dict(QUAD=0),                 # forward: every wave walks the whole tile list
    dict(QUAD=0, CULL=0),         # ... over the reference's lists
    dict(BIN=1),                  # whole-image slices instead of bands (culled lists)
    dict(BIN=0),                  # 64-bit radix sort as the reference (reference lists by construction)
    dict(LPT=0),                  # row-major tile launch
    dict(MAILBOX=0),              # num_rendered through memcpy + stream sync
    dict(EARLY=0),                # no early launch of the binning head
    dict(BWD_FASTEXP=0),          # gs_expf in the backward
    dict(BWD_NQ=1),               # backward: four waves per tile (one quadrant each)
    dict(BWD_NQ=2),               # backward: two waves per tile (a 16x8 half each)
    dict(BWD_NQ=1, BWD_FASTEXP=0), dict(BWD_NQ=2, CULL=0),
    dict(KEYS=1),                 # sorted keys written during the forward
    dict(WAVE8X8=0),              # forward: 16x4 pixel strips per wave
    dict(BANDS=16),
    dict(GB_BLOCK=256),
    dict(PRE_DIRECT=0),           # SH rows through the LDS tile in preprocess
    dict(GB_DIRECT=0),            # ... and in the per-Gaussian backward
    dict(GB_RK=2),                # two gradient rows per thread and chunk
    dict(GB_BLOCK=256, GB_RK=2),
    dict(GB_BALANCED=0),          # per-Gaussian backward: 128 consecutive Gaussians per workgroup, every thread adds its own rows
    dict(GB_BALANCED=0, GB_BLOCK=256),
    dict(GB_BALANCED=0, GB_DIRECT=0),
    dict(EMIT_SCAN=0),            # block scan as its own launch (scan_blocks_kernel) instead of inside emit_partition_kernel
    dict(EMIT_SCAN=0, MAILBOX=0),
    dict(EMIT_SCAN=1, MAILBOX=0),
    dict(MID_HINT=0),             # tile_sort_mid_kernel launched every frame
    dict(MID_HINT=2),             # ... never: lists of more than 2048 entries are left to tile_sort_small_kernel's in-place network
    dict(BLEND_LOG=0),            # no blend log: the backward re-derives every blend decision (render_backward_kernel, rounds 1-4)
    dict(BLEND_LOG=0, BWD_NQ=1), dict(BLEND_LOG=0, BWD_NQ=2, CULL=0), dict(BLEND_LOG=0, BWD_FASTEXP=0),
    dict(BLEND_LOG=1, BWD_NQ=1), dict(BLEND_LOG=1, BWD_NQ=2), dict(BLEND_LOG=1, CULL=0),  # the log with every work split / the reference's lists
    dict(BLEND_LOG_LIST=1),       # the optional list-length limit says "lists too long": no log
    dict(BWD_NQ=2, BWD_PADLDS=6144),  # the occupancy experiments' switches: fewer workgroups per CU, same results
    dict(FWD_PADLDS=20000, BWD_PADLDS=9728),
]


@pytest.mark.parametrize("alt", ALTERNATIVES, ids=lambda a: ",".join(f"{k}={v}" for k, v in a.items()))
def test_alternative_kernels_keep_the_results(alt):
    from based_on_point_cloud_partitions_amd import _lib

    g, cam = scene(12000, 330, 250, 210.0, seed=23, scale_lo=0.008, scale_hi=0.12)
    bg = (0.1, 0.6, 0.2)
    st = oracle_forward(g, cam, bg=bg)
    dL = np.random.default_rng(8).standard_normal((3, 250, 330)).astype(np.float32)
    try:
        for k, v in alt.items():
            _lib.set_variant(k, v)
        ctx = hip_forward_raw(g, cam, bg=bg)
        assert_forward_bit_exact(ctx, st)
        two_stage_check(st, hip_backward_raw(ctx, dL, capture=True), dL)
    finally:
        for k in alt:
            _lib.set_variant(k, DEFAULTS[k])
