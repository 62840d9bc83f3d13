"""Per-op DEVICE tests: the drivers of tests/emu/emu_full.cpp compiled with hipcc -DOPS_DEVICE launch the
real gfx950 kernels - convolutions of every kind / input mode / tiling (ragged 24x24 and 8x32 tiles, K split, pose-vector
and concatenated sources), norm finalize (Instance / GroupNorm + FiLM, near-zero variance), gemv, attention (large
logits), the warp + blend tail with offsets beyond every border and the bilinear x2 edges - against torch fp64.
The test bodies are those of tests/test_emu_full.py (CPU emulator); only the library differs."""
import ctypes as C
import os

import pytest

import test_emu_full as T

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lib():
    import torch  # noqa: F401  (its HIP runtime first, as for libtha4_hip.so)
    import __graft_entry__ as g
    return C.CDLL(g.build_ops_device())          # no-op when the in-tree .so is newer than its sources


@pytest.mark.parametrize("case", T.FUSED_CASES)
def test_fused_norm_convs_on_device(lib, case):
    T.test_conv_with_fused_norm(lib, case)


@pytest.mark.parametrize("case", T.CASES)
def test_conv_kernels_on_device(lib, case):
    T.test_conv_kernel_matches_torch(lib, case)


def test_norm_finalize_on_device(lib):
    T.test_norm_finalize_instance_and_group(lib)
    T.test_norm_finalize_near_zero_variance(lib)


def test_gemv_and_attention_on_device(lib):
    T.test_gemv_and_attention(lib)
    T.test_attention_large_logits(lib)


@pytest.mark.parametrize("size", [256, 512])
def test_warp_blend_tail_borders_on_device(lib, size):
    T.check_warp_blend_tail(lib, size)


def test_upscaler_input_on_device(lib):
    T.test_upscaler_input_bilinear_edges_and_warp(lib)


# ---- round 5 -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pg_prod,pg_cons,h,w,c0,c1,cmid", [(12, 12, 32, 32, 32, 0, 32), (54, 52, 32, 32, 16, 16, 64), (11, 14, 32, 32, 48, 0, 32), (22, 12, 16, 16, 64, 0, 32)])
def test_moment_accumulators_on_device(lib, pg_prod, pg_cons, h, w, c0, c1, cmid):
    """The producers' integer-atomic moment accumulators and the consumer folding them (a tuning-only plan option: measured neutral, csrc/full_net.h
    acc_planned) with the REAL device atomics: the workgroups of the producing launch arrive in any order, the sums must not depend on it."""
    T.test_moment_accumulators_producer_to_consumer(lib, pg_prod, pg_cons, h, w, c0, c1, cmid)


@pytest.mark.parametrize("split", [1, 0])
@pytest.mark.parametrize("tiles", [1024, 1031, 72])
def test_norm_finalize_many_tiles_on_device(lib, tiles, split):
    T.test_norm_finalize_many_tiles_narrow_split(lib, tiles, split)


def test_xcd_aware_order_on_device(lib):
    """conv_tile_kernel on the 1-D XCD-aware grid (ConvArgs::xcd_remap) - the three remap cases are also members of T.CASES; here the launch counter
    proves the device harness took the remapped grid for them."""
    T.test_xcd_aware_order_is_taken_where_the_product_takes_it(lib)
