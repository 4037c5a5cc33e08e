"""No-GPU tests of the C-ABI library: it loads, exports every symbol include/mof.h declares,
its geometry helpers follow the reference constructors, and it refuses to run without a device."""
import ctypes as C
import os
import re

import pytest

from mrs_optic_flow_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mof.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mof_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mof.h but not exported"
    assert sorted(_capi.SYMBOLS) == declared, "ctypes table and header disagree"
    assert lib.mof_version().startswith(b"mof-hip")


def test_fft_reference_geometry():
    lib = _capi.load()
    cfg = _capi.FftConfig()
    # FftMethod.cpp:1706-1720: odd frame size made even, sqNum = fs / sps
    _capi.check(lib.mof_fft_config_reference(C.byref(cfg), 481, 120, 80.0))
    assert (cfg.frame_width, cfg.patch_size, cfg.grid_x, cfg.grid_y, cfg.stride_x) == (480, 120, 4, 4, 120)
    # not a multiple -> one window
    _capi.check(lib.mof_fft_config_reference(C.byref(cfg), 480, 100, 80.0))
    assert (cfg.patch_size, cfg.grid_x, cfg.grid_y) == (480, 1, 1)
    assert lib.mof_fft_config_reference(None, 480, 100, 80.0) == _capi.MOF_ERR_BAD_ARG
    assert b"geometry" in lib.mof_last_error()


def test_bm_geometry_helpers():
    lib = _capi.load()
    cfg = _capi.BmConfig()
    _capi.check(lib.mof_bm_config_block_method(C.byref(cfg), 272, 32, 8))
    assert (cfg.grid_x, cfg.grid_y, cfg.step_size, cfg.low_contrast_rule) == (8, 8, 0, 0)
    _capi.check(lib.mof_bm_config_fast_spaced(C.byref(cfg), 752, 480, 16, 8, 16))
    assert (cfg.grid_x, cfg.grid_y, cfg.low_contrast_rule) == (30, 18, 1)


def test_argument_validation_happens_before_any_device_use():
    lib = _capi.load()
    h = C.c_void_p()
    bad = _capi.FftConfig(2000, 2000, 1000, 2, 2, 0, 0, 1000, 1000, 80.0, 0)  # pads to 1000: beyond the planned transforms (<= 960)
    assert lib.mof_fft_create(C.byref(bad), C.byref(h)) == _capi.MOF_ERR_UNSUPPORTED
    bad = _capi.FftConfig(752, 480, 62, 8, 8, 0, 0, 90, 59, 80.0, 0, 1, 55)  # the OpenCL model cannot plan 62 = 2 * 31 (nor can the reference)
    assert lib.mof_fft_create(C.byref(bad), C.byref(h)) == _capi.MOF_ERR_UNSUPPORTED
    bad = _capi.FftConfig(752, 480, 1, 8, 8, 0, 0, 90, 59, 80.0, 0)
    assert lib.mof_fft_create(C.byref(bad), C.byref(h)) == _capi.MOF_ERR_BAD_ARG
    ok = _capi.FftConfig(752, 480, 60, 8, 8, 0, 0, 90, 59, 80.0, 0)  # any other patch size has a kernel: only the device is missing here
    assert lib.mof_fft_create(C.byref(ok), C.byref(h)) in (_capi.MOF_ERR_NO_DEVICE, _capi.MOF_OK)
    if h:
        lib.mof_fft_destroy(h)
        h = C.c_void_p()
    bad = _capi.FftConfig(752, 480, 64, 8, 8, 0, 0, 100, 59, 80.0, 0)  # 7*100+64 > 752
    assert lib.mof_fft_create(C.byref(bad), C.byref(h)) == _capi.MOF_ERR_BAD_ARG
    assert b"leaves the frame" in lib.mof_last_error()
    badb = _capi.BmConfig(752, 480, 18, 8, 16, 30, 18, 1, 0)  # block not a multiple of 4
    assert lib.mof_bm_create(C.byref(badb), C.byref(h)) == _capi.MOF_ERR_UNSUPPORTED
    assert lib.mof_fft_process(None, None, 0, None, None) == _capi.MOF_ERR_NOT_INIT


def test_no_cpu_fallback_without_a_device():
    lib = _capi.load()
    if lib.mof_device_count() > 0:
        pytest.skip("a HIP device is present")
    from mrs_optic_flow_amd import FftMethod, MofError

    with pytest.raises(MofError) as exc:
        FftMethod(448, 64)
    assert exc.value.code == _capi.MOF_ERR_NO_DEVICE


def test_shard_partition_is_the_contiguous_ceil_split():
    """mof_shard_partition: rank g of G takes pairs [g ceil(B/G), min(B, (g+1) ceil(B/G))) -- the same split as
    mrs_optic_flow_amd.sharding.shard_bounds -- for G in {1, 2, 4, 8} and ragged B. Pure host arithmetic: no device needed."""
    from mrs_optic_flow_amd import sharding

    lib = _capi.load()
    for G in (1, 2, 3, 4, 8):
        for B in (0, 1, 7, 8, 9, 37, 1000, 1024, 8191, 8192):
            slab = lib.mof_shard_slab_pairs(B, G)
            assert slab == -(-B // G)
            covered = 0
            for g in range(G):
                first, count = C.c_int(-1), C.c_int(-1)
                assert lib.mof_shard_partition(B, G, g, C.byref(first), C.byref(count)) == _capi.MOF_OK
                lo, hi = sharding.shard_bounds(B, g, G)
                assert (first.value, first.value + count.value) == (lo, hi) or (count.value == 0 and hi == lo)
                covered += count.value
            assert covered == B
    f, c = C.c_int(), C.c_int()
    assert lib.mof_shard_partition(8, 4, 4, C.byref(f), C.byref(c)) == _capi.MOF_ERR_BAD_ARG
    assert lib.mof_shard_partition(-1, 4, 0, C.byref(f), C.byref(c)) == _capi.MOF_ERR_BAD_ARG
    assert lib.mof_shard_fft_devices(None) == 0 and lib.mof_shard_fft_sync(None) == _capi.MOF_ERR_NOT_INIT
    # r05: the explicit gather set-up and the block-matching group, on null groups (no device needed)
    assert lib.mof_shard_fft_init_gather(None) == _capi.MOF_ERR_NOT_INIT and lib.mof_shard_fft_gather_ready(None) == 0
    assert lib.mof_shard_bm_devices(None) == 0 and lib.mof_shard_bm_sync(None) == _capi.MOF_ERR_NOT_INIT
    assert lib.mof_shard_bm_init_gather(None) == _capi.MOF_ERR_NOT_INIT and lib.mof_shard_bm_slab_bytes(None, 8) == 0
    assert lib.mof_shard_bm_locate(None, 8, 0, None, None, None) == _capi.MOF_ERR_NOT_INIT
    h = C.c_void_p()
    assert lib.mof_shard_bm_create(None, None, 2, C.byref(h)) == _capi.MOF_ERR_BAD_ARG and not h
