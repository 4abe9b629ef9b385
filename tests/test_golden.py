"""The oracle against its frozen answers (tests/golden/cprocess_small.npz, made by tests/golden/make_golden.py),
and -- on the GPU box -- the library against the same file.  See make_golden.py for what these vectors pin."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.util import assert_same_f16, canon_f16, canon_f32

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "cprocess_small.npz"))


def test_oracle_reproduces_every_golden_vector(orc, golden):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden
    fresh = make_golden.cases()
    assert sorted(fresh) == sorted(golden.files)
    for k in golden.files:
        a, b = fresh[k], golden[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if a.dtype == np.float32:
            assert np.array_equal(canon_f32(a), canon_f32(b)), k
        elif a.dtype == np.uint16:
            assert np.array_equal(canon_f16(a), canon_f16(b)), k
        else:
            assert np.array_equal(a, b), k


@pytest.fixture(scope="module")
def full_size_sums():
    import json
    with open(os.path.join(HERE, "golden", "full_size_sha256.json")) as f:
        return json.load(f)


def test_oracle_reproduces_full_size_checksums(orc, full_size_sums):
    """Configs 2 and 3 at 3840x2160 (the 8K and the graph cases are left to make_checksums.py: they take half a minute)."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_checksums
    names = ["config2_3840x2160", "config3_3840x2160_to_1920x1080"]
    fresh = make_checksums.checksums(only=names)
    for n in names:
        assert fresh[n] == full_size_sums[n], n
    # the checker's other build (clang, a * b + c fused): what the library's contracted flavour is held to
    fresh = make_checksums.checksums(only=names[:1], flavour="contracted")
    assert fresh[names[0] + "@contracted"] == full_size_sums[names[0] + "@contracted"]
    assert full_size_sums[names[0] + "@contracted"]["sha256"] != full_size_sums[names[0]]["sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("flavour", ["separate", "contracted"])
def test_library_reproduces_full_size_checksums(full_size_sums, flavour):
    """Every BASELINE config at its full size, frame 0, in both arithmetic flavours: the library's output hashes to the
    committed value (no oracle run on the GPU box: the fixtures were made by the oracle's two builds in the build container)."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_checksums import canon_sha256
    from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth
    from canvas_amd.device import DeviceFrame, chain_color_over
    from canvas_amd.stream import GraphStream
    lib = _lib.load()
    assert lib.cvs_init(0) == 0
    lib.init_half()
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    f32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    sfx = "@contracted" if flavour == "contracted" else ""
    lib.cvs_set_arithmetic(_lib.ARITH_CONTRACTED if sfx else _lib.ARITH_SEPARATE)
    try:
        got = _render_every_config(lib, _lib, m, f32p)
    finally:
        lib.cvs_set_arithmetic(_lib.ARITH_SEPARATE)
    assert sorted(n + sfx for n in got) == sorted(n for n in full_size_sums if n.endswith("@contracted") == bool(sfx))
    for name, arr in got.items():
        assert list(arr.shape) == full_size_sums[name + sfx]["shape"], name
        assert canon_sha256(arr) == full_size_sums[name + sfx]["sha256"], name + sfx


def _render_every_config(lib, _lib, m, f32p):
    from canvas_amd import synth
    from canvas_amd.device import DeviceFrame, chain_color_over
    from canvas_amd.stream import GraphStream

    def chain(w, h, n, matrix, pre):
        dl = [DeviceFrame.from_host(synth.layer_frame(w, h, k, 0)) for k in range(n)]
        out = DeviceFrame((0, 0, w - 1, h - 1), np.uint16)
        chain_color_over([(out, dl)], matrix, pre, _lib.LUT_NONE)
        _lib.check(lib.cvs_stream_sync(None))
        assert lib.cvs_chain_last_was_fused() == 1
        got = out.download().array
        for d in dl + [out]:
            d.free()
        return got

    got = {"config2_3840x2160": chain(3840, 2160, 2, m, _lib.LUT_REC709_TO_LINEAR_SCENE)}
    taps = synth.gaussian_taps(9, 1.5)
    src = DeviceFrame.from_host(synth.layer_frame(3840, 2160, 1, 0))
    small = DeviceFrame((0, 0, 1919, 1079), np.uint16)
    _lib.check(lib.cvs_blur_lanczos_f16_dev(small.ref(), src.ref(), f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, None))
    got["config3_3840x2160_to_1920x1080"] = small.download().array
    src.free(); small.free()
    got["config4_7680x4320"] = chain(7680, 4320, 3, None, _lib.LUT_NONE)
    g = GraphStream(3840, 2160, ring=1)
    out = g.render(0)
    _lib.check(lib.cvs_stream_sync(None))
    got["config5_3840x2160"] = out.download().array
    return got


@pytest.mark.gpu
def test_library_reproduces_golden_chain_and_tables(golden):
    from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth
    from canvas_amd.device import DeviceFrame, chain_color_over
    lib = _lib.load()
    assert lib.cvs_init(0) == 0
    lib.init_half()
    for k in range(4):
        tab = np.ctypeslib.as_array(lib.cvs_lut_host(k), shape=(65536,))
        assert_same_f16(tab, golden["lut%d" % k], "lut %d" % k)
    out = np.empty(golden["f2h_in"].shape, np.uint16)
    _lib.half_pointer("half_convert_from_float")(out.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                 golden["f2h_in"].ctypes.data_as(C.POINTER(C.c_float)), out.size)
    assert np.array_equal(out, golden["f2h_out"])
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    for nl in (2, 3, 4):
        dl = [DeviceFrame.from_host(synth.layer_frame(64, 36, k, 0)) for k in range(nl)]
        res = DeviceFrame((0, 0, 63, 35), np.uint16)
        chain_color_over([(res, dl)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
        _lib.check(lib.cvs_stream_sync(None))
        assert_same_f16(res.download().array, golden["chain%d" % nl], "chain %d" % nl)
    taps = synth.gaussian_taps(9, 1.5)
    src = DeviceFrame.from_host(synth.layer_frame(96, 54, 1, 0))
    small = DeviceFrame((0, 0, 47, 26), np.uint16)
    _lib.check(lib.cvs_blur_lanczos_f16_dev(small.ref(), src.ref(), taps.ctypes.data_as(C.POINTER(C.c_float)), 9,
                                            C.c_float(0.5), C.c_float(0.5), 3, None))
    assert_same_f16(small.download().array, golden["config3_96x54"], "config 3")


@pytest.mark.gpu
def test_library_reproduces_golden_graph_and_edges(golden):
    from canvas_amd import _lib, synth
    from canvas_amd.abi import HostFrame
    from canvas_amd.stream import GraphStream
    lib = _lib.load()
    assert lib.cvs_init(0) == 0
    lib.init_half()
    g = GraphStream(96, 54, ring=1, first_frame=1)
    out = g.render(0)
    _lib.check(lib.cvs_stream_sync(None))
    assert_same_f16(out.download().array, golden["config5_96x54"], "config 5 graph")
    disp = HostFrame((0, 0, 31, 15), np.uint16, golden["display_in"])
    for tag, pre, mode in [("rgba8_srgb", _lib.LUT_LINEAR_TO_SRGB, _lib.DISPLAY_RGBA8), ("argb32", _lib.LUT_NONE, _lib.DISPLAY_ARGB32_PREMUL)]:
        got = np.zeros((16, 32), np.uint32)
        _lib.check(lib.video_frame_to_bytes(got.ctypes.data, disp.ref(), pre, mode))
        assert np.array_equal(got, golden["display_" + tag]), tag
    got = np.zeros((16, 32), np.uint32)
    _lib.check(lib.video_frame_to_rgba8_intent(got.ctypes.data, disp.ref(), _lib.LUT_LINEAR_TO_SRGB, C.c_float(1.25)))
    assert np.array_equal(got, golden["display_widget"]), "widget"
    # the widget's ramp itself, through an identity frame of all codes (no transfer table)
    allc = HostFrame((0, 0, 127, 127), np.uint16, np.arange(65536, dtype=np.uint16).reshape(128, 128, 4))
    got = np.zeros((128, 128), np.uint32)
    _lib.check(lib.video_frame_to_rgba8_intent(got.ctypes.data, allc.ref(), _lib.LUT_NONE, C.c_float(1.25)))
    assert np.array_equal(got.view(np.uint8).reshape(-1), golden["widget_ramp_125"]), "widget ramp"
    img = _lib.coded_image()
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden
    planes = make_golden.dv_planes()
    for p, s in enumerate((720, 180, 180)):
        img.data[p], img.stride[p], img.line_count[p] = planes[p].ctypes.data, s, 480
    dv = HostFrame((0, -1, 719, 478), np.uint16)
    lib.video_reconstruct_dv(dv.ref(), C.byref(img))
    assert_same_f16(dv.array[101:105], golden["dv_frame_rows_100_103"], "DV reconstruct")
    back = lib.video_subsample_dv(dv.ref())
    try:
        for p, (key, s) in enumerate((("dv_back_y_rows", 720), ("dv_back_cb_rows", 180), ("dv_back_cr_rows", 180))):
            plane = np.ctypeslib.as_array(C.cast(back.contents.data[p], C.POINTER(C.c_uint8)), shape=(480, s))
            assert np.array_equal(plane[101:105], golden[key]), key
    finally:
        C.CFUNCTYPE(None, C.c_void_p)(back.contents.free_func)(C.cast(back, C.c_void_p))
