"""north_star's tolerance, tested: "within 1 ulp in half of the reference src/cprocess CPU path".

The reference has two builds (SConstruct:46-48,75-83): gcc -std=c99, which rounds every multiply and add on its own, and
clang (preferred when installed), which contracts a * b + c inside an expression into one fused multiply-add.  The same
restatement compiled both ways gives the two flavours of the oracle (oracle/Makefile): liboracle.so and liboracle_fma.so.

  * CPU: on the colour + over chain (configs 2 and 4) the two flavours differ in a few values per hundred thousand, by
    one half code -- "1 ulp in half" is the width of the reference's own build-to-build spread.  One exception class,
    measured: where the Y'PbPr matrix cancels (Pb, Pr near zero: the result is orders of magnitude smaller than the
    terms it is the sum of) one f32 rounding difference is many half codes of the tiny result, and the two BUILDS OF THE
    REFERENCE land tens of codes apart -- on the BASELINE input too (3840 x 256 px: 48 codes at worst), more so where a
    small blended alpha then divides (1 value in 147 456 on random-alpha input, 31 codes, |difference| 7e-6 at 4e-4).
    A second, milder class: the colour filter's result is truncated to half BEFORE the stack (color.c:132), so a
    one-code difference there can come out of the blend and the final truncation as two codes.
    The bound below is therefore: at most one code apart, except for at most 1 value in 10 000, each of those within
    one half-ulp of the frame's value range in absolute terms (2^-10 of the largest magnitude);
  * GPU: the library's output is bit-equal to the gcc flavour and inside the same bound from the clang flavour.
"""
import numpy as np
import pytest

from canvas_amd import REC709_RGB_TO_YPBPR, synth
from canvas_amd.abi import HostFrame
from tests.util import canon_f16

M = np.array(REC709_RGB_TO_YPBPR, np.float32)


def half_code_distance(a, b):
    """Distance in half codes on the number line (sign-magnitude codes mapped to a monotonic integer); NaNs must pair."""
    a, b = canon_f16(a).astype(np.int32), canon_f16(b).astype(np.int32)
    nan_a, nan_b = (a & 0x7FFF) > 0x7C00, (b & 0x7FFF) > 0x7C00
    assert np.array_equal(nan_a, nan_b)
    lin = lambda c: np.where(c & 0x8000, -(c & 0x7FFF), c & 0x7FFF)       # noqa: E731
    d = np.abs(lin(a) - lin(b))
    d[nan_a] = 0
    return d


def assert_within_build_spread(orc, a, b, what):
    d = half_code_distance(a, b)
    far = d > 1
    assert far.sum() <= max(1, d.size // 10000), (what, int(far.sum()), int(d.max()))
    if far.any():
        va, vb = orc.half_to_float(a).astype(np.float64), orc.half_to_float(b).astype(np.float64)
        assert np.abs(va - vb)[far].max() <= 2.0 ** -10 * max(1.0, np.abs(va[np.isfinite(va)]).max()), what
    return d


def layers_for(case, w, h):
    rng = np.random.default_rng(77)
    if case == "baseline":                   # BASELINE's synthetic input: opaque bottom layer
        return [synth.layer_frame(w, h, k, 0) for k in range(3)]
    if case == "translucent":                # random alpha everywhere: every divide of the over operator is live
        return [HostFrame((0, 0, w - 1, h - 1), np.uint16, synth.layer_pixels(w, h, k, 3, opaque_base=False)) for k in range(3)]
    # wide range: negatives, values above 1, tiny alphas
    out = []
    for k in range(3):
        px = rng.normal(0.3, 1.5, (h, w, 4)).astype(np.float32)
        px[..., 3] = rng.uniform(0, 1, (h, w)) ** 3
        out.append(HostFrame((0, 0, w - 1, h - 1), np.uint16, px.astype(np.float16).view(np.uint16)))
    return out


def both_flavours(orc, layers, matrix, lut):
    want_gcc = orc.chain_color_over(layers, matrix, lut, None).array
    with orc.flavour("fma"):
        want_fma = orc.chain_color_over(layers, matrix, None if lut is None else orc.transfer_table(0), None).array
    return want_gcc, want_fma


@pytest.mark.parametrize("case", ["baseline", "translucent", "wide"])
@pytest.mark.parametrize("config", [2, 4])
def test_the_two_reference_builds_differ_by_at_most_one_half_code(orc, case, config):
    w, h = 256, 144
    layers = layers_for(case, w, h)[:2 if config == 2 else 3]
    gcc, fma = both_flavours(orc, layers, M if config == 2 else None, orc.transfer_table(0) if config == 2 else None)
    d = assert_within_build_spread(orc, gcc, fma, (case, config))
    assert d.max() >= 1, "contraction changed nothing here: the case does not exercise the tolerance"


def test_transfer_tables_barely_depend_on_the_flavour(orc):
    """Two of the four transfer functions contain an a * powf(x) - b (gammatab.c:58-66,201-211), which clang contracts:
    measured here, ONE of the 4 x 65536 entries differs between the builds (linear -> Rec.709, code 0x789b), by one
    code.  The two tables config 2 can use as pre-tables (Rec.709 -> linear) are flavour-independent."""
    differing = 0
    for which in range(4):
        a = orc.transfer_table(which)
        with orc.flavour("fma"):
            b = orc.transfer_table(which)
        d = half_code_distance(a, b)
        assert d.max() <= 1, which
        if which in (0, 1):
            assert d.max() == 0, which
        differing += int((d > 0).sum())
    assert differing <= 8


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["baseline", "translucent", "wide"])
@pytest.mark.parametrize("config,w,h", [(2, 256, 144), (4, 256, 144), (2, 3840, 2160), (4, 7680, 4320)])
def test_library_is_bit_equal_to_gcc_build_and_within_one_code_of_clang_build(cvs, orc, case, config, w, h):
    from canvas_amd import _lib
    from canvas_amd.device import DeviceFrame, chain_color_over
    if w > 256:
        if case != "baseline":
            pytest.skip("full size: the BASELINE input only")
        h = 256 if config == 2 else 128            # the top rows of the full-width frame: the oracle finishes in seconds
    layers = layers_for(case, w, h)[:2 if config == 2 else 3]
    gcc, fma = both_flavours(orc, layers, M if config == 2 else None, orc.transfer_table(0) if config == 2 else None)
    dl = [DeviceFrame.from_host(l) for l in layers]
    out = DeviceFrame((0, 0, w - 1, h - 1), np.uint16)
    chain_color_over([(out, dl)], M if config == 2 else None, _lib.LUT_REC709_TO_LINEAR_SCENE if config == 2 else _lib.LUT_NONE, _lib.LUT_NONE)
    _lib.check(cvs.cvs_stream_sync(None))
    assert cvs.cvs_chain_last_was_fused() == 1
    got = out.download().array
    assert np.array_equal(canon_f16(got), canon_f16(gcc)), "not bit-equal to the gcc / no-contraction build of the reference"
    assert_within_build_spread(orc, got, fma, "library vs the clang / contraction build")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["baseline", "translucent", "wide"])
@pytest.mark.parametrize("config,w,h", [(2, 256, 144), (4, 256, 144), (2, 3840, 2160), (4, 7680, 4320)])
def test_contracted_library_is_bit_equal_to_clang_build_and_not_to_gcc_build(cvs, orc, case, config, w, h):
    """cvs_set_arithmetic(CVS_ARITH_CONTRACTED): the mirror image of the test above -- bit-equal to the contraction-on build
    of the reference, and (on inputs where the two builds differ at all) NOT equal to the other one: the flavour is in force."""
    from canvas_amd import _lib
    from canvas_amd.device import DeviceFrame, chain_color_over
    if w > 256:
        if case != "baseline":
            pytest.skip("full size: the BASELINE input only")
        h = 256 if config == 2 else 128
    layers = layers_for(case, w, h)[:2 if config == 2 else 3]
    gcc, fma = both_flavours(orc, layers, M if config == 2 else None, orc.transfer_table(0) if config == 2 else None)
    dl = [DeviceFrame.from_host(l) for l in layers]
    out = DeviceFrame((0, 0, w - 1, h - 1), np.uint16)
    assert cvs.cvs_set_arithmetic(_lib.ARITH_CONTRACTED) == _lib.ARITH_SEPARATE
    try:
        assert cvs.cvs_get_arithmetic() == _lib.ARITH_CONTRACTED
        chain_color_over([(out, dl)], M if config == 2 else None, _lib.LUT_REC709_TO_LINEAR_SCENE if config == 2 else _lib.LUT_NONE, _lib.LUT_NONE)
        _lib.check(cvs.cvs_stream_sync(None))
        assert cvs.cvs_chain_last_was_fused() == 1
    finally:
        cvs.cvs_set_arithmetic(_lib.ARITH_SEPARATE)
    got = out.download().array
    assert np.array_equal(canon_f16(got), canon_f16(fma)), "not bit-equal to the clang / contraction-on build of the reference"
    if not np.array_equal(canon_f16(gcc), canon_f16(fma)):
        assert not np.array_equal(canon_f16(got), canon_f16(gcc)), "the contracted flavour computed the gcc build's pixels"
    assert_within_build_spread(orc, got, gcc, "contracted library vs the gcc build")
