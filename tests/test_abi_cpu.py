"""CPU-side checks of the product library: it loads, exports every symbol include/canvas_hip.h
declares, fails loudly without a GPU, and its host-only logic (FIR taps, workspace items, time
helpers, window helpers) is right.  No pixel work happens here.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from canvas_amd import _lib
from canvas_amd.abi import HostFrame, fir_filter, rational

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "canvas_hip.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
    funcs = re.findall(r"CVS_EXPORT\s+[^;(]*?\b(\w+)\s*\(", text)
    ptrs = re.findall(r"CVS_EXPORT\s+extern\s+\w+\s*\(\*(\w+)\)", text)
    return sorted(set(funcs) - {"void"}), sorted(set(ptrs))


def test_every_declared_symbol_is_exported(lib):
    funcs, ptrs = declared_symbols()
    assert len(funcs) > 80 and len(ptrs) == 5
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    missing = [s for s in funcs + ptrs if s not in exported]
    assert not missing, "declared in canvas_hip.h but not exported: %s" % missing
    # and the ctypes table covers the same set, so no binding silently rots
    assert sorted(_lib.SIGNATURES) == funcs
    assert sorted(_lib.HALF_POINTER_GLOBALS) == ptrs


def test_header_compiles_as_c_and_cpp(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "framework.h"\n#include "half.h"\n'
                   "int main(void){ box2i a, b, r; box2i_set(&a,0,0,3,3); box2i_set_empty(&b); box2i_union(&r,&a,&b);\n"
                   " rgba_frame_f16 f; (void)f; return sizeof(video_frame_source_funcs) == 32 && sizeof(rgba_f16) == 8 ? 0 : 1; }\n")
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++14")):
        exe = tmp_path / ("t_" + cc)
        subprocess.run([cc, std, "-Wall", "-Werror", "-x", "c" if cc == "gcc" else "c++", str(src), "-I", os.path.join(ROOT, "include"),
                        "-o", str(exe)], check=True)
        assert subprocess.run([str(exe)]).returncode == 0


def test_struct_layouts_match_the_reference_abi():
    # include/framework.h:46-75,155-194: sizes and offsets a compiled consumer depends on
    from canvas_amd.abi import box2i, rgba_frame_f16, rgba_frame_f32, video_frame_source_funcs, video_source
    assert C.sizeof(box2i) == 16 and C.sizeof(rational) == 8
    assert C.sizeof(rgba_frame_f16) == 40 and rgba_frame_f16.full_window.offset == 8 and rgba_frame_f16.current_window.offset == 24
    assert C.sizeof(rgba_frame_f32) == 40
    assert C.sizeof(video_frame_source_funcs) == 32 and video_frame_source_funcs.get_frame.offset == 8
    assert video_frame_source_funcs.get_frame_32.offset == 16 and video_frame_source_funcs.get_frame_dev.offset == 24
    assert C.sizeof(video_source) == 16 and C.sizeof(fir_filter) == 16


def test_no_gpu_means_loud_failure_not_fallback(lib):
    if lib.cvs_device_count() > 0:
        pytest.skip("a GPU is present; the no-device behaviour is exercised in the CPU container")
    assert lib.cvs_init(0) != 0
    assert "no CPU path" in _lib.last_error()
    # frame functions signal failure the way the reference does: empty current_window (main.c:35-38)
    a = HostFrame((0, 0, 3, 3), np.float32, fill=1.0)
    b = HostFrame((0, 0, 3, 3), np.float32, fill=0.5)
    before = a.array.copy()
    lib.video_mix_over_f32(a.ref(), b.ref(), C.c_float(1.0))
    assert a.current_window.is_empty()
    assert np.array_equal(a.array, before)              # and nothing was computed on the CPU
    assert lib.cvs_malloc(16) is None
    lib.init_half()
    out = np.full(4, 7.0, np.float32)
    codes = np.array([0x3C00] * 4, np.uint16)
    _lib.half_pointer("half_convert_to_float")(out.ctypes.data_as(C.POINTER(C.c_float)), codes.ctypes.data_as(C.POINTER(C.c_uint16)), 4)
    assert (out == 7.0).all()


@pytest.mark.parametrize("sub", [0.25, 0.5, 0.75, 1.0, 2.0, 3.5])
@pytest.mark.parametrize("offset", [0.0, 0.25, 0.5, 0.999])
def test_fir_generators_equal_oracle(lib, orc, sub, offset):
    # parameter-sized host math (filter.c), no GPU involved
    def mine(fn, *args):
        f = fir_filter(None, 0, 0)
        fn(*args, C.byref(f))
        taps = np.ctypeslib.as_array(f.coeff, shape=(f.width,)).copy()
        c = f.center
        lib.filter_free(C.byref(f))
        return taps, c
    t, c = mine(lib.filter_createTriangle, C.c_float(sub), C.c_float(offset))
    ot, oc = orc.fir_triangle(sub, offset)
    assert c == oc and np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    t, c = mine(lib.filter_createLanczos, C.c_float(sub), 3, C.c_float(offset))
    ot, oc = orc.fir_lanczos(sub, 3, offset)
    assert c == oc and np.array_equal(t.view(np.uint32), ot.view(np.uint32))


def test_frame_time_helpers(lib):
    # src/cprocess/main.c:23-31
    ntsc = rational(30000, 1001)
    assert lib.get_frame_time(C.byref(ntsc), 0) == 1
    assert lib.get_frame_time(C.byref(ntsc), 30) == (30 * 10 ** 9 * 1001) // 30000 + 1
    for f in (0, 1, 29, 30, 1000, 123456):
        assert lib.get_time_frame(C.byref(ntsc), lib.get_frame_time(C.byref(ntsc), f)) == f
    t0 = lib.gettime()
    assert lib.gettime() >= t0 > 0


def test_workspace_item_bookkeeping(lib):
    # workspace.c:309-492: add / get (ordered by x then z) / update / remove
    ws = lib.workspace_create()
    a = lib.workspace_add_item(ws, 101, 10, 5, 0, 3, 1001)
    b = lib.workspace_add_item(ws, 102, 0, 20, 7, 9, 1002)
    c = lib.workspace_add_item(ws, 103, 10, 5, 0, -1, 1003)
    assert lib.workspace_get_length(ws) == 3
    assert [lib.workspace_get_item(ws, i) for i in range(3)] == [b, c, a]      # x=0; then x=10 z=-1, z=3
    assert lib.workspace_get_item_source(b) == 102 and lib.workspace_get_item_tag(c) == 1003
    assert lib.workspace_get_item_offset(b) == 7
    x, ln, z = C.c_int64(), C.c_int64(), C.c_int64()
    lib.workspace_get_item_pos(a, C.byref(x), C.byref(ln), C.byref(z))
    assert (x.value, ln.value, z.value) == (10, 5, 3)
    nx = C.c_int64(-4)
    lib.workspace_update_item(a, C.byref(nx), None, None, None, None, None)
    assert lib.workspace_get_item(ws, 0) == a
    lib.workspace_set_item_offset(a, 42)
    assert lib.workspace_get_item_offset(a) == 42
    lib.workspace_remove_item(b)
    assert lib.workspace_get_length(ws) == 2
    lib.workspace_free(ws)


def test_workspace_random_edit_fuzz(lib):
    """tests/process/video/VideoWorkspace.py:12-38 without the pulls: 10 000 random edits must keep
    the item list consistent (ordered, right length)."""
    import random
    rnd = random.Random(1)
    ws = lib.workspace_create()
    live = []
    for _ in range(10000):
        action = rnd.randint(1, 7)
        if action <= 4 and live:
            it = rnd.choice(live)
            v = C.c_int64(rnd.randint(-20, 1000))
            args = [None] * 6
            args[{1: 0, 2: 2, 3: 1, 4: 3}[action]] = C.byref(v)
            if action == 3:
                v.value = rnd.randint(1, 100)
            lib.workspace_update_item(it, *args)
        elif action == 5 and live:
            it = live.pop(rnd.randrange(len(live)))
            lib.workspace_remove_item(it)
        else:
            live.append(lib.workspace_add_item(ws, 1, rnd.randint(0, 1000), rnd.randint(1, 100), rnd.randint(-20, 20), rnd.randint(-10, 10), None))
        assert lib.workspace_get_length(ws) == len(live)
    keys = []
    x, ln, z = C.c_int64(), C.c_int64(), C.c_int64()
    for i in range(len(live)):
        lib.workspace_get_item_pos(lib.workspace_get_item(ws, i), C.byref(x), C.byref(ln), C.byref(z))
        keys.append((x.value, z.value))
    assert keys == sorted(keys)
    lib.workspace_free(ws)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under canvas_amd/ may import, link or call it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "canvas_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"\boracle\b|liboracle|orc_", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    out = subprocess.run(["ldd", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out


def test_hot_kernels_keep_their_state_in_registers():
    """The FIR ring, the chain's pixel sets and the divide paths index their register arrays with compile-time
    constants only; one runtime index would move an array to scratch memory (private segment) and cost a factor
    of 2-3 without failing any parity test.  Read the code objects' metadata: no hot kernel may use scratch or spill."""
    import glob
    import re
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    build = os.path.join(ROOT, "canvas_amd", "csrc", "build")
    if not (os.path.exists(readelf) and os.path.exists(objdump) and os.path.isdir(build)):
        pytest.skip("no ROCm LLVM tools or no object files in tree")
    import tempfile
    checked = hand = 0
    with tempfile.TemporaryDirectory(dir=build) as tmp:
        for obj in ("blur_ops.hip.o", "blur_long_ops.hip.o", "blur_even_ops.hip.o", "chain_ops.hip.o", "chain_deep_ops.hip.o", "color_ops.hip.o", "display_ops.hip.o", "resample_ops.hip.o", "sweep_ops.hip.o", "sweep_vh_ops.hip.o"):
            src = os.path.join(build, obj)
            if not os.path.exists(src):
                continue
            local = os.path.join(tmp, obj)
            with open(src, "rb") as f, open(local, "wb") as g:
                g.write(f.read())
            subprocess.run([objdump, "-d", "--offloading", obj], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            for co in glob.glob(os.path.join(tmp, obj + "*gfx950")):
                notes = subprocess.run([readelf, "--notes", co], stdout=subprocess.PIPE, text=True).stdout
                for name, scratch, spills in re.findall(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", notes, re.S):
                    if "k_chain_v0" in name or "k_chain_tail" in name:
                        continue            # not hot: the first version (mixed batches, huge frames) and the one-lane-per-frame
                                            # kernel for the last pixel of odd-sized frames
                    assert int(scratch) == 0 and int(spills) == 0, (name, scratch, spills)
                    checked += 1
                # sweep_ops.hip: an instance that fetches its rows through hand-written asm loads (last template argument
                # true) must have registers to spare -- under pressure hipcc moves values about, and a register with a load
                # in flight into it must not be touched before the hand-written wait
                for agprs, name, vgprs in re.findall(r"\.agpr_count:\s+(\d+)(?:(?!\.agpr_count).)*?\.name:\s+(\S*k_fir_lanes\S+)(?:(?!\.agpr_count).)*?\.vgpr_count:\s+(\d+)", notes, re.S):
                    if name.endswith("Lb1EEEv16cvk_fir2d_paramsi"):
                        assert int(agprs) == 0 and int(vgprs) <= 224, (name, agprs, vgprs)
                        hand += 1
                # sweep_vh_ops.hip: every instance fetches that way
                for agprs, name, vgprs in re.findall(r"\.agpr_count:\s+(\d+)(?:(?!\.agpr_count).)*?\.name:\s+(\S*k_fir_vh\S+)(?:(?!\.agpr_count).)*?\.vgpr_count:\s+(\d+)", notes, re.S):
                    assert int(agprs) == 0 and int(vgprs) <= 224, (name, agprs, vgprs)
                    hand += 1
    assert checked >= 20, checked
    assert hand >= 20, hand
