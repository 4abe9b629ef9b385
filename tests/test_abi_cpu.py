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
        for obj in ("blur_ops.hip.o", "blur_pair_ops.hip.o", "blur_halve_pair_ops.hip.o", "blur_long_ops.hip.o", "blur_even_ops.hip.o", "chain_ops.hip.o", "chain_deep_ops.hip.o", "color_ops.hip.o", "display_ops.hip.o", "tile_vh_ops.hip.o", "sweep_vh_ops.hip.o", "sweep_hv_ops.hip.o", "blur_halve_ops.hip.o",
                    "blur_ops.fma.hip.o", "blur_pair_ops.fma.hip.o", "blur_halve_pair_ops.fma.hip.o", "blur_halve_ops.fma.hip.o", "chain_ops.fma.hip.o", "color_ops.fma.hip.o", "sweep_vh_ops.fma.hip.o"):
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
    assert checked >= 20, checked


def _asm_checker():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_asm_loads", os.path.join(ROOT, "tools", "check_asm_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_kernel_touches_a_register_with_a_load_in_flight():
    """The hand-pipelined kernel (k_chain, in both arithmetic flavours) issues loads from inline asm and wait by hand; hipcc believes the
    loaded value exists when the asm statement ends and may copy or reuse its register before the wait (round 2: wrong
    pixels, then a fault; round 3: the checker found sixteen such copies in k_chain<6 layers, pre-LUT>).  The rule is
    checked on the disassembly of EVERY kernel of the library -- the same check the Makefile runs as a build step."""
    chk = _asm_checker()
    build = os.path.join(ROOT, "canvas_amd", "csrc", "build")
    import glob
    objs = sorted(glob.glob(os.path.join(build, "*.hip.o")))
    if not os.path.exists(chk.OBJDUMP):
        pytest.skip("no ROCm LLVM tools")
    assert len(objs) >= 15, "no kernel objects in tree: build first (python -c 'import __graft_entry__ as g; g.build()')"
    checked, problems = chk.check_paths(objs)
    assert checked >= 250, checked
    assert not problems, "\n".join(problems[:10])
    # the hand-pipelined ones were among them
    hand, _ = chk.check_paths([o for o in objs if os.path.basename(o) in ("chain_ops.hip.o", "chain_ops.fma.hip.o")], only=r"k_chainILi2ELi0ELb1ELb0")
    assert hand >= 2


_BROKEN = """
0000000000001000 <k_broken>:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0                         // 000000001000: C0060002 00000000
\ts_waitcnt lgkmcnt(0)                                       // 000000001008: BF8CC07F
\tglobal_load_dwordx4 v[4:7], v1, s[0:1]                     // 00000000100C: DC5C8000 04000001
\tv_mov_b32_e32 v10, 0                                       // 000000001014: 7E140280
\tv_mov_b64_e32 v[8:9], v[4:5]                               // 000000001018: 7E107104
\ts_waitcnt vmcnt(0)                                         // 00000000101C: BF8C0F70
\tv_add_f32_e32 v10, v8, v6                                  // 000000001020: 02140D08
\ts_endpgm                                                   // 000000001024: BF810000
"""


def test_the_checker_finds_a_copy_in_front_of_the_wait():
    """The shape of the fault: a register pair copied between the load into it and the wait.  And the shapes that are
    fine: a counted wait that retires the older of two loads, a second load into the same registers (in order), a loop
    that waits at its top for what it requested at its bottom, a structurizer flag that sends the exit path round the body."""
    chk = _asm_checker()
    bad = chk.parse(_BROKEN)
    problems = chk.check_function("k_broken", bad["k_broken"])
    assert len(problems) == 1 and "v_mov_b64_e32 v[8:9], v[4:5]" in problems[0] and "0x100c" in problems[0], problems

    fine = """
0000000000002000 <k_fine>:
\tglobal_load_dwordx2 v[2:3], v0, s[0:1]                     // 000000002000: DC548000 02000000
\tglobal_load_dwordx2 v[4:5], v0, s[2:3]                     // 000000002008: DC548000 04020000
\ts_waitcnt vmcnt(1)                                         // 000000002010: BF8C0F71
\tv_add_f32_e32 v6, v2, v3                                   // 000000002014: 020C0702
\tglobal_load_dwordx2 v[4:5], v0, s[4:5]                     // 000000002018: DC548000 04040000
\ts_mov_b64 s[10:11], 0                                      // 000000002020: BE8A0180
\ts_cmp_lt_i32 s6, s7                                        // 000000002024: BF040706
\ts_cbranch_scc1 2                                           // 000000002028: BF850002 <k_fine+0x34>
\ts_mov_b64 s[10:11], -1                                     // 00000000202C: BE8A01C1
\ts_branch 3                                                 // 000000002030: BF820003 <k_fine+0x40>
\ts_waitcnt vmcnt(0)                                         // 000000002034: BF8C0F70
\tv_add_f32_e32 v6, v4, v5                                   // 000000002038: 020C0B04
\tglobal_load_dwordx2 v[4:5], v0, s[4:5]                     // 00000000203C: DC548000 04040000
\ts_and_b64 vcc, exec, s[10:11]                              // 000000002040: 86EA0A7E
\ts_cbranch_vccnz 2                                          // 000000002044: BF870002 <k_fine+0x50>
\ts_branch 65531                                             // 000000002048: BF82FFFB <k_fine+0x34>
\ts_nop 0                                                    // 00000000204C: BF800000
\ts_endpgm                                                   // 000000002050: BF810000
"""
    ok = chk.parse(fine)
    assert chk.check_function("k_fine", ok["k_fine"]) == []

    # hipcc's structurizer: the exit path (loads still in flight) and the loop path (waited) meet in a shared block that
    # tests a flag; only with the flag known does the exit path not run on into the body
    flagged = """
0000000000003000 <k_flag>:
\tglobal_load_dwordx2 v[4:5], v0, s[0:1]                     // 000000003000: DC548000 04000000
\ts_cmp_lt_i32 s6, s7                                        // 000000003008: BF040706
\ts_cbranch_scc1 2                                           // 00000000300C: BF850002 <k_flag+0x18>
\ts_mov_b64 s[10:11], -1                                     // 000000003010: BE8A01C1
\ts_branch 2                                                 // 000000003014: BF820002 <k_flag+0x20>
\ts_waitcnt vmcnt(0)                                         // 000000003018: BF8C0F70
\ts_mov_b64 s[10:11], 0                                      // 00000000301C: BE8A0180
\ts_and_b64 vcc, exec, s[10:11]                              // 000000003020: 86EA0A7E
\ts_cbranch_vccnz 2                                          // 000000003024: BF870002 <k_flag+0x30>
\tv_add_f32_e32 v6, v4, v5                                   // 000000003028: 020C0B04
\ts_nop 0                                                    // 00000000302C: BF800000
\ts_endpgm                                                   // 000000003030: BF810000
"""
    assert chk.check_function("k_flag", chk.parse(flagged)["k_flag"]) == []
    leaky = flagged.replace("s_and_b64 vcc, exec, s[10:11]", "s_and_b64 vcc, exec, s[12:13]").replace("k_flag", "k_leaky")
    problems = chk.check_function("k_leaky", chk.parse(leaky)["k_leaky"])
    assert len(problems) == 1 and "v_add_f32_e32 v6, v4, v5" in problems[0], problems


# ---------------------------------------------------------------- round 4: arithmetic flavours, refusals that need no device

def test_every_contracted_launcher_has_its_twin():
    """kernels.h renames the launchers of the units built with -DCVS_CONTRACT to *_fma: every name of that list must be defined
    by the library in both forms, and the host must know both (a missing twin would be a link error at best)."""
    header = open(os.path.join(ROOT, "canvas_amd", "csrc", "kernels", "kernels.h")).read()
    renamed = re.findall(r"#define\s+(cvk_\w+)\s+(cvk_\w+_fma)\b", header)
    assert len(renamed) >= 20
    assert all(b == a + "_fma" for a, b in renamed)
    out = subprocess.run(["nm", "--defined-only", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    defined = set(re.findall(r"\b[Tt]\s+(cvk_\w+)", out))
    for a, b in renamed:
        assert a in defined and b in defined, (a, b)
        assert re.search(r"\b%s\s*\(" % b, header), "kernels.h does not declare " + b


def test_arithmetic_mode_is_a_process_wide_switch(lib):
    assert lib.cvs_get_arithmetic() in (_lib.ARITH_SEPARATE, _lib.ARITH_CONTRACTED)
    before = lib.cvs_set_arithmetic(_lib.ARITH_CONTRACTED)
    try:
        assert lib.cvs_get_arithmetic() == _lib.ARITH_CONTRACTED
        assert lib.cvs_set_arithmetic(7) == -1 and "unknown mode" in _lib.last_error()
        assert lib.cvs_get_arithmetic() == _lib.ARITH_CONTRACTED              # a refused mode changes nothing
        assert lib.cvs_set_arithmetic(_lib.ARITH_SEPARATE) == _lib.ARITH_CONTRACTED
    finally:
        lib.cvs_set_arithmetic(before)


def test_batch_entries_refuse_what_the_single_calls_refuse(lib):
    """the batch of cvs_blur_lanczos_f16_dev skipped the window check of the single entry -- a source whose
    current_window reaches outside its buffer must be refused before anything is launched (here: before a device is even
    looked for), every target marked empty; null entries of the arrays likewise."""
    from canvas_amd.abi import rgba_frame_f16
    fp16 = C.POINTER(rgba_frame_f16)
    taps = np.array([0.25, 0.5, 0.25], np.float32)
    tp = taps.ctypes.data_as(C.POINTER(C.c_float))
    good = [HostFrame((0, 0, 63, 35), np.uint16) for _ in range(2)]
    bad = HostFrame((0, 0, 63, 35), np.uint16, current_window=(0, 0, 64, 35))        # one column beyond the buffer
    outs = [HostFrame((0, 0, 31, 17), np.uint16) for _ in range(3)]
    srcs = (fp16 * 3)(C.pointer(good[0].c), C.pointer(bad.c), C.pointer(good[1].c))
    dsts = (fp16 * 3)(*[C.pointer(o.c) for o in outs])
    assert lib.cvs_blur_lanczos_f16_batch_dev(dsts, srcs, 3, tp, 3, C.c_float(0.5), C.c_float(0.5), 3, None) == -1
    assert "outside its buffer (frame 1)" in _lib.last_error()
    assert all(o.current_window.is_empty() for o in outs)
    holes = (fp16 * 3)(C.pointer(good[0].c), None, C.pointer(good[1].c))
    assert lib.cvs_blur_lanczos_f16_batch_dev(dsts, holes, 3, tp, 3, C.c_float(0.5), C.c_float(0.5), 3, None) == -1
    assert "null pointer" in _lib.last_error()
    # the scaler's batch entries: same two rules
    from canvas_amd.abi import v2f
    big = [HostFrame((0, 0, 127, 71), np.uint16) for _ in range(3)]
    bdst = (fp16 * 3)(*[C.pointer(o.c) for o in big])
    assert lib.cvs_scale_bilinear_f16_batch_dev(bdst, v2f(0, 0), srcs, v2f(0, 0), v2f(2.0, 2.0), 3, None) == -1
    assert "outside its buffer (frame 1)" in _lib.last_error()
    assert all(o.current_window.is_empty() for o in big)
    assert lib.cvs_scale_bilinear_f16_batch_dev(bdst, v2f(0, 0), holes, v2f(0, 0), v2f(2.0, 2.0), 3, None) == -1
    assert "null pointer" in _lib.last_error()
    assert lib.cvs_scale_bilinear_f16_batch_dev(bdst, v2f(0, 0), holes, v2f(0, 0), v2f(2.0, 2.0), 0, None) == 0      # nothing to do
    fp32 = C.POINTER(_lib.rgba_frame_f32_t)
    f32src = [HostFrame((0, 0, 63, 35), np.float32) for _ in range(2)]
    f32dst = [HostFrame((0, 0, 127, 71), np.float32) for _ in range(2)]
    s32 = (fp32 * 2)(C.pointer(f32src[0].c), None)
    d32 = (fp32 * 2)(*[C.pointer(o.c) for o in f32dst])
    assert lib.cvs_scale_bilinear_f32_batch_dev(d32, v2f(0, 0), s32, v2f(0, 0), v2f(2.0, 2.0), 2, None) == -1
    assert "null pointer" in _lib.last_error()
    # the blur + over batch: same two rules
    full = [HostFrame((0, 0, 63, 35), np.uint16) for _ in range(3)]
    fdst = (fp16 * 3)(*[C.pointer(o.c) for o in full])
    layers = (fp16 * 3)(*[C.pointer(g.c) for g in (good[0], good[1], good[0])])
    assert lib.cvs_blur_over_f16_batch_dev(fdst, srcs, tp, 3, layers, 1, 3, None) == -1
    assert "outside its buffer (frame 1)" in _lib.last_error()
    assert all(o.current_window.is_empty() for o in full)
    assert lib.cvs_blur_over_f16_batch_dev(fdst, holes, tp, 3, layers, 1, 3, None) == -1 and "null pointer" in _lib.last_error()


def test_forced_pull_of_a_source_without_a_device_slot_is_empty(lib):
    """src/cprocess/main.c:78-103: video_get_frame_f16_gl on a source with no slot-3 entry leaves the window empty.  Slot 3 is
    the device slot here; a host-only source pulled with force_gl=True must come back empty too, its pixel
    callback never called."""
    from canvas_amd.abi import GET_FRAME_F16, GET_FRAME_F32, video_frame_source_funcs, video_source
    called = []

    def get16(obj, idx, frame):
        called.append(idx)
    funcs = video_frame_source_funcs()
    funcs.flags = 0
    keep = GET_FRAME_F16(get16)
    funcs.get_frame = keep
    src = video_source(None, C.pointer(funcs))
    for dtype, fn in ((np.uint16, lib.video_get_frame_f16_gl), (np.float32, lib.video_get_frame_f32_gl)):
        out = HostFrame((0, 0, 7, 7), dtype)
        fn(C.byref(src), 3, out.ref())
        assert out.current_window.is_empty()
    assert called == []


def test_frame_to_device_rule_and_contexts_without_a_device(lib):
    """Several GPUs in one process: frame g belongs to context g mod n -- the same rule canvas_amd/shard.py
    uses across processes -- and a context cannot be chosen before it exists.  No device is needed for either."""
    from canvas_amd import shard
    for n in (1, 2, 3, 4, 8):
        owners = [lib.cvs_frame_owner(g, n) for g in range(-8, 40)]
        assert owners == [g % n for g in range(-8, 40)]
        for r in range(n):
            mine = [g for g in range(40) if lib.cvs_frame_owner(g, n) == r]
            assert mine == shard.frames_of_rank(r, n, len(mine)) and all(shard.owner_of_frame(g, n) == r for g in mine)
    assert lib.cvs_frame_owner(5, 0) == -1
    count = lib.cvs_context_count()
    assert lib.cvs_set_context(count + 3) == -2 and "no context" in _lib.last_error()
    if lib.cvs_device_count() == 0:
        assert count == 0 and lib.cvs_current_context() == -1 and lib.cvs_current_device() == -1
        assert lib.cvs_context_open(0) == -1 and "no CPU path" in _lib.last_error()
        assert lib.cvs_context_count() == 0
