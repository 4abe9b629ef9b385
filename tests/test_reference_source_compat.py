"""The drop-in boundary as a maintainer meets it: the reference's CPython layer,
src/process/*.c, compiled where it lies against THIS repo's include/ -- framework.h, half.h and pyframework.h shadow the
reference's headers -- and linked against libcanvas_hip.so with no undefined symbol allowed.

Runs only where the reference tree is mounted (the build container; the GPU box has no /root/reference) and where glib's
headers exist.  Nothing of the reference is copied into the repo: files are read in place, the ones that need an edit go
through tools/degl_reference.py (the recipe INTEGRATION.md lists as "required edits") into pytest's tmp directory."""
import glob
import os
import subprocess
import sys
import sysconfig

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/process"
sys.path.insert(0, os.path.join(ROOT, "tools"))

# the video side of src/process (audio, clock, codec packets and the GL-only MPEG-2 filter leave the link)
NO_EDIT = ["AnimationFunc", "CodedImageSource", "DVSubsampleFilter", "FrameFuncPassThroughFilter", "RgbaFrameF16",
           "RgbaFrameF32", "VideoPullQueue", "VideoScaler", "VideoSource", "basetypes", "basicframefuncs"]
# file -> what the recipe has to take out ("required edits")
EDITED = {
    "DVReconstructionFilter": ["DVReconstructionFilter_get_frame_gl"],
    "EmptyVideoSource": ["EmptyVideoSource_getFrameGL"],
    "Pulldown23RemovalFilter": ["gl_shader_state", "destroy_shader", "Pulldown23RemovalFilter_getFrameGL"],
    "SolidColorVideoSource": ["gl_solid_color_shader_state", "destroy_shader", "SolidColorVideoSource_getFrameGL"],
    "VideoGainOffsetFilter": ["VideoGainOffsetFilter_get_frame_gl"],
    "VideoMixFilter": ["VideoMixFilter_getFrameGL"],
    "VideoPassThroughFilter": ["VideoPassThroughFilter_getFrameGL"],
    "VideoSequence": ["VideoSequence_getFrameGL"],
    "VideoWorkspace": ["Workspace_get_frame_gl"],
    "main": ["py_audio_take_source", "py_destroy_offscreen_gl_context", "py_create_offscreen_gl_context",
             "py_set_current_gl_context", "py_check_context_supported"],
}


def _glib_flags():
    for inc, cfg in (("/usr/include/glib-2.0", "/usr/lib/x86_64-linux-gnu/glib-2.0/include"),
                     ("/opt/conda/include/glib-2.0", "/opt/conda/lib/glib-2.0/include")):
        if os.path.exists(os.path.join(inc, "glib.h")) and os.path.exists(os.path.join(cfg, "glibconfig.h")):
            return ["-I" + inc, "-I" + cfg]
    return None


def _cc(args, **kw):
    return subprocess.run(["gcc"] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)


@pytest.fixture(scope="module")
def env():
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present")
    glib = _glib_flags()
    if glib is None:
        pytest.skip("no glib headers in this image")
    # -std=gnu99 as the reference builds (SConstruct:46); an implicit declaration = a call with nothing behind it
    flags = ["-std=gnu99", "-fPIC", "-fvisibility=hidden", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
             "-I" + os.path.join(ROOT, "include"), "-I" + sysconfig.get_paths()["include"]] + glib
    return flags


@pytest.mark.parametrize("name", NO_EDIT)
def test_reference_file_compiles_unchanged(env, name):
    """These files need no edit at all: they compile from where they lie, and the recipe leaves them byte-identical."""
    import degl_reference
    path = os.path.join(REF, name + ".c")
    text = open(path).read()
    edited, dropped = degl_reference.edit(text)
    assert edited == text and dropped == []
    p = _cc(env + ["-fsyntax-only", path])
    assert p.returncode == 0, p.stdout[-3000:]


@pytest.mark.parametrize("name", sorted(EDITED))
def test_reference_file_compiles_with_the_listed_edit(env, name, tmp_path):
    """These files name the removed GL slot (or, main.c, the audio holder and the GL context helpers): unchanged they
    must NOT compile -- that is what makes the edit 'required' -- and with exactly the recipe's removals they do."""
    import degl_reference
    path = os.path.join(REF, name + ".c")
    assert _cc(env + ["-fsyntax-only", path]).returncode != 0
    edited, dropped = degl_reference.edit(open(path).read())
    assert dropped == EDITED[name]
    assert "get_frame_gl" not in edited and "rgba_frame_gl" not in edited
    out = tmp_path / (name + ".c")
    out.write_text(edited)
    p = _cc(env + ["-fsyntax-only", "-Wall", str(out)])
    assert p.returncode == 0, p.stdout[-3000:]
    assert "warning" not in p.stdout, p.stdout[-3000:]         # e.g. a static left without its only user


def test_reference_python_layer_links_against_the_library_and_imports(env, tmp_path):
    """All 21 files -> objects -> one shared object against libcanvas_hip.so + glib + libpython with --no-undefined:
    every symbol the reference's CPython layer takes from src/cprocess on the video path is exported by the library
    under the same name.  Then the result is imported as fluggo.media.process (in a child process, beside this repo's
    fluggo.media.basetypes): its types construct, and a pull without a HIP device comes back with an empty window --
    the reference's error behaviour (src/cprocess/main.c:35-38) and no CPU fallback."""
    import degl_reference
    objs = []
    for name in NO_EDIT + sorted(EDITED):
        src = os.path.join(REF, name + ".c")
        if name in EDITED:
            out = tmp_path / (name + ".c")
            out.write_text(degl_reference.edit(open(src).read())[0])
            src = str(out)
        obj = str(tmp_path / (name + ".o"))
        p = _cc(env + ["-O1", "-Wno-deprecated-declarations", "-c", src, "-o", obj])
        assert p.returncode == 0, p.stdout[-3000:]
        objs.append(obj)
    pkg = tmp_path / "pkg" / "fluggo" / "media"
    pkg.mkdir(parents=True)
    (pkg.parent / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "basetypes.py").write_text(open(os.path.join(ROOT, "fluggo", "media", "basetypes.py")).read())
    so = str(pkg / ("process" + sysconfig.get_config_var("EXT_SUFFIX")))
    libdir = sysconfig.get_config_var("LIBDIR")
    glibdir = "/opt/conda/lib" if not glob.glob("/usr/lib/x86_64-linux-gnu/libglib-2.0.so") else "/usr/lib/x86_64-linux-gnu"
    p = _cc(["-shared", "-o", so] + objs + ["-L" + os.path.join(ROOT, "canvas_amd"), "-lcanvas_hip", "-L" + glibdir, "-lglib-2.0",
             "-L" + libdir, "-lpython" + sysconfig.get_config_var("LDVERSION"), "-lm", "-Wl,--no-undefined",
             "-Wl,-rpath," + os.path.join(ROOT, "canvas_amd")])
    undefined = [l for l in p.stdout.splitlines() if "undefined reference" in l]
    assert p.returncode == 0 and not undefined, "\n".join(undefined[:20]) or p.stdout[-3000:]

    script = r"""
import sys
sys.dont_write_bytecode = True
from fluggo.media import process
from fluggo.media.basetypes import box2i
need = ['VideoSource', 'RgbaFrameF16', 'RgbaFrameF32', 'SolidColorVideoSource', 'EmptyVideoSource', 'VideoGainOffsetFilter',
        'VideoMixFilter', 'VideoScaler', 'VideoPassThroughFilter', 'VideoSequence', 'VideoWorkspace', 'VideoPullQueue',
        'AnimationFunc', 'LerpFunc', 'LinearFrameFunc', 'get_frame_time', 'get_time_frame', 'time_get_frame']
missing = [n for n in need if not hasattr(process, n)]
assert not missing, missing
solid = process.SolidColorVideoSource((0.25, 0.5, 0.75, 1.0), box2i(0, 0, 3, 3))
seq = process.VideoSequence()
seq.append((solid, 0, 10))
ws = process.VideoWorkspace()
item = ws.add(source=process.VideoPassThroughFilter(seq), offset=0, x=2, length=5, z=1)
assert len(ws) == 1 and item.x == 2 and item.length == 5
mix = process.VideoMixFilter(src_a=solid, src_b=ws, mix_b=process.LerpFunc((0.0,), (1.0,), 10))
frame = mix.get_frame_f32(3, box2i(0, 0, 3, 3))
from canvas_amd_probe import device_present
if not device_present:
    cw = frame.current_window
    assert cw.max.x < cw.min.x or cw.max.y < cw.min.y, cw          # loud failure: empty window, no CPU path
print('reference layer over libcanvas_hip: ok')
"""
    (tmp_path / "pkg" / "canvas_amd_probe.py").write_text(
        "import os\ndevice_present = os.path.exists('/dev/kfd')\n")
    env_vars = dict(os.environ, PYTHONPATH=str(tmp_path / "pkg"), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", script], cwd=str(tmp_path), env=env_vars,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "reference layer over libcanvas_hip: ok" in r.stdout, r.stdout[-3000:]
