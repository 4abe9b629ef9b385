#!/usr/bin/env python3
"""The "required edits" of INTEGRATION.md section 3 as a recipe: take one file of the Canvas tree's
src/process (the CPython layer that stays), return the text a maintainer ends up with after removing
what libcanvas_hip.so no longer has behind it -- the GL render path (removed, not wrapped) and the
audio / clock / codec holders that are not part of this library.

Nothing of the reference is stored here: the recipe names identifiers, not text.  It works on whole
top-level items of a C file:

  1. a top-level item (function definition, typedef, static variable) that mentions one of REMOVED
     is dropped;
  2. the names of the functions dropped that way are then removed where other items mention them:
     a designated initialiser `.slot = (cast) name`, an element `{ "name", (PyCFunction) name, ... }`
     of a method table, a prototype or a call statement `name( ... );`.

usage: degl_reference.py <in.c> [<out.c>]      (prints to stdout without <out.c>)
       degl_reference.py --list <in.c>         (names of the items dropped, one per line)
"""
import re
import sys

# Identifiers with nothing behind them any more.  GL: include/framework.h:254-330,405-466 of the reference
# (rgba_frame_gl, video_filter_program, the gl_* context functions, *_gl entry points -- except the two forced pulls
# video_get_frame_f16_gl / _f32_gl, which the library keeps as "pull through the device slot"); audio / clock / codec:
# pyframework.h:54-66,95-119.
REMOVED = [
    r"rgba_frame_gl", r"video_filter_program", r"video_(?!get_frame_f(?:16|32)_gl\b)\w+_gl\w*", r"video_\w*_gl_\w+", r"getCurrentGLContext",
    r"gl_\w*shader_state", r"gl_create_offscreen_context", r"gl_destroy_offscreen_context",
    r"gl_set_current_context", r"gl_ensure_context", r"GLEW_\w+", r"GLuint", r"GLint", r"GLenum",
    r"AudioSourceHolder", r"audio_source", r"audio_frame", r"AudioFrameSourceFuncs",
    r"PresentationClockHolder", r"presentation_clock", r"CodecPacketSourceHolder", r"codec_packet_source",
]
# init_* calls of src/process/main.c for files that leave the link (audio, clock, codec packets, the GL-only MPEG-2 filter)
DROPPED_UNITS = ["AudioSource", "CodecPacketSource", "SystemPresentationClock", "AudioPassThroughFilter",
                 "AudioWorkspace", "AudioFrame", "MPEG2SubsampleFilter"]

_removed_re = re.compile(r"\b(?:%s)\b" % "|".join(REMOVED))


def _blank_comments_and_strings(text):
    """Same length as text, comments / string / char literals replaced by spaces (newlines kept)."""
    out = list(text)
    i, n = 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            j = text.find("\n", i)
            j = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            j = n if j < 0 else j + 2
        elif c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            j += 1
        else:
            i += 1
            continue
        for k in range(i, min(j, n)):
            if out[k] != "\n":
                out[k] = " "
        i = j
    return "".join(out)


def _top_level_items(text):
    """[(start, end)] of the top-level items: up to a ';' or a closing '}' at brace depth 0 (a '}' followed by
    struct body or an initialiser, as in `typedef struct { ... } name;` or `= { ... };`, runs on to the ';').
    Preprocessor lines are items of their own."""
    code = _blank_comments_and_strings(text)
    items, i, n, start = [], 0, len(text), 0
    depth, body = 0, False
    while i < n:
        c = code[i]
        if depth == 0 and c == "#" and code[start:i].strip() == "":
            j = i
            while True:
                j = code.find("\n", j)
                if j < 0:
                    j = n
                    break
                if code[j - 1] != "\\":
                    break
                j += 1
            items.append((start, j + 1 if j < n else n))
            i = start = items[-1][1]
            continue
        if c == "{":
            if depth == 0:
                body = code[start:i].rstrip().endswith(")")      # a function body, not a struct / initialiser
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0 and body:
                items.append((start, i + 1))
                start = i + 1
        elif c == ";" and depth == 0:
            items.append((start, i + 1))
            start = i + 1
        i += 1
    if start < n:
        items.append((start, n))
    return items


_fn_name_re = re.compile(r"\b([A-Za-z_]\w*)\s*\([^;{]*\)\s*\{", re.S)


def edit(text):
    """-> (edited text, [names of the functions and types dropped])"""
    # the vtable slot itself: `.get_frame_gl = (video_get_frame_gl_func) name` (framework.h:193 of the reference)
    text = re.sub(r"[ \t]*\.get_frame_gl\s*=[^,}]*(?:,[ \t]*\n?|(?=\}))", "", text)
    code = _blank_comments_and_strings(text)
    keep, dropped = [], []
    for a, b in _top_level_items(text):
        item_code = code[a:b]
        if item_code.lstrip().startswith("#") or not _removed_re.search(item_code):
            keep.append([a, b])
            continue
        m = _fn_name_re.search(item_code)
        if m:
            dropped.append(m.group(1))
        else:
            m = re.search(r"\}\s*(\w+)\s*;\s*$", item_code) or re.search(r"\b(\w+)\s*(?:=|;)", item_code)
            if m:
                dropped.append(m.group(1))
    out = "".join(text[a:b] for a, b in keep)
    names = [d for d in dropped if d]
    names += ["init_" + u for u in DROPPED_UNITS]
    if names:
        alt = "|".join(re.escape(x) for x in names)
        # designated initialiser `.slot = (cast) name,`
        out = re.sub(r"[ \t]*\.\w+\s*=\s*(?:\([^)]*\)\s*)?(?:%s)\s*,?[ \t]*\n" % alt, "", out)
        # method-table element `{ "x", (PyCFunction) name, FLAGS,\n "doc" },`
        out = re.sub(r"[ \t]*\{\s*\"[^\"]*\"\s*,\s*(?:\([^)]*\)\s*)?(?:%s)\s*,(?:[^{}\"]|\"(?:[^\"\\]|\\.)*\")*\}\s*,?[ \t]*\n" % alt, "", out)
        # prototype or call statement on a line of its own
        out = re.sub(r"(?m)^[ \t]*(?:[A-Za-z_][\w \t\*]*\s+)?(?:%s)\s*\([^;{}]*\)\s*;[ \t]*\n" % alt, "", out)
    return out, dropped


def main(argv):
    if len(argv) >= 3 and argv[1] == "--list":
        print("\n".join(edit(open(argv[2]).read())[1]))
        return 0
    if len(argv) < 2:
        print(__doc__)
        return 2
    out, _ = edit(open(argv[1]).read())
    if len(argv) > 2:
        open(argv[2], "w").write(out)
    else:
        sys.stdout.write(out)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
