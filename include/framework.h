/* Source-compatibility shim: code written against the reference's include/framework.h
 * (video path only; GL, audio and codec declarations are not part of this library).
 * Like the reference's header (framework.h:26-35) it defines EXPORT and brings in glib, so the
 * CPython layer of the reference (src/process, which uses GRWLock, GArray, gpointer ...) keeps
 * compiling; the library itself does not need glib, so its absence is not an error here. */
#ifndef fluggo_framework
#define fluggo_framework
#include <inttypes.h>
#include "half.h"
#if defined(WINNT)
#define EXPORT __attribute__((dllexport))
#else
#define EXPORT __attribute__((visibility("default")))
#endif
#if defined(__has_include)
#if __has_include(<glib.h>)
#include <glib.h>
#endif
#endif
#include "canvas_hip.h"
#endif
