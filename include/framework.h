/* Source-compatibility shim: code written against the reference's include/framework.h
 * (video path only; GL, audio and codec declarations are not part of this library). */
#ifndef fluggo_framework
#define fluggo_framework
#include "canvas_hip.h"
#define EXPORT CVS_EXPORT
#endif
