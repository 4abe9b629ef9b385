/* Source-compatibility shim for the reference's include/half.h: `half`, HALF_COUNT, init_half()
 * and the half_convert_* / half_lookup function-pointer globals live in canvas_hip.h. */
#ifndef fluggo_half
#define fluggo_half
#include "canvas_hip.h"
#endif
