/*
 * halfconv.c -- the half<->float entry points of src/cprocess/half.c, and the transfer tables of
 * src/cprocess/gammatab.c, on top of the HIP kernels.
 *
 * half.c:87-105 exports five function-pointer VARIABLES that init_half() fills in; callers in the
 * reference (and third-party modules linked against it) go through those pointers, so they are
 * kept as pointers here.  Host buffers are staged through HBM; the `cvs_*_dev` twins take device
 * pointers directly.
 *
 * gammatab.c builds each table once as table[i] = f2h(func(h2f(i))) (:87-106, :132-155, :175-194,
 * :227-246).  Same here: h2f and f2h run on the GPU through the entry points above (so the tables
 * use exactly the conversions the pixel path uses), func is evaluated with the host's libm powf as
 * in the reference, and the result is kept both on the host and in HBM.
 */
#define _GNU_SOURCE
#include "internal.h"
#include <math.h>
#include <pthread.h>

/* ---------------------------------------------------------------- flat conversions */

CVS_EXPORT int cvs_half_to_float_dev(float *out, const half *in, size_t count, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_KERNEL(cvk_half_to_float(out, in, count, 0, cvs_pick_stream(s)));
    return 0;
}
CVS_EXPORT int cvs_float_to_half_dev(half *out, const float *in, size_t count, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_KERNEL(cvk_float_to_half(out, in, count, 0, cvs_pick_stream(s)));
    return 0;
}
CVS_EXPORT int cvs_half_to_float_fast_dev(float *out, const half *in, size_t count, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_KERNEL(cvk_half_to_float(out, in, count, 1, cvs_pick_stream(s)));
    return 0;
}
CVS_EXPORT int cvs_float_to_half_fast_dev(half *out, const float *in, size_t count, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_KERNEL(cvk_float_to_half(out, in, count, 1, cvs_pick_stream(s)));
    return 0;
}
CVS_EXPORT int cvs_half_lookup_dev(const half *table_dev, half *out, const half *in, size_t count, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_KERNEL(cvk_half_lookup(table_dev, out, in, count, cvs_cus(), cvs_pick_stream(s)));
    return 0;
}

/* host buffers: stage in, convert, stage out */
static int host_convert(void *out, size_t out_bytes, const void *in, size_t in_bytes, size_t count, int to_float, int fast) {
    if (cvs_enter() != 0) return -1;
    if (!count) return 0;
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged din, dout;
    int rc = cvs_stage_in(&din, in, in_bytes, 1, s);
    if (rc == 0) rc = cvs_stage_in(&dout, NULL, out_bytes, 0, s);
    else dout.dev = NULL;
    if (rc == 0) {
        rc = to_float ? cvk_half_to_float((float *)dout.dev, (const uint16_t *)din.dev, count, fast, s)
                      : cvk_float_to_half((uint16_t *)dout.dev, (const float *)din.dev, count, fast, s);
        if (rc != 0) cvs_set_error("conversion kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
    if (rc == 0) rc = cvs_stage_out(&dout, out, s);
    cvs_stage_free(&din);
    cvs_stage_free(&dout);
    return rc;
}

static void host_h2f(float *out, const half *in, int count) { if (count > 0) host_convert(out, (size_t)count * 4, in, (size_t)count * 2, (size_t)count, 1, 0); }
static void host_f2h(half *out, const float *in, int count) { if (count > 0) host_convert(out, (size_t)count * 2, in, (size_t)count * 4, (size_t)count, 0, 0); }
static void host_h2f_fast(float *out, const half *in, int count) { if (count > 0) host_convert(out, (size_t)count * 4, in, (size_t)count * 2, (size_t)count, 1, 1); }
static void host_f2h_fast(half *out, const float *in, int count) { if (count > 0) host_convert(out, (size_t)count * 2, in, (size_t)count * 4, (size_t)count, 0, 1); }

static void host_lookup(const half *table, half *out, const half *in, int count) {
    if (count <= 0 || cvs_enter() != 0) return;
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged dt, din, dout;
    dt.dev = din.dev = dout.dev = NULL;
    int rc = cvs_stage_in(&dt, table, HALF_COUNT * sizeof(half), 1, s);
    if (rc == 0) rc = cvs_stage_in(&din, in, (size_t)count * 2, 1, s);
    if (rc == 0) rc = cvs_stage_in(&dout, NULL, (size_t)count * 2, 0, s);
    if (rc == 0) {
        rc = cvk_half_lookup((const uint16_t *)dt.dev, (uint16_t *)dout.dev, (const uint16_t *)din.dev, (size_t)count, cvs_cus(), s);
        if (rc != 0) cvs_set_error("lookup kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
    if (rc == 0) cvs_stage_out(&dout, out, s);
    cvs_stage_free(&dt); cvs_stage_free(&din); cvs_stage_free(&dout);
}

/* before init_half() the reference's pointers are NULL (half.c:87-91) */
CVS_EXPORT void (*half_convert_to_float)(float *, const half *, int);
CVS_EXPORT void (*half_convert_from_float)(half *, const float *, int);
CVS_EXPORT void (*half_convert_to_float_fast)(float *, const half *, int);
CVS_EXPORT void (*half_convert_from_float_fast)(half *, const float *, int);
CVS_EXPORT void (*half_lookup)(const half *, half *, const half *, int);

CVS_EXPORT void init_half(void) {                                     /* half.c:93-105 */
    half_convert_to_float = host_h2f;
    half_convert_from_float = host_f2h;
    half_convert_to_float_fast = host_h2f_fast;
    half_convert_from_float_fast = host_f2h_fast;
    half_lookup = host_lookup;
}

/* ---------------------------------------------------------------- transfer tables */

/* `fl`: the arithmetic flavour (canvas_hip.h cvs_set_arithmetic).  Two of the four functions hold an a * pow - b in one
 * expression, which the reference's clang build fuses (measured between the two builds of the same C: ONE of the 4 x 65536 entries
 * differs, linear -> Rec.709 at code 0x789b); the other two are the same in both flavours and exist once. */
static float tf_rec709_to_linear(float in, int fl) {                  /* gammatab.c:48-56 */
    const float knee = 4.5f * 0.018f;
    (void)fl;
    return in < knee ? in / 4.5f : powf((in + 0.099f) / 1.099f, 1.0f / 0.45f);
}
static float tf_rec709_display(float in, int fl) {                    /* gammatab.c:145-150 */
    (void)fl;
    return in < 0.0f ? 0.0f : powf(in, 2.5f);
}
static float tf_linear_to_rec709(float in, int fl) {                  /* gammatab.c:58-66 */
    if (in < 0.018f) return in * 4.5f;
    return fl ? fmaf(1.099f, powf(in, 0.45f), -0.099f) : 1.099f * powf(in, 0.45f) - 0.099f;
}
static float tf_linear_to_srgb(float in, int fl) {                    /* gammatab.c:201-211 */
    const float a = 0.055;
    if (in <= 0.0031308f) return in * 12.92f;
    return fl ? fmaf(1.0f + a, powf(in, 1.0f / 2.4f), -a) : (1.0f + a) * powf(in, 1.0f / 2.4f) - a;
}
/* the slot of a table: [flavour][which], the flavour-independent tables always in flavour 0's row */
static inline int lut_row(int which) { return (which == CVS_LUT_LINEAR_TO_REC709 || which == CVS_LUT_LINEAR_TO_SRGB) && cvs_arith() == CVS_ARITH_CONTRACTED; }

/* The tables themselves are host values (one copy per flavour row, built once); every device context gets its own copy in HBM
 * the first time it asks -- lut_dev_gen records which generation of the host table that copy holds. */
static pthread_mutex_t lut_lock = PTHREAD_MUTEX_INITIALIZER;
static half *lut_host2[2][CVS_LUT_COUNT];
static unsigned lut_gen2[2][CVS_LUT_COUNT];                            /* bumped whenever the host table changes (0: not built yet) */
static half *lut_dev3[CVS_MAX_CONTEXTS][2][CVS_LUT_COUNT];
static unsigned lut_dev_gen[CVS_MAX_CONTEXTS][2][CVS_LUT_COUNT];
static uint8_t *ramp45;
static float *codes_as_float;       /* h2f of 0..65535, computed once on the GPU */

static int ensure_codes(void) {
    if (codes_as_float) return 0;
    half *codes = malloc(HALF_COUNT * sizeof(half));
    float *f = malloc(HALF_COUNT * sizeof(float));
    if (!codes || !f) { free(codes); free(f); return -1; }
    for (int i = 0; i < HALF_COUNT; i++) codes[i] = (half)i;
    int rc = host_convert(f, HALF_COUNT * 4, codes, HALF_COUNT * 2, HALF_COUNT, 1, 0);
    free(codes);
    if (rc != 0) { free(f); return rc; }
    codes_as_float = f;
    return 0;
}

/* lut_lock held: a new host table (every context's device copy is now stale) */
static int install_locked(int row, int which, const half *table) {
    half **lut_host = lut_host2[row];
    if (!lut_host[which]) lut_host[which] = malloc(HALF_COUNT * sizeof(half));
    if (!lut_host[which]) return -1;
    if (table != lut_host[which]) memcpy(lut_host[which], table, HALF_COUNT * sizeof(half));
    lut_gen2[row][which]++;
    return 0;
}

/* lut_lock held, cvs_enter() done: the calling thread's context holds the current host table */
static int upload_locked(int row, int which) {
    const int c = cvs_ctx();
    if (lut_dev3[c][row][which] && lut_dev_gen[c][row][which] == lut_gen2[row][which]) return 0;
    if (!lut_dev3[c][row][which]) CVS_HIP(hipMalloc((void **)&lut_dev3[c][row][which], HALF_COUNT * sizeof(half)));
    CVS_HIP(hipMemcpy(lut_dev3[c][row][which], lut_host2[row][which], HALF_COUNT * sizeof(half), hipMemcpyHostToDevice));
    lut_dev_gen[c][row][which] = lut_gen2[row][which];
    return 0;
}

/* h2f of every half code (computed once, on the GPU): what ramps over the whole code space are built from */
const float *cvs_codes_as_float(void) {
    if (cvs_enter() != 0) return NULL;
    pthread_mutex_lock(&lut_lock);
    int rc = ensure_codes();
    pthread_mutex_unlock(&lut_lock);
    return rc == 0 ? codes_as_float : NULL;
}

/* (counts installs of either flavour's table, so that a composed table made from one is rebuilt when the flavour's table changes) */
unsigned cvs_lut_generation(int which) {
    return (which >= 0 && which < CVS_LUT_COUNT) ? __atomic_load_n(&lut_gen2[lut_row(which)][which], __ATOMIC_ACQUIRE) * 2u + (unsigned)lut_row(which) : 0;
}

static int ensure_lut(int which) {
    if (which < 0 || which >= CVS_LUT_COUNT) { cvs_set_error("no such transfer table: %d", which); return -1; }
    if (cvs_enter() != 0) return -1;
    const int row = lut_row(which);
    pthread_mutex_lock(&lut_lock);
    int rc = 0;
    if (!lut_gen2[row][which]) {
        float (*fn[CVS_LUT_COUNT])(float, int) = { tf_rec709_to_linear, tf_rec709_display, tf_linear_to_rec709, tf_linear_to_srgb };
        rc = ensure_codes();
        if (rc == 0) {
            float *g = malloc(HALF_COUNT * sizeof(float));
            half *t = malloc(HALF_COUNT * sizeof(half));
            if (!g || !t) rc = -1;
            if (rc == 0) {
                for (int i = 0; i < HALF_COUNT; i++) g[i] = fn[which](codes_as_float[i], row);
                rc = host_convert(t, HALF_COUNT * 2, g, HALF_COUNT * 4, HALF_COUNT, 0, 0);
            }
            if (rc == 0) rc = install_locked(row, which, t);
            free(g); free(t);
        }
    }
    if (rc == 0) rc = upload_locked(row, which);
    pthread_mutex_unlock(&lut_lock);
    return rc;
}

CVS_EXPORT const half *cvs_lut_device(int which) { return ensure_lut(which) == 0 ? lut_dev3[cvs_ctx()][lut_row(which)][which] : NULL; }
CVS_EXPORT const half *cvs_lut_host(int which) { return ensure_lut(which) == 0 ? lut_host2[lut_row(which)][which] : NULL; }

const half *cvs_lut_dev_or_null(int which) { return which == CVS_LUT_NONE ? NULL : cvs_lut_device(which); }

CVS_EXPORT int cvs_lut_install(int which, const half *table_host) {
    if (which < 0 || which >= CVS_LUT_COUNT || !table_host) { cvs_set_error("cvs_lut_install: bad arguments"); return -1; }
    if (cvs_enter() != 0) return -1;
    /* a caller's table replaces the built-in one of BOTH flavours */
    pthread_mutex_lock(&lut_lock);
    int rc = install_locked(0, which, table_host);
    if (rc == 0 && (which == CVS_LUT_LINEAR_TO_REC709 || which == CVS_LUT_LINEAR_TO_SRGB)) rc = install_locked(1, which, table_host);
    pthread_mutex_unlock(&lut_lock);
    return rc;
}

static void host_transfer(int which, half *out, const half *in, size_t count) {
    if (!count || ensure_lut(which) != 0) return;
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged din, dout;
    din.dev = dout.dev = NULL;
    int rc = cvs_stage_in(&din, in, count * 2, 1, s);
    if (rc == 0) rc = cvs_stage_in(&dout, NULL, count * 2, 0, s);
    if (rc == 0) {
        rc = cvk_half_lookup(lut_dev3[cvs_ctx()][lut_row(which)][which], (uint16_t *)dout.dev, (const uint16_t *)din.dev, count, cvs_cus(), s);
        if (rc != 0) cvs_set_error("transfer kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
    if (rc == 0) cvs_stage_out(&dout, out, s);
    cvs_stage_free(&din); cvs_stage_free(&dout);
}

CVS_EXPORT void video_transfer_rec709_to_linear_scene(half *out, const half *in, size_t count) { host_transfer(CVS_LUT_REC709_TO_LINEAR_SCENE, out, in, count); }
CVS_EXPORT void video_transfer_rec709_to_linear_display(half *out, const half *in, size_t count) { host_transfer(CVS_LUT_REC709_TO_LINEAR_DISPLAY, out, in, count); }
CVS_EXPORT void video_transfer_linear_to_rec709(half *out, const half *in, size_t count) { host_transfer(CVS_LUT_LINEAR_TO_REC709, out, in, count); }
CVS_EXPORT void video_transfer_linear_to_sRGB(half *out, const half *in, size_t count) { host_transfer(CVS_LUT_LINEAR_TO_SRGB, out, in, count); }

CVS_EXPORT const uint8_t *video_get_gamma45_ramp(void) {              /* gammatab.c:13-38 */
    if (cvs_enter() != 0) return NULL;
    pthread_mutex_lock(&lut_lock);
    if (!ramp45 && ensure_codes() == 0) {
        uint8_t *r = malloc(HALF_COUNT);
        if (r) {
            for (int i = 0; i < HALF_COUNT; i++)
                r[i] = (uint8_t)clampf(powf(codes_as_float[i], 0.45f) * 255.0f, 0.0f, 255.0f);
            ramp45 = r;
        }
    }
    pthread_mutex_unlock(&lut_lock);
    return ramp45;
}
