/*
 * runtime.c -- device binding, streams, memory, error reporting, host<->HBM staging.
 *
 * The reference has no device layer (its frames are malloc'd host buffers).  What it does have is
 * a threading contract this file has to keep: get_frame may be entered from any thread, with or
 * without the GIL (SURVEY.md section 8b "threading").  HIP's current device is per thread, so every
 * entry point goes through cvs_enter(), and each thread that does not bring its own stream gets one
 * of its own -- no shared stream, no global lock on the pixel path.
 */
#define _GNU_SOURCE
#include "internal.h"
#include <pthread.h>
#include <stdarg.h>
#include <time.h>

/* ---- device contexts.  A context = one HIP device + everything the library keeps ON that device for its own use: the
 * scratch pool (below), the transfer tables (halfconv.c), the FIR tap tables (scale.c), the byte tables (display.c), and one
 * stream per calling thread.  A process starts with none; cvs_init(device) -- or the first entry point, with CVS_DEVICE --
 * opens context 0 and makes it the DEFAULT: what every thread runs in that never chose another.  cvs_context_open(device)
 * opens further ones (another GPU of the node -- or the same one again: two contexts on one device share nothing but the
 * device), cvs_set_context() binds the calling thread.  One pull-queue worker per context is how an editor process uses the
 * eight GPUs of a node (the reference's only frame-parallel consumer is its thread pool, src/process/VideoPullQueue.c:99-113):
 * frames are independent, tables are small and rebuilt per context on first use, nothing is exchanged between devices.
 * Device pointers belong to the context (device) they were allocated in; handing them to a call made in a context on another
 * device is the caller's error. */
#define POOL_SLOTS 256
typedef struct { void *ptr; size_t bytes; hipStream_t stream; hipEvent_t ev; int live; uint64_t stamp; } pool_slot;
typedef struct {
    int device;                        /* HIP device ordinal */
    int cus;
    char name[640];
    pool_slot pool[POOL_SLOTS];        /* the scratch pool of this context (see below) */
    uint64_t pool_clock;
    size_t pool_parked;                /* bytes sitting idle in the pool */
} cvs_context;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static cvs_context g_ctx[CVS_MAX_CONTEXTS];
static int g_nctx;                     /* contexts opened so far (ids 0 .. g_nctx - 1; never closed) */
static int g_default_ctx = -1;         /* the context of threads that never called cvs_set_context */
static __thread int t_ctx_choice = -1; /* this thread's cvs_set_context, -1: the default */
static __thread int t_ctx = 0;         /* the context of the call this thread is in (snapshot by cvs_enter) */
int cvs_ctx(void) { return t_ctx; }

static int g_arith = -1;              /* CVS_ARITH_*; -1: not set yet (the environment decides at the first entry) */
static __thread int t_arith;          /* the flavour of the call this thread is in */
static __thread char t_error[512];
static __thread hipStream_t t_streams[CVS_MAX_CONTEXTS];      /* this thread's own stream in each context it has called into */

/* where diagnostics go besides cvs_last_error(): stderr unless the host installs a handler (the reference routes its
 * g_log domains to Python's logging the same way, src/process/main.c:272-329) */
static cvs_log_func g_log_handler;
static void *g_log_user;

CVS_EXPORT void cvs_set_log_handler(cvs_log_func handler, void *user_data) {
    __atomic_store_n(&g_log_user, user_data, __ATOMIC_RELEASE);
    __atomic_store_n(&g_log_handler, handler, __ATOMIC_RELEASE);
}

void cvs_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_error, sizeof t_error, fmt, ap);
    va_end(ap);
    cvs_log_func handler = __atomic_load_n(&g_log_handler, __ATOMIC_ACQUIRE);
    if (handler) handler("fluggo.media.cprocess", CVS_LOG_WARNING, t_error, __atomic_load_n(&g_log_user, __ATOMIC_ACQUIRE));
    else fprintf(stderr, "canvas_hip: %s\n", t_error);
}

void cvs_log_warning(const char *fmt, ...) {
    char text[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(text, sizeof text, fmt, ap);
    va_end(ap);
    cvs_log_func handler = __atomic_load_n(&g_log_handler, __ATOMIC_ACQUIRE);
    if (handler) handler("fluggo.media.cprocess", CVS_LOG_WARNING, text, __atomic_load_n(&g_log_user, __ATOMIC_ACQUIRE));
    else fprintf(stderr, "canvas_hip: %s\n", text);
}

void cvs_clear_error(void) { t_error[0] = 0; }

CVS_EXPORT const char *cvs_last_error(void) { return t_error; }
CVS_EXPORT void cvs_clear_last_error(void) { t_error[0] = 0; }

CVS_EXPORT int cvs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int arith_now(void) {
    int a = __atomic_load_n(&g_arith, __ATOMIC_RELAXED);
    if (a < 0) {
        const char *env = getenv("CVS_ARITHMETIC");
        a = env && (strcmp(env, "contracted") == 0 || strcmp(env, "fma") == 0 || strcmp(env, "1") == 0) ? CVS_ARITH_CONTRACTED : CVS_ARITH_SEPARATE;
        int unset = -1;
        if (!__atomic_compare_exchange_n(&g_arith, &unset, a, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) a = unset;      /* someone set it meanwhile */
    }
    return a;
}

CVS_EXPORT int cvs_set_arithmetic(int mode) {
    if (mode != CVS_ARITH_SEPARATE && mode != CVS_ARITH_CONTRACTED) { cvs_set_error("cvs_set_arithmetic: unknown mode %d", mode); return -1; }
    const int before = arith_now();
    __atomic_store_n(&g_arith, mode, __ATOMIC_RELAXED);
    t_arith = mode;
    return before;
}
CVS_EXPORT int cvs_get_arithmetic(void) { return arith_now(); }
int cvs_arith(void) { return t_arith; }

/* g_lock held.  A new context on `device`; its id, or -1 (message set). */
static int open_context_locked(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        cvs_set_error("no HIP device available (%s); this library has no CPU path", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
        return -1;
    }
    if (device < 0 || device >= n) { cvs_set_error("device %d out of range (0..%d)", device, n - 1); return -1; }
    if (g_nctx >= CVS_MAX_CONTEXTS) { cvs_set_error("no more than %d device contexts", CVS_MAX_CONTEXTS); return -1; }
    if ((e = hipSetDevice(device)) != hipSuccess) { cvs_set_error("hipSetDevice(%d): %s", device, hipGetErrorString(e)); return -1; }
    cvs_context *c = &g_ctx[g_nctx];
    memset(c, 0, sizeof *c);
    c->device = device;
    c->cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        c->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        /* some boxes report an empty marketing name */
        snprintf(c->name, sizeof c->name, "%s (%s, %d CUs)", prop.name[0] ? prop.name : "AMD GPU", prop.gcnArchName, prop.multiProcessorCount);
    }
    return g_nctx++;
}

/* Binds the library (the default context) and the calling thread to HIP device `device`: the first context on that device, or
 * a new one.  Idempotent per device. */
CVS_EXPORT int cvs_init(int device) {
    pthread_mutex_lock(&g_lock);
    int id = -1;
    for (int i = 0; i < g_nctx && id < 0; i++) if (g_ctx[i].device == device) id = i;
    if (id < 0) id = open_context_locked(device);
    if (id >= 0) g_default_ctx = id;
    pthread_mutex_unlock(&g_lock);
    if (id < 0) return -1;
    t_ctx_choice = -1;
    t_ctx = id;
    return hipSetDevice(device) == hipSuccess ? 0 : -1;
}

CVS_EXPORT int cvs_context_open(int device) {
    pthread_mutex_lock(&g_lock);
    const int id = open_context_locked(device);
    if (id >= 0 && g_default_ctx < 0) g_default_ctx = id;
    pthread_mutex_unlock(&g_lock);
    return id;
}

CVS_EXPORT int cvs_context_count(void) { return __atomic_load_n(&g_nctx, __ATOMIC_ACQUIRE); }

CVS_EXPORT int cvs_context_device(int ctx) {
    if (ctx < 0 || ctx >= cvs_context_count()) return -1;
    return g_ctx[ctx].device;
}

CVS_EXPORT int cvs_current_context(void) { return t_ctx_choice >= 0 ? t_ctx_choice : __atomic_load_n(&g_default_ctx, __ATOMIC_ACQUIRE); }

CVS_EXPORT int cvs_set_context(int ctx) {
    const int before = cvs_current_context();
    if (ctx >= cvs_context_count() || ctx < -1) { cvs_set_error("cvs_set_context: no context %d (%d open)", ctx, cvs_context_count()); return -2; }
    t_ctx_choice = ctx;                                   /* -1: back to the default context */
    return before;
}

/* round-robin owner of a frame among `nowners` devices / contexts / ranks: the sharding rule of the whole library
 * (frames are independent random-access units; canvas_amd/shard.py uses the same rule across processes) */
CVS_EXPORT int cvs_frame_owner(int64_t frame_index, int nowners) {
    if (nowners <= 0) return -1;
    const int64_t m = frame_index % nowners;
    return (int)(m < 0 ? m + nowners : m);
}

int cvs_enter(void) {
    t_arith = arith_now();
    int id = t_ctx_choice;
    if (id < 0) {
        id = __atomic_load_n(&g_default_ctx, __ATOMIC_ACQUIRE);
        if (id < 0) {
            const char *env = getenv("CVS_DEVICE");
            if (cvs_init(env ? atoi(env) : 0) != 0) return -1;
            id = __atomic_load_n(&g_default_ctx, __ATOMIC_ACQUIRE);
        }
    }
    t_ctx = id;
    if (hipSetDevice(g_ctx[id].device) != hipSuccess) { cvs_set_error("hipSetDevice(%d) failed", g_ctx[id].device); return -1; }
    return 0;
}

CVS_EXPORT int cvs_current_device(void) { const int c = cvs_current_context(); return c >= 0 ? g_ctx[c].device : -1; }
CVS_EXPORT const char *cvs_device_name(void) { return cvs_enter() == 0 ? g_ctx[t_ctx].name : ""; }
CVS_EXPORT int cvs_compute_units(void) { return cvs_enter() == 0 ? g_ctx[t_ctx].cus : 0; }
int cvs_cus(void) { return g_ctx[t_ctx].cus; }

static __thread char t_have_stream[CVS_MAX_CONTEXTS];
hipStream_t cvs_pick_stream(cvs_stream_t s) {
    if (s) return (hipStream_t)s;
    if (!t_have_stream[t_ctx]) {                           /* (the thread is bound to the context's device: cvs_enter) */
        if (hipStreamCreateWithFlags(&t_streams[t_ctx], hipStreamNonBlocking) != hipSuccess) t_streams[t_ctx] = NULL;
        t_have_stream[t_ctx] = 1;
    }
    return t_streams[t_ctx];
}

CVS_EXPORT void *cvs_malloc(size_t bytes) {
    void *p = NULL;
    if (cvs_enter() != 0) return NULL;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { cvs_set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return NULL; }
    return p;
}

/* A block from cvs_malloc: the library does not know which streams its user ran on it, so wait for the whole device
 * (hipFree alone does not wait for kernels on non-blocking streams: a queued kernel faulted on a just-freed block).
 * Pool blocks know their stream and wait for that alone (release_to_driver below). */
static void free_when_idle(void *dev) {
    (void)hipDeviceSynchronize();
    (void)hipFree(dev);
}

CVS_EXPORT void cvs_free(void *dev) {
    if (dev && cvs_enter() == 0) free_when_idle(dev);
}

CVS_EXPORT int cvs_memcpy_h2d(void *dev, const void *host, size_t bytes, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, cvs_pick_stream(s)));
    if (!s) CVS_HIP(hipStreamSynchronize(cvs_pick_stream(s)));
    return 0;
}

CVS_EXPORT int cvs_memcpy_d2h(void *host, const void *dev, size_t bytes, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, cvs_pick_stream(s)));
    CVS_HIP(hipStreamSynchronize(cvs_pick_stream(s)));
    return 0;
}

CVS_EXPORT int cvs_memcpy_d2d(void *dst, const void *src, size_t bytes, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, cvs_pick_stream(s)));
    return 0;
}

CVS_EXPORT int cvs_memset(void *dev, int value, size_t bytes, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_HIP(hipMemsetAsync(dev, value, bytes, cvs_pick_stream(s)));
    return 0;
}

CVS_EXPORT cvs_stream_t cvs_stream_create(void) {
    hipStream_t s = NULL;
    if (cvs_enter() != 0) return NULL;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { cvs_set_error("hipStreamCreate failed"); return NULL; }
    return s;
}

CVS_EXPORT void cvs_stream_destroy(cvs_stream_t s) {
    if (s && cvs_enter() == 0) hipStreamDestroy((hipStream_t)s);
}

CVS_EXPORT int cvs_stream_sync(cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_HIP(hipStreamSynchronize(cvs_pick_stream(s)));
    return 0;
}

/* ---- scratch pool: intermediates of a filter graph (f32 temps, pulled layers) come and go once per frame; hipMalloc
 * costs tens of microseconds and hipFree stalls, so freed blocks are parked and handed back to later requests.
 *
 * Lifetime rule (what the first version got wrong, and then hid behind a device-wide wait before every hipFree): a block
 * is "freed" while the kernels that use it are still queued on the freeing stream.  Every parked block therefore carries
 * a hipEvent recorded on that stream at the moment of the free:
 *   - handing the block to the SAME stream needs nothing (stream order);
 *   - handing it to ANOTHER stream makes that stream wait for the event on the device (hipStreamWaitEvent: no host stall,
 *     and no stale stream handle -- the first version called hipStreamSynchronize on a remembered handle that its owner
 *     may have destroyed meanwhile, the error was ignored and the block went out while still in use);
 *   - returning it to the driver (eviction, trim) waits for the event on the host first: hipFree of memory that a kernel
 *     on a non-blocking stream still reads is what faulted.
 * A request is served by the smallest parked block of at least its size and at most 1/8 more (animated windows ask for a
 * slightly different size every frame); when the table is full the least recently parked block is evicted. */
#define GRAPH_BLOCKS 64
typedef struct { void (*release)(void *); void *arg; } cvs_hold;
static __thread struct { int active, overflow; hipStream_t stream; void *blocks[GRAPH_BLOCKS]; int n; cvs_hold holds[GRAPH_BLOCKS]; int nholds; } t_capture;
/* the pool of the context the calling thread's call runs in (g_lock covers every context's table) */
#define g_pool (g_ctx[t_ctx].pool)
#define g_pool_clock (g_ctx[t_ctx].pool_clock)
#define g_pool_parked (g_ctx[t_ctx].pool_parked)
static const size_t kPoolParkedMax = (size_t)8 << 30;

static void release_to_driver(void *ptr, hipEvent_t ev) {      /* g_lock NOT held */
    if (ev) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); }
    (void)hipFree(ptr);
}

CVS_EXPORT void *cvs_pool_malloc(size_t bytes, cvs_stream_t s) {
    if (cvs_enter() != 0) return NULL;
    if (!bytes) bytes = 1;
    hipStream_t st = cvs_pick_stream(s);
    pthread_mutex_lock(&g_lock);
    int best = -1;
    for (int i = 0; i < POOL_SLOTS; i++)
        if (g_pool[i].ptr && !g_pool[i].live && g_pool[i].bytes >= bytes && g_pool[i].bytes - bytes <= bytes / 8 &&
            (best < 0 || g_pool[i].bytes < g_pool[best].bytes)) best = i;
    if (best >= 0) {
        pool_slot *b = &g_pool[best];
        b->live = 1;
        g_pool_parked -= b->bytes;
        const hipStream_t prev = b->stream;
        const hipEvent_t ev = b->ev;
        void *p = b->ptr;
        b->stream = st;
        pthread_mutex_unlock(&g_lock);
        if (prev != st && ev) {                      /* the old user's kernels finish before ours start */
            if (t_capture.active && st == t_capture.stream) (void)hipEventSynchronize(ev);      /* no outside events inside a capture */
            else (void)hipStreamWaitEvent(st, ev, 0);
        }
        return p;
    }
    pthread_mutex_unlock(&g_lock);
    void *p = NULL;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        cvs_pool_trim();                                                  /* parked blocks may be what is in the way */
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) { cvs_set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return NULL; }
    void *evict_ptr = NULL; hipEvent_t evict_ev = NULL;
    pthread_mutex_lock(&g_lock);
    int slot = -1, oldest = -1;
    for (int i = 0; i < POOL_SLOTS; i++) {
        if (!g_pool[i].ptr) { slot = i; break; }
        if (!g_pool[i].live && (oldest < 0 || g_pool[i].stamp < g_pool[oldest].stamp)) oldest = i;
    }
    if (slot < 0 && oldest >= 0) {                  /* table full: the least recently parked block goes back to the driver */
        slot = oldest;
        evict_ptr = g_pool[slot].ptr; evict_ev = g_pool[slot].ev;
        g_pool_parked -= g_pool[slot].bytes;
    }
    if (slot >= 0) { pool_slot fresh = { p, bytes, st, NULL, 1, 0 }; g_pool[slot] = fresh; }
    pthread_mutex_unlock(&g_lock);                  /* every slot live: the block is untracked, cvs_pool_free handles that */
    if (evict_ptr) release_to_driver(evict_ptr, evict_ev);
    return p;
}

/* ---- graph capture state of the calling thread (see cvs_graph_begin below) */
/* (t_capture itself is declared above the pool, which needs to know whether its stream is capturing) */

/* A cached device table (FIR taps, display bytes) that a launch has just been handed: outside a capture the caller
 * lets go of it as soon as the launch is enqueued; inside one, the recorded kernels will read it at every replay, so
 * the hold passes to the graph and `release(arg)` runs when the graph is destroyed.  Returns 1 when the graph took it. */
int cvs_capture_hold(hipStream_t st, void (*release)(void *), void *arg) {
    if (!t_capture.active || st != t_capture.stream) return 0;
    if (t_capture.nholds < GRAPH_BLOCKS) { t_capture.holds[t_capture.nholds].release = release; t_capture.holds[t_capture.nholds].arg = arg; t_capture.nholds++; return 1; }
    t_capture.overflow = 1;
    return 0;
}

CVS_EXPORT void cvs_pool_free(void *dev, cvs_stream_t s) {
    if (!dev || cvs_enter() != 0) return;
    hipStream_t st = cvs_pick_stream(s);
    if (t_capture.active && st == t_capture.stream) {
        /* the captured kernels will use this block at every replay: it stays out of the pool, owned by the graph */
        if (t_capture.n < GRAPH_BLOCKS) t_capture.blocks[t_capture.n++] = dev;
        else t_capture.overflow = 1;
        return;
    }
    hipEvent_t ev = NULL;
    pthread_mutex_lock(&g_lock);
    int slot = -1;
    for (int i = 0; i < POOL_SLOTS; i++)
        if (g_pool[i].ptr == dev) { slot = i; break; }
    if (slot >= 0) { ev = g_pool[slot].ev; g_pool[slot].ev = NULL; }
    pthread_mutex_unlock(&g_lock);
    /* the kernels that use the block are queued on `st`: mark the point after them */
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = NULL;
    if (ev && hipEventRecord(ev, st) != hipSuccess) { (void)hipEventDestroy(ev); ev = NULL; }
    if (!ev) (void)hipStreamSynchronize(st);                            /* no event to be had: wait here instead */
    bool parked = false;
    pthread_mutex_lock(&g_lock);
    if (slot >= 0 && g_pool[slot].ptr == dev && g_pool[slot].live) {
        if (g_pool_parked + g_pool[slot].bytes <= kPoolParkedMax) {
            g_pool[slot].live = 0;
            g_pool[slot].stream = st;
            g_pool[slot].ev = ev;
            g_pool[slot].stamp = ++g_pool_clock;
            g_pool_parked += g_pool[slot].bytes;
            parked = true;
        } else {
            g_pool[slot].ptr = NULL;
        }
    }
    pthread_mutex_unlock(&g_lock);
    if (!parked) release_to_driver(dev, ev);        /* over the parking limit, or a block the full table never tracked */
}

CVS_EXPORT int cvs_mem_info(size_t *free_bytes, size_t *total_bytes) {
    if (cvs_enter() != 0) return -1;
    size_t f = 0, t = 0;
    CVS_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

CVS_EXPORT void cvs_pool_trim(void) {
    if (cvs_enter() != 0) return;
    for (;;) {                                      /* one block at a time: the wait and the hipFree happen outside the lock */
        void *ptr = NULL; hipEvent_t ev = NULL;
        pthread_mutex_lock(&g_lock);
        for (int i = 0; i < POOL_SLOTS; i++)
            if (g_pool[i].ptr && !g_pool[i].live) {
                ptr = g_pool[i].ptr; ev = g_pool[i].ev;
                g_pool_parked -= g_pool[i].bytes;
                g_pool[i].ptr = NULL; g_pool[i].ev = NULL;
                break;
            }
        pthread_mutex_unlock(&g_lock);
        if (!ptr) break;
        release_to_driver(ptr, ev);
    }
}

/* ---- HIP graphs: a launch-bound sequence (a node graph on small frames is a dozen 10-20 us kernels) recorded once
 * and replayed with one submission.  Everything between begin and end must be device-frame entry points on the
 * capturing stream, called from the capturing thread, and must have run once before (so that tables, occupancy
 * figures and pool blocks exist: nothing may allocate or synchronise while a stream is capturing).  Scratch blocks
 * the sequence takes from the pool, and the cached tables its kernels read, belong to the graph until it is destroyed.  Frame pointers and parameters are
 * baked in; the CONTENTS of the frames are whatever they hold at replay time. */
typedef struct { hipGraph_t graph; hipGraphExec_t exec; void *blocks[GRAPH_BLOCKS]; int n; cvs_hold holds[GRAPH_BLOCKS]; int nholds; hipStream_t stream; } cvs_graph;

CVS_EXPORT int cvs_graph_begin(cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    if (t_capture.active) { cvs_set_error("graph capture: already capturing on this thread"); return -1; }
    hipStream_t st = cvs_pick_stream(s);
    CVS_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    memset(&t_capture, 0, sizeof t_capture);
    t_capture.active = 1;
    t_capture.stream = st;
    return 0;
}

CVS_EXPORT cvs_graph_t cvs_graph_end(cvs_stream_t s) {
    if (cvs_enter() != 0 || !t_capture.active) { cvs_set_error("graph capture: not capturing"); return NULL; }
    hipStream_t st = t_capture.stream;
    t_capture.active = 0;
    hipGraph_t graph = NULL;
    hipError_t e = hipStreamEndCapture(st, &graph);
    cvs_graph *g = (e == hipSuccess && graph && !t_capture.overflow) ? calloc(1, sizeof *g) : NULL;
    if (g) {
        g->graph = graph;
        e = hipGraphInstantiate(&g->exec, graph, NULL, NULL, 0);
        if (e != hipSuccess) { free(g); g = NULL; }
    }
    if (!g) {
        cvs_set_error("graph capture failed: %s", t_capture.overflow ? "too many scratch blocks" : hipGetErrorString(e));
        if (graph) hipGraphDestroy(graph);
        for (int i = 0; i < t_capture.n; i++) cvs_pool_free(t_capture.blocks[i], st);     /* not capturing any more: back to the pool */
        for (int i = 0; i < t_capture.nholds; i++) t_capture.holds[i].release(t_capture.holds[i].arg);
        return NULL;
    }
    memcpy(g->blocks, t_capture.blocks, sizeof(void *) * (size_t)t_capture.n);
    g->n = t_capture.n;
    memcpy(g->holds, t_capture.holds, sizeof(cvs_hold) * (size_t)t_capture.nholds);
    g->nholds = t_capture.nholds;
    g->stream = st;
    return g;
}

CVS_EXPORT int cvs_graph_launch(cvs_graph_t graph, cvs_stream_t s) {
    cvs_graph *g = graph;
    if (cvs_enter() != 0 || !g) return -1;
    CVS_HIP(hipGraphLaunch(g->exec, cvs_pick_stream(s)));
    return 0;
}

CVS_EXPORT void cvs_graph_destroy(cvs_graph_t graph) {
    cvs_graph *g = graph;
    if (!g || cvs_enter() != 0) return;
    (void)hipDeviceSynchronize();                 /* a replay may still be running */
    hipGraphExecDestroy(g->exec);
    hipGraphDestroy(g->graph);
    for (int i = 0; i < g->n; i++) cvs_pool_free(g->blocks[i], g->stream);
    for (int i = 0; i < g->nholds; i++) g->holds[i].release(g->holds[i].arg);
    free(g);
}

CVS_EXPORT cvs_event_t cvs_event_create(void) {
    hipEvent_t e = NULL;
    if (cvs_enter() != 0) return NULL;
    if (hipEventCreate(&e) != hipSuccess) { cvs_set_error("hipEventCreate failed"); return NULL; }
    return e;
}

CVS_EXPORT void cvs_event_destroy(cvs_event_t e) {
    if (e && cvs_enter() == 0) hipEventDestroy((hipEvent_t)e);
}

CVS_EXPORT int cvs_event_record(cvs_event_t e, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_HIP(hipEventRecord((hipEvent_t)e, cvs_pick_stream(s)));
    return 0;
}

CVS_EXPORT int cvs_event_sync(cvs_event_t e) {
    if (cvs_enter() != 0) return -1;
    CVS_HIP(hipEventSynchronize((hipEvent_t)e));
    return 0;
}

CVS_EXPORT float cvs_event_elapsed_ms(cvs_event_t start, cvs_event_t stop) {
    float ms = -1.0f;
    if (cvs_enter() != 0) return -1.0f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) { cvs_set_error("hipEventElapsedTime failed"); return -1.0f; }
    return ms;
}

/* ---- staging */

int cvs_stage_in(cvs_staged *st, const void *host, size_t bytes, int upload, hipStream_t s) {
    st->dev = NULL;
    st->bytes = bytes;
    st->stream = s;
    if (!bytes) return 0;
    st->dev = cvs_pool_malloc(bytes, s);
    if (!st->dev) return -1;
    if (upload) CVS_HIP(hipMemcpyAsync(st->dev, host, bytes, hipMemcpyHostToDevice, s));
    return 0;
}

int cvs_stage_out(cvs_staged *st, void *host, hipStream_t s) {
    if (st->dev && st->bytes) CVS_HIP(hipMemcpyAsync(host, st->dev, st->bytes, hipMemcpyDeviceToHost, s));
    CVS_HIP(hipStreamSynchronize(s));
    return 0;
}

void cvs_stage_free(cvs_staged *st) {
    if (st->dev) cvs_pool_free(st->dev, st->stream);      /* stream-ordered: whatever is still queued on it finishes first */
    st->dev = NULL;
}

/* ---- timing helpers of the reference's C-ABI */

CVS_EXPORT int64_t gettime(void) {                                   /* src/cprocess/clock.c:28-52 */
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * NS_PER_SEC + (int64_t)ts.tv_nsec;
}

CVS_EXPORT int64_t get_frame_time(const rational *rate, int frame) { /* src/cprocess/main.c:23-26 */
    return ((int64_t)frame * NS_PER_SEC * (int64_t)rate->d) / (int64_t)rate->n + INT64_C(1);
}

CVS_EXPORT int get_time_frame(const rational *rate, int64_t time) {  /* src/cprocess/main.c:28-31 */
    return (int)((time * (int64_t)rate->n) / (NS_PER_SEC * (int64_t)rate->d));
}
