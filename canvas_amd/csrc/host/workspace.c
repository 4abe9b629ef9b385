/*
 * workspace.c -- the timeline compositor (video half of src/cprocess/workspace.c).
 *
 * Item API: workspace.c:107-492.  Frame: workspace_get_frame_f32, workspace.c:494-550 -- the items
 * alive at frame_index (x <= index < x+length, :257,272-275), lowest z first (cmpz, :102-105, list
 * walked from its end, :510-520); the bottom item is pulled straight into the output, every other
 * item into a full-size f32 temp and mixed on with video_mix_over_f32(.., 1.0f); an item's frame is
 * index - x + offset.
 *
 * The reference keeps three GSequences and two cursors so that stepping to a neighbouring frame is
 * cheap on one thread.  Frames here are spread over GPUs and arrive in any order, so membership is
 * recomputed per call from one x-sorted array: O(n) for the handful of clips alive in an editor
 * timeline, no cursor state to keep coherent between threads.  Items with EQUAL z stack in the
 * order they were added (the reference's order for ties depends on cursor history; unpinned).
 *
 * Two vtable entries: get_frame_32 on host frames (pulls host frames, every over staged through
 * HBM) and, in slot 3, get_frame_dev, which keeps the whole stack in HBM.
 */
#define _GNU_SOURCE
#include "internal.h"
#include <pthread.h>

struct workspace_item_t_tag {
    workspace_t *workspace;
    int64_t x, z, length, offset;
    void *source, *tag;
    uint64_t serial;          /* add order: tie-break for equal z */
};

struct workspace_t_tag {
    pthread_mutex_t mutex;    /* protects items[] and the fields of each item (workspace.c:59-60) */
    workspace_item_t **items; /* sorted by (x, z) like leftsort (workspace.c:75-84) */
    int count, cap;
    uint64_t next_serial;
};

static int cmp64(int64_t a, int64_t b) { return (a > b) - (a < b); }

static int left_order(const workspace_item_t *a, const workspace_item_t *b) {
    int r = cmp64(a->x, b->x);
    return r ? r : cmp64(a->z, b->z);
}

static void insert_sorted(workspace_t *w, workspace_item_t *item) {
    if (w->count == w->cap) {
        w->cap = w->cap ? w->cap * 2 : 16;
        w->items = realloc(w->items, sizeof(*w->items) * (size_t)w->cap);
    }
    int i = w->count;
    while (i > 0 && left_order(w->items[i - 1], item) > 0) { w->items[i] = w->items[i - 1]; i--; }   /* after equals */
    w->items[i] = item;
    w->count++;
}

static void take_out(workspace_t *w, workspace_item_t *item) {
    for (int i = 0; i < w->count; i++)
        if (w->items[i] == item) {
            memmove(&w->items[i], &w->items[i + 1], sizeof(*w->items) * (size_t)(w->count - i - 1));
            w->count--;
            return;
        }
}

CVS_EXPORT workspace_t *workspace_create(void) {
    workspace_t *w = calloc(1, sizeof *w);
    if (w) pthread_mutex_init(&w->mutex, NULL);
    return w;
}

CVS_EXPORT int workspace_get_length(workspace_t *w) { return w->count; }

CVS_EXPORT workspace_item_t *workspace_add_item(workspace_t *w, void *source, int64_t x, int64_t length, int64_t offset, int64_t z, void *tag) {
    workspace_item_t *item = calloc(1, sizeof *item);
    if (!item) return NULL;
    item->workspace = w;
    item->x = x; item->z = z; item->length = length; item->offset = offset;
    item->source = source; item->tag = tag;
    pthread_mutex_lock(&w->mutex);
    item->serial = w->next_serial++;
    insert_sorted(w, item);
    pthread_mutex_unlock(&w->mutex);
    return item;
}

CVS_EXPORT workspace_item_t *workspace_get_item(workspace_t *w, int index) {
    return (index >= 0 && index < w->count) ? w->items[index] : NULL;
}

CVS_EXPORT void workspace_get_item_pos(workspace_item_t *item, int64_t *x, int64_t *length, int64_t *z) {
    if (x) *x = item->x;
    if (length) *length = item->length;
    if (z) *z = item->z;
}

CVS_EXPORT int64_t workspace_get_item_offset(workspace_item_t *item) { return item->offset; }
CVS_EXPORT void workspace_set_item_offset(workspace_item_t *item, int64_t offset) { item->offset = offset; }
CVS_EXPORT void *workspace_get_item_source(workspace_item_t *item) { return item->source; }
CVS_EXPORT void workspace_set_item_source(workspace_item_t *item, void *source) { item->source = source; }
CVS_EXPORT void *workspace_get_item_tag(workspace_item_t *item) { return item->tag; }
CVS_EXPORT void workspace_set_item_tag(workspace_item_t *item, void *tag) { item->tag = tag; }

CVS_EXPORT void workspace_update_item(workspace_item_t *item, int64_t *x, int64_t *length, int64_t *z, int64_t *offset, void **source, void **tag) {
    workspace_t *w = item->workspace;
    pthread_mutex_lock(&w->mutex);
    if (x || length || z) {
        take_out(w, item);
        if (x) item->x = *x;
        if (length) item->length = *length;
        if (z) item->z = *z;
        insert_sorted(w, item);
    }
    if (offset) item->offset = *offset;
    if (source) item->source = *source;
    if (tag) item->tag = *tag;
    pthread_mutex_unlock(&w->mutex);
}

CVS_EXPORT void workspace_remove_item(workspace_item_t *item) {
    workspace_t *w = item->workspace;
    pthread_mutex_lock(&w->mutex);
    take_out(w, item);
    item->workspace = NULL;
    pthread_mutex_unlock(&w->mutex);
    free(item);
}

CVS_EXPORT void workspace_free(workspace_t *w) {
    if (!w) return;
    for (int i = 0; i < w->count; i++) free(w->items[i]);
    free(w->items);
    pthread_mutex_destroy(&w->mutex);
    free(w);
}

/* ---- frame ---- */

typedef struct { void *source; int frame; int64_t z; uint64_t serial; } live_item;

static int stack_order(const void *pa, const void *pb) {
    const live_item *a = pa, *b = pb;
    int r = cmp64(a->z, b->z);
    return r ? r : (a->serial > b->serial) - (a->serial < b->serial);
}

/* snapshot under the mutex, composite outside it (workspace.c:496-522) */
static live_item *snapshot(workspace_t *w, int frame_index, int *n_out) {
    pthread_mutex_lock(&w->mutex);
    live_item *live = malloc(sizeof(*live) * (size_t)(w->count ? w->count : 1));
    int n = 0;
    for (int i = 0; live && i < w->count; i++) {
        const workspace_item_t *it = w->items[i];
        if (it->x <= frame_index && frame_index < it->x + it->length) {
            live[n].source = it->source;
            live[n].frame = (int)(frame_index - it->x + it->offset);
            live[n].z = it->z;
            live[n].serial = it->serial;
            n++;
        }
    }
    pthread_mutex_unlock(&w->mutex);
    if (live) qsort(live, (size_t)n, sizeof *live, stack_order);
    *n_out = live ? n : 0;
    return live;
}

static void workspace_get_frame_f32(workspace_t *w, int frame_index, rgba_frame_f32 *frame) {
    int n;
    live_item *live = snapshot(w, frame_index, &n);
    if (!n) { box2i_set_empty(&frame->current_window); free(live); return; }

    video_get_frame_f32((video_source *)live[0].source, live[0].frame, frame);
    if (n > 1) {
        rgba_frame_f32 tmp;
        size_t px = cvs_box_pixels(&frame->full_window);
        tmp.data = malloc(sizeof(rgba_f32) * (px ? px : 1));
        tmp.full_window = frame->full_window;
        for (int i = 1; tmp.data && i < n; i++) {
            box2i_set_empty(&tmp.current_window);
            video_get_frame_f32((video_source *)live[i].source, live[i].frame, &tmp);
            video_mix_over_f32(frame, &tmp, 1.0f);
        }
        free(tmp.data);
    }
    free(live);
}

/* the same stack with every frame resident in HBM */
static void workspace_get_frame_dev(workspace_t *w, int frame_index, rgba_frame_dev *frame) {
    int n;
    live_item *live = snapshot(w, frame_index, &n);
    if (!n || cvs_enter() != 0) { box2i_set_empty(&frame->current_window); free(live); return; }
    const size_t px = cvs_box_pixels(&frame->full_window);
    int rc = 0;

    /* Fast path.  A source that fills only the f16 slot is half-native: an f32 pull of it is "pull f16, widen"
     * (main.c:105-144).  When every live item is such a source and the caller wants f16, the layers are pulled as
     * f16 device frames and the whole stack -- widen, over at mix 1.0 per layer, truncate -- is one launch of the
     * chain kernel without a colour stage (8 B per layer pixel + 8 B written instead of f32 frames through HBM);
     * the chain entry itself goes node by node when a layer's window is not the whole frame. */
    bool all_half = frame->format == CVS_FORMAT_F16 && n <= CVS_CHAIN_MAX_LAYERS && px > 0;
    for (int i = 0; all_half && i < n; i++) {
        const video_source *src = live[i].source;
        all_half = src && src->funcs && src->funcs->get_frame && !src->funcs->get_frame_32;
    }
    if (all_half) {
        rgba_frame_f16 layers[CVS_CHAIN_MAX_LAYERS];
        rgba_frame_f16 out = { frame->data, frame->full_window, frame->full_window };
        cvs_chain_job job;
        memset(&job, 0, sizeof job);
        memset(layers, 0, sizeof layers);
        job.out = &out;
        job.nlayers = n;
        for (int i = 0; rc == 0 && i < n; i++) {
            rgba_frame_dev d = { cvs_pool_malloc(px * sizeof(rgba_f16), frame->stream), CVS_FORMAT_F16, frame->full_window, frame->full_window, frame->stream };
            if (!d.data) { rc = -1; break; }
            video_get_frame_dev((video_source *)live[i].source, live[i].frame, &d);
            layers[i].data = d.data; layers[i].full_window = d.full_window; layers[i].current_window = d.current_window;
            job.layers[i] = &layers[i];
        }
        if (rc == 0) rc = cvs_chain_color_over_f16_dev(&job, 1, NULL, CVS_LUT_NONE, CVS_LUT_NONE, frame->stream);
        for (int i = 0; i < n; i++) cvs_pool_free(layers[i].data, frame->stream);
        if (rc == 0) frame->current_window = out.current_window;
        else box2i_set_empty(&frame->current_window);
        free(live);
        return;
    }

    /* the stack is computed in f32 whatever the caller's format is (workspace.c:530-544) */
    rgba_frame_dev acc = { NULL, CVS_FORMAT_F32, frame->full_window, frame->full_window, frame->stream };
    rgba_frame_dev tmp = acc;
    bool own_acc = frame->format != CVS_FORMAT_F32;
    acc.data = own_acc ? cvs_pool_malloc(px * sizeof(rgba_f32), frame->stream) : frame->data;
    if (!acc.data) rc = -1;

    if (rc == 0) video_get_frame_dev((video_source *)live[0].source, live[0].frame, &acc);
    if (rc == 0 && n > 1) {
        tmp.data = cvs_pool_malloc(px * sizeof(rgba_f32), frame->stream);
        if (!tmp.data) rc = -1;
        for (int i = 1; rc == 0 && i < n; i++) {
            box2i_set_empty(&tmp.current_window);
            tmp.current_window = tmp.full_window;
            video_get_frame_dev((video_source *)live[i].source, live[i].frame, &tmp);
            rgba_frame_f32 fa = { acc.data, acc.full_window, acc.current_window };
            rgba_frame_f32 fb = { tmp.data, tmp.full_window, tmp.current_window };
            rc = cvs_mix_over_f32_dev(&fa, &fb, 1.0f, frame->stream);
            acc.current_window = fa.current_window;
        }
    }
    if (rc == 0 && own_acc) {
        rgba_frame_f32 fa = { acc.data, acc.full_window, acc.current_window };
        rgba_frame_f16 fo = { frame->data, frame->full_window, frame->full_window };
        rc = cvs_frame_f32_to_f16_dev(&fo, &fa, frame->stream);
        acc.current_window = fo.current_window;
    }
    /* scratch goes back to the stream-ordered pool: nothing waits here */
    if (tmp.data) cvs_pool_free(tmp.data, frame->stream);
    if (own_acc && acc.data) cvs_pool_free(acc.data, frame->stream);
    if (rc == 0) frame->current_window = acc.current_window;
    else box2i_set_empty(&frame->current_window);
    free(live);
}

static video_frame_source_funcs workspace_video_funcs = {
    .flags = VIDEO_SOURCE_FLAG_DEVICE,
    .get_frame = NULL,
    .get_frame_32 = (video_get_frame_32_func)workspace_get_frame_f32,
    .get_frame_dev = (video_get_frame_dev_func)workspace_get_frame_dev,
};

CVS_EXPORT void workspace_as_video_source(workspace_t *workspace, video_source *source) {   /* workspace.c:604-613 */
    source->obj = workspace;
    source->funcs = &workspace_video_funcs;
}
