/*
 * dv.c -- CPU restatement of the DV 4:1:1 edge of the path: planar 8-bit Y'CbCr <-> half RGBA.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *   orc_reconstruct_dv   src/cprocess/video_reconstruct.c:50-137
 *   orc_subsample_dv     src/cprocess/video_subsample.c:99-187
 *
 * Both work on the fixed 720x480 NTSC DV raster, placed on the frame plane with its first line at y = -1
 * ("line zero is part of the first field"), chroma subsampled 4:1 horizontally and co-sited with the left
 * pixel, Rec.709 matrix and transfer function.  No reference test covers them: parity unpinned beyond this
 * line-by-line restatement.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define PX(f, X, Y) (&(f)->data[(ptrdiff_t)((Y) - (f)->full_window.min.y) * \
        ((f)->full_window.max.x - (f)->full_window.min.x + 1) + ((X) - (f)->full_window.min.x)])

enum { DV_W = 720, DV_H = 480, DV_SUB = 4, DV_OFF_Y = -1 };

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

typedef struct { float cb, cr; } cbcr;

void orc_reconstruct_dv(orc_frame16 *frame, const uint8_t *const planes[3], const int strides[3]) {
    /* video_reconstruct.c:62-66 */
    const float m[3][3] = { { 1.0f, 0.0f, 1.5748f }, { 1.0f, -0.187324f, -0.468124f }, { 1.0f, 1.8556f, 0.0f } };
    orc_box2i *cw = &frame->current_window;
    cw->min.x = imax(0, frame->full_window.min.x);                                   /* :72-76 */
    cw->min.y = imax(DV_OFF_Y, frame->full_window.min.y);
    cw->max.x = imin(DV_W - 1, frame->full_window.max.x);
    cw->max.y = imin(DV_H + DV_OFF_Y - 1, frame->full_window.max.y);
    orc_fir tri = { NULL, 0, 0 };
    orc_fir_triangle((float)DV_SUB, 0.0f, &tri);                                       /* :83-84 */
    orc_px32 *row32 = malloc(sizeof(orc_px32) * DV_W);
    cbcr *chroma = malloc(sizeof(cbcr) * DV_W);
    for (int row = cw->min.y - DV_OFF_Y; row <= cw->max.y - DV_OFF_Y; row++) {          /* :91 */
        const uint8_t *yrow = planes[0] + (ptrdiff_t)row * strides[0];
        const uint8_t *cbrow = planes[1] + (ptrdiff_t)row * strides[1];
        const uint8_t *crrow = planes[2] + (ptrdiff_t)row * strides[2];
        memset(chroma, 0, sizeof(cbcr) * DV_W);
        for (int x = 0; x <= (DV_W - 1) / DV_SUB; x++) {                                /* :98-109: scatter each chroma sample */
            const float cb = (cbrow[x] - 128.0f) / 224.0f, cr = (crrow[x] - 128.0f) / 224.0f;    /* :30-33 */
            for (int i = imax(cw->min.x, x * DV_SUB - tri.center); i <= imin(cw->max.x, x * DV_SUB + (tri.width - tri.center - 1)); i++) {
                chroma[i].cb += cb * tri.coeff[i - x * DV_SUB + tri.center];
                chroma[i].cr += cr * tri.coeff[i - x * DV_SUB + tri.center];
            }
        }
        for (int x = cw->min.x; x <= cw->max.x; x++) {                                  /* :111-124 */
            const float y = (yrow[x] - 16.0f) / 219.0f;                                 /* :35-38 */
            row32[x].r = y * m[0][0] + chroma[x].cb * m[0][1] + chroma[x].cr * m[0][2];
            row32[x].g = y * m[1][0] + chroma[x].cb * m[1][1] + chroma[x].cr * m[1][2];
            row32[x].b = y * m[2][0] + chroma[x].cb * m[2][1] + chroma[x].cr * m[2][2];
            row32[x].a = 1.0f;
        }
        if (cw->max.x >= cw->min.x) {
            orc_px16 *out = PX(frame, cw->min.x, row + DV_OFF_Y);
            const int n = cw->max.x - cw->min.x + 1;
            orc_float_to_half(&out->r, &row32[cw->min.x].r, 4 * n);                     /* :128-129 */
            orc_transfer(ORC_LUT_REC709_TO_LINEAR_SCENE, &out->r, &out->r, (size_t)4 * n);     /* :130-131: alpha too */
        }
    }
    orc_fir_free(&tri);
    free(row32); free(chroma);
}

/* planes: Y 720x480, Cb and Cr 180x480, zero-filled first (coded_image_alloc0, :125).  The frame's rows inside
 * the window are transfer-encoded IN PLACE on the way (:144) -- the caller sees that, so it is kept. */
void orc_subsample_dv(uint8_t *const planes[3], const int strides[3], orc_frame16 *frame) {
    const float m[3][3] = { { 0.2126f, 0.7152f, 0.0722f }, { -0.114572f, -0.385428f, 0.5f }, { 0.5f, -0.454153f, -0.045847f } };   /* :104-108 */
    const int lines[3] = { DV_H, DV_H, DV_H }, widths[3] = { DV_W, DV_W / DV_SUB, DV_W / DV_SUB };
    for (int p = 0; p < 3; p++)
        for (int r = 0; r < lines[p]; r++) memset(planes[p] + (ptrdiff_t)r * strides[p], 0, (size_t)widths[p]);
    orc_box2i w;                                                                        /* :121-126 */
    w.min.x = imax(0, frame->current_window.min.x);
    w.min.y = imax(DV_OFF_Y, frame->current_window.min.y);
    w.max.x = imin(DV_W - 1, frame->current_window.max.x);
    w.max.y = imin(DV_H + DV_OFF_Y - 1, frame->current_window.max.y);
    const int ww = w.max.x - w.min.x + 1;
    if (ww <= 0 || w.max.y < w.min.y) return;
    orc_fir tri = { NULL, 0, 0 };
    orc_fir_triangle(1.0f / (float)DV_SUB, 0.0f, &tri);                                  /* :132-133 */
    orc_px32 *row32 = malloc(sizeof(orc_px32) * (size_t)ww);
    cbcr *chroma = malloc(sizeof(cbcr) * (size_t)ww);
    for (int row = w.min.y - DV_OFF_Y; row <= w.max.y - DV_OFF_Y; row++) {
        uint8_t *yrow = planes[0] + (ptrdiff_t)row * strides[0];
        uint8_t *cbrow = planes[1] + (ptrdiff_t)row * strides[1];
        uint8_t *crrow = planes[2] + (ptrdiff_t)row * strides[2];
        orc_px16 *in = PX(frame, w.min.x, row + DV_OFF_Y);
        orc_transfer(ORC_LUT_LINEAR_TO_REC709, &in->r, &in->r, (size_t)4 * ww);          /* :144, in place */
        orc_half_to_float(&row32->r, &in->r, 4 * ww);
        for (int x = 0; x < ww; x++) {                                                  /* :147-161 */
            const float y = (row32[x].r * m[0][0] + row32[x].g * m[0][1] + row32[x].b * m[0][2]) * 219.0f + 16.0f;
            chroma[x].cb = row32[x].r * m[1][0] + row32[x].g * m[1][1] + row32[x].b * m[1][2];
            chroma[x].cr = row32[x].r * m[2][0] + row32[x].g * m[2][1] + row32[x].b * m[2][2];
            yrow[x + w.min.x] = (uint8_t)(int32_t)y;        /* (uint8_t)float as the reference's x86 build does it: truncate to int, keep the low byte */
        }
        for (int tx = w.min.x / DV_SUB; tx <= w.max.x / DV_SUB; tx++) {                  /* :163-175: gather */
            float cb = 0.0f, cr = 0.0f;
            for (int sx = imax(w.min.x, tx * DV_SUB - tri.center); sx <= imin(w.max.x, tx * DV_SUB + (tri.width - tri.center - 1)); sx++) {
                cb += chroma[sx - w.min.x].cb * tri.coeff[sx - tx * DV_SUB + tri.center];
                cr += chroma[sx - w.min.x].cr * tri.coeff[sx - tx * DV_SUB + tri.center];
            }
            cbrow[tx] = (uint8_t)(int32_t)(cb * 224.0f + 128.0f);                       /* :62-65 */
            crrow[tx] = (uint8_t)(int32_t)(cr * 224.0f + 128.0f);
        }
    }
    orc_fir_free(&tri);
    free(row32); free(chroma);
}
