/*
 * kernels.h -- internal seam between the host C code (csrc/host) and the HIP kernels
 * (the .hip files next to this header).  Plain C types only; every launcher enqueues on `stream` and returns a
 * hipError_t as int (0 = ok).  Host code has already done all window arithmetic: the kernels
 * get explicit rectangles and never look at box2i / current_window.
 */
#ifndef CVS_KERNELS_H
#define CVS_KERNELS_H

#include <stddef.h>
#include <stdint.h>

/* Measurement knobs of tools/ exist only in the diagnostic build (make diag, -DCVS_DIAG); in the shipped library the
 * macro is a null pointer constant and the branches that read it fold away -- no environment variable changes what the
 * product computes or launches. */
#ifdef CVS_DIAG
#include <stdlib.h>
#define CVS_DIAG_ENV(name) getenv(name)
#else
#define CVS_DIAG_ENV(name) ((const char *)0)
#endif

/* Arithmetic flavours (pixel_math.hpp): the translation units that hold f32 arithmetic are compiled twice, plain and with
 * -DCVS_CONTRACT (a * b + c inside one expression of the reference's C becomes ONE fused multiply-add, as the reference's
 * preferred clang build computes it).  The contracted objects export the same launchers under *_fma names; the host picks per
 * call (internal.h CVK()).  Units without arithmetic (conversions, copies, table look-ups, the display edge) exist once. */
#ifdef CVS_CONTRACT
#define cvk_gain_offset_f16          cvk_gain_offset_f16_fma
#define cvk_mix                      cvk_mix_fma
#define cvk_color_matrix             cvk_color_matrix_fma
#define cvk_chain_color_over         cvk_chain_color_over_fma
#define cvk_chain_cross              cvk_chain_cross_fma
#define cvk_chain_count_reset        cvk_chain_count_reset_fma
#define cvk_chain_count              cvk_chain_count_fma
#define cvk_fir_gather               cvk_fir_gather_fma
#define cvk_fir2d_lds_bytes          cvk_fir2d_lds_bytes_fma
#define cvk_fir2d                    cvk_fir2d_fma
#define cvk_fir_vh_supported         cvk_fir_vh_supported_fma
#define cvk_fir_vh                   cvk_fir_vh_fma
#define cvk_fir_tvh_supported        cvk_fir_tvh_supported_fma
#define cvk_fir_tvh_preferred        cvk_fir_tvh_preferred_fma
#define cvk_fir_tvh                  cvk_fir_tvh_fma
#define cvk_blur_supported           cvk_blur_supported_fma
#define cvk_blur_takes_pairs         cvk_blur_takes_pairs_fma
#define cvk_blur                     cvk_blur_fma
#define cvk_blur_pair_supported      cvk_blur_pair_supported_fma
#define cvk_blur_pair                cvk_blur_pair_fma
#define cvk_blur_halve_pair_supported cvk_blur_halve_pair_supported_fma
#define cvk_blur_halve_pair          cvk_blur_halve_pair_fma
#define cvk_blur_halve_supported     cvk_blur_halve_supported_fma
#define cvk_blur_halve_takes_pairs   cvk_blur_halve_takes_pairs_fma
#define cvk_blur_halve               cvk_blur_halve_fma
#define cvk_dv_reconstruct           cvk_dv_reconstruct_fma
#define cvk_dv_subsample             cvk_dv_subsample_fma
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* a frame buffer as the kernels see it: base pointer + the full window it covers */
typedef struct {
    void *data;          /* device pointer to pixel (fx0, fy0) */
    int pitch;           /* pixels per row = fx1 - fx0 + 1 */
    int fx0, fy0, fx1, fy1;
} cvk_view;

typedef struct { int x0, y0, x1, y1; } cvk_rect;     /* inclusive */

/* flat arrays */
int cvk_half_to_float(float *out, const uint16_t *in, size_t count, int fast, void *stream);
int cvk_float_to_half(uint16_t *out, const float *in, size_t count, int fast, void *stream);
int cvk_half_lookup(const uint16_t *table, uint16_t *out, const uint16_t *in, size_t count, int cus, void *stream);

/* windowed frame ops; rect must lie inside every view involved */
int cvk_copy_f16(cvk_view out, cvk_view in, cvk_rect r, void *stream);
int cvk_copy_alpha_f32(cvk_view out, cvk_view in, cvk_rect r, float alpha, void *stream);
/* even rows of `cur` in `frame` <- rows of `other` (packed, allocated for exactly `cur`), Pulldown23RemovalFilter.c:88-104 */
int cvk_weave_f16(cvk_view frame, cvk_rect cur, const void *other, cvk_rect other_cur, void *stream);
int cvk_widen(cvk_view out32, cvk_view in16, cvk_rect r, void *stream);
int cvk_narrow(cvk_view out16, cvk_view in32, cvk_rect r, void *stream);
int cvk_fill_f16(cvk_view out, cvk_rect r, const float rgba[4], void *stream);      /* truncates to half in the kernel */
int cvk_fill_f32(cvk_view out, cvk_rect r, const float rgba[4], void *stream);
int cvk_gain_offset_f16(cvk_view out, cvk_view in, cvk_rect r, float gain, float offset, void *stream);

/* two-input mixers (video_mix.c region walk, decided on the host) */
enum { CVK_MIX_CROSS = 0, CVK_MIX_OVER = 1 };
typedef struct {
    cvk_view out, p, q;            /* for OVER p aliases out */
    cvk_rect outer, inner;         /* inner already normalised (gap form when empty) */
    cvk_rect pw, qw;               /* current windows of p and q */
    int gap_x, gap_y;
    int top_is_p, bottom_is_p, left_is_p, right_is_p;
    float wp, wq;                  /* alpha weights of lone pixels / blend mixes */
    int mode;                      /* CVK_MIX_* */
    int p_in_place;                /* p's own pixels are already in out: leave them */
} cvk_mix_params;
int cvk_mix(const cvk_mix_params *mp, void *stream);

/* colour matrix on an f16 frame (dst may be the same buffer as src), color.c structure; LUT pointers are
 * device tables or NULL */
int cvk_color_matrix(cvk_view dst, cvk_view src, cvk_rect r, const float m[9], const uint16_t *pre_lut,
                     const uint16_t *post_lut, int cus, void *stream);

/* fused colour + over chain; `jobs` is a HOST array (records are passed as kernel arguments,
 * 32 per launch); the buffers they point to are in HBM */
#define CVK_CHAIN_MAX_LAYERS 8
typedef struct {
    void *out;                               /* rgba_f16 device buffer, npixels long */
    const void *layer[CVK_CHAIN_MAX_LAYERS]; /* rgba_f16 device buffers, bottom first */
    int nlayers;
    int pad;
    uint64_t npixels;
} cvk_chain_job;
/* m == NULL: no colour stage (plain over stack of the f16 layers) */
int cvk_chain_color_over(const cvk_chain_job *jobs, int njobs, int uniform_layers, const float *m,
                         const uint16_t *pre_lut, const uint16_t *post_lut, int cus, void *stream);

/* launches of the trip-loop kernel since the calling thread's last reset (a batch is cut into launches of ~8 4K frames) */
void cvk_chain_count_reset(void);
int cvk_chain_count(void);

/* the same machinery crossfading two-layer jobs: out = cross(layer[0], layer[1]), weights wa = 1 - mix_b, wb = mix_b */
int cvk_chain_cross(const cvk_chain_job *jobs, int njobs, float wa, float wb, int cus, void *stream);

/* separable FIR passes (video_scale.c structure) */
typedef struct {
    cvk_view target, source;
    int axis;                 /* 0: resample along y, 1: along x */
    int t0, t1;               /* target lines to produce along the axis */
    int lo, hi;               /* span on the other axis */
    int s0, s1;               /* valid source lines along the axis (taps outside are skipped) */
    const int *ntaps;         /* device: per target line, tap count */
    const int *tap_src;       /* device: per target line, `stride` source line indices, ascending */
    const float *taps;        /* device: per target line, `stride` coefficients */
    int stride;
    int in_half, out_half;    /* 0: rgba_f32 pixels, 1: rgba_f16 pixels (widened on load / truncated on store) */
} cvk_fir_params;
int cvk_fir_gather(const cvk_fir_params *fp, void *stream);
int cvk_zero_f32(cvk_view v, void *stream);

/* separable FIR, both passes in one launch with the source tile and the horizontal result staged in LDS.
 * One device table per axis (cvk_fir_axis): for target line i (0-based within the rect) ntaps[i] taps,
 * source indices src[i*stride + k] (ascending) and weights taps[i*stride + k]; foot[2*t], foot[2*t+1] =
 * first and last source index any line of tile t touches (first > last: the tile touches nothing); the list is padded with
 * such entries to a multiple of four. */
typedef struct {
    const int *ntaps, *src;
    const float *taps;
    const int *foot;
    int stride;
    int lines;                 /* target lines the table covers (ntaps is also the start of the table's device block: CVK_AXIS_OFF_*) */
    /* facts about the table the host worked out when it built it */
    int max_taps;              /* longest tap list */
    int wide_foot;             /* widest source footprint of any run of 128 lines starting at a multiple of 128 */
    int max_active;            /* as the vertical axis: the longest run of lines i..j such that line j starts at or before line i's last tap */
    int streamable;            /* every list is consecutive source lines, first and last taps never decrease from line to line */
    /* the table by TARGET line, one record of CVK_FIR_LREC dwords per line (present when streamable and no list is longer
     * than CVK_FIR_LREC - 2): [0] tap count, [1] first source line (INT_MIN for a line without taps), [2 + k] weight of
     * tap k; unused entries 0.  One spare all-zero record (count 0) follows the last.  sweep_vh_ops.hip reads one per line. */
    const uint32_t *lrec;
    /* the table by target line for short lists (longest <= 4; width pack_width = 2 or 4): pack_width source lines, then
     * pack_width weights per line, 32-bit each; entries past the line's count: source line INT_MIN, weight 0.  One aligned
     * 16- or 32-byte read per line instead of 1 + 2 x count scattered ones (tile_vh_ops.hip: lane = line).  NULL otherwise. */
    const uint32_t *pack;
    int pack_width;
    int span_lines[3];         /* (streamable tables) most source lines any 16, 32, 64 consecutive target lines reach, first tap of the first to last tap of the last */
} cvk_fir_axis;
#define CVK_FIR_LREC 32
/* One device block per table: ntaps | src | taps | foot | lrec, each part on a 256-byte boundary, ntaps first.  A kernel that
 * is handed the block's start can form the other pointers itself (tile_vh_ops.hip does, to have them before its arguments
 * arrive); the host lays the block out with the same macros (scale.c axis_upload). */
#define CVK_AXIS_ALIGN(bytes)              (((size_t)(bytes) + 255) & ~(size_t)255)
#define CVK_AXIS_NLINES(lines)             ((size_t)((lines) > 0 ? (lines) : 1))
#define CVK_AXIS_OFF_SRC(lines)            CVK_AXIS_ALIGN(CVK_AXIS_NLINES(lines) * 4)
#define CVK_AXIS_OFF_TAPS(lines, stride)   (CVK_AXIS_OFF_SRC(lines) + CVK_AXIS_ALIGN(CVK_AXIS_NLINES(lines) * (size_t)(stride) * 4))
#define CVK_AXIS_OFF_FOOT(lines, stride)   (CVK_AXIS_OFF_TAPS(lines, stride) + CVK_AXIS_ALIGN(CVK_AXIS_NLINES(lines) * (size_t)(stride) * 4))
#define CVK_FIR_TVH_LINES 16     /* the shortest segment of tile_vh_ops.hip (span_lines[0]); [1], [2]: twice, four times as many */
typedef struct {
    cvk_view target, source;
    int in_half, out_half;     /* 0: rgba_f32 pixels, 1: rgba_f16 pixels */
    int tx0, ty0, tx1, ty1;    /* target rectangle */
    cvk_fir_axis h, v;         /* h: per target column, v: per target row */
    int max_sw, max_sh;        /* largest tile footprint in source pixels (sizes the LDS tile) */
    /* cvk_fir_tvh and cvk_fir_vh only: nframes > 1 = that many frames of this geometry in one launch (grid.z = frame); their
     * data pointers replace target.data / source.data */
    int nframes, pad_;
    const void *frame_source[8];
    void *frame_target[8];
} cvk_fir2d_params;
#define CVK_FIR2D_TILE_X 32
#define CVK_FIR2D_TILE_Y 16
/* the same tables, horizontal pass first, as a gather per target line (sweep_hv_ops.hip): needs v.streamable and
 * v.lrec (vertical lists <= CVK_FIR_LREC - 2), horizontal lists <= 24; first choice for these tables */
int cvk_fir_hv_supported(const cvk_fir2d_params *fp);
int cvk_fir_hv(const cvk_fir2d_params *fp, int cus, void *stream);
/* the same tables with the VERTICAL pass first (sweep_vh_ops.hip): what video_scale_bilinear_f32 does when the factors are
 * equal or the vertical one is smaller.  fp->ty0 = first line of the vertical table; lines ty0 + line0 .. ty1 are produced */
int cvk_fir_vh_supported(const cvk_fir2d_params *fp);
int cvk_fir_vh(const cvk_fir2d_params *fp, int line0, int cus, void *stream);
/* the same, a workgroup per tile of 128 columns x CVK_FIR_TVH_LINES lines with both passes through LDS (tile_vh_ops.hip):
 * for tables with short lists and narrow footprints (enlarging); needs v.span_lines and h.wide_foot (columns counted from fp->tx0) */
int cvk_fir_tvh_supported(const cvk_fir2d_params *fp);
int cvk_fir_tvh_preferred(const cvk_fir2d_params *fp);    /* supported, and measured faster than cvk_fir_vh for this format and size */
int cvk_fir_tvh(const cvk_fir2d_params *fp, int line0, void *stream);
size_t cvk_fir2d_lds_bytes(const cvk_fir2d_params *fp);     /* dynamic LDS the launch would need */
int cvk_fir2d(const cvk_fir2d_params *fp, void *stream);

/* Several frames of one geometry in ONE launch of a strip x segment kernel (grid.z = frame): the chip is filled by
 * frames x strips x segments workgroups, so the segments of each frame are taller and re-filter fewer halo rows
 * (a 4K frame alone: 48 useful source rows per 65 of config 3's sweep; four frames: 192 per 209).  n == 0: one frame,
 * the views' own data pointers. */
#define CVK_FRAME_BATCH 8
#define CVK_BLUR_MAX_OVER 4
typedef struct {
    int n, pad;
    const void *source[CVK_FRAME_BATCH];
    void *target[CVK_FRAME_BATCH];
    const void *over[CVK_FRAME_BATCH][CVK_BLUR_MAX_OVER];
} cvk_frame_batch;

/* separable FIR with one tap list for every line (3..31 taps odd, 4..16 even, all finite), optionally decimating by 2:
 * both passes in one sweep, the vertical window in registers.  Source pixels outside (sx0..sx1, sy0..sy1) count as skipped taps. */
typedef struct {
    cvk_view target, source;
    int in_half, out_half;     /* 0: rgba_f32 pixels, 1: rgba_f16 pixels */
    int tx0, ty0, tx1, ty1;    /* target rectangle */
    int sx0, sy0, sx1, sy1;    /* the source's current window */
    int ntaps;
    int rows_per_wg;           /* 0: let the launcher choose */
    int step;                  /* target line t reads source lines step*t - ntaps/2 + k; 0 or 1: blur, 2: halving resampler */
    float taps[32];
    int nover;                 /* f16 frames blended over the blur result before the (f16) store; 0..CVK_BLUR_MAX_OVER */
    int flags;                 /* CVK_BLUR_* */
    const void *over[CVK_BLUR_MAX_OVER];       /* rgba_f16 device buffers laid out exactly like `target` */
    cvk_frame_batch batch;                     /* batch.n frames of this geometry (their pointers replace target / source / over) */
} cvk_blur_params;
#define CVK_BLUR_ONE_COLUMN  1   /* never the two-columns-per-lane form */
#define CVK_BLUR_TWO_COLUMNS 2   /* that form wherever it takes the launch (by itself cvk_blur keeps it for frames of 256 columns and more) */
int cvk_blur_supported(int ntaps, int step);
int cvk_blur(const cvk_blur_params *bp, int cus, void *stream);
int cvk_blur_takes_pairs(const cvk_blur_params *bp);      /* would cvk_blur launch k_blur_pair for this? */
/* the same blur with two target columns per lane (blur_pair_ops.hip): f16 in and out, 1:1, 3..13 taps odd, every buffer,
 * window and pitch such that a pair of columns is one whole, aligned 16-byte access.  cvk_blur goes there by itself. */
int cvk_blur_pair_supported(const cvk_blur_params *bp);
int cvk_blur_pair(const cvk_blur_params *bp, int cus, void *stream);

/* blur (ntaps1 odd, one list for every line) followed by the Lanczos halving resampler (ntaps2 taps, target line t reads
 * blurred lines 2t - ntaps2/2 + k), both separable, in one sweep: blur_halve_ops.hip.  The blurred frame exists only
 * inside the source's current window (sx0..sy1); target rectangle in target coordinates. */
typedef struct {
    cvk_view target, source;
    int in_half, out_half;
    int tx0, ty0, tx1, ty1;
    int sx0, sy0, sx1, sy1;
    int ntaps1, ntaps2;
    int rows_per_wg;           /* 0: let the launcher choose */
    int flags;                 /* CVK_BLUR_ONE_COLUMN / CVK_BLUR_TWO_COLUMNS */
    float taps1[16], taps2[16];
    cvk_frame_batch batch;                     /* batch.n frames of this geometry (over[] unused) */
} cvk_blur_halve_params;
int cvk_blur_halve_supported(int ntaps1, int ntaps2);
/* the same sweep with two source columns per lane and one-wave workgroups (blur_halve_pair_ops.hip): f16 in and out, every
 * pair of source columns one whole, aligned 16-byte access; also ntaps1 == 1 with taps1[0] == 1.0f, the resampler alone.  cvk_blur_halve goes there by itself (cvk_blur_halve_takes_pairs). */
int cvk_blur_halve_pair_supported(const cvk_blur_halve_params *bp);
int cvk_blur_halve_pair(const cvk_blur_halve_params *bp, int cus, void *stream);
int cvk_blur_halve_takes_pairs(const cvk_blur_halve_params *bp);
int cvk_blur_halve(const cvk_blur_halve_params *bp, int cus, void *stream);

/* display / export edge: f16 RGBA -> 4 bytes per pixel through a 65536-entry half->u8 table (device pointer,
 * 16-byte aligned); dst is packed over the rectangle */
enum { CVK_DISPLAY_RGBA8 = 0, CVK_DISPLAY_ARGB32_PREMUL = 1 };
int cvk_display(void *dst, cvk_view src, cvk_rect r, const uint8_t *table, int mode, int cus, void *stream);

/* DV 4:1:1 edge (video_reconstruct.c:50-137, video_subsample.c:99-187): device planes + strides in bytes */
typedef struct { uint8_t *y, *cb, *cr; int sy, scb, scr; } cvk_dv_planes;
typedef struct { float coeff[16]; int width, center; } cvk_dv_taps;
int cvk_dv_reconstruct(cvk_view frame, cvk_rect cur, const cvk_dv_planes *pl, const cvk_dv_taps *tri, const uint16_t *lut, void *stream);
int cvk_dv_subsample(const cvk_dv_planes *pl, cvk_view frame, cvk_rect w, const cvk_dv_taps *tri, const uint16_t *lut, int encode_in_place, void *stream);

/* the contracted twins, as the host sees them (same signatures; built from the same sources with -DCVS_CONTRACT) */
#ifndef CVS_CONTRACT
int cvk_gain_offset_f16_fma(cvk_view out, cvk_view in, cvk_rect r, float gain, float offset, void *stream);
int cvk_mix_fma(const cvk_mix_params *mp, void *stream);
int cvk_color_matrix_fma(cvk_view dst, cvk_view src, cvk_rect r, const float m[9], const uint16_t *pre_lut, const uint16_t *post_lut, int cus, void *stream);
int cvk_chain_color_over_fma(const cvk_chain_job *jobs, int njobs, int uniform_layers, const float *m, const uint16_t *pre_lut, const uint16_t *post_lut, int cus, void *stream);
int cvk_chain_cross_fma(const cvk_chain_job *jobs, int njobs, float wa, float wb, int cus, void *stream);
void cvk_chain_count_reset_fma(void);
int cvk_chain_count_fma(void);
int cvk_fir_gather_fma(const cvk_fir_params *fp, void *stream);
size_t cvk_fir2d_lds_bytes_fma(const cvk_fir2d_params *fp);
int cvk_fir2d_fma(const cvk_fir2d_params *fp, void *stream);
int cvk_fir_vh_supported_fma(const cvk_fir2d_params *fp);
int cvk_fir_vh_fma(const cvk_fir2d_params *fp, int line0, int cus, void *stream);
int cvk_fir_tvh_supported_fma(const cvk_fir2d_params *fp);
int cvk_fir_tvh_preferred_fma(const cvk_fir2d_params *fp);
int cvk_fir_tvh_fma(const cvk_fir2d_params *fp, int line0, void *stream);
int cvk_blur_supported_fma(int ntaps, int step);
int cvk_blur_takes_pairs_fma(const cvk_blur_params *bp);
int cvk_blur_fma(const cvk_blur_params *bp, int cus, void *stream);
int cvk_blur_pair_supported_fma(const cvk_blur_params *bp);
int cvk_blur_pair_fma(const cvk_blur_params *bp, int cus, void *stream);
int cvk_blur_halve_pair_supported_fma(const cvk_blur_halve_params *bp);
int cvk_blur_halve_pair_fma(const cvk_blur_halve_params *bp, int cus, void *stream);
int cvk_blur_halve_supported_fma(int ntaps1, int ntaps2);
int cvk_blur_halve_takes_pairs_fma(const cvk_blur_halve_params *bp);
int cvk_blur_halve_fma(const cvk_blur_halve_params *bp, int cus, void *stream);
int cvk_dv_reconstruct_fma(cvk_view frame, cvk_rect cur, const cvk_dv_planes *pl, const cvk_dv_taps *tri, const uint16_t *lut, void *stream);
int cvk_dv_subsample_fma(const cvk_dv_planes *pl, cvk_view frame, cvk_rect w, const cvk_dv_taps *tri, const uint16_t *lut, int encode_in_place, void *stream);
#endif

#ifdef __cplusplus
}
#endif
#endif
