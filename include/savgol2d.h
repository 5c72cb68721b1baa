/*
 * savgol2d.h -- 2-D Savitzky-Golay filter (images), MI355X (gfx950) implementation.
 *
 * Drop-in for the reference's include/iterative/savgol2d.h: same config/filter structs
 * (16 / 48 bytes; `weights` is a heap array the caller may read), boundary enum, the eight
 * exported functions (:126-269) and the two header-inline helpers (:250-264).
 *
 * The kernel is the dense (2ny+1) x (2nx+1) least-squares kernel of a total-degree polynomial
 * fit (x^i y^j, i+j <= order), computed on the host in double exactly as the reference does
 * (src/savgol2d.c:188-265).  Applying it is done on the GPU: LDS-tiled direct convolution, or --
 * because that kernel is exactly low rank -- a sum of a few row/column separable passes
 * (see DESIGN.md).  x = columns, y = rows.
 */
#ifndef SAVGOL2D_H
#define SAVGOL2D_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAVGOL2D_MAX_HALF_WINDOW 16
#define SAVGOL2D_MAX_POLY_ORDER  6
#define SAVGOL2D_MAX_TERMS       28
#define SAVGOL2D_MAX_WINDOW_AREA ((2*SAVGOL2D_MAX_HALF_WINDOW+1)*(2*SAVGOL2D_MAX_HALF_WINDOW+1))

typedef struct {
    uint8_t half_window_x;   /* columns: window width  = 2nx+1                 */
    uint8_t half_window_y;   /* rows:    window height = 2ny+1                 */
    uint8_t poly_order;      /* total degree of the fitted surface             */
    uint8_t deriv_x;         /* d^dx/dx^dx ...                                 */
    uint8_t deriv_y;         /* ... d^dy/dy^dy, dx+dy <= poly_order            */
    float   delta_x;         /* sample spacing along x, > 0                    */
    float   delta_y;         /* sample spacing along y, > 0                    */
} Savgol2DConfig;

typedef struct Savgol2DFilter {
    Savgol2DConfig config;
    int    window_width;
    int    window_height;
    int    window_area;
    int    num_terms;        /* (order+1)(order+2)/2                           */
    float  scale;            /* 1 / (delta_x^dx * delta_y^dy)                  */
    float *weights;          /* [window_height][window_width], row-major       */
} Savgol2DFilter;

typedef enum {
    SAVGOL2D_BOUNDARY_VALID = 0,   /* only pixels whose whole window exists    */
    SAVGOL2D_BOUNDARY_CONSTANT,    /* clamp coordinates to the frame           */
    SAVGOL2D_BOUNDARY_REFLECT      /* half-sample mirror, then clamp           */
} Savgol2DBoundary;

Savgol2DFilter *savgol2d_create(const Savgol2DConfig *config);
void            savgol2d_destroy(Savgol2DFilter *filter);

/* (rows-2ny) x (cols-2nx) outputs written at output[0..]; strides are in elements. -1 on error. */
int savgol2d_apply_valid(const Savgol2DFilter *filter,
                         const float *input, int rows, int cols, int in_stride,
                         float *output, int out_stride);

/* Same-size output.  VALID writes only the interior and leaves the border untouched. */
int savgol2d_apply(const Savgol2DFilter *filter,
                   const float *input, int rows, int cols, int in_stride,
                   float *output, int out_stride,
                   Savgol2DBoundary boundary);

/* First derivatives; either output may be NULL. */
int savgol2d_gradient(int half_win_x, int half_win_y, int poly_order,
                      const float *input, int rows, int cols, int stride,
                      float *grad_x, float *grad_y,
                      float delta_x, float delta_y,
                      Savgol2DBoundary boundary);

/* Second derivatives (poly_order >= 2); any output may be NULL. */
int savgol2d_hessian(int half_win_x, int half_win_y, int poly_order,
                     const float *input, int rows, int cols, int stride,
                     float *hess_xx, float *hess_xy, float *hess_yy,
                     float delta_x, float delta_y,
                     Savgol2DBoundary boundary);

/* d2/dx2 + d2/dy2 (poly_order >= 2). */
int savgol2d_laplacian(int half_win_x, int half_win_y, int poly_order,
                       const float *input, int rows, int cols, int stride,
                       float *output,
                       float delta_x, float delta_y,
                       Savgol2DBoundary boundary);

static inline void savgol2d_valid_size(const Savgol2DFilter *filter,
                                       int in_rows, int in_cols,
                                       int *out_rows, int *out_cols)
{
    *out_rows = in_rows - 2 * filter->config.half_window_y;
    *out_cols = in_cols - 2 * filter->config.half_window_x;
}

static inline int savgol2d_num_terms(int poly_order)
{
    return (poly_order + 1) * (poly_order + 2) / 2;
}

bool savgol2d_config_valid(const Savgol2DConfig *config);

#ifdef __cplusplus
}
#endif
#endif /* SAVGOL2D_H */
