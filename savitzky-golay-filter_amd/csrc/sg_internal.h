/* sg_internal.h -- declarations shared by the host-side C/C++ sources and the HIP launchers. */
#ifndef SG_INTERNAL_H
#define SG_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include "savgolFilter.h"
#include "savgol_stream.h"
#include "savgol2d.h"
#include "savgol_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- sg_weights.c (host tables; compiled with -ffp-contract=off) ---- */
int   sg_weights_valid(int n, int m, int d, float time_step);
void  sg_weights_fill(SavgolFilter *f);                 /* config already copied into *f */
int   sg2d_term(int px, int py);
int   sg2d_config_ok(const Savgol2DConfig *c);
int   sg2d_weights_fill(const Savgol2DConfig *c, float *W, double *coef);
float sg2d_scale(const Savgol2DConfig *c);

/* ---- process-wide switches (sg_api_1d.cpp, savgol_hip_set_option) ---- */
int   sg_option_boundary_aware(void);                    /* SAVGOL_HIP_OPT_BOUNDARY_AWARE */

/* ---- error reporting (sg_runtime.cpp) ---- */
void  sg_set_error(const char *fmt, ...);

#ifdef __cplusplus
}
#endif
#endif
