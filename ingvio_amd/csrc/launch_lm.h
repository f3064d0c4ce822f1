// launch_lm.h — host-side launchers of kernels_lm.hip (SLAM-landmark covariance operations)
#pragma once
#include <hip/hip_runtime.h>

#include "dev_common.h"

// Givens QR of H_new (m x s, column stride mld) applied to H_old (m x nc, mld) and res (m); in place.  -1: capacity
int launch_delayed_qr(double* H_old, double* res, double* H_new, int m, int s, int nc, int mld, hipStream_t st);
// addVariableDelayedInvertible with the top s rows of H_old (Hx) / H_new (Hf); Y: n x s scratch (ld = ldp); n[b] += s
int launch_delayed_add(CovView cv, int b, const double* Hx, const double* Hf, const int* colmap, int s, int nc, int mld, double var,
                       double* Y, hipStream_t st);
// replaceVarLinear: H ts x nc (mld)
int launch_replace_var(CovView cv, int b, const double* H, const int* colmap, int tidx, int ts, int nc, int mld, double* Y, hipStream_t st);
