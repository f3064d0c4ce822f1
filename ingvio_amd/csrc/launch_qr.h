// launch_qr.h — general blocked-Householder QR compression (kernels_qr.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// device workspace (doubles) for an m x n problem
size_t qr_dense_workspace_doubles(int m, int n);
// dH m x n column-major (ldh), dres [m] -> dHt n x n upper triangular column-major (ldt), drt [n]; -1: m too large
int launch_qr_dense(const double* dH, int ldh, const double* dres, int m, int n, double* ws, double* dHt, int ldt, double* drt, hipStream_t st);

// Cholesky-QR for tall stacks (m >> n): Gram GEMM + blocked Cholesky (kernels_chol.hip); same outputs, diagonal of R positive
size_t qr_chol_workspace_doubles(int m, int n);
int launch_qr_chol(const double* dH, int ldh, const double* dres, int m, int n, double* ws, double* dHt, int ldt, double* drt, hipStream_t st);
