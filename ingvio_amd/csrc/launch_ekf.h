// launch_ekf.h — host-side launch descriptors of kernels_ekf.hip / kernels_cov.hip.
#pragma once
#include "dev_common.h"

struct EkfLaunch {
    CovView cv;
    int b0, nb;
    const double* H;        // [nb][hstride], column-major ld = mld
    const double* res;      // [nb][mld]
    const int* colmap;      // [nb][cstride]
    int* m;                 // [nb] (the gates may shrink it)
    const int* nc;          // [nb]
    const double* noise;    // [nb][nstride]
    int r_kind, mld, hstride, cstride, nstride;
    double* Y;              // [nb][ystride]
    int ystride;
    double* dx;             // [B][ldp]
    int* status;            // [B]
    int m_cap, nc_cap;      // LDS sizing
    const double* chi2;     // optional block gate of k_ekf_core (GnssUpdate.cpp:286): device table, chi2[dof]
    int chi2_len, gate_max_rows;   // gate_max_rows > 0: updates with m <= gate_max_rows rows are gated as one block
    // In-frame GNSS update (one pass over P): the columns var_order of the covariance are NOT read from P but from W
    // [nb][wstride], column c of W = column colmap[c] of the MSCKF posterior that k_info_apply has yet to write (k_post_cols);
    // Y is padded with zero columns up to ypad; dx is written in the index space AFTER the frame's marginalisation of
    // [marg_idx[bl], + marg_size) (rows inside it are skipped)
    const double* W;
    size_t wstride;
    int ypad;
    const int* marg_idx;
    int marg_size;
};

void launch_ekf_core(const EkfLaunch& L, hipStream_t st);
// ldy / ystride (0: the covariance's ld / L.ystride): leading dimension and per-filter stride of Y when it lives in another workspace
// marg_idx [nb] (optional, with marg_size): fused marginalisation - where the LDS-blocked variant runs (returns true) the updated
// covariance goes compacted into the other ping-pong half (launch_post_marg must follow); false: not fused, nothing changed
bool launch_downdate(const EkfLaunch& L, int n_cap, hipStream_t st, const double* Yb = nullptr, int ldy = 0, size_t ystride = 0,
                     const int* marg_idx = nullptr, int marg_size = 0);
struct RowsGateIn {         // the staged candidate rows of a batch (read-only; the gate writes its compacted copy into EkfLaunch's H/res/noise/m)
    const double *H, *res, *noise;
    const int *m, *colmap, *nc;
    int hstride, cstride;
};
int launch_rows_gate(const EkfLaunch& L, const RowsGateIn& in, double thr, double* gamma, int* keep, hipStream_t st);
void launch_gamma_multi(CovView cv, int b, int nblk, const double* dbuf, const int* ibuf, const int* desc, const double* noise,
                        double* gamma_out, size_t lds_bytes, hipStream_t st);
void launch_gamma(CovView cv, int b, const double* H, const double* res, const int* colmap, int m, int nc,
                  const double* noise, int r_kind, int mld, double* gamma_out, hipStream_t st);

// kernels_cov.hip
void launch_propagate(CovView cv, int b0, int nb, int n_cap, const double* Phi, const double* G, const double* dt, int k,
                      const int* gnss_idx, const double sigma[4], int enable_gnss, double scb, double srw, hipStream_t st,
                      const double* augR = nullptr, int* status_clear = nullptr,      // optional fused K2 / status reset
                      const double* snap = nullptr, const int* n_snap = nullptr);      // optional: start from the snapshot (when propagate_can_restore and augR)
bool propagate_can_restore(int n_cap);
void launch_augment(CovView cv, int b0, int nb, const double* R, hipStream_t st);
void launch_augment_one(CovView cv, int b, const double* R_host, hipStream_t st);      // one filter, R (host memory) as a kernel argument
void launch_marginalize(CovView cv, int b0, int nb, int n_cap, const int* idx, int size, hipStream_t st, int idx_imm = -1);      // idx == nullptr: nb = 1, index idx_imm
void launch_post_marg(CovView cv, int b0, int nb, const int* idx, int size, hipStream_t st);
void launch_append(CovView cv, int b0, int nb, int size, const double* blk, hipStream_t st);
void launch_copy_ints(int* dst, const int* src, int count, hipStream_t st);
void launch_upload_words(void* dst, const void* src_pinned, size_t bytes, hipStream_t st);      // bytes % 4 == 0; src: hipHostMalloc memory
void launch_snapshot(CovView cv, int n_cap, double* snap, int* n_snap, hipStream_t st);
void launch_restore_strips(CovView cv, int b0, int nb, int n_cap, const double* snap, const int* n_snap, const int* gnss_idx, hipStream_t st);      // filters [b0, b0 + nb)
void launch_restore(CovView cv, int b0, int nb, int n_cap, const double* snap, const int* n_snap, hipStream_t st);
int dbg_read_cov(long long* out, int n);
