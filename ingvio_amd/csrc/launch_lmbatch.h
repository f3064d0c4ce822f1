// launch_lmbatch.h — batched SLAM-landmark update and the dense-H route of large generic updates (kernels_lmbatch.hip)
#pragma once
#include "dev_common.h"

#define LM_MAX 64          // in-state landmarks per filter that the batched update carries (the reference's configs: max_landmark_features <= 50)

// staged landmark inputs of a batch, SoA over the context's filters (indexed by the ABSOLUTE filter b)
struct LmView {
    const double* pose;       // [B][24]  R_i2w 9 | p_i2w 3 | R_cl2i 9 | p_c2i 3 (row-major rotations)
    const int* idx;           // [B][2]   state index of the extended pose (9 columns) and of the camera extrinsics (6)
    const int* n_lm;          // [B]
    const int* lm_idx;        // [B][lmax] state index of the landmark (3 columns)
    const int* anchor_idx;    // [B][lmax] state index of its anchor clone (6 columns)
    const double* pf;         // [B][lmax][3] landmark position in the world frame (AnchoredLandmark::valuePosXyz)
    const double* uv;         // [B][lmax][4] current observation (u0, v0, u1, v1)
    const int* tracked;       // [B][lmax] observed in the current frame
    int lmax;
};

struct LmOpts {
    double R_lr[9], t_lr[3];  // T_cl2cr
    double var;               // noise^2
    double chi2_thr;          // quantile(chi_squared(rows), 0.95)
    int stereo;
};

struct LmBuild {
    CovView cv; LmView lv; LmOpts op;
    int b0, nb;
    double* Hd; size_t hstride; int n_ld, m_cap;        // compact blocks [LM_MAX][100] per filter (in the dense-row buffer of the generic route)
    int* cidx; int n_rows;                              // [nb][LM_MAX][4] column bases; rows of P H^T to write (n32)
    double* X; size_t xstride; int ldx, res_row;        // the Cholesky working matrix: residual into row res_row
    double* gamma; int* accept; int* m_out;             // [nb][LM_MAX], [nb][LM_MAX], [nb]
    double* dx;
};
void launch_lm_build(const LmBuild& L, hipStream_t st);
// fused rows + products + gate for states of up to 256 rows (k_lm_front): S and P H^T for every tracked landmark, the accepted rows
// named by rowmap [nb][m_cap] (what launch_lm_chol reads through).  false: shape not covered, use launch_lm_build.
bool launch_lm_front(const LmBuild& L, int lcap, int* rowmap, hipStream_t st);
void launch_lm_finish(CovView cv, int b0, int nb, const double* Y, size_t ystride, int ldy, int y_row0, int z_row, const int* m, double* dx,
                      const int* status /* [B] absolute: bit 4 = the sweep's fail bit, filter skipped */, hipStream_t st);
void launch_add_noise(double* X, size_t xstride, int ldx, const double* noise, int nstride, int r_kind, const int* m, int m_cap, int nb,
                      hipStream_t st);
int dbg_read_lmbatch(long long* out, int n);
