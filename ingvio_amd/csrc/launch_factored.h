// launch_factored.h — host-side launch descriptor of kernels_factored.hip.
#pragma once
#include "dev_common.h"

struct FactoredLaunch {
    int stage;            // 0 gate2, 1 gram, 2 info update
    int stereo;
    CovView cv;
    FrameView fv;
    MsckfOpts op;
    int b0, nb, fmax_used;
    double* gamma;
    int* accept;
    int* used;
    double* rec;          // [B][fmax][rec_size] per-feature records (gate -> gram)
    double* Apart;
    int* chunk_used;
    int G, rstride;
    const double* noise;
    double* T;
    double* Pc;
    int ystride;
    double* dx;
    int* m_out;
    int* nc_out;
    int* status;
};

int launch_factored(const FactoredLaunch& L, hipStream_t st);
int factored_rec_size(int cmax);
