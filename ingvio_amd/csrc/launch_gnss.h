// launch_gnss.h — host-side launch descriptor of kernels_gnss.hip.
#pragma once
#include "dev_common.h"

// flat records (doubles), the same layout as include/ingvio_hip.h documents for ingvio_gnss_epoch
enum { GE_SYS = 0, GE_PRN, GE_TOE, GE_TOE_SYS, GE_TOC, GE_A, GE_E, GE_I0, GE_OMG, GE_OMG0, GE_M0, GE_DELTA_N, GE_OMG_DOT, GE_I_DOT,
       GE_CUC, GE_CUS, GE_CRC, GE_CRS, GE_CIC, GE_CIS, GE_AF0, GE_AF1, GE_AF2, GE_TGD, GE_URA, GE_N };
// GLONASS record (GE_SYS == 1), same GE_N doubles: sys, prn, toe (GPS week seconds), -, -, pos[3], vel[3], acc[3] (PZ-90 ECEF), tau_n, gamma; GE_URA
enum { GE_GLO_POS = 5, GE_GLO_VEL = 8, GE_GLO_ACC = 11, GE_GLO_TAUN = 14, GE_GLO_GAMMA = 15 };
enum { GO_TOW = 0, GO_PSR, GO_DOPP, GO_PSR_STD, GO_DOPP_STD, GO_FREQ, GO_N };
// per-filter receiver record
enum { GR_NSAT = 0, GR_DOY, GR_HAVE_ION, GR_ION, GR_PW = GR_ION + 8, GR_VW = GR_PW + 3, GR_CB = GR_VW + 3, GR_FS = GR_CB + 4, GR_YAW,
       GR_RENU, GR_ANCHOR = GR_RENU + 9, GR_IDX_SE23 = GR_ANCHOR + 3, GR_IDX_YOF, GR_IDX_FS, GR_IDX_CB, GR_PSR_AMP = GR_IDX_CB + 4,
       GR_DOPP_AMP, GR_N };
enum { GF_N = 20 };            // per-satellite outputs: res_pos, res_vel, los 3, az, el, ion, tro, usable, then the SatState: pos 3, vel 3, dt, ddt, tgd, ttx
#define GNSS_FRONT_NCW 32      // == GNSS_NCW of capi.hip: column stride of the staged rows

struct GnssFrontLaunch {
    const double* eph;         // [nb][smax][GE_N]
    const double* obs;         // [nb][smax][GO_N]
    const double* rcv;         // [nb][GR_N]
    int smax;
    double* front;             // [nb][64][GF_N]
    double *H, *res, *noise;   // staged candidate rows: H [nb][hstride] column-major ld = mld, res / noise [nb][mld]; H == nullptr: no rows (ingvio_gnss_sat_eval)
    int *m, *nc, *colmap;      // [nb], [nb], [nb][GNSS_FRONT_NCW]
    int mld, hstride;
};

void launch_gnss_front(const GnssFrontLaunch& L, int nb, hipStream_t st);
