// kernels_tracks.hip — device-resident track store and the per-frame DELTA hand-over (round 6).
//
// The reference's MapServer adds ONE observation per live feature per camera frame (MapServerManager::collectStereoMeas,
// MapServerManager.cpp:146-217) and erases the observations of the clones that leave the window (KeyframeUpdate::cleanStereoObsAtMargTime,
// KeyframeUpdate.cpp:737-761; SwMargUpdate.cpp:425-446).  ingvio_frame_stage re-sends a filter's whole [F][C][4] measurement array and
// the k IMU transition matrices with every frame: 61 KB per update, of which 4.8 KB are new.  Here the observations of every live
// track stay on the device between frames:
//     store.uv   [B][T][C][4]   measurement of track t at window slot s (FeatureInfo::_stereo_obs as the window sees it)
//     store.mask [B][T]         bit s: track t has an observation at slot s
//     store.pf   [B][T][3]      the track's triangulated point (AnchoredLandmark::valuePosXyz), when the host has sent one
// and a frame travels as a delta (drop window slots, append the new clone's column, release tracks, points that changed) + the list
// of tracks the update uses + the raw IMU samples; three small kernels turn that into the staged arrays the MSCKF kernels read:
//     k_imu_steps      ImuPropagator::stateAndCovTransition (ImuPropagator.cpp:98-162) for the k samples: Phi, G, dt, the clone rotation
//     k_tracks_apply   the delta on the store
//     k_tracks_gather  FrameView arrays (uv, mask, pf, anchor, dof, clone table) of the listed tracks
// gfx950 only.
#include <hip/hip_runtime.h>

#include "launch_tracks.h"

namespace {

struct M3 { double m[9]; };
struct V3 { double v[3]; };

__device__ __forceinline__ M3 m3_zero() { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = 0.0; return r; }
__device__ __forceinline__ M3 m3_eye(double s = 1.0) { M3 r = m3_zero(); r.m[0] = r.m[4] = r.m[8] = s; return r; }
__device__ __forceinline__ M3 operator*(const M3& a, const M3& b)
{
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return r;
}
__device__ __forceinline__ M3 operator*(double s, const M3& a) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = s * a.m[i]; return r; }
__device__ __forceinline__ M3 operator+(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i]; return r; }
__device__ __forceinline__ M3 operator-(const M3& a) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = -a.m[i]; return r; }
__device__ __forceinline__ V3 operator*(const M3& a, const V3& x)
{
    V3 r;
    for (int i = 0; i < 3; ++i) r.v[i] = a.m[3 * i] * x.v[0] + a.m[3 * i + 1] * x.v[1] + a.m[3 * i + 2] * x.v[2];
    return r;
}
__device__ __forceinline__ V3 operator*(double s, const V3& a) { return V3{ { s * a.v[0], s * a.v[1], s * a.v[2] } }; }
__device__ __forceinline__ V3 operator+(const V3& a, const V3& b) { return V3{ { a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2] } }; }
__device__ __forceinline__ V3 operator-(const V3& a, const V3& b) { return V3{ { a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2] } }; }
__device__ __forceinline__ double norm(const V3& a) { return sqrt(a.v[0] * a.v[0] + a.v[1] * a.v[1] + a.v[2] * a.v[2]); }
__device__ __forceinline__ M3 skew(const V3& v)                       // AuxGammaFunc.cpp:28-35
{
    M3 r = m3_zero();
    r.m[1] = -v.v[2]; r.m[2] = v.v[1];
    r.m[3] = v.v[2]; r.m[5] = -v.v[0];
    r.m[6] = -v.v[1]; r.m[7] = v.v[0];
    return r;
}

// sin / cos of the step's rotation angle theta = |w| dt, evaluated ONCE per step: every Gamma_m and both Psi functions of a step share
// it (Gamma_m(-w dt) has the same theta; sin 2 theta, cos 2 theta by the double-angle formulas).  Called per function as the host
// does, a lane spent its time in ~17 sin / cos evaluations per step.
struct SinCos { double theta, s, c; };
__device__ __forceinline__ SinCos sincos_of(double theta) { SinCos r; r.theta = theta; sincos(theta, &r.s, &r.c); return r; }

__device__ M3 gamma_func(const V3& vec, int m, const SinCos& sc)      // AuxGammaFunc.cpp:46-113 (same guards, same formulas); |vec| = sc.theta
{
    const double theta = sc.theta;
    if (fabs(theta) < 1e-06) return m3_eye(m == 3 ? 1.0 / 6.0 : (m == 2 ? 0.5 : 1.0));
    const M3 nx = skew((1.0 / theta) * vec);
    const M3 nx2 = nx * nx;
    const double s = sc.s, c = sc.c;
    double f0, f1, f2;
    if (m == 1) { f0 = 1.0; f1 = (1.0 - c) / theta; f2 = (theta - s) / theta; }
    else if (m == 2) { f0 = 0.5; f1 = (theta - s) / (theta * theta); f2 = (theta * theta + 2.0 * c - 2.0) / (2.0 * theta * theta); }
    else if (m == 3) { const double t3 = theta * theta * theta; f0 = 1.0 / 6.0; f1 = (theta * theta + 2.0 * c - 2.0) / (2.0 * t3); f2 = (t3 - 6.0 * theta + 6.0 * s) / (6.0 * t3); }
    else { f0 = 1.0; f1 = s; f2 = 1.0 - c; }
    return m3_eye(f0) + f1 * nx + f2 * nx2;
}

struct SkewProducts { M3 WA, WAW, WAW2, W2A, W2AW, W2AW2; };
__device__ SkewProducts skew_products(const V3& w, const V3& a)      // AuxGammaFunc.cpp:123-133
{
    SkewProducts p;
    const M3 W = skew(w);
    p.WA = W * skew(a);
    p.WAW = p.WA * W;
    p.WAW2 = p.WAW * W;
    p.W2A = W * p.WA;
    p.W2AW = p.W2A * W;
    p.W2AW2 = p.W2AW * W;
    return p;
}
__device__ M3 psi1_func(const V3& w, const V3& a, double dt, const SinCos& sc)          // AuxGammaFunc.cpp:115-166; sc: theta = |w dt|
{
    if (sc.theta < 1e-08) return m3_zero();
    const M3 M1 = (dt * dt) * (skew(a) * gamma_func((-dt) * w, 2, sc));
    const SkewProducts p = skew_products(w, a);
    const double eta = norm(w), xi = eta * dt, xi2 = xi * xi;
    const double sx = sc.s, cx = sc.c, s2 = 2.0 * sc.s * sc.c, c2 = (sc.c - sc.s) * (sc.c + sc.s);
    const double eta3 = eta * eta * eta, eta4 = eta * eta3, eta5 = eta * eta4, eta6 = eta * eta5;
    const double c1 = (sx - xi * cx) / eta3;
    const double cc2 = (c2 - 4 * cx + 3) / (4 * eta4);
    const double c3 = (4 * sx + s2 - 4 * xi * cx - 2 * xi) / (4 * eta5);
    const double c4 = (xi2 - 2 * xi * sx - 2 * cx + 2) / (2 * eta4);
    const double c5 = (6 * xi - 8 * sx + s2) / (4 * eta5);
    const double c6 = (2 * xi2 - 4 * xi * sx - c2 + 1) / (4 * eta6);
    return M1 * (c1 * p.WA + cc2 * p.WAW + c3 * p.WAW2 + c4 * p.W2A + c5 * p.W2AW + c6 * p.W2AW2);
}
__device__ M3 psi2_func(const V3& w, const V3& a, double dt, const SinCos& sc)          // AuxGammaFunc.cpp:168-225
{
    if (sc.theta < 1e-07) return m3_zero();
    const M3 M1 = (dt * dt * dt) * (skew(a) * gamma_func((-dt) * w, 3, sc));
    const SkewProducts p = skew_products(w, a);
    const double eta = norm(w), xi = eta * dt, xi2 = xi * xi, xi3 = xi * xi2;
    const double sx = sc.s, cx = sc.c, s2 = 2.0 * sc.s * sc.c, c2 = (sc.c - sc.s) * (sc.c + sc.s);
    const double eta3 = eta * eta * eta, eta4 = eta * eta3, eta5 = eta * eta4, eta6 = eta * eta5, eta7 = eta * eta6;
    const double c1 = (xi * sx + 2 * cx - 2) / eta4;
    const double cc2 = (6 * xi - 8 * sx + s2) / (8 * eta5);
    const double c3 = (2 * xi2 + 8 * xi * sx + 16 * cx + c2 - 17) / (8 * eta6);
    const double c4 = (xi3 + 6 * xi - 12 * sx + 6 * xi * cx) / (6 * eta5);
    const double c5 = (6 * xi2 + 16 * cx - c2 - 15) / (8 * eta6);
    const double c6 = (4 * xi3 + 6 * xi - 24 * sx - 3 * s2 + 24 * xi * cx) / (24 * eta7);
    return M1 * (c1 * p.WA + cc2 * p.WAW + c3 * p.WAW2 + c4 * p.W2A + c5 * p.W2AW + c6 * p.W2AW2);
}

__device__ __forceinline__ void put(double* M, int r0, int c0, const M3& B)      // column-major 15 x n, as Eigen::Matrix<double, 15, n>::data()
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[(size_t)(c0 + j) * 15 + r0 + i] = B.m[3 * i + j];
}

}  // namespace

// One wave per filter, lane s = IMU sample s of the frame (ImuPropagator::propagateUntil's loop over stateAndCovTransition, the analytic
// branch), from the nominal state the host holds at the frame's start.  The nominal recursion (R, p, v) is sequential in the samples
// but cheap (three Gamma functions and two matrix-vector products per step), the transition matrices are not (Psi_1, Psi_2: a dozen
// 3 x 3 products and eight sin / cos each): every lane runs the cheap recursion up to ITS sample and forms only its own Phi and G; the
// matrices go through LDS so that the stores are coalesced.  (The first version ran one THREAD per filter over all k samples: 165-200 us
// per 512 filters on the copy stream, under the update's apply kernel.)
// imu [k][7] = gyro (3), accel (3), dt; st0 [24] = R (9, row-major), p, v, bg, ba, gravity.  Writes Phi [k][225], G [k][180], dt [k]
// of the filter's input slot and the IMU rotation at clone time R_out [9] (StateManager::augmentSlidingWindowPose reads it).
__global__ __launch_bounds__(64) void k_imu_steps(TrackStage ts, int b0, int nb, int kcap /* = k of every filter */, double* __restrict__ PhiAll, double* __restrict__ GAll,
                                                  double* __restrict__ dtAll, double* __restrict__ Rall)
{
    extern __shared__ double sm[];                                     // [k][405]: Phi | G of sample s
    const int bl = blockIdx.x, lane = threadIdx.x;
    const int* h = ts.hdr + (size_t)bl * TRK_HDR;
    const int k = h[TRK_K];
    const double* d = ts.dpool + h[TRK_OFF_D];
    const double* imu = d + h[TRK_D_IMU];
    const double* s0 = d + h[TRK_D_STATE];
    M3 R; V3 p, v, bg, ba, g;
    for (int i = 0; i < 9; ++i) R.m[i] = s0[i];
    for (int i = 0; i < 3; ++i) { p.v[i] = s0[9 + i]; v.v[i] = s0[12 + i]; bg.v[i] = s0[15 + i]; ba.v[i] = s0[18 + i]; g.v[i] = s0[21 + i]; }
    const M3 I3 = m3_eye();
    const int mine = lane < k ? lane : k - 1;                          // lanes beyond k shadow the last sample
    // the cheap recursion, the SAME k iterations on every lane (no divergence); a lane keeps the state in front of its own sample
    M3 Rk = R; V3 pk = p, vk = v;
    for (int s = 0; s < k; ++s) {
        if (s == mine) { Rk = R; pk = p; vk = v; }
        const V3 gyro{ { imu[7 * s], imu[7 * s + 1], imu[7 * s + 2] } }, acc{ { imu[7 * s + 3], imu[7 * s + 4], imu[7 * s + 5] } };
        const double dt = imu[7 * s + 6];
        const V3 w = gyro - bg, a = acc - ba;                                           // ImuPropagator.cpp:121-122
        const SinCos sc = sincos_of(norm(dt * w));
        const M3 G0m = gamma_func(dt * w, 0, sc), G1m = gamma_func(dt * w, 1, sc), G2m = gamma_func(dt * w, 2, sc);      // :126-128
        const V3 vn = v + dt * g + dt * ((R * G1m) * a);                               // :133
        const V3 pn = p + dt * v + (0.5 * dt * dt) * g + (dt * dt) * ((R * G2m) * a);    // :136
        R = R * G0m; p = pn; v = vn;
    }
    // every lane's own sample, all lanes at once: Phi and G from the state in front of it
    {
        const int sm_s = mine;
        const V3 gyro{ { imu[7 * sm_s], imu[7 * sm_s + 1], imu[7 * sm_s + 2] } }, acc{ { imu[7 * sm_s + 3], imu[7 * sm_s + 4], imu[7 * sm_s + 5] } };
        const double dt = imu[7 * sm_s + 6];
        const V3 w = gyro - bg, a = acc - ba;
        const SinCos sc = sincos_of(norm(dt * w));
        const M3 RG1 = Rk * gamma_func(dt * w, 1, sc), RG2 = Rk * gamma_func(dt * w, 2, sc);
        const V3 vn = vk + dt * g + dt * (RG1 * a);
        const V3 pn = pk + dt * vk + (0.5 * dt * dt) * g + (dt * dt) * (RG2 * a);
        const M3 P1 = Rk * psi1_func(w, a, dt, sc), P2 = Rk * psi2_func(w, a, dt, sc);
        if (lane < k) {
            double* Phi = sm + (size_t)sm_s * 405;
            double* G = Phi + 225;
            for (int i = 0; i < 405; ++i) Phi[i] = 0.0;
            for (int i = 0; i < 15; ++i) Phi[i * 15 + i] = 1.0;                        // :105
            put(G, 0, 0, Rk);                                                           // :112-117
            put(G, 3, 0, skew(pk) * Rk);
            put(G, 6, 0, skew(vk) * Rk);
            put(G, 6, 3, Rk);
            put(G, 9, 6, I3);
            put(G, 12, 9, I3);
            put(Phi, 3, 0, (0.5 * dt * dt) * skew(g));                                 // :150
            put(Phi, 3, 6, dt * I3);
            put(Phi, 6, 0, dt * skew(g));
            put(Phi, 0, 9, -(dt * RG1));
            put(Phi, 6, 12, -(dt * RG1));
            put(Phi, 3, 12, -((dt * dt) * RG2));
            put(Phi, 6, 9, -(dt * (skew(vn) * RG1)) + P1);                             // :159
            put(Phi, 3, 9, -(dt * (skew(pn) * RG1)) + P2);                             // :161
        }
    }
    __syncthreads();
    double* Phi0 = PhiAll + (size_t)(b0 + bl) * kcap * 225;
    double* G0 = GAll + (size_t)(b0 + bl) * kcap * 180;
    for (int e = lane; e < k * 405; e += 64) {
        const int s = e / 405, r = e - s * 405;
        if (r < 225) Phi0[(size_t)s * 225 + r] = sm[e]; else G0[(size_t)s * 180 + r - 225] = sm[e];
    }
    if (lane < k) dtAll[(size_t)(b0 + bl) * kcap + lane] = imu[7 * lane + 6];
    if (lane == 0) {                                                  // every lane's recursion ended behind the last sample
        double* Ro = Rall + (size_t)(b0 + bl) * 9;
        for (int i = 0; i < 9; ++i) Ro[i] = R.m[i];
    }
}

// One workgroup per filter, one thread per track: the delta on the store.
//   1. window slots that leave (ascending list): the track's mask and its row of measurements close up (a thread moves its own
//      C x 4 doubles; nothing is shared between tracks);
//   2. released tracks: mask cleared;
//   3. the new clone's observations: uv at the append slot, the mask bit;
//   4. points that changed.
__global__ __launch_bounds__(256) void k_tracks_apply(TrackStage ts, TrackStore st, int b0)
{
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x;
    const int* h = ts.hdr + (size_t)bl * TRK_HDR;
    const int* ip = ts.ipool + h[TRK_OFF_I];
    const double* dp = ts.dpool + h[TRK_OFF_D];
    const int n_drop = h[TRK_N_DROP], n_obs = h[TRK_N_OBS], n_free = h[TRK_N_FREE], n_pf = h[TRK_N_PF], app = h[TRK_APPEND];
    const int* drop = ip + h[TRK_I_DROP];
    const int C = st.cmax, T = st.tmax;
    double* uvB = st.uv + (size_t)b * T * C * 4;
    unsigned long long* mkB = st.mask + (size_t)b * T;
    if (n_drop > 0) {
        unsigned long long dm = 0ULL;
        for (int q = 0; q < n_drop; ++q) dm |= 1ULL << drop[q];
        for (int t = tid; t < T; t += 256) {
            const unsigned long long m = mkB[t];
            if (m == 0ULL) continue;
            double* row = uvB + (size_t)t * C * 4;
            unsigned long long nm = 0ULL;
            int w = 0;
            for (int s = 0; s < C; ++s) {
                if ((dm >> s) & 1ULL) continue;
                if ((m >> s) & 1ULL) {
                    nm |= 1ULL << w;
                    if (w != s) { const double4 x = *reinterpret_cast<const double4*>(row + 4 * s); *reinterpret_cast<double4*>(row + 4 * w) = x; }
                }
                ++w;
            }
            mkB[t] = nm;
        }
    }
    __syncthreads();
    const int* fr = ip + h[TRK_I_FREE];
    for (int q = tid; q < n_free; q += 256) mkB[fr[q]] = 0ULL;
    __syncthreads();
    const int* ot = ip + h[TRK_I_OBS];
    const double* ouv = dp + h[TRK_D_OBS];
    for (int q = tid; q < n_obs; q += 256) {
        const int t = ot[q];
        *reinterpret_cast<double4*>(uvB + ((size_t)t * C + app) * 4) = *reinterpret_cast<const double4*>(ouv + 4 * (size_t)q);
        mkB[t] |= 1ULL << app;                                      // one observation per track and frame: no two threads share t
    }
    const int* pt = ip + h[TRK_I_PF];
    const double* pv = dp + h[TRK_D_PF];
    double* pfB = st.pf + (size_t)b * T * 3;
    for (int q = tid; q < n_pf; q += 256) {
        const int t = pt[q];
        pfB[3 * t] = pv[3 * q]; pfB[3 * t + 1] = pv[3 * q + 1]; pfB[3 * t + 2] = pv[3 * q + 2];
    }
}

// One workgroup per filter: the staged frame (FrameView arrays of the context's current input set) of the listed tracks.
// feat word = track | anchor slot << 16 | dof << 24; sel [n_feat] optional u64 masks ANDed onto the stored ones (the
// Selected-timestamp updates, SwMargUpdate.cpp:236-257).
__global__ __launch_bounds__(256) void k_tracks_gather(TrackStage ts, TrackStore st, FrameOut fv, int b0, int* __restrict__ idx_marg, int* __restrict__ gnss_idx)
{
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x;
    const int* h = ts.hdr + (size_t)bl * TRK_HDR;
    const int* ip = ts.ipool + h[TRK_OFF_I];
    const double* dp = ts.dpool + h[TRK_OFF_D];
    const int F = h[TRK_N_FEAT], Cn = h[TRK_N_CLONES], C = st.cmax, T = st.tmax;
    const int* feat = ip + h[TRK_I_FEAT];
    const int* cidx = ip + h[TRK_I_CIDX];
    const double* cR = dp + h[TRK_D_CR];
    const double* cp = dp + h[TRK_D_CP];
    const unsigned long long* sel = h[TRK_HAS_SEL] ? ts.mpool + h[TRK_OFF_M] : nullptr;
    if (tid == 0) {
        fv.n_clones[b] = Cn; fv.n_feat[b] = F;
        idx_marg[b] = h[TRK_MARG];
    }
    if (tid < 5) gnss_idx[(size_t)b * 5 + tid] = ip[h[TRK_I_GNSS] + tid];
    for (int q = tid; q < Cn; q += 256) fv.clone_idx[(size_t)b * fv.cmax + q] = cidx[q];
    for (int q = tid; q < Cn * 9; q += 256) fv.clone_R[(size_t)b * fv.cmax * 9 + q] = cR[q];
    for (int q = tid; q < Cn * 3; q += 256) fv.clone_p[(size_t)b * fv.cmax * 3 + q] = cp[q];
    const double* uvB = st.uv + (size_t)b * T * C * 4;
    const unsigned long long* mkB = st.mask + (size_t)b * T;
    const double* pfB = st.pf + (size_t)b * T * 3;
    for (int j = tid; j < F; j += 256) {
        const unsigned w = (unsigned)feat[j];
        const int t = (int)(w & 0xffffu);
        unsigned long long m = mkB[t];
        if (sel) m &= sel[j];
        const size_t o = (size_t)b * fv.fmax + j;
        fv.anchor[o] = (int)((w >> 16) & 0xffu);
        fv.dof[o] = (int)(w >> 24);
        fv.obs_mask[o] = m;
        fv.pf[3 * o] = pfB[3 * t]; fv.pf[3 * o + 1] = pfB[3 * t + 1]; fv.pf[3 * o + 2] = pfB[3 * t + 2];
    }
    for (int j = F + tid; j < fv.fmax; j += 256) fv.obs_mask[(size_t)b * fv.fmax + j] = 0ULL;      // as pack_frames leaves the rows behind the frame's features
    // measurements: F rows of C x 4 doubles, coalesced by (feature, element) pairs
    const int per = fv.cmax * 4;
    for (int e = tid; e < F * per; e += 256) {
        const int j = e / per, r = e - j * per;
        const int t = (int)((unsigned)feat[j] & 0xffffu);
        fv.uv[((size_t)b * fv.fmax + j) * per + r] = r < C * 4 ? uvB[(size_t)t * C * 4 + r] : 0.0;
    }
}

void launch_imu_steps(const TrackStage& ts, int b0, int nb, int kcap, double* Phi, double* G, double* dt, double* R, hipStream_t st)
{
    hipLaunchKernelGGL(k_imu_steps, dim3(nb), dim3(64), sizeof(double) * 405 * (size_t)kcap, st, ts, b0, nb, kcap, Phi, G, dt, R);
}
void launch_tracks_apply(const TrackStage& ts, const TrackStore& store, int b0, int nb, hipStream_t st)
{
    hipLaunchKernelGGL(k_tracks_apply, dim3(nb), dim3(256), 0, st, ts, store, b0);
}
void launch_tracks_gather(const TrackStage& ts, const TrackStore& store, const FrameOut& fv, int b0, int nb, int* idx_marg, int* gnss_idx, hipStream_t st)
{
    hipLaunchKernelGGL(k_tracks_gather, dim3(nb), dim3(256), 0, st, ts, store, fv, b0, idx_marg, gnss_idx);
}
