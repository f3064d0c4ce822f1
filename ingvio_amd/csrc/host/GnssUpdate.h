// GnssUpdate.h — the covariance slice of ingvio_estimator/src/GnssUpdate.cpp:84-293
// (GnssUpdate::updateTrackedSys): H assembly for pseudo-range / Doppler rows, the per-row and block
// chi^2 gates and ekfUpdate with a diagonal R.  Its inputs are the OUTPUTS of gnss_comm's
// psr_res / dopp_res / sat_states (line-of-sight units, residuals, elevation, ura/std), which stay
// out of scope.
#pragma once
#include <memory>
#include <vector>

#include <cmath>
#include <iomanip>
#include <iostream>
#include <map>
#include <queue>
#include <unordered_set>

#include "GnssComm.h"
#include "IngvioParams.h"
#include "Mat3.h"
#include "Update.h"

namespace ingvio {

class IngvioParams;
class State;

struct GnssResiduals {
    // one entry per satellite, in gnss_meas order
    std::vector<Vec3d> unit_rv2sv;        // = -J_pos_ecef.row(i).head<3>()  (GnssUpdate.cpp:150)
    std::vector<int> sys;                 // gnss_comm::sys2idx: GPS 0, GLO 1, GAL 2, BDS 3
    std::vector<double> res_pos, res_vel; // psr_res / dopp_res outputs
    std::vector<double> sin_el;           // sin(all_sv_azel[i].y())
    std::vector<double> ura, psr_std;     // ephem ura, obs psr_std[l1]
    std::vector<double> dopp_std_mps;     // dopp_std[l1] * LIGHT_SPEED / L1 frequency
    Mat3d R_w2ecef;                       // getRenu2ecef() * calcRw2enu(state)
    // is_adjust_yof (GnssUpdate.cpp:164-167, 239-242): the yaw-offset column of a row is -u^T Renu2ecef dotRw2enu(state) p (v for
    // Doppler rows).  dRw2ecef_dyof = getRenu2ecef() * GnssManager::dotRw2enu(state); used only when has_yof_jac is set.
    Mat3d R_enu2ecef = Mat3d::Identity();
    Mat3d dRw2ecef_dyof;
    bool has_yof_jac = false;
};

// GnssManager::dotRw2enu (GnssManager.cpp:101-113): d/d(yaw offset) of calcRw2enu = AngleAxis(yo, UnitZ); nine entries from yo.
Mat3d dotRw2enu(double yaw_offset);

// candidate rows of updateTrackedSys (all pseudo-range rows, then all Doppler rows; H column-major, ldh >= 2 nsat, 15 columns;
// vidx / vsize hold up to 7 variables); returns the row count
int gnssCandidateRows(const GnssResiduals& g, const Vec3d& p_w, const Vec3d& v_w, int idx_se23, int idx_yof, const int idx_cb[4],
                      int idx_fs, double psr_amp, double dopp_amp, double* H, int ldh, double* res, double* Rd, int* vidx, int* vsize,
                      int* nvar, bool adjust_yof = false);

// One GNSS epoch as GnssProcessor::callbackGnssMeas hands it on (GnssProcessor.cpp:119-220: valid L1 observations with their
// ephemerides; here with the satellite states already evaluated) and one SPP fix (GnssSync.h SppMeas).
struct GnssMeas {
    double stamp = 0;
    std::vector<gnss::SatObs> sats;
    // the raw content of the epoch (GnssData.h GnssMeas = (obs, ephems) + latest_gnss_iono_params), flat records in the layout of
    // ingvio_gnss_epoch: only GvioAligner::batchAlign reads it (the alignment needs the ephemerides, not satellite states at
    // one receiver position); empty = the alignment must be provided with IngvioFilter::setGnssAlignment
    std::vector<double> raw_eph, raw_obs, iono;
    double doy = 0.0;
};
struct SppMeas { double stamp = 0; double posSpp[7] = { 0 }; double velSpp[4] = { 0 }; };      // (ecef xyz, clock biases) / (ecef vel, drift)

// What GnssUpdate reads from GvioAligner (GvioAligner.h: isAlign, getYawOffset, getRenu2ecef, getTenu2ecef); the batch
// alignment itself (GvioAligner.cpp:199-383) is out of scope: the result is set by the caller.
struct GvioAlignment {
    bool aligned = false;
    double yaw_offset = 0.0;
    Mat3d R_enu2ecef = Mat3d::Identity();
    Vec3d anchor_ecef;
    bool isAlign() const { return aligned; }
};

// GnssSync (GnssSync.cpp:25-236): bounded FIFO buffers and the pick-by-frame-time rule of getGnssMeasAt / getSppAt.  The
// ROS-time <-> GNSS-time offset estimation (storeTimePair) is plumbing: stamps are local times here, setSync() marks the
// buffers usable.
class GnssSync {
public:
    // the reference's threshold is 0.13 s (GnssSync.h:61): it gates the arrival pairing AND the window in which getGnssMeasAt /
    // getSppAt match an epoch to a frame - epochs older than (frame - 0.13 s) are dropped, the first one before (frame + 0.13 s) taken
    explicit GnssSync(double unsync_thres = 0.13) : _unsync_thres(unsync_thres) {}
    // GnssSync.h:60-74: nothing without enable_gnss; with use_fix_time_offset the configured offset is taken and the buffers open at once
    explicit GnssSync(const IngvioParams& fp, double unsync_thres = 0.13) : _unsync_thres(unsync_thres)
    {
        if (!fp._enable_gnss) return;
        if (fp._use_fix_time_offset) { _gnss2local_time_offset = fp._gnss_local_offset; _isSync = true; }
    }
    bool isSync() const { return _isSync; }
    void setSync(bool s = true) { _isSync = s; }
    double getUnsyncTime() const { return _gnss2local_time_offset; }                                                              // GnssSync.h:91-92
    // GnssSync::storeTimePair (GnssSync.cpp:66-134): both overloads record (arrival time on the local clock -> stamp); once three
    // GNSS epochs and three sensor headers are held, the pair of arrivals closest to each other defines the offset that maps GNSS
    // time onto the sensor HEADER clock, corrected by the arrival difference; accepted only if the two arrived within
    // _unsync_thres of each other.  Either way both tables are cleared and the collection starts over.
    void storeTimePairGnss(double curr_time, double gnss_time_sec)
    {
        if (_isSync) return;
        if (_arrival_gnss.find(curr_time) == _arrival_gnss.end()) _arrival_gnss[curr_time] = gnss_time_sec;
        tryPair();
    }
    void storeTimePairHeader(double curr_time, double header_time)
    {
        if (_isSync) return;
        if (_arrival_header.find(curr_time) == _arrival_header.end()) _arrival_header[curr_time] = header_time;
        tryPair();
    }
    void bufferGnssMeas(const GnssMeas& m) { if (!_isSync) return; while (_gnss.size() > 100) _gnss.pop(); _gnss.push(m); }      // :27-46
    void bufferSppMeas(const SppMeas& m) { if (!_isSync) return; while (_spp.size() > 100) _spp.pop(); _spp.push(m); }           // :48-68
    bool getGnssMeasAt(double target_time, GnssMeas& out) { return pick(_gnss, target_time, out); }                                // :136-164
    bool getSppAt(double target_time, SppMeas& out) { return pick(_spp, target_time, out); }                                        // :166-194
private:
    void tryPair()
    {
        if (_arrival_gnss.size() < 3 || _arrival_header.size() < 3) return;
        double delta_time = INFINITY, idx_g = 0, idx_h = 0;
        for (const auto& g : _arrival_gnss)
            for (const auto& h : _arrival_header)
                if (std::fabs(g.first - h.first) < delta_time) { idx_g = g.first; idx_h = h.first; delta_time = std::fabs(idx_g - idx_h); }
        if (delta_time < _unsync_thres) {
            _gnss2local_time_offset = _arrival_header.at(idx_h) - _arrival_gnss.at(idx_g) - (idx_h - idx_g);
            _isSync = true;
            std::cout << "[GnssSync]: gnss time to local time offset = " << std::setprecision(10) << _gnss2local_time_offset << " (s) " << std::endl;
        }
        _arrival_gnss.clear();
        _arrival_header.clear();
    }
    template <class T> bool pick(std::queue<T>& q, double target_time, T& out)
    {
        if (!_isSync) return false;
        while (!q.empty()) {
            const double g_time = q.front().stamp;
            if (g_time < target_time - _unsync_thres) { q.pop(); continue; }
            if (g_time >= target_time + _unsync_thres) break;
            out = q.front(); q.pop();
            return true;
        }
        return false;
    }
    double _unsync_thres;
    bool _isSync = false;
    double _gnss2local_time_offset = 0.0;
    std::map<double, double> _arrival_gnss, _arrival_header;      // arrival (local clock) -> GNSS time (s) / sensor header stamp (s)
    std::queue<GnssMeas> _gnss;
    std::queue<SppMeas> _spp;
};

class GnssUpdate : public UpdateBase {
public:
    GnssUpdate(const IngvioParams& filter_params);
    // returns rows handed to ekfUpdate (0: nothing done)
    int updateTrackedSys(std::shared_ptr<State> state, const GnssResiduals& g);
    // GnssUpdate.cpp:33-44, :84-293, :319-469 on a GNSS epoch: residuals by the restated psr_res / dopp_res (GnssComm.h), then the
    // row assembly, gates and update on the device
    void checkYofStatus(std::shared_ptr<State> state, const GvioAlignment& aligner);
    int updateTrackedSys(std::shared_ptr<State> state, const GnssMeas& gnss_meas, const GvioAlignment& aligner);
    int addNewTrackedSys(std::shared_ptr<State> state, const GnssMeas& gnss_meas, const SppMeas& spp_meas, const GvioAlignment& aligner);
    void removeUntrackedSys(std::shared_ptr<State> state, const GnssMeas& gnss_meas);                    // :65-82
    // psr_res + dopp_res at the current state (:98-122; `xyzt` / `dopp` may carry SPP values for systems about to be added)
    GnssResiduals residualsAt(std::shared_ptr<State> state, const GnssMeas& gnss_meas, const GvioAlignment& aligner,
                              const double* cb_override = nullptr, const double* fs_override = nullptr);

    const std::vector<int>& lastKeep() const { return _last_keep; }
    void clearLastKeep() { _last_keep.clear(); }      // candidate rows of the last updateTrackedSys that passed their per-row gate

protected:
    std::vector<int> _last_keep;
    double _psr_noise_amp, _dopp_noise_amp;
    bool _is_gnss_chi2_test, _is_gnss_strong_reject, _is_adjust_yof;
};

}  // namespace ingvio
