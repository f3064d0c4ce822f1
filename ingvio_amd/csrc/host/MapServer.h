// MapServer.h — mirrors the data model of ingvio_estimator/src/MapServer.h:30-134 (MonoMeas, StereoMeas,
// FeatureInfo, MapServer).  The pointer-chasing bookkeeping of MapServerManager stays host-side and out
// of scope; the Update classes flatten this structure into the SoA of the C ABI.
// Licence note: class, member and accessor names below are those of the reference (InGVIO, (C) 2022 Changwu Liu, GNU GPL v3 or later)
// because the drop-in contract is source compatibility with code written against them; distributed under the same licence.
#pragma once
#include <map>
#include <memory>

#include "AnchoredLandmark.h"

namespace ingvio {

struct MonoMeas { int _id = -1; double _u0 = 0, _v0 = 0; };
struct StereoMeas { int _id = -1; double _u0 = 0, _v0 = 0, _u1 = 0, _v1 = 0; };

class FeatureInfo {
public:
    enum FeatureType { MSCKF = 0, SLAM };
    FeatureInfo() : _id(-1), _ftype(MSCKF), _isToMarg(false), _isTri(false), _numOfTri(0) { _landmark = std::make_shared<AnchoredLandmark>(); }
    // the read accessors of MapServer.h:84-113 (the reference's tests and update classes use both these and the members)
    int getId() const { return _id; }
    FeatureType getFeatureType() const { return _ftype; }
    bool isToMarg() const { return _isToMarg; }
    bool isTri() const { return _isTri; }
    bool hasMonoObsAt(double t) const { return _mono_obs.find(t) != _mono_obs.end(); }
    bool hasStereoObsAt(double t) const { return _stereo_obs.find(t) != _stereo_obs.end(); }
    const std::shared_ptr<AnchoredLandmark> landmark() const { return _landmark; }
    int numOfMonoFrames() const { return (int)_mono_obs.size(); }
    int numOfStereoFrames() const { return (int)_stereo_obs.size(); }
    const std::shared_ptr<SE3> anchor() const { return _landmark->getAnchoredPose(); }
    int _id;
    FeatureType _ftype;
    bool _isToMarg;
    bool _isTri;
    int _numOfTri;
    std::shared_ptr<AnchoredLandmark> _landmark;
    std::map<double, std::shared_ptr<MonoMeas>> _mono_obs;
    std::map<double, std::shared_ptr<StereoMeas>> _stereo_obs;
};

typedef std::map<int, std::shared_ptr<FeatureInfo>> MapServer;

}  // namespace ingvio
