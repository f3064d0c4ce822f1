// ingvio_node.cpp — ROS1 node of the MI355X-backed estimator: the drop-in for ingvio_estimator/src/IngvioNode.cpp:25-39 +
// IngvioFilter::initIO (IngvioFilter.cpp:50-122) + the ROS half of the callbacks (:124-498) and of GnssProcessor.cpp:32-220.
// Same node name, parameter ("~config_file": the reference's OpenCV YAML), topics (feature_topic / imu_topic / gnss_*_topic of
// that YAML) and outputs (pose_w, path_w, pose_spp, path_spp, tf world -> uav).  Everything numerical is the host shim
// (ingvio_amd/csrc/host, libingvio_host.so) over libingvio_hip.so; this file only moves fields (ros1/include/RosAdapter.h).
// Build: catkin, see ros1/CMakeLists.txt.  ROS is not part of the build image of this repository: the file is compiled where
// ROS1 + feature_tracker + gnss_comm message packages are installed; its conversion core is unit-tested without ROS
// (tests/cpp/test_ros_adapter.cpp).
#include <fstream>
#include <memory>
#include <sstream>

#include <ros/ros.h>
#include <sensor_msgs/Imu.h>
#include <nav_msgs/Odometry.h>
#include <nav_msgs/Path.h>
#include <geometry_msgs/PoseStamped.h>
#include <tf/transform_broadcaster.h>

#include <feature_tracker/MonoFrame.h>
#include <feature_tracker/StereoFrame.h>
#include <gnss_comm/GnssEphemMsg.h>
#include <gnss_comm/GnssGloEphemMsg.h>
#include <gnss_comm/GnssMeasMsg.h>
#include <gnss_comm/StampedFloat64Array.h>

#include "IngvioFilter.h"
#include "Replay.h"
#include "RosAdapter.h"
#include "StateManager.h"

namespace ingvio {

// The reference's YAML is OpenCV FileStorage text; the scalar keys are "key: value" lines, which applyParamsText (Replay.h) reads
// with the same key names (IngvioParams.cpp:27-174).  The camera files carry T_cam_imu as an !!opencv-matrix: its 16 numbers.
static bool readOpencvMatrix(const std::string& path, const std::string& key, double out[16])
{
    std::ifstream f(path);
    if (!f) return false;
    std::stringstream ss; ss << f.rdbuf();
    const std::string s = ss.str();
    size_t p = s.find(key + ":");
    if (p == std::string::npos) return false;
    p = s.find("data:", p);
    if (p == std::string::npos) return false;
    p = s.find('[', p);
    const size_t q = s.find(']', p);
    if (p == std::string::npos || q == std::string::npos) return false;
    std::string body = s.substr(p + 1, q - p - 1);
    for (char& c : body) if (c == ',') c = ' ';
    std::istringstream is(body);
    for (int i = 0; i < 16; ++i) if (!(is >> out[i])) return false;
    return true;
}

static std::string yamlString(const std::string& text, const std::string& key)
{
    size_t p = text.find("\n" + key + ":");
    if (p == std::string::npos) return "";
    p = text.find(':', p) + 1;
    size_t q = text.find('\n', p);
    std::string v = text.substr(p, q - p);
    while (!v.empty() && (v.front() == ' ' || v.front() == '"')) v.erase(v.begin());
    while (!v.empty() && (v.back() == ' ' || v.back() == '"' || v.back() == '\r')) v.pop_back();
    return v;
}

class IngvioRosNode {
public:
    explicit IngvioRosNode(ros::NodeHandle& n) : _nh(n) {}

    bool initIO()                                                                       // IngvioFilter.cpp:50-122
    {
        std::string config_file;
        _nh.param<std::string>("config_file", config_file, "");
        std::ifstream f(config_file);
        if (!f) { ROS_ERROR("[IngvioParams]: cannot open config_file %s", config_file.c_str()); return false; }
        std::stringstream ss; ss << f.rdbuf();
        const std::string text = "\n" + ss.str();
        applyParamsText(text, _params);
        // camera extrinsics (IngvioParams.cpp:126-174): T_cam_imu of the camera files is T_imu2cam; the filter keeps T_cam2imu
        double M[16];
        const std::string left = yamlString(text, "cam_left_file_path"), right = yamlString(text, "cam_right_file_path");
        auto toIso = [&](const double* m) { Iso3 T; for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T.R(r, c) = m[4 * r + c]; T.t[r] = m[4 * r + 3]; } return T.inverse(); };
        if (!left.empty() && readOpencvMatrix(left, "T_cam_imu", M)) _params._T_cl2i = toIso(M);
        if (_params._cam_nums == 2 && !right.empty() && readOpencvMatrix(right, "T_cam_imu", M)) _params._T_cr2i = toIso(M);
        const std::string feature_topic = yamlString(text, "feature_topic"), imu_topic = yamlString(text, "imu_topic");
        // device capacity from the window (the shim's State creates the libingvio_hip context): 21 + 6 GNSS scalars + clones + landmarks
        _params._hip_n_max = std::max(_params._hip_n_max, 21 + 6 + 6 * (_params._max_sw_clones + 2) + 3 * _params._max_lm_feats + 16);
        _filter.reset(new IngvioFilter(_params, std::make_shared<Triangulator>(_params)));

        if (_params._cam_nums == 2) _sub_frame = _nh.subscribe(feature_topic, 100, &IngvioRosNode::callbackStereoFrame, this);
        else {
            if (_params._cam_nums != 1) std::cout << "[IngvioParams]: Cam num " << _params._cam_nums << " not supported! Init as mono config!" << std::endl;
            _sub_frame = _nh.subscribe(feature_topic, 100, &IngvioRosNode::callbackMonoFrame, this);
        }
        _sub_imu = _nh.subscribe(imu_topic, 500, &IngvioRosNode::callbackIMU, this);
        _odom_w_pub = _nh.advertise<nav_msgs::Odometry>("pose_w", 5);
        _path_w_pub = _nh.advertise<nav_msgs::Path>("path_w", 1);
        if (_params._enable_gnss) {
            _sub_ephem = _nh.subscribe(yamlString(text, "gnss_ephem_topic"), 100, &IngvioRosNode::callbackEphem, this);
            _sub_glo_ephem = _nh.subscribe(yamlString(text, "gnss_glo_ephem_topic"), 100, &IngvioRosNode::callbackGloEphem, this);
            _sub_gnss_meas = _nh.subscribe(yamlString(text, "gnss_meas_topic"), 100, &IngvioRosNode::callbackGnssMeas, this);
            _sub_iono = _nh.subscribe(yamlString(text, "gnss_iono_params_topic"), 100, &IngvioRosNode::callbackIonoParams, this);
            _odom_spp_pub = _nh.advertise<nav_msgs::Odometry>("pose_spp", 5);
            _path_spp_pub = _nh.advertise<nav_msgs::Path>("path_spp", 1);
            std::istringstream a(yamlString(text, "gnss_psr_std_thres")), b(yamlString(text, "gnss_dopp_std_thres")), c(yamlString(text, "gnss_track_num_thres"));
            a >> _gnss.psr_std_thres; b >> _gnss.dopp_std_thres; c >> _gnss.track_num_thres;
        }
        return true;
    }

private:
    void callbackIMU(const sensor_msgs::ImuConstPtr& m)                                                            // :381-407
    {
        const double aux_time = ros::Time::now().toSec();
        if (_params._enable_gnss) {                                                                               // :385-391
            _filter->gnssSync()->storeTimePairHeader(aux_time, m->header.stamp.toSec());
            if (!_filter->gnssSync()->isSync()) return;
        }
        _filter->callbackIMU(ros1::imuFromRos(*m));
    }

    void callbackMonoFrame(const feature_tracker::MonoFrameConstPtr& m)                                            // :124-250
    {
        _filter->callbackMonoFrame(ros1::monoFrameFromRos(*m));
        visualize(ros1::headerFromRos(m->header));
    }
    void callbackStereoFrame(const feature_tracker::StereoFrameConstPtr& m)                                        // :252-379
    {
        _filter->callbackStereoFrame(ros1::stereoFrameFromRos(*m));
        visualize(ros1::headerFromRos(m->header));
    }

    void visualize(const msg::Header& header)                                                                      // :409-447
    {
        msg::Odometry od;
        if (!_filter->odometry(header, od)) return;
        nav_msgs::Odometry odom;
        ros1::odometryToRos(od, odom);
        tf::Transform T;
        T.setOrigin(tf::Vector3(od.position.x, od.position.y, od.position.z));
        T.setRotation(tf::Quaternion(od.orientation.x, od.orientation.y, od.orientation.z, od.orientation.w));
        _tf_pub.sendTransform(tf::StampedTransform(T, odom.header.stamp, "world", "uav"));
        _odom_w_pub.publish(odom);
        geometry_msgs::PoseStamped ps;
        ros1::poseStampedFromOdometry(odom, ps);
        _path_w.header = ps.header;
        _path_w.poses.push_back(ps);
        _path_w_pub.publish(_path_w);
    }

    // ---- GNSS (GnssProcessor.cpp:32-220) ---------------------------------------------------------------------------------------
    void callbackEphem(const gnss_comm::GnssEphemMsgConstPtr& m) { _gnss.addEphem(m->sat, ros1::ephemFromRos(*m)); }            // :32-66
    void callbackGloEphem(const gnss_comm::GnssGloEphemMsgConstPtr& m) { _gnss.addEphem(m->sat, ros1::gloEphemFromRos(*m)); }    // :68-100
    void callbackIonoParams(const gnss_comm::StampedFloat64ArrayConstPtr& m) { _gnss.setIono(*m); }                              // :102-117

    void callbackGnssMeas(const gnss_comm::GnssMeasMsgConstPtr& m)                                                                // :119-220
    {
        const double now = ros::Time::now().toSec();
        if (!_params._enable_gnss || _gnss.iono.size() != 8 || m->meas.empty()) return;
        const double t_gnss = ros1::gpstAbs(m->meas[0].time.week, m->meas[0].time.tow);
        // GnssSync::storeTimePair (GnssSync.cpp:66-99, called from GnssProcessor.cpp:130): arrival times of GNSS epochs and of sensor
        // headers are paired by the shim's GnssSync; nothing is buffered before the offset GNSS time -> header clock is known
        _filter->gnssSync()->storeTimePairGnss(now, t_gnss);
        if (!_filter->gnssSync()->isSync()) return;
        const double gnss2local = _filter->gnssSync()->getUnsyncTime();
        // day of year for the troposphere model (gnss_comm::time2doy): GPS epoch 1980-01-06 = day 6
        const double days = t_gnss / 86400.0 + 5.0;
        const double doy = std::fmod(days, 365.25) + 1.0;
        RawGnssEpoch raw;
        if (_gnss.epochFromRos(*m, doy, raw) <= 0) return;
        const double stamp = t_gnss + gnss2local;
        // satellite states + atmosphere at the SPP position of this epoch (psr_pos), the SPP fix itself for the buffer
        auto aligner = _filter->gvioAligner();
        aligner->setIono(_gnss.iono);
        double xyzt[7], vel[4];
        std::vector<const RawGnssEpoch*> one{ &raw };
        const bool have_spp = raw.n_sat() >= 4 && aligner->psrPos(one, xyzt) && aligner->doppVel(raw, xyzt, vel);
        ingvio_gnss_epoch e;
        std::memset(&e, 0, sizeof e);
        e.n_sat = raw.n_sat(); e.eph = raw.eph.data(); e.obs = raw.obs.data(); e.ion = _gnss.iono.data(); e.doy = doy;
        e.R_enu2ecef[0] = e.R_enu2ecef[4] = e.R_enu2ecef[8] = 1.0;
        // Where the epoch is evaluated (elevation, Klobuchar / Saastamoinen delays ride on the buffered record; the reference's psr_res
        // computes them at the state's ECEF position, GnssUpdate.cpp:98-122): the SPP fix of this epoch; without one (fewer than four
        // satellites, no convergence) the filter's own ECEF position once aligned, else the last point used.  An epoch with none of
        // the three is dropped: a receiver "at the geocentre" has no atmosphere and 90 degree elevations.
        if (have_spp) { std::memcpy(e.anchor_ecef, xyzt, 24); std::memcpy(e.cb, xyzt + 3, 32); }
        else if (aligner->isAlign()) {
            const Vec3d p_ecef = aligner->getTecef2w().inverse() * _filter->state()->_extended_pose->valueTrans1();
            for (int i = 0; i < 3; ++i) e.anchor_ecef[i] = p_ecef[i];
        }
        else if (_have_anchor) std::memcpy(e.anchor_ecef, _last_anchor, 24);
        else return;
        std::memcpy(_last_anchor, e.anchor_ecef, 24); _have_anchor = true;
        std::vector<double> rec((size_t)INGVIO_GNSS_MAX_SAT * INGVIO_GNSS_SAT_REC);
        if (ingvio_gnss_sat_eval(StateManager::ctx(_filter->state()), 1, &e, rec.data()) != INGVIO_OK) return;
        _filter->callbackGnssMeas(ros1::gnssMeasFromEval(stamp, raw, _gnss.iono, rec.data()));
        if (have_spp) {
            SppMeas s;
            s.stamp = stamp;
            for (int i = 0; i < 7; ++i) s.posSpp[i] = xyzt[i];
            for (int i = 0; i < 4; ++i) s.velSpp[i] = vel[i];
            _filter->callbackSppMeas(s);
            visualizeSpp(stamp, s);
        }
    }

    void visualizeSpp(double stamp, const SppMeas& s)                                                              // :449-498
    {
        auto al = _filter->gvioAligner();
        if (!al->isAlign()) return;
        const Vec3d p = al->getTecef2w() * Vec3d(s.posSpp), v = al->getRecef2enu() * Vec3d(s.velSpp);
        if (p[0] != p[0] || v[0] != v[0]) return;
        nav_msgs::Odometry od;
        od.header.stamp = ros::Time(stamp); od.header.frame_id = "world"; od.child_frame_id = "spp";
        od.pose.pose.position.x = p[0]; od.pose.pose.position.y = p[1]; od.pose.pose.position.z = p[2];
        od.pose.pose.orientation.w = 1.0;
        od.twist.twist.linear.x = v[0]; od.twist.twist.linear.y = v[1]; od.twist.twist.linear.z = v[2];
        geometry_msgs::PoseStamped ps;
        ros1::poseStampedFromOdometry(od, ps);
        _path_spp.header = ps.header; _path_spp.poses.push_back(ps);
        _path_spp_pub.publish(_path_spp);
        _odom_spp_pub.publish(od);
    }

    ros::NodeHandle _nh;
    IngvioParams _params;
    std::unique_ptr<IngvioFilter> _filter;
    ros1::GnssFrontEnd _gnss;
    double _last_anchor[3] = { 0, 0, 0 };            // ECEF point the last epoch's satellite geometry / atmosphere was evaluated at
    bool _have_anchor = false;
    ros::Subscriber _sub_frame, _sub_imu, _sub_ephem, _sub_glo_ephem, _sub_gnss_meas, _sub_iono;
    ros::Publisher _odom_w_pub, _path_w_pub, _odom_spp_pub, _path_spp_pub;
    nav_msgs::Path _path_w, _path_spp;
    tf::TransformBroadcaster _tf_pub;
};

}  // namespace ingvio

int main(int argc, char** argv)                                                        // IngvioNode.cpp:25-39
{
    ros::init(argc, argv, "ingvio_estimator");
    ros::NodeHandle n("~");
    ros::console::set_logger_level(ROSCONSOLE_DEFAULT_NAME, ros::console::levels::Info);
    ingvio::IngvioRosNode node(n);
    if (!node.initIO()) return 1;
    ros::spin();
    return 0;
}
