// Update.h — mirrors ingvio_estimator/src/Update.h:36-96 (UpdateBase): the chi-squared table
// (Boost quantile replaced by a self-contained inverse regularised gamma, equal to
// boost::math::quantile(chi_squared(k), p) to 1e-12) and the three testChiSquared overloads; the
// Mahalanobis arithmetic of whitenResidual runs on the device against the resident covariance.
#pragma once
#include <map>
#include <memory>
#include <vector>

#include "MatX.h"

namespace ingvio {

class Type;
class State;

double chi2Quantile(int dof, double p);      // == boost::math::quantile(chi_squared(dof), p)

class UpdateBase {
public:
    UpdateBase(const int& max_dof, const double& thres) : _thres(thres) { this->setChiSquaredTable(max_dof, _thres); }
    UpdateBase(const double& thres) : _thres(thres) { this->setChiSquaredTable(150, _thres); }
    UpdateBase() : _thres(0.95) { this->setChiSquaredTable(150, _thres); }
    virtual ~UpdateBase() {}
    UpdateBase(const UpdateBase&) = delete;

    // dense copy of the table for the C ABI: t[d] = quantile(d), t[0] unused
    std::vector<double> chi2TableDense(int min_len = 0);

protected:
    double _thres;
    std::map<int, double> _chi_squared_table;
    void setChiSquaredTable(const int& max_dof, const double& thres);                               // Update.cpp:27-34
    void extendTable(int dof);                                                                       // :91-96
    double whitenResidual(const std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                          const std::vector<std::shared_ptr<Type>>& var_order, double noise);        // :36-56
    double whitenResidual(const std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                          const std::vector<std::shared_ptr<Type>>& var_order, const MatXd& R);      // :58-79
    virtual bool testChiSquared(const std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                const std::vector<std::shared_ptr<Type>>& var_order, double noise);  // :81-102
    virtual bool testChiSquared(const std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                const std::vector<std::shared_ptr<Type>>& var_order, double noise, int dof);   // :104-124
    virtual bool testChiSquared(const std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                const std::vector<std::shared_ptr<Type>>& var_order, const MatXd& R, int dof); // :126-149
};

}  // namespace ingvio
