#include "Update.h"

#include <cmath>

#include "StateManager.h"

namespace ingvio {

namespace {
double gammaincLowerReg(double a, double x)      // P(a, x): series for x < a+1, Lentz continued fraction otherwise
{
    if (x <= 0.0) return 0.0;
    if (x < a + 1.0) {
        double term = 1.0 / a, s = term, n = a;
        for (int i = 0; i < 10000; ++i) { n += 1.0; term *= x / n; s += term; if (std::fabs(term) < std::fabs(s) * 1e-17) break; }
        return s * std::exp(-x + a * std::log(x) - std::lgamma(a));
    }
    const double tiny = 1e-300;
    double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
    for (int i = 1; i < 10000; ++i) {
        const double an = -i * (i - a);
        b += 2.0;
        d = an * d + b; if (std::fabs(d) < tiny) d = tiny;
        c = b + an / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double delta = d * c;
        h *= delta;
        if (std::fabs(delta - 1.0) < 1e-16) break;
    }
    return 1.0 - std::exp(-x + a * std::log(x) - std::lgamma(a)) * h;
}
double normPpf(double p)      // Acklam + one Newton step
{
    static const double a[6] = { -3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                                 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00 };
    static const double b[5] = { -5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                                 6.680131188771972e+01, -1.328068155288572e+01 };
    static const double c[6] = { -7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                                 -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00 };
    static const double d[4] = { 7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00 };
    double x;
    if (p < 0.02425) { const double q = std::sqrt(-2 * std::log(p)); x = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1); }
    else if (p > 1 - 0.02425) { const double q = std::sqrt(-2 * std::log(1 - p)); x = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1); }
    else { const double q = p - 0.5, r = q * q; x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1); }
    const double e = 0.5 * std::erfc(-x / std::sqrt(2.0)) - p;
    x -= e * std::sqrt(2.0 * M_PI) * std::exp(0.5 * x * x);
    return x;
}
}  // namespace

double chi2Quantile(int k, double p)
{
    const double a = 0.5 * k, z = normPpf(p);
    double x = k * std::pow(1.0 - 2.0 / (9.0 * k) + z * std::sqrt(2.0 / (9.0 * k)), 3);      // Wilson-Hilferty start
    if (!(x > 1e-8)) x = 1e-8;
    for (int it = 0; it < 100; ++it) {
        const double f = gammaincLowerReg(a, 0.5 * x) - p;
        const double pdf = std::exp((a - 1.0) * std::log(0.5 * x) - 0.5 * x - std::lgamma(a)) * 0.5;
        double xn = x - f / pdf;
        if (xn <= 0) xn = 0.5 * x;
        if (std::fabs(xn - x) < 1e-14 * (x > 1.0 ? x : 1.0)) { x = xn; break; }
        x = xn;
    }
    return x;
}

void UpdateBase::setChiSquaredTable(const int& max_dof, const double& thres)
{
    for (int i = 1; i <= max_dof; ++i) this->_chi_squared_table[i] = chi2Quantile(i, thres);
}

void UpdateBase::extendTable(int dof)
{
    if (_chi_squared_table.find(dof) == _chi_squared_table.end())
        for (int i = _chi_squared_table.rbegin()->first + 1; i <= dof; ++i) this->_chi_squared_table[i] = chi2Quantile(i, _thres);
}

std::vector<double> UpdateBase::chi2TableDense(int min_len)
{
    if (min_len > 1) extendTable(min_len - 1);
    std::vector<double> t(_chi_squared_table.rbegin()->first + 1, 0.0);
    for (const auto& kv : _chi_squared_table) t[kv.first] = kv.second;
    return t;
}

double UpdateBase::whitenResidual(const std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                  const std::vector<std::shared_ptr<Type>>& var_order, double noise)
{
    return StateManager::whitenResidual(state, res, H, var_order, noise);
}

double UpdateBase::whitenResidual(const std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                  const std::vector<std::shared_ptr<Type>>& var_order, const MatXd& R)
{
    return StateManager::whitenResidual(state, res, H, var_order, R);
}

bool UpdateBase::testChiSquared(const std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                const std::vector<std::shared_ptr<Type>>& var_order, double noise)
{
    const double prob = this->whitenResidual(state, res, H, var_order, noise);
    const int dof = (int)res.size();
    extendTable(dof);
    return prob < _chi_squared_table.at(dof);
}

bool UpdateBase::testChiSquared(const std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                const std::vector<std::shared_ptr<Type>>& var_order, double noise, int dof)
{
    const double prob = this->whitenResidual(state, res, H, var_order, noise);
    extendTable(dof);
    return prob < _chi_squared_table.at(dof);
}

bool UpdateBase::testChiSquared(const std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                const std::vector<std::shared_ptr<Type>>& var_order, const MatXd& R, int dof)
{
    if (dof <= 0) return false;
    const double prob = this->whitenResidual(state, res, H, var_order, R);
    extendTable(dof);
    return prob < _chi_squared_table.at(dof);
}

}  // namespace ingvio
