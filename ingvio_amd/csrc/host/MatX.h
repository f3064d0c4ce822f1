// MatX.h — tiny column-major dynamic matrix standing in for Eigen::MatrixXd / VectorXd in the shim's
// signatures (Eigen is not installed here).  data() + rows() is exactly what Eigen::Map needs, so a
// reference-side adapter is `Eigen::Map<const Eigen::MatrixXd>(m.data(), m.rows(), m.cols())`.
#pragma once
#include <vector>

namespace ingvio {

class MatXd {
public:
    MatXd() : _r(0), _c(0) {}
    MatXd(int r, int c) : _r(r), _c(c), _d((size_t)r * c, 0.0) {}
    static MatXd Identity(int n) { MatXd m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1.0; return m; }
    int rows() const { return _r; }
    int cols() const { return _c; }
    double& operator()(int i, int j) { return _d[(size_t)j * _r + i]; }
    double operator()(int i, int j) const { return _d[(size_t)j * _r + i]; }
    double* data() { return _d.data(); }
    const double* data() const { return _d.data(); }
    void resize(int r, int c) { _r = r; _c = c; _d.assign((size_t)r * c, 0.0); }

private:
    int _r, _c;
    std::vector<double> _d;
};

typedef std::vector<double> VecXd;

}  // namespace ingvio
