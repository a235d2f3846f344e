// oracle/dual.hpp -- forward-mode dual numbers for the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  The reference obtains its gradient
// by running the SAME templated objective with Type = CppAD::AD<double>
// (TMB::MakeADFun, /root/reference/R/sde.R:656-658; objective template
// /root/reference/src/smoothSDE.cpp:9-28).  CppAD is not in this image, so the oracle
// instantiates the restated templates with this Dual<N> type instead: it differentiates
// the branch that is actually taken, exactly like a tape does.
#ifndef SSDE_ORACLE_DUAL_HPP
#define SSDE_ORACLE_DUAL_HPP
#include <cmath>

namespace ssde_oracle {

template <int N>
struct Dual {
    double v;
    double d[N];
    Dual() : v(0.0) { for (int k = 0; k < N; k++) d[k] = 0.0; }
    Dual(double x) : v(x) { for (int k = 0; k < N; k++) d[k] = 0.0; }
    Dual(int x) : v((double)x) { for (int k = 0; k < N; k++) d[k] = 0.0; }
};

#define SSDE_DUAL_BIN(OP, VEXPR, DEXPR)                                         \
    template <int N> inline Dual<N> operator OP(const Dual<N>& a, const Dual<N>& b) { \
        Dual<N> r; r.v = VEXPR; for (int k = 0; k < N; k++) r.d[k] = DEXPR; return r; }

SSDE_DUAL_BIN(+, a.v + b.v, a.d[k] + b.d[k])
SSDE_DUAL_BIN(-, a.v - b.v, a.d[k] - b.d[k])
SSDE_DUAL_BIN(*, a.v * b.v, a.d[k] * b.v + a.v * b.d[k])
template <int N> inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.v = a.v / b.v;
    for (int k = 0; k < N; k++) r.d[k] = (a.d[k] - r.v * b.d[k]) / b.v;
    return r;
}
#undef SSDE_DUAL_BIN

template <int N> inline Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator+(double a, const Dual<N>& b) { return b + a; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> inline Dual<N> operator-(double a, const Dual<N>& b) {
    Dual<N> r; r.v = a - b.v; for (int k = 0; k < N; k++) r.d[k] = -b.d[k]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a) {
    Dual<N> r; r.v = -a.v; for (int k = 0; k < N; k++) r.d[k] = -a.d[k]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double b) {
    Dual<N> r; r.v = a.v * b; for (int k = 0; k < N; k++) r.d[k] = a.d[k] * b; return r; }
template <int N> inline Dual<N> operator*(double a, const Dual<N>& b) { return b * a; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double b) {
    Dual<N> r; r.v = a.v / b; for (int k = 0; k < N; k++) r.d[k] = a.d[k] / b; return r; }
template <int N> inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N> inline Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N> inline Dual<N>& operator-=(Dual<N>& a, const Dual<N>& b) { a = a - b; return a; }

template <int N> inline bool operator<=(const Dual<N>& a, double b) { return a.v <= b; }
template <int N> inline bool operator>(const Dual<N>& a, double b) { return a.v > b; }
template <int N> inline bool operator<(const Dual<N>& a, double b) { return a.v < b; }

template <int N> inline Dual<N> exp(const Dual<N>& a) {
    Dual<N> r; r.v = std::exp(a.v); for (int k = 0; k < N; k++) r.d[k] = r.v * a.d[k]; return r; }
template <int N> inline Dual<N> log(const Dual<N>& a) {
    Dual<N> r; r.v = std::log(a.v); for (int k = 0; k < N; k++) r.d[k] = a.d[k] / a.v; return r; }
template <int N> inline Dual<N> sqrt(const Dual<N>& a) {
    Dual<N> r; r.v = std::sqrt(a.v); for (int k = 0; k < N; k++) r.d[k] = a.d[k] / (2.0 * r.v); return r; }
template <int N> inline Dual<N> fabs(const Dual<N>& a) { return a.v < 0 ? -a : a; }

// digamma for the derivative of lgamma (x > 0): recurrence to x >= 10, then the asymptotic series
inline double digamma_d(double x) {
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    double i2 = 1.0 / (x * x);
    return r + std::log(x) - 0.5 / x - i2 * (1.0 / 12.0 - i2 * (1.0 / 120.0 - i2 * (1.0 / 252.0 - i2 * (1.0 / 240.0 - i2 * (1.0 / 132.0)))));
}
inline double lgamma(double x) { return std::lgamma(x); }
template <int N> inline Dual<N> lgamma(const Dual<N>& a) {
    Dual<N> r; r.v = std::lgamma(a.v); double psi = digamma_d(a.v); for (int k = 0; k < N; k++) r.d[k] = psi * a.d[k]; return r; }

inline double asDouble(double x) { return x; }
template <int N> inline double asDouble(const Dual<N>& x) { return x.v; }

using std::exp;
using std::log;
using std::sqrt;
using std::fabs;

}  // namespace ssde_oracle
#endif
