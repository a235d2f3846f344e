// oracle/quad.hpp -- IEEE binary128 scalar (gcc __float128 + libquadmath) for the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  The restated templates of ssde_oracle.hpp are instantiated with
// this type to evaluate the reference's formulas (/root/reference/src/nllk/*.hpp) with a 113-bit mantissa at the SAME
// double-precision inputs.  Where the double-precision oracle (a literal restatement) and the HIP engine (algebraically
// rearranged) disagree, the binary128 value says which of the two carries the rounding noise: it is the arbiter of
// tools/extreme_triage.py and tests/test_oracle_quad.py, never a product dependency.
#ifndef SSDE_ORACLE_QUAD_HPP
#define SSDE_ORACLE_QUAD_HPP
#include <quadmath.h>

namespace ssde_oracle {

struct Quad {
    __float128 v;
    Quad() : v(0) {}
    Quad(double x) : v(x) {}
    Quad(int x) : v(x) {}
    Quad(long x) : v(x) {}
    explicit Quad(__float128 x, int) : v(x) {}
};
inline Quad q128(__float128 x) { return Quad(x, 0); }

inline Quad operator+(const Quad& a, const Quad& b) { return q128(a.v + b.v); }
inline Quad operator-(const Quad& a, const Quad& b) { return q128(a.v - b.v); }
inline Quad operator*(const Quad& a, const Quad& b) { return q128(a.v * b.v); }
inline Quad operator/(const Quad& a, const Quad& b) { return q128(a.v / b.v); }
inline Quad operator-(const Quad& a) { return q128(-a.v); }
inline Quad operator+(const Quad& a, double b) { return q128(a.v + b); }
inline Quad operator+(double a, const Quad& b) { return q128(a + b.v); }
inline Quad operator-(const Quad& a, double b) { return q128(a.v - b); }
inline Quad operator-(double a, const Quad& b) { return q128(a - b.v); }
inline Quad operator*(const Quad& a, double b) { return q128(a.v * b); }
inline Quad operator*(double a, const Quad& b) { return q128(a * b.v); }
inline Quad operator/(const Quad& a, double b) { return q128(a.v / b); }
inline Quad operator/(double a, const Quad& b) { return q128(a / b.v); }
inline Quad& operator+=(Quad& a, const Quad& b) { a.v += b.v; return a; }
inline Quad& operator-=(Quad& a, const Quad& b) { a.v -= b.v; return a; }
inline bool operator<=(const Quad& a, double b) { return a.v <= b; }
inline bool operator>(const Quad& a, double b) { return a.v > b; }
inline bool operator<(const Quad& a, double b) { return a.v < b; }

inline Quad exp(const Quad& a) { return q128(expq(a.v)); }
inline Quad log(const Quad& a) { return q128(logq(a.v)); }
inline Quad sqrt(const Quad& a) { return q128(sqrtq(a.v)); }
inline Quad fabs(const Quad& a) { return q128(fabsq(a.v)); }
inline Quad lgamma(const Quad& a) { return q128(lgammaq(a.v)); }
inline double asDouble(const Quad& x) { return (double)x.v; }

}  // namespace ssde_oracle
#endif
