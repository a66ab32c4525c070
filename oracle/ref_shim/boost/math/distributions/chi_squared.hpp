// oracle/ref_shim/boost/math/distributions/chi_squared.hpp -- TEST INFRASTRUCTURE ONLY.
//
// Boost.Math is not installed in this image.  The reference uses exactly two names from it
// (/root/reference/include/msckf_mono/msckf.h:93-94): boost::math::chi_squared(dof) and
// boost::math::quantile(dist, p).  This header provides both: the quantile is the inverse of the regularized lower
// incomplete gamma function P(k/2, x/2), evaluated by series / continued fraction and inverted by a bracketed
// Newton iteration to double rounding.  tests/test_ref_vs_oracle.py holds it against scipy.stats.chi2.ppf.
#ifndef MSCKF_REF_SHIM_BOOST_CHI_SQUARED_HPP
#define MSCKF_REF_SHIM_BOOST_CHI_SQUARED_HPP

#include <cmath>
#include <limits>

namespace boost { namespace math {

namespace shim_detail {
// regularized lower incomplete gamma P(a, x)
inline double gamma_p(double a, double x) {
  if (x <= 0) return 0.0;
  const double lg = std::lgamma(a);
  if (x < a + 1.0) {               // series
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 100000; ++n) {
      ap += 1.0; del *= x / ap; sum += del;
      if (std::fabs(del) < std::fabs(sum) * 1e-17) break;
    }
    return sum * std::exp(-x + a * std::log(x) - lg);
  }
  // continued fraction for Q(a, x) (modified Lentz)
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 100000; ++i) {
    const double an = -i * (i - a);
    b += 2.0;
    d = an * d + b; if (std::fabs(d) < tiny) d = tiny;
    c = b + an / c; if (std::fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (std::fabs(del - 1.0) < 1e-17) break;
  }
  return 1.0 - std::exp(-x + a * std::log(x) - lg) * h;
}
inline double gamma_p_inv(double a, double p) {
  if (p <= 0) return 0.0;
  if (p >= 1) return std::numeric_limits<double>::infinity();
  double lo = 0.0, hi = a + 1.0;
  while (gamma_p(a, hi) < p) hi *= 2.0;
  double x = 0.5 * (lo + hi);
  const double lg = std::lgamma(a);
  for (int it = 0; it < 400; ++it) {
    const double f = gamma_p(a, x) - p;
    if (f > 0) hi = x; else lo = x;
    const double pdf = std::exp(-x + (a - 1.0) * std::log(x) - lg);
    double xn = x - f / pdf;
    if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
    if (std::fabs(xn - x) <= 4e-16 * std::fabs(xn)) { x = xn; break; }
    x = xn;
  }
  return x;
}
}  // namespace shim_detail

template <class RealType = double> class chi_squared_distribution {
  RealType dof_;

 public:
  typedef RealType value_type;
  chi_squared_distribution(RealType dof) : dof_(dof) {}
  RealType degrees_of_freedom() const { return dof_; }
};
typedef chi_squared_distribution<double> chi_squared;

template <class RealType, class P> inline RealType quantile(const chi_squared_distribution<RealType>& d, const P& p) {
  return RealType(2.0 * shim_detail::gamma_p_inv(0.5 * (double)d.degrees_of_freedom(), (double)p));
}
template <class RealType, class X> inline RealType cdf(const chi_squared_distribution<RealType>& d, const X& x) {
  return RealType(shim_detail::gamma_p(0.5 * (double)d.degrees_of_freedom(), 0.5 * (double)x));
}

}}  // namespace boost::math
#endif
