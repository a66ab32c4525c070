// oracle/la.hpp -- TEST INFRASTRUCTURE ONLY (CPU oracle for parity checks; never shipped, never
// linked by the product library).
//
// A small dense linear-algebra kit that restates, in plain C++17 without Eigen, the third-party
// arithmetic the reference calls from include/msckf_mono/msckf.h:
//   * Eigen::Quaternion            (toRotationMatrix, operator*, inverse, normalize, _transformVector,
//                                   angularDistance)                     msckf.h:116-160,851-872,1070
//   * MatrixBase::exp()            (unsupported/MatrixFunctions, Higham-2005 Pade scaling&squaring)
//                                                                         msckf.h:111
//   * HouseholderQR                (unblocked, beta=-sign(c0)*norm, tau=0 on a zero tail)  msckf.h:1343
//   * LDLT().solve                                                         msckf.h:1115,1222
//   * PartialPivLU inverse / determinant                                   msckf.h:176,1370
// Eigen/Boost are absent from the build image and un-pinned by the reference (CMakeLists.txt:22,39);
// only the *values* of these operations matter for parity, so they are restated from their published
// algorithms.  The reference ships no golden vectors; parity is PINNED by running the
// reference's own unmodified msckf.h (oracle/_ref/lib_ref.so, built by oracle/Makefile against oracle/ref_shim)
// on the same inputs: tests/test_ref_vs_oracle.py.
#ifndef ORACLE_LA_HPP
#define ORACLE_LA_HPP

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>
#include <vector>

namespace oracle {

template <class S>
struct Mat {  // column-major dense matrix
  int r = 0, c = 0;
  std::vector<S> a;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, S(0)) {}
  void resize(int r_, int c_) { r = r_; c = c_; a.assign((size_t)r_ * c_, S(0)); }
  S& operator()(int i, int j) { return a[(size_t)j * r + i]; }
  const S& operator()(int i, int j) const { return a[(size_t)j * r + i]; }
  static Mat identity(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = S(1); return m; }
  Mat block(int i0, int j0, int nr, int nc) const {
    Mat m(nr, nc);
    for (int j = 0; j < nc; ++j) for (int i = 0; i < nr; ++i) m(i, j) = (*this)(i0 + i, j0 + j);
    return m;
  }
  void set_block(int i0, int j0, const Mat& m) {
    for (int j = 0; j < m.c; ++j) for (int i = 0; i < m.r; ++i) (*this)(i0 + i, j0 + j) = m(i, j);
  }
  Mat t() const {
    Mat m(c, r);
    for (int j = 0; j < c; ++j) for (int i = 0; i < r; ++i) m(j, i) = (*this)(i, j);
    return m;
  }
};

template <class S>
Mat<S> mul(const Mat<S>& A, const Mat<S>& B) {  // C = A*B
  Mat<S> C(A.r, B.c);
  for (int j = 0; j < B.c; ++j)
    for (int k = 0; k < A.c; ++k) {
      const S b = B(k, j);
      if (b == S(0)) continue;
      const S* ak = &A.a[(size_t)k * A.r];
      S* cj = &C.a[(size_t)j * C.r];
      for (int i = 0; i < A.r; ++i) cj[i] += ak[i] * b;
    }
  return C;
}
template <class S>
Mat<S> mul_abt(const Mat<S>& A, const Mat<S>& B) {  // C = A*B^T
  Mat<S> C(A.r, B.r);
  for (int k = 0; k < A.c; ++k)
    for (int j = 0; j < B.r; ++j) {
      const S b = B(j, k);
      if (b == S(0)) continue;
      const S* ak = &A.a[(size_t)k * A.r];
      S* cj = &C.a[(size_t)j * C.r];
      for (int i = 0; i < A.r; ++i) cj[i] += ak[i] * b;
    }
  return C;
}
template <class S>
Mat<S> mul_atb(const Mat<S>& A, const Mat<S>& B) {  // C = A^T*B
  Mat<S> C(A.c, B.c);
  for (int j = 0; j < B.c; ++j)
    for (int i = 0; i < A.c; ++i) {
      const S* ai = &A.a[(size_t)i * A.r];
      const S* bj = &B.a[(size_t)j * B.r];
      S s = 0;
      for (int k = 0; k < A.r; ++k) s += ai[k] * bj[k];
      C(i, j) = s;
    }
  return C;
}
template <class S>
Mat<S> add(const Mat<S>& A, const Mat<S>& B, S sb = S(1)) {
  Mat<S> C = A;
  for (size_t i = 0; i < C.a.size(); ++i) C.a[i] += sb * B.a[i];
  return C;
}
template <class S>
void symmetrize(Mat<S>& A) {  // (A + A^T)/2, msckf.h:143,197,1401-1403
  for (int j = 0; j < A.c; ++j)
    for (int i = 0; i < j; ++i) {
      S v = (A(i, j) + A(j, i)) / S(2);
      A(i, j) = v; A(j, i) = v;
    }
}

// ---------------------------------------------------------------- small fixed-size helpers
template <class S> struct V3 { S x = 0, y = 0, z = 0; };
template <class S> V3<S> operator+(V3<S> a, V3<S> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class S> V3<S> operator-(V3<S> a, V3<S> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class S> V3<S> operator*(S s, V3<S> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class S> S dot(V3<S> a, V3<S> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class S> V3<S> cross(V3<S> a, V3<S> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class S> S norm(V3<S> a) { return std::sqrt(dot(a, a)); }

template <class S> struct M3 {  // row-major 3x3
  S m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  static M3 identity() { M3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = 1; return r; }
  M3 t() const { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[j][i]; return r; }
};
template <class S> M3<S> operator*(const M3<S>& a, const M3<S>& b) {
  M3<S> r;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    S s = 0; for (int k = 0; k < 3; ++k) s += a.m[i][k] * b.m[k][j]; r.m[i][j] = s;
  }
  return r;
}
template <class S> V3<S> operator*(const M3<S>& a, V3<S> v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
          a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
template <class S> M3<S> operator-(const M3<S>& a, const M3<S>& b) {
  M3<S> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] - b.m[i][j]; return r;
}
template <class S> M3<S> neg(const M3<S>& a) {
  M3<S> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = -a.m[i][j]; return r;
}
// vectorToSkewSymmetric, matrix_utils.h:8-17
template <class S> M3<S> skew(V3<S> v) {
  M3<S> r;
  r.m[0][1] = -v.z; r.m[0][2] = v.y;
  r.m[1][0] = v.z;  r.m[1][2] = -v.x;
  r.m[2][0] = -v.y; r.m[2][1] = v.x;
  return r;
}

// Eigen::Quaternion restated (coefficients w,x,y,z; Hamilton product).
template <class S> struct Quat {
  S w = 1, x = 0, y = 0, z = 0;
  M3<S> toRot() const {  // Eigen QuaternionBase::toRotationMatrix
    const S tx = S(2) * x, ty = S(2) * y, tz = S(2) * z;
    const S twx = tx * w, twy = ty * w, twz = tz * w;
    const S txx = tx * x, txy = ty * x, txz = tz * x;
    const S tyy = ty * y, tyz = tz * y, tzz = tz * z;
    M3<S> r;
    r.m[0][0] = S(1) - (tyy + tzz); r.m[0][1] = txy - twz; r.m[0][2] = txz + twy;
    r.m[1][0] = txy + twz; r.m[1][1] = S(1) - (txx + tzz); r.m[1][2] = tyz - twx;
    r.m[2][0] = txz - twy; r.m[2][1] = tyz + twx; r.m[2][2] = S(1) - (txx + tyy);
    return r;
  }
  S norm2() const { return w * w + x * x + y * y + z * z; }
  void normalize() { S n = std::sqrt(norm2()); w /= n; x /= n; y /= n; z /= n; }
  Quat normalized() const { Quat q = *this; q.normalize(); return q; }
  Quat conj() const { return {w, -x, -y, -z}; }
  Quat inverse() const {  // Eigen: conjugate / squaredNorm
    S n2 = norm2();
    if (n2 > S(0)) return {w / n2, -x / n2, -y / n2, -z / n2};
    return {0, 0, 0, 0};
  }
  V3<S> rotate(V3<S> v) const {  // Eigen _transformVector: v + w*uv + vec x uv, uv = 2*vec x v
    V3<S> qv{x, y, z};
    V3<S> uv = cross(qv, v);
    uv = uv + uv;
    return v + (w * uv) + cross(qv, uv);
  }
  S angularDistance(const Quat& o) const {  // Eigen 3.3: d = this*conj(o); 2*atan2(|vec|,|w|)
    Quat d = (*this) * o.conj();
    return S(2) * std::atan2(std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z), std::abs(d.w));
  }
  Quat operator*(const Quat& b) const {
    return {w * b.w - x * b.x - y * b.y - z * b.z,
            w * b.x + x * b.w + y * b.z - z * b.y,
            w * b.y + y * b.w + z * b.x - x * b.z,
            w * b.z + z * b.w + x * b.y - y * b.x};
  }
};

// ---------------------------------------------------------------- factorizations
// In-place unblocked Householder QR with Eigen's makeHouseholder convention:
//   tail^2 <= numeric_limits::min  -> tau = 0, beta = c0 (step is the identity)
//   else beta = -sign(c0)*sqrt(c0^2+tail^2), essential = tail/(c0-beta), tau = (beta-c0)/beta.
// On return A holds R in its upper triangle and the essential parts below; tau[k] per step.
// tail_tol > 0: the zero-tail rule with a rounding threshold -- a tail below tail_tol * |column| (a column that depends
// on the previous ones: exactly zero in exact arithmetic) is treated as the zero it stands for.
template <class S>
void householder_qr_inplace(Mat<S>& A, std::vector<S>& tau, double tail_tol = 0) {
  const int m = A.r, n = A.c, steps = std::min(m, n);
  tau.assign(steps, S(0));
  std::vector<S> w(n);
  for (int k = 0; k < steps; ++k) {
    S* ck = &A.a[(size_t)k * m];
    S tail2 = 0;
    for (int i = k + 1; i < m; ++i) tail2 += ck[i] * ck[i];
    const S c0 = ck[k];
    S zero2 = std::numeric_limits<S>::min();
    if (tail_tol > 0) { S head2 = 0; for (int i = 0; i <= k; ++i) head2 += ck[i] * ck[i]; zero2 = std::max(zero2, S(tail_tol * tail_tol) * (head2 + tail2)); }
    if (tail2 <= zero2) {
      tau[k] = 0;
      for (int i = k + 1; i < m; ++i) ck[i] = 0;
      continue;
    }
    S beta = std::sqrt(c0 * c0 + tail2);
    if (c0 >= S(0)) beta = -beta;
    const S inv = S(1) / (c0 - beta);
    for (int i = k + 1; i < m; ++i) ck[i] *= inv;
    tau[k] = (beta - c0) / beta;
    ck[k] = beta;
    // apply H = I - tau v v^T (v = [1; essential]) to the trailing columns
    for (int j = k + 1; j < n; ++j) {
      S* cj = &A.a[(size_t)j * m];
      S s = cj[k];
      for (int i = k + 1; i < m; ++i) s += ck[i] * cj[i];
      s *= tau[k];
      cj[k] -= s;
      for (int i = k + 1; i < m; ++i) cj[i] -= s * ck[i];
    }
  }
}
// Column-pivoted variant: before step k the remaining column (rows k..) of largest squared norm is swapped into
// place.  This is the QR preconditioner Eigen's JacobiSVD runs on a tall matrix (ColPivHouseholderQR); the last
// rows-cols columns of its full Q ARE the trailing columns of JacobiSVD::matrixU() that the reference reads as the
// left null space (msckf.h:954-955).  Norms are recomputed per step (Eigen downdates them: same choice away from ties).
template <class S>
void householder_qr_colpiv_inplace(Mat<S>& A, std::vector<S>& tau) {
  const int m = A.r, n = A.c, steps = std::min(m, n);
  tau.assign(steps, S(0));
  for (int k = 0; k < steps; ++k) {
    int big = k; S best = S(-1);
    for (int j = k; j < n; ++j) {
      S s = 0; for (int i = k; i < m; ++i) s += A(i, j) * A(i, j);
      if (s > best) { best = s; big = j; }
    }
    if (big != k) for (int i = 0; i < m; ++i) std::swap(A(i, k), A(i, big));
    S* ck = &A.a[(size_t)k * m];
    S tail2 = 0;
    for (int i = k + 1; i < m; ++i) tail2 += ck[i] * ck[i];
    const S c0 = ck[k];
    if (tail2 <= std::numeric_limits<S>::min()) { tau[k] = 0; for (int i = k + 1; i < m; ++i) ck[i] = 0; continue; }
    S beta = std::sqrt(c0 * c0 + tail2);
    if (c0 >= S(0)) beta = -beta;
    const S inv = S(1) / (c0 - beta);
    for (int i = k + 1; i < m; ++i) ck[i] *= inv;
    tau[k] = (beta - c0) / beta;
    ck[k] = beta;
    for (int j = k + 1; j < n; ++j) {
      S* cj = &A.a[(size_t)j * m];
      S s = cj[k];
      for (int i = k + 1; i < m; ++i) s += ck[i] * cj[i];
      s *= tau[k];
      cj[k] -= s;
      for (int i = k + 1; i < m; ++i) cj[i] -= s * ck[i];
    }
  }
}
// Apply Q^T = H_{s-1}...H_1 H_0 to the columns of X (m x p), reflectors stored in QR/tau.
template <class S>
void apply_qt(const Mat<S>& QR, const std::vector<S>& tau, Mat<S>& X) {
  const int m = QR.r;
  for (int k = 0; k < (int)tau.size(); ++k) {
    if (tau[k] == S(0)) continue;
    const S* vk = &QR.a[(size_t)k * m];
    for (int j = 0; j < X.c; ++j) {
      S* xj = &X.a[(size_t)j * m];
      S s = xj[k];
      for (int i = k + 1; i < m; ++i) s += vk[i] * xj[i];
      s *= tau[k];
      xj[k] -= s;
      for (int i = k + 1; i < m; ++i) xj[i] -= s * vk[i];
    }
  }
}
// Form columns [c0, c0+nc) of Q = H_0 H_1 ... H_{s-1} (m x m): apply reflectors in reverse to e_j.
template <class S>
Mat<S> form_q_cols(const Mat<S>& QR, const std::vector<S>& tau, int c0, int nc) {
  const int m = QR.r;
  Mat<S> Q(m, nc);
  for (int j = 0; j < nc; ++j) Q(c0 + j, j) = S(1);
  for (int k = (int)tau.size() - 1; k >= 0; --k) {
    if (tau[k] == S(0)) continue;
    const S* vk = &QR.a[(size_t)k * m];
    for (int j = 0; j < nc; ++j) {
      S* qj = &Q.a[(size_t)j * m];
      S s = qj[k];
      for (int i = k + 1; i < m; ++i) s += vk[i] * qj[i];
      s *= tau[k];
      qj[k] -= s;
      for (int i = k + 1; i < m; ++i) qj[i] -= s * vk[i];
    }
  }
  return Q;
}

// LDL^T (no pivoting; the matrices it is used on are SPD: msckf.h:1115 S = HPH^T + sigma^2 I,
// msckf.h:1222 A + lambda I) followed by solve.  Eigen's LDLT pivots on the diagonal; for SPD input
// the solution is the same up to rounding.
template <class S>
std::vector<S> ldlt_solve(Mat<S> A, std::vector<S> b) {
  const int n = A.r;
  for (int j = 0; j < n; ++j) {
    S d = A(j, j);
    for (int k = 0; k < j; ++k) d -= A(j, k) * A(j, k) * A(k, k);
    A(j, j) = d;
    for (int i = j + 1; i < n; ++i) {
      S s = A(i, j);
      for (int k = 0; k < j; ++k) s -= A(i, k) * A(j, k) * A(k, k);
      A(i, j) = s / d;
    }
  }
  for (int i = 0; i < n; ++i) for (int k = 0; k < i; ++k) b[i] -= A(i, k) * b[k];
  for (int i = 0; i < n; ++i) b[i] /= A(i, i);
  for (int i = n - 1; i >= 0; --i) for (int k = i + 1; k < n; ++k) b[i] -= A(k, i) * b[k];
  return b;
}

// LU with partial pivoting (Eigen PartialPivLU restated): returns false when singular.
template <class S>
bool lu_factor(Mat<S>& A, std::vector<int>& piv, int& sign) {
  const int n = A.r;
  piv.resize(n);
  sign = 1;
  for (int k = 0; k < n; ++k) {
    int p = k; S best = std::abs(A(k, k));
    for (int i = k + 1; i < n; ++i) if (std::abs(A(i, k)) > best) { best = std::abs(A(i, k)); p = i; }
    piv[k] = p;
    if (best == S(0)) return false;
    if (p != k) { for (int j = 0; j < n; ++j) std::swap(A(k, j), A(p, j)); sign = -sign; }
    const S inv = S(1) / A(k, k);
    for (int i = k + 1; i < n; ++i) A(i, k) *= inv;
    for (int j = k + 1; j < n; ++j) {
      const S akj = A(k, j);
      if (akj == S(0)) continue;
      for (int i = k + 1; i < n; ++i) A(i, j) -= A(i, k) * akj;
    }
  }
  return true;
}
template <class S>
Mat<S> lu_solve(const Mat<S>& LU, const std::vector<int>& piv, Mat<S> B) {
  const int n = LU.r;
  for (int k = 0; k < n; ++k) if (piv[k] != k) for (int j = 0; j < B.c; ++j) std::swap(B(k, j), B(piv[k], j));
  for (int j = 0; j < B.c; ++j) {
    for (int i = 0; i < n; ++i) { S s = B(i, j); for (int k = 0; k < i; ++k) s -= LU(i, k) * B(k, j); B(i, j) = s; }
    for (int i = n - 1; i >= 0; --i) {
      S s = B(i, j); for (int k = i + 1; k < n; ++k) s -= LU(i, k) * B(k, j); B(i, j) = s / LU(i, i);
    }
  }
  return B;
}
template <class S>
Mat<S> inverse(const Mat<S>& A) {
  Mat<S> LU = A; std::vector<int> piv; int sign;
  lu_factor(LU, piv, sign);
  return lu_solve(LU, piv, Mat<S>::identity(A.r));
}
template <class S>
S determinant(const Mat<S>& A) {
  Mat<S> LU = A; std::vector<int> piv; int sign;
  if (!lu_factor(LU, piv, sign)) return S(0);
  S d = S(sign);
  for (int i = 0; i < A.r; ++i) d *= LU(i, i);
  return d;
}

// Matrix exponential: Higham (2005) scaling-and-squaring Pade, with the degree switch points of
// Eigen's unsupported/MatrixFunctions/MatrixExponential.h (float: 3/5/7, double: 3/5/7/9/13).
namespace detail {
template <class S> Mat<S> scaled(const Mat<S>& A, S s) { Mat<S> B = A; for (auto& v : B.a) v *= s; return B; }
template <class S>
void pade_uv(const Mat<S>& A, const double* b, int deg, Mat<S>& U, Mat<S>& V) {
  const int n = A.r;
  const Mat<S> I = Mat<S>::identity(n);
  const Mat<S> A2 = mul(A, A);
  std::vector<Mat<S>> pw;  // even powers: A^0, A^2, A^4, ...
  pw.push_back(I); pw.push_back(A2);
  for (int k = 2; 2 * k <= deg; ++k) pw.push_back(mul(pw.back(), A2));
  Mat<S> Uo(n, n), Ve(n, n);
  for (int k = 0; 2 * k <= deg; ++k) {
    if (2 * k + 1 <= deg) Uo = add(Uo, pw[k], S(b[2 * k + 1]));
    Ve = add(Ve, pw[k], S(b[2 * k]));
  }
  U = mul(A, Uo);
  V = Ve;
}
}  // namespace detail
template <class S>
Mat<S> expm(const Mat<S>& Ain) {
  static const double b3[] = {120., 60., 12., 1.};
  static const double b5[] = {30240., 15120., 3360., 420., 30., 1.};
  static const double b7[] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
  static const double b9[] = {17643225600., 8821612800., 2075673600., 302702400., 30270240.,
                              2162160., 110880., 3960., 90., 1.};
  static const double b13[] = {64764752532480000., 32382376266240000., 7771770303897600.,
                               1187353796428800., 129060195264000., 10559470521600., 670442572800.,
                               33522128640., 1323241920., 40840800., 960960., 16380., 182., 1.};
  double l1 = 0;
  for (int j = 0; j < Ain.c; ++j) { double s = 0; for (int i = 0; i < Ain.r; ++i) s += std::abs((double)Ain(i, j)); l1 = std::max(l1, s); }
  int squarings = 0;
  Mat<S> A = Ain, U, V;
  if (sizeof(S) == sizeof(float)) {
    if (l1 < 4.258730016922831e-001) detail::pade_uv(A, b3, 3, U, V);
    else if (l1 < 1.880152677804762e+000) detail::pade_uv(A, b5, 5, U, V);
    else {
      const double maxnorm = 3.925724783138660;
      std::frexp(l1 / maxnorm, &squarings); if (squarings < 0) squarings = 0;
      A = detail::scaled(Ain, S(std::ldexp(1.0, -squarings)));
      detail::pade_uv(A, b7, 7, U, V);
    }
  } else {
    if (l1 < 1.495585217958292e-002) detail::pade_uv(A, b3, 3, U, V);
    else if (l1 < 2.539398330063230e-001) detail::pade_uv(A, b5, 5, U, V);
    else if (l1 < 9.504178996162932e-001) detail::pade_uv(A, b7, 7, U, V);
    else if (l1 < 2.097847961257068e+000) detail::pade_uv(A, b9, 9, U, V);
    else {
      const double maxnorm = 5.371920351148152;
      std::frexp(l1 / maxnorm, &squarings); if (squarings < 0) squarings = 0;
      A = detail::scaled(Ain, S(std::ldexp(1.0, -squarings)));
      detail::pade_uv(A, b13, 13, U, V);
    }
  }
  Mat<S> numer = add(U, V), denom = add(V, U, S(-1));
  Mat<S> LU = denom; std::vector<int> piv; int sign;
  lu_factor(LU, piv, sign);
  Mat<S> R = lu_solve(LU, piv, numer);
  for (int i = 0; i < squarings; ++i) R = mul(R, R);
  return R;
}

}  // namespace oracle
#endif
