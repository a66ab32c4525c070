// oracle/ref_shim/Eigen/src/MiniEigen.h -- TEST INFRASTRUCTURE ONLY.
//
// Eigen 3 and Boost.Math are not installed in this image and there is no network, so the reference
// (/root/reference/include/msckf_mono/msckf.h:15-19) cannot be compiled as shipped.  This header provides exactly
// the part of Eigen's public surface that msckf.h, types.h and matrix_utils.h use, so that the reference's own,
// UNMODIFIED source compiles into oracle/_ref/lib_ref.so (recipe: oracle/Makefile, target _ref) and its control
// flow and formulas -- not a restatement of them -- can be run beside oracle/msckf_oracle.hpp and the HIP path.
//
// It is not Eigen: expressions are evaluated eagerly into plain column-major matrices, there are no expression
// templates, no alignment tricks and no blocking.  The numerical algorithms behind the decompositions follow
// Eigen's documented ones so that results agree with an Eigen build to rounding:
//   * Householder reflectors with Eigen's makeHouseholder convention (beta = -sign(c0)*norm, tau = 0 and the
//     column left untouched when the tail is exactly zero)                       -> HouseholderQR, msckf.h:1343
//   * JacobiSVD(ComputeFullU) of a tall matrix = column-pivoted Householder QR preconditioner whose full Q gives
//     the last rows-cols columns of U unchanged (only those are read, msckf.h:954-955); the leading columns are
//     finished by one-sided Jacobi rotations
//   * MatrixBase::exp() = Higham (2005) scaling-and-squaring Pade with Eigen's degree switch points, msckf.h:111
//   * LDLT with diagonal pivoting, msckf.h:1115,1222; PartialPivLU inverse / determinant, msckf.h:176,1370;
//     closed-form inverses up to 3x3, msckf.h:121,944,1140
//   * Quaternion / Transform conventions of Eigen/Geometry (Hamilton product, coefficient order x,y,z,w)
#ifndef MSCKF_REF_SHIM_MINI_EIGEN_H
#define MSCKF_REF_SHIM_MINI_EIGEN_H

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>
#include <memory>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_WORLD_VERSION 3
#define EIGEN_MAJOR_VERSION 3
#define EIGEN_MINOR_VERSION 0

namespace Eigen {

const int Dynamic = -1;
enum StorageOptions { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2 };
enum DecompositionOptions { ComputeFullU = 0x04, ComputeThinU = 0x08, ComputeFullV = 0x10, ComputeThinV = 0x20 };
enum UpLoType { Lower = 0x1, Upper = 0x2 };
enum TransformTraits { Isometry = 0x1, Affine = 0x2, AffineCompact = 0x10 | Affine, Projective = 0x20 };
typedef std::ptrdiff_t Index;

template <class T> using aligned_allocator = std::allocator<T>;

template <class D> struct traits;
template <class Derived> class MatrixBase;
template <class S, int R, int C, int O = 0, int MR = R, int MC = C> class Matrix;
template <class M, int BR, int BC> class Block;
template <class S, int O = 0> class Quaternion;
template <class MatrixType> class LDLT;

template <class S, int R, int C, int O, int MR, int MC> struct traits<Matrix<S, R, C, O, MR, MC>> {
  typedef S Scalar;
  enum { Rows = R, Cols = C };
};
template <class M, int BR, int BC> struct traits<Block<M, BR, BC>> {
  typedef typename traits<typename std::remove_const<M>::type>::Scalar Scalar;
  enum { Rows = BR, Cols = BC };
};

namespace internal {
constexpr int merge_dim(int a, int b) { return a != Dynamic ? a : b; }
template <class D> using plain_t = Matrix<typename traits<D>::Scalar, traits<D>::Rows, traits<D>::Cols>;
template <class S> using dyn_t = Matrix<S, Dynamic, Dynamic>;

template <class S, int R, int C, bool Dyn = (R == Dynamic || C == Dynamic)> struct Storage;
template <class S, int R, int C> struct Storage<S, R, C, false> {
  S d[R * C];
  Index rows() const { return R; }
  Index cols() const { return C; }
  void resize(Index, Index) {}
  S* data() { return d; }
  const S* data() const { return d; }
};
template <class S, int R, int C> struct Storage<S, R, C, true> {
  std::vector<S> d;
  Index r = (R == Dynamic ? 0 : R), c = (C == Dynamic ? 0 : C);
  Index rows() const { return r; }
  Index cols() const { return c; }
  void resize(Index rr, Index cc) { r = rr; c = cc; d.resize((size_t)(rr * cc)); }
  S* data() { return d.data(); }
  const S* data() const { return d.data(); }
};

template <class D, class S, bool OneByOne> struct ScalarConv {};
template <class D, class S> struct ScalarConv<D, S, true> {
  operator S() const { return static_cast<const D*>(this)->coeff(0, 0); }
};
}  // namespace internal

template <class Derived> class CommaInitializer {
  Derived& m_;
  Index row_ = 0, col_ = 0, blk_ = 1;

 public:
  typedef typename traits<Derived>::Scalar S;
  template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
  CommaInitializer(Derived& m, const T& s) : m_(m) { m_.coeffRef(0, 0) = S(s); col_ = 1; blk_ = 1; }
  template <class O> CommaInitializer(Derived& m, const MatrixBase<O>& o) : m_(m) { put(o); }
  template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
  CommaInitializer& operator,(const T& s) {
    if (col_ == m_.cols()) { row_ += blk_; col_ = 0; blk_ = 1; }
    m_.coeffRef(row_, col_++) = S(s);
    return *this;
  }
  template <class O> CommaInitializer& operator,(const MatrixBase<O>& o) {
    if (col_ == m_.cols()) { row_ += blk_; col_ = 0; }
    put(o);
    return *this;
  }

 private:
  template <class O> void put(const MatrixBase<O>& o) {
    if (col_ == 0) blk_ = o.rows();
    for (Index j = 0; j < o.cols(); ++j)
      for (Index i = 0; i < o.rows(); ++i) m_.coeffRef(row_ + i, col_ + j) = S(o.coeff(i, j));
    col_ += o.cols();
  }
};

// rowwise().any(): Eigen returns a bool vector; here the 0/1 values are held in the matrix's own scalar type
// (std::vector<bool> cannot back a plain matrix), which is what msckf.h:1347 converts it to anyway
template <class Derived> struct RowwiseOp {
  const Derived& m;
  Matrix<typename traits<Derived>::Scalar, Dynamic, 1> any() const;
};

// -------------------------------------------------------------------------------------------------------------
template <class Derived> class MatrixBase {
 public:
  typedef typename traits<Derived>::Scalar Scalar;
  typedef Scalar S;
  enum { Rows = traits<Derived>::Rows, Cols = traits<Derived>::Cols };
  typedef internal::plain_t<Derived> Plain;

  Derived& derived() { return *static_cast<Derived*>(this); }
  const Derived& derived() const { return *static_cast<const Derived*>(this); }
  Index rows() const { return derived().rows(); }
  Index cols() const { return derived().cols(); }
  Index size() const { return rows() * cols(); }
  S coeff(Index i, Index j) const { return derived().coeff(i, j); }
  S coeff(Index i) const { return cols() == 1 ? coeff(i, 0) : coeff(0, i); }
  S operator()(Index i, Index j) const { return coeff(i, j); }
  S operator()(Index i) const { return coeff(i); }
  S& operator()(Index i, Index j) { return derived().coeffRef(i, j); }
  S& operator()(Index i) { return cols() == 1 ? derived().coeffRef(i, 0) : derived().coeffRef(0, i); }
  S operator[](Index i) const { return coeff(i); }
  S& operator[](Index i) { return (*this)(i); }

  Plain eval() const { return Plain(*this); }
  Plain array() const { return Plain(*this); }
  Plain matrix() const { return Plain(*this); }

  // ---- blocks (views into the underlying plain matrix)
  template <int BR, int BC> auto block(Index i, Index j) { return derived().template mk_block<BR, BC>(i, j, BR, BC); }
  template <int BR, int BC> auto block(Index i, Index j) const { return derived().template mk_block<BR, BC>(i, j, BR, BC); }
  auto block(Index i, Index j, Index r, Index c) { return derived().template mk_block<Dynamic, Dynamic>(i, j, r, c); }
  auto block(Index i, Index j, Index r, Index c) const { return derived().template mk_block<Dynamic, Dynamic>(i, j, r, c); }
  auto row(Index i) { return derived().template mk_block<1, Cols>(i, 0, 1, cols()); }
  auto row(Index i) const { return derived().template mk_block<1, Cols>(i, 0, 1, cols()); }
  auto col(Index j) { return derived().template mk_block<Rows, 1>(0, j, rows(), 1); }
  auto col(Index j) const { return derived().template mk_block<Rows, 1>(0, j, rows(), 1); }
  template <int N> auto leftCols() { return derived().template mk_block<Rows, N>(0, 0, rows(), N); }
  template <int N> auto leftCols() const { return derived().template mk_block<Rows, N>(0, 0, rows(), N); }
  auto leftCols(Index n) { return derived().template mk_block<Rows, Dynamic>(0, 0, rows(), n); }
  auto leftCols(Index n) const { return derived().template mk_block<Rows, Dynamic>(0, 0, rows(), n); }
  template <int N> auto rightCols() { return derived().template mk_block<Rows, N>(0, cols() - N, rows(), N); }
  template <int N> auto rightCols() const { return derived().template mk_block<Rows, N>(0, cols() - N, rows(), N); }
  auto rightCols(Index n) { return derived().template mk_block<Rows, Dynamic>(0, cols() - n, rows(), n); }
  auto rightCols(Index n) const { return derived().template mk_block<Rows, Dynamic>(0, cols() - n, rows(), n); }
  auto topRows(Index n) { return derived().template mk_block<Dynamic, Cols>(0, 0, n, cols()); }
  auto topRows(Index n) const { return derived().template mk_block<Dynamic, Cols>(0, 0, n, cols()); }
  auto bottomRows(Index n) { return derived().template mk_block<Dynamic, Cols>(rows() - n, 0, n, cols()); }
  auto bottomRows(Index n) const { return derived().template mk_block<Dynamic, Cols>(rows() - n, 0, n, cols()); }
  // vector segments: along the rows of a column vector, along the columns of a row vector
  template <int N> auto segment(Index i) { return seg_<N>(i, N); }
  template <int N> auto segment(Index i) const { return seg_<N>(i, N); }
  auto segment(Index i, Index n) { return seg_<Dynamic>(i, n); }
  auto segment(Index i, Index n) const { return seg_<Dynamic>(i, n); }
  template <int N> auto head() { return seg_<N>(0, N); }
  template <int N> auto head() const { return seg_<N>(0, N); }
  auto head(Index n) { return seg_<Dynamic>(0, n); }
  auto head(Index n) const { return seg_<Dynamic>(0, n); }
  template <int N> auto tail() { return seg_<N>(size() - N, N); }
  template <int N> auto tail() const { return seg_<N>(size() - N, N); }
  auto tail(Index n) { return seg_<Dynamic>(size() - n, n); }
  auto tail(Index n) const { return seg_<Dynamic>(size() - n, n); }

  // ---- reductions and unary functions
  S squaredNorm() const { S s = 0; for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) { const S v = coeff(i, j); s += v * v; } return s; }
  S norm() const { using std::sqrt; return sqrt(squaredNorm()); }
  S sum() const { S s = 0; for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) s += coeff(i, j); return s; }
  S trace() const { S s = 0; for (Index i = 0; i < std::min(rows(), cols()); ++i) s += coeff(i, i); return s; }
  Plain normalized() const { Plain p(*this); const S n = p.norm(); if (n > S(0)) p /= n; return p; }
  Matrix<S, Cols, Rows> transpose() const {
    Matrix<S, Cols, Rows> t; t.resize(cols(), rows());
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) t.coeffRef(j, i) = coeff(i, j);
    return t;
  }
  Matrix<S, Dynamic, 1> diagonal() const {
    const Index n = std::min(rows(), cols());
    Matrix<S, Dynamic, 1> d(n);
    for (Index i = 0; i < n; ++i) d.coeffRef(i, 0) = coeff(i, i);
    return d;
  }
  internal::dyn_t<S> asDiagonal() const {
    const Index n = size();
    internal::dyn_t<S> d = internal::dyn_t<S>::Zero(n, n);
    for (Index i = 0; i < n; ++i) d.coeffRef(i, i) = coeff(i);
    return d;
  }
  internal::dyn_t<S> replicate(Index rf, Index cf) const {
    internal::dyn_t<S> out(rows() * rf, cols() * cf);
    for (Index J = 0; J < cf; ++J) for (Index I = 0; I < rf; ++I)
      for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) out.coeffRef(I * rows() + i, J * cols() + j) = coeff(i, j);
    return out;
  }
  template <unsigned Mode> Plain triangularView() const {
    Plain p(*this);
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i)
      if ((Mode == Upper && i > j) || (Mode == Lower && i < j)) p.coeffRef(i, j) = S(0);
    return p;
  }
  RowwiseOp<Derived> rowwise() const { return RowwiseOp<Derived>{derived()}; }
  template <class O> S dot(const MatrixBase<O>& o) const { S s = 0; for (Index i = 0; i < size(); ++i) s += coeff(i) * o.coeff(i); return s; }
  template <class O> Matrix<S, 3, 1> cross(const MatrixBase<O>& o) const {
    return Matrix<S, 3, 1>(coeff(1) * o.coeff(2) - coeff(2) * o.coeff(1), coeff(2) * o.coeff(0) - coeff(0) * o.coeff(2),
                           coeff(0) * o.coeff(1) - coeff(1) * o.coeff(0));
  }
  Plain inverse() const;
  S determinant() const;
  Plain exp() const;
  LDLT<Plain> ldlt() const;

  // ---- in-place
  Derived& setZero() { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) derived().coeffRef(i, j) = S(0); return derived(); }
  Derived& setIdentity() { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) derived().coeffRef(i, j) = S(i == j ? 1 : 0); return derived(); }
  template <class T> Derived& setConstant(const T& v) { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) derived().coeffRef(i, j) = S(v); return derived(); }
  void normalize() { const S n = norm(); if (n > S(0)) *this /= n; }
  template <class O> Derived& operator+=(const MatrixBase<O>& o) { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) derived().coeffRef(i, j) += o.coeff(i, j); return derived(); }
  template <class O> Derived& operator-=(const MatrixBase<O>& o) { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) derived().coeffRef(i, j) -= o.coeff(i, j); return derived(); }
  template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
  Derived& operator*=(const T& s) { const S v = S(s); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) derived().coeffRef(i, j) *= v; return derived(); }
  template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
  Derived& operator/=(const T& s) { const S v = S(s); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) derived().coeffRef(i, j) /= v; return derived(); }
  template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
  CommaInitializer<Derived> operator<<(const T& s) { return CommaInitializer<Derived>(derived(), s); }
  template <class O> CommaInitializer<Derived> operator<<(const MatrixBase<O>& o) { return CommaInitializer<Derived>(derived(), o); }

 private:
  template <int N> auto seg_(Index i, Index n) {
    if constexpr (Rows == 1 && Cols != 1) return derived().template mk_block<1, N>(0, i, 1, n);
    else return derived().template mk_block<N, 1>(i, 0, n, 1);
  }
  template <int N> auto seg_(Index i, Index n) const {
    if constexpr (Rows == 1 && Cols != 1) return derived().template mk_block<1, N>(0, i, 1, n);
    else return derived().template mk_block<N, 1>(i, 0, n, 1);
  }
};

// -------------------------------------------------------------------------------------------------------------
template <class S, int R, int C, int O, int MR, int MC>
class Matrix : public MatrixBase<Matrix<S, R, C, O, MR, MC>>, public internal::ScalarConv<Matrix<S, R, C, O, MR, MC>, S, (R == 1 && C == 1)> {
  internal::Storage<S, R, C> st_;
  typedef MatrixBase<Matrix> Base;

 public:
  typedef S Scalar;
  using Base::operator();
  Matrix() {}
  template <class T, class = typename std::enable_if<std::is_integral<T>::value>::type>
  explicit Matrix(T n) { if (C == 1 || R != 1) st_.resize(R == Dynamic ? (Index)n : R, C == Dynamic ? 1 : C); else st_.resize(1, (Index)n); }
  template <class T0, class T1, class = typename std::enable_if<std::is_arithmetic<T0>::value && std::is_arithmetic<T1>::value>::type>
  Matrix(const T0& a, const T1& b) {
    if (R == Dynamic || C == Dynamic) st_.resize(R == Dynamic ? (Index)a : R, C == Dynamic ? (Index)b : C);
    else { data()[0] = S(a); data()[1] = S(b); }
  }
  template <class T0, class T1, class T2, class = typename std::enable_if<std::is_arithmetic<T0>::value>::type>
  Matrix(const T0& x, const T1& y, const T2& z) { st_.resize(R == Dynamic ? 3 : R, C == Dynamic ? 1 : C); data()[0] = S(x); data()[1] = S(y); data()[2] = S(z); }
  template <class T0, class T1, class T2, class T3, class = typename std::enable_if<std::is_arithmetic<T0>::value>::type>
  Matrix(const T0& x, const T1& y, const T2& z, const T3& w) { st_.resize(R == Dynamic ? 4 : R, C == Dynamic ? 1 : C); data()[0] = S(x); data()[1] = S(y); data()[2] = S(z); data()[3] = S(w); }
  template <class Ot> Matrix(const MatrixBase<Ot>& o) { assign(o); }
  template <int QO> Matrix(const Quaternion<S, QO>& q) { *this = q.toRotationMatrix(); }
  template <class Ot> Matrix& operator=(const MatrixBase<Ot>& o) { assign(o); return *this; }

  Index rows() const { return st_.rows(); }
  Index cols() const { return st_.cols(); }
  S* data() { return st_.data(); }
  const S* data() const { return st_.data(); }
  S coeff(Index i, Index j) const { return st_.data()[j * st_.rows() + i]; }
  S& coeffRef(Index i, Index j) { return st_.data()[j * st_.rows() + i]; }
  void resize(Index r, Index c) { st_.resize(r, c); }
  void resize(Index n) { if (C == 1) st_.resize(n, 1); else st_.resize(1, n); }
  void conservativeResize(Index r, Index c) {
    if (r == rows() && c == cols()) return;
    Matrix t; t.st_.resize(r, c);
    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) t.coeffRef(i, j) = (i < rows() && j < cols()) ? coeff(i, j) : S(0);
    *this = t;
  }
  void conservativeResize(Index n) { if (C == 1) conservativeResize(n, 1); else conservativeResize(1, n); }

  static Matrix Zero() { Matrix m; m.setZero(); return m; }
  template <class T> static Matrix Zero(T n) { Matrix m(n); m.setZero(); return m; }
  template <class T0, class T1> static Matrix Zero(T0 r, T1 c) { Matrix m; m.resize((Index)r, (Index)c); m.setZero(); return m; }
  static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
  template <class T0, class T1> static Matrix Identity(T0 r, T1 c) { Matrix m; m.resize((Index)r, (Index)c); m.setIdentity(); return m; }
  template <class T> static Matrix Constant(const T& v) { Matrix m; m.setConstant(v); return m; }
  template <class T0, class T> static Matrix Constant(T0 n, const T& v) { Matrix m((Index)n); m.setConstant(v); return m; }
  template <class T0, class T1, class T> static Matrix Constant(T0 r, T1 c, const T& v) { Matrix m; m.resize((Index)r, (Index)c); m.setConstant(v); return m; }
  template <class T0, class T1> static Matrix LinSpaced(Index n, T0 lo, T1 hi) {
    Matrix m(n);
    for (Index i = 0; i < n; ++i) m(i) = n == 1 ? S(hi) : S(S(lo) + (S(hi) - S(lo)) * S(i) / S(n - 1));
    return m;
  }

  template <int BR, int BC> Block<Matrix, BR, BC> mk_block(Index i, Index j, Index r, Index c) { return Block<Matrix, BR, BC>(this, i, j, r, c); }
  template <int BR, int BC> Block<const Matrix, BR, BC> mk_block(Index i, Index j, Index r, Index c) const { return Block<const Matrix, BR, BC>(this, i, j, r, c); }

 private:
  template <class Ot> void assign(const MatrixBase<Ot>& o) {
    // evaluate through a temporary when the source may alias this matrix (eager evaluation makes that rare)
    const Index r = o.rows(), c = o.cols();
    if ((const void*)&o.derived() == (const void*)this) return;
    std::vector<S> tmp((size_t)(r * c));
    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) tmp[(size_t)(j * r + i)] = S(o.coeff(i, j));
    st_.resize(r, c);
    std::copy(tmp.begin(), tmp.end(), st_.data());
  }
};

template <class M, int BR, int BC>
class Block : public MatrixBase<Block<M, BR, BC>>, public internal::ScalarConv<Block<M, BR, BC>, typename traits<Block<M, BR, BC>>::Scalar, (BR == 1 && BC == 1)> {
  M* m_;
  Index i0_, j0_, r_, c_;
  typedef MatrixBase<Block> Base;

 public:
  typedef typename traits<Block>::Scalar S;
  using Base::operator();
  Block(M* m, Index i0, Index j0, Index r, Index c) : m_(m), i0_(i0), j0_(j0), r_(r), c_(c) {}
  Block(const Block&) = default;
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  S coeff(Index i, Index j) const { return m_->coeff(i0_ + i, j0_ + j); }
  S& coeffRef(Index i, Index j) { return m_->coeffRef(i0_ + i, j0_ + j); }
  template <class Ot> Block& operator=(const MatrixBase<Ot>& o) {
    internal::plain_t<Ot> t(o);   // the source may alias the target's matrix
    for (Index j = 0; j < c_; ++j) for (Index i = 0; i < r_; ++i) coeffRef(i, j) = S(t.coeff(i, j));
    return *this;
  }
  Block& operator=(const Block& o) { return this->template operator=<Block>(o); }
  template <int R2, int C2> Block<M, R2, C2> mk_block(Index i, Index j, Index r, Index c) { return Block<M, R2, C2>(m_, i0_ + i, j0_ + j, r, c); }
  template <int R2, int C2> Block<const M, R2, C2> mk_block(Index i, Index j, Index r, Index c) const { return Block<const M, R2, C2>(m_, i0_ + i, j0_ + j, r, c); }
};

template <class Derived> Matrix<typename traits<Derived>::Scalar, Dynamic, 1> RowwiseOp<Derived>::any() const {
  Matrix<typename traits<Derived>::Scalar, Dynamic, 1> out(m.rows());
  for (Index i = 0; i < m.rows(); ++i) {
    bool a = false;
    for (Index j = 0; j < m.cols(); ++j) a = a || (m.coeff(i, j) != typename traits<Derived>::Scalar(0));
    out.coeffRef(i, 0) = typename traits<Derived>::Scalar(a ? 1 : 0);
  }
  return out;
}

// ---- arithmetic (eager) ----------------------------------------------------------------------------------------
template <class A, class B>
Matrix<typename traits<A>::Scalar, internal::merge_dim(traits<A>::Rows, traits<B>::Rows), internal::merge_dim(traits<A>::Cols, traits<B>::Cols)>
operator+(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  Matrix<typename traits<A>::Scalar, internal::merge_dim(traits<A>::Rows, traits<B>::Rows), internal::merge_dim(traits<A>::Cols, traits<B>::Cols)> r;
  r.resize(a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) + b.coeff(i, j);
  return r;
}
template <class A, class B>
Matrix<typename traits<A>::Scalar, internal::merge_dim(traits<A>::Rows, traits<B>::Rows), internal::merge_dim(traits<A>::Cols, traits<B>::Cols)>
operator-(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  Matrix<typename traits<A>::Scalar, internal::merge_dim(traits<A>::Rows, traits<B>::Rows), internal::merge_dim(traits<A>::Cols, traits<B>::Cols)> r;
  r.resize(a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) - b.coeff(i, j);
  return r;
}
template <class A> internal::plain_t<A> operator-(const MatrixBase<A>& a) {
  internal::plain_t<A> r; r.resize(a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = -a.coeff(i, j);
  return r;
}
template <class A, class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
internal::plain_t<A> operator*(const MatrixBase<A>& a, const T& s) {
  typedef typename traits<A>::Scalar S;
  internal::plain_t<A> r; r.resize(a.rows(), a.cols());
  const S v = S(s);
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) * v;
  return r;
}
template <class A, class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
internal::plain_t<A> operator*(const T& s, const MatrixBase<A>& a) {
  typedef typename traits<A>::Scalar S;
  internal::plain_t<A> r; r.resize(a.rows(), a.cols());
  const S v = S(s);
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = v * a.coeff(i, j);
  return r;
}
template <class A, class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
internal::plain_t<A> operator/(const MatrixBase<A>& a, const T& s) {
  typedef typename traits<A>::Scalar S;
  internal::plain_t<A> r; r.resize(a.rows(), a.cols());
  const S v = S(s);
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) / v;
  return r;
}

namespace internal {
// C (m x n) += A (m x k) * B (k x n), all column-major with leading dimensions; cache-blocked over k and n with the
// inner loop running down a column of C (vectorises under -O3)
template <class S> void gemm_acc(Index m, Index n, Index k, const S* A, Index lda, const S* B, Index ldb, S* C, Index ldc) {
  const Index KB = 256, MB = 512;
  for (Index i0 = 0; i0 < m; i0 += MB) {
    const Index mb = std::min(MB, m - i0);
    for (Index k0 = 0; k0 < k; k0 += KB) {
      const Index kb = std::min(KB, k - k0);
      for (Index j = 0; j < n; ++j) {
        S* c = C + j * ldc + i0;
        const S* b = B + j * ldb + k0;
        Index p = 0;
        for (; p + 4 <= kb; p += 4) {
          const S b0 = b[p], b1 = b[p + 1], b2 = b[p + 2], b3 = b[p + 3];
          const S* a0 = A + (k0 + p) * lda + i0; const S* a1 = a0 + lda; const S* a2 = a1 + lda; const S* a3 = a2 + lda;
          for (Index i = 0; i < mb; ++i) c[i] += a0[i] * b0 + a1[i] * b1 + a2[i] * b2 + a3[i] * b3;
        }
        for (; p < kb; ++p) {
          const S bp = b[p];
          const S* a = A + (k0 + p) * lda + i0;
          for (Index i = 0; i < mb; ++i) c[i] += a[i] * bp;
        }
      }
    }
  }
}
}  // namespace internal

template <class A, class B>
Matrix<typename traits<A>::Scalar, traits<A>::Rows, traits<B>::Cols> operator*(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  typedef typename traits<A>::Scalar S;
  Matrix<S, traits<A>::Rows, traits<B>::Cols> r;
  const Index m = a.rows(), n = b.cols(), k = a.cols();
  r.resize(m, n);
  r.setZero();
  if (m * n * k <= 4096) {
    for (Index j = 0; j < n; ++j) for (Index p = 0; p < k; ++p) { const S bp = b.coeff(p, j); for (Index i = 0; i < m; ++i) r.coeffRef(i, j) += a.coeff(i, p) * bp; }
  } else {
    const internal::dyn_t<S> ae(a), be(b);
    internal::gemm_acc<S>(m, n, k, ae.data(), m, be.data(), k, r.data(), m);
  }
  return r;
}

// ---- dense decompositions ------------------------------------------------------------------------------------
namespace internal {
// Eigen's makeHouseholder on x[0..n): returns beta, tau; x[1..n) overwritten by the essential part
template <class S> void make_householder(S* x, Index n, S& tau, S& beta) {
  using std::sqrt;
  S tail = 0;
  for (Index i = 1; i < n; ++i) tail += x[i] * x[i];
  const S c0 = x[0];
  if (n <= 1 || tail <= (std::numeric_limits<S>::min)()) {
    tau = S(0); beta = c0;
    for (Index i = 1; i < n; ++i) x[i] = S(0);
  } else {
    beta = sqrt(c0 * c0 + tail);
    if (c0 >= S(0)) beta = -beta;
    const S d = c0 - beta;
    for (Index i = 1; i < n; ++i) x[i] /= d;
    tau = (beta - c0) / beta;
  }
}
// apply H = I - tau [1;ess][1;ess]^T on the left of the n x nc block starting at c (column-major, ld)
template <class S> void apply_householder_left(S* c, Index ld, Index n, Index nc, const S* ess, S tau) {
  if (tau == S(0)) return;
  for (Index j = 0; j < nc; ++j) {
    S* col = c + j * ld;
#ifdef MSCKF_REF_SHIM_ALT_ROUNDING   // same arithmetic, dot product summed from the bottom up: a second, equally valid rounding
    S t = 0;
    for (Index i = n - 1; i >= 1; --i) t += ess[i - 1] * col[i];
    t += col[0];
#else
    S t = col[0];
    for (Index i = 1; i < n; ++i) t += ess[i - 1] * col[i];
#endif
    t *= tau;
    col[0] -= t;
    for (Index i = 1; i < n; ++i) col[i] -= t * ess[i - 1];
  }
}
template <class S> struct PartialPivLU {
  dyn_t<S> lu; std::vector<Index> perm; int sign = 1;
  explicit PartialPivLU(const dyn_t<S>& a) : lu(a) {
    using std::abs;
    const Index n = lu.rows();
    perm.resize((size_t)n);
    for (Index k = 0; k < n; ++k) {
      Index piv = k; S best = abs(lu.coeff(k, k));
      for (Index i = k + 1; i < n; ++i) if (abs(lu.coeff(i, k)) > best) { best = abs(lu.coeff(i, k)); piv = i; }
      perm[(size_t)k] = piv;
      if (piv != k) { sign = -sign; for (Index j = 0; j < n; ++j) std::swap(lu.coeffRef(k, j), lu.coeffRef(piv, j)); }
      const S d = lu.coeff(k, k);
      if (d != S(0)) for (Index i = k + 1; i < n; ++i) lu.coeffRef(i, k) /= d;
      for (Index j = k + 1; j < n; ++j) {
        const S u = lu.coeff(k, j);
        if (u == S(0)) continue;
        S* cj = lu.data() + j * n; const S* ck = lu.data() + k * n;
        for (Index i = k + 1; i < n; ++i) cj[i] -= ck[i] * u;
      }
    }
  }
  S determinant() const { S d = S(sign); for (Index i = 0; i < lu.rows(); ++i) d *= lu.coeff(i, i); return d; }
  dyn_t<S> solve(const dyn_t<S>& b) const {
    const Index n = lu.rows(), nc = b.cols();
    dyn_t<S> x(b);
    for (Index k = 0; k < n; ++k) if (perm[(size_t)k] != k) for (Index j = 0; j < nc; ++j) std::swap(x.coeffRef(k, j), x.coeffRef(perm[(size_t)k], j));
    for (Index j = 0; j < nc; ++j) {
      S* xj = x.data() + j * n;
      for (Index k = 0; k < n; ++k) { const S v = xj[k]; if (v == S(0)) continue; const S* ck = lu.data() + k * n; for (Index i = k + 1; i < n; ++i) xj[i] -= ck[i] * v; }
      for (Index k = n - 1; k >= 0; --k) { xj[k] /= lu.coeff(k, k); const S v = xj[k]; const S* ck = lu.data() + k * n; for (Index i = 0; i < k; ++i) xj[i] -= ck[i] * v; }
    }
    return x;
  }
};
}  // namespace internal

template <class Derived> typename MatrixBase<Derived>::S MatrixBase<Derived>::determinant() const {
  const Index n = rows();
  if (n == 1) return coeff(0, 0);
  if (n == 2) return coeff(0, 0) * coeff(1, 1) - coeff(0, 1) * coeff(1, 0);
  if (n == 3)
    return coeff(0, 0) * (coeff(1, 1) * coeff(2, 2) - coeff(1, 2) * coeff(2, 1)) - coeff(0, 1) * (coeff(1, 0) * coeff(2, 2) - coeff(1, 2) * coeff(2, 0)) +
           coeff(0, 2) * (coeff(1, 0) * coeff(2, 1) - coeff(1, 1) * coeff(2, 0));
  return internal::PartialPivLU<S>(internal::dyn_t<S>(*this)).determinant();
}
template <class Derived> typename MatrixBase<Derived>::Plain MatrixBase<Derived>::inverse() const {
  const Index n = rows();
  Plain r; r.resize(n, n);
  if (n == 1) { r.coeffRef(0, 0) = S(1) / coeff(0, 0); return r; }
  if (n == 2) {
    const S id = S(1) / determinant();
    r.coeffRef(0, 0) = coeff(1, 1) * id; r.coeffRef(1, 0) = -coeff(1, 0) * id; r.coeffRef(0, 1) = -coeff(0, 1) * id; r.coeffRef(1, 1) = coeff(0, 0) * id;
    return r;
  }
  if (n == 3) {
    auto cof = [&](int i, int j) { const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return coeff(i1, j1) * coeff(i2, j2) - coeff(i1, j2) * coeff(i2, j1); };
    const S c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const S id = S(1) / (c00 * coeff(0, 0) + c10 * coeff(1, 0) + c20 * coeff(2, 0));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.coeffRef(j, i) = cof(i, j) * id;
    return r;
  }
  r = internal::PartialPivLU<S>(internal::dyn_t<S>(*this)).solve(internal::dyn_t<S>::Identity(n, n));
  return r;
}

// Higham (2005) scaling and squaring, degrees and switch points of Eigen's unsupported MatrixExponential.h
template <class Derived> typename MatrixBase<Derived>::Plain MatrixBase<Derived>::exp() const {
  typedef internal::dyn_t<S> M;
  const Index n = rows();
  M A(*this);
  double l1 = 0;
  for (Index j = 0; j < n; ++j) { double s = 0; for (Index i = 0; i < n; ++i) s += std::fabs((double)A.coeff(i, j)); l1 = std::max(l1, s); }
  const M I = M::Identity(n, n);
  int squarings = 0, deg;
  if (sizeof(S) == 4) {
    if (l1 < 4.258730016922831e-001) deg = 3;
    else if (l1 < 1.880152677804762e+000) deg = 5;
    else { deg = 7; std::frexp(l1 / 3.925724783138660, &squarings); if (squarings < 0) squarings = 0; }
  } else {
    if (l1 < 1.495585217958292e-002) deg = 3;
    else if (l1 < 2.539398330063230e-001) deg = 5;
    else if (l1 < 9.504178996162932e-001) deg = 7;
    else if (l1 < 2.097847961257068e+000) deg = 9;
    else { deg = 13; std::frexp(l1 / 5.371920351148152, &squarings); if (squarings < 0) squarings = 0; }
  }
  if (squarings > 0) A = A * S(std::ldexp(1.0, -squarings));
  M U, V;
  const M A2 = A * A;
  if (deg == 3) {
    const S b[] = {120, 60, 12, 1};
    const M tmp = b[3] * A2 + b[1] * I;
    U = A * tmp; V = b[2] * A2 + b[0] * I;
  } else if (deg == 5) {
    const S b[] = {30240, 15120, 3360, 420, 30, 1};
    const M A4 = A2 * A2;
    const M tmp = b[5] * A4 + b[3] * A2 + b[1] * I;
    U = A * tmp; V = b[4] * A4 + b[2] * A2 + b[0] * I;
  } else if (deg == 7) {
    const S b[] = {17297280, 8648640, 1995840, 277200, 25200, 1512, 56, 1};
    const M A4 = A2 * A2, A6 = A4 * A2;
    const M tmp = b[7] * A6 + b[5] * A4 + b[3] * A2 + b[1] * I;
    U = A * tmp; V = b[6] * A6 + b[4] * A4 + b[2] * A2 + b[0] * I;
  } else if (deg == 9) {
    const S b[] = {S(17643225600.), S(8821612800.), S(2075673600.), 302702400, 30270240, 2162160, 110880, 3960, 90, 1};
    const M A4 = A2 * A2, A6 = A4 * A2, A8 = A6 * A2;
    const M tmp = b[9] * A8 + b[7] * A6 + b[5] * A4 + b[3] * A2 + b[1] * I;
    U = A * tmp; V = b[8] * A8 + b[6] * A6 + b[4] * A4 + b[2] * A2 + b[0] * I;
  } else {
    const S b[] = {S(64764752532480000.), S(32382376266240000.), S(7771770303897600.), S(1187353796428800.), S(129060195264000.),
                   S(10559470521600.), S(670442572800.), S(33522128640.), S(1323241920.), 40840800, 960960, 16380, 182, 1};
    const M A4 = A2 * A2, A6 = A4 * A2;
    V = b[13] * A6 + b[11] * A4 + b[9] * A2;
    M tmp = A6 * V;
    tmp += b[7] * A6 + b[5] * A4 + b[3] * A2 + b[1] * I;
    U = A * tmp;
    tmp = b[12] * A6 + b[10] * A4 + b[8] * A2;
    V = A6 * tmp;
    V += b[6] * A6 + b[4] * A4 + b[2] * A2 + b[0] * I;
  }
  const M numer = U + V, denom = V - U;
  M res = internal::PartialPivLU<S>(denom).solve(numer);
  for (int i = 0; i < squarings; ++i) res = res * res;
  Plain out; out = res;
  return out;
}

// LDL^T with diagonal pivoting (largest remaining |diagonal| first), as Eigen's LDLT
template <class MatrixType> class LDLT {
  typedef typename traits<MatrixType>::Scalar S;
  internal::dyn_t<S> m_;
  std::vector<Index> tr_;

 public:
  template <class D> explicit LDLT(const MatrixBase<D>& a) : m_(a) {
    using std::abs;
    const Index n = m_.rows();
    tr_.resize((size_t)n);
    for (Index k = 0; k < n; ++k) {
      Index piv = k; S best = abs(m_.coeff(k, k));
      for (Index i = k + 1; i < n; ++i) if (abs(m_.coeff(i, i)) > best) { best = abs(m_.coeff(i, i)); piv = i; }
      tr_[(size_t)k] = piv;
      if (piv != k) {   // symmetric swap of rows/columns k and piv of the lower triangle
        for (Index j = 0; j < k; ++j) std::swap(m_.coeffRef(k, j), m_.coeffRef(piv, j));
        for (Index i = piv + 1; i < n; ++i) std::swap(m_.coeffRef(i, k), m_.coeffRef(i, piv));
        std::swap(m_.coeffRef(k, k), m_.coeffRef(piv, piv));
        for (Index i = k + 1; i < piv; ++i) std::swap(m_.coeffRef(i, k), m_.coeffRef(piv, i));
      }
      // d_k = a_kk - sum_j l_kj^2 d_j ; column k of L below the diagonal
      for (Index j = 0; j < k; ++j) m_.coeffRef(k, k) -= m_.coeff(k, j) * m_.coeff(k, j) * m_.coeff(j, j);
      const S d = m_.coeff(k, k);
      for (Index i = k + 1; i < n; ++i) {
        S v = m_.coeff(i, k);
        for (Index j = 0; j < k; ++j) v -= m_.coeff(i, j) * m_.coeff(k, j) * m_.coeff(j, j);
        m_.coeffRef(i, k) = (d != S(0)) ? v / d : S(0);
      }
    }
  }
  template <class B> internal::plain_t<B> solve(const MatrixBase<B>& b) const {
    using std::abs;
    const Index n = m_.rows(), nc = b.cols();
    internal::dyn_t<S> x(b);
    for (Index k = 0; k < n; ++k) if (tr_[(size_t)k] != k) for (Index j = 0; j < nc; ++j) std::swap(x.coeffRef(k, j), x.coeffRef(tr_[(size_t)k], j));
    S dmax = 0;
    for (Index i = 0; i < n; ++i) dmax = std::max(dmax, abs(m_.coeff(i, i)));
    const S tol = (std::numeric_limits<S>::min)();
    for (Index j = 0; j < nc; ++j) {
      for (Index k = 0; k < n; ++k) { const S v = x.coeff(k, j); for (Index i = k + 1; i < n; ++i) x.coeffRef(i, j) -= m_.coeff(i, k) * v; }
      for (Index k = 0; k < n; ++k) { const S d = m_.coeff(k, k); x.coeffRef(k, j) = abs(d) > tol ? x.coeff(k, j) / d : S(0); }
      for (Index k = n - 1; k >= 0; --k) { S v = x.coeff(k, j); for (Index i = k + 1; i < n; ++i) v -= m_.coeff(i, k) * x.coeff(i, j); x.coeffRef(k, j) = v; }
    }
    for (Index k = n - 1; k >= 0; --k) if (tr_[(size_t)k] != k) for (Index j = 0; j < nc; ++j) std::swap(x.coeffRef(k, j), x.coeffRef(tr_[(size_t)k], j));
    internal::plain_t<B> out; out = x;
    return out;
  }
};
template <class Derived> LDLT<typename MatrixBase<Derived>::Plain> MatrixBase<Derived>::ldlt() const { return LDLT<Plain>(*this); }

// Unpivoted Householder QR, column by column (Eigen's unblocked kernel; its blocked variant applies the same
// reflectors panel-wise and differs by rounding only).
template <class MatrixType> class HouseholderQR {
  typedef typename traits<MatrixType>::Scalar S;
  internal::dyn_t<S> qr_;
  std::vector<S> tau_;

 public:
  template <class D> explicit HouseholderQR(const MatrixBase<D>& a) : qr_(a) {
    const Index m = qr_.rows(), n = qr_.cols(), size = std::min(m, n);
    tau_.assign((size_t)size, S(0));
    for (Index k = 0; k < size; ++k) {
      S* ck = qr_.data() + k * m + k;
      S tau, beta;
      internal::make_householder(ck, m - k, tau, beta);
      ck[0] = beta;
      tau_[(size_t)k] = tau;
      internal::apply_householder_left(qr_.data() + (k + 1) * m + k, m, m - k, n - k - 1, ck + 1, tau);
    }
  }
  const internal::dyn_t<S>& matrixQR() const { return qr_; }
  // Q = H_0 H_1 ... H_{size-1} as a dense rows x rows matrix (what `MatrixX Q = qr.householderQ()` evaluates)
  internal::dyn_t<S> householderQ() const {
    const Index m = qr_.rows(), size = (Index)tau_.size();
    internal::dyn_t<S> Q = internal::dyn_t<S>::Identity(m, m);
    for (Index k = size - 1; k >= 0; --k)
      internal::apply_householder_left(Q.data() + k * m + k, m, m - k, m - k, qr_.data() + k * m + k + 1, tau_[(size_t)k]);
    return Q;
  }
};

// JacobiSVD of a tall (rows >= cols) matrix with Eigen's default column-pivoting QR preconditioner.
template <class MatrixType> class JacobiSVD {
  typedef typename traits<MatrixType>::Scalar S;
  internal::dyn_t<S> U_, V_;
  Matrix<S, Dynamic, 1> sv_;

 public:
  template <class D> explicit JacobiSVD(const MatrixBase<D>& a, unsigned options = 0) {
    using std::abs; using std::sqrt;
    internal::dyn_t<S> A(a);
    const Index m = A.rows(), n = A.cols();
    const bool tall = m >= n;
    if (!tall) A = internal::dyn_t<S>(a.transpose());
    const Index mm = A.rows(), nn = A.cols();
    // scale as JacobiSVD::compute does (leaves the reflectors unchanged up to rounding)
    S scale = 0;
    for (Index j = 0; j < nn; ++j) for (Index i = 0; i < mm; ++i) scale = std::max(scale, abs(A.coeff(i, j)));
    if (scale == S(0)) scale = S(1);
    A = A / scale;
    // column-pivoted Householder QR
    std::vector<Index> perm((size_t)nn);
    for (Index j = 0; j < nn; ++j) perm[(size_t)j] = j;
    std::vector<S> tau((size_t)nn, S(0));
    for (Index k = 0; k < nn; ++k) {
      Index big = k; S best = -1;
      for (Index j = k; j < nn; ++j) { S s = 0; for (Index i = k; i < mm; ++i) s += A.coeff(i, j) * A.coeff(i, j); if (s > best) { best = s; big = j; } }
      if (big != k) { for (Index i = 0; i < mm; ++i) std::swap(A.coeffRef(i, k), A.coeffRef(i, big)); std::swap(perm[(size_t)k], perm[(size_t)big]); }
      S* ck = A.data() + k * mm + k;
      S t, beta;
      internal::make_householder(ck, mm - k, t, beta);
      ck[0] = beta; tau[(size_t)k] = t;
      internal::apply_householder_left(A.data() + (k + 1) * mm + k, mm, mm - k, nn - k - 1, ck + 1, t);
    }
    internal::dyn_t<S> Q = internal::dyn_t<S>::Identity(mm, mm);
    for (Index k = nn - 1; k >= 0; --k) internal::apply_householder_left(Q.data() + k * mm + k, mm, mm - k, mm - k, A.data() + k * mm + k + 1, tau[(size_t)k]);
    // one-sided Jacobi on W = R (nn x nn): W <- W J, Vs <- Vs J until the columns are orthogonal
    internal::dyn_t<S> W = internal::dyn_t<S>::Zero(nn, nn), Vs = internal::dyn_t<S>::Identity(nn, nn);
    for (Index j = 0; j < nn; ++j) for (Index i = 0; i <= j; ++i) W.coeffRef(i, j) = A.coeff(i, j);
    for (int sweep = 0; sweep < 60; ++sweep) {
      bool rotated = false;
      for (Index p = 0; p < nn; ++p) for (Index q = p + 1; q < nn; ++q) {
        S app = 0, aqq = 0, apq = 0;
        for (Index i = 0; i < nn; ++i) { app += W.coeff(i, p) * W.coeff(i, p); aqq += W.coeff(i, q) * W.coeff(i, q); apq += W.coeff(i, p) * W.coeff(i, q); }
        if (abs(apq) <= std::numeric_limits<S>::epsilon() * sqrt(app * aqq) || apq == S(0)) continue;
        rotated = true;
        const S zeta = (aqq - app) / (S(2) * apq);
        const S t = (zeta >= 0 ? S(1) : S(-1)) / (abs(zeta) + sqrt(S(1) + zeta * zeta));
        const S c = S(1) / sqrt(S(1) + t * t), s = c * t;
        for (Index i = 0; i < nn; ++i) {
          const S wp = W.coeff(i, p), wq = W.coeff(i, q);
          W.coeffRef(i, p) = c * wp - s * wq; W.coeffRef(i, q) = s * wp + c * wq;
          const S vp = Vs.coeff(i, p), vq = Vs.coeff(i, q);
          Vs.coeffRef(i, p) = c * vp - s * vq; Vs.coeffRef(i, q) = s * vp + c * vq;
        }
      }
      if (!rotated) break;
    }
    std::vector<S> sig((size_t)nn);
    internal::dyn_t<S> Us = internal::dyn_t<S>::Identity(nn, nn);
    for (Index j = 0; j < nn; ++j) {
      S s = 0; for (Index i = 0; i < nn; ++i) s += W.coeff(i, j) * W.coeff(i, j);
      sig[(size_t)j] = sqrt(s);
      if (sig[(size_t)j] > S(0)) for (Index i = 0; i < nn; ++i) Us.coeffRef(i, j) = W.coeff(i, j) / sig[(size_t)j];
    }
    std::vector<Index> ord((size_t)nn);
    for (Index j = 0; j < nn; ++j) ord[(size_t)j] = j;
    std::stable_sort(ord.begin(), ord.end(), [&](Index x, Index y) { return sig[(size_t)x] > sig[(size_t)y]; });
    internal::dyn_t<S> Ufull(Q), Vfull = internal::dyn_t<S>::Zero(nn, nn);
    sv_.resize(nn);
    for (Index jj = 0; jj < nn; ++jj) {
      const Index j = ord[(size_t)jj];
      sv_(jj) = sig[(size_t)j] * scale;
      for (Index i = 0; i < mm; ++i) { S s = 0; for (Index p = 0; p < nn; ++p) s += Q.coeff(i, p) * Us.coeff(p, j); Ufull.coeffRef(i, jj) = s; }
      for (Index p = 0; p < nn; ++p) Vfull.coeffRef(perm[(size_t)p], jj) = Vs.coeff(p, j);
    }
    (void)options;
    if (tall) { U_ = Ufull; V_ = Vfull; } else { U_ = Vfull; V_ = Ufull; }
  }
  const internal::dyn_t<S>& matrixU() const { return U_; }
  const internal::dyn_t<S>& matrixV() const { return V_; }
  const Matrix<S, Dynamic, 1>& singularValues() const { return sv_; }
};

// ---- Geometry ------------------------------------------------------------------------------------------------
template <class S, int O> class Quaternion {
  S c_[4];   // x y z w

 public:
  typedef S Scalar;
  Quaternion() {}
  Quaternion(const S& w, const S& x, const S& y, const S& z) { c_[0] = x; c_[1] = y; c_[2] = z; c_[3] = w; }
  template <class D> explicit Quaternion(const MatrixBase<D>& m) {
    using std::sqrt;
    if (m.rows() == 4 && m.cols() == 1) { for (int i = 0; i < 4; ++i) c_[i] = m.coeff(i, 0); return; }
    S t = m.coeff(0, 0) + m.coeff(1, 1) + m.coeff(2, 2);   // rotation matrix -> quaternion (Eigen's branches)
    if (t > S(0)) {
      t = sqrt(t + S(1.0)); w() = S(0.5) * t; t = S(0.5) / t;
      x() = (m.coeff(2, 1) - m.coeff(1, 2)) * t; y() = (m.coeff(0, 2) - m.coeff(2, 0)) * t; z() = (m.coeff(1, 0) - m.coeff(0, 1)) * t;
    } else {
      int i = 0;
      if (m.coeff(1, 1) > m.coeff(0, 0)) i = 1;
      if (m.coeff(2, 2) > m.coeff(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = sqrt(m.coeff(i, i) - m.coeff(j, j) - m.coeff(k, k) + S(1.0));
      c_[i] = S(0.5) * t; t = S(0.5) / t;
      w() = (m.coeff(k, j) - m.coeff(j, k)) * t; c_[j] = (m.coeff(j, i) + m.coeff(i, j)) * t; c_[k] = (m.coeff(k, i) + m.coeff(i, k)) * t;
    }
  }
  static Quaternion Identity() { return Quaternion(S(1), S(0), S(0), S(0)); }
  Quaternion& setIdentity() { *this = Identity(); return *this; }
  S& x() { return c_[0]; } S& y() { return c_[1]; } S& z() { return c_[2]; } S& w() { return c_[3]; }
  S x() const { return c_[0]; } S y() const { return c_[1]; } S z() const { return c_[2]; } S w() const { return c_[3]; }
  Matrix<S, 4, 1> coeffs() const { return Matrix<S, 4, 1>(c_[0], c_[1], c_[2], c_[3]); }
  Matrix<S, 3, 1> vec() const { return Matrix<S, 3, 1>(c_[0], c_[1], c_[2]); }
  S squaredNorm() const { return c_[0] * c_[0] + c_[1] * c_[1] + c_[2] * c_[2] + c_[3] * c_[3]; }
  S norm() const { using std::sqrt; return sqrt(squaredNorm()); }
  void normalize() { const S n = norm(); for (int i = 0; i < 4; ++i) c_[i] /= n; }
  Quaternion normalized() const { Quaternion q(*this); q.normalize(); return q; }
  Quaternion conjugate() const { return Quaternion(w(), -x(), -y(), -z()); }
  Quaternion inverse() const {
    const S n2 = squaredNorm();
    if (n2 > S(0)) return Quaternion(w() / n2, -x() / n2, -y() / n2, -z() / n2);
    return Quaternion(S(0), S(0), S(0), S(0));
  }
  Quaternion operator*(const Quaternion& b) const {
    const Quaternion& a = *this;
    return Quaternion(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(), a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                      a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(), a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
  }
  Quaternion& operator*=(const Quaternion& b) { *this = *this * b; return *this; }
  // rotate a vector: v + w*2(u x v) + u x 2(u x v), Eigen's _transformVector
  template <class D> Matrix<S, 3, 1> operator*(const MatrixBase<D>& v) const {
    const Matrix<S, 3, 1> u = vec(), vv(v);
    Matrix<S, 3, 1> uv = u.cross(vv);
    uv += uv;
    return vv + w() * uv + u.cross(uv);
  }
  Matrix<S, 3, 3> toRotationMatrix() const {
    Matrix<S, 3, 3> r;
    const S tx = S(2) * x(), ty = S(2) * y(), tz = S(2) * z();
    const S twx = tx * w(), twy = ty * w(), twz = tz * w(), txx = tx * x(), txy = ty * x(), txz = tz * x(), tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
    r(0, 0) = S(1) - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
    r(1, 0) = txy + twz; r(1, 1) = S(1) - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = S(1) - (txx + tyy);
    return r;
  }
  S dot(const Quaternion& o) const { return c_[0] * o.c_[0] + c_[1] * o.c_[1] + c_[2] * o.c_[2] + c_[3] * o.c_[3]; }
  S angularDistance(const Quaternion& o) const {
    using std::atan2; using std::abs;
    const Quaternion d = (*this) * o.conjugate();
    return S(2) * atan2(d.vec().norm(), abs(d.w()));
  }
};
typedef Quaternion<float> Quaternionf;
typedef Quaternion<double> Quaterniond;

template <class S, int Dim, int Mode, int O = 0> class Transform {
  Matrix<S, Dim, Dim> lin_;
  Matrix<S, Dim, 1> t_;

 public:
  Transform() {}
  static Transform Identity() { Transform T; T.lin_.setIdentity(); T.t_.setZero(); return T; }
  Matrix<S, Dim, Dim>& linear() { return lin_; }
  const Matrix<S, Dim, Dim>& linear() const { return lin_; }
  Matrix<S, Dim, Dim>& rotation() { return lin_; }
  const Matrix<S, Dim, Dim>& rotation() const { return lin_; }
  Matrix<S, Dim, 1>& translation() { return t_; }
  const Matrix<S, Dim, 1>& translation() const { return t_; }
  Transform inverse() const {
    Transform r;
    if (Mode == Isometry) r.lin_ = lin_.transpose(); else r.lin_ = lin_.inverse();
    r.t_ = -(r.lin_ * t_);
    return r;
  }
  Transform operator*(const Transform& o) const { Transform r; r.lin_ = lin_ * o.lin_; r.t_ = lin_ * o.t_ + t_; return r; }
  template <class D> Matrix<S, Dim, 1> operator*(const MatrixBase<D>& v) const { return lin_ * v + t_; }
};

// ---- the usual typedefs ----------------------------------------------------------------------------------------
template <class S, int R, int C, int O = 0, int MR = R, int MC = C> using Array = Matrix<S, R, C, O, MR, MC>;
#define MSCKF_REF_SHIM_TYPEDEFS(T, sfx)                      \
  typedef Matrix<T, 2, 2> Matrix2##sfx;                      \
  typedef Matrix<T, 3, 3> Matrix3##sfx;                      \
  typedef Matrix<T, 4, 4> Matrix4##sfx;                      \
  typedef Matrix<T, Dynamic, Dynamic> MatrixX##sfx;          \
  typedef Matrix<T, 2, 1> Vector2##sfx;                      \
  typedef Matrix<T, 3, 1> Vector3##sfx;                      \
  typedef Matrix<T, 4, 1> Vector4##sfx;                      \
  typedef Matrix<T, Dynamic, 1> VectorX##sfx;                \
  typedef Matrix<T, 1, 3> RowVector3##sfx;                   \
  typedef Matrix<T, 1, Dynamic> RowVectorX##sfx;
MSCKF_REF_SHIM_TYPEDEFS(float, f)
MSCKF_REF_SHIM_TYPEDEFS(double, d)
MSCKF_REF_SHIM_TYPEDEFS(int, i)
#undef MSCKF_REF_SHIM_TYPEDEFS
typedef Transform<float, 3, Isometry> Isometry3f;
typedef Transform<double, 3, Isometry> Isometry3d;

}  // namespace Eigen
#endif
