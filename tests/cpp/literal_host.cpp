// literal_host.cpp -- TEST HARNESS: msckf_mono_amd/csrc/literal_core.h (the device's literal anisotropic compression, written
// for two compilers) compiled for the host with -DLIT_HOST, so that its algorithm can be held against the oracle on the
// CPU (tests/test_literal_core.py).  The product runs the same header inside kernels_literal.hip.
#include <cstdlib>
#include <vector>

#define LIT_HOST
#include "../../msckf_mono_amd/csrc/literal_core.h"

// route: 0 the compact route; 1 the sweep over the dense stack.  LamIn: [H_o | r_o]^T [H_o | r_o]
// ((6N+1)^2, element (hi, lo) at hi * (6N+1) + lo; what k_gram accumulates on the device), may be null for route 1.
extern "C" int lit_host_compress(int F, int m_cap, int N, const int* included, const int* M, const int* slots /*[F][m_cap]*/,
                                 const double* Hx /*[F][m_cap][12]*/, const double* rw /*[F][2 m_cap]*/, double u_var, double v_var,
                                 double tol, int route, const double* LamIn, double* Lam /*[(6N+1)^2], (hi, lo) at hi * (6N+1) + lo*/,
                                 int* info8, double* TH_out /*[(6N+15) x (6N+1)] column-major or null*/) {
  using namespace msckf::lit;
  const int n = 6 * N;
  int m = 0, mobs = 0;
  for (int t = 0; t < F; ++t) if (included[t]) { m += 2 * M[t] - 3; mobs += M[t]; }
  Args<double> a;
  a.F = F; a.m_cap = m_cap; a.N = N; a.status = included; a.inc_bit = 1; a.M = M; a.slots = slots; a.off = nullptr;
  a.Hx = Hx; a.rw = rw; a.u_var = u_var; a.v_var = v_var; a.tol = tol;
  a.ldx = m + 8;
  std::vector<double> X((size_t)a.ldx * (n + 1)), tau(2 * (n + 1) + 2), Vf((size_t)F * 2 * m_cap * 3), Tf((size_t)F * 9);
  std::vector<int> row0(F + 1), obs0(F + 1), kept(6 * (n + 16) + 64), otrk(mobs + 8);
  a.X = X.data(); a.tau = tau.data(); a.Vf = Vf.data(); a.Tf = Tf.data(); a.row0 = row0.data(); a.obs0 = obs0.data(); a.kept = kept.data();
  a.otrk = otrk.data();
  a.r_cap = n + 15;
  std::vector<double> TH((size_t)a.r_cap * (n + 1)), G((size_t)(mobs + 8) * a.r_cap);
  a.TH = TH.data(); a.ldg = mobs + 8; a.G = G.data();
  a.ldz = a.r_cap + n + 1;
  std::vector<double> Z((size_t)a.ldz * a.ldz);
  a.Z = Z.data();
  std::vector<double> W2((size_t)compact_ws_doubles(n, m_cap, a.r_cap, a.ldg));
  a.W2 = W2.data();
  a.LamIn = LamIn; a.lam_part = 0; a.gram_parts = 1;
  a.Lam = Lam; a.ldL = n + 1; a.info = info8;
  // Gam = sum over the stacked tracks of (u-rows of the projected Jacobian)^T (the same): on the device k_lit_pre writes six
  // rows per track (a wavefront per track) and k_lit_gamma multiplies them on the matrix cores; here the serial reference of the
  // rows (literal_core.h: gamma_rows, after the tracks' null spaces) and plain loops
  std::vector<double> Gam((size_t)(n + 1) * (n + 1), 0.0);
  a.Gam = Gam.data(); a.ldGam = n + 1;
  if (route != 1) {
    std::vector<double> rows((size_t)6 * (n + 8));
    const long ldc = n + 8;
    for (int t = 0; t < F; ++t) {
      if (!included[t]) continue;
      track_null_space(a, t);
      std::fill(rows.begin(), rows.end(), 0.0);
      gamma_rows(a, t, rows.data(), ldc);
      for (int o = 0; o < M[t]; ++o) {                       // Du: u-row of the observation's 2 x 6 block, squared
        const int col = 6 * slots[(size_t)t * m_cap + o];
        for (int x = 0; x < 6; ++x) for (int y = 0; y <= x; ++y) Gam[(size_t)(col + x) * (n + 1) + col + y] += Hx[((size_t)t * m_cap + o) * 12 + x] * Hx[((size_t)t * m_cap + o) * 12 + y];
      }
      for (int hi = 0; hi < n; ++hi)
        for (int lo = 0; lo <= hi; ++lo) {
          double sacc = 0;
          for (int x = 0; x < 3; ++x) sacc += rows[x * ldc + hi] * rows[(3 + x) * ldc + lo] + rows[(3 + x) * ldc + hi] * rows[x * ldc + lo];
          Gam[(size_t)hi * (n + 1) + lo] -= sacc;
        }
    }
  }
  Ctx c;
  // what the device has as LDS (kernels_literal.hip: LIT_LDS_DOUBLES); LIT_HOST_STAGE overrides the size (tests: the blocked
  // elimination must give the same matrix with narrower panels and with none)
  const char* se = getenv("LIT_HOST_STAGE");
  std::vector<double> stage((size_t)(se ? atol(se) : 12000));
  c.lds = stage.data(); c.lds_doubles = (int)stage.size();
  literal_compress(c, a, route);
  if (TH_out) for (size_t i = 0; i < (size_t)(n + 15) * (n + 1); ++i) { const size_t col = i / (n + 15), row = i % (n + 15); TH_out[i] = TH[row + (size_t)a.r_cap * col]; }
  return 0;
}
