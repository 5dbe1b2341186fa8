// TEST INFRASTRUCTURE -- CPU oracle (see problem.h for what this restates).
#include "problem.h"
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <typeinfo>
#include <algorithm>

namespace orc {

Problem::~Problem() {
  std::map<void*, int> seen;
  for (auto& rb : residuals_) {
    if (rb.cost && !seen[rb.cost]++) delete rb.cost;
    if (rb.loss && !seen[rb.loss]++) delete rb.loss;
  }
  for (auto& pb : params_)
    if (pb.lp && !seen[pb.lp]++) delete pb.lp;
}

void Problem::AddParameterBlock(double* values, int size, LocalParameterization* lp) {
  auto it = index_.find(values);
  if (it != index_.end()) {
    if (lp) params_[it->second].lp = lp;
    return;
  }
  ParamBlock pb;
  pb.user = values;
  pb.size = size;
  pb.lp = lp;
  index_[values] = (int)params_.size();
  params_.push_back(pb);
}
void Problem::SetParameterBlockConstant(double* values) { params_[index_.at(values)].constant = true; }
void Problem::AddResidualBlock(CostFunction* cost, LossFunction* loss, const std::vector<double*>& blocks) {
  ResidualBlock rb;
  rb.cost = cost;
  rb.loss = loss;
  const std::vector<int>& sizes = cost->parameter_block_sizes();
  for (size_t i = 0; i < blocks.size(); ++i) {
    if (!index_.count(blocks[i])) AddParameterBlock(blocks[i], sizes[i], nullptr);
    rb.blocks.push_back(index_.at(blocks[i]));
  }
  residuals_.push_back(rb);
}

namespace {

// One Jacobian block of a residual block, in LOCAL coordinates, row-major rows x cols.
struct JBlock {
  int active;  // index into Program::active
  int cols;
  std::vector<double> v;
};
struct RowBlock {
  int row0, rows;
  std::vector<JBlock> blocks;
};

// The reduced program (ceres: constant blocks removed, e-blocks ordered first).
struct Program {
  Problem* p;
  std::vector<int> active;      // -> problem param index
  std::vector<int> col0;        // local column offset
  std::vector<int> st0;         // ambient (global) offset in the state vector
  std::vector<int> act_of;      // problem param index -> active index or -1
  int num_e = 0, e_cols = 0, num_cols = 0, num_state = 0, num_rows = 0;
  std::vector<int> row0;        // per residual block

  explicit Program(Problem* prob) : p(prob) {
    const int np = (int)p->params_.size();
    std::vector<char> used(np, 0);
    for (auto& rb : p->residuals_)
      for (int b : rb.blocks) used[b] = 1;
    act_of.assign(np, -1);
    // e-blocks: the landmark blocks (inverse depth, size 1; line orth, size 4).  Any valid
    // ceres Schur ordering gives the same step up to rounding; see DESIGN.md.
    for (int pass = 0; pass < 2; ++pass)
      for (int i = 0; i < np; ++i) {
        auto& pb = p->params_[i];
        if (!used[i] || pb.constant) continue;
        bool is_e = pb.local_size() <= 4;
        if ((pass == 0) != is_e) continue;
        act_of[i] = (int)active.size();
        active.push_back(i);
        col0.push_back(num_cols);
        st0.push_back(num_state);
        num_cols += pb.local_size();
        num_state += pb.size;
        if (is_e) { ++num_e; e_cols = num_cols; }
      }
    for (auto& rb : p->residuals_) {
      row0.push_back(num_rows);
      num_rows += rb.cost->num_residuals();
    }
  }
  int lsize(int a) const { return p->params_[active[a]].local_size(); }
  int gsize(int a) const { return p->params_[active[a]].size; }

  void StateFromUser(VecX& x) const {
    x.assign(num_state, 0.0);
    for (size_t a = 0; a < active.size(); ++a)
      std::memcpy(&x[st0[a]], p->params_[active[a]].user, sizeof(double) * gsize((int)a));
  }
  void StateToUser(const VecX& x) const {
    for (size_t a = 0; a < active.size(); ++a)
      std::memcpy(p->params_[active[a]].user, &x[st0[a]], sizeof(double) * gsize((int)a));
  }
  bool Plus(const VecX& x, const VecX& delta, VecX& out) const {
    out.assign(num_state, 0.0);
    for (size_t a = 0; a < active.size(); ++a) {
      auto& pb = p->params_[active[a]];
      if (pb.lp) {
        if (!pb.lp->Plus(&x[st0[a]], &delta[col0[a]], &out[st0[a]])) return false;
      } else {
        for (int k = 0; k < pb.size; ++k) out[st0[a] + k] = x[st0[a] + k] + delta[col0[a] + k];
      }
    }
    for (double v : out)
      if (!std::isfinite(v)) return false;
    return true;
  }

  // ceres ProgramEvaluator::Evaluate + ResidualBlock::Evaluate
  bool Evaluate(const VecX& x, double* cost, VecX* residuals, VecX* gradient, std::vector<RowBlock>* jac) const {
    *cost = 0.0;
    if (residuals) residuals->assign(num_rows, 0.0);
    if (gradient) gradient->assign(num_cols, 0.0);
    if (jac) jac->assign(p->residuals_.size(), RowBlock());
    std::vector<double> r, gj[8], lpj;
    for (size_t i = 0; i < p->residuals_.size(); ++i) {
      auto& rb = p->residuals_[i];
      const int nres = rb.cost->num_residuals();
      const int nb = (int)rb.blocks.size();
      std::vector<const double*> params(nb);
      std::vector<double*> jptr(nb, nullptr);
      std::vector<std::vector<double>> gjac(nb);
      for (int b = 0; b < nb; ++b) {
        int pi = rb.blocks[b];
        int a = act_of[pi];
        params[b] = a >= 0 ? &x[st0[a]] : p->params_[pi].user;
        if (jac && a >= 0) {
          gjac[b].assign((size_t)nres * p->params_[pi].size, 0.0);
          jptr[b] = gjac[b].data();
        }
      }
      r.assign(nres, 0.0);
      if (!rb.cost->Evaluate(params.data(), r.data(), jac ? jptr.data() : nullptr)) return false;
      for (double v : r)
        if (!std::isfinite(v)) return false;
      double sq = 0;
      for (double v : r) sq += v * v;
      RowBlock* out = jac ? &(*jac)[i] : nullptr;
      if (out) { out->row0 = row0[i]; out->rows = nres; }
      // local parameterisation first (residual_block.cc), then the loss corrector
      if (jac) {
        for (int b = 0; b < nb; ++b) {
          int pi = rb.blocks[b];
          int a = act_of[pi];
          if (a < 0) continue;
          auto& pb = p->params_[pi];
          JBlock jb;
          jb.active = a;
          jb.cols = pb.local_size();
          jb.v.assign((size_t)nres * jb.cols, 0.0);
          if (pb.lp) {
            lpj.assign((size_t)pb.size * jb.cols, 0.0);
            pb.lp->ComputeJacobian(params[b], lpj.data());
            for (int rr = 0; rr < nres; ++rr)
              for (int c = 0; c < jb.cols; ++c) {
                double s = 0;
                for (int k = 0; k < pb.size; ++k) s += gjac[b][(size_t)rr * pb.size + k] * lpj[(size_t)k * jb.cols + c];
                jb.v[(size_t)rr * jb.cols + c] = s;
              }
          } else {
            jb.v = gjac[b];
          }
          for (double v : jb.v)
            if (!std::isfinite(v)) return false;
          out->blocks.push_back(std::move(jb));
        }
      }
      if (rb.loss) {
        double rho[3];
        rb.loss->Evaluate(sq, rho);
        *cost += 0.5 * rho[0];
        if (residuals || jac) {
          Corrector corr(sq, rho);
          if (jac)
            for (auto& jb : out->blocks) corr.CorrectJacobian(nres, jb.cols, r.data(), jb.v.data());
          corr.CorrectResiduals(nres, r.data());
        }
      } else {
        *cost += 0.5 * sq;
      }
      if (residuals && nres > 0) std::memcpy(&(*residuals)[row0[i]], r.data(), sizeof(double) * nres);   // (a prior without rows)
      if (gradient && jac)
        for (auto& jb : out->blocks)
          for (int rr = 0; rr < nres; ++rr)
            for (int c = 0; c < jb.cols; ++c) (*gradient)[col0[jb.active] + c] += jb.v[(size_t)rr * jb.cols + c] * r[rr];
    }
    return true;
  }
};

// ORC_DEBUG=2: cost of a state split by cost-function class (diagnosis of rejected steps, tools/diag_rho.py)
void DebugCostByType(const Program& pr, const VecX& x, const char* tag) {
  std::map<std::string, double> by;
  std::vector<double> r;
  for (auto& rb : pr.p->residuals_) {
    const int nres = rb.cost->num_residuals();
    std::vector<const double*> params(rb.blocks.size());
    for (size_t b = 0; b < rb.blocks.size(); ++b) {
      int a = pr.act_of[rb.blocks[b]];
      params[b] = a >= 0 ? &x[pr.st0[a]] : pr.p->params_[rb.blocks[b]].user;
    }
    r.assign(nres, 0.0);
    rb.cost->Evaluate(params.data(), r.data(), nullptr);
    double sq = 0;
    for (double v : r) sq += v * v;
    double rho[3] = {sq, 1, 0};
    if (rb.loss) rb.loss->Evaluate(sq, rho);
    by[typeid(*rb.cost).name()] += 0.5 * rho[0];
  }
  fprintf(stderr, "   %s:", tag);
  for (auto& kv : by) fprintf(stderr, " %s=%.6g", kv.first.c_str(), kv.second);
  fprintf(stderr, "\n");
}

// ---- block sparse matrix ops on the row blocks --------------------------------
void SquaredColumnNorm(const Program& pr, const std::vector<RowBlock>& J, VecX& out) {
  out.assign(pr.num_cols, 0.0);
  for (auto& rb : J)
    for (auto& jb : rb.blocks)
      for (int r = 0; r < rb.rows; ++r)
        for (int c = 0; c < jb.cols; ++c) {
          double v = jb.v[(size_t)r * jb.cols + c];
          out[pr.col0[jb.active] + c] += v * v;
        }
}
void ScaleColumns(const Program& pr, std::vector<RowBlock>& J, const VecX& s) {
  for (auto& rb : J)
    for (auto& jb : rb.blocks)
      for (int r = 0; r < rb.rows; ++r)
        for (int c = 0; c < jb.cols; ++c) jb.v[(size_t)r * jb.cols + c] *= s[pr.col0[jb.active] + c];
}
// y += J x
void RightMultiply(const Program& pr, const std::vector<RowBlock>& J, const VecX& x, VecX& y) {
  for (auto& rb : J)
    for (auto& jb : rb.blocks)
      for (int r = 0; r < rb.rows; ++r) {
        double s = 0;
        for (int c = 0; c < jb.cols; ++c) s += jb.v[(size_t)r * jb.cols + c] * x[pr.col0[jb.active] + c];
        y[rb.row0 + r] += s;
      }
}
// y += J^T x
void LeftMultiply(const Program& pr, const std::vector<RowBlock>& J, const VecX& x, VecX& y) {
  for (auto& rb : J)
    for (auto& jb : rb.blocks)
      for (int r = 0; r < rb.rows; ++r)
        for (int c = 0; c < jb.cols; ++c) y[pr.col0[jb.active] + c] += jb.v[(size_t)r * jb.cols + c] * x[rb.row0 + r];
}

// ---- DENSE_SCHUR: solve (J^T J + D^T D) y = J^T b  (ceres SchurComplementSolver,
//      SchurEliminator::Eliminate / BackSubstitute, dense Cholesky of the reduced system)
enum LinearSolverStatus { LS_SUCCESS, LS_FAILURE };
LinearSolverStatus DenseSchurSolve(const Program& pr, const std::vector<RowBlock>& J, const VecX& b, const VecX& D,
                                   VecX& y) {
  const int ne = pr.num_e, ecols = pr.e_cols, nf = pr.num_cols - ecols;
  MatX lhs(nf, nf);
  VecX rhs(nf, 0.0);
  // chunks: residual blocks grouped by their (single) e-block
  std::vector<std::vector<int>> chunk(ne);
  std::vector<int> no_e;
  for (size_t i = 0; i < J.size(); ++i) {
    int e = -1;
    for (auto& jb : J[i].blocks)
      if (jb.active < ne) e = jb.active;
    if (e >= 0) chunk[e].push_back((int)i); else no_e.push_back((int)i);
  }
  // D on the f-blocks' diagonal
  for (int c = 0; c < nf; ++c) lhs(c, c) = D[ecols + c] * D[ecols + c];
  auto add_ftf = [&](const RowBlock& rb) {
    for (auto& bi : rb.blocks) {
      if (bi.active < ne) continue;
      const int ci = pr.col0[bi.active] - ecols;
      for (int r = 0; r < rb.rows; ++r)
        for (int c = 0; c < bi.cols; ++c) rhs[ci + c] += bi.v[(size_t)r * bi.cols + c] * b[rb.row0 + r];
      for (auto& bj : rb.blocks) {
        if (bj.active < ne) continue;
        const int cj = pr.col0[bj.active] - ecols;
        for (int r = 0; r < rb.rows; ++r)
          for (int c = 0; c < bi.cols; ++c) {
            const double v = bi.v[(size_t)r * bi.cols + c];
            for (int d = 0; d < bj.cols; ++d) lhs(ci + c, cj + d) += v * bj.v[(size_t)r * bj.cols + d];
          }
      }
    }
  };
  for (int i : no_e) add_ftf(J[i]);

  std::vector<MatX> inv_ete(ne);
  std::vector<VecX> g_e(ne);
  for (int e = 0; e < ne; ++e) {
    const int es = pr.lsize(e), ec0 = pr.col0[e];
    MatX ete(es, es);
    for (int k = 0; k < es; ++k) ete(k, k) = D[ec0 + k] * D[ec0 + k];
    VecX g(es, 0.0);
    // E^T F accumulated per f column (dense row over the f space, sparse in practice)
    MatX etf(es, nf);
    std::vector<char> touched(nf, 0);
    for (int i : chunk[e]) {
      const RowBlock& rb = J[i];
      const JBlock* E = nullptr;
      for (auto& jb : rb.blocks)
        if (jb.active == e) E = &jb;
      for (int r = 0; r < rb.rows; ++r)
        for (int k = 0; k < es; ++k) {
          const double ek = E->v[(size_t)r * es + k];
          g[k] += ek * b[rb.row0 + r];
          for (int l = 0; l < es; ++l) ete(k, l) += ek * E->v[(size_t)r * es + l];
          for (auto& jb : rb.blocks) {
            if (jb.active < ne) continue;
            const int cj = pr.col0[jb.active] - ecols;
            for (int d = 0; d < jb.cols; ++d) { etf(k, cj + d) += ek * jb.v[(size_t)r * jb.cols + d]; touched[cj + d] = 1; }
          }
        }
      add_ftf(rb);
    }
    // ceres InvertPSDMatrix: Cholesky-based inverse of the (small) e-block
    MatX L = ete;
    if (!cholesky_lower(L)) return LS_FAILURE;
    MatX inv(es, es);
    for (int c = 0; c < es; ++c) {
      VecX col(es, 0.0);
      col[c] = 1.0;
      cholesky_solve(L, col);
      for (int k = 0; k < es; ++k) inv(k, c) = col[k];
    }
    inv_ete[e] = inv;
    g_e[e] = g;
    std::vector<int> cols;
    for (int c = 0; c < nf; ++c)
      if (touched[c]) cols.push_back(c);
    // lhs -= (E^T F)^T inv (E^T F) ; rhs -= (E^T F)^T inv g
    VecX ig(es, 0.0);
    for (int k = 0; k < es; ++k)
      for (int l = 0; l < es; ++l) ig[k] += inv(k, l) * g[l];
    for (int c : cols) {
      double s = 0;
      for (int k = 0; k < es; ++k) s += etf(k, c) * ig[k];
      rhs[c] -= s;
      VecX t(es, 0.0);
      for (int k = 0; k < es; ++k)
        for (int l = 0; l < es; ++l) t[k] += inv(k, l) * etf(l, c);
      for (int d : cols) {
        double u = 0;
        for (int k = 0; k < es; ++k) u += etf(k, d) * t[k];
        lhs(d, c) -= u;
      }
    }
  }
  // dense Cholesky of the reduced camera system (ceres uses Eigen LLT on the upper triangle)
  MatX L = lhs;
  if (!cholesky_lower(L)) return LS_FAILURE;
  VecX yf = rhs;
  cholesky_solve(L, yf);
  y.assign(pr.num_cols, 0.0);
  for (int c = 0; c < nf; ++c) y[ecols + c] = yf[c];
  // back substitution: y_e = inv_ete (g_e - E^T F y_f)
  for (int e = 0; e < ne; ++e) {
    const int es = pr.lsize(e), ec0 = pr.col0[e];
    VecX t = g_e[e];
    for (int i : chunk[e]) {
      const RowBlock& rb = J[i];
      const JBlock* E = nullptr;
      for (auto& jb : rb.blocks)
        if (jb.active == e) E = &jb;
      for (int r = 0; r < rb.rows; ++r) {
        double fy = 0;
        for (auto& jb : rb.blocks) {
          if (jb.active < ne) continue;
          const int cj = pr.col0[jb.active] - ecols;
          for (int d = 0; d < jb.cols; ++d) fy += jb.v[(size_t)r * jb.cols + d] * yf[cj + d];
        }
        for (int k = 0; k < es; ++k) t[k] -= E->v[(size_t)r * es + k] * fy;
      }
    }
    for (int k = 0; k < es; ++k) {
      double s = 0;
      for (int l = 0; l < es; ++l) s += inv_ete[e](k, l) * t[l];
      y[ec0 + k] = s;
    }
  }
  for (double v : y)
    if (!std::isfinite(v)) return LS_FAILURE;
  return LS_SUCCESS;
}

// ---- ceres DoglegStrategy (TRADITIONAL_DOGLEG), dogleg_strategy.cc ------------
struct Dogleg {
  double radius, max_radius, min_diagonal, max_diagonal;
  double mu = 1e-8, min_mu = 1e-8, max_mu = 1.0, mu_increase_factor = 10.0;
  double increase_threshold = 0.75, decrease_threshold = 0.25;
  double dogleg_step_norm = 0.0, alpha = 0.0;
  bool reuse = false;
  VecX diagonal, gradient, gauss_newton_step;

  explicit Dogleg(const SolverOptions& o)
      : radius(o.initial_trust_region_radius), max_radius(o.max_trust_region_radius),
        min_diagonal(o.min_lm_diagonal), max_diagonal(o.max_lm_diagonal) {}

  LinearSolverStatus ComputeStep(const Program& pr, const std::vector<RowBlock>& J, const VecX& residuals, VecX& step) {
    const int n = pr.num_cols;
    if (reuse) {
      ComputeTraditionalDoglegStep(step);
      return LS_SUCCESS;
    }
    reuse = true;
    SquaredColumnNorm(pr, J, diagonal);
    for (int i = 0; i < n; ++i) diagonal[i] = std::sqrt(std::min(std::max(diagonal[i], min_diagonal), max_diagonal));
    // ComputeGradient
    gradient.assign(n, 0.0);
    LeftMultiply(pr, J, residuals, gradient);
    for (int i = 0; i < n; ++i) gradient[i] /= diagonal[i];
    // ComputeCauchyPoint
    {
      VecX Jg(pr.num_rows, 0.0), sg(n);
      for (int i = 0; i < n; ++i) sg[i] = gradient[i] / diagonal[i];
      RightMultiply(pr, J, sg, Jg);
      double g2 = 0, jg2 = 0;
      for (double v : gradient) g2 += v * v;
      for (double v : Jg) jg2 += v * v;
      alpha = g2 / jg2;
    }
    // ComputeGaussNewtonStep
    LinearSolverStatus st = LS_FAILURE;
    while (mu < max_mu) {
      VecX lm(n);
      for (int i = 0; i < n; ++i) lm[i] = diagonal[i] * std::sqrt(mu);
      st = DenseSchurSolve(pr, J, residuals, lm, gauss_newton_step);
      if (st == LS_FAILURE) { mu *= mu_increase_factor; continue; }
      break;
    }
    if (st != LS_FAILURE) {
      for (int i = 0; i < n; ++i) gauss_newton_step[i] *= -diagonal[i];
      ComputeTraditionalDoglegStep(step);
    }
    return st;
  }

  void ComputeTraditionalDoglegStep(VecX& dogleg) {
    const int n = (int)gradient.size();
    dogleg.assign(n, 0.0);
    double gn2 = 0, g2 = 0;
    for (int i = 0; i < n; ++i) { g2 += gradient[i] * gradient[i]; gn2 += gauss_newton_step[i] * gauss_newton_step[i]; }
    const double gradient_norm = std::sqrt(g2), gauss_newton_norm = std::sqrt(gn2);
    if (gauss_newton_norm <= radius) {
      for (int i = 0; i < n; ++i) dogleg[i] = gauss_newton_step[i] / diagonal[i];
      dogleg_step_norm = gauss_newton_norm;
      return;
    }
    if (gradient_norm * alpha >= radius) {
      for (int i = 0; i < n; ++i) dogleg[i] = -(radius / gradient_norm) * gradient[i] / diagonal[i];
      dogleg_step_norm = radius;
      return;
    }
    double gdotgn = 0;
    for (int i = 0; i < n; ++i) gdotgn += gradient[i] * gauss_newton_step[i];
    const double b_dot_a = -alpha * gdotgn;
    const double a_squared_norm = std::pow(alpha * gradient_norm, 2.0);
    const double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + std::pow(gauss_newton_norm, 2);
    const double c = b_dot_a - a_squared_norm;
    const double d = std::sqrt(c * c + b_minus_a_squared_norm * (std::pow(radius, 2.0) - a_squared_norm));
    const double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + c);
    double nn = 0;
    for (int i = 0; i < n; ++i) {
      dogleg[i] = (-alpha * (1.0 - beta)) * gradient[i] + beta * gauss_newton_step[i];
      nn += dogleg[i] * dogleg[i];
    }
    dogleg_step_norm = std::sqrt(nn);
    for (int i = 0; i < n; ++i) dogleg[i] /= diagonal[i];
  }
  void StepAccepted(double step_quality) {
    if (step_quality < decrease_threshold) radius *= 0.5;
    if (step_quality > increase_threshold) radius = std::max(radius, 3.0 * dogleg_step_norm);
    mu = std::max(min_mu, 2.0 * mu / mu_increase_factor);
    reuse = false;
  }
  void StepRejected(double) { radius *= 0.5; reuse = true; }
  void StepIsInvalid() { mu *= mu_increase_factor; reuse = false; }
};

// ---- ceres LevenbergMarquardtStrategy, levenberg_marquardt_strategy.cc (ceres 1.12, restated from the published
//      algorithm): D = sqrt(clamp(diag(J^T J)) / radius), step = -(J^T J + D^2)^-1 J^T r through the same dense Schur
//      solver; accepted: radius /= max(1/3, 1 - (2 rho - 1)^3), decrease_factor = 2; rejected: radius /= decrease_factor,
//      decrease_factor *= 2.  The diagonal is re-used after a rejected step.
struct LevenbergMarquardt {
  double radius, max_radius, min_diagonal, max_diagonal;
  double decrease_factor = 2.0;
  bool reuse_diagonal = false;
  VecX diagonal, lm_diagonal;
  // members the debug print of Solve() reads for the dogleg strategy
  VecX gauss_newton_step;
  double dogleg_step_norm = 0.0, mu = 0.0;

  explicit LevenbergMarquardt(const SolverOptions& o)
      : radius(o.initial_trust_region_radius), max_radius(o.max_trust_region_radius),
        min_diagonal(o.min_lm_diagonal), max_diagonal(o.max_lm_diagonal) {}

  LinearSolverStatus ComputeStep(const Program& pr, const std::vector<RowBlock>& J, const VecX& residuals, VecX& step) {
    const int n = pr.num_cols;
    if (!reuse_diagonal) {
      SquaredColumnNorm(pr, J, diagonal);
      for (int i = 0; i < n; ++i) diagonal[i] = std::min(std::max(diagonal[i], min_diagonal), max_diagonal);
    }
    lm_diagonal.resize(n);
    for (int i = 0; i < n; ++i) lm_diagonal[i] = std::sqrt(diagonal[i] / radius);
    LinearSolverStatus st = DenseSchurSolve(pr, J, residuals, lm_diagonal, step);
    if (st == LS_FAILURE) return st;
    for (double& v : step) v = -v;
    reuse_diagonal = true;
    return st;
  }
  void StepAccepted(double step_quality) {
    radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * step_quality - 1.0, 3));
    radius = std::min(max_radius, radius);
    decrease_factor = 2.0;
    reuse_diagonal = false;
  }
  void StepRejected(double) {
    radius = radius / decrease_factor;
    decrease_factor *= 2.0;
    reuse_diagonal = true;
  }
  void StepIsInvalid() {}
};

}  // namespace

// ---- ceres TrustRegionMinimizer::Minimize (monotonic, unconstrained) ------------
template <class Strategy>
static void SolveWith(const SolverOptions& opt, Problem* problem, SolverSummary* sum) {
  Program pr(problem);
  VecX x, candidate_x, residuals, gradient, scale, step, delta, model_residuals;
  std::vector<RowBlock> J;
  pr.StateFromUser(x);
  double x_cost = 0, candidate_cost = 0, model_cost_change = 0;
  Strategy strategy(opt);
  int num_consecutive_invalid_steps = 0;
  IterationSummary it;
  auto norm = [](const VecX& v) { double s = 0; for (double a : v) s += a * a; return std::sqrt(s); };
  double x_norm = norm(x);

  auto EvaluateGradientAndJacobian = [&]() -> bool {
    if (!pr.Evaluate(x, &x_cost, &residuals, &gradient, &J)) return false;
    it.cost = x_cost;
    if (opt.jacobi_scaling) {
      if (it.iteration == 0) {
        SquaredColumnNorm(pr, J, scale);
        for (double& s : scale) s = 1.0 / (1.0 + std::sqrt(s));
      }
      ScaleColumns(pr, J, scale);
    } else if (it.iteration == 0) {
      scale.assign(pr.num_cols, 1.0);
    }
    VecX neg(gradient.size()), proj;
    for (size_t i = 0; i < gradient.size(); ++i) neg[i] = -gradient[i];
    if (!pr.Plus(x, neg, proj)) return false;
    double mx = 0;
    for (size_t i = 0; i < x.size(); ++i) mx = std::max(mx, std::fabs(x[i] - proj[i]));
    it.gradient_max_norm = mx;
    return true;
  };

  // IterationZero
  it = IterationSummary();
  if (!EvaluateGradientAndJacobian()) {
    sum->termination_type = FAILURE;
    sum->message = "Initial residual and Jacobian evaluation failed.";
    return;
  }
  sum->initial_cost = x_cost;
  it.step_is_valid = it.step_is_successful = true;

  for (;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it.step_is_successful) ++sum->num_successful_steps; else ++sum->num_unsuccessful_steps;
    it.trust_region_radius = strategy.radius;
    sum->iterations.push_back(it);
    if (it.iteration >= opt.max_num_iterations) { sum->termination_type = NO_CONVERGENCE; sum->message = "Maximum number of iterations reached."; break; }
    if (it.step_is_successful && it.gradient_max_norm <= opt.gradient_tolerance) { sum->termination_type = CONVERGENCE; sum->message = "Gradient tolerance reached."; break; }
    if (it.trust_region_radius <= opt.min_trust_region_radius) { sum->termination_type = CONVERGENCE; sum->message = "Minimum trust region radius reached."; break; }

    const int iter = sum->iterations.back().iteration + 1;
    const double prev_gmax = sum->iterations.back().gradient_max_norm;
    it = IterationSummary();
    it.iteration = iter;

    // ComputeTrustRegionStep
    it.step_is_valid = false;
    LinearSolverStatus st = strategy.ComputeStep(pr, J, residuals, step);
    if (st == LS_SUCCESS) {
      model_residuals.assign(pr.num_rows, 0.0);
      RightMultiply(pr, J, step, model_residuals);
      double mc = 0;
      for (int r = 0; r < pr.num_rows; ++r) mc += model_residuals[r] * (residuals[r] + model_residuals[r] / 2.0);
      model_cost_change = -mc;
      it.step_is_valid = (model_cost_change > 0.0);
      if (it.step_is_valid) {
        delta.resize(step.size());
        for (size_t i = 0; i < step.size(); ++i) delta[i] = step[i] * scale[i];
        num_consecutive_invalid_steps = 0;
      }
    }
    if (!it.step_is_valid) {
      // HandleInvalidStep
      if (++num_consecutive_invalid_steps >= opt.max_num_consecutive_invalid_steps) {
        sum->termination_type = FAILURE;
        sum->message = "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps";
        break;
      }
      strategy.StepIsInvalid();
      it.cost = x_cost;
      it.gradient_max_norm = prev_gmax;
      continue;
    }
    // ComputeCandidatePointAndEvaluateCost
    if (!pr.Plus(x, delta, candidate_x)) {
      candidate_cost = std::numeric_limits<double>::max();
    } else if (!pr.Evaluate(candidate_x, &candidate_cost, nullptr, nullptr, nullptr)) {
      candidate_cost = std::numeric_limits<double>::max();
    }
    // ParameterToleranceReached
    {
      double s = 0;
      for (size_t i = 0; i < x.size(); ++i) s += (x[i] - candidate_x[i]) * (x[i] - candidate_x[i]);
      it.step_norm = std::sqrt(s);
      if (it.step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) {
        sum->termination_type = CONVERGENCE;
        sum->message = "Parameter tolerance reached.";
        break;
      }
    }
    // FunctionToleranceReached
    it.cost_change = x_cost - candidate_cost;
    if (getenv("ORC_DEBUG"))
      fprintf(stderr, "it %d ftol check: |dcost| %.6g vs %.6g (ratio %.4g)\n", it.iteration, std::fabs(it.cost_change),
              opt.function_tolerance * x_cost, std::fabs(it.cost_change) / (opt.function_tolerance * x_cost));
    if (std::fabs(it.cost_change) <= opt.function_tolerance * x_cost) {
      sum->termination_type = CONVERGENCE;
      sum->message = "Function tolerance reached.";
      break;
    }
    // IsStepSuccessful (monotonic TrustRegionStepEvaluator)
    it.relative_decrease = (x_cost - candidate_cost) / model_cost_change;
    if (getenv("ORC_DEBUG"))
      fprintf(stderr, "it %d cost %.6g cand %.6g model_change %.6g rho %.4g radius %.4g gn_norm %.4g step_norm %.4g mu %.3g\n", it.iteration,
              x_cost, candidate_cost, model_cost_change, it.relative_decrease, strategy.radius,
              norm(strategy.gauss_newton_step), strategy.dogleg_step_norm, strategy.mu);
    if (getenv("ORC_DEBUG") && atoi(getenv("ORC_DEBUG")) >= 2) {
      DebugCostByType(pr, x, "x   ");
      DebugCostByType(pr, candidate_x, "cand");
    }
    if (getenv("ORC_DEBUG") && atoi(getenv("ORC_DEBUG")) >= 3) {
      // the residual blocks whose cost rises most, with the last parameter block (the landmark) before / after
      std::vector<std::pair<double, int>> inc;
      std::vector<double> r;
      auto cost_of = [&](int i, const VecX& xx) {
        auto& rb = problem->residuals_[i];
        std::vector<const double*> params(rb.blocks.size());
        for (size_t b = 0; b < rb.blocks.size(); ++b) {
          int a = pr.act_of[rb.blocks[b]];
          params[b] = a >= 0 ? &xx[pr.st0[a]] : problem->params_[rb.blocks[b]].user;
        }
        r.assign(rb.cost->num_residuals(), 0.0);
        rb.cost->Evaluate(params.data(), r.data(), nullptr);
        double sq = 0;
        for (double v : r) sq += v * v;
        double rho[3] = {sq, 1, 0};
        if (rb.loss) rb.loss->Evaluate(sq, rho);
        return 0.5 * rho[0];
      };
      for (size_t i = 0; i < problem->residuals_.size(); ++i) inc.push_back({cost_of((int)i, candidate_x) - cost_of((int)i, x), (int)i});
      std::sort(inc.begin(), inc.end(), [](auto& a, auto& b) { return a.first > b.first; });
      for (int k = 0; k < 12 && k < (int)inc.size(); ++k) {
        auto& rb = problem->residuals_[inc[k].second];
        int a = pr.act_of[rb.blocks.back()];
        fprintf(stderr, "   +%.4g blk %d %s cost_x %.4g lm#%d:", inc[k].first, inc[k].second, typeid(*rb.cost).name(), cost_of(inc[k].second, x), a);
        for (int c = 0; c < pr.gsize(a); ++c) fprintf(stderr, " %.5f->%.5f", x[pr.st0[a] + c], candidate_x[pr.st0[a] + c]);
        fprintf(stderr, " | delta");
        for (int c = 0; c < pr.lsize(a); ++c) fprintf(stderr, " %.4g", delta[pr.col0[a] + c]);
        fprintf(stderr, "\n");
      }
    }
    it.gradient_max_norm = prev_gmax;
    if (it.relative_decrease > opt.min_relative_decrease) {
      x = candidate_x;
      x_norm = norm(x);
      if (!EvaluateGradientAndJacobian()) {
        sum->termination_type = FAILURE;
        sum->message = "Residual and Jacobian evaluation failed.";
        break;
      }
      it.step_is_successful = true;
      strategy.StepAccepted(it.relative_decrease);
    } else {
      it.step_is_successful = false;
      strategy.StepRejected(it.relative_decrease);
      it.cost = candidate_cost;
    }
  }
  sum->final_cost = x_cost;
  pr.StateToUser(x);
}

void Solve(const SolverOptions& opt, Problem* problem, SolverSummary* sum) {
  if (opt.use_dogleg) SolveWith<Dogleg>(opt, problem, sum);
  else SolveWith<LevenbergMarquardt>(opt, problem, sum);
}

}  // namespace orc
