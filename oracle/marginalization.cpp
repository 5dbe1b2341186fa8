// TEST INFRASTRUCTURE -- CPU oracle (see marginalization.h for the reference map).
#include "marginalization.h"
#include <functional>
#include <numeric>
#include <thread>

namespace orc {

// TEST SWITCH (0 = the reference's rule).  > 0: the spectrum of the kept block is cut at max(1e-8, rel * max diagonal) instead of
// the reference's absolute 1e-8 (marginalization_factor.cpp:349-357) -- the cut the device's pivoted Cholesky makes
// (csrc/ba_marg.h, kMargNoiseRel); tests/test_gpu_chain.py uses it to measure what that deviation does over a long chain.
double g_marg_rel_eps = 0.0;
// TEST SWITCH (0 = as the reference writes the product).  2: the inner sums of both Schur complements accumulate in long double and
// are rounded once.  1: the inner sums of the last Schur complement A = Arr - (Arm Amm^+) Amr
// (marginalization_factor.cpp:345) run from the last term to the first -- the same numbers, another rounding.  Entries of
// the IMU rows reach 5e14 (bias random walk), the kept block's weak eigenvalues sit near 1e-8: tools/fuzz_parity.py uses
// the switch to tell windows whose prior is decided by that rounding (Eigen's blocked products round differently again)
// from a wrong device.
int g_marg_reverse_sums = 0;
int g_marg_threads = 1;   // NUM_THREADS of marginalization_factor.h:13 is 4; 1 = the same sums without threads

// marginalization_factor.cpp:3-69
void ResidualBlockInfo::Evaluate() {
  const int nres = cost_function->num_residuals();
  residuals.assign(nres, 0.0);
  const std::vector<int>& block_sizes = cost_function->parameter_block_sizes();
  std::vector<double*> raw(block_sizes.size());
  jacobians.resize(block_sizes.size());
  for (size_t i = 0; i < block_sizes.size(); ++i) {
    jacobians[i].assign((size_t)nres * block_sizes[i], 0.0);
    raw[i] = jacobians[i].data();
  }
  cost_function->Evaluate(parameter_blocks.data(), residuals.data(), raw.data());

  if (loss_function) {
    double sq_norm = 0, rho[3];
    for (double v : residuals) sq_norm += v * v;
    loss_function->Evaluate(sq_norm, rho);
    Corrector corr(sq_norm, rho);  // same arithmetic as :46-60
    for (size_t i = 0; i < block_sizes.size(); ++i)
      corr.CorrectJacobian(nres, block_sizes[i], residuals.data(), jacobians[i].data());
    corr.CorrectResiduals(nres, residuals.data());
  }
}

MarginalizationInfo::~MarginalizationInfo() {
  for (auto& it : parameter_block_data) delete[] it.second;
  for (auto* f : factors) {
    delete f->cost_function;
    delete f;
  }
}

// marginalization_factor.cpp:89-108
void MarginalizationInfo::addResidualBlockInfo(ResidualBlockInfo* info) {
  factors.emplace_back(info);
  const std::vector<int>& sizes = info->cost_function->parameter_block_sizes();
  for (size_t i = 0; i < info->parameter_blocks.size(); ++i)
    parameter_block_size[reinterpret_cast<long>(info->parameter_blocks[i])] = sizes[i];
  for (int d : info->drop_set) parameter_block_idx[reinterpret_cast<long>(info->parameter_blocks[d])] = 0;
}

// marginalization_factor.cpp:110-129
void MarginalizationInfo::preMarginalize() {
  for (auto* it : factors) {
    it->Evaluate();
    const std::vector<int>& sizes = it->cost_function->parameter_block_sizes();
    for (size_t i = 0; i < sizes.size(); ++i) {
      long addr = reinterpret_cast<long>(it->parameter_blocks[i]);
      if (parameter_block_data.find(addr) == parameter_block_data.end()) {
        double* data = new double[sizes[i]];
        std::memcpy(data, it->parameter_blocks[i], sizeof(double) * sizes[i]);
        parameter_block_data[addr] = data;
      }
    }
  }
}

// marginalization_factor.cpp:177-363
void MarginalizationInfo::marginalize() {
  int pos = 0;
  int marg_pose_size = 0;
  std::map<int, int, std::greater<int>> marg_pose_index_size;
  for (auto& it : parameter_block_idx) {
    if (localSize(parameter_block_size[it.first]) > 4) {
      marg_pose_index_size.insert(std::make_pair(pos, localSize(parameter_block_size[it.first])));
      marg_pose_size += localSize(parameter_block_size[it.first]);
    }
    it.second = pos;
    pos += localSize(parameter_block_size[it.first]);
  }
  m = pos;
  for (const auto& it : parameter_block_size) {
    if (parameter_block_idx.find(it.first) == parameter_block_idx.end()) {
      parameter_block_idx[it.first] = pos;
      pos += localSize(it.second);
    }
  }
  n = pos - m;

  MatX A(pos, pos);
  VecX b(pos, 0.0);
  // ThreadsConstructA (:144-175): factor i goes to thread i % NUM_THREADS, every thread fills its own pos x pos matrix,
  // the partial sums are added afterwards (:253-280).  g_marg_threads == 1 adds everything in one pass (same sums).
  auto construct = [&](int t, int nt, MatX& At, VecX& bt) {
    for (size_t fi = t; fi < factors.size(); fi += nt) {
      ResidualBlockInfo* it = factors[fi];
      const int nres = (int)it->residuals.size();
      const std::vector<int>& sizes = it->cost_function->parameter_block_sizes();
      for (size_t i = 0; i < it->parameter_blocks.size(); ++i) {
        int idx_i = parameter_block_idx.at(reinterpret_cast<long>(it->parameter_blocks[i]));
        int gs_i = sizes[i];
        int size_i = localSize(parameter_block_size.at(reinterpret_cast<long>(it->parameter_blocks[i])));
        const double* Ji = it->jacobians[i].data();
        for (size_t j = i; j < it->parameter_blocks.size(); ++j) {
          int idx_j = parameter_block_idx.at(reinterpret_cast<long>(it->parameter_blocks[j]));
          int gs_j = sizes[j];
          int size_j = localSize(parameter_block_size.at(reinterpret_cast<long>(it->parameter_blocks[j])));
          const double* Jj = it->jacobians[j].data();
          for (int a = 0; a < size_i; ++a)
            for (int c = 0; c < size_j; ++c) {
              double s = 0;
              for (int r = 0; r < nres; ++r) s += Ji[(size_t)r * gs_i + a] * Jj[(size_t)r * gs_j + c];
              At(idx_i + a, idx_j + c) += s;
              if (i != j) At(idx_j + c, idx_i + a) = At(idx_i + a, idx_j + c);
            }
        }
        for (int a = 0; a < size_i; ++a) {
          double s = 0;
          for (int r = 0; r < nres; ++r) s += Ji[(size_t)r * gs_i + a] * it->residuals[r];
          bt[idx_i + a] += s;
        }
      }
    }
  };
  if (g_marg_threads <= 1) {
    construct(0, 1, A, b);
  } else {
    const int nt = g_marg_threads;
    std::vector<MatX> As(nt, MatX(pos, pos));
    std::vector<VecX> bs(nt, VecX(pos, 0.0));
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(construct, t, nt, std::ref(As[t]), std::ref(bs[t]));
    for (int t = 0; t < nt; ++t) {
      th[t].join();
      for (size_t k = 0; k < A.a.size(); ++k) A.a[k] += As[t].a[k];
      for (int k = 0; k < pos; ++k) b[k] += bs[t][k];
    }
  }

  // marg landmarks first (:282-327): move pose-like blocks to the end of the m range
  int m1 = m - marg_pose_size;
  int n1 = n + marg_pose_size;
  if (m1 > 0) {
    // permutation equivalent to the row/column moves at :291-309
    std::vector<int> perm(pos);
    std::iota(perm.begin(), perm.end(), 0);
    for (auto& iter : marg_pose_index_size) {
      int idx = iter.first, size = iter.second;
      std::vector<int> tmp(perm.begin() + idx, perm.begin() + idx + size);
      perm.erase(perm.begin() + idx, perm.begin() + idx + size);
      perm.insert(perm.begin() + (m - size), tmp.begin(), tmp.end());
    }
    MatX Ar(pos, pos);
    VecX br(pos);
    for (int i = 0; i < pos; ++i) {
      br[i] = b[perm[i]];
      for (int j = 0; j < pos; ++j) Ar(i, j) = A(perm[i], perm[j]);
    }
    MatX Amm1(m1, m1), Amm_inv1;
    for (int i = 0; i < m1; ++i)
      for (int j = 0; j < m1; ++j) Amm1(i, j) = Ar(i, j);
    inverse_lu(Amm1, Amm_inv1);  // Eigen MatrixXd::inverse() == PartialPivLU
    // tempA = Arm1 * Amm_inv1 ; A = Arr1 - tempA*Amr1 ; b = brr1 - tempA*bmm1
    MatX tempA(n1, m1);
    for (int i = 0; i < n1; ++i)
      for (int j = 0; j < m1; ++j) {
        double s = 0;
        for (int k = 0; k < m1; ++k) s += Ar(m1 + i, k) * Amm_inv1(k, j);
        tempA(i, j) = s;
      }
    MatX A2(n1, n1);
    VecX b2(n1);
    for (int i = 0; i < n1; ++i) {
      for (int j = 0; j < n1; ++j) {
        double s = 0;
        if (g_marg_reverse_sums == 2) {   // test switch: the sum in extended precision, rounded once
          long double sl = 0;
          for (int k = 0; k < m1; ++k) sl += (long double)tempA(i, k) * Ar(k, m1 + j);
          s = (double)sl;
        } else
        for (int k = 0; k < m1; ++k) s += tempA(i, k) * Ar(k, m1 + j);
        A2(i, j) = Ar(m1 + i, m1 + j) - s;
      }
      double s = 0;
      for (int k = 0; k < m1; ++k) s += tempA(i, k) * br[k];
      b2[i] = br[m1 + i] - s;
    }
    A = A2;
    b = b2;
  }

  // then marg pose (:329-346)
  int m2 = m - m1;
  int n2 = n;
  MatX Amm(m2, m2);
  for (int i = 0; i < m2; ++i)
    for (int j = 0; j < m2; ++j) Amm(i, j) = 0.5 * (A(i, j) + A(j, i));
  VecX ev;
  MatX V;
  sym_eigen(Amm, ev, V);
  MatX Amm_inv(m2, m2);
  for (int i = 0; i < m2; ++i)
    for (int j = 0; j < m2; ++j) {
      double s = 0;
      for (int k = 0; k < m2; ++k) s += V(i, k) * (ev[k] > eps ? 1.0 / ev[k] : 0.0) * V(j, k);
      Amm_inv(i, j) = s;
    }
  MatX tempB(n2, m2);
  for (int i = 0; i < n2; ++i)
    for (int j = 0; j < m2; ++j) {
      double s = 0;
      for (int k = 0; k < m2; ++k) s += A(m2 + i, k) * Amm_inv(k, j);
      tempB(i, j) = s;
    }
  MatX A3(n2, n2);
  VecX b3(n2);
  for (int i = 0; i < n2; ++i) {
    for (int j = 0; j < n2; ++j) {
      double s = 0;
      if (!g_marg_reverse_sums) for (int k = 0; k < m2; ++k) s += tempB(i, k) * A(k, m2 + j);
      else if (g_marg_reverse_sums == 2) {
        long double sl = 0;
        for (int k = 0; k < m2; ++k) sl += (long double)tempB(i, k) * A(k, m2 + j);
        s = (double)sl;
      } else for (int k = m2 - 1; k >= 0; --k) s += tempB(i, k) * A(k, m2 + j);   // test switch: the same sum, last term first
      A3(i, j) = A(m2 + i, m2 + j) - s;
    }
    double s = 0;
    for (int k = 0; k < m2; ++k) s += tempB(i, k) * b[k];
    b3[i] = b[m2 + i] - s;
  }
  A_final = A3;
  b_final = b3;

  // :349-357
  VecX S;
  MatX V2;
  sym_eigen(A3, S, V2);
  linearized_jacobians.resize(n2, n2);
  linearized_residuals.assign(n2, 0.0);
  double cut = eps;
  if (g_marg_rel_eps > 0.0) {
    double dmax = 0.0;
    for (int i = 0; i < n2; ++i) dmax = std::max(dmax, A3(i, i));
    cut = std::max(eps, g_marg_rel_eps * dmax);
  }
  for (int k = 0; k < n2; ++k) {
    const double s = S[k] > cut ? S[k] : 0.0;
    const double sinv = S[k] > cut ? 1.0 / S[k] : 0.0;
    const double s_sqrt = std::sqrt(s), sinv_sqrt = std::sqrt(sinv);
    double vb = 0;
    for (int i = 0; i < n2; ++i) {
      linearized_jacobians(k, i) = s_sqrt * V2(i, k);
      vb += V2(i, k) * b3[i];
    }
    linearized_residuals[k] = sinv_sqrt * vb;
  }
}

// marginalization_factor.cpp:458-478
std::vector<double*> MarginalizationInfo::getParameterBlocks(std::map<long, double*>& addr_shift) {
  std::vector<double*> keep_block_addr;
  keep_block_size.clear();
  keep_block_idx.clear();
  keep_block_data.clear();
  for (const auto& it : parameter_block_idx) {
    if (it.second >= m) {
      keep_block_size.push_back(parameter_block_size[it.first]);
      keep_block_idx.push_back(parameter_block_idx[it.first]);
      keep_block_data.push_back(parameter_block_data[it.first]);
      keep_block_addr.push_back(addr_shift[it.first]);
    }
  }
  sum_block_size = std::accumulate(keep_block_size.begin(), keep_block_size.end(), 0);
  return keep_block_addr;
}

MarginalizationFactor::MarginalizationFactor(MarginalizationInfo* info) : marginalization_info(info) {
  for (int s : info->keep_block_size) sizes_.push_back(s);
  nres_ = info->n;
}

// marginalization_factor.cpp:492-542
bool MarginalizationFactor::Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
  const MarginalizationInfo* mi = marginalization_info;
  const int n = mi->n, m = mi->m;
  VecX dx(n, 0.0);
  for (size_t i = 0; i < mi->keep_block_size.size(); ++i) {
    int size = mi->keep_block_size[i];
    int idx = mi->keep_block_idx[i] - m;
    const double* x = parameters[i];
    const double* x0 = mi->keep_block_data[i];
    if (size != 7) {
      for (int k = 0; k < size; ++k) dx[idx + k] = x[k] - x0[k];
    } else {
      for (int k = 0; k < 3; ++k) dx[idx + k] = x[k] - x0[k];
      Quat q0(x0[6], x0[3], x0[4], x0[5]), q(x[6], x[3], x[4], x[5]);
      Quat dq = q0.inverse() * q;
      Vec3 v = dq.vec() * 2.0;
      if (!(dq.w >= 0)) v = -v;
      for (int k = 0; k < 3; ++k) dx[idx + 3 + k] = v[k];
    }
  }
  for (int r = 0; r < n; ++r) {
    double s = mi->linearized_residuals[r];
    for (int c = 0; c < n; ++c) s += mi->linearized_jacobians(r, c) * dx[c];
    residuals[r] = s;
  }
  if (jacobians) {
    for (size_t i = 0; i < mi->keep_block_size.size(); ++i) {
      if (!jacobians[i]) continue;
      int size = mi->keep_block_size[i], local_size = MarginalizationInfo::localSize(size);
      int idx = mi->keep_block_idx[i] - m;
      for (int r = 0; r < n; ++r) {
        for (int c = 0; c < size; ++c) jacobians[i][(size_t)r * size + c] = 0.0;
        for (int c = 0; c < local_size; ++c) jacobians[i][(size_t)r * size + c] = mi->linearized_jacobians(r, idx + c);
      }
    }
  }
  return true;
}

}  // namespace orc
