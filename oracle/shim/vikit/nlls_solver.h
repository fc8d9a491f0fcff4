// -*- C++ -*-
// oracle/shim/vikit/nlls_solver.h -- TEST INFRASTRUCTURE ONLY.
// vk::NLLSSolver<D,T> restated from rpg_vikit (nlls_solver.h / nlls_solver_impl.hpp): the
// Gauss-Newton driver that svo::SparseImgAlign derives from (svo/include/svo/
// sparse_img_align.h:33).  Levenberg-Marquardt is not restated (SVO never selects it:
// frame_handler_mono.cpp:136-138 passes GaussNewton).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <Eigen/Core>
#include <vikit/math_utils.h>
#include <vikit/robust_cost.h>
namespace vk {
using namespace Eigen;
template <int D, typename T> class NLLSSolver {
 public:
  typedef T ModelType;
  enum Method { GaussNewton, LevenbergMarquardt };
  enum ScaleEstimatorType { UnitScale, TDistScale, MADScale, NormalScale };
  enum WeightFunctionType { UnitWeight, TDistWeight, TukeyWeight, HuberWeight };

 protected:
  Matrix<double, D, D> H_;
  Matrix<double, D, 1> Jres_;
  Matrix<double, D, 1> x_;
  bool have_prior_;
  ModelType prior_;
  Matrix<double, D, D> I_prior_;
  double chi2_;
  double rho_;
  Method method_;

  virtual double computeResiduals(const ModelType& model, bool linearize_system, bool compute_weight_scale) = 0;
  virtual int solve() = 0;
  virtual void update(const ModelType& old_model, ModelType& new_model) = 0;
  virtual void applyPrior(const ModelType&) {}
  virtual void startIteration() {}
  virtual void finishIteration() {}
  virtual void finish() {}

 public:
  double mu_init_, mu_;
  double nu_init_, nu_;
  size_t n_iter_init_, n_iter_;
  size_t n_trials_;
  size_t n_trials_max_;
  size_t n_meas_;
  bool stop_;
  bool verbose_;
  double eps_;
  size_t iter_;
  bool use_weights_;
  float scale_;
  robust_cost::ScaleEstimatorPtr scale_estimator_;
  robust_cost::WeightFunctionPtr weight_function_;

  NLLSSolver()
      : have_prior_(false), method_(LevenbergMarquardt), mu_init_(0.01f), mu_(mu_init_), nu_init_(2.0), nu_(nu_init_),
        n_iter_init_(15), n_iter_(n_iter_init_), n_trials_(0), n_trials_max_(5), n_meas_(0), stop_(false),
        verbose_(true), eps_(0.0000000001), iter_(0), use_weights_(false), scale_(0.0), scale_estimator_(), weight_function_() {}
  virtual ~NLLSSolver() {}

  void optimize(ModelType& model) {
    if (method_ == GaussNewton) optimizeGaussNewton(model);
    else { fprintf(stderr, "shim: LevenbergMarquardt not restated\n"); abort(); }
  }

  void optimizeGaussNewton(ModelType& model) {
    // Compute weight scale
    if (use_weights_) computeResiduals(model, false, true);
    // Save the old model to rollback in case of unsuccessful update
    ModelType old_model(model);
    // perform iterative estimation
    for (iter_ = 0; iter_ < n_iter_; ++iter_) {
      rho_ = 0;
      startIteration();
      H_.setZero();
      Jres_.setZero();
      // compute initial error
      n_meas_ = 0;
      double new_chi2 = computeResiduals(model, true, false);
      // add prior
      if (have_prior_) applyPrior(model);
      // solve the linear system
      if (!solve()) {
        // matrix was singular and could not be computed
        stop_ = true;
      }
      // check if error increased since last optimization
      if ((iter_ > 0 && new_chi2 > chi2_) || stop_) {
        model = old_model;  // rollback
        break;
      }
      // update the model
      ModelType new_model;
      update(model, new_model);
      old_model = model;
      model = new_model;
      chi2_ = new_chi2;
      finishIteration();
      // stop when converged, i.e. update step too small
      if (norm_max(x_) <= eps_) break;
    }
  }

  void reset() {
    have_prior_ = false;
    chi2_ = 1e10;
    mu_ = mu_init_;
    nu_ = nu_init_;
    n_meas_ = 0;
    n_iter_ = n_iter_init_;
    iter_ = 0;
    stop_ = false;
  }
  const double& getChi2() const { return chi2_; }
  const Matrix<double, D, D>& getInformationMatrix() const { return H_; }
};
}  // namespace vk
