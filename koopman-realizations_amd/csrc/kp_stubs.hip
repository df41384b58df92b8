// Entry points not implemented yet in this build: they fail loudly.
#include "kp_internal.h"

int kp_lasso_dev(kp_ctx* ctx, const double*, const double*, int, int, double, int, double, double*, int*) {
  return ctx->fail(KP_ERR_ARG, "kp_fit_lasso: not implemented in this build");
}
extern "C" int kp_fit_lasso(kp_ctx* ctx, const double*, const double*, int, int, double, int, double, double*, int*) {
  return ctx ? ctx->fail(KP_ERR_ARG, "kp_fit_lasso: not implemented in this build") : KP_ERR_ARG;
}
extern "C" int kp_model_project(kp_ctx* ctx, const double*, const double*, const double*, int, int, double*, double*, double*) {
  return ctx ? ctx->fail(KP_ERR_ARG, "kp_model_project: not implemented in this build") : KP_ERR_ARG;
}
extern "C" int kp_rollout(kp_ctx* ctx, int, int, const double*, const double*, int, int, const double*, const double*, int, int, double*) {
  return ctx ? ctx->fail(KP_ERR_ARG, "kp_rollout: not implemented in this build") : KP_ERR_ARG;
}
extern "C" int kp_mpc_create(kp_ctx* ctx, int, const double*, const double*, int, int, int, const double*, int, double, double,
                             const double*, const double*, const double*, double, double, kp_mpc**) {
  return ctx ? ctx->fail(KP_ERR_ARG, "kp_mpc_create: not implemented in this build") : KP_ERR_ARG;
}
extern "C" int kp_mpc_destroy(kp_mpc*) { return KP_OK; }
extern "C" int kp_mpc_dims(const kp_mpc*, int*, int*) { return KP_ERR_ARG; }
extern "C" int kp_mpc_step(kp_mpc*, const double*, const double*, const double*, int, double*, int*) { return KP_ERR_ARG; }
extern "C" int kp_mpc_step_zeta(kp_mpc*, const kp_basis*, const double*, const double*, const double*, int, double*, double*, int*) { return KP_ERR_ARG; }
extern "C" int kp_mpc_step_batch(kp_mpc*, int, const double*, const double*, const double*, double*, int*) { return KP_ERR_ARG; }
extern "C" int kp_mpc_last_qp(kp_mpc*, double*, double*, double*, double*) { return KP_ERR_ARG; }
extern "C" int kp_qp_solve(kp_ctx* ctx, const double*, const double*, const double*, const double*, int, int, double*, int*) {
  return ctx ? ctx->fail(KP_ERR_ARG, "kp_qp_solve: not implemented in this build") : KP_ERR_ARG;
}
