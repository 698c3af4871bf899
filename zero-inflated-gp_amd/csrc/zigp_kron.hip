// Kronecker (space x time) variant -- placeholder until the factored kernels land (this round).
#include "zigp_ctx.h"
extern "C" {
int zigp_kron_elbo(zigp_ctx* c, const zigp_kron_params*, const double*, const double*, int64_t, double, double, double, int32_t,
                   double*, double*, zigp_kron_grads*) {
  if (!c) return ZIGP_EARG;
  c->err = "zigp_kron_elbo: not implemented yet";
  return ZIGP_EARG;
}
int zigp_kron_predict(zigp_ctx* c, const zigp_kron_params*, const double*, int64_t, double, double, double*) {
  if (!c) return ZIGP_EARG;
  c->err = "zigp_kron_predict: not implemented yet";
  return ZIGP_EARG;
}
}
