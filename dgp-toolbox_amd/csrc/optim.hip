// Adam on the unconstrained variables of GPflow-style Parameters (tf.optimizers.Adam, dgp.py:267-276):
// Softplus for kernel variance / lengthscales, Softplus + Shift(1e-6) for the Gaussian variance,
// FillTriangular for q_sqrt (only the lower-triangular entries are variables), identity otherwise.
#include "dgp_internal.h"

namespace dgp {

__global__ void adam_kernel(double* __restrict__ params, const double* __restrict__ grad_elbo, double* __restrict__ m,
                            double* __restrict__ v, const ParamSeg* __restrict__ segs, int nseg, long total, double lr_t,
                            double beta1, double beta2, double eps) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int s = 0;
  while (s + 1 < nseg && idx >= segs[s + 1].off) ++s;
  const ParamSeg sg = segs[s];
  if (!sg.trainable) return;
  const long k = idx - sg.off;
  if (sg.transform == TR_TRIL) {
    const long M = sg.rows;
    const long i = (k / M) % M, j = k % M;
    if (j > i) return;
  }
  const double x = params[idx];
  double g = -grad_elbo[idx];          // objective = -ELBO
  double u = x;
  if (sg.transform == TR_SOFTPLUS || sg.transform == TR_SOFTPLUS_SHIFT) {
    const double xs = x - (sg.transform == TR_SOFTPLUS_SHIFT ? kLikVarLower : 0.0);
    u = xs + log(-expm1(-xs));         // softplus^-1
    g *= 1.0 / (1.0 + exp(-u));        // dx/du = sigmoid(u)
  }
  const double mn = beta1 * m[idx] + (1.0 - beta1) * g;
  const double vn = beta2 * v[idx] + (1.0 - beta2) * g * g;
  m[idx] = mn;
  v[idx] = vn;
  u -= lr_t * mn / (sqrt(vn) + eps);
  double xn = u;
  if (sg.transform == TR_SOFTPLUS || sg.transform == TR_SOFTPLUS_SHIFT) {
    xn = fmax(u, 0.0) + log1p(exp(-fabs(u)));
    if (sg.transform == TR_SOFTPLUS_SHIFT) xn += kLikVarLower;
  }
  params[idx] = xn;
}

hipError_t adam_apply(hipStream_t st, double* params, const double* grad_elbo, double* m, double* v,
                      const ParamSeg* segs_dev, int nseg, long total, double lr_t, double beta1, double beta2,
                      double eps) {
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, params, grad_elbo, m, v,
                     segs_dev, nseg, total, lr_t, beta1, beta2, eps);
  return hipGetLastError();
}

}  // namespace dgp
