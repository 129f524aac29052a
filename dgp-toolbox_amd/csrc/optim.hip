// Adam on the unconstrained variables of GPflow-style Parameters (tf.optimizers.Adam, dgp.py:267-276):
// Softplus for kernel variance / lengthscales, Softplus + Shift(1e-6) for the Gaussian variance,
// FillTriangular for q_sqrt (only the lower-triangular entries are variables), identity otherwise.
#include "dgp_internal.h"

namespace dgp {

__global__ void adam_kernel(double* __restrict__ params, const double* __restrict__ grad_elbo, double* __restrict__ m,
                            double* __restrict__ v, const ParamSeg* __restrict__ segs, int nseg, long total, double lr_t,
                            double beta1, double beta2, double eps, const double* __restrict__ t_dev) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  if (t_dev) {                         // Keras Adam: lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t), t from device memory
    const double t = t_dev[0];
    lr_t = lr_t * sqrt(1.0 - pow(beta2, t)) / (1.0 - pow(beta1, t));
  }
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
                      double eps, const double* t_dev) {
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, params, grad_elbo, m, v,
                     segs_dev, nseg, total, lr_t, beta1, beta2, eps, t_dev);
  return hipGetLastError();
}

// a[0 .. na) = 0 and b[0 .. nb) = 0 in one launch (the four ELBO / KL scalars and the partial-sum buffer at the head of an
// iteration: two fill launches of ~5 us each were two of a small model's ~23)
__global__ void zero_two_kernel(double* __restrict__ a, long na, double* __restrict__ b, long nb) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < na) a[i] = 0.0;
  if (i < nb) b[i] = 0.0;
}
hipError_t zero_two(hipStream_t st, double* a, long na, double* b, long nb) {
  const long n = na > nb ? na : nb;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(zero_two_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, na, b, nb);
  return hipGetLastError();
}
// the tail of a gradient evaluation: d ELBO / d likelihood variance out of the partial-sum buffer, ELBO = data term - sum KL
__global__ void finish_tail_kernel(const double* __restrict__ acc, double* __restrict__ scal, double* __restrict__ grad_last) {
  grad_last[0] = acc[1];
  scal[1] = acc[0] - scal[0];
}
hipError_t finish_tail_launch(hipStream_t st, const double* acc, double* scal, double* grad_last) {
  hipLaunchKernelGGL(finish_tail_kernel, dim3(1), dim3(1), 0, st, acc, scal, grad_last);
  return hipGetLastError();
}

// Q' = sum_p cbar_p c_p^T without a pass over the points.  With cbar = u mbar + sum_d 2 vbar_d (W_d W_d^T - I) c (SURVEY App. C
// step 3):  Q' = u du^T + sum_d (W_d dW_d^T - 2 G_d),  du = sum_p c_p mbar_p^T,  G_d = sum_p vbar_pd c_p c_p^T (symmetric),  dW_d =
// 2 G_d W_d - all of them already on hand when a layer's gradient is finished (the DATA terms, before the KL terms join).
// T[d] = dW_d W_d^T arrives as a batched product; this writes  Q'[i][j] = sum_d (u[i][d] du[j][d] + T[d][j][i] - 2 G[d][i][j]),
// the d terms in order (no atomics: reproducible).
__global__ void qprime_from_sums_kernel(const double* __restrict__ T, const double* __restrict__ G, const double* __restrict__ u,
                                        const double* __restrict__ du, int Mp, int D, double* __restrict__ Q) {
  __shared__ double tile[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const long MM = (long)Mp * Mp;
  for (int d = 0; d < D; ++d) {
    __syncthreads();
    for (int r = ty; r < 32; r += 8) tile[r][tx] = T[d * MM + (long)(bj * 32 + r) * Mp + bi * 32 + tx];    // T[d] tile (bj, bi)
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q;
      const long i = bi * 32 + r, j = bj * 32 + tx;
      acc[q] += u[i * D + d] * du[j * D + d] + tile[tx][r] - 2.0 * G[d * MM + i * Mp + j];
    }
  }
  for (int q = 0; q < 4; ++q) Q[(long)(bi * 32 + ty + 8 * q) * Mp + bj * 32 + tx] = acc[q];
}
// out[i][j] = sum_d (a[i][d] b[j][d] + T[d][i][j]): the K = D Mp product  a b^T + sum_d dW_d W_d^T  of the finish chain as ONE
// batched Mp^3 product (T[d] = dW_d W_d^T, 32 x 32-tile kernel) plus this sum over d, in order - instead of a split-K launch of the
// 128 x 64 engine that added its D partial products with atomics (35 us, and an order that changed from run to run)
__global__ void sum_dprod_kernel(const double* __restrict__ T, const double* __restrict__ a, const double* __restrict__ b, int Mp, int D,
                                 double* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long MM = (long)Mp * Mp;
  if (idx >= MM) return;
  const int i = (int)(idx / Mp), j = (int)(idx % Mp);
  double acc = 0.0;
  for (int d = 0; d < D; ++d) acc += a[i * D + d] * b[j * D + d] + T[d * MM + idx];
  out[idx] = acc;
}
hipError_t sum_dprod(hipStream_t st, const double* T, const double* a, const double* b, int Mp, int D, double* out) {
  const long MM = (long)Mp * Mp;
  hipLaunchKernelGGL(sum_dprod_kernel, dim3((unsigned)((MM + 255) / 256)), dim3(256), 0, st, T, a, b, Mp, D, out);
  return hipGetLastError();
}

hipError_t qprime_from_sums(hipStream_t st, const double* T, const double* G, const double* u, const double* du, int Mp, int D,
                            double* Q) {
  hipLaunchKernelGGL(qprime_from_sums_kernel, dim3((unsigned)(Mp / 32), (unsigned)(Mp / 32)), dim3(256), 0, st, T, G, u, du, Mp, D, Q);
  return hipGetLastError();
}

__global__ void iter_bump_kernel(double* it, int seed_inc, int t_inc) {
  uint64_t* seed = reinterpret_cast<uint64_t*>(it);
  seed[0] += (uint64_t)seed_inc;
  it[1] += (double)t_inc;
}
hipError_t iter_bump(hipStream_t st, double* it, int seed_inc, int t_inc) {
  hipLaunchKernelGGL(iter_bump_kernel, dim3(1), dim3(1), 0, st, it, seed_inc, t_inc);
  return hipGetLastError();
}
__global__ void iter_log_elbo_kernel(double* it, const double* scal, double* elbo_log, int cap) {
  const int i = (int)it[2];
  if (i >= 0 && i < cap) elbo_log[i] = scal[1];
  it[2] = (double)(i + 1);
}
hipError_t iter_log_elbo(hipStream_t st, double* it, const double* scal, double* elbo_log, int cap) {
  hipLaunchKernelGGL(iter_log_elbo_kernel, dim3(1), dim3(1), 0, st, it, scal, elbo_log, cap);
  return hipGetLastError();
}

// ---- transport form of the partial-sum buffer (dgp_ctx.h: acc_xfer).  A lower triangle (diagonal included) of an Mp x Mp
// matrix, Mp even, in rectangular packed form [Mp / 2][Mp + 1]: row r = row Mp/2 + r of the triangle (Mp/2 + r + 1 entries)
// followed by row Mp/2 - 1 - r (Mp/2 - r entries) - no square roots, exactly Mp (Mp + 1) / 2 doubles.
__device__ __forceinline__ void rfp_index(long e, long Mp, long& i, long& j) {
  const long h = Mp / 2, r = e / (Mp + 1), c = e - r * (Mp + 1);
  if (c <= h + r) { i = h + r; j = c; }
  else { i = h - 1 - r; j = c - (h + r + 1); }
}
__global__ void acc_xfer_kernel(const long* __restrict__ tab, int nseg, double* __restrict__ sq, double* __restrict__ packed, int unpack) {
  const long* t = tab + 4L * blockIdx.y;
  const long kind = t[0], so = t[1], po = t[2], n = t[3];
  const long count = kind ? n * (n + 1) / 2 : n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x) {
    if (!kind) {
      if (unpack) sq[so + e] = packed[po + e]; else packed[po + e] = sq[so + e];
    } else {
      long i, j;
      rfp_index(e, n, i, j);
      if (unpack) {                 // mirrored: the summed matrices are symmetric, and some consumers read the whole square
        const double v = packed[po + e];
        sq[so + i * n + j] = v;
        sq[so + j * n + i] = v;
      } else {
        packed[po + e] = sq[so + i * n + j];
      }
    }
  }
}
hipError_t acc_xfer(hipStream_t st, const long* tab, int nseg, double* square, double* packed, bool unpack) {
  if (nseg <= 0) return hipSuccess;
  hipLaunchKernelGGL(acc_xfer_kernel, dim3(32, (unsigned)nseg), dim3(256), 0, st, tab, nseg, square, packed, unpack ? 1 : 0);
  return hipGetLastError();
}

}  // namespace dgp
