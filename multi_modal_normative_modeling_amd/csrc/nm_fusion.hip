// nm_fusion.hip -- the expert-fusion operators of the reference's class surface as stand-alone launches (gfx950).
//
// Inside a train step the fusion lives in the step kernel (nmhip.hip: fuse_fwd / fuse_bwd).  The reference also exposes
// it as public methods that take tensors and return tensors -- cVAE_multimodal.combine_latent (cVAE.py:1144-1164),
// product_of_experts / mixture_of_experts / mixture_of_product_of_experts (:1118-1126, classes :986-1083), the same on
// cVAE_multimodal_regression (:2265-2307), mvtCAE's variants (ProductOfExperts2 :1481-1489, clamp :1823,
// total_correlation :1859-1866) and mmJSD.combine_latent (:1399-1402).  These are the forward-only entry points behind
// those methods: elementwise over [M][n] fp32 tensors, one element per thread, 16-byte accesses when n % 4 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "nmhip.h"

namespace {

constexpr int FUSE_MAX = 8;     // experts per call

struct FuseArgs {
  const float* mus;             // [M][n]
  const float* vars;            // [M][n]  variances, or log-variances with in_log
  const float* alpha_raw;       // [M] un-normalised gPoE weights (alpha_m_list), or nullptr
  float* out_mu;                // [n]
  float* out_var;               // [n]  variance, or its logarithm with out_log
  int64_t n;
  int M, combine, in_log, out_log, single_bypass;
  float var_floor;
};

__device__ __forceinline__ void fuse_one(const FuseArgs& a, const float (&al)[FUSE_MAX], const float (&mu)[FUSE_MAX],
                                         const float (&vin)[FUSE_MAX], float& omu, float& ovar) {
  const int M = a.M;
  if (M == 1 && a.single_bypass) { omu = mu[0]; ovar = a.in_log ? expf(vin[0]) : vin[0]; return; }   // cVAE.py:1146-1147
  float S = 0.f, Smu = 0.f, sm = 0.f, sv = 0.f;
#pragma unroll
  for (int m = 0; m < FUSE_MAX; ++m) {
    if (m < M) {
      const float var = a.in_log ? expf(vin[m]) : vin[m];
      const float w = (a.combine == NM_COMBINE_GPOE) ? al[m] / var : 1.0f / var;
      S += w; Smu += mu[m] * w;
      sm += mu[m]; sv += var;
    }
  }
  if (a.combine == NM_COMBINE_MOE) { omu = sm / (float)M; ovar = sv / (float)M; }
  else {
    omu = Smu / S; ovar = 1.0f / S;
    if (a.combine == NM_COMBINE_MOPOE) { omu = (sm + omu) / (float)(M + 1); ovar = (sv + ovar) / (float)(M + 1); }
  }
}

__global__ void combine_latent_kernel(FuseArgs a) {
  float al[FUSE_MAX];
  {
    // softmax of alpha_m_list (cVAE.py:1155): M scalars, every thread evaluates it
    float mx = -INFINITY, s = 0.f;
#pragma unroll
    for (int m = 0; m < FUSE_MAX; ++m) if (m < a.M && a.alpha_raw) mx = fmaxf(mx, a.alpha_raw[m]);
#pragma unroll
    for (int m = 0; m < FUSE_MAX; ++m) { al[m] = (m < a.M && a.alpha_raw) ? expf(a.alpha_raw[m] - mx) : 0.f; s += al[m]; }
#pragma unroll
    for (int m = 0; m < FUSE_MAX; ++m) al[m] = s > 0.f ? al[m] / s : 0.f;
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  float mu[FUSE_MAX], vin[FUSE_MAX];
#pragma unroll
  for (int m = 0; m < FUSE_MAX; ++m) {
    mu[m] = m < a.M ? a.mus[(int64_t)m * a.n + i] : 0.f;
    vin[m] = m < a.M ? a.vars[(int64_t)m * a.n + i] : 1.f;
  }
  float omu, ovar;
  fuse_one(a, al, mu, vin, omu, ovar);
  if (a.out_log) ovar = logf(ovar);                                    // ProductOfExperts2 returns log(var), cVAE.py:1487
  if (a.var_floor > 0.f) ovar = fmaxf(ovar, a.var_floor);              // torch.clamp(variance_multimodal, min=1e-6), :1823
  a.out_mu[i] = omu;
  a.out_var[i] = ovar;
}

// mvtCAE.total_correlation (cVAE.py:1859-1866): sum over the latent columns of
//   [logsumexp_rows(qz_x[:, i]) - mean(that scalar)]  -  mean_j logsumexp_rows(qz_xs[j][:, i]);
// the first bracket is a scalar minus its own mean, i.e. exactly zero.  One wave per (expert, column): max and sum over
// the rows by shuffles in a fixed order, the column results summed by one thread in index order (reproducible).
__global__ void total_correlation_kernel(const float* __restrict__ qz_xs, int M, int B, int Z, float* __restrict__ out) {
  __shared__ float lse[FUSE_MAX * 128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int col = wave; col < M * Z; col += nw) {
    const int m = col / Z, z = col - m * Z;
    float mx = -3.0e38f;
    for (int r = lane; r < B; r += 64) mx = fmaxf(mx, qz_xs[((int64_t)m * B + r) * Z + z]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sx = 0.f;
    for (int r = lane; r < B; r += 64) sx += expf(qz_xs[((int64_t)m * B + r) * Z + z] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sx += __shfl_xor(sx, o, 64);
    if (lane == 0) lse[col] = mx + logf(sx);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tc = 0.f;
    for (int z = 0; z < Z; ++z) {
      float s = 0.f;
      for (int m = 0; m < M; ++m) s += lse[m * Z + z];
      tc -= s / (float)M;
    }
    out[0] = tc;
  }
}

}  // namespace

extern "C" {

int nm_combine_latent(const float* mus, const float* variances, int M, int64_t n, int combine, const float* alpha_raw,
                      int single_bypass, int in_log, int out_log, float var_floor, float* out_mu, float* out_var,
                      void* stream) {
  if (!mus || !variances || !out_mu || !out_var) return -1;
  if (M < 1 || M > FUSE_MAX) return -2;
  if (n < 1) return -8;
  if (combine < NM_COMBINE_POE || combine > NM_COMBINE_MOPOE) return -9;
  if (combine == NM_COMBINE_GPOE && !alpha_raw) return -1;
  FuseArgs a{mus, variances, alpha_raw, out_mu, out_var, n, M, combine, in_log, out_log, single_bypass, var_floor};
  const int threads = 256;
  const int64_t blocks = (n + threads - 1) / threads;
  if (blocks > 0x7fffffff) return -8;
  hipLaunchKernelGGL(combine_latent_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int nm_total_correlation(const float* qz_xs, int M, int B, int Z, float* out, void* stream) {
  if (!qz_xs || !out) return -1;
  if (M < 1 || M > FUSE_MAX) return -2;
  if (B < 1 || Z < 1 || Z > 128) return -5;
  hipLaunchKernelGGL(total_correlation_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, qz_xs, M, B, Z, out);
  return (int)hipGetLastError();
}

}  // extern "C"
