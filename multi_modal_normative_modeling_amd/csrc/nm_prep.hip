// nm_prep.hip -- input preparation on the device: a fold's ROI tables are built from the raw
// cohort resident in HBM, with no pass over the data on the host.
//
//   nm_prep_scaler_fit   sklearn RobustScaler().fit on the fold's rows: per ROI the median and the 25..75 % range
//                        (multimodal_kfold_train_cvae_supervised.py:101-102); one workgroup per column sorts the
//                        column's rows in LDS (bitonic, <= 8192 rows) and interpolates exactly as numpy does
//   nm_prep_onehot       pd.qcut(col.rank(method='first'), q, labels=range(q)) one-hot blocks for AGE (27) and
//                        PTGENDER (2) (:107-126): stable rank by an LDS sort of (value, row) pairs, bins by the edges
//                        numpy computes for the ranks 1..n (they depend on n only: the host passes them in)
//   nm_pack_table_raw    (x - center) / scale in fp64, cast to fp32 (RobustScaler.transform + astype), early-fusion
//                        column concat of several source tables (early_fusion_modalities.py:23-32), and the packing
//                        of nm_pack_table (xb chunk images, fp32 copy, cz block) in one kernel
//
// Integer / ordering work is exact; the floating-point steps are the same IEEE fp64 operations numpy performs, in the
// same order (no contraction into FMAs), so the results are bit-identical to prep.py (tests/test_gpu_prep.py).
// Inputs are assumed free of NaNs (the reference's nanmedian / nanpercentile then equal median / percentile).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nmhip.h"

namespace {

constexpr int PREP_WG = 512;
constexpr int PREP_MAX_N = 8192;
constexpr int LDX = 72, XCH = 64, ROWS = NM_BATCH;

// order-preserving map double -> uint64 (ascending)
__device__ __forceinline__ uint64_t key_of(double v) {
  uint64_t u = (uint64_t)__double_as_longlong(v);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double val_of(uint64_t k) {
  uint64_t u = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ const double* src_col(const double* const* srcs, const int32_t* src_D, int n_src, int d, int* pitch) {
  int off = 0;
  for (int s = 0; s < n_src; ++s) {
    if (d < off + src_D[s]) { *pitch = src_D[s]; return srcs[s] + (d - off); }
    off += src_D[s];
  }
  *pitch = src_D[n_src - 1];
  return srcs[n_src - 1];
}

// ascending bitonic sort of P (power of two) 64-bit keys in LDS, optional payload
template <bool PAYLOAD>
__device__ __forceinline__ void bitonic(uint64_t* k, uint32_t* p, int P) {
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        uint64_t a = k[lo], b = k[hi];
        bool swap = up ? (a > b) : (a < b);
        if (PAYLOAD && a == b) { const uint32_t pa = p[lo], pb = p[hi]; swap = up ? (pa > pb) : (pa < pb); }
        if (swap) {
          k[lo] = b; k[hi] = a;
          if (PAYLOAD) { const uint32_t x = p[lo]; p[lo] = p[hi]; p[hi] = x; }
        }
      }
    }
  }
  __syncthreads();
}

// numpy's linear-interpolated quantile of a SORTED array a[0..n): virtual index n q + (1 - q) - 1, _lerp()
__device__ __forceinline__ double np_quantile_sorted(const uint64_t* keys, int n, double q) {
  const double vi = __dadd_rn(__dadd_rn(__dmul_rn((double)n, q), __dadd_rn(1.0, __dmul_rn(q, -1.0))), -1.0);
  double fl = floor(vi);
  int lo = (int)fl;
  lo = lo < 0 ? 0 : (lo > n - 1 ? n - 1 : lo);
  const int hi = lo + 1 > n - 1 ? n - 1 : lo + 1;
  const double g = __dadd_rn(vi, -fl);
  const double a = val_of(keys[lo]), b = val_of(keys[hi]);
  const double diff = __dadd_rn(b, -a);
  if (g >= 0.5) return __dadd_rn(b, -__dmul_rn(diff, __dadd_rn(1.0, -g)));
  return __dadd_rn(a, __dmul_rn(diff, g));
}

__global__ __launch_bounds__(PREP_WG) void scaler_fit_kernel(const double* const* srcs, const int32_t* src_D, int n_src, int D,
                                                             const int32_t* rows, int n, double* center, double* scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  const int d = blockIdx.x;
  if (d >= D) return;
  int P = 1;
  while (P < n) P <<= 1;
  int pitch;
  const double* col = src_col(srcs, src_D, n_src, d, &pitch);
  for (int i = threadIdx.x; i < P; i += blockDim.x)
    keys[i] = i < n ? key_of(col[(int64_t)rows[i] * pitch]) : 0xFFFFFFFFFFFFFFFFull;
  bitonic<false>(keys, nullptr, P);
  if (threadIdx.x == 0) {
    // np.nanmedian: middle element, or the mean of the two middle ones ((a + b) / 2)
    double med;
    if (n & 1) med = val_of(keys[n >> 1]);
    else med = __ddiv_rn(__dadd_rn(val_of(keys[(n >> 1) - 1]), val_of(keys[n >> 1])), 2.0);
    const double q25 = np_quantile_sorted(keys, n, 0.25), q75 = np_quantile_sorted(keys, n, 0.75);
    double sc = __dadd_rn(q75, -q25);
    if (sc < 10.0 * 2.220446049250313e-16) sc = 1.0;          // sklearn _handle_zeros_in_scale
    center[d] = med;
    scale[d] = sc;
  }
}

// one workgroup per covariate (0: AGE, 1: PTGENDER): stable rank, bin by the edges, write the one-hot block
__global__ __launch_bounds__(PREP_WG) void onehot_kernel(const double* age, const double* gender, const int32_t* rows, int n,
                                                         const double* age_edges, int age_bins, const double* gender_edges,
                                                         int gender_bins, float* c_out, int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int P = 1;
  while (P < n) P <<= 1;
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  uint32_t* idx = reinterpret_cast<uint32_t*>(keys + P);
  const bool is_age = blockIdx.x == 0;
  const double* col = is_age ? age : gender;
  const double* edges = is_age ? age_edges : gender_edges;
  const int q = is_age ? age_bins : gender_bins, col0 = is_age ? 0 : age_bins;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    keys[i] = i < n ? key_of(col[rows[i]]) : 0xFFFFFFFFFFFFFFFFull;
    idx[i] = (uint32_t)i;
  }
  bitonic<true>(keys, idx, P);
  // sorted position s (0-based) of local row idx[s]: rank = s + 1; bin = searchsorted(edges, rank, 'left') - 1,
  // rank <= edges[0] -> 0, clipped to [0, q)
  for (int s = threadIdx.x; s < n; s += blockDim.x) {
    const double r = (double)(s + 1);
    int b = 0;
    while (b < q + 1 && edges[b] < r) ++b;          // first edge >= r  (searchsorted side = 'left')
    b -= 1;
    if (r <= edges[0]) b = 0;
    b = b < 0 ? 0 : (b > q - 1 ? q - 1 : b);
    float* out = c_out + (int64_t)idx[s] * C + col0;
    for (int k = 0; k < q; ++k) out[k] = k == b ? 1.0f : 0.0f;
  }
}

// scaled fp32 value of (table row r, column k): (x - center) / scale in fp64, then the cast
__device__ __forceinline__ float scaled(const double* const* srcs, const int32_t* src_D, int n_src, const int32_t* rows, int r,
                                        int k, const double* center, const double* scale) {
  int pitch;
  const double* col = src_col(srcs, src_D, n_src, k, &pitch);
  const double v = col[(int64_t)rows[r] * pitch];
  return (float)__ddiv_rn(__dadd_rn(v, -center[k]), scale[k]);
}

__global__ void pack_raw_kernel(const double* const* srcs, const int32_t* src_D, int n_src, const int32_t* rows, int n_rows,
                                const double* center, const double* scale, const float* cc, int rows_alloc, int D, int C, int Kx,
                                uint16_t* xb, float* xf, int xp, uint16_t* cz, int Cz) {
  const int nch = (Kx + XCH - 1) / XCH;
  const int64_t total = (int64_t)rows_alloc * nch * LDX;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % LDX);
    const int64_t q = i / LDX;
    const int rl = (int)(q % ROWS);
    const int64_t q2 = q / ROWS;
    const int kc = (int)(q2 % nch), tile = (int)(q2 / nch);
    const int r = tile * ROWS + rl, k = kc * XCH + j;
    float v = 0.f;
    if (r < n_rows && j < XCH) {
      if (k < D) v = scaled(srcs, src_D, n_src, rows, r, k, center, scale);
      else if (k < D + C) v = cc[(int64_t)r * C + (k - D)];
      else if (k == D + C) v = 1.0f;
    }
    __bf16 h = (__bf16)v;
    xb[i] = __builtin_bit_cast(uint16_t, h);
  }
  if (xf) {
    const int64_t tf = (int64_t)rows_alloc * xp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tf; i += (int64_t)gridDim.x * blockDim.x) {
      const int r = (int)(i / xp), k = (int)(i - (int64_t)r * xp);
      xf[i] = (r < n_rows && k < D) ? scaled(srcs, src_D, n_src, rows, r, k, center, scale) : 0.f;
    }
  }
  if (cz) {
    const int64_t tc = (int64_t)rows_alloc * Cz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tc; i += (int64_t)gridDim.x * blockDim.x) {
      const int r = (int)(i / Cz), k = (int)(i - (int64_t)r * Cz);
      float v = 0.f;
      if (r < n_rows) v = k < C ? cc[(int64_t)r * C + k] : (k == C ? 1.0f : 0.f);
      __bf16 h = (__bf16)v;
      cz[i] = __builtin_bit_cast(uint16_t, h);
    }
  }
}

}  // namespace

extern "C" {

int nm_prep_scaler_fit(const double* const* srcs_dev, const int32_t* src_D_dev, int n_src, int D, const int32_t* rows_dev,
                       int n_rows, double* center_dev, double* scale_dev, void* stream) {
  if (!srcs_dev || !src_D_dev || !rows_dev || !center_dev || !scale_dev) return -1;
  if (n_src < 1 || D < 1 || n_rows < 1 || n_rows > PREP_MAX_N) return -17;
  int P = 1;
  while (P < n_rows) P <<= 1;
  const int smem = P * 8;
  hipError_t e = hipFuncSetAttribute((const void*)scaler_fit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(scaler_fit_kernel, dim3(D), dim3(PREP_WG), smem, (hipStream_t)stream, srcs_dev, src_D_dev, n_src, D, rows_dev,
                     n_rows, center_dev, scale_dev);
  return (int)hipGetLastError();
}

int nm_prep_onehot(const double* age_dev, const double* gender_dev, const int32_t* rows_dev, int n_rows, const double* age_edges_dev,
                   int age_bins, const double* gender_edges_dev, int gender_bins, float* c_out_dev, void* stream) {
  if (!age_dev || !gender_dev || !rows_dev || !age_edges_dev || !gender_edges_dev || !c_out_dev) return -1;
  if (n_rows < 1 || n_rows > PREP_MAX_N || age_bins < 1 || gender_bins < 1) return -17;
  int P = 1;
  while (P < n_rows) P <<= 1;
  const int smem = P * 12;
  hipError_t e = hipFuncSetAttribute((const void*)onehot_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(onehot_kernel, dim3(2), dim3(PREP_WG), smem, (hipStream_t)stream, age_dev, gender_dev, rows_dev, n_rows,
                     age_edges_dev, age_bins, gender_edges_dev, gender_bins, c_out_dev, age_bins + gender_bins);
  return (int)hipGetLastError();
}

int nm_pack_table_raw(const double* const* srcs_dev, const int32_t* src_D_dev, int n_src, const int32_t* rows_dev, int n_rows,
                      const double* center_dev, const double* scale_dev, const float* c_dev, int rows_alloc, int D, int C, int Kx,
                      uint16_t* xb, float* x_f32_out, int x_pitch, uint16_t* cz_out, int Cz, void* stream) {
  if (!srcs_dev || !src_D_dev || !rows_dev || !center_dev || !scale_dev || !xb || (C > 0 && !c_dev)) return -1;
  if (Kx % 32 != 0 || Kx < D + C + 1 || rows_alloc < n_rows || rows_alloc % NM_BATCH != 0) return -7;
  if (x_f32_out && (x_pitch % 4 != 0 || x_pitch < D || x_pitch > Kx)) return -7;
  if (cz_out && (Cz % 8 != 0 || Cz < C + 1)) return -7;
  const int64_t total = nm_xb_elems(rows_alloc, Kx);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_raw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, srcs_dev, src_D_dev, n_src, rows_dev, n_rows,
                     center_dev, scale_dev, c_dev, rows_alloc, D, C, Kx, xb, x_f32_out, x_pitch, cz_out, Cz);
  return (int)hipGetLastError();
}

}  // extern "C"
