// Post-hoc metrics of the sweep on the device: the numbers the final RCCL gather carries.
//
//   nm_posthoc_metrics    per-subject deviation scores + class labels -> ROC-AUC, Youden-J threshold, accuracy,
//                         sensitivity, specificity, significance ratio
//                         (compute_classification_performance, multimodal_kfold_cvae_group_analysis_1x1.py:105-157;
//                          sklearn.metrics.roc_curve / auc as called there at :125-126)
//   nm_confusion_metrics  hard predictions + labels -> accuracy, auroc, sensitivity, specificity, f1, precision
//                         (evaluate, multimodal_kfold_cvae_nmpmcont.py:29-70)
//
// One workgroup per score set (a (fold, procedure) cell); sets are segments of one concatenated array.  The
// whole set lives in LDS: order-preserving 64-bit keys (score, label) are bitonic-sorted descending, label
// prefix sums give (tps, fps) at every distinct-score boundary exactly as sklearn's _binary_clf_curve builds
// them, and counts stay integers until the final divisions (fp64, IEEE) -- so counts, thresholds and the
// chosen operating point are bit-exact against the CPU restatement; the AUC is the exactly rounded value of
// the integer trapezoid sum.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "nmhip.h"

namespace {

constexpr int MT = 256;                 // threads per workgroup
constexpr int MAXN = NM_METRICS_MAX_N;  // scores per set (power of two)

__device__ __forceinline__ uint32_t desc_key(float x) {
  if (x == 0.0f) x = 0.0f;                               // -0 and +0 are one threshold (np.diff == 0)
  uint32_t b = __float_as_uint(x);
  uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // ascending order-preserving key
  return ~asc;                                           // ascending sort of this = descending scores
}
__device__ __forceinline__ float key_score(uint32_t k) {
  uint32_t asc = ~k;
  uint32_t b = (asc & 0x80000000u) ? (asc & 0x7FFFFFFFu) : ~asc;
  return __uint_as_float(b);
}

// inclusive prefix sum of v[0..npad) in place (npad a power of two >= MT or smaller), all threads call
__device__ __forceinline__ void block_scan(int32_t* v, int npad, int32_t* part) {
  const int per = (npad + MT - 1) / MT;
  const int t = threadIdx.x;
  const int lo = min(t * per, npad), hi = min(lo + per, npad);
  int32_t s = 0;
  for (int i = lo; i < hi; ++i) { s += v[i]; v[i] = s; }
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < MT; off <<= 1) {
    int32_t add = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  const int32_t base = (t > 0) ? part[t - 1] : 0;
  for (int i = lo; i < hi; ++i) v[i] += base;
  __syncthreads();
}

__global__ __launch_bounds__(MT) void posthoc_kernel(const float* __restrict__ scores, const int32_t* __restrict__ labels,
                                                     const int32_t* __restrict__ offsets, const double* __restrict__ thr_in,
                                                     double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* key = reinterpret_cast<uint64_t*>(smem);            // [MAXN]
  int32_t* cum = reinterpret_cast<int32_t*>(key + MAXN);        // [MAXN] label prefix sums (tps)
  int32_t* bnd = cum + MAXN;                                    // [MAXN] boundary flags -> compacted indices
  __shared__ int32_t part[MT];
  __shared__ double redJ[MT];
  __shared__ int32_t redI[MT];
  __shared__ long long redA[MT];

  const int s = blockIdx.x, t = threadIdx.x;
  const int base = offsets[s], n = offsets[s + 1] - base;
  double* o = out + (int64_t)s * NM_METRICS_STRIDE;
  const double qnan = __longlong_as_double(0x7FF8000000000000ll);
  if (n <= 0 || n > MAXN) {
    if (t < NM_METRICS_STRIDE) o[t] = qnan;
    return;
  }
  int npad = 2;
  while (npad < n) npad <<= 1;
  for (int i = t; i < npad; i += MT)
    key[i] = (i < n) ? (((uint64_t)desc_key(scores[base + i]) << 32) | (uint64_t)(labels[base + i] != 0)) : ~0ull;
  __syncthreads();
  // bitonic sort, ascending in key = descending in score
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < npad; i += MT) {
        int p = i ^ j;
        if (p > i) {
          uint64_t a = key[i], b = key[p];
          bool up = (i & k) == 0;
          if ((a > b) == up) { key[i] = b; key[p] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = t; i < npad; i += MT) {
    cum[i] = (i < n) ? (int32_t)(key[i] & 1ull) : 0;
    bnd[i] = (i < n && (i == n - 1 || (key[i] >> 32) != (key[i + 1] >> 32))) ? 1 : 0;   // distinct_value_indices + last
  }
  __syncthreads();
  block_scan(cum, npad, part);
  // compact the boundaries: bnd[j] = sorted position of the j-th threshold.  After the scan, position i is a
  // boundary iff its inclusive count differs from its predecessor's; targets lie at or below i, so every
  // thread first collects its (target, position) pairs, then all write after a barrier.
  block_scan(bnd, npad, part);
  const int K = bnd[npad - 1];
  __syncthreads();
  {
    const int per = (npad + MT - 1) / MT;
    const int lo = min(t * per, npad), hi = min(lo + per, npad);
    int32_t mypos[32], myidx[32];          // per <= MAXN / MT = 32
    int cnt = 0;
    for (int i = lo; i < hi; ++i) {
      int32_t c = bnd[i], p = (i > 0) ? bnd[i - 1] : 0;
      if (c != p) { mypos[cnt] = c - 1; myidx[cnt] = i; ++cnt; }
    }
    __syncthreads();
    for (int q = 0; q < cnt; ++q) bnd[mypos[q]] = myidx[q];
    __syncthreads();
  }
  const int P = cum[n - 1], Nn = n - P;
  if (P == 0 || Nn == 0) {                 // roc_curve is undefined with one class
    if (t < NM_METRICS_STRIDE) o[t] = qnan;
    if (t == 0) { o[6] = (double)P; o[7] = (double)Nn; }
    return;
  }
  // points j = 0..K-1: (fps_j, tps_j); origin (0,0) precedes them with threshold +inf (sklearn >= 1.3)
  double bestJ = 0.0;                      // the origin's J; a point must beat it strictly (argmax takes the first)
  int bestj = -1;
  long long area2 = 0;
  for (int j = t; j < K; j += MT) {
    const int i = bnd[j];
    const long long tp = cum[i], fp = (long long)i + 1 - tp;
    long long tp0 = 0, fp0 = 0;
    if (j > 0) { const int i0 = bnd[j - 1]; tp0 = cum[i0]; fp0 = (long long)i0 + 1 - tp0; }
    area2 += (fp - fp0) * (tp + tp0);
    bool kept = (j == 0) || (j == K - 1);
    if (!kept) {                           // drop_intermediate: keep only corners of the curve
      const int i1 = bnd[j + 1];
      const long long tp1 = cum[i1], fp1 = (long long)i1 + 1 - tp1;
      kept = (fp1 - 2 * fp + fp0 != 0) || (tp1 - 2 * tp + tp0 != 0);
    }
    if (kept) {
      const double J = (double)tp / (double)P - (double)fp / (double)Nn;      // tpr - fpr as numpy forms it
      if (J > bestJ) { bestJ = J; bestj = j; }                               // j ascends per thread: first max kept
    }
  }
  redJ[t] = bestJ; redI[t] = bestj; redA[t] = area2;
  __syncthreads();
  if (t == 0) {
    double bj = 0.0; int bi = -1; long long a2 = 0;
    for (int w = 0; w < MT; ++w) {
      a2 += redA[w];
      if (redI[w] >= 0 && (redJ[w] > bj || (redJ[w] == bj && bi >= 0 && redI[w] < bi))) { bj = redJ[w]; bi = redI[w]; }
    }
    const double auc = (double)a2 / (2.0 * (double)P * (double)Nn);
    double thr;
    long long TP, FP;
    if (thr_in) {
      thr = thr_in[s];
      TP = -1; FP = -1;                    // counted below by everyone
    } else if (bi < 0) {
      thr = __longlong_as_double(0x7FF0000000000000ll); TP = 0; FP = 0;
    } else {
      const int i = bnd[bi];
      thr = (double)key_score((uint32_t)(key[i] >> 32));
      TP = cum[i]; FP = (long long)i + 1 - TP;
    }
    o[0] = auc; o[1] = thr; o[5] = auc / (1.0 - auc); o[6] = (double)P; o[7] = (double)Nn;
    redA[0] = TP; redA[1] = FP;
  }
  __syncthreads();
  long long TP = redA[0], FP = redA[1];
  if (thr_in) {                            // given threshold: predicted = score >= thr
    const double thr = thr_in[s];
    int tp = 0, fp = 0;
    for (int i = t; i < n; i += MT) {
      const bool pos = (double)key_score((uint32_t)(key[i] >> 32)) >= thr;
      const bool lab = (key[i] & 1ull) != 0;
      tp += (pos && lab) ? 1 : 0;
      fp += (pos && !lab) ? 1 : 0;
    }
    __syncthreads();
    part[t] = tp; redI[t] = fp;
    __syncthreads();
    if (t == 0) {
      long long a = 0, b = 0;
      for (int w = 0; w < MT; ++w) { a += part[w]; b += redI[w]; }
      redA[0] = a; redA[1] = b;
    }
    __syncthreads();
    TP = redA[0]; FP = redA[1];
  }
  if (t == 0) {
    const long long FN = P - TP, TN = Nn - FP;
    o[2] = (double)(TP + TN) / (double)n;                // accuracy
    o[3] = (double)TP / (double)(TP + FN);               // recall / sensitivity
    o[4] = (double)TN / (double)(TN + FP);               // specificity
  }
}

__global__ __launch_bounds__(MT) void confusion_kernel(const int32_t* __restrict__ pred, const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ offsets, double* __restrict__ out) {
  __shared__ int32_t cnt[4][MT];
  const int s = blockIdx.x, t = threadIdx.x;
  const int base = offsets[s], n = offsets[s + 1] - base;
  int tp = 0, fp = 0, tn = 0, fn = 0;
  for (int i = t; i < n; i += MT) {
    const bool p = pred[base + i] != 0, l = labels[base + i] != 0;
    tp += (p && l); fp += (p && !l); tn += (!p && !l); fn += (!p && l);
  }
  cnt[0][t] = tp; cnt[1][t] = fp; cnt[2][t] = tn; cnt[3][t] = fn;
  __syncthreads();
  if (t == 0) {
    long long TP = 0, FP = 0, TN = 0, FN = 0;
    for (int w = 0; w < MT; ++w) { TP += cnt[0][w]; FP += cnt[1][w]; TN += cnt[2][w]; FN += cnt[3][w]; }
    double* o = out + (int64_t)s * NM_METRICS_STRIDE;
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    const double sens = (TP + FN) ? (double)TP / (double)(TP + FN) : 0.0;          // recall_score: 0 when undefined
    const double spec = (double)TN / (double)(TN + FP);                            // numpy division: nan when 0/0
    o[0] = n > 0 ? (double)(TP + TN) / (double)n : qnan;                           // accuracy_score
    // roc_auc_score on hard predictions = mean of the two rates; ValueError (-> nan) with one class present
    o[1] = ((TP + FN) && (TN + FP)) ? 0.5 * ((double)TP / (double)(TP + FN) + (double)TN / (double)(TN + FP)) : qnan;
    o[2] = sens;
    o[3] = spec;
    o[4] = (2 * TP + FP + FN) ? 2.0 * (double)TP / (double)(2 * TP + FP + FN) : 0.0;   // f1_score
    o[5] = (TP + FP) ? (double)TP / (double)(TP + FP) : 0.0;                       // precision_score
    o[6] = (double)(TP + FN);
    o[7] = (double)(TN + FP);
  }
}

constexpr int METRICS_SMEM = MAXN * (8 + 4 + 4);

}  // namespace

extern "C" {

int nm_posthoc_metrics(const float* scores, const int32_t* labels, const int32_t* offsets, int n_sets, int max_set,
                       const double* thr_in, double* out, void* stream) {
  if (!scores || !labels || !offsets || !out) return -1;
  if (n_sets < 1 || max_set < 1 || max_set > MAXN) return -12;
  hipError_t e = hipFuncSetAttribute((const void*)posthoc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, METRICS_SMEM);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(posthoc_kernel, dim3(n_sets), dim3(MT), METRICS_SMEM, (hipStream_t)stream, scores, labels, offsets,
                     thr_in, out);
  return (int)hipGetLastError();
}

int nm_confusion_metrics(const int32_t* pred, const int32_t* labels, const int32_t* offsets, int n_sets, double* out,
                         void* stream) {
  if (!pred || !labels || !offsets || !out) return -1;
  if (n_sets < 1) return -12;
  hipLaunchKernelGGL(confusion_kernel, dim3(n_sets), dim3(MT), 0, (hipStream_t)stream, pred, labels, offsets, out);
  return (int)hipGetLastError();
}

}  // extern "C"
