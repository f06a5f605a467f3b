// nmhip.hip -- conditional-VAE train step / forward / deviation pass for MI355X (gfx950, CDNA4).
//
// One 512-thread workgroup (8 wavefronts of 64 lanes, two per SIMD: 256 VGPRs each) owns one model ("job") and
// runs whole train steps for it: encoder MLPs -> expert fusion -> reparameterisation -> decoder
// MLPs -> Gaussian NLL + KL -> backward -> Adam, with no inter-workgroup communication.  The
// sweep fills the chip with independent jobs (one workgroup per CU), see DESIGN.md.
//
// Data placement per workgroup
//   LDS  P [256][136] bf16 : the running activation / delta of the layer chain (updated in place)
//        Q [256][136] bf16 : two [128][136] weight-image halves, or a saved activation, or x-chunk stages, or the
//                            delta chunk + an output-chunk slot + transposition patches of the decoder output layer
//        S 18 KiB          : the other output-chunk slot / overflow of the second x stage + patches
//   HBM  fp32 parameters + Adam moments as 16 x 16 tiles (1 KiB each, lane-linear in the Adam units);
//        bf16 shadow images of every weight matrix in exactly the LDS layout (LDS-DMA copies, requested one phase
//        ahead); workspace: fp32 latent statistics, bf16 activation images saved for the backward pass.
//
// Every contraction is a v_mfma_f32_16x16x32_bf16 (fp32 accumulate), issued "transposed":
// the FEATURE index of the result lives in the accumulator registers (4 consecutive features per
// lane) and the batch ROW on the lane, so every epilogue touches 8 or 16 contiguous bytes:
//   forward  out[r][n] = sum_k P[r][k] W[n][k]      A = ds_read_b128 of the weight image, B = ds_read_b128 of P rows
//   dgrad    din[r][k] = sum_n P[r][n] W[n][k]      A = the same image through ds_read_b64_tr_b16, B = P rows
//   wgrad    dW[n][k]  = sum_r P[r][n] Q[r][k]      A, B = ds_read_b64_tr_b16 (transposing LDS read)
//
// Reference semantics restated here (paths relative to the reference checkout):
//   Encoder/Decoder            cVAE.py:140-206        expert fusion  cVAE.py:986-1083, 1144-1164
//   reparameterise / KL / LL   cVAE.py:14-15, 1130-1142
//   forward_multimodal / loss  cVAE.py:1166-1196      Adam           cVAE.py:1111-1116
//   deviation (x - x_hat)^2    multimodal_kfold_train_cvae_supervised_regression.py:183-188
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "nmhip.h"

namespace {

constexpr int NWM = 2;           // wave grid: row groups
constexpr int NWN = 4;           //            feature-tile groups
constexpr int NWAVES = NWM * NWN;
constexpr int WG = NWAVES * 64;  // threads per workgroup
constexpr int RT = NM_BATCH / (NWM * 16);   // 16-row tiles per wave
constexpr int WROWS = RT * 16;   // rows per wave
constexpr int ROWS = NM_BATCH;   // 256 rows per tile
constexpr int PW = 128;          // padded feature width held in P/Q
constexpr int LDP = 136;         // P/Q row pitch (elements): +8 breaks the 256-B bank period
constexpr int LDX = 72;          // row pitch of a staged 64-column x chunk inside Q
constexpr int XCH = 64;          // columns per staged x chunk
constexpr int STAGE_FLOATS = 4608;   // S: 18 KiB (an output-chunk blob, a [64][136] bf16 half tile, or overflow + patches)
constexpr int IMG_ROWS = 128;
constexpr int IMG_BYTES = IMG_ROWS * LDP * 2;        // 34,816: a [128][136] bf16 weight image = one half of Q
constexpr int VEC_BYTES = 1024;                      // the fp32 vectors that travel with an image (one DMA piece)
constexpr int BLOB_BYTES = IMG_BYTES + VEC_BYTES;    // 35,840
constexpr int XIMG_BYTES = ROWS * LDX * 2;           // 36,864: one 64-column chunk of a 256-row tile of xb
constexpr int W0IMG_BYTES = 128 * LDX * 2;           // 18,432: the matching chunk of the first encoder layer's weights
constexpr int OCH = 64;                              // ROI columns per output chunk
constexpr int OIMG_BYTES = OCH * LDP * 2;            // 17,408: [64][136] rows of decoder_mean_layer
constexpr int OBLOB_BYTES = 18432;                   // image + bias[64] + logvar_out[64] (fp32), padded to 18 pieces
constexpr int ACT_BYTES = ROWS * LDP * 2;            // 69,632: a saved activation in LDS layout
constexpr int PATCH_FLOATS = 16 * 16;                // one wave's 16 x 16 fp32 transposition patch (block-swizzled, see wgrad_adam)
constexpr int SPATCH_OFF = 8192;                     // patches inside S (bytes): above the 4 KiB the second x slot runs into S
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // 16-byte copies (builtin vector: address-space safe)

// explicit global-address-space pointer types (see asg())
#define GAS __attribute__((address_space(1)))
typedef GAS float* gf32;
typedef const GAS float* gcf32;
typedef GAS __bf16* gbf16;
typedef const GAS __bf16* gcbf16;

// Phase timers (NM_F_PROFILE): shader-clock cycles of workgroup (0,0), thread 0, accumulated per phase.
__device__ unsigned long long nm_prof_cycles[32];
// Per-wave interval timers (NM_F_TRACE): cycles between consecutive stamps of each wave of workgroup (0,0),
// attributed to the tag of the later stamp.
__device__ unsigned long long nm_trace_cycles[8][64];
// Start / end of every workgroup of the last NM_F_TRACE launch of nm_step_kernel on the constant-rate (100 MHz) counter
// that all XCDs share: [workgroup][0 = start, 1 = end], first 512 workgroups.
__device__ unsigned long long nm_wg_times[512][2];

struct Ctx {
  unsigned long long t_last;
  const nm_job_t* job;
  __bf16* P;
  __bf16* Q;
  float* stage;      // [STAGE_FLOATS] S
  float* vec;        // [2][256] fp32 vector slots (biases / logvar_out that travel with a weight image)
  int part, nparts;  // NM_F_SPLIT: this workgroup runs modality `part` of the job (of nparts); -1: the whole job
  int lstep;         // step index inside this launch (hand-off targets)
  float slope;       // negative slope of the activation (nm_job_t.act_slope)
  float* red;        // [64] reduction scratch
  float* colacc;     // [128] per-column accumulators
  float* rowacc;     // [256] per-row accumulators
  float* lse;        // [256] logsumexp over the rows of every expert's mu column (mvtCAE total correlation)
  float* bgrad;      // [128] bias gradients of the current weight-gradient pass (the ones column), see wgrad_adam
  unsigned* abort;   // [1] set by split_handoff on a time-out: the workgroup leaves the launch
  unsigned long long* tlast;   // [8] last stamp per wave (NM_F_TRACE)
  int wave_s;        // wave index of this wavefront inside the workgroup (wave-uniform, set once at kernel entry)
  int tid, lane, wave, wm, wn, g, c16;
  int row0;          // first table row of this tile
  int nrows;         // valid rows in this tile (<= 256)
  int flags;
  float inv_b;       // 1 / nrows
  // Adam scalars of the current step
  float step_size;   // lr / (1 - beta1^t)
  float inv_bc2_sqrt;
  GAS char* ws;      // workspace of this tile
};

__host__ __device__ inline int rup(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ inline int wpad(int n) { return rup(n + 1, 32); }   // width incl. the ones column

// ---- workspace layout (shared by host and device) ------------------------------------------
// One workspace per concurrently running tile of a job.  When the modalities of a model run as separate workgroups
// (NM_F_SPLIT) they share it: the expert statistics are double-buffered by step parity (a part may already be one
// step ahead), and every part has its own joint statistics / d z / decoder activations / z|c slot.
struct WsLayout {
  int64_t sync, mu_m, lv_m, mu_j, lv_j, es, dz, enc_act, dec_act, zc, total;
  int64_t lat, act;        // bytes of one [256][Zs] fp32 array / of one activation image
  int Zs;
};
constexpr int WS_SYNC_BYTES = 256;       // hand-off counters of the split mode: A at +0, B at +64, error flag at +128
constexpr int WS_SYNC_ERR_WORD = 32;
__host__ __device__ inline WsLayout ws_layout(int M, int L, int Z) {
  WsLayout w;
  w.Zs = rup(Z, 16);
  int64_t o = 0;
  w.sync = o; o += WS_SYNC_BYTES;
  const int64_t lat = (int64_t)ROWS * w.Zs * 4;
  w.lat = lat;
  w.mu_m = o; o += lat * M * 2;                // [step parity][expert]
  w.lv_m = o; o += lat * M * 2;
  w.mu_j = o; o += lat * M;                    // [part]
  w.lv_j = o; o += lat * M;
  w.es = o; o += lat * M;
  w.dz = o; o += lat * M;
  const int64_t act = (int64_t)ROWS * LDP * 2; // activation images (LDS layout, reloaded by LDS-DMA)
  w.act = act;
  w.enc_act = o; o += act * M * L;
  w.dec_act = o; o += act * M * L;             // [part][layer]
  w.zc = o; o += act * M;                      // [part]
  w.total = (o + 255) / 256 * 256;
  return w;
}

// ---- small helpers ---------------------------------------------------------------------------
enum { PH_ENC_L0 = 0, PH_ENC_REST, PH_HEADS, PH_LATENT, PH_DEC_ZC, PH_DEC_HID, PH_OUT_GEMM, PH_OUT_DLV, PH_OUT_DGRAD,
       PH_OUT_WGRAD, PH_NLL_RED, PH_DEC_FINISH, PH_DEC_LOAD, PH_DEC_DGRAD, PH_DEC_WGRAD, PH_DEC_DELTA, PH_ALPHA,
       PH_ENCB_PREP, PH_ENCB_HEADS_DGRAD, PH_ENCB_HEADS_WGRAD, PH_ENCB_LOAD, PH_ENCB_DGRAD, PH_ENCB_WGRAD,
       PH_ENCB_DELTA, PH_ENCB_L0_WGRAD, PH_X_LOADS, PH_X_MFMA, PH_X_EPI, PH_COUNT };
// Both timers branch on wave-uniform conditions only and let every lane of the wave do the same
// read-modify-write (same address, same value): a lane-divergent `if (lane == 0)` here would put dozens of
// EXEC-masked regions into the kernel, and register spills next to such regions are not safe with this
// compiler (tools/check_spill_exec.py).
__device__ __forceinline__ void tr(const Ctx& c, int tag) {
  if ((c.flags & 64) && blockIdx.x == 0 && blockIdx.y == 0) {
    unsigned long long t = clock64();
    const int w = c.wave_s;
    nm_trace_cycles[w][tag] += t - c.tlast[w];
    c.tlast[w] = t;
  }
}
__device__ __forceinline__ void prof(Ctx& c, int phase) {
  if ((c.flags & NM_F_PROFILE) && blockIdx.x == 0 && blockIdx.y == 0 && c.wave_s == 0) {
    unsigned long long t = clock64();
    nm_prof_cycles[phase] += t - c.t_last;
    c.t_last = t;
  }
}

// Re-derive the lane/wave indices per phase from opaque copies.  Without this the compiler hoists every per-lane
// LDS/global address of every phase out of the persistent step loop and then spills hundreds of them; re-deriving per
// phase keeps live ranges phase-local.  The wave index is a scalar kept in the context (read once from threadIdx.x at
// kernel entry), the lane index comes from mbcnt (EXEC is all ones wherever this is called): the workitem-id VGPR does
// not have to stay alive across the whole step loop -- it was the one value the allocator spilled when a phase gained a
// register.  Wave-level work splits compile to scalar branches.
__device__ __forceinline__ void relaunder(Ctx& c) {
  int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  int w = c.wave_s;
  asm volatile("" : "+s"(w));
  c.lane = l;
  c.tid = w * 64 + l;
  c.wave = w;
  c.wm = w / NWN;
  c.wn = w % NWN;
  c.g = c.lane >> 4;
  c.c16 = c.lane & 15;
}

// Pointers read out of the job descriptor are generic to the compiler; routing them through an
// explicit global-address-space pointer type lets it emit global_* (saddr + 32-bit offset), not flat_*.
template <class T>
__device__ __forceinline__ GAS T* asg(T* p) {
  return (GAS T*)p;
}

// Workgroup barrier that orders LDS traffic only: global loads, stores and LDS-DMA copies in flight stay in
// flight across it (a __syncthreads() drains vmcnt(0) whenever an LDS-DMA is pending).  Used wherever the barrier
// protects P / Q / S; hand-offs through GLOBAL memory between threads use handoff_barrier().
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ float lrelu(float v, bool nl, float slope) { return (nl && v < 0.f) ? v * slope : v; }

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

// exact a / b for 0 <= a < 2^22, 0 < b: float reciprocal + one-step fix (no integer division)
__device__ __forceinline__ int idiv(int a, int b, float rb) {
  int q = (int)((float)a * rb);
  q += ((q + 1) * b <= a) ? 1 : 0;
  q -= (q * b > a) ? 1 : 0;
  return q;
}

// D = A * B + C with A = 16 features x 32 k, B = 32 k x 16 rows: lane (c16, g) supplies
// A[feature c16][k 8g..8g+7] and B[k 8g..8g+7][row c16], and receives D[feature 4g+i][row c16].
__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// fragment of a row-major bf16 LDS tile: lane holds buf[row][k .. k+7]
__device__ __forceinline__ bf16x8 lds_frag(const __bf16* buf, int ld, int row, int k) {
  return *reinterpret_cast<const bf16x8*>(buf + row * ld + k);
}

// Transposed fragment: lane (c16, g) receives buf[r0 + 8g + j][c0 + c16], j = 0..7, i.e. the
// operand of a contraction over the ROW index of a row-major tile.  Two ds_read_b64_tr_b16:
// within each 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 and lane
// i receives column i of the four rows (cdna_hip_programming.md T10).  EXEC is all ones here.
__device__ __forceinline__ unsigned tr_addr(const __bf16* buf, int ld, int r0, int c0, int lane) {
  int i = lane & 15, g = lane >> 4;
  int q = i >> 2, p = i & 3;
  return lds_addr(buf + (r0 + 8 * g + q) * ld + c0 + 4 * p);
}
// Interleaved variant for products whose BOTH operands are transposed reads of the same rows (wgrad): the
// contraction index may be permuted freely as long as both sides use the same permutation, so group g takes
// rows r0 + 8g + {0, 2, 4, 6} with the first read and the odd rows (+1 row) with the second.  With the row
// pitches used here (68 or 36 dwords) the eight rows of a 32-lane half then start 8 banks apart (no two rows on
// one bank; the natural order puts two).  Measured effect on the step: within noise (-1.5 % on forward+backward).
__device__ __forceinline__ unsigned tr_addr_il(const __bf16* buf, int ld, int r0, int c0, int lane) {
  int i = lane & 15, g = lane >> 4;
  int q = i >> 2, p = i & 3;
  return lds_addr(buf + (r0 + 8 * g + 2 * q) * ld + c0 + 4 * p);
}
#define NM_TR_READ(dst, addr, OFF) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))

__device__ __forceinline__ bf16x8 join4(bf16x4 lo, bf16x4 hi) {
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// Scalar reference form of the transposed fragment (unit tests compare the two).
__device__ __forceinline__ bf16x8 lds_frag_tr_scalar(const __bf16* buf, int ld, int r0, int c0, int lane) {
  int c = c0 + (lane & 15), g = lane >> 4;
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = buf[(r0 + 8 * g + j) * ld + c];
  return r;
}

// Weight matrices live in the flat parameter buffer as 16 x 16 fp32 tiles, [ceil(N/16)][ceil(K/16)][16][16], zero
// padded: 1 KiB per tile.  The trunk touches them only in the Adam sweep (one lane-linear 16-byte access per lane and
// tile); the head kernels also read their GEMM fragments from them.
__host__ __device__ inline int ktiles(int K) { return (K + 15) >> 4; }
__host__ __device__ inline int64_t wt_elems(int N, int K) { return (int64_t)((N + 15) >> 4) * ktiles(K) * 256; }
__host__ __device__ inline int64_t wt_off(int n, int k, int KT) {
  return ((int64_t)((n >> 4) * KT + (k >> 4)) << 8) + ((n & 15) << 4) + (k & 15);
}
// Forward weight fragment: W[n][k0 .. k0+7] -> bf16x8 (k0 a multiple of 8), zero outside.
__device__ __forceinline__ bf16x8 w_frag(gcf32 W, int N, int K, int n, int k0) {
  const int KT = ktiles(K);
  const GAS f32x4* p = (const GAS f32x4*)(W + wt_off(min(n, N - 1), min(k0, KT * 16 - 8), KT));
  f32x4 a = p[0], b = p[1];
  const bool ok = (n < N) && (k0 < KT * 16);       // pad columns are zeros in memory
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = (__bf16)(ok ? a[j] : 0.f); r[4 + j] = (__bf16)(ok ? b[j] : 0.f); }
  return r;
}

// Dgrad weight fragment: W[n0 + j][k] for j = 0..7 (contraction over the OUTPUT index n; n0 a multiple of 8).
__device__ __forceinline__ bf16x8 w_frag_t(gcf32 W, int N, int K, int n0, int k) {
  const int KT = ktiles(K), NP = rup(N, 16);
  bf16x8 r;
  const int kc = min(k, KT * 16 - 1);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = W[wt_off(min(n0 + j, NP - 1), kc, KT)];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)((n0 + j < N && k < K) ? v[j] : 0.f);
  return r;
}

// Block-wide sum; every thread gets the result.  Fixed summation order (bitwise reproducible).
__device__ __forceinline__ float block_sum(const Ctx& c, float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  lds_barrier();
  if (c.lane == 0) c.red[c.wave] = v;
  lds_barrier();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < NWAVES; ++w) s += c.red[w];
  return s;
}

// Counter-based standard normal for the in-kernel draw (eps == NULL): splitmix64 + Box-Muller.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float randn_ctr(uint64_t seed, uint32_t step, uint32_t row, uint32_t z) {
  uint64_t h = splitmix64(seed ^ ((uint64_t)step << 32) ^ ((uint64_t)row << 8) ^ z);
  uint64_t h2 = splitmix64(h);
  float u1 = ((uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777217.0f);   // (0, 1]
  float u2 = (uint32_t)(h2 >> 40) * (1.0f / 16777216.0f);           // [0, 1)
  // hardware transcendentals (v_log_f32 = log2, v_cos_f32 takes revolutions): the draw is a random number, not
  // a parity quantity -- parity runs inject eps
  return __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
}
// two draws from one hash (both branches of the Box-Muller pair): the 16-byte latent path, keyed by the column pair
__device__ __forceinline__ void randn2_ctr(uint64_t seed, uint32_t step, uint32_t row, uint32_t zpair, float& n0, float& n1) {
  const uint64_t h = splitmix64(seed ^ 0x2D0B1E5Dull ^ ((uint64_t)step << 32) ^ ((uint64_t)row << 8) ^ zpair);
  const float u1 = ((uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777217.0f);   // (0, 1]
  const float u2 = (uint32_t)(h & 0xFFFFFFu) * (1.0f / 16777216.0f);      // [0, 1)
  const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  n0 = rad * __builtin_amdgcn_cosf(u2);
  n1 = rad * __builtin_amdgcn_sinf(u2);
}

// ---- Adam (torch.optim.Adam as configured at cVAE.py:1111-1116) -------------------------------
struct AdamK { float b1, b2, eps, step_size, inv_bc2_sqrt; };
__device__ __forceinline__ AdamK adam_consts(const Ctx& c) {
  const nm_job_t* J = c.job;
  return AdamK{J->beta1, J->beta2, J->adam_eps, c.step_size, c.inv_bc2_sqrt};
}
__device__ __forceinline__ void adam1(const AdamK& a, float g, float& p, float& m, float& v) {
  m = m + (g - m) * (1.0f - a.b1);                    // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.b2 + (1.0f - a.b2) * g * g;               // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
  // v_sqrt_f32 / v_rcp_f32 (1 ulp): the update is ~lr * O(1), so the difference to the correctly rounded
  // sequence is ~1e-11 absolute, far below fp32 resolution of the parameters
  float denom = __builtin_amdgcn_sqrtf(v) * a.inv_bc2_sqrt + a.eps;
  p = p - a.step_size * (m * __builtin_amdgcn_rcpf(denom));
}

// scalar gradient sink (a handful of elements per step: alpha, d logvar_out)
// (sh: fp32 copy of the element inside a shadow image's vector piece, or nullptr)
__device__ __forceinline__ void apply_grad(const Ctx& c, int64_t idx, float g, GAS float* sh = nullptr) {
  const nm_job_t* J = c.job;
  if (c.flags & NM_F_GRADS) asg(J->grads)[idx] = g;
  if (c.flags & NM_F_ADAM) {
    gf32 P_ = asg(J->params); gf32 M_ = asg(J->adam_m); gf32 V_ = asg(J->adam_v);
    float p = P_[idx], m = M_[idx], v = V_[idx];
    adam1(adam_consts(c), g, p, m, v);
    P_[idx] = p; M_[idx] = m; V_[idx] = v;
    if (sh) *sh = p;
  }
}

// ---- LDS-DMA ---------------------------------------------------------------------------------------
// global_load_lds_dwordx4: every lane moves 16 bytes from its own global address to LDS at (wave-uniform base +
// lane * 16) -- one wave instruction lands 1 KiB of contiguous LDS, no VGPR in between, tracked by vmcnt like any
// other vector-memory operation.  Everything the forward and dgrad GEMMs stage (weight images, x chunks, saved
// activations) is therefore kept in global memory in exactly the LDS layout, in 1-KiB pieces, and is requested
// one phase ahead of its use.
typedef __attribute__((address_space(3))) void* lds_vp;
// POL = 2: streaming (nt) -- for bytes this workgroup alone reads, once per step (weight images, saved activations);
// the ROI tables, which the models of a fold share through L2 / Infinity Cache, keep the default policy.
template <int POL = 0>
__device__ __forceinline__ void dma16(const GAS char* src_lane, char* dst_wave) {
  __builtin_amdgcn_global_load_lds((const GAS void*)src_lane, (lds_vp)dst_wave, 16, 0, POL);
}
// pieces [0, npieces) of 1 KiB, contiguous on both sides; wave w takes w, w + 8, ...  Returns the number of
// instructions THIS wave issued (wave-uniform): the count a later s_waitcnt vmcnt(N) needs.
template <int POL = 0>
__device__ __forceinline__ int dma_lin(const Ctx& c, const GAS char* src, char* dst, int npieces) {
  int n = 0;
  for (int p = c.wave; p < npieces; p += NWAVES) {
    dma16<POL>(src + (p << 10) + (c.lane << 4), dst + (p << 10));
    ++n;
  }
  return n;
}
// s_waitcnt vmcnt(n): returns once at most n of this wave's vector-memory operations are outstanding, i.e. once
// everything issued BEFORE the youngest n has completed.  n is wave-uniform; a smaller n than necessary only waits
// longer.
__device__ __forceinline__ void wait_vm(int n) {
  if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if (n <= 9) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n <= 11) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (n <= 13) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n <= 15) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (n <= 19) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
}
// Barrier for hand-offs that go through GLOBAL memory between threads of the workgroup (latent statistics, d z):
// every wave first drains its own stores, then the workgroup meets.  (One CU's vector L1 is in order for its own
// wavefronts; the explicit drain makes the hand-off independent of that.)
__device__ __forceinline__ void handoff_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// Hand-off between the workgroups that share a model (NM_F_SPLIT): every part arrives at a monotonic counter once
// its stores are globally visible, and leaves once all parts of the job have arrived.  Form: every wave drains its
// own stores, the workgroup meets, one lane releases at agent scope (write-back of this XCD's L2), adds its arrival,
// polls relaxed, acquires at agent scope (this CU's L1 is invalidated) -- MI355X_MICROARCH.md, "Valid forms".
// The spin is bounded (~1 s): on a time-out the job's error word (workspace + 128, sticky: nm_split_errors reads it,
// only the host clears it) is set and the workgroup LEAVES the launch -- the other parts' statistics are stale, so
// nothing computed from them may reach the parameters.  Returns false in that case (for every thread).
__device__ __forceinline__ bool split_handoff(const Ctx& c, GAS unsigned* cnt, GAS unsigned* err, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (c.tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add((unsigned*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    bool ok = true;
    while (__hip_atomic_load((unsigned*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      // (another part that has already given up also ends this wait: no part is left spinning for its full bound)
      if (++spins > (1 << 22) || __hip_atomic_load((unsigned*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store((unsigned*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *c.abort = ok ? 0u : 1u;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  return *c.abort == 0u;
}

// Hidden / heads / decoder weight blobs in memory: the matrix COMPACT, [rows][kp] bf16 row-major with kp = blob_kp(K) (the
// ones column and the 16-column tiles of the Adam stores included), then -- 1-KiB aligned -- the fp32 vector piece.
// global_load_lds takes a per-lane source address, so the copy into LDS still lands as the [128][136] image the GEMMs
// read: segments outside the matrix (pad rows / pad columns) come from one line of zeros at the start of job.wsh.
// (round 2 stored the padded image itself: 34.8 KB read twice per step for a 110 x 111 layer, 7-25 KB now)
__host__ __device__ inline int blob_kp(int K) { const int a = rup(K + 1, 8), b = rup(K, 16); return a > b ? a : b; }
// (rows rounded to 16 in memory: the Adam units store whole 16-row tiles, pad rows stay zero)
__host__ __device__ inline int cimg_bytes(int rows, int K) { return rup(rup(rows, 16) * blob_kp(K) * 2, 1024); }
__host__ __device__ inline int cblob_bytes(int rows, int K) { return cimg_bytes(rows, K) + VEC_BYTES; }
constexpr int WSH_ZERO_BYTES = 1024;     // job.wsh starts with zeros (never written): the source of every pad segment
__device__ __forceinline__ int dma_img(const Ctx& c, const GAS char* src, char* dst, int rows, int kp, const GAS char* zero) {
  int n = 0;
  const int segs = kp >> 3;
  for (int p = c.wave; p < (IMG_BYTES >> 10); p += NWAVES) {
    const int q = (p << 6) + c.lane;                       // 16-byte segment of the LDS image
    const int r = idiv(q, LDP / 8, 8.0f / (float)LDP), sg = q - (LDP / 8) * r;
    const GAS char* a = (r < rows && sg < segs) ? src + (((int64_t)r * kp + sg * 8) << 1) : zero;
    dma16<0>(a, dst + (p << 10));
    ++n;
  }
  return n;
}
// What a GEMM phase requests for the phase after it: `np` 1-KiB pieces src -> dst (linear copy; rows > 0: a compact weight
// matrix [rows][kp] into a [128][136] image, dma_img) and, optionally, one more piece (the fp32 vectors that travel with a
// weight image) vsrc -> vdst.
struct Next {
  const GAS char* src; char* dst; int np;
  const GAS char* vsrc; char* vdst;
  int rows, kp;
};
__device__ __forceinline__ Next no_next() { return Next{nullptr, nullptr, 0, nullptr, nullptr, 0, 0}; }
__device__ __forceinline__ int issue_next(const Ctx& c, const Next& nx) {
  int n = 0;
  if (nx.src) n = nx.rows > 0 ? dma_img(c, nx.src, nx.dst, nx.rows, nx.kp, (const GAS char*)c.job->wsh)
                              : dma_lin(c, nx.src, nx.dst, nx.np);
  if (nx.vsrc && c.wave == 2) { dma16<0>(nx.vsrc + (c.lane << 4), nx.vdst); ++n; }
  return n;
}
// the weight blob of a layer with `rows` output rows and K inputs (+ its vector piece) into half `half` of Q / vector slot `half`
__device__ __forceinline__ Next blob_to_half(const Ctx& c, const GAS char* blob, int half, int rows, int K) {
  return Next{blob, reinterpret_cast<char*>(c.Q) + half * IMG_BYTES, IMG_BYTES >> 10, blob + cimg_bytes(rows, K),
              reinterpret_cast<char*>(c.vec) + half * VEC_BYTES, rows, blob_kp(K)};
}

// ---- cooperative copies ----------------------------------------------------------------------
// Saved activations in the workspace: COMPACT, [256][nsegs * 8] bf16 with nsegs = act_segs(width) 16-byte segments per row
// (the real columns + the ones column; 17 = the whole LDS row), reloaded into the [256][136] LDS layout by per-lane-address
// LDS-DMA with the pad segments taken from the zero line of job.wsh (dma_act).  z | c | 1 is 5 segments of 17, a 110-wide
// layer 14.
__host__ __device__ inline int act_segs(int N) { const int s = rup(N + 1, 8) / 8; return s < LDP / 8 ? s : LDP / 8; }
// store count a wave can rely on (lower bound: the iterations every thread takes part in)
__host__ __device__ inline int act_stores(int nsegs) { return (ROWS * nsegs) / WG; }
__device__ __forceinline__ void store_act_img(const Ctx& c, gbf16 dst, const __bf16* src, int nsegs) {
  const float rs = 1.0f / (float)nsegs;
#pragma unroll 2
  for (int p = c.tid; p < ROWS * nsegs; p += WG) {
    const int row = idiv(p, nsegs, rs), seg = p - row * nsegs;
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + row * LDP + seg * 8);
    *(GAS u32x4*)(dst + ((int64_t)row * nsegs + seg) * 8) = v;
  }
}
// rows [r0, r0 + nr) of a compact activation -> LDS at `dst` (= the LDS address of row r0); nr a multiple of 64
__device__ __forceinline__ int dma_act(const Ctx& c, const GAS char* src, char* dst, int r0, int nr, int nsegs) {
  const GAS char* zero = (const GAS char*)c.job->wsh;
  int n = 0;
  for (int p = c.wave; p < (nr * (LDP / 8)) >> 6; p += NWAVES) {
    const int q = (p << 6) + c.lane;
    const int r = idiv(q, LDP / 8, 8.0f / (float)LDP), sg = q - (LDP / 8) * r;
    const GAS char* a = sg < nsegs ? src + (((int64_t)(r0 + r) * nsegs + sg) << 4) : zero;
    dma16<0>(a, dst + (p << 10));
    ++n;
  }
  return n;
}
// legacy [256][PW] workspace tiles (head kernels, fusion-backward hand-off)
__device__ __forceinline__ void load_act(const Ctx& c, __bf16* dst, gcbf16 src, int width) {
  const int segs = width >> 3;                   // 16-byte pieces per row (width is a multiple of 32)
  const float rs = 1.0f / (float)segs;
#pragma unroll 4
  for (int p = c.tid; p < ROWS * segs; p += WG) {
    int row = idiv(p, segs, rs), seg = p - row * segs;
    u32x4 v = *(const GAS u32x4*)(src + row * PW + seg * 8);
    *reinterpret_cast<u32x4*>(dst + row * LDP + seg * 8) = v;
  }
}
__device__ __forceinline__ void store_act(const Ctx& c, gbf16 dst, const __bf16* src, int width) {
  const int segs = width >> 3;
  const float rs = 1.0f / (float)segs;
#pragma unroll 4
  for (int p = c.tid; p < ROWS * segs; p += WG) {
    int row = idiv(p, segs, rs), seg = p - row * segs;
    u32x4 v = *reinterpret_cast<const u32x4*>(src + row * LDP + seg * 8);
    *(GAS u32x4*)(dst + row * PW + seg * 8) = v;
  }
}

// ---- register-staged fp32 weight blocks (head kernels only: the trunk reads bf16 shadow images) -----------------
// A [NR][128] block of a tiled fp32 weight matrix (rows row0.., columns col0..): 32 pieces of 4 floats per row;
// bf16 [NR][ld] in LDS, zeros past the matrix.
template <int NR>
struct WBlk { f32x4 v[(NR * 128 / 4) / WG]; };
template <int NR>
__device__ __forceinline__ void wblk_load(const Ctx& c, WBlk<NR>& s, gcf32 W, int N, int K, int row0, int col0) {
  const int KT = ktiles(K), NP = rup(N, 16);
#pragma unroll
  for (int j = 0; j < (NR * 128 / 4) / WG; ++j) {
    const int p = c.tid + j * WG, row = row0 + (p >> 5), col = col0 + (p & 31) * 4;
    s.v[j] = *(const GAS f32x4*)(W + wt_off(min(row, NP - 1), min(col, KT * 16 - 4), KT));
  }
}
template <int NR>
__device__ __forceinline__ void wblk_store(const Ctx& c, const WBlk<NR>& s, __bf16* dst, int ld, int N, int K, int row0,
                                           int col0) {
  const int KT = ktiles(K);
#pragma unroll
  for (int j = 0; j < (NR * 128 / 4) / WG; ++j) {
    const int p = c.tid + j * WG, lr = p >> 5, lc = (p & 31) * 4;
    const bool ok = row0 + lr < N && col0 + lc < KT * 16;       // pad rows / columns inside a tile are zeros in memory
    bf16x4 pk;
#pragma unroll
    for (int i = 0; i < 4; ++i) pk[i] = (__bf16)(ok ? s.v[j][i] : 0.f);
    *reinterpret_cast<bf16x4*>(dst + lr * ld + lc) = pk;
  }
}

// [z | c | 1 | 0] rows of the decoder input (cVAE.py:199) into an LDS buffer: the covariate / ones columns from
// the table's cz block (16-byte pieces, scattered to the unaligned destination with 2-byte LDS stores), zero pad,
// then the z columns from the latent workspace.
// (DMVAE family: the last S of the Z latent columns are the modality's private latent = columns [0, S) of its own
// encoder's mu, `priv`, taken as they are)
// (zsrc != nullptr: the z columns come from LDS, bf16 [256][32] -- fwd_heads' in-register draw)
__device__ __forceinline__ void build_zc(const Ctx& c, __bf16* dst, const nm_modality_t& md, gcf32 mu_j, gcf32 es, int Z,
                                         int C, int Zs, int S, gcf32 priv, const __bf16* zsrc = nullptr) {
  const int wz = wpad(Z + C);
  const float rz = 1.0f / (float)Z;
  const GAS uint16_t* cz = asg(md.cz);
  {
    const int npc = (C + 1 + 7) >> 3;
    const float rnp = 1.0f / (float)npc;
    for (int p = c.tid; p < ROWS * npc; p += WG) {
      const int r = idiv(p, npc, rnp), col0 = 8 * (p - r * npc);
      const u32x4 v = *(const GAS u32x4*)(cz + (int64_t)(c.row0 + r) * md.Cz + col0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = col0 + i;
        const uint16_t h = (uint16_t)(v[i >> 1] >> (16 * (i & 1)));
        if (j <= C) reinterpret_cast<uint16_t*>(dst)[r * LDP + Z + j] = h;
      }
    }
  }
  {                                                 // zero pad columns (Z + C, wz)
    const int nz = wz - (Z + C + 1);
    if (nz > 0) {
      const float rnz = 1.0f / (float)nz;
      for (int e = c.tid; e < ROWS * nz; e += WG) {
        const int r = idiv(e, nz, rnz);
        dst[r * LDP + Z + C + 1 + (e - r * nz)] = (__bf16)0.0f;
      }
    }
  }
  const int Zc = Z - S;                              // shared columns first, then the private ones
  if (zsrc) {
#pragma unroll 4
    for (int e = c.tid; e < ROWS * Z; e += WG) {
      int r = idiv(e, Z, rz), k = e - r * Z;
      dst[r * LDP + k] = zsrc[r * 32 + k];
    }
    return;
  }
#pragma unroll 4
  for (int e = c.tid; e < ROWS * Z; e += WG) {
    int r = idiv(e, Z, rz), k = e - r * Z;
    const int ks = min(k, max(Zc - 1, 0)), kp = min(max(k - Zc, 0), max(S - 1, 0));
    const float shared = mu_j[r * Zs + ks] + es[r * Zs + ks];
    const float v = (k < Zc) ? shared : (S > 0 ? priv[r * Zs + kp] : 0.f);
    dst[r * LDP + k] = (__bf16)v;
  }
}

// ---- accumulator tile bookkeeping --------------------------------------------------------------
// acc[t][rt]: feature tile ft = wn + 4 t, row tile rt; lane holds features ft*16 + 4g + i (i = 0..3)
// of row wm*128 + rt*16 + c16.
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[2][RT]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void bias_acc(const Ctx& c, f32x4 (&acc)[2][RT], gcf32 b, int N, int f_base) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int f0 = f_base + (c.wn + 4 * t) * 16 + 4 * c.g;
    f32x4 bv;
#pragma unroll
    for (int i = 0; i < 4; ++i) { float x = b[min(f0 + i, N - 1)]; bv[i] = (f0 + i < N) ? x : 0.f; }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[t][rt] = bv;
  }
}
// activation epilogue: P[r][f] = act(acc + bias[f]) for f < N, 1 at f == N (ones column), 0 beyond.  bias: LDS
// vector (zero padded to 128) or nullptr when the accumulators already carry it.
__device__ __forceinline__ void act_to_P(const Ctx& c, const f32x4 (&acc)[2][RT], const float* bias, int N, int ntn,
                                         bool act) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int ft = c.wn + 4 * t;
    if (ft >= ntn) continue;
    int f0 = ft * 16 + 4 * c.g;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + f0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      int r = c.wm * WROWS + rt * 16 + c.c16;
      bf16x4 pk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[t][rt][i] + bv[i];
        v = (f0 + i < N) ? lrelu(v, act, c.slope) : (f0 + i == N ? 1.0f : 0.0f);
        pk[i] = (__bf16)v;
      }
      *reinterpret_cast<bf16x4*>(c.P + r * LDP + f0) = pk;
    }
  }
}

// ---- GEMM phase: forward layer, P -> P in place ---------------------------------------------
// out[r][n] = act(sum_k P[r][k] W[n][k] + b[n]), n < N; column N := 1 (ones column feeding the next layer's
// bias gradient), columns (N, wpad(N)) := 0.  The layer's weight image [128][LDP] and bias vector were requested
// by the PREVIOUS phase into half `half` of Q / vector slot `half`; this phase first requests `nx` (the next
// phase's image), then waits for its own.  Optionally saved to `save` (activation image) for the backward pass.
// `younger` = vector-memory operations this wave issued AFTER the request of this layer's image and before this call
// (the previous phase's activation save: act_stores(..) of them at least): they may stay in flight.
__device__ __forceinline__ void fwd_layer(const Ctx& cc, int half, const Next& nx, int N, int K, bool act, gbf16 save,
                                          int younger) {
  Ctx c = cc;
  relaunder(c);
  const int ksteps = wpad(K) / 32;       // <= 4
  const int ntn = wpad(N) / 16;
  const __bf16* Wt = c.Q + half * (IMG_ROWS * LDP);
  const int n_next = issue_next(c, nx);
  wait_vm(n_next + younger);
  lds_barrier();
  f32x4 acc[2][RT];
  zero_acc(acc);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (ks < ksteps) {
      bf16x8 wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) wf[t] = lds_frag(Wt, LDP, (c.wn + 4 * t) * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        bf16x8 a = lds_frag(c.P, LDP, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][rt] = mfma(wf[t], a, acc[t][rt]);
      }
    }
  }
  lds_barrier();                       // every wave has finished reading P
  act_to_P(c, acc, c.vec + half * (VEC_BYTES / 4), N, ntn, act);
  lds_barrier();
  if (save) store_act_img(c, save, c.P, act_segs(N));
}

// ---- GEMM phase: first encoder layer ----------------------------------------------------------
// x | c | 1 and the layer's weights stream through LDS in 64-column chunks, both as LDS-DMA copies of ready-made
// images (xb chunk [256][LDX], weight chunk [128][LDX]); two stages (one in P, one in Q: both are dead here), so
// chunk i + 1 is in flight while chunk i is multiplied.  The last chunk sits in P; `nx` (the next layer's image,
// into Q) is requested as soon as the Q stage is drained.
// (xsrc: the tile's chunk images [nch][256][LDX]; wsrc: the layer's weights -- compact = false: chunk images
//  [nch][128][LDX] followed by the bias piece (the regression head's first layer); compact = true: the matrix itself,
//  [N rounded to 16][Kx] bf16 row-major, then the bias piece (l0_img_bytes): the trunk's encoders, each chunk gathered into
//  its [128][LDX] stage by per-lane-address copies with the pad rows / the pad segment taken from the zero line.)
__host__ __device__ inline int l0_img_bytes(int N, int Kx) { return rup(rup(N, 16) * Kx * 2, 1024); }
__device__ __forceinline__ int dma_w0chunk(const Ctx& c, const GAS char* w, char* dst, int N, int Kx, int i) {
  const GAS char* zero = (const GAS char*)c.job->wsh;
  int n = 0;
  for (int p = c.wave; p < (W0IMG_BYTES >> 10); p += NWAVES) {
    const int q = (p << 6) + c.lane;
    const int r = idiv(q, LDX / 8, 8.0f / (float)LDX), sg = q - (LDX / 8) * r;
    const int k = i * XCH + sg * 8;
    const GAS char* a = (r < N && sg < XCH / 8 && k < Kx) ? w + (((int64_t)r * Kx + k) << 1) : zero;
    dma16<0>(a, dst + (p << 10));
    ++n;
  }
  return n;
}
__device__ __forceinline__ void fwd_first_layer(const Ctx& cc, const GAS char* xsrc, int Kx, const GAS char* wsrc, const Next& nx,
                                                int N, bool act, gbf16 save, bool compact = false) {
  Ctx c = cc;
  relaunder(c);
  const int nch = (Kx + XCH - 1) / XCH;
  const int ntn = wpad(N) / 16;
  float* bias = c.vec + (VEC_BYTES / 4);            // slot 1 (slot 0 receives nx's vectors)
  auto stage = [&](int i) { return reinterpret_cast<char*>(((nch - 1 - i) & 1) ? c.Q : c.P); };
  auto issue_chunk = [&](int i) {
    char* st = stage(i);
    int n = dma_lin<0>(c, xsrc + (int64_t)i * XIMG_BYTES, st, XIMG_BYTES >> 10);
    n += compact ? dma_w0chunk(c, wsrc, st + XIMG_BYTES, N, Kx, i)
                 : dma_lin(c, wsrc + (int64_t)i * W0IMG_BYTES, st + XIMG_BYTES, W0IMG_BYTES >> 10);
    return n;
  };
  int n_nxt = 0, n_blob = 0;
  issue_chunk(0);
  if (c.wave == 3)
    dma16<0>(wsrc + (compact ? (int64_t)l0_img_bytes(N, Kx) : (int64_t)nch * W0IMG_BYTES) + (c.lane << 4), reinterpret_cast<char*>(bias));
  if (nch > 1) n_nxt = issue_chunk(1);
  else n_blob = issue_next(c, nx);
  f32x4 acc[2][RT];
  zero_acc(acc);
  for (int i = 0; i < nch; ++i) {
    wait_vm(n_nxt + n_blob);
    lds_barrier();                       // chunk i has landed for every wave
    const __bf16* X = reinterpret_cast<const __bf16*>(stage(i));
    const __bf16* Wq = X + ROWS * LDX;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (i * XCH + ks * 32 < Kx) {
        bf16x8 wf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) wf[t] = lds_frag(Wq, LDX, (c.wn + 4 * t) * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          bf16x8 a = lds_frag(X, LDX, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t][rt] = mfma(wf[t], a, acc[t][rt]);
        }
      }
    }
    lds_barrier();                       // stage(i) is drained
    int n_new = 0;
    if (i + 2 < nch) n_new = issue_chunk(i + 2);
    else if (i + 2 == nch) n_blob = issue_next(c, nx);        // stage(i) was the Q stage: Q is free from here on
    n_nxt = n_new;
  }
  act_to_P(c, acc, bias, N, ntn, act);
  lds_barrier();
  if (save) store_act_img(c, save, c.P, act_segs(N));
  tr(c, 0);
}

// ---- GEMM phase: encoder heads, P (= last hidden) -> fp32 mu / logvar in the workspace --------
// Heads image: rows [0, Z) = enc_mean_layer, rows [Zs, Zs + Z) = enc_logvar_layer, zeros elsewhere; the vector
// piece holds the biases at the same row indices.
// `zdst` != nullptr (forward-only launches of a single-expert model, the deviation pass): the latent draw is made right
// here from the accumulators -- z = mu + eps exp(logvar / 2) with the exp -> log round trip of cVAE.py:1175-1178 -- and
// goes to LDS as bf16 [256][32] (the decoder's input operand); nothing is stored to the workspace, the fusion phase and
// its two hand-offs through global memory are skipped.  kl_out receives this thread's share of the KL sum.
__device__ __forceinline__ void fwd_heads(const Ctx& cc, int half, const Next& nx, int Z, int K, gf32 mu_out, gf32 lv_out,
                                          int Zs, int younger, __bf16* zdst = nullptr, int step = 0, bool pair_draw = false,
                                          float* kl_out = nullptr) {
  Ctx c = cc;
  relaunder(c);
  const int ksteps = wpad(K) / 32;
  const int nzt = Zs / 16;
  const __bf16* Wt = c.Q + half * (IMG_ROWS * LDP);
  const float* bias = c.vec + half * (VEC_BYTES / 4);
  const int n_next = issue_next(c, nx);
  wait_vm(n_next + younger);
  lds_barrier();
  // unit = (feature tile, row tile of the wave's row half): the 4 waves of a row half share them round-robin,
  // so all 8 waves work even when the latent fits one feature tile
  for (int u = c.wn; u < nzt * RT; u += NWN) {
    const int ft = u / RT, rt = u - ft * RT;
    const int f0 = ft * 16 + 4 * c.g;
    f32x4 am = {0.f, 0.f, 0.f, 0.f}, al = am;
    for (int ks = 0; ks < ksteps; ++ks) {
      const bf16x8 fm = lds_frag(Wt, LDP, ft * 16 + c.c16, ks * 32 + 8 * c.g);
      const bf16x8 fl = lds_frag(Wt, LDP, Zs + ft * 16 + c.c16, ks * 32 + 8 * c.g);
      const bf16x8 a = lds_frag(c.P, LDP, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
      am = mfma(fm, a, am);
      al = mfma(fl, a, al);
    }
    const int r = c.wm * WROWS + rt * 16 + c.c16;
    am += *reinterpret_cast<const f32x4*>(bias + f0);            // features >= Z: zero weight rows, zero bias
    al += *reinterpret_cast<const f32x4*>(bias + Zs + f0);
    if (zdst) {
      const nm_job_t* J = c.job;
      float ep[4];
      if (J->eps) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ep[i] = asg(J->eps)[((int64_t)(step % J->eps_cap) * ROWS + r) * Z + min(f0 + i, Z - 1)];
      } else if (pair_draw) {                                    // the generators of the fusion phase, same keys
        randn2_ctr(J->seed, (uint32_t)step, (uint32_t)(c.row0 + r), (uint32_t)(f0 >> 1), ep[0], ep[1]);
        randn2_ctr(J->seed, (uint32_t)step, (uint32_t)(c.row0 + r), (uint32_t)(f0 >> 1) + 1u, ep[2], ep[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ep[i] = randn_ctr(J->seed, (uint32_t)step, (uint32_t)(c.row0 + r), (uint32_t)(f0 + i));
      }
      bf16x4 zk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lv2 = logf(expf(al[i]));                     // variance and back, as the reference does
        const float zz = am[i] + ep[i] * expf(0.5f * lv2);
        zk[i] = (__bf16)((f0 + i < Z) ? zz : 0.f);
        if (kl_out && r < c.nrows && f0 + i < Z) *kl_out += -0.5f * (1.0f + lv2 - am[i] * am[i] - expf(lv2));
      }
      *reinterpret_cast<bf16x4*>(zdst + r * 32 + f0) = zk;
    } else {
      *(GAS f32x4*)(mu_out + r * Zs + f0) = am;
      *(GAS f32x4*)(lv_out + r * Zs + f0) = al;
    }
  }
  lds_barrier();                          // P and the image are drained (the latent hand-off has its own barrier)
  tr(c, 2);
}

// ---- dgrad: acc[k][r] += sum_n A[r][n] W[n][k]  (contraction over the columns of A) ------------
// k tiles {wn, wn+4} of wpad(K); nsteps = 32-wide steps over A's columns; n_base = index of A's
// column 0 in W's row space.  Weights straight from the (tiled) fp32 master: head kernels only.
__device__ __forceinline__ void dgrad_acc(const Ctx& cc, f32x4 (&acc)[2][RT], const __bf16* A, gcf32 W, int N, int K,
                                          int nsteps, int n_base) {
  Ctx c = cc;
  relaunder(c);
  for (int s0 = 0; s0 < nsteps; s0 += 2) {
    bf16x8 wf[2][2];
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        wf[ss][t] = w_frag_t(W, N, K, n_base + min(s0 + ss, nsteps - 1) * 32 + 8 * c.g, (c.wn + 4 * t) * 16 + c.c16);
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      if (s0 + ss < nsteps) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          bf16x8 a = lds_frag(A, LDP, c.wm * WROWS + rt * 16 + c.c16, (s0 + ss) * 32 + 8 * c.g);
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t][rt] = mfma(wf[ss][t], a, acc[t][rt]);
        }
      }
    }
  }
}

// dgrad with the weights taken from an LDS tile T[n][k] (bf16, row pitch ld) through the transposing read:
// lane (c16, g) gets T[s*32 + 8g + j][ktile*16 + c16], j = 0..7.  A = delta rows in LDS (row pitch lda), column
// n_col0 + s*32 onwards.
__device__ __forceinline__ void dgrad_tile(const Ctx& c, f32x4 (&acc)[2][RT], const __bf16* A, int lda, int n_col0,
                                           const __bf16* T, int ld, int nsteps) {
  for (int s = 0; s < nsteps; ++s) {
    bf16x4 l0, h0, l1, h1;
    const unsigned a0 = tr_addr(T, ld, s * 32, (c.wn + 0) * 16, c.lane), a1 = tr_addr(T, ld, s * 32, (c.wn + 4) * 16, c.lane);
    const unsigned r4 = 4u * ld * 2u;
    NM_TR_READ(l0, a0, 0); NM_TR_READ(h0, a0 + r4, 0);
    NM_TR_READ(l1, a1, 0); NM_TR_READ(h1, a1 + r4, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1));
    const bf16x8 wf0 = join4(l0, h0), wf1 = join4(l1, h1);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      bf16x8 a = lds_frag(A, lda, c.wm * WROWS + rt * 16 + c.c16, n_col0 + s * 32 + 8 * c.g);
      acc[0][rt] = mfma(wf0, a, acc[0][rt]);
      acc[1][rt] = mfma(wf1, a, acc[1][rt]);
    }
  }
}
// Hidden-layer backward, first half: P = delta of the layer's output [256][N].  The layer's weight image goes to
// the lower half of Q and, meanwhile, rows 128..255 of the layer's saved INPUT activation to the upper half; after
// the dgrad the lower 128 rows follow (still in flight on return: the caller's wgrad_adam(..., pending = 0) waits), so that
// Q = input activation for wgrad and the ReLU mask.
__device__ __forceinline__ void dgrad_hidden(const Ctx& cc, f32x4 (&acc)[2][RT], const GAS char* w_img, const GAS char* act_img,
                                             int N, int K) {
  Ctx c = cc;
  relaunder(c);
  char* Qb = reinterpret_cast<char*>(c.Q);
  dma_img(c, w_img, Qb, N, blob_kp(K), (const GAS char*)c.job->wsh);
  const int n_hi = dma_act(c, act_img, Qb + IMG_BYTES, ROWS / 2, ROWS / 2, act_segs(K));
  wait_vm(n_hi);
  lds_barrier();
  dgrad_tile(c, acc, c.P, LDP, 0, c.Q, LDP, wpad(N) / 32);
  lds_barrier();                              // weight image fully read
  dma_act(c, act_img, Qb, 0, ROWS / 2, act_segs(K));   // waited for by the weight-gradient pass that follows (wgrad_adam, pending = 0)
}

// P[r][k] = acc[k][r] * leaky_relu'(src[r][k]) for k < K, 0 for the ones/pad columns.
__device__ __forceinline__ void finish_delta(const Ctx& cc, const f32x4 (&acc)[2][RT], const __bf16* src, int K,
                                             bool act) {
  Ctx c = cc;
  relaunder(c);
  const int ntk = wpad(K) / 16;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int kt = c.wn + 4 * t;
    if (kt >= ntk) continue;
    int k0 = kt * 16 + 4 * c.g;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      int r = c.wm * WROWS + rt * 16 + c.c16;
      bf16x4 a = *reinterpret_cast<const bf16x4*>(src + r * LDP + k0);
      bf16x4 pk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float d = acc[t][rt][i];
        if (act && !((float)a[i] > 0.f)) d *= c.slope;
        if (k0 + i >= K) d = 0.f;
        pk[i] = (__bf16)d;
      }
      *reinterpret_cast<bf16x4*>(c.P + r * LDP + k0) = pk;
    }
  }
}

// ---- wgrad + Adam ------------------------------------------------------------------------------
// dW[n][k] = sum_r A[r][a_col0 + n] * B[r][kk], n in [0,N), B columns kk in [0, ncols) map to the weight column
// k = k_base + kk (k_base a multiple of 16); k < K is W[n][k], k == K the bias b[n] (ones column), beyond: nothing.
// A unit = one 16-row n tile x two adjacent 16-column k tiles (the n-side fragment is shared).  The waves take
// units round-robin and run them INDEPENDENTLY -- no workgroup barrier, no shared slab: per unit a wave
//   (i)   requests p / m / v of its NEXT unit: the master keeps every 16 x 16 tile as 1 KiB of contiguous memory,
//         so each request is one lane-linear 16-byte load per lane (full lines, streaming);
//   (ii)  runs the unit's MFMAs through the transposing LDS reads;
//   (iii) turns each accumulator tile (lane = row n, 4 consecutive k) into the master's lane order through a
//         private 16 x 16 fp32 LDS patch (the LDS traffic of one wave is in order: no barrier);
//   (iv)  applies Adam, stores p / m / v lane-linear again and the new weights as bf16 into the shadow image.
// While one wave waits for its moments the others are in their MFMA loops.  One barrier at the end (the caller
// may overwrite the operands), then the bias vector (its gradient = the ones column, parked in LDS by the unit
// that owns it) is updated by one thread per row.
//
// The p / m / v requests are hand-issued (inline asm) and hand-waited: the compiler's own s_waitcnt placement
// cannot count vector-memory operations across the wave-uniform branches of the unit loop (tile validity, shadow,
// gradient export) and falls back to vmcnt(0) at every use -- which waited for the NEXT unit's requests and for the
// previous tile's store acknowledgements, i.e. two exposed memory round trips per unit (round 2: 12.6 B/clk per CU
// in these phases on an EMPTY chip).  Here every request is unconditional (out-of-range tiles read a clamped
// address and are not stored), so a wave knows how many operations are younger than the set it is about to use:
// 6 requests of the next unit + the stores of the previous one, which may all stay in flight.  Two register sets
// (A / B) swap roles from unit to unit; the loop is unrolled by two so that no set is ever copied while its loads
// are in flight.
struct WgT {
  int64_t w_off;      // master offset of tile (0, 0) of this row block
  int64_t b_off;      // master offset of the bias of row 0 (< 0: the pass has no bias column)
  GAS char* sh;       // shadow image of (row 0, pass column 0); nullptr: none
  int sh_pitch;       // bytes per image row
  GAS float* sh_b;    // fp32 bias copy that travels with the image (row 0); nullptr: none
  float* patch;       // LDS, NWAVES * PATCH_FLOATS floats
};
struct WgGeom { int N, K, k_base, ncols; WgT T; };
struct PMV { f32x4 p0, m0, v0, p1, m1, v1; };
__device__ __forceinline__ int wg_units(const WgGeom& G) { return ((G.N + 15) >> 4) * ((((G.ncols + 15) >> 4) + 1) >> 1); }
__device__ __forceinline__ int wg_bias_pair(const WgGeom& G) {
  const bool has_bias = (G.T.b_off >= 0) && (G.K >= G.k_base) && (G.K < G.k_base + G.ncols);
  return has_bias ? ((G.K - G.k_base) >> 5) : -1;
}
// 16 bytes per lane from (wave-uniform 64-bit base in SGPRs) + (32-bit byte offset per lane); NT: streaming policy
// (the moments are touched once per step: they should not evict what is re-read).  "+v": the destination is tied, so
// the register allocator never has a reason to move the value between the request and the wait.
#define NM_GLOAD16(dst, voff, sbase) \
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory")
#define NM_GLOAD16_NT(dst, voff, sbase) \
  asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory")
#define NM_GLOAD4(dst, voff, sbase) \
  asm volatile("global_load_dword %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory")
constexpr int WG_LOADS = 6;      // vector-memory operations of one wg_issue

// Returns a LOWER bound of the vector-memory operations this wave issued here (for the caller's counted waits on
// copies it requested before the call).
// `pending` >= 0: the caller has LDS-DMA copies of the operands in flight (and `pending` vector-memory operations issued
// after them); they are waited for here, AFTER this pass's first requests are on their way (one memory round trip saved
// per call), followed by the workgroup barrier that makes them visible.
template <bool SCALAR_TR>
__device__ __forceinline__ int wgrad_adam(const Ctx& cc, const __bf16* A, int lda, int a_col0, const __bf16* B, int ldb,
                                          const WgGeom& G, int pending = -1) {
  Ctx c = cc;
  relaunder(c);
  const nm_job_t* J = c.job;
  const WgT& T = G.T;
  const int N = G.N, K = G.K, k_base = G.k_base, ncols = G.ncols;
  const int KT = ktiles(K), kt0 = k_base >> 4;
  const int nktp = (ncols + 15) >> 4;           // k tiles of this pass
  const int nkp = (nktp + 1) >> 1;              // pairs
  const int nunits = wg_units(G);
  const int kb = K - k_base;                    // pass column of the ones column
  const int kb_pair = wg_bias_pair(G), kb_j = (kb >> 4) & 1, kb_col = kb & 15;
  const bool do_adam = (c.flags & NM_F_ADAM) != 0;
  const bool do_grads = (c.flags & NM_F_GRADS) != 0;
  const AdamK ak = adam_consts(c);
  gf32 Pp = asg(J->params), Mp = asg(J->adam_m), Vp = asg(J->adam_v);
  float* patch = T.patch + c.wave * PATCH_FLOATS;
  const int prow = c.lane >> 2, pcol = (c.lane & 3) * 4;          // this lane's element group inside a tile
  // patch swizzle: 16-byte block b of row r sits at block b ^ ((r >> 1) & 3) -- the accumulator write (lane = row,
  // block = lane group) and the master-order read (lane = 4 row + block) are both conflict-free
  float* const pw = patch + c.c16 * 16 + 4 * (c.g ^ ((c.c16 >> 1) & 3));
  const float* const pr = patch + prow * 16 + 4 * ((c.lane & 3) ^ ((prow >> 1) & 3));
  const float* const pb = patch + prow * 16 + 4 * ((kb_col >> 2) ^ ((prow >> 1) & 3)) + (kb_col & 3);
  const unsigned lane16 = (unsigned)c.lane << 4;

  // the bias rows' p / m / v (waves 0 and 1, one row per lane): requested first, used after the unit loop
  const bool bias_wave = kb_pair >= 0 && do_adam && c.wave < 2;
  float bp, bm, bv;
  asm volatile("" : "=v"(bp), "=v"(bm), "=v"(bv));
  if (bias_wave) {
    const unsigned boff = (unsigned)(T.b_off + min(c.tid, N - 1)) << 2;
    NM_GLOAD4(bp, boff, Pp); NM_GLOAD4(bm, boff, Mp); NM_GLOAD4(bv, boff, Vp);
  }
  int young = 0;                                 // this wave's vector-memory operations since the bias request (capped)

  // request p / m / v of both tiles of unit u (wave-uniform) into set s: always WG_LOADS operations
  auto issue = [&](int u, PMV& s) {
    const int nt = u / nkp, kp = u - nt * nkp;            // wave-uniform: scalar division
    // a tile outside the pass (the odd tile of the last pair, a bias-only column block) is requested anyway -- the count of
    // operations must not depend on the unit -- from the first KiB of the buffers: every wave's filler hits the same hot
    // lines (L1 / L2), so it costs no memory traffic; it is never stored
    const int ktl0 = 2 * kp, ktl1 = 2 * kp + 1;
    const bool v0 = ktl0 < nktp && kt0 + ktl0 < KT, v1 = ktl1 < nktp && kt0 + ktl1 < KT;
    const unsigned o0 = (v0 ? (unsigned)((T.w_off + ((int64_t)(nt * KT + kt0 + ktl0) << 8)) << 2) : 0u) + lane16;
    const unsigned o1 = (v1 ? (unsigned)((T.w_off + ((int64_t)(nt * KT + kt0 + ktl1) << 8)) << 2) : 0u) + lane16;
    NM_GLOAD16(s.p0, o0, Pp); NM_GLOAD16_NT(s.m0, o0, Mp); NM_GLOAD16_NT(s.v0, o0, Vp);
    NM_GLOAD16(s.p1, o1, Pp); NM_GLOAD16_NT(s.m1, o1, Mp); NM_GLOAD16_NT(s.v1, o1, Vp);
  };
  // set s is complete once at most `younger` of this wave's vector-memory operations are outstanding
  auto wait_set = [&](PMV& s, int younger) {
    wait_vm(younger);
    asm volatile("" : "+v"(s.p0), "+v"(s.m0), "+v"(s.v0), "+v"(s.p1), "+v"(s.m1), "+v"(s.v1));
  };
  // MFMAs of unit u: acc[j] = k tile 2 kp + j of n tile nt
  auto unit_mfma = [&](int u, f32x4 (&acc)[2]) {
    const int nt = u / nkp, kp = u - nt * nkp;
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
    const int ncol0 = a_col0 + nt * 16;
    if (SCALAR_TR) {
      for (int rs = 0; rs < ROWS / 32; ++rs) {
        bf16x8 nf = lds_frag_tr_scalar(A, lda, rs * 32, ncol0, c.lane);
        acc[0] = mfma(lds_frag_tr_scalar(B, ldb, rs * 32, kp * 32, c.lane), nf, acc[0]);
        acc[1] = mfma(lds_frag_tr_scalar(B, ldb, rs * 32, kp * 32 + 16, c.lane), nf, acc[1]);
      }
    } else {
      unsigned na = tr_addr_il(A, lda, 0, ncol0, c.lane);
      unsigned ka = tr_addr_il(B, ldb, 0, kp * 32, c.lane);
      const unsigned n_step = 32u * lda * 2u, k_step = 32u * ldb * 2u;
      const unsigned n4 = 1u * lda * 2u, k4 = 1u * ldb * 2u;         // second read of a pair: the odd rows
#pragma unroll 2
      for (int rs = 0; rs < ROWS / 32; rs += 2) {
        // two row steps per wait: 12 transposing reads in flight (n side once, two k tiles)
        bf16x4 n0v, n1v, n2v, n3v, a0, a1, a2, a3, b0, b1, b2, b3;
        unsigned na1 = na + n4, ka1 = ka + k4, na2 = na + n_step, ka2 = ka + k_step;
        unsigned na3 = na2 + n4, ka3 = ka2 + k4;
        NM_TR_READ(n0v, na, 0);  NM_TR_READ(n1v, na1, 0);
        NM_TR_READ(a0, ka, 0);   NM_TR_READ(a1, ka1, 0);
        NM_TR_READ(b0, ka, 32);  NM_TR_READ(b1, ka1, 32);
        NM_TR_READ(n2v, na2, 0); NM_TR_READ(n3v, na3, 0);
        NM_TR_READ(a2, ka2, 0);  NM_TR_READ(a3, ka3, 0);
        NM_TR_READ(b2, ka2, 32); NM_TR_READ(b3, ka3, 32);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(n0v), "+v"(n1v), "+v"(n2v), "+v"(n3v), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0),
                       "+v"(b1), "+v"(b2), "+v"(b3));
        bf16x8 nf0 = join4(n0v, n1v), nf1 = join4(n2v, n3v);
        acc[0] = mfma(join4(a0, a1), nf0, acc[0]);
        acc[1] = mfma(join4(b0, b1), nf0, acc[1]);
        acc[0] = mfma(join4(a2, a3), nf1, acc[0]);
        acc[1] = mfma(join4(b2, b3), nf1, acc[1]);
        na += 2 * n_step; ka += 2 * k_step;
      }
    }
  };
  // Per tile: accumulators -> master lane order -> Adam -> stores.  MFMA lane (c16, g) holds
  // dW[n = nt*16 + c16][kk = ktl*16 + 4g .. +3]; master lane L holds row L / 4, columns 4 (L % 4) .. +3 of the tile.
  // Returns the number of vector-memory stores this wave issued (wave-uniform).
  auto finish_unit = [&](int u, PMV& s, const f32x4 (&acc)[2]) {
    const int nt = u / nkp, kp = u - nt * nkp;
    const int n = nt * 16 + prow;
    int nst = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ktl = 2 * kp + j;
      *reinterpret_cast<f32x4*>(pw) = acc[j];
      f32x4 g = *reinterpret_cast<const f32x4*>(pr);
      const float bg = *pb;
      const int k0 = k_base + ktl * 16 + pcol;
      if (nt * 16 + 16 > N || k_base + ktl * 16 + 16 > K) {      // wave-uniform: an edge tile -- pad rows / columns keep zero gradient
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = (n < N && k0 + i < K) ? g[i] : 0.f;
      }
      if (ktl < nktp && kt0 + ktl < KT) {                                            // wave-uniform
        const int64_t idx = T.w_off + ((int64_t)(nt * KT + kt0 + ktl) << 8) + c.lane * 4;
        if (do_grads) { *(GAS f32x4*)(asg(J->grads) + idx) = g; ++nst; }
        if (do_adam) {
          f32x4 p4 = j ? s.p1 : s.p0, m4 = j ? s.m1 : s.m0, v4 = j ? s.v1 : s.v0;
#pragma unroll
          for (int i = 0; i < 4; ++i) { float pp = p4[i], mm = m4[i], v2 = v4[i]; adam1(ak, g[i], pp, mm, v2); p4[i] = pp; m4[i] = mm; v4[i] = v2; }
          *(GAS f32x4*)(Pp + idx) = p4;
          __builtin_nontemporal_store(m4, (GAS f32x4*)(Mp + idx));
          __builtin_nontemporal_store(v4, (GAS f32x4*)(Vp + idx));
          nst += 3;
          if (T.sh) {
            bf16x4 pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) pk[i] = (__bf16)p4[i];
            *(GAS bf16x4*)(T.sh + (int64_t)n * T.sh_pitch + (ktl * 16 + pcol) * 2) = pk;
            ++nst;
          }
        }
      }
      if (kp == kb_pair && j == kb_j) {              // wave-uniform: this tile carries the ones column
        if (pcol == 0 && n < N) c.bgrad[n] = bg;     // one lane per row; consumed after the barrier below
      }
    }
    return nst;
  };

  PMV sa, sb;
  asm volatile("" : "=v"(sa.p0), "=v"(sa.m0), "=v"(sa.v0), "=v"(sa.p1), "=v"(sa.m1), "=v"(sa.v1));
  asm volatile("" : "=v"(sb.p0), "=v"(sb.m0), "=v"(sb.v0), "=v"(sb.p1), "=v"(sb.m1), "=v"(sb.v1));
  int u = c.wave, st_prev = 0;
  if (do_adam && u < nunits) { issue(u, sa); young += WG_LOADS; }
  if (pending >= 0) {
    wait_vm(min(pending + young + (bias_wave ? 3 : 0), 20));
    lds_barrier();
  }
  while (u < nunits) {
    f32x4 acc[2];
    const int u1 = u + NWAVES, u2 = u + 2 * NWAVES;
    const bool h1 = do_adam && u1 < nunits;
    if (h1) { issue(u1, sb); young += WG_LOADS; }
    unit_mfma(u, acc);
    if (do_adam) wait_set(sa, st_prev + (h1 ? WG_LOADS : 0));
    st_prev = finish_unit(u, sa, acc);
    young += st_prev;
    if (u1 >= nunits) break;
    const bool h2 = do_adam && u2 < nunits;
    if (h2) { issue(u2, sa); young += WG_LOADS; }
    unit_mfma(u1, acc);
    if (do_adam) wait_set(sb, st_prev + (h2 ? WG_LOADS : 0));
    st_prev = finish_unit(u1, sb, acc);
    young += st_prev;
    u = u2;
  }
  lds_barrier();                                    // every wave has finished reading A / B; the bias gradients are parked
  if (kb_pair >= 0 && c.wave < 2) {                 // wave-uniform
    relaunder(c);
    if (bias_wave) { wait_vm(min(young, 20)); asm volatile("" : "+v"(bp), "+v"(bm), "+v"(bv)); }
    const int n = c.tid;
    if (n < N) {
      const float bg = c.bgrad[n];
      const int64_t bidx = T.b_off + n;
      if (do_grads) asg(J->grads)[bidx] = bg;
      if (do_adam) {
        adam1(ak, bg, bp, bm, bv);
        Pp[bidx] = bp; Mp[bidx] = bm; Vp[bidx] = bv;
        if (T.sh_b) T.sh_b[n] = bp;
      }
    }
  }
  return young;
}

// ---- expert fusion (cVAE.py:1144-1164) on one (row, z) element --------------------------------
__device__ __forceinline__ int experts(const nm_job_t* J) { return J->M_enc > 0 ? J->M_enc : J->M; }
// Regression head: its first layer sees the residuals of the first `experts` modalities side by side, every modality
// padded to whole 64-column chunks (nm_job_t.reg_w): chunks of modality m start at chunk head_chunk0(J, m).
__host__ __device__ inline int head_chunk0(const nm_job_t* J, int m) {
  int q = 0;
  for (int i = 0; i < m; ++i) q += (J->mod[i].D + XCH - 1) / XCH;
  return q;
}
struct Fuse { float mu, lv, var; };
struct Lat { float mu[NM_MAX_EXP], lv[NM_MAX_EXP]; };     // always indexed by unrolled constants
__device__ __forceinline__ void softmax_alpha(const nm_job_t* J, float (&al)[NM_MAX_EXP]) {
  float mx = -INFINITY;
#pragma unroll
  for (int m = 0; m < NM_MAX_EXP; ++m)
    if (m < experts(J)) mx = fmaxf(mx, asg(J->params)[J->mod[m].alpha]);
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < NM_MAX_EXP; ++m) {
    al[m] = (m < experts(J)) ? expf(asg(J->params)[J->mod[m].alpha] - mx) : 0.f;
    s += al[m];
  }
#pragma unroll
  for (int m = 0; m < NM_MAX_EXP; ++m) al[m] /= s;
}
__device__ __forceinline__ Fuse fuse_fwd(const nm_job_t* J, const Lat& L, const float (&al)[NM_MAX_EXP]) {
  const int M = experts(J);
  Fuse f;
  if (M == 1 && J->single_bypass) { f.mu = L.mu[0]; f.var = expf(L.lv[0]); f.lv = logf(f.var); return f; }
  const int cb = J->combine;
  float S = 0.f, Smu = 0.f, sm = 0.f, sv = 0.f;
#pragma unroll
  for (int m = 0; m < NM_MAX_EXP; ++m) {
    if (m < M) {
      float var = expf(L.lv[m]);
      // POE2V (mvtCAE 'poe', cVAE.py:1782-1783 + 1481-1489): the variances stand where ProductOfExperts2 expects log
      // variances, so the precisions are exp(-var_m)
      float w = (cb == NM_COMBINE_POE2V) ? expf(-var) : ((cb == NM_COMBINE_GPOE) ? al[m] / var : 1.0f / var);
      S += w; Smu += L.mu[m] * w;
      sm += L.mu[m]; sv += var;
    }
  }
  if (cb == NM_COMBINE_MOE) { f.mu = sm / M; f.var = sv / M; }
  else {
    f.mu = Smu / S; f.var = 1.0f / S;
    if (cb == NM_COMBINE_MOPOE) { f.mu = (sm + f.mu) / (M + 1); f.var = (sv + f.var) / (M + 1); }
    if (cb == NM_COMBINE_POE2V) f.var = logf(1.0f / S);          // ... and its "logvar" is taken as the joint variance
  }
  if (J->var_floor > 0.f) f.var = fmaxf(f.var, J->var_floor);   // torch.clamp(variance_multimodal, min=1e-6), cVAE.py:1823
  f.lv = logf(f.var);
  return f;
}
// backward of the fusion: (d mu_j, d lv_j) -> (d mu_m, d lv_m) and d alpha_m (gPoE) for EVERY expert
struct FuseGrad { float dmu[NM_MAX_EXP], dlv[NM_MAX_EXP], dal[NM_MAX_EXP]; };
__device__ __forceinline__ FuseGrad fuse_bwd(const nm_job_t* J, const Lat& L, const float (&al)[NM_MAX_EXP], float dmu_j,
                                             float dlv_j) {
  const int M = experts(J);
  FuseGrad G;
#pragma unroll
  for (int m = 0; m < NM_MAX_EXP; ++m) { G.dmu[m] = 0.f; G.dlv[m] = 0.f; G.dal[m] = 0.f; }
  if (M == 1 && J->single_bypass) { G.dmu[0] = dmu_j; G.dlv[0] = dlv_j; return G; }
  const int cb = J->combine;
  float S = 0.f, Smu = 0.f, sv = 0.f;
#pragma unroll
  for (int m = 0; m < NM_MAX_EXP; ++m) {
    if (m < M) {
      float w = (cb == NM_COMBINE_POE2V) ? expf(-expf(L.lv[m])) : expf(-L.lv[m]) * ((cb == NM_COMBINE_GPOE) ? al[m] : 1.0f);
      S += w; Smu += L.mu[m] * w;
      sv += expf(L.lv[m]);
    }
  }
  if (J->var_floor > 0.f) {                        // a clamped joint variance passes no gradient
    const float var_p0 = 1.0f / S;
    const float var_u = (cb == NM_COMBINE_MOE) ? sv / M : (cb == NM_COMBINE_MOPOE) ? (sv + var_p0) / (M + 1)
                        : (cb == NM_COMBINE_POE2V) ? logf(var_p0) : var_p0;
    if (!(var_u > J->var_floor)) dlv_j = 0.f;
  }
  if (cb == NM_COMBINE_POE2V) {
    // p_m = exp(-v_m), v_m = exp(lv_m); mu_j = sum mu_m p_m / S; u = -log S; lv_j = log u:
    //   d mu_j / d mu_m = p_m / S,  d mu_j / d lv_m = -(mu_m - mu_j) p_m v_m / S,  d lv_j / d lv_m = v_m p_m / (u S)
    const float mu_p = Smu / S, u = logf(1.0f / S);
#pragma unroll
    for (int m = 0; m < NM_MAX_EXP; ++m) {
      if (m < M) {
        const float v = expf(L.lv[m]), r = expf(-v) / S;
        G.dmu[m] = dmu_j * r;
        G.dlv[m] = -dmu_j * (L.mu[m] - mu_p) * r * v + dlv_j * r * v / u;
      }
    }
    return G;
  }
  if (cb == NM_COMBINE_MOE) {
#pragma unroll
    for (int m = 0; m < NM_MAX_EXP; ++m)
      if (m < M) { G.dmu[m] = dmu_j / M; G.dlv[m] = dlv_j * expf(L.lv[m]) / sv; }   // d log(mean var) / d lv_m
    return G;
  }
  const float var_p = 1.0f / S, mu_p = Smu * var_p;
  float dmu_p = dmu_j, dlv_p = dlv_j, e_mu = 0.f, e_lv = 0.f;    // e_*: direct MoE branch of MoPoE
  if (cb == NM_COMBINE_MOPOE) {
    float var_j = (sv + var_p) / (M + 1);
    dmu_p = dmu_j / (M + 1);
    dlv_p = dlv_j * var_p / ((M + 1) * var_j);          // through var_p = exp(log var_p)
    e_mu = dmu_j / (M + 1);
    e_lv = dlv_j / ((M + 1) * var_j);
  }
#pragma unroll
  for (int m = 0; m < NM_MAX_EXP; ++m) {
    if (m < M) {
      float pm = expf(-L.lv[m]);
      float r = var_p * pm * ((cb == NM_COMBINE_GPOE) ? al[m] : 1.0f);     // sigma^2 * p_m
      G.dmu[m] = dmu_p * r + e_mu;
      G.dlv[m] = -dmu_p * r * (L.mu[m] - mu_p) + dlv_p * r + e_lv * expf(L.lv[m]);
      if (cb == NM_COMBINE_GPOE) G.dal[m] = dmu_p * var_p * pm * (L.mu[m] - mu_p) - dlv_p * var_p * pm;
    }
  }
  return G;
}
__device__ __forceinline__ float pick(const float (&a)[NM_MAX_EXP], int m) {
  // a select chain, kept opaque: left alone the compiler turns it back into a[m], i.e. a private array in
  // scratch memory (12 floats stored and one reloaded per element of the fusion backward loop)
  float r = a[0];
#pragma unroll
  for (int q = 1; q < NM_MAX_EXP; ++q) {
    float t = (q == m) ? a[q] : r;
    asm volatile("" : "+v"(t));
    r = t;
  }
  return r;
}

// ----------------------------------------------------------------------------------------------
// The step: all phases for one tile of 256 rows.
// ----------------------------------------------------------------------------------------------
// MODE 0: the whole step.  Head models (regression / end-to-end, nm_head_step_kernel) run it as two passes around the
// head: MODE 1 = encoders, fusion and every decoder forward (exports on; all activations incl. every decoder's last
// hidden one saved), MODE 2 = the decoders' output chunks again from the saved activation -- now with the head's extra
// gradients -- and the whole backward; no second encoder / fusion / hidden-decoder forward.
// MODE 3: forward only (deviation pass, predictions): the one-pass step with the backward compiled out.
template <bool SCALAR_TR, int MODE = 0>
__device__ __forceinline__ void run_step(Ctx& c, int step) {
  const nm_job_t* J = c.job;
  const int M = J->M, L = J->L, Z = J->Z, C = J->C;
  const int Me = experts(J);                    // modalities that have an encoder
  const bool nl = J->non_linear != 0;
  const bool bwd = MODE != 1 && MODE != 3 && (c.flags & NM_F_BACKWARD) != 0;
  constexpr bool FWD_ONLY = MODE == 1 || MODE == 3;   // output chunks export only: see the chunk loop
  const bool save = bwd || MODE == 1;           // activations go to the workspace
  const bool exportf = MODE != 2 && (c.flags & NM_F_EXPORT) != 0;
  const WsLayout wl = ws_layout(M, L, Z);
  const int Zs = wl.Zs;
  const bool split = c.part >= 0;               // this workgroup runs one modality of the model (NM_F_SPLIT)
  const int part = split ? c.part : 0;
  const int S = J->n_private, Zc = Z - S;       // DMVAE family: private / shared latent columns (S = 0: all shared)
  const float rZc = Zc > 0 ? 1.0f / (float)Zc : 0.f;
  // latent phases four columns at a time when the rows divide evenly (measured: with Z = 10 the 12-column groups
  // leave half the threads a second, mostly padded round -- slower than the element loop; Z = 64: 2.5x faster)
  const bool vec4 = S == 0 && (Z & 3) == 0;
  // forward-only launch of a single-expert model (deviation pass, predictions from one modality): the latent draw is
  // made inside the heads' epilogue (fwd_heads), see there
  const bool fastlat = MODE == 3 && M == 1 && Me == 1 && J->single_bypass != 0 && !split && S == 0 && J->tc_weight == 0.f &&
                       J->w_off < 0 && !(c.flags & NM_F_ZGIVEN) && wl.Zs <= 32 &&
                       !((c.flags & NM_F_EXPORT) && (J->out_mu || J->out_logvar || J->out_z));
  __bf16* const zlds = reinterpret_cast<__bf16*>(c.stage);      // [256][32] bf16 in S (free until the last decoder layer)
  float kl_fast = 0.f;
  const bool sigm = J->out_kind == 1;           // sigmoid output, ll = -0.5 sum (x - x_hat)^2
  gf32 ws_mu_m = (gf32)(c.ws + wl.mu_m + (int64_t)(step & 1) * M * wl.lat);
  gf32 ws_lv_m = (gf32)(c.ws + wl.lv_m + (int64_t)(step & 1) * M * wl.lat);
  gf32 ws_mu_j = (gf32)(c.ws + wl.mu_j + part * wl.lat);
  gf32 ws_lv_j = (gf32)(c.ws + wl.lv_j + part * wl.lat);
  gf32 ws_es = (gf32)(c.ws + wl.es + part * wl.lat);
  gf32 ws_dz0 = (gf32)(c.ws + wl.dz);            // d z of decoder m at + m * 256 * Zs (one copy per decoder)
  GAS char* ws_enc = c.ws + wl.enc_act;         // activation images [256][LDP], ACT_BYTES each
  GAS char* ws_dec0 = c.ws + wl.dec_act + (int64_t)part * L * wl.act;
  GAS char* ws_zc0 = c.ws + wl.zc + part * wl.act;
  GAS unsigned* sync_a = (GAS unsigned*)(c.ws + wl.sync);
  GAS unsigned* sync_b = sync_a + 16;
  GAS unsigned* sync_err = sync_a + 32;
  const unsigned sync_target = (unsigned)(c.lstep + 1) * (unsigned)c.nparts;
  GAS char* wsh = (GAS char*)J->wsh;
  char* const Sb = reinterpret_cast<char*>(c.stage);
  char* const Qb = reinterpret_cast<char*>(c.Q);
  char* const Pb = reinterpret_cast<char*>(c.P);

  // ================= encoders =================
  for (int m = 0; m < (MODE == 2 ? 0 : Me); ++m) {
    if (split && m != part) continue;
    relaunder(c);
    const nm_modality_t& md = J->mod[m];
    gbf16 save0 = save ? (gbf16)(ws_enc + (int64_t)(m * L + 0) * ACT_BYTES) : (gbf16)nullptr;
    // the image of the phase after the first layer goes to the lower half of Q
    // (rows / inputs of that blob: the second encoder layer, or the heads' [mu | logvar] rows)
    const GAS char* after0 = wsh + (L > 1 ? md.enc_s[1] : md.heads_s);
    fwd_first_layer(c, (const GAS char*)asg(md.xb) + (int64_t)(c.row0 / ROWS) * ((md.Kx + XCH - 1) / XCH) * XIMG_BYTES, md.Kx,
                    wsh + md.enc_s[0], blob_to_half(c, after0, 0, L > 1 ? J->H[1] : 2 * Zs, J->H[0]), J->H[0], nl, save0, true);
    prof(c, PH_ENC_L0);
    int half = 0;
    for (int e = 1; e < L; ++e) {
      gbf16 sv = save ? (gbf16)(ws_enc + (int64_t)(m * L + e) * ACT_BYTES) : (gbf16)nullptr;
      const GAS char* nxt = wsh + (e + 1 < L ? md.enc_s[e + 1] : md.heads_s);
      fwd_layer(c, half, blob_to_half(c, nxt, half ^ 1, e + 1 < L ? J->H[e + 1] : 2 * Zs, J->H[e]), J->H[e], J->H[e - 1], nl, sv,
                save ? act_stores(act_segs(J->H[e - 1])) : 0);
      half ^= 1;
    }
    tr(c, 1);
    prof(c, PH_ENC_REST);
    fwd_heads(c, half, no_next(), Z, J->H[L - 1], ws_mu_m + (int64_t)m * ROWS * Zs, ws_lv_m + (int64_t)m * ROWS * Zs, Zs,
              save ? act_stores(act_segs(J->H[L - 1])) : 0, fastlat ? zlds : (__bf16*)nullptr, step, vec4, &kl_fast);
    prof(c, PH_HEADS);
  }

  // ================= fusion + reparameterisation + KL =================
  // the heads' mu / logvar stores are complete (split: of every part, made visible across workgroups)
  if (split) { if (!split_handoff(c, sync_a, sync_err, sync_target)) return; }
  else handoff_barrier();
  // first decoder layer's image: requested now, lands during the latent arithmetic
  if (MODE != 2) issue_next(c, blob_to_half(c, wsh + J->mod[split ? part : 0].dec_s[0], 0, J->H[L - 1], Z + C));
  float al[NM_MAX_EXP] = {0.f, 0.f, 0.f, 0.f};
  if (J->combine == NM_COMBINE_GPOE && !(Me == 1 && J->single_bypass)) softmax_alpha(J, al);
  auto load_lat = [&](Lat& Lt, int r, int z) {
#pragma unroll
    for (int m = 0; m < NM_MAX_EXP; ++m) {
      Lt.mu[m] = (m < Me) ? ws_mu_m[((int64_t)m * ROWS + r) * Zs + z] : 0.f;
      Lt.lv[m] = (m < Me) ? ws_lv_m[((int64_t)m * ROWS + r) * Zs + z] : 0.f;
    }
  };
  // learnable per-modality loss weights (WeightedDMVAE.weights, cVAE.py:1650, 1693-1697): read once per step, before
  // any of them is updated
  // mvtCAE's total-correlation term (cVAE.py:1862-1869): tc = - sum_z mean_m logsumexp_rows(mu_m[:, z]) -- the joint
  // posterior's half of it is a scalar minus its own mean, identically zero.  One wave per (expert, latent column):
  // max and sum over the rows by shuffles (fixed order), kept in LDS for the backward pass.
  float tc = 0.f;
  if (MODE != 2 && J->tc_weight != 0.f) {
    relaunder(c);
    for (int col = c.wave; col < Me * Z; col += NWAVES) {
      const int m = col / Z, z = col - m * Z;
      float v[4], mx = -3.0e38f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = c.lane + 64 * k;
        const float x = ws_mu_m[((int64_t)m * ROWS + min(r, c.nrows - 1)) * Zs + z];
        v[k] = (r < c.nrows) ? x : -3.0e38f;
        mx = fmaxf(mx, v[k]);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      float sx = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) sx += (c.lane + 64 * k < c.nrows) ? expf(v[k] - mx) : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sx += __shfl_xor(sx, o, 64);
      if (c.lane == 0) c.lse[col] = mx + logf(sx);
    }
    lds_barrier();
    for (int z = 0; z < Z; ++z) {
      float sm_ = 0.f;
      for (int m = 0; m < Me; ++m) sm_ += c.lse[m * Z + z];
      tc -= sm_ / (float)Me;
    }
  }
  // (weights[m] is read at the start of decoder m and updated at its end, by that decoder only; their sum, the
  //  weight of the KL term, is formed here, before any of them moves)
  float kl_w = J->kl_weight;
  if (J->w_off >= 0) {
    kl_w = 0.f;
    for (int m = 0; m < M; ++m) kl_w += asg(J->params)[J->w_off + m];
  }
  float kl_part = 0.f;
  relaunder(c);
  // Four latent columns of one row per iteration (16-byte loads of every expert's mu / logvar, 16-byte stores of the
  // joint statistics).  The element-at-a-time
  // loop spent its time waiting -- each iteration's loads queue behind the previous iteration's stores.
  if (fastlat) {
    kl_part = kl_fast;                             // (the draw and the KL terms were formed in the heads' epilogue)
  } else if (vec4) {
    const int nq4 = (Z + 3) >> 2;
    const float rq4 = 1.0f / (float)nq4;
#pragma unroll 2
    for (int e = c.tid; e < (MODE == 2 ? 0 : ROWS * nq4); e += WG) {
      const int r = idiv(e, nq4, rq4), z0 = 4 * (e - r * nq4);
      f32x4 mu4[NM_MAX_EXP], lv4[NM_MAX_EXP];
#pragma unroll
      for (int m = 0; m < NM_MAX_EXP; ++m) {
        mu4[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        lv4[m] = mu4[m];
        if (m < Me) {
          mu4[m] = *(const GAS f32x4*)(ws_mu_m + ((int64_t)m * ROWS + r) * Zs + z0);
          lv4[m] = *(const GAS f32x4*)(ws_lv_m + ((int64_t)m * ROWS + r) * Zs + z0);
        }
      }
      float ep[4];
      if (J->eps) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ep[i] = asg(J->eps)[((int64_t)(step % J->eps_cap) * ROWS + r) * Z + min(z0 + i, Z - 1)];
      } else {
        randn2_ctr(J->seed, (uint32_t)step, (uint32_t)(c.row0 + r), (uint32_t)(z0 >> 1), ep[0], ep[1]);
        randn2_ctr(J->seed, (uint32_t)step, (uint32_t)(c.row0 + r), (uint32_t)(z0 >> 1) + 1u, ep[2], ep[3]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) ep[i] = (z0 + i < Z) ? ep[i] : 0.f;
      f32x4 omu, olv, oes;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Lat Lt;
#pragma unroll
        for (int m = 0; m < NM_MAX_EXP; ++m) { Lt.mu[m] = mu4[m][i]; Lt.lv[m] = lv4[m][i]; }
        Fuse f = fuse_fwd(J, Lt, al);
        float es = ep[i] * expf(0.5f * f.lv);
        if (c.flags & NM_F_ZGIVEN) { f.mu = ep[i]; es = 0.f; }     // decode(z, c, m): the draw buffer holds z itself
        omu[i] = f.mu; olv[i] = f.lv; oes[i] = es;
        if (r < c.nrows && z0 + i < Z) {
          kl_part += -0.5f * (1.0f + f.lv - f.mu * f.mu - expf(f.lv));
          if (exportf && part == 0) {
            int64_t gr = (int64_t)(c.row0 + r) * Z + z0 + i;
            if (J->out_mu) asg(J->out_mu)[gr] = f.mu;
            if (J->out_logvar) asg(J->out_logvar)[gr] = f.lv;
            if (J->out_z) asg(J->out_z)[gr] = f.mu + es;
          }
        }
      }
      *(GAS f32x4*)(ws_mu_j + r * Zs + z0) = omu;
      *(GAS f32x4*)(ws_lv_j + r * Zs + z0) = olv;
      *(GAS f32x4*)(ws_es + r * Zs + z0) = oes;
    }
  } else {
  // (private columns: element at a time) shared latent column z = head column S + z
#pragma unroll 2
  for (int e = c.tid; e < (MODE == 2 ? 0 : ROWS * Zc); e += WG) {
    int r = idiv(e, Zc, rZc), z = e - r * Zc;
    Lat Lt;
    load_lat(Lt, r, S + z);
    Fuse f = fuse_fwd(J, Lt, al);
    float ep = J->eps ? asg(J->eps)[((int64_t)(step % J->eps_cap) * ROWS + r) * Z + z]
                      : randn_ctr(J->seed, (uint32_t)step, (uint32_t)(c.row0 + r), (uint32_t)z);
    float es = ep * expf(0.5f * f.lv);
    if (c.flags & NM_F_ZGIVEN) { f.mu = ep; es = 0.f; }     // decode(z, c, m): the draw buffer holds z itself
    float zz = f.mu + es;
    ws_mu_j[r * Zs + z] = f.mu;
    ws_lv_j[r * Zs + z] = f.lv;
    ws_es[r * Zs + z] = es;
    if (r < c.nrows) {
      kl_part += -0.5f * (1.0f + f.lv - f.mu * f.mu - expf(f.lv));
      if (exportf && part == 0) {
        int64_t gr = (int64_t)(c.row0 + r) * Z + z;
        if (J->out_mu) asg(J->out_mu)[gr] = f.mu;
        if (J->out_logvar) asg(J->out_logvar)[gr] = f.lv;
        if (J->out_z) asg(J->out_z)[gr] = zz;
      }
    }
  }
  }
  float kl = (MODE == 2) ? 0.f : block_sum(c, kl_part) * c.inv_b;          // calc_kl: sum over z, mean over rows
  if (fastlat) lds_barrier();                          // (z sits in LDS; the decoder image just requested stays in flight)
  else handoff_barrier();                              // mu_j / es are complete for build_zc
  tr(c, 3);
  prof(c, PH_LATENT);

  // ================= decoders (forward, NLL, and the whole decoder backward) =================
  float ll_sum = 0.f;
  for (int m = 0; m < M; ++m) {
    if (split && m != part) continue;
    relaunder(c);
    const nm_modality_t& md = J->mod[m];
    const int D = md.D;
    const int Kd0 = Z + C;
    const int nck = (D + OCH - 1) / OCH;
    const GAS char* oblob = wsh + md.out_s;
    // z | c | 1: built by the first decoder; the others reuse it when all tables carry the same covariates
    const bool reuse_zc = J->shared_cov && M > 1 && !split && S == 0;
    // (two-pass modes: every decoder keeps its own activations -- and its own z | c | 1 unless that one is shared)
    GAS char* const ws_dec = ws_dec0 + (MODE != 0 ? (int64_t)m * L * wl.act : 0);
    GAS char* const ws_zc = ws_zc0 + ((MODE != 0 && !reuse_zc) ? (int64_t)m * wl.act : 0);
    if (MODE == 2) {
      // second pass: the last hidden activation comes back from the workspace, chunk 0 of the output layer with it
      lds_barrier();                                   // P / S are drained by whatever ran before
      dma_act(c, ws_dec + (int64_t)(L - 1) * ACT_BYTES, Pb, 0, ROWS, act_segs(J->H[0]));
      dma_lin(c, oblob, Sb, OBLOB_BYTES >> 10);
    } else {
    if (m > 0 && !split) issue_next(c, blob_to_half(c, wsh + md.dec_s[0], 0, J->H[L - 1], Z + C));
    if (m == 0 || !reuse_zc) {
      build_zc(c, c.P, md, ws_mu_j, ws_es, Z, C, Zs, S, ws_mu_m + (int64_t)min(m, Me - 1) * ROWS * Zs,
               fastlat ? zlds : (const __bf16*)nullptr);
      lds_barrier();
      if (save || reuse_zc) store_act_img(c, (gbf16)ws_zc, c.P, act_segs(Kd0));
    } else {
      dma_act(c, ws_zc, Pb, 0, ROWS, act_segs(Kd0));   // waited for by the first layer (it waits for everything older)
    }
    tr(c, 4);
    prof(c, PH_DEC_ZC);
    // --- hidden decoder layers; the last one requests output chunk 0 into slot B (= S) ---
    int half = 0;
    for (int d = 0; d < L; ++d) {
      int Kin = (d == 0) ? Kd0 : J->H[L - d];
      int Nout = J->H[L - 1 - d];
      gbf16 sv = (save && (d < L - 1 || MODE == 1)) ? (gbf16)(ws_dec + (int64_t)d * ACT_BYTES) : (gbf16)nullptr;
      Next nx = (d + 1 < L) ? blob_to_half(c, wsh + md.dec_s[d + 1], half ^ 1, J->H[L - 2 - d], Nout)
                            : Next{oblob, Sb, OBLOB_BYTES >> 10, nullptr, nullptr, 0, 0};
      // (d == 0: the z | c | 1 build / reload sits between the image request and here -- wait for everything)
      fwd_layer(c, half, nx, Nout, Kin, nl, sv, (d > 0 && save) ? act_stores(act_segs(Kin)) : 0);
      half ^= 1;
    }
    }
    tr(c, 5);
    prof(c, PH_DEC_HID);
    // --- output layer in chunks of 64 ROI columns, fused with NLL, its backward and Adam ---
    // LDS during the chunk loop: P = last hidden activation; Q = [delta chunk [256][LDX] | slot A | patches];
    // S = slot B.  Chunk ch's blob ([64][LDP] weight rows, bias[64], logvar_out[64]) sits in slot B for even ch,
    // slot A for odd ch; the other slot receives chunk ch + 1 while chunk ch is processed.
    const int Hl = J->H[0];                       // width feeding the output layer
    const int KTo = ktiles(Hl);
    gcf32 xf = asg(md.x_f32);
    const int xp = md.x_pitch;
    __bf16* const Dq = c.Q;                       // delta chunk, row pitch LDX
    char* const slotA = Qb + XIMG_BYTES;
    float* const opatch = reinterpret_cast<float*>(Qb + XIMG_BYTES + OBLOB_BYTES);
    f32x4 accg[2][RT];
    zero_acc(accg);
    float nll_part = 0.f;
    if (exportf && md.out_rowdev) { for (int r = c.tid; r < ROWS; r += WG) c.rowacc[r] = 0.f; }
    // per row, the same for every chunk: the hinge's row coefficient (read once), the row's squared deviation (summed in
    // registers over the chunks, reduced once after the loop)
    // (two-pass modes only: the one-pass step has no registers to spare for them)
    constexpr int NRC = MODE == 2 ? RT : 1, NRD = FWD_ONLY ? RT : 1;
    float rcv[NRC], rdev[NRD];
#pragma unroll
    for (int rt = 0; rt < NRD; ++rt) rdev[rt] = 0.f;
#pragma unroll
    for (int rt = 0; rt < NRC; ++rt)
      rcv[rt] = (MODE == 2 && md.dloc_rowcoef) ? asg(md.dloc_rowcoef)[c.row0 + c.wm * WROWS + rt * 16 + c.c16] : 0.f;
    // read once, outside the per-lane selects below: a descriptor load inside `cond ? load * x : 0` becomes a
    // lane-divergent branch, and register spills placed around such branches are not safe with this compiler
    // (tools/check_spill_exec.py)
    const float ll_w = (J->w_off >= 0) ? asg(J->params)[J->w_off + m] : J->ll_weight;
    const float llw_b = ll_w * c.inv_b;
    // regression head: residual chunk images out (export), d loss / d x_hat chunk images in (second pass)
    const int hq_all = (J->reg_head && m < Me) ? head_chunk0(J, Me) : 0;
    const int64_t hq_m = (J->reg_head && m < Me) ? head_chunk0(J, m) : 0;
    GAS char* const res_out = (exportf && hq_all > 0 && J->reg_resid)           // (one set of images per 256-row tile)
        ? (GAS char*)asg(J->reg_resid) + ((int64_t)(c.row0 / ROWS) * hq_all + hq_m) * XIMG_BYTES : (GAS char*)nullptr;
    const GAS char* const dres_in = (MODE == 2 && hq_all > 0 && J->reg_dres)    // (one set: the batch in flight)
        ? (const GAS char*)asg(J->reg_dres) + hq_m * XIMG_BYTES : (const GAS char*)nullptr;
    int young_prev = RT;                          // this wave's vector-memory operations younger than the next blob request
    // fp32 inputs of chunk `chx` for this lane: 4 consecutive ROI of RT rows
    auto load_xin = [&](int chx, f32x4 (&xv)[RT]) {
      const int dcl = min(chx * OCH + c.wn * 16 + 4 * c.g, xp - 4);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        xv[rt] = *(const GAS f32x4*)(xf + (int64_t)(c.row0 + c.wm * WROWS + rt * 16 + c.c16) * xp + dcl);
    };
    // One output chunk.  xin: this chunk's fp32 inputs -- requested here, in flight during the MFMAs (training), or
    // already requested by the previous chunk (forward only: the chunk is too short to hide them, so chunk ch + 1's are
    // requested as soon as chunk ch's have arrived, into the other register set `xnx`).
    auto chunk = [&](const int ch, f32x4 (&xin)[RT], f32x4 (&xnx)[RT]) {
      relaunder(c);
      const int d0 = ch * OCH;
      const int valid = min(OCH, D - d0);
      char* const slot = (ch & 1) ? slotA : Sb;
      char* const other = (ch & 1) ? Sb : slotA;
      const __bf16* Wc = reinterpret_cast<const __bf16*>(slot);
      const float* vb = reinterpret_cast<const float*>(slot + OIMG_BYTES);      // bias[64], then logvar_out[64]
      // this chunk's weight-gradient target; p / m / v of this wave's two units are requested well ahead of the
      // weight-gradient phase: the first right below (in flight during the GEMM and the epilogue), the second
      // after the epilogue
      GAS char* const oimg = wsh + md.out_s + (int64_t)ch * OBLOB_BYTES;
      const WgGeom Go{valid, Hl, 0, rup(Hl + 1, 16),
                      WgT{md.out_w + (int64_t)(d0 >> 4) * KTo * 256, md.out_b + d0, oimg, LDP * 2, (GAS float*)(oimg + OIMG_BYTES), opatch}};
      if (c.tid < OCH) c.colacc[c.tid] = 0.f;
      // d logvar_out of this chunk is applied after the epilogue by one lane per column (wave 0): its p / m / v are
      // requested now (hand-issued: a plain load there would wait for every older store of this wave), and are complete
      // by then -- the epilogue consumes this chunk's fp32 inputs, which are requested after them (in-order return)
      const bool lvo_adam = bwd && (c.flags & NM_F_ADAM) && !sigm && c.wave == 0;
      float lvp, lvm, lvv;
      asm volatile("" : "=v"(lvp), "=v"(lvm), "=v"(lvv));
      if (lvo_adam) {
        const unsigned lo = (unsigned)(md.logvar_out + d0 + min(c.tid, valid - 1)) << 2;
        gf32 Pq = asg(J->params), Mq = asg(J->adam_m), Vq = asg(J->adam_v);
        NM_GLOAD4(lvp, lo, Pq); NM_GLOAD4(lvm, lo, Mq); NM_GLOAD4(lvv, lo, Vq);
      }
      // chunk ch's blob (requested a chunk ago) and everything older; from the second chunk on at least the RT
      // fp32 input loads of the previous chunk are younger than it and may stay in flight (with Adam: its last stores;
      // forward only: also the previous chunk's export stores -- waiting for THEIR acknowledgements was most of a
      // forward-only chunk)
      wait_vm(ch > 0 ? min(young_prev + (lvo_adam ? 3 : 0), 20) : 0);
      lds_barrier();                              // ... for every wave; also: the previous chunk is finished everywhere
      int n_blob = 0;
      if (ch + 1 < nck) n_blob = dma_lin(c, oblob + (int64_t)(ch + 1) * OBLOB_BYTES, other, OBLOB_BYTES >> 10);
      // fp32 inputs of the residual (rows are always inside the zero-padded table): in flight during the MFMAs
      const int dl0 = c.wn * 16 + 4 * c.g;        // first of the lane's 4 columns inside the chunk
      const int dg0 = d0 + dl0;
      if (!FWD_ONLY) load_xin(ch, xin);
      bf16x4 exh[RT];                               // the head's gradient on this lane's 4 x RT outputs (second pass only)
      if (MODE == 2 && dres_in) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          exh[rt] = *(const GAS bf16x4*)(dres_in + (int64_t)ch * XIMG_BYTES + ((c.wm * WROWS + rt * 16 + c.c16) * LDX + dl0) * 2);
      }
      // x_hat chunk: acc[rt] = features dl0..dl0+3 of row (wm, rt, c16)
      f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        const int ksteps = wpad(Hl) / 32;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks < ksteps) {
            const bf16x8 wf = lds_frag(Wc, LDP, c.wn * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
              bf16x8 a = lds_frag(c.P, LDP, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
              acc[rt] = mfma(wf, a, acc[rt]);
            }
          }
        }
      }
      tr(c, 6);
      if (FWD_ONLY) {
        // this chunk's inputs have been in flight since the previous chunk's epilogue: make sure of them, then
        // request the next chunk's
        // (younger than them and free to stay in flight: the previous chunk's export stores, the blob just requested)
        wait_vm(ch > 0 ? young_prev - RT + n_blob : 0);
        if (ch + 1 < nck) load_xin(ch + 1, xnx);
      }
      // epilogue: residual, NLL, d logvar_out, delta chunk -> Dq.  Lane: 4 consecutive ROI of one row.
      {
        const f32x4 bo = *reinterpret_cast<const f32x4*>(vb + dl0);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(vb + OCH + dl0);
        // per column: q = sum_r diff^2 (valid rows only).  Then  NLL = sum_d [0.5 e^{-s} q + n (0.5 s + log sqrt(2 pi))]
        // and d(-LL)/d s_d = (0.5 n - 0.5 e^{-s} q) / B: one masked square-accumulate per element instead of
        // evaluating both sums element by element.
        // (sigmoid / squared-error output, cVAE.py:1478, 1560: x_hat = sigmoid(a), ll = -0.5 sum (x - x_hat)^2 -- the same
        //  epilogue with unit precision, no logvar_out, and the sigmoid's derivative on the way back)
        float inv[4], colq[4], coef[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          inv[i] = sigm ? 1.0f : expf(-sv[i]);
          colq[i] = 0.f;
          coef[i] = (dg0 + i < D) ? llw_b * inv[i] : 0.f;                      // d total / d x_hat = coef * diff
        }
        int nvalid = 0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int r = c.wm * WROWS + rt * 16 + c.c16;
          const bool rv = r < c.nrows;
          nvalid += rv ? 1 : 0;
          acc[rt] += bo;                                                       // x_hat (pre-sigmoid with out_kind 1)
          float dsig[4] = {1.f, 1.f, 1.f, 1.f};
          if (sigm) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float sg = 1.0f / (1.0f + expf(-acc[rt][i]));
              acc[rt][i] = sg;
              dsig[i] = sg * (1.0f - sg);
            }
          }
          bf16x4 pk;
          f32x4 ex = {0.f, 0.f, 0.f, 0.f};
          if (md.dloc_extra)               // extra loss gradient on x_hat
            ex = *(const GAS f32x4*)(asg(md.dloc_extra) + (int64_t)(c.row0 + r) * xp + min(dg0, xp - 4));
          if (MODE == 2 && dres_in) {      // ... of the regression head (bf16 chunk image, requested before the GEMM)
#pragma unroll
            for (int i = 0; i < 4; ++i) ex[i] += (float)exh[rt][i];
          }
          float rc = rcv[MODE == 2 ? rt : 0];                              // contrastive hinge: rc * (x_hat - x)
          if (MODE != 2 && md.dloc_rowcoef) rc = asg(md.dloc_rowcoef)[c.row0 + r];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float diff = rv ? acc[rt][i] - xin[rt][i] : 0.f;
            colq[i] = fmaf(diff, diff, colq[i]);
            const bool dv = dg0 + i < D;
            pk[i] = (__bf16)((diff * (coef[i] + (dv ? rc : 0.f)) + ((rv && dv) ? ex[i] : 0.f)) * dsig[i]);
          }
          if (bwd) *reinterpret_cast<bf16x4*>(Dq + r * LDX + dl0) = pk;
          if (res_out) {                   // x - x_hat, zero on pad rows / columns: the head's first-layer operand
            bf16x4 rk;
#pragma unroll
            for (int i = 0; i < 4; ++i) rk[i] = (__bf16)((rv && dg0 + i < D) ? xin[rt][i] - acc[rt][i] : 0.f);
            *(GAS bf16x4*)(res_out + (int64_t)ch * XIMG_BYTES + (r * LDX + dl0) * 2) = rk;
          }
        }
        float colsum[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool dv = dg0 + i < D;
          const float hq = 0.5f * inv[i] * colq[i];
          nll_part += dv ? hq + (sigm ? 0.f : (float)nvalid * (0.5f * sv[i] + LOG_SQRT_2PI)) : 0.f;
          colsum[i] = dv ? 0.5f * (float)nvalid - hq : 0.f;
        }
        // exports share the fp32 table's row pitch: one 16-byte store each.  Forward only: stored for every row of the
        // tile (zeros on the rows past the table's end -- the buffers hold whole tiles) under a wave-uniform column
        // test, so that the number of stores a wave issues is known to the next chunk's wait.
        const bool wave_cols = d0 + c.wn * 16 < xp;
        if (exportf && (FWD_ONLY ? wave_cols : dg0 < xp)) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const int r = c.wm * WROWS + rt * 16 + c.c16;
            const bool rv = r < c.nrows;
            if (FWD_ONLY || rv) {
              f32x4 lo, sq;
              float rs = 0.f;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const bool dv = dg0 + i < D && rv;
                const float xh = acc[rt][i], diff = xh - xin[rt][i];
                lo[i] = dv ? xh : 0.f;
                sq[i] = dv ? diff * diff : 0.f;
                rs += sq[i];
              }
              const int64_t gi = (int64_t)(c.row0 + r) * xp + dg0;
              if (!FWD_ONLY || dg0 < xp) {
                if (MODE == 1) {                  // read back by the head phase of this workgroup
                  if (md.out_loc) *(GAS f32x4*)(asg(md.out_loc) + gi) = lo;
                  if (md.out_sqerr) *(GAS f32x4*)(asg(md.out_sqerr) + gi) = sq;
                } else {
                  if (md.out_loc) __builtin_nontemporal_store(lo, (GAS f32x4*)(asg(md.out_loc) + gi));       // written once,
                  if (md.out_sqerr) __builtin_nontemporal_store(sq, (GAS f32x4*)(asg(md.out_sqerr) + gi));   // read elsewhere
                }
              }
              if (FWD_ONLY) rdev[FWD_ONLY ? rt : 0] += rs;
              else if (md.out_rowdev) atomicAdd(&c.rowacc[r], rs);
            }
          }
        }
        if (FWD_ONLY)
          young_prev = RT + (res_out ? RT : 0) +
                       ((exportf && wave_cols) ? RT * ((md.out_loc ? 1 : 0) + (md.out_sqerr ? 1 : 0)) : 0);
        if (bwd) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float s = colsum[i];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 8, 64);
            if (c.c16 == 0 && dg0 + i < D) atomicAdd(&c.colacc[dl0 + i], s);
          }
        }
      }
      prof(c, PH_X_EPI);
      tr(c, 7);
      if (!bwd) return;                           // forward only: the next chunk's barrier protects the slots
      lds_barrier();                              // delta chunk and column sums complete
      relaunder(c);
      // d logvar_out for this chunk (master + the copy that travels with the chunk's image)
      if (lvo_adam) {                               // wave-uniform (wave 0)
        asm volatile("" : "+v"(lvp), "+v"(lvm), "+v"(lvv));
        if (c.tid < valid) {
          const float g = ll_w * c.colacc[c.tid] * c.inv_b;
          const int64_t idx = md.logvar_out + d0 + c.tid;
          if (c.flags & NM_F_GRADS) asg(J->grads)[idx] = g;
          adam1(adam_consts(c), g, lvp, lvm, lvv);
          asg(J->params)[idx] = lvp; asg(J->adam_m)[idx] = lvm; asg(J->adam_v)[idx] = lvv;
          ((GAS float*)(oblob + (int64_t)ch * OBLOB_BYTES + OIMG_BYTES))[OCH + c.tid] = lvp;
        }
      } else if (c.tid < valid && !sigm) {
        apply_grad(c, md.logvar_out + d0 + c.tid, ll_w * c.colacc[c.tid] * c.inv_b,
                   (GAS float*)(oblob + (int64_t)ch * OBLOB_BYTES + OIMG_BYTES) + OCH + c.tid);
      }
      // dgrad into the last hidden activation: accg[k][r] += sum_d Dq[r][d] Wo[d0 + d][k], weights from the slot
      dgrad_tile(c, accg, Dq, LDX, 0, Wc, LDP, OCH / 32);
      tr(c, 8);
      prof(c, PH_OUT_DGRAD);
      // wgrad + Adam of this chunk of decoder_mean_layer: dWo[d][k] = sum_r Dq[r][d] P[r][k]
      const int n_wg = wgrad_adam<SCALAR_TR>(c, Dq, LDX, 0, c.P, LDP, Go);
      young_prev = RT + n_wg;                     // all younger than the next chunk's blob request
      tr(c, 9);
      prof(c, PH_OUT_WGRAD);
    };
    if (FWD_ONLY) {
      // two register sets that swap roles from chunk to chunk (unrolled by two so that the sets keep their names)
      f32x4 xa[RT], xb[RT];
      load_xin(0, xa);
      for (int ch = 0; ch < nck; ch += 2) {
        chunk(ch, xa, xb);
        if (ch + 1 < nck) chunk(ch + 1, xb, xa);
      }
    } else {
      f32x4 xa[RT];
      for (int ch = 0; ch < nck; ++ch) chunk(ch, xa, xa);
    }
    float nll = block_sum(c, nll_part);
    float ll_this = -nll * c.inv_b;                 // compute_ll: sum over ROI, mean over rows
    if (J->w_off >= 0) {                            // WeightedDMVAE: ll_i * weights[i]; d total / d weights[i] = KL - ll_i
      if (c.tid == 0 && bwd) apply_grad(c, J->w_off + m, kl - ll_this, nullptr);
      ll_this *= ll_w;
    }
    ll_sum += ll_this;
    if (c.tid == 0 && J->loss_log)
      asg(J->loss_log)[(int64_t)(step % J->loss_cap) * NM_LOSS_STRIDE + NM_LOSS_LL_M + m] = ll_this;
    prof(c, PH_NLL_RED);
    if (exportf && md.out_rowdev) {
      if (FWD_ONLY) {
#pragma unroll
        for (int rt = 0; rt < NRD; ++rt) {          // the row's 4 column groups of this wave, then the 4 waves of the row half
          float v = rdev[rt];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          if (c.g == 0) atomicAdd(&c.rowacc[c.wm * WROWS + rt * 16 + c.c16], v);
        }
      }
      lds_barrier();
      for (int r = c.tid; r < c.nrows; r += WG) asg(md.out_rowdev)[c.row0 + r] = c.rowacc[r] / (float)D;
    }
    if (!bwd) { lds_barrier(); continue; }

    // --- decoder hidden layers, backward ---
    // state: P = activation g_{L-1}, accg = pre-mask delta of g_{L-1}
    finish_delta(c, accg, c.P, Hl, nl);             // mask source is P itself (same element)
    lds_barrier();
    prof(c, PH_DEC_FINISH);
    float* const spatch = reinterpret_cast<float*>(Sb + SPATCH_OFF);
    for (int d = L - 1; d >= 0; --d) {
      relaunder(c);
      int Kin = (d == 0) ? Kd0 : J->H[L - d];
      int Nout = J->H[L - 1 - d];
      f32x4 acc[2][RT];
      zero_acc(acc);
      const GAS char* act_img = d == 0 ? ws_zc : ws_dec + (int64_t)(d - 1) * ACT_BYTES;
      GAS char* const dimg = wsh + md.dec_s[d];
      const WgGeom Gd{Nout, Kin, 0, rup(Kin + 1, 16),
                      WgT{md.dec_w[d], md.dec_b[d], dimg, blob_kp(Kin) * 2, (GAS float*)(dimg + cimg_bytes(Nout, Kin)), spatch}};
      dgrad_hidden(c, acc, dimg, act_img, Nout, Kin);
      tr(c, 10);
      prof(c, PH_DEC_DGRAD);
      wgrad_adam<SCALAR_TR>(c, c.P, LDP, 0, c.Q, LDP, Gd, 0);
      tr(c, 11);
      prof(c, PH_DEC_WGRAD);
      if (d > 0) {
        finish_delta(c, acc, c.Q, Kin, nl);
        lds_barrier();
      } else {
        // d z of this decoder
        const int ntk = wpad(Kin) / 16;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          int kt = c.wn + 4 * t;
          if (kt >= ntk) continue;
          int k0 = kt * 16 + 4 * c.g;
          if (k0 < Z) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
              int r = c.wm * WROWS + rt * 16 + c.c16;
              // decoder m's own copy (columns >= Z of the row are never read); the sum over the decoders is formed
              // where it is used, in decoder order
              *(GAS f32x4*)(ws_dz0 + (int64_t)m * ROWS * Zs + r * Zs + k0) = acc[t][rt];
            }
          }
        }
        lds_barrier();
      }
      prof(c, PH_DEC_DELTA);
    }
  }

  // ================= loss log =================
  // d z (and, split: ll_m) of every decoder is complete
  if (split) { if (!split_handoff(c, sync_b, sync_err, sync_target)) return; }
  else handoff_barrier();
  if (MODE != 2 && c.tid == 0 && J->loss_log && part == 0) {
    gf32 row = asg(J->loss_log) + (int64_t)(step % J->loss_cap) * NM_LOSS_STRIDE;
    if (split) {                                  // the other parts logged their ll_m before they arrived
      ll_sum = 0.f;
      for (int m = 0; m < M; ++m) ll_sum += row[NM_LOSS_LL_M + m];
    }
    const float llw_tot = (J->w_off >= 0) ? 1.0f : J->ll_weight;        // (weighted per modality already)
    row[NM_LOSS_KL] = kl_w * kl;
    row[NM_LOSS_LL] = ll_sum;
    row[NM_LOSS_TC] = tc;
    row[NM_LOSS_TOTAL] = kl_w * kl - llw_tot * ll_sum + J->tc_weight * tc;
  }
  if (!bwd) return;
  if (split && part >= Me) return;                // a decoder-only part has no encoder to differentiate
  prof(c, PH_ALPHA);
  // d z = sum over the decoders, in decoder order (single workgroup: accumulated in place in that order)
  // d z of shared column z = sum over the decoders, in decoder order
  // (every copy is requested before the first is used: a loop over M would wait for each in turn)
  auto load_dz = [&](int r, int z) {
    float dq[NM_MAX_MOD];
#pragma unroll
    for (int q = 0; q < NM_MAX_MOD; ++q) dq[q] = (q < M) ? ws_dz0[(int64_t)q * ROWS * Zs + r * Zs + z] : 0.f;
    float d = dq[0];
#pragma unroll
    for (int q = 1; q < NM_MAX_MOD; ++q) d += dq[q];
    return d;
  };
  auto load_dz4 = [&](int r, int z0) {
    f32x4 dq[NM_MAX_MOD];
#pragma unroll
    for (int q = 0; q < NM_MAX_MOD; ++q) {
      dq[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (q < M) dq[q] = *(const GAS f32x4*)(ws_dz0 + (int64_t)q * ROWS * Zs + r * Zs + z0);
    }
    f32x4 d = dq[0];
#pragma unroll
    for (int q = 1; q < NM_MAX_MOD; ++q) d += dq[q];
    return d;
  };

  // ================= fusion backward: alpha gradients (gPoE) =================
  const bool fused = !(Me == 1 && J->single_bypass);
  const float klw = kl_w * c.inv_b;
  const float tcw = J->tc_weight / (float)Me;     // (the softmax over the rows is normalised: no 1 / B)
  gbf16 ws_fz = (gbf16)ws_zc0;                    // the (dead) z|c slot, legacy [256][PW] layout
  // With several experts the fusion backward (8 exponentials per element) is evaluated ONCE: the deltas of every
  // expert go side by side into Q (expert m in columns [m 2Zs, (m+1) 2Zs) = [d mu_m | d logvar_m]), from there into
  // the (dead) z|c slot of the workspace, and each encoder's backward below starts from a 16-byte copy of its
  // columns.  Falls back to one evaluation per encoder when the deltas do not fit in 128 columns.
  const bool once = fused && Me >= 2 && Me * 2 * Zs <= PW && !split && S == 0;
  // Wider latents (config 5: Z = 64, three experts): still ONE evaluation, every expert's [d mu | d logvar] block
  // written straight to its own (dead) z|c slot of the workspace, [256][PW] like the hand-off above.
  const bool once_ws = fused && Me >= 2 && !once && 2 * Zs <= PW && !split && S == 0;
  // (split: the alpha sums ride on the part's own evaluation of the fusion backward, below)
  if (once || once_ws || (fused && J->combine == NM_COMBINE_GPOE && !split)) {
    relaunder(c);
    float dal[NM_MAX_EXP] = {0.f, 0.f, 0.f, 0.f};
    if (once) {                                    // zero pads of every expert's block: columns [Z, Zs) of both halves
      const int npz = Zs - Z, cols = Me * 2 * npz;
      const float rc_ = cols > 0 ? 1.0f / (float)cols : 0.f;
      for (int e = c.tid; e < ROWS * cols; e += WG) {
        const int r = idiv(e, cols, rc_), j = e - r * cols;
        const int blk = idiv(j, npz, 1.0f / (float)npz), k = j - blk * npz;      // blk = 2 m + half
        c.Q[r * LDP + blk * Zs + Z + k] = (__bf16)0.0f;
      }
    }
    if (once_ws && Zs > Z) {                       // the same pads, in the experts' workspace blocks
      const int npz = Zs - Z, cols = Me * 2 * npz;
      const float rc_ = 1.0f / (float)cols;
      for (int e = c.tid; e < ROWS * cols; e += WG) {
        const int r = idiv(e, cols, rc_), j = e - r * cols;
        const int blk = idiv(j, npz, 1.0f / (float)npz), k = j - blk * npz;      // blk = 2 m + half
        ((gbf16)(ws_zc0 + (int64_t)(blk >> 1) * wl.act))[r * PW + (blk & 1) * Zs + Z + k] = (__bf16)0.0f;
      }
    }
    if (vec4) {
      // four latent columns of a row per iteration, as in the forward pass
      const int nq4 = (Z + 3) >> 2;
      const float rq4 = 1.0f / (float)nq4;
      for (int e = c.tid; e < ROWS * nq4; e += WG) {
        const int r = idiv(e, nq4, rq4), z0 = 4 * (e - r * nq4);
        f32x4 mu4[NM_MAX_EXP], lv4[NM_MAX_EXP];
#pragma unroll
        for (int m = 0; m < NM_MAX_EXP; ++m) {
          mu4[m] = f32x4{0.f, 0.f, 0.f, 0.f};
          lv4[m] = mu4[m];
          if (m < Me) {
            mu4[m] = *(const GAS f32x4*)(ws_mu_m + ((int64_t)m * ROWS + r) * Zs + z0);
            lv4[m] = *(const GAS f32x4*)(ws_lv_m + ((int64_t)m * ROWS + r) * Zs + z0);
          }
        }
        const f32x4 mj4 = *(const GAS f32x4*)(ws_mu_j + r * Zs + z0), lj4 = *(const GAS f32x4*)(ws_lv_j + r * Zs + z0);
        const f32x4 es4 = *(const GAS f32x4*)(ws_es + r * Zs + z0);
        f32x4 dz4 = load_dz4(r, z0);
        if (J->dz_extra) {
#pragma unroll
          for (int i = 0; i < 4; ++i) dz4[i] += asg(J->dz_extra)[(int64_t)(c.row0 + r) * Z + min(z0 + i, Z - 1)];
        }
        bf16x4 pmu[NM_MAX_EXP], plv[NM_MAX_EXP];
        const bool rv = r < c.nrows;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          Lat Lt;
#pragma unroll
          for (int m = 0; m < NM_MAX_EXP; ++m) { Lt.mu[m] = mu4[m][i]; Lt.lv[m] = lv4[m][i]; }
          const float dmu_j = dz4[i] + klw * mj4[i];
          const float dlv_j = 0.5f * dz4[i] * es4[i] + klw * 0.5f * (expf(lj4[i]) - 1.0f);
          FuseGrad G = fuse_bwd(J, Lt, al, dmu_j, dlv_j);
          const bool ok = rv && z0 + i < Z;
#pragma unroll
          for (int m = 0; m < NM_MAX_EXP; ++m) {
            dal[m] += ok ? G.dal[m] : 0.f;
            if ((once || once_ws) && m < Me) {
              // d (tc_weight tc) / d mu_m[r][z] = -(tc_weight / Me) softmax over the rows
              if (tcw != 0.f) G.dmu[m] -= tcw * expf(Lt.mu[m] - c.lse[m * Z + min(z0 + i, Z - 1)]);
              pmu[m][i] = (__bf16)(ok ? G.dmu[m] : 0.f);
              plv[m][i] = (__bf16)(ok ? G.dlv[m] : 0.f);
            }
          }
        }
#pragma unroll
        for (int m = 0; m < NM_MAX_EXP; ++m) {
          if (once && m < Me) {
            *reinterpret_cast<bf16x4*>(c.Q + r * LDP + m * 2 * Zs + z0) = pmu[m];
            *reinterpret_cast<bf16x4*>(c.Q + r * LDP + m * 2 * Zs + Zs + z0) = plv[m];
          }
          if (once_ws && m < Me) {
            gbf16 blk = (gbf16)(ws_zc0 + (int64_t)m * wl.act);
            *(GAS bf16x4*)(blk + r * PW + z0) = pmu[m];
            *(GAS bf16x4*)(blk + r * PW + Zs + z0) = plv[m];
          }
        }
      }
    } else {
      for (int e = c.tid; e < ROWS * Zc; e += WG) {
        int r = idiv(e, Zc, rZc), z = e - r * Zc;
        Lat Lt;
        load_lat(Lt, r, S + z);
        float mj = ws_mu_j[r * Zs + z], lj = ws_lv_j[r * Zs + z], es = ws_es[r * Zs + z], dz = load_dz(r, z);
        if (J->dz_extra) dz += asg(J->dz_extra)[(int64_t)(c.row0 + r) * Z + z];
        float dmu_j = dz + klw * mj;
        float dlv_j = 0.5f * dz * es + klw * 0.5f * (expf(lj) - 1.0f);
        FuseGrad G = fuse_bwd(J, Lt, al, dmu_j, dlv_j);
        const bool rv = r < c.nrows;
  #pragma unroll
        for (int m = 0; m < NM_MAX_EXP; ++m) {
          dal[m] += rv ? G.dal[m] : 0.f;
          if (once && m < Me) {
            // d (tc_weight tc) / d mu_m[r][z] = -(tc_weight / Me) softmax over the rows
            if (tcw != 0.f) G.dmu[m] -= tcw * expf(Lt.mu[m] - c.lse[m * Z + z]);
            c.Q[r * LDP + m * 2 * Zs + z] = (__bf16)(rv ? G.dmu[m] : 0.f);
            c.Q[r * LDP + m * 2 * Zs + Zs + z] = (__bf16)(rv ? G.dlv[m] : 0.f);
          }
          if (once_ws && m < Me) {
            if (tcw != 0.f) G.dmu[m] -= tcw * expf(Lt.mu[m] - c.lse[m * Z + z]);
            gbf16 blk = (gbf16)(ws_zc0 + (int64_t)m * wl.act);
            blk[r * PW + z] = (__bf16)(rv ? G.dmu[m] : 0.f);
            blk[r * PW + Zs + z] = (__bf16)(rv ? G.dlv[m] : 0.f);
          }
        }
      }
    }
    if (J->combine == NM_COMBINE_GPOE) {
      float tot[NM_MAX_EXP];
#pragma unroll
      for (int m = 0; m < NM_MAX_EXP; ++m) tot[m] = block_sum(c, dal[m]);
      if (c.tid == 0) {
        float dot = 0.f;
#pragma unroll
        for (int m = 0; m < NM_MAX_EXP; ++m) dot += al[m] * tot[m];
#pragma unroll
        for (int m = 0; m < NM_MAX_EXP; ++m)
          if (m < Me && (!split || m == part)) apply_grad(c, J->mod[m].alpha, al[m] * (tot[m] - dot), nullptr);   // softmax backward
      }
    }
    lds_barrier();
    if (once) store_act(c, ws_fz, c.Q, Me * 2 * Zs);
    handoff_barrier();
  }
  tr(c, 12);

  // ================= encoders, backward =================
  float* const spatch = reinterpret_cast<float*>(Sb + SPATCH_OFF);
  for (int m = 0; m < Me; ++m) {
    if (split && m != part) continue;
    relaunder(c);
    const nm_modality_t& md = J->mod[m];
    const int Hh = J->H[L - 1];
    const int whp = rup(2 * Zs, 32);
    // heads image -> lower half of Q, upper rows of the last hidden activation -> upper half: in flight while
    // P <- [d mu_m | 0 | d logvar_m | 0] is put together
    const GAS char* act_last = ws_enc + (int64_t)(m * L + (L - 1)) * ACT_BYTES;
    dma_img(c, wsh + md.heads_s, Qb, 2 * Zs, blob_kp(Hh), (const GAS char*)J->wsh);
    dma_act(c, act_last, Qb + IMG_BYTES, ROWS / 2, ROWS / 2, act_segs(Hh));
    if (once || once_ws) {                         // this expert's columns of the saved fusion backward
      const int segs = (2 * Zs) >> 3;              // 16-byte pieces per row
      const float rs_ = 1.0f / (float)segs;
      gcbf16 src = once ? (gcbf16)(ws_fz + m * 2 * Zs) : (gcbf16)(ws_zc0 + (int64_t)m * wl.act);
      for (int p_ = c.tid; p_ < ROWS * segs; p_ += WG) {
        const int row = idiv(p_, segs, rs_), seg = p_ - row * segs;
        *reinterpret_cast<u32x4*>(c.P + row * LDP + seg * 8) = *(const GAS u32x4*)(src + row * PW + seg * 8);
      }
    } else {
      float dal_s[NM_MAX_EXP] = {0.f, 0.f, 0.f, 0.f};
      const int npad = whp - 2 * Z;
      const float rnp = npad > 0 ? 1.0f / (float)npad : 0.f;
      for (int e = c.tid; e < ROWS * npad; e += WG) {
        int r = idiv(e, npad, rnp), j = e - r * npad;
        int k = (j < Zs - Z) ? Z + j : Zs + Z + (j - (Zs - Z));
        c.P[r * LDP + k] = (__bf16)0.0f;
      }
      // private columns (DMVAE family): d mu_m[:, i] = d z of THIS decoder's private input column; logvar unused
      for (int e = c.tid; e < ROWS * S; e += WG) {
        const int r = e / S, i = e - r * S;
        const float d = ws_dz0[(int64_t)m * ROWS * Zs + r * Zs + Zc + i];
        c.P[r * LDP + i] = (__bf16)(r < c.nrows ? d : 0.f);
        c.P[r * LDP + Zs + i] = (__bf16)0.0f;
      }
      if (vec4) {
        // four latent columns of a row per iteration (as in the forward pass); this expert's deltas only
        const int nq4 = (Z + 3) >> 2;
        const float rq4 = 1.0f / (float)nq4;
        for (int e = c.tid; e < ROWS * nq4; e += WG) {
          const int r = idiv(e, nq4, rq4), z0 = 4 * (e - r * nq4);
          f32x4 mu4[NM_MAX_EXP], lv4[NM_MAX_EXP];
#pragma unroll
          for (int q = 0; q < NM_MAX_EXP; ++q) {
            mu4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            lv4[q] = mu4[q];
            if (q < Me) {
              mu4[q] = *(const GAS f32x4*)(ws_mu_m + ((int64_t)q * ROWS + r) * Zs + z0);
              lv4[q] = *(const GAS f32x4*)(ws_lv_m + ((int64_t)q * ROWS + r) * Zs + z0);
            }
          }
          const f32x4 mj4 = *(const GAS f32x4*)(ws_mu_j + r * Zs + z0), lj4 = *(const GAS f32x4*)(ws_lv_j + r * Zs + z0);
          const f32x4 es4 = *(const GAS f32x4*)(ws_es + r * Zs + z0);
          f32x4 dz4 = load_dz4(r, z0);
          if (J->dz_extra) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dz4[i] += asg(J->dz_extra)[(int64_t)(c.row0 + r) * Z + min(z0 + i, Z - 1)];
          }
          bf16x4 pmu, plv;
          const bool rv = r < c.nrows;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            Lat Lt;
#pragma unroll
            for (int q = 0; q < NM_MAX_EXP; ++q) { Lt.mu[q] = mu4[q][i]; Lt.lv[q] = lv4[q][i]; }
            const float dmu_j = dz4[i] + klw * mj4[i];
            const float dlv_j = 0.5f * dz4[i] * es4[i] + klw * 0.5f * (expf(lj4[i]) - 1.0f);
            FuseGrad G = fuse_bwd(J, Lt, al, dmu_j, dlv_j);
            const bool ok = rv && z0 + i < Z;
            float dmu_m = pick(G.dmu, m);
            if (tcw != 0.f) dmu_m -= tcw * expf(pick(Lt.mu, m) - c.lse[m * Z + min(z0 + i, Z - 1)]);
            pmu[i] = (__bf16)(ok ? dmu_m : 0.f);
            plv[i] = (__bf16)(ok ? pick(G.dlv, m) : 0.f);
#pragma unroll
            for (int q = 0; q < NM_MAX_EXP; ++q) dal_s[q] += ok ? G.dal[q] : 0.f;
          }
          *reinterpret_cast<bf16x4*>(c.P + r * LDP + z0) = pmu;
          *reinterpret_cast<bf16x4*>(c.P + r * LDP + Zs + z0) = plv;
        }
      } else {
#pragma unroll 2
        for (int e = c.tid; e < ROWS * Zc; e += WG) {
          int r = idiv(e, Zc, rZc), z = e - r * Zc;
          Lat Lt;
          load_lat(Lt, r, S + z);
          float mj = ws_mu_j[r * Zs + z], lj = ws_lv_j[r * Zs + z], es = ws_es[r * Zs + z], dz = load_dz(r, z);
          if (J->dz_extra) dz += asg(J->dz_extra)[(int64_t)(c.row0 + r) * Z + z];
          float dmu_j = dz + klw * mj;
          float dlv_j = 0.5f * dz * es + klw * 0.5f * (expf(lj) - 1.0f);
          FuseGrad G = fuse_bwd(J, Lt, al, dmu_j, dlv_j);
          const bool rv = r < c.nrows;
          float dmu_m = pick(G.dmu, m);
          if (tcw != 0.f) dmu_m -= tcw * expf(pick(Lt.mu, m) - c.lse[m * Z + z]);
          c.P[r * LDP + S + z] = (__bf16)(rv ? dmu_m : 0.f);
          c.P[r * LDP + Zs + S + z] = (__bf16)(rv ? pick(G.dlv, m) : 0.f);
#pragma unroll
          for (int q = 0; q < NM_MAX_EXP; ++q) dal_s[q] += rv ? G.dal[q] : 0.f;
        }
      }
      if (split && fused && J->combine == NM_COMBINE_GPOE) {       // this part's alpha: softmax backward of the sums
        float tot[NM_MAX_EXP];
#pragma unroll
        for (int q = 0; q < NM_MAX_EXP; ++q) tot[q] = block_sum(c, dal_s[q]);
        if (c.tid == 0) {
          float dot = 0.f;
#pragma unroll
          for (int q = 0; q < NM_MAX_EXP; ++q) dot += al[q] * tot[q];
          apply_grad(c, md.alpha, pick(al, m) * (pick(tot, m) - dot), nullptr);
        }
      }
    }
    wait_vm(0);
    lds_barrier();
    prof(c, PH_ENCB_PREP);
    // dgrad through both heads from one image (rows = [d mu | d logvar] columns of P), then Q <- activation
    f32x4 acc[2][RT];
    zero_acc(acc);
    dgrad_tile(c, acc, c.P, LDP, 0, c.Q, LDP, (2 * Zs) / 32);
    lds_barrier();                                  // image fully read
    prof(c, PH_ENCB_HEADS_DGRAD);
    dma_act(c, act_last, Qb, 0, ROWS / 2, act_segs(Hh));   // (waited for inside the first weight-gradient pass)
    tr(c, 13);
    {
      GAS char* img = wsh + md.heads_s;
      const int hkp = blob_kp(Hh);
      GAS float* const hvec = (GAS float*)(img + cimg_bytes(2 * Zs, Hh));
      const WgGeom Gm{Z, Hh, 0, rup(Hh + 1, 16), WgT{md.mu_w, md.mu_b, img, hkp * 2, hvec, spatch}};
      wgrad_adam<SCALAR_TR>(c, c.P, LDP, 0, c.Q, LDP, Gm, 0);
      const WgGeom Gl{Z, Hh, 0, rup(Hh + 1, 16),
                      WgT{md.lv_w, md.lv_b, img + (int64_t)Zs * hkp * 2, hkp * 2, hvec + Zs, spatch}};
      // (pending = 0: its barrier also separates this pass's bias hand-off through LDS from the previous pass's)
      wgrad_adam<SCALAR_TR>(c, c.P, LDP, Zs, c.Q, LDP, Gl, 0);
    }
    prof(c, PH_ENCB_HEADS_WGRAD);
    finish_delta(c, acc, c.Q, Hh, nl);              // P = delta of h_{L-1}
    lds_barrier();
    prof(c, PH_ENCB_DELTA);
    for (int e = L - 1; e >= 1; --e) {
      int Kin = J->H[e - 1], Nout = J->H[e];
      zero_acc(acc);
      GAS char* const eimg = wsh + md.enc_s[e];
      const WgGeom Ge{Nout, Kin, 0, rup(Kin + 1, 16),
                      WgT{md.enc_w[e], md.enc_b[e], eimg, blob_kp(Kin) * 2, (GAS float*)(eimg + cimg_bytes(Nout, Kin)), spatch}};
      dgrad_hidden(c, acc, eimg, ws_enc + (int64_t)(m * L + (e - 1)) * ACT_BYTES, Nout, Kin);
      prof(c, PH_ENCB_DGRAD);
      wgrad_adam<SCALAR_TR>(c, c.P, LDP, 0, c.Q, LDP, Ge, 0);
      prof(c, PH_ENCB_WGRAD);
      finish_delta(c, acc, c.Q, Kin, nl);
      lds_barrier();
      prof(c, PH_ENCB_DELTA);
    }
    tr(c, 14);
    // first encoder layer: dW[n][k] = sum_r P[r][n] xc[r][k]; the x chunk images stream through two slots of Q
    // (the second one runs 4 KiB into S, below the patches): chunk kc + 1 lands while chunk kc is processed
    {
      const int Kx = md.Kx, K0 = md.D + C, N0 = J->H[0];
      const int nch = (Kx + XCH - 1) / XCH;
      const GAS char* xsrc = (const GAS char*)asg(md.xb) + (int64_t)(c.row0 / ROWS) * nch * XIMG_BYTES;
      GAS char* img = wsh + md.enc_s[0];
      dma_lin<0>(c, xsrc, Qb, XIMG_BYTES >> 10);
      auto geom = [&](int kc) {
        return WgGeom{N0, K0, kc * XCH, min(XCH, Kx - kc * XCH),
                      WgT{md.enc_w[0], md.enc_b[0], img + (int64_t)kc * XCH * 2, Kx * 2,
                          (GAS float*)(img + l0_img_bytes(N0, Kx)), spatch}};
      };
      // chunk kc + 1 is requested as soon as every wave is done with chunk kc - 1 (the barrier that ends pass kc - 1) and
      // lands during pass kc; pass kc waits for chunk kc itself, after its own first requests (pending = what this
      // wave issued after that copy: the next chunk's pieces)
      for (int kc = 0; kc < nch; ++kc) {
        int n_next = 0;
        if (kc + 1 < nch) n_next = dma_lin<0>(c, xsrc + (int64_t)(kc + 1) * XIMG_BYTES, Qb + ((kc + 1) & 1) * XIMG_BYTES, XIMG_BYTES >> 10);
        const __bf16* Xc = reinterpret_cast<const __bf16*>(Qb + (kc & 1) * XIMG_BYTES);
        wgrad_adam<SCALAR_TR>(c, c.P, LDP, 0, Xc, LDX, geom(kc), n_next);
      }
      prof(c, PH_ENCB_L0_WGRAD);
    }
    tr(c, 15);
  }
}

// ----------------------------------------------------------------------------------------------
constexpr int SMEM_BYTES = 2 * ROWS * LDP * 2 + STAGE_FLOATS * 4 + 2 * VEC_BYTES + (64 + 128 + 256 + 256 + 128 + 16 + 4) * 4;
static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
static_assert(XIMG_BYTES + OBLOB_BYTES + NWAVES * PATCH_FLOATS * 4 <= ACT_BYTES, "output-chunk layout of Q");
static_assert(2 * XIMG_BYTES - ACT_BYTES <= SPATCH_OFF && SPATCH_OFF + NWAVES * PATCH_FLOATS * 4 <= STAGE_FLOATS * 4, "S layout");

__device__ __forceinline__ void carve_lds(Ctx& c, unsigned char* smem) {
  c.wave_s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  c.P = reinterpret_cast<__bf16*>(smem);
  c.Q = c.P + ROWS * LDP;
  c.stage = reinterpret_cast<float*>(c.Q + ROWS * LDP);
  c.vec = c.stage + STAGE_FLOATS;
  c.red = c.vec + 2 * (VEC_BYTES / 4);
  c.colacc = c.red + 64;
  c.rowacc = c.colacc + 128;
  c.lse = c.rowacc + 256;
  c.bgrad = c.lse + 256;
  c.tlast = reinterpret_cast<unsigned long long*>(c.bgrad + 128);
  c.abort = reinterpret_cast<unsigned*>(c.tlast + 8);
}

template <bool SCALAR_TR, int MODE = 0>
__global__ __launch_bounds__(WG) void nm_step_kernel(const nm_job_t* __restrict__ jobs, int step0, int steps_per_tile,
                                                     int flags, int n_jobs, int nparts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int job_idx = blockIdx.x, part = -1;
  if ((flags & 64) && blockIdx.y == 0 && blockIdx.x < 512 && threadIdx.x == 0) nm_wg_times[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
  if (flags & NM_F_SPLIT) {
    // workgroups b and b + 8 share an XCD (observed placement; speed only): the parts of a job are consecutive
    // workgroups of ONE XCD, so that their hand-offs and shared expert statistics stay inside one L2
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    job_idx = (idx / nparts) * 8 + xcd;
    part = idx % nparts;
    if (job_idx >= n_jobs) return;
    if ((flags & NM_F_FAULT_INJECT) && part == 1) return;    // diagnostic: a part that never arrives (time-out test)
  }
  const int tile_idx = blockIdx.y;
  const nm_job_t* J = jobs + job_idx;
  Ctx c;
  c.job = J;
  c.part = part;
  c.nparts = nparts;
  c.slope = J->act_slope;
  carve_lds(c, smem);
  relaunder(c);
  c.flags = flags;
  c.t_last = 0;
  c.ws = (GAS char*)J->workspace + (int64_t)tile_idx * J->workspace_stride;
  // zero LDS once: padded columns are multiplied by zero weights and must stay finite
  for (int i = c.tid; i < SMEM_BYTES / 4; i += WG) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  __syncthreads();
  // De-phase the workgroups of a launch: identical models otherwise run their HBM-heavy weight-gradient / Adam phases
  // in lockstep and share the DRAM 256 ways at once.  nm_job_t.dephase = this job's start offset in microseconds,
  // waited for on the constant-rate counter (s_sleep counts are not shader cycles: the first version of this, a fixed
  // number of s_sleep(127), spread the workgroups over three steps instead of one -- tools/wg_spread.py).  The offset
  // is pure cost at the end of the launch, so short launches get less of it and very short ones none.
  {
#ifndef NM_FWD_DEPHASE_DIV
#define NM_FWD_DEPHASE_DIV 4
#endif
    // (forward only, many row tiles: the first workgroup of every CU -- tile 0 of each job -- starts late by a fraction
    //  of a tile's time, the following tiles inherit the stagger)
    int us = steps_per_tile >= 64 ? J->dephase : (steps_per_tile >= 8 ? (J->dephase >> 2) : 0);
    if (MODE == 3) us = (NM_FWD_DEPHASE_DIV > 0 && tile_idx == 0 && gridDim.y > 1) ? J->dephase / NM_FWD_DEPHASE_DIV : 0;
    if (us > 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), ticks = 100ull * (unsigned long long)min(us, 20000);
      while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    }
  }
  const int nb = (J->n_rows + ROWS - 1) / ROWS;
  const int s_begin = step0 + tile_idx * steps_per_tile;
  for (int s = s_begin; s < s_begin + steps_per_tile; ++s) {
    int b = s % nb;
    c.lstep = s - s_begin;
    c.row0 = b * ROWS;
    c.nrows = min(ROWS, J->n_rows - c.row0);
    c.inv_b = 1.0f / (float)c.nrows;
    // bias corrections in double, as torch.optim.Adam computes them on the host
    const int64_t t_opt = J->adam_off + (int64_t)s + 1;
    const double tt = (double)t_opt;
    // learning rate of this optimizer step: the schedule table (param_group['lr'] = clr per step) or the constant
    const double lr_t = (J->lr_table && J->lr_cap > 0) ? J->lr_table[(t_opt - 1) % J->lr_cap] : (double)J->lr;
    c.step_size = (float)(lr_t / (1.0 - pow((double)J->beta1, tt)));
    c.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)J->beta2, tt)));
    if (flags & NM_F_PROFILE) c.t_last = clock64();
    if (flags & 64) c.tlast[c.wave_s] = clock64();
    lds_barrier();
    relaunder(c);
    run_step<SCALAR_TR, MODE>(c, s);
    if ((flags & NM_F_SPLIT) && *c.abort != 0u) break;       // a hand-off timed out (wave-uniform: LDS word read by all)
    tr(c, 62);
    // the next step reads what this one stored (weights, shadow images, workspace): drain, then meet
    handoff_barrier();
    tr(c, 63);
  }
  if ((flags & 64) && blockIdx.y == 0 && blockIdx.x < 512 && c.tid == 0) nm_wg_times[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
}

#include "nm_wide.inc"

// bytes of the trunk's part of a tile's workspace (the heads' own region sits behind it)
__host__ __device__ inline int64_t trunk_ws_bytes(const nm_job_t* J) {
  return J->wide ? wide_ws_layout(J).total : ws_layout(J->M, J->L, J->Z).total;
}

// ---- regression head (cVAE.py:2249-2253 regressor, 2318-2321 forward, 2330-2346 loss) ----------------
// fi_pred = W3 relu(W2 relu(W1 cat_m(x_m - x_hat_m) + b1) + b2) + b3;  loss = mean_r (fi_pred - FI)^2.
// One workgroup per (job, 256-row tile); same operand conventions as the trunk: bf16 MFMA operands, fp32 accumulate,
// fp32 parameters.  The first layer is the big one (128 x sum D) and runs like the trunk's first encoder layer: the
// trunk's output chunks leave the residual as bf16 chunk images [256][72] (nm_job_t.reg_resid, one per 64 ROI columns
// of a modality), W1 lives in the master with every modality's columns padded to whole chunks and has bf16 chunk
// images [128][72] in the shadow (nm_job_t.reg_s); forward = both streamed through LDS by LDS-DMA, two stages;
// backward per chunk = weight gradient + Adam in wave-independent units from the residual image, then
// d loss / d x_hat = -(delta h1 W1[:, chunk]) from the (pre-update) weight image, written as a bf16 chunk image
// (nm_job_t.reg_dres) that the trunk's second pass adds to its NLL gradient.
// P[r][f] = relu(acc) for f < N (N a multiple of 16, <= 128)
__device__ __forceinline__ void relu_to_P(const Ctx& c, const f32x4 (&acc)[2][RT], int N) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int f0 = (c.wn + 4 * t) * 16 + 4 * c.g;
    if (f0 >= N) continue;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int r = c.wm * WROWS + rt * 16 + c.c16;
      bf16x4 pk;
#pragma unroll
      for (int i = 0; i < 4; ++i) pk[i] = (__bf16)fmaxf(acc[t][rt][i], 0.f);
      *reinterpret_cast<bf16x4*>(c.P + r * LDP + f0) = pk;
    }
  }
}

// Set-up shared by the stand-alone head kernels: LDS carved and zeroed, the tile's rows, Adam scalars of `step`.
__device__ __forceinline__ bool head_setup(Ctx& c, unsigned char* smem, const nm_job_t* J, int step, int tile0, int flags) {
  c.job = J;
  carve_lds(c, smem);
  relaunder(c);
  c.flags = flags;
  c.part = -1; c.nparts = 1; c.lstep = 0; c.slope = J->act_slope;
  c.t_last = 0;
  c.ws = (GAS char*)J->workspace + (int64_t)blockIdx.y * J->workspace_stride;
  for (int i = c.tid; i < SMEM_BYTES / 4; i += WG) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  __syncthreads();
  c.row0 = (tile0 + (int)blockIdx.y) * ROWS;
  c.nrows = min(ROWS, J->n_rows - c.row0);
  if (c.nrows <= 0) return false;
  c.inv_b = 1.0f / (float)c.nrows;
  const int64_t t_opt = J->adam_off + (int64_t)step + 1;
  const double tt = (double)t_opt;
  const double lr_t = (J->lr_table && J->lr_cap > 0) ? J->lr_table[(t_opt - 1) % J->lr_cap] : (double)J->lr;
  c.step_size = (float)(lr_t / (1.0 - pow((double)J->beta1, tt)));
  c.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)J->beta2, tt)));
  return true;
}

// The head of one 256-row tile; `c` is fully set up (rows, Adam scalars, c.flags = NM_F_BACKWARD / ADAM / GRADS).
// `hws` = the head's own workspace (behind the trunk's: the trunk's saved activations stay intact); `log` = this
// tile writes the loss row.  Leaves LDS in an arbitrary (finite) state.
__device__ __forceinline__ void reg_head_body(Ctx& c, const nm_job_t* J, int step, GAS char* hws, bool log) {
  const int flags = c.flags;
  const bool bwd = (flags & NM_F_BACKWARD) != 0;
  const int M = experts(J);
  constexpr int N1 = 128, N2 = 64;
  const int nq = head_chunk0(J, M), Kh = nq * XCH;           // chunks / columns of the padded concatenation
  gcf32 prm = asg(J->params);
  gcf32 W2 = prm + J->reg_w[1], b2 = prm + J->reg_b[1], W3 = prm + J->reg_w[2];
  GAS char* const wimg = (GAS char*)J->wsh + J->reg_s;       // W1 chunk images, then b1
  const int64_t tile_off = (int64_t)(c.row0 / ROWS) * nq * XIMG_BYTES;
  const GAS char* const res = (const GAS char*)asg(J->reg_resid) + tile_off;
  GAS char* const dres = (GAS char*)asg(J->reg_dres);        // (one set of images: the batch in flight)
  gbf16 ws_h1 = (gbf16)hws;                                  // h1 as an activation image (backward)
  char* const Qb = reinterpret_cast<char*>(c.Q);
  char* const Pb = reinterpret_cast<char*>(c.P);
  const float slope_keep = c.slope;
  c.slope = 0.f;                                             // ReLU

  // ---- layer 1: h1 = relu(W1 resid + b1) ----
  lds_barrier();                                             // P / Q are drained by whatever ran before
  fwd_first_layer(c, res, Kh, wimg, no_next(), N1, true, bwd ? ws_h1 : (gbf16)nullptr);
  c.slope = slope_keep;
  tr(c, 21);
  // ---- layer 2: h2 = relu(W2 h1 + b2), in place ----
  relaunder(c);
  f32x4 acc[2][RT];
  bias_acc(c, acc, b2, N2, 0);
  for (int ks = 0; ks < N1 / 32; ++ks) {
    bf16x8 wf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) wf[t] = w_frag(W2, N2, N1, (c.wn + 4 * t) * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      bf16x8 a = lds_frag(c.P, LDP, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t][rt] = mfma(wf[t], a, acc[t][rt]);
    }
  }
  lds_barrier();
  relu_to_P(c, acc, N2);
  lds_barrier();
  // ---- layer 3 + MSE: one row per thread ----
  relaunder(c);
  float pred = 0.f, err = 0.f;
  if (c.tid < ROWS) {
    float s = prm[J->reg_b[2]];
    for (int n = 0; n < N2; ++n) s = fmaf((float)c.P[c.tid * LDP + n], (float)(__bf16)W3[wt_off(0, n, N2 / 16)], s);
    pred = s;
    if (c.tid < c.nrows) {
      if (J->out_fi_pred) asg(J->out_fi_pred)[c.row0 + c.tid] = pred;
      if (J->fi_target) err = pred - asg(J->fi_target)[c.row0 + c.tid];
    }
  }
  const float sse = block_sum(c, err * err);
  if (c.tid == 0 && J->loss_log && J->fi_target && log)   // one tile's MSE (training: the step's batch)
    asg(J->loss_log)[(int64_t)(step % J->loss_cap) * NM_LOSS_STRIDE + NM_LOSS_REG] = sse * c.inv_b;
  tr(c, 22);
  if (!bwd) return;

  // ---- backward ----
  // d pred (rowacc), dW3 / db3 from h2, then P <- delta h2 = d pred * W3 * relu'(h2) in place
  if (c.tid < ROWS) c.rowacc[c.tid] = J->reg_lambda * 2.0f * err * c.inv_b;     // err = 0 on padded rows
  __syncthreads();
  float g3 = 0.f;
  if (c.tid < N2) {
    for (int r = 0; r < ROWS; ++r) g3 = fmaf(c.rowacc[r], (float)c.P[r * LDP + c.tid], g3);
  } else if (c.tid == N2) {
    for (int r = 0; r < ROWS; ++r) g3 += c.rowacc[r];
  }
  __syncthreads();
  for (int e = c.tid; e < ROWS * N2; e += WG) {
    const int r = e >> 6, n = e & 63;
    const float h = (float)c.P[r * LDP + n];
    c.P[r * LDP + n] = (__bf16)(h > 0.f ? c.rowacc[r] * (float)(__bf16)W3[wt_off(0, n, N2 / 16)] : 0.f);
  }
  __syncthreads();                               // W3 fully read before its update
  if (c.tid < N2) apply_grad(c, J->reg_w[2] + wt_off(0, c.tid, N2 / 16), g3);
  else if (c.tid == N2) apply_grad(c, J->reg_b[2], g3);
  if (c.tid < N2) {                              // db2 = column sums of delta h2
    float g = 0.f;
    for (int r = 0; r < ROWS; ++r) g += (float)c.P[r * LDP + c.tid];
    apply_grad(c, J->reg_b[1] + c.tid, g);
  }
  // layer 2 backward: Q <- h1; delta h1 (pre-mask) = delta h2 W2; dW2 = delta h2^T h1
  dma_act(c, (const GAS char*)ws_h1, Qb, 0, ROWS, act_segs(N1));
  wait_vm(0);
  lds_barrier();
  zero_acc(acc);
  dgrad_acc(c, acc, c.P, W2, N2, N1, N2 / 32, 0);
  lds_barrier();                                 // W2 fully read before its update
  float* const hpatch = c.stage + SPATCH_OFF / 4;
  wgrad_adam<false>(c, c.P, LDP, 0, c.Q, LDP, WgGeom{N2, N1, 0, N1, WgT{J->reg_w[1], -1, nullptr, 0, nullptr, hpatch}});
  relaunder(c);
#pragma unroll
  for (int t = 0; t < 2; ++t) {                  // P <- delta h1 = acc * relu'(h1)
    const int k0 = (c.wn + 4 * t) * 16 + 4 * c.g;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int r = c.wm * WROWS + rt * 16 + c.c16;
      bf16x4 a = *reinterpret_cast<const bf16x4*>(c.Q + r * LDP + k0);
      bf16x4 pk;
#pragma unroll
      for (int i = 0; i < 4; ++i) pk[i] = (__bf16)(((float)a[i] > 0.f) ? acc[t][rt][i] : 0.f);
      *reinterpret_cast<bf16x4*>(c.P + r * LDP + k0) = pk;
    }
  }
  lds_barrier();
  tr(c, 23);
  // ---- layer 1 backward ----
  // LDS: P = delta h1; Q = [residual chunk [256][72] | W1 chunk slot A [128][72] | slot B] (slot B runs 4 KiB into S,
  // below the patches).  Per chunk: weight gradient + Adam from the residual image (the LDS copy of the chunk's
  // weights was complete before: the dgrad below sees the weights of THIS step), the next residual chunk is requested
  // as soon as every wave is done with the current one, then d loss / d x_hat = -(delta h1 W1[:, chunk]).
  {
    char* const slotR = Qb;
    auto slotW = [&](int q) { return Qb + XIMG_BYTES + (q & 1) * W0IMG_BYTES; };
    dma_lin(c, res, slotR, XIMG_BYTES >> 10);
    dma_lin(c, wimg, slotW(0), W0IMG_BYTES >> 10);
    if (c.tid < N1) {                              // db1 (and its fp32 copy behind the chunk images)
      float g = 0.f;
      for (int r = 0; r < ROWS; ++r) g += (float)c.P[r * LDP + c.tid];
      apply_grad(c, J->reg_b[0] + c.tid, g, (GAS float*)(wimg + (int64_t)nq * W0IMG_BYTES) + c.tid);
    }
    for (int q = 0; q < nq; ++q) {
      relaunder(c);
      wait_vm(0);
      lds_barrier();                               // chunk q's residual and weights have landed everywhere
      if (q + 1 < nq) dma_lin(c, wimg + (int64_t)(q + 1) * W0IMG_BYTES, slotW(q + 1), W0IMG_BYTES >> 10);
      wgrad_adam<false>(c, c.P, LDP, 0, reinterpret_cast<const __bf16*>(slotR), LDX,
                        WgGeom{N1, Kh, q * XCH, XCH, WgT{J->reg_w[0], -1, wimg + (int64_t)q * W0IMG_BYTES, LDX * 2, nullptr, hpatch}});
      tr(c, 24);
      if (q + 1 < nq) dma_lin(c, res + (int64_t)(q + 1) * XIMG_BYTES, slotR, XIMG_BYTES >> 10);
      // d resid chunk: lane (c16, g) of row tile rt holds columns wn * 16 + 4 g .. + 3 of row wm * 128 + rt * 16 + c16
      const __bf16* Wq = reinterpret_cast<const __bf16*>(slotW(q));
      f32x4 dr[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) dr[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < N1 / 32; ++s) {
        bf16x4 l0, h0;
        const unsigned a0 = tr_addr(Wq, LDX, s * 32, c.wn * 16, c.lane);
        NM_TR_READ(l0, a0, 0); NM_TR_READ(h0, a0 + 4u * LDX * 2u, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l0), "+v"(h0));
        const bf16x8 wf = join4(l0, h0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          bf16x8 a = lds_frag(c.P, LDP, c.wm * WROWS + rt * 16 + c.c16, s * 32 + 8 * c.g);
          dr[rt] = mfma(wf, a, dr[rt]);
        }
      }
      const int dl0 = c.wn * 16 + 4 * c.g;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int r = c.wm * WROWS + rt * 16 + c.c16;
        bf16x4 pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) pk[i] = (__bf16)(-dr[rt][i]);                 // d / d x_hat = - d / d resid
        *(GAS bf16x4*)(dres + (int64_t)q * XIMG_BYTES + (r * LDX + dl0) * 2) = pk;
      }
      tr(c, 25);
    }
  }
}
__global__ __launch_bounds__(WG) void nm_reghead_kernel(const nm_job_t* __restrict__ jobs, int step, int tile0, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const nm_job_t* J = jobs + blockIdx.x;
  Ctx c;
  if (!head_setup(c, smem, J, step, tile0, flags & (NM_F_BACKWARD | NM_F_ADAM | NM_F_GRADS))) return;
  reg_head_body(c, J, step, c.ws + trunk_ws_bytes(J), blockIdx.y == 0);
}

// ---- classifier head of the end-to-end model (cVAE.py:2004-2018 Classifier, 2117 logits, 2140-2200 loss) ---
// One workgroup per (job, 256-row tile).  Blocks of Linear - BatchNorm1d - ReLU - Dropout, then Linear to the
// class logits; cross entropy (mean over rows) and the contrastive hinge on the per-subject deviations the
// trunk exported.  Backward returns d CE / d z (dz_out) and, per decoder, the row coefficient of the hinge
// gradient (rowcoef_out), and applies / stores the classifier's own gradients.
// classifier workspace of one tile: the blocks' inputs hin[0 .. NM_MAX_CLS] (bf16 [256][128]), the normalised activations
// xhat[i] (fp32 [256][128]) and 1 / sigma rstd[i] (fp32 [128]) of every BatchNorm -- addresses computed, not tabulated: a
// pointer table indexed by the block number is a private-memory array for this compiler
struct ClsWs {
  GAS char* base;
  __device__ __forceinline__ gbf16 hin(int i) const { return (gbf16)(base + (int64_t)i * ROWS * PW * 2); }
  __device__ __forceinline__ gf32 xhat(int i) const {
    return (gf32)(base + (int64_t)(NM_MAX_CLS + 1) * ROWS * PW * 2 + (int64_t)i * ROWS * PW * 4);
  }
  __device__ __forceinline__ gf32 rstd(int i) const {
    return (gf32)(base + (int64_t)(NM_MAX_CLS + 1) * ROWS * PW * 2 + (int64_t)NM_MAX_CLS * ROWS * PW * 4 + (int64_t)i * PW * 4);
  }
};
__host__ __device__ inline int64_t cls_ws_bytes() {
  return (int64_t)(NM_MAX_CLS + 1) * ROWS * PW * 2 + (int64_t)NM_MAX_CLS * ROWS * PW * 4 + (int64_t)NM_MAX_CLS * PW * 4;
}
// column sums over the rows of per-lane values v[t][i] (feature (wn+4t)*16+4g+i) into dst[feature]
__device__ __forceinline__ void col_reduce(const Ctx& c, const float (&v)[2][4], float* dst, int N) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = v[t][i];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);
      const int f = (c.wn + 4 * t) * 16 + 4 * c.g + i;
      if (c.c16 == 0 && f < N) atomicAdd(&dst[f], s);
    }
}
// Dropout draws: one 64-bit hash per (step, layer, row, group of 4 features) gives the group's four uniforms (16 bits
// each) -- the hash was most of the classifier's forward time when every element had its own.
__device__ __forceinline__ f32x4 uniform4_ctr(uint64_t seed, uint32_t step, uint32_t layer, uint32_t row, uint32_t fgroup) {
  const uint64_t h = splitmix64(seed ^ 0xC1A551F1E5ull ^ ((uint64_t)step << 32) ^ ((uint64_t)layer << 28) ^ ((uint64_t)row << 8) ^ fgroup);
  f32x4 u;
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = (float)(uint32_t)((h >> (16 * i)) & 0xFFFFu) * (1.0f / 65536.0f);
  return u;
}

// (same contract as reg_head_body; `bn_stats`: update the BatchNorm running statistics)
__device__ __forceinline__ void cls_head_body(Ctx& c, const nm_job_t* J, int step, GAS char* hws, bool log, bool bn_stats) {
  const int flags = c.flags;
  const bool train = J->cls_train != 0;
  const bool bwd = (flags & NM_F_BACKWARD) != 0 && train && J->labels != nullptr;
  const int Lc = J->cls_layers, C = J->cls_classes, Z = J->Z;
  const float Bf = (float)c.nrows;
  gcf32 prm = asg(J->params);
  float* col1 = c.colacc;
  float* col2 = c.stage;
  const ClsWs W{hws};
  const float keep_scale = (train && J->cls_dropout > 0.f) ? 1.0f / (1.0f - J->cls_dropout) : 1.0f;

  // ---- P <- z (or the joint mean for predict) ----
  {
    const int Kz = rup(Z, 32);
    gcf32 zsrc = asg((const float*)(J->cls_use_mu ? J->out_mu : J->out_z));
    if ((Z & 3) == 0) {                              // 16-byte pieces (rows are 16-byte aligned then)
      const int nq = Kz >> 2;
      const float rq = 1.0f / (float)nq;
      for (int e = c.tid; e < ROWS * nq; e += WG) {
        const int r = idiv(e, nq, rq), k = 4 * (e - r * nq);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < Z && r < c.nrows) v = *(const GAS f32x4*)(zsrc + (int64_t)(c.row0 + r) * Z + k);
        bf16x4 pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) pk[i] = (__bf16)v[i];
        *reinterpret_cast<bf16x4*>(c.P + r * LDP + k) = pk;
      }
    } else {
      const float rk = 1.0f / (float)Kz;
      for (int e = c.tid; e < ROWS * Kz; e += WG) {
        const int r = idiv(e, Kz, rk), k = e - r * Kz;
        const float v = zsrc[(int64_t)(c.row0 + min(r, c.nrows - 1)) * Z + min(k, Z - 1)];
        c.P[r * LDP + k] = (__bf16)((k < Z && r < c.nrows) ? v : 0.f);
      }
    }
  }
  lds_barrier();

  f32x4 acc[2][RT];
  // ---- hidden blocks ----
  for (int li = 0; li < Lc; ++li) {
    relaunder(c);
    const int K = li == 0 ? Z : J->cls_width[li - 1], N = J->cls_width[li], K32 = rup(K, 32);
    gcf32 Wl = prm + J->cls_w[li];
    // the layer's weights as one coalesced block through registers into Q (free in the forward pass) and the BatchNorm
    // affine parameters: all requested before anything waits (fragment loads from global inside the GEMM loop and
    // parameter loads behind the statistics' barriers were most of this head's forward time)
    WBlk<128> wb;
    wblk_load<128>(c, wb, Wl, N, K, 0, 0);
    float mean[2][4], rstd[2][4], gam[2][4], bet[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int fc = min((c.wn + 4 * t) * 16 + 4 * c.g + i, N - 1);
        gam[t][i] = prm[J->cls_bn_w[li] + fc];
        bet[t][i] = prm[J->cls_bn_b[li] + fc];
      }
    if (bwd) store_act(c, W.hin(li), c.P, K32);
    bias_acc(c, acc, prm + J->cls_b[li], N, 0);
    wblk_store<128>(c, wb, c.Q, LDP, N, K, 0, 0);
    lds_barrier();
    for (int ks = 0; ks < K32 / 32; ++ks) {
      bf16x8 wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) wf[t] = lds_frag(c.Q, LDP, (c.wn + 4 * t) * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        bf16x8 a = lds_frag(c.P, LDP, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][rt] = mfma(wf[t], a, acc[t][rt]);
      }
    }
    if (c.tid < PW) { col1[c.tid] = 0.f; col2[c.tid] = 0.f; }
    __syncthreads();                               // P fully read; column accumulators cleared
    if (train) {                                   // batch statistics over the valid rows (biased variance)
      float v[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float s = 0.f;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) s += (c.wm * WROWS + rt * 16 + c.c16 < c.nrows) ? acc[t][rt][i] : 0.f;
          v[t][i] = s;
        }
      col_reduce(c, v, col1, N);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = min((c.wn + 4 * t) * 16 + 4 * c.g + i, PW - 1);
          mean[t][i] = col1[f] / Bf;
          float s = 0.f;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const float d = acc[t][rt][i] - mean[t][i];
            s += (c.wm * WROWS + rt * 16 + c.c16 < c.nrows) ? d * d : 0.f;
          }
          v[t][i] = s;
        }
      col_reduce(c, v, col2, N);
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = (c.wn + 4 * t) * 16 + 4 * c.g + i, fc = min(f, N - 1);
        float var;
        if (train) { var = col2[min(f, PW - 1)] / Bf; }
        else { mean[t][i] = prm[J->cls_bn_mean[li] + fc]; var = prm[J->cls_bn_var[li] + fc]; }
        rstd[t][i] = 1.0f / sqrtf(var + 1e-5f);
      }
    if (c.tid < N && train) {                      // per-feature rstd for the backward pass; running statistics
      const float m = col1[c.tid] / Bf, var = col2[c.tid] / Bf;
      if (bwd) W.rstd(li)[c.tid] = 1.0f / sqrtf(var + 1e-5f);
      if (bn_stats && log) {
        gf32 rm = asg(J->params) + J->cls_bn_mean[li] + c.tid, rv = asg(J->params) + J->cls_bn_var[li] + c.tid;
        const float unb = c.nrows > 1 ? var * Bf / (Bf - 1.0f) : var;      // running_var takes the unbiased estimate
        *rm = 0.9f * *rm + 0.1f * m;
        *rv = 0.9f * *rv + 0.1f * unb;
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int f0 = (c.wn + 4 * t) * 16 + 4 * c.g;
      if (f0 >= rup(N, 32)) continue;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int r = c.wm * WROWS + rt * 16 + c.c16;
        f32x4 xh;
        bf16x4 pk;
        f32x4 u4 = {1.f, 1.f, 1.f, 1.f};
        if (train && J->cls_dropout > 0.f) u4 = uniform4_ctr(J->seed, (uint32_t)step, (uint32_t)li, (uint32_t)(c.row0 + r), (uint32_t)(f0 >> 2));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xh[i] = (acc[t][rt][i] - mean[t][i]) * rstd[t][i];
          float h = fmaxf(gam[t][i] * xh[i] + bet[t][i], 0.f);
          if (train && J->cls_dropout > 0.f) h = u4[i] >= J->cls_dropout ? h * keep_scale : 0.f;
          pk[i] = (__bf16)((f0 + i < N && r < c.nrows) ? h : 0.f);
        }
        if (bwd) *(GAS f32x4*)(W.xhat(li) + r * PW + f0) = xh;
        *reinterpret_cast<bf16x4*>(c.P + r * LDP + f0) = pk;
      }
    }
    lds_barrier();
  }

  tr(c, 26);
  // ---- output layer, cross entropy ----
  relaunder(c);
  const int Kl = Lc ? J->cls_width[Lc - 1] : Z, Kl32 = rup(Kl, 32);
  gcf32 Wo = prm + J->cls_w[Lc];
  if (bwd) store_act(c, W.hin(Lc), c.P, Kl32);
  bias_acc(c, acc, prm + J->cls_b[Lc], C, 0);
  for (int ks = 0; ks < Kl32 / 32; ++ks) {
    bf16x8 wf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) wf[t] = w_frag(Wo, C, Kl, (c.wn + 4 * t) * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      bf16x8 a = lds_frag(c.P, LDP, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t][rt] = mfma(wf[t], a, acc[t][rt]);
    }
  }
  lds_barrier();
  for (int e = c.tid; e < ROWS * 32; e += WG) c.P[(e >> 5) * LDP + (e & 31)] = (__bf16)0.0f;   // d logits land here
  lds_barrier();
  float ce = 0.f;
  if (c.wn == 0 && c.g == 0) {                     // these lanes hold logits 0..3 of their rows
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int r = c.wm * WROWS + rt * 16 + c.c16;
      if (r < c.nrows) {
        float l[NM_MAX_CLASSES], mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < NM_MAX_CLASSES; ++k) { l[k] = acc[0][rt][k]; if (k < C) mx = fmaxf(mx, l[k]); }
        if (J->out_logits)
#pragma unroll
          for (int k = 0; k < NM_MAX_CLASSES; ++k)
            asg(J->out_logits)[(int64_t)(c.row0 + r) * NM_MAX_CLASSES + k] = k < C ? l[k] : 0.f;
        if (J->labels) {
          const int y = asg(J->labels)[c.row0 + r];
          float se = 0.f;
#pragma unroll
          for (int k = 0; k < NM_MAX_CLASSES; ++k) se += k < C ? expf(l[k] - mx) : 0.f;
          const float lse = mx + logf(se);
#pragma unroll
          for (int k = 0; k < NM_MAX_CLASSES; ++k) {
            if (k == y) ce += lse - l[k];
            if (bwd && k < C) c.P[r * LDP + k] = (__bf16)((expf(l[k] - lse) - (k == y ? 1.f : 0.f)) * J->cls_w_ce * c.inv_b);
          }
        }
      }
    }
  }
  const float ce_sum = block_sum(c, ce);
  // ---- contrastive hinge on the per-subject deviations (cVAE.py:2166-2182) ----
  const int Me = experts(J);
  float hinge = 0.f;
  if (J->labels && c.tid < c.nrows && J->M >= 2 * Me) {
    const int gr = c.row0 + c.tid;
    float dh = 0.f, dd = 0.f;
    bool have = true;
    for (int m = 0; m < Me; ++m) {
      have = have && J->mod[m].out_rowdev && J->mod[Me + m].out_rowdev;
      if (have) { dh += asg(J->mod[m].out_rowdev)[gr]; dd += asg(J->mod[Me + m].out_rowdev)[gr]; }
    }
    if (have) {
      dh /= (float)Me; dd /= (float)Me;
      const int y = asg(J->labels)[gr];
      const float tval = y ? J->cls_margin + dd - dh : J->cls_margin + dh - dd;
      hinge = fmaxf(tval, 0.f);
      if (bwd) {
        const float gt = tval > 0.f ? J->cls_w_contrast * c.inv_b : 0.f;      // d total / d tval
        const float g_h = y ? -gt : gt;                                       // d / d dev_health; disease = -g_h
        for (int m = 0; m < Me; ++m) {
          if (J->rowcoef_out[m]) asg(J->rowcoef_out[m])[gr] = g_h * 2.0f / ((float)Me * (float)J->mod[m].D);
          if (J->rowcoef_out[Me + m]) asg(J->rowcoef_out[Me + m])[gr] = -g_h * 2.0f / ((float)Me * (float)J->mod[Me + m].D);
        }
      }
    }
  }
  const float hinge_sum = block_sum(c, hinge);
  if (c.tid == 0 && J->loss_log && J->labels && log) {
    gf32 row = asg(J->loss_log) + (int64_t)(step % J->loss_cap) * NM_LOSS_STRIDE;
    row[NM_LOSS_CE] = ce_sum * c.inv_b;
    row[NM_LOSS_CONTRAST] = hinge_sum * c.inv_b;
  }
  tr(c, 27);
  if (!bwd) return;

  // ---- backward: output layer ----
  relaunder(c);
  load_act(c, c.Q, W.hin(Lc), Kl32);
  lds_barrier();
  zero_acc(acc);
  dgrad_acc(c, acc, c.P, Wo, C, Kl, 1, 0);
  float gb = 0.f;
  if (c.tid < C) for (int r = 0; r < ROWS; ++r) gb += (float)c.P[r * LDP + c.tid];
  lds_barrier();                                   // Wo fully read before its update
  float* const hpatch = c.stage + SPATCH_OFF / 4;
  wgrad_adam<false>(c, c.P, LDP, 0, c.Q, LDP, WgGeom{C, Kl, 0, rup(Kl, 16), WgT{J->cls_w[Lc], -1, nullptr, 0, nullptr, hpatch}});
  if (c.tid < C) apply_grad(c, J->cls_b[Lc] + c.tid, gb);
  // ---- backward: hidden blocks ----
  for (int li = Lc - 1; li >= 0; --li) {
    relaunder(c);
    const int K = li == 0 ? Z : J->cls_width[li - 1], N = J->cls_width[li], K32 = rup(K, 32);
    gcf32 Wl = prm + J->cls_w[li];
    // acc = d h (pre-mask), Q = h of this block.  d y = d h * relu'/dropout mask; BatchNorm backward needs the
    // column sums S1 = sum d y, S2 = sum d y * x_hat
    float grs[2][4];                               // gamma * rstd of this lane's features: requested before the barriers
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int fc = min((c.wn + 4 * t) * 16 + 4 * c.g + i, N - 1);
        grs[t][i] = prm[J->cls_bn_w[li] + fc] * W.rstd(li)[fc];
      }
    if (c.tid < PW) { col1[c.tid] = 0.f; col2[c.tid] = 0.f; }
    __syncthreads();
    float v1[2][4], v2[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int f0 = (c.wn + 4 * t) * 16 + 4 * c.g;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v1[t][i] = 0.f; v2[t][i] = 0.f; }
      if (f0 >= rup(N, 32)) continue;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int r = c.wm * WROWS + rt * 16 + c.c16;
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(c.Q + r * LDP + f0);
        const f32x4 xh = *(const GAS f32x4*)(W.xhat(li) + r * PW + f0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float dy = ((float)h[i] > 0.f && r < c.nrows && f0 + i < N) ? acc[t][rt][i] * keep_scale : 0.f;
          acc[t][rt][i] = dy;
          v1[t][i] += dy;
          v2[t][i] = fmaf(dy, xh[i], v2[t][i]);
        }
      }
    }
    col_reduce(c, v1, col1, N);
    col_reduce(c, v2, col2, N);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int f0 = (c.wn + 4 * t) * 16 + 4 * c.g;
      if (f0 >= rup(N, 32)) continue;
      float s1[4], s2[4], gr_[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int fc = min(f0 + i, N - 1);
        s1[i] = col1[min(f0 + i, PW - 1)] * c.inv_b;
        s2[i] = col2[min(f0 + i, PW - 1)] * c.inv_b;
        gr_[i] = grs[t][i];
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int r = c.wm * WROWS + rt * 16 + c.c16;
        const f32x4 xh = *(const GAS f32x4*)(W.xhat(li) + r * PW + f0);
        bf16x4 pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float da = gr_[i] * (acc[t][rt][i] - s1[i] - xh[i] * s2[i]);
          pk[i] = (__bf16)((r < c.nrows && f0 + i < N) ? da : 0.f);
        }
        *reinterpret_cast<bf16x4*>(c.P + r * LDP + f0) = pk;
      }
    }
    __syncthreads();                               // gamma has been read by everyone: its update may go ahead
    if (c.tid < N) {                               // d gamma = S2, d beta = S1
      apply_grad(c, J->cls_bn_w[li] + c.tid, col2[c.tid]);
      apply_grad(c, J->cls_bn_b[li] + c.tid, col1[c.tid]);
    }
    load_act(c, c.Q, W.hin(li), K32);
    lds_barrier();
    float gbi = 0.f;
    if (c.tid < N) for (int r = 0; r < ROWS; ++r) gbi += (float)c.P[r * LDP + c.tid];
    zero_acc(acc);
    dgrad_acc(c, acc, c.P, Wl, N, K, rup(N, 32) / 32, 0);
    lds_barrier();
    wgrad_adam<false>(c, c.P, LDP, 0, c.Q, LDP, WgGeom{N, K, 0, rup(K, 16), WgT{J->cls_w[li], -1, nullptr, 0, nullptr, hpatch}});
    if (c.tid < N) apply_grad(c, J->cls_b[li] + c.tid, gbi);
  }
  // ---- d CE / d z ----
  if (J->dz_out) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int k0 = (c.wn + 4 * t) * 16 + 4 * c.g;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int r = c.wm * WROWS + rt * 16 + c.c16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (r < c.nrows && k0 + i < Z) asg(J->dz_out)[(int64_t)(c.row0 + r) * Z + k0 + i] = acc[t][rt][i];
      }
    }
  }
}
__global__ __launch_bounds__(WG) void nm_clshead_kernel(const nm_job_t* __restrict__ jobs, int step, int tile0, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const nm_job_t* J = jobs + blockIdx.x;
  Ctx c;
  if (!head_setup(c, smem, J, step, tile0, flags & (NM_F_BACKWARD | NM_F_ADAM | NM_F_GRADS))) return;
  cls_head_body(c, J, step, c.ws + trunk_ws_bytes(J), blockIdx.y == 0, (flags & NM_F_BNSTATS) != 0);
}

// ---- head models in one persistent launch ------------------------------------------------------------------------
// One workgroup per model, n_steps train steps of a regression / end-to-end model (the loops of
// multimodal_kfold_train_cvae_supervised_regression.py:112-125 and multimodal_kfold_cvae_nmpmcont.py:257-303), per step:
//   pass 1  trunk forward: encoders, fusion, every decoder; reconstructions / latent / per-subject deviations exported,
//           every activation saved (run_step MODE 1);
//   head    forward, its loss, backward, its Adam update; d loss / d x_hat (regression) or d CE / d z and the hinge row
//           coefficients (classifier) left in the job's exchange buffers;
//   pass 2  per decoder: output chunks again from the saved last hidden activation, now with those extra gradients,
//           NLL backward, dgrad, wgrad + Adam; then the decoders' hidden layers, fusion and encoders backward (MODE 2).
// The trunk's forward runs once (the three-launch form ran it twice), nothing returns to the host between steps.
__global__ __launch_bounds__(WG) void nm_head_step_kernel(const nm_job_t* __restrict__ jobs, int step0, int n_steps, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const nm_job_t* J = jobs + blockIdx.x;
  Ctx c;
  c.job = J;
  c.part = -1; c.nparts = 1; c.slope = J->act_slope;
  carve_lds(c, smem);
  relaunder(c);
  c.t_last = 0;
  c.ws = (GAS char*)J->workspace;
  for (int i = c.tid; i < SMEM_BYTES / 4; i += WG) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  __syncthreads();
  GAS char* const hws = c.ws + trunk_ws_bytes(J);
  const int nb = (J->n_rows + ROWS - 1) / ROWS;
  const int tflags = flags & (NM_F_PROFILE | NM_F_TRACE);
  // NM_F_GRADS: gradients of the step's total into job.grads, no update (the eager facade's backward)
  const int bflags = NM_F_BACKWARD | ((flags & NM_F_GRADS) ? NM_F_GRADS : NM_F_ADAM);
  for (int s = step0; s < step0 + n_steps; ++s) {
    c.lstep = s - step0;
    c.row0 = (s % nb) * ROWS;
    c.nrows = min(ROWS, J->n_rows - c.row0);
    c.inv_b = 1.0f / (float)c.nrows;
    const int64_t t_opt = J->adam_off + (int64_t)s + 1;
    const double tt = (double)t_opt;
    const double lr_t = (J->lr_table && J->lr_cap > 0) ? J->lr_table[(t_opt - 1) % J->lr_cap] : (double)J->lr;
    c.step_size = (float)(lr_t / (1.0 - pow((double)J->beta1, tt)));
    c.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)J->beta2, tt)));
    if (flags & 64) c.tlast[c.wave_s] = clock64();
    lds_barrier();
    relaunder(c);
    c.flags = NM_F_EXPORT | tflags;
    run_step<false, 1>(c, s);
    handoff_barrier();                            // exports and saved activations are complete
    tr(c, 20);
    relaunder(c);
    c.flags = bflags | tflags;
    if (J->reg_head) reg_head_body(c, J, s, hws, true);
    else if (J->cls_classes > 0) cls_head_body(c, J, s, hws, true, (flags & NM_F_BNSTATS) != 0);
    handoff_barrier();                            // the head's gradients for the trunk are complete
    tr(c, 29);
    relaunder(c);
    c.flags = bflags | tflags;
    run_step<false, 2>(c, s);
    handoff_barrier();                            // the next step reads what this one stored
  }
}

// ---- stand-alone kernels ----------------------------------------------------------------------
__global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, int64_t n, float b1, float b2, float eps, float step_size,
                                 float inv_bc2_sqrt) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const AdamK a{b1, b2, eps, step_size, inv_bc2_sqrt};
  for (; i < n; i += stride) {
    float pp = p[i], mm = m[i], vv = v[i];
    adam1(a, g[i], pp, mm, vv);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

// Shadow images from the fp32 master: one workgroup per job.  A matrix is walked in master (tile) order; element
// (n, k) goes to image row row0 + n, column k of `img` (row pitch `pitch` bytes), k-chunked every `kchunk` columns
// with `kstride` bytes between chunk images, n-chunked every `nchunk` rows with `nstride` bytes between chunk blobs.
// (the shadow rebuild runs as gridDim.y blocks per job that share the element loops)
__device__ __forceinline__ int sh_tid() { return (int)(blockIdx.y * blockDim.x + threadIdx.x); }
__device__ __forceinline__ int sh_nthr() { return (int)(blockDim.x * gridDim.y); }
__device__ __forceinline__ void sync_matrix(const float* prm, int64_t w_off, int N, int K, char* img, int pitch, int kchunk,
                                            int64_t kstride, int nchunk, int64_t nstride) {
  const int KT = ktiles(K);
  const int64_t total = wt_elems(N, K);
  for (int64_t e = sh_tid(); e < total; e += sh_nthr()) {
    const int tile = (int)(e >> 8), r = (int)(e >> 4) & 15, cidx = (int)e & 15;
    const int n = (tile / KT) * 16 + r, k = (tile % KT) * 16 + cidx;
    const __bf16 h = (__bf16)prm[w_off + e];
    char* dst = img + (int64_t)(k / kchunk) * kstride + (int64_t)(n / nchunk) * nstride + (int64_t)(n % nchunk) * pitch +
                (k % kchunk) * 2;
    *reinterpret_cast<__bf16*>(dst) = h;
  }
}
// zero the split-mode hand-off counters of every job (before each NM_F_SPLIT launch)
__global__ void sync_reset_kernel(const nm_job_t* __restrict__ jobs, int n_jobs) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_jobs * (WS_SYNC_BYTES / 4)) {
    const nm_job_t* J = jobs + j / (WS_SYNC_BYTES / 4);
    const int w = j % (WS_SYNC_BYTES / 4);
    if (w != WS_SYNC_ERR_WORD)              // the error word is sticky: read by nm_split_errors, cleared by its `clear`
      ((unsigned*)((char*)J->workspace + ws_layout(J->M, J->L, J->Z).sync))[w] = 0u;
  }
}
// out[j] = error word of job j (a split launch's hand-off timed out); clear != 0 zeroes the words afterwards
__global__ void split_errors_kernel(const nm_job_t* __restrict__ jobs, int n_jobs, int* __restrict__ out, int clear) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_jobs) {
    const nm_job_t* J = jobs + j;
    unsigned* w = (unsigned*)((char*)J->workspace + ws_layout(J->M, J->L, J->Z).sync) + WS_SYNC_ERR_WORD;
    out[j] = (int)*w;
    if (clear) *w = 0u;
  }
}

__global__ void sync_shadow_kernel(const nm_job_t* __restrict__ jobs) {
  const nm_job_t* J = jobs + blockIdx.x;
  const float* prm = J->params;
  char* wsh = (char*)J->wsh;
  if (!wsh) return;
  const int L = J->L, Z = J->Z, C = J->C, Zs = rup(Z, 16);
  const int Me = J->M_enc > 0 ? J->M_enc : J->M;
  const int BIG = 1 << 30;
  for (int m = 0; m < J->M; ++m) {
    const nm_modality_t& md = J->mod[m];
    if (m < Me) {
      const int nch = (md.Kx + XCH - 1) / XCH;
      sync_matrix(prm, md.enc_w[0], J->H[0], md.D + C, wsh + md.enc_s[0], md.Kx * 2, BIG, 0, BIG, 0);
      float* b0 = (float*)(wsh + md.enc_s[0] + l0_img_bytes(J->H[0], md.Kx));
      for (int i = sh_tid(); i < J->H[0]; i += sh_nthr()) b0[i] = prm[md.enc_b[0] + i];
      for (int e = 1; e < L; ++e) {
        sync_matrix(prm, md.enc_w[e], J->H[e], J->H[e - 1], wsh + md.enc_s[e], blob_kp(J->H[e - 1]) * 2, BIG, 0, BIG, 0);
        float* b = (float*)(wsh + md.enc_s[e] + cimg_bytes(J->H[e], J->H[e - 1]));
        for (int i = sh_tid(); i < J->H[e]; i += sh_nthr()) b[i] = prm[md.enc_b[e] + i];
      }
      const int hkp = blob_kp(J->H[L - 1]);
      sync_matrix(prm, md.mu_w, Z, J->H[L - 1], wsh + md.heads_s, hkp * 2, BIG, 0, BIG, 0);
      sync_matrix(prm, md.lv_w, Z, J->H[L - 1], wsh + md.heads_s + (int64_t)Zs * hkp * 2, hkp * 2, BIG, 0, BIG, 0);
      float* bh = (float*)(wsh + md.heads_s + cimg_bytes(2 * Zs, J->H[L - 1]));
      for (int i = sh_tid(); i < Z; i += sh_nthr()) { bh[i] = prm[md.mu_b + i]; bh[Zs + i] = prm[md.lv_b + i]; }
    }
    for (int d = 0; d < L; ++d) {
      const int Kin = d == 0 ? Z + C : J->H[L - d], Nout = J->H[L - 1 - d];
      sync_matrix(prm, md.dec_w[d], Nout, Kin, wsh + md.dec_s[d], blob_kp(Kin) * 2, BIG, 0, BIG, 0);
      float* b = (float*)(wsh + md.dec_s[d] + cimg_bytes(Nout, Kin));
      for (int i = sh_tid(); i < Nout; i += sh_nthr()) b[i] = prm[md.dec_b[d] + i];
    }
    sync_matrix(prm, md.out_w, md.D, J->H[0], wsh + md.out_s, LDP * 2, BIG, 0, OCH, OBLOB_BYTES);
    for (int i = sh_tid(); i < md.D; i += sh_nthr()) {
      float* vb = (float*)(wsh + md.out_s + (int64_t)(i / OCH) * OBLOB_BYTES + OIMG_BYTES);
      vb[i % OCH] = prm[md.out_b + i];
      if (J->out_kind == 0) vb[OCH + i % OCH] = prm[md.logvar_out + i];
    }
  }
  if (J->reg_head) {                               // regressor.0: chunk images [128][72] over the padded concatenation + bias
    const int nq = head_chunk0(J, Me);
    sync_matrix(prm, J->reg_w[0], 128, nq * XCH, wsh + J->reg_s, LDX * 2, XCH, W0IMG_BYTES, BIG, 0);
    float* b = (float*)(wsh + J->reg_s + (int64_t)nq * W0IMG_BYTES);
    for (int i = sh_tid(); i < 128; i += sh_nthr()) b[i] = prm[J->reg_b[0] + i];
  }
}

// xb images [tile][chunk][256][LDX] (x | c | 1 | 0 in 64-column chunks, 8 pad columns per row), the fp32 copy of x
// and the covariate block cz = c | 1 | 0.
__global__ void pack_table_kernel(const float* __restrict__ x, const float* __restrict__ cc, int n_rows, int rows_alloc,
                                  int D, int C, int Kx, uint16_t* __restrict__ xb, float* __restrict__ xf, int xp,
                                  uint16_t* __restrict__ cz, int Cz) {
  const int nch = (Kx + XCH - 1) / XCH;
  const int64_t total = (int64_t)rows_alloc * nch * LDX;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // i -> (tile, chunk, row in tile, column in chunk)
    const int j = (int)(i % LDX);
    const int64_t q = i / LDX;
    const int rl = (int)(q % ROWS);
    const int64_t q2 = q / ROWS;
    const int kc = (int)(q2 % nch), tile = (int)(q2 / nch);
    const int r = tile * ROWS + rl, k = kc * XCH + j;
    float v = 0.f;
    if (r < n_rows && j < XCH) {
      if (k < D) v = x[(int64_t)r * D + k];
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
      xf[i] = (r < n_rows && k < D) ? x[(int64_t)r * D + k] : 0.f;
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

// Unit-test kernel: one workgroup, through the production fragment loaders and lane maps.
//   mode 0  forward form : C[r][n] = sum_k A[r][k] B[n][k]   (A [256][K] via P, B fp32 [N][K])
//   mode 1  dgrad form   : C[r][k] = sum_n A[r][n] B[n][k]   (A [256][N] via P, B fp32 [N][K])
//   mode 2/3 wgrad form  : C[n][k] = sum_r A[r][n] B[r][k]   (ds_read_b64_tr_b16 / scalar loaders)
__global__ __launch_bounds__(WG) void test_gemm_kernel(int mode, const float* A, const float* B, float* Cout, int M,
                                                       int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Ctx c;
  carve_lds(c, smem);
  relaunder(c);
  for (int i = c.tid; i < SMEM_BYTES / 4; i += WG) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  __syncthreads();
  if (mode == 0 || mode == 1) {
    int KA = (mode == 0) ? K : N;
    for (int e = c.tid; e < ROWS * KA; e += WG) { int r = e / KA, k = e - r * KA; c.P[r * LDP + k] = (__bf16)A[e]; }
    __syncthreads();
    f32x4 acc[2][RT];
    zero_acc(acc);
    int ncols = (mode == 0) ? N : K;
    if (mode == 0) {
      for (int ks = 0; ks < rup(K, 32) / 32; ++ks) {
        bf16x8 wf[2];
        for (int t = 0; t < 2; ++t) wf[t] = w_frag(asg(B), N, K, (c.wn + 4 * t) * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          bf16x8 a = lds_frag(c.P, LDP, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t][rt] = mfma(wf[t], a, acc[t][rt]);
        }
      }
    } else {
      dgrad_acc(c, acc, c.P, asg(B), N, K, rup(N, 32) / 32, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      int f0 = (c.wn + 4 * t) * 16 + 4 * c.g;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int r = c.wm * WROWS + rt * 16 + c.c16;
          if (f0 + i < ncols) Cout[(int64_t)r * ncols + f0 + i] = acc[t][rt][i];
        }
    }
  } else {
    // A [256][N], B [256][K]; output [N][K]
    for (int e = c.tid; e < ROWS * N; e += WG) { int r = e / N, k = e - r * N; c.P[r * LDP + k] = (__bf16)A[e]; }
    for (int e = c.tid; e < ROWS * K; e += WG) { int r = e / K, k = e - r * K; c.Q[r * LDP + k] = (__bf16)B[e]; }
    __syncthreads();
    const int ntn = (N + 15) / 16, nkt = (K + 15) / 16, kpairs = (nkt + 1) / 2;
    for (int u = c.wave; u < ntn * kpairs; u += NWAVES) {
      int nt = u / kpairs, kp = u % kpairs;
      f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
      for (int rs = 0; rs < ROWS / 32; ++rs) {
        bf16x8 bn, ak[2];
        if (mode == 3) {
          bn = lds_frag_tr_scalar(c.P, LDP, rs * 32, nt * 16, c.lane);
          for (int t = 0; t < 2; ++t) ak[t] = lds_frag_tr_scalar(c.Q, LDP, rs * 32, (kp * 2 + t) * 16, c.lane);
        } else {
          unsigned na = tr_addr_il(c.P, LDP, rs * 32, nt * 16, c.lane);
          unsigned ka = tr_addr_il(c.Q, LDP, rs * 32, kp * 32, c.lane);
          unsigned na1 = na + 1u * LDP * 2u, ka1 = ka + 1u * LDP * 2u;
          bf16x4 n0v, n1v, k0v[2], k1v[2];
          NM_TR_READ(n0v, na, 0); NM_TR_READ(n1v, na1, 0);
          NM_TR_READ(k0v[0], ka, 0);  NM_TR_READ(k1v[0], ka1, 0);
          NM_TR_READ(k0v[1], ka, 32); NM_TR_READ(k1v[1], ka1, 32);
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(n0v), "+v"(n1v), "+v"(k0v[0]), "+v"(k1v[0]), "+v"(k0v[1]), "+v"(k1v[1]));
          bn = join4(n0v, n1v);
          for (int t = 0; t < 2; ++t) ak[t] = join4(k0v[t], k1v[t]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = mfma(ak[t], bn, acc[t]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        int n = nt * 16 + c.c16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int k = (kp * 2 + t) * 16 + 4 * c.g + i;
          if (n < N && k < K) Cout[(int64_t)n * K + k] = acc[t][i];
        }
      }
    }
  }
}

}  // namespace

// ================================= C ABI ========================================================
extern "C" {

int nm_version(void) { return 8; }

/* phase profile (NM_F_PROFILE): read / reset the per-phase shader-clock accumulators */
int nm_prof_read(unsigned long long* out32, int reset) {
  if (!out32) return -1;
  hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(nm_prof_cycles), sizeof(unsigned long long) * 32);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    unsigned long long z[32] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(nm_prof_cycles), z, sizeof(z));
  }
  return (int)e;
}

/* NM_F_TRACE read-out: [8 waves][64 tags] interval cycles of workgroup (0,0); reset != 0 clears. */
int nm_trace_read(unsigned long long* out512, int reset) {
  if (!out512) return -1;
  hipError_t e = hipMemcpyFromSymbol(out512, HIP_SYMBOL(nm_trace_cycles), sizeof(unsigned long long) * 512);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    static unsigned long long z[512];
    e = hipMemcpyToSymbol(HIP_SYMBOL(nm_trace_cycles), z, sizeof(z));
  }
  return (int)e;
}

int nm_wgtimes_read(unsigned long long* out1024) {
  if (!out1024) return -1;
  return (int)hipMemcpyFromSymbol(out1024, HIP_SYMBOL(nm_wg_times), sizeof(unsigned long long) * 1024);
}

int nm_abi_sizes(int64_t* sizeof_job, int64_t* sizeof_modality) {
  if (!sizeof_job || !sizeof_modality) return -1;
  *sizeof_job = (int64_t)sizeof(nm_job_t);
  *sizeof_modality = (int64_t)sizeof(nm_modality_t);
  return 0;
}

const char* nm_status_string(int status) {
  switch (status) {
    case 0: return "ok";
    case -1: return "null pointer";
    case -2: return "modalities out of range (1..NM_MAX_MOD decoders, 1..NM_MAX_EXP experts)";
    case -3: return "hidden layers out of range (1..NM_MAX_HID)";
    case -4: return "hidden width out of range (1..NM_MAX_WIDTH)";
    case -5: return "latent out of range (1..NM_MAX_LATENT)";
    case -6: return "latent + c_dim exceeds NM_MAX_WIDTH";
    case -7: return "table pitch: Kx must be a multiple of 32 and >= D + C + 1, x_pitch a multiple of 4 and >= D, Cz a multiple of 8 and >= C + 1";
    case -14: return "n_rows, loss_cap and eps_cap must be >= 1";
    case -18: return "out_kind must be 0 or 1, 0 <= n_private <= Z, and a private latent needs an encoder per decoder";
    case -19: return "general-shape path (wide): cVAE / cVAE_multimodal / end-to-end trunk only (no regression head, no DMVAE-family or mvtCAE switches)";
    case -17: return "input preparation: 1 <= rows <= NM_PREP_MAX_ROWS, at least one source / column / bin";
    case -16: return "split launch: jobs x parts exceeds the number of CUs (the parts of a model wait for each other and must all be resident)";
    case -15: return "wsh (shadow images) missing: allocate nm_fill_shadow() bytes, zero them and call nm_sync_shadow()";
    case -8: return "bad launch geometry";
    case -9: return "unknown combine";
    case -10: return "parameter tensor offsets must be multiples of 4 floats (weight matrices: of 256)";
    case -13: return "classifier head: 0..NM_MAX_CLS blocks of width 1..128, 2..NM_MAX_CLASSES classes, offsets multiples of 4, out_mu/out_z export";
    case -12: return "metrics: n_sets >= 1 and 1 <= max_set <= NM_METRICS_MAX_N";
    case -11: return "regression head: needs reg_w / reg_b offsets (weights: multiples of 256, biases: of 4) and the reg_resid / reg_dres image buffers (16-byte aligned)";
    default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown argument error";
  }
}

int nm_validate_job(const nm_job_t* j) {
  if (!j) return -1;
  if (j->M < 1 || j->M > NM_MAX_MOD) return -2;
  if (j->M_enc < 0 || j->M_enc > j->M || (j->M_enc == 0 ? j->M : j->M_enc) > NM_MAX_EXP) return -2;
  if (j->L < 1 || j->L > NM_MAX_HID) return -3;
  if (j->wide) {
    // the general-shape path (nm_launch_wide): any width, latent <= 128; the plain cVAE / cVAE_multimodal model only
    for (int i = 0; i < j->L; ++i)
      if (j->H[i] < 1 || j->H[i] > NM_WIDE_MAX_WIDTH) return -4;
    if (j->Z < 1 || j->Z > NM_WIDE_MAX_LATENT) return -5;
    // (the classifier head of the end-to-end model runs as its own kernel, nm_head_classifier, on any trunk)
    if (j->out_kind != 0 || j->n_private != 0 || j->tc_weight != 0.f || j->w_off >= 0 || j->reg_head ||
        j->combine == NM_COMBINE_POE2V)
      return -19;
  } else {
  for (int i = 0; i < j->L; ++i)
    if (j->H[i] < 1 || j->H[i] > NM_MAX_WIDTH) return -4;
  if (j->Z < 1 || j->Z > NM_MAX_LATENT) return -5;
  if (j->Z + j->C > NM_MAX_WIDTH) return -6;
  }
  if (j->combine < 0 || j->combine > NM_COMBINE_POE2V) return -9;
  if (j->out_kind < 0 || j->out_kind > 1 || j->n_private < 0 || j->n_private > j->Z) return -18;
  if (j->n_private > 0 && j->M_enc != 0 && j->M_enc != j->M) return -18;      // a private latent needs the modality's own encoder
  if (j->n_rows < 1 || j->loss_cap < 1 || j->eps_cap < 1) return -14;       // modulo divisors / batch count in the kernel
  if (!j->wsh && !j->wide) return -15;
  for (int m = 0; m < j->M; ++m) {
    const nm_modality_t& md = j->mod[m];
    if (md.D < 1 || md.Kx % 32 != 0 || md.Kx < md.D + j->C + 1) return -7;
    if (md.x_pitch % 4 != 0 || md.x_pitch < md.D) return -7;
    if (md.Cz % 8 != 0 || md.Cz < j->C + 1) return -7;
    for (int i = 0; i < j->L; ++i) {
      if ((md.enc_b[i] | md.dec_b[i]) & 3) return -10;
      if ((md.enc_w[i] | md.dec_w[i]) & 255) return -10;
    }
    if ((md.mu_b | md.lv_b | md.out_b) & 3) return -10;
    if (j->out_kind == 0 && (md.logvar_out < 0 || (md.logvar_out & 3))) return -10;
    if ((md.mu_w | md.lv_w | md.out_w) & 255) return -10;
  }
  if (j->cls_classes > 0) {
    if (j->cls_layers < 0 || j->cls_layers > NM_MAX_CLS || j->cls_classes < 2 || j->cls_classes > NM_MAX_CLASSES) return -13;
    for (int i = 0; i < j->cls_layers; ++i) {
      if (j->cls_width[i] < 1 || j->cls_width[i] > PW) return -13;
      if ((j->cls_b[i] | j->cls_bn_w[i] | j->cls_bn_b[i] | j->cls_bn_mean[i] | j->cls_bn_var[i]) & 3) return -13;
      if (j->cls_w[i] & 255) return -13;
    }
    if ((j->cls_b[j->cls_layers] & 3) || (j->cls_w[j->cls_layers] & 255)) return -13;
    if (!(j->cls_use_mu ? j->out_mu : j->out_z)) return -13;
  }
  if (j->reg_head) {
    for (int i = 0; i < 3; ++i)
      if (j->reg_w[i] < 0 || j->reg_b[i] < 0 || (j->reg_w[i] & 255) || (j->reg_b[i] & 3)) return -11;
    if (!j->reg_resid || !j->reg_dres || ((uintptr_t)j->reg_resid & 15) || ((uintptr_t)j->reg_dres & 15)) return -11;
  }
  return 0;
}

int64_t nm_fill_shadow(nm_job_t* j) {
  if (!j) return -1;
  if (j->M < 1 || j->M > NM_MAX_MOD || j->L < 1 || j->L > NM_MAX_HID) return -2;
  const int Me = j->M_enc > 0 ? j->M_enc : j->M;
  const int Zs = rup(j->Z, 16), L = j->L;
  int64_t o = WSH_ZERO_BYTES;                      // a line of zeros first: the source of every pad segment (dma_img)
  for (int m = 0; m < j->M; ++m) {
    nm_modality_t& md = j->mod[m];
    if (md.Kx < 32 || md.D < 1) return -7;
    for (int i = 0; i < NM_MAX_HID; ++i) { md.enc_s[i] = 0; md.dec_s[i] = 0; }
    md.heads_s = 0;
    if (m < Me) {
      const int nch = (md.Kx + XCH - 1) / XCH;
      md.enc_s[0] = o; o += (int64_t)l0_img_bytes(j->H[0], md.Kx) + VEC_BYTES;
      for (int e = 1; e < L; ++e) { md.enc_s[e] = o; o += cblob_bytes(j->H[e], j->H[e - 1]); }
      md.heads_s = o; o += cblob_bytes(2 * Zs, j->H[L - 1]);
    }
    for (int d = 0; d < L; ++d) {
      md.dec_s[d] = o;
      o += cblob_bytes(j->H[L - 1 - d], d == 0 ? j->Z + j->C : j->H[L - d]);
    }
    md.out_s = o; o += (int64_t)((md.D + OCH - 1) / OCH) * OBLOB_BYTES;
  }
  j->reg_s = 0;
  if (j->reg_head) { j->reg_s = o; o += (int64_t)head_chunk0(j, Me) * W0IMG_BYTES + VEC_BYTES; }
  return o;
}

int nm_sync_shadow(const nm_job_t* jobs_dev, int n_jobs, void* stream) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1) return -8;
  // few jobs (the eager facade: one): many blocks per job, so that the rebuild is not one workgroup's serial loop
  const int slices = n_jobs >= 64 ? 4 : (n_jobs >= 8 ? 16 : 64);
  hipLaunchKernelGGL(sync_shadow_kernel, dim3(n_jobs, slices), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  return (int)hipGetLastError();
}

int64_t nm_workspace_bytes(const nm_job_t* j) {
  if (!j) return -1;
  int64_t b = trunk_ws_bytes(j);                      // the head's region sits behind the trunk's
  int64_t hb = 0;
  if (j->reg_head) hb = ACT_BYTES;        // regression head: its first hidden activation, kept for the backward pass
  if (j->cls_layers > 0 || j->cls_classes > 0) hb = hb > cls_ws_bytes() ? hb : cls_ws_bytes();
  return b + (hb + 255) / 256 * 256;
}

static int launch_impl(const nm_job_t* jobs_dev, int n_jobs, int step0, int steps_per_tile, int n_tiles, int flags,
                       void* stream, bool scalar_tr, int parts = 1) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1 || steps_per_tile < 1 || n_tiles < 1 || step0 < 0) return -8;
  // concurrent tiles of one job share its parameters, moments and gradient buffer: forward-only
  if (n_tiles > 1 && (flags & (NM_F_BACKWARD | NM_F_ADAM | NM_F_GRADS))) return -8;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(n_jobs, n_tiles), block(WG);
  flags &= ~NM_F_SPLIT;
  if (parts <= 1) flags &= ~NM_F_FAULT_INJECT;
  if (parts > 1) {
    // several workgroups per model: they wait for each other inside the launch, so every one of them must be
    // resident at once -- one workgroup per CU (LDS), hence at most as many workgroups as the device has CUs
    if (n_tiles != 1 || parts > NM_MAX_MOD || !(flags & NM_F_BACKWARD)) return -8;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -8;
    const int wgs = (n_jobs + 7) / 8 * 8 * parts;
    if (wgs > cus) return -16;
    grid = dim3(wgs, 1);
    flags |= NM_F_SPLIT;
    hipLaunchKernelGGL(sync_reset_kernel, dim3((n_jobs * (WS_SYNC_BYTES / 4) + 255) / 256), dim3(256), 0, st, jobs_dev, n_jobs);
  }
  hipError_t e;
  if (!scalar_tr && !(flags & NM_F_BACKWARD)) {           // forward only: the instantiation without the backward pass
    e = hipFuncSetAttribute((const void*)nm_step_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((nm_step_kernel<false, 3>), grid, block, SMEM_BYTES, st, jobs_dev, step0, steps_per_tile, flags, n_jobs, parts);
  } else if (scalar_tr) {
    e = hipFuncSetAttribute((const void*)nm_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nm_step_kernel<true>, grid, block, SMEM_BYTES, st, jobs_dev, step0, steps_per_tile, flags, n_jobs, parts);
  } else {
    e = hipFuncSetAttribute((const void*)nm_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nm_step_kernel<false>, grid, block, SMEM_BYTES, st, jobs_dev, step0, steps_per_tile, flags, n_jobs, parts);
  }
  return (int)hipGetLastError();
}

/* The general-shape path (csrc/nm_wide.inc): jobs with nm_job_t.wide = 1 -- hidden widths > 127, latent > 64 or
 * latent + c_dim > 127 -- one workgroup per (job, tile), same launch contract as nm_launch. */
int nm_launch_wide(const nm_job_t* jobs_dev, int n_jobs, int step0, int steps_per_tile, int n_tiles, int flags, void* stream) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1 || steps_per_tile < 1 || n_tiles < 1 || step0 < 0) return -8;
  if (n_tiles > 1 && (flags & (NM_F_BACKWARD | NM_F_ADAM | NM_F_GRADS))) return -8;
  hipError_t e = hipFuncSetAttribute((const void*)nm_wide_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
  if (e != hipSuccess) return (int)e;
  flags &= (NM_F_BACKWARD | NM_F_ADAM | NM_F_GRADS | NM_F_EXPORT | NM_F_ZGIVEN | NM_F_TRACE);
  hipLaunchKernelGGL(nm_wide_step_kernel, dim3(n_jobs, n_tiles), dim3(WG), SMEM_BYTES, (hipStream_t)stream, jobs_dev, step0,
                     steps_per_tile, flags);
  return (int)hipGetLastError();
}

/* Small sweeps: every model runs as `parts` workgroups, one per modality (decoder), which meet twice per step (after
 * the encoders: expert statistics; after the decoders: d z).  All jobs of the launch must have M == parts decoders.
 * Results equal the one-workgroup launch bit for bit.  -16: more workgroups than CUs (they could not all be resident). */
int nm_launch_split(const nm_job_t* jobs_dev, int n_jobs, int parts, int step0, int n_steps, int flags, void* stream) {
  return launch_impl(jobs_dev, n_jobs, step0, n_steps, 1, flags, stream, false, parts);
}

/* Error words of the jobs' split launches: out_dev[j] != 0 <=> a hand-off of job j timed out in some nm_launch_split
 * since the words were last cleared (its parts left the launch; the parameters are not to be trusted). */
int nm_split_errors(const nm_job_t* jobs_dev, int n_jobs, int* out_dev, int clear, void* stream) {
  if (!jobs_dev || !out_dev) return -1;
  if (n_jobs < 1) return -8;
  hipLaunchKernelGGL(split_errors_kernel, dim3((n_jobs + 255) / 256), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs, out_dev, clear);
  return (int)hipGetLastError();
}

int nm_launch(const nm_job_t* jobs_dev, int n_jobs, int step0, int steps_per_tile, int n_tiles, int flags,
              void* stream) {
  return launch_impl(jobs_dev, n_jobs, step0, steps_per_tile, n_tiles, flags, stream, false);
}

/* same as nm_launch but with the scalar transposing loader (validation of ds_read_b64_tr_b16) */
int nm_launch_scalar_tr(const nm_job_t* jobs_dev, int n_jobs, int step0, int steps_per_tile, int n_tiles, int flags,
                        void* stream) {
  return launch_impl(jobs_dev, n_jobs, step0, steps_per_tile, n_tiles, flags, stream, true);
}

int nm_head_regression(const nm_job_t* jobs_dev, int n_jobs, int step, int tile0, int n_tiles, int flags, void* stream) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1 || n_tiles < 1 || step < 0 || tile0 < 0) return -8;
  // concurrent tiles of one job share its parameters, moments, gradient buffer and reg_dres: forward-only
  if (n_tiles > 1 && (flags & (NM_F_BACKWARD | NM_F_ADAM | NM_F_GRADS))) return -8;
  hipError_t e = hipFuncSetAttribute((const void*)nm_reghead_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nm_reghead_kernel, dim3(n_jobs, n_tiles), dim3(WG), SMEM_BYTES, (hipStream_t)stream, jobs_dev, step,
                     tile0, flags);
  return (int)hipGetLastError();
}

int nm_head_classifier(const nm_job_t* jobs_dev, int n_jobs, int step, int tile0, int n_tiles, int flags, void* stream) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1 || n_tiles < 1 || step < 0 || tile0 < 0) return -8;
  if (n_tiles > 1 && (flags & (NM_F_BACKWARD | NM_F_ADAM | NM_F_GRADS))) return -8;
  hipError_t e = hipFuncSetAttribute((const void*)nm_clshead_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nm_clshead_kernel, dim3(n_jobs, n_tiles), dim3(WG), SMEM_BYTES, (hipStream_t)stream, jobs_dev, step,
                     tile0, flags);
  return (int)hipGetLastError();
}

/* n_steps train steps of head models (every job: regression head, or end-to-end with classifier, labels / targets and
 * exchange buffers set) in ONE persistent launch, one workgroup per job: see nm_head_step_kernel. */
int nm_train_steps_head(const nm_job_t* jobs_dev, int n_jobs, int step0, int n_steps, int flags, void* stream) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1 || n_steps < 1 || step0 < 0) return -8;
  hipError_t e = hipFuncSetAttribute((const void*)nm_head_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
  if (e != hipSuccess) return (int)e;
  if ((flags & NM_F_GRADS) && n_steps != 1) return -8;
  hipLaunchKernelGGL(nm_head_step_kernel, dim3(n_jobs), dim3(WG), SMEM_BYTES, (hipStream_t)stream, jobs_dev, step0, n_steps,
                     flags & (NM_F_PROFILE | NM_F_TRACE | NM_F_GRADS | NM_F_BNSTATS));
  return (int)hipGetLastError();
}

int nm_train_steps(const nm_job_t* jobs_dev, int n_jobs, int step0, int n_steps, void* stream) {
  return nm_launch(jobs_dev, n_jobs, step0, n_steps, 1, NM_F_BACKWARD | NM_F_ADAM, stream);
}

int nm_grads(const nm_job_t* jobs_dev, int n_jobs, int step, void* stream) {
  return nm_launch(jobs_dev, n_jobs, step, 1, 1, NM_F_BACKWARD | NM_F_GRADS | NM_F_EXPORT, stream);
}

int nm_forward(const nm_job_t* jobs_dev, int n_jobs, int tile0, int n_tiles, void* stream) {
  return nm_launch(jobs_dev, n_jobs, tile0, 1, n_tiles, NM_F_EXPORT, stream);
}

int nm_train_steps_persistent(const nm_job_t* jobs_dev, int n_jobs, int step0, int n_steps, void* stream) {
  return nm_train_steps(jobs_dev, n_jobs, step0, n_steps, stream);
}

int nm_deviation(const nm_job_t* jobs_dev, int n_jobs, int tile0, int n_tiles, void* stream) {
  return nm_forward(jobs_dev, n_jobs, tile0, n_tiles, stream);
}

int nm_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                 float eps, int64_t t, void* stream) {
  if (!params || !grads || !m || !v) return -1;
  if (n <= 0 || t < 1) return -8;
  double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
  float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_flat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n, beta1,
                     beta2, eps, step_size, inv_bc2_sqrt);
  return (int)hipGetLastError();
}

int64_t nm_xb_elems(int rows_alloc, int Kx) {
  if (rows_alloc < 1 || rows_alloc % NM_BATCH != 0 || Kx < 32 || Kx % 32 != 0) return -7;
  return (int64_t)rows_alloc * ((Kx + XCH - 1) / XCH) * LDX;
}

int nm_pack_table(const float* x, const float* c, int n_rows, int rows_alloc, int D, int C, int Kx, uint16_t* xb,
                  float* x_f32_out, int x_pitch, uint16_t* cz_out, int Cz, void* stream) {
  if (!x || !xb || (C > 0 && !c)) return -1;
  if (Kx % 32 != 0 || Kx < D + C + 1 || rows_alloc < n_rows || rows_alloc % NM_BATCH != 0) return -7;
  if (x_f32_out && (x_pitch % 4 != 0 || x_pitch < D || x_pitch > Kx)) return -7;
  if (cz_out && (Cz % 8 != 0 || Cz < C + 1)) return -7;
  int64_t total = nm_xb_elems(rows_alloc, Kx);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_table_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, c, n_rows, rows_alloc, D, C,
                     Kx, xb, x_f32_out, x_pitch, cz_out, Cz);
  return (int)hipGetLastError();
}

int nm_test_gemm(int mode, const float* A, const float* B, float* Cout, int M, int N, int K, void* stream) {
  if (!A || !B || !Cout) return -1;
  if (M != ROWS || mode < 0 || mode > 3) return -8;
  if (mode == 0 && (K > PW || N > PW)) return -8;
  if (mode == 1 && (N > PW || K > 96)) return -8;
  if (mode >= 2 && (N > PW || K > PW)) return -8;
  hipError_t e = hipFuncSetAttribute((const void*)test_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(test_gemm_kernel, dim3(1), dim3(WG), SMEM_BYTES, (hipStream_t)stream, mode, A, B, Cout, M, N, K);
  return (int)hipGetLastError();
}

}  // extern "C"
