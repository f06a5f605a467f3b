// nmhip.hip -- conditional-VAE train step / forward / deviation pass for MI355X (gfx950, CDNA4).
//
// One 512-thread workgroup (8 wavefronts of 64 lanes, two per SIMD: 256 VGPRs each) owns one model ("job") and
// runs whole train steps for it: encoder MLPs -> expert fusion -> reparameterisation -> decoder
// MLPs -> Gaussian NLL + KL -> backward -> Adam, with no inter-workgroup communication.  The
// sweep fills the chip with independent jobs (one workgroup per CU), see DESIGN.md.
//
// Data placement per workgroup
//   LDS  P [256][136] bf16 : the running activation / delta of the layer chain (updated in place)
//        Q [256][136] bf16 : the other operand of the current layer (saved activation, staged
//                            x-chunk, or delta chunk of the decoder output layer)
//        S [4352] fp32     : staging slab of a weight-gradient tile group for the coalesced Adam sweep
//   HBM/L2  fp32 parameters + Adam moments in the reference's own tensor layout.  Weights are read
//           straight into MFMA fragments (fp32 -> bf16 in registers), each element once per pass;
//           Adam streams p/m/v as contiguous 16-byte-per-lane sweeps over each gradient slab.
//   workspace (L2-resident): fp32 latent statistics, bf16 activations saved for backward.
//
// Every contraction is a v_mfma_f32_16x16x32_bf16 (fp32 accumulate), issued "transposed":
// the FEATURE index of the result lives in the accumulator registers (4 consecutive features per
// lane) and the batch ROW on the lane, so every epilogue touches 8 or 16 contiguous bytes:
//   forward  out[r][n] = sum_k P[r][k] W[n][k]      A = W rows (global),   B = ds_read_b128 of P rows
//   dgrad    din[r][k] = sum_n P[r][n] W[n][k]      A = W columns (global), B = ds_read_b128 of P rows
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
constexpr int STAGE_FLOATS = 4352;   // 32 x (128 + 4) or 64 x (64 + 4) fp32
constexpr float SLOPE = 0.01f;   // F.leaky_relu default (cVAE.py:167,203)
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

struct Ctx {
  unsigned long long t_last;
  const nm_job_t* job;
  __bf16* P;
  __bf16* Q;
  float* stage;      // [STAGE_FLOATS] gradient slab
  float* red;        // [64] reduction scratch
  float* colacc;     // [128] per-column accumulators
  float* rowacc;     // [256] per-row accumulators
  unsigned long long* tlast;   // [8] last stamp per wave (NM_F_TRACE)
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
__host__ __device__ inline int kpitch(int K) { return rup(K, 8); }      // row pitch of a weight matrix in the flat buffer

// ---- workspace layout (shared by host and device) ------------------------------------------
struct WsLayout {
  int64_t mu_m, lv_m, mu_j, lv_j, es, dz, enc_act, dec_act, zc, total;
  int Zs;
};
__host__ __device__ inline WsLayout ws_layout(int M, int L, int Z) {
  WsLayout w;
  w.Zs = rup(Z, 16);
  int64_t o = 0;
  int64_t lat = (int64_t)ROWS * w.Zs * 4;
  w.mu_m = o; o += lat * M;
  w.lv_m = o; o += lat * M;
  w.mu_j = o; o += lat;
  w.lv_j = o; o += lat;
  w.es = o; o += lat;
  w.dz = o; o += lat;
  int64_t act = (int64_t)ROWS * PW * 2;
  w.enc_act = o; o += act * M * L;
  w.dec_act = o; o += act * L;
  w.zc = o; o += act;
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
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    nm_trace_cycles[w][tag] += t - c.tlast[w];
    c.tlast[w] = t;
  }
}
__device__ __forceinline__ void prof(Ctx& c, int phase) {
  if ((c.flags & NM_F_PROFILE) && blockIdx.x == 0 && blockIdx.y == 0 &&
      __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
    unsigned long long t = clock64();
    nm_prof_cycles[phase] += t - c.t_last;
    c.t_last = t;
  }
}

// Re-derive the lane/wave indices from an opaque copy of threadIdx.x.  Without this the compiler
// hoists every per-lane LDS/global address of every phase out of the persistent step loop and then
// spills hundreds of them; re-deriving per phase keeps live ranges phase-local.  The wave index goes
// through readfirstlane so that wave-level work splits compile to scalar branches.
__device__ __forceinline__ void relaunder(Ctx& c) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  c.tid = t;
  c.lane = t & 63;
  int w = __builtin_amdgcn_readfirstlane(t >> 6);
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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0): every global
// load/store in flight (weight / p-m-v / x prefetches, activation saves) would be waited for at every
// phase boundary.  Used wherever the barrier protects P / Q / the slab; hand-offs through global
// memory between threads keep __syncthreads().
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ float lrelu(float v, bool nl) { return (nl && v < 0.f) ? v * SLOPE : v; }

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

// Weight matrices live in the flat parameter buffer as [N][kpitch(K)] fp32 (rows padded to a
// multiple of 8 with zeros, every row 32-byte aligned), so a fragment is always two 16-byte loads.
// Forward weight fragment: W[n][k0 .. k0+7] -> bf16x8, zero outside.
__device__ __forceinline__ bf16x8 w_frag(gcf32 W, int N, int K, int n, int k0) {
  const int Kp = kpitch(K);
  const GAS f32x4* p = (const GAS f32x4*)(W + (int64_t)min(n, N - 1) * Kp + min(k0, Kp - 8));
  f32x4 a = p[0], b = p[1];
  const bool ok = (n < N) && (k0 < Kp);            // pad columns are zeros in memory
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = (__bf16)(ok ? a[j] : 0.f); r[4 + j] = (__bf16)(ok ? b[j] : 0.f); }
  return r;
}

// Dgrad weight fragment: W[n0 + j][k] for j = 0..7 (contraction over the OUTPUT index n).
__device__ __forceinline__ bf16x8 w_frag_t(gcf32 W, int N, int K, int n0, int k) {
  const int Kp = kpitch(K);
  bf16x8 r;
  const int kc = min(k, Kp - 1);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = W[(int64_t)min(n0 + j, N - 1) * Kp + kc];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)((n0 + j < N && k < K) ? v[j] : 0.f);
  return r;
}

// Block-wide sum; every thread gets the result.  Fixed summation order (bitwise reproducible).
__device__ __forceinline__ float block_sum(const Ctx& c, float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (c.lane == 0) c.red[c.wave] = v;
  __syncthreads();
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
__device__ __forceinline__ void apply_grad(const Ctx& c, int64_t idx, float g) {
  const nm_job_t* J = c.job;
  if (c.flags & NM_F_GRADS) asg(J->grads)[idx] = g;
  if (c.flags & NM_F_ADAM) {
    gf32 P_ = asg(J->params); gf32 M_ = asg(J->adam_m); gf32 V_ = asg(J->adam_v);
    float p = P_[idx], m = M_[idx], v = V_[idx];
    adam1(adam_consts(c), g, p, m, v);
    P_[idx] = p; M_[idx] = m; V_[idx] = v;
  }
}

// ---- cooperative copies ----------------------------------------------------------------------
// global bf16 [256][PW] (saved activation) <-> LDS [256][LDP]; only the first `width` columns move
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

// one 64-column chunk of the packed table xb into registers / into Q (pitch LDX)
constexpr int XPIECES = (ROWS * XCH / 8) / WG;     // 16-byte pieces of an x chunk per thread
struct XStage { u32x4 v[XPIECES]; };
__device__ __forceinline__ void xchunk_load(const Ctx& c, XStage& s, const GAS uint16_t* xb, int Kx, int kc) {
#pragma unroll
  for (int i = 0; i < XPIECES; ++i) {
    int p = c.tid + i * WG;            // 2048 pieces of 16 B
    int row = p >> 3, seg = p & 7;
    int col = kc * XCH + seg * 8;
    u32x4 z = {0u, 0u, 0u, 0u};
    u32x4 ld = *(const GAS u32x4*)(xb + (int64_t)(c.row0 + row) * Kx + min(col, Kx - 8));
    s.v[i] = (col < Kx) ? ld : z;
  }
}
__device__ __forceinline__ void xchunk_store(const Ctx& c, const XStage& s, __bf16* Q) {
#pragma unroll
  for (int i = 0; i < XPIECES; ++i) {
    int p = c.tid + i * WG;
    int row = p >> 3, seg = p & 7;
    *reinterpret_cast<u32x4*>(Q + row * LDX + seg * 8) = s.v[i];
  }
}

// ---- LDS-staged weight tiles ---------------------------------------------------------------------
// A block of up to 128 weight rows is contiguous in the flat buffer (row pitch = kpitch(K)), so the workgroup
// copies it with coalesced 16-byte loads, converts to bf16 and lays it out [rows][ld] in LDS; the MFMA
// A-fragments are then ds_read_b128 like the activations (dgrad: the transposing read).  Thread -> (row, piece)
// is a shift and a mask (32 pieces of 4 floats per row; lanes past the row end idle), no division.
// A [NR][128] block of a weight matrix with row pitch Kp (rows row0.., columns col0.. of the chunk walk): 32
// 16-byte pieces per row = 512 contiguous bytes per row; bf16 [NR][ld] in LDS, zeros past the matrix (rows >= N,
// columns >= Kp).  NR = 128 (8 pieces per thread) or 64 (4).
template <int NR>
struct WBlk { f32x4 v[(NR * 128 / 4) / WG]; };
template <int NR>
__device__ __forceinline__ void wblk_load(const Ctx& c, WBlk<NR>& s, gcf32 W, int N, int Kp, int row0, int col0) {
#pragma unroll
  for (int j = 0; j < (NR * 128 / 4) / WG; ++j) {
    const int p = c.tid + j * WG, row = row0 + (p >> 5), col = col0 + (p & 31) * 4;
    s.v[j] = *(const GAS f32x4*)(W + (int64_t)min(row, N - 1) * Kp + min(col, Kp - 4));
  }
}
template <int NR>
__device__ __forceinline__ void wblk_store(const Ctx& c, const WBlk<NR>& s, __bf16* dst, int ld, int N, int Kp, int row0,
                                           int col0) {
#pragma unroll
  for (int j = 0; j < (NR * 128 / 4) / WG; ++j) {
    const int p = c.tid + j * WG, lr = p >> 5, lc = (p & 31) * 4;
    const bool ok = row0 + lr < N && col0 + lc < Kp;
    bf16x4 pk;
#pragma unroll
    for (int i = 0; i < 4; ++i) pk[i] = (__bf16)(ok ? s.v[j][i] : 0.f);
    *reinterpret_cast<bf16x4*>(dst + lr * ld + lc) = pk;
  }
}

// 64-column chunk kc of the first encoder layer's weights [N][Kp]: 16 pieces of 16 bytes per row, 256
// contiguous bytes per row; bf16 [128][LDX] in LDS, zeros for rows >= N and columns >= Kp.
constexpr int W0PIECES = (128 * XCH / 4) / WG;
struct W0Stage { f32x4 v[W0PIECES]; };
__device__ __forceinline__ void w0chunk_load(const Ctx& c, W0Stage& s, gcf32 W, int N, int Kp, int kc) {
#pragma unroll
  for (int j = 0; j < W0PIECES; ++j) {
    const int p = c.tid + j * WG, row = p >> 4, col = kc * XCH + (p & 15) * 4;
    s.v[j] = *(const GAS f32x4*)(W + (int64_t)min(row, N - 1) * Kp + min(col, Kp - 4));
  }
}
__device__ __forceinline__ void w0chunk_store(const Ctx& c, const W0Stage& s, __bf16* dst, int N, int Kp, int kc) {
#pragma unroll
  for (int j = 0; j < W0PIECES; ++j) {
    const int p = c.tid + j * WG, row = p >> 4, col = kc * XCH + (p & 15) * 4;
    const bool ok = row < N && col < Kp;
    bf16x4 pk;
#pragma unroll
    for (int i = 0; i < 4; ++i) pk[i] = (__bf16)(ok ? s.v[j][i] : 0.f);
    *reinterpret_cast<bf16x4*>(dst + row * LDX + (p & 15) * 4) = pk;
  }
}

// [z | c | 1 | 0] rows of the decoder input (cVAE.py:199) into an LDS buffer.  Two branch-free
// passes (clamped, unconditional loads): the covariate/ones/pad columns from the packed table (its
// columns D .. D+C hold c | 1), then the z columns from the latent workspace.
__device__ __forceinline__ void build_zc(const Ctx& c, __bf16* dst, const nm_modality_t& md, gcf32 mu_j, gcf32 es, int Z,
                                         int C, int Zs) {
  const int wz = wpad(Z + C);
  const float rz = 1.0f / (float)Z;
  const GAS uint16_t* xb = asg(md.xb);
  // covariates and the ones column: 16-byte pieces of the packed table starting at the aligned column at or
  // below D (the row pitch Kx is a multiple of 32 and >= D + C + 1, so every piece is inside the row), scattered
  // into the unaligned destination with 2-byte LDS stores
  {
    const int c0 = md.D & ~7;
    const int npc = (md.D + C + 1 - c0 + 7) >> 3;
    const float rnp = 1.0f / (float)npc;
    for (int p = c.tid; p < ROWS * npc; p += WG) {
      const int r = idiv(p, npc, rnp), col0 = c0 + 8 * (p - r * npc);
      const u32x4 v = *(const GAS u32x4*)(xb + (int64_t)(c.row0 + r) * md.Kx + col0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = col0 + i - md.D;
        const uint16_t h = (uint16_t)(v[i >> 1] >> (16 * (i & 1)));
        if (j >= 0 && j <= C) reinterpret_cast<uint16_t*>(dst)[r * LDP + Z + j] = h;
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
#pragma unroll 4
  for (int e = c.tid; e < ROWS * Z; e += WG) {
    int r = idiv(e, Z, rz), k = e - r * Z;
    dst[r * LDP + k] = (__bf16)(mu_j[r * Zs + k] + es[r * Zs + k]);
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
// activation epilogue: P[r][f] = act(acc) for f < N, 1 at f == N (ones column), 0 beyond
__device__ __forceinline__ void act_to_P(const Ctx& c, const f32x4 (&acc)[2][RT], int N, int ntn, bool act) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int ft = c.wn + 4 * t;
    if (ft >= ntn) continue;
    int f0 = ft * 16 + 4 * c.g;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      int r = c.wm * WROWS + rt * 16 + c.c16;
      bf16x4 pk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[t][rt][i];
        v = (f0 + i < N) ? lrelu(v, act) : (f0 + i == N ? 1.0f : 0.0f);
        pk[i] = (__bf16)v;
      }
      *reinterpret_cast<bf16x4*>(c.P + r * LDP + f0) = pk;
    }
  }
}

// ---- GEMM phase: forward layer, P -> P in place ---------------------------------------------
// out[r][n] = act(sum_k P[r][k] W[n][k] + b[n]), n < N; column N := 1 (ones column feeding the
// next layer's bias gradient), columns (N, wpad(N)) := 0.  Optionally saved to `save` (bf16
// [256][PW]) for the backward pass.
__device__ __forceinline__ void fwd_layer_inplace(const Ctx& cc, gcf32 W, gcf32 b, int N, int K, bool act,
                                                  gbf16 save) {
  Ctx c = cc;
  relaunder(c);
  const int ksteps = wpad(K) / 32;       // <= 4
  const int ntn = wpad(N) / 16;
  const int Kp = kpitch(K);
  // the layer's weights: coalesced global -> registers -> bf16 tile in Q (free during the forward chain)
  WBlk<128> wsg;
  wblk_load<128>(c, wsg, W, N, Kp, 0, 0);
  tr(c, 0);
  f32x4 acc[2][RT];
  bias_acc(c, acc, b, N, 0);
  wblk_store<128>(c, wsg, c.Q, LDP, N, Kp, 0, 0);
  lds_barrier();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (ks < ksteps) {
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
  }
  tr(c, 1);
  lds_barrier();                       // every wave has finished reading P
  tr(c, 2);
  act_to_P(c, acc, N, ntn, act);
  tr(c, 3);
  lds_barrier();
  tr(c, 4);
  if (save) store_act(c, save, c.P, wpad(N));
  tr(c, 5);
}

// ---- GEMM phase: first encoder layer, x streamed through Q in 64-column chunks ----------------
__device__ __forceinline__ void fwd_first_layer(const Ctx& cc, const nm_modality_t& md, gcf32 W, gcf32 b, int N, int K,
                                                bool act, gbf16 save) {
  Ctx c = cc;
  relaunder(c);
  const int Kx = md.Kx;
  const int nch = (Kx + XCH - 1) / XCH;
  const int ntn = wpad(N) / 16;
  const int Kp = kpitch(K);
  __bf16* Wq = c.Q + ROWS * LDX;          // [128][LDX] weight chunk behind the [256][LDX] x chunk
  f32x4 acc[2][RT];
  bias_acc(c, acc, b, N, 0);
  XStage st;
  W0Stage wst;
  xchunk_load(c, st, asg(md.xb), Kx, 0);
  w0chunk_load(c, wst, W, N, Kp, 0);
  tr(c, 6);
  for (int kc = 0; kc < nch; ++kc) {
    xchunk_store(c, st, c.Q);
    w0chunk_store(c, wst, Wq, N, Kp, kc);
    tr(c, 7);
    lds_barrier();
    tr(c, 8);
    if (kc + 1 < nch) {                   // next chunk's inputs and weights fly during this chunk's MFMAs
      xchunk_load(c, st, asg(md.xb), Kx, kc + 1);
      w0chunk_load(c, wst, W, N, Kp, kc + 1);
    }
    tr(c, 9);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (kc * XCH + ks * 32 < Kx) {
        bf16x8 wf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) wf[t] = lds_frag(Wq, LDX, (c.wn + 4 * t) * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          bf16x8 a = lds_frag(c.Q, LDX, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t][rt] = mfma(wf[t], a, acc[t][rt]);
        }
      }
    }
    tr(c, 10);
    lds_barrier();
    tr(c, 11);
  }
  act_to_P(c, acc, N, ntn, act);
  lds_barrier();
  if (save) store_act(c, save, c.P, wpad(N));
  tr(c, 12);
}

// Both encoder heads as ONE LDS tile [2 Zs][LDP]: rows [0, Z) = enc_mean_layer, rows [Zs, Zs + Z) =
// enc_logvar_layer, zeros elsewhere (2 Zs <= 128 rows).  Two block copies of up to 64 rows each.
__device__ __forceinline__ void stage_heads(const Ctx& c, __bf16* dst, gcf32 Wmu, gcf32 Wlv, int Z, int K, int Zs) {
  const int Kp = kpitch(K);
  WBlk<64> a, b;
  wblk_load<64>(c, a, Wmu, Z, Kp, 0, 0);
  wblk_load<64>(c, b, Wlv, Z, Kp, 0, 0);
  // a 64-row block covers Zs <= 64 rows of each head; rows >= Z are written as zeros
  const int rows = Zs;                          // rows of each half actually used
#pragma unroll
  for (int j = 0; j < (64 * 128 / 4) / WG; ++j) {
    const int p = c.tid + j * WG, lr = p >> 5, lc = (p & 31) * 4;
    if (lr < rows) {
      const bool ok = lr < Z && lc < Kp;
      bf16x4 pa, pb;
#pragma unroll
      for (int i = 0; i < 4; ++i) { pa[i] = (__bf16)(ok ? a.v[j][i] : 0.f); pb[i] = (__bf16)(ok ? b.v[j][i] : 0.f); }
      *reinterpret_cast<bf16x4*>(dst + lr * LDP + lc) = pa;
      *reinterpret_cast<bf16x4*>(dst + (Zs + lr) * LDP + lc) = pb;
    }
  }
}

// ---- GEMM phase: encoder heads, P (= last hidden) -> fp32 mu / logvar in the workspace --------
__device__ __forceinline__ void fwd_heads(const Ctx& cc, gcf32 Wmu, gcf32 bmu, gcf32 Wlv, gcf32 blv, int Z, int K,
                                          gf32 mu_out, gf32 lv_out, int Zs) {
  Ctx c = cc;
  relaunder(c);
  const int ksteps = wpad(K) / 32;
  const int nzt = Zs / 16;
  stage_heads(c, c.Q, Wmu, Wlv, Z, K, Zs);          // Q is free during the forward chain
  lds_barrier();
  // unit = (feature tile, row tile of the wave's row half): the 4 waves of a row half share them round-robin,
  // so all 8 waves work even when the latent fits one feature tile
  for (int u = c.wn; u < nzt * RT; u += NWN) {
    const int ft = u / RT, rt = u - ft * RT;
    const int f0 = ft * 16 + 4 * c.g;
    f32x4 am, al;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x0 = bmu[min(f0 + i, Z - 1)], x1 = blv[min(f0 + i, Z - 1)];
      am[i] = (f0 + i < Z) ? x0 : 0.f;
      al[i] = (f0 + i < Z) ? x1 : 0.f;
    }
    for (int ks = 0; ks < ksteps; ++ks) {
      const bf16x8 fm = lds_frag(c.Q, LDP, ft * 16 + c.c16, ks * 32 + 8 * c.g);
      const bf16x8 fl = lds_frag(c.Q, LDP, Zs + ft * 16 + c.c16, ks * 32 + 8 * c.g);
      const bf16x8 a = lds_frag(c.P, LDP, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
      am = mfma(fm, a, am);
      al = mfma(fl, a, al);
    }
    const int r = c.wm * WROWS + rt * 16 + c.c16;
    *(GAS f32x4*)(mu_out + r * Zs + f0) = am;        // features >= Z are exactly 0 (zero weight rows, zero bias)
    *(GAS f32x4*)(lv_out + r * Zs + f0) = al;
  }
  tr(c, 13);
  __syncthreads();
  tr(c, 14);
}

// ---- dgrad: acc[k][r] += sum_n A[r][n] W[n][k]  (contraction over the columns of A) ------------
// k tiles {wn, wn+4} of wpad(K); nsteps = 32-wide steps over A's columns; n_base = index of A's
// column 0 in W's row space.
__device__ __forceinline__ void dgrad_acc(const Ctx& cc, f32x4 (&acc)[2][RT], const __bf16* A, gcf32 W, int N, int K,
                                          int nsteps, int n_base) {
  Ctx c = cc;
  relaunder(c);
  tr(c, 33);
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
// lane (c16, g) gets T[s*32 + 8g + j][ktile*16 + c16], j = 0..7 -- the fragment the 8 strided dword loads of
// w_frag_t assemble from global memory.  A = delta rows in LDS, column n_col0 + s*32 onwards.
__device__ __forceinline__ void dgrad_tile(const Ctx& c, f32x4 (&acc)[2][RT], const __bf16* A, int n_col0, const __bf16* T,
                                           int ld, int nsteps) {
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
      bf16x8 a = lds_frag(A, LDP, c.wm * WROWS + rt * 16 + c.c16, n_col0 + s * 32 + 8 * c.g);
      acc[0][rt] = mfma(wf0, a, acc[0][rt]);
      acc[1][rt] = mfma(wf1, a, acc[1][rt]);
    }
  }
}
// Hidden-layer dgrad: stage W[N][K] in Q (coalesced copy, as in the forward pass), then dgrad_tile.  On return
// the tile is still being read by other waves: the caller puts a barrier before reusing Q.
__device__ __forceinline__ void dgrad_staged(const Ctx& cc, f32x4 (&acc)[2][RT], const __bf16* A, gcf32 W, int N, int K) {
  Ctx c = cc;
  relaunder(c);
  const int Kp = kpitch(K);
  WBlk<128> wsg;
  wblk_load<128>(c, wsg, W, N, Kp, 0, 0);
  tr(c, 33);
  wblk_store<128>(c, wsg, c.Q, LDP, N, Kp, 0, 0);
  lds_barrier();
  dgrad_tile(c, acc, A, 0, c.Q, LDP, wpad(N) / 32);
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
        if (act && !((float)a[i] > 0.f)) d *= SLOPE;
        if (k0 + i >= K) d = 0.f;
        pk[i] = (__bf16)d;
      }
      *reinterpret_cast<bf16x4*>(c.P + r * LDP + k0) = pk;
    }
  }
}

// ---- wgrad + Adam ------------------------------------------------------------------------------
// dW[n][k] = sum_r A[r][a_col0 + n] * B[r][kk], n in [0,N), B columns kk in [0, ncols) map to the
// weight column k = k_base + kk; k < K is W[n][k], k == K the bias b[n] (ones column), beyond:
// nothing.  The output is produced in slabs of SR rows x SC columns (32 x 128 or 64 x 64, = one
// 16-byte parameter group per thread).  Per slab every thread first ISSUES the loads of its group's
// p/m/v (so their HBM latency hides under the MFMA work), the 16 waves then write their accumulator
// tiles (4 consecutive k per lane) into the LDS slab S, and after one barrier every thread applies
// Adam to its group with 16-byte stores: the sweep walks contiguous parameter memory.
template <bool SCALAR_TR>
__device__ __forceinline__ void wgrad_adam(const Ctx& cc, const __bf16* A, int lda, int a_col0, const __bf16* B, int ldb,
                                           int N, int K, int k_base, int ncols, int64_t w_off, int64_t b_off) {
  Ctx c = cc;
  relaunder(c);
  const nm_job_t* J = c.job;
  const bool wide = ncols > 64;                 // slab shape
  const int SR = wide ? 32 : 64, SP = (wide ? 128 : 64) + 4;      // slab rows, slab pitch (floats)
  const int Kp = kpitch(K);
  const int nkt = (ncols + 15) / 16;            // k tiles in this pass
  const int gpr = max(0, min(ncols, Kp - k_base)) >> 2;          // 16-byte parameter groups per row in this pass
  const float rg = gpr > 0 ? 1.0f / (float)gpr : 0.f;
  const bool has_bias = (b_off >= 0) && (K >= k_base) && (K < k_base + ncols);   // b_off < 0: no bias column in B
  const bool do_adam = (c.flags & NM_F_ADAM) != 0;
  const AdamK ak = adam_consts(c);
  constexpr int NG = 1024 / WG;                 // 16-byte parameter groups per thread and slab
  for (int n0 = 0; n0 < N; n0 += SR) {
    const int nr = min(SR, N - n0);
    const int nts = (nr + 15) / 16;
    // ---- this thread's parameter groups: issue the p/m/v loads now ----
    bool mine[NG];
    int qn[NG], qk[NG];
    int64_t pidx[NG];
    f32x4 pv[NG], mv[NG], vv[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int q = c.tid + j * WG;
      mine[j] = q < nr * gpr;
      qn[j] = mine[j] ? idiv(q, gpr, rg) : 0;
      qk[j] = mine[j] ? (q - qn[j] * gpr) * 4 : 0;                              // column inside the pass
      pidx[j] = w_off + (int64_t)(n0 + qn[j]) * Kp + k_base + qk[j];            // 16-byte aligned
      pv[j] = f32x4{0.f, 0.f, 0.f, 0.f}; mv[j] = pv[j]; vv[j] = pv[j];
      if (do_adam && mine[j]) {
        pv[j] = *(const GAS f32x4*)(asg(J->params) + pidx[j]);
        // the moments are touched once per step: streaming (nt) accesses keep them from evicting the weights
        // and activations that are re-read within the step (measured: -9 % config A, -2.5 % 3-modality)
        mv[j] = __builtin_nontemporal_load((const GAS f32x4*)(asg(J->adam_m) + pidx[j]));
        vv[j] = __builtin_nontemporal_load((const GAS f32x4*)(asg(J->adam_v) + pidx[j]));
      }
    }
    // bias element of this thread (first nr threads): its p/m/v fly with the slab's groups
    const bool bmine = has_bias && c.tid < nr;
    const int64_t bidx = b_off + n0 + (bmine ? c.tid : 0);
    float bp = 0.f, bm = 0.f, bv = 0.f;
    if (do_adam && bmine) { bp = asg(J->params)[bidx]; bm = asg(J->adam_m)[bidx]; bv = asg(J->adam_v)[bidx]; }
    tr(c, 26);
    // ---- tiles -> slab: a unit is one k tile x two n tiles (the k-side fragment is shared) ----
    const int npairs = (nts + 1) / 2;
    for (int u = c.wave; u < nkt * npairs; u += NWAVES) {
      const int kt = u % nkt, np = u / nkt;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const int ncol0 = a_col0 + n0 + np * 32;
      if (SCALAR_TR) {
        for (int rs = 0; rs < ROWS / 32; ++rs) {
          bf16x8 akf = lds_frag_tr_scalar(B, ldb, rs * 32, kt * 16, c.lane);
          acc0 = mfma(akf, lds_frag_tr_scalar(A, lda, rs * 32, ncol0, c.lane), acc0);
          acc1 = mfma(akf, lds_frag_tr_scalar(A, lda, rs * 32, ncol0 + 16, c.lane), acc1);
        }
      } else {
        unsigned na = tr_addr_il(A, lda, 0, ncol0, c.lane);
        unsigned ka = tr_addr_il(B, ldb, 0, kt * 16, c.lane);
        const unsigned n_step = 32u * lda * 2u, k_step = 32u * ldb * 2u;
        const unsigned n4 = 1u * lda * 2u, k4 = 1u * ldb * 2u;         // second read of a pair: the odd rows
#pragma unroll 2
        for (int rs = 0; rs < ROWS / 32; rs += 2) {
          // two row steps per wait: 12 transposing reads in flight (k side once, two n tiles)
          bf16x4 k0v, k1v, k2v, k3v, a0, a1, a2, a3, b0, b1, b2, b3;
          unsigned na1 = na + n4, ka1 = ka + k4, na2 = na + n_step, ka2 = ka + k_step;
          unsigned na3 = na2 + n4, ka3 = ka2 + k4;
          NM_TR_READ(k0v, ka, 0);  NM_TR_READ(k1v, ka1, 0);
          NM_TR_READ(a0, na, 0);   NM_TR_READ(a1, na1, 0);
          NM_TR_READ(b0, na, 32);  NM_TR_READ(b1, na1, 32);
          NM_TR_READ(k2v, ka2, 0); NM_TR_READ(k3v, ka3, 0);
          NM_TR_READ(a2, na2, 0);  NM_TR_READ(a3, na3, 0);
          NM_TR_READ(b2, na2, 32); NM_TR_READ(b3, na3, 32);
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(k0v), "+v"(k1v), "+v"(k2v), "+v"(k3v), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0),
                         "+v"(b1), "+v"(b2), "+v"(b3));
          bf16x8 kf0 = join4(k0v, k1v), kf1 = join4(k2v, k3v);
          acc0 = mfma(kf0, join4(a0, a1), acc0);
          acc1 = mfma(kf0, join4(b0, b1), acc1);
          acc0 = mfma(kf1, join4(a2, a3), acc0);
          acc1 = mfma(kf1, join4(b2, b3), acc1);
          na += 2 * n_step; ka += 2 * k_step;
        }
      }
      // lane holds dW[n = (2 np + h)*16 + c16][kk = kt*16 + 4g .. +3]
      *reinterpret_cast<f32x4*>(c.stage + (np * 32 + c.c16) * SP + kt * 16 + 4 * c.g) = acc0;
      if (np * 2 + 1 < nts) *reinterpret_cast<f32x4*>(c.stage + (np * 32 + 16 + c.c16) * SP + kt * 16 + 4 * c.g) = acc1;
    }
    tr(c, 27);
    lds_barrier();
    tr(c, 28);
    // ---- Adam on this thread's group; bias column by the first nr threads ----
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      if (mine[j]) {
        f32x4 g = *reinterpret_cast<const f32x4*>(c.stage + qn[j] * SP + qk[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = (k_base + qk[j] + i < K) ? g[i] : 0.f;      // pad columns keep zero gradient
        if (c.flags & NM_F_GRADS) *(GAS f32x4*)(asg(J->grads) + pidx[j]) = g;
        if (do_adam) {
          f32x4 p4 = pv[j], m4 = mv[j], v4 = vv[j];
#pragma unroll
          for (int i = 0; i < 4; ++i) { float pp = p4[i], mm = m4[i], v2 = v4[i]; adam1(ak, g[i], pp, mm, v2); p4[i] = pp; m4[i] = mm; v4[i] = v2; }
          *(GAS f32x4*)(asg(J->params) + pidx[j]) = p4;
          __builtin_nontemporal_store(m4, (GAS f32x4*)(asg(J->adam_m) + pidx[j]));
          __builtin_nontemporal_store(v4, (GAS f32x4*)(asg(J->adam_v) + pidx[j]));
        }
      }
    }
    if (bmine) {
      const float bg = c.stage[c.tid * SP + (K - k_base)];
      if (c.flags & NM_F_GRADS) asg(J->grads)[bidx] = bg;
      if (do_adam) {
        adam1(ak, bg, bp, bm, bv);
        asg(J->params)[bidx] = bp; asg(J->adam_m)[bidx] = bm; asg(J->adam_v)[bidx] = bv;
      }
    }
    tr(c, 29);
    lds_barrier();
    tr(c, 30);
  }
}

// ---- expert fusion (cVAE.py:1144-1164) on one (row, z) element --------------------------------
__device__ __forceinline__ int experts(const nm_job_t* J) { return J->M_enc > 0 ? J->M_enc : J->M; }
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
      float w = (cb == NM_COMBINE_GPOE) ? al[m] / var : 1.0f / var;
      S += w; Smu += L.mu[m] * w;
      sm += L.mu[m]; sv += var;
    }
  }
  if (cb == NM_COMBINE_MOE) { f.mu = sm / M; f.var = sv / M; }
  else {
    f.mu = Smu / S; f.var = 1.0f / S;
    if (cb == NM_COMBINE_MOPOE) { f.mu = (sm + f.mu) / (M + 1); f.var = (sv + f.var) / (M + 1); }
  }
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
      float w = expf(-L.lv[m]) * ((cb == NM_COMBINE_GPOE) ? al[m] : 1.0f);
      S += w; Smu += L.mu[m] * w;
      sv += expf(L.lv[m]);
    }
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
template <bool SCALAR_TR>
__device__ __forceinline__ void run_step(Ctx& c, int step) {
  const nm_job_t* J = c.job;
  const int M = J->M, L = J->L, Z = J->Z, C = J->C;
  const int Me = experts(J);                    // modalities that have an encoder
  const bool nl = J->non_linear != 0;
  const bool bwd = (c.flags & NM_F_BACKWARD) != 0;
  const bool exportf = (c.flags & NM_F_EXPORT) != 0;
  const WsLayout wl = ws_layout(M, L, Z);
  const int Zs = wl.Zs;
  const float rZ = 1.0f / (float)Z;
  gf32 ws_mu_m = (gf32)(c.ws + wl.mu_m);
  gf32 ws_lv_m = (gf32)(c.ws + wl.lv_m);
  gf32 ws_mu_j = (gf32)(c.ws + wl.mu_j);
  gf32 ws_lv_j = (gf32)(c.ws + wl.lv_j);
  gf32 ws_es = (gf32)(c.ws + wl.es);
  gf32 ws_dz = (gf32)(c.ws + wl.dz);
  gbf16 ws_enc = (gbf16)(c.ws + wl.enc_act);
  gbf16 ws_dec = (gbf16)(c.ws + wl.dec_act);
  gbf16 ws_zc = (gbf16)(c.ws + wl.zc);
  gcf32 prm = asg(J->params);

  // ================= encoders =================
  for (int m = 0; m < Me; ++m) {
    relaunder(c);
    const nm_modality_t& md = J->mod[m];
    gbf16 save0 = bwd ? ws_enc + (int64_t)(m * L + 0) * ROWS * PW : (gbf16)nullptr;
    fwd_first_layer(c, md, prm + md.enc_w[0], prm + md.enc_b[0], J->H[0], md.D + C, nl, save0);
    prof(c, PH_ENC_L0);
    for (int e = 1; e < L; ++e) {
      gbf16 sv = bwd ? ws_enc + (int64_t)(m * L + e) * ROWS * PW : (gbf16)nullptr;
      fwd_layer_inplace(c, prm + md.enc_w[e], prm + md.enc_b[e], J->H[e], J->H[e - 1], nl, sv);
    }
    prof(c, PH_ENC_REST);
    fwd_heads(c, prm + md.mu_w, prm + md.mu_b, prm + md.lv_w, prm + md.lv_b, Z, J->H[L - 1],
              ws_mu_m + (int64_t)m * ROWS * Zs, ws_lv_m + (int64_t)m * ROWS * Zs, Zs);
    prof(c, PH_HEADS);
  }

  // ================= fusion + reparameterisation + KL =================
  float al[NM_MAX_EXP] = {0.f, 0.f, 0.f, 0.f};
  if (J->combine == NM_COMBINE_GPOE && !(Me == 1 && J->single_bypass)) softmax_alpha(J, al);
  auto load_lat = [&](Lat& Lt, int r, int z) {
#pragma unroll
    for (int m = 0; m < NM_MAX_EXP; ++m) {
      Lt.mu[m] = (m < Me) ? ws_mu_m[((int64_t)m * ROWS + r) * Zs + z] : 0.f;
      Lt.lv[m] = (m < Me) ? ws_lv_m[((int64_t)m * ROWS + r) * Zs + z] : 0.f;
    }
  };
  float kl_part = 0.f;
  relaunder(c);
#pragma unroll 2
  for (int e = c.tid; e < ROWS * Z; e += WG) {
    int r = idiv(e, Z, rZ), z = e - r * Z;
    Lat Lt;
    load_lat(Lt, r, z);
    Fuse f = fuse_fwd(J, Lt, al);
    float ep = J->eps ? asg(J->eps)[((int64_t)(step % J->eps_cap) * ROWS + r) * Z + z]
                      : randn_ctr(J->seed, (uint32_t)step, (uint32_t)(c.row0 + r), (uint32_t)z);
    float es = ep * expf(0.5f * f.lv);
    if (c.flags & NM_F_ZGIVEN) { f.mu = ep; es = 0.f; }     // decode(z, c, m): the draw buffer holds z itself
    float zz = f.mu + es;
    ws_mu_j[r * Zs + z] = f.mu;
    ws_lv_j[r * Zs + z] = f.lv;
    ws_es[r * Zs + z] = es;
    ws_dz[r * Zs + z] = 0.f;
    if (r < c.nrows) {
      kl_part += -0.5f * (1.0f + f.lv - f.mu * f.mu - expf(f.lv));
      if (exportf) {
        int64_t gr = (int64_t)(c.row0 + r) * Z + z;
        if (J->out_mu) asg(J->out_mu)[gr] = f.mu;
        if (J->out_logvar) asg(J->out_logvar)[gr] = f.lv;
        if (J->out_z) asg(J->out_z)[gr] = zz;
      }
    }
  }
  tr(c, 15);
  float kl = block_sum(c, kl_part) * c.inv_b;          // calc_kl: sum over z, mean over rows
  tr(c, 22);
  prof(c, PH_LATENT);

  // ================= decoders (forward, NLL, and the whole decoder backward) =================
  float ll_sum = 0.f;
  for (int m = 0; m < M; ++m) {
    relaunder(c);
    const nm_modality_t& md = J->mod[m];
    const int D = md.D;
    const int Kd0 = Z + C;
    // z | c | 1: built by the first decoder; the others reuse it when all tables carry the same covariates
    const bool reuse_zc = J->shared_cov && M > 1;
    if (m == 0 || !reuse_zc) {
      build_zc(c, c.P, md, ws_mu_j, ws_es, Z, C, Zs);
      tr(c, 23);
      lds_barrier();
      if (bwd || reuse_zc) store_act(c, ws_zc, c.P, wpad(Kd0));
    } else {
      __syncthreads();                             // the first decoder's copy is complete in memory
      load_act(c, c.P, ws_zc, wpad(Kd0));
      tr(c, 23);
      lds_barrier();
    }
    tr(c, 24);
    prof(c, PH_DEC_ZC);
    // --- hidden decoder layers ---
    for (int d = 0; d < L; ++d) {
      int Kin = (d == 0) ? Kd0 : J->H[L - d];
      int Nout = J->H[L - 1 - d];
      gbf16 sv = (bwd && d < L - 1) ? ws_dec + (int64_t)d * ROWS * PW : (gbf16)nullptr;
      fwd_layer_inplace(c, prm + md.dec_w[d], prm + md.dec_b[d], Nout, Kin, nl, sv);
    }
    prof(c, PH_DEC_HID);
    // --- output layer in chunks of 128 ROI columns, fused with NLL, its backward and Adam ---
    const int Hl = J->H[0];                       // width feeding the output layer
    gcf32 Wo = prm + md.out_w;
    gcf32 bo = prm + md.out_b;
    gcf32 lvo = prm + md.logvar_out;
    gcf32 xf = asg(md.x_f32);
    const int xp = md.x_pitch;
    f32x4 accg[2][RT];
    zero_acc(accg);
    float nll_part = 0.f;
    if (exportf && md.out_rowdev) { for (int r = c.tid; r < ROWS; r += WG) c.rowacc[r] = 0.f; }
    const int nchunks = (D + PW - 1) / PW;
    // read once, outside the per-lane selects below: a descriptor load inside `cond ? load * x : 0` becomes a
    // lane-divergent branch, and register spills placed around such branches are not safe with this compiler
    // (tools/check_spill_exec.py)
    const float llw_b = J->ll_weight * c.inv_b;
    for (int ch = 0; ch < nchunks; ++ch) {
      relaunder(c);
      const int d0 = ch * PW;
      const int valid = min(PW, D - d0);
      tr(c, 16);
      if (c.tid < PW) c.colacc[c.tid] = 0.f;
      // x_hat chunk: acc[d][r]
      {
        const int ksteps = wpad(Hl) / 32;
        // everything this chunk reads from memory is requested first: the chunk's weight rows (contiguous:
        // coalesced copy into a bf16 tile in Q, free until the epilogue), the fp32 inputs of the residual
        // (rows are always inside the zero-padded table) and logvar_out
        const int Kpo = kpitch(Hl);
        WBlk<128> wsg;
        wblk_load<128>(c, wsg, Wo + (int64_t)d0 * Kpo, valid, Kpo, 0, 0);
        f32x4 xin[RT];                     // inputs of feature tile t = 0 now, t = 1 after tile 0's epilogue
        float sv[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int dg0 = d0 + (c.wn + 4 * t) * 16 + 4 * c.g;
#pragma unroll
          for (int i = 0; i < 4; ++i) sv[t][i] = lvo[min(dg0 + i, D - 1)];
        }
        {
          const int dcl = min(d0 + c.wn * 16 + 4 * c.g, xp - 4);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            xin[rt] = *(const GAS f32x4*)(xf + (int64_t)(c.row0 + c.wm * WROWS + rt * 16 + c.c16) * xp + dcl);
        }
        prof(c, PH_X_LOADS);
        tr(c, 18);
        f32x4 acc[2][RT];
        bias_acc(c, acc, bo, D, d0);
        wblk_store<128>(c, wsg, c.Q, LDP, valid, Kpo, 0, 0);
        lds_barrier();
        tr(c, 17);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks < ksteps) {
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
        }
        lds_barrier();                     // the weight tile is fully read: the epilogue (delta) or the next chunk's tile may overwrite Q
        prof(c, PH_X_MFMA);
        tr(c, 19);
        // epilogue: residual, NLL, d logvar_out, delta chunk -> Q.  Lane: 4 consecutive ROI of one row.
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int dl0 = (c.wn + 4 * t) * 16 + 4 * c.g;       // first of the lane's 4 columns inside the chunk
          const int dg0 = d0 + dl0;
          // per column: q = sum_r diff^2 (valid rows only).  Then  NLL = sum_d [0.5 e^{-s} q + n (0.5 s + log sqrt(2 pi))]
          // and d(-LL)/d s_d = (0.5 n - 0.5 e^{-s} q) / B: one masked square-accumulate per element instead of
          // evaluating both sums element by element.
          float inv[4], colq[4], coef[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            inv[i] = expf(-sv[t][i]);
            colq[i] = 0.f;
            coef[i] = (dg0 + i < D) ? llw_b * inv[i] : 0.f;                      // d total / d x_hat = coef * diff
          }
          f32x4 xcur[RT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) xcur[rt] = xin[rt];
          if (t == 0) {                      // request tile 1's inputs while tile 0 is processed
            const int dcl = min(d0 + (c.wn + 4) * 16 + 4 * c.g, xp - 4);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              xin[rt] = *(const GAS f32x4*)(xf + (int64_t)(c.row0 + c.wm * WROWS + rt * 16 + c.c16) * xp + dcl);
          }
          int nvalid = 0;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const int r = c.wm * WROWS + rt * 16 + c.c16;
            const bool rv = r < c.nrows;
            nvalid += rv ? 1 : 0;
            bf16x4 pk;
            f32x4 ex = {0.f, 0.f, 0.f, 0.f};
            if (md.dloc_extra)               // extra loss gradient on x_hat (regression head)
              ex = *(const GAS f32x4*)(asg(md.dloc_extra) + (int64_t)(c.row0 + r) * xp + min(dg0, xp - 4));
            float rc = 0.f;
            if (md.dloc_rowcoef) rc = asg(md.dloc_rowcoef)[c.row0 + r];      // contrastive hinge: rc * (x_hat - x)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float diff = rv ? acc[t][rt][i] - xcur[rt][i] : 0.f;
              colq[i] = fmaf(diff, diff, colq[i]);
              const bool dv = dg0 + i < D;
              pk[i] = (__bf16)(diff * (coef[i] + (dv ? rc : 0.f)) + ((rv && dv) ? ex[i] : 0.f));
            }
            if (bwd) *reinterpret_cast<bf16x4*>(c.Q + r * LDP + dl0) = pk;
          }
          float colsum[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool dv = dg0 + i < D;
            const float hq = 0.5f * inv[i] * colq[i];
            nll_part += dv ? hq + (float)nvalid * (0.5f * sv[t][i] + LOG_SQRT_2PI) : 0.f;
            colsum[i] = dv ? 0.5f * (float)nvalid - hq : 0.f;
          }
          if (exportf && dg0 < xp) {           // exports share the fp32 table's row pitch: one 16-byte store each
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
              const int r = c.wm * WROWS + rt * 16 + c.c16;
              if (r < c.nrows) {
                f32x4 lo, sq;
                float rs = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const bool dv = dg0 + i < D;
                  const float xh = acc[t][rt][i], diff = xh - xcur[rt][i];
                  lo[i] = dv ? xh : 0.f;
                  sq[i] = dv ? diff * diff : 0.f;
                  rs += sq[i];
                }
                const int64_t gi = (int64_t)(c.row0 + r) * xp + dg0;
                if (md.out_loc) __builtin_nontemporal_store(lo, (GAS f32x4*)(asg(md.out_loc) + gi));       // written once,
                if (md.out_sqerr) __builtin_nontemporal_store(sq, (GAS f32x4*)(asg(md.out_sqerr) + gi));   // read elsewhere
                if (md.out_rowdev) atomicAdd(&c.rowacc[r], rs);
              }
            }
          }
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
      }
      prof(c, PH_X_EPI);
      tr(c, 20);
      if (!bwd) continue;
      // dgrad into the last hidden activation: accg[k][r] += sum_d Q[r][d] Wo[d0 + d][k].  The chunk's weight
      // rows come back from L2 (the forward tile was copied from them) as two 64-row half tiles through the
      // gradient slab, read with the transposing LDS load; both halves are requested now.
      relaunder(c);
      __bf16* Th = reinterpret_cast<__bf16*>(c.stage);
      const int Kph = kpitch(Hl);
      const int rows_a = min(valid, 64), rows_b = valid - rows_a;
      WBlk<64> ha, hb;
      wblk_load<64>(c, ha, Wo + (int64_t)d0 * Kph, valid, Kph, 0, 0);
      wblk_load<64>(c, hb, Wo + (int64_t)d0 * Kph, valid, Kph, 64, 0);      // rows clamp to the chunk's last row
      wblk_store<64>(c, ha, Th, LDP, valid, Kph, 0, 0);
      lds_barrier();                              // delta chunk in Q and half tile A complete
      tr(c, 21);
      prof(c, PH_OUT_GEMM);
      // d logvar_out for this chunk
      if (c.tid < valid) apply_grad(c, md.logvar_out + d0 + c.tid, J->ll_weight * c.colacc[c.tid] * c.inv_b);
      prof(c, PH_OUT_DLV);
      dgrad_tile(c, accg, c.Q, 0, Th, LDP, rup(rows_a, 32) / 32);
      if (rows_b > 0) {
        lds_barrier();                            // half A fully read
        wblk_store<64>(c, hb, Th, LDP, valid, Kph, 64, 0);
        lds_barrier();
        dgrad_tile(c, accg, c.Q, 64, Th, LDP, rup(rows_b, 32) / 32);
      }
      tr(c, 34);
      lds_barrier();                              // the slab is free again; old Wo fully read
      tr(c, 35);
      prof(c, PH_OUT_DGRAD);
      // wgrad + Adam of this chunk of decoder_mean_layer: dWo[d][k] = sum_r Q[r][d] P[r][k]
      wgrad_adam<SCALAR_TR>(c, c.Q, LDP, 0, c.P, LDP, valid, Hl, 0, rup(Hl + 1, 16),
                            md.out_w + (int64_t)d0 * kpitch(Hl), md.out_b + d0);
      prof(c, PH_OUT_WGRAD);
    }
    float nll = block_sum(c, nll_part);
    const float ll_this = -nll * c.inv_b;           // compute_ll: sum over ROI, mean over rows
    ll_sum += ll_this;
    if (c.tid == 0 && J->loss_log)
      asg(J->loss_log)[(int64_t)(step % J->loss_cap) * NM_LOSS_STRIDE + NM_LOSS_LL_M + m] = ll_this;
    prof(c, PH_NLL_RED);
    if (exportf && md.out_rowdev) {
      lds_barrier();
      for (int r = c.tid; r < c.nrows; r += WG) asg(md.out_rowdev)[c.row0 + r] = c.rowacc[r] / (float)D;
    }
    if (!bwd) { __syncthreads(); continue; }

    // --- decoder hidden layers, backward ---
    // state: P = activation g_{L-1}, accg = pre-mask delta of g_{L-1}
    tr(c, 43);
    finish_delta(c, accg, c.P, Hl, nl);             // mask source is P itself (same element)
    tr(c, 44);
    lds_barrier();
    tr(c, 45);
    prof(c, PH_DEC_FINISH);
    for (int d = L - 1; d >= 0; --d) {
      relaunder(c);
      int Kin = (d == 0) ? Kd0 : J->H[L - d];
      int Nout = J->H[L - 1 - d];
      // dgrad from the layer's weight tile staged in Q, then Q <- input activation of decoder layer d
      f32x4 acc[2][RT];
      zero_acc(acc);
      dgrad_staged(c, acc, c.P, prm + md.dec_w[d], Nout, Kin);
      tr(c, 34);
      lds_barrier();                              // weight tile fully read
      tr(c, 35);
      prof(c, PH_DEC_DGRAD);
      tr(c, 31);
      load_act(c, c.Q, d == 0 ? ws_zc : ws_dec + (int64_t)(d - 1) * ROWS * PW, wpad(Kin));
      tr(c, 32);
      lds_barrier();
      tr(c, 36);
      prof(c, PH_DEC_LOAD);
      wgrad_adam<SCALAR_TR>(c, c.P, LDP, 0, c.Q, LDP, Nout, Kin, 0, rup(Kin + 1, 16), md.dec_w[d], md.dec_b[d]);
      prof(c, PH_DEC_WGRAD);
      if (d > 0) {
        finish_delta(c, acc, c.Q, Kin, nl);
      } else {
        // d z: accumulate over decoders (fixed element -> thread ownership, no race)
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
              GAS f32x4* p = (GAS f32x4*)(ws_dz + r * Zs + k0);    // columns >= Z of the row are never read
              *p = *p + acc[t][rt];
            }
          }
        }
      }
      __syncthreads();
      prof(c, PH_DEC_DELTA);
    }
  }

  // ================= loss log =================
  if (c.tid == 0 && J->loss_log) {
    gf32 row = asg(J->loss_log) + (int64_t)(step % J->loss_cap) * NM_LOSS_STRIDE;
    row[NM_LOSS_KL] = J->kl_weight * kl;
    row[NM_LOSS_LL] = ll_sum;
    row[NM_LOSS_TOTAL] = J->kl_weight * kl - J->ll_weight * ll_sum;
  }
  if (!bwd) return;
  prof(c, PH_ALPHA);

  // ================= fusion backward: alpha gradients (gPoE) =================
  const bool fused = !(Me == 1 && J->single_bypass);
  const float klw = J->kl_weight * c.inv_b;
  // With several experts the fusion backward (8 exponentials per element) is evaluated ONCE: the deltas of every
  // expert go side by side into Q (expert m in columns [m 2Zs, (m+1) 2Zs) = [d mu_m | d logvar_m]), from there into
  // the (dead) z|c slot of the workspace, and each encoder's backward below starts from a 16-byte copy of its
  // columns.  Falls back to one evaluation per encoder when the deltas do not fit in 128 columns.
  const bool once = fused && Me >= 2 && Me * 2 * Zs <= PW;
  if (once || (fused && J->combine == NM_COMBINE_GPOE)) {
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
    for (int e = c.tid; e < ROWS * Z; e += WG) {
      int r = idiv(e, Z, rZ), z = e - r * Z;
      Lat Lt;
      load_lat(Lt, r, z);
      float mj = ws_mu_j[r * Zs + z], lj = ws_lv_j[r * Zs + z], es = ws_es[r * Zs + z], dz = ws_dz[r * Zs + z];
      if (J->dz_extra) dz += asg(J->dz_extra)[(int64_t)(c.row0 + r) * Z + z];
      float dmu_j = dz + klw * mj;
      float dlv_j = 0.5f * dz * es + klw * 0.5f * (expf(lj) - 1.0f);
      FuseGrad G = fuse_bwd(J, Lt, al, dmu_j, dlv_j);
      const bool rv = r < c.nrows;
#pragma unroll
      for (int m = 0; m < NM_MAX_EXP; ++m) {
        dal[m] += rv ? G.dal[m] : 0.f;
        if (once && m < Me) {
          c.Q[r * LDP + m * 2 * Zs + z] = (__bf16)(rv ? G.dmu[m] : 0.f);
          c.Q[r * LDP + m * 2 * Zs + Zs + z] = (__bf16)(rv ? G.dlv[m] : 0.f);
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
          if (m < Me) apply_grad(c, J->mod[m].alpha, al[m] * (tot[m] - dot));   // softmax backward
      }
    }
    lds_barrier();
    if (once) store_act(c, ws_zc, c.Q, Me * 2 * Zs);
    __syncthreads();
  }

  // ================= encoders, backward =================
  for (int m = 0; m < Me; ++m) {
    relaunder(c);
    const nm_modality_t& md = J->mod[m];
    const int Hh = J->H[L - 1];
    const int whp = rup(2 * Zs, 32);
    const float rwh = 1.0f / (float)whp;
    // P <- [d mu_m | 0 | d logvar_m | 0], Q <- last hidden activation
    if (once) {                                    // this expert's columns of the saved fusion backward
      const int segs = (2 * Zs) >> 3;              // 16-byte pieces per row
      const float rs_ = 1.0f / (float)segs;
      for (int p_ = c.tid; p_ < ROWS * segs; p_ += WG) {
        const int row = idiv(p_, segs, rs_), seg = p_ - row * segs;
        *reinterpret_cast<u32x4*>(c.P + row * LDP + seg * 8) = *(const GAS u32x4*)(ws_zc + row * PW + m * 2 * Zs + seg * 8);
      }
    } else {
      const int npad = whp - 2 * Z;
      const float rnp = npad > 0 ? 1.0f / (float)npad : 0.f;
      for (int e = c.tid; e < ROWS * npad; e += WG) {
        int r = idiv(e, npad, rnp), j = e - r * npad;
        int k = (j < Zs - Z) ? Z + j : Zs + Z + (j - (Zs - Z));
        c.P[r * LDP + k] = (__bf16)0.0f;
      }
#pragma unroll 2
      for (int e = c.tid; e < ROWS * Z; e += WG) {
        int r = idiv(e, Z, rZ), z = e - r * Z;
        Lat Lt;
        load_lat(Lt, r, z);
        float mj = ws_mu_j[r * Zs + z], lj = ws_lv_j[r * Zs + z], es = ws_es[r * Zs + z], dz = ws_dz[r * Zs + z];
        if (J->dz_extra) dz += asg(J->dz_extra)[(int64_t)(c.row0 + r) * Z + z];
        float dmu_j = dz + klw * mj;
        float dlv_j = 0.5f * dz * es + klw * 0.5f * (expf(lj) - 1.0f);
        FuseGrad G = fuse_bwd(J, Lt, al, dmu_j, dlv_j);
        const bool rv = r < c.nrows;
        c.P[r * LDP + z] = (__bf16)(rv ? pick(G.dmu, m) : 0.f);
        c.P[r * LDP + Zs + z] = (__bf16)(rv ? pick(G.dlv, m) : 0.f);
      }
    }
    tr(c, 37);
    // dgrad through both heads from one staged tile (rows = [d mu | d logvar] columns of P), then Q <- activation
    stage_heads(c, c.Q, prm + md.mu_w, prm + md.lv_w, Z, Hh, Zs);
    lds_barrier();
    prof(c, PH_ENCB_PREP);
    f32x4 acc[2][RT];
    zero_acc(acc);
    dgrad_tile(c, acc, c.P, 0, c.Q, LDP, (2 * Zs) / 32);
    lds_barrier();                                  // tile fully read
    prof(c, PH_ENCB_HEADS_DGRAD);
    load_act(c, c.Q, ws_enc + (int64_t)(m * L + (L - 1)) * ROWS * PW, wpad(Hh));
    tr(c, 38);
    lds_barrier();
    tr(c, 39);
    wgrad_adam<SCALAR_TR>(c, c.P, LDP, 0, c.Q, LDP, Z, Hh, 0, rup(Hh + 1, 16), md.mu_w, md.mu_b);
    wgrad_adam<SCALAR_TR>(c, c.P, LDP, Zs, c.Q, LDP, Z, Hh, 0, rup(Hh + 1, 16), md.lv_w, md.lv_b);
    prof(c, PH_ENCB_HEADS_WGRAD);
    finish_delta(c, acc, c.Q, Hh, nl);              // P = delta of h_{L-1}
    lds_barrier();
    prof(c, PH_ENCB_DELTA);
    for (int e = L - 1; e >= 1; --e) {
      int Kin = J->H[e - 1], Nout = J->H[e];
      zero_acc(acc);
      dgrad_staged(c, acc, c.P, prm + md.enc_w[e], Nout, Kin);
      lds_barrier();                              // weight tile fully read
      prof(c, PH_ENCB_DGRAD);
      load_act(c, c.Q, ws_enc + (int64_t)(m * L + (e - 1)) * ROWS * PW, wpad(Kin));
      lds_barrier();
      prof(c, PH_ENCB_LOAD);
      wgrad_adam<SCALAR_TR>(c, c.P, LDP, 0, c.Q, LDP, Nout, Kin, 0, rup(Kin + 1, 16), md.enc_w[e], md.enc_b[e]);
      prof(c, PH_ENCB_WGRAD);
      finish_delta(c, acc, c.Q, Kin, nl);
      lds_barrier();
      prof(c, PH_ENCB_DELTA);
    }
    // first encoder layer: dW[n][k] = sum_r P[r][n] xc[r][k], x streamed through Q
    {
      const int Kx = md.Kx, K0 = md.D + C, N0 = J->H[0];
      const int nch = (Kx + XCH - 1) / XCH;
      XStage st;
      xchunk_load(c, st, asg(md.xb), Kx, 0);
      for (int kc = 0; kc < nch; ++kc) {
        tr(c, 40);
        xchunk_store(c, st, c.Q);
        tr(c, 41);
        lds_barrier();
        tr(c, 42);
        if (kc + 1 < nch) xchunk_load(c, st, asg(md.xb), Kx, kc + 1);
        int cols = min(XCH, Kx - kc * XCH);
        wgrad_adam<SCALAR_TR>(c, c.P, LDP, 0, c.Q, LDX, N0, K0, kc * XCH, cols, md.enc_w[0], md.enc_b[0]);
      }
      prof(c, PH_ENCB_L0_WGRAD);
    }
  }
}

// ----------------------------------------------------------------------------------------------
constexpr int SMEM_BYTES = 2 * ROWS * LDP * 2 + (STAGE_FLOATS + 64 + 128 + 256 + 16) * 4;

__device__ __forceinline__ void carve_lds(Ctx& c, unsigned char* smem) {
  c.P = reinterpret_cast<__bf16*>(smem);
  c.Q = c.P + ROWS * LDP;
  c.stage = reinterpret_cast<float*>(c.Q + ROWS * LDP);
  c.red = c.stage + STAGE_FLOATS;
  c.colacc = c.red + 64;
  c.rowacc = c.colacc + 128;
  c.tlast = reinterpret_cast<unsigned long long*>(c.rowacc + 256);
}

template <bool SCALAR_TR>
__global__ __launch_bounds__(WG) void nm_step_kernel(const nm_job_t* __restrict__ jobs, int step0, int steps_per_tile,
                                                     int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const nm_job_t* J = jobs + blockIdx.x;
  Ctx c;
  c.job = J;
  carve_lds(c, smem);
  relaunder(c);
  c.flags = flags;
  c.t_last = 0;
  c.ws = (GAS char*)J->workspace + (int64_t)blockIdx.y * J->workspace_stride;
  // zero LDS once: padded columns are multiplied by zero weights and must stay finite
  for (int i = c.tid; i < SMEM_BYTES / 4; i += WG) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  __syncthreads();
  // De-phase the workgroups of a long launch: identical models otherwise run their HBM-heavy weight-gradient /
  // Adam phases in lockstep and share the DRAM 256 ways at once (measured: -3 % per step; the sleep itself costs
  // up to 7/8 of one step per launch, hence only for launches of 64 steps or more).
  if (steps_per_tile >= 64 && J->dephase > 0)
    for (int i = 0; i < (int)(blockIdx.x & 7) * J->dephase; ++i) __builtin_amdgcn_s_sleep(127);
  const int nb = (J->n_rows + ROWS - 1) / ROWS;
  const int s_begin = step0 + blockIdx.y * steps_per_tile;
  for (int s = s_begin; s < s_begin + steps_per_tile; ++s) {
    int b = s % nb;
    c.row0 = b * ROWS;
    c.nrows = min(ROWS, J->n_rows - c.row0);
    c.inv_b = 1.0f / (float)c.nrows;
    // bias corrections in double, as torch.optim.Adam computes them on the host
    const double tt = (double)(J->adam_off + (int64_t)s + 1);
    c.step_size = (float)((double)J->lr / (1.0 - pow((double)J->beta1, tt)));
    c.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)J->beta2, tt)));
    if (flags & NM_F_PROFILE) c.t_last = clock64();
    if (flags & 64) c.tlast[threadIdx.x >> 6] = clock64();
    lds_barrier();
    relaunder(c);
    run_step<SCALAR_TR>(c, s);
    tr(c, 62);
    __syncthreads();
    tr(c, 63);
  }
}

// ---- regression head (cVAE.py:2249-2253 regressor, 2318-2321 forward, 2330-2346 loss) ----------------
// fi_pred = W3 relu(W2 relu(W1 cat_m(x_m - x_hat_m) + b1) + b2) + b3;  loss = mean_r (fi_pred - FI)^2.
// One workgroup per (job, 256-row tile); same operand conventions as the trunk: bf16 MFMA operands,
// fp32 accumulate, fp32 parameters.  The concatenated residual [256][sum D] is walked in 128-column
// chunks of the CONCATENATED column space (so chunk boundaries stay 16-byte aligned inside W1's rows);
// a chunk may straddle two modalities.
struct CatCol { int m, d; };
__device__ __forceinline__ CatCol cat_col(const nm_job_t* J, int M, int col) {
  CatCol r{0, col};
#pragma unroll
  for (int m = 0; m < NM_MAX_EXP - 1; ++m) {
    const int Dm = J->mod[m].D;
    if (m < M - 1 && r.m == m && r.d >= Dm) { r.m = m + 1; r.d -= Dm; }
  }
  return r;
}
// Q[r][j] = x[r][col] - x_hat[r][col] for concatenated column col = k0 + j < SD, valid rows; 0 elsewhere.
// Walked per modality segment in the modality's own column space: x_f32 and the exported x_hat share the
// row pitch x_pitch (a multiple of 4), so a group of four columns is two aligned 16-byte loads.
__device__ __forceinline__ void resid_chunk_to_Q(const Ctx& c, int M, int SD, int k0) {
  const nm_job_t* J = c.job;
  int koff = 0;
  for (int m = 0; m < M; ++m) {
    const nm_modality_t& md = J->mod[m];
    const int lo = max(0, k0 - koff), hi = min(md.D, k0 + PW - koff);     // this modality's columns inside the chunk
    if (lo < hi) {
      gcf32 xf = asg(md.x_f32);
      gcf32 xh = asg((const float*)md.out_loc);
      const int g0 = lo >> 2, ng = ((hi + 3) >> 2) - g0;
      const float rng = 1.0f / (float)ng;
      for (int e = c.tid; e < ROWS * ng; e += WG) {
        const int r = idiv(e, ng, rng), d0 = 4 * (g0 + e - r * ng);
        const int64_t o = (int64_t)(c.row0 + r) * md.x_pitch + d0;
        const f32x4 xv = *(const GAS f32x4*)(xf + o), hv = *(const GAS f32x4*)(xh + o);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (d0 + i >= lo && d0 + i < hi) c.Q[r * LDP + koff + d0 + i - k0] = (__bf16)(r < c.nrows ? xv[i] - hv[i] : 0.f);
      }
    }
    koff += md.D;
  }
  if (k0 + PW > SD) {                             // columns past the end of the concatenation
    const int z0 = SD - k0, nz = PW - z0;
    const float rz = 1.0f / (float)nz;
    for (int e = c.tid; e < ROWS * nz; e += WG) {
      const int r = idiv(e, nz, rz);
      c.Q[r * LDP + z0 + (e - r * nz)] = (__bf16)0.0f;
    }
  }
}
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

__global__ __launch_bounds__(WG) void nm_reghead_kernel(const nm_job_t* __restrict__ jobs, int step, int tile0, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const nm_job_t* J = jobs + blockIdx.x;
  Ctx c;
  c.job = J;
  carve_lds(c, smem);
  relaunder(c);
  c.flags = flags & ~(NM_F_PROFILE | NM_F_TRACE);
  c.t_last = 0;
  c.ws = (GAS char*)J->workspace + (int64_t)blockIdx.y * J->workspace_stride;
  for (int i = c.tid; i < SMEM_BYTES / 4; i += WG) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  __syncthreads();
  c.row0 = (tile0 + (int)blockIdx.y) * ROWS;
  c.nrows = min(ROWS, J->n_rows - c.row0);
  if (c.nrows <= 0) return;
  c.inv_b = 1.0f / (float)c.nrows;
  const double tt = (double)(J->adam_off + (int64_t)step + 1);
  c.step_size = (float)((double)J->lr / (1.0 - pow((double)J->beta1, tt)));
  c.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)J->beta2, tt)));
  const bool bwd = (flags & NM_F_BACKWARD) != 0;
  const int M = experts(J);
  int SD = 0;
  for (int m = 0; m < M; ++m) SD += J->mod[m].D;
  constexpr int N1 = 128, N2 = 64;
  gcf32 prm = asg(J->params);
  gcf32 W1 = prm + J->reg_w[0], b1 = prm + J->reg_b[0], W2 = prm + J->reg_w[1], b2 = prm + J->reg_b[1];
  gcf32 W3 = prm + J->reg_w[2];
  const WsLayout wl = ws_layout(J->M, J->L, J->Z);
  // the trunk's workspace is dead between its two launches: h1 first, then the residual chunks (bf16), kept for
  // the backward pass so that x / x_hat are read once
  gbf16 ws_h1 = (gbf16)c.ws;
  gbf16 ws_res = (gbf16)(c.ws + (int64_t)ROWS * PW * 2);
  (void)wl;
  const int nch = (SD + PW - 1) / PW;

  // ---- layer 1: h1 = relu(W1 resid + b1): the residual streamed through Q, the matching 128-column block of
  // W1 through P (free until h1 exists), both as coalesced copies; next block's weights fly during the MFMAs ----
  f32x4 acc[2][RT];
  bias_acc(c, acc, b1, N1, 0);
  const int Kp1 = kpitch(SD);
  WBlk<128> wb;
  wblk_load<128>(c, wb, W1, N1, Kp1, 0, 0);
  for (int ch = 0; ch < nch; ++ch) {
    relaunder(c);
    const int k0 = ch * PW, valid = min(PW, SD - k0), ksteps = rup(valid, 32) / 32;
    resid_chunk_to_Q(c, M, SD, k0);
    wblk_store<128>(c, wb, c.P, LDP, N1, Kp1, 0, k0);
    lds_barrier();
    if (ch + 1 < nch) wblk_load<128>(c, wb, W1, N1, Kp1, 0, k0 + PW);
    if (bwd) store_act(c, ws_res + (int64_t)ch * ROWS * PW, c.Q, PW);
    for (int ks = 0; ks < ksteps; ++ks) {
      bf16x8 wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) wf[t] = lds_frag(c.P, LDP, (c.wn + 4 * t) * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        bf16x8 a = lds_frag(c.Q, LDP, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][rt] = mfma(wf[t], a, acc[t][rt]);
      }
    }
    lds_barrier();
  }
  relu_to_P(c, acc, N1);
  lds_barrier();
  if (bwd) store_act(c, ws_h1, c.P, N1);
  // ---- layer 2: h2 = relu(W2 h1 + b2), in place ----
  relaunder(c);
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
    for (int n = 0; n < N2; ++n) s = fmaf((float)c.P[c.tid * LDP + n], (float)(__bf16)W3[n], s);
    pred = s;
    if (c.tid < c.nrows) {
      if (J->out_fi_pred) asg(J->out_fi_pred)[c.row0 + c.tid] = pred;
      if (J->fi_target) err = pred - asg(J->fi_target)[c.row0 + c.tid];
    }
  }
  const float sse = block_sum(c, err * err);
  if (c.tid == 0 && J->loss_log && J->fi_target && blockIdx.y == 0)   // one tile's MSE (training: the step's batch)
    asg(J->loss_log)[(int64_t)(step % J->loss_cap) * NM_LOSS_STRIDE + NM_LOSS_REG] = sse * c.inv_b;
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
    c.P[r * LDP + n] = (__bf16)(h > 0.f ? c.rowacc[r] * (float)(__bf16)W3[n] : 0.f);
  }
  __syncthreads();                               // W3 fully read before its update
  if (c.tid < N2) apply_grad(c, J->reg_w[2] + c.tid, g3);
  else if (c.tid == N2) apply_grad(c, J->reg_b[2], g3);
  if (c.tid < N2) {                              // db2 = column sums of delta h2
    float g = 0.f;
    for (int r = 0; r < ROWS; ++r) g += (float)c.P[r * LDP + c.tid];
    apply_grad(c, J->reg_b[1] + c.tid, g);
  }
  // layer 2 backward: Q <- h1; delta h1 (pre-mask) = delta h2 W2; dW2 = delta h2^T h1
  load_act(c, c.Q, ws_h1, N1);
  lds_barrier();
  zero_acc(acc);
  dgrad_acc(c, acc, c.P, W2, N2, N1, N2 / 32, 0);
  lds_barrier();                                 // W2 fully read before its update
  wgrad_adam<false>(c, c.P, LDP, 0, c.Q, LDP, N2, N1, 0, N1, J->reg_w[1], -1);
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
  if (c.tid < N1) {                              // db1
    float g = 0.f;
    for (int r = 0; r < ROWS; ++r) g += (float)c.P[r * LDP + c.tid];
    apply_grad(c, J->reg_b[0] + c.tid, g);
  }
  // layer 1 backward per chunk: d resid = delta h1 W1 (-> dloc_extra = -d resid), dW1 chunk = delta h1^T resid
  for (int ch = 0; ch < nch; ++ch) {
    relaunder(c);
    const int k0 = ch * PW, valid = min(PW, SD - k0);
    // the W1 block of this chunk comes as two 64-row half tiles through the gradient slab (transposing reads)
    __bf16* Th = reinterpret_cast<__bf16*>(c.stage);
    WBlk<64> ha, hb;
    wblk_load<64>(c, ha, W1, N1, Kp1, 0, k0);
    wblk_load<64>(c, hb, W1, N1, Kp1, 64, k0);
    load_act(c, c.Q, ws_res + (int64_t)ch * ROWS * PW, PW);
    wblk_store<64>(c, ha, Th, LDP, N1, Kp1, 0, k0);
    lds_barrier();
    zero_acc(acc);
    dgrad_tile(c, acc, c.P, 0, Th, LDP, 2);
    lds_barrier();
    wblk_store<64>(c, hb, Th, LDP, N1, Kp1, 64, k0);
    lds_barrier();
    dgrad_tile(c, acc, c.P, 64, Th, LDP, 2);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = k0 + (c.wn + 4 * t) * 16 + 4 * c.g + i;
        if (col < SD) {
          const CatCol cc = cat_col(J, M, col);
          const nm_modality_t& md = J->mod[cc.m];
          if (md.dloc_extra) {
            gf32 dst = asg((float*)md.dloc_extra);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
              const int r = c.wm * WROWS + rt * 16 + c.c16;
              if (r < c.nrows) dst[(int64_t)(c.row0 + r) * md.x_pitch + cc.d] = -acc[t][rt][i];
            }
          }
        }
      }
    }
    lds_barrier();                               // W1 chunk fully read before its update
    wgrad_adam<false>(c, c.P, LDP, 0, c.Q, LDP, N1, SD, k0, rup(valid, 16), J->reg_w[0], -1);
  }
}

// ---- classifier head of the end-to-end model (cVAE.py:2004-2018 Classifier, 2117 logits, 2140-2200 loss) ---
// One workgroup per (job, 256-row tile).  Blocks of Linear - BatchNorm1d - ReLU - Dropout, then Linear to the
// class logits; cross entropy (mean over rows) and the contrastive hinge on the per-subject deviations the
// trunk exported.  Backward returns d CE / d z (dz_out) and, per decoder, the row coefficient of the hinge
// gradient (rowcoef_out), and applies / stores the classifier's own gradients.
struct ClsWs {
  gbf16 hin[NM_MAX_CLS + 1];
  gf32 xhat[NM_MAX_CLS];
  gf32 rstd[NM_MAX_CLS];
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
__device__ __forceinline__ float uniform_ctr(uint64_t seed, uint32_t step, uint32_t layer, uint32_t row, uint32_t f) {
  uint64_t h = splitmix64(seed ^ 0xC1A551F1E5ull ^ ((uint64_t)step << 32) ^ ((uint64_t)layer << 28) ^ ((uint64_t)row << 8) ^ f);
  return (uint32_t)(h >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(WG) void nm_clshead_kernel(const nm_job_t* __restrict__ jobs, int step, int tile0, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const nm_job_t* J = jobs + blockIdx.x;
  Ctx c;
  c.job = J;
  carve_lds(c, smem);
  relaunder(c);
  c.flags = flags & (NM_F_BACKWARD | NM_F_ADAM | NM_F_GRADS);
  c.t_last = 0;
  c.ws = (GAS char*)J->workspace + (int64_t)blockIdx.y * J->workspace_stride;
  for (int i = c.tid; i < SMEM_BYTES / 4; i += WG) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  __syncthreads();
  c.row0 = (tile0 + (int)blockIdx.y) * ROWS;
  c.nrows = min(ROWS, J->n_rows - c.row0);
  if (c.nrows <= 0) return;
  c.inv_b = 1.0f / (float)c.nrows;
  const double tt = (double)(J->adam_off + (int64_t)step + 1);
  c.step_size = (float)((double)J->lr / (1.0 - pow((double)J->beta1, tt)));
  c.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)J->beta2, tt)));
  const bool train = J->cls_train != 0;
  const bool bwd = (flags & NM_F_BACKWARD) != 0 && train && J->labels != nullptr;
  const int Lc = J->cls_layers, C = J->cls_classes, Z = J->Z;
  const float Bf = (float)c.nrows;
  gcf32 prm = asg(J->params);
  float* col1 = c.colacc;
  float* col2 = c.stage;
  ClsWs W;
  {
    GAS char* p = c.ws;
    for (int i = 0; i <= NM_MAX_CLS; ++i) { W.hin[i] = (gbf16)p; p += (int64_t)ROWS * PW * 2; }
    for (int i = 0; i < NM_MAX_CLS; ++i) { W.xhat[i] = (gf32)p; p += (int64_t)ROWS * PW * 4; }
    for (int i = 0; i < NM_MAX_CLS; ++i) { W.rstd[i] = (gf32)p; p += (int64_t)PW * 4; }
  }
  const float keep_scale = (train && J->cls_dropout > 0.f) ? 1.0f / (1.0f - J->cls_dropout) : 1.0f;

  // ---- P <- z (or the joint mean for predict) ----
  {
    const int Kz = rup(Z, 32);
    const float rk = 1.0f / (float)Kz;
    gcf32 zsrc = asg((const float*)(J->cls_use_mu ? J->out_mu : J->out_z));
    for (int e = c.tid; e < ROWS * Kz; e += WG) {
      const int r = idiv(e, Kz, rk), k = e - r * Kz;
      const float v = zsrc[(int64_t)(c.row0 + min(r, c.nrows - 1)) * Z + min(k, Z - 1)];
      c.P[r * LDP + k] = (__bf16)((k < Z && r < c.nrows) ? v : 0.f);
    }
  }
  lds_barrier();

  f32x4 acc[2][RT];
  // ---- hidden blocks ----
  for (int li = 0; li < Lc; ++li) {
    relaunder(c);
    const int K = li == 0 ? Z : J->cls_width[li - 1], N = J->cls_width[li], K32 = rup(K, 32);
    gcf32 Wl = prm + J->cls_w[li];
    if (bwd) store_act(c, W.hin[li], c.P, K32);
    bias_acc(c, acc, prm + J->cls_b[li], N, 0);
    for (int ks = 0; ks < K32 / 32; ++ks) {
      bf16x8 wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) wf[t] = w_frag(Wl, N, K, (c.wn + 4 * t) * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        bf16x8 a = lds_frag(c.P, LDP, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][rt] = mfma(wf[t], a, acc[t][rt]);
      }
    }
    if (c.tid < PW) { col1[c.tid] = 0.f; col2[c.tid] = 0.f; }
    __syncthreads();                               // P fully read; column accumulators cleared
    float mean[2][4], rstd[2][4], gam[2][4], bet[2][4];
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
        gam[t][i] = prm[J->cls_bn_w[li] + fc];
        bet[t][i] = prm[J->cls_bn_b[li] + fc];
      }
    if (c.tid < N && train) {                      // per-feature rstd for the backward pass; running statistics
      const float m = col1[c.tid] / Bf, var = col2[c.tid] / Bf;
      if (bwd) W.rstd[li][c.tid] = 1.0f / sqrtf(var + 1e-5f);
      if ((flags & NM_F_BNSTATS) && blockIdx.y == 0) {
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
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xh[i] = (acc[t][rt][i] - mean[t][i]) * rstd[t][i];
          float h = fmaxf(gam[t][i] * xh[i] + bet[t][i], 0.f);
          if (train && J->cls_dropout > 0.f)
            h = uniform_ctr(J->seed, (uint32_t)step, (uint32_t)li, (uint32_t)(c.row0 + r), (uint32_t)(f0 + i)) >= J->cls_dropout
                    ? h * keep_scale : 0.f;
          pk[i] = (__bf16)((f0 + i < N && r < c.nrows) ? h : 0.f);
        }
        if (bwd) *(GAS f32x4*)(W.xhat[li] + r * PW + f0) = xh;
        *reinterpret_cast<bf16x4*>(c.P + r * LDP + f0) = pk;
      }
    }
    lds_barrier();
  }

  // ---- output layer, cross entropy ----
  relaunder(c);
  const int Kl = Lc ? J->cls_width[Lc - 1] : Z, Kl32 = rup(Kl, 32);
  gcf32 Wo = prm + J->cls_w[Lc];
  if (bwd) store_act(c, W.hin[Lc], c.P, Kl32);
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
  if (c.tid == 0 && J->loss_log && J->labels && blockIdx.y == 0) {
    gf32 row = asg(J->loss_log) + (int64_t)(step % J->loss_cap) * NM_LOSS_STRIDE;
    row[NM_LOSS_CE] = ce_sum * c.inv_b;
    row[NM_LOSS_CONTRAST] = hinge_sum * c.inv_b;
  }
  if (!bwd) return;

  // ---- backward: output layer ----
  relaunder(c);
  load_act(c, c.Q, W.hin[Lc], Kl32);
  lds_barrier();
  zero_acc(acc);
  dgrad_acc(c, acc, c.P, Wo, C, Kl, 1, 0);
  float gb = 0.f;
  if (c.tid < C) for (int r = 0; r < ROWS; ++r) gb += (float)c.P[r * LDP + c.tid];
  lds_barrier();                                   // Wo fully read before its update
  wgrad_adam<false>(c, c.P, LDP, 0, c.Q, LDP, C, Kl, 0, rup(Kl, 16), J->cls_w[Lc], -1);
  if (c.tid < C) apply_grad(c, J->cls_b[Lc] + c.tid, gb);
  // ---- backward: hidden blocks ----
  for (int li = Lc - 1; li >= 0; --li) {
    relaunder(c);
    const int K = li == 0 ? Z : J->cls_width[li - 1], N = J->cls_width[li], K32 = rup(K, 32);
    gcf32 Wl = prm + J->cls_w[li];
    // acc = d h (pre-mask), Q = h of this block.  d y = d h * relu'/dropout mask; BatchNorm backward needs the
    // column sums S1 = sum d y, S2 = sum d y * x_hat
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
        const f32x4 xh = *(const GAS f32x4*)(W.xhat[li] + r * PW + f0);
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
        gr_[i] = prm[J->cls_bn_w[li] + fc] * W.rstd[li][fc];
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int r = c.wm * WROWS + rt * 16 + c.c16;
        const f32x4 xh = *(const GAS f32x4*)(W.xhat[li] + r * PW + f0);
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
    load_act(c, c.Q, W.hin[li], K32);
    lds_barrier();
    float gbi = 0.f;
    if (c.tid < N) for (int r = 0; r < ROWS; ++r) gbi += (float)c.P[r * LDP + c.tid];
    zero_acc(acc);
    dgrad_acc(c, acc, c.P, Wl, N, K, rup(N, 32) / 32, 0);
    lds_barrier();
    wgrad_adam<false>(c, c.P, LDP, 0, c.Q, LDP, N, K, 0, rup(K, 16), J->cls_w[li], -1);
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

__global__ void pack_table_kernel(const float* __restrict__ x, const float* __restrict__ cc, int n_rows, int rows_alloc,
                                  int D, int C, int Kx, uint16_t* __restrict__ xb, float* __restrict__ xf, int xp) {
  int64_t total = (int64_t)rows_alloc * Kx;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(i / Kx), k = (int)(i - (int64_t)r * Kx);
    float v = 0.f;
    if (r < n_rows) {
      if (k < D) v = x[(int64_t)r * D + k];
      else if (k < D + C) v = cc[(int64_t)r * C + (k - D)];
      else if (k == D + C) v = 1.0f;
    }
    __bf16 h = (__bf16)v;
    xb[i] = __builtin_bit_cast(uint16_t, h);
    if (xf && k < xp) xf[(int64_t)r * xp + k] = (r < n_rows && k < D) ? x[(int64_t)r * D + k] : 0.f;
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

int nm_version(void) { return 4; }

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
    case -7: return "table pitch: Kx must be a multiple of 32 and >= D + C + 1, x_pitch a multiple of 4 and >= D";
    case -8: return "bad launch geometry";
    case -9: return "unknown combine";
    case -10: return "parameter tensor offsets must be multiples of 4 floats";
    case -13: return "classifier head: 0..NM_MAX_CLS blocks of width 1..128, 2..NM_MAX_CLASSES classes, offsets multiples of 4, out_mu/out_z export";
    case -12: return "metrics: n_sets >= 1 and 1 <= max_set <= NM_METRICS_MAX_N";
    case -11: return "regression head: needs reg_w / reg_b offsets (multiples of 4) and every expert's out_loc export";
    default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown argument error";
  }
}

int nm_validate_job(const nm_job_t* j) {
  if (!j) return -1;
  if (j->M < 1 || j->M > NM_MAX_MOD) return -2;
  if (j->M_enc < 0 || j->M_enc > j->M || (j->M_enc == 0 ? j->M : j->M_enc) > NM_MAX_EXP) return -2;
  if (j->L < 1 || j->L > NM_MAX_HID) return -3;
  for (int i = 0; i < j->L; ++i)
    if (j->H[i] < 1 || j->H[i] > NM_MAX_WIDTH) return -4;
  if (j->Z < 1 || j->Z > NM_MAX_LATENT) return -5;
  if (j->Z + j->C > NM_MAX_WIDTH) return -6;
  if (j->combine < 0 || j->combine > NM_COMBINE_MOPOE) return -9;
  for (int m = 0; m < j->M; ++m) {
    const nm_modality_t& md = j->mod[m];
    if (md.Kx % 32 != 0 || md.Kx < md.D + j->C + 1) return -7;
    if (md.x_pitch % 4 != 0 || md.x_pitch < md.D) return -7;
    for (int i = 0; i < j->L; ++i)
      if ((md.enc_w[i] | md.enc_b[i] | md.dec_w[i] | md.dec_b[i]) & 3) return -10;
    if ((md.mu_w | md.mu_b | md.lv_w | md.lv_b | md.logvar_out | md.out_w | md.out_b) & 3) return -10;
  }
  if (j->cls_classes > 0) {
    if (j->cls_layers < 0 || j->cls_layers > NM_MAX_CLS || j->cls_classes < 2 || j->cls_classes > NM_MAX_CLASSES) return -13;
    for (int i = 0; i < j->cls_layers; ++i) {
      if (j->cls_width[i] < 1 || j->cls_width[i] > PW) return -13;
      if ((j->cls_w[i] | j->cls_b[i] | j->cls_bn_w[i] | j->cls_bn_b[i] | j->cls_bn_mean[i] | j->cls_bn_var[i]) & 3) return -13;
    }
    if ((j->cls_w[j->cls_layers] | j->cls_b[j->cls_layers]) & 3) return -13;
    if (!(j->cls_use_mu ? j->out_mu : j->out_z)) return -13;
  }
  if (j->reg_head) {
    for (int i = 0; i < 3; ++i)
      if (j->reg_w[i] < 0 || j->reg_b[i] < 0 || ((j->reg_w[i] | j->reg_b[i]) & 3)) return -11;
    for (int m = 0; m < (j->M_enc == 0 ? j->M : j->M_enc); ++m)
      if (!j->mod[m].out_loc) return -11;
  }
  return 0;
}

int64_t nm_workspace_bytes(const nm_job_t* j) {
  if (!j) return -1;
  int64_t b = ws_layout(j->M, j->L, j->Z).total;
  if (j->reg_head) {                      // regression head: h1 + one bf16 tile per 128 concatenated residual columns
    int sd = 0;
    for (int m = 0; m < (j->M_enc > 0 ? j->M_enc : j->M); ++m) sd += j->mod[m].D;
    int64_t hb = (int64_t)(1 + (sd + PW - 1) / PW) * ROWS * PW * 2;
    b = b > hb ? b : (hb + 255) / 256 * 256;
  }
  if (j->cls_layers > 0 || j->cls_classes > 0) b = b > cls_ws_bytes() ? b : (cls_ws_bytes() + 255) / 256 * 256;
  return b;
}

static int launch_impl(const nm_job_t* jobs_dev, int n_jobs, int step0, int steps_per_tile, int n_tiles, int flags,
                       void* stream, bool scalar_tr) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1 || steps_per_tile < 1 || n_tiles < 1 || step0 < 0) return -8;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(n_jobs, n_tiles), block(WG);
  hipError_t e;
  if (scalar_tr) {
    e = hipFuncSetAttribute((const void*)nm_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nm_step_kernel<true>, grid, block, SMEM_BYTES, st, jobs_dev, step0, steps_per_tile, flags);
  } else {
    e = hipFuncSetAttribute((const void*)nm_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nm_step_kernel<false>, grid, block, SMEM_BYTES, st, jobs_dev, step0, steps_per_tile, flags);
  }
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
  hipError_t e = hipFuncSetAttribute((const void*)nm_reghead_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nm_reghead_kernel, dim3(n_jobs, n_tiles), dim3(WG), SMEM_BYTES, (hipStream_t)stream, jobs_dev, step,
                     tile0, flags);
  return (int)hipGetLastError();
}

int nm_head_classifier(const nm_job_t* jobs_dev, int n_jobs, int step, int tile0, int n_tiles, int flags, void* stream) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1 || n_tiles < 1 || step < 0 || tile0 < 0) return -8;
  hipError_t e = hipFuncSetAttribute((const void*)nm_clshead_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nm_clshead_kernel, dim3(n_jobs, n_tiles), dim3(WG), SMEM_BYTES, (hipStream_t)stream, jobs_dev, step,
                     tile0, flags);
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

int nm_pack_table(const float* x, const float* c, int n_rows, int rows_alloc, int D, int C, int Kx, uint16_t* xb,
                  float* x_f32_out, int x_pitch, void* stream) {
  if (!x || !xb || (C > 0 && !c)) return -1;
  if (Kx % 32 != 0 || Kx < D + C + 1 || rows_alloc < n_rows || rows_alloc % NM_BATCH != 0) return -7;
  if (x_f32_out && (x_pitch % 4 != 0 || x_pitch < D || x_pitch > Kx)) return -7;
  int64_t total = (int64_t)rows_alloc * Kx;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_table_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, c, n_rows, rows_alloc, D, C,
                     Kx, xb, x_f32_out, x_pitch);
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
