// nmhip.hip -- libnmhip.so: whole-batch step kernels, head models, general-shape path and the C ABI.
// The device functions of the step live in nm_core.inc (shared with nm_rowsplit.hip).
#include "nm_core.inc"

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
// [256][128] tiles per activation of the classifier: 1 = every block fits the fused head's tile; > 1: blocks wider than 128
// (-Layers "256 128 64" of commands_list9_endtoend.sh:21) -- cls_head_body_wide, activations in workspace tiles
__host__ __device__ inline int cls_tiles(const nm_job_t* J) {
  int tb = 1;
  for (int i = 0; i < J->cls_layers && i < NM_MAX_CLS; ++i) tb = tb > wblocks(J->cls_width[i]) ? tb : wblocks(J->cls_width[i]);
  return tb;
}
// the tile form of the classifier workspace: inputs of Linear 0 .. NM_MAX_CLS as TB tiles each (bf16), x_hat of every BatchNorm
// (fp32 tiles), 1 / sigma, and two sets of delta tiles (bf16) the backward alternates between
struct ClsWsW {
  GAS char* base;
  int TB;
  __device__ __forceinline__ int64_t o_xhat() const { return (int64_t)(NM_MAX_CLS + 1) * TB * WTILE * 2; }
  __device__ __forceinline__ int64_t o_rstd() const { return o_xhat() + (int64_t)NM_MAX_CLS * TB * WTILE * 4; }
  __device__ __forceinline__ int64_t o_dy() const { return o_rstd() + (int64_t)NM_MAX_CLS * TB * PW * 4; }
  __device__ __forceinline__ gbf16 hin(int i) const { return (gbf16)(base + (int64_t)i * TB * WTILE * 2); }
  __device__ __forceinline__ gf32 xhat(int i) const { return (gf32)(base + o_xhat() + (int64_t)i * TB * WTILE * 4); }
  __device__ __forceinline__ gf32 rstd(int i) const { return (gf32)(base + o_rstd() + (int64_t)i * TB * PW * 4); }
  __device__ __forceinline__ gbf16 dy(int s) const { return (gbf16)(base + o_dy() + (int64_t)s * TB * WTILE * 2); }
};
__host__ __device__ inline int64_t cls_ws_bytes(const nm_job_t* J) {
  const int64_t tb = cls_tiles(J);
  if (tb <= 1) return (int64_t)(NM_MAX_CLS + 1) * ROWS * PW * 2 + (int64_t)NM_MAX_CLS * ROWS * PW * 4 + (int64_t)NM_MAX_CLS * PW * 4;
  return (int64_t)(NM_MAX_CLS + 1) * tb * WTILE * 2 + (int64_t)NM_MAX_CLS * tb * WTILE * 4 + (int64_t)NM_MAX_CLS * tb * PW * 4 +
         2 * tb * WTILE * 2;
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

__device__ __forceinline__ void cls_head_body_wide(Ctx& c, const nm_job_t* J, int step, GAS char* hws, bool log, bool bn_stats);
// (same contract as reg_head_body; `bn_stats`: update the BatchNorm running statistics)
// TILED_OK: blocks wider than 128 go to cls_head_body_wide (nm_clshead_kernel); the persistent head kernel compiles the
// one-tile head only (with both it spills vector registers, which the ISA guard refuses) and marks such a job's loss row
// NaN instead of running it -- JobSet.train_endtoend sends those models through the three-launch form.
template <bool TILED_OK>
__device__ __forceinline__ void cls_head_body(Ctx& c, const nm_job_t* J, int step, GAS char* hws, bool log, bool bn_stats) {
  if (cls_tiles(J) > 1) {
    if constexpr (TILED_OK) cls_head_body_wide(c, J, step, hws, log, bn_stats);
    else if (c.tid < NM_LOSS_STRIDE && J->loss_log)
      asg(J->loss_log)[(int64_t)(step % J->loss_cap) * NM_LOSS_STRIDE + c.tid] = __builtin_nanf("");
    return;
  }
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

// The same head with blocks wider than 128 (cls_tiles(J) > 1): every activation lives in the workspace as [256][128] bf16
// tiles, a Linear loops over (output block, input block) with the general-shape path's block GEMMs (nm_wide.inc: weights
// from the tiled fp32 master, staged as bf16 one block ahead), BatchNorm statistics / ReLU / dropout per output block in the
// accumulator's lanes exactly as above, the weight gradients + Adam through wide_linear_bwd.  Same arithmetic per element
// as cls_head_body; a block <= 128 wide gives the same numbers up to the bf16 rounding of the tiles it passes through.
__device__ __forceinline__ void cls_head_body_wide(Ctx& c, const nm_job_t* J, int step, GAS char* hws, bool log, bool bn_stats) {
  const int flags = c.flags;
  const bool train = J->cls_train != 0;
  const bool bwd = (flags & NM_F_BACKWARD) != 0 && train && J->labels != nullptr;
  const int Lc = J->cls_layers, C = J->cls_classes, Z = J->Z;
  const float Bf = (float)c.nrows;
  gcf32 prm = asg(J->params);
  float* col1 = c.colacc;
  float* col2 = c.stage;
  const ClsWsW W{hws, cls_tiles(J)};
  const float keep_scale = (train && J->cls_dropout > 0.f) ? 1.0f / (1.0f - J->cls_dropout) : 1.0f;
  auto zero_P = [&]() {
    for (int e = c.tid; e < ROWS * (LDP / 8); e += WG) reinterpret_cast<u32x4*>(c.P)[e] = u32x4{0u, 0u, 0u, 0u};
  };

  // ---- tile 0 of hin(0) <- z (or the joint mean for predict) ----
  lds_barrier();
  zero_P();
  lds_barrier();
  {
    gcf32 zsrc = asg((const float*)(J->cls_use_mu ? J->out_mu : J->out_z));
    const float rk = 1.0f / (float)Z;
    for (int e = c.tid; e < c.nrows * Z; e += WG) {
      const int r = idiv(e, Z, rk), k = e - r * Z;
      c.P[r * LDP + k] = (__bf16)zsrc[(int64_t)(c.row0 + r) * Z + k];
    }
  }
  lds_barrier();
  store_act(c, W.hin(0), c.P, PW);
  handoff_barrier();

  f32x4 acc[2][RT];
  // ---- hidden blocks ----
  for (int li = 0; li < Lc; ++li) {
    const int K = li == 0 ? Z : J->cls_width[li - 1], N = J->cls_width[li];
    gcf32 Wl = prm + J->cls_w[li];
    for (int nb = 0; nb < wblocks(N); ++nb) {
      relaunder(c);
      const int fb = nb * WT, nv = min(WT, N - fb);            // this block: features [fb, fb + nv)
      float mean[2][4], rstd[2][4], gam[2][4], bet[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int fc = fb + min((c.wn + 4 * t) * 16 + 4 * c.g + i, nv - 1);
          gam[t][i] = prm[J->cls_bn_w[li] + fc];
          bet[t][i] = prm[J->cls_bn_b[li] + fc];
        }
      bias_acc(c, acc, prm + J->cls_b[li], N, fb);
      wide_gemm_fwd(c, acc, W.hin(li), (gcbf16)nullptr, 0, Wl, N, K, nb);
      relaunder(c);
      if (c.tid < PW) { col1[c.tid] = 0.f; col2[c.tid] = 0.f; }
      __syncthreads();                             // P / Q fully read; column accumulators cleared
      if (train) {                                 // batch statistics over the valid rows (biased variance)
        float v[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float sm = 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) sm += (c.wm * WROWS + rt * 16 + c.c16 < c.nrows) ? acc[t][rt][i] : 0.f;
            v[t][i] = sm;
          }
        col_reduce(c, v, col1, nv);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int f = min((c.wn + 4 * t) * 16 + 4 * c.g + i, PW - 1);
            mean[t][i] = col1[f] / Bf;
            float sm = 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
              const float d = acc[t][rt][i] - mean[t][i];
              sm += (c.wm * WROWS + rt * 16 + c.c16 < c.nrows) ? d * d : 0.f;
            }
            v[t][i] = sm;
          }
        col_reduce(c, v, col2, nv);
        __syncthreads();
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = (c.wn + 4 * t) * 16 + 4 * c.g + i, fc = fb + min(f, nv - 1);
          float var;
          if (train) { var = col2[min(f, PW - 1)] / Bf; }
          else { mean[t][i] = prm[J->cls_bn_mean[li] + fc]; var = prm[J->cls_bn_var[li] + fc]; }
          rstd[t][i] = 1.0f / sqrtf(var + 1e-5f);
        }
      if (c.tid < nv && train) {                   // per-feature rstd for the backward pass; running statistics
        const float m = col1[c.tid] / Bf, var = col2[c.tid] / Bf;
        if (bwd) W.rstd(li)[fb + c.tid] = 1.0f / sqrtf(var + 1e-5f);
        if (bn_stats && log) {
          gf32 rm = asg(J->params) + J->cls_bn_mean[li] + fb + c.tid, rv = asg(J->params) + J->cls_bn_var[li] + fb + c.tid;
          const float unb = c.nrows > 1 ? var * Bf / (Bf - 1.0f) : var;
          *rm = 0.9f * *rm + 0.1f * m;
          *rv = 0.9f * *rv + 0.1f * unb;
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int f0 = (c.wn + 4 * t) * 16 + 4 * c.g;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int r = c.wm * WROWS + rt * 16 + c.c16;
          f32x4 xh;
          bf16x4 pk;
          f32x4 u4 = {1.f, 1.f, 1.f, 1.f};
          if (train && J->cls_dropout > 0.f)
            u4 = uniform4_ctr(J->seed, (uint32_t)step, (uint32_t)li, (uint32_t)(c.row0 + r), (uint32_t)((fb + f0) >> 2));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            xh[i] = (acc[t][rt][i] - mean[t][i]) * rstd[t][i];
            float h = fmaxf(gam[t][i] * xh[i] + bet[t][i], 0.f);
            if (train && J->cls_dropout > 0.f) h = u4[i] >= J->cls_dropout ? h * keep_scale : 0.f;
            pk[i] = (__bf16)((f0 + i < nv && r < c.nrows) ? h : 0.f);
          }
          if (bwd) *(GAS f32x4*)(W.xhat(li) + (int64_t)nb * WTILE + r * PW + f0) = xh;
          *reinterpret_cast<bf16x4*>(c.P + r * LDP + f0) = pk;
        }
      }
      lds_barrier();
      store_act(c, W.hin(li + 1) + (int64_t)nb * WTILE, c.P, PW);
    }
    handoff_barrier();                             // the block's tiles are read back by the next Linear
  }

  tr(c, 26);
  // ---- output layer, cross entropy ----
  relaunder(c);
  const int Kl = Lc ? J->cls_width[Lc - 1] : Z;
  bias_acc(c, acc, prm + J->cls_b[Lc], C, 0);
  wide_gemm_fwd(c, acc, W.hin(Lc), (gcbf16)nullptr, 0, prm + J->cls_w[Lc], C, Kl, 0);
  relaunder(c);
  lds_barrier();
  zero_P();                                        // d logits land here
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
        const float gt = tval > 0.f ? J->cls_w_contrast * c.inv_b : 0.f;
        const float g_h = y ? -gt : gt;
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

  // ---- backward ----
  lds_barrier();
  store_act(c, W.dy(0), c.P, PW);                  // d logits as a delta tile
  handoff_barrier();
  float* const hpatch = c.stage + SPATCH_OFF / 4;
  int cur = 0;                                     // dy(cur): delta at the output of Linear li
  for (int li = Lc; li >= 0; --li) {
    const int K = li == 0 ? Z : J->cls_width[li - 1], N = li == Lc ? C : J->cls_width[li];
    gcf32 Wl = prm + J->cls_w[li];
    gcbf16 dyt = W.dy(cur);
    if (li > 0) {
      const int lb = li - 1;                       // the block whose output is Linear li's input
      for (int kb = 0; kb < wblocks(K); ++kb) {
        relaunder(c);
        const int fb = kb * WT, kv = min(WT, K - fb);
        zero_acc(acc);
        wide_gemm_dgrad(c, acc, dyt, Wl, N, K, kb);          // d h (pre-mask) of features [fb, fb + kv)
        relaunder(c);
        float grs[2][4];                           // gamma * rstd of this lane's features
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int fc = fb + min((c.wn + 4 * t) * 16 + 4 * c.g + i, kv - 1);
            grs[t][i] = prm[J->cls_bn_w[lb] + fc] * W.rstd(lb)[fc];
          }
        if (c.tid < PW) { col1[c.tid] = 0.f; col2[c.tid] = 0.f; }
        __syncthreads();                           // P / Q fully read; column accumulators cleared
        gcbf16 ht = W.hin(li) + (int64_t)kb * WTILE;
        gcf32 xt = W.xhat(lb) + (int64_t)kb * WTILE;
        float v1[2][4], v2[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int f0 = (c.wn + 4 * t) * 16 + 4 * c.g;
#pragma unroll
          for (int i = 0; i < 4; ++i) { v1[t][i] = 0.f; v2[t][i] = 0.f; }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const int r = c.wm * WROWS + rt * 16 + c.c16;
            const bf16x4 h = *(const GAS bf16x4*)(ht + r * PW + f0);
            const f32x4 xh = *(const GAS f32x4*)(xt + r * PW + f0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float dy = ((float)h[i] > 0.f && r < c.nrows && f0 + i < kv) ? acc[t][rt][i] * keep_scale : 0.f;
              acc[t][rt][i] = dy;
              v1[t][i] += dy;
              v2[t][i] = fmaf(dy, xh[i], v2[t][i]);
            }
          }
        }
        col_reduce(c, v1, col1, kv);
        col_reduce(c, v2, col2, kv);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int f0 = (c.wn + 4 * t) * 16 + 4 * c.g;
          float s1[4], s2[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s1[i] = col1[min(f0 + i, PW - 1)] * c.inv_b;
            s2[i] = col2[min(f0 + i, PW - 1)] * c.inv_b;
          }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const int r = c.wm * WROWS + rt * 16 + c.c16;
            const f32x4 xh = *(const GAS f32x4*)(xt + r * PW + f0);
            bf16x4 pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float da = grs[t][i] * (acc[t][rt][i] - s1[i] - xh[i] * s2[i]);
              pk[i] = (__bf16)((r < c.nrows && f0 + i < kv) ? da : 0.f);
            }
            *reinterpret_cast<bf16x4*>(c.P + r * LDP + f0) = pk;
          }
        }
        __syncthreads();                           // gamma has been read by everyone: its update may go ahead
        if (c.tid < kv) {                          // d gamma = S2, d beta = S1
          apply_grad(c, J->cls_bn_w[lb] + fb + c.tid, col2[c.tid]);
          apply_grad(c, J->cls_bn_b[lb] + fb + c.tid, col1[c.tid]);
        }
        store_act(c, W.dy(cur ^ 1) + (int64_t)kb * WTILE, c.P, PW);
      }
    } else if (J->dz_out) {                        // d CE / d z
      relaunder(c);
      zero_acc(acc);
      wide_gemm_dgrad(c, acc, dyt, Wl, N, K, 0);
      relaunder(c);
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
    // the weights of Linear li have been read by the dgrad: their gradient / update
    wide_linear_bwd(c, dyt, W.hin(li), (gcbf16)nullptr, 0, J->cls_w[li], J->cls_b[li], N, K, false, (gbf16)nullptr, (gf32)nullptr, 0,
                    0, hpatch);
    handoff_barrier();
    cur ^= 1;
  }
}
__global__ __launch_bounds__(WG) void nm_clshead_kernel(const nm_job_t* __restrict__ jobs, int step, int tile0, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const nm_job_t* J = jobs + blockIdx.x;
  Ctx c;
  if (!head_setup(c, smem, J, step, tile0, flags & (NM_F_BACKWARD | NM_F_ADAM | NM_F_GRADS))) return;
  cls_head_body<true>(c, J, step, c.ws + trunk_ws_bytes(J), blockIdx.y == 0, (flags & NM_F_BNSTATS) != 0);
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
  if (J->wide) {          // a trunk of the general-shape path trains in the three-launch form (JobSet.train_regression / train_endtoend)
    if (c.tid < NM_LOSS_STRIDE && J->loss_log)
      for (int s = step0; s < step0 + n_steps; ++s) asg(J->loss_log)[(int64_t)(s % J->loss_cap) * NM_LOSS_STRIDE + c.tid] = __builtin_nanf("");
    return;
  }
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
    else if (J->cls_classes > 0) cls_head_body<false>(c, J, s, hws, true, (flags & NM_F_BNSTATS) != 0);
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
  for (int m = 0; m < (J->wide ? 0 : J->M); ++m) {             // (general-shape jobs: no trunk images, the fp32 master is read)
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

int nm_version(void) { return 10; }

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
    case -19: return "general-shape path (wide): mvtCAE's total correlation needs experts x latent <= 256";
    case -17: return "input preparation: 1 <= rows <= NM_PREP_MAX_ROWS, at least one source / column / bin";
    case -16: return "split launch: jobs x parts exceeds the number of CUs (the parts of a model wait for each other and must all be resident)";
    case -20: return "row-split launch: the job uses a switch that needs the whole batch in one workgroup, or lacks gpart / workspace tiles";
    case -22: return "deviation-pass kernel: one expert with the single-expert bypass, first hidden width <= 112, latent <= 32, Gaussian output only";
    case -21: return "n_params must be set and stay below 2^30 floats (32-bit byte offsets into params / adam_m / adam_v)";
    case -15: return "wsh (shadow images) missing: allocate nm_fill_shadow() bytes, zero them and call nm_sync_shadow()";
    case -8: return "bad launch geometry";
    case -9: return "unknown combine";
    case -10: return "parameter tensor offsets must be multiples of 4 floats (weight matrices: of 256)";
    case -13: return "classifier head: 0..NM_MAX_CLS blocks of width 1..NM_MAX_CLS_WIDTH, 2..NM_MAX_CLASSES classes, offsets multiples of 4, out_mu/out_z export";
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
    // (the heads of the end-to-end and the regression model run as their own kernels, nm_head_classifier /
    //  nm_head_regression, on any trunk)
    // (mvtCAE's switches -- ProductOfExperts2 on variances, the variance floor, the total-correlation term -- are served; its
    //  log-sum-exps sit in 256 floats of LDS)
    // (so are the DMVAE family's: private latents, sigmoid output, learnable loss weights)
    if (j->tc_weight != 0.f && (j->M_enc > 0 ? j->M_enc : j->M) * j->Z > 256) return -19;
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
  if (j->n_params < 1 || j->n_params >= ((int64_t)1 << 30)) return -21;      // (unsigned)(offset << 2) in the Adam units
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
      if (j->cls_width[i] < 1 || j->cls_width[i] > NM_MAX_CLS_WIDTH) return -13;
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
    md.out_s = 0;
    if (j->wide) continue;                         // general-shape trunk: no images (only a regression head's, below)
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

/* Where a tile's workspace keeps the experts' statistics after a launch of the fused kernels: byte offset of
 * mu_m (what = 0) / logvar_m (what = 1), fp32 [step parity][expert][256][Z rounded to 16]; < 0: none (general-shape jobs). */
int64_t nm_workspace_offset(const nm_job_t* j, int what) {
  if (!j || what < 0 || what > 1) return -1;
  if (j->wide) {                                   // (general-shape jobs: [expert][256][Z rounded to 16], no step parity)
    const WideWs ww = wide_ws_layout(j);
    return what == 0 ? ww.mu_m : ww.lv_m;
  }
  const WsLayout w = ws_layout(j->M, j->L, j->Z);
  return what == 0 ? w.mu_m : w.lv_m;
}

int64_t nm_workspace_bytes(const nm_job_t* j) {
  if (!j) return -1;
  int64_t b = trunk_ws_bytes(j);                      // the head's region sits behind the trunk's
  int64_t hb = 0;
  if (j->reg_head) hb = ACT_BYTES;        // regression head: its first hidden activation, kept for the backward pass
  if (j->cls_layers > 0 || j->cls_classes > 0) hb = hb > cls_ws_bytes(j) ? hb : cls_ws_bytes(j);
  return b + (hb + 255) / 256 * 256;
}

int nm_sync_reset(const nm_job_t* jobs_dev, int n_jobs, void* stream) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1) return -8;
  hipLaunchKernelGGL(sync_reset_kernel, dim3((n_jobs * (WS_SYNC_BYTES / 4) + 255) / 256), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
  return (int)hipGetLastError();
}

int nm_rowsplit_ok(const nm_job_t* j) {
  if (!j) return -1;
  if (j->wide || j->reg_head || j->cls_classes > 0 || j->cls_layers > 0) return -20;
  if (j->w_off >= 0 || j->tc_weight != 0.f || j->n_private != 0 || j->out_kind != 0) return -20;
  if (j->M_enc != 0 && j->M_enc != j->M) return -20;
  if (j->M > NM_MAX_EXP) return -20;
  for (int m = 0; m < j->M; ++m) {                 // the sweep's tables (nm_rowsplit.hip): passes, vector segments and elements
    const int nck = (j->mod[m].D + OCH - 1) / OCH, nch = (j->mod[m].Kx + XCH - 1) / XCH;
    if (nck + 2 * j->L + 1 + nch > NM_RS_MAX_PASSES || 2 * nck + 2 * j->L + 3 > NM_RS_MAX_VSEGS) return -20;
    int64_t vtot = 2 * (int64_t)j->mod[m].D + 2 * j->Z + 1;
    for (int i = 0; i < j->L; ++i) vtot += 2 * j->H[i];
    if (vtot > 2 * WG * 3) return -20;              // three vector elements per thread at k = 2 (nm_rowsplit.hip: SW_NV)
  }
  // (the sweep addresses slice q's partials at a 32-bit byte offset q * gpart_stride * 4 from slice 0's, q < 4)
  if (!j->gpart || j->gpart_stride < j->n_params || (j->gpart_stride & 255) || j->gpart_stride >= ((int64_t)1 << 28)) return -20;
  return 0;
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
    nm_sync_reset(jobs_dev, n_jobs, stream);
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

