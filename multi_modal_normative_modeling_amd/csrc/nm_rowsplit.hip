// nm_rowsplit.hip -- the row-split launch of the train step (libnmhip.so, second translation unit).
//
// Small sweeps -- the reference's own unit of work is 5 folds x 4 procedures trained one after another
// (multimodal_kfold_train_cvae_supervised.py:68,82; commands_list_deviation.sh:13-23) -- leave most of the chip idle when
// a model is one workgroup (nm_launch) or one workgroup per modality (nm_launch_split): 5 SE models use 15 of 256 CUs
// and a step is one CU's serial chain.  Here k = 2 or 4 workgroups share one (model, modality): each owns 256 / k rows
// of the batch (run_step<.., RTV = 8 / k> of nm_core.inc: forward and backward on its rows), writes its fp32
// weight-gradient PARTIALS tile-linear to a per-slice buffer (nm_job_t.gpart, the master's own offsets, write-through),
// meets the other workgroups of the model once (hand-off C), and then sums the k partials IN SLICE ORDER for its 1 / k of
// the Adam units and updates them (rs_sweep: p / m / v, the bf16 shadow images and their fp32 vector pieces); a last
// hand-off (D, the k slices of the modality) publishes the new shadow images before the next step's forward reads them.
// Bitwise reproducible run to run (fixed summation order, fixed unit -> workgroup map); NOT bit-equal to the whole-batch
// kernels (a different association of the same fp32 sums): parity is held against the oracle at the same bounds.
//
// Hand-offs follow cdna_hip_programming.md Guideline 16, recipe R1: every byte another workgroup reads (expert
// statistics, d z, gradient partials, loss shares, shadow images / vector pieces, alpha) is stored write-through (sc1)
// and drained by its storing wave; ONE lane adds to a monotonic counter and polls it relaxed; ONE agent-scope acquire;
// then plain loads / LDS-DMA.  No release fence, hence no write-back of the XCD's dirty p / m / v lines per hand-off.
//
// Reference semantics: the step is the one of nm_core.inc (cVAE.py:1166-1196 forward_multimodal / loss_function_multimodal,
// cVAE.py:1111-1116 Adam; train loop multimodal_kfold_train_cvae_supervised.py:177-199).
#include "nm_core.inc"

// ---- Adam sweep over the summed partials ---------------------------------------------------------------------------
// One weight-gradient pass G (the geometry run_step used for it): its 16 x 16 tiles are dealt round-robin over the
// KS * 8 waves of the modality's KS slices, starting at global tile number `base` (the running tile count of the
// passes before it, so that consecutive small passes keep all waves busy).  Per tile: p / m / v and the KS partials
// (lane-linear 1-KiB tiles), g = ((g_0 + g_1) + g_2) + g_3, Adam, p / m / v back, the new weights as bf16 into the shadow
// image (write-through: every slice's next forward reads it).  Returns the tile count of the pass.
template <int KS>
__device__ __forceinline__ int rs_sweep_pass(const Ctx& cc, const WgGeom& G, int base) {
  Ctx c = cc;
  relaunder(c);
  const nm_job_t* J = c.job;
  const WgT& T = G.T;
  const int KT = ktiles(G.K), kt0 = G.k_base >> 4;
  const int nkt = min((G.ncols + 15) >> 4, KT - kt0);          // k tiles of this pass that exist in the master
  const int ntn = (G.N + 15) >> 4;
  const int ntiles = ntn * nkt;
  const bool do_adam = (c.flags & NM_F_ADAM) != 0, do_grads = (c.flags & NM_F_GRADS) != 0;
  const AdamK ak = adam_consts(c);
  gf32 Pp = asg(J->params), Mp = asg(J->adam_m), Vp = asg(J->adam_v);
  const GAS float* const g0 = c.gpart - (int64_t)c.rsq * c.gp_stride;      // slice 0's partials
  const int stride = KS * NWAVES, gw = c.rsq * NWAVES + c.wave;
  const int prow = c.lane >> 2, pcol = (c.lane & 3) * 4;
  int t = gw - base % stride;
  t += t < 0 ? stride : 0;
  // two tiles per iteration: both tiles' loads are issued before the first store
  for (; t < ntiles; t += 2 * stride) {
    const int t1 = t + stride;
    const bool has1 = t1 < ntiles;
    int64_t idx[2];
    int nt[2], ktl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int tt = j ? (has1 ? t1 : t) : t;
      nt[j] = tt / nkt;                                          // wave-uniform
      ktl[j] = tt - nt[j] * nkt;
      idx[j] = T.w_off + ((int64_t)(nt[j] * KT + kt0 + ktl[j]) << 8) + c.lane * 4;
    }
    f32x4 gq[2][KS], p4[2], m4[2], v4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < KS; ++q) gq[j][q] = *(const GAS f32x4*)(g0 + (int64_t)q * c.gp_stride + idx[j]);
      if (do_adam) {
        p4[j] = *(const GAS f32x4*)(Pp + idx[j]);
        m4[j] = __builtin_nontemporal_load((const GAS f32x4*)(Mp + idx[j]));
        v4[j] = __builtin_nontemporal_load((const GAS f32x4*)(Vp + idx[j]));
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j == 1 && !has1) break;                                // wave-uniform
      f32x4 g = gq[j][0];
#pragma unroll
      for (int q = 1; q < KS; ++q) g += gq[j][q];                // slice order
      if (do_grads) *(GAS f32x4*)(asg(J->grads) + idx[j]) = g;
      if (do_adam) {
        f32x4 pn = p4[j], mn = m4[j], vn = v4[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) { float pp = pn[i], mm = mn[i], v2 = vn[i]; adam1(ak, g[i], pp, mm, v2); pn[i] = pp; mn[i] = mm; vn[i] = v2; }
        *(GAS f32x4*)(Pp + idx[j]) = pn;
        __builtin_nontemporal_store(mn, (GAS f32x4*)(Mp + idx[j]));
        __builtin_nontemporal_store(vn, (GAS f32x4*)(Vp + idx[j]));
        if (T.sh) {
          bf16x4 pk;
#pragma unroll
          for (int i = 0; i < 4; ++i) pk[i] = (__bf16)pn[i];
          st8_wt(T.sh + (int64_t)(nt[j] * 16 + prow) * T.sh_pitch + (ktl[j] * 16 + pcol) * 2, pk);
        }
      }
    }
  }
  return ntiles;
}

// One vector parameter segment: elements [idx0, idx0 + n) of the master (a bias, a chunk of logvar_out, alpha), with an
// optional fp32 copy inside a shadow image's vector piece.  All segments of a modality form one flat element range that
// is dealt thread-linear over the KS slices: one memory round trip for all of them.
struct VSeg { int64_t idx0; int n; GAS float* copy; };

template <int KS>
__device__ __forceinline__ void rs_sweep_vectors(const Ctx& cc, int m) {
  Ctx c = cc;
  relaunder(c);
  const nm_job_t* J = c.job;
  const nm_modality_t& md = J->mod[m];
  const int L = J->L, Me = experts(J);
  const int nck = (md.D + OCH - 1) / OCH;
  const bool sigm = J->out_kind == 1;
  const bool has_alpha = m < Me && md.alpha >= 0 && J->combine == NM_COMBINE_GPOE && !(Me == 1 && J->single_bypass);
  const bool do_adam = (c.flags & NM_F_ADAM) != 0, do_grads = (c.flags & NM_F_GRADS) != 0;
  // segment s (wave-uniform): 2 per output chunk (bias, logvar_out), decoder layers, then the encoder's
  const int n_out = 2 * nck, n_dec = L, n_enc = m < Me ? 2 + L : 0;          // heads (2), hidden layers (L - 1), first layer (1)
  const int nseg = n_out + n_dec + n_enc + (has_alpha ? 1 : 0);
  auto seg = [&](int s) -> VSeg {
    if (s < n_out) {
      const int ch = s >> 1, d0 = ch * OCH, valid = min(OCH, md.D - d0);
      GAS float* const vec = (GAS float*)((GAS char*)J->wsh + md.out_s + (int64_t)ch * OBLOB_BYTES + OIMG_BYTES);
      if (s & 1) return VSeg{md.logvar_out + d0, sigm ? 0 : valid, vec + OCH};
      return VSeg{md.out_b + d0, valid, vec};
    }
    s -= n_out;
    if (s < n_dec) { const WgGeom G = geom_dec(J, md, s, nullptr); return VSeg{G.T.b_off, G.N, G.T.sh_b}; }
    s -= n_dec;
    if (s < 2 && n_enc) { const WgGeom G = geom_head(J, md, s, nullptr); return VSeg{G.T.b_off, G.N, G.T.sh_b}; }
    s -= 2;
    if (s < L - 1 && n_enc) { const WgGeom G = geom_enc(J, md, s + 1, nullptr); return VSeg{G.T.b_off, G.N, G.T.sh_b}; }
    s -= L - 1;
    if (s == 0 && n_enc) { const WgGeom G = geom_l0(J, md, 0, nullptr); return VSeg{G.T.b_off, G.N, G.T.sh_b}; }
    return VSeg{md.alpha, 1, nullptr};
  };
  // this thread's element of the concatenation: walk the segments (wave-uniform loop, per-lane selects)
  const int e = c.rsq * WG + c.tid;                  // flat element index handled by this thread, then + KS * WG
  const GAS float* const g0 = c.gpart - (int64_t)c.rsq * c.gp_stride;
  gf32 Pp = asg(J->params), Mp = asg(J->adam_m), Vp = asg(J->adam_v);
  const AdamK ak = adam_consts(c);
  int total = 0;
  for (int s = 0; s < nseg; ++s) total += seg(s).n;
  for (int e0 = e; e0 - c.tid - c.rsq * WG < total; e0 += KS * WG) {      // (wave-uniform trip count)
    int64_t idx = -1;
    GAS float* copy = nullptr;
    int off = 0;
    for (int s = 0; s < nseg; ++s) {
      const VSeg sg = seg(s);
      const bool in = e0 >= off && e0 < off + sg.n;
      idx = in ? sg.idx0 + (e0 - off) : idx;
      copy = in ? (sg.copy ? sg.copy + (e0 - off) : (GAS float*)nullptr) : copy;
      off += sg.n;
    }
    if (idx >= 0) {
      float g = g0[idx];
#pragma unroll
      for (int q = 1; q < KS; ++q) g += g0[(int64_t)q * c.gp_stride + idx];
      if (do_grads) asg(J->grads)[idx] = g;
      if (do_adam) {
        float p = Pp[idx], mm = Mp[idx], v = Vp[idx];
        adam1(ak, g, p, mm, v);
        st4_wt(Pp + idx, p);                         // (alpha is read by every part; the others only through `copy`)
        Mp[idx] = mm; Vp[idx] = v;
        if (copy) st4_wt(copy, p);
      }
    }
  }
}

// All weight-gradient passes of modality m, in the order run_step issued them.
template <int KS>
__device__ __forceinline__ void rs_sweep(const Ctx& c, int m) {
  const nm_job_t* J = c.job;
  const nm_modality_t& md = J->mod[m];
  const int L = J->L, Me = experts(J);
  const int nck = (md.D + OCH - 1) / OCH;
  int base = 0;
  for (int ch = 0; ch < nck; ++ch) base += rs_sweep_pass<KS>(c, geom_out(J, md, ch, nullptr), base);
  for (int d = L - 1; d >= 0; --d) base += rs_sweep_pass<KS>(c, geom_dec(J, md, d, nullptr), base);
  if (m < Me) {
    base += rs_sweep_pass<KS>(c, geom_head(J, md, 0, nullptr), base);
    base += rs_sweep_pass<KS>(c, geom_head(J, md, 1, nullptr), base);
    for (int e = L - 1; e >= 1; --e) base += rs_sweep_pass<KS>(c, geom_enc(J, md, e, nullptr), base);
    const int nch = (md.Kx + XCH - 1) / XCH;
    for (int kc = 0; kc < nch; ++kc) base += rs_sweep_pass<KS>(c, geom_l0(J, md, kc, nullptr), base);
  }
  rs_sweep_vectors<KS>(c, m);
}

// ---- kernel ----------------------------------------------------------------------------------------------------------
// Grid: groups of KS workgroups = the row slices of one (job, modality), every group on ONE XCD (workgroups b and b + 8
// share an XCD -- observed placement, speed only: the partials of a group then meet in that XCD's L2; correctness comes from
// the hand-off protocol): workgroup b = ((slot * KS + q) << 3) + xcd runs slice q of group slot * 8 + xcd = job * M + m.
template <int KS>
__global__ __launch_bounds__(WG) void nm_rs_kernel(const nm_job_t* __restrict__ jobs, int step0, int n_steps, int flags,
                                                   int n_jobs, int M) {
  constexpr int RTV = 8 / KS;
  NM_GEOM(RTV);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int q = idx % KS, group = (idx / KS) * 8 + xcd;
  const int job_idx = group / M, part = group - job_idx * M;
  if (job_idx >= n_jobs) return;
  if ((flags & NM_F_FAULT_INJECT) && part == M - 1 && q == KS - 1) return;   // diagnostic: a workgroup that never arrives
  if ((flags & 64) && blockIdx.x < 512 && threadIdx.x == 0) nm_wg_times[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
  const nm_job_t* J = jobs + job_idx;
  Ctx c;
  c.job = J;
  c.part = part;
  c.nparts = M;
  c.slope = J->act_slope;
  carve_lds(c, smem);
  relaunder(c);
  c.flags = flags;
  c.t_last = 0;
  c.rsk = KS; c.rsq = q; c.rloc0 = q * ROWS; c.xwg = 1;
  c.ws0 = (GAS char*)J->workspace;
  c.ws = c.ws0 + (int64_t)q * J->workspace_stride;
  c.gp_stride = J->gpart_stride;
  c.gpart = (GAS float*)J->gpart + (int64_t)q * J->gpart_stride;
  for (int i = c.tid; i < SMEM_BYTES / 4; i += WG) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  __syncthreads();
  const WsLayout wl = ws_layout(J->M, J->L, J->Z);
  GAS unsigned* const sync0 = (GAS unsigned*)(c.ws0 + wl.sync);
  GAS unsigned* const sync_c = sync0 + WS_SYNC_C_WORD;
  GAS unsigned* const sync_d = sync0 + WS_SYNC_D_WORD + part;
  GAS unsigned* const sync_err = sync0 + WS_SYNC_ERR_WORD;
  const int nb = (J->n_rows + TROWS - 1) / TROWS;
  for (int s = step0; s < step0 + n_steps; ++s) {
    const int b = s % nb;
    c.lstep = s - step0;
    const int brows = min(TROWS, J->n_rows - b * TROWS);       // valid rows of the batch
    c.row0 = b * TROWS + c.rloc0;
    c.nrows = max(0, min(ROWS, brows - c.rloc0));               // ... of this slice (0: a slice past a ragged batch's end)
    c.inv_b = 1.0f / (float)brows;                              // means run over the whole batch
    const int64_t t_opt = J->adam_off + (int64_t)s + 1;
    const double tt = (double)t_opt;
    const double lr_t = (J->lr_table && J->lr_cap > 0) ? J->lr_table[(t_opt - 1) % J->lr_cap] : (double)J->lr;
    c.step_size = (float)(lr_t / (1.0 - pow((double)J->beta1, tt)));
    c.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)J->beta2, tt)));
    if (flags & 64) c.tlast[c.wave_s] = clock64();
    lds_barrier();
    relaunder(c);
    run_step<false, 0, RTV>(c, s);
    if (*c.abort != 0u) break;                                  // a hand-off timed out (wave-uniform: LDS word read by all)
    tr(c, 40);
    // C: every workgroup of the model has stored its partials and loss shares
    if (!split_handoff(c, sync_c, sync_err, (unsigned)(c.lstep + 1) * (unsigned)(M * KS))) break;
    tr(c, 41);
    if (part == 0 && q == 0 && c.tid == 0 && J->loss_log) {     // loss row: the slices' shares, in slice order
      gf32 row = asg(J->loss_log) + (int64_t)(s % J->loss_cap) * NM_LOSS_STRIDE;
      float kl = 0.f, ll_sum = 0.f;
      for (int qq = 0; qq < KS; ++qq)
        kl += ((const GAS float*)(c.ws0 + (int64_t)qq * J->workspace_stride + wl.sync))[WS_SYNC_LOSS_WORD];
      for (int m = 0; m < M; ++m) {
        float ll = 0.f;
        for (int qq = 0; qq < KS; ++qq)
          ll += ((const GAS float*)(c.ws0 + (int64_t)qq * J->workspace_stride + wl.sync))[WS_SYNC_LOSS_WORD + 1 + m];
        row[NM_LOSS_LL_M + m] = ll;
        ll_sum += ll;
      }
      row[NM_LOSS_KL] = J->kl_weight * kl;
      row[NM_LOSS_LL] = ll_sum;
      row[NM_LOSS_TC] = 0.f;
      row[NM_LOSS_TOTAL] = J->kl_weight * kl - J->ll_weight * ll_sum;
    }
    rs_sweep<KS>(c, part);
    tr(c, 42);
    // D: the modality's new shadow images / vector pieces are complete (the next forward of every slice reads them)
    if (!split_handoff(c, sync_d, sync_err, (unsigned)(c.lstep + 1) * (unsigned)KS)) break;
    tr(c, 43);
  }
  if ((flags & 64) && blockIdx.x < 512 && c.tid == 0) nm_wg_times[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace

extern "C" {

/* Row-split launch (include/nmhip.h): n_jobs models of M modalities each, k in {2, 4} row slices per (model, modality).
 * Every job needs k workspace tiles and gpart / gpart_stride; -16: the launch would not be resident at once;
 * -20: a job of the launch cannot run row-split (see nm_rowsplit_ok). */
int nm_launch_rowsplit(const nm_job_t* jobs_dev, int n_jobs, int M, int k, int step0, int n_steps, int flags, void* stream) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1 || n_steps < 1 || step0 < 0 || M < 1 || M > NM_MAX_EXP || (k != 2 && k != 4)) return -8;
  if (!(flags & NM_F_BACKWARD) || !(flags & (NM_F_ADAM | NM_F_GRADS))) return -8;
  if ((flags & NM_F_GRADS) && n_steps != 1) return -8;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return -8;
  const int groups = (n_jobs * M + 7) / 8 * 8;
  const int wgs = groups * k;
  if (wgs > cus) return -16;            // the workgroups of a model wait for each other: all must be resident
  hipStream_t st = (hipStream_t)stream;
  nm_sync_reset(jobs_dev, n_jobs, stream);
  flags &= (NM_F_BACKWARD | NM_F_ADAM | NM_F_GRADS | NM_F_TRACE | NM_F_FAULT_INJECT);
  hipError_t e;
  if (k == 2) {
    e = hipFuncSetAttribute((const void*)nm_rs_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nm_rs_kernel<2>, dim3(wgs), dim3(WG), SMEM_BYTES, st, jobs_dev, step0, n_steps, flags, n_jobs, M);
  } else {
    e = hipFuncSetAttribute((const void*)nm_rs_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nm_rs_kernel<4>, dim3(wgs), dim3(WG), SMEM_BYTES, st, jobs_dev, step0, n_steps, flags, n_jobs, M);
  }
  return (int)hipGetLastError();
}

/* NM_F_TRACE read-out of the row-split kernels (this translation unit has its own copy of the timers). */
int nm_trace_read_rs(unsigned long long* out512, int reset) {
  if (!out512) return -1;
  hipError_t e = hipMemcpyFromSymbol(out512, HIP_SYMBOL(nm_trace_cycles), sizeof(unsigned long long) * 512);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    static unsigned long long z[512];
    e = hipMemcpyToSymbol(HIP_SYMBOL(nm_trace_cycles), z, sizeof(z));
  }
  return (int)e;
}

}  // extern "C"
