// nm_devpass.hip -- the ROI-wise deviation pass as its own kernel (libnmhip.so, third translation unit).
//
// The pass of multimodal_kfold_train_cvae_supervised_regression.py:163-192 (and utils_vae.py:147-152): per modality the
// unimodal encoder -> sampled z -> decoder, (x - x_hat)^2 per ROI and its row mean, for ALL subjects.  Forward only, one
// expert, nothing saved: the general forward-only step (nm_step_kernel<false, 3>) spends a 256-row tile's 178 k cycles
// mostly waiting -- eight dependent phases of one 8-wave workgroup, each behind a staged weight image and a barrier --
// with the CU's other half idle, because its LDS plan (P, Q as [256][136], S) takes 159 of 160 KB.
//
// Here a workgroup owns 128 rows (RT = 4) and 75 KB of LDS, so TWO workgroups share a CU and fill each other's waits:
//   P  [128][136] bf16  the running activation
//   W  [128][136] bf16  ONE weight image at a time (requested as soon as the layer before it has drained W; the other
//                        workgroup on the CU works meanwhile), or two first-layer stages together with P, or the two
//                        [64][136] output-chunk slots (their bias / logvar_out pieces go to the two vector slots), or --
//                        beside the 32-row heads image -- the sampled z as bf16 [128][32]
// 16-row MFMA tiles at or beyond the tile's valid rows are skipped in every phase (a 1064-row table costs 1064 rows
// rounded to 16, not 5 x 256), and the export epilogue makes one pass: residual, its square (stored), row sum.
// Arithmetic, draws (keyed by absolute row) and exports are those of the general kernel, row by row: the two agree bit for
// bit on out_sqerr / out_rowdev / out_loc (tests/test_gpu_devpass.py).  No loss log, no latent exports: launches that want
// those, models with several experts, a first hidden layer wider than 112 or a latent wider than 32 stay on nm_forward.
#include "nm_core.inc"

// export stores: plain.  NM_DV_NT = 1 makes them non-temporal ("written once, read by another kernel") for the A/B that
// decided it: 458 us per pass non-temporal against 328 us plain (profiles/r04k_ab_devpass_export_stores.txt) -- a row's
// 256 bytes of a chunk leave as four 64-byte stores; plain stores merge in L2 into full lines before they go to memory,
// the streaming policy sent them on as partial lines.
#ifndef NM_DV_NT
#define NM_DV_NT 0
#endif
#if NM_DV_NT
#define NM_DV_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define NM_DV_STORE(v, p) (*(p) = (v))
#endif
constexpr int DV_RT = 4;
constexpr int DV_ROWS = DV_RT * 32;                                   // 128
constexpr int DV_P_BYTES = DV_ROWS * LDP * 2;                         // 34,816
constexpr int DV_W_BYTES = IMG_BYTES;                                 // 34,816
constexpr int DV_X_PIECES = (DV_ROWS * LDX * 2) >> 10;                // 18: a [128][72] x chunk
constexpr int DV_W0_PIECES = (DV_P_BYTES >> 10) - DV_X_PIECES;        // 16: what is left of a stage for the weight chunk
constexpr int DV_MAX_H0 = (DV_W0_PIECES << 10) / (LDX * 2) / 16 * 16; // 112 rows
constexpr int DV_Z_OFF = 64 * LDP * 2;                                // z [128][32] bf16 inside W, behind the (<= 64-row) heads image
constexpr int DV_MISC_FLOATS = 64 + 128 + 128 + 16 + 4;
constexpr int DV_SMEM = DV_P_BYTES + DV_W_BYTES + 2 * VEC_BYTES + DV_MISC_FLOATS * 4;
static_assert(2 * DV_SMEM <= 160 * 1024, "two workgroups per CU");
static_assert(2 * OIMG_BYTES <= DV_W_BYTES && DV_Z_OFF % 16 == 0 && DV_Z_OFF + DV_ROWS * 32 * 2 <= DV_W_BYTES, "W layout");

__device__ __forceinline__ void dv_carve(Ctx& c, unsigned char* smem) {
  c.wave_s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  c.rsk = 1; c.rsq = 0; c.rloc0 = 0; c.gpart = nullptr; c.gp_stride = 0; c.ws0 = nullptr; c.xwg = 0; c.gwt = 0;
  c.P = reinterpret_cast<__bf16*>(smem);
  c.Q = c.P + DV_ROWS * LDP;                                         // = W
  c.stage = nullptr;                                                  // (no S in this plan)
  c.vec = reinterpret_cast<float*>(smem + DV_P_BYTES + DV_W_BYTES);
  c.red = c.vec + 2 * (VEC_BYTES / 4);
  c.colacc = c.red + 64;
  c.rowacc = c.colacc + 128;
  c.lse = nullptr; c.bgrad = nullptr;
  c.tlast = reinterpret_cast<unsigned long long*>(c.rowacc + 128);
  c.abort = reinterpret_cast<unsigned*>(c.tlast + 8);
}

// One hidden layer, P -> P in place, image in W (requested by the phase before, its vector piece in slot `vs`): wait, GEMM
// over the live row tiles, then -- W drained -- request `nx` into W, then the activation epilogue.
__device__ __forceinline__ void dv_layer(const Ctx& cc, int vs, const Next& nx, int N, int K, bool act, int live) {
  constexpr int RT = DV_RT, WROWS = DV_RT * 16;
  Ctx c = cc;
  relaunder(c);
  const int ksteps = wpad(K) / 32, ntn = wpad(N) / 16;
  wait_vm(0);
  lds_barrier();
  f32x4 acc[2][RT];
  zero_acc(acc);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (ks < ksteps) {
      bf16x8 wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) wf[t] = lds_frag(c.Q, LDP, (c.wn + 4 * t) * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        if (c.wm * WROWS + rt * 16 < live) {
          bf16x8 a = lds_frag(c.P, LDP, c.wm * WROWS + rt * 16 + c.c16, ks * 32 + 8 * c.g);
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t][rt] = mfma(wf[t], a, acc[t][rt]);
        }
      }
    }
  }
  lds_barrier();                       // every wave has finished reading P and W
  issue_next(c, nx);
  act_to_P(c, acc, c.vec + vs * (VEC_BYTES / 4), N, ntn, act);
}

__global__ __launch_bounds__(WG, 4) void nm_devpass_kernel(const nm_job_t* __restrict__ jobs, int tile0, int flags) {
  constexpr int RT = DV_RT, ROWS = DV_ROWS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const nm_job_t* J = jobs + blockIdx.x;
  const int t128 = tile0 + (int)blockIdx.y;
  const int row0 = t128 * ROWS;
  if (row0 >= J->n_rows) {
    // the second half of a ragged last 256-row tile: its export rows come back as zeros, as from the general kernel
    const nm_modality_t& m0 = J->mod[0];
    const int xp0 = m0.x_pitch;
    if (row0 < (J->n_rows + TROWS - 1) / TROWS * TROWS) {
      for (int e = threadIdx.x; e < ROWS * (xp0 >> 2); e += WG) {
        const int64_t gi = (int64_t)row0 * xp0 + (int64_t)e * 4;
        if (m0.out_loc) *(GAS f32x4*)(asg(m0.out_loc) + gi) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m0.out_sqerr) *(GAS f32x4*)(asg(m0.out_sqerr) + gi) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    return;
  }
  Ctx c;
  c.job = J;
  c.part = -1; c.nparts = 1; c.lstep = 0;
  c.slope = J->act_slope;
  dv_carve(c, smem);
  relaunder(c);
  c.flags = NM_F_EXPORT | (flags & NM_F_TRACE);
  c.t_last = 0;
  c.ws = nullptr;
  for (int i = c.tid; i < DV_SMEM / 4; i += WG) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  __syncthreads();
  if (c.flags & 64) c.tlast[c.wave_s] = clock64();
  c.row0 = row0;
  c.rloc0 = row0 % TROWS;                         // (the draw buffer holds whole 256-row batches)
  c.nrows = min(ROWS, J->n_rows - row0);
  c.inv_b = 1.0f / (float)c.nrows;
  const int live = c.nrows;
  const int step = row0 / TROWS;                  // the batch this tile belongs to = the general kernel's step index
  const nm_modality_t& md = J->mod[0];
  const int L = J->L, Z = J->Z, C = J->C, D = md.D;
  const int Zs = rup(Z, 16);
  const bool nl = J->non_linear != 0;
  const bool vec4 = (Z & 3) == 0;
  GAS char* const wsh = (GAS char*)J->wsh;
  char* const Wb = reinterpret_cast<char*>(c.Q);
  __bf16* const zlds = reinterpret_cast<__bf16*>(Wb + DV_Z_OFF);
  auto to_W = [&](const GAS char* blob, int vs, int rows, int K) {
    return Next{blob, Wb, IMG_BYTES >> 10, blob + cimg_bytes(rows, K), reinterpret_cast<char*>(c.vec) + vs * VEC_BYTES, rows, blob_kp(K)};
  };

  // ---- encoder ----
  const int nch = (md.Kx + XCH - 1) / XCH;
  const GAS char* after0 = wsh + (L > 1 ? md.enc_s[1] : md.heads_s);
  fwd_first_layer<RT>(c, (const GAS char*)asg(md.xb) + (int64_t)(row0 / TROWS) * nch * XIMG_TILE_BYTES + (int64_t)c.rloc0 * (LDX * 2),
                      md.Kx, wsh + md.enc_s[0], to_W(after0, 0, L > 1 ? J->H[1] : 2 * Zs, J->H[0]), J->H[0], nl, (gbf16)nullptr, true,
                      DV_W0_PIECES, live);
  int vs = 0;
  for (int e = 1; e < L; ++e) {
    const GAS char* nxt = wsh + (e + 1 < L ? md.enc_s[e + 1] : md.heads_s);
    dv_layer(c, vs, to_W(nxt, vs ^ 1, e + 1 < L ? J->H[e + 1] : 2 * Zs, J->H[e]), J->H[e], J->H[e - 1], nl, live);
    vs ^= 1;
  }
  tr(c, 1);
  // heads + the latent draw in their epilogue (z -> W behind the heads image; nothing requested meanwhile: W is in use)
  wait_vm(0);
  float kl_unused = 0.f;
  fwd_heads<RT>(c, 0, no_next(), Z, J->H[L - 1], (gf32)nullptr, (gf32)nullptr, Zs, 0, zlds, step, vec4, &kl_unused,
                c.vec + vs * (VEC_BYTES / 4));
  tr(c, 2);
  // ---- decoder ----
  relaunder(c);
  build_zc<RT>(c, c.P, md, (gcf32)nullptr, (gcf32)nullptr, Z, C, Zs, 0, (gcf32)nullptr, zlds);
  lds_barrier();                                   // z is consumed: W is free
  issue_next(c, to_W(wsh + md.dec_s[0], 0, J->H[L - 1], Z + C));
  tr(c, 4);
  vs = 0;
  const GAS char* oblob = wsh + md.out_s;
  const int nck = (D + OCH - 1) / OCH;
  // chunk ch: its [64][136] rows into half ch & 1 of W, its vectors into slot (ch + ob) & 1 -- ob such that chunk 0's vectors
  // do not land on the bias the last hidden layer's epilogue is still reading (L layers: that one sits in slot (L - 1) & 1)
  const int ob = L & 1;
  auto out_blob = [&](int ch) {
    return Next{oblob + (int64_t)ch * OBLOB_BYTES, Wb + (ch & 1) * OIMG_BYTES, OIMG_BYTES >> 10,
                oblob + (int64_t)ch * OBLOB_BYTES + OIMG_BYTES, reinterpret_cast<char*>(c.vec) + ((ch + ob) & 1) * VEC_BYTES, 0, 0};
  };
  for (int d = 0; d < L; ++d) {
    const int Kin = (d == 0) ? Z + C : J->H[L - d], Nout = J->H[L - 1 - d];
    const Next nx = (d + 1 < L) ? to_W(wsh + md.dec_s[d + 1], vs ^ 1, J->H[L - 2 - d], Nout) : out_blob(0);
    dv_layer(c, vs, nx, Nout, Kin, nl, live);
    vs ^= 1;
  }
  tr(c, 5);
  // ---- output layer in 64-ROI chunks: x_hat, (x - x_hat)^2, row sums ----
  // Wave grid of this phase: wave w owns rows [16 w, 16 w + 16) of the tile and ALL 64 columns of a chunk (four feature
  // tiles): a row's 256 bytes of a chunk are then stored by four consecutive instructions of ONE wave (64 bytes each),
  // which the memory system merges into full lines -- with the GEMM phases' 2 x 4 grid the two halves of every 128-byte
  // line came from two waves at different times, and the pass was bound by partial-line writes.  A row's sum needs no
  // cross-wave step either.  Same MFMA and LDS-read counts as before (one P fragment and four weight fragments per k step).
  const int Hl = J->H[0];
  gcf32 xf = asg(md.x_f32);
  const int xp = md.x_pitch;
  const bool sigm = J->out_kind == 1;
  float rdev = 0.f;
  const int orow = c.wave * 16 + c.c16;            // this lane's row of the tile
  const bool wave_live = c.wave * 16 < live;       // (wave-uniform: a dead 16-row tile's wave only keeps the barriers)
  auto load_xin = [&](int chx, f32x4 (&xv)[4]) {
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const int dcl = min(chx * OCH + ft * 16 + 4 * c.g, xp - 4);
      xv[ft] = *(const GAS f32x4*)(xf + (int64_t)(row0 + orow) * xp + dcl);
    }
  };
  int stores_prev = 0;                             // export stores this wave issued in the previous chunk (wave-uniform)
  auto chunk = [&](const int ch, f32x4 (&xin)[4], f32x4 (&xnx)[4]) {
    relaunder(c);
    const int d0 = ch * OCH;
    const __bf16* Wc = reinterpret_cast<const __bf16*>(Wb + (ch & 1) * OIMG_BYTES);
    const float* vb = c.vec + ((ch + ob) & 1) * (VEC_BYTES / 4);       // bias[64], then logvar_out[64]
    // chunk ch's rows, vectors and inputs (requested a chunk ago) have landed; the previous chunk's export stores, issued
    // after them, may stay in flight
    wait_vm(ch > 0 ? stores_prev : 0);
    lds_barrier();                                  // ... for every wave; the previous chunk is finished everywhere
    if (ch + 1 < nck) { issue_next(c, out_blob(ch + 1)); if (wave_live) load_xin(ch + 1, xnx); }
    stores_prev = 0;
    if (!wave_live) return;
    f32x4 acc[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) acc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ksteps = wpad(Hl) / 32;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < ksteps) {
        const bf16x8 a = lds_frag(c.P, LDP, orow, ks * 32 + 8 * c.g);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) acc[ft] = mfma(lds_frag(Wc, LDP, ft * 16 + c.c16, ks * 32 + 8 * c.g), a, acc[ft]);
      }
    }
    tr(c, 6);
    const bool rv = orow < c.nrows;
    const bool full = live == ROWS && d0 + OCH <= D && !sigm;          // no row / column masks needed (wave-uniform)
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const int dg0 = d0 + ft * 16 + 4 * c.g;
      const f32x4 bo = *reinterpret_cast<const f32x4*>(vb + ft * 16 + 4 * c.g);
      f32x4 lo, sq;
      if (full) {
        lo = acc[ft] + bo;
        const f32x4 diff = lo - xin[ft];
        sq = diff * diff;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float xh = acc[ft][i] + bo[i];
          if (sigm) xh = 1.0f / (1.0f + expf(-xh));
          const bool dv = rv && dg0 + i < D;
          const float diff = xh - xin[ft][i];
          lo[i] = dv ? xh : 0.f;
          sq[i] = dv ? diff * diff : 0.f;
        }
      }
      rdev += ((sq[0] + sq[1]) + sq[2]) + sq[3];
      // (whole tiles are stored, zeros on the rows past the table's end -- the export buffers hold whole 256-row tiles)
      if (d0 + ft * 16 < xp) {                      // wave-uniform
        if (dg0 < xp) {
          const int64_t gi = (int64_t)(row0 + orow) * xp + dg0;
          if (md.out_loc) NM_DV_STORE(lo, (GAS f32x4*)(asg(md.out_loc) + gi));
          if (md.out_sqerr) NM_DV_STORE(sq, (GAS f32x4*)(asg(md.out_sqerr) + gi));
        }
        stores_prev += (md.out_loc ? 1 : 0) + (md.out_sqerr ? 1 : 0);
      }
    }
    tr(c, 7);
  };
  {
    f32x4 xa[4], xb[4];
    if (wave_live) load_xin(0, xa);
    for (int ch = 0; ch < nck; ch += 2) {
      chunk(ch, xa, xb);
      if (ch + 1 < nck) chunk(ch + 1, xb, xa);
    }
  }
  if (md.out_rowdev && wave_live) {                 // the row's four column groups sit in the lanes c16, c16 + 16, + 32, + 48
    float v = rdev;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (c.g == 0 && orow < c.nrows) asg(md.out_rowdev)[row0 + orow] = v / (float)D;
  }
  tr(c, 8);
}

}  // namespace

extern "C" {

/* 0: the job's deviation pass can run on the compact kernel (one expert with the single-expert bypass, no private latent /
 * learnable weights / total correlation, first hidden width <= 112, latent <= 32, Gaussian output); -22 otherwise. */
int nm_devpass_ok(const nm_job_t* j) {
  if (!j) return -1;
  const int Me = j->M_enc > 0 ? j->M_enc : j->M;
  if (j->wide || j->M != 1 || Me != 1 || !j->single_bypass || j->n_private != 0 || j->tc_weight != 0.f || j->w_off >= 0) return -22;
  if (j->H[0] > DV_MAX_H0 || rup(j->Z, 16) > 32 || j->out_kind != 0) return -22;
  return 0;
}

/* The ROI-wise deviation pass (multimodal_kfold_train_cvae_supervised_regression.py:163-192) over table rows
 * [tile0 * 128, (tile0 + n_tiles) * 128): out_sqerr / out_rowdev / out_loc of modality 0, nothing else (no loss log, no
 * latent exports).  Every job must pass nm_devpass_ok. */
int nm_devpass(const nm_job_t* jobs_dev, int n_jobs, int tile0, int n_tiles, int flags, void* stream) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1 || n_tiles < 1 || tile0 < 0) return -8;
  hipError_t e = hipFuncSetAttribute((const void*)nm_devpass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DV_SMEM);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nm_devpass_kernel, dim3(n_jobs, n_tiles), dim3(WG), DV_SMEM, (hipStream_t)stream, jobs_dev, tile0,
                     flags & NM_F_TRACE);
  return (int)hipGetLastError();
}

/* NM_F_TRACE read-out of nm_devpass ([8 waves][64 tags] interval cycles of workgroup (0, 0), as nm_trace_read) */
int nm_trace_read_dv(unsigned long long* out512, int reset) {
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
