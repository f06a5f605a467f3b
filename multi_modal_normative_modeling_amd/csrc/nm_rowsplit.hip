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
// All weight-gradient passes of a modality (the geometries run_step used) form ONE stream of 16 x 16 master tiles, dealt
// round-robin over the KS * 8 waves of the modality's KS slices (the same tile always goes to the same wave of the same
// workgroup: p / m / v of a tile are private to it).  Per tile: p / m / v and the KS partials (lane-linear 1-KiB tiles),
// g = ((g_0 + g_1) + g_2) + g_3, Adam, p / m / v back, the new weights as bf16 into the shadow image (write-through: every
// slice's next forward reads it).
//
// A wave keeps SW_DEPTH tiles in flight: the requests are hand-issued (asm, destination tied, always 3 + KS per tile) and
// hand-waited with a counted vmcnt -- per pass a wave has only a tile or two, so a loop per pass with the compiler's
// waits costs one exposed memory round trip per pass (the first version: 90 k cycles of a 345 k-cycle step).  The passes'
// geometries sit in a small table in S (all of P / Q / S is dead during the sweep), built by one thread per pass.
#ifndef NM_SW_DEPTH
#define NM_SW_DEPTH 4
#endif
constexpr int SW_DEPTH = NM_SW_DEPTH;
constexpr int SW_NV = 3;                 // vector elements per thread (biases, logvar_out, alpha: 2 D + ... per modality)
struct SwRec { int w_off, KT, kt0, nkt; float rnkt; int sh_pitch; GAS char* sh; };     // 32 bytes
// One vector parameter segment: elements [idx0, idx0 + n) of the master (a bias, a chunk of logvar_out, alpha), with an
// optional fp32 copy inside a shadow image's vector piece; `off` = its first element in the concatenation of all segments
// of the modality, which is dealt thread-linear over the KS slices: one memory round trip for all of them.
struct SwVec { int idx0, n, off, pad; GAS float* copy; };                                 // 24 bytes
// The sweep's tables of one (job, modality): built once per launch (rs_build_tables) into the slice's workspace tile
// (WsLayout::rs_tab), copied into S at the start of every sweep (the step's phases use S in between).
struct SwTab {
  int npass, ntot, nseg, vtot;
  int tbase[NM_RS_MAX_PASSES + 1];        // running tile count of the passes
  int pad_[3];
  SwRec rec[NM_RS_MAX_PASSES];
  SwVec vec[NM_RS_MAX_VSEGS];
};
static_assert(sizeof(SwTab) % 16 == 0 && (int)sizeof(SwTab) <= STAGE_FLOATS * 4 && (int)sizeof(SwTab) <= WS_RS_TAB_BYTES, "sweep tables fit S and their workspace slot");

// pass number -> geometry, in the order the NEXT forward reads the shadow images the sweep rewrites: the first encoder layer's
// chunks -- read right behind hand-off D -- are the sweep's oldest stores (swept last, as the backward issues the passes,
// the first chunk waited ~2 us longer for lines still on their way)
__device__ __forceinline__ WgGeom sw_geom(const nm_job_t* J, int m, int g) {
  const nm_modality_t& md = J->mod[m];
  const int L = J->L, nck = (md.D + OCH - 1) / OCH, nch = (md.Kx + XCH - 1) / XCH;
  if (m < experts(J)) {
    if (g < nch) return geom_l0(J, md, g, nullptr);
    g -= nch;
    if (g < L - 1) return geom_enc(J, md, 1 + g, nullptr);
    g -= L - 1;
    if (g < 2) return geom_head(J, md, g, nullptr);
    g -= 2;
  }
  if (g < L) return geom_dec(J, md, g, nullptr);
  g -= L;
  return geom_out(J, md, g, nullptr);
}
// vector segment number -> segment: per output chunk its bias and its logvar_out columns, the decoder layers' biases, then
// the encoder's (heads, hidden layers, first layer) and alpha
__device__ __forceinline__ SwVec sw_vseg(const nm_job_t* J, int m, int s) {
  const nm_modality_t& md = J->mod[m];
  const int L = J->L, nck = (md.D + OCH - 1) / OCH;
  if (s < 2 * nck) {
    const int ch = s >> 1, d0 = ch * OCH, valid = min(OCH, md.D - d0);
    GAS float* const vec = (GAS float*)((GAS char*)J->wsh + md.out_s + (int64_t)ch * OBLOB_BYTES + OIMG_BYTES);
    if (s & 1) return SwVec{(int)md.logvar_out + d0, J->out_kind == 1 ? 0 : valid, 0, 0, vec + OCH};
    return SwVec{(int)md.out_b + d0, valid, 0, 0, vec};
  }
  s -= 2 * nck;
  WgGeom G;
  if (s < L) G = geom_dec(J, md, s, nullptr);
  else if (s < L + 2) G = geom_head(J, md, s - L, nullptr);
  else if (s < 2 * L + 1) G = geom_enc(J, md, s - L - 1, nullptr);
  else if (s == 2 * L + 1) G = geom_l0(J, md, 0, nullptr);
  else return SwVec{(int)md.alpha, 1, 0, 0, nullptr};
  return SwVec{(int)G.T.b_off, G.N, 0, 0, G.T.sh_b};
}

// Once per launch: the tables of modality m into S, from there into the workspace slot `gtab`.
__device__ __forceinline__ void rs_build_tables(const Ctx& cc, int m, GAS char* gtab) {
  Ctx c = cc;
  relaunder(c);
  const nm_job_t* J = c.job;
  const nm_modality_t& md = J->mod[m];
  SwTab* const tab = reinterpret_cast<SwTab*>(c.stage);
  const int L = J->L, Me = experts(J);
  const int nck = (md.D + OCH - 1) / OCH, nch = (md.Kx + XCH - 1) / XCH;
  const bool enc = m < Me;
  const bool has_alpha = enc && md.alpha >= 0 && J->combine == NM_COMBINE_GPOE && !(Me == 1 && J->single_bypass);
  const int npass = min(nck + L + (enc ? 2 + (L - 1) + nch : 0), NM_RS_MAX_PASSES);
  const int nseg = min(2 * nck + L + (enc ? 2 + L : 0) + (has_alpha ? 1 : 0), NM_RS_MAX_VSEGS);
  for (int g = c.tid; g < npass; g += WG) {
    const WgGeom G = sw_geom(J, m, g);
    const int KT = ktiles(G.K), kt0 = G.k_base >> 4;
    const int nkt = min((G.ncols + 15) >> 4, KT - kt0);                      // k tiles of the pass that exist in the master
    tab->rec[g] = SwRec{(int)G.T.w_off, KT, kt0, nkt, 1.0f / (float)nkt, G.T.sh_pitch, G.T.sh};
    tab->tbase[g + 1] = ((G.N + 15) >> 4) * nkt;
  }
  for (int sI = c.tid; sI < nseg; sI += WG) tab->vec[sI] = sw_vseg(J, m, sI);   // (a decoder-only modality ends after its decoder layers)
  lds_barrier();
  if (c.tid == 0) {
    int acc = 0;
    tab->tbase[0] = 0;
    for (int g = 0; g < npass; ++g) { acc += tab->tbase[g + 1]; tab->tbase[g + 1] = acc; }
    int off = 0;
    for (int sI = 0; sI < nseg; ++sI) { tab->vec[sI].off = off; off += tab->vec[sI].n; }
    tab->npass = npass; tab->ntot = acc; tab->nseg = nseg; tab->vtot = off;
  }
  lds_barrier();
  for (int i = c.tid; i < (int)sizeof(SwTab) / 16; i += WG)       // (write-through: the helpers read slice 0's copy)
    st16_wt(gtab + i * 16, reinterpret_cast<const f32x4*>(tab)[i]);
  handoff_barrier();
}

// The sweep is dealt over KH >= KS workgroups: the KS row slices of the modality and, when the set leaves CUs idle, KH - KS
// HELPER workgroups that take no part in the step itself (c.rsq = this workgroup's index among the KH).
template <int KS>
// (tpre: this thread's 16 bytes of the tables, requested by the caller BEFORE hand-off C so that the round trip to the
//  workspace -- a step's traffic has pushed the tables out of the L2 -- runs under the wait)
__device__ __forceinline__ void rs_sweep(const Ctx& cc, int m, u32x4 tpre, int KH) {
  Ctx c = cc;
  relaunder(c);
  const nm_job_t* J = c.job;
  SwTab* const tab = reinterpret_cast<SwTab*>(c.stage);
  static_assert((int)sizeof(SwTab) / 16 <= WG, "one 16-byte piece of the tables per thread");
  lds_barrier();                                   // S is drained by whatever ran before
  if (c.tid < (int)sizeof(SwTab) / 16) reinterpret_cast<u32x4*>(tab)[c.tid] = tpre;
  lds_barrier();
  const SwRec* const rec = tab->rec;
  const int* const tbase = tab->tbase;
  const int ntot = tab->ntot;
  tr(c, 44);
  const bool do_adam = (c.flags & NM_F_ADAM) != 0, do_grads = (c.flags & NM_F_GRADS) != 0;
  const AdamK ak = adam_consts(c);
  gf32 Pp = asg(J->params), Mp = asg(J->adam_m), Vp = asg(J->adam_v);
  const GAS float* const g0 = asg((const float*)J->gpart);                   // slice 0's partials
  const unsigned gstride_b = (unsigned)(c.gp_stride << 2);                  // (k * gpart_stride * 4 < 2^32: nm_rowsplit_ok)
  const int stride = KH * NWAVES;
  const int prow = c.lane >> 2, pcol = (c.lane & 3) * 4;
  const unsigned lane16 = (unsigned)c.lane << 4;
  constexpr int L = 3 + KS;                                                  // vector-memory operations of one request
  const int S_lb = (do_adam ? 4 : 0) + (do_grads ? 1 : 0);                   // stores of one finished tile

  // ---- the vector segments first: up to SW_NV elements per thread, requested now, finished after the tile stream ----
  // (vtot <= 2 * WG * SW_NV: nm_rowsplit_ok)
  int64_t vidx[SW_NV];
  GAS float* vcopy[SW_NV];
  float vg[SW_NV][KS], vp[SW_NV], vm[SW_NV], vv[SW_NV];
#pragma unroll
  for (int i = 0; i < SW_NV; ++i) {
    const int e0 = (i * KH + c.rsq) * WG + c.tid;
    vidx[i] = -1; vcopy[i] = nullptr; vp[i] = 0.f; vm[i] = 0.f; vv[i] = 0.f;
#ifdef NM_RS_NO_VEC
    if (false) {
#else
    if (e0 < tab->vtot) {
#endif
      int sgi = 0;
      for (int sI = 1; sI < tab->nseg; ++sI) sgi += (e0 >= tab->vec[sI].off) ? 1 : 0;
      const SwVec sg = tab->vec[sgi];
      vidx[i] = sg.idx0 + (e0 - sg.off);
      vcopy[i] = sg.copy ? sg.copy + (e0 - sg.off) : (GAS float*)nullptr;
    }
    const int64_t li = vidx[i] >= 0 ? vidx[i] : 0;
#pragma unroll
    for (int q = 0; q < KS; ++q) vg[i][q] = g0[(int64_t)q * c.gp_stride + li];
    if (do_adam) { vp[i] = Pp[li]; vm[i] = Mp[li]; vv[i] = Vp[li]; }
  }

  struct Slot { f32x4 p, m, v, g[KS]; unsigned boff; GAS char* shp; };
  Slot sl[SW_DEPTH];
#pragma unroll
  for (int s = 0; s < SW_DEPTH; ++s) {
    asm volatile("" : "=v"(sl[s].p), "=v"(sl[s].m), "=v"(sl[s].v));
#pragma unroll
    for (int q = 0; q < KS; ++q) asm volatile("" : "=v"(sl[s].g[q]));
  }
  int gi = 0;                                      // pass cursor of the NEXT tile to request (tiles come in ascending order)
  int Tn = c.rsq * NWAVES + c.wave;                // next tile of this wave
  auto issue = [&](Slot& z) {
    while (Tn >= tbase[gi + 1]) ++gi;
    const SwRec r = rec[gi];
    const int lt = Tn - tbase[gi];
    const int nt = idiv(lt, r.nkt, r.rnkt), ktl = lt - nt * r.nkt;
    z.boff = ((unsigned)(r.w_off + ((nt * r.KT + r.kt0 + ktl) << 8)) << 2) + lane16;
    z.shp = r.sh + (int64_t)(nt * 16 + prow) * r.sh_pitch + (ktl * 16 + pcol) * 2;
    NM_GLOAD16(z.p, z.boff, Pp); NM_GLOAD16_NT(z.m, z.boff, Mp); NM_GLOAD16_NT(z.v, z.boff, Vp);
#pragma unroll
    for (int q = 0; q < KS; ++q) { const unsigned o = z.boff + (unsigned)q * gstride_b; NM_GLOAD16(z.g[q], o, g0); }
    Tn += stride;
  };
  auto finish = [&](Slot& z) {
    asm volatile("" : "+v"(z.p), "+v"(z.m), "+v"(z.v));
#pragma unroll
    for (int q = 0; q < KS; ++q) asm volatile("" : "+v"(z.g[q]));
    f32x4 g = z.g[0];
#pragma unroll
    for (int q = 1; q < KS; ++q) g += z.g[q];      // slice order
    const int64_t idx = (int64_t)(z.boff >> 2);
    if (do_grads) *(GAS f32x4*)(asg(J->grads) + idx) = g;
    if (do_adam) {
      f32x4 pn = z.p, mn = z.m, vn = z.v;
#pragma unroll
      for (int i = 0; i < 4; ++i) { float pp = pn[i], mm = mn[i], v2 = vn[i]; adam1(ak, g[i], pp, mm, v2); pn[i] = pp; mn[i] = mm; vn[i] = v2; }
      *(GAS f32x4*)(Pp + idx) = pn;
      __builtin_nontemporal_store(mn, (GAS f32x4*)(Mp + idx));
      __builtin_nontemporal_store(vn, (GAS f32x4*)(Vp + idx));
      bf16x4 pk;
#pragma unroll
      for (int i = 0; i < 4; ++i) pk[i] = (__bf16)pn[i];
      st8_g(c, z.shp, pk);
    }
  };
  bool valid[SW_DEPTH];
  int inflight = 0, done = 0;
#pragma unroll
  for (int s = 0; s < SW_DEPTH; ++s) {
    valid[s] = Tn < ntot;
    if (valid[s]) { issue(sl[s]); ++inflight; }
  }
  while (inflight > 0) {
#pragma unroll
    for (int s = 0; s < SW_DEPTH; ++s) {
      if (valid[s]) {                                // wave-uniform
        // younger than this tile's request: the requests of the tiles behind it, the stores of the tiles finished since
        wait_vm_le((inflight - 1) * L + min(done, SW_DEPTH - 1) * S_lb);
        finish(sl[s]);
        ++done;
        valid[s] = Tn < ntot;
        if (valid[s]) issue(sl[s]); else --inflight;
      }
    }
  }
  tr(c, 45);
  // ---- the vector elements of this thread (their loads were the first of the sweep) ----
#pragma unroll
  for (int i = 0; i < SW_NV; ++i) {
    if (vidx[i] >= 0) {
      float g = vg[i][0];
#pragma unroll
      for (int q = 1; q < KS; ++q) g += vg[i][q];
      if (do_grads) asg(J->grads)[vidx[i]] = g;
      if (do_adam) {
        float pp = vp[i], mm = vm[i], v2 = vv[i];
        adam1(ak, g, pp, mm, v2);
        st4_wt(Pp + vidx[i], pp);                  // (alpha is read by every part; the others only through `copy`)
        Mp[vidx[i]] = mm; Vp[vidx[i]] = v2;
        if (vcopy[i]) st4_g(c, vcopy[i], pp);
      }
    }
  }
}

// A helper's side of hand-off C: it has nothing to publish, so it only waits for the counter (bounded, as split_handoff),
// then acquires.  Returns false on a time-out / when another workgroup of the job has given up.
__device__ __forceinline__ bool wait_counter(const Ctx& c, GAS unsigned* cnt, GAS unsigned* err, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (c.tid == 0) {
    int spins = 0;
    bool ok = true;
    while (__hip_atomic_load((unsigned*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1 << 22) || __hip_atomic_load((unsigned*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store((unsigned*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *c.abort = ok ? 0u : 1u;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  return *c.abort == 0u;
}

// ---- kernel ----------------------------------------------------------------------------------------------------------
// Grid: groups of KS workgroups = the row slices of one (job, modality), every group on ONE XCD (workgroups b and b + 8
// share an XCD -- observed placement, speed only: the partials of a group then meet in that XCD's L2; correctness comes from
// the hand-off protocol): workgroup b = ((slot * KH + q) << 3) + xcd is member q of group slot * 8 + xcd = job * M + m; members
// q < KS are the row slices, members KS <= q < KH = KS + H are HELPERS: a small set leaves most CUs idle, and the Adam sweep --
// a quarter of a slice's step, bound by what ONE CU pulls from memory -- needs nothing but the partials in memory, so the
// idle CUs take a share of its tiles.  A helper waits for hand-off C (it publishes nothing), sweeps, arrives at D.
template <int KS>
__global__ __launch_bounds__(WG) void nm_rs_kernel(const nm_job_t* __restrict__ jobs, int step0, int n_steps, int flags,
                                                   int n_jobs, int M, int spread_us, int H) {
  constexpr int RTV = 8 / KS;
  NM_GEOM(RTV);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int KH = KS + H;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int q = idx % KH, group = (idx / KH) * 8 + xcd;
  const bool helper = q >= KS;
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
  c.rsk = KS; c.rsq = q; c.rloc0 = q * ROWS; c.xwg = 1; c.gwt = 1;
  c.ws0 = (GAS char*)J->workspace;
  c.ws = c.ws0 + (int64_t)(helper ? 0 : q) * J->workspace_stride;          // (a helper has no workspace tile of its own)
  c.gp_stride = J->gpart_stride;
  c.gpart = (GAS float*)J->gpart + (int64_t)(helper ? 0 : q) * J->gpart_stride;
  for (int i = c.tid; i < SMEM_BYTES / 4; i += WG) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  __syncthreads();
  const WsLayout wl = ws_layout(J->M, J->L, J->Z);
  GAS char* const gtab = c.ws + wl.rs_tab + (int64_t)part * WS_RS_TAB_BYTES;    // (helpers: slice 0's tables, complete at D below)
  if (!helper) rs_build_tables(c, part, gtab);
#ifdef NM_RS_DEBUG_TABLES
  {   // diagnostic build: dump the table header of workgroup (job 0, part 0, slice 0) into the loss log and leave
    if (blockIdx.x == 0 && c.tid == 0 && J->loss_log) {
      const GAS int* t = (const GAS int*)gtab;
      gf32 row = asg(J->loss_log);
      for (int i = 0; i < 16; ++i) row[i] = (float)t[i];
      const GAS int* v = (const GAS int*)(gtab + 544 + NM_RS_MAX_PASSES * 32);
      for (int i = 0; i < 16; ++i) row[16 + i] = (float)v[i];
    }
    return;
  }
#endif
  GAS unsigned* const sync0 = (GAS unsigned*)(c.ws0 + wl.sync);
  GAS unsigned* const sync_c = sync0 + WS_SYNC_C_WORD;
  GAS unsigned* const sync_d = sync0 + WS_SYNC_D_WORD + part;
  GAS unsigned* const sync_err = sync0 + WS_SYNC_ERR_WORD;
  // Where do the KS workgroups of this group run?  Each publishes its XCD (XCC_ID, bits 3:0 of hardware register 20), the
  // group meets once (arrival 1 of the D counter), and only if all KS ids agree -- they share one L2, which is then their
  // coherence point -- the group's internal payload (gradient partials, shadow images) is stored plain and served from
  // that L2 instead of being written through to memory and fetched back at the cross-XCD rate.  A placement that differs
  // from the expected one (workgroups b and b + 8 on one XCD) costs speed, never correctness.
  // (one word per modality part in tile 0: every member ORs in the bit of its XCD; one bit set = one XCD)
  {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;
    GAS unsigned* const word = sync0 + WS_SYNC_XCC_WORD + part;
    if (c.tid == 0) __hip_atomic_fetch_or((unsigned*)word, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!split_handoff(c, sync_d, sync_err, (unsigned)KH)) return;
    const unsigned seen = __hip_atomic_load((unsigned*)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool same = seen == (1u << xcc);
    c.gwt = (same && !(flags & NM_F_PROFILE)) ? 0 : 1;        // (NM_F_PROFILE here: force the write-through path, for A/B runs)
  }
  // start offsets: the models of a full chip otherwise reach their Adam sweeps -- the step's burst of memory traffic --
  // together; job j starts j / n_jobs of spread_us late (constant-rate counter, as nm_step_kernel's dephase)
  if (spread_us > 0 && n_jobs > 1) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long ticks = 100ull * (unsigned long long)min(spread_us, 20000) * (unsigned long long)job_idx / (unsigned long long)n_jobs;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  }
  const int nb = (J->n_rows + TROWS - 1) / TROWS;
  float ss_lane = 0.f, ib_lane = 0.f;
  for (int s = step0; s < step0 + n_steps; ++s) {
    const int b = s % nb;
    c.lstep = s - step0;
    const int brows = min(TROWS, J->n_rows - b * TROWS);       // valid rows of the batch
    c.row0 = b * TROWS + c.rloc0;
    c.nrows = max(0, min(ROWS, brows - c.rloc0));               // ... of this slice (0: a slice past a ragged batch's end)
    c.inv_b = 1.0f / (float)brows;                              // means run over the whole batch
    // Adam's step constants (double precision, as torch forms them): two pow() calls are ~2 us of dependent arithmetic --
    // 2 % of a 100-us step -- so every 64 steps lane l of each wave computes those of step s + l, and a step reads its lane
    if ((c.lstep & 63) == 0) {
      const int64_t t_opt = J->adam_off + (int64_t)s + 1 + c.lane;
      const double tt = (double)t_opt;
      const double lr_t = (J->lr_table && J->lr_cap > 0) ? J->lr_table[(t_opt - 1) % J->lr_cap] : (double)J->lr;
      ss_lane = (float)(lr_t / (1.0 - pow((double)J->beta1, tt)));
      ib_lane = (float)(1.0 / sqrt(1.0 - pow((double)J->beta2, tt)));
    }
    c.step_size = __shfl(ss_lane, c.lstep & 63, 64);
    c.inv_bc2_sqrt = __shfl(ib_lane, c.lstep & 63, 64);
    if (flags & 64) c.tlast[c.wave_s] = clock64();
    lds_barrier();
    relaunder(c);
    u32x4 tpre = {0u, 0u, 0u, 0u};
    if (helper) {
      if (c.tid < (int)sizeof(SwTab) / 16) tpre = *(const GAS u32x4*)(gtab + c.tid * 16);
      if (!wait_counter(c, sync_c, sync_err, (unsigned)(c.lstep + 1) * (unsigned)(M * KS))) break;
    } else {
      run_step<false, 0, RTV>(c, s);
      if (*c.abort != 0u) break;                                // a hand-off timed out (wave-uniform: LDS word read by all)
      if (c.tid < (int)sizeof(SwTab) / 16) tpre = *(const GAS u32x4*)(gtab + c.tid * 16);
      tr(c, 40);
      // C: every workgroup of the model has stored its partials and loss shares
      if (!split_handoff(c, sync_c, sync_err, (unsigned)(c.lstep + 1) * (unsigned)(M * KS))) break;
      // the next step's first x chunks travel while the sweep runs (P and Q are dead until then)
      c.x_pre = 0;
      if (s + 1 < step0 + n_steps) {
        const nm_modality_t& md = J->mod[part];
        const int bn = (s + 1) % nb;
        c.x_pre = fwd_first_layer_prefetch<RTV>(c, (const GAS char*)asg(md.xb) + (int64_t)bn * ((md.Kx + XCH - 1) / XCH) * XIMG_TILE_BYTES +
                                                       (int64_t)c.rloc0 * (LDX * 2), md.Kx);
      }
    }
    tr(c, 41);
    // loss row: the slices' shares (word 0: KL, word 1 + m: log-likelihood of modality m), summed in slice order by the
    // model's first workgroup.  Lane qq * (1 + M) + w of its wave 0 requests share w of slice qq now and the row is formed
    // after the sweep (the next step rewrites the shares only after hand-off D): the round trip is off the step's path.
    const bool loss_wg = part == 0 && q == 0 && c.wave_s == 0 && J->loss_log != nullptr;
    float share = 0.f;
    if (loss_wg && c.lane < KS * (1 + M)) {
      const int qq = c.lane / (1 + M), w = c.lane - qq * (1 + M);
      share = ((const GAS float*)(c.ws0 + (int64_t)qq * J->workspace_stride + wl.sync))[WS_SYNC_LOSS_WORD + w];
    }
    rs_sweep<KS>(c, part, tpre, KH);
    if (loss_wg) {                                              // (wave-uniform)
      gf32 row = asg(J->loss_log) + (int64_t)(s % J->loss_cap) * NM_LOSS_STRIDE;
      float kl = 0.f, ll_sum = 0.f;
      for (int qq = 0; qq < KS; ++qq) kl += __shfl(share, qq * (1 + M), 64);
      for (int m = 0; m < M; ++m) {
        float ll = 0.f;
        for (int qq = 0; qq < KS; ++qq) ll += __shfl(share, qq * (1 + M) + 1 + m, 64);
        if (c.lane == 0) row[NM_LOSS_LL_M + m] = ll;
        ll_sum += ll;
      }
      if (c.lane == 0) {
        row[NM_LOSS_KL] = J->kl_weight * kl;
        row[NM_LOSS_LL] = ll_sum;
        row[NM_LOSS_TC] = 0.f;
        row[15] = (float)c.gwt;                                 // (diagnostic: 0 = the group's payload stayed in one XCD's L2)
        row[NM_LOSS_TOTAL] = J->kl_weight * kl - J->ll_weight * ll_sum;
      }
    }
    tr(c, 42);
    // D: the modality's new shadow images / vector pieces are complete (the next forward of every slice reads them)
    if (!split_handoff(c, sync_d, sync_err, (unsigned)(c.lstep + 2) * (unsigned)KH)) break;    // (arrival 1: the placement check)
    tr(c, 43);
  }
  if ((flags & 64) && blockIdx.x < 512 && c.tid == 0) nm_wg_times[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace

extern "C" {

/* Row-split launch (include/nmhip.h): n_jobs models of M modalities each, k in {2, 4} row slices per (model, modality).
 * Every job needs k workspace tiles and gpart / gpart_stride; -16: the launch would not be resident at once;
 * -20: a job of the launch cannot run row-split (see nm_rowsplit_ok). */
int nm_launch_rowsplit(const nm_job_t* jobs_dev, int n_jobs, int M, int k, int helpers, int step0, int n_steps, int flags,
                       int spread_us, void* stream) {
  if (!jobs_dev) return -1;
  if (n_jobs < 1 || n_steps < 1 || step0 < 0 || M < 1 || M > NM_MAX_EXP || (k != 2 && k != 4)) return -8;
  if (!(flags & NM_F_BACKWARD) || !(flags & (NM_F_ADAM | NM_F_GRADS))) return -8;
  if ((flags & NM_F_GRADS) && n_steps != 1) return -8;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return -8;
  if (helpers < 0 || helpers > 60) return -8;
  const int groups = (n_jobs * M + 7) / 8 * 8;
  const int wgs = groups * (k + helpers);
  if (wgs > cus) return -16;            // the workgroups of a model wait for each other: all must be resident
  hipStream_t st = (hipStream_t)stream;
  nm_sync_reset(jobs_dev, n_jobs, stream);
  flags &= (NM_F_BACKWARD | NM_F_ADAM | NM_F_GRADS | NM_F_EXPORT | NM_F_TRACE | NM_F_FAULT_INJECT | NM_F_PROFILE);
  if (spread_us < 0 || n_steps < 16) spread_us = 0;          // (an offset is pure cost at the end of a short launch)
  hipError_t e;
  if (k == 2) {
    e = hipFuncSetAttribute((const void*)nm_rs_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nm_rs_kernel<2>, dim3(wgs), dim3(WG), SMEM_BYTES, st, jobs_dev, step0, n_steps, flags, n_jobs, M, spread_us, helpers);
  } else {
    e = hipFuncSetAttribute((const void*)nm_rs_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nm_rs_kernel<4>, dim3(wgs), dim3(WG), SMEM_BYTES, st, jobs_dev, step0, n_steps, flags, n_jobs, M, spread_us, helpers);
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
