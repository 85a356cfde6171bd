// One pass of the network over a batch: workspace, the plan of a pass (PassPlan, device-free) and its launches (Pass).
// Part of api.hip (included there after gru_model.hpp).
#pragma once
// ------------------------------------------------------------------------------------------
// forward
// Default column budget of one pass: 16 Mi columns = 51.5 GB gi + 2 x 17.2 GB activations, sized for
// 288 GB of HBM (option "max_rows_per_pass" overrides it; tests use a tiny value).
static const size_t kMaxRowsPerPass = (size_t)16 << 20;

// gi (3072 B per column and direction: 6.1 GB at 200 x 10000) only for the passes that touch it: in the throughput regime
// layer 1's pre-activations live in registers and layer 0's in the packed-x fragments, and the only other reader is the
// exact-projection fallback for input beyond fp16 range -- which such a pass then leaves to its caller (PassPlan::need_gi).
static int ensure_workspace(mdk_gru *m, size_t rows, bool need_gi) {
    const size_t D = m->D;
    if (rows > m->ws_rows) {
        free_dev(m->act[0]); free_dev(m->act[1]); free_dev(m->lpart);
        m->act[0] = m->act[1] = m->lpart = nullptr;
        m->ws_rows = 0;
        HIP_TRY(hipMalloc((void **)&m->act[0], rows * D * kH * sizeof(float)));
        if (m->desc.num_layers > 1) HIP_TRY(hipMalloc((void **)&m->act[1], rows * D * kH * sizeof(float)));
        if (m->desc.num_layers > 1) HIP_TRY(hipMalloc((void **)&m->lpart, rows * D * 5 * sizeof(float)));
        m->ws_rows = rows;
    }
    if (need_gi && rows > m->gi_rows) {
        free_dev(m->gi); m->gi = nullptr; m->gi_rows = 0;
        HIP_TRY(hipMalloc((void **)&m->gi, D * rows * kG * sizeof(float)));
        m->gi_rows = rows;
    }
    return MDK_OK;
}

struct EvTimer {
    struct Span { int slot; size_t e0, e1; hipStream_t st; };
    mdk_gru *m;
    hipStream_t s;
    size_t next = 0;
    std::vector<Span> spans;
    // begin a span on stream `on` (default: the forward's stream); returns its index through *idx
    int begin(int slot, hipStream_t on = (hipStream_t)-1, size_t *idx = nullptr) {
        if (!m->timing) return MDK_OK;
        hipStream_t st = (on != (hipStream_t)-1) ? on : s;
        while (m->ev.size() < next + 2) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            m->ev.push_back(e);
        }
        HIP_TRY(hipEventRecord(m->ev[next], st));
        if (idx) *idx = spans.size();
        spans.push_back({slot, next, next + 1, st});
        next += 2;
        return MDK_OK;
    }
    int end() { return spans.empty() ? MDK_OK : end_at(spans.size() - 1); }
    int end_at(size_t idx) {
        if (!m->timing) return MDK_OK;
        HIP_TRY(hipEventRecord(m->ev[spans[idx].e1], spans[idx].st));
        return MDK_OK;
    }
};

enum { SLOT_GI0 = 0, SLOT_REC0 = 4, SLOT_HEAD = 8 };

// Host path of mdk_gru_forward (reference TorchModel.predict_on_batch, models.py:303-313: host tensor in,
// host tensor out): x arrives and the probabilities leave in TIME SLABS while the recurrences run.
//   in : scan step s of a bidirectional layer needs column s (forward) and T-1-s (reverse), so the slabs
//        come from both ends towards the middle -- [0,T/16)+[15T/16,T) first, doubling -- as strided 2-D
//        copies (one row of slab columns per window) into the natural (B,T,F) device layout; layer 0's
//        recurrence is cut at the same boundaries and each piece waits only for its own slabs;
//   out: finished columns are copied out as soon as they exist (again 2-D: nt columns x nb windows) -- behind the chunks
//        of a side-stream classifier head where the recurrence leaves CUs idle for one (sequential scan of a small
//        batch), behind the launches of the last layer's second half where that half writes the probabilities itself
//        (rec_fused.hpp HEAD = 2: split scans, batches that fill the chip).
// 80 MB in + 40 MB out per 200 x 10000 batch cost 2.1 ms of PCIe time (profiles/r2_host_path_probe.txt);
// what stays exposed is the first slab pair (10 MB; a split call: all of x) and the last chunk of columns.
struct HostIO {
    const float *x_host = nullptr;   // (nb, T, F) of this pass, or null: x is already on the device
    float *p_host = nullptr;         // (nb, T, C) of this pass, or null: probabilities stay on the device
    float *p_host_dev = nullptr;     // the device's view of p_host where the caller's buffer is page-locked AND no other forward will run
                                     // behind this one (the cold host entry): the last result chunks then leave by kernel (k_tail_to_host)
};

// the streamed host path of a split call cuts the last layer's scan into launches (forward_pass): only for virtual windows
// long enough for that to be worth them
constexpr int kSplitStreamMinT = 512;

static int pool_event(mdk_gru *m, hipEvent_t *out) {
    if (m->ov_next == m->ov_ev.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m->ov_ev.push_back(e);
    }
    *out = m->ov_ev[m->ov_next++];
    return MDK_OK;
}

// ---- one pass of the network over nb windows of T columns --------------------------------------------------------------
// `sp` (split scan): x holds the REAL batch, the pass runs the virtual one (nb = sp->S * sp->B windows of T = sp->Tv
// columns) and `probs` is the REAL (sp->B, sp->T, C) result, filled by the head with every chunk's own columns.
//
// PassPlan: every decision about the pass, and no device work -- also what ensure_workspace() asks how much it needs.
constexpr int kOvMaxWgs = 208;   // profiles/run_overlap_sweep.sh: +10 % at 128 work-groups, +5 % at 160, +-1 % at 200-256
constexpr int kOvChunks = 6;     // (a finer, shrinking schedule measured no better: the GEMM is the longer leg)

struct PassPlan {
    int nb = 0, T = 0, D = 2, L = 1, n_tiles = 0, nq = 1, n_wg = 0;
    bool exact = false, hp = false;
    bool io_in = false, io_out = false, sp_out = false;
    bool can_chunk = false, can_chunk_sp = false;
    bool fuse0 = false;        // layer 0: K <= 16 projection inside the recurrence (k_pack_x operands; device-side fallback on range)
    bool fuse_proj = false;    // layers >= 1: projection inside the recurrence (rec_fused.hpp): the throughput regime
    bool fuse_head = false;    // ... and the classifier's Linear
    bool final_head = false;   // ... and the softmax: the scan's second half delivers probabilities (HEAD = 2)
    bool overlap = false;      // latency regime: layer 1's GEMM / the head on a side stream under the recurrences' tails
    bool stream_in = false, stream_out = false;
    int abl = 0;               // debug builds: timing-only ablation mask
    bool ablated = false;
    // Who needs gi in HBM: the exact kernels, an unfused layer 0, unfused layers >= 1 -- and the exact-projection FALLBACK of
    // a fused layer 0 (input beyond fp16 range), enqueued behind it as empty launches that the range flag arms on the
    // device.  A caller that synchronises anyway and promises to look at the flag itself (`host_checks_range`: split calls,
    // the host entries) lets the throughput regime run WITHOUT gi and without those launches; if the flag is up it marks
    // the model (`oor_seen`) and repeats the call -- from then on with gi and the device-side decision.
    bool need_gi = true;
};

static int plan_pass(const mdk_gru *m, int nb, int T, const HostIO *io, const SplitPlan *sp, PassPlan &P, bool host_checks_range = false,
                     bool lean = false) {
    P = PassPlan{};
    P.nb = nb; P.T = T; P.D = m->D; P.L = m->desc.num_layers;
    P.exact = (m->variant == MDK_VARIANT_EXACT);
    P.n_tiles = (nb + kTileWin - 1) / kTileWin;
    P.io_in = io && io->x_host;
    P.io_out = io && io->p_host && !sp;
    P.sp_out = sp && io && io->p_host;          // split call: `probs` is the real (B, T, C) result, io->p_host the caller's buffer
    if (P.exact) return MDK_OK;
    if (m->layers[0].K > 16)
        return fail(MDK_ERR_ARG, "num_features %d > 16 is only supported by MDK_VARIANT_EXACT", m->layers[0].K);
#ifdef MDK_DEBUG_HOOKS
    // "ablate" option / MDK_ABLATE=<mask>: timing-only ablations (wrong results)
    static const int env_abl = getenv("MDK_ABLATE") ? atoi(getenv("MDK_ABLATE")) : 0;
    P.abl = m->opt_ablate ? m->opt_ablate : env_abl;
#endif
    const int D = P.D, L = P.L;
    // work-group granularity of the recurrence: 4 windows while that fits the chip in one round of
    // work-groups (latency-bound regime), else 8, else (half-precision mode only) 16
    P.hp = (m->precision == MDK_PREC_FP16);
    const int n_win = P.n_tiles * kTileWin;
    int nq = 1;
    // (profiles/run_tile_sweep.sh: at 256 work-groups of 4 windows the 8-window variant + overlap is
    // already 3 % ahead, at 200 it is 3 % behind)
    // one work-group owns a CU (8 waves x <= 256 VGPRs), so a call wants all its work-groups resident at once; with
    // `gpu_share` processes on the GPU each takes its share of the 256 CUs (work-groups of different processes
    // do run side by side, profiles/r3_procs_per_gpu.txt), otherwise the surplus queues behind the others
    const int cu_budget = 232 / m->opt_gpu_share;
    while (nq < (P.hp ? 4 : 2) && ((n_win + 4 * nq - 1) / (4 * nq)) * D > cu_budget) nq *= 2;
    // half precision: 8-window work-groups while they fit the chip, so that layers >= 1 can run fused (rec_fused.hpp carries
    // 8 windows; 16-window groups fill only half the CUs at 1000 chunk-windows)
    if (P.hp && nq == 4 && m->opt_fuse_proj && L >= 2 && ((n_win + 7) / 8) * D * m->opt_gpu_share <= 256) nq = 2;
    // `lean` (the audit's sequential scan): whatever the batch, the regime that needs no gi in HBM -- 8-window work-groups with
    // the projection inside the recurrence -- so that an audit allocates nothing (and, above all, FREES nothing: see run_forward)
    if (lean && nq < 2 && m->opt_fuse_proj && L >= 2) nq = 2;
    if (m->opt_tile_windows == 4) nq = 1;
    if (m->opt_tile_windows == 8) nq = 2;
    if (m->opt_tile_windows == 16 && P.hp) nq = 4;
    P.nq = nq;
    P.n_wg = (n_win + 4 * nq - 1) / (4 * nq);
    // Overlap plan (bidirectional, >= 2 layers): gi of layer 1 at column t needs layer 0's forward h_t
    // (ready after scan step t) and backward h_t (ready after scan step T-1-t), i.e. columns
    // [T-s, s) after s steps.  The second half of layer 0's recurrence is cut into chunks; after each,
    // the newly complete column ranges are projected on a side stream by the CUs the latency-bound
    // recurrence leaves idle.  Needs its own gi buffer: layer 0's unfused fallback may still read gi.
    // Measured at B=200: 14.8 -> 13.5 ms per batch; the recurrence itself gets 9 % slower while the
    // GEMM runs (chip clock drops with the extra power draw -- padding its LDS so that no GEMM
    // work-group can share its CUs changed nothing), at B >= 1000 there are no idle CUs and no gain.
    P.ablated = (P.abl != 0 && !P.hp && nq <= 2);
    const bool can_chunk_any = D == 2 && !P.ablated && T % (2 * kGemmSteps) == 0;
    P.can_chunk = can_chunk_any && T >= 2048;
    P.can_chunk_sp = can_chunk_any && T >= kSplitStreamMinT;      // a split call's result can leave in column chunks
    const bool overlap_ok = m->opt_overlap && P.can_chunk && L >= 2 &&
                            (P.n_wg * D * m->opt_gpu_share <= kOvMaxWgs || m->opt_overlap == 2);   // only while the recurrence leaves CUs idle (2 = force)
    // Layers >= 1 in the throughput regime (every CU holds a recurrence work-group: nothing is idle to hide a projection
    // GEMM under): the projection runs INSIDE the recurrence kernel, strip by strip, and gi never exists in HBM
    // (rec_fused.hpp; bit-identical to the GEMM + recurrence pair).  fp32-parity or half mode, 8-window work-groups, T a
    // multiple of the strip.  "fuse_proj" = 2 prefers it to the side-stream GEMM as well.
    // (rec_fused.hpp addresses a tile's activations through a buffer resource: 32-bit byte offsets t * D * 4096 inside a 2 GB
    // window -- beyond T * D * 4096 = 2^31 the offsets would wrap, loads return 0 and stores are dropped: such a scan takes the
    // GEMM + k_rec_mfma pair, whose addresses are 64-bit)
    const bool fused_addressable = (long long)T * D * 4096 < (1LL << 31);
    P.fuse_proj = L >= 2 && nq == 2 && !P.ablated && T % kFusedSteps == 0 && fused_addressable &&
                  (m->opt_fuse_proj == 2 || (m->opt_fuse_proj == 1 && (lean || P.n_wg * D * m->opt_gpu_share > kOvMaxWgs)));   // auto: the recurrence fills the chip
    P.overlap = overlap_ok && !P.fuse_proj;
    // ... and with it the classifier's Linear (rec_fused.hpp HEAD): the last layer leaves partial logits, k_head_combine
    // finishes them (fp16x2-split MFMA instead of fp32 FMAs: ~1e-7 relative on the logits, not bit for bit)
    P.fuse_head = P.fuse_proj && m->opt_fuse_head && m->desc.num_classes == 5;
    P.fuse0 = m->opt_fuse_l0 && m->layers[0].wx_frag != nullptr && !P.ablated;
    P.stream_in = P.io_in && P.can_chunk && P.fuse0 && m->opt_stream_host;    // x in time slabs under layer 0's (fused) recurrence
    // ... and where the scan's second half can deliver the probabilities itself (rec_fused.hpp HEAD = 2: a step's column is
    // complete once the other direction has passed it): no head kernel, and finished columns can go home by DMA under the
    // rest of the scan.  Bidirectional: the scan is cut at T/2, a multiple of the strip.  "final_head" = 0: k_head_combine.
    P.final_head = P.fuse_head && m->opt_final_head && (D == 1 || T % (2 * kFusedSteps) == 0);
    // the result leaves in column chunks under the last recurrence: behind a side-stream head where the recurrence leaves
    // CUs idle for one (sequential scan of a small batch), behind the launches of a final-head scan (DMA only; a head
    // KERNEL beside a recurrence that holds every CU crawls: a split call without the final head leaves as one copy)
    P.stream_out = ((P.io_out && P.can_chunk) || (P.sp_out && P.can_chunk_sp && P.final_head)) && L >= 2 && m->opt_stream_host;
    P.need_gi = !P.fuse0 || (L >= 2 && !P.fuse_proj) || !host_checks_range || m->oor_seen;
    return MDK_OK;
}

// Pass: the launches.  One object per pass; the methods are the regimes of DESIGN.md section 4.
struct Pass {
    mdk_gru *m;
    const PassPlan &P;
    const float *x;            // device input of the pass (natural (nb, T, F); a split call: the real batch)
    float *probs;
    hipStream_t s;
    EvTimer &tm;
    const HostIO *io;
    const SplitPlan *sp;
    std::vector<hipEvent_t> *join_later;     // split call: the events behind its last result copies (run_split waits for them
                                             // after its certificate kernel) instead of a wait on `s`
    struct OutRange { hipEvent_t ready; int t0, nt; int launch; };
    std::vector<OutRange> out_ranges;        // column ranges to copy out; issued after every launch is enqueued, because a
                                             // copy into pageable memory may block the calling thread until it is done
    const float *in = nullptr;               // input of the current layer
    const float *gi_l1 = nullptr;            // where layer 1 finds its gi
    bool gemm_done = false, head_done = false;
    int reverse_mask() const { return P.D == 2 ? 2 : 0; }
    dim3 rgrid() const { return dim3(P.n_wg, P.D); }

    int run();
    int run_exact();
    int layer(int l);
    int layer_final_head(int l, const LayerDev &Ld, float *outp);
    int layer_phased(int l, const LayerDev &Ld, float *outp, bool fuse, bool slabs, bool dev_slabs, bool side_gemm, bool side_head);
    int copy_out();
    // launches
    void launch_gemm(const LayerDev &Lg, const float *src, float *gi_out, hipStream_t st, int strip0, int n_strips,
                     const int *gcond = nullptr, int gwant = 0);
    void launch_head(const float *src, hipStream_t st, int t0, int nt);
    void pack_cols(const LayerDev &Lp, const float *src, int t0, int nt, hipStream_t st);
    int copy_in_cols(int t0, int nt);
    void launch_gi_small(int l, const LayerDev &Ld, const int *cond);
    void launch_rec(int l, const LayerDev &Ld, const float *gi_src, float *outp, bool xin, const int *cnd, int want, int rs0, int rns, bool fin = false);
    void launch_rec_fallback(const LayerDev &Ld, const float *gi_src, float *outp, const int *cnd, int rs0, int rns);
#ifdef MDK_DEBUG_HOOKS
    int launch_rec_ablated(const LayerDev &Ld, const float *gi_src, float *outp, int rs0, int rns);
#endif
};

// projection GEMM of a layer over the 8-step strips [strip0, strip0 + n_strips): 64-row work-groups, two per CU
// (128-row work-groups -- half the L2 traffic for W_ih, one per CU -- are bit-identical and measured 7 % SLOWER at
// 1000 x 10000: with one work-group per CU nothing overlaps the staging; profiles/r3_experiments/README.md)
void Pass::launch_gemm(const LayerDev &Lg, const float *src, float *gi_out, hipStream_t st, int strip0, int n_strips,
                       const int *gcond, int gwant) {
    if (n_strips <= 0) return;
    const int T = P.T, D = P.D, n_tiles = P.n_tiles;
    const int t_end = std::min(T, (strip0 + n_strips) * kGemmSteps);
    const dim3 grid((unsigned)n_strips * n_tiles);
#define MDK_GEMM(KS, HPF)                                                                          \
    hipLaunchKernelGGL((k_gi_gemm<KS, HPF>), grid, dim3(512), (size_t)2 * kGemmMT * KS * 64 * sizeof(half8), st, \
                       src, Lg.wih_frag, Lg.bias_gi, gi_out, n_tiles, T, D, Lg.inv_scale_gi, Lg.up_scale_rec, kActScale, strip0, \
                       gcond, gwant, t_end)
    if (D == 2) { if (P.hp) MDK_GEMM(8, true); else MDK_GEMM(8, false); }
    else { if (P.hp) MDK_GEMM(4, true); else MDK_GEMM(4, false); }
#undef MDK_GEMM
}

// classifier head over the columns [t0, t0 + nt) of every window
void Pass::launch_head(const float *src, hipStream_t st, int t0, int nt) {
    if (nt <= 0) return;
    const int T = P.T, D = P.D, nb = P.nb, n_tiles = P.n_tiles;
    if (P.fuse_head) {
        const long n = (long)n_tiles * nt * kTileWin;
        const unsigned blocks = (unsigned)std::min<long>((n + 255) / 256, 256 * 8);
        if (sp) hipLaunchKernelGGL(k_head_combine<true>, dim3(blocks), dim3(256), 0, st, (const float *)m->lpart, m->lin_b, probs, nb, T,
                                   n_tiles, D, m->desc.normalise, t0, nt, *sp);
        else hipLaunchKernelGGL(k_head_combine<false>, dim3(blocks), dim3(256), 0, st, (const float *)m->lpart, m->lin_b, probs, nb, T,
                                n_tiles, D, m->desc.normalise, t0, nt, SplitPlan{});
        return;
    }
    const long n_blocks = (long)n_tiles * nt;
    const long blocks = std::min<long>((n_blocks + 3) / 4, 256 * 8);
    if (sp)       // (plan_split: bidirectional models only)
        hipLaunchKernelGGL((k_head_tiled<2, true>), dim3((unsigned)blocks), dim3(256), 0, st, src, m->lin_w, m->lin_b,
                           probs, nb, T, n_tiles, m->desc.normalise, t0, nt, *sp);
    else if (D == 2)
        hipLaunchKernelGGL(k_head_tiled<2>, dim3((unsigned)blocks), dim3(256), 0, st, src, m->lin_w, m->lin_b,
                           probs, nb, T, n_tiles, m->desc.normalise, t0, nt, SplitPlan{});
    else
        hipLaunchKernelGGL(k_head_tiled<1>, dim3((unsigned)blocks), dim3(256), 0, st, src, m->lin_w, m->lin_b,
                           probs, nb, T, n_tiles, m->desc.normalise, t0, nt, SplitPlan{});
}

void Pass::pack_cols(const LayerDev &Lp, const float *src, int t0, int nt, hipStream_t st) {
    if (nt <= 0) return;
    const size_t need = (size_t)P.n_wg * nt * kXfragLanes;
    hipLaunchKernelGGL(k_pack_x, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, st, src, m->xfrag, P.nb, P.T,
                       Lp.K, P.nq, P.hp ? 1 : 0, P.n_wg, Lp.x_scale, m->oor_flag, t0, nt, sp ? *sp : SplitPlan{});
}

// host -> device copy of the columns [t0, t0 + nt) of every window of this pass
int Pass::copy_in_cols(int t0, int nt) {
    if (nt <= 0) return MDK_OK;
    const int F = m->desc.num_features, T = P.T;
    HIP_TRY(hipMemcpy2DAsync(const_cast<float *>(x) + (size_t)t0 * F, (size_t)T * F * sizeof(float),
                             io->x_host + (size_t)t0 * F, (size_t)T * F * sizeof(float),
                             (size_t)nt * F * sizeof(float), (size_t)P.nb, hipMemcpyHostToDevice, m->copy_in));
    return MDK_OK;
}

// unfused layer-0 projection: the only path without fusion, the on-device fallback (input beyond
// fp16 range) with it.  It reads all of x, so with slabs it is enqueued after the last of them.
void Pass::launch_gi_small(int l, const LayerDev &Ld, const int *cond) {
    const int tpb = 128, T = P.T;
    const float *src = in;
    if (sp && l == 0) {
        // split scan: `in` is the REAL batch (k_pack_x maps the virtual windows onto it); the exact projection wants
        // the virtual batch in memory -- gathered only if the range flag is up (unfused layer 0: always)
        const int F = m->desc.num_features;
        const int vec = (F % 2 == 0 && reinterpret_cast<uintptr_t>(in) % 8 == 0) ? 2 : 1;
        const size_t n = (size_t)P.nb * T * F / vec;
        hipLaunchKernelGGL(k_split_gather, dim3((unsigned)std::min<size_t>((n + 255) / 256, 256 * 16)), dim3(256), 0, s,
                           in, m->xv, *sp, F, vec, 0, T, cond);
        src = m->xv;
    }
    hipLaunchKernelGGL(k_gi_small<16>, dim3(P.n_tiles, P.D, (T + tpb - 1) / tpb), dim3(768), 0, s, src,
                       Ld.w_ih_t, Ld.bias_gi, m->gi, P.nb, T, Ld.K, P.n_tiles, tpb, Ld.up_scale_rec, cond, 1);
}

#define MDK_LAUNCH_REC_T(NQV, XIN, HPF, A, DSV, CND, WANT)                                         \
    hipLaunchKernelGGL((k_rec_mfma<MDK_PF, NQV, XIN, HPF, 0, A, DSV>), rgrid(), dim3(512), 0, s, gi_src, m->xfrag, \
                       Ld.wx_frag, Ld.whh_frag, Ld.b_hn, outp, P.n_tiles, P.T, P.D, Ld.inv_scale_rec,    \
                       reverse_mask(), CND, WANT, rs0, rns)
// deferred HBM store of h_t (default) or the store behind the gate math; ablation builds use the latter
#define MDK_LAUNCH_REC(NQV, XIN, HPF, A, CND, WANT)                                                \
    do { if ((A) == 0 && m->opt_deferred_store) MDK_LAUNCH_REC_T(NQV, XIN, HPF, 0, true, CND, WANT); \
         else MDK_LAUNCH_REC_T(NQV, XIN, HPF, A, false, CND, WANT); } while (0)

// one recurrence launch over the scan steps [rs0, rs0 + rns) of layer l.
// `fin`: this launch's columns are complete (second half of a bidirectional scan, any step of a one-directional
// one): the fused head writes probabilities instead of partial logits (rec_fused.hpp HEAD = 2)
void Pass::launch_rec(int l, const LayerDev &Ld, const float *gi_src, float *outp, bool xin, const int *cnd, int want, int rs0, int rns, bool fin) {
    const int nq = P.nq, D = P.D, L = P.L;
    if (l >= 1 && P.fuse_proj) {
        const int hd = (P.fuse_head && l == L - 1) ? (fin ? 2 : 1) : 0;
#define MDK_LAUNCH_FUSED(KS, HD, HPF)                                                                                         \
    hipLaunchKernelGGL((k_rec_fused<KS, HD, HPF>), rgrid(), dim3(512), fused_lds_bytes(KS, HPF), s, in, Ld.wih_frag, Ld.bias_gi, \
                       Ld.whh_frag, Ld.b_hn, outp, P.n_tiles, P.T, D, Ld.inv_scale_rec, Ld.inv_scale_gi, Ld.up_scale_rec,   \
                       kActScale, reverse_mask(), rs0, rns, (const half8 *)m->wlin_frag, m->lin_inv_scale, m->lpart,        \
                       (const float *)m->lin_b, probs, P.nb, (int)m->desc.normalise, sp ? *sp : SplitPlan{})
#define MDK_LAUNCH_FUSED_P(KS, HD) do { if (P.hp) MDK_LAUNCH_FUSED(KS, HD, true); else MDK_LAUNCH_FUSED(KS, HD, false); } while (0)
#define MDK_LAUNCH_FUSED_H(KS) do { if (hd == 2) MDK_LAUNCH_FUSED_P(KS, 2); else if (hd == 1) MDK_LAUNCH_FUSED_P(KS, 1); else MDK_LAUNCH_FUSED_P(KS, 0); } while (0)
        if (D == 2) MDK_LAUNCH_FUSED_H(8); else MDK_LAUNCH_FUSED_H(4);
#undef MDK_LAUNCH_FUSED_H
#undef MDK_LAUNCH_FUSED_P
#undef MDK_LAUNCH_FUSED
        if (hd) m->last.fused_layers |= 1 << 8;
        if (hd == 2) m->last.fused_layers |= 1 << 9;
        m->last.fused_layers |= 1 << l;
        return;
    }
    if (P.hp) {
        if (nq == 1) { if (xin) MDK_LAUNCH_REC(1, true, true, 0, cnd, want); else MDK_LAUNCH_REC(1, false, true, 0, cnd, want); }
        else if (nq == 2) { if (xin) MDK_LAUNCH_REC(2, true, true, 0, cnd, want); else MDK_LAUNCH_REC(2, false, true, 0, cnd, want); }
        else { if (xin) MDK_LAUNCH_REC(4, true, true, 0, cnd, want); else MDK_LAUNCH_REC(4, false, true, 0, cnd, want); }
    } else {
        if (nq == 1) { if (xin) MDK_LAUNCH_REC(1, true, false, 0, cnd, want); else MDK_LAUNCH_REC(1, false, false, 0, cnd, want); }
        else { if (xin) MDK_LAUNCH_REC(2, true, false, 0, cnd, want); else MDK_LAUNCH_REC(2, false, false, 0, cnd, want); }
    }
}

// the unfused twin of a fused layer 0: runs only if the range flag is up.  It is instantiated with a different ring depth
// only so that profilers show it under its own symbol (its launches are empty unless the range flag is raised)
void Pass::launch_rec_fallback(const LayerDev &Ld, const float *gi_src, float *outp, const int *cnd, int rs0, int rns) {
#define MDK_LAUNCH_FB(NQV, HPF)                                                                    \
    hipLaunchKernelGGL((k_rec_mfma<MDK_PF - 1, NQV, false, HPF>), rgrid(), dim3(512), 0, s, gi_src, m->xfrag, \
                       Ld.wx_frag, Ld.whh_frag, Ld.b_hn, outp, P.n_tiles, P.T, P.D, Ld.inv_scale_rec,    \
                       reverse_mask(), cnd, 1, rs0, rns)
    const int nq = P.nq;
    if (P.hp) { if (nq == 1) MDK_LAUNCH_FB(1, true); else if (nq == 2) MDK_LAUNCH_FB(2, true); else MDK_LAUNCH_FB(4, true); }
    else { if (nq == 1) MDK_LAUNCH_FB(1, false); else MDK_LAUNCH_FB(2, false); }
#undef MDK_LAUNCH_FB
}

#ifdef MDK_DEBUG_HOOKS
// timing-only ablations: fp32-parity mode, unfused input, 4- or 8-window work-groups
int Pass::launch_rec_ablated(const LayerDev &Ld, const float *gi_src, float *outp, int rs0, int rns) {
    const int abl = P.abl, nq = P.nq;
#define MDK_ABL_CASE(A)                                                                            \
    case A:                                                                                        \
        if (nq == 1) MDK_LAUNCH_REC(1, false, false, A, (A & 64) ? m->oor_flag : (const int *)nullptr, 0); \
        else MDK_LAUNCH_REC(2, false, false, A, (A & 64) ? m->oor_flag : (const int *)nullptr, 0);  \
        break;
    if (abl & 64) HIP_TRY(hipMemsetAsync(m->oor_flag, 0, sizeof(int), s));
    switch (abl) {
        MDK_ABL_CASE(1) MDK_ABL_CASE(2) MDK_ABL_CASE(4) MDK_ABL_CASE(8) MDK_ABL_CASE(16)
        MDK_ABL_CASE(7) MDK_ABL_CASE(31) MDK_ABL_CASE(64)
        default: return fail(MDK_ERR_ARG, "unsupported ablation mask %d", abl);
    }
#undef MDK_ABL_CASE
    return MDK_OK;
}
#endif
#undef MDK_LAUNCH_REC
#undef MDK_LAUNCH_REC_T

// natural [window][t][f] layouts, plain fp32 kernels
int Pass::run_exact() {
    const int D = P.D, L = P.L, nb = P.nb, T = P.T;
    const long M = (long)nb * T;
    int rc;
    const size_t x_bytes = (size_t)M * m->desc.num_features * sizeof(float);
    const size_t p_bytes = (size_t)M * m->desc.num_classes * sizeof(float);
    if (P.io_in) HIP_TRY(hipMemcpyAsync(const_cast<float *>(x), io->x_host, x_bytes, hipMemcpyHostToDevice, s));
    const size_t gi_dir_stride = (size_t)M * kG;
    const int out_stride = D * kH;
    in = x;
    for (int l = 0; l < L; ++l) {
        const LayerDev &Ld = m->layers[l];
        float *outp = m->act[l & 1];
        if ((rc = tm.begin(SLOT_GI0 + l))) return rc;
        hipLaunchKernelGGL(k_gi_exact, dim3((unsigned)(3 * M), D), dim3(128), 0, s, in, Ld.w_ih_t,
                           Ld.bias_gi, m->gi, M, Ld.K, gi_dir_stride, Ld.ones);
        if ((rc = tm.end())) return rc;
        if ((rc = tm.begin(SLOT_REC0 + l))) return rc;
        hipLaunchKernelGGL(k_rec_exact, dim3(nb, D), dim3(128), 0, s, m->gi, Ld.w_hh_t, Ld.b_hn, outp,
                           nb, T, out_stride, gi_dir_stride, reverse_mask());
        if ((rc = tm.end())) return rc;
        m->last.rec_launches++;
        in = outp;
    }
    if ((rc = tm.begin(SLOT_HEAD))) return rc;
    long blocks = std::min<long>((M + 15) / 16, 256 * 16);
    if (D == 2)
        hipLaunchKernelGGL(k_linear_softmax<4>, dim3((unsigned)blocks), dim3(256), 0, s, in, m->lin_w,
                           m->lin_b, probs, M, m->desc.normalise);
    else
        hipLaunchKernelGGL(k_linear_softmax<2>, dim3((unsigned)blocks), dim3(256), 0, s, in, m->lin_w,
                           m->lin_b, probs, M, m->desc.normalise);
    if ((rc = tm.end())) return rc;
    HIP_TRY(hipGetLastError());
    if (P.io_out) HIP_TRY(hipMemcpyAsync(io->p_host, probs, p_bytes, hipMemcpyDeviceToHost, s));
    return MDK_OK;
}

// Last layer, fused head (throughput regime): [0, T/2) leaves partial logits, the launches after T/2 (every launch of a
// one-directional scan) deliver probabilities; on the host path the second half is cut again so that what it has
// finished -- columns [T - s', T - s) + [s, s') after the launch [s, s') -- crosses PCIe under the next launch.
int Pass::layer_final_head(int l, const LayerDev &Ld, float *outp) {
    const int T = P.T, D = P.D;
    int rc;
    std::vector<int> ph{0};
    if (D == 2) {
        ph.push_back(T / 2);
        // (a launch's columns must have crossed PCIe before the next launch ends: ~0.9 us per column pair of a
        // 200-window batch + ~10 us per copy against 1.8 us per step -- halvings keep that.  A split scan's LAST launch
        // is its outer margin, [T - G, T): only the two edge chunks deliver anything from it -- the first and last G
        // columns of every window, two copies -- so all but those have left when the scan ends.)
        if (P.stream_out) {
            const int last_cut = sp ? T - sp->G : T;
            for (int k = 1; k <= (sp ? 3 : 4); ++k) {
                const int cut = T / 2 + ((T / 2) - ((T / 2) >> k)) / kFusedSteps * kFusedSteps;
                if (cut > ph.back() && cut < T && (!sp || cut + 64 < last_cut)) ph.push_back(cut);
            }
            if (sp && last_cut > ph.back() && last_cut % kFusedSteps == 0) ph.push_back(last_cut);
        }
    }
    ph.push_back(T);
    for (size_t p = 0; p + 1 < ph.size(); ++p) {
        const bool fin = D == 1 || p >= 1;
        launch_rec(l, Ld, m->gi, outp, false, nullptr, 0, ph[p], ph[p + 1] - ph[p], fin);
        m->last.rec_launches++;
        if (!(P.stream_out && fin && D == 2)) continue;
        hipEvent_t ev;
        if ((rc = pool_event(m, &ev))) return rc;
        HIP_TRY(hipEventRecord(ev, s));
        const int lo0 = T - ph[p + 1], hi0 = ph[p], len = ph[p + 1] - ph[p];
        if (lo0 + len == hi0) out_ranges.push_back({ev, lo0, 2 * len, (int)p});
        else { out_ranges.push_back({ev, lo0, len, (int)p}); out_ranges.push_back({ev, hi0, len, (int)p}); }
    }
    m->last.rec_launches--;   // (the caller counts the layer once)
    head_done = true;
    return MDK_OK;
}

// Latency regime (and the slab-wise start of any layer 0).  The scan is cut into phases [ph[p], ph[p+1]).  First half: one
// phase, or -- when x is still arriving -- four that double in length, each behind the copy of its two slabs.  Second half:
// one phase, or kOvChunks with, behind each on the side stream, what the newly complete columns
// [T-s', T-s) + [s, s') feed: layer 1's projection (l = 0) or the classifier head (last layer).
int Pass::layer_phased(int l, const LayerDev &Ld, float *outp, bool fuse, bool slabs, bool dev_slabs, bool side_gemm, bool side_head) {
    const int T = P.T;
    const int *cond = fuse ? m->oor_flag : nullptr;
    const float *gi_src = (l == 1 && gemm_done) ? gi_l1 : m->gi;
    int rc;
    std::vector<int> ph{0};
    if (slabs) for (int sh = (T >= 8192 ? 5 : 4); sh >= 2; --sh) ph.push_back((T >> sh) / kGemmSteps * kGemmSteps);
    ph.push_back(T / 2);
    const int n_first = (int)ph.size() - 1;
    if (side_gemm) {
        for (int j = 1; j < kOvChunks; ++j) ph.push_back(T / 2 + (int)((long)(T / 2) * j / kOvChunks) / kGemmSteps * kGemmSteps);
    } else if (side_head) {
        // halving chunks: what follows the last recurrence launch (its head chunk, and on the host path
        // the copy of that chunk) is T/32 columns instead of T/12
        for (int k = 1; k <= 4; ++k) ph.push_back(T / 2 + ((T / 2) - ((T / 2) >> k)) / kGemmSteps * kGemmSteps);
    }
    ph.push_back(T);
    const int n_ph = (int)ph.size() - 1;
    size_t gspan = 0;
    bool gspan_open = false;
    hipEvent_t slab_ev[8] = {};
    for (int p = 0; p < n_ph; ++p) {
        const int rs0 = ph[p], rns = ph[p + 1] - ph[p];
        if (slabs && p < n_first && dev_slabs) {
            if (p == 0) {
                pack_cols(Ld, in, 0, ph[1], s);
                pack_cols(Ld, in, T - ph[1], ph[1], s);
                hipEvent_t ev0;                        // x may come from earlier work on `s`
                if ((rc = pool_event(m, &ev0))) return rc;
                HIP_TRY(hipEventRecord(ev0, s));
                HIP_TRY(hipStreamWaitEvent(m->side, ev0, 0));
                for (int pp = 1; pp < n_first; ++pp) {
                    const int lo = ph[pp], len = ph[pp + 1] - ph[pp];
                    pack_cols(Ld, in, lo, len, m->side);
                    pack_cols(Ld, in, T - lo - len, len, m->side);
                    if ((rc = pool_event(m, &slab_ev[pp]))) return rc;
                    HIP_TRY(hipEventRecord(slab_ev[pp], m->side));
                }
            } else {
                HIP_TRY(hipStreamWaitEvent(s, slab_ev[p], 0));
            }
        } else if (slabs && p < n_first) {
            // columns [rs0, rs0+rns) and their mirror [T-rs0-rns, T-rs0); the last pair is adjacent
            const int lo = rs0, hi = T - rs0 - rns;
            if (lo + rns == hi) { if ((rc = copy_in_cols(lo, 2 * rns))) return rc; }
            else { if ((rc = copy_in_cols(lo, rns)) || (rc = copy_in_cols(hi, rns))) return rc; }
            hipEvent_t ev;
            if ((rc = pool_event(m, &ev))) return rc;
            HIP_TRY(hipEventRecord(ev, m->copy_in));
            HIP_TRY(hipStreamWaitEvent(s, ev, 0));
            if (fuse) { pack_cols(Ld, in, lo, rns, s); pack_cols(Ld, in, hi, rns, s); }
        }
        if (fuse) launch_rec(l, Ld, gi_src, outp, true, cond, 0, rs0, rns);   // (the unfused twin runs once, after the phases: see below)
        else launch_rec(l, Ld, gi_src, outp, false, nullptr, 0, rs0, rns);
        m->last.rec_launches++;
        if (p < n_first || !(side_gemm || side_head)) continue;   // before T/2 steps no column has both directions
        hipEvent_t ev;
        if ((rc = pool_event(m, &ev))) return rc;
        HIP_TRY(hipEventRecord(ev, s));
        HIP_TRY(hipStreamWaitEvent(m->side, ev, 0));
        const int lo0 = T - ph[p + 1], hi0 = ph[p], len = ph[p + 1] - ph[p];
        if (side_gemm) {
            if (!gspan_open) { if ((rc = tm.begin(SLOT_GI0 + 1, m->side, &gspan))) return rc; gspan_open = true; }
            launch_gemm(m->layers[1], outp, m->gi2, m->side, lo0 / kGemmSteps, len / kGemmSteps);
            launch_gemm(m->layers[1], outp, m->gi2, m->side, hi0 / kGemmSteps, len / kGemmSteps);
        } else {
            launch_head(outp, m->side, lo0, len);
            launch_head(outp, m->side, hi0, len);
            if (P.stream_out) {
                hipEvent_t hv;
                if ((rc = pool_event(m, &hv))) return rc;
                HIP_TRY(hipEventRecord(hv, m->side));
                if (lo0 + len == hi0) out_ranges.push_back({hv, lo0, 2 * len, p});
                else { out_ranges.push_back({hv, lo0, len, p}); out_ranges.push_back({hv, hi0, len, p}); }
            }
        }
    }
    m->last.rec_launches--;   // (the caller counts the layer once)
    if (gspan_open && (rc = tm.end_at(gspan))) return rc;
    if (side_gemm || side_head) {
        hipEvent_t done;
        if ((rc = pool_event(m, &done))) return rc;
        HIP_TRY(hipEventRecord(done, m->side));
        HIP_TRY(hipStreamWaitEvent(s, done, 0));
    }
    if (l == 0 && fuse) {
        // out-of-range input (flag raised by k_pack_x): the fused phases were no-ops and the side
        // stream projected stale activations.  The unfused twin now runs the whole layer and a
        // conditional GEMM redoes the projection; all are empty launches otherwise.
        if (slabs && P.need_gi) launch_gi_small(l, Ld, cond);
        if (P.need_gi) launch_rec_fallback(Ld, gi_src, outp, cond, 0, T);
        if (side_gemm)
            launch_gemm(m->layers[1], outp, m->gi2, s, 0, (T + kGemmSteps - 1) / kGemmSteps, cond, 1);
    }
    if (side_gemm) gemm_done = true;
    if (side_head) head_done = true;
    return MDK_OK;
}

// one layer: its projection (unless fused), its recurrence in the form the plan chose
int Pass::layer(int l) {
    const int T = P.T, L = P.L;
    const LayerDev &Ld = m->layers[l];
    float *outp = m->act[l & 1];
    const float *gi_src = (l == 1 && gemm_done) ? gi_l1 : m->gi;
    const bool fuse = (l == 0) && P.fuse0;
    const int *cond = fuse ? m->oor_flag : nullptr;
    const bool fused_proj = l >= 1 && P.fuse_proj;
    // device-resident x: the packing of all but the first slab pair runs on the side stream under the
    // first recurrence phases instead of in front of them (0.25 ms of k_pack_x at 200 x 10000)
    const bool dev_slabs = !P.io_in && fuse && P.can_chunk && l == 0 && m->opt_overlap;
    const bool slabs = (P.stream_in || dev_slabs) && l == 0;       // this layer's recurrence starts slab by slab
    const bool side_gemm = P.overlap && l == 0;                    // layer 1's projection behind this layer's chunks
    const bool side_head = (P.overlap || P.stream_out) && l == L - 1 && L >= 2;   // classifier head behind the chunks
    int rc;
    if (l == 1 && m->wait_before_l1) HIP_TRY(hipStreamWaitEvent(s, m->wait_before_l1, 0));
    if ((rc = tm.begin(SLOT_GI0 + l))) return rc;
    if (fuse) {
        const size_t need = (size_t)P.n_wg * T * kXfragLanes;
        if (need > m->xfrag_cap) {
            free_dev(m->xfrag); m->xfrag = nullptr; m->xfrag_cap = 0;
            HIP_TRY(hipMalloc((void **)&m->xfrag, need * sizeof(half8)));
            m->xfrag_cap = need;
        }
        HIP_TRY(hipMemsetAsync(m->oor_flag, 0, sizeof(int), s));
        if (!slabs) pack_cols(Ld, in, 0, T, s);
    }
    if (l == 0) {
        if (!slabs && P.need_gi) launch_gi_small(l, Ld, cond);
    } else {
        if (!(l == 1 && gemm_done) && !fused_proj) launch_gemm(Ld, in, m->gi, s, 0, (T + kGemmSteps - 1) / kGemmSteps);
    }
    if ((rc = tm.end())) return rc;
    size_t rspan = 0;
    if ((rc = tm.begin(SLOT_REC0 + l, (hipStream_t)-1, &rspan))) return rc;
    if (P.ablated) {
#ifdef MDK_DEBUG_HOOKS
        if ((rc = launch_rec_ablated(Ld, gi_src, outp, 0, T))) return rc;
#endif
    } else if (P.final_head && l == L - 1) {
        if ((rc = layer_final_head(l, Ld, outp))) return rc;
    } else if (slabs || side_gemm || side_head) {
        if ((rc = layer_phased(l, Ld, outp, fuse, slabs, dev_slabs, side_gemm, side_head))) return rc;
    } else if (fuse) {
        launch_rec(l, Ld, gi_src, outp, true, cond, 0, 0, T);     // fused: runs unless the range flag is up
        if (P.need_gi) launch_rec_fallback(Ld, gi_src, outp, cond, 0, T);        // unfused twin: runs only on the flag
    } else {
        launch_rec(l, Ld, gi_src, outp, false, nullptr, 0, 0, T);
    }
    if ((rc = tm.end_at(rspan))) return rc;
    m->last.rec_launches++;
    if (l == 0) HIP_TRY(hipEventRecord(m->l0_done, s));
    in = outp;
    return MDK_OK;
}

// the probabilities' way home (host entries)
int Pass::copy_out() {
    const int T = P.T, nb = P.nb, C = m->desc.num_classes;
    const size_t p_bytes = (size_t)nb * T * C * sizeof(float);
    int rc;
    if (P.sp_out && !P.stream_out) {
        HIP_TRY(hipMemcpyAsync(io->p_host, probs, (size_t)sp->B * sp->T * C * sizeof(float), hipMemcpyDeviceToHost, s));
    } else if (P.sp_out) {
        m->last.host_streamed |= 2;
        // split host path: each launch of the final-head scan delivered, for chunk k, the real columns core_k /\ (start[k] +
        // [t0, t0 + nt)): they leave for the caller's buffer behind the launch's event as 2-D DMA copies (B rows of a few KB:
        // 37-50 GB/s, profiles/r4_experiments/dma2d_probe.txt) -- DMA, not a copy kernel: any kernel that talks to host memory
        // from the recurrence's CUs stalls it (profiles/r4_experiments/README.md)
        // The ranges of the LAST TWO launches do not go to the DMA queue where the CUs can take them (HostIO::p_host_dev): the
        // queue is still behind when the scan ends -- in half precision by 0.45 ms -- and a kernel behind the last recurrence
        // writes them home at the full PCIe rate while the queue finishes what it has (k_tail_to_host).
        TailRanges tail{};
        const int last_launch = out_ranges.empty() ? 0 : out_ranges.back().launch;
        int n_copy = 0;
        for (const OutRange &r : out_ranges) {
            // (half precision only: in fp32-parity mode the scan is slow enough for the queue to keep up, and the kernel would only
            // add its own 0.1 ms behind the last recurrence -- measured 7.95 -> 8.05 ms; half precision 5.74 -> 5.58 ms)
            // (two launches: one 5.68, two 5.57, three 5.76, four 6.2 ms per half-precision call -- what the kernel takes it takes
            // AFTER the scan, what the queue takes it takes under it)
            if (io->p_host_dev && P.hp && r.launch + 2 > last_launch && tail.n < 4 && out_ranges.size() > 4) {
                tail.t0[tail.n] = r.t0; tail.nt[tail.n] = r.nt; tail.n++;
                continue;
            }
            HIP_TRY(hipStreamWaitEvent(m->copy_out, r.ready, 0));
            HIP_TRY(hipStreamWaitEvent(m->copy_out2, r.ready, 0));
            for (int k = 0; k < sp->S; ++k) {
                const int a = std::max(sp->core0[k], sp->start[k] + r.t0), b = std::min(sp->core0[k + 1], sp->start[k] + r.t0 + r.nt);
                if (a >= b) continue;
                // (copies alternate between two streams: each costs ~10 us of set-up on top of its bytes, and two DMA engines
                // work side by side)
                HIP_TRY(hipMemcpy2DAsync(io->p_host + (size_t)a * C, (size_t)sp->T * C * sizeof(float),
                                         probs + (size_t)a * C, (size_t)sp->T * C * sizeof(float),
                                         (size_t)(b - a) * C * sizeof(float), (size_t)sp->B, hipMemcpyDeviceToHost,
                                         (n_copy++ & 1) ? m->copy_out2 : m->copy_out));
            }
        }
        if (tail.n) {
            hipLaunchKernelGGL(k_tail_to_host, dim3((unsigned)sp->B, (unsigned)(tail.n * sp->S)), dim3(256), 0, s, (const float *)probs,
                               io->p_host_dev, *sp, tail, C);
            m->last.host_streamed |= 32;
        }
        for (hipStream_t cs : {m->copy_out, m->copy_out2}) {
            hipEvent_t done;
            if ((rc = pool_event(m, &done))) return rc;
            HIP_TRY(hipEventRecord(done, cs));
            if (join_later) join_later->push_back(done);
            else HIP_TRY(hipStreamWaitEvent(s, done, 0));
        }
    } else if (P.io_out) {
        if (out_ranges.empty()) {      // head not chunked, or its chunks were not streamed: one copy behind it
            HIP_TRY(hipMemcpyAsync(io->p_host, probs, p_bytes, hipMemcpyDeviceToHost, s));
        } else {
            // every kernel of the pass is enqueued: now the copies, each behind its head chunk
            for (const OutRange &r : out_ranges) {
                HIP_TRY(hipStreamWaitEvent(m->copy_out, r.ready, 0));
                HIP_TRY(hipMemcpy2DAsync(io->p_host + (size_t)r.t0 * C, (size_t)T * C * sizeof(float),
                                         probs + (size_t)r.t0 * C, (size_t)T * C * sizeof(float),
                                         (size_t)r.nt * C * sizeof(float), (size_t)nb, hipMemcpyDeviceToHost,
                                         m->copy_out));
            }
            hipEvent_t done;
            if ((rc = pool_event(m, &done))) return rc;
            HIP_TRY(hipEventRecord(done, m->copy_out));
            HIP_TRY(hipStreamWaitEvent(s, done, 0));   // a synchronize on `s` then covers the copies
        }
    }
    return MDK_OK;
}

int Pass::run() {
    int rc;
    m->ov_next = 0;
    if (P.exact) return run_exact();
    const int T = P.T, L = P.L;
    const size_t x_bytes = (size_t)P.nb * T * m->desc.num_features * sizeof(float);
    if (P.io_in && !P.stream_in)
        HIP_TRY(hipMemcpyAsync(const_cast<float *>(x), io->x_host, x_bytes, hipMemcpyHostToDevice, s));
    gi_l1 = m->gi;
    if (P.overlap) {
        const size_t rows = (size_t)P.n_tiles * kTileWin * T;
        if (rows > m->gi2_rows) {
            free_dev(m->gi2); m->gi2 = nullptr; m->gi2_rows = 0;
            HIP_TRY(hipMalloc((void **)&m->gi2, (size_t)P.D * rows * kG * sizeof(float)));
            m->gi2_rows = rows;
        }
        gi_l1 = m->gi2;
    }
    in = x;
    for (int l = 0; l < L; ++l)
        if ((rc = layer(l))) return rc;
    if ((rc = tm.begin(SLOT_HEAD))) return rc;
    if (!head_done) launch_head(in, s, 0, T);
    if ((rc = tm.end())) return rc;
    HIP_TRY(hipGetLastError());
    // every kernel of the pass is enqueued (a split call adds its certificate kernel and records again): what the OTHER context's
    // next forward waits for where two passes cannot share the chip -- not for the result copies that follow
    HIP_TRY(hipEventRecord(m->kernels_done, s));
    m->last_wgs = P.n_wg * P.D * m->opt_gpu_share;
    return copy_out();
}

static int forward_pass(mdk_gru *m, const PassPlan &P, const float *x, float *probs, hipStream_t s,
                        EvTimer &tm, const HostIO *io, const SplitPlan *sp = nullptr, std::vector<hipEvent_t> *join_later = nullptr) {
    Pass pass{m, P, x, probs, s, tm, io, sp, join_later};
    return pass.run();
}

// the range flag of the pass(es) just enqueued, for callers that promised to look (PassPlan::need_gi): true = the input left
// the fp16 range and nothing was there to take over -- the model is marked and the call has to be repeated
static int range_flag_raised(mdk_gru *m, hipStream_t s, bool *raised) {
    if (!m->oor_host) HIP_TRY(hipHostMalloc((void **)&m->oor_host, sizeof(int), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(m->oor_host, m->oor_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *raised = *m->oor_host != 0;
    if (*raised && !m->oor_seen) {
        m->oor_seen = true;
        fprintf(stderr, "[medaka_amd] input beyond fp16 range (un-normalised counts?): the exact fp32 projection takes over -- this call is "
                        "repeated, later ones decide on the device\n");
    }
    return MDK_OK;
}

static int finish_timing(mdk_gru *m, EvTimer &tm, hipStream_t s) {
    if (!m->timing) return MDK_OK;
    HIP_TRY(hipStreamSynchronize(s));
    for (auto &sp : tm.spans) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, m->ev[sp.e0], m->ev[sp.e1]));
        const int slot = sp.slot;
        if (slot >= SLOT_GI0 && slot < SLOT_GI0 + 4) m->last.gi_ms[slot - SLOT_GI0] += ms;
        else if (slot >= SLOT_REC0 && slot < SLOT_REC0 + 4) m->last.rec_ms[slot - SLOT_REC0] += ms;
        else if (slot == SLOT_HEAD) m->last.head_ms += ms;
    }
    if (!tm.spans.empty()) {   // first event recorded .. last event of the last (head) span, both on `s`
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, m->ev[tm.spans.front().e0], m->ev[tm.spans.back().e1]));
        m->last.total_ms = ms;
    }
    return MDK_OK;
}

// all passes of one call; x_host / probs_host (may be null) select the streamed host path per pass.  `lean`: plan for the
// regime without gi (plan_pass); the CALLER looks at the range flag afterwards (range_flag_raised) and repeats without it
static int run_passes(mdk_gru *m, const float *x_dev, int B, int T, float *probs_dev, hipStream_t s,
                      const float *x_host, float *probs_host, bool lean = false) {
    memset(&m->last, 0, sizeof(m->last));
    m->last.n_layers = m->desc.num_layers;
    // windows per pass, bounded so that the workspace stays within a fixed column budget
    // and balanced: equal passes keep every launch's grid full (a 838 + 162 split of 1000 windows
    // costs two full-length recurrences; 2 x 500 costs the same two, 1 x 1000 costs one)
    const size_t budget = m->max_rows_per_pass ? m->max_rows_per_pass : kMaxRowsPerPass;
    const size_t fit = std::max<size_t>(1, budget / (size_t)T);
    const size_t n_pass = ((size_t)B + fit - 1) / fit;
    size_t per_pass = ((size_t)B + n_pass - 1) / n_pass;
    if (n_pass > 1 && fit >= kTileWin)             // full recurrence tiles in all but the last pass
        per_pass = std::min(fit - fit % kTileWin, (per_pass + kTileWin - 1) / kTileWin * kTileWin);
    int rc;
    if (n_pass > 1) lean = false;                  // (the range flag is per pass: only a single pass can leave it to the caller)
    bool need_gi = !lean;
    if (lean) {
        PassPlan P;
        if ((rc = plan_pass(m, (int)std::min(per_pass, (size_t)B), T, nullptr, nullptr, P, true, true))) return rc;
        need_gi = P.need_gi;
    }
    if ((rc = ensure_workspace(m, ((per_pass + kTileWin - 1) / kTileWin * kTileWin) * (size_t)T, need_gi))) return rc;
    EvTimer tm{m, s};
    const size_t F = m->desc.num_features, C = m->desc.num_classes;
    for (size_t b0 = 0; b0 < (size_t)B; b0 += per_pass) {
        const int nb = (int)std::min(per_pass, (size_t)B - b0);
        HostIO io;
        if (x_host) io.x_host = x_host + b0 * T * F;
        if (probs_host) io.p_host = probs_host + b0 * T * C;
        const HostIO *iop = (x_host || probs_host) ? &io : nullptr;
        PassPlan P;                                  // (the range flag is per pass: the fallback stays on the device here)
        if ((rc = plan_pass(m, nb, T, iop, nullptr, P, lean, lean))) return rc;
        if ((rc = forward_pass(m, P, x_dev + b0 * T * F, probs_dev + b0 * T * C, s, tm, iop))) return rc;
    }
    return finish_timing(m, tm, s);
}
