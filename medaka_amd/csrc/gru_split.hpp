// The split scan on the host side: shape arithmetic, enqueue / finish halves of a split call, the call-level state machine
// (run_forward: certificate, margin learner, probe, audits, fallback) and the first attempt of a call started ahead (start_call).
// Part of api.hip (included there after gru_pass.hpp).
#pragma once
// ---- split scan (scan_split.hpp): plan, run on the virtual batch, certify, fall back
// The shape arithmetic of a split, free of any model state (also exported as mdk_split_plan for hosts and CPU tests).
//   mode: 1 auto, n >= 2 forced chunk count; share: processes on this GPU; G: margin; budget: column budget of a pass
static bool plan_split_shape(int B, int T, int share, int mode, int G, size_t budget, SplitPlan &p) {
    p.S = 1; p.B = B; p.T = T; p.Tv = T; p.G = 0;
    if (B < 1 || T < 1 || mode < 1 || G < 8 || share < 1) return false;
    // The recurrence holds 8 windows per work-group and direction at most (fp32-parity mode): 1024 chunk-windows are
    // one round of work-groups on 256 CUs -- more than that queues (profiles/r3_experiments/scan_split/time_probe.txt).
    // K processes sharing the GPU (launch.py --procs-per-gpu): their kernels interleave -- one is in its projection
    // while another is in a recurrence -- and 1600 / K chunk-windows each measured best (profiles/r3_fed_loop_shared.txt:
    // K = 3 at batch 200, whole fed loop: 249 M columns/s unsplit, 290 M with 2 chunks, 284 M with 3)
    const int max_win = share == 1 ? 1024 : 1600 / share;
    int S = (mode >= 2) ? mode : max_win / B;
    // alone, two chunks of 500 windows gain 8 % on the device and nothing host to host: not worth the margins
    if (mode == 1 && S < (share == 1 ? 3 : 2)) return false;
    S = std::min({S, kMaxSplit, T / (4 * G)});      // a chunk's own columns are at least twice its two margins
    if (S < 2) return false;
    int max_core = 0, core0[kMaxSplit + 1];
    for (int k = 0; k <= S; ++k) core0[k] = (int)((long)T * k / S);
    for (int k = 0; k < S; ++k) max_core = std::max(max_core, core0[k + 1] - core0[k]);
    const int Tv = (max_core + 2 * G + 15) / 16 * 16;
    if (Tv >= T || (size_t)S * B * Tv > budget) return false;
    p.S = S; p.G = G; p.Tv = Tv;
    for (int k = 0; k <= S; ++k) p.core0[k] = core0[k];
    for (int k = 0; k < S; ++k) p.start[k] = std::min(std::max(core0[k] - G, 0), T - Tv);
    return true;
}

// The margin learner on a model that certifies iff the margin is >= `need` (0: never), with differences at the noise floor:
// n_calls calls from `start`; margins[i] = the margin call i was ANSWERED at (0: sequentially), forwards[i] = split forwards
// it cost (rejected ones included).  Device-free: the CPU tests drive the state machine through this.
extern "C" int mdk_margin_sim(int start, int adapt, int need, int n_calls, int *margins, int *forwards) {
    if (start < 16 || start > 4096 || adapt < 0 || need < 0 || n_calls < 0 || !margins || !forwards)
        return fail(MDK_ERR_ARG, "bad argument");
    MarginLearner L;
    bool disabled = false;
    for (int i = 0; i < n_calls; ++i) {
        margins[i] = 0; forwards[i] = 0;
        if (disabled) continue;
        for (;;) {
            const int G = L.cur ? L.cur : start;
            forwards[i]++;
            if (need > 0 && G >= need) { L.certified(G, 0.f, 1.f, adapt); margins[i] = G; break; }
            int back = 0;
            if (L.rejected(G, &back) == MarginLearner::GIVE_UP) { disabled = true; break; }
        }
    }
    return MDK_OK;
}

// plan_pass on a model that exists on paper only (default options): nothing here touches a device
extern "C" int mdk_pass_plan(const mdk_gru_desc *desc, int precision, int gpu_share, int windows, int T, int host_io, int split_chunks,
                             int mode, mdk_pass_shape *out) {
    if (!desc || !out) return fail(MDK_ERR_ARG, "null argument");
    if (windows < 1 || T < 1 || gpu_share < 1 || gpu_share > 8 || split_chunks < 0 || split_chunks > kMaxSplit ||
        (precision != MDK_PREC_FP32 && precision != MDK_PREC_FP16) || desc->num_layers < 1 || desc->num_features < 1)
        return fail(MDK_ERR_ARG, "bad argument (windows=%d T=%d gpu_share=%d split_chunks=%d precision=%d)", windows, T, gpu_share, split_chunks, precision);
    mdk_gru m;
    m.desc = *desc;
    m.D = desc->bidirectional ? 2 : 1;
    m.precision = precision;
    m.opt_gpu_share = gpu_share;
    m.oor_seen = (mode & 4) != 0;
    m.layers.resize((size_t)desc->num_layers);
    m.layers[0].K = desc->num_features;
    // (the fused layer-0 projection exists when the features + the bias row fit one 16-slot k-group: mdk_gru_create)
    m.layers[0].wx_frag = desc->num_features + 1 <= 16 ? reinterpret_cast<half8 *>(sizeof(half8)) : nullptr;
    static const float dummy = 0.f;
    HostIO io;
    if (host_io & 1) io.x_host = &dummy;
    if (host_io & 2) io.p_host = const_cast<float *>(&dummy);
    SplitPlan sp;
    sp.S = split_chunks;
    PassPlan P;
    const int rc = plan_pass(&m, windows, T, (host_io & 3) ? &io : nullptr, split_chunks > 1 ? &sp : nullptr, P, (mode & 1) != 0, (mode & 2) != 0);
    if (rc) return rc;
    memset(out, 0, sizeof(*out));
    out->windows_per_group = 4 * P.nq; out->work_groups = P.n_wg;
    out->fuse_layer0 = P.fuse0; out->fuse_projection = P.fuse_proj; out->fuse_head = P.fuse_head; out->final_head = P.final_head;
    out->overlap_gemm = P.overlap; out->stream_in = P.stream_in; out->stream_out = P.stream_out;
    out->needs_gi = P.need_gi;
    return MDK_OK;
}

extern "C" int mdk_split_plan(int B, int T, int gpu_share, int scan_split, int margin, mdk_split_shape *out) {
    if (!out) return fail(MDK_ERR_ARG, "null argument");
    if (B < 0 || T < 0 || gpu_share < 1 || gpu_share > 8 || scan_split < 0 || scan_split > kMaxSplit || margin < 16 || margin > 4096 || margin % 8)
        return fail(MDK_ERR_ARG, "bad argument (B=%d T=%d gpu_share=%d scan_split=%d margin=%d)", B, T, gpu_share, scan_split, margin);
    SplitPlan p;
    plan_split_shape(B, T, gpu_share, scan_split, margin, kMaxRowsPerPass, p);
    memset(out, 0, sizeof(*out));
    out->chunks = p.S; out->columns = p.S > 1 ? p.Tv : T; out->margin = p.S > 1 ? p.G : 0;
    for (int k = 0; k < p.S && p.S > 1; ++k) { out->start[k] = p.start[k]; out->first[k] = p.core0[k]; out->last[k] = p.core0[k + 1]; }
    if (p.S == 1) { out->start[0] = 0; out->first[0] = 0; out->last[0] = T; }
    return MDK_OK;
}

static bool plan_split(const mdk_gru *m, int B, int T, SplitPlan &p) {
    static const int env_abl = getenv("MDK_ABLATE") ? atoi(getenv("MDK_ABLATE")) : 0;
    p.S = 1;
    if (m->opt_scan_split == 0 || (m->split_disabled && m->opt_scan_split == 1)) return false;
    if (m->variant != MDK_VARIANT_MFMA || m->D != 2 || m->desc.num_layers != 2 || m->opt_ablate || env_abl) return false;
    if (m->layers[0].K > 16) return false;
    return plan_split_shape(B, T, m->opt_gpu_share, m->opt_scan_split, m->margin.cur ? m->margin.cur : m->opt_split_margin,
                            m->max_rows_per_pass ? m->max_rows_per_pass : kMaxRowsPerPass, p);
}

// A split call in two halves, so that the staged entry can enqueue the NEXT batch's forward before it waits for this one's
// certificate: split_enqueue = every launch and copy of the call (nothing here waits for the device), split_finish = the wait,
// the range flag, the certificate.  run_split = one after the other.
static int split_enqueue(mdk_gru *m, const SplitPlan &sp, const float *x_dev, float *probs_dev, hipStream_t s,
                         const float *x_host, float *probs_host, EvTimer &tm, bool *need_gi) {
    const size_t F = m->desc.num_features;
    const int Bv = sp.S * sp.B;
    const size_t cols = (size_t)Bv * sp.Tv;
    memset(&m->last, 0, sizeof(m->last));
    m->last.n_layers = m->desc.num_layers;
    if (cols * F > m->xv_cap) {
        free_dev(m->xv); m->xv = nullptr; m->xv_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->xv, cols * F * sizeof(float)));
        m->xv_cap = cols * F;
    }
    if (!m->split_flag) HIP_TRY(hipMalloc((void **)&m->split_flag, kSplitFlagWords * sizeof(unsigned)));
    if (!m->split_host) HIP_TRY(hipHostMalloc((void **)&m->split_host, kSplitFlagWords * sizeof(unsigned), hipHostMallocDefault));
    if (!m->oor_host) HIP_TRY(hipHostMalloc((void **)&m->oor_host, sizeof(int), hipHostMallocDefault));
    HostIO io;
    io.p_host = probs_host;
    if (probs_host && probs_host == m->tail_host) io.p_host_dev = m->tail_dev;      // (the cold host entry: the last chunks may leave by kernel)
    PassPlan P;                    // this call synchronises for its certificate anyway: it looks at the range flag itself
    int rc = plan_pass(m, Bv, sp.Tv, probs_host ? &io : nullptr, &sp, P, /*host_checks_range=*/true);
    if (rc) return rc;
    *need_gi = P.need_gi;
    if ((rc = ensure_workspace(m, (((size_t)Bv + kTileWin - 1) / kTileWin * kTileWin) * (size_t)sp.Tv, P.need_gi))) return rc;
    static const bool dbg_spans = getenv("MDK_EARLY_DEBUG") != nullptr;
    if (dbg_spans) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventRecord(a, s));
        m->dbg_spans.push_back({a, b});
    }
    HIP_TRY(hipMemsetAsync(m->split_flag, 0, kSplitFlagWords * sizeof(unsigned), s));
    // Host buffers.  x crosses PCIe whole, one contiguous copy in front of the forward: all of it is needed within the
    // first half of layer 0 (1 ms of work against 1.4 ms of PCIe), so slabs gain nothing -- measured both as DMA slabs and
    // as copy kernels on the mapped buffer (profiles/r4_experiments/README.md); callers that can, hand x over early
    // (medaka_amd.torch_ext: the batch is on its way to the device while the previous one is still being computed).
    // The probabilities leave in column chunks, as 2-D DMA copies under the rest of the last layer's scan, whose second
    // half writes them itself (rec_fused.hpp HEAD = 2; forward_pass decides: `host_streamed` bit 1) -- behind a separate
    // head kernel they did not (a kernel beside a recurrence that holds every CU crawls until the recurrence is over: 9.4 ms
    // against 9.1, profiles/r4_experiments/host_path_timeline_v4_dma_out.txt; "stream_host" = 2 still forces that form).
    // What stays exposed is the last launch's chunk; a shape that cannot be chunked leaves as one copy behind the forward.
    if (x_host)
        HIP_TRY(hipMemcpyAsync(const_cast<float *>(x_dev), x_host, (size_t)sp.B * sp.T * F * sizeof(float), hipMemcpyHostToDevice, s));
    std::vector<hipEvent_t> out_done;      // (the last result chunks are still crossing PCIe while the certificate is computed)
    // (x_dev is the REAL batch: layer 0's operands are packed straight from it, chunk by chunk; m->xv -- the virtual batch in
    // memory -- is written only if the exact-projection fallback needs it)
    rc = forward_pass(m, P, x_dev, probs_dev, s, tm, probs_host ? &io : nullptr, &sp, &out_done);
    if (rc) return rc;
    hipLaunchKernelGGL(k_split_verify, dim3((unsigned)((sp.B + kVerifyWin - 1) / kVerifyWin), (unsigned)(8 * (sp.S - 1))), dim3(128), 0, s,
                       (const float *)m->act[0], (const float *)m->act[1], sp, m->split_flag);
    HIP_TRY(hipMemcpyAsync(m->split_host, m->split_flag, kSplitFlagWords * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    // (no gi, hence no device-side fallback in this pass: the range flag goes home with the certificate)
    if (!P.need_gi) HIP_TRY(hipMemcpyAsync(m->oor_host, m->oor_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(m->kernels_done, s));        // the call's last kernel: the other context's next forward may start behind it
    if (dbg_spans) HIP_TRY(hipEventRecord(m->dbg_spans.back().second, s));
    for (hipEvent_t e : out_done) HIP_TRY(hipStreamWaitEvent(s, e, 0));
    HIP_TRY(hipGetLastError());
    return MDK_OK;
}

static int run_split(mdk_gru *m, const SplitPlan &sp, const float *x_dev, float *probs_dev, hipStream_t s,
                     const float *x_host, float *probs_host, bool *certified);

static int split_finish(mdk_gru *m, const SplitPlan &sp, bool need_gi, EvTimer &tm, const float *x_dev, float *probs_dev, hipStream_t s,
                        float *probs_host, bool *certified) {
    int rc;
    if ((rc = finish_timing(m, tm, s))) return rc;
    HIP_TRY(hipStreamSynchronize(s));      // the certificate decides what this call returns
    if (!need_gi && *m->oor_host != 0) {
        // the input left the fp16 range and nothing was there to take over: the model is marked and the call repeated, with gi
        // and the device-side decision, which later calls keep
        if (!m->oor_seen) {
            m->oor_seen = true;
            fprintf(stderr, "[medaka_amd] input beyond fp16 range (un-normalised counts?): the exact fp32 projection takes over -- this call is "
                            "repeated, later ones decide on the device\n");
        }
        return run_split(m, sp, x_dev, probs_dev, s, nullptr, probs_host, certified);
    }
    const float eps = m->precision == MDK_PREC_FP16 ? kSplitEpsHalf : kSplitEps;
    float worst = 0.f;
    for (int y = 0; y < 8 * (sp.S - 1); ++y) {
        float d;
        memcpy(&d, &m->split_host[y], sizeof(float));
        worst = std::max(worst, d);
    }
    *certified = worst <= eps;
    static const bool dbg = getenv("MDK_SPLIT_DEBUG") != nullptr;
    if (dbg) {
        fprintf(stderr, "[mdk split] %d x %d as %d chunks of %d columns (margin %d): %s, worst %.3g\n", sp.B, sp.T, sp.S, sp.Tv, sp.G,
                *certified ? "certified" : "REJECTED", worst);
        for (int y = 0; y < 8 * (sp.S - 1); ++y) {
            float d;
            memcpy(&d, &m->split_host[y], sizeof(float));
            fprintf(stderr, "    junction %d (column %d) layer %d direction %d point %d: %.3g\n", y >> 3, sp.core0[(y >> 3) + 1], (y >> 2) & 1,
                    (y >> 1) & 1, y & 1, d);
        }
    }
    m->last_split.chunks = sp.S; m->last_split.margin = sp.G; m->last_split.columns = sp.Tv;
    m->last_split.max_delta = worst;
    m->last_split.status = *certified ? MDK_SPLIT_CERTIFIED : MDK_SPLIT_REJECTED;
    return MDK_OK;
}

static int run_split(mdk_gru *m, const SplitPlan &sp, const float *x_dev, float *probs_dev, hipStream_t s,
                     const float *x_host, float *probs_host, bool *certified) {
    EvTimer tm{m, s};
    bool need_gi = true;
    int rc = split_enqueue(m, sp, x_dev, probs_dev, s, x_host, probs_host, tm, &need_gi);
    if (rc) return rc;
    return split_finish(m, sp, need_gi, tm, x_dev, probs_dev, s, probs_host, certified);
}

static void report_audits(mdk_gru *m) {
    m->last_split.audits = (int)std::min<long>(m->audits_done, 0x7fffffff);
    m->last_split.audit_failures = m->audit_failures;
    m->last_split.audit_worst_dp = m->audit_worst;
    m->last_split.probes = (int)std::min<long>(m->probes_done, 0x7fffffff);
    m->last_split.probe_max_delta = m->probe_last_delta;
}

// one call: split scan when the shape is latency-bound and the certificate holds, the sequential passes otherwise
// `pre` (staged entry only): the call's FIRST attempt is already enqueued on `s` in this context (start_call) -- a split scan whose
// certificate is still unread, or the sequential passes.  It is taken over if it is what this function would have enqueued now;
// otherwise (an option, the learner or the back-off moved in between) it is waited for and forgotten.
static bool same_split(const SplitPlan &a, const SplitPlan &b) {
    if (a.S != b.S || a.B != b.B || a.T != b.T || a.Tv != b.Tv || a.G != b.G) return false;
    for (int k = 0; k < a.S; ++k) if (a.start[k] != b.start[k] || a.core0[k] != b.core0[k]) return false;
    return a.core0[a.S] == b.core0[b.S];
}

static bool split_probe_due(const mdk_gru *m, const SplitPlan &sp) {
    return m->precision == MDK_PREC_FP16 && m->opt_scan_split == 1 && m->opt_split_probe &&
           (std::find(m->probed_ok.begin(), m->probed_ok.end(), sp.G) == m->probed_ok.end() ||
            (m->opt_split_audit == 1 && m->opt_split_audit_every > 0 && m->split_calls_since_audit + 1 >= m->opt_split_audit_every));
}

static int run_forward(mdk_gru *m, const float *x_dev, int B, int T, float *probs_dev, hipStream_t s,
                       const float *x_host, float *probs_host, mdk_gru::Started *pre = nullptr) {
    SplitPlan sp;
    int rc;
    bool first_attempt = true;
    auto forget_pre = [&]() -> int {
        if (pre && pre->valid) {
            pre->valid = false;
            m->early_dropped++;
            HIP_TRY(hipStreamSynchronize(s));       // (its result copies target the caller's buffer: nothing of it may still be running)
        }
        return MDK_OK;
    };
    const int fallbacks = m->last_split.fallbacks;
    memset(&m->last_split, 0, sizeof(m->last_split));
    m->last_split.chunks = 1; m->last_split.columns = T; m->last_split.fallbacks = fallbacks;
    // A rejection at the largest margin may be the INPUT's doing (a zero-coverage run, a stretch the model was never
    // trained on: dynamics that do not forget THERE), not the model's: the split is tried again after a back-off of
    // 64, 128, ... 4096 calls, at the largest margin (one rejected forward per retry, < 1 % of the calls in between).
    if (m->split_disabled && m->split_retry_in > 0 && --m->split_retry_in == 0) m->split_disabled = false;
    m->last_split.status = m->split_disabled ? MDK_SPLIT_DISABLED : MDK_SPLIT_NOT_USED;
    report_audits(m);
#ifdef MDK_DEBUG_HOOKS
    static const bool keep = getenv("MDK_SPLIT_KEEP") != nullptr;   // debug builds only: deliver a rejected split as it is
#else
    const bool keep = false;
#endif
    while (plan_split(m, B, T, sp)) {
        bool ok = false;
        // Half precision (what `medaka inference` runs by default, prediction.py:164-168).  Its certificate compares the fp16
        // images the scan keeps of h: two merged scans still differ by 1e-4 .. 3e-4 of rounding noise there, the threshold is
        // 2^-10, and a state that has NOT merged by up to 1e-3 passes unseen -- the margin learner then walks down to margins the
        // fp32-parity certificate rejects for the same weights (round 5: 64 where fp32 parity needs 128).  So in auto mode a margin
        // is used in half mode only after a call certified at it in FP32-PARITY mode: the call is run once more with the hi/lo
        // operands and the 2^-18 threshold (result discarded, x stays on the device), once per margin the learner visits and again
        // with every standing audit; a rejected probe is a rejected certificate (the margin climbs / the trial goes back).
        const bool probe_due = split_probe_due(m, sp);
        const bool use_pre = first_attempt && pre && pre->valid && pre->split && pre->precision == m->precision && !probe_due &&
                             same_split(sp, pre->sp);
        if (first_attempt && !use_pre && (rc = forget_pre())) return rc;
        first_attempt = false;
        bool probe_rejected = false;
        if (probe_due) {
            m->precision = MDK_PREC_FP32;
            bool pok = false;
            rc = run_split(m, sp, x_dev, probs_dev, s, x_host, nullptr, &pok);
            m->precision = MDK_PREC_FP16;
            if (rc) return rc;
            x_host = nullptr;                     // x is on the device from here on
            m->probes_done++;
            m->probe_last_delta = m->last_split.max_delta;
            m->probed_ok.erase(std::remove(m->probed_ok.begin(), m->probed_ok.end(), sp.G), m->probed_ok.end());
            if (pok) m->probed_ok.push_back(sp.G);
            else probe_rejected = true;
        }
        if (!probe_rejected) {
            if (use_pre) {
                pre->valid = false;
                m->early_used++;
                EvTimer none{m, s};
                rc = split_finish(m, sp, pre->need_gi, none, x_dev, probs_dev, s, probs_host, &ok);
            } else {
                rc = run_split(m, sp, x_dev, probs_dev, s, x_host, probs_host, &ok);
            }
            if (rc) return rc;
        }
        report_audits(m);
        if (keep) return MDK_OK;
        if (ok) {
            m->split_backoff = 0;
            // The margin is the split's price (12.8 % of all columns at 128, 5.7 % at 64) and what it has to be is the MODEL's
            // forgetting length: after `scan_split_adapt` certified calls in a row whose largest junction difference sat at the
            // rounding-noise floor (a quarter of the threshold), the next call tries one rung less.  A trial that is rejected
            // costs that one forward: the call is repeated at the margin that worked, and no shrink goes below it again.
            const float quiet_thr = 0.25f * (m->precision == MDK_PREC_FP16 ? kSplitEpsHalf : kSplitEps);
            const int was = m->margin.certified(sp.G, m->last_split.max_delta, quiet_thr, m->opt_scan_split == 1 ? m->opt_split_adapt : 0);
            if (was) fprintf(stderr, "[medaka_amd] split scan: certified at a margin of %d columns (was %d): kept\n", sp.G, was);
            // Audit.  The certificate argues from the states at the junctions; the audit looks at what is delivered: the call is
            // ALSO run as the sequential scan on the device and the two (B, T, C) results are compared in full.  Audited are the
            // first certified call of a model (and the first at every margin / precision it moves to) and, as a STANDING check on
            // whatever input the model meets later, every `scan_split_audit_every`-th certified call after that (default 256:
            // one sequential forward of ~2x a split forward's time per 256 calls, < 1 %; a concurrent low-priority audit was
            // tried first and cost far more -- any second tenant keeps the recurrence's work-groups from being resident
            // together).  A mismatch delivers the sequential result and turns the split off for the model.
            const int audit_key = sp.G | (m->precision << 16) | (1 << 24);
            const bool first = m->split_audited_key != audit_key;
            const bool periodic = !first && m->opt_split_audit_every > 0 && ++m->split_calls_since_audit >= m->opt_split_audit_every;
            if (m->opt_split_audit == 0 || (m->opt_split_audit == 1 && !first && !periodic)) return MDK_OK;
            m->split_calls_since_audit = 0;
            const size_t n = (size_t)B * T * m->desc.num_classes;
            if (n > m->audit_cap) {
                free_dev(m->audit); m->audit = nullptr; m->audit_cap = 0;
                HIP_TRY(hipMalloc((void **)&m->audit, n * sizeof(float)));
                m->audit_cap = n;
            }
            const mdk_gru_split certified = m->last_split;
            // (x_dev holds x also on the host path.)  The audit's scan is planned `lean`: it needs no gi -- 6 GB per buffer at
            // 200 x 10 000, which an audit used to allocate and give back: memory handed back to the driver is wiped by the
            // kernel ON THE DMA ENGINES, in the background, and while that ran (0.45 s for the two buffers) every strided copy of
            // the host path took 130 us longer -- the "slow DMA state" of the first 40 calls after every audit, found in round 5
            // (profiles/r5_experiments/README.md section 9).
            rc = run_passes(m, x_dev, B, T, m->audit, s, nullptr, nullptr, /*lean=*/true);
            if (rc) return rc;
            if (!m->oor_seen) {             // (possibly) no gi, no device-side fallback: was x inside fp16 range?  (if not: once more, with it)
                bool raised = false;
                if ((rc = range_flag_raised(m, s, &raised))) return rc;
                if (raised && (rc = run_passes(m, x_dev, B, T, m->audit, s, nullptr, nullptr))) return rc;
            }
            HIP_TRY(hipMemsetAsync(m->split_flag, 0, sizeof(unsigned), s));
            hipLaunchKernelGGL(k_split_audit, dim3((unsigned)std::min<size_t>((n + 255) / 256, 256 * 8)), dim3(256), 0, s,
                               (const float *)probs_dev, (const float *)m->audit, n, m->split_flag);
            HIP_TRY(hipMemcpyAsync(m->split_host, m->split_flag, sizeof(unsigned), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            // (a shape whose sequential scan cannot run fused -- T not a multiple of the strip -- did allocate gi: it STAYS, the
            // next audit of the shape needs it again and a hipFree of that size is 0.5 s of slow strided DMA, see above)
            float dp;
            memcpy(&dp, &m->split_host[0], sizeof(float));
            m->audits_done++;
            m->audit_worst = std::max(m->audit_worst, dp);
            m->last_split = certified;
            m->last_split.audited = 1;
            m->last_split.audit_max_dp = dp;
            if (dp <= (m->precision == MDK_PREC_FP16 ? kAuditTolHalf : kAuditTol)) {
                m->split_audited_key = audit_key;
                report_audits(m);
                return MDK_OK;
            }
            // never seen: certified junctions, different probabilities.  The sequential result is already there.
            fprintf(stderr, "[medaka_amd] split scan: an audit found |p_split - p_sequential| = %.3g behind a certified split (margin %d, "
                            "%s call): the sequential result is delivered and the split scan is off for this model\n", dp, sp.G,
                    first ? "first" : "a later");
            m->audit_failures++;
            m->last_split.status = MDK_SPLIT_REJECTED;
            m->last_split.fallbacks++;
            m->split_disabled = true;
            report_audits(m);
            HIP_TRY(hipMemcpyAsync(probs_dev, m->audit, n * sizeof(float), hipMemcpyDeviceToDevice, s));
            if (probs_host) HIP_TRY(hipMemcpyAsync(probs_host, m->audit, n * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            return MDK_OK;
        }
        // Some junction did not merge: this model remembers further back than the margin.  Auto mode tries again with
        // twice the margin and keeps it for later calls (said once on stderr).  A shape that no longer splits at the new
        // margin is answered sequentially -- this call only; the model is given up (sequential scans from then on) only by
        // a rejection AT kSplitMarginMax: a very long or chaotic memory.  A forced chunk count is not second-guessed: the
        // call is answered sequentially.
        m->last_split.fallbacks++;
        m->margin.quiet = 0;
        if (m->opt_scan_split != 1) break;
        int was_trial = 0;
        const MarginLearner::Next nx = m->margin.rejected(sp.G, &was_trial);
        if (was_trial) {
            // a shrink on trial did not certify: back to the margin that did (this call is repeated there)
            fprintf(stderr, "[medaka_amd] split scan: a margin of %d columns does not certify (junction states differ by %.3g): back to %d\n",
                    sp.G, m->last_split.max_delta, m->margin.cur);
            continue;
        }
        const int next = m->margin.cur;
        if (nx == MarginLearner::GIVE_UP) {
            m->split_disabled = true;
            m->split_backoff = m->split_backoff ? std::min<long>(2 * m->split_backoff, 4096) : 64;
            m->split_retry_in = m->split_backoff;
            if (m->split_backoff == 64)
                fprintf(stderr, "[medaka_amd] split scan: junction states still differ by %.3g at a margin of %d columns: sequential scans "
                                "for the next %ld calls, then another try (back-off doubling up to 4096 calls)\n",
                        m->last_split.max_delta, sp.G, m->split_backoff);
            break;
        }
        fprintf(stderr, "[medaka_amd] split scan: junction states differed by %.3g at a margin of %d columns: margin %d from now on\n",
                m->last_split.max_delta, sp.G, next);
    }
    if (first_attempt && pre && pre->valid && !pre->split && pre->precision == m->precision) {
        pre->valid = false;            // the sequential passes are what start_call enqueued: the caller's synchronize ends them
        m->early_used++;
        report_audits(m);
        return MDK_OK;
    }
    if ((rc = forget_pre())) return rc;
    rc = run_passes(m, x_dev, B, T, probs_dev, s, x_host, probs_host);
    report_audits(m);
    return rc;
}

// The first attempt of a call, enqueue only: what run_forward would launch for (x_dev, B, T) right now -- a split scan at the
// margin in use, or the sequential passes -- WITHOUT waiting for anything.  Not started (st->valid stays false; run_forward then
// does everything): timing on, a probe due, more than one pass, the exact kernels.  `prev`: the other context; where the two
// forwards cannot share the chip this one's kernels are ordered behind that one's (its result copies are not waited for).
static int start_call(mdk_gru *m, const float *x_dev, int B, int T, float *probs_dev, hipStream_t s, float *probs_host,
                      mdk_gru::Started *st, const Ctx *prev) {
    st->valid = false;
    if (m->timing || m->variant != MDK_VARIANT_MFMA) return MDK_OK;
    SplitPlan sp;
    int rc;
    // (the back-off of a model whose certificate was rejected at the largest margin counts calls in run_forward: a call that
    // would end it is left to run_forward)
    if (m->split_disabled && m->split_retry_in == 1) return MDK_OK;
    const bool split = plan_split(m, B, T, sp);
    if (split && split_probe_due(m, sp)) return MDK_OK;
    const size_t budget = m->max_rows_per_pass ? m->max_rows_per_pass : kMaxRowsPerPass;
    if (!split && (size_t)B * T > budget) return MDK_OK;
    int wgs = 256;
    if (!split) {
        PassPlan P;
        HostIO io;
        io.p_host = probs_host;
        if ((rc = plan_pass(m, B, T, &io, nullptr, P))) return rc;
        wgs = P.n_wg * P.D * m->opt_gpu_share;
    }
    // two forwards side by side only where both leave the other its CUs (sequential scans of the reference's batch sizes: 100 of
    // 256 CUs each); a recurrence that holds every CU tolerates nothing beside it (profiles/r4_experiments/README.md)
    m->wait_before_l1 = nullptr;
    if (prev && prev->kernels_done && prev->last_wgs > 0 && (split || wgs + prev->last_wgs > 256)) {
        // Stage overlap (option "stage_overlap"): this batch's LAYER 0 beside the previous batch's LAYER 1 -- a layer-0 work-group
        // (8 KB of LDS, a latency chain that leaves the matrix pipe idle two thirds of its step in half precision) fits on a CU
        // beside a fused layer-1 work-group; layers of the same kind still follow each other
        static const int env_so = getenv("MDK_STAGE_OVERLAP") ? atoi(getenv("MDK_STAGE_OVERLAP")) : -1;
        const int so = env_so >= 0 ? env_so : m->opt_stage_overlap;
        const bool stage = split && m->desc.num_layers == 2 && (so == 2 || (so == 1 && m->precision == MDK_PREC_FP16));
        if (stage) {
            HIP_TRY(hipStreamWaitEvent(s, prev->l0_done, 0));
            m->wait_before_l1 = prev->kernels_done;
        } else {
            HIP_TRY(hipStreamWaitEvent(s, prev->kernels_done, 0));
        }
    }
    st->split = split;
    st->precision = m->precision;
    if (split) {
        EvTimer none{m, s};
        st->sp = sp;
        if ((rc = split_enqueue(m, sp, x_dev, probs_dev, s, nullptr, probs_host, none, &st->need_gi))) return rc;
    } else {
        if ((rc = run_passes(m, x_dev, B, T, probs_dev, s, nullptr, probs_host))) return rc;
    }
    m->wait_before_l1 = nullptr;
    st->valid = true;
    return MDK_OK;
}
