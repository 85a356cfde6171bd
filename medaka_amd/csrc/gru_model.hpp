// The model object of the consensus GRU engine: margin learner, weights, the two contexts, create / destroy / options.
// Part of api.hip (one translation unit; included there, in this order: gru_model, gru_pass, gru_split, gru_entries).
#pragma once
// ---- the margin of the split scan, learned per model (scan_split.hpp).  Pure state machine, no
// device: also exported as mdk_margin_sim for the CPU property tests.
// The margins a model can learn: a ladder instead of doublings (a set that needs 192 should not pay for 256: 19 % of all
// columns against 25 %).  Margins outside the ladder (option "scan_split_margin") join it at the next rung.
static const int kMarginLadder[] = {64, 96, 128, 192, 256, 384, 512};
static int split_margin_up(int G) {
    for (int r : kMarginLadder) if (r > G) return r;
    return 2 * kSplitMarginMax;                      // above the ladder: the caller gives the model up
}
static int split_margin_down(int G, int floor_) {
    int best = 0;
    for (int r : kMarginLadder) if (r < G && r >= floor_) best = r;
    return best;                                     // 0: nothing smaller is allowed
}
struct MarginLearner {
    int cur = 0;          // margin in use (0: the option's starting margin)
    int floor_ = 0;       // no shrink below this: one rung above the largest margin a certificate was ever rejected at
    int quiet = 0;        // consecutive certified calls at the current margin whose differences sat at the noise floor
    int trial_back = 0;   // != 0: the current margin is a shrink on trial; a rejection returns to this one
    enum Next { RETRY = 0, GIVE_UP = 1 };
    void reset(bool forget_rejections) { cur = quiet = trial_back = 0; if (forget_rejections) floor_ = 0; }
    // a certified call at margin G; returns the margin a kept trial came from (0: none).  `adapt` = quiet calls before a smaller
    // margin is tried (0: never), `noise_floor` = largest junction difference that still counts as quiet
    int certified(int G, float worst, float noise_floor, int adapt) {
        const int was = trial_back;
        trial_back = 0;
        quiet = worst <= noise_floor ? quiet + 1 : 0;
        if (adapt > 0 && quiet >= adapt) {
            const int down = split_margin_down(G, floor_);
            quiet = 0;
            if (down) { trial_back = G; cur = down; }
        }
        return was;
    }
    // a rejected certificate at margin G: RETRY = run the call again at `cur` (a failed trial goes back, anything else one rung
    // up), GIVE_UP = nothing larger is left.  `*back` = 1 if this was a trial
    Next rejected(int G, int *back) {
        quiet = 0;
        floor_ = std::max(floor_, split_margin_up(G));          // never shrink to a rejected margin again
        *back = 0;
        if (trial_back) { cur = trial_back; trial_back = 0; *back = 1; return RETRY; }
        const int next = split_margin_up(G);
        if (next > kSplitMarginMax) return GIVE_UP;
        cur = next;
        return RETRY;
    }
};

// ------------------------------------------------------------------------------------------
// model object
struct LayerDev {
    int K = 0;                    // input width of this layer
    float *w_ih_t = nullptr;      // [D][K][384] fp32
    float *w_hh_t = nullptr;      // [D][128][384] fp32
    float *bias_gi = nullptr;     // [D][384]
    float *b_hn = nullptr;        // [D][128]
    half8 *whh_frag = nullptr;    // [D][8 waves][4 ks][3 gates][2 hi/lo][64 lanes]
    float *ones = nullptr;        // [D] = 1
    half8 *wx_frag = nullptr;     // layer 0 only: fused input projection B-fragments [D][8][3][2][64]
    float x_scale = 0.f;          // layer 0 only: sx (0 = fusion unavailable)
    float *up_scale_rec = nullptr;   // [D] = 1/inv_scale_rec
    half8 *wih_frag = nullptr;    // [D][8 waves][K/32][3 gates][2][64] (layers >= 1)
    float *inv_scale_rec = nullptr;  // [D]
    float *inv_scale_gi = nullptr;   // [D]
};

// Everything ONE call in flight owns: workspace, flags, streams, event pools.  A model has two of these (mdk_gru below): the
// staged entry (mdk_gru_forward_pipelined) enqueues the NEXT batch's forward into the one that is idle while the caller still
// waits for the current batch's last result chunks -- every function of this file reaches these fields through `m->`, and
// switching contexts is a swap of this base object (swap_ctx), not a change of the code that uses them.
struct Ctx {
    float *lpart = nullptr;     // partial logits of the fused head [D][n_tiles][T][8][5]
    half8 *xfrag = nullptr;     // packed layer-0 input fragments
    size_t xfrag_cap = 0;
    int *oor_flag = nullptr;    // device flag: layer-0 input out of fp16 range -> unfused path
    int *oor_host = nullptr;    // page-locked copy (host-checked fallback: calls that did not allocate gi)
    size_t gi_rows = 0;         // rows gi is allocated for (0: not yet -- the throughput regime never touches it)
    // workspace (grown on demand)
    float *gi = nullptr;
    float *act[2] = {nullptr, nullptr};
    size_t ws_rows = 0;
    float *p_dev = nullptr;     // the probabilities on the device (host entries)
    size_t p_cap = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;              // projection GEMM of layer 1 under the tail of layer 0
    hipStream_t copy_in = nullptr;           // host path: time slabs of x, host -> device, ahead of the layer-0 recurrence
    hipStream_t copy_out = nullptr;          // host path: finished probability columns, device -> host
    hipStream_t copy_out2 = nullptr;         // split host path: every other chunk copy (two DMA engines side by side)
    std::vector<hipEvent_t> ov_ev;           // event pool of one forward pass (no timing)
    size_t ov_next = 0;
    float *gi2 = nullptr;                    // its own gi buffer (layer 0's fallback may still read gi)
    size_t gi2_rows = 0;
    float *xv = nullptr;                     // the virtual batch
    size_t xv_cap = 0;
    unsigned *split_flag = nullptr;          // device: bits of the largest junction difference per certificate point
    unsigned *split_host = nullptr;          // page-locked copy of split_flag
    mdk_gru_timing last{};
    std::vector<hipEvent_t> ev;
    hipEvent_t kernels_done = nullptr;       // behind the last KERNEL of the pass(es) this context enqueued last (its result copies may still run)
    bool shares_copy_streams = false;        // copy_in / copy_out / copy_out2 belong to the other context (init_ctx)
    hipEvent_t l0_done = nullptr;            // behind layer 0 of that pass
    hipEvent_t wait_before_l1 = nullptr;     // this pass: layers >= 1 start behind this event of the OTHER context (stage overlap, start_call)
    int last_wgs = 0;                        // recurrence work-groups (x directions x gpu_share) of that pass: 0 = nothing enqueued yet
};

struct mdk_gru : Ctx {
    Ctx other;                  // the second context (streams and buffers created on first use: init_ctx)
    mdk_gru_desc desc{};
    int device = 0;
    int D = 2;
    int precision = MDK_PREC_FP32;
    int variant = MDK_VARIANT_MFMA;
    int opt_tile_windows = 0;   // 0 auto, 4, 8
    int opt_ablate = 0;         // timing-only ablation mask of the recurrence kernel
    size_t max_rows_per_pass = 0;   // 0 = kMaxRowsPerPass
    int opt_fuse_l0 = 1;        // fuse the layer-0 input projection into the recurrence
    int opt_final_head = 1;     // fused head: the scan's second half writes probabilities itself (rec_fused.hpp HEAD = 2); 0: k_head_combine
    int opt_fuse_head = 1;      // last layer with a fused projection: Linear(D*128 -> 5) inside the recurrence kernel too (rec_fused.hpp HEAD)
    half8 *wlin_frag = nullptr; // [D][4 ksteps][2 hi/lo][64 lanes] B-fragments of linear.weight (classes padded to 16 columns)
    float lin_inv_scale = 1.f;  // 1 / (kActScale * their operand scale)
    int opt_fuse_proj = 1;      // layers >= 1: projection fused into the recurrence (rec_fused.hpp): 0 off, 1 when the call fills the chip, 2 always
    bool oor_seen = false;      // an input left the fp16 range once: gi stays allocated and the fallback decides on the device again
    std::vector<LayerDev> layers;
    float *lin_w = nullptr, *lin_b = nullptr;
    // host-API staging
    float *x_dev = nullptr;     // x of a host call (the staged entry reads its staging slot instead)
    size_t x_cap = 0;
    unsigned char *aux_dev = nullptr;   // raw counts + depth in, decoded classes + probabilities out
    size_t aux_cap = 0;
    int opt_overlap = 1;
    int opt_deferred_store = 1;              // recurrence: HBM store of h_t from inside step t+1 (rec_mfma.hpp DS)
    int opt_gpu_share = 1;                   // processes sharing this GPU (launch.py --procs-per-gpu): divides the CU budgets below
    int opt_stream_host = 1;                 // host path: x in / probabilities out in time slabs under the recurrences
    // split scan (scan_split.hpp)
    int opt_scan_split = 1;                  // 0 off, 1 auto, n >= 2: n chunks per window whenever the shape allows it
    int opt_split_margin = 128;              // G: columns of warm-up on either side of a chunk (where the model starts)
    MarginLearner margin;                    // the margin in use, LEARNED per model: one rung up the ladder 64 .. 512 on a rejected
                                             // certificate, one rung down after `opt_split_adapt` certified calls at the noise floor
    int opt_split_adapt = 8;                 // certified calls at the noise floor before a smaller margin is tried (0: never shrink)
    // half precision: a margin is used only after a call CERTIFIED AT IT IN FP32-PARITY MODE (a "probe": the same call, run once
    // more with the hi/lo operands, threshold 2^-18, result discarded) -- half mode's own certificate compares fp16 images of h
    // (threshold 2^-10) and cannot see an un-merged state below ~1e-3; see run_forward
    int opt_split_probe = 1;                 // 0: half mode trusts its own certificate (round 5's behaviour)
    std::vector<int> probed_ok;              // margins a probe certified
    long probes_done = 0;
    float probe_last_delta = 0.f;
    bool split_disabled = false;             // a certificate failed at the largest margin (or an audit failed): sequential scans (auto mode)
    long split_retry_in = 0;                 // ... for this many calls; then one more try at the largest margin (0: for good -- failed audits)
    long split_backoff = 0;                  // the last back-off (doubles per rejection at the largest margin: 64 .. 4096 calls)
    mdk_gru_split last_split{};
    int opt_split_audit = 1;                 // 0 never, 1 the first certified call of every margin, 2 every certified call
    int split_audited_key = 0;               // margin | precision << 16 whose first certified call has been audited (0 = none yet)
    float *audit = nullptr;                  // the sequential scan's probabilities of an audited call
    size_t audit_cap = 0;
    // early hand-over of the next batch (mdk_gru_stage_input): its host -> device copy runs on `stage_stream` while the
    // caller's thread is still inside the forward of the previous one
    struct StageSlot { unsigned long long token = 0; bool busy = false; float *dev = nullptr; size_t cap = 0; int B = 0, T = 0; hipEvent_t ready = nullptr; };
    StageSlot stage[10];     // the reference's loader runs up to 8 batches ahead of the model (prediction.py:229, batch_cache_size): those, the
                             // one the forward is reading and the one being filled; buffers are allocated on first use, sized to the batch
    unsigned long long stage_next_token = 1;
    hipStream_t stage_stream = nullptr;
    std::mutex stage_mu;
    long staged_used = 0;
    int stage_unredeemed = 0;                // slots overwritten in a row whose token nobody had redeemed
    int stage_pause = 0;                     // > 0: the next this many hand-overs are skipped (nobody was redeeming them)
    // standing audit: every `opt_split_audit_every`-th certified call is ALSO run as the sequential scan (run_forward)
    int opt_split_audit_every = 256;
    long split_calls_since_audit = 0;
    long audits_done = 0;
    int audit_failures = 0;
    float audit_worst = 0.f;
    // timing
    bool timing = false;
    // the NEXT batch's forward, enqueued ahead of its call (mdk_gru_forward_pipelined): lives in `other` while valid
    struct Started {
        bool valid = false;
        bool split = false;                  // enqueued as a split scan (its certificate is still unread) / as sequential passes
        SplitPlan sp{};
        bool need_gi = false;
        int precision = 0;
        std::vector<hipEvent_t> out_done;    // (unused after the enqueue: the stream waits for them itself)
    };
    struct Pending { Started st; unsigned long long token = 0; StageSlot *slot = nullptr; int B = 0, T = 0; float *probs_host = nullptr; } pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> dbg_spans;   // MDK_EARLY_DEBUG: first .. last kernel of every split forward (timing events)
    int opt_tail_blit = 1;                   // mdk_gru_forward, page-locked result buffer: the last result chunks of a split call leave by kernel
    float *tail_host = nullptr, *tail_dev = nullptr;   // ... the buffer of the call in progress and the device's view of it (else null)
    int opt_early_start = 1;                 // 0: the staged entry never starts the next batch's forward ahead of its call
    int opt_stage_overlap = 2;               // a batch started ahead: its layer 0 beside the previous batch's layer 1 (0 off, 1 half precision, 2 both)
    long early_started = 0, early_used = 0, early_dropped = 0;
};


// ---- the two contexts of a model
static void swap_ctx(mdk_gru *m) { std::swap(static_cast<Ctx &>(*m), m->other); }

// streams and flags of the CURRENT context (create: the first; the second on its first use -- swap, init, swap back)
static int init_ctx(mdk_gru *m) {
    if (m->stream) return MDK_OK;
    // The SECOND context's main stream gets a priority of its own: HIP hands its streams out over a small pool of hardware queues
    // PER PRIORITY (GPU_MAX_HW_QUEUES = 4 by default) and streams that land on one queue are serialised -- a second main stream
    // from the same pool could share the first one's queue, and a batch started ahead could then never run its layer 0 beside the
    // previous batch's layer 1 ("stage_overlap").  As the only high-priority stream of the model it has a queue to itself; the
    // first context's streams are created exactly as they were before there was a second (a high-priority main stream there
    // costs the cold host-to-host call 0.15 ms: profiles/r6_experiments/README.md).
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // (the copy streams are shared by the two contexts -- `other` holds them already when the second one is initialised: their
    // work is DMA behind events, in the order the forwards were enqueued, and every stream less is one hardware queue less to alias)
    const bool share = m->other.copy_in != nullptr;
    if (share) { m->copy_in = m->other.copy_in; m->copy_out = m->other.copy_out; m->copy_out2 = m->other.copy_out2; m->shares_copy_streams = true; }
    // (a runtime without stream priorities: an ordinary stream -- stage overlap may then find the two main streams on one queue)
    if (share && hipStreamCreateWithPriority(&m->stream, hipStreamNonBlocking, prio_hi) != hipSuccess) { (void)hipGetLastError(); m->stream = nullptr; }
    if ((!m->stream && hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) ||
        hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess ||
        (!share && (hipStreamCreateWithFlags(&m->copy_in, hipStreamNonBlocking) != hipSuccess ||
                    hipStreamCreateWithFlags(&m->copy_out, hipStreamNonBlocking) != hipSuccess ||
                    hipStreamCreateWithFlags(&m->copy_out2, hipStreamNonBlocking) != hipSuccess)) ||
        hipEventCreateWithFlags(&m->kernels_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->l0_done, hipEventDisableTiming) != hipSuccess)
        return fail(MDK_ERR_DEVICE, "hipStreamCreate failed");
    if (hipMalloc((void **)&m->oor_flag, 8192) != hipSuccess) return fail(MDK_ERR_OOM, "hipMalloc failed");
    (void)hipMemset(m->oor_flag, 0, 8192);
    return MDK_OK;
}

static void free_ctx(Ctx &c) {
    free_dev(c.lpart); free_dev(c.gi); free_dev(c.act[0]); free_dev(c.act[1]); free_dev(c.gi2); free_dev(c.p_dev);
    free_dev(c.xfrag); free_dev(c.oor_flag); free_dev(c.xv); free_dev(c.split_flag);
    if (c.split_host) (void)hipHostFree(c.split_host);
    if (c.oor_host) (void)hipHostFree(c.oor_host);
    for (auto e : c.ev) (void)hipEventDestroy(e);
    for (auto e : c.ov_ev) (void)hipEventDestroy(e);
    if (c.kernels_done) (void)hipEventDestroy(c.kernels_done);
    if (c.l0_done) (void)hipEventDestroy(c.l0_done);
    if (c.shares_copy_streams) c.copy_in = c.copy_out = c.copy_out2 = nullptr;       // (the other context's)
    for (hipStream_t st : {c.stream, c.side, c.copy_in, c.copy_out, c.copy_out2})
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    c = Ctx{};
}

static void drop_pending(mdk_gru *m);

extern "C" void mdk_gru_destroy(mdk_gru *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    drop_pending(m);
    if (getenv("MDK_EARLY_DEBUG") && m->dbg_spans.size() > 12) {
        // the last forwards of the model: duration of each, idle time between one's last kernel and the next one's first
        (void)hipDeviceSynchronize();
        const size_t n = m->dbg_spans.size(), lo = n - 12;
        fprintf(stderr, "[medaka_amd] last split forwards (ms) / gap to the next (ms):");
        for (size_t i = lo; i < n; ++i) {
            float d = 0.f, g = 0.f;
            (void)hipEventElapsedTime(&d, m->dbg_spans[i].first, m->dbg_spans[i].second);
            if (i + 1 < n) (void)hipEventElapsedTime(&g, m->dbg_spans[i].second, m->dbg_spans[i + 1].first);
            fprintf(stderr, " %.3f/%.3f", d, g);
        }
        fprintf(stderr, "\n");
    }
    for (auto &pr : m->dbg_spans) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (getenv("MDK_EARLY_DEBUG"))
        fprintf(stderr, "[medaka_amd] staged calls %ld: forwards started ahead %ld, taken over %ld, dropped %ld\n", m->staged_used, m->early_started,
                m->early_used, m->early_dropped);
    for (auto &L : m->layers) {
        free_dev(L.w_ih_t); free_dev(L.w_hh_t); free_dev(L.bias_gi); free_dev(L.b_hn);
        free_dev(L.whh_frag); free_dev(L.ones); free_dev(L.wx_frag); free_dev(L.up_scale_rec); free_dev(L.wih_frag); free_dev(L.inv_scale_rec); free_dev(L.inv_scale_gi);
    }
    free_dev(m->wlin_frag);
    free_dev(m->lin_w); free_dev(m->lin_b);
    (void)hipDeviceSynchronize();
    if (m->shares_copy_streams) { free_ctx(static_cast<Ctx &>(*m)); free_ctx(m->other); }
    else { free_ctx(m->other); free_ctx(static_cast<Ctx &>(*m)); }
    free_dev(m->aux_dev); free_dev(m->x_dev); free_dev(m->audit);
    if (m->stage_stream) { (void)hipStreamSynchronize(m->stage_stream); (void)hipStreamDestroy(m->stage_stream); }
    for (auto &sl : m->stage) { free_dev(sl.dev); if (sl.ready) (void)hipEventDestroy(sl.ready); }
    delete m;
}

// The classifier: Linear(D * 128 -> 5) as fp32 (k_head_tiled, the exact kernels) and, for the head fused into the last
// layer's kernel (rec_fused.hpp HEAD), as fp16 hi/lo B-fragments per direction: k = hidden unit in the A image's order
// (slot (ks, lane-group gq, i) = unit 32 ks + 8 gq + i), column n = class (columns 5..15 zero).
static int upload_classifier(mdk_gru *m, const float *lin_w, const float *lin_b) {
    const int D = m->D, H = kH, C = m->desc.num_classes;
    int rc;
    std::vector<float> lw(lin_w, lin_w + (size_t)C * D * H);
    std::vector<float> lb(lin_b, lin_b + C);
    if ((rc = upload(&m->lin_w, lw))) return rc;
    if ((rc = upload(&m->lin_b, lb))) return rc;
    const float swl = pick_scale(lw.data(), lw.size());
    m->lin_inv_scale = 1.0f / (kActScale * swl);
    std::vector<half8> wl((size_t)D * 4 * 2 * 64);
    for (int d = 0; d < D; ++d)
        for (int ks = 0; ks < 4; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 15, gq = lane >> 4;
                half8 hi, lo;
                for (int i = 0; i < 8; ++i) {
                    const int u = 32 * ks + 8 * gq + i;
                    _Float16 a = (_Float16)0.f, b = (_Float16)0.f;
                    if (n < C) split_host(lw[(size_t)n * D * H + (size_t)d * H + u] * swl, a, b);
                    hi[i] = a; lo[i] = b;
                }
                wl[((size_t)(d * 4 + ks) * 2 + 0) * 64 + lane] = hi;
                wl[((size_t)(d * 4 + ks) * 2 + 1) * 64 + lane] = lo;
            }
    return upload(&m->wlin_frag, wl);
}

extern "C" int mdk_gru_create(const mdk_gru_desc *desc, const float *const *weights,
                              int n_weights, int device, mdk_gru **out) {
    if (!desc || !weights || !out) return fail(MDK_ERR_ARG, "null argument");
    *out = nullptr;
    const int I = desc->num_features, H = desc->hidden, L = desc->num_layers;
    const int D = desc->bidirectional ? 2 : 1, C = desc->num_classes;
    if (H != kH) return fail(MDK_ERR_ARG, "unsupported gru_size %d (engine supports 128)", H);
    if (L < 1 || L > 4) return fail(MDK_ERR_ARG, "unsupported num_layers %d (1..4)", L);
    if (I < 1 || I > 256) return fail(MDK_ERR_ARG, "unsupported num_features %d (1..256)", I);
    if (C != 5) return fail(MDK_ERR_ARG, "unsupported num_classes %d (reference Linear is fixed at 5)", C);
    if (n_weights != 4 * L * D + 2) return fail(MDK_ERR_ARG, "expected %d weight tensors, got %d", 4 * L * D + 2, n_weights);
    for (int i = 0; i < n_weights; ++i)
        if (!weights[i]) return fail(MDK_ERR_ARG, "weight tensor %d is null", i);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(MDK_ERR_DEVICE, "device %d not available (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    mdk_gru *m = new mdk_gru();
    m->desc = *desc;
    m->device = device;
    m->D = D;
    m->layers.resize(L);
    // process-wide defaults of the split scan (the options of the same names override them per model)
    if (const char *e = getenv("MDK_SCAN_SPLIT")) m->opt_scan_split = std::min(std::max(atoi(e), 0), kMaxSplit);
    if (const char *e = getenv("MDK_SCAN_SPLIT_ADAPT")) m->opt_split_adapt = std::max(atoi(e), 0);
    if (const char *e = getenv("MDK_SCAN_SPLIT_PROBE")) m->opt_split_probe = atoi(e) ? 1 : 0;
    if (const char *e = getenv("MDK_EARLY_START")) m->opt_early_start = atoi(e) ? 1 : 0;
    if (const char *e = getenv("MDK_TAIL_BLIT")) m->opt_tail_blit = atoi(e) ? 1 : 0;
    if (const char *e = getenv("MDK_SCAN_SPLIT_MARGIN")) {
        const int g = atoi(e);
        if (g >= 16 && g <= 4096 && g % 8 == 0) m->opt_split_margin = g;
    }
    int rc = MDK_OK;
    auto bail = [&](int code) { mdk_gru_destroy(m); return code; };
    if ((rc = init_ctx(m))) return bail(rc);

    for (int l = 0; l < L; ++l) {
        LayerDev &Ld = m->layers[l];
        const int K = (l == 0) ? I : D * H;
        Ld.K = K;
        std::vector<float> w_ih_t((size_t)D * K * kG), w_hh_t((size_t)D * kH * kG);
        std::vector<float> bias_gi((size_t)D * kG), b_hn((size_t)D * kH);
        std::vector<float> inv_rec(D), inv_gi(D);
        std::vector<half8> whh_frag((size_t)D * 8 * 4 * 3 * 2 * 64);
        std::vector<float> ones(D, 1.0f), up_rec(D);
        const int KS = (K % 32 == 0) ? K / 32 : 0;
        std::vector<half8> wih_frag;
        if (l > 0) wih_frag.resize((size_t)D * 4 * KS * 6 * 2 * 64);
        for (int d = 0; d < D; ++d) {
            const float *w_ih = weights[4 * (l * D + d) + 0];
            const float *w_hh = weights[4 * (l * D + d) + 1];
            const float *b_ih = weights[4 * (l * D + d) + 2];
            const float *b_hh = weights[4 * (l * D + d) + 3];
            for (int j = 0; j < kG; ++j) {
                for (int k = 0; k < K; ++k) w_ih_t[((size_t)d * K + k) * kG + j] = w_ih[(size_t)j * K + k];
                for (int k = 0; k < kH; ++k) w_hh_t[((size_t)d * kH + k) * kG + j] = w_hh[(size_t)j * kH + k];
                bias_gi[(size_t)d * kG + j] = b_ih[j] + (j < 2 * kH ? b_hh[j] : 0.0f);
            }
            for (int j = 0; j < kH; ++j) b_hn[(size_t)d * kH + j] = b_hh[2 * kH + j];
            // recurrent B-fragments (rec_mfma.hpp): wave w8 owns units 16*w8..+15, column n = lane&15
            const float sw = pick_scale(w_hh, (size_t)kG * kH);
            inv_rec[d] = 1.0f / (kActScale * sw);
            up_rec[d] = kActScale * sw;
            for (int w8 = 0; w8 < 8; ++w8)
                for (int ks = 0; ks < 4; ++ks)
                    for (int gate = 0; gate < 3; ++gate)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int j = gate * kH + 16 * w8 + (lane & 15);
                            const int gq = lane >> 4;
                            half8 hi, lo;
                            for (int i = 0; i < 8; ++i) {
                                _Float16 a, b;
                                split_host(w_hh[(size_t)j * kH + 32 * ks + 8 * gq + i] * sw, a, b);
                                hi[i] = a; lo[i] = b;
                            }
                            const size_t base = ((((size_t)(d * 8 + w8) * 4 + ks) * 3 + gate) * 2) * 64 + lane;
                            whh_frag[base] = hi;
                            whh_frag[base + 64] = lo;
                        }
            if (l > 0) {
                const float swi = pick_scale(w_ih, (size_t)kG * K);
                inv_gi[d] = 1.0f / (kActScale * swi);
                for (int w8 = 0; w8 < 8; ++w8)
                    for (int ks = 0; ks < KS; ++ks)
                        for (int nt = 0; nt < 3; ++nt)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int col = nt * kH + 16 * w8 + (lane & 15);   // gate nt, unit
                                const int gq = lane >> 4;
                                half8 hi, lo;
                                for (int i = 0; i < 8; ++i) {
                                    const int k = 32 * ks + 8 * gq + i;
                                    _Float16 a, b;
                                    split_host(w_ih[(size_t)col * K + k] * swi, a, b);
                                    hi[i] = a; lo[i] = b;
                                }
                                const size_t base = (((((size_t)(d * 8 + w8)) * KS + ks) * 3 + nt) * 2) * 64 + lane;
                                wih_frag[base] = hi;
                                wih_frag[base + 64] = lo;
                            }
            } else {
                inv_gi[d] = 1.0f;
            }
        }
        if ((rc = upload(&Ld.w_ih_t, w_ih_t))) return bail(rc);
        if ((rc = upload(&Ld.w_hh_t, w_hh_t))) return bail(rc);
        if ((rc = upload(&Ld.bias_gi, bias_gi))) return bail(rc);
        if ((rc = upload(&Ld.b_hn, b_hn))) return bail(rc);
        if ((rc = upload(&Ld.whh_frag, whh_frag))) return bail(rc);
        if ((rc = upload(&Ld.ones, ones))) return bail(rc);
        if ((rc = upload(&Ld.up_scale_rec, up_rec))) return bail(rc);
        if (l > 0 && (rc = upload(&Ld.wih_frag, wih_frag))) return bail(rc);
        if ((rc = upload(&Ld.inv_scale_rec, inv_rec))) return bail(rc);
        if ((rc = upload(&Ld.inv_scale_gi, inv_gi))) return bail(rc);
        if (l == 0 && K + 1 <= 8 * (kXfragLanes / 16)) {      // features + the bias row inside the packed block's k-slots
            // fused layer-0 projection: one sx for all directions (the packed x is shared),
            // per-direction W_ih scale swx = S_d / sx
            float sx = 16.0f;
            for (int d = 0; d < D; ++d) {
                const float *w_ih = weights[4 * d + 0];
                float mx = 0.f;
                for (size_t i = 0; i < (size_t)kG * K; ++i) mx = std::max(mx, std::fabs(w_ih[i]));
                for (int j = 0; j < kG; ++j) mx = std::max(mx, std::fabs(bias_gi[(size_t)d * kG + j]));
                const float need = up_rec[d] * mx / 32768.0f;   // sx >= S * max / 2^15
                while (sx < need) sx *= 2.0f;
            }
            if (sx <= 8192.0f) {
                std::vector<half8> wx((size_t)D * 8 * 3 * 2 * 64);
                for (int d = 0; d < D; ++d) {
                    const float *w_ih = weights[4 * d + 0];
                    const float swx = up_rec[d] / sx;
                    for (int w8 = 0; w8 < 8; ++w8)
                        for (int gate = 0; gate < 3; ++gate)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int j = gate * kH + 16 * w8 + (lane & 15);
                                const int gq = lane >> 4;
                                half8 hi, lo;
                                for (int i = 0; i < 8; ++i) {
                                    const int f = 8 * gq + i;
                                    float v = 0.f;
                                    if (f < K) v = w_ih[(size_t)j * K + f] * swx;
                                    else if (f == K) v = bias_gi[(size_t)d * kG + j] * swx;
                                    _Float16 a, b;
                                    split_host(v, a, b);
                                    hi[i] = a; lo[i] = b;
                                }
                                const size_t base = ((((size_t)(d * 8 + w8)) * 3 + gate) * 2) * 64 + lane;
                                wx[base] = hi;
                                wx[base + 64] = lo;
                            }
                }
                if ((rc = upload(&Ld.wx_frag, wx))) return bail(rc);
                Ld.x_scale = sx;
            }
        }
    }
    if ((rc = upload_classifier(m, weights[4 * L * D], weights[4 * L * D + 1]))) return bail(rc);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gi_gemm<8, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kGemmMT * 8 * 64 * 16));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_fused<8, 0, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds_bytes(8)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_fused<8, 1, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds_bytes(8)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_fused<8, 2, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds_bytes(8)));
    if (const char *e = getenv("MDK_FUSE_HEAD")) m->opt_fuse_head = atoi(e) ? 1 : 0;
    if (const char *e = getenv("MDK_FINAL_HEAD")) m->opt_final_head = atoi(e) ? 1 : 0;
    if (const char *e = getenv("MDK_FUSE_PROJ")) m->opt_fuse_proj = std::min(std::max(atoi(e), 0), 2);

    *out = m;
    return MDK_OK;
}

extern "C" int mdk_gru_set_precision(mdk_gru *m, int precision) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if (precision != MDK_PREC_FP32 && precision != MDK_PREC_FP16) return fail(MDK_ERR_ARG, "bad precision %d", precision);
    if (precision != m->precision) { (void)hipSetDevice(m->device); drop_pending(m); }
    m->precision = precision;
    return MDK_OK;
}
extern "C" int mdk_gru_set_variant(mdk_gru *m, int variant) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if (variant != MDK_VARIANT_MFMA && variant != MDK_VARIANT_EXACT) return fail(MDK_ERR_ARG, "bad variant %d", variant);
    if (variant != m->variant) { (void)hipSetDevice(m->device); drop_pending(m); }
    m->variant = variant;
    return MDK_OK;
}
extern "C" int mdk_gru_set_normalise(mdk_gru *m, int normalise) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if ((normalise ? 1 : 0) != m->desc.normalise) { (void)hipSetDevice(m->device); drop_pending(m); }
    m->desc.normalise = normalise ? 1 : 0;
    return MDK_OK;
}
extern "C" int mdk_gru_set_option(mdk_gru *m, const char *key, int value) {
    if (!m || !key) return fail(MDK_ERR_ARG, "null argument");
    (void)hipSetDevice(m->device);
    drop_pending(m);                 // (a batch started ahead was planned under the old options)
    if (!strcmp(key, "tail_blit")) {
        m->opt_tail_blit = value ? 1 : 0;
    } else if (!strcmp(key, "early_start")) {
        m->opt_early_start = value ? 1 : 0;
    } else if (!strcmp(key, "stage_overlap")) {
        if (value < 0 || value > 2) return fail(MDK_ERR_ARG, "stage_overlap must be 0, 1 (half precision) or 2 (both precisions)");
        m->opt_stage_overlap = value;
    } else if (!strcmp(key, "rec_windows_per_tile")) {
        if (value != 0 && value != 4 && value != 8 && value != 16)
            return fail(MDK_ERR_ARG, "rec_windows_per_tile must be 0, 4, 8 or 16 (16: half precision only)");
        m->opt_tile_windows = value;
    } else if (!strcmp(key, "max_rows_per_pass")) {
        if (value < 0) return fail(MDK_ERR_ARG, "max_rows_per_pass must be >= 0 (0 = default)");
        m->max_rows_per_pass = (size_t)value;
#ifdef MDK_DEBUG_HOOKS
    } else if (!strcmp(key, "ablate")) {
        m->opt_ablate = value;
#endif
    } else if (!strcmp(key, "fuse_l0")) {
        m->opt_fuse_l0 = value ? 1 : 0;
    } else if (!strcmp(key, "fuse_head")) {
        m->opt_fuse_head = value ? 1 : 0;
    } else if (!strcmp(key, "final_head")) {
        m->opt_final_head = value ? 1 : 0;
    } else if (!strcmp(key, "fuse_proj")) {
        if (value < 0 || value > 2) return fail(MDK_ERR_ARG, "fuse_proj must be 0 (off), 1 (auto) or 2 (always)");
        m->opt_fuse_proj = value;
    } else if (!strcmp(key, "overlap_gemm")) {
        m->opt_overlap = value < 0 ? 0 : (value > 2 ? 2 : value);   // 0 off, 1 auto, 2 force (experiments)
    } else if (!strcmp(key, "deferred_store")) {
        m->opt_deferred_store = value ? 1 : 0;
    } else if (!strcmp(key, "stream_host")) {
        m->opt_stream_host = value ? 1 : 0;
    } else if (!strcmp(key, "gpu_share")) {
        if (value < 1 || value > 8) return fail(MDK_ERR_ARG, "gpu_share must be 1..8");
        m->opt_gpu_share = value;
    } else if (!strcmp(key, "scan_split")) {
        if (value < 0 || value > kMaxSplit) return fail(MDK_ERR_ARG, "scan_split must be 0 (off), 1 (auto) or 2..%d chunks", kMaxSplit);
        m->opt_scan_split = value;
        m->split_disabled = false;           // setting the option re-arms a model that fell back
        m->split_retry_in = m->split_backoff = 0;
        m->margin.reset(true);
        m->probed_ok.clear();
    } else if (!strcmp(key, "scan_split_audit")) {
        if (value < 0 || value > 2) return fail(MDK_ERR_ARG, "scan_split_audit must be 0, 1 or 2");
        m->opt_split_audit = value;
    } else if (!strcmp(key, "scan_split_audit_every")) {
        if (value < 0) return fail(MDK_ERR_ARG, "scan_split_audit_every must be >= 0 (0 = only the first call of a margin)");
        m->opt_split_audit_every = value;
    } else if (!strcmp(key, "scan_split_adapt")) {
        if (value < 0) return fail(MDK_ERR_ARG, "scan_split_adapt must be >= 0 (certified calls at the noise floor before a smaller margin is tried; 0 = never)");
        m->opt_split_adapt = value;
        m->margin.quiet = 0;
    } else if (!strcmp(key, "scan_split_probe")) {
        m->opt_split_probe = value ? 1 : 0;
    } else if (!strcmp(key, "scan_split_margin")) {
        if (value < 16 || value > 4096 || value % 8) return fail(MDK_ERR_ARG, "scan_split_margin must be a multiple of 8 in 16..4096");
        m->opt_split_margin = value;
        m->margin.reset(false);      // (what the certificates rejected so far stays learned: "scan_split" re-arms)
        m->split_disabled = false;
        m->split_retry_in = m->split_backoff = 0;
    } else {
        return fail(MDK_ERR_ARG, "unknown option '%s'", key);
    }
    return MDK_OK;
}
#ifdef MDK_DEBUG_HOOKS
// debug: per-phase cycle counters written by the ablate=64 build of the recurrence kernel
extern "C" int mdk_gru_debug_read(mdk_gru *m, unsigned long long *dst, int n) {
    if (!m || !dst || n < 0 || n > 768) return fail(MDK_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(dst, reinterpret_cast<char *>(m->oor_flag) + 64, (size_t)n * 8, hipMemcpyDeviceToHost));
    return MDK_OK;
}
#endif
extern "C" int mdk_gru_enable_timing(mdk_gru *m, int on) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    m->timing = on != 0;
    return MDK_OK;
}
extern "C" int mdk_gru_get_timing(mdk_gru *m, mdk_gru_timing *out) {
    if (!m || !out) return fail(MDK_ERR_ARG, "null argument");
    *out = m->last;
    return MDK_OK;
}
extern "C" int mdk_gru_get_split(mdk_gru *m, mdk_gru_split *out) {
    if (!m || !out) return fail(MDK_ERR_ARG, "null argument");
    *out = m->last_split;
    return MDK_OK;
}
extern "C" int mdk_gru_device(const mdk_gru *m) { return m ? m->device : -1; }
