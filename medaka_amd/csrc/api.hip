// C ABI of the MI355X consensus-inference engine (include/medaka_amd.h).
// Host side: weight packing, workspace management, launch sequencing, hipEvent timing.
// No CPU fallback exists here: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/medaka_amd.h"
#include "common.hpp"
#include "exact.hpp"
#include "gi_proj.hpp"
#include "layout.hpp"
#include "head.hpp"
#include "rec_mfma.hpp"
#include "rec_fused.hpp"
#include "scan_split.hpp"

using namespace mdk;

#ifndef MDK_PF
#define MDK_PF 5   // gi / x prefetch ring depth (effective look-ahead PF-1 steps)
#endif


#include "host_common.hpp"

extern "C" const char *mdk_last_error(void) { return g_mdk_err.c_str(); }
extern "C" const char *mdk_version(void) { return "medaka_amd 0.1 (gfx950)"; }

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
    if (const char *e = getenv("MDK_STREAM_PRIO")) { if (!atoi(e)) prio_hi = prio_lo = 0; }
    // (the copy streams are shared by the two contexts -- `other` holds them already when the second one is initialised: their
    // work is DMA behind events, in the order the forwards were enqueued, and every stream less is one hardware queue less to alias)
    const bool share = m->other.copy_in != nullptr;
    if (share) { m->copy_in = m->other.copy_in; m->copy_out = m->other.copy_out; m->copy_out2 = m->other.copy_out2; m->shares_copy_streams = true; }
    if ((share ? hipStreamCreateWithPriority(&m->stream, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking)) != hipSuccess ||
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
    if (!strcmp(key, "early_start")) {
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
    struct OutRange { hipEvent_t ready; int t0, nt; };
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
        if (lo0 + len == hi0) out_ranges.push_back({ev, lo0, 2 * len});
        else { out_ranges.push_back({ev, lo0, len}); out_ranges.push_back({ev, hi0, len}); }
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
                if (lo0 + len == hi0) out_ranges.push_back({hv, lo0, 2 * len});
                else { out_ranges.push_back({hv, lo0, len}); out_ranges.push_back({hv, hi0, len}); }
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
        int n_copy = 0;
        static const bool one_copy_stream = getenv("MDK_ONE_COPY_STREAM") && atoi(getenv("MDK_ONE_COPY_STREAM"));
        for (const OutRange &r : out_ranges) {
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
                                         ((n_copy++ & 1) && !one_copy_stream) ? m->copy_out2 : m->copy_out));
            }
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

extern "C" int mdk_gru_forward_dev(mdk_gru *m, const float *x_dev, int B, int T, float *probs_dev,
                                   void *stream) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if (B < 0 || T < 0) return fail(MDK_ERR_ARG, "negative shape B=%d T=%d", B, T);
    if (B == 0 || T == 0) { memset(&m->last, 0, sizeof(m->last)); m->last.n_layers = m->desc.num_layers; return MDK_OK; }
    if (!x_dev || !probs_dev) return fail(MDK_ERR_ARG, "null buffer");
    HIP_TRY(hipSetDevice(m->device));
    drop_pending(m);
    // NULL = the legacy default stream, as for any HIP call
    return run_forward(m, x_dev, B, T, probs_dev, (hipStream_t)stream, nullptr, nullptr);
}

static int ensure_staging(mdk_gru *m, size_t nx, size_t np) {
    if (nx > m->x_cap) {
        free_dev(m->x_dev); m->x_dev = nullptr; m->x_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->x_dev, nx * sizeof(float)));
        m->x_cap = nx;
    }
    if (np > m->p_cap) {
        free_dev(m->p_dev); m->p_dev = nullptr; m->p_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->p_dev, np * sizeof(float)));
        m->p_cap = np;
    }
    return MDK_OK;
}

// ---- early hand-over of a batch (the engine's Batch.collate calls this from the reference's Batcher thread) --------
extern "C" int mdk_gru_stage_input(mdk_gru *m, const float *x_host, int B, int T, unsigned long long *token) {
    if (!m || !token) return fail(MDK_ERR_ARG, "null argument");
    *token = 0;
    if (B <= 0 || T <= 0 || !x_host) return fail(MDK_ERR_ARG, "bad batch B=%d T=%d", B, T);
    HIP_TRY(hipSetDevice(m->device));
    // Pick a slot under the lock, fill it outside: the (re)allocation of its buffer and the wait for an unredeemed copy
    // synchronise the device, and mdk_gru_forward_staged -- the caller's main thread -- needs the same lock.
    mdk_gru::StageSlot *sl = nullptr;
    {
        std::lock_guard<std::mutex> lock(m->stage_mu);
        if (!m->stage_stream) HIP_TRY(hipStreamCreateWithFlags(&m->stage_stream, hipStreamNonBlocking));
        // nobody is redeeming the tokens (another model took the batches, or the caller uses the counts / decoded entries):
        // every copy would cross PCIe for nothing -- pause, and look again later
        if (m->stage_pause > 0) { m->stage_pause--; return MDK_OK; }
        // a free slot, else the one staged longest ago (a token nobody redeemed in time simply stops being valid); never the
        // slot a forward is reading or another stager is filling
        for (auto &c : m->stage)
            if (!c.busy && (!sl || c.token < sl->token)) sl = &c;
        if (!sl) return fail(MDK_ERR_ARG, "no staging slot free");
        if (sl->token != 0 && ++m->stage_unredeemed >= 4) { m->stage_unredeemed = 0; m->stage_pause = 64; }
        sl->busy = true;
        sl->token = 0;
    }
    const size_t n = (size_t)B * T * m->desc.num_features;
    int rc = MDK_OK;
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == MDK_OK) rc = fail(e == hipErrorOutOfMemory ? MDK_ERR_OOM : MDK_ERR_DEVICE, "%s failed: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    if (sl->ready) hip_ok(hipEventSynchronize(sl->ready), "hipEventSynchronize");        // (an unredeemed copy into this slot may still be running)
    if (rc == MDK_OK && n > sl->cap) {
        free_dev(sl->dev); sl->dev = nullptr; sl->cap = 0;
        if (hip_ok(hipMalloc((void **)&sl->dev, n * sizeof(float)), "hipMalloc")) sl->cap = n;
    }
    if (rc == MDK_OK && !sl->ready) hip_ok(hipEventCreateWithFlags(&sl->ready, hipEventDisableTiming), "hipEventCreate");
    if (rc == MDK_OK) hip_ok(hipMemcpyAsync(sl->dev, x_host, n * sizeof(float), hipMemcpyHostToDevice, m->stage_stream), "hipMemcpyAsync");
    if (rc == MDK_OK) hip_ok(hipEventRecord(sl->ready, m->stage_stream), "hipEventRecord");
    std::lock_guard<std::mutex> lock(m->stage_mu);
    sl->busy = false;
    if (rc != MDK_OK) return rc;
    sl->B = B; sl->T = T;
    sl->token = m->stage_next_token++;
    *token = sl->token;
    return MDK_OK;
}

// ---- the next batch's forward, started ahead of its call ---------------------------------------------------------------------
// A staged call returns when its last result chunk has crossed PCIe and its certificate has been read: 0.5 - 1 ms during
// which the GPU has nothing to do (the second half of the last scan produces 40 MB of probabilities about as fast as one
// DMA engine ships them), then the caller's own work between two calls, then the launches of the next forward.  With the
// reference's loader (prediction.py:225-370) the next batch is usually on the device already (mdk_gru_stage_input): its
// forward is enqueued -- into the model's second context, results straight into the buffer the caller promises for it --
// BEFORE this call waits, ordered behind this call's last kernel (two recurrences that each hold every CU cannot share the
// chip; two sequential scans of the reference's batch sizes can, and then run side by side).  The call that redeems the next
// token finds its work in flight or done and only reads the certificate.  Bits: those of a lone call (same plan, same
// kernels, same margin -- a batch started ahead whose plan has moved by its call is waited for and recomputed).
static void release_slot(mdk_gru *m, mdk_gru::StageSlot *sl) {
    std::lock_guard<std::mutex> lock(m->stage_mu);
    sl->busy = false;
}

// nothing of a batch started ahead may survive: wait for it, free its slot (its token is spent: the caller's ordinary host
// entry answers).  Every entry but the pipelined one starts with this.
static void drop_pending(mdk_gru *m) {
    if (!m->pending.st.valid) return;
    m->pending.st.valid = false;
    m->early_dropped++;
    swap_ctx(m);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    swap_ctx(m);
    if (m->pending.slot) release_slot(m, m->pending.slot);
    m->pending.slot = nullptr;
}

extern "C" int mdk_gru_drop_pending(mdk_gru *m) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    HIP_TRY(hipSetDevice(m->device));
    drop_pending(m);
    return MDK_OK;
}

// enqueue the forward of the batch staged right after `token` (same shape), if it is there, into the other context
static int try_early_start(mdk_gru *m, unsigned long long token, int B, int T, float *next_probs_host) {
    if (!m->opt_early_start || !next_probs_host || m->pending.st.valid || m->timing) return MDK_OK;
    // a call that the learner will move (a smaller margin on trial) or that an audit / probe will repeat is not worth starting:
    // its plan is not known before the current call has been judged
    const int g_now = m->margin.cur ? m->margin.cur : m->opt_split_margin;
    if (m->opt_scan_split == 1 && m->opt_split_adapt > 0 && m->margin.quiet + 2 >= m->opt_split_adapt &&
        split_margin_down(g_now, m->margin.floor_) != 0) return MDK_OK;
    if (m->margin.trial_back) return MDK_OK;
    if (m->opt_scan_split && m->opt_split_audit == 1 &&
        (m->split_audited_key == 0 || (m->opt_split_audit_every > 0 && m->split_calls_since_audit + 2 >= m->opt_split_audit_every))) return MDK_OK;
    if (m->opt_split_audit == 2) return MDK_OK;
    mdk_gru::StageSlot *sl = nullptr;
    {
        std::lock_guard<std::mutex> lock(m->stage_mu);
        for (auto &c : m->stage)
            if (c.token == token + 1 && c.B == B && c.T == T && !c.busy) { sl = &c; c.busy = true; c.token = 0; }
    }
    if (!sl) return MDK_OK;
    swap_ctx(m);                               // the idle context becomes the current one
    int rc = init_ctx(m);
    if (!rc) rc = ensure_staging(m, 0, (size_t)B * T * m->desc.num_classes);
    if (!rc && hipStreamWaitEvent(m->stream, sl->ready, 0) != hipSuccess) rc = fail(MDK_ERR_DEVICE, "hipStreamWaitEvent failed");
    mdk_gru::Started st;
    if (!rc) rc = start_call(m, sl->dev, B, T, m->p_dev, m->stream, next_probs_host, &st, &m->other);
    if (rc) (void)hipStreamSynchronize(m->stream);
    swap_ctx(m);
    if (rc || !st.valid) {
        std::lock_guard<std::mutex> lock(m->stage_mu);      // not started: the token is good again
        sl->token = token + 1;
        sl->busy = false;
        return rc;
    }
    m->pending.st = st;
    m->pending.token = token + 1; m->pending.slot = sl; m->pending.B = B; m->pending.T = T; m->pending.probs_host = next_probs_host;
    m->early_started++;
    return MDK_OK;
}

extern "C" int mdk_gru_forward_pipelined(mdk_gru *m, unsigned long long token, int B, int T, float *probs_host, float *next_probs_host) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if (!probs_host || token == 0) return fail(MDK_ERR_ARG, "null buffer / token");
    HIP_TRY(hipSetDevice(m->device));
    const auto t_entry = std::chrono::steady_clock::now();
    mdk_gru::StageSlot *sl = nullptr;
    mdk_gru::Started pre;
    bool from_pending = false;
    if (m->pending.st.valid) {
        if (m->pending.token == token && m->pending.B == B && m->pending.T == T && m->pending.probs_host == probs_host) {
            swap_ctx(m);                       // the context this batch was started in becomes the current one
            from_pending = true;
            pre = m->pending.st;
            sl = m->pending.slot;
            m->pending.st.valid = false;
            m->pending.slot = nullptr;
        } else {
            drop_pending(m);                   // another batch, or another buffer than the one promised: its token is spent
        }
    }
    if (!sl) {
        std::lock_guard<std::mutex> lock(m->stage_mu);
        for (auto &c : m->stage)
            if (c.token == token && c.B == B && c.T == T && !c.busy) { sl = &c; c.busy = true; c.token = 0; m->stage_unredeemed = 0; }
    }
    if (!sl) return fail(MDK_ERR_ARG, "unknown or expired staging token (use mdk_gru_forward)");
    const size_t np = (size_t)B * T * m->desc.num_classes;
    int rc = ensure_staging(m, 0, np);
    if (!rc && !pre.valid) {
        if (hipStreamWaitEvent(m->stream, sl->ready, 0) != hipSuccess) rc = fail(MDK_ERR_DEVICE, "hipStreamWaitEvent failed");
        // this call's own first attempt, enqueue only -- so that the next batch's can follow it before anything is waited for
        if (!rc && next_probs_host) rc = start_call(m, sl->dev, B, T, m->p_dev, m->stream, probs_host, &pre, m->other.stream ? &m->other : nullptr);
    }
    static const bool dbg_t = getenv("MDK_EARLY_DEBUG") != nullptr;
    const auto t_a = std::chrono::steady_clock::now();
    // (a batch that cannot be started ahead -- no memory for the second context, say -- is no reason to fail THIS call: the
    // early start is switched off for the model and the batch takes the ordinary way when its call comes)
    auto start_next = [&]() {
        if (try_early_start(m, token, B, T, next_probs_host) != MDK_OK) {
            fprintf(stderr, "[medaka_amd] the next batch's forward could not be started ahead (%s): early start off for this model\n", g_mdk_err.c_str());
            m->opt_early_start = 0;
        }
    };
    if (!rc && pre.valid) start_next();
    const auto t_b = std::chrono::steady_clock::now();
    const long used_before = m->early_used;
    if (!rc) rc = run_forward(m, sl->dev, B, T, m->p_dev, m->stream, nullptr, probs_host, &pre);
    const auto t_c = std::chrono::steady_clock::now();
    if (rc) (void)hipDeviceSynchronize();
    else if (hipStreamSynchronize(m->stream) != hipSuccess) rc = fail(MDK_ERR_DEVICE, "hipStreamSynchronize failed");
    if (dbg_t) {
        const auto t_d = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        if (ms(t_entry, t_d) > 8.0)
            fprintf(stderr, "[medaka_amd] slow staged call: own enqueue %.2f ms, next batch's enqueue %.2f ms, run_forward (wait + certificate) %.2f ms, "
                            "final synchronize %.2f ms\n", ms(t_entry, t_a), ms(t_a, t_b), ms(t_b, t_c), ms(t_c, t_d));
    }
    release_slot(m, sl);
    m->staged_used++;
    m->last.host_streamed |= 4;
    if (from_pending && m->early_used != used_before) m->last.host_streamed |= 8;
    // the batch behind this one may have landed only now: its forward then runs under whatever the caller does between two calls
    if (!rc) start_next();
    return rc;
}

extern "C" int mdk_gru_forward_staged(mdk_gru *m, unsigned long long token, int B, int T, float *probs_host) {
    return mdk_gru_forward_pipelined(m, token, B, T, probs_host, nullptr);
}

extern "C" int mdk_gru_forward(mdk_gru *m, const float *x_host, int B, int T, float *probs_host) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if (B < 0 || T < 0) return fail(MDK_ERR_ARG, "negative shape B=%d T=%d", B, T);
    if (B == 0 || T == 0) { memset(&m->last, 0, sizeof(m->last)); return MDK_OK; }
    if (!x_host || !probs_host) return fail(MDK_ERR_ARG, "null buffer");
    HIP_TRY(hipSetDevice(m->device));
    drop_pending(m);
    const size_t nx = (size_t)B * T * m->desc.num_features, np = (size_t)B * T * m->desc.num_classes;
    int rc = ensure_staging(m, nx, np);
    if (rc) return rc;
    // x streams in and the probabilities stream out while the recurrences run (forward_pass, HostIO)
    rc = run_forward(m, m->x_dev, B, T, m->p_dev, m->stream, x_host, probs_host);
    if (rc) { (void)hipDeviceSynchronize(); return rc; }   // nothing of ours may still touch the caller's buffers
    HIP_TRY(hipStreamSynchronize(m->stream));
    return MDK_OK;
}

// ------------------------------------------------------------------------------------------
// f2 / f3: device-side normalisation of raw counts and argmax decode (PCIe diet)
extern "C" int mdk_normalise_counts_dev(const uint16_t *counts_dev, const uint32_t *depth_dev, long n_cols,
                                        int n_features, float *x_dev, int device, void *stream) {
    if (n_cols < 0 || n_features < 1) return fail(MDK_ERR_ARG, "bad shape n_cols=%ld n_features=%d", n_cols, n_features);
    if (n_cols == 0) return MDK_OK;
    if (!counts_dev || !depth_dev || !x_dev) return fail(MDK_ERR_ARG, "null buffer");
    HIP_TRY(hipSetDevice(device));
    const long n = n_cols * n_features;
    hipLaunchKernelGGL(k_normalise_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       counts_dev, depth_dev, x_dev, n_cols, n_features);
    HIP_TRY(hipGetLastError());
    return MDK_OK;
}

extern "C" int mdk_decode_dev(const float *probs_dev, long n_cols, int n_classes, uint8_t *cls_dev, float *pmax_dev,
                              int device, void *stream) {
    if (n_cols < 0 || n_classes < 1 || n_classes > 255) return fail(MDK_ERR_ARG, "bad shape n_cols=%ld n_classes=%d", n_cols, n_classes);
    if (n_cols == 0) return MDK_OK;
    if (!probs_dev || !cls_dev || !pmax_dev) return fail(MDK_ERR_ARG, "null buffer");
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(k_decode, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, probs_dev,
                       cls_dev, pmax_dev, n_cols, n_classes);
    HIP_TRY(hipGetLastError());
    return MDK_OK;
}

// shared body of the two host entries: exactly one of x_host / counts_host is given
static int forward_any(mdk_gru *m, const float *x_host, const uint16_t *counts_host, const uint32_t *depth_host, int B,
                       int T, float *probs_host, uint8_t *cls_host, float *pmax_host) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if (B < 0 || T < 0) return fail(MDK_ERR_ARG, "negative shape B=%d T=%d", B, T);
    if (B == 0 || T == 0) { memset(&m->last, 0, sizeof(m->last)); return MDK_OK; }
    if (!x_host && !(counts_host && depth_host)) return fail(MDK_ERR_ARG, "null input buffer");
    if (!probs_host && !(cls_host && pmax_host)) return fail(MDK_ERR_ARG, "no output requested (probs, or cls + pmax)");
    if ((cls_host == nullptr) != (pmax_host == nullptr)) return fail(MDK_ERR_ARG, "cls and pmax go together");
    HIP_TRY(hipSetDevice(m->device));
    drop_pending(m);
    const int F = m->desc.num_features, C = m->desc.num_classes;
    const size_t cols = (size_t)B * T, nx = cols * F, np = cols * C;
    { int rc0 = ensure_staging(m, nx, np); if (rc0) return rc0; }
    // aux: [depth u32 | pmax f32 (cols)] [counts u16 (cols*F)] [cls u8 (cols)], 16-byte aligned pieces
    const size_t off_counts = (cols * 4 + 15) / 16 * 16, off_cls = off_counts + (cols * F * 2 + 15) / 16 * 16;
    const size_t aux_need = off_cls + cols;
    if (aux_need > m->aux_cap) {
        free_dev(m->aux_dev); m->aux_dev = nullptr; m->aux_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->aux_dev, aux_need));
        m->aux_cap = aux_need;
    }
    hipStream_t s = m->stream;
    if (counts_host) {
        uint32_t *dd = reinterpret_cast<uint32_t *>(m->aux_dev);
        uint16_t *cd = reinterpret_cast<uint16_t *>(m->aux_dev + off_counts);
        HIP_TRY(hipMemcpyAsync(dd, depth_host, cols * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(cd, counts_host, cols * F * 2, hipMemcpyHostToDevice, s));
        int rc = mdk_normalise_counts_dev(cd, dd, (long)cols, F, m->x_dev, m->device, s);
        if (rc) return rc;
    }
    // float features stream in, probabilities (if wanted) stream out under the recurrences (HostIO)
    int rc = run_forward(m, m->x_dev, B, T, m->p_dev, s, counts_host ? nullptr : x_host, probs_host);
    if (rc) { (void)hipDeviceSynchronize(); return rc; }
    if (cls_host) {
        float *pm = reinterpret_cast<float *>(m->aux_dev);          // depth is dead by now
        uint8_t *cl = m->aux_dev + off_cls;
        rc = mdk_decode_dev(m->p_dev, (long)cols, C, cl, pm, m->device, s);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(cls_host, cl, cols, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(pmax_host, pm, cols * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return MDK_OK;
}

extern "C" int mdk_gru_forward_counts(mdk_gru *m, const uint16_t *counts_host, const uint32_t *depth_host, int B,
                                      int T, float *probs_host, uint8_t *cls_host, float *pmax_host) {
    if (m && B > 0 && T > 0 && !(counts_host && depth_host)) return fail(MDK_ERR_ARG, "null input buffer");
    return forward_any(m, nullptr, counts_host, depth_host, B, T, probs_host, cls_host, pmax_host);
}

extern "C" int mdk_gru_forward_decoded(mdk_gru *m, const float *x_host, int B, int T, uint8_t *cls_host,
                                       float *pmax_host) {
    if (m && B > 0 && T > 0 && !x_host) return fail(MDK_ERR_ARG, "null input buffer");
    return forward_any(m, x_host, nullptr, nullptr, B, T, nullptr, cls_host, pmax_host);
}

// ------------------------------------------------------------------------------------------
// majority-vote model
extern "C" int mdk_majority_forward_dev(const float *x_dev, long n_cols, float *probs_dev, int device,
                                        void *stream) {
    if (n_cols < 0) return fail(MDK_ERR_ARG, "negative n_cols");
    if (n_cols == 0) return MDK_OK;
    if (!x_dev || !probs_dev) return fail(MDK_ERR_ARG, "null buffer");
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(k_majority, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x_dev, probs_dev, n_cols);
    HIP_TRY(hipGetLastError());
    return MDK_OK;
}

// (the host entry's device buffers are kept per device, grow-only: a hipMalloc / hipFree pair per call costs milliseconds, and
// freed device memory is wiped by the driver on the DMA engines -- behind which any engine's strided result copies wait)
namespace {
struct MajorityBuffers { std::mutex mu; float *x = nullptr, *p = nullptr; size_t cols = 0; };
MajorityBuffers g_majority[16];
}

extern "C" int mdk_majority_forward(const float *x_host, long n_cols, float *probs_host, int device) {
    if (n_cols < 0) return fail(MDK_ERR_ARG, "negative n_cols");
    if (n_cols == 0) return MDK_OK;
    if (!x_host || !probs_host) return fail(MDK_ERR_ARG, "null buffer");
    if (device < 0 || device >= 16) return fail(MDK_ERR_ARG, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    MajorityBuffers &b = g_majority[device];
    std::lock_guard<std::mutex> lock(b.mu);
    if ((size_t)n_cols > b.cols) {
        free_dev(b.x); free_dev(b.p); b.x = b.p = nullptr; b.cols = 0;
        HIP_TRY(hipMalloc((void **)&b.x, (size_t)n_cols * 10 * sizeof(float)));
        hipError_t e = hipMalloc((void **)&b.p, (size_t)n_cols * 5 * sizeof(float));
        if (e != hipSuccess) { free_dev(b.x); b.x = nullptr; return fail(MDK_ERR_OOM, "hipMalloc failed: %s", hipGetErrorString(e)); }
        b.cols = (size_t)n_cols;
    }
    if (hipMemcpy(b.x, x_host, (size_t)n_cols * 10 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return fail(MDK_ERR_DEVICE, "H2D copy failed");
    int rc = mdk_majority_forward_dev(b.x, n_cols, b.p, device, nullptr);
    if (!rc && hipMemcpy(probs_host, b.p, (size_t)n_cols * 5 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(MDK_ERR_DEVICE, "D2H copy failed");
    return rc;
}

// ------------------------------------------------------------------------------------------
// raw device helpers
extern "C" int mdk_device_count(int *count) {
    if (!count) return fail(MDK_ERR_ARG, "null argument");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return fail(MDK_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return MDK_OK;
}
extern "C" int mdk_device_name(int device, char *buf, size_t buflen) {
    if (!buf || buflen == 0) return fail(MDK_ERR_ARG, "null argument");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return MDK_OK;
}
extern "C" int mdk_dev_alloc(int device, size_t bytes, void **ptr) {
    if (!ptr) return fail(MDK_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(ptr, bytes));
    return MDK_OK;
}
extern "C" int mdk_dev_free(int device, void *ptr) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(ptr));
    return MDK_OK;
}
// page-locked host memory: buffers handed to mdk_gru_forward / mdk_rl_forward from here are copied by
// DMA without a staging pass and are never page-faulted in by the copy (a fresh 40 MB malloc costs 3.6 ms
// of first-touch faults as a copy target: profiles/r2_host_path_probe.txt)
extern "C" int mdk_host_alloc(size_t bytes, void **ptr) {
    if (!ptr) return fail(MDK_ERR_ARG, "null argument");
    HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return MDK_OK;
}
extern "C" int mdk_host_free(void *ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return MDK_OK;
}
// Batch assembly (reference Batch.collate -> torch.stack, torch_ext.py:147-148) with a few host threads.
extern "C" int mdk_gather_rows(void *dst, const void *const *rows, int n_rows, size_t row_bytes, int n_threads) {
    if (n_rows < 0) return fail(MDK_ERR_ARG, "negative n_rows");
    if (n_rows == 0 || row_bytes == 0) return MDK_OK;
    if (!dst || !rows) return fail(MDK_ERR_ARG, "null buffer");
    for (int i = 0; i < n_rows; ++i)
        if (!rows[i]) return fail(MDK_ERR_ARG, "row %d is null", i);
    n_threads = std::max(1, std::min(std::min(n_threads, 64), n_rows));
    auto work = [=](int k) {
        const int lo = (int)((long)n_rows * k / n_threads), hi = (int)((long)n_rows * (k + 1) / n_threads);
        for (int i = lo; i < hi; ++i) memcpy(static_cast<char *>(dst) + (size_t)i * row_bytes, rows[i], row_bytes);
    };
    if (n_threads == 1) { work(0); return MDK_OK; }
    std::vector<std::thread> pool;
    try {
        for (int k = 1; k < n_threads; ++k) pool.emplace_back(work, k);
    } catch (...) {            // thread creation failed: finish what was not handed out on this thread
        const int started = (int)pool.size();
        for (int k = started + 1; k < n_threads; ++k) work(k);
    }
    work(0);
    for (auto &t : pool) t.join();
    return MDK_OK;
}
extern "C" int mdk_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return MDK_OK;
}
extern "C" int mdk_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return MDK_OK;
}
extern "C" int mdk_device_synchronize(int device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return MDK_OK;
}

#ifdef MDK_DEBUG_HOOKS
// ------------------------------------------------------------------------------------------
// Test hook: keep `blocks` CUs busy with a compute-bound loop (profiles/soak_wide.py uses it as the competing
// tenant of the LSTM(384) cluster recurrence).  Synchronous.
__global__ __launch_bounds__(512, 1) void k_burn(float *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) { a = fmaf(a, b, c); c = fmaf(c, b, a); }
    if (a + c == 12345.678f) out[0] = a;
}
extern "C" int mdk_selftest_burn(int device, int blocks, int iters) {
    if (blocks < 1 || iters < 0) return fail(MDK_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 4));
    hipLaunchKernelGGL(k_burn, dim3(blocks), dim3(512), 0, nullptr, d, iters);
    hipError_t e = hipDeviceSynchronize();
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MDK_ERR_DEVICE, "burn kernel failed: %s", hipGetErrorString(e));
    return MDK_OK;
}

// Test hook: hold `blocks` CUs EXCLUSIVELY (one 512-thread work-group each with `lds_bytes` of LDS, e.g. 140 KB, so
// that nothing else fits beside it) for `milliseconds` of wall clock.  Synchronous; tests/test_parity_gpu.py uses it from
// a second thread as the tenant that leaves the LSTM(384) cluster recurrence fewer CUs than it needs.
__global__ __launch_bounds__(512, 1) void k_hold(unsigned long long ticks) {
    extern __shared__ unsigned char hold_lds[];
    hold_lds[threadIdx.x] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" int mdk_selftest_hold(int device, int blocks, int milliseconds, int lds_bytes) {
    if (blocks < 1 || milliseconds < 0 || milliseconds > 10000 || lds_bytes < 512 || lds_bytes > 160 * 1024)
        return fail(MDK_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hold), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_hold, dim3(blocks), dim3(512), (size_t)lds_bytes, st, (unsigned long long)milliseconds * 100000ull);   // 100 MHz
    hipError_t e = hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    if (e != hipSuccess) return fail(MDK_ERR_DEVICE, "hold kernel failed: %s", hipGetErrorString(e));
    return MDK_OK;
}
#endif   // MDK_DEBUG_HOOKS


// ------------------------------------------------------------------------------------------
// MFMA self-test: D = A(16x32) B(32x16) with the fragment layout the kernels assume, on
// asymmetric integer data (exact in fp16/fp32), plus an fp16-subnormal operand probe.
__global__ void k_selftest(const _Float16 *A /*[16][32]*/, const _Float16 *Bm /*[32][16]*/,
                           float *Dm /*[16][16]*/) {
    const int lane = threadIdx.x;
    half8 a, b;
    for (int i = 0; i < 8; ++i) {
        const int k = (lane >> 4) * 8 + i;
        a[i] = A[(lane & 15) * 32 + k];
        b[i] = Bm[k * 16 + (lane & 15)];
    }
    floatx4 c = {0.f, 0.f, 0.f, 0.f};
    c = mfma16(a, b, c);
    for (int r = 0; r < 4; ++r) Dm[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = c[r];
}

extern "C" int mdk_selftest_mfma(int device, float *max_abs_err, int *subnormal_preserved) {
    if (!max_abs_err || !subnormal_preserved) return fail(MDK_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    std::vector<_Float16> A(16 * 32), Bm(32 * 16);
    std::vector<float> ref(256, 0.f), got(256);
    for (int pass = 0; pass < 2; ++pass) {
        for (int m_ = 0; m_ < 16; ++m_)
            for (int k = 0; k < 32; ++k) A[m_ * 32 + k] = (_Float16)(float)((m_ * 7 + k * 3) % 11 - 5);
        for (int k = 0; k < 32; ++k)
            for (int n = 0; n < 16; ++n) Bm[k * 16 + n] = (_Float16)(float)((k * 5 + n * 2 + k * n) % 13 - 6);
        if (pass == 1) {
            // subnormal probe: A[0][0] = 2^-20 (fp16 subnormal), B[0][0] = 1024, rest of row/col 0 zero
            for (int k = 0; k < 32; ++k) { A[k] = (_Float16)0.f; Bm[k * 16] = (_Float16)0.f; }
            A[0] = (_Float16)9.5367431640625e-07f;
            Bm[0] = (_Float16)1024.f;
        }
        for (int m_ = 0; m_ < 16; ++m_)
            for (int n = 0; n < 16; ++n) {
                float acc = 0.f;
                for (int k = 0; k < 32; ++k) acc += (float)A[m_ * 32 + k] * (float)Bm[k * 16 + n];
                ref[m_ * 16 + n] = acc;
            }
        _Float16 *dA = nullptr, *dB = nullptr;
        float *dD = nullptr;
        HIP_TRY(hipMalloc((void **)&dA, A.size() * 2));
        HIP_TRY(hipMalloc((void **)&dB, Bm.size() * 2));
        HIP_TRY(hipMalloc((void **)&dD, 256 * 4));
        HIP_TRY(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dB, Bm.data(), Bm.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, nullptr, dA, dB, dD);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(got.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
        if (pass == 0) {
            float e = 0.f;
            for (int i = 0; i < 256; ++i) e = std::max(e, std::fabs(got[i] - ref[i]));
            *max_abs_err = e;
        } else {
            *subnormal_preserved = (std::fabs(got[0] - ref[0]) <= 1e-6f * std::fabs(ref[0])) ? 1 : 0;
        }
    }
    return MDK_OK;
}
