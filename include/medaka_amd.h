/*
 * medaka_amd.h -- C ABI of the MI355X-native consensus-inference engine.
 *
 * This is the drop-in boundary for ONE path of nanoporetech/medaka: the network forward
 * pass behind `model.predict_on_batch(batch)` (reference medaka/prediction.py:46 ->
 * medaka/models.py:303-313 -> medaka/architectures/gru.py:58-72).  Everything above it
 * (prediction.py's run_prediction loop, torch_ext.Batch, model loading) and either side of it
 * (features.py / src/medaka_counts.c, stitch.py) is unchanged reference code.
 *
 * Plain pointers and sizes only: no torch / HIP types appear in any signature (a HIP stream is
 * passed as `void*`).  All functions return 0 on success and a non-zero code on failure;
 * `mdk_last_error()` then returns a thread-local human-readable message.  Nothing here calls
 * `exit()` (the reference C library does, src/medaka_counts.c:203-206 -- deliberately not
 * mirrored).
 *
 * There is NO CPU fallback behind this interface: if no HIP device is usable every compute
 * entry point fails with MDK_ERR_DEVICE.
 */
#ifndef MEDAKA_AMD_H
#define MEDAKA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDK_OK 0
#define MDK_ERR_ARG 1      /* bad argument / unsupported architecture */
#define MDK_ERR_DEVICE 2   /* HIP runtime error (message has hipGetErrorString) */
#define MDK_ERR_OOM 3      /* device allocation failed: lower the batch size (reference README.md:156-159) */

/* Precision of the recurrent / projection contractions. */
#define MDK_PREC_FP32 0  /* parity mode: fp16 hi+lo split operands, fp32 accumulate (<=1e-6 of fp32) */
#define MDK_PREC_FP16 1  /* what `TorchModel.half()` selects (models.py:298-301): fp16 operands */

/* Kernel family used by mdk_gru_forward*. */
#define MDK_VARIANT_MFMA 0   /* production kernels (MFMA, LDS-resident hidden state) */
#define MDK_VARIANT_EXACT 1  /* plain fp32 VALU kernels: slow, bit-simple, for on-device cross-checks */

typedef struct mdk_gru mdk_gru;

/*
 * Architecture of reference `GRUModel.__init__` (medaka/architectures/gru.py:13-56).
 * Supported: hidden == 128, 1 <= num_layers <= 4, bidirectional 0/1, num_features <= 256,
 * num_classes == 5 (the reference hard-codes Linear(.., 5), gru.py:53-55).
 */
typedef struct {
    int num_features;  /* 10: channels a c g t A C G T d D (src/medaka_counts.h:19-30) */
    int hidden;        /* gru_size, 128 for every bundled consensus / variant model */
    int num_layers;    /* 2 */
    int bidirectional; /* 1 */
    int num_classes;   /* 5: '*ACGT' (medaka/labels.py:342) */
    int normalise;     /* 1: softmax over classes (gru.py:68-71); 0: logits */
} mdk_gru_desc;

/* Per-kernel device time of the last timed forward, milliseconds (hipEvent based). */
typedef struct {
    float h2d_ms;         /* always 0 since the host copies stream under the recurrences (mdk_gru_forward): there is no
                             separate copy span to report; kept for ABI stability -- time the call itself instead */
    float gi_ms[4];       /* input projection per layer */
    float rec_ms[4];      /* recurrence per layer (both directions) -- the dominant kernel */
    float head_ms;        /* Linear + softmax */
    float d2h_ms;         /* always 0, as h2d_ms */
    float total_ms;       /* first kernel start -> last kernel end (device-resident region) */
    int rec_launches;     /* recurrence launches in the last forward */
    int n_layers;
    int host_streamed;    /* split calls through mdk_gru_forward: bit 1 = the probabilities left in column chunks under the
                             second half of the last layer's scan (option "stream_host"; else one copy after the forward; bit 5: the last of them by
                             kernel, "tail_blit"); bit 2 = x had
                             been handed over early (mdk_gru_forward_staged: no PCIe wait for the input inside the call); bit 3 = the
                             forward itself had been enqueued ahead of the call (mdk_gru_forward_pipelined) */
    int fused_layers;     /* bit l set: layer l ran with its input projection fused into the recurrence (option
                             "fuse_proj"; its gi_ms is then 0 and its rec_ms covers both); bit 8: the classifier's Linear
                             ran inside the last layer's kernel as well ("fuse_head": head_ms is the combine kernel); bit 9: ... and the
                             scan's second half wrote the probabilities itself ("final_head": no head kernel, head_ms ~ 0) */
} mdk_gru_timing;

/* What the last forward did about splitting the scan (option "scan_split" below). */
#define MDK_SPLIT_NOT_USED 0   /* shape not latency-bound, option off, or model outside the split path */
#define MDK_SPLIT_CERTIFIED 1  /* ran as `chunks` chunks per window; every junction certified */
#define MDK_SPLIT_REJECTED 2   /* a junction differed by more than 2^-18 (half precision: 2^-10) at every margin tried:
                                  the call was answered by the sequential scan */
#define MDK_SPLIT_DISABLED 3   /* an earlier call was rejected at the largest margin: sequential scans for a back-off of 64 .. 4096
                                  calls, then one more try (auto mode); after a failed AUDIT: for good */
typedef struct {
    int chunks;       /* chunks per window of the last forward (1 = sequential scan) */
    int margin;       /* warm-up columns on either side of a chunk (the margin the model has escalated to) */
    int columns;      /* columns of one virtual window (T when not split) */
    int status;       /* MDK_SPLIT_* */
    float max_delta;  /* largest |h_warm - h_carried| over all certificate points of the last split forward */
    int fallbacks;    /* rejected certificates since the model was created (each cost one repeated forward) */
    int audited;      /* 1: this call was also run as the sequential scan and the two results compared in full */
    float audit_max_dp; /* largest |p_split - p_sequential| of that comparison */
    int audits;           /* audited calls since the model was created (first calls + every "scan_split_audit_every"-th) */
    int audit_failures;   /* ... of which found a difference above the audit tolerance (the split is then turned off) */
    float audit_worst_dp; /* largest |p_split - p_sequential| any audit has seen */
    int probes;           /* half precision: calls that were also run once in fp32-parity mode to certify the margin ("scan_split_probe") */
    float probe_max_delta; /* largest junction difference of the last such probe (threshold 2^-18) */
} mdk_gru_split;

/*
 * Replaces: `ModelStoreTGZ.load_model` -> `GRUModel(...)`; `load_state_dict`; `.to(device)`
 * (medaka/datastore.py:135-157, medaka/architectures/gru.py:46-56).
 * `weights` holds host pointers to contiguous fp32 tensors in torch `state_dict()` order:
 *   for layer l, for direction d in (fwd[, reverse]):
 *       weight_ih (3H x K_l), weight_hh (3H x H), bias_ih (3H), bias_hh (3H)
 *   then linear.weight (C x D*H), linear.bias (C);   n_weights = 4*L*D + 2.
 * Gate row blocks are ordered r, z, n (PyTorch nn.GRU).  Weights are copied; the caller keeps
 * ownership of its buffers.  `device` is the HIP device ordinal among the visible devices
 * (the reference uses device 0 of the visible set, medaka/prediction.py:136-138).
 */
int mdk_gru_create(const mdk_gru_desc *desc, const float *const *weights, int n_weights,
                   int device, mdk_gru **out);

/*
 * Replaces: `TorchModel.predict_on_batch` (medaka/models.py:303-313) for CountsMatrixModel
 * input (`batch.counts_matrix`, medaka/architectures/base_classes.py:9-11):
 * x_host: B x T x num_features contiguous fp32 (host);  probs_host: B x T x num_classes fp32
 * (host), fully overwritten.  Synchronous: returns when probs_host is complete.  B, T >= 0;
 * any B (last batch is short, medaka/common.py:903-916) and any T (the un-chunked B=1 second
 * pass, medaka/prediction.py:196-209).  Internally x is copied in and the probabilities out in time
 * slabs while the recurrences run (bidirectional models, T >= 2048, T % 16 == 0; otherwise one copy
 * each side); a call that runs as a split scan (option "scan_split") copies x in once, in front of the
 * forward, and sends the probabilities home in column chunks under the second half of the last layer's
 * scan (options "final_head", "stream_host"; mdk_gru_timing.host_streamed says what the last call did).
 * Pageable and page-locked buffers are both accepted.
 */
int mdk_gru_forward(mdk_gru *m, const float *x_host, int B, int T, float *probs_host);

/* Early hand-over of the NEXT batch (no reference counterpart; what the engine's `Batch.collate` does from the reference's
 * Batcher thread, prediction.py:356-370): starts the host -> device copy of x_host (B x T x num_features fp32, page-locked
 * for a truly asynchronous copy) on a stream of its own and returns at once with a token.  mdk_gru_forward_staged(token,
 * ...) later answers exactly like mdk_gru_forward(x_host, ...) without waiting for PCIe: the 1.4 ms an 80 MB batch
 * takes to cross overlap the previous batch's forward.  x_host must stay untouched until the staged forward returns.
 * Ten batches can be staged (the reference's loader keeps up to 8 in its queue, one is being read by the forward, one may have been started ahead); a token that was pushed out (or never redeemed) makes mdk_gru_forward_staged
 * return MDK_ERR_ARG -- call mdk_gru_forward instead.  Callable from another thread than the forwards. */
int mdk_gru_stage_input(mdk_gru *m, const float *x_host, int B, int T, unsigned long long *token);
int mdk_gru_forward_staged(mdk_gru *m, unsigned long long token, int B, int T, float *probs_host);
/* mdk_gru_forward_staged that also STARTS THE NEXT BATCH'S FORWARD before it waits for this one's last result chunk (no reference
 * counterpart; `run_prediction` keeps one call in flight, prediction.py:44-52 -- this keeps the GPU busy between two of them).
 * `next_probs_host` (B x T x num_classes fp32, page-locked): the buffer the caller will pass as `probs_host` of its NEXT call.  If
 * the batch staged right after `token` (same B, T) is there, its forward is enqueued into the model's second context -- behind
 * this call's last kernel where the two cannot share the chip, beside it where they can (sequential scans of <= 128
 * work-groups) -- and its probabilities stream into `next_probs_host`; the call that redeems that token with that buffer finds
 * its work in flight or done (mdk_gru_timing.host_streamed bit 3).  Same bits as a lone call.  The buffer must stay valid and
 * untouched until that call has returned, or until mdk_gru_drop_pending / mdk_gru_destroy / any other entry of the model (all of
 * which wait for the batch started ahead and forget it; its token is then spent: mdk_gru_forward answers).  NULL: exactly
 * mdk_gru_forward_staged.  Option "early_start" = 0 (environment MDK_EARLY_START) turns the early start off.
 * Option "stage_overlap" = 2 | 1 | 0 (environment MDK_STAGE_OVERLAP): a split call started ahead runs its LAYER 0 beside the
 * previous batch's LAYER 1 -- a layer-0 work-group (8 KB of LDS, a latency chain that leaves the matrix pipe idle most of its
 * step) shares a CU with a fused layer-1 work-group; layers of the same kind still follow each other (2: both precisions,
 * 1: half precision only, 0: a batch started ahead waits for the previous batch's last kernel). */
int mdk_gru_forward_pipelined(mdk_gru *m, unsigned long long token, int B, int T, float *probs_host, float *next_probs_host);
int mdk_gru_drop_pending(mdk_gru *m);

/*
 * Same contraction with device-resident buffers (what `GRUModel.forward`, gru.py:58-72, is to
 * `predict_on_batch`).  x_dev / probs_dev are device pointers valid on the model's device;
 * `stream` is a hipStream_t (NULL = the default stream); all work is enqueued on it, in order.
 * Asynchronous with respect to the host unless timing is enabled or the call runs as a split scan ("scan_split":
 * the certificate is read back, and a rejected call repeated, before the function returns).
 */
int mdk_gru_forward_dev(mdk_gru *m, const float *x_dev, int B, int T, float *probs_dev,
                        void *stream);

/* Replaces `TorchModel.half()` (models.py:298-301): MDK_PREC_FP32 (default) / MDK_PREC_FP16. */
int mdk_gru_set_precision(mdk_gru *m, int precision);
int mdk_gru_set_variant(mdk_gru *m, int variant);
/* `normalise` attribute of GRUModel (gru.py:56,68): 1 softmax, 0 logits. */
int mdk_gru_set_normalise(mdk_gru *m, int normalise);

/* Tuning knobs (no reference counterpart):
 *   "rec_windows_per_tile" = 0 (auto) | 4 | 8      recurrence work-group granularity
 *   "fuse_l0"              = 1 | 0                  fuse the layer-0 input projection (default 1)
 *   "fuse_proj"            = 1 (auto) | 0 | 2 (always)  layers >= 1: the input projection runs inside the recurrence
 *                                                   kernel, strip by strip, and its result never exists in HBM
 *                                                   (bit-identical to the separate GEMM; both precisions, 8-window
 *                                                   work-groups, T % 8 == 0; auto: when the recurrence's work-groups fill
 *                                                   the chip (> 208 of them); environment MDK_FUSE_PROJ)
 *   "fuse_head"            = 1 | 0                  with a fused last layer: Linear(D*128 -> 5) inside its kernel as well
 *                                                   (fp16x2-split MFMA on the h image already in LDS; the logits agree with
 *                                                   the fp32 FMA head to ~1e-7, not bit for bit; environment MDK_FUSE_HEAD)
 *   "final_head"           = 1 | 0                  with a fused head: the launches of the scan's second half (T % 16 == 0;
 *                                                   every launch of a one-directional model) add the other direction's
 *                                                   partial logits, the bias and the softmax themselves and store the
 *                                                   probabilities -- the arithmetic of the combine kernel, same bits; no
 *                                                   head kernel, and finished columns can leave for the host while the
 *                                                   scan runs on (environment MDK_FINAL_HEAD)
 *   "overlap_gemm"         = 1 (auto) | 0 | 2 (force)  project layer 1 (and the head) on a side stream under
 *                                                   the tails of the recurrences (bidirectional, T >= 2048,
 *                                                   T % 16 == 0; auto: while the recurrence leaves CUs idle)
 *   "deferred_store"       = 1 | 0                  recurrence: h_t leaves for HBM from inside step t+1 (default 1)
 *   "gpu_share"            = 1 .. 8                 processes sharing this GPU (medaka_amd.launch --procs-per-gpu):
 *                                                   work-group sizes are chosen so that all of them fit the chip
 *   "scan_split"           = 1 (auto) | 0 | 2..16   split the scan.  A batch that leaves most of the GPU idle
 *                                                   (chunks = min(W / B, T / (4 * margin)) >= 3 -- or 2 when T is the limit or
 *                                                   the GPU is shared -- with W = 1024 chunk-windows, 1600 / gpu_share
 *                                                   for a process that shares the GPU) runs as `chunks` chunks per window, each warmed up over
 *                                                   "scan_split_margin" columns on either side, as ONE batch of
 *                                                   B * chunks windows of about T / chunks + 2 * margin columns.  The
 *                                                   states at every junction are compared on the device (both layers,
 *                                                   both directions, at the junction and margin / 2 columns past it);
 *                                                   if any differs by more than 2^-18 (2^-10 in half-precision mode)
 *                                                   the call is repeated with the next larger margin of the ladder 64, 96,
 *                                                   128, 192, 256, 384, 512 -- kept for later calls -- and beyond a margin of 512 as the sequential scan, which
 *                                                   the model then stays on for a back-off of 64 calls (doubling up to
 *                                                   4096 per further rejection) before the split is tried again: the
 *                                                   rejection may have been that input's doing.  n >= 2 forces n chunks (no escalation:
 *                                                   a rejected call is answered by the sequential scan).
 *                                                   Bidirectional 2-layer models, T >= 8 * margin.  Results agree with
 *                                                   the sequential scan to ~1e-7 (not bit for bit) and depend, at that
 *                                                   level, on B and on the margin the model has escalated to.
 *   "scan_split_adapt"     = 8 | 0 | n              the margin is LEARNED per model: after n certified calls in a row whose largest junction
 *                                                   difference sat at the rounding-noise floor (a quarter of the threshold) the next call tries
 *                                                   the next smaller margin of the ladder; a trial that is rejected is repeated at the margin
 *                                                   that worked, and no shrink goes below a rejected margin again.  0: margins only grow.
 *   "scan_split_probe"     = 1 | 0                  half precision, auto mode: a margin is used only after a call certified at it in
 *                                                   FP32-PARITY mode (the same call run once more with hi/lo operands and the 2^-18
 *                                                   threshold, result discarded: one extra forward per margin the learner visits
 *                                                   and one per standing audit).  Half mode's own certificate compares fp16 images
 *                                                   of h (threshold 2^-10) and cannot see an un-merged state below ~1e-3; without
 *                                                   the probe its learner shrinks to margins fp32 parity rejects for the same
 *                                                   weights.  0: trust the half certificate alone (environment MDK_SCAN_SPLIT_PROBE)
 *   "scan_split_audit"     = 1 | 0 | 2              1: the first certified call of a model -- and the first at every margin
 *                                                   it escalates to, and every "scan_split_audit_every"-th after that -- is
 *                                                   also run as the sequential scan and the two
 *                                                   results are compared in full (1e-5; half precision 4e-4); a mismatch
 *                                                   delivers the sequential result and turns the split off.  One extra
 *                                                   forward per model (it allocates no workspace of its own and frees none:
 *                                                   device memory handed back to the driver is wiped on the DMA engines, and
 *                                                   the host path's result copies wait behind that).  2: every certified
 *                                                   call (debug), 0: never
 *   "scan_split_audit_every" = 256 | n >= 0         standing audit: with "scan_split_audit" = 1, every n-th certified call after
 *                                                   the first is audited the same way (0: only the first of every margin);
 *                                                   mdk_gru_split.audits / audit_failures / audit_worst_dp count them
 *   "scan_split_margin"    = 128 | multiple of 8 in 16..4096   (environment MDK_SCAN_SPLIT / MDK_SCAN_SPLIT_MARGIN, read
 *                                                   when a model is created, set the defaults of these two options)
 *   "tail_blit"            = 1 | 0                  mdk_gru_forward with a page-locked result buffer: the result chunks of a split call's last
 *                                                   two launches are written home by a kernel behind the last recurrence instead of queueing
 *                                                   behind the DMA copies of the earlier ones (environment MDK_TAIL_BLIT)
 *   "early_start"          = 1 | 0                  mdk_gru_forward_pipelined: enqueue the next staged batch's forward before waiting for
 *                                                   the current one (see that entry; environment MDK_EARLY_START)
 *   "stage_overlap"        = 2 | 1 | 0              ... and run its layer 0 beside the current batch's layer 1 (2: both precisions, 1: half
 *                                                   precision only, 0: whole forwards follow each other; environment MDK_STAGE_OVERLAP)
 *   "stream_host"          = 1 | 0                  mdk_gru_forward, sequential scan: copy x in / probabilities out in time
 *                                                   slabs under the recurrences (0: one copy before, one after).  A split
 *                                                   call copies x in once (all of it is needed at once) and sends the
 *                                                   probabilities home in column chunks under the second half of the last
 *                                                   layer's scan when that half writes them itself ("final_head"; DMA
 *                                                   only: no kernel can run beside recurrences that hold every CU),
 *                                                   as one copy behind the forward otherwise
 *   "max_rows_per_pass"    = 0 (16 Mi) | n          column budget (B*T) of one pass over the workspace;
 *                                                   larger batches run as equal passes
 * (the debug library adds "ablate": see the MDK_DEBUG_HOOKS block at the end of this header) */
int mdk_gru_set_option(mdk_gru *m, const char *key, int value);

/* hipEvent timing of every kernel of the following forwards (adds a stream sync per forward). */
int mdk_gru_enable_timing(mdk_gru *m, int on);
int mdk_gru_get_timing(mdk_gru *m, mdk_gru_timing *out);

/* Chunks, margin and certificate of the last forward (see "scan_split"). */
int mdk_gru_get_split(mdk_gru *m, mdk_gru_split *out);

/* The shape arithmetic of "scan_split" on its own (no device needed): how a batch of B windows of T columns would be
 * split by a process that is one of `gpu_share` on its GPU, with option values `scan_split` (1 auto, n >= 2 forced) and
 * `margin`.  chunks = 1: not split.  Chunk k is columns [start[k], start[k] + columns) of every window and delivers
 * columns [first[k], last[k]); the delivered ranges tile [0, T). */
typedef struct {
    int chunks, columns, margin;
    int start[16], first[16], last[16];
} mdk_split_shape;
int mdk_split_plan(int B, int T, int gpu_share, int scan_split, int margin, mdk_split_shape *out);
/* The margin learner (option "scan_split_adapt") on a model that certifies iff the margin is >= `need` (0: never): n_calls calls
 * from margin `start`; margins[i] = the margin call i was answered at (0: sequentially), forwards[i] = the split forwards it cost.
 * Device-free, for tests and for reasoning about what a model with a known forgetting length will pay. */
int mdk_margin_sim(int start, int adapt, int need, int n_calls, int *margins, int *forwards);

/* How a pass of `windows` windows of T columns would be launched (device-free: the decisions of api.hip's plan_pass, for tests,
 * for `python -m medaka_amd.validate --plan-only` and for sizing): work-group granularity, what is fused, what streams, and
 * whether the pass needs the gi workspace (3 KB per column and direction; the throughput regime and an audit's scan do not).
 *   precision   MDK_PREC_*;  gpu_share  processes sharing the GPU;  host_io  bit 0: x comes from host memory, bit 1: the
 *   probabilities go to host memory;  split_chunks  > 1: the pass is the virtual batch of a split call;
 *   mode  bit 0: the caller looks at the fp16-range flag itself (split calls, host entries), bit 1: `lean` (an audit's scan),
 *         bit 2: the model has met out-of-range input before */
typedef struct mdk_pass_shape {
    int windows_per_group;   /* 4, 8 or 16 */
    int work_groups;         /* per direction */
    int fuse_layer0, fuse_projection, fuse_head, final_head;
    int overlap_gemm, stream_in, stream_out;
    int needs_gi;
} mdk_pass_shape;
int mdk_pass_plan(const mdk_gru_desc *desc, int precision, int gpu_share, int windows, int T, int host_io, int split_chunks,
                  int mode, mdk_pass_shape *out);

/* Device ordinal the model lives on (`TorchModel.device()`, models.py:291-296). */
int mdk_gru_device(const mdk_gru *m);
void mdk_gru_destroy(mdk_gru *m);

/*
 * Replaces `MajorityVoteModel.forward` (medaka/architectures/majority_vote_model.py:37-53):
 * x_dev: n_cols x 10 fp32, probs_dev: n_cols x 5 fp32, both device pointers.
 */
int mdk_majority_forward_dev(const float *x_dev, long n_cols, float *probs_dev, int device,
                             void *stream);
int mdk_majority_forward(const float *x_host, long n_cols, float *probs_host, int device);

/* ------------------------------------------------------------------------------------------
 * Read-level model (reference `LatentSpaceLSTM`, medaka/architectures/latent_space_lstm.py:35-207):
 * Embedding + q-score (+ dwell) -> Conv1d(k=1) -> ReLU -> BN -> Conv1d(k=17) -> ReLU -> BN ->
 * Linear -> masked mean over reads -> 2-layer bi-LSTM or 4 alternating LSTMs -> Linear -> softmax.
 * Supported: cnn_size == 128 with lstm_size == 128 (bi- or 4 x uni-directional) or lstm_size == 384
 * uni-directional (the bundled rl_lstm384); kernel_sizes [1, 17], embedding size 6, alphabet <= 8, 5 classes.
 */
typedef struct mdk_rl mdk_rl;
typedef struct {
    int lstm_size;       /* 128 | 384 */
    int cnn_size;        /* 128 */
    int kernel_size0;    /* 1 */
    int kernel_size1;    /* 17 */
    int use_dwells;      /* 0/1: extra dwell feature, input has 5 bytes per read position */
    int alphabet_size;   /* bases_alphabet_size, 6 */
    int embedding_size;  /* bases_embedding_size, 6 */
    int bidirectional;   /* 1: 2-layer bi-LSTM; 0: reverse-forward-reverse-forward stack */
    int num_classes;     /* 5 */
    int normalise;       /* 1: softmax */
} mdk_rl_desc;

/*
 * `weights`: 34 host fp32 tensors in `state_dict()` order without `num_batches_tracked` and the
 * unused `read_level_conv.expansion_layer.*`:
 *   base_embedder.weight, strand_embedder.weight,
 *   convs.0.{weight,bias}, convs.2.{weight,bias,running_mean,running_var},
 *   convs.3.{weight,bias}, convs.5.{weight,bias,running_mean,running_var},
 *   pre_pool_expansion_layer.{weight,bias},
 *   16 LSTM tensors (weight_ih, weight_hh, bias_ih, bias_hh per layer/direction in state_dict order),
 *   linear.{weight,bias}.
 */
int mdk_rl_create(const mdk_rl_desc *desc, const float *const *weights, int n_weights, int device,
                  mdk_rl **out);
/* Replaces `TorchModel.predict_on_batch` for ReadLevelFeaturesModel input
 * (`batch.read_level_features`, base_classes.py:28-34): x uint8 (B, P, D, F) -> probs (B, P, 5). */
int mdk_rl_forward(mdk_rl *m, const unsigned char *x_host, int B, int P, int D, int F, float *probs_host);
/* Device-resident variant, enqueued on `stream`.  lstm_size 128: asynchronous.  lstm_size 384 (rl_lstm384): the
 * cluster recurrence verifies its cross-CU exchange, so by default the call SYNCHRONISES `stream`, and after a
 * time-out (a late cluster member: the path wants 192 CUs of the GPU to itself) re-runs on the plain schedule.
 * Every try is bounded on the device (the clusters' placement handshake: 50 ms of wall clock); the host retries with
 * a growing pause (20 ... 320 ms) until the forward goes through or "wide_wait_ms" (default 3000) are spent, and only
 * then fails with MDK_ERR_DEVICE: a co-tenant that holds CUs for a few hundred milliseconds is waited out, a GPU that
 * cannot host the kernel is reported after the budget.  Option "wide_async" = 1 makes it asynchronous (no retry);
 * mdk_rl_check() then reports a time-out of any earlier forward. */
int mdk_rl_forward_dev(mdk_rl *m, const unsigned char *x_dev, int B, int P, int D, int F, float *probs_dev,
                       void *stream);
int mdk_rl_check(mdk_rl *m, void *stream);
/* hipEvent timing of the following mdk_rl_forward_dev calls (adds a sync per forward): the fused read-level
 * front end (k_rl_front: embedding + conv1 + BN + conv17 on MFMA + BN + masked mean), the dominant kernel. */
typedef struct {
    float front_ms;      /* k_rl_front */
    float total_ms;      /* whole forward on the stream */
    int wide_retries;    /* lstm_size 384: tries re-run on the plain schedule after an exchange time-out (cumulative) */
} mdk_rl_timing;
int mdk_rl_enable_timing(mdk_rl *m, int on);
int mdk_rl_get_timing(mdk_rl *m, mdk_rl_timing *out);
int mdk_rl_set_precision(mdk_rl *m, int precision);
int mdk_rl_set_normalise(mdk_rl *m, int normalise);
/* Tuning / test knobs (no reference counterpart):
 *   "rec_windows_per_tile" = 0 (auto) | 4 | 8 | 16   lstm_size 128: recurrence work-group granularity
 *   "wide_async"           = 0 | 1                  lstm_size 384: do not synchronise in mdk_rl_forward_dev (see above)
 *   "overlap_gemm"         = 1 | 0                  lstm_size 384: next layer's projection on a side stream
 *                                                   behind resumable recurrence chunks (P >= 1024)
 *   "wide_wait_ms"         = 3000 | 0..60000        lstm_size 384: wall-clock budget of the host's retries after a
 *                                                   cluster time-out (0: the second time-out is the error)
 *   "wide_write_through"   = 0 | 1                  lstm_size 384: always exchange h through write-through
 *                                                   granules, even when a cluster shares one XCD
 *   "wide_groups_per_cluster" = 0 (auto) | 1 | 2     lstm_size 384: 8-window groups interleaved per cluster
 *   "wide_poll_delay"      = 0..64 (default 7)      lstm_size 384, 1 group: 64-clock sleeps before the first poll */
int mdk_rl_set_option(mdk_rl *m, const char *key, int value);
int mdk_rl_device(const mdk_rl *m);
void mdk_rl_destroy(mdk_rl *m);

/* ---- PCIe diet either side of the network (opt-in fast paths) ----------
 * f2: CountsFeatureEncoder(normalise='total') on the device -- reference medaka/features.py:907-911,
 *     feature = (counts / np.maximum(1, depth)).astype(float32); bit-identical (float64 divide, then
 *     float32 round, as numpy).  counts: (n_cols, n_features) uint16 raw pileup counts
 *     (src/medaka_counts.c), depth: (n_cols) uint32, the depth of the parent major column
 *     (features.py:884-885).  22-24 bytes per column cross PCIe instead of 40.
 * f3: argmax decode on the device -- reference medaka/labels.py:1061-1065 (`np.argmax`, first maximum,
 *     a NaN wins as in numpy) and the probability of that class (`np.take_along_axis`): 5 bytes per
 *     column come back instead of 20.  Symbols / phred strings are then made on the host exactly as
 *     labels.py:1066-1085 does. */
int mdk_normalise_counts_dev(const uint16_t *counts_dev, const uint32_t *depth_dev, long n_cols, int n_features,
                             float *x_dev, int device, void *stream);
int mdk_decode_dev(const float *probs_dev, long n_cols, int n_classes, uint8_t *cls_dev, float *pmax_dev,
                   int device, void *stream);
/* Host entry: raw counts + depth in; any of probs (B,T,5) / cls (B,T) / pmax (B,T) out (NULL = not wanted,
 * at least one must be given). */
int mdk_gru_forward_counts(mdk_gru *m, const uint16_t *counts_host, const uint32_t *depth_host, int B, int T,
                           float *probs_host, uint8_t *cls_host, float *pmax_host);
/* As mdk_gru_forward but returning only the decoded classes (x: normalised fp32 features). */
int mdk_gru_forward_decoded(mdk_gru *m, const float *x_host, int B, int T, uint8_t *cls_host, float *pmax_host);

/* Raw device helpers for hosts that do not carry their own HIP runtime binding (bench, tests). */
int mdk_device_count(int *count);
int mdk_device_name(int device, char *buf, size_t buflen);
int mdk_dev_alloc(int device, size_t bytes, void **ptr);
int mdk_dev_free(int device, void *ptr);
/* Page-locked host buffers (hipHostMalloc): as x_host / probs_host of the forward entry points they are
 * copied by DMA directly and never page-fault inside the call.  Any host memory is accepted there;
 * these are for callers that own their buffers (the Python layer takes its output tensors from
 * torch's pinned allocator instead). */
int mdk_host_alloc(size_t bytes, void **ptr);
int mdk_host_free(void *ptr);
/* Batch assembly on the host.  Replaces the `torch.stack([...]).float()` of `Batch.collate`
 * (medaka/torch_ext.py:147-148), which the reference's single "Batcher" thread (prediction.py:356-370)
 * runs for every batch: n_rows equal-sized sample blocks (row i = the contiguous `features` array of sample i,
 * row_bytes each) are copied into the contiguous batch buffer `dst` by n_threads host threads (1..64; rows are
 * dealt out in contiguous runs).  With a page-locked, recycled `dst` nothing is page-faulted: 17 ms -> 2 ms per
 * 200 x 10000 x 10 fp32 batch.  No device is touched; callable without a GPU. */
int mdk_gather_rows(void *dst, const void *const *rows, int n_rows, size_t row_bytes, int n_threads);
int mdk_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes);
int mdk_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes);
int mdk_device_synchronize(int device);

/* MFMA fragment-layout / subnormal self-test run on the device (used by the gpu tests). */
int mdk_selftest_mfma(int device, float *max_abs_err, int *subnormal_preserved);

#ifdef MDK_DEBUG_HOOKS
/* ---- test / profiling hooks: compiled ONLY into the debug library (medaka_amd/libmedaka_amd_debug.so, built with
 * -DMDK_DEBUG_HOOKS by medaka_amd/build.py; `nm -D` of the release library shows none of them).  The debug library also
 * accepts the options "ablate" (timing-only ablation masks of the recurrence kernel, wrong results) and
 * "wide_inject_timeout" (the next n tries of the wide read-level forward find the time-out flag already raised), and the
 * environment variables MDK_ABLATE and MDK_SPLIT_KEEP. */
/* occupy `blocks` CUs with a compute-bound loop of `iters` FMA pairs per lane (synchronous) */
int mdk_selftest_burn(int device, int blocks, int iters);
/* hold `blocks` CUs exclusively (one work-group with `lds_bytes` of LDS each) for `milliseconds` (synchronous) */
int mdk_selftest_hold(int device, int blocks, int milliseconds, int lds_bytes);
/* per-phase cycle counters written by the ablate = 64 build of the recurrence kernel */
int mdk_gru_debug_read(mdk_gru *m, unsigned long long *dst, int n);
#endif

const char *mdk_last_error(void);
const char *mdk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MEDAKA_AMD_H */
