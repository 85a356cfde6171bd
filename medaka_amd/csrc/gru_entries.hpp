// The forward entries of include/medaka_amd.h: device entry, staging and the pipelined staged entry, host entry, counts / decoded.
// Part of api.hip (included there after gru_split.hpp).
#pragma once
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
    // a page-locked result buffer is visible to the device: the last chunks of a split call's result may then leave by kernel
    // behind the last recurrence (Pass::copy_out, k_tail_to_host) -- here only: no other forward runs behind a cold call
    m->tail_host = m->tail_dev = nullptr;
    if (m->opt_tail_blit) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, probs_host) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) {
            m->tail_host = probs_host;
            m->tail_dev = static_cast<float *>(at.devicePointer);
        } else {
            (void)hipGetLastError();       // pageable memory: not an error, the DMA queue takes every chunk
        }
    }
    // x streams in and the probabilities stream out while the recurrences run (forward_pass, HostIO)
    rc = run_forward(m, m->x_dev, B, T, m->p_dev, m->stream, x_host, probs_host);
    m->tail_host = m->tail_dev = nullptr;
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
