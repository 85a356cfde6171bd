// C ABI of the read-level model (reference `LatentSpaceLSTM`, medaka/architectures/
// latent_space_lstm.py): fused read-level front end (rl_front.hpp) -> LSTM stack on the same
// MFMA recurrence / projection kernels as the GRU model (CELL = 1, four gate tiles) -> head.
#include <hip/hip_runtime.h>

#include <chrono>
#include <thread>

#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "gi_proj.hpp"
#include "head.hpp"
#include "host_common.hpp"
#include "layout.hpp"
#include "lstm_wide.hpp"
#include "rec_mfma.hpp"
#include "rl_front.hpp"

using namespace mdk;

#ifndef MDK_PF
#define MDK_PF 5
#endif
#ifndef MDK_WIDE_PF
#define MDK_WIDE_PF 3
#endif

namespace {

struct LstmLayer {
    int K = 0, D = 1, reverse_mask = 0;
    half8 *whh_frag = nullptr;   // [D][8][4][4][2][64]
    half8 *wih_frag = nullptr;   // [D][8][K/32][4][2][64]
    float *bias = nullptr;       // [D][512]  b_ih + b_hh
    float *inv_rec = nullptr, *up_rec = nullptr, *inv_gi = nullptr;   // [D]
    float a_scale = kActScale;   // operand scale of this layer's input activations
};

// one uni-directional LSTM(384) layer of the wide model (lstm_wide.hpp)
struct WideLayer {
    int KS = kWKS, reverse = 0;
    half8 *whh_frag = nullptr;   // [12][8][12][2][64]
    half8 *wih_frag = nullptr;   // [96 tiles][KS][2][64]   permuted gate columns
    float *bias = nullptr;       // [1536] permuted, multiplied by the recurrence scale
    float inv_rec = 1.f, alpha = 1.f, a_scale = 1.f;
};

}  // namespace

struct mdk_rl {
    mdk_rl_desc desc{};
    int device = 0;
    int precision = MDK_PREC_FP32;
    int opt_tile_windows = 0;
    int opt_force_wt = 0;        // wide model: always use write-through granules (test hook)
    int opt_wide_groups = 0;     // wide model: groups per cluster, 0 = auto
    int opt_poll_delay = 7;      // wide model, one group per cluster: 64-clock sleeps before the first poll
    int opt_async = 0;           // wide model: 1 = mdk_rl_forward_dev does not synchronise (no retry; see mdk_rl_check)
    int n_cus = 0;
    int wide_retries = 0;        // forwards that were re-run on the plain schedule after a time-out
    int opt_wait_ms = 3000;      // wide model: wall-clock budget of the host's retries after a time-out ("wide_wait_ms")
    int inject_timeouts = 0;     // test hook ("wide_inject_timeout"): the next n tries find the device flag already up
    bool timing = false;         // hipEvent timing of the front end and of the whole forward (adds a sync)
    hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};   // forward start, front start, front end, forward end
    mdk_rl_timing last{};
    // front end
    float *base_emb = nullptr, *strand_emb = nullptr, *w1 = nullptr, *b1 = nullptr, *a1 = nullptr, *c1 = nullptr;
    half8 *w2frag = nullptr;
    float *b2 = nullptr, *a2 = nullptr, *c2 = nullptr;
    float s1 = 1.f, inv2 = 1.f, s2 = 1.f;   // s2: operand scale of the pooled conv features
    int nf = 7;
    // recurrent stack + head
    bool wide = false;           // lstm_size == 384: cluster recurrence (lstm_wide.hpp)
    std::vector<WideLayer> wlayers;
    unsigned long long *exch = nullptr;
    int *status = nullptr;
    float *gi2 = nullptr, *cstate = nullptr;   // second projection buffer / cell state between chunks
    size_t cstate_cap = 0;
    hipStream_t side = nullptr;               // next layer's projection under this layer's recurrence
    std::vector<hipEvent_t> ov_ev;
    int opt_overlap = 1;
    std::vector<LstmLayer> layers;
    float *lin_w = nullptr, *lin_b = nullptr;
    // workspace
    int *mask = nullptr;
    size_t mask_cap = 0;
    float *gi = nullptr, *act[2] = {nullptr, nullptr};
    size_t ws_rows = 0;
    unsigned char *x_dev = nullptr;
    float *p_dev = nullptr;
    size_t x_cap = 0, p_cap = 0;
    hipStream_t stream = nullptr;
};

extern "C" void mdk_rl_destroy(mdk_rl *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (auto e : m->tev) if (e) (void)hipEventDestroy(e);
    for (void *p : {(void *)m->base_emb, (void *)m->strand_emb, (void *)m->w1, (void *)m->b1, (void *)m->a1,
                    (void *)m->c1, (void *)m->w2frag, (void *)m->b2, (void *)m->a2,
                    (void *)m->c2, (void *)m->lin_w, (void *)m->lin_b, (void *)m->mask,
                    (void *)m->gi, (void *)m->act[0], (void *)m->act[1], (void *)m->x_dev,
                    (void *)m->p_dev})
        free_dev(p);
    for (auto &L : m->layers) {
        free_dev(L.whh_frag); free_dev(L.wih_frag); free_dev(L.bias);
        free_dev(L.inv_rec); free_dev(L.up_rec); free_dev(L.inv_gi);
    }
    for (auto &L : m->wlayers) { free_dev(L.whh_frag); free_dev(L.wih_frag); free_dev(L.bias); }
    free_dev(m->exch); free_dev(m->status); free_dev(m->gi2); free_dev(m->cstate);
    for (auto e : m->ov_ev) (void)hipEventDestroy(e);
    if (m->side) (void)hipStreamDestroy(m->side);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

// fold: the pre-pool Linear (w_pp [128][128], b_pp [128]) when this is the first layer -- mean-pooling
// and the linear layer commute and nothing non-linear separates it from W_ih, so the front end pools
// the conv features and this layer projects them with W_ih W_pp (bias W_ih b_pp + b_ih + b_hh).
static int build_lstm_layer(LstmLayer &Ld, int K, int D, int reverse_mask, const float *const *w,
                            const float *w_pp = nullptr, const float *b_pp = nullptr, float a_scale = kActScale) {
    constexpr int NG = 4, G4 = 4 * kH;
    Ld.K = K; Ld.D = D; Ld.reverse_mask = reverse_mask; Ld.a_scale = a_scale;
    const int KS = K / 32;
    std::vector<half8> whh((size_t)D * 8 * 4 * NG * 2 * 64), wih((size_t)D * 8 * KS * NG * 2 * 64);
    std::vector<float> bias((size_t)D * G4), inv_rec(D), up_rec(D), inv_gi(D), wfold;
    for (int d = 0; d < D; ++d) {
        const float *w_ih = w[4 * d + 0], *w_hh = w[4 * d + 1], *b_ih = w[4 * d + 2], *b_hh = w[4 * d + 3];
        for (int j = 0; j < G4; ++j) bias[(size_t)d * G4 + j] = b_ih[j] + b_hh[j];
        if (w_pp) {
            wfold.assign((size_t)G4 * K, 0.f);
            std::vector<double> row(K);
            for (int j = 0; j < G4; ++j) {
                std::fill(row.begin(), row.end(), 0.0);
                double bj = (double)b_ih[j] + (double)b_hh[j];
                for (int k = 0; k < kH; ++k) {
                    const double a = w_ih[(size_t)j * kH + k];
                    bj += a * b_pp[k];
                    for (int c = 0; c < K; ++c) row[c] += a * w_pp[(size_t)k * K + c];
                }
                for (int c = 0; c < K; ++c) wfold[(size_t)j * K + c] = (float)row[c];
                bias[(size_t)d * G4 + j] = (float)bj;
            }
            w_ih = wfold.data();
        }
        const float sw = pick_scale(w_hh, (size_t)G4 * kH);
        inv_rec[d] = 1.0f / (kActScale * sw);
        up_rec[d] = kActScale * sw;
        const float swi = pick_scale(w_ih, (size_t)G4 * K);
        inv_gi[d] = 1.0f / (a_scale * swi);
        for (int w8 = 0; w8 < 8; ++w8)
            for (int gate = 0; gate < NG; ++gate)
                for (int lane = 0; lane < 64; ++lane) {
                    const int j = gate * kH + 16 * w8 + (lane & 15), gq = lane >> 4;
                    for (int ks = 0; ks < 4; ++ks) {
                        half8 hi, lo;
                        for (int i = 0; i < 8; ++i) {
                            _Float16 a, b;
                            split_host(w_hh[(size_t)j * kH + 32 * ks + 8 * gq + i] * sw, a, b);
                            hi[i] = a; lo[i] = b;
                        }
                        const size_t base = ((((size_t)(d * 8 + w8) * 4 + ks) * NG + gate) * 2) * 64 + lane;
                        whh[base] = hi; whh[base + 64] = lo;
                    }
                    for (int ks = 0; ks < KS; ++ks) {
                        half8 hi, lo;
                        for (int i = 0; i < 8; ++i) {
                            _Float16 a, b;
                            split_host(w_ih[(size_t)j * K + 32 * ks + 8 * gq + i] * swi, a, b);
                            hi[i] = a; lo[i] = b;
                        }
                        const size_t base = ((((size_t)(d * 8 + w8) * KS + ks) * NG + gate) * 2) * 64 + lane;
                        wih[base] = hi; wih[base + 64] = lo;
                    }
                }
    }
    int rc;
    if ((rc = upload(&Ld.whh_frag, whh))) return rc;
    if ((rc = upload(&Ld.wih_frag, wih))) return rc;
    if ((rc = upload(&Ld.bias, bias))) return rc;
    if ((rc = upload(&Ld.inv_rec, inv_rec))) return rc;
    if ((rc = upload(&Ld.up_rec, up_rec))) return rc;
    if ((rc = upload(&Ld.inv_gi, inv_gi))) return rc;
    return MDK_OK;
}

// Wide layer: w_ih is [1536][K] (K = 128 for the folded first layer, 384 after), w_hh [1536][384],
// bias [1536] already summed.  Gate columns are permuted into the order k_lstm_wide's waves own:
// tile nt = member * 8 + wave, column n of the tile = gate (n & 3) of unit 32*member + 4*wave + (n >> 2).
static int build_wide_layer(WideLayer &Ld, int K, const float *w_ih, const float *w_hh, const float *bias,
                            float a_scale, int reverse) {
    Ld.KS = K / 32; Ld.reverse = reverse; Ld.a_scale = a_scale;
    const float sw = pick_scale(w_hh, (size_t)kWG4 * kWH);
    const float swi = pick_scale(w_ih, (size_t)kWG4 * K);
    const float up_rec = kActScale * sw;
    Ld.inv_rec = 1.0f / up_rec;
    Ld.alpha = up_rec / (a_scale * swi);
    std::vector<half8> whh((size_t)96 * kWKS * 2 * 64), wih((size_t)96 * Ld.KS * 2 * 64);
    std::vector<float> bp(kWG4);
    for (int nt = 0; nt < 96; ++nt) {
        const int member = nt / 8, w8 = nt % 8;
        for (int lane = 0; lane < 64; ++lane) {
            const int n = lane & 15, kg = lane >> 4;
            const int j = (n & 3) * kWH + 32 * member + 4 * w8 + (n >> 2);
            if (kg == 0) bp[nt * 16 + n] = bias[j] * up_rec;
            for (int ks = 0; ks < kWKS; ++ks) {
                half8 hi, lo;
                for (int i = 0; i < 8; ++i) {
                    _Float16 a, b;
                    split_host(w_hh[(size_t)j * kWH + 32 * ks + 8 * kg + i] * sw, a, b);
                    hi[i] = a; lo[i] = b;
                }
                const size_t base = (((size_t)nt * kWKS + ks) * 2) * 64 + lane;
                whh[base] = hi; whh[base + 64] = lo;
            }
            for (int ks = 0; ks < Ld.KS; ++ks) {
                half8 hi, lo;
                for (int i = 0; i < 8; ++i) {
                    _Float16 a, b;
                    split_host(w_ih[(size_t)j * K + 32 * ks + 8 * kg + i] * swi, a, b);
                    hi[i] = a; lo[i] = b;
                }
                const size_t base = (((size_t)nt * Ld.KS + ks) * 2) * 64 + lane;
                wih[base] = hi; wih[base + 64] = lo;
            }
        }
    }
    int rc;
    if ((rc = upload(&Ld.whh_frag, whh))) return rc;
    if ((rc = upload(&Ld.wih_frag, wih))) return rc;
    if ((rc = upload(&Ld.bias, bp))) return rc;
    return MDK_OK;
}

extern "C" int mdk_rl_create(const mdk_rl_desc *desc, const float *const *w, int n_weights, int device,
                             mdk_rl **out) {
    if (!desc || !w || !out) return fail(MDK_ERR_ARG, "null argument");
    *out = nullptr;
    const bool wide = desc->lstm_size == kWH;
    if ((desc->lstm_size != kH && !wide) || desc->cnn_size != kRlC)
        return fail(MDK_ERR_ARG, "unsupported lstm_size %d / cnn_size %d (engine supports 128 or 384 / 128)",
                    desc->lstm_size, desc->cnn_size);
    if (wide && desc->bidirectional)
        return fail(MDK_ERR_ARG, "lstm_size 384 is supported as the 4 x uni-directional stack only (bidirectional=False)");
    if (desc->kernel_size0 != 1 || desc->kernel_size1 != kRlTaps)
        return fail(MDK_ERR_ARG, "unsupported kernel_sizes [%d, %d] (engine supports [1, 17])",
                    desc->kernel_size0, desc->kernel_size1);
    if (desc->embedding_size != 6 || desc->alphabet_size < 1 || desc->alphabet_size > 8)
        return fail(MDK_ERR_ARG, "unsupported embedding %d x %d", desc->alphabet_size, desc->embedding_size);
    if (desc->num_classes != 5) return fail(MDK_ERR_ARG, "unsupported num_classes %d", desc->num_classes);
    if (n_weights != 34) return fail(MDK_ERR_ARG, "expected 34 weight tensors, got %d", n_weights);
    for (int i = 0; i < n_weights; ++i)
        if (!w[i]) return fail(MDK_ERR_ARG, "weight tensor %d is null", i);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(MDK_ERR_DEVICE, "device %d not available (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    mdk_rl *m = new mdk_rl();
    m->desc = *desc;
    m->device = device;
    m->wide = wide;
    int rc = MDK_OK;
    auto bail = [&](int code) { mdk_rl_destroy(m); return code; };
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(MDK_ERR_DEVICE, "hipStreamCreate failed"));
    const int nf = 6 + 1 + (desc->use_dwells ? 1 : 0);
    m->nf = nf;
    const int A = desc->alphabet_size;
    const float eps = 1e-5f;

    // ---- front end: embeddings, conv1 (rows padded to 8 features), BN folded to a*x + c
    {
        std::vector<float> be(w[0], w[0] + A * 6), se(w[1], w[1] + 18), w1(128 * 8, 0.f);
        for (int c = 0; c < 128; ++c)
            for (int f = 0; f < nf; ++f) w1[c * 8 + f] = w[2][(size_t)c * nf + f];
        std::vector<float> b1(w[3], w[3] + 128), a1(128), c1(128), b2(w[9], w[9] + 128), a2(128), c2(128);
        for (int c = 0; c < 128; ++c) {
            a1[c] = w[4][c] / std::sqrt(w[7][c] + eps);
            c1[c] = w[5][c] - w[6][c] * a1[c];
            a2[c] = w[10][c] / std::sqrt(w[13][c] + eps);
            c2[c] = w[11][c] - w[12][c] * a2[c];
        }
        // analytic bounds -> operand scales.  |feature| <= max|emb_b| + max|emb_s|, 9.2 (q), 255 (dwell)
        float eb = 0.f, es = 0.f;
        for (float v : be) eb = std::max(eb, std::fabs(v));
        for (float v : se) es = std::max(es, std::fabs(v));
        float y1max = 0.f;
        for (int c = 0; c < 128; ++c) {
            float bound = std::fabs(b1[c]);
            for (int f = 0; f < nf; ++f) {
                const float fm = f < 6 ? (eb + es) : (f == 6 ? 9.2f : 255.f);
                bound += std::fabs(w1[c * 8 + f]) * fm;
            }
            y1max = std::max(y1max, std::max(std::fabs(c1[c]), std::fabs(a1[c] * bound + c1[c])));
        }
        m->s1 = pick_scale_max(y1max);
        const float sw2 = pick_scale(w[8], (size_t)128 * 128 * kRlTaps);
        m->inv2 = 1.0f / (m->s1 * sw2);
        float y2max = 0.f;
        for (int co = 0; co < 128; ++co) {
            float bound = std::fabs(b2[co]);
            for (size_t i = 0; i < (size_t)128 * kRlTaps; ++i) bound += std::fabs(w[8][(size_t)co * 128 * kRlTaps + i]) * y1max;
            y2max = std::max(y2max, std::max(std::fabs(c2[co]), std::fabs(a2[co] * bound + c2[co])));
        }
        m->s2 = pick_scale_max(y2max);
        // conv2 B-fragments [17][4 kb][4 waves][2 nt][2][64]: W2[co][ci][tau]
        std::vector<half8> w2f((size_t)kRlTaps * 4 * 4 * 2 * 2 * 64);
        for (int tau = 0; tau < kRlTaps; ++tau)
            for (int kb = 0; kb < 4; ++kb)
                for (int wv = 0; wv < 4; ++wv)
                    for (int nt = 0; nt < 2; ++nt)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = 32 * wv + 16 * nt + (lane & 15), gq = lane >> 4;
                            half8 hi, lo;
                            for (int i = 0; i < 8; ++i) {
                                const int ci = 32 * kb + 8 * gq + i;
                                _Float16 a, b;
                                split_host(w[8][((size_t)co * 128 + ci) * kRlTaps + tau] * sw2, a, b);
                                hi[i] = a; lo[i] = b;
                            }
                            const size_t base = (((((size_t)tau * 4 + kb) * 4 + wv) * 2 + nt) * 2) * 64 + lane;
                            w2f[base] = hi; w2f[base + 64] = lo;
                        }
        if ((rc = upload(&m->base_emb, be)) || (rc = upload(&m->strand_emb, se)) || (rc = upload(&m->w1, w1)) ||
            (rc = upload(&m->b1, b1)) || (rc = upload(&m->a1, a1)) || (rc = upload(&m->c1, c1)) ||
            (rc = upload(&m->w2frag, w2f)) || (rc = upload(&m->b2, b2)) || (rc = upload(&m->a2, a2)) ||
            (rc = upload(&m->c2, c2)))
            return bail(rc);
    }
    // ---- LSTM stack
    if (wide) {
        // mean-pooling and Linear(128 -> 384) commute and no non-linearity separates the linear layer
        // from the first LSTM projection: W' = W_ih0 W_pp, b' = W_ih0 b_pp + b_ih0 + b_hh0 (in double)
        const float *w_pp = w[14], *b_pp = w[15];
        std::vector<float> w0((size_t)kWG4 * 128), bias(kWG4);
        {
            std::vector<double> row(128);
            for (int j = 0; j < kWG4; ++j) {
                std::fill(row.begin(), row.end(), 0.0);
                double bj = (double)w[18][j] + (double)w[19][j];
                for (int k = 0; k < kWH; ++k) {
                    const double a = w[16][(size_t)j * kWH + k];
                    bj += a * b_pp[k];
                    const float *wr = w_pp + (size_t)k * 128;
                    for (int c = 0; c < 128; ++c) row[c] += a * wr[c];
                }
                for (int c = 0; c < 128; ++c) w0[(size_t)j * 128 + c] = (float)row[c];
                bias[j] = (float)bj;
            }
        }
        m->wlayers.resize(4);   // reverse - forward - reverse - forward (latent_space_lstm.py:141-149)
        if ((rc = build_wide_layer(m->wlayers[0], 128, w0.data(), w[17], bias.data(), m->s2, 1))) return bail(rc);
        for (int i = 1; i < 4; ++i) {
            for (int j = 0; j < kWG4; ++j) bias[j] = w[16 + 4 * i + 2][j] + w[16 + 4 * i + 3][j];
            if ((rc = build_wide_layer(m->wlayers[i], kWH, w[16 + 4 * i], w[16 + 4 * i + 1], bias.data(), kActScale,
                                       (i % 2 == 0) ? 1 : 0)))
                return bail(rc);
        }
        HIP_TRY(hipMalloc((void **)&m->exch, kWExchWords * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc((void **)&m->status, 64));
        HIP_TRY(hipMemset(m->status, 0, 64));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_rows<12, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 12 * 4 * kWGemmBlk));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_rows<12, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 12 * 4 * kWGemmBlk));
    } else if (desc->bidirectional) {
        m->layers.resize(2);
        if ((rc = build_lstm_layer(m->layers[0], 128, 2, 2, w + 16, w[14], w[15], m->s2))) return bail(rc);
        if ((rc = build_lstm_layer(m->layers[1], 256, 2, 2, w + 24))) return bail(rc);
    } else {
        m->layers.resize(4);   // reverse - forward - reverse - forward (latent_space_lstm.py:141-149)
        for (int i = 0; i < 4; ++i)
            if ((rc = build_lstm_layer(m->layers[i], 128, 1, (i % 2 == 0) ? 1 : 0, w + 16 + 4 * i,
                                       i == 0 ? w[14] : nullptr, i == 0 ? w[15] : nullptr, i == 0 ? m->s2 : kActScale)))
                return bail(rc);
    }
    {
        const int Dl = desc->bidirectional ? 2 : 1;
        std::vector<float> lw(w[32], w[32] + (size_t)5 * Dl * desc->lstm_size), lb(w[33], w[33] + 5);
        if ((rc = upload(&m->lin_w, lw)) || (rc = upload(&m->lin_b, lb))) return bail(rc);
    }
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gi_gemm<8, false, 4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kGemmMT * 8 * 64 * 16));
    *out = m;
    return MDK_OK;
}

extern "C" int mdk_rl_set_precision(mdk_rl *m, int precision) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if (precision != MDK_PREC_FP32 && precision != MDK_PREC_FP16) return fail(MDK_ERR_ARG, "bad precision %d", precision);
    m->precision = precision;
    return MDK_OK;
}
extern "C" int mdk_rl_set_normalise(mdk_rl *m, int normalise) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    m->desc.normalise = normalise ? 1 : 0;
    return MDK_OK;
}
extern "C" int mdk_rl_set_option(mdk_rl *m, const char *key, int value) {
    if (!m || !key) return fail(MDK_ERR_ARG, "null argument");
    if (!strcmp(key, "rec_windows_per_tile")) {
        if (value != 0 && value != 4 && value != 8 && value != 16)
            return fail(MDK_ERR_ARG, "rec_windows_per_tile must be 0, 4, 8 or 16 (16: half precision only)");
        m->opt_tile_windows = value;
    } else if (!strcmp(key, "wide_async")) {
        m->opt_async = value ? 1 : 0;
    } else if (!strcmp(key, "overlap_gemm")) {
        m->opt_overlap = value ? 1 : 0;
    } else if (!strcmp(key, "wide_wait_ms")) {
        if (value < 0 || value > 60000) return fail(MDK_ERR_ARG, "wide_wait_ms must be 0..60000");
        m->opt_wait_ms = value;
#ifdef MDK_DEBUG_HOOKS
    } else if (!strcmp(key, "wide_inject_timeout")) {
        if (value < 0 || value > 1000) return fail(MDK_ERR_ARG, "wide_inject_timeout must be 0..1000");
        m->inject_timeouts = value;
#endif
    } else if (!strcmp(key, "wide_write_through")) {
        m->opt_force_wt = value ? 1 : 0;
    } else if (!strcmp(key, "wide_poll_delay")) {
        if (value < 0 || value > 64) return fail(MDK_ERR_ARG, "wide_poll_delay must be 0..64");
        m->opt_poll_delay = value;
    } else if (!strcmp(key, "wide_groups_per_cluster")) {
        if (value < 0 || value > 2) return fail(MDK_ERR_ARG, "wide_groups_per_cluster must be 0 (auto), 1 or 2");
        m->opt_wide_groups = value;
    } else {
        return fail(MDK_ERR_ARG, "unknown option '%s'", key);
    }
    return MDK_OK;
}
extern "C" int mdk_rl_device(const mdk_rl *m) { return m ? m->device : -1; }

static int rl_workspace(mdk_rl *m, int B, int Dp, size_t rows) {
    if ((size_t)B * Dp > m->mask_cap) {
        free_dev(m->mask); m->mask = nullptr; m->mask_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->mask, (size_t)B * Dp * sizeof(int)));
        m->mask_cap = (size_t)B * Dp;
    }
    if (rows > m->ws_rows) {
        free_dev(m->gi); free_dev(m->act[0]); free_dev(m->act[1]);
        m->gi = m->act[0] = m->act[1] = nullptr; m->ws_rows = 0;
        HIP_TRY(hipMalloc((void **)&m->gi, (size_t)2 * rows * 4 * kH * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&m->act[0], rows * 2 * kH * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&m->act[1], rows * 2 * kH * sizeof(float)));
        m->ws_rows = rows;
    }
    return MDK_OK;
}

// lstm_size = 384: front end (pool-only) -> 4 x [k_gemm_rows -> k_lstm_wide] -> head, natural layouts
static int rl_forward_wide_once(mdk_rl *m, const unsigned char *x_dev, int B, int P, int Dp, int F,
                                float *probs_dev, hipStream_t s, bool allow_overlap, int *timed_out) {
    const size_t rows = (size_t)B * P;
    *timed_out = 0;
    if ((size_t)B * Dp > m->mask_cap) {
        free_dev(m->mask); m->mask = nullptr; m->mask_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->mask, (size_t)B * Dp * sizeof(int)));
        m->mask_cap = (size_t)B * Dp;
    }
    if (rows > m->ws_rows) {
        free_dev(m->gi); free_dev(m->act[0]); free_dev(m->act[1]);
        m->gi = m->act[0] = m->act[1] = nullptr; m->ws_rows = 0;
        free_dev(m->gi2); m->gi2 = nullptr;
        HIP_TRY(hipMalloc((void **)&m->gi, rows * kWG4 * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&m->gi2, rows * kWG4 * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&m->act[0], rows * kWH * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&m->act[1], rows * kWH * sizeof(float)));
        m->ws_rows = rows;
    }
    if ((size_t)B * kWH > m->cstate_cap) {
        free_dev(m->cstate); m->cstate = nullptr; m->cstate_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->cstate, (size_t)B * kWH * sizeof(float)));
        m->cstate_cap = (size_t)B * kWH;
    }
    if (!m->side) {
        // lowest priority: a projection GEMM that becomes ready together with the next recurrence chunk must
        // not take CUs before all 12 members of every cluster are resident (they spin on each other)
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, lo));
    }
    constexpr int kChunks = 8;
    while (m->ov_ev.size() < kChunks + 1) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m->ov_ev.push_back(e);
    }
    HIP_TRY(hipMemsetAsync(m->mask, 0, (size_t)B * Dp * sizeof(int), s));
    hipLaunchKernelGGL(k_rl_mask, dim3((P + 255) / 256, B), dim3(256), 0, s, x_dev, P, Dp, F, m->mask);
    RlFrontArgs fa;
    fa.x = x_dev; fa.mask = m->mask; fa.base_emb = m->base_emb; fa.strand_emb = m->strand_emb;
    fa.w1 = m->w1; fa.b1 = m->b1; fa.a1 = m->a1; fa.c1 = m->c1; fa.w2frag = m->w2frag; fa.b2 = m->b2; fa.a2 = m->a2;
    fa.c2 = m->c2; fa.pooled = m->act[1];   // (B, P, 128) rows
    fa.B = B; fa.P = P; fa.Dp = Dp; fa.F = F; fa.nf = m->nf; fa.n_alpha = m->desc.alphabet_size;
    fa.s1 = m->s1; fa.inv2 = m->inv2;
    const bool hp = (m->precision == MDK_PREC_FP16);
    if (m->timing) HIP_TRY(hipEventRecord(m->tev[1], s));
    if (hp) hipLaunchKernelGGL((k_rl_front<false, true>), dim3((P + kRlPos - 1) / kRlPos, B), dim3(256), 0, s, fa);
    else hipLaunchKernelGGL((k_rl_front<false, false>), dim3((P + kRlPos - 1) / kRlPos, B), dim3(256), 0, s, fa);
    if (m->timing) HIP_TRY(hipEventRecord(m->tev[2], s));

    // groups of 8 windows (16 in half precision); up to 16 groups: one per cluster; more: two interleaved
    const int gw = hp ? 2 * kWWin : kWWin;
    const int n_groups = (B + gw - 1) / gw;
    int ngrp = n_groups > kWMaxClusters ? 2 : 1;
    if (m->opt_wide_groups == 1 || m->opt_wide_groups == 2) ngrp = m->opt_wide_groups;
    const int n_units = (n_groups + ngrp - 1) / ngrp;
    const int n_clusters = std::min(n_units, kWMaxClusters);
    const unsigned rec_grid = 8u * kWC * (unsigned)((n_clusters + 7) / 8);
    // The stack is uni-directional, so the next layer's projection of column t only needs this
    // layer's h_t: each layer's recurrence runs as kChunks resumable launches and, behind every
    // chunk, the next layer's k_gemm_rows for the columns just produced runs on a side stream, on the
    // >= 64 CUs the clusters never occupy.  (The next recurrence scans the other way and still has
    // to wait for the whole layer.)  Projections alternate between two gi buffers.
    // co-residency: one 512-thread work-group per CU (launch bounds), so the grid must fit the CUs
    if (m->n_cus == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, m->device));
        m->n_cus = prop.multiProcessorCount;
    }
    if ((int)rec_grid > m->n_cus)
        return fail(MDK_ERR_DEVICE, "the LSTM(384) cluster recurrence needs %u co-resident work-groups, the device has %d CUs",
                    rec_grid, m->n_cus);
    const bool ovl = allow_overlap && m->opt_overlap && P >= 1024;
    const int n_chunks = ovl ? kChunks : 1;
    auto launch_gemm = [&](const WideLayer &Lg, const float *src, float *gi_out, hipStream_t st, int t_begin, int t_len) {
        if (t_len <= 0) return;
        const dim3 grid((unsigned)((t_len + kWGemmRows - 1) / kWGemmRows), (unsigned)B);
#define MDK_WGEMM(KSV, HPF)                                                                              \
    hipLaunchKernelGGL((k_gemm_rows<KSV, HPF>), grid, dim3(512), (size_t)2 * KSV * 4 * kWGemmBlk, st, src, \
                       Lg.wih_frag, Lg.bias, gi_out, P, t_begin, t_len, Lg.a_scale, Lg.alpha)
        if (Lg.KS == 4) { if (hp) MDK_WGEMM(4, true); else MDK_WGEMM(4, false); }
        else { if (hp) MDK_WGEMM(12, true); else MDK_WGEMM(12, false); }
#undef MDK_WGEMM
    };
    const float *in = m->act[1];
    launch_gemm(m->wlayers[0], in, m->gi, s, 0, P);
    for (size_t l = 0; l < m->wlayers.size(); ++l) {
        const WideLayer &Ld = m->wlayers[l];
        float *outp = m->act[l & 1];
        const float *gi_cur = (l & 1) ? m->gi2 : m->gi;
        float *gi_next = (l & 1) ? m->gi : m->gi2;
        const bool has_next = l + 1 < m->wlayers.size();
        for (int j = 0; j < n_chunks; ++j) {
            const int s0 = (int)((long)P * j / n_chunks) / 8 * 8;
            const int s1 = (j + 1 == n_chunks) ? P : (int)((long)P * (j + 1) / n_chunks) / 8 * 8;
            HIP_TRY(hipMemsetAsync(m->exch, 0, kWExchWords * sizeof(unsigned long long), s));
#define MDK_WIDE_N(NG, HPF, ABLV)                                                                        \
    hipLaunchKernelGGL((k_lstm_wide<MDK_WIDE_PF, NG, HPF, ABLV>), dim3(rec_grid), dim3(512), 0, s, gi_cur, Ld.whh_frag, \
                       outp, m->exch, m->status, B, P, Ld.reverse, Ld.inv_rec, n_clusters, n_units, m->opt_force_wt, \
                       hp ? 2 * m->opt_poll_delay : m->opt_poll_delay, s0, s1 - s0, m->cstate, m->opt_async ? 0 : 1)   /* half: fewer MFMAs, later stores */
#define MDK_WIDE(ABLV)                                                                                   \
    do {                                                                                                 \
        if (hp) { if (ngrp == 2) MDK_WIDE_N(2, true, ABLV); else MDK_WIDE_N(1, true, ABLV); }            \
        else { if (ngrp == 2) MDK_WIDE_N(2, false, ABLV); else MDK_WIDE_N(1, false, ABLV); }             \
    } while (0)
#ifdef MDK_WIDE_ABLATE   // timing experiments only (profiles/): MDK_WIDE_ABL selects a garbage-result variant
            switch (getenv("MDK_WIDE_ABL") ? atoi(getenv("MDK_WIDE_ABL")) : 0) {
                case 1: MDK_WIDE(1); break;
                case 2: MDK_WIDE(2); break;
                case 4: MDK_WIDE(4); break;
                case 8: MDK_WIDE(8); break;
                default: MDK_WIDE(0);
            }
#else
            MDK_WIDE(0);
#endif
#undef MDK_WIDE
#undef MDK_WIDE_N
            if (!has_next) continue;
            const int t_begin = Ld.reverse ? P - s1 : s0;
            if (ovl) {
                HIP_TRY(hipEventRecord(m->ov_ev[1 + j], s));
                HIP_TRY(hipStreamWaitEvent(m->side, m->ov_ev[1 + j], 0));
                launch_gemm(m->wlayers[l + 1], outp, gi_next, m->side, t_begin, s1 - s0);
            } else {
                launch_gemm(m->wlayers[l + 1], outp, gi_next, s, t_begin, s1 - s0);
            }
        }
        if (has_next && ovl) {
            HIP_TRY(hipEventRecord(m->ov_ev[0], m->side));
            HIP_TRY(hipStreamWaitEvent(s, m->ov_ev[0], 0));
        }
        in = outp;
    }
    {
        const long blocks = std::min<long>(((long)rows + 15) / 16, 256 * 8);
        hipLaunchKernelGGL(k_linear_softmax<6>, dim3((unsigned)blocks), dim3(256), 0, s, in, m->lin_w, m->lin_b,
                           probs_dev, (long)rows, m->desc.normalise);
    }
    HIP_TRY(hipGetLastError());
    if (m->opt_async) return MDK_OK;        // status is read by the next call or by mdk_rl_check()
    // the cluster recurrence spins across work-groups with bounded waits: read the time-out flag
    int st = 0;
    HIP_TRY(hipMemcpyAsync(&st, m->status, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (st != 0) {
        HIP_TRY(hipMemsetAsync(m->status, 0, sizeof(int), s));
        *timed_out = 1;
    }
    return MDK_OK;
}

// The cluster exchange needs every member of a cluster on a CU at the same time.  A time-out (a late
// member: another tenant on the GPU, or the side-stream projection competing for CUs) is not an error
// yet: the layer stack is run once more on the plain schedule -- nothing on the side stream, one launch
// per layer.  If that times out as well somebody else is holding CUs: every try is bounded on the device (50 ms of
// wall clock in the placement handshake, every later launch of a lost forward returns at once), so the HOST keeps
// retrying the plain schedule with a growing pause -- 20, 40, ... 320 ms -- until the forward goes through or
// "wide_wait_ms" (3 s by default) of wall clock are spent; only then MDK_ERR_DEVICE.  A co-tenant or a long foreign
// kernel that holds the CUs for a few hundred milliseconds is waited out; a GPU that cannot host the kernel at all
// is reported after the budget, not after minutes of spinning and never with a wrong result.
static int rl_forward_wide(mdk_rl *m, const unsigned char *x_dev, int B, int P, int Dp, int F,
                           float *probs_dev, hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    auto inject = [&]() -> int {          // test hook (debug builds: option "wide_inject_timeout"; never armed otherwise): raise the device flag before the try
        if (m->inject_timeouts <= 0 || !m->status) return MDK_OK;
        m->inject_timeouts--;
        const int one = 1;
        HIP_TRY(hipMemcpyAsync(m->status, &one, sizeof(int), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        return MDK_OK;
    };
    int timed_out = 0;
    int rc = inject();
    if (rc) return rc;
    rc = rl_forward_wide_once(m, x_dev, B, P, Dp, F, probs_dev, s, true, &timed_out);
    if (rc || !timed_out) return rc;
    int pause_ms = 0, tries = 1;
    for (;;) {
        m->wide_retries++;
        tries++;
        if ((rc = inject())) return rc;
        timed_out = 0;
        rc = rl_forward_wide_once(m, x_dev, B, P, Dp, F, probs_dev, s, false, &timed_out);
        if (rc || !timed_out) return rc;
        const long spent = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        pause_ms = pause_ms ? std::min(2 * pause_ms, 320) : 20;
        if (spent + pause_ms > m->opt_wait_ms)
            return fail(MDK_ERR_DEVICE, "LSTM(384) cluster exchange timed out %d times in %ld ms (the later tries without the "
                                        "overlapped projection): the rl_lstm384 path needs %d CUs of the GPU to itself",
                        tries, spent, 8 * kWC * 2);
        std::this_thread::sleep_for(std::chrono::milliseconds(pause_ms));
    }
}

// asynchronous mode ("wide_async" = 1): surfaces a time-out of an earlier mdk_rl_forward_dev
extern "C" int mdk_rl_check(mdk_rl *m, void *stream) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if (!m->wide || !m->status) return MDK_OK;
    HIP_TRY(hipSetDevice(m->device));
    int st = 0;
    HIP_TRY(hipMemcpyAsync(&st, m->status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (st != 0) {
        HIP_TRY(hipMemsetAsync(m->status, 0, sizeof(int), (hipStream_t)stream));
        return fail(MDK_ERR_DEVICE, "LSTM(384) cluster exchange timed out in an earlier asynchronous forward");
    }
    return MDK_OK;
}

static int rl_forward_dev_inner(mdk_rl *m, const unsigned char *x_dev, int B, int P, int Dp, int F,
                                float *probs_dev, void *stream);

extern "C" int mdk_rl_enable_timing(mdk_rl *m, int on) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    HIP_TRY(hipSetDevice(m->device));
    if (on && !m->tev[0])
        for (auto &e : m->tev) HIP_TRY(hipEventCreate(&e));
    m->timing = on != 0;
    return MDK_OK;
}
extern "C" int mdk_rl_get_timing(mdk_rl *m, mdk_rl_timing *out) {
    if (!m || !out) return fail(MDK_ERR_ARG, "null argument");
    *out = m->last;
    out->wide_retries = m->wide_retries;
    return MDK_OK;
}

extern "C" int mdk_rl_forward_dev(mdk_rl *m, const unsigned char *x_dev, int B, int P, int Dp, int F,
                                  float *probs_dev, void *stream) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if (!m->timing || B <= 0 || P <= 0) return rl_forward_dev_inner(m, x_dev, B, P, Dp, F, probs_dev, stream);
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipEventRecord(m->tev[0], s));
    int rc = rl_forward_dev_inner(m, x_dev, B, P, Dp, F, probs_dev, stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(m->tev[3], s));
    HIP_TRY(hipEventSynchronize(m->tev[3]));
    HIP_TRY(hipEventElapsedTime(&m->last.front_ms, m->tev[1], m->tev[2]));
    HIP_TRY(hipEventElapsedTime(&m->last.total_ms, m->tev[0], m->tev[3]));
    return MDK_OK;
}

static int rl_forward_dev_inner(mdk_rl *m, const unsigned char *x_dev, int B, int P, int Dp, int F,
                                float *probs_dev, void *stream) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if (B < 0 || P < 0 || Dp < 0) return fail(MDK_ERR_ARG, "negative shape");
    if (B == 0 || P == 0) return MDK_OK;
    if (!x_dev || !probs_dev) return fail(MDK_ERR_ARG, "null buffer");
    if (Dp < 1 || Dp > 256) return fail(MDK_ERR_ARG, "read depth %d outside 1..256", Dp);
    const int need_f = m->desc.use_dwells ? 5 : 4;
    if (F < need_f) return fail(MDK_ERR_ARG, "expected >= %d features per read position, got %d", need_f, F);
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    if (m->wide) return rl_forward_wide(m, x_dev, B, P, Dp, F, probs_dev, s);
    const int T = P, n_tiles = (B + kTileWin - 1) / kTileWin;
    const size_t rows = (size_t)n_tiles * kTileWin * T;
    int rc = rl_workspace(m, B, Dp, rows);
    if (rc) return rc;
    const bool hp = (m->precision == MDK_PREC_FP16);

    // ---- front end -> pooled (act[0], tile-major, one "direction")
    HIP_TRY(hipMemsetAsync(m->act[0], 0, rows * kH * sizeof(float), s));   // padding windows of the last tile
    HIP_TRY(hipMemsetAsync(m->mask, 0, (size_t)B * Dp * sizeof(int), s));
    hipLaunchKernelGGL(k_rl_mask, dim3((P + 255) / 256, B), dim3(256), 0, s, x_dev, P, Dp, F, m->mask);
    RlFrontArgs fa;
    fa.x = x_dev; fa.mask = m->mask; fa.base_emb = m->base_emb; fa.strand_emb = m->strand_emb;
    fa.w1 = m->w1; fa.b1 = m->b1; fa.a1 = m->a1; fa.c1 = m->c1; fa.w2frag = m->w2frag; fa.b2 = m->b2; fa.a2 = m->a2;
    fa.c2 = m->c2; fa.pooled = m->act[0];
    fa.B = B; fa.P = P; fa.Dp = Dp; fa.F = F; fa.nf = m->nf; fa.n_alpha = m->desc.alphabet_size;
    fa.s1 = m->s1; fa.inv2 = m->inv2;
    if (m->timing) HIP_TRY(hipEventRecord(m->tev[1], s));
    if (hp) hipLaunchKernelGGL((k_rl_front<true, true>), dim3((P + kRlPos - 1) / kRlPos, B), dim3(256), 0, s, fa);
    else hipLaunchKernelGGL((k_rl_front<true, false>), dim3((P + kRlPos - 1) / kRlPos, B), dim3(256), 0, s, fa);
    if (m->timing) HIP_TRY(hipEventRecord(m->tev[2], s));

    // ---- LSTM stack
    const int n_win = n_tiles * kTileWin;
    const float *in = m->act[0];
    int din = 1;
    for (size_t l = 0; l < m->layers.size(); ++l) {
        const LstmLayer &Ld = m->layers[l];
        float *outp = m->act[(l + 1) & 1];
        const int D = Ld.D;
        int nq = 1;
        while (nq < (hp ? 4 : 2) && ((n_win + 4 * nq - 1) / (4 * nq)) * D > 256) nq *= 2;
        if (m->opt_tile_windows == 4) nq = 1;
        if (m->opt_tile_windows == 8) nq = 2;
        if (m->opt_tile_windows == 16 && hp) nq = 4;
        const dim3 rgrid((n_win + 4 * nq - 1) / (4 * nq), D);
        const dim3 ggrid(((T + kGemmSteps - 1) / kGemmSteps) * n_tiles);
#define MDK_GEMM(KS, HPF)                                                                          \
    hipLaunchKernelGGL((k_gi_gemm<KS, HPF, 4>), ggrid, dim3(512), (size_t)2 * kGemmMT * KS * 64 * sizeof(half8), s, \
                       in, Ld.wih_frag, Ld.bias, m->gi, n_tiles, T, D, Ld.inv_gi, Ld.up_rec, Ld.a_scale, 0, (const int *)nullptr, 0, T)
        if (din == 2) { if (hp) MDK_GEMM(8, true); else MDK_GEMM(8, false); }
        else { if (hp) MDK_GEMM(4, true); else MDK_GEMM(4, false); }
#undef MDK_GEMM
#define MDK_REC(NQV, HPF)                                                                          \
    hipLaunchKernelGGL((k_rec_mfma<MDK_PF, NQV, false, HPF, 1>), rgrid, dim3(512), 0, s, m->gi,     \
                       (const half8 *)nullptr, (const half8 *)nullptr, Ld.whh_frag, Ld.bias, outp, n_tiles, T, D, \
                       Ld.inv_rec, Ld.reverse_mask, (const int *)nullptr, 0, 0, T)
        if (hp) { if (nq == 1) MDK_REC(1, true); else if (nq == 2) MDK_REC(2, true); else MDK_REC(4, true); }
        else { if (nq == 1) MDK_REC(1, false); else MDK_REC(2, false); }
#undef MDK_REC
        in = outp;
        din = D;
    }
    // ---- head
    {
        const long n_blocks = (long)n_tiles * T;
        const long blocks = std::min<long>((n_blocks + 3) / 4, 256 * 8);
        if (din == 2)
            hipLaunchKernelGGL(k_head_tiled<2>, dim3((unsigned)blocks), dim3(256), 0, s, in, m->lin_w, m->lin_b,
                               probs_dev, B, T, n_tiles, m->desc.normalise, 0, T, SplitPlan{});
        else
            hipLaunchKernelGGL(k_head_tiled<1>, dim3((unsigned)blocks), dim3(256), 0, s, in, m->lin_w, m->lin_b,
                               probs_dev, B, T, n_tiles, m->desc.normalise, 0, T, SplitPlan{});
    }
    HIP_TRY(hipGetLastError());
    return MDK_OK;
}

extern "C" int mdk_rl_forward(mdk_rl *m, const unsigned char *x_host, int B, int P, int Dp, int F,
                              float *probs_host) {
    if (!m) return fail(MDK_ERR_ARG, "null model");
    if (B < 0 || P < 0 || Dp < 0) return fail(MDK_ERR_ARG, "negative shape");
    if (B == 0 || P == 0) return MDK_OK;
    if (!x_host || !probs_host) return fail(MDK_ERR_ARG, "null buffer");
    HIP_TRY(hipSetDevice(m->device));
    const size_t nx = (size_t)B * P * Dp * F, np = (size_t)B * P * 5;
    if (nx > m->x_cap) {
        free_dev(m->x_dev); m->x_dev = nullptr; m->x_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->x_dev, nx));
        m->x_cap = nx;
    }
    if (np > m->p_cap) {
        free_dev(m->p_dev); m->p_dev = nullptr; m->p_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->p_dev, np * sizeof(float)));
        m->p_cap = np;
    }
    HIP_TRY(hipMemcpyAsync(m->x_dev, x_host, nx, hipMemcpyHostToDevice, m->stream));
    int rc = mdk_rl_forward_dev(m, m->x_dev, B, P, Dp, F, m->p_dev, m->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(probs_host, m->p_dev, np * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return MDK_OK;
}
