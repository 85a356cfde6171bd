/*
 * oracle/gru_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C fp32 restatement of the reference consensus network forward pass
 *     medaka/architectures/gru.py:58-72   (GRUModel.forward: nn.GRU -> nn.Linear -> softmax)
 *     medaka/architectures/gru.py:46-56   (2-layer bidirectional GRU, batch_first, Linear(2H -> 5))
 *     medaka/architectures/majority_vote_model.py:37-53 (MajorityVoteModel.forward)
 * The arithmetic of nn.GRU itself lives in PyTorch (torch~=2.3, requirements.txt:19 of the
 * reference), whose published cell definition is restated here:
 *     r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)
 *     z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
 *     n = tanh  (W_in x + b_in + r * (W_hn h + b_hn))
 *     h = (1 - z) * n + z * h            h_0 = 0
 * gate row blocks ordered r,z,n; the reverse direction scans t = T-1..0 and writes its output at
 * t; layer l>0 consumes concat(fwd, bwd) of layer l-1.
 *
 * Parity pinning: the reference's own tests assert only shapes at this boundary
 * (medaka/test/test_architectures.py:58-64), so this file is pinned against outputs of the
 * UNMODIFIED reference classes run on PyTorch-CPU (oracle/make_golden.py -> tests/golden/),
 * see tests/test_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* One direction of one GRU layer for one window.
 * x:   T x I  (row stride ldx)
 * out: T x H written at column offset of the caller's choosing (row stride ldo)
 * w_ih: 3H x I, w_hh: 3H x H (torch layout, row-major), b_ih, b_hh: 3H
 * wt_hh: H x 3H transposed copy of w_hh (so the inner loop runs over outputs). */
static void gru_dir(const float *x, int T, int I, int ldx, float *out, int ldo, int H,
                    const float *w_ih, const float *wt_hh, const float *b_ih,
                    const float *b_hh, int reverse, float *scratch)
{
    const int G = 3 * H;
    float *h = scratch;          /* H  */
    float *gi = scratch + H;     /* 3H */
    float *gh = gi + G;          /* 3H */
    memset(h, 0, sizeof(float) * H);
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? T - 1 - s : s;
        const float *xt = x + (size_t)t * ldx;
        for (int j = 0; j < G; ++j) {
            const float *w = w_ih + (size_t)j * I;
            float acc = 0.0f;
            for (int k = 0; k < I; ++k) acc += w[k] * xt[k];
            gi[j] = acc + b_ih[j];
        }
        for (int j = 0; j < G; ++j) gh[j] = 0.0f;
        for (int k = 0; k < H; ++k) {
            const float hk = h[k];
            const float *w = wt_hh + (size_t)k * G;
            for (int j = 0; j < G; ++j) gh[j] += w[j] * hk;
        }
        float *ot = out + (size_t)t * ldo;
        for (int j = 0; j < H; ++j) {
            const float r = sigmoidf_(gi[j] + (gh[j] + b_hh[j]));
            const float z = sigmoidf_(gi[H + j] + (gh[H + j] + b_hh[H + j]));
            const float n = tanhf(gi[2 * H + j] + r * (gh[2 * H + j] + b_hh[2 * H + j]));
            const float hn = (1.0f - z) * n + z * h[j];
            h[j] = hn;
            ot[j] = hn;
        }
    }
}

/*
 * weights: for layer l in [0,L), direction d in [0,D): w_ih, w_hh, b_ih, b_hh  (torch state_dict
 * order: gru.weight_ih_l{l}[_reverse], weight_hh, bias_ih, bias_hh), then linear.weight (C x D*H),
 * linear.bias (C).  x: B x T x I contiguous, probs: B x T x C.  normalise != 0 -> softmax
 * (gru.py:68-71), else logits.  Returns 0 on success.
 */
int mdk_oracle_gru_forward(const float *x, int B, int T, int I, int H, int L, int bidir, int C,
                           const float *const *weights, int normalise, float *probs)
{
    const int D = bidir ? 2 : 1;
    const int G = 3 * H;
    if (B < 0 || T < 0 || I <= 0 || H <= 0 || L <= 0 || C <= 0) return 1;
    if (B == 0 || T == 0) return 0;
    /* transposed recurrent weights, shared by all windows */
    float **wt = (float **)malloc(sizeof(float *) * L * D);
    for (int ld = 0; ld < L * D; ++ld) {
        const float *w_hh = weights[4 * ld + 1];
        wt[ld] = (float *)malloc(sizeof(float) * (size_t)H * G);
        for (int j = 0; j < G; ++j)
            for (int k = 0; k < H; ++k) wt[ld][(size_t)k * G + j] = w_hh[(size_t)j * H + k];
    }
    const float *lin_w = weights[4 * L * D];
    const float *lin_b = weights[4 * L * D + 1];
    int err = 0;
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; ++b) {
        float *buf0 = (float *)malloc(sizeof(float) * (size_t)T * D * H);
        float *buf1 = (float *)malloc(sizeof(float) * (size_t)T * D * H);
        float *scratch = (float *)malloc(sizeof(float) * (H + 2 * G));
        if (!buf0 || !buf1 || !scratch) { err = 2; free(buf0); free(buf1); free(scratch); continue; }
        const float *in = x + (size_t)b * T * I;
        int in_w = I;
        float *cur = buf0, *nxt = buf1;
        for (int l = 0; l < L; ++l) {
            for (int d = 0; d < D; ++d) {
                const int ld = l * D + d;
                gru_dir(in, T, in_w, in_w, cur + d * H, D * H, H, weights[4 * ld], wt[ld],
                        weights[4 * ld + 2], weights[4 * ld + 3], d, scratch);
            }
            in = cur; in_w = D * H;
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        /* linear + softmax (gru.py:67-71); softmax as torch: subtract max, exp, divide */
        for (int t = 0; t < T; ++t) {
            const float *ht = in + (size_t)t * in_w;
            float *p = probs + ((size_t)b * T + t) * C;
            float m = -INFINITY;
            for (int c = 0; c < C; ++c) {
                const float *w = lin_w + (size_t)c * in_w;
                float acc = 0.0f;
                for (int k = 0; k < in_w; ++k) acc += w[k] * ht[k];
                p[c] = acc + lin_b[c];
                if (p[c] > m) m = p[c];
            }
            if (normalise) {
                float sum = 0.0f;
                for (int c = 0; c < C; ++c) { p[c] = expf(p[c] - m); sum += p[c]; }
                for (int c = 0; c < C; ++c) p[c] = p[c] / sum;
            }
        }
        free(buf0); free(buf1); free(scratch);
    }
    for (int ld = 0; ld < L * D; ++ld) free(wt[ld]);
    free(wt);
    return err;
}

/* MajorityVoteModel.forward (majority_vote_model.py:37-53): channels a c g t A C G T d D ->
 * classes [d+D, a+A, c+C, g+G, t+T]; class 0 += 1 - sum. */
int mdk_oracle_majority_forward(const float *x, long n_cols, float *probs)
{
    for (long i = 0; i < n_cols; ++i) {
        const float *r = x + i * 10;
        float *p = probs + i * 5;
        p[0] = r[8] + r[9];
        for (int c = 0; c < 4; ++c) p[1 + c] = r[c] + r[4 + c];
        float s = p[0];
        for (int c = 1; c < 5; ++c) s += p[c];
        p[0] += 1.0f - s;
    }
    return 0;
}

int mdk_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
