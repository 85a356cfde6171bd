/* A plain-C host of the engine: what a non-Python embedder (or a cffi cdef) sees.  Builds with
 *     gcc -std=c99 -I include tests/c/abi_smoke.c -L medaka_amd -lmedaka_amd -lm
 * and, on a GPU box, runs a 2-window forward through mdk_gru_create / mdk_gru_forward and checks
 * the softmax rows.  tests/test_host.py compiles and links it (no GPU needed for that),
 * tests/test_parity_gpu.py runs it. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "medaka_amd.h"

static float frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xffff) / 65536.0f - 0.5f; }

int main(void) {
    const int I = 10, H = 128, G = 3 * H, B = 2, T = 300;
    /* state_dict order: per layer and direction weight_ih, weight_hh, bias_ih, bias_hh; linear w, b */
    const size_t sizes[18] = {(size_t)G * I, (size_t)G * H, G, G, (size_t)G * I, (size_t)G * H, G, G,
                              (size_t)G * 2 * H, (size_t)G * H, G, G, (size_t)G * 2 * H, (size_t)G * H, G, G,
                              (size_t)5 * 2 * H, 5};
    const float *w[18];
    unsigned seed = 7;
    for (int i = 0; i < 18; ++i) {
        float *p = (float *)malloc(sizes[i] * sizeof(float));
        for (size_t j = 0; j < sizes[i]; ++j) p[j] = 0.2f * frand(&seed);
        w[i] = p;
    }
    mdk_gru_desc desc = {I, H, 2, 1, 5, 1};
    mdk_gru *m = NULL;
    int n_dev = 0;
    if (mdk_device_count(&n_dev) != MDK_OK || n_dev < 1) {
        printf("no HIP device: %s\n", mdk_last_error());
        return 77;   /* "skipped" */
    }
    if (mdk_gru_create(&desc, w, 18, 0, &m) != MDK_OK) {
        printf("mdk_gru_create: %s\n", mdk_last_error());
        return 1;
    }
    float *x = (float *)malloc((size_t)B * T * I * sizeof(float));
    float *p = (float *)malloc((size_t)B * T * 5 * sizeof(float));
    for (int i = 0; i < B * T * I; ++i) x[i] = frand(&seed) + 0.5f;
    if (mdk_gru_forward(m, x, B, T, p) != MDK_OK) {
        printf("mdk_gru_forward: %s\n", mdk_last_error());
        return 1;
    }
    double worst = 0.0;
    for (int r = 0; r < B * T; ++r) {
        double s = 0.0;
        for (int c = 0; c < 5; ++c) {
            if (!(p[r * 5 + c] >= 0.0f && p[r * 5 + c] <= 1.0f)) { printf("bad probability\n"); return 1; }
            s += p[r * 5 + c];
        }
        if (fabs(s - 1.0) > worst) worst = fabs(s - 1.0);
    }
    /* the streamed host path: a window long enough for time slabs (T >= 2048, T % 16 == 0), page-locked buffers
     * from the ABI; must equal the same call with one copy each side ("stream_host" = 0), bit for bit */
    {
        const int T2 = 2304;
        float *x2 = NULL, *pa = NULL, *pb = NULL;
        if (mdk_host_alloc((size_t)B * T2 * I * sizeof(float), (void **)&x2) != MDK_OK ||
            mdk_host_alloc((size_t)B * T2 * 5 * sizeof(float), (void **)&pa) != MDK_OK ||
            mdk_host_alloc((size_t)B * T2 * 5 * sizeof(float), (void **)&pb) != MDK_OK) {
            printf("mdk_host_alloc: %s\n", mdk_last_error());
            return 1;
        }
        for (int i = 0; i < B * T2 * I; ++i) x2[i] = frand(&seed) + 0.5f;
        if (mdk_gru_forward(m, x2, B, T2, pa) != MDK_OK || mdk_gru_set_option(m, "stream_host", 0) != MDK_OK ||
            mdk_gru_forward(m, x2, B, T2, pb) != MDK_OK || mdk_gru_set_option(m, "stream_host", 1) != MDK_OK) {
            printf("streamed forward: %s\n", mdk_last_error());
            return 1;
        }
        for (int i = 0; i < B * T2 * 5; ++i)
            if (pa[i] != pb[i]) { printf("streamed and plain host paths differ at %d\n", i); return 1; }
        /* early hand-over: the copy of x2 starts now, the forward redeems the token later; a token is good once */
        unsigned long long token = 0;
        for (int i = 0; i < B * T2 * 5; ++i) pb[i] = -1.0f;
        if (mdk_gru_stage_input(m, x2, B, T2, &token) != MDK_OK || token == 0 ||
            mdk_gru_forward_staged(m, token, B, T2, pb) != MDK_OK) {
            printf("staged forward: %s\n", mdk_last_error());
            return 1;
        }
        for (int i = 0; i < B * T2 * 5; ++i)
            if (pa[i] != pb[i]) { printf("staged and ordinary forwards differ at %d\n", i); return 1; }
        if (mdk_gru_forward_staged(m, token, B, T2, pb) != MDK_ERR_ARG) { printf("a spent token must be refused\n"); return 1; }
        mdk_host_free(x2); mdk_host_free(pa); mdk_host_free(pb);
    }
    /* argument errors come back as codes + message, never as exit() */
    if (mdk_gru_forward(m, NULL, B, T, p) != MDK_ERR_ARG) { printf("expected MDK_ERR_ARG\n"); return 1; }
    mdk_gru_destroy(m);
    printf("ok %s rows %d max|sum-1| %.2e\n", mdk_version(), B * T, worst);
    return worst < 1e-5 ? 0 : 1;
}
