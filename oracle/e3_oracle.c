/*
 * e3_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, not product code) of the
 * arithmetic on elektronn3's 3D U-Net hot path.
 *
 * The reference (ELEKTRONN/elektronn3) is pure Python; every FLOP of
 * elektronn3/models/unet.py is issued through torch (ATen), a third-party
 * dependency that is NOT under /root/reference (requirements.txt:1 `torch>=1.6.0`,
 * present here as torch 2.10.0).  This file therefore restates the *published
 * semantics* of the ATen ops the reference calls, anchored on the reference's
 * call sites:
 *
 *   orc_conv3d_*      nn.Conv3d(k=3|(1,3,3)|1, stride 1, zero padding)   unet.py:131-149,178-180
 *   orc_convT_*       nn.ConvTranspose3d(kernel=stride=2|(1,2,2))        unet.py:152-165
 *   orc_bn_*          nn.BatchNorm3d(eps=1e-5, momentum=0.1)             unet.py:77-105
 *   orc_relu_*        nn.ReLU()                                          unet.py:183-186
 *   orc_maxpool_*     nn.MaxPool3d(k=2|(1,2,2), ceil_mode=True)          unet.py:67-74,225-230
 *   orc_softmax_c     nn.Softmax(1)                                      inference.py:443-444
 *   orc_adamw_step    torch.optim.AdamW(lr, weight_decay)                examples/train_unet_neurodata.py:257-262
 *
 * Layout: contiguous NCDHW fp32 exactly like the reference's tensors.  Sums are
 * accumulated in double so that the oracle is at least as accurate as the fp32
 * reference (the reference's own fp32-vs-fp64 noise floor is what
 * tolerances are stated against).
 *
 * Pinned against golden vectors generated from the imported reference
 * (tests/golden/make_golden.py -> the .npz files in tests/golden; tests/test_oracle_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path (elektronn3_amd/) never does.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX5(n, c, d, h, w, C, D, H, W) \
    (((((size_t)(n) * (C) + (c)) * (D) + (d)) * (H) + (h)) * (W) + (w))

/* y[n,co,d,h,w] = b[co] + sum_{ci,kd,kh,kw} x[n,ci,d+kd-pd,h+kh-ph,w+kw-pw] * w[co,ci,kd,kh,kw]
 * stride 1; output size = in + 2p - k + 1 per axis.  (torch.nn.Conv3d cross-correlation.) */
void orc_conv3d_fwd(const float *x, const float *wt, const float *b, float *y,
                    int N, int Cin, int D, int H, int W, int Cout,
                    int kd, int kh, int kw, int pd, int ph, int pw)
{
    const int Do = D + 2 * pd - kd + 1, Ho = H + 2 * ph - kh + 1, Wo = W + 2 * pw - kw + 1;
#pragma omp parallel for collapse(3) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int d = 0; d < Do; ++d) {
                double *acc = (double *)malloc(sizeof(double) * (size_t)Wo);
                for (int h = 0; h < Ho; ++h) {
                    const double b0 = b ? (double)b[co] : 0.0;
                    for (int w = 0; w < Wo; ++w) acc[w] = b0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int a = 0; a < kd; ++a) {
                            const int id = d + a - pd;
                            if (id < 0 || id >= D) continue;
                            for (int e = 0; e < kh; ++e) {
                                const int ih = h + e - ph;
                                if (ih < 0 || ih >= H) continue;
                                const float *xr = x + IDX5(n, ci, id, ih, 0, Cin, D, H, W);
                                const float *wr = wt + ((((size_t)co * Cin + ci) * kd + a) * kh + e) * kw;
                                for (int f = 0; f < kw; ++f) {
                                    const double wv = wr[f];
                                    const int off = f - pw;
                                    int w0 = off < 0 ? -off : 0;
                                    int w1 = Wo < W - off ? Wo : W - off;
                                    for (int w = w0; w < w1; ++w) acc[w] += wv * (double)xr[w + off];
                                }
                            }
                        }
                    float *yr = y + IDX5(n, co, d, h, 0, Cout, Do, Ho, Wo);
                    for (int w = 0; w < Wo; ++w) yr[w] = (float)acc[w];
                }
                free(acc);
            }
}

/* dx = conv_transpose of dy with w (autograd of Conv3d w.r.t. its input). */
void orc_conv3d_bwd_data(const float *dy, const float *wt, float *dx,
                         int N, int Cin, int D, int H, int W, int Cout,
                         int kd, int kh, int kw, int pd, int ph, int pw)
{
    const int Do = D + 2 * pd - kd + 1, Ho = H + 2 * ph - kh + 1, Wo = W + 2 * pw - kw + 1;
#pragma omp parallel for collapse(3) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int ci = 0; ci < Cin; ++ci)
            for (int d = 0; d < D; ++d) {
                double *acc = (double *)malloc(sizeof(double) * (size_t)W);
                for (int h = 0; h < H; ++h) {
                    for (int w = 0; w < W; ++w) acc[w] = 0.0;
                    for (int co = 0; co < Cout; ++co)
                        for (int a = 0; a < kd; ++a) {
                            const int od = d - a + pd;
                            if (od < 0 || od >= Do) continue;
                            for (int e = 0; e < kh; ++e) {
                                const int oh = h - e + ph;
                                if (oh < 0 || oh >= Ho) continue;
                                const float *gr = dy + IDX5(n, co, od, oh, 0, Cout, Do, Ho, Wo);
                                const float *wr = wt + ((((size_t)co * Cin + ci) * kd + a) * kh + e) * kw;
                                for (int f = 0; f < kw; ++f) {
                                    const double wv = wr[f];
                                    /* ow = w - f + pw */
                                    const int off = pw - f;
                                    int w0 = off < 0 ? -off : 0;
                                    int w1 = W < Wo - off ? W : Wo - off;
                                    for (int w = w0; w < w1; ++w) acc[w] += wv * (double)gr[w + off];
                                }
                            }
                        }
                    float *xr = dx + IDX5(n, ci, d, h, 0, Cin, D, H, W);
                    for (int w = 0; w < W; ++w) xr[w] = (float)acc[w];
                }
                free(acc);
            }
}

/* dw[co,ci,kd,kh,kw] = sum_{n,d,h,w} dy[n,co,d,h,w] * x[n,ci,d+kd-pd,...];  db[co] = sum dy. */
void orc_conv3d_bwd_weight(const float *x, const float *dy, float *dw, float *db,
                           int N, int Cin, int D, int H, int W, int Cout,
                           int kd, int kh, int kw, int pd, int ph, int pw)
{
    const int Do = D + 2 * pd - kd + 1, Ho = H + 2 * ph - kh + 1, Wo = W + 2 * pw - kw + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int a = 0; a < kd; ++a)
                for (int e = 0; e < kh; ++e)
                    for (int f = 0; f < kw; ++f) {
                        double acc = 0.0;
                        for (int n = 0; n < N; ++n)
                            for (int d = 0; d < Do; ++d) {
                                const int id = d + a - pd;
                                if (id < 0 || id >= D) continue;
                                for (int h = 0; h < Ho; ++h) {
                                    const int ih = h + e - ph;
                                    if (ih < 0 || ih >= H) continue;
                                    const float *gr = dy + IDX5(n, co, d, h, 0, Cout, Do, Ho, Wo);
                                    const float *xr = x + IDX5(n, ci, id, ih, 0, Cin, D, H, W);
                                    const int off = f - pw;
                                    int w0 = off < 0 ? -off : 0;
                                    int w1 = Wo < W - off ? Wo : W - off;
                                    double s = 0.0;
                                    for (int w = w0; w < w1; ++w) s += (double)gr[w] * (double)xr[w + off];
                                    acc += s;
                                }
                            }
                        dw[((((size_t)co * Cin + ci) * kd + a) * kh + e) * kw + f] = (float)acc;
                    }
    if (db) {
#pragma omp parallel for schedule(static)
        for (int co = 0; co < Cout; ++co) {
            double acc = 0.0;
            for (int n = 0; n < N; ++n) {
                const float *gr = dy + IDX5(n, co, 0, 0, 0, Cout, Do, Ho, Wo);
                for (size_t i = 0; i < (size_t)Do * Ho * Wo; ++i) acc += gr[i];
            }
            db[co] = (float)acc;
        }
    }
}

/* ConvTranspose3d with kernel == stride == (sd,sh,sw), no padding:
 * y[n,co,sd*d+a,sh*h+e,sw*w+f] = b[co] + sum_ci x[n,ci,d,h,w] * w[ci,co,a,e,f]   (weight is (Cin,Cout,k...)) */
void orc_convT_fwd(const float *x, const float *wt, const float *b, float *y,
                   int N, int Cin, int D, int H, int W, int Cout, int sd, int sh, int sw)
{
    const int Do = D * sd, Ho = H * sh, Wo = W * sw;
#pragma omp parallel for collapse(3) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int od = 0; od < Do; ++od) {
                const int d = od / sd, a = od % sd;
                for (int oh = 0; oh < Ho; ++oh) {
                    const int h = oh / sh, e = oh % sh;
                    for (int ow = 0; ow < Wo; ++ow) {
                        const int w = ow / sw, f = ow % sw;
                        double acc = b ? (double)b[co] : 0.0;
                        for (int ci = 0; ci < Cin; ++ci)
                            acc += (double)x[IDX5(n, ci, d, h, w, Cin, D, H, W)] *
                                   (double)wt[((((size_t)ci * Cout + co) * sd + a) * sh + e) * sw + f];
                        y[IDX5(n, co, od, oh, ow, Cout, Do, Ho, Wo)] = (float)acc;
                    }
                }
            }
}

void orc_convT_bwd_data(const float *dy, const float *wt, float *dx,
                        int N, int Cin, int D, int H, int W, int Cout, int sd, int sh, int sw)
{
    const int Do = D * sd, Ho = H * sh, Wo = W * sw;
#pragma omp parallel for collapse(3) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int ci = 0; ci < Cin; ++ci)
            for (int d = 0; d < D; ++d)
                for (int h = 0; h < H; ++h)
                    for (int w = 0; w < W; ++w) {
                        double acc = 0.0;
                        for (int co = 0; co < Cout; ++co)
                            for (int a = 0; a < sd; ++a)
                                for (int e = 0; e < sh; ++e)
                                    for (int f = 0; f < sw; ++f)
                                        acc += (double)dy[IDX5(n, co, d * sd + a, h * sh + e, w * sw + f, Cout, Do, Ho, Wo)] *
                                               (double)wt[((((size_t)ci * Cout + co) * sd + a) * sh + e) * sw + f];
                        dx[IDX5(n, ci, d, h, w, Cin, D, H, W)] = (float)acc;
                    }
}

void orc_convT_bwd_weight(const float *x, const float *dy, float *dw, float *db,
                          int N, int Cin, int D, int H, int W, int Cout, int sd, int sh, int sw)
{
    const int Do = D * sd, Ho = H * sh, Wo = W * sw;
#pragma omp parallel for collapse(2) schedule(static)
    for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co)
            for (int a = 0; a < sd; ++a)
                for (int e = 0; e < sh; ++e)
                    for (int f = 0; f < sw; ++f) {
                        double acc = 0.0;
                        for (int n = 0; n < N; ++n)
                            for (int d = 0; d < D; ++d)
                                for (int h = 0; h < H; ++h)
                                    for (int w = 0; w < W; ++w)
                                        acc += (double)x[IDX5(n, ci, d, h, w, Cin, D, H, W)] *
                                               (double)dy[IDX5(n, co, d * sd + a, h * sh + e, w * sw + f, Cout, Do, Ho, Wo)];
                        dw[((((size_t)ci * Cout + co) * sd + a) * sh + e) * sw + f] = (float)acc;
                    }
    if (db) {
#pragma omp parallel for schedule(static)
        for (int co = 0; co < Cout; ++co) {
            double acc = 0.0;
            for (int n = 0; n < N; ++n) {
                const float *gr = dy + IDX5(n, co, 0, 0, 0, Cout, Do, Ho, Wo);
                for (size_t i = 0; i < (size_t)Do * Ho * Wo; ++i) acc += gr[i];
            }
            db[co] = (float)acc;
        }
    }
}

/* BatchNorm (training): per-channel mean / biased variance over (N, spatial); S = D*H*W.
 * y = gamma*(x-mean)/sqrt(var+eps)+beta;
 * running_mean <- (1-m)*running_mean + m*mean;  running_var <- (1-m)*running_var + m*var*n/(n-1).
 * save_mean / save_invstd are returned for the backward. */
void orc_bn_train_fwd(const float *x, const float *gamma, const float *beta, float *y,
                      float *save_mean, float *save_invstd, float *running_mean, float *running_var,
                      double momentum, double eps, int N, int C, size_t S)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const double cnt = (double)N * (double)S;
        double s = 0.0;
        for (int n = 0; n < N; ++n) {
            const float *p = x + ((size_t)n * C + c) * S;
            for (size_t i = 0; i < S; ++i) s += p[i];
        }
        const double mean = s / cnt;
        double m2 = 0.0;
        for (int n = 0; n < N; ++n) {
            const float *p = x + ((size_t)n * C + c) * S;
            for (size_t i = 0; i < S; ++i) { const double dlt = p[i] - mean; m2 += dlt * dlt; }
        }
        const double var = m2 / cnt;
        const double invstd = 1.0 / sqrt(var + eps);
        const double g = gamma ? gamma[c] : 1.0, bb = beta ? beta[c] : 0.0;
        for (int n = 0; n < N; ++n) {
            const float *p = x + ((size_t)n * C + c) * S;
            float *q = y + ((size_t)n * C + c) * S;
            for (size_t i = 0; i < S; ++i) q[i] = (float)((p[i] - mean) * invstd * g + bb);
        }
        if (save_mean) save_mean[c] = (float)mean;
        if (save_invstd) save_invstd[c] = (float)invstd;
        if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        if (running_var) {
            const double unbiased = cnt > 1.0 ? m2 / (cnt - 1.0) : var;
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
    }
}

void orc_bn_eval_fwd(const float *x, const float *gamma, const float *beta,
                     const float *running_mean, const float *running_var, float *y,
                     double eps, int N, int C, size_t S)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const double invstd = 1.0 / sqrt((double)running_var[c] + eps);
        const double g = gamma ? gamma[c] : 1.0, bb = beta ? beta[c] : 0.0, mean = running_mean[c];
        for (int n = 0; n < N; ++n) {
            const float *p = x + ((size_t)n * C + c) * S;
            float *q = y + ((size_t)n * C + c) * S;
            for (size_t i = 0; i < S; ++i) q[i] = (float)((p[i] - mean) * invstd * g + bb);
        }
    }
}

/* BatchNorm backward (training-mode statistics):
 * dgamma = sum dy*xhat; dbeta = sum dy; dx = gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat)). */
void orc_bn_train_bwd(const float *dy, const float *x, const float *gamma,
                      const float *save_mean, const float *save_invstd,
                      float *dx, float *dgamma, float *dbeta, int N, int C, size_t S)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const double cnt = (double)N * (double)S;
        const double mean = save_mean[c], invstd = save_invstd[c];
        double sdy = 0.0, sdyx = 0.0;
        for (int n = 0; n < N; ++n) {
            const float *g = dy + ((size_t)n * C + c) * S;
            const float *p = x + ((size_t)n * C + c) * S;
            for (size_t i = 0; i < S; ++i) { sdy += g[i]; sdyx += (double)g[i] * ((p[i] - mean) * invstd); }
        }
        const double gm = gamma ? gamma[c] : 1.0;
        const double k = gm * invstd, mdy = sdy / cnt, mdyx = sdyx / cnt;
        for (int n = 0; n < N; ++n) {
            const float *g = dy + ((size_t)n * C + c) * S;
            const float *p = x + ((size_t)n * C + c) * S;
            float *q = dx + ((size_t)n * C + c) * S;
            for (size_t i = 0; i < S; ++i) {
                const double xh = (p[i] - mean) * invstd;
                q[i] = (float)(k * (g[i] - mdy - xh * mdyx));
            }
        }
        if (dgamma) dgamma[c] = (float)sdyx;
        if (dbeta) dbeta[c] = (float)sdy;
    }
}

void orc_relu_fwd(const float *x, float *y, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) y[i] = x[i] > 0.f ? x[i] : 0.f;
}

/* torch threshold_backward: dx = dy where out > 0 else 0 */
void orc_relu_bwd(const float *dy, const float *out, float *dx, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) dx[i] = out[i] > 0.f ? dy[i] : 0.f;
}

/* MaxPool3d(kernel=stride=(kd,kh,kw), ceil_mode=True, no padding): out = ceil(in/k);
 * windows are clipped at the high edge.  idx = flat (d*H+h)*W+w index of the FIRST maximum in
 * scan order (ATen max_pool3d_with_indices: `val > maxval || isnan(val)`). */
void orc_maxpool_fwd(const float *x, float *y, int64_t *idx,
                     int N, int C, int D, int H, int W, int kd, int kh, int kw)
{
    const int Do = (D + kd - 1) / kd, Ho = (H + kh - 1) / kh, Wo = (W + kw - 1) / kw;
#pragma omp parallel for schedule(static)
    for (int nc = 0; nc < N * C; ++nc) {
        const float *p = x + (size_t)nc * D * H * W;
        for (int d = 0; d < Do; ++d)
            for (int h = 0; h < Ho; ++h)
                for (int w = 0; w < Wo; ++w) {
                    float best = -INFINITY; int64_t bi = -1;
                    for (int a = d * kd; a < (d + 1) * kd && a < D; ++a)
                        for (int e = h * kh; e < (h + 1) * kh && e < H; ++e)
                            for (int f = w * kw; f < (w + 1) * kw && f < W; ++f) {
                                const int64_t i = ((int64_t)a * H + e) * W + f;
                                const float v = p[i];
                                if (bi < 0 || v > best || isnan(v)) { best = v; bi = i; }
                            }
                    const size_t o = ((size_t)nc * Do + d) * Ho * Wo + (size_t)h * Wo + w;
                    y[o] = best;
                    if (idx) idx[o] = bi;
                }
    }
}

void orc_maxpool_bwd(const float *dy, const int64_t *idx, float *dx,
                     int N, int C, int D, int H, int W, int kd, int kh, int kw)
{
    const int Do = (D + kd - 1) / kd, Ho = (H + kh - 1) / kh, Wo = (W + kw - 1) / kw;
    memset(dx, 0, sizeof(float) * (size_t)N * C * D * H * W);
#pragma omp parallel for schedule(static)
    for (int nc = 0; nc < N * C; ++nc) {
        float *q = dx + (size_t)nc * D * H * W;
        const size_t so = (size_t)Do * Ho * Wo;
        for (size_t o = 0; o < so; ++o) q[idx[(size_t)nc * so + o]] += dy[(size_t)nc * so + o];
    }
}

/* Softmax over the channel axis of an (N,C,S) tensor. */
void orc_softmax_c(const float *x, float *y, int N, int C, size_t S)
{
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (size_t i = 0; i < S; ++i) {
            double m = -INFINITY;
            for (int c = 0; c < C; ++c) { const double v = x[((size_t)n * C + c) * S + i]; if (v > m) m = v; }
            double s = 0.0;
            for (int c = 0; c < C; ++c) s += exp((double)x[((size_t)n * C + c) * S + i] - m);
            for (int c = 0; c < C; ++c)
                y[((size_t)n * C + c) * S + i] = (float)(exp((double)x[((size_t)n * C + c) * S + i] - m) / s);
        }
}

/* torch.optim.AdamW single step (decoupled weight decay, no amsgrad), the optimizer of the reference example
 * (examples/train_unet_neurodata.py:257-262; stepped in training/trainer.py:539-542).  Published semantics
 * (torch/optim/adamw.py, "_single_tensor_adam" with decoupled_weight_decay):
 *   p <- p * (1 - lr*wd);  m <- m + (1-b1)(g - m);  v <- b2*v + (1-b2) g*g
 *   p <- p - lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)            t = step count AFTER the increment
 * Arithmetic in double, state stored as fp32 like torch's. */
void orc_adamw_step(float *p, const float *g, float *m, float *v, size_t n, int t,
                    double lr, double b1, double b2, double eps, double wd)
{
    const double bc1 = 1.0 - pow(b1, (double)t), bc2s = sqrt(1.0 - pow(b2, (double)t));
    for (size_t i = 0; i < n; ++i) {
        double pi = (double)p[i] * (1.0 - lr * wd);
        const double gi = g[i];
        const double mi = (double)m[i] + (1.0 - b1) * (gi - (double)m[i]);
        const double vi = (double)v[i] * b2 + (1.0 - b2) * gi * gi;
        m[i] = (float)mi; v[i] = (float)vi;
        pi -= lr / bc1 * ((double)m[i] / (sqrt((double)v[i]) / bc2s + eps));
        p[i] = (float)pi;
    }
}
