/*
 * oracle/daf_oracle.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Plain-C CPU restatement of the reference's two CUDA kernels for the
 * deformable_aggregation operator.  It exists so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can check / time the HIP path against the reference
 * algorithm on a box that has neither CUDA nor /root/reference.
 *
 * Followed text (reference @ /root/reference, read as text, not compiled):
 *   projects/mmdet3d_plugin/ops/src/deformable_aggregation_cuda.cu
 *     :13-59    bilinear_sampling            -> bilinear_fwd()
 *     :62-126   bilinear_sampling_grad       -> bilinear_bwd()
 *     :129-187  deformable_aggregation_kernel       -> hipad_oracle_daf_forward()
 *     :190-262  deformable_aggregation_grad_kernel  -> hipad_oracle_daf_backward()
 *
 * The reference launches one CUDA thread per flat index over
 * (batch, anchor, pts, cam, scale, channel) and combines with float atomicAdd, so its
 * summation order is undefined.  This restatement walks the same flat index space in
 * increasing order (the order a sequential machine would retire the threads) and keeps
 * every arithmetic expression in the reference's type and order:
 *   - index math in fp32 exactly as written there: `loc * size - 0.5` is a float product
 *     followed by a subtraction carried out in double and rounded to float (cu:180-181);
 *     floorf; int offsets;
 *   - sample dropped iff loc_w<=0 || loc_w>=1 || loc_h<=0 || loc_h>=1 (cu:168-171);
 *   - corner contributes iff inside the map (cu:33-52).
 * Accumulation is available in fp32 (what the GPU does, order aside) or fp64
 * (acc64 != 0: a tighter yardstick for tolerance tests).
 *
 * Build with -ffp-contract=off so the host compiler cannot fuse the index arithmetic.
 *
 * Parity status: the CUDA op itself cannot be built here (no nvcc, no GPU, and it
 * includes THC/THCAtomics.cuh which current torch no longer ships) -- this file is
 * pinned instead against the reference's own PyTorch fallback
 * (models/blocks.py:227-264, imported in the build container) through the fixtures in
 * tests/golden/ (see tests/golden/make_golden.py and tests/test_oracle_golden.py).
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/*
 * Threads (bench.py's cpu_baseline leg only; the checker default is ONE thread = the sequential order described above).
 * Anchors are dealt to threads for everything an anchor owns (its output row, grad_loc, grad_w); grad_feat is scattered
 * in a second pass in which thread t applies the updates of pyramid rows r with r % threads == t, in the sequential
 * order.  Both passes give bitwise the one-thread result (tests/test_oracle_golden.py checks it).
 */
static int g_threads = 1;
void hipad_oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int hipad_oracle_get_threads(void) { return g_threads; }

typedef struct {
    int h_low, w_low, h_high, w_high;
    float lh, lw, hh, hw;
    int64_t p1, p2, p3, p4; /* element offsets of the four corners (channel 0) */
    int in1, in2, in3, in4;
} taps_t;

/* cu:18-31 + cu:33-52 (bounds tests) */
static void make_taps(int height, int width, int num_embeds, float h_im, float w_im,
                      int64_t base, taps_t *t) {
    /* float->int of NaN is undefined in C; the GPU conversion (cvt.rzi / v_cvt_i32_f32) gives 0 */
    t->h_low = isnan(h_im) ? 0 : (int)floorf(h_im);
    t->w_low = isnan(w_im) ? 0 : (int)floorf(w_im);
    t->h_high = t->h_low + 1;
    t->w_high = t->w_low + 1;
    t->lh = h_im - (float)t->h_low;
    t->lw = w_im - (float)t->w_low;
    t->hh = 1 - t->lh;
    t->hw = 1 - t->lw;
    const int64_t w_stride = num_embeds;
    const int64_t h_stride = (int64_t)width * w_stride;
    const int64_t hl = (int64_t)t->h_low * h_stride, hhp = hl + h_stride;
    const int64_t wl = (int64_t)t->w_low * w_stride, whp = wl + w_stride;
    t->in1 = (t->h_low >= 0 && t->w_low >= 0);
    t->in2 = (t->h_low >= 0 && t->w_high <= width - 1);
    t->in3 = (t->h_high <= height - 1 && t->w_low >= 0);
    t->in4 = (t->h_high <= height - 1 && t->w_high <= width - 1);
    t->p1 = hl + wl + base;
    t->p2 = hl + whp + base;
    t->p3 = hhp + wl + base;
    t->p4 = hhp + whp + base;
}

/* cu:180-181: float product, subtraction of the double literal 0.5, rounded to float */
static inline float pix(float loc, int size) {
    float prod = loc * (float)size;
    return (float)((double)prod - 0.5);
}

/* Sample valid?  cu:168-171 (NaN falls through as in the reference: neither test fires). */
static inline int loc_rejected(float loc_w, float loc_h) {
    if (loc_w <= 0 || loc_w >= 1) return 1;
    if (loc_h <= 0 || loc_h >= 1) return 1;
    return 0;
}

/*
 * Forward.  Layouts (ops/src/deformable_aggregation.cpp:23-29):
 *   feat  [bs, num_feat, C] f32      spatial_shape [cams, scales, 2] i32 (h, w)
 *   scale_start_index [cams, scales] i32
 *   loc   [bs, A, P, cams, 2] f32 (x=w, y=h, normalised)
 *   w     [bs, A, P, cams, scales, G] f32        out [bs, A, C] f32 (overwritten)
 */
int hipad_oracle_daf_forward(const float *feat, const int32_t *spatial_shape,
                             const int32_t *scale_start_index, const float *loc,
                             const float *weights, float *out, int batch_size, int num_cams,
                             int num_feat, int num_embeds, int num_scale, int num_anchors,
                             int num_pts, int num_groups, int acc64) {
    if (num_groups <= 0 || num_embeds % num_groups) return -1;
    const int gdim = num_embeds / num_groups;
    const size_t n_out = (size_t)batch_size * num_anchors * num_embeds;
    double *acc = NULL;
    if (acc64) {
        acc = (double *)__builtin_malloc(n_out * sizeof(double));
        if (!acc) return -2;
        memset(acc, 0, n_out * sizeof(double));
    }
    memset(out, 0, n_out * sizeof(float));
#pragma omp parallel for collapse(2) schedule(dynamic, 4) num_threads(g_threads)
    for (int b = 0; b < batch_size; ++b)
        for (int a = 0; a < num_anchors; ++a) {
            const int64_t anchor_index = (int64_t)b * num_anchors + a;
            for (int p = 0; p < num_pts; ++p)
                for (int cam = 0; cam < num_cams; ++cam) {
                    const int64_t loc_offset = ((anchor_index * num_pts + p) * num_cams + cam) << 1;
                    const float loc_w = loc[loc_offset], loc_h = loc[loc_offset + 1];
                    if (loc_rejected(loc_w, loc_h)) continue;
                    for (int s = 0; s < num_scale; ++s) {
                        const int cs = cam * num_scale + s;
                        const int64_t base =
                            ((int64_t)b * num_feat + scale_start_index[cs]) * num_embeds;
                        const int h = spatial_shape[2 * cs], w = spatial_shape[2 * cs + 1];
                        taps_t t;
                        make_taps(h, w, num_embeds, pix(loc_h, h), pix(loc_w, w), base, &t);
                        const float w1 = t.hh * t.hw, w2 = t.hh * t.lw, w3 = t.lh * t.hw,
                                    w4 = t.lh * t.lw;
                        /* cu:149: weight index = flat thread index / (C/G) */
                        const float *wrow =
                            weights + ((loc_offset >> 1) * num_scale + s) * num_groups;
                        for (int c = 0; c < num_embeds; ++c) {
                            const float v1 = t.in1 ? feat[t.p1 + c] : 0.f;
                            const float v2 = t.in2 ? feat[t.p2 + c] : 0.f;
                            const float v3 = t.in3 ? feat[t.p3 + c] : 0.f;
                            const float v4 = t.in4 ? feat[t.p4 + c] : 0.f;
                            const float val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                            const float contrib = val * wrow[c / gdim];
                            if (acc64)
                                acc[anchor_index * num_embeds + c] += (double)contrib;
                            else
                                out[anchor_index * num_embeds + c] += contrib;
                        }
                    }
                }
        }
    if (acc64) {
        for (size_t i = 0; i < n_out; ++i) out[i] = (float)acc[i];
        __builtin_free(acc);
    }
    return 0;
}

/* grad_feat of the rows owned by thread `tid` of `nt` (row r belongs to thread r % nt): every thread walks the whole
 * index space in the sequential order but applies only its rows' updates, so each row receives its addends in exactly the
 * order of the one-thread code (bitwise the same result), without atomics. */
static void feat_scatter_rows(const int32_t *spatial_shape, const int32_t *scale_start_index, const float *loc,
                              const float *weights, const float *grad_out, float *grad_feat, double *gf64,
                              int batch_size, int num_cams, int num_feat, int num_embeds, int num_scale, int num_anchors,
                              int num_pts, int num_groups, int tid, int nt) {
    const int gdim = num_embeds / num_groups;
    for (int b = 0; b < batch_size; ++b)
        for (int a = 0; a < num_anchors; ++a) {
            const int64_t anchor_index = (int64_t)b * num_anchors + a;
            const float *go = grad_out + anchor_index * num_embeds;
            for (int p = 0; p < num_pts; ++p)
                for (int cam = 0; cam < num_cams; ++cam) {
                    const int64_t loc_offset = ((anchor_index * num_pts + p) * num_cams + cam) << 1;
                    const float loc_w = loc[loc_offset], loc_h = loc[loc_offset + 1];
                    if (loc_rejected(loc_w, loc_h)) continue;
                    for (int s = 0; s < num_scale; ++s) {
                        const int cs = cam * num_scale + s;
                        const int64_t base = ((int64_t)b * num_feat + scale_start_index[cs]) * num_embeds;
                        const int h = spatial_shape[2 * cs], w = spatial_shape[2 * cs + 1];
                        taps_t t;
                        make_taps(h, w, num_embeds, pix(loc_h, h), pix(loc_w, w), base, &t);
                        const float cw[4] = {t.hh * t.hw, t.hh * t.lw, t.lh * t.hw, t.lh * t.lw};
                        const int64_t cp[4] = {t.p1, t.p2, t.p3, t.p4};
                        const int cin[4] = {t.in1, t.in2, t.in3, t.in4};
                        const int64_t wbase = ((loc_offset >> 1) * num_scale + s) * num_groups;
                        for (int k = 0; k < 4; ++k) {
                            if (!cin[k] || (int)((cp[k] / num_embeds) % nt) != tid) continue;
                            for (int c = 0; c < num_embeds; ++c) {
                                const float top = go[c] * weights[wbase + c / gdim];
                                if (gf64) gf64[cp[k] + c] += (double)(cw[k] * top);
                                else grad_feat[cp[k] + c] += cw[k] * top;
                            }
                        }
                    }
                }
        }
}

/*
 * Backward (cu:190-262 + cu:62-126).  grad_* are ACCUMULATED INTO (the reference's
 * caller passes zero-initialised tensors, ops/deformable_aggregation.py:55-57).
 *   grad_out [bs, A, C]; grad_feat like feat; grad_loc like loc; grad_w like w.
 * With acc64 the three gradients are accumulated in double scratch and added once.
 */
int hipad_oracle_daf_backward(const float *feat, const int32_t *spatial_shape,
                              const int32_t *scale_start_index, const float *loc,
                              const float *weights, const float *grad_out, float *grad_feat,
                              float *grad_loc, float *grad_w, int batch_size, int num_cams,
                              int num_feat, int num_embeds, int num_scale, int num_anchors,
                              int num_pts, int num_groups, int acc64) {
    if (num_groups <= 0 || num_embeds % num_groups) return -1;
    const int gdim = num_embeds / num_groups;
    const size_t n_feat = (size_t)batch_size * num_feat * num_embeds;
    double *gf64 = NULL;
    if (acc64) {
        gf64 = (double *)__builtin_malloc(n_feat * sizeof(double));
        if (!gf64) return -2;
        memset(gf64, 0, n_feat * sizeof(double));
    }
    const int mt = g_threads > 1; /* several threads: grad_feat is scattered in the second, row-partitioned pass */
#pragma omp parallel for collapse(2) schedule(dynamic, 4) num_threads(g_threads)
    for (int b = 0; b < batch_size; ++b)
        for (int a = 0; a < num_anchors; ++a) {
            const int64_t anchor_index = (int64_t)b * num_anchors + a;
            const float *go = grad_out + anchor_index * num_embeds;
            for (int p = 0; p < num_pts; ++p)
                for (int cam = 0; cam < num_cams; ++cam) {
                    const int64_t loc_offset = ((anchor_index * num_pts + p) * num_cams + cam) << 1;
                    const float loc_w = loc[loc_offset], loc_h = loc[loc_offset + 1];
                    if (loc_rejected(loc_w, loc_h)) continue;
                    double gl0 = 0, gl1 = 0;
                    float gl0f = 0, gl1f = 0;
                    for (int s = 0; s < num_scale; ++s) {
                        const int cs = cam * num_scale + s;
                        const int64_t base =
                            ((int64_t)b * num_feat + scale_start_index[cs]) * num_embeds;
                        const int h = spatial_shape[2 * cs], w = spatial_shape[2 * cs + 1];
                        taps_t t;
                        make_taps(h, w, num_embeds, pix(loc_h, h), pix(loc_w, w), base, &t);
                        const float w1 = t.hh * t.hw, w2 = t.hh * t.lw, w3 = t.lh * t.hw,
                                    w4 = t.lh * t.lw;
                        const int64_t wbase = ((loc_offset >> 1) * num_scale + s) * num_groups;
                        for (int c = 0; c < num_embeds; ++c) {
                            const float weight = weights[wbase + c / gdim];
                            const float top = go[c] * weight; /* cu:86 */
                            float gh = 0, gw = 0;
                            float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                            if (t.in1) {
                                v1 = feat[t.p1 + c];
                                gh -= t.hw * v1;
                                gw -= t.hh * v1;
                                if (!mt) {
                                    if (acc64) gf64[t.p1 + c] += (double)(w1 * top);
                                    else grad_feat[t.p1 + c] += w1 * top;
                                }
                            }
                            if (t.in2) {
                                v2 = feat[t.p2 + c];
                                gh -= t.lw * v2;
                                gw += t.hh * v2;
                                if (!mt) {
                                    if (acc64) gf64[t.p2 + c] += (double)(w2 * top);
                                    else grad_feat[t.p2 + c] += w2 * top;
                                }
                            }
                            if (t.in3) {
                                v3 = feat[t.p3 + c];
                                gh += t.hw * v3;
                                gw -= t.lh * v3;
                                if (!mt) {
                                    if (acc64) gf64[t.p3 + c] += (double)(w3 * top);
                                    else grad_feat[t.p3 + c] += w3 * top;
                                }
                            }
                            if (t.in4) {
                                v4 = feat[t.p4 + c];
                                gh += t.lw * v4;
                                gw += t.lh * v4;
                                if (!mt) {
                                    if (acc64) gf64[t.p4 + c] += (double)(w4 * top);
                                    else grad_feat[t.p4 + c] += w4 * top;
                                }
                            }
                            const float val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                            /* cu:122-125 */
                            const float gwt = go[c] * val;
                            const float g0 = (float)w * gw * top;
                            const float g1 = (float)h * gh * top;
                            if (acc64) {
                                /* per-group partial in double, flushed below */
                                gl0 += (double)g0;
                                gl1 += (double)g1;
                            } else {
                                gl0f += g0;
                                gl1f += g1;
                            }
                            grad_w[wbase + c / gdim] += gwt; /* <=32 addends: fp32 is fine */
                        }
                    }
                    if (acc64) {
                        grad_loc[loc_offset] += (float)gl0;
                        grad_loc[loc_offset + 1] += (float)gl1;
                    } else {
                        grad_loc[loc_offset] += gl0f;
                        grad_loc[loc_offset + 1] += gl1f;
                    }
                }
        }
    if (mt) {
#pragma omp parallel num_threads(g_threads)
        {
#ifdef _OPENMP
            const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
            const int tid = 0, nt = 1;
#endif
            feat_scatter_rows(spatial_shape, scale_start_index, loc, weights, grad_out, grad_feat, gf64, batch_size,
                              num_cams, num_feat, num_embeds, num_scale, num_anchors, num_pts, num_groups, tid, nt);
        }
    }
    if (acc64) {
        for (size_t i = 0; i < n_feat; ++i) grad_feat[i] += (float)gf64[i];
        __builtin_free(gf64);
    }
    return 0;
}

/*
 * Index work only (bit-exact class of the north star): for every (b, a, p, cam) the
 * valid flag, and for every scale the integer corner coordinates and in-bounds mask.
 *   valid [bs*A*P*cams] u8;  taps [bs*A*P*cams*scales*4] i32 = (h_low, w_low, mask, base_row)
 * where mask bit k = corner k in bounds and base_row = b*num_feat + scale_start_index.
 */
int hipad_oracle_daf_taps(const int32_t *spatial_shape, const int32_t *scale_start_index,
                          const float *loc, uint8_t *valid, int32_t *taps, int batch_size,
                          int num_cams, int num_feat, int num_scale, int num_anchors,
                          int num_pts) {
    const int64_t n = (int64_t)batch_size * num_anchors * num_pts * num_cams;
    for (int64_t i = 0; i < n; ++i) {
        const int cam = (int)(i % num_cams);
        const int b = (int)(i / ((int64_t)num_cams * num_pts * num_anchors));
        const float loc_w = loc[2 * i], loc_h = loc[2 * i + 1];
        const int rej = loc_rejected(loc_w, loc_h);
        valid[i] = (uint8_t)!rej;
        for (int s = 0; s < num_scale; ++s) {
            int32_t *o = taps + (i * num_scale + s) * 4;
            if (rej) {
                o[0] = o[1] = o[2] = o[3] = 0;
                continue;
            }
            const int cs = cam * num_scale + s;
            const int h = spatial_shape[2 * cs], w = spatial_shape[2 * cs + 1];
            taps_t t;
            make_taps(h, w, 1, pix(loc_h, h), pix(loc_w, w), 0, &t);
            o[0] = t.h_low;
            o[1] = t.w_low;
            o[2] = t.in1 | (t.in2 << 1) | (t.in3 << 2) | (t.in4 << 3);
            o[3] = b * num_feat + scale_start_index[cs];
        }
    }
    return 0;
}
