// cvx_vit_encode / cvx_head_forward: the two hot-path calls as single C-ABI entry points.  They are launch sequences
// over the op-level kernels (no allocation, no synchronisation, caller's stream: hipGraph-capturable); weights and
// workspaces are caller-owned device buffers described by plain structs.
#include "../../include/cryovit_hip.h"
#include "host_util.h"

#define CVX_TRY(expr)            \
    do {                         \
        int _rc = (expr);        \
        if (_rc) return _rc;     \
    } while (0)

static cvx_gemm_desc gemm_base(int epi, const void* a, long lda, const void* w, long ldw, long m, long n, long n_pad, long k_pad,
                               void* out, long ldc, const float* bias) {
    cvx_gemm_desc d = {};
    d.epilogue = epi; d.a = a; d.lda = lda; d.w = w; d.ldw = ldw; d.m = m; d.n = n; d.n_pad = n_pad; d.k_pad = k_pad;
    d.out = out; d.ldc = ldc; d.bias = bias;
    return d;
}
static long rup(long x, long m) { return (x + m - 1) / m * m; }

// The product path: the residual stream lives as a bf16 (hi, lo) pair, `hi` IS the A operand of the qk / V^T / FFN-in GEMMs, whose
// epilogues apply the LayerNorm from per-row constants (rstd, -mean * rstd); those come out of the previous residual GEMM's
// epilogue as 64-column partial sums (cvx_rowstat_finalize: one ~5-us launch).  No LayerNorm pass, no xn buffer.
static int vit_blocks_folded(const cvx_vit_desc* v, const cvx_vit_ws* ws, int b, int hp, int wp, int nt, int ntp, int kp, void* feats_f16,
                             long d_total, long d0, void* feats_cl, float* tokens_f32, hipStream_t st) {
    const int C = v->dim, tok0 = 1 + v->n_reg;
    const long M = (long)b * ntp;
    const long c128 = rup(C, 128), c2 = rup(2L * C, 128);
    const long rows = rup(M, 256) + 256;  // allocation contract of cvx_vit_ws
    if (!ws->xh || !ws->xl || !ws->stat_part || !ws->rowstat) return cvx_fail("vit_encode: ln_fold needs xh, xl, stat_part and rowstat workspaces");
    if (C % 64) return cvx_fail("vit_encode: ln_fold needs dim % 64 == 0");
    CVX_TRY(cvx_split_stream((const float*)ws->x, C, ws->xh, ws->xl, C, ws->rowstat, M, C, v->ln_eps, st));
    auto resid = [&](const void* a, long lda, const void* w, long kpad, const float* bias, const float* gamma) {
        cvx_gemm_desc d = gemm_base(CVX_EPI_RESID_HL, a, lda, w, kpad, M, C, c128, kpad, ws->xh, C, bias);
        d.gamma = gamma; d.out2 = ws->xl; d.stat_part = ws->stat_part; d.stat_rows = rows;
        return cvx_gemm_bf16(&d, st);
    };
    for (int i = 0; i < v->depth; ++i) {
        const cvx_vit_layer* L = &v->layers[i];
        if (v->qkv_merged) {  // ONE pass over hi for Q, K and V; attention reads V row-major from the same buffer
            const long c3 = rup(3L * C, 128);
            cvx_gemm_desc d = gemm_base(CVX_EPI_BF16, ws->xh, C, L->qk_w, C, M, 3L * C, c3, C, ws->qk, 3L * C, L->qk_b);
            d.ln_rowstat = ws->rowstat;
            CVX_TRY(cvx_gemm_bf16(&d, st));
            CVX_TRY(cvx_attention_qkv_bf16(ws->qk, 3L * C, ws->ao, C, b, v->heads, nt, ntp, st));
        } else {
        {
            cvx_gemm_desc d = gemm_base(CVX_EPI_BF16, ws->xh, C, L->qk_w, C, M, 2L * C, c2, C, ws->qk, 2L * C, L->qk_b);
            d.ln_rowstat = ws->rowstat;
            CVX_TRY(cvx_gemm_bf16(&d, st));
        }
        {
            cvx_gemm_desc d = gemm_base(CVX_EPI_VT, ws->xh, C, L->v_w, C, M, C, c128, C, ws->vt, 0, L->v_b);
            d.heads = v->heads; d.ntp = ntp; d.kp = kp; d.ln_rowstat = ws->rowstat;
            CVX_TRY(cvx_gemm_bf16(&d, st));
        }
        CVX_TRY(cvx_attention_bf16(ws->qk, 2L * C, ws->vt, ws->ao, C, b, v->heads, nt, ntp, kp, st));
        }
        CVX_TRY(resid(ws->ao, C, L->proj_w, C, L->proj_b, L->ls1));
        CVX_TRY(cvx_rowstat_finalize(ws->stat_part, C / 64, rows, ws->rowstat, M, C, v->ln_eps, st));
        {
            const long n1 = v->ffn_swiglu ? 2L * v->hid_pad : (long)v->hid_pad;
            cvx_gemm_desc d = gemm_base(v->ffn_swiglu ? CVX_EPI_SWIGLU : CVX_EPI_BF16_GELU, ws->xh, C, L->ffn1_w, C, M, n1, n1, C, ws->hid,
                                        v->hid_pad, L->ffn1_b);
            d.ln_rowstat = ws->rowstat;
            CVX_TRY(cvx_gemm_bf16(&d, st));
        }
        CVX_TRY(resid(ws->hid, v->hid_pad, L->ffn2_w, v->hid_pad, L->ffn2_b, L->ls2));
        if (i + 1 < v->depth)  // (the final LayerNorm computes its own statistics from the rows it reads)
            CVX_TRY(cvx_rowstat_finalize(ws->stat_part, C / 64, rows, ws->rowstat, M, C, v->ln_eps, st));
    }
    return cvx_final_norm_features_hl(ws->xh, ws->xl, C, v->norm_w, v->norm_b, v->ln_eps, b, ntp, tok0, hp, wp, C, feats_f16, d_total, d0,
                                      feats_cl, tokens_f32, st);
}

extern "C" int cvx_vit_encode(const cvx_vit_desc* v, const cvx_vit_ws* ws, int b, int hp, int wp, const void* patches, long patches_ld,
                              const void* pe_w, const float* pos, const float* cls_pos0, void* feats_f16, long d_total, long d0,
                              void* feats_cl, float* tokens_f32, hipStream_t st) {
    if (!v || !ws || !v->layers) return cvx_fail("vit_encode: null descriptor");
    if (b <= 0) return 0;
    if (v->dim % 128 || v->dim / v->heads != 64 || v->dim % v->heads) return cvx_fail("vit_encode: dim must be heads*64 and a multiple of 128");
    const int C = v->dim, npatch = hp * wp, tok0 = 1 + v->n_reg, nt = npatch + tok0;
    const int ntp = (int)rup(nt, 8), kp = (int)rup(nt, 64);
    const long M = (long)b * ntp;
    const long c128 = rup(C, 128), c2 = rup(2L * C, 128);
    CVX_TRY(cvx_init_tokens((float*)ws->x, C, cls_pos0, v->reg, v->n_reg, b, nt, ntp, C, st));
    {
        cvx_gemm_desc d = gemm_base(CVX_EPI_PATCH, patches, patches_ld, pe_w, patches_ld, (long)b * npatch, C, c128, patches_ld, ws->x, C, v->pe_b);
        d.pos = pos; d.ldpos = C; d.npatch = npatch; d.ntp = ntp; d.tok0 = tok0;
        CVX_TRY(cvx_gemm_bf16(&d, st));
    }
    if (v->ln_fold) return vit_blocks_folded(v, ws, b, hp, wp, nt, ntp, kp, feats_f16, d_total, d0, feats_cl, tokens_f32, st);
    for (int i = 0; i < v->depth; ++i) {
        const cvx_vit_layer* L = &v->layers[i];
        CVX_TRY(cvx_layernorm_bf16((const float*)ws->x, C, L->ln1_w, L->ln1_b, ws->xn, C, M, C, v->ln_eps, st));
        {
            cvx_gemm_desc d = gemm_base(CVX_EPI_BF16, ws->xn, C, L->qk_w, C, M, 2L * C, c2, C, ws->qk, 2L * C, L->qk_b);
            CVX_TRY(cvx_gemm_bf16(&d, st));
        }
        {
            cvx_gemm_desc d = gemm_base(CVX_EPI_VT, ws->xn, C, L->v_w, C, M, C, c128, C, ws->vt, 0, L->v_b);
            d.heads = v->heads; d.ntp = ntp; d.kp = kp;
            CVX_TRY(cvx_gemm_bf16(&d, st));
        }
        CVX_TRY(cvx_attention_bf16(ws->qk, 2L * C, ws->vt, ws->ao, C, b, v->heads, nt, ntp, kp, st));
        {
            cvx_gemm_desc d = gemm_base(CVX_EPI_RESID, ws->ao, C, L->proj_w, C, M, C, c128, C, ws->x, C, L->proj_b);
            d.gamma = L->ls1;
            CVX_TRY(cvx_gemm_bf16(&d, st));
        }
        CVX_TRY(cvx_layernorm_bf16((const float*)ws->x, C, L->ln2_w, L->ln2_b, ws->xn, C, M, C, v->ln_eps, st));
        if (v->ffn_swiglu) {
            cvx_gemm_desc d = gemm_base(CVX_EPI_SWIGLU, ws->xn, C, L->ffn1_w, C, M, 2L * v->hid_pad, 2L * v->hid_pad, C, ws->hid, v->hid_pad, L->ffn1_b);
            CVX_TRY(cvx_gemm_bf16(&d, st));
        } else {
            cvx_gemm_desc d = gemm_base(CVX_EPI_BF16_GELU, ws->xn, C, L->ffn1_w, C, M, v->hid_pad, v->hid_pad, C, ws->hid, v->hid_pad, L->ffn1_b);
            CVX_TRY(cvx_gemm_bf16(&d, st));
        }
        {
            cvx_gemm_desc d = gemm_base(CVX_EPI_RESID, ws->hid, v->hid_pad, L->ffn2_w, v->hid_pad, M, C, c128, v->hid_pad, ws->x, C, L->ffn2_b);
            d.gamma = L->ls2;
            CVX_TRY(cvx_gemm_bf16(&d, st));
        }
    }
    return cvx_final_norm_features((const float*)ws->x, C, v->norm_w, v->norm_b, v->ln_eps, b, ntp, tok0, hp, wp, C, feats_f16, d_total, d0,
                                   feats_cl, tokens_f32, st);
}


extern "C" int cvx_head_forward(const cvx_head_desc* hd, const cvx_head_ws* ws, const void* feats_cl, int D, int h, int w, float* logits,
                                float* probs, const int8_t* labels, float* dice, uint8_t* mask, float mask_threshold, hipStream_t st) {
    if (!hd || !ws || !hd->blocks) return cvx_fail("head_forward: null descriptor");
    if (hd->n_blocks < 1 || hd->n_blocks > CVX_HEAD_MAX_BLOCKS) return cvx_fail("head_forward: 1..8 synthesis blocks supported");
    if (hd->c_tail != 8) return cvx_fail("head_forward: the fused output kernel is specialised for 8 tail channels");
    if (D <= 0 || h <= 0 || w <= 0) return 0;
    long nv = (long)D * h * w;
    {
        cvx_gemm_desc d = gemm_base(CVX_EPI_BF16_GELU, feats_cl, hd->c_in, hd->proj_w, hd->proj_kpad, nv, hd->c0, hd->proj_npad, hd->proj_kpad,
                                    ws->act0, hd->c0, hd->proj_b);
        d.dtype = CVX_DTYPE_F16;  // the whole head stores fp16 (features, activations, GEMM-side weights)
        CVX_TRY(cvx_gemm_bf16(&d, st));
    }
    const void* act = ws->act0;
    int H = h, W = w;
    for (int i = 0; i < hd->n_blocks; ++i) {
        const cvx_head_block* B = &hd->blocks[i];
        nv = (long)D * H * W;
        CVX_TRY(cvx_groupnorm_f16(act, B->gn_w, B->gn_b, ws->gn[i], ws->gn_stats, nv, B->c1, B->groups, 1e-3f, st));
        {
            cvx_conv3d_desc c = {ws->gn[i], B->conv1_w, B->conv1_b, hd->zero_page, ws->t1[i], B->c1, D, H, W, B->d1, B->c2, B->conv1_npad,
                                 B->conv1_kpad, 1};
            CVX_TRY(cvx_conv3d_f16(&c, st));
        }
        {
            cvx_conv3d_desc c = {ws->t1[i], B->conv2_w, B->conv2_b, hd->zero_page, ws->t2[i], B->c2, D, H, W, B->d2, B->c2, B->conv2_npad,
                                 B->conv2_kpad, 1};
            CVX_TRY(cvx_conv3d_f16(&c, st));
        }
        {
            cvx_gemm_desc d = gemm_base(CVX_EPI_CONVT, ws->t2[i], B->c2, B->convt_w, B->convt_kpad, nv, 4L * B->c3, B->convt_npad, B->convt_kpad,
                                        ws->up[i], B->c3, B->convt_b);
            d.H = H; d.W = W; d.cout = B->c3; d.act = 1; d.dtype = CVX_DTYPE_F16;
            CVX_TRY(cvx_gemm_bf16(&d, st));
        }
        act = ws->up[i];
        H *= 2; W *= 2;
    }
    {
        cvx_conv3d_desc c = {act, hd->out0_w, hd->out0_b, hd->zero_page, ws->mid, hd->c_tail, D, H, W, 1, hd->c_tail, hd->out0_npad,
                             hd->out0_kpad, 1};
        CVX_TRY(cvx_conv3d_f16(&c, st));
    }
    return cvx_conv3_out_fused(ws->mid, hd->out2_w, hd->out2_b, logits, probs, labels, dice, ws->dice_scratch, mask, mask_threshold, D, H,
                               W, st);
}
