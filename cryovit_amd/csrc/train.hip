// Training-side pieces of the segmentation head: the Dice loss of models/losses.py:8-32 with its
// gradient, and the AdamW update the reference's BaseModel.configure_optimizers asks torch for (models/base_model.py:57-63).
// All HBM-bound streaming kernels: 16-B accesses, grid-stride, no atomics (block partials + a fixed-order finalize, so a loss
// value and every gradient are bitwise reproducible).
#include "common.h"
#include "../../include/cryovit_hip.h"
#include "host_util.h"

namespace cvx {

// ---- Dice loss: loss = 1 - 2 I / (Sy + Sp + 1e-3), I = sum y p, Sy = sum y, Sp = sum p over voxels with label > -1 ----
// (the reference applies the mask by torch.masked_select before the loss, base_model.py:91-112; here the mask is the label's sign)
__global__ __launch_bounds__(256) void k_dice_loss_partials(const float* __restrict__ probs, const int8_t* __restrict__ labels, long n,
                                                            float* __restrict__ partials) {
    __shared__ float red[3][4];
    float inter = 0.f, ysum = 0.f, psum = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 p = *(const float4*)(probs + 4 * i);
        const uint32_t lw = *(const uint32_t*)(labels + 4 * i);
        const float pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int lab = (int)(int8_t)(lw >> (8 * e));
            if (lab > -1) { inter += (float)lab * pv[e]; ysum += (float)lab; psum += pv[e]; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // the last 1..3 elements
        const long i = (n4 << 2) + threadIdx.x;
        const int lab = labels[i];
        if (lab > -1) { inter += (float)lab * probs[i]; ysum += (float)lab; psum += probs[i]; }
    }
    inter = wave_sum(inter); ysum = wave_sum(ysum); psum = wave_sum(psum);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = inter; red[1][wave] = ysum; red[2][wave] = psum; }
    __syncthreads();
    if (threadIdx.x < 3) partials[(long)blockIdx.x * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// out[0..2] = I, Sy, Sp (fp64 accumulation of the block partials in a fixed order), out[3] = the loss
__global__ __launch_bounds__(64) void k_dice_loss_finalize(const float* __restrict__ partials, int nblk, float* __restrict__ out) {
    const int lane = threadIdx.x;
    double s[3];
    for (int k = 0; k < 3; ++k) {
        double a = 0.0;
        for (int b = lane; b < nblk; b += 64) a += (double)partials[(long)b * 3 + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        s[k] = a;
    }
    if (lane == 0) {
        const float I = (float)s[0], Sy = (float)s[1], Sp = (float)s[2];
        out[0] = I; out[1] = Sy; out[2] = Sp;
        out[3] = 1.0f - (2.0f * I) / ((Sy + Sp) + 1e-3f);
    }
}

// d loss / d p_i = -2 y_i / den + 2 I / den^2 for label > -1 (0 elsewhere), times the upstream gradient; through == 1 chains it
// through p = sigmoid(clip(logit, -5, 5)): * p (1 - p), and zero where the stored (clipped) logit sits on a clip boundary
__global__ __launch_bounds__(256) void k_dice_loss_backward(const float* __restrict__ probs, const float* __restrict__ logits,
                                                            const int8_t* __restrict__ labels, long n, const float* __restrict__ sums,
                                                            float gout, int through, float* __restrict__ grad) {
    const float I = sums[0], den = (sums[1] + sums[2]) + 1e-3f;
    const float ga = -2.0f * gout / den, gb = 2.0f * gout * I / (den * den);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int lab = labels[i];
        float g = 0.f;
        if (lab > -1) {
            g = fmaf((float)lab, ga, gb);
            if (through) {
                const float p = probs[i];
                g *= p * (1.0f - p);
                if (logits && fabsf(logits[i]) >= 5.0f) g = 0.f;
            }
        }
        grad[i] = g;
    }
}

// ---- AdamW, decoupled weight decay; the order of operations of torch.optim.AdamW's single-tensor path:
//   p *= 1 - lr wd;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, long n, float decay, float w1, float beta2, float w2, float step_size,
                                               float bc2_sqrt, float eps) {
    auto upd = [&](float& pi, float gi, float& mi, float& vi) {
        pi *= decay;
        mi = mi + w1 * (gi - mi);           // lerp(m, g, 1 - beta1), weight < 0.5 form
        vi = vi * beta2 + (w2 * gi) * gi;   // mul_(beta2).addcmul_(g, g, value = 1 - beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi + (-step_size * mi) / denom; // addcdiv_(m, denom, value = -step_size)
    };
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 P = *(float4*)(p + 4 * i), M = *(float4*)(m + 4 * i), V = *(float4*)(v + 4 * i);
        const float4 G = *(const float4*)(g + 4 * i);
        upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
        *(float4*)(p + 4 * i) = P; *(float4*)(m + 4 * i) = M; *(float4*)(v + 4 * i) = V;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        upd(p[i], g[i], m[i], v[i]);
    }
}


// ---- Focal loss (models/losses.py:35-64 -> torchvision.ops.sigmoid_focal_loss, reduction "mean"): per element, x = the model output
// the reference passes as "logits", t the label:  p = sigmoid(x), ce = max(x,0) - x t + log(1 + exp(-|x|)), p_t = p t + (1-p)(1-t),
// loss = alpha_t (1 - p_t)^gamma ce with alpha_t = alpha t + (1-alpha)(1-t) and alpha = (n - sum t) / n over the labelled voxels
// (losses.py:58: weight.item() -- a constant for the gradient).
__device__ __forceinline__ void focal_terms(float x, float t, float alpha, float gamma, float& loss, float& dldx) {
    const float p = 1.0f / (1.0f + __expf(-x));
    const float ce = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
    const float pt = p * t + (1.0f - p) * (1.0f - t);
    const float m = 1.0f - pt;
    const float at = alpha * t + (1.0f - alpha) * (1.0f - t);
    // m^(gamma-1) and m^gamma.  gamma == 0 is plain weighted cross-entropy: m^0 = 1 also at m = 0 (the product form
    // m^(-1) * m gave 0 there), and the d m^gamma term vanishes
    const float mg1 = gamma == 2.0f ? m : (gamma == 0.0f ? 0.0f : powf(fmaxf(m, 1e-30f), gamma - 1.0f));
    const float mg = gamma == 0.0f ? 1.0f : mg1 * m;
    loss = at * mg * ce;
    const float dm = -(2.0f * t - 1.0f) * p * (1.0f - p);
    dldx = at * (gamma * mg1 * dm * ce + mg * (p - t));
}

// mode 0: partials = (count, sum t, 0) over labels > -1;  mode 1: partials = (sum loss, 0, 0) with alpha from stats4
__global__ __launch_bounds__(256) void k_focal_partials(const float* __restrict__ x, const int8_t* __restrict__ labels, long n, int mode,
                                                        const float* __restrict__ stats4, float gamma, float* __restrict__ partials) {
    __shared__ float red[2][4];
    float a = 0.f, b = 0.f;
    const float alpha = mode ? stats4[2] : 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int lab = labels[i];
        if (lab > -1) {
            if (mode == 0) { a += 1.0f; b += (float)lab; }
            else { float l, g; focal_terms(x[i], (float)lab, alpha, gamma, l, g); a += l; }
        }
    }
    a = wave_sum(a); b = wave_sum(b);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x < 2) partials[(long)blockIdx.x * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    if (threadIdx.x == 2) partials[(long)blockIdx.x * 3 + 2] = 0.f;
}

// mode 0: out4 = {count, sum t, alpha = (count - sum t) / count, -};  mode 1: out4[3] = sum loss / count
__global__ __launch_bounds__(64) void k_focal_finalize(const float* __restrict__ partials, int nblk, int mode, float* __restrict__ out4) {
    const int lane = threadIdx.x;
    double s[2];
    for (int k = 0; k < 2; ++k) {
        double a = 0.0;
        for (int b = lane; b < nblk; b += 64) a += (double)partials[(long)b * 3 + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        s[k] = a;
    }
    if (lane == 0) {
        if (mode == 0) {
            const float cnt = (float)s[0], st = (float)s[1];
            out4[0] = cnt; out4[1] = st; out4[2] = cnt > 0.f ? (cnt - st) / cnt : 0.f;
        } else {
            out4[3] = out4[0] > 0.f ? (float)s[0] / out4[0] : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void k_focal_backward(const float* __restrict__ x, const int8_t* __restrict__ labels, long n,
                                                        const float* __restrict__ stats4, float gamma, float gout, float* __restrict__ grad) {
    const float alpha = stats4[2], scale = stats4[0] > 0.f ? gout / stats4[0] : 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int lab = labels[i];
        float g = 0.f;
        if (lab > -1) { float l; focal_terms(x[i], (float)lab, alpha, gamma, l, g); g *= scale; }
        grad[i] = g;
    }
}

}  // namespace cvx

using namespace cvx;

static unsigned stream_blocks(long n, long per_block) {
    long b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (unsigned)(b < CVX_DICE_BLOCKS ? b : CVX_DICE_BLOCKS);
}

extern "C" int cvx_dice_loss_forward(const float* probs, const int8_t* labels, long n, float* scratch, float* out4, hipStream_t st) {
    if (!probs || !labels || !scratch || !out4) return cvx_fail("dice_loss_forward: null pointer");
    if (n < 0) return cvx_fail("dice_loss_forward: n < 0");
    if (((uintptr_t)probs & 15) || ((uintptr_t)labels & 3)) return cvx_fail("dice_loss_forward: probs must be 16-B and labels 4-B aligned");
    const unsigned nblk = stream_blocks(n, 4096);
    hipLaunchKernelGGL(k_dice_loss_partials, dim3(nblk), dim3(256), 0, st, probs, labels, n, scratch);
    int rc = cvx_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(k_dice_loss_finalize, dim3(1), dim3(64), 0, st, scratch, (int)nblk, out4);
    return cvx_check_launch();
}

extern "C" int cvx_dice_loss_backward(const float* probs, const float* logits, const int8_t* labels, long n, const float* sums4,
                                      float grad_out, int through_sigmoid, float* grad, hipStream_t st) {
    if (!labels || !sums4 || !grad || (through_sigmoid && !probs)) return cvx_fail("dice_loss_backward: null pointer");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_dice_loss_backward, dim3(stream_blocks(n, 2048)), dim3(256), 0, st, probs, logits, labels, n, sums4, grad_out,
                       through_sigmoid, grad);
    return cvx_check_launch();
}

extern "C" int cvx_adamw_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2, double eps,
                              double weight_decay, int step, hipStream_t st) {
    if (!p || !g || !m || !v) return cvx_fail("adamw_step: null pointer");
    if (n <= 0) return 0;
    if (step < 1) return cvx_fail("adamw_step: step counts from 1");
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)) return cvx_fail("adamw_step: buffers must be 16-B aligned");
    // scalar factors in double, as torch's Python does, then one rounding to fp32
    // (hyper-parameters arrive as doubles: 1 - beta2 formed from a float-rounded 0.999 would be off by 1.3e-5 relative)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float decay = (float)(1.0 - lr * weight_decay);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(k_adamw, dim3(stream_blocks(n, 2048)), dim3(256), 0, st, p, g, m, v, n, decay, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), step_size, bc2_sqrt, (float)eps);
    return cvx_check_launch();
}

extern "C" int cvx_focal_loss_forward(const float* x, const int8_t* labels, long n, float gamma, float* scratch, float* out4, hipStream_t st) {
    if (!x || !labels || !scratch || !out4) return cvx_fail("focal_loss_forward: null pointer");
    if (n < 0) return cvx_fail("focal_loss_forward: n < 0");
    const unsigned nblk = stream_blocks(n, 2048);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k_focal_partials, dim3(nblk), dim3(256), 0, st, x, labels, n, mode, out4, gamma, scratch);
        int rc = cvx_check_launch();
        if (rc) return rc;
        hipLaunchKernelGGL(k_focal_finalize, dim3(1), dim3(64), 0, st, scratch, (int)nblk, mode, out4);
        rc = cvx_check_launch();
        if (rc) return rc;
    }
    return 0;
}

extern "C" int cvx_focal_loss_backward(const float* x, const int8_t* labels, long n, float gamma, const float* stats4, float grad_out,
                                       float* grad, hipStream_t st) {
    if (!x || !labels || !stats4 || !grad) return cvx_fail("focal_loss_backward: null pointer");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_focal_backward, dim3(stream_blocks(n, 2048)), dim3(256), 0, st, x, labels, n, stats4, gamma, grad_out, grad);
    return cvx_check_launch();
}
