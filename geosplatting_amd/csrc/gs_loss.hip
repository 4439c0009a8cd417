// gs_loss.hip -- SURVEY section 8f rank 2: the loss side, the step immediately after the render path
// (rfstudio/trainer/geosplat_trainer.py:171-195): per view
//     x = rgb + (1 - alpha) * bg                      bg = per-pixel random background (torch.rand_like)
//     y = gt_lin * mask + (1 - mask) * bg             gt_lin = srgb2rgb(gt)  (graphics/_images.py:287-311), mask = gt alpha
//     loss = lambda * (1 - SSIM(y, x)) + (1 - lambda) * L1(x, y) + mask_weight * mean((mask - alpha)^2)
// with SSIM = torchmetrics.functional.structural_similarity_index_measure defaults (pinned torchmetrics~=1.3.1;
// not vendored in the reference, restated from its published algorithm): 11x11 Gaussian window, sigma 1.5,
// k1 0.01, k2 0.03, data_range 1, variances clamped at 0, reflect padding followed by a crop of the same 5-pixel
// border -> only windows that lie entirely inside the image contribute, the mean is over (H-10)(W-10)*3 values.
// The sRGB-space PSNR metric of the same view (:191-195) falls out of the first pass.
//
// The loss is the end of the graph, so value AND gradient come out of one call -- two LDS-tiled passes:
//   pass 1 (16x16 tiles, 26x26 halo): build x,y on the fly, separable blur of {x, y, xx, yy, xy}, SSIM value and
//           its partials w.r.t. (mu_x, E[xx], E[xy]) pre-multiplied by the loss weight -> 9 planar maps; per-block
//           partial sums of SSIM / L1 / mask-MSE / sRGB-SSE;
//   pass 2: blur the three partial maps with the same (symmetric) window and apply the chain rule
//           dL/dx = G*(dmu) + 2x G*(dExx) + y G*(dExy) + L1 term;  v_rgb = dL/dx, v_alpha = -bg . dL/dx + mask term;
//   finalize: one block sums the partials in a fixed order in double (deterministic).
// HBM-bound: ~28 B/px read + 36 B/px map write (pass 1), 36+28 B/px read + 16 B/px write (pass 2).
#include "gs_common.h"

#define LT 16
#define LHALO 5
#define LS (LT + 2 * LHALO)

__device__ __forceinline__ float gs_win(int k)
{
    // exp(-(d/1.5)^2/2) normalised, d = -5..5, exactly the fp32 values torchmetrics' _gaussian produces
    constexpr float w[11] = { 0.0010283803567290306f, 0.0075987558811903f, 0.036000773310661316f, 0.10936067998409271f,
                              0.21300552785396576f,   0.26601171493530273f, 0.21300552785396576f, 0.10936067998409271f,
                              0.036000773310661316f,  0.0075987558811903f,  0.0010283803567290306f };
    return w[k];
}

__device__ __forceinline__ float srgb_to_linear(float c)
{
    return c <= 0.04045f ? c / 12.92f : powf((fmaxf(c, 0.04045f) + 0.055f) / 1.055f, 2.4f);
}
__device__ __forceinline__ float linear_to_srgb(float c)
{
    return c <= 0.0031308f ? c * 12.92f : powf(fmaxf(c, 0.0031308f), 1.0f / 2.4f) * 1.055f - 0.055f;
}

struct LossPx {
    float x[3], y[3], bg[3], a, m;
};

__device__ __forceinline__ LossPx loss_load(int px, int py, int W, const float* __restrict__ rgb,
                                            const float* __restrict__ alpha, const float* __restrict__ gt,
                                            int gt_is_srgb, const float* __restrict__ bg)
{
    LossPx p;
    const size_t i = (size_t)py * W + px;
    const float4 g = *reinterpret_cast<const float4*>(gt + 4 * i);
    p.a = alpha[i];
    p.m = g.w;
    const float gl[3] = { g.x, g.y, g.z };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p.bg[c] = bg[3 * i + c];
        const float lin = gt_is_srgb ? srgb_to_linear(gl[c]) : gl[c];
        p.x[c] = rgb[3 * i + c] + (1.0f - p.a) * p.bg[c];
        p.y[c] = lin * p.m + (1.0f - p.m) * p.bg[c];
    }
    return p;
}

__global__ void __launch_bounds__(256)
loss_stats_kernel(int W, int H, const float* __restrict__ rgb, const float* __restrict__ alpha,
                  const float* __restrict__ gt, int gt_is_srgb, const float* __restrict__ bg,
                  const float* __restrict__ metric_bg, float w_ssim, float* __restrict__ maps, float* __restrict__ partials)
{
    __shared__ float sx[3][LS][LS + 1], sy[3][LS][LS + 1];
    __shared__ float tmp[5][LS][LT];
    __shared__ float red[4][4];
    const int tid = threadIdx.x, tx = tid % LT, ty = tid / LT;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    for (int i = tid; i < LS * LS; i += 256) {
        const int r = i / LS, c = i % LS, gx = x0 - LHALO + c, gy = y0 - LHALO + r;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
            const LossPx p = loss_load(gx, gy, W, rgb, alpha, gt, gt_is_srgb, bg);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { sx[ch][r][c] = p.x[ch]; sy[ch][r][c] = p.y[ch]; }
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { sx[ch][r][c] = 0.0f; sy[ch][r][c] = 0.0f; }
        }
    }
    __syncthreads();
    const int px = x0 + tx, py = y0 + ty;
    const bool inside = px < W && py < H;
    const bool valid = inside && px >= LHALO && px < W - LHALO && py >= LHALO && py < H - LHALO;
    const size_t plane = (size_t)W * H, pix = (size_t)py * W + px;
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    float s_ssim = 0.0f, s_l1 = 0.0f, s_mask = 0.0f, s_sse = 0.0f;
    for (int ch = 0; ch < 3; ++ch) {
        for (int i = tid; i < LS * LT; i += 256) {
            const int r = i / LT, c = i % LT;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, a4 = 0.0f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float w = gs_win(k), xv = sx[ch][r][c + k], yv = sy[ch][r][c + k];
                a0 += w * xv; a1 += w * yv; a2 += w * (xv * xv); a3 += w * (yv * yv); a4 += w * (xv * yv);
            }
            tmp[0][r][c] = a0; tmp[1][r][c] = a1; tmp[2][r][c] = a2; tmp[3][r][c] = a3; tmp[4][r][c] = a4;
        }
        __syncthreads();
        float mu_x = 0.0f, mu_y = 0.0f, exx = 0.0f, eyy = 0.0f, exy = 0.0f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = gs_win(k);
            mu_x += w * tmp[0][ty + k][tx]; mu_y += w * tmp[1][ty + k][tx];
            exx += w * tmp[2][ty + k][tx]; eyy += w * tmp[3][ty + k][tx]; exy += w * tmp[4][ty + k][tx];
        }
        __syncthreads();
        const float mxx = mu_x * mu_x, myy = mu_y * mu_y, mxy = mu_x * mu_y;
        const float vxr = exx - mxx, vyr = eyy - myy, cxy = exy - mxy;
        const float vx = fmaxf(vxr, 0.0f), vy = fmaxf(vyr, 0.0f);
        const float A1 = 2.0f * mxy + c1, A2 = 2.0f * cxy + c2, B1 = mxx + myy + c1, B2 = vx + vy + c2;
        const float inv = 1.0f / (B1 * B2);
        const float S = A1 * A2 * inv;
        const float dvx = vxr >= 0.0f ? -S / B2 : 0.0f;
        const float dcxy = 2.0f * A1 * inv;
        const float dmu = 2.0f * mu_y * A2 * inv - S * 2.0f * mu_x / B1 - dcxy * mu_y - dvx * 2.0f * mu_x;
        const float wv = valid ? w_ssim : 0.0f;
        if (inside && maps) {
            maps[(size_t)(ch * 3 + 0) * plane + pix] = wv * dmu;
            maps[(size_t)(ch * 3 + 1) * plane + pix] = wv * dvx;
            maps[(size_t)(ch * 3 + 2) * plane + pix] = wv * dcxy;
        }
        if (valid) s_ssim += S;
        if (inside) s_l1 += fabsf(sx[ch][ty + LHALO][tx + LHALO] - sy[ch][ty + LHALO][tx + LHALO]);
    }
    if (inside) {
        const float a = alpha[pix];
        const float4 g = *reinterpret_cast<const float4*>(gt + 4 * pix);
        s_mask = (g.w - a) * (g.w - a);
        if (metric_bg) {
            // sRGB-space PSNR (geosplat_trainer.py:191-195): rgb2srgb(render) over bg (clamped) vs the sRGB ground truth over bg
            const float gs[3] = { g.x, g.y, g.z };
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float gsr = gt_is_srgb ? gs[ch] : linear_to_srgb(gs[ch]);
                const float o = fminf(fmaxf(linear_to_srgb(rgb[3 * pix + ch]) + (1.0f - a) * metric_bg[ch], 0.0f), 1.0f);
                const float t = g.w * gsr + metric_bg[ch] * (1.0f - g.w);
                s_sse += (o - t) * (o - t);
            }
        }
    }
    s_ssim = gs_wave_sum(s_ssim); s_l1 = gs_wave_sum(s_l1); s_mask = gs_wave_sum(s_mask); s_sse = gs_wave_sum(s_sse);
    const int wave = tid >> 6;
    if ((tid & 63) == 0) { red[wave][0] = s_ssim; red[wave][1] = s_l1; red[wave][2] = s_mask; red[wave][3] = s_sse; }
    __syncthreads();
    if (tid < 4) {
        const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partials[4 * b + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

__global__ void __launch_bounds__(256)
loss_grad_kernel(int W, int H, const float* __restrict__ rgb, const float* __restrict__ alpha,
                 const float* __restrict__ gt, int gt_is_srgb, const float* __restrict__ bg,
                 const float* __restrict__ maps, float w_l1, float w_mask, float* __restrict__ v_rgb,
                 float* __restrict__ v_alpha)
{
    __shared__ float sm[3][LS][LS + 1];
    __shared__ float tmp[3][LS][LT];
    const int tid = threadIdx.x, tx = tid % LT, ty = tid / LT;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    const int px = x0 + tx, py = y0 + ty;
    const bool inside = px < W && py < H;
    const size_t plane = (size_t)W * H, pix = (size_t)py * W + px;
    LossPx p;
    if (inside) p = loss_load(px, py, W, rgb, alpha, gt, gt_is_srgb, bg);
    float va = 0.0f;
    for (int ch = 0; ch < 3; ++ch) {
        for (int i = tid; i < LS * LS; i += 256) {
            const int r = i / LS, c = i % LS, gx = x0 - LHALO + c, gy = y0 - LHALO + r;
            const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
            const size_t o = (size_t)gy * W + gx;
#pragma unroll
            for (int q = 0; q < 3; ++q) sm[q][r][c] = in ? maps[(size_t)(ch * 3 + q) * plane + o] : 0.0f;
        }
        __syncthreads();
        for (int i = tid; i < LS * LT; i += 256) {
            const int r = i / LT, c = i % LT;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float w = gs_win(k);
                a0 += w * sm[0][r][c + k]; a1 += w * sm[1][r][c + k]; a2 += w * sm[2][r][c + k];
            }
            tmp[0][r][c] = a0; tmp[1][r][c] = a1; tmp[2][r][c] = a2;
        }
        __syncthreads();
        float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = gs_win(k);
            b0 += w * tmp[0][ty + k][tx]; b1 += w * tmp[1][ty + k][tx]; b2 += w * tmp[2][ty + k][tx];
        }
        __syncthreads();
        if (inside) {
            const float d = p.x[ch] - p.y[ch];
            const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
            const float g = b0 + 2.0f * p.x[ch] * b1 + p.y[ch] * b2 + w_l1 * sgn;
            v_rgb[3 * pix + ch] = g;
            va -= p.bg[ch] * g;
        }
    }
    if (inside) v_alpha[pix] = va + w_mask * (p.a - p.m);
}

__global__ void __launch_bounds__(256)
loss_finalize_kernel(int nblocks, const float* __restrict__ partials, double n_ssim, double n_px, float ssim_lambda,
                     float mask_weight, float* __restrict__ out)
{
    __shared__ double red[256][4];
    double s[4] = { 0.0, 0.0, 0.0, 0.0 };
    for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += (double)partials[4 * (size_t)b + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x][k] = s[k];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[threadIdx.x][k] += red[threadIdx.x + st][k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double ssim_loss = 1.0 - red[0][0] / n_ssim, l1 = red[0][1] / (3.0 * n_px), mmse = red[0][2] / n_px;
        const double mse = red[0][3] / (3.0 * n_px);
        out[0] = (float)(ssim_lambda * ssim_loss + (1.0 - ssim_lambda) * l1 + mask_weight * mmse);
        out[1] = (float)ssim_loss; out[2] = (float)l1; out[3] = (float)mmse; out[4] = (float)mse;
        out[5] = (float)(-10.0 * log10(mse));
    }
}

extern "C" size_t gs_photo_loss_ws_bytes(int W, int H)
{
    const size_t nb = (size_t)gs_cdiv(W, LT) * gs_cdiv(H, LT);
    return sizeof(float) * (9 * (size_t)W * H + 4 * nb);
}

extern "C" int gs_photo_loss(int W, int H, const float* rgb, const float* alpha, const float* gt_rgba, int gt_is_srgb,
                             const float* train_bg, const float* metric_bg, float ssim_lambda, float mask_weight,
                             float grad_scale, float* out, float* v_rgb, float* v_alpha, void* ws, size_t ws_bytes,
                             void* stream)
{
    GS_CHECK_ARG(W > 2 * LHALO && H > 2 * LHALO, "SSIM needs W, H > 10 (11x11 window; torchmetrics yields NaN otherwise)");
    GS_CHECK_ARG(ws_bytes >= gs_photo_loss_ws_bytes(W, H), "workspace too small");
    GS_CHECK_ARG((v_rgb == nullptr) == (v_alpha == nullptr), "v_rgb and v_alpha go together");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(gs_cdiv(W, LT), gs_cdiv(H, LT));
    const int nb = grid.x * grid.y;
    float* maps = (float*)ws;
    float* partials = maps + 9 * (size_t)W * H;
    const double n_ssim = 3.0 * (double)(W - 2 * LHALO) * (double)(H - 2 * LHALO), n_px = (double)W * H;
    const float w_ssim = (float)(-(double)ssim_lambda / n_ssim * grad_scale);
    hipLaunchKernelGGL(loss_stats_kernel, grid, dim3(256), 0, s, W, H, rgb, alpha, gt_rgba, gt_is_srgb, train_bg, metric_bg,
                       w_ssim, v_rgb ? maps : (float*)nullptr, partials);
    if (v_rgb) {
        const float w_l1 = (float)((1.0 - (double)ssim_lambda) / (3.0 * n_px) * grad_scale);
        const float w_mask = (float)(2.0 * (double)mask_weight / n_px * grad_scale);
        hipLaunchKernelGGL(loss_grad_kernel, grid, dim3(256), 0, s, W, H, rgb, alpha, gt_rgba, gt_is_srgb, train_bg, maps,
                           w_l1, w_mask, v_rgb, v_alpha);
    }
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, nb, partials, n_ssim, n_px, ssim_lambda, mask_weight, out);
    GS_CHECK_LAUNCH();
    return GS_OK;
}
