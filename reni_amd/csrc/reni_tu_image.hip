// translation unit of libreni_hip.so: HDR image epilogue / prologue of the RENI path.
//
// Reference:
//   UnMinMaxNormlise   src/utils/custom_transforms.py:14-21   y = exp(0.5 (x + 1)(m1 - m0) + m0)
//   MinMaxNormalise    src/utils/custom_transforms.py:4-12    clip(x, min positive, max finite) -> log -> 2 (. - m0)/(m1 - m0) - 1
//   sRGB               src/utils/utils.py:30-42               x / q_b -> clamp [0,1] -> sRGB transfer curve, with
//                      q_b = quantile_0.98 over W of ( quantile_0.98 over H of ( quantile_0.98 over C of x[b] ) )
//
// The model's output is a [B, P, 3] tensor of log-normalised radiance; a viewer (and the FIT_INVERSE renderer,
// RENI_module.py:108) needs linear HDR and its sRGB view.  One call does un-normalise -> nested quantile -> divide ->
// clamp -> gamma on the device, reading the model output through strides (channel-last or channel-planar alike), so
// nothing is permuted or copied on the host.
//
// torch.quantile semantics are kept exactly: rank = q * (n - 1) evaluated in fp32, value = lerp(sorted[floor],
// sorted[ceil], frac) with ATen's two-sided lerp.  Order statistics come from rank-by-counting in LDS (n <= 4096: every
// thread counts the elements smaller than its own, ties broken by index): exact, deterministic, O(n^2 / 256) per thread
// and the grids are small (B*W columns of H values, then B rows of W values) -- HBM traffic is the two passes over the
// image (12 + 12 B per pixel and channel), which is what bounds it.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "reni_hip.h"
#include "reni_internal.h"

#define DEV __device__ __forceinline__

namespace reni {

constexpr int QMAX = 4096;  // longest axis a quantile is taken over (LDS: 16 KB)

DEV float lerp_aten(float a, float b, float w) {  // at::native::lerp (Lerp.h): two-sided for monotonicity
  return (fabsf(w) < 0.5f) ? a + w * (b - a) : b - (b - a) * (1.f - w);
}

struct ImgArgs {
  const float* in;   // element (b, c, h, w) at in[b*sb + c*sc + h*sh + w*sw]
  long long sb, sc, sh, sw;
  float* lin;        // [B][3][H][W] linear HDR (un-normalised input, or a copy of the input)
  float* q1;         // [B][H][W] quantile over the channels
  float* out;        // [B][3][H][W] sRGB
  const float* q;    // [B] the nested quantile
  int B, H, W;
  int unnorm;        // 1: apply UnMinMaxNormlise first
  float range, m0;   // (float)(m1 - m0) of the double difference, (float)m0
  float qf;          // the quantile (0.98)
};

// pass 1: (un-normalise,) write the linear image, quantile over the 3 channels of every pixel
__global__ void __launch_bounds__(256) k_img_pass1(const ImgArgs a) {
  const long long n = (long long)a.B * a.H * a.W;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int w = (int)(i % a.W), h = (int)((i / a.W) % a.H), b = (int)(i / ((long long)a.W * a.H));
  float v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float x = a.in[b * a.sb + c * a.sc + h * a.sh + w * a.sw];
    // 0.5 * (img + 1) * (m1 - m0) + m0, rounded after every operation as the reference's tensor ops are
    if (a.unnorm) x = expf(__fadd_rn(__fmul_rn(__fmul_rn(0.5f, __fadd_rn(x, 1.f)), a.range), a.m0));
    v[c] = x;
    if (a.lin) a.lin[(((long long)b * 3 + c) * a.H + h) * a.W + w] = x;
  }
  if (a.q1) {
    float s[3] = {v[0], v[1], v[2]};  // three compare-exchanges
    if (s[0] > s[1]) { const float t = s[0]; s[0] = s[1]; s[1] = t; }
    if (s[1] > s[2]) { const float t = s[1]; s[1] = s[2]; s[2] = t; }
    if (s[0] > s[1]) { const float t = s[0]; s[0] = s[1]; s[1] = t; }
    const float rank = a.qf * 2.f;
    const float fl = floorf(rank);
    a.q1[i] = lerp_aten(s[(int)fl], s[(int)ceilf(rank)], rank - fl);
  }
}

// quantile of n values taken with stride `es` from base(o): one workgroup per output o
__global__ void __launch_bounds__(256) k_img_quantile(const float* in, int n, long long es, int inner, long long outer_stride,
                                                      long long inner_stride, float qf, float* out) {
  __shared__ float v[QMAX];
  __shared__ float pick[2];
  const int o = blockIdx.x;
  const float* base = in + (long long)(o / inner) * outer_stride + (long long)(o % inner) * inner_stride;
  // (a NaN in the column: every comparison below is false, ranks collide and a pick may never be written -- torch.quantile
  // propagates NaN, and so does this: the picks start as NaN and a column holding one keeps them)
  if (threadIdx.x < 2) pick[threadIdx.x] = __builtin_nanf("");
  __shared__ int has_nan;
  if (threadIdx.x == 0) has_nan = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const float x = base[(long long)i * es];
    v[i] = x;
    if (x != x) has_nan = 1;
  }
  __syncthreads();
  if (has_nan) {
    if (threadIdx.x == 0) out[o] = __builtin_nanf("");
    return;
  }
  const float rank = qf * (float)(n - 1);
  const float fl = floorf(rank);
  const int lo = (int)fl, hi = (int)ceilf(rank);
  for (int i = threadIdx.x; i < n; i += 256) {
    const float x = v[i];
    int r = 0;
    for (int j = 0; j < n; ++j) {
      const float y = v[j];
      r += (y < x || (y == x && j < i)) ? 1 : 0;
    }
    if (r == lo) pick[0] = x;
    if (r == hi) pick[1] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) out[o] = lerp_aten(pick[0], pick[1], rank - fl);
}

// pass 2: divide by the image's quantile, clamp, sRGB transfer curve (utils.py:35-41)
__global__ void __launch_bounds__(256) k_img_pass2(const ImgArgs a) {
  const long long per = (long long)3 * a.H * a.W;
  const long long n = per * a.B;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int b = (int)(i / per);
  float x = a.lin[i] / a.q[b];
  x = fminf(fmaxf(x, 0.f), 1.f);
  a.out[i] = (x <= 0.0031308f) ? 12.92f * x : 1.055f * powf(fabsf(x), 1.f / 2.4f) - 0.055f;
}

// MinMaxNormalise: smallest positive and largest finite value of the tensor (positive floats order like their bits)
__global__ void __launch_bounds__(256) k_img_minmax(const float* in, long long n, unsigned* mm) {
  unsigned lo = 0x7f800000u, hi = 0u;  // +inf / +0: identities of (min over positives, max over finite values)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float x = in[i];
    if (x > 0.f) lo = min(lo, __float_as_uint(x));
    if (x < INFINITY && x >= 0.f) hi = max(hi, __float_as_uint(x));  // (radiance is never negative: a finite maximum below
  }                                                                    //  zero is not representable in this encoding)
  for (int d = 32; d >= 1; d >>= 1) {
    lo = min(lo, (unsigned)__shfl_xor((int)lo, d, 64));
    hi = max(hi, (unsigned)__shfl_xor((int)hi, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&mm[0], lo);
    atomicMax(&mm[1], hi);
  }
}

__global__ void k_img_minmax_init(unsigned* mm) { mm[0] = 0x7f800000u; mm[1] = 0u; }

__global__ void __launch_bounds__(256) k_img_normalise(const float* in, long long n, const unsigned* mm, float m0, float range,
                                                       float* out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float lo = __uint_as_float(mm[0]), hi = __uint_as_float(mm[1]);
  const float xi = in[i];
  const float x = (xi != xi) ? xi : fminf(fmaxf(xi, lo), hi);  // torch.clip: min(max(x, lo), hi), NaN stays NaN (fmaxf would drop it)
  out[i] = __fsub_rn(__fdiv_rn(__fmul_rn(2.f, __fsub_rn(logf(x), m0)), range), 1.f);  // 2 * (log x - m0) / (m1 - m0) - 1
}

}  // namespace reni

using reni::reni_set_error;

extern "C" {

size_t reni_image_workspace_bytes(int64_t B, int64_t H, int64_t W) {
  if (B < 1 || H < 1 || W < 1) return 0;
  // linear image (when the caller does not keep it), channel quantiles, column quantiles, image quantiles, min/max words
  return (size_t)B * 3 * H * W * 4 + (size_t)B * H * W * 4 + (size_t)B * W * 4 + (size_t)B * 4 + 1024;
}

int reni_unnormalise_srgb(int64_t B, int64_t H, int64_t W, const float* img, const int64_t strides[4], int32_t unnormalise,
                          double minmax0, double minmax1, int32_t srgb, float* out_srgb, float* out_linear, void* ws,
                          size_t ws_bytes, void* stream) {
  if (B < 1 || H < 1 || W < 1 || B * H * W > 0x3fffffffLL) return reni_set_error(RENI_EINVAL, "image: bad B/H/W");
  if (!img || !strides) return reni_set_error(RENI_EINVAL, "image: NULL input");
  if (srgb && !out_srgb) return reni_set_error(RENI_EINVAL, "image: sRGB requested but out_srgb is NULL");
  if (!srgb && !out_linear) return reni_set_error(RENI_EINVAL, "image: nothing to compute");
  if (srgb && (H > reni::QMAX || W > reni::QMAX)) return reni_set_error(RENI_EUNSUPPORTED, "image: quantile axis longer than 4096");
  if (srgb && (!ws || ws_bytes < reni_image_workspace_bytes(B, H, W) || ((uintptr_t)ws & 255)))
    return reni_set_error(RENI_EWORKSPACE, "image: workspace missing, too small or not 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  reni::ImgArgs a;
  a.in = img; a.sb = strides[0]; a.sc = strides[1]; a.sh = strides[2]; a.sw = strides[3];
  a.B = (int)B; a.H = (int)H; a.W = (int)W;
  a.unnorm = unnormalise ? 1 : 0;
  a.range = (float)(minmax1 - minmax0); a.m0 = (float)minmax0;
  a.qf = 0.98f;
  size_t o = 0;
  float* lin_ws = nullptr;
  if (srgb) { lin_ws = (float*)(w + o); o += ((size_t)B * 3 * H * W * 4 + 255) & ~(size_t)255; }
  a.lin = out_linear ? out_linear : lin_ws;
  a.q1 = nullptr; a.out = out_srgb; a.q = nullptr;
  float *q2 = nullptr, *q3 = nullptr;
  if (srgb) {
    a.q1 = (float*)(w + o); o += ((size_t)B * H * W * 4 + 255) & ~(size_t)255;
    q2 = (float*)(w + o); o += ((size_t)B * W * 4 + 255) & ~(size_t)255;
    q3 = (float*)(w + o);
  }
  const long long npix = (long long)B * H * W;
  hipLaunchKernelGGL(reni::k_img_pass1, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, a);
  if (srgb) {
    // over H (dim 1 of [B,H,W]) -> [B,W]; then over W (dim 1 of [B,W]) -> [B]
    hipLaunchKernelGGL(reni::k_img_quantile, dim3((unsigned)(B * W)), dim3(256), 0, s, (const float*)a.q1, (int)H, (long long)W,
                       (int)W, (long long)H * W, 1LL, a.qf, q2);
    hipLaunchKernelGGL(reni::k_img_quantile, dim3((unsigned)B), dim3(256), 0, s, (const float*)q2, (int)W, 1LL, 1, (long long)W, 0LL,
                       a.qf, q3);
    a.q = q3;
    const long long n = npix * 3;
    hipLaunchKernelGGL(reni::k_img_pass2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return reni_set_error(RENI_EHIP, hipGetErrorString(e));
  return RENI_OK;
}

int reni_minmax_normalise(int64_t n, const float* img, double minmax0, double minmax1, float* out, void* ws, size_t ws_bytes,
                          void* stream) {
  if (n < 1 || n > 0x3fffffffffLL) return reni_set_error(RENI_EINVAL, "normalise: bad element count");
  if (!img || !out) return reni_set_error(RENI_EINVAL, "normalise: NULL argument");
  if (!ws || ws_bytes < 256 || ((uintptr_t)ws & 255)) return reni_set_error(RENI_EWORKSPACE, "normalise: workspace of 256 aligned bytes needed");
  if (!(minmax1 > minmax0)) return reni_set_error(RENI_EINVAL, "normalise: minmax[1] must exceed minmax[0]");
  hipStream_t s = (hipStream_t)stream;
  unsigned* mm = (unsigned*)ws;
  hipLaunchKernelGGL(reni::k_img_minmax_init, dim3(1), dim3(1), 0, s, mm);
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(reni::k_img_minmax, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, s, img, (long long)n, mm);
  hipLaunchKernelGGL(reni::k_img_normalise, dim3((unsigned)nb), dim3(256), 0, s, img, (long long)n, (const unsigned*)mm, (float)minmax0,
                     (float)(minmax1 - minmax0), out);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return reni_set_error(RENI_EHIP, hipGetErrorString(e));
  return RENI_OK;
}

}  // extern "C"
