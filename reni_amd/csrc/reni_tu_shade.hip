// translation unit of libreni_hip.so: environment-map Blinn-Phong shading of a G-buffer (FIT_INVERSE task).
//
// Reference: blinn_phong_shading_env_map, src/utils/pytorch3d_envmap_shader.py:46-116 (the part after the two
// interpolate_face_attributes calls; the rasteriser itself is pytorch3d's and out of scope).  With N_p the pixel
// normal, V_p the unit view vector, L_j the direction and C_bj the (sine-weighted) colour of environment-map texel j:
//
//   M(p, j)       = kd * clamp(N_p . L_j, 0, 1) + ks * norm(s) * clamp(N_p . normalize(V_p + L_j), 0, 1) ^ s
//   colors[b,p,:] = sum_j M(p, j) C[b,j,:]                                    (:89-114)
//   dC[b,j,:]     = sum_p M(p, j) dcolors[b,p,:]                              (autograd of the two einsums, :99,:111)
//
// The reference materialises diffuse / half-way / specular tensors of B x H x W x J (x 3) elements; here M is a
// function evaluated in registers.  Forward and backward are the same kernel with the roles of the two axes swapped:
// a thread OWNS one element of one axis (its geometry and 3 * BC accumulators in registers) and walks the other axis,
// whose geometry and source rows are staged through LDS and read by broadcast.  M does not depend on the image when
// the texel directions are shared (the reference repeats one grid, RENI_module.py:376,112), so one evaluation of M
// serves BC images.  The other axis is split over blockIdx.y to fill the chip; partial sums are combined by a second
// kernel in a fixed order (deterministic, no atomics).  VALU-bound: ~25 fp32 operations and three transcendentals per
// (p, j) pair against 6 * BC flops of accumulation -- this is not GEMM-shaped work for the matrix cores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "reni_hip.h"
#include "reni_internal.h"

#define DEV __device__ __forceinline__

namespace reni {

struct ShadeArgs {
  const float* nrm;        // [NP][3] interpolated vertex normals (not normalised; 0 where no face covers the pixel)
  const float* pos;        // [NP][3] interpolated surface positions
  const float* ldir;       // [J][3] texel directions of the image chunk's first image
  const float* src;        // forward: C [B][J][3]; backward: dcolors [B][NP][3]
  float* part;             // [S][B][NOWN][3] partial sums (S == 1: the output itself)
  float cam[3];
  float shin, kd, ksn;     // ksn = ks * (s + 2) / (4 (2 - exp(-s / 2)))     (:112-114)
  int B, NP, J, b0, nb;    // images b0 .. b0 + nb - 1 are handled by this launch (nb <= BC)
  int per_split;           // elements of the other axis per blockIdx.y
};

DEV void normalize3(float (&v)[3]) {  // F.normalize(p=2, eps=1e-6): v / max(|v|, eps)      (:82,:93,:107)
  const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float inv = 1.f / fmaxf(n, 1e-6f);
  v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

DEV float shade_coeff(const float (&N)[3], const float (&V)[3], const float (&L)[3], float shin, float kd, float ksn) {
  const float nl = fminf(fmaxf(N[0] * L[0] + N[1] * L[1] + N[2] * L[2], 0.f), 1.f);        // :87-88
  const float hx = V[0] + L[0], hy = V[1] + L[1], hz = V[2] + L[2];                          // :105
  const float inv = __builtin_amdgcn_rsqf(fmaxf(hx * hx + hy * hy + hz * hz, 1e-12f));       // 1 / max(|h|, 1e-6)  :107
  // normalise first, then dot, as the reference does
  const float nh = fminf(fmaxf(N[0] * (hx * inv) + N[1] * (hy * inv) + N[2] * (hz * inv), 0.f), 1.f);  // :108-109
  const float spec = __builtin_amdgcn_exp2f(shin * __builtin_amdgcn_logf(nh));               // pow(x, s), x in [0, 1]; 0 -> 0
  return kd * nl + ksn * spec;
}

constexpr int SH_TILE = 128;  // elements of the other axis per LDS stage

// OWN_PIXEL: a thread owns pixel p and sums over texels j (forward); else it owns texel j and sums over pixels
template <bool OWN_PIXEL, int BC>
__global__ void __launch_bounds__(256) k_envmap_shade(const ShadeArgs a) {
  __shared__ float4 g0[SH_TILE];       // other axis geometry: texel (Lx, Ly, Lz, -) | pixel (Nx, Ny, Nz, Vx)
  __shared__ float4 g1[SH_TILE];       //                                            | pixel (Vy, Vz, -, -)
  __shared__ float4 sv[SH_TILE][(3 * BC + 3) / 4];  // source rows of the BC images
  const int tid = threadIdx.x;
  const int n_own = OWN_PIXEL ? a.NP : a.J, n_oth = OWN_PIXEL ? a.J : a.NP;
  const int o = blockIdx.x * 256 + tid;
  const bool live = o < n_own;
  float N[3] = {0.f, 0.f, 0.f}, V[3] = {0.f, 0.f, 0.f}, L[3] = {0.f, 0.f, 0.f};
  auto load_pixel = [&](int p, float (&n)[3], float (&v)[3]) {
    n[0] = a.nrm[(size_t)p * 3]; n[1] = a.nrm[(size_t)p * 3 + 1]; n[2] = a.nrm[(size_t)p * 3 + 2];
    normalize3(n);
    v[0] = a.cam[0] - a.pos[(size_t)p * 3]; v[1] = a.cam[1] - a.pos[(size_t)p * 3 + 1]; v[2] = a.cam[2] - a.pos[(size_t)p * 3 + 2];  // :91
    normalize3(v);
  };
  if (live) {
    if (OWN_PIXEL) load_pixel(o, N, V);
    else { L[0] = a.ldir[(size_t)o * 3]; L[1] = a.ldir[(size_t)o * 3 + 1]; L[2] = a.ldir[(size_t)o * 3 + 2]; }
  }
  float acc[BC][3];
#pragma unroll
  for (int b = 0; b < BC; ++b) { acc[b][0] = 0.f; acc[b][1] = 0.f; acc[b][2] = 0.f; }

  const int t_begin = blockIdx.y * a.per_split, t_end = min(n_oth, t_begin + a.per_split);
  for (int t0 = t_begin; t0 < t_end; t0 += SH_TILE) {
    __syncthreads();
    if (tid < SH_TILE) {
      const int t = t0 + tid;
      float4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
      if (t < t_end) {
        if (OWN_PIXEL) {
          q0.x = a.ldir[(size_t)t * 3]; q0.y = a.ldir[(size_t)t * 3 + 1]; q0.z = a.ldir[(size_t)t * 3 + 2];
        } else {
          float n[3], v[3];
          load_pixel(t, n, v);
          q0 = float4{n[0], n[1], n[2], v[0]};
          q1 = float4{v[1], v[2], 0.f, 0.f};
        }
      }
      g0[tid] = q0;
      if (!OWN_PIXEL) g1[tid] = q1;
    }
    {  // source rows: 3 * BC floats per element, zero beyond the range / the chunk's images (they then add nothing)
      float* svf = (float*)sv;
      constexpr int ROW = 4 * ((3 * BC + 3) / 4);
      for (int i = tid; i < SH_TILE * 3 * BC; i += 256) {
        const int e = i / (3 * BC), r = i - e * (3 * BC), b = r / 3, c = r - b * 3;
        const int t = t0 + e;
        float x = 0.f;
        if (t < t_end && b < a.nb) x = a.src[((size_t)(a.b0 + b) * n_oth + t) * 3 + c];
        svf[e * ROW + r] = x;
      }
    }
    __syncthreads();
    const int cnt = min(SH_TILE, t_end - t0);
    for (int e = 0; e < cnt; ++e) {
      const float4 q0 = g0[e];
      float m;
      if (OWN_PIXEL) {
        const float Lo[3] = {q0.x, q0.y, q0.z};
        m = shade_coeff(N, V, Lo, a.shin, a.kd, a.ksn);
      } else {
        const float4 q1 = g1[e];
        const float No[3] = {q0.x, q0.y, q0.z}, Vo[3] = {q0.w, q1.x, q1.y};
        m = shade_coeff(No, Vo, L, a.shin, a.kd, a.ksn);
      }
      const float* s = (const float*)sv[e];
#pragma unroll
      for (int b = 0; b < BC; ++b) {
        acc[b][0] += m * s[3 * b]; acc[b][1] += m * s[3 * b + 1]; acc[b][2] += m * s[3 * b + 2];
      }
    }
  }
  if (live) {
#pragma unroll
    for (int b = 0; b < BC; ++b) {
      if (b < a.nb) {
        float* dst = a.part + (((size_t)blockIdx.y * a.B + (a.b0 + b)) * n_own + o) * 3;
        dst[0] = acc[b][0]; dst[1] = acc[b][1]; dst[2] = acc[b][2];
      }
    }
  }
}

// out[i] = sum_s part[s][i], s ascending
__global__ void __launch_bounds__(256) k_shade_reduce(const float* part, size_t n, int S, float* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += part[(size_t)k * n + i];
  out[i] = s;
}

}  // namespace reni

namespace {

using reni::reni_set_error;
constexpr int SHADE_BC = 4;

int n_splits(int64_t n_own, int64_t n_oth) {  // enough workgroups for 256 CUs, whole LDS tiles per split
  const int64_t wg_x = (n_own + 255) / 256;
  int64_t s = (2048 + wg_x - 1) / wg_x;
  const int64_t max_s = (n_oth + reni::SH_TILE - 1) / reni::SH_TILE;
  if (s > max_s) s = max_s;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return (int)s;
}

int shade_common(bool forward, int64_t B, int64_t NP, int64_t J, const float* normals, const float* positions,
                 float cx, float cy, float cz, const float* light_dirs, int64_t dirs_bstride, const float* src,
                 float shininess, float kd, float ks, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (B < 1 || NP < 1 || J < 1) return reni_set_error(RENI_EINVAL, "envmap shade: B, NP and J must be >= 1");
  if (NP > 0x3fffffff || J > 0x3fffffff || B > 0xffff) return reni_set_error(RENI_EINVAL, "envmap shade: problem too large");
  if (!normals || !positions || !light_dirs || !src || !out) return reni_set_error(RENI_EINVAL, "envmap shade: NULL argument");
  if (dirs_bstride != 0 && dirs_bstride < J * 3) return reni_set_error(RENI_EINVAL, "envmap shade: direction batch stride must be 0 or >= 3 J");
  if (!(shininess > 0.f)) return reni_set_error(RENI_EINVAL, "envmap shade: shininess must be positive");
  const int64_t n_own = forward ? NP : J, n_oth = forward ? J : NP;
  const int S = n_splits(n_own, n_oth);
  const size_t need = S > 1 ? (size_t)S * B * n_own * 3 * sizeof(float) : 0;
  if (need > 0 && (!ws || ws_bytes < need)) return reni_set_error(RENI_EWORKSPACE, "envmap shade: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  reni::ShadeArgs a;
  a.nrm = normals; a.pos = positions; a.src = src;
  a.part = S > 1 ? (float*)ws : out;
  a.cam[0] = cx; a.cam[1] = cy; a.cam[2] = cz;
  a.shin = shininess; a.kd = kd;
  a.ksn = ks * (shininess + 2.f) / (4.f * (2.f - expf(-shininess * 0.5f)));
  a.B = (int)B; a.NP = (int)NP; a.J = (int)J;
  const int64_t tiles = (n_oth + reni::SH_TILE - 1) / reni::SH_TILE;
  a.per_split = (int)((tiles + S - 1) / S) * reni::SH_TILE;
  const dim3 grid((unsigned)((n_own + 255) / 256), (unsigned)S);
  // images that share their texel directions share M: chunks of SHADE_BC; otherwise one image per launch
  const int step = dirs_bstride == 0 ? SHADE_BC : 1;
  for (int64_t b0 = 0; b0 < B; b0 += step) {
    a.b0 = (int)b0; a.nb = (int)((B - b0) < step ? (B - b0) : step);
    a.ldir = light_dirs + (size_t)b0 * dirs_bstride;
    if (step == 1) {
      if (forward) hipLaunchKernelGGL((reni::k_envmap_shade<true, 1>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((reni::k_envmap_shade<false, 1>), grid, dim3(256), 0, s, a);
    } else {
      if (forward) hipLaunchKernelGGL((reni::k_envmap_shade<true, SHADE_BC>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((reni::k_envmap_shade<false, SHADE_BC>), grid, dim3(256), 0, s, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return reni_set_error(RENI_EHIP, hipGetErrorString(e));
  }
  if (S > 1) {
    const size_t n = (size_t)B * n_own * 3;
    hipLaunchKernelGGL(reni::k_shade_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)ws, n, S, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return reni_set_error(RENI_EHIP, hipGetErrorString(e));
  }
  return RENI_OK;
}

}  // namespace

extern "C" {

size_t reni_envmap_shade_workspace_bytes(int64_t B, int64_t NP, int64_t J) {
  if (B < 1 || NP < 1 || J < 1) return 0;
  const size_t f = (size_t)n_splits(NP, J) * B * NP * 3 * sizeof(float);
  const size_t g = (size_t)n_splits(J, NP) * B * J * 3 * sizeof(float);
  return (f > g ? f : g) + 256;
}

int reni_envmap_shade(int64_t B, int64_t NP, int64_t J, const float* normals, const float* positions, float cam_x,
                      float cam_y, float cam_z, const float* light_dirs, int64_t dirs_batch_stride,
                      const float* light_colors, float shininess, float kd, float ks, float* colors, void* ws,
                      size_t ws_bytes, void* stream) {
  return shade_common(true, B, NP, J, normals, positions, cam_x, cam_y, cam_z, light_dirs, dirs_batch_stride, light_colors,
                      shininess, kd, ks, colors, ws, ws_bytes, stream);
}

int reni_envmap_shade_backward(int64_t B, int64_t NP, int64_t J, const float* normals, const float* positions, float cam_x,
                               float cam_y, float cam_z, const float* light_dirs, int64_t dirs_batch_stride,
                               const float* dcolors, float shininess, float kd, float ks, float* dlight_colors, void* ws,
                               size_t ws_bytes, void* stream) {
  return shade_common(false, B, NP, J, normals, positions, cam_x, cam_y, cam_z, light_dirs, dirs_batch_stride, dcolors,
                      shininess, kd, ks, dlight_colors, ws, ws_bytes, stream);
}

}  // extern "C"
