// reni_kernels.hip -- hand-written gfx950 (CDNA4) kernels for the RENI forward / training hot path.
//
// What is computed (reference: /root/reference/src/models/RENI.py:23-60 invariant encodings,
// :63-87 SineLayer, :132-178 decoder; src/utils/loss_functions.py:6-71 losses) and how it is
// re-factored for the GPU is described in DESIGN.md.  Short version:
//
//   * Every column of the reference's concatenated MLP input is either constant per image
//     (Gram entries, Z_y, vec(Z)) or linear in the direction d (inner products d.Z^T, |d_xz|,
//     d_y).  The first SineLayer therefore collapses to a per-image affine map
//         a0 = A_b . (dx, dy, dz, r, 1),   A_b in R^{H x 5}   (k_prep_image)
//     and its backward to a per-image H x 5 reduction dA_b (fused kernel) followed by tiny
//     per-image tails (k_tail_dz, k_tail_dw0).
//   * The dense part runs on MFMA with activations laid out [feature][sample]: the weights are
//     the A operand (read from LDS), the activations the B operand, so the accumulator of layer l
//     (lane = sample, registers = features) IS the B operand of layer l+1 after sin() -- the
//     chain never leaves registers.
//   * One 256-thread workgroup = 4 waves x 32 samples = a 128-sample tile.  Forward stashes the
//     sine arguments of every layer (u16 phase for bf16, fp32 pre-activation for fp32) to a
//     per-workgroup scratch ring that stays cache resident; backward replays them in reverse.
//   * Weight gradients are MFMA GEMMs with K = samples: gradients/activations are transposed
//     through LDS once per layer and tile; per-workgroup partial sums are reduced by
//     k_reduce_partials in a fixed order (deterministic, no float atomics).
//
// gfx950 only.  No portability layer.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "reni_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define DEV __device__ __forceinline__

namespace reni {

// Row (within a 32-row block) held by accumulator register r of a 32x32 MFMA result in a lane
// of the upper (hi=1) or lower (hi=0) half-wave.  (cdna_hip_programming.md section 3.)
DEV constexpr int rowmap(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ------------------------------------------------------------------------------------------
// precision policies
// ------------------------------------------------------------------------------------------
struct PolF32 {
  static constexpr bool BF = false;
  static constexpr int FRAGB = 4;        // bytes per lane of one A/B fragment
  static constexpr int KF = 2;           // features contracted per MFMA (k-step)
  static constexpr int TS = 2;           // samples contracted per MFMA in the dW GEMMs
  static constexpr int WPS = 1;          // waves per SIMD to compile for
  static constexpr int HEAD_BWD_KS = 3;  // k-steps that cover the 3 real head outputs
  static constexpr int STASH_CH_PER_RB = 4;  // 16-byte chunks per lane per 32-feature block
  using Frag = float;
  static DEV f32x16 mfma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  // Wait out the write-back of the last MFMA of a chain before anything else may touch its result.
  // hipcc's own hazard padding cannot be relied on across branches (ROCm 7.2: it counts a taken
  // s_cbranch as a wait state and, with an MFMA at the end of a block, was seen to place the pad AFTER
  // the first read of the result: H = 256 bf16 head dW, nondeterministic rows).  The generic kernels
  // therefore end every chain with their own pad; tests/test_build_audit.py checks the emitted ISA.
  static DEV void drain() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 3");  // 16-pass XDL: 18 states (+2)
    __builtin_amdgcn_sched_barrier(0);
  }
};
struct PolBF16 {
  static constexpr bool BF = true;
  static constexpr int FRAGB = 16;
  static constexpr int KF = 16;
  static constexpr int TS = 16;
  static constexpr int WPS = 2;
  static constexpr int HEAD_BWD_KS = 1;
  static constexpr int STASH_CH_PER_RB = 2;
  using Frag = bf16x8;
  static DEV f32x16 mfma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static DEV void drain() {  // see PolF32::drain
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 11");  // 8-pass XDL: 12 states
    __builtin_amdgcn_sched_barrier(0);
  }
};

// feature index (within the layer input) that k-slot (ks, hi, e) of a fragment stands for.
// It is the order in which a D-layout register file enumerates features, so that accumulators
// can be fed back as B operands without any data movement.
template <class Pol>
__host__ __device__ constexpr int kfeat(int ks, int hi, int e) {
  if (Pol::BF) return 16 * ks + (e & 3) + 8 * (e >> 2) + 4 * hi;
  return 32 * (ks >> 4) + ((ks & 15) & 3) + 8 * ((ks & 15) >> 2) + 4 * hi;
}

// LDS geometry ------------------------------------------------------------------------------
template <class Pol, int H>
struct Geo {
  static constexpr int NRB = H / 32;
  static constexpr int NKS = H / Pol::KF;                       // k-steps of an H-wide contraction
  static constexpr int IMG_HID = NRB * NKS * 64 * Pol::FRAGB;   // hidden layer image (fwd or bwd)
  static constexpr int IMG_HF = 1 * NKS * 64 * Pol::FRAGB;      // head forward image
  static constexpr int IMG_HB = NRB * Pol::HEAD_BWD_KS * 64 * Pol::FRAGB;  // head backward image
  static constexpr int BIAS_HID = H * 4;
  static constexpr int BIAS_HEAD = 32 * 4;
  // a hidden image is staged into LDS CH_RB row blocks at a time (the whole layer up to H = 128)
  static constexpr int CH_RB = (H > 128) ? 2 : NRB;
  static constexpr int NCHUNK = NRB / CH_RB;
  static constexpr int CHUNK_BYTES = CH_RB * NKS * 64 * Pol::FRAGB;
  // waves whose samples share one dW transposition round
  static constexpr int ROUND_WAVES = Pol::BF ? (H > 128 ? 2 : 4) : 1;
  static constexpr int ROUND_SAMPLES = 32 * ROUND_WAVES;
  // transposition images: bf16 [feature][sample] rows padded by 16 B; fp32 [sample][feature+1]
  static constexpr int T_ROWB = ROUND_SAMPLES * 2 + 16;
  static constexpr int T_BYTES = Pol::BF ? (H * T_ROWB) : (ROUND_SAMPLES * (H + 1) * 4);
  static constexpr int T_BYTES_AL = (T_BYTES + 255) & ~255;
  static constexpr int WB_A = CHUNK_BYTES > T_BYTES_AL ? CHUNK_BYTES : T_BYTES_AL;
  static constexpr int WB_B = IMG_HF > IMG_HB ? IMG_HF : IMG_HB;
  static constexpr int WB_RAW = WB_A > WB_B ? WB_A : WB_B;
  static constexpr int WB_BYTES = (WB_RAW + 255) & ~255;
  static constexpr int BIAS_LDS = (BIAS_HID + 255) & ~255;      // the layer's bias lives after the images
  static constexpr int LDS_BYTES = WB_BYTES + T_BYTES_AL + BIAS_LDS;
  static constexpr int STASH_CH = NRB * Pol::STASH_CH_PER_RB;   // 16-B chunks per lane per layer
  static constexpr int STASH_LAYER_BYTES = STASH_CH * 256 * 16;
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
DEV float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}


// sin and cos of an fp32 argument, ~1 ulp for |x| < 1e4: three-term Cody-Waite reduction by pi/2
// with FMAs, then degree-9 / degree-8 minimax polynomials on [-pi/4, pi/4].  Used by the fp32
// (parity-grade) kernels instead of the library sinf/cosf, whose huge-argument slow path costs
// scratch memory and a loop.
DEV void sincos_f32(float x, float& s, float& c) {
  const float kf = rintf(x * 0.63661977236758134f);
  float r = __builtin_fmaf(-kf, 1.57079637050628662109375f, x);
  r = __builtin_fmaf(-kf, -4.37113900018624283e-8f, r);
  r = __builtin_fmaf(-kf, -1.71512449e-15f, r);
  const float r2 = r * r;
  float ps = __builtin_fmaf(r2, 2.7557314297e-06f, -1.9841270114e-04f);
  ps = __builtin_fmaf(ps, r2, 8.3333337680e-03f);
  ps = __builtin_fmaf(ps, r2, -1.6666667163e-01f);
  const float sr = __builtin_fmaf(ps * r2, r, r);
  float pc = __builtin_fmaf(r2, -2.7557314297e-07f, 2.4801587642e-05f);
  pc = __builtin_fmaf(pc, r2, -1.3888889225e-03f);
  pc = __builtin_fmaf(pc, r2, 4.1666667908e-02f);
  const float cr = __builtin_fmaf(pc * r2, r2, __builtin_fmaf(-0.5f, r2, 1.0f));
  const int q = (int)kf;
  const float s0 = (q & 1) ? cr : sr;
  const float c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}
DEV float sin_f32(float x) { float s, c; sincos_f32(x, s, c); return s; }
DEV float cos_f32(float x) { float s, c; sincos_f32(x, s, c); return c; }

DEV void stage_to_lds(char* dst, const char* src, int bytes, int tid) {
  for (int o = tid * 16; o < bytes; o += 256 * 16) *(u32x4*)(dst + o) = *(const u32x4*)(src + o);
}

template <class Pol, int NRB_OUT>
DEV void acc_init_bias(f32x16 (&acc)[NRB_OUT], const float* bias_lds, int hi) {
#pragma unroll
  for (int rb = 0; rb < NRB_OUT; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = bias_lds[32 * rb + rowmap(r, hi)];
}

template <int NRB_OUT>
DEV void acc_zero(f32x16 (&acc)[NRB_OUT]) {
#pragma unroll
  for (int rb = 0; rb < NRB_OUT; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
}

// D[32*rbo.., sample] += sum_k Wimg[rbo][k] * vin[k]  for the wave's 32 samples.
// vin is a D-layout register file (lane = sample, register = feature).
template <class Pol, int NRB_OUT, int NRB_IN, int NKS_USE>
DEV void gemm_lds(const char* wb, const float (&vin)[NRB_IN][16], f32x16 (&acc)[NRB_OUT], int lane) {
  if constexpr (Pol::BF) {
    bf16x8 bop[NKS_USE];
#pragma unroll
    for (int ks = 0; ks < NKS_USE; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) bop[ks][e] = (__bf16)vin[ks >> 1][(ks & 1) * 8 + e];
#pragma unroll
    for (int ks = 0; ks < NKS_USE; ++ks)
#pragma unroll
      for (int rbo = 0; rbo < NRB_OUT; ++rbo) {
        bf16x8 a = *(const bf16x8*)(wb + ((rbo * NKS_USE + ks) * 64 + lane) * 16);
        acc[rbo] = PolBF16::mfma(a, bop[ks], acc[rbo]);
      }
  } else {
#pragma unroll
    for (int ks = 0; ks < NKS_USE; ++ks)
#pragma unroll
      for (int rbo = 0; rbo < NRB_OUT; ++rbo) {
        float a = *(const float*)(wb + ((rbo * NKS_USE + ks) * 64 + lane) * 4);
        acc[rbo] = PolF32::mfma(a, vin[ks >> 4][ks & 15], acc[rbo]);
      }
  }
  Pol::drain();
}

// v: pre-activation a (in)  ->  h = sin(omega a) (out); the sine argument is stashed for backward.
template <class Pol, int NRB, bool STASH>
DEV void act_forward(float (&v)[NRB][16], float omega, char* stash_layer, int tid) {
  if constexpr (Pol::BF) {
    const float sc = omega * 0.15915494309189535f;  // argument in revolutions
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      unsigned q[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float th = __builtin_amdgcn_fractf(v[rb][r] * sc);
        q[r] = ((unsigned)(th * 65536.f + 0.5f)) & 0xffffu;
        v[rb][r] = __builtin_amdgcn_sinf(th);
      }
      if constexpr (STASH) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          u32x4 w;
#pragma unroll
          for (int d = 0; d < 4; ++d) w[d] = q[8 * c + 2 * d] | (q[8 * c + 2 * d + 1] << 16);
          *(u32x4*)(stash_layer + ((rb * 2 + c) * 256 + tid) * 16) = w;
        }
      }
    }
  } else {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      if constexpr (STASH) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          f32x4 w = {v[rb][4 * c], v[rb][4 * c + 1], v[rb][4 * c + 2], v[rb][4 * c + 3]};
          *(f32x4*)(stash_layer + ((rb * 4 + c) * 256 + tid) * 16) = w;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) v[rb][r] = sin_f32(omega * v[rb][r]);
    }
  }
}

// Reload a layer's stash; WHAT = 0: cos(omega a) * scale * g (in place on g), WHAT = 1: sin(omega a) -> out
template <class Pol, int NRB, int WHAT>
DEV void act_replay(float (&x)[NRB][16], const f32x16* g, float omega, float scale,
                    const char* stash_layer, int tid) {
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
    if constexpr (Pol::BF) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        u32x4 w = *(const u32x4*)(stash_layer + ((rb * 2 + c) * 256 + tid) * 16);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          float t0 = (float)(w[d] & 0xffffu) * (1.f / 65536.f);
          float t1 = (float)(w[d] >> 16) * (1.f / 65536.f);
          const int r = 8 * c + 2 * d;
          if constexpr (WHAT == 0) {
            x[rb][r] = g[rb][r] * scale * __builtin_amdgcn_cosf(t0);
            x[rb][r + 1] = g[rb][r + 1] * scale * __builtin_amdgcn_cosf(t1);
          } else {
            x[rb][r] = __builtin_amdgcn_sinf(t0);
            x[rb][r + 1] = __builtin_amdgcn_sinf(t1);
          }
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 w = *(const f32x4*)(stash_layer + ((rb * 4 + c) * 256 + tid) * 16);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int r = 4 * c + d;
          if constexpr (WHAT == 0) x[rb][r] = g[rb][r] * scale * cos_f32(omega * w[d]);
          else x[rb][r] = sin_f32(omega * w[d]);
        }
      }
    }
  }
}

// ---- transposition through LDS for the K = samples GEMMs ------------------------------------
template <class Pol, int H, int NB>
DEV void t_write(char* T, const float (&x)[NB][16], int wslot, int hi, int j) {
  using G = Geo<Pol, H>;
#pragma unroll
  for (int rb = 0; rb < NB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * rb + rowmap(r, hi);
      if constexpr (Pol::BF) {
        *(__bf16*)(T + f * G::T_ROWB + (wslot * 32 + j) * 2) = (__bf16)x[rb][r];
      } else {
        *(float*)(T + (j * (32 * NB + 1) + f) * 4) = x[rb][r];
      }
    }
}

template <class Pol, int H, int NB>
DEV typename Pol::Frag t_read(const char* T, int blk, int ks, int i, int khi) {
  using G = Geo<Pol, H>;
  if constexpr (Pol::BF) {
    return *(const bf16x8*)(T + (32 * blk + i) * G::T_ROWB + (16 * ks + 8 * khi) * 2);
  } else {
    return *(const float*)(T + ((2 * ks + khi) * (32 * NB + 1) + 32 * blk + i) * 4);
  }
}

template <class Pol>
DEV float frag_sum(typename Pol::Frag a) {
  if constexpr (Pol::BF) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)a[e];
    return s;
  } else {
    return a;
  }
}

enum { DW_HIDDEN = 0, DW_HEAD = 1, DW_L0 = 2 };

// MFMA with the accumulator PINNED to the AGPR half ("a") or the VGPR half ("v") of the register
// file.  hipcc puts every MFMA accumulator of a kernel into AGPRs (256 = sixteen 32x32 tiles); the
// persistent training kernel needs twenty-odd live tiles, so it pins them by hand.  Operands come
// straight from ds_read (the compiler's s_waitcnt covers them) and the accumulator is only ever
// touched by these statements, so no MFMA hazard crosses the asm boundary (cdna guide, 5.7).
// ---- hand-owned AGPRs ------------------------------------------------------------------------------
// The persistent training kernel owns ALL 256 AGPRs as sixteen literal 32x32 fp32 tiles a[16t:16t+15]
// (t = 4*(layer-2) + column block): the dW accumulators of hidden layers 2..5.  The compiler never sees
// them as values -- every statement that touches them names the registers in its text and lists all AGPRs
// as clobbers (which also makes the kernel descriptor allocate them); the library is built with
// -amdgpu-spill-vgpr-to-agpr=0 and every MFMA of that kernel is an asm statement, so no compiler-generated
// instruction ever reads or writes an AGPR (checked on the emitted ISA by tests/test_build_audit.py).
//
// hipcc neither pads hazards inside an asm statement nor knows that the statement is an MFMA, so the wait
// states are part of the string: `s_nop 4` in front covers a VALU write of an operand just before; LAST =
// the final MFMA of a chain is followed by 20 wait states so that a read of the result never overtakes the
// 8-pass XDL write-back.
#define RENI_A10(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define RENI_AGPR_ALL                                                                                       \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", RENI_A10(1), RENI_A10(2), RENI_A10(3),         \
      RENI_A10(4), RENI_A10(5), RENI_A10(6), RENI_A10(7), RENI_A10(8), RENI_A10(9), RENI_A10(10),            \
      RENI_A10(11), RENI_A10(12), RENI_A10(13), RENI_A10(14), RENI_A10(15), RENI_A10(16), RENI_A10(17),      \
      RENI_A10(18), RENI_A10(19), RENI_A10(20), RENI_A10(21), RENI_A10(22), RENI_A10(23), RENI_A10(24),      \
      "a250", "a251", "a252", "a253", "a254", "a255"

template <int T, bool LAST>
DEV void mfma_bf16_agpr_tile(bf16x8 a, bf16x8 b) {
  if constexpr (LAST)
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]\n\ts_nop 15\n\ts_nop 3"
                 :: "v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15) : RENI_AGPR_ALL);
  else
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]"
                 :: "v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15) : RENI_AGPR_ALL);
}
template <int N>
DEV void agpr_zero_one() { asm volatile("v_accvgpr_write_b32 a[%c0], 0" :: "i"(N) : RENI_AGPR_ALL); }
template <int... I>
DEV void agpr_zero_seq(std::integer_sequence<int, I...>) { (agpr_zero_one<I>(), ...); }
template <int N>
DEV float agpr_read_one() { float x; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(N)); return x; }
template <int T, int... R>
DEV f32x16 agpr_read_tile_seq(std::integer_sequence<int, R...>) { f32x16 v = {agpr_read_one<16 * T + R>()...}; return v; }
template <int T>
DEV f32x16 agpr_read_tile() { return agpr_read_tile_seq<T>(std::make_integer_sequence<int, 16>{}); }

template <bool LAST>
DEV void mfma_bf16_pin_v(f32x16& acc, bf16x8 a, bf16x8 b) {
  if constexpr (LAST)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(acc) : "v"(a), "v"(b));
  else
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// K = samples GEMM from the transposed LDS images: acc[m] += rows-block x cols-block over one round
// PIN: 0 = compiler-allocated accumulator, 1 = pinned to AGPRs, 2 = pinned to VGPRs (bf16 only)
template <class Pol, int H, int NBR, int NBC, int KIND, int MYB, int PIN = 0, int M0 = 0>
DEV void dw_gemm(const char* TA, const char* TB, f32x16 (&acc)[MYB], float& dbacc, int wave, int lane,
                 float* dbacc2 = nullptr) {  // dbacc2: bias sum of the wave's second row block (NBR == 8)
  using G = Geo<Pol, H>;
  using Frag = typename Pol::Frag;
  constexpr int NBLK = NBR * NBC;
  constexpr int NKS_T = G::ROUND_SAMPLES / Pol::TS;
  const int hi = lane >> 5, j = lane & 31;
  if constexpr (PIN == 0) {
#pragma unroll
    for (int ks = 0; ks < NKS_T; ++ks) {
#pragma unroll
      for (int m = 0; m < MYB; ++m) {
        const int blk = wave + 4 * (M0 + m);
        if (blk < NBLK) {
          const int rbo = blk % NBR, cb = blk / NBR;
          Frag fa = t_read<Pol, H, NBR>(TA, rbo, ks, j, hi);
          Frag fb = t_read<Pol, H, NBC>(TB, cb, ks, j, hi);
          acc[m] = Pol::mfma(fa, fb, acc[m]);
          // bias sums: the owner test is a 0/1 factor, never a branch -- hipcc counts a taken s_cbranch
          // as one of the 12 wait states between an 8-pass MFMA and the first read of its result, and a
          // wave that skipped the sum read stale accumulator rows (H = 256 head dW, waves 1..3)
          if (KIND == DW_HIDDEN && M0 + m == 0) {  // (m is an unrolled constant)
            if (NBR >= 4) dbacc += frag_sum<Pol>(fa);
            else dbacc += (cb == 0 ? 1.f : 0.f) * frag_sum<Pol>(fa);
          }
          if (KIND == DW_HIDDEN && NBR > 4 && M0 + m == 1 && dbacc2 != nullptr) *dbacc2 += frag_sum<Pol>(fa);
          if (KIND == DW_HEAD && M0 + m == 0) dbacc += (wave == 0 ? 1.f : 0.f) * frag_sum<Pol>(fb);
        }
      }
    }
    Pol::drain();
  } else {
    // pinned accumulators (persistent training kernel: NBR == 4, every wave owns row block `wave`
    // and all MYB column blocks).  Fragments are fetched one k-step ahead and the schedule is fenced
    // per k-step so that no more than two k-steps of fragments are ever live.
    static_assert(NBLK == 4 * MYB && M0 == 0, "pinned form: every wave owns exactly MYB blocks");
#pragma unroll
    for (int ks = 0; ks < NKS_T; ++ks) {
#pragma unroll
      for (int m = 0; m < MYB; ++m) {
        const int blk = wave + 4 * m;
        Frag fa = t_read<Pol, H, NBR>(TA, blk % NBR, ks, j, hi);
        Frag fb = t_read<Pol, H, NBC>(TB, blk / NBR, ks, j, hi);
        if (ks == NKS_T - 1 && m == MYB - 1) mfma_bf16_pin_v<true>(acc[m], fa, fb);
        else mfma_bf16_pin_v<false>(acc[m], fa, fb);
        if (KIND == DW_HIDDEN && m == 0 && (wave / NBR) == 0) dbacc += frag_sum<Pol>(fa);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// write the accumulated block(s) to the flat-layout destination (plain store when `first`, else +=)
template <class Pol, int H, int NBR, int NBC, int KIND, int MYB, int M0 = 0, bool DO_W = true, bool DO_B = true>
DEV void dw_flush(const f32x16 (&acc)[MYB], float dbacc, float* dst_w, float* dst_b, int ldw, bool first,
                  int wave, int lane, int amode = 0) {
  constexpr int NBLK = NBR * NBC;
  const int hi = lane >> 5, j = lane & 31;
#pragma unroll
  for (int m = 0; m < (DO_W ? MYB : 0); ++m) {
    const int blk = wave + 4 * (M0 + m);
    if (blk < NBLK) {
      const int rbo = blk % NBR, cb = blk / NBR;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * rbo + rowmap(r, hi);
        if constexpr (KIND == DW_HIDDEN) {
          float* p = dst_w + (size_t)row * ldw + 32 * cb + j;
          if (amode == 1) {
            __hip_atomic_fetch_add(p, acc[m][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else if (amode == 2) {
            const float t = acc[m][r] * 4096.f;  // 2^(44-32)
            const float h = floorf(t);
            const unsigned lo = (unsigned)((t - h) * 4294967296.f);
            const unsigned long long q = ((unsigned long long)(unsigned)(int)h << 32) | lo;
            __hip_atomic_fetch_add((unsigned long long*)dst_w + ((size_t)row * ldw + 32 * cb + j), q, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            *p = first ? acc[m][r] : (*p + acc[m][r]);
          }
        } else if constexpr (KIND == DW_HEAD) {
          if (j < 3) {
            float* p = dst_w + (size_t)j * ldw + row;
            *p = first ? acc[m][r] : (*p + acc[m][r]);
          }
        } else {
          if (j < 16) dst_w[(size_t)row * 16 + j] = acc[m][r];
        }
      }
    }
  }
  if constexpr (!DO_B) {
    return;
  } else if constexpr (KIND == DW_HIDDEN) {
    dbacc += __shfl_xor(dbacc, 32, 64);
    if (wave < NBR && wave < NBLK && hi == 0) {
      float* p = dst_b + 32 * (wave % NBR) + j;
      *p = first ? dbacc : (*p + dbacc);
    }
  } else if constexpr (KIND == DW_HEAD) {
    dbacc += __shfl_xor(dbacc, 32, 64);
    if (wave == 0 && hi == 0 && j < 3) {
      float* p = dst_b + j;
      *p = first ? dbacc : (*p + dbacc);
    }
  }
}

// dM[row-feature][col-feature] (+)= sum over the tile's 128 samples rows[.][s] * cols[.][s].
//   DW_HIDDEN: rows = g_a (layer outputs), cols = h_{l-1}; flushed to dW_l [out][in] and db_l
//   DW_HEAD  : rows = h_L, cols = g_y (3 real);            flushed to dW_out[c][feature], db_out
//   DW_L0    : rows = g_a0, cols = (x_hi[5], x_lo[4]);     stored to the per-tile dA partial
// dbg: ablation mask for profiling experiments (1 = no flush, 2 = no GEMM, 4 = no LDS transposition)
template <class Pol, int H, int NBR, int NBC, int KIND>
DEV void dw_phase(char* TA, char* TB, const float (&rows)[NBR][16], const float (&cols)[NBC][16],
                  float* dst_w, float* dst_b, int ldw, bool first, int wave, int lane, int dbg = 0) {
  constexpr int MYB = (NBR * NBC + 3) / 4;
  constexpr int NROUND = 4 / Geo<Pol, H>::ROUND_WAVES;
  const int hi = lane >> 5, j = lane & 31;
  if constexpr (MYB <= 4) {
    f32x16 acc[MYB];
    acc_zero<MYB>(acc);
    float dbacc = 0.f;
#pragma unroll 1
    for (int round = 0; round < NROUND; ++round) {
      __syncthreads();
      if (wave / Geo<Pol, H>::ROUND_WAVES == round && !(dbg & 4)) {
        t_write<Pol, H, NBR>(TA, rows, wave % Geo<Pol, H>::ROUND_WAVES, hi, j);
        t_write<Pol, H, NBC>(TB, cols, wave % Geo<Pol, H>::ROUND_WAVES, hi, j);
      }
      __syncthreads();
      if (dbg & 2) continue;
      dw_gemm<Pol, H, NBR, NBC, KIND, MYB>(TA, TB, acc, dbacc, wave, lane);
    }
    if (dbg & 1) {
      float keep = dbacc;
#pragma unroll
      for (int m = 0; m < MYB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep += acc[m][r];
      if (keep == 123.456f) dst_w[0] = keep;
      return;
    }
    dw_flush<Pol, H, NBR, NBC, KIND, MYB>(acc, dbacc, dst_w, dst_b, ldw, first, wave, lane, (dbg >> 8) & 3);
  } else {
    // wide layers (H = 256: 16 blocks per wave): four blocks at a time, every (round, group) result is added
    // to the partial right away -- more += traffic, bounded register use
    static_assert(MYB % 4 == 0, "block groups of four");
    static_assert(NBR == 8, "wide path: a wave owns row blocks w and w + 4");
    float dbacc = 0.f, dbacc2 = 0.f;
#pragma unroll 1
    for (int round = 0; round < NROUND; ++round) {
      __syncthreads();
      if (wave / Geo<Pol, H>::ROUND_WAVES == round) {
        t_write<Pol, H, NBR>(TA, rows, wave % Geo<Pol, H>::ROUND_WAVES, hi, j);
        t_write<Pol, H, NBC>(TB, cols, wave % Geo<Pol, H>::ROUND_WAVES, hi, j);
      }
      __syncthreads();
      const bool fw = first && round == 0;
      auto group = [&](auto gc) {
        constexpr int M0 = decltype(gc)::value * 4;
        f32x16 acc[4];
        acc_zero<4>(acc);
        dw_gemm<Pol, H, NBR, NBC, KIND, 4, 0, M0>(TA, TB, acc, dbacc, wave, lane, &dbacc2);
        dw_flush<Pol, H, NBR, NBC, KIND, 4, M0, true, false>(acc, 0.f, dst_w, dst_b, ldw, fw, wave, lane);
      };
      group(std::integral_constant<int, 0>{});
      if constexpr (MYB > 4) group(std::integral_constant<int, 1>{});
      if constexpr (MYB > 8) group(std::integral_constant<int, 2>{});
      if constexpr (MYB > 12) group(std::integral_constant<int, 3>{});
    }
    if constexpr (KIND == DW_HIDDEN) {  // bias gradients of row blocks w and w + 4
      dbacc += __shfl_xor(dbacc, 32, 64);
      dbacc2 += __shfl_xor(dbacc2, 32, 64);
      if (hi == 0) {
        float* p0 = dst_b + 32 * wave + j;
        float* p1 = dst_b + 32 * (wave + 4) + j;
        *p0 = first ? dbacc : (*p0 + dbacc);
        *p1 = first ? dbacc2 : (*p1 + dbacc2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// the fused kernel
// ------------------------------------------------------------------------------------------
enum { MODE_FWD = 0, MODE_STATS = 1, MODE_FWD_BWD = 2 };

template <class Pol, int H, int MODE>
__global__ void __launch_bounds__(256, (H > 128 ? 1 : Pol::WPS)) k_reni_main(const MainArgs a) {
  using G = Geo<Pol, H>;
  constexpr int NRB = G::NRB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const WB = smem;
  char* const TA = smem;  // aliases WB: the weight image is dead while a dW phase runs
  char* const TB = smem + G::WB_BYTES;
  float* const BL = (float*)(TB + G::T_BYTES_AL);  // bias of the layer being evaluated

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, hi = lane >> 5, j = lane & 31;
  char* const stash = a.stash + (size_t)blockIdx.x * a.stash_per_wg;
  float* dwp_sel = a.dwp + (size_t)blockIdx.x * a.dwp_per_wg;
  if (a.dbg & 0x300) {  // experiment: atomics into per-XCD (or one) shared partial buffers
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
    dwp_sel = a.dwp + (size_t)((a.dbg & 0x400) ? 0 : xcc) * 2 * a.dwp_per_wg;
  }
  float* const dwp = dwp_sel;
  const int L = a.L;
  bool first = true;

  // out[NRB] = image(l) x vin : the image is staged into LDS CH_RB row blocks at a time
  auto layer_gemm = [&](const char* img, const float (&vin)[NRB][16], f32x16 (&out)[NRB], bool with_bias) {
    auto chunk = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      __syncthreads();
      if (!(a.dbg & 32)) stage_to_lds(WB, img + c * G::CHUNK_BYTES, G::CHUNK_BYTES, tid);
      if (c == 0 && with_bias) stage_to_lds((char*)BL, img + G::IMG_HID, G::BIAS_HID, tid);
      __syncthreads();
      f32x16 part[G::CH_RB];
      if (with_bias) acc_init_bias<Pol, G::CH_RB>(part, BL + 32 * G::CH_RB * c, hi);
      else acc_zero<G::CH_RB>(part);
      gemm_lds<Pol, G::CH_RB, NRB, G::NKS>(WB, vin, part, lane);
#pragma unroll
      for (int r = 0; r < G::CH_RB; ++r) out[c * G::CH_RB + r] = part[r];
    };
    chunk(std::integral_constant<int, 0>{});
    if constexpr (G::NCHUNK > 1) chunk(std::integral_constant<int, 1>{});
    if constexpr (G::NCHUNK > 2) chunk(std::integral_constant<int, 2>{});
    if constexpr (G::NCHUNK > 3) chunk(std::integral_constant<int, 3>{});
    static_assert(G::NCHUNK <= 4, "at most four staging chunks");
  };

#pragma unroll 1
  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_image;
    const int p = (tile - b * a.tiles_per_image) * 128 + wave * 32 + j;
    const bool valid = p < a.P;

    // ---- direction -> x = (dx, dy, dz, r, 1)
    float x[5];
    {
      const float* dp = a.D + (size_t)b * a.d_bstride + (size_t)(valid ? p : 0) * 3;
      x[0] = dp[0]; x[1] = dp[1]; x[2] = dp[2];
      x[3] = sqrtf(x[0] * x[0] + x[2] * x[2]);
      x[4] = 1.f;
    }
    float v[NRB][16];
    // ---- layer 0: a0 = A_b x  (A_b = per-image affine map, k_prep_image)
    {
      f32x16 acc[NRB];
      acc_zero<NRB>(acc);
      const float* Ab = a.Apre + (size_t)b * H * 8;
      if constexpr (Pol::BF) {
        // x = xh + xl, A = Ah + Al (bf16 pairs); A x ~= Ah xh + Al xh + Ah xl in ONE k=16 MFMA
        __bf16 xh[5], xl[4];
#pragma unroll
        for (int k = 0; k < 5; ++k) xh[k] = (__bf16)x[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) xl[k] = (__bf16)(x[k] - (float)xh[k]);
        bf16x8 bop;
        if (hi == 0) { bop[0] = xh[0]; bop[1] = xh[1]; bop[2] = xh[2]; bop[3] = xh[3]; bop[4] = xh[4]; bop[5] = xh[0]; bop[6] = xh[1]; bop[7] = xh[2]; }
        else { bop[0] = xh[3]; bop[1] = xh[4]; bop[2] = xl[0]; bop[3] = xl[1]; bop[4] = xl[2]; bop[5] = xl[3]; bop[6] = (__bf16)0.f; bop[7] = (__bf16)0.f; }
#pragma unroll
        for (int rbo = 0; rbo < NRB; ++rbo) {
          const f32x4 lo4 = *(const f32x4*)(Ab + (32 * rbo + j) * 8);
          const float a4 = Ab[(32 * rbo + j) * 8 + 4];
          const float A5[5] = {lo4[0], lo4[1], lo4[2], lo4[3], a4};
          __bf16 Ah[5], Al[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) { Ah[k] = (__bf16)A5[k]; Al[k] = (__bf16)(A5[k] - (float)Ah[k]); }
          bf16x8 aop;
          if (hi == 0) { aop[0] = Ah[0]; aop[1] = Ah[1]; aop[2] = Ah[2]; aop[3] = Ah[3]; aop[4] = Ah[4]; aop[5] = Al[0]; aop[6] = Al[1]; aop[7] = Al[2]; }
          else { aop[0] = Al[3]; aop[1] = Al[4]; aop[2] = Ah[0]; aop[3] = Ah[1]; aop[4] = Ah[2]; aop[5] = Ah[3]; aop[6] = (__bf16)0.f; aop[7] = (__bf16)0.f; }
          acc[rbo] = PolBF16::mfma(aop, bop, acc[rbo]);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const int k = 2 * ks + hi;
          const float bx = (k == 0) ? x[0] : (k == 1) ? x[1] : (k == 2) ? x[2] : (k == 3) ? x[3] : (k == 4) ? x[4] : 0.f;
#pragma unroll
          for (int rbo = 0; rbo < NRB; ++rbo) {
            const float av = Ab[(32 * rbo + j) * 8 + k];
            acc[rbo] = PolF32::mfma(av, bx, acc[rbo]);
          }
        }
      }
      Pol::drain();
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[rb][r] = acc[rb][r];
    }
    act_forward<Pol, NRB, MODE == MODE_FWD_BWD>(v, a.w_first, stash, tid);

    // ---- hidden layers 1..L
#pragma unroll 1
    for (int l = 1; l <= L; ++l) {
      f32x16 acc[NRB];
      layer_gemm(a.wimg + a.fwd_off[l], v, acc, true);
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[rb][r] = acc[rb][r];
      act_forward<Pol, NRB, MODE == MODE_FWD_BWD>(v, a.w_hidden, stash + (size_t)l * G::STASH_LAYER_BYTES, tid);
    }

    // ---- head: y = W_out h_L + b_out (rows 0..2 of a 32-row block)
    float y[3], ylin[3], outv[3];
    {
      __syncthreads();
      stage_to_lds(WB, a.wimg + a.fwd_off[L + 1], G::IMG_HF, tid);
      stage_to_lds((char*)BL, a.wimg + a.fwd_off[L + 1] + G::IMG_HF, G::BIAS_HEAD, tid);
      __syncthreads();
      f32x16 acc[1];
      acc_init_bias<Pol, 1>(acc, BL, hi);
      gemm_lds<Pol, 1, NRB, G::NKS>(WB, v, acc, lane);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        ylin[c] = acc[0][c];
        y[c] = a.last_linear ? ylin[c] : sin_f32(a.w_hidden * ylin[c]);
        outv[c] = (a.act == 1) ? tanhf(y[c]) : (a.act == 2) ? expf(y[c]) : y[c];
      }
    }
    const bool owner = valid && (hi == 0);
    if (a.out != nullptr && owner) {
      float* op = a.out + ((size_t)b * a.P + p) * 3;
      op[0] = outv[0]; op[1] = outv[1]; op[2] = outv[2];
    }
    if constexpr (MODE == MODE_FWD) continue;

    // ---- loss and d loss / d out
    float gy[3] = {0.f, 0.f, 0.f};
    if constexpr (MODE == MODE_STATS) {
      float s9[9];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float t = owner ? a.target[(size_t)b * a.ts0 + (size_t)p * a.ts1 + (size_t)c * a.ts2] : 0.f;
        const float o = owner ? outv[c] : 0.f;
        s9[c] = o * t; s9[3 + c] = o * o; s9[6 + c] = t * t;
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const float s = wave_sum(s9[q]);
        if (lane == 0) a.loss_part[((size_t)tile * 4 + wave) * 16 + 1 + q] = s;
      }
      continue;
    } else {
      float e = 0.f;
      if (a.loss_kind == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gy[c] = owner ? a.dout[((size_t)b * a.P + p) * 3 + c] : 0.f;
      } else {
        const float inv3p = 1.0f / (3.0f * (float)a.P);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (owner) {
            const float t = a.target[(size_t)b * a.ts0 + (size_t)p * a.ts1 + (size_t)c * a.ts2];
            const float s = a.weight[(size_t)b * a.ws0 + (size_t)p * a.ws1 + (size_t)c * a.ws2];
            const float d = outv[c] - t;
            e += s * d * d;
            gy[c] = 2.f * s * d * inv3p;
            if (a.loss_kind == 1) {
              const float cA = a.stats[(size_t)b * 16 + c], cB = a.stats[(size_t)b * 16 + 4 + c];
              gy[c] += cA * t - cB * outv[c];
            }
          }
        }
      }
      const float es = wave_sum(e);
      if (lane == 0) a.loss_part[((size_t)tile * 4 + wave) * 16] = es;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (a.act == 1) gy[c] *= (1.f - outv[c] * outv[c]);
        else if (a.act == 2) gy[c] *= outv[c];
        if (!a.last_linear) gy[c] *= a.w_hidden * cos_f32(a.w_hidden * ylin[c]);
        if (!owner) gy[c] = 0.f;
      }
    }

    // =============================== backward ===============================
    float g[NRB][16];  // gradient w.r.t. the current layer's pre-activation, D layout
    f32x16 acc[NRB];
    {
      float gh[1][16];
#pragma unroll
      for (int r = 0; r < 16; ++r) gh[0][r] = (r < 3) ? gy[r] : 0.f;
      if (a.need_dw && !(a.dbg & 128))
        dw_phase<Pol, H, NRB, 1, DW_HEAD>(TA, TB, v, gh, dwp + a.p_off_w[L + 1], dwp + a.p_off_b[L + 1], H, first, wave, lane, a.dbg);
      __syncthreads();
      stage_to_lds(WB, a.wimg + a.bwd_off[L + 1], G::IMG_HB, tid);
      __syncthreads();
      acc_zero<NRB>(acc);
      gemm_lds<Pol, NRB, 1, Pol::HEAD_BWD_KS>(WB, gh, acc, lane);
    }
#pragma unroll 1
    for (int l = L; l >= 1; --l) {
      act_replay<Pol, NRB, 0>(g, acc, a.w_hidden, a.w_hidden, stash + (size_t)l * G::STASH_LAYER_BYTES, tid);
      if (a.need_dw && !(a.dbg & 128)) {
        float hp[NRB][16];
        act_replay<Pol, NRB, 1>(hp, nullptr, (l - 1 == 0) ? a.w_first : a.w_hidden, 0.f,
                                stash + (size_t)(l - 1) * G::STASH_LAYER_BYTES, tid);
        dw_phase<Pol, H, NRB, NRB, DW_HIDDEN>(TA, TB, g, hp, dwp + a.p_off_w[l], dwp + a.p_off_b[l], H, first, wave, lane, a.dbg);
      }
      layer_gemm(a.wimg + a.bwd_off[l], g, acc, false);
    }
    act_replay<Pol, NRB, 0>(g, acc, a.w_first, a.w_first, stash, tid);
    {
      float xc[1][16];
#pragma unroll
      for (int r = 0; r < 16; ++r) xc[0][r] = 0.f;
      if (hi == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if constexpr (Pol::BF) {
            const float xhk = (float)(__bf16)x[k];
            xc[0][k] = xhk; xc[0][4 + k] = x[k] - xhk;
          } else {
            xc[0][k] = x[k];
          }
        }
      } else {
        xc[0][0] = 1.f;
      }
      dw_phase<Pol, H, NRB, 1, DW_L0>(TA, TB, g, xc, a.dA_part + (size_t)tile * H * 16, nullptr, 16, true, wave, lane, a.dbg & ~3);
    }
    first = false;
  }
}

// ---- helpers of the bf16 training path ---------------------------------------------------------
template <int N, class F>
DEV void static_for_down(F&& f) {
  if constexpr (N >= 1) {
    f(std::integral_constant<int, N>{});
    static_for_down<N - 1>(f);
  }
}

template <int H>
struct GeoP {
  using G = Geo<PolBF16, H>;
  static constexpr int WBSZ = (G::IMG_HID + G::BIAS_HID + 1023) & ~1023;  // DMA moves 1 KB pieces
  static constexpr int TSZ = 128 * 256;  // sample-major image, see ts_write / ts_read
  static constexpr int LDS_BYTES = 2 * WBSZ + 2 * TSZ + 2048;  // + head dW accumulators
};

// async global -> LDS copy of `bytes` (rounded up to 1 KB; the image regions are padded) split over 4 waves
template <int BYTES>
DEV void dma_image(char* lds_dst, const char* gsrc, int wave, int lane16) {
  constexpr int NP = (BYTES + 1023) / 1024;
  const char* src = gsrc + lane16;  // lane16 is opaque per tile iteration: keeps this address math out of LICM
#pragma unroll
  for (int pi = 0; pi < (NP + 3) / 4; ++pi) {
    const int piece = pi * 4 + wave;
    if (piece < NP)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024),
                                       (__attribute__((address_space(3))) void*)(lds_dst + piece * 1024), 16, 0, 0);
  }
}

// acc[rbo] (+)= W-image[rbo] . bop   (row-block-outer order so the caller's per-row-block epilogue
// overlaps the next row block's MFMAs)
// All MFMAs of the training kernel name their accumulator's register file explicitly: the 256 AGPRs belong
// to the sixteen persistent dW tiles, so the chain GEMMs accumulate in VGPRs ("+v").  The last MFMA of a
// chain carries the wait states the following VALU reads of the result need.
template <int NKS>
DEV f32x16 gemm_rb(const char* wb, int rbo, const bf16x8 (&bop)[NKS], f32x16 acc, int lane) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    bf16x8 a = *(const bf16x8*)(wb + ((rbo * NKS + ks) * 64 + lane) * 16);
    if (ks == NKS - 1) mfma_bf16_pin_v<true>(acc, a, bop[ks]);
    else mfma_bf16_pin_v<false>(acc, a, bop[ks]);
  }
  return acc;
}

// two row blocks at once: their MFMA chains are independent, so alternating them keeps the matrix pipe
// busy instead of waiting out each dependent-accumulator latency
template <int NKS>
DEV void gemm_rb2(const char* wb, int rbo0, const bf16x8 (&bop)[NKS], f32x16& acc0, f32x16& acc1, int lane) {
  // A fragments are fetched two k-steps ahead of the MFMAs that use them (LDS latency ~ 2 MFMA pairs)
  auto fa = [&](int r, int ks) { return *(const bf16x8*)(wb + (((rbo0 + r) * NKS + ks) * 64 + lane) * 16); };
  constexpr int D = NKS >= 2 ? 2 : 1;
  bf16x8 q0[D], q1[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { q0[d] = fa(0, d); q1[d] = fa(1, d); }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const bf16x8 a0 = q0[ks % D], a1 = q1[ks % D];
    if (ks + D < NKS) { q0[ks % D] = fa(0, ks + D); q1[ks % D] = fa(1, ks + D); }
    mfma_bf16_pin_v<false>(acc0, a0, bop[ks]);
    if (ks == NKS - 1) mfma_bf16_pin_v<true>(acc1, a1, bop[ks]);
    else mfma_bf16_pin_v<false>(acc1, a1, bop[ks]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int H, int NB>
DEV void t_write_b(char* T, const bf16x8 (&xb)[2 * NB], int wave, int hi, int j) {
  using G = Geo<PolBF16, H>;
#pragma unroll
  for (int rb = 0; rb < NB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      *(__bf16*)(T + (32 * rb + rowmap(r, hi)) * G::T_ROWB + (wave * 32 + j) * 2) = xb[2 * rb + (r >> 3)][r & 7];
}

// phase (revolutions, u16) pair in one dword -> two floats in [1,2): sin/cos of 2*pi*x are periodic in 1
DEV float phase_lo(unsigned w) { return __uint_as_float(((w & 0xffffu) << 7) | 0x3f800000u); }
DEV float phase_hi(unsigned w) { return __uint_as_float(((w >> 9) & 0x007fff80u) | 0x3f800000u); }

// ---- sample-major transposition image of the bf16 training path (H = 128) --------------------------
// Row s (one of the tile's 128 samples) = 256 B = 32 pieces of 8 B; logical piece c holds features
// 4c..4c+3 of that sample.  Pieces are stored at c ^ tswz(s), which makes both the 8-byte writes (a lane
// owns a sample) and the ds_read_b64_tr_b16 transpose reads (a lane ends up owning a feature) free of LDS
// bank conflicts.  A fragment read returns, to lane l, feature 32*blk + (l & 31) for
// samples 16*ks + 8*(l >> 5) + 0..7 -- the MFMA operand layout with K = samples.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int TS_BYTES = 128 * 256;
DEV int tswz(int s) { return ((s & 1) << 4) | ((s & 2) << 2) | ((s & 1) << 2) | ((s & 8) >> 2) | ((s & 4) >> 2); }

// write NB 32-feature blocks of packed bf16 (D layout) for sample s
template <int NB>
DEV void ts_write(char* T, const bf16x8 (&xb)[2 * NB], int s, int hi) {
  char* row = T + s * 256;
  const int x = tswz(s);
#pragma unroll
  for (int rb = 0; rb < NB; ++rb)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const u32x4 w = __builtin_bit_cast(u32x4, xb[2 * rb + (u >> 1)]);
      const u32x2 h = (u & 1) ? u32x2{w[2], w[3]} : u32x2{w[0], w[1]};
      *(u32x2*)(row + (((8 * rb + 2 * u + hi) ^ x) << 3)) = h;
    }
}
// the two 8-byte halves of one bf16x8 (features 8*(c&1).. of block c>>1, i.e. chunk index c) for sample s
DEV void ts_write_chunk(char* T, bf16x8 v, int c, int s, int hi) {
  char* row = T + s * 256;
  const int x = tswz(s);
  const u32x4 w = __builtin_bit_cast(u32x4, v);
  const int rb = c >> 1, u0 = 2 * (c & 1);
  *(u32x2*)(row + (((8 * rb + 2 * u0 + hi) ^ x) << 3)) = u32x2{w[0], w[1]};
  *(u32x2*)(row + (((8 * rb + 2 * (u0 + 1) + hi) ^ x) << 3)) = u32x2{w[2], w[3]};
}
// per-lane byte offsets (block 0, k-step 0) of the two transpose reads that make one fragment
DEV void ts_lane_offsets(int lane, int (&a0)[2]) {
  const int khi = lane >> 5, g = (lane >> 4) & 1, u = lane & 15;
  const int s0 = (u >> 2) & 1, s1 = (u >> 3) & 1;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int x = (s0 << 4) | (s1 << 3) | (s0 << 2) | (khi << 1) | q;
    const int phys = ((4 * g + (u & 3)) ^ (x & 7)) | ((x >> 3) << 3);
    a0[q] = (8 * khi + 4 * q + (u >> 2)) * 256 + phys * 8;
  }
}
DEV bf16x8 ts_read(const char* T, const int (&a0)[2], int blk, int ks) {
  typedef __attribute__((address_space(3))) s16x4* lp;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(T + ((a0[0] ^ (blk << 6)) + ks * 4096)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(T + ((a0[1] ^ (blk << 6)) + ks * 4096)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// K = samples GEMM from sample-major images: rows block(s) x cols block(s), 128 samples.
// Wave w owns row block w (NBR == 4) and all NBC column blocks.  PIN: 0 compiler-allocated accumulator,
// 2 accumulator pinned to VGPRs (asm MFMA).
template <int NBC, int KIND, int PIN>
DEV void dw_gemm_s(const char* TA, const char* TB, f32x16 (&acc)[NBC], float& dbacc, int wave, int lane) {
  int a0[2];
  ts_lane_offsets(lane, a0);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const bf16x8 fa = ts_read(TA, a0, wave, ks);
#pragma unroll
    for (int m = 0; m < NBC; ++m) {
      const bf16x8 fb = ts_read(TB, a0, m, ks);
      if constexpr (PIN == 2) {
        if (ks == 7 && m == NBC - 1) mfma_bf16_pin_v<true>(acc[m], fa, fb);
        else mfma_bf16_pin_v<false>(acc[m], fa, fb);
      } else {
        acc[m] = PolBF16::mfma(fa, fb, acc[m]);
      }
      if (KIND == DW_HEAD && m == 0 && wave == 0) dbacc += frag_sum<PolBF16>(fb);
    }
    if (KIND == DW_HIDDEN) dbacc += frag_sum<PolBF16>(fa);
    if constexpr (PIN != 0) __builtin_amdgcn_sched_barrier(0);
  }
}

// hidden-layer dW into the hand-owned AGPR tiles BASE..BASE+3 (rows = this wave's block, 4 column blocks)
template <int BASE>
DEV void dw_gemm_s_agpr(const char* TA, const char* TB, float& dbacc, int wave, int lane) {
  int a0[2];
  ts_lane_offsets(lane, a0);
  bf16x8 fa = ts_read(TA, a0, wave, 0);
  bf16x8 f0 = ts_read(TB, a0, 0, 0), f1 = ts_read(TB, a0, 1, 0), f2 = ts_read(TB, a0, 2, 0), f3 = ts_read(TB, a0, 3, 0);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    bf16x8 na = fa, n0 = f0, n1 = f1, n2 = f2, n3 = f3;
    if (ks < 7) {  // next k-step's fragments are in flight while this k-step's MFMAs run
      na = ts_read(TA, a0, wave, ks + 1);
      n0 = ts_read(TB, a0, 0, ks + 1); n1 = ts_read(TB, a0, 1, ks + 1);
      n2 = ts_read(TB, a0, 2, ks + 1); n3 = ts_read(TB, a0, 3, ks + 1);
    }
    mfma_bf16_agpr_tile<BASE + 0, false>(fa, f0);
    mfma_bf16_agpr_tile<BASE + 1, false>(fa, f1);
    mfma_bf16_agpr_tile<BASE + 2, false>(fa, f2);
    if (ks == 7) mfma_bf16_agpr_tile<BASE + 3, true>(fa, f3);
    else mfma_bf16_agpr_tile<BASE + 3, false>(fa, f3);
    dbacc += frag_sum<PolBF16>(fa);
    __builtin_amdgcn_sched_barrier(0);
    fa = na; f0 = n0; f1 = n1; f2 = n2; f3 = n3;
  }
}

// ==========================================================================================
// bf16 TRAINING kernel with register-persistent weight-gradient accumulators (H = 128, L <= 5)
// ==========================================================================================
// Same arithmetic as k_reni_main<PolBF16, 128, MODE_FWD_BWD>, different schedule:
//   * the dW_l accumulators of all hidden layers (5 x 128 x 128 fp32 = 320 KB per workgroup, 62 % of a
//     CU's register file) stay in registers across the workgroup's whole tile loop and are written
//     ONCE per launch.  hipcc allocates every MFMA accumulator of a kernel in AGPRs (256 = sixteen
//     32x32 tiles), so the tiles are pinned by hand: layers 2..5 in AGPRs, layer 1 in VGPRs
//     (mfma_bf16_agpr_tile).  The generic kernel's per-tile "+=" flush of those partials was 65 % of
//     its run time (profiles/r01_b_ablation.md);
//   * one workgroup (4 waves, one per SIMD, up to 512 registers each) per CU;
//   * activations / gradients travel between layers as packed bf16 MFMA operands;
//   * weight images are double-buffered in LDS and fetched one step ahead by LDS-DMA
//     (global_load_lds): one barrier per forward layer, two per backward layer, no staging stall.
template <int H>
__global__ void __launch_bounds__(256, 1) k_reni_train_bf16(const MainArgs a) {
  using Pol = PolBF16;
  using G = Geo<Pol, H>;
  using GP = GeoP<H>;
  static_assert(H == 128, "sample-major transposition image is laid out for H = 128");
  constexpr int NRB = G::NRB, NKS = G::NKS;
  constexpr int MYB_H = (NRB * NRB + 3) / 4;
  constexpr int PMAX = 5;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const TA = smem + 2 * GP::WBSZ;
  char* const TB = TA + GP::TSZ;
  float* const hdw = (float*)(TB + GP::TSZ);  // [3][H] dW_out + [3] db_out accumulated over the tile loop

#ifdef RENI_TRACE
  int trace_n = 0;
#define TRACE(tag) do { __builtin_amdgcn_sched_barrier(0); if (a.trace && blockIdx.x == 0 && threadIdx.x == 0 && trace_n < 250) { a.trace[2 * trace_n] = (long long)(tag); a.trace[2 * trace_n + 1] = (long long)__builtin_amdgcn_s_memtime(); ++trace_n; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TRACE(tag) do {} while (0)
#endif
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, hi = lane >> 5, j = lane & 31;
  char* const stash = a.stash + (size_t)blockIdx.x * a.stash_per_wg;
  float* const dwp = a.dwp + (size_t)blockIdx.x * a.dwp_per_wg;
  const int L = a.L;
  const int nstep = 2 * L + 2;  // weight images consumed per tile

  static_assert(MYB_H == 4, "sixteen AGPR tiles = 4 layers x 4 column blocks");
  agpr_zero_seq(std::make_integer_sequence<int, 256>{});  // dW_2..dW_5 accumulators: a[0:255]
  float pdb[PMAX];
#pragma unroll
  for (int l = 0; l < PMAX; ++l) pdb[l] = 0.f;
  for (int i = tid; i < 3 * H + 4; i += 256) hdw[i] = 0.f;

  int lane16 = lane * 16, tid16 = tid * 16;  // re-made opaque at the top of every tile iteration
  // image of step k of a tile: k < L hidden fwd (layer k+1), k == L head fwd, k == L+1 head bwd,
  // then hidden bwd for layer 2L+2-k.  Buffer = k & 1.
  auto issue_dma = [&](int k) {
    char* dst = smem + (k & 1) * GP::WBSZ;
    if (k < L) dma_image<G::IMG_HID + G::BIAS_HID>(dst, a.wimg + a.fwd_off[k + 1], wave, lane16);
    else if (k == L) dma_image<G::IMG_HF + G::BIAS_HEAD>(dst, a.wimg + a.fwd_off[L + 1], wave, lane16);
    else if (k == L + 1) dma_image<G::IMG_HB>(dst, a.wimg + a.bwd_off[L + 1], wave, lane16);
    else dma_image<G::IMG_HID>(dst, a.wimg + a.bwd_off[2 * L + 2 - k], wave, lane16);
  };
  issue_dma(0);
  const float sc_first = a.w_first * 0.15915494309189535f, sc_hidden = a.w_hidden * 0.15915494309189535f;

#pragma unroll 1
  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    asm volatile("" : "+v"(lane16), "+v"(tid16));
    TRACE(1);
    const int b = tile / a.tiles_per_image;
    const int p = (tile - b * a.tiles_per_image) * 128 + wave * 32 + j;
    const bool valid = p < a.P;
    float x[5];
    {
      const float* dp = a.D + (size_t)b * a.d_bstride + (size_t)(valid ? p : 0) * 3;
      x[0] = dp[0]; x[1] = dp[1]; x[2] = dp[2];
      x[3] = sqrtf(x[0] * x[0] + x[2] * x[2]);
      x[4] = 1.f;
    }
    const bool owner = valid && (hi == 0);
    float tgt[3] = {0.f, 0.f, 0.f}, swt[3] = {0.f, 0.f, 0.f};  // issued now, consumed after the forward pass
    if (owner) {
      if (a.loss_kind == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) tgt[c] = a.dout[((size_t)b * a.P + p) * 3 + c];
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          tgt[c] = a.target[(size_t)b * a.ts0 + (size_t)p * a.ts1 + (size_t)c * a.ts2];
          swt[c] = a.weight[(size_t)b * a.ws0 + (size_t)p * a.ws1 + (size_t)c * a.ws2];
        }
      }
    }
    __bf16 xh[5], xl[4];
#pragma unroll
    for (int k = 0; k < 5; ++k) xh[k] = (__bf16)x[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) xl[k] = (__bf16)(x[k] - (float)xh[k]);

    bf16x8 hb[NKS];       // current activations as MFMA B operands
    u32x4 qst[NRB * 2];   // phases of the layer just computed; stored at the START of the next step so
                          // that the barrier after it never waits for these stores (vmcnt counts stores)

    // activation of one finished row block: keep the phase (for the stash) and the bf16 sin
    auto activate = [&](const f32x16& acc, int rb, float sc, bf16x8 (&dst)[NKS]) {
      float th[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        th[r] = __builtin_amdgcn_fractf(acc[r] * sc);
        dst[2 * rb + (r >> 3)][r & 7] = (__bf16)__builtin_amdgcn_sinf(th[r]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d)  // two phases -> two unorm16 in ONE instruction (v_cvt_pknorm_u16_f32)
          qst[2 * rb + c][d] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(th[8 * c + 2 * d], th[8 * c + 2 * d + 1]));
    };
    auto store_stash = [&](int layer) {
      char* sl = stash + (size_t)layer * G::STASH_LAYER_BYTES;
#pragma unroll
      for (int c = 0; c < NRB * 2; ++c) *(u32x4*)(sl + c * 4096 + tid16) = qst[c];
    };

    // ---- layer 0 (per-image affine map, split-bf16 operands: see k_reni_main)
    {
      const float* Ab = a.Apre + (size_t)b * H * 8;
      bf16x8 bop;
      if (hi == 0) { bop[0] = xh[0]; bop[1] = xh[1]; bop[2] = xh[2]; bop[3] = xh[3]; bop[4] = xh[4]; bop[5] = xh[0]; bop[6] = xh[1]; bop[7] = xh[2]; }
      else { bop[0] = xh[3]; bop[1] = xh[4]; bop[2] = xl[0]; bop[3] = xl[1]; bop[4] = xl[2]; bop[5] = xl[3]; bop[6] = (__bf16)0.f; bop[7] = (__bf16)0.f; }
#pragma unroll
      for (int rbo = 0; rbo < NRB; ++rbo) {
        const f32x4 lo4 = *(const f32x4*)(Ab + (32 * rbo + j) * 8);
        const float a4 = Ab[(32 * rbo + j) * 8 + 4];
        const float A5[5] = {lo4[0], lo4[1], lo4[2], lo4[3], a4};
        __bf16 Ah[5], Al[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { Ah[k] = (__bf16)A5[k]; Al[k] = (__bf16)(A5[k] - (float)Ah[k]); }
        bf16x8 aop;
        if (hi == 0) { aop[0] = Ah[0]; aop[1] = Ah[1]; aop[2] = Ah[2]; aop[3] = Ah[3]; aop[4] = Ah[4]; aop[5] = Al[0]; aop[6] = Al[1]; aop[7] = Al[2]; }
        else { aop[0] = Al[3]; aop[1] = Al[4]; aop[2] = Ah[0]; aop[3] = Ah[1]; aop[4] = Ah[2]; aop[5] = Ah[3]; aop[6] = (__bf16)0.f; aop[7] = (__bf16)0.f; }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mfma_bf16_pin_v<true>(acc, aop, bop);
        activate(acc, rbo, sc_first, hb);
      }
    }

    TRACE(2);
    // ---- hidden layers (step k = l - 1)
#pragma unroll 1
    for (int l = 1; l <= L; ++l) {
      __syncthreads();  // image of this step landed (vmcnt(0) before the barrier); the other buffer is free
      TRACE(10 + l);
      store_stash(l - 1);
      issue_dma(l);
      const char* wb = smem + ((l - 1) & 1) * GP::WBSZ;
      const float* bias = (const float*)(wb + G::IMG_HID);
      bf16x8 hn[NKS];
#pragma unroll
      for (int rbo = 0; rbo < NRB; rbo += 2) {
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc0[r] = bias[32 * rbo + rowmap(r, hi)];
          acc1[r] = bias[32 * rbo + 32 + rowmap(r, hi)];
        }
        gemm_rb2<NKS>(wb, rbo, hb, acc0, acc1, lane);
        activate(acc0, rbo, sc_hidden, hn);
        activate(acc1, rbo + 1, sc_hidden, hn);
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) hb[ks] = hn[ks];
    }

    TRACE(3);
    // ---- head (step L)
    float ylin[3], outv[3];
    {
      __syncthreads();
      TRACE(4);
      store_stash(L);
      issue_dma(L + 1);
      const char* wb = smem + (L & 1) * GP::WBSZ;
      const float* bias = (const float*)(wb + G::IMG_HF);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bias[rowmap(r, hi)];
      acc = gemm_rb<NKS>(wb, 0, hb, acc, lane);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        ylin[c] = acc[c];
        const float yv = a.last_linear ? ylin[c] : sin_f32(a.w_hidden * ylin[c]);
        // tanh(y) = 1 - 2 / (exp(2y) + 1) on the hardware exp2 / rcp (abs error ~1e-6, far inside bf16's)
        const float e2 = __builtin_amdgcn_exp2f(fminf(yv * 2.885390081777927f, 60.f));
        outv[c] = (a.act == 1) ? (1.f - 2.f * __builtin_amdgcn_rcpf(e2 + 1.f)) : (a.act == 2) ? expf(yv) : yv;
      }
    }
    if (a.out != nullptr && owner) {
      float* op = a.out + ((size_t)b * a.P + p) * 3;
      op[0] = outv[0]; op[1] = outv[1]; op[2] = outv[2];
    }

    TRACE(5);
    // ---- loss and d loss / d out
    float gy[3] = {0.f, 0.f, 0.f};
    {
      float e = 0.f;
      if (a.loss_kind == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gy[c] = tgt[c];
      } else {
        const float inv3p = 1.0f / (3.0f * (float)a.P);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (owner) {
            const float t = tgt[c], sw = swt[c];
            const float d = outv[c] - t;
            e += sw * d * d;
            gy[c] = 2.f * sw * d * inv3p;
            if (a.loss_kind == 1) {
              const float cA = a.stats[(size_t)b * 16 + c], cB = a.stats[(size_t)b * 16 + 4 + c];
              gy[c] += cA * t - cB * outv[c];
            }
          }
        }
      }
      const float es = wave_sum(e);
      if (lane == 0) a.loss_part[((size_t)tile * 4 + wave) * 16] = es;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (a.act == 1) gy[c] *= (1.f - outv[c] * outv[c]);
        else if (a.act == 2) gy[c] *= outv[c];
        if (!a.last_linear) gy[c] *= a.w_hidden * cos_f32(a.w_hidden * ylin[c]);
        if (!owner) gy[c] = 0.f;
      }
    }

    TRACE(6);
    // =============================== backward ===============================
    bf16x8 gb[NKS];   // gradient w.r.t. the current layer's pre-activation (g_a), packed
    bf16x8 ghb[NKS];  // dX result of the current step (g_h), packed as soon as a row block is done
    // one row block of a dX GEMM -> packed bf16 (only ONE fp32 accumulator tile is ever live)
    auto dx_rowblock = [&](const char* wb, int rbi, auto& bop, auto nks_tag) {  // row blocks rbi, rbi+1
      constexpr int NK = decltype(nks_tag)::value;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      gemm_rb2<NK>(wb, rbi, bop, acc0, acc1, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ghb[2 * rbi + (r >> 3)][r & 7] = (__bf16)acc0[r];
        ghb[2 * rbi + 2 + (r >> 3)][r & 7] = (__bf16)acc1[r];
      }
    };
    // g_a = g_h * omega * cos(phase) for one layer's stash
    auto load_phases = [&](const char* sl, u32x4 (&w)[NRB * 2]) {  // all loads in flight together
#pragma unroll
      for (int c = 0; c < NRB * 2; ++c) w[c] = *(const u32x4*)(sl + c * 4096 + tid16);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto make_g = [&](const u32x4 (&w)[NRB * 2], float omega) {
#pragma unroll
      for (int c = 0; c < NRB * 2; ++c) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          gb[c][2 * d] = (__bf16)((float)ghb[c][2 * d] * omega * __builtin_amdgcn_cosf(phase_lo(w[c][d])));
          gb[c][2 * d + 1] = (__bf16)((float)ghb[c][2 * d + 1] * omega * __builtin_amdgcn_cosf(phase_hi(w[c][d])));
        }
      }
    };
    // ---- head (step L+1): dW_out += h_L g_y^T ; g_h = W_out^T g_y
    {
      bf16x8 gyb[2];
#pragma unroll
      for (int e = 0; e < 8; ++e) { gyb[0][e] = (__bf16)((e < 3) ? gy[e] : 0.f); gyb[1][e] = (__bf16)0.f; }
      __syncthreads();  // B1: TA/TB free (previous tile's layer-0 phase done)
      ts_write<NRB>(TA, hb, wave * 32 + j, hi);
      ts_write<1>(TB, gyb, wave * 32 + j, hi);
      __syncthreads();  // B2
      TRACE(7);
      issue_dma(L + 2 < nstep ? L + 2 : 0);
      {
        f32x16 ah[1];
        acc_zero<1>(ah);
        float dbh = 0.f;
        dw_gemm_s<1, DW_HEAD, 2>(TA, TB, ah, dbh, wave, lane);
        if (j < 3) {  // element (c = j, feature) is owned by exactly one lane of one wave: plain LDS +=
#pragma unroll
          for (int r = 0; r < 16; ++r) hdw[j * H + 32 * wave + rowmap(r, hi)] += ah[0][r];
        }
        dbh += __shfl_xor(dbh, 32, 64);
        if (wave == 0 && hi == 0 && j < 3) hdw[3 * H + j] += dbh;
      }
      const char* wb = smem + ((L + 1) & 1) * GP::WBSZ;
      const bf16x8 g1[1] = {gyb[0]};
#pragma unroll
      for (int rbi = 0; rbi < NRB; rbi += 2) dx_rowblock(wb, rbi, g1, std::integral_constant<int, 1>{});
    }
    // ---- hidden layers l = L..1 (step k = 2L+2-l)
#pragma unroll 1
    for (int l = L; l >= 1; --l) {
      const int k = 2 * L + 2 - l;
      TRACE(20 + l);
      u32x4 wph[NRB * 2];
      load_phases(stash + (size_t)l * G::STASH_LAYER_BYTES, wph);
      make_g(wph, a.w_hidden);
      load_phases(stash + (size_t)(l - 1) * G::STASH_LAYER_BYTES, wph);  // for h_{l-1}; lands while we sync
      if (l == 1) {  // the sixteen AGPR tiles hold dW_2..dW_5; dW_1 = sum g_1 h_0^T is finished by k_reni_dw1
        char* gp = a.g1 + (size_t)tile * (NKS * 4096);
#pragma unroll
        for (int c = 0; c < NKS; ++c) *(bf16x8*)(gp + c * 4096 + tid16) = gb[c];
      }
      TRACE(30 + l);
      __syncthreads();  // B1: every wave finished the previous step's LDS reads
      TRACE(40 + l);
      ts_write<NRB>(TA, gb, wave * 32 + j, hi);
      TRACE(80 + l);
      // h_{l-1} = sin(phase_{l-1}) straight into the transposed image (dW_l's column operand)
#pragma unroll
      for (int c = 0; c < NRB * 2; ++c) {
        bf16x8 hv;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          hv[2 * d] = (__bf16)__builtin_amdgcn_sinf(phase_lo(wph[c][d]));
          hv[2 * d + 1] = (__bf16)__builtin_amdgcn_sinf(phase_hi(wph[c][d]));
        }
        ts_write_chunk(TB, hv, c, wave * 32 + j, hi);
      }
      TRACE(50 + l);
      __syncthreads();  // B2
      TRACE(60 + l);
      if (k + 1 < nstep) issue_dma(k + 1);
      else if (tile + (int)gridDim.x < a.n_tiles) issue_dma(0);
      switch (l) {  // one static register set per layer
        case 5: dw_gemm_s_agpr<12>(TA, TB, pdb[4], wave, lane); break;
        case 4: dw_gemm_s_agpr<8>(TA, TB, pdb[3], wave, lane); break;
        case 3: dw_gemm_s_agpr<4>(TA, TB, pdb[2], wave, lane); break;
        case 2: dw_gemm_s_agpr<0>(TA, TB, pdb[1], wave, lane); break;
        default: break;  // layer 1: handled by k_reni_dw1 from the g_1 stream stored below
      }
      TRACE(70 + l);
      const char* wb = smem + (k & 1) * GP::WBSZ;
#pragma unroll
      for (int rbi = 0; rbi < NRB; rbi += 2) dx_rowblock(wb, rbi, gb, std::integral_constant<int, NKS>{});
    }
    TRACE(8);
    // ---- layer 0: dA (per tile) = g_0 (x_hi | x_lo)^T
    {
      u32x4 wph[NRB * 2];
      load_phases(stash, wph);
      make_g(wph, a.w_first);
      bf16x8 xcb[2];
#pragma unroll
      for (int e = 0; e < 8; ++e) { xcb[0][e] = (__bf16)0.f; xcb[1][e] = (__bf16)0.f; }
      if (hi == 0) {
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) { xcb[0][k2] = xh[k2]; xcb[0][4 + k2] = xl[k2]; }
      } else {
        xcb[0][0] = (__bf16)1.f;
      }
      __syncthreads();
      ts_write<NRB>(TA, gb, wave * 32 + j, hi);
      ts_write<1>(TB, xcb, wave * 32 + j, hi);
      __syncthreads();
      f32x16 accA[1];
      acc_zero<1>(accA);
      float dummy = 0.f;
      dw_gemm_s<1, DW_L0, 2>(TA, TB, accA, dummy, wave, lane);
      dw_flush<Pol, H, NRB, 1, DW_L0, 1>(accA, 0.f, a.dA_part + (size_t)tile * H * 16, nullptr, 16, true, wave, lane);
    }
    TRACE(9);
  }
  // ---- the only write of the hidden-layer weight-gradient partials: once per workgroup per launch
  __syncthreads();
  for (int i = tid; i < 3 * H; i += 256) dwp[a.p_off_w[L + 1] + i] = hdw[i];
  if (tid < 3) dwp[a.p_off_b[L + 1] + tid] = hdw[3 * H + tid];
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  auto flush_layer = [&](auto lc) {
    constexpr int l = decltype(lc)::value;
    if (l <= L) {
      f32x16 t[MYB_H] = {agpr_read_tile<4 * (l - 2) + 0>(), agpr_read_tile<4 * (l - 2) + 1>(),
                         agpr_read_tile<4 * (l - 2) + 2>(), agpr_read_tile<4 * (l - 2) + 3>()};
      dw_flush<Pol, H, NRB, NRB, DW_HIDDEN, MYB_H>(t, pdb[l - 1], dwp + a.p_off_w[l], dwp + a.p_off_b[l], H, true, wave, lane);
    }
  };
  flush_layer(std::integral_constant<int, 2>{});
  flush_layer(std::integral_constant<int, 3>{});
  flush_layer(std::integral_constant<int, 4>{});
  flush_layer(std::integral_constant<int, 5>{});
}
template __global__ void k_reni_train_bf16<128>(const MainArgs);

// dW_1 / db_1 for the persistent training path: K = samples GEMM of the stored g_1 stream (bf16,
// D layout, written by k_reni_train_bf16) with h_0 = sin(w0 (A_b x)) recomputed from the directions.
// Persistent workgroups; the 128 x 128 accumulator stays in registers over all of a workgroup's tiles.
template <int H>
__global__ void __launch_bounds__(256, 1) k_reni_dw1(const MainArgs a) {
  using Pol = PolBF16;
  using G = Geo<Pol, H>;
  constexpr int NRB = G::NRB, NKS = G::NKS;
  constexpr int MYB_H = (NRB * NRB + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const TA = smem;
  char* const TB = smem + TS_BYTES;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, hi = lane >> 5, j = lane & 31;
  f32x16 acc[MYB_H];
  acc_zero<MYB_H>(acc);
  float db = 0.f;
  const float sc_first = a.w_first * 0.15915494309189535f;

  // inputs of a tile: the g_1 fragments, this lane's direction and its four rows of A_b.  They are
  // fetched one tile ahead so their HBM latency hides behind the previous tile's work.
  bf16x8 gbN[NKS];
  float dN[3];
  f32x4 aloN[NRB];
  float a4N[NRB];
  auto fetch = [&](int tile) {
    const int b = tile / a.tiles_per_image;
    const int p = (tile - b * a.tiles_per_image) * 128 + wave * 32 + j;
    const float* dp = a.D + (size_t)b * a.d_bstride + (size_t)(p < a.P ? p : 0) * 3;
    dN[0] = dp[0]; dN[1] = dp[1]; dN[2] = dp[2];
    const char* gp = a.g1 + (size_t)tile * (NKS * 4096);
#pragma unroll
    for (int c = 0; c < NKS; ++c) gbN[c] = *(const bf16x8*)(gp + c * 4096 + tid * 16);
    const float* Ab = a.Apre + (size_t)b * H * 8;
#pragma unroll
    for (int rbo = 0; rbo < NRB; ++rbo) {
      aloN[rbo] = *(const f32x4*)(Ab + (32 * rbo + j) * 8);
      a4N[rbo] = Ab[(32 * rbo + j) * 8 + 4];
    }
  };
  if ((int)blockIdx.x < a.n_tiles) fetch(blockIdx.x);
#pragma unroll 1
  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    bf16x8 gb[NKS];
    float x[5];
    f32x4 alo[NRB];
    float a4[NRB];
#pragma unroll
    for (int c = 0; c < NKS; ++c) gb[c] = gbN[c];
    x[0] = dN[0]; x[1] = dN[1]; x[2] = dN[2];
#pragma unroll
    for (int rbo = 0; rbo < NRB; ++rbo) { alo[rbo] = aloN[rbo]; a4[rbo] = a4N[rbo]; }
    if (tile + (int)gridDim.x < a.n_tiles) fetch(tile + gridDim.x);
    x[3] = sqrtf(x[0] * x[0] + x[2] * x[2]);
    x[4] = 1.f;
    __bf16 xh[5], xl[4];
#pragma unroll
    for (int k = 0; k < 5; ++k) xh[k] = (__bf16)x[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) xl[k] = (__bf16)(x[k] - (float)xh[k]);
    bf16x8 hb[NKS];
    {
      bf16x8 bop;
      if (hi == 0) { bop[0] = xh[0]; bop[1] = xh[1]; bop[2] = xh[2]; bop[3] = xh[3]; bop[4] = xh[4]; bop[5] = xh[0]; bop[6] = xh[1]; bop[7] = xh[2]; }
      else { bop[0] = xh[3]; bop[1] = xh[4]; bop[2] = xl[0]; bop[3] = xl[1]; bop[4] = xl[2]; bop[5] = xl[3]; bop[6] = (__bf16)0.f; bop[7] = (__bf16)0.f; }
#pragma unroll
      for (int rbo = 0; rbo < NRB; ++rbo) {
        const float A5[5] = {alo[rbo][0], alo[rbo][1], alo[rbo][2], alo[rbo][3], a4[rbo]};
        __bf16 Ah[5], Al[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { Ah[k] = (__bf16)A5[k]; Al[k] = (__bf16)(A5[k] - (float)Ah[k]); }
        bf16x8 aop;
        if (hi == 0) { aop[0] = Ah[0]; aop[1] = Ah[1]; aop[2] = Ah[2]; aop[3] = Ah[3]; aop[4] = Ah[4]; aop[5] = Al[0]; aop[6] = Al[1]; aop[7] = Al[2]; }
        else { aop[0] = Al[3]; aop[1] = Al[4]; aop[2] = Ah[0]; aop[3] = Ah[1]; aop[4] = Ah[2]; aop[5] = Ah[3]; aop[6] = (__bf16)0.f; aop[7] = (__bf16)0.f; }
        f32x16 a0;
#pragma unroll
        for (int r = 0; r < 16; ++r) a0[r] = 0.f;
        a0 = PolBF16::mfma(aop, bop, a0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // same phase quantisation as the training kernel's stash, so h_0 is bit-identical to what it used
          const float th = __builtin_amdgcn_fractf(a0[r] * sc_first);
          const unsigned q = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(th, 0.f)) & 0xffffu;
          hb[2 * rbo + (r >> 3)][r & 7] = (__bf16)__builtin_amdgcn_sinf(__uint_as_float((q << 7) | 0x3f800000u));
        }
      }
    }
    __syncthreads();
    ts_write<NRB>(TA, gb, wave * 32 + j, hi);
    ts_write<NRB>(TB, hb, wave * 32 + j, hi);
    __syncthreads();
    dw_gemm_s<NRB, DW_HIDDEN, 0>(TA, TB, acc, db, wave, lane);
  }
  float* const dwp = a.dwp + (size_t)blockIdx.x * a.dwp_per_wg;
  dw_flush<Pol, H, NRB, NRB, DW_HIDDEN, MYB_H>(acc, db, dwp + a.p_off_w[1], dwp + a.p_off_b[1], H, true, wave, lane);
}
template __global__ void k_reni_dw1<128>(const MainArgs);

// ------------------------------------------------------------------------------------------
// prologue: per-image constant input columns and the affine map A_b
// ------------------------------------------------------------------------------------------
// xconst[b][col] = value of input column col if it is constant over the image, else 0
// A[b][h][0..4] = coefficients of (dx, dy, dz, r, 1)
__global__ void __launch_bounds__(256) k_prep_image(const PrepArgs a) {
  // grid (B, H/8): every block rebuilds the image's constant columns in LDS (cheap) and does 8 rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xc = (float*)smem;  // F_in
  const int b = blockIdx.x, tid = threadIdx.x, nd = a.nd, F = a.F_in;
  const float* z = a.Z + (size_t)b * nd * 3;
  for (int col = tid; col < F; col += 256) {
    float val = 0.f;
    if (a.eq == 1) {  // SO2: [ip nd | G nd^2 | r | zy nd | dy]
      const int c1 = col - nd;
      if (c1 >= 0 && c1 < nd * nd) {
        const int i = c1 / nd, k = c1 - i * nd;
        val = z[i * 3] * z[k * 3] + z[i * 3 + 2] * z[k * 3 + 2];
      } else if (c1 > nd * nd && c1 <= nd * nd + nd) {
        val = z[(c1 - nd * nd - 1) * 3 + 1];
      }
    } else if (a.eq == 2) {  // SO3: [ip nd | G nd^2]
      const int c1 = col - nd;
      if (c1 >= 0) {
        const int i = c1 / nd, k = c1 - i * nd;
        val = z[i * 3] * z[k * 3] + z[i * 3 + 1] * z[k * 3 + 1] + z[i * 3 + 2] * z[k * 3 + 2];
      }
    } else {  // None: [ip nd | vec(Z) 3nd]
      const int c1 = col - nd;
      if (c1 >= 0) val = z[c1];
    }
    xc[col] = val;
    if (blockIdx.y == 0) a.xconst[(size_t)b * F + col] = val;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int h = blockIdx.y * 8 + wave; h < min(a.H, (int)blockIdx.y * 8 + 8); h += 4) {
    const float* w = a.W0 + (size_t)h * F;
    float c = 0.f, ux = 0.f, uy = 0.f, uz = 0.f;
    for (int col = lane; col < F; col += 64) {
      const float wv = w[col];
      c += wv * xc[col];
      if (col < nd) {
        ux += wv * z[col * 3];
        uy += wv * z[col * 3 + 1];
        uz += wv * z[col * 3 + 2];
      }
    }
    c = wave_sum(c); ux = wave_sum(ux); uy = wave_sum(uy); uz = wave_sum(uz);
    if (lane == 0) {
      float* o = a.A + ((size_t)b * a.H + h) * 8;
      if (a.eq == 1) {
        o[0] = ux; o[1] = w[F - 1]; o[2] = uz; o[3] = w[nd + nd * nd];
      } else {
        o[0] = ux; o[1] = uy; o[2] = uz; o[3] = 0.f;
      }
      o[4] = c + a.b0[h];
      o[5] = 0.f; o[6] = 0.f; o[7] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight repacking into MFMA fragment order (once per call; the weights change every step)
// ------------------------------------------------------------------------------------------
template <class Pol>
__global__ void __launch_bounds__(256) k_pack(const PackArgs a) {
  const PackDesc d = a.d[blockIdx.y];
  const int nfrag_lanes = d.nrb * d.nks * 64;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < nfrag_lanes; t += gridDim.x * 256) {
    const int lane = t & 63, fr = t >> 6;
    const int ks = fr % d.nks, rb = fr / d.nks;
    const int R = 32 * rb + (lane & 31), hi = lane >> 5;
    if constexpr (Pol::BF) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int Fk = kfeat<Pol>(ks, hi, e);
        float val = 0.f;
        if (!d.transposed) { if (R < d.M && Fk < d.K) val = a.params[d.src + (size_t)R * d.K + Fk]; }
        else { if (Fk < d.M && R < d.K) val = a.params[d.src + (size_t)Fk * d.K + R]; }
        o[e] = (__bf16)val;
      }
      *(bf16x8*)(a.wimg + d.dst + (size_t)t * 16) = o;
    } else {
      const int Fk = kfeat<Pol>(ks, hi, 0);
      float val = 0.f;
      if (!d.transposed) { if (R < d.M && Fk < d.K) val = a.params[d.src + (size_t)R * d.K + Fk]; }
      else { if (Fk < d.M && R < d.K) val = a.params[d.src + (size_t)Fk * d.K + R]; }
      *(float*)(a.wimg + d.dst + (size_t)t * 4) = val;
    }
  }
  if (d.bias_n_pad > 0 && blockIdx.x == 0) {
    float* bo = (float*)(a.wimg + d.dst + (size_t)nfrag_lanes * Pol::FRAGB);
    for (int i = threadIdx.x; i < d.bias_n_pad; i += 256) bo[i] = (i < d.bias_n) ? a.params[d.bias_src + i] : 0.f;
  }
}
#ifndef RENI_ONLY_TRAIN
template __global__ void k_pack<PolF32>(const PackArgs);
template __global__ void k_pack<PolBF16>(const PackArgs);

// ------------------------------------------------------------------------------------------
// epilogue kernels (all fixed-order sums: deterministic)
// ------------------------------------------------------------------------------------------
// dparams[i] = sum over workgroups of their partial, for i in [lo, n)
__global__ void __launch_bounds__(256) k_reduce_partials(const float* part, size_t per_wg, int nwg,
                                                          float* dparams, int lo, int n) {
  const int i = lo + blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // four interleaved partial sums (fixed order -> deterministic) keep four loads in flight
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int w = 0;
  for (; w + 3 < nwg; w += 4) {
    s0 += part[(size_t)w * per_wg + i];
    s1 += part[(size_t)(w + 1) * per_wg + i];
    s2 += part[(size_t)(w + 2) * per_wg + i];
    s3 += part[(size_t)(w + 3) * per_wg + i];
  }
  for (; w < nwg; ++w) s0 += part[(size_t)w * per_wg + i];
  dparams[i] = (s0 + s1) + (s2 + s3);
}

// Per-image reduction of the per-tile partials, two fixed-order stages.
// stage A: grid (B, nchunk): sum RT_CHUNK consecutive tiles -> part2[b][chunk][H*8 + 16]
//          (dA[h][k] = x_hi part + x_lo part for k < 5; 16 loss/statistics slots summed over tiles and waves)
// stage B: grid B: sum the chunks -> dA[b][h][8], img_loss[b][16]
constexpr int RT_CHUNK = 16;
__global__ void __launch_bounds__(256) k_reduce_tiles_a(const float* dA_part, const float* loss_part,
                                                        int tiles_per_image, int H, float* part2, int have_dA) {
  const int b = blockIdx.x, ch = blockIdx.y, nch = gridDim.y;
  const int t0 = ch * RT_CHUNK, t1 = min(t0 + RT_CHUNK, tiles_per_image);
  float* out = part2 + ((size_t)b * nch + ch) * (H * 8 + 16);
  if (have_dA) {
    for (int i = threadIdx.x; i < H * 8; i += 256) {
      const int h = i >> 3, k = i & 7;
      float s = 0.f;
      if (k < 5) {
        for (int t = t0; t < t1; ++t) {
          const float* q = dA_part + ((size_t)(b * tiles_per_image + t) * H + h) * 16;
          s += q[k] + ((k < 4) ? q[8 + k] : 0.f);
        }
      }
      out[i] = s;
    }
  }
  if (threadIdx.x < 16) {
    float s = 0.f;
    for (int t = t0 * 4; t < t1 * 4; ++t) s += loss_part[((size_t)b * tiles_per_image * 4 + t) * 16 + threadIdx.x];
    out[H * 8 + threadIdx.x] = s;
  }
}

__global__ void __launch_bounds__(256) k_reduce_tiles_b(const float* part2, int nch, int H, float* dA,
                                                        float* img_loss, int have_dA) {
  const int b = blockIdx.x;
  const float* in = part2 + (size_t)b * nch * (H * 8 + 16);
  if (have_dA) {
    for (int i = threadIdx.x; i < H * 8; i += 256) {
      float s = 0.f;
      for (int c = 0; c < nch; ++c) s += in[(size_t)c * (H * 8 + 16) + i];
      dA[(size_t)b * H * 8 + i] = s;
    }
  }
  if (threadIdx.x < 16) {
    float s = 0.f;
    for (int c = 0; c < nch; ++c) s += in[(size_t)c * (H * 8 + 16) + H * 8 + threadIdx.x];
    img_loss[(size_t)b * 16 + threadIdx.x] = s;
  }
}

// cosine-term coefficients per (image, channel) from the forward statistics
// stats[b][c] = cA, stats[b][4+c] = cB, stats[b][8+c] = cosine similarity * pixel-0 weight
__global__ void k_cos_coeffs(const float* img_loss, const float* weight, long long ws0, long long ws2,
                             float beta, int B, float* stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 3) return;
  const int b = i / 3, c = i - 3 * b;
  const float sot = img_loss[(size_t)b * 16 + 1 + c], soo = img_loss[(size_t)b * 16 + 4 + c], stt = img_loss[(size_t)b * 16 + 7 + c];
  const float no = sqrtf(soo), nt = sqrtf(stt);
  const float den = fmaxf(no * nt, 1e-20f);
  const float cs = sot / den;
  const float s0 = weight[(size_t)b * ws0 + (size_t)c * ws2];  // weight of pixel 0 (loss_functions.py:29)
  stats[(size_t)b * 16 + c] = -beta * s0 / (3.f * den);
  stats[(size_t)b * 16 + 4 + c] = -beta * s0 * cs / (3.f * soo);
  stats[(size_t)b * 16 + 8 + c] = cs * s0;
}

// loss_terms = (loss, mse, prior, cosine)
__global__ void k_finalize_loss(const float* img_loss, const float* stats, const float* Z, int B, int P,
                                int nz, int has_cos, int has_prior, float alpha, float beta, float* loss_terms) {
  __shared__ float red[256];
  float mse = 0.f, cosv = 0.f, prior = 0.f;
  if (threadIdx.x == 0) {
    for (int b = 0; b < B; ++b) mse += img_loss[(size_t)b * 16] / (3.f * (float)P);
    if (has_cos)
      for (int b = 0; b < B; ++b) {
        const float m = (stats[(size_t)b * 16 + 8] + stats[(size_t)b * 16 + 9] + stats[(size_t)b * 16 + 10]) / 3.f;
        cosv += 1.f - m;
      }
  }
  float zz = 0.f;
  if (has_prior)
    for (int i = threadIdx.x; i < nz; i += 256) zz += Z[i] * Z[i];
  red[threadIdx.x] = zz;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    prior = alpha * red[0];
    cosv *= beta;
    loss_terms[0] = mse + prior + cosv;
    loss_terms[1] = mse;
    loss_terms[2] = prior;
    loss_terms[3] = cosv;
  }
}

// per image: m[col] = W0[:,col]^T g_c (g_c = constant-column gradient dA[:,4]) and, for the inner-product
// block, u[col][xyz] = W_ip[:,col]^T dA[:,xyz].   grid (B, ceil(F/64)); 4 row groups x 64 columns per block
__global__ void __launch_bounds__(256) k_tail_m(const TailArgs a) {
  __shared__ float red[4][64][4];
  const int F = a.F_in, nd = a.nd, H = a.H, b = blockIdx.x;
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + c;
  float s = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
  if (col < F) {
    const float* dAb = a.dA + (size_t)b * H * 8;
    const int h0 = rg * (H / 4), h1 = h0 + H / 4;
    for (int h = h0; h < h1; ++h) {
      const float w = a.W0[(size_t)h * F + col];
      const float* q = dAb + h * 8;
      s += w * q[4];
      if (col < nd) { sx += w * q[0]; sy += w * q[1]; sz += w * q[2]; }
    }
  }
  red[rg][c][0] = s; red[rg][c][1] = sx; red[rg][c][2] = sy; red[rg][c][3] = sz;
  __syncthreads();
  if (rg == 0 && col < F) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = (red[0][c][k] + red[1][c][k]) + (red[2][c][k] + red[3][c][k]);
    a.mcol[(size_t)b * F + col] = t[0];
    if (col < nd) {
      float* u = a.ucol + ((size_t)b * nd + col) * 3;
      u[0] = t[1]; u[1] = t[2]; u[2] = t[3];
    }
  }
}

// assemble dZ from m / u.  grid B
__global__ void __launch_bounds__(256) k_tail_dz(const TailArgs a) {
  const int F = a.F_in, nd = a.nd, b = blockIdx.x, tid = threadIdx.x;
  const float* m = a.mcol + (size_t)b * F;
  const float* u = a.ucol + (size_t)b * nd * 3;
  const float* z = a.Z + (size_t)b * nd * 3;
  for (int t = tid; t < nd * 3; t += 256) {
    const int i = t / 3, k = t - 3 * i;
    float val = 0.f;
    if (a.eq == 1) {
      if (k == 1) {
        val = m[nd + nd * nd + 1 + i];
      } else {
        val = u[i * 3 + k];
        for (int q = 0; q < nd; ++q) val += (m[nd + i * nd + q] + m[nd + q * nd + i]) * z[q * 3 + k];
      }
    } else if (a.eq == 2) {
      val = u[i * 3 + k];
      for (int q = 0; q < nd; ++q) val += (m[nd + i * nd + q] + m[nd + q * nd + i]) * z[q * 3 + k];
    } else {
      val = u[i * 3 + k] + m[nd + i * 3 + k];
    }
    val += 2.f * a.alpha2 * z[t];  // prior term d(alpha |Z|^2) (0 unless RENITestLoss)
    a.dZ[(size_t)b * nd * 3 + t] = val;
  }
}

// dW0[h][col] and db0[h]: sums over the batch of per-image outer products
__global__ void __launch_bounds__(256) k_tail_dw0(const TailArgs a) {
  const int F = a.F_in, nd = a.nd, H = a.H, B = a.B;
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int h = blockIdx.y;
  if (col > F) return;
  if (col == F) {  // bias
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += a.dA[((size_t)b * H + h) * 8 + 4];
    a.db0[h] = s;
    return;
  }
  float s = 0.f;
  int kind = 0;  // 0: constant column, 1: inner-product column, 2: r, 3: dy
  if (col < nd) kind = 1;
  else if (a.eq == 1 && col == nd + nd * nd) kind = 2;
  else if (a.eq == 1 && col == F - 1) kind = 3;
  for (int b = 0; b < B; ++b) {
    const float* q = a.dA + ((size_t)b * H + h) * 8;
    if (kind == 0) s += q[4] * a.xconst[(size_t)b * F + col];
    else if (kind == 1) {
      const float* z = a.Z + ((size_t)b * nd + col) * 3;
      if (a.eq == 1) s += q[0] * z[0] + q[2] * z[2];
      else s += q[0] * z[0] + q[1] * z[1] + q[2] * z[2];
    } else if (kind == 2) s += q[3];
    else s += q[1];
  }
  a.dW0[(size_t)h * F + col] = s;
}

__global__ void __launch_bounds__(256) k_adam(float* p, const float* g, float* m, float* v, long long n,
                                              float lr, float b1, float b2, float eps, float bc1,
                                              float bc2_sqrt, float gscale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  // torch.optim.Adam (non-amsgrad): denom = sqrt(v)/sqrt(bias_correction2) + eps; p -= lr/bias_correction1 * m/denom
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}

// ------------------------------------------------------------------------------------------
// MFMA layout probes (reni_selftest_layouts)
// ------------------------------------------------------------------------------------------
// probe 0: fp32 32x32x2 : D = A(32x2) B(2x32) with the operand/result maps the kernels assume
// probe 1: bf16 32x32x16
__global__ void k_probe(float* outD, int which) {
  const int lane = threadIdx.x, hi = lane >> 5, j = lane & 31;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (which == 0) {
    // A[i][k] = 1 + i + 100 k ; B[k][n] = 3 + 2 n + 7 k
    const float av = 1.f + j + 100.f * hi;
    const float bv = 3.f + 2.f * j + 7.f * hi;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  } else {
    // A[i][k] = ((i + 3k) % 7) - 3 ; B[k][n] = ((2n + k) % 5) - 2 ; k = 8*hi + e  (exact in bf16)
    bf16x8 av, bv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * hi + e;
      av[e] = (__bf16)(float)(((j + 3 * k) % 7) - 3);
      bv[e] = (__bf16)(float)(((2 * j + k) % 5) - 2);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) outD[rowmap(r, hi) * 32 + j] = acc[r];
}

// probe of ds_read_b64_tr_b16: LDS holds u16 element i = i; lane l reads at byte address 8*l (mode 0) or
// at a caller-given per-lane byte address (mode 1); out[l*4 + e] = element e returned to lane l
__global__ void k_probe_tr(unsigned short* out, const int* lane_addr, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)lds + (mode == 0 ? threadIdx.x * 8 : lane_addr[threadIdx.x]);
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (unsigned short)(v >> (16 * e));
}

// explicit instantiations of the fused kernel
#define RENI_INST(POL, HH)                                                     \
  template __global__ void k_reni_main<POL, HH, MODE_FWD>(const MainArgs);     \
  template __global__ void k_reni_main<POL, HH, MODE_STATS>(const MainArgs);   \
  template __global__ void k_reni_main<POL, HH, MODE_FWD_BWD>(const MainArgs);
RENI_INST(PolF32, 32)
RENI_INST(PolF32, 64)
RENI_INST(PolF32, 128)
RENI_INST(PolBF16, 32)
RENI_INST(PolBF16, 64)
RENI_INST(PolBF16, 128)
RENI_INST(PolF32, 256)
RENI_INST(PolBF16, 256)
#endif  // RENI_ONLY_TRAIN

}  // namespace reni

#ifndef RENI_ONLY_TRAIN
#include "reni_capi.inc"
#endif
