// reni_internal.h -- kernel argument blocks shared by reni_device.inc (the reni_tu_*.hip translation units) and reni_capi.inc
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace reni {

constexpr int MAX_LAYERS = 15;  // hidden_layers <= MAX_LAYERS - 1

struct MainArgs {
  // geometry
  int B, P, tiles_per_image, n_tiles;
  int L, last_linear, act, loss_kind, need_dw;
  int dbg;  // ablation mask for profiling experiments (RENI_DEBUG_MASK); 0 in production
  float w_first, w_hidden, beta;
  // Concat calls on the persistent bf16 kernels (round 6): the W^T images are EXACT power-of-two
  // multiples (x 8) of the forward images' rounded weights -- the backward pass is the transpose of the very network the forward pass
  // evaluates -- and the constant that is then missing from the gradient, (2 pi)^L omega_first / 8^(L+1), rides on d loss / d y (the
  // weight gradients' per-layer constants are multiplied back where their partials are summed: reni::LayerScale).  1 everywhere else.
  float gy_scale;
  // inputs
  const float* Z;
  const float* D;
  long long d_bstride;
  const float* Apre;  // [B][H][8] per-image affine map of the first layer
  const char* afrag;  // the same as split-bf16 MFMA fragments (PrepArgs::afrag), persistent training path
  const char* wimg;   // packed weight images
  unsigned fwd_off[MAX_LAYERS + 2];  // byte offsets into wimg, index 1..L hidden, L+1 head
  unsigned bwd_off[MAX_LAYERS + 2];
  const float* target;
  long long ts0, ts1, ts2;
  const float* weight;
  long long ws0, ws1, ws2;
  const float* dout;
  const float* stats;  // [B][16] cosine-term coefficients
  // outputs / scratch
  float* out;
  char* stash;
  size_t stash_per_wg;
  float* dwp;  // per-workgroup decoder-gradient partials, flat parameter layout
  size_t dwp_per_wg;
  unsigned p_off_w[MAX_LAYERS + 2];  // float offsets of W_l / b_l in the flat layout
  unsigned p_off_b[MAX_LAYERS + 2];
  float* dA_part;    // [n_tiles][H][8]: columns dx, dy, dz, r, 1 (hi + lo parts summed), 3 unused
  float* loss_part;  // [n_tiles][4][16]
  char* g1;          // [n_tiles][H/16][256][16 B] bf16 g_1 stream (persistent training path)
  float* dw1p;       // [2 nwg][H * H + H] k_reni_dw1's weight-gradient partials (layer 1)
  long long* trace;  // optional (tag, s_memtime) pairs from workgroup 0 (RENI_TRACE builds)
  // FiLM conditioning (k_reni_main<..., FILM = true>): hidden layer l in 1..L of image b applies
  // sin(freq . (W_l h + b_l) + phase); film[b][l-1][0][.] = freq, [1][.] = phase
  const float* film;
  float* dfp_part;      // [n_tiles][L][2][H] per-tile d(freq), d(phase)
  const float* params;  // flat fp32 parameters (d(freq) needs W_l and b_l)
  unsigned n_first;     // p_off_* are relative to the partial buffer: params offset = p_off + n_first
  // wide bf16 training (H = 256): a layer's 256 x 256 weight gradient does not fit a workgroup's registers next to
  // the chain, so k_reni_main writes the transposed operand images of every (layer, tile, 64-sample round) to this
  // stream and k_dw_stream256 finishes dW_l with one layer's accumulator resident in its AGPRs.
  // [L][n_tiles][2 rounds][2 images (g_l | h_{l-1})][256 features][T_ROWB bytes]
  char* dws;
  // H = 256: instead of transposed operand images the chain stores what it has in registers -- the
  // gradient g_l as bf16 fragments, gfs[(l-1) n_tiles + tile][H/32][2][256 threads][16 B] -- and keeps the phase stash of
  // EVERY tile (stash + tile * stash_per_wg, same fragment layout: h_{l-1} = sin(2 pi phase)); k_dw_frag builds the two
  // operand images of a record in its own LDS.  NULL: the image stream above.
  char* gfs;
  // FiLM on the stream path: a workgroup's contiguous record range is cut into per-image runs; run k of workgroup w
  // writes dfr[((w * kmax + k) * L + l-1)][2][H] = (sum_k W (.) M rows, d(phase)) and run_image[w * kmax + k] = image
  float* dfr;
  int* run_image;
  int kmax;
  // persistent TRAINING kernel (k_reni_train_bf16<H, true>): a workgroup owns a CONTIGUOUS range of tiles and keeps the
  // per-image sums (layer-0 dA [H][8], loss) on chip; they leave the chip once per image run: run k of workgroup w is
  // pruns[(w * pkmax + k)][H * 8 + 16] (dA, then one loss slot per wave) with prun_image[w * pkmax + k] = image (-1: unused)
  float* pruns;
  int* prun_image;
  int pkmax;
  int nwg_main;  // grid of the training kernel (k_reni_dw1 walks each workgroup's range backwards: most recent first)
  // RENI_WEIGHT_SPARSE (frozen-decoder and statistics instances of the persistent kernel): the tiles to visit, ascending, and their
  // number -- built on the device from the loss weight by k_tile_flags / k_tile_compact.  NULL: every tile.
  const int* tlist;
  const int* tcount;
  // RENI_WEIGHT_COMPACT: the image's pixels with a non-zero weight, ascending, packed into its first tiles: position k of image b is
  // pixel plist[b * P + k], k < nlive[b] (k_pix_scatter).  NULL: position = pixel.
  const int* plist;
  const int* nlive;
};

// sets reni_last_error()'s thread-local message and returns `code` (defined next to the C ABI, reni_capi.inc)
int reni_set_error(int code, const char* msg);

// host launchers of the fused kernel, one per translation unit (reni_device.inc)
hipError_t launch_main_f32(int H, int mode, const MainArgs& a, int nwg, hipStream_t s);
hipError_t launch_main_bf16(int H, int mode, const MainArgs& a, int nwg, hipStream_t s);
hipError_t launch_film_f32(int H, int mode, const MainArgs& a, int nwg, hipStream_t s);
hipError_t launch_film_bf16(int H, int mode, const MainArgs& a, int nwg, hipStream_t s);
// FiLM instances of the persistent kernels (reni_tu_train_film.hip): which = 0 training instance, 1 forward / statistics
hipError_t launch_train_film(int which, const MainArgs& a, int grid, hipStream_t s);
hipError_t launch_dw1_film(const MainArgs& a, int grid, hipStream_t s);
int train_film_lds_bytes(int which);
// the H = 256 persistent chain (reni_tu_wide.hip): which = 0 forward / statistics, 1 frozen-decoder forward + loss + backward
hipError_t launch_wide256(int which, const MainArgs& a, int grid, hipStream_t s);  // (2: the training form, with launch_wide_head_dw behind it; 3: FiLM forward / statistics; 4: FiLM training form)
int wide256_film_max_layers();  // hidden FiLM layers whose per-image tables fit the forward instance's LDS region
int wide256_film_train_max_layers();  // ... and the training form's (4: the default model's; its tables alias the dA exchange alone)
hipError_t launch_wide_head_dw(const MainArgs& a, int grid, hipStream_t s);

struct PrepArgs {
  const long long* idx;  // optional: image b's latent is row idx[b] of Z (a latent TABLE); the batch's rows are copied to Zc
  long long n_rows;      // rows of that table: an index outside [0, n_rows) poisons the image's results with NaN
  float* Zc;             // [B][nd][3] compact copy of the gathered rows (read by the epilogue kernels), with idx only
  long long* idx_copy;   // [B] optional: idx as this prologue saw it (reni_train_step_rows checks a staged batch against it)
  const float* Z;
  const float* W0;
  const float* b0;
  float* xconst;  // [B][F_in]
  float* A;       // [B][H][8]
  char* afrag;    // optional [B][H/32][64 lanes][16 B]: afrag_scale * A_b as split-bf16 MFMA A operands (persistent kernels)
  float afrag_scale;  // omega_first / 2 pi: the persistent kernels' layer-0 MFMA then yields the sine argument in revolutions
  int eq, nd, F_in, H;
};

struct PackDesc {
  unsigned src;       // float offset of the weight matrix in params
  unsigned bias_src;  // float offset of its bias
  unsigned dst;       // byte offset in wimg
  int M, K;           // logical matrix [M][K] row-major
  int transposed;     // 0: image rows = M rows, k = K cols ; 1: image rows = K cols, k = M rows
  int nrb, nks;       // image geometry
  int bias_n, bias_n_pad;
  float scale;        // every weight and bias of the image is multiplied by this (1 = plain copy)
};

struct PackArgs {
  const float* params;
  char* wimg;
  PackDesc d[2 * (MAX_LAYERS + 1)];
};

struct TailArgs {
  const float* dA;  // [B][H][8]
  const float* W0;
  const float* Z;
  const float* xconst;
  float* mcol;  // [B][F_in] scratch: W0^T g_c
  float* ucol;  // [B][ND][3] scratch: W_ip^T dA[:,xyz]
  float* dZ;
  float* dW0;
  float* db0;
  float alpha2;
  int eq, nd, F_in, H, B;
};

// FiLM per-image glue (k_film_minput / k_film_linear / k_film_fold; k_film_dout / k_film_linear_t / k_film_grads): the mapping network
// (src/models/RENI.py:482-505), freq = 15 f + 30, and the first FiLM layer folded into the per-image affine map
constexpr int MAX_MAP_LAYERS = 8;
struct FilmGlueArgs {
  const float* Z;        // [B][nd][3]
  const float* params;   // flat net.* / final_layer.* (net.0 = W0 [H][F0], b0 [H] first)
  const float* mparams;  // flat mapping_network.network.{0,2,..}.{weight,bias}
  float* A;              // [B][H][8]   out (forward)
  float* film;           // [B][L][2][H] out (forward)
  float* mapm;           // [B][M_in]      saved mapping input
  float* mapx;           // [B][ML][Hm]    saved post-activation inputs of mapping layers 1..ML
  float* fo;             // [B][N_out]     saved mapping output
  const float* dA;       // [B][H][8]   in (backward)
  const float* dfilm;    // [B][L][2][H] in (backward)
  float* dAb;            // [B][H][8]  freq_0 . dA: gradient w.r.t. the un-modulated first-layer map
  float* dly;            // deltas of the mapping layers' outputs, block i = [B][N_i] at dly_off[i]
  float* dZ;             // [B][nd][3] out (backward)
  float alpha2;          // 2 alpha of the latent prior alpha |Z|^2 (RENITestLoss), 0 otherwise
  int eq, nd, H, L, F0, M_in, Hm, ML, N_out, B;
  unsigned mw_off[MAX_MAP_LAYERS + 1], mb_off[MAX_MAP_LAYERS + 1];
  unsigned dly_off[MAX_MAP_LAYERS + 1];
};

}  // namespace reni
