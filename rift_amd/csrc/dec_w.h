// Planning decoder, wave-private form (round 2): the 4 DecoderLayers (planning_decoder.py:42-86) of ONE scene in ONE workgroup, with
// every activation register-resident.  A wave owns one 16-row query tile for a whole sub-block; the residual stream lives in the MFMA
// C/D register layout (lane = 4 consecutive channels of row lane&15, 8 n-tiles = 32 VGPRs) and every GEMM output chains into the next
// GEMM's operand through K-permuted weight images (nat_l0w.h), so LayerNorm -> qkv -> attention -> out_proj -> residual -> FFN never
// touch LDS.  Two tilings of the R x 12 queries are used:
//   * r2r (attention over the R reference lines of one mode, :56-60): tile = a mode PAIR, slot l15 = (mode & 1) * 8 + r;
//   * m2m / cross / FFN (:62-83): tile = one reference line, slot l15 = mode (12 of 16 used);
// the fp32 residual changes tiling twice per layer through a 96-row LDS buffer.  What LDS mostly holds is the WEIGHT STREAM: all
// waves consume the same 1 KiB operand fragments in the same order, so the layer is a sequence of 20 groups of <= 32 fragments
// (18 weight groups + the scene's cross-attention K | V^T operands in two halves, written fragment-major by the encoder kernel's
// tail) that LDS-DMA (`global_load_lds_dwordx4`, no staging registers) drops into a two-slot ring one group ahead; one barrier per
// group both publishes the next slot and retires the previous one, and the residual hand-overs ride on those barriers.  20 barriers
// per layer instead of ~28 barrier-separated LDS round trips, no activation traffic through LDS at all.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opfmt.h"
#include "dropstats.h"

namespace RIFT_NS {

#define DECW_WGROUPS 18                         // weight groups per layer, 32 fragments each (consumption order, see the kernel)
#define DECW_LAYER_FRAGS (DECW_WGROUPS * 32)
#define DECW_KV_FRAGS 48                        // per (scene, layer): heads {0,1}: 12 K + 12 V^T fragments | heads {2,3}: likewise
// fp32 parameter block of a layer: region E (r2r + m2m) and region L (cross attention + FFN), resident in LDS on their own schedules
#define DECW_E_LN1 0                            // g 128 | b 128
#define DECW_E_BR2R 256                         // q (pre-scaled by head_dim^-0.5 log2 e) 128 | k 128 | v 128
#define DECW_E_BR2RO 640
#define DECW_E_LN2 768
#define DECW_E_PB 1024                          // [12][DECW_PBS]: mode m -> (m_pos W_qk^T + b_qk)[q pre-scaled 128 | k 128]  (:62-64)
#define DECW_PBS 260
#define DECW_E_BM2MV (1024 + 12 * DECW_PBS)
#define DECW_E_BM2MO (DECW_E_BM2MV + 128)
#define DECW_E_N 4608                           // 18 KiB
#define DECW_L_LN3 0
#define DECW_L_BCQ 256                          // pre-scaled
#define DECW_L_BCO 384
#define DECW_L_LN4 512
#define DECW_L_BF1 768
#define DECW_L_BF2 1280
#define DECW_L_N 1536                           // 6 KiB
#define DECW_PAR_LAYER (DECW_E_N + DECW_L_N)
#define DECW_XS 132
#define DECW_LDS_BYTES (2 * 32768 + 96 * DECW_XS * 4 + DECW_PAR_LAYER * 4 + 96 * 4 + 96 * 4 + 16)
#define DECW_LDS_MID_BYTES (DECW_LDS_BYTES + 32 * 4)                   // eight key tiles: 128 mask entries
#define DECW_LDS_DENSE_BYTES (2 * 32768 + 2 * DECW_PAR_LAYER * 4 + 192 * 4 + 12 * 16 * 4 + 16)
#define DECW_ROWS 96                            // dropout counter stride per scene

struct DecWSrc {
  struct L {
    const float* ln[8];                         // norm1.weight, norm1.bias, ..., norm4.bias
    const float *r2r_w, *r2r_b, *r2ro_w, *r2ro_b, *m2m_w, *m2m_b, *m2mo_w, *m2mo_b;   // in_proj (384,128) / out_proj (128,128)
    const float *c_w, *c_b, *co_w, *co_b;       // cross_attn in_proj (rows 0..127 = q) / out_proj
    const float *f1_w, *f1_b, *f2_w, *f2_b;     // ffn.0 (512,128), ffn.3 (128,512)
  } l[4];
  const float* m_pos;                           // (12, 128)
};

struct DecWP {
  float* Q;                     // (bs*R*12, 128) fp32 decoder queries, updated in place
  const uint8_t* kpm;           // (bs*N) encoder key padding
  int compact;                  // 1: the K | V^T fragments come from an encoder that compacted its rows (enc_fused.h, RIFT_ENC_COMPACT): kpm is that kernel's
                                // kpm_c (bs, 96) -- a scene's valid keys are a PREFIX, so the sixth key tile exists exactly when key 80 is valid
  const uint8_t* r_kpm;         // (bs*R) reference-line padding
  const uint8_t* q_kpm;         // (q_bs*R) padding rows the r2r quirk indexes (see DecFusedP)
  int q_bs, q_off;
  int bs, N, R;                 // N <= 96 tokens and R <= 8 reference lines: the standard kernel; up to N <= 192, R <= 16: the dense-traffic variant
  const unsigned short* KV;     // (bs, 4, DECW_KV_FRAGS, 512) bf16 fragment images from the encoder kernel's tail; dense variant: (bs, 4, 96, 512)
                                // from dec_kv_frag_kernel (per head: 12 K fragments, then V^T (dim tile, key pair))
  const unsigned short* img;    // pack_decw_kernel
  const float* par;
  float dropout; uint32_t seed, stream;
  int dbg;                      // diagnostic: 1 = skip all compute (weight stream + barriers only), 8 = no stream, 16 = all eight waves carry the stream
  long long* ts;                // optional: clock of wave 0 of workgroup 0 at every group boundary (diagnostic, RIFT_DEC_TS)
  int* nonfinite;               // device flag: raised when a query row leaves the last layer with a NaN / Inf (planning_decoder.py:175)
  int l0, l1;                   // layers [l0, l1) of the four (l0 even): a launch per half lets a small-batch step run the halves on two queues
  uint32_t* rng_io;             // (bs, 512, 4) dropout stream states: written by a launch that stops before layer 4, read by one that starts behind layer 0
                                // (the draws of the two halves are then the ones of a single launch)
  DropStats ds;                 // diagnostic build only (dropstats.h)
};

// batches the standard kernel serves with EIGHT key tiles out of the dense K | V^T image (dec_kv.h): R <= 8 lines, 96 < N <= 128 token slots
inline bool decw_mid_shape(int R, int N) { return R <= 8 && N > 96 && N <= 128; }
// host entry points of the dec_w.hip translation unit (see build.py)
int decw_set_attributes();
void decw_pack(const DecWSrc& src, unsigned short* img, float* par, hipStream_t stream);
void decw_launch(const DecWP& p, hipStream_t stream);

}  // namespace RIFT_NS
