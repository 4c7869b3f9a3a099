// Scene encoder (the 4 pre-LN encoder blocks + final LayerNorm, transformer.py:73-94 / pluto_model.py:144-150) for the dense-traffic
// shapes (97..192 tokens per scene), in the wave-private, weight-streaming form of dec_w.h: a wave owns one 16-token tile per round of
// eight tiles with the residual in the MFMA C/D register layout; a layer is two passes over the scene's tiles --
//   pass 1 (per tile): LayerNorm -> K and V projections -> the tile's K | V^T MFMA operand fragments into a per-scene scratch image
//                      (the per-head fragment order of dec_kv.h), two weight groups;
//   (last layer, optional tail per tile: the decoder's cross-attention K | V projections of its four layers, eight more groups)
//   pass 2 (per tile): LayerNorm -> Q -> attention over ALL tokens with the scratch image streamed back as four per-head groups (the
//                      decoder's cross-attention code path; key-padding mask as the score accumulator) -> out_proj -> DropPath residual ->
//                      LayerNorm -> fc1 / GELU / fc2 in four chunks -> DropPath residual, fourteen groups;
// the residual rows live in the output array between passes and rounds (workgroup-scope fences around the group barriers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opfmt.h"
#include "dropstats.h"

namespace RIFT_NS {

#define ENCW_WGROUPS 12                          // weight groups per layer: k | v | q | out | (fc1 chunk c, fc2 chunk c) x 4
#define ENCW_LAYER_FRAGS (ENCW_WGROUPS * 32)
// fp32 parameters of a layer (resident in LDS for all four): ln1 g,b 256 | bq (pre-scaled) 128 | bk 128 | bv 128 | bo 128 | ln2 g,b 256 | b1 512 | b2 128
#define ENCW_P_LN1 0
#define ENCW_P_BQ 256
#define ENCW_P_BK 384
#define ENCW_P_BV 512
#define ENCW_P_BO 640
#define ENCW_P_LN2 768
#define ENCW_P_B1 1024
#define ENCW_P_B2 1536
#define ENCW_P_LAYER 1664
#define ENCW_P_FN (4 * ENCW_P_LAYER)             // final norm g 128 | b 128
#define ENCW_P_BKV (ENCW_P_FN + 256)               // the decoder's cross-attention K | V biases: [4 layers][k 128 | v 128]
#define ENCW_NPAR (ENCW_P_BKV + 1024)
#define ENCW_TAIL_FRAGS (8 * 32)                  // image tail: the decoder's K | V projection weights, per decoder layer (k | v)
#define ENCW_LDS (2 * 32768 + ENCW_NPAR * 4 + 192 * 4)

struct EncWSrc {
  struct L { const float *ln1_g, *ln1_b, *w_in, *b_in, *wo, *bo, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2; } l[4];
  const float* fn_g; const float* fn_b;
  const float* dkv_w[4]; const float* dkv_b[4];   // planning_decoder.decoder_blocks.l.cross_attn.in_proj_weight / bias (rows 128..383 = k | v)
};

struct EncWP {
  const float* X;               // (bs*N, 128) tokens + positional embedding
  float* Y;                     // (bs*N, 128) encoder output after the final LayerNorm; also the working residual between passes
  const uint8_t* kpm;           // (bs*N) key padding mask (1 = padded)
  int bs, N;                    // 96 < N <= 192 (works for any N <= 192)
  unsigned short* KVs;          // (bs, 96, 512) bf16 scratch: the current layer's K | V^T fragments of every scene
  unsigned short* DKV;          // optional (bs, 4, 96, 512): the planning decoder's cross-attention K | V^T fragments of its four layers (dec_w.h,
                                // dense layout), projected from the final output while it is still in registers (planning_decoder.py:74-79)
  const unsigned short* img; const float* par;
  float droppath[4]; uint32_t seed, stream;
  int* nonfinite;               // device flag: raised when a valid token row leaves the encoder with a NaN / Inf
  DropStats ds;                 // diagnostic build only (dropstats.h)
};

int encw_set_attributes();
void encw_pack(const EncWSrc& src, unsigned short* img, float* par, hipStream_t stream);
void encw_launch(const EncWP& p, hipStream_t stream);

}  // namespace RIFT_NS
