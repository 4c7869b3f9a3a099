// Level 2 of the NAT-FPN history encoder (dim 128, 8 heads of 16, kernel 5 over L = 5 steps: every step sees all five; embedding.py:93-99,
// 196-202), wave-private and weight-streaming like dec_w.h: a wave owns 3 agents x 5 steps (rows 5 a + t of one 16-row tile) for both
// NATLayers with the residual in the MFMA C/D register layout; the two layers' weights (2 x 320 KiB) stream through a three-slot LDS ring
// as 20 groups of 32 operand fragments per round of 8 tiles (LDS-DMA, two groups ahead); the 5 x 5 neighbourhood attention of the three
// agents of a tile is one 16 x 16 MFMA score tile per head (other agents' keys masked, rpb as the score accumulator).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opfmt.h"
#include "dropstats.h"

namespace RIFT_NS {

#define L2W_GROUPS 10                           // per NATLayer: q | k | v | proj | (fc1 chunk c, fc2 k-range c) x 3
#define L2W_BLK_FRAGS (L2W_GROUPS * 32)
// fp32 parameters in LDS: per block 1664: ln1 g,b 256 | bqkv 384 (q pre-scaled) | rpb 8 x 9 (x log2 e) padded to 128 | bproj 128 | ln2 g,b 256 | b1 384 | b2 128
#define L2W_P_BLK(b) (1664 * (b))
#define L2W_PB_LN1 0
#define L2W_PB_BQKV 256
#define L2W_PB_RPB 640
#define L2W_PB_BP 768
#define L2W_PB_LN2 896
#define L2W_PB_B1 1152
#define L2W_PB_B2 1536
#define L2W_P_FN 3328                           // norm2 of the encoder (the FPN's level-2 input): g 128 | b 128
#define L2W_NPAR 3584
#define L2W_SLOTS 3
#define L2W_LDS (L2W_SLOTS * 32768 + L2W_NPAR * 4)

struct NatL2WSrc {
  struct Blk { const float *ln1_g, *ln1_b, *wqkv, *bqkv, *rpb, *wproj, *bproj, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2; } blk[2];
  const float* fn_g; const float* fn_b;
};

struct NatL2WP {
  const float* X; int nseq;                // (nseq * 5, 128) level input (level 1's downsample output)
  const unsigned short* img; const float* par;
  float* Oc;                               // (nseq * 3, 128) LayerNorm(norm2) of steps 2..4
  unsigned short* Ocb;                     // if set: the same rows as bf16 instead (what fpn_tail_kernel rounds them to anyway: half the bytes both ways)
  float droppath[2]; uint32_t seed, stream;
  long long* ts;                           // optional: clock of wave 0 of workgroup 0 at every group boundary (diagnostic)
  const int* cnt;                          // if set: the three class counts of the compacted launch (nat_l0w.h; common.h: SeqCount); nseq is the bound
  DropStats ds;                            // diagnostic build only (dropstats.h)
};

int l2w_set_attributes();
void l2w_pack(const NatL2WSrc& src, unsigned short* img, float* par, hipStream_t stream);
void l2w_launch(const NatL2WP& p, int grid, hipStream_t stream);

}  // namespace RIFT_NS
