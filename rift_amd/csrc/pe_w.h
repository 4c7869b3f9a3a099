// PointsEncoder pass B, wave-private and weight-streaming (round 2).  Same arithmetic as pe_mid_kernel (pe_fused.h; embedding.py:254-296:
// first_mlp -> max over the points of a polyline -> second_mlp.0 with its BatchNorm-2 statistics), rebuilt like the decoder / NAT kernels:
//   * a ROUND is 240 point rows (12 map polygons of 20 points, or 2 reference lines of 120); a wave owns two 16-row tiles of it through
//     the whole pass with every activation in registers: x -> h1 = relu(bn1(x W1^T + b1)) -> f = h1 W2^T + b2 -> g = f W3a^T + gp, each
//     GEMM output in the MFMA C/D layout chaining into the next GEMM's operand through K-permuted weight images (nat_l0w.h);
//   * LDS holds the WEIGHT STREAM: W1 | W2 | W3a as 7 groups of <= 32 one-KiB fragments per round, LDS-DMA'd into a two-slot ring one group
//     ahead, one barrier per group; a fragment read feeds the MFMAs of both tiles of the wave (pe_w_gemm.h);
//   * the max over a polyline's points crosses lanes (rows = lanes in the C/D layout): each wave transposes its bf16 f tile through a
//     private 4 KiB LDS scratch (rows in-lane -> 4 VALU per row and channel pair), writes per-(tile, segment) partial maxima, and the
//     workgroup combines them into the 16 x 256 pooled operand; gp = pooled W3b^T + b3 takes its 16 fragments per wave straight from L2
//     into registers (streaming W3b's 128 KiB through the ring for 16 MFMAs per wave would be 40 % of the stream);
//   * g goes to HBM as fp16 in 32-byte pieces per lane (W3a's image permutes the output channels so that a lane's 4 n-tiles x 4 values
//     are 16 consecutive channels); BatchNorm-2 statistics from the fp32 accumulators, one partial per workgroup;
//   * persistent workgroups (one per CU) walk the rounds with the stream running across round boundaries; in train mode only the rounds
//     that hold a valid point (pass A's counts -> the live lists of PeLiveP, pe_fused.h), dealt evenly: list positions wg, wg + G, ...
// Replaces: pe_mid_kernel (kept for RIFT_PE_W=0 and as the parity reference of tests/test_gpu_parity.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opfmt.h"

namespace RIFT_NS {

#define PEW_FRAGS 200                     // per encoder: W1 8 | W2 2 x 32 | W3a 4 x 32 fragments, consumption order
#define PEW_ROUND_ROWS 240
#define PEW_LDS_BYTES 128576

struct PeWSrc { const float *w1, *w2, *w3; int Cin; };   // first_mlp.0 (128, Cin), first_mlp.3 (256, 128), second_mlp.0 (256, 512: [feat | pool])

struct PeWSide {
  const float* F; int Cin;                // (rows, Cin) fp32 point features
  const uint8_t* valid;                   // (rows)
  int rows, npts, nrounds;                // npts = 20 | 120 points per polyline; nrounds = ceil(rows / 240)
  const unsigned short* img;              // pack_pew_kernel: PEW_FRAGS fragments
  const unsigned short* w3b;              // second_mlp.0 pool half, fragment-major bf16 image [256][256] (engine pack_cols)
  const float *b1, *b2, *b3, *s1, *t1;    // biases; BatchNorm-1 folded to y = x s1 + t1
  const int* live;                        // the rounds that hold a valid point, ascending (pe_fused.h: PeLiveP), or null: all nrounds rounds
  const int* ptab; const int* phdr;       // (side b) the packed rounds of pe_pack_lines_kernel (pe_fused.h): records and their number; null: unpacked
  float* part2;                           // [2][256][nwg] sums of g, g^2 over the valid rows of a workgroup's rounds (nwg = its share of the grid)
  int* cnt2;                              // [nwg] valid rows of a workgroup's rounds
  unsigned short* Fmid;                   // (rows, 256) fp16 bits of g
};

struct PeWP {
  PeWSide a, b;                           // a: map polygons (20 points), b: reference lines (120 points)
  int ga;                                 // workgroups [0, ga) take a's rounds, the rest b's
  const int* hdr;                         // if set (with a.live / b.live): n_live a, n_live b, ga, grid - ga from device memory (PeLiveP) instead
  int do_stats;
  int dbg;                                // diagnostics (timing only, results invalid): 1 no g stores, 2 no weight stream, 4 no statistics write-out (RIFT_PEW_DBG)
  long long* ts;                          // optional: clock of wave 0 of workgroup 0 at the group boundaries of its first round
};

int pew_set_attributes();
void pew_pack(const PeWSrc& src, unsigned short* img, hipStream_t stream);
void pew_split(int ra, int rb, int* grid, int* ga);        // grid = persistent workgroups (<= *grid, one per CU): [0, ga) for a's rounds, the rest for b's
void pew_launch(const PeWP& p, int grid, hipStream_t stream);

}  // namespace RIFT_NS
