// FourierEmbedding (fourier_embedding.py:45-55), wave-private and weight-streaming (round 2); same arithmetic as fourier_fused_kernel:
//   for each input dim d: [cos(2 pi f x_d) (64), sin (64), x_d] -> Linear(129,128) -> LayerNorm -> ReLU -> Linear(128,128); sum over d
//   -> LayerNorm -> ReLU -> Linear(128,128).
// Every step is row-wise, so a wave owns one 16-row tile from the inputs to the output with all activations in registers (MFMA C/D layout,
// GEMM outputs chained as the next operand through K-permuted weight images, LayerNorm across the four lanes of a row by permlane swaps);
// LDS holds the weight stream: 2 D + 1 groups of 32 one-KiB fragments per pass of eight tiles, LDS-DMA into a two-slot ring, one barrier
// per group.  fourier_fused_kernel gave each 64-row workgroup the same 224 KiB of weights (as register fragments, behind ~20 barrier
// separated LDS phases): 3.5 KiB of L2 traffic per row against 1.75 here, and the benchmark's 2160 tiles fit ONE pass of 238 workgroups
// (the one-dimensional speed-limit embedding runs two passes per workgroup: a pass there is 3 groups instead of 7).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opfmt.h"

namespace RIFT_NS {

#define FOW_PAR_DIM 640                          // per dim: b0 | ln gamma | ln beta | w0[:, 128] | b3
#define FOW_PAR_OUT 1920                         // to_out: ln gamma | ln beta | bias
#define FOW_PAR_FREQ 2304                        // [3][64]
#define FOW_PAR_B3SUM 2496                       // sum over the dims of b3
#define FOW_NPAR 2624
#define FOW_LDS_BYTES (2 * 32768 + FOW_NPAR * 4)

struct FoWSrc {                                  // fp32 parameters of one FourierEmbedding
  int D;
  const float* w0[3]; const float* b0[3];        // mlps.d.0 (128, 129)
  const float* lng[3]; const float* lnb[3];      // mlps.d.1
  const float* w3[3]; const float* b3[3];        // mlps.d.3 (128, 128)
  const float *og, *ob, *wo, *bo;                // to_out.0, to_out.2
  const float* freqs;                            // (D, 64)
};

struct FoWSide {
  const float* in; int in_ld, rows, D, wrap_dim;
  const unsigned short* img;                     // pack_fow_kernel: (2 D + 1) groups of 32 fragments
  const float* par;                              // pack_fow_kernel: FOW_NPAR floats
  float* Y; int accumulate;                      // (rows, 128) = or += the embedding
  int nwg, rep;                                  // workgroups of this embedding; passes of eight tiles per workgroup
};

struct FoWP { FoWSide e[3]; int count; };

int fow_set_attributes();
void fow_pack(const FoWSrc& src, unsigned short* img, float* par, hipStream_t stream);
void fow_launch(const FoWP& p, hipStream_t stream);

}  // namespace RIFT_NS
