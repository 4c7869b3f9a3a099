// Fused PointsEncoder for gfx950 (embedding.py:254-296: first_mlp Linear-BN-ReLU-Linear -> max-pool over the points
// of a polyline -> concat -> second_mlp Linear-BN-ReLU-Linear -> max-pool), used for the map polygons
// (map_encoder.py:42-44, 20 points each) and the reference lines (planning_decoder.py:143-147, 120 points each).
//
// BatchNorm's batch statistics are a dependency on EVERY valid point of the minibatch, so the encoder is cut
// into three passes around the two statistics; inside a pass a tile of 120 point rows (6 polygons or 1 reference
// line) stays in LDS and nothing but the bf16 256-wide intermediate `f` goes to HBM, once:
//   pass A  pe_stats1_kernel : h1 = x W1^T + b1                         -> per-tile sum / sum of squares (BN1)
//   pass B  pe_mid_kernel    : h1 -> BN1 -> ReLU -> f = . W2^T + b2 (invalid rows 0) -> pooled = max over points
//                              gp = pooled W3b^T + b3 ;  g = f W3a^T + gp   -> per-tile sum / sum of squares (BN2)
//                              writes g as fp16 (the only intermediate that crosses HBM, 512 B per point)
//   pass C  pe_out_kernel    : g -> BN2 -> ReLU -> o = . W4^T + b4 (invalid rows 0) -> max over points -> out
// g is handed over in fp16, not bf16: BatchNorm subtracts the channel mean next, so the absolute rounding error of g is what
// counts; 11 significand bits keep it at the level of the bf16 rounding the normalised value gets anyway as an MFMA operand
// (|g| stays far inside the fp16 range).  The statistics are taken from the fp32 accumulators before rounding.
// All contractions are issued with the weight fragment as the MFMA A operand (see enc_fused.h): a lane holds four
// consecutive output channels of one point row.
#pragma once
#include <hip/hip_fp16.h>
#include "common.h"

namespace RIFT_NS {

struct PeP {
  const float* F; int Cin;              // (rows, Cin) fp32 point features, Cin <= 32
  const uint8_t* valid;                 // (rows) 1 = valid point
  int rows, ntiles;                     // tile t covers rows [120 t, 120 t + 120)
  const unsigned short *w1, *w2, *w3a, *w3b, *w4;   // bf16 [128][32] [256][128] [256][256] [256][256] [128][256]
  const float *b1, *b2, *b3, *b4;
  const float *s1, *t1, *s2, *t2;       // BatchNorm folded to y = x * s + t
  float* part1;                         // [2][128][ntiles] tile sums of h1, h1^2 over valid rows
  float* part2;                         // [2][256][ntiles] tile sums of g, g^2
  int* cnt;                             // [ntiles] valid rows per tile
  float* part1w; int* cnt1w; int nwg1;  // persistent pass A (pe_stats1p_kernel): [2][128][nwg1] sums over a workgroup's tiles, [nwg1] valid rows
  unsigned short* Fmid;                 // (rows, 256) fp16 bits: g = second_mlp.0 pre-activation
  float* gp;                            // (rows / NPTS, 256)
  float* out;                           // (rows / NPTS, 128)
  int do_stats;
  long long* ts;                        // optional phase timestamps of one workgroup (diagnostic)
  int ts_tile;
};

// ---- reference lines packed at tile granularity (round 4) --------------------------------------------------------------------------------
// pe_w_kernel used to walk the reference-line rows in rounds of two whole 120-point lines (240 rows = 15 tiles) whatever the lines' valid
// lengths -- at the benchmark's ragged lines (valid prefix ~ U{30..120}) 7.5 tiles per line where ~5.2 hold a valid point.  Here every
// line contributes only the 16-row tiles up to its LAST valid point, and a round is filled with whole lines (next-fit, in line order) up
// to 16 tiles / 16 lines: 512 -> ~300 rounds at the benchmark.  A tile then belongs to ONE line (no segment boundary inside a tile), only a
// line's eighth tile has rows that do not exist (points 120..127), and a line with dropped tiles takes 0 into its max as the reference's
// zero rows of invalid points do (embedding.py:282-293: invalid points are zero rows of the max; a row past the last valid point is one).
// Round record: 32 ints -- [T] (T < 16) tile descriptor (first point row of the tile << 8) | (tile index within its line << 4) | line slot of
// the round, or -1; [16 + s] line slot s: first tile | last tile << 8 | (tiles dropped) << 16 | (exists) << 17.
#define PEW_TAB_INTS 32
struct PePackP {
  const uint8_t* tiles; int nlines;     // (nlines) tiles of every line up to its last valid point (prep_kernel: refline_mask_body)
  int* tab;                             // out: [max rounds][PEW_TAB_INTS]
  int* hdr;                             // out: [0] packed rounds
  int max_rounds;
};
// Packing = bounded-space first fit (three open rounds; a line goes into the first open round with room, else the oldest round is closed
// and a new one opened): 16-tile rounds come out ~14.2 tiles full at the benchmark's lengths (plain next-fit: 13.6; ~330 rounds instead
// of 512 two-line rounds).  It sits on the map chain, i.e. on the step's critical path, so it has to be short: the tiles per line come from
// the preparation (prep_kernel reads the masks anyway), the line sequence is cut into up to 32 segments packed independently by 32 lanes
// in lock step (one sequential loop over all 1536 lines: > 100 us; 8 segments: 80 us; the segment ends cost ~10 rounds), the segments'
// rounds are numbered one after the other, and the whole thing runs in the extra block the BatchNorm-1 finalize launch already has for
// rounds are numbered one after the other.  WHERE it runs decided whether it paid: as a launch of its own at the head of the map chain
// (24 us of one workgroup) or as part of the BatchNorm-1 finalize launch (+15 us) it cost the step what the shorter pass B returned
// (0.662 / 0.655 against 0.651 ms), behind the ranking on the prepare stream more (0.673: the history chain is the step's longest); as
// one extra block of pass A's launch -- ~40 us of ~1000 blocks, the map chain's first big kernel -- it costs the chain nothing.
// the BatchNorm layers of the two encoders (same position in their pipelines) in one launch: blocks [0, a.C) are a's channels
// The rounds of pe_w_kernel (240 rows = two of pass A's 120-row tiles) that hold a valid point, per encoder, in ascending order, and the split
// of pe_w_kernel's persistent workgroups over the two encoders in proportion to those counts: built by one extra block of the BatchNorm-1
// finalize launch (it sits between pass A and pass B anyway).  A workgroup then walks list positions wg, wg + G, ...: every workgroup of an
// encoder gets the same number of live rounds to within one.  (Before: the split went by ALL rounds and a workgroup skipped its empty ones,
// so the reference-line workgroups -- a third of their rounds empty at the benchmark's ragged R -- finished at two thirds of the map
// encoder's five rounds.)
struct PeLiveP {
  const int* cnt[2]; int nt[2];    // pass A's valid-point counts per 120-row tile
  int nr[2];                       // rounds per encoder
  int grid;                        // pe_w_kernel's workgroups
  int* live[2];                    // out: live rounds
  int* hdr;                        // out: n_live a, n_live b, workgroups of a, workgroups of b
  const int* packed_b;             // if set: encoder b's rounds are PACKED (pe_pack_body): their number comes from here, no live list for b
};

#define PEW_PACK_SEGS 32
// (runs as one extra block of pe_stats1p_kernel, or as pe_pack_lines_kernel in eval mode: 256 threads; `pk_lds`: 2 * nlines ints)
__device__ __forceinline__ void pe_pack_body(const PePackP& q, int* __restrict__ pk_lds, int* __restrict__ seg_rounds /*[PEW_PACK_SEGS + 1]*/) {
  const int tid = threadIdx.x, NT = blockDim.x;
  int* const enc = pk_lds + q.nlines;   // (segment-local round << 16) | (tiles << 8) | (first tile << 4) | slot
  for (int l = tid; l < q.nlines; l += NT) pk_lds[l] = q.tiles[l];
  __syncthreads();
  const int nseg = min(PEW_PACK_SEGS, max(1, q.nlines / 32));
  const int per = (q.nlines + nseg - 1) / nseg;
  if (tid < PEW_PACK_SEGS) {
    int nround = 0;
    if (tid < nseg) {
      int bt0 = 0, bt1 = 0, bt2 = 0, bs0 = 0, bs1 = 0, bs2 = 0, br0 = 0, br1 = 0, br2 = 0, nopen = 0;
      const int lbeg = tid * per, lend = min(q.nlines, lbeg + per);
      for (int l = lbeg; l < lend; ++l) {
        const int t = pk_lds[l];
        if (t == 0) { enc[l] = -1; continue; }
        bool f0 = nopen > 0 && bt0 + t <= 16 && bs0 < 16;
        bool f1 = !f0 && nopen > 1 && bt1 + t <= 16 && bs1 < 16;
        bool f2 = !f0 && !f1 && nopen > 2 && bt2 + t <= 16 && bs2 < 16;
        if (!(f0 || f1 || f2)) {
          if (nopen == 3) { bt0 = bt1; bs0 = bs1; br0 = br1; bt1 = bt2; bs1 = bs2; br1 = br2; nopen = 2; }
          f0 = nopen == 0; f1 = nopen == 1; f2 = nopen == 2;
          if (f0) { bt0 = 0; bs0 = 0; br0 = nround; }
          if (f1) { bt1 = 0; bs1 = 0; br1 = nround; }
          if (f2) { bt2 = 0; bs2 = 0; br2 = nround; }
          ++nround; ++nopen;
        }
        const int bt = f0 ? bt0 : (f1 ? bt1 : bt2), bsl = f0 ? bs0 : (f1 ? bs1 : bs2), br = f0 ? br0 : (f1 ? br1 : br2);
        enc[l] = (br << 16) | (t << 8) | (bt << 4) | bsl;
        bt0 += f0 ? t : 0; bt1 += f1 ? t : 0; bt2 += f2 ? t : 0;
        bs0 += f0 ? 1 : 0; bs1 += f1 ? 1 : 0; bs2 += f2 ? 1 : 0;
      }
    }
    seg_rounds[tid + 1] = nround;
  }
  __syncthreads();
  if (tid == 0) {
    seg_rounds[0] = 0;
    for (int i = 1; i <= PEW_PACK_SEGS; ++i) seg_rounds[i] += seg_rounds[i - 1];
    q.hdr[0] = seg_rounds[PEW_PACK_SEGS];
  }
  __syncthreads();
  const int nr = seg_rounds[PEW_PACK_SEGS];
  for (int i = tid; i < nr * PEW_TAB_INTS; i += NT) q.tab[i] = (i & 31) < 16 ? -1 : 0;      // unused tiles / line slots
  __syncthreads();                       // (the same block writes the records below: ordered by the barrier)
  for (int l = tid; l < q.nlines; l += NT) {
    const int e = enc[l];
    if (e < 0) continue;
    const int round = (e >> 16) + seg_rounds[l / per], t = (e >> 8) & 0xff, t0 = (e >> 4) & 15, slot = e & 15;
    int* rec = q.tab + (size_t)round * PEW_TAB_INTS;
    for (int k = 0; k < t; ++k) rec[t0 + k] = ((l * 120 + 16 * k) << 8) | (k << 4) | slot;
    rec[16 + slot] = t0 | ((t0 + t - 1) << 8) | ((t < 8 ? 1 : 0) << 16) | (1 << 17);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void pe_pack_lines_kernel(PePackP q) {      // (eval mode: no pass A to ride on)
  extern __shared__ int pack_dyn[];
  __shared__ int seg_rounds[PEW_PACK_SEGS + 1];
  pe_pack_body(q, pack_dyn, seg_rounds);
}

template <int KS, int NTW>
struct PFrags { h16x8 f[KS][NTW]; };

// weight fragments of n-tiles (j * NW + wave), j < NTW, k in [k0, k0 + 32 KS)
template <int NW, int KS, int NTW>
__device__ __forceinline__ void p_load_w(PFrags<KS, NTW>& B, const unsigned short* W, int ldw, int k0, int wave, int l15, int l4) {
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      B.f[ks][j] = fm_load(W, ldw, (j * NW + wave) * 16, k0 + ks * 32, l4 * 16 + l15);
}

// acc[mt][j][r] (+)= sum_k A[mt*16 + l15][k0 + k] * W[ntile*16 + 4*l4 + r][k0 + k]
template <int MT, int KS, int NTW>
__device__ __forceinline__ void p_mma(f32x4 (&acc)[MT][NTW], const unsigned short* A, int lda, int k0, const PFrags<KS, NTW>& B,
                                      int l15, int l4) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    h16x8 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const h16x8*>(A + (mt * 16 + l15) * lda + k0 + ks * 32 + l4 * 8);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[mt][j] = mfma_h(B.f[ks][j], a[mt], acc[mt][j], 0, 0, 0);
  }
}

// the same over the first K row tiles only (K a compile-time count: a run-time predicate inside the loops made hipcc index the fragments
// dynamically -- 1 KB of scratch per lane -- or, row tile by row tile, double the registers)
template <int K, int MT, int KS, int NTW>
__device__ __forceinline__ void p_mma_first(f32x4 (&acc)[MT][NTW], const unsigned short* A, int lda, int k0, const PFrags<KS, NTW>& B,
                                            int l15, int l4) {
  static_assert(K <= MT, "row tiles");
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    h16x8 a[K];
#pragma unroll
    for (int mt = 0; mt < K; ++mt) a[mt] = *reinterpret_cast<const h16x8*>(A + (mt * 16 + l15) * lda + k0 + ks * 32 + l4 * 8);
#pragma unroll
    for (int mt = 0; mt < K; ++mt)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[mt][j] = mfma_h(B.f[ks][j], a[mt], acc[mt][j], 0, 0, 0);
  }
}

template <int MT, int NTW>
__device__ __forceinline__ void p_zero(f32x4 (&acc)[MT][NTW]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

#define PE_ROWS 128
#define PE_USED 120
#define PE_XS 40
#define PE_HS 136
#define PE_FS 264     // pass B's f / g tile: written twice per element by row-strided epilogues -> the +8 padding (tools/lds_conflicts.py)
#define PE_GS 272     // pass C's normalised-g tile: read as MFMA operand only -> conflict-free +16 padding

// stage the tile's point features as bf16 [128][PE_XS] (k >= Cin and unused rows zero) and the row flags
// (0 = not a row of this tile, 1 = valid point, 2 = invalid point: an all-zero row that still takes part in the max)
template <int NT>
__device__ __forceinline__ int pe_stage_x(const PeP& p, int tile, unsigned short* xin, unsigned char* sval, int tid) {
  const int row0 = tile * PE_USED;
  // every global load of the phase is issued before the first LDS store (no load -> store round trips)
  unsigned char fl = 0;
  if (tid < PE_ROWS && tid < PE_USED && row0 + tid < p.rows) fl = p.valid[row0 + tid] ? 1 : 2;
  constexpr int NX = PE_ROWS * 32 / NT;
  float xv[NX];
#pragma unroll
  for (int u = 0; u < NX; ++u) {
    const int i = tid + u * NT;
    const int r = i >> 5, k = i & 31;
    xv[u] = 0.f;
    if (k < p.Cin && r < PE_USED && row0 + r < p.rows) xv[u] = p.F[(size_t)(row0 + r) * p.Cin + k];
  }
  if (tid < PE_ROWS) sval[tid] = fl;
#pragma unroll
  for (int u = 0; u < NX; ++u) { const int i = tid + u * NT; xin[(i >> 5) * PE_XS + (i & 31)] = f2h(xv[u]); }
  return __syncthreads_count(fl == 1);   // also the barrier after staging
}

// per-channel sum and sum of squares over the valid rows of the tile: acc layout as p_mma; every wave owns its columns.
// addend(row, col) returns the four per-column terms added to acc[.][.][0..3] (bias / per-polyline term) as one float4.
template <int MT, int NTW, int NW, class F>
__device__ __forceinline__ void pe_tile_stats(const f32x4 (&acc)[MT][NTW], const unsigned char* sval, F&& addend, float* part, int C,
                                              int ntiles, int tile, int wave, int l15, int l4) {
  f32x2_t okf[MT];                                   // 1 / 0 row mask, applied with packed multiplies (two columns per instruction)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) okf[mt] = (f32x2_t)(sval[mt * 16 + l15] == 1 ? 1.0f : 0.0f);
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int col = (j * NW + wave) * 16 + l4 * 4;
    f32x2_t s01 = (f32x2_t)0.f, s23 = s01, q01 = s01, q23 = s01;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float4 a = addend(mt * 16 + l15, col);
      f32x2_t v01, v23, a01, a23;
      v01.x = acc[mt][j][0]; v01.y = acc[mt][j][1]; v23.x = acc[mt][j][2]; v23.y = acc[mt][j][3];
      a01.x = a.x; a01.y = a.y; a23.x = a.z; a23.y = a.w;
      v01 = (v01 + a01) * okf[mt]; v23 = (v23 + a23) * okf[mt];
      s01 += v01; s23 += v23;
      q01 = __builtin_elementwise_fma(v01, v01, q01); q23 = __builtin_elementwise_fma(v23, v23, q23);
    }
    float s[4] = {s01.x, s01.y, s23.x, s23.y}, q[4] = {q01.x, q01.y, q23.x, q23.y};
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[r] = sum16(s[r]); q[r] = sum16(q[r]); }
    if (l15 == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        part[(size_t)(col + r) * ntiles + tile] = s[r];
        part[(size_t)(C + col + r) * ntiles + tile] = q[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// pass A: BatchNorm-1 statistics of h1 = x W1^T + b1 (bf16 MFMA, exactly the values pass B recomputes)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pe_stats1_body(const PeP& p, const int tile) {
  __shared__ __attribute__((aligned(16))) unsigned short xin[PE_ROWS * PE_XS];
  __shared__ unsigned char sval[PE_ROWS];
  __shared__ __attribute__((aligned(16))) float b1s[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  PFrags<1, 2> W1;
  p_load_w<4, 1, 2>(W1, p.w1, 32, 0, wave, l15, l4);
  if (tid < 128) b1s[tid] = p.b1[tid];
  const int nv = pe_stage_x<256>(p, tile, xin, sval, tid);
  if (tid == 0) p.cnt[tile] = nv;
  if (nv == 0) {
    for (int i = tid; i < 256; i += 256) p.part1[(size_t)i * p.ntiles + tile] = 0.f;
    return;
  }
  f32x4 acc[8][2];
  p_zero(acc);
  p_mma<8, 1, 2>(acc, xin, PE_XS, 0, W1, l15, l4);
  pe_tile_stats<8, 2, 4>(acc, sval, [&](int, int c) { return *reinterpret_cast<const float4*>(b1s + c); }, p.part1, 128, p.ntiles, tile, wave, l15, l4);
}

// two encoders (map polygons: 20 points per polyline, reference lines: 120) in one launch: tiles [0, a.ntiles) belong to `a`
struct PeP2 { PeP a, b; };

// Pass A, persistent: a workgroup walks the tiles wg, wg + G, ... of one encoder with the per-lane partial sums of h1, h1^2 in registers and
// reduces / stores them ONCE (the one-tile-per-workgroup form below pays a 256-float scattered store, a cross-lane reduction and the launch
// floor of 2900 workgroups for 16 MFMAs of work each).  Still writes the per-tile valid counts (pass B skips empty rounds by them).
__device__ __forceinline__ void pe_stats1p_body(const PeP& p, const int wg, const int G) {
  __shared__ __attribute__((aligned(16))) unsigned short xin[PE_ROWS * PE_XS];
  __shared__ unsigned char sval[PE_ROWS];
  __shared__ __attribute__((aligned(16))) float b1s[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  PFrags<1, 2> W1;
  p_load_w<4, 1, 2>(W1, p.w1, 32, 0, wave, l15, l4);
  if (tid < 128) b1s[tid] = p.b1[tid];
  f32x2_t s01[2], s23[2], q01[2], q23[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { s01[j] = (f32x2_t)0.f; s23[j] = s01[j]; q01[j] = s01[j]; q23[j] = s01[j]; }
  int nvw = 0;
  // The tile's features are one contiguous run of 120 Cin floats: fetched linearly (five coalesced loads per thread; the [row][32]-shaped
  // gather of pe_stage_x is 16 mostly masked loads per thread, and those load instructions were what this pass took its 20 us for), the
  // next tile's before this tile's MFMAs.  Columns k >= Cin of the operand tile are zeroed once.
  constexpr int NXL = (PE_USED * 10 + 255) / 256;                  // Cin <= 10
  const int Cin = p.Cin;
  const float rcin = 1.0f / (float)Cin;
  float xv[NXL];
  unsigned char fl = 0;
  auto request = [&](int tile) {
    const int row0 = tile * PE_USED;
    const int nfl = tile < p.ntiles ? min(PE_USED, p.rows - row0) * Cin : 0;
    fl = 0;
    if (tile < p.ntiles && tid < PE_USED && row0 + tid < p.rows) fl = p.valid[row0 + tid] ? 1 : 2;
#pragma unroll
    for (int u = 0; u < NXL; ++u) { const int i = tid + u * 256; xv[u] = i < nfl ? p.F[(size_t)row0 * Cin + i] : 0.f; }
  };
  for (int i = tid; i < PE_ROWS * PE_XS / 2; i += 256) reinterpret_cast<unsigned int*>(xin)[i] = 0u;
  request(wg);
  __syncthreads();
  for (int tile = wg; tile < p.ntiles; tile += G) {
    if (tid < PE_ROWS) sval[tid] = fl;
#pragma unroll
    for (int u = 0; u < NXL; ++u) {
      const int i = tid + u * 256;
      if (i < PE_USED * Cin) {
        const int r = (int)(((float)i + 0.5f) * rcin), k = i - r * Cin;
        xin[r * PE_XS + k] = f2h(xv[u]);
      }
    }
    const int nv = __syncthreads_count(fl == 1);                  // (the barrier also publishes b1s the first time)
    request(tile + G);
    if (tid == 0) p.cnt[tile] = nv;
    nvw += nv;
    if (nv) {
      f32x4 acc[8][2];
      p_zero(acc);
      p_mma<8, 1, 2>(acc, xin, PE_XS, 0, W1, l15, l4);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float4 b = *reinterpret_cast<const float4*>(b1s + (j * 4 + wave) * 16 + l4 * 4);
        f32x2_t b01, b23;
        b01.x = b.x; b01.y = b.y; b23.x = b.z; b23.y = b.w;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          const f32x2_t ok = (f32x2_t)(sval[mt * 16 + l15] == 1 ? 1.0f : 0.0f);
          f32x2_t v01, v23;
          v01.x = acc[mt][j][0]; v01.y = acc[mt][j][1]; v23.x = acc[mt][j][2]; v23.y = acc[mt][j][3];
          v01 = (v01 + b01) * ok; v23 = (v23 + b23) * ok;
          s01[j] += v01; s23[j] += v23;
          q01[j] = __builtin_elementwise_fma(v01, v01, q01[j]); q23[j] = __builtin_elementwise_fma(v23, v23, q23[j]);
        }
      }
    }
    __syncthreads();                                               // the tile buffers are free again
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = (j * 4 + wave) * 16 + l4 * 4;
    float sv[4] = {s01[j].x, s01[j].y, s23[j].x, s23[j].y}, qv[4] = {q01[j].x, q01[j].y, q23[j].x, q23[j].y};
#pragma unroll
    for (int r = 0; r < 4; ++r) { sv[r] = sum16(sv[r]); qv[r] = sum16(qv[r]); }
    if (l15 == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p.part1w[(size_t)(col + r) * G + wg] = sv[r];
        p.part1w[(size_t)(128 + col + r) * G + wg] = qv[r];
      }
    }
  }
  if (tid == 0) p.cnt1w[wg] = nvw;
}

__global__ __launch_bounds__(256) void pe_stats1p_kernel(PeP2 q, PePackP pk) {
  // one extra block when pk is filled: the packed rounds of pass B.  It is block 0 -- dispatched first, so that its ~18 us run beside the
  // other blocks' ~20 us (as the LAST block it started when the first of the resident ones had finished: pass A 25 -> 38 us)
  const int blk = (int)blockIdx.x - (pk.tiles ? 1 : 0);
  if (blk < 0) {
    extern __shared__ int pack_dyn[];
    __shared__ int seg_rounds[PEW_PACK_SEGS + 1];
    pe_pack_body(pk, pack_dyn, seg_rounds);
    return;
  }
  if (blk < q.a.nwg1) pe_stats1p_body(q.a, blk, q.a.nwg1);
  else pe_stats1p_body(q.b, blk - q.a.nwg1, q.b.nwg1);
}

__global__ __launch_bounds__(256) void pe_stats1_kernel(PeP2 q) {
  if ((int)blockIdx.x < q.a.ntiles) pe_stats1_body(q.a, blockIdx.x);
  else pe_stats1_body(q.b, blockIdx.x - q.a.ntiles);
}

// ---------------------------------------------------------------------------------------------------------------
// pass B
// ---------------------------------------------------------------------------------------------------------------
#define PE_MID_LDS (PE_ROWS * PE_XS * 2 + PE_ROWS * PE_HS * 2 + PE_ROWS * PE_FS * 2 + 16 * PE_FS * 2 + 8 * 256 * 4 + 1024 * 4 + PE_ROWS)

template <int NPTS>
__device__ __forceinline__ void pe_mid_body(const PeP& p, const int tile) {
  constexpr int MT = 8, GPT = PE_USED / NPTS, NW = 8;
  constexpr int SPL = GPT >= 4 ? 1 : 4;            // row splits per polyline for the max-pool (units = GPT * SPL >= 4)
  constexpr int UNITS = GPT * SPL, RPU = NPTS / SPL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* xin = reinterpret_cast<unsigned short*>(smem_raw);
  unsigned short* h1 = xin + PE_ROWS * PE_XS;
  unsigned short* fl = h1 + PE_ROWS * PE_HS;
  unsigned short* pool = fl + PE_ROWS * PE_FS;                     // [16][PE_FS] bf16 (rows >= GPT zero)
  float* gpl = reinterpret_cast<float*>(pool + 16 * PE_FS);        // [8][256]: gp of the tile's polylines; max-pool scratch before
  float* par = gpl + 8 * 256;                                      // b1 128 | s1 128 | t1 128 | b2 256 | b3 256
  unsigned char* sval = reinterpret_cast<unsigned char*>(par + 1024);
  constexpr int P_B1 = 0, P_S1 = 128, P_T1 = 256, P_B2 = 384, P_B3 = 640;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int row0 = tile * PE_USED;
  int tsn = 0;
#define PTS() do { if (p.ts && tile == p.ts_tile && tid == 0) p.ts[tsn++] = clock64(); } while (0)
  PTS();

  PFrags<1, 1> W1;
  PFrags<4, 2> Wa, Wb;
  p_load_w<NW, 1, 1>(W1, p.w1, 32, 0, wave, l15, l4);
  p_load_w<NW, 4, 2>(Wa, p.w2, 128, 0, wave, l15, l4);
  {
    float pv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * 512;
      const float* src = e < 128 ? p.b1 + e : e < 256 ? p.s1 + (e - 128) : e < 384 ? p.t1 + (e - 256) : e < 640 ? p.b2 + (e - 384) : p.b3 + (e - 640);
      pv[u] = e < 896 ? *src : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int e = tid + u * 512; if (e < 896) par[e] = pv[u]; }
  }
  for (int i = tid; i < 16 * PE_FS / 2; i += 512) reinterpret_cast<unsigned int*>(pool)[i] = 0u;
  const int nv = pe_stage_x<512>(p, tile, xin, sval, tid);
  if (nv == 0) {   // no valid point: f = 0, gp unused, zero statistics
    if (p.do_stats) for (int i = tid; i < 512; i += 512) p.part2[(size_t)i * p.ntiles + tile] = 0.f;
    return;
  }

  PTS();
  // ---- h1 = relu(bn1(x W1^T + b1)) -> LDS bf16
  {
    f32x4 acc[MT][1];
    p_zero(acc);
    p_mma<MT, 1, 1>(acc, xin, PE_XS, 0, W1, l15, l4);
    const int col = wave * 16 + l4 * 4;
    const float4 b = *reinterpret_cast<const float4*>(par + P_B1 + col), s = *reinterpret_cast<const float4*>(par + P_S1 + col),
                 t = *reinterpret_cast<const float4*>(par + P_T1 + col);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      *reinterpret_cast<uint2*>(h1 + (mt * 16 + l15) * PE_HS + col) =
          pack_h4(fmaxf((acc[mt][0][0] + b.x) * s.x + t.x, 0.f), fmaxf((acc[mt][0][1] + b.y) * s.y + t.y, 0.f),
                      fmaxf((acc[mt][0][2] + b.z) * s.z + t.z, 0.f), fmaxf((acc[mt][0][3] + b.w) * s.w + t.w, 0.f));
  }
  __syncthreads();
  PTS();

  // ---- f = h1 W2^T + b2, invalid rows zero -> LDS bf16
  {
    f32x4 acc[MT][2];
    p_zero(acc);
    p_mma<MT, 4, 2>(acc, h1, PE_HS, 0, Wa, l15, l4);
    p_load_w<NW, 4, 2>(Wa, p.w3b, 256, 0, wave, l15, l4);
    p_load_w<NW, 4, 2>(Wb, p.w3b, 256, 128, wave, l15, l4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (j * NW + wave) * 16 + l4 * 4;
      const float4 b = *reinterpret_cast<const float4*>(par + P_B2 + col);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 16 + l15;
        const bool ok = sval[row] == 1;
        *reinterpret_cast<uint2*>(fl + row * PE_FS + col) =
            ok ? pack_h4(acc[mt][j][0] + b.x, acc[mt][j][1] + b.y, acc[mt][j][2] + b.z, acc[mt][j][3] + b.w) : make_uint2(0u, 0u);
      }
    }
  }
  __syncthreads();
  PTS();

  // ---- the per-polyline max over its points
  {
    const int cp = (tid & 127) * 2, sub = tid >> 7;
    for (int u = sub; u < UNITS; u += 4) {
      const int g = u / SPL, r0 = g * NPTS + (u % SPL) * RPU;
      float m0 = -INFINITY, m1 = -INFINITY;
      for (int r = r0; r < r0 + RPU; ++r) {
        const unsigned int v = *reinterpret_cast<const unsigned int*>(fl + r * PE_FS + cp);
        m0 = fmaxf(m0, h_lo(v));
        m1 = fmaxf(m1, h_hi(v));
      }
      gpl[u * 256 + cp] = m0; gpl[u * 256 + cp + 1] = m1;
    }
  }
  __syncthreads();
  if (tid < 256) {
#pragma unroll
    for (int g = 0; g < GPT; ++g) {
      float m = gpl[g * SPL * 256 + tid];
#pragma unroll
      for (int s = 1; s < SPL; ++s) m = fmaxf(m, gpl[(g * SPL + s) * 256 + tid]);
      pool[g * PE_FS + tid] = (unsigned short)(h_pair_exact(m, 0.f) & 0xffffu);   // already an operand-format value
    }
  }
  __syncthreads();
  PTS();

  // ---- gp = pooled W3b^T + b3 (one row per polyline)
  {
    f32x4 acc[1][2];
    p_zero(acc);
    p_mma<1, 4, 2>(acc, pool, PE_FS, 0, Wa, l15, l4);
    p_mma<1, 4, 2>(acc, pool, PE_FS, 128, Wb, l15, l4);
    p_load_w<NW, 4, 2>(Wa, p.w3a, 256, 0, wave, l15, l4);
    p_load_w<NW, 4, 2>(Wb, p.w3a, 256, 128, wave, l15, l4);
    if (l15 < GPT) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = (j * NW + wave) * 16 + l4 * 4;
        const float4 b = *reinterpret_cast<const float4*>(par + P_B3 + col);
        const float4 v = make_float4(acc[0][j][0] + b.x, acc[0][j][1] + b.y, acc[0][j][2] + b.z, acc[0][j][3] + b.w);
        *reinterpret_cast<float4*>(gpl + l15 * 256 + col) = v;
        const int grp = tile * GPT + l15;
        if ((size_t)grp * NPTS < (size_t)p.rows) *reinterpret_cast<float4*>(p.gp + (size_t)grp * 256 + col) = v;
      }
    }
  }
  __syncthreads();
  PTS();

  // ---- g = f W3a^T + gp: BatchNorm-2 statistics from the fp32 accumulators, the values to HBM as fp16
  {
    f32x4 acc[MT][2];
    p_zero(acc);
    p_mma<MT, 4, 2>(acc, fl, PE_FS, 0, Wa, l15, l4);
    p_mma<MT, 4, 2>(acc, fl, PE_FS, 128, Wb, l15, l4);
    // rows 120..127 of the MFMA tile belong to no polyline: they read polyline 0's term (their statistics weight is zero and their g is never stored)
    auto addend = [&](int row, int c) { return *reinterpret_cast<const float4*>(gpl + (GPT == 1 || row >= PE_USED ? 0 : row / NPTS) * 256 + c); };
    if (p.do_stats) pe_tile_stats<MT, 2, NW>(acc, sval, addend, p.part2, 256, p.ntiles, tile, wave, l15, l4);
    __syncthreads();                       // every wave is done reading f: its tile becomes the fp16 staging area of g
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (j * NW + wave) * 16 + l4 * 4;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 16 + l15;
        const float4 a = addend(row, col);
        const __half2 lo = __floats2half2_rn(acc[mt][j][0] + a.x, acc[mt][j][1] + a.y), hi = __floats2half2_rn(acc[mt][j][2] + a.z, acc[mt][j][3] + a.w);
        uint2 u;
        u.x = *reinterpret_cast<const unsigned int*>(&lo); u.y = *reinterpret_cast<const unsigned int*>(&hi);
        *reinterpret_cast<uint2*>(fl + row * PE_FS + col) = u;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < PE_USED * 32; i += 512) {       // 16 B per lane, whole rows
    const int r = i >> 5, c8 = (i & 31) * 8;
    if (row0 + r < p.rows) *reinterpret_cast<uint4*>(p.Fmid + (size_t)(row0 + r) * 256 + c8) = *reinterpret_cast<const uint4*>(fl + r * PE_FS + c8);
  }
  PTS();
#undef PTS
}

__global__ __launch_bounds__(512) void pe_mid_kernel(PeP2 q) {
  if ((int)blockIdx.x < q.a.ntiles) pe_mid_body<20>(q.a, blockIdx.x);
  else pe_mid_body<120>(q.b, blockIdx.x - q.a.ntiles);
}

// ---------------------------------------------------------------------------------------------------------------
// pass C
// ---------------------------------------------------------------------------------------------------------------
#define PE_OUT_LDS (PE_ROWS * PE_GS * 2 + 4 * 128 * 4 + 640 * 4 + PE_ROWS)

template <int NPTS>
__device__ __forceinline__ void pe_out_body(const PeP& p, const int tile) {
  constexpr int MT = 8, GPT = PE_USED / NPTS, NW = 8, OS = 132;
  constexpr int SPL = GPT >= 4 ? 1 : 4;
  constexpr int UNITS = GPT * SPL, RPU = NPTS / SPL;
  static_assert(PE_ROWS * OS * 4 <= PE_ROWS * PE_GS * 2, "o tile must fit the activation tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* fl = reinterpret_cast<unsigned short*>(smem_raw);      // relu(bn2(g)) as bf16, then o (fp32) over the same bytes
  float* ol = reinterpret_cast<float*>(smem_raw);
  float* pm = reinterpret_cast<float*>(fl + PE_ROWS * PE_GS);            // [4][128] max-pool partials
  float* par = pm + 4 * 128;                                             // s2 256 | t2 256 | b4 128
  unsigned char* sval = reinterpret_cast<unsigned char*>(par + 640);
  constexpr int P_S2 = 0, P_T2 = 256, P_B4 = 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int row0 = tile * PE_USED;

  PFrags<4, 1> Wc, Wd;
  p_load_w<NW, 4, 1>(Wc, p.w4, 256, 0, wave, l15, l4);
  p_load_w<NW, 4, 1>(Wd, p.w4, 256, 128, wave, l15, l4);
  unsigned char fv = 0;
  if (tid < PE_ROWS && tid < PE_USED && row0 + tid < p.rows) fv = p.valid[row0 + tid] ? 1 : 2;
  // the row tiles behind the last valid point hold zero rows only (a reference line's valid points are a prefix of its 120): their g is
  // not fetched and their rows take no part in the contraction -- the o rows of invalid points are zero whatever the GEMM says
  unsigned long long* sbal = reinterpret_cast<unsigned long long*>(pm);
  { const unsigned long long bal = __ballot(fv == 1); if (lane == 0 && wave < 2) sbal[wave] = bal; }
  const int nv = __syncthreads_count(fv == 1);
  const unsigned long long bal0 = sbal[0], bal1 = sbal[1];
  const int nmt = __builtin_amdgcn_readfirstlane(((bal1 ? 128 - __builtin_clzll(bal1) : bal0 ? 64 - __builtin_clzll(bal0) : 0) + 15) >> 4);
  if (nv == 0) {   // every row zero -> the max is zero
    for (int i = tid; i < GPT * 128; i += 512) {
      const int grp = tile * GPT + i / 128;
      if ((size_t)grp * NPTS < (size_t)p.rows) p.out[(size_t)grp * 128 + (i & 127)] = 0.f;
    }
    return;
  }
  // ---- g (fp16) -> relu(bn2(g)) as the bf16 MFMA operand tile; all loads first
  {
    uint4 gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + u * 512, r = i >> 5, c8 = (i & 31) * 8;
      gv[u] = make_uint4(0u, 0u, 0u, 0u);
      if (r < PE_USED && row0 + r < p.rows && r < 16 * nmt) gv[u] = *reinterpret_cast<const uint4*>(p.Fmid + (size_t)(row0 + r) * 256 + c8);
    }
    float pv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int e = tid + u * 512; pv[u] = e < 256 ? p.s2[e] : e < 512 ? p.t2[e - 256] : e < 640 ? p.b4[e - 512] : 0.f; }
    if (tid < PE_ROWS) sval[tid] = fv;
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int e = tid + u * 512; if (e < 640) par[e] = pv[u]; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + u * 512, r = i >> 5, c8 = (i & 31) * 8;
      const unsigned int w[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
      float x[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float2 f2 = __half22float2(*reinterpret_cast<const __half2*>(&w[k]));
        x[2 * k] = f2.x; x[2 * k + 1] = f2.y;
      }
      const float4 s0 = *reinterpret_cast<const float4*>(par + P_S2 + c8), s1 = *reinterpret_cast<const float4*>(par + P_S2 + c8 + 4);
      const float4 t0 = *reinterpret_cast<const float4*>(par + P_T2 + c8), t1 = *reinterpret_cast<const float4*>(par + P_T2 + c8 + 4);
      uint4 o;
      o.x = pack_h2(fmaxf(x[0] * s0.x + t0.x, 0.f), fmaxf(x[1] * s0.y + t0.y, 0.f));
      o.y = pack_h2(fmaxf(x[2] * s0.z + t0.z, 0.f), fmaxf(x[3] * s0.w + t0.w, 0.f));
      o.z = pack_h2(fmaxf(x[4] * s1.x + t1.x, 0.f), fmaxf(x[5] * s1.y + t1.y, 0.f));
      o.w = pack_h2(fmaxf(x[6] * s1.z + t1.z, 0.f), fmaxf(x[7] * s1.w + t1.w, 0.f));
      if (r < 16 * nmt) *reinterpret_cast<uint4*>(fl + r * PE_GS + c8) = o;     // (rows that do not exist carry relu(t2): masked below)
    }
  }
  __syncthreads();

  // ---- o = . W4^T + b4, invalid rows zero (fp32 tile over the same LDS), then the max over each polyline
  {
    f32x4 acc[MT][1];
    p_zero(acc);
    if (nmt <= 2) { p_mma_first<2, MT, 4, 1>(acc, fl, PE_GS, 0, Wc, l15, l4); p_mma_first<2, MT, 4, 1>(acc, fl, PE_GS, 128, Wd, l15, l4); }
    else if (nmt <= 4) { p_mma_first<4, MT, 4, 1>(acc, fl, PE_GS, 0, Wc, l15, l4); p_mma_first<4, MT, 4, 1>(acc, fl, PE_GS, 128, Wd, l15, l4); }
    else if (nmt <= 6) { p_mma_first<6, MT, 4, 1>(acc, fl, PE_GS, 0, Wc, l15, l4); p_mma_first<6, MT, 4, 1>(acc, fl, PE_GS, 128, Wd, l15, l4); }
    else { p_mma<MT, 4, 1>(acc, fl, PE_GS, 0, Wc, l15, l4); p_mma<MT, 4, 1>(acc, fl, PE_GS, 128, Wd, l15, l4); }
    __syncthreads();
    const int col = wave * 16 + l4 * 4;
    const float4 b = *reinterpret_cast<const float4*>(par + P_B4 + col);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 16 + l15;
      const bool ok = sval[row] == 1;
      *reinterpret_cast<float4*>(ol + row * OS + col) =
          ok ? make_float4(acc[mt][0][0] + b.x, acc[mt][0][1] + b.y, acc[mt][0][2] + b.z, acc[mt][0][3] + b.w) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  {
    const int c = tid & 127, sub = tid >> 7;
    for (int u = sub; u < UNITS; u += 4) {
      const int g = u / SPL, r0 = g * NPTS + (u % SPL) * RPU;
      float m = -INFINITY;
      for (int r = r0; r < r0 + RPU; ++r) m = fmaxf(m, ol[r * OS + c]);
      if (SPL == 1) {
        const int grp = tile * GPT + g;
        if ((size_t)grp * NPTS < (size_t)p.rows) p.out[(size_t)grp * 128 + c] = m;
      } else pm[u * 128 + c] = m;
    }
  }
  if (SPL > 1) {
    __syncthreads();
    if (tid < 128) {
#pragma unroll
      for (int g = 0; g < GPT; ++g) {
        float m = pm[g * SPL * 128 + tid];
#pragma unroll
        for (int s = 1; s < SPL; ++s) m = fmaxf(m, pm[(g * SPL + s) * 128 + tid]);
        const int grp = tile * GPT + g;
        if ((size_t)grp * NPTS < (size_t)p.rows) p.out[(size_t)grp * 128 + tid] = m;
      }
    }
  }
}

__global__ __launch_bounds__(512) void pe_out_kernel(PeP2 q) {
  if ((int)blockIdx.x < q.a.ntiles) pe_out_body<20>(q.a, blockIdx.x);
  else pe_out_body<120>(q.b, blockIdx.x - q.a.ntiles);
}

// BatchNorm finalize over tile partials laid out [2][C][nblk] (fp32 tile sums, fp64 accumulation).  One 256-thread workgroup per
// channel, coalesced over the tiles with every load of a thread issued before the first add (the kernel is a pure latency chain:
// one wave per channel walking 24 dependent rounds cost 11 us per launch).  Same semantics as bn_finalize_kernel.
struct BnFinP {
  const float* part; const int* cnt; int nblk, C;
  const int* nblk_dev;             // if set: the number of partials comes from device memory (pe_w_kernel's workgroup split, PeLiveP)
  const float* gamma; const float* beta; float* running_mean; float* running_var; long long* num_batches;
  float* scale; float* shift;
  double* sums; int sums_mode;     // data parallel: [2C+1] (sum, sum of squares, count); 0 local, 1 emit this rank's sums, 2 consume reduced sums
};

__device__ __forceinline__ void bn_finalize_t_body(const BnFinP& p, int train, int update_running, float eps, int c) {
  constexpr int NT = 256, MAXU = 8;                      // up to 2048 tiles in one unrolled round; more take further rounds
  const int tid = threadIdx.x;
  const int nblk = p.nblk_dev ? *p.nblk_dev : p.nblk;
  float mean, var;
  if (train) {
    double s = 0.0, q = 0.0;
    long long n = 0;
    for (int b0 = 0; b0 < (p.sums_mode == 2 ? 0 : nblk); b0 += NT * MAXU) {
      float sv[MAXU], qv[MAXU]; int nv[MAXU];
#pragma unroll
      for (int u = 0; u < MAXU; ++u) {
        const int b = b0 + tid + u * NT;
        const bool ok = b < nblk;
        sv[u] = ok ? p.part[(size_t)c * nblk + b] : 0.f;
        qv[u] = ok ? p.part[(size_t)(p.C + c) * nblk + b] : 0.f;
        nv[u] = ok ? p.cnt[b] : 0;
      }
#pragma unroll
      for (int u = 0; u < MAXU; ++u) { s += (double)sv[u]; q += (double)qv[u]; n += nv[u]; }
    }
    s = wave_sum_d(s); q = wave_sum_d(q);
    double nd = wave_sum_d((double)n);                    // counts stay far below 2^53: exact
    __shared__ double red[3][4];
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = q; red[2][tid >> 6] = nd; }
    __syncthreads();
    s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    nd = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    if (p.sums_mode == 1) {
      if (tid == 0) { p.sums[c] = s; p.sums[p.C + c] = q; if (c == 0) p.sums[2 * p.C] = nd; }
      return;
    }
    if (p.sums_mode == 2) { s = p.sums[c]; q = p.sums[p.C + c]; nd = p.sums[2 * p.C]; }
    const double mu = s / nd;
    double v = q / nd - mu * mu;
    v = v < 0.0 ? 0.0 : v;
    mean = (float)mu; var = (float)v;
    if (update_running && tid == 0) {
      const double unb = nd > 1.0 ? v * nd / (nd - 1.0) : v;
      p.running_mean[c] = 0.9f * p.running_mean[c] + 0.1f * mean;
      p.running_var[c] = 0.9f * p.running_var[c] + 0.1f * (float)unb;
      if (c == 0 && p.num_batches) *p.num_batches += 1;
    }
  } else { mean = p.running_mean[c]; var = p.running_var[c]; }
  if (tid == 0) {
    const float sc = p.gamma[c] * rsqrtf(var + eps);
    p.scale[c] = sc;
    p.shift[c] = p.beta[c] - mean * sc;
  }
}


__device__ __forceinline__ void pe_live_body(const PeLiveP& q) {
  __shared__ int sc[256];
  __shared__ int tot[2];
  const int tid = threadIdx.x;
  for (int e = 0; e < 2; ++e) {
    if (e == 1 && q.packed_b) { if (tid == 255) tot[1] = *q.packed_b; __syncthreads(); continue; }
    if (!q.cnt[e]) { if (tid == 255) tot[e] = q.nr[e]; __syncthreads(); continue; }      // (eval mode: no pass A, every round of this encoder)
    const int R = q.nr[e], K = (R + 255) / 256, r0 = tid * K;
    auto is_live = [&](int r) { return r < R && (q.cnt[e][2 * r] + (2 * r + 1 < q.nt[e] ? q.cnt[e][2 * r + 1] : 0)) > 0; };
    int mine = 0;
    for (int j = 0; j < K; ++j) mine += is_live(r0 + j) ? 1 : 0;
    sc[tid] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int v = tid >= d ? sc[tid - d] : 0;
      __syncthreads();
      sc[tid] += v;
      __syncthreads();
    }
    int o = sc[tid] - mine;
    for (int j = 0; j < K; ++j) if (is_live(r0 + j)) q.live[e][o++] = r0 + j;
    if (tid == 255) tot[e] = sc[255];
    __syncthreads();
  }
  if (tid == 0) {
    const int na = tot[0], nb = tot[1], g = q.grid;
    int a = na + nb > 0 ? (int)(((long long)g * na + (na + nb) / 2) / (na + nb)) : 0;
    if (na > 0 && a < 1) a = 1;
    if (nb > 0 && a > g - 1) a = g - 1;
    if (nb == 0) a = g;
    q.hdr[0] = na; q.hdr[1] = nb; q.hdr[2] = a; q.hdr[3] = g - a;
  }
}

__global__ __launch_bounds__(256) void bn_finalize_t_kernel(BnFinP a, BnFinP b, int train, int update_running, float eps, PeLiveP lv) {
  if ((int)blockIdx.x >= a.C + b.C) { pe_live_body(lv); return; }      // (one extra block, launched only when lv is filled)
  if ((int)blockIdx.x < a.C) bn_finalize_t_body(a, train, update_running, eps, blockIdx.x);
  else bn_finalize_t_body(b, train, update_running, eps, blockIdx.x - a.C);
}

}  // namespace RIFT_NS
