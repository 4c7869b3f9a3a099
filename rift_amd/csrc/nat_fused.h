// Fused NAT level for gfx950: both NATLayers of one level of the agent history encoder
// (embedding.py:196-202 x2: LN -> qkv -> 1-D neighbourhood attention -> proj -> +res, LN -> fc1 -> GELU
// -> fc2 -> +res) in ONE kernel.  A workgroup keeps 80 rows (= 4/8/16 whole agent sequences at
// L = 20/10/5) resident in LDS for the whole level: the fp32 residual stream, the bf16 LayerNorm output,
// a 192-column bf16 chunk buffer (q|k|v of 4 heads, or a slice of the MLP hidden layer) and the attention
// output.  HBM traffic of a level drops from ~26 activation passes to one read + one write; weights
// (<= 0.65 MB per level) stream from L2 as MFMA B fragments, each fragment feeding 5 MFMAs.
#pragma once
#include "common.h"

namespace rift {

struct NatBlockW {
  const float* ln1_g; const float* ln1_b; const float* ln2_g; const float* ln2_b;
  const unsigned short* wqkv;   // bf16 [3C][C], rows in head-major order (head, part{q,k,v}, 16)
  const float* bqkv;            // fp32 [3C] same order
  const float* rpb;             // fp32 [H][2K-1]
  const unsigned short* wproj;  // bf16 [C][C]
  const float* bproj;
  const unsigned short* w1;     // bf16 [3C][C]
  const float* b1;
  const unsigned short* w2;     // bf16 [C][3C]
  const float* b2;
  float droppath;               // train-mode stochastic depth probability of this layer
};

struct NatLevelP {
  float* X;                     // (nseq*L, C) fp32 level input (read unless F9 is set); written back only if write_x
  int nseq;
  NatBlockW blk[2];
  uint32_t seed, stream;
  int dbg;                      // timing experiments only (RIFT_NAT_DBG bitmask); 0 in production
  // ---- fused neighbours of the level (all optional) ----
  const float* F9;              // level 0: (nseq*L, 9) agent features; the ConvTokenizer (embedding.py:57, k=3 pad 1) runs as prologue
  const unsigned short* w_tok;  // bf16 [32][32] tap-major (k, cin), K = 27 padded
  const float* b_tok;
  float* Oc;                    // (nseq*3, C): LayerNorm(norm_lv) of the last 3 steps -- all the FPN reads of this level
  const float* fn_g; const float* fn_b;
  float* Xnext;                 // (nseq*L/2, 2C): downsample conv (k=3 stride 2 pad 1, no bias) + LayerNorm(2C) (embedding.py:93-99)
  const unsigned short* w_ds;   // bf16 [2C][3C] tap-major
  const float* ds_g; const float* ds_b;
  int write_x;
  long long* ts;                // optional phase timestamps of workgroup 0 (diagnostic)
};

// reorder qkv rows of natten's (3, H, 16) layout to (H, 3, 16) and convert to bf16 (+ bias reorder)
__global__ void pack_qkv_headmajor_kernel(const float* __restrict__ w, const float* __restrict__ b, int C, int H,
                                          unsigned short* __restrict__ wo, float* __restrict__ bo) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 3 * C * C) return;
  const int pr = idx / C, k = idx - pr * C;
  const int head = pr / 48, part = (pr % 48) / 16, d = pr % 16;
  const int src = part * C + head * 16 + d;
  wo[fm_index(pr, k, C)] = f2bf(w[(size_t)src * C + k]);   // fragment-major image
  if (k == 0) bo[pr] = b[src];
}

// B fragments (weights) of one GEMM phase, fetched from L2 ahead of time so that the latency hides under
// the preceding LayerNorm / attention / epilogue phase.  W: bf16 [rows][ldw]; this wave's n-tiles are
// nt = j*4 + wave (j < NTW), n-tile nt covers rows n0 + nt*16 .. +15; k-step ks covers k0 + ks*32 .. +31.
template <int KS, int NTW>
struct BFrags { bf16x8 f[KS][NTW]; };

template <int NWV> struct NWaves {};
template <int KS, int NTW, int NW>
__device__ __forceinline__ void load_b(BFrags<KS, NTW>& B, const unsigned short* W, int ldw, int n0, int k0, int ntiles,
                                       int wave, int l15, int l4, bool skip, NWaves<NW>) {
  if (skip) ntiles = 0;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int nt = j * NW + wave;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      B.f[ks][j] = (nt < ntiles) ? fm_load(W, ldw, n0 + nt * 16, k0 + ks * 32, l4 * 16 + l15)
                                 : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  }
}

// acc[5][NTW] += A[80][K] (bf16 in LDS, row stride lda) . B
template <int KS, int NTW>
__device__ __forceinline__ void mma80(f32x4 (&acc)[5][NTW], const unsigned short* A, int lda, const BFrags<KS, NTW>& B,
                                      int l15, int l4) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 a[5];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(A + (mt * 16 + l15) * lda + ks * 32 + l4 * 8);
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
      for (int j = 0; j < NTW; ++j)   // swapped operands: acc[r] = out[row = mt*16 + (lane&15)][col = ntile*16 + 4*(lane>>4) + r]
        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B.f[ks][j], a[mt], acc[mt][j], 0, 0, 0);
  }
}

template <int MTN, int KS, int NTW>
__device__ __forceinline__ void mma_rows(f32x4 (&acc)[MTN][NTW], const unsigned short* A, int lda, const BFrags<KS, NTW>& B,
                                         int l15, int l4) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 a[MTN];
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(A + (mt * 16 + l15) * lda + ks * 32 + l4 * 8);
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B.f[ks][j], a[mt], acc[mt][j], 0, 0, 0);
  }
}

// dynamic LDS bytes of nat_level_kernel<C, NHEAD, L, KSZ, NW, CWMAX>
constexpr size_t nat_lds_bytes(int C, int NHEAD, int KSZ, int CWMAX, int ROWS = 80) {
  const int DSR = (ROWS / 2 + 15) / 16 * 16;
  const int C3 = 3 * C, CWK = C3 < CWMAX ? C3 : CWMAX, CB = CWK + 16, DSB = C3 + 16;
  const int cbsz = (C <= 64 && DSR * DSB > ROWS * CB) ? DSR * DSB : ROWS * CB;
  const int nrpb = NHEAD * (2 * KSZ - 1);
  return (size_t)ROWS * (C + 4) * 4 + (size_t)ROWS * (C + 16) * 2 * 2 + (size_t)cbsz * 2 + (size_t)2 * (12 * C + ((nrpb + 3) & ~3)) * 4 + (size_t)6 * C * 4;
}

// NW waves per workgroup (n-tiles and rows are dealt round-robin to the waves); CWMAX = widest qkv / hidden chunk
// WPE = waves per SIMD the register allocation must leave room for (2 = one 8-wave workgroup per CU, 4 = two)
// ROWS_ = rows per tile (a multiple of 16 and of L): small-C levels take larger tiles so that a phase carries enough work
template <int C, int NHEAD, int L, int KSZ, int NW = 4, int CWMAX = 192, int WPE = 1, int ROWS_ = 80>
__global__ __launch_bounds__(64 * NW, WPE) void nat_level_kernel(NatLevelP p) {
  constexpr int ROWS = ROWS_, MT = ROWS / 16;
  static_assert(ROWS % 16 == 0 && ROWS % L == 0 && (ROWS / 2) % (L / 2 > 0 ? L / 2 : 1) == 0, "tile = whole sequences, whole MFMA tiles");
  constexpr int C3 = 3 * C;
  constexpr int NTH = 64 * NW;
  constexpr int CWK = C3 < CWMAX ? C3 : CWMAX;        // chunk width (columns of qkv / hidden processed at once)
  constexpr int NCH = C3 / CWK;                   // 1, 1, 2
  constexpr int HPC = CWK / 48;                   // heads per qkv chunk
  // bf16 operand tiles: row strides = 16 (mod 32) elements -> conflict-free ds_read_b128 fragment reads (tools/lds_conflicts.py)
  constexpr int XS = C + 4, XN = C + 16, CB = CWK + 16;
  constexpr int KS1 = C / 32;                     // k-steps with K = C
  constexpr int KSC = CWK / 32;                   // k-steps with K = chunk
  constexpr int NT_CH = CWK / 16;                 // n-tiles of a chunk (6 or 12)
  constexpr int NTW_CH = (NT_CH + NW - 1) / NW;         // per wave (2 or 3)
  constexpr int NT_C = C / 16;                    // n-tiles of a C-wide output (2, 4, 8)
  constexpr int NTW_C = (NT_C + NW - 1) / NW;           // per wave (1, 1, 2)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* xs = reinterpret_cast<float*>(smem_raw);
  unsigned short* xn = reinterpret_cast<unsigned short*>(xs + ROWS * XS);
  unsigned short* cb = xn + ROWS * XN;
  constexpr int DSB = C3 + 16;                    // row stride of the downsample conv's A tile (staged in cb)
  constexpr int DSR = (ROWS / 2 + 15) / 16 * 16;   // rows of the downsample conv's A tile
  constexpr int CBSZ = (C <= 64 && DSR * DSB > ROWS * CB) ? DSR * DSB : ROWS * CB;
  unsigned short* ao = cb + CBSZ;
  // both layers' bias / LayerNorm / rpb vectors live in LDS (fetched once at kernel start): no epilogue or
  // LayerNorm begins with a dependent global load
  constexpr int NRPB = NHEAD * (2 * KSZ - 1);
  constexpr int P_LN1G = 0, P_LN1B = C, P_LN2G = 2 * C, P_LN2B = 3 * C, P_BQKV = 4 * C, P_BP = 7 * C, P_B1 = 8 * C,
                P_B2 = 11 * C, P_RPB = 12 * C, NPAR = 12 * C + ((NRPB + 3) & ~3);
  float* par = reinterpret_cast<float*>(ao + ROWS * XN);    // [2][NPAR]
  float* par2 = par + 2 * NPAR;                             // fn_g C | fn_b C | ds_g 2C | ds_b 2C
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int total_rows = p.nseq * L;
  const int ntiles = (total_rows + ROWS - 1) / ROWS;
  int row0 = blockIdx.x * ROWS;          // persistent workgroup: tiles blockIdx.x, blockIdx.x + gridDim.x, ...
  int tsn = 0;
#define NTS() do { if (p.ts && blockIdx.x == 0 && tid == 0 && tsn < 250) p.ts[tsn++] = clock64(); } while (0)
  NTS();

  BFrags<KS1, NTW_CH> Bq;      // qkv / fc1 weights of the current chunk
  BFrags<KS1, NTW_C> Bp;       // proj weights
  BFrags<KSC, NTW_C> B2;       // fc2 weights of the current hidden chunk
  load_b(Bq, p.blk[0].wqkv, C, 0, 0, NT_CH, wave, l15, l4, (p.dbg & 32) != 0, NWaves<NW>());
  constexpr bool tokenized = (C == 32);   // level 0 always starts from the agent features (p.F9) through the ConvTokenizer
  // The tile's input is fetched into registers one tile ahead (issued in the middle of the previous tile's second block),
  // so that no tile starts with an exposed HBM round trip.
  constexpr int NFV = ROWS * 32 / NTH;                          // ConvTokenizer window values per thread (level 0)
  constexpr int NV = (ROWS * (C / 4) + NTH - 1) / NTH;          // float4 of the residual tile per thread
  float fv[tokenized ? NFV : 1];
  float4 tv[tokenized ? 1 : NV];
  BFrags<1, NTW_C> Wt;
  if constexpr (tokenized) { load_b(Wt, p.w_tok, 32, 0, 0, NT_C, wave, l15, l4, (p.dbg & 32) != 0, NWaves<NW>()); }
  auto fetch_tile = [&](int r0) {
    if constexpr (tokenized) {
#pragma unroll
      for (int u = 0; u < NFV; ++u) {
        const int i = tid + u * NTH;
        const int r = i >> 5, kk = i & 31;
        const int tap = kk / 9, cin = kk - tap * 9;
        const int t = r % L, tt = t - 1 + tap;
        fv[u] = 0.f;
        if (kk < 27 && tt >= 0 && tt < L && r0 + r < total_rows) fv[u] = p.F9[(size_t)(r0 + r - t + tt) * 9 + cin];
      }
    } else {
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int i = tid + u * NTH;
        const int r = i / (C / 4), c4 = (i - r * (C / 4)) * 4;
        tv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < ROWS * (C / 4) && r0 + r < total_rows) tv[u] = *reinterpret_cast<const float4*>(p.X + (size_t)(r0 + r) * C + c4);
      }
    }
  };
  // every global load of the prologue -- the first tile, both layers' parameter vectors, the level-tail vectors -- is issued before
  // the first LDS store: one exposed round trip per workgroup instead of three (level 0 runs one tile per workgroup, so its prologue
  // is a quarter of its run time)
  fetch_tile(row0);
  {
    constexpr int NPV = (2 * NPAR + NTH - 1) / NTH;
    constexpr int NP2 = (6 * C + NTH - 1) / NTH;
    float pv[NPV], pv2[NP2];
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      const int i = tid + u * NTH;
      const NatBlockW& w = p.blk[i >= NPAR ? 1 : 0];
      const int e = i >= NPAR ? i - NPAR : i;
      const float* src = e < C ? w.ln1_g + e : e < 2 * C ? w.ln1_b + (e - C) : e < 3 * C ? w.ln2_g + (e - 2 * C)
                       : e < 4 * C ? w.ln2_b + (e - 3 * C) : e < 7 * C ? w.bqkv + (e - 4 * C) : e < 8 * C ? w.bproj + (e - 7 * C)
                       : e < 11 * C ? w.b1 + (e - 8 * C) : e < 12 * C ? w.b2 + (e - 11 * C) : w.rpb + (e - 12 * C);
      pv[u] = (i < 2 * NPAR && e < 12 * C + NRPB) ? *src : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NP2; ++u) {
      const int i = tid + u * NTH;
      pv2[u] = 0.f;
      if (i < 2 * C) { if (p.Oc) pv2[u] = i < C ? p.fn_g[i] : p.fn_b[i - C]; }
      else if (i < 6 * C && p.Xnext) pv2[u] = i < 4 * C ? p.ds_g[i - 2 * C] : p.ds_b[i - 4 * C];
    }
#pragma unroll
    for (int u = 0; u < NPV; ++u) { const int i = tid + u * NTH; if (i < 2 * NPAR) par[i] = pv[u]; }
#pragma unroll
    for (int u = 0; u < NP2; ++u) { const int i = tid + u * NTH; if (i < 6 * C) par2[i] = pv2[u]; }
  }

  auto commit_tile = [&]() {       // registers -> xs (through the ConvTokenizer MFMA step on level 0); ends with a barrier
    if constexpr (tokenized) {
      {
        // ---- ConvTokenizer prologue: xs = conv1d(F9, k=3, pad 1) + b as one K=32 MFMA step over the 3x9 window
#pragma unroll
        for (int u = 0; u < NFV; ++u) { const int i = tid + u * NTH; xn[(i >> 5) * XN + (i & 31)] = f2bf(fv[u]); }
        lds_barrier();
        f32x4 acc[MT][NTW_C];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < NTW_C; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_rows<MT, 1, NTW_C>(acc, xn, XN, Wt, l15, l4);
#pragma unroll
        for (int j = 0; j < NTW_C; ++j) {
          const int nt = j * NW + wave;
          if (nt >= NT_C) continue;
          const int col = nt * 16 + l4 * 4;
          const float4 b4 = *reinterpret_cast<const float4*>(p.b_tok + col);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<float4*>(xs + (mt * 16 + l15) * XS + col) =
                make_float4(acc[mt][j][0] + b4.x, acc[mt][j][1] + b4.y, acc[mt][j][2] + b4.z, acc[mt][j][3] + b4.w);
        }
        lds_barrier();
      }
    } else {
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int i = tid + u * NTH;
        const int r = i / (C / 4), c4 = (i - r * (C / 4)) * 4;
        if (i < ROWS * (C / 4)) *reinterpret_cast<float4*>(xs + r * XS + c4) = tv[u];
      }
      lds_barrier();
    }
  };

  // LayerNorm xs -> xn (bf16): C/8 lanes per row, 8 columns each (two 16-byte LDS reads, DPP statistics, packed fp32 math)
  auto layer_norm = [&](const float* g, const float* b) {
    constexpr int LPR = C / 8, RPS = 64 / LPR;            // lanes per row, rows per wave step
    const int lr = lane % LPR, rsub = lane / LPR;
    const float4 g0 = *reinterpret_cast<const float4*>(g + lr * 4), g1 = *reinterpret_cast<const float4*>(g + C / 2 + lr * 4);
    const float4 b0 = *reinterpret_cast<const float4*>(b + lr * 4), b1 = *reinterpret_cast<const float4*>(b + C / 2 + lr * 4);
#pragma unroll
    for (int r = wave * RPS + rsub; r < ROWS; r += NW * RPS) ln_row8<LPR>(xs + r * XS, xn + r * XN, g0, g1, b0, b1, lr);
  };

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  row0 = tile * ROWS;
  commit_tile();
  NTS();
  for (int bi = 0; bi < 2; ++bi) {
    const NatBlockW& w = p.blk[bi];
    const float* pb = par + bi * NPAR;
    // ======== attention half ========
    if (!(p.dbg & 1)) layer_norm(pb + P_LN1G, pb + P_LN1B);
    lds_barrier();
    NTS();
    for (int ch = 0; ch < NCH; ++ch) {
      // ---- qkv chunk GEMM: cb[80][CWK] = xn[80][C] . Wqkv[ch*CWK .. +CWK][C]^T + b ; q pre-scaled by 16^-0.5
      {
        f32x4 acc[MT][NTW_CH];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < NTW_CH; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(p.dbg & 8)) mma_rows<MT, KS1, NTW_CH>(acc, xn, XN, Bq, l15, l4);
        // prefetch the next phase's weights (next qkv chunk, or proj)
        if (ch + 1 < NCH) load_b(Bq, w.wqkv, C, (ch + 1) * CWK, 0, NT_CH, wave, l15, l4, (p.dbg & 32) != 0, NWaves<NW>());
        else load_b(Bp, w.wproj, C, 0, 0, NT_C, wave, l15, l4, (p.dbg & 32) != 0, NWaves<NW>());
#pragma unroll
        for (int j = 0; j < NTW_CH; ++j) {
          const int nt = j * NW + wave;
          if (nt >= NT_CH) continue;
          const int col = nt * 16 + l4 * 4;
          const float4 b4 = *reinterpret_cast<const float4*>(pb + P_BQKV + ch * CWK + col);
          const float sc = ((col % 48) < 16) ? 0.25f : 1.0f;    // q pre-scaled by 16^-0.5
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<uint2*>(cb + (mt * 16 + l15) * CB + col) =
                pack_bf16x4((acc[mt][j][0] + b4.x) * sc, (acc[mt][j][1] + b4.y) * sc, (acc[mt][j][2] + b4.z) * sc, (acc[mt][j][3] + b4.w) * sc);
        }
      }
      lds_barrier();
      NTS();
      // ---- neighbourhood attention of the chunk's heads (VALU, fp32 softmax); one (row, head) per work item
      for (int it = tid; it < ((p.dbg & 2) ? 0 : ROWS * HPC); it += NTH) {
        const int hh = it % HPC, row = it / HPC;
        const int a0 = (row / L) * L, i = row - a0;
        const int head = ch * HPC + hh;
        // q.k through v_dot2c_f32_bf16 on the packed operands (no unpacking), P.V as packed fp32 FMAs on unpacked pairs
        const uint4 q0 = *reinterpret_cast<const uint4*>(cb + row * CB + hh * 48), q1 = *reinterpret_cast<const uint4*>(cb + row * CB + hh * 48 + 8);
        int start = i - KSZ / 2;
        start = start < 0 ? 0 : start;
        start = start > L - KSZ ? L - KSZ : start;
        float sc[KSZ];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < KSZ; ++j) {
          const int nb = start + j;
          const unsigned short* kp = cb + (a0 + nb) * CB + hh * 48 + 16;
          const uint4 k0 = *reinterpret_cast<const uint4*>(kp), k1 = *reinterpret_cast<const uint4*>(kp + 8);
          float s0 = pb[P_RPB + head * (2 * KSZ - 1) + (nb - i) + KSZ - 1], s1 = 0.f;
          s0 = dot2_bf16(q0.x, k0.x, s0); s1 = dot2_bf16(q0.y, k0.y, s1); s0 = dot2_bf16(q0.z, k0.z, s0); s1 = dot2_bf16(q0.w, k0.w, s1);
          s0 = dot2_bf16(q1.x, k1.x, s0); s1 = dot2_bf16(q1.y, k1.y, s1); s0 = dot2_bf16(q1.z, k1.z, s0); s1 = dot2_bf16(q1.w, k1.w, s1);
          sc[j] = s0 + s1;
          mx = fmaxf(mx, sc[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < KSZ; ++j) { sc[j] = __expf(sc[j] - mx); den += sc[j]; }
        const float inv = __builtin_amdgcn_rcpf(den);
        f32x2_t o[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] = (f32x2_t)0.f;
#pragma unroll
        for (int j = 0; j < KSZ; ++j) {
          const unsigned short* vp = cb + (a0 + start + j) * CB + hh * 48 + 32;
          const uint4 v0 = *reinterpret_cast<const uint4*>(vp), v1 = *reinterpret_cast<const uint4*>(vp + 8);
          const f32x2_t wj = (f32x2_t)(sc[j] * inv);
          o[0] = __builtin_elementwise_fma(wj, unpack_bf16x2(v0.x), o[0]); o[1] = __builtin_elementwise_fma(wj, unpack_bf16x2(v0.y), o[1]);
          o[2] = __builtin_elementwise_fma(wj, unpack_bf16x2(v0.z), o[2]); o[3] = __builtin_elementwise_fma(wj, unpack_bf16x2(v0.w), o[3]);
          o[4] = __builtin_elementwise_fma(wj, unpack_bf16x2(v1.x), o[4]); o[5] = __builtin_elementwise_fma(wj, unpack_bf16x2(v1.y), o[5]);
          o[6] = __builtin_elementwise_fma(wj, unpack_bf16x2(v1.z), o[6]); o[7] = __builtin_elementwise_fma(wj, unpack_bf16x2(v1.w), o[7]);
        }
        uint4 o0, o1;
        o0.x = pack_bf16x2(o[0].x, o[0].y); o0.y = pack_bf16x2(o[1].x, o[1].y); o0.z = pack_bf16x2(o[2].x, o[2].y); o0.w = pack_bf16x2(o[3].x, o[3].y);
        o1.x = pack_bf16x2(o[4].x, o[4].y); o1.y = pack_bf16x2(o[5].x, o[5].y); o1.z = pack_bf16x2(o[6].x, o[6].y); o1.w = pack_bf16x2(o[7].x, o[7].y);
        *reinterpret_cast<uint4*>(ao + row * XN + head * 16) = o0;
        *reinterpret_cast<uint4*>(ao + row * XN + head * 16 + 8) = o1;
      }
      lds_barrier();
      NTS();
    }
    // ---- proj GEMM + residual: xs += droppath( ao . Wproj^T + b )
    {
      f32x4 acc[MT][NTW_C];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW_C; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      mma_rows<MT, KS1, NTW_C>(acc, ao, XN, Bp, l15, l4);
      load_b(Bq, w.w1, C, 0, 0, NT_CH, wave, l15, l4, (p.dbg & 32) != 0, NWaves<NW>());     // fc1 weights of hidden chunk 0
      float dps[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        dps[mt] = 1.f;
        if (w.droppath > 0.f)
          dps[mt] = (uniform01(p.seed, p.stream + 2 * bi, (uint32_t)((row0 + mt * 16 + l15) / L)) < w.droppath) ? 0.f : 1.0f / (1.0f - w.droppath);
      }
#pragma unroll
      for (int j = 0; j < NTW_C; ++j) {
        const int nt = j * NW + wave;
        if (nt >= NT_C) continue;
        const int col = nt * 16 + l4 * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(pb + P_BP + col);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          float4* xp = reinterpret_cast<float4*>(xs + (mt * 16 + l15) * XS + col);
          float4 x = *xp;
          x.x += (acc[mt][j][0] + b4.x) * dps[mt]; x.y += (acc[mt][j][1] + b4.y) * dps[mt];
          x.z += (acc[mt][j][2] + b4.z) * dps[mt]; x.w += (acc[mt][j][3] + b4.w) * dps[mt];
          *xp = x;
        }
      }
    }
    lds_barrier();
    NTS();
    // ======== MLP half ========
    if (!(p.dbg & 1)) layer_norm(pb + P_LN2G, pb + P_LN2B);
    if (bi == 1 && tile + (int)gridDim.x < ntiles) fetch_tile((tile + (int)gridDim.x) * ROWS);
    lds_barrier();
    NTS();
    {
      f32x4 acc2[MT][NTW_C];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW_C; ++j) acc2[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int ch = 0; ch < NCH; ++ch) {
        // ---- fc1 chunk: cb = gelu( xn . W1[ch*CWK..]^T + b1 )
        {
          f32x4 acc[MT][NTW_CH];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NTW_CH; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (!(p.dbg & 8)) mma_rows<MT, KS1, NTW_CH>(acc, xn, XN, Bq, l15, l4);
          load_b(B2, w.w2, C3, 0, ch * CWK, NT_C, wave, l15, l4, (p.dbg & 32) != 0, NWaves<NW>());          // fc2 weights of this hidden chunk
          if (ch > 0) lds_barrier();   // the previous chunk's fc2 reads of cb are complete
#pragma unroll
          for (int j = 0; j < NTW_CH; ++j) {
            const int nt = j * NW + wave;
            if (nt >= NT_CH) continue;
            const int col = nt * 16 + l4 * 4;
            const float4 b4 = *reinterpret_cast<const float4*>(pb + P_B1 + ch * CWK + col);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              *reinterpret_cast<uint2*>(cb + (mt * 16 + l15) * CB + col) = gelu4_pack(acc[mt][j], b4);
            }
          }
        }
        // weights needed after this fc2: next hidden chunk's fc1, or the next block's qkv chunk 0
        if (ch + 1 < NCH) load_b(Bq, w.w1, C, (ch + 1) * CWK, 0, NT_CH, wave, l15, l4, (p.dbg & 32) != 0, NWaves<NW>());
        else load_b(Bq, p.blk[1 - bi].wqkv, C, 0, 0, NT_CH, wave, l15, l4, (p.dbg & 32) != 0, NWaves<NW>());   // next block, or block 0 of the next tile
        lds_barrier();
        NTS();
        // ---- fc2 partial: acc2 += cb[80][CWK] . W2[:, ch*CWK..]^T
        if (!(p.dbg & 8)) mma_rows<MT, KSC, NTW_C>(acc2, cb, CB, B2, l15, l4);
        NTS();
      }
      float dps[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        dps[mt] = 1.f;
        if (w.droppath > 0.f)
          dps[mt] = (uniform01(p.seed, p.stream + 2 * bi + 1, (uint32_t)((row0 + mt * 16 + l15) / L)) < w.droppath) ? 0.f : 1.0f / (1.0f - w.droppath);
      }
#pragma unroll
      for (int j = 0; j < NTW_C; ++j) {
        const int nt = j * NW + wave;
        if (nt >= NT_C) continue;
        const int col = nt * 16 + l4 * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(pb + P_B2 + col);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          float4* xp = reinterpret_cast<float4*>(xs + (mt * 16 + l15) * XS + col);
          float4 x = *xp;
          x.x += (acc2[mt][j][0] + b4.x) * dps[mt]; x.y += (acc2[mt][j][1] + b4.y) * dps[mt];
          x.z += (acc2[mt][j][2] + b4.z) * dps[mt]; x.w += (acc2[mt][j][3] + b4.w) * dps[mt];
          *xp = x;
        }
      }
    }
    lds_barrier();
    NTS();
  }
  // ---- level output: what the FPN reads (normalised last 3 steps), the next level's input (downsample + LN), and X itself
  if (p.write_x) {
    for (int i = tid; i < ROWS * (C / 4); i += NTH) {
      const int r = i / (C / 4), c4 = (i - r * (C / 4)) * 4;
      if (row0 + r < total_rows)
        *reinterpret_cast<float4*>(p.X + (size_t)(row0 + r) * C + c4) = *reinterpret_cast<const float4*>(xs + r * XS + c4);
    }
  }
  if (p.Oc) {
    constexpr int LPR = C / 4, RPS = 64 / LPR, NSEQ = ROWS / L;
    const int lr = lane % LPR, rsub = lane / LPR;
    const float4 g4 = *reinterpret_cast<const float4*>(par2 + lr * 4), b4 = *reinterpret_cast<const float4*>(par2 + C + lr * 4);
    for (int it = wave * RPS + rsub; it < NSEQ * 3; it += NW * RPS) {
      const int a = it / 3, j = it - a * 3;
      const int r = a * L + L - 3 + j;
      const float4 v = *reinterpret_cast<const float4*>(xs + r * XS + lr * 4);
      float sm = (v.x + v.y) + (v.z + v.w);
      sm = group_sum<LPR>(sm);
      const float mean = sm * (1.0f / C);
      const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
      float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      q = group_sum<LPR>(q);
      const float rstd = rsqrtf(q * (1.0f / C) + 1e-5f);
      const int seq = row0 / L + a;
      if (seq < p.nseq)
        *reinterpret_cast<float4*>(p.Oc + ((size_t)seq * 3 + j) * C + lr * 4) =
            make_float4(d0 * rstd * g4.x + b4.x, d1 * rstd * g4.y + b4.y, d2 * rstd * g4.z + b4.z, d3 * rstd * g4.w + b4.w);
    }
  }
  if constexpr (C <= 64) {
    if (p.Xnext) {
      constexpr int RD = ROWS / 2, MD = DSR / 16, L2 = L / 2, C2 = 2 * C;   // ROWS / 2 output rows
      constexpr int NT2 = C2 / 16, NTW2 = (NT2 + NW - 1) / NW, DS = C2 + 4;
      static_assert(C > 64 || RD * DS <= ROWS * XS, "downsample output tile must fit the residual tile");
      BFrags<C3 / 32, NTW2> Wd;
      load_b(Wd, p.w_ds, C3, 0, 0, NT2, wave, l15, l4, (p.dbg & 32) != 0, NWaves<NW>());
      for (int i = tid; i < DSR * (C3 / 4); i += NTH) {
        const int m = i / (C3 / 4), k4 = (i - m * (C3 / 4)) * 4;
        const int tap = k4 / C, cin = k4 - tap * C;
        const int a = m / L2, j = m - a * L2;
        const int t = 2 * j - 1 + tap;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < RD && t >= 0 && t < L) v = *reinterpret_cast<const float4*>(xs + (a * L + t) * XS + cin);
        *reinterpret_cast<uint2*>(cb + m * DSB + k4) = pack_bf16x4(v.x, v.y, v.z, v.w);
      }
      lds_barrier();
      f32x4 acc[MD][NTW2];
#pragma unroll
      for (int mt = 0; mt < MD; ++mt)
#pragma unroll
        for (int j = 0; j < NTW2; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      mma_rows<MD, C3 / 32, NTW2>(acc, cb, DSB, Wd, l15, l4);
      float* ds = xs;                                                      // [40][DS] fp32 over the (dead) residual tile
#pragma unroll
      for (int j = 0; j < NTW2; ++j) {
        const int nt = j * NW + wave;
        if (nt >= NT2) continue;
        const int col = nt * 16 + l4 * 4;
#pragma unroll
        for (int mt = 0; mt < MD; ++mt) {
          const int m = mt * 16 + l15;
          if (m < RD) *reinterpret_cast<float4*>(ds + m * DS + col) = make_float4(acc[mt][j][0], acc[mt][j][1], acc[mt][j][2], acc[mt][j][3]);
        }
      }
      lds_barrier();
      constexpr int LPR = C2 / 4, RPS = 64 / LPR;
      const int lr = lane % LPR, rsub = lane / LPR;
      const float4 g4 = *reinterpret_cast<const float4*>(par2 + 2 * C + lr * 4), b4 = *reinterpret_cast<const float4*>(par2 + 4 * C + lr * 4);
      const int orow0 = row0 / 2, ototal = total_rows / 2;
      for (int m = wave * RPS + rsub; m < RD; m += NW * RPS) {
        const float4 v = *reinterpret_cast<const float4*>(ds + m * DS + lr * 4);
        float sm = (v.x + v.y) + (v.z + v.w);
        sm = group_sum<LPR>(sm);
        const float mean = sm * (1.0f / C2);
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        q = group_sum<LPR>(q);
        const float rstd = rsqrtf(q * (1.0f / C2) + 1e-5f);
        if (orow0 + m < ototal)
          *reinterpret_cast<float4*>(p.Xnext + (size_t)(orow0 + m) * C2 + lr * 4) =
              make_float4(d0 * rstd * g4.x + b4.x, d1 * rstd * g4.y + b4.y, d2 * rstd * g4.z + b4.z, d3 * rstd * g4.w + b4.w);
      }
    }
  }
  lds_barrier();   // xs / cb are re-used by the next tile
  }
  NTS();
#undef NTS
}

}  // namespace rift
