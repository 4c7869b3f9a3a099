// Fused FPN tail of the agent history encoder for gfx950 (embedding.py:60-87 restricted to what out[:, :, -1] reads):
// the three lateral convs (k = 3) on the LayerNorm'ed last three steps of each NAT level, the top-down merge
// (F.interpolate(scale 2, linear, align_corners = False) -> weights .25/.75 and 1) and the last-step fpn conv
// (taps 0, 1 only), for 64 agents per workgroup.  Replaces three conv GEMMs, the merge kernel and a GEMM, and their
// five HBM round trips, by one launch that reads 3 x (3, C) rows and writes one 128-vector per agent.
// Row r of the 128-row MFMA tile is (agent r >> 1, position r & 1); operands swapped as in enc_fused.h, so a lane holds
// four output channels of one row and the other position of the same agent sits in the neighbouring lane (DPP swap).
#pragma once
#include "common.h"
#include "pe_fused.h"

namespace RIFT_NS {

struct FpnP {
  const float* oc[3];            // level i: (nA, 3, C_i) normalised last three steps, C = 32, 64, 128
  const unsigned short* ocb[3];  // if set: the same rows already rounded to bf16 by the NAT kernel that wrote them
  const unsigned short* wl[3];   // lateral conv weights, fragment-major bf16 [128][3 C_i] tap-major
  const float* bl[3];
  const unsigned short* wf;      // fpn conv at the last step, taps 0 and 1: [128][256]
  const float* bf_;
  float* out;                    // (nA, 128)
  int nA;
  const int* cnt; const int* aidx;   // compacted launch (nat_l0w.h; common.h: SeqCount): live row r is agent slot aidx[r] of `out`
};

#define FPN_AG 64
#define FPN_LDA 400
#define FPN_LDS (128 * FPN_LDA * 2)

// the rows of level lv that this thread stages (issued for all three levels before the first one is used: one global round trip instead of three)
template <int C>
struct FpnRows { uint2 u[(FPN_AG * 3 * (C / 4) + 511) / 512]; };

template <int C>
__device__ __forceinline__ void fpn_load(const FpnP& p, int lv, int a0, FpnRows<C>& R, int tid, int sq_n, int sq_c0, int sq_c1, int sq_c2) {
  constexpr int V = FPN_AG * 3 * (C / 4), N = (V + 511) / 512;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int i = tid + k * 512;
    uint2 u = make_uint2(0u, 0u);
    if (i < V) {
      const int a = i / (3 * (C / 4)), rem = i - a * (3 * (C / 4));
      const int pz = rem / (C / 4), c4 = (rem - pz * (C / 4)) * 4;
      if (RIFT_SEQ_LIVE(a0 + a)) {
        if (p.ocb[lv]) u = *reinterpret_cast<const uint2*>(p.ocb[lv] + ((size_t)(a0 + a) * 3 + pz) * C + c4);
        else { const float4 v = *reinterpret_cast<const float4*>(p.oc[lv] + ((size_t)(a0 + a) * 3 + pz) * C + c4); u = pack_h4(v.x, v.y, v.z, v.w); }
      }
    }
    R.u[k] = u;
  }
}

template <int C, int KS>
__device__ __forceinline__ void fpn_lateral(const FpnP& p, int lv, const FpnRows<C>& R, unsigned short* At, f32x4 (&acc)[8][1], int tid, int wave,
                                            int l15, int l4) {
  PFrags<KS, 1> W;
  p_load_w<8, KS, 1>(W, p.wl[lv], 3 * C, 0, wave, l15, l4);
  // A tile: row (a, j) = [x[a][j], x[a][j+1], x[a][j+2] or 0], x = oc[lv][a] (3 x C)
  constexpr int V = FPN_AG * 3 * (C / 4), N = (V + 511) / 512;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int i = tid + k * 512;
    if (i < V) {
      const int a = i / (3 * (C / 4)), rem = i - a * (3 * (C / 4));
      const int pz = rem / (C / 4), c4 = (rem - pz * (C / 4)) * 4;
      *reinterpret_cast<uint2*>(At + (a * 2) * FPN_LDA + pz * C + c4) = R.u[k];
      if (pz >= 1) *reinterpret_cast<uint2*>(At + (a * 2 + 1) * FPN_LDA + (pz - 1) * C + c4) = R.u[k];
    }
  }
  for (int i = tid; i < FPN_AG * (C / 4); i += 512) {      // the third tap of position 1 looks past the sequence end
    const int a = i / (C / 4), c4 = (i - a * (C / 4)) * 4;
    *reinterpret_cast<uint2*>(At + (a * 2 + 1) * FPN_LDA + 2 * C + c4) = make_uint2(0u, 0u);
  }
  __syncthreads();
  p_zero(acc);
  p_mma<8, KS, 1>(acc, At, FPN_LDA, 0, W, l15, l4);
  __syncthreads();
}

__global__ __launch_bounds__(512) void fpn_tail_kernel(FpnP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* At = reinterpret_cast<unsigned short*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int a0 = blockIdx.x * FPN_AG;
  RIFT_SEQ_COUNT(p.cnt, p.nA);
  if (a0 >= sq_n) return;
  const int col = wave * 16 + l4 * 4;
  const bool odd = l15 & 1;                       // position 1 of its agent
  PFrags<8, 1> Wf;
  p_load_w<8, 8, 1>(Wf, p.wf, 256, 0, wave, l15, l4);
  f32x4 a2[8][1], a1[8][1], a0c[8][1];
  FpnRows<128> r2; FpnRows<64> r1; FpnRows<32> r0;
  fpn_load<128>(p, 2, a0, r2, tid, sq_n, sq_c0, sq_c1, sq_c2); fpn_load<64>(p, 1, a0, r1, tid, sq_n, sq_c0, sq_c1, sq_c2); fpn_load<32>(p, 0, a0, r0, tid, sq_n, sq_c0, sq_c1, sq_c2);
  fpn_lateral<128, 12>(p, 2, r2, At, a2, tid, wave, l15, l4);
  fpn_lateral<64, 6>(p, 1, r1, At, a1, tid, wave, l15, l4);
  fpn_lateral<32, 3>(p, 0, r0, At, a0c, tid, wave, l15, l4);
  const float4 b2 = *reinterpret_cast<const float4*>(p.bl[2] + col), b1 = *reinterpret_cast<const float4*>(p.bl[1] + col),
               b0 = *reinterpret_cast<const float4*>(p.bl[0] + col);
  const float bb2[4] = {b2.x, b2.y, b2.z, b2.w}, bb1[4] = {b1.x, b1.y, b1.z, b1.w}, bb0[4] = {b0.x, b0.y, b0.z, b0.w};
  // top-down merge (see fpn_merge_kernel): positions a = even row, b = odd row of the same agent
  unsigned short* Zt = At;                        // [64][272] bf16: [Z0 | Z1] per agent (the A tile is dead)
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    float z[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float l2 = a2[mt][0][r] + bb2[r], l1 = a1[mt][0][r] + bb1[r], l0 = a0c[mt][0][r] + bb0[r];
      const float l2o = dpp_f<0xB1>(l2);          // the other position of this agent
      const float l2b = odd ? l2 : l2o, l2a = odd ? l2o : l2;
      const float l1m = odd ? l1 + l2b : l1 + (0.25f * l2a + 0.75f * l2b);      // l1b' : l1a'
      const float l1o = dpp_f<0xB1>(l1m);
      const float l1b = odd ? l1m : l1o, l1a = odd ? l1o : l1m;
      z[r] = odd ? l0 + l1b : l0 + (0.25f * l1a + 0.75f * l1b);
    }
    const int agent = (mt * 16 + l15) >> 1;
    *reinterpret_cast<uint2*>(Zt + agent * 272 + (odd ? 128 : 0) + col) = pack_h4(z[0], z[1], z[2], z[3]);
  }
  __syncthreads();
  f32x4 acc[4][1];
  p_zero(acc);
  p_mma<4, 8, 1>(acc, Zt, 272, 0, Wf, l15, l4);
  const float4 bf4 = *reinterpret_cast<const float4*>(p.bf_ + col);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int agent = a0 + mt * 16 + l15;
    if (RIFT_SEQ_LIVE(agent))
      *reinterpret_cast<float4*>(p.out + (size_t)(p.aidx ? p.aidx[agent] : agent) * 128 + col) =
          make_float4(acc[mt][0][0] + bf4.x, acc[mt][0][1] + bf4.y, acc[mt][0][2] + bf4.z, acc[mt][0][3] + bf4.w);
  }
}

}  // namespace RIFT_NS
