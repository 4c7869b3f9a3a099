// Wave-private, weight-streaming PointsEncoder pass B (see pe_w.h).  Its own translation unit: built with `-fno-honor-nans -mno-amdgpu-ieee`.
#include <hip/hip_fp16.h>
#include "common.h"
#include "pe_w.h"
#include "wp_stream.h"
#include "pe_w_gemm.h"
#include <type_traits>

namespace RIFT_NS {

// Weight image: PEW_FRAGS one-KiB fragments [lane][8] in consumption order.
//   W1  (fragments 0..7):   n-tile fr; element j of lane quarter l4 = feature k = 4 j + l4 (j < 3, k < Cin <= 12; the rest zero), so that
//                           every lane fetches three consecutive-quarter floats of its point row instead of one quarter fetching eight;
//   W2  (2 groups of 32):   f = ks * 8 + nt, output channel (8 g + nt) * 16 + l15, K-permuted (an n-tile pair of h1 is a k-step);
//   W3a (4 groups of 32):   f = ks * 4 + j, K-permuted likewise (an n-tile pair of f is a k-step), and the OUTPUT channels of a group
//                           permuted so that lane quarter l4 ends up with the 16 consecutive channels 64 q + 16 l4 + 4 j + r.
__global__ void pack_pew_kernel(PeWSrc s, unsigned short* __restrict__ img) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= PEW_FRAGS * 512) return;
  const int j = e & 7, lane = (e >> 3) & 63, fr = e >> 9, l15 = lane & 15, l4 = lane >> 4;
  float v;
  if (fr < 8) {
    const int k = 4 * j + l4;
    v = (j < 3 && k < s.Cin) ? s.w1[(fr * 16 + l15) * s.Cin + k] : 0.f;
  } else if (fr < 72) {
    const int g = (fr - 8) >> 5, f = (fr - 8) & 31, ks = f >> 3, nt = f & 7;
    v = s.w2[((g * 8 + nt) * 16 + l15) * 128 + l0w_chan(l4, j, 2 * ks)];
  } else {
    const int q = (fr - 72) >> 5, f = (fr - 72) & 31, ks = f >> 2, jn = f & 3;
    const int o = 64 * q + 16 * (l15 >> 2) + 4 * jn + (l15 & 3);
    v = s.w3[o * 512 + l0w_chan(l4, j, 2 * ks)];
  }
  img[e] = f2h(v);
}


// sum over the 16 lanes of a row of eight values at once: one fused v_add_f32_dpp per value and step (hipcc's own lowering of the same
// reduction was v_mov_b32_dpp + v_pk_add_f32: 1.5 instructions per value and step).  The four steps of a value are 8 instructions apart
// (a DPP source written by the previous VALU needs two wait states); `s_nop 1` covers the producer of the inputs.
__device__ __forceinline__ void pew_sum16x8(float (&v)[8]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}

// One encoder's rounds ri = wg, wg + G, ...
template <bool map20>   // 20 points per polyline (map polygons) or 120 (reference lines)
__device__ __forceinline__ void pe_w_body(const PeWSide& s0, const PeWP& p, const int wg, const int G) {
  const PeWSide* sp = &s0;
#define s (*sp)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t OFF_SCR = 65536, SCR_W = 4352, OFF_GPT = OFF_SCR + 8 * SCR_W, OFF_POOL = OFF_GPT + 16 * 260 * 4, OFF_PAR = OFF_POOL + 16 * 264 * 2;
  constexpr int GPS = 260, PLS = 264, TRS = 272;
  static_assert(OFF_PAR + 768 * 4 + 64 == PEW_LDS_BYTES, "LDS layout");
  unsigned char* ring = smem_raw;
  float* gpt = reinterpret_cast<float*>(smem_raw + OFF_GPT);            // [16][GPS] gp of the round's polylines ...
  uint32_t* pmax = reinterpret_cast<uint32_t*>(smem_raw + OFF_GPT);     // ... after the partial maxima [16 tiles][2 segments][128 channel pairs]
  unsigned short* pool = reinterpret_cast<unsigned short*>(smem_raw + OFF_POOL);   // [16][PLS] bf16 pooled operand
  float* par = reinterpret_cast<float*>(smem_raw + OFF_PAR);            // s1 128 | u1 = b1 s1 + t1 128 | b2 256 | b3 256
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* scr = smem_raw + OFF_SCR + wv * SCR_W;                 // wave-private: 16 x TRS transposition tile, later [2][256] statistics + count
  float* st = reinterpret_cast<float*>(scr);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem_raw);
  const uint32_t voff = (uint32_t)lane * 16u;
  // transposition read-back: dword `lane` of a scratch row = elements 2 (lane & 3), +1 of the operand fragment (pair lane >> 4, quarter (lane >> 2) & 3)
  const int pch = ((2 * (lane >> 4) + ((lane & 3) >> 1)) * 16 + 4 * ((lane >> 2) & 3) + 2 * (lane & 1)) >> 1;     // its channel pair within the half
  const int R = s.nrounds, Cin = s.Cin;
  constexpr int NPTS = map20 ? 20 : 120;
  auto pdiv = [&](int x) { return x / NPTS; };
  const unsigned char* img = reinterpret_cast<const unsigned char*>(s.img);
  const f32x4 Z = {0.f, 0.f, 0.f, 0.f};
#if RIFT_PEW_DIAG
  int tsn = 0;
#endif
#if RIFT_PEW_DIAG
#define PWTS() do { if (p.ts && blockIdx.x == 0 && tid == 0 && tsn < 120) p.ts[tsn++] = clock64(); } while (0)
#define PWDBG(bit) (p.dbg & (bit))
#else
#define PWTS() do { } while (0)
#define PWDBG(bit) 0
#endif

  // this workgroup's rounds: positions wg, wg + G, ... of the live list (train mode) or of 0 .. R - 1
  // (side b with packed rounds, pe_fused.h: pe_pack_lines_kernel -- round = record `pos` of the table, NL of them)
  const bool packed = !map20 && s.ptab != nullptr;
  const int NL = packed ? __builtin_amdgcn_readfirstlane(s.phdr[0]) : (s.live ? __builtin_amdgcn_readfirstlane(p.hdr[map20 ? 0 : 1]) : R);
  auto round_at = [&](int pos) { return pos >= NL ? R : ((s.live && !packed) ? __builtin_amdgcn_readfirstlane(s.live[pos]) : pos); };
  // tile descriptor of tile T of packed round ri: (first point row << 8) | (tile index within its line << 4) | line slot, or -1
  auto tdesc = [&](int ri, int T) { return __builtin_amdgcn_readfirstlane(s.ptab[(size_t)ri * 32 + T]); };
  int li = wg;
  auto dma = [&](const unsigned char* src, uint32_t dst, int nfrag) {
    if (PWDBG(2)) return;                     // (diagnostic: no weight stream -- compute on whatever the ring holds)
    decw_dma_share(src, voff, lds0 + dst, nfrag, wv, 8);
  };
  auto sync = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); PWTS(); };
  // statistics of a finished round: the eight waves' partials are added to this thread's running sum (thread = kind * 256 + channel); the
  // workgroup writes ONE partial at its end (fp32 over its <= ~6 rounds x 240 rows; bn_finalize_t_kernel adds the partials in fp64).
  // Called behind a barrier that follows the round's last group.
  float stat_acc = 0.f;
  int cnt_acc = 0;
  auto finish_stats = [&]() {
    if (!p.do_stats || PWDBG(4)) return;    // (diagnostic 4: no statistics)
#pragma unroll
    for (int w = 0; w < 8; ++w) stat_acc += *reinterpret_cast<const float*>(smem_raw + OFF_SCR + w * SCR_W + tid * 4);
    if (tid < 8) cnt_acc += *reinterpret_cast<const int*>(smem_raw + OFF_SCR + tid * SCR_W + 2048);
  };
  // the point rows of this wave's two tiles T = 2 wv + mt in round ri: validity byte (0xff = no such row) and features, raw
  auto load_rows = [&](int ri, unsigned (&vb)[2], float (&xv)[2][3]) {
    const int row0 = ri * PEW_ROUND_ROWS, nex = min(PEW_ROUND_ROWS, s.rows - row0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      int r = 32 * wv + 16 * mt + l15, rbase = row0;
      bool ex = r < nex;
      if (!map20 && packed) {                                        // this tile's sixteen rows: points 16 k .. of its line (the eighth tile ends at point 119)
        const int d = tdesc(ri, 2 * wv + mt);
        rbase = d >> 8; r = l15;
        ex = d >= 0 && 16 * ((d >> 4) & 15) + l15 < 120;
        if (d < 0) rbase = 0;
      }
      const int rr = min(rbase + r, s.rows - 1);                     // every lane loads (clamped address), then selects: no branch, no wait between the loads
      const unsigned v = s.valid[rr];
      float t[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) t[j] = s.F[min(rr * Cin + 4 * j + l4, s.rows * Cin - 1)];
      vb[mt] = ex ? v : 0xffu;
#pragma unroll
      for (int j = 0; j < 3; ++j) xv[mt][j] = (ex && 4 * j + l4 < Cin) ? t[j] : 0.f;
    }
  };

  int gc = 0;                                 // groups consumed so far: group gc sits in ring slot gc & 1
  int ri = round_at(li);
  if (ri >= R) {
    if (p.do_stats) { s.part2[(size_t)tid * G + wg] = 0.f; if (tid == 0) s.cnt2[wg] = 0; }
    return;
  }
  dma(img, 0, 8);
  for (int i = tid; i < 768; i += 512) {      // parameter block (published by the first group barrier)
    float v;
    if (i < 128) v = s.s1[i];
    else if (i < 256) v = s.b1[i - 128] * s.s1[i - 128] + s.t1[i - 128];
    else if (i < 512) v = s.b2[i - 256];
    else v = s.b3[i - 512];
    par[i] = v;
  }
  auto to_operand = [&](const float (&xv)[2][3], h16x8 (&xb)[2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      xb[mt] = l0w_from_u2(make_uint2(pack_h2(xv[mt][0], xv[mt][1]), pack_h2(xv[mt][2], 0.f)), make_uint2(0u, 0u));
  };
  unsigned vb[2];
  h16x8 xb[2];
  {
    float xv[2][3];
    load_rows(ri, vb, xv);
    to_operand(xv, xb);
  }
  int prev_ri = -1;
  // g of a group leaves one boundary late (a store issued just ahead of a boundary would be waited for there: vmcnt counts stores too)
  uint32_t hold[2][8];
  bool hex[2] = {false, false};               // rows of the held tiles are valid points
  int hrow[2] = {0, 0};                       // first point row of the held tiles
  auto store_hold = [&](int q) {
    if (PWDBG(1)) return;                     // (diagnostic: no g stores)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (hex[mt]) {
        unsigned short* dst = s.Fmid + (size_t)(hrow[mt] + l15) * 256 + 64 * q + 16 * l4;
        *reinterpret_cast<uint4*>(dst) = make_uint4(hold[mt][0], hold[mt][1], hold[mt][2], hold[mt][3]);
        *reinterpret_cast<uint4*>(dst + 8) = make_uint4(hold[mt][4], hold[mt][5], hold[mt][6], hold[mt][7]);
      }
    }
  };
  bool pend = false;                          // the last group of the previous round is still held

  while (ri < R) {
    {   // the side's fields are re-read from the kernel arguments where a round uses them (an opaque zero keeps them from being hoisted
        // out of the loop into ~40 long-lived SGPRs / VGPR pairs, which is what spilled)
      int z;
      asm volatile("s_mov_b32 %0, 0" : "=s"(z));
      sp = &s0 + z;
    }
    const int row0 = ri * PEW_ROUND_ROWS;
    const int nex = min(PEW_ROUND_ROWS, s.rows - row0);            // rows of the round that exist
    // row flags: 0 = no such row, 1 = valid point, 2 = invalid point (a zero row that takes part in the max)
    int fl[2], poly[2];
    int td[2] = {-1, -1};                      // (packed) descriptors of this wave's two tiles
    float okf[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      fl[mt] = vb[mt] == 0xffu ? 0 : (vb[mt] ? 1 : 2);
      okf[mt] = fl[mt] == 1 ? 1.0f : 0.0f;
      poly[mt] = pdiv(32 * wv + 16 * mt + l15);
      if (!map20 && packed) { td[mt] = tdesc(ri, 2 * wv + mt); poly[mt] = td[mt] & 15; }
    }
    const bool wact = __builtin_amdgcn_ballot_w64(fl[0] == 1 || fl[1] == 1) != 0;        // a valid point among this wave's 32 rows
    const int nval = __builtin_popcountll(__builtin_amdgcn_ballot_w64(fl[0] == 1 && l4 == 0)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(fl[1] == 1 && l4 == 0));
    const int nri = round_at(li + G);

    // group k of the round has landed for everybody and nobody reads the other slot any more: request group k + 1 into it
    // (fragment offsets: W1 0 | W2 8, 40 | W3a 72, 104, 136, 168; behind the last group the next round's W1)
    auto nothing = [] {};
    auto boundary = [&](int k, auto&& first) {
      sync();
      first();                               // work that must not queue behind the request below (a scratch reload would wait for the DMA)
      if (k < 6) dma(img + (8 + 32 * k) * 1024, ((gc + 1) & 1) * 32768u, 32);
      else if (nri < R) dma(img, ((gc + 1) & 1) * 32768u, 8);
    };
    auto load_wb = [&](h16x8 (&wb)[16], int k0, int k1) {       // W3b fragments of this wave's gp columns (n-tiles 2 wv, 2 wv + 1), straight from L2
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
          if (ks >= k0 && ks < k1) wb[u * 8 + ks] = fm_load(s.w3b, 256, (2 * wv + u) * 16, ks * 32, lane);
    };
    // partial maxima -> pooled operand; gp = pooled W3b^T + b3 of the round's polylines -> gpt (every wave: 32 of the 256 columns)
    auto pooled_gp = [&](const h16x8 (&wb)[16]) {
      const int npoly = pdiv(nex);                                   // polylines of this round that exist (whole ones)
#pragma unroll
      for (int u = 0; u < 2; ++u) {                                // wave wv: polylines 2 wv, 2 wv + 1 (uniform tile loop); a lane: channel pairs lane, lane + 64
        const int pl = 2 * wv + u;
        float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
        if (!map20 && packed) {                                     // line slot pl of the packed round: tiles t0 .. t1, each wholly this line's
          const int li = __builtin_amdgcn_readfirstlane(s.ptab[(size_t)ri * 32 + 16 + pl]);
          if (li & (1 << 17)) {
            a0 = b0 = a1 = b1 = (li & (1 << 16)) ? 0.f : -INFINITY;   // dropped tiles = zero rows of invalid points in the reference's max
            for (int T = li & 0xff; T <= ((li >> 8) & 0xff); ++T) {
              const uint32_t* src = pmax + (T * 2) * 128;
              const uint32_t w0 = src[lane], w1 = src[lane + 64];
              a0 = fmaxf(a0, h_lo(w0)); b0 = fmaxf(b0, h_hi(w0)); a1 = fmaxf(a1, h_lo(w1)); b1 = fmaxf(b1, h_hi(w1));
            }
          }
        } else if (pl < npoly) {
          a0 = b0 = a1 = b1 = -INFINITY;
          const int r0 = pl * NPTS, T1 = (r0 + NPTS - 1) >> 4;
          for (int T = r0 >> 4; T <= T1; ++T) {                    // a tile that starts ahead of the polyline holds it as its second segment
            const uint32_t* src = pmax + (T * 2 + (16 * T < r0 ? 1 : 0)) * 128;
            const uint32_t w0 = src[lane], w1 = src[lane + 64];
            a0 = fmaxf(a0, h_lo(w0)); b0 = fmaxf(b0, h_hi(w0)); a1 = fmaxf(a1, h_lo(w1)); b1 = fmaxf(b1, h_hi(w1));
          }
        }
        reinterpret_cast<uint32_t*>(pool)[pl * (PLS / 2) + lane] = h_pair_exact(a0, b0);
        reinterpret_cast<uint32_t*>(pool)[pl * (PLS / 2) + lane + 64] = h_pair_exact(a1, b1);
      }
      lds_barrier();                   // (not __syncthreads: that would also wait for the group just requested)
      f32x4 ga[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4 b = *reinterpret_cast<const float4*>(par + 512 + (2 * wv + u) * 16 + l4 * 4);
        ga[u] = (f32x4){b.x, b.y, b.z, b.w};
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const h16x8 pb = *reinterpret_cast<const h16x8*>(pool + l15 * PLS + ks * 32 + l4 * 8);
#pragma unroll
        for (int u = 0; u < 2; ++u) ga[u] = mfma_h(wb[u * 8 + ks], pb, ga[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) *reinterpret_cast<float4*>(gpt + l15 * GPS + (2 * wv + u) * 16 + l4 * 4) = make_float4(ga[u][0], ga[u][1], ga[u][2], ga[u][3]);
      lds_barrier();                   // (not __syncthreads: that would also wait for the group just requested)
    };
    // rows [0, B) of a tile belong to its first polyline, rows [B, 16) to the next one, B in {4, 8, 12, 16} (rows that do not exist only
    // ever fall into polylines that do not exist: a round ends on a polyline boundary): maxima of the four 4-row blocks, combined per segment
    auto tile_max = [&](int T, int h, int d) {
      // (packed: the whole tile is ONE line's -- B = its existing rows: 8 in a line's eighth tile (points 112..119), else 16; blocks past B
      // fall into the unused second segment)
      const int B = (!map20 && packed) ? (((d >> 4) & 15) == 7 ? 8 : 16) : min(16, (pdiv(16 * T) + 1) * NPTS - 16 * T);
      uint32_t w[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) w[r] = *reinterpret_cast<const uint32_t*>(scr + r * TRS + lane * 4);
      float ba[4], bb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ba[k] = fmaxf(fmaxf(h_lo(w[4 * k]), h_lo(w[4 * k + 1])), fmaxf(h_lo(w[4 * k + 2]), h_lo(w[4 * k + 3])));
        bb[k] = fmaxf(fmaxf(h_hi(w[4 * k]), h_hi(w[4 * k + 1])), fmaxf(h_hi(w[4 * k + 2]), h_hi(w[4 * k + 3])));
      }
      float m0a = ba[0], m0b = bb[0], m1a = -INFINITY, m1b = -INFINITY;
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        if (4 * k < B) { m0a = fmaxf(m0a, ba[k]); m0b = fmaxf(m0b, bb[k]); }
        else { m1a = fmaxf(m1a, ba[k]); m1b = fmaxf(m1b, bb[k]); }
      }
      pmax[(T * 2 + 0) * 128 + 64 * h + pch] = h_pair_exact(m0a, m0b);
      pmax[(T * 2 + 1) * 128 + 64 * h + pch] = h_pair_exact(m1a, m1b);
    };

    if (wact) {
      // ---- group 0: h1 = relu(bn1(x W1^T + b1)) as the k-steps of W2
      boundary(0, [&] { if (prev_ri >= 0) finish_stats(); if (pend) { store_hold(3); pend = false; } });
      PWTS();
      PWTS();
      h16x8 hb[2][4];
      {
        f32x4 acc[2][8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const h16x8 w = *reinterpret_cast<const h16x8*>(ring + (gc & 1) * 32768 + nt * 1024 + lane * 16);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = mfma_h(w, xb[mt], Z, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          f32x4 sc[2], sh[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const float4 a = *reinterpret_cast<const float4*>(par + (2 * ks + u) * 16 + l4 * 4), b = *reinterpret_cast<const float4*>(par + 128 + (2 * ks + u) * 16 + l4 * 4);
            sc[u] = (f32x4){a.x, a.y, a.z, a.w}; sh[u] = (f32x4){b.x, b.y, b.z, b.w};
          }
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            f32x4 y[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              y[u] = acc[mt][2 * ks + u] * sc[u] + sh[u];
              y[u] = (f32x4){fmaxf(y[u][0], 0.f), fmaxf(y[u][1], 0.f), fmaxf(y[u][2], 0.f), fmaxf(y[u][3], 0.f)};
            }
            hb[mt][ks] = l0w_pack8(y[0], y[1]);
          }
        }
      }
      PWTS();
      ++gc;

      // ---- groups 1, 2: f = h1 W2^T + b2 (invalid rows zero) as the k-steps of W3a; per-(tile, segment) maxima of the bf16 values
      h16x8 fb[2][8];
      h16x8 wb[16];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        boundary(1 + h, nothing);
        {
          f32x4 c0[8], c1[8];
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) {
            const float4 b = *reinterpret_cast<const float4*>(par + 256 + (8 * h + nt) * 16 + l4 * 4);
            c0[nt] = c1[nt] = (f32x4){b.x, b.y, b.z, b.w};
          }
          pew_gemm_k4n8((uint32_t)(uintptr_t)ring + (uint32_t)(gc & 1) * 32768u + voff, hb[0], hb[1], c0, c1);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const bool ok = fl[mt] == 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 a = mt ? c1[2 * q] : c0[2 * q], b = mt ? c1[2 * q + 1] : c0[2 * q + 1];
              uint2 w0 = pack_h4(a[0], a[1], a[2], a[3]), w1 = pack_h4(b[0], b[1], b[2], b[3]);
              if (!ok) { w0 = make_uint2(0u, 0u); w1 = w0; }
              fb[mt][4 * h + q] = l0w_from_u2(w0, w1);
            }
          }
        }
        if (h == 1) load_wb(wb, 0, 8);          // (behind the GEMM: its accumulators are dead, the fragments land under the max below)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
          for (int q = 0; q < 4; ++q)          // n-tiles 2 q (elements 0..3) and 2 q + 1 (elements 4..7) of this half, row l15
            *reinterpret_cast<h16x8*>(scr + l15 * TRS + q * 64 + l4 * 16) = fb[mt][4 * h + q];
          // rows in-lane: lane t owns two channels of the 16 rows just written (same wave: LDS operations stay in order)
          tile_max(2 * wv + mt, h, td[mt]);
        }
        ++gc;
      }

      // ---- groups 3..6: g = f W3a^T + gp, 64 output channels per group; ahead of the first one the pooled operand and gp
      unsigned nvb[2];
      float nxv[2][3];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // (the held stores go out AHEAD of the request: behind it they queue in the texture path after the group's 32 requests)
        boundary(3 + q, [&] { if (q == 1 && nri < R) { to_operand(nxv, xb); vb[0] = nvb[0]; vb[1] = nvb[1]; } if (q > 0) store_hold(q - 1); });
        if (q == 0) {                                          // (g of an invalid point is never read: pass C masks the row)
          hex[0] = fl[0] == 1; hex[1] = fl[1] == 1;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) hrow[mt] = (!map20 && packed) ? (td[mt] >> 8) : row0 + 32 * wv + 16 * mt;
          pooled_gp(wb);
        }
        if (q == 0 && nri < R) load_rows(nri, nvb, nxv);      // the next round's rows: in flight under this group
        f32x4 c0[4], c1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 g0 = *reinterpret_cast<const float4*>(gpt + poly[0] * GPS + 64 * q + 16 * l4 + 4 * j);
          const float4 g1 = *reinterpret_cast<const float4*>(gpt + poly[1] * GPS + 64 * q + 16 * l4 + 4 * j);
          c0[j] = (f32x4){g0.x, g0.y, g0.z, g0.w}; c1[j] = (f32x4){g1.x, g1.y, g1.z, g1.w};
        }
        PWTS();
        pew_gemm_k8n4((uint32_t)(uintptr_t)ring + (uint32_t)(gc & 1) * 32768u + voff, fb[0], fb[1], c0, c1);
        PWTS();
        if (p.do_stats) {
#pragma unroll
          for (int jj = 0; jj < 4; jj += 2) {
            float sm[8], sq[8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const f32x4 v0 = c0[jj + u] * okf[0], v1 = c1[jj + u] * okf[1];
              const f32x4 a = v0 + v1, b = v0 * v0 + v1 * v1;
#pragma unroll
              for (int r = 0; r < 4; ++r) { sm[4 * u + r] = a[r]; sq[4 * u + r] = b[r]; }
            }
            pew_sum16x8(sm);
            pew_sum16x8(sq);
            if (l15 == 0) {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                *reinterpret_cast<float4*>(st + 64 * q + 16 * l4 + 4 * (jj + u)) = make_float4(sm[4 * u], sm[4 * u + 1], sm[4 * u + 2], sm[4 * u + 3]);
                *reinterpret_cast<float4*>(st + 256 + 64 * q + 16 * l4 + 4 * (jj + u)) = make_float4(sq[4 * u], sq[4 * u + 1], sq[4 * u + 2], sq[4 * u + 3]);
              }
            }
          }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 v = mt ? c1[j] : c0[j];
            const __half2 lo = __floats2half2_rn(v[0], v[1]), hi = __floats2half2_rn(v[2], v[3]);
            hold[mt][2 * j] = *reinterpret_cast<const uint32_t*>(&lo); hold[mt][2 * j + 1] = *reinterpret_cast<const uint32_t*>(&hi);
          }
        }
        PWTS();
        ++gc;
      }
      pend = true;
    } else {
      // ---- no valid point among this wave's rows: it keeps the barriers, carries its share of the stream and of gp, and its
      // existing rows are zero rows (they take part in the max with 0; their g is never read)
      boundary(0, [&] { if (prev_ri >= 0) finish_stats(); });
      if (pend) { store_hold(3); pend = false; }
      ++gc;
      boundary(1, nothing);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int T = 2 * wv + mt;
        int ext = max(0, min(16, nex - 16 * T)), B = min(16, (pdiv(16 * T) + 1) * NPTS - 16 * T);
        if (!map20 && packed) { ext = td[mt] >= 0 ? 16 : 0; B = 16; }     // a tile in use whose points are all invalid: zero rows; an unused tile: no line reads it
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          pmax[(T * 2 + 0) * 128 + 64 * u + lane] = ext > 0 ? 0u : RIFT_H_NEG_INF2;
          pmax[(T * 2 + 1) * 128 + 64 * u + lane] = ext > B ? 0u : RIFT_H_NEG_INF2;
        }
      }
      ++gc;
      boundary(2, nothing);
      h16x8 wb[16];
      load_wb(wb, 0, 8);
      ++gc;
      boundary(3, nothing);
      pooled_gp(wb);
      if (p.do_stats) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(st + (i * 64 + lane) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      ++gc;
      boundary(4, nothing);
      float nxv[2][3];
      if (nri < R) load_rows(nri, vb, nxv);
      ++gc;
      boundary(5, nothing);
      if (nri < R) to_operand(nxv, xb);
      ++gc;
      boundary(6, nothing); ++gc;
    }
    if (lane == 0) *reinterpret_cast<int*>(scr + 2048) = nval;
    prev_ri = ri;
    ri = nri; li += G;
  }
  __syncthreads();
  if (pend) store_hold(3);
  if (prev_ri >= 0) finish_stats();
  if (p.do_stats) {
    s.part2[(size_t)tid * G + wg] = stat_acc;
    int n = cnt_acc;                           // lanes 0..7 of wave 0 hold the eight waves' counts, every other lane 0
    n += __shfl_xor(n, 1, 64); n += __shfl_xor(n, 2, 64); n += __shfl_xor(n, 4, 64);
    if (tid == 0) s.cnt2[wg] = n;
  }
#undef PWTS
#undef PWDBG
#undef s
}

// workgroups [0, ga) walk the map encoder's rounds, the others the reference-line encoder's
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pe_w_kernel(PeWP p) {
  const int ga = p.hdr ? __builtin_amdgcn_readfirstlane(p.hdr[2]) : p.ga;
  if ((int)blockIdx.x < ga) pe_w_body<true>(p.a, p, blockIdx.x, ga);
  else pe_w_body<false>(p.b, p, blockIdx.x - ga, gridDim.x - ga);
}

int pew_set_attributes() {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(pe_w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PEW_LDS_BYTES);
}

void pew_pack(const PeWSrc& src, unsigned short* img, hipStream_t stream) {
  hipLaunchKernelGGL(pack_pew_kernel, dim3((PEW_FRAGS * 512 + 255) / 256), dim3(256), 0, stream, src, img);
}

void pew_split(int ra, int rb, int* grid, int* ga) {
  int g = *grid < ra + rb ? *grid : ra + rb;
  int a = ra + rb > 0 ? (int)(((long long)g * ra + (ra + rb) / 2) / (ra + rb)) : 0;
  if (ra > 0 && a < 1) a = 1;
  if (rb > 0 && a > g - 1) a = g - 1;
  if (rb == 0) a = g;
  if (a > ra) a = ra;
  *grid = g; *ga = a;
}

void pew_launch(const PeWP& p, int grid, hipStream_t stream) {
  if (grid <= 0) return;
  hipLaunchKernelGGL(pe_w_kernel, dim3(grid), dim3(512), PEW_LDS_BYTES, stream, p);
}

}  // namespace RIFT_NS
