"""Re-applies the round-4 experiment "decoder FFN shared between a tile's owner wave and an idle wave" to rift_amd/csrc/dec_w.hip (measured neutral: kept as a patch)."""
p='/root/repo/rift_amd/csrc/dec_w.hip'
s=open(p).read()
s=s.replace('#include "wp_stream.h"\n','#include "wp_stream.h"\n#include "wp_gemm_half.h"\n',1)
old='''  derive(0, 0);
  set_tiles(0, 0);
'''
new='''  derive(0, 0);
  set_tiles(0, 0);
  // ---- FFN shared between a tile's owner and an idle wave (R <= 6: six tiles on four SIMDs -- SIMDs 0 / 1 carry two working waves, waves
  // 6 / 7 on SIMDs 2 / 3 own no tile).  Owner wave 4 + p hands the FFN's operand rows (LayerNorm 4 output) to helper wave 6 + p; ffn.0 chunks
  // are split by output tiles (owner 0..3, helper 4..7 = the k-steps each then holds of ffn.3), ffn.3 by k-steps into two partial sums; the
  // helper returns its partial sum and its half of the last chunk's hidden rows one group early (the last ffn.3 group is the owner's alone),
  // so that every hand-over rides on a group barrier.  Mailbox = LDS nobody reads meanwhile: the tile's own rows of the residual buffer (dead
  // from the m2m read to the FFN write) and rows 72..95 (no reference line there).
  const bool pair_on = !DENSE && R <= 6 && !(p.dbg & 32);
  auto mb = [&](int unit, int ln) -> unsigned char* {        // 1 KiB unit `unit` (0..11) of this pair's mailbox + lane ln's 16 bytes
    const int pr = wv & 1;
    const uint32_t off = unit < 6 ? (uint32_t)((4 + pr) * 12 * XS * 4 + unit * 1024) : (uint32_t)(72 * XS * 4 + pr * 6144 + (unit - 6) * 1024);
    return smem_raw + 2 * 32768 + off + ln * 16;
  };
'''
assert old in s; s=s.replace(old,new,1)
old='''        layer_norm(res, xb, parL + DECW_L_LN4);
        DSITE(6);
        f32x4 acc2[8];
        init8(acc2, parL + DECW_L_BF2);
#pragma unroll 1
        for (int hc = 0; hc < 4; ++hc) {                        // ---- ffn.0 chunk -> ReLU, dropout -> ffn.3 partial
'''
new='''        layer_norm(res, xb, parL + DECW_L_LN4);
        DSITE(6);
        f32x4 acc2[8];
        init8(acc2, parL + DECW_L_BF2);
        if (!DENSE && pair_on && wv >= 4) {                     // ---- FFN shared with the helper wave
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<h16x8*>(mb(8 + k, lane)) = xb[k];
          half_ffn(li, acc2, xb, 0, true);
        } else
#pragma unroll 1
        for (int hc = 0; hc < 4; ++hc) {                        // ---- ffn.0 chunk -> ReLU, dropout -> ffn.3 partial
'''
assert old in s; s=s.replace(old,new,1)
old='''#pragma unroll 1
  for (int li = 0; li < 4; ++li) {
    {   // the lane / wave indices pass through opaque zeros once per layer'''
new='''  // one wave's half of a tile's FFN (see pair_on): nb = first ffn.0 output tile of this wave (owner 0, helper 4); the groups sit at positions
  // 12 .. 19 of the layer, ffn.0 chunks in ring slot 0, ffn.3 chunks in slot 1
  auto half_ffn = [&](int li, f32x4 (&acc2)[8], h16x8 (&xb)[4], const int nb, const bool own) {
    const int ln = lane;
    const uint32_t r0 = (uint32_t)(uintptr_t)ring + voff;
    h16x8 hb2[2];
#pragma unroll 1
    for (int hc = 0; hc < 4; ++hc) {
      f32x4 a4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float4 v = *reinterpret_cast<const float4*>(parL + DECW_L_BF1 + hc * 128 + (nb + j) * 16 + l4 * 4); a4[j] = (f32x4){v.x, v.y, v.z, v.w}; }
      bnd(li, 12 + 2 * hc);
      if (!own && hc == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) xb[k] = *reinterpret_cast<const h16x8*>(mb(8 + k, ln));
      }
      decw_gemm_nhalf(r0 + (uint32_t)nb * 1024u, xb, a4);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 a = a4[2 * q + u];
          v[u] = (f32x4){fmaxf(a[0], 0.f), fmaxf(a[1], 0.f), fmaxf(a[2], 0.f), fmaxf(a[3], 0.f)};
          if (DROP) v[u] *= keep4(2 * q + u);
        }
        hb2[q] = l0w_pack8(v[0], v[1]);
      }
      if (hc < 3) {
        bnd(li, 13 + 2 * hc);
        decw_gemm_khalf(r0 + 32768u + (uint32_t)(nb >> 1) * 8192u, hb2, acc2);
      }
    }
    if (!own) {                      // partial sum + the last chunk's hidden rows to the owner; the last group is a boundary only
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) *reinterpret_cast<f32x4*>(mb(nt, ln)) = acc2[nt];
      *reinterpret_cast<h16x8*>(mb(8, ln)) = hb2[0];
      *reinterpret_cast<h16x8*>(mb(9, ln)) = hb2[1];
      bnd(li, 19);
    } else {
      bnd(li, 19);
      h16x8 hb[4];
      hb[0] = hb2[0]; hb[1] = hb2[1];
      hb[2] = *reinterpret_cast<const h16x8*>(mb(8, ln)); hb[3] = *reinterpret_cast<const h16x8*>(mb(9, ln));
      gemm(1, hb, acc2);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc2[nt] += *reinterpret_cast<const f32x4*>(mb(nt, ln));
    }
  };

  // A helper wave runs a loop nest of its own (the same boundaries, its half of the FFN groups and nothing else): inside the working waves'
  // loop its accumulators and operand registers counted against theirs (255 VGPRs and spills, against 252 this way).
  if (!DENSE && pair_on && wv0 >= 6 && wv0 - 2 < R) {
#pragma unroll 1
    for (int li = 0; li < 4; ++li) {
      int zv, zs;
      asm volatile("v_mov_b32 %0, 0" : "=v"(zv));
      asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
      derive(zv, zs);
#pragma unroll 1
      for (int k = 0; k < 12; ++k) bnd(li, k);
      h16x8 xb[4];
      f32x4 acc2[8];
      zero8(acc2);
      DSITE(6);
      half_ffn(li, acc2, xb, 4, false);
    }
  } else
#pragma unroll 1
  for (int li = 0; li < 4; ++li) {
    {   // the lane / wave indices pass through opaque zeros once per layer'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)
