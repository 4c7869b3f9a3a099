// Shared pieces of the wave-private, weight-streaming kernels (dec_w.hip, nat_l2w.hip): the LDS-DMA fragment load and the hand-scheduled
// 32-fragment group GEMM.  Device inline functions only (each kernel is its own translation unit).
#pragma once
#include "common.h"

namespace RIFT_NS {

// one 1 KiB fragment, global -> LDS, no staging registers: lane i's 16 bytes land at lds_dst + 16 i.  `src` and `lds_dst` are wave-uniform.
__device__ __forceinline__ void decw_glds(const void* src, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  // the scalar operands are pinned to SGPRs by hand: an "s" constraint on a value hipcc's divergence analysis does not prove uniform
  // is NOT legalised with a readfirstlane, it reaches the assembler as a VGPR
  const uint64_t a = reinterpret_cast<uint64_t>(src);
  const uint64_t sa = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t sd = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);
  // `s_nop 4` first: the SGPR operands may come straight out of v_readfirstlane, and a VALU-written SGPR needs five wait states before a
  // VMEM instruction reads it as its base -- hipcc pads its own instructions, not the inside of an asm statement (builds with the
  // readfirstlane four instructions ahead of the load exist)
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sa), "s"(sd) : "memory");
}

// four CONSECUTIVE fragments behind one M0 write: the instruction offset moves the global and the LDS address alike (checked byte for byte
// in tools/ubench/lds_dma_rate.hip).  Writing M0 ahead of every fragment holds a wave at ~210 cycles per request; this form reaches
// 58 B/cycle/CU with two groups in flight against 39.
__device__ __forceinline__ void decw_glds4(const void* src, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  const uint64_t a = reinterpret_cast<uint64_t>(src);
  const uint64_t sa = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t sd = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sa), "s"(sd) : "memory");
}
// This loader's share of an nfrag-fragment group (nfrag >= 4), loader lw of ln: consecutive fragments in runs of four; a run that would
// pass the end of the group starts at nfrag - 4 instead (re-requesting a neighbour's fragments is harmless and keeps this to one loop).
__device__ __forceinline__ void decw_dma_share(const unsigned char* src, uint32_t voff, uint32_t lds_dst, int nfrag, int lw, int ln) {
  const int per = max(4, (nfrag + ln - 1) / ln);
  const int f1 = min(lw * per + per, nfrag);
#pragma unroll 1
  for (int f = lw * per; f < f1; f += 4) {
    const int ff = min(f, nfrag - 4);
    decw_glds4(src + (size_t)ff * 1024, voff, lds_dst + (uint32_t)ff * 1024u);
  }
}

// One group GEMM (16 rows x K = 128 against the 32 fragments at LDS address `addr` + 1024 f) as a hand-scheduled stream: the 8 fragments of
// k-step ks + 1 are requested under the 8 MFMAs of k-step ks (lgkmcnt(8) = the fragment 8 requests back has landed), so an MFMA never waits a
// full LDS round trip; hipcc's own schedule of the same loop kept 1-2 reads in flight.  PLAIN = activations as the A operand (V^T tiles).
// HID: the weight fragments and x are operands of the MLP hidden layer (common.h: RIFT_MFMA_HID_ASM -- fp16 words in the bf16 build)
#define DECW_GEMM_SWAPPED_ASM(MN) \
        "ds_read_b128 %[w0], %[a] offset:0\n\t" \
        "ds_read_b128 %[w1], %[a] offset:1024\n\t" \
        "ds_read_b128 %[w2], %[a] offset:2048\n\t" \
        "ds_read_b128 %[w3], %[a] offset:3072\n\t" \
        "ds_read_b128 %[w4], %[a] offset:4096\n\t" \
        "ds_read_b128 %[w5], %[a] offset:5120\n\t" \
        "ds_read_b128 %[w6], %[a] offset:6144\n\t" \
        "ds_read_b128 %[w7], %[a] offset:7168\n\t" \
        "ds_read_b128 %[w8], %[a] offset:8192\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c0], %[w0], %[x0], %[c0]\n\t" \
        "ds_read_b128 %[w9], %[a] offset:9216\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c1], %[w1], %[x0], %[c1]\n\t" \
        "ds_read_b128 %[w10], %[a] offset:10240\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c2], %[w2], %[x0], %[c2]\n\t" \
        "ds_read_b128 %[w11], %[a] offset:11264\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c3], %[w3], %[x0], %[c3]\n\t" \
        "ds_read_b128 %[w12], %[a] offset:12288\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c4], %[w4], %[x0], %[c4]\n\t" \
        "ds_read_b128 %[w13], %[a] offset:13312\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c5], %[w5], %[x0], %[c5]\n\t" \
        "ds_read_b128 %[w14], %[a] offset:14336\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c6], %[w6], %[x0], %[c6]\n\t" \
        "ds_read_b128 %[w15], %[a] offset:15360\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c7], %[w7], %[x0], %[c7]\n\t" \
        "ds_read_b128 %[w0], %[a] offset:16384\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c0], %[w8], %[x1], %[c0]\n\t" \
        "ds_read_b128 %[w1], %[a] offset:17408\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c1], %[w9], %[x1], %[c1]\n\t" \
        "ds_read_b128 %[w2], %[a] offset:18432\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c2], %[w10], %[x1], %[c2]\n\t" \
        "ds_read_b128 %[w3], %[a] offset:19456\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c3], %[w11], %[x1], %[c3]\n\t" \
        "ds_read_b128 %[w4], %[a] offset:20480\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c4], %[w12], %[x1], %[c4]\n\t" \
        "ds_read_b128 %[w5], %[a] offset:21504\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c5], %[w13], %[x1], %[c5]\n\t" \
        "ds_read_b128 %[w6], %[a] offset:22528\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c6], %[w14], %[x1], %[c6]\n\t" \
        "ds_read_b128 %[w7], %[a] offset:23552\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c7], %[w15], %[x1], %[c7]\n\t" \
        "ds_read_b128 %[w8], %[a] offset:24576\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c0], %[w0], %[x2], %[c0]\n\t" \
        "ds_read_b128 %[w9], %[a] offset:25600\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c1], %[w1], %[x2], %[c1]\n\t" \
        "ds_read_b128 %[w10], %[a] offset:26624\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c2], %[w2], %[x2], %[c2]\n\t" \
        "ds_read_b128 %[w11], %[a] offset:27648\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c3], %[w3], %[x2], %[c3]\n\t" \
        "ds_read_b128 %[w12], %[a] offset:28672\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c4], %[w4], %[x2], %[c4]\n\t" \
        "ds_read_b128 %[w13], %[a] offset:29696\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c5], %[w5], %[x2], %[c5]\n\t" \
        "ds_read_b128 %[w14], %[a] offset:30720\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c6], %[w6], %[x2], %[c6]\n\t" \
        "ds_read_b128 %[w15], %[a] offset:31744\n\t" \
        "s_waitcnt lgkmcnt(8)\n\t" \
        MN " %[c7], %[w7], %[x2], %[c7]\n\t" \
        "s_waitcnt lgkmcnt(7)\n\t" \
        MN " %[c0], %[w8], %[x3], %[c0]\n\t" \
        "s_waitcnt lgkmcnt(6)\n\t" \
        MN " %[c1], %[w9], %[x3], %[c1]\n\t" \
        "s_waitcnt lgkmcnt(5)\n\t" \
        MN " %[c2], %[w10], %[x3], %[c2]\n\t" \
        "s_waitcnt lgkmcnt(4)\n\t" \
        MN " %[c3], %[w11], %[x3], %[c3]\n\t" \
        "s_waitcnt lgkmcnt(3)\n\t" \
        MN " %[c4], %[w12], %[x3], %[c4]\n\t" \
        "s_waitcnt lgkmcnt(2)\n\t" \
        MN " %[c5], %[w13], %[x3], %[c5]\n\t" \
        "s_waitcnt lgkmcnt(1)\n\t" \
        MN " %[c6], %[w14], %[x3], %[c6]\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        MN " %[c7], %[w15], %[x3], %[c7]\n\t" \
        "s_nop 15\n\t"
template <bool PLAIN, bool HID = false>
__device__ __forceinline__ void decw_gemm(uint32_t addr, const h16x8 (&x)[4], f32x4 (&c)[8]) {
  h16x8 w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11, w12, w13, w14, w15;
  if (PLAIN) {
    asm volatile(
        "ds_read_b128 %[w0], %[a] offset:0\n\t"
        "ds_read_b128 %[w1], %[a] offset:1024\n\t"
        "ds_read_b128 %[w2], %[a] offset:2048\n\t"
        "ds_read_b128 %[w3], %[a] offset:3072\n\t"
        "ds_read_b128 %[w4], %[a] offset:4096\n\t"
        "ds_read_b128 %[w5], %[a] offset:5120\n\t"
        "ds_read_b128 %[w6], %[a] offset:6144\n\t"
        "ds_read_b128 %[w7], %[a] offset:7168\n\t"
        "ds_read_b128 %[w8], %[a] offset:8192\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c0], %[x0], %[w0], %[c0]\n\t"
        "ds_read_b128 %[w9], %[a] offset:9216\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c1], %[x0], %[w1], %[c1]\n\t"
        "ds_read_b128 %[w10], %[a] offset:10240\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c2], %[x0], %[w2], %[c2]\n\t"
        "ds_read_b128 %[w11], %[a] offset:11264\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c3], %[x0], %[w3], %[c3]\n\t"
        "ds_read_b128 %[w12], %[a] offset:12288\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c4], %[x0], %[w4], %[c4]\n\t"
        "ds_read_b128 %[w13], %[a] offset:13312\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c5], %[x0], %[w5], %[c5]\n\t"
        "ds_read_b128 %[w14], %[a] offset:14336\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c6], %[x0], %[w6], %[c6]\n\t"
        "ds_read_b128 %[w15], %[a] offset:15360\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c7], %[x0], %[w7], %[c7]\n\t"
        "ds_read_b128 %[w0], %[a] offset:16384\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c0], %[x1], %[w8], %[c0]\n\t"
        "ds_read_b128 %[w1], %[a] offset:17408\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c1], %[x1], %[w9], %[c1]\n\t"
        "ds_read_b128 %[w2], %[a] offset:18432\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c2], %[x1], %[w10], %[c2]\n\t"
        "ds_read_b128 %[w3], %[a] offset:19456\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c3], %[x1], %[w11], %[c3]\n\t"
        "ds_read_b128 %[w4], %[a] offset:20480\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c4], %[x1], %[w12], %[c4]\n\t"
        "ds_read_b128 %[w5], %[a] offset:21504\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c5], %[x1], %[w13], %[c5]\n\t"
        "ds_read_b128 %[w6], %[a] offset:22528\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c6], %[x1], %[w14], %[c6]\n\t"
        "ds_read_b128 %[w7], %[a] offset:23552\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c7], %[x1], %[w15], %[c7]\n\t"
        "ds_read_b128 %[w8], %[a] offset:24576\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c0], %[x2], %[w0], %[c0]\n\t"
        "ds_read_b128 %[w9], %[a] offset:25600\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c1], %[x2], %[w1], %[c1]\n\t"
        "ds_read_b128 %[w10], %[a] offset:26624\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c2], %[x2], %[w2], %[c2]\n\t"
        "ds_read_b128 %[w11], %[a] offset:27648\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c3], %[x2], %[w3], %[c3]\n\t"
        "ds_read_b128 %[w12], %[a] offset:28672\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c4], %[x2], %[w4], %[c4]\n\t"
        "ds_read_b128 %[w13], %[a] offset:29696\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c5], %[x2], %[w5], %[c5]\n\t"
        "ds_read_b128 %[w14], %[a] offset:30720\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c6], %[x2], %[w6], %[c6]\n\t"
        "ds_read_b128 %[w15], %[a] offset:31744\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        RIFT_MFMA_H_ASM " %[c7], %[x2], %[w7], %[c7]\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        RIFT_MFMA_H_ASM " %[c0], %[x3], %[w8], %[c0]\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        RIFT_MFMA_H_ASM " %[c1], %[x3], %[w9], %[c1]\n\t"
        "s_waitcnt lgkmcnt(5)\n\t"
        RIFT_MFMA_H_ASM " %[c2], %[x3], %[w10], %[c2]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        RIFT_MFMA_H_ASM " %[c3], %[x3], %[w11], %[c3]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        RIFT_MFMA_H_ASM " %[c4], %[x3], %[w12], %[c4]\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        RIFT_MFMA_H_ASM " %[c5], %[x3], %[w13], %[c5]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        RIFT_MFMA_H_ASM " %[c6], %[x3], %[w14], %[c6]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        RIFT_MFMA_H_ASM " %[c7], %[x3], %[w15], %[c7]\n\t"
        "s_nop 15\n\t"
        : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [w4] "=&v"(w4), [w5] "=&v"(w5), [w6] "=&v"(w6), [w7] "=&v"(w7), [w8] "=&v"(w8), [w9] "=&v"(w9), [w10] "=&v"(w10), [w11] "=&v"(w11), [w12] "=&v"(w12), [w13] "=&v"(w13), [w14] "=&v"(w14), [w15] "=&v"(w15)
        : [a] "v"(addr), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3])
        : "memory");
  }
  if (!PLAIN) {
    if (HID) asm volatile(DECW_GEMM_SWAPPED_ASM(RIFT_MFMA_HID_ASM)
        : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [w4] "=&v"(w4), [w5] "=&v"(w5), [w6] "=&v"(w6), [w7] "=&v"(w7), [w8] "=&v"(w8), [w9] "=&v"(w9), [w10] "=&v"(w10), [w11] "=&v"(w11), [w12] "=&v"(w12), [w13] "=&v"(w13), [w14] "=&v"(w14), [w15] "=&v"(w15)
        : [a] "v"(addr), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3])
        : "memory");
    else asm volatile(DECW_GEMM_SWAPPED_ASM(RIFT_MFMA_H_ASM)
        : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [w4] "=&v"(w4), [w5] "=&v"(w5), [w6] "=&v"(w6), [w7] "=&v"(w7), [w8] "=&v"(w8), [w9] "=&v"(w9), [w10] "=&v"(w10), [w11] "=&v"(w11), [w12] "=&v"(w12), [w13] "=&v"(w13), [w14] "=&v"(w14), [w15] "=&v"(w15)
        : [a] "v"(addr), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3])
        : "memory");

  }
}

}  // namespace RIFT_NS
