// Wave-private NAT levels 0 and 1 (nat_l0w.h, nat_l1w.h): their own translation unit, built with -fno-honor-nans -mno-amdgpu-ieee like the
// other wave-private kernels (rift_amd/build.py: FAST) -- no `v_max_f32 x, x, x` canonicalisation in front of every max over an MFMA result
// (round 5: the softmax of the MFMA neighbourhood attention had 28 of them per head; these kernels test no NaN themselves, the policy-head
// kernels' bit-pattern test does).
#define RIFT_NAT_L01_IMPL 1
#include "common.h"
#include "nat_l0w.h"
#include "nat_l1w.h"

namespace RIFT_NS {

int l0w_set_attributes() {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&nat_l0w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L0W_LDS);
}
void l0w_pack(const NatL0WSrc& src, unsigned short* img, float* par, hipStream_t stream) {
  hipLaunchKernelGGL(pack_l0w_kernel, dim3((L0W_NFRAG * 512 + 255) / 256), dim3(256), 0, stream, src, img, par);
}
void l0w_launch(const NatL0WP& p, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(nat_l0w_kernel, dim3(grid), dim3(64 * L0W_NWV), (size_t)L0W_LDS, stream, p);
}

int l1w_set_attributes() {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&nat_l1w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L1W_LDS);
}
void l1w_pack(const NatL1WSrc& src, unsigned short* img, float* par, hipStream_t stream) {
  hipLaunchKernelGGL(pack_l1w_kernel, dim3((L1W_NFRAG * 512 + 255) / 256), dim3(256), 0, stream, src, img, par);
}
void l1w_launch(const NatL1WP& p, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(nat_l1w_kernel, dim3(grid), dim3(64 * L1W_NWV), (size_t)L1W_LDS, stream, p);
}

}  // namespace RIFT_NS
