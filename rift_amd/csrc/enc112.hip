// The fused scene encoder on its 112-row layout (enc_fused.h: EncLay<112>, enc_fused112_kernel) -- the shapes train_cbv collates, 97 .. 112
// token slots per scene -- in a translation unit of its own: instantiated beside the 96-row kernel in engine.hip it changed THAT kernel's
// register allocation (2 VGPR spills next to its 9 SGPR spills: the state rift_amd/build.py refuses, DESIGN.md "hazards").
#define RIFT_ENC112_TU 1
#include "common.h"
#include "enc_fused.h"

namespace RIFT_NS {

int enc112_set_attributes() {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_fused112_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, EncLay<112>::BYTES);
}
void enc112_launch(const EncFusedP& p, hipStream_t stream) {
  hipLaunchKernelGGL(enc_fused112_kernel<8>, dim3(p.bs), dim3(512), (size_t)EncLay<112>::BYTES, stream, p);
}

}  // namespace RIFT_NS
