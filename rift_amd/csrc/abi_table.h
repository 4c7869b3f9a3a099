// Table of the C-ABI entry points of one engine build (bf16 or fp16 MFMA operands).  The member types come from the declarations of
// include/rift_hip.h, so an implementation whose signature drifts from the header does not compile.
#pragma once
#include "../../include/rift_hip.h"

struct RiftVTable {
#define RIFT_FN(name) decltype(&::name) name;
#include "abi_list.h"
#undef RIFT_FN
};
