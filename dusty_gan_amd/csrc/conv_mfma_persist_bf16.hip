// bf16 instantiations of the persistent implicit-GEMM conv (conv_mfma_persist_impl.h)
#include "conv_mfma_persist_impl.h"

int dg_conv_mfma_persist_launch_bf16(const ConvP* p, hipStream_t stream, int auto_rule, int wg_cap, DgConvPlan* plan) {
  return persist::launch_dtype<bf16>(p, stream, auto_rule != 0, wg_cap, plan);
}
