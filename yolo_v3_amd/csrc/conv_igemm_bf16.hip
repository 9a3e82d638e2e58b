// bf16 implicit-GEMM convolution (bf16 activations/weights, fp32 accumulate + fp32 epilogue).
// Placeholder until the bf16 MFMA kernel lands: the entry point exists so the ABI is stable,
// and it refuses loudly rather than silently computing in another precision.
#include "yv3_common.h"

int yv3_conv2d_bf16(const yv3_conv_desc*, hipStream_t) { return YV3_EDTYPE; }
