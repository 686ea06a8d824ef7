// f16x3.h -- host entry points of the opt-in split-fp16 conv layer (dncnn_f16x3.hip), used by the DnCNN plan.
#pragma once
#include "common.h"

namespace pnp {
size_t f16x3_weight_bytes(int n_mid);
void f16x3_pack_weights(const float* w_mid, int n_mid, void* out_h8);           // host -> host buffer
int f16x3_layer(const void* in_a16, void* out, const void* wpack_layer, const float* bias, const void* zeros, int H, int W,
                int batch, int num_cu, int out_f32, float slope, hipStream_t s, unsigned long long* stamps = nullptr);
}  // namespace pnp
