// wino44b.h -- host entry points of the 3 x bf16 split F(4x4,3x3) Winograd conv layer (dncnn_wino44b.hip), used by the DnCNN plan
// (conv mode 6: opt-in, fp32-class accuracy on the bf16 matrix cores; not the reference's arithmetic operation for operation).
#pragma once
#include "common.h"

namespace pnp {
bool wino44b_supports(int H, int W);                                         // H % 8 == 0 and W % 64 == 0
size_t wino44b_weight_halfwords(int n_mid);                                  // uint16 elements of the packed weights
void wino44b_pack_weights(const float* w_mid, int n_mid, uint16_t* out);     // host -> host buffer (U = G g G^T, split in three)
int wino44b_layer(const float* in, float* out, const uint16_t* upack_layer, const float* bias, int H, int W, int batch, int num_cu,
                  float slope, hipStream_t s);
int wino44b_debug_clock(const float* in, float* out, const uint16_t* upack_layer, const float* bias, int H, int W, int batch,
                        int num_cu, int reps, unsigned long long* stamps_dev, hipStream_t s);   // 4 values per workgroup
}  // namespace pnp
