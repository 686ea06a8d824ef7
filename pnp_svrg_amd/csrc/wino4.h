// wino4.h -- host entry points of the F(4,3) Winograd conv layer (dncnn_wino4.hip), used by the DnCNN plan.
#pragma once
#include "common.h"

namespace pnp {
bool wino4_supports(int H, int W);                                           // H % 4 == 0 and W % 64 == 0
size_t wino4_weight_floats(int n_mid);
void wino4_pack_weights(const float* w_mid, int n_mid, float* out);          // host -> host buffer (G g along dx)
int wino4_layer(const float* in, float* out, const float* upack_layer, const float* bias, const float* zeros, int H, int W,
                int batch, int num_cu, float slope, hipStream_t s);
}  // namespace pnp
