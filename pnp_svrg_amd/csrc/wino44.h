// wino44.h -- host entry points of the F(4x4,3x3) Winograd conv layer (dncnn_wino44.hip), used by the DnCNN plan.
#pragma once
#include "common.h"

namespace pnp {
bool wino44_supports(int H, int W);                                          // H % 8 == 0 and W % 64 == 0
size_t wino44_weight_floats(int n_mid);
void wino44_pack_weights(const float* w_mid, int n_mid, float* out);         // host -> host buffer (U = G g G^T)
int wino44_layer(const float* in, float* out, const float* upack_layer, const float* bias, const float* zeros, int H, int W,
                 int batch, int num_cu, float slope, hipStream_t s, int force_rows = 0);   // force_rows 1 / 2: test hook only
int wino44_debug_clock(const float* in, float* out, const float* upack_layer, const float* bias, int H, int W, int batch,
                       int num_cu, int reps, unsigned long long* stamps_dev, hipStream_t s);   // 4 values per workgroup
}  // namespace pnp
