// Separable binomial blur of an RGBA8 frame (see blur.hip).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace game {
// out rows [row0, row0 + rows) of a w x h frame := blur of `raw` (same size) with radius 1..4; rows of `raw` up to
// `radius` beyond the band are read (clamped at the frame's edges). Queued on `stream`. Returns a GAME_E_* code.
int launch_blur(const uint32_t* raw, uint32_t* out, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows, uint32_t radius,
                hipStream_t stream);
}
