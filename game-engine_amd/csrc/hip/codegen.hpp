// Kernel generator interface (see codegen.cpp).
#pragma once
#include "../front/front.hpp"
#include <string>

namespace game {

struct KernelShape {
    uint32_t pixels_per_lane = 1;   // consecutive pixels in x handled by one lane
    uint32_t rows_per_lane = 1;     // rows a lane walks (one workgroup row of the grid covers this many image rows)
    uint32_t time_slot = 0;         // index in the frame row of fract(u_time/120)*120
    uint32_t aspect_slot = 0;       // index in the frame row of u_resolution.x / u_resolution.y
    bool wgsl = false;              // builtins as worded by the reference's WebGPU emitter (wgsl.rs) where they differ
    bool animate = false;           // expression-valued stage arguments read their per-frame value (Ir::derived)
    uint32_t derived_slot = 0;      // index in the frame row of derived value 0
    bool lens_post = false;         // run the lens post chain on the composited colour
    bool project = false;           // apply the cinematic's projection block (Ir::project) to v_uv
    uint32_t wg = 256;              // threads per workgroup (a multiple of 64)
    bool stream_prev = false;       // the previous frame is read with a non-temporal hint (frames too large to stay in cache)
    bool row_shared = true;         // with several rows per lane: per-row values worked out once per workgroup, in shared memory
    bool speculate = false;         // four-pixel lanes: range tests recorded instead of branched on, one check per wave and row
                                    // (and a second, fully tested evaluation when some lane left a range)
};

std::string float_literal(float v);
std::string generate_frame_kernel(const Ir& ir, const KernelShape& shape);
// fast: root and reciprocal of the pair distance from one v_rsq_f32 (not bit-reproducible on a CPU; within tolerance)
std::string generate_gravity_kernel(const Ir& ir, bool fast = false);

}  // namespace game
