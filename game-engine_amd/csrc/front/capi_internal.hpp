// Shared definitions behind the opaque C handles of include/game_hip.h.
#pragma once
#include "front.hpp"
#include "../../../include/game_hip.h"
#include "../../../include/game_hip_diag.h"
#include <string>
#include <vector>

struct game_ir {
    const game::Ir* ir = nullptr;
    const std::vector<uint8_t>* blob = nullptr;
    const std::string* json = nullptr;
};

struct game_program {
    game::CompiledProgram prog;
    std::vector<game_ir> irs;
};

namespace game {
extern thread_local std::string g_last_error;
int set_error(int code, const std::string& msg);
// Tables of the current HIP device (see game_device.h): kind 0 = lattice hash pairs, 1 = voronoi feature
// points, 2 = simplex corners. Built on first use and kept for the life of the process. Defined with the ahead-of-time
// kernels; returns a GAME_E_* code.
int lut_for_current_device(int kind, const float** table);
// Frees every table built so far (game_hip_shutdown).
int release_device_tables();
// Render contexts alive right now (created and not yet destroyed): they hold pointers into those tables.
int live_context_count();
int error_code_of(const CompileError& e);
}  // namespace game
