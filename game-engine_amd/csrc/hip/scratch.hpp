// Static device scratch of the allocation-free entry point (see scratch.hip).
#pragma once
#include <cstddef>

// Bytes of the file-scope device array: room for 65536 floats of uniform rows (a 12-float row: 5461 frames per call).
#define GAME_HIP_SCRATCH_BYTES (256u * 1024u)

namespace game {
// Device address of the scratch array on the CURRENT device (hipGetSymbolAddress, once per device).
int scratch_for_current_device(void** ptr);
}
