// Static device scratch of the allocation-free entry point (see scratch.hip).
#pragma once
#include <cstddef>

// Bytes of the file-scope device array: room for 65536 floats of uniform rows (a 12-float row: 5461 frames per call).
#define GAME_HIP_SCRATCH_BYTES (256u * 1024u)

namespace game {
// Device address of the scratch array on the CURRENT device (hipGetSymbolAddress, once per device), and that device's
// "last reader" event (a hipEvent_t, made with the address): every context of a GPU shares the array, so a call that
// refills it first makes its stream wait for this event, and records it again behind the kernels that read the array.
int scratch_for_current_device(void** ptr, void** last_reader_event);
}
