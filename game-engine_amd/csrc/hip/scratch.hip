// Static device scratch for the entry point that allocates nothing (game_hip_render_into): the staging copy of the
// per-frame uniform table, which every other render call keeps in a buffer from hipMalloc, lives here in a
// file-scope device array instead. Each GPU the library runs on holds its own copy of the array, at its own
// address, until the process ends; host code must never use the array's name as an address (it is a host-side
// stand-in), only the per-device pointer hipGetSymbolAddress returns.
#include <hip/hip_runtime.h>
#include <mutex>
#include "../front/capi_internal.hpp"
#include "scratch.hpp"

static __device__ __align__(256) unsigned char scratch[GAME_HIP_SCRATCH_BYTES];

namespace game {

int scratch_for_current_device(void** ptr, void** last_reader_event)
{
    static std::mutex mu;
    static void* per_device[64] = {};
    static hipEvent_t reader[64] = {};
    if (!ptr || !last_reader_event) return set_error(GAME_E_ARGUMENT, "scratch_for_current_device: null argument");
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_error(GAME_E_HIP, std::string("hipGetDevice: ") + hipGetErrorString(e));
    if (dev < 0 || dev >= 64) return set_error(GAME_E_ARGUMENT, "scratch_for_current_device: device ordinal beyond 63");
    std::lock_guard<std::mutex> lock(mu);
    if (!per_device[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return set_error(GAME_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return set_error(GAME_E_HIP, std::string("hipEventCreate(scratch reader): ") + hipGetErrorString(e));
        reader[dev] = ev;
        per_device[dev] = p;
    }
    *ptr = per_device[dev];
    *last_reader_event = (void*)reader[dev];
    return GAME_OK;
}

}  // namespace game
