// frames_small.hip -- frame launchers of tile size(s) 16, 32, 64, 128, 256 (one translation unit per group: parallel build).
#include "ocean_launch.h"

hipError_t ocean_launch_frame_small(ocean_ctx* c, const FrameArgs& a, int stream_maps, hipStream_t st, hipEvent_t* marks)
{
    switch (c->n) {
        case 16: return launch_frame<16>(c, a, stream_maps, st, marks);
        case 32: return launch_frame<32>(c, a, stream_maps, st, marks);
        case 64: return launch_frame<64>(c, a, stream_maps, st, marks);
        case 128: return launch_frame<128>(c, a, stream_maps, st, marks);
        case 256: return launch_frame<256>(c, a, stream_maps, st, marks);
        default: return hipErrorInvalidValue;
    }
}
