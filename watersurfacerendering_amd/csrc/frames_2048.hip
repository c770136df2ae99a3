// frames_2048.hip -- frame launchers of tile size(s) 2048 (one translation unit per group: parallel build).
#include "ocean_launch.h"

hipError_t ocean_launch_frame_2048(ocean_ctx* c, const FrameArgs& a, int stream_maps, hipStream_t st, hipEvent_t* marks)
{
    switch (c->n) {
        case 2048: return launch_frame<2048>(c, a, stream_maps, st, marks);
        default: return hipErrorInvalidValue;
    }
}
