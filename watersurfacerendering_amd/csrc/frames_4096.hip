// frames_4096.hip -- frame launchers of tile size(s) 4096 (one translation unit per group: parallel build).
#include "ocean_launch.h"

hipError_t ocean_launch_frame_4096(ocean_ctx* c, const FrameArgs& a, int stream_maps, hipStream_t st, hipEvent_t* marks)
{
    switch (c->n) {
        case 4096: return launch_frame<4096>(c, a, stream_maps, st, marks);
        default: return hipErrorInvalidValue;
    }
}
