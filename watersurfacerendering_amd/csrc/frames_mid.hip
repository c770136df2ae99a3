// frames_mid.hip -- frame launchers of tile size(s) 512, 1024 (one translation unit per group: parallel build).
#include "ocean_launch.h"

hipError_t ocean_launch_frame_mid(ocean_ctx* c, const FrameArgs& a, int stream_maps, hipStream_t st, hipEvent_t* marks)
{
    switch (c->n) {
        case 512: return launch_frame<512>(c, a, stream_maps, st, marks);
        case 1024: return launch_frame<1024>(c, a, stream_maps, st, marks);
        default: return hipErrorInvalidValue;
    }
}
