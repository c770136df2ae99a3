// Drives include/WSTessendorf.hpp the way the reference's only caller does
// (src/scene/WaterSurfaceMesh.cpp:123-154, 701-755): Prepare, ComputeWaves per
// frame, then both maps memcpy'd back to back into one staging buffer.
// Prints "N A min max sum_disp sum_nrm" so the GPU test can compare it with the
// Python binding on the same seed.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "WSTessendorf.hpp"

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : WSTessendorf::s_kDefaultTileSize;
    const float t = argc > 2 ? (float)std::atof(argv[2]) : 1.5f;
    try {
        WSTessendorf model(n, WSTessendorf::s_kDefaultTileLength);
        model.SetWindDirection(WSTessendorf::vec2(1.0f, 0.5f));
        model.SetWindSpeed(20.0f);
        model.SetLambda(-1.5f);
        model.Prepare(42);
        const float amp = model.ComputeWaves(t);

        const size_t dispBytes = sizeof(WSTessendorf::Displacement) * model.GetDisplacementCount();
        const size_t nrmBytes = sizeof(WSTessendorf::Normal) * model.GetNormalCount();
        std::vector<unsigned char> staging(dispBytes + nrmBytes);
        std::memcpy(staging.data(), model.GetDisplacements().data(), dispBytes);
        std::memcpy(staging.data() + dispBytes, model.GetNormals().data(), nrmBytes);

        double sd = 0.0, sn = 0.0;
        const float* f = reinterpret_cast<const float*>(staging.data());
        for (size_t i = 0; i < dispBytes / 4; ++i) sd += (double)f[i] * (double)((i % 7) + 1);
        for (size_t i = dispBytes / 4; i < (dispBytes + nrmBytes) / 4; ++i) sn += (double)f[i] * (double)((i % 5) + 1);
        std::printf("%u %.9g %.9g %.9g %.12g %.12g\n", model.GetTileSize(), amp, model.GetMinHeight(),
                    model.GetMaxHeight(), sd, sn);
        // optional third argument: time that many frames the way WaterSurfaceMesh::Update does
        // (ComputeWaves + both maps in host memory), reported on stderr
        const int frames = argc > 3 ? std::atoi(argv[3]) : 0;
        if (frames > 0) {
            for (int j = 0; j < 5; ++j) model.ComputeWaves(t + 0.05f * (float)j);
            const auto t0 = std::chrono::steady_clock::now();
            for (int j = 0; j < frames; ++j) model.ComputeWaves(t + 0.05f * (float)j);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            std::fprintf(stderr, "adaptor_demo: N=%u ComputeWaves + read-out of both maps: %.1f us/frame\n",
                         model.GetTileSize(), us / frames);
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "adaptor_demo: %s\n", e.what());
        return 3;
    }
}
