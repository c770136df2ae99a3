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

#include "ocean_dev.h"          // (SelectFastestQueue below: the adaptor offers it only to hosts that include the developer header)
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
        if (argc > 5 && std::strcmp(argv[5], "select") == 0) model.SelectFastestQueue(20);   // must not change a bit of what follows
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
            std::fprintf(stderr, "adaptor_demo: N=%u ComputeWaves + read-out of both maps: %.1f us/frame = %.1f GB/s over PCIe\n",
                         n, us / frames, (double)(dispBytes + nrmBytes) * frames / us * 1e-3);
        }
        // optional fourth argument "async": the opt-in ComputeWavesAsync() / Wait() pair must deliver the same frame, keep the
        // previous one readable in between, and return A before the copy has landed.  Prints "async <ok> <us to A> <us to maps>".
        if (argc > 4 && std::strcmp(argv[4], "async") == 0) {
            const float t2 = t + 2.0f;
            const float prevFirst = model.GetDisplacements()[1].x, prevAmp = amp;
            const auto t0 = std::chrono::steady_clock::now();
            const float a2 = model.ComputeWavesAsync(t2);
            const auto t1 = std::chrono::steady_clock::now();
            bool ok = model.Pending() && model.GetDisplacements()[1].x == prevFirst;      // front pair untouched while the copy flies
            model.Wait();
            const auto t3 = std::chrono::steady_clock::now();
            std::vector<WSTessendorf::Displacement> d2 = model.GetDisplacements();
            std::vector<WSTessendorf::Normal> q2 = model.GetNormals();
            const float mn2 = model.GetMinHeight(), mx2 = model.GetMaxHeight();
            const float a3 = model.ComputeWaves(t2);                                      // the blocking call on the same time
            ok = ok && a2 == a3 && a2 != prevAmp && mn2 == model.GetMinHeight() && mx2 == model.GetMaxHeight() &&
                 std::memcmp(d2.data(), model.GetDisplacements().data(), dispBytes) == 0 &&
                 std::memcmp(q2.data(), model.GetNormals().data(), nrmBytes) == 0;
            std::printf("async %d %.1f %.1f\n", ok ? 1 : 0, std::chrono::duration<double, std::micro>(t1 - t0).count(),
                        std::chrono::duration<double, std::micro>(t3 - t0).count());
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "adaptor_demo: %s\n", e.what());
        return 3;
    }
}
