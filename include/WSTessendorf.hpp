// include/WSTessendorf.hpp -- header-only C++ adaptor with the reference's class
// surface (/root/reference/src/scene/WSTessendorf.h:58-122) over the C ABI of
// include/ocean.h, so the one caller of the hot path
// (src/scene/WaterSurfaceMesh.cpp:127,131,151,172,179,478-486,701-755,804-902)
// compiles unchanged against the MI355X library instead of the FFTW/OpenMP
// implementation.
//
// Vector types: glm::vec2 / glm::vec4 when <glm/glm.hpp> is on the include path
// (as in the reference build), otherwise layout-identical PODs.
//
// Behavioural notes (all mirror the reference):
//   * setters other than SetLambda take effect at the next Prepare()
//     (WSTessendorf.cpp:459-505; GUI "Apply" path WaterSurfaceMesh.cpp:888-900);
//   * SetTileSize ignores a non power of two (.cpp:461-467);
//   * Prepare() draws NEW gaussian noise every call (.cpp:87-103, std::rand
//     seeded from the clock in core/Application.cpp:21) -- here: a fresh 64-bit
//     seed per call, or Prepare(seed) for reproducible runs;
//   * ComputeWaves returns the height amplitude A and leaves both maps in host
//     vectors valid until the next Prepare()/ComputeWaves (.h:95-107);
//   * errors: the reference has none (asserts -> SIGTRAP).  The adaptor throws
//     std::runtime_error if the device library reports a failure (no GPU, HIP
//     error): there is deliberately no CPU fallback.
#ifndef WS_TESSENDORF_ADAPTOR_HPP_
#define WS_TESSENDORF_ADAPTOR_HPP_

#include <chrono>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "ocean.h"

#if defined(__has_include)
#if __has_include(<glm/glm.hpp>)
#include <glm/glm.hpp>
#define WS_TESSENDORF_HAVE_GLM 1
#endif
#endif

#ifndef WS_TESSENDORF_HAVE_GLM
namespace wsvec {
struct vec2 { float x, y; vec2(float a = 0.f, float b = 0.f) : x(a), y(b) {} };
struct vec4 { float x, y, z, w; vec4(float a = 0.f, float b = 0.f, float c = 0.f, float d = 0.f) : x(a), y(b), z(c), w(d) {} };
}  // namespace wsvec
#endif

class WSTessendorf
{
public:
#ifdef WS_TESSENDORF_HAVE_GLM
    using vec2 = glm::vec2;
    using vec4 = glm::vec4;
#else
    using vec2 = wsvec::vec2;
    using vec4 = wsvec::vec4;
#endif
    static constexpr uint32_t s_kDefaultTileSize{ 512 };
    static constexpr float    s_kDefaultTileLength{ 1000.0f };
    static inline const vec2  s_kDefaultWindDir{ 1.0f, 1.0f };
    static constexpr float    s_kDefaultWindSpeed{ 30.0f };
    static constexpr float    s_kDefaultAnimPeriod{ 200.0f };
    static constexpr float    s_kDefaultPhillipsConst{ 3e-7f };
    static constexpr float    s_kDefaultPhillipsDamping{ 0.1f };

    using Displacement = vec4;
    using Normal       = vec4;
    static_assert(sizeof(vec4) == 16, "maps are RGBA32F");

    explicit WSTessendorf(uint32_t tileSize = s_kDefaultTileSize, float tileLength = s_kDefaultTileLength,
                          int device = 0)
    {
        if (tileSize == 0 || (tileSize & (tileSize - 1))) tileSize = s_kDefaultTileSize;
        Check(ocean_create(&m_Ctx, tileSize, 1, device), "ocean_create");
        ocean_default_params(&m_Params);
        m_Params.tile_length = tileLength;
        Push();
    }
    ~WSTessendorf() { if (m_Pending) (void)ocean_synchronize(m_Ctx); Unpin(); ocean_destroy(m_Ctx); }
    WSTessendorf(const WSTessendorf&) = delete;
    WSTessendorf& operator=(const WSTessendorf&) = delete;

    // WSTessendorf.cpp:36-58
    void Prepare()
    {
        const uint64_t now = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
        Prepare(now ^ (0x9E3779B97F4A7C15ull * ++m_PrepareCount));
    }
    void Prepare(uint64_t seed, const float* gaussRandomOrNull = nullptr)
    {
        if (m_Pending) Wait();
        Check(ocean_prepare(m_Ctx, seed, gaussRandomOrNull), "ocean_prepare");
        const size_t n = ocean_tile_size(m_Ctx);
        Unpin();
        m_Front = 0;
        m_Displacements[1].clear(); m_Normals[1].clear();                      // the back pair exists only once ComputeWavesAsync is used
        m_Displacements[0].assign(n * n, Displacement(0.f, 0.f, 0.f, 0.f));    // .cpp:48-51
        m_Normals[0].assign(n * n, Normal(0.f, 1.f, 0.f, 0.f));                // .cpp:53-54
        // page-lock the two host vectors so the per-frame read-out is a direct DMA (best effort:
        // a pageable vector still works, the copy is then staged by the runtime)
        m_Pinned[0] = Pin(0);
    }

    // WSTessendorf.cpp:284-455: returns the amplitude of the normalised heights
    float ComputeWaves(float time)
    {
        if (m_Pending) Wait();
        float amp = 0.f;
        // synthesis + both maps into the host vectors, one blocking call: the normal map's copy runs beside the displacement pass
        // (ocean_compute_waves_read; the same maps as ocean_compute_waves + ocean_read_maps)
        Check(ocean_compute_waves_read(m_Ctx, time, &amp, reinterpret_cast<float*>(m_Displacements[m_Front].data()),
                                       reinterpret_cast<float*>(m_Normals[m_Front].data())), "ocean_compute_waves_read");
        float a;
        Check(ocean_get_heights(m_Ctx, 0, &a, &m_MinHeight, &m_MaxHeight), "ocean_get_heights");
        return amp;
    }

    // Opt-in non-blocking pair, beyond the reference -- whose own note on its DOUBLE_BUFFERED switch says "should be on dedicated
    // thread" (WaterSurfaceMesh.h:26-34): ComputeWaves above is synthesis + a blocking copy of both maps to the host, and the copy
    // is 10-40 x the synthesis (INTEGRATION.md section A has the measured call).  ComputeWavesAsync enqueues the frame and the DMA of both maps
    // into a BACK pair of host vectors and returns A as soon as the frame's kernels have finished (ocean_wait_frame: a poll of
    // the frame's completion records) -- the copy is still in flight.  Wait() blocks until it has landed and makes that pair the
    // front one.  In between, GetDisplacements() / GetNormals() / GetMinHeight() / GetMaxHeight() keep returning the previous
    // frame, untouched: exactly what DOUBLE_BUFFERED does with its two texture pairs (WaterSurfaceMesh.cpp:187-199, 233-239).
    float ComputeWavesAsync(float time)
    {
        if (m_Pending) Wait();
        const int back = m_Front ^ 1;
        const size_t n2 = m_Displacements[m_Front].size();
        if (!m_Tracking) { Check(ocean_set_frame_tracking(m_Ctx, 1), "ocean_set_frame_tracking"); m_Tracking = true; }
        if (m_Displacements[back].size() != n2) {
            m_Displacements[back].assign(n2, Displacement(0.f, 0.f, 0.f, 0.f));
            m_Normals[back].assign(n2, Normal(0.f, 1.f, 0.f, 0.f));
            m_Pinned[back] = Pin(back);
        }
        Check(ocean_compute_waves_async(m_Ctx, time), "ocean_compute_waves_async");
        Check(ocean_read_maps_async(m_Ctx, 0, 1, reinterpret_cast<float*>(m_Displacements[back].data()),
                                    reinterpret_cast<float*>(m_Normals[back].data())), "ocean_read_maps_async");
        float amp = 0.f, a;
        Check(ocean_wait_frame(m_Ctx, &amp), "ocean_wait_frame");
        Check(ocean_get_heights(m_Ctx, 0, &a, &m_PendingMin, &m_PendingMax), "ocean_get_heights");
        m_Pending = true;
        return amp;
    }
    void Wait()
    {
        if (!m_Pending) return;
        Check(ocean_synchronize(m_Ctx), "ocean_synchronize");
        m_Front ^= 1;
        m_MinHeight = m_PendingMin; m_MaxHeight = m_PendingMax;
        m_Pending = false;
    }
    bool Pending() const { return m_Pending; }

#ifdef OCEAN_DEV_H_      // (only for a host that included the developer header first: ocean_select_streams is bench plumbing, not part of the boundary)
    // Opt-in, once after Prepare(): put the model's work on the fastest of the process's hardware queues (ocean_select_streams: the queues
    // of an MI355X process differ by up to 1 us per kernel).  Costs 4 x 55 frames; the next ComputeWaves delivers as ever.
    void SelectFastestQueue(uint32_t framesPerQueue = 50)
    {
        Wait();
        Check(ocean_select_streams(m_Ctx, framesPerQueue, nullptr), "ocean_select_streams");
    }
#endif

    // Getters: WSTessendorf.h:82-107
    auto GetTileSize() const { return ocean_tile_size(m_Ctx); }
    auto GetTileLength() const { return m_Params.tile_length; }
    auto GetWindDir() const
    {
        const float inv = 1.0f / std::sqrt(m_Params.wind_dir_x * m_Params.wind_dir_x +
                                           m_Params.wind_dir_y * m_Params.wind_dir_y);
        return vec2(m_Params.wind_dir_x * inv, m_Params.wind_dir_y * inv);
    }
    auto GetWindSpeed() const { return m_Params.wind_speed; }
    auto GetAnimationPeriod() const { return m_Params.anim_period; }
    auto GetPhillipsConst() const { return m_Params.phillips_const; }
    auto GetDamping() const { return m_Params.damping; }
    auto GetDisplacementLambda() const { return m_Params.lambda; }
    float GetMinHeight() const { return m_MinHeight; }
    float GetMaxHeight() const { return m_MaxHeight; }
    size_t GetDisplacementCount() const { return m_Displacements[m_Front].size(); }
    const std::vector<Displacement>& GetDisplacements() const { return m_Displacements[m_Front]; }
    size_t GetNormalCount() const { return m_Normals[m_Front].size(); }
    const std::vector<Normal>& GetNormals() const { return m_Normals[m_Front]; }

    // Setters: WSTessendorf.cpp:459-505
    void SetTileSize(uint32_t size)
    {
        if (size == 0 || (size & (size - 1))) return;
        if (m_Pending) Wait();
        Check(ocean_set_tile_size(m_Ctx, size), "ocean_set_tile_size");
    }
    void SetTileLength(float length) { m_Params.tile_length = length; Push(); }
    void SetWindDirection(const vec2& w) { m_Params.wind_dir_x = w.x; m_Params.wind_dir_y = w.y; Push(); }
    void SetWindSpeed(float v) { m_Params.wind_speed = v > 0.0001f ? v : 0.0001f; Push(); }
    void SetAnimationPeriod(float T) { m_Params.anim_period = T; Push(); }
    void SetPhillipsConst(float A) { m_Params.phillips_const = A; Push(); }
    void SetLambda(float lambda)
    {
        m_Params.lambda = lambda;
        Check(ocean_set_lambda(m_Ctx, OCEAN_ALL_TILES, lambda), "ocean_set_lambda");
    }
    void SetDamping(float damping) { m_Params.damping = damping; Push(); }

    // Beyond the reference: the device-resident maps (no host copy), for interop.
    ocean_t* Context() const { return m_Ctx; }

private:
    bool Pin(int i)
    {
        const size_t bytes = m_Displacements[i].size() * sizeof(Displacement);
        if (ocean_host_register(m_Displacements[i].data(), bytes) != OCEAN_OK) return false;
        if (ocean_host_register(m_Normals[i].data(), bytes) != OCEAN_OK) {
            ocean_host_unregister(m_Displacements[i].data());
            return false;
        }
        return true;
    }
    void Unpin()
    {
        for (int i = 0; i < 2; ++i) {
            if (!m_Pinned[i]) continue;
            ocean_host_unregister(m_Displacements[i].data());
            ocean_host_unregister(m_Normals[i].data());
            m_Pinned[i] = false;
        }
    }
    void Push() { Check(ocean_set_params(m_Ctx, OCEAN_ALL_TILES, &m_Params), "ocean_set_params"); }
    static void Check(int rc, const char* what)
    {
        if (rc != OCEAN_OK)
            throw std::runtime_error(std::string(what) + ": " + ocean_strerror(rc) +
                                     " (hip " + std::to_string(ocean_last_hip_error()) + ")");
    }

    ocean_t* m_Ctx{ nullptr };
    ocean_params m_Params{};
    std::vector<Displacement> m_Displacements[2];     // [m_Front]: what the getters return; the other pair: ComputeWavesAsync's target
    std::vector<Normal> m_Normals[2];
    int m_Front{ 0 };
    bool m_Pending{ false };                          // a ComputeWavesAsync whose copy has not been waited for
    bool m_Tracking{ false };                         // asynchronous frames leave completion records (ocean_set_frame_tracking)
    float m_PendingMin{ -1.0f }, m_PendingMax{ 1.0f };
    float m_MinHeight{ -1.0f };     // WSTessendorf.h:227-228
    float m_MaxHeight{ 1.0f };
    uint64_t m_PrepareCount{ 0 };
    bool m_Pinned[2]{ false, false };
};

#endif  // WS_TESSENDORF_ADAPTOR_HPP_
