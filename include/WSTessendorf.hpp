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
    ~WSTessendorf() { Unpin(); ocean_destroy(m_Ctx); }
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
        Check(ocean_prepare(m_Ctx, seed, gaussRandomOrNull), "ocean_prepare");
        const size_t n = ocean_tile_size(m_Ctx);
        Unpin();
        m_Displacements.assign(n * n, Displacement(0.f, 0.f, 0.f, 0.f));     // .cpp:48-51
        m_Normals.assign(n * n, Normal(0.f, 1.f, 0.f, 0.f));                 // .cpp:53-54
        // page-lock the two host vectors so the per-frame read-out is a direct DMA (best effort:
        // a pageable vector still works, the copy is then staged by the runtime)
        m_Pinned = ocean_host_register(m_Displacements.data(), n * n * sizeof(Displacement)) == OCEAN_OK;
        if (m_Pinned && ocean_host_register(m_Normals.data(), n * n * sizeof(Normal)) != OCEAN_OK) {
            ocean_host_unregister(m_Displacements.data());
            m_Pinned = false;
        }
    }

    // WSTessendorf.cpp:284-455: returns the amplitude of the normalised heights
    float ComputeWaves(float time)
    {
        float amp = 0.f;
        Check(ocean_compute_waves(m_Ctx, time, &amp), "ocean_compute_waves");
        Check(ocean_read_maps(m_Ctx, 0, 1, reinterpret_cast<float*>(m_Displacements.data()),
                              reinterpret_cast<float*>(m_Normals.data())), "ocean_read_maps");
        float a;
        Check(ocean_get_heights(m_Ctx, 0, &a, &m_MinHeight, &m_MaxHeight), "ocean_get_heights");
        return amp;
    }

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
    size_t GetDisplacementCount() const { return m_Displacements.size(); }
    const std::vector<Displacement>& GetDisplacements() const { return m_Displacements; }
    size_t GetNormalCount() const { return m_Normals.size(); }
    const std::vector<Normal>& GetNormals() const { return m_Normals; }

    // Setters: WSTessendorf.cpp:459-505
    void SetTileSize(uint32_t size)
    {
        if (size == 0 || (size & (size - 1))) return;
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
    void Unpin()
    {
        if (!m_Pinned) return;
        ocean_host_unregister(m_Displacements.data());
        ocean_host_unregister(m_Normals.data());
        m_Pinned = false;
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
    std::vector<Displacement> m_Displacements;
    std::vector<Normal> m_Normals;
    float m_MinHeight{ -1.0f };     // WSTessendorf.h:227-228
    float m_MaxHeight{ 1.0f };
    uint64_t m_PrepareCount{ 0 };
    bool m_Pinned{ false };
};

#endif  // WS_TESSENDORF_ADAPTOR_HPP_
