// The importing side of ocean_export_maps, as a process of its own: takes a dma-buf file descriptor it inherited, imports it
// with hipImportExternalMemory, maps it and writes the two maps it finds at the given offsets to a file -- what a second API
// (the reference's Vulkan renderer: INTEGRATION.md section B) or a second process does with the exported maps instead of the
// reference's staging-buffer round trip (WaterSurfaceMesh.cpp:642-755).
//   import_demo <fd> <bytes> <disp_offset> <nrm_offset> <map_bytes> <out_file>
// With "-" as the file it stays alive as the importer of a running producer: every line "r" on stdin makes it read both maps
// through the import at once and answer "SUM <checksum>" (sum of word_i * (i + 1) over the 32-bit words, mod 2^64); "q" ends it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "import_demo: %s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 7) return 1;
    const int fd = std::atoi(argv[1]);
    const size_t bytes = std::strtoull(argv[2], nullptr, 10), doff = std::strtoull(argv[3], nullptr, 10),
                 noff = std::strtoull(argv[4], nullptr, 10), map_bytes = std::strtoull(argv[5], nullptr, 10);
    CHECK(hipSetDevice(0));
    hipExternalMemoryHandleDesc hd = {};
    hd.type = hipExternalMemoryHandleTypeOpaqueFd;       // a dma-buf descriptor is what amdgpu's opaque fds are
    hd.handle.fd = fd;
    hd.size = bytes;
    hipExternalMemory_t ext = nullptr;
    CHECK(hipImportExternalMemory(&ext, &hd));
    hipExternalMemoryBufferDesc bd = {};
    bd.offset = 0; bd.size = bytes;
    void* base = nullptr;
    CHECK(hipExternalMemoryGetMappedBuffer(&base, ext, &bd));
    std::vector<unsigned char> host(2 * map_bytes);
    if (std::strcmp(argv[6], "-") == 0) {
        std::printf("IMPORT_READY\n");
        std::fflush(stdout);
        char line[64];
        while (std::fgets(line, sizeof line, stdin) && line[0] == 'r') {
            CHECK(hipMemcpy(host.data(), static_cast<char*>(base) + doff, map_bytes, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(host.data() + map_bytes, static_cast<char*>(base) + noff, map_bytes, hipMemcpyDeviceToHost));
            uint64_t sum = 0;
            const uint32_t* w = reinterpret_cast<const uint32_t*>(host.data());
            for (size_t i = 0; i < host.size() / 4; ++i) sum += (uint64_t)w[i] * (uint64_t)(i + 1);
            std::printf("SUM %llu\n", (unsigned long long)sum);
            std::fflush(stdout);
        }
        CHECK(hipDestroyExternalMemory(ext));
        return 0;
    }
    CHECK(hipMemcpy(host.data(), static_cast<char*>(base) + doff, map_bytes, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(host.data() + map_bytes, static_cast<char*>(base) + noff, map_bytes, hipMemcpyDeviceToHost));
    FILE* f = std::fopen(argv[6], "wb");
    if (!f || std::fwrite(host.data(), 1, host.size(), f) != host.size()) return 3;
    std::fclose(f);
    CHECK(hipDestroyExternalMemory(ext));
    std::printf("IMPORT_OK %zu\n", host.size());
    return 0;
}
