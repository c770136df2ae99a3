// The tile-sharded batch mode driven by a plain C++ host through the C ABI of include/ocean.h: one process per GPU
// (fork, before anything touches a GPU), tiles sharded in contiguous blocks, NO data-path collective during synthesis,
// one gather of the packed maps per batch to rank 0 over RCCL (ocean_gather_maps), overlapped with the synthesis of the
// next batch (pipeline depth 2).  This is the BASELINE.json config-5 shape ("64 independent 1024x1024 tiles sharded
// 8/GPU, RCCL gather over xGMI") without Python or torch; the RCCL unique id travels from rank 0 to the
// others through pipes (any channel would do: MPI, a file).
//
//   gather_demo <ranks> <tile size> <tiles per rank> <batches>
//
// Prints "GATHER_OK ranks N tiles T ms_per_batch X" when every tile of every rank arrived on the root bit for bit
// (own tiles compared in full, the other ranks' by an order-independent checksum each rank computes on its own maps).
#include <hip/hip_runtime_api.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ocean.h"

#define CHECK(call)                                                                                          \
    do {                                                                                                     \
        int rc_ = (call);                                                                                    \
        if (rc_ != OCEAN_OK) {                                                                               \
            std::fprintf(stderr, "rank %d: %s -> %s (hip %d, rccl %d)\n", rank, #call, ocean_strerror(rc_),  \
                         ocean_last_hip_error(), ocean_last_rccl_error());                                   \
            return 10;                                                                                       \
        }                                                                                                    \
    } while (0)

static uint64_t checksum(const float* p, size_t floats)
{
    uint64_t s = 0;                        // sum of the bit patterns: order independent, exact
    for (size_t i = 0; i < floats; ++i) { uint32_t u; std::memcpy(&u, p + i, 4); s += u; }
    return s;
}

static int run_rank(int rank, int ranks, uint32_t n, uint32_t tiles, int batches, const unsigned char* id, int report_fd, const int* peer_fds)
{
    const uint64_t seed = 0x5EED0000ull;
    ocean_t* oc = nullptr;
    CHECK(ocean_create(&oc, n, tiles, rank));                                   // rank r drives GPU r
    CHECK(ocean_prepare(oc, seed + (uint64_t)rank * tiles, nullptr));           // global tile g has seed + g
    CHECK(ocean_comm_init(oc, ranks, rank, id));
    CHECK(ocean_set_pipeline_depth(oc, 2));
    const size_t map_floats = (size_t)tiles * n * n * 4;
    void *d_disp = nullptr, *d_nrm = nullptr;
    if (rank == 0) {
        if (hipMalloc(&d_disp, map_floats * 4 * ranks) != hipSuccess || hipMalloc(&d_nrm, map_floats * 4 * ranks) != hipSuccess) return 11;
    }
    for (int j = 0; j < 3; ++j) {                                               // warm-up: first touch of every chain, RCCL channels
        CHECK(ocean_compute_waves_async(oc, 0.05f * j));
        CHECK(ocean_gather_maps(oc, 0, d_disp, d_nrm));
    }
    CHECK(ocean_synchronize(oc));
    const auto t0 = std::chrono::steady_clock::now();
    float t_last = 0.f;
    for (int j = 0; j < batches; ++j) {
        t_last = 1.0f + 0.05f * j;
        CHECK(ocean_compute_waves_async(oc, t_last));                           // batch j+1 is synthesised while batch j is gathered
        CHECK(ocean_gather_maps(oc, 0, d_disp, d_nrm));
    }
    CHECK(ocean_synchronize(oc));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / batches;
    // what this rank produced last
    std::vector<float> disp(map_floats), nrm(map_floats);
    CHECK(ocean_read_maps(oc, 0, tiles, disp.data(), nrm.data()));
    const uint64_t mine[2] = {checksum(disp.data(), map_floats), checksum(nrm.data(), map_floats)};
    int bad = 0;
    if (rank != 0) {
        if (write(report_fd, mine, sizeof mine) != (ssize_t)sizeof mine) bad = 1;
    } else {
        std::vector<float> got(map_floats);
        for (int r = 0; r < ranks && !bad; ++r) {
            uint64_t want[2] = {mine[0], mine[1]};
            if (r > 0 && read(peer_fds[r], want, sizeof want) != (ssize_t)sizeof want) { bad = 1; break; }
            for (int m = 0; m < 2 && !bad; ++m) {
                const char* src = static_cast<const char*>(m ? d_nrm : d_disp) + (size_t)r * map_floats * 4;
                if (hipMemcpy(got.data(), src, map_floats * 4, hipMemcpyDeviceToHost) != hipSuccess) { bad = 1; break; }
                if (r == 0 && std::memcmp(got.data(), m ? nrm.data() : disp.data(), map_floats * 4) != 0) bad = 1;
                if (checksum(got.data(), map_floats) != want[m]) bad = 1;
            }
        }
        std::printf("%s ranks %d tiles %u ms_per_batch %.3f\n", bad ? "GATHER_MISMATCH" : "GATHER_OK", ranks, tiles * ranks, ms);
        (void)hipFree(d_disp); (void)hipFree(d_nrm);
    }
    CHECK(ocean_comm_destroy(oc));
    ocean_destroy(oc);
    return bad ? 12 : 0;
}

int main(int argc, char** argv)
{
    const int ranks = argc > 1 ? std::atoi(argv[1]) : 1;
    const uint32_t n = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 256;
    const uint32_t tiles = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 2;
    const int batches = argc > 4 ? std::atoi(argv[4]) : 10;
    if (ranks < 1 || ranks > 64) return 2;
    // pipes: id_pipe[r] root -> rank r (the RCCL id), sum_pipe[r] rank r -> root (its checksums); forks come FIRST,
    // no process has made a HIP call yet
    std::vector<int> id_rd(ranks, -1), id_wr(ranks, -1), sum_rd(ranks, -1), sum_wr(ranks, -1);
    for (int r = 1; r < ranks; ++r) {
        int a[2], b[2];
        if (pipe(a) || pipe(b)) return 3;
        id_rd[r] = a[0]; id_wr[r] = a[1]; sum_rd[r] = b[0]; sum_wr[r] = b[1];
    }
    std::vector<pid_t> kids;
    for (int r = 1; r < ranks; ++r) {
        const pid_t pid = fork();
        if (pid < 0) return 4;
        if (pid == 0) {
            unsigned char id[OCEAN_COMM_ID_BYTES];
            if (read(id_rd[r], id, sizeof id) != (ssize_t)sizeof id) _exit(5);
            _exit(run_rank(r, ranks, n, tiles, batches, id, sum_wr[r], nullptr));
        }
        kids.push_back(pid);
    }
    const int rank = 0;
    unsigned char id[OCEAN_COMM_ID_BYTES];
    CHECK(ocean_comm_unique_id(id));
    for (int r = 1; r < ranks; ++r)
        if (write(id_wr[r], id, sizeof id) != (ssize_t)sizeof id) return 6;
    int rc = run_rank(0, ranks, n, tiles, batches, id, -1, sum_rd.data());
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 20;
    }
    return rc;
}
