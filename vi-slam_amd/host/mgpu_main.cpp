// mgpu_main.cpp -- the C++ multi-GPU launcher of the frame-sharded path (BASELINE configs[3]): host code in
// C++ like the reference's (src/main_vi_slamGPU.cpp), HIP through the C ABI, the ONE collective through RCCL directly.
//
//   vislam_mgpu --gpus N [--steps K] [--warmup W] [--batch B] [--timeout SECONDS]
//
// The parent never touches the GPU: it forks N ranks (one process per GPU, rank r <-> device r <-> camera stream r).
// Rank 0 creates the ncclUniqueId and hands it to the others through a file; every rank then
//   1. ncclCommInitRank, 2. ncclBroadcast of the vis_params POD from rank 0 ("RCCL broadcast of intrinsics only", < 256 B),
//   3. generates its own S-752 stream on its device (seed 0xE0C00010 + r), plans, warms up,
//   4. barrier (ncclAllReduce of one int), times K steps of vis_batch_run over frames resident in HBM, barrier,
//   5. ncclAllReduce(MAX) of the elapsed time; rank 0 prints one JSON line: frames of ALL ranks / slowest rank's time.
// Failure handling: every rank checks the visible device count before it touches the communicator; the parent reaps its ranks
// as they exit (waitpid(-1)), and on the FIRST rank that fails, is killed or outlives --timeout it kills the others (a rank whose
// peer died would otherwise block in ncclCommInitRank / a collective forever) and returns that rank's code.  Ranks die with the
// parent (PR_SET_PDEATHSIG).  Nothing is ever re-exec'ed.
// No data-path collective exists: streams are independent (a frame needs only the previous frame of its own stream,
// src/Camera.cpp:149-150); xGMI bandwidth is irrelevant at this payload.  bench.py (the driver's contract) does the same
// through torch.distributed; this program is the path a C++ deployment of the reference would use.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/vislam_hip.h"

#define CK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_)); return 3; } } while (0)
#define CK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", rank, #x, ncclGetErrorString(r_)); return 4; } } while (0)
#define CK_VIS(x) do { int r_ = (x); if (r_ != VIS_OK) { std::fprintf(stderr, "rank %d: %s: %s (%s)\n", rank, #x, vis_strerror(r_), ctx ? vis_last_error(ctx) : ""); return 5; } } while (0)

static int run_rank(int rank, int world, int steps, int warmup, int B, const std::string& id_path) {
    vis_ctx* ctx = nullptr;
    const int W = 752, H = 480, R = 2, DIM = 4096;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    if (ndev < world) { std::fprintf(stderr, "rank %d: --gpus %d but %d HIP device(s) visible\n", rank, world, ndev); return 3; }
    CK_HIP(hipSetDevice(rank));
    // ---- communicator: rank 0 publishes the unique id, the others poll for the file
    ncclUniqueId id;
    if (rank == 0) {
        CK_NCCL(ncclGetUniqueId(&id));
        const std::string tmp = id_path + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) { std::fprintf(stderr, "rank 0: cannot write %s\n", tmp.c_str()); return 2; }
        std::fclose(f);
        std::rename(tmp.c_str(), id_path.c_str());
    } else {
        FILE* f = nullptr;
        for (int tries = 0; tries < 6000 && !(f = std::fopen(id_path.c_str(), "rb")); tries++) usleep(10000);
        if (!f || std::fread(&id, sizeof(id), 1, f) != 1) { std::fprintf(stderr, "rank %d: no unique id\n", rank); return 2; }
        std::fclose(f);
    }
    ncclComm_t comm;
    CK_NCCL(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t cs;
    CK_HIP(hipStreamCreate(&cs));
    // ---- parameters: rank 0 owns them, one broadcast of the POD
    vis_params p;
    std::memset(&p, 0, sizeof(p));
    if (rank == 0) { vis_default_params(&p); p.fy = p.fx; }
    void* d_p = nullptr;
    CK_HIP(hipMalloc(&d_p, sizeof(p)));
    CK_HIP(hipMemcpy(d_p, &p, sizeof(p), hipMemcpyHostToDevice));
    CK_NCCL(ncclBroadcast(d_p, d_p, sizeof(p), ncclChar, 0, comm, cs));
    CK_HIP(hipStreamSynchronize(cs));
    CK_HIP(hipMemcpy(&p, d_p, sizeof(p), hipMemcpyDeviceToHost));
    // ---- this rank's stream, generated on its own device
    CK_VIS(vis_create(rank, &ctx));
    CK_VIS(vis_set_params(ctx, &p));
    const unsigned long long seed = world == 1 ? 0xE0C00001ULL : 0xE0C00010ULL + (unsigned)rank;
    std::vector<uint8_t> canvas((size_t)DIM * DIM);
    CK_VIS(vis_synth_canvas(canvas.data(), DIM, seed));
    uint8_t* d_canvas = nullptr; uint8_t* d_frames = nullptr;
    CK_HIP(hipMalloc((void**)&d_canvas, canvas.size()));
    CK_HIP(hipMemcpy(d_canvas, canvas.data(), canvas.size(), hipMemcpyHostToDevice));
    const size_t fb = (size_t)W * H;
    CK_HIP(hipMalloc((void**)&d_frames, fb * B * R));
    for (int t0 = 0; t0 < B * R; t0 += 256)
        CK_VIS(vis_synth_frames_device(ctx, d_canvas, DIM, seed, t0, std::min(256, B * R - t0), W, H, W, 0, d_frames + fb * t0));
    CK_VIS(vis_batch_plan(ctx, W, H, W, B));
    for (int i = 0; i < warmup; i++) CK_VIS(vis_batch_run(ctx, d_frames + fb * B * (i % R), B, VIS_STAGE_FRAME));
    CK_VIS(vis_batch_sync(ctx));
    int flags = 0;
    CK_VIS(vis_batch_status(ctx, &flags));
    if (flags) { std::fprintf(stderr, "rank %d: device capacity flag %d\n", rank, flags); return 6; }
    // ---- barrier, timed region, barrier
    int* d_one = nullptr; double* d_t = nullptr;
    CK_HIP(hipMalloc((void**)&d_one, sizeof(int))); CK_HIP(hipMalloc((void**)&d_t, sizeof(double)));
    CK_HIP(hipMemset(d_one, 0, sizeof(int)));
    CK_HIP(hipDeviceSynchronize());
    CK_NCCL(ncclAllReduce(d_one, d_one, 1, ncclInt, ncclSum, comm, cs)); CK_HIP(hipStreamSynchronize(cs));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; i++) CK_VIS(vis_batch_run(ctx, d_frames + fb * B * ((warmup + i) % R), B, VIS_STAGE_FRAME));
    CK_VIS(vis_batch_sync(ctx));
    CK_HIP(hipDeviceSynchronize());
    CK_NCCL(ncclAllReduce(d_one, d_one, 1, ncclInt, ncclSum, comm, cs)); CK_HIP(hipStreamSynchronize(cs));
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double dt_rank = dt;
    CK_HIP(hipMemcpy(d_t, &dt, sizeof(double), hipMemcpyHostToDevice));
    CK_NCCL(ncclAllReduce(d_t, d_t, 1, ncclDouble, ncclMax, comm, cs)); CK_HIP(hipStreamSynchronize(cs));
    CK_HIP(hipMemcpy(&dt, d_t, sizeof(double), hipMemcpyDeviceToHost));
    CK_VIS(vis_batch_status(ctx, &flags));
    if (flags) { std::fprintf(stderr, "rank %d: device capacity flag %d\n", rank, flags); return 6; }
    // ---- who ran where: one ncclAllGather of a 64-byte POD per rank behind the timed region (rank, HIP device, PCI address, frames/s,
    // seconds); rank 0 prints them and refuses a run in which two ranks shared a device
    struct RankRec { int32_t rank, device; double fps, seconds; char bus[32]; char pad[8]; };
    static_assert(sizeof(RankRec) == 64, "the layout of vislam/dist.py pack_rank_record");
    RankRec mine; std::memset(&mine, 0, sizeof(mine));
    mine.rank = rank; mine.device = rank; mine.fps = (double)steps * B / dt_rank; mine.seconds = dt_rank;
    (void)vis_device_pci_bus_id(rank, mine.bus, (int)sizeof(mine.bus));
    RankRec* d_rec = nullptr;
    CK_HIP(hipMalloc((void**)&d_rec, sizeof(RankRec) * (size_t)(world + 1)));
    CK_HIP(hipMemcpy(d_rec + world, &mine, sizeof(mine), hipMemcpyHostToDevice));
    CK_NCCL(ncclAllGather(d_rec + world, d_rec, sizeof(RankRec), ncclChar, comm, cs)); CK_HIP(hipStreamSynchronize(cs));
    std::vector<RankRec> recs((size_t)world);
    CK_HIP(hipMemcpy(recs.data(), d_rec, sizeof(RankRec) * (size_t)world, hipMemcpyDeviceToHost));
    (void)hipFree(d_rec);
    std::string ranks_json = "[";
    for (int i = 0; i < world; i++) {
        for (int j = 0; j < i; j++)
            if (std::strncmp(recs[i].bus, recs[j].bus, sizeof(recs[i].bus)) == 0 && recs[i].bus[0]) { std::fprintf(stderr, "ranks %d and %d share device %s\n", j, i, recs[i].bus); return 7; }
        char buf[256];
        std::snprintf(buf, sizeof(buf), "%s{\"rank\": %d, \"device\": %d, \"pci_bus_id\": \"%.31s\", \"frames_per_s\": %.3f, \"seconds\": %.6f}", i ? ", " : "",
                      recs[i].rank, recs[i].device, recs[i].bus, recs[i].fps, recs[i].seconds);
        ranks_json += buf;
    }
    ranks_json += "]";
    if (rank == 0)
        std::printf("{\"metric\": \"frames/sec detect+match+pose, 752x480 mono8\", \"value\": %.3f, \"unit\": \"frames/s\", \"n_gpus\": %d, \"steps\": %d, "
                    "\"warmup\": %d, \"ms_per_step\": %.6f, \"higher_is_better\": true, \"scaling\": \"weak\", \"dtype\": \"u8\", \"data\": \"synthetic\", "
                    "\"config\": {\"workload\": \"S-752 stream per rank, launcher = C++ (vislam_mgpu), RCCL: 1 broadcast of vis_params (%zu B) + timing reductions\", "
                    "\"frames_per_step_per_gpu\": %d}, \"ranks\": %s}\n",
                    (double)world * steps * B / dt, world, steps, warmup, dt / steps * 1e3, sizeof(vis_params), B, ranks_json.c_str());
    std::fflush(stdout);                                             // the rank leaves through _exit()
    vis_destroy(ctx);
    (void)hipFree(d_frames); (void)hipFree(d_canvas); (void)hipFree(d_p); (void)hipFree(d_one); (void)hipFree(d_t);
    (void)hipStreamDestroy(cs);
    ncclCommDestroy(comm);
    return 0;
}

int main(int argc, char** argv) {
    int gpus = 1, steps = 40, warmup = 5, B = 1024, timeout_s = 900;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--gpus") gpus = std::atoi(argv[i + 1]);
        else if (k == "--steps") steps = std::atoi(argv[i + 1]);
        else if (k == "--warmup") warmup = std::atoi(argv[i + 1]);
        else if (k == "--batch") B = std::atoi(argv[i + 1]);
        else if (k == "--timeout") timeout_s = std::atoi(argv[i + 1]);
    }
    if (gpus < 1 || gpus > 64 || steps < 1 || B < 1 || timeout_s < 1) {
        std::fprintf(stderr, "usage: vislam_mgpu --gpus N [--steps K] [--warmup W] [--batch B] [--timeout SECONDS]\n"); return 2;
    }
    char tmpl[] = "/tmp/vislam_nccl_id_XXXXXX";
    const int fd = mkstemp(tmpl);
    if (fd >= 0) { close(fd); unlink(tmpl); }
    const std::string id_path = tmpl;
    // one process per GPU, forked BEFORE anything initialises HIP in this process
    const pid_t parent = getpid();
    std::vector<pid_t> kids;
    for (int r = 0; r < gpus; r++) {
        const pid_t pid = fork();
        if (pid < 0) { std::perror("fork"); for (pid_t k : kids) kill(k, SIGKILL); return 2; }
        if (pid == 0) {
            prctl(PR_SET_PDEATHSIG, SIGKILL);                        // a rank never outlives the launcher
            if (getppid() != parent) _exit(2);
            _exit(run_rank(r, gpus, steps, warmup, B, id_path));
        }
        kids.push_back(pid);
    }
    // reap in completion order; first failure (or the wall-clock limit) ends the job for everyone
    int rc = 0, alive = gpus;
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s);
    while (alive > 0) {
        int st = 0;
        const pid_t done = waitpid(-1, &st, WNOHANG);
        if (done > 0) {
            alive--;
            for (pid_t& k : kids) if (k == done) k = -1;
            const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
            if (code != 0 && rc == 0) {
                rc = code;
                std::fprintf(stderr, "vislam_mgpu: a rank ended with code %d: stopping the other %d\n", code, alive);
                for (pid_t k : kids) if (k > 0) kill(k, SIGKILL);
            }
            continue;
        }
        if (done < 0) break;                                         // no children left
        if (std::chrono::steady_clock::now() > t_end) {
            if (rc == 0) rc = 124;
            std::fprintf(stderr, "vislam_mgpu: %d s wall-clock limit reached: stopping %d rank(s)\n", timeout_s, alive);
            for (pid_t k : kids) if (k > 0) kill(k, SIGKILL);
            while (waitpid(-1, &st, 0) > 0) { }
            break;
        }
        usleep(20000);
    }
    unlink(id_path.c_str());
    unlink((id_path + ".tmp").c_str());
    return rc;
}
