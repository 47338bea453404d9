// vis_internal.h -- internal declarations of libvislam_hip (MI355X / gfx950 only).
// Product code: never includes or links anything under oracle/.
#ifndef VIS_INTERNAL_H_
#define VIS_INTERNAL_H_

#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/vislam_hip.h"

#define VIS_BATCH_SETS 2          // record sets of a batch plan (see Plan)
#define VIS_NSLOTS 32            // device keyframe slots of the single-frame API (Camera::frameList)
#define VIS_RANSAC_MAX_M 8192    // max correspondences per RANSAC problem
#define VIS_MAX_MODELS 10
// k_fast (detect.hip) works on ITEMS: a strip of 32 lanes x 4 pixels = 128 pixel columns (120 of them emit; 1 score halo + 3 ring
// columns on either side) marched down 8 score rows at a time, at most VIS_FS_NCH chunks = 8 * NCH - 2 emitting rows.  An item owns a
// candidate slot sized by the 3x3-NMS bound of its emit region (60 x 31 = 1860 <= 2048): a slot cannot overflow.
#define VIS_FS_EMIT_W 120
#define VIS_FS_NCH 8
#define VIS_FS_EMIT_H (8 * VIS_FS_NCH - 2)
#define VIS_TILE_CAND_CAP 2048
static_assert(((VIS_FS_EMIT_W + 1) / 2) * ((VIS_FS_EMIT_H + 1) / 2) <= VIS_TILE_CAND_CAP, "NMS bound of an item");

struct LevelInfo {
    int w, h, stride;            // stride: bytes per row of the level buffer (level 0: caller's)
    int quota;                   // nfeaturesPerLevel
    float scale;                 // layerScale
    size_t frame_bytes;          // stride*h
    int cand_cap;                // tiles * VIS_TILE_CAND_CAP: every FAST item owns a slot sized by the 3x3-NMS bound
    int tile_base;               // first global tile index of this level
    int surv_cap;                // survivors of the FAST cut (2*quota + ties), LDS sort size
    int keep_cap;                // kept per level (quota + ties)
    int tiles_x, tiles_y;        // FAST items: strips x segments
    int wave_base, nwaves;       // k_fast waves of this level (two items each) in the plan's FastWave table
};

// device-resident compact kNN entry: key = (dist << 16) | trainIdx, 0xFFFFFFFF = none
typedef vis_pose_result PoseOut;       // E, R, t (double) + n_inliers, n_pose_good, iters_run, n_points, n_models

struct Plan {
    int w = 0, h = 0, stride = 0, B = 0, L = 0;
    int fs_nch = VIS_FS_NCH;     // chunks (8 score rows) per k_fast segment of this plan: 8 for batches, fewer for a handful of frames (more, shorter waves)
    int nrec = 0;                // keypoint/descriptor records (batch: B+1, single: VIS_NSLOTS)
    int kcap = 0;                // keypoints per record (sum of keep_cap)
    int npairs = 0;              // pair result capacity
    int root = 0;                // grid root
    int pose_mcap = 0;           // correspondences per pair the pose stage is sized for (root^2, or kcap with VIS_POSE_SYM)
    LevelInfo lv[VIS_MAX_LEVELS];
    // ---- device buffers
    uint8_t* d_stage = nullptr;              // single-frame upload staging (stride x h)
    uint8_t* d_pyr[VIS_MAX_LEVELS] = {};     // level l >= 1: B x h_l x stride_l
    uint32_t* d_rs_tab[VIS_MAX_LEVELS] = {}; // level l >= 1: coefficient tables of the resize step l-1 -> l (detect.hip k_resize_tab; built on first use)
    uint32_t* d_cand[VIS_MAX_LEVELS] = {};   // B x tiles_l x VIS_TILE_CAND_CAP packed (score<<24 | y<<12 | x)
    int32_t* d_tile_cnt = nullptr;           // B x total_tiles candidates per tile
    int total_tiles = 0;
    void* d_fast_tiles = nullptr;            // total_waves x FastWave (detect.hip): per-wave record of k_fast (two items of one level)
    int total_waves = 0;
    int32_t* d_seg_cnt = nullptr;            // B x L kept counts
    float4*  d_seg_kp[VIS_MAX_LEVELS] = {};  // B x keep_cap (x, y, response, unused)
    int32_t* d_flags = nullptr;              // device error flags (1 word)
    uint32_t* d_angle_tab = nullptr;         // IC-angle byte weight/mask table for k_describe
    // speculative FAST threshold of batched streams (detect.hip fast_tile): per-level prediction, per-(frame, level) cuts of the
    // last batch, and the work list of the pairs whose prediction was too high
    bool speculate = false;
    int32_t* d_tau = nullptr;                // L
    int32_t* d_seg_cut = nullptr;            // B x L
    int32_t* d_fix = nullptr;                // 1 + B x L
    // VIS_STAGE_UPDATE: B x vis_grad_frame_elems(w, h) half pyramids; VIS_STAGE_GRADIENT: the Scharr gradients of the batch
    // (Camera::computeGradient) beside the detect chain.  Plan-owned, allocated on first use, TWO sets used in turn: vis_batch_align of
    // step i reads set i & 1 on the pose stream while the side stream of step i + 1 fills the other one (with one set the alignment, the
    // next step's gradients and -- through the detect stream's join -- the next matcher formed a serial chain as long as the step).
    // d_half / d_gx / d_gy / d_g = the set the last vis_batch_run wrote (not owned); grad_reader[s] = the event of the last alignment
    // that read set s (nullptr: none pending).
    uint8_t* d_half_set[2] = {nullptr, nullptr}; int16_t* d_gx_set[2] = {nullptr, nullptr}; int16_t* d_gy_set[2] = {nullptr, nullptr}; uint8_t* d_g_set[2] = {nullptr, nullptr};
    int grad_set = 0;
    hipEvent_t grad_reader[2] = {nullptr, nullptr};
    uint8_t* d_half = nullptr;
    bool half_valid = false;
    int16_t* d_gx = nullptr; int16_t* d_gy = nullptr; uint8_t* d_g = nullptr;
    bool grad_valid = false;
    // records
    vis_keypoint* d_kps = nullptr;           // nrec x kcap
    uint8_t* d_desc = nullptr;               // nrec x kcap x 32
    int32_t* d_nkp = nullptr;                // nrec
    int8_t* d_descx = nullptr;               // nrec x kcap x 128: descriptor bits as FP4 (e2m1) +1/-1, two per byte (MFMA matcher operand)
    // pairs
    int32_t* d_pair_q = nullptr;             // npairs: query record index (-1 = no pair)
    int32_t* d_pair_t = nullptr;
    uint32_t* d_knn12 = nullptr;             // npairs x kcap x 2 keys
    uint32_t* d_knn21 = nullptr;
    vis_dmatch* d_sym = nullptr;             // npairs x kcap
    int32_t* d_nsym = nullptr;
    vis_dmatch* d_good = nullptr;            // npairs x root^2
    int32_t* d_ngood = nullptr;
    float* d_p1 = nullptr;                   // npairs x root^2 x 2
    float* d_p2 = nullptr;
    // The six arrays above are what k_filter writes and the pose stage / vis_batch_align / the results download read.  A batch plan
    // owns TWO sets of them, used in turn like the record sets: with one set the filter of step i + 1 had to wait for the pose stage and
    // the download of step i, and the pose stage of step i + 1 for that filter -- a serial loop (pose chain + download + filter) that
    // set the step's period as soon as it grew longer than the detect chain (results download: - 8 %).  d_sym ... d_p2 = the set of
    // the last vis_batch_run (not owned); mo_set[s] = {sym, nsym, good, ngood, p1, p2} of set s (set 0 only for single-frame plans);
    // mo_pose / mo_results / mo_align[s] = the event behind the last reader of set s on the pose stream (nullptr: none pending).
    void* mo_set[2][6] = {{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}};
    hipEvent_t mo_pose[2] = {nullptr, nullptr}, mo_results[2] = {nullptr, nullptr}, mo_align[2] = {nullptr, nullptr};
    int last_cur = 0;
    float* d_hf = nullptr;                   // root band limits (float accumulation on host)
    float* d_wf = nullptr;
    // pose
    double* d_n1 = nullptr;                  // npairs x root^2 x 2 normalised points
    double* d_n2 = nullptr;
    uint8_t* d_mask = nullptr;               // npairs x root^2 inlier mask of the winning E
    int32_t* d_pair_q_noprev = nullptr;      // same as d_pair_q with pair 0 disabled
    int32_t* d_samples = nullptr;            // npairs x max_iters x 5
    double* d_models = nullptr;              // npairs x max_iters x 10 x 9
    int32_t* d_counts = nullptr;             // npairs x max_iters x 10  (-1 = no model)
    int32_t* d_rstate = nullptr;             // npairs x 4: niters, maxGood, bestIter, bestModel
    PoseOut* d_pose = nullptr;               // npairs
    int32_t* d_worklist = nullptr;           // 2 + npairs: count, pairs that need RANSAC chunks beyond the first, item counter of the roots kernel
    double* d_hyp = nullptr;                 // npairs x max_iters hypothesis records (VIS_HYP_DOUBLES each, element-major)
    int max_iters = 0;
    bool have_prev = false;                  // batch: record 0 holds the previous batch's last frame
    int last_n = 0;                          // frames in the last batch
    int carry_from = 0;                      // absolute record to copy into the next set's record 0 (0 = none)
    // batch plans keep VIS_BATCH_SETS sets of records (keypoints, descriptors, expanded descriptors) and walk them round-robin: the detect
    // chain of batch i + SETS waits for the matcher of batch i.  (Round 5: three sets instead of two change nothing -- in steady state the
    // low-priority matcher / pose streams progress only as fast as the detect chain leaves them room, and the detect stream ends up waiting
    // for them however far ahead it may run: 393.4 k frames/s with three sets against 395-400 k with two, same build.)
    int nsets = 1, rec_per_set = 0, run_count = 0, last_base = 0;
    int32_t* d_pq[VIS_BATCH_SETS] = {}; int32_t* d_pt[VIS_BATCH_SETS] = {}; int32_t* d_pqn[VIS_BATCH_SETS] = {};
    bool match_pending[VIS_BATCH_SETS] = {};
    bool pair0_valid = false;                // last run: frame 0 had a predecessor
};

#define VIS_POSE_TABLE_M 64                  // the frame-at-a-time pose entry points take their RANSAC samples from the table for M <= 64 (root^2 = 49 in the reference)
struct vis_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t pose_stream = nullptr;       // RANSAC/pose of batch i overlaps detect/match of batch i+1
    hipEvent_t ev_filter_done = nullptr, ev_pose_done = nullptr, ev_pose_start = nullptr;
    hipStream_t match_stream = nullptr;      // knn + filters of batch i overlap the detect chain of batch i+1
    hipEvent_t ev_detect_done = nullptr, ev_match_start = nullptr, ev_match_done[VIS_BATCH_SETS] = {};
    hipStream_t update_stream = nullptr;     // VIS_STAGE_UPDATE (Camera::Update): streaming work beside the VALU-bound detect chain
    hipEvent_t ev_update_fork = nullptr, ev_update_done = nullptr;
    bool pose_pending = false;
    // vis_batch_align runs on the pose stream (beside the next batch's detect chain): what may not overtake it waits for this event
    // (ev_align_done = the event of the LAST alignment, one of ev_align_done2[] used in turn, so that a gradient set can wait for its own reader)
    hipEvent_t ev_align_fork = nullptr, ev_align_done = nullptr, ev_align_done2[2] = {nullptr, nullptr}; int align_k = 0; bool align_pending = false;
    bool pose_attr_set = false;              // > 64 KiB LDS opt-in of the RANSAC solver kernels done on this context's device
    bool pose_grids_set = false; int pose_grid[4] = {0, 0, 0, 0};   // resident-workgroup grids of the work-list pose kernels on this device (pose.hip pose_grids)
    hipEvent_t ev_pose_done_set[2] = {nullptr, nullptr}, ev_results_done_set[2] = {nullptr, nullptr};   // per match-output set (Plan::mo_set)
    hipEvent_t ev_results_done = nullptr; bool results_pending = false;   // D2H of the last batch's results (vis_batch_results_async)
    vis_params p;
    std::string err;
    Plan* single = nullptr;
    Plan* batch = nullptr;
    vis_timings tm;
    hipEvent_t ev[12];
    bool ev_ok = false;
    // grow-only scratch for the *_host entry points
    void* d_scratch = nullptr; size_t scratch_bytes = 0;
    // grow-only PINNED host block of the single-frame entry points: a caller's pageable buffer is copied through it, so that every
    // upload / download of a call is an asynchronous copy on the context's stream and the call blocks ONCE, at its end (HostStage, api.hip)
    void* h_pin = nullptr; size_t h_pin_bytes = 0; void* h_pin_dev = nullptr;   // (h_pin_dev: the block's device address; nullptr = not device-accessible)
    int stage_live = 0;                      // HostStage objects alive on this context: vis_ensure_pin refuses to replace the block under one (VIS_E_STATE)
    // diagnostics of the single-frame path (vis_debug_counters): times the host blocked on the device / copies queued since the context was made
    unsigned long long n_host_waits = 0, n_copies = 0;
    int slot_valid[VIS_NSLOTS];
    // cv::RNG sample tables for M in [6, sample_max_m], built on the host for (seed, max_iters)
    int32_t* d_sample_table = nullptr; int sample_max_m = 0; int sample_iters = 0; unsigned long long sample_seed = 0;
};

// roctx ranges around the stage families of a batched step (readable rocprofv3 --marker-trace timelines); compiled in only by
// `make ROCTX=1` (-DVIS_HAVE_ROCTX -lrocprofiler-sdk-roctx: a diagnostic build), otherwise no-ops
#ifdef VIS_HAVE_ROCTX
#include <rocprofiler-sdk-roctx/roctx.h>
struct VisRange { explicit VisRange(const char* n) { roctxRangePushA(n); } ~VisRange() { roctxRangePop(); } };
#else
struct VisRange { explicit VisRange(const char*) {} };
#endif

// every kernel launch of the library is counted (vis_debug_counters: launches per frame of the single-frame path, bench.py's
// single_frame_api leg): one relaxed atomic increment beside a launch that costs microseconds (contexts of different host threads
// launch concurrently: a plain global would be a data race)
extern std::atomic<unsigned long long> vis_g_launches;
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    do { vis_g_launches.fetch_add(1, std::memory_order_relaxed); kernel<<<(grid), (block), (lds), (stream)>>>(__VA_ARGS__); } while (0)

#define HIPCHK(ctx, call)                                                          \
    do { hipError_t e_ = (call);                                                   \
         if (e_ != hipSuccess) { (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); \
                                 return VIS_E_HIP; } } while (0)

// carve typed, 256-byte aligned pieces out of the context scratch block
struct Carver {
    char* base; size_t off;
    template <class T> T* take(size_t count) { off = (off + 255) & ~(size_t)255; T* p = (T*)(base + off); off += count * sizeof(T); return p; }
};
int vis_ensure_scratch(vis_ctx* ctx, size_t bytes);

// ---- host staging of the single-frame entry points (api.hip explains; vis_ensure_pin grows the context's pinned block) ----
#include <cstring>
#include <algorithm>
int vis_ensure_pin(vis_ctx* ctx, size_t bytes);
int launch_copy_jobs(vis_ctx* ctx, hipStream_t st, int njobs, void* const* dst, const void* const* src, const size_t* bytes);
// The block may only be replaced (vis_ensure_pin frees and re-allocates it when it grows) while NO HostStage is alive: a stage hands out
// addresses inside the block (take / down) that the caller reads after wait().  Round 5's host SIGSEGV (gpurun_out/r5r_gdb.log: libc's copy
// faulting on its first destination byte, 0x7ff600c00000 = 2 MiB-aligned, unmapped, 360 960 bytes = the first up2d of a call, i.e. offset 0 of
// a block that was no longer there) was a stage built on a block that a later vis_ensure_pin of the same call freed; every entry point sizes
// the block BEFORE it builds its stage since f950f5e, and since round 6 the rule is enforced instead of kept by convention: the stage reads
// the block's address from the context at every use, counts itself in ctx->stage_live, and vis_ensure_pin answers VIS_E_STATE instead of
// freeing a block under a live stage (vi-slam_amd/host/stage_selftest.cpp drives both refusals on the CPU).
struct HostStage {
    vis_ctx* ctx; size_t off = 0; hipError_t err = hipSuccess; bool overflow = false;
    explicit HostStage(vis_ctx* c) : ctx(c) { ctx->stage_live++; }
    ~HostStage() { ctx->stage_live--; }
    HostStage(const HostStage&) = delete; HostStage& operator=(const HostStage&) = delete;
    char* base_now() const { return (char*)ctx->h_pin; }
    void* take(size_t bytes) {
        off = (off + 63) & ~(size_t)63;
        if (!ctx->h_pin || off + bytes > ctx->h_pin_bytes) { overflow = true; return nullptr; }
        void* p = base_now() + off; off += bytes; return p;
    }
    // host (pageable) -> device, asynchronous: through the pinned block.  Dword-granular uploads are collected like the downloads and
    // read out of the block by one kernel of the library per six of them -- at flush_ups(), which every entry point calls behind its
    // last upload, in front of its first kernel (down() and wait() flush too, so a forgotten call shows as a wrong result in the parity
    // tests, not as a missing transfer).
    void* up_dst[6]; const void* up_src[6]; size_t up_bytes[6]; int up_n = 0;
    void flush_ups() {
        if (!up_n) return;
        if (launch_copy_jobs(ctx, ctx->stream, up_n, up_dst, up_src, up_bytes) != VIS_OK && err == hipSuccess) err = hipErrorLaunchFailure;
        ctx->n_copies++;
        up_n = 0;
    }
    void up(void* d, const void* h, size_t bytes) {
        if (!bytes) return;
        void* p = take(bytes);
        if (!p) return;
        std::memcpy(p, h, bytes);
        if (!(bytes & 3) && !((uintptr_t)d & 3) && ctx->h_pin_dev) {
            if (up_n == 6) flush_ups();
            up_dst[up_n] = d; up_src[up_n] = (const char*)ctx->h_pin_dev + ((char*)p - base_now()); up_bytes[up_n] = bytes; up_n++;
            return;
        }
        flush_ups();                                               // (stream order among the uploads)
        const hipError_t e = hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess && err == hipSuccess) err = e;
        ctx->n_copies++;
    }
    // rows of an image with a host stride -> a device image with its own stride
    void up2d(void* d, size_t dpitch, const void* h, size_t hpitch, size_t width, size_t height) {
        char* p = (char*)take(width * height);
        if (!p) return;
        for (size_t y = 0; y < height; y++) std::memcpy(p + y * width, (const char*)h + y * hpitch, width);
        if (dpitch == width && !((width * height) & 3) && !((uintptr_t)d & 3) && ctx->h_pin_dev) {      // dense on the device: one copy job
            if (up_n == 6) flush_ups();
            up_dst[up_n] = d; up_src[up_n] = (const char*)ctx->h_pin_dev + (p - base_now()); up_bytes[up_n] = width * height; up_n++;
            return;
        }
        flush_ups();
        const hipError_t e = hipMemcpy2DAsync(d, dpitch, p, width, width, height, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess && err == hipSuccess) err = e;
        ctx->n_copies++;
    }
    // device -> the pinned block, asynchronous; valid after wait().  Dword-granular downloads are collected and written by ONE kernel of
    // the library per six of them (launch_copy_jobs; the block is device-accessible) when wait() is called -- every call site requests
    // its downloads right in front of its wait(), behind the kernels that produce them -- instead of one runtime copy each.
    void* dn_dst[6]; const void* dn_src[6]; size_t dn_bytes[6]; int dn_n = 0;
    void flush_downs() {
        if (!dn_n) return;
        if (launch_copy_jobs(ctx, ctx->stream, dn_n, dn_dst, dn_src, dn_bytes) != VIS_OK && err == hipSuccess) err = hipErrorLaunchFailure;
        ctx->n_copies++;
        dn_n = 0;
    }
    void* down(const void* d, size_t bytes) {
        flush_ups();
        void* p = take(std::max(bytes, (size_t)4));
        if (!p || !bytes) return p;
        if (!(bytes & 3) && !((uintptr_t)d & 3) && ctx->h_pin_dev) {
            if (dn_n == 6) flush_downs();
            dn_dst[dn_n] = (char*)ctx->h_pin_dev + ((char*)p - base_now()); dn_src[dn_n] = d; dn_bytes[dn_n] = bytes; dn_n++;
            return p;
        }
        const hipError_t e = hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess && err == hipSuccess) err = e;
        ctx->n_copies++;
        return p;
    }
    // the one place a single-frame entry point blocks
    int wait() {
        if (overflow) { ctx->err = "host staging block too small (internal)"; return VIS_E_NOMEM; }
        flush_ups(); flush_downs();
        if (err != hipSuccess) { ctx->err = std::string("staged copy: ") + hipGetErrorString(err); return VIS_E_HIP; }
        ctx->n_host_waits++;
        const hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); return VIS_E_HIP; }
        return VIS_OK;
    }
};


// ---- host-side geometry / tables (geometry.cpp) ----
int  vis_compute_levels(const vis_params& p, int w, int h, int stride0, LevelInfo* lv, int fs_nch = VIS_FS_NCH);
void vis_grid_limits(const vis_params& p, int* root, std::vector<float>& hf, std::vector<float>& wf);

struct SynthOrigin;
int  vis_synth_origin(int dim, uint64_t seed, int t, int w, int h, SynthOrigin* o);

// ---- plan management (plan.hip) ----
int  plan_create(vis_ctx* ctx, int w, int h, int stride, int B, int nrec, int npairs, Plan** out, int nsets = 1);
void plan_destroy(Plan* pl);

// ---- kernel launchers (each enqueues on ctx->stream) ----
int launch_copy_jobs(vis_ctx* ctx, hipStream_t st, int njobs, void* const* dst, const void* const* src, const size_t* bytes);   // detect.hip: <= 6 dword-granular copies in one launch
int launch_detect(vis_ctx* ctx, Plan* pl, const uint8_t* d_frames, int n, int rec0, int carry_rec = -1, hipEvent_t after_resize = nullptr, hipEvent_t records_free = nullptr);   // carry_rec >= 0: copy that record to rec0 - 1 before k_describe; after_resize: recorded behind the pyramid launches; records_free: waited for before the chain's first write to the record set
int build_fast_tiles(vis_ctx* ctx, Plan* pl);
int launch_expand(vis_ctx* ctx, Plan* pl, int rec_first, int rec_count);
int launch_match(vis_ctx* ctx, Plan* pl, int npairs);
int launch_filter(vis_ctx* ctx, Plan* pl, int npairs);
int launch_pose(vis_ctx* ctx, Plan* pl, int npairs);
int launch_half_pyramid(vis_ctx* ctx, const uint8_t* d_img, int w, int h, int stride, uint8_t* d_out[5]);
// gradient.hip: Camera::Update / computeGradient / patch builders, batched
size_t vis_grad_frame_elems(int w, int h);
// cvRound(n * 0.5), round half to even: the size cv::resize(src, dst, Size(), 0.5, 0.5) gives the next level (include/vislam_hip.h)
static inline int vis_half_dim(int n) { return (n >> 1) + ((n & 1) & ((n >> 1) & 1)); }
static inline void vis_half_dims(int w, int h, int lw[5], int lh[5]) {
    lw[0] = w; lh[0] = h;
    for (int l = 1; l < 5; l++) { lw[l] = vis_half_dim(lw[l - 1]); lh[l] = vis_half_dim(lh[l - 1]); }
}
int launch_half_pyramid_batch(vis_ctx* ctx, const uint8_t* d_frames, int w, int h, int stride, size_t frame_bytes, int n, uint8_t* d_pyr);
int launch_gradient(vis_ctx* ctx, const uint8_t* d_frames, int w, int h, int stride, size_t frame_bytes, int n,
                    const uint8_t* d_pyr, int scale, int16_t* d_gx, int16_t* d_gy, uint8_t* d_g);
int launch_patch_points(vis_ctx* ctx, const vis_keypoint* d_good, int n, int w, int h, float* d_patch, float* d_debug, int cap,
                        int32_t* d_counts);
#define VIS_HYP_DOUBLES 97                // 96 doubles + 2 int32 per (pair, iteration): see pose.hip HR_*
int pose_run(vis_ctx* ctx, int npairs, int mcap, int max_iters, const float* d_p1, const float* d_p2, const int32_t* d_npts,
             double* d_n1, double* d_n2, int32_t* d_samples, double* d_models, int32_t* d_counts, int32_t* d_rstate,
             const double* d_E_in, uint8_t* d_mask, PoseOut* d_pose, int do_ransac, int do_pose, int32_t* d_worklist, double* d_hyp);
int  vis_build_sample_table(vis_ctx* ctx, int max_m);
int f2f_run(vis_ctx* ctx, const vis_keypoint* d_pts1, const vis_keypoint* d_pts2, int m, const float* d_rot,
            const int32_t* d_idx, int iters, double* d_nv, float* d_counts);
#define VIS_RSTATE_WORDS 16

#endif
