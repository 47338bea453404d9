// api.hip -- the C ABI of libvislam_hip.so (include/vislam_hip.h): context, plans, single-frame
// entry points (one per OpenCV-CUDA call site of the reference) and the batched stream path.
// Host code only; kernels live in detect.hip / match.hip / pose.hip.
#include "vis_internal.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <new>

static const char* VIS_VERSION_STR = "vislam_hip 0.1 (gfx950)";

extern "C" const char* vis_version(void) { return VIS_VERSION_STR; }

extern "C" const char* vis_strerror(int code) {
    switch (code) {
        case VIS_OK: return "ok";
        case VIS_E_INVALID: return "invalid argument";
        case VIS_E_NODEVICE: return "no HIP device";
        case VIS_E_HIP: return "HIP runtime error";
        case VIS_E_CAPACITY: return "capacity exceeded";
        case VIS_E_STATE: return "invalid state / call order";
        case VIS_E_NOMEM: return "out of memory";
        default: return "unknown error";
    }
}

// "dddd:bb:dd.f" of a HIP device: what the multi-GPU launchers gather per rank to show that N ranks ran on N different devices
extern "C" int vis_device_pci_bus_id(int device, char* out, int len) {
    if (!out || len < 16) return VIS_E_INVALID;
    out[0] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return VIS_E_NODEVICE; }
    if (device < 0 || device >= n) return VIS_E_INVALID;
    if (hipDeviceGetPCIBusId(out, len, device) != hipSuccess) { (void)hipGetLastError(); return VIS_E_HIP; }
    return VIS_OK;
}

extern "C" int vis_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void vis_default_params(vis_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->nfeatures = 1000; p->nlevels = 8; p->scale_factor = 1.2f; p->edge_threshold = 31; p->patch_size = 31;
    p->fast_threshold = 20; p->ratio = 0.8f; p->n_cells = 49; p->w_size = 752; p->h_size = 480;
    p->sym_mode = VIS_SYM_REFERENCE_EFFECTIVE;
    p->ransac_prob = 0.999; p->ransac_threshold = 1.0; p->ransac_max_iters = 1000; p->ransac_adaptive = 1;
    p->ransac_seed = 0xFFFFFFFFFFFFFFFFULL;
    p->fx = 458.654; p->fy = 457.296; p->cx = 367.215; p->cy = 248.375;     // calibrationEUROC.xml:20
    p->f2f_iters = 1000; p->f2f_threshold = 370.0;
    p->pose_input = VIS_POSE_GOOD;
}

static int validate_params(const vis_params& p) {
    if (p.nfeatures < 1 || p.nlevels < 1 || p.nlevels > VIS_MAX_LEVELS) return VIS_E_INVALID;
    if (!(p.scale_factor > 1.0f) || p.scale_factor > 3.0f) return VIS_E_INVALID;   // k_resize fetches a 12-byte source window per 4 outputs
    if (p.patch_size != 31) return VIS_E_INVALID;                 // the rBRIEF pattern is learned for 31x31
    if (p.edge_threshold < 22 || p.edge_threshold > 255) return VIS_E_INVALID;   // 43x43 raw patch must stay inside
    if (p.fast_threshold < 1 || p.fast_threshold > 254) return VIS_E_INVALID;
    if (p.n_cells < 1 || p.w_size < 1 || p.h_size < 1) return VIS_E_INVALID;
    if (p.ransac_max_iters < 1 || p.ransac_max_iters > 100000) return VIS_E_INVALID;
    if (!(p.ransac_prob > 0 && p.ransac_prob < 1) || !(p.fx > 0) || !(p.fy > 0)) return VIS_E_INVALID;
    if (p.f2f_iters < 0) return VIS_E_INVALID;
    if (p.pose_input != VIS_POSE_GOOD && p.pose_input != VIS_POSE_SYM) return VIS_E_INVALID;
    if (p.keypoint_capacity < 0 || p.keypoint_capacity > 65535) return VIS_E_INVALID;
    return VIS_OK;
}

extern "C" int vis_create(int device, vis_ctx** out) {
    if (!out) return VIS_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VIS_E_NODEVICE;
    if (device < 0 || device >= n) return VIS_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return VIS_E_HIP;
    vis_ctx* ctx = new (std::nothrow) vis_ctx();
    if (!ctx) return VIS_E_NOMEM;
    ctx->device = device;
    vis_default_params(&ctx->p);
    std::memset(&ctx->tm, 0, sizeof(ctx->tm));
    for (int i = 0; i < VIS_NSLOTS; i++) ctx->slot_valid[i] = 0;
    // Stream priorities of the batch pipeline.  Rounds 2-4: detect chain high, everything beside it low.  Round 5: with the streaming k_fast the
    // low-priority matcher / pose queues hardly got a turn (k_knn_mfma 1.4 ms of wall clock for 0.29 ms of work) and the detect stream ended
    // up waiting for them at the start of every chain; matcher and pose at the detect stream's priority: 397.1 k -> 402.7 k frames/s (three
    // alternations on one box; all four equal: 402.5 k).  Camera::Update's streaming side stream stays low.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);            // lo = numerically largest = least urgent
    if (hipStreamCreateWithPriority(&ctx->own_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) { delete ctx; return VIS_E_HIP; }
    ctx->stream = ctx->own_stream;
    const int prio_pose = prio_hi, prio_match = prio_hi;
    if (hipStreamCreateWithPriority(&ctx->pose_stream, hipStreamNonBlocking, prio_pose) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_filter_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_pose_done, hipEventDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_pose_start, hipEventDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_results_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_pose_done_set[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_pose_done_set[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_results_done_set[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_results_done_set[1], hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithPriority(&ctx->match_stream, hipStreamNonBlocking, prio_match) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_detect_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_match_start, hipEventDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_match_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_match_done[1], hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithPriority(&ctx->update_stream, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_update_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_update_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_align_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_align_done2[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_align_done2[1], hipEventDisableTiming) != hipSuccess) { delete ctx; return VIS_E_HIP; }
    ctx->ev_align_done = ctx->ev_align_done2[0];
    ctx->ev_ok = true;
    for (int i = 0; i < 12; i++) if (hipEventCreate(&ctx->ev[i]) != hipSuccess) ctx->ev_ok = false;
    *out = ctx;
    return VIS_OK;
}

extern "C" void vis_destroy(vis_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->match_stream) (void)hipStreamSynchronize(ctx->match_stream);
    if (ctx->pose_stream) (void)hipStreamSynchronize(ctx->pose_stream);
    plan_destroy(ctx->single); plan_destroy(ctx->batch);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->d_sample_table) (void)hipFree(ctx->d_sample_table);
    for (int i = 0; i < 12; i++) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->pose_stream) { (void)hipStreamSynchronize(ctx->pose_stream); (void)hipStreamDestroy(ctx->pose_stream); }
    if (ctx->ev_filter_done) (void)hipEventDestroy(ctx->ev_filter_done);
    if (ctx->ev_pose_done) (void)hipEventDestroy(ctx->ev_pose_done);
    if (ctx->ev_pose_start) (void)hipEventDestroy(ctx->ev_pose_start);
    if (ctx->ev_results_done) (void)hipEventDestroy(ctx->ev_results_done);
    for (int i = 0; i < 2; i++) { if (ctx->ev_pose_done_set[i]) (void)hipEventDestroy(ctx->ev_pose_done_set[i]); if (ctx->ev_results_done_set[i]) (void)hipEventDestroy(ctx->ev_results_done_set[i]); }
    if (ctx->ev_align_fork) (void)hipEventDestroy(ctx->ev_align_fork);
    for (int i = 0; i < 2; i++) if (ctx->ev_align_done2[i]) (void)hipEventDestroy(ctx->ev_align_done2[i]);
    if (ctx->match_stream) { (void)hipStreamSynchronize(ctx->match_stream); (void)hipStreamDestroy(ctx->match_stream); }
    if (ctx->ev_detect_done) (void)hipEventDestroy(ctx->ev_detect_done);
    if (ctx->ev_match_start) (void)hipEventDestroy(ctx->ev_match_start);
    for (int i = 0; i < VIS_BATCH_SETS; i++) if (ctx->ev_match_done[i]) (void)hipEventDestroy(ctx->ev_match_done[i]);
    if (ctx->update_stream) { (void)hipStreamSynchronize(ctx->update_stream); (void)hipStreamDestroy(ctx->update_stream); }
    if (ctx->ev_update_fork) (void)hipEventDestroy(ctx->ev_update_fork);
    if (ctx->ev_update_done) (void)hipEventDestroy(ctx->ev_update_done);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

static void sync_all(vis_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->update_stream) (void)hipStreamSynchronize(ctx->update_stream);      // (joined by the detect stream in vis_batch_run -- unless that call failed half way, or the caller swapped ctx->stream since)
    if (ctx->match_stream) (void)hipStreamSynchronize(ctx->match_stream);
    if (ctx->pose_stream) (void)hipStreamSynchronize(ctx->pose_stream);
    ctx->pose_pending = false; ctx->results_pending = false; ctx->align_pending = false;
    if (ctx->batch) { for (int i = 0; i < VIS_BATCH_SETS; i++) ctx->batch->match_pending[i] = false; ctx->batch->grad_reader[0] = ctx->batch->grad_reader[1] = nullptr;
                      for (int i = 0; i < 2; i++) ctx->batch->mo_pose[i] = ctx->batch->mo_results[i] = ctx->batch->mo_align[i] = nullptr; }
}

extern "C" const char* vis_last_error(vis_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// the fields that fix the per-record layout (keypoint capacity per level, descriptor pattern)
static bool same_record_geometry(const vis_params& a, const vis_params& b) {
    return a.nfeatures == b.nfeatures && a.nlevels == b.nlevels && a.scale_factor == b.scale_factor &&
           a.edge_threshold == b.edge_threshold && a.patch_size == b.patch_size && a.keypoint_capacity == b.keypoint_capacity;
}

extern "C" int vis_set_params(vis_ctx* ctx, const vis_params* p) {
    if (!ctx || !p) return VIS_E_INVALID;
    int rc = validate_params(*p);
    if (rc) return rc;
    sync_all(ctx);
    // The reference changes matcher / pose settings while its keyframes keep their keypoints and descriptors
    // (Matcher::setImageDimensions, computeBestMatches(n_cells)): when only such fields change, the device slots of the
    // single-frame API survive -- the records move into the re-created plan.  A change of the detector geometry drops them
    // (vis_bf_knn2_hamming / vis_good_matches on a dropped slot return VIS_E_STATE, never stale data).
    Plan* old = ctx->single;
    const bool keep = old && same_record_geometry(ctx->p, *p);
    ctx->p = *p;
    plan_destroy(ctx->batch); ctx->batch = nullptr;
    ctx->single = nullptr;
    // the cv::RNG sample table of the frame-at-a-time pose entry points, for every M they take from the table (<= VIS_POSE_TABLE_M): built
    // HERE, where parameters change, not inside the first vis_essential_ransac whose correspondence count exceeds the last one's (round 5's
    // single_frame_api leg: one call of 200 took 57 ms -- a table freed, rebuilt and re-uploaded when M grew from 33 to 39 -- against a
    // p50 of 0.14 ms).  A failure here is not fatal: the entry point asks again.
    (void)hipSetDevice(ctx->device);
    (void)vis_build_sample_table(ctx, VIS_POSE_TABLE_M);
    if (keep) {
        Plan* np = nullptr;
        rc = plan_create(ctx, old->w, old->h, old->stride, 1, VIS_NSLOTS, 1, &np);
        if (rc == VIS_OK && np->kcap == old->kcap && np->nrec == old->nrec) {
            std::swap(np->d_kps, old->d_kps); std::swap(np->d_desc, old->d_desc);
            std::swap(np->d_nkp, old->d_nkp); std::swap(np->d_descx, old->d_descx);
            ctx->single = np;
            plan_destroy(old);
            return VIS_OK;
        }
        plan_destroy(np);
        // the parameters are set, but the keyframe slots could not be carried over: say so (callers that rely on the slots see
        // VIS_E_STATE from the next matcher call; the reason is here)
        ctx->err = std::string("vis_set_params: keyframe slots dropped, re-planning failed (") + vis_strerror(rc) + ")" + (ctx->err.empty() ? "" : ": " + ctx->err);
    }
    plan_destroy(old);
    for (int i = 0; i < VIS_NSLOTS; i++) ctx->slot_valid[i] = 0;
    return VIS_OK;
}

extern "C" int vis_get_params(vis_ctx* ctx, vis_params* p) {
    if (!ctx || !p) return VIS_E_INVALID;
    *p = ctx->p;
    return VIS_OK;
}

extern "C" int vis_set_stream(vis_ctx* ctx, void* s) {
    if (!ctx) return VIS_E_INVALID;
    sync_all(ctx);
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    return VIS_OK;
}

extern "C" int vis_last_timings(vis_ctx* ctx, vis_timings* t) {
    if (!ctx || !t) return VIS_E_INVALID;
    *t = ctx->tm;
    return VIS_OK;
}

extern "C" int vis_level_geometry(vis_ctx* ctx, int w, int h, int32_t* widths, int32_t* heights,
                                  float* scales, int32_t* quotas) {
    if (!ctx) return VIS_E_INVALID;
    LevelInfo lv[VIS_MAX_LEVELS];
    int rc = vis_compute_levels(ctx->p, w, h, w, lv);
    if (rc) return rc;
    for (int l = 0; l < ctx->p.nlevels; l++) {
        if (widths) widths[l] = lv[l].w;
        if (heights) heights[l] = lv[l].h;
        if (scales) scales[l] = lv[l].scale;
        if (quotas) quotas[l] = lv[l].quota;
    }
    return VIS_OK;
}

// ------------------------------------------------------------------------------------------------ plans
template <class T> static int dalloc(vis_ctx* ctx, T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIPCHK(ctx, hipMalloc((void**)p, count * sizeof(T)));
    return VIS_OK;
}
#define DALLOC(ptr, count) do { int rc_ = dalloc(ctx, &(ptr), (count)); if (rc_) { plan_destroy(pl); return rc_; } } while (0)

void plan_destroy(Plan* pl) {
    if (!pl) return;
    auto F = [](void* p) { if (p) (void)hipFree(p); };
    F(pl->d_stage);
    for (int l = 0; l < VIS_MAX_LEVELS; l++) {
        F(pl->d_pyr[l]); F(pl->d_rs_tab[l]);
        F(pl->d_cand[l]); F(pl->d_seg_kp[l]);
    }
    F(pl->d_fast_tiles); F(pl->d_tile_cnt); F(pl->d_seg_cnt); F(pl->d_flags); F(pl->d_angle_tab); for (int i = 0; i < 2; i++) { F(pl->d_half_set[i]); F(pl->d_gx_set[i]); F(pl->d_gy_set[i]); F(pl->d_g_set[i]); } F(pl->d_tau); F(pl->d_seg_cut); F(pl->d_fix);
    F(pl->d_kps); F(pl->d_desc); F(pl->d_nkp); F(pl->d_descx);
    for (int i = 0; i < VIS_BATCH_SETS; i++) { F(pl->d_pq[i]); F(pl->d_pt[i]); F(pl->d_pqn[i]); }
    F(pl->d_knn12); F(pl->d_knn21);
    if (pl->mo_set[0][0] || pl->mo_set[1][0]) { for (int s_ = 0; s_ < 2; s_++) for (int k = 0; k < 6; k++) F(pl->mo_set[s_][k]); }    // (d_sym ... d_p2 alias one of the sets)
    else { F(pl->d_sym); F(pl->d_nsym); F(pl->d_good); F(pl->d_ngood); F(pl->d_p1); F(pl->d_p2); }                                  // a plan that failed before the sets were registered
    F(pl->d_hf); F(pl->d_wf);
    F(pl->d_n1); F(pl->d_n2); F(pl->d_mask); F(pl->d_samples); F(pl->d_models); F(pl->d_counts); F(pl->d_rstate); F(pl->d_pose); F(pl->d_worklist); F(pl->d_hyp);
    delete pl;
}

int plan_create(vis_ctx* ctx, int w, int h, int stride, int B, int nrec, int npairs, Plan** out, int nsets) {
    *out = nullptr;
    if (B < 1 || nrec < 1 || npairs < 1 || stride < w || (stride & 3)) return VIS_E_INVALID;
    Plan* pl = new (std::nothrow) Plan();
    if (!pl) return VIS_E_NOMEM;
    pl->w = w; pl->h = h; pl->stride = stride; pl->B = B; pl->L = ctx->p.nlevels; pl->npairs = npairs;
    pl->nsets = nsets; pl->rec_per_set = nrec; nrec *= nsets; pl->nrec = nrec;
    // k_fast segment height: a wave marches 8 fs_nch - 2 emitting rows.  Long segments amortise the prologue and the halo rows of a wave --
    // right for a batch, whose tens of thousands of waves fill the chip anyway; ONE frame has 68 such waves for 256 CUs and the kernel's
    // latency is a wave's lifetime, so the single-frame plan (and small batches) cut the same frame into more, shorter waves
    pl->fs_nch = B >= 32 ? VIS_FS_NCH : (B >= 8 ? 4 : 2);
    int rc = vis_compute_levels(ctx->p, w, h, stride, pl->lv, pl->fs_nch);
    if (rc) { delete pl; return rc; }
    const int L = pl->L;
    int kcap = 0; for (int l = 0; l < L; l++) kcap += pl->lv[l].quota + pl->lv[l].quota / 8 + 32;      // default: every level's own slack
    kcap = std::max(kcap, ctx->p.keypoint_capacity);             // (the levels' keep_cap share whatever the caller asked for beyond it)
    if (kcap > 65535) { delete pl; return VIS_E_INVALID; }       // packed 16-bit indices in the matcher
    pl->kcap = kcap;
    std::vector<float> hf, wf; vis_grid_limits(ctx->p, &pl->root, hf, wf);
    const int ncell = pl->root * pl->root;
    // correspondences per pair the pose stage sees: the grid-filtered good matches (reference pipeline) or every symmetric match
    const int mcap = ctx->p.pose_input == VIS_POSE_SYM ? kcap : ncell;
    if (mcap > VIS_RANSAC_MAX_M) { delete pl; return VIS_E_INVALID; }
    pl->pose_mcap = mcap;
    pl->max_iters = ctx->p.ransac_max_iters;
    DALLOC(pl->d_stage, (size_t)stride * h);
    for (int l = 1; l < L; l++) {
        const LevelInfo& V = pl->lv[l];
        DALLOC(pl->d_pyr[l], V.frame_bytes * B);
    }
    for (int l = 0; l < L; l++) {
        DALLOC(pl->d_cand[l], (size_t)pl->lv[l].cand_cap * B);
        DALLOC(pl->d_seg_kp[l], (size_t)pl->lv[l].keep_cap * B);
    }
    pl->total_tiles = pl->lv[L - 1].tile_base + pl->lv[L - 1].tiles_x * pl->lv[L - 1].tiles_y;
    DALLOC(pl->d_tile_cnt, (size_t)B * pl->total_tiles); DALLOC(pl->d_seg_cnt, (size_t)B * L + VIS_MAX_LEVELS);   // + padding: k_describe reads VIS_MAX_LEVELS counts per frame
    if (pl->total_tiles > 65535) { plan_destroy(pl); return VIS_E_INVALID; }        // k_fast: gridDim.y
    { int rc2 = build_fast_tiles(ctx, pl); if (rc2) { plan_destroy(pl); return rc2; } }
    if (nsets > 1) {                                               // batched stream plans predict the FAST threshold from batch to batch
        pl->speculate = true;
        DALLOC(pl->d_tau, L); DALLOC(pl->d_seg_cut, (size_t)B * L); DALLOC(pl->d_fix, (size_t)B * L + 1);
        std::vector<int32_t> t0((size_t)L, ctx->p.fast_threshold);
        HIPCHK(ctx, hipMemcpy(pl->d_tau, t0.data(), (size_t)L * 4, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemset(pl->d_fix, 0, ((size_t)B * L + 1) * 4));
    }
    DALLOC(pl->d_flags, 4);
    HIPCHK(ctx, hipMemset(pl->d_flags, 0, 16));
    DALLOC(pl->d_kps, (size_t)nrec * kcap); DALLOC(pl->d_desc, (size_t)nrec * kcap * 32); DALLOC(pl->d_nkp, nrec);
    HIPCHK(ctx, hipMemset(pl->d_nkp, 0, (size_t)nrec * 4));
    HIPCHK(ctx, hipMemset(pl->d_desc, 0, (size_t)nrec * kcap * 32));
    DALLOC(pl->d_descx, (size_t)nrec * kcap * 128);
    for (int sidx = 0; sidx < nsets; sidx++) {
        DALLOC(pl->d_pq[sidx], npairs); DALLOC(pl->d_pt[sidx], npairs); DALLOC(pl->d_pqn[sidx], npairs);
        const int base = sidx * pl->rec_per_set;
        std::vector<int32_t> q(npairs), t(npairs), qn(npairs);
        for (int i = 0; i < npairs; i++) { q[i] = base + i; t[i] = base + i + 1; qn[i] = i == 0 ? -1 : base + i; }
        HIPCHK(ctx, hipMemcpy(pl->d_pq[sidx], q.data(), (size_t)npairs * 4, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(pl->d_pt[sidx], t.data(), (size_t)npairs * 4, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(pl->d_pqn[sidx], qn.data(), (size_t)npairs * 4, hipMemcpyHostToDevice));
    }
    pl->d_pair_q = pl->d_pq[0]; pl->d_pair_t = pl->d_pt[0]; pl->d_pair_q_noprev = pl->d_pqn[0];
    DALLOC(pl->d_knn12, (size_t)npairs * kcap * 2); DALLOC(pl->d_knn21, (size_t)npairs * kcap * 2);
    DALLOC(pl->d_sym, (size_t)npairs * kcap); DALLOC(pl->d_nsym, npairs);
    DALLOC(pl->d_good, (size_t)npairs * ncell); DALLOC(pl->d_ngood, npairs);
    DALLOC(pl->d_p1, (size_t)npairs * mcap * 2); DALLOC(pl->d_p2, (size_t)npairs * mcap * 2);
    DALLOC(pl->d_hf, pl->root); DALLOC(pl->d_wf, pl->root);
    HIPCHK(ctx, hipMemcpy(pl->d_hf, hf.data(), (size_t)pl->root * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(pl->d_wf, wf.data(), (size_t)pl->root * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemset(pl->d_nsym, 0, (size_t)npairs * 4));
    HIPCHK(ctx, hipMemset(pl->d_ngood, 0, (size_t)npairs * 4));
    DALLOC(pl->d_n1, (size_t)npairs * mcap * 2); DALLOC(pl->d_n2, (size_t)npairs * mcap * 2);
    DALLOC(pl->d_mask, (size_t)npairs * mcap);
    DALLOC(pl->d_samples, (size_t)npairs * pl->max_iters * 5);
    DALLOC(pl->d_models, (size_t)npairs * pl->max_iters * 90);
    DALLOC(pl->d_counts, (size_t)npairs * pl->max_iters * 10);
    DALLOC(pl->d_rstate, (size_t)npairs * VIS_RSTATE_WORDS);
    DALLOC(pl->d_pose, npairs);
    DALLOC(pl->d_worklist, (size_t)npairs + 2);        // count, pairs, k_hyp_roots_packed item counter
    DALLOC(pl->d_hyp, (size_t)npairs * pl->max_iters * VIS_HYP_DOUBLES);
    // defence in depth: nothing should read these before writing them, but a recycled allocation must never
    // turn a missed guard into an out-of-bounds index (see the inactive-lane fix in k_ransac_hyp)
    HIPCHK(ctx, hipMemset(pl->d_samples, 0, (size_t)npairs * pl->max_iters * 5 * sizeof(int32_t)));
    HIPCHK(ctx, hipMemset(pl->d_counts, 0xFF, (size_t)npairs * pl->max_iters * 10 * sizeof(int32_t)));
    HIPCHK(ctx, hipMemset(pl->d_rstate, 0, (size_t)npairs * VIS_RSTATE_WORDS * sizeof(int32_t)));
    HIPCHK(ctx, hipMemset(pl->d_worklist, 0, ((size_t)npairs + 2) * sizeof(int32_t)));
    HIPCHK(ctx, hipMemset(pl->d_knn12, 0xFF, (size_t)npairs * kcap * 2 * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemset(pl->d_knn21, 0xFF, (size_t)npairs * kcap * 2 * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemset(pl->d_p1, 0, (size_t)npairs * mcap * 2 * sizeof(float)));
    HIPCHK(ctx, hipMemset(pl->d_p2, 0, (size_t)npairs * mcap * 2 * sizeof(float)));
    {   // the matcher-output sets (vis_internal.h): set 0 = the arrays above; a second one for plans that pipeline steps (nsets > 1)
        void* s0[6] = {pl->d_sym, pl->d_nsym, pl->d_good, pl->d_ngood, pl->d_p1, pl->d_p2};
        for (int k = 0; k < 6; k++) pl->mo_set[0][k] = s0[k];
#ifndef VIS_MO_SETS
#define VIS_MO_SETS 2
#endif
        if (pl->nsets > 1 && VIS_MO_SETS > 1) {
            vis_dmatch* sym2 = nullptr; int32_t* nsym2 = nullptr; vis_dmatch* good2 = nullptr; int32_t* ngood2 = nullptr; float* p12 = nullptr; float* p22 = nullptr;
            DALLOC(sym2, (size_t)npairs * kcap); pl->mo_set[1][0] = sym2;
            DALLOC(nsym2, npairs); pl->mo_set[1][1] = nsym2;
            DALLOC(good2, (size_t)npairs * ncell); pl->mo_set[1][2] = good2;
            DALLOC(ngood2, npairs); pl->mo_set[1][3] = ngood2;
            DALLOC(p12, (size_t)npairs * mcap * 2); pl->mo_set[1][4] = p12;
            DALLOC(p22, (size_t)npairs * mcap * 2); pl->mo_set[1][5] = p22;
            HIPCHK(ctx, hipMemset(nsym2, 0, (size_t)npairs * 4));
            HIPCHK(ctx, hipMemset(ngood2, 0, (size_t)npairs * 4));
            HIPCHK(ctx, hipMemset(p12, 0, (size_t)npairs * mcap * 2 * sizeof(float)));
            HIPCHK(ctx, hipMemset(p22, 0, (size_t)npairs * mcap * 2 * sizeof(float)));
        }
    }
    HIPCHK(ctx, hipMemset(pl->d_pose, 0, (size_t)npairs * sizeof(PoseOut)));
    // cv::RNG sample tables only for small M (the reference pipeline: M <= root^2); larger M replays the stream on the device
    { int rc2 = vis_build_sample_table(ctx, std::min(mcap, 8192)); if (rc2) { plan_destroy(pl); return rc2; } }
    *out = pl;
    return VIS_OK;
}

// cv::RNG((uint64)seed) + getSubset replayed on the host for every M in [6, max_m]: the RANSAC sample
// stream depends only on (seed, M), so the per-pair sequential RNG walk is replaced by a table row.
int vis_build_sample_table(vis_ctx* ctx, int max_m) {
    const int iters = ctx->p.ransac_max_iters;
    // as many rows as fit in 256 MB (2000 iterations: M <= 6715, i.e. config 3's ~3100 symmetric matches come from the table; pairs with
    // more correspondences than the table has rows replay the stream on the device, one thread per pair: 0.43 ms at M = 3100)
    max_m = (int)std::min<size_t>((size_t)max_m, 5 + ((size_t)256 << 20) / ((size_t)std::max(iters, 1) * 20));
    if (ctx->d_sample_table && ctx->sample_max_m >= max_m && ctx->sample_iters == iters && ctx->sample_seed == ctx->p.ransac_seed) return VIS_OK;
    if (ctx->d_sample_table) { sync_all(ctx); (void)hipFree(ctx->d_sample_table); ctx->d_sample_table = nullptr; ctx->sample_max_m = 0; }
    if (max_m < 6) return VIS_OK;                                  // no table: device replay
    std::vector<int32_t> tab;                                      // up to 256 MB of host memory: an allocation failure is an error code, not an exception across the C ABI
    try { tab.resize((size_t)(max_m - 5) * iters * 5); } catch (...) { ctx->err = "sample table: host allocation failed"; return VIS_E_NOMEM; }
    for (int M = 6; M <= max_m; M++) {
        unsigned long long state = ctx->p.ransac_seed ? ctx->p.ransac_seed : 0xffffffffULL;
        int32_t* dst = tab.data() + (size_t)(M - 6) * iters * 5;
        for (int it = 0; it < iters; it++) {
            int idx[5];
            for (int i = 0; i < 5; i++) {
                for (;;) {
                    state = (unsigned long long)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32);   // RNG::next
                    const int v = idx[i] = (int)((unsigned)state % (unsigned)M);                               // uniform(0, M)
                    int j = 0;
                    for (; j < i; j++) if (v == idx[j]) break;
                    if (j == i) break;
                }
            }
            for (int k = 0; k < 5; k++) dst[5 * it + k] = idx[k];
        }
    }
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_sample_table, tab.size() * 4));
    HIPCHK(ctx, hipMemcpy(ctx->d_sample_table, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    ctx->sample_max_m = max_m; ctx->sample_iters = iters; ctx->sample_seed = ctx->p.ransac_seed;
    return VIS_OK;
}

int launch_pose(vis_ctx* ctx, Plan* pl, int npairs) {
    if (npairs <= 0) return VIS_OK;
    return pose_run(ctx, npairs, pl->pose_mcap, pl->max_iters, pl->d_p1, pl->d_p2,
                    ctx->p.pose_input == VIS_POSE_SYM ? pl->d_nsym : pl->d_ngood, pl->d_n1, pl->d_n2,
                    pl->d_samples, pl->d_models, pl->d_counts, pl->d_rstate, nullptr, pl->d_mask, pl->d_pose, 1, 1, pl->d_worklist, pl->d_hyp);
}

// grow-only scratch of the host-pointer entry points.  Every stream of the context is drained before the old block is
// freed: batch work queued on the matcher / pose streams may still read buffers carved from it.
int vis_ensure_scratch(vis_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return VIS_OK;
    sync_all(ctx);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    ctx->d_scratch = nullptr; ctx->scratch_bytes = 0;
    HIPCHK(ctx, hipMalloc(&ctx->d_scratch, bytes));
    ctx->scratch_bytes = bytes;
    return VIS_OK;
}
static inline int ensure_scratch(vis_ctx* ctx, size_t bytes) { return vis_ensure_scratch(ctx, bytes); }

std::atomic<unsigned long long> vis_g_launches{0};

// ---- host staging of the single-frame entry points --------------------------------------------------------------------------------
// The reference's main hands over pageable host memory (cv::Mat, std::vector) once per camera frame (src/main_vi_slamGPU.cpp:118-123).  A
// blocking hipMemcpy on such memory is a full host <-> device round trip each; round 4's vis_orb_detect_compute made four of them, the
// matcher entry six.  Here every transfer of a call goes through the context's pinned block as an ASYNCHRONOUS copy on the context's
// stream -- uploads are copied into the block first, downloads are copied out of it after the call's ONE wait -- so an entry point blocks
// once (twice when a result decides what else to fetch).
int vis_ensure_pin(vis_ctx* ctx, size_t bytes) {
    if (ctx->h_pin_bytes >= bytes) return VIS_OK;
    if (ctx->stage_live > 0) { ctx->err = "staging block would be replaced under a live HostStage (internal: size the block before building the stage)"; return VIS_E_STATE; }
    (void)hipStreamSynchronize(ctx->stream);                       // nothing queued may still use the old block
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    ctx->h_pin = nullptr; ctx->h_pin_bytes = 0; ctx->h_pin_dev = nullptr;
    const size_t want = std::max(bytes + bytes / 4, (size_t)1 << 20);
    if (hipHostMalloc(&ctx->h_pin, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->err = "hipHostMalloc of the staging block failed"; return VIS_E_NOMEM; }
    ctx->h_pin_bytes = want;
    if (hipHostGetDevicePointer(&ctx->h_pin_dev, ctx->h_pin, 0) != hipSuccess) { (void)hipGetLastError(); ctx->h_pin_dev = nullptr; }   // (downloads then go through the runtime's copies)
    return VIS_OK;
}
// diagnostics: out[0] = kernel launches of this process, out[1] = times a single-frame entry point of this context blocked on the device,
// out[2] = asynchronous copies it queued (bench.py `single_frame_api`: launches and round trips per frame)
extern "C" int vis_debug_counters(vis_ctx* ctx, unsigned long long out[4]) {
    if (!ctx || !out) return VIS_E_INVALID;
    out[0] = vis_g_launches.load(std::memory_order_relaxed); out[1] = ctx->n_host_waits; out[2] = ctx->n_copies; out[3] = 0;
    return VIS_OK;
}

__global__ void k_set_pair(int32_t* q, int32_t* t, int32_t vq, int32_t vt) { *q = vq; *t = vt; }

static int ensure_single(vis_ctx* ctx, int w, int h) {
    (void)hipSetDevice(ctx->device);
    if (ctx->single && ctx->single->w == w && ctx->single->h == h) return VIS_OK;
    (void)hipStreamSynchronize(ctx->stream);
    plan_destroy(ctx->single); ctx->single = nullptr;
    for (int i = 0; i < VIS_NSLOTS; i++) ctx->slot_valid[i] = 0;
    const int stride = ((w + 63) / 64) * 64;
    return plan_create(ctx, w, h, stride, 1, VIS_NSLOTS, 1, &ctx->single);
}

// re-create the single-frame plan with a larger per-frame keypoint capacity; the valid slots keep their records
static int grow_single(vis_ctx* ctx, int capacity) {
    Plan* old = ctx->single;
    if (!old) return VIS_E_STATE;
    sync_all(ctx);
    const int cap_before = ctx->p.keypoint_capacity;
    ctx->p.keypoint_capacity = capacity;
    Plan* np = nullptr;
    int rc = plan_create(ctx, old->w, old->h, old->stride, 1, VIS_NSLOTS, 1, &np);
    if (rc == VIS_OK && (np->kcap < old->kcap || np->nrec != old->nrec)) { plan_destroy(np); rc = VIS_E_STATE; }
    if (rc) { ctx->p.keypoint_capacity = cap_before; return rc; }       // (the old plan and its slots stay as they were)
    // every failure from here on leaves the context as it was: the new plan is destroyed, the capacity restored (the early
    // returns of HIPCHK leaked `np` and left ctx->p ahead of the live plan)
    auto copy_records = [&]() -> hipError_t {
        std::vector<int32_t> nk((size_t)old->nrec);
        hipError_t e = hipMemcpy(nk.data(), old->d_nkp, nk.size() * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(np->d_nkp, nk.data(), nk.size() * 4, hipMemcpyHostToDevice);
        for (int r = 0; e == hipSuccess && r < old->nrec && r < VIS_NSLOTS; r++) {
            const size_t n = (size_t)std::min(std::max(nk[r], 0), old->kcap);
            if (!ctx->slot_valid[r] || !n) continue;
            e = hipMemcpy(np->d_kps + (size_t)r * np->kcap, old->d_kps + (size_t)r * old->kcap, n * sizeof(vis_keypoint), hipMemcpyDeviceToDevice);
            if (e == hipSuccess) e = hipMemcpy(np->d_desc + (size_t)r * np->kcap * 32, old->d_desc + (size_t)r * old->kcap * 32, n * 32, hipMemcpyDeviceToDevice);
            if (e == hipSuccess) e = hipMemcpy(np->d_descx + (size_t)r * np->kcap * 128, old->d_descx + (size_t)r * old->kcap * 128, n * 128, hipMemcpyDeviceToDevice);
        }
        return e;
    };
    const hipError_t ce = copy_records();
    if (ce != hipSuccess) {
        ctx->err = std::string("grow_single: ") + hipGetErrorString(ce);
        plan_destroy(np); ctx->p.keypoint_capacity = cap_before;
        return VIS_E_HIP;
    }
    plan_destroy(old);
    ctx->single = np;
    // a batch plan was sized with the old parameters: it goes (sync_all above has drained its streams; results of a finished batch that
    // the caller has not fetched yet are lost with it -- vis_set_params documents the same for every parameter change)
    plan_destroy(ctx->batch); ctx->batch = nullptr;
    return VIS_OK;
}

static void key_to_dmatch(uint32_t key, int q, vis_dmatch* m) {
    m->queryIdx = q;
    if (key == 0xFFFFFFFFu) { m->trainIdx = -1; m->imgIdx = -1; m->distance = FLT_MAX; }
    else { m->trainIdx = (int)(key & 0xFFFF); m->imgIdx = 0; m->distance = (float)(key >> 16); }
}

static int check_flags(vis_ctx* ctx, Plan* pl) {
    int32_t fl = 0;
    HIPCHK(ctx, hipMemcpy(&fl, pl->d_flags, 4, hipMemcpyDeviceToHost));
    if (fl) {
        ctx->err = "device capacity flag set: " + std::to_string(fl) + ((fl & 12) ? " (more tied keypoints than vis_params.keypoint_capacity holds: " +
                   std::to_string(pl->kcap) + " per frame; raise it)" : "");
        HIPCHK(ctx, hipMemset(pl->d_flags, 0, 4)); return VIS_E_CAPACITY;
    }
    return VIS_OK;
}

// ------------------------------------------------------------------------------------------------ single-frame API
extern "C" int vis_camera_update(vis_ctx* ctx, const uint8_t* img, int w, int h, int stride, uint8_t* const out_levels[5]) {
    if (!ctx || !img || !out_levels || w < 16 || h < 16 || stride < w) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    int lw[5], lh[5]; vis_half_dims(w, h, lw, lh);
    size_t lvl_bytes[5]; size_t total = (size_t)w * h;
    lvl_bytes[0] = (size_t)w * h;
    for (int l = 1; l < 5; l++) { lvl_bytes[l] = (size_t)lw[l] * lh[l]; total += lvl_bytes[l] + 256; }
    int rc = ensure_scratch(ctx, total + 1024);
    if (!rc) rc = vis_ensure_pin(ctx, 2 * total + 4096);
    if (rc) return rc;
    Carver cv{(char*)ctx->d_scratch, 0};
    uint8_t* d[5];
    for (int l = 0; l < 5; l++) d[l] = cv.take<uint8_t>(lvl_bytes[l]);
    HostStage hs(ctx);
    hs.up2d(d[0], w, img, stride, w, h);
    hs.flush_ups();
    rc = launch_half_pyramid(ctx, d[0], w, h, w, d);
    if (rc) return rc;
    const void* got[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int l = 0; l < 5; l++) if (out_levels[l]) got[l] = hs.down(d[l], lvl_bytes[l]);
    rc = hs.wait();
    if (rc) return rc;
    for (int l = 0; l < 5; l++) if (out_levels[l]) std::memcpy(out_levels[l], got[l], lvl_bytes[l]);
    return VIS_OK;
}

extern "C" size_t vis_gradient_frame_elems(int w, int h) { return (w < 16 || h < 16) ? 0 : vis_grad_frame_elems(w, h); }
extern "C" void vis_half_pyramid_dims(int w, int h, int32_t lw[5], int32_t lh[5]) { if (lw && lh) vis_half_dims(w, h, lw, lh); }

extern "C" int vis_gradient_batch(vis_ctx* ctx, const uint8_t* d_frames, int w, int h, int stride, int n, int scale,
                                  uint8_t* d_gray, int16_t* d_gx, int16_t* d_gy, uint8_t* d_g) {
    if (!ctx || !d_frames || !d_gray || !d_gx || !d_gy || !d_g) return VIS_E_INVALID;
    if (w < 16 || h < 16 || stride < w || (stride & 3) || n < 1 || scale < 1 || scale > 8) {
        ctx->err = "vis_gradient_batch: w, h >= 16, stride >= w and % 4 == 0, n >= 1, 1 <= scale <= 8"; return VIS_E_INVALID;
    }
    if (((uintptr_t)d_gx | (uintptr_t)d_gy | (uintptr_t)d_g | (uintptr_t)d_gray) & 15) { ctx->err = "vis_gradient_batch: output buffers must be 16-byte aligned"; return VIS_E_INVALID; }
    (void)hipSetDevice(ctx->device);
    // a vis_batch_align still in flight on the pose stream reads the previous gradients (usually these very buffers)
    if (ctx->align_pending) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_align_done, 0));
    const size_t frame_bytes = (size_t)stride * h;
    int rc = launch_half_pyramid_batch(ctx, d_frames, w, h, stride, frame_bytes, n, d_gray);
    if (rc) return rc;
    return launch_gradient(ctx, d_frames, w, h, stride, frame_bytes, n, d_gray, scale, d_gx, d_gy, d_g);
}

extern "C" int vis_compute_gradient(vis_ctx* ctx, const uint8_t* img, int w, int h, int stride, int scale,
                                    int16_t* const gx[5], int16_t* const gy[5], uint8_t* const g[5]) {
    if (!ctx || !img || !gx || !gy || !g || w < 16 || h < 16 || stride < w || scale < 1 || scale > 8) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    const size_t fe = vis_grad_frame_elems(w, h);
    const int ws = (w + 15) & ~15;                              // device row stride of the copy (the batched entry wants % 4 == 0)
    int rc = ensure_scratch(ctx, (size_t)ws * h + fe * 6 + 4096);
    if (rc) return rc;
    Carver cv{(char*)ctx->d_scratch, 0};
    uint8_t* d_img = cv.take<uint8_t>((size_t)ws * h);
    uint8_t* d_gray = cv.take<uint8_t>(fe);
    int16_t* d_gx = cv.take<int16_t>(fe); int16_t* d_gy = cv.take<int16_t>(fe);
    uint8_t* d_g = cv.take<uint8_t>(fe);
    rc = vis_ensure_pin(ctx, (size_t)w * h + fe * 5 + 4096);
    if (rc) return rc;
    HostStage hs(ctx);
    hs.up2d(d_img, ws, img, stride, w, h);
    hs.flush_ups();
    rc = vis_gradient_batch(ctx, d_img, w, h, ws, 1, scale, d_gray, d_gx, d_gy, d_g);
    if (rc) return rc;
    int lw[5], lh[5]; vis_half_dims(w, h, lw, lh);
    // the three outputs are dense over the five levels: one copy each, cut into levels on the host
    const int16_t* h_gx = (const int16_t*)hs.down(d_gx, fe * 2);
    const int16_t* h_gy = (const int16_t*)hs.down(d_gy, fe * 2);
    const uint8_t* h_g = (const uint8_t*)hs.down(d_g, fe);
    rc = hs.wait();
    if (rc) return rc;
    size_t off = 0;
    for (int l = 0; l < 5; l++) {
        const size_t cnt = (size_t)lw[l] * lh[l];
        if (gx[l]) std::memcpy(gx[l], h_gx + off, cnt * 2);
        if (gy[l]) std::memcpy(gy[l], h_gy + off, cnt * 2);
        if (g[l]) std::memcpy(g[l], h_g + off, cnt);
        off += cnt;
    }
    return VIS_OK;
}

extern "C" int vis_patch_points(vis_ctx* ctx, const vis_keypoint* good, int n, int cap,
                                float* const patch[5], int n_patch[5], float* const debug[5], int n_debug[5]) {
    if (!ctx || (n > 0 && !good) || n < 0 || cap < 0 || !patch || !debug || !n_patch || !n_debug) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    const int m = std::min(n, 200);
    int rc = ensure_scratch(ctx, (size_t)200 * sizeof(vis_keypoint) + (size_t)2 * 5 * std::max(cap, 1) * 16 + 4096);
    if (rc) return rc;
    Carver cv{(char*)ctx->d_scratch, 0};
    vis_keypoint* d_good = cv.take<vis_keypoint>(200);
    float* d_patch = cv.take<float>((size_t)5 * std::max(cap, 1) * 4);
    float* d_debug = cv.take<float>((size_t)5 * std::max(cap, 1) * 4);
    int32_t* d_cnt = cv.take<int32_t>(10);
    rc = vis_ensure_pin(ctx, (size_t)200 * sizeof(vis_keypoint) + (size_t)10 * std::max(cap, 1) * 16 + 4096);
    if (rc) return rc;
    HostStage hs(ctx);
    hs.up(d_good, good, (size_t)m * sizeof(vis_keypoint));
    hs.flush_ups();
    rc = launch_patch_points(ctx, d_good, m, ctx->p.w_size, ctx->p.h_size, d_patch, d_debug, cap, d_cnt);
    if (rc) return rc;
    // counts and both point lists of all five levels behind the kernels.  The list lengths are not known before the wait, but their
    // bounds are: min(m, 200) keypoints x at most (2 sp + 2)^2 points of a level's window (sp = 5, 3, 2, 5, 5: k_patch_points) and one
    // debug point per keypoint -- that much of every level is fetched, not the caller's whole capacity (200 x 121 x 16 B x 5 levels x 2
    // lists = 3.9 MB per call for 49 good matches; the GPU main calls this twice per frame)
    const int32_t* cnt = (const int32_t*)hs.down(d_cnt, 40);
    const float* h_patch[5]; const float* h_debug[5]; int bp[5], bd[5];
    const int mk = std::min(m, 200);
    for (int l = 0; l < 5; l++) {
        const int sp = l == 1 ? 3 : (l == 2 ? 2 : 5);
        bp[l] = (int)std::min<long long>(cap, (long long)mk * (2 * sp + 2) * (2 * sp + 2)); bd[l] = std::min(cap, mk);
        h_patch[l] = bp[l] ? (const float*)hs.down(d_patch + (size_t)l * cap * 4, (size_t)bp[l] * 16) : nullptr;
        h_debug[l] = bd[l] ? (const float*)hs.down(d_debug + (size_t)l * cap * 4, (size_t)bd[l] * 16) : nullptr;
    }
    rc = hs.wait();
    if (rc) return rc;
    bool over = false;
    for (int l = 0; l < 5; l++) {
        n_patch[l] = cnt[l]; n_debug[l] = cnt[5 + l];
        over = over || cnt[l] > cap || cnt[5 + l] > cap;
        const int np = std::min(cnt[l], bp[l]), nd = std::min(cnt[5 + l], bd[l]);         // (the bounds hold by construction; never read past what was fetched)
        if (patch[l] && np) std::memcpy(patch[l], h_patch[l], (size_t)np * 16);
        if (debug[l] && nd) std::memcpy(debug[l], h_debug[l], (size_t)nd * 16);
    }
    return over ? VIS_E_CAPACITY : VIS_OK;
}

// elapsed time between two events, false when either was never recorded (a stage that did not run, a sync before the first run).
// The failed query must not stay behind as the thread's "last error": the launch checks (hipGetLastError after a kernel launch)
// would report it for the next, valid, call.
static bool ev_elapsed(float* ms, hipEvent_t a, hipEvent_t b) {
    if (hipEventElapsedTime(ms, a, b) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

static void collect_detect_timings(vis_ctx* ctx) {
    if (!ctx->ev_ok) return;
    float a = 0;
    if (ev_elapsed(&a, ctx->ev[0], ctx->ev[1])) ctx->tm.ms_pyramid = a;
    if (ev_elapsed(&a, ctx->ev[1], ctx->ev[2])) ctx->tm.ms_fast = a;
    if (ev_elapsed(&a, ctx->ev[2], ctx->ev[3])) ctx->tm.ms_select = a;
    if (ev_elapsed(&a, ctx->ev[3], ctx->ev[4])) ctx->tm.ms_describe = a;
}

extern "C" int vis_orb_detect_compute(vis_ctx* ctx, const uint8_t* img, int w, int h, int stride, int frame_slot,
                                      vis_keypoint* kps_out, uint8_t* desc_out, int cap, int* n_out) {
    if (!ctx || !img || !n_out || frame_slot < 0 || frame_slot >= VIS_NSLOTS || stride < w) return VIS_E_INVALID;
    int rc = ensure_single(ctx, w, h);
    if (rc) return rc;
    Plan* pl = ctx->single;
    // everything the call may hand back, fetched behind the detect chain in one go: count, device flags and as many keypoint /
    // descriptor records as the caller can take (the count is not known before the wait; a record set is a few tens of KB)
    const size_t nfetch = (size_t)std::max(0, std::min(cap, pl->kcap));
    rc = vis_ensure_pin(ctx, (size_t)w * h + nfetch * (sizeof(vis_keypoint) + 32) + 4096);
    if (rc) return rc;
    std::memset(&ctx->tm, 0, sizeof(ctx->tm));
    int grow_to = 0;                                               // > 0: more ties than the plan's records hold; grow and detect again (below, once the stage is gone)
    {
        HostStage hs(ctx);
        hs.up2d(pl->d_stage, pl->stride, img, stride, w, h);
        hs.flush_ups();
        if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[0], ctx->stream);
        rc = launch_detect(ctx, pl, pl->d_stage, 1, frame_slot);
        if (rc) return rc;
        const int32_t* h_n = (const int32_t*)hs.down(pl->d_nkp + frame_slot, 4);
        const int32_t* h_fl = (const int32_t*)hs.down(pl->d_flags, 4);
        const void* h_kps = kps_out && nfetch ? hs.down(pl->d_kps + (size_t)frame_slot * pl->kcap, nfetch * sizeof(vis_keypoint)) : nullptr;
        const void* h_desc = desc_out && nfetch ? hs.down(pl->d_desc + (size_t)frame_slot * pl->kcap * 32, nfetch * 32) : nullptr;
        rc = hs.wait();
        if (rc) return rc;
        collect_detect_timings(ctx);
        { float a = 0; if (ctx->ev_ok && ev_elapsed(&a, ctx->ev[0], ctx->ev[4])) ctx->tm.ms_total = a; }
        const int32_t n = *h_n, fl = *h_fl;
        if (fl) {
            // More tied keypoints than the plan's records hold (KeyPointsFilter::retainBest keeps every tie at its cut: a checkerboard)?  The
            // caller's `cap` says how many it is prepared to take: grow the per-frame capacity towards it -- the other slots' records move
            // into the re-created plan -- and detect again.  Only what exceeds `cap` (or 65535) is an error.
            if ((fl & 12) && !(fl & ~12) && cap > pl->kcap && pl->kcap < 65535) grow_to = std::min(65535, std::max(cap, 2 * pl->kcap));
            else {
                rc = check_flags(ctx, pl);                         // (reads and clears the flags: the error path may block again)
                if (rc) return rc;
            }
        }
        if (!grow_to) {
            ctx->slot_valid[frame_slot] = 1;
            *n_out = n;
            if (n > cap && (kps_out || desc_out)) return VIS_E_CAPACITY;
            if (kps_out && n) std::memcpy(kps_out, h_kps, (size_t)n * sizeof(vis_keypoint));
            if (desc_out && n) std::memcpy(desc_out, h_desc, (size_t)n * 32);
            return VIS_OK;
        }
    }
    // The retry runs with NO stage alive: it sizes the staging block for the larger record count, which may replace the block (round 6: the
    // retry used to re-enter this function under the first pass's live stage -- harmless only because that stage was never touched again;
    // vis_ensure_pin now refuses it, tests/test_edge_cases_gpu.py::test_ties_beyond_the_default_slack_are_all_returned on a fresh context)
    HIPCHK(ctx, hipMemset(pl->d_flags, 0, 4));
    rc = grow_single(ctx, grow_to);
    if (rc) return rc;
    return vis_orb_detect_compute(ctx, img, w, h, stride, frame_slot, kps_out, desc_out, cap, n_out);
}

static int set_single_pair(vis_ctx* ctx, Plan* pl, int slot_q, int slot_t) {
    if (!pl) return VIS_E_STATE;
    if (slot_q < 0 || slot_q >= VIS_NSLOTS || slot_t < 0 || slot_t >= VIS_NSLOTS) return VIS_E_INVALID;
    if (!ctx->slot_valid[slot_q] || !ctx->slot_valid[slot_t]) return VIS_E_STATE;
    hipLaunchKernelGGL(k_set_pair, dim3(1), dim3(1), 0, ctx->stream, pl->d_pair_q, pl->d_pair_t, (int32_t)slot_q, (int32_t)slot_t);   // (two blocking 4-byte uploads before)
    HIPCHK(ctx, hipGetLastError());
    return VIS_OK;
}

static void keys_to_dmatches(const uint32_t* k, int n, vis_dmatch* out) {
    if (!out) return;
    for (int q = 0; q < n; q++) { key_to_dmatch(k[2 * q], q, out + 2 * q); key_to_dmatch(k[2 * q + 1], q, out + 2 * q + 1); }
}

extern "C" int vis_bf_knn2_hamming(vis_ctx* ctx, int slot_q, int slot_t, vis_dmatch* out12, vis_dmatch* out21) {
    if (!ctx) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    Plan* pl = ctx->single;
    int rc = pl ? vis_ensure_pin(ctx, (size_t)pl->kcap * 16 + 4096) : VIS_E_STATE;
    if (!rc) rc = set_single_pair(ctx, pl, slot_q, slot_t);
    if (rc) return rc;
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[4], ctx->stream);
    rc = launch_expand(ctx, pl, slot_q, 1);
    if (!rc && slot_t != slot_q) rc = launch_expand(ctx, pl, slot_t, 1);
    if (!rc) rc = launch_match(ctx, pl, 1);
    if (rc) return rc;
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[5], ctx->stream);
    HostStage hs(ctx);
    const int32_t* h_nq = (const int32_t*)hs.down(pl->d_nkp + slot_q, 4);
    const int32_t* h_nt = (const int32_t*)hs.down(pl->d_nkp + slot_t, 4);
    const uint32_t* h12 = out12 ? (const uint32_t*)hs.down(pl->d_knn12, (size_t)pl->kcap * 8) : nullptr;
    const uint32_t* h21 = out21 ? (const uint32_t*)hs.down(pl->d_knn21, (size_t)pl->kcap * 8) : nullptr;
    rc = hs.wait();
    if (rc) return rc;
    { float a = 0; if (ctx->ev_ok && ev_elapsed(&a, ctx->ev[4], ctx->ev[5])) ctx->tm.ms_knn = a; }
    keys_to_dmatches(h12, std::min(*h_nq, pl->kcap), out12);
    keys_to_dmatches(h21, std::min(*h_nt, pl->kcap), out21);
    return VIS_OK;
}

extern "C" int vis_bf_knn2_hamming_host(vis_ctx* ctx, const uint8_t* desc_q, int n_q, const uint8_t* desc_t, int n_t,
                                        vis_dmatch* out12, vis_dmatch* out21) {
    if (!ctx || n_q < 0 || n_t < 0 || (n_q && !desc_q) || (n_t && !desc_t)) return VIS_E_INVALID;
    if (n_q > 65535 || n_t > 65535) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    const int kcap = std::max(std::max(n_q, n_t), 1);
    int rc = ensure_scratch(ctx, (size_t)kcap * 32 * 2 + (size_t)kcap * 8 * 2 + (size_t)kcap * 256 + 8192);
    if (!rc) rc = vis_ensure_pin(ctx, (size_t)kcap * (64 + 16) + 4096);
    if (rc) return rc;
    Carver cv{(char*)ctx->d_scratch, 0};
    Plan tp;
    tp.kcap = kcap;
    tp.d_desc = cv.take<uint8_t>((size_t)kcap * 64);
    tp.d_nkp = cv.take<int32_t>(2); tp.d_pair_q = cv.take<int32_t>(1); tp.d_pair_t = cv.take<int32_t>(1);
    tp.d_knn12 = cv.take<uint32_t>((size_t)kcap * 2); tp.d_knn21 = cv.take<uint32_t>((size_t)kcap * 2);
    tp.d_descx = cv.take<int8_t>((size_t)kcap * 256);
    const int32_t nk[2] = {n_q, n_t};
    HostStage hs(ctx);
    hs.up(tp.d_desc, desc_q, (size_t)n_q * 32);
    hs.up(tp.d_desc + (size_t)kcap * 32, desc_t, (size_t)n_t * 32);
    hs.up(tp.d_nkp, nk, 8);
    hs.flush_ups();
    hipLaunchKernelGGL(k_set_pair, dim3(1), dim3(1), 0, ctx->stream, tp.d_pair_q, tp.d_pair_t, 0, 1);
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[4], ctx->stream);
    rc = launch_expand(ctx, &tp, 0, 2);
    if (!rc) rc = launch_match(ctx, &tp, 1);
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[5], ctx->stream);
    const uint32_t* h12 = out12 && n_q ? (const uint32_t*)hs.down(tp.d_knn12, (size_t)n_q * 8) : nullptr;
    const uint32_t* h21 = out21 && n_t ? (const uint32_t*)hs.down(tp.d_knn21, (size_t)n_t * 8) : nullptr;
    tp = Plan();       // scratch-owned pointers: nothing to free
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    rc = hs.wait();
    if (rc) return rc;
    { float a = 0; if (ctx->ev_ok && ev_elapsed(&a, ctx->ev[4], ctx->ev[5])) ctx->tm.ms_knn = a; }
    if (h12) keys_to_dmatches(h12, n_q, out12);
    if (h21) keys_to_dmatches(h21, n_t, out21);
    return VIS_OK;
}

// the filter results of one pair: queued behind the filter kernel, copied out after the call's wait
struct MatchFetch { const int32_t* ng; const int32_t* ns; const void* good; const void* sym; size_t ngood_cap, nsym_cap; };
static MatchFetch fetch_matches(HostStage& hs, Plan* pl, int pair, bool want_good, int cap, bool want_sym, int sym_cap) {
    MatchFetch f = {};
    f.ng = (const int32_t*)hs.down(pl->d_ngood + pair, 4);
    f.ns = (const int32_t*)hs.down(pl->d_nsym + pair, 4);
    f.ngood_cap = (size_t)std::max(0, std::min(cap, pl->root * pl->root));
    f.nsym_cap = (size_t)std::max(0, std::min(sym_cap, pl->kcap));
    if (want_good && f.ngood_cap) f.good = hs.down(pl->d_good + (size_t)pair * pl->root * pl->root, f.ngood_cap * sizeof(vis_dmatch));
    if (want_sym && f.nsym_cap) f.sym = hs.down(pl->d_sym + (size_t)pair * pl->kcap, f.nsym_cap * sizeof(vis_dmatch));
    return f;
}
static int deliver_matches(const MatchFetch& f, vis_dmatch* good, int cap, int* n_good, vis_dmatch* sym_out, int sym_cap, int* n_sym) {
    const int ng = *f.ng, ns = *f.ns;
    if (n_good) *n_good = ng;
    if (n_sym) *n_sym = ns;
    if (good) {
        if (ng > cap) return VIS_E_CAPACITY;
        if (ng) std::memcpy(good, f.good, (size_t)ng * sizeof(vis_dmatch));
    }
    if (sym_out) {
        if (ns > sym_cap) return VIS_E_CAPACITY;
        if (ns) std::memcpy(sym_out, f.sym, (size_t)ns * sizeof(vis_dmatch));
    }
    return VIS_OK;
}

extern "C" int vis_good_matches(vis_ctx* ctx, int slot_prev, int slot_cur, vis_dmatch* good, int cap, int* n_good,
                                vis_dmatch* sym_out, int sym_cap, int* n_sym) {
    if (!ctx) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    Plan* pl = ctx->single;
    int rc = pl ? vis_ensure_pin(ctx, ((size_t)pl->kcap + (size_t)pl->root * pl->root) * sizeof(vis_dmatch) + 4096) : VIS_E_STATE;
    if (!rc) rc = set_single_pair(ctx, pl, slot_prev, slot_cur);
    if (rc) return rc;
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[4], ctx->stream);
    rc = launch_expand(ctx, pl, slot_prev, 1);
    if (!rc && slot_cur != slot_prev) rc = launch_expand(ctx, pl, slot_cur, 1);
    if (!rc) rc = launch_match(ctx, pl, 1);
    if (rc) return rc;
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[5], ctx->stream);
    rc = launch_filter(ctx, pl, 1);
    if (rc) return rc;
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[6], ctx->stream);
    HostStage hs(ctx);
    const MatchFetch f = fetch_matches(hs, pl, 0, good != nullptr, cap, sym_out != nullptr, sym_cap);
    rc = hs.wait();
    if (rc) return rc;
    { float a = 0;
      if (ctx->ev_ok && ev_elapsed(&a, ctx->ev[4], ctx->ev[5])) ctx->tm.ms_knn = a;
      if (ctx->ev_ok && ev_elapsed(&a, ctx->ev[5], ctx->ev[6])) ctx->tm.ms_filter = a; }
    return deliver_matches(f, good, cap, n_good, sym_out, sym_cap, n_sym);
}

static uint32_t dmatch_to_key(const vis_dmatch& m) {
    if (m.trainIdx < 0) return 0xFFFFFFFFu;
    return ((uint32_t)m.distance << 16) | (uint32_t)(m.trainIdx & 0xFFFF);
}

extern "C" int vis_good_matches_host(vis_ctx* ctx, const vis_keypoint* kps1, int n1, const vis_keypoint* kps2, int n2,
                                     const vis_dmatch* knn12, const vis_dmatch* knn21,
                                     vis_dmatch* good, int cap, int* n_good,
                                     vis_dmatch* sym_out, int sym_cap, int* n_sym) {
    if (!ctx || n1 < 0 || n2 < 0 || n1 > 65535 || n2 > 65535) return VIS_E_INVALID;
    if ((n1 && (!kps1 || !knn12)) || (n2 && (!kps2 || !knn21))) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    const int kcap = std::max(std::max(n1, n2), 1);
    int root; std::vector<float> hf, wf; vis_grid_limits(ctx->p, &root, hf, wf);
    const int ncell = root * root;
    size_t need = (size_t)kcap * (2 * sizeof(vis_keypoint) + 16 + sizeof(vis_dmatch)) + (size_t)ncell * (sizeof(vis_dmatch) + 16) + 16384;
    int rc = ensure_scratch(ctx, need);
    if (rc) return rc;
    Carver cv{(char*)ctx->d_scratch, 0};
    Plan tp; tp.kcap = kcap; tp.root = root;
    tp.d_kps = cv.take<vis_keypoint>((size_t)kcap * 2);
    tp.d_nkp = cv.take<int32_t>(2); tp.d_pair_q = cv.take<int32_t>(1); tp.d_pair_t = cv.take<int32_t>(1);
    tp.d_knn12 = cv.take<uint32_t>((size_t)kcap * 2); tp.d_knn21 = cv.take<uint32_t>((size_t)kcap * 2);
    tp.d_sym = cv.take<vis_dmatch>(kcap); tp.d_nsym = cv.take<int32_t>(1);
    tp.d_good = cv.take<vis_dmatch>(ncell); tp.d_ngood = cv.take<int32_t>(1);
    tp.d_p1 = cv.take<float>((size_t)ncell * 2); tp.d_p2 = cv.take<float>((size_t)ncell * 2);
    tp.d_hf = cv.take<float>(root); tp.d_wf = cv.take<float>(root);
    std::vector<uint32_t> k12(2 * (size_t)kcap, 0xFFFFFFFFu), k21(2 * (size_t)kcap, 0xFFFFFFFFu);
    for (int i = 0; i < 2 * n1; i++) k12[i] = dmatch_to_key(knn12[i]);
    for (int i = 0; i < 2 * n2; i++) k21[i] = dmatch_to_key(knn21[i]);
    const int32_t nk[2] = {n1, n2};
    rc = vis_ensure_pin(ctx, (size_t)kcap * (2 * sizeof(vis_keypoint) + 16 + sizeof(vis_dmatch)) + (size_t)ncell * sizeof(vis_dmatch) + (size_t)root * 8 + 8192);
    if (rc) { tp = Plan(); return rc; }
    HostStage hs(ctx);
    hs.up(tp.d_kps, kps1, (size_t)n1 * sizeof(vis_keypoint));
    hs.up(tp.d_kps + kcap, kps2, (size_t)n2 * sizeof(vis_keypoint));
    hs.up(tp.d_nkp, nk, 8);
    hs.flush_ups();
    hipLaunchKernelGGL(k_set_pair, dim3(1), dim3(1), 0, ctx->stream, tp.d_pair_q, tp.d_pair_t, 0, 1);
    hs.up(tp.d_knn12, k12.data(), k12.size() * 4);
    hs.up(tp.d_knn21, k21.data(), k21.size() * 4);
    hs.up(tp.d_hf, hf.data(), (size_t)root * 4);
    hs.up(tp.d_wf, wf.data(), (size_t)root * 4);
    hs.flush_ups();
    rc = launch_filter(ctx, &tp, 1);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); tp = Plan(); return rc; }
    const MatchFetch f = fetch_matches(hs, &tp, 0, good != nullptr, cap, sym_out != nullptr, sym_cap);
    tp = Plan();
    rc = hs.wait();
    if (rc) return rc;
    return deliver_matches(f, good, cap, n_good, sym_out, sym_cap, n_sym);
}

// shared by vis_essential_ransac / vis_recover_pose
static int pose_host(vis_ctx* ctx, const float* p1xy, const float* p2xy, int m, const double* E_in,
                     int do_ransac, int do_pose, uint8_t* mask, PoseOut* out) {
    (void)hipSetDevice(ctx->device);
    if (m > VIS_RANSAC_MAX_M) return VIS_E_CAPACITY;
    const int mcap = std::max(m, 1);
    const int iters = ctx->p.ransac_max_iters;
    // scratch sized for a power-of-two correspondence count >= 256: a stream whose match count creeps up does not re-allocate per call
    int mroom = 256; while (mroom < mcap) mroom <<= 1;
    size_t need = (size_t)mroom * (16 + 32 + 1) + (size_t)iters * (20 + 720 + 40 + 8 * VIS_HYP_DOUBLES) + 65536;
    int rc = ensure_scratch(ctx, need);
    if (rc) return rc;
    Carver cv{(char*)ctx->d_scratch, 0};
    float* d_p1 = cv.take<float>((size_t)mcap * 2); float* d_p2 = cv.take<float>((size_t)mcap * 2);
    int32_t* d_npts = cv.take<int32_t>(1);
    double* d_n1 = cv.take<double>((size_t)mcap * 2); double* d_n2 = cv.take<double>((size_t)mcap * 2);
    int32_t* d_samples = cv.take<int32_t>((size_t)iters * 5);
    double* d_models = cv.take<double>((size_t)iters * 90);
    int32_t* d_counts = cv.take<int32_t>((size_t)iters * 10);
    int32_t* d_rstate = cv.take<int32_t>(VIS_RSTATE_WORDS);
    double* d_E = cv.take<double>(9);
    uint8_t* d_mask = cv.take<uint8_t>(mcap);
    PoseOut* d_pose = cv.take<PoseOut>(1);
    int32_t* d_worklist = cv.take<int32_t>(3);
    double* d_hyp = cv.take<double>((size_t)iters * VIS_HYP_DOUBLES);
    rc = vis_ensure_pin(ctx, (size_t)mcap * 17 + sizeof(PoseOut) + 4096);
    if (rc) return rc;
    HostStage hs(ctx);
    hs.up(d_p1, p1xy, (size_t)m * 8);
    hs.up(d_p2, p2xy, (size_t)m * 8);
    const int32_t mm = m;
    hs.up(d_npts, &mm, 4);
    if (E_in) hs.up(d_E, E_in, 72);
    hs.flush_ups();
    rc = vis_build_sample_table(ctx, VIS_POSE_TABLE_M);           // (already there since vis_set_params / the first call: never rebuilt because M grew)
    if (rc) return rc;
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[6], ctx->stream);
    rc = pose_run(ctx, 1, mcap, iters, d_p1, d_p2, d_npts, d_n1, d_n2, d_samples, d_models, d_counts, d_rstate,
                  E_in ? d_E : nullptr, d_mask, d_pose, do_ransac, do_pose, d_worklist, d_hyp);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[7], ctx->stream);
    const void* h_pose = hs.down(d_pose, sizeof(PoseOut));
    const void* h_mask = mask && m ? hs.down(d_mask, (size_t)m) : nullptr;
    rc = hs.wait();
    if (rc) return rc;
    { float a = 0; if (ctx->ev_ok && ev_elapsed(&a, ctx->ev[6], ctx->ev[7])) ctx->tm.ms_pose = a; }
    std::memcpy(out, h_pose, sizeof(PoseOut));
    if (h_mask) std::memcpy(mask, h_mask, (size_t)m);
    return VIS_OK;
}

extern "C" int vis_essential_ransac(vis_ctx* ctx, const float* p1xy, const float* p2xy, int m,
                                    double E[9], uint8_t* mask, int* n_inliers, int* iters_run) {
    if (!ctx || m < 0 || (m && (!p1xy || !p2xy)) || !E) return VIS_E_INVALID;
    PoseOut o;
    int rc = pose_host(ctx, p1xy, p2xy, m, nullptr, 1, 0, mask, &o);
    if (rc) return rc;
    std::memcpy(E, o.E, 72);
    if (n_inliers) *n_inliers = o.n_inliers;
    if (iters_run) *iters_run = o.iters_run;
    return VIS_OK;
}

extern "C" int vis_recover_pose(vis_ctx* ctx, const double E[9], const float* p1xy, const float* p2xy, int m,
                                double R[9], double t[3], int* n_good) {
    if (!ctx || !E || m < 0 || (m && (!p1xy || !p2xy)) || !R || !t) return VIS_E_INVALID;
    PoseOut o;
    int rc = pose_host(ctx, p1xy, p2xy, m, E, 0, 1, nullptr, &o);
    if (rc) return rc;
    std::memcpy(R, o.R, 72); std::memcpy(t, o.t, 24);
    if (n_good) *n_good = o.n_pose_good;
    return VIS_OK;
}

extern "C" int vis_f2f_ransac(vis_ctx* ctx, const vis_keypoint* pts1, const vis_keypoint* pts2, int m, const float rot[9],
                              const int32_t* sample_idx, int iters, float scale, float out_t[3], int* count_max) {
    if (!ctx || !out_t || m < 0 || iters < 0 || !rot || (iters && !sample_idx) || (m && (!pts1 || !pts2))) return VIS_E_INVALID;
    out_t[0] = out_t[1] = out_t[2] = 0.f;
    if (count_max) *count_max = 0;
    if (m < 2 || iters == 0) return VIS_OK;                       // SPEC: M < 2 -> zero vector
    for (int i = 0; i < 2 * iters; i++) if (sample_idx[i] < 0 || sample_idx[i] >= m) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    int rc = ensure_scratch(ctx, (size_t)m * (2 * sizeof(vis_keypoint) + 24) + (size_t)iters * 24 + 8192);
    if (rc) return rc;
    Carver cv{(char*)ctx->d_scratch, 0};
    vis_keypoint* d1 = cv.take<vis_keypoint>(m); vis_keypoint* d2 = cv.take<vis_keypoint>(m);
    float* d_rot = cv.take<float>(9); int32_t* d_idx = cv.take<int32_t>((size_t)iters * 2);
    double* d_nv = cv.take<double>((size_t)m * 3); float* d_cnt = cv.take<float>((size_t)iters * 4);
    HIPCHK(ctx, hipMemcpy(d1, pts1, (size_t)m * sizeof(vis_keypoint), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(d2, pts2, (size_t)m * sizeof(vis_keypoint), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(d_rot, rot, 36, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(d_idx, sample_idx, (size_t)iters * 8, hipMemcpyHostToDevice));
    rc = f2f_run(ctx, d1, d2, m, d_rot, d_idx, iters, d_nv, d_cnt);
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<float> c((size_t)iters * 4);
    HIPCHK(ctx, hipMemcpy(c.data(), d_cnt, c.size() * 4, hipMemcpyDeviceToHost));
    // `if (count > countMax)`: first strictly larger count wins, iterations in order (src/VISystem.cpp:737-741)
    float countMax = 0; float best[3] = {0, 0, 0};
    for (int i = 0; i < iters; i++)
        if (c[4 * (size_t)i] > countMax) { countMax = c[4 * (size_t)i]; best[0] = c[4 * (size_t)i + 1]; best[1] = c[4 * (size_t)i + 2]; best[2] = c[4 * (size_t)i + 3]; }
    for (int k = 0; k < 3; k++) out_t[k] = scale * best[k];
    if (count_max) *count_max = (int)countMax;
    return VIS_OK;
}

// blocking downloads of the batch getters (vis_batch_get_*: test / inspection entry points behind a full synchronisation)
static int download_knn(vis_ctx* ctx, const uint32_t* d_keys, int n, vis_dmatch* out) {
    if (!out || n <= 0) return VIS_OK;
    std::vector<uint32_t> k(2 * (size_t)n);
    HIPCHK(ctx, hipMemcpy(k.data(), d_keys, k.size() * 4, hipMemcpyDeviceToHost));
    keys_to_dmatches(k.data(), n, out);
    return VIS_OK;
}
static int download_matches(vis_ctx* ctx, Plan* pl, int pair, vis_dmatch* good, int cap, int* n_good,
                            vis_dmatch* sym_out, int sym_cap, int* n_sym) {
    int32_t ng = 0, ns = 0;
    HIPCHK(ctx, hipMemcpy(&ng, pl->d_ngood + pair, 4, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(&ns, pl->d_nsym + pair, 4, hipMemcpyDeviceToHost));
    if (n_good) *n_good = ng;
    if (n_sym) *n_sym = ns;
    if (good) {
        if (ng > cap) return VIS_E_CAPACITY;
        if (ng) HIPCHK(ctx, hipMemcpy(good, pl->d_good + (size_t)pair * pl->root * pl->root, (size_t)ng * sizeof(vis_dmatch), hipMemcpyDeviceToHost));
    }
    if (sym_out) {
        if (ns > sym_cap) return VIS_E_CAPACITY;
        if (ns) HIPCHK(ctx, hipMemcpy(sym_out, pl->d_sym + (size_t)pair * pl->kcap, (size_t)ns * sizeof(vis_dmatch), hipMemcpyDeviceToHost));
    }
    return VIS_OK;
}

// ------------------------------------------------------------------------------------------------ batch API
extern "C" int vis_batch_plan(vis_ctx* ctx, int w, int h, int stride, int max_frames) {
    if (!ctx || max_frames < 1 || max_frames > 4096) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    sync_all(ctx);
    plan_destroy(ctx->batch); ctx->batch = nullptr;
    return plan_create(ctx, w, h, stride, max_frames, max_frames + 1, max_frames, &ctx->batch, VIS_BATCH_SETS);
}

extern "C" int vis_batch_reset(vis_ctx* ctx) {
    if (!ctx || !ctx->batch) return VIS_E_STATE;
    sync_all(ctx);
    ctx->batch->have_prev = false; ctx->batch->last_n = 0; ctx->batch->carry_from = 0; ctx->batch->pair0_valid = false;
    if (ctx->batch->speculate) {                                   // a new stream: no prediction
        std::vector<int32_t> t0((size_t)ctx->batch->L, ctx->p.fast_threshold);
        HIPCHK(ctx, hipMemcpy(ctx->batch->d_tau, t0.data(), t0.size() * 4, hipMemcpyHostToDevice));
    }
    return VIS_OK;
}

// Three streams per context: A = detect chain (caller's / own stream), M = expand + knn + filters,
// P = RANSAC + recoverPose.  Batch i+1's detect chain overlaps batch i's matcher (MFMA + LDS, little VALU)
// and pose (latency-bound FP64): the records (keypoints, descriptors, expanded descriptors) are double
// buffered, everything else is single buffered and ordered with events.
extern "C" int vis_batch_run(vis_ctx* ctx, const uint8_t* d_frames, int n, int stages) {
    if (!ctx || !d_frames) return VIS_E_INVALID;
    Plan* pl = ctx->batch;
    if (!pl) return VIS_E_STATE;
    if (n < 1 || n > pl->B || ((uintptr_t)d_frames & 3)) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    hipStream_t sA = ctx->stream, sM = ctx->match_stream, sP = ctx->pose_stream;
    ctx->tm.launches_total = 0;
    const bool detect = (stages & VIS_STAGE_DETECT) != 0;
    const int cur = detect ? (pl->run_count % pl->nsets) : (pl->last_base / pl->rec_per_set);   // run_count is committed only when every launch succeeded
    const int base = cur * pl->rec_per_set;
    int rc = VIS_OK;
    bool have_prev = pl->have_prev;
    if (detect) {
        // this record set was last read by the matcher nsets batches ago: launch_detect waits for it where the chain first writes the set
        if (pl->carry_from > 0) have_prev = true;      // previous batch's last frame -> record 0 of this set: copied by launch_detect's first kernel
    }
    // Camera::Update (src/Camera.cpp:63-72): the half pyramid of every frame of the batch.  Nothing of the detect chain reads it
    // and it is pure streaming work, so it runs on a stream of its own beside the (vector-ALU bound) detect kernels; the detect
    // stream joins it at the end of its chain, so "the detect stream is done" still means "d_frames may be reused".
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[0], sA);
    if (detect) { VisRange r_("vis: ORB detect + describe"); rc = launch_detect(ctx, pl, d_frames, n, base + 1, pl->carry_from > 0 ? pl->carry_from : -1,
                                                                             (stages & (VIS_STAGE_UPDATE | VIS_STAGE_GRADIENT)) ? ctx->ev_update_fork : nullptr,
                                                                             pl->match_pending[cur] ? ctx->ev_match_done[cur] : nullptr); if (rc) return rc; }
    else if (ctx->ev_ok) for (int i = 1; i <= 4; i++) (void)hipEventRecord(ctx->ev[i], sA);
    hipStream_t sU = ctx->update_stream;
    pl->half_valid = false; pl->grad_valid = false;
    bool update_queued = false;
    if (stages & (VIS_STAGE_UPDATE | VIS_STAGE_GRADIENT)) {
        VisRange r_("vis: Camera::Update half pyramid");
        // this step's half pyramid / gradient set: the one the step before last used.  The plan's view of it (grad_set, d_half ... d_g) is
        // switched only once every launch of the stage has been queued; a failure leaves the last step's set and pointers in place
        const int gs = pl->grad_set ^ 1;
        const size_t fe = vis_grad_frame_elems(pl->w, pl->h);
        if (!pl->d_half_set[gs]) HIPCHK(ctx, hipMalloc((void**)&pl->d_half_set[gs], (size_t)pl->B * fe));
        if ((stages & VIS_STAGE_GRADIENT) && !pl->d_gx_set[gs]) {
            HIPCHK(ctx, hipMalloc((void**)&pl->d_gx_set[gs], (size_t)pl->B * fe * 2)); HIPCHK(ctx, hipMalloc((void**)&pl->d_gy_set[gs], (size_t)pl->B * fe * 2));
            HIPCHK(ctx, hipMalloc((void**)&pl->d_g_set[gs], (size_t)pl->B * fe));
        }
        // the frames were produced on the detect stream (or before the call): the side stream is ordered behind it -- behind the
        // PYRAMID launches of this batch's detect chain (launch_detect recorded the event there): the resize chain streams at HBM
        // speed itself, the kernels after it are vector-ALU bound, and that is where streaming work fits beside them
        if (!detect) HIPCHK(ctx, hipEventRecord(ctx->ev_update_fork, sA));
        HIPCHK(ctx, hipStreamWaitEvent(sU, ctx->ev_update_fork, 0));
        // an alignment two steps back (vis_batch_align on the pose stream) may still read THIS set's half pyramid / gradients
        if (pl->grad_reader[gs]) HIPCHK(ctx, hipStreamWaitEvent(sU, pl->grad_reader[gs], 0));
        if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[10], sU);
        ctx->stream = sU;
        rc = launch_half_pyramid_batch(ctx, d_frames, pl->w, pl->h, pl->stride, (size_t)pl->stride * pl->h, n, pl->d_half_set[gs]);
        if (!rc && (stages & VIS_STAGE_GRADIENT))
            rc = launch_gradient(ctx, d_frames, pl->w, pl->h, pl->stride, (size_t)pl->stride * pl->h, n, pl->d_half_set[gs], 3 /* the reference's Scharr scale, src/Camera.cpp:172 */,
                                 pl->d_gx_set[gs], pl->d_gy_set[gs], pl->d_g_set[gs]);
        ctx->stream = sA;
        if (rc) {
            // whatever was queued on the side stream before the failure still reads d_frames / writes set gs: the detect stream joins
            // it before this call returns, so "ctx->stream is done" keeps meaning "the batch's buffers are free"
            if (hipEventRecord(ctx->ev_update_done, sU) == hipSuccess) (void)hipStreamWaitEvent(sA, ctx->ev_update_done, 0);
            return rc;
        }
        pl->grad_set = gs; pl->d_half = pl->d_half_set[gs];
        if (stages & VIS_STAGE_GRADIENT) { pl->d_gx = pl->d_gx_set[gs]; pl->d_gy = pl->d_gy_set[gs]; pl->d_g = pl->d_g_set[gs]; pl->grad_valid = true; }
        if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[11], sU);
        HIPCHK(ctx, hipEventRecord(ctx->ev_update_done, sU));
        update_queued = true;
        pl->half_valid = true;
    }
    // the matcher needs the records (detect chain), not the side stream's half pyramid / gradients: its event comes BEFORE the join (with
    // the join in front, the matcher -- and through it the alignment and the next step's gradients -- waited for the gradient stage)
    HIPCHK(ctx, hipEventRecord(ctx->ev_detect_done, sA));
    if (update_queued) HIPCHK(ctx, hipStreamWaitEvent(sA, ctx->ev_update_done, 0));
    if (stages & (VIS_STAGE_MATCH | VIS_STAGE_POSE)) HIPCHK(ctx, hipStreamWaitEvent(sM, ctx->ev_detect_done, 0));
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev_match_start, sM);
    // pair i: query = record base+i (frame i-1, or the carried frame for i = 0), train = record base+i+1 (frame i)
    pl->d_pair_q = have_prev ? pl->d_pq[cur] : pl->d_pqn[cur];
    pl->d_pair_t = pl->d_pt[cur];
    // the matcher-output set of this step: a step that runs the matcher takes the next one, a pose-only step reads the last one
    const int mo = pl->mo_set[1][0] ? ((stages & VIS_STAGE_MATCH) ? (pl->last_cur ^ 1) : pl->last_cur) : 0;
    auto use_mo = [&](int s_) {
        pl->d_sym = (vis_dmatch*)pl->mo_set[s_][0]; pl->d_nsym = (int32_t*)pl->mo_set[s_][1]; pl->d_good = (vis_dmatch*)pl->mo_set[s_][2];
        pl->d_ngood = (int32_t*)pl->mo_set[s_][3]; pl->d_p1 = (float*)pl->mo_set[s_][4]; pl->d_p2 = (float*)pl->mo_set[s_][5];
    };
    use_mo(mo);
    ctx->stream = sM;
    if (stages & VIS_STAGE_MATCH) {
        VisRange r_("vis: knn + match filters");
        rc = launch_expand(ctx, pl, base, n + 1);
        if (!rc) rc = launch_match(ctx, pl, n);
        if (!rc) {
            if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[5], sM);
            // the filter writes matcher-output set `mo` (sets in turn, like the records): wait for what still reads THAT set -- the pose
            // stage, the alignment and the results download of the step before last; those of the last step read the other set
            if (pl->mo_pose[mo]) { hipError_t e = hipStreamWaitEvent(sM, pl->mo_pose[mo], 0); if (e != hipSuccess) rc = VIS_E_HIP; }
            if (!rc && pl->mo_align[mo]) { hipError_t e = hipStreamWaitEvent(sM, pl->mo_align[mo], 0); if (e != hipSuccess) rc = VIS_E_HIP; }
            if (!rc && pl->mo_results[mo]) { hipError_t e = hipStreamWaitEvent(sM, pl->mo_results[mo], 0); if (e != hipSuccess) rc = VIS_E_HIP; }
            if (!rc) rc = launch_filter(ctx, pl, n);
        }
        if (!rc && ctx->ev_ok) (void)hipEventRecord(ctx->ev[6], sM);
        if (!rc) { (void)hipEventRecord(ctx->ev_match_done[cur], sM); pl->match_pending[cur] = true; }
    } else if (ctx->ev_ok) { (void)hipEventRecord(ctx->ev[5], sM); (void)hipEventRecord(ctx->ev[6], sM); }
    if (!rc && (stages & VIS_STAGE_POSE)) {
        (void)hipEventRecord(ctx->ev_filter_done, sM);
        (void)hipStreamWaitEvent(sP, ctx->ev_filter_done, 0);
        ctx->stream = sP;
        (void)hipEventRecord(ctx->ev_pose_start, sP);
        VisRange r_("vis: essential RANSAC + recoverPose");
        rc = launch_pose(ctx, pl, n);
        (void)hipEventRecord(ctx->ev_pose_done, sP);
        ctx->pose_pending = true;
        if (hipEventRecord(ctx->ev_pose_done_set[mo], sP) == hipSuccess) pl->mo_pose[mo] = ctx->ev_pose_done_set[mo];
    }
    ctx->stream = sA;
    if (rc) { use_mo(pl->last_cur); pl->half_valid = pl->grad_valid = false; return rc; }   // nothing of the stream state (carried frame, record set, matcher-output set) has been committed
    pl->have_prev = have_prev; pl->pair0_valid = have_prev;
    if (detect) { pl->run_count++; pl->carry_from = base + n; }
    if (ctx->ev_ok) (void)hipEventRecord(ctx->ev[8], sA);
    pl->last_n = n; pl->last_base = base; pl->last_cur = mo;
    return VIS_OK;
}

extern "C" int vis_batch_sync(vis_ctx* ctx) {
    if (!ctx) return VIS_E_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->update_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->update_stream));
    if (ctx->match_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->match_stream));
    const bool had_pose = ctx->pose_pending;
    if (ctx->pose_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->pose_stream));
    ctx->pose_pending = false; ctx->results_pending = false; ctx->align_pending = false;
    if (ctx->batch) { for (int i = 0; i < VIS_BATCH_SETS; i++) ctx->batch->match_pending[i] = false; ctx->batch->grad_reader[0] = ctx->batch->grad_reader[1] = nullptr;
                      for (int i = 0; i < 2; i++) ctx->batch->mo_pose[i] = ctx->batch->mo_results[i] = ctx->batch->mo_align[i] = nullptr; }
    if (ctx->ev_ok) {
        collect_detect_timings(ctx);
        float a = 0;
        if (ev_elapsed(&a, ctx->ev_match_start, ctx->ev[5])) ctx->tm.ms_knn = a;
        if (ev_elapsed(&a, ctx->ev[5], ctx->ev[6])) ctx->tm.ms_filter = a;
        ctx->tm.ms_pose = 0;
        if (had_pose && ev_elapsed(&a, ctx->ev_pose_start, ctx->ev_pose_done)) ctx->tm.ms_pose = a;
        ctx->tm.ms_update = 0;
        const bool upd = ctx->batch && ctx->batch->half_valid;
        if (upd && ev_elapsed(&a, ctx->ev[10], ctx->ev[11])) ctx->tm.ms_update = a;
        hipEvent_t e0 = ctx->ev[0];
        if (ev_elapsed(&a, e0, ctx->ev[8])) ctx->tm.ms_total = a;
        if (ev_elapsed(&a, e0, ctx->ev[6]) && a > ctx->tm.ms_total) ctx->tm.ms_total = a;
        if (had_pose && ev_elapsed(&a, e0, ctx->ev_pose_done) && a > ctx->tm.ms_total) ctx->tm.ms_total = a;
    }
    return VIS_OK;
}

extern "C" int vis_batch_half_pyramid(vis_ctx* ctx, const uint8_t** d_half, size_t* frame_elems) {
    if (!ctx || !ctx->batch || !d_half) return VIS_E_STATE;
    Plan* pl = ctx->batch;
    if (!pl->half_valid || !pl->d_half) return VIS_E_STATE;
    *d_half = pl->d_half;
    if (frame_elems) *frame_elems = vis_grad_frame_elems(pl->w, pl->h);
    return VIS_OK;
}

extern "C" int vis_batch_gradients(vis_ctx* ctx, const uint8_t** d_gray, const int16_t** d_gx, const int16_t** d_gy, const uint8_t** d_g, size_t* frame_elems) {
    if (!ctx || !ctx->batch) return VIS_E_STATE;
    Plan* pl = ctx->batch;
    if (!pl->grad_valid || !pl->d_gx) return VIS_E_STATE;
    if (d_gray) *d_gray = pl->d_half;
    if (d_gx) *d_gx = pl->d_gx;
    if (d_gy) *d_gy = pl->d_gy;
    if (d_g) *d_g = pl->d_g;
    if (frame_elems) *frame_elems = vis_grad_frame_elems(pl->w, pl->h);
    return VIS_OK;
}

extern "C" int vis_batch_results_async(vis_ctx* ctx, vis_pose_result* h_pose, vis_dmatch* h_good, int32_t* h_ngood, int n_cap) {
    if (!ctx || !ctx->batch) return VIS_E_STATE;
    Plan* pl = ctx->batch;
    const int n = pl->last_n;
    if (n < 1) return VIS_E_STATE;
    if (n_cap < n) { ctx->err = "vis_batch_results_async: caller buffers hold fewer frames than the last batch"; return VIS_E_CAPACITY; }
    (void)hipSetDevice(ctx->device);
    // the pose stream is ordered behind the matcher of the same batch (ev_filter_done); without a pose stage the matcher's own stream
    hipStream_t s = ctx->pose_pending ? ctx->pose_stream : ctx->match_stream;
    const int ncell = pl->root * pl->root;
    // Pinned (device-accessible) destinations are written by ONE small kernel of the library instead of three hipMemcpyAsync: every
    // device-to-host copy of the runtime ends in a system-scope release, and with one per step the whole pipeline lost 8 % (round 5:
    // 402 k -> 371 k frames/s with a single 196 KB copy per step).  The host reads after vis_batch_sync, which orders it.  Anything
    // else (pageable memory, unaligned pointers) goes through the runtime's copies as before.
    void* dsts[3]; const void* srcs[3]; size_t bytes[3]; int nj = 0; bool mapped = true;
    auto add = [&](void* h, const void* d, size_t b) {
        if (!h) return;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, h) != hipSuccess) { (void)hipGetLastError(); mapped = false; }
        else if (at.type != hipMemoryTypeHost || !at.devicePointer || ((uintptr_t)at.devicePointer & 3)) mapped = false;
        else { dsts[nj] = at.devicePointer; }
        if (!mapped) dsts[nj] = h;
        srcs[nj] = d; bytes[nj] = b; nj++;
    };
    add(h_pose, pl->d_pose, (size_t)n * sizeof(vis_pose_result));
    add(h_good, pl->d_good, (size_t)n * ncell * sizeof(vis_dmatch));
    add(h_ngood, pl->d_ngood, (size_t)n * sizeof(int32_t));
    if (nj && mapped) { const int rc = launch_copy_jobs(ctx, s, nj, dsts, srcs, bytes); if (rc) return rc; ctx->n_copies++; }
    else {
        if (h_pose) HIPCHK(ctx, hipMemcpyAsync(h_pose, pl->d_pose, (size_t)n * sizeof(vis_pose_result), hipMemcpyDeviceToHost, s));
        if (h_good) HIPCHK(ctx, hipMemcpyAsync(h_good, pl->d_good, (size_t)n * ncell * sizeof(vis_dmatch), hipMemcpyDeviceToHost, s));
        if (h_ngood) HIPCHK(ctx, hipMemcpyAsync(h_ngood, pl->d_ngood, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_results_done, s));
    ctx->results_pending = true;
    HIPCHK(ctx, hipEventRecord(ctx->ev_results_done_set[pl->last_cur], s));
    pl->mo_results[pl->last_cur] = ctx->ev_results_done_set[pl->last_cur];
    return VIS_OK;
}

extern "C" int vis_batch_fast_thresholds(vis_ctx* ctx, int32_t* tau_next, int32_t* n_redone) {
    if (!ctx || !ctx->batch) return VIS_E_STATE;
    Plan* pl = ctx->batch;
    sync_all(ctx);
    if (!pl->speculate) { if (n_redone) *n_redone = 0; if (tau_next) for (int l = 0; l < pl->L; l++) tau_next[l] = ctx->p.fast_threshold; return VIS_OK; }
    if (tau_next) HIPCHK(ctx, hipMemcpy(tau_next, pl->d_tau, (size_t)pl->L * 4, hipMemcpyDeviceToHost));
    if (n_redone) HIPCHK(ctx, hipMemcpy(n_redone, pl->d_fix, 4, hipMemcpyDeviceToHost));
    return VIS_OK;
}

extern "C" int vis_batch_status(vis_ctx* ctx, int* flags) {
    if (!ctx || !ctx->batch || !flags) return VIS_E_STATE;
    sync_all(ctx);
    int32_t fl = 0;
    HIPCHK(ctx, hipMemcpy(&fl, ctx->batch->d_flags, 4, hipMemcpyDeviceToHost));
    *flags = fl;
    return VIS_OK;
}

extern "C" int vis_batch_get_keypoints(vis_ctx* ctx, int frame, vis_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    if (!ctx || !ctx->batch) return VIS_E_STATE;
    Plan* pl = ctx->batch;
    if (frame < 0 || frame >= pl->last_n) return VIS_E_INVALID;
    sync_all(ctx);
    int32_t n = 0;
    const int rbase = pl->last_base;
    HIPCHK(ctx, hipMemcpy(&n, pl->d_nkp + rbase + frame + 1, 4, hipMemcpyDeviceToHost));
    if (n_out) *n_out = n;
    if ((kps || desc) && n > cap) return VIS_E_CAPACITY;
    if (kps && n) HIPCHK(ctx, hipMemcpy(kps, pl->d_kps + (size_t)(rbase + frame + 1) * pl->kcap, (size_t)n * sizeof(vis_keypoint), hipMemcpyDeviceToHost));
    if (desc && n) HIPCHK(ctx, hipMemcpy(desc, pl->d_desc + (size_t)(rbase + frame + 1) * pl->kcap * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
    return VIS_OK;
}

extern "C" int vis_batch_get_knn(vis_ctx* ctx, int frame, vis_dmatch* out12, int cap12, int* n12,
                                 vis_dmatch* out21, int cap21, int* n21) {
    if (!ctx || !ctx->batch) return VIS_E_STATE;
    Plan* pl = ctx->batch;
    if (frame < 0 || frame >= pl->last_n) return VIS_E_INVALID;
    sync_all(ctx);
    int32_t nq = 0, nt = 0;
    HIPCHK(ctx, hipMemcpy(&nq, pl->d_nkp + pl->last_base + frame, 4, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(&nt, pl->d_nkp + pl->last_base + frame + 1, 4, hipMemcpyDeviceToHost));
    if (frame == 0 && !pl->pair0_valid) { nq = 0; nt = 0; }      // first frame of a stream has no pair
    if (n12) *n12 = nq;
    if (n21) *n21 = nt;
    if (out12) { if (2 * nq > cap12) return VIS_E_CAPACITY; int rc = download_knn(ctx, pl->d_knn12 + (size_t)frame * pl->kcap * 2, nq, out12); if (rc) return rc; }
    if (out21) { if (2 * nt > cap21) return VIS_E_CAPACITY; int rc = download_knn(ctx, pl->d_knn21 + (size_t)frame * pl->kcap * 2, nt, out21); if (rc) return rc; }
    return VIS_OK;
}

extern "C" int vis_batch_get_matches(vis_ctx* ctx, int frame, vis_dmatch* good, int cap, int* n_good, int* n_sym) {
    if (!ctx || !ctx->batch) return VIS_E_STATE;
    Plan* pl = ctx->batch;
    if (frame < 0 || frame >= pl->last_n) return VIS_E_INVALID;
    sync_all(ctx);
    return download_matches(ctx, pl, frame, good, cap, n_good, nullptr, 0, n_sym);
}

// the inlier mask findEssentialMat left for pair `frame` (one byte per correspondence the pose stage saw: the grid-filtered good
// matches, or the symmetric matches in their list order with VIS_POSE_SYM)
extern "C" int vis_batch_get_inlier_mask(vis_ctx* ctx, int frame, uint8_t* mask, int cap, int* n_points) {
    if (!ctx || !ctx->batch) return VIS_E_STATE;
    Plan* pl = ctx->batch;
    if (frame < 0 || frame >= pl->last_n) return VIS_E_INVALID;
    sync_all(ctx);
    PoseOut o;
    HIPCHK(ctx, hipMemcpy(&o, pl->d_pose + frame, sizeof(PoseOut), hipMemcpyDeviceToHost));
    const int n = std::min(std::max(o.n_points, 0), pl->pose_mcap);
    if (n_points) *n_points = n;
    if (mask) {
        if (n > cap) return VIS_E_CAPACITY;
        if (n) HIPCHK(ctx, hipMemcpy(mask, pl->d_mask + (size_t)frame * pl->pose_mcap, (size_t)n, hipMemcpyDeviceToHost));
    }
    return VIS_OK;
}

extern "C" int vis_batch_get_pose(vis_ctx* ctx, int frame, double E[9], double R[9], double t[3],
                                  int* n_inliers, int* n_pose_good, int* iters_run) {
    if (!ctx || !ctx->batch) return VIS_E_STATE;
    Plan* pl = ctx->batch;
    if (frame < 0 || frame >= pl->last_n) return VIS_E_INVALID;
    sync_all(ctx);
    PoseOut o;
    HIPCHK(ctx, hipMemcpy(&o, pl->d_pose + frame, sizeof(PoseOut), hipMemcpyDeviceToHost));
    if (E) std::memcpy(E, o.E, 72);
    if (R) std::memcpy(R, o.R, 72);
    if (t) std::memcpy(t, o.t, 24);
    if (n_inliers) *n_inliers = o.n_inliers;
    if (n_pose_good) *n_pose_good = o.n_pose_good;
    if (iters_run) *iters_run = o.iters_run;
    return VIS_OK;
}
