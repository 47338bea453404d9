// ingest.hip -- frame ingest for the throughput path: image directory listing and timestamp
// stems as /root/reference/src/ImageReader.cpp:22-78 does them, a PGM (P5) / raw 8-bit / greyscale PNG reader in place of
// cv::imread(..., CV_LOAD_IMAGE_GRAYSCALE) (:80-82), and a pinned-host double-buffered H2D feeder so that batch k+1
// is copied while batch k is processed.  No kernels here: host code + HIP runtime copies.
// PNG (round 6): EuRoC ships cam0 as 8-bit greyscale PNG, which is what the reference's imread decodes.  The decoder here
// handles exactly the lossless cases in which imread's answer is defined by the PNG specification alone -- colour type 0 (grey)
// and 4 (grey + alpha: the alpha channel is dropped), bit depth 8 and 16 (the high byte, as libpng's strip_16), non-interlaced --
// chunk CRCs checked, zlib's inflate for the stream.  Colour PNGs are refused: imread would go through libpng's rgb_to_gray,
// whose rounding is not restated here.
#include "vis_internal.h"
#include <dirent.h>
#include <sys/stat.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <zlib.h>

// ---- directory listing (ImageReader::searchImages, :49-74) -------------------------------------------------------
// The reference sorts ALL directory entries and drops the first two, assuming they are "." and ".."; here the two
// entries are dropped by name and only regular files whose extension is .pgm / .raw are kept, still in byte order.
extern "C" int vis_image_list(const char* dir, char* names_out, int cap_bytes, int* count) {
    if (!dir || !count || cap_bytes < 0 || (cap_bytes > 0 && !names_out)) return VIS_E_INVALID;
    DIR* d = opendir(dir);
    if (!d) return VIS_E_STATE;
    std::vector<std::string> names;
    std::string base(dir);
    if (!base.empty() && base.back() != '/') base += '/';
    while (struct dirent* ent = readdir(d)) {
        const std::string n(ent->d_name);
        if (n == "." || n == "..") continue;
        const size_t dot = n.rfind('.');
        if (dot == std::string::npos) continue;
        std::string ext = n.substr(dot);
        for (auto& c : ext) c = (char)tolower((unsigned char)c);
        if (ext != ".pgm" && ext != ".raw" && ext != ".png") continue;
        struct stat st;
        if (stat((base + n).c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
        names.push_back(n);
    }
    closedir(d);
    std::sort(names.begin(), names.end());
    *count = (int)names.size();
    size_t need = 0;
    for (auto& n : names) need += n.size() + 1;
    if (need > (size_t)cap_bytes) return names_out ? VIS_E_CAPACITY : VIS_OK;
    char* o = names_out;
    for (auto& n : names) { std::memcpy(o, n.c_str(), n.size()); o += n.size(); *o++ = '\n'; }
    if (need < (size_t)cap_bytes) *o = 0;
    return VIS_OK;
}

// ImageReader::getImageName + getImageTime (:22-47): strip the directory, cut at the FIRST '.', atol
extern "C" long vis_image_time(const char* file_name) {
    if (!file_name) return 0;
    const char* s = std::strrchr(file_name, '/');
    s = s ? s + 1 : file_name;
    std::string stem(s);
    const size_t dot = stem.find('.');
    if (dot != std::string::npos && dot != 0) stem = stem.substr(0, dot);
    return std::atol(stem.c_str());
}

// ---- PGM (P5, maxval <= 255) -------------------------------------------------------------------------------------
static bool pgm_token(FILE* f, int* v) {               // next unsigned integer of the header, '#' comments skipped
    int c;
    for (;;) {
        c = fgetc(f);
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
        if (c == EOF) return false;
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') continue;
        break;
    }
    if (c < '0' || c > '9') return false;
    long x = 0;
    while (c >= '0' && c <= '9') { x = x * 10 + (c - '0'); if (x > 1 << 24) return false; c = fgetc(f); }
    *v = (int)x;                                       // the single whitespace after the token has been consumed
    return c == ' ' || c == '\t' || c == '\n' || c == '\r';
}

static int pgm_open(const char* path, FILE** out, int* w, int* h) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return VIS_E_STATE;
    int maxv = 0;
    if (fgetc(f) != 'P' || fgetc(f) != '5' || !pgm_token(f, w) || !pgm_token(f, h) || !pgm_token(f, &maxv) ||
        *w < 1 || *h < 1 || maxv < 1 || maxv > 255) { std::fclose(f); return VIS_E_INVALID; }
    *out = f;
    return VIS_OK;
}

extern "C" int vis_pgm_info(const char* path, int* w, int* h) {
    if (!path || !w || !h) return VIS_E_INVALID;
    FILE* f = nullptr;
    const int rc = pgm_open(path, &f, w, h);
    if (f) std::fclose(f);
    return rc;
}

// ---- PNG (greyscale, 8 / 16 bit, optional alpha, non-interlaced) ------------------------------------------------------
struct PngHead { int w = 0, h = 0, depth = 0, color = 0, interlace = 0; };
static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static bool is_png_magic(const uint8_t* p) { static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A}; return std::memcmp(p, sig, 8) == 0; }

// walks the chunks of a PNG file: header fields, and (idat != nullptr) the concatenated IDAT payload.  Every chunk's CRC is checked.
static int png_parse(const char* path, PngHead* hd, std::vector<uint8_t>* idat) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return VIS_E_STATE;
    std::vector<uint8_t> buf;
    {
        uint8_t tmp[65536]; size_t n;
        while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) { buf.insert(buf.end(), tmp, tmp + n); if (buf.size() > ((size_t)1 << 28)) break; }
        std::fclose(f);
        if (buf.size() > ((size_t)1 << 28)) return VIS_E_INVALID;          // (a 16384 x 16384 16-bit grey + alpha image is 1 GiB raw; no camera frame's file is 256 MiB)
    }
    if (buf.size() < 8 + 25 || !is_png_magic(buf.data())) return VIS_E_INVALID;
    size_t pos = 8; bool have_head = false, have_end = false;
    while (pos + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[pos]);
        if (len > buf.size() || pos + 12 + (size_t)len > buf.size()) return VIS_E_INVALID;                 // truncated
        const uint8_t* type = &buf[pos + 4]; const uint8_t* data = &buf[pos + 8];
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != be32(data + len)) return VIS_E_INVALID;
        if (!have_head) {
            if (std::memcmp(type, "IHDR", 4) != 0 || len != 13) return VIS_E_INVALID;
            hd->w = (int)be32(data); hd->h = (int)be32(data + 4); hd->depth = data[8]; hd->color = data[9]; hd->interlace = data[12];
            if (hd->w < 1 || hd->h < 1 || hd->w > (1 << 16) || hd->h > (1 << 16) || data[10] != 0 || data[11] != 0) return VIS_E_INVALID;
            have_head = true;
            if (!idat) return VIS_OK;
        } else if (std::memcmp(type, "IDAT", 4) == 0) idat->insert(idat->end(), data, data + len);
        else if (std::memcmp(type, "IEND", 4) == 0) { have_end = true; break; }
        pos += 12 + (size_t)len;
    }
    return have_head && have_end ? VIS_OK : VIS_E_INVALID;
}

static int png_read_impl(const char* path, uint8_t* out, int out_stride, int w, int h);
// (files are untrusted input: whatever they make the containers throw ends as an error code at the C boundary, never as an exception)
static int png_read(const char* path, uint8_t* out, int out_stride, int w, int h) {
    try { return png_read_impl(path, out, out_stride, w, h); } catch (...) { return VIS_E_NOMEM; }
}
static int png_read_impl(const char* path, uint8_t* out, int out_stride, int w, int h) {
    PngHead hd; std::vector<uint8_t> idat;
    int rc = png_parse(path, &hd, &idat);
    if (rc) return rc;
    if (hd.w != w || hd.h != h) return VIS_E_INVALID;
    if ((hd.color != 0 && hd.color != 4) || (hd.depth != 8 && hd.depth != 16) || hd.interlace != 0) return VIS_E_INVALID;   // see the file header
    const int bpp = (hd.color == 4 ? 2 : 1) * (hd.depth / 8);            // bytes per pixel = the filters' left-neighbour distance
    const size_t rowb = (size_t)w * bpp;
    std::vector<uint8_t> raw((rowb + 1) * (size_t)h);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) return VIS_E_INVALID;
    std::vector<uint8_t> zero(rowb, 0);
    const uint8_t* prev = zero.data();
    for (int y = 0; y < h; y++) {
        uint8_t* line = raw.data() + (size_t)y * (rowb + 1);
        const int ft = line[0]; uint8_t* cur = line + 1;
        if (ft > 4) return VIS_E_INVALID;
        for (size_t i = 0; i < rowb; i++) {                               // PNG specification 9.2: Sub, Up, Average, Paeth on bytes
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int pr = 0;
            if (ft == 1) pr = a; else if (ft == 2) pr = b; else if (ft == 3) pr = (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            cur[i] = (uint8_t)(cur[i] + pr);
        }
        uint8_t* o = out + (size_t)y * out_stride;
        for (int x = 0; x < w; x++) o[x] = cur[(size_t)x * bpp];          // grey sample, its high byte at 16 bit; alpha dropped
        prev = cur;
    }
    return VIS_OK;
}

// width and height of a PGM (P5) or PNG file, by its magic bytes
extern "C" int vis_image_info(const char* path, int* w, int* h) {
    if (!path || !w || !h) return VIS_E_INVALID;
    FILE* f = std::fopen(path, "rb");
    if (!f) return VIS_E_STATE;
    uint8_t m[8] = {0}; const size_t n = std::fread(m, 1, 8, f);
    std::fclose(f);
    if (n == 8 && is_png_magic(m)) {
        PngHead hd; int rc;
        try { rc = png_parse(path, &hd, nullptr); } catch (...) { return VIS_E_NOMEM; }
        *w = hd.w; *h = hd.h; return rc;
    }
    return vis_pgm_info(path, w, h);
}

// reads a w x h image into out (row stride out_stride); a .raw file is w*h bytes without header, anything else is told by its magic bytes
extern "C" int vis_image_read(const char* path, uint8_t* out, int out_stride, int w, int h) {
    if (!path || !out || w < 1 || h < 1 || out_stride < w) return VIS_E_INVALID;
    const char* dot = std::strrchr(path, '.');
    const bool raw = dot && (std::strcmp(dot, ".raw") == 0 || std::strcmp(dot, ".RAW") == 0);
    FILE* f = nullptr;
    if (raw) { f = std::fopen(path, "rb"); if (!f) return VIS_E_STATE; }
    else {
        f = std::fopen(path, "rb");
        if (!f) return VIS_E_STATE;
        uint8_t m[8] = {0}; const size_t n = std::fread(m, 1, 8, f);
        std::fclose(f); f = nullptr;
        if (n == 8 && is_png_magic(m)) return png_read(path, out, out_stride, w, h);
        int fw = 0, fh = 0;
        const int rc = pgm_open(path, &f, &fw, &fh);
        if (rc) return rc;
        if (fw != w || fh != h) { std::fclose(f); return VIS_E_INVALID; }
    }
    for (int y = 0; y < h; y++)
        if (std::fread(out + (size_t)y * out_stride, 1, (size_t)w, f) != (size_t)w) { std::fclose(f); return VIS_E_INVALID; }   // truncated
    std::fclose(f);
    return VIS_OK;
}

// ---- pinned double-buffered H2D feeder -----------------------------------------------------------------------------
struct vis_feeder {
    vis_ctx* ctx; int w, h, batch; size_t frame_bytes;
    uint8_t* h_buf[2]; uint8_t* d_buf[2];
    hipStream_t copy_stream; hipEvent_t copied[2]; hipEvent_t consumed[2]; bool busy[2]; bool released[2];
};

extern "C" int vis_feeder_create(vis_ctx* ctx, int w, int h, int batch, vis_feeder** out) {
    if (!ctx || !out || w < 16 || h < 16 || (w & 3) || batch < 1) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    vis_feeder* f = new vis_feeder();
    f->ctx = ctx; f->w = w; f->h = h; f->batch = batch; f->frame_bytes = (size_t)w * h;
    for (int k = 0; k < 2; k++) { f->h_buf[k] = nullptr; f->d_buf[k] = nullptr; f->copied[k] = nullptr; f->consumed[k] = nullptr; f->busy[k] = false; f->released[k] = true; }
    f->copy_stream = nullptr;
    bool ok = hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < 2 && ok; k++) {
        ok = ok && hipHostMalloc((void**)&f->h_buf[k], f->frame_bytes * batch, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipMalloc((void**)&f->d_buf[k], f->frame_bytes * batch) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&f->copied[k], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&f->consumed[k], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        ctx->err = "vis_feeder_create: allocation failed";
        for (int k = 0; k < 2; k++) {
            if (f->h_buf[k]) (void)hipHostFree(f->h_buf[k]);
            if (f->d_buf[k]) (void)hipFree(f->d_buf[k]);
            if (f->copied[k]) (void)hipEventDestroy(f->copied[k]);
            if (f->consumed[k]) (void)hipEventDestroy(f->consumed[k]);
        }
        if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
        delete f;
        return VIS_E_NOMEM;
    }
    *out = f;
    return VIS_OK;
}

extern "C" void vis_feeder_destroy(vis_feeder* f) {
    if (!f) return;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->copy_stream);
    for (int k = 0; k < 2; k++) {
        (void)hipHostFree(f->h_buf[k]); (void)hipFree(f->d_buf[k]);
        (void)hipEventDestroy(f->copied[k]); (void)hipEventDestroy(f->consumed[k]);
    }
    (void)hipStreamDestroy(f->copy_stream);
    delete f;
}

// the pinned staging buffer the caller's reader fills (batch x h x w, dense).  Blocks until the previous copy out of
// this buffer has finished, so the buffer can be overwritten.
extern "C" uint8_t* vis_feeder_host_buffer(vis_feeder* f, int which) {
    if (!f || which < 0 || which > 1) return nullptr;
    if (f->busy[which]) (void)hipEventSynchronize(f->copied[which]);
    return f->h_buf[which];
}

// enqueue the H2D copy of the first n frames of buffer `which` and make the context's detect stream wait for it:
// a following vis_batch_run(ctx, *d_frames, n, ...) is ordered after the copy; the copy of the OTHER buffer overlaps
// the processing of this one.  The device buffer is reused only after the work queued on it so far has finished.
extern "C" int vis_feeder_submit(vis_feeder* f, int which, int n, const uint8_t** d_frames) {
    if (!f || which < 0 || which > 1 || n < 1 || n > f->batch || !d_frames) return VIS_E_INVALID;
    vis_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    // the previous use of this device buffer must have been released (vis_feeder_release records the event the copy waits
    // for): without it the next copy could overwrite frames the detect chain is still reading
    if (f->busy[which] && !f->released[which]) { ctx->err = "vis_feeder_submit: previous batch of this buffer was not released"; return VIS_E_STATE; }
    if (ctx->batch && (ctx->batch->w != f->w || ctx->batch->h != f->h || ctx->batch->stride != f->w)) {
        ctx->err = "vis_feeder_submit: the batch plan's geometry (w, h, stride == w) does not match the feeder"; return VIS_E_INVALID;
    }
    if (f->busy[which]) HIPCHK(ctx, hipStreamWaitEvent(f->copy_stream, f->consumed[which], 0));    // detect of the batch that used d_buf[which]
    if (ctx->align_pending) HIPCHK(ctx, hipStreamWaitEvent(f->copy_stream, ctx->ev_align_done, 0));  // ... and an alignment that may still read it
    HIPCHK(ctx, hipMemcpyAsync(f->d_buf[which], f->h_buf[which], f->frame_bytes * n, hipMemcpyHostToDevice, f->copy_stream));
    HIPCHK(ctx, hipEventRecord(f->copied[which], f->copy_stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, f->copied[which], 0));
    f->busy[which] = true; f->released[which] = false;
    *d_frames = f->d_buf[which];
    return VIS_OK;
}

// call after the vis_batch_run that consumes buffer `which`: marks the point on the detect stream after which the
// device buffer may be overwritten by the next copy into it
extern "C" int vis_feeder_release(vis_feeder* f, int which) {
    if (!f || which < 0 || which > 1) return VIS_E_INVALID;
    HIPCHK(f->ctx, hipEventRecord(f->consumed[which], f->ctx->stream));
    f->released[which] = true;
    return VIS_OK;
}
