// stage_selftest -- the two refusals of the frame-at-a-time staging block, driven on the CPU (no device, no HIP call is reached):
//   1. HostStage::take() beyond the pinned block -> nullptr, and the call's wait() answers VIS_E_NOMEM instead of copying past the block
//   2. vis_ensure_pin() asked to grow (= free + re-allocate) the block while a HostStage is alive -> VIS_E_STATE, block untouched
// (round 5's host SIGSEGV, gpurun_out/r5r_gdb.log: a stage that kept the address of a block a later vis_ensure_pin had freed; see
// csrc/vis_internal.h at HostStage).  Compiled by `make -C vi-slam_amd/csrc selftest` with hipcc as host code against the library;
// tests/test_abi.py runs it.  Prints one line per check and exits non-zero on the first failure.
#include "../csrc/vis_internal.h"
#include <cstdio>
#include <cstdlib>

static int fails = 0;
#define CHECK(cond) do { const bool ok_ = (cond); std::printf("%s  %s\n", ok_ ? "ok  " : "FAIL", #cond); if (!ok_) fails++; } while (0)

int main() {
    vis_ctx c;                                         // (default members only: no stream, no device)
    std::vector<char> block(1024);
    c.h_pin = block.data(); c.h_pin_bytes = block.size(); c.h_pin_dev = nullptr;
    {
        HostStage hs(&c);
        CHECK(c.stage_live == 1);
        void* a = hs.take(512);
        CHECK(a == block.data());
        void* b = hs.take(500);                        // 512 + 500 fits (offsets are 64-byte aligned: 512 is)
        CHECK(b == block.data() + 512);
        void* d = hs.take(64);                         // 1012 -> 1024 aligned + 64 > 1024
        CHECK(d == nullptr && hs.overflow);
        const unsigned long long waits = c.n_host_waits;
        CHECK(hs.wait() == VIS_E_NOMEM && c.n_host_waits == waits);      // refused before any device call
        // a block that must grow while the stage lives: refused, nothing freed, nothing allocated
        CHECK(vis_ensure_pin(&c, 4096) == VIS_E_STATE && c.h_pin == block.data() && c.h_pin_bytes == block.size());
        CHECK(vis_ensure_pin(&c, 1000) == VIS_OK);     // large enough already: no replacement, allowed under a live stage
    }
    CHECK(c.stage_live == 0);
    {
        vis_ctx e;                                     // no block at all
        HostStage hs(&e);
        CHECK(hs.take(4) == nullptr && hs.overflow && hs.wait() == VIS_E_NOMEM);
        char src[8] = {0};
        hs.up((void*)0x1000, src, 8);                  // an upload on an overflowing stage copies nothing and queues nothing
        CHECK(hs.up_n == 0 && e.n_copies == 0);
    }
    c.h_pin = nullptr; c.h_pin_bytes = 0;              // (the vector owns the memory)
    std::printf(fails ? "stage_selftest FAILED (%d)\n" : "stage_selftest passed\n", fails);
    return fails ? 1 : 0;
}
