// guard_alloc.cpp -- a guard-band device allocator for PyTorch-ROCm (TEST INFRASTRUCTURE, never loaded by the product).
//
// one 8-rank rehearsal died with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION inside somebody else's
// kernel, which is what an out-of-bounds device write looks like, and the same round found a real out-of-bounds scratch
// write in gnr_wgrad.hip outside every tested size.  The caching allocator hides such writes: neighbouring tensors share
// one segment, an overrun lands in live data or in a free block and nothing complains.
//
// This library is plugged into torch with torch.cuda.memory.CUDAPluggableAllocator (tests/guard/run_guarded.py does it,
// before the first device allocation).  From then on EVERY device allocation of the process -- the workspaces, scratch
// buffers and outputs the bindings hand to gnr_fwd / gnr_bwd / gnr_upsample_* / gnr_merge_* / gnr_resample, the saved
// activations, the gradients, torch's own temporaries -- is its own hipMalloc with a guard band of GUARD bytes (default
// 1 MiB, GNR_GUARD_BYTES) of a known pattern on BOTH sides, and optionally a poisoned body (GNR_GUARD_POISON=1: every byte
// 0xFF = a NaN in fp32, so a kernel that reads workspace it never wrote shows up as NaN in results that are then compared
// with the oracle).  The bands are checked when the allocation is freed and whenever guard_check_all() is called (the
// harness calls it after every library call, so a violation names the call that made it).
//
// Build: hipcc -shared -fPIC -O2 tests/guard/guard_alloc.cpp -o tests/guard/_guard_alloc.so   (gazenerf_amd.build.build_guard)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr unsigned char PATTERN = 0xA5;

struct Block {
    size_t size;     // bytes the caller asked for
    size_t guard;    // bytes of guard on each side
    int device;
    uint64_t serial;
};

// never destructed: torch frees its last tensors during interpreter shutdown, after this library's static destructors
std::mutex& mu = *new std::mutex;
std::map<void*, Block>& live = *new std::map<void*, Block>;           // user pointer -> block
uint64_t n_alloc = 0, n_free = 0, n_violations = 0, n_checks = 0;
size_t bytes_live = 0, bytes_peak = 0;
std::string& report = *new std::string;                    // one line per violation
std::vector<unsigned char>& host = *new std::vector<unsigned char>;       // staging for the read-back

size_t guard_bytes() {
    static size_t g = [] {
        const char* e = getenv("GNR_GUARD_BYTES");
        size_t v = e ? strtoull(e, nullptr, 10) : (size_t)1 << 20;
        return (v + 255) & ~(size_t)255;          // keep the user pointer 256-byte aligned like the caching allocator
    }();
    return g;
}

bool poison() {
    static bool p = [] { const char* e = getenv("GNR_GUARD_POISON"); return e && e[0] == '1'; }();
    return p;
}

void fail(const char* what, hipError_t e) {
    fprintf(stderr, "guard_alloc: %s failed: %s\n", what, hipGetErrorString(e));
    fflush(stderr);
}

// caller holds `mu` and has synchronised the device.  Reads both bands back and records every band that was written to.
void check_block(void* user, const Block& b, const char* when) {
    const size_t g = b.guard;
    if (host.size() < g) host.resize(g);
    char* base = (char*)user - g;
    for (int side = 0; side < 2; ++side) {
        const char* src = side == 0 ? base : (char*)user + b.size;
        hipError_t e = hipMemcpy(host.data(), src, g, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { fail("hipMemcpy (guard read-back)", e); continue; }
        size_t bad = 0, first = 0, last = 0;
        for (size_t i = 0; i < g; ++i)
            if (host[i] != PATTERN) {
                if (!bad) first = i;
                last = i;
                ++bad;
            }
        if (bad) {
            ++n_violations;
            char line[512];
            // offsets relative to the user buffer: negative = before its first byte, >= size = past its end
            const long long lo = side == 0 ? (long long)first - (long long)g : (long long)(b.size + first);
            const long long hi = side == 0 ? (long long)last - (long long)g : (long long)(b.size + last);
            unsigned int word = 0;
            memcpy(&word, host.data() + (first & ~(size_t)3), 4);
            snprintf(line, sizeof line,
                     "VIOLATION %s: allocation #%llu (%zu bytes, device %d): %zu byte(s) of the %s guard overwritten, offsets %lld..%lld "
                     "relative to the buffer (first overwritten word 0x%08x)\n",
                     when, (unsigned long long)b.serial, b.size, b.device, bad, side == 0 ? "LEADING" : "TRAILING", lo, hi, word);
            report += line;
            fputs("guard_alloc: ", stderr);
            fputs(line, stderr);
            fflush(stderr);
            // restore the pattern so the same overrun is reported once, not at every later check
            hipMemset((void*)src, PATTERN, g);
        }
    }
    ++n_checks;
}

}  // namespace

extern "C" {

void* guard_malloc(ssize_t size, int device, hipStream_t stream) {
    (void)stream;
    if (size <= 0) size = 1;
    const size_t g = guard_bytes();
    int prev = 0;
    hipGetDevice(&prev);
    if (prev != device) hipSetDevice(device);
    char* base = nullptr;
    hipError_t e = hipMalloc((void**)&base, (size_t)size + 2 * g);
    if (e != hipSuccess) {
        fail("hipMalloc", e);
        if (prev != device) hipSetDevice(prev);
        return nullptr;
    }
    // the trailing band begins at the first byte past the request: a one-element overrun is seen, not absorbed by padding
    hipMemset(base, PATTERN, g);
    hipMemset(base + g + (size_t)size, PATTERN, g);
    if (poison()) hipMemset(base + g, 0xFF, (size_t)size);
    hipDeviceSynchronize();
    if (prev != device) hipSetDevice(prev);
    std::lock_guard<std::mutex> lk(mu);
    live[base + g] = Block{(size_t)size, g, device, ++n_alloc};
    bytes_live += (size_t)size;
    if (bytes_live > bytes_peak) bytes_peak = bytes_live;
    return base + g;
}

void guard_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
    (void)size; (void)stream;
    if (!ptr) return;
    int prev = 0;
    hipGetDevice(&prev);
    if (prev != device) hipSetDevice(device);
    hipDeviceSynchronize();            // torch frees a tensor while kernels that use it may still be queued: no caching here
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(ptr);
        if (it == live.end()) {
            fprintf(stderr, "guard_alloc: free of an unknown pointer %p\n", ptr);
        } else {
            check_block(ptr, it->second, "at free");
            bytes_live -= it->second.size;
            hipFree((char*)ptr - it->second.guard);
            live.erase(it);
            ++n_free;
        }
    }
    if (prev != device) hipSetDevice(prev);
}

// Synchronise and check the bands of every live allocation.  -> number of violations recorded so far (all checks).
unsigned long long guard_check_all(const char* when) {
    hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : live) {
        int prev = 0;
        hipGetDevice(&prev);
        if (prev != kv.second.device) hipSetDevice(kv.second.device);
        check_block(kv.first, kv.second, when ? when : "check");
        if (prev != kv.second.device) hipSetDevice(prev);
    }
    return n_violations;
}

unsigned long long guard_violations(void) { std::lock_guard<std::mutex> lk(mu); return n_violations; }

// {allocations, frees, live, band checks, peak live bytes, guard bytes per side}
void guard_stats(unsigned long long* out6) {
    std::lock_guard<std::mutex> lk(mu);
    out6[0] = n_alloc; out6[1] = n_free; out6[2] = live.size(); out6[3] = n_checks; out6[4] = bytes_peak; out6[5] = guard_bytes();
}

// Copies the violation lines (NUL-terminated, truncated to n) -> bytes needed.
size_t guard_report(char* buf, size_t n) {
    std::lock_guard<std::mutex> lk(mu);
    if (buf && n) {
        size_t k = report.size() < n - 1 ? report.size() : n - 1;
        memcpy(buf, report.data(), k);
        buf[k] = 0;
    }
    return report.size() + 1;
}

}  // extern "C"
