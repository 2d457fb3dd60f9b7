// Test infrastructure (never loaded by the product): an "electric fence" device allocator for the -m gpu tests.
// torch.cuda.memory.CUDAPluggableAllocator entry points (efence_malloc / efence_free): every tensor is its own hipMalloc of a multiple
// of 2 MiB and ENDS where that allocation ends.  A kernel that reads or writes past the end of ANY tensor -- program buffers, packed
// weights, the tensors a test hands to the C ABI -- then leaves the mapped range and raises a memory fault on every run (unless the
// driver happened to place another allocation right behind it: the fence can miss, it never cries wolf), instead of reading whatever
// lies behind the tensor in the caching allocator's segment.  New memory is filled with 0xFF bytes (NaN in fp16 and fp32): reads of
// never-written memory that reach a result show up as NaN.  Slow by design (hipMalloc / hipFree and a device synchronisation per
// tensor); enabled with EOD_TEST_EFENCE=1 (tests/conftest.py).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

namespace {
std::mutex g_mu;
std::unordered_map<void*, void*> g_base;  // user pointer -> hipMalloc'd base
constexpr size_t GRAN = (size_t)2 << 20;

void die(const char* what, hipError_t e) {
    std::fprintf(stderr, "efence_alloc: %s failed: %s\n", what, hipGetErrorString(e));
    std::abort();
}
#define EF_CHECK(call)                        \
    do {                                      \
        hipError_t e_ = (call);               \
        if (e_ != hipSuccess) die(#call, e_); \
    } while (0)
}  // namespace

extern "C" void* efence_malloc(ssize_t size, int device, hipStream_t stream) {
    (void)stream;
    if (size <= 0) return nullptr;
    EF_CHECK(hipSetDevice(device));
    const size_t user = ((size_t)size + 15) & ~(size_t)15;
    const size_t total = (user + GRAN - 1) / GRAN * GRAN;
    void* base = nullptr;
    EF_CHECK(hipMalloc(&base, total));
    EF_CHECK(hipMemset(base, 0xFF, total));
    EF_CHECK(hipDeviceSynchronize());  // (the fill is complete before anybody, on any stream, touches the tensor)
    void* p = (char*)base + (total - user);
    std::lock_guard<std::mutex> lk(g_mu);
    g_base[p] = base;
    return p;
}

extern "C" void efence_free(void* p, ssize_t size, int device, hipStream_t stream) {
    (void)size;
    (void)stream;
    if (!p) return;
    EF_CHECK(hipSetDevice(device));
    EF_CHECK(hipDeviceSynchronize());  // (a kernel still using the tensor must not find it unmapped)
    void* base = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_base.find(p);
        if (it == g_base.end()) {
            std::fprintf(stderr, "efence_alloc: free of an unknown pointer %p\n", p);
            std::abort();
        }
        base = it->second;
        g_base.erase(it);
    }
    EF_CHECK(hipFree(base));
}
