// Library state: device binding, per-thread error text, scratch arenas.
#include "xmi_common.h"

#include <mutex>
#include <string>

namespace xmi {

static thread_local char g_err[512] = "";
static std::mutex g_mu;
static int g_device = -1;
static constexpr int kSlots = 12;
static void *g_scratch[kSlots] = {};
static size_t g_scratch_bytes[kSlots] = {};

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static int init_locked(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible: libx265_amd has no CPU fallback");
        return XMI_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (0..%d)", device, n - 1);
        return XMI_ERR_ARG;
    }
    hipDeviceProp_t prop;
    XMI_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return XMI_ERR_NO_DEVICE;
    }
    XMI_HIP(hipSetDevice(device));
    int rc = sad_surface_prepare_device();
    if (rc != XMI_OK) return rc;
    rc = deblock_prepare_device(device);
    if (rc != XMI_OK) return rc;
    rc = rdoq_prepare_device();
    if (rc != XMI_OK) return rc;
    g_device = device;
    return XMI_OK;
}

// Implicit binding (no xmi_init before the first call): the calling thread's current device - HIP's own default, device 0
// unless the caller chose another one - so a host that only links the drop-in works on a box with several GPUs as any HIP
// program does.
static int bind_implicitly_locked()
{
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) cur = 0;
    return init_locked(cur);
}

// Entry points that take the caller's device pointers and stream: the library never switches such a caller's device behind
// its back. A thread whose current device is another one than the bound one holds pointers of the wrong device and is told so.
int require_device()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_device < 0) return bind_implicitly_locked();
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) {
        set_error("hipGetDevice failed");
        return XMI_ERR_HIP;
    }
    if (cur != g_device) {
        set_error("the calling thread's current device is %d but the library is bound to device %d: make it current "
                  "(hipSetDevice / torch.cuda.set_device) before calling",
                  cur, g_device);
        return XMI_ERR_ARG;
    }
    return XMI_OK;
}

// Entry points that take host pointers or one of the library's handles (per-call primitives, frame session, analysis,
// encoder): everything they touch on the GPU is the library's own, on the bound device, so the bound device is made current
// for the calling thread - a fresh host thread starts on device 0 whatever the process was bound to, and the worker threads of
// the handles do the same for themselves.
int enter_device()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_device < 0) return bind_implicitly_locked();
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == g_device) return XMI_OK;
    if (hipSetDevice(g_device) != hipSuccess) {
        set_error("hipSetDevice(%d) failed", g_device);
        return XMI_ERR_HIP;
    }
    return XMI_OK;
}

int bound_device()
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_device;
}

std::mutex &picture_call_mutex()
{
    static std::mutex m;
    return m;
}

void *scratch(int slot, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot < 0 || slot >= kSlots) return nullptr;
    if (g_scratch_bytes[slot] >= bytes && g_scratch[slot]) return g_scratch[slot];
    if (g_scratch[slot]) (void)hipFree(g_scratch[slot]);
    g_scratch[slot] = nullptr;
    g_scratch_bytes[slot] = 0;
    size_t want = bytes < 4096 ? 4096 : bytes;
    if (hipMalloc(&g_scratch[slot], want) != hipSuccess) {
        set_error("hipMalloc(%zu) failed", want);
        return nullptr;
    }
    g_scratch_bytes[slot] = want;
    return g_scratch[slot];
}

}  // namespace xmi

extern "C" {

int xmi_init(int device)
{
    std::lock_guard<std::mutex> lk(xmi::g_mu);
    if (xmi::g_device == device) return XMI_OK;
    if (xmi::g_device >= 0) {
        xmi::set_error("already bound to device %d; call xmi_shutdown() first", xmi::g_device);
        return XMI_ERR_ARG;
    }
    return xmi::init_locked(device);
}

void xmi_shutdown(void)
{
    std::lock_guard<std::mutex> lk(xmi::g_mu);
    for (int i = 0; i < xmi::kSlots; i++) {
        if (xmi::g_scratch[i]) (void)hipFree(xmi::g_scratch[i]);
        xmi::g_scratch[i] = nullptr;
        xmi::g_scratch_bytes[i] = 0;
    }
    xmi::g_device = -1;
}

const char *xmi_last_error(void) { return xmi::g_err; }

int xmi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *xmi_version(void) { return "x265_amd 0.1 (gfx950)"; }

}  // extern "C"
