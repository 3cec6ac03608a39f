// svi_api.hip — C-ABI glue of libsvi_hip.so: error plumbing and the operator-level entry points
// (the seams the reference exposes: flash_attention, LayerNorm+modulate, RMSNorm+RoPE, Linear, CFG step).
#include <algorithm>
#include <string>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <deque>
#include <mutex>
#include <vector>

#include "svi_common.h"

static thread_local char g_err[512] = "";

void svi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* svi_last_error(void) { return g_err; }

static int env_int(const char* name, int lo, int dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    char* end = nullptr;
    const long x = strtol(v, &end, 10);
    if (end == v || x < lo || x > (1 << 20)) { fprintf(stderr, "libsvi_hip: ignoring %s=%s (expected an integer >= %d)\n", name, v, lo); return dflt; }
    return (int)x;
}
static SviSwitches parse_switches() {
    SviSwitches s;
    s.flash_kernel = env_int("SVI_FLASH_KERNEL", 1, 0);
    if (s.flash_kernel > 2) s.flash_kernel = 0;
    s.gemm_kernel = env_int("SVI_GEMM_KERNEL", 128, 0);
#ifdef SVI_GEMM_EXPERIMENTS
    const bool experiment = s.gemm_kernel == 264 || s.gemm_kernel == 265;      // the four-wave tiles of round 6 (variant builds only)
#else
    const bool experiment = false;
#endif
    if (s.gemm_kernel != 128 && s.gemm_kernel != 192 && s.gemm_kernel != 259 && s.gemm_kernel != 260 && !experiment) {
        if (s.gemm_kernel != 0)      // 256 / 257 / 258 were kernels of rounds 1-3: an old A/B script must not silently measure "auto" instead
            fprintf(stderr, "libsvi_hip: ignoring SVI_GEMM_KERNEL=%d (kernels: 128, 192, 259 = 256^2 four phases, 260 = 256^2 two phases; 256 / 257 / 258 are retired)\n", s.gemm_kernel);
        s.gemm_kernel = 0;
    }
    s.gemm_gm = env_int("SVI_GEMM_GM", 1, 0);
    s.gemm_pf = env_int("SVI_GEMM_PF", 1, 0);
    if (s.gemm_pf != 0 && s.gemm_pf != 1 && s.gemm_pf != 4 && s.gemm_pf != 8) s.gemm_pf = 0;
    s.vae_exact_fp32 = getenv("SVI_VAE_EXACT_FP32") != nullptr;
    s.flash_two_pass = env_int("SVI_FLASH_TWO_PASS", 0, 1);
    s.flash_m16 = env_int("SVI_FLASH_M16", 0, 1);
    s.vae_no_x2h = env_int("SVI_VAE_X2H", 0, 1) == 0;
    s.cross_dedup = env_int("SVI_CROSS_DEDUP", 0, 1);
    s.cross_fused = env_int("SVI_CROSS_FUSED", 0, 1);
    s.vae_dma = env_int("SVI_VAE_DMA", 0, 1);
    s.vae_up_phases = env_int("SVI_VAE_UP_PHASES", 0, 1);
    s.vae_tile_order = env_int("SVI_VAE_TILE_ORDER", 0, 1);
    s.vae_pair = env_int("SVI_VAE_PAIR", 0, 1);
    s.mx8_fused = env_int("SVI_MX8_FUSED", 0, 1);
    s.qk_fused = env_int("SVI_QK_FUSED", 0, 1);
    s.attn_qk8 = env_int("SVI_ATTN_QK8", 0, 0) != 0;
    s.qk8_fused = env_int("SVI_QK8_FUSED", 0, 1);
    s.rms_rows = env_int("SVI_RMS_ROWS", 0, 1);
    s.flash_split = env_int("SVI_FLASH_SPLIT", 0, 0);
    if (s.flash_split > 4) s.flash_split = 4;
    s.ws_limit_mb = env_int("SVI_WS_LIMIT_MB", 1, 0);
    { const char* v = getenv("SVI_T5_BUCKETS"); s.t5_host_buckets = v && strcmp(v, "host") == 0; }
#ifdef SVI_ABLATIONS
    s.flash_abl = env_int("SVI_FLASH_ABL", 0, 0);
    s.gemm_epi_abl = env_int("SVI_GEMM_EPI_ABL", 0, 0);
    s.vae_abl = env_int("SVI_VAE_ABL", 0, 0);
    s.flash_assume_prescaled = getenv("SVI_FLASH_ASSUME_PRESCALED") != nullptr;
#endif
    return s;
}
// The parsed switches live behind an atomic pointer: a reload publishes a NEW immutable struct (the old ones are kept — a few hundred bytes per reload, A/B
// tooling only), so a launcher on another thread reads either the old set or the new one, never a half-written struct.
static std::atomic<const SviSwitches*>& switches_ptr() {
    static std::atomic<const SviSwitches*> p{new SviSwitches(parse_switches())};
    return p;
}
const SviSwitches& svi_switches() { return *switches_ptr().load(std::memory_order_acquire); }
static std::atomic<unsigned long long> g_stream_buffer_generation{0};      // moves whenever a per-stream library buffer is freed or the switches are re-read
extern "C" svi_status svi_switches_reload(void) {
    switches_ptr().store(new SviSwitches(parse_switches()), std::memory_order_release);
    // what a captured step graph has baked in includes the kernels the switches selected (and, with SVI_ATTN_QK8, the arithmetic): move the
    // generation svi_dit_generation reports so that every DenoiseLoop re-captures
    g_stream_buffer_generation.fetch_add(1, std::memory_order_relaxed);
    return SVI_OK;
}

// What the library PARSED for a switch (not what the environment says now): 1 / 0, or -1 for a name this query does not know.
extern "C" int32_t svi_switch_state(const char* name) {
    if (!name) return -1;
    const SviSwitches& sw = svi_switches();
    const std::string n(name);
    if (n == "SVI_ATTN_QK8") return sw.attn_qk8 ? 1 : 0;
    if (n == "SVI_CROSS_FUSED") return sw.cross_fused ? 1 : 0;
    if (n == "SVI_CROSS_DEDUP") return sw.cross_dedup ? 1 : 0;
    if (n == "SVI_QK_FUSED") return sw.qk_fused ? 1 : 0;
    if (n == "SVI_FLASH_TWO_PASS") return sw.flash_two_pass ? 1 : 0;
    if (n == "SVI_FLASH_M16") return sw.flash_m16 ? 1 : 0;
    return -1;
}

int svi_current_device() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); svi_set_error("no usable HIP device"); return -1; }
    return d;
}

svi_status svi_claim_device(int* handle_device) {
    const int d = svi_current_device();
    if (d < 0) return SVI_ERR_HIP;
    if (*handle_device < 0) *handle_device = d;
    if (*handle_device != d) {
        svi_set_error("handle belongs to device %d but the current device is %d (one handle per device)", *handle_device, d);
        return SVI_ERR_INVALID;
    }
    return SVI_OK;
}

// Process-wide tables (LDS attributes, per-stream buffers, profiler records) are shared by every handle and every host thread:
// one mutex guards them.  Handles themselves stay single-threaded (one handle per device, driven from one thread at a time).
static std::mutex& table_mutex() {
    static std::mutex m;
    return m;
}

svi_status svi_ensure_lds(const void* kernel, int bytes) {
    struct Seen { int dev; const void* fn; };
    static std::vector<Seen> seen;
    const int dev = svi_current_device();
    if (dev < 0) return SVI_ERR_HIP;
    std::lock_guard<std::mutex> lock(table_mutex());
    for (const Seen& s : seen)
        if (s.dev == dev && s.fn == kernel) return SVI_OK;
    SVI_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    seen.push_back(Seen{dev, kernel});
    return SVI_OK;
}

// Device buffers owned by the library outside any handle (the attention kernels' flag words, the operator seams' scratch) are
// keyed by (device, stream, kind): work enqueued on one stream is ordered, so a buffer is only ever used by one launch sequence
// at a time; two streams (or two host threads driving two streams) get buffers of their own and cannot clear or overwrite each
// other's.  Growing a buffer frees the old one (hipFree drains the device first) — never in steady state.
// A captured hipGraph bakes these buffers' addresses in: every time one of them is freed (grown, or released) the process-wide generation
// below moves, svi_dit_generation() includes it, and DenoiseLoop re-captures instead of replaying against freed memory.
namespace {
struct StreamSlot { int dev, kind; hipStream_t st; void* p; size_t bytes; long user; };
std::deque<StreamSlot>& stream_slots() {
    static std::deque<StreamSlot> slots;    // deque: a slot's address (its host-side `user` word) stays valid as the table grows
    return slots;
}
}  // namespace
unsigned long long svi_stream_buffer_generation() { return g_stream_buffer_generation.load(std::memory_order_relaxed); }

// Frees the library buffers keyed to `stream` on the current device (a capture stream that is being retired); stream = nullptr: those of
// every stream of the device.  Drains the device first (hipFree).  Graphs captured on such a stream must be dropped by the caller; the
// generation moves so that DenoiseLoop does.
extern "C" svi_status svi_stream_buffers_release(svi_stream stream, int32_t all_streams) {
    const int dev = svi_current_device();
    if (dev < 0) return SVI_ERR_HIP;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lock(table_mutex());
    bool any = false;
    for (StreamSlot& c : stream_slots())
        if (c.dev == dev && c.p && (all_streams || c.st == st)) {
            SVI_CHECK_HIP(hipFree(c.p));
            c.p = nullptr; c.bytes = 0; c.user = 0; any = true;
        }
    if (any) g_stream_buffer_generation.fetch_add(1, std::memory_order_relaxed);
    return SVI_OK;
}

svi_status svi_stream_buffer(int kind, hipStream_t st, size_t bytes, void** out, long** user_out) {
    typedef StreamSlot Slot;
    std::deque<Slot>& slots = stream_slots();
    const int dev = svi_current_device();
    if (dev < 0) return SVI_ERR_HIP;
    std::lock_guard<std::mutex> lock(table_mutex());
    Slot* s = nullptr;
    for (Slot& c : slots)
        if (c.dev == dev && c.kind == kind && c.st == st) { s = &c; break; }
    if (!s) {
        slots.push_back(Slot{dev, kind, st, nullptr, 0, 0});
        s = &slots.back();
    }
    if (s->bytes < bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            svi_set_error("a library buffer (kind %d, %zu B) would have to be allocated while its stream is being captured: run the call once on "
                          "the capture stream before capturing", kind, bytes);
            return SVI_ERR_INVALID;
        }
        (void)hipGetLastError();
        if (s->p) { SVI_CHECK_HIP(hipFree(s->p)); s->p = nullptr; s->bytes = 0; g_stream_buffer_generation.fetch_add(1, std::memory_order_relaxed); }
        hipError_t e = hipMalloc(&s->p, bytes);
        if (e != hipSuccess) { svi_set_error("hipMalloc(%zu B, buffer kind %d) failed: %s", bytes, kind, hipGetErrorString(e)); return SVI_ERR_OOM; }
        s->bytes = bytes;
    }
    *out = s->p;
    if (user_out) *user_out = &s->user;
    return SVI_OK;
}
extern "C" int32_t svi_abi_version(void) { return SVI_HIP_ABI_VERSION; }
extern "C" int32_t svi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// ---- event profiler (one recorder per process, guarded by the table mutex; tags are paired per (tag, stream)) ------------------
bool g_svi_prof_on = false;
unsigned g_svi_prof_mask = 0xFFFFFFFFu;
namespace {
struct ProfRec { int tag; hipEvent_t a, b; };
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
struct ProfOpen { int tag; hipStream_t st; hipEvent_t e; };
std::vector<ProfOpen> g_prof_open;
const char* kProfNames[PROF_NTAGS] = {"ln_modulate", "gemm_qkv", "rmsnorm_rope", "flash_self", "gemm_attn_out",
                                      "gemm_cross", "flash_cross", "gemm_ffn1", "gemm_ffn2", "embed", "head",
                                      "vae_conv", "vae_other"};
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    // timing-only events: no system-scope fence when the event completes — the default event's cache writeback / invalidation between
    // every two kernels of the block cost the step it was measuring 1-2 % (eager 437-440 ms against 431 ms replayed from a graph, which records none)
#ifdef SVI_PROF_FENCED_EVENTS      // A/B aid (tools/build_variant.py): the default events
    (void)hipEventCreate(&e);
    return e;
#endif
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) { (void)hipGetLastError(); (void)hipEventCreate(&e); }
    return e;
}
}  // namespace
// A stream that is being captured takes no event pairs (they would be recorded INTO the graph and read back as garbage): the scope is a
// no-op there — bench.py times graphed steps as wholes and takes the per-kernel figures from eager steps.
static bool prof_stream_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}
void svi_prof_begin_impl(int tag, hipStream_t st) {
    if (prof_stream_capturing(st)) return;
    std::lock_guard<std::mutex> lock(table_mutex());
    hipEvent_t e = prof_event();
    (void)hipEventRecord(e, st);
    g_prof_open.push_back(ProfOpen{tag, st, e});
}
void svi_prof_end_impl(int tag, hipStream_t st) {
    if (prof_stream_capturing(st)) return;
    std::lock_guard<std::mutex> lock(table_mutex());
    for (size_t i = g_prof_open.size(); i-- > 0;) {
        if (g_prof_open[i].tag != tag || g_prof_open[i].st != st) continue;
        hipEvent_t e = prof_event();
        (void)hipEventRecord(e, st);
        g_prof_recs.push_back(ProfRec{tag, g_prof_open[i].e, e});
        g_prof_open.erase(g_prof_open.begin() + (long)i);
        return;
    }
}
extern "C" svi_status svi_prof_enable(int32_t on) {
    std::lock_guard<std::mutex> lock(table_mutex());
    for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
    g_prof_recs.clear();
    for (auto& o : g_prof_open) g_prof_pool.push_back(o.e);
    g_prof_open.clear();
    g_svi_prof_on = on != 0;
    return SVI_OK;
}
// Which tags are recorded: a comma-separated list of tag names (as svi_prof_summary prints them), or NULL / "" for all.  Every event record is
// a packet between two kernels of the stream being measured; a caller that needs one kernel's durations over a long timed region (bench.py:
// the dominant kernel) selects that tag there and takes the full breakdown in a shorter pass.  Takes effect for launches after the call.
extern "C" svi_status svi_prof_select(const char* tags) {
    unsigned mask = 0;
    if (!tags || !*tags) mask = 0xFFFFFFFFu;
    else {
        const char* p = tags;
        while (*p) {
            const char* e = strchr(p, ',');
            const size_t n = e ? (size_t)(e - p) : strlen(p);
            int found = -1;
            for (int i = 0; i < PROF_NTAGS; ++i)
                if (strlen(kProfNames[i]) == n && strncmp(kProfNames[i], p, n) == 0) found = i;
            if (found < 0) { svi_set_error("svi_prof_select: unknown tag '%.*s'", (int)n, p); return SVI_ERR_INVALID; }
            mask |= 1u << found;
            p += n + (e ? 1 : 0);
        }
    }
    std::lock_guard<std::mutex> lock(table_mutex());
    g_svi_prof_mask = mask;
    return SVI_OK;
}
extern "C" svi_status svi_prof_summary(char* buf, int64_t buflen) {
    SVI_REQUIRE(buf && buflen > 64, "svi_prof_summary: buffer too small");
    std::lock_guard<std::mutex> lock(table_mutex());
    double ms[PROF_NTAGS] = {0};
    long cnt[PROF_NTAGS] = {0};
    for (auto& r : g_prof_recs) {
        SVI_CHECK_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        SVI_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
        ms[r.tag] += t; cnt[r.tag] += 1;
    }
    int64_t o = 0;
    o += snprintf(buf + o, buflen - o, "{");
    bool first = true;
    for (int i = 0; i < PROF_NTAGS; ++i) {
        if (!cnt[i]) continue;
        if (buflen - o < 96) break;
        o += snprintf(buf + o, buflen - o, "%s\"%s\": {\"count\": %ld, \"ms\": %.6f}", first ? "" : ", ", kProfNames[i], cnt[i], ms[i]);
        first = false;
    }
    snprintf(buf + o, buflen - o, "}");
    return SVI_OK;
}

// ---- scratch owned by the operator seams (grown on demand, never in steady state; one per device and stream) ----------------
namespace {
svi_status scratch_reserve(size_t bytes, hipStream_t st, char** out, int kind = SVI_BUF_SEAM_SCRATCH) {
    void* p = nullptr;
    SVI_TRY(svi_stream_buffer(kind, st, bytes, &p, nullptr));
    *out = reinterpret_cast<char*>(p);
    return SVI_OK;
}

__global__ void f32_prepare_kernel(const bf16* __restrict__ in, float* __restrict__ out, int n, int one_plus) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = (float)in[i];
    out[i] = one_plus ? rbf(1.0f + v) : v;
}
}  // namespace

extern "C" svi_status svi_attention_fwd(const void* q, const void* k, const void* v, void* out, int32_t b, int32_t s_q,
                                        int32_t s_kv, int32_t n, int32_t d, svi_stream stream) {
    SVI_REQUIRE(q && k && v && out, "svi_attention_fwd: null argument");
    SVI_REQUIRE(d == 128, "svi_attention_fwd: head dim %d unsupported (Wan DiT heads are 128 wide)", d);
    SVI_REQUIRE(b > 0 && s_q > 0 && s_kv > 0 && n > 0, "svi_attention_fwd: bad sizes");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int D = n * d;
    const int ldvt = ((s_kv + 7) / 8) * 8;
    char* scr = nullptr;
    SVI_TRY(scratch_reserve((size_t)D * ldvt * 2, st, &scr));
    bf16* vt = reinterpret_cast<bf16*>(scr);
    if (ldvt != s_kv) SVI_CHECK_HIP(hipMemsetAsync(vt, 0, (size_t)D * ldvt * 2, st));
    for (int i = 0; i < b; ++i) {
        const bf16* qi = reinterpret_cast<const bf16*>(q) + (size_t)i * s_q * D;
        const bf16* ki = reinterpret_cast<const bf16*>(k) + (size_t)i * s_kv * D;
        const bf16* vi = reinterpret_cast<const bf16*>(v) + (size_t)i * s_kv * D;
        bf16* oi = reinterpret_cast<bf16*>(out) + (size_t)i * s_q * D;
        SVI_TRY(svi_launch_transpose(vi, D, vt, ldvt, s_kv, D, st));
#ifdef SVI_ABLATIONS       // timing aid of tools/attn_abl.py (results wrong by the scale factor): variant builds only
        const int prescaled = svi_switches().flash_assume_prescaled;
#else
        const int prescaled = 0;
#endif
        SVI_TRY(svi_launch_flash(qi, D, ki, D, vt, ldvt, oi, D, s_q, s_kv, n, prescaled, st));
    }
    return SVI_OK;
}

// Attention on the layouts the DiT keeps internally: q / k token-major with row strides, V already transposed
// (vt [n*128, ldvt], zero beyond s_kv), q optionally pre-multiplied by softmax_scale * log2(e).  The sequence-parallel driver
// calls it on a head group after the token <-> head exchange.
extern "C" svi_status svi_attention_vt_fwd(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* vt, int32_t ldvt,
                                           void* out, int32_t ldo, int32_t s_q, int32_t s_kv, int32_t n, int32_t q_prescaled,
                                           svi_stream stream) {
    SVI_REQUIRE(q && k && vt && out, "svi_attention_vt_fwd: null argument");
    SVI_REQUIRE(s_q > 0 && s_kv > 0 && n > 0 && ldq >= n * 128 && ldk >= n * 128 && ldo >= n * 128 && ldvt >= s_kv,
                "svi_attention_vt_fwd: bad sizes");
    SVI_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "svi_attention_vt_fwd: strides must be multiples of 8 elements");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SviProfScope _p(PROF_FLASH_SELF, st);
    return svi_launch_flash(reinterpret_cast<const bf16*>(q), ldq, reinterpret_cast<const bf16*>(k), ldk, reinterpret_cast<const bf16*>(vt), ldvt,
                            reinterpret_cast<bf16*>(out), ldo, s_q, s_kv, n, q_prescaled ? 1 : 0, st);
}

extern "C" svi_status svi_video_to_u8(const float* video, uint8_t* frames, int32_t T, int32_t H, int32_t W, svi_stream stream) {
    SVI_REQUIRE(video && frames && T > 0 && H > 0 && W > 0, "svi_video_to_u8: bad argument");
    return svi_launch_video_to_u8(video, frames, (long)T * H * W, reinterpret_cast<hipStream_t>(stream));
}
extern "C" svi_status svi_u8_to_video(const uint8_t* frames, float* video, int32_t n, int32_t H, int32_t W, svi_stream stream) {
    SVI_REQUIRE(video && frames && n > 0 && H > 0 && W > 0, "svi_u8_to_video: bad argument");
    return svi_launch_u8_to_video(frames, video, n, (long)H * W, reinterpret_cast<hipStream_t>(stream));
}

extern "C" svi_status svi_layernorm_modulate(const void* x, void* out, int32_t rows, int32_t dim, float eps, const void* w,
                                             const void* b, const void* shift, const void* scale, svi_stream stream) {
    SVI_REQUIRE(x && out, "svi_layernorm_modulate: null argument");
    SVI_REQUIRE((shift == nullptr) == (scale == nullptr), "shift and scale must come together");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float *fs = nullptr, *fc = nullptr;
    if (shift) {
        char* scr = nullptr;
        SVI_TRY(scratch_reserve((size_t)2 * dim * 4, st, &scr, SVI_BUF_SEAM_SMALL));
        float* f = reinterpret_cast<float*>(scr);
        hipLaunchKernelGGL(f32_prepare_kernel, dim3((dim + 255) / 256), dim3(256), 0, st, reinterpret_cast<const bf16*>(shift), f, dim, 0);
        hipLaunchKernelGGL(f32_prepare_kernel, dim3((dim + 255) / 256), dim3(256), 0, st, reinterpret_cast<const bf16*>(scale), f + dim, dim, 1);
        SVI_LAUNCH_CHECK();
        fs = f; fc = f + dim;
    }
    return svi_launch_ln_mod(reinterpret_cast<const bf16*>(x), dim, reinterpret_cast<bf16*>(out), dim, rows, dim, eps,
                             reinterpret_cast<const bf16*>(w), reinterpret_cast<const bf16*>(b), fs, fc, st);
}

extern "C" svi_status svi_rmsnorm_rope(void* x, int32_t ld, int32_t rows, int32_t dim, const void* weight, float eps,
                                       int32_t rope, int32_t num_heads, int32_t f, int32_t h, int32_t w, svi_stream stream) {
    SVI_REQUIRE(x && weight, "svi_rmsnorm_rope: null argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!rope)
        return svi_launch_rmsnorm_rope(reinterpret_cast<bf16*>(x), ld, rows, dim, reinterpret_cast<const bf16*>(weight), eps, nullptr, 1.0f, st);
    SVI_REQUIRE(num_heads > 0 && dim == num_heads * 128, "rope needs head_dim 128");
    SVI_REQUIRE(f > 0 && h > 0 && w > 0 && f * h * w == rows, "rope grid %dx%dx%d != rows %d", f, h, w, rows);
    // table built on the host in fp64 exactly as precompute_freqs_cis_3d does (dit:161-175)
    const int dh = 128, d_hw = dh / 3, d_f = dh - 2 * d_hw;
    const int npf = d_f / 2, nph = d_hw / 2, npw = d_hw / 2;
    const size_t cnt = (size_t)f * npf + (size_t)h * nph + (size_t)w * npw;
    float2* host = (float2*)malloc(cnt * sizeof(float2));
    if (!host) { svi_set_error("out of host memory"); return SVI_ERR_OOM; }
    size_t o = 0;
    const int lens[3] = {f, h, w}, dims[3] = {d_f, d_hw, d_hw}, nps[3] = {npf, nph, npw};
    for (int a = 0; a < 3; ++a)
        for (int p = 0; p < lens[a]; ++p)
            for (int i = 0; i < nps[a]; ++i) {
                const double ang = (double)p / pow(10000.0, (double)(2 * i) / (double)dims[a]);
                host[o++] = make_float2((float)cos(ang), (float)sin(ang));
            }
    char* scr = nullptr;
    svi_status s = scratch_reserve(cnt * sizeof(float2), st, &scr, SVI_BUF_SEAM_SMALL);
    if (s != SVI_OK) { free(host); return s; }
    // the scratch may still be read by work enqueued earlier (a previous call's table): drain the stream before rewriting it,
    // and the device after, since `st` need not be ordered after the null stream the copy runs on (operator seam, not the hot path)
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(scr, host, cnt * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    free(host);
    if (e != hipSuccess) { svi_set_error("hipMemcpy(rope table) failed: %s", hipGetErrorString(e)); return SVI_ERR_HIP; }
    SviRope r{};
    r.tab_f = reinterpret_cast<const float2*>(scr);
    r.tab_h = r.tab_f + (size_t)f * npf;
    r.tab_w = r.tab_h + (size_t)h * nph;
    r.npf = npf; r.nph = nph; r.npw = npw; r.f = f; r.h = h; r.w = w;
    return svi_launch_rmsnorm_rope(reinterpret_cast<bf16*>(x), ld, rows, dim, reinterpret_cast<const bf16*>(weight), eps, &r, 1.0f, st);
}

extern "C" svi_status svi_gemm_bf16(const void* A, int32_t lda, const void* W, int32_t ldw, void* C, int32_t ldc, int32_t M,
                                    int32_t N, int32_t K, const void* bias, int32_t bias_along_m, int32_t epilogue,
                                    const float* gate, const void* res, int32_t ldres, svi_stream stream) {
    SVI_REQUIRE(A && W && C, "svi_gemm_bf16: null argument");
    SviGemmArgs g{};
    g.A = reinterpret_cast<const bf16*>(A); g.lda = lda; g.W = reinterpret_cast<const bf16*>(W); g.ldw = ldw;
    g.C = reinterpret_cast<bf16*>(C); g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = reinterpret_cast<const bf16*>(bias); g.bias_along_m = bias_along_m; g.epi = epilogue; g.gate = gate;
    g.res = reinterpret_cast<const bf16*>(res); g.ldres = ldres;
    return svi_launch_gemm(g, reinterpret_cast<hipStream_t>(stream));
}

// The cross-attention query path as the DiT block runs it, in two seams (CrossAttention.forward, models/wan_video_dit.py:266-303: q = norm_q(self.q(x)), then
// attention against the prompt's K / V): the projection that also leaves the RMSNorm statistic, and the short-key attention that normalises q as it reads it.
extern "C" svi_status svi_linear_row_stats(const void* A, int32_t lda, const void* W, int32_t ldw, void* C, int32_t ldc, int32_t M, int32_t N, int32_t K,
                                           const void* bias, float eps, float* row_sumsq, int32_t ldss, float* rs_out, svi_stream stream) {
    SVI_REQUIRE(A && W && C && row_sumsq && rs_out, "svi_linear_row_stats: null argument");
    SVI_REQUIRE(N > 0 && N % 64 == 0 && ldss >= M, "svi_linear_row_stats: N = %d must be a multiple of 64 and ldss >= M", N);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SviGemmArgs g{};
    g.A = reinterpret_cast<const bf16*>(A); g.lda = lda; g.W = reinterpret_cast<const bf16*>(W); g.ldw = ldw;
    g.C = reinterpret_cast<bf16*>(C); g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = reinterpret_cast<const bf16*>(bias); g.epi = SVI_EPI_BIAS;
    g.rowss = row_sumsq; g.ldss = ldss;
    SVI_TRY(svi_launch_gemm(g, st));
    return svi_launch_row_rs(row_sumsq, N / 64, ldss, M, N, eps, rs_out, st);
}
extern "C" svi_status svi_cross_attention_fwd(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* vt, int32_t ldvt, void* out, int32_t ldo,
                                              int32_t s_q, int32_t s_kv, int32_t n, const int32_t* key_tail, const float* q_rs, const void* q_gain,
                                              float q_out_scale, svi_stream stream) {
    SVI_REQUIRE(q && k && vt && out, "svi_cross_attention_fwd: null argument");
    SVI_REQUIRE(s_q > 0 && s_kv > 0 && n > 0 && ldq >= n * 128 && ldk >= n * 128 && ldo >= n * 128 && ldvt >= s_kv, "svi_cross_attention_fwd: bad sizes");
    SVI_REQUIRE((q_rs == nullptr) == (q_gain == nullptr), "svi_cross_attention_fwd: q_rs and q_gain come together");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SviProfScope _p(PROF_FLASH_CROSS, st);
    const SviQNorm qn{q_rs, reinterpret_cast<const bf16*>(q_gain), q_out_scale};
    int key_blocks = 0;          // the host's copy of how many 32-key blocks are walked (the DiT reads it once per prompt; here per call, outside a capture)
    if (key_tail) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        int n = 0;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone &&
            hipMemcpyAsync(&n, key_tail, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess && n > 0)
            key_blocks = (std::min(n, s_kv) + 31) / 32;
        else (void)hipGetLastError();
    }
    return svi_launch_flash_cross(reinterpret_cast<const bf16*>(q), ldq, reinterpret_cast<const bf16*>(k), ldk, reinterpret_cast<const bf16*>(vt), ldvt,
                                  reinterpret_cast<bf16*>(out), ldo, s_q, s_kv, n, st, key_tail, q_rs ? &qn : nullptr, key_blocks);
}

// Launch planners (no device work): what the GEMM / attention launchers would do with a problem of these sizes under the current switches.
extern "C" svi_status svi_gemm_plan(int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t skinny, int32_t compute_units, int32_t* kernel_out) {
    SVI_REQUIRE(kernel_out && M > 0 && N > 0 && K > 0 && compute_units > 0, "svi_gemm_plan: bad argument");
    SviGemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = (N + 7) / 8 * 8; g.epi = epilogue; g.skinny = skinny;
    *kernel_out = svi_gemm_choose(g, compute_units);
    return SVI_OK;
}
extern "C" svi_status svi_attention_plan(int32_t s_q, int32_t s_kv, int32_t heads, int32_t compute_units, int32_t* out4) {
    SVI_REQUIRE(out4 && s_q > 0 && s_kv > 0 && heads > 0 && compute_units > 0, "svi_attention_plan: bad argument");
    int kernel = 0;
    const SviFlashSplit sp = svi_flash_plan(s_q, s_kv, heads, compute_units, &kernel);
    const int items = sp.qblocks * sp.heads;
    out4[0] = kernel; out4[1] = sp.whole; out4[2] = sp.pieces; out4[3] = sp.whole + (items - sp.whole) * sp.pieces;
    return SVI_OK;
}

// MX-fp8 operator seams (opt-in path; see csrc/svi_gemm.hip): quantise bf16 activations, and the block-scaled GEMM itself.
extern "C" svi_status svi_mx8_quantize(const void* x, int32_t ldx, int32_t rows, int32_t K, void* q, int32_t ldq, void* scales, int32_t sc_rows,
                                       svi_stream stream) {
    SVI_REQUIRE(x && q && scales, "svi_mx8_quantize: null argument");
    return svi_launch_mx8_quantize(reinterpret_cast<const bf16*>(x), ldx, rows, K, reinterpret_cast<unsigned char*>(q), ldq, reinterpret_cast<unsigned*>(scales), sc_rows,
                                   reinterpret_cast<hipStream_t>(stream));
}
extern "C" svi_status svi_gemm_mx8(const void* A8, int32_t lda, const void* a_scales, int32_t sc_rows, const void* W8, int32_t ldw, void* C, int32_t ldc,
                                   int32_t M, int32_t N, int32_t K, const void* bias, int32_t epilogue, const float* gate, const void* res, int32_t ldres,
                                   svi_stream stream) {
    SVI_REQUIRE(A8 && a_scales && W8 && C, "svi_gemm_mx8: null argument");
    SviGemmArgs g{};
    g.A = reinterpret_cast<const bf16*>(A8); g.lda = lda; g.W = reinterpret_cast<const bf16*>(W8); g.ldw = ldw;
    g.C = reinterpret_cast<bf16*>(C); g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = reinterpret_cast<const bf16*>(bias); g.epi = epilogue; g.gate = gate; g.res = reinterpret_cast<const bf16*>(res); g.ldres = ldres;
    return svi_launch_gemm_mx8(g, reinterpret_cast<const unsigned*>(a_scales), sc_rows, reinterpret_cast<hipStream_t>(stream));
}

extern "C" svi_status svi_gemm_mx8_wscaled(const void* A8, int32_t lda, const void* W8, int32_t ldw, const void* w_scales, int32_t sc_rows, void* C, int32_t ldc,
                                           int32_t M, int32_t N, int32_t K, const void* bias, int32_t bias_along_m, svi_stream stream) {
    SVI_REQUIRE(A8 && w_scales && W8 && C, "svi_gemm_mx8_wscaled: null argument");
    SviGemmArgs g{};
    g.A = reinterpret_cast<const bf16*>(A8); g.lda = lda; g.W = reinterpret_cast<const bf16*>(W8); g.ldw = ldw;
    g.C = reinterpret_cast<bf16*>(C); g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = reinterpret_cast<const bf16*>(bias); g.bias_along_m = bias_along_m; g.epi = SVI_EPI_BIAS;
    return svi_launch_gemm_mx8_wscaled(g, reinterpret_cast<const unsigned*>(w_scales), sc_rows, reinterpret_cast<hipStream_t>(stream));
}

extern "C" svi_status svi_cfg_step(void* latents, const void* cond, const void* uncond, int64_t n, float cfg_scale,
                                   float dsigma, svi_stream stream) {
    SVI_REQUIRE(latents && cond, "svi_cfg_step: null argument");
    return svi_launch_cfg_step(reinterpret_cast<bf16*>(latents), reinterpret_cast<const bf16*>(cond),
                               reinterpret_cast<const bf16*>(uncond), n, cfg_scale, dsigma,
                               reinterpret_cast<hipStream_t>(stream));
}

extern "C" svi_status svi_fp8_e4m3_to_bf16(const void* in, void* out, int64_t n, svi_stream stream) {
    SVI_REQUIRE(in && out && n >= 0, "svi_fp8_e4m3_to_bf16: bad argument");
    return svi_launch_fp8_e4m3_to_bf16(reinterpret_cast<const unsigned char*>(in), reinterpret_cast<bf16*>(out), n, reinterpret_cast<hipStream_t>(stream));
}

extern "C" svi_status svi_cfg3_step(void* latents, const void* cond, const void* uncond, const void* drop_text, int64_t n, float s_text,
                                    float s_audio, float dsigma, svi_stream stream) {
    SVI_REQUIRE(latents && cond && uncond && drop_text && n >= 0, "svi_cfg3_step: bad argument");
    return svi_launch_cfg3_step(reinterpret_cast<bf16*>(latents), reinterpret_cast<const bf16*>(cond), reinterpret_cast<const bf16*>(uncond),
                                reinterpret_cast<const bf16*>(drop_text), n, s_text, s_audio, dsigma, reinterpret_cast<hipStream_t>(stream));
}
