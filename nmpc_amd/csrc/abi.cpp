// Host side of libcnmpc.so: batch objects, the hidden single-instance
// controller behind the reference's symbols, and the extern "C" boundary
// declared in include/cnmpc.h.  All arithmetic is in kernels.hip; this file
// only owns device memory, ring offsets, staging copies and launch order.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cnmpc.h"
#include "kernels.hpp"

using namespace nmpc;

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess)                                                      \
            return fail(NMPC_E_HIP, "%s failed: %s (%s:%d)", #expr,                \
                        hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)

#define TRY(expr)                 \
    do {                          \
        int r_ = (expr);          \
        if (r_ != 0) return r_;   \
    } while (0)

bool is_device_pointer(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// A diagonal weight block within 1e-15 of the identity carries no data in the
// reference's QP solver and behaves as exact ones
// (ccs-c66x/qpDUNES/utils.c:106-117, setup_qp.c:229-258).
bool is_identity(const real_t* w, int n) {
    for (int i = 0; i < n; i++)
        if (std::fabs(w[i] - 1.0) > 1e-15) return false;
    return true;
}

struct Staging {        // grow-only device scratch for host-side inputs/outputs
    void* ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (ptr) HIP_TRY(hipFree(ptr));
        ptr = nullptr; cap = 0;
        HIP_TRY(hipMalloc(&ptr, bytes));
        cap = bytes;
        return 0;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

// Small host-memory inputs (a measurement, a reference point, a wind vector of a few
// instances) are copied into a slot of page-locked memory first and go to the device
// from there: the caller's buffer is free when the call returns, so the call need not
// wait for the stream, and a sequence of such calls (a control step through the
// reference's interface is four of them) waits once, where results are read.
// A slot is reused after kPinnedSlots further inputs, once the copy out of it is done.
constexpr size_t kPinnedSlotBytes = 4096;
constexpr unsigned kPinnedSlots = 64;
struct PinnedRing {
    char* base = nullptr;
    hipEvent_t done[kPinnedSlots] = {};
    unsigned next = 0;
    bool failed = false;
};

// The entry point that allocates nothing (nmpc_batch_set_reference_from_waypoints_noalloc)
// keeps its scratch in one fixed array per GPU, shared by the batches on that GPU
// whatever their streams: a call records this event behind its last reader and
// waits for it before its first writer, so that calls which follow each other in
// host order do not overlap on the device.  Made by nmpc_batch_create (creating an
// event is no business of a call that must allocate nothing).
constexpr int kMaxDevices = 64;
hipEvent_t g_scratch_free[kMaxDevices];

// start may be the stop event of the launch before (owned == false): launches
// queued back to back share the event between them.
struct TimedLaunch { int kind; hipEvent_t start, stop; bool owned; };

}  // namespace

struct nmpc_batch {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    BatchView v{};
    void* arena = nullptr;
    size_t arena_bytes = 0;
    size_t reset_bytes = 0;     // leading part of the arena that nmpc_batch_reset clears
    real_t wq[12], wr[3], wt[12], lb[3], ub[3];
    real_t init_lb[3], init_ub[3];      // control bounds as nmpc_batch_init saw them
    bool initialised = false;
    // Real-time-iteration preparation (nmpc_batch_set_rti): fresh[k] != 0 while
    // horizon position k has been linearised about the reference since the last
    // feedback step, i.e. carries no solution to linearise about.
    bool rti = false;
    std::vector<char> fresh;
    Staging in_refs, in_meas, in_wind, out_u, out_status, out_iters, scratch;
    PinnedRing pinned;
    // nmpc_batch_init leaves the Jacobians C and residuals c logically zero; the
    // clear itself is postponed until something is about to read or partly
    // overwrite them, and dropped when every stage is linearised first.
    bool jacobians_zero_pending = false;
    // nmpc_batch_init also re-stages every stage QP; a stage's linearisation
    // rewrites all of that, so the stages 0..N-1 are left for later under the
    // same rule (k_init_stages part 2).
    bool stage_init_pending = false;
    // Likewise for nmpc_batch_reset: the stage vectors zLow .. y are logically
    // zero after it, but nmpc_batch_init rewrites six of the nine, the first
    // feedback step rewrites z and y, and the trial-point scratch is never
    // looked at; clears happen only where an earlier reader turns up.
    bool stage_zero_pending = false;    // zLow, zUpp, q, z, qStep, zUnc, dz, tmp, y
    bool zy_zero_pending = false;       // z, y only (after nmpc_batch_init)
    // [B][kCfgStride] weights and bounds of every instance, allocated by the first
    // per-instance setter; until then every instance has the batch's (v.cfg null)
    real_t* cfg = nullptr;
    Staging in_cfg;
    real_t* plant = nullptr;    // [B][13] simulated vehicle states (nmpc_batch_plant_*)
    real_t* plant_next = nullptr;   // output of the next roll-forward step, then swapped
    bool profiling = false;
    unsigned profile_kinds = ~0u;   // bit k: launches of kernel id k are timed
    std::vector<TimedLaunch> pending;
    std::vector<hipEvent_t> free_events;
    bool untimed = false;           // inside an entry point that must not create events
    unsigned solve_lists = 0;       // instance lists made for the Newton solve so far
    // A Newton iteration of a mid-sized batch runs as two parts side by side: the
    // instances with long factorisation sweeps on the batch's stream, the others
    // on this one, between a fork and a join event.
    hipStream_t side_stream = nullptr;
    hipEvent_t fork_event = nullptr, join_event = nullptr;
    bool chain_launches = false;    // inside a sequence of launches with nothing between them
    hipEvent_t chain_event = nullptr;
    double k_ms[NMPC_K_COUNT] = {0};
    uint64_t k_count[NMPC_K_COUNT] = {0};
};

namespace {

int use_device(const nmpc_batch* b) {
    HIP_TRY(hipSetDevice(b->device));
    return 0;
}

int get_event(nmpc_batch* b, hipEvent_t* e) {
    if (!b->free_events.empty()) {
        *e = b->free_events.back();
        b->free_events.pop_back();
        return 0;
    }
    HIP_TRY(hipEventCreate(e));
    return 0;
}

// Runs `launch` on the batch's stream; with profiling on, brackets it with
// HIP events whose elapsed time is charged to `kind`.
template <class F>
int timed_on(nmpc_batch* b, hipStream_t stream, int kind, F launch) {
    if (!b->profiling || b->untimed || !((b->profile_kinds >> kind) & 1u)) {
        launch();
        HIP_TRY(hipGetLastError());
        b->chain_event = nullptr;       // the next timed launch starts its own interval
        return 0;
    }
    TimedLaunch t{kind, nullptr, nullptr, true};
    if (b->chain_launches && b->chain_event) {
        t.start = b->chain_event;
        t.owned = false;
    } else {
        TRY(get_event(b, &t.start));
        HIP_TRY(hipEventRecord(t.start, stream));
    }
    TRY(get_event(b, &t.stop));
    launch();
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(t.stop, stream));
    b->chain_event = b->chain_launches ? t.stop : nullptr;
    b->pending.push_back(t);
    return 0;
}

template <class F>
int timed(nmpc_batch* b, int kind, F launch) {
    return timed_on(b, b->stream, kind, launch);
}

int drain_timings(nmpc_batch* b) {
    for (auto& t : b->pending) {
        HIP_TRY(hipEventSynchronize(t.stop));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, t.start, t.stop));
        b->k_ms[t.kind] += ms;
        b->k_count[t.kind] += 1;
        if (t.owned) b->free_events.push_back(t.start);
        b->free_events.push_back(t.stop);
    }
    b->pending.clear();
    b->chain_event = nullptr;
    return 0;
}

// Device address of `bytes` of input: the caller's pointer when it already is
// device memory, otherwise a staged copy.
// Host memory -> dst on the batch's stream.  *consumed: the source has been copied out of
// (through page-locked memory) and the caller need not wait for the stream on its account.
int copy_in(nmpc_batch* b, void* dst, const void* src, size_t bytes, bool* consumed) {
    *consumed = false;
    PinnedRing& r = b->pinned;
    if (bytes <= kPinnedSlotBytes && !r.failed && !b->untimed) {
        if (!r.base) {
            if (hipHostMalloc((void**)&r.base, kPinnedSlotBytes * kPinnedSlots, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                r.base = nullptr;
                r.failed = true;
            }
            for (unsigned i = 0; r.base && i < kPinnedSlots; i++)
                if (hipEventCreateWithFlags(&r.done[i], hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    r.failed = true;            // (slots and events made so far are released with the batch)
                    break;
                }
        }
        if (r.base && !r.failed) {
            const unsigned slot = r.next++ % kPinnedSlots;
            HIP_TRY(hipEventSynchronize(r.done[slot]));     // the copy that last used the slot
            char* at = r.base + (size_t)slot * kPinnedSlotBytes;
            memcpy(at, src, bytes);
            HIP_TRY(hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipEventRecord(r.done[slot], b->stream));
            *consumed = true;
            return 0;
        }
    }
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, b->stream));
    return 0;
}

// consumed (may be null): as copy_in's; always true for device memory.
int stage_in(nmpc_batch* b, Staging& st, const void* src, size_t bytes, const void** dev,
             bool* consumed = nullptr) {
    if (consumed) *consumed = true;
    if (is_device_pointer(src)) { *dev = src; return 0; }
    TRY(st.reserve(bytes));
    bool done = false;
    TRY(copy_in(b, st.ptr, src, bytes, &done));
    if (consumed) *consumed = done;
    *dev = st.ptr;
    return 0;
}

void sync_config(nmpc_batch* b) {
    memcpy(b->v.lb, b->lb, sizeof b->lb);
    memcpy(b->v.ub, b->ub, sizeof b->ub);
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// Carries out the postponed clears of the stage vectors.
int materialise_stage_vectors(nmpc_batch* b) {
    const BatchView& v = b->v;
    const size_t vec = (size_t)v.B * (v.N + 1) * 15 * sizeof(real_t);
    if (b->stage_zero_pending) {
        HIP_TRY(hipMemsetAsync(v.zLow, 0, (size_t)((char*)v.p - (char*)v.zLow), b->stream));
    } else if (b->zy_zero_pending) {
        HIP_TRY(hipMemsetAsync(v.z, 0, vec, b->stream));
        HIP_TRY(hipMemsetAsync(v.y, 0, 2 * vec, b->stream));
    }
    b->stage_zero_pending = b->zy_zero_pending = false;
    return 0;
}

// The stage bounds nmpc_init writes are the control bounds of that moment
// (ccs-c66x/cnmpc.c:920-931), whatever the setters have been given since.
int timed_init_part2(nmpc_batch* b) {
    BatchView at_init = b->v;
    memcpy(at_init.lb, b->init_lb, sizeof at_init.lb);
    memcpy(at_init.ub, b->init_ub, sizeof at_init.ub);
    return timed(b, NMPC_K_OTHER, [&] { launch_init_stages(at_init, 2, b->stream); });
}

// Carries out the postponed clear of C and c and the postponed re-staging of
// the stages 0..N-1.
int materialise_jacobians(nmpc_batch* b) {
    if (b->stage_init_pending) {
        b->stage_init_pending = false;
        if (int rc = timed_init_part2(b)) return rc;
    }
    if (!b->jacobians_zero_pending) return 0;
    const BatchView& v = b->v;
    HIP_TRY(hipMemsetAsync(v.C, 0, (size_t)v.B * v.N * 180 * sizeof(real_t), b->stream));
    HIP_TRY(hipMemsetAsync(v.c, 0, (size_t)v.B * v.N * 12 * sizeof(real_t), b->stream));
    b->jacobians_zero_pending = false;
    return 0;
}

}  // namespace

extern "C" {

const char* nmpc_last_error(void) { return g_error.c_str(); }

int nmpc_batch_create(nmpc_batch_t** out, uint32_t instances, uint32_t horizon, int device,
                      void* stream) {
    if (!out || instances == 0 || horizon < 2)
        return fail(NMPC_E_ARG, "nmpc_batch_create: need out != NULL, instances >= 1, horizon >= 2");
    // the kernels index (instance, stage) pairs with 32-bit integers
    if ((uint64_t)instances * ((uint64_t)horizon + 1) >= (1ull << 31))
        return fail(NMPC_E_ARG, "nmpc_batch_create: instances x (horizon + 1) = %llu beyond 2^31",
                    (unsigned long long)instances * (horizon + 1ull));
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(NMPC_E_ARG, "nmpc_batch_create: device %d not present (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    if (device < kMaxDevices && !g_scratch_free[device] &&
        hipEventCreateWithFlags(&g_scratch_free[device], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        g_scratch_free[device] = nullptr;   // the no-alloc entry point then reports it
    }
    nmpc_batch* b = new nmpc_batch();
    b->device = device;
    if (stream) {
        b->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete b; return fail(NMPC_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        b->own_stream = true;
    }
    const size_t B = instances, N = horizon, N1 = N + 1;
    BatchView& v = b->v;
    v.B = (int)instances; v.N = (int)horizon;
    v.b0 = 0; v.nb = (int)instances;
    v.head = v.xhead = v.lhead = 0;

    struct Field { void** p; size_t bytes; };
    const size_t d = sizeof(real_t);
    // The last four fields are rebuilt in full before they are read, by every
    // Newton step (wabs), by nmpc_batch_init (C) and by the first solve after it
    // (hess, chol); nmpc_batch_reset skips them.
    Field fields[] = {
        {(void**)&v.xref, B * N1 * 13 * d}, {(void**)&v.uref, B * N * 3 * d},
        {(void**)&v.wind, B * 3 * d},       {(void**)&v.c, B * N * 12 * d},
        {(void**)&v.zLow, B * N1 * 15 * d}, {(void**)&v.zUpp, B * N1 * 15 * d},
        {(void**)&v.q, B * N1 * 15 * d},    {(void**)&v.z, B * N1 * 15 * d},
        {(void**)&v.qStep, B * N1 * 15 * d},{(void**)&v.zUnc, B * N1 * 15 * d},
        {(void**)&v.dz, B * N1 * 15 * d},   {(void**)&v.tmp, B * N1 * 15 * d},
        {(void**)&v.y, B * N1 * 30 * d},    {(void**)&v.p, B * N1 * d},
        {(void**)&v.pStep, B * N1 * d},     {(void**)&v.lam, B * N * 12 * d},
        {(void**)&v.dl, B * N * 12 * d},    {(void**)&v.grad, B * N * 12 * d},
        {(void**)&v.ieq, B * N1 * 15},      {(void**)&v.prev, B * N1 * 15},
        {(void**)&v.chg, B * N1},           {(void**)&v.sc, B * sizeof(Scalars)},
        {(void**)&v.tailCount, 32 * sizeof(int)}, {(void**)&v.tailList, 4 * B * sizeof(int)},
        {(void**)&v.solveCount, 2 * kSolveListStride * sizeof(int)}, {(void**)&v.solveList, B * sizeof(int)},
        {(void**)&v.wabs, B * N1 * 16 * d},
        {(void**)&v.C, B * N * 180 * d},
        {(void**)&v.hess, (B * N * kCholRow + kCholSlack) * d}, {(void**)&v.chol, (B * N * kCholRow + kCholSlack) * d},
    };
    size_t total = 0;
    const size_t n_fields = sizeof fields / sizeof fields[0];
    for (size_t i = 0; i < n_fields; i++) {
        if (i == n_fields - 4) b->reset_bytes = total;
        total += align_up(fields[i].bytes);
    }
    hipError_t e = hipMalloc(&b->arena, total);
    if (e != hipSuccess) {
        if (b->own_stream) (void)hipStreamDestroy(b->stream);
        delete b;
        return fail(NMPC_E_HIP, "hipMalloc(%zu bytes) failed: %s", total, hipGetErrorString(e));
    }
    b->arena_bytes = total;
    e = hipMemsetAsync(b->arena, 0, total, b->stream);
    if (e != hipSuccess) {
        (void)hipFree(b->arena);
        if (b->own_stream) (void)hipStreamDestroy(b->stream);
        delete b;
        return fail(NMPC_E_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
    size_t off = 0;
    for (auto& f : fields) {
        *f.p = (char*)b->arena + off;
        off += align_up(f.bytes);
    }
    memset(b->wq, 0, sizeof b->wq); memset(b->wr, 0, sizeof b->wr);
    memset(b->wt, 0, sizeof b->wt); memset(b->lb, 0, sizeof b->lb);
    memset(b->ub, 0, sizeof b->ub);
    b->fresh.assign(N, 1);
    if (instances >= 512) {     // batches that run their Newton iterations as two parts
        // The part on the batch's stream is the one an iteration waits for: the second
        // stream gets the lowest priority, so that the long part's step is not queued
        // behind the thousands of workgroups of the short part's.
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
        if (hipStreamCreateWithPriority(&b->side_stream, hipStreamNonBlocking, least) != hipSuccess ||
            hipEventCreateWithFlags(&b->fork_event, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&b->join_event, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (b->fork_event) (void)hipEventDestroy(b->fork_event);
            if (b->join_event) (void)hipEventDestroy(b->join_event);
            if (b->side_stream) (void)hipStreamDestroy(b->side_stream);
            b->side_stream = nullptr;       // one part then
            b->fork_event = b->join_event = nullptr;
        }
    }
    *out = b;
    return 0;
}

int nmpc_batch_destroy(nmpc_batch_t* b) {
    if (!b) return 0;
    (void)hipSetDevice(b->device);
    (void)hipStreamSynchronize(b->stream);
    for (auto& t : b->pending) {
        if (t.owned) (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    for (auto ev : b->free_events) (void)hipEventDestroy(ev);
    b->in_refs.release(); b->in_meas.release(); b->in_wind.release();
    b->out_u.release(); b->out_status.release(); b->out_iters.release(); b->scratch.release();
    if (b->plant) (void)hipFree(b->plant);
    if (b->plant_next) (void)hipFree(b->plant_next);
    if (b->cfg) (void)hipFree(b->cfg);
    b->in_cfg.release();
    for (auto ev : b->pinned.done) if (ev) (void)hipEventDestroy(ev);
    if (b->pinned.base) (void)hipHostFree(b->pinned.base);
    if (b->arena) (void)hipFree(b->arena);
    if (b->side_stream) {
        (void)hipStreamSynchronize(b->side_stream);
        (void)hipEventDestroy(b->fork_event);
        (void)hipEventDestroy(b->join_event);
        (void)hipStreamDestroy(b->side_stream);
    }
    if (b->own_stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return 0;
}

int nmpc_batch_reset(nmpc_batch_t* b) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    TRY(use_device(b));
    const BatchView& v = b->v;
    char* base = (char*)b->arena;
    HIP_TRY(hipMemsetAsync(base, 0, (size_t)((char*)v.zLow - base), b->stream));
    HIP_TRY(hipMemsetAsync(v.p, 0, (size_t)(base + b->reset_bytes - (char*)v.p), b->stream));
    b->stage_zero_pending = true;
    b->zy_zero_pending = false;
    b->stage_init_pending = false;      // what it would write is zero again
    b->v.head = b->v.xhead = b->v.lhead = 0;
    b->initialised = false;
    std::fill(b->fresh.begin(), b->fresh.end(), 1);
    return 0;
}

int nmpc_batch_synchronize(nmpc_batch_t* b) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    TRY(use_device(b));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

// The batch-wide setters keep the host copy and, once instances have records of
// their own, write the value into every record as well.
#define SETTER(name, field, n, at)                                      \
    int name(nmpc_batch_t* b, const real_t* coeffs) {                   \
        if (!b || !coeffs) return fail(NMPC_E_ARG, #name ": null argument"); \
        memcpy(b->field, coeffs, (n) * sizeof(real_t));                 \
        if (b->cfg) {                                                   \
            TRY(use_device(b));                                         \
            launch_cfg_set(b->cfg, 0, b->v.B, at, n, nullptr, b->field, b->stream); \
            HIP_TRY(hipGetLastError());                                 \
        }                                                               \
        return 0;                                                       \
    }
SETTER(nmpc_batch_set_state_weights, wq, 12, kCfgWq)
SETTER(nmpc_batch_set_control_weights, wr, 3, kCfgWr)
SETTER(nmpc_batch_set_terminal_weights, wt, 12, kCfgWt)
SETTER(nmpc_batch_set_lower_control_bound, lb, 3, kCfgLb)
SETTER(nmpc_batch_set_upper_control_bound, ub, 3, kCfgUb)
#undef SETTER

namespace {
// Records for every instance, filled with what the batch has had so far: the
// setters' values, and what the last nmpc_batch_init latched from them.
int ensure_cfg(nmpc_batch* b) {
    if (b->cfg) return 0;
    const int B = b->v.B;
    HIP_TRY(hipMalloc((void**)&b->cfg, (size_t)B * kCfgStride * sizeof(real_t)));
    HIP_TRY(hipMemsetAsync(b->cfg, 0, (size_t)B * kCfgStride * sizeof(real_t), b->stream));
    const struct { int at, n; const real_t* src; } parts[] = {
        {kCfgWq, 12, b->wq}, {kCfgWr, 3, b->wr}, {kCfgWt, 12, b->wt}, {kCfgLb, 3, b->lb}, {kCfgUb, 3, b->ub},
        {kCfgH, 15, b->v.H}, {kCfgHt, 12, b->v.Ht}, {kCfgLb0, 3, b->init_lb}, {kCfgUb0, 3, b->init_ub}};
    for (const auto& p : parts) launch_cfg_set(b->cfg, 0, B, p.at, p.n, nullptr, p.src, b->stream);
    HIP_TRY(hipGetLastError());
    b->v.cfg = b->cfg;
    return 0;
}

// values[count][n] (host or device memory, null: leave as it is) into the records.
int set_cfg_part(nmpc_batch* b, uint32_t first, uint32_t count, int at, int n, const real_t* values,
                 bool* host) {
    if (!values) return 0;
    const void* dev = nullptr;
    if (!is_device_pointer(values)) {
        // one staging buffer for the parts of one call: each part behind the one before
        *host = true;
    }
    TRY(stage_in(b, b->in_cfg, values, (size_t)count * n * sizeof(real_t), &dev));
    launch_cfg_set(b->cfg, (int)first, (int)count, at, n, (const real_t*)dev, nullptr, b->stream);
    HIP_TRY(hipGetLastError());
    // the staging buffer is reused by the next part
    if (!is_device_pointer(values)) HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}
}  // namespace

int nmpc_batch_set_instance_weights(nmpc_batch_t* b, uint32_t first, uint32_t count, const real_t* state,
                                    const real_t* control, const real_t* terminal) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    if ((uint64_t)first + count > (uint64_t)b->v.B)
        return fail(NMPC_E_ARG, "nmpc_batch_set_instance_weights: instances %u..%u beyond the batch's %d",
                    first, first + count - 1, b->v.B);
    if (count == 0) return 0;
    TRY(use_device(b));
    TRY(ensure_cfg(b));
    bool host = false;
    TRY(set_cfg_part(b, first, count, kCfgWq, 12, state, &host));
    TRY(set_cfg_part(b, first, count, kCfgWr, 3, control, &host));
    TRY(set_cfg_part(b, first, count, kCfgWt, 12, terminal, &host));
    return 0;
}

int nmpc_batch_set_instance_control_bounds(nmpc_batch_t* b, uint32_t first, uint32_t count,
                                           const real_t* lower, const real_t* upper) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    if ((uint64_t)first + count > (uint64_t)b->v.B)
        return fail(NMPC_E_ARG, "nmpc_batch_set_instance_control_bounds: instances %u..%u beyond the batch's %d",
                    first, first + count - 1, b->v.B);
    if (count == 0) return 0;
    TRY(use_device(b));
    TRY(ensure_cfg(b));
    bool host = false;
    TRY(set_cfg_part(b, first, count, kCfgLb, 3, lower, &host));
    TRY(set_cfg_part(b, first, count, kCfgUb, 3, upper, &host));
    return 0;
}

int nmpc_batch_init(nmpc_batch_t* b) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    TRY(use_device(b));
    BatchView& v = b->v;
    const bool idq = is_identity(b->wq, 12), idr = is_identity(b->wr, 3);
    for (int i = 0; i < 12; i++) v.H[i] = idq ? 1. : b->wq[i];
    for (int i = 0; i < 3; i++) v.H[12 + i] = idr ? 1. : b->wr[i];
    const bool idt = is_identity(b->wt, 12);
    for (int i = 0; i < 12; i++) v.Ht[i] = idt ? 1. : b->wt[i];
    sync_config(b);
    memcpy(b->init_lb, b->lb, sizeof b->init_lb);
    memcpy(b->init_ub, b->ub, sizeof b->init_ub);
    if (b->cfg) {       // the same, instance by instance
        launch_cfg_latch(b->cfg, v.B, b->stream);
        HIP_TRY(hipGetLastError());
    }
    v.head = 0;   // the interval order is rebuilt (ccs-c66x/cnmpc.c:817-828)
    // Jacobians and residuals start at zero (ccs-c66x/cnmpc.c:913-917)
    b->jacobians_zero_pending = true;
    if (b->stage_zero_pending) {        // the kernel below writes all but z, y and the scratch
        b->stage_zero_pending = false;
        b->zy_zero_pending = true;
    }
    TRY(timed(b, NMPC_K_OTHER, [&] { launch_init_stages(v, 1, b->stream); }));
    b->stage_init_pending = true;
    b->initialised = true;
    std::fill(b->fresh.begin(), b->fresh.end(), 1);    // no solution to linearise about yet
    return 0;
}

int nmpc_batch_set_wind_velocity(nmpc_batch_t* b, const real_t* wind) {
    if (!b || !wind) return fail(NMPC_E_ARG, "nmpc_batch_set_wind_velocity: null argument");
    TRY(use_device(b));
    const size_t bytes = (size_t)b->v.B * 3 * sizeof(real_t);
    if (is_device_pointer(wind)) {
        HIP_TRY(hipMemcpyAsync(b->v.wind, wind, bytes, hipMemcpyDeviceToDevice, b->stream));
        return 0;
    }
    bool consumed = false;
    TRY(copy_in(b, b->v.wind, wind, bytes, &consumed));
    if (!consumed) HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

// Reference points already in device memory ([B][count][16]) -> rings, then the
// linearisation of the stages they complete.
static int store_and_linearise(nmpc_batch* b, const real_t* dev, uint32_t first, uint32_t count) {
    sync_config(b);
    if (b->stage_zero_pending) TRY(materialise_stage_vectors(b));
    TRY(timed(b, NMPC_K_OTHER, [&] {
        launch_store_refs(b->v, dev, (int)first, (int)count, b->stream);
    }));
    // point i linearises stage i-1
    const int lin_first = first == 0 ? 0 : (int)first - 1;
    const int lin_count = first == 0 ? (int)count - 1 : (int)count;
    if (lin_first == 0 && lin_count == b->v.N)
        b->jacobians_zero_pending = b->stage_init_pending = false;  // every stage is rewritten below
    else
        TRY(materialise_jacobians(b));
    if (lin_count > 0) {
        TRY(timed(b, NMPC_K_LINEARISE, [&] {
            launch_linearise(b->v, lin_first, lin_count, b->stream);
        }));
        std::fill(b->fresh.begin() + lin_first, b->fresh.begin() + lin_first + lin_count, 1);
    }
    return 0;
}

int nmpc_batch_set_reference_points(nmpc_batch_t* b, const real_t* refs, uint32_t first,
                                    uint32_t count) {
    if (!b || !refs) return fail(NMPC_E_ARG, "nmpc_batch_set_reference_points: null argument");
    if (count == 0) return 0;
    if ((uint64_t)first + count > (uint64_t)b->v.N + 1)
        return fail(NMPC_E_ARG, "reference index %u beyond horizon %d", first + count - 1, b->v.N);
    TRY(use_device(b));
    const void* dev = nullptr;
    bool consumed = true;
    TRY(stage_in(b, b->in_refs, refs, (size_t)b->v.B * count * 16 * sizeof(real_t), &dev, &consumed));
    TRY(store_and_linearise(b, (const real_t*)dev, first, count));
    if (!consumed) HIP_TRY(hipStreamSynchronize(b->stream));  // caller may reuse its buffer
    return 0;
}

int nmpc_batch_set_reference_from_waypoints(nmpc_batch_t* b, const real_t* waypoints,
                                            uint32_t n_waypoints, real_t t0, real_t dt,
                                            uint32_t first, uint32_t count) {
    if (!b || !waypoints || n_waypoints == 0)
        return fail(NMPC_E_ARG, "nmpc_batch_set_reference_from_waypoints: null or empty argument");
    if (count == 0) return 0;
    if ((uint64_t)first + count > (uint64_t)b->v.N + 1)
        return fail(NMPC_E_ARG, "reference index %u beyond horizon %d", first + count - 1, b->v.N);
    TRY(use_device(b));
    const size_t B = (size_t)b->v.B;
    const void* dwp = nullptr;
    const bool host = !is_device_pointer(waypoints);
    TRY(stage_in(b, b->in_wind, waypoints, B * n_waypoints * 17 * sizeof(real_t), &dwp));
    TRY(b->scratch.reserve(B * count * 16 * sizeof(real_t)));
    TRY(timed(b, NMPC_K_OTHER, [&] {
        launch_refs_from_waypoints((int)B, (const real_t*)dwp, (int)n_waypoints, t0, dt, (int)first,
                                   (int)count, (real_t*)b->scratch.ptr, b->stream);
    }));
    TRY(store_and_linearise(b, (const real_t*)b->scratch.ptr, first, count));
    if (host) HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

// ---- the entry point that allocates nothing -------------------------------------
// The interpolated reference points of nmpc_batch_set_reference_from_waypoints
// pass from one kernel to the next through a buffer the library allocates on
// demand.  Here that buffer is a fixed array in the code object: 65 536 points
// of 16 doubles, 8 MiB on every GPU the library runs on, held until the
// process ends and shared by all batches on that GPU.
constexpr uint32_t kStaticPoints = NMPC_STATIC_SCRATCH_POINTS;
static __device__ __align__(256) real_t scratch[(size_t)kStaticPoints * 16];
static real_t* g_scratch_ptr[kMaxDevices];      // device address of `scratch`, per GPU

// The array's name is only a handle on the host; its memory on the current
// device is found once per device.
static int static_scratch(int device, real_t** out) {
    if (device < 0 || device >= kMaxDevices)
        return fail(NMPC_E_ARG, "device %d beyond the %d the library keeps scratch addresses for",
                    device, kMaxDevices);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(NMPC_E_HIP, "hipGetSymbolAddress returned a null address");
        g_scratch_ptr[device] = (real_t*)ptr;
    }
    *out = g_scratch_ptr[device];
    return 0;
}

int nmpc_batch_set_reference_from_waypoints_noalloc(nmpc_batch_t* b, const real_t* waypoints,
                                                    uint32_t n_waypoints, real_t t0, real_t dt,
                                                    uint32_t first, uint32_t count) {
    if (!b || !waypoints || n_waypoints == 0)
        return fail(NMPC_E_ARG, "nmpc_batch_set_reference_from_waypoints_noalloc: null or empty argument");
    if (count == 0) return 0;
    if ((uint64_t)first + count > (uint64_t)b->v.N + 1)
        return fail(NMPC_E_ARG, "reference index %u beyond horizon %d", first + count - 1, b->v.N);
    const uint64_t points = (uint64_t)b->v.B * count;
    if (points > kStaticPoints)
        return fail(NMPC_E_ARG, "nmpc_batch_set_reference_from_waypoints_noalloc: %llu reference "
                    "points (instances x count) exceed the fixed scratch of %u points; load fewer "
                    "points per call or use nmpc_batch_set_reference_from_waypoints",
                    (unsigned long long)points, kStaticPoints);
    TRY(use_device(b));
    if (!is_device_pointer(waypoints))
        return fail(NMPC_E_ARG, "nmpc_batch_set_reference_from_waypoints_noalloc: waypoints must be "
                    "device memory (a host buffer would need a staging allocation)");
    real_t* points_dev = nullptr;
    TRY(static_scratch(b->device, &points_dev));
    struct Untimed {        // timing a launch may create events
        nmpc_batch* b;
        explicit Untimed(nmpc_batch* bb) : b(bb) { b->untimed = true; }
        ~Untimed() { b->untimed = false; }
    } untimed(b);
    hipEvent_t scratch_free = b->device < kMaxDevices ? g_scratch_free[b->device] : nullptr;
    if (!scratch_free) return fail(NMPC_E_HIP, "no event to order the users of the fixed scratch with");
    // the call before this one on this GPU, on whatever stream, has read its points
    HIP_TRY(hipStreamWaitEvent(b->stream, scratch_free, 0));
    // every entry of the points the kernels below read is written first: nothing
    // of an earlier call's contents is looked at
    launch_refs_from_waypoints(b->v.B, (const real_t*)waypoints, (int)n_waypoints, t0, dt, (int)first,
                               (int)count, points_dev, b->stream);
    HIP_TRY(hipGetLastError());
    const int rc = store_and_linearise(b, points_dev, first, count);
    HIP_TRY(hipEventRecord(scratch_free, b->stream));
    return rc;
}

int nmpc_batch_set_rti(nmpc_batch_t* b, int enabled) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    b->rti = enabled != 0;
    return 0;
}

int nmpc_batch_preparation_step(nmpc_batch_t* b) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    if (!b->rti || !b->initialised) return 0;      // the reference's preparation step is empty
    TRY(use_device(b));
    sync_config(b);
    TRY(materialise_jacobians(b));
    if (b->stage_zero_pending) TRY(materialise_stage_vectors(b));
    // Nothing to do unless some stage carries a solution of the last feedback step.
    const int N = b->v.N;
    if (std::find(b->fresh.begin(), b->fresh.begin() + N, 0) == b->fresh.begin() + N) return 0;
    // Per instance, on the device: an instance whose last solve did not converge
    // is left exactly as it is (no solution to linearise about, and its
    // multipliers are the progress made so far).  For the others every stage is
    // linearised again, about its solution where it has one and about the
    // reference otherwise, and the duals start afresh: multipliers of the old
    // linearisation would, with Jacobians that moved, put the stage solutions
    // into the control bounds all along the horizon, and the dual Newton
    // iteration then needs more than its five steps to recover.
    const BatchView& v = b->v;
    TRY(timed(b, NMPC_K_OTHER, [&] {
        launch_rti_reset(v, b->stream);
        launch_init_stages(v, 3, b->stream);    // terminal stage against the new multipliers
    }));
    for (int k = 0; k < N;) {
        const bool about_solution = !b->fresh[k];
        int e = k;
        while (e < N && !b->fresh[e] == about_solution) ++e;
        TRY(timed(b, NMPC_K_LINEARISE, [&] {
            launch_linearise(b->v, k, e - k, b->stream, about_solution, true);
        }));
        k = e;
    }
    std::fill(b->fresh.begin(), b->fresh.begin() + N, 1);
    return 0;
}

int nmpc_batch_relinearise(nmpc_batch_t* b) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    if (!b->initialised) return fail(NMPC_E_STATE, "nmpc_batch_relinearise before nmpc_batch_init");
    TRY(use_device(b));
    sync_config(b);
    b->jacobians_zero_pending = b->stage_init_pending = false;      // every stage is rewritten
    TRY(timed(b, NMPC_K_LINEARISE, [&] { launch_linearise(b->v, 0, b->v.N, b->stream); }));
    std::fill(b->fresh.begin(), b->fresh.end(), 1);
    return 0;
}

namespace {

// Newton system, step and step-length searches of iteration `it` for the instances of view v.
int newton_iteration(nmpc_batch* b, const BatchView& v, int it, hipStream_t s, bool cold) {
    TRY(timed_on(b, s, NMPC_K_NEWTON_SOLVE, [&] { launch_newton_solve(v, s, cold); }));
    TRY(timed_on(b, s, NMPC_K_NEWTON_STEP, [&] { launch_newton_step(v, it + 1, s); }));
    if (line_search_tail_enabled(v))
        TRY(timed_on(b, s, NMPC_K_LINE_SEARCH_TAIL, [&] { launch_line_search_tail(v, it + 1, s); }));
    return 0;
}

// A launch of the solve lasts as long as the longest factorisation sweep, which after
// the first iteration a sixth of the instances have, and below the size at which the
// solve fills the chip twice over most of the chip idles meanwhile; the instances with
// short sweeps then go through their iteration beside it.  Larger batches gain less
// (2 % at 8 192 to 32 768 instances: the short part's bandwidth-bound step beside the
// long part's solve).  Small batches are bound by the number of launches, which this doubles.
bool two_parts(const nmpc_batch* b) {
    static const bool off = getenv("NMPC_ONE_PART") != nullptr;
    return !off && b->side_stream && b->v.nb >= 512;
}

}  // namespace

int nmpc_batch_feedback_step(nmpc_batch_t* b, const real_t* measurement) {
    if (!b || !measurement) return fail(NMPC_E_ARG, "nmpc_batch_feedback_step: null argument");
    if (!b->initialised) return fail(NMPC_E_STATE, "nmpc_batch_feedback_step before nmpc_batch_init");
    TRY(use_device(b));
    sync_config(b);
    TRY(materialise_jacobians(b));
    b->zy_zero_pending = false;             // the opening step writes every z and y
    const void* dev = nullptr;
    bool consumed = true;
    TRY(stage_in(b, b->in_meas, measurement, (size_t)b->v.B * 13 * sizeof(real_t), &dev, &consumed));
    const bool host = !consumed;
    const real_t* dmeas = (const real_t*)dev;
    const BatchView& v = b->v;
    // the launches below follow each other with nothing between them
    struct ChainScope {
        nmpc_batch* b;
        explicit ChainScope(nmpc_batch* bb) : b(bb) { b->chain_launches = true; b->chain_event = nullptr; }
        ~ChainScope() { b->chain_launches = false; b->chain_event = nullptr; }
    } chain(b);
    // After a linearisation of every stage the first Newton system is factorised in
    // full for every instance: nothing to run beside it.
    const bool all_fresh = std::all_of(b->fresh.begin(), b->fresh.end(), [](char f) { return f != 0; });
    TRY(timed(b, NMPC_K_QP_BEGIN, [&] { launch_qp_begin(v, dmeas, b->stream); }));
    for (int it = 1; it <= max_newton_iterations(); ++it) {
        if (two_parts(b) && !(it == 1 && all_fresh)) {
            TRY(timed(b, NMPC_K_OTHER, [&] { launch_list_solves(v, ++b->solve_lists, b->stream); }));
            HIP_TRY(hipEventRecord(b->fork_event, b->stream));
            HIP_TRY(hipStreamWaitEvent(b->side_stream, b->fork_event, 0));
            // Whatever fails from here on, the second stream is joined back before the
            // call returns: nothing queued on it may outlive the order of the batch's stream.
            const int rc_long = newton_iteration(b, solve_list_view(v, b->solve_lists, 1), it, b->stream, all_fresh);
            int rc_short;
            {
                // intervals timed on the other stream do not continue the ones of this stream
                hipEvent_t kept = b->chain_event;
                b->chain_launches = false;
                rc_short = newton_iteration(b, solve_list_view(v, b->solve_lists, 2), it, b->side_stream, all_fresh);
                b->chain_launches = true;
                b->chain_event = kept;
            }
            const hipError_t e_rec = hipEventRecord(b->join_event, b->side_stream);
            const hipError_t e_wait = e_rec == hipSuccess ? hipStreamWaitEvent(b->stream, b->join_event, 0) : e_rec;
            if (e_wait != hipSuccess) {
                // no event to order the streams with: wait for the second one here
                (void)hipStreamSynchronize(b->side_stream);
                return fail(NMPC_E_HIP, "joining the second stream failed: %s", hipGetErrorString(e_wait));
            }
            TRY(rc_long);
            TRY(rc_short);
            b->chain_event = nullptr;
        } else if (newton_solve_two_waves(v)) {
            // a grid that fills the chip several times over: the solve takes the running
            // instances in the order of the list, the kernels with one workgroup per instance
            // gain nothing from it
            TRY(timed(b, NMPC_K_OTHER, [&] { launch_list_solves(v, ++b->solve_lists, b->stream); }));
            const BatchView listed = solve_list_view(v, b->solve_lists, 0);
            TRY(timed(b, NMPC_K_NEWTON_SOLVE, [&] { launch_newton_solve(listed, b->stream); }));
            TRY(timed(b, NMPC_K_NEWTON_STEP, [&] { launch_newton_step(v, it + 1, b->stream); }));
            if (line_search_tail_enabled(v))
                TRY(timed(b, NMPC_K_LINE_SEARCH_TAIL, [&] { launch_line_search_tail(v, it + 1, b->stream); }));
        } else {
            TRY(newton_iteration(b, v, it, b->stream, all_fresh));
        }
    }
    TRY(timed(b, NMPC_K_OTHER, [&] { launch_qp_finish(v, b->stream); }));
    std::fill(b->fresh.begin(), b->fresh.end(), 0);     // every stage now carries a solution
    if (host) HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

int nmpc_batch_get_controls(nmpc_batch_t* b, real_t* controls, int32_t* status,
                            int32_t* iterations) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    TRY(use_device(b));
    const size_t B = (size_t)b->v.B;
    const bool du = controls && is_device_pointer(controls);
    const bool ds = status && is_device_pointer(status);
    const bool di = iterations && is_device_pointer(iterations);
    if (!du) TRY(b->out_u.reserve(B * 3 * sizeof(real_t)));
    if (!ds) TRY(b->out_status.reserve(B * sizeof(int)));
    if (!di) TRY(b->out_iters.reserve(B * sizeof(int)));
    real_t* pu = du ? controls : (real_t*)b->out_u.ptr;
    int* ps = ds ? status : (int*)b->out_status.ptr;
    int* pi = di ? iterations : (int*)b->out_iters.ptr;
    TRY(timed(b, NMPC_K_OTHER, [&] { launch_gather_results(b->v, pu, ps, pi, b->stream); }));
    if (controls && !du)
        HIP_TRY(hipMemcpyAsync(controls, pu, B * 3 * sizeof(real_t), hipMemcpyDeviceToHost, b->stream));
    if (status && !ds)
        HIP_TRY(hipMemcpyAsync(status, ps, B * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    if (iterations && !di)
        HIP_TRY(hipMemcpyAsync(iterations, pi, B * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    const bool any_host = (controls && !du) || (status && !ds) || (iterations && !di);
    if (any_host) HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

int nmpc_batch_update_horizon(nmpc_batch_t* b, const real_t* new_reference) {
    if (!b || !new_reference) return fail(NMPC_E_ARG, "nmpc_batch_update_horizon: null argument");
    TRY(use_device(b));
    sync_config(b);
    TRY(materialise_jacobians(b));
    if (b->stage_zero_pending) TRY(materialise_stage_vectors(b));
    BatchView& v = b->v;
    const void* dev = nullptr;
    bool consumed = true;
    TRY(stage_in(b, b->in_refs, new_reference, (size_t)v.B * 16 * sizeof(real_t), &dev, &consumed));
    const bool host = !consumed;
    // shift references, multipliers and stage records by one position
    v.xhead = (v.xhead + 1) % (v.N + 1);
    v.lhead = (v.lhead + 1) % v.N;
    v.head = (v.head + 1) % v.N;
    TRY(timed(b, NMPC_K_OTHER, [&] { launch_shift_lambda_tail(v, b->stream); }));
    TRY(timed(b, NMPC_K_OTHER, [&] {
        launch_store_refs(v, (const real_t*)dev, v.N, 1, b->stream);
    }));
    TRY(timed(b, NMPC_K_LINEARISE, [&] { launch_linearise(v, v.N - 1, 1, b->stream); }));
    b->fresh.erase(b->fresh.begin());       // the stage records moved up by one position
    b->fresh.push_back(1);
    if (host) HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

// ---- plant roll-forward --------------------------------------------------------
static int plant_buffer(nmpc_batch* b) {
    if (b->plant) return 0;
    const size_t bytes = (size_t)b->v.B * 13 * sizeof(real_t);
    HIP_TRY(hipMalloc((void**)&b->plant, bytes));
    HIP_TRY(hipMalloc((void**)&b->plant_next, bytes));
    HIP_TRY(hipMemsetAsync(b->plant, 0, bytes, b->stream));
    return 0;
}

int nmpc_batch_plant_set_state(nmpc_batch_t* b, const real_t* state) {
    if (!b || !state) return fail(NMPC_E_ARG, "nmpc_batch_plant_set_state: null argument");
    TRY(use_device(b));
    TRY(plant_buffer(b));
    HIP_TRY(hipMemcpyAsync(b->plant, state, (size_t)b->v.B * 13 * sizeof(real_t), hipMemcpyDefault,
                           b->stream));
    if (!is_device_pointer(state)) HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

int nmpc_batch_plant_get_state(nmpc_batch_t* b, real_t* state) {
    if (!b || !state) return fail(NMPC_E_ARG, "nmpc_batch_plant_get_state: null argument");
    TRY(use_device(b));
    TRY(plant_buffer(b));
    HIP_TRY(hipMemcpyAsync(state, b->plant, (size_t)b->v.B * 13 * sizeof(real_t), hipMemcpyDefault,
                           b->stream));
    if (!is_device_pointer(state)) HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

int nmpc_batch_plant_integrate(nmpc_batch_t* b, real_t dt, const real_t* controls,
                               const real_t* wind) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    TRY(use_device(b));
    TRY(plant_buffer(b));
    const size_t B = (size_t)b->v.B;
    const void* du = nullptr;
    const void* dw = b->v.wind;
    bool host = false;
    if (controls) {
        host = host || !is_device_pointer(controls);
        TRY(stage_in(b, b->in_meas, controls, B * 3 * sizeof(real_t), &du));
    } else {
        // the controls every instance would report now
        TRY(b->out_u.reserve(B * 3 * sizeof(real_t)));
        TRY(b->out_status.reserve(B * sizeof(int)));
        TRY(b->out_iters.reserve(B * sizeof(int)));
        TRY(timed(b, NMPC_K_OTHER, [&] {
            launch_gather_results(b->v, (real_t*)b->out_u.ptr, (int*)b->out_status.ptr,
                                  (int*)b->out_iters.ptr, b->stream);
        }));
        du = b->out_u.ptr;
    }
    if (wind) {
        host = host || !is_device_pointer(wind);
        TRY(stage_in(b, b->in_wind, wind, B * 3 * sizeof(real_t), &dw));
    }
    TRY(timed(b, NMPC_K_OTHER, [&] {
        launch_rk4_free((long)B, b->plant, (const real_t*)du, (const real_t*)dw, dt, b->plant_next,
                        b->stream);
    }));
    std::swap(b->plant, b->plant_next);
    if (host) HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

int nmpc_batch_feedback_step_from_plant(nmpc_batch_t* b) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    if (!b->plant) return fail(NMPC_E_STATE, "nmpc_batch_feedback_step_from_plant before nmpc_batch_plant_set_state");
    return nmpc_batch_feedback_step(b, b->plant);
}

uint32_t nmpc_batch_instances(const nmpc_batch_t* b) { return b ? (uint32_t)b->v.B : 0; }
uint32_t nmpc_batch_horizon(const nmpc_batch_t* b) { return b ? (uint32_t)b->v.N : 0; }

// ---- introspection ---------------------------------------------------------
static int d2h(nmpc_batch* b, void* dst, const void* src, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, b->stream));
    return 0;
}

int nmpc_batch_get_stage(nmpc_batch_t* b, uint32_t instance, uint32_t k, real_t* o) {
    if (!b || !o) return fail(NMPC_E_ARG, "nmpc_batch_get_stage: null argument");
    const BatchView& v = b->v;
    if (instance >= (uint32_t)v.B || k > (uint32_t)v.N)
        return fail(NMPC_E_ARG, "nmpc_batch_get_stage: index out of range");
    TRY(use_device(b));
    TRY(materialise_jacobians(b));
    TRY(materialise_stage_vectors(b));
    const size_t N = v.N, N1 = N + 1, ib = instance;
    const size_t s = k < N ? (size_t)((v.head + k) % v.N) : N;
    const size_t nv = k < N ? 15 : 12;
    memset(o, 0, 332 * sizeof(real_t));
    const size_t d = sizeof(real_t);
    if (k < N) {
        TRY(d2h(b, o, v.C + (ib * N + s) * 180, 180 * d));
        TRY(d2h(b, o + 180, v.c + (ib * N + s) * 12, 12 * d));
    }
    const size_t so = (ib * N1 + s) * 15;
    TRY(d2h(b, o + 192, v.zLow + so, nv * d));
    TRY(d2h(b, o + 207, v.zUpp + so, nv * d));
    TRY(d2h(b, o + 222, v.q + so, nv * d));
    TRY(d2h(b, o + 237, v.p + ib * N1 + s, d));
    TRY(d2h(b, o + 238, v.z + so, nv * d));
    TRY(d2h(b, o + 253, v.y + 2 * so, 2 * nv * d));
    TRY(d2h(b, o + 283, v.qStep + so, nv * d));
    TRY(d2h(b, o + 298, v.pStep + ib * N1 + s, d));
    TRY(d2h(b, o + 299, v.zUnc + so, nv * d));
    TRY(d2h(b, o + 314, v.dz + so, nv * d));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

uint32_t nmpc_batch_global_size(const nmpc_batch_t* b) {
    if (!b) return 0;
    const uint32_t N = (uint32_t)b->v.N;
    return 3 * 12 * N + 2 * 288 * N + 2 * 15 * (N + 1) + 13 * (N + 1) + 3 * N + 3;
}

int nmpc_batch_get_global(nmpc_batch_t* b, uint32_t instance, real_t* o) {
    if (!b || !o) return fail(NMPC_E_ARG, "nmpc_batch_get_global: null argument");
    const BatchView& v = b->v;
    if (instance >= (uint32_t)v.B) return fail(NMPC_E_ARG, "instance out of range");
    TRY(use_device(b));
    const size_t N = v.N, N1 = N + 1, ib = instance, d = sizeof(real_t);
    // lambda, position order
    for (size_t k = 0; k < N; k++) {
        const size_t ls = (size_t)((v.lhead + (int)k) % v.N);
        TRY(d2h(b, o + 12 * k, v.lam + (ib * N + ls) * 12, 12 * d));
    }
    o += 12 * N;
    TRY(d2h(b, o, v.dl + ib * N * 12, 12 * N * d)); o += 12 * N;
    TRY(d2h(b, o, v.grad + ib * N * 12, 12 * N * d)); o += 12 * N;
    // Newton Hessian and factor in the reference's [12][24] block rows (types.h:265-283);
    // the entries right of the diagonal, which the factorisation never reads, come out
    // as zero (the reference computes them in the Hessian and leaves them zero in the factor)
    std::vector<real_t> packedH(kCholRow * N), packed(kCholRow * N);
    TRY(d2h(b, packedH.data(), v.hess + ib * N * kCholRow, kCholRow * N * d));
    real_t* oh = o; o += 288 * N;
    TRY(d2h(b, packed.data(), v.chol + ib * N * kCholRow, kCholRow * N * d));
    real_t* of = o; o += 288 * N;
    std::vector<int8_t> ie(N1 * 15), pr(N1 * 15);
    TRY(d2h(b, ie.data(), v.ieq + ib * N1 * 15, N1 * 15));
    TRY(d2h(b, pr.data(), v.prev + ib * N1 * 15, N1 * 15));
    real_t* oi = o; o += 15 * N1;
    real_t* op = o; o += 15 * N1;
    for (size_t k = 0; k < N1; k++) {
        const size_t xs = (size_t)((v.xhead + (int)k) % (v.N + 1));
        TRY(d2h(b, o + 13 * k, v.xref + (ib * N1 + xs) * 13, 13 * d));
    }
    o += 13 * N1;
    for (size_t k = 0; k < N; k++) {
        const size_t s = (size_t)((v.head + (int)k) % v.N);
        TRY(d2h(b, o + 3 * k, v.uref + (ib * N + s) * 3, 3 * d));
    }
    o += 3 * N;
    Scalars sc;
    TRY(d2h(b, &sc, v.sc + ib, sizeof sc));
    HIP_TRY(hipStreamSynchronize(b->stream));
    for (size_t k = 0; k < N; k++)
        for (size_t l = 0; l < 12; l++)
            for (size_t j = 0; j < 12; j++) {
                of[k * 288 + l * 24 + j] = packed[k * kCholRow + 12 * l + j];
                of[k * 288 + l * 24 + 12 + j] = j <= l ? packed[k * kCholRow + kCholDiagRow[l] + j] : (real_t)0;
                oh[k * 288 + l * 24 + j] = packedH[k * kCholRow + 12 * l + j];
                oh[k * 288 + l * 24 + 12 + j] = j <= l ? packedH[k * kCholRow + kCholDiagRow[l] + j] : (real_t)0;
            }
    for (size_t k = 0; k < N1; k++)
        for (size_t i = 0; i < 15; i++) {
            const bool valid = i < (k < N ? 15u : 12u);
            oi[k * 15 + i] = valid ? (real_t)ie[k * 15 + i] : 0.0;
            op[k * 15 + i] = valid ? (real_t)pr[k * 15 + i] : 0.0;
        }
    o[0] = sc.u[0]; o[1] = sc.u[1]; o[2] = sc.u[2];
    return 0;
}

int nmpc_batch_get_solver_info(nmpc_batch_t* b, uint32_t instance, int32_t info[6],
                               real_t* alpha) {
    if (!b || !info) return fail(NMPC_E_ARG, "nmpc_batch_get_solver_info: null argument");
    if (instance >= (uint32_t)b->v.B) return fail(NMPC_E_ARG, "instance out of range");
    TRY(use_device(b));
    Scalars sc;
    TRY(d2h(b, &sc, b->v.sc + instance, sizeof sc));
    HIP_TRY(hipStreamSynchronize(b->stream));
    info[0] = sc.status; info[1] = sc.numIter; info[2] = sc.last;
    info[3] = sc.regularized; info[4] = sc.lsIter; info[5] = sc.lastResult;
    if (alpha) *alpha = sc.alpha;
    return 0;
}

int nmpc_batch_set_profiling(nmpc_batch_t* b, int enabled) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    TRY(use_device(b));
    HIP_TRY(hipStreamSynchronize(b->stream));
    TRY(drain_timings(b));
    b->profiling = enabled != 0;
    if (enabled)
        for (int i = 0; i < NMPC_K_COUNT; i++) { b->k_ms[i] = 0; b->k_count[i] = 0; }
    return 0;
}

int nmpc_batch_set_profiling_kinds(nmpc_batch_t* b, unsigned kinds) {
    if (!b) return fail(NMPC_E_ARG, "null batch");
    TRY(use_device(b));
    HIP_TRY(hipStreamSynchronize(b->stream));
    TRY(drain_timings(b));
    b->profile_kinds = kinds ? kinds : ~0u;
    return 0;
}

int nmpc_batch_get_kernel_time(nmpc_batch_t* b, int kernel, double* total_ms, uint64_t* launches) {
    if (!b || kernel < 0 || kernel >= NMPC_K_COUNT)
        return fail(NMPC_E_ARG, "nmpc_batch_get_kernel_time: bad argument");
    TRY(use_device(b));
    HIP_TRY(hipStreamSynchronize(b->stream));
    TRY(drain_timings(b));
    if (total_ms) *total_ms = b->k_ms[kernel];
    if (launches) *launches = b->k_count[kernel];
    return 0;
}

// ---- stateless sample kernels --------------------------------------------------
namespace {
struct TempCopy {       // device copy of a host input / device home of a host output
    void* dev = nullptr;
    bool owned = false;
    ~TempCopy() { if (owned && dev) (void)hipFree(dev); }
};
int temp_in(TempCopy& t, const void* src, size_t bytes, hipStream_t s) {
    if (is_device_pointer(src)) { t.dev = (void*)src; return 0; }
    HIP_TRY(hipMalloc(&t.dev, bytes));
    t.owned = true;
    HIP_TRY(hipMemcpyAsync(t.dev, src, bytes, hipMemcpyHostToDevice, s));
    return 0;
}
int temp_out(TempCopy& t, void* dst, size_t bytes) {
    if (is_device_pointer(dst)) { t.dev = dst; return 0; }
    HIP_TRY(hipMalloc(&t.dev, bytes));
    t.owned = true;
    return 0;
}
int finish_out(TempCopy& t, void* dst, size_t bytes, hipStream_t s) {
    if (t.owned) HIP_TRY(hipMemcpyAsync(dst, t.dev, bytes, hipMemcpyDeviceToHost, s));
    return 0;
}
}  // namespace

int nmpc_linearise_samples(int device, void* stream, uint64_t n, const real_t* x, const real_t* u,
                           const real_t* x_next, const real_t* wind, const real_t lower[3],
                           const real_t upper[3], real_t* jacobian, real_t* residual) {
    if (!x || !u || !x_next || !wind || !lower || !upper || !jacobian || !residual)
        return fail(NMPC_E_ARG, "nmpc_linearise_samples: null argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    const size_t d = sizeof(real_t);
    TempCopy tx, tu, tn, tw, tj, tr;
    TRY(temp_in(tx, x, n * 13 * d, s)); TRY(temp_in(tu, u, n * 3 * d, s));
    TRY(temp_in(tn, x_next, n * 13 * d, s)); TRY(temp_in(tw, wind, n * 3 * d, s));
    TRY(temp_out(tj, jacobian, n * 180 * d)); TRY(temp_out(tr, residual, n * 12 * d));
    Bounds bnd;
    memcpy(bnd.lb, lower, sizeof bnd.lb); memcpy(bnd.ub, upper, sizeof bnd.ub);
    launch_linearise_free((long)n, (const real_t*)tx.dev, (const real_t*)tu.dev,
                          (const real_t*)tn.dev, (const real_t*)tw.dev, bnd, (real_t*)tj.dev,
                          (real_t*)tr.dev, s);
    HIP_TRY(hipGetLastError());
    TRY(finish_out(tj, jacobian, n * 180 * d, s));
    TRY(finish_out(tr, residual, n * 12 * d, s));
    const bool any_host = tx.owned || tu.owned || tn.owned || tw.owned || tj.owned || tr.owned;
    if (any_host) HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int nmpc_rk4_samples(int device, void* stream, uint64_t n, const real_t* x, const real_t* u,
                     const real_t* wind, real_t delta, real_t* out) {
    if (!x || !u || !wind || !out) return fail(NMPC_E_ARG, "nmpc_rk4_samples: null argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    const size_t d = sizeof(real_t);
    TempCopy tx, tu, tw, to;
    TRY(temp_in(tx, x, n * 13 * d, s)); TRY(temp_in(tu, u, n * 3 * d, s));
    TRY(temp_in(tw, wind, n * 3 * d, s)); TRY(temp_out(to, out, n * 13 * d));
    launch_rk4_free((long)n, (const real_t*)tx.dev, (const real_t*)tu.dev, (const real_t*)tw.dev,
                    delta, (real_t*)to.dev, s);
    HIP_TRY(hipGetLastError());
    TRY(finish_out(to, out, n * 13 * d, s));
    if (tx.owned || tu.owned || tw.owned || to.owned) HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// ============================================================================
// Reference controller interface: one hidden instance.
// ============================================================================
namespace {

struct Single {
    nmpc_batch* batch = nullptr;
    real_t wq[12] = {0}, wr[3] = {0}, wt[12] = {0}, lb[3] = {0}, ub[3] = {0};
    real_t wind[3] = {0};
    real_t plant[13] = {0};     // c/cnmpc.cpp:33 `current`
};
Single g_single;

[[noreturn]] void die(const char* what) {
    fprintf(stderr, "libcnmpc: %s: %s\n", what, g_error.c_str());
    abort();
}
#define MUST(expr) do { if ((expr) != 0) die(#expr); } while (0)

nmpc_batch* single() {
    if (!g_single.batch) {
        int dev = 0;
        if (const char* e = getenv("NMPC_DEVICE")) dev = atoi(e);
        MUST(nmpc_batch_create(&g_single.batch, 1, OCP_HORIZON_LENGTH, dev, nullptr));
    }
    return g_single.batch;
}

void push_config(nmpc_batch* b) {
    memcpy(b->wq, g_single.wq, sizeof b->wq); memcpy(b->wr, g_single.wr, sizeof b->wr);
    memcpy(b->wt, g_single.wt, sizeof b->wt); memcpy(b->lb, g_single.lb, sizeof b->lb);
    memcpy(b->ub, g_single.ub, sizeof b->ub);
}

}  // namespace

void nmpc_init(void) {
    nmpc_batch* b = single();
    push_config(b);
    MUST(nmpc_batch_init(b));
    MUST(nmpc_batch_set_wind_velocity(b, g_single.wind));
}

void nmpc_preparation_step(void) {
    if (g_single.batch) MUST(nmpc_batch_preparation_step(g_single.batch));
}

void nmpc_set_rti(int enabled) { MUST(nmpc_batch_set_rti(single(), enabled)); }

void nmpc_feedback_step(real_t measurement[NMPC_STATE_DIM]) {
    nmpc_batch* b = single();
    push_config(b);
    MUST(nmpc_batch_feedback_step(b, measurement));
}

enum nmpc_result_t nmpc_get_controls(real_t controls[NMPC_CONTROL_DIM]) {
    int32_t status = 0;
    MUST(nmpc_batch_get_controls(single(), controls, &status, nullptr));
    return status == 0 ? NMPC_OK : NMPC_INFEASIBLE;
}

void nmpc_update_horizon(real_t new_reference[NMPC_REFERENCE_DIM]) {
    nmpc_batch* b = single();
    push_config(b);
    MUST(nmpc_batch_update_horizon(b, new_reference));
}

void nmpc_set_state_weights(real_t coeffs[NMPC_DELTA_DIM]) { memcpy(g_single.wq, coeffs, sizeof g_single.wq); }
void nmpc_set_control_weights(real_t coeffs[NMPC_CONTROL_DIM]) { memcpy(g_single.wr, coeffs, sizeof g_single.wr); }
void nmpc_set_terminal_weights(real_t coeffs[NMPC_DELTA_DIM]) { memcpy(g_single.wt, coeffs, sizeof g_single.wt); }
void nmpc_set_lower_control_bound(real_t coeffs[NMPC_CONTROL_DIM]) { memcpy(g_single.lb, coeffs, sizeof g_single.lb); }
void nmpc_set_upper_control_bound(real_t coeffs[NMPC_CONTROL_DIM]) { memcpy(g_single.ub, coeffs, sizeof g_single.ub); }

void nmpc_set_reference_point(real_t coeffs[NMPC_REFERENCE_DIM], uint32_t i) {
    nmpc_batch* b = single();
    push_config(b);
    MUST(nmpc_batch_set_reference_points(b, coeffs, i, 1));
}

void nmpc_set_wind_velocity(real_t x, real_t y, real_t z) {
    g_single.wind[0] = x; g_single.wind[1] = y; g_single.wind[2] = z;
    if (g_single.batch) MUST(nmpc_batch_set_wind_velocity(g_single.batch, g_single.wind));
}

void nmpc_fixedwingdynamics_set_position(real_t lat, real_t lon, real_t alt) {
    g_single.plant[0] = lat; g_single.plant[1] = lon; g_single.plant[2] = alt;
}
void nmpc_fixedwingdynamics_set_velocity(real_t x, real_t y, real_t z) {
    g_single.plant[3] = x; g_single.plant[4] = y; g_single.plant[5] = z;
}
void nmpc_fixedwingdynamics_set_attitude(real_t w, real_t x, real_t y, real_t z) {
    g_single.plant[6] = x; g_single.plant[7] = y; g_single.plant[8] = z; g_single.plant[9] = w;
}
void nmpc_fixedwingdynamics_set_angular_velocity(real_t x, real_t y, real_t z) {
    g_single.plant[10] = x; g_single.plant[11] = y; g_single.plant[12] = z;
}
void nmpc_fixedwingdynamics_set_state(struct nmpc_state_t* in) { memcpy(g_single.plant, in, sizeof g_single.plant); }
void nmpc_fixedwingdynamics_get_state(struct nmpc_state_t* in) { memcpy(in, g_single.plant, sizeof g_single.plant); }

void nmpc_fixedwingdynamics_integrate(float dt, real_t control_vector[NMPC_CONTROL_DIM]) {
    nmpc_batch* b = single();
    real_t next[13];
    MUST(nmpc_rk4_samples(b->device, nullptr, 1, g_single.plant, control_vector, g_single.wind,
                          (real_t)dt, next));
    memcpy(g_single.plant, next, sizeof next);
}

uint32_t nmpc_config_get_state_dim(void) { return NMPC_STATE_DIM; }
uint32_t nmpc_config_get_control_dim(void) { return NMPC_CONTROL_DIM; }
uint32_t nmpc_config_get_horizon_length(void) { return OCP_HORIZON_LENGTH; }
real_t nmpc_config_get_step_length(void) { return OCP_STEP_LENGTH; }
enum nmpc_precision_t nmpc_config_get_precision(void) {
    return sizeof(real_t) == 8 ? NMPC_PRECISION_DOUBLE : NMPC_PRECISION_FLOAT;
}


#ifdef NMPC_TAIL_STATS
// Debug build only: the step-length search counters of the last solve.
int nmpc_debug_tail_stats(nmpc_batch_t* b, int* out16) {
    HIP_TRY(hipStreamSynchronize(b->stream));
    HIP_TRY(hipMemcpy(out16, b->v.tailCount, 16 * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}
#endif
}  // extern "C"
