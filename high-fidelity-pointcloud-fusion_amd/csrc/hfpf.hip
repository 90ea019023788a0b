// csrc/hfpf.hip -- host side of libhfpf.so: handle, HBM pools, kernel orchestration, C ABI (include/hfpf.h).
//
// One HIP stream per handle; every entry point enqueues on it.  clean/extract are the only calls
// that read device counters back (they are synchronisation points in the reference too: clean holds
// grid_mtx_, node.cpp:305-321).  No CPU fallback exists: without a usable HIP device hfpf_create fails.
#include <hip/hip_runtime.h>

#include <cstring>  // rocprim's texture_cache_iterator.hpp uses memset without including it

#include <rocprim/rocprim.hpp>

#include <dlfcn.h>
#include <sched.h>
#include <sys/mman.h>

#if !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#endif
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hfpf.h"
#include "../../include/hfpf_probe.h"
#include "kernels.hpp"

using namespace hfpf;

static_assert(sizeof(hfpf_row) == sizeof(Row), "hfpf_row layout");

namespace {

thread_local std::string g_create_error;  // last error of a call that has no handle (create, dist_unique_id)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct StageSlot {  // pose / frame-id staging for one in-flight integrate call: n poses (12 doubles each), then n frame ids -- ONE upload
    double* h_pose = nullptr;
    uint32_t* h_ids = nullptr;  // = h_pose + 12 * (frames of the call)
    double* d_pose = nullptr;
    uint32_t* d_ids = nullptr;
    uint32_t cap = 0;
    hipEvent_t done = nullptr;
    bool pending = false;
};

struct FrameSlot {  // host-frame staging: a pinned bounce buffer + one slot of the device ring (hfpf_handle::ring_d)
    void* h = nullptr;       // pinned bounce buffer (hfpf_integrate only; hfpf_integrate_pinned copies from the caller's memory)
    size_t cap_h = 0;
    hipEvent_t done = nullptr;    // main stream: the kernels that read the slot have finished
    hipEvent_t copied = nullptr;  // copy stream: the upload into the slot has finished
    bool pending = false;
};

// Helper threads for the bounce copy of hfpf_integrate (caller's pageable frame -> pinned staging): one core moves ~12 GB/s, the
// link takes 55, so the copy is split over the caller + the helpers.  The helpers spin for a fraction of a millisecond after a
// job (a stream of frames keeps them hot) and sleep on a condition variable otherwise.
// Copy with non-temporal stores (dst 16-byte aligned): the staging buffer is only read by the DMA engine afterwards, so the
// lines need not be fetched for ownership nor kept in the cache -- a third less memory traffic than memcpy below glibc's
// non-temporal threshold (which a 1-2 MB share of a frame does not reach).
#if !defined(__HIP_DEVICE_COMPILE__)
inline void stream_copy(char* dst, const char* src, size_t n)
{
    if (((uintptr_t)dst & 15u) != 0) {
        memcpy(dst, src, n);
        return;
    }
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 16));
        const __m128i c = _mm_loadu_si128((const __m128i*)(src + i + 32)), d = _mm_loadu_si128((const __m128i*)(src + i + 48));
        _mm_stream_si128((__m128i*)(dst + i), a);
        _mm_stream_si128((__m128i*)(dst + i + 16), b);
        _mm_stream_si128((__m128i*)(dst + i + 32), c);
        _mm_stream_si128((__m128i*)(dst + i + 48), d);
    }
    _mm_sfence();
    if (i < n) memcpy(dst + i, src + i, n - i);
}
#else
inline void stream_copy(char* dst, const char* src, size_t n) { memcpy(dst, src, n); }
#endif

class StagePool {
  public:
    explicit StagePool(int helpers)
    {
        for (int i = 0; i < helpers; i++) th_.emplace_back([this, i] { run(i + 1); });
    }
    ~StagePool()
    {
        stop_.store(true, std::memory_order_release);
        {
            std::lock_guard<std::mutex> lk(m_);
            cv_.notify_all();
        }
        for (auto& t : th_) t.join();
    }
    void copy(void* dst, const void* src, size_t bytes)
    {
        dst_ = (char*)dst, src_ = (const char*)src, bytes_ = bytes;
        remaining_.store((int)th_.size(), std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        if (sleepers_.load(std::memory_order_acquire) > 0) {
            std::lock_guard<std::mutex> lk(m_);
            cv_.notify_all();
        }
        part(0);
        while (remaining_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
    }

  private:
    void part(int i) const
    {
        const size_t parts = th_.size() + 1;
        const size_t chunk = ((bytes_ + parts - 1) / parts + 4095) & ~(size_t)4095;
        const size_t lo = std::min(bytes_, chunk * (size_t)i), hi = std::min(bytes_, lo + chunk);
        if (hi > lo) stream_copy(dst_ + lo, src_ + lo, hi - lo);
    }
    void run(int i)
    {
        uint64_t seen = 0;
        for (;;) {
            uint64_t g;
            int spins = 0;
            while ((g = gen_.load(std::memory_order_acquire)) == seen && !stop_.load(std::memory_order_acquire)) {
                if (++spins < 20000) {
                    __builtin_ia32_pause();
                } else {
                    std::unique_lock<std::mutex> lk(m_);
                    sleepers_.fetch_add(1, std::memory_order_acq_rel);
                    cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen || stop_.load(std::memory_order_acquire); });
                    sleepers_.fetch_sub(1, std::memory_order_acq_rel);
                    spins = 0;
                }
            }
            if (stop_.load(std::memory_order_acquire)) return;
            seen = g;
            part(i);
            remaining_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> remaining_{0}, sleepers_{0};
    std::atomic<bool> stop_{false};
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t bytes_ = 0;
};

constexpr int kStageSlots = 8;
constexpr size_t kXferChunk = 16u << 20;  // bytes per pinned chunk of extract's row download
#ifndef HFPF_PROBE_FRAMES
#define HFPF_PROBE_FRAMES 8
#endif
constexpr int kProbeFrames = HFPF_PROBE_FRAMES;  // frames of a plan-less batch that go ahead of the rest to measure the per-brick demand
constexpr int kFrameSlots = 8;  // uploads run ahead of the kernels by up to this many frames

}  // namespace

struct hfpf_handle {
    std::mutex mtx;
    hfpf_config cfg;
    GridParams g;
    Tables t;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // host-frame uploads, overlapped with the kernels of earlier frames
    // ... of ring slots 1, 2, ... modulo n_copy_streams: uploads in flight side by side keep the link busy across the gap between two copies
    // of one stream (HFPF_COPY_STREAMS=1..4)
    hipStream_t copy_more[3] = {nullptr, nullptr, nullptr};
    int n_copy_streams = 2;
    std::string err;
    std::vector<void*> allocs;
    uint64_t device_bytes = 0;
    size_t dir_entries = 0;
    uint64_t n_slots = 0;
    uint64_t max_touched = 0;

    // host mirrors
    bool dirty = false;
    uint64_t n_linked[kLogRegions] = {0};  // per log region: entries already chained
    unsigned long long* h_log_ctr = nullptr;  // pinned mirror of the region counters
    int integrate_grid = 1536;
    int upd_shape_forced = -1;  // HFPF_UPD_SHAPE
    bool upd_wide = false;      // the 352-slot record table overflowed in this session: k_update_cells takes the 512-slot shape
    bool trace_shape = false;   // HFPF_TRACE_SHAPE=1: one stderr line per pick_update_shape() window
    unsigned long long upd_miss_seen = 0, upd_member_seen = 0;
    uint32_t launch_seq = 0;  // integrate launches so far (rotates the log append regions)
    uint64_t frames_integrated = 0;
    bool normals_possible = false;   // a clean pass has run since the last clear (the host mirror of C_NORMALS may lag behind a no-wait pass)
    bool stream_replay = true;       // HFPF_STREAM_REPLAY=0: every replay walks the chains
    float bin_slack = 2.0f;          // planned capacity of a bin region = the brick's demand in the previous launch x this (HFPF_BIN_SLACK)
    bool clean_small_nowait = true;  // small clean passes run without a mid-pass read-back (HFPF_CLEAN_NOWAIT=0 restores it)
    uint64_t gate_done = 0; // occ_list entries already examined by a gate pass
    // A capacity / HIP / collective error in the middle of a clean pass leaves the tables half updated: the handle then refuses
    // further work (HFPF_ERR_STATE) until hfpf_clear, instead of silently losing candidates on a retry.
    bool poisoned = false;
    std::string poison_msg;
    bool pend_valid = false;
    uint64_t direct_linked = 0;  // points buffered by k_integrate's direct form (C_BUFFERED) that k_link_log has already chained // pend_a holds C_PEND cells that a gate pass examined and left without a normal
    DevBuf pend_a, pend_b;
    uint64_t clean_passes = 0;
    uint32_t next_frame_id = 0;

    StageSlot stage[kStageSlots];
    int stage_next = 0;
    FrameSlot fslot[kFrameSlots];
    // Device side of the host-frame path: ONE allocation of kFrameSlots slots, ring_cap bytes apart, so that consecutive slots
    // form a batch k_integrate can take in one launch (frame_stride = ring_cap).  Frames that arrive while the engine's stream
    // is still busy with earlier ones are uploaded at once but handed to the kernels together (up to host_batch of them): one
    // bin plan, one k_integrate, one pass of the per-brick kernels for the lot instead of one each.  A frame that finds the
    // stream idle is launched immediately, so a sensor slower than the engine sees no added latency.
    void* ring_d = nullptr;
    size_t ring_cap = 0;
    uint32_t pend_n = 0, pend_first = 0;  // uploaded, not yet launched: slots [pend_first, pend_first + pend_n)
    uint32_t pend_pts = 0;
    uint32_t pend_lay[5] = {0, 0, 0, 0, 0};  // point_step, off_x, off_y, off_z, off_rgb of the pending frames
    double pend_pose[kFrameSlots * 12];
    int host_batch = 4;                   // HFPF_HOST_BATCH (1 = every frame launches on its own)
    hipEvent_t busy_ev = nullptr;         // recorded behind the last launch of the host-frame path
    bool busy_pending = false;
    bool update_cells = true;  // k_update_cells (cell-sorted form); HFPF_UPDATE_FORM=points: k_update (per-point form), A/B and tests
    void* xfer_pin[2] = {nullptr, nullptr};  // pinned staging of extract's row download (two chunks in flight)
    hipEvent_t xfer_ev[2] = {nullptr, nullptr};
    StagePool* stage_pool = nullptr;  // created by the first large bounce copy (HFPF_STAGE_THREADS helpers, default 4; 0 = none)
    int stage_threads = -1;
    int fslot_next = 0;

    // scratch
    DevBuf sort_tmp, keys_a, keys_b, vals_a, vals_b, rows_dev, probe_a, probe_b, probe_c, probe_d, probe_e, probe_f;
    unsigned long long* h_ctr = nullptr;  // pinned mirror of the counters
    unsigned long long* mbox = nullptr;   // coherent pinned mailbox k_publish_counters writes (HFPF_MAILBOX=0: blit copies + synchronize)
    unsigned long long mbox_seq = 0;
    unsigned long long pub_seq = 0;  // sequence number of a publish enqueued behind the last integrate call and still current (0: none)

    // two-pass (binned) dependant update (default; HFPF_FLAG_DIRECT_UPDATE switches it off)
    bool binned = false;
    bool bin_have_hist = false;   // bin_fill holds the demand of the previous launch
    bool bin_from_probe = false;  // ... and that launch was the dry run of a session's first frames
    double bin_prev_points = 0;   // points presented by that launch (to scale the plan)
    uint64_t bin_pool = 0;        // entries in bin_pt
    uint64_t n_bricks_known = 0;  // bricks allocated at the last counter read-back
    uint64_t n_bricks_before = 0; // ... and at the read-back before the count last changed
    bool bin_spare = true;        // HFPF_BIN_SPARE=0: no bin regions for bricks the launch discovers
    float test_bin_scale = 1.f;   // tests only (HFPF_TEST_BIN_SCALE): shrinks the planned bin regions so that they overflow into the direct forms
    DevBuf bin_pt_buf, bin_rgb_buf, bin_sums;
    DevBuf ovf_pt_buf, ovf_aux_buf;  // overflow list of one integrate launch (points that found no room in a bin)

    // multi-GPU: RCCL is resolved at run time so a single-GPU user needs no librccl
    bool dist_on = false;
    int rank = 0, world = 1;
    void* rccl_lib = nullptr;
    void* comm = nullptr;  // ncclComm_t
    uint64_t occ_exported = 0;  // occ_list entries already exchanged
    uint64_t frames_exported = 0, frames_seen = 0;  // frame_list entries already exchanged / as of the last export
    uint64_t ex_cap_records = 0;  // records the exchange buffers hold per rank; kept EQUAL on every rank (same initial value, same growth rule)
    DevBuf ex_send, ex_recv, ex_counts, stats_total;
    unsigned long long* h_counts = nullptr;  // pinned, world entries

    // kernel timing
    bool timing = false;
    bool timing_detail = false;  // hfpf_kernel_timing(h, 2): also one event pair per kernel of an integrate call (ids 2..4)
    std::vector<hipEvent_t> ev_detail;  // 4 events per call: before k_integrate, after it, after k_update*, after k_buffer
    std::vector<uint8_t> ev_detail_ran;  // per call: bit k = the kernel between events k and k + 1 was launched
    double t_detail_ms[3] = {0, 0, 0};
    uint64_t n_detail[3] = {0, 0, 0};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending_clean;
    double t_clean_ms = 0;
    uint64_t n_clean_timed = 0;
    std::vector<hipEvent_t> ev_free;
    double t_integrate_ms = 0;
    uint64_t n_integrate_launches = 0;
};

namespace {

int fail(hfpf_handle* h, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_error = buf;
    return code;
}

int check_usable(hfpf_handle* h)
{
    if (!h->poisoned) return HFPF_OK;
    return fail(h, HFPF_ERR_STATE, "handle failed earlier (%s); hfpf_clear resets it", h->poison_msg.c_str());
}
int poison_on_error(hfpf_handle* h, int rc)
{
    if ((rc == HFPF_ERR_CAPACITY || rc == HFPF_ERR_HIP || rc == HFPF_ERR_DIST) && !h->poisoned) {
        h->poisoned = true;
        h->poison_msg = h->err;
    }
    return rc;
}

#define HIPCHK(h, call)                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) return fail(h, HFPF_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int dev_alloc(hfpf_handle* h, T** out, uint64_t count, int memset_byte = 0, bool do_memset = true)
{
    void* p = nullptr;
    const size_t bytes = std::max<uint64_t>(count, 1) * sizeof(T);
    HIPCHK(h, hipMalloc(&p, bytes));
    h->allocs.push_back(p);
    h->device_bytes += bytes;
    if (do_memset) HIPCHK(h, hipMemsetAsync(p, memset_byte, bytes, h->stream));
    *out = (T*)p;
    return HFPF_OK;
}

// Entries the bin pool of a launch of `pts` points needs: every brick has two regions, each planned for the brick's whole demand of
// the previous launch x slack (x 1.5 at least when the plan comes from the dry run's sample) + 64.
uint64_t bin_pool_entries(const hfpf_handle* h, uint64_t pts, uint64_t bricks)
{
    const double per_point = 2.0 * std::max(1.5, (double)h->bin_slack) + 0.125;
    return (uint64_t)(per_point * (double)pts) + 128ull * (bricks + 1);
}

hipError_t sync_copy_streams(hfpf_handle* h)
{
    hipError_t e = hipStreamSynchronize(h->copy_stream);
    for (hipStream_t cs : h->copy_more)
        if (cs && e == hipSuccess) e = hipStreamSynchronize(cs);
    return e;
}

int scratch(hfpf_handle* h, DevBuf& b, size_t bytes)
{
    if (b.bytes >= bytes && b.p) return HFPF_OK;
    if (b.p) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipFree(b.p));
        h->device_bytes -= b.bytes;
        b.p = nullptr;
        b.bytes = 0;
    }
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    HIPCHK(h, hipMalloc(&b.p, want));
    b.bytes = want;
    h->device_bytes += want;
    return HFPF_OK;
}

inline unsigned blocks_for(uint64_t n, unsigned bs) { return (unsigned)std::max<uint64_t>(1, (n + bs - 1) / bs); }

__global__ void k_set_ctr(unsigned long long* ctr, int idx, unsigned long long v) { ctr[idx] = v; }
// up to three counters in one launch (the clean pass resets them in groups; a launch costs ~5 us even for one thread)
__global__ void k_set_ctr3(unsigned long long* ctr, int i0, unsigned long long v0, int i1, unsigned long long v1, int i2, unsigned long long v2)
{
    ctr[i0] = v0;
    ctr[i1] = v1;
    if (i2 >= 0) ctr[i2] = v2;
}

// Counter read-back through a mailbox in coherent pinned host memory: one small kernel copies the counters (and the four
// used words of every region-counter line) over the link and then publishes a sequence number with system-scope release; the
// host spins on that number.  Against two blit copies + hipStreamSynchronize this saves ~20 us per read-back (the interrupt
// and wake-up of the synchronize), and a clean pass needs two of them with the GPU idle meanwhile.  Stream order makes the
// arrival of the number equivalent to a synchronize for everything enqueued before it.
constexpr int kMboxLogWords = 5;  // words 0..4 of each 16-word region-counter line are in use
constexpr int kMboxWords = C_COUNT + kLogRegions * kMboxLogWords;  // + the sequence number in its own 64-byte line
__global__ __launch_bounds__(256) void k_publish_counters(const unsigned long long* __restrict__ ctr, const unsigned long long* __restrict__ log_ctr,
                                                          unsigned long long* mbox, unsigned long long seq)
{
    const unsigned i = threadIdx.x;
    if (i < (unsigned)C_COUNT) mbox[i] = ctr[i];
    for (unsigned w = i; w < (unsigned)(kLogRegions * kMboxLogWords); w += blockDim.x) mbox[C_COUNT + w] = log_ctr[(w / kMboxLogWords) * 16 + (w % kMboxLogWords)];
    __threadfence_system();
    __syncthreads();
    if (i == 0) __hip_atomic_store(&mbox[kMboxWords + 7], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int read_counters(hfpf_handle* h)
{
    if (h->mbox) {
        // An integrate call ends with a publish of its own (nothing has touched the counters since): the snapshot is already on
        // its way, so the host only waits -- no launch of its own behind a stream that has just drained.
        unsigned long long seq = h->pub_seq;
        h->pub_seq = 0;
        if (seq == 0) {
            seq = ++h->mbox_seq;
            k_publish_counters<<<1, 256, 0, h->stream>>>(h->t.ctr, h->t.log_ctr, h->mbox, seq);
            HIPCHK(h, hipGetLastError());
        }
        volatile unsigned long long* flag = h->mbox + kMboxWords + 7;
        for (uint64_t spins = 1;; spins++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
            __builtin_ia32_pause();
            if ((spins & 0xFFFF) == 0) {  // every ~1 ms: a failed stream would never publish
                const hipError_t q = hipStreamQuery(h->stream);
                if (q != hipSuccess && q != hipErrorNotReady) HIPCHK(h, q);
                if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
                    return fail(h, HFPF_ERR_HIP, "counter mailbox: the stream drained without publishing sequence %llu", seq);
            }
        }
        memcpy(h->h_ctr, h->mbox, C_COUNT * sizeof(unsigned long long));
        for (int r = 0; r < kLogRegions; r++)
            for (int w = 0; w < kMboxLogWords; w++) h->h_log_ctr[r * 16 + w] = h->mbox[C_COUNT + r * kMboxLogWords + w];
    } else {
        HIPCHK(h, hipMemcpyAsync(h->h_ctr, h->t.ctr, C_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->h_log_ctr, h->t.log_ctr, kLogRegions * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    unsigned long long total = 0;
    for (int r = 0; r < kLogRegions; r++) total += std::min<unsigned long long>(h->h_log_ctr[r * 16], h->t.log_region_cap);
    h->h_ctr[C_LOG] = total;
    unsigned long long replayed = 0;  // striped diagnostic counter of k_replay (word 1 of every log_ctr line)
    for (int r = 0; r < kLogRegions; r++) replayed += h->h_log_ctr[r * 16 + 1];
    h->h_ctr[C_REPLAY_MEMBER] = replayed;
    unsigned long long upd_tested = 0, upd_member = 0;  // k_update's striped pair counters (words 2, 3); k_integrate's direct path adds to ctr[]
    for (int r = 0; r < kLogRegions; r++) {
        upd_tested += h->h_log_ctr[r * 16 + 2];
        upd_member += h->h_log_ctr[r * 16 + 3];
    }
    h->h_ctr[C_DEP_TESTED] += upd_tested;
    h->h_ctr[C_DEP_MEMBER] += upd_member;
    unsigned long long single = 0;  // touched cells of the last dependant-table update that lie in single-run bricks (word 4)
    for (int r = 0; r < kLogRegions; r++) single += h->h_log_ctr[r * 16 + 4];
    h->h_ctr[C_TOUCHED_SINGLE] = single;
    {
        const uint64_t nbk = std::min<uint64_t>(h->h_ctr[C_BRICKS], h->t.max_bricks);
        if (nbk != h->n_bricks_known) h->n_bricks_before = h->n_bricks_known;  // how fast the session discovers bricks (spare bin regions)
        h->n_bricks_known = nbk;
    }
    return HFPF_OK;
}

int check_device_errors(hfpf_handle* h)
{
    const unsigned long long e = h->h_ctr[C_ERR];
    if (!e) return HFPF_OK;
    std::string what;
    if (e & E_BRICKS) what += " brick pool (max_bricks)";
    if (e & E_LOG) what += " point log (max_log_points)";
    if (e & E_OCC) what += " occupied list (max_normals*4)";
    if (e & E_NORMALS) what += " normal records (max_normals)";
    if (e & E_REG) what += " registrations (max_normals*7)";
    if (e & E_DEP) what += " dependant table";
    if (e & E_SPIN) what += " brick-claim spin bound";
    if (e & E_DEPCNT) what += " more than 65535 dependants on one cell";
    if (e & E_FRAME) what += " frame id >= max_frames";
    if (e & E_OVF) what += " integrate overflow list";
    if (e & E_CHAIN) what += " point-log chain / run record out of range (internal)";
    return fail(h, HFPF_ERR_CAPACITY, "device pool overflow:%s", what.c_str());
}

// Power-of-two scale that keeps one contribution of magnitude < bound below 2^27 (stats.hpp).
float stat_scale_for(double bound) { return (float)std::ldexp(1.0, 26 - (int)std::floor(std::log2(bound))); }

int setup_params(hfpf_handle* h)
{
    const hfpf_config& c = h->cfg;
    GridParams& g = h->g;
    if (!(c.resolution > 0.f)) return fail(h, HFPF_ERR_BAD_CONFIG, "resolution must be > 0");
    for (int a = 0; a < 3; a++)
        if (!(c.bbox[2 * a + 1] > c.bbox[2 * a])) return fail(h, HFPF_ERR_BAD_CONFIG, "bounding_box axis %d: max <= min", a);
    if (c.k != 2) return fail(h, HFPF_ERR_BAD_CONFIG, "k must be 2 (the reference probes exactly 125 neighbours, OccupancyGrid.hpp:334)");
    if (c.K < 0 || c.K > 16) return fail(h, HFPF_ERR_BAD_CONFIG, "K out of range [0,16]");
    if (!(c.cylinder_radius > 0) || !(c.ball_radius > 0)) return fail(h, HFPF_ERR_BAD_CONFIG, "radii must be > 0");
    g.res = (double)c.resolution;  // float -> double, grid.hpp:614-619
    g.inv_res = 1.0 / g.res;
    for (int a = 0; a < 3; a++) {
        g.min[a] = c.bbox[2 * a];
        g.max[a] = c.bbox[2 * a + 1];
        const double d = (g.max[a] - g.min[a]) / g.res;  // grid.hpp:623-625 (int truncation)
        if (!(d < 2097151.0)) return fail(h, HFPF_ERR_BAD_CONFIG, "axis %d: more than 2^21 cells", a);
        g.dim[a] = (int32_t)d;
        g.bdim[a] = (g.dim[a] + 1 + 7) / 8;  // storage is dim+1 cells, grid.hpp:626
    }
    {
        auto bits_for = [](int32_t dim) {  // bits that hold 0..dim (storage is dim+1 cells per axis, grid.hpp:626)
            uint32_t b = 1;
            while ((1ll << b) <= (long long)dim) b++;
            return b;
        };
        g.key_sy = bits_for(g.dim[2]);
        g.key_sx = g.key_sy + bits_for(g.dim[1]);
        g.key_bits = g.key_sx + bits_for(g.dim[0]);
    }
    g.zclip_min = c.z_clip_min;
    g.zclip_max = c.z_clip_max;
    {   // float neighbours of the double clip constants (geometry.hpp, GridParams::bb_lo): compares in f32 with identical decisions
        auto float_at_or_above = [](double m) {
            float f = (float)m;
            if ((double)f < m) f = nextafterf(f, INFINITY);
            return f;
        };
        auto float_at_or_below = [](double m) {
            float f = (float)m;
            if ((double)f > m) f = nextafterf(f, -INFINITY);
            return f;
        };
        for (int a = 0; a < 3; a++) {
            g.bb_hi[a] = float_at_or_above(g.max[a]);
            g.bb_lo[a] = float_at_or_below(g.min[a]);
        }
        g.zc_hi = float_at_or_above(g.zclip_max);
        g.zc_lo = float_at_or_below(g.zclip_min);
    }
    g.cyl_r = c.cylinder_radius;
    g.ball_r = (float)c.ball_radius;
    g.K = c.K;
    g.gate = c.gate;
    g.cov_shifted = (c.flags & HFPF_FLAG_PCL_SHIFTED_COV) ? 1 : 0;
    // s = 0.5 + t / (2 r) with t the point's offset along the normal from the cell centre: a point updates a voxel only from
    // a cell on that voxel's line, so |t| <= K*res + sqrt(3)*res
    const double Bs = 0.5 + ((double)c.K + 2.0) * g.res / (2.0 * c.ball_radius);
    if (!(Bs < 1.0e6)) return fail(h, HFPF_ERR_BAD_CONFIG, "ball_radius too small for this resolution and K");
    const double Bm = HFPF_CENTERED_MOMENTS ? Bs - 0.5 : Bs;  // bound of the accumulated variable (u = s - 0.5 or s, stats.hpp)
    g.fs_scale = stat_scale_for(Bm);
    g.fss_scale = stat_scale_for(Bm * Bm);
    g.fd_scale = stat_scale_for(g.cyl_r);
    g.fdd_scale = stat_scale_for(g.cyl_r * g.cyl_r);
    {   // largest f32 u with (double)sqrtf(u) < cyl_r (sqrtf is correctly rounded on the host: IEEE 754), by bisection on the bits
        uint32_t lo_b = 0, hi_b = 0x7F800000u;  // invariant: f(lo) passes, f(hi) fails (sqrt(+inf) = inf)
        while (hi_b - lo_b > 1) {
            const uint32_t mid = lo_b + (hi_b - lo_b) / 2;
            float u;
            memcpy(&u, &mid, 4);
            if ((double)sqrtf(u) < g.cyl_r) lo_b = mid;
            else hi_b = mid;
        }
        memcpy(&g.d2_max, &lo_b, 4);
    }
    const double dir_entries = (double)g.bdim[0] * (double)g.bdim[1] * (double)g.bdim[2];
    if (dir_entries > 4.0e9) return fail(h, HFPF_ERR_BAD_CONFIG, "brick directory too large (%.3g entries)", dir_entries);
    h->dir_entries = (size_t)dir_entries;
    return HFPF_OK;
}

// bricks_used / normals_used: how far the session got into the brick pool and the record table (hfpf_clear knows; everything at
// create).  Per-cell and per-brick arrays are indexed by brick id, records by record id, and both are handed out in sequence, so a
// reset only has to cover the prefix the session used: the bench's handle is sized for 300,000 bricks and uses 22,000 -- 7 GB of
// resets become 0.6 GB.
int reset_state(hfpf_handle* h, uint64_t bricks_used = ~0ull, uint64_t normals_used = ~0ull)
{
    Tables& t = h->t;
    hipStream_t s = h->stream;
    const uint64_t nb = std::min<uint64_t>(bricks_used, t.max_bricks);                     // ids 1..nb (+ the unused id 0)
    const uint64_t nsl = std::min<uint64_t>(h->n_slots, (nb + 1) * (uint64_t)kBrickCells);  // their cells
    const uint64_t nn = std::min<uint64_t>(normals_used, t.max_normals);
    HIPCHK(h, hipMemsetAsync(t.dir, 0, h->dir_entries * 4, s));
    HIPCHK(h, hipMemsetAsync(t.info, 0, nsl * 8, s));
    HIPCHK(h, hipMemsetAsync(t.first_frame, 0xFF, nsl * 4, s));
    HIPCHK(h, hipMemsetAsync(t.buf_head, 0, nsl * 4 * kChains, s));
    HIPCHK(h, hipMemsetAsync(t.stat_id, 0, nsl * 4, s));
    HIPCHK(h, hipMemsetAsync(t.pre_dep, 0, nsl * 4, s));
    HIPCHK(h, hipMemsetAsync(t.dep_tmp, 0, nsl * 4, s));
    HIPCHK(h, hipMemsetAsync(t.occ_mask, 0, (nb + 1) * 8 * 8, s));
    HIPCHK(h, hipMemsetAsync(t.stats, 0, (nn + 1) * kStatWords * 8, s));
    HIPCHK(h, hipMemsetAsync(t.nd_mask, 0, (nb + 1) * 8 * 2 * 8, s));
    HIPCHK(h, hipMemsetAsync(t.ctr, 0, C_COUNT * 8, s));
    HIPCHK(h, hipMemsetAsync(t.log_ctr, 0, kLogRegions * 16 * 8, s));
    HIPCHK(h, hipMemsetAsync(t.bin_fill, 0, 2 * (t.max_bricks + 2) * 4, s));
    HIPCHK(h, hipMemsetAsync(t.bin_off, 0, 2 * (t.max_bricks + 2) * 4, s));
    HIPCHK(h, hipMemsetAsync(t.bin_capb, 0, 2 * (t.max_bricks + 2) * 4, s));
    HIPCHK(h, hipMemsetAsync(t.run_cnt, 0, (t.max_bricks + 2) * 4, s));
    h->bin_have_hist = false;
    h->normals_possible = false;
    h->pend_n = 0;
    h->pub_seq = 0;
    h->upd_miss_seen = h->upd_member_seen = 0;  // (upd_wide stays: the next session on this handle fuses the same kind of scene)
    h->direct_linked = 0;
    h->n_bricks_known = 0;
    h->n_bricks_before = 0;
    if (h->h_ctr) memset(h->h_ctr, 0, C_COUNT * sizeof(unsigned long long));  // host mirror follows the device counters
    h->dirty = false;
    for (int r = 0; r < kLogRegions; r++) h->n_linked[r] = 0;
    h->gate_done = 0;
    h->pend_valid = false;
    h->poisoned = false;
    h->poison_msg.clear();
    h->occ_exported = 0;
    h->frames_exported = h->frames_seen = 0;
    h->next_frame_id = 0;
    return HFPF_OK;
}

int alloc_tables(hfpf_handle* h)
{
    hfpf_config& c = h->cfg;
    Tables& t = h->t;
    memset(&t, 0, sizeof t);
    if (c.max_bricks == 0) c.max_bricks = 131072;
    if (c.max_log_points == 0) c.max_log_points = 64ull << 20;
    if (c.max_normals == 0) c.max_normals = 8ull << 20;
    if (c.max_frames == 0) c.max_frames = 65536;
    if (c.max_bricks > 8388606ull) return fail(h, HFPF_ERR_BAD_CONFIG, "max_bricks must be < 2^23");
    c.max_log_points = std::max<uint64_t>(c.max_log_points, 64 * kLogRegions);
    if (c.max_log_points > 0x7FFFFFF0ull) return fail(h, HFPF_ERR_BAD_CONFIG, "max_log_points must be < 2^31 (bit 31 of a log link marks an unchained entry)");
    if (c.max_frames > (1ull << 23)) return fail(h, HFPF_ERR_BAD_CONFIG, "max_frames must be <= 2^23 (a parked point carries its frame id in 23 bits)");
    if (c.max_normals > 4294967294ull) return fail(h, HFPF_ERR_BAD_CONFIG, "max_normals must be < 2^32-1");
    t.max_bricks = c.max_bricks;
    t.max_log = c.max_log_points;
    t.max_normals = c.max_normals;
    t.max_occ = c.max_normals * 4;
    t.max_reg = c.max_normals * (2ull * (uint64_t)c.K + 1ull);
    t.max_dep = 2 * t.max_reg;  // room for the lists the incremental update relocates
    if (t.max_dep > 0xFFFFFFF0ull) return fail(h, HFPF_ERR_BAD_CONFIG, "max_normals too large: the dependant table must stay below 2^32 entries");
    t.max_frames = c.max_frames;
    h->n_slots = (t.max_bricks + 1) * (uint64_t)kBrickCells;
    h->max_touched = t.max_reg;
    int rc;
#define ALLOC(field, count, ...)                                         \
    if ((rc = dev_alloc(h, &t.field, (count), ##__VA_ARGS__)) != HFPF_OK) return rc;
    ALLOC(dir, h->dir_entries, 0, false);
    ALLOC(brick_lin, t.max_bricks + 1);
    ALLOC(info, h->n_slots, 0, false);
    ALLOC(first_frame, h->n_slots, 0, false);
    ALLOC(buf_head, h->n_slots * kChains, 0, false);
    ALLOC(stat_id, h->n_slots, 0, false);
    ALLOC(pre_dep, h->n_slots, 0, false);
    ALLOC(dep_tmp, h->n_slots, 0, false);
    ALLOC(occ_mask, (t.max_bricks + 1) * 8, 0, false);
    ALLOC(log_pt, t.max_log + 1, 0, false);
    if (c.flags & HFPF_FLAG_FUSE_COLOR) { ALLOC(log_rgb, t.max_log + 1, 0, false); }
    ALLOC(occ_list, t.max_occ, 0, false);
    ALLOC(nv_key, t.max_normals + 1, 0, false);
    ALLOC(nv_slot, t.max_normals + 1, 0, false);
    ALLOC(nv_c, 3 * (t.max_normals + 1), 0, false);
    ALLOC(nv_n, 3 * (t.max_normals + 1), 0, false);
    ALLOC(stats, (t.max_normals + 1) * kStatWords, 0, false);
    t.color = (c.flags & HFPF_FLAG_FUSE_COLOR) ? 1u : 0u;
    {
        const char* ts = getenv("HFPF_TEST_TABLE_SKIP");
        t.test_table_skip = (ts && ts[0] == '1') ? 1u : 0u;
    }
    ALLOC(nd_mask, (t.max_bricks + 1) * 8 * 2, 0, false);
    ALLOC(reg_occ, t.max_reg, 0, false);
    ALLOC(dep, t.max_dep + t.max_normals + 1, 0, false);  // dep[] and, behind it, the records' lines (one 32-byte entry each): one entry space
    t.nv_line = reinterpret_cast<float4*>(t.dep + t.max_dep);
    ALLOC(prereg_list, t.max_reg, 0, false);
    ALLOC(touched_list, h->max_touched, 0, false);
    ALLOC(run_start, t.max_bricks + 2, 0, false);
    ALLOC(run_len, t.max_bricks + 2, 0, false);
    ALLOC(run_cnt, t.max_bricks + 2);
    ALLOC(cand_key, t.max_occ, 0, false);
    ALLOC(frame_vp, 3 * t.max_frames);
    ALLOC(frame_list, t.max_frames, 0, false);
    ALLOC(ctr, C_COUNT);
    ALLOC(log_ctr, kLogRegions * 16);
    ALLOC(bin_fill, 2 * (t.max_bricks + 2));
    ALLOC(bin_off, 2 * (t.max_bricks + 2));
    ALLOC(bin_capb, 2 * (t.max_bricks + 2));
    h->binned = (c.flags & HFPF_FLAG_DIRECT_UPDATE) == 0;
#undef ALLOC
    t.log_region_cap = t.max_log / kLogRegions;
    t.ent_base = t.dep;
    t.ent_dep_first = 0;
    t.ent_nv_first = t.max_dep;  // (max_dep < 2^32 and max_normals < 2^32: far inside the 40-bit entry index of k_update_cells)
    return reset_state(h);
}

// (rocPRIM sorts fewer than a million keys by block sort + log2(n / 1024) merge launches -- nine launches, 57 us, for the 74 K candidate
// keys of a steady clean pass.  Forcing its radix path for small inputs, radix_sort_config<..., 8192>, was measured in round 4: a
// histogram, a scan and per 8-bit digit two buffer fills and one sweep, fifteen launches and ~100 us for the same keys: the merge path stays.)
// What does help: blocks of 4,096 sorted keys instead of 1,024 -- two merge launches less for those 74 K keys (clean passes 3.66 -> 3.57 ms
// per 1000-frame run; 8,192-key blocks: 3.64).
#ifndef HFPF_SORT_BLOCK_ITEMS
#define HFPF_SORT_BLOCK_ITEMS 4
#endif
using sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512, 1024, HFPF_SORT_BLOCK_ITEMS>, rocprim::default_config>;

// Cell keys: only the low GridParams::key_bits bits are significant (an all-ones sentinel still sorts behind every valid key:
// a valid cell has x < dim <= 2^bits_x - 1, so its key is never all ones).
int sort_keys_u64(hfpf_handle* h, uint64_t* in, uint64_t* out, uint64_t n, unsigned bits = 0)
{
    const unsigned kb = bits ? bits : h->g.key_bits;
    size_t bytes = 0;
    HIPCHK(h, rocprim::radix_sort_keys<sort_config>(nullptr, bytes, in, out, (size_t)n, 0, kb, h->stream));
    int rc = scratch(h, h->sort_tmp, bytes);
    if (rc) return rc;
    bytes = h->sort_tmp.bytes;
    HIPCHK(h, rocprim::radix_sort_keys<sort_config>(h->sort_tmp.p, bytes, in, out, (size_t)n, 0, kb, h->stream));
    return HFPF_OK;
}

int sort_keys_u32(hfpf_handle* h, uint32_t* in, uint32_t* out, uint64_t n, unsigned bits = 32)
{
    size_t bytes = 0;
    HIPCHK(h, rocprim::radix_sort_keys<sort_config>(nullptr, bytes, in, out, (size_t)n, 0, bits, h->stream));
    int rc = scratch(h, h->sort_tmp, bytes);
    if (rc) return rc;
    bytes = h->sort_tmp.bytes;
    HIPCHK(h, rocprim::radix_sort_keys<sort_config>(h->sort_tmp.p, bytes, in, out, (size_t)n, 0, bits, h->stream));
    return HFPF_OK;
}

int sort_pairs_u64(hfpf_handle* h, uint64_t* kin, uint64_t* kout, uint32_t* vin, uint32_t* vout, uint64_t n)
{
    const unsigned kb = h->g.key_bits;
    size_t bytes = 0;
    HIPCHK(h, rocprim::radix_sort_pairs<sort_config>(nullptr, bytes, kin, kout, vin, vout, (size_t)n, 0, kb, h->stream));
    int rc = scratch(h, h->sort_tmp, bytes);
    if (rc) return rc;
    bytes = h->sort_tmp.bytes;
    HIPCHK(h, rocprim::radix_sort_pairs<sort_config>(h->sort_tmp.p, bytes, kin, kout, vin, vout, (size_t)n, 0, kb, h->stream));
    return HFPF_OK;
}

int acquire_stage(hfpf_handle* h, uint32_t n_frames, StageSlot** out)
{
    StageSlot& s = h->stage[h->stage_next];
    h->stage_next = (h->stage_next + 1) % kStageSlots;
    if (s.pending) {
        HIPCHK(h, hipEventSynchronize(s.done));
        s.pending = false;
    }
    if (!s.done) HIPCHK(h, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    if (s.cap < n_frames) {
        if (s.h_pose) {
            HIPCHK(h, hipHostFree(s.h_pose));
            HIPCHK(h, hipFree(s.d_pose));
            s.h_pose = nullptr, s.d_pose = nullptr, s.cap = 0;
        }
        const uint32_t cap = std::max<uint32_t>(n_frames, 64);
        const size_t bytes = (size_t)cap * (12 * sizeof(double) + sizeof(uint32_t));
        HIPCHK(h, hipHostMalloc((void**)&s.h_pose, bytes, hipHostMallocDefault));
        HIPCHK(h, hipMalloc((void**)&s.d_pose, bytes));
        s.cap = cap;
    }
    s.h_ids = reinterpret_cast<uint32_t*>(s.h_pose + 12 * (size_t)n_frames);
    s.d_ids = reinterpret_cast<uint32_t*>(s.d_pose + 12 * (size_t)n_frames);
    *out = &s;
    return HFPF_OK;
}

// Which instantiation of k_update_cells a launch takes (kernels.hpp, UpdShape): 0 dense, 1 wide table.  HFPF_UPD_SHAPE=0|1 forces
// one.  Otherwise the dense shape, until the items that found no slot in its table exceed one in 500 member pairs (counted by the
// kernel, seen by the host at its counter read-backs -- every clean pass); then the wide table for the rest of the session.
// The streaming replay of a clean pass is the same kernel with the same table and counts its misses into the same word, so its
// members belong in the denominator too: a caller that reads the counters between a clean pass and the next integrate call
// (hfpf_get_counters, hfpf_get_kernel_time) makes a window that holds the replay alone -- a few hundred misses against no update
// member at all, which used to switch the session to the wide shape (bench.py's per-kernel pass did exactly that).
int pick_update_shape(hfpf_handle* h, double points, uint32_t nb)
{
    (void)points;
    (void)nb;
    if (h->upd_shape_forced >= 0) return h->upd_shape_forced;
    const unsigned long long miss = h->h_ctr[C_TABLE_MISS], member = h->h_ctr[C_DEP_MEMBER] + h->h_ctr[C_REPLAY_MEMBER];
    if (h->trace_shape)
        fprintf(stderr, "hfpf: update shape window: %llu table misses, %llu members (%llu of them replayed so far), %s\n", miss - std::min(miss, h->upd_miss_seen),
                member - std::min(member, h->upd_member_seen), h->h_ctr[C_REPLAY_MEMBER], h->upd_wide ? "wide" : "dense");
    if (!h->upd_wide && miss > h->upd_miss_seen && (miss - h->upd_miss_seen) * 500ull > member - std::min(member, h->upd_member_seen)) h->upd_wide = true;
    h->upd_miss_seen = miss;
    h->upd_member_seen = member;
    return h->upd_wide ? 1 : 0;
}

int integrate_device_locked(hfpf_handle* h, const void* dev_base, uint32_t n_frames, uint64_t frame_stride, uint32_t n_points,
                            uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t off_rgb, const double* poses,
                            const uint32_t* frame_ids)
{
    if (!dev_base || !poses || n_frames == 0) return fail(h, HFPF_ERR_BAD_ARG, "integrate: null buffer/poses or zero frames");
    if (n_points == 0) return HFPF_OK;
    if ((point_step & 3) || (off_x & 3) || (off_y & 3) || (off_z & 3) || (off_rgb & 3) || ((uintptr_t)dev_base & 3) || (frame_stride & 3))
        return fail(h, HFPF_ERR_BAD_ARG, "integrate: fields must be 4-byte aligned");
    if (std::max(std::max(off_x, off_y), std::max(off_z, off_rgb)) + 4 > point_step)
        return fail(h, HFPF_ERR_BAD_ARG, "integrate: field offset beyond point_step");
    if (n_frames > 65535) return fail(h, HFPF_ERR_BAD_ARG, "integrate: at most 65535 frames per call");
    h->pub_seq = 0;  // kernels are about to be enqueued: a counter snapshot already on its way is no longer the latest
    StageSlot* s = nullptr;
    int rc = acquire_stage(h, n_frames, &s);
    if (rc) return rc;
    memcpy(s->h_pose, poses, (size_t)n_frames * 12 * sizeof(double));
    for (uint32_t f = 0; f < n_frames; f++) {
        const uint32_t id = frame_ids ? frame_ids[f] : h->next_frame_id + f;
        if (id >= h->t.max_frames) return fail(h, HFPF_ERR_CAPACITY, "frame id %u >= max_frames %llu", id, (unsigned long long)h->t.max_frames);
        s->h_ids[f] = id;
    }
    if (!frame_ids) h->next_frame_id += n_frames;
    HIPCHK(h, hipMemcpyAsync(s->d_pose, s->h_pose, (size_t)n_frames * (12 * sizeof(double) + sizeof(uint32_t)), hipMemcpyHostToDevice, h->stream));  // poses + ids

    const FrameLayout lay{point_step, off_x, off_y, off_z, off_rgb};
    const bool packed = point_step == 16 && off_x == 0 && off_y == 4 && off_z == 8 && off_rgb == 12 && ((uintptr_t)dev_base & 15) == 0 &&
                        (frame_stride & 15) == 0;
    const dim3 block(256);
    const uint64_t n_tiles = (uint64_t)blocks_for(n_points, 256) * n_frames;
    // 16x16-pixel tiles when the caller told us the image width and the frame tiles exactly (hfpf_config.frame_width)
    const uint32_t fw = h->cfg.frame_width;
    const uint32_t row_w = (fw >= 16 && fw % 16 == 0 && n_points % fw == 0 && (n_points / fw) % 16 == 0) ? fw : 0u;
    const dim3 grid((unsigned)std::min<uint64_t>(n_tiles, (uint64_t)h->integrate_grid));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing) {
        auto get = [&](hipEvent_t& e) -> hipError_t {
            if (!h->ev_free.empty()) {
                e = h->ev_free.back();
                h->ev_free.pop_back();
                return hipSuccess;
            }
            return hipEventCreate(&e);
        };
        HIPCHK(h, get(e0));
        HIPCHK(h, get(e1));
        HIPCHK(h, hipEventRecord(e0, h->stream));
    }
    const bool color = h->t.color != 0;
    const bool bin = h->binned;
    const uint32_t pre_possible = (h->h_ctr[C_NORMALS] > 0 || h->normals_possible) ? 1u : 0u;  // a clean pass has run: unoccupied cells may carry a dependant
    uint32_t launch_frames = n_frames, probe = 0;
    // frames of the dry run: twice as many for a long batch.  For 150 frames of 640x480 with random poses the first 8 find 46 % of
    // the bricks the batch touches and 16 find 50 % (32: 57 %), and the plan made from the larger sample sends 30 % fewer points
    // through the overflow list (268 K instead of 382 K per 1000-frame pass): whole job +1.5 %; 32 frames add nothing.
    const uint32_t probe_frames = n_frames >= 8u * (uint32_t)kProbeFrames ? 2u * (uint32_t)kProbeFrames : (uint32_t)kProbeFrames;
    if (bin && !h->bin_have_hist && n_frames > probe_frames) {
        // No plan for the per-brick bins yet (first batch of a session): a dry run of the batch's first frames claims their
        // bricks and records the per-region demand, so that the real launch below parks from its first point.  One extra
        // read-back (the brick count), once per session; batches of up to kProbeFrames frames just take the direct forms.
        HIPCHK(h, hipMemsetAsync(h->t.bin_fill, 0, 2 * (h->t.max_bricks + 2) * 4, h->stream));
        HIPCHK(h, hipMemsetAsync(h->t.bin_capb, 0, 2 * (h->t.max_bricks + 2) * 4, h->stream));
        launch_frames = probe_frames;
        probe = 1;
        const uint32_t log_rot = 0;
        const dim3 pgrid((unsigned)std::min<uint64_t>((uint64_t)blocks_for(n_points, 256) * launch_frames, (uint64_t)h->integrate_grid));
#define HFPF_LAUNCH_PROBE(P, C)                                                                                                                    \
    hipLaunchKernelGGL((k_integrate<P, C, true>), pgrid, block, 0, h->stream,                                                                        \
                       IntegrateArgs{h->g, h->t}, (const uint8_t*)dev_base, frame_stride, n_points, launch_frames, lay, (const double*)s->d_pose, \
                       (const uint32_t*)s->d_ids, row_w, log_rot, probe, pre_possible)
        if (packed && !color) HFPF_LAUNCH_PROBE(true, false);
        else if (packed && color) HFPF_LAUNCH_PROBE(true, true);
        else if (!color) HFPF_LAUNCH_PROBE(false, false);
        else HFPF_LAUNCH_PROBE(false, true);
#undef HFPF_LAUNCH_PROBE
        HIPCHK(h, hipGetLastError());
        int rcp = read_counters(h);  // bricks the dry run claimed
        if (rcp) return rcp;
        h->bin_have_hist = true;
        h->bin_from_probe = true;  // the plan of the launch below comes from a sample: more slack per region
        h->bin_prev_points = (double)n_points * probe_frames;
        launch_frames = n_frames;
        probe = 0;
    }
    const uint32_t nb_known = (uint32_t)h->n_bricks_known;
    // a plan = per-brick bin regions sized from the previous launch's demand; without one nothing is parked (direct forms)
    const bool have_plan = bin && h->bin_have_hist && nb_known > 0;
    // Bricks this launch may discover get spare regions of an average brick's size: as many as the session found between its last
    // two counter read-backs (x2), i.e. half the known bricks right after the dry run and a few hundred in the steady state.
    uint32_t spare = 0, spare_cap = 0;
    if (have_plan && h->bin_spare) {
        const uint64_t grown = h->n_bricks_known - std::min(h->n_bricks_known, h->n_bricks_before);
        spare = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2 * grown, 256), std::max<uint64_t>(nb_known / 2, 256));
        spare = (uint32_t)std::min<uint64_t>(spare, h->t.max_bricks - std::min<uint64_t>(h->t.max_bricks, nb_known));
    }
    const uint32_t nb = nb_known + spare;  // bricks the per-brick kernels of this call look at
    if (bin) {
        // pool for this launch's parked points (+25 % plan slack, +64 per brick)
        const uint64_t pts = (uint64_t)n_points * n_frames;
        uint64_t pool = bin_pool_entries(h, pts, nb_known);  // two regions per brick, each sized for the whole brick
        if (spare) {  // an average brick's share of the batch, both regions, within what a 32-bit index still addresses
            spare_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(pts / std::max(1u, nb_known), 64), 1u << 15);
            const uint64_t room = 0xFFFFFFFFull - std::min<uint64_t>(pool, 0xFFFFFFFFull);
            if (2ull * spare * spare_cap > room) spare_cap = (uint32_t)(room / (2ull * spare));
            pool += 2ull * spare * spare_cap;
        }
        if (pool > 0xFFFFFFFFull) return fail(h, HFPF_ERR_BAD_ARG, "integrate: batch too large for the binned update (split the call)");
        if (h->bin_pool < pool) {
            int rc2 = scratch(h, h->bin_pt_buf, pool * sizeof(float4));
            if (rc2) return rc2;
            if (color && (rc2 = scratch(h, h->bin_rgb_buf, pool * 4))) return rc2;
            h->bin_pool = pool;
        }
        h->t.bin_pt = (float4*)h->bin_pt_buf.p;
        h->t.bin_rgb = (uint32_t*)h->bin_rgb_buf.p;
        // the overflow list holds a whole launch: a batch without a plan, or one that looks at a new part of the scene, parks nothing
        if (h->ovf_pt_buf.bytes < pts * sizeof(float4) || h->ovf_aux_buf.bytes < pts * sizeof(uint2)) {
            int rc2 = scratch(h, h->ovf_pt_buf, pts * sizeof(float4));
            if (!rc2) rc2 = scratch(h, h->ovf_aux_buf, pts * sizeof(uint2));
            if (rc2) return rc2;
        }
        h->t.ovf_pt = (float4*)h->ovf_pt_buf.p;
        h->t.ovf_aux = (uint2*)h->ovf_aux_buf.p;
        h->t.ovf_cap = pts;
        if (have_plan) {
            const float scale = (float)((double)pts / std::max(1.0, h->bin_prev_points)) * h->test_bin_scale;
            const uint32_t n_regions = 2u * (nb_known + 1u);  // two per brick: cells with / without a normal
            const uint32_t n_planned = 2u * (nb + 1u);        // ... and the spare ones behind them
            const uint32_t all_regions = 2u * (uint32_t)(h->t.max_bricks + 2);
            int rc2 = scratch(h, h->bin_sums, (size_t)blocks_for(all_regions, kBinPlanTile) * sizeof(uint32_t));
            if (rc2) return rc2;
            hipLaunchKernelGGL(k_bin_plan, dim3(blocks_for(n_planned, kBinPlanTile)), dim3(256), 0, h->stream, h->t, n_regions, n_planned, spare_cap, scale,
                               h->bin_from_probe ? std::max(1.5f, h->bin_slack) : h->bin_slack, (uint32_t*)h->bin_sums.p);
            h->bin_from_probe = false;
            hipLaunchKernelGGL(k_bin_place, dim3(blocks_for(all_regions, kBinPlanTile)), dim3(256), 0, h->stream, h->t, n_planned, all_regions, h->bin_pool,
                               (const uint32_t*)h->bin_sums.p);
        } else {  // no plan yet: no region exists, every lane takes the direct forms, the demand is recorded
            HIPCHK(h, hipMemsetAsync(&h->t.ctr[C_OVF], 0, sizeof(unsigned long long), h->stream));  // (k_bin_place does this where there is a plan)
            HIPCHK(h, hipMemsetAsync(h->t.bin_fill, 0, 2 * (h->t.max_bricks + 2) * 4, h->stream));
            HIPCHK(h, hipMemsetAsync(h->t.bin_capb, 0, 2 * (h->t.max_bricks + 2) * 4, h->stream));
        }
    }
    const uint32_t log_rot = (uint32_t)((h->launch_seq++ * 17u) & (kLogRegions - 1));
    // Detail timing: four events per call and one byte saying which of the three kernels between them ran.  A call that leaves
    // early gives its events back, so the records stay aligned.
    struct DetailGuard {
        hfpf_handle* h;
        size_t first;
        bool done = false;
        ~DetailGuard()
        {
            if (done) return;
            while (h->ev_detail.size() > first) {
                h->ev_free.push_back(h->ev_detail.back());
                h->ev_detail.pop_back();
            }
        }
    } detail_guard{h, h->ev_detail.size()};
    uint8_t detail_ran = 1;  // k_integrate (+ its overflow kernel) always runs
    auto detail_mark = [&]() -> hipError_t {  // per-kernel boundaries of this call (detail timing only)
        if (!h->timing_detail) return hipSuccess;
        hipEvent_t e = nullptr;
        if (!h->ev_free.empty()) {
            e = h->ev_free.back();
            h->ev_free.pop_back();
        } else if (hipError_t r = hipEventCreate(&e)) {
            return r;
        }
        h->ev_detail.push_back(e);
        return hipEventRecord(e, h->stream);
    };
    HIPCHK(h, detail_mark());
#define HFPF_LAUNCH_INTEGRATE(P, C, B)                                                                                                              \
    hipLaunchKernelGGL((k_integrate<P, C, B>), grid, block, 0, h->stream,                                                                           \
                       IntegrateArgs{h->g, h->t}, (const uint8_t*)dev_base, frame_stride, n_points, launch_frames, lay, (const double*)s->d_pose, \
                       (const uint32_t*)s->d_ids, row_w, log_rot, probe, pre_possible)
    if (!bin) {
        if (packed && !color) HFPF_LAUNCH_INTEGRATE(true, false, false);
        else if (packed && color) HFPF_LAUNCH_INTEGRATE(true, true, false);
        else if (!color) HFPF_LAUNCH_INTEGRATE(false, false, false);
        else HFPF_LAUNCH_INTEGRATE(false, true, false);
        for (int k = 0; k < 3; k++) HIPCHK(h, detail_mark());
    } else {
        if (packed && !color) HFPF_LAUNCH_INTEGRATE(true, false, true);
        else if (packed && color) HFPF_LAUNCH_INTEGRATE(true, true, true);
        else if (!color) HFPF_LAUNCH_INTEGRATE(false, false, true);
        else HFPF_LAUNCH_INTEGRATE(false, true, true);
        {  // the points that found no room in a bin (usually a few thousand, everything for a batch without a plan): direct forms
            const unsigned ogrid = (unsigned)std::min<uint64_t>(blocks_for((uint64_t)n_points * n_frames, 256), 8ull * 256);
            if (color) hipLaunchKernelGGL(k_integrate_overflow<true>, dim3(ogrid), dim3(256), 0, h->stream, h->g, h->t, log_rot);
            else hipLaunchKernelGGL(k_integrate_overflow<false>, dim3(ogrid), dim3(256), 0, h->stream, h->g, h->t, log_rot);
        }
        HIPCHK(h, detail_mark());
        if (have_plan) {
            if (h->h_ctr[C_NORMALS] > 0 || h->normals_possible) {  // without a normal record no cell has dependants
                detail_ran |= 2;
                if (h->update_cells) {
                    const int shape = pick_update_shape(h, (double)n_points * n_frames, nb);
#define HFPF_LAUNCH_UPDATE(C, S) \
    hipLaunchKernelGGL((k_update_cells<C, S.threads, S.cap, S.slots, S.desc, S.waves>), dim3(nb), dim3(S.threads), 0, h->stream, h->g, h->t, nb)
                    if (color && shape == 1) HFPF_LAUNCH_UPDATE(true, kUpdWideColor);
                    else if (color) HFPF_LAUNCH_UPDATE(true, kUpdDenseColor);
                    else if (shape == 1) HFPF_LAUNCH_UPDATE(false, kUpdWide);
                    else HFPF_LAUNCH_UPDATE(false, kUpdDense);
#undef HFPF_LAUNCH_UPDATE
                } else {
                    if (color) hipLaunchKernelGGL(k_update<true>, dim3(nb), dim3(kUpdThreads), 0, h->stream, h->g, h->t, nb);
                    else hipLaunchKernelGGL(k_update<false>, dim3(nb), dim3(kUpdThreads), 0, h->stream, h->g, h->t, nb);
                }
            }
            HIPCHK(h, detail_mark());
            detail_ran |= 4;
            if (color) hipLaunchKernelGGL(k_buffer<true>, dim3(nb), dim3(256), 0, h->stream, h->g, h->t, nb);
            else hipLaunchKernelGGL(k_buffer<false>, dim3(nb), dim3(256), 0, h->stream, h->g, h->t, nb);
            HIPCHK(h, detail_mark());
        } else {
            HIPCHK(h, detail_mark());
            HIPCHK(h, detail_mark());
        }
        h->bin_have_hist = true;
        h->bin_prev_points = (double)n_points * n_frames;
    }
#undef HFPF_LAUNCH_INTEGRATE
    HIPCHK(h, hipGetLastError());
    if (h->timing_detail) h->ev_detail_ran.push_back(detail_ran);
    detail_guard.done = true;
    if (h->timing) {
        HIPCHK(h, hipEventRecord(e1, h->stream));
        h->ev_pending.emplace_back(e0, e1);
    }
    HIPCHK(h, hipEventRecord(s->done, h->stream));
    s->pending = true;
    h->dirty = true;  // state_changed = true, grid.hpp:189
    h->frames_integrated += n_frames;
    if (h->mbox && n_frames >= 4) {  // a batch: the next call is probably a clean pass, which starts by reading the counters
        h->pub_seq = ++h->mbox_seq;
        k_publish_counters<<<1, 256, 0, h->stream>>>(h->t.ctr, h->t.log_ctr, h->mbox, h->pub_seq);
        HIPCHK(h, hipGetLastError());
    } else {
        h->pub_seq = 0;
    }
    return HFPF_OK;
}

int resolve_timing(hfpf_handle* h)
{
    if (h->ev_pending.empty() && h->ev_pending_clean.empty() && h->ev_detail.empty()) return HFPF_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (auto& pr : h->ev_pending_clean) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, pr.first, pr.second));
        h->t_clean_ms += (double)ms;
        h->n_clean_timed++;
        h->ev_free.push_back(pr.first);
        h->ev_free.push_back(pr.second);
    }
    h->ev_pending_clean.clear();
    for (auto& pr : h->ev_pending) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, pr.first, pr.second));
        h->t_integrate_ms += (double)ms;
        h->n_integrate_launches++;
        h->ev_free.push_back(pr.first);
        h->ev_free.push_back(pr.second);
    }
    h->ev_pending.clear();
    for (size_t c = 0; 4 * c + 3 < h->ev_detail.size() && c < h->ev_detail_ran.size(); c++) {
        for (int k = 0; k < 3; k++) {
            if (!(h->ev_detail_ran[c] & (1u << k))) continue;  // not launched in this call (e.g. no dependants yet: no k_update_cells)
            float ms = 0.f;
            HIPCHK(h, hipEventElapsedTime(&ms, h->ev_detail[4 * c + k], h->ev_detail[4 * c + k + 1]));
            h->t_detail_ms[k] += (double)ms;
            h->n_detail[k]++;
        }
    }
    for (hipEvent_t e : h->ev_detail) h->ev_free.push_back(e);
    h->ev_detail.clear();
    h->ev_detail_ran.clear();
    return HFPF_OK;
}


// ---- RCCL, resolved with dlopen/dlsym ---------------------------------------------------------------
// (same ABI as <rccl/rccl.h>; declared here so libhfpf.so has no link-time dependency on librccl)
typedef struct ncclComm* ncclComm_t_;
typedef struct { char internal[128]; } ncclUniqueId_;
enum { ncclSuccess_ = 0 };
enum { ncclChar_ = 0, ncclUint64_ = 5 };  // ncclInt8/ncclChar = 0, ncclUint64 = 5 (rccl.h ncclDataType_t)
enum { ncclSum_ = 0 };
struct RcclApi {
    int (*GetUniqueId)(ncclUniqueId_*) = nullptr;
    int (*CommInitRank)(ncclComm_t_*, int, ncclUniqueId_, int) = nullptr;
    int (*CommDestroy)(ncclComm_t_) = nullptr;
    int (*CommCount)(const ncclComm_t_, int*) = nullptr;
    int (*CommUserRank)(const ncclComm_t_, int*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t_, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* lib = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mtx;

int load_rccl(std::string& err)
{
    std::lock_guard<std::mutex> lk(g_rccl_mtx);
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    for (const char* n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) {
        err = std::string("cannot load librccl: ") + dlerror();
        return -1;
    }
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.CommCount = (decltype(a.CommCount))dlsym(lib, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(lib, "ncclCommUserRank");
    a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
    a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.AllReduce) {
        err = "librccl lacks a required symbol";
        return -1;
    }
    g_rccl = a;
    return 0;
}

#define NCCLCHK(h, call)                                                                                              \
    do {                                                                                                              \
        int r_ = (call);                                                                                              \
        if (r_ != ncclSuccess_)                                                                                       \
            return fail(h, HFPF_ERR_DIST, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
    } while (0)

// Export the cells this handle occupied since the last exchange into h->ex_send; returns the count.
int epoch_export_locked(hfpf_handle* h, uint64_t* n_out, uint64_t min_capacity_records)
{
    int rc = read_counters(h);
    if (rc) return rc;
    if ((rc = check_device_errors(h))) return rc;
    const uint64_t n_occ = std::min<uint64_t>(h->h_ctr[C_OCC], h->t.max_occ);
    const uint64_t n_cells = n_occ - std::min(n_occ, h->occ_exported);
    const uint64_t n_fr_all = std::min<uint64_t>(h->h_ctr[C_FRAMES], h->t.max_frames);
    const uint64_t n_frames = n_fr_all - std::min(n_fr_all, h->frames_exported);
    const uint64_t n_new = n_cells + 2 * n_frames;  // one record per cell, two per frame (its viewpoint)
    if ((rc = scratch(h, h->ex_send, std::max<uint64_t>(std::max(n_new, min_capacity_records), 1) * sizeof(EpochRec)))) return rc;
    if (n_new) {
        hipLaunchKernelGGL(k_epoch_export, dim3(blocks_for(n_new, 256)), dim3(256), 0, h->stream, h->g, h->t, h->occ_exported, n_cells, h->frames_exported, n_frames,
                           (EpochRec*)h->ex_send.p);
        HIPCHK(h, hipGetLastError());
    }
    h->frames_seen = n_fr_all;  // (the clean pass that follows the exchange marks them exchanged, like the cells)
    *n_out = n_new;
    return HFPF_OK;
}

int epoch_import_locked(hfpf_handle* h, const void* dev_records, uint64_t n)
{
    if (n == 0) return HFPF_OK;
    h->pub_seq = 0;  // the import changes counters behind any snapshot already on its way
    hipLaunchKernelGGL(k_epoch_import, dim3(blocks_for(n, 256 * kImportTiles)), dim3(256), 0, h->stream, h->g, h->t, (const EpochRec*)dev_records, n);
    HIPCHK(h, hipGetLastError());
    return HFPF_OK;
}

// Device records [0, n) of one rank's slice of a gathered exchange buffer -> this handle's tables.  The one place both
// transports (RCCL all-gather below, hfpf_epoch_import for host-staged / virtual ranks) go through.
int import_rank_slice_locked(hfpf_handle* h, const void* dev_buffer, uint64_t slice_stride_bytes, int src_rank, uint64_t n_records)
{
    if (n_records == 0) return HFPF_OK;
    return epoch_import_locked(h, (const char*)dev_buffer + (size_t)src_rank * slice_stride_bytes, n_records);
}

// Every other rank's slice of an all-gathered buffer (world slices of slice_stride_bytes; slice r holds counts[r] records, the
// rest of it is padding that is never read).
int import_gathered_locked(hfpf_handle* h, const void* dev_buffer, uint64_t slice_stride_bytes, int world, int my_rank, const unsigned long long* counts)
{
    for (int r = 0; r < world; r++) {
        if (r == my_rank) continue;
        if (counts[r] * sizeof(EpochRec) > slice_stride_bytes) return fail(h, HFPF_ERR_BAD_ARG, "gathered slice %d: %llu records do not fit the stride", r, counts[r]);
        if (int rc = import_rank_slice_locked(h, dev_buffer, slice_stride_bytes, r, counts[r])) return rc;
    }
    return HFPF_OK;
}

// Failure consensus in front of a data collective: every rank contributes ONE status word (a payload below 2^63, bit 63 = "this
// rank has failed") to an all-gather that it enters whatever happened to it locally -- the status gather itself needs nothing but
// the (world+1)-word buffer allocated by hfpf_dist_init.  Afterwards every rank knows whether any rank failed and all of them
// leave the collective sequence at the same point: the failed rank with its own error, the others with HFPF_ERR_DIST.  Without
// it a rank that returns early (capacity overflow, poisoned handle, failed allocation) leaves its peers blocked in the next
// ncclAllGather / ncclAllReduce for ever.  payloads_out (world entries, may be null) receives every rank's payload.
int dist_status_gather_locked(hfpf_handle* h, int local_rc, uint64_t payload, const char* where, unsigned long long* payloads_out)
{
    constexpr unsigned long long kFailBit = 1ull << 63;
    const std::string local_err = h->err;  // keep the text of the local failure: the calls below may overwrite it
    unsigned long long* d_counts = (unsigned long long*)h->ex_counts.p;
    if (!d_counts || !h->h_counts) return fail(h, HFPF_ERR_DIST, "%s: no status buffer (hfpf_dist_init did not complete)", where);
    h->h_counts[h->world] = (payload & ~kFailBit) | (local_rc ? kFailBit : 0ull);
    HIPCHK(h, hipMemcpyAsync(d_counts + h->world, h->h_counts + h->world, 8, hipMemcpyHostToDevice, h->stream));
    NCCLCHK(h, g_rccl.AllGather(d_counts + h->world, d_counts, 1, ncclUint64_, (ncclComm_t_)h->comm, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_counts, d_counts, (size_t)h->world * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int failed_rank = -1;
    for (int r = 0; r < h->world; r++) {
        if ((h->h_counts[r] & kFailBit) && failed_rank < 0) failed_rank = r;
        h->h_counts[r] &= ~kFailBit;
        if (payloads_out) payloads_out[r] = h->h_counts[r];
    }
    if (local_rc) {
        h->err = local_err;
        return local_rc;
    }
    if (failed_rank >= 0) return fail(h, HFPF_ERR_DIST, "rank %d failed before the %s; the call is abandoned on every rank", failed_rank, where);
    return HFPF_OK;
}

// Exchange buffers for `records` records per rank.  The send buffer keeps its first `keep` records.
int grow_exchange_buffers_locked(hfpf_handle* h, uint64_t records, uint64_t keep)
{
    int rc;
    if (h->ex_send.bytes < records * sizeof(EpochRec)) {
        DevBuf bigger;
        if ((rc = scratch(h, bigger, records * sizeof(EpochRec)))) return rc;
        if (keep) HIPCHK(h, hipMemcpyAsync(bigger.p, h->ex_send.p, keep * sizeof(EpochRec), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->ex_send.p) {
            HIPCHK(h, hipFree(h->ex_send.p));
            h->device_bytes -= h->ex_send.bytes;
        }
        h->ex_send = bigger;
    }
    return scratch(h, h->ex_recv, (size_t)h->world * records * sizeof(EpochRec));
}

// RCCL exchange at the head of a clean pass: status gather (with the per-rank record counts as payload), then the all-gather of
// the padded record lists.  pre_rc != 0: this rank failed before the call (poisoned handle); it still takes part in the status
// gather so that its peers return an error instead of waiting for it.  Nothing that can fail sits between two collectives
// without a status gather behind it: the buffers hold ex_cap_records per rank -- the same number on every rank, because it
// starts from the same configuration and grows by the same rule from the same gathered maximum -- so "must grow" is the same
// decision everywhere, and the ranks that grow agree on the outcome before the record gather.
int dist_exchange_locked(hfpf_handle* h, int pre_rc)
{
    uint64_t n_mine = 0;
    int local_rc = pre_rc;
    if (!local_rc) local_rc = epoch_export_locked(h, &n_mine, h->ex_cap_records);
    if (local_rc) n_mine = 0;
    int rc = dist_status_gather_locked(h, local_rc, n_mine, "epoch exchange", nullptr);
    if (rc) return rc;
    uint64_t maxc = 0;
    for (int r = 0; r < h->world; r++) maxc = std::max<uint64_t>(maxc, h->h_counts[r]);
    if (maxc == 0) return HFPF_OK;
    if (maxc > h->ex_cap_records) {
        const uint64_t new_cap = maxc + maxc / 4;
        const int grc = grow_exchange_buffers_locked(h, new_cap, n_mine);
        std::vector<unsigned long long> counts(h->h_counts, h->h_counts + h->world);  // the second gather reuses h_counts
        if ((rc = dist_status_gather_locked(h, grc, new_cap, "growth of the exchange buffers", nullptr))) return rc;
        std::copy(counts.begin(), counts.end(), h->h_counts);
        h->ex_cap_records = new_cap;
    }
    NCCLCHK(h, g_rccl.AllGather(h->ex_send.p, h->ex_recv.p, maxc * sizeof(EpochRec), ncclChar_, (ncclComm_t_)h->comm, h->stream));
    return import_gathered_locked(h, h->ex_recv.p, maxc * sizeof(EpochRec), h->world, h->rank, h->h_counts);
}

// k_gate with four tiles per workgroup only when the input is large enough to fill the chip that way (one reservation per
// list and workgroup; small inputs keep one tile so that the latency-heavy stencil probes spread over as many CUs as possible).
void launch_gate(hfpf_handle* h, const uint32_t* cells_a, uint64_t n_a, const uint32_t* cells_b, uint64_t n_b, uint32_t* pend_out)
{
    const uint64_t n = n_a + n_b;
    if (n >= (1ull << 20))
        hipLaunchKernelGGL(k_gate<4>, dim3(blocks_for(n, 256 * 4)), dim3(256), 0, h->stream, h->g, h->t, cells_a, n_a, cells_b, n_b, pend_out);
    else
        hipLaunchKernelGGL(k_gate<1>, dim3(blocks_for(n, 256)), dim3(256), 0, h->stream, h->g, h->t, cells_a, n_a, cells_b, n_b, pend_out);
}

// One clean pass.  Two host read-backs: at the start (what the integrate launches since the last pass produced) and after the
// dependant-table update (how many cells to replay, overflow bits).  Everything between them is sized from upper bounds on the
// host and reads its exact counts from the device counters.
int clean_locked(hfpf_handle* h, int pre_rc)
{
    Tables& t = h->t;
    hipStream_t s = h->stream;
    int rc;
    // collective: every rank cleans at the same schedule point; a rank that is already unusable (pre_rc) still says so to its peers
    if (h->dist_on && (rc = dist_exchange_locked(h, pre_rc))) return rc;
    if (pre_rc) return pre_rc;
    if ((rc = read_counters(h))) return rc;
    if ((rc = check_device_errors(h))) return rc;
    const uint64_t n_occ = std::min<uint64_t>(h->h_ctr[C_OCC], t.max_occ);
    const uint64_t n_normals = h->h_ctr[C_NORMALS];
    const uint64_t n_pend = h->pend_valid ? h->h_ctr[C_PEND] : 0;  // cells the previous pass left without a normal (pend_a)
    h->occ_exported = n_occ;  // everything occupied so far (locally or imported) has been exchanged
    h->frames_exported = std::max(h->frames_exported, h->frames_seen);  // ... and the viewpoints of the frames the last export covered
    h->dirty = false;  // state_changed = false, grid.hpp:313
    h->clean_passes++;

    {
        LinkRanges lr;
        uint64_t max_new = 0;
        for (int r = 0; r < kLogRegions; r++) {
            const uint64_t n_r = std::min<uint64_t>(h->h_log_ctr[r * 16], t.log_region_cap);
            const uint64_t base = (uint64_t)r * t.log_region_cap;
            lr.first[r] = (uint32_t)(base + h->n_linked[r] + 1);
            lr.last[r] = (uint32_t)(base + n_r);
            max_new = std::max(max_new, n_r - h->n_linked[r]);
            h->n_linked[r] = n_r;
        }
        // entries appended by k_buffer arrive chained; only k_integrate's direct form leaves marked entries behind
        if (max_new && !h->binned && h->h_ctr[C_BUFFERED] != h->direct_linked) {  // (the binned form chains its few direct appends itself)
            hipLaunchKernelGGL(k_link_log, dim3(blocks_for(max_new, 256), kLogRegions), dim3(256), 0, s, t, lr);
            HIPCHK(h, hipGetLastError());
        }
        h->direct_linked = h->h_ctr[C_BUFFERED];
    }
    if (n_occ == 0) return HFPF_OK;

    // candidates: the cells that failed the gate last time (pending list) + the cells occupied since (new tail of occ_list)
    const uint64_t n_new_occ = n_occ - std::min(n_occ, h->gate_done);
    const uint64_t n_in = n_pend + n_new_occ;  // upper bound of everything this pass can produce per candidate
    if (n_in == 0) return HFPF_OK;
    if ((rc = scratch(h, h->pend_b, n_in * 4))) return rc;
    if ((rc = scratch(h, h->keys_a, n_in * 8))) return rc;
    // The registration counts of this pass are bounded by (2K+1) * n_in; the kernels below read the exact counts from the
    // device counters (kCountOnDevice), and the host picks the values up at the read-back after them.
    const uint64_t reg_ub = (2ull * (uint64_t)h->g.K + 1ull) * n_in;
    // Incremental update of the dependant table, or a compacting rebuild?  A conservative space estimate decides (C_DEP as of the
    // read-back above: nothing has changed it since).  What one incremental update can take from dep[]: a block of the next
    // power-of-two capacity for every list that outgrows its own (kernels.hpp dep_capacity) -- below twice its new length, i.e.
    // at most twice (all live entries: registrations + filed pre-dependants so far, + the registration bound of this pass) -- and
    // one entry per cell occupied since the last pass (the pre-dependants filed at the head of the pass).
    const uint64_t live_ub = std::min<uint64_t>(h->h_ctr[C_REG], t.max_reg) + std::min<uint64_t>(h->h_ctr[C_PREREG], t.max_reg);
    bool full = h->h_ctr[C_DEP] + 2 * live_ub + 2 * reg_ub + n_new_occ > t.max_dep;
    // sentinels + the pass's list counters + the pre-dependants of the cells occupied since the last pass become their lists
    hipLaunchKernelGGL(k_clean_begin, dim3(blocks_for(n_in, 256)), dim3(256), 0, s, t, n_in, (const uint32_t*)(t.occ_list + h->gate_done), n_new_occ, full ? 1u : 0u);
    launch_gate(h, (const uint32_t*)h->pend_a.p, n_pend, (const uint32_t*)(t.occ_list + h->gate_done), n_new_occ, (uint32_t*)h->pend_b.p);
    HIPCHK(h, hipGetLastError());
    h->gate_done = n_occ;
    h->normals_possible = true;
    std::swap(h->pend_a, h->pend_b);  // cells that got a normal in this pass are dropped by the next gate's kNormal test
    h->pend_valid = true;             // C_PEND now counts pend_a; the host reads it at the start of the next pass

    // canonical order: ascending (x,y,z) key; record id = n_normals + rank + 1
    // (with HFPF_MORTON_IDS the candidate keys are Z-order codes: three interleaved axes of the widest axis' bits)
    const unsigned cand_bits = HFPF_MORTON_IDS ? 3u * std::max(h->g.key_sy, std::max(h->g.key_sx - h->g.key_sy, h->g.key_bits - h->g.key_sx)) : h->g.key_bits;
    if ((rc = sort_keys_u64(h, t.cand_key, (uint64_t*)h->keys_a.p, n_in, cand_bits))) return rc;
    hipLaunchKernelGGL(k_normal, dim3(blocks_for(n_in, 128)), dim3(128), 0, s, h->g, t, (const uint64_t*)h->keys_a.p, kCountOnDevice, n_normals);
    const bool reg_small = n_in < (1ull << 18);  // tiles per workgroup: kernels.hpp k_register
    const uint64_t reg_tile = 256ull * (reg_small ? kRegTilesSmall : kRegTilesLarge);  // step-major, whole workgroups per step
    const uint64_t reg_blocks = ((n_in + reg_tile - 1) / reg_tile) * (2ull * (uint64_t)h->g.K + 1ull);
    const uint32_t count_deps = full ? 0u : 1u;  // (the compacting rebuild counts for itself, from zero)
    if (reg_small) hipLaunchKernelGGL(k_register<kRegTilesSmall>, dim3((unsigned)std::max<uint64_t>(reg_blocks, 1)), dim3(256), 0, s, h->g, t, kCountOnDevice, n_normals, count_deps);
    else hipLaunchKernelGGL(k_register<kRegTilesLarge>, dim3((unsigned)std::max<uint64_t>(reg_blocks, 1)), dim3(256), 0, s, h->g, t, kCountOnDevice, n_normals, count_deps);
    HIPCHK(h, hipGetLastError());
    uint64_t n_reg = 0, n_pre = 0, inc_touched = 0;
    // registrations already present in dep[]: every pass files all of its own, so that is the counter as this pass found it
    const uint64_t reg_first = std::min<uint64_t>(h->h_ctr[C_REG], t.max_reg);
    // A pass is SMALL when its upper bounds are: then the replay is launched over the bound and reads the touched-cell count on
    // the device, and the host does not wait for the pass at all (no mid-pass read-back: the GPU is not left idle for a host
    // round trip, and the next integrate call is enqueued behind the replay at once).  The bound above makes sure the pass
    // cannot run out of dep[] half-way (the one overflow the host would have to repair by compacting).
    const bool no_wait = !full && reg_ub < (1ull << 21) && h->clean_small_nowait;
    if (!full) {
        hipLaunchKernelGGL(k_depinc_offsets, dim3(blocks_for(reg_ub, 256 * kListTiles)), dim3(256), 0, s, t, kCountOnDevice);
        hipLaunchKernelGGL(k_depinc_fill, dim3(blocks_for(reg_ub, 256)), dim3(256), 0, s, t, reg_first, kCountOnDevice);
        HIPCHK(h, hipGetLastError());
        if (no_wait) {  // errors of this pass (capacity) surface at the next read-back and poison the handle there
            if (t.color)
                hipLaunchKernelGGL(k_replay<true>, dim3(blocks_for(reg_ub * kChains, 256)), dim3(256), 0, s, h->g, t, (const uint32_t*)t.touched_list, 1u, 0u,
                                   kCountOnDevice, n_normals);
            else
                hipLaunchKernelGGL(k_replay<false>, dim3(blocks_for(reg_ub * kChains, 256)), dim3(256), 0, s, h->g, t, (const uint32_t*)t.touched_list, 1u, 0u,
                                   kCountOnDevice, n_normals);
            HIPCHK(h, hipGetLastError());
            return HFPF_OK;
        }
        if ((rc = read_counters(h))) return rc;
        n_reg = std::min<uint64_t>(h->h_ctr[C_REG], t.max_reg);
        n_pre = std::min<uint64_t>(h->h_ctr[C_PREREG], t.max_reg);
        inc_touched = h->h_ctr[C_TOUCHED];
        if (h->h_ctr[C_ERR] == (unsigned long long)E_DEP) {  // dep[] ran out mid-way: compact
            hipLaunchKernelGGL(k_set_ctr, dim3(1), dim3(1), 0, s, t.ctr, (int)C_ERR, 0ull);
            if (inc_touched) hipLaunchKernelGGL(k_dep_reset, dim3(blocks_for(inc_touched, 256)), dim3(256), 0, s, t, inc_touched);  // poisoned cursors
            full = true;
        } else if ((rc = check_device_errors(h))) {
            return rc;
        }
    } else {
        if ((rc = read_counters(h))) return rc;
        if ((rc = check_device_errors(h))) return rc;
        n_reg = std::min<uint64_t>(h->h_ctr[C_REG], t.max_reg);
        n_pre = std::min<uint64_t>(h->h_ctr[C_PREREG], t.max_reg);
    }
    if (full) {
        const uint64_t n_all = n_reg + n_pre;
        if (2 * n_all > t.max_dep)  // (lists own power-of-two blocks: below twice their length)
            return fail(h, HFPF_ERR_CAPACITY, "dependant table: %llu entries do not fit %llu with their blocks", (unsigned long long)n_all, (unsigned long long)t.max_dep);
        // touched_list holds one entry per distinct cell among the n_all registrations: every normal record registers on at most
        // 2K+1 cells, so n_reg + n_pre <= max_normals * (2K+1) = max_reg = max_touched
        if (n_all > h->max_touched) return fail(h, HFPF_ERR_CAPACITY, "registrations: %llu > %llu", (unsigned long long)n_all, (unsigned long long)h->max_touched);
        hipLaunchKernelGGL(k_set_ctr3, dim3(1), dim3(1), 0, s, t.ctr, (int)C_DEP, 0ull, (int)C_TOUCHED, 0ull, -1, 0ull);
        if (n_all) {
            hipLaunchKernelGGL(k_dep_count, dim3(blocks_for(n_all, 256)), dim3(256), 0, s, t, n_reg, n_pre);
            HIPCHK(h, hipGetLastError());
            if ((rc = read_counters(h))) return rc;
            const uint64_t n_touched = h->h_ctr[C_TOUCHED];
            hipLaunchKernelGGL(k_dep_offsets, dim3(blocks_for(n_touched, 256)), dim3(256), 0, s, t, n_touched);
            hipLaunchKernelGGL(k_dep_fill, dim3(blocks_for(n_all, 256)), dim3(256), 0, s, t, n_reg, n_pre);
            hipLaunchKernelGGL(k_dep_reset, dim3(blocks_for(n_touched, 256)), dim3(256), 0, s, t, n_touched);
            HIPCHK(h, hipGetLastError());
            inc_touched = n_touched;  // superset of the cells touched by this pass; k_replay filters by record id
        }
        if ((rc = read_counters(h))) return rc;
        if ((rc = check_device_errors(h))) return rc;
    }
    // buffer replay of the cells that gained registrants in this pass (touched_list is still intact)
    if (inc_touched) {
        const uint32_t use_marks = full ? 0u : 1u;  // (a compacting rebuild leaves no notes and the lists in no particular order)
        // Bricks whose buffered points are ONE contiguous run of the log (the epoch was handed over in one integrate call, the
        // brick is new in it) replay by streaming that run: worth its launch over every brick when enough of the touched cells
        // lie in such bricks.  The chain walk takes the others.
        const uint64_t single = full ? 0 : std::min<uint64_t>(h->h_ctr[C_TOUCHED_SINGLE], inc_touched);
        const bool stream = !full && h->binned && h->stream_replay && single >= (1ull << 16);
        if (stream) {
            const uint32_t nbk = (uint32_t)h->n_bricks_known;
#define HFPF_LAUNCH_STREAM(C, S) hipLaunchKernelGGL((k_update_cells<C, S.threads, S.cap, S.slots, S.desc, S.waves, true>), dim3(nbk), dim3(S.threads), 0, s, h->g, t, nbk)
            const bool wide = h->upd_shape_forced >= 0 ? h->upd_shape_forced == 1 : h->upd_wide;  // the shape the dependant updates of this session take
            if (t.color && wide) HFPF_LAUNCH_STREAM(true, kUpdWideColor);
            else if (t.color) HFPF_LAUNCH_STREAM(true, kUpdDenseColor);
            else if (wide) HFPF_LAUNCH_STREAM(false, kUpdWide);
            else HFPF_LAUNCH_STREAM(false, kUpdDense);
#undef HFPF_LAUNCH_STREAM
            HIPCHK(h, hipGetLastError());
        }
        const uint64_t walked = stream ? inc_touched - single : inc_touched;
        if (walked) {
            // a 256-point tile appends consecutive log entries for neighbouring cells, so walking the cells in slot (brick-major)
            // order lets adjacent lanes share cache lines of the log; a short list is not worth the sort's launches
            const uint32_t* cells = t.touched_list;
            if (walked >= (1ull << 18)) {
                if ((rc = scratch(h, h->vals_a, inc_touched * 4))) return rc;
                unsigned slot_bits = 9;  // slot = brick * 512 + cell; the brick count is as of the read-back just above
                while ((1ull << slot_bits) < (h->n_bricks_known + 2) * (uint64_t)kBrickCells && slot_bits < 32) slot_bits++;
                if ((rc = sort_keys_u32(h, t.touched_list, (uint32_t*)h->vals_a.p, inc_touched, slot_bits))) return rc;
                cells = (const uint32_t*)h->vals_a.p;
            }
            if (t.color)
                hipLaunchKernelGGL(k_replay<true>, dim3(blocks_for(inc_touched * kChains, 256)), dim3(256), 0, s, h->g, t, cells, use_marks, stream ? 1u : 0u, inc_touched, n_normals);
            else
                hipLaunchKernelGGL(k_replay<false>, dim3(blocks_for(inc_touched * kChains, 256)), dim3(256), 0, s, h->g, t, cells, use_marks, stream ? 1u : 0u, inc_touched, n_normals);
            HIPCHK(h, hipGetLastError());
        }
    }
    return HFPF_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

int hfpf_abi_version(void) { return HFPF_ABI_VERSION; }

void hfpf_default_config(hfpf_config* c)
{
    if (!c) return;
    memset(c, 0, sizeof *c);
    c->struct_size = sizeof *c;
    c->resolution = 0.005f;  // kResolution node.cpp:91
    const double box[6] = {-0.80, 1.80, -1.5, 1.5, 0.0, 1.0};  // launch file line 7
    memcpy(c->bbox, box, sizeof box);
    c->k = 2;                    // node.cpp:163
    c->K = 3;                    // node.cpp:311
    c->gate = 20;                // grid.hpp:352
    c->cylinder_radius = 0.001;  // grid.hpp:36
    c->ball_radius = 0.015;      // grid.hpp:35
    c->z_clip_min = 0.28;        // node.cpp:92
    c->z_clip_max = 0.6;         // node.cpp:93
    c->device = 0;
}

const char* hfpf_last_error(const hfpf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int hfpf_create(const hfpf_config* cfg, hfpf_handle** out)
{
    if (!cfg || !out) return fail(nullptr, HFPF_ERR_BAD_ARG, "hfpf_create: null argument");
    if (cfg->struct_size != sizeof(hfpf_config)) return fail(nullptr, HFPF_ERR_BAD_CONFIG, "hfpf_config.struct_size mismatch (ABI %d)", HFPF_ABI_VERSION);
    *out = nullptr;
    hfpf_handle* h = new hfpf_handle();
    h->cfg = *cfg;
    auto bail = [&](int rc) {
        g_create_error = h->err;
        for (void* p : h->allocs) (void)hipFree(p);
        if (h->h_ctr) (void)hipHostFree(h->h_ctr);
        if (h->h_log_ctr) (void)hipHostFree(h->h_log_ctr);
        if (h->mbox) (void)hipHostFree(h->mbox);
        if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
        for (hipStream_t cs : h->copy_more)
            if (cs) (void)hipStreamDestroy(cs);
        if (h->stream) (void)hipStreamDestroy(h->stream);
        delete h;
        return rc;
    };
    int rc = setup_params(h);
    if (rc) return bail(rc);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return bail(fail(h, HFPF_ERR_HIP, "no HIP device available (%s); this engine has no CPU path", hipGetErrorString(e)));
    if (cfg->device < 0 || cfg->device >= ndev) return bail(fail(h, HFPF_ERR_BAD_CONFIG, "device %d out of range (%d devices)", cfg->device, ndev));
    if ((e = hipSetDevice(cfg->device)) != hipSuccess) return bail(fail(h, HFPF_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e)));
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail(fail(h, HFPF_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)));
    if ((e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking)) != hipSuccess) return bail(fail(h, HFPF_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)));
    if (const char* cs = getenv("HFPF_COPY_STREAMS")) h->n_copy_streams = std::max(1, std::min(atoi(cs), 4));
    for (int k = 0; k + 1 < h->n_copy_streams; k++)
        if ((e = hipStreamCreateWithFlags(&h->copy_more[k], hipStreamNonBlocking)) != hipSuccess) return bail(fail(h, HFPF_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)));
    if ((e = hipHostMalloc((void**)&h->h_ctr, C_COUNT * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess)
        return bail(fail(h, HFPF_ERR_HIP, "hipHostMalloc: %s", hipGetErrorString(e)));
    memset(h->h_ctr, 0, C_COUNT * sizeof(unsigned long long));
    if ((e = hipHostMalloc((void**)&h->h_log_ctr, kLogRegions * 16 * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess)
        return bail(fail(h, HFPF_ERR_HIP, "hipHostMalloc: %s", hipGetErrorString(e)));
    memset(h->h_log_ctr, 0, kLogRegions * 16 * sizeof(unsigned long long));
    {
        const char* uf = getenv("HFPF_UPDATE_FORM");
        h->update_cells = !(uf && uf[0] == 'p');
        if (const char* hb = getenv("HFPF_HOST_BATCH")) h->host_batch = std::max(1, std::min(atoi(hb), kFrameSlots));
        if (const char* us = getenv("HFPF_UPD_SHAPE")) h->upd_shape_forced = std::max(0, std::min(atoi(us), 1));
        if (const char* tr = getenv("HFPF_TRACE_SHAPE")) h->trace_shape = tr[0] != '0';
        if (const char* sp = getenv("HFPF_BIN_SPARE")) h->bin_spare = sp[0] != '0';
        if (const char* sr = getenv("HFPF_STREAM_REPLAY")) h->stream_replay = sr[0] != '0';
        if (const char* bs = getenv("HFPF_BIN_SLACK")) h->bin_slack = std::max(1.0f, std::min(4.0f, (float)atof(bs)));
        if (const char* nw = getenv("HFPF_CLEAN_NOWAIT")) h->clean_small_nowait = nw[0] != '0';
        if (const char* bs = getenv("HFPF_TEST_BIN_SCALE")) h->test_bin_scale = std::max(0.f, std::min(1.f, (float)atof(bs)));
        const char* mb = getenv("HFPF_MAILBOX");
        if (!mb || mb[0] != '0') {
            if ((e = hipHostMalloc((void**)&h->mbox, (kMboxWords + 8) * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped)) != hipSuccess)
                return bail(fail(h, HFPF_ERR_HIP, "hipHostMalloc (mailbox): %s", hipGetErrorString(e)));
            memset(h->mbox, 0, (kMboxWords + 8) * sizeof(unsigned long long));
        }
    }
    {
        int per_cu = 0, cus = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, h->binned ? k_integrate<true, false, true> : k_integrate<true, false, false>, 256, 0) != hipSuccess) per_cu = 4;
        h->integrate_grid = std::max(1, per_cu) * std::max(1, cus);
    }
    if ((rc = alloc_tables(h))) return bail(rc);
    {
        // Work buffers that would otherwise grow on first use are sized here like the pools: the clean
        // passes' key / value / sort scratch from max_normals, the per-call brick bins from max_call_points when given.
        const uint64_t n0 = h->cfg.max_normals;
        if ((rc = scratch(h, h->keys_a, n0 * 8)) || (rc = scratch(h, h->keys_b, n0 * 8)) || (rc = scratch(h, h->vals_a, n0 * 4)) ||
            (rc = scratch(h, h->vals_b, n0 * 4)) || (rc = scratch(h, h->pend_b, n0 * 4)) || (rc = scratch(h, h->sort_tmp, n0 * 16)))
            return bail(rc);
        if (h->cfg.max_call_points && h->binned) {
            const uint64_t pts = h->cfg.max_call_points;
            const uint64_t pool = std::min<uint64_t>(bin_pool_entries(h, pts, h->cfg.max_bricks), 0xFFFFFFFFull);
            if ((rc = scratch(h, h->bin_pt_buf, pool * sizeof(float4)))) return bail(rc);
            if (h->t.color && (rc = scratch(h, h->bin_rgb_buf, pool * 4))) return bail(rc);
            h->bin_pool = pool;
            if ((rc = scratch(h, h->ovf_pt_buf, pts * sizeof(float4))) || (rc = scratch(h, h->ovf_aux_buf, pts * sizeof(uint2)))) return bail(rc);
        }
    }
    for (int k = 0; k < 2; k++) {  // (a failure here only costs speed: extract then copies straight into pageable memory)
        if (hipHostMalloc(&h->xfer_pin[k], kXferChunk, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&h->xfer_ev[k], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (h->xfer_pin[0]) (void)hipHostFree(h->xfer_pin[0]);
            h->xfer_pin[0] = nullptr;
            break;
        }
    }
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return bail(fail(h, HFPF_ERR_HIP, "sync after init: %s", hipGetErrorString(e)));
    *out = h;
    return HFPF_OK;
}

int hfpf_destroy(hfpf_handle* h)
{
    if (!h) return HFPF_OK;
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    for (void* p : h->allocs) (void)hipFree(p);
    for (DevBuf* b : {&h->sort_tmp, &h->keys_a, &h->keys_b, &h->vals_a, &h->vals_b, &h->rows_dev, &h->probe_a, &h->probe_b, &h->probe_c, &h->probe_d,
                      &h->probe_e, &h->probe_f})
        if (b->p) (void)hipFree(b->p);
    for (DevBuf* b : {&h->ex_send, &h->ex_recv, &h->ex_counts, &h->stats_total, &h->bin_pt_buf, &h->bin_rgb_buf, &h->bin_sums, &h->ovf_pt_buf, &h->ovf_aux_buf, &h->pend_a, &h->pend_b})
        if (b->p) (void)hipFree(b->p);
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t_)h->comm);
    if (h->h_counts) (void)hipHostFree(h->h_counts);
    for (auto& s : h->stage) {
        if (s.h_pose) (void)hipHostFree(s.h_pose);
        if (s.d_pose) (void)hipFree(s.d_pose);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    delete h->stage_pool;
    h->stage_pool = nullptr;
    for (int k = 0; k < 2; k++) {
        if (h->xfer_pin[k]) (void)hipHostFree(h->xfer_pin[k]);
        if (h->xfer_ev[k]) (void)hipEventDestroy(h->xfer_ev[k]);
    }
    for (auto& f : h->fslot) {
        if (f.h) (void)hipHostFree(f.h);
        if (f.done) (void)hipEventDestroy(f.done);
        if (f.copied) (void)hipEventDestroy(f.copied);
    }
    if (h->ring_d) (void)hipFree(h->ring_d);
    if (h->busy_ev) (void)hipEventDestroy(h->busy_ev);
    if (h->copy_stream) {
        (void)hipStreamSynchronize(h->copy_stream);
        (void)hipStreamDestroy(h->copy_stream);
    }
    for (hipStream_t cs : h->copy_more)
        if (cs) {
            (void)hipStreamSynchronize(cs);
            (void)hipStreamDestroy(cs);
        }
    for (auto& pr : h->ev_pending) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (auto& pr : h->ev_pending_clean) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (auto e : h->ev_detail) (void)hipEventDestroy(e);
    for (auto e : h->ev_free) (void)hipEventDestroy(e);
    if (h->h_ctr) (void)hipHostFree(h->h_ctr);
    if (h->h_log_ctr) (void)hipHostFree(h->h_log_ctr);
    if (h->mbox) (void)hipHostFree(h->mbox);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return HFPF_OK;
}

int hfpf_get_dims(const hfpf_handle* h, int32_t dims[3], double* resolution)
{
    if (!h) return HFPF_ERR_BAD_ARG;
    if (dims) memcpy(dims, h->g.dim, 3 * sizeof(int32_t));
    if (resolution) *resolution = h->g.res;
    return HFPF_OK;
}

static int flush_pending_locked(hfpf_handle* h);

int hfpf_integrate_device(hfpf_handle* h, const void* dev_base, uint32_t n_frames, uint64_t frame_stride, uint32_t n_points, uint32_t point_step,
                          uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t off_rgb, const double* poses, const uint32_t* frame_ids)
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    if (int rc = check_usable(h)) return rc;
    return integrate_device_locked(h, dev_base, n_frames, frame_stride, n_points, point_step, off_x, off_y, off_z, off_rgb, poses, frame_ids);
}

// Large host-to-host copies (the bounce copy of hfpf_integrate, the row download of extract) are split over the caller and the
// helper threads of the handle's StagePool (HFPF_STAGE_THREADS, default half the process's cores - 1, at most 7; 0 = none).
static void host_copy(hfpf_handle* h, void* dst, const void* src, size_t bytes)
{
    if (h->stage_threads < 0) {
        const char* e = getenv("HFPF_STAGE_THREADS");
        int cores = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cores = std::min(cores > 0 ? cores : 1 << 20, CPU_COUNT(&set));  // the process's share
        h->stage_threads = e ? std::max(0, std::min(atoi(e), 15)) : std::max(0, std::min(7, cores / 2 - 1));
    }
    if (h->stage_threads > 0 && bytes >= (1u << 20)) {
        if (!h->stage_pool) h->stage_pool = new StagePool(h->stage_threads);
        h->stage_pool->copy(dst, src, bytes);
    } else {
        stream_copy((char*)dst, (const char*)src, bytes);
    }
}

// Hand the uploaded-but-not-launched host frames to the kernels: one integrate launch for the batch.
static int flush_pending_locked(hfpf_handle* h)
{
    if (h->pend_n == 0) return HFPF_OK;
    const uint32_t first = h->pend_first, n = h->pend_n;
    h->pend_n = 0;
    // the handle refuses work until hfpf_clear: frames accepted before the failure surfaced are dropped, and the caller is told
    if (h->poisoned) return fail(h, HFPF_ERR_STATE, "%u accepted host frame(s) dropped: handle failed earlier (%s); hfpf_clear resets it", n, h->poison_msg.c_str());
    // The slots of a batch alternate over the copy streams (slot % n_copy_streams): the kernels wait for the LAST upload of every
    // stream that carried one of them -- a stream is in order, so its earlier uploads are done too.  (f.done, recorded behind the
    // kernels below, then also means "this slot's upload has left its pinned bounce buffer".)
    for (uint32_t k = n, seen = 0; k-- > 0;) {
        const uint32_t which = (first + k) % (uint32_t)h->n_copy_streams;
        if (seen & (1u << which)) continue;
        seen |= 1u << which;
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->fslot[first + k].copied, 0));
    }
    int rc = integrate_device_locked(h, (const char*)h->ring_d + (size_t)first * h->ring_cap, n, h->ring_cap, h->pend_pts, h->pend_lay[0], h->pend_lay[1],
                                     h->pend_lay[2], h->pend_lay[3], h->pend_lay[4], h->pend_pose, nullptr);
    // frames that were accepted with HFPF_OK are lost: whichever entry point found them waiting, the handle stops until hfpf_clear
    if (rc) return poison_on_error(h, rc);
    for (uint32_t k = 0; k < n; k++) {
        FrameSlot& f = h->fslot[first + k];
        HIPCHK(h, hipEventRecord(f.done, h->stream));
        f.pending = true;
    }
    if (!h->busy_ev) HIPCHK(h, hipEventCreateWithFlags(&h->busy_ev, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(h->busy_ev, h->stream));
    h->busy_pending = true;
    return HFPF_OK;
}

// One host frame: upload on the copy stream into the next slot of the device ring; launch it -- together with the frames still
// waiting in front of it -- when the engine's stream is idle or the batch is full, otherwise leave it pending (the next frame,
// or any other call on the handle, launches it).  The upload of frame k+1 overlaps the kernels of frame k; `bounce` = copy the
// caller's (pageable) buffer into the slot's pinned buffer first, so that the caller's memory is free again when the call returns.
static int integrate_host_locked(hfpf_handle* h, const void* base, bool bounce, uint32_t n_points, uint32_t point_step, uint32_t off_x,
                                 uint32_t off_y, uint32_t off_z, uint32_t off_rgb, const double pose[12])
{
    if (!base || !pose) return fail(h, HFPF_ERR_BAD_ARG, "integrate: null buffer or pose");
    if (int rc0 = check_usable(h)) return rc0;
    int rc;
    if (n_points == 0) {
        if ((rc = flush_pending_locked(h))) return rc;  // frame ids stay in arrival order
        h->next_frame_id++;
        h->frames_integrated++;
        h->dirty = true;
        return HFPF_OK;
    }
    if ((point_step & 3) || (off_x & 3) || (off_y & 3) || (off_z & 3) || (off_rgb & 3))
        return fail(h, HFPF_ERR_BAD_ARG, "integrate: fields must be 4-byte aligned");
    if (std::max(std::max(off_x, off_y), std::max(off_z, off_rgb)) + 4 > point_step)
        return fail(h, HFPF_ERR_BAD_ARG, "integrate: field offset beyond point_step");
    const uint32_t lay[5] = {point_step, off_x, off_y, off_z, off_rgb};
    const size_t bytes = (size_t)n_points * point_step;
    // a frame of another shape, or a slot that is not the batch's neighbour (ring wrap), closes the pending batch
    if (h->pend_n && (h->pend_pts != n_points || memcmp(h->pend_lay, lay, sizeof lay) != 0 || (uint32_t)h->fslot_next != h->pend_first + h->pend_n)) {
        if ((rc = flush_pending_locked(h))) return rc;
    }
    if (h->ring_cap < bytes) {  // (re)size the ring: nothing may be in flight in it
        if ((rc = flush_pending_locked(h))) return rc;
        HIPCHK(h, sync_copy_streams(h));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (auto& f : h->fslot) f.pending = false;
        if (h->ring_d) {
            HIPCHK(h, hipFree(h->ring_d));
            h->device_bytes -= h->ring_cap * kFrameSlots;
        }
        h->ring_d = nullptr;
        h->ring_cap = 0;
        const size_t cap = (bytes + 4095) & ~(size_t)4095;
        HIPCHK(h, hipMalloc(&h->ring_d, cap * kFrameSlots));
        h->ring_cap = cap;
        h->device_bytes += cap * kFrameSlots;
        h->fslot_next = 0;
    }
    const uint32_t slot = (uint32_t)h->fslot_next;
    FrameSlot& f = h->fslot[slot];
    h->fslot_next = (h->fslot_next + 1) % kFrameSlots;
    if (f.pending) {  // the kernels that read this slot (kFrameSlots frames ago)
        HIPCHK(h, hipEventSynchronize(f.done));
        f.pending = false;
    }
    if (!f.done) HIPCHK(h, hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
    if (!f.copied) HIPCHK(h, hipEventCreateWithFlags(&f.copied, hipEventDisableTiming));
    const void* src = base;
    if (bounce) {
        if (f.cap_h < bytes) {
            if (f.h) HIPCHK(h, hipHostFree(f.h));
            f.h = nullptr;
            HIPCHK(h, hipHostMalloc(&f.h, h->ring_cap, hipHostMallocDefault));
            f.cap_h = h->ring_cap;
        }
        // the caller's buffer is free again when this call returns
        host_copy(h, f.h, base, bytes);
        src = f.h;
    }
    const uint32_t which = slot % (uint32_t)h->n_copy_streams;
    hipStream_t cs = which ? h->copy_more[which - 1] : h->copy_stream;
    HIPCHK(h, hipMemcpyAsync((char*)h->ring_d + (size_t)slot * h->ring_cap, src, bytes, hipMemcpyHostToDevice, cs));
    HIPCHK(h, hipEventRecord(f.copied, cs));
    if (h->pend_n == 0) {
        h->pend_first = slot;
        h->pend_pts = n_points;
        memcpy(h->pend_lay, lay, sizeof lay);
    }
    memcpy(h->pend_pose + 12 * h->pend_n, pose, 12 * sizeof(double));
    h->pend_n++;
    h->dirty = true;  // state_changed = true, grid.hpp:189 (the frame is accepted; its kernels follow)
    bool launch = h->pend_n >= (uint32_t)std::max(1, h->host_batch) || h->fslot_next == 0;  // batch full, or the ring wraps next
    if (!launch) {
        if (!h->busy_pending) {
            launch = true;
        } else {
            const hipError_t q = hipEventQuery(h->busy_ev);
            if (q == hipSuccess) h->busy_pending = false, launch = true;
            else if (q != hipErrorNotReady) HIPCHK(h, q);
        }
    }
    return launch ? flush_pending_locked(h) : HFPF_OK;
}

int hfpf_integrate(hfpf_handle* h, const void* base, uint32_t n_points, uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                   uint32_t off_rgb, const double pose[12])
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return integrate_host_locked(h, base, true, n_points, point_step, off_x, off_y, off_z, off_rgb, pose);
}

int hfpf_integrate_pinned(hfpf_handle* h, const void* pinned_base, uint32_t n_points, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                          uint32_t off_z, uint32_t off_rgb, const double pose[12])
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (pinned_base) {
        hipPointerAttribute_t attr;
        const hipError_t e = hipPointerGetAttributes(&attr, pinned_base);
        if (e != hipSuccess || attr.type != hipMemoryTypeHost) {
            (void)hipGetLastError();
            return fail(h, HFPF_ERR_BAD_ARG, "integrate_pinned: the buffer is not page-locked host memory (hfpf_host_alloc / hipHostRegister); use hfpf_integrate");
        }
    }
    return integrate_host_locked(h, pinned_base, false, n_points, point_step, off_x, off_y, off_z, off_rgb, pose);
}

int hfpf_host_alloc(hfpf_handle* h, uint64_t bytes, void** host_ptr)
{
    if (!h || !host_ptr) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipHostMalloc(host_ptr, (size_t)std::max<uint64_t>(bytes, 1), hipHostMallocDefault));
    return HFPF_OK;
}

int hfpf_host_free(hfpf_handle* h, void* host_ptr)
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    HIPCHK(h, sync_copy_streams(h));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipHostFree(host_ptr));
    return HFPF_OK;
}

int hfpf_is_dirty(hfpf_handle* h)
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    return h->dirty ? 1 : 0;
}

int hfpf_clean(hfpf_handle* h)
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    // Everything that can fail in front of the collectives of a distributed clean is folded into `pre`: a rank whose device
    // cannot be selected or whose deferred host-frame launch fails (scratch allocation, frame id >= max_frames, HIP error) still
    // enters the status gather of clean_locked, so its peers leave with HFPF_ERR_DIST instead of waiting in ncclAllGather for ever.
    int pre = HFPF_OK;
    if (hipError_t e_ = hipSetDevice(h->cfg.device)) pre = fail(h, HFPF_ERR_HIP, "hipSetDevice failed: %s", hipGetErrorString(e_));
    if (!pre) pre = poison_on_error(h, flush_pending_locked(h));  // host frames still waiting for their launch
    if (!pre) pre = check_usable(h);
    if (pre && !h->dist_on) return pre;
    if (!h->timing || pre) return poison_on_error(h, clean_locked(h, pre));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto get = [&](hipEvent_t& e) -> hipError_t {
        if (!h->ev_free.empty()) {
            e = h->ev_free.back();
            h->ev_free.pop_back();
            return hipSuccess;
        }
        return hipEventCreate(&e);
    };
    HIPCHK(h, get(e0));
    HIPCHK(h, get(e1));
    HIPCHK(h, hipEventRecord(e0, h->stream));  // after every queued integrate: measures the clean pass alone
    const int rc = poison_on_error(h, clean_locked(h, 0));
    HIPCHK(h, hipEventRecord(e1, h->stream));
    h->ev_pending_clean.emplace_back(e0, e1);
    return rc;
}

// Shared tail of extract: `stats` are the (possibly merged) sums to finalise.
static int extract_locked(hfpf_handle* h, const unsigned long long* stats, const hfpf_extract_opts* o, hfpf_row** rows, uint64_t* n_rows)
{
    ExtractOpts opt{0.0, -1, 0};
    if (o) {
        opt.min_count = o->min_count;
        opt.classify_threshold = o->classify_threshold;
        opt.paint_white = o->paint_white ? 1 : 0;
    }
    Tables& t = h->t;
    int rc;
    const uint64_t n = h->h_ctr[C_NORMALS];
    if (n == 0) return HFPF_OK;
    if ((rc = scratch(h, h->keys_a, n * 8))) return rc;
    if ((rc = scratch(h, h->keys_b, n * 8))) return rc;
    if ((rc = scratch(h, h->vals_a, n * 4))) return rc;
    if ((rc = scratch(h, h->vals_b, n * 4))) return rc;
    hipLaunchKernelGGL(k_set_ctr, dim3(1), dim3(1), 0, h->stream, t.ctr, (int)C_ROWS, 0ull);
    hipLaunchKernelGGL(k_extract_keys, dim3(blocks_for(n, 256 * kExtractTiles)), dim3(256), 0, h->stream, h->g, t, stats, n, opt, (uint64_t*)h->keys_a.p, (uint32_t*)h->vals_a.p);
    HIPCHK(h, hipGetLastError());
    if ((rc = sort_pairs_u64(h, (uint64_t*)h->keys_a.p, (uint64_t*)h->keys_b.p, (uint32_t*)h->vals_a.p, (uint32_t*)h->vals_b.p, n))) return rc;
    if ((rc = read_counters(h))) return rc;
    const uint64_t nr = h->h_ctr[C_ROWS];
    if (nr == 0) return HFPF_OK;
    if ((rc = scratch(h, h->rows_dev, nr * sizeof(Row)))) return rc;
    hipLaunchKernelGGL(k_extract_rows, dim3(blocks_for(nr, 256)), dim3(256), 0, h->stream, h->g, t, stats, nr, opt, (const uint64_t*)h->keys_b.p,
                       (const uint32_t*)h->vals_b.p, (Row*)h->rows_dev.p);
    HIPCHK(h, hipGetLastError());
    // (2 MB-aligned and advised as huge pages: a fresh 125 MB result is then ~60 page faults instead of 30,000 while it is filled)
    hfpf_row* host = nullptr;
    {
        void* mem = nullptr;
        const size_t want = nr * sizeof(hfpf_row);
        if (want >= (4u << 20) && posix_memalign(&mem, 2u << 20, (want + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1)) == 0) {
            (void)madvise(mem, (want + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1), MADV_HUGEPAGE);
            host = (hfpf_row*)mem;
        } else {
            host = (hfpf_row*)malloc(want);
        }
    }
    if (!host) return fail(h, HFPF_ERR_CAPACITY, "extract: host allocation of %llu rows failed", (unsigned long long)nr);
    // The rows go to pageable memory the caller will free(): a direct device-to-pageable copy runs at ~10 GB/s through the
    // runtime's own staging.  Two pinned 16 MB buffers instead: chunk i+1 crosses the link while chunk i is copied out by the
    // caller and the helper threads.
    const size_t total = nr * sizeof(Row);
    hipError_t e = hipSuccess;
    if (h->xfer_pin[0] && total >= 2 * kXferChunk) {
        const size_t n_chunks = (total + kXferChunk - 1) / kXferChunk;
        auto issue = [&](size_t i) -> hipError_t {
            const size_t off = i * kXferChunk, len = std::min(kXferChunk, total - off);
            const hipError_t r = hipMemcpyAsync(h->xfer_pin[i & 1], (const char*)h->rows_dev.p + off, len, hipMemcpyDeviceToHost, h->stream);
            return r != hipSuccess ? r : hipEventRecord(h->xfer_ev[i & 1], h->stream);
        };
        e = issue(0);
        if (e == hipSuccess && n_chunks > 1) e = issue(1);
        for (size_t i = 0; i < n_chunks && e == hipSuccess; i++) {
            const size_t off = i * kXferChunk, len = std::min(kXferChunk, total - off);
            e = hipEventSynchronize(h->xfer_ev[i & 1]);
            if (e != hipSuccess) break;
            host_copy(h, (char*)host + off, h->xfer_pin[i & 1], len);
            if (i + 2 < n_chunks) e = issue(i + 2);  // this buffer is free again
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    } else {
        e = hipMemcpyAsync(host, h->rows_dev.p, total, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    if (e != hipSuccess) {
        free(host);
        return fail(h, HFPF_ERR_HIP, "extract copy: %s", hipGetErrorString(e));
    }
    *rows = host;
    *n_rows = nr;
    return HFPF_OK;
}

int hfpf_extract(hfpf_handle* h, hfpf_row** rows, uint64_t* n_rows) { return hfpf_extract_filtered(h, nullptr, rows, n_rows); }

int hfpf_extract_filtered(hfpf_handle* h, const hfpf_extract_opts* opts, hfpf_row** rows, uint64_t* n_rows)
{
    if (!h || !rows || !n_rows) return HFPF_ERR_BAD_ARG;
    if (opts && opts->struct_size != sizeof(hfpf_extract_opts)) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    *rows = nullptr;
    *n_rows = 0;
    Tables& t = h->t;
    // (as in hfpf_clean: a local failure in front of the collective travels through the status gather, it does not skip it)
    int rc = HFPF_OK;
    if (hipError_t e_ = hipSetDevice(h->cfg.device)) rc = fail(h, HFPF_ERR_HIP, "hipSetDevice failed: %s", hipGetErrorString(e_));
    if (!rc) rc = poison_on_error(h, flush_pending_locked(h));  // host frames still waiting for their launch
    if (!rc) rc = check_usable(h);
    if (!rc) rc = read_counters(h);
    if (!rc) rc = poison_on_error(h, check_device_errors(h));
    if (rc && !h->dist_on) return rc;
    const unsigned long long* stats = t.stats;
    if (h->dist_on) {
        // Sum the ranks' private partial records (exact integer adds) into scratch; the partials stay intact.
        // Normal records are replicated, so every rank holds the same n and the same record ids -- checked: the count travels as
        // the payload of the status gather that makes every rank agree to enter the all-reduce (or to leave together).
        // The whole table is reduced, (n + 1) x 64 bytes = 125 MB at 2 M voxels: one collective per extract, a few ms over xGMI
        // against an extract that sorts and downloads the same records; a list of touched records would need a second collective.
        const uint64_t n_mine = h->h_ctr[C_NORMALS];
        const uint64_t words = (n_mine + 1) * kStatWords;
        if (!rc) rc = scratch(h, h->stats_total, words * 8);
        std::vector<unsigned long long> n_of(h->world, 0ull);
        if ((rc = dist_status_gather_locked(h, rc, n_mine, "statistics all-reduce of extract", n_of.data()))) return rc;
        for (int r = 0; r < h->world; r++)  // every rank sees the same payloads, so all of them take this exit or none does
            if (n_of[r] != n_of[0])
                return fail(h, HFPF_ERR_DIST, "rank %d holds %llu normal records, rank 0 %llu: the ranks did not run the same clean schedule", r,
                            (unsigned long long)n_of[r], (unsigned long long)n_of[0]);
        NCCLCHK(h, g_rccl.AllReduce(t.stats, h->stats_total.p, words, ncclUint64_, ncclSum_, (ncclComm_t_)h->comm, h->stream));
        stats = (const unsigned long long*)h->stats_total.p;
    }
    return extract_locked(h, stats, opts, rows, n_rows);
}

int hfpf_extract_with_stats(hfpf_handle* h, const void* dev_words, const void* dev_cwords, hfpf_row** rows, uint64_t* n_rows)
{
    if (!h || !rows || !n_rows || !dev_words) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    *rows = nullptr;
    *n_rows = 0;
    int rc = read_counters(h);
    if (rc) return rc;
    if ((rc = check_device_errors(h))) return rc;
    (void)dev_cwords;  // colour sums travel in words 5-7 of the statistics records since ABI 3
    return extract_locked(h, (const unsigned long long*)dev_words, nullptr, rows, n_rows);
}

int hfpf_stats_export(hfpf_handle* h, const void** dev_words, uint64_t* n_words, const void** dev_cwords, uint64_t* n_cwords)
{
    if (!h || !dev_words || !n_words) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    int rc = check_usable(h);
    if (!rc) rc = read_counters(h);
    if (!rc) rc = poison_on_error(h, check_device_errors(h));
    if (rc) return rc;
    *dev_words = h->t.stats;
    *n_words = (h->h_ctr[C_NORMALS] + 1) * kStatWords;
    if (dev_cwords) *dev_cwords = nullptr;  // colour sums are words 5-7 of the same records
    if (n_cwords) *n_cwords = 0;
    return HFPF_OK;
}

int hfpf_epoch_export(hfpf_handle* h, const void** dev_records, uint64_t* n_records)
{
    if (!h || !dev_records || !n_records) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    uint64_t n = 0;
    int rc = check_usable(h);
    if (!rc) rc = epoch_export_locked(h, &n, 0);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *dev_records = h->ex_send.p;
    *n_records = n;
    return HFPF_OK;
}

int hfpf_epoch_import(hfpf_handle* h, const void* dev_records, uint64_t n_records)
{
    if (!h || (!dev_records && n_records)) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    int rc = epoch_import_locked(h, dev_records, n_records);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));  // the caller may reuse / free the record buffer
    h->dirty = true;
    return HFPF_OK;
}

int hfpf_epoch_import_gathered(hfpf_handle* h, const void* dev_buffer, uint64_t slice_stride_bytes, int32_t world, int32_t my_rank, const uint64_t* counts)
{
    if (!h || !dev_buffer || !counts || world < 1 || my_rank < 0 || my_rank >= world) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    if (int rc0 = check_usable(h)) return rc0;
    std::vector<unsigned long long> c(counts, counts + world);
    int rc = import_gathered_locked(h, dev_buffer, slice_stride_bytes, world, my_rank, c.data());
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));  // the caller may reuse / free the buffer
    h->dirty = true;
    return HFPF_OK;
}

int hfpf_device_copy(hfpf_handle* h, void* dev_dst, const void* dev_src, uint64_t bytes)
{
    if (!h || (!dev_dst && bytes) || (!dev_src && bytes)) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (bytes) HIPCHK(h, hipMemcpyAsync(dev_dst, dev_src, (size_t)bytes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HFPF_OK;
}

int hfpf_dist_unique_id(void* id128)
{
    if (!id128) return HFPF_ERR_BAD_ARG;
    std::string err;
    if (load_rccl(err)) return fail(nullptr, HFPF_ERR_DIST, "%s", err.c_str());
    ncclUniqueId_ id;
    int r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess_) return fail(nullptr, HFPF_ERR_DIST, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    memcpy(id128, &id, sizeof id);
    return HFPF_OK;
}

int hfpf_dist_init(hfpf_handle* h, int rank, int world, const void* id128)
{
    if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->dist_on) return fail(h, HFPF_ERR_STATE, "hfpf_dist_init called twice");
    if (h->h_counts) { (void)hipHostFree(h->h_counts); h->h_counts = nullptr; }
    std::string err;
    if (load_rccl(err)) return fail(h, HFPF_ERR_DIST, "%s", err.c_str());
    ncclUniqueId_ id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t_ comm = nullptr;
    NCCLCHK(h, g_rccl.CommInitRank(&comm, world, id, rank));
    h->comm = comm;
    h->rank = rank;
    h->world = world;
    HIPCHK(h, hipHostMalloc((void**)&h->h_counts, (size_t)(world + 1) * 8, hipHostMallocDefault));
    {  // exchange buffers for epochs of up to 4 M newly occupied cells per rank exist from here on; larger epochs grow them
        const uint64_t recs = std::min<uint64_t>(h->t.max_occ, 4ull << 20);
        int rc;
        if ((rc = scratch(h, h->ex_counts, (size_t)(world + 1) * 8)) || (rc = scratch(h, h->ex_send, recs * sizeof(EpochRec))) ||
            (rc = scratch(h, h->ex_recv, (size_t)world * recs * sizeof(EpochRec))))
            return rc;
        h->ex_cap_records = recs;
    }
    h->dist_on = true;
    return HFPF_OK;
}

int hfpf_dist_info(hfpf_handle* h, int32_t* rank, int32_t* world)
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    int r = 0, w = 1;
    if (h->dist_on && h->comm) {  // what the communicator itself reports, not what the caller passed to hfpf_dist_init
        if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(h, HFPF_ERR_DIST, "librccl lacks ncclCommCount / ncclCommUserRank");
        NCCLCHK(h, g_rccl.CommCount((ncclComm_t_)h->comm, &w));
        NCCLCHK(h, g_rccl.CommUserRank((ncclComm_t_)h->comm, &r));
    }
    if (rank) *rank = r;
    if (world) *world = w;
    return HFPF_OK;
}

int hfpf_dist_disable(hfpf_handle* h)
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t_)h->comm);
    h->comm = nullptr;
    h->dist_on = false;
    h->rank = 0;
    h->world = 1;
    return HFPF_OK;
}

int hfpf_device_download(hfpf_handle* h, void* host_dst, const void* dev_src, uint64_t bytes)
{
    if (!h || !host_dst || !dev_src) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(host_dst, dev_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return HFPF_OK;
}

void hfpf_free_rows(hfpf_row* rows) { free(rows); }


}  // extern "C" (the helper below is a template)

namespace {
// Formats rows [0,n) with `fmt_row` on several host threads (one contiguous chunk each) and writes the chunks in
// order: ASCII formatting, not I/O, dominates the reference's savePCDFileASCII-style outputs.
template <typename F>
bool write_rows_parallel(FILE* f, uint64_t n, size_t bytes_per_row_hint, F fmt_row)
{
    if (n == 0) return true;
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_thr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(hw ? hw : 1, 16), n / 4096 + 1));
    std::vector<std::string> parts(n_thr);
    std::vector<std::thread> thr;
    const uint64_t per = (n + n_thr - 1) / n_thr;
    for (unsigned t = 0; t < n_thr; t++) {
        thr.emplace_back([&, t] {
            const uint64_t a = t * per, b = std::min<uint64_t>(n, a + per);
            std::string& s = parts[t];
            if (b > a) s.reserve((size_t)(b - a) * bytes_per_row_hint);
            char line[256];
            for (uint64_t i = a; i < b; i++) {
                const int len = fmt_row(i, line, sizeof line);
                if (len > 0) s.append(line, (size_t)len);
            }
        });
    }
    for (auto& t : thr) t.join();
    for (auto& s : parts)
        if (!s.empty() && fwrite(s.data(), 1, s.size(), f) != s.size()) return false;
    return true;
}
}  // namespace

extern "C" {

int hfpf_write_pcd(const hfpf_row* rows, uint64_t n, const char* path)
{
    if (!path || (!rows && n)) return HFPF_ERR_BAD_ARG;
    FILE* f = fopen(path, "w");
    if (!f) return HFPF_ERR_IO;
    // PCL PCDWriter::writeASCII layout for PointXYZRGBNormal (pcl::io::savePCDFileASCII, grid.hpp:485), precision 8.
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb normal_x normal_y normal_z curvature\n");
    fprintf(f, "SIZE 4 4 4 4 4 4 4 4\nTYPE F F F U F F F F\nCOUNT 1 1 1 1 1 1 1 1\n");
    fprintf(f, "WIDTH %llu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %llu\nDATA ascii\n", (unsigned long long)n, (unsigned long long)n);
    // the reference never writes rgb (grid.hpp:471-479): a default-constructed PCL point has r=g=b=0, a=255
    bool ok = write_rows_parallel(f, n, 96, [rows](uint64_t i, char* line, size_t cap) {
        const hfpf_row& r = rows[i];
        return snprintf(line, cap, "%.8g %.8g %.8g %u %.8g %.8g %.8g 0\n", r.x, r.y, r.z, 0xFF000000u | r.rgb, r.nx, r.ny, r.nz);
    });
    ok = ok && !ferror(f);
    return (fclose(f) == 0 && ok) ? HFPF_OK : HFPF_ERR_IO;
}

int hfpf_write_meta_csv(const hfpf_row* rows, uint64_t n, const char* path)
{
    if (!path || (!rows && n)) return HFPF_ERR_BAD_ARG;
    FILE* f = fopen(path, "w");
    if (!f) return HFPF_ERR_IO;
    fprintf(f, "Id,sdx,sdy,sdz,mean distance from normal, distance from normal sd, points in cylinder\n");  // grid.hpp:462 verbatim
    // default ostream float formatting = %g (6 significant digits), grid.hpp:478
    bool ok = write_rows_parallel(f, n, 80, [rows](uint64_t i, char* line, size_t cap) {
        const hfpf_row& r = rows[i];
        return snprintf(line, cap, "%llu,%g,%g,%g,%g,%g,%d\n", (unsigned long long)i, r.sdx, r.sdy, r.sdz, r.mean_dist, r.sd_dist, (int)r.count);
    });
    ok = ok && !ferror(f);
    return (fclose(f) == 0 && ok) ? HFPF_OK : HFPF_ERR_IO;
}

// downloadHQ / downloadClassified / download(XYZRGB) (grid.hpp:491-575; only referenced inside `#if 0`, node.cpp:399-437)
// as one writer over already extracted rows: PointXYZRGB cloud, optional count filter and colour coding.
int hfpf_write_pcd_xyzrgb(const hfpf_row* rows, uint64_t n, const char* path, uint32_t min_count, int32_t classify_threshold, int32_t white)
{
    if (!path || (!rows && n)) return HFPF_ERR_BAD_ARG;
    uint64_t kept = 0;
    for (uint64_t i = 0; i < n; i++) kept += rows[i].count >= min_count ? 1 : 0;  // `if(data->count<threshold) continue;` grid.hpp:561
    FILE* f = fopen(path, "w");
    if (!f) return HFPF_ERR_IO;
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n");
    fprintf(f, "WIDTH %llu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %llu\nDATA ascii\n", (unsigned long long)kept, (unsigned long long)kept);
    for (uint64_t i = 0; i < n; i++) {
        const hfpf_row& r = rows[i];
        if (r.count < min_count) continue;
        uint32_t rgb = white ? 0x00FFFFFFu : r.rgb;                                                // pt.r=g=b=255, grid.hpp:527-529,558-560
        if (classify_threshold >= 0 && (int64_t)r.count > (int64_t)classify_threshold) rgb = 0x00FF0000u;  // g=b=0, grid.hpp:530-534
        fprintf(f, "%.8g %.8g %.8g %u\n", r.x, r.y, r.z, 0xFF000000u | rgb);
    }
    const bool ok = !ferror(f);
    return (fclose(f) == 0 && ok) ? HFPF_OK : HFPF_ERR_IO;
}

// Same fields as hfpf_write_pcd with DATA binary (40 bytes/point): for outputs where ASCII formatting would dominate.
int hfpf_write_pcd_binary(const hfpf_row* rows, uint64_t n, const char* path)
{
    if (!path || (!rows && n)) return HFPF_ERR_BAD_ARG;
    FILE* f = fopen(path, "wb");
    if (!f) return HFPF_ERR_IO;
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb normal_x normal_y normal_z curvature\n");
    fprintf(f, "SIZE 4 4 4 4 4 4 4 4\nTYPE F F F U F F F F\nCOUNT 1 1 1 1 1 1 1 1\n");
    fprintf(f, "WIDTH %llu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %llu\nDATA binary\n", (unsigned long long)n, (unsigned long long)n);
    std::vector<char> buf;
    buf.reserve(32 * 4096);
    for (uint64_t i = 0; i < n; i++) {
        const hfpf_row& r = rows[i];
        const float zero = 0.f;
        const uint32_t rgba = 0xFF000000u | r.rgb;
        const void* fields[8] = {&r.x, &r.y, &r.z, &rgba, &r.nx, &r.ny, &r.nz, &zero};
        for (int k = 0; k < 8; k++) buf.insert(buf.end(), (const char*)fields[k], (const char*)fields[k] + 4);
        if (buf.size() >= 32 * 4096 || i + 1 == n) {
            fwrite(buf.data(), 1, buf.size(), f);
            buf.clear();
        }
    }
    const bool ok = !ferror(f);
    return (fclose(f) == 0 && ok) ? HFPF_OK : HFPF_ERR_IO;
}

int hfpf_clear(hfpf_handle* h)
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    h->pend_n = 0;  // host frames still waiting for their launch would be wiped with the rest
    // how far the session got (the counters as the device has them now); a handle whose stream has failed is reset in full
    uint64_t bricks_used = ~0ull, normals_used = ~0ull;
    if (read_counters(h) == HFPF_OK) {
        bricks_used = h->h_ctr[C_BRICKS];
        normals_used = h->h_ctr[C_NORMALS];
    }
    int rc = reset_state(h, bricks_used, normals_used);
    h->dirty = true;  // clearVoxels sets state_changed, grid.hpp:169
    return rc;
}

int hfpf_sync(hfpf_handle* h)
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    int rc = read_counters(h);
    if (rc) return rc;
    return poison_on_error(h, check_device_errors(h));
}

int hfpf_get_counters(hfpf_handle* h, hfpf_counters* out)
{
    if (!h || !out) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    int rc = read_counters(h);
    if (rc) return rc;
    const unsigned long long* c = h->h_ctr;
    out->points_presented = c[C_PRESENTED];
    out->points_zclip_pass = c[C_ZPASS];
    out->points_in_bbox = c[C_INBOX];
    out->points_buffered = c[C_LOG];
    out->dep_pairs_tested = c[C_DEP_TESTED];
    out->dep_pairs_member = c[C_DEP_MEMBER];
    out->voxels_occupied = c[C_OCC];
    out->voxels_with_normal = c[C_NORMALS];
    out->bricks_allocated = std::min<uint64_t>(c[C_BRICKS], h->t.max_bricks);
    out->registrations = c[C_REG];
    out->dep_entries = c[C_DEP];  /* includes relocated (garbage) lists until the next compaction */
    out->frames_integrated = h->frames_integrated;
    out->clean_passes = h->clean_passes;
    out->device_bytes = h->device_bytes;
    out->replay_members = c[C_REPLAY_MEMBER];
    out->points_direct = c[C_BUFFERED];
    out->table_misses = c[C_TABLE_MISS];
    out->update_extra_rounds = c[C_UPD_ROUNDS];
    return HFPF_OK;
}

int hfpf_get_occupied(hfpf_handle* h, int32_t* xyz, uint64_t cap, uint64_t* n_out)
{
    if (!h || !n_out) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    int rc = read_counters(h);
    if (rc) return rc;
    const uint64_t n = std::min<uint64_t>(h->h_ctr[C_OCC], h->t.max_occ);
    *n_out = n;
    if (!xyz || n == 0) return HFPF_OK;
    if ((rc = scratch(h, h->keys_a, n * 8))) return rc;
    if ((rc = scratch(h, h->keys_b, n * 8))) return rc;
    hipLaunchKernelGGL(k_occupied_keys, dim3(blocks_for(n, 256)), dim3(256), 0, h->stream, h->g, h->t, n, (uint64_t*)h->keys_a.p);
    HIPCHK(h, hipGetLastError());
    if ((rc = sort_keys_u64(h, (uint64_t*)h->keys_a.p, (uint64_t*)h->keys_b.p, n))) return rc;
    std::vector<uint64_t> keys(n);
    HIPCHK(h, hipMemcpyAsync(keys.data(), h->keys_b.p, n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (uint64_t i = 0; i < std::min(n, cap); i++) key_coords(h->g, keys[i], xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return HFPF_OK;
}

int hfpf_device_alloc(hfpf_handle* h, uint64_t bytes, void** dev_ptr)
{
    if (!h || !dev_ptr) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMalloc(dev_ptr, (size_t)std::max<uint64_t>(bytes, 1)));
    return HFPF_OK;
}

int hfpf_device_free(hfpf_handle* h, void* dev_ptr)
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(dev_ptr));
    return HFPF_OK;
}

int hfpf_device_upload(hfpf_handle* h, void* dev_dst, const void* host_src, uint64_t bytes)
{
    if (!h || !dev_dst || !host_src) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpy(dev_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice));
    return HFPF_OK;
}

int hfpf_kernel_timing(hfpf_handle* h, int enable)
{
    if (!h) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    int rc = resolve_timing(h);
    if (rc) return rc;
    h->timing = enable != 0;
    h->timing_detail = enable == 2;
    if (enable) {
        for (int k = 0; k < 3; k++) h->t_detail_ms[k] = 0, h->n_detail[k] = 0;
        h->t_integrate_ms = 0;
        h->n_integrate_launches = 0;
        h->t_clean_ms = 0;
        h->n_clean_timed = 0;
    }
    return HFPF_OK;
}

int hfpf_get_kernel_time(hfpf_handle* h, int kernel_id, double* total_ms, uint64_t* launches)
{
    if (!h || kernel_id < 0 || kernel_id > 4) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rcf = flush_pending_locked(h)) return rcf;  // host frames still waiting for their launch
    int rc = resolve_timing(h);
    if (rc) return rc;
    if (kernel_id >= 2) {
        if (total_ms) *total_ms = h->t_detail_ms[kernel_id - 2];
        if (launches) *launches = h->n_detail[kernel_id - 2];
        return HFPF_OK;
    }
    if (total_ms) *total_ms = kernel_id == 0 ? h->t_integrate_ms : h->t_clean_ms;
    if (launches) *launches = kernel_id == 0 ? h->n_integrate_launches : h->n_clean_timed;
    return HFPF_OK;
}

// ---- leaf probes (include/hfpf_probe.h) ----------------------------------------------------------
#define PROBE_UP(buf, src, bytes)                                                  \
    if ((rc = scratch(h, buf, (bytes)))) return rc;                                \
    HIPCHK(h, hipMemcpyAsync(buf.p, (src), (bytes), hipMemcpyHostToDevice, h->stream));
#define PROBE_DOWN(dst, buf, bytes) HIPCHK(h, hipMemcpyAsync((dst), buf.p, (bytes), hipMemcpyDeviceToHost, h->stream));

int hfpf_probe_points(hfpf_handle* h, const double pose[12], const float* xyz, uint64_t n, float* q_out, int32_t* idx_out, uint8_t* flags_out)
{
    if (!h || !pose || !xyz || !q_out || !idx_out || !flags_out) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (n == 0) return HFPF_OK;
    int rc;
    PROBE_UP(h->probe_a, pose, 12 * sizeof(double));
    PROBE_UP(h->probe_b, xyz, n * 12);
    if ((rc = scratch(h, h->probe_c, n * 12))) return rc;
    if ((rc = scratch(h, h->probe_d, n * 12))) return rc;
    if ((rc = scratch(h, h->probe_e, n))) return rc;
    hipLaunchKernelGGL(k_probe_points, dim3(blocks_for(n, 256)), dim3(256), 0, h->stream, h->g, (const double*)h->probe_a.p, (const float*)h->probe_b.p, n,
                       (float*)h->probe_c.p, (int32_t*)h->probe_d.p, (uint8_t*)h->probe_e.p);
    HIPCHK(h, hipGetLastError());
    PROBE_DOWN(q_out, h->probe_c, n * 12);
    PROBE_DOWN(idx_out, h->probe_d, n * 12);
    PROBE_DOWN(flags_out, h->probe_e, n);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HFPF_OK;
}

int hfpf_probe_normals(hfpf_handle* h, uint64_t n, const int32_t* cells, const uint8_t* occ, const float* vps, float* normals_out, int32_t* totals_out)
{
    if (!h || !cells || !occ || !vps || !normals_out || !totals_out) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (n == 0) return HFPF_OK;
    int rc;
    PROBE_UP(h->probe_a, cells, n * 12);
    PROBE_UP(h->probe_b, occ, n * 125);
    PROBE_UP(h->probe_c, vps, n * 12);
    if ((rc = scratch(h, h->probe_d, n * 12))) return rc;
    if ((rc = scratch(h, h->probe_e, n * 4))) return rc;
    hipLaunchKernelGGL(k_probe_normals, dim3(blocks_for(n, 64)), dim3(64), 0, h->stream, h->g, n, (const int32_t*)h->probe_a.p, (const uint8_t*)h->probe_b.p,
                       (const float*)h->probe_c.p, (float*)h->probe_d.p, (int32_t*)h->probe_e.p);
    HIPCHK(h, hipGetLastError());
    PROBE_DOWN(normals_out, h->probe_d, n * 12);
    PROBE_DOWN(totals_out, h->probe_e, n * 4);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HFPF_OK;
}

int hfpf_probe_project(hfpf_handle* h, uint64_t n, const float* pts, const float* centres, const float* normals, float* proj_out, double* dist_out,
                       uint8_t* member_out)
{
    if (!h || !pts || !centres || !normals || !proj_out || !dist_out || !member_out) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (n == 0) return HFPF_OK;
    int rc;
    PROBE_UP(h->probe_a, pts, n * 12);
    PROBE_UP(h->probe_b, centres, n * 12);
    PROBE_UP(h->probe_c, normals, n * 12);
    if ((rc = scratch(h, h->probe_d, n * 12))) return rc;
    if ((rc = scratch(h, h->probe_e, n * 8))) return rc;
    if ((rc = scratch(h, h->probe_f, n))) return rc;
    hipLaunchKernelGGL(k_probe_project, dim3(blocks_for(n, 256)), dim3(256), 0, h->stream, h->g, n, (const float*)h->probe_a.p, (const float*)h->probe_b.p,
                       (const float*)h->probe_c.p, (float*)h->probe_d.p, (double*)h->probe_e.p, (uint8_t*)h->probe_f.p);
    HIPCHK(h, hipGetLastError());
    PROBE_DOWN(proj_out, h->probe_d, n * 12);
    PROBE_DOWN(dist_out, h->probe_e, n * 8);
    PROBE_DOWN(member_out, h->probe_f, n);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HFPF_OK;
}

int hfpf_probe_trig(hfpf_handle* h, uint64_t n, const float* y, const float* x, float* atan2_out, float* cos_out, float* sin_out)
{
    if (!h || !y || !x || !atan2_out || !cos_out || !sin_out) return HFPF_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mtx);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (n == 0) return HFPF_OK;
    int rc;
    PROBE_UP(h->probe_a, y, n * 4);
    PROBE_UP(h->probe_b, x, n * 4);
    if ((rc = scratch(h, h->probe_c, n * 4))) return rc;
    if ((rc = scratch(h, h->probe_d, n * 4))) return rc;
    if ((rc = scratch(h, h->probe_e, n * 4))) return rc;
    hipLaunchKernelGGL(k_probe_trig, dim3(blocks_for(n, 256)), dim3(256), 0, h->stream, n, (const float*)h->probe_a.p, (const float*)h->probe_b.p,
                       (float*)h->probe_c.p, (float*)h->probe_d.p, (float*)h->probe_e.p);
    HIPCHK(h, hipGetLastError());
    PROBE_DOWN(atan2_out, h->probe_c, n * 4);
    PROBE_DOWN(cos_out, h->probe_d, n * 4);
    PROBE_DOWN(sin_out, h->probe_e, n * 4);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HFPF_OK;
}

}  // extern "C"
