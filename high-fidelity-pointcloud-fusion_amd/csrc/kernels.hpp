// csrc/kernels.hpp -- HIP kernels of the fusion path (gfx950, 64-wide waves).
//
//   K1 k_integrate        decode + z-clip + SE(3) + bbox clip + voxel index + brick claim + first occupancy; what is left to do
//                         with a point (buffer it / update its cell's dependants) is parked in its brick's bin, region A
//                         (cell has a normal) or B (it has none) (node.cpp:190-214,251-255,289; grid.hpp:194-243)
//   K2 k_update           region A: dependant updates per brick, LDS-staged statistic records, one flush per record (grid.hpp:244-277)
//      k_buffer           region B: the brick's buffered points go to the point log as one run, chained per cell (grid.hpp:210-211,230,239)
//      (k_integrate<.., BIN=false>, and lanes whose bin region is full, keep the direct forms: log append + one memory-side
//      atomic segment per (point, dependant) pair)
//      k_bin_plan/clamp   per-brick bin regions of the next launch from the previous launch's demand
//   K3 k_gate             5x5x5 occupancy count and the >gate test            (grid.hpp:322-352)
//   K4 k_normal           plane fit on occupied neighbour centres, orientation (grid.hpp:356-398)
//   K5 k_register         +-K line walk, dependant registration (grid.hpp:403-417,443-449)
//      k_link_log         chains the point-log entries the direct form appended since the last clean (4 interleaved chains per cell)
//      k_replay           buffer replay of the cells that gained registrants (grid.hpp:418-440)
//      k_depinc_*/k_dep_* incremental update / compacting rebuild of the per-cell dependant table
//   K6 k_extract_*        ordered compaction of normal_found voxels            (grid.hpp:463-480)
//      k_epoch_*          multi-GPU exchange of newly occupied cells
#pragma once
#include "stats.hpp"

namespace hfpf {

struct FrameLayout {
    uint32_t point_step, off_x, off_y, off_z, off_rgb;
};

// ------------------------------------------------------------------------------------------------
// K1.  Persistent grid: block b walks tiles b, b+gridDim.x, ... of 256 consecutive points; a tile never
// straddles two frames, so the pose stays in scalar registers.  Loop trip counts are wave-uniform, which
// keeps the ballot/shuffle helpers convergent.
//
// Hot-counter discipline: one same-address device atomic costs ~12 ns at the memory side whatever the
// issuer (MI355X_MICROARCH.md "fanin"), so diagnostics stay in registers until the block retires and the
// point log is striped over kLogRegions append regions with one counter per 128-byte line.
//
// Dependant updates, direct form (BIN = false, and the fallback of the binned form): every (point, dependant) pair
// inside the 1 mm cylinder adds 5 int64 words (8 with colour) to ONE 64-byte statistics record.  A lane-per-pair loop would issue
// as many fully scattered atomic instructions per round (320 memory-side requests); instead the member lanes park their deltas
// in a per-wave LDS queue and the wave replays the queue with 8 lanes per record, so one wave-instruction carries 8 whole
// records as one 64-byte segment each.  That form saturates the chip's ~20 G requests/s atomic unit (41 M per launch).
// Binned form (BIN = true, default): the point is parked in its brick's bin and k_update does the pairs brick by brick.
#ifndef HFPF_REPLAY_B
#define HFPF_REPLAY_B 3  // registrants per chain walk held in registers by k_replay
#endif
constexpr int kLogRegions = 64;
#ifndef HFPF_LOG_CHAINS
#define HFPF_LOG_CHAINS 4
#endif
constexpr uint32_t kLogChains = HFPF_LOG_CHAINS;  // interleaved chains per cell (= kChains below: entry e joins chain e mod kLogChains of its cell)
constexpr uint32_t kLogUnlinked = 0x80000000u;  // log entry .w = slot | this bit until the entry is chained (then: index of the next entry)
#ifndef HFPF_REG_TILES
#define HFPF_REG_TILES 8  // 256-voxel tiles one workgroup of k_register takes per list reservation
#endif
constexpr int kRegTilesLarge = HFPF_REG_TILES, kRegTilesSmall = 2;
constexpr int kImportTiles = HFPF_REG_TILES;  // k_epoch_import: tiles per workgroup (one occ_list reservation)
// Passed instead of an element count: the kernel reads the exact count from its device counter and the host sizes the grid
// from an upper bound, which saves a host round trip between two kernels of a clean pass.
constexpr uint64_t kCountOnDevice = ~0ull;
constexpr int kListTiles = 4;  // same idea for the k_depinc_* list builders (not k_gate: it is latency-heavy per cell and needs every workgroup it can get)
typedef float vf4 __attribute__((ext_vector_type(4)));  // native vector type (the nontemporal builtins do not take HIP's float4)

// Wave-cooperative flush of statistic deltas: lanes with `member` park their delta (5 words, + 3 colour sums, + record id)
// in the wave's LDS queue, kQueueRows at a time, and the wave replays the queue with 8 lanes per record, so one
// wave-instruction carries 8 whole 64-byte records (one memory-side atomic segment each).  Convergent (all 64 lanes must call).
constexpr int kQueueStride = 11;  // u64 words per queued delta; odd stride spreads LDS banks
constexpr int kQueueRows = 8;     // deltas staged per round (the queue is 704 bytes per wave: the direct form is the rare path)
template <bool COLOR>
__device__ __forceinline__ void wave_flush_members(const Tables& t, unsigned long long* q, bool member, const StatDeltaT<COLOR>& d, uint32_t sid)
{
    constexpr int W = COLOR ? 8 : kStatUsed;
    const unsigned long long mm = __ballot(member);
    if (mm == 0) return;
    const uint32_t lane = lane_id();
    const uint32_t n_mem = (uint32_t)__popcll(mm);
    const uint32_t rank = (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
    for (uint32_t r0 = 0; r0 < n_mem; r0 += (uint32_t)kQueueRows) {  // wave-uniform trip count
        if (member && rank >= r0 && rank < r0 + (uint32_t)kQueueRows) {
            unsigned long long* r = q + (rank - r0) * kQueueStride;
#pragma unroll
            for (int w = 0; w < kStatUsed; w++) r[w] = (unsigned long long)d.v[w];
            if constexpr (COLOR) {
                r[SW_R] = (unsigned long long)d.rgb[0];
                r[SW_G] = (unsigned long long)d.rgb[1];
                r[SW_B] = (unsigned long long)d.rgb[2];
            }
            r[8] = sid;
        }
        // same-wave LDS hand-off: DS operations of one wave execute in order; the fences only pin the compiler
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const uint32_t w = lane & 7u, row = lane >> 3;
            if (r0 + row < n_mem && w < (uint32_t)W) {
                const unsigned long long* r = q + row * kQueueStride;
                atomicAdd(&t.stats[(uint64_t)r[8] * kStatWords + w], r[w]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

constexpr int kOccStage = 256;  // newly occupied cells ONE WAVE of k_integrate stages before appending them to occ_list
// Append the wave's staged slots with ONE atomic on C_OCC.  Wave-level only (no workgroup barrier, so waves never have to agree on
// when to flush); `n` is wave-uniform.  Convergent for the wave.
__device__ __forceinline__ void flush_occ_stage(const Tables& t, const uint32_t* s_occ, uint32_t n)
{
    const uint32_t lane = lane_id();
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&t.ctr[C_OCC], (unsigned long long)n);
    base = __shfl(base, 0);
    // same-wave LDS hand-off: DS operations of one wave execute in order; the fences only pin the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bool overflow = false;
    for (uint32_t k = lane; k < n; k += 64) {
        const unsigned long long oi = base + k;
        if (oi < t.max_occ) t.occ_list[oi] = s_occ[k];
        else overflow = true;
    }
    if (overflow) atomicOr(&t.ctr[C_ERR], (unsigned long long)E_OCC);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inclusive prefix sum over the 64 lanes of a wave on the DPP data path: four row shifts and two row broadcasts, six VALU
// instructions and no LDS traffic (__shfl_up is a ds_bpermute per step: ~60 cycles of LDS-crossbar latency each, in a chain).
#ifndef HFPF_DPP_SCAN
#define HFPF_DPP_SCAN 1
#endif
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
#if HFPF_DPP_SCAN
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1, zero shifted in
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
#else
    const uint32_t lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = __shfl_up(v, o);
        if (lane >= (uint32_t)o) v += a;
    }
    return v;
#endif
}

__device__ __forceinline__ DepEntry make_dep_entry(const Tables& t, uint32_t nid)  // (nv_line holds every record's line in the layout of a dependant entry)
{
    return *reinterpret_cast<const DepEntry*>(&t.nv_line[2 * (uint64_t)nid]);
}
// A cell gained its first dependant: set its bit in the brick's flag line (read by k_integrate).
__device__ __forceinline__ void set_dep_flag(const Tables& t, uint32_t slot)
{
    uint64_t plane, bit;
    slot_plane_bit(slot, plane, bit);
    atomicOr(reinterpret_cast<unsigned long long*>(&t.nd_mask[plane * 2 + 1]), (unsigned long long)bit);
}
// The ONE dependant a cell can carry while it is unoccupied (grid.hpp:443-449: the last registrant wins) lives in pre_dep[slot]
// alone -- no list entry, no flag bit, nothing a clean pass has to write per unoccupied target but the atomicMax that settles the
// winner (most registration targets are unoccupied: 8 M of the 13.7 M in the bench's first pass, and on 0.5 mm voxels the slot
// arrays they used to touch fall out of the caches).  A cell that has become occupied gets the entry filed as its dependant list
// at the head of the next clean pass (k_materialize_new: list entry, info word, flag bit, a place in prereg_list for compacting
// rebuilds).  In between -- the cell was occupied in the running epoch -- whoever reads dependant lists while points arrive
// (k_update_cells, k_update, k_integrate_overflow, the un-binned k_integrate) takes an EMPTY list of an occupied cell to mean
// "look at pre_dep": the one entry is the record's own line, which nv_line keeps in the layout of a dependant entry.
__device__ __forceinline__ void pre_list_entry(const Tables& t, uint32_t nid, float4& e0, float4& e1)  // = the two halves of make_dep_entry(t, nid)
{
    e0 = t.nv_line[2 * (uint64_t)nid];
    e1 = t.nv_line[2 * (uint64_t)nid + 1];
}

// The direct forms of buffering (grid.hpp:205-243) and of the dependant update (grid.hpp:244-277) for one point per lane: the whole
// path of the un-binned engine (BIN = false, inline in k_integrate), and in the binned one what k_integrate_overflow does with the
// few points that found no room in their brick's bin.  Convergent for the wave (every lane calls; `todo` says which have a point).
constexpr uint32_t kOvfHasNormal = 0x80000000u, kOvfHasDeps = 0x40000000u;  // flags beside the frame id (< 2^23, host-checked) in Tables::ovf_aux
template <bool COLOR, bool BIN>
__device__ __forceinline__ void direct_buffer(const Tables& t, unsigned long long* log_ctr, const uint64_t log_base, const F3 p, const uint32_t slot, const uint32_t b,
                                              const uint32_t fid, const uint32_t rgb, const bool todo, const bool has_n, uint32_t& c_buf)
{
    // direct buffering; the viewpoint latch (smallest frame id that touched the cell, grid.hpp:229,238) is only ever
    // read before the normal exists
    const bool buf = todo && !has_n;
    if (buf && fid < t.first_frame[slot]) atomicMin(&t.first_frame[slot], fid);
    const unsigned long long li = wave_reserve(log_ctr, buf);
    if (buf) {
        if (li < t.log_region_cap) {
            const uint64_t e = log_base + li + 1;
            // Binned form: this is the rare lane whose bin region was full, so it chains its entry right here (one returning
            // atomic) and no pass over the log is needed afterwards.  Direct form (every point comes this way): .w carries the
            // marked slot until k_link_log chains the entries of the epoch in one go.
            uint32_t link = slot | kLogUnlinked;
            if (BIN) {
                link = atomicExch(&t.buf_head[(uint64_t)slot * kLogChains + ((uint32_t)e & (kLogChains - 1))], (uint32_t)e);
                if (!(t.run_cnt[b] & 0x100u)) atomicOr(&t.run_cnt[b], 0x100u);  // an entry outside the brick's runs: its replay walks the chains
            }
            t.log_pt[e] = make_float4(p.x, p.y, p.z, __uint_as_float(link));
            if (COLOR) t.log_rgb[e] = rgb;
        } else {
            atomicOr(&t.ctr[C_ERR], (unsigned long long)E_LOG);
        }
    }
    c_buf += buf;
}

template <bool COLOR, bool BIN>
__device__ __forceinline__ void direct_forms(const GridParams& g, const Tables& t, unsigned long long* q, unsigned long long* log_ctr, const uint64_t log_base,
                                             const F3 p, const uint32_t slot, const uint32_t b, const uint32_t fid, const uint32_t rgb, const bool todo,
                                             const bool has_n, const bool has_d, const uint32_t pre_nid, uint32_t& c_buf, uint32_t& c_tested, uint32_t& c_member)
{
    direct_buffer<COLOR, BIN>(t, log_ctr, log_base, p, slot, b, fid, rgb, todo, has_n, c_buf);

    // direct dependant updates.  pre_nid != 0: the cell was occupied in the running epoch and has no list yet, only the dependant it
    // was given while unoccupied (pre_list_entry): the point is tested against that record's own line.
    const bool direct = todo && has_d;
    uint32_t cnt = 0;
    uint64_t off = 0;
    if (direct) {
        if (pre_nid) {
            cnt = 1;
        } else {
            const uint64_t info = t.info[slot];
            cnt = (uint32_t)((info >> kDepCntShift) & kDepCntMask);
            off = info >> kDepOffShift;
        }
    }
    uint32_t max_cnt = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_cnt = max(max_cnt, (uint32_t)__shfl_xor((int)max_cnt, o));
    for (uint32_t j = 0; j < max_cnt; j++) {
        bool member = false;
        StatDeltaT<COLOR> d;
        stat_delta_zero(d);
        uint32_t sid = 0;
        if (j < cnt) {
            const DepEntry e = pre_nid ? make_dep_entry(t, pre_nid) : t.dep[off + j];
            float sp, distf;
            c_tested++;
            if (line_member(g, p, F3{e.ax, e.ay, e.az}, F3{e.abx, e.aby, e.abz}, e.dd, sp, distf)) {
                member = true;
                c_member++;
                sid = e.sid;
                stat_delta_add(d, pair_delta(g, sp, distf), rgb);
            }
        }
        wave_flush_members(t, q, member, d, sid);
    }
}

// Grid constants and table descriptor of k_integrate as ONE struct and its FIRST argument, so that a field's place in the kernarg
// segment is offsetof(IntegrateArgs, field).  (The frame, pose and frame-id pointers stay separate __restrict__ arguments: that
// is what lets the compiler read the wave-uniform pose with scalar loads.)
struct IntegrateArgs {
    GridParams g;
    Tables t;
};
// The table descriptor as the RARE paths of the tile loop see it (a new brick, a newly occupied cell's list flush, a point
// without room in its bin, the one lane a frame that files the viewpoint): read from the kernarg segment where it is needed.
// The compiler treats by-value kernel arguments as loop invariants and keeps every field the loop mentions anywhere in scalar
// registers for its whole length -- some forty table bases and limits here, 37 of which it then parked in VGPR lanes, with a
// v_readlane / v_writelane wherever the tile body wanted one back.  The empty asm hides where the pointer comes from, so the
// scalar loads behind it stay inside the branch they are written in.
typedef const __attribute__((address_space(4))) char* kernarg_ptr;
__device__ __forceinline__ const Tables& kernarg_tables()
{
    kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const Tables*)(p + offsetof(IntegrateArgs, t));
}

// Waves per SIMD: the tile loop is a chain of dependent table round trips, so resident waves are what hides them.  Seven for both
// forms.  The binned one (direct forms moved out to k_integrate_overflow) needs only 59 VGPRs, but 94 SGPRs a wave keep it at seven
// workgroups per CU all the same; asking the compiler for eight shrinks its SGPR budget (94 -> 78, 51 spills instead of 37) for
// nothing.  The un-binned form fits seven in 72 VGPRs without scratch.
#ifndef HFPF_INT_WAVES
#define HFPF_INT_WAVES 7
#endif
#ifndef HFPF_INT_WAVES_BIN
#define HFPF_INT_WAVES_BIN 7
#endif
template <bool PACKED16, bool COLOR, bool BIN>
__global__ __launch_bounds__(256, BIN ? HFPF_INT_WAVES_BIN : HFPF_INT_WAVES) void k_integrate(const IntegrateArgs A, const uint8_t* __restrict__ frames,
                                                   const uint64_t frame_stride, const uint32_t n_pts, const uint32_t n_frames,
                                                   const FrameLayout lay, const double* __restrict__ poses,
                                                   const uint32_t* __restrict__ frame_ids, const uint32_t row_w, const uint32_t log_rot,
                                                   const uint32_t probe, const uint32_t pre_possible)
{
    // pre_possible: a clean pass has run, so unoccupied cells may carry a dependant (k_materialize_new)
    const GridParams& g = A.g;
    const Tables& t = A.t;
    // probe != 0: dry run of the batch's first frames for a session that has no bin plan yet -- transform, index, claim the
    // bricks and record the per-region demand; nothing else is touched (the frames come again in the real launch).
    __shared__ unsigned long long queue[4][kQueueRows * kQueueStride];
    __shared__ unsigned int blk_ctr[6];
    __shared__ uint32_t s_occ_all[4][kOccStage];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    unsigned long long* q = queue[wave];
    uint32_t* s_occ = s_occ_all[wave];
    uint32_t occ_n = 0;  // wave-uniform: slots staged by this wave
    if (threadIdx.x < 6) blk_ctr[threadIdx.x] = 0;
    __syncthreads();

    const uint32_t tiles_per_frame = (n_pts + 255u) >> 8;
    const uint64_t n_tiles = (uint64_t)tiles_per_frame * n_frames;  // < 2^32 (host-checked)
    // log_rot changes from launch to launch, so that launches with fewer than kLogRegions workgroups (small frames through
    // hfpf_integrate, one frame per call) still fill every append region of the log
    const uint32_t region = (blockIdx.x + log_rot) & (kLogRegions - 1);
    unsigned long long* log_ctr = &t.log_ctr[region * 16];
    const uint64_t log_base = (uint64_t)region * t.log_region_cap;
    uint32_t c_present = 0, c_z = 0, c_in = 0, c_buf = 0, c_tested = 0, c_member = 0;

    // Where tile `tl` sits: frame and point index of this thread.  row_w != 0 (host-checked: organised frame, width and rows
    // multiples of 16): the tile is a 16x16-pixel patch, whose points fall into ~2x2 bricks and half as many table lines as a
    // 256-pixel run of one image row; each wave takes one 8x8 quadrant of the patch (8 pixels x 16 B = one 128-byte line per
    // image row).  32-bit arithmetic: the host keeps n_tiles below 2^32.
    auto locate = [&](uint32_t tl, uint32_t& f_out, uint32_t& i_out) {
        f_out = tl / tiles_per_frame;
        const uint32_t tile_in_frame = tl - f_out * tiles_per_frame;
        i_out = tile_in_frame * 256u + threadIdx.x;
        if (row_w) {
            const uint32_t tiles_x = row_w >> 4, ty = tile_in_frame / tiles_x, tx = tile_in_frame - ty * tiles_x;
            const uint32_t px = ((threadIdx.x >> 6) & 1u) * 8u + (threadIdx.x & 7u), py = (threadIdx.x >> 7) * 8u + ((threadIdx.x >> 3) & 7u);
            i_out = (ty * 16u + py) * row_w + tx * 16u + px;
        }
    };
    // The frame read of the NEXT tile is issued before this tile's table lookups (packed records only): the stream from HBM
    // is the longest latency of a tile and nothing in the tile depends on it but its own first instruction.
    uint32_t pre_f = 0, pre_i = 0;
    vf4 pre = {0.f, 0.f, 0.f, 0.f};
    const uint32_t n_tiles32 = (uint32_t)n_tiles;
    if (blockIdx.x < n_tiles32) {
        locate(blockIdx.x, pre_f, pre_i);
        if (PACKED16 && pre_i < n_pts)
            pre = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(frames + (uint64_t)pre_f * frame_stride) + pre_i);  // read once
    }
    for (uint32_t tile = blockIdx.x; tile < n_tiles32; tile += gridDim.x) {
        const uint32_t f = pre_f, i = pre_i;
        const vf4 nv = pre;
        if (tile + gridDim.x < n_tiles32) {
            locate(tile + gridDim.x, pre_f, pre_i);
            if (PACKED16 && pre_i < n_pts)
                pre = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(frames + (uint64_t)pre_f * frame_stride) + pre_i);
        }
        double T[12];
#pragma unroll
        for (int k = 0; k < 12; k++) T[k] = poses[12 * f + k];
        const uint32_t fid = frame_ids[f];
        if (i == 0) {  // viewpoint = float(translation), node.cpp:290
            const Tables& tr = BIN ? kernarg_tables() : t;
            float* vp = tr.frame_vp + 3 * (uint64_t)fid;
            vp[0] = (float)T[3];
            vp[1] = (float)T[7];
            vp[2] = (float)T[11];
            if (!probe) {  // the frames this rank has integrated: what the next epoch exchange sends viewpoints for
                const unsigned long long fi = atomicAdd(&tr.ctr[C_FRAMES], 1ull);
                if (fi < tr.max_frames) tr.frame_list[fi] = fid;  // (ids are below max_frames: the list only fills up when ids repeat, and then they are in it)
            }
        }
        const uint8_t* __restrict__ base = frames + (uint64_t)f * frame_stride;
        bool act = i < n_pts;
        float x = 0.f, y = 0.f, z = 0.f;
        uint32_t rgb = 0;
        if (act) {
            if (PACKED16) {
                x = nv.x;
                y = nv.y;
                z = nv.z;
                rgb = __float_as_uint(nv.w);
            } else {
                const uint8_t* rec = base + (uint64_t)i * lay.point_step;
                x = *reinterpret_cast<const float*>(rec + lay.off_x);
                y = *reinterpret_cast<const float*>(rec + lay.off_y);
                z = *reinterpret_cast<const float*>(rec + lay.off_z);
                rgb = *reinterpret_cast<const uint32_t*>(rec + lay.off_rgb);
            }
        }
        c_present += act;
        act = act && zclip_pass(g, z);
        c_z += act;
        const F3 p = transform_point(T, x, y, z);
        int32_t ix, iy, iz;
        voxel_coords(g, p, ix, iy, iz);
        // A NaN coordinate passes validPoints in the reference and then indexes out of bounds (crash); dropped here.
        act = act && valid_point(g, p) && ix != INT_MIN && iy != INT_MIN && iz != INT_MIN;
        c_in += act;

        const uint32_t bidx = act ? brick_index(g, ix, iy, iz) : 0u;
        uint32_t dir_word = 0;
        if (act) dir_word = t.dir[bidx];  // issued here, needed after the grouping loop below
        // Lanes of the same brick, as a mask per lane (registers and scalar lane reads only): the loop runs while the directory
        // read is in flight, and the bin reservation further down needs no loop of its own.
        unsigned long long same_brick = 0;
        if (BIN) {
            unsigned long long m = __ballot(act);
            while (m) {
                const int leader = __ffsll((long long)m) - 1;
                const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)bidx, leader);
                const bool same = act && bidx == lb;
                const unsigned long long sm = __ballot(same);
                if (same) same_brick = sm;
                m &= ~sm;
            }
        }
        const uint32_t b = BIN ? brick_acquire_groups(t, bidx, act, dir_word, same_brick, true) : brick_acquire_wave(t, bidx, act, dir_word);
        act = act && b != 0;
        const uint32_t lcell = local_index(ix, iy, iz);
        const uint32_t slot = b * kBrickCells + lcell;
        // What to do with the point is two bits of its cell: normal_found (stop buffering, grid.hpp:210) and has-dependants.
        // Both come from the brick's 128-byte flag line (L2-resident) instead of the cell's 8-byte info word.  The occupancy
        // word of the cell's plane is read in the same round trip (it is only needed while the cell has no normal).
        const uint64_t plane = (uint64_t)b * 8u + ((uint32_t)ix & 7u);
        const uint64_t bit = 1ull << ((((uint32_t)iy & 7u) << 3) | ((uint32_t)iz & 7u));
        bool has_n = false, has_d = false;
        unsigned long long occ_word = ~0ull;
        if (act) {
            const ulonglong2 nd = *reinterpret_cast<const ulonglong2*>(&t.nd_mask[plane * 2]);
            occ_word = *reinterpret_cast<const unsigned long long*>(&t.occ_mask[plane]);
            has_n = (nd.x & bit) != 0;
            has_d = (nd.y & bit) != 0;
        }

        // first occupancy (grid.hpp:219-243): the returning atomic on the brick's occupancy word decides who is first
        bool first = false;
        if (act && !has_n && !probe && !(occ_word & bit)) {  // a stale (cached) word only sends the lane through the atomic
            unsigned long long* om = reinterpret_cast<unsigned long long*>(&t.occ_mask[plane]);
            const unsigned long long old = atomicOr(om, (unsigned long long)bit);
            first = !(old & bit);  // (occupancy lives in occ_mask alone; the cell's info word is not touched)
        }
        // A cell occupied since the last clean pass has no list and no flag yet, but it may carry the ONE dependant it was given
        // while unoccupied (pre_dep, see k_materialize_new): every reader of the lists falls back on it.  The binned form parks
        // every point of a cell without a normal anyway and its per-brick kernels do the looking; the direct form looks here.
        uint32_t pre_nid = 0;
        if (!BIN && pre_possible && act && !has_n && !has_d && !probe) {
            pre_nid = t.pre_dep[slot];
            has_d = pre_nid != 0;
        }
        // newly occupied cells are staged in LDS (per wave) and appended to occ_list in batches: C_OCC is one address for the
        // whole chip (a same-address atomic retires every ~12 ns), so it gets one atomic per flush, not one per tile.  The
        // count lives in a wave-uniform register, so no two waves ever have to agree on a flush (no barrier in the tile loop).
        {
            const unsigned long long fm = __ballot(first);
            if (fm) {
                if (occ_n + 64u > (uint32_t)kOccStage) {
                    flush_occ_stage(BIN ? kernarg_tables() : t, s_occ, occ_n);
                    occ_n = 0;
                }
                if (first) s_occ[occ_n + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = slot;
                occ_n += (uint32_t)__popcll(fm);
            }
        }

        // What is left to do with the point: buffer it while its voxel has no normal (grid.hpp:210-211,230,239) and update the
        // voxel's dependants (grid.hpp:244-277).  Binned form (BIN = true, default): both are handed to the per-brick kernels --
        // the point is parked in its brick's bin, k_update accumulates the brick's records in LDS and flushes each once per
        // launch, k_buffer appends the brick's buffered points to the log as one run and links them into their cells' chains
        // with LDS exchanges.  One reservation per distinct brick per wave (ballot grouping); a lane whose brick region is
        // full or unplanned stays `todo` and takes the direct forms below, so correctness never depends on the plan.
        bool todo = act && (has_d || !has_n);
        if (BIN) {
            const bool want_bin = todo;
            // phase 1 (registers only): the lane's group = the lanes of its bin region (brick x {cell has a normal, cell has
            // none}) -> leader lane, rank in group, group size
            const uint32_t rg = 2u * b + (has_n ? 0u : 1u);
            const unsigned long long hn = __ballot(has_n);
            const unsigned long long grp = same_brick & __ballot(want_bin) & (has_n ? hn : ~hn);
            const uint32_t grp_leader = want_bin ? (uint32_t)(__ffsll((long long)grp) - 1) : lane;
            const uint32_t grp_rank = (uint32_t)__popcll(grp & ((1ull << lane) - 1ull));
            const uint32_t grp_size = (uint32_t)__popcll(grp);
            // phase 2: every group leader reserves for its group in ONE wave-instruction (one memory round trip per tile);
            // the counter also records the demand the next launch's plan is made from
            uint32_t base = 0, cap = 0, roff = 0;
            if (want_bin && grp_leader == lane) {
                base = atomicAdd(&t.bin_fill[rg], grp_size);
                cap = t.bin_capb[rg];
                roff = t.bin_off[rg];
            }
            base = __shfl(base, (int)grp_leader);
            cap = __shfl(cap, (int)grp_leader);
            roff = __shfl(roff, (int)grp_leader);
            if (todo && !probe) {
                const uint32_t pos = base + grp_rank;
                if (pos < cap) {
                    const uint64_t e = (uint64_t)roff + pos;
                    // the point + its cell inside the brick + its frame id (viewpoint latch); read back by k_update / k_buffer
                    t.bin_pt[e] = make_float4(p.x, p.y, p.z, __uint_as_float(lcell | (fid << 9)));
                    if (COLOR) t.bin_rgb[e] = rgb;
                    todo = false;
                }
            }
        }
        if (probe || __ballot(todo) == 0) continue;  // wave-uniform; the common case of the binned form

        if constexpr (BIN) {
            // The rare lane whose bin region was full or unplanned: handed, with what the direct forms need, to k_integrate_overflow,
            // which runs behind this kernel.  Keeping the direct forms out of the tile loop halves the scalar state this kernel
            // spills into VGPR lanes (60 -> 37 SGPRs) and 13 VGPRs, and no wave waits in the loop for a lane that walks a dependant list.
            const Tables& tr = kernarg_tables();
            const unsigned long long oi = wave_reserve(&tr.ctr[C_OVF], todo);
            if (todo) {
                if (oi < tr.ovf_cap) {
                    tr.ovf_pt[oi] = make_float4(p.x, p.y, p.z, __uint_as_float(slot));
                    tr.ovf_aux[oi] = make_uint2(fid | (has_n ? kOvfHasNormal : 0u) | (has_d ? kOvfHasDeps : 0u), rgb);
                } else {
                    atomicOr(&tr.ctr[C_ERR], (unsigned long long)E_OVF);  // (the list holds a whole launch: cannot happen)
                }
            }
        } else {
            direct_forms<COLOR, false>(g, t, q, log_ctr, log_base, p, slot, b, fid, rgb, todo, has_n, has_d, pre_nid, c_buf, c_tested, c_member);
        }
    }
    if (occ_n) flush_occ_stage(t, s_occ, occ_n);  // wave-uniform
    // block-level reduction of the diagnostics: one device atomic per counter per block
    uint32_t cv[6] = {c_present, c_z, c_in, c_buf, c_tested, c_member};
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint32_t v = cv[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        cv[k] = v;
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++)
            if (cv[k]) atomicAdd(&blk_ctr[k], cv[k]);
    }
    __syncthreads();
    if (!probe && threadIdx.x < 6 && blk_ctr[threadIdx.x]) atomicAdd(&t.ctr[C_PRESENTED + threadIdx.x], (unsigned long long)blk_ctr[threadIdx.x]);
}


// K1b (binned form): the points k_integrate could not park (bin region full, brick unplanned or found within the launch after the
// spare regions ran out: 0.3 % of the bench's points) take the direct forms here, right behind it on the stream.  The list is sized
// for a whole launch, so a session without a bin plan (its first small batch) comes through here entirely.  A workgroup takes 256
// points at a time: one lane per point for the buffering, then one lane per (point, dependant) PAIR -- the pairs of the 256 points
// are numbered through a prefix sum of the list lengths, so a point on a cell with twenty dependants does not hold its wave for
// twenty dependent reads.  The list is reset by the next launch's bin plan (k_bin_place; the host where there is no plan).
template <bool COLOR>
__global__ __launch_bounds__(256) void k_integrate_overflow(const GridParams g, const Tables t, const uint32_t log_rot)
{
    __shared__ unsigned long long queue[4][kQueueRows * kQueueStride];
    __shared__ unsigned int blk_ctr[3];
    __shared__ float s_x[256], s_y[256], s_z[256];
    __shared__ uint32_t s_rgb[COLOR ? 256 : 1];
    __shared__ uint32_t s_off[256];    // first dependant entry of the point's cell (dep[] stays below 2^32 entries, host-checked), or ...
    __shared__ uint8_t s_pre[256];     // ... 1: the record id of the cell's pre-dependant (cell occupied in this epoch: pre_list_entry)
    __shared__ uint32_t s_pref[257];   // pairs in front of the point
    __shared__ uint32_t s_wsum[4];
    const uint64_t n = min((uint64_t)t.ctr[C_OVF], t.ovf_cap);  // the same in every thread of the grid (nothing appends while this kernel runs)
    if (n == 0) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned long long* q = queue[wave];
    if (tid < 3) blk_ctr[tid] = 0;
    __syncthreads();
    const uint32_t region = (blockIdx.x + log_rot) & (kLogRegions - 1);
    unsigned long long* log_ctr = &t.log_ctr[region * 16];
    const uint64_t log_base = (uint64_t)region * t.log_region_cap;
    uint32_t c_buf = 0, c_tested = 0, c_member = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * 256u; base < n; base += (uint64_t)gridDim.x * 256u) {  // block-uniform trip count
        const uint64_t i = base + tid;
        const bool todo = i < n;
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        uint2 aux = make_uint2(0u, 0u);
        if (todo) {
            rec = t.ovf_pt[i];
            aux = t.ovf_aux[i];
        }
        const uint32_t slot = __float_as_uint(rec.w);
        uint32_t cnt = 0, off = 0;
        bool pre = false;
        if (todo && ((aux.x & kOvfHasDeps) || !(aux.x & kOvfHasNormal))) {  // issued before the buffering's atomics
            const uint64_t info = t.info[slot];
            cnt = (uint32_t)((info >> kDepCntShift) & kDepCntMask);
            off = (uint32_t)(info >> kDepOffShift);
            if (cnt == 0 && !(aux.x & kOvfHasNormal)) {  // no list yet: a cell occupied in this epoch may carry a pre-dependant
                off = t.pre_dep[slot];
                pre = off != 0;
                cnt = pre ? 1u : 0u;
            }
        }
        direct_buffer<COLOR, true>(t, log_ctr, log_base, F3{rec.x, rec.y, rec.z}, slot, slot >> 9, aux.x & ~(kOvfHasNormal | kOvfHasDeps), aux.y, todo,
                                   (aux.x & kOvfHasNormal) != 0, c_buf);
        const uint32_t inc = wave_inclusive_scan(cnt);
        if (lane == 63) s_wsum[wave] = inc;
        s_x[tid] = rec.x, s_y[tid] = rec.y, s_z[tid] = rec.z;
        if (COLOR) s_rgb[tid] = aux.y;
        s_off[tid] = off;
        s_pre[tid] = pre ? 1 : 0;
        __syncthreads();
        uint32_t before = inc - cnt;
        for (uint32_t w2 = 0; w2 < wave; w2++) before += s_wsum[w2];
        s_pref[tid] = before;
        if (tid == 255) s_pref[256] = before + cnt;
        __syncthreads();
        const uint32_t total = s_pref[256];
        for (uint32_t k0 = 0; k0 < total; k0 += 256u) {  // block-uniform trip count
            const uint32_t k = k0 + tid;
            bool member = false;
            StatDeltaT<COLOR> d;
            stat_delta_zero(d);
            uint32_t sid = 0;
            if (k < total) {
                uint32_t lo = 0, hi = 256;  // s_pref[lo] <= k < s_pref[hi]
#pragma unroll
                for (int st = 0; st < 8; st++) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_pref[mid] <= k) lo = mid;
                    else hi = mid;
                }
                const DepEntry e = s_pre[lo] ? make_dep_entry(t, s_off[lo]) : t.dep[(uint64_t)s_off[lo] + (k - s_pref[lo])];
                float sp, distf;
                c_tested++;
                if (line_member(g, F3{s_x[lo], s_y[lo], s_z[lo]}, F3{e.ax, e.ay, e.az}, F3{e.abx, e.aby, e.abz}, e.dd, sp, distf)) {
                    member = true;
                    c_member++;
                    sid = e.sid;
                    stat_delta_add(d, pair_delta(g, sp, distf), COLOR ? s_rgb[lo] : 0u);
                }
            }
            wave_flush_members(t, q, member, d, sid);
        }
        __syncthreads();  // the staged points are rewritten by the next round
    }
    uint32_t cv[3] = {c_buf, c_tested, c_member};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t v = cv[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (lane == 0 && v) atomicAdd(&blk_ctr[k], v);
    }
    __syncthreads();
    if (tid < 3 && blk_ctr[tid]) atomicAdd(&t.ctr[C_BUFFERED + tid], (unsigned long long)blk_ctr[tid]);
#ifdef HFPF_OVF_COUNT_DEBUG
    if (tid == 0 && blockIdx.x == 0) atomicAdd(&t.ctr[C_BUFFERED], (unsigned long long)n);
#endif
}

// ------------------------------------------------------------------------------------------------
// K2 (binned form): one workgroup per brick.  The brick's parked points are read back coalesced; a point finds its cell's
// dependant list through the brick's 512 info words (staged in LDS: k_integrate no longer reads them), and every
// (point, dependant) member pair adds its contribution (stats.hpp) to an LDS table keyed by record id (open addressing,
// LDS atomics).  The table is flushed with 8 lanes per record, so a record costs one 64-byte memory-side request per
// brick per launch however many of the brick's cells and points touched it.  A full table falls back to direct device
// atomics.  Tried instead and slower on the bench: points counting-sorted by cell in LDS with
// rows of lanes sharing a list (1.3 ms against 1.0: the sort, the row reductions and the barriers cost more than the
// broadcast reads save) and lists staged in LDS with a table indexed by list position (1.35 ms: 3.7x the flushes).
#ifndef HFPF_UPD_BITS
#define HFPF_UPD_BITS 9  // log2 of the LDS table size of k_update
#endif
constexpr int kUpdSlots = 1 << HFPF_UPD_BITS;
constexpr unsigned long long kPreListBit = 1ull << 63;  // in a staged info word: the list is the cell's pre-dependant alone (pre_list_entry)
constexpr int kUpdThreads = 256;
#ifndef HFPF_UPD_LANES
#define HFPF_UPD_LANES 2  // lanes that share one point in k_update (1, 2, 4 or 8)
#endif
constexpr uint32_t kUpdLanes = HFPF_UPD_LANES;
__device__ __forceinline__ uint32_t upd_hash(uint32_t sid) { return (sid * 2654435761u) >> (32 - HFPF_UPD_BITS); }

template <bool COLOR>
__global__ __launch_bounds__(256) void k_update(const GridParams g, const Tables t, const uint32_t n_bricks)
{
    constexpr int W = COLOR ? 8 : kStatUsed;
    __shared__ uint64_t s_info[kBrickCells];
    __shared__ uint32_t keys[kUpdSlots];
    __shared__ unsigned long long vals[kUpdSlots * W];
    __shared__ unsigned int blk_ctr[2];
    const uint32_t b = blockIdx.x + 1;
    if (b > n_bricks) return;
    // the brick's two bin regions (cells with / without a normal) as one index space [0, fill)
    const uint32_t fill_a = min(t.bin_fill[2 * b], t.bin_capb[2 * b]), fill_b = min(t.bin_fill[2 * b + 1], t.bin_capb[2 * b + 1]);
    const uint32_t fill = fill_a + fill_b;
    if (fill == 0) return;  // block-uniform
    const uint32_t tid = threadIdx.x;
    const uint64_t first_a = t.bin_off[2 * b], first_b = t.bin_off[2 * b + 1];
    auto entry = [&](uint32_t i) -> uint64_t { return i < fill_a ? first_a + i : first_b + (i - fill_a); };
    float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid / kUpdLanes < fill) pe = t.bin_pt[entry(tid / kUpdLanes)];  // in flight while the tables are set up
    {
        ulonglong2 inf = *reinterpret_cast<const ulonglong2*>(&t.info[(uint64_t)b * kBrickCells + 2u * tid]);
        for (uint32_t i = tid; i < (uint32_t)kUpdSlots; i += 256) keys[i] = 0;
        for (uint32_t i = tid; i < (uint32_t)(kUpdSlots * W); i += 256) vals[i] = 0;
        if (tid < 2) blk_ctr[tid] = 0;
        // an empty list may belong to a cell occupied in this epoch that carries a pre-dependant: a list of one, marked by bit 63,
        // whose "offset" is the record id (pre_list_entry)
        auto with_pre = [&](unsigned long long info, uint32_t cell) -> unsigned long long {
            if ((info >> kDepCntShift) & kDepCntMask) return info;
            const uint32_t pd = t.pre_dep[(uint64_t)b * kBrickCells + cell];
            return pd ? ((info & 3ull) | (1ull << kDepCntShift) | ((unsigned long long)pd << kDepOffShift) | kPreListBit) : info;
        };
        inf.x = with_pre(inf.x, 2u * tid);
        inf.y = with_pre(inf.y, 2u * tid + 1u);
        s_info[2u * tid] = inf.x;
        s_info[2u * tid + 1] = inf.y;
    }
    __syncthreads();
    const float4* __restrict__ dep4 = reinterpret_cast<const float4*>(t.dep);
    uint32_t c_tested = 0, c_member = 0;
    // kUpdLanes lanes share a point and take every kUpdLanes-th entry of its cell's list: the lanes of a point read
    // consecutive 32-byte entries (the vector L1's access rate bounds this kernel), and a list of up to kUpdLanes entries
    // is one step for all of them.  Measured per 150-frame launch: 1 lane 0.84 ms, 2 lanes 0.80, 4 lanes 0.89, 8 lanes 1.4.
    const uint32_t sub = tid % kUpdLanes;
    for (uint32_t i = tid / kUpdLanes; i < fill; i += 256 / kUpdLanes) {
        const float4 cur = pe;
        const uint32_t rgb = COLOR ? t.bin_rgb[entry(i)] : 0u;
        if (i + 256 / kUpdLanes < fill) pe = t.bin_pt[entry(i + 256 / kUpdLanes)];  // next point: in flight during this one's pairs
        const F3 p = F3{cur.x, cur.y, cur.z};
        const uint64_t info = s_info[__float_as_uint(cur.w) & (kBrickCells - 1)];
        const uint32_t cnt = (uint32_t)((info >> kDepCntShift) & kDepCntMask);
        const uint64_t off = (info & ~kPreListBit) >> kDepOffShift;
        if (sub == 0) c_tested += cnt;
        float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0;
        if (sub < cnt) {
            if (info & kPreListBit) {
                pre_list_entry(t, (uint32_t)off, n0, n1);
            } else {
                n0 = dep4[2 * (off + sub)];
                n1 = dep4[2 * (off + sub) + 1];
            }
        }
        for (uint32_t j = sub; j < cnt; j += kUpdLanes) {
            const float4 e0 = n0, e1 = n1;
            if (j + kUpdLanes < cnt) {  // next entry: in flight during this pair (without it the loop is latency-bound: 1.16 ms against 0.84)
                n0 = dep4[2 * (off + j + kUpdLanes)];
                n1 = dep4[2 * (off + j + kUpdLanes) + 1];
            }
            float sp, distf;
            if (!line_member(g, p, F3{e0.y, e0.z, e0.w}, F3{e1.x, e1.y, e1.z}, e1.w, sp, distf)) continue;
            c_member++;
            const PairDelta q = pair_delta(g, sp, distf);
            const uint32_t sid = __float_as_uint(e0.x);
            uint32_t h = upd_hash(sid);
            bool placed = false;
            for (int probe = 0; probe < ((t.test_table_skip & sid) ? 0 : 16); probe++) {
                const uint32_t old = atomicCAS(&keys[h], 0u, sid);
                if (old == 0u || old == sid) {
                    placed = true;
                    break;
                }
                h = (h + 1) & (kUpdSlots - 1);
            }
            if (placed) {
                unsigned long long* sv = &vals[h * W];
                atomicAdd(&sv[SW_COUNT], 1ull);
                atomicAdd(&sv[SW_S], (unsigned long long)(long long)q.s);
                atomicAdd(&sv[SW_SS], (unsigned long long)(long long)q.ss);
                atomicAdd(&sv[SW_D], (unsigned long long)(long long)q.d);
                atomicAdd(&sv[SW_DD], (unsigned long long)(long long)q.dd);
                if constexpr (COLOR) {
                    atomicAdd(&sv[SW_R], (unsigned long long)((rgb >> 16) & 255u));
                    atomicAdd(&sv[SW_G], (unsigned long long)((rgb >> 8) & 255u));
                    atomicAdd(&sv[SW_B], (unsigned long long)(rgb & 255u));
                }
            } else {  // table full: straight to HBM
                unsigned long long* v = &t.stats[(uint64_t)sid * kStatWords];
                atomicAdd(&v[SW_COUNT], 1ull);
                atomicAdd(&v[SW_S], (unsigned long long)(long long)q.s);
                atomicAdd(&v[SW_SS], (unsigned long long)(long long)q.ss);
                atomicAdd(&v[SW_D], (unsigned long long)(long long)q.d);
                atomicAdd(&v[SW_DD], (unsigned long long)(long long)q.dd);
                if constexpr (COLOR) {
                    atomicAdd(&v[SW_R], (unsigned long long)((rgb >> 16) & 255u));
                    atomicAdd(&v[SW_G], (unsigned long long)((rgb >> 8) & 255u));
                    atomicAdd(&v[SW_B], (unsigned long long)(rgb & 255u));
                }
            }
        }
    }
    __syncthreads();
    {  // flush: 8 lanes per record, 32 records per pass
        const uint32_t w = tid & 7u;
        for (uint32_t sl = tid >> 3; sl < (uint32_t)kUpdSlots; sl += 32) {
            const uint32_t key = keys[sl];
            if (key != 0u && w < (uint32_t)W) atomicAdd(&t.stats[(uint64_t)key * kStatWords + w], vals[sl * W + w]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c_tested += __shfl_down(c_tested, o);
        c_member += __shfl_down(c_member, o);
    }
    if ((tid & 63u) == 0) {
        if (c_tested) atomicAdd(&blk_ctr[0], c_tested);
        if (c_member) atomicAdd(&blk_ctr[1], c_member);
    }
    __syncthreads();
    // striped like k_replay's counter: words 2 and 3 of the 64 log_ctr lines, summed by the host
    if (tid < 2 && blk_ctr[tid])
        atomicAdd(&t.log_ctr[(blockIdx.x & (kLogRegions - 1)) * 16 + 2 + tid], (unsigned long long)blk_ctr[tid]);
}

// ------------------------------------------------------------------------------------------------
// K2, cell-sorted form.  The per-point form above reads a 32-byte dependant entry from global memory and does five LDS atomics
// for EVERY (point, dependant) pair.  Here the brick's parked points are counting-sorted by cell into LDS (kUpd2Cap at a time),
// and a work item is (cell, dependant entry, run of up to kUpd2Chunk of the cell's points): the entry is read once per item, the
// points come from LDS (lanes of one cell read the same address), the contributions are summed in registers (int32: one is below
// 2^27, stats.hpp) and the item ends with ONE hash insert + five LDS atomics.  Same LDS record table and flush as above; the
// sums are integers, so the result is bit-identical whatever the order.
//
// Thread t owns cell t of the brick for the whole launch (its dependant count and list offset stay in registers).  A round:
//   1. histogram of the round's points by cell with returning LDS atomics (= rank of the point inside its cell)      | barrier
//   2. owner threads: exclusive scans over the cells (first sorted position, first work item) -- wave scan             | barrier
//   3. owner threads: per-cell record (list offset, first position, points) and one 32-bit DESCRIPTOR per work item
//      (cell, chunk, entry) written into LDS in item order                                                             | barrier
//   4. points scattered into LDS in cell order; the first item's dependant entry is already in flight                  | barrier
//   5. items: descriptor -> cell record -> entry (global, read one item ahead) -> pair loop over the run -> table insert
// An item finds its work with two dependent LDS reads (round 2: a 9-step binary search over the item prefix sums, i.e. nine
// dependent LDS reads per item, ~1000 cycles of latency in front of every entry read).  Items beyond the descriptor array
// (kUpd2Desc per round) and entries past 32766 keep the search.
#ifndef HFPF_UPD2_CHUNK
#define HFPF_UPD2_CHUNK 8  // points of a cell one work item takes
#endif
constexpr int kUpd2Chunk = HFPF_UPD2_CHUNK;
static_assert(kUpd2Chunk >= 1 && kUpd2Chunk <= 15, "k_update_cells: int32 item sums hold 15 contributions");
constexpr uint32_t kUpd2NoDesc = ~0u;
// The kernel is a chain of latencies per brick (dependent loads, four barriers a round), so what decides its speed is how many
// bricks a CU works on at once and how many rounds a brick takes, i.e. how its LDS is spent -- the record table (44 bytes a slot),
// the sorted points (12 bytes each).  Two shapes are instantiated and the host picks per launch (hfpf.hip pick_update_shape):
//   kUpdDense  512 threads, 2048 points a round, 352-slot table: 52 KB, three workgroups (24 waves, 80 VGPRs) per CU.  Bricks that
//              update ~100 records each: the 640x480 @ 1 mm bench.  Measured on one box, per 150-frame launch: 1024-point rounds
//              at four workgroups 587 us, 1536 at three 602 us, 2048 at three 537 us (a 320-slot table with 1024 descriptors: 552).
//   kUpdWide   512 threads, 1536 points, 512 slots: 53 KB, three workgroups.  Taken for the rest of the session once the small
//              table has overflowed more than rarely (the kernel counts the items that found no slot, the host sees the count at
//              its counter read-backs): 0.5 mm voxels put 200-300 records on a brick, and an item without a slot costs five
//              scattered memory-side atomics (2048 x 1536 @ 0.5 mm: 1.34 ms per launch; 1.48 ms with 1024-point rounds, 1.71 ms
//              with 2048-point rounds at two workgroups, 2.35 ms with the dense table).
// Measured and dropped: 256 threads with 512-point rounds for bricks with few points (1.9 ms on that workload: the barriers cost
// less, the lanes per brick are missed more).  The table takes any size (the hash is range-reduced with a multiply, not masked).
struct UpdShape {
    int threads, cap, slots, desc, waves;
};
#ifndef HFPF_UPD_DENSE_SHAPE
#define HFPF_UPD_DENSE_SHAPE 512, 2048, 352, 1280, 6
#endif
#ifndef HFPF_UPD_WIDE_SHAPE
#define HFPF_UPD_WIDE_SHAPE 512, 1536, 512, 1024, 6
#endif
constexpr UpdShape kUpdDense{HFPF_UPD_DENSE_SHAPE}, kUpdWide{HFPF_UPD_WIDE_SHAPE};
// With colour the sorted points carry 4 more bytes and a table slot 24 more: 1024-point rounds keep three workgroups on a CU.
constexpr UpdShape kUpdDenseColor{512, 1024, 352, 1024, 6}, kUpdWideColor{512, 1024, 512, 1024, 4};

// REPLAY = true is the buffer replay of a clean pass (grid.hpp:418-440) for bricks whose buffered points form ONE contiguous run of
// the point log (Tables::run_*): the same kernel with the run as its input instead of the bin -- the cell of a logged point is
// recomputed from its coordinates, exactly as k_integrate indexed it -- and with every cell's dependant list cut down to the
// registrants of the running pass (the note k_depinc_fill left in the cell's scratch word, cleared here).  Coalesced reads of the
// run instead of one dependent 16-byte chain hop per point.
template <bool COLOR, int THREADS, int CAP, int SLOTS, int DESC, int WAVES, bool REPLAY = false>
__global__ __launch_bounds__(THREADS, WAVES) void k_update_cells(const GridParams g, const Tables t, const uint32_t n_bricks)
{
    static_assert(THREADS == kBrickCells, "k_update_cells: thread t owns cell t");
    static_assert(CAP % THREADS == 0 && CAP <= 4095 && (CAP + kUpd2Chunk - 1) / kUpd2Chunk <= 256, "k_update_cells: cell record holds 12-bit positions, descriptor 8-bit chunks");
    constexpr int kUpd2Threads = THREADS, kUpd2Cap = CAP, kUpd2Slots = SLOTS, kUpd2Desc = DESC;
    auto upd2_hash = [](uint32_t sid) -> uint32_t { return __umulhi(sid * 2654435761u, (uint32_t)SLOTS); };
    constexpr int W = COLOR ? 8 : kStatUsed;
    constexpr uint32_t T = kUpd2Threads, CPT = 1, PER = kUpd2Cap / T, CH = kUpd2Chunk;
    __shared__ float s_px[kUpd2Cap], s_py[kUpd2Cap], s_pz[kUpd2Cap];  // sorted points, one array per coordinate (12 bytes a point)
    __shared__ uint32_t s_rgb[COLOR ? kUpd2Cap : 1];
    __shared__ uint32_t s_cnt[kBrickCells];
    // per cell and round: first entry of the dependant list (40 bits: an index into ONE entry space that holds dep[] and, for a cell whose
    // "list" is its pre-dependant, the records' own lines -- Tables::ent_base) | first sorted position (12) | points (12)
    __shared__ uint64_t s_pack[kBrickCells];
    __shared__ uint32_t s_items[kBrickCells + 1];  // first work item of each cell (only the search path and the total read it)
    __shared__ uint32_t s_desc[kUpd2Desc];         // per work item: cell (9) | chunk (8) | entry (15)
    __shared__ uint32_t s_wsum[2][T / 64];
    __shared__ uint32_t keys[kUpd2Slots];
    __shared__ unsigned long long vals[kUpd2Slots * W];
    __shared__ unsigned int blk_ctr[3];
    const uint32_t b = blockIdx.x + 1;
    if (b > n_bricks) return;
    if (REPLAY && t.run_cnt[b] != 1u) return;  // block-uniform: several runs or stray entries -- the chain walk replays this brick
    const uint32_t fill_a = REPLAY ? t.run_len[b] : min(t.bin_fill[2 * b], t.bin_capb[2 * b]);
    const uint32_t fill_b = REPLAY ? 0u : min(t.bin_fill[2 * b + 1], t.bin_capb[2 * b + 1]);
    const uint32_t fill = fill_a + fill_b;
    if (fill == 0) return;  // block-uniform
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t first_a = REPLAY ? t.run_start[b] : t.bin_off[2 * b], first_b = REPLAY ? 0u : t.bin_off[2 * b + 1];
    if (REPLAY && first_a + fill_a > t.max_log + 1) {  // block-uniform; a run record that leaves the log (cannot happen: k_buffer records a run only when it fits)
        if (threadIdx.x == 0) atomicOr(&t.ctr[C_ERR], (unsigned long long)E_CHAIN);
        return;
    }
    auto entry = [&](uint32_t i) -> uint64_t { return i < fill_a ? first_a + i : first_b + (i - fill_a); };
    const bool scanner = tid < (uint32_t)kBrickCells / CPT;  // wave-uniform: the threads that own cells
    uint32_t own_cnt[CPT], own_off[CPT], own_pre = 0;
    bool any_note = false;
#pragma unroll
    for (uint32_t k = 0; k < CPT; k++) {
        const uint64_t cslot = (uint64_t)b * kBrickCells + tid * CPT + k;
        const uint64_t info = scanner ? t.info[cslot] : 0ull;
        const uint32_t pd = (!REPLAY && scanner) ? t.pre_dep[cslot] : 0u;  // (read beside the info word, not behind it: one round trip)
        own_cnt[k] = (uint32_t)((info >> kDepCntShift) & kDepCntMask);
        own_off[k] = (uint32_t)(info >> kDepOffShift);  // dep[] stays below 2^32 entries (host-checked)
        // no list: a cell occupied in this epoch may carry a pre-dependant (a list of one: pre_list_entry)
        if (!REPLAY && own_cnt[k] == 0 && pd) own_cnt[k] = 1, own_off[k] = pd, own_pre |= 1u << k;
        if (REPLAY) {  // only the registrants of the running pass: the entries behind the old list length
            const uint32_t note = scanner ? t.dep_tmp[cslot] : 0u;
            const bool noted = (note & kTouchedMark) != 0;
            const uint32_t old_len = noted ? min(note & kDepOldMax, own_cnt[k]) : own_cnt[k];
            own_off[k] += old_len;
            own_cnt[k] -= old_len;
            if (noted) t.dep_tmp[cslot] = 0;
            any_note = any_note || noted;
        }
    }
    if (REPLAY && !__syncthreads_or(any_note ? 1 : 0)) return;  // block-uniform: no cell of this brick gained a registrant
    for (uint32_t i = tid; i < (uint32_t)kBrickCells; i += T) s_cnt[i] = 0;
    for (uint32_t i = tid; i < (uint32_t)kUpd2Slots; i += T) keys[i] = 0;
    for (uint32_t i = tid; i < (uint32_t)(kUpd2Slots * W); i += T) vals[i] = 0;
    if (tid < 3) blk_ctr[tid] = 0;
    const float4* __restrict__ ent4 = reinterpret_cast<const float4*>(t.ent_base);  // dep[] and nv_line as one array of 32-byte entries
    uint32_t c_tested = 0, c_member = 0, c_miss = 0;
    // Software pipeline: the points of round r+1 are read from the bin while round r's items are worked (without it the streaming
    // replay, whose bricks take several rounds, is a third slower), and an item's dependant entry is read one item ahead (the first
    // one of a round ahead of the scatter).
    float4 pt[PER];
    uint32_t col[PER];
    auto load_round = [&](uint32_t r0) {
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = r0 + tid + k * T;
            col[k] = 0;
            pt[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < fill) {
                if (REPLAY) {
                    pt[k] = t.log_pt[entry(i)];
                    if (COLOR) col[k] = t.log_rgb[entry(i)];
                    int32_t ix, iy, iz;  // the cell the point was buffered under (grid.hpp:630-637 on the same f32 point)
                    voxel_coords(g, F3{pt[k].x, pt[k].y, pt[k].z}, ix, iy, iz);
                    pt[k].w = __uint_as_float(local_index(ix, iy, iz));
                } else {
                    pt[k] = t.bin_pt[entry(i)];
                    if (COLOR) col[k] = t.bin_rgb[entry(i)];
                }
            }
        }
    };
    struct Item {
        uint32_t first, p_lo, p_hi;
        float4 e0, e1;
    };
    auto locate = [&](uint32_t item, Item& it) {  // work item -> (cell, entry, run of points) + the entry's two 16-byte halves
        uint32_t d = kUpd2NoDesc;
        if (item < (uint32_t)kUpd2Desc) d = s_desc[item];
        uint32_t c, j, ch;
        if (d != kUpd2NoDesc) {
            c = d & (kBrickCells - 1);
            ch = (d >> 9) & 255u;
            j = d >> 17;
        } else {  // no descriptor: search the item prefix sums (s_items[lo] <= item < s_items[hi])
            uint32_t lo = 0, hi = kBrickCells;
#pragma unroll 1
            for (int st = 0; st < 9; st++) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_items[mid] <= item) lo = mid;
                else hi = mid;
            }
            c = lo;
            const uint32_t local = item - s_items[lo];
            const uint32_t chunks = ((uint32_t)((s_pack[lo] >> 52) & 0xFFFu) + CH - 1) / CH;
            j = local / chunks;
            ch = local - j * chunks;
        }
        const uint64_t pk = s_pack[c];
        const uint32_t n_c = (uint32_t)((pk >> 52) & 0xFFFu);
        it.first = (uint32_t)((pk >> 40) & 0xFFFu);
        const uint64_t e = (pk & ((1ull << 40) - 1ull)) + j;
        it.e0 = ent4[2 * e];
        it.e1 = ent4[2 * e + 1];
        it.p_lo = ch * CH;
        it.p_hi = min(n_c, it.p_lo + CH);
    };
    load_round(0);  // in flight while the tables above are cleared
    __syncthreads();
    for (uint32_t r0 = 0; r0 < fill; r0 += (uint32_t)kUpd2Cap) {  // block-uniform trip count
        const uint32_t n_round = min((uint32_t)kUpd2Cap, fill - r0);

        // 1. rank within the cell from the histogram's returning atomic
        uint32_t rk[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            rk[k] = 0;
            if (tid + k * T < n_round) rk[k] = atomicAdd(&s_cnt[__float_as_uint(pt[k].w) & (kBrickCells - 1)], 1u);
        }
        __syncthreads();
        // 2. exclusive scans over the cells: sorted position of the cell's first point, index of its first work item
        uint32_t n[CPT], it[CPT], sum_n = 0, sum_it = 0, inc_n = 0, inc_it = 0;
        {
#pragma unroll
            for (uint32_t k = 0; k < CPT; k++) {
                const uint32_t c = tid * CPT + k;
                n[k] = s_cnt[c];
                s_cnt[c] = 0;  // for the next round's histogram (three barriers away)
                it[k] = n[k] ? own_cnt[k] * ((n[k] + CH - 1) / CH) : 0u;
                sum_n += n[k];
                sum_it += it[k];
            }
            inc_n = wave_inclusive_scan(sum_n);
            inc_it = wave_inclusive_scan(sum_it);
            if (lane == 63) s_wsum[0][wave] = inc_n, s_wsum[1][wave] = inc_it;
        }
        __syncthreads();
        // 3. cell records and item descriptors
        {
            uint32_t pre_n = inc_n - sum_n, pre_it = inc_it - sum_it;
            for (uint32_t w2 = 0; w2 < wave; w2++) pre_n += s_wsum[0][w2], pre_it += s_wsum[1][w2];
#pragma unroll
            for (uint32_t k = 0; k < CPT; k++) {
                const uint32_t c = tid * CPT + k;
                s_pack[c] = ((((own_pre >> k) & 1u) ? t.ent_nv_first : t.ent_dep_first) + own_off[k]) | ((uint64_t)pre_n << 40) | ((uint64_t)n[k] << 52);
                s_items[c] = pre_it;
                if (it[k]) {
                    const uint32_t chunks = (n[k] + CH - 1) / CH;
                    uint32_t idx = pre_it;
                    for (uint32_t j = 0; j < own_cnt[k] && idx < (uint32_t)kUpd2Desc; j++)
                        for (uint32_t ch = 0; ch < chunks && idx < (uint32_t)kUpd2Desc; ch++, idx++)
                            s_desc[idx] = j < 32767u ? (c | (ch << 9) | (j << 17)) : kUpd2NoDesc;
                }
                pre_n += n[k];
                pre_it += it[k];
            }
            if (tid == (uint32_t)kBrickCells / CPT - 1) s_items[kBrickCells] = pre_it;
        }
        __syncthreads();
        const uint32_t total = s_items[kBrickCells];
        Item nxt;
        nxt.first = nxt.p_lo = nxt.p_hi = 0;
        nxt.e0 = nxt.e1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < total) locate(tid, nxt);  // entry read in flight during the scatter
        // 4. scatter into cell order
#pragma unroll
        for (uint32_t k = 0; k < PER; k++)
            if (tid + k * T < n_round) {
                const uint32_t pos = (uint32_t)((s_pack[__float_as_uint(pt[k].w) & (kBrickCells - 1)] >> 40) & 0xFFFu) + rk[k];
                s_px[pos] = pt[k].x;
                s_py[pos] = pt[k].y;
                s_pz[pos] = pt[k].z;
                if (COLOR) s_rgb[pos] = col[k];
            }
        __syncthreads();
        if (r0 + (uint32_t)kUpd2Cap < fill) load_round(r0 + (uint32_t)kUpd2Cap);  // in flight during the items
        // 5. items.  No barrier behind them: the next round's histogram only touches s_cnt, and its first barrier keeps every
        // writer of s_pack / s_desc / the sorted points behind the last reader of this round.
        for (uint32_t item = tid; item < total; item += T) {
            const Item cur = nxt;
            if (item + T < total) locate(item + T, nxt);
            const F3 la = F3{cur.e0.y, cur.e0.z, cur.e0.w}, lab = F3{cur.e1.x, cur.e1.y, cur.e1.z};
            const LineDiv dv = line_div_of(cur.e1.w);
            int32_t a_n = 0, a_s = 0, a_ss = 0, a_d = 0, a_dd = 0, a_r = 0, a_g = 0, a_b = 0;
            F3 qn = F3{0.f, 0.f, 0.f};  // the next point of the run is read while this one is tested
            if (cur.p_lo < cur.p_hi) qn = F3{s_px[cur.first + cur.p_lo], s_py[cur.first + cur.p_lo], s_pz[cur.first + cur.p_lo]};
            for (uint32_t pi = cur.p_lo; pi < cur.p_hi; pi++) {
                const F3 q3 = qn;
                if (pi + 1 < cur.p_hi) qn = F3{s_px[cur.first + pi + 1], s_py[cur.first + pi + 1], s_pz[cur.first + pi + 1]};
                float sp, distf;
                if (!line_member_hoisted(g, q3, la, lab, dv, sp, distf)) continue;
                const PairDelta q = pair_delta(g, sp, distf);
                a_n++;
                a_s += q.s;
                a_ss += q.ss;
                a_d += q.d;
                a_dd += q.dd;
                if constexpr (COLOR) {
                    const uint32_t rgb = s_rgb[cur.first + pi];
                    a_r += (int32_t)((rgb >> 16) & 255u);
                    a_g += (int32_t)((rgb >> 8) & 255u);
                    a_b += (int32_t)(rgb & 255u);
                }
            }
            c_tested += cur.p_hi - cur.p_lo;
            c_member += (uint32_t)a_n;
            if (a_n == 0) continue;
            const uint32_t sid = __float_as_uint(cur.e0.x);
            uint32_t h = upd2_hash(sid);
            bool placed = false;
            for (int probe = 0; probe < ((t.test_table_skip & sid) ? 0 : 16); probe++) {
                const uint32_t old = atomicCAS(&keys[h], 0u, sid);
                if (old == 0u || old == sid) {
                    placed = true;
                    break;
                }
                h = h + 1 == (uint32_t)kUpd2Slots ? 0u : h + 1;
            }
            c_miss += placed ? 0u : 1u;  // the host widens the table when this stops being rare
            unsigned long long* sv = placed ? &vals[h * W] : nullptr;
            unsigned long long* gv = &t.stats[(uint64_t)sid * kStatWords];  // table full: straight to HBM
#define HFPF_UPD2_ADD(word, val)                                                \
    if (placed) atomicAdd(&sv[word], (unsigned long long)(long long)(val));     \
    else atomicAdd(&gv[word], (unsigned long long)(long long)(val))
            HFPF_UPD2_ADD(SW_COUNT, a_n);
            HFPF_UPD2_ADD(SW_S, a_s);
            HFPF_UPD2_ADD(SW_SS, a_ss);
            HFPF_UPD2_ADD(SW_D, a_d);
            HFPF_UPD2_ADD(SW_DD, a_dd);
            if constexpr (COLOR) {
                HFPF_UPD2_ADD(SW_R, a_r);
                HFPF_UPD2_ADD(SW_G, a_g);
                HFPF_UPD2_ADD(SW_B, a_b);
            }
#undef HFPF_UPD2_ADD
        }
    }
    __syncthreads();
    {  // flush: 8 lanes per record
        const uint32_t w = tid & 7u;
        for (uint32_t sl = tid >> 3; sl < (uint32_t)kUpd2Slots; sl += T / 8) {
            const uint32_t key = keys[sl];
            if (key != 0u && w < (uint32_t)W) atomicAdd(&t.stats[(uint64_t)key * kStatWords + w], vals[sl * W + w]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c_tested += __shfl_down(c_tested, o);
        c_member += __shfl_down(c_member, o);
        c_miss += __shfl_down(c_miss, o);
    }
    if (lane == 0) {
        if (c_tested) atomicAdd(&blk_ctr[0], c_tested);
        if (c_member) atomicAdd(&blk_ctr[1], c_member);
        if (c_miss) atomicAdd(&blk_ctr[2], c_miss);
    }
    __syncthreads();
    if (REPLAY) {  // members found at replay time are counted apart (word 1 of the striped lines, like k_replay)
        if (tid == 1 && blk_ctr[1]) atomicAdd(&t.log_ctr[(blockIdx.x & (kLogRegions - 1)) * 16 + 1], (unsigned long long)blk_ctr[1]);
    } else if (tid < 2 && blk_ctr[tid]) {
        atomicAdd(&t.log_ctr[(blockIdx.x & (kLogRegions - 1)) * 16 + 2 + tid], (unsigned long long)blk_ctr[tid]);
    }
    if (tid == 2 && blk_ctr[2]) atomicAdd(&t.ctr[C_TABLE_MISS], (unsigned long long)blk_ctr[2]);  // rare by construction
    if (tid == 3 && fill > (uint32_t)kUpd2Cap) atomicAdd(&t.ctr[C_UPD_ROUNDS], (unsigned long long)((fill - 1u) / (uint32_t)kUpd2Cap));  // (few bricks)
}

// ------------------------------------------------------------------------------------------------
// K2b: the brick's parked points whose cell has no normal yet (its second bin region) are appended to the point log as ONE
// run (one reservation per brick and launch) and chained into their cells' four chains.  A long run (the first epoch:
// everything is buffered) chains with LDS exchanges, the chain heads travelling as one coalesced 8 KB read and write per
// brick; a short run (steady state: a few edge cells) exchanges straight on the heads in global memory.  Replaces one
// returning device atomic per buffered point in a separate pass over the log (k_link_log's atomicExch, ~1.2 ms for the
// first epoch's 39 M entries) and keeps a cell's entries of one launch within a few KB of each other for the replay's
// chain walks.
constexpr uint32_t kChains = kLogChains;  // interleaved chains per cell: k_replay walks them with 4 lanes in parallel
constexpr uint32_t kBufLdsMin = 256;  // runs at least this long chain in LDS
template <bool COLOR>
__global__ __launch_bounds__(256) void k_buffer(const GridParams g, const Tables t, const uint32_t n_bricks)
{
    __shared__ uint32_t s_head[kBrickCells * kChains];  // chain heads of the brick's cells
    __shared__ uint32_t s_minfid[kBrickCells];          // smallest frame id that buffered into the cell in this launch
    __shared__ unsigned long long s_base;
    const uint32_t b = blockIdx.x + 1;
    if (b > n_bricks) return;
    const uint32_t n_buf = min(t.bin_fill[2 * b + 1], t.bin_capb[2 * b + 1]);
    if (n_buf == 0) return;  // block-uniform
    const uint32_t tid = threadIdx.x;
    const uint64_t first = t.bin_off[2 * b + 1];
    const uint32_t region = b & (kLogRegions - 1);
    const bool in_lds = n_buf >= kBufLdsMin;  // block-uniform
    if (tid == 0) {
        s_base = atomicAdd(&t.log_ctr[region * 16], (unsigned long long)n_buf);
        const bool fits = s_base + n_buf <= t.log_region_cap;
        t.run_start[b] = (uint32_t)((uint64_t)region * t.log_region_cap + s_base + 1);
        t.run_len[b] = n_buf;
        t.run_cnt[b] += fits ? 1u : 0x100u;  // (one workgroup per brick and launch: no other writer)
    }
    if (in_lds) {
        for (uint32_t i = tid; i < (uint32_t)(kBrickCells * kChains); i += 256) s_head[i] = t.buf_head[(uint64_t)b * kBrickCells * kChains + i];
        for (uint32_t i = tid; i < (uint32_t)kBrickCells; i += 256) s_minfid[i] = kNoFrame;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    const uint64_t log_base = (uint64_t)region * t.log_region_cap;
    bool overflow = false;
    for (uint32_t i = tid; i < n_buf; i += 256) {
        const unsigned long long k = base + i;
        if (k >= t.log_region_cap) {
            overflow = true;
            continue;
        }
        const float4 pe = t.bin_pt[first + i];
        const uint32_t e = (uint32_t)(log_base + k + 1);
        const uint32_t w = __float_as_uint(pe.w), lcell = w & (kBrickCells - 1), fid = w >> 9;
        uint32_t prev;  // entry e joins chain e mod 4 of its cell
        if (in_lds) {
            prev = atomicExch(&s_head[lcell * kChains + (e & (kChains - 1))], e);
            atomicMin(&s_minfid[lcell], fid);
        } else {
            const uint64_t slot = (uint64_t)b * kBrickCells + lcell;
            prev = atomicExch(&t.buf_head[slot * kChains + (e & (kChains - 1))], e);
            if (fid < t.first_frame[slot]) atomicMin(&t.first_frame[slot], fid);  // viewpoint latch, grid.hpp:229,238
        }
        t.log_pt[e] = make_float4(pe.x, pe.y, pe.z, __uint_as_float(prev));
        if (COLOR) t.log_rgb[e] = t.bin_rgb[first + i];
    }
    if (overflow) atomicOr(&t.ctr[C_ERR], (unsigned long long)E_LOG);
    if (!in_lds) return;
    __syncthreads();
    for (uint32_t i = tid; i < (uint32_t)(kBrickCells * kChains); i += 256) t.buf_head[(uint64_t)b * kBrickCells * kChains + i] = s_head[i];
    for (uint32_t i = tid; i < (uint32_t)kBrickCells; i += 256) {  // viewpoint latch: smallest frame id that touched the cell
        const uint32_t mf = s_minfid[i];
        if (mf != kNoFrame) {
            uint32_t* ff = &t.first_frame[(uint64_t)b * kBrickCells + i];
            if (mf < *ff) atomicMin(ff, mf);
        }
    }
}

// Plan the bin regions of the next launch from the demand of the previous one: cap = demand * scale * slack + 64 (slack 2: a brick's
// share of a launch moves with the poses of its frames -- 15-frame launches of the 0.5 mm workload left 1 % of their points without
// room at 1.25, 0.5 % at 2), where the
// demand is the BRICK's (both regions): a clean pass between two launches moves cells from "no normal" to "normal", so either
// region must be able to take all of the brick's points.  n_regions = 2 * (bricks + 1); regions 0 and 1 belong to the null
// brick and stay empty.
// Regions [n_regions, n_planned) belong to brick ids the host has not seen yet: ids are handed out in order, so the next bricks a
// launch discovers find a region of `spare_cap` entries waiting (without one all their points go through the overflow list and
// take the direct forms).
// Two launches: k_bin_plan writes the capacities and one sum per workgroup of kBinPlanTile regions; k_bin_place turns them into
// region offsets (the sums in front of its tile + a scan inside the tile), switches off what does not fit the pool and restarts
// the demand counters.  (Until round 4: plan, a two-launch library scan, clamp -- four dependent launches of ~5 us in front of
// every integrate call.)
constexpr uint32_t kBinPlanTile = 1024;  // regions per workgroup: 256 threads x 4 consecutive regions
__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* s_w)  // sum over a 256-thread workgroup (convergent; s_w: 4 words)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t r = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(256) void k_bin_plan(const Tables t, const uint32_t n_regions, const uint32_t n_planned, const uint32_t spare_cap, const float scale,
                                                  const float slack, uint32_t* __restrict__ tile_sums)
{
    __shared__ uint32_t s_w[4];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t r = blockIdx.x * kBinPlanTile + threadIdx.x * 4u + k;
        if (r >= n_planned) continue;
        uint32_t cap = 0;
        if (r >= n_regions) {
            cap = spare_cap;
        } else if (r >= 2) {
            const uint32_t demand = t.bin_fill[r & ~1u] + t.bin_fill[r | 1u];
            if (demand) cap = (uint32_t)fminf((float)demand * scale * slack, 2.0e9f) + 64u;  // (saturated: a float above 2^32 does not convert)
        }
        t.bin_capb[r] = cap;
        sum += cap;
    }
    sum = block_sum_256(sum, s_w);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = sum;
}

// Offsets, clamp, restart.  Regions that do not fit the pool are switched off, and the demand counters of ALL regions restart --
// also those of bricks the host has not heard of yet (claimed since its last counter read-back: they have no region until then and
// keep recording their demand): left alone, their counters would add up over every launch until the next clean pass and the plan
// made from them would be inflated by that factor.  With every counter holding ONE launch's demand the planned capacities add up
// to at most 2 x slack x points + 128 x bricks < the pool size (hfpf.hip bin_pool_entries), so the 32-bit sums cannot wrap.
__global__ __launch_bounds__(256) void k_bin_place(const Tables t, const uint32_t n_planned, const uint32_t all_regions, const uint64_t pool,
                                                   const uint32_t* __restrict__ tile_sums)
{
    __shared__ uint32_t s_w[4];
    const uint32_t r0 = blockIdx.x * kBinPlanTile + threadIdx.x * 4u;
    if (blockIdx.x * kBinPlanTile < n_planned) {  // block-uniform
        uint32_t before = 0;
        for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256u) before += tile_sums[b];
        before = block_sum_256(before, s_w);
        uint32_t cap[4], mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            cap[k] = r0 + k < n_planned ? t.bin_capb[r0 + k] : 0u;
            mine += cap[k];
        }
        const uint32_t incl = wave_inclusive_scan(mine);
        if ((threadIdx.x & 63u) == 63u) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t off = before + incl - mine;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) off += s_w[w];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t r = r0 + k;
            if (r < n_planned) {
                t.bin_off[r] = off;
                if ((uint64_t)off + cap[k] > pool) t.bin_capb[r] = 0;
                off += cap[k];
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t r = r0 + k;
        if (r >= all_regions) continue;
        if (r >= n_planned) t.bin_capb[r] = 0;  // (an earlier launch may have planned more spare regions than this one)
        t.bin_fill[r] = 0;
        if (r == 0) t.ctr[C_OVF] = 0;  // the overflow list of the previous launch has been worked off (k_integrate_overflow)
    }
}

// ------------------------------------------------------------------------------------------------
// Link the log entries appended since the last clean into their cells' chains.  grid.y = log region;
// [first[r], last[r]] are 1-based global entry indices (empty when first > last).
struct LinkRanges {
    uint32_t first[kLogRegions];
    uint32_t last[kLogRegions];
};
__global__ __launch_bounds__(256) void k_link_log(const Tables t, const LinkRanges lr)
{
    const uint32_t r = blockIdx.y;
    const uint64_t e = (uint64_t)lr.first[r] + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e > lr.last[r]) return;
    uint32_t* w = reinterpret_cast<uint32_t*>(&t.log_pt[e]) + 3;
    const uint32_t v = *w;
    if (!(v & kLogUnlinked)) return;  // appended and chained by k_buffer
    const uint32_t slot = v & ~kLogUnlinked;
    *w = atomicExch(&t.buf_head[(uint64_t)slot * kChains + (e & (kChains - 1))], (uint32_t)e);  // entry e joins chain e mod 4 of its cell
}

// 125-bit occupancy stencil around (x,y,z): bit d = ((dx+2)*5 + (dy+2))*5 + (dz+2), the setK table order
// (grid.hpp:138-149); only cells with validCoord (grid.hpp:337) can be set.
// The window [c-2, c+2] spans at most two bricks per axis.  All directory entries (<= 8) are read first, then all plane
// masks (<= 20 u64), each batch independent loads, and the 5-bit z-runs are cut out with shifts: two memory round trips
// per call instead of one per (plane, brick), no data-dependent branches.
__device__ inline void neighbourhood(const GridParams& g, const Tables& t, int32_t x, int32_t y, int32_t z, uint64_t& lo, uint64_t& hi)
{
    lo = 0;
    hi = 0;
    const int32_t bx0 = (x - 2) >> 3, by0 = (y - 2) >> 3, bz0 = (z - 2) >> 3;  // arithmetic shifts: -1 below the grid
    const bool two_x = ((x + 2) >> 3) != bx0, two_y = ((y + 2) >> 3) != by0, two_z = ((z + 2) >> 3) != bz0;
    uint32_t bid[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int i = q >> 2, j = (q >> 1) & 1, k = q & 1;
        const int32_t bx = bx0 + i, by = by0 + j, bz = bz0 + k;
        const bool need = (i == 0 || two_x) && (j == 0 || two_y) && (k == 0 || two_z) && bx >= 0 && by >= 0 && bz >= 0 && bx < g.bdim[0] &&
                          by < g.bdim[1] && bz < g.bdim[2];
        uint32_t v = 0;
        if (need) v = t.dir[((uint32_t)bx * (uint32_t)g.bdim[1] + (uint32_t)by) * (uint32_t)g.bdim[2] + (uint32_t)bz];
        bid[q] = v == kLock ? 0u : v;
    }
    // plane masks of the 5 x-planes in the (up to) 2 x 2 bricks of the y/z window
    uint64_t pm[5][4];
#pragma unroll
    for (int a = 0; a < 5; a++) {
        const int32_t xx = x + a - 2;
        const bool vx = xx >= 0 && xx < g.dim[0];
        const bool hi_x = (xx >> 3) != bx0;
#pragma unroll
        for (int jk = 0; jk < 4; jk++) {
            const uint32_t b = hi_x ? bid[4 + jk] : bid[jk];
            uint64_t m = 0;
            if (vx && b) m = t.occ_mask[(uint64_t)b * 8 + ((uint32_t)xx & 7u)];
            pm[a][jk] = m;
        }
    }
    // z clipping: which of the 5 dz positions are valid cells
    uint32_t zclip = 0;
#pragma unroll
    for (int c = 0; c < 5; c++)
        if (z + c - 2 >= 0 && z + c - 2 < g.dim[2]) zclip |= 1u << c;
    const uint32_t zsh = (uint32_t)((z - 2) - bz0 * 8);  // 0..7: first z of the window inside the 16-wide line of the two z-bricks
#pragma unroll
    for (int a = 0; a < 5; a++) {
#pragma unroll
        for (int dy = 0; dy < 5; dy++) {
            const int32_t yy = y + dy - 2;
            const bool vy = yy >= 0 && yy < g.dim[1];
            const bool hi_y = (yy >> 3) != by0;
            const uint32_t sh = ((uint32_t)yy & 7u) * 8u;
            const uint64_t m0 = hi_y ? pm[a][2] : pm[a][0], m1 = hi_y ? pm[a][3] : pm[a][1];
            const uint32_t line = (uint32_t)((m0 >> sh) & 0xFFull) | ((uint32_t)((m1 >> sh) & 0xFFull) << 8);
            uint64_t f = vy ? (uint64_t)((line >> zsh) & 31u & zclip) : 0ull;
            const int d0 = (a * 5 + dy) * 5;
            if (d0 < 64) {
                lo |= f << d0;
                if (d0 + 5 > 64) hi |= f >> (64 - d0);
            } else {
                hi |= f << (d0 - 64);
            }
        }
    }
}

// K3: every occupied cell without a normal is a candidate (the reference's unprocessed_data_ set is
// a superset whose extra members fail the same gate, grid.hpp:315,352).
template <int TILES>
__global__ __launch_bounds__(256) void k_gate(const GridParams g, const Tables t, const uint32_t* __restrict__ cells_a, const uint64_t n_a,
                                              const uint32_t* __restrict__ cells_b, const uint64_t n_b, uint32_t* __restrict__ pend_out)
{
    // Input = the cells that failed the gate last time (cells_a) followed by the cells occupied since (cells_b), one launch.
    // TILES tiles per workgroup, one reservation per output list (hot list counters: see k_register).  The host uses
    // TILES = 4 only for large inputs: the stencil probe is latency-heavy and a small grid needs every workgroup it can get.
    const uint64_t n_cells = n_a + n_b;
    uint64_t key_[TILES];
    uint32_t slot_[TILES];
    uint32_t n_pass[TILES], n_pend[TILES];
#pragma unroll
    for (int tt = 0; tt < TILES; tt++) {
        const uint64_t j = ((uint64_t)blockIdx.x * TILES + tt) * 256u + threadIdx.x;
        key_[tt] = 0;
        slot_[tt] = 0;
        n_pass[tt] = n_pend[tt] = 0;
        if (j < n_cells) {
            const uint32_t slot = j < n_a ? cells_a[j] : cells_b[j - n_a];
            slot_[tt] = slot;
            if (!(t.info[slot] & kNormal)) {
                int32_t x, y, z;
                slot_coords(g, t, slot, x, y, z);
                uint64_t lo, hi;
                neighbourhood(g, t, x, y, z, lo, hi);
                const int total = __popcll(lo) + __popcll(hi);
                if (total > g.gate) n_pass[tt] = 1;
                else n_pend[tt] = 1;  // still without a normal: look at it again next pass (its neighbourhood may fill up)
                key_[tt] = HFPF_MORTON_IDS ? morton_key(x, y, z) : make_key(g, x, y, z);  // the order the pass numbers its records in
            }
        }
    }
    __shared__ TileReserveScratch<TILES> trs;
    unsigned long long ci[TILES], pi[TILES];
    block_reserve_tiles<TILES>(&t.ctr[C_CAND], n_pass, ci, trs);
    block_reserve_tiles<TILES>(&t.ctr[C_PEND], n_pend, pi, trs);
#pragma unroll
    for (int tt = 0; tt < TILES; tt++) {
        if (n_pass[tt]) t.cand_key[ci[tt]] = key_[tt];  // capacity = max_occ >= cells examined
        if (n_pend[tt]) pend_out[pi[tt]] = slot_[tt];
    }
}

// Candidates of the running pass as the kernels after the gate see them: the device counter, clamped to the room left in the
// normal records (the host learns the exact number at its next read-back; E_NORMALS reports the clamp).
__device__ __forceinline__ uint64_t cand_count(const Tables& t, uint64_t base)
{
    const uint64_t n = t.ctr[C_CAND];
    const uint64_t room = t.max_normals > base ? t.max_normals - base : 0;
    return n < room ? n : room;
}
// Head of a clean pass, one launch: all-ones sentinels behind the candidate keys the gate is about to write (they sort to the
// end), the pass's list counters back to zero, and (k_materialize_new in the comments: this part) the cells occupied since the
// previous pass -- the new tail of occ_list, imported ones included -- that carry a pre-dependant get it filed as their list: entry,
// info word, flag bit, and a place in prereg_list (what a compacting rebuild re-creates the entry from); one reservation per list
// and workgroup.  lists_only: a compacting rebuild follows in this pass and writes every entry and info word itself.  No room in
// dep[] is E_DEP, which the host answers by compacting.  n_in >= n_new (the gate's input contains the new cells).
__global__ __launch_bounds__(256) void k_clean_begin(const Tables t, const uint64_t n_in, const uint32_t* __restrict__ new_cells, const uint64_t n_new,
                                                     const uint32_t lists_only)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_in) t.cand_key[i] = ~0ull;
    if (i == 0) {
        t.ctr[C_CAND] = 0;
        t.ctr[C_PEND] = 0;
        t.ctr[C_TOUCHED] = 0;
    }
    if (i < (uint64_t)kLogRegions) t.log_ctr[i * 16 + 4] = 0;  // touched cells in single-run bricks (k_depinc_offsets)
    if ((uint64_t)blockIdx.x * blockDim.x >= n_new) return;  // block-uniform
    uint32_t slot = 0, nid = 0;
    if (i < n_new) {
        slot = new_cells[i];
        nid = t.pre_dep[slot];
    }
    __shared__ BlockReserveScratch brs;
    const unsigned long long li = block_reserve(&t.ctr[C_PREREG], nid != 0, brs);
    const unsigned long long off = lists_only ? 0ull : block_reserve(&t.ctr[C_DEP], nid != 0, brs);
    if (!nid) return;
    if (li < t.max_reg) t.prereg_list[li] = slot;
    else atomicOr(&t.ctr[C_ERR], (unsigned long long)E_REG);
    set_dep_flag(t, slot);
    if (lists_only) return;
    if (off >= t.max_dep) {
        atomicOr(&t.ctr[C_ERR], (unsigned long long)E_DEP);
        return;
    }
    t.dep[off] = make_dep_entry(t, nid);
    t.info[slot] = (t.info[slot] & 3ull) | (1ull << kDepCntShift) | ((uint64_t)off << kDepOffShift);
}

// K4: one thread per candidate, in ascending order of the sorted keys (Z-order codes with HFPF_MORTON_IDS); record id = base + rank + 1.
__global__ __launch_bounds__(128) void k_normal(const GridParams g, const Tables t, const uint64_t* __restrict__ sorted_keys,
                                                const uint64_t n_cand_arg, const uint64_t base)
{
    // n_cand_arg = kCountOnDevice: the grid covers an upper bound (the gate's input size), the sorted keys behind the real
    // candidates are all-ones sentinels
    const uint64_t n_cand = n_cand_arg == kCountOnDevice ? cand_count(t, base) : n_cand_arg;
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0 && n_cand_arg == kCountOnDevice) {  // publish the new record count (nothing in this pass reads it on the device)
        if (t.ctr[C_CAND] > n_cand) atomicOr(&t.ctr[C_ERR], (unsigned long long)E_NORMALS);
        t.ctr[C_NORMALS] = base + n_cand;
    }
    if (r >= n_cand) return;
    int32_t x, y, z;
    uint64_t key = sorted_keys[r];
    if (HFPF_MORTON_IDS) {
        morton_coords(key, x, y, z);
        key = make_key(g, x, y, z);  // what the record keeps: the canonical (x, y, z) key (extract order, registration contests)
    } else {
        key_coords(g, key, x, y, z);
    }
    const uint32_t slot = slot_lookup(g, t, x, y, z);
    uint64_t lo, hi;
    neighbourhood(g, t, x, y, z, lo, hi);
    // the cell centre is separable: 5 values per axis (grid.hpp:131-135 evaluated once each) instead of 125 x 3
    float cxs[5], cys[5], czs[5];
#pragma unroll
    for (int a = 0; a < 5; a++) {
        const F3 c = voxel_center(g, x + a - 2, y + a - 2, z + a - 2);
        cxs[a] = c.x;
        cys[a] = c.y;
        czs[a] = c.z;
    }
    Moments m;
    m.clear();
    int total = 0;
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
        for (int bq = 0; bq < 5; bq++) {
            const int d0 = (a * 5 + bq) * 5;
            const uint32_t bits5 = (uint32_t)((d0 < 64 ? (lo >> d0) | (d0 + 5 > 64 ? hi << (64 - d0) : 0ull) : hi >> (d0 - 64)) & 31ull);
#pragma unroll
            for (int c = 0; c < 5; c++)
                if ((bits5 >> c) & 1u) {
                    m.add(F3{cxs[a], cys[bq], czs[c]}, g.cov_shifted != 0);  // grid.hpp:364-369, setK order (x outer, z inner)
                    total++;
                }
        }
    F3 normal = m.normal(total);
    const F3 centre = voxel_center(g, x, y, z);  // grid.hpp:391
    const uint32_t ff = t.first_frame[slot];
    F3 vp = {0.f, 0.f, 0.f};
    if (ff != kNoFrame && ff < t.max_frames) vp = F3{t.frame_vp[3 * (uint64_t)ff], t.frame_vp[3 * (uint64_t)ff + 1], t.frame_vp[3 * (uint64_t)ff + 2]};
    normal = orient_normal(normal, vp, centre);
    const uint64_t nid = base + r + 1;
    t.nv_key[nid] = key;
    t.nv_slot[nid] = slot;
    t.nv_c[3 * nid + 0] = centre.x;
    t.nv_c[3 * nid + 1] = centre.y;
    t.nv_c[3 * nid + 2] = centre.z;
    t.nv_n[3 * nid + 0] = normal.x;
    t.nv_n[3 * nid + 1] = normal.y;
    t.nv_n[3 * nid + 2] = normal.z;
    {  // the per-voxel half of the cylinder test (grid.hpp:40-49), evaluated once here and copied into dependant entries
        F3 a, ab;
        float dd;
        line_of(g, centre, normal, a, ab, dd);
        t.nv_line[2 * nid] = make_float4(__uint_as_float((uint32_t)nid), a.x, a.y, a.z);  // = DepEntry{sid, a, ab, |ab|^2}
        t.nv_line[2 * nid + 1] = make_float4(ab.x, ab.y, ab.z, dd);
    }
    t.stat_id[slot] = (uint32_t)nid;
    atomicOr(reinterpret_cast<unsigned int*>(&t.info[slot]), 2u);  // normal_found, grid.hpp:398
    atomicOr(reinterpret_cast<unsigned long long*>(&t.nd_mask[((uint64_t)(slot >> 9) * 8u + ((uint32_t)x & 7u)) * 2]),
             1ull << ((((uint32_t)y & 7u) << 3) | ((uint32_t)z & 7u)));
}

// K5: one thread per (new normal, line step).  Occupancy is frozen during a clean pass, so the steps are
// independent; "last registrant wins" on unoccupied cells (grid.hpp:443-449) is an atomicMax over record ids -- they ascend from
// pass to pass -- with the contests inside a pass settled by the canonical (x, y, z) key (see below: ids follow the Z-order there).
template <int kRegTiles>
__global__ __launch_bounds__(256) void k_register(const GridParams g, const Tables t, const uint64_t n_cand_arg, const uint64_t base, const uint32_t count_deps)
{
    // count_deps: the incremental dependant-table update follows (no compacting rebuild was decided): a registration on an occupied
    // cell is counted into the cell's scratch word right here, and the first one of a cell files the cell in touched_list -- what a
    // separate pass over reg_occ (k_depinc_count) did until round 4, one launch and one re-read of the list earlier.
    const uint64_t n_cand = n_cand_arg == kCountOnDevice ? cand_count(t, base) : n_cand_arg;  // the same in every thread
    if (n_cand == 0) return;
    // Step-major mapping: a wave holds 64 key-adjacent voxels at the SAME step, so its targets sit in the same few bricks.
    // A workgroup takes kRegTiles consecutive 256-voxel tiles of one step and reserves its list entries once: same-address device
    // atomics retire one per ~12 ns whoever issues them.  The host takes 8 tiles for passes of millions of candidates (6.7 K
    // reservations in the bench's first pass; 27 K would be a third of a millisecond) and 2 for the small steady ones, where the
    // ~20 K candidates of a pass would otherwise sit on 70 of the 256 CUs.
    const uint32_t steps = 2u * (uint32_t)g.K + 1u;
    const uint64_t tile_items = 256ull * kRegTiles;
    const uint64_t per_step = ((n_cand + tile_items - 1) / tile_items) * tile_items;  // whole workgroups per step keep them uniform in i
    const uint64_t blk_first = (uint64_t)blockIdx.x * tile_items;
    const uint32_t step_idx = (uint32_t)(blk_first / per_step);
    const int i = (int)step_idx - g.K;
    uint32_t slot_[kRegTiles], nid_[kRegTiles];
    uint32_t f_occ = 0, f_fresh = 0;
#pragma unroll
    for (int tt = 0; tt < kRegTiles; tt++) {
        const uint64_t r = blk_first % per_step + (uint64_t)tt * 256u + threadIdx.x;
        bool want = r < n_cand && step_idx < steps;
        uint64_t nid = 0;
        int32_t xx = 0, yy = 0, zz = 0;
        if (want) {
            nid = base + r + 1;
            const F3 c = F3{t.nv_c[3 * nid], t.nv_c[3 * nid + 1], t.nv_c[3 * nid + 2]};
            const F3 n = F3{t.nv_n[3 * nid], t.nv_n[3 * nid + 1], t.nv_n[3 * nid + 2]};
            const F3 nb = line_step(g, c, n, i);  // grid.hpp:405
            want = valid_point(g, nb);            // grid.hpp:406
            voxel_coords(g, nb, xx, yy, zz);      // grid.hpp:409
            want = want && xx != INT_MIN && yy != INT_MIN && zz != INT_MIN && valid_coord(g, xx, yy, zz);  // grid.hpp:410
        }
        const uint32_t bidx = want ? brick_index(g, xx, yy, zz) : 0u;
        const uint32_t b = brick_acquire_wave(t, bidx, want);
        want = want && b != 0;
        const uint32_t slot = b * kBrickCells + local_index(xx, yy, zz);
        uint64_t o_plane, o_bit;
        slot_plane_bit(slot, o_plane, o_bit);
        const bool occ = want && (t.occ_mask[o_plane] & o_bit);
        slot_[tt] = slot;
        nid_[tt] = (uint32_t)nid;
        if (occ) {
            f_occ |= 1u << tt;
            if (count_deps && atomicAdd(&t.dep_tmp[slot], 1u) == 0u) f_fresh |= 1u << tt;  // (scratch words are zero between passes)
        }
        if (want && !occ) {
            // An unoccupied target keeps ONE dependant, in pre_dep[slot] and nowhere else until the cell is occupied (k_materialize_new).
            const uint32_t old = atomicMax(&t.pre_dep[slot], (uint32_t)nid);
#ifdef HFPF_TEST_NO_KEY_CONTEST  // (tests only: shows that test_registration_contest_... notices a contest left to the ids)
            if (false) {
#else
            if (HFPF_MORTON_IDS && old > base && old != (uint32_t)nid) {
#endif
                // Another voxel of this pass registered here as well.  Ids follow the Z-order inside a pass, the contest is about the
                // canonical order: the registrant with the LARGEST (x, y, z) KEY stays (the reference walks its candidates in that
                // order and the last one overwrites, grid.hpp:443-449).  The atomicMax above has put the larger ID into the cell and
                // told this lane whom it met; the lane now sees to it that the cell holds a key at least as large as the better of the
                // two.  Whoever displaces a holder learns of it the same way and takes over that duty, so once every registrant is
                // through, the cell holds the largest key of the pass.  (Uncontested cells -- most -- cost the one atomic they always did.)
                const uint64_t my_key = t.nv_key[nid], old_key = t.nv_key[old];
                const uint32_t best = old_key > my_key ? old : (uint32_t)nid;
                const uint64_t best_key = old_key > my_key ? old_key : my_key;
                uint32_t cur = max(old, (uint32_t)nid);  // what the atomicMax left behind
                for (int spin = 0; spin < kMaxSpin; ++spin) {
                    if (cur == best || t.nv_key[cur] >= best_key) break;  // (every value the cell takes in this pass is an id of this pass)
                    const uint32_t seen = atomicCAS(&t.pre_dep[slot], cur, best);
                    if (seen == cur) break;
                    cur = seen;
                }
            }
        }
    }
    __shared__ TileReserveScratch<kRegTiles> trs;
    uint32_t n_occ[kRegTiles];
    unsigned long long ri[kRegTiles];
#pragma unroll
    for (int tt = 0; tt < kRegTiles; tt++) n_occ[tt] = (f_occ >> tt) & 1u;
    block_reserve_tiles<kRegTiles>(&t.ctr[C_REG], n_occ, ri, trs);
    bool overflow = false;
#pragma unroll
    for (int tt = 0; tt < kRegTiles; tt++) {
        if (n_occ[tt]) {
            if (ri[tt] < t.max_reg) t.reg_occ[ri[tt]] = make_uint2(slot_[tt], nid_[tt]);  // dependants.push_back, grid.hpp:417
            else overflow = true;
        }
    }
    if (overflow) atomicOr(&t.ctr[C_ERR], (unsigned long long)E_REG);
    if (count_deps) {  // (wave-uniform) the cells this workgroup was the first to count a registration for
#pragma unroll
        for (int tt = 0; tt < kRegTiles; tt++) n_occ[tt] = (f_fresh >> tt) & 1u;
        block_reserve_tiles<kRegTiles>(&t.ctr[C_TOUCHED], n_occ, ri, trs);  // capacity max_touched = max_reg >= the registrations of any pass
#pragma unroll
        for (int tt = 0; tt < kRegTiles; tt++)
            if (n_occ[tt]) t.touched_list[ri[tt]] = slot_[tt];
    }
}

// K5b: buffer replay (grid.hpp:418-440), cell-centric.  The reference replays a cell's buffer once per voxel that
// registers on it; here one thread walks the chain of a touched cell ONCE and tests every buffered point against all
// of the cell's registrants of this pass (dependant entries with record id > base), four at a time in registers.
// Random 16-byte chain reads are the cost (sector amplification makes them HBM-bound), so reading each entry once
// instead of once per registrant is the lever.  Runs after the dependant table has been updated.
template <bool COLOR>
__global__ __launch_bounds__(256) void k_replay(const GridParams g, const Tables t, const uint32_t* __restrict__ cells, const uint32_t use_marks,
                                                const uint32_t skip_single, const uint64_t n_touched_arg, const uint64_t base)
{
    // use_marks: the incremental update left kTouchedMark | old list length in the cell's scratch word.  It appends, so the
    // registrants of this pass are the entries behind that position and the walk over the older ones -- one dependent 32-byte read
    // each -- is skipped; the note is cleared here.  Without marks (compacting rebuild: any order) every entry is looked at and
    // filtered by record id.  skip_single: cells of single-run bricks are left (with their note) to the streaming replay.
    __shared__ unsigned long long queue[4][kQueueRows * kQueueStride];
    unsigned long long* q = queue[threadIdx.x >> 6];
    const uint64_t n_touched = n_touched_arg == kCountOnDevice ? (uint64_t)t.ctr[C_TOUCHED] : n_touched_arg;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t j = gid / kChains;           // cell
    const uint32_t sub = (uint32_t)(gid % kChains);  // which of the cell's interleaved chains this lane walks
    uint32_t slot = 0, cnt = 0, head = 0;
    uint64_t off = 0;
    bool mine = j < n_touched;
    if (mine) {
        slot = cells[j];  // touched cells in slot (brick-major) order: adjacent lanes walk chains that share cache lines
        if (skip_single && t.run_cnt[slot >> 9] == 1u) mine = false;  // (with its note) left to the streaming replay
    }
    if (mine) {
        const uint64_t info = t.info[slot];
        cnt = (uint32_t)((info >> kDepCntShift) & kDepCntMask);
        off = info >> kDepOffShift;
        head = t.buf_head[(uint64_t)slot * kChains + sub];
    }
    {  // a cell none of whose chains holds a point (an unoccupied cell that gained its single dependant) has nothing to replay
        uint32_t any = head;
#pragma unroll
        for (uint32_t o = 1; o < kChains; o <<= 1) any |= (uint32_t)__shfl_xor((int)any, (int)o);
        if (any == 0u) cnt = 0;
    }
    uint32_t replayed = 0;
    uint32_t next = 0;  // next dependant entry to look at
    if (use_marks && mine) {
        const uint32_t m = t.dep_tmp[slot];  // the lanes of a cell read it in the same instruction, then lane 0 clears it
        if (m & kTouchedMark) {
            next = min(m & kDepOldMax, cnt);
            if (sub == 0) t.dep_tmp[slot] = 0;
        }
    }
    // wave-uniform outer loop: every lane keeps calling the flush helper until all lanes are done
    while (__ballot(next < cnt) != 0) {
        constexpr int B = HFPF_REPLAY_B;
        uint32_t sid[B];
        F3 la[B], lab[B];
        float ldd[B];
        LineDiv ldv[B];
        bool use[B];
        const float4* __restrict__ dep4 = reinterpret_cast<const float4*>(t.dep);
        float4 e0[B], e1[B];
#pragma unroll
        for (int k = 0; k < B; k++) {  // the next B entries, B independent reads in flight together
            e0[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            e1[k] = make_float4(0.f, 0.f, 1.f, 1.f);
            if (next + (uint32_t)k < cnt) {
                e0[k] = dep4[2 * (off + next + k)];
                e1[k] = dep4[2 * (off + next + k) + 1];
            }
        }
        bool any = false;
#pragma unroll
        for (int k = 0; k < B; k++) {
            sid[k] = __float_as_uint(e0[k].x);
            use[k] = next + (uint32_t)k < cnt && sid[k] > base;  // a registrant of this pass
            any = any || use[k];
            la[k] = F3{e0[k].y, e0[k].z, e0[k].w};
            lab[k] = F3{e1[k].x, e1[k].y, e1[k].z};
            ldd[k] = e1[k].w;
        }
        next = min(cnt, next + (uint32_t)B);
        StatDeltaT<COLOR> d[B];
#pragma unroll
        for (int k = 0; k < B; k++) {
            stat_delta_zero(d[k]);
            ldv[k] = line_div_of(ldd[k]);  // the divisor's share of the division, once per registrant (geometry.hpp)
        }
        if (any) {
            uint32_t e = head;
            // (Every link is a head that k_buffer / direct_buffer exchanged, i.e. the index of an entry they wrote.  A range check per
            // hop was measured in round 4: +12 % on this kernel, whose time is the chain of dependent reads; left out.)
            while (e) {
                const float4 p = t.log_pt[e];
                const uint32_t rgb = COLOR ? t.log_rgb[e] : 0u;
                const F3 pt = F3{p.x, p.y, p.z};
#pragma unroll
                for (int k = 0; k < B; k++) {
                    if (use[k]) {
                        float sp, distf;
                        if (line_member_hoisted(g, pt, la[k], lab[k], ldv[k], sp, distf)) stat_delta_add(d[k], pair_delta(g, sp, distf), rgb);
                    }
                }
                e = __float_as_uint(p.w);  // one 16-byte read per hop: the link travels with the point
            }
        }
#pragma unroll
        for (int k = 0; k < B; k++) {
            // sum the sub-chains of the cell (kChains adjacent lanes) before the flush: one record update per (cell, registrant)
#pragma unroll
            for (int w = 0; w < kStatUsed; w++) {
                long long v = d[k].v[w];
#pragma unroll
                for (uint32_t o = 1; o < kChains; o <<= 1) v += __shfl_xor(v, (int)o);
                d[k].v[w] = v;
            }
            if constexpr (COLOR) {
#pragma unroll
                for (int w = 0; w < 3; w++) {
                    long long v = d[k].rgb[w];
#pragma unroll
                    for (uint32_t o = 1; o < kChains; o <<= 1) v += __shfl_xor(v, (int)o);
                    d[k].rgb[w] = v;
                }
            }
            const bool member = sub == 0 && d[k].v[SW_COUNT] != 0;
            if (member) replayed += (uint32_t)d[k].v[SW_COUNT];
            wave_flush_members(t, q, member, d[k], sid[k]);
        }
    }
    // Diagnostic count of replayed members: reduced per workgroup and striped over the 64 counter lines of log_ctr (word 1 of
    // each line; the host sums them in read_counters).  One counter bumped once per wave would be ~1e5 same-address atomics
    // at ~12 ns each in the first pass -- most of this kernel's time.
    __shared__ unsigned int s_rep;
    if (threadIdx.x == 0) s_rep = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) replayed += __shfl_down(replayed, o);
    if ((threadIdx.x & 63u) == 0 && replayed) atomicAdd(&s_rep, replayed);
    __syncthreads();
    if (threadIdx.x == 0 && s_rep) atomicAdd(&t.log_ctr[(blockIdx.x & (kLogRegions - 1)) * 16 + 1], (unsigned long long)s_rep);
}

// ---- dependant table rebuild --------------------------------------------------------------------
__device__ __forceinline__ uint32_t reg_slot(const Tables& t, uint64_t j, uint64_t n_reg) { return j < n_reg ? t.reg_occ[j].x : t.prereg_list[j - n_reg]; }

// A cell's dependant list owns a power-of-two number of entries (1, 2, 4, ...: dep_capacity of its length).  A clean pass that
// extends the list appends IN PLACE while the new length fits and relocates it -- into a block of the next capacity -- only when it
// does not, so a list that grows by an entry or two per pass moves a logarithmic number of times instead of once per pass (on the
// 0.5 mm workload a touched cell holds ~8 entries: relocating them all every pass was most of k_depinc_offsets' 32-byte traffic).
// Every writer of info words allocates by this rule: k_depinc_offsets, the compacting rebuild (k_dep_offsets), k_clean_begin (1).
__device__ __forceinline__ uint32_t dep_capacity(uint32_t cnt) { return cnt <= 1u ? cnt : 1u << (32 - __clz((int)(cnt - 1u))); }

__global__ __launch_bounds__(256) void k_dep_count(const Tables t, const uint64_t n_reg, const uint64_t n_pre)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool fresh = false;
    uint32_t slot = 0;
    if (j < n_reg + n_pre) {
        slot = reg_slot(t, j, n_reg);
        fresh = atomicAdd(&t.dep_tmp[slot], 1u) == 0u;
    }
    __shared__ BlockReserveScratch brs;
    const unsigned long long ti = block_reserve(&t.ctr[C_TOUCHED], fresh, brs);
    if (fresh) t.touched_list[ti] = slot;  // capacity max_touched = max_reg >= distinct cells among n_reg + n_pre entries (host-checked: n_reg + n_pre <= max_reg)
}

__global__ __launch_bounds__(256) void k_dep_offsets(const Tables t, const uint64_t n_touched)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = j < n_touched;
    const uint32_t slot = act ? t.touched_list[j] : 0u;
    uint32_t cnt = act ? t.dep_tmp[slot] : 0u;
    if (cnt > kDepCntMask) {
        atomicOr(&t.ctr[C_ERR], (unsigned long long)E_DEPCNT);
        cnt = (uint32_t)kDepCntMask;
    }
    const unsigned long long off = wave_reserve_n(&t.ctr[C_DEP], dep_capacity(cnt));
    if (!act) return;
    t.info[slot] = (t.info[slot] & 3ull) | ((uint64_t)cnt << kDepCntShift) | ((uint64_t)off << kDepOffShift);
    if (cnt) set_dep_flag(t, slot);
    t.dep_tmp[slot] = 0;
}

__global__ __launch_bounds__(256) void k_dep_fill(const Tables t, const uint64_t n_reg, const uint64_t n_pre)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_reg + n_pre) return;
    const uint32_t slot = reg_slot(t, j, n_reg);
    const uint32_t nid = j < n_reg ? t.reg_occ[j].y : t.pre_dep[slot];
    const uint64_t info = t.info[slot];
    const uint32_t k = atomicAdd(&t.dep_tmp[slot], 1u);
    if (k >= ((info >> kDepCntShift) & kDepCntMask)) return;  // clamped list
    t.dep[(info >> kDepOffShift) + k] = make_dep_entry(t, nid);
}

__global__ __launch_bounds__(256) void k_dep_reset(const Tables t, const uint64_t n_touched)
{
    const uint64_t n = n_touched == kCountOnDevice ? (uint64_t)t.ctr[C_TOUCHED] : n_touched;
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) t.dep_tmp[t.touched_list[j]] = 0;
}


// ---- incremental dependant-table update ---------------------------------------------------------------
// A clean pass only appends registrations to the lists of occupied cells, so only the cells touched by THIS pass are looked at
// (k_register counts the pass's registrations per cell and files the touched cells): k_depinc_offsets extends a list in place while it
// fits the power-of-two block it owns (dep_capacity) and otherwise copies it into a fresh block at the end of dep[], k_depinc_fill
// writes the new entries.  The space of a relocated list is garbage until the next full rebuild (k_dep_count / k_dep_offsets /
// k_dep_fill), which the host runs when dep[] fills up.

__global__ __launch_bounds__(256) void k_depinc_offsets(const Tables t, const uint64_t n_touched_arg)
{
    const uint64_t n_touched = n_touched_arg == kCountOnDevice ? (uint64_t)t.ctr[C_TOUCHED] : n_touched_arg;
    uint32_t slot_[kListTiles], old_cnt_[kListTiles], new_cnt_[kListTiles], take_[kListTiles];
    uint64_t info_[kListTiles];
#pragma unroll
    for (int tt = 0; tt < kListTiles; tt++) {
        const uint64_t j = ((uint64_t)blockIdx.x * kListTiles + tt) * 256u + threadIdx.x;
        slot_[tt] = 0, old_cnt_[tt] = 0, new_cnt_[tt] = 0, take_[tt] = 0, info_[tt] = 0;
        if (j < n_touched) {
            slot_[tt] = t.touched_list[j];
            info_[tt] = t.info[slot_[tt]];
            old_cnt_[tt] = (uint32_t)((info_[tt] >> kDepCntShift) & kDepCntMask);
            new_cnt_[tt] = old_cnt_[tt] + t.dep_tmp[slot_[tt]];
            // the list stays where it is while it fits the block it owns (dep_capacity); otherwise it moves into a block of the next capacity
            if (new_cnt_[tt] > dep_capacity(old_cnt_[tt])) take_[tt] = dep_capacity(min(new_cnt_[tt], (uint32_t)kDepCntMask));
        }
    }
    __shared__ TileReserveScratch<kListTiles> trs;
    unsigned long long off_[kListTiles];
    uint32_t n_single = 0;
    block_reserve_tiles<kListTiles>(&t.ctr[C_DEP], take_, off_, trs);  // one atomic per workgroup; lists of neighbouring cells stay adjacent
#pragma unroll
    for (int tt = 0; tt < kListTiles; tt++) {
        const uint64_t j = ((uint64_t)blockIdx.x * kListTiles + tt) * 256u + threadIdx.x;
        if (j >= n_touched) continue;
        const uint32_t slot = slot_[tt], old_cnt = old_cnt_[tt], new_cnt = new_cnt_[tt];
        const uint64_t info = info_[tt], old_off = info >> kDepOffShift;
        const bool moves = take_[tt] != 0;
        const unsigned long long off = moves ? off_[tt] : old_off;
        // No room left in dep[], or an old list too long for the 15-bit note of the cursor word (kDepOldMax; the info word itself
        // counts to 65535): E_DEP, which the host answers with the compacting rebuild (k_dep_*, no such limit).  More than 65535
        // entries on one cell fit nowhere: E_DEPCNT, a capacity error.  (A cell's registrants all lie within K cells of it along
        // their normals, a few hundred voxels at most, so neither limit is reachable by a real scene.)
        if ((moves && off + take_[tt] > t.max_dep) || new_cnt > kDepCntMask || old_cnt > kDepOldMax) {
            atomicOr(&t.ctr[C_ERR], (unsigned long long)(new_cnt > kDepCntMask ? E_DEPCNT : E_DEP));
            t.dep_tmp[slot] = kDepPoison;  // k_depinc_fill skips this cell
            continue;
        }
        n_single += t.run_cnt[slot >> 9] == 1u ? 1u : 0u;
        if (moves) {
            for (uint32_t k = 0; k < old_cnt; k++) t.dep[off + k] = t.dep[old_off + k];
        }
        t.info[slot] = (info & 3ull) | ((uint64_t)new_cnt << kDepCntShift) | ((uint64_t)off << kDepOffShift);
        if (old_cnt == 0 && new_cnt) set_dep_flag(t, slot);
        t.dep_tmp[slot] = (old_cnt << 16) | old_cnt;  // old length | append cursor
    }
    // cells in single-run bricks, for the host's choice of replay form: word 4 of the striped counter lines, one atomic per workgroup
    __shared__ unsigned int s_single;
    if (threadIdx.x == 0) s_single = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_single += __shfl_down(n_single, o);
    if ((threadIdx.x & 63u) == 0 && n_single) atomicAdd(&s_single, n_single);
    __syncthreads();
    if (threadIdx.x == 0 && s_single) atomicAdd(&t.log_ctr[(blockIdx.x & (kLogRegions - 1)) * 16 + 4], (unsigned long long)s_single);
}

__global__ __launch_bounds__(256) void k_depinc_fill(const Tables t, const uint64_t reg_first, const uint64_t n_reg_arg)
{
    const uint64_t n_reg = n_reg_arg == kCountOnDevice ? min((uint64_t)t.ctr[C_REG], t.max_reg) : n_reg_arg;
    const uint64_t j = reg_first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_reg) return;
    const uint2 r = t.reg_occ[j];
    const uint32_t v = atomicAdd(&t.dep_tmp[r.x], 1u);
    if (v & kDepPoison) return;
    const uint32_t k = v & 0xFFFFu;
    const uint64_t info = t.info[r.x];
    t.dep[(info >> kDepOffShift) + k] = make_dep_entry(t, r.y);
    // exactly (new length - old length) tickets are drawn per cell: whoever draws the last one turns the cursor word into the
    // note for the replay -- this cell gained registrants, they sit behind entry `old length` -- which the replay clears again
    if (k + 1 == (uint32_t)((info >> kDepCntShift) & kDepCntMask)) t.dep_tmp[r.x] = kTouchedMark | ((v >> 16) & kDepOldMax);
}


// ---- K6 extract -----------------------------------------------------------------------------------
// Keys of the records that downloadData would emit: x<xdim && y<ydim && z<zdim (grid.hpp:463-465);
// others get the all-ones key and sort to the end.
// The reference's alternate extractors (grid.hpp:491-601) are this one scan with two knobs (= hfpf_extract_opts):
//   min_count           downloadHQ(cloud, threshold): `if (data->count < threshold) continue;` -- applied HERE, so filtered
//                       rows never reach the sort's front, the row kernel or the host
//   classify / paint    downloadClassified / download(XYZRGB): colour coding in k_extract_rows
struct ExtractOpts {
    double min_count;
    int32_t classify_threshold;  // < 0: off; else rows with count > threshold are painted red, the others white (grid.hpp:527-534)
    int32_t paint_white;         // 1: r = g = b = 255 as downloadHQ / downloadClassified set it (grid.hpp:527-529,558-560)
};
constexpr int kExtractTiles = 8;  // 256-record tiles per workgroup of k_extract_keys: one atomic on the row counter per workgroup
__global__ __launch_bounds__(256) void k_extract_keys(const GridParams g, const Tables t, const unsigned long long* __restrict__ stats,
                                                      const uint64_t n_normals, const ExtractOpts opt, uint64_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals)
{
    __shared__ unsigned int s_rows;
    if (threadIdx.x == 0) s_rows = 0;
    __syncthreads();
    uint32_t n_valid = 0;
#pragma unroll
    for (int tt = 0; tt < kExtractTiles; tt++) {
        const uint64_t j = ((uint64_t)blockIdx.x * kExtractTiles + tt) * 256u + threadIdx.x;
        if (j >= n_normals) continue;
        const uint64_t nid = j + 1;
        const uint64_t key = t.nv_key[nid];
        int32_t x, y, z;
        key_coords(g, key, x, y, z);
        uint32_t valid = valid_coord(g, x, y, z) ? 1u : 0u;
        if (valid && opt.min_count > 0.0) {
            const long long cnt = (long long)stats[nid * kStatWords + SW_COUNT];
            if ((double)(int)cnt < opt.min_count) valid = 0;  // int count against a double threshold, grid.hpp:561
        }
        keys[j] = valid ? key : ~0ull;
        vals[j] = (uint32_t)nid;
        n_valid += valid;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_valid += __shfl_down(n_valid, o);
    if ((threadIdx.x & 63u) == 0 && n_valid) atomicAdd(&s_rows, n_valid);
    __syncthreads();
    if (threadIdx.x == 0 && s_rows) atomicAdd(&t.ctr[C_ROWS], (unsigned long long)s_rows);  // (one same-address device atomic costs ~12 ns)
}

struct Row {  // = hfpf_row
    int32_t ix, iy, iz;
    uint32_t count;
    float x, y, z;
    float nx, ny, nz;
    float sdx, sdy, sdz;
    float mean_dist, sd_dist;
    uint32_t rgb;
};
static_assert(sizeof(Row) == 64, "row is 64 bytes");

__global__ __launch_bounds__(256) void k_extract_rows(const GridParams g, const Tables t, const unsigned long long* __restrict__ stats,
                                                      const uint64_t n_rows, const ExtractOpts opt, const uint64_t* __restrict__ keys,
                                                      const uint32_t* __restrict__ vals, Row* __restrict__ rows)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_rows) return;
    const uint64_t nid = vals[j];
    Row r;
    key_coords(g, keys[j], r.ix, r.iy, r.iz);
    const long long* s = reinterpret_cast<const long long*>(&stats[nid * kStatWords]);
    const long long cnt = s[SW_COUNT];
    r.count = (uint32_t)cnt;
    r.nx = t.nv_n[3 * nid];
    r.ny = t.nv_n[3 * nid + 1];
    r.nz = t.nv_n[3 * nid + 2];
    if (cnt <= 0) {  // count==0 rows emit the zero centroid (grid.hpp:472-476)
        r.x = r.y = r.z = 0.f;
        r.sdx = r.sdy = r.sdz = 0.f;
        r.mean_dist = r.sd_dist = 0.f;
        r.rgb = 0;
    } else {
        const float4 l0 = t.nv_line[2 * nid], l1 = t.nv_line[2 * nid + 1];
        const double ax = l0.y, ay = l0.z, az = l0.w, abx = l1.x, aby = l1.y, abz = l1.z;
        const double inv = 1.0 / (double)cnt;
        // every projection is a - s*ab (stats.hpp): centroid = a - E[s]*ab, per-axis variance = ab_i^2 * var(s)
        const double em = ((double)s[SW_S] / (double)g.fs_scale) * inv;  // mean of s, or of u = s - 0.5 (stats.hpp)
        const double es = HFPF_CENTERED_MOMENTS ? 0.5 + em : em;
        r.x = (float)(ax - es * abx);
        r.y = (float)(ay - es * aby);
        r.z = (float)(az - es * abz);
        double vs = ((double)s[SW_SS] / (double)g.fss_scale) * inv - em * em;
        const double md = ((double)s[SW_D] / (double)g.fd_scale) * inv;
        double vd = ((double)s[SW_DD] / (double)g.fdd_scale) * inv - md * md;
        if (cnt == 1) vs = vd = 0.0;  // the recurrence gives exactly 0 for a single sample
        vs = fmax(vs, 0.0);
        r.sdx = (float)(abx * abx * vs);
        r.sdy = (float)(aby * aby * vs);
        r.sdz = (float)(abz * abz * vs);
        r.mean_dist = (float)md;
        r.sd_dist = (float)fmax(vd, 0.0);
        r.rgb = 0;
        if (t.color) {  // EXTENSION: mean colour of the cylinder members, round half up
            const uint32_t cr = (uint32_t)((2 * s[SW_R] + cnt) / (2 * cnt));  // exact integer form of floor(sum/cnt + 1/2)
            const uint32_t cg = (uint32_t)((2 * s[SW_G] + cnt) / (2 * cnt));
            const uint32_t cb = (uint32_t)((2 * s[SW_B] + cnt) / (2 * cnt));
            r.rgb = (min(cr, 255u) << 16) | (min(cg, 255u) << 8) | min(cb, 255u);
        }
    }
    if (opt.paint_white) r.rgb = 0x00FFFFFFu;
    if (opt.classify_threshold >= 0) r.rgb = (int)r.count > opt.classify_threshold ? 0x00FF0000u : 0x00FFFFFFu;
    rows[j] = r;
}


// ---- multi-GPU epoch exchange ----------------------------------------------------------
// Frames shard across ranks; what must be agreed before a clean pass is the occupancy set and, per cell, the
// smallest frame id that touched it (the viewpoint latch).  Each rank exports the cells IT occupied since
// the last exchange and imports everybody else's; normals and registrations are then computed redundantly
// and deterministically on every rank, while buffers and statistic sums stay private partial state.
// One 16-byte record per newly occupied cell (key, smallest frame id that touched it) and TWO per frame the rank has integrated since
// the last exchange (the frame's viewpoint: ranks only know their own frames' poses, and a cell's latch may name anybody's frame).
// Until round 4 every cell record carried its frame's viewpoint: 32 bytes a cell, 338 MB received per rank in the first epoch of
// eight cameras (profiles/r04_virtual_ranks.md); the viewpoints now travel once per frame.
struct __attribute__((aligned(16))) EpochRec {
    uint64_t key;  // cell key (below 2^63), or kEpochFrame | half << 62 | frame id
    union {
        struct {
            uint32_t first_frame, pad;
        };
        struct {
            float a, b;  // half 0: viewpoint x, y; half 1: viewpoint z, 0
        };
    };
};
static_assert(sizeof(EpochRec) == 16, "EpochRec is 16 bytes");
constexpr uint64_t kEpochFrame = 1ull << 63, kEpochFrameHalf = 1ull << 62;

// Records [0, n_cells): the cells occ_list[first .. first + n_cells); behind them two records per frame of frame_list[frame_first .. + n_frames).
__global__ __launch_bounds__(256) void k_epoch_export(const GridParams g, const Tables t, const uint64_t first, const uint64_t n_cells, const uint64_t frame_first,
                                                      const uint64_t n_frames, EpochRec* __restrict__ out)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cells + 2 * n_frames) return;
    EpochRec r;
    if (j < n_cells) {
        const uint32_t slot = t.occ_list[first + j];
        int32_t x, y, z;
        slot_coords(g, t, slot, x, y, z);
        r.key = make_key(g, x, y, z);
        r.first_frame = t.first_frame[slot];
        r.pad = 0;
    } else {
        const uint64_t k = j - n_cells;
        const uint32_t fid = t.frame_list[frame_first + (k >> 1)];
        const uint32_t half = (uint32_t)(k & 1u);
        r.key = kEpochFrame | (half ? kEpochFrameHalf : 0ull) | fid;
        r.a = t.frame_vp[3 * (uint64_t)fid + 2 * half];
        r.b = half ? 0.f : t.frame_vp[3 * (uint64_t)fid + 1];
    }
    out[j] = r;
}

__global__ __launch_bounds__(256) void k_epoch_import(const GridParams g, const Tables t, const EpochRec* __restrict__ in, const uint64_t n)
{
    // kImportTiles tiles per workgroup, one occ_list reservation (hot counter: see k_register)
    uint32_t slot_[kImportTiles];
    uint32_t f_first = 0;
#pragma unroll
    for (int tt = 0; tt < kImportTiles; tt++) {
        const uint64_t j = ((uint64_t)blockIdx.x * kImportTiles + tt) * 256u + threadIdx.x;
        bool want = j < n;
        EpochRec r;
        r.key = 0;
        r.first_frame = kNoFrame;
        r.pad = 0;
        int32_t x = 0, y = 0, z = 0;
        if (want) {
            r = in[j];
            if (r.key & kEpochFrame) {  // a frame's viewpoint (the same value from every exporter)
                const uint32_t fid = (uint32_t)r.key;
                if (fid < t.max_frames) {
                    if (r.key & kEpochFrameHalf) {
                        t.frame_vp[3 * (uint64_t)fid + 2] = r.a;
                    } else {
                        t.frame_vp[3 * (uint64_t)fid] = r.a;
                        t.frame_vp[3 * (uint64_t)fid + 1] = r.b;
                    }
                }
                want = false;
            } else {
                key_coords(g, r.key, x, y, z);
                want = x <= g.dim[0] && y <= g.dim[1] && z <= g.dim[2];  // storage extent is dim+1 (grid.hpp:626)
            }
        }
        const uint32_t bidx = want ? brick_index(g, x, y, z) : 0u;
        const uint32_t b = brick_acquire_wave(t, bidx, want);
        want = want && b != 0;
        const uint32_t slot = b * kBrickCells + local_index(x, y, z);
        slot_[tt] = slot;
        if (want) {
            const unsigned long long bit = 1ull << (((y & 7) << 3) | (z & 7));
            const unsigned long long old = atomicOr(reinterpret_cast<unsigned long long*>(&t.occ_mask[(uint64_t)b * 8 + (x & 7)]), bit);
            if (!(old & bit)) f_first |= 1u << tt;  // (joins occ_list: the clean pass this import opens files its pre-dependant, k_clean_begin)
            if (r.first_frame < t.max_frames) atomicMin(&t.first_frame[slot], r.first_frame);
        }
    }
    __shared__ BlockReserveScratch brs;
    unsigned long long oi = block_reserve_n(&t.ctr[C_OCC], (uint32_t)__popc(f_first), brs);
    bool overflow = false;
#pragma unroll
    for (int tt = 0; tt < kImportTiles; tt++) {
        if (f_first & (1u << tt)) {
            if (oi < t.max_occ) t.occ_list[oi] = slot_[tt];
            else overflow = true;
            oi++;
        }
    }
    if (overflow) atomicOr(&t.ctr[C_ERR], (unsigned long long)E_OCC);
}

__global__ __launch_bounds__(256) void k_add_u64(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, const uint64_t n)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] += src[j];
}

// Diagnostic: keys of every occupied cell.
__global__ __launch_bounds__(256) void k_occupied_keys(const GridParams g, const Tables t, const uint64_t n_occ, uint64_t* __restrict__ keys)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_occ) return;
    int32_t x, y, z;
    slot_coords(g, t, t.occ_list[j], x, y, z);
    keys[j] = make_key(g, x, y, z);
}

// ---- leaf probes (tests only; same device functions as the kernels above) -------------------------
__global__ void k_probe_points(const GridParams g, const double* __restrict__ pose, const float* __restrict__ xyz, const uint64_t n,
                               float* __restrict__ q_out, int32_t* __restrict__ idx_out, uint8_t* __restrict__ flags_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double T[12];
    for (int k = 0; k < 12; k++) T[k] = pose[k];
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const F3 q = transform_point(T, x, y, z);
    int32_t ix, iy, iz;
    voxel_coords(g, q, ix, iy, iz);
    q_out[3 * i] = q.x;
    q_out[3 * i + 1] = q.y;
    q_out[3 * i + 2] = q.z;
    idx_out[3 * i] = ix;
    idx_out[3 * i + 1] = iy;
    idx_out[3 * i + 2] = iz;
    flags_out[i] = (zclip_pass(g, z) ? 1 : 0) | (valid_point(g, q) ? 2 : 0);
}

__global__ void k_probe_normals(const GridParams g, const uint64_t n, const int32_t* __restrict__ cells, const uint8_t* __restrict__ occ,
                                const float* __restrict__ vps, float* __restrict__ normals_out, int32_t* __restrict__ totals_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t x = cells[3 * i], y = cells[3 * i + 1], z = cells[3 * i + 2];
    Moments m;
    m.clear();
    int total = 0;
    for (int d = 0; d < 125; d++) {
        const int a = d / 25 - 2, b = (d / 5) % 5 - 2, c = d % 5 - 2;
        if (!occ[125 * i + d] || !valid_coord(g, x + a, y + b, z + c)) continue;
        m.add(voxel_center(g, x + a, y + b, z + c), g.cov_shifted != 0);
        total++;
    }
    totals_out[i] = total;
    F3 nrm = {0, 0, 0};
    if (total >= 3) {
        nrm = m.normal(total);
        nrm = orient_normal(nrm, F3{vps[3 * i], vps[3 * i + 1], vps[3 * i + 2]}, voxel_center(g, x, y, z));
    }
    normals_out[3 * i] = nrm.x;
    normals_out[3 * i + 1] = nrm.y;
    normals_out[3 * i + 2] = nrm.z;
}

__global__ void k_probe_project(const GridParams g, const uint64_t n, const float* __restrict__ pts, const float* __restrict__ centres,
                                const float* __restrict__ normals, float* __restrict__ proj_out, double* __restrict__ dist_out,
                                uint8_t* __restrict__ member_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 proj;
    double dist;
    const bool mem = cylinder_member(g, F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, F3{centres[3 * i], centres[3 * i + 1], centres[3 * i + 2]},
                                     F3{normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]}, proj, dist);
    proj_out[3 * i] = proj.x;
    proj_out[3 * i + 1] = proj.y;
    proj_out[3 * i + 2] = proj.z;
    dist_out[i] = dist;
    F3 a, ab;
    float dd, sp, distf;
    line_of(g, F3{centres[3 * i], centres[3 * i + 1], centres[3 * i + 2]}, F3{normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]}, a, ab, dd);
    const bool mem2 = line_member(g, F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, a, ab, dd, sp, distf);  // the kernels' form
    float sp3, distf3;  // the form with the hoisted division (k_update_cells): must agree bit for bit
    const bool mem3 = line_member_hoisted(g, F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, a, ab, line_div_of(dd), sp3, distf3);
    const bool same = mem3 == mem2 && __float_as_uint(sp3) == __float_as_uint(sp) && __float_as_uint(distf3) == __float_as_uint(distf);
    member_out[i] = (mem ? 1 : 0) | (mem2 ? 2 : 0) | (same ? 4 : 0);
}

__global__ void k_probe_trig(const uint64_t n, const float* __restrict__ y, const float* __restrict__ x, float* __restrict__ a_out,
                             float* __restrict__ c_out, float* __restrict__ s_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a_out[i] = det_atan2f(y[i], x[i]);
    c_out[i] = det_cosf(x[i]);
    s_out[i] = det_sinf(x[i]);
}

}  // namespace hfpf
