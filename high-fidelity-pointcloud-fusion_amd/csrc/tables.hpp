// csrc/tables.hpp -- HBM layout of the occupancy grid and the device-side lookup / allocation helpers.
//
// Layout (ours; the reference keeps vector<vector<vector<Voxel>>> + heap VoxelInfo, grid.hpp:51-82,108):
//
//   dir[bdim.x*bdim.y*bdim.z]  u32   brick directory, direct-mapped over the bounded bbox (a perfect
//                                    hash: the bbox is a required launch parameter, node.cpp:451).
//                                    0 = untouched, kLock = being allocated, else brick id (1-based).
//   brick b, cell (lx,ly,lz)         slot = b*512 + morton3(lx,ly,lz) (2x2x2-blocked); brick 0 is a permanent
//                                    all-zero "null brick" so a lookup of an untouched region needs no branch.
//   per slot:  info u64              bit0 unused (Voxel::occupied lives in occ_mask), bit1 normal_found,
//                                    bits 2..17 dependant count, bits 18..63 offset into dep[]
//              first_frame u32       smallest frame id that touched the cell (-> VoxelInfo::viewpoint, grid.hpp:229,238)
//              buf_head u32[4]       heads of the cell's 4 interleaved chains in the point log (VoxelInfo::buffer)
//              stat_id u32           1-based id of the cell's normal/statistics record, 0 = none
//              pre_dep u32           the one dependant registered while the cell was unoccupied
//                                    (grid.hpp:443-449 overwrite semantics: last registrant wins); it becomes the cell's
//                                    dependant list at the first clean pass after the cell was occupied (kernels.hpp k_materialize_new)
//              dep_tmp u32           scratch for the dependant-table rebuild
//   per brick: occ_mask[8] u64       occupancy bits, one u64 per x-plane (bit = ly*8+lz): the 5x5x5 stencil
//                                    of grid.hpp:334-349 becomes <= 20 u64 loads.
//              nd_mask[8][2] u64     normal_found bits and has-dependants bits, same bit order: one 16-byte read tells
//                                    k_integrate what to do with a point (22 k bricks x 128 B stay in L2, where the
//                                    8-byte info words of 10^7 cells do not).
//   point log: log_pt float4 (x,y,z, w = slot until linked, then `next`) (+ log_rgb u32 with colour fusion): the
//              reference's per-voxel buffers (grid.hpp:70,211,230) as one append-only array.
//   normals:   nv_key u64, nv_slot u32, nv_c/nv_n float3, stats[8] i64 per record (stats.hpp)
//   dependants: reg_occ (slot, stat id) pairs, append-only; dep[] = 32-byte entries grouped per slot,
//              rebuilt by every clean pass (VoxelInfo::dependants, grid.hpp:71,417,447).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geometry.hpp"

namespace hfpf {

constexpr uint32_t kLock = 0xFFFFFFFFu;
constexpr int kBrickShift = 3;
constexpr int kBrickCells = 512;
constexpr int kMaxSpin = 1 << 20;
constexpr uint32_t kNoFrame = 0xFFFFFFFFu;

// info word (bit 0 is unused: occupancy lives in the brick's occ_mask alone, one returning atomic per first touch)
constexpr uint64_t kNormal = 2ull;
// dep_tmp (per slot, scratch of the dependant-table update): during the incremental fill  old length << 16 | append cursor;
// afterwards, until the replay has taken note,  kTouchedMark | old length  (the entries behind `old length` are the registrants
// of the running pass); kDepPoison marks a cell the update had to give up on.  Zero between passes.
constexpr uint32_t kTouchedMark = 0x40000000u, kDepPoison = 0x80000000u, kDepOldMax = 0x7FFFu;
constexpr int kDepCntShift = 2;
constexpr uint64_t kDepCntMask = 0xFFFFull;
constexpr int kDepOffShift = 18;

// counters (device u64 array)
enum Ctr : int {
    C_BRICKS = 0,   // bricks allocated (ids 1..n)
    C_LOG,          // point-log entries (1-based indices 1..n)
    C_OCC,          // entries in occ_list
    C_NORMALS,      // normal records (ids 1..n)
    C_REG,          // entries in reg_occ
    C_DEP,          // entries in dep[]
    C_PREREG,       // entries in prereg_list: occupied cells whose pre-dependant was filed as their list (kernels.hpp k_materialize_new)
    C_TOUCHED,      // entries in touched_list (rebuild scratch)
    C_CAND,         // candidates of the running clean pass
    C_ERR,          // error bits
    C_PRESENTED,
    C_ZPASS,
    C_INBOX,
    C_BUFFERED,
    C_DEP_TESTED,
    C_DEP_MEMBER,
    C_ROWS,         // rows valid at extract
    C_REPLAY_MEMBER, // buffered points that fell inside a cylinder during clean-time replay
    C_PEND,         // occupied cells still without a normal after the running gate pass
    C_UNUSED19,     // (until round 4: unoccupied cells whose single dependant changed in the running clean pass)
    C_TOUCHED_SINGLE, // host mirror only (sum of word 4 of the striped lines): touched cells of the running pass in single-run bricks
    C_TABLE_MISS,   // work items of k_update_cells that found no slot in the LDS record table (they update HBM directly)
    C_OVF,          // entries in the overflow list of the running integrate launch (k_integrate -> k_integrate_overflow)
    C_FRAMES,       // entries in frame_list (frames this handle has integrated, in any order)
    C_UPD_ROUNDS,   // sort rounds beyond the first that bricks of k_update_cells took (diagnostic: which bricks exceed one LDS round)
    C_COUNT = 32
};

enum ErrBits : uint64_t {
    E_BRICKS = 1,
    E_LOG = 2,
    E_OCC = 4,
    E_NORMALS = 8,
    E_REG = 16,
    E_DEP = 32,
    E_SPIN = 64,
    E_DEPCNT = 128,
    E_FRAME = 256,
    E_OVF = 512,
    E_CHAIN = 1024,  // a point-log link or run record that names no log entry (internal consistency; the readers stop instead of reading)
};

struct __attribute__((aligned(32))) DepEntry {  // one dependant of a cell, denormalised for the per-point loop
    uint32_t sid;                               // statistics record to update
    float ax, ay, az;                           // a = centre - r*n: one end of the voxel's line segment (geometry.hpp line_of)
    float abx, aby, abz;                        // ab = a - b
    float dd;                                   // |ab|^2
};
static_assert(sizeof(DepEntry) == 32, "DepEntry must be 32 bytes");

constexpr int kStatWords = 8;  // i64 words per statistics record (64 B = one atomic segment): see stats.hpp

struct Tables {
    uint32_t* dir;
    uint32_t* brick_lin;  // [max_bricks+1] linear directory index of brick id
    uint64_t* info;
    uint32_t* first_frame;
    uint32_t* buf_head;
    uint32_t* stat_id;
    uint32_t* pre_dep;
    uint32_t* dep_tmp;
    uint64_t* occ_mask;
    float4* log_pt;
    uint32_t* log_rgb;  // NULL unless HFPF_FLAG_FUSE_COLOR
    uint32_t* occ_list;
    uint64_t* nv_key;
    uint32_t* nv_slot;
    float* nv_c;  // 3 per record
    float* nv_n;  // 3 per record
    float4* nv_line;  // 2 per record = one DepEntry: (record id, a.xyz), (ab.xyz, |ab|^2): the per-voxel half of the cylinder test
    unsigned long long* stats;
    uint32_t color;   // 1 with HFPF_FLAG_FUSE_COLOR: words 5-7 of a statistics record carry the members' colour sums
    uint32_t test_table_skip;  // tests only (HFPF_TEST_TABLE_SKIP=1): records with an odd id bypass the LDS record tables, as if they were full
    uint64_t* nd_mask;  // per brick and x-plane, 2 words: normal_found bits, has-dependants bits of the plane's 64 cells
    uint2* reg_occ;
    DepEntry* dep;
    // dep[] and nv_line (one 32-byte entry per normal record, same layout; allocated behind dep[]) addressed as ONE array of entries:
    // *_first = where each starts in it (in entries).  k_update_cells keeps a 40-bit entry index per cell, so the list of a cell --
    // entries of dep[], or the record line of its pre-dependant -- is read the same way.
    const DepEntry* ent_base;
    uint64_t ent_dep_first, ent_nv_first;
    uint32_t* prereg_list;
    uint32_t* touched_list;
    // per brick: the point-log run k_buffer appended last (first entry, length) and how many appends the brick has seen (+ 0x100 for
    // every entry that reached the log outside a run).  A brick with exactly ONE run holds all its buffered points contiguously:
    // its replay streams that run (k_update_cells in replay mode) instead of walking the cells' chains.
    uint32_t* run_start;
    uint32_t* run_len;
    uint32_t* run_cnt;
    uint64_t* cand_key;    // clean scratch (unsorted / sorted ping-pong handled by the host)
    float* frame_vp;       // 3 per frame id
    uint32_t* frame_list;  // ids of the frames this handle has integrated (C_FRAMES entries): the epoch exchange sends their viewpoints
    unsigned long long* ctr;
    // brick bins of the two-pass dependant update (kernels.hpp, k_integrate<BIN> + k_update)
    float4* bin_pt;       // (x, y, z, bits of the cell's index inside the brick) of points parked for k_update, grouped per brick
    uint32_t* bin_rgb;    // their colour (HFPF_FLAG_FUSE_COLOR only)
    float4* ovf_pt;       // overflow list of the running integrate launch: (x, y, z, bits of the slot) of the points that found no room in a bin
    uint2* ovf_aux;       // ... and (frame id | kOvfHasNormal | kOvfHasDeps, colour)
    uint64_t ovf_cap;     // entries both hold (= the points of the launch)
    // Two regions per brick, index 2*brick + kind: kind 0 = points whose cell has a normal (dependant updates only),
    // kind 1 = points whose cell has none yet (to be buffered by k_buffer, and updated like the others by k_update)
    uint32_t* bin_fill;   // per region: entries requested in the running launch (may exceed the region)
    uint32_t* bin_off;    // per region: first entry
    uint32_t* bin_capb;   // per region: entries it can hold (0 = not planned: direct forms in k_integrate)
    unsigned long long* log_ctr;  // kLogRegions append counters, one per 128-byte line (index r*16)
    uint64_t log_region_cap;      // entries per log region
    uint64_t max_bricks, max_log, max_occ, max_normals, max_reg, max_dep, max_frames;
};

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

HFPF_HD uint64_t make_key(const GridParams& g, int32_t x, int32_t y, int32_t z)
{
    return ((uint64_t)x << g.key_sx) | ((uint64_t)y << g.key_sy) | (uint64_t)z;
}
HFPF_HD void key_coords(const GridParams& g, uint64_t k, int32_t& x, int32_t& y, int32_t& z)
{
    x = (int32_t)(k >> g.key_sx);
    y = (int32_t)((k >> g.key_sy) & ((1ull << (g.key_sx - g.key_sy)) - 1ull));
    z = (int32_t)(k & ((1ull << g.key_sy) - 1ull));
}
// Z-order (Morton) code of a cell: the order the candidates of a clean pass are numbered in (HFPF_MORTON_IDS).  Records of one brick
// -- 8 x 8 x 8 cells = the low 9 bits of the code -- get consecutive ids, so whatever reads or updates records by id (the per-brick
// flush of k_update_cells, the registration walk, the dependant-table fill) touches neighbouring lines.  21 bits per axis.
#ifndef HFPF_MORTON_IDS
#define HFPF_MORTON_IDS 1
#endif
HFPF_HD uint64_t morton_spread(uint32_t v)
{
    uint64_t x = v & 0x1FFFFFull;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
HFPF_HD uint32_t morton_compact(uint64_t x)
{
    x &= 0x1249249249249249ull;
    x = (x | (x >> 2)) & 0x10C30C30C30C30C3ull;
    x = (x | (x >> 4)) & 0x100F00F00F00F00Full;
    x = (x | (x >> 8)) & 0x1F0000FF0000FFull;
    x = (x | (x >> 16)) & 0x1F00000000FFFFull;
    x = (x | (x >> 32)) & 0x1FFFFFull;
    return (uint32_t)x;
}
HFPF_HD uint64_t morton_key(int32_t x, int32_t y, int32_t z) { return (morton_spread((uint32_t)x) << 2) | (morton_spread((uint32_t)y) << 1) | morton_spread((uint32_t)z); }
HFPF_HD void morton_coords(uint64_t k, int32_t& x, int32_t& y, int32_t& z)
{
    x = (int32_t)morton_compact(k >> 2);
    y = (int32_t)morton_compact(k >> 1);
    z = (int32_t)morton_compact(k);
}
HFPF_HD uint32_t brick_index(const GridParams& g, int32_t x, int32_t y, int32_t z)
{
    return ((uint32_t)(x >> kBrickShift) * (uint32_t)g.bdim[1] + (uint32_t)(y >> kBrickShift)) * (uint32_t)g.bdim[2] +
           (uint32_t)(z >> kBrickShift);
}
// Cell order inside a brick: 2x2x2-blocked (3-level Morton), so the 8 cells of one 64-byte line of a per-slot u64 array
// form a 2x2x2 cube and a surface patch crossing the brick touches about half as many lines as with z-runs.
HFPF_HD uint32_t spread3(uint32_t v)  // 3 bits abc -> a00b00c
{
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4);
}
HFPF_HD uint32_t compact3(uint32_t v)  // inverse of spread3 on bits 0,3,6
{
    return (v & 1u) | ((v >> 2) & 2u) | ((v >> 4) & 4u);
}
HFPF_HD uint32_t spread3_lut(uint32_t v)  // the same for v in 0..7: byte v of a packed table (two instructions)
{
    return (uint32_t)(0x4948414009080100ull >> (v * 8u)) & 0xFFu;
}
HFPF_HD uint32_t local_index(int32_t x, int32_t y, int32_t z)
{
    return (spread3_lut((uint32_t)x & 7u) << 2) | (spread3_lut((uint32_t)y & 7u) << 1) | spread3_lut((uint32_t)z & 7u);
}

// Cell coordinates of a slot (inverse of the two functions above).
__device__ __forceinline__ void slot_coords(const GridParams& g, const Tables& t, uint32_t slot, int32_t& x, int32_t& y, int32_t& z)
{
    const uint32_t b = slot >> 9, l = slot & 511;
    const uint32_t lin = t.brick_lin[b];
    const uint32_t bz = lin % (uint32_t)g.bdim[2];
    const uint32_t r = lin / (uint32_t)g.bdim[2];
    const uint32_t by = r % (uint32_t)g.bdim[1];
    const uint32_t bx = r / (uint32_t)g.bdim[1];
    x = (int32_t)(bx * 8 + compact3(l >> 2));
    y = (int32_t)(by * 8 + compact3(l >> 1));
    z = (int32_t)(bz * 8 + compact3(l));
}

// Word index into occ_mask (and /2 of nd_mask) and bit of a slot's cell.
__device__ __forceinline__ void slot_plane_bit(uint32_t slot, uint64_t& plane, uint64_t& bit)
{
    const uint32_t l = slot & 511u;
    plane = (uint64_t)(slot >> 9) * 8u + compact3(l >> 2);
    bit = 1ull << ((compact3(l >> 1) << 3) | compact3(l));
}

// Read-only lookup: slot of a cell, inside the null brick (all zeros) when the brick was never touched.
__device__ __forceinline__ uint32_t slot_lookup(const GridParams& g, const Tables& t, int32_t x, int32_t y, int32_t z)
{
    uint32_t b = t.dir[brick_index(g, x, y, z)];
    if (b == kLock) b = 0;  // cannot happen between kernels; defensive
    return b * kBrickCells + local_index(x, y, z);
}

// One lane claims (or finds) the brick of directory entry `bidx`.  Lock-free for readers; a writer
// that loses the CAS waits for the winner's id, which another *wave* is guaranteed to publish (the
// wave-level election below makes sure two lanes of one wave never wait on each other).  Bounded.
__device__ inline uint32_t brick_acquire_single(const Tables& t, uint32_t bidx)
{
    uint32_t* p = &t.dir[bidx];
    for (int spin = 0; spin < kMaxSpin; ++spin) {
        uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == 0) {
            const uint32_t old = atomicCAS(p, 0u, kLock);
            if (old == 0) {
                const unsigned long long id = atomicAdd(&t.ctr[C_BRICKS], 1ull) + 1ull;
                if (id > t.max_bricks) {
                    atomicOr(&t.ctr[C_ERR], (unsigned long long)E_BRICKS);
                    __hip_atomic_store(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return 0;
                }
                t.brick_lin[id] = bidx;
                __hip_atomic_store(p, (uint32_t)id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                return (uint32_t)id;
            }
            v = old;
        }
        if (v != kLock) return v;
        __builtin_amdgcn_s_sleep(4);
    }
    atomicOr(&t.ctr[C_ERR], (unsigned long long)E_SPIN);
    return 0;
}

// The claims of one wave, side by side: `leader` lanes hold DISTINCT directory entries.  Three phases in program order, so that
// every lane that won an entry has published its id before any lane of the wave starts to wait for another wave's:
//   1. every leader reads its entry and tries the CAS 0 -> kLock
//   2. the winners take their ids with ONE atomic on the brick counter for the wave (a claim used to take its own: the counter is
//      one address for the whole chip, ~12 ns a hit, and a dry run or a session's first launch claims ~10,000 bricks -- more than
//      a tenth of a millisecond of nothing but that counter) and publish them
//   3. a leader that found its entry locked by another wave waits for the id (bounded).
// Convergent: all 64 lanes call it; returns the id for leader lanes (0 on failure), 0 for the others.
__device__ inline uint32_t brick_claim_leaders(const Tables& t, uint32_t bidx, bool leader)
{
    const uint32_t lane = lane_id();
    uint32_t* p = &t.dir[leader ? bidx : 0u];
    uint32_t v = 0;
    bool won = false;
    if (leader) {
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == 0) {
            const uint32_t old = atomicCAS(p, 0u, kLock);
            won = old == 0;
            v = won ? kLock : old;
        }
    }
    const unsigned long long wm = __ballot(won);
    if (wm) {
        const int first = __ffsll((long long)wm) - 1;
        unsigned long long base = 0;
        if (lane == (uint32_t)first) base = atomicAdd(&t.ctr[C_BRICKS], (unsigned long long)__popcll(wm));
        base = __shfl(base, first);
        if (won) {
            const unsigned long long id = base + (unsigned long long)__popcll(wm & ((1ull << lane) - 1ull)) + 1ull;
            if (id > t.max_bricks) {
                atomicOr(&t.ctr[C_ERR], (unsigned long long)E_BRICKS);
                __hip_atomic_store(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = 0;
            } else {
                t.brick_lin[id] = bidx;
                __hip_atomic_store(p, (uint32_t)id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                v = (uint32_t)id;
            }
        }
    }
    if (leader && v == kLock) v = brick_acquire_single(t, bidx);  // another wave holds the entry: its id is on its way
    return leader ? v : 0u;
}

// Wave-level election: every lane that needs a brick which is not there yet takes part; one lane per
// distinct directory entry runs the claim, the others receive its id through a shuffle.
// Must be reached by all 64 lanes of the wave (convergent); `want` masks the lanes with real work.
// `v` = the lane's directory word, read by the caller with a plain (cacheable) load (0 for lanes that do not want a brick); a
// stale 0 only sends the lane through the atomic path.
__device__ inline uint32_t brick_acquire_wave(const Tables& t, uint32_t bidx, bool want, uint32_t v)
{
    const bool need = want && (v == 0 || v == kLock);
    unsigned long long m = __ballot(need);
    if (m == 0) return v;
    // The lanes that need a brick are grouped by brick with registers and scalar lane reads only; then the first lane of every
    // group claims its brick, all of them side by side (a claim is a chain of three memory round trips: one after the other
    // they cost a wave ~3 us per new brick).  No lane waits on another lane of its own wave: the leaders hold distinct bricks.
    const uint32_t lane = lane_id();
    unsigned long long grp = 0;
    while (m) {
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)bidx, leader);
        const bool same = need && bidx == lb;
        const unsigned long long sm = __ballot(same);
        if (same) grp = sm;
        m &= ~sm;
    }
    const uint32_t leader_lane = need ? (uint32_t)(__ffsll((long long)grp) - 1) : lane;
    uint32_t id = v;
    const bool claims = need && leader_lane == lane;
    const uint32_t claimed = brick_claim_leaders(t, bidx, claims);
    if (claims) id = claimed;
    const uint32_t got = (uint32_t)__shfl((int)id, (int)leader_lane);
    return need ? got : v;
}

// The same when the caller already knows, per lane, the mask of the lanes that want the same brick (k_integrate groups its lanes
// by brick anyway): the first lane of every group claims its brick in ONE pass -- the leaders of different bricks run the claim
// side by side instead of one after the other -- and hands the id to its group.  No lane waits on another lane of its own wave
// (leaders hold distinct bricks), only on other waves, as before.
// rare_tables: the claim (a rare branch of the caller's loop) reads the table descriptor from the kernarg segment on the spot
// (kernels.hpp kernarg_tables) instead of keeping the caller's copy of its fields alive across the loop.
__device__ __forceinline__ const Tables& kernarg_tables();
__device__ inline uint32_t brick_acquire_groups(const Tables& t_in, uint32_t bidx, bool want, uint32_t v, unsigned long long same_brick, bool rare_tables = false)
{
    const bool need = want && (v == 0 || v == kLock);
    if (__ballot(need) == 0) return v;
    const Tables& t = rare_tables ? kernarg_tables() : t_in;
    const uint32_t lane = lane_id();
    const uint32_t leader = want ? (uint32_t)(__ffsll((long long)same_brick) - 1) : lane;
    // a group's lanes read the same directory word, but not necessarily the same value (another wave may publish in between):
    // the leader claims if ANY lane of its group still needs the brick
    const unsigned long long need_mask = __ballot(need);
    uint32_t id = v;
    const bool claims = want && leader == lane && (need_mask & same_brick) != 0;
    const uint32_t claimed = brick_claim_leaders(t, bidx, claims);
    if (claims) id = claimed;
    const uint32_t got = (uint32_t)__shfl((int)id, (int)leader);
    return (want && (need_mask & same_brick) != 0) ? got : v;
}

__device__ inline uint32_t brick_acquire_wave(const Tables& t, uint32_t bidx, bool want)
{
    uint32_t v = 0;
    if (want) v = t.dir[bidx];
    return brick_acquire_wave(t, bidx, want, v);
}

// Wave-aggregated bump allocation: one atomic per wave, ranks by prefix popcount of the ballot.
// Returns the 0-based index reserved for this lane (meaningless when !want). Convergent.
__device__ inline unsigned long long wave_reserve(unsigned long long* ctr, bool want)
{
    const unsigned long long m = __ballot(want);
    if (m == 0) return 0;
    const uint32_t lane = lane_id();
    const int leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if (lane == (uint32_t)leader) base = atomicAdd(ctr, (unsigned long long)__popcll(m));
    base = __shfl(base, leader);
    return base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
}

// Block-aggregated bump allocation for 256-thread blocks: ONE device atomic per block (a same-address device atomic
// costs ~12 ns at the memory side, so per-wave reservations from millions of threads serialise on the counter).
// Every thread of the block must call it (it synchronises); scratch = 5 words of LDS reused across calls.
struct BlockReserveScratch {
    uint32_t wave_cnt[4];
    unsigned long long base;
};
__device__ inline unsigned long long block_reserve(unsigned long long* ctr, bool want, BlockReserveScratch& s)
{
    const unsigned long long m = __ballot(want);
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    if (lane == 0) s.wave_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = s.wave_cnt[0] + s.wave_cnt[1] + s.wave_cnt[2] + s.wave_cnt[3];
        s.base = total ? atomicAdd(ctr, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    uint32_t prefix = 0;
    for (uint32_t w = 0; w < wave; w++) prefix += s.wave_cnt[w];
    const unsigned long long r = s.base + prefix + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();  // scratch may be reused by the next call
    return r;
}

// Block-aggregated reservation of a per-thread amount n: ONE device atomic per workgroup of 256 threads; returns the first
// index of the calling thread's n entries.  Convergent (contains barriers).
__device__ inline unsigned long long block_reserve_n(unsigned long long* ctr, uint32_t n, BlockReserveScratch& s)
{
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == 63) s.wave_cnt[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = s.wave_cnt[0] + s.wave_cnt[1] + s.wave_cnt[2] + s.wave_cnt[3];
        s.base = total ? atomicAdd(ctr, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    uint32_t prefix = 0;
    for (uint32_t w = 0; w < wave; w++) prefix += s.wave_cnt[w];
    const unsigned long long r = s.base + prefix + (unsigned long long)(incl - n);
    __syncthreads();  // scratch may be reused by the next call
    return r;
}

// The same for T tiles per thread, handing the indices out in tile-major order (tile, then wave, then lane) -- the order T
// separate workgroups would have produced -- so neighbouring items stay neighbours in the list.  n[tt] = amount of the
// calling thread for tile tt; idx[tt] = first index of that amount.  ONE device atomic.  Convergent (contains barriers).
template <int T>
struct TileReserveScratch {
    uint32_t cnt[T][4];
    unsigned long long base;
};
template <int T>
__device__ inline void block_reserve_tiles(unsigned long long* ctr, const uint32_t (&n)[T], unsigned long long (&idx)[T], TileReserveScratch<T>& s)
{
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t excl[T];
#pragma unroll
    for (int tt = 0; tt < T; tt++) {
        uint32_t incl = n[tt];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += v;
        }
        excl[tt] = incl - n[tt];
        if (lane == 63) s.cnt[tt][wave] = incl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int tt = 0; tt < T; tt++) total += s.cnt[tt][0] + s.cnt[tt][1] + s.cnt[tt][2] + s.cnt[tt][3];
        s.base = total ? atomicAdd(ctr, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long run = s.base;
#pragma unroll
    for (int tt = 0; tt < T; tt++) {
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; w++) before += s.cnt[tt][w];
        idx[tt] = run + before + excl[tt];
        run += s.cnt[tt][0] + s.cnt[tt][1] + s.cnt[tt][2] + s.cnt[tt][3];
    }
    __syncthreads();  // scratch may be reused by the next call
}

// Wave-aggregated reservation of a per-lane amount n (0 for idle lanes): one atomic per wave, exclusive
// prefix sum across the lanes.  Convergent.
__device__ inline unsigned long long wave_reserve_n(unsigned long long* ctr, uint32_t n)
{
    const uint32_t lane = lane_id();
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += v;
    }
    const uint32_t total = __shfl(incl, 63);
    unsigned long long base = 0;
    if (lane == 63 && total) base = atomicAdd(ctr, (unsigned long long)total);
    base = __shfl(base, 63);
    return base + (unsigned long long)(incl - n);
}

// Wave-reduced counter add (diagnostic counters): convergent.
__device__ inline void wave_count(unsigned long long* ctr, uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane_id() == 0 && v) atomicAdd(ctr, (unsigned long long)v);
}

}  // namespace hfpf
