// bf_kernels.h -- launch interface between the C-ABI files (bf_context / bf_upload / bf_operators / bf_run / bf_extras .cpp) and
// bf_kernels.hip (gfx950 kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "bf_device.h"

namespace bf {

constexpr int kHistCopies = 8;   // copies of the re-bin's global histogram (work-group b adds to copy b % 8: contention)
constexpr int kBinTileLdsMax = 156 * 1024;   // dynamic LDS of a scatter work-group (160 KiB per CU minus its static part)
constexpr int kPrepBlocks = 1024;   // work-groups of k_prepare == SliceStats records it writes

// min / max / sum statistics of one staged slice (k_prepare), one record per work-group.
struct SliceStats {
    int32_t xmin, xmax, ymin, ymax, tmin, tmax;
    long long tsum;
};

struct WarpScatterArgs {
    const uint32_t* xy;
    const int32_t* t;
    float2* p;
    const uint8_t* noise;          // may be NULL
    double2* nxny;                 // written only by the final warp, at [perm[i]]
    double2* uv;                   // optional: Event::compute_uv fused into the final warp
    const uint32_t* perm;          // original index of slot i (NULL: identity)
    unsigned long long* plane;     // point-scatter accumulator (current buffer)
    uint32_t* cplane;              // SPLIT mode count plane (current buffer)
    const DevState* st;
    long long n;
    int check_done;                // 1 inside the fused loop: return at once if st->done; 2: run only if done
    EvSets sets;                   // pick_set: take xy / t / p / perm from sets.s[hot.cs ^ hot.flip] instead
    int pick_set;
    int sorted_out;                // write nxny / uv at the slot index (coalesced) instead of perm[slot]
    bool packed;
};

struct StencilArgs {
    const DevState* st;
    int check_done;
    int R, C, scale, tbits;
    long long tmin;
    const unsigned long long* plane;   // SRC 0 / 1
    const uint32_t* cplane;            // SRC 1
    const float* time_in;              // SRC 2
    float* time_out;                   // optional
    uint32_t* count_out;               // optional
    float* gx_out;                     // optional (gy_out must be set with it)
    float* gy_out;
    MomentAcc* acc;                    // optional: moment sums are added to these exact accumulators
    MomentAcc* acc_zero;               // tile-binned loop: the other parity's accumulators, cleared here
    const uint32_t* ovf_cur;           // tile-binned loop: overflow events of this / the previous iteration, and the
    const uint32_t* ovf_prev;          //   next iteration's counter (cleared here)
    uint32_t* ovf_next;
    unsigned long long* zero_plane;    // optional: the OTHER plane buffer, zeroed here
    uint32_t* zero_cplane;
    // tile-binned loop: one bit per image pixel that an overflow event of this iteration touched (set by the scatter kernel;
    // row pitch ovf_pitch words, column Y in word (Y >> 5) + 1), and the other buffer's bitmap, cleared with its planes
    const uint32_t* ovf_bits;
    uint32_t* zero_bits;
    int ovf_pitch;
    int zero_full;                     // clear every pixel of the other buffer, not only the flagged ones (its dirt may predate the bitmap: first launch of a run)
    // fused reduction + model / loop update by the last work-group (NULL ticket: accumulate only)
    unsigned int* ticket;
    DevState* st_rw;                   // where the updated state goes (tile-binned loop, update here: NOT the buffer `st` is read from)
    DevState* snap;                    // optional: pinned host copy of the updated state, polled by the host
    bf_trace_rec* trace;
    int update_mode;                   // 1: full iteration_step / run() update, 0: model only
    unsigned long long* tl;            // debug timeline (BF_TIMELINE builds), usually NULL
    int tl_launch;
    // SRC 3 (tile-binned): per-bin slabs + the overflow planes of buffer `cur`
    const unsigned long long* slabs;
    const uint16_t* cidx;              // compact lists (hot.fmt == 1): see BinScatterArgs
    const uint32_t* chdr;
    int compact;
    const unsigned long long* m_cur;   // interior + margin format (compact == 3): the margin plane of buffer `cur`
    BinGrid g;
    int cur;
};

// Profiling hook: when armed (by ProfScope, bf_ctx.h), the next launch of a loop kernel goes through
// hipExtLaunchKernelGGL with these events, which then carry the kernel's own begin / end timestamps (what
// rocprofv3 reports) instead of bracketing the launch with two extra barrier packets (~1.5 us more).
struct LaunchTimer {
    hipEvent_t start = nullptr, stop = nullptr;
    bool consumed = false;
};
LaunchTimer& launch_timer();   // thread local
// Plain launch, or (profiling armed) an extended launch whose events carry the kernel's own timestamps.
template <class K, class... A>
static inline void launch_timed(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args) {
    LaunchTimer& t = launch_timer();
    if (t.start && !t.consumed) {
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, t.start, t.stop, 0, args...);
        t.consumed = true;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
    }
}

void launch_set_state(DevState* st, const DevState& v, hipStream_t s);
void launch_project_dn(const uint32_t* xy, const int32_t* t, float2* p, double2* nxny, const uint32_t* perm, int have_n,
                       const DevState* st, long long n, hipStream_t s);
// the run's final warp with compute_uv fused (outputs in slot order): one event per thread
void launch_final_warp(const WarpScatterArgs& a, hipStream_t s);
void launch_warp_scatter(const WarpScatterArgs& a, bool warp, bool scatter, bool write_n,
                         hipStream_t s);
void launch_prepare(const int32_t* fr_x, const int32_t* fr_y, const int32_t* t_in, uint32_t* xy,
                    int32_t* t_out, float2* p, long long n, long long n_pad, SliceStats* stats,
                    hipStream_t s);
void launch_local_time(const unsigned long long* ts, unsigned long long t0, int32_t* t_out, long long n, hipStream_t s);
// (ts32: `ts` holds 32-bit words, the low halves of the timestamps)
void launch_local_time16(const unsigned long long* ts, bool ts32, const uint16_t* row, const uint16_t* col, unsigned long long t0,
                         int32_t* x_out, int32_t* y_out, int32_t* t_out, long long n, hipStream_t s);
void stencil_grid(int R, int C, int* gx, int* gy);
void launch_stencil(const StencilArgs& a, int src, hipStream_t s, int n_cus = 0);   // n_cus: lets the tile-binned form pick its build by how often the grid fills the GPU
// applies a pending update (sums in `acc`) to `st` in place: the tile-binned loop's update outside a warp+scatter launch
// (snap_seq / seq: after the snapshot has been stored and fenced, `seq` goes to the pinned word the host spins on)
void launch_finish_update(DevState* st, MomentAcc* acc, const uint32_t* ovf_prev, int j, int cur_prev, bf_trace_rec* trace,
                          DevState* snap, hipStream_t s, const uint32_t* lost = nullptr, unsigned long long* snap_seq = nullptr,
                          unsigned long long seq = 0);
void launch_compute_uv(const double2* nxny, double2* uv, long long n, hipStream_t s);
void launch_unpermute(const double2* src, const uint32_t* perm, double2* dst, long long n, hipStream_t s);
void launch_expand_pr(const uint32_t* xy, const float2* p, const uint32_t* perm, double2* pr, long long n,
                      hipStream_t s);

// bf_tiles.hip -- grid of independent per-tile optimizers (BASELINE config 4)
struct TileGrid {
    int32_t rows, cols, res_x, res_y;   // tile (r, c) = (fr_x * rows / res_x, fr_y * cols / res_y)
};
struct TileArgs {
    const uint32_t* xy;
    const int32_t* t;
    float2* p;
    const uint32_t* perm;
    double2* nxny;
    const uint32_t* tile_start;
    DevState* states;                  // one per tile: in = zero-model template, out = final state
    int32_t scale, seed_res_x, seed_res_y, guard_res_x, guard_res_y, min_events, max_px;
};
void launch_tile_sort(const uint32_t* xy, const int32_t* t, const uint32_t* perm_in, long long n, const TileGrid& g,
                      uint32_t* hist, uint32_t* start, uint32_t* cursor, uint32_t* oxy, int32_t* ot, float2* op,
                      uint32_t* operm, hipStream_t s);
int launch_tile_optimizer(const TileArgs& a, int ntiles, hipStream_t s);
int launch_tile_optimizer_many(const TileArgs* slices, int nslices, int ntiles, int scale, int max_px, uint32_t* counter, int n_cus,
                               hipStream_t s);
void launch_fill_states(DevState* states, const DevState& tmpl, int nt, hipStream_t s);

// bf_rebin.hip / bf_scatter.hip / bf_stencil.hip / bf_fused.hip (the tile-binned loops)
int bin_kernel_setup();
void launch_stencil_binned(const StencilArgs& a, dim3 grid, hipStream_t s, int n_cus);
// interior + margin format: clear what the bins' lists name in `mplane`, empty the lists (a run that cannot rely on the
// loop's own clean-up: the dirty margin plane is the one its first iteration adds to, or the bin grid changes)
void launch_margin_clean(unsigned long long* mplane, const uint32_t* mlist, uint32_t* mcount, int nbins, int mcap, hipStream_t s);
void launch_rebin(const EvSets& sets, int has_perm, long long n, DevState* st, const BinGrid& g,
                  uint16_t* binid, uint32_t* hist_cnt, uint32_t* bin_start,
                  uint32_t* cursor, uint32_t* armed, const WarpParams* prewarp, int pack_limit, hipStream_t s,
                  uint32_t* ftab = nullptr, uint32_t* lost = nullptr);   // (g.fz != 0: the one-kernel iteration's range table and flag)
struct BinScatterArgs {
    EvSets sets;
    const uint32_t* bin_start;
    unsigned long long* slabs;       // per bin: the dense tile, or the values of its compact list
    uint16_t* cidx;                  // compact lists: tile-local pixel index of every entry (per bin: L * LR slots)
    uint32_t* chdr;                  // compact lists: first entry of every tile row, per bin (LR + 1 words)
    int compact;                     // this slice's scatter writes compact lists (bf_set_cloud's choice)
    unsigned long long* ovf_plane;   // overflow planes of buffer `cur`
    uint32_t* ovf_cplane;
    uint32_t* ovf_bits;              // ... and its dirty bitmap (StencilArgs)
    int ovf_pitch;
    const DevState* st_in;           // state as of the previous launch ...
    DevState* st_out;                // ... and with the pending update applied (written by work-group 0)
    DevState* snap;                  // optional: pinned host copy of st_out, polled by the host
    MomentAcc* acc;                  // moment sums of the previous iteration (parity (j - 1) & 1)
    uint32_t* ovf_cur;               // this iteration's overflow counter
    const uint32_t* ovf_prev;        // the previous iteration's (final)
    bf_trace_rec* trace;
    BinGrid g;
    int cur, j;                      // plane buffer of this iteration; number of stencil launches completed before it
    unsigned long long* tl;
    // interior + margin format (compact == 3): the bin's own TSR x TS pixels go to `slabs` (plain stores, nothing shared), what
    // its events left in the margin of its LDS tile is ADDED to the margin plane of buffer `cur` and listed, and the pixels
    // this bin listed in its previous executed launch are cleared in the other buffer's margin plane
    unsigned long long* m_cur;
    unsigned long long* m_prev;
    uint32_t* mlist;                 // per bin: mcap linear pixel indices
    uint32_t* mcount;                // per bin: entries of its list
    int mcap;
};
// work-group size of the scatter kernel for a slice: format (0 dense slabs, 2 event lists, 3 own pixels + margin plane), where the
// update runs, -- lists -- whether the grid is thousands of small bins, and -- dense tiles -- the events per bin
int bin_scatter_threads(int fmt, bool head, bool many_small_bins, double events_per_bin);
hipError_t launch_bin_warp_scatter(const BinScatterArgs& a, bool warp, int threads, int per_thread, hipStream_t s);
// The one-kernel iteration (k_fused_pass, bf_fused.hip): warp + scatter + stencil + moments of one image tile per work-group.
struct FusedArgs {
    EvSets sets;
    const uint32_t* ftab;            // FusedTab per tile (written by the counting sort's scan)
    const DevState* st_in;           // state as of the previous launch ...
    DevState* st_out;                // ... and with the pending update applied (written by work-group 0)
    DevState* snap;                  // optional: pinned host copy of st_out, polled by the host
    MomentAcc* acc_in;               // moment sums of the previous pass
    MomentAcc* acc_out;              // this pass adds here
    MomentAcc* acc_zero;             // cleared for the next pass
    uint32_t* lost;                  // [3], by launch number mod 3: raised by a pass that cannot vouch for its sums (an event moved
                                     // further than the bins allow)
    bf_trace_rec* trace;
    int nbr, nbc;                    // image tiles
    int R, C;
    int j;                           // launch number (the update of iteration j runs at the head of launch j + 1)
    int warp;                        // 0: scatter the events where their stored products put them (first pass of a cold run)
    unsigned long long* tl;          // debug timeline (`make tl` build only)
};
hipError_t launch_fused_pass(const FusedArgs& a, int half_scale, int rows_per_tile, hipStream_t s);
// The persistent form of the one-kernel iteration (k_fused_loop, bf_loop.hip): up to max_passes iterations per launch, the
// work-groups resident, the moment sums exchanged through tagged records in memory instead of a launch boundary.
struct FusedLoopArgs {
    EvSets sets;
    const uint32_t* ftab;            // FusedTab per tile
    DevState* st;                    // read at entry; written (with st_other and snap) by work-group 0 at exit
    DevState* st_other;
    DevState* snap;                  // pinned host copy: (done, it) after every pass, the whole state at exit
    unsigned long long* rec;         // [2][tiles * NSUB][16][2]: per-sub-tile records (payload, tag), by pass parity
    unsigned long long* red;         // [2][16][16][2]: the reducers' records
    float2* scratch[4];              // private product arrays (lists longer than a pass): the strips' readers' [0 .. 2], the owners' [3]
    bf_trace_rec* trace;
    int nbr, nbc, R, C;
    int max_passes;
    int first_warp;                  // 0: the first pass of the run scatters the stored products as they are
    unsigned long long* tl;          // debug timeline (`make tl` build only)
    int debug_abort;                 // >= 0: every work-group gives up at that pass (BF_DEBUG_PERSIST_ABORT: exercises the undo + fall-back)
    int debug_mute;                  // >= 0: from that pass on the LAST work-group publishes no records, as if it had never become
                                     // resident (BF_DEBUG_PERSIST_MUTE): one reducer really times out, the others do not
    int debug_split, debug_split_late;   // >= 0: the LAST work-group alone "times out" at that pass (BF_DEBUG_PERSIST_SPLIT=<pass>[,late];
                                     // the host makes it the launch's last pass): at once -- its ABORT decides --, or ~100 us late --
                                     // the others have committed and it catches up
    unsigned long long* verdict;     // one word per context: (launch id << 2) | COMMIT / ABORT, decided by compare-and-swap
    int* broken;                     // pinned host word, set if a committed launch's records cannot be read back (cannot happen)
};
hipError_t launch_fused_loop(const FusedLoopArgs& a, int half_scale, int rows_per_tile, int n_cus, hipStream_t s);
// can `ntiles` work-groups of that kernel be resident at once on this device (n_cus compute units)?
bool fused_loop_resident(int half_scale, int rows_per_tile, int n_cus, int ntiles);
void launch_run_init(DevState* st, const DevState& v, uint32_t* ovf, uint32_t prev_dirty, MomentAcc* acc, bool init_loop, hipStream_t s);

// bf_local.hip -- contrast-score evaluation of OptimizerLocal (optimizer_sampler.cpp:120-153)
struct LocalGeom {
    int32_t scale, wsx, wsy, R, C, pad;
    float kx, ky;              // float(n) / nz of Event::apply_project, the same for every event
    double x_shift, y_shift;   // -event_c.pr * scale + metric_wsize / 2.0 (optimizer_sampler.cpp:126-127)
};
void launch_local_project_count(const uint32_t* xy, const int32_t* t, long long n, const LocalGeom& g, uint32_t* plane,
                                hipStream_t s);
int launch_local_blur_score(const uint32_t* plane, uint32_t* zero_plane, const LocalGeom& g, unsigned long long* score,
                            uint8_t* img_out, hipStream_t s);

// a grid of OptimizerLocal windows, one work-group each (bf_local.hip); < 0: scale above 7 / kernel attributes / window too large for the LDS
int launch_local_tile_optimizer(const uint32_t* xy, const int32_t* t, const uint32_t* tile_start, bf_local_state* states, int32_t* rcs,
                                const TileGrid& g, int scale, int wsz, int guard_res_x, int guard_res_y, long long max_evaluations,
                                hipStream_t s);

void launch_proj_count(const uint32_t* xy, const float2* p, const uint8_t* noise, long long n, int scale, int res_x,
                       int res_y, int show_final, uint32_t* plane, hipStream_t s);
void launch_proj_scale(uint8_t* img, long long n, const unsigned long long* score, hipStream_t s);

// EventFile::color_time_img (event_file.h:649-747)
struct ColorGeom {
    int32_t scale, show_final, mx, my, R, C;   // mx = scale * res_x (:668-669), R = mx + scale (:670-671)
    long long t_min, t_range;                  // min t; max(t_max, 0) - t_min (:659-662)
    double x_shift, y_shift;                   // :677-678
};
void launch_color_time(const uint32_t* xy, const int32_t* t, const float2* p, const uint8_t* noise, long long n,
                       const ColorGeom& g, uint32_t* cnt, unsigned long long* pc, unsigned long long* ps, uint8_t* bgr,
                       hipStream_t s);

void launch_copy(const void* src, void* dst, long long bytes, int blocks, bool nontemporal, hipStream_t s);
void launch_eval_sincos(const double* x, long long n, int table, double* sn, double* cs, hipStream_t s);

}  // namespace bf
