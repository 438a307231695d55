// K4: IoU tracker for all streams of a process (replaces tracker.py:45-147).
//
// Layout in HBM: one structure-of-arrays table per stream, `cap` rows, row order == the reference's
// dict insertion order (ids are handed out monotonically, so it is also ascending-id order).
//
// k4_update: one wavefront per stream.  The table is staged in LDS; detections are consumed in
// order (tracker.py:55) because each one may create a track that the next one can match and may
// overwrite the box a later one is compared against.  Per detection the 64 lanes
// stride over the live tracks computing float64 IoU (tracker.py:129-147), then a wave reduction
// picks the maximum IoU with the earliest row on ties -- exactly what the reference's
// `iou >= thr and iou > best_iou` scan in dict order yields.  Pruning (tracker.py:111-126) is a
// ballot/popcount stable compaction written back to HBM.
// New tracks get provisional ids -(k+1); k4_assign_ids turns them into the reference's global
// counter values in canonical order (tick-major, stream-minor), using the new-track counts of ALL
// streams of the job (all-gathered over RCCL when streams are sharded across GPUs).
//
// Under load (float32 source = the post-process output, up to `dm` detections per frame) the float64 IoUs leave the
// serial loop ("K4 matrix on GPU, sequential assign per stream"):
//   k4_iou     all CUs: row d of a per-stream matrix V holds the IoU of detection d with every track the frame started
//              with (columns 0..T0) and with every earlier detection of the frame (columns T0r + d'), class mismatch
//              stored as 0.0 (the reference skips such tracks, and a 0.0 can never win `iou > best`, best >= 0).  Same
//              iou64(), same argument order, -ffp-contract=off: the values the serial scan would have computed, because
//              a row's box at any point of the frame is either its old box or the box of the last detection matched to it.
//              The block that writes a row also reduces it: A[d] = the row the reference's scan would pick if NO row had
//              been touched by an earlier detection of the frame (maximum over columns 0..T0, earliest row on ties), and
//              a flag saying whether ANY earlier detection's box qualifies against d (a column T0r + d' >= min_iou).
//   k4_update  one wave per stream, 64 detections per step, one per lane.  A detection whose flag is clear and whose
//              A row is untouched so far (or which has no A row: a new track) needs no scan: untouched rows still hold the
//              boxes A was computed against, and no touched or new row can reach min_iou.  All such detections up to the
//              first one that fails the test are applied at once (distinct rows, new rows numbered by a ballot prefix);
//              the one that fails -- two detections sharing a best track, or overlapping an earlier detection -- takes the
//              reference's scan over the live rows through a per-row column pointer, then the step resumes behind it.
#include <climits>
#include <cstring>

#include "rva_internal.h"

#define RVA_MAX_TRACKER_STREAMS 512

struct rva_tracker {
    rva_ctx *ctx = nullptr;
    int n_streams = 0, cap = 0, max_age = 0, min_hits = 0;
    double min_iou = 0.0;
    // device state
    int64_t *id = nullptr;     // [S][cap]
    double *box = nullptr;     // [S][cap][4]
    double *conf = nullptr;    // [S][cap]
    int32_t *cls = nullptr;    // [S][cap]
    int32_t *age = nullptr;    // [S][cap]
    int32_t *hits = nullptr;   // [S][cap]
    int32_t *last_det = nullptr;  // [S][cap] index of the last detection of the latest update that wrote the row, -1 if none
    int32_t *n_tracks = nullptr;  // [S]
    int32_t *n_new = nullptr;     // [S]
    int64_t *next_id = nullptr;   // [1] the reference's itertools.count(1) (tracker.py:47)
    int32_t *id_ticket = nullptr; // [1] arrival counter of k4_assign_ids (returns to 0 every launch)
    int32_t *flags = nullptr;     // [1]
    int32_t *d_slot = nullptr;    // [S] per-tick: slot / active
    int32_t *d_offs = nullptr;    // [S+1]
    int32_t *d_gidx = nullptr;    // [S]
    double *d_bscale = nullptr;   // [S] box scale of _rescale_detections (1.0 = no downsample)
    double *d_V = nullptr;        // [S][dm][ld] IoU matrix of the tick (k4_iou -> k4_update), null: always the in-loop form
    int32_t *d_A = nullptr;       // [S][dm] per detection: (row picked among the untouched tracks + 1) | (an earlier detection qualifies) << 16
    int dm = 0, ld = 0;           // detections per frame the matrix holds; row stride in doubles (multiple of 128)
    // pre-detector gates decided on the device (pipeline.py:156-170, 242-262), see rva_tracker_set_gates
    int32_t *gate_cfg = nullptr;    // [S][4] adaptive enabled, max_process_every, idle_tolerance, motion minimum count
    int32_t *gate_state = nullptr;  // [S][4] frame_index, idle_frames, process_every, -
    int32_t *emitted = nullptr;     // [S] len(filtered) of the last update (pipeline.py:187)
    int32_t *processed = nullptr;   // [S] last update: 1 processed, 0 skipped frame, -1 no frame
    // pinned host staging
    int32_t *h_slot = nullptr, *h_offs = nullptr;
    void *h_read[RVA_SNAPSHOT_SLOTS] = {};       // pinned snapshot slots
    void *h_read_dev[RVA_SNAPSHOT_SLOTS] = {};   // the same slots as the device sees them
    hipEvent_t snap_done[RVA_SNAPSHOT_SLOTS] = {};
    size_t h_read_bytes = 0;
    hipEvent_t staged = nullptr;  // completion of the last async copy out of h_slot/h_offs
    std::vector<int32_t> gidx_cached;
};

namespace {

struct K4Args {
    int64_t *id; double *box; double *conf; int32_t *cls, *age, *hits, *last_det, *n_tracks, *n_new, *flags;
    int cap, max_age, min_hits;
    double min_iou;
    const int32_t *slot;  // f64 path: device [S] active flags (staged); f32 path: unused
    int16_t kslot[RVA_MAX_TRACKER_STREAMS];  // f32 path: batch row, -1 no frame, -2 skipped frame, -3 not part of this
                                             // launch (kernarg-resident: no staging buffer to race with, graph-capturable)
    int16_t mrow[RVA_MAX_TRACKER_STREAMS];   // f32 path, device gates: row of motion_cnt holding the stream's K5 count, -1 = no gate
    const int32_t *motion_cnt;               // K5 counts (-1 = first frame of the stream: always processed)
    int32_t *gate_cfg, *gate_state;          // null: gates are decided by the host (kslot -2)
    int32_t *emitted, *processed;
    // f32 source (post-process outputs)
    const float4 *boxes32; const float *scores32; const int32_t *cls32; const int32_t *counts32; int max_det;
    const double *bscale;   // per-stream multiplier applied to the widened box (pipeline.py:237)
    double filter_thr;
    // f64 source (host API)
    const int32_t *offs; const double *boxes64; const double *conf64; const int64_t *cls64;
    // IoU matrix path (f32 source only)
    double *V; int32_t *A; int dm, ld;
};

constexpr int K4_MATRIX_CAP = 1024;   // matrix path: capacities up to this many rows per stream

__device__ __forceinline__ int k4_round128(int v) { return (v + 127) & ~127; }
inline int k4_round128_host(int v) { return (v + 127) & ~127; }

// Pre-detector gates decided on the device (f32 path): motion gate first (utils/frame_filter.py:26-40 via the K5 count,
// first frame always passes), then the adaptive-fps gate (pipeline.py:165-170).  Reads the state the previous tick left.
struct K4Gate { bool process; int fi, idle, pe, on, maxpe, tol; };

__device__ __forceinline__ K4Gate k4_gate(const K4Args &a, int s, int slot)
{
    K4Gate g{slot >= 0, 0, 0, 1, 0, 1, 0};
    if (a.gate_cfg) {
        const int32_t *cfg = a.gate_cfg + 4 * s;
        const int32_t *st = a.gate_state + 4 * s;
        g.on = cfg[0]; g.maxpe = cfg[1]; g.tol = cfg[2];
        g.fi = st[0] + 1; g.idle = st[1]; g.pe = st[2];            // pipeline.py:144: the frame index advances for every frame
        const int mr = a.mrow[s];
        if (mr >= 0) {
            const int cnt = a.motion_cnt[mr];
            if (cnt >= 0 && cnt < cfg[3]) g.process = false;       // ratio < motion_threshold (host turned it into a count)
        }
        if (g.on && g.pe > 1 && (g.fi - 1) % g.pe != 0) g.process = false;
    }
    return g;
}

// tracker.py:129-147, float64; a = track, b = detection
__device__ __forceinline__ double iou64(const double a0, const double a1, const double a2, const double a3,
                                        const double b0, const double b1, const double b2, const double b3)
{
    const double ix1 = a0 > b0 ? a0 : b0, iy1 = a1 > b1 ? a1 : b1;
    const double ix2 = a2 < b2 ? a2 : b2, iy2 = a3 < b3 ? a3 : b3;
    double iw = ix2 - ix1, ih = iy2 - iy1;
    iw = iw > 0.0 ? iw : 0.0;
    ih = ih > 0.0 ? ih : 0.0;
    const double inter = iw * ih;
    double aw = a2 - a0, ah = a3 - a1, bw = b2 - b0, bh = b3 - b1;
    aw = aw > 0.0 ? aw : 0.0; ah = ah > 0.0 ? ah : 0.0;
    bw = bw > 0.0 ? bw : 0.0; bh = bh > 0.0 ? bh : 0.0;
    const double area_a = aw * ah, area_b = bw * bh;
    const double uni = area_a + area_b - inter;
    if (uni <= 0.0) return 0.0;
    return __ddiv_rn(inter, uni);
}

extern __shared__ __attribute__((aligned(16))) unsigned char k4_smem[];

// a detection of the float32 source as the tracker sees it: exact widening, then _rescale_detections (float64 multiply)
__device__ __forceinline__ void k4_det_box(const K4Args &a, size_t o, double bs, double &b0, double &b1, double &b2, double &b3)
{
    const float4 f = a.boxes32[o];
    b0 = (double)f.x; b1 = (double)f.y; b2 = (double)f.z; b3 = (double)f.w;
    if (bs != 1.0) { b0 *= bs; b1 *= bs; b2 *= bs; b3 *= bs; }
}

// ---- IoU matrix of the tick -------------------------------------------------------------------------------------------
// grid = (K4_IOU_BX, streams), 256 threads.  A block takes tiles of 4 detections (rows of V) and sweeps the columns: the
// T0 tracks the frame starts with, then the detections before the row's own; every thread keeps the best qualifying track
// column of its share (ascending, `>`: earliest on ties) and the block reduces them (LDS max of the value's bit pattern --
// IoUs are non-negative doubles, so the patterns order like the values -- then min of the row among the holders of the
// maximum).  Streams that k4_update will not run the matrix form for (no frame, skipped frame, gated out, more than dm
// detections) return at once.
constexpr int K4_IOU_BX = 16;

__global__ void __launch_bounds__(256) k4_iou(K4Args a)
{
    __shared__ unsigned long long mx[4];
    __shared__ int mi[4], bf[4];
    const int s = blockIdx.y, tid = threadIdx.x;
    const int slot = (int)a.kslot[s];
    if (slot < 0) return;
    if (!k4_gate(a, s, slot).process) return;
    const int D = a.counts32[slot];
    if (D <= 0 || D > a.dm) return;
    const int T0 = a.n_tracks[s], T0r = k4_round128(T0);
    const double bs = a.bscale[s];
    const size_t tb = (size_t)s * a.cap;
    double *Vs = a.V + (size_t)s * a.dm * a.ld;
    for (int d0 = blockIdx.x * 4; d0 < D; d0 += K4_IOU_BX * 4) {
        if (tid < 4) { mx[tid] = 0ull; mi[tid] = INT_MAX; bf[tid] = 0; }
        __syncthreads();
        double db[4][4], lb[4] = {0.0, 0.0, 0.0, 0.0};   // tracker.py:100 best_iou = 0.0
        int dc[4], li[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d = d0 + q < D ? d0 + q : D - 1;
            const size_t o = (size_t)slot * a.max_det + d;
            k4_det_box(a, o, bs, db[q][0], db[q][1], db[q][2], db[q][3]);
            dc[q] = a.cls32[o];
        }
        for (int j = tid; j < T0; j += 256) {                  // columns 0..T0: the tracks as the frame finds them
            const double2 t01 = reinterpret_cast<const double2 *>(a.box)[2 * (tb + j)];
            const double2 t23 = reinterpret_cast<const double2 *>(a.box)[2 * (tb + j) + 1];
            const int tc = a.cls[tb + j];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (d0 + q < D) {                                // tracker.py:103-105; a = track, b = detection
                    const double v = tc == dc[q] ? iou64(t01.x, t01.y, t23.x, t23.y, db[q][0], db[q][1], db[q][2], db[q][3]) : 0.0;
                    Vs[(size_t)(d0 + q) * a.ld + j] = v;
                    if (v >= a.min_iou && v > lb[q]) { lb[q] = v; li[q] = j; }      // :106
                }
        }
        for (int e = tid; e < d0 + 3 && e < D; e += 256) {     // columns T0r + e: a row that detection e (re)wrote earlier in the frame
            const size_t o = (size_t)slot * a.max_det + e;
            double e0, e1, e2, e3;
            k4_det_box(a, o, bs, e0, e1, e2, e3);
            const int ec = a.cls32[o];                           // a row's class is the class of every detection matched to it
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (d0 + q < D && e < d0 + q) {
                    const double v = ec == dc[q] ? iou64(e0, e1, e2, e3, db[q][0], db[q][1], db[q][2], db[q][3]) : 0.0;
                    Vs[(size_t)(d0 + q) * a.ld + T0r + e] = v;
                    if (v >= a.min_iou && v > 0.0) bf[q] = 1;    // could win the scan if its row is live: k4_update decides
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (lb[q] > 0.0) __hip_atomic_fetch_max(&mx[q], (unsigned long long)__double_as_longlong(lb[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (lb[q] > 0.0 && (unsigned long long)__double_as_longlong(lb[q]) == mx[q])
                __hip_atomic_fetch_min(&mi[q], li[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        if (tid < 4 && d0 + tid < D) a.A[(size_t)s * a.dm + d0 + tid] = (mi[tid] == INT_MAX ? 0 : mi[tid] + 1) | (bf[tid] << 16);
    }
}

template <bool F64SRC>
__global__ void __launch_bounds__(64) k4_update(K4Args a)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const int slot = F64SRC ? a.slot[s] : (int)a.kslot[s];
    if (F64SRC ? slot == 2 : slot == -3) return;   // another launch of this tick owns the stream: leave everything alone
    if (F64SRC ? slot == 0 : slot == -1) {  // stream not updated this tick
        if (lane == 0) { a.n_new[s] = 0; a.processed[s] = -1; a.emitted[s] = 0; }
        return;
    }
    K4Gate g{true, 0, 0, 1, 0, 1, 0};
    if (!F64SRC) g = k4_gate(a, s, slot);
    const bool process = g.process;
    const int cap = a.cap;
    const size_t tb = (size_t)s * cap;
    int n = a.n_tracks[s];
    int D = 0, d0 = 0;
    if (F64SRC) { d0 = a.offs[s]; D = a.offs[s + 1] - d0; }
    else if (process) D = a.counts32[slot];
    int created = 0, n_emit = 0, base = 0;
    bool overflow = false;

    if (!F64SRC && a.V && D <= a.dm) {
        // ---------------- matrix form: the IoUs are in V, the picks of untouched rows in A (k4_iou) ------------------------
        const int T0 = n, T0r = k4_round128(T0);
        int32_t *l_cnt = (int32_t *)k4_smem;        // [cap] matches of the frame per row (a new row starts at 1)
        int32_t *l_src = l_cnt + cap;               // [cap] the column of V that stands for the row now: itself, or T0r + last matching detection
        int32_t *l_first = l_src + cap;             // [cap] earliest detection of the frame that picked / touched the row
        const double *Vs = a.V + (size_t)s * a.dm * a.ld;
        const int32_t *As = a.A + (size_t)s * a.dm;
        const size_t ob = (size_t)slot * a.max_det;
        int nx_info = 0;
        float nx_sc = 0.f;
        if (lane < D) { nx_info = As[lane]; nx_sc = a.scores32[ob + lane]; }
        for (int k = lane; k < cap; k += 64) { l_cnt[k] = 0; l_src[k] = k; l_first[k] = INT_MAX; }
        __syncthreads();
        const unsigned long long below = (1ull << lane) - 1ull;
        for (int db = 0; db < D; db += 64) {
            const int d = db + lane;
            const int info = nx_info;
            const float sc = nx_sc;
            if (d + 64 < D) { nx_info = As[d + 64]; nx_sc = a.scores32[ob + d + 64]; }
            // filter_detections (pipeline.py:182): detections below the threshold do not exist for the tracker
            const bool pass = d < D && (double)sc >= a.filter_thr;
            n_emit += __popcll(__ballot(pass));
            const int arow = (info & 0xffff) - 1;
            const bool later = (info >> 16) & 1;
            if (pass && arow >= 0) __hip_atomic_fetch_min(&l_first[arow], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
            int start = 0;
            for (;;) {
                const bool open = pass && lane >= start;
                const bool scan = open && (later || (arow >= 0 && l_first[arow] != d));
                const unsigned long long sm = __ballot(scan);
                const int c = sm ? __builtin_ctzll(sm) : 64;
                // lanes [start, c): distinct untouched rows or new tracks, none of them visible to another -- all at once
                const bool act = open && lane < c;
                const bool fresh = act && arow < 0;             // tracker.py:69-80 new track, appended in detection order
                const unsigned long long nm = __ballot(fresh);
                const int room = cap - n, rank = __popcll(nm & below), nn = __popcll(nm);
                if (fresh && rank < room) { l_cnt[n + rank] = 1; l_src[n + rank] = T0r + d; }
                if (act && arow >= 0) { l_cnt[arow] += 1; l_src[arow] = T0r + d; }   // :81-92
                if (nn > room) overflow = true;
                n += nn < room ? nn : room;
                created += nn < room ? nn : room;
                if (c == 64) break;
                __syncthreads();
                // detection db + c: the reference's scan over the live rows (tracker.py:100-109), each at its current column
                const int dc = db + c;
                const double *row = Vs + (size_t)dc * a.ld;
                double best = 0.0;
                int bi = INT_MAX;
                for (int k = lane; k < n; k += 64) {
                    const double v = row[l_src[k]];
                    if (v >= a.min_iou && v > best) { best = v; bi = k; }     // rows ascend within a lane
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double vb = __shfl_xor(best, off);
                    const int vi = __shfl_xor(bi, off);
                    if (vb > best || (vb == best && vi < bi)) { best = vb; bi = vi; }
                }
                if (bi == INT_MAX) {
                    if (n >= cap) overflow = true;
                    else {
                        if (lane == 0) { l_cnt[n] = 1; l_src[n] = T0r + dc; }
                        ++n; ++created;
                    }
                } else if (lane == 0) {
                    l_cnt[bi] += 1; l_src[bi] = T0r + dc;
                    if (dc < l_first[bi]) l_first[bi] = dc;     // a later detection that picked this row must scan too
                }
                __syncthreads();
                start = c + 1;
            }
            __syncthreads();
        }
        // every row: its final values (old ones, or those of the last detection matched to it), prune, stable compaction in
        // place (a group of 64 rows is read whole before any of its rows is written, and only rows <= the group are written)
        const double bs = a.bscale[s];
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            const bool valid = k < n, old = valid && k < T0;
            const int col = valid ? l_src[k] : 0;
            const bool touched = valid && col >= T0r;
            double b0 = 0, b1 = 0, b2 = 0, b3 = 0, cf = 0;
            long long id = 0;
            int cl = 0, ag = 0, hi = 0, match = 0;
            if (old) {
                id = a.id[tb + k]; cl = a.cls[tb + k]; ag = a.age[tb + k]; hi = a.hits[tb + k];
                if (!touched) {
                    const double2 t01 = reinterpret_cast<const double2 *>(a.box)[2 * (tb + k)];
                    const double2 t23 = reinterpret_cast<const double2 *>(a.box)[2 * (tb + k) + 1];
                    b0 = t01.x; b1 = t01.y; b2 = t23.x; b3 = t23.y; cf = a.conf[tb + k];
                }
            }
            bool keep = false;
            if (touched) {
                const int dd = col - T0r;
                k4_det_box(a, ob + dd, bs, b0, b1, b2, b3);
                cf = (double)a.scores32[ob + dd];
                if (!old) { id = -(long long)(k - T0 + 1); cl = a.cls32[ob + dd]; }
                hi += l_cnt[k]; ag = 0; match = dd + 1;
                keep = true;
            } else if (valid) {
                ag += 1;
                keep = !(ag > a.max_age || hi < a.min_hits);
            }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const size_t o = tb + base + __popcll(m & below);
                reinterpret_cast<double2 *>(a.box)[2 * o] = make_double2(b0, b1);
                reinterpret_cast<double2 *>(a.box)[2 * o + 1] = make_double2(b2, b3);
                a.conf[o] = cf; a.id[o] = id; a.cls[o] = cl; a.age[o] = ag; a.hits[o] = hi;
                a.last_det[o] = match - 1;
            }
            base += __popcll(m);
        }
    } else {
    // ---------------- in-loop form: table staged in LDS, float64 IoU inside the detection loop ------------------------
    double *l_box = (double *)k4_smem;                 // [4][cap]  (component-major: conflict-free)
    double *l_conf = l_box + 4 * (size_t)cap;          // [cap]
    int64_t *l_id = (int64_t *)(l_conf + cap);         // [cap]
    int32_t *l_cls = (int32_t *)(l_id + cap);          // [cap]
    int32_t *l_age = l_cls + cap;
    int32_t *l_hits = l_age + cap;
    int32_t *l_match = l_hits + cap;

    for (int k = lane; k < n; k += 64) {
        const double *gb = a.box + (tb + k) * 4;
        l_box[k] = gb[0]; l_box[cap + k] = gb[1]; l_box[2 * cap + k] = gb[2]; l_box[3 * cap + k] = gb[3];
        l_conf[k] = a.conf[tb + k];
        l_id[k] = a.id[tb + k];
        l_cls[k] = a.cls[tb + k];
        l_age[k] = a.age[tb + k];
        l_hits[k] = a.hits[tb + k];
        l_match[k] = 0;
    }
    __syncthreads();

    for (int d = 0; d < D; ++d) {
        double b0, b1, b2, b3, dconf;
        int dcls;
        if (F64SRC) {
            const double *p = a.boxes64 + (size_t)(d0 + d) * 4;
            b0 = p[0]; b1 = p[1]; b2 = p[2]; b3 = p[3];
            dconf = a.conf64[d0 + d];
            dcls = (int)a.cls64[d0 + d];
        } else {
            const size_t o = (size_t)slot * a.max_det + d;
            k4_det_box(a, o, a.bscale[s], b0, b1, b2, b3);
            dconf = (double)a.scores32[o];
            dcls = a.cls32[o];
            if (!(dconf >= a.filter_thr)) continue;  // filter_detections, pipeline.py:182 (wave-uniform)
        }
        ++n_emit;
        double best = 0.0;  // tracker.py:100
        int bi = INT_MAX;
        for (int k = lane; k < n; k += 64) {
            if (l_cls[k] != dcls) continue;  // :103-104
            const double v = iou64(l_box[k], l_box[cap + k], l_box[2 * cap + k], l_box[3 * cap + k], b0, b1, b2, b3);
            if (v >= a.min_iou && v > best) { best = v; bi = k; }  // :106 (rows ascend within a lane)
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (bi == INT_MAX) {  // :69-80 new track, appended immediately
            if (n >= cap) { overflow = true; continue; }
            if (lane == 0) {
                l_box[n] = b0; l_box[cap + n] = b1; l_box[2 * cap + n] = b2; l_box[3 * cap + n] = b3;
                l_conf[n] = dconf; l_id[n] = -(int64_t)(created + 1);
                l_cls[n] = dcls; l_age[n] = 0; l_hits[n] = 1; l_match[n] = d + 1;
            }
            ++n; ++created;
        } else if (lane == 0) {  // :81-92 (class_id is not updated)
            l_box[bi] = b0; l_box[cap + bi] = b1; l_box[2 * cap + bi] = b2; l_box[3 * cap + bi] = b3;
            l_conf[bi] = dconf; l_hits[bi] += 1; l_age[bi] = 0; l_match[bi] = d + 1;
        }
        __syncthreads();
    }
    // :111-126 prune, stable compaction back to HBM
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        bool keep = false;
        int ag = 0;
        if (k < n) {
            ag = l_age[k];
            if (l_match[k]) keep = true;
            else { ag += 1; keep = !(ag > a.max_age || l_hits[k] < a.min_hits); }
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const size_t o = tb + base + __popcll(m & ((1ull << lane) - 1ull));
            double *gb = a.box + o * 4;
            gb[0] = l_box[k]; gb[1] = l_box[cap + k]; gb[2] = l_box[2 * cap + k]; gb[3] = l_box[3 * cap + k];
            a.conf[o] = l_conf[k]; a.id[o] = l_id[k]; a.cls[o] = l_cls[k]; a.age[o] = ag; a.hits[o] = l_hits[k];
            a.last_det[o] = l_match[k] - 1;
        }
        base += __popcll(m);
    }
    }
    if (lane == 0) {
        a.n_tracks[s] = base;
        a.n_new[s] = created;
        a.emitted[s] = n_emit;
        a.processed[s] = process ? 1 : 0;
        if (overflow) atomicOr(a.flags, 1);
        if (!F64SRC && a.gate_cfg) {                 // _adjust_adaptive_state(len(filtered), len(tracks)), pipeline.py:242-262
            if (g.on) {
                if (n_emit > 0 || base > 0) { g.idle = 0; g.pe = 1; }
                else { ++g.idle; if (g.idle >= g.tol) g.pe = g.maxpe > 1 ? g.maxpe : 1; }
            }
            int32_t *st = a.gate_state + 4 * s;
            st[0] = g.fi; st[1] = g.idle; st[2] = g.pe;
        }
    }
}

// One block per local stream.  counts_all[n_global] in canonical order; local stream s sits at
// gidx[s].  Every block reads the id counter, then takes a ticket; the LAST block to arrive advances
// the counter (so no block can read the advanced value) and re-arms the ticket: safe under any
// dispatch order and identical on every replay of a captured graph.
__global__ void __launch_bounds__(64) k4_assign_ids(int64_t *id, const int32_t *n_tracks, const int32_t *counts_all,
                                                    int n_global, const int32_t *gidx, int cap,
                                                    int64_t *next_id, int32_t *ticket)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const int g = gidx ? gidx[s] : s;
    long long before = 0, total = 0;   // exclusive prefix at g, and the grand total
    for (int i = lane; i < n_global; i += 64) {
        const long long c = counts_all[i];
        total += c;
        if (i < g) before += c;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        before += __shfl_xor(before, off);
        total += __shfl_xor(total, off);
    }
    const long long first = __hip_atomic_load(next_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long b = first + before;
    const int n = n_tracks[s];
    // provisional ids are only ever at the tail of the table (new rows are appended)
    for (int k = n - 1 - lane; k >= 0; k -= 64) {
        const long long v = id[(size_t)s * cap + k];
        const bool neg = v < 0;
        if (neg) id[(size_t)s * cap + k] = b + (-v - 1);
        if (!__any(neg)) break;
    }
    if (lane == 0) {
        __threadfence();  // the counter read above is complete before the ticket is taken
        if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
            __hip_atomic_store(next_id, first + total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// dynamic LDS of k4_update: the staged table of the in-loop form (the matrix form's three int32 arrays fit inside it)
size_t k4_smem_bytes(int cap) { return (size_t)cap * (4 * 8 + 8 + 8 + 4 * 4) + 64; }

int launch_update(rva_tracker *t, K4Args &a, bool f64, hipStream_t stream)
{
    a.id = t->id; a.box = t->box; a.conf = t->conf; a.cls = t->cls; a.age = t->age; a.hits = t->hits; a.last_det = t->last_det;
    a.n_tracks = t->n_tracks; a.n_new = t->n_new; a.flags = t->flags;
    a.emitted = t->emitted; a.processed = t->processed;
    a.cap = t->cap; a.max_age = t->max_age; a.min_hits = t->min_hits; a.min_iou = t->min_iou;
    const size_t smem = k4_smem_bytes(t->cap);
    a.V = f64 ? nullptr : t->d_V; a.A = t->d_A; a.dm = t->dm; a.ld = t->ld;
    if (f64) k4_update<true><<<t->n_streams, 64, smem, stream>>>(a);
    else {
        if (a.V && a.boxes32) k4_iou<<<dim3(K4_IOU_BX, t->n_streams), 256, 0, stream>>>(a);
        k4_update<false><<<t->n_streams, 64, smem, stream>>>(a);
    }
    RVA_HIP(t->ctx, hipGetLastError());
    return RVA_OK;
}

}  // namespace

extern "C" {

int rva_tracker_create(rva_ctx *ctx, int n_streams, int capacity, int max_age, double max_iou_distance, int min_hits,
                       rva_tracker **out)
{
    if (!ctx || !out || n_streams <= 0 || n_streams > RVA_MAX_TRACKER_STREAMS || capacity <= 0)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_tracker_create: bad argument (1 <= n_streams <= %d)", RVA_MAX_TRACKER_STREAMS);
    // IoU matrix of a tick (k4_iou -> k4_update): up to dm detections per frame against cap tracks + dm detections, kept
    // within 512 MiB for all streams; busier frames (and capacities beyond 1024 rows) use the in-loop form
    int dm = 0, ld = 0;
    if (capacity <= K4_MATRIX_CAP) {
        const int cap_r = k4_round128_host(capacity);
        dm = cap_r < 512 ? cap_r : 512;
        while (dm > 128 && (size_t)n_streams * dm * (cap_r + dm) * 8 > ((size_t)512 << 20)) dm -= 128;
        if ((size_t)n_streams * dm * (cap_r + dm) * 8 > ((size_t)512 << 20)) dm = 0;
        ld = cap_r + dm;
    }
    const size_t smem = k4_smem_bytes(capacity);
    if (smem > 160 * 1024) return rva_fail(ctx, RVA_ERR_CAPACITY, "tracker capacity %d needs %zu B of LDS (max 160 KiB)", capacity, smem);
    RVA_HIP(ctx, hipSetDevice(ctx->device));
    RVA_HIP(ctx, rva_func_smem((const void *)k4_update<true>, smem));
    RVA_HIP(ctx, rva_func_smem((const void *)k4_update<false>, smem));
    rva_tracker *t = new rva_tracker();
    t->ctx = ctx; t->n_streams = n_streams; t->cap = capacity; t->max_age = max_age; t->min_hits = min_hits;
    t->min_iou = max_iou_distance;
    const size_t sc = (size_t)n_streams * capacity;
    RVA_HIP(ctx, hipMalloc(&t->id, sc * 8));
    RVA_HIP(ctx, hipMalloc(&t->box, sc * 32));
    RVA_HIP(ctx, hipMalloc(&t->conf, sc * 8));
    RVA_HIP(ctx, hipMalloc(&t->cls, sc * 4));
    RVA_HIP(ctx, hipMalloc(&t->age, sc * 4));
    RVA_HIP(ctx, hipMalloc(&t->hits, sc * 4));
    RVA_HIP(ctx, hipMalloc(&t->last_det, sc * 4));
    RVA_HIP(ctx, hipMalloc(&t->n_tracks, n_streams * 4));
    RVA_HIP(ctx, hipMalloc(&t->n_new, n_streams * 4));
    RVA_HIP(ctx, hipMalloc(&t->next_id, 8));
    RVA_HIP(ctx, hipMalloc(&t->id_ticket, 4));
    RVA_HIP(ctx, hipMemset(t->id_ticket, 0, 4));
    RVA_HIP(ctx, hipMalloc(&t->flags, 4));
    RVA_HIP(ctx, hipMalloc(&t->d_slot, n_streams * 4));
    RVA_HIP(ctx, hipMalloc(&t->d_offs, (n_streams + 1) * 4));
    RVA_HIP(ctx, hipMalloc(&t->d_gidx, n_streams * 4));
    RVA_HIP(ctx, hipMalloc(&t->d_bscale, n_streams * 8));
    {
        std::vector<double> ones(n_streams, 1.0);
        RVA_HIP(ctx, hipMemcpy(t->d_bscale, ones.data(), n_streams * 8, hipMemcpyHostToDevice));
    }
    if (dm > 0) {
        // the IoU matrix of the busy-scene form (k4_iou): n_streams x dm x ld doubles, up to 512 MiB (32 streams x capacity 1024:
        // 201 MB).  It is an accelerator, not a requirement: when HBM cannot give it, the tracker keeps the in-loop form (dm = 0,
        // same results) instead of failing the creation.
        if (hipMalloc(&t->d_V, (size_t)n_streams * dm * ld * sizeof(double)) == hipSuccess &&
            hipMalloc(&t->d_A, (size_t)n_streams * dm * sizeof(int32_t)) == hipSuccess) {
            t->dm = dm; t->ld = ld;
        } else {
            (void)hipGetLastError();                       // clear the sticky allocation error
            if (t->d_V) { (void)hipFree(t->d_V); t->d_V = nullptr; }
            t->d_A = nullptr; t->dm = 0; t->ld = 0;
        }
    }
    RVA_HIP(ctx, hipMalloc(&t->gate_cfg, n_streams * 16));
    RVA_HIP(ctx, hipMalloc(&t->gate_state, n_streams * 16));
    RVA_HIP(ctx, hipMalloc(&t->emitted, n_streams * 4));
    RVA_HIP(ctx, hipMalloc(&t->processed, n_streams * 4));
    RVA_HIP(ctx, hipMemset(t->gate_cfg, 0, n_streams * 16));
    RVA_HIP(ctx, hipMemset(t->gate_state, 0, n_streams * 16));
    RVA_HIP(ctx, hipMemset(t->emitted, 0, n_streams * 4));
    RVA_HIP(ctx, hipMemset(t->processed, 0xff, n_streams * 4));
    RVA_HIP(ctx, hipHostMalloc(&t->h_slot, n_streams * 4));
    RVA_HIP(ctx, hipHostMalloc(&t->h_offs, (n_streams + 1) * 4));
    RVA_HIP(ctx, hipEventCreateWithFlags(&t->staged, hipEventDisableTiming));
    t->h_read_bytes = sc * (8 + 32 + 8 + 4 + 4 + 4 + 4) + n_streams * 12 + 64 * 12;
    for (int i = 0; i < RVA_SNAPSHOT_SLOTS; ++i) {
        RVA_HIP(ctx, hipHostMalloc(&t->h_read[i], t->h_read_bytes, hipHostMallocMapped));
        RVA_HIP(ctx, hipHostGetDevicePointer(&t->h_read_dev[i], t->h_read[i], 0));
        RVA_HIP(ctx, hipEventCreateWithFlags(&t->snap_done[i], hipEventDisableTiming));
    }
    RVA_HIP(ctx, hipMemset(t->n_tracks, 0, n_streams * 4));
    RVA_HIP(ctx, hipMemset(t->n_new, 0, n_streams * 4));
    RVA_HIP(ctx, hipMemset(t->flags, 0, 4));
    const int64_t one = 1;  // itertools.count(1), tracker.py:47
    RVA_HIP(ctx, hipMemcpy(t->next_id, &one, 8, hipMemcpyHostToDevice));
    *out = t;
    return RVA_OK;
}

void rva_tracker_destroy(rva_tracker *t)
{
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipDeviceSynchronize();
    void *dev[] = {t->id, t->box, t->conf, t->cls, t->age, t->hits, t->last_det, t->n_tracks, t->n_new, t->next_id, t->id_ticket, t->flags,
                   t->d_slot, t->d_offs, t->d_gidx, t->d_bscale, t->gate_cfg, t->gate_state, t->emitted, t->processed, t->d_V, t->d_A};
    for (void *p : dev) (void)hipFree(p);
    void *host[] = {t->h_slot, t->h_offs};
    for (int i = 0; i < RVA_SNAPSHOT_SLOTS; ++i) {
        if (t->snap_done[i]) (void)hipEventDestroy(t->snap_done[i]);
        (void)hipHostFree(t->h_read[i]);
    }
    for (void *p : host) (void)hipHostFree(p);
    if (t->staged) (void)hipEventDestroy(t->staged);
    delete t;
}

static int update_f32_common(rva_tracker *t, const int32_t *slot_of_stream, const float *boxes, const float *scores,
                             const int32_t *cls, const int32_t *counts, int max_det, double filter_thr, bool gated,
                             const int32_t *motion_counts, const int32_t *motion_row, hipStream_t stream)
{
    if (!t || !slot_of_stream) return RVA_ERR_ARG;
    bool any = false;
    K4Args a{};
    for (int s = 0; s < t->n_streams; ++s) {
        const int v = slot_of_stream[s];
        if (v < -3 || v > 32767) return rva_fail(t->ctx, RVA_ERR_ARG, "rva_tracker_update_f32: slot_of_stream[%d] = %d out of range", s, v);
        a.kslot[s] = (int16_t)v;
        any |= v >= 0;
        const int mr = gated && motion_row ? motion_row[s] : -1;
        if (mr < -1 || mr > 32767 || (mr >= 0 && !motion_counts))
            return rva_fail(t->ctx, RVA_ERR_ARG, "rva_tracker_update_gated_f32: motion_row[%d] = %d without counts / out of range", s, mr);
        a.mrow[s] = (int16_t)mr;
    }
    if (any && (!boxes || !scores || !cls || !counts || max_det <= 0))
        return rva_fail(t->ctx, RVA_ERR_ARG, "rva_tracker_update_f32: detection arrays missing");
    a.boxes32 = (const float4 *)boxes; a.scores32 = scores; a.cls32 = cls; a.counts32 = counts; a.max_det = max_det;
    a.filter_thr = filter_thr;
    a.bscale = t->d_bscale;
    a.motion_cnt = motion_counts;
    a.gate_cfg = gated ? t->gate_cfg : nullptr;
    a.gate_state = gated ? t->gate_state : nullptr;
    return launch_update(t, a, false, stream);
}

int rva_tracker_update_f32(rva_tracker *t, const int32_t *slot_of_stream, const float *boxes, const float *scores,
                           const int32_t *cls, const int32_t *counts, int max_det, double filter_thr,
                           rva_stream_t stream_)
{
    return update_f32_common(t, slot_of_stream, boxes, scores, cls, counts, max_det, filter_thr, false, nullptr, nullptr,
                             (hipStream_t)stream_);
}

int rva_tracker_update_gated_f32(rva_tracker *t, const int32_t *slot_of_stream, const float *boxes, const float *scores,
                                 const int32_t *cls, const int32_t *counts, int max_det, double filter_thr,
                                 const int32_t *motion_counts, const int32_t *motion_row, rva_stream_t stream_)
{
    return update_f32_common(t, slot_of_stream, boxes, scores, cls, counts, max_det, filter_thr, true, motion_counts, motion_row,
                             (hipStream_t)stream_);
}

int rva_tracker_set_gates(rva_tracker *t, const int32_t *adaptive_enabled, const int32_t *max_process_every,
                          const int32_t *idle_tolerance, const int32_t *motion_min_count, int reset_state)
{
    if (!t || !adaptive_enabled || !max_process_every || !idle_tolerance || !motion_min_count) return RVA_ERR_ARG;
    std::vector<int32_t> cfg((size_t)t->n_streams * 4), st((size_t)t->n_streams * 4, 0);
    for (int s = 0; s < t->n_streams; ++s) {
        cfg[4 * s] = adaptive_enabled[s] ? 1 : 0; cfg[4 * s + 1] = max_process_every[s]; cfg[4 * s + 2] = idle_tolerance[s];
        cfg[4 * s + 3] = motion_min_count[s];
        st[4 * s + 2] = 1;                                   // process_every starts at 1 (pipeline.py:106)
    }
    RVA_HIP(t->ctx, hipDeviceSynchronize());
    RVA_HIP(t->ctx, hipMemcpy(t->gate_cfg, cfg.data(), cfg.size() * 4, hipMemcpyHostToDevice));
    if (reset_state) RVA_HIP(t->ctx, hipMemcpy(t->gate_state, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    return RVA_OK;
}

int rva_tracker_update_f64(rva_tracker *t, const int32_t *active, const int32_t *offsets, const double *boxes,
                           const double *conf, const int64_t *cls, rva_stream_t stream_)
{
    if (!t || !active || !offsets) return RVA_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    RVA_HIP(t->ctx, hipEventSynchronize(t->staged));  // previous tick's copies have left the staging buffers
    for (int s = 0; s < t->n_streams; ++s) { t->h_slot[s] = active[s] == 2 ? 2 : (active[s] ? 1 : 0); t->h_offs[s] = offsets[s]; }
    t->h_offs[t->n_streams] = offsets[t->n_streams];
    RVA_HIP(t->ctx, hipMemcpyAsync(t->d_slot, t->h_slot, t->n_streams * 4, hipMemcpyHostToDevice, stream));
    RVA_HIP(t->ctx, hipMemcpyAsync(t->d_offs, t->h_offs, (t->n_streams + 1) * 4, hipMemcpyHostToDevice, stream));
    RVA_HIP(t->ctx, hipEventRecord(t->staged, stream));
    K4Args a{};
    a.slot = t->d_slot; a.offs = t->d_offs; a.boxes64 = boxes; a.conf64 = conf; a.cls64 = cls;
    return launch_update(t, a, true, stream);
}

int rva_tracker_set_box_scale(rva_tracker *t, const double *scales)
{
    if (!t || !scales) return RVA_ERR_ARG;
    RVA_HIP(t->ctx, hipDeviceSynchronize());
    RVA_HIP(t->ctx, hipMemcpy(t->d_bscale, scales, (size_t)t->n_streams * 8, hipMemcpyHostToDevice));
    return RVA_OK;
}

int32_t *rva_tracker_new_counts(rva_tracker *t) { return t ? t->n_new : nullptr; }

int rva_tracker_assign_ids(rva_tracker *t, const int32_t *counts_all, int n_global, const int32_t *global_index,
                           rva_stream_t stream_)
{
    if (!t) return RVA_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    const int32_t *gidx = nullptr;
    if (!counts_all) { counts_all = t->n_new; n_global = t->n_streams; global_index = nullptr; }
    if (n_global <= 0 || n_global > 4096) return rva_fail(t->ctx, RVA_ERR_ARG, "n_global out of range");
    if (global_index) {
        bool same = (int)t->gidx_cached.size() == t->n_streams;
        for (int s = 0; s < t->n_streams; ++s) {
            if (global_index[s] < 0 || global_index[s] >= n_global) return rva_fail(t->ctx, RVA_ERR_ARG, "global_index out of range");
            same = same && t->gidx_cached[s] == global_index[s];
        }
        if (!same) {  // the mapping is fixed for a job: uploaded once, synchronously
            t->gidx_cached.assign(global_index, global_index + t->n_streams);
            RVA_HIP(t->ctx, hipMemcpy(t->d_gidx, global_index, t->n_streams * 4, hipMemcpyHostToDevice));
        }
        gidx = t->d_gidx;
    } else if (n_global != t->n_streams) {
        return rva_fail(t->ctx, RVA_ERR_ARG, "global_index required when n_global != n_streams");
    }
    k4_assign_ids<<<t->n_streams, 64, 0, stream>>>(t->id, t->n_tracks, counts_all, n_global, gidx, t->cap,
                                                   t->next_id, t->id_ticket);
    RVA_HIP(t->ctx, hipGetLastError());
    return RVA_OK;
}

// Snapshot layout inside a pinned slot: fixed offsets, 64-byte aligned sections.
static size_t snap_off(const rva_tracker *t, int section)
{
    const size_t sc = (size_t)t->n_streams * t->cap;
    const size_t sizes[11] = {sc * 8, sc * 32, sc * 8, sc * 4, sc * 4, sc * 4, sc * 4, (size_t)t->n_streams * 4,
                              (size_t)t->n_streams * 4, (size_t)t->n_streams * 4, 8};
    size_t off = 0;
    for (int i = 0; i < section; ++i) off += (sizes[i] + 63) & ~(size_t)63;
    return off;
}

// One launch instead of eight D2H copies: each block (= stream) writes the LIVE rows of its tables straight into the
// pinned, device-mapped snapshot slot (same section layout, rows beyond n_tracks[s] are left untouched).  At ~13-120
// tracks per stream that is tens of KB over PCIe instead of the full S x cap tables (2 MB at cap = 1024), and no
// blit-kernel launches with host-side gaps between them.
struct SnapArgs {
    const int64_t *id; const double *box; const double *conf; const int32_t *cls, *age, *hits, *last_det, *n_tracks;
    int64_t *h_id; double *h_box; double *h_conf; int32_t *h_cls, *h_age, *h_hits, *h_last, *h_n;
    int cap;
    const int32_t *emitted, *processed, *trk_flags, *post_flags;
    int32_t *h_emitted, *h_processed, *h_flags;
};

__global__ void __launch_bounds__(256) k4_snapshot(SnapArgs a)
{
    const int s = blockIdx.x;
    const int n = min(a.n_tracks[s], a.cap);
    if (threadIdx.x == 0) {
        a.h_n[s] = a.n_tracks[s];
        a.h_emitted[s] = a.emitted[s];
        a.h_processed[s] = a.processed[s];
        if (s == 0) { a.h_flags[0] = a.trk_flags[0]; a.h_flags[1] = a.post_flags[0]; }   // overflow / truncation flags ride along
    }
    const size_t base = (size_t)s * a.cap;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const size_t r = base + i;
        a.h_id[r] = a.id[r];
        a.h_conf[r] = a.conf[r];
        a.h_cls[r] = a.cls[r];
        a.h_age[r] = a.age[r];
        a.h_hits[r] = a.hits[r];
        a.h_last[r] = a.last_det[r];
        reinterpret_cast<double2 *>(a.h_box)[2 * r] = reinterpret_cast<const double2 *>(a.box)[2 * r];
        reinterpret_cast<double2 *>(a.h_box)[2 * r + 1] = reinterpret_cast<const double2 *>(a.box)[2 * r + 1];
    }
}

int rva_tracker_snapshot_async(rva_tracker *t, int slot, rva_stream_t stream_)
{
    if (!t || slot < 0 || slot >= RVA_SNAPSHOT_SLOTS) return RVA_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    char *h = (char *)t->h_read_dev[slot];
    SnapArgs a{t->id, t->box, t->conf, t->cls, t->age, t->hits, t->last_det, t->n_tracks,
               (int64_t *)(h + snap_off(t, 0)), (double *)(h + snap_off(t, 1)), (double *)(h + snap_off(t, 2)),
               (int32_t *)(h + snap_off(t, 3)), (int32_t *)(h + snap_off(t, 4)), (int32_t *)(h + snap_off(t, 5)),
               (int32_t *)(h + snap_off(t, 6)), (int32_t *)(h + snap_off(t, 7)), t->cap,
               t->emitted, t->processed, t->flags, t->ctx->post_flags,
               (int32_t *)(h + snap_off(t, 8)), (int32_t *)(h + snap_off(t, 9)), (int32_t *)(h + snap_off(t, 10))};
    k4_snapshot<<<t->n_streams, 256, 0, stream>>>(a);
    RVA_HIP(t->ctx, hipGetLastError());
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    if (cap == hipStreamCaptureStatusNone)   // inside a graph capture the caller synchronises on the replay instead
        RVA_HIP(t->ctx, hipEventRecord(t->snap_done[slot], stream));
    return RVA_OK;
}

int rva_tracker_snapshot_fetch(rva_tracker *t, int slot, int wait, int64_t *ids, int32_t *cls, int32_t *age,
                               int32_t *hits, double *conf, double *boxes, int32_t *last_det, int32_t *counts)
{
    if (!t || slot < 0 || slot >= RVA_SNAPSHOT_SLOTS) return RVA_ERR_ARG;
    if (wait) RVA_HIP(t->ctx, hipEventSynchronize(t->snap_done[slot]));
    const char *h = (const char *)t->h_read[slot];
    const int32_t *hn = (const int32_t *)(h + snap_off(t, 7));
    if (counts) std::memcpy(counts, hn, (size_t)t->n_streams * 4);
    // only the live rows of every stream exist in the slot (k4_snapshot): copy those, leave the rest of the caller's
    // [S][cap] arrays untouched
    for (int s = 0; s < t->n_streams; ++s) {
        const size_t n = (size_t)(hn[s] < t->cap ? (hn[s] > 0 ? hn[s] : 0) : t->cap), r = (size_t)s * t->cap;
        if (!n) continue;
        if (ids) std::memcpy(ids + r, h + snap_off(t, 0) + r * 8, n * 8);
        if (boxes) std::memcpy(boxes + r * 4, h + snap_off(t, 1) + r * 32, n * 32);
        if (conf) std::memcpy(conf + r, h + snap_off(t, 2) + r * 8, n * 8);
        if (cls) std::memcpy(cls + r, h + snap_off(t, 3) + r * 4, n * 4);
        if (age) std::memcpy(age + r, h + snap_off(t, 4) + r * 4, n * 4);
        if (hits) std::memcpy(hits + r, h + snap_off(t, 5) + r * 4, n * 4);
        if (last_det) std::memcpy(last_det + r, h + snap_off(t, 6) + r * 4, n * 4);
    }
    return RVA_OK;
}

int rva_tracker_snapshot_status(rva_tracker *t, int slot, int32_t *emitted, int32_t *processed, int32_t *flags)
{
    if (!t || slot < 0 || slot >= RVA_SNAPSHOT_SLOTS) return RVA_ERR_ARG;
    const char *h = (const char *)t->h_read[slot];
    if (emitted) std::memcpy(emitted, h + snap_off(t, 8), (size_t)t->n_streams * 4);
    if (processed) std::memcpy(processed, h + snap_off(t, 9), (size_t)t->n_streams * 4);
    if (flags) {
        const int32_t *f = (const int32_t *)(h + snap_off(t, 10));
        *flags = (f[0] & 0xff) | ((f[1] & 0xff) << 8);
    }
    return RVA_OK;
}

int rva_tracker_read_all(rva_tracker *t, int64_t *ids, int32_t *cls, int32_t *age, int32_t *hits, double *conf,
                         double *boxes, int32_t *last_det, int32_t *counts, rva_stream_t stream_)
{
    int rc = rva_tracker_snapshot_async(t, 0, stream_);
    if (rc != RVA_OK) return rc;
    return rva_tracker_snapshot_fetch(t, 0, 1, ids, cls, age, hits, conf, boxes, last_det, counts);
}

int rva_tracker_read(rva_tracker *t, int stream_id, int cap, int64_t *ids, int32_t *cls, int32_t *age, int32_t *hits,
                     double *conf, double *boxes, int32_t *last_det, int32_t *n, rva_stream_t stream_)
{
    if (!t || stream_id < 0 || stream_id >= t->n_streams || !n || cap < 0) return RVA_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    int32_t cnt = 0;
    RVA_HIP(t->ctx, hipMemcpyAsync(&cnt, t->n_tracks + stream_id, 4, hipMemcpyDeviceToHost, stream));
    RVA_HIP(t->ctx, hipStreamSynchronize(stream));
    *n = cnt;
    const int m = cnt < cap ? cnt : cap;
    if (m <= 0) return RVA_OK;
    const size_t tb = (size_t)stream_id * t->cap;
    if (ids) RVA_HIP(t->ctx, hipMemcpyAsync(ids, t->id + tb, (size_t)m * 8, hipMemcpyDeviceToHost, stream));
    if (boxes) RVA_HIP(t->ctx, hipMemcpyAsync(boxes, t->box + tb * 4, (size_t)m * 32, hipMemcpyDeviceToHost, stream));
    if (conf) RVA_HIP(t->ctx, hipMemcpyAsync(conf, t->conf + tb, (size_t)m * 8, hipMemcpyDeviceToHost, stream));
    if (cls) RVA_HIP(t->ctx, hipMemcpyAsync(cls, t->cls + tb, (size_t)m * 4, hipMemcpyDeviceToHost, stream));
    if (age) RVA_HIP(t->ctx, hipMemcpyAsync(age, t->age + tb, (size_t)m * 4, hipMemcpyDeviceToHost, stream));
    if (hits) RVA_HIP(t->ctx, hipMemcpyAsync(hits, t->hits + tb, (size_t)m * 4, hipMemcpyDeviceToHost, stream));
    if (last_det) RVA_HIP(t->ctx, hipMemcpyAsync(last_det, t->last_det + tb, (size_t)m * 4, hipMemcpyDeviceToHost, stream));
    RVA_HIP(t->ctx, hipStreamSynchronize(stream));
    return RVA_OK;
}

int rva_tracker_state(rva_tracker *t, int64_t *next_id, int *flags, rva_stream_t stream_)
{
    if (!t) return RVA_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    int64_t nid = 0;
    int32_t fl = 0;
    RVA_HIP(t->ctx, hipMemcpyAsync(&nid, t->next_id, 8, hipMemcpyDeviceToHost, stream));
    RVA_HIP(t->ctx, hipMemcpyAsync(&fl, t->flags, 4, hipMemcpyDeviceToHost, stream));
    RVA_HIP(t->ctx, hipStreamSynchronize(stream));
    if (next_id) *next_id = nid;
    if (flags) *flags = fl;
    return RVA_OK;
}

int rva_tracker_set_next_id(rva_tracker *t, int64_t next_id, rva_stream_t stream_)
{
    if (!t) return RVA_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    RVA_HIP(t->ctx, hipMemcpyAsync(t->next_id, &next_id, 8, hipMemcpyHostToDevice, stream));
    RVA_HIP(t->ctx, hipStreamSynchronize(stream));
    return RVA_OK;
}

}  // extern "C"
