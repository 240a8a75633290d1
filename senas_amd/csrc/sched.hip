// Lane scheduler: replays a CAPTURED multi-stream HIP graph as linear segments on streams of its own.
//
// Why this exists.  The step drivers capture forward + backward of a network whose macro-grid columns run on several HIP
// streams (senas_amd/grid.py: Lanes), so the captured graph is a DAG with ~5 branches in flight.  The runtime's own graph
// executor (ROCm 7.0 libamdhip64 under this torch build) does not run such a DAG well: it re-derives a node -> stream
// assignment of its own by a depth-first walk, in which every branch forked off the origin stream lands on the same
// internal stream (measured: 2 graph queues give the same step time as 8), and `hip::Graph::UpdateStreams` indexes its
// stream table out of bounds for some fork / join shapes (SIGSEGV inside hipGraphLaunch: profiles/r4_graph_executor.txt).
// A linear graph on one stream is the path of that executor that has replayed this package's steps since round 1.
//
// So: take the captured hipGraph_t (never instantiated), read its nodes and edges, cut it into linear pieces ("segments"),
// rebuild every piece as a single-branch graph (kernel nodes re-added from their own parameters, 1-D memsets as a fill kernel
// of this library, a memcpy node as the one survivor of a clone of the captured graph; each node depending on its predecessor
// only), and at launch time issue the pieces in a topological order on a few streams (one hardware queue each) with an event
// per dependency that crosses streams.  Memory safety is the capture's (torch's caching allocator saw every lane as a stream
// of its own).
//
// Two policies decide which piece runs where (SENAS_SCHED_POLICY):
//   "critical" (default, round 5): the pieces are the maximal linear runs of a path cover of the DAG; every piece is TIMED once
//       (a serial replay inside senas_sched_create), and a list scheduler deals them to the streams in the order a simulation of
//       the streams starts them -- whenever a stream falls idle it takes, among the pieces whose dependencies have finished, the
//       one with the longest remaining path to the end of the pass.  The pass's critical path (head -> column 0 -> first down
//       cell) is then never queued behind a weight-gradient batch or a small cell that happens to share its stream.
//   "chain" (round 4): cover the DAG with at most L chains ("lanes") greedily at node level, cut the chains where a dependency
//       crosses lanes, lanes pinned to streams by node count.
// Dependencies are the captured ones, with one exception that is the point of the typed markers (below): a reader behind a
// CONSUMER marker depends on the PRODUCER-marked parents of the RELAY marker it waited for, not on the rest of the origin stream's
// history.
#include "common.h"

#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <queue>
#include <unordered_map>
#include <vector>

namespace senas {

struct Segment {
    int lane = 0;                // the stream (index into Sched::lanes) this piece is launched on
    std::vector<int> nodes;      // indices into Sched::node (topological positions)
    std::vector<int> deps;       // segments whose `done` event this one waits for (those on its own stream need no event)
    bool signals = false;        // somebody on another stream waits for it
    float dur_us = 0.f;          // measured by the serial timing replay of senas_sched_create ("critical" policy)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEvent_t done = nullptr;
};

struct Sched {
    std::vector<Segment> segs;
    std::vector<int> issue;              // the order the segments are launched in (a topological order of the segments)
    int policy = 1;                      // 1 critical, 0 chain
    std::vector<hipStream_t> lanes;      // from a process-wide pool (one lane only: empty, everything runs on the caller's stream)
    std::vector<hipEvent_t> lane_done;
    hipEvent_t start = nullptr;
    int n_nodes = 0, n_lanes = 0, n_cross = 0, n_kernel = 0, n_memset = 0, n_memcpy = 0, n_empty = 0, n_marker = 0, n_captured = 0;
};

static void sched_free(Sched* s) {
    if (!s) return;
    for (auto& g : s->segs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
        if (g.done) (void)hipEventDestroy(g.done);
    }
    for (auto e : s->lane_done) if (e) (void)hipEventDestroy(e);
    if (s->start) (void)hipEventDestroy(s->start);
    delete s;
}

// Lane streams are shared by every scheduler of a device (the two passes of a search step, successive step drivers) and are
// chosen so that no two of them sit on one hardware queue.  The runtime runs GPU_MAX_HW_QUEUES (4) hardware queues per process
// and deals streams onto them by load; two lanes on one queue run one after the other however independent their work is, and
// more than four busy queues made every step slower (measured: profiles/r4_lanes_queues.txt).  So the pool is built by
// measurement: candidate streams are created one by one and a candidate is kept only if a 150 us spin kernel on it overlaps
// with the same kernel on every stream already kept.  Never destroyed (process lifetime).
__global__ void lane_probe_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}

static bool lanes_overlap(hipStream_t a, hipStream_t b) {
    // two 150 us spins: ~150 us when the streams run side by side, ~300 us when they share a queue
    const long long ticks = 15000;                                   // 100 MHz clock
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        hipEvent_t e0 = nullptr, e1 = nullptr, eb = nullptr;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventCreateWithFlags(&eb, hipEventDisableTiming);
        (void)hipEventRecord(e0, a);
        hipLaunchKernelGGL(lane_probe_kernel, dim3(1), dim3(1), 0, a, ticks);
        hipLaunchKernelGGL(lane_probe_kernel, dim3(1), dim3(1), 0, b, ticks);
        (void)hipEventRecord(eb, b);
        (void)hipStreamWaitEvent(a, eb, 0);
        (void)hipEventRecord(e1, a);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, (double)ms);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipEventDestroy(eb);
    }
    return best < 0.225;
}

static std::vector<hipStream_t>& lane_pool() {
    static std::vector<hipStream_t> pool[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return pool[dev & 63];
}

// Stream priorities -- measured and NOT used (profiles/r5_sched_policies.txt).  The idea: streams 0 and 1 of the pool at the device's
// greatest priority for the light segments (launch-bound kernels of a few workgroups: the small-map cells, which are the critical
// path of a pass in both directions), the default priority for the heavy ones (kernels that fill the chip), the two classes
// never queueing behind each other.  The measurement: on this runtime a kernel chain on a high-priority stream runs 3.5x SLOWER
// (down1 forward: 372 -> 1 203 us; the search step 27.2 -> 59.3 ms, the train step 15.7 -> 27.4) -- whatever the queue
// arbitration does with the priority, every launch of such a queue pays for it.  SENAS_SCHED_PRIORITY=1 turns it on again
// (the classes and the priorities together) for a re-measurement on another runtime.
// SENAS_SCHED_PRIORITY=2: the other way round -- the light streams at the default priority, the HEAVY ones at the device's least.
// SENAS_SCHED_PRIORITY=3: no priorities; the heavy streams are created with a CU mask that leaves every 8th CU to the light ones
// (hipExtStreamCreateWithCUMask): a light kernel then always finds free CUs beside a kernel that fills the rest of the chip.
static int pool_priority_mode() {
    const char* e = getenv("SENAS_SCHED_PRIORITY");
    return e ? (e[0] == '1' ? 1 : (e[0] == '2' ? 2 : (e[0] == '3' ? 3 : (e[0] == '4' ? 4 : 0)))) : 0;      // (4: the two classes on plain streams)
}
static bool pool_priorities() { return pool_priority_mode() != 0; }
constexpr int kHighStreams = 2;

// grow the pool to `want` streams on distinct hardware queues (fewer if the device does not give that many)
static void lane_pool_grow(int want) {
    auto& pool = lane_pool();
    std::vector<hipStream_t> rejected;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    for (int tries = 0; (int)pool.size() < want && tries < 16; ++tries) {
        hipStream_t st = nullptr;
        const int mode = pool_priority_mode();
        const bool light_stream = (int)pool.size() < kHighStreams;
        const int prio = mode == 1 ? (light_stream ? greatest : 0) : (mode == 2 ? (light_stream ? 0 : least) : 0);
        if (getenv("SENAS_SCHED_VERBOSE") && pool.empty()) fprintf(stderr, "[sched] stream priorities of this device: least %d, greatest %d\n", least, greatest);
        if (mode == 3 && !light_stream) {
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
            std::vector<uint32_t> mask((cus + 31) / 32, 0u);
            int keep_every = 8;
            if (const char* e = getenv("SENAS_SCHED_CUMASK_EVERY")) keep_every = atoi(e) > 1 ? atoi(e) : 8;
            for (int cu = 0; cu < cus; ++cu) if (cu % keep_every != keep_every - 1) mask[cu >> 5] |= 1u << (cu & 31);
            if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) break;
        } else if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) != hipSuccess) break;
        bool ok = true;
        for (hipStream_t kept : pool) ok = ok && lanes_overlap(kept, st);
        if (ok) pool.push_back(st); else rejected.push_back(st);      // (kept alive until the end: a destroyed stream's queue slot is the next one handed out)
    }
    for (hipStream_t st : rejected) (void)hipStreamDestroy(st);
    (void)hipGetLastError();
}

// ---- the plan: host arithmetic only (senas_sched_plan exposes it to the CPU tests) -----------------------------------------
// Nodes are numbered in a topological order (every parent before its children).  Chain cover with at most L chains ("lanes"):
// a node continues the lane of the latest parent that still ends its lane, else takes an unused lane, else the lane whose tail has
// nothing left to feed and has idled longest, else the lane idle longest.  Lane 0 is the lane of node 0.  Segments: cut before a
// node with a parent on another lane and after a node with a child on another lane; a `solo` node is a segment of its own.  A
// segment waits for the LATEST segment of every other lane that holds a parent of its first node (stream order covers the earlier
// ones).  Segment indices are the issue order and a topological order of the segments.
struct PlanSeg {
    int lane = 0;
    std::vector<int> nodes, deps;
    bool signals = false;
};

static void plan_schedule(int n, const std::vector<std::vector<int>>& par, const std::vector<std::vector<int>>& chi, const std::vector<char>& solo,
                          int L, std::vector<int>& lane, std::vector<int>& seg_of, std::vector<PlanSeg>& segs, int& used, int& n_cross) {
    lane.assign(n, -1);
    std::vector<int> tail(L, -1), placed_children(n, 0), last_use(L, -1);
    for (int v = 0; v < n; ++v) {
        int best = -1;
        for (int p : par[v])                                   // continue the lane of a parent that still ends its lane
            if (tail[lane[p]] == p && (best < 0 || p > best)) best = p;
        int l;
        if (best >= 0) {
            l = lane[best];
        } else {
            l = -1;
            for (int q = 0; q < L && l < 0; ++q) if (tail[q] < 0) l = q;                       // an unused lane
            if (l < 0) {                                                                       // a lane whose tail has nothing left to feed, idle longest
                for (int q = 0; q < L; ++q) {
                    const bool dead = placed_children[tail[q]] == (int)chi[tail[q]].size();
                    if (dead && (l < 0 || last_use[q] < last_use[l])) l = q;
                }
            }
            if (l < 0) { l = 0; for (int q = 1; q < L; ++q) if (last_use[q] < last_use[l]) l = q; }
        }
        lane[v] = l;
        tail[l] = v;
        last_use[l] = v;
        for (int p : par[v]) ++placed_children[p];
    }
    // lane 0 = the lane of the first node (the caller's stream carries what the capture's origin stream started with)
    if (n > 0 && lane[0] != 0) { const int a = lane[0]; for (auto& x : lane) x = (x == a ? 0 : (x == 0 ? a : x)); }
    used = 0;
    for (int v = 0; v < n; ++v) used = std::max(used, lane[v] + 1);
    seg_of.assign(n, -1);
    segs.clear();
    std::vector<int> open(used, -1);
    for (int v = 0; v < n; ++v) {
        const int l = lane[v];
        bool waits = false, feeds = false;
        for (int p : par[v]) waits |= lane[p] != l;
        for (int c : chi[v]) feeds |= lane[c] != l;
        if (waits || solo[v] || open[l] < 0) {
            segs.emplace_back();
            segs.back().lane = l;
            open[l] = (int)segs.size() - 1;
        }
        PlanSeg& sg = segs[open[l]];
        sg.nodes.push_back(v);
        seg_of[v] = open[l];
        if (feeds) sg.signals = true;
        if (feeds || solo[v]) open[l] = -1;
    }
    n_cross = 0;
    for (auto& sg : segs) {
        std::vector<int> latest(used, -1);                     // per source lane only the latest segment matters
        for (int p : par[sg.nodes[0]])
            if (lane[p] != sg.lane) { latest[lane[p]] = std::max(latest[lane[p]], seg_of[p]); ++n_cross; }
        for (int q = 0; q < used; ++q) if (latest[q] >= 0) sg.deps.push_back(latest[q]);
    }
}

// ---- typed contraction (host arithmetic; senas_sched_contract exposes it to the CPU tests) -----------------------------------
// kind[v]: -1 a node that runs; 0 RELAY, 1 PRODUCER, 2 CONSUMER marker; 3 an empty node.  Nodes numbered topologically.  On
// return par[] of the surviving nodes holds surviving nodes only (chi[] is rebuilt from it); the others have par[] cleared.
static void contract_markers(int n, std::vector<std::vector<int>>& par, std::vector<std::vector<int>>& chi, const std::vector<int>& kind) {
    std::vector<std::vector<int>> via(n);                    // via[v]: the parents v inherited through a PRODUCER marker
    auto add = [](std::vector<int>& to, int x) { if (std::find(to.begin(), to.end(), x) == to.end()) to.push_back(x); };
    for (int i = 0; i < n; ++i) {
        if (kind[i] < 0) continue;
        // (topological order: every parent of i that was a marker is gone already, so par[i] holds running nodes only)
        for (int c : chi[i]) {
            auto& cp = par[c];
            cp.erase(std::remove(cp.begin(), cp.end(), i), cp.end());
            const bool narrow = kind[i] == 0 && kind[c] == 2 && !via[i].empty();      // RELAY -> CONSUMER: the hand-over's source only
            for (int p : (narrow ? via[i] : par[i])) add(cp, p);
            if (kind[i] == 1) for (int p : par[i]) add(via[c], p);
        }
        par[i].clear();
    }
    for (int v = 0; v < n; ++v) chi[v].clear();
    for (int v = 0; v < n; ++v) for (int p : par[v]) chi[p].push_back(v);
}

// ---- the "critical" plan (host arithmetic; senas_sched_plan2 exposes it to the CPU tests) ----------------------------------
// 1. Path cover: in topological order a node becomes the heir of its latest parent that has no heir yet.  2. Segments: the
// maximal runs of a path in which no node but the first has a parent outside the run and no node but the last a child that is not
// its heir (a `solo` node is a run of its own) -- so every parent of a segment's first node is the LAST node of its segment and
// "wait for that segment" is exact.  Segments are numbered by their first node: the numbering is a topological order.
struct Piece {
    std::vector<int> nodes, deps, kids;
};

static void cut_pieces(int n, const std::vector<std::vector<int>>& par, const std::vector<std::vector<int>>& chi, const std::vector<char>& solo,
                       std::vector<int>& seg_of, std::vector<Piece>& segs) {
    std::vector<int> pred(n, -1), heir(n, -1);
    for (int v = 0; v < n; ++v) {
        if (solo[v]) continue;
        int best = -1;
        for (int p : par[v]) if (heir[p] < 0 && !solo[p] && p > best) best = p;
        if (best >= 0) { pred[v] = best; heir[best] = v; }
    }
    seg_of.assign(n, -1);
    segs.clear();
    for (int v = 0; v < n; ++v) {
        bool fresh = pred[v] < 0;
        if (!fresh) {
            const int p = pred[v];
            for (int c : chi[p]) fresh |= c != v;                                   // the predecessor feeds somebody else too: it ends its segment
            for (int q : par[v]) fresh |= q != p && seg_of[q] != seg_of[p];        // a parent outside the open segment: a wait in front of v
        }
        if (fresh) {
            segs.emplace_back();
            seg_of[v] = (int)segs.size() - 1;
            for (int q : par[v]) {
                auto& d = segs.back().deps;
                if (std::find(d.begin(), d.end(), seg_of[q]) == d.end()) d.push_back(seg_of[q]);
            }
        } else {
            seg_of[v] = seg_of[pred[v]];
        }
        segs[seg_of[v]].nodes.push_back(v);
    }
    for (size_t k = 0; k < segs.size(); ++k) for (int d : segs[k].deps) segs[d].kids.push_back((int)k);
}

// 3. List scheduling on S streams with the measured durations: a simulation in which, whenever a stream is idle, it takes --
// among the segments whose dependencies have FINISHED -- the one with the longest remaining path (its own duration + the longest
// chain of dependants); an idle stream that ran one of the segment's dependencies is preferred (no event, no cross-queue
// latency).  `stream[k]`, and `issue` = the segments in the order the simulation starts them (topological; per stream it is the
// stream's FIFO order).
// `light[k]` / `high` (optional): a light segment only goes to streams [0, high), a heavy one only to [high, S) -- the two classes
// never queue behind each other (high == 0 or high >= S: one class).
static void list_schedule(const std::vector<Piece>& segs, const std::vector<double>& dur, int S, std::vector<int>& stream, std::vector<int>& issue,
                          const std::vector<char>* light = nullptr, int high = 0) {
    const int K = (int)segs.size();
    const bool classes = light != nullptr && high > 0 && high < S;
    auto fits = [&](int k, int q) { return !classes || ((*light)[k] ? q < high : q >= high); };
    std::vector<double> bottom(K, 0.0), finish(K, 0.0);
    for (int k = K - 1; k >= 0; --k) {
        double b = 0.0;
        for (int c : segs[k].kids) b = std::max(b, bottom[c]);
        bottom[k] = b + std::max(dur[k], 1e-3);
    }
    std::vector<int> waiting(K), state(K, 0);                  // state: 0 not ready, 1 ready, 2 started
    for (int k = 0; k < K; ++k) { waiting[k] = (int)segs[k].deps.size(); if (!waiting[k]) state[k] = 1; }
    std::vector<double> free_at(S, 0.0);
    std::vector<int> running(S, -1);
    stream.assign(K, 0);
    issue.clear();
    double t = 0.0;
    int started = 0;
    while (started < K) {
        // retire what has finished by t
        for (int q = 0; q < S; ++q)
            if (running[q] >= 0 && free_at[q] <= t) {
                for (int c : segs[running[q]].kids) if (--waiting[c] == 0) state[c] = 1;
                running[q] = -1;
            }
        for (;;) {
            // the ready segment with the longest remaining path among those an idle stream of their class can take
            int best = -1, q = -1;
            for (int k = 0; k < K; ++k) {
                if (state[k] != 1 || (best >= 0 && bottom[k] <= bottom[best])) continue;
                int mine = -1;
                double latest = -1.0;
                for (int d : segs[k].deps)                     // an idle stream that ran a dependency (the one that finished last)
                    if (running[stream[d]] < 0 && fits(k, stream[d]) && finish[d] > latest) { latest = finish[d]; mine = stream[d]; }
                for (int r = 0; r < S && mine < 0; ++r) if (running[r] < 0 && fits(k, r)) mine = r;
                if (mine >= 0) { best = k; q = mine; }
            }
            if (best < 0) break;                               // nothing is ready, or no stream of the ready segments' class is idle
            stream[best] = q;
            running[q] = best;
            state[best] = 2;
            finish[best] = free_at[q] = t + std::max(dur[best], 1e-3);
            issue.push_back(best);
            ++started;
        }
        double next = 1e300;
        for (int q = 0; q < S; ++q) if (running[q] >= 0) next = std::min(next, free_at[q]);
        if (next < 1e300) { t = next; continue; }
        // nothing runs and nothing is ready although segments remain: only a malformed dependency list gets here -- issue the
        // rest in their numbering (a topological order) on stream 0 rather than spin
        for (int k = 0; k < K; ++k) if (state[k] != 2) { state[k] = 2; stream[k] = 0; issue.push_back(k); ++started; }
    }
}

// How many lanes a captured pass may be spread over.  The schedule is tuned to the runtime's default of FOUR hardware queues per
// process.  Measured (profiles/r5_queue_guard.txt; search step, ms): default 28.5; GPU_MAX_HW_QUEUES=6 with lanes 43.7, =8 54.7
// (round 4) against 34.9 on one stream -- MORE queues than four make the step slower than no lanes at all; =3 with lanes 28.8, =2
// 31.3 -- FEWER queues still beat one stream (the lanes share the queues there are).  So: an override above 4 keeps the serial
// schedule, one line on stderr, never a slower step; below 4 the lanes run on what the probe finds.  No distinct second queue at
// all: serial.  SENAS_SCHED_TRUST_QUEUES=1 skips the guard (measurement).
static int lanes_allowed(int want) {
    if (want <= 1) return want;
    if (const char* t = getenv("SENAS_SCHED_TRUST_QUEUES")) if (t[0] == '1') return want;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    static int warned = 0;
    if (const char* q = getenv("GPU_MAX_HW_QUEUES")) {
        const int v = atoi(q);
        if (v > 4) {
            if (!warned++) fprintf(stderr, "[senas sched] GPU_MAX_HW_QUEUES=%d (more hardware queues than the default 4 make the lane schedule slower than one stream): captured passes keep the serial schedule\n", v);
            return 1;
        }
    }
    lane_pool_grow(2);
    const int have = (int)lane_pool().size();
    if (have < 2) {
        if (!warned++) fprintf(stderr, "[senas sched] no second hardware queue: captured passes keep the serial schedule\n");
        return 1;
    }
    return want;
}

#define SCHED_HIP(call, what)                                                   \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) { set_error(what, e__); sched_free(S); return SENAS_ELAUNCH; } \
    } while (0)

}  // namespace senas

// A hand-over between two lanes passes through the capture's origin stream (grid.Lanes: the star topology).  The origin stream
// records no kernel between two hand-overs, so the runtime's capture bookkeeping makes every hand-over depend on the producers
// of all earlier ones.  A marker launched on the origin stream at each hand-over gives that chain nodes the scheduler can
// recognise and contract away.  CUTTING the chain at the markers (marker -> marker edges dropped: tried in round 4) would
// leave consumer <- producer, the dependency that was meant -- but a marker also absorbs every wait the origin stream itself
// made since its last kernel (autograd's hand-overs TO the origin stream, join_lanes), and the origin's next real kernel
// reaches those only through the chain: the cut loses them (measured: dirty weight gradients of the first down cell).  So the
// chain is kept; telling the two kinds of children of a marker apart needs a marker on the consumer side too (not built).
// Kinds (senas_marker): 0 RELAY -- on the capture's origin stream at a hand-over; 1 PRODUCER -- on the lane that made the tensor
// (the gradient, on the way back), in front of the event the origin stream waits for; 2 CONSUMER -- on the lane that reads it,
// behind its wait for the origin stream.  A RELAY marker R has parents {origin stream's previous node and whatever else that
// stream had waited for, PRODUCER markers}; its children are CONSUMER markers (the readers) and the origin stream's next node.
// Contraction: a CONSUMER child inherits only what R's PRODUCER parents stand for; any other child inherits everything (the
// origin stream's own chain keeps every wait it ever made -- the cut that lost some of them in round 4 is not made).
__global__ void relay_marker_kernel(int kind) {}

// A captured 1-D memset as a kernel: count elements of esz bytes (1, 2 or 4) set to the low esz bytes of value; dst is aligned
// to esz (it is an array of such elements), so every aligned 32-bit word holds whole elements and takes the replicated pattern.
__global__ void sched_fill_kernel(void* dst, unsigned long long count, unsigned value, int esz) {
    const unsigned long long bytes = count * (unsigned long long)esz;
    unsigned pattern = value;
    if (esz == 1) { pattern &= 0xffu; pattern |= pattern << 8; pattern |= pattern << 16; }
    else if (esz == 2) { pattern &= 0xffffu; pattern |= pattern << 16; }
    unsigned char* base = reinterpret_cast<unsigned char*>(dst);
    unsigned long long head = (4 - (reinterpret_cast<uintptr_t>(base) & 3)) & 3;      // bytes in front of the first aligned word
    if (head > bytes) head = bytes;
    const unsigned long long words = (bytes - head) / 4;
    const unsigned long long tail0 = head + words * 4;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned* w = reinterpret_cast<unsigned*>(base + head);
    for (unsigned long long i = tid; i < words; i += stride) w[i] = pattern;
    for (unsigned long long i = tid; i < head; i += stride) base[i] = (unsigned char)(value >> (8 * (i % esz)));
    for (unsigned long long i = tail0 + tid; i < bytes; i += stride) base[i] = (unsigned char)(value >> (8 * (i % esz)));
}

using namespace senas;

extern "C" int senas_stream_create(void** out) {
    SENAS_REQUIRE(out != nullptr, "stream_create: bad argument");
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) { set_error("hipStreamCreateWithFlags", e); return SENAS_ELAUNCH; }
    *out = st;
    return SENAS_OK;
}

// One thread writes the device's constant-rate wall clock (100 MHz) into *slot: a time stamp IN stream order -- under HIP-graph
// replay across several streams the only timeline that shows what really overlaps (a tracing profiler serialises the queues).
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

extern "C" int senas_stamp(uint64_t* slot, void* stream) {
    SENAS_REQUIRE(slot != nullptr, "stamp: bad argument");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, as_stream(stream), reinterpret_cast<unsigned long long*>(slot));
    return launch_status("stamp");
}

extern "C" int senas_marker(int kind, void* stream) {
    SENAS_REQUIRE(kind >= 0 && kind <= 2, "marker: kind must be 0 (relay), 1 (producer) or 2 (consumer)");
    hipLaunchKernelGGL(relay_marker_kernel, dim3(1), dim3(1), 0, as_stream(stream), kind);
    return launch_status("marker");
}

extern "C" int senas_relay_marker(void* stream) { return senas_marker(0, stream); }

extern "C" int senas_sched_create(void* hip_graph, int max_lanes, void** out) {
    SENAS_REQUIRE(hip_graph && out && max_lanes >= 1 && max_lanes <= 16, "sched_create: bad argument");
    *out = nullptr;
    hipGraph_t G = reinterpret_cast<hipGraph_t>(hip_graph);
    Sched* S = new Sched();
    size_t n = 0, m = 0;
    SCHED_HIP(hipGraphGetNodes(G, nullptr, &n), "hipGraphGetNodes");
    if (n == 0) { set_error_msg("sched_create: the graph has no nodes"); sched_free(S); return SENAS_EINVAL; }
    std::vector<hipGraphNode_t> raw(n);
    SCHED_HIP(hipGraphGetNodes(G, raw.data(), &n), "hipGraphGetNodes");
    SCHED_HIP(hipGraphGetEdges(G, nullptr, nullptr, &m), "hipGraphGetEdges");
    std::vector<hipGraphNode_t> ef(m), et(m);
    if (m) SCHED_HIP(hipGraphGetEdges(G, ef.data(), et.data(), &m), "hipGraphGetEdges");
    std::unordered_map<hipGraphNode_t, int> index;
    for (size_t i = 0; i < n; ++i) index[raw[i]] = (int)i;
    std::vector<std::vector<int>> par0(n), chi0(n);
    for (size_t e = 0; e < m; ++e) {
        auto a = index.find(ef[e]), b = index.find(et[e]);
        if (a == index.end() || b == index.end()) { set_error_msg("sched_create: an edge names a node the graph does not list"); sched_free(S); return SENAS_EINVAL; }
        par0[b->second].push_back(a->second);
        chi0[a->second].push_back(b->second);
    }
    // ---- topological order of everything captured, ties broken by the runtime's own node order (creation order)
    std::vector<int> topo, pos(n);
    {
        std::vector<int> indeg(n);
        for (size_t i = 0; i < n; ++i) indeg[i] = (int)par0[i].size();
        std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
        for (size_t i = 0; i < n; ++i) if (!indeg[i]) ready.push((int)i);
        while (!ready.empty()) {
            const int v = ready.top();
            ready.pop();
            pos[v] = (int)topo.size();
            topo.push_back(v);
            for (int c : chi0[v]) if (--indeg[c] == 0) ready.push(c);
        }
        if (topo.size() != n) { set_error_msg("sched_create: the captured graph has a cycle"); sched_free(S); return SENAS_EINVAL; }
    }
    // ---- markers and empty nodes: contracted out of the graph (contract_markers: a removed node hands its parents to its
    // children; a CONSUMER marker behind a RELAY marker only takes what the relay's PRODUCER markers stand for)
    {
        std::vector<std::vector<int>> par(n), chi(n);
        std::vector<int> kind(n, -1);
        for (size_t i = 0; i < n; ++i) {
            for (int p : par0[i]) par[pos[i]].push_back(pos[p]);
            for (int c : chi0[i]) chi[pos[i]].push_back(pos[c]);
            hipGraphNodeType t;
            SCHED_HIP(hipGraphNodeGetType(raw[i], &t), "hipGraphNodeGetType");
            if (t == hipGraphNodeTypeEmpty) { kind[pos[i]] = 3; ++S->n_empty; }
            if (t == hipGraphNodeTypeKernel) {
                hipKernelNodeParams p;
                SCHED_HIP(hipGraphKernelNodeGetParams(raw[i], &p), "hipGraphKernelNodeGetParams");
                if (p.func == reinterpret_cast<void*>(relay_marker_kernel)) {
                    int k = 0;
                    if (p.kernelParams && p.kernelParams[0]) k = *reinterpret_cast<int*>(p.kernelParams[0]);
                    kind[pos[i]] = (k >= 0 && k <= 2) ? k : 0;
                    ++S->n_marker;
                }
            }
        }
        size_t edges_before = 0, edges_after = 0;
        for (size_t v = 0; v < n; ++v) edges_before += par[v].size();
        contract_markers((int)n, par, chi, kind);
        // compact: only the surviving nodes take part from here on, numbered in the same topological order
        std::vector<int> newidx(n, -1);
        std::vector<hipGraphNode_t> kept;
        for (size_t v = 0; v < n; ++v) if (kind[v] < 0) { newidx[v] = (int)kept.size(); kept.push_back(raw[topo[v]]); }
        if (kept.empty()) { set_error_msg("sched_create: the graph has no nodes to run"); sched_free(S); return SENAS_EINVAL; }
        par0.assign(kept.size(), {});
        chi0.assign(kept.size(), {});
        for (size_t v = 0; v < n; ++v) {
            if (kind[v] >= 0) continue;
            for (int p : par[v]) { par0[newidx[v]].push_back(newidx[p]); chi0[newidx[p]].push_back(newidx[v]); ++edges_after; }
        }
        if (getenv("SENAS_SCHED_VERBOSE"))
            fprintf(stderr, "[sched] %zu captured nodes, %zu edges; %d markers and %d empty nodes contracted: %zu nodes, %zu edges\n", n, edges_before,
                    S->n_marker, S->n_empty, kept.size(), edges_after);
        S->n_captured = (int)n;
        raw.swap(kept);
        n = raw.size();
    }
    // from here on a node is its (compacted) topological position; raw[v] is its captured node
    std::vector<std::vector<int>>& par = par0;
    std::vector<std::vector<int>>& chi = chi0;
    std::vector<hipGraphNodeType> type(n);
    for (size_t v = 0; v < n; ++v) SCHED_HIP(hipGraphNodeGetType(raw[v], &type[v]), "hipGraphNodeGetType");

    // ---- the plan (host arithmetic: plan_schedule / cut_pieces + list_schedule above)
    std::vector<char> solo(n, 0);
    for (size_t v = 0; v < n; ++v) solo[v] = type[v] == hipGraphNodeTypeMemcpy;      // (not rebuilt from parameters: a segment of its own)
    const int L = lanes_allowed(max_lanes);
    int policy = 1;
    if (const char* pol = getenv("SENAS_SCHED_POLICY")) policy = (pol[0] == 'c' && pol[1] == 'h') ? 0 : 1;      // "chain" | "critical"
    S->policy = policy;
    int used = 0;
    std::vector<Piece> pieces;
    if (policy == 0 || L <= 1) {
        std::vector<int> lane, seg_of;
        std::vector<PlanSeg> planned;
        plan_schedule((int)n, par, chi, solo, L, lane, seg_of, planned, used, S->n_cross);
        S->segs.resize(planned.size());
        for (size_t k = 0; k < planned.size(); ++k) {
            S->segs[k].lane = planned[k].lane;
            S->segs[k].nodes.swap(planned[k].nodes);
            S->segs[k].deps.swap(planned[k].deps);
            S->segs[k].signals = planned[k].signals;
            S->issue.push_back((int)k);
        }
        S->policy = 0;
    } else {
        std::vector<int> seg_of;
        cut_pieces((int)n, par, chi, solo, seg_of, pieces);
        S->segs.resize(pieces.size());
        for (size_t k = 0; k < pieces.size(); ++k) {
            S->segs[k].nodes = pieces[k].nodes;
            S->segs[k].deps = pieces[k].deps;
        }
    }

    // ---- every segment as a single-branch graph
    for (auto& sg : S->segs) {
        {
            SCHED_HIP(hipGraphCreate(&sg.graph, 0), "hipGraphCreate");
            hipGraphNode_t prev = nullptr;
            for (int v : sg.nodes) {
                hipGraphNode_t src = raw[v], made = nullptr;
                const hipGraphNode_t* deps = prev ? &prev : nullptr;
                const size_t nd = prev ? 1 : 0;
                switch (type[v]) {
                    case hipGraphNodeTypeKernel: {
                        hipKernelNodeParams p;
                        SCHED_HIP(hipGraphKernelNodeGetParams(src, &p), "hipGraphKernelNodeGetParams");
                        SCHED_HIP(hipGraphAddKernelNode(&made, sg.graph, deps, nd, &p), "hipGraphAddKernelNode");
                        ++S->n_kernel;
                        break;
                    }
                    case hipGraphNodeTypeMemset: {
                        hipMemsetParams p;
                        SCHED_HIP(hipGraphMemsetNodeGetParams(src, &p), "hipGraphMemsetNodeGetParams");
                        if (getenv("SENAS_SCHED_VERBOSE") && S->n_memset < 4)
                            fprintf(stderr, "[sched] memset node: dst %p elementSize %u width %zu height %zu pitch %zu value %u\n", p.dst,
                                    p.elementSize, p.width, p.height, p.pitch, p.value);
                        // a 1-D fill (what hipMemsetAsync / hipMemsetD32Async capture) is re-issued as a kernel of this library: a
                        // memset node rebuilt from these parameters did NOT fill its buffer on this runtime (round 4: the zeroed
                        // scratch of the atomically accumulated weight gradients stayed dirty from the second replay on)
                        if (p.height <= 1 && (p.elementSize == 1 || p.elementSize == 2 || p.elementSize == 4)) {
                            void* dst = p.dst;
                            unsigned long long count = (unsigned long long)p.width;
                            unsigned value = p.value;
                            int esz = (int)p.elementSize;
                            void* args[] = {&dst, &count, &value, &esz};
                            hipKernelNodeParams kp{};
                            kp.func = reinterpret_cast<void*>(sched_fill_kernel);
                            const unsigned long long words = (count * esz + 3) / 4;
                            kp.gridDim = dim3((unsigned)std::min<unsigned long long>((words + 255) / 256, 4096ull));
                            kp.blockDim = dim3(256);
                            kp.sharedMemBytes = 0;
                            kp.kernelParams = args;
                            kp.extra = nullptr;
                            SCHED_HIP(hipGraphAddKernelNode(&made, sg.graph, deps, nd, &kp), "hipGraphAddKernelNode (fill)");
                        } else {
                            // a 2-D fill (or an element size the fill kernel does not write): the runtime's own memset node is the
                            // path that did not fill its buffer -- refuse; the step driver then captures the pass on one stream
                            set_error_msg("sched_create: the captured graph holds a 2-D memset node, which the lane scheduler does not re-issue");
                            sched_free(S);
                            return SENAS_EUNSUPPORTED;
                        }
                        ++S->n_memset;
                        break;
                    }
                    case hipGraphNodeTypeMemcpy: {
                        // (alone in its segment: see the segmentation above)
                        SCHED_HIP(hipGraphDestroy(sg.graph), "hipGraphDestroy");
                        sg.graph = nullptr;
                        SCHED_HIP(hipGraphClone(&sg.graph, G), "hipGraphClone");
                        hipGraphNode_t keep = nullptr;
                        SCHED_HIP(hipGraphNodeFindInClone(&keep, src, sg.graph), "hipGraphNodeFindInClone");
                        size_t cn = 0;
                        SCHED_HIP(hipGraphGetNodes(sg.graph, nullptr, &cn), "hipGraphGetNodes");
                        std::vector<hipGraphNode_t> all(cn);
                        SCHED_HIP(hipGraphGetNodes(sg.graph, all.data(), &cn), "hipGraphGetNodes");
                        for (hipGraphNode_t x : all) if (x != keep) SCHED_HIP(hipGraphDestroyNode(x), "hipGraphDestroyNode");
                        made = keep;
                        ++S->n_memcpy;
                        break;
                    }
                    default:
                        set_error_msg("sched_create: the captured graph holds a node type the lane scheduler does not rebuild");
                        sched_free(S);
                        return SENAS_EUNSUPPORTED;
                }
                prev = made;
            }
            SCHED_HIP(hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0), "hipGraphInstantiate");
        }
    }
    auto& pool = lane_pool();
    if (S->policy == 0) {
        // ---- "chain": lanes pinned to streams by node count
        S->lanes.assign(used, nullptr);
        if (used > 1) {
            static std::mutex mu;
            std::lock_guard<std::mutex> lock(mu);
            lane_pool_grow(used);
            if (pool.empty()) { set_error_msg("sched_create: no lane stream could be created"); sched_free(S); return SENAS_ELAUNCH; }
            // fewer hardware queues than lanes: the lanes with the fewest nodes share streams, the heaviest keep theirs to themselves
            // (rank by node count; rank r < P owns stream r; rank P + i shares with rank P - 1 - (i mod P), the lightest owners first)
            const int P = (int)pool.size();
            std::vector<int> weight(used, 0), order(used);
            for (auto& sg : S->segs) weight[sg.lane] += (int)sg.nodes.size();
            for (int q = 0; q < used; ++q) order[q] = q;
            std::sort(order.begin(), order.end(), [&](int a, int b) { return weight[a] != weight[b] ? weight[a] > weight[b] : a < b; });
            // (which owner the fifth lane shares with moves the search step by 0.9 ms: lightest 28.4, heaviest 29.3 -- r4_lanes_queues.txt)
            for (int r = 0; r < used; ++r) S->lanes[order[r]] = pool[r < P ? r : P - 1 - ((r - P) % P)];
        }
    } else {
        // ---- "critical": time every segment once (a serial replay on one stream: the segments' numbering is a topological order, and
        // the pass is self-contained -- it zeroes what it accumulates into), then deal them to the streams there are
        {
            static std::mutex mu;
            std::lock_guard<std::mutex> lock(mu);
            lane_pool_grow(std::min(L, 4));
        }
        if (pool.empty()) { set_error_msg("sched_create: no lane stream could be created"); sched_free(S); return SENAS_ELAUNCH; }
        const int P = std::min((int)pool.size(), L);
        const int K = (int)S->segs.size();
        std::vector<double> dur(K, 1.0);
        if (const char* t = getenv("SENAS_SCHED_NO_TIMING"); t && t[0] == '1') {
            for (int k = 0; k < K; ++k) dur[k] = (double)S->segs[k].nodes.size();      // (measurement: node counts instead of times)
        } else {
            std::vector<hipEvent_t> ev(K + 1, nullptr);
            hipStream_t st = pool[0];
            bool ok = hipDeviceSynchronize() == hipSuccess;
            for (int k = 0; k <= K && ok; ++k) ok = hipEventCreate(&ev[k]) == hipSuccess;
            for (int rep = 0; rep < 2 && ok; ++rep) {
                for (int k = 0; k < K && ok; ++k) {
                    ok = hipEventRecord(ev[k], st) == hipSuccess;
                    if (ok && S->segs[k].exec) ok = hipGraphLaunch(S->segs[k].exec, st) == hipSuccess;
                }
                ok = ok && hipEventRecord(ev[K], st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            }
            for (int k = 0; k < K && ok; ++k) {
                float ms = 0.f;
                ok = hipEventElapsedTime(&ms, ev[k], ev[k + 1]) == hipSuccess;
                dur[k] = 1e3 * (double)ms;
            }
            for (auto e : ev) if (e) (void)hipEventDestroy(e);
            if (!ok) { set_error("sched_create: the timing replay", hipGetLastError()); sched_free(S); return SENAS_ELAUNCH; }
        }
        std::vector<int> stream;
        // light / heavy: by the average kernel time of the segment in the serial replay.  Measured on the search step (serial trace,
        // profiles/r4_search_cells.txt): cells on maps <= 64 x 64 average 8.6 - 13.2 us per launch (the launch floor and a few
        // workgroups each), cells on 128 x 128 maps 17 - 20 us, the head 53 us.
        double light_us = 15.0;
        if (const char* e = getenv("SENAS_SCHED_LIGHT_US")) light_us = atof(e);
        std::vector<char> light(K, 0);
        int n_light = 0;
        for (int k = 0; k < K; ++k) { light[k] = dur[k] / (double)std::max<size_t>(1, S->segs[k].nodes.size()) < light_us; n_light += light[k]; }
        const bool classes = pool_priorities() && P >= 4 && n_light > 0 && n_light < K;
        list_schedule(pieces, dur, P, stream, S->issue, classes ? &light : nullptr, classes ? kHighStreams : 0);
        // ---- refinement: the serial replay times every segment ALONE; side by side with others on the chip it takes longer (a small
        // cell beside a 128 x 128 cell: 1.3 - 1.5x), so the simulation's time line drifts from the real one and a segment of the
        // critical path ends up queued behind a side cell on its stream.  So: run the plan itself with a pair of timing events around
        // every segment, plan again with the durations the segments had IN that plan, and keep whichever plan measured the shortest
        // pass (the first one included: the result is never worse than the unrefined plan by its own measurement).
        int refine = 2;
        if (const char* e = getenv("SENAS_SCHED_REFINE")) refine = atoi(e);
        if (refine > 0 && P > 1 && K > 1) {
            std::vector<hipEvent_t> e0(K, nullptr), e1(K, nullptr);
            hipEvent_t t0 = nullptr, t1 = nullptr;
            bool ok = hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess;
            for (int k = 0; k < K && ok; ++k) ok = hipEventCreate(&e0[k]) == hipSuccess && hipEventCreate(&e1[k]) == hipSuccess;
            auto run = [&](const std::vector<int>& st_of, const std::vector<int>& order, std::vector<double>& seen, double& span_us) {
                bool fine = hipDeviceSynchronize() == hipSuccess;
                for (int rep = 0; rep < 2 && fine; ++rep) {
                    fine = hipEventRecord(t0, pool[0]) == hipSuccess;
                    for (int q = 1; q < P && fine; ++q) fine = hipStreamWaitEvent(pool[q], t0, 0) == hipSuccess;
                    for (int k : order) {
                        if (!fine) break;
                        hipStream_t st = pool[st_of[k]];
                        for (int d : S->segs[k].deps)
                            if (st_of[d] != st_of[k]) fine = fine && hipStreamWaitEvent(st, e1[d], 0) == hipSuccess;
                        fine = fine && hipEventRecord(e0[k], st) == hipSuccess;
                        if (S->segs[k].exec) fine = fine && hipGraphLaunch(S->segs[k].exec, st) == hipSuccess;
                        fine = fine && hipEventRecord(e1[k], st) == hipSuccess;
                    }
                    // join on stream 0: the last segment issued on every stream
                    std::vector<int> last(P, -1);
                    for (int k : order) last[st_of[k]] = k;
                    for (int q = 1; q < P && fine; ++q) if (last[q] >= 0) fine = hipStreamWaitEvent(pool[0], e1[last[q]], 0) == hipSuccess;
                    fine = fine && hipEventRecord(t1, pool[0]) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
                }
                float ms = 0.f;
                fine = fine && hipEventElapsedTime(&ms, t0, t1) == hipSuccess;
                span_us = 1e3 * (double)ms;
                seen.assign(K, 0.0);
                for (int k = 0; k < K && fine; ++k) {
                    fine = hipEventElapsedTime(&ms, e0[k], e1[k]) == hipSuccess;
                    seen[k] = 1e3 * (double)ms;
                }
                return fine;
            };
            std::vector<int> best_stream = stream, best_issue = S->issue;
            std::vector<double> seen;
            double best_span = 0.0, span = 0.0;
            ok = ok && run(stream, S->issue, seen, best_span);
            const bool verbose = getenv("SENAS_SCHED_VERBOSE") != nullptr;
            if (verbose && ok) fprintf(stderr, "[sched] plan 0 (durations of the serial replay): the pass takes %.1f us\n", best_span);
            for (int it = 1; it <= refine && ok; ++it) {
                std::vector<int> st2, is2;
                list_schedule(pieces, seen, P, st2, is2, classes ? &light : nullptr, classes ? kHighStreams : 0);
                std::vector<double> seen2;
                ok = run(st2, is2, seen2, span);
                if (verbose && ok) fprintf(stderr, "[sched] plan %d (durations seen in plan %d): the pass takes %.1f us\n", it, it - 1, span);
                if (ok && span < best_span) { best_span = span; best_stream = st2; best_issue = is2; }
                seen.swap(seen2);
            }
            for (auto e : e0) if (e) (void)hipEventDestroy(e);
            for (auto e : e1) if (e) (void)hipEventDestroy(e);
            if (t0) (void)hipEventDestroy(t0);
            if (t1) (void)hipEventDestroy(t1);
            if (!ok) { set_error("sched_create: the refinement replay", hipGetLastError()); sched_free(S); return SENAS_ELAUNCH; }
            stream = best_stream;
            S->issue = best_issue;
        }
        used = 0;
        for (int k = 0; k < K; ++k) { S->segs[k].lane = stream[k]; S->segs[k].dur_us = (float)dur[k]; used = std::max(used, stream[k] + 1); }
        for (int k = 0; k < K; ++k)
            for (int d : S->segs[k].deps) if (stream[d] != stream[k]) { S->segs[d].signals = true; ++S->n_cross; }
        S->lanes.assign(used, nullptr);
        for (int q = 0; q < used; ++q) S->lanes[q] = pool[q];
    }
    S->lane_done.assign(used, nullptr);
    for (auto& sg : S->segs)
        if (sg.signals) SCHED_HIP(hipEventCreateWithFlags(&sg.done, hipEventDisableTiming), "hipEventCreateWithFlags");
    if (const char* path = getenv("SENAS_SCHED_DUMP")) {
        // tools/lane_timeline.py --segments: which segment and stream every time stamp of the pass (i.e. every cell boundary) sits on,
        // which segments each one waits for, its measured duration; in issue order
        if (FILE* f = fopen(path, "a")) {
            fprintf(f, "sched nodes %zu lanes %d segments %zu policy %s\n", n, used, S->segs.size(), S->policy ? "critical" : "chain");
            for (int k : S->issue) {
                const Segment& sg = S->segs[k];
                int stream = -1;
                for (size_t r = 0; r < pool.size(); ++r) if (used > 1 && pool[r] == S->lanes[sg.lane]) stream = (int)r;
                fprintf(f, "seg %d lane %d stream %d nodes %zu first %d last %d dur_us %.1f deps", k, sg.lane, stream, sg.nodes.size(), sg.nodes.front(),
                        sg.nodes.back(), sg.dur_us);
                for (int d : sg.deps) fprintf(f, " %d", d);
                fprintf(f, " stamps");
                for (int v : sg.nodes) {
                    if (type[v] != hipGraphNodeTypeKernel) continue;
                    hipKernelNodeParams p;
                    if (hipGraphKernelNodeGetParams(raw[v], &p) != hipSuccess || p.func != reinterpret_cast<void*>(stamp_kernel) || !p.kernelParams) continue;
                    fprintf(f, " %llu", (unsigned long long)(uintptr_t)*reinterpret_cast<unsigned long long**>(p.kernelParams[0]));
                }
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    for (int q = 0; q < used; ++q) SCHED_HIP(hipEventCreateWithFlags(&S->lane_done[q], hipEventDisableTiming), "hipEventCreateWithFlags");
    SCHED_HIP(hipEventCreateWithFlags(&S->start, hipEventDisableTiming), "hipEventCreateWithFlags");
    S->n_nodes = (int)n;
    S->n_lanes = used;
    *out = S;
    return SENAS_OK;
}

extern "C" int senas_sched_plan(int32_t n, int32_t m, const int32_t* from, const int32_t* to, const uint8_t* solo, int32_t max_lanes,
                                int32_t* node_lane, int32_t* node_segment, int32_t* n_segments, int32_t* seg_lane,
                                int32_t* seg_dep_begin, int32_t* seg_deps) {
    SENAS_REQUIRE(n >= 1 && m >= 0 && (m == 0 || (from && to)) && max_lanes >= 1 && max_lanes <= 16 && node_lane && node_segment &&
                  n_segments && seg_lane && seg_dep_begin && seg_deps, "sched_plan: bad argument");
    std::vector<std::vector<int>> par(n), chi(n);
    for (int e = 0; e < m; ++e) {
        SENAS_REQUIRE(from[e] >= 0 && to[e] < n && from[e] < to[e], "sched_plan: nodes must be numbered topologically (from < to)");
        par[to[e]].push_back(from[e]);
        chi[from[e]].push_back(to[e]);
    }
    std::vector<char> so(n, 0);
    if (solo) for (int v = 0; v < n; ++v) so[v] = solo[v] != 0;
    std::vector<int> lane, seg_of;
    std::vector<PlanSeg> segs;
    int used = 0, cross = 0;
    plan_schedule(n, par, chi, so, max_lanes, lane, seg_of, segs, used, cross);
    for (int v = 0; v < n; ++v) { node_lane[v] = lane[v]; node_segment[v] = seg_of[v]; }
    *n_segments = (int)segs.size();
    int off = 0;
    for (size_t k = 0; k < segs.size(); ++k) {
        seg_lane[k] = segs[k].lane;
        seg_dep_begin[k] = off;
        for (int d : segs[k].deps) seg_deps[off++] = d;            // (at most one per other lane and never more than the edges: <= m)
    }
    seg_dep_begin[segs.size()] = off;
    return SENAS_OK;
}

extern "C" int senas_sched_contract(int32_t n, int32_t m, const int32_t* from, const int32_t* to, const int32_t* kind, int32_t* out_m,
                                    int32_t* out_from, int32_t* out_to, int32_t cap) {
    SENAS_REQUIRE(n >= 1 && m >= 0 && (m == 0 || (from && to)) && kind && out_m && out_from && out_to && cap >= 0, "sched_contract: bad argument");
    std::vector<std::vector<int>> par(n), chi(n);
    for (int e = 0; e < m; ++e) {
        SENAS_REQUIRE(from[e] >= 0 && to[e] < n && from[e] < to[e], "sched_contract: nodes must be numbered topologically (from < to)");
        par[to[e]].push_back(from[e]);
        chi[from[e]].push_back(to[e]);
    }
    std::vector<int> k(kind, kind + n);
    contract_markers(n, par, chi, k);
    int cnt = 0;
    for (int v = 0; v < n; ++v)
        for (int p : par[v]) {
            SENAS_REQUIRE(cnt < cap, "sched_contract: the output arrays are too small");
            out_from[cnt] = p;
            out_to[cnt++] = v;
        }
    *out_m = cnt;
    return SENAS_OK;
}

extern "C" int senas_sched_plan2(int32_t n, int32_t m, const int32_t* from, const int32_t* to, const uint8_t* solo, const double* node_us,
                                 int32_t streams, int32_t* node_segment, int32_t* n_segments, int32_t* seg_stream, int32_t* seg_issue,
                                 int32_t* seg_dep_begin, int32_t* seg_deps) {
    SENAS_REQUIRE(n >= 1 && m >= 0 && (m == 0 || (from && to)) && streams >= 1 && streams <= 16 && node_segment && n_segments && seg_stream &&
                  seg_issue && seg_dep_begin && seg_deps, "sched_plan2: bad argument");
    std::vector<std::vector<int>> par(n), chi(n);
    for (int e = 0; e < m; ++e) {
        SENAS_REQUIRE(from[e] >= 0 && to[e] < n && from[e] < to[e], "sched_plan2: nodes must be numbered topologically (from < to)");
        par[to[e]].push_back(from[e]);
        chi[from[e]].push_back(to[e]);
    }
    std::vector<char> so(n, 0);
    if (solo) for (int v = 0; v < n; ++v) so[v] = solo[v] != 0;
    std::vector<int> seg_of, stream, issue;
    std::vector<Piece> segs;
    cut_pieces(n, par, chi, so, seg_of, segs);
    std::vector<double> dur(segs.size(), 0.0);
    for (size_t k = 0; k < segs.size(); ++k) for (int v : segs[k].nodes) dur[k] += node_us ? node_us[v] : 1.0;
    list_schedule(segs, dur, streams, stream, issue);
    for (int v = 0; v < n; ++v) node_segment[v] = seg_of[v];
    *n_segments = (int)segs.size();
    int off = 0;
    for (size_t k = 0; k < segs.size(); ++k) {
        seg_stream[k] = stream[k];
        seg_issue[k] = issue[k];
        seg_dep_begin[k] = off;
        for (int d : segs[k].deps) seg_deps[off++] = d;            // (one per parent segment of the first node: <= m)
    }
    seg_dep_begin[segs.size()] = off;
    return SENAS_OK;
}

extern "C" int senas_sched_launch(void* sched, void* stream) {
    SENAS_REQUIRE(sched != nullptr, "sched_launch: bad argument");
    Sched* S = reinterpret_cast<Sched*>(sched);
    hipStream_t main = as_stream(stream);
#define LAUNCH_HIP(call, what) do { hipError_t e__ = (call); if (e__ != hipSuccess) { set_error(what, e__); return SENAS_ELAUNCH; } } while (0)
    // the caller's stream only forks and joins: the lanes are the pool's streams, one hardware queue each
    const bool forked = S->n_lanes > 1;
    if (forked) {
        LAUNCH_HIP(hipEventRecord(S->start, main), "hipEventRecord");
        for (int q = 0; q < S->n_lanes; ++q) LAUNCH_HIP(hipStreamWaitEvent(S->lanes[q], S->start, 0), "hipStreamWaitEvent");
    }
    for (int k : S->issue) {
        Segment& sg = S->segs[k];
        hipStream_t s = forked ? S->lanes[sg.lane] : main;
        for (int d : sg.deps)                                   // (a dependency on the segment's own stream is the stream's order)
            if (forked && S->segs[d].lane != sg.lane) LAUNCH_HIP(hipStreamWaitEvent(s, S->segs[d].done, 0), "hipStreamWaitEvent");
        if (sg.exec) LAUNCH_HIP(hipGraphLaunch(sg.exec, s), "hipGraphLaunch");
        if (forked && sg.signals) LAUNCH_HIP(hipEventRecord(sg.done, s), "hipEventRecord");
    }
    for (int q = 0; forked && q < S->n_lanes; ++q) {
        LAUNCH_HIP(hipEventRecord(S->lane_done[q], S->lanes[q]), "hipEventRecord");
        LAUNCH_HIP(hipStreamWaitEvent(main, S->lane_done[q], 0), "hipStreamWaitEvent");
    }
#undef LAUNCH_HIP
    return SENAS_OK;
}

extern "C" int senas_sched_info(void* sched, int32_t* out8) {
    SENAS_REQUIRE(sched != nullptr && out8 != nullptr, "sched_info: bad argument");
    Sched* S = reinterpret_cast<Sched*>(sched);
    out8[0] = S->n_nodes; out8[1] = S->n_lanes; out8[2] = (int)S->segs.size(); out8[3] = S->n_cross;
    out8[4] = S->n_kernel; out8[5] = S->n_memset; out8[6] = S->n_memcpy; out8[7] = S->n_empty + S->n_marker;
    return SENAS_OK;
}

extern "C" void senas_sched_destroy(void* sched) { sched_free(reinterpret_cast<Sched*>(sched)); }
