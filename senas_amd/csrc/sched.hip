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
// So: take the captured hipGraph_t (never instantiated), read its nodes and edges, cover the DAG with at most L chains
// ("lanes"), cut every chain where a dependency crosses lanes, rebuild every piece as a single-branch graph (kernel nodes
// re-added from their own parameters, 1-D memsets as a fill kernel of this library, a memcpy node as the one survivor of a
// clone of the captured graph; each node depending on its predecessor only), and at launch time issue the
// pieces in topological order on L streams with an event per cross-lane dependency.  Dependencies are exactly the captured
// ones (+ the chain order inside a lane); memory safety is the capture's (torch's caching allocator saw every lane as a
// stream of its own).
#include "common.h"

#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <queue>
#include <unordered_map>
#include <vector>

namespace senas {

struct Segment {
    int lane = 0;
    std::vector<int> nodes;      // indices into Sched::node (topological positions)
    std::vector<int> deps;       // segments whose `done` event this one waits for
    bool signals = false;        // somebody waits for it
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEvent_t done = nullptr;
};

struct Sched {
    std::vector<Segment> segs;
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

// grow the pool to `want` streams on distinct hardware queues (fewer if the device does not give that many)
static void lane_pool_grow(int want) {
    auto& pool = lane_pool();
    std::vector<hipStream_t> rejected;
    for (int tries = 0; (int)pool.size() < want && tries < 16; ++tries) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
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

// How many lanes a captured pass may be spread over.  The schedule is tuned to the runtime's default of FOUR hardware queues per
// process: with GPU_MAX_HW_QUEUES = 5 / 6 / 8 the same step ran 1.4 - 2x SLOWER than on one stream, and with fewer than four
// distinct queues the lanes share them (profiles/r4_lanes_queues.txt, r5_queue_guard.txt).  So: an override of that variable to
// anything but 4, or a device that does not give `want` streams on distinct queues, keeps the serial schedule -- one line on
// stderr, never a slower step.  SENAS_SCHED_TRUST_QUEUES=1 skips the guard (measurement).
static int lanes_allowed(int want) {
    if (want <= 1) return want;
    if (const char* t = getenv("SENAS_SCHED_TRUST_QUEUES")) if (t[0] == '1') return want;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    static int warned = 0;
    if (const char* q = getenv("GPU_MAX_HW_QUEUES")) {
        const int v = atoi(q);
        if (v != 4) {
            if (!warned++) fprintf(stderr, "[senas sched] GPU_MAX_HW_QUEUES=%d (the lane schedule is tuned to the default of 4): captured passes keep the serial schedule\n", v);
            return 1;
        }
    }
    const int need = std::min(want, 4);
    lane_pool_grow(need);
    const int have = (int)lane_pool().size();
    if (have < need) {
        if (!warned++) fprintf(stderr, "[senas sched] %d distinct hardware queue(s) where the lane schedule needs %d: captured passes keep the serial schedule\n", have, need);
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
__global__ void relay_marker_kernel() {}

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

extern "C" int senas_relay_marker(void* stream) {
    hipLaunchKernelGGL(relay_marker_kernel, dim3(1), dim3(1), 0, as_stream(stream));
    return launch_status("relay_marker");
}

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
    // ---- relay markers and empty nodes: contracted out of the graph (a removed node hands its parents to its children; the
    // marker -> marker chain itself is kept: see the note at relay_marker_kernel)
    {
        std::vector<char> gone(n, 0);
        for (size_t i = 0; i < n; ++i) {
            hipGraphNodeType t;
            SCHED_HIP(hipGraphNodeGetType(raw[i], &t), "hipGraphNodeGetType");
            if (t == hipGraphNodeTypeEmpty) { gone[i] = 1; ++S->n_empty; }
            if (t == hipGraphNodeTypeKernel) {
                hipKernelNodeParams p;
                SCHED_HIP(hipGraphKernelNodeGetParams(raw[i], &p), "hipGraphKernelNodeGetParams");
                if (p.func == reinterpret_cast<void*>(relay_marker_kernel)) { gone[i] = 1; ++S->n_marker; }
            }
        }
        // contract in creation order: a removed node hands its parents to its children
        for (size_t i = 0; i < n; ++i) {
            if (!gone[i]) continue;
            for (int c : chi0[i]) {
                auto& cp = par0[c];
                cp.erase(std::remove(cp.begin(), cp.end(), (int)i), cp.end());
                for (int p : par0[i]) if (std::find(cp.begin(), cp.end(), p) == cp.end()) cp.push_back(p);
            }
            for (int p : par0[i]) {
                auto& pc = chi0[p];
                pc.erase(std::remove(pc.begin(), pc.end(), (int)i), pc.end());
                for (int c : chi0[i]) if (std::find(pc.begin(), pc.end(), c) == pc.end()) pc.push_back(c);
            }
            par0[i].clear();
            chi0[i].clear();
        }
        // compact: only the surviving nodes take part from here on
        std::vector<int> newidx(n, -1);
        std::vector<hipGraphNode_t> kept;
        for (size_t i = 0; i < n; ++i) if (!gone[i]) { newidx[i] = (int)kept.size(); kept.push_back(raw[i]); }
        if (kept.empty()) { set_error_msg("sched_create: the graph has no nodes to run"); sched_free(S); return SENAS_EINVAL; }
        std::vector<std::vector<int>> np(kept.size()), nc(kept.size());
        for (size_t i = 0; i < n; ++i) {
            if (gone[i]) continue;
            for (int p : par0[i]) np[newidx[i]].push_back(newidx[p]);
            for (int c : chi0[i]) nc[newidx[i]].push_back(newidx[c]);
        }
        S->n_captured = (int)n;
        raw.swap(kept);
        par0.swap(np);
        chi0.swap(nc);
        n = raw.size();
    }
    if (getenv("SENAS_SCHED_VERBOSE")) {
        size_t back = 0, edges = 0;
        for (size_t i = 0; i < n; ++i) for (int p : par0[i]) { ++edges; if ((size_t)p > i) ++back; }
        fprintf(stderr, "[sched] %zu nodes, %zu edges, %zu edges point from a later node of hipGraphGetNodes to an earlier one\n", n, edges, back);
    }
    // topological order, ties broken by the runtime's own node order (creation order)
    std::vector<int> indeg(n), topo, pos(n);
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
    // from here on a node is its topological position
    std::vector<std::vector<int>> par(n), chi(n);
    for (size_t i = 0; i < n; ++i) {
        for (int p : par0[i]) par[pos[i]].push_back(pos[p]);
        for (int c : chi0[i]) chi[pos[i]].push_back(pos[c]);
    }
    std::vector<hipGraphNodeType> type(n);
    for (size_t v = 0; v < n; ++v) SCHED_HIP(hipGraphNodeGetType(raw[topo[v]], &type[v]), "hipGraphNodeGetType");

    // ---- chain cover, segment cut, cross-lane dependencies (host arithmetic: plan_schedule above)
    std::vector<char> solo(n, 0);
    for (size_t v = 0; v < n; ++v) solo[v] = type[v] == hipGraphNodeTypeMemcpy;      // (not rebuilt from parameters: a segment of its own)
    int L = lanes_allowed(max_lanes);
    std::vector<int> lane, seg_of;
    std::vector<PlanSeg> planned;
    int used = 0;
    plan_schedule((int)n, par, chi, solo, L, lane, seg_of, planned, used, S->n_cross);
    S->segs.resize(planned.size());
    for (size_t k = 0; k < planned.size(); ++k) {
        S->segs[k].lane = planned[k].lane;
        S->segs[k].nodes.swap(planned[k].nodes);
        S->segs[k].deps.swap(planned[k].deps);
        S->segs[k].signals = planned[k].signals;
    }

    // ---- every segment as a single-branch graph
    for (auto& sg : S->segs) {
        {
            SCHED_HIP(hipGraphCreate(&sg.graph, 0), "hipGraphCreate");
            hipGraphNode_t prev = nullptr;
            for (int v : sg.nodes) {
                hipGraphNode_t src = raw[topo[v]], made = nullptr;
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
        if (sg.signals) SCHED_HIP(hipEventCreateWithFlags(&sg.done, hipEventDisableTiming), "hipEventCreateWithFlags");
    }
    S->lanes.assign(used, nullptr);
    S->lane_done.assign(used, nullptr);
    if (used > 1) {
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        lane_pool_grow(used);
        auto& pool = lane_pool();
        if (pool.empty()) { set_error_msg("sched_create: no lane stream could be created"); sched_free(S); return SENAS_ELAUNCH; }
        // fewer hardware queues than lanes: the lanes with the fewest nodes share streams, the heaviest keep theirs to themselves
        // (rank by node count; rank r < P owns stream r; rank P + i shares with rank P - 1 - (i mod P), the lightest owners first)
        const int P = (int)pool.size();
        std::vector<int> weight(used, 0), order(used);
        for (int v = 0; v < (int)n; ++v) ++weight[lane[v]];
        for (int q = 0; q < used; ++q) order[q] = q;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return weight[a] != weight[b] ? weight[a] > weight[b] : a < b; });
        // (which owner the fifth lane shares with moves the search step by 0.9 ms: lightest 28.4, heaviest 29.3 -- r4_lanes_queues.txt)
        for (int r = 0; r < used; ++r) S->lanes[order[r]] = pool[r < P ? r : P - 1 - ((r - P) % P)];
    }
    if (const char* path = getenv("SENAS_SCHED_DUMP")) {
        // tools/lane_timeline.py --segments: which segment, lane and stream every time stamp of the pass (i.e. every cell boundary) sits
        // on, and which segments each one waits for
        if (FILE* f = fopen(path, "a")) {
            fprintf(f, "sched nodes %zu lanes %d segments %zu\n", n, used, S->segs.size());
            auto& pool = lane_pool();
            for (size_t k = 0; k < S->segs.size(); ++k) {
                const Segment& sg = S->segs[k];
                int stream = -1;
                for (size_t r = 0; r < pool.size(); ++r) if (used > 1 && pool[r] == S->lanes[sg.lane]) stream = (int)r;
                fprintf(f, "seg %zu lane %d stream %d nodes %zu first %d last %d deps", k, sg.lane, stream, sg.nodes.size(), sg.nodes.front(), sg.nodes.back());
                for (int d : sg.deps) fprintf(f, " %d", d);
                fprintf(f, " stamps");
                for (int v : sg.nodes) {
                    if (type[v] != hipGraphNodeTypeKernel) continue;
                    hipKernelNodeParams p;
                    if (hipGraphKernelNodeGetParams(raw[topo[v]], &p) != hipSuccess || p.func != reinterpret_cast<void*>(stamp_kernel) || !p.kernelParams) continue;
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
    for (auto& sg : S->segs) {
        hipStream_t s = forked ? S->lanes[sg.lane] : main;
        for (int d : sg.deps) LAUNCH_HIP(hipStreamWaitEvent(s, S->segs[d].done, 0), "hipStreamWaitEvent");
        if (sg.exec) LAUNCH_HIP(hipGraphLaunch(sg.exec, s), "hipGraphLaunch");
        if (sg.signals) LAUNCH_HIP(hipEventRecord(sg.done, s), "hipEventRecord");
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
