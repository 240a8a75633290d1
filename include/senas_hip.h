/*
 * senas_hip.h -- C ABI of libsenas_hip.so, the MI355X (gfx950) device library behind the SENAS
 * hot path (supernet MixedOp DAG + derived-genotype encoder/decoder).
 *
 * Boundary rules:
 *   - plain pointers and sizes only; no torch types.  All tensor pointers are DEVICE pointers
 *     borrowed from the caller (the caller owns them and keeps them alive until the stream has
 *     passed the launch).  Nothing here allocates, frees or synchronises.
 *   - activations are fp32, NHWC ("channels_last"): element (n,y,x,c) at ((n*H + y)*W + x)*C + c.
 *   - weights keep the reference's (torch) parameter layout so that reference checkpoints load
 *     unchanged: Conv2d [c_out][c_in/groups][kh][kw], ConvTranspose2d [c_in][c_out/groups][kh][kw].
 *   - every entry point launches on `stream` (a hipStream_t passed as void*) and returns
 *     SENAS_OK or a negative error code; launch errors are reported, never swallowed.
 *
 * The reference has no FFI on this path: it is pure Python on torch.nn, so
 * each entry point cites the reference call site whose arithmetic it replaces
 * (paths relative to the reference repository root).
 */
#ifndef SENAS_HIP_H
#define SENAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SENAS_OK            0
#define SENAS_EINVAL       -1   /* bad argument (null pointer, shape mismatch, unsupported size) */
#define SENAS_ELAUNCH      -2   /* hipLaunch / runtime error; see senas_last_error()            */
#define SENAS_EUNSUPPORTED -3

#define SENAS_MAX_TERMS 32
#define SENAS_SKIP_MAX 8     /* tensors one senas_skipcat_* call stacks (a column of a depth-9 macro grid) */

/* Geometry of one convolution, always stated for the FORWARD op y = op(x):
 *   x: [n][hi][wi][ci]  ->  y: [n][ho][wo][co]
 * transposed == 0 : nn.Conv2d           (utils/operations.py:128-129)
 * transposed == 1 : nn.ConvTranspose2d  (utils/operations.py:125-126), output_padding folded in ho/wo
 * groups is 1 (dense) or ci == co (depthwise, utils/operations.py:110).                        */
typedef struct senas_conv_geom {
    int32_t n, hi, wi, ci, ho, wo, co;
    int32_t kh, kw, stride, pad, dil;
    int32_t transposed;
    int32_t groups;
} senas_conv_geom;

/* ---- convolution (dense and depthwise) ------------------------------------------------------
 * Replaces the nn.Conv2d / nn.ConvTranspose2d calls made by ConvBn, ConvBnSe, DepSepConv,
 * AdapterBlock.conv, ShrinkBlock, RectifyBlock, ReLUConv, build_rectify and BasicBlock
 * (utils/operations.py:81-130,141-152,167-183,206-268).
 *
 * in_relu != 0 applies ReLU to x on load (the reference's separate nn.ReLU in front of the conv,
 * e.g. operations.py:84,142,210); zero padding is applied after that ReLU, as in the reference.
 * stats (optional, may be NULL): double[n][co][2], the kernel ADDS per-image per-channel sum and
 * sum of squares of y into it (producer-side batch-norm statistics; caller zeroes it).          */
/* ws: device scratch of at least senas_conv2d_ws_bytes(g) bytes (repacked weights / partial sums),
 * private to the call until the stream has passed it.                                          */
int64_t senas_conv2d_ws_bytes(const senas_conv_geom* g);     /* forward and data gradient */
/* packed (optional, may be NULL): the MFMA fragment image of w for this direction, kept up to date by
 * the caller with senas_pack_batched (layout from senas_conv2d_pack_layout).  NULL: the launcher
 * repacks w into ws itself.  Ignored by the non-MFMA fallback kernels.                            */
int senas_conv2d_fwd(const senas_conv_geom* g, const float* x, const float* w, float* y,
                     int in_relu, double* stats, void* ws, const float* packed, void* stream);
/* The same with the output in PLANAR groups of 8 channels: channel ch of pixel p (p over n*ho*wo) goes to
 * y[(ch / 8) * y_plane + p * 8 + ch % 8], y_plane = n*ho*wo*8 floats, co a multiple of 8 -- every 8-channel group is a dense
 * [n][ho][wo][8] tensor of its own.  For the STACKED convolutions of a search cell (search/cell.py:100-106: the same-named
 * candidates of the edges that leave one state run as one convolution, weights stacked along c_out; every edge's 8-channel
 * slice is then read by a different node kernel -- interleaved, each of those reads would fetch whole 128-byte pixels for 32
 * bytes of payload).  Statistics are per channel as before.  SENAS_EUNSUPPORTED (nothing launched) when the geometry does not
 * land on a kernel with that epilogue (the stride-1 LDS-window kernel, the stride-2 transposed one): the caller then makes the
 * interleaved call.  senas_conv2d_fwd_pair_planar: the pair launch (below) likewise, both outputs planar.                      */
int senas_conv2d_fwd_planar(const senas_conv_geom* g, const float* x, const float* w, float* y, int64_t y_plane,
                            int in_relu, double* stats, void* ws, const float* packed, void* stream);
/* Inference forward (experiments/testing_model.py:150-190 runs the model under model.eval()): eval-mode
 * nn.BatchNorm2d (operations.py:133-134) is a per-channel affine of the convolution output, so it -- together with
 * the node sum and ReLU of the cell (models/senas_model.py:55-63) -- rides in the epilogue of the producer:
 *     y = act( scale[n][c] * conv(x, w) + bias[n][c] + add_scale[n][c] * addend[n,p,c] )
 * scale / bias: float[n][co] (per image so that an SE gate, operations.py:203, can ride along); addend: NHWC tensor
 * shaped like y, or NULL; add_scale: float[n][co] or NULL (= 1).  Returns SENAS_EUNSUPPORTED without launching when
 * the geometry has no epilogue kernel (transposed, thin, off the LDS-window path; depthwise with an addend).      */
typedef struct senas_conv_epilogue {
    const float* scale;
    const float* bias;
    const float* addend;
    const float* add_scale;
    int32_t relu;
} senas_conv_epilogue;
int senas_conv2d_fwd_epilogue(const senas_conv_geom* g, const float* x, const float* w, float* y, int in_relu,
                              const senas_conv_epilogue* e, void* ws, const float* packed, void* stream);
/* dx = d loss / d x.  If in_relu != 0, x must be given and dx is masked by (x > 0).            */
int senas_conv2d_bwd_data(const senas_conv_geom* g, const float* dy, const float* w, float* dx,
                          int in_relu, const float* x, void* ws, const float* packed, void* stream);
/* Packed-weight cache.  direction 0 = forward image, 1 = data-gradient image.  *elems == 0: this
 * convolution has no MFMA image (depthwise, or reduction channels not a multiple of 8).          */
typedef struct senas_pack_item {
    const float* src;            /* weights, torch layout [d0][d1][taps]                          */
    float* dst;                  /* float[elems]                                                   */
    int32_t d0, d1, taps, swap;
    int64_t elems;
} senas_pack_item;
int senas_conv2d_pack_layout(const senas_conv_geom* g, int direction, int32_t* d0, int32_t* d1, int32_t* swap,
                             int64_t* elems);
/* One launch repacks n weight tensors; items_dev is a DEVICE array, max_elems = max over items.   */
int senas_pack_batched(const senas_pack_item* items_dev, int n, int64_t max_elems, void* stream);
/* ---- two convolutions in one launch ---------------------------------------------------------------------------------------
 * dil_3_conv_5 and dil_2_conv_5 of the same edges (utils/operations.py:69-72) read the same tensor with the same shapes and
 * differ in the dilation only: forward and data gradient of both share a launch (grid.z doubled) on the 8-channel MFMA kernel
 * (inner edges) and the LDS-window kernels (stacked 32 -> 32 candidates, stride 1 and 2).  Arguments as the single calls, once
 * per problem.  Returns SENAS_EUNSUPPORTED without launching when the two do not take the same kernel of those families (the
 * caller then makes the two single calls).                                                                               */
int senas_conv2d_fwd_pair(const senas_conv_geom* ga, const senas_conv_geom* gb, const float* x, const float* wa, const float* wb,
                          float* ya, float* yb, int in_relu, double* stats_a, double* stats_b, void* ws_a, void* ws_b,
                          const float* packed_a, const float* packed_b, void* stream);
int senas_conv2d_fwd_pair_planar(const senas_conv_geom* ga, const senas_conv_geom* gb, const float* x, const float* wa, const float* wb,
                                 float* ya, float* yb, int64_t y_plane, int in_relu, double* stats_a, double* stats_b, void* ws_a,
                                 void* ws_b, const float* packed_a, const float* packed_b, void* stream);
int senas_conv2d_bwd_data_pair(const senas_conv_geom* ga, const senas_conv_geom* gb, const float* dya, const float* dyb,
                               const float* wa, const float* wb, float* dxa, float* dxb, int in_relu, const float* x,
                               void* ws_a, void* ws_b, const float* packed_a, const float* packed_b, void* stream);

/* ---- the same convolutions on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulation) -------------------------
 * Stride-1 "same" nn.Conv2d 3x3 / 5x5 with c_in a multiple of 16 (32 for terms == 1) and c_out a multiple of 32 on maps at
 * least 32 wide and 8 high -- the dense candidates and pre/post-process convolutions of the derived cell
 * (utils/operations.py:89-95,118-130,206-218; models/senas_model.py:11-16) -- forward and data gradient.  `terms` selects the
 * operand form; both operands are split the same way, the activations while they are staged in LDS, the weights once per
 * step in the packed image:
 *     1  "bf16"   : operands rounded to bf16 (round to nearest even)
 *     3  "bf16x3" : x = hi + lo, products hi*hi + hi*lo + lo*hi               (per-product error ~2^-17 |x||w|)
 *     6  "bf16x6" : x = hi + mid + lo (exact), 6 products, lo*lo-order terms dropped (<= 2^-25 |x||w|: below fp32 rounding)
 * Everything else is as in senas_conv2d_fwd / _bwd_data (fp32 NHWC tensors in HBM, stats, in_relu, ws).  Shapes off this
 * path return SENAS_EUNSUPPORTED and launch nothing (the caller then uses the fp32 entry points).  packed_lp: the image
 * described by senas_conv2d_pack_layout_lp (elems in 4-byte units), refreshed with senas_pack_batched_lp -- items as for
 * senas_pack_batched with the `swap` value that the layout call returns (it carries `terms`); NULL: repacked into ws.   */
int senas_conv2d_fwd_lp(const senas_conv_geom* g, const float* x, const float* w, float* y, int in_relu, double* stats,
                        void* ws, const void* packed_lp, int terms, void* stream);
int senas_conv2d_bwd_data_lp(const senas_conv_geom* g, const float* dy, const float* w, float* dx, int in_relu, const float* x,
                             void* ws, const void* packed_lp, int terms, void* stream);
int senas_conv2d_pack_layout_lp(const senas_conv_geom* g, int direction, int terms, int32_t* d0, int32_t* d1, int32_t* swap,
                                int64_t* elems);
int senas_pack_batched_lp(const senas_pack_item* items_dev, int n, int64_t max_elems, void* stream);
/* kernel symbol of the lp forward (which = 0) / data gradient (1) / weight gradient (2); "" when the geometry is off the path */
const char* senas_conv2d_kernel_name_lp(const senas_conv_geom* g, int which, int terms);

/* Stacked weights of the search cell (see senas_unstack_fwd): item i copies rows x row_len floats from src rows that
 * are src_stride floats apart into dst rows that are dst_stride floats apart -- every per-edge weight of the model into
 * its slice of a stacked buffer in ONE launch, and (the other way round) every slice of the stacked weight gradients
 * into the per-edge gradients (accumulate != 0: dst += src).  items_dev is a DEVICE array, max_elems = max
 * rows*row_len over the items.                                                                                       */
typedef struct senas_copy_item {
    const float* src;
    float* dst;
    int64_t rows, row_len, src_stride, dst_stride;
    int64_t accumulate;
} senas_copy_item;
int senas_copy_rows_batched(const senas_copy_item* items_dev, int n, int64_t max_elems, void* stream);
/* Workspace of the weight gradient: *bytes to allocate and whether it must be zero-filled on entry
 * (*needs_zero != 0; pass ws_is_zero accordingly, or 0 to let the call clear it itself).            */
int senas_conv2d_bwd_weight_ws(const senas_conv_geom* g, int64_t* bytes, int32_t* needs_zero);
/* dw (same layout as w) is OVERWRITTEN.  ws_is_zero != 0: the caller guarantees that ws is zero-filled
 * (lets the launcher skip its own memset of the split-K accumulation image).                    */
int senas_conv2d_bwd_weight(const senas_conv_geom* g, const float* x, int in_relu, const float* dy,
                            float* dw, void* ws, int ws_is_zero, void* stream);

/* Two-stage weight gradients (per-block partial images, then a sum in block order -- bitwise reproducible) with the
 * second stage DEFERRED: the call launches the partial kernel only and describes the remaining sum in *defer
 * (defer->kind == 0: nothing is left, dw is complete when the stream has passed the call; defer == NULL: as the plain
 * entry point).  ws must then stay alive and untouched until senas_wgrad_sum_batched has run for that item: one launch
 * folds up to SENAS_MAX_SUMS such items (items: HOST array).  A backward pass of the derived network has ~100 of these
 * 5 us sums, the supernet's ~320 (weight gradients of nn.Conv2d / nn.ConvTranspose2d, utils/operations.py:118-130).   */
#define SENAS_MAX_SUMS 64
typedef struct senas_sum_item {
    const float* part;
    float* dw;
    int32_t kind;               /* 0 none, 1 flat partials [nblk][n_elem], 2 wgrad_lds images [nblk][taps * A/32][32][32] */
    int32_t A, B, taps;         /* kind 2 */
    int32_t n_elem;             /* kind 1 */
    int32_t nblk;
} senas_sum_item;
int senas_conv2d_bwd_weight_deferred(const senas_conv_geom* g, const float* x, int in_relu, const float* dy,
                                     float* dw, void* ws, int ws_is_zero, senas_sum_item* defer, void* stream);
int senas_wgrad_sum_batched(const senas_sum_item* items, int n, void* stream);
/* The weight gradients of the two convolutions of senas_conv2d_fwd_pair (one tensor, shapes and kernel size in common, their
 * own dilation / padding: utils/operations.py:69-72; models/senas_model.py:55-63 where two nodes take candidates of one state)
 * as ONE first-stage launch (problem 2 on blockIdx.y) on the 8-channel MFMA kernel or the LDS kernel.  ws_a / ws_b: what
 * senas_conv2d_bwd_weight_ws asks for each; defer_a / defer_b: both NULL (the sums run here) or both given.  dwa / dwb (torch
 * layout) are OVERWRITTEN.  Returns SENAS_EUNSUPPORTED without launching when the two do not share a kernel and tile list
 * (the caller then makes the two single calls).                                                                           */
int senas_conv2d_bwd_weight_pair(const senas_conv_geom* ga, const senas_conv_geom* gb, const float* x, int in_relu,
                                 const float* dya, const float* dyb, float* dwa, float* dwb, void* ws_a, void* ws_b,
                                 senas_sum_item* defer_a, senas_sum_item* defer_b, void* stream);

/* Weight gradient of the same convolutions on the bf16 pipe (both operands split while they are staged in LDS; fragments
 * by transposing LDS reads).  *bytes == 0: the geometry is off the path (use senas_conv2d_bwd_weight).  ws: *bytes of
 * scratch for the per-block partial images (need not be zeroed); defer as in senas_conv2d_bwd_weight_deferred (NULL:
 * the sum runs at once).  dw (torch layout) is OVERWRITTEN.                                                              */
int senas_conv2d_bwd_weight_ws_lp(const senas_conv_geom* g, int terms, int64_t* bytes);
int senas_conv2d_bwd_weight_lp(const senas_conv_geom* g, const float* x, int in_relu, const float* dy, float* dw, void* ws,
                               int terms, senas_sum_item* defer, void* stream);
/* ---- "bf16s": bf16-STORED convolution outputs and their gradients (math mode 'bf16s', senas_amd/functional.py) -----------------
 * The bf16-pipe convolutions above in their plain bf16 form (terms = 1) with the tensor they PRODUCE kept in bf16: the forward
 * pass writes y as bf16 (fp32 accumulators rounded to nearest even; `stats` are taken from the accumulators, before the rounding),
 * and the gradient that comes back for y is a bf16 tensor too, which the data- and weight-gradient kernels stage by a copy.  x, dx,
 * w and dw stay fp32.  Pointers named *_bf16 point to 2-byte elements in the same NHWC order.  Serves what senas_conv2d_fwd_lp
 * serves with terms = 1 (stride-1 "same" 3x3 / 5x5, c_in % 32 == 0, c_out % 32 == 0, maps >= 32 wide); SENAS_EUNSUPPORTED
 * otherwise.  The cell node reads such a term and writes its gradient through senas_node_fwd / _bwd with a NEGATIVE pixel stride
 * for that term (|stride| elements of 2 bytes; at most 4 terms, c % 4 == 0).  Same nn.Conv2d call sites as the _lp entry points
 * (models/senas_model.py:11-16,55-63); the reference has no reduced-precision path (requirements.txt:5, no AMP anywhere).     */
int senas_conv2d_fwd_bf16s(const senas_conv_geom* g, const float* x, const float* w, void* y_bf16, int in_relu, double* stats,
                           void* ws, const void* packed_lp, void* stream);
int senas_conv2d_bwd_data_bf16s(const senas_conv_geom* g, const void* dy_bf16, const float* w, float* dx, int in_relu,
                                const float* x, void* ws, const void* packed_lp, void* stream);
int senas_conv2d_bwd_weight_bf16s(const senas_conv_geom* g, const float* x, int in_relu, const void* dy_bf16, float* dw, void* ws,
                                  senas_sum_item* defer, void* stream);

/* ---- pooling / resampling --------------------------------------------------------------------
 * nn.AvgPool2d(3, stride, 1, count_include_pad=False)  (operations.py:62,150)
 * nn.MaxPool2d(3, stride, 1)                            (operations.py:64; senas_search.py:31)
 * nn.Upsample(scale_factor=2, 'bilinear', align_corners=False) (operations.py:13,145)
 * x: [n][h][w][c]; stride in {1,2}.  in_relu as above.  stats as above (may be NULL).           */
int senas_avgpool3_fwd(int n, int h, int w, int c, int stride, const float* x, int in_relu, float* y,
                       double* stats, void* stream);
int senas_avgpool3_bwd(int n, int h, int w, int c, int stride, const float* dy, int in_relu,
                       const float* x, float* dx, void* stream);
/* argmax: uint8[n][ho][wo][c], window-local index (ky*3+kx) of the first maximum.                */
int senas_maxpool3_fwd(int n, int h, int w, int c, int stride, const float* x, int in_relu, float* y,
                       uint8_t* argmax, double* stats, void* stream);
int senas_maxpool3_bwd(int n, int h, int w, int c, int stride, const float* dy, const uint8_t* argmax,
                       int in_relu, const float* x, float* dx, void* stream);
int senas_bilinear2x_fwd(int n, int h, int w, int c, const float* x, float* y, double* stats, void* stream);
int senas_bilinear2x_bwd(int n, int h, int w, int c, const float* dy, float* dx, void* stream);

/* ---- channel un-stacking ---------------------------------------------------------------------------
 * The edges that LEAVE one state of a search cell (search/cell.py:100-106) read the same tensor with the same
 * geometry, so their same-named candidates run as ONE convolution with the weights stacked along c_out; this splits
 * its output src [n][hw][k*c] into the k per-edge tensors dst[e] [n][hw][c] the nodes consume, adding the per-image
 * channel sums of every part into stats[e] (double[n][c][2], caller zeroes; stats or any stats[e] may be NULL).  A NULL
 * dst[e] (e > 0) skips part e -- the zero-weight padding part that fills a stack of three to a full 32-channel tile.
 * dst / stats: HOST arrays of k device pointers, k <= SENAS_MAX_STACK.                                          */
#define SENAS_MAX_STACK 4
int senas_unstack_fwd(int n, int64_t hw, int c, int k, const float* src, float* const* dst, double* const* stats,
                      void* stream);

/* ---- lane scheduler: a captured multi-stream HIP graph replayed as linear segments on streams of its own ---------------
 * Replaces the runtime's executor for the step drivers' captured passes (senas_amd/step.py; why: csrc/sched.hip).  The
 * reference has no counterpart -- its step loop launches eagerly (experiments/train_model.py:264-305, search_arc.py:252-299).
 * hip_graph: a captured, NOT instantiated hipGraph_t that outlives the scheduler (kernel / memset / memcpy / empty nodes only,
 * else SENAS_EUNSUPPORTED).  max_lanes: streams the pieces are spread over, 1..16 (lane 0 is the `stream` given at launch).
 * senas_sched_launch enqueues one replay: the lanes wait for `stream`, and `stream` waits for the lanes at the end, so
 * successive launches on one stream are ordered like ordinary graph launches.  Not thread-safe per scheduler.
 * senas_sched_info: out8 = {nodes run, lanes used, segments, cross-lane dependencies, kernel, memset, memcpy nodes, nodes
 * contracted away (empty nodes and relay markers)}.
 * A 2-D memset node is refused with SENAS_EUNSUPPORTED as well (the caller then captures the pass on one stream).  With
 * GPU_MAX_HW_QUEUES overridden to more than the runtime's default of 4 (measured: slower than one stream), or no second
 * hardware queue at all, the scheduler keeps everything on the caller's stream (one line on stderr;
 * SENAS_SCHED_TRUST_QUEUES=1 skips the guard).
 * senas_relay_marker: an empty kernel launched on the capture's origin stream at every hand-over between two lanes: it gives
 * the chain of hand-overs the capture records on that stream nodes the scheduler recognises and CONTRACTS out of the graph
 * (their parents become their children's parents; the chain itself is kept -- csrc/sched.hip says why).
 * senas_sched_plan: the scheduler's plan alone, host arithmetic (no device, no graph): nodes 0..n-1 numbered topologically,
 * edges from[e] -> to[e] (from < to), solo[v] != 0 (or solo NULL) for a node that must be a segment of its own.  Outputs:
 * node_lane[n], node_segment[n]; *n_segments; seg_lane[n]; the segments segment k waits for are
 * seg_deps[seg_dep_begin[k] .. seg_dep_begin[k+1]) (seg_dep_begin: n + 1 entries, seg_deps: max(m, 1)).  Segment indices are
 * the issue order.  tests/test_host_logic.py checks the plan's invariants with it.  */
int senas_relay_marker(void* stream);
/* The typed markers of a hand-over between two lanes of a captured pass (senas_amd/grid.py Lanes.hand; csrc/sched.hip):
 * kind 0 RELAY (what senas_relay_marker launches: on the capture's origin stream), 1 PRODUCER (on the lane that made the tensor,
 * in front of the event the origin stream waits for), 2 CONSUMER (on the lane that reads it, behind its wait for the origin
 * stream).  All are empty kernels the scheduler contracts out of the graph; a CONSUMER behind a RELAY takes as its
 * dependencies what the relay's PRODUCER parents stand for and nothing else of the origin stream's history.
 * senas_sched_contract: that contraction alone, host arithmetic: nodes numbered topologically, kind[v] = -1 for a node that runs,
 * 0 / 1 / 2 for the marker kinds, 3 for an empty node; the surviving edges come back in out_from / out_to (cap entries).
 * senas_sched_plan2: the "critical" policy's plan alone (path cover, segments, list scheduling on `streams` streams with the
 * per-node durations node_us, NULL = 1 each): node_segment[n]; *n_segments; seg_stream[k]; seg_issue = the segments in launch
 * order; dependencies as in senas_sched_plan.  tests/test_host_logic.py checks their invariants.  */
int senas_marker(int kind, void* stream);
int senas_sched_contract(int32_t n, int32_t m, const int32_t* from, const int32_t* to, const int32_t* kind, int32_t* out_m,
                         int32_t* out_from, int32_t* out_to, int32_t cap);
int senas_sched_plan2(int32_t n, int32_t m, const int32_t* from, const int32_t* to, const uint8_t* solo, const double* node_us,
                      int32_t streams, int32_t* node_segment, int32_t* n_segments, int32_t* seg_stream, int32_t* seg_issue,
                      int32_t* seg_dep_begin, int32_t* seg_deps);
int senas_sched_plan(int32_t n, int32_t m, const int32_t* from, const int32_t* to, const uint8_t* solo, int32_t max_lanes,
                     int32_t* node_lane, int32_t* node_segment, int32_t* n_segments, int32_t* seg_lane,
                     int32_t* seg_dep_begin, int32_t* seg_deps);
/* A new non-blocking HIP stream of the current device (never destroyed).  The lanes of a captured pass must be streams of their
 * own: torch hands its streams out of a pool of 32 round-robin, and two "different" torch streams that are one hipStream_t turn the
 * star topology of the capture into lane-to-lane waits (the runtime then never returns from hipStreamEndCapture: csrc/sched.hip). */
int senas_stream_create(void** out);
int senas_sched_create(void* hip_graph, int max_lanes, void** out);
int senas_sched_launch(void* sched, void* stream);
int senas_sched_info(void* sched, int32_t* out8);
void senas_sched_destroy(void* sched);

/* ---- time stamp in stream order (measurement only: tools/lane_timeline.py) ------------------------------------------
 * *slot = the device's constant-rate wall clock (100 MHz ticks) when the stream reaches this launch.                   */
int senas_stamp(uint64_t* slot, void* stream);

/* ---- elementwise ReLU (Cell.preprocess1, search/cell.py:66,94; senas_model.py:15,52) ---------- */
int senas_relu_fwd(int64_t numel, const float* x, float* y, void* stream);
int senas_relu_bwd(int64_t numel, const float* dy, const float* y, float* dx, void* stream);

/* ---- gamma-gated blend of two skip candidates (search/senas_search.py:98-102) ---------------------------------------
 * y = g[0] * x1 + g[1] * x2 with g a DEVICE float[2] (a row of softmax(gamma)); backward: dx1 = g[0] * dy,
 * dx2 = g[1] * dy (either may be NULL), dg[0] += sum dy * x1, dg[1] += sum dy * x2 (double[2], caller zeroes).
 * numel % 4 == 0, tensors 16-byte aligned and identically laid out.                                             */
int senas_blend2_fwd(int64_t numel, const float* x1, const float* x2, const float* g, float* y, void* stream);
int senas_blend2_bwd(int64_t numel, const float* dy, const float* x1, const float* x2, const float* g, float* dx1,
                     float* dx2, double* dg, void* stream);

/* ---- in0 of a supernet up cell in one pass (search/senas_search.py:96-103) -------------------------------------------
 * The reference concatenates, along channels, the column's down-path output and the gamma-gated blends of neighbouring
 * outputs below it.  xs: m NHWC tensors [npix][c] (2 <= m <= SENAS_SKIP_MAX, c % 4 == 0, 16-byte aligned); table: DEVICE
 * float[rows][2] = softmax(gamma); idx[k] (k >= 1; idx[0] is ignored): the row that gates slice k.
 *     y[npix][m * c]:  slice 0 = xs[0],  slice k = table[idx[k]][0] * xs[k-1] + table[idx[k]][1] * xs[k]
 * Backward: dy [npix][m * c] -> dxs[k] [npix][c] (NULL: not wanted) = [k == 0] dy_0 + [k >= 1] table[idx[k]][1] dy_k
 * + [k + 1 < m] table[idx[k+1]][0] dy_{k+1};  acc[idx[k]] += (sum dy_k xs[k-1], sum dy_k xs[k]) -- acc: DEVICE
 * double[rows][2], zeroed by the caller once per pass (the d loss / d softmax(gamma) table every blend of the pass adds into). */
int senas_skipcat_fwd(int64_t npix, int c, int m, const float* const* xs, const float* table, int rows, const int32_t* idx,
                      float* y, void* stream);
int senas_skipcat_bwd(int64_t npix, int c, int m, const float* dy, const float* const* xs, const float* table, int rows,
                      const int32_t* idx, float* const* dxs, double* acc, void* stream);

/* ---- the architecture tensors of the supernet, one launch per direction --------------------------------------------------
 * NAS.forward (search/senas_search.py:246-260): row softmax of alphas_dn / alphas_up / alphas_dn_nm / alphas_up_nm ([k][ops]),
 * softmax of betas_dn / betas_up over the node windows [i : 2i + 2] (the reference's overlapping slices, :254-257), row
 * softmax of gamma ([grows][2]); and what every MixedOp / Cell of a kind then multiplies in (search/cell.py:33-36,100-106):
 *     M[kind][e][o] = beta_soft[kind][e] * alpha_soft[kind, NORM edge ? the *_nm table : the dn / up table][e][o]
 * kind 0 = down cell (NORM edges: input state >= 2), kind 1 = up cell (NORM edges: every input state but 1); edges in the
 * order node 0 (states 0, 1), node 1 (states 0..2), ...  k = sum over nodes of (2 + i).
 * Backward: dM[kind] / dG are the tables the cell nodes (senas_node_bwd, dmix_accumulate) and the blends (senas_blend2_bwd)
 * ADDED their gradients into -- dM[kind] with one [k][ops] row block ("slot") per cell of the kind, because the cells of one
 * macro-grid column run on their own HIP stream and a plain read-modify-write table must have one writer at a time; the seven parameter gradients are OVERWRITTEN.  d_alpha[3] == NULL: alphas_up_nm IS alphas_dn_nm
 * (NAS(use_sharing=True), senas_search.py:148-150) and d_alpha[2] receives both contributions.                          */
typedef struct senas_arch_mix {
    const float* alpha[4];      /* dn, up, dn_nm, up_nm */
    const float* beta[2];       /* dn, up */
    const float* gamma;
    float* s_alpha[4];
    float* s_beta[2];
    float* s_gamma;
    float* M[2];
    const float* dM[2];
    const double* dG;
    float* d_alpha[4];
    float* d_beta[2];
    float* d_gamma;
    int32_t k, ops, nodes, grows;
    int32_t slots;              /* backward: dM[kind] is [slots][k][ops]; the rows are folded in slot order (1..256)    */
} senas_arch_mix;
int senas_arch_mix_fwd(const senas_arch_mix* a, void* stream);
int senas_arch_mix_bwd(const senas_arch_mix* a, void* stream);

/* ---- batch-norm statistics --------------------------------------------------------------------
 * Per-image per-channel sum / sum-of-squares of x [n][hw][c] ADDED into stats double[n][c][2].   */
int senas_chan_stats(int n, int64_t hw, int c, const float* x, double* stats, void* stream);

/* nn.BatchNorm2d training/eval semantics (operations.py:133-134; torch defaults eps=1e-5,
 * momentum=0.1, biased variance for normalisation, unbiased for running_var).
 * stats: double[n][c][2] as produced above (ignored when training == 0).
 * Outputs (float[c] each): mean, invstd, scale = gamma*invstd, shift = beta - mean*scale.
 * When training != 0 the running buffers are updated in place and *num_batches_tracked += 1.   */
int senas_bn_finalize(int n, int64_t hw, int c, const double* stats, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked,
                      float momentum, float eps, int training,
                      float* mean, float* invstd, float* scale, float* shift, void* stream);

/* ---- k depthwise convolutions of ONE input -----------------------------------------------------------------------
 * The same-named DepSepConv candidates (utils/operations.py:107-115) of the edges that leave one state of a search
 * cell: one input x, k depthwise weight sets w[p] ([c][1][kh][kw]), k outputs.  One launch forward; the data gradient
 * dx = sum_p dgrad(dy[p], w[p]) in one launch; the k weight gradients in two.  w / y / stats / dy / dw: HOST arrays of
 * k pointers, k <= SENAS_MAX_DWMULTI.  g describes ONE of the convolutions (groups == ci == co).  Returns
 * SENAS_EUNSUPPORTED (nothing launched) off the fast path (channels not a power-of-two multiple of 4, taps not 3x3 / 5x5);
 * ws: senas_dwconv_multi_ws_bytes(g, k) bytes of scratch (per-block partial sums, overwritten).                       */
#define SENAS_MAX_DWMULTI 12
int senas_dwconv_multi_fwd(const senas_conv_geom* g, int k, const float* x, const float* const* w, float* const* y,
                           double* const* stats, void* stream);
int senas_dwconv_multi_bwd_data(const senas_conv_geom* g, int k, const float* const* dy, const float* const* w, float* dx,
                                void* stream);
int64_t senas_dwconv_multi_ws_bytes(const senas_conv_geom* g, int k);
int senas_dwconv_multi_bwd_weight(const senas_conv_geom* g, int k, const float* x, const float* const* dy, float* const* dw,
                                  void* ws, void* stream);
/* as above with the k sums deferred (defer: k items, see senas_wgrad_sum_batched)                                       */
int senas_dwconv_multi_bwd_weight_deferred(const senas_conv_geom* g, int k, const float* x, const float* const* dy,
                                           float* const* dw, void* ws, senas_sum_item* defer, void* stream);

/* The same for TWO groups of problems that differ in the kernel size only: ka problems of geometry ga (3x3) followed by kb
 * of geometry gb (5x5) -- dep_sep_conv_3 and dep_sep_conv_5 of the same edges (utils/operations.py:73-76) share every
 * launch; ka + kb <= SENAS_MAX_DWMULTI, gb == NULL with kb == 0 is the single-group form.  bwd_weight: defer == NULL sums at
 * once, else leaves the ka + kb second stages to senas_wgrad_sum_batched.                                                 */
int senas_dwconv_pair_fwd(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb, const float* x,
                          const float* const* w, float* const* y, double* const* stats, void* stream);
int senas_dwconv_pair_bwd_data(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb, const float* const* dy,
                               const float* const* w, float* dx, void* stream);
/* ... and with one input PER PROBLEM (xs: HOST array of ka + kb device pointers; NULL: all read x): the candidates of BOTH
 * input states of a search cell (search/cell.py:81-93: the edges from s0 and from s1 have one geometry) share the forward
 * and the weight-gradient launch; the data gradient stays one launch per input (senas_dwconv_pair_bwd_data on its problems). */
int senas_dwconv_pair_fwd_xs(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb, const float* x,
                             const float* const* xs, const float* const* w, float* const* y, double* const* stats, void* stream);
int senas_dwconv_pair_bwd_weight_xs(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb, const float* x,
                                    const float* const* xs, const float* const* dy, float* const* dw, void* ws,
                                    senas_sum_item* defer, void* stream);
int64_t senas_dwconv_pair_ws_bytes(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb);
int senas_dwconv_pair_bwd_weight(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb, const float* x,
                                 const float* const* dy, float* const* dw, void* ws, senas_sum_item* defer, void* stream);

/* ---- k independent pointwise convolutions of one shape ------------------------------------------------------------
 * The 1x1 halves of the DepSepConv candidates (utils/operations.py:107-115) of the edges that leave one state: each
 * has its own input x[p] [n][hw][cin], weights w[p] [cout][cin] and output y[p] [n][hw][cout]; one launch forward (stats[p]:
 * producer-side statistics as in senas_conv2d_fwd, may be NULL), one for all data gradients (dx[p] NULL: skipped), two
 * for all weight gradients.  HOST arrays of k pointers, k <= SENAS_MAX_PWMULTI; cin in {4..64} a power-of-two multiple
 * of 4, cout 4 or 8; otherwise SENAS_EUNSUPPORTED and nothing is launched.                                            */
#define SENAS_MAX_PWMULTI 8
int senas_pw_multi_fwd(int k, int n, int64_t hw, int cin, int cout, const float* const* x, const float* const* w,
                       float* const* y, double* const* stats, void* stream);
int senas_pw_multi_bwd_data(int k, int n, int64_t hw, int cin, int cout, const float* const* dy, const float* const* w,
                            float* const* dx, void* stream);
int64_t senas_pw_multi_ws_bytes(int k, int n, int64_t hw, int cin, int cout);
int senas_pw_multi_bwd_weight(int k, int n, int64_t hw, int cin, int cout, const float* const* x, const float* const* dy,
                              float* const* dw, void* ws, void* stream);

/* ---- batched BatchNorm2d + ReLU over k independent tensors of one shape -------------------------------------------
 * DepSepConv's depthwise half (utils/operations.py:107-115: depthwise conv -> BatchNorm2d(c_in) -> ReLU).  The k
 * depthwise outputs that leave one state of a search cell share ONE forward launch and TWO backward launches instead of
 * 1 + 3 each.  Arithmetic as senas_node_fwd / _bwd with one term and relu.  items: HOST array of k descriptors
 * (k <= SENAS_MAX_BNRELU); tensors [n][hw][c], c % 4 == 0, c <= 64.
 *   forward : y = relu(BN(z)), mask8 (one byte per 16-byte piece of y), mean_invstd float[2][c] saved for backward;
 *             training != 0: statistics from stats (double[n][c][2], producer-side sums), running buffers updated.
 *   backward: dz, dgamma, dbeta from dy (pixel stride dy_pixel_stride floats, 0 = c), z, mask8, mean_invstd;
 *             sums: double[n][c][2] scratch, zero on entry.                                                          */
#define SENAS_MAX_BNRELU 8
typedef struct senas_bnrelu_item {
    const float* z;
    float* y;
    uint8_t* mask8;
    const double* stats;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    int64_t* num_batches_tracked;
    float* mean_invstd;
    const float* dy;
    int64_t dy_pixel_stride;
    float* dz;
    float* dgamma;
    float* dbeta;
    double* sums;
} senas_bnrelu_item;
int senas_bnrelu_multi_fwd(const senas_bnrelu_item* items, int k, int n, int64_t hw, int c, int training, float momentum,
                           float eps, void* stream);
int senas_bnrelu_multi_bwd(const senas_bnrelu_item* items, int k, int n, int64_t hw, int c, void* stream);

/* ---- the second half of k DepSepConv candidates in one pass per direction ---------------------------------------------
 * utils/operations.py:107-115: depthwise conv -> BatchNorm2d(c_in) -> ReLU -> 1x1 conv (c_in -> c_out) -> BatchNorm2d.
 * Given the depthwise outputs z1 (with their producer-side statistics), problem p computes
 *     z2 = W relu(BN1(z1))
 * with BN1 + ReLU applied on load (the activated tensor is never stored) and the statistics of z2 for the BatchNorm2d
 * that follows (applied by the cell node, senas_node_fwd).  Backward (two launches for all k problems): dz1, d gamma1,
 * d beta1 and dW from dz2, recomputing W^T dz2 and the ReLU mask per pixel instead of storing them.
 * cin a power-of-two multiple of 4 up to 64, cout 4 or 8; otherwise SENAS_EUNSUPPORTED and nothing is launched.
 * items: HOST array of k descriptors (k <= SENAS_MAX_DSTAIL).
 *   forward : reads z1, stats1 (training != 0; else running_mean1 / running_var1), gamma1, beta1, w; writes z2, stats2
 *             (ADDED into, may be NULL), mean_invstd (float[2][cin], kept for backward), the running buffers of BN1.
 *   backward: reads z1, dz2 (pixel stride dz2_pixel_stride floats, 0 = cout), w, gamma1, beta1, mean_invstd; sums:
 *             double[n][cin][2], ZERO on entry; writes dz1 (NULL: skipped), dgamma1, dbeta1 (float[cin]) and -- when dw
 *             is given (for all problems or none) -- dw [cout][cin], accumulated in dw_acc (several double[cout][cin]
 *             images per batch image: senas_dstail_ws_bytes(...) bytes per problem, ZERO on entry).                  */
#define SENAS_MAX_DSTAIL 12
typedef struct senas_dstail_item {
    const float* z1;
    const double* stats1;
    const float* gamma1;
    const float* beta1;
    float* running_mean1;
    float* running_var1;
    int64_t* num_batches_tracked1;
    float* mean_invstd;
    const float* w;
    float* z2;
    double* stats2;
    const float* dz2;
    int64_t dz2_pixel_stride;
    double* sums;
    float* dz1;
    float* dgamma1;
    float* dbeta1;
    float* dw;
    double* dw_acc;
} senas_dstail_item;
int senas_dstail_fwd(const senas_dstail_item* items, int k, int n, int64_t hw, int cin, int cout, int training,
                     float momentum, float eps, void* stream);
int64_t senas_dstail_ws_bytes(int k, int n, int64_t hw, int cin, int cout);
int senas_dstail_bwd(const senas_dstail_item* items, int k, int n, int64_t hw, int cin, int cout, void* stream);

/* ---- fused "normalise, gate, mix, add, activate" ------------------------------------------------
 * y[n,p,c] = act( sum_t coef[t][n][c] * z_t[n,p,c] + bias[n][c] (+ residual[n,p,c]) )
 * This single pass replaces, per cell node, the BatchNorm2d of every candidate op, the SE channel
 * scale (operations.py:203), the alpha-weighted MixedOp sum (search/cell.py:34-36), the
 * beta-weighted edge sum (search/cell.py:104-105), the node add of the derived cell
 * (senas_model.py:62), the residual add of BasicBlock (operations.py:266) and the node ReLU
 * (search/cell.py:107).  coef: float[nterms][n][c]; bias: float[n][c] (device);
 * z / dz: HOST arrays of nterms device pointers (copied into the launch arguments).             */
int senas_combine_fwd(int n, int64_t hw, int c, int nterms, const float* const* z, const float* coef,
                      const float* bias, const float* residual, int relu, float* y, void* stream);
/* Backward reductions: with ds = dy * (relu ? y > 0 : 1),
 *   p1[n][c]      += sum_p ds          (double)
 *   p2[t][n][c]   += sum_p ds * z_t    (double)       -- caller zeroes p1/p2.                    */
int senas_combine_bwd_reduce(int n, int64_t hw, int c, int nterms, const float* const* z, const float* dy,
                             const float* y, int relu, double* p1, double* p2, void* stream);
/* Backward apply: dz_t = a[t][n][c] * ds + b[t][n][c] * z_t + k[t][n][c]  (dz_t may be NULL: skipped)
 * ds_out (optional): receives ds itself (gradient of the residual input).                       */
int senas_combine_bwd_apply(int n, int64_t hw, int c, int nterms, const float* const* z, const float* dy,
                            const float* y, int relu, const float* a, const float* b, const float* k,
                            float* const* dz, float* ds_out, void* stream);

/* ---- one cell node in two / three launches --------------------------------------------------------
 * The production path of the fused node: the per-term bookkeeping (batch statistics -> scale/shift,
 * running-stat update of every nn.BatchNorm2d, the SE gate of se_conv_3 -- operations.py:186-203 --
 * and the alpha*beta mixing weight, search/cell.py:34-36,104) is done by one small "prepare" kernel
 * instead of dozens of elementwise launches; senas_combine_* above stay as the building blocks.
 *
 * Term t (t < nterms <= SENAS_MAX_TERMS) is described column-wise; z[t] == NULL marks the all-zero
 * input of the 'none' op (operations.py:9,155-164), which contributes its batch-norm bias only.
 *   stats[t]      : double[n][c][2] per-image sum / sum-of-squares of z_t (NULL for a 'none' term)
 *   gamma/beta    : float[c] BatchNorm2d weight / bias
 *   running_mean, running_var, num_batches_tracked: updated in place when training != 0 (may be
 *                   NULL in training mode; required in eval mode)
 *   se_w1[t]      : float[mid][c] or NULL (no SE);  se_w2[t]: float[c][mid];  se_mid[t] <= 16
 *   mix           : device float[nterms] mixing weights, or NULL for all ones                      */
typedef struct senas_node_desc {
    int32_t nterms, n, c, training, relu;
    int64_t hw;
    float eps, momentum;
    const double* stats[SENAS_MAX_TERMS];
    const float* gamma[SENAS_MAX_TERMS];
    const float* beta[SENAS_MAX_TERMS];
    float* running_mean[SENAS_MAX_TERMS];
    float* running_var[SENAS_MAX_TERMS];
    int64_t* num_batches_tracked[SENAS_MAX_TERMS];
    const float* se_w1[SENAS_MAX_TERMS];
    const float* se_w2[SENAS_MAX_TERMS];
    int32_t se_mid[SENAS_MAX_TERMS];
    int32_t stats_image_stride[SENAS_MAX_TERMS];   /* doubles between the statistics of consecutive images; 0 = 2c (dense) */
    const float* mix;
} senas_node_desc;

/* y = act(sum_t mix_t * gate_t * BN_t(z_t) + residual).
 * z_pixel_stride (may be NULL = all dense): floats between consecutive pixels of z_t; larger than c when z_t is a channel
 * slice of a wider NHWC tensor -- the per-edge part of a stacked convolution output (search/cell.py:100-106 sums the
 * candidates of several edges; their same-named convolutions run as one), read in place together with the matching
 * slice of the stacked statistics (stats_image_stride).  Saved for backward (caller-allocated):
 *   coefs  float[nterms][4][c]  (mean, invstd, scale, shift)      gate   float[nterms][n][c]
 *   coef   float[nterms][n][c]  shiftc float[nterms][n][c]        (scratch of the forward pass)
 *   se_m   float[nterms][n][c], se_a1 float[nterms][n][16]        (only touched for SE terms; may be NULL without)
 *   mask8  uint8[n*hw*c/4] or NULL: with relu and c % 4 == 0, byte k holds (y > 0) of the 4 floats of 16-byte
 *          piece k in bits 0..3 -- lets the backward pass read 1 byte where it would read 16 of y
 *   out_stats (optional, c = 4 * 2^k): double[n][c][2], the per-image channel sums of y itself are ADDED into it (caller
 *          zeroes) -- the statistics the BatchNorm2d of an 'identity' candidate reading this node needs
 *   y2 (optional): y is ALSO written as a channel slice of a wider NHWC tensor (y2_pixel_stride floats between pixels) --
 *          the node's place in the concatenation the cell's post-process convolution reads (models/senas_model.py:64),
 *          instead of a torch.cat copy; y may then be NULL (a node that nothing but the concatenation reads);
 *          y2_zero_pad zero channels are written behind the slice (search/cell.py:110: the 24-channel concatenation of a
 *          search cell is padded to a full 32-channel tile for the post-process convolution)                              */
int senas_node_fwd(const senas_node_desc* desc, const float* const* z, const int32_t* z_pixel_stride, const float* residual, float* y,
                   float* coefs, float* gate, float* coef, float* shiftc, float* se_m, float* se_a1, uint8_t* mask8,
                   double* out_stats, float* y2, int64_t y2_pixel_stride, int y2_zero_pad, void* stream);
/* Backward of the above.  p1: double[n][c], p2: double[nterms][n][c], both ZEROED by the caller.
 *   dgamma[t], dbeta[t]: float[c] destinations, one pair per term (host arrays of device pointers);
 *   dmix: float[nterms] or NULL, overwritten -- or, with dmix_accumulate != 0, added to: the cells of one kind share
 *   their mixing weights (search/senas_search.py:252-259 computes them once per forward), so their nodes sum
 *   d loss / d mix into ONE buffer in stream order; dse_w1[t] / dse_w2[t]: like se_w1 / se_w2
 *   abk: float[3][nterms][n][c] scratch; dz[t]: gradient of z_t or NULL (skipped); ds_out: gradient of
 *   the residual input or NULL.  With relu, the mask comes from mask8 (as written by senas_node_fwd) if
 *   given, else from y; one of the two must be non-NULL.
 *   dy_pixel_stride: floats between consecutive pixels of dy (0 or c: dense NHWC; larger: dy is a channel slice of a
 *   wider NHWC tensor, e.g. the gradient of a torch.cat along channels -- read in place, no copy).             */
int senas_node_bwd(const senas_node_desc* desc, const float* const* z, const int32_t* z_pixel_stride, const float* dy, int64_t dy_pixel_stride,
                   const float* y, const uint8_t* mask8, const float* coefs, const float* gate, const float* se_m, const float* se_a1,
                   double* p1, double* p2, float* const* dgamma, float* const* dbeta, float* dmix, int dmix_accumulate,
                   float* const* dse_w1, float* const* dse_w2, float* abk, float* const* dz, const int32_t* dz_pixel_stride,
                   float* ds_out, void* stream);

/* out = sum of n dense fp32 tensors of numel elements (n <= SENAS_MAX_TERMS, 16-byte aligned): the gradient of a
 * tensor with n consumers (a cell state feeding several edges) in one pass instead of n-1 binary accumulations.  */
int senas_sum_n(int n, int64_t numel, const float* const* srcs, float* out, void* stream);
/* The same for NHWC tensors [npix][c] whose sources may be channel slices of wider tensors: src_pixel_stride[k] floats
 * between consecutive pixels of source k (>= c, multiple of 4) -- the gradient that torch.cat along channels hands one of
 * its inputs (models/senas_model.py:64, search/cell.py:110 concatenate the node outputs) is read in place.            */
int senas_sum_n_strided(int n, int64_t npix, int c, const float* const* srcs, const int32_t* src_pixel_stride, float* out,
                        void* stream);

/* ---- loss and metric ------------------------------------------------
 * DiceCrossEntropyLoss (utils/loss/loss.py:45-70,124-228): w_ce * CrossEntropy(mean over pixels) +
 * w_dice * (1 - mean over classes [1:] (or [0:] with do_bg) of (2 tp + smooth) / (2 tp + fp + fn + smooth + 1e-8))
 * with soft tp/fp/fn over batch and space.  logits: float [npix][c] (NHWC), c <= 8; target: int64 [npix].
 *   acc : double[1 + 3c], ZEROED by the caller (sum of -log p[t], sum p_c, sum p_c [t=c], sum [t=c])
 *   loss: float[1];  coef: float[3c + 1], saved for senas_dice_ce_bwd
 * bwd: dlogits = dloss[0] (1 if NULL) * d loss / d logits.                                                     */
int senas_dice_ce_fwd(int64_t npix, int c, const float* logits, const int64_t* target, float w_ce, float w_dice,
                      float smooth, int do_bg, double* acc, float* loss, float* coef, void* stream);
int senas_dice_ce_bwd(int64_t npix, int c, const float* logits, const int64_t* target, const float* coef,
                      const float* dloss, float* dlogits, void* stream);
/* SegmentationMetric.update (utils/metrics.py:127-173) without host syncs: arg-max over c, then
 *   counts[c-1][3] (int64: tp, fp, fn of classes 1..c-1) += this batch
 *   acc_sum[0] += mean over images of (correct_i + eps) / (labelled_i + eps)      (mean_pix_accuracy, :127-142)
 * part_zeroed: (2n + 3(c-1)) x 8 bytes of zeroed scratch.                                                    */
int senas_seg_metric_update(int n, int64_t hw, int c, const float* logits, const int64_t* target, float eps,
                            void* part_zeroed, int64_t* counts, double* acc_sum, void* stream);

/* ---- optimizer step -------------------------------------------------------------------------
 * nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.SGD.step() (momentum, dampening,
 * weight decay, nesterov; experiments/train_model.py:284-289, search_arc.py:280-285) over n tensors in two
 * launches (three with clipping).  items_dev: DEVICE array; grad == NULL: the tensor is skipped (like a parameter
 * without .grad); buf: momentum buffer (ignored when momentum == 0); first != 0: this tensor's buffer is initialised
 * with the gradient (torch's first step for it).  partial64: double[1 + n * ceil(max_numel / 1024)] scratch (one slot
 * per block, folded in a fixed order: the norm is bit-identical on every rank).  max_norm <= 0: no clipping.
 * first_step != 0: every tensor is treated as first.  Gradients are left scaled by the clip coefficient, as
 * clip_grad_norm_ leaves them.  total_norm_out: float[1] or NULL.                                                  */
typedef struct senas_sgd_item {
    float* param;
    float* grad;
    float* buf;
    int64_t numel;
    int64_t first;
} senas_sgd_item;
int senas_sgd_clip_step(const senas_sgd_item* items_dev, int n, int64_t max_numel, double* partial64,
                        float max_norm, float lr, float momentum, float dampening, float weight_decay,
                        int nesterov, int first_step, float* total_norm_out, void* stream);

/* Name of the kernel a convolution call dispatches to, as rocprofv3 prints the symbol (without the
 * `senas::` prefix and argument list).  which: 0 forward, 1 data gradient, 2 weight gradient.
 * For attributing measured time to profile rows; the string is owned by the library.              */
const char* senas_conv2d_kernel_name(const senas_conv_geom* g, int which);

/* ---- misc ---------------------------------------------------------------------------------- */
const char* senas_last_error(void);
int senas_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SENAS_HIP_H */
