// The architecture tensors of the supernet in one launch per direction.
//
// Forward (search/senas_search.py:252-259 NAS.forward + search/cell.py:33-36,100-106): row softmax of the four alpha tables,
// the per-node softmax of the two beta vectors over the reference's OVERLAPPING windows [i : 2i + 2] (offset = node index,
// senas_search.py:254-257 -- kept bit for bit in structure), row softmax of gamma, and from those the per-edge mixing matrix
// of each cell kind,  M[e][o] = beta_soft[e] * alpha_soft(kind of edge e)[e][o]  (NORM edges read the *_nm table: in a down
// cell the edges from states >= 2, in an up cell all but the edge from state 1).  The cell nodes read rows of M in place and
// ADD d loss / d M into one buffer per kind; the blends read rows of softmax(gamma) and add into one fp64 table.
// Backward: from those accumulated tables straight to the gradients of the seven parameters.
// ~20 softmax / select / multiply / concatenate launches per pass (and as many again backward) become two.
#include "common.h"

namespace senas {

struct ArchMix {            // = senas_arch_mix
    const float* alpha[4];  // dn, up, dn_nm, up_nm: [k][ops]
    const float* beta[2];   // dn, up: [k]
    const float* gamma;     // [grows][2]
    float* s_alpha[4];      // softmaxes (kept for backward and for the caller's bookkeeping)
    float* s_beta[2];
    float* s_gamma;
    float* M[2];            // dn, up: [k][ops]
    const float* dM[2];     // backward: accumulated d loss / d M, [slots][k][ops] -- one slot per cell of the kind
    const double* dG;       // backward: accumulated d loss / d softmax(gamma)  [grows][2]
    float* d_alpha[4];      // backward outputs (d_alpha[3] == nullptr: up_nm IS dn_nm, its gradient is added to d_alpha[2])
    float* d_beta[2];
    float* d_gamma;
    int k, ops, nodes, grows;
    int slots;              // backward: rows of dM[kind] (the cells of a kind run on several streams: each adds into its own)
};

__device__ __forceinline__ bool edge_is_norm(int kind, int j) { return kind == 0 ? j >= 2 : j != 1; }

// edge e -> (node, input state j) for nodes with 2 + i inputs
__device__ __forceinline__ void edge_pos(int e, int& node, int& j) {
    node = 0;
    int off = 0;
    while (e >= off + 2 + node) { off += 2 + node; ++node; }
    j = e - off;
}

__global__ __launch_bounds__(256) void arch_mix_fwd_kernel(ArchMix a) {
    const int t = threadIdx.x;
    // alpha rows: thread = (table, edge)
    if (t < 4 * a.k) {
        const int tab = t / a.k, e = t % a.k;
        const float* row = a.alpha[tab] + (size_t)e * a.ops;
        float mx = row[0];
        for (int o = 1; o < a.ops; ++o) mx = fmaxf(mx, row[o]);
        float sum = 0.f;
        for (int o = 0; o < a.ops; ++o) sum += expf(row[o] - mx);
        for (int o = 0; o < a.ops; ++o) a.s_alpha[tab][(size_t)e * a.ops + o] = expf(row[o] - mx) / sum;
    }
    // beta windows: thread = (kind, node); window of node i = beta[i .. 2i + 1], written at the concatenation offset
    if (t >= 64 && t < 64 + 2 * a.nodes) {
        const int kind = (t - 64) / a.nodes, i = (t - 64) % a.nodes;
        int off = 0;
        for (int q = 0; q < i; ++q) off += 2 + q;
        const float* b = a.beta[kind] + i;
        const int cnt = i + 2;
        float mx = b[0];
        for (int w = 1; w < cnt; ++w) mx = fmaxf(mx, b[w]);
        float sum = 0.f;
        for (int w = 0; w < cnt; ++w) sum += expf(b[w] - mx);
        for (int w = 0; w < cnt; ++w) a.s_beta[kind][off + w] = expf(b[w] - mx) / sum;
    }
    if (t >= 128 && t < 128 + a.grows) {
        const int rr = t - 128;
        const float g0 = a.gamma[2 * rr], g1 = a.gamma[2 * rr + 1], mx = fmaxf(g0, g1);
        const float e0 = expf(g0 - mx), e1 = expf(g1 - mx);
        a.s_gamma[2 * rr] = e0 / (e0 + e1);
        a.s_gamma[2 * rr + 1] = e1 / (e0 + e1);
    }
    __syncthreads();
    for (int i = t; i < 2 * a.k * a.ops; i += 256) {
        const int kind = i / (a.k * a.ops), e = (i / a.ops) % a.k, o = i % a.ops;
        int node, j;
        edge_pos(e, node, j);
        const int tab = edge_is_norm(kind, j) ? 2 + kind : kind;
        a.M[kind][(size_t)e * a.ops + o] = a.s_beta[kind][e] * a.s_alpha[tab][(size_t)e * a.ops + o];
    }
}

__global__ __launch_bounds__(256) void arch_mix_bwd_kernel(ArchMix a) {
    __shared__ float dbs[2][64];          // d loss / d beta_soft
    __shared__ float dMs[2][60 * 16];     // d loss / d M per kind: the cells' slots folded in slot order (fixed: bitwise reproducible)
    const int t = threadIdx.x;
    const int ko = a.k * a.ops;
    for (int i = t; i < 2 * ko; i += 256) {
        const int kind = i / ko, idx = i % ko;
        float s = 0.f;
        for (int q = 0; q < a.slots; ++q) s += a.dM[kind][(size_t)q * ko + idx];
        dMs[kind][idx] = s;
    }
    __syncthreads();
    // per (kind, edge): d beta_soft, and the softmax backward of the alpha row the edge reads; the other table's row gets 0
    if (t < 2 * a.k) {
        const int kind = t / a.k, e = t % a.k;
        int node, j;
        edge_pos(e, node, j);
        const bool norm = edge_is_norm(kind, j);
        const int used = norm ? 2 + kind : kind, other = norm ? kind : 2 + kind;
        const float* S = a.s_alpha[used] + (size_t)e * a.ops;
        const float* dM = dMs[kind] + e * a.ops;
        const float bs = a.s_beta[kind][e];
        float db = 0.f, dot = 0.f;
        for (int o = 0; o < a.ops; ++o) { db = fmaf(dM[o], S[o], db); dot = fmaf(dM[o] * bs, S[o], dot); }
        dbs[kind][e] = db;
        // up_nm shared with dn_nm (one Parameter under two names): both kinds' NORM rows land in d_alpha[2] -- kind 0 writes, kind 1 adds
        // (below, after the barrier); unshared: each table is written by exactly one kind
        float* du = a.d_alpha[used];
        if (du != nullptr)
            for (int o = 0; o < a.ops; ++o) du[(size_t)e * a.ops + o] = S[o] * (dM[o] * bs - dot);
        float* dz = a.d_alpha[other];
        if (dz != nullptr && !(other == 2 && a.d_alpha[3] == nullptr))
            for (int o = 0; o < a.ops; ++o) dz[(size_t)e * a.ops + o] = 0.f;
    }
    __syncthreads();
    if (a.d_alpha[3] == nullptr && t < a.k) {
        // shared *_nm table: d_alpha[2][e] = (row read by the down cell, if NORM there) + (row read by the up cell, if NORM there)
        const int e = t;
        int node, j;
        edge_pos(e, node, j);
        const float* S = a.s_alpha[2] + (size_t)e * a.ops;
        float* d = a.d_alpha[2] + (size_t)e * a.ops;
        float acc[16];
        for (int o = 0; o < a.ops; ++o) acc[o] = 0.f;
        for (int kind = 0; kind < 2; ++kind) {
            if (!edge_is_norm(kind, j)) continue;
            const float* dM = dMs[kind] + e * a.ops;
            const float bs = a.s_beta[kind][e];
            float dot = 0.f;
            for (int o = 0; o < a.ops; ++o) dot = fmaf(dM[o] * bs, S[o], dot);
            for (int o = 0; o < a.ops; ++o) acc[o] += S[o] * (dM[o] * bs - dot);
        }
        for (int o = 0; o < a.ops; ++o) d[o] = acc[o];
    }
    // beta: one thread per kind walks the overlapping windows
    if (t >= 64 && t < 66) {
        const int kind = t - 64;
        float* d = a.d_beta[kind];
        for (int e = 0; e < a.k; ++e) d[e] = 0.f;
        int off = 0;
        for (int i = 0; i < a.nodes; ++i) {
            const int cnt = i + 2;
            float dot = 0.f;
            for (int w = 0; w < cnt; ++w) dot = fmaf(dbs[kind][off + w], a.s_beta[kind][off + w], dot);
            for (int w = 0; w < cnt; ++w) d[i + w] += a.s_beta[kind][off + w] * (dbs[kind][off + w] - dot);
            off += cnt;
        }
    }
    if (t >= 128 && t < 128 + a.grows) {
        const int rr = t - 128;
        const float s0 = a.s_gamma[2 * rr], s1 = a.s_gamma[2 * rr + 1];
        const float g0 = (float)a.dG[2 * rr], g1 = (float)a.dG[2 * rr + 1];
        const float dot = g0 * s0 + g1 * s1;
        a.d_gamma[2 * rr] = s0 * (g0 - dot);
        a.d_gamma[2 * rr + 1] = s1 * (g1 - dot);
    }
}

}  // namespace senas

static_assert(sizeof(senas_arch_mix) == sizeof(senas::ArchMix), "senas_arch_mix layout");

static bool arch_mix_ok(const senas_arch_mix* a, bool bwd) {
    if (!a || a->k < 2 || a->k > 60 || a->ops < 1 || a->ops > 16 || a->nodes < 1 || a->nodes > 8 || a->grows < 0 || a->grows > 120) return false;
    int k = 0;
    for (int i = 0; i < a->nodes; ++i) k += 2 + i;
    if (k != a->k) return false;
    for (int i = 0; i < 4; ++i) if (!a->alpha[i] || !a->s_alpha[i]) return false;
    for (int i = 0; i < 2; ++i) if (!a->beta[i] || !a->s_beta[i] || !a->M[i]) return false;
    if (a->grows > 0 && (!a->gamma || !a->s_gamma)) return false;
    if (bwd) {
        for (int i = 0; i < 3; ++i) if (!a->d_alpha[i]) return false;
        for (int i = 0; i < 2; ++i) if (!a->dM[i] || !a->d_beta[i]) return false;
        if (a->slots < 1 || a->slots > 256) return false;
        if (a->grows > 0 && (!a->dG || !a->d_gamma)) return false;
    }
    return true;
}

extern "C" int senas_arch_mix_fwd(const senas_arch_mix* a, void* stream) {
    SENAS_REQUIRE(arch_mix_ok(a, false), "arch_mix_fwd: bad argument");
    hipLaunchKernelGGL(senas::arch_mix_fwd_kernel, dim3(1), dim3(256), 0, senas::as_stream(stream), *reinterpret_cast<const senas::ArchMix*>(a));
    return senas::launch_status("arch_mix_fwd");
}

extern "C" int senas_arch_mix_bwd(const senas_arch_mix* a, void* stream) {
    SENAS_REQUIRE(arch_mix_ok(a, true), "arch_mix_bwd: bad argument");
    hipLaunchKernelGGL(senas::arch_mix_bwd_kernel, dim3(1), dim3(256), 0, senas::as_stream(stream), *reinterpret_cast<const senas::ArchMix*>(a));
    return senas::launch_status("arch_mix_bwd");
}
