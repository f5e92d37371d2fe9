// SAH kd-tree build on the device (rtx_kd_build, include/rtx.h) — included at the end of rt_hip.hip.
//
// The reference builds the tree by recursion, one node at a time, with one pass over the node's
// primitives per candidate plane (src/rt/accelerators/kdtree.c:244-383: 3 axes x ~31 planes).  A node's
// decision depends only on its own box and primitive list, so all nodes of one LEVEL are independent:
//
//   per level   k_kd_decide   one 256-thread block per node: candidate planes (the reference's f32
//                             accumulation loop), both histograms of every axis in LDS (one pass over
//                             the primitives), the SAH of every candidate in parallel, then the
//                             reference's strict-'<' scan over (axis, plane) in its order
//               k_kd_scan     exclusive sums over the level's nodes: children, child references,
//                             leaf references
//               k_kd_emit     one block per node: a leaf copies its list to the leaf pool; an inner
//                             node partitions its list STABLY into the children's lists
//                             (kdtree.c:331-347 keeps the order of the parent's list) and emits
//                             the two child records
//   at the end  the topology goes to the host, which numbers it in the reference's order
//               (BuildParentNode, kdtree.c:137-171: parent, RIGHT subtree, LEFT subtree) — a linear
//               pass over the nodes.
//
// All f32 arithmetic is the reference's, operation by operation (-ffp-contract=off, IEEE division);
// the counts are integers.  tests/test_kd_build_gpu.py: node words and leaf lists identical to the
// host build's (which tests/test_host_build.py pins to dumps of the reference's own trees).

#define KD_MIN_LEAF_LOAD 4u            // kdtree.c:15-37
#define KD_NUM_BUCKETS 32
#define KD_INTERSECT_COST (1.0f)
#define KD_RIGHT_NODE_RELATIVE_COST (0.95f)
#define KD_EMPTY_BONUS (0.5f)
#define KD_TRAVERSAL_COST (1.0f)
#define KD_LEFT_NODE_RELATIVE_COST (1.0f + (1.0f - KD_RIGHT_NODE_RELATIVE_COST))
#define KD_MAX_CAND (2 * KD_NUM_BUCKETS + 2) // cap of the candidate loop (rt_host.cpp's, never reached by sane boxes)

struct KdNode { // a node of the level being built
    float    lo[3], hi[3];
    uint32_t ref_begin, ref_count;
};
struct KdDecision {
    uint32_t split_bits; // inner: the plane
    uint32_t axis;       // 0..2, 3 = leaf
    uint32_t n_left, n_right;
};
struct KdTopo { // the finished node, by creation order
    uint32_t w0;       // plane bits | leaf length
    uint32_t axis;     // 0..2, 3 = leaf
    uint32_t left;     // inner: creation index of the left child (right = left - 1) | leaf: offset in the leaf pool
    uint32_t pad;
};

DEV float kd_surface_area(const float lo[3], const float hi[3]) // kdtree.c:198-205
{
    const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return 2 * x * y + 2 * x * z + 2 * y * z;
}

// KD_BLOCK: 256 threads per node on wide levels, 1024 on the narrow top levels, whose lists are long
// A long list cut into segments of KD_SEG references, one block each (the narrow top levels: the root's list is every
// primitive of the scene, and one block per NODE left 255 of the 256 CUs idle for a third of the build's time)
#define KD_SEG 4096u
#define KD_HIST_WORDS (3 * (KD_MAX_CAND + 2)) // per node and histogram
struct KdSeg {
    uint32_t node, begin, count, first; // node of the level, range within its list, index of the node's first segment
};

// one pass over references [i0, i1) of a node's list: a = #{planes <= box.max}, g = #{planes < box.min} per axis, into LDS
template <int KD_BLOCK>
DEV void kd_histogram_pass(const KdNode& nd, const uint32_t* __restrict__ refs, const float* __restrict__ boxes, uint32_t i0, uint32_t i1,
                           const float (*s_cand)[KD_MAX_CAND], const int* s_k, uint32_t (*s_a)[KD_MAX_CAND + 2], uint32_t (*s_g)[KD_MAX_CAND + 2])
{
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += KD_BLOCK) {
        const float* bx = boxes + 6ull * refs[nd.ref_begin + i];
        for (int axis = 0; axis < 3; axis++) {
            const int k = s_k[axis];
            if (k == 0) continue;
            const float* cand = s_cand[axis];
            const float  inv  = 1.0f / ((nd.hi[axis] - nd.lo[axis]) / KD_NUM_BUCKETS);
            const float  bmin = bx[axis], bmax = bx[3 + axis];
            int          a    = (int)((bmax - cand[0]) * inv) + 1; // a guess; the two loops make it exact
            a                 = a < 0 ? 0 : (a > k ? k : a);
            while (a < k && cand[a] <= bmax) a++;
            while (a > 0 && !(cand[a - 1] <= bmax)) a--;
            int g = (int)((bmin - cand[0]) * inv) + 1;
            g     = g < 0 ? 0 : (g > k ? k : g);
            while (g < k && cand[g] < bmin) g++;
            while (g > 0 && !(cand[g - 1] < bmin)) g--;
            atomicAdd(&s_a[axis][a], 1u);
            atomicAdd(&s_g[axis][g], 1u);
        }
    }
}

// candidate planes of one axis (thread `axis` of the block): kdtree.c:299-301, `bucket += stride` in f32
DEV void kd_candidates(const KdNode& nd, int axis, float (*s_cand)[KD_MAX_CAND], int* s_k)
{
    const float cmin = nd.lo[axis], cmax = nd.hi[axis];
    const float stride = (cmax - cmin) / KD_NUM_BUCKETS;
    int         k = 0;
    if (stride > 0.0f && cmin + stride != cmin) {
        for (float bucket = cmin + stride; bucket <= cmax - stride; bucket += stride) {
            if (k == KD_MAX_CAND) break;
            s_cand[axis][k++] = bucket;
        }
    }
    s_k[axis] = k;
}

// narrow levels, first half of k_kd_decide's work: the histograms of ONE SEGMENT of a node's list, added to the node's
// histograms in global memory (integer counts: the order of the additions does not matter)
__global__ __launch_bounds__(1024) void k_kd_hist_seg(const KdNode* __restrict__ nodes, const uint32_t* __restrict__ refs, const float* __restrict__ boxes,
                                                      const KdSeg* __restrict__ segs, uint32_t depth_left, uint32_t* __restrict__ g_hist)
{
    __shared__ float    s_cand[3][KD_MAX_CAND];
    __shared__ int      s_k[3];
    __shared__ uint32_t s_a[3][KD_MAX_CAND + 2], s_g[3][KD_MAX_CAND + 2];
    const KdSeg    sg  = segs[blockIdx.x];
    const KdNode   nd  = nodes[sg.node];
    const uint32_t tid = threadIdx.x;
    if (nd.ref_count <= KD_MIN_LEAF_LOAD || depth_left == 0u) return; // a leaf by rule (kdtree.c:286): k_kd_decide says so
    for (uint32_t i = tid; i < KD_HIST_WORDS; i += 1024) (&s_a[0][0])[i] = 0u, (&s_g[0][0])[i] = 0u;
    if (tid < 3) kd_candidates(nd, (int)tid, s_cand, s_k);
    __syncthreads();
    kd_histogram_pass<1024>(nd, refs, boxes, sg.begin, sg.begin + sg.count, s_cand, s_k, s_a, s_g);
    __syncthreads();
    uint32_t* h = g_hist + 2ull * KD_HIST_WORDS * sg.node;
    for (uint32_t i = tid; i < KD_HIST_WORDS; i += 1024) {
        const uint32_t a = (&s_a[0][0])[i], g = (&s_g[0][0])[i];
        if (a) atomicAdd(h + i, a);
        if (g) atomicAdd(h + KD_HIST_WORDS + i, g);
    }
}

// g_hist != nullptr: the node's histograms have been counted by k_kd_hist_seg (narrow levels); else they are counted here
template <int KD_BLOCK>
__global__ __launch_bounds__(KD_BLOCK) void k_kd_decide(const KdNode* __restrict__ nodes, const uint32_t* __restrict__ refs,
                                                        const float* __restrict__ boxes, uint32_t depth_left,
                                                        KdDecision* __restrict__ out, const uint32_t* __restrict__ g_hist)
{
    __shared__ float    s_cand[3][KD_MAX_CAND];
    __shared__ int      s_k[3];
    __shared__ uint32_t s_a[3][KD_MAX_CAND + 2], s_g[3][KD_MAX_CAND + 2]; // histograms, then their prefix sums
    __shared__ float    s_sah[3][KD_MAX_CAND];
    const KdNode   nd  = nodes[blockIdx.x];
    const uint32_t n   = nd.ref_count;
    const uint32_t tid = threadIdx.x;
    if (n <= KD_MIN_LEAF_LOAD || depth_left == 0u) { // kdtree.c:286
        if (tid == 0) out[blockIdx.x] = KdDecision{n, 3u, 0u, 0u};
        return;
    }
    if (g_hist) {
        const uint32_t* h = g_hist + 2ull * KD_HIST_WORDS * blockIdx.x;
        for (uint32_t i = tid; i < KD_HIST_WORDS; i += KD_BLOCK) (&s_a[0][0])[i] = h[i], (&s_g[0][0])[i] = h[KD_HIST_WORDS + i];
    } else {
        for (uint32_t i = tid; i < KD_HIST_WORDS; i += KD_BLOCK) (&s_a[0][0])[i] = 0u, (&s_g[0][0])[i] = 0u;
    }
    if (tid < 3) kd_candidates(nd, (int)tid, s_cand, s_k);
    __syncthreads();
    // one pass over the primitives: a = #{planes <= box.max}, g = #{planes < box.min} per axis
    if (!g_hist) kd_histogram_pass<KD_BLOCK>(nd, refs, boxes, 0u, n, s_cand, s_k, s_a, s_g);
    __syncthreads();
    if (tid < 3) { // inclusive prefix sums: s_a[j] = #{box.max < plane j}, s_g[j] = #{box.min <= plane j}
        uint32_t ra = 0, rg = 0;
        for (int j = 0; j < s_k[tid]; j++) {
            ra += s_a[tid][j], rg += s_g[tid][j];
            s_a[tid][j] = ra, s_g[tid][j] = rg;
        }
    }
    __syncthreads();
    const float parentSA = kd_surface_area(nd.lo, nd.hi);
    if (tid < 3 * KD_MAX_CAND) { // ComputeSplitSAH (kdtree.c:244-271) of one candidate
        const int axis = tid / KD_MAX_CAND, j = tid % KD_MAX_CAND;
        if (j < s_k[axis]) {
            const uint32_t leftPrims  = s_g[axis][j];     // n - #{box.min > plane}
            const uint32_t rightPrims = n - s_a[axis][j]; // n - #{box.max < plane}
            const float    split      = s_cand[axis][j];
            float          llo[3] = {nd.lo[0], nd.lo[1], nd.lo[2]}, lhi[3] = {nd.hi[0], nd.hi[1], nd.hi[2]};
            float          rlo[3] = {nd.lo[0], nd.lo[1], nd.lo[2]}, rhi[3] = {nd.hi[0], nd.hi[1], nd.hi[2]};
            lhi[axis] = split, rlo[axis] = split; // SplitBox, kdtree.c:232-242
            const float leftCost   = (kd_surface_area(llo, lhi) / parentSA) * (float)leftPrims * KD_INTERSECT_COST * KD_LEFT_NODE_RELATIVE_COST;
            const float rightCost  = (kd_surface_area(rlo, rhi) / parentSA) * (float)rightPrims * KD_INTERSECT_COST * KD_RIGHT_NODE_RELATIVE_COST;
            const float emptyBonus = (leftPrims == 0u || rightPrims == 0u) ? KD_EMPTY_BONUS : 0.0f;
            s_sah[axis][j]         = KD_TRAVERSAL_COST + (1.0f - emptyBonus) * (leftCost + rightCost);
        }
    }
    __syncthreads();
    if (tid == 0) { // kdtree.c:295-325: first strictly smaller SAH in (axis, plane) order, against the cost of a leaf
        float bestSAH = RTX_INF, bestSplit = 0.0f;
        int   bestAxis = 0, bestJ = -1;
        for (int axis = 0; axis < 3; axis++)
            for (int j = 0; j < s_k[axis]; j++)
                if (s_sah[axis][j] < bestSAH) bestSAH = s_sah[axis][j], bestAxis = axis, bestJ = j, bestSplit = s_cand[axis][j];
        const float parentSAH = (float)n * KD_INTERSECT_COST;
        if (parentSAH <= bestSAH) out[blockIdx.x] = KdDecision{n, 3u, 0u, 0u};
        else out[blockIdx.x] = KdDecision{__float_as_uint(bestSplit), (uint32_t)bestAxis, s_g[bestAxis][bestJ], n - s_a[bestAxis][bestJ]};
    }
}

// exclusive sums over the m nodes of a level (one block): off[0][i] children, off[1][i] child references,
// off[2][i] leaf references; totals[0..2]
__global__ __launch_bounds__(1024) void k_kd_scan(const KdDecision* __restrict__ dec, uint32_t m, uint32_t* __restrict__ off, uint64_t* totals)
{
    __shared__ uint32_t s_w[3][16];
    __shared__ uint32_t s_run[3];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 3) s_run[tid] = 0;
    __syncthreads();
    for (uint32_t base = 0; base < m; base += 1024) {
        const uint32_t i = base + tid;
        uint32_t       v[3] = {0, 0, 0};
        if (i < m) {
            const KdDecision d = dec[i];
            if (d.axis == 3u) v[2] = d.split_bits;
            else v[0] = 2u, v[1] = d.n_left + d.n_right;
        }
        uint32_t inc[3];
        for (int q = 0; q < 3; q++) {
            inc[q] = wave_scan_add(v[q]);
            if (lane == 63) s_w[q][wave] = inc[q];
        }
        __syncthreads();
        for (int q = 0; q < 3; q++) {
            uint32_t before = s_run[q];
            for (uint32_t w = 0; w < wave; w++) before += s_w[q][w];
            if (i < m) off[(uint64_t)q * m + i] = before + inc[q] - v[q];
        }
        __syncthreads();
        if (tid < 3) {
            uint32_t t = s_run[tid];
            for (int w = 0; w < 16; w++) t += s_w[tid][w];
            s_run[tid] = t;
        }
        __syncthreads();
    }
    if (tid < 3) totals[tid] = s_run[tid];
}

template <int KD_BLOCK>
__global__ __launch_bounds__(KD_BLOCK) void k_kd_emit(const KdNode* __restrict__ nodes, const uint32_t* __restrict__ refs,
                                                      const float* __restrict__ boxes, const KdDecision* __restrict__ dec,
                                                      const uint32_t* __restrict__ off, uint32_t m, KdTopo* __restrict__ topo,
                                                      uint32_t topo_base, uint32_t next_base, uint32_t* __restrict__ leaf_pool,
                                                      uint32_t leaf_base, KdNode* __restrict__ next_nodes, uint32_t* __restrict__ next_refs)
{
    __shared__ uint32_t s_wl[KD_BLOCK / 64], s_wr[KD_BLOCK / 64];
    const uint32_t   b   = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const KdNode     nd  = nodes[b];
    const KdDecision d   = dec[b];
    const uint32_t   n   = nd.ref_count;
    if (d.axis == 3u) { // BuildLeafNode, kdtree.c:173-196
        const uint32_t at = leaf_base + off[2ull * m + b];
        for (uint32_t i = tid; i < n; i += KD_BLOCK) leaf_pool[at + i] = refs[nd.ref_begin + i];
        if (tid == 0) topo[topo_base + b] = KdTopo{n, 3u, at, 0u};
        return;
    }
    const uint32_t child  = off[b];                 // children: [child] = right, [child + 1] = left
    const uint32_t rbegin = off[1ull * m + b], lbegin = rbegin + d.n_right;
    const float    split  = __uint_as_float(d.split_bits);
    if (tid == 0) {
        topo[topo_base + b] = KdTopo{d.split_bits, d.axis, next_base + child + 1u, 0u};
        KdNode r = nd, l = nd;
        r.lo[d.axis] = split, l.hi[d.axis] = split; // SplitBox
        r.ref_begin = rbegin, r.ref_count = d.n_right;
        l.ref_begin = lbegin, l.ref_count = d.n_left;
        next_nodes[child]     = r;
        next_nodes[child + 1] = l;
    }
    // kdtree.c:331-347, order kept: left unless box.min > plane, right unless box.max < plane
    uint32_t doneL = 0, doneR = 0; // block-uniform
    for (uint32_t base = 0; base < n; base += KD_BLOCK) {
        const uint32_t i = base + tid;
        uint32_t       r = 0;
        bool           fl = false, fr = false;
        if (i < n) {
            r               = refs[nd.ref_begin + i];
            const float* bx = boxes + 6ull * r;
            const float  bmin = bx[d.axis], bmax = bx[3 + d.axis];
            if (bmax < split) fl = true;
            else if (bmin > split) fr = true;
            else fl = fr = true;
        }
        const uint64_t ml = __ballot(fl), mr = __ballot(fr);
        const uint64_t lt = (1ull << lane) - 1ull;
        if (lane == 0) s_wl[wave] = (uint32_t)__popcll(ml), s_wr[wave] = (uint32_t)__popcll(mr);
        __syncthreads();
        uint32_t pl = doneL + (uint32_t)__popcll(ml & lt), pr = doneR + (uint32_t)__popcll(mr & lt);
        uint32_t tl = 0, tr = 0;
        for (uint32_t w = 0; w < KD_BLOCK / 64; w++) {
            if (w < wave) pl += s_wl[w], pr += s_wr[w];
            tl += s_wl[w], tr += s_wr[w];
        }
        if (fl && pl < d.n_left) next_refs[lbegin + pl] = r;   // (the counts always match for finite boxes with min <= max,
        if (fr && pr < d.n_right) next_refs[rbegin + pr] = r; //  which rtx_kd_build insists on; the bounds keep a bug from faulting)
        doneL += tl, doneR += tr;
        __syncthreads();
    }
}

// narrow levels, k_kd_emit by segments.  First the number of references a segment sends left and right ...
__global__ __launch_bounds__(1024) void k_kd_count_seg(const KdNode* __restrict__ nodes, const uint32_t* __restrict__ refs, const float* __restrict__ boxes,
                                                       const KdDecision* __restrict__ dec, const KdSeg* __restrict__ segs, uint2* __restrict__ seg_cnt)
{
    __shared__ uint32_t s_l, s_r;
    const KdSeg      sg = segs[blockIdx.x];
    const KdNode     nd = nodes[sg.node];
    const KdDecision d  = dec[sg.node];
    if (threadIdx.x == 0) s_l = 0u, s_r = 0u;
    __syncthreads();
    if (d.axis != 3u) {
        const float split = __uint_as_float(d.split_bits);
        uint32_t    cl = 0, cr = 0;
        for (uint32_t i = sg.begin + threadIdx.x; i < sg.begin + sg.count; i += 1024) {
            const float* bx = boxes + 6ull * refs[nd.ref_begin + i];
            const float  bmin = bx[d.axis], bmax = bx[3 + d.axis];
            if (bmax < split) cl++;
            else if (bmin > split) cr++;
            else cl++, cr++;
        }
        for (int off = 32; off > 0; off >>= 1) cl += __shfl_down(cl, off), cr += __shfl_down(cr, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_l, cl), atomicAdd(&s_r, cr);
    }
    __syncthreads();
    if (threadIdx.x == 0) seg_cnt[blockIdx.x] = make_uint2(s_l, s_r);
}

// ... then the stable partition (kdtree.c:331-347), every segment starting behind what the node's earlier segments send
__global__ __launch_bounds__(1024) void k_kd_emit_seg(const KdNode* __restrict__ nodes, const uint32_t* __restrict__ refs, const float* __restrict__ boxes,
                                                      const KdDecision* __restrict__ dec, const uint32_t* __restrict__ off, uint32_t m, KdTopo* __restrict__ topo,
                                                      uint32_t topo_base, uint32_t next_base, uint32_t* __restrict__ leaf_pool, uint32_t leaf_base,
                                                      KdNode* __restrict__ next_nodes, uint32_t* __restrict__ next_refs, const KdSeg* __restrict__ segs,
                                                      const uint2* __restrict__ seg_cnt)
{
    __shared__ uint32_t s_wl[16], s_wr[16];
    const KdSeg      sg  = segs[blockIdx.x];
    const uint32_t   b   = sg.node, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const KdNode     nd  = nodes[b];
    const KdDecision d   = dec[b];
    const uint32_t   n   = nd.ref_count;
    const bool       head = blockIdx.x == sg.first; // the node's first segment writes the node itself
    if (d.axis == 3u) { // BuildLeafNode, kdtree.c:173-196
        const uint32_t at = leaf_base + off[2ull * m + b];
        for (uint32_t i = sg.begin + tid; i < sg.begin + sg.count; i += 1024) leaf_pool[at + i] = refs[nd.ref_begin + i];
        if (head && tid == 0) topo[topo_base + b] = KdTopo{n, 3u, at, 0u};
        return;
    }
    const uint32_t child  = off[b];
    const uint32_t rbegin = off[1ull * m + b], lbegin = rbegin + d.n_right;
    const float    split  = __uint_as_float(d.split_bits);
    if (head && tid == 0) {
        topo[topo_base + b] = KdTopo{d.split_bits, d.axis, next_base + child + 1u, 0u};
        KdNode r = nd, l = nd;
        r.lo[d.axis] = split, l.hi[d.axis] = split; // SplitBox
        r.ref_begin = rbegin, r.ref_count = d.n_right;
        l.ref_begin = lbegin, l.ref_count = d.n_left;
        next_nodes[child]     = r;
        next_nodes[child + 1] = l;
    }
    uint32_t doneL = 0, doneR = 0; // block-uniform: what the node's earlier segments send left and right
    for (uint32_t q = sg.first; q < blockIdx.x; q++) doneL += seg_cnt[q].x, doneR += seg_cnt[q].y;
    for (uint32_t base = sg.begin; base < sg.begin + sg.count; base += 1024) {
        const uint32_t i = base + tid;
        uint32_t       r = 0;
        bool           fl = false, fr = false;
        if (i < sg.begin + sg.count) {
            r               = refs[nd.ref_begin + i];
            const float* bx = boxes + 6ull * r;
            const float  bmin = bx[d.axis], bmax = bx[3 + d.axis];
            if (bmax < split) fl = true;
            else if (bmin > split) fr = true;
            else fl = fr = true;
        }
        const uint64_t ml = __ballot(fl), mr = __ballot(fr);
        const uint64_t lt = (1ull << lane) - 1ull;
        if (lane == 0) s_wl[wave] = (uint32_t)__popcll(ml), s_wr[wave] = (uint32_t)__popcll(mr);
        __syncthreads();
        uint32_t pl = doneL + (uint32_t)__popcll(ml & lt), pr = doneR + (uint32_t)__popcll(mr & lt);
        uint32_t tl = 0, tr = 0;
        for (uint32_t w = 0; w < 16; w++) {
            if (w < wave) pl += s_wl[w], pr += s_wr[w];
            tl += s_wl[w], tr += s_wr[w];
        }
        if (fl && pl < d.n_left) next_refs[lbegin + pl] = r;
        if (fr && pr < d.n_right) next_refs[rbegin + pr] = r;
        doneL += tl, doneR += tr;
        __syncthreads();
    }
}

// a device array that grows (contents kept)
struct KdBuf {
    void*  p     = nullptr;
    size_t bytes = 0;
    ~KdBuf() { if (p) (void)hipFree(p); }
    hipError_t need(size_t want, size_t keep)
    {
        if (want <= bytes) return hipSuccess;
        size_t cap = std::max(want, bytes * 2);
        void*  q   = nullptr;
        hipError_t e = hipMalloc(&q, cap);
        if (e != hipSuccess) return e;
        if (keep) e = hipMemcpy(q, p, keep, hipMemcpyDeviceToDevice);
        if (p) (void)hipFree(p);
        p = q, bytes = cap;
        return e;
    }
    template <typename T> T* as() { return (T*)p; }
};

static double kd_now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

extern "C" void rtx_kd_free(rtx_kd_tree* t)
{
    if (!t) return;
    free(t->node_w0), free(t->node_w1), free(t->leaf_ref);
    memset(t, 0, sizeof *t);
}

extern "C" int rtx_kd_build(int device, const float* box_min_max, uint64_t n, uint64_t max_depth, const float* world_min_max, rtx_kd_tree* out)
{
    if (!box_min_max || !world_min_max || !out || n == 0) return fail(RTX_E_INVALID, "rtx_kd_build: null argument or no boxes");
    if (n > 0x3FFFFFFFull) return fail(RTX_E_LIMIT, "rtx_kd_build: more than 2^30 primitives (kdtree.c:412)");
    memset(out, 0, sizeof *out);
    for (uint64_t i = 0; i < n; i++) { // the kernels assume finite boxes with min <= max (CreateKDBBs pads by +-1e-4: min < max)
        const float* b = box_min_max + 6 * i;
        for (int a = 0; a < 3; a++)
            if (!(b[a] <= b[3 + a]) || !std::isfinite(b[a]) || !std::isfinite(b[3 + a]))
                return fail(RTX_E_INVALID, "rtx_kd_build: box %llu is not finite or has min > max", (unsigned long long)i);
    }
    for (int a = 0; a < 3; a++)
        if (!(world_min_max[a] <= world_min_max[3 + a]) || !std::isfinite(world_min_max[a]) || !std::isfinite(world_min_max[3 + a]))
            return fail(RTX_E_INVALID, "rtx_kd_build: the world box is not finite or has min > max");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RTX_E_NO_DEVICE, "rtx_kd_build: no HIP device visible");
    if (device < 0 || device >= ndev) return fail(RTX_E_NO_DEVICE, "rtx_kd_build: device %d of %d", device, ndev);
    HIPCHECK(hipSetDevice(device));
    const double t0 = kd_now_ms();

    KdBuf boxes, nodes[2], refs[2], dec, off, totals, topo, pool, segs, seg_cnt, hist;
    HIPCHECK(boxes.need(n * 24, 0));
    HIPCHECK(hipMemcpy(boxes.p, box_min_max, n * 24, hipMemcpyHostToDevice));
    HIPCHECK(totals.need(3 * sizeof(uint64_t), 0));
    HIPCHECK(nodes[0].need(sizeof(KdNode), 0));
    HIPCHECK(refs[0].need(n * 4, 0));
    {
        KdNode root;
        for (int a = 0; a < 3; a++) root.lo[a] = world_min_max[a], root.hi[a] = world_min_max[3 + a];
        root.ref_begin = 0, root.ref_count = (uint32_t)n;
        HIPCHECK(hipMemcpy(nodes[0].p, &root, sizeof root, hipMemcpyHostToDevice));
        std::vector<uint32_t> all(n);
        for (uint64_t i = 0; i < n; i++) all[i] = (uint32_t)i;
        HIPCHECK(hipMemcpy(refs[0].p, all.data(), n * 4, hipMemcpyHostToDevice));
    }
    const double t1 = kd_now_ms();
    uint64_t m = 1, n_topo = 0, n_pool = 0, level = 0, level_refs = n; // level_refs: references in the lists of the level
    int      cur = 0;
    for (;; level++) {
        if (m > 0x7FFFFFFFull) return fail(RTX_E_LIMIT, "rtx_kd_build: level too wide");
        HIPCHECK(dec.need(m * sizeof(KdDecision), 0));
        HIPCHECK(off.need(3 * m * 4, 0));
        HIPCHECK(topo.need((n_topo + m) * sizeof(KdTopo), n_topo * sizeof(KdTopo)));
        const uint32_t depth_left = (uint32_t)(max_depth > level ? max_depth - level : 0);
        const bool narrow = m <= 4096; // few nodes, long lists: four times the threads per node
        // The top levels — a handful of nodes with lists of tens of thousands of references — are cut into segments of KD_SEG
        // references, a block each, instead of one block per node (root of the 278 k-triangle stand-in: 0.8 ms on one CU).
        std::vector<KdSeg> sv;
        if (m <= 512 && level_refs > 2ull * KD_SEG * m / 4 + KD_SEG) {
            std::vector<KdNode> hn(m);
            HIPCHECK(hipMemcpy(hn.data(), nodes[cur].p, m * sizeof(KdNode), hipMemcpyDeviceToHost));
            uint32_t longest = 0;
            for (const KdNode& k : hn) longest = std::max(longest, k.ref_count);
            if (longest > 2 * KD_SEG) {
                for (uint32_t b = 0; b < (uint32_t)m; b++) {
                    const uint32_t first = (uint32_t)sv.size(), cnt = hn[b].ref_count;
                    if (cnt == 0) sv.push_back(KdSeg{b, 0u, 0u, first});
                    for (uint32_t at = 0; at < cnt; at += KD_SEG) sv.push_back(KdSeg{b, at, std::min(KD_SEG, cnt - at), first});
                }
            }
        }
        const bool segmented = !sv.empty();
        if (segmented) {
            HIPCHECK(segs.need(sv.size() * sizeof(KdSeg), 0));
            HIPCHECK(seg_cnt.need(sv.size() * sizeof(uint2), 0));
            HIPCHECK(hist.need(m * 2ull * KD_HIST_WORDS * 4, 0));
            HIPCHECK(hipMemcpy(segs.p, sv.data(), sv.size() * sizeof(KdSeg), hipMemcpyHostToDevice));
            HIPCHECK(hipMemsetAsync(hist.p, 0, m * 2ull * KD_HIST_WORDS * 4, 0));
            hipLaunchKernelGGL(k_kd_hist_seg, dim3((unsigned)sv.size()), dim3(1024), 0, 0, nodes[cur].as<KdNode>(), refs[cur].as<uint32_t>(), boxes.as<float>(),
                               segs.as<KdSeg>(), depth_left, hist.as<uint32_t>());
            hipLaunchKernelGGL(k_kd_decide<256>, dim3((unsigned)m), dim3(256), 0, 0, nodes[cur].as<KdNode>(), refs[cur].as<uint32_t>(), boxes.as<float>(), depth_left,
                               dec.as<KdDecision>(), (const uint32_t*)hist.as<uint32_t>());
        } else {
            hipLaunchKernelGGL(narrow ? k_kd_decide<1024> : k_kd_decide<256>, dim3((unsigned)m), dim3(narrow ? 1024 : 256), 0, 0, nodes[cur].as<KdNode>(),
                               refs[cur].as<uint32_t>(), boxes.as<float>(), depth_left, dec.as<KdDecision>(), (const uint32_t*)nullptr);
        }
        hipLaunchKernelGGL(k_kd_scan, dim3(1), dim3(1024), 0, 0, dec.as<KdDecision>(), (uint32_t)m, off.as<uint32_t>(), totals.as<uint64_t>());
        HIPCHECK(hipGetLastError());
        uint64_t tot[3];
        HIPCHECK(hipMemcpy(tot, totals.p, sizeof tot, hipMemcpyDeviceToHost)); // children, child references, leaf references
        if (n_topo + m + tot[0] > 0x40000000ull || n_pool + tot[2] > 0x40000000ull || tot[1] > 0xFFFFFFF0ull)
            return fail(RTX_E_LIMIT, "rtx_kd_build: more than 2^30 nodes or leaf references (kdtree.c:280,412)");
        HIPCHECK(nodes[cur ^ 1].need(std::max<uint64_t>(tot[0], 1) * sizeof(KdNode), 0));
        HIPCHECK(refs[cur ^ 1].need(std::max<uint64_t>(tot[1], 1) * 4, 0));
        HIPCHECK(pool.need(std::max<uint64_t>(n_pool + tot[2], 1) * 4, n_pool * 4));
        if (segmented) {
            hipLaunchKernelGGL(k_kd_count_seg, dim3((unsigned)sv.size()), dim3(1024), 0, 0, nodes[cur].as<KdNode>(), refs[cur].as<uint32_t>(), boxes.as<float>(),
                               dec.as<KdDecision>(), segs.as<KdSeg>(), seg_cnt.as<uint2>());
            hipLaunchKernelGGL(k_kd_emit_seg, dim3((unsigned)sv.size()), dim3(1024), 0, 0, nodes[cur].as<KdNode>(), refs[cur].as<uint32_t>(), boxes.as<float>(),
                               dec.as<KdDecision>(), off.as<uint32_t>(), (uint32_t)m, topo.as<KdTopo>(), (uint32_t)n_topo, (uint32_t)(n_topo + m),
                               pool.as<uint32_t>(), (uint32_t)n_pool, nodes[cur ^ 1].as<KdNode>(), refs[cur ^ 1].as<uint32_t>(), segs.as<KdSeg>(),
                               seg_cnt.as<uint2>());
        } else {
            hipLaunchKernelGGL(narrow ? k_kd_emit<1024> : k_kd_emit<256>, dim3((unsigned)m), dim3(narrow ? 1024 : 256), 0, 0, nodes[cur].as<KdNode>(),
                               refs[cur].as<uint32_t>(), boxes.as<float>(),
                               dec.as<KdDecision>(), off.as<uint32_t>(), (uint32_t)m, topo.as<KdTopo>(), (uint32_t)n_topo, (uint32_t)(n_topo + m),
                               pool.as<uint32_t>(), (uint32_t)n_pool, nodes[cur ^ 1].as<KdNode>(), refs[cur ^ 1].as<uint32_t>());
        }
        HIPCHECK(hipGetLastError());
        level_refs = tot[1];
        n_topo += m;
        n_pool += tot[2];
        if (tot[0] == 0) break;
        m   = tot[0];
        cur ^= 1;
    }
    HIPCHECK(hipDeviceSynchronize());
    const double t2 = kd_now_ms();

    // numbering in the reference's order (BuildParentNode, kdtree.c:137-171): a node takes the next index when it is
    // entered, its right subtree follows, then its left subtree; leaves append their lists in that order
    std::vector<KdTopo>   T(n_topo);
    std::vector<uint32_t> P(std::max<uint64_t>(n_pool, 1));
    HIPCHECK(hipMemcpy(T.data(), topo.p, n_topo * sizeof(KdTopo), hipMemcpyDeviceToHost));
    if (n_pool) HIPCHECK(hipMemcpy(P.data(), pool.p, n_pool * 4, hipMemcpyDeviceToHost));
    uint32_t* w0 = (uint32_t*)malloc(n_topo * 4);
    uint32_t* w1 = (uint32_t*)malloc(n_topo * 4);
    uint32_t* lr = (uint32_t*)malloc(std::max<uint64_t>(n_pool, 1) * 4);
    if (!w0 || !w1 || !lr) {
        free(w0), free(w1), free(lr);
        return fail(RTX_E_NOMEM, "rtx_kd_build: host allocation failed");
    }
    struct Item { uint32_t topo, parent; }; // parent: final index of the node whose LEFT child this is, or ~0
    std::vector<Item> stack;
    stack.push_back(Item{0u, 0xFFFFFFFFu});
    uint32_t next = 0, nref = 0;
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        const KdTopo& t  = T[it.topo];
        const uint32_t me = next++;
        if (it.parent != 0xFFFFFFFFu) w1[it.parent] |= me;
        w0[me] = t.w0;
        if (t.axis == 3u) {
            w1[me] = (3u << 30) | nref;
            memcpy(lr + nref, P.data() + t.left, (size_t)t.w0 * 4);
            nref += t.w0;
        } else {
            w1[me] = t.axis << 30;
            stack.push_back(Item{t.left, me});                 // left subtree: after the whole right subtree
            stack.push_back(Item{t.left - 1u, 0xFFFFFFFFu});  // right child: next index
        }
    }
    out->node_w0 = w0, out->node_w1 = w1, out->n_nodes = n_topo;
    out->leaf_ref = lr, out->n_refs = n_pool;
    out->root_index    = 0;
    out->depth_reached = level;
    out->ms_device     = t2 - t1;
    out->ms_total      = kd_now_ms() - t0;
    return RTX_OK;
}
