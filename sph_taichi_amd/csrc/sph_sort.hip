// sph_sort.hip -- K1..K3 of the hot path: grid hash + histogram, wave64 prefix
// sum, STABLE counting sort.  Replaces particle_system.py:311-375 and
// scan_single_buffer.py:44-146 (reference root).
//
// The reference ranks particles with atomic_sub inside a parallel loop
// (particle_system.py:325-330); a serial run of that loop yields the stable
// order (ascending previous index inside a cell).  We reproduce exactly that
// order, deterministically, in three bandwidth-bound passes:
//   hash_histogram : key[i] = cell(x_i); off[i] = arbitrary unique offset in
//                    the cell (one returning atomic per RUN of equal cells in
//                    a wave -- particles arrive almost sorted, so ~8x fewer
//                    atomics than one per particle)
//   scan           : cell_end = inclusive prefix (ONE launch, wave64 shuffles; each tile publishes its total in an atomic
//                    word and adds up the totals of the tiles before it)
//   unstable_place : idx_unstable[start(c)+off[i]] = i
//   stable_scatter : rank of i among its cell's members by previous index,
//                    then move the 48-byte hot record (ping-pong, no copy-back)
#include "sph_internal.h"
#include "sph_bricks.h"

#define TPB 256
#define SCAN_IPT 8
#define SCAN_TILE (TPB * SCAN_IPT)

__global__ __launch_bounds__(TPB) void k_hash_histogram(DevView d, int* __restrict__ cell_cnt,
                                                        int* __restrict__ rank_off, int* __restrict__ brick_count) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (i == 0) brick_count[0] = brick_count[1] = 0;  // (heavy, light) of the list the place kernel builds on the new cell array
    int c = -1 - lane;  // inactive lanes get distinct sentinels
    if (i < d.N) {
        const float4 p = d.xm[i];
        // particle_system.py:287-298 get_flatten_grid_index
        const int cx = sph_cell_coord(p.x, d.grid_size, d.ox, d.nx);
        const int cy = sph_cell_coord(p.y, d.grid_size, d.oy, d.ny);
        const int cz = sph_cell_coord(p.z, d.grid_size, d.oz, d.nz);
        c = sph_flatten(d, cx, cy, cz);
        if (d.drop_outside) {  // slab rank: not in any local x layer -> virtual cell G, sorted behind everything
            const int raw = (int)(p.x / d.grid_size) - d.ox;
            if (raw < 0 || raw >= d.nx) c = d.G;
        }
        d.key[i] = c;  // particle_system.py:315
    }
    // run detection inside the wave (wavefront ballot primitive)
    const int prev = __shfl_up(c, 1, 64);
    const bool head = (lane == 0) || (prev != c);
    const unsigned long long heads = __ballot(head);
    const unsigned long long below = heads & (~0ull >> (63 - lane));  // heads at lanes <= lane
    const int head_lane = 63 - __clzll(below);
    const unsigned long long above = (head_lane == 63) ? 0ull : (heads >> (head_lane + 1));
    const int run_len = above ? (__ffsll((long long)above)) : (64 - head_lane);
    int base = 0;
    if (head && c >= 0) base = atomicAdd(&cell_cnt[c], run_len);  // particle_system.py:316
    base = __shfl(base, head_lane, 64);
    if (i < d.N) rank_off[i] = base + (lane - head_lane);
}

// ---- scan: in-place inclusive i32 prefix sum over the (padded) cell array ----
__device__ __forceinline__ int block_exclusive_offset(int thread_total, int* s_wave, int& block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = sph_wave_inclusive_scan(thread_total, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) {
        const int s = s_wave[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    block_total = tot;
    return wave_off + incl - thread_total;
}

// One launch (round 5; rounds 1-4: k_scan_reduce + k_scan_apply, the tile loaded twice and a kernel boundary -- ~3.3 us on
// this part -- between them).  Every tile publishes its total as ONE 64-bit word (launch epoch << 32 | total) with a
// device-scope atomic, then adds up the totals of the tiles before it -- waiting for the epoch to appear in each -- and
// writes its scanned values.  Only the TOTALS cross workgroups inside the kernel, and they travel in the atomic word
// itself: no fence, no cache write-back (an agent-scope release costs what a kernel boundary costs here, DESIGN_HISTORY
// r04); the scanned cells are ordinary stores that the next kernel reads.  No chained look-back either: a tile waits for
// totals, never for another tile's prefix, so the critical path is one load + reduce + atomic, whatever the tile count.
// Deadlock-free without co-residency: a tile waits only for tiles with SMALLER block ids, which the dispatcher started
// before it and which wait for nothing but still smaller ones.  The wait is bounded all the same: a tile that does not see
// a predecessor's word within SCAN_SPIN_LIMIT polls raises *err instead of hanging the GPU.
#define SCAN_SPIN_LIMIT (1 << 22)
__global__ __launch_bounds__(TPB) void k_scan_fused(int4* __restrict__ data, unsigned long long* __restrict__ status,
                                                    unsigned epoch, int* __restrict__ err) {
    __shared__ int s_wave[TPB / 64];
    const int base = (blockIdx.x * TPB + threadIdx.x) * (SCAN_IPT / 4);
    int4 v[SCAN_IPT / 4];
    int t = 0;
#pragma unroll
    for (int k = 0; k < SCAN_IPT / 4; ++k) {
        v[k] = data[base + k];
        t += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    int tot;
    const int ex = block_exclusive_offset(t, s_wave, tot);
    if (threadIdx.x == 0)
        __hip_atomic_store(&status[blockIdx.x], ((unsigned long long)epoch << 32) | (unsigned)tot, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    int before = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += TPB) {
        unsigned long long w;
        int spins = 0;
        do {
            w = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((unsigned)(w >> 32) != epoch && ++spins < SCAN_SPIN_LIMIT);
        if ((unsigned)(w >> 32) != epoch) { *err = 1; w = 0ull; }
        before += (int)(unsigned)w;
    }
    __syncthreads();  // s_wave is reused
    int before_tot;
    (void)block_exclusive_offset(before, s_wave, before_tot);
    int run = before_tot + ex;
#pragma unroll
    for (int k = 0; k < SCAN_IPT / 4; ++k) {
        run += v[k].x; v[k].x = run;
        run += v[k].y; v[k].y = run;
        run += v[k].z; v[k].z = run;
        run += v[k].w; v[k].w = run;
        data[base + k] = v[k];
    }
}

__global__ __launch_bounds__(TPB) void k_unstable_place(DevView d, const int* __restrict__ rank_off,
                                                        int* __restrict__ idx_unstable, int* __restrict__ dyn_count,
                                                        CellIdx16 off_ix, int* __restrict__ off_out, int off_stamp) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i == 0) *dyn_count = 0;  // counted by the scatter that follows (saves a memset launch)
    // slab ranks: the record counts below a few x layers (scanned cells) go straight into mapped host memory for the driver,
    // followed -- behind a system-scope fence -- by the sort's stamp: the host spins on the stamp instead of waiting for an event
    // (an event record is a packet of its own on the stream: 4 us between this kernel and the scatter, r06d trace)
    if (off_out != nullptr && i == 0) {
        int v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = off_ix.v[k] >= 0 ? d.cell_end[off_ix.v[k]] : 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) __hip_atomic_store(&off_out[k], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __atomic_thread_fence(__ATOMIC_RELEASE);   // system scope by default in HIP: the values are visible before the stamp
        __hip_atomic_store(&off_out[17], off_stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (i >= d.N) return;
    const int c = d.key[i];
    const int b = c > 0 ? d.cell_end[c - 1] : 0;  // particle_system.py:327-329 base_offset
    idx_unstable[b + rank_off[i]] = i;
}

// + the step's brick list (sph_bricks.h) in the first bl.nblocks workgroups of the scatter: the list needs the scanned
// cell array and nothing else; as a launch of its own its ~13 us chain of dependent phases (304 workgroups) sat on the
// critical path, beside the 6.5 us place kernel it still stuck out by half -- beside the 33 us scatter it disappears.
template <bool SORT_ACC>
__global__ __launch_bounds__(TPB) void k_stable_scatter(DevView d, const int* __restrict__ idx_unstable,
                                                        float4* __restrict__ xm_out, float4* __restrict__ vf_out,
                                                        float4* __restrict__ aux_out, int* __restrict__ key_out,
                                                        float4* __restrict__ acc_out, int* __restrict__ dyn_list,
                                                        int* __restrict__ dyn_count, int4* __restrict__ zero_dst,
                                                        int zero_n4, BrickListArgs bl) {
    extern __shared__ int sm_bl[];
    if ((int)blockIdx.x < bl.nblocks) {
        sph_brick_list_block<SPH_BRICK_BX, SPH_BRICK_BY, SPH_BRICK_BZ>(bl.d, bl.nbx, bl.nby, bl.list, bl.count, bl.list_cap, bl.tmax,
                                                                        bl.smax, bl.fixed_bz, (int)blockIdx.x, sm_bl);
        return;
    }
    const int s = ((int)blockIdx.x - bl.nblocks) * TPB + threadIdx.x;
    // the OTHER cell array (the previous step's, dead by now) is zeroed here for the next histogram: 2 MB of stores in
    // a 180 MB kernel instead of a memset launch of its own (~10 us per step)
    for (int z = s; z < zero_n4; z += ((int)gridDim.x - bl.nblocks) * TPB) zero_dst[z] = make_int4(0, 0, 0, 0);
    const bool live = s < d.N;
    const int lane = threadIdx.x & 63;
    const int i = live ? idx_unstable[s] : 0;
    const int c = live ? d.key[i] : 0;
    const int b = c > 0 ? d.cell_end[c - 1] : 0;
    const int e = live ? d.cell_end[c] : 0;
    const int my = live ? (d.sort_by_pid ? __float_as_int(d.aux[i].w) : i) : 0;  // the key a cell is ordered by: previous index (the
    // reference's serial order) or persistent id (the canonical order two ranks agree on, slab DFSPH)
    int rank = s - b;  // virtual cell G (dropped slab strays): any order will do, and the cell can be huge
    // Rank among the cell's members = how many of them have a smaller key.  A lane walks its own cell (<= ~30 members,
    // L1-resident) -- but a crowded cell (a collapsed scene, a pile-up in a corner) would keep ONE lane busy for
    // hundreds of reads while 63 idle: cells with more than 64 members are ranked by the whole wave, one target at a
    // time, every lane taking a 64th of the members.
    const bool ranked = live && c != d.G;
    const bool crowded = ranked && e - b > 64;
    if (ranked && !crowded) {
        rank = 0;
        if (d.sort_by_pid) {
            for (int t = b; t < e; ++t) rank += (__float_as_int(d.aux[idx_unstable[t]].w) < my) ? 1 : 0;
        } else {
            for (int t = b; t < e; ++t) rank += (idx_unstable[t] < my) ? 1 : 0;
        }
    }
    unsigned long long todo = __ballot(crowded);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const int bb = __shfl(b, src, 64), ee = __shfl(e, src, 64), kk = __shfl(my, src, 64);
        int cnt = 0;
        for (int t = bb + lane; t < ee; t += 64) {
            const int v = idx_unstable[t];
            cnt += ((d.sort_by_pid ? __float_as_int(d.aux[v].w) : v) < kk) ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        if (lane == src) rank = cnt;
    }
    if (!live) return;
    const int dst = b + rank;  // == grid_ids_new[i] of a serial run (particle_system.py:330)
    const float4 xm = d.xm[i];
    const float4 vf = d.vf[i];
    const float4 aux = d.aux[i];
    xm_out[dst] = xm;
    vf_out[dst] = vf;
    aux_out[dst] = aux;
    key_out[dst] = c;
    if (SORT_ACC) acc_out[dst] = d.acc[i];
    if (c != d.G && sph_is_dynamic_rigid(__float_as_int(vf.w))) dyn_list[atomicAdd(dyn_count, 1)] = dst;
}

// ---------------------------------------------------------------------------
int sphk_hash_histogram(SphContext* c) {
    const int Gpad = c->scan_blocks * SCAN_TILE;
    // flip to the other cell array; the last scatter normally left it zeroed
    int* nxt = c->cell_buf[c->cell_cur ^ 1];
    if (!c->next_cells_zero) SPH_HIP(c, hipMemsetAsync(nxt, 0, sizeof(int) * (size_t)Gpad, c->stream));
    c->cell_cur ^= 1;
    c->cell_end = nxt;
    c->next_cells_zero = false;
    c->brick_count_zero = c->N > 0;
    if (c->N > 0) {
        DevView d = sph_view(c);
        hipLaunchKernelGGL(k_hash_histogram, dim3((c->N + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d, c->cell_end,
                           c->rank_off, c->brick_count);
        SPH_LAUNCH_CHECK(c);
    }
    return 0;
}

int sphk_scan(SphContext* c) {
    const int nb = c->scan_blocks;
    c->scan_epoch += 1u;
    if (c->scan_epoch == 0u) {  // (wrapped after 2^32 scans: the words of an old epoch 1 could be mistaken for new ones)
        SPH_HIP(c, hipMemsetAsync(c->scan_status, 0, (size_t)(nb + 1) * 8, c->stream));
        c->scan_epoch = 1u;
    }
    hipLaunchKernelGGL(k_scan_fused, dim3(nb), dim3(TPB), 0, c->stream, (int4*)c->cell_end, c->scan_status, c->scan_epoch,
                       c->scan_err);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_sort_scatter(SphContext* c, bool sort_acc) {
    if (c->N <= 0) return 0;
    DevView d = sph_view(c);
    const int nb = (c->N + TPB - 1) / TPB;
    const int o = c->cur ^ 1;
    int4* zero_dst = reinterpret_cast<int4*>(c->cell_buf[c->cell_cur ^ 1]);
    const int zero_n4 = c->scan_blocks * SCAN_TILE / 4;
    BrickListArgs bl;
    int rc = sphk_brick_list_prepare(c, &bl);
    if (rc) return rc;
    if (bl.nblocks > 0 && !c->brick_count_zero) {  // (a sort without its own hash pass: sph_counting_sort called twice)
        SPH_HIP(c, hipMemsetAsync(c->brick_count, 0, 2 * sizeof(int), c->stream));
    }
    CellIdx16 off_ix;
    int* off_out = nullptr;
    for (int k = 0; k < 16; ++k) off_ix.v[k] = -1;
    if (c->off_in_sort) {
        off_ix = c->off_ix;
        SPH_HIP(c, hipHostGetDevicePointer((void**)&off_out, c->h_pinned, 0));
    }
    if (c->off_in_sort) {
        c->off_stamp = c->off_stamp == 0x7fffffff ? 1 : c->off_stamp + 1;
        c->off_stamp_pending = true;
    }
    hipLaunchKernelGGL(k_unstable_place, dim3(nb), dim3(TPB), 0, c->stream, d, c->rank_off, c->idx_unstable, c->dyn_count, off_ix, off_out,
                       c->off_stamp);
    SPH_LAUNCH_CHECK(c);
    if (sort_acc)
        hipLaunchKernelGGL(k_stable_scatter<true>, dim3(nb + bl.nblocks), dim3(TPB), bl.lds_bytes, c->stream, d, c->idx_unstable, c->xm[o],
                           c->vf[o], c->aux[o], c->key[o], c->acc_tmp, c->dyn_list, c->dyn_count, zero_dst, zero_n4, bl);
    else
        hipLaunchKernelGGL(k_stable_scatter<false>, dim3(nb + bl.nblocks), dim3(TPB), bl.lds_bytes, c->stream, d, c->idx_unstable, c->xm[o],
                           c->vf[o], c->aux[o], c->key[o], c->acc_tmp, c->dyn_list, c->dyn_count, zero_dst, zero_n4, bl);
    SPH_LAUNCH_CHECK(c);
    c->next_cells_zero = true;
    c->cur = o;
    sph_invalidate_lists(c);
    c->brick_count_zero = false;
    c->bricks_valid = bl.nblocks > 0;  // (key recorded by sphk_brick_list_prepare)
    c->gcnt_written = false;
    c->aux_stale = false;  // (eos2 is in the old order: whoever needed density / pressure called sph_ensure_aux before)
    if (sort_acc) {
        float4* t = c->acc;
        c->acc = c->acc_tmp;
        c->acc_tmp = t;
    }
    return 0;
}
