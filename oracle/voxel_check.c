// voxel_check.c -- TEST INFRASTRUCTURE (oracle/): an independent check of the mesh -> particle path of
// /root/reference/particle_system.py:421-447 (load_rigid_body: trimesh.load, apply_scale, rotation about the vertex
// mean, repair.fill_holes, mesh.voxelized(pitch).fill().points), whose product-side restatement is
// sph_taichi_amd/voxelizer.py.  trimesh is a third-party dependency that is absent from this image (the reference's
// requirements.txt names it unpinned), so what is restated here is its published algorithm
// (trimesh.voxel.creation.voxelize_subdivide + VoxelGrid.fill(method="holes")), a SECOND time and differently built, plus
// a purely geometric classifier that does not know that algorithm at all:
//
//   vc_sample_shell : the voxels hit by the vertices of the mesh subdivided until every edge is <= pitch / 2, voxel =
//                     rint(vertex / pitch) (half-even, numpy.round) -- written as a per-face depth-first recursion
//                     (voxelizer.py subdivides level by level over arrays of all faces);
//   vc_fill         : holes filled = everything the background cannot reach from outside through FACE neighbours
//                     (scipy.ndimage.binary_fill_holes' default structure), as a breadth-first flood (voxelizer.py calls scipy);
//   vc_box_shell    : which voxel cubes [c - d/2, c + d/2]^3 does the SURFACE touch at all -- exact triangle / box
//                     separating-axis test (Akenine-Moller 2001), no sampling;
//   vc_parity       : which perturbed voxel centres lie INSIDE the closed surface -- crossings of an axis-parallel ray,
//                     one call per axis (the caller takes the majority of the three).
//
// tests/test_voxelizer_crosscheck.py holds voxelizer.py to the first pair (identical voxel SETS) and explains the result
// with the second pair (every sampled voxel's cube touches the surface; every interior centre is filled; what else is
// filled is an enclosed exterior pocket, counted).  Nothing here is linked into or called by the product.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { long o[3]; int n[3]; double pitch; } Grid;   // voxel (i, j, k) has the lattice index o + (i, j, k), centre index * pitch

static inline size_t gidx(const Grid* g, int i, int j, int k) { return ((size_t)i * g->n[1] + j) * g->n[2] + k; }

static void mark(const Grid* g, unsigned char* out, const double v[3]) {
    long q[3];
    for (int a = 0; a < 3; ++a) {
        q[a] = (long)nearbyint(v[a] / g->pitch) - g->o[a];   // numpy.round: round-half-even on the f64 quotient
        if (q[a] < 0 || q[a] >= g->n[a]) return;              // (the caller's grid covers the bounding box + a margin)
    }
    out[gidx(g, (int)q[0], (int)q[1], (int)q[2])] = 1;
}

static double dist(const double a[3], const double b[3]) {
    const double x = b[0] - a[0], y = b[1] - a[1], z = b[2] - a[2];
    return sqrt(x * x + y * y + z * z);
}

// trimesh.remesh.subdivide_to_size, one face at a time: a face with an edge longer than max_edge becomes four (corner,
// corner, corner, middle) over its edge midpoints (a + b) / 2, and so on; returns 1 if max_iter levels did not suffice
static int subdivide_mark(const Grid* g, unsigned char* out, const double a[3], const double b[3], const double c[3],
                          double max_edge, int depth, int max_iter) {
    if (!(dist(a, b) > max_edge || dist(b, c) > max_edge || dist(c, a) > max_edge)) {
        mark(g, out, a); mark(g, out, b); mark(g, out, c);
        return 0;
    }
    if (depth >= max_iter) return 1;
    double ab[3], bc[3], ca[3];
    for (int k = 0; k < 3; ++k) { ab[k] = (a[k] + b[k]) * 0.5; bc[k] = (b[k] + c[k]) * 0.5; ca[k] = (c[k] + a[k]) * 0.5; }
    int bad = 0;
    bad |= subdivide_mark(g, out, a, ab, ca, max_edge, depth + 1, max_iter);
    bad |= subdivide_mark(g, out, ab, b, bc, max_edge, depth + 1, max_iter);
    bad |= subdivide_mark(g, out, ca, bc, c, max_edge, depth + 1, max_iter);
    bad |= subdivide_mark(g, out, ab, bc, ca, max_edge, depth + 1, max_iter);
    return bad;
}

int vc_sample_shell(const double* V, const int* F, int nF, const long o[3], const int n[3], double pitch, double edge_factor,
                    int max_iter, unsigned char* out) {
    Grid g = {{o[0], o[1], o[2]}, {n[0], n[1], n[2]}, pitch};
    memset(out, 0, (size_t)n[0] * n[1] * n[2]);
    int bad = 0;
    for (int f = 0; f < nF; ++f)
        bad |= subdivide_mark(&g, out, V + 3 * (size_t)F[3 * f], V + 3 * (size_t)F[3 * f + 1], V + 3 * (size_t)F[3 * f + 2],
                              pitch / edge_factor, 0, max_iter);
    return bad;
}

// out = in with every background region that does not reach the grid's border through face neighbours filled
void vc_fill(const unsigned char* in, const int n[3], unsigned char* out) {
    const size_t total = (size_t)n[0] * n[1] * n[2];
    unsigned char* seen = (unsigned char*)calloc(total, 1);
    size_t* queue = (size_t*)malloc(total * sizeof(size_t));
    size_t head = 0, tail = 0;
#define PUSH(i_, j_, k_) do { const size_t q_ = ((size_t)(i_) * n[1] + (j_)) * n[2] + (k_); \
                              if (!in[q_] && !seen[q_]) { seen[q_] = 1; queue[tail++] = q_; } } while (0)
    for (int i = 0; i < n[0]; ++i)
        for (int j = 0; j < n[1]; ++j)
            for (int k = 0; k < n[2]; ++k)
                if (i == 0 || j == 0 || k == 0 || i == n[0] - 1 || j == n[1] - 1 || k == n[2] - 1) PUSH(i, j, k);
    while (head < tail) {
        const size_t q = queue[head++];
        const int k = (int)(q % n[2]), j = (int)((q / n[2]) % n[1]), i = (int)(q / ((size_t)n[2] * n[1]));
        if (i > 0) PUSH(i - 1, j, k);
        if (i < n[0] - 1) PUSH(i + 1, j, k);
        if (j > 0) PUSH(i, j - 1, k);
        if (j < n[1] - 1) PUSH(i, j + 1, k);
        if (k > 0) PUSH(i, j, k - 1);
        if (k < n[2] - 1) PUSH(i, j, k + 1);
    }
#undef PUSH
    for (size_t q = 0; q < total; ++q) out[q] = (unsigned char)(in[q] || !seen[q]);
    free(seen);
    free(queue);
}

// ---- exact triangle / axis-aligned box overlap (separating axes: 3 box normals, the triangle normal, 9 edge cross products) ----
static int axis_separates(const double v[3][3], const double ax[3], double h) {
    double lo = 1e300, hi = -1e300;
    for (int k = 0; k < 3; ++k) {
        const double p = v[k][0] * ax[0] + v[k][1] * ax[1] + v[k][2] * ax[2];
        if (p < lo) lo = p;
        if (p > hi) hi = p;
    }
    const double r = h * (fabs(ax[0]) + fabs(ax[1]) + fabs(ax[2]));
    return lo > r || hi < -r;
}

static int tri_box_overlap(const double c[3], double h, const double* a, const double* b, const double* cc) {
    double v[3][3];
    for (int k = 0; k < 3; ++k) { v[0][k] = a[k] - c[k]; v[1][k] = b[k] - c[k]; v[2][k] = cc[k] - c[k]; }
    const double e[3][3] = {{v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]},
                            {v[2][0] - v[1][0], v[2][1] - v[1][1], v[2][2] - v[1][2]},
                            {v[0][0] - v[2][0], v[0][1] - v[2][1], v[0][2] - v[2][2]}};
    for (int k = 0; k < 3; ++k) {
        double ax[3] = {0.0, 0.0, 0.0};
        ax[k] = 1.0;
        if (axis_separates(v, ax, h)) return 0;
    }
    const double nrm[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2], e[0][0] * e[1][1] - e[0][1] * e[1][0]};
    if (axis_separates(v, nrm, h)) return 0;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            double u[3] = {0.0, 0.0, 0.0}, ax[3];
            u[k] = 1.0;
            ax[0] = u[1] * e[i][2] - u[2] * e[i][1]; ax[1] = u[2] * e[i][0] - u[0] * e[i][2]; ax[2] = u[0] * e[i][1] - u[1] * e[i][0];
            if (ax[0] == 0.0 && ax[1] == 0.0 && ax[2] == 0.0) continue;
            if (axis_separates(v, ax, h)) return 0;
        }
    return 1;
}

// out[voxel] = 1 iff some triangle touches the closed cube of half width `half` around the voxel's centre
void vc_box_shell(const double* V, const int* F, int nF, const long o[3], const int n[3], double pitch, double half, unsigned char* out) {
    memset(out, 0, (size_t)n[0] * n[1] * n[2]);
    Grid g = {{o[0], o[1], o[2]}, {n[0], n[1], n[2]}, pitch};
    for (int f = 0; f < nF; ++f) {
        const double* a = V + 3 * (size_t)F[3 * f];
        const double* b = V + 3 * (size_t)F[3 * f + 1];
        const double* c = V + 3 * (size_t)F[3 * f + 2];
        int lo[3], hi[3];
        for (int k = 0; k < 3; ++k) {
            const double mn = fmin(a[k], fmin(b[k], c[k])), mx = fmax(a[k], fmax(b[k], c[k]));
            lo[k] = (int)(floor((mn - half) / pitch) - g.o[k]) - 1;
            hi[k] = (int)(ceil((mx + half) / pitch) - g.o[k]) + 1;
            if (lo[k] < 0) lo[k] = 0;
            if (hi[k] > n[k] - 1) hi[k] = n[k] - 1;
        }
        for (int i = lo[0]; i <= hi[0]; ++i)
            for (int j = lo[1]; j <= hi[1]; ++j)
                for (int k = lo[2]; k <= hi[2]; ++k) {
                    const size_t q = gidx(&g, i, j, k);
                    if (out[q]) continue;
                    const double ctr[3] = {(double)(g.o[0] + i) * pitch, (double)(g.o[1] + j) * pitch, (double)(g.o[2] + k) * pitch};
                    if (tri_box_overlap(ctr, half, a, b, c)) out[q] = 1;
                }
    }
}

// ---- inside / outside by ray parity along one axis ----
// The ray of voxel (i, j, k) starts at its centre + eps and runs along +axis.  A voxel whose cube the surface does not touch
// is on one side of the surface as a whole, so the small generic offset eps (which keeps the rays off vertices and edges)
// cannot change its answer.  Edge functions are evaluated with the edge's endpoints in a canonical (vertex-index) order, so
// the two triangles that share an edge agree on which side of it a ray passes: a ray is never counted by both or neither.
static double edge_fn(const double* V, int ia, int ib, int u, int w, double pu, double pw) {
    int flip = 0;
    if (ia > ib) { const int t = ia; ia = ib; ib = t; flip = 1; }
    const double* a = V + 3 * (size_t)ia;
    const double* b = V + 3 * (size_t)ib;
    const double s = (b[u] - a[u]) * (pw - a[w]) - (b[w] - a[w]) * (pu - a[u]);
    return flip ? -s : s;
}

void vc_parity(const double* V, const int* F, int nF, const long o[3], const int n[3], double pitch, int axis, const double eps[3],
               unsigned char* inside) {
    const int u = (axis + 1) % 3, w = (axis + 2) % 3;
    const int nu = n[u], nw = n[w], na = n[axis];
    // cnt[(ju * nw + jw) * (na + 1) + m] = crossings of line (ju, jw) that have exactly m lattice centres below them
    unsigned* cnt = (unsigned*)calloc((size_t)nu * nw * (na + 1), sizeof(unsigned));
    for (int f = 0; f < nF; ++f) {
        const int ia = F[3 * f], ib = F[3 * f + 1], ic = F[3 * f + 2];
        const double* a = V + 3 * (size_t)ia;
        const double* b = V + 3 * (size_t)ib;
        const double* c = V + 3 * (size_t)ic;
        const double umin = fmin(a[u], fmin(b[u], c[u])), umax = fmax(a[u], fmax(b[u], c[u]));
        const double wmin = fmin(a[w], fmin(b[w], c[w])), wmax = fmax(a[w], fmax(b[w], c[w]));
        int ju0 = (int)(ceil((umin - eps[u]) / pitch) - o[u]) - 1, ju1 = (int)(floor((umax - eps[u]) / pitch) - o[u]) + 1;
        int jw0 = (int)(ceil((wmin - eps[w]) / pitch) - o[w]) - 1, jw1 = (int)(floor((wmax - eps[w]) / pitch) - o[w]) + 1;
        if (ju0 < 0) ju0 = 0;
        if (jw0 < 0) jw0 = 0;
        if (ju1 > nu - 1) ju1 = nu - 1;
        if (jw1 > nw - 1) jw1 = nw - 1;
        for (int ju = ju0; ju <= ju1; ++ju)
            for (int jw = jw0; jw <= jw1; ++jw) {
                const double pu = (double)(o[u] + ju) * pitch + eps[u], pw = (double)(o[w] + jw) * pitch + eps[w];
                const double s0 = edge_fn(V, ia, ib, u, w, pu, pw), s1 = edge_fn(V, ib, ic, u, w, pu, pw), s2 = edge_fn(V, ic, ia, u, w, pu, pw);
                const int pos = (s0 >= 0.0) && (s1 >= 0.0) && (s2 >= 0.0), neg = (s0 < 0.0) && (s1 < 0.0) && (s2 < 0.0);
                if (!pos && !neg) continue;
                const double area = s0 + s1 + s2;     // twice the projected area, signed (the three edge functions add up to it)
                if (area == 0.0) continue;
                // barycentric interpolation of the crossing's coordinate along the axis: weight of vertex c is s0 / area, ...
                const double t = (s1 * a[axis] + s2 * b[axis] + s0 * c[axis]) / area;
                // lattice centres strictly below the crossing: index m with (o + m) * pitch + eps < t
                double mf = ceil((t - eps[axis]) / pitch - (double)o[axis]);
                if (mf < 0.0) mf = 0.0;
                if (mf > (double)na) mf = (double)na;
                cnt[((size_t)ju * nw + jw) * (na + 1) + (int)mf] += 1u;
            }
    }
    for (int ju = 0; ju < nu; ++ju)
        for (int jw = 0; jw < nw; ++jw) {
            const unsigned* row = cnt + ((size_t)ju * nw + jw) * (na + 1);
            unsigned above = 0;   // crossings above centre m = those with more than m centres below them
            for (int m = na - 1; m >= 0; --m) {
                above += row[m + 1];
                int ijk[3];
                ijk[axis] = m; ijk[u] = ju; ijk[w] = jw;
                inside[((size_t)ijk[0] * n[1] + ijk[1]) * n[2] + ijk[2]] = (unsigned char)(above & 1u);
            }
        }
    free(cnt);
}
