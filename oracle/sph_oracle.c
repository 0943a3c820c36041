/*
 * sph_oracle.c -- CPU restatement of erizmr/SPH_Taichi's WCSPH step (and, further down, of its DFSPH step).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (sph_taichi_amd/) may
 * import, link or call this file.  It is used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker /
 * reported CPU baseline -- never as the thing shipped.
 *
 * PARITY PINNING: the reference is Python that only runs under Taichi, and
 * Taichi cannot be installed in this image.  The restatement is pinned in two
 * ways (see oracle/README.md and DESIGN.md):
 *   1. tests/golden/ref_*.npz are produced by EXECUTING THE REFERENCE'S OWN
 *      SOURCE FILES (particle_system.py, sph_base.py, WCSPH.py, DFSPH.py, unmodified,
 *      from /root/reference) under a serial pure-Python stand-in for the
 *      `taichi` module (oracle/taichi_shim/, generator oracle/gen_golden.py).
 *      That pins every formula and the traversal order; it cannot pin Taichi's
 *      own compiler rounding (fast-math, pow lowering) nor ti.polar_decompose.
 *   2. the known-answer table of SURVEY.md section 4 (tests/test_oracle_kat.py)
 *      and an independent O(N^2) NumPy brute force (oracle/brute.py).
 *
 * Conventions: f32 everywhere, i32 indices, vectors are AoS [N][3] like a
 * Taichi Vector.field, loops visit particles / neighbour cells in exactly the
 * reference order, atomics are executed serially (=> the stable counting-sort
 * order of a single-threaded run).  Compile with -ffp-contract=off and without
 * -ffast-math so each f32 op rounds once.
 *
 * Every function cites the reference lines it follows (paths are relative to
 * /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct OracleState {
    /* sizes */
    int32_t N;               /* particle_max_num          particle_system.py:83 */
    int32_t grid_num[3];     /* particle_system.py:44 */
    int32_t G;               /* number of cells           particle_system.py:96 */
    int32_t n_objects;       /* length of rigid_rest_cm   particle_system.py:93 */
    /* scalars (all f32 when they enter a kernel) */
    float grid_size;         /* = support_radius          particle_system.py:43 */
    float support_radius;    /* 4 r                       particle_system.py:37 */
    float particle_diameter; /* 2 r                       particle_system.py:36 */
    float m_V0;              /* 0.8 d^3                   particle_system.py:38 */
    float density_0;         /* sph_base.py:18 */
    float stiffness;         /* WCSPH.py:13 */
    float exponent;          /* WCSPH.py:10 */
    float viscosity;         /* 0.01  sph_base.py:15 */
    float surface_tension;   /* 0.01  WCSPH.py:15 */
    float dt;                /* WCSPH.py:16 */
    float g[3];              /* sph_base.py:13 */
    float domain_size[3];    /* particle_system.py:22 */
    float padding;           /* particle_system.py:46 */
    float wall_hi[3];        /* domain_size - padding folded in f64 then cast, as Taichi folds Python-scope constants */
    /* f32 kernel constants, folded in f64 on the Python side like Taichi does */
    float k_w;               /* 8/pi/h^3        sph_base.py:33-35 */
    float k_dw;              /* 6*8/pi/h^3      sph_base.py:57 */
    float visc_d_nu;         /* 2*(dim+2)*viscosity   WCSPH.py:104,112 */
    float visc_eps;          /* 0.01*h^2              WCSPH.py:113 */
    int32_t omp_threads;     /* 1 = deterministic serial order */
    int32_t rigid_sums_f64;  /* test switch (not in the reference): accumulate the shape-matching sums in f64 */
    /* per-particle state   particle_system.py:101-113 */
    int32_t *object_id; float *x; float *x_0; float *v; float *acceleration;
    float *m_V; float *m; float *density; float *pressure;
    int32_t *material; int32_t *color; int32_t *is_dynamic;
    /* sort buffers          particle_system.py:120-131 */
    int32_t *object_id_buffer; float *x_buffer; float *x_0_buffer; float *v_buffer;
    float *acceleration_buffer; float *m_V_buffer; float *m_buffer; float *density_buffer;
    float *pressure_buffer; int32_t *material_buffer; int32_t *color_buffer;
    int32_t *is_dynamic_buffer;
    /* grid                  particle_system.py:96-97, 138-140 */
    int32_t *grid_ids; int32_t *grid_ids_buffer; int32_t *grid_ids_new;
    int32_t *grid_particles_num; int32_t *grid_particles_num_temp;
    float *rigid_rest_cm;    /* [n_objects][3]  particle_system.py:93 */
    /* extra (not in the reference): persistent particle id carried through the
     * sort so tests can compare per particle. */
    int32_t *pid; int32_t *pid_buffer;
    /* ---- DFSPH (simulationMethod 4) ---- */
    int32_t simulation_method;        /* 0 WCSPH, 4 DFSPH            particle_system.py:27, 214-221 */
    int32_t fluid_particle_num;       /* particle_system.py:57-62 */
    int32_t enable_divergence_solver; /* DFSPH.py:12 */
    int32_t m_max_iterations_v;       /* DFSPH.py:14 */
    int32_t m_max_iterations;         /* DFSPH.py:15 */
    float m_eps;                      /* DFSPH.py:17 */
    double max_error_V;               /* DFSPH.py:19 (Python-scope floats stay f64) */
    double max_error;                 /* DFSPH.py:20 */
    int32_t last_iterations_v;        /* what the reference prints at DFSPH.py:258 / :353 */
    int32_t last_iterations;
    double last_avg_err_v;
    double last_avg_err;
    float *dfsph_factor; float *density_adv;               /* particle_system.py:116-117 */
    float *dfsph_factor_buffer; float *density_adv_buffer; /* particle_system.py:134-135 */
    /* Test switch (not in the reference): deliberate ulp-level perturbations of the SAME formulas, to measure how far two
     * correct f32 evaluations drift apart on a given scene (tools/fastmath_ab.py: what a post-impact "error" is made of).
     * bit 0: neighbours are visited in the reverse order (cells and particles) => every sum is added up in another order;
     * bit 1: the Tait exponent, when an integer, by repeated multiplication instead of powf (what the HIP path does). */
    int32_t perturb;
} OracleState;

#define MATERIAL_SOLID 0 /* particle_system.py:30 */
#define MATERIAL_FLUID 1 /* particle_system.py:31 */

static void set_threads(const OracleState *s) {
#ifdef _OPENMP
    omp_set_num_threads(s->omp_threads > 0 ? s->omp_threads : 1);
#else
    (void)s;
#endif
}

/* particle_system.py:287-294  pos_to_index + flatten_grid_index.
 * (pos / grid_size).cast(int): f32 division, truncation toward zero.  The
 * reference does no bounds handling (out-of-domain positions are UB there);
 * we clamp the cell coordinate so a stray particle cannot corrupt memory. */
static inline int32_t cell_coord(float p, float grid_size, int32_t n) {
    int32_t c = (int32_t)(p / grid_size);
    if (c < 0) c = 0;
    if (c > n - 1) c = n - 1;
    return c;
}
static inline int32_t flatten(const OracleState *s, int32_t cx, int32_t cy, int32_t cz) {
    return cx * s->grid_num[1] * s->grid_num[2] + cy * s->grid_num[2] + cz;
}
static inline int32_t get_flatten_grid_index(const OracleState *s, const float *p) {
    return flatten(s, cell_coord(p[0], s->grid_size, s->grid_num[0]),
                   cell_coord(p[1], s->grid_size, s->grid_num[1]),
                   cell_coord(p[2], s->grid_size, s->grid_num[2]));
}

/* particle_system.py:301-308 */
static inline int is_static_rigid_body(const OracleState *s, int32_t p) {
    return s->material[p] == MATERIAL_SOLID && !s->is_dynamic[p];
}
static inline int is_dynamic_rigid_body(const OracleState *s, int32_t p) {
    return s->material[p] == MATERIAL_SOLID && s->is_dynamic[p];
}

/* particle_system.py:311-320  update_grid_id */
void oracle_update_grid_id(OracleState *s) {
    for (int32_t c = 0; c < s->G; ++c) s->grid_particles_num[c] = 0;
    for (int32_t i = 0; i < s->N; ++i) {
        int32_t gi = get_flatten_grid_index(s, &s->x[3 * i]);
        s->grid_ids[i] = gi;
        s->grid_particles_num[gi] += 1; /* ti.atomic_add */
    }
    for (int32_t c = 0; c < s->G; ++c) s->grid_particles_num_temp[c] = s->grid_particles_num[c];
}

/* particle_system.py:374  prefix_sum_executor.run(grid_particles_num): in-place
 * INCLUSIVE i32 scan (in-repo twin scan_single_buffer.py:108-146). */
void oracle_prefix_sum(OracleState *s) {
    int32_t acc = 0;
    for (int32_t c = 0; c < s->G; ++c) {
        acc += s->grid_particles_num[c];
        s->grid_particles_num[c] = acc;
    }
}

/* particle_system.py:322-369  counting_sort.  The rank loop is serial (reverse
 * index order, atomic_sub returns the OLD value) => stable order.  The scatter
 * and copy-back move the same 13 arrays the reference moves (+ pid). */
void oracle_counting_sort(OracleState *s) {
    const int32_t N = s->N;
    set_threads(s);
    for (int32_t i = 0; i < N; ++i) {
        int32_t I = N - 1 - i;
        int32_t base_offset = 0;
        if (s->grid_ids[I] - 1 >= 0) base_offset = s->grid_particles_num[s->grid_ids[I] - 1];
        int32_t old = s->grid_particles_num_temp[s->grid_ids[I]];
        s->grid_particles_num_temp[s->grid_ids[I]] = old - 1; /* ti.atomic_sub */
        s->grid_ids_new[I] = old - 1 + base_offset;
    }
#pragma omp parallel for schedule(static)
    for (int32_t I = 0; I < N; ++I) {
        int32_t n = s->grid_ids_new[I];
        s->grid_ids_buffer[n] = s->grid_ids[I];
        s->object_id_buffer[n] = s->object_id[I];
        memcpy(&s->x_0_buffer[3 * n], &s->x_0[3 * I], 12);
        memcpy(&s->x_buffer[3 * n], &s->x[3 * I], 12);
        memcpy(&s->v_buffer[3 * n], &s->v[3 * I], 12);
        memcpy(&s->acceleration_buffer[3 * n], &s->acceleration[3 * I], 12);
        s->m_V_buffer[n] = s->m_V[I];
        s->m_buffer[n] = s->m[I];
        s->density_buffer[n] = s->density[I];
        s->pressure_buffer[n] = s->pressure[I];
        s->material_buffer[n] = s->material[I];
        memcpy(&s->color_buffer[3 * n], &s->color[3 * I], 12);
        s->is_dynamic_buffer[n] = s->is_dynamic[I];
        s->pid_buffer[n] = s->pid[I];
        if (s->simulation_method == 4) { /* particle_system.py:348-350 */
            s->dfsph_factor_buffer[n] = s->dfsph_factor[I];
            s->density_adv_buffer[n] = s->density_adv[I];
        }
    }
#pragma omp parallel for schedule(static)
    for (int32_t I = 0; I < N; ++I) {
        s->grid_ids[I] = s->grid_ids_buffer[I];
        s->object_id[I] = s->object_id_buffer[I];
        memcpy(&s->x_0[3 * I], &s->x_0_buffer[3 * I], 12);
        memcpy(&s->x[3 * I], &s->x_buffer[3 * I], 12);
        memcpy(&s->v[3 * I], &s->v_buffer[3 * I], 12);
        memcpy(&s->acceleration[3 * I], &s->acceleration_buffer[3 * I], 12);
        s->m_V[I] = s->m_V_buffer[I];
        s->m[I] = s->m_buffer[I];
        s->density[I] = s->density_buffer[I];
        s->pressure[I] = s->pressure_buffer[I];
        s->material[I] = s->material_buffer[I];
        memcpy(&s->color[3 * I], &s->color_buffer[3 * I], 12);
        s->is_dynamic[I] = s->is_dynamic_buffer[I];
        s->pid[I] = s->pid_buffer[I];
        if (s->simulation_method == 4) { /* particle_system.py:367-369 */
            s->dfsph_factor[I] = s->dfsph_factor_buffer[I];
            s->density_adv[I] = s->density_adv_buffer[I];
        }
    }
}

/* particle_system.py:372-375 */
void oracle_initialize_particle_system(OracleState *s) {
    oracle_update_grid_id(s);
    oracle_prefix_sum(s);
    oracle_counting_sort(s);
}

/* sph_base.py:23-44  cubic_kernel */
static inline float cubic_kernel(const OracleState *s, float r_norm) {
    float res = 0.0f;
    const float h = s->support_radius;
    const float k = s->k_w;
    const float q = r_norm / h;
    if (q <= 1.0f) {
        if (q <= 0.5f) {
            const float q2 = q * q;
            const float q3 = q2 * q;
            res = k * (6.0f * q3 - 6.0f * q2 + 1.0f);
        } else {
            res = k * 2.0f * powf(1.0f - q, 3.0f);
        }
    }
    return res;
}

/* sph_base.py:46-68  cubic_kernel_derivative */
static inline void cubic_kernel_derivative(const OracleState *s, const float r[3], float res[3]) {
    const float h = s->support_radius;
    const float k = s->k_dw;
    const float r_norm = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    const float q = r_norm / h;
    res[0] = res[1] = res[2] = 0.0f;
    if (r_norm > 1e-5f && q <= 1.0f) {
        const float inv = r_norm * h;
        const float gq[3] = {r[0] / inv, r[1] / inv, r[2] / inv};
        float c;
        if (q <= 0.5f) {
            c = k * q * (3.0f * q - 2.0f);
        } else {
            const float factor = 1.0f - q;
            c = k * (-factor * factor);
        }
        res[0] = c * gq[0]; res[1] = c * gq[1]; res[2] = c * gq[2];
    }
}

static inline float norm3(const float a[3]) { return sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

/* particle_system.py:378-385  for_all_neighbors.  Offsets x-outer .. z-inner,
 * ascending p_j inside a cell.  Cell range = [P[max(0,c-1)], P[c]) -- note
 * this makes cell 0's own range empty, exactly as in the reference.  The
 * reference computes the flat index of out-of-grid neighbour cells without a
 * bounds check (UB / aliases a far cell whose particles fail the distance
 * test); we skip such cells.
 * NEIGHBOR_LOOP(p_i, p_j) { body } */
#define NEIGHBOR_LOOP_BEGIN(s, p_i, p_j)                                                   \
    {                                                                                      \
        const float *xi_ = &(s)->x[3 * (p_i)];                                             \
        const int32_t ccx_ = cell_coord(xi_[0], (s)->grid_size, (s)->grid_num[0]);          \
        const int32_t ccy_ = cell_coord(xi_[1], (s)->grid_size, (s)->grid_num[1]);          \
        const int32_t ccz_ = cell_coord(xi_[2], (s)->grid_size, (s)->grid_num[2]);          \
        const int32_t sg_ = ((s)->perturb & 1) ? -1 : 1; /* (test switch: reversed traversal) */ \
        for (int32_t ox_ = -1; ox_ <= 1; ++ox_)                                            \
            for (int32_t oy_ = -1; oy_ <= 1; ++oy_)                                        \
                for (int32_t oz_ = -1; oz_ <= 1; ++oz_) {                                  \
                    const int32_t nx_ = ccx_ + sg_ * ox_, ny_ = ccy_ + sg_ * oy_, nz_ = ccz_ + sg_ * oz_; \
                    if (nx_ < 0 || ny_ < 0 || nz_ < 0 || nx_ >= (s)->grid_num[0] ||        \
                        ny_ >= (s)->grid_num[1] || nz_ >= (s)->grid_num[2])                \
                        continue;                                                          \
                    const int32_t gi_ = flatten((s), nx_, ny_, nz_);                       \
                    const int32_t beg_ = (s)->grid_particles_num[gi_ - 1 > 0 ? gi_ - 1 : 0]; \
                    const int32_t end_ = (s)->grid_particles_num[gi_];                     \
                    for (int32_t t_ = beg_; t_ < end_; ++t_) {                             \
                        const int32_t p_j = sg_ > 0 ? t_ : end_ - 1 - (t_ - beg_);         \
                        const float *xj_ = &(s)->x[3 * p_j];                               \
                        const float rr_[3] = {xi_[0] - xj_[0], xi_[1] - xj_[1], xi_[2] - xj_[2]}; \
                        if ((p_i) != p_j && norm3(rr_) < (s)->support_radius) {
#define NEIGHBOR_LOOP_END \
    }                     \
    }                     \
    }                     \
    }

/* sph_base.py:100-103  compute_boundary_volume_task;
 * sph_base.py:91-98 (static, init only) / :106-113 (dynamic, every step) */
static void boundary_volume(OracleState *s, int dynamic) {
    set_threads(s);
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t p_i = 0; p_i < s->N; ++p_i) {
        if (dynamic ? !is_dynamic_rigid_body(s, p_i) : !is_static_rigid_body(s, p_i)) continue;
        float delta = cubic_kernel(s, 0.0f);
        NEIGHBOR_LOOP_BEGIN(s, p_i, p_j)
        if (s->material[p_j] == MATERIAL_SOLID) delta += cubic_kernel(s, norm3(rr_));
        NEIGHBOR_LOOP_END
        s->m_V[p_i] = 1.0f / delta * 3.0f;
    }
}
void oracle_compute_static_boundary_volume(OracleState *s) { boundary_volume(s, 0); }
void oracle_compute_moving_boundary_volume(OracleState *s) { boundary_volume(s, 1); }

/* WCSPH.py:19-43  compute_densities (+ task): both materials add m_V_j * W. */
void oracle_compute_densities(OracleState *s) {
    set_threads(s);
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t p_i = 0; p_i < s->N; ++p_i) {
        if (s->material[p_i] != MATERIAL_FLUID) continue;
        float rho = s->m_V[p_i] * cubic_kernel(s, 0.0f);
        float den = 0.0f;
        NEIGHBOR_LOOP_BEGIN(s, p_i, p_j)
        if (s->material[p_j] == MATERIAL_FLUID || s->material[p_j] == MATERIAL_SOLID)
            den += s->m_V[p_j] * cubic_kernel(s, norm3(rr_));
        NEIGHBOR_LOOP_END
        rho += den;
        rho *= s->density_0;
        s->density[p_i] = rho;
    }
}

static inline void atomic_add3(float *dst, const float v[3]) {
    for (int d = 0; d < 3; ++d) {
#pragma omp atomic
        dst[d] += v[d];
    }
}

/* WCSPH.py:88-140  compute_non_pressure_forces (+ task) */
void oracle_compute_non_pressure_forces(OracleState *s) {
    set_threads(s);
    const float diameter2 = s->particle_diameter * s->particle_diameter;
    const float w_d = cubic_kernel(s, s->particle_diameter); /* W(|(d,0,0)|)  WCSPH.py:102 */
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t p_i = 0; p_i < s->N; ++p_i) {
        float *a = &s->acceleration[3 * p_i];
        if (is_static_rigid_body(s, p_i)) { a[0] = a[1] = a[2] = 0.0f; continue; }
        float d_v[3] = {s->g[0], s->g[1], s->g[2]};
        if (s->material[p_i] != MATERIAL_FLUID) {
            /* dynamic rigid: a = g.  (Fluid threads add -f_v*rho0/rho_j == 0 to it
             * concurrently in the reference, WCSPH.py:124-125; numerically void.) */
            a[0] = d_v[0]; a[1] = d_v[1]; a[2] = d_v[2];
            continue;
        }
        NEIGHBOR_LOOP_BEGIN(s, p_i, p_j)
        const float *r = rr_;
        if (s->material[p_j] == MATERIAL_FLUID) { /* surface tension  WCSPH.py:93-102 */
            const float r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
            const float c = s->surface_tension / s->m[p_i] * s->m[p_j];
            const float w = (r2 > diameter2) ? cubic_kernel(s, norm3(r)) : w_d;
            d_v[0] -= c * r[0] * w; d_v[1] -= c * r[1] * w; d_v[2] -= c * r[2] * w;
        }
        /* viscosity  WCSPH.py:105-125 */
        const float *vi = &s->v[3 * p_i], *vj = &s->v[3 * p_j];
        const float v_xy = (vi[0] - vj[0]) * r[0] + (vi[1] - vj[1]) * r[1] + (vi[2] - vj[2]) * r[2];
        const float rn = norm3(r);
        if (s->material[p_j] == MATERIAL_FLUID) {
            float gw[3];
            cubic_kernel_derivative(s, r, gw);
            const float c = s->visc_d_nu * (s->m[p_j] / s->density[p_j]) * v_xy / (rn * rn + s->visc_eps);
            d_v[0] += c * gw[0]; d_v[1] += c * gw[1]; d_v[2] += c * gw[2];
        }
        /* solid neighbour: boundary_viscosity = 0.0 => f_v == 0 (WCSPH.py:117-125);
         * the scatter to a dynamic body adds -0 and is omitted. */
        NEIGHBOR_LOOP_END
        a[0] = d_v[0]; a[1] = d_v[1]; a[2] = d_v[2];
    }
}

/* WCSPH.py:46-85  compute_pressure_forces (+ task) */
void oracle_compute_pressure_forces(OracleState *s) {
    set_threads(s);
    const float rho0 = s->density_0;
#pragma omp parallel for schedule(static)
    for (int32_t p_i = 0; p_i < s->N; ++p_i) {
        if (s->material[p_i] != MATERIAL_FLUID) continue;
        s->density[p_i] = fmaxf(s->density[p_i], rho0);
        const float xr = s->density[p_i] / rho0;
        float pw;
        if ((s->perturb & 2) && s->exponent >= 1.0f && s->exponent <= 32.0f && s->exponent == (float)(int)s->exponent) {
            float r = 1.0f, b = xr; /* (test switch) square-and-multiply */
            for (int e = (int)s->exponent; e; e >>= 1) { if (e & 1) r *= b; b *= b; }
            pw = r;
        } else {
            pw = powf(xr, s->exponent);
        }
        s->pressure[p_i] = s->stiffness * (pw - 1.0f);
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t p_i = 0; p_i < s->N; ++p_i) {
        float *a = &s->acceleration[3 * p_i];
        if (is_static_rigid_body(s, p_i)) { a[0] = a[1] = a[2] = 0.0f; continue; }
        if (is_dynamic_rigid_body(s, p_i)) continue;
        float dv[3] = {0.0f, 0.0f, 0.0f};
        const float dpi = s->pressure[p_i] / (s->density[p_i] * s->density[p_i]);
        NEIGHBOR_LOOP_BEGIN(s, p_i, p_j)
        float gw[3];
        cubic_kernel_derivative(s, rr_, gw);
        if (s->material[p_j] == MATERIAL_FLUID) {
            const float density_j = s->density[p_j] * rho0 / rho0; /* WCSPH.py:53 */
            const float dpj = s->pressure[p_j] / (density_j * density_j);
            const float c = -rho0 * s->m_V[p_j] * (dpi + dpj);
            dv[0] += c * gw[0]; dv[1] += c * gw[1]; dv[2] += c * gw[2];
        } else if (s->material[p_j] == MATERIAL_SOLID) {
            const float dpj = s->pressure[p_i] / (rho0 * rho0);
            const float c = -rho0 * s->m_V[p_j] * (dpi + dpj);
            const float f_p[3] = {c * gw[0], c * gw[1], c * gw[2]};
            dv[0] += f_p[0]; dv[1] += f_p[1]; dv[2] += f_p[2];
            if (is_dynamic_rigid_body(s, p_j)) { /* two-way coupling scatter  WCSPH.py:67-68 */
                const float sc = rho0 / s->density[p_j];
                const float back[3] = {-f_p[0] * sc, -f_p[1] * sc, -f_p[2] * sc};
                atomic_add3(&s->acceleration[3 * p_j], back);
            }
        }
        NEIGHBOR_LOOP_END
        a[0] += dv[0]; a[1] += dv[1]; a[2] += dv[2]; /* fluid p_i: only this thread touches it */
    }
}

/* WCSPH.py:143-149  advect (symplectic Euler) */
void oracle_advect(OracleState *s) {
    set_threads(s);
#pragma omp parallel for schedule(static)
    for (int32_t p = 0; p < s->N; ++p) {
        if (!s->is_dynamic[p]) continue;
        for (int d = 0; d < 3; ++d) {
            s->v[3 * p + d] += s->dt * s->acceleration[3 * p + d];
            s->x[3 * p + d] += s->dt * s->v[3 * p + d];
        }
    }
}

/* sph_base.py:118-123, 149-179  simulate_collisions + enforce_boundary_3D */
void oracle_enforce_boundary_3D(OracleState *s, int32_t particle_type) {
    set_threads(s);
#pragma omp parallel for schedule(static)
    for (int32_t p = 0; p < s->N; ++p) {
        if (!(s->material[p] == particle_type && s->is_dynamic[p])) continue;
        float *x = &s->x[3 * p], *v = &s->v[3 * p];
        const float pos[3] = {x[0], x[1], x[2]};
        float n[3] = {0.0f, 0.0f, 0.0f};
        for (int d = 0; d < 3; ++d) {
            if (pos[d] > s->wall_hi[d]) { n[d] += 1.0f; x[d] = s->wall_hi[d]; }
            if (pos[d] <= s->padding) { n[d] += -1.0f; x[d] = s->padding; }
        }
        const float len = norm3(n);
        if (len > 1e-6f) {
            const float vec[3] = {n[0] / len, n[1] / len, n[2] / len};
            const float c_f = 0.5f;
            const float vd = v[0] * vec[0] + v[1] * vec[1] + v[2] * vec[2];
            for (int d = 0; d < 3; ++d) v[d] -= (1.0f + c_f) * vd * vec[d];
        }
    }
}

/* sph_base.py:182-192  compute_com */
static void compute_com(const OracleState *s, int32_t object_id, float cm[3]) {
    if (s->rigid_sums_f64) { /* same formula, f64 accumulators: isolates the f32 summation noise of the reference */
        double sm = 0.0, c[3] = {0.0, 0.0, 0.0};
        for (int32_t p = 0; p < s->N; ++p)
            if (is_dynamic_rigid_body(s, p) && s->object_id[p] == object_id) {
                const float mass = s->m_V0 * s->density[p];
                for (int d = 0; d < 3; ++d) c[d] += (double)(mass * s->x[3 * p + d]);
                sm += mass;
            }
        for (int d = 0; d < 3; ++d) cm[d] = (float)c[d] / (float)sm;
        return;
    }
    float sum_m = 0.0f;
    cm[0] = cm[1] = cm[2] = 0.0f;
    for (int32_t p = 0; p < s->N; ++p) {
        if (is_dynamic_rigid_body(s, p) && s->object_id[p] == object_id) {
            const float mass = s->m_V0 * s->density[p];
            cm[0] += mass * s->x[3 * p]; cm[1] += mass * s->x[3 * p + 1]; cm[2] += mass * s->x[3 * p + 2];
            sum_m += mass;
        }
    }
    cm[0] /= sum_m; cm[1] /= sum_m; cm[2] /= sum_m;
}

/* sph_base.py:87-89 */
void oracle_compute_rigid_rest_cm(OracleState *s, int32_t object_id) {
    compute_com(s, object_id, &s->rigid_rest_cm[3 * object_id]);
}

/* Rotation factor of the polar decomposition A = R S (ti.polar_decompose,
 * sph_base.py:212 -- third-party Taichi code, NOT in the reference tree:
 * parity unpinned for this function).  Taichi computes U,sig,V = svd3d(A) with
 * det U = det V = +1 and returns R = U V^T, i.e. the closest proper rotation.
 * Restated in f64: Jacobi eigen-decomposition of A^T A, U = A V sig^-1 with the
 * smallest singular direction completed by a cross product. */
static void polar_rotation(const float A_[9], float R_[9]) {
    double A[3][3], S[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = A_[3 * i + j];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        S[i][j] = 0; for (int k = 0; k < 3; ++k) S[i][j] += A[k][i] * A[k][j];
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = fabs(S[0][1]) + fabs(S[0][2]) + fabs(S[1][2]);
        if (off <= 1e-30 * (fabs(S[0][0]) + fabs(S[1][1]) + fabs(S[2][2])) + 1e-300) break;  /* cyclic Jacobi: quadratic, ~6 sweeps */
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            if (fabs(S[p][q]) < 1e-300) continue;
            double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 3; ++k) { /* S = S J */
                double skp = S[k][p], skq = S[k][q];
                S[k][p] = c * skp - sn * skq; S[k][q] = sn * skp + c * skq;
            }
            for (int k = 0; k < 3; ++k) { /* S = J^T S */
                double spk = S[p][k], sqk = S[q][k];
                S[p][k] = c * spk - sn * sqk; S[q][k] = sn * spk + c * sqk;
            }
            for (int k = 0; k < 3; ++k) {
                double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq;
            }
        }
    }
    /* sort eigenvalues descending */
    int idx[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j)
        if (S[idx[j]][idx[j]] > S[idx[i]][idx[i]]) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    double Vs[3][3], U[3][3], sig[3];
    for (int c = 0; c < 3; ++c) {
        for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][idx[c]];
        double e = S[idx[c]][idx[c]]; sig[c] = e > 0 ? sqrt(e) : 0.0;
    }
    /* make V a proper rotation */
    double detV = Vs[0][0] * (Vs[1][1] * Vs[2][2] - Vs[1][2] * Vs[2][1]) -
                  Vs[0][1] * (Vs[1][0] * Vs[2][2] - Vs[1][2] * Vs[2][0]) +
                  Vs[0][2] * (Vs[1][0] * Vs[2][1] - Vs[1][1] * Vs[2][0]);
    if (detV < 0) for (int r = 0; r < 3; ++r) Vs[r][2] = -Vs[r][2];
    if (sig[0] <= 1e-300) { /* A == 0: R = 0 (caller turns it into identity) */
        for (int i = 0; i < 9; ++i) R_[i] = 0.0f;
        return;
    }
    for (int c = 0; c < 2; ++c) {
        double u[3];
        for (int r = 0; r < 3; ++r) { u[r] = 0; for (int k = 0; k < 3; ++k) u[r] += A[r][k] * Vs[k][c]; }
        double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (c == 1) { /* orthogonalise against u0 for rank-deficient A */
            double d = u[0] * U[0][0] + u[1] * U[1][0] + u[2] * U[2][0];
            for (int r = 0; r < 3; ++r) u[r] -= d * U[r][0];
            n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        }
        if (n < 1e-300) { /* pick any unit vector orthogonal to the previous column(s) */
            double a[3] = {1, 0, 0};
            if (c == 1) { if (fabs(U[0][0]) > 0.9) { a[0] = 0; a[1] = 1; }
                double d = a[0] * U[0][0] + a[1] * U[1][0] + a[2] * U[2][0];
                for (int r = 0; r < 3; ++r) u[r] = a[r] - d * U[r][0]; }
            else for (int r = 0; r < 3; ++r) u[r] = a[r];
            n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        }
        for (int r = 0; r < 3; ++r) U[r][c] = u[r] / n;
    }
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double r = 0; for (int k = 0; k < 3; ++k) r += U[i][k] * Vs[j][k];
        R_[3 * i + j] = (float)r;
    }
}
void oracle_polar_rotation(const float A[9], float R[9]) { polar_rotation(A, R); }

/* sph_base.py:200-222  solve_constraints (shape matching) */
void oracle_solve_constraints(OracleState *s, int32_t object_id, float R_out[9]) {
    float cm[3];
    compute_com(s, object_id, cm);
    float A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double Ad[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float *rc = &s->rigid_rest_cm[3 * object_id];
    for (int32_t p = 0; p < s->N; ++p) {
        if (is_dynamic_rigid_body(s, p) && s->object_id[p] == object_id) {
            const float q[3] = {s->x_0[3 * p] - rc[0], s->x_0[3 * p + 1] - rc[1], s->x_0[3 * p + 2] - rc[2]};
            const float pp[3] = {s->x[3 * p] - cm[0], s->x[3 * p + 1] - cm[1], s->x[3 * p + 2] - cm[2]};
            const float w = s->m_V0 * s->density[p];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                A[3 * i + j] += w * (pp[i] * q[j]);
                Ad[3 * i + j] += (double)(w * (pp[i] * q[j]));
            }
        }
    }
    if (s->rigid_sums_f64) for (int i = 0; i < 9; ++i) A[i] = (float)Ad[i];
    float R[9];
    polar_rotation(A, R);
    int all_small = 1;
    for (int i = 0; i < 9; ++i) if (!(fabsf(R[i]) < 1e-6f)) all_small = 0;
    if (all_small) { for (int i = 0; i < 9; ++i) R[i] = 0.0f; R[0] = R[4] = R[8] = 1.0f; }
    for (int32_t p = 0; p < s->N; ++p) {
        if (is_dynamic_rigid_body(s, p) && s->object_id[p] == object_id) {
            const float q[3] = {s->x_0[3 * p] - rc[0], s->x_0[3 * p + 1] - rc[1], s->x_0[3 * p + 2] - rc[2]};
            for (int i = 0; i < 3; ++i) {
                const float goal = cm[i] + (R[3 * i] * q[0] + R[3 * i + 1] * q[1] + R[3 * i + 2] * q[2]);
                const float corr = (goal - s->x[3 * p + i]) * 1.0f;
                s->x[3 * p + i] += corr;
            }
        }
    }
    if (R_out) for (int i = 0; i < 9; ++i) R_out[i] = R[i];
}

/* WCSPH.py:152-156  substep */
void oracle_substep(OracleState *s) {
    oracle_compute_densities(s);
    oracle_compute_non_pressure_forces(s);
    oracle_compute_pressure_forces(s);
    oracle_advect(s);
}


/* ======================================================================================
 * DFSPH (DFSPH.py).  Same neighbour traversal, same f32 conventions.  Host-side loop
 * arithmetic (eta, averages, 1/dt) is Python f64 in the reference and double here.
 * ==================================================================================== */

/* DFSPH.py:22-47  compute_densities: formula identical to WCSPH.py:19-43 */
void oracle_dfsph_compute_densities(OracleState *s) { oracle_compute_densities(s); }

/* DFSPH.py:49-97  compute_non_pressure_forces: surface tension + viscosity as WCSPH.py:88-140; the solid
 * branch carries boundary_viscosity = 0.0, so it (and its scatter to dynamic bodies) adds exactly 0. */
void oracle_dfsph_compute_non_pressure_forces(OracleState *s) { oracle_compute_non_pressure_forces(s); }

/* grad_p_j = -m_V[p_j] * cubic_kernel_derivative(x_i - x_j)   (DFSPH.py:144, 151, 297, 303, 376, 383) */
static inline void grad_p(const OracleState *s, int32_t p_j, const float r[3], float g[3]) {
    float gw[3];
    cubic_kernel_derivative(s, r, gw);
    const float c = -s->m_V[p_j];
    g[0] = c * gw[0]; g[1] = c * gw[1]; g[2] = c * gw[2];
}
static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* DFSPH.py:116-154  compute_DFSPH_factor (+ task) */
void oracle_dfsph_compute_factor(OracleState *s) {
    set_threads(s);
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t p_i = 0; p_i < s->N; ++p_i) {
        if (s->material[p_i] != MATERIAL_FLUID) continue;
        float ret[4] = {0.0f, 0.0f, 0.0f, 0.0f}; /* grad_p_i, sum_grad_p_k */
        NEIGHBOR_LOOP_BEGIN(s, p_i, p_j)
        float g[3];
        grad_p(s, p_j, rr_, g);
        if (s->material[p_j] == MATERIAL_FLUID) ret[3] += dot3(g, g);
        ret[0] -= g[0]; ret[1] -= g[1]; ret[2] -= g[2]; /* both materials */
        NEIGHBOR_LOOP_END
        float sum_grad_p_k = ret[3];
        sum_grad_p_k += dot3(ret, ret);
        s->dfsph_factor[p_i] = sum_grad_p_k > 1e-6f ? -1.0f / sum_grad_p_k : 0.0f;
    }
}

/* sum_j m_V_j (v_i - v_j) . gradW(x_i - x_j) over both materials, and the neighbour count
 * (DFSPH.py:183-197 compute_density_change_task / :212-221 compute_density_adv_task) */
static inline float velocity_divergence(const OracleState *s, int32_t p_i, int32_t *count) {
    float acc = 0.0f;
    int32_t n = 0;
    const float *vi = &s->v[3 * p_i];
    NEIGHBOR_LOOP_BEGIN(s, p_i, p_j)
    float gw[3];
    cubic_kernel_derivative(s, rr_, gw);
    const float *vj = &s->v[3 * p_j];
    const float dv[3] = {vi[0] - vj[0], vi[1] - vj[1], vi[2] - vj[2]};
    acc += s->m_V[p_j] * dot3(dv, gw);
    n += 1;
    NEIGHBOR_LOOP_END
    if (count) *count = n;
    return acc;
}

/* DFSPH.py:157-180  compute_density_change */
void oracle_dfsph_compute_density_change(OracleState *s) {
    set_threads(s);
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t p_i = 0; p_i < s->N; ++p_i) {
        if (s->material[p_i] != MATERIAL_FLUID) continue;
        int32_t num_neighbors = 0;
        float density_adv = fmaxf(velocity_divergence(s, p_i, &num_neighbors), 0.0f); /* only positive divergence */
        if (num_neighbors < 20) density_adv = 0.0f;                                    /* dim == 3  DFSPH.py:172-174 */
        s->density_adv[p_i] = density_adv;
    }
}

/* DFSPH.py:200-209  compute_density_adv */
void oracle_dfsph_compute_density_adv(OracleState *s) {
    set_threads(s);
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t p_i = 0; p_i < s->N; ++p_i) {
        if (s->material[p_i] != MATERIAL_FLUID) continue;
        const float delta = velocity_divergence(s, p_i, NULL);
        const float density_adv = s->density[p_i] / s->density_0 + s->dt * delta;
        s->density_adv[p_i] = fmaxf(density_adv, 1.0f);
    }
}

/* DFSPH.py:224-230  compute_density_error: f32 sum in index order (a serial run of the reference's reduction) */
float oracle_dfsph_compute_density_error(const OracleState *s, float offset) {
    float density_error = 0.0f;
    for (int32_t I = 0; I < s->N; ++I)
        if (s->material[I] == MATERIAL_FLUID) density_error += s->density_0 * s->density_adv[I] - offset;
    return density_error;
}

/* DFSPH.py:233-237  multiply_time_step (only ever applied to dfsph_factor) */
void oracle_dfsph_multiply_time_step(OracleState *s, float time_step) {
    for (int32_t I = 0; I < s->N; ++I)
        if (s->material[I] == MATERIAL_FLUID) s->dfsph_factor[I] *= time_step;
}

/* DFSPH.py:285-321 divergence_solver_iteration_kernel (+ task)  [pressure = 0]
 * DFSPH.py:356-394 pressure_solve_iteration_kernel (+ task)     [pressure = 1]
 * The two differ in b (density_adv vs density_adv - 1), in accumulating dv vs updating v[p_i] in place, and in
 * how the reaction on a dynamic body divides by dt. */
static void solver_iteration_kernel(OracleState *s, int pressure) {
    set_threads(s);
    const float dt = s->dt;
    const float off = pressure ? 1.0f : 0.0f;
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t p_i = 0; p_i < s->N; ++p_i) {
        if (s->material[p_i] != MATERIAL_FLUID) continue;
        const float b_i = s->density_adv[p_i] - off;
        const float k_i = b_i * s->dfsph_factor[p_i];
        float dv[3] = {0.0f, 0.0f, 0.0f};
        float *v = &s->v[3 * p_i];
        NEIGHBOR_LOOP_BEGIN(s, p_i, p_j)
        if (s->material[p_j] == MATERIAL_FLUID) {
            const float b_j = s->density_adv[p_j] - off;
            const float k_j = b_j * s->dfsph_factor[p_j];
            const float k_sum = k_i + s->density_0 / s->density_0 * k_j;
            if (fabsf(k_sum) > s->m_eps) {
                float g[3];
                grad_p(s, p_j, rr_, g);
                const float c = dt * k_sum;
                if (pressure) { v[0] -= c * g[0]; v[1] -= c * g[1]; v[2] -= c * g[2]; }
                else { dv[0] -= c * g[0]; dv[1] -= c * g[1]; dv[2] -= c * g[2]; }
            }
        } else if (s->material[p_j] == MATERIAL_SOLID) {
            if (fabsf(k_i) > s->m_eps) {
                float g[3];
                grad_p(s, p_j, rr_, g);
                const float c = -dt * 1.0f * k_i;
                const float vel_change[3] = {c * g[0], c * g[1], c * g[2]};
                if (pressure) { v[0] += vel_change[0]; v[1] += vel_change[1]; v[2] += vel_change[2]; }
                else { dv[0] += vel_change[0]; dv[1] += vel_change[1]; dv[2] += vel_change[2]; }
                if (is_dynamic_rigid_body(s, p_j)) {
                    float back[3];
                    for (int d = 0; d < 3; ++d)
                        back[d] = pressure ? -vel_change[d] * 1.0f / dt * s->density[p_i] / s->density[p_j]           /* :394 */
                                           : -vel_change[d] * (1.0f / dt) * s->density[p_i] / s->density[p_j];        /* :321 */
                    atomic_add3(&s->acceleration[3 * p_j], back);
                }
            }
        }
        NEIGHBOR_LOOP_END
        if (!pressure) { v[0] += dv[0]; v[1] += dv[1]; v[2] += dv[2]; }
    }
}
void oracle_dfsph_divergence_solver_iteration_kernel(OracleState *s) { solver_iteration_kernel(s, 0); }
void oracle_dfsph_pressure_solve_iteration_kernel(OracleState *s) { solver_iteration_kernel(s, 1); }

/* DFSPH.py:278-283 */
double oracle_dfsph_divergence_solver_iteration(OracleState *s) {
    oracle_dfsph_divergence_solver_iteration_kernel(s);
    oracle_dfsph_compute_density_change(s);
    const float density_err = oracle_dfsph_compute_density_error(s, 0.0f);
    return (double)density_err / s->fluid_particle_num;
}

/* DFSPH.py:240-275 */
void oracle_dfsph_divergence_solve(OracleState *s) {
    oracle_dfsph_compute_density_change(s);
    const double dt = (double)s->dt;
    const double inv_dt = 1 / dt;
    oracle_dfsph_multiply_time_step(s, (float)inv_dt);
    int32_t m_iterations_v = 0;
    double avg_density_err = 0.0;
    while (m_iterations_v < 1 || m_iterations_v < s->m_max_iterations_v) {
        avg_density_err = oracle_dfsph_divergence_solver_iteration(s);
        const double eta = 1.0 / dt * s->max_error_V * 0.01 * (double)s->density_0;
        if (avg_density_err <= eta) break;
        m_iterations_v += 1;
    }
    s->last_iterations_v = m_iterations_v;
    s->last_avg_err_v = avg_density_err;
    oracle_dfsph_multiply_time_step(s, (float)dt);
}

/* DFSPH.py:350-354 */
double oracle_dfsph_pressure_solve_iteration(OracleState *s) {
    oracle_dfsph_pressure_solve_iteration_kernel(s);
    oracle_dfsph_compute_density_adv(s);
    const float density_err = oracle_dfsph_compute_density_error(s, s->density_0);
    return (double)density_err / s->fluid_particle_num;
}

/* DFSPH.py:324-348 */
void oracle_dfsph_pressure_solve(OracleState *s) {
    const double dt = (double)s->dt;
    const double inv_dt2 = 1 / (dt * dt);
    oracle_dfsph_compute_density_adv(s);
    oracle_dfsph_multiply_time_step(s, (float)inv_dt2);
    int32_t m_iterations = 0;
    double avg_density_err = 0.0;
    while (m_iterations < 1 || m_iterations < s->m_max_iterations) {
        avg_density_err = oracle_dfsph_pressure_solve_iteration(s);
        const double eta = s->max_error * 0.01 * (double)s->density_0;
        if (avg_density_err <= eta) break;
        m_iterations += 1;
    }
    s->last_iterations = m_iterations;
    s->last_avg_err = avg_density_err;
}

/* DFSPH.py:388-394  predict_velocity */
void oracle_dfsph_predict_velocity(OracleState *s) {
    for (int32_t p = 0; p < s->N; ++p)
        if (s->is_dynamic[p] && s->material[p] == MATERIAL_FLUID)
            for (int d = 0; d < 3; ++d) s->v[3 * p + d] += s->dt * s->acceleration[3 * p + d];
}

/* DFSPH.py:100-107  advect: only dynamic rigid particles integrate their acceleration here */
void oracle_dfsph_advect(OracleState *s) {
    for (int32_t p = 0; p < s->N; ++p) {
        if (!s->is_dynamic[p]) continue;
        for (int d = 0; d < 3; ++d) {
            if (is_dynamic_rigid_body(s, p)) s->v[3 * p + d] += s->dt * s->acceleration[3 * p + d];
            s->x[3 * p + d] += s->dt * s->v[3 * p + d];
        }
    }
}

/* DFSPH.py:400-408  substep */
void oracle_dfsph_substep(OracleState *s) {
    oracle_dfsph_compute_densities(s);
    oracle_dfsph_compute_factor(s);
    if (s->enable_divergence_solver) oracle_dfsph_divergence_solve(s);
    oracle_dfsph_compute_non_pressure_forces(s);
    oracle_dfsph_predict_velocity(s);
    oracle_dfsph_pressure_solve(s);
    oracle_dfsph_advect(s);
}

/* sph_base.py:247-260  solve_rigid_body;  dyn_ids = ids of dynamic RigidBodies */
void oracle_solve_rigid_body(OracleState *s, const int32_t *dyn_ids, int32_t n_dyn) {
    for (int32_t k = 0; k < n_dyn; ++k) {
        oracle_solve_constraints(s, dyn_ids[k], NULL);
        oracle_enforce_boundary_3D(s, MATERIAL_SOLID);
    }
}

/* sph_base.py:263-271  step.  phase_ms (nullable) receives sort / neighbour /
 * force / integrate wall-clock milliseconds accumulated over the call. */
static double now_ms(void) {
#ifdef _OPENMP
    return omp_get_wtime() * 1e3;
#else
    return 0.0;
#endif
}
void oracle_step(OracleState *s, const int32_t *dyn_ids, int32_t n_dyn, int32_t n_steps, double *phase_ms) {
    for (int32_t it = 0; it < n_steps; ++it) {
        double t0 = now_ms();
        oracle_initialize_particle_system(s);
        double t1 = now_ms();
        oracle_compute_moving_boundary_volume(s);
        double t2, t3;
        if (s->simulation_method == 4) { /* DFSPHSolver.substep: density + factor | solvers + forces | advect */
            oracle_dfsph_compute_densities(s);
            oracle_dfsph_compute_factor(s);
            t2 = now_ms();
            if (s->enable_divergence_solver) oracle_dfsph_divergence_solve(s);
            oracle_dfsph_compute_non_pressure_forces(s);
            oracle_dfsph_predict_velocity(s);
            oracle_dfsph_pressure_solve(s);
            t3 = now_ms();
            oracle_dfsph_advect(s);
        } else {
            oracle_compute_densities(s);
            t2 = now_ms();
            oracle_compute_non_pressure_forces(s);
            oracle_compute_pressure_forces(s);
            t3 = now_ms();
            oracle_advect(s);
        }
        oracle_solve_rigid_body(s, dyn_ids, n_dyn);
        oracle_enforce_boundary_3D(s, MATERIAL_FLUID);
        double t4 = now_ms();
        if (phase_ms) { phase_ms[0] += t1 - t0; phase_ms[1] += t2 - t1; phase_ms[2] += t3 - t2; phase_ms[3] += t4 - t3; }
    }
}

/* sph_base.py:80-85  initialize */
void oracle_initialize(OracleState *s, const int32_t *rigid_body_ids, int32_t n_rb) {
    oracle_initialize_particle_system(s);
    for (int32_t k = 0; k < n_rb; ++k) oracle_compute_rigid_rest_cm(s, rigid_body_ids[k]);
    oracle_compute_static_boundary_volume(s);
    oracle_compute_moving_boundary_volume(s);
}

/* exposed helpers for the KAT tests */
float oracle_cubic_kernel(const OracleState *s, float r) { return cubic_kernel(s, r); }
void oracle_cubic_kernel_derivative(const OracleState *s, const float r[3], float out[3]) {
    cubic_kernel_derivative(s, r, out);
}
int32_t oracle_sizeof_state(void) { return (int32_t)sizeof(OracleState); }
int32_t oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
