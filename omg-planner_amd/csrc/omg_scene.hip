// omg_scene.hip — scenes that change while they are resident in HBM (include/omg_hip.h section 8, ABI 8).
//
// The reference rebuilds its per-object parameters on every call (Cost.compute_obstacle_cost_layer, omg/cost.py:296-335) and a
// perception frame replaces an obstacle's volume with a fresh point-cloud SDF (PointEnv.compute_sdf_from_points,
// omg/core.py:426-457).  Here the object table and the SDF pool live on the device; what a change needs is
//   * a new pose: 12 floats of the record (DeviceScenes.set_object_pose — a 48-byte copy, no kernel);
//   * a new volume: the grid itself (already on the device: omgx_point_cloud_sdf writes it), the record's limits and the
//     object's INFLUENCE REGION — the rounded box outside which a lookup adds nothing, which the goal-set kernels cull with.
// omgx_fit_influence_region computes that region on the device: the algorithm of scenes.influence_rbox / tighten_far_boxes
// (omg-planner_amd/scenes.py, the specification; float64 throughout, the same expressions in the same order), as five small
// launches on the caller's stream — no host pass over the voxels, no synchronisation.  omgx_object_set_grid then writes the
// record's grid fields on the device (derived constants included), so a frame's update is stream-ordered end to end.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "omg_device.h"
#include "omg_host.h"

namespace {

#define RG_REFINE 4   // scenes.RBOX_REFINE
#define RG_CANDS 14   // 9 eroded bounding boxes of the needed windows + 5 shrunk bounding boxes of the non-positive voxels

struct RegionStats {
    unsigned int maxabs_bits;      // max |g| over the finite voxels (float bits; non-negative floats order like unsigned ints)
    int nmin[3], nmax[3];          // bounding box of the needed windows (window indices)
    int qmin[3], qmax[3];          // bounding box of the voxels <= 0
    unsigned long long count;      // needed windows
    int ncand, pad_;
    double cand_c[RG_CANDS][3], cand_h[RG_CANDS][3];  // candidate inner boxes (centre, half-widths) in grid coordinates
    unsigned long long r2bits[RG_CANDS];              // max over the kept boxes of the squared distance (double bits, >= 0)
};

struct RegionArgs {
    const float* g;     // [X][Y][Z]
    int X, Y, Z;
    double eps, clr;    // float(np.float32(epsilon)), float(np.float32(clearance))
    double vox[3];      // voxel extents (float32 quotient widened)
    RegionStats* st;
    unsigned char* need;  // [X*Y*Z]
    omgx_object* rec;
};

__device__ __forceinline__ void rg_init(RegionStats* st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->maxabs_bits = 0u;
    for (int k = 0; k < 3; ++k) { st->nmin[k] = 0x7fffffff; st->nmax[k] = -1; st->qmin[k] = 0x7fffffff; st->qmax[k] = -1; }
    st->count = 0ull;
    st->ncand = 0;
    for (int c = 0; c < RG_CANDS; ++c) st->r2bits[c] = 0ull;
}

__device__ __forceinline__ void rg_maxabs(const float* __restrict__ g, int64_t N, RegionStats* st) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = __builtin_fabsf(g[i]);
        if (a <= 3.4028234663852886e38f) m = a > m ? a : m;  // finite (NaN fails the comparison)
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(&st->maxabs_bits, __float_as_uint(m));
}

// the grid in float64 with the linear extension layer on every axis (voxel index -1), numpy's order of construction: axis 0 first,
// then axis 1 on the extended array, then axis 2 (scenes._needed_windows)
__device__ __forceinline__ double rg_f0(const RegionArgs& a, int x, int y, int z) {
    const int64_t sy = a.Z, sx = (int64_t)a.Y * a.Z;
    if (x >= 0) return (double)a.g[x * sx + y * sy + z];
    return (double)a.g[y * sy + z] * 2.0 - (double)a.g[sx + y * sy + z];
}
__device__ __forceinline__ double rg_f1(const RegionArgs& a, int x, int y, int z) {
    if (y >= 0) return rg_f0(a, x, y, z);
    return rg_f0(a, x, 0, z) * 2.0 - rg_f0(a, x, 1, z);
}
__device__ __forceinline__ double rg_ge(const RegionArgs& a, int x, int y, int z) {
    const double v = z >= 0 ? rg_f1(a, x, y, z) : rg_f1(a, x, y, 0) * 2.0 - rg_f1(a, x, y, 1);
    return (v - v == 0.0) ? v : -__builtin_inf();  // non-finite (NaN, +-inf): reachable
}

__device__ __forceinline__ double rg_margin(const RegionStats* st) {
    const double mx = (double)__uint_as_float(st->maxabs_bits);
    return 1e-5 * (mx > 1.0 ? mx : 1.0);
}

// one thread per window w (shape = grid dims): window w is the trilinear polynomial of voxels w - 1 .. w per axis
__device__ __forceinline__ void rg_need(const RegionArgs& a) {
    const int64_t N = (int64_t)a.X * a.Y * a.Z;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double margin = rg_margin(a.st);
    const double eps_t = a.eps + margin, clr_t = a.clr + margin;
    bool need = false, neg = false;
    int x = 0, y = 0, z = 0;
    if (id < N) {
        z = (int)(id % a.Z); y = (int)((id / a.Z) % a.Y); x = (int)(id / ((int64_t)a.Z * a.Y));
        double m = __builtin_inf();
#pragma unroll
        for (int dx = -1; dx <= 0; ++dx)
#pragma unroll
            for (int dy = -1; dy <= 0; ++dy)
#pragma unroll
                for (int dz = -1; dz <= 0; ++dz) {
                    const double v = rg_ge(a, x + dx, y + dy, z + dz);
                    m = v < m ? v : m;
                }
        need = (m <= eps_t) || (m < clr_t);
        a.need[id] = need ? 1 : 0;
        neg = a.g[id] <= 0.0f;
    }
    // bounding boxes: wave-level reduction, then one atomic per wave and bound
    const unsigned long long bn = __ballot(need), bq = __ballot(neg);
    const int big = 0x7fffffff;
    int v[12] = {need ? x : big, need ? y : big, need ? z : big, need ? x : -1, need ? y : -1, need ? z : -1,
                 neg ? x : big, neg ? y : big, neg ? z : big, neg ? x : -1, neg ? y : -1, neg ? z : -1};
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const bool is_min = (k % 6) < 3;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(v[k], off, 64);
            v[k] = is_min ? (o < v[k] ? o : v[k]) : (o > v[k] ? o : v[k]);
        }
    }
    if ((threadIdx.x & 63) == 0) {
        if (bn) {
            for (int k = 0; k < 3; ++k) { atomicMin(&a.st->nmin[k], v[k]); atomicMax(&a.st->nmax[k], v[3 + k]); }
            atomicAdd(&a.st->count, (unsigned long long)__popcll(bn));
        }
        if (bq)
            for (int k = 0; k < 3; ++k) { atomicMin(&a.st->qmin[k], v[6 + k]); atomicMax(&a.st->qmax[k], v[9 + k]); }
    }
}

// the family of candidate inner boxes (scenes.influence_rbox): one thread
__device__ __forceinline__ void rg_cands(const RegionArgs& a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    RegionStats* st = a.st;
    if (st->count == 0ull) return;
    double bc[3], bh[3];
    double rmax = __builtin_inf();
    for (int k = 0; k < 3; ++k) {
        const double nlo = (double)st->nmin[k] - 0.5, nhi = (double)st->nmax[k] + 0.5;
        bc[k] = (nlo + nhi) / 2; bh[k] = (nhi - nlo) / 2;
        const double e = bh[k] * a.vox[k];
        rmax = e < rmax ? e : rmax;
    }
    int n = 0;
    for (int j = 0; j < 9; ++j, ++n)
        for (int k = 0; k < 3; ++k) {
            st->cand_c[n][k] = bc[k];
            const double h = bh[k] - (rmax * (double)j / 8.0) / a.vox[k];
            st->cand_h[n][k] = h > 0.0 ? h : 0.0;
        }
    if (st->qmax[0] >= 0) {
        const double sh[5] = {0.0, 0.25, 0.5, 0.75, 1.0};
        for (int j = 0; j < 5; ++j, ++n)
            for (int k = 0; k < 3; ++k) {
                const double qlo = (double)st->qmin[k] + 0.5, qhi = (double)st->qmax[k] + 0.5;  // voxel i sits at grid coordinate i + 0.5
                st->cand_c[n][k] = (qlo + qhi) / 2;
                st->cand_h[n][k] = (qhi - qlo) / 2 * (1.0 - sh[j]);
            }
    }
    st->ncand = n;
}

// One wave per 64 consecutive windows.  A needed window with a face neighbour that is not needed (the outside of the array is not)
// is a BOUNDARY window: it is split RG_REFINE times per axis — lane = sub-cell — and contributes the bounding box of the
// sub-cells that can still reach the thresholds; every other needed window contributes its whole box (they add nothing to the
// maximum beyond the layer next to the boundary, which scenes.influence_rbox lists: a deeper window lies between two listed
// boxes along every axis, so none of its corners is an extreme point).  Every lane keeps, per candidate, the largest squared
// distance from the candidate's inner box to the far corner of a box it has seen.
__device__ __forceinline__ void rg_boxes(const RegionArgs& a, double (*sc)[3], double (*sh)[3]) {
    RegionStats* st = a.st;
    const int ncand = st->ncand;
    if (ncand == 0) return;
    for (int e = threadIdx.x; e < RG_CANDS * 3; e += blockDim.x) { sc[e / 3][e % 3] = st->cand_c[e / 3][e % 3]; sh[e / 3][e % 3] = st->cand_h[e / 3][e % 3]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t N = (int64_t)a.X * a.Y * a.Z;
    const double margin = rg_margin(st);
    const double eps_t = a.eps + margin, clr_t = a.clr + margin;
    double r2[RG_CANDS];
#pragma unroll
    for (int c = 0; c < RG_CANDS; ++c) r2[c] = 0.0;
    auto feed = [&](const double* blo, const double* bhi) {
#pragma unroll
        for (int c = 0; c < RG_CANDS; ++c) {
            if (c < ncand) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double u = __builtin_fabs(blo[k] - sc[c][k]), w = __builtin_fabs(bhi[k] - sc[c][k]);
                    double d = (u > w ? u : w) - sh[c][k];
                    d = (d > 0.0 ? d : 0.0) * a.vox[k];
                    s = k == 0 ? d * d : s + d * d;
                }
                r2[c] = s > r2[c] ? s : r2[c];
            }
        }
    };
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t base = wave_id * 64; base < N; base += nwaves * 64) {
        const int64_t id = base + lane;
        bool need = false, boundary = false;
        int x = 0, y = 0, z = 0;
        if (id < N) {
            need = a.need[id] != 0;
            z = (int)(id % a.Z); y = (int)((id / a.Z) % a.Y); x = (int)(id / ((int64_t)a.Z * a.Y));
            if (need) {
                const int64_t sy = a.Z, sx = (int64_t)a.Y * a.Z;
                const bool inner = x > 0 && x < a.X - 1 && y > 0 && y < a.Y - 1 && z > 0 && z < a.Z - 1 && a.need[id - sx] && a.need[id + sx] &&
                                   a.need[id - sy] && a.need[id + sy] && a.need[id - 1] && a.need[id + 1];
                boundary = !inner;
                if (inner) {
                    const double blo[3] = {x - 0.5, y - 0.5, z - 0.5}, bhi[3] = {x + 0.5, y + 0.5, z + 0.5};
                    feed(blo, bhi);
                }
            }
        }
        unsigned long long todo = __ballot(boundary);
        while (todo) {  // wave-uniform: one boundary window at a time, lane = sub-cell (i, j, k)
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int wx = __builtin_amdgcn_readlane(x, src), wy = __builtin_amdgcn_readlane(y, src), wz = __builtin_amdgcn_readlane(z, src);
            double c[2][2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 2; ++r) c[p][q][r] = rg_ge(a, wx - 1 + p, wy - 1 + q, wz - 1 + r);
            const int i = lane >> 4, j = (lane >> 2) & 3, k = lane & 3;
            double m = __builtin_inf();
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const double fx = (double)(i + p) / RG_REFINE, fy = (double)(j + q) / RG_REFINE, fz = (double)(k + r) / RG_REFINE;
                        auto lerp = [](double u, double w, double t) { return u + (w - u) * t; };
                        double v = lerp(lerp(lerp(c[0][0][0], c[1][0][0], fx), lerp(c[0][1][0], c[1][1][0], fx), fy),
                                        lerp(lerp(c[0][0][1], c[1][0][1], fx), lerp(c[0][1][1], c[1][1][1], fx), fy), fz);
                        v = (v - v == 0.0) ? v : -__builtin_inf();
                        m = v < m ? v : m;
                    }
            const unsigned long long kept = __ballot((m <= eps_t) || (m < clr_t));
            if (kept == 0ull) continue;
            unsigned long long m4 = kept | (kept >> 16) | (kept >> 32) | (kept >> 48);
            m4 &= 0xffffull;                                   // OR over i: bit 4 j + k
            const unsigned mz = (unsigned)((m4 | (m4 >> 4) | (m4 >> 8) | (m4 >> 12)) & 0xfull);  // OR over i, j: bit k
            unsigned mx = 0, my = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if ((kept >> (16 * t)) & 0xffffull) mx |= 1u << t;
                if ((m4 >> (4 * t)) & 0xfull) my |= 1u << t;
            }
            const int w3[3] = {wx, wy, wz};
            const unsigned ms[3] = {mx, my, mz};
            double blo[3], bhi[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int first = __builtin_ctz(ms[t]), last = 31 - __builtin_clz(ms[t]);
                blo[t] = (double)w3[t] - 0.5 + (double)first / RG_REFINE;
                bhi[t] = (double)w3[t] - 0.5 + (double)(last + 1) / RG_REFINE;
            }
            feed(blo, bhi);
        }
    }
#pragma unroll
    for (int c = 0; c < RG_CANDS; ++c) {
        double v = r2[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
        if (lane == 0 && c < ncand && v > 0.0) atomicMax(&st->r2bits[c], (unsigned long long)__double_as_longlong(v));
    }
}

__device__ __forceinline__ float rg_next_up(float v) {  // np.nextafter(v, +inf) for finite v
    if (v == 0.0f) return __uint_as_float(1u);
    const unsigned b = __float_as_uint(v);
    return __uint_as_float(v > 0.0f ? b + 1u : b - 1u);
}

// the smallest rounded box among the candidates -> the record (scenes.tighten_far_boxes): one thread
__device__ __forceinline__ void rg_pick(const RegionArgs& a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    RegionStats* st = a.st;
    omgx_object* r = a.rec;
    if (st->count == 0ull) {  // nothing reachable: an empty region rejects every point
        for (int k = 0; k < 3; ++k) { r->rb_c[k] = 0.0f; r->rb_h[k] = 0.0f; }
        r->rb_r = 0.0f; r->rb_r2 = -1.0f;
        return;
    }
    int best = -1;
    double best_vol = 0.0, bR = 0.0;
    const double pi = 3.141592653589793;
    for (int c = 0; c < st->ncand; ++c) {
        const double R = sqrt(__longlong_as_double((long long)st->r2bits[c]));
        const double h0 = st->cand_h[c][0] * a.vox[0], h1 = st->cand_h[c][1] * a.vox[1], h2 = st->cand_h[c][2] * a.vox[2];
        const double vol = 8 * ((h0 * h1) * h2) + 8 * R * (h0 * h1 + h1 * h2 + h0 * h2) + 2 * pi * R * R * ((h0 + h1) + h2) + 4.0 / 3.0 * pi * (R * R * R);
        if (best < 0 || vol < best_vol) { best = c; best_vol = vol; bR = R; }
    }
    double vmin = a.vox[0] < a.vox[1] ? a.vox[0] : a.vox[1];
    vmin = a.vox[2] < vmin ? a.vox[2] : vmin;
    const double R = bR + 0.01 * vmin + 1e-6;
    for (int k = 0; k < 3; ++k) {
        r->rb_c[k] = (float)(st->cand_c[best][k] * a.vox[k]);
        r->rb_h[k] = rg_next_up((float)(st->cand_h[best][k] * a.vox[k]));
    }
    const float rr = rg_next_up((float)R);
    r->rb_r = rr;
    r->rb_r2 = rg_next_up((float)((double)rr * (double)rr));
}

// ---- one object (omgx_fit_influence_region) ---------------------------------------------------------------------------------
__global__ void k_region_init(RegionStats* st) { rg_init(st); }
__global__ __launch_bounds__(256) void k_region_maxabs(const float* __restrict__ g, int64_t N, RegionStats* st) { rg_maxabs(g, N, st); }
__global__ __launch_bounds__(256) void k_region_need(RegionArgs a) { rg_need(a); }
__global__ void k_region_cands(RegionArgs a) { rg_cands(a); }
__global__ __launch_bounds__(256) void k_region_boxes(RegionArgs a) {
    __shared__ double sc[RG_CANDS][3], sh[RG_CANDS][3];
    rg_boxes(a, sc, sh);
}
__global__ void k_region_pick(RegionArgs a) { rg_pick(a); }

// ---- a whole table (omgx_fit_influence_regions, ABI 10): blockIdx.y = entry of the fit list; every launch reads what it needs —
// dims, limits, thresholds, the volume's place in the pool — from the object's record ON THE DEVICE.  Same bodies, same bits.
struct RegionBatch {
    omgx_object* objects;
    const float* pool;
    const int32_t* fit_list;   // [n_fit] object indices: one per distinct (volume, epsilon, clearance)
    const int64_t* need_off;   // [n_fit] byte offset of the entry's window flags in `need`
    RegionStats* stats;        // [n_fit]
    unsigned char* need;
    int n_fit;
};
__device__ __forceinline__ RegionArgs rg_entry(const RegionBatch& b, int f) {
    omgx_object* r = b.objects + b.fit_list[f];
    RegionArgs a;
    a.g = b.pool + r->grid_offset;
    a.X = r->dim[0]; a.Y = r->dim[1]; a.Z = r->dim[2];
    a.eps = (double)r->epsilon; a.clr = (double)r->clearance;
    for (int k = 0; k < 3; ++k) a.vox[k] = (double)((r->hi[k] - r->lo[k]) / (float)r->dim[k]);
    a.st = b.stats + f;
    a.need = b.need + b.need_off[f];
    a.rec = r;
    return a;
}
__global__ void k_regions_init(RegionBatch b) { rg_init(b.stats + blockIdx.y); }
__global__ __launch_bounds__(256) void k_regions_maxabs(RegionBatch b) {
    const RegionArgs a = rg_entry(b, blockIdx.y);
    rg_maxabs(a.g, (int64_t)a.X * a.Y * a.Z, a.st);
}
__global__ __launch_bounds__(256) void k_regions_need(RegionBatch b) {
    const RegionArgs a = rg_entry(b, blockIdx.y);
    if ((int64_t)blockIdx.x * blockDim.x >= (int64_t)a.X * a.Y * a.Z) return;  // the grid is sized for the largest volume
    rg_need(a);
}
__global__ void k_regions_cands(RegionBatch b) { rg_cands(rg_entry(b, blockIdx.y)); }
__global__ __launch_bounds__(256) void k_regions_boxes(RegionBatch b) {
    __shared__ double sc[RG_CANDS][3], sh[RG_CANDS][3];
    rg_boxes(rg_entry(b, blockIdx.y), sc, sh);
}
__global__ void k_regions_pick(RegionBatch b) { rg_pick(rg_entry(b, blockIdx.y)); }
// the records that share a fitted entry's volume and thresholds take its region: copy_src[o] = the fitted object, or -1
__global__ __launch_bounds__(256) void k_regions_copy(omgx_object* objects, const int32_t* copy_src, int n) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    const int s = copy_src[o];
    if (s < 0 || s == o) return;
    for (int k = 0; k < 3; ++k) { objects[o].rb_c[k] = objects[s].rb_c[k]; objects[o].rb_h[k] = objects[s].rb_h[k]; }
    objects[o].rb_r = objects[s].rb_r; objects[o].rb_r2 = objects[s].rb_r2;
}

// the grid fields of a record and everything derived from them (scenes.pack_table + finish_records): one thread
struct SetGridArgs {
    omgx_object* rec;
    float lo[3], hi[3];
    int dim[3];
    float delta;
    int64_t grid_offset;
};
__global__ void k_object_set_grid(SetGridArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    omgx_object* r = a.rec;
    for (int k = 0; k < 3; ++k) {
        r->lo[k] = a.lo[k]; r->hi[k] = a.hi[k]; r->dim[k] = a.dim[k];
        const float w = a.hi[k] - a.lo[k];
        r->inv_extent[k] = 1.0 / (double)w;
        const bool ok = w > 0.0f && a.dim[k] > 0;
        const float vox = w / (float)(a.dim[k] > 0 ? a.dim[k] : 1);
        r->rb_c[k] = ok ? 0.5f * w : 0.0f;                             // the loose region: the grid with 1.5 voxels of slack
        r->rb_h[k] = ok ? 0.5f * w + 1.5f * vox : __builtin_inff();
    }
    r->rb_r = 0.0f; r->rb_r2 = 0.0f;
    r->delta = a.delta;
    r->inv_delta = 1.0 / (double)a.delta;
    r->grid_offset = a.grid_offset;
}

}  // namespace

extern "C" int64_t omgx_region_scratch_bytes(int32_t dx, int32_t dy, int32_t dz) {
    if (dx < 1 || dy < 1 || dz < 1) return 0;
    return (int64_t)((sizeof(RegionStats) + 255) & ~(size_t)255) + (int64_t)dx * dy * dz;
}

extern "C" int omgx_object_set_grid(omgx_object* object, const float* h_lo, const float* h_hi, const int32_t* h_dims, float delta,
                                    int64_t grid_offset, void* stream) {
    if (!object || !h_lo || !h_hi || !h_dims || !(delta > 0.0f) || grid_offset < 0) return OMGX_ERR_INVALID;
    SetGridArgs a;
    a.rec = object;
    for (int k = 0; k < 3; ++k) {
        if (h_dims[k] < 1) return OMGX_ERR_INVALID;
        a.lo[k] = h_lo[k]; a.hi[k] = h_hi[k]; a.dim[k] = h_dims[k];
    }
    a.delta = delta; a.grid_offset = grid_offset;
    hipLaunchKernelGGL(k_object_set_grid, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    OMGX_CHECK_LAUNCH("k_object_set_grid");
    return OMGX_OK;
}

extern "C" int omgx_fit_influence_region(omgx_object* object, const float* grid, const int32_t* h_dims, const float* h_lo,
                                         const float* h_hi, float epsilon, float clearance, void* scratch, void* stream) {
    if (!object || !grid || !h_dims || !h_lo || !h_hi || !scratch) return OMGX_ERR_INVALID;
    if (h_dims[0] < 1 || h_dims[1] < 1 || h_dims[2] < 1) return OMGX_ERR_INVALID;
    const int64_t N = (int64_t)h_dims[0] * h_dims[1] * h_dims[2];
    if (N > (int64_t)1 << 31) return OMGX_ERR_UNSUPPORTED;
    // where the kernels do not cull (an out-of-range lookup's 1.0 still adds something) or the grid has no interior, the loose
    // region stays (scenes.tighten_far_boxes skips these records too)
    if (!(epsilon < 1.0f && clearance <= 1.0f) || h_dims[0] < 2 || h_dims[1] < 2 || h_dims[2] < 2) return OMGX_OK;
    RegionArgs a;
    a.g = grid; a.X = h_dims[0]; a.Y = h_dims[1]; a.Z = h_dims[2];
    a.eps = (double)epsilon; a.clr = (double)clearance;
    for (int k = 0; k < 3; ++k) {
        const float w = h_hi[k] - h_lo[k];
        if (!(w > 0.0f)) return OMGX_OK;
        a.vox[k] = (double)(w / (float)h_dims[k]);
    }
    a.st = reinterpret_cast<RegionStats*>(scratch);
    a.need = reinterpret_cast<unsigned char*>(scratch) + ((sizeof(RegionStats) + 255) & ~(size_t)255);
    a.rec = object;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(k_region_init, dim3(1), dim3(64), 0, st, a.st);
    hipLaunchKernelGGL(k_region_maxabs, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, st, grid, N, a.st);
    hipLaunchKernelGGL(k_region_need, dim3(blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_region_cands, dim3(1), dim3(64), 0, st, a);
    const unsigned wblocks = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(k_region_boxes, dim3(wblocks < 2048 ? wblocks : 2048), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_region_pick, dim3(1), dim3(64), 0, st, a);
    OMGX_CHECK_LAUNCH("omgx_fit_influence_region");
    return OMGX_OK;
}


// Content hash of every object's volume: two independent 64-bit sums of mixed (float bits, position) words — commutative, so the
// lanes add in any order; equal for equal volumes, different otherwise with probability 1 - 2^-128.  DeviceScenes.fit_all fits
// volumes that occur several times (private copies of one model in many scenes) once.
__device__ __forceinline__ unsigned long long rg_mix(unsigned long long z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_volume_hashes(const omgx_object* objects, const float* pool, unsigned long long* out) {
    const omgx_object* r = objects + blockIdx.y;
    const int64_t N = (int64_t)r->dim[0] * r->dim[1] * r->dim[2];
    const uint32_t* g = reinterpret_cast<const uint32_t*>(pool + r->grid_offset);
    unsigned long long h0 = 0ull, h1 = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long w = ((unsigned long long)g[i] << 32) ^ (unsigned long long)i;
        h0 += rg_mix(w + 0x9e3779b97f4a7c15ull);
        h1 += rg_mix(w ^ 0xd1b54a32d192ed03ull);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { h0 += __shfl_xor(h0, off, 64); h1 += __shfl_xor(h1, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out + 2 * blockIdx.y, h0); atomicAdd(out + 2 * blockIdx.y + 1, h1); }
}

extern "C" int omgx_volume_hashes(const omgx_object* objects, int32_t num_objects, const float* pool, uint64_t* hashes, void* stream) {
    if (num_objects < 0) return OMGX_ERR_INVALID;
    if (num_objects == 0) return OMGX_OK;
    if (!objects || !pool || !hashes) return OMGX_ERR_INVALID;
    if (num_objects > 65535) return OMGX_ERR_UNSUPPORTED;
    hipError_t e = hipMemsetAsync(hashes, 0, (size_t)num_objects * 16, (hipStream_t)stream);
    if (e != hipSuccess) return omgx_set_error("hipMemsetAsync", e);
    hipLaunchKernelGGL(k_volume_hashes, dim3(16, (unsigned)num_objects), dim3(256), 0, (hipStream_t)stream, objects, pool,
                       reinterpret_cast<unsigned long long*>(hashes));
    OMGX_CHECK_LAUNCH("k_volume_hashes");
    return OMGX_OK;
}

extern "C" int64_t omgx_regions_scratch_bytes(int32_t n_fit, int64_t need_bytes) {
    if (n_fit < 0 || need_bytes < 0) return 0;
    return (int64_t)n_fit * (int64_t)sizeof(RegionStats) + 256 + need_bytes;
}

extern "C" int omgx_fit_influence_regions(omgx_object* objects, int32_t num_objects, const float* pool, const int32_t* fit_list,
                                          const int64_t* need_offsets, int32_t n_fit, int64_t max_voxels, const int32_t* copy_src,
                                          void* scratch, void* stream) {
    if (n_fit < 0 || num_objects < 0 || max_voxels < 0) return OMGX_ERR_INVALID;
    if (n_fit == 0) return OMGX_OK;
    if (!objects || !pool || !fit_list || !need_offsets || !scratch || max_voxels < 1) return OMGX_ERR_INVALID;
    if (max_voxels > (int64_t)1 << 31 || n_fit > 65535) return OMGX_ERR_UNSUPPORTED;
    RegionBatch b;
    b.objects = objects; b.pool = pool; b.fit_list = fit_list; b.need_off = need_offsets; b.n_fit = n_fit;
    b.stats = reinterpret_cast<RegionStats*>(scratch);
    b.need = reinterpret_cast<unsigned char*>(scratch) + (((size_t)n_fit * sizeof(RegionStats) + 255) & ~(size_t)255);
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((max_voxels + 255) / 256), ny = (unsigned)n_fit;
    hipLaunchKernelGGL(k_regions_init, dim3(1, ny), dim3(64), 0, st, b);
    hipLaunchKernelGGL(k_regions_maxabs, dim3(blocks < 64 ? blocks : 64, ny), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_regions_need, dim3(blocks, ny), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_regions_cands, dim3(1, ny), dim3(64), 0, st, b);
    hipLaunchKernelGGL(k_regions_boxes, dim3(blocks < 256 ? blocks : 256, ny), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_regions_pick, dim3(1, ny), dim3(64), 0, st, b);
    if (copy_src) hipLaunchKernelGGL(k_regions_copy, dim3((unsigned)((num_objects + 255) / 256)), dim3(256), 0, st, objects, copy_src, (int)num_objects);
    OMGX_CHECK_LAUNCH("omgx_fit_influence_regions");
    return OMGX_OK;
}
