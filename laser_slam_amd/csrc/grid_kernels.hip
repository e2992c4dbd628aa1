// Counting-grid update on gfx950: the reference's CPMap
//   CPMap::updateMap          src/drawmap/PMap.cpp:46-91
//   CPMap::mapPosfromPoint    src/drawmap/PMap.cpp:100-117
//   grid_line(_core)          src/drawmap/line.cpp:5-120
//   CPMap::computeProbsOfMap  src/drawmap/PMap.cpp:119-136
//
// Semantics kept bit-exact: integer DDA generated from the
// endpoint with the smaller major-axis coordinate, never reversed; every in-bounds
// cell of the line sum+1, the last generated cell hit+1, the beam's end cell hit+1.
// Counts are u32 (exact; the reference's float counts saturate at 2^24).
//
// This path is HBM/atomic bound: N*(L*8+16) bytes of read-modify-write per scan.
#include "lsa_device.h"

namespace lsa {

struct GridParam {
    int size_x, size_y, center_x, center_y;
    float off_x, off_y;
    double res;
};

// mapPosfromPoint: the point is widened to double, offset and the 0.0005 tail stay float
__device__ __host__ __forceinline__ int grid_cell1(float pf, float off, int center, double res)
{
    const float tail = 0.0005f;
    const double p = (double)pf;
    if (p < (double)off) return (int)((double)center + ((p - (double)off + (double)tail) / res) - 1.0);
    return (int)((double)center + ((p - (double)off + (double)tail) / res));
}

__device__ __forceinline__ bool inb(const GridParam& G, int x, int y)
{
    return x >= 0 && x < G.size_x && y >= 0 && y < G.size_y;
}

// One lane per beam; each lane walks its own Bresenham line.
__global__ __launch_bounds__(256) void grid_update_kernel(
    GridParam G, int S, int n_beams, const float* __restrict__ gx, const float* __restrict__ gy,
    const float* __restrict__ robot_xy, const int* __restrict__ n_valid,
    uint32_t* __restrict__ hit, uint32_t* __restrict__ sum)
{
    const long long total = (long long)S * n_beams;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(t / n_beams), i = (int)(t - (long long)s * n_beams);
        if (n_valid && i >= n_valid[s]) continue;
        const int sx = grid_cell1(robot_xy[2 * s], G.off_x, G.center_x, G.res);
        const int sy = grid_cell1(robot_xy[2 * s + 1], G.off_y, G.center_y, G.res);
        const int ex = grid_cell1(gx[t], G.off_x, G.center_x, G.res);
        const int ey = grid_cell1(gy[t], G.off_y, G.center_y, G.res);
        if (inb(G, ex, ey)) atomicAdd(&hit[(size_t)ex * G.size_y + ey], 1u);          // :72-75

        const int dx = abs(ex - sx), dy = abs(ey - sy);
        int x, y, len, d, inc_keep, inc_step, minor_step;
        bool x_major = dy <= dx;
        if (x_major) {                                                                 // line.cpp:17-58
            d = 2 * dy - dx; inc_keep = 2 * dy; inc_step = 2 * (dy - dx);
            int flag;
            if (sx > ex) { x = ex; y = ey; flag = -1; } else { x = sx; y = sy; flag = 1; }
            minor_step = ((ey - sy) * flag) > 0 ? 1 : -1;
            len = dx;
        } else {                                                                       // line.cpp:59-101
            d = 2 * dx - dy; inc_keep = 2 * dx; inc_step = 2 * (dx - dy);
            int flag;
            if (sy > ey) { y = ey; x = ex; flag = -1; } else { y = sy; x = sx; flag = 1; }
            minor_step = ((ex - sx) * flag) > 0 ? 1 : -1;
            len = dy;
        }
        for (int k = 0; k <= len; ++k) {
            if (k > 0) {
                if (x_major) { x++; if (d < 0) d += inc_keep; else { y += minor_step; d += inc_step; } }
                else         { y++; if (d < 0) d += inc_keep; else { x += minor_step; d += inc_step; } }
            }
            if (inb(G, x, y)) {
                const size_t c = (size_t)x * G.size_y + y;
                if (k == len) atomicAdd(&hit[c], 1u);                                  // j > cnt-2, :84-86
                atomicAdd(&sum[c], 1u);                                                // :87
            }
        }
    }
}

// Same update, one workgroup per scan with the robot's neighbourhood staged in LDS.
// Every ray of a scan starts in the robot cell, so the cells around it are hit by hundreds of
// rays: their increments are accumulated in an LDS tile (packed u16 hit | u16 visit, a scan
// contributes < 65536 to either) and flushed once with row-contiguous atomics; only the part
// of a ray that leaves the tile goes to global atomics directly.  Counts are additive, so the
// totals are identical to the per-cell version.
constexpr int GRID_TILE = 128;          // cells per tile side (64 KiB of LDS)

__global__ __launch_bounds__(256) void grid_update_tile_kernel(
    GridParam G, int S, int n_beams, const float* __restrict__ gx, const float* __restrict__ gy,
    const float* __restrict__ robot_xy, const int* __restrict__ n_valid,
    uint32_t* __restrict__ hit, uint32_t* __restrict__ sum)
{
    __shared__ uint32_t tile[GRID_TILE * GRID_TILE];
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        for (int k = threadIdx.x; k < GRID_TILE * GRID_TILE; k += blockDim.x) tile[k] = 0;
        const int sx = grid_cell1(robot_xy[2 * s], G.off_x, G.center_x, G.res);
        const int sy = grid_cell1(robot_xy[2 * s + 1], G.off_y, G.center_y, G.res);
        const int ox = sx - GRID_TILE / 2, oy = sy - GRID_TILE / 2;
        const int nb = n_valid ? n_valid[s] : n_beams;
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += blockDim.x) {
            const size_t t = (size_t)s * n_beams + i;
            const int ex = grid_cell1(gx[t], G.off_x, G.center_x, G.res);
            const int ey = grid_cell1(gy[t], G.off_y, G.center_y, G.res);
            if (inb(G, ex, ey)) {                                                     // :72-75
                const unsigned tx = (unsigned)(ex - ox), ty = (unsigned)(ey - oy);
                if (tx < GRID_TILE && ty < GRID_TILE) atomicAdd(&tile[tx * GRID_TILE + ty], 1u << 16);
                else atomicAdd(&hit[(size_t)ex * G.size_y + ey], 1u);
            }
            const int dx = abs(ex - sx), dy = abs(ey - sy);
            int x, y, len, d, inc_keep, inc_step, minor_step;
            const bool x_major = dy <= dx;
            if (x_major) {                                                            // line.cpp:17-58
                d = 2 * dy - dx; inc_keep = 2 * dy; inc_step = 2 * (dy - dx);
                int flag;
                if (sx > ex) { x = ex; y = ey; flag = -1; } else { x = sx; y = sy; flag = 1; }
                minor_step = ((ey - sy) * flag) > 0 ? 1 : -1;
                len = dx;
            } else {                                                                  // line.cpp:59-101
                d = 2 * dx - dy; inc_keep = 2 * dx; inc_step = 2 * (dx - dy);
                int flag;
                if (sy > ey) { y = ey; x = ex; flag = -1; } else { y = sy; x = sx; flag = 1; }
                minor_step = ((ex - sx) * flag) > 0 ? 1 : -1;
                len = dy;
            }
            for (int k = 0; k <= len; ++k) {
                if (k > 0) {
                    if (x_major) { x++; if (d < 0) d += inc_keep; else { y += minor_step; d += inc_step; } }
                    else         { y++; if (d < 0) d += inc_keep; else { x += minor_step; d += inc_step; } }
                }
                if (inb(G, x, y)) {
                    const unsigned tx = (unsigned)(x - ox), ty = (unsigned)(y - oy);
                    const uint32_t inc = (k == len ? (1u << 16) : 0u) + 1u;           // last cell: hit too (:84-87)
                    if (tx < GRID_TILE && ty < GRID_TILE) atomicAdd(&tile[tx * GRID_TILE + ty], inc);
                    else {
                        const size_t c = (size_t)x * G.size_y + y;
                        if (k == len) atomicAdd(&hit[c], 1u);
                        atomicAdd(&sum[c], 1u);
                    }
                }
            }
        }
        __syncthreads();
        // flush: consecutive lanes -> consecutive y of one x row (contiguous in the [x][y] layers)
        for (int k = threadIdx.x; k < GRID_TILE * GRID_TILE; k += blockDim.x) {
            const uint32_t v = tile[k];
            if (v) {
                const int x = ox + k / GRID_TILE, y = oy + k % GRID_TILE;      // in bounds: only in-bounds cells were counted
                const size_t c = (size_t)x * G.size_y + y;
                if (v & 0xffffu) atomicAdd(&sum[c], v & 0xffffu);
                if (v >> 16) atomicAdd(&hit[c], v >> 16);
            }
        }
        __syncthreads();
    }
}

// computeProbsOfMap: prob = hit / (double) sum clipped to 1, unseen cells 0.3.
// Counts are converted to float first (the reference stores them as float).
__global__ __launch_bounds__(256) void grid_prob_kernel(
    size_t cells, const uint32_t* __restrict__ hit, const uint32_t* __restrict__ sum, float* __restrict__ prob)
{
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x) {
        const float s = (float)sum[c], h = (float)hit[c];
        float p;
        if (s > 0) {
            p = (float)((double)h / (double)s);
            if ((double)p > 1.0) p = 1.0f;
        } else {
            p = (float)0.3;
        }
        prob[c] = p;
    }
}

// u32 count layer -> float layer (download in the reference's storage type)
__global__ __launch_bounds__(256) void grid_to_float_kernel(size_t cells, const uint32_t* __restrict__ in,
                                                            float* __restrict__ out)
{
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x)
        out[c] = (float)in[c];
}

// getPointCloud (:138-149): cells with prob >= thre in [x][y] order.  Two passes:
// per-row counts, host-free exclusive scan by one block, ordered scatter.
__global__ __launch_bounds__(256) void cloud_count_kernel(GridParam G, const float* __restrict__ prob, float thre,
                                                          int* __restrict__ row_cnt)
{
    // one wave per row x
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, lane = lane_id();
    if (wave >= G.size_x) return;
    int cnt = 0;
    for (int y = lane; y < G.size_y; y += WAVE) cnt += prob[(size_t)wave * G.size_y + y] >= thre ? 1 : 0;
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) row_cnt[wave] = cnt;
}

__global__ __launch_bounds__(1024) void cloud_scan_kernel(int n, int* __restrict__ row_cnt, int* __restrict__ total)
{
    // single block exclusive scan (n = size_x, a few thousand)
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    int s = 0;
    for (int k = 0; k < per; ++k) { int i = tid * per + k; if (i < n) s += row_cnt[i]; }
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int k = 0; k < per; ++k) {
        int i = tid * per + k;
        if (i < n) { int c = row_cnt[i]; row_cnt[i] = run; run += c; }
    }
    if (tid == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void cloud_emit_kernel(GridParam G, const float* __restrict__ prob, float thre,
                                                         const int* __restrict__ row_off, int cap,
                                                         float* __restrict__ ox, float* __restrict__ oy)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, lane = lane_id();
    if (wave >= G.size_x) return;
    int base = row_off[wave];
    for (int y0 = 0; y0 < G.size_y; y0 += WAVE) {
        const int y = y0 + lane;
        const bool on = y < G.size_y && prob[(size_t)wave * G.size_y + y] >= thre;
        const unsigned long long m = __ballot(on);
        if (on) {
            const int pos = base + prefix_popc(m);
            if (pos < cap) {
                ox[pos] = (float)((double)(wave - G.center_x) * G.res);
                oy[pos] = (float)((double)(y - G.center_y) * G.res);
            }
        }
        base += __popcll(m);
    }
}

} // namespace lsa
