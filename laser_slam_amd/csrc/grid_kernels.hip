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
// (Cell numbers are kept within +-2^28: a point that far outside any map - or a NaN - has no defined cell in the
//  reference either, and the walks' integer arithmetic must not overflow on it.)
__device__ __host__ __forceinline__ int grid_cell1(float pf, float off, int center, double res)
{
    const float tail = 0.0005f;
    const double p = (double)pf;
    double c = (double)center + ((p - (double)off + (double)tail) / res);
    if (p < (double)off) c -= 1.0;
    if (!(c > -268435456.0)) c = -268435456.0;
    if (!(c < 268435456.0)) c = 268435456.0;
    return (int)c;
}

__device__ __forceinline__ bool inb(const GridParam& G, int x, int y)
{
    return x >= 0 && x < G.size_x && y >= 0 && y < G.size_y;
}

// The reference's line from (sx, sy) to (ex, ey) (line.cpp:17-101: generated from the end with the smaller major
// coordinate), entered at the first cell whose MAJOR coordinate lies inside the map and left after the last such
// cell - a beam that ends (or a robot that stands) millions of cells outside the map costs no more than one that
// crosses it.  A standard integer DDA (d0 = 2 dmin - len; a minor step whenever d >= 0) has taken
// floor((2 dmin t + len) / (2 len)) minor steps before its cell t, which gives the exact state at any t.
struct LineWalk {
    bool x_major;
    int major, minor, mstep, d, inc_keep, inc_step, k, k_hi, len;
    __device__ __forceinline__ int x() const { return x_major ? major : minor; }
    __device__ __forceinline__ int y() const { return x_major ? minor : major; }
    __device__ __forceinline__ void step() { ++k; ++major; if (d < 0) d += inc_keep; else { minor += mstep; d += inc_step; } }
};

__device__ __forceinline__ LineWalk line_begin(const GridParam& G, int sx, int sy, int ex, int ey)
{
    LineWalk w;
    const int dx = abs(ex - sx), dy = abs(ey - sy);
    w.x_major = dy <= dx;
    const bool from_end = w.x_major ? sx > ex : sy > ey;
    const int len = w.x_major ? dx : dy, dmin = w.x_major ? dy : dx;
    const int m0 = w.x_major ? (from_end ? ex : sx) : (from_end ? ey : sy);
    const int n0 = w.x_major ? (from_end ? ey : sy) : (from_end ? ex : sx);
    w.mstep = ((w.x_major ? ey - sy : ex - sx) * (from_end ? -1 : 1)) > 0 ? 1 : -1;
    w.len = len; w.inc_keep = 2 * dmin; w.inc_step = 2 * (dmin - len);
    const int size_major = w.x_major ? G.size_x : G.size_y;
    const int t_lo = m0 < 0 ? -m0 : 0;
    w.k_hi = size_major - 1 - m0 < len ? size_major - 1 - m0 : len;
    w.k = t_lo;
    int steps = 0;
    if (t_lo > 0 && len > 0) steps = (int)((2ll * dmin * t_lo + len) / (2ll * len));
    w.major = m0 + t_lo;
    w.minor = n0 + w.mstep * steps;
    w.d = (int)(2ll * dmin * (t_lo + 1) - len - 2ll * len * steps);
    return w;
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

        for (LineWalk w = line_begin(G, sx, sy, ex, ey); w.k <= w.k_hi; w.step()) {
            const int x = w.x(), y = w.y();
            if (inb(G, x, y)) {
                const size_t c = (size_t)x * G.size_y + y;
                if (w.k == w.len) atomicAdd(&hit[c], 1u);                              // j > cnt-2, :84-86
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
            for (LineWalk w = line_begin(G, sx, sy, ex, ey); w.k <= w.k_hi; w.step()) {
                const int x = w.x(), y = w.y();
                if (inb(G, x, y)) {
                    const unsigned tx = (unsigned)(x - ox), ty = (unsigned)(y - oy);
                    const uint32_t inc = (w.k == w.len ? (1u << 16) : 0u) + 1u;       // last cell: hit too (:84-87)
                    if (tx < GRID_TILE && ty < GRID_TILE) atomicAdd(&tile[tx * GRID_TILE + ty], inc);
                    else {
                        const size_t c = (size_t)x * G.size_y + y;
                        if (w.k == w.len) atomicAdd(&hit[c], 1u);
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

// ---------------------------------------------------------------------------------------------
// Batched update with tile ownership (the throughput path): no global atomics at all.
//
//   grid_prep_kernel   one workgroup per scan: robot cell, end cell of every beam (stored as
//                      int2, the double-precision cell arithmetic is done once), the scan's
//                      cell bounding box and the batch's bounding box.
//   grid_owned_kernel  one workgroup per 64x64-cell tile of the grid: the tile's two count
//                      layers live in LDS for the whole batch; the workgroup walks through the
//                      scans whose bounding box meets the tile and, for every ray that crosses
//                      it, exactly the part of the reference's Bresenham line inside the tile.
//                      A standard integer DDA (d0 = 2 dmin - len; a minor step whenever d >= 0)
//                      has taken floor((2 dmin t + len) / (2 len)) minor steps after t major
//                      steps, so the walk can start anywhere on the line with the reference's
//                      exact state.  At the end the tile is added to the layers with plain,
//                      row-contiguous read-modify-writes - the tile is nobody else's.
//                      All rays of a scan meet in the robot cell (hundreds of increments of one
//                      address): that cell is skipped by the walks and receives its totals once
//                      per wave.  With `splits` > 1 the scans are divided among several
//                      workgroups per tile (more parallelism when the batch covers few tiles)
//                      and the flush uses atomic adds instead.
// Counts are additive and integer, so the result equals the sequential update bit for bit.
struct ScanInfo { int sx, sy, xmin, xmax, ymin, ymax, nb, nh, s, pad; };   // nh: rays whose last generated cell is the robot cell

constexpr int OWN_T = 64;               // cells per owned tile side (2 x 16 KiB of LDS)
constexpr int OWN_CHUNK = 256;          // scans looked at per culling round (fewer when a scan has many blocks)
constexpr int OWN_LIST = 256 * 17;      // candidate blocks per round

// Binned path, shared by the counting pass (grid_prep_kernel) and the fill pass (grid_bin_fill_kernel): which
// 64x64 tiles do the 64-beam blocks of one scan concern?  A workgroup owns the scan.  When the scan's tile
// rectangle is small enough, the blocks' boxes are first merged in LDS into one 64-bit block mask per tile, so
// that the layers' global counters see one atomic per (scan, tile) instead of one per (block, tile) - the hot
// tiles around the robot are otherwise hit by every block of every scan.
constexpr int BIN_T = 64;                // cells per tile side
constexpr int BIN_RECT_MAX = 1024;       // tiles of a scan's rectangle held as LDS masks

__device__ __forceinline__ bool bin_tile_rect(const GridParam& G, int4 bb, int& tx0, int& tx1, int& ty0, int& ty1)
{
    if (bb.y < 0 || bb.w < 0 || bb.x >= G.size_x || bb.z >= G.size_y) return false;
    tx0 = (bb.x > 0 ? bb.x : 0) / BIN_T; tx1 = (bb.y < G.size_x - 1 ? bb.y : G.size_x - 1) / BIN_T;
    ty0 = (bb.z > 0 ? bb.z : 0) / BIN_T; ty1 = (bb.w < G.size_y - 1 ? bb.w : G.size_y - 1) / BIN_T;
    return true;
}

constexpr int BIN_GROUP = 8;             // scans binned by one workgroup

// One workgroup bins BIN_GROUP consecutive scans.  FILL = false: tile_ctr[t] += the blocks of these scans whose
// box meets tile t.  FILL = true: the same blocks are appended to the tiles' lists (tile_ctr = fill cursors).
// The blocks' boxes are merged in LDS - per scan into a 64-bit block mask per tile, then into a per-tile count of
// the whole group - so that a tile's global counter sees one atomic per group and tile, not one per block.
template <bool FILL>
__global__ __launch_bounds__(256) void grid_bin_tiles_kernel(GridParam G, int S, int n_beams, const ScanInfo* __restrict__ info,
                                                             const int4* __restrict__ blk, int nty_all, int* __restrict__ tile_ctr,
                                                             const int* __restrict__ tile_off, uint32_t* __restrict__ list)
{
    __shared__ int4 boxes[BIN_GROUP][64];
    __shared__ int4 sbox[BIN_GROUP];
    __shared__ int snb[BIN_GROUP];
    __shared__ unsigned long long mask[BIN_RECT_MAX];
    __shared__ int cnt[BIN_RECT_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nblk = (n_beams + 63) >> 6;
    for (int g0 = blockIdx.x * BIN_GROUP; g0 < S; g0 += gridDim.x * BIN_GROUP) {
        const int ng = S - g0 < BIN_GROUP ? S - g0 : BIN_GROUP;
        __syncthreads();
        if (nblk <= 64) for (int k = tid; k < ng * nblk; k += 256) boxes[k / nblk][k % nblk] = blk[(size_t)g0 * nblk + k];
        if (tid < ng) {
            const ScanInfo si = info[g0 + tid];
            sbox[tid] = make_int4(si.xmin, si.xmax, si.ymin, si.ymax);
            snb[tid] = (si.nb + 63) >> 6 < nblk ? (si.nb + 63) >> 6 : nblk;      // blocks that hold rays
        }
        __syncthreads();
        // the group's tile rectangle
        int4 ub = make_int4(INT_MAX, INT_MIN, INT_MAX, INT_MIN);
        for (int g = 0; g < ng; ++g) if (snb[g] > 0) {
            ub.x = sbox[g].x < ub.x ? sbox[g].x : ub.x; ub.y = sbox[g].y > ub.y ? sbox[g].y : ub.y;
            ub.z = sbox[g].z < ub.z ? sbox[g].z : ub.z; ub.w = sbox[g].w > ub.w ? sbox[g].w : ub.w;
        }
        int sx0, sx1, sy0, sy1;
        if (ub.x > ub.y || !bin_tile_rect(G, ub, sx0, sx1, sy0, sy1)) continue;
        const int swy = sy1 - sy0 + 1, area = (sx1 - sx0 + 1) * swy;
        if (area <= BIN_RECT_MAX && nblk <= 64) {
            for (int k = tid; k < area; k += 256) cnt[k] = 0;
            for (int pass = 0; pass < (FILL ? 2 : 1); ++pass) {
                for (int g = 0; g < ng; ++g) {
                    __syncthreads();
                    for (int k = tid; k < area; k += 256) mask[k] = 0ull;
                    __syncthreads();
                    for (int bk = wave; bk < snb[g]; bk += 4) {
                        int tx0, tx1, ty0, ty1;
                        if (!bin_tile_rect(G, boxes[g][bk], tx0, tx1, ty0, ty1)) continue;
                        const int wy = ty1 - ty0 + 1, nrect = (tx1 - tx0 + 1) * wy;
                        for (int k = lane; k < nrect; k += 64)
                            atomicOr(&mask[(tx0 + k / wy - sx0) * swy + (ty0 + k % wy - sy0)], 1ull << bk);
                    }
                    __syncthreads();
                    for (int k = tid; k < area; k += 256) {              // (tile k belongs to one thread throughout)
                        unsigned long long m = mask[k];
                        if (!m) continue;
                        if (pass == 0) cnt[k] += __popcll(m);
                        else {
                            int pos = cnt[k];
                            cnt[k] = pos + __popcll(m);
                            while (m) { const int bk = __ffsll((long long)m) - 1; m &= m - 1; list[pos++] = ((uint32_t)(g0 + g) << 8) | (uint32_t)bk; }
                        }
                    }
                }
                if (pass == 0) {
                    for (int k = tid; k < area; k += 256) {
                        const int c = cnt[k];
                        if (!c) continue;
                        const int t = (sx0 + k / swy) * nty_all + sy0 + k % swy;
                        if (!FILL) atomicAdd(&tile_ctr[t], c);
                        else cnt[k] = tile_off[t] + atomicAdd(&tile_ctr[t], c);  // the group's slots in tile t's list
                    }
                }
            }
        } else {                                    // a huge rectangle (or more than 64 blocks per scan): block by block
            for (int g = 0; g < ng; ++g)
                for (int bk = wave; bk < snb[g]; bk += 4) {
                    int tx0, tx1, ty0, ty1;
                    const int4 bb = nblk <= 64 ? boxes[g][bk] : blk[(size_t)(g0 + g) * nblk + bk];
                    if (!bin_tile_rect(G, bb, tx0, tx1, ty0, ty1)) continue;
                    const int wy = ty1 - ty0 + 1, nrect = (tx1 - tx0 + 1) * wy;
                    for (int k = lane; k < nrect; k += 64) {
                        const int t = (tx0 + k / wy) * nty_all + ty0 + k % wy;
                        if (!FILL) atomicAdd(&tile_ctr[t], 1);
                        else list[tile_off[t] + atomicAdd(&tile_ctr[t], 1)] = ((uint32_t)(g0 + g) << 8) | (uint32_t)bk;
                    }
                }
        }
    }
}

__global__ __launch_bounds__(256) void grid_prep_kernel(
    GridParam G, int S, int n_beams, const float* __restrict__ gx, const float* __restrict__ gy,
    const float* __restrict__ robot_xy, const int* __restrict__ n_valid,
    int2* __restrict__ ends, ScanInfo* __restrict__ info, int4* __restrict__ blk /* [S][n_beams/64]: xmin,xmax,ymin,ymax */,
    int* __restrict__ bbox /* xmin, ymin, xmax, ymax (owned-tile path; may be null) */,
    uint32_t* __restrict__ hit = nullptr, uint32_t* __restrict__ sum = nullptr /* binned path: robot-cell totals */)
{
    __shared__ int red[4][5];
    const int nblk = (n_beams + 63) >> 6;
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        const int sx = grid_cell1(robot_xy[2 * s], G.off_x, G.center_x, G.res);
        const int sy = grid_cell1(robot_xy[2 * s + 1], G.off_y, G.center_y, G.res);
        const int nb = n_valid ? (n_valid[s] < n_beams ? n_valid[s] : n_beams) : n_beams;
        int x0 = sx, x1 = sx, y0 = sy, y1 = sy, nh = 0;
        for (int i = threadIdx.x; i < nblk * 64; i += blockDim.x) {
            const size_t t = (size_t)s * n_beams + i;
            int ex = sx, ey = sy;
            if (i < nb) {
                ex = grid_cell1(gx[t], G.off_x, G.center_x, G.res);
                ey = grid_cell1(gy[t], G.off_y, G.center_y, G.res);
                ends[t] = make_int2(ex, ey);
            }
            {   // bounding box of this wave's 64 consecutive rays (all of them start in the robot cell)
                int a0 = ex < sx ? ex : sx, a1 = ex > sx ? ex : sx, c0 = ey < sy ? ey : sy, c1 = ey > sy ? ey : sy;
                for (int o = 32; o; o >>= 1) {
                    const int p0 = __shfl_xor(a0, o), p1 = __shfl_xor(a1, o), q0 = __shfl_xor(c0, o), q1 = __shfl_xor(c1, o);
                    a0 = p0 < a0 ? p0 : a0; a1 = p1 > a1 ? p1 : a1; c0 = q0 < c0 ? q0 : c0; c1 = q1 > c1 ? q1 : c1;
                }
                if ((threadIdx.x & 63) == 0) blk[(size_t)s * nblk + (i >> 6)] = make_int4(a0, a1, c0, c1);
            }
            if (i >= nb) continue;
            x0 = ex < x0 ? ex : x0; x1 = ex > x1 ? ex : x1;
            y0 = ey < y0 ? ey : y0; y1 = ey > y1 ? ey : y1;
            // the robot cell is the last generated cell when the line comes from the beam's end
            // (the end has the smaller major coordinate, line.cpp:17-27, 61-70) or is a single cell
            const int dx = abs(ex - sx), dy = abs(ey - sy);
            const bool from_end = dy <= dx ? sx > ex : sy > ey;
            nh += (from_end || (dx == 0 && dy == 0)) ? 1 : 0;
        }
        for (int o = 32; o; o >>= 1) {
            const int a = __shfl_xor(x0, o), b = __shfl_xor(x1, o), c = __shfl_xor(y0, o), d = __shfl_xor(y1, o);
            x0 = a < x0 ? a : x0; x1 = b > x1 ? b : x1; y0 = c < y0 ? c : y0; y1 = d > y1 ? d : y1;
            nh += __shfl_xor(nh, o);
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { int w = threadIdx.x >> 6; red[w][0] = x0; red[w][1] = x1; red[w][2] = y0; red[w][3] = y1; red[w][4] = nh; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) {
                x0 = red[w][0] < x0 ? red[w][0] : x0; x1 = red[w][1] > x1 ? red[w][1] : x1;
                y0 = red[w][2] < y0 ? red[w][2] : y0; y1 = red[w][3] > y1 ? red[w][3] : y1;
                nh += red[w][4];
            }
            ScanInfo si; si.sx = sx; si.sy = sy; si.xmin = x0; si.xmax = x1; si.ymin = y0; si.ymax = y1;
            si.nb = nb > 0 ? nb : 0; si.nh = nh; si.s = s; si.pad = 0;
            info[s] = si;
            if (sum && inb(G, sx, sy)) {                    // every ray of the scan starts in the robot cell; the walks leave it out
                const size_t c = (size_t)sx * G.size_y + sy;
                if (si.nb) atomicAdd(&sum[c], (uint32_t)si.nb);
                if (nh) atomicAdd(&hit[c], (uint32_t)nh);
            }
            if (bbox) { atomicMin(&bbox[0], x0); atomicMin(&bbox[1], y0); atomicMax(&bbox[2], x1); atomicMax(&bbox[3], y1); }
        }
    }
}

// The part of one ray inside the tile [X0, X0+T) x [Y0, Y0+T); the robot cell (sx, sy) is left out when `rin`.
// `inside`: the whole tile lies in the grid (no per-cell bounds test).  (Neighbouring beams add to the
// same LDS cells; starting every lane's walk at a different place of its range was measured: the
// second closed-form start costs more than the serialised adds.)
__device__ __forceinline__ void own_walk(const GridParam& G, uint32_t* lh, uint32_t* ls, int X0, int Y0, bool inside,
                                         int sx, int sy, bool rin, int ex, int ey)
{
    const int X1 = X0 + OWN_T - 1, Y1 = Y0 + OWN_T - 1;
    if (ex >= X0 && ex <= X1 && ey >= Y0 && ey <= Y1 && inb(G, ex, ey))                  // :72-75
        atomicAdd(&lh[(ex - X0) * OWN_T + (ey - Y0)], 1u);
    const int dx = abs(ex - sx), dy = abs(ey - sy);
    const bool x_major = dy <= dx;
    const bool from_end = x_major ? sx > ex : sy > ey;
    const int len = x_major ? dx : dy, dmin = x_major ? dy : dx;
    const int m0 = x_major ? (from_end ? ex : sx) : (from_end ? ey : sy);                  // major / minor of the first cell
    const int n0 = x_major ? (from_end ? ey : sy) : (from_end ? ex : sx);
    const int mstep = ((x_major ? ey - sy : ex - sx) * (from_end ? -1 : 1)) > 0 ? 1 : -1;
    const int M0 = x_major ? X0 : Y0, N0 = x_major ? Y0 : X0;
    const int t_lo = M0 - m0 > 0 ? M0 - m0 : 0;
    const int t_hi = M0 + OWN_T - 1 - m0 < len ? M0 + OWN_T - 1 - m0 : len;
    if (t_lo > t_hi) return;
    const int robot_cc = rin ? (sx - X0) * OWN_T + (sy - Y0) : -1;
    const int c_major = x_major ? OWN_T : 1, c_minor = x_major ? mstep : mstep * OWN_T;   // cell index steps
    const int inc_keep = 2 * dmin, inc_step = 2 * (dmin - len);
    int t = t_lo;
    int steps = 0;                                                                       // minor steps taken before cell t
    if (len > 0) {
        if (len < 32768) steps = (int)((2u * (unsigned)dmin * (unsigned)t + (unsigned)len) / (2u * (unsigned)len));   // t <= len
        else steps = (int)((2ll * dmin * t + len) / (2ll * len));
    }
    int minor = n0 + mstep * steps;
    int d = (int)(2ll * dmin * (t + 1) - len - 2ll * len * steps);                      // |d| <= 2 len
    int cc = x_major ? (m0 + t - X0) * OWN_T + (minor - Y0) : (minor - X0) * OWN_T + (m0 + t - Y0);
    for (; t <= t_hi; ++t) {
        bool ok = (unsigned)(minor - N0) < (unsigned)OWN_T && cc != robot_cc;
        if (!inside) ok = ok && inb(G, x_major ? m0 + t : minor, x_major ? minor : m0 + t);
        if (ok) {
            atomicAdd(&ls[cc], 1u);                                                      // :87
            if (t == len) atomicAdd(&lh[cc], 1u);                                        // last generated cell, :84-86
        }
        cc += c_major;
        if (d < 0) d += inc_keep; else { minor += mstep; cc += c_minor; d += inc_step; }
    }
}

#ifdef LSA_GRID_STAMPS       // diagnostic build only: workgroup cycles per phase (thread 0), items, longest item, cells walked
__device__ unsigned long long lsa_grid_cycles[16];
#define GSTAMP(slot) do { if (threadIdx.x == 0) { unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&lsa_grid_cycles[slot], t_ - gt_); gt_ = t_; } } while (0)
#else
#define GSTAMP(slot) do {} while (0)
#endif

__global__ __launch_bounds__(256) void grid_owned_kernel(
    GridParam G, int S, int n_beams, const int2* __restrict__ ends, const ScanInfo* __restrict__ info,
    const int4* __restrict__ blk, int* __restrict__ bbox /* [0..3] box, [4] item counter */, int splits_req,
    uint32_t* __restrict__ hit, uint32_t* __restrict__ sum)
{
    __shared__ uint32_t lh[OWN_T * OWN_T], ls[OWN_T * OWN_T];
    __shared__ ScanInfo cand[OWN_CHUNK];
    __shared__ uint16_t list[OWN_LIST];             // (scan of the round << 8) | block
    __shared__ int nlist;
    const int tid = threadIdx.x, lane = tid & 63;
    // work items = (tile of the batch's bounding box, share of the scans); the launch is a fixed number of
    // workgroups that stride over them (launching one workgroup per tile of a 4096^2 grid costs more
    // than the update itself when the batch covers a few dozen tiles)
    const int bx0 = (bbox[0] > 0 ? bbox[0] : 0) / OWN_T, by0 = (bbox[1] > 0 ? bbox[1] : 0) / OWN_T;
    const int bx1 = (bbox[2] < G.size_x - 1 ? bbox[2] : G.size_x - 1) / OWN_T, by1 = (bbox[3] < G.size_y - 1 ? bbox[3] : G.size_y - 1) / OWN_T;
    if (bbox[2] < 0 || bbox[3] < 0 || bbox[0] >= G.size_x || bbox[1] >= G.size_y || bx1 < bx0 || by1 < by0) return;
    const int ntx = bx1 - bx0 + 1, nty = by1 - by0 + 1, nt = ntx * nty;
    int splits = splits_req;
    if (splits <= 0) {                      // about 16 scans per item (measured: 2048 scans, 128 shares per tile)
        splits = S / 16;
        if (splits > 128) splits = 128;
        if (splits < 1) splits = 1;
    }
    if (splits > S) splits = S;
    // items are claimed from a counter (bbox[4], zeroed with the box) so that a workgroup that drew a
    // light tile moves on: the work per item differs by orders of magnitude
    __shared__ int next_item;
#ifdef LSA_GRID_STAMPS
    unsigned long long gt_ = __builtin_readcyclecounter(), gstart_ = gt_, gitem_ = gt_;
#endif
    for (;;) {
    __syncthreads();
#ifdef LSA_GRID_STAMPS
    if (tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicMax(&lsa_grid_cycles[6], t_ - gitem_); gitem_ = t_; }
#endif
    if (tid == 0) next_item = atomicAdd((int*)&bbox[4], 1);
    __syncthreads();
    const int item = next_item;
    if (item >= nt * splits) break;
    GSTAMP(4);
#ifdef LSA_GRID_STAMPS
    if (tid == 0) atomicAdd(&lsa_grid_cycles[5], 1ull);
#endif
    const int tile = item % nt, part = item / nt;
    const int X0 = (bx0 + tile / nty) * OWN_T, Y0 = (by0 + tile % nty) * OWN_T;
    const int X1 = X0 + OWN_T - 1, Y1 = Y0 + OWN_T - 1;
    const bool inside = X1 < G.size_x && Y1 < G.size_y;          // (X0, Y0 >= 0 by construction)
    __syncthreads();
    for (int k = tid; k < OWN_T * OWN_T; k += blockDim.x) { lh[k] = 0; ls[k] = 0; }
    const int s_lo = (int)((long long)S * part / splits), s_hi = (int)((long long)S * (part + 1) / splits);
    bool touched = false;
    const int wave = tid >> 6, nblk = (n_beams + 63) >> 6;
    const int chunk = OWN_LIST / nblk < OWN_CHUNK ? OWN_LIST / nblk : OWN_CHUNK;     // scans per round: the list holds all their blocks
    for (int c0 = s_lo; c0 < s_hi; c0 += chunk) {
        const int ns = s_hi - c0 < chunk ? s_hi - c0 : chunk;
        __syncthreads();
        if (tid == 0) nlist = 0;
        // the round's scans: header into LDS; all rays of a scan meet in its robot cell - that cell gets
        // its totals here and the walks skip it
        if (tid < ns) {
            const ScanInfo si = info[c0 + tid];
            cand[tid] = si;
            if (si.sx >= X0 && si.sx <= X1 && si.sy >= Y0 && si.sy <= Y1 && inb(G, si.sx, si.sy)) {
                const int cc = (si.sx - X0) * OWN_T + (si.sy - Y0);
                if (si.nb) atomicAdd(&ls[cc], (uint32_t)si.nb);
                if (si.nh) atomicAdd(&lh[cc], (uint32_t)si.nh);
            }
        }
        __syncthreads();
        // culling: every (scan, 64-beam block) whose bounding box (grid_prep_kernel) meets the tile goes to a list
        for (int pr = tid; pr < ((ns * nblk + 63) & ~63); pr += blockDim.x) {
            bool ov = false;
            if (pr < ns * nblk) {
                const int4 bb = blk[(size_t)c0 * nblk + pr];               // the round's boxes are contiguous
                ov = !(bb.y < X0 || bb.x > X1 || bb.w < Y0 || bb.z > Y1);
            }
            const unsigned long long m = __ballot(ov);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(&nlist, __popcll(m));
            base = __shfl(base, 0);
            if (ov) list[base + prefix_popc(m)] = (uint16_t)(((pr / nblk) << 8) | (pr % nblk));
        }
        __syncthreads();
        GSTAMP(1);
        const int nl = nlist;
        if (nl) touched = true;
#ifdef LSA_GRID_STAMPS
        if (tid == 0) atomicAdd(&lsa_grid_cycles[8], (unsigned long long)nl);
#endif
        // every wave walks listed blocks, 64 rays at a time; the next block's end cells are in flight meanwhile
        int q = wave;
        int2 e = make_int2(0, 0);
        if (q < nl) {
            const int j = list[q] >> 8, bk = list[q] & 255;
            if (64 * bk + lane < cand[j].nb) e = ends[(size_t)cand[j].s * n_beams + 64 * bk + lane];
        }
        for (; q < nl; q += 4) {
            const int j = list[q] >> 8, bk = list[q] & 255;
            int2 en = make_int2(0, 0);
            if (q + 4 < nl) {
                const int jn = list[q + 4] >> 8, bn = list[q + 4] & 255;
                if (64 * bn + lane < cand[jn].nb) en = ends[(size_t)cand[jn].s * n_beams + 64 * bn + lane];
            }
            const int sx = cand[j].sx, sy = cand[j].sy;
            const bool rin = sx >= X0 && sx <= X1 && sy >= Y0 && sy <= Y1;
            if (64 * bk + lane < cand[j].nb) own_walk(G, lh, ls, X0, Y0, inside, sx, sy, rin, e.x, e.y);
            e = en;
        }
        __syncthreads();
        GSTAMP(2);
    }
    __syncthreads();
    GSTAMP(0);
    if (!touched) continue;
    // flush: consecutive lanes -> consecutive y of one x row (contiguous in the [x][y] layers)
    for (int k = tid; k < OWN_T * OWN_T; k += blockDim.x) {
        const int x = X0 + k / OWN_T, y = Y0 + k % OWN_T;
        if (!inb(G, x, y)) continue;
        const uint32_t vh = lh[k], vs = ls[k];
        const size_t c = (size_t)x * G.size_y + y;
        if (splits == 1) { if (vh) hit[c] += vh; if (vs) sum[c] += vs; }
        else { if (vh) atomicAdd(&hit[c], vh); if (vs) atomicAdd(&sum[c], vs); }
    }
    __syncthreads();
    GSTAMP(3);
    }
#ifdef LSA_GRID_STAMPS
    if (tid == 0) { atomicAdd(&lsa_grid_cycles[7], __builtin_readcyclecounter() - gstart_); atomicAdd(&lsa_grid_cycles[9], 1ull); }
#endif
}

// Batched update, binned (the throughput path).  The owned-tile kernel above spends most of its time finding out
// which rays concern a tile; here that question is answered once, rasteriser style:
//   grid_prep_kernel      (above) end cells, per-scan header, the bounding box of every 64-beam block; adds the
//                         robot cell's totals (every ray of a scan starts there) to the layers.
//   grid_bin_tiles_kernel<false>  per 64x64 tile of the grid, the number of blocks whose box meets it.
//   grid_bin_scan_kernel  one workgroup: tile counts -> list offsets and work-item offsets (an item = up to
//                         `chunk` list entries of one tile), totals for the host (list capacity, launch size).
//   grid_bin_tiles_kernel<true>   every (scan, block) is appended to the lists of the tiles its box meets.
//   grid_binned_kernel    one workgroup per item: the tile's two count layers, packed hit << 16 | sum, live in
//                         16 KiB of LDS; every wave walks list entries, 64 rays at a time, through exactly the
//                         part of the reference's Bresenham line inside the tile (closed-form restart as above,
//                         inverted to clip the walk in both coordinates: one LDS add per cell and no test); the
//                         tile is then added to the layers (plain read-modify-write when it has a single item).
// The work per item is bounded by `chunk`, the number of items follows the work, and a tile nobody's box meets
// costs nothing.  Counts are additive integers: the order of the list entries does not matter.
constexpr int BIN_CHUNK_MAX = 256;       // packed 16-bit counters: sum <= 64 chunk, hit <= 128 chunk per cell and item

__global__ __launch_bounds__(256) void grid_bin_scan_kernel(int* __restrict__ tile_cnt, int ntiles, int chunk_req,
                                                            int* __restrict__ tile_off, int* __restrict__ item_off,
                                                            int* __restrict__ totals /* entries, items, chunk */)
{
    __shared__ int sh[256];
    const int tid = threadIdx.x, per = (ntiles + 255) / 256, a = tid * per, b = a + per < ntiles ? a + per : ntiles;
    // list offsets
    int e = 0;
    for (int t = a; t < b; ++t) e += tile_cnt[t];
    sh[tid] = e;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int x = tid >= off ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += x;
        __syncthreads();
    }
    const int entries = sh[255];
    int eo = sh[tid] - e;
    // entries per work item: about 4096 items (two rounds of resident workgroups) unless the caller fixed it;
    // small chunks pay the tile's zeroing and flush more often, large ones leave compute units idle at the end
    int chunk = chunk_req;
    if (chunk <= 0) {
        chunk = (entries / 4096 + 15) & ~15;
        chunk = chunk < 32 ? 32 : (chunk > BIN_CHUNK_MAX ? BIN_CHUNK_MAX : chunk);
    }
    int it = 0;
    for (int t = a; t < b; ++t) it += (tile_cnt[t] + chunk - 1) / chunk;
    __syncthreads();
    sh[tid] = it;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int x = tid >= off ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += x;
        __syncthreads();
    }
    int io = sh[tid] - it;
    for (int t = a; t < b; ++t) {
        const int c = tile_cnt[t];
        tile_off[t] = eo; item_off[t] = io;
        eo += c; io += (c + chunk - 1) / chunk;
        tile_cnt[t] = 0;                                    // becomes the fill cursor
    }
    if (tid == 255) { tile_off[ntiles] = entries; item_off[ntiles] = sh[255]; totals[0] = entries; totals[1] = sh[255]; totals[2] = chunk; }
}

// The part of one ray inside the tile [X0, X1] x [Y0, Y1] (inclusive, already clipped to the grid).  lt: packed
// counters, hit << 16 | sum.  The reference's line is generated from the end with the smaller major coordinate
// (cell t = 0 .. len along the major axis); a standard integer DDA has taken floor((2 dmin t + len) / (2 len))
// minor steps before cell t.  That closed form is inverted here to clip the walk to the cells that lie in the
// tile in BOTH coordinates, so the loop adds without testing anything:
//   minor steps >= a  <=>  t >= ceil(len (2a - 1) / (2 dmin)),   minor steps <= b  <=>  t <= floor((len (2b + 1) - 1) / (2 dmin)).
// The robot cell (t = 0, or t = len when the line comes from the beam's end) is left out: grid_prep_kernel added
// its totals.  The beam's end cell receives its end hit (:72-75) and, when it is also the last generated cell
// (:84-86), a second one.
// floor(n / d) for n < 2^31, 1 <= d < 2^16: exact while the quotient is below 2^16; larger quotients come out as
// large (within 2^-20 of the truth) - the callers clamp those away.  A third of the compiler's 32-bit division.
__device__ __forceinline__ int bin_udiv(unsigned n, unsigned d)
{
    unsigned q = (unsigned)((float)n * __frcp_rn((float)d));
    const int r = (int)(n - q * d);
    q = r < 0 ? q - 1 : q;
    q = r >= (int)d ? q + 1 : q;
    return (int)q;
}

__device__ __forceinline__ void bin_walk(uint32_t* lt, int X0, int Y0, int X1, int Y1, int sx, int sy, int ex, int ey)
{
    const int dx = abs(ex - sx), dy = abs(ey - sy);
    const bool x_major = dy <= dx;
    const bool from_end = x_major ? sx > ex : sy > ey;
    const int len = x_major ? dx : dy, dmin = x_major ? dy : dx;
    if (ex >= X0 && ex <= X1 && ey >= Y0 && ey <= Y1)
        atomicAdd(&lt[(ex - X0) * BIN_T + (ey - Y0)], !from_end && len > 0 ? 0x20000u : 0x10000u);
    const int m0 = x_major ? (from_end ? ex : sx) : (from_end ? ey : sy);                  // major / minor of the first cell
    const int n0 = x_major ? (from_end ? ey : sy) : (from_end ? ex : sx);
    const int mstep = ((x_major ? ey - sy : ex - sx) * (from_end ? -1 : 1)) > 0 ? 1 : -1;
    const int M0 = x_major ? X0 : Y0, M1 = x_major ? X1 : Y1, N0 = x_major ? Y0 : X0, N1 = x_major ? Y1 : X1;
    int t_lo = M0 - m0 > 0 ? M0 - m0 : 0;
    int t_hi = M1 - m0 < len ? M1 - m0 : len;
    if (from_end) t_hi = t_hi < len - 1 ? t_hi : len - 1; else t_lo = t_lo > 1 ? t_lo : 1;
    const int a_lo = mstep > 0 ? N0 - n0 : n0 - N1;         // minor steps needed to enter the tile's minor range
    const int b_hi = mstep > 0 ? N1 - n0 : n0 - N0;         // minor steps after which it is left
    if (b_hi < 0 || a_lo > dmin || t_lo > t_hi) return;
    const bool small = len < 32768;                         // every product below fits 32 bits
    if (dmin > 0) {
        if (a_lo > 0) {
            const long long num = (long long)len * (2 * a_lo - 1) + 2 * dmin - 1;
            const int q = small ? bin_udiv((unsigned)num, (unsigned)(2 * dmin)) : (int)(num / (2ll * dmin));
            t_lo = q > t_lo ? q : t_lo;
        }
        if (b_hi < dmin) {
            const long long num = (long long)len * (2 * b_hi + 1) - 1;
            const int q = small ? bin_udiv((unsigned)num, (unsigned)(2 * dmin)) : (int)(num / (2ll * dmin));
            t_hi = q < t_hi ? q : t_hi;
        }
        if (t_lo > t_hi) return;
    }
    const int c_major = x_major ? BIN_T : 1, c_minor = x_major ? mstep : mstep * BIN_T;   // cell index steps
    const int inc_keep = 2 * dmin, inc_step = 2 * (dmin - len);
    int steps;                                                                           // minor steps taken before cell t_lo (len > 0 here)
    if (small) steps = bin_udiv(2u * (unsigned)dmin * (unsigned)t_lo + (unsigned)len, 2u * (unsigned)len);
    else steps = (int)((2ll * dmin * t_lo + len) / (2ll * len));
    const int minor = n0 + mstep * steps;
    int d = (int)(2ll * dmin * (t_lo + 1) - len - 2ll * len * steps);                   // |d| <= 2 len
    int cc = x_major ? (m0 + t_lo - X0) * BIN_T + (minor - Y0) : (minor - X0) * BIN_T + (m0 + t_lo - Y0);
    int n = t_hi - t_lo + 1;
    // Neighbouring beams run through the same cells while they are close to the robot, and 64 adds of one wave
    // instruction to one LDS address are executed one after the other (measured: half of this kernel's time).
    // The lanes are in bearing order, so equal cells sit in runs of adjacent lanes: the first lane of a run adds
    // the run's length, the others nothing (runs are cut at lane 32 so that 32-bit masks do).  Starting lane l
    // l mod k iterations late instead - rays of one scan that are a different number of steps from the robot
    // are in different cells - was measured slower for every k (the extra iterations cost more).
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t sentinel = 0x80000000u >> (lane & 31);
    do {
        const int prev = __builtin_amdgcn_update_dpp(-2, cc, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);   // lane 0 / idle neighbour: -2
        const unsigned long long act = __ballot(1);
        const unsigned long long heads = __ballot(cc != prev) | ~act | 0x100000001ull;   // run starts, idle lanes, lanes 0 and 32
        if (cc != prev || (lane & 31) == 0) {                                            // :87  (a branch for "no two neighbours agree" was measured: slower)
            const uint32_t h = lane < 32 ? (uint32_t)heads : (uint32_t)(heads >> 32);
            atomicAdd(&lt[cc], (uint32_t)__ffs((int)((h >> 1 >> (lane & 31)) | sentinel)));   // distance to the next run start above
        }
        const bool keep = d < 0;
        cc += keep ? c_major : c_major + c_minor;
        d += keep ? inc_keep : inc_step;
    } while (--n);
}

__global__ __launch_bounds__(256) void grid_binned_kernel(
    GridParam G, int n_beams, const int2* __restrict__ ends, const ScanInfo* __restrict__ info, int ntiles, int nty_all, int chunk,
    const int* __restrict__ tile_off, const int* __restrict__ item_off, const uint32_t* __restrict__ list,
    uint32_t* __restrict__ hit, uint32_t* __restrict__ sum)
{
    __shared__ uint32_t lt[BIN_T * BIN_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int item = blockIdx.x;
#ifdef LSA_GRID_STAMPS
    unsigned long long gt_ = __builtin_readcyclecounter();
#endif
    // the item's tile: the last tile whose first item is <= item (uniform; the offsets sit in L2)
    int lo = 0, hi = ntiles - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (item_off[mid] <= item) lo = mid; else hi = mid - 1;
    }
    const int tile = lo, part = item - item_off[tile], items_here = item_off[tile + 1] - item_off[tile];
    const int e0 = tile_off[tile] + part * chunk;
    const int e1 = e0 + chunk < tile_off[tile + 1] ? e0 + chunk : tile_off[tile + 1];
    const int X0 = (tile / nty_all) * BIN_T, Y0 = (tile % nty_all) * BIN_T;
    const int X1 = X0 + BIN_T - 1 < G.size_x - 1 ? X0 + BIN_T - 1 : G.size_x - 1;      // the tile clipped to the grid
    const int Y1 = Y0 + BIN_T - 1 < G.size_y - 1 ? Y0 + BIN_T - 1 : G.size_y - 1;
    GSTAMP(4);                                              // item -> tile search
    for (int k = tid; k < BIN_T * BIN_T; k += 256) lt[k] = 0;
    __syncthreads();
    GSTAMP(0);
    // every wave walks list entries, 64 rays at a time; the next entry's end cells are in flight meanwhile
    int q = e0 + wave;
    int2 e = make_int2(0, 0);
    int hs = 0, hx = 0, hy = 0, hn = 0;
    if (q < e1) {
        const uint32_t en = list[q];
        hs = (int)(en >> 8); const int bk = (int)(en & 255u);
        hx = info[hs].sx; hy = info[hs].sy; hn = info[hs].nb - 64 * bk;
        if (lane < hn) e = ends[(size_t)hs * n_beams + 64 * bk + lane];
    }
    for (; q < e1; q += 4) {
        const int sx = hx, sy = hy, nv = hn;
        const int2 ec = e;
        if (q + 4 < e1) {
            const uint32_t en = list[q + 4];
            hs = (int)(en >> 8); const int bk = (int)(en & 255u);
            hx = info[hs].sx; hy = info[hs].sy; hn = info[hs].nb - 64 * bk;
            if (lane < hn) e = ends[(size_t)hs * n_beams + 64 * bk + lane];
        }
        if (lane < nv) bin_walk(lt, X0, Y0, X1, Y1, sx, sy, ec.x, ec.y);
    }
    __syncthreads();
    GSTAMP(2);
    // flush: consecutive lanes -> consecutive y of one x row (contiguous in the [x][y] layers)
    for (int k = tid; k < BIN_T * BIN_T; k += 256) {
        const uint32_t v = lt[k];
        if (!v) continue;
        const int x = X0 + k / BIN_T, y = Y0 + k % BIN_T;            // (cells outside the grid were never added to)
        const uint32_t vh = v >> 16, vs = v & 0xffffu;
        const size_t c = (size_t)x * G.size_y + y;
        if (items_here == 1) { if (vh) hit[c] += vh; if (vs) sum[c] += vs; }
        else { if (vh) atomicAdd(&hit[c], vh); if (vs) atomicAdd(&sum[c], vs); }
    }
#ifdef LSA_GRID_STAMPS
    __syncthreads();
    GSTAMP(3);
    if (tid == 0) { atomicAdd(&lsa_grid_cycles[9], 1ull); atomicAdd(&lsa_grid_cycles[5], 1ull); }
#endif
}

// computeProbsOfMap: prob = hit / (double) sum clipped to 1, unseen cells 0.3 (PMap.cpp:119-136).
// Counts are converted to float first (the reference stores them as float).  The reference divides in double and rounds the
// quotient to float; for counts below 2^24 that is the correctly rounded FLOAT quotient (the exact quotient of two 24-bit
// integers is either a float or at least 2^-49 away, relatively, from every midpoint between two floats - too far for the
// 53-bit intermediate to land on one), which the build's IEEE float division gives at a third of the instructions.  Counts
// of 2^24 and more (the reference's float counters stop there) take the double path.  One cell per lane and trip: the pass is
// a pure stream (8 B read, 4 B written per cell; 4.2 TB/s cold, 8.7 TB/s from the Infinity Cache) - four cells per lane with
// 16-byte accesses measured 20 % SLOWER here, cold and cached alike, at every grid size tried.
__device__ __forceinline__ float grid_prob_cell(uint32_t hit, uint32_t sum)
{
    const float s = (float)sum, h = (float)hit;
    if (!(s > 0)) return (float)0.3;
    float p;
    if (__builtin_expect((hit | sum) >> 24, 0)) p = (float)((double)h / (double)s);
    else p = h / s;
    return (double)p > 1.0 ? 1.0f : p;
}

__global__ __launch_bounds__(256) void grid_prob_kernel(
    size_t cells, const uint32_t* __restrict__ hit, const uint32_t* __restrict__ sum, float* __restrict__ prob)
{
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x)
        prob[c] = grid_prob_cell(hit[c], sum[c]);
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
