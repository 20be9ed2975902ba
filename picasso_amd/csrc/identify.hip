// identify.hip — spot identification on a frame stack (HBM-bound stage).
//
// Replaces picasso/localize.py:97-134 (_local_maxima), :202-244 (_net_gradient),
// :247-292 (identify_in_image), :295-337 (ROI crop + float32 cast) and the
// frame loop of :604-636.  One workgroup owns one TH x TW tile of one frame:
//   1. tile + (h+1)-pixel halo -> LDS as float32 (wrapping row/col -1 to the
//      last row/col exactly like numba's negative indexing at :179-180);
//   2. separable first-argmax test: a pixel is a maximum iff it is strictly
//      greater than everything before it in row-major window order and >=
//      everything after it (np.argmax returns the FIRST maximum);
//   3. net gradient of every surviving candidate, float32, in the reference's
//      (k, l) order with unfused multiply/add so the value is bit-identical;
//   4. survivors (ng > min_ng) appended to an unordered record list.
// Three small kernels then order the records by (frame, y, x): per-frame
// counts -> exclusive scan -> scatter by frame -> rank sort inside each frame.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <mutex>
#include <vector>

#include "pmi_common.h"

// The net gradient must round like the reference's unfused float32 arithmetic:
// no a*b+c contraction anywhere in this translation unit.
#pragma clang fp contract(off)

namespace pmi {

// frames per chunk of the uint16 copy a 32-bit movie goes through (0 = as many as fit 1 GiB); pmi_identify_set_narrow_chunk
static int64_t g_narrow_chunk_frames = 0;

// float32 ops that must round exactly like the reference's unfused arithmetic.  They are
// defined HERE, under the pragma above, so the instructions carry no `contract` flag
// (the __f*_rn helpers of the HIP headers are compiled with contraction allowed).
static __device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
static __device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
static __device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }

constexpr int ID_TH = 32;    // tile rows
constexpr int ID_TW = 128;   // tile cols
constexpr int ID_NT = 256;   // threads per workgroup
constexpr int ID_SHARDS = 8; // record-list shards (= XCDs); counters[0..7] = shard counts, counters[8] = total

struct IdParams {
    int64_t Y, X;        // full frame
    int y0, x0, cy, cx;  // crop (ROI) origin and size
    int64_t f_lo;        // label of the first frame scanned
    int64_t frame_label_offset;  // added to labels (host chunking)
    int nframes;
    int box;
    int tiles_y, tiles_x;
    double min_ng;
    const int *gate;     // optional device flag: the launch does nothing if it is 0 (the packed scan took these frames)
};

template <typename T>
__global__ __launch_bounds__(ID_NT) void identify_scan_kernel(
    const T *__restrict__ movie, IdParams p, const float *__restrict__ uxy,
    Record *__restrict__ recs, long long cap, unsigned long long *__restrict__ shard_cnt,
    int *__restrict__ frame_count)
{
    extern __shared__ float smem[];
    const int box = p.box, h = box / 2, halo = h + 1;
    const int LW = ID_TW + 2 * halo, LH = ID_TH + 2 * halo;
    float *f = smem;                       // LH x LW pixels
    float *H = f + LH * LW;                // LH x LW row-window maxima
    float *sux = H + LH * LW;              // box x box
    float *suy = sux + box * box;
    unsigned *cand = (unsigned *)(suy + box * box);
    __shared__ int ncand;

    const int tid = threadIdx.x;
    const int tiles = p.tiles_y * p.tiles_x;
    const int fi = blockIdx.x / tiles;
    const int t = blockIdx.x - fi * tiles;
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    const T *src = movie + ((int64_t)(p.f_lo + fi) * p.Y + p.y0) * p.X + p.x0;

    if (p.gate && *p.gate == 0) return;
    if (tid == 0) ncand = 0;
    for (int idx = tid; idx < LH * LW; idx += ID_NT) {
        int r = idx / LW, c = idx - r * LW;
        int gy = ty * ID_TH - halo + r, gx = tx * ID_TW - halo + c;
        if (gy < 0) gy += p.cy;            // numba negative-index wrap
        if (gx < 0) gx += p.cx;
        gy = min(max(gy, 0), p.cy - 1);    // rows/cols past the far edge are never read
        gx = min(max(gx, 0), p.cx - 1);
        f[idx] = px_f32(src, (int64_t)gy * p.X + gx);
    }
    for (int idx = tid; idx < box * box; idx += ID_NT) {
        sux[idx] = uxy[idx];
        suy[idx] = uxy[box * box + idx];
    }
    __syncthreads();

    // pass A: horizontal window maximum for LDS rows 1 .. LH-2
    for (int idx = tid; idx < (ID_TH + 2 * h) * ID_TW; idx += ID_NT) {
        int r = 1 + idx / ID_TW, c = halo + idx % ID_TW;
        const float *row = f + r * LW + c;
        float m = row[-h];
        for (int d = -h + 1; d <= h; d++) m = fmaxf(m, row[d]);
        H[r * LW + c] = m;
    }
    __syncthreads();

    // pass B: first-argmax test
    for (int idx = tid; idx < ID_TH * ID_TW; idx += ID_NT) {
        int rr = idx / ID_TW, cc = idx - rr * ID_TW;
        int i = ty * ID_TH + rr, j = tx * ID_TW + cc;
        bool valid = i >= h && i < p.cy - h - 1 && j >= h && j < p.cx - h - 1;   // localize.py:122-123
        if (!valid) continue;
        int r = halo + rr, c = halo + cc;
        float v = f[r * LW + c];
        float before = H[(r - 1) * LW + c], after = H[(r + 1) * LW + c];
        for (int d = 2; d <= h; d++) {
            before = fmaxf(before, H[(r - d) * LW + c]);
            after = fmaxf(after, H[(r + d) * LW + c]);
        }
        for (int d = 1; d <= h; d++) {
            before = fmaxf(before, f[r * LW + c - d]);
            after = fmaxf(after, f[r * LW + c + d]);
        }
        if (v > before && v >= after) {
            int slot = atomicAdd(&ncand, 1);
            cand[slot] = ((unsigned)rr << 16) | (unsigned)cc;
        }
    }
    __syncthreads();

    // net gradient of each candidate: float32, (k, l) order, no FMA contraction
    const int nc = ncand;
    for (int q = tid; q < nc; q += ID_NT) {
        int rr = cand[q] >> 16, cc = cand[q] & 0xffff;
        int r = halo + rr, c = halo + cc;
        float ng = 0.0f;
        for (int k = 0; k < box; k++) {
            const float *rowm = f + (r - h + k - 1) * LW + (c - h);
            const float *row0 = rowm + LW;
            const float *rowp = row0 + LW;
            for (int l = 0; l < box; l++) {
                if (k == h && l == h) continue;
                float gy = sub_rn(rowp[l], rowm[l]);
                float gx = sub_rn(row0[l + 1], row0[l - 1]);
                float s = add_rn(mul_rn(gy, suy[k * box + l]), mul_rn(gx, sux[k * box + l]));
                ng = add_rn(ng, s);
            }
        }
        if ((double)ng > p.min_ng) {                                   // localize.py:288
            const int shard = blockIdx.x & (ID_SHARDS - 1);     // one slot counter per XCD: a single hot word
            unsigned long long pos = atomicAdd(&shard_cnt[shard], 1ull);   // saturates at ~88 atomics/us
            atomicAdd(&frame_count[fi], 1);
            if ((long long)pos < cap) {
                Record rec;
                rec.frame = (int32_t)(p.f_lo + fi + p.frame_label_offset);
                rec.yx = pack_yx(ty * ID_TH + rr + p.y0, tx * ID_TW + cc + p.x0);
                rec.slot = -1;
                rec.ng = ng;
                recs[(long long)shard * cap + pos] = rec;
            }
        }
    }
}

// exclusive scan of per-frame counts (single workgroup), also publishes the total
__global__ __launch_bounds__(1024) void frame_scan_kernel(const int *__restrict__ count, int *__restrict__ base,
                                                          int *__restrict__ cursor, int nframes,
                                                          unsigned long long *__restrict__ counters,
                                                          long long *__restrict__ out_n)
{
    // exclusive prefix sum of the per-frame counts, one workgroup: every thread sums a run of K consecutive frames,
    // the 1024 run totals are scanned with wave shuffles (two levels, two barriers), the runs are written out
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = (nframes + 1023) / 1024;
    const int i0 = tid * K;
    int local = 0;
    for (int k = 0; k < K; k++) local += (i0 + k < nframes) ? count[i0 + k] : 0;
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) before += w < wv ? wave_tot[w] : 0;
    int run = before + incl - local;
    for (int k = 0; k < K; k++) {
        const int i = i0 + k;
        if (i < nframes) { base[i] = run; cursor[i] = 0; run += count[i]; }
    }
    if (tid == 0) {
        unsigned long long total = 0;
        for (int sh = 0; sh < ID_SHARDS; sh++) total += counters[sh];
        counters[ID_SHARDS] = total;
        if (out_n) *out_n = (long long)total;
    }
}

__global__ void scatter_by_frame_kernel(const Record *__restrict__ recs, const unsigned long long *__restrict__ counters,
                                        long long cap, long long f_first, const int *__restrict__ base,
                                        int *__restrict__ cursor, Record *__restrict__ grouped)
{
    if ((long long)counters[ID_SHARDS] > cap) return;   // overflow: the caller retries with a larger capacity
    const int shard = blockIdx.y;
    const long long n = (long long)counters[shard];
    // fixed grid, grid-stride loop: the shard sizes are only known on the device.  A shard holds runs of
    // records of the same frame (one flush of the scan kernel appends several), so the lanes of a run
    // share ONE cursor atomic: the head lane of the run reserves the slots of the whole run.
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i0 = (long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63); i0 < n; i0 += stride) {
        const long long i = i0 + lane;
        const bool valid = i < n;
        Record r = {};
        int fi = -1;
        if (valid) { r = recs[(long long)shard * cap + i]; fi = (int)(r.frame - f_first); }
        const int prev = __shfl_up(fi, 1);
        const bool head = valid && (lane == 0 || prev != fi);
        const unsigned long long heads = __ballot(head);
        const unsigned long long vmask = __ballot(valid);
        if (valid) {
            const unsigned long long below = heads & ((2ull << lane) - 1ull);        // heads at or below this lane
            const int hl = 63 - __clzll(below);
            const unsigned long long above = heads & ~((2ull << lane) - 1ull);       // next head above, else end of valid lanes
            const int next = above ? __ffsll((long long)above) - 1 : __popcll(vmask);
            int slot0 = 0;
            if (head) slot0 = base[fi] + atomicAdd(&cursor[fi], next - lane);
            slot0 = __shfl(slot0, hl);
            grouped[slot0 + (lane - hl)] = r;
        }
    }
}

// Rank of each record among its frame's records by (y, x).  grid = (frames, split): every block stages the
// frame's keys in LDS (they fit unless a frame holds more than SORT_LDS maxima) and ranks its own slice of the
// records, so a dense frame (2048 x 2048: ~1600 maxima) is spread over `split` blocks instead of one wave.  With
// Y, X <= 65536 a key is (y << 16 | x) in 32 bits and one 16-byte LDS broadcast feeds four compares.
constexpr int SORT_LDS = 4096;
constexpr int SORT_THREADS = 256;
template <bool K32>
__global__ __launch_bounds__(SORT_THREADS) void sort_in_frame_kernel(const Record *__restrict__ grouped,
                                                                     const int *__restrict__ base,
                                                                     const int *__restrict__ count,
                                                                     const unsigned long long *__restrict__ counters,
                                                                     long long cap,
                                                                     int32_t *__restrict__ o_frame, int32_t *__restrict__ o_y,
                                                                     int32_t *__restrict__ o_x, float *__restrict__ o_ng,
                                                                     int32_t *__restrict__ o_slot)
{
    using key_t = typename std::conditional<K32, unsigned, unsigned long long>::type;
    __shared__ __attribute__((aligned(16))) key_t s_key[SORT_LDS];
    if ((long long)counters[ID_SHARDS] > cap) return;   // overflow: the caller retries with a larger capacity
    const int fi = blockIdx.x;
    const int m = count[fi], b = base[fi];
    const int per = (((m + (int)gridDim.y - 1) / (int)gridDim.y) + 63) & ~63;
    const int q0 = (int)blockIdx.y * per, q1 = min(m, q0 + per);
    if (q0 >= m) return;
    auto key_of = [](const Record &o) -> key_t { return (key_t)o.yx; };       // y << 16 | x: row-major order
    // keys go through LDS in tiles of SORT_LDS; a frame with fewer maxima than that (the usual case) is staged once
    auto stage = [&](int t0, int tm) {
        const int tmp = (tm + 3) & ~3;
        for (int q = threadIdx.x; q < tmp; q += SORT_THREADS)
            s_key[q] = q < tm ? key_of(grouped[b + t0 + q]) : ~(key_t)0;   // padding never counts: no key is above it
    };
    const bool single = m <= SORT_LDS;
    if (single) {
        stage(0, m);
        __syncthreads();
    }
    for (int qb = q0; qb < q1; qb += SORT_THREADS) {
        const int q = qb + (int)threadIdx.x;
        const bool act = q < q1;
        Record r = {};
        key_t key = 0;
        if (act) {
            r = grouped[b + q];
            key = key_of(r);
        }
        int rank = 0;
        for (int t0 = 0; t0 < m; t0 += SORT_LDS) {
            const int tm = min(SORT_LDS, m - t0);
            if (!single) {
                __syncthreads();
                stage(t0, tm);
                __syncthreads();
            }
            if constexpr (K32) {
                for (int t = 0; t < tm; t += 4) {                            // uniform address: one LDS broadcast per step
                    const uint4 k4 = *reinterpret_cast<const uint4 *>(&s_key[t]);
                    rank += (int)(k4.x < key) + (int)(k4.y < key) + (int)(k4.z < key) + (int)(k4.w < key);
                }
            } else {
                for (int t = 0; t < tm; t++) rank += s_key[t] < key;
            }
        }
        if (act) {
            o_frame[b + rank] = r.frame;
            o_y[b + rank] = (int32_t)(r.yx >> 16);
            o_x[b + rank] = (int32_t)(r.yx & 0xffffu);
            o_ng[b + rank] = r.ng;
            if (o_slot) o_slot[b + rank] = r.slot;
        }
    }
}


// float32 unit-vector tables, built on the host with IEEE float ops exactly as
// localize.py:279-286 does, cached on the device per box size.
static float *g_unit_tables[PMI_MAX_DEVICES][PMI_MAX_BOX + 1] = {};      // a table lives on the device it was made on
static std::mutex g_unit_tables_mu;

static int unit_table(int box, const float **d_tab)
{
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(g_unit_tables_mu);
    if (!g_unit_tables[dev][box]) {
        std::vector<float> tab(2 * box * box);
        int h = box / 2;
        for (int k = 0; k < box; k++)
            for (int l = 0; l < box; l++) {
                volatile float vx = (float)(h - l), vy = (float)(h - k);
                volatile float n2 = vx * vx + vy * vy;
                volatile float n = sqrtf(n2);
                tab[k * box + l] = vx / n;
                tab[box * box + k * box + l] = vy / n;
            }
        float *d = nullptr;
        PMI_HIP(hipMalloc(&d, tab.size() * sizeof(float)));
        PMI_HIP(hipMemcpy(d, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
        g_unit_tables[dev][box] = d;
    }
    *d_tab = g_unit_tables[dev][box];
    return PMI_OK;
}

static size_t scan_lds_bytes(int box)
{
    int halo = box / 2 + 1;
    size_t LW = ID_TW + 2 * halo, LH = ID_TH + 2 * halo;
    return (2 * LW * LH + 2 * box * box) * sizeof(float) + (size_t)(ID_TH * ID_TW / 4 + 64) * sizeof(unsigned);
}

template <typename T>
static int launch_scan(const void *d_movie, const IdParams &p, const float *d_tab, Record *recs, long long cap,
                       unsigned long long *n_total, int *frame_count, hipStream_t s)
{
    size_t lds = scan_lds_bytes(p.box);
    PMI_HIP(hipFuncSetAttribute((const void *)identify_scan_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long long blocks = (long long)p.nframes * p.tiles_y * p.tiles_x;
    if (blocks > 0x7fffffffLL) { set_error("identify: too many tiles (%lld)", blocks); return PMI_ERR_ARG; }
    hipLaunchKernelGGL(identify_scan_kernel<T>, dim3((unsigned)blocks), dim3(ID_NT), lds, s,
                       (const T *)d_movie, p, d_tab, recs, cap, n_total, frame_count);
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

int launch_scan_u16_fast(const void *d_movie, int dtype, int64_t Y, int64_t X, int y0, int x0, int cy, int cx, int64_t f_lo,
                         int64_t label_off, int nframes, int box, double min_ng, const float *d_tab, Record *recs,
                         long long cap, unsigned long long *n_total, int *frame_count, hipStream_t s, bool *handled,
                         const int *gate = nullptr, uint32_t *pix = nullptr, unsigned *pix_cnt = nullptr, unsigned pix_cap = 0,
                         bool defer = false, const float *fmovie = nullptr, int gate_want = 0);

// float32 / int32 / uint32 movies that hold 16-bit counts (a camera's counts saved wide): the frames are narrowed to
// uint16 — exactly, or not at all: any pixel that is not an integer in 0..65535 raises the chunk's flag — and take the
// packed scan; a flagged chunk takes the generic kernel.  Both launches are queued, each looks at the flag.
template <typename T>
__global__ __launch_bounds__(256) void narrow_to_u16_kernel(const T *__restrict__ src, long long n, uint16_t *__restrict__ dst,
                                                            int *__restrict__ flag)
{
    bool bad = false;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
        // another block has seen a pixel that is not a count: the copy is void already (an atomic load: `flag` is restrict-qualified
        // and a plain one is hoisted out of the loop)
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        T v[4];
        const bool full = i + 4 <= n;
        if (full) { const uint4 q = *reinterpret_cast<const uint4 *>(src + i); __builtin_memcpy(v, &q, 16); }
        else for (int k = 0; k < 4; k++) v[k] = i + k < n ? src[i + k] : (T)0;
        uint16_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if constexpr (sizeof(T) == 4 && !std::is_integral<T>::value) {
                const float f = (float)v[k];
                bad = bad || !(f >= 0.0f && f <= 65535.0f && f == __builtin_truncf(f));     // NaN fails every test
                o[k] = (uint16_t)(int)f;
            } else if constexpr (std::is_signed<T>::value) {
                bad = bad || v[k] < 0 || v[k] > 65535;
                o[k] = (uint16_t)v[k];
            } else {
                bad = bad || v[k] > 65535u;
                o[k] = (uint16_t)v[k];
            }
        }
        if (full) { uint2 w; w.x = o[0] | ((unsigned)o[1] << 16); w.y = o[2] | ((unsigned)o[3] << 16); *reinterpret_cast<uint2 *>(dst + i) = w; }
        else for (int k = 0; k < 4 && i + k < n; k++) dst[i + k] = o[k];
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// float32 movies with any other content (fractions, negatives, NaN, infinities): a chunk the count narrowing above has flagged is
// scanned by the packed kernel on 16-bit keys — the upper half of the order-preserving integer image of a float32 — which it
// builds in registers from the float32 rows (identify_fast.hip, PT_KEY; round 4 made a 16-bit copy first), and everything exact
// is decided on the float32 pixels.

// d_movie points at frame 0 of a stack holding at least frames [f_lo, f_hi].
// Labels written = frame index + label_offset.
int identify_impl(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X, int box, double min_ng,
                  const int64_t *roi4, int64_t f_lo, int64_t f_hi, int64_t label_offset,
                  int32_t *d_frame, int32_t *d_y, int32_t *d_x, float *d_ng, int64_t cap, int64_t *d_out_n,
                  hipStream_t s)
{
    if (box < 3 || box > PMI_MAX_BOX || (box & 1) == 0) { set_error("box must be odd, 3..%d (got %d)", PMI_MAX_BOX, box); return PMI_ERR_ARG; }
    if (F < 0 || Y <= 0 || X <= 0 || Y > 65535 || X > 65535) { set_error("bad movie shape (%lld,%lld,%lld)", (long long)F, (long long)Y, (long long)X); return PMI_ERR_ARG; }
    if (cap < 0) { set_error("negative capacity"); return PMI_ERR_ARG; }
    int64_t y0 = 0, x0 = 0, y1 = Y, x1 = X;
    if (roi4) { y0 = roi4[0]; x0 = roi4[1]; y1 = roi4[2]; x1 = roi4[3]; }
    if (y0 < 0 || x0 < 0 || y1 > Y || x1 > X) { set_error("roi outside the frame"); return PMI_ERR_ARG; }
    if (f_lo < 0) f_lo = 0;
    if (f_hi > F - 1) f_hi = F - 1;
    int64_t nf = f_hi - f_lo + 1;
    int64_t cy = y1 - y0, cx = x1 - x0;

    unsigned long long *d_total = nullptr;
    void *ptr = nullptr;
    int rc;
    if ((rc = scratch(SCR_COUNTERS, 256, &ptr)) != PMI_OK) return rc;
    d_total = (unsigned long long *)ptr;
    unsigned *pix_cnt = (unsigned *)((char *)ptr + 128);              // slot counters of the pixel hand-off, one per shard
    PMI_HIP(hipMemsetAsync(d_total, 0, 256, s));
    g_handoff.used = false;
    if (nf <= 0 || cy < box + 1 || cx < box + 1) {   // no interior pixel can be scanned
        PMI_HIP(hipMemsetAsync(d_out_n, 0, sizeof(int64_t), s));
        return PMI_OK;
    }
    if (nf > 0x7fffffffLL) { set_error("too many frames"); return PMI_ERR_ARG; }

    IdParams p;
    p.Y = Y; p.X = X; p.y0 = (int)y0; p.x0 = (int)x0; p.cy = (int)cy; p.cx = (int)cx;
    p.f_lo = f_lo; p.frame_label_offset = label_offset; p.nframes = (int)nf; p.box = box;
    p.tiles_y = (int)((cy + ID_TH - 1) / ID_TH); p.tiles_x = (int)((cx + ID_TW - 1) / ID_TW);
    p.min_ng = min_ng;

    const float *d_tab = nullptr;
    if ((rc = unit_table(box, &d_tab)) != PMI_OK) return rc;
    Record *recs = nullptr, *grouped = nullptr;
    int *fc = nullptr;
    size_t rec_bytes = (size_t)std::max<int64_t>(cap, 1) * sizeof(Record);
    if ((rc = scratch(SCR_RECORDS, rec_bytes * ID_SHARDS, &ptr)) != PMI_OK) return rc; recs = (Record *)ptr;
    if ((rc = scratch(SCR_RECORDS2, rec_bytes, &ptr)) != PMI_OK) return rc; grouped = (Record *)ptr;
    if ((rc = scratch(SCR_FRAME_COUNT, (size_t)nf * 3 * sizeof(int), &ptr)) != PMI_OK) return rc; fc = (int *)ptr;
    int *count = fc, *base = fc + nf, *cursor = fc + 2 * nf;
    PMI_HIP(hipMemsetAsync(count, 0, (size_t)nf * sizeof(int), s));

    {
        ScopedKernelTimer tm(s, &g_last_times.scan_ms);
        bool fast = false;
        rc = PMI_OK;
        // register-pipelined packed-u16 scan (identify_fast.hip: uint16, uint8, int16) when the layout allows
        // (a fused call may leave the exact stage to its fit's start-value kernel: g_defer_exact, pmi_common.h)
        const bool hand = g_handoff.pix && dtype == PMI_U16 && !g_defer_exact;
        rc = launch_scan_u16_fast(d_movie, dtype, Y, X, p.y0, p.x0, p.cy, p.cx, f_lo, label_offset, p.nframes, box, min_ng,
                                      d_tab, recs, cap, d_total, count, s, &fast, nullptr, hand ? g_handoff.pix : nullptr,
                                      hand ? pix_cnt : nullptr, hand ? g_handoff.cap_per_shard : 0u, g_defer_exact);
        g_handoff.used = hand && fast;
        p.gate = nullptr;
        const bool wide = dtype == PMI_U32 || dtype == PMI_I32 || dtype == PMI_F32;
        static const bool no_narrow = tuning_env("PMI_IDENTIFY_NO_NARROW") != nullptr;
        if (rc == PMI_OK && !fast && wide && !no_narrow) {
            // float32 movies, whatever they hold: the packed scan on 16-bit keys it builds from the float32 rows as it loads
            // them, every exact decision on the float32 pixels — one pass over 4 bytes per pixel (a movie that holds 16-bit
            // counts used to be narrowed to a uint16 copy first: 8 bytes moved per pixel, 2.3 against 3.2 TB/s of float32).
            // 32-bit integer movies (round 6): the same scan with the reference's cast to float32 (localize.py:332) in front
            // of the keys and of every exact read; frames too narrow for one row range per lane set keep the routes below.
            rc = launch_scan_u16_fast(d_movie, dtype, Y, X, p.y0, p.x0, p.cy, p.cx, f_lo, label_offset, p.nframes, box, min_ng,
                                      d_tab, recs, cap, d_total, count, s, &fast, nullptr, nullptr, nullptr, 0u, false,
                                      (const float *)d_movie, 0);
        }
        if (rc == PMI_OK && !fast && wide && !no_narrow && ((uintptr_t)d_movie & 15) == 0 && ((Y * X) & 3) == 0) {
            // chunks of frames through a uint16 copy of at most 1 GiB
            const int64_t frame_px = Y * X;
            const int64_t chunk = g_narrow_chunk_frames > 0 ? g_narrow_chunk_frames : std::max<int64_t>(1, ((int64_t)1 << 29) / frame_px);
            const int64_t nchunks = (nf + chunk - 1) / chunk;
            void *tmp = nullptr, *gptr = nullptr;
            if ((rc = scratch(SCR_NARROW, (size_t)std::min<int64_t>(nf, chunk) * frame_px * 2 + 64, &tmp)) != PMI_OK) return rc;
            if ((rc = scratch(SCR_GATES, (size_t)nchunks * sizeof(int), &gptr)) != PMI_OK) return rc;
            int *gates = (int *)gptr;
            PMI_HIP(hipMemsetAsync(gates, 0, (size_t)nchunks * sizeof(int), s));
            bool all_fast = true;
            for (int64_t ci = 0; ci < nchunks && rc == PMI_OK; ci++) {
                const int64_t c0 = ci * chunk, n = std::min<int64_t>(chunk, nf - c0);
                const long long npx = (long long)n * frame_px;
                const unsigned nb = (unsigned)std::min<long long>((npx / 4 + 255) / 256, 256 * 32);
                const char *srcp = (const char *)d_movie + (size_t)(f_lo + c0) * frame_px * 4;
                if (dtype == PMI_F32) hipLaunchKernelGGL(narrow_to_u16_kernel<float>, dim3(nb), dim3(256), 0, s, (const float *)srcp, npx, (uint16_t *)tmp, gates + ci);
                else if (dtype == PMI_I32) hipLaunchKernelGGL(narrow_to_u16_kernel<int32_t>, dim3(nb), dim3(256), 0, s, (const int32_t *)srcp, npx, (uint16_t *)tmp, gates + ci);
                else hipLaunchKernelGGL(narrow_to_u16_kernel<uint32_t>, dim3(nb), dim3(256), 0, s, (const uint32_t *)srcp, npx, (uint16_t *)tmp, gates + ci);
                PMI_HIP(hipGetLastError());
                bool f2 = false;
                rc = launch_scan_u16_fast(tmp, PMI_U16, Y, X, p.y0, p.x0, p.cy, p.cx, 0, f_lo + label_offset + c0, (int)n, box, min_ng,
                                          d_tab, recs, cap, d_total, count + c0, s, &f2, gates + ci);
                if (rc != PMI_OK) return rc;
                if (!f2) { all_fast = false; break; }          // the geometry rules the packed scan out: nothing was queued by it
                if (dtype == PMI_F32) {
                    // a flagged chunk of a float32 movie: the packed scan again, on keys it builds from the float32 rows as it
                    // loads them; exact decisions on the float32 pixels
                    bool f3 = false;
                    rc = launch_scan_u16_fast(srcp, PMI_F32, Y, X, p.y0, p.x0, p.cy, p.cx, 0, f_lo + label_offset + c0, (int)n, box, min_ng,
                                              d_tab, recs, cap, d_total, count + c0, s, &f3, gates + ci, nullptr, nullptr, 0u, false,
                                              (const float *)srcp, 1);
                    if (rc != PMI_OK) return rc;
                    if (f3) continue;
                }
                IdParams pc = p;
                pc.f_lo = f_lo + c0; pc.nframes = (int)n; pc.gate = gates + ci;
                switch (dtype) {
                case PMI_U32: rc = launch_scan<uint32_t>(d_movie, pc, d_tab, recs, cap, d_total, count + c0, s); break;
                case PMI_I32: rc = launch_scan<int32_t>(d_movie, pc, d_tab, recs, cap, d_total, count + c0, s); break;
                default: rc = launch_scan<float>(d_movie, pc, d_tab, recs, cap, d_total, count + c0, s); break;
                }
            }
            fast = all_fast;
        }
        if (rc == PMI_OK && !fast) snprintf(g_last_scan_kernel, sizeof(g_last_scan_kernel), "identify_scan_kernel (dtype %d)", dtype);
        if (rc == PMI_OK && !fast) switch (dtype) {
        case PMI_U16: rc = launch_scan<uint16_t>(d_movie, p, d_tab, recs, cap, d_total, count, s); break;
        case PMI_U8:  rc = launch_scan<uint8_t>(d_movie, p, d_tab, recs, cap, d_total, count, s); break;
        case PMI_I16: rc = launch_scan<int16_t>(d_movie, p, d_tab, recs, cap, d_total, count, s); break;
        case PMI_U32: rc = launch_scan<uint32_t>(d_movie, p, d_tab, recs, cap, d_total, count, s); break;
        case PMI_I32: rc = launch_scan<int32_t>(d_movie, p, d_tab, recs, cap, d_total, count, s); break;
        case PMI_F32: rc = launch_scan<float>(d_movie, p, d_tab, recs, cap, d_total, count, s); break;
        default: set_error("unknown dtype code %d", dtype); rc = PMI_ERR_ARG;
        }
        tm.stop();
        if (rc != PMI_OK) return rc;
    }
    hipLaunchKernelGGL(frame_scan_kernel, dim3(1), dim3(1024), 0, s, count, base, cursor, (int)nf, d_total,
                       (long long *)d_out_n);
    if (cap > 0) {
        const unsigned sb = (unsigned)std::min<long long>((cap + 255) / 256, 512);
        hipLaunchKernelGGL(scatter_by_frame_kernel, dim3(sb, ID_SHARDS), dim3(256), 0, s, recs, d_total, (long long)cap,
                           (long long)(f_lo + label_offset), base, cursor, grouped);
        const unsigned split = (unsigned)std::min<int64_t>(16, std::max<int64_t>(1, (Y * X + 262143) / 262144));
        if (Y <= 65536 && X <= 65536)
            hipLaunchKernelGGL(sort_in_frame_kernel<true>, dim3((unsigned)nf, split), dim3(SORT_THREADS), 0, s, grouped, base,
                               count, d_total, (long long)cap, d_frame, d_y, d_x, d_ng, g_handoff.d_slot);
        else
            hipLaunchKernelGGL(sort_in_frame_kernel<false>, dim3((unsigned)nf, split), dim3(SORT_THREADS), 0, s, grouped, base,
                               count, d_total, (long long)cap, d_frame, d_y, d_x, d_ng, g_handoff.d_slot);
    }
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

// Net gradient at given pixels of one float32 image with caller-supplied unit vectors
// (picasso/localize.py:202-244 _net_gradient, :153-181 _gradient_at): float32 accumulator, window rows
// then columns, centre skipped, one rounding per operation; row / column -1 wraps to the last one.
__global__ __launch_bounds__(256) void net_gradient_kernel(const float *__restrict__ img, int Y, int X,
                                                           const int32_t *__restrict__ py, const int32_t *__restrict__ px,
                                                           int64_t n, int box, const float *__restrict__ uy,
                                                           const float *__restrict__ ux, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int h = box / 2, yi = py[i], xi = px[i];
    float ng = 0.0f;
    for (int kk = 0; kk < box; kk++) {
        const int k = yi - h + kk;
        for (int ll = 0; ll < box; ll++) {
            const int m = xi - h + ll;
            if (k == yi && m == xi) continue;
            auto wrap = [](int v, int N) { return v < 0 ? v + N : v; };      // each index wraps on its own
            const int kw = wrap(k, Y), mw = wrap(m, X);
            const float gy = img[(int64_t)wrap(k + 1, Y) * X + mw] - img[(int64_t)wrap(k - 1, Y) * X + mw];
            const float gx = img[(int64_t)kw * X + wrap(m + 1, X)] - img[(int64_t)kw * X + wrap(m - 1, X)];
            const float t1 = gy * uy[kk * box + ll];
            const float t2 = gx * ux[kk * box + ll];
            const float sum = t1 + t2;
            ng = ng + sum;
        }
    }
    out[i] = ng;
}

}  // namespace pmi

extern "C" {

int pmi_identify_set_narrow_chunk(int64_t frames)
{
    if (frames < 0) { pmi::set_error("chunk length must not be negative"); return PMI_ERR_ARG; }
    pmi::g_narrow_chunk_frames = frames;
    return PMI_OK;
}

int pmi_identify_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X, int box, double min_ng,
                     const int64_t *roi4, int64_t f_lo, int64_t f_hi, int32_t *d_frame, int32_t *d_y,
                     int32_t *d_x, float *d_ng, int64_t cap, int64_t *d_out_n, void *stream)
{
    return pmi::identify_impl(d_movie, dtype, F, Y, X, box, min_ng, roi4, f_lo, f_hi, 0, d_frame, d_y, d_x, d_ng,
                              cap, d_out_n, (hipStream_t)stream);
}

static size_t dtype_size(int dtype)
{
    switch (dtype) {
    case PMI_U8: return 1;
    case PMI_U16: case PMI_I16: return 2;
    default: return 4;
    }
}

// Host-buffer form: streams the movie through HBM in frame chunks.
int pmi_identify(const void *movie, int dtype, int64_t F, int64_t Y, int64_t X, int box, double min_ng,
                 const int64_t *roi4, int64_t f_lo, int64_t f_hi, int32_t *out_frame, int32_t *out_y,
                 int32_t *out_x, float *out_ng, int64_t cap, int64_t *out_n)
{
    using namespace pmi;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    if (!movie || !out_n) { set_error("null pointer"); return PMI_ERR_ARG; }
    if (dtype < 0 || dtype > PMI_F32) { set_error("unknown dtype code %d", dtype); return PMI_ERR_ARG; }
    if (f_lo < 0) f_lo = 0;
    if (f_hi > F - 1) f_hi = F - 1;
    *out_n = 0;
    if (f_hi < f_lo) return PMI_OK;
    const size_t frame_bytes = (size_t)Y * X * dtype_size(dtype);
    const int64_t chunk = std::max<int64_t>(1, (int64_t)((size_t)1 << 30) / (int64_t)frame_bytes);   // ~1 GiB of frames
    void *d_chunk = nullptr, *ptr = nullptr;
    int rc;
    if ((rc = scratch(SCR_STAGE_A, (size_t)std::min<int64_t>(chunk, f_hi - f_lo + 1) * frame_bytes, &d_chunk)) != PMI_OK) return rc;
    int64_t dcap = std::max<int64_t>(cap, 1);
    if ((rc = scratch(SCR_STAGE_B, (size_t)dcap * 16 + 64, &ptr)) != PMI_OK) return rc;
    int32_t *d_frame = (int32_t *)ptr, *d_y = d_frame + dcap, *d_x = d_y + dcap;
    float *d_ng = (float *)(d_x + dcap);
    int64_t *d_n = (int64_t *)(d_ng + dcap);
    int64_t total = 0;
    bool overflow = false;
    for (int64_t c0 = f_lo; c0 <= f_hi; c0 += chunk) {
        int64_t c1 = std::min(f_hi, c0 + chunk - 1), nfc = c1 - c0 + 1;
        PMI_HIP(hipMemcpy(d_chunk, (const char *)movie + (size_t)c0 * frame_bytes, (size_t)nfc * frame_bytes, hipMemcpyHostToDevice));
        int64_t room = overflow ? 0 : std::max<int64_t>(cap - total, 0);
        rc = identify_impl(d_chunk, dtype, nfc, Y, X, box, min_ng, roi4, 0, nfc - 1, c0, d_frame, d_y, d_x, d_ng,
                           room, d_n, nullptr);
        if (rc != PMI_OK) return rc;
        int64_t n = 0;
        PMI_HIP(hipMemcpy(&n, d_n, sizeof(n), hipMemcpyDeviceToHost));
        if (n > room) overflow = true;
        else if (n > 0) {
            PMI_HIP(hipMemcpy(out_frame + total, d_frame, (size_t)n * 4, hipMemcpyDeviceToHost));
            PMI_HIP(hipMemcpy(out_y + total, d_y, (size_t)n * 4, hipMemcpyDeviceToHost));
            PMI_HIP(hipMemcpy(out_x + total, d_x, (size_t)n * 4, hipMemcpyDeviceToHost));
            PMI_HIP(hipMemcpy(out_ng + total, d_ng, (size_t)n * 4, hipMemcpyDeviceToHost));
        }
        total += n;
    }
    *out_n = total;
    if (overflow) { set_error("identify: capacity %lld too small, %lld rows needed", (long long)cap, (long long)total); return PMI_ERR_CAPACITY; }
    return PMI_OK;
}


int pmi_net_gradient(const float *image, int64_t Y, int64_t X, const int32_t *y, const int32_t *x, int64_t n, int box,
                     const float *uy, const float *ux, float *out_ng)
{
    using namespace pmi;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    if (n < 0 || (n > 0 && (!image || !y || !x || !uy || !ux || !out_ng))) { set_error("null pointer"); return PMI_ERR_ARG; }
    if (box < 1 || box > PMI_MAX_BOX || (box & 1) == 0) { set_error("box must be odd in [1, %d], got %d", PMI_MAX_BOX, box); return PMI_ERR_ARG; }
    if (n == 0) return PMI_OK;
    const int h = box / 2;
    // what the reference's unchecked indexing can reach without leaving the array: y + h + 1 and x + h + 1 inside,
    // y - h - 1 and x - h - 1 not below -1 ... -Y (negative indices wrap once)
    for (int64_t i = 0; i < n; i++)
        if (y[i] + h + 1 >= Y || x[i] + h + 1 >= X || y[i] - h - 1 < -Y || x[i] - h - 1 < -X) {
            set_error("net_gradient: position %lld (%d, %d) reaches outside the %lld x %lld image", (long long)i, y[i], x[i], (long long)Y, (long long)X);
            return PMI_ERR_ARG;
        }
    void *pa = nullptr, *pb = nullptr;
    int rc;
    const size_t img_bytes = (size_t)Y * X * 4, tab = (size_t)box * box * 4;
    if ((rc = scratch(SCR_STAGE_A, img_bytes, &pa)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_B, (size_t)n * 12 + 2 * tab + 64, &pb)) != PMI_OK) return rc;
    int32_t *d_y = (int32_t *)pb, *d_x = d_y + n;
    float *d_out = (float *)(d_x + n), *d_uy = d_out + n, *d_ux = d_uy + box * box;
    PMI_HIP(hipMemcpy(pa, image, img_bytes, hipMemcpyHostToDevice));
    PMI_HIP(hipMemcpy(d_y, y, (size_t)n * 4, hipMemcpyHostToDevice));
    PMI_HIP(hipMemcpy(d_x, x, (size_t)n * 4, hipMemcpyHostToDevice));
    PMI_HIP(hipMemcpy(d_uy, uy, tab, hipMemcpyHostToDevice));
    PMI_HIP(hipMemcpy(d_ux, ux, tab, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(net_gradient_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const float *)pa, (int)Y,
                       (int)X, d_y, d_x, n, box, d_uy, d_ux, d_out);
    PMI_HIP(hipGetLastError());
    PMI_HIP(hipMemcpy(out_ng, d_out, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PMI_OK;
}

}  // extern "C"
