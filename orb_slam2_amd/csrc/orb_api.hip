// Host side of liborbhip: context, level geometry, resize tap tables, buffer
// management and the extraction entry points of include/orb_hip.h.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "orb_internal.h"
#include "quadtree_core.h"  // qt_code_x / qt_code_y for the path tables

// ---- error reporting ----------------------------------------------------------------
static thread_local char t_err[512] = "";

void orb_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* orb_last_error(void) { return t_err; }
extern "C" const char* orb_version(void) { return "orbhip 0.1 (gfx950)"; }


static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_floor(double v) {
    int i = (int)v;
    return i - (i > v);
}
static inline int cv_ceil(double v) {
    int i = (int)v;
    return i + (i < v);
}

// ---- context --------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return ORB_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) {
            orb_set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
            return ORB_ERR_NOMEM;
        }
        bytes = need;
        return ORB_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct PinBuf {  // page-locked host staging: one DMA per direction instead of per-row / pageable copies
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return ORB_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipHostMalloc(&p, need, hipHostMallocDefault);
        if (e != hipSuccess) {
            orb_set_error("hipHostMalloc(%zu) failed: %s", need, hipGetErrorString(e));
            return ORB_ERR_NOMEM;
        }
        bytes = need;
        return ORB_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// Host threads a context keeps for staging batches of pageable frames (created when the first such batch arrives,
// asleep on a condition variable between batches): run(n, fn) calls fn(0) on the caller and fn(1) .. fn(n - 1) on
// the workers and returns when all have finished.
class StagePool {
public:
    ~StagePool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cvStart_.notify_all();
        for (std::thread& t : th_) t.join();
    }
    void run(int nthreads, const std::function<void(int)>& fn) {
        if (nthreads <= 1) {
            fn(0);
            return;
        }
        while ((int)th_.size() < nthreads - 1) {
            const int id = (int)th_.size() + 1;
            th_.emplace_back([this, id] { worker(id); });
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            active_ = nthreads;
            pending_ = nthreads - 1;
            gen_++;
        }
        cvStart_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(m_);
        cvDone_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void worker(int id) {
        unsigned seen = 0;
        for (;;) {
            const std::function<void(int)>* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cvStart_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                if (id < active_) job = job_;
            }
            if (!job) continue;
            (*job)(id);
            bool last;
            {
                std::lock_guard<std::mutex> lk(m_);
                last = --pending_ == 0;
            }
            if (last) cvDone_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cvStart_, cvDone_;
    const std::function<void(int)>* job_ = nullptr;
    unsigned gen_ = 0;
    int active_ = 0, pending_ = 0;
    bool stop_ = false;
};

#define ORB_MAX_PARTS 4
#define ORB_STAGE_THREADS 16     // most host threads that stage a batch of host frames (orb_ctx::stageThreads are used)
#define ORB_STAGE_SLICE 4        // batches below this many frames are staged and copied in one piece
#define ORB_STAGE_SLICE_BYTES (5 << 20)  // bytes per host-to-device copy of a larger batch, about
#define ORB_MAX_STAGE_SLICES 64
#define ORB_MIN_PART_FRAMES 8   // groups smaller than this do not fill the device; the batch is then not split

struct orb_ctx {
    orb_params prm;
    double scaleFactorD;  // the reference keeps scaleFactor in a double member (include/ORBextractor.h:88)
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> perLevel;
    int umax[16];
    OrbGeom geom;          // for the current frame size
    bool geomValid = false;
    std::vector<OrbAreaTap> areaTaps;
    std::vector<OrbLinTap> linTaps;
    std::vector<uint32_t> codeTab;  // quadtree path words per column / row of every level
    hipStream_t stream = nullptr;
    // A batch is enqueued as up to ORB_MAX_PARTS groups of frames, the first on the caller's stream and
    // the others on these, so that the latency-bound stages of one group (quadtree, small pyramid
    // levels) run under the throughput-bound stages of another.
    hipStream_t aux[ORB_MAX_PARTS - 1] = {};
    hipEvent_t evFork = nullptr, evJoin[ORB_MAX_PARTS - 1] = {};
    hipEvent_t evIn[ORB_MAX_PARTS] = {};  // host-pointer batches: entry i marks the arrival of frame group i's frames (i >= 1)
    int parts = 2;
    int hostParts = 3;  // frame groups of a host-pointer batch: the groups are also what overlaps the frames' way over the link with the extraction
    int stageThreads = 4;
    DevBuf dGeom, dAreaTaps, dLinTaps, dCodeTab, dInput, dPyr, dCellKeys, dCellCount, dSort, dNodes, dInts,
        dSelXY, dSelScore, dLevelCount, dLevelKeys, dQtRetry, dTmpA, dTmpB;
    // One block for what the kernels report and what the host-pointer entry points read back, so that one copy
    // fetches it all: [error flag word (64 bytes) | keypoint counts | keypoints (cap per frame) | descriptors].  The
    // error flag word is used by every entry point; the
    // rest is laid out per call for the call's frame count (orb_extract_batch).
    DevBuf dHostOut;
    int* dErr() const { return (int*)dHostOut.p; }
    PinBuf pinIn, pinOut;  // host-pointer entry points only, sized on first use
    StagePool* pool = nullptr;  // staging threads of the host-pointer entry points, made on first use
    // a host-pointer batch between its two halves (extract_batch_begin / extract_batch_finish)
    struct {
        bool open = false, direct = false;
        int n = 0, cap = 0;
        int ngroups = 1, groupStart[ORB_MAX_PARTS + 1] = {};  // group i > 0 = frames [groupStart[i], groupStart[i + 1]), done at evJoin[i - 1]
        double t0 = 0;               // ORBHIP_HOST_TRACE: when the call began
        size_t offCnt = 0, offKps = 0, offDesc = 0, total = 0;
        size_t offUn = 0;  // undistorted keypoints in the result block (0: none)
    } hostBatch;
    int lastFrames = 0;
    int lastGroups = 0, lastGroupFrame0[ORB_MAX_PARTS] = {};  // frame groups of the last batch whose quadtree ran in two passes
    int inputStride = 0;
    const uint8_t* lastImg0 = nullptr;
    int lastStride0 = 0;
    int64_t lastFrameStride0 = 0;
    // Frame::mnMinX, mnMaxX, mnMinY, mnMaxY (static members of Frame in the reference, set from the
    // first frame): used by the windowed matchers when set, else the image rectangle
    bool haveBounds = false;
    float bounds[4] = {0, 0, 0, 0};
    // matcher scratch lives in kernels_match.hip
    void* matchState = nullptr;
    // bag-of-words matcher scratch lives in kernels_bow.hip
    void* bowState = nullptr;
    // stage timing: ring of event sets, one set (5 events) per recorded batch
    bool timing = false;
    std::vector<hipEvent_t> events;
    int evCount = 0;
};
#define ORB_EV_RING 64

void orb_match_state_free(void* p);
void orb_bow_state_free(void* p);

// accessors for the matcher's translation unit
int orb_ctx_device(const orb_ctx* c) { return c->prm.device; }
unsigned orb_ctx_quirks(const orb_ctx* c) { return c->prm.quirks; }
int orb_ctx_max_batch(const orb_ctx* c) { return c->prm.max_batch; }
void** orb_ctx_match_state(orb_ctx* c) { return &c->matchState; }
void* orb_ctx_stream(orb_ctx* c) { return (void*)c->stream; }
void** orb_ctx_bow_state(orb_ctx* c) { return &c->bowState; }
int orb_ctx_get_bounds(const orb_ctx* c, float b[4]) {
    if (c->haveBounds) memcpy(b, c->bounds, sizeof(c->bounds));
    return c->haveBounds ? 1 : 0;
}
void orb_ctx_store_bounds(orb_ctx* c, const float* b) {
    c->haveBounds = b != nullptr;
    if (b) memcpy(c->bounds, b, sizeof(c->bounds));
}

// ORBextractor::ORBextractor, src/ORBextractor.cc:369-436 (inverse tables divide by the forward tables).
static int build_tables(orb_ctx* c) {
    const orb_params& p = c->prm;
    const int L = p.nlevels;
    c->scaleFactorD = (double)p.scale_factor;
    c->scale.assign(L, 1.f);
    c->sigma2.assign(L, 1.f);
    for (int i = 1; i < L; i++) {
        c->scale[i] = (float)(c->scale[i - 1] * c->scaleFactorD);
        c->sigma2[i] = c->scale[i] * c->scale[i];
    }
    c->invScale.resize(L);
    c->invSigma2.resize(L);
    for (int i = 0; i < L; i++) {
        c->invScale[i] = 1.0f / c->scale[i];
        c->invSigma2[i] = 1.0f / c->sigma2[i];
    }
    c->perLevel.assign(L, 0);
    const float factor = (float)(1.0f / c->scaleFactorD);
    float nDesired = p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) {
        c->perLevel[l] = cv_round(nDesired);
        sum += c->perLevel[l];
        nDesired *= factor;
    }
    c->perLevel[L - 1] = std::max(p.nfeatures - sum, 0);
    // patch row half-widths, :420-435
    for (int i = 0; i < 16; i++) c->umax[i] = 0;
    const int HALF = 15;
    const int vmax = cv_floor(HALF * sqrt(2.f) / 2 + 1);
    const int vmin = cv_ceil(HALF * sqrt(2.f) / 2);
    const double hp2 = HALF * HALF;
    int v, v0;
    for (v = 0; v <= vmax; ++v) c->umax[v] = (int)lrint(sqrt(hp2 - v * v));
    for (v = HALF, v0 = 0; v >= vmin; --v) {
        if (p.quirks & ORB_Q_UMAX_FORK) {
            while (c->umax[v] == c->umax[v0 + 1]) ++v0;
        } else {
            while (c->umax[v0] == c->umax[v0 + 1]) ++v0;
        }
        c->umax[v] = v0;
        ++v0;
    }
    return ORB_OK;
}

// cv::resize(INTER_AREA) per-axis table (computeResizeAreaTab): covered source cells and fractions.
static int area_taps(int ssize, int dsize, std::vector<OrbAreaTap>& out) {
    const double scale = 1. / ((double)dsize / ssize);
    if (!(scale >= 1.0)) {
        orb_set_error("area resize needs a shrinking level ratio (got %f)", scale);
        return ORB_ERR_UNSUPPORTED;
    }
    if (std::fabs(scale - (int)scale) < 2.220446049250313e-16) {
        orb_set_error("level ratio %d:%d is an exact integer; cv::resize takes a different INTER_AREA path there", ssize,
                      dsize);
        return ORB_ERR_UNSUPPORTED;
    }
    for (int d = 0; d < dsize; d++) {
        OrbAreaTap t;
        memset(&t, 0, sizeof(t));
        int k = 0;
        const double f1 = d * scale, f2 = f1 + scale;
        const double cell = std::min(scale, ssize - f1);
        int s1 = cv_ceil(f1), s2 = cv_floor(f2);
        s2 = std::min(s2, ssize - 1);
        s1 = std::min(s1, s2);
        auto push = [&](int si, float a) -> bool {
            if (k >= ORB_MAX_TAPS) return false;
            t.si[k] = si;
            t.a[k] = a;
            k++;
            return true;
        };
        bool ok = true;
        if (s1 - f1 > 1e-3) ok = ok && push(s1 - 1, (float)((s1 - f1) / cell));
        for (int s = s1; s < s2; s++) ok = ok && push(s, (float)(1.0 / cell));
        if (f2 - s2 > 1e-3) ok = ok && push(s2, (float)(std::min(std::min(f2 - s2, 1.), cell) / cell));
        if (!ok) {
            orb_set_error("scale factor too large: more than %d source cells per destination pixel", ORB_MAX_TAPS);
            return ORB_ERR_UNSUPPORTED;
        }
        for (int j = k; j < ORB_MAX_TAPS; j++) t.si[j] = t.si[k - 1];  // unused taps: weight 0 on a cell that is staged anyway
        out.push_back(t);
    }
    return ORB_OK;
}

// cv::resize(INTER_LINEAR) per-axis table: source index pair and 11-bit weights.
static void linear_taps(int ssize, int dsize, bool clampWeights, std::vector<OrbLinTap>& out) {
    const double scale = 1. / ((double)dsize / ssize);
    auto sat16 = [](int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); };
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        OrbLinTap t;
        if (clampWeights) {  // horizontal: out-of-range taps get weight 0
            if (s < 0) f = 0, s = 0;
            if (s >= ssize - 1) f = 0, s = ssize - 1;
            t.s0 = s;
            t.s1 = std::min(s + 1, ssize - 1);
        } else {  // vertical: rows are clamped, weights kept
            t.s0 = std::min(std::max(s, 0), ssize - 1);
            t.s1 = std::min(std::max(s + 1, 0), ssize - 1);
        }
        t.a0 = sat16(cv_round((1.f - f) * 2048));
        t.a1 = sat16(cv_round(f * 2048));
        out.push_back(t);
    }
}

// Every tap of every resize tile must lie inside the source span the kernel stages for that tile.
static int check_tile_spans(int ssize, int dsize, int tile, const OrbAreaTap* at, const OrbLinTap* lt) {
    const uint64_t scale = orb_resize_scale_fx(ssize, dsize);
    for (int d = 0; d < dsize; d++) {
        const int d0 = d / tile * tile, d1 = std::min(d0 + tile, dsize);
        int s0, s1, lo, hi;
        orb_resize_span(d0, d1, scale, ssize, &s0, &s1);
        if (at) {
            lo = hi = at[d].si[0];
            for (int k = 1; k < ORB_MAX_TAPS && at[d].a[k] != 0.f; k++) hi = at[d].si[k];
        } else {
            lo = std::min(lt[d].s0, lt[d].s1);
            hi = std::max(lt[d].s0, lt[d].s1);
        }
        if (lo < s0 || hi > s1) {
            orb_set_error("resize %d -> %d: taps of destination index %d leave the staged source span", ssize, dsize, d);
            return ORB_ERR_UNSUPPORTED;
        }
    }
    return ORB_OK;
}

// Largest source extent of a resize tile of `tile` destination indices when the kernel reads a tile's span off the
// tap table (first tap of its first index .. last tap of its last): bytes per LDS row from a 4-byte-aligned start for
// the horizontal axis, rows for the vertical one.  0 if some tap of a tile lies outside that span (taps not ordered).
static int exact_tile_extent(int dsize, int tile, bool dwordAligned, const OrbAreaTap* at, const OrbLinTap* lt) {
    int best = 0;
    for (int d0 = 0; d0 < dsize; d0 += tile) {
        const int d1 = std::min(d0 + tile, dsize);
        const int s0 = at ? at[d0].si[0] : lt[d0].s0;
        const int s1 = at ? at[d1 - 1].si[ORB_MAX_TAPS - 1] : lt[d1 - 1].s1;
        for (int d = d0; d < d1; d++) {
            int lo, hi;
            if (at) {
                lo = hi = at[d].si[0];
                for (int k = 0; k < ORB_MAX_TAPS; k++) lo = std::min(lo, at[d].si[k]), hi = std::max(hi, at[d].si[k]);
            } else {
                lo = std::min(lt[d].s0, lt[d].s1);
                hi = std::max(lt[d].s0, lt[d].s1);
            }
            if (lo < s0 || hi > s1) return 0;
        }
        best = std::max(best, dwordAligned ? 4 * ((s1 - (s0 & ~3) + 4) >> 2) : s1 - s0 + 1);
    }
    return best;
}

static inline int halfUp(int v) { return (v + 1) >> 1; }

// Regions of the few-frames pyramid kernel (OrbGeom::chainX / chainY) from the tap tables: top level first, every
// lower level's region is what the tile owns there joined with every source index the region above is made from.
static void build_chain(OrbGeom& g, const std::vector<OrbAreaTap>& at, const std::vector<OrbLinTap>& lt) {
    const bool area = (g.quirks & ORB_Q_RESIZE_AREA) != 0;
    const int L = g.nlevels;
    g.chainGX = g.chainGY = g.chainLds = 0;
    if (L < 2) return;
    // tiles of about 40 pixels at level 0: at 640x480 192 workgroups, each computing ~4x what it owns (borders
    // included); coarser tiles recompute less but leave most of the device idle (measured: 80 -> 31 us, 40 -> 20 us)
    const int tileSide = 40;
    const int GX = std::max(1, std::min(ORB_CHAIN_G, (g.w + tileSide - 1) / tileSide)), GY = std::max(1, std::min(ORB_CHAIN_G, (g.h + tileSide - 1) / tileSide));
    auto axis = [&](bool isX, int G, short tab[][ORB_CHAIN_G][4]) {
        for (int i = 0; i < G; i++) {
            int r0 = 0, r1 = 0;
            for (int k = L - 1; k >= 0; k--) {
                const int size = isX ? g.lv[k].w : g.lv[k].h;
                const int o0 = (int)((int64_t)i * size / G), o1 = (int)((int64_t)(i + 1) * size / G);
                int n0 = o0, n1 = o1;  // what is needed at this level
                if (k < L - 1 && r1 > r0) {  // sources of the region above
                    const int tb = isX ? g.lv[k + 1].tabX : g.lv[k + 1].tabY;
                    int lo = size, hi = -1;
                    for (int d = r0; d < r1; d++) {
                        if (area) {
                            const OrbAreaTap& t = at[(size_t)tb + d];
                            lo = std::min(lo, t.si[0]);
                            for (int j = 0; j < ORB_MAX_TAPS; j++) hi = std::max(hi, t.si[j]);  // (unused taps name the last used cell)
                        } else {
                            const OrbLinTap& t = lt[(size_t)tb + d];
                            lo = std::min(lo, std::min(t.s0, t.s1));
                            hi = std::max(hi, std::max(t.s0, t.s1));
                        }
                    }
                    if (o1 > o0) {
                        n0 = std::min(o0, lo);
                        n1 = std::max(o1, hi + 1);
                    } else {
                        n0 = lo;
                        n1 = hi + 1;
                    }
                }
                if (k == 0 && isX) n0 &= ~3;  // level 0 is read as dwords
                tab[k][i][0] = (short)n0;
                tab[k][i][1] = (short)n1;
                tab[k][i][2] = (short)o0;
                tab[k][i][3] = (short)o1;
                r0 = n0;
                r1 = n1;
            }
        }
    };
    axis(true, GX, g.chainX);
    axis(false, GY, g.chainY);
    size_t most = 0;
    int widest = 0;
    for (int k = 0; k < L; k++)
        for (int i = 0; i < GX; i++)
            for (int j = 0; j < GY; j++) {
                const int pw = ((g.chainX[k][i][1] - g.chainX[k][i][0]) + 3) & ~3;
                most = std::max(most, (size_t)pw * (size_t)(g.chainY[k][j][1] - g.chainY[k][j][0]));
                widest = std::max(widest, pw);
            }
    // two buffers, each with room for the dword reads that start at a row's last pixel; level 0 is staged by at most
    // 128 threads per row
    const size_t lds = 2 * ((most + 16 + 15) & ~(size_t)15);
    if (lds > 64 * 1024 || widest / 4 > 128 || g.w > 32767 || g.h > 32767) return;
    g.chainGX = GX;
    g.chainGY = GY;
    g.chainLds = (int)lds;
}

// Level sizes (src/ORBextractor.cc:917-918), FAST cell grid (:808-823), quadtree
// roots (:574-576) and every derived capacity for a w x h frame.
static int compute_geom(orb_ctx* c, int w, int h) {
    if (c->geomValid && c->geom.w == w && c->geom.h == h) return ORB_OK;
    const orb_params& p = c->prm;
    OrbGeom g;
    memset(&g, 0, sizeof(g));
    g.nlevels = p.nlevels;
    g.w = w;
    g.h = h;
    g.iniTh = std::min(std::max(p.ini_th_fast, 0), 255);
    g.minTh = std::min(std::max(p.min_th_fast, 0), 255);
    g.quirks = p.quirks;
    for (int off = 0; off < 4; off++)
        for (int r = 0; r < 31; r++)
            for (int qi = 0; qi < 9; qi++) {
                const int v = r - 15, lim = c->umax[v < 0 ? -v : v] & 15, u0 = 4 * (qi + 1) - off - 21;
                uint32_t ones = 0u, weights = 0u;
                for (int i = 0; i < 4; i++) {
                    const int u = u0 + i;
                    if (u < -lim || u > lim) continue;
                    ones |= 1u << (8 * i);
                    weights |= (uint32_t)(u + 32) << (8 * i);
                }
                g.icMask[off][r][qi][0] = ones;
                g.icMask[off][r][qi][1] = weights;
            }
    std::vector<OrbAreaTap> at;
    std::vector<OrbLinTap> lt;
    int64_t pyrOff = 0;
    int cellBase = 0, outBase = 0, cellCap = 1;
    int fastCells[2] = {0, 0};
    int64_t keyBase = 0, nodeBase = 0;
    for (int l = 0; l < p.nlevels; l++) {
        OrbLevel& L = g.lv[l];
        const float sc = c->invScale[l];
        L.w = cv_round((float)w * sc);
        L.h = cv_round((float)h * sc);
        L.pitch = (L.w + 63) & ~63;
        L.frameStride = (int64_t)L.pitch * L.h;
        L.off = pyrOff;  // multiplied by the batch size at launch time (levels are stored level-major)
        if (l > 0) pyrOff += L.frameStride;
        L.maxBorderX = L.w - ORB_EDGE + 3;
        L.maxBorderY = L.h - ORB_EDGE + 3;
        const float width = (float)(L.maxBorderX - ORB_BORDER), height = (float)(L.maxBorderY - ORB_BORDER);
        L.nCols = (int)(width / 30.f);
        L.nRows = (int)(height / 30.f);
        if (L.nCols < 1 || L.nRows < 1) {
            orb_set_error("level %d of a %dx%d frame is %dx%d: smaller than one 30-pixel FAST cell plus borders", l, w,
                          h, L.w, L.h);
            return ORB_ERR_UNSUPPORTED;
        }
        L.nColsMagic = orb_div_magic((uint32_t)L.nCols, (uint64_t)L.nCols * L.nRows);
        L.wCell = (int)ceilf(width / L.nCols);
        L.hCell = (int)ceilf(height / L.nRows);
        if (L.wCell > 59 || L.hCell > 59) {
            orb_set_error("internal: cell %dx%d exceeds the kernel tile", L.wCell, L.hCell);
            return ORB_ERR_UNSUPPORTED;
        }
        L.cellBase = cellBase;
        cellBase += L.nCols * L.nRows;
        L.nIni = (int)lroundf((float)(L.maxBorderX - ORB_BORDER) / (L.maxBorderY - ORB_BORDER));
        if (L.nIni < 1 || L.nIni > 16) {
            orb_set_error("aspect ratio of level %d (%dx%d) gives %d quadtree roots; supported 1..16", l, L.w, L.h,
                          L.nIni);
            return ORB_ERR_UNSUPPORTED;
        }
        L.hX = (float)(L.maxBorderX - ORB_BORDER) / L.nIni;
        L.quota = c->perLevel[l];
        L.outCap = std::max(L.quota + 3, 4 * L.nIni);
        L.outBase = outBase;
        outBase += L.outCap;
        L.nodeCap = L.outCap + 8;
        L.nodeBase = nodeBase;
        nodeBase += L.nodeCap;
        // most keys FAST can emit: strict local maxima take at most one pixel of every 2x2 block of a cell interior
        int64_t keys = 0;
        int levelCellW = 0, levelCellH = 0;
        for (int i = 0; i < L.nRows; i++) {
            const int iniY = ORB_BORDER + i * L.hCell;
            if (iniY >= L.maxBorderY - 3) continue;
            const int maxY = std::min(iniY + L.hCell + 6, L.maxBorderY);
            const int ch = maxY - 3 - (iniY + 3);
            if (ch <= 0) continue;
            for (int j = 0; j < L.nCols; j++) {
                const int iniX = ORB_BORDER + j * L.wCell;
                if (iniX >= L.maxBorderX - 6) continue;
                const int maxX = std::min(iniX + L.wCell + 6, L.maxBorderX);
                const int cw = maxX - 3 - (iniX + 3);
                if (cw <= 0) continue;
                const int cap = halfUp(cw) * halfUp(ch);
                g.maxCellW = std::max(g.maxCellW, cw);
                g.maxCellH = std::max(g.maxCellH, ch);
                levelCellW = std::max(levelCellW, cw);
                levelCellH = std::max(levelCellH, ch);
                keys += cap;
                cellCap = std::max(cellCap, cap);
            }
        }
        if (keys > 0x7fffffff) {
            orb_set_error("frame too large");
            return ORB_ERR_UNSUPPORTED;
        }
        // the FAST kernel's two instances: a level whose cells fit the small tile runs at twice the occupancy
        L.fastClass = (levelCellW <= ORB_FAST_SMALL_W && levelCellH <= ORB_FAST_SMALL_H) ? 0 : 1;
        L.classBase = fastCells[L.fastClass];
        fastCells[L.fastClass] += L.nCols * L.nRows;
        L.keyCap = (int)keys;
        L.keyBase = keyBase;
        keyBase += keys;
        L.scale = c->scale[l];
        L.patchSize = (float)(int)(31 * c->scale[l]);  // :881, int * float truncated
        L.tabX = L.tabY = 0;
        if (l > 0) {
            const OrbLevel& P = g.lv[l - 1];
            if (p.quirks & ORB_Q_RESIZE_AREA) {
                L.tabX = (int)at.size();
                int rc = area_taps(P.w, L.w, at);
                if (rc) return rc;
                L.tabY = (int)at.size();
                rc = area_taps(P.h, L.h, at);
                if (rc) return rc;
                if ((rc = check_tile_spans(P.w, L.w, ORB_RS_TW, &at[L.tabX], nullptr))) return rc;
                if ((rc = check_tile_spans(P.h, L.h, ORB_RS_TH, &at[L.tabY], nullptr))) return rc;
                L.rsTpExact = exact_tile_extent(L.w, ORB_RS_TW, true, &at[L.tabX], nullptr);
                L.rsRows8 = exact_tile_extent(L.h, 8, false, &at[L.tabY], nullptr);
            } else {
                if (P.w == 2 * L.w && P.h == 2 * L.h) {
                    orb_set_error("exact 2:1 level ratio: cv::resize(INTER_LINEAR) switches to INTER_AREA there");
                    return ORB_ERR_UNSUPPORTED;
                }
                L.tabX = (int)lt.size();
                linear_taps(P.w, L.w, true, lt);
                L.tabY = (int)lt.size();
                linear_taps(P.h, L.h, false, lt);
                int rc;
                if ((rc = check_tile_spans(P.w, L.w, ORB_RS_TW, nullptr, &lt[L.tabX]))) return rc;
                if ((rc = check_tile_spans(P.h, L.h, ORB_RS_TH, nullptr, &lt[L.tabY]))) return rc;
                L.rsTpExact = exact_tile_extent(L.w, ORB_RS_TW, true, nullptr, &lt[L.tabX]);
                L.rsRows8 = exact_tile_extent(L.h, 8, false, nullptr, &lt[L.tabY]);
            }
        }
    }
    // quadtree path digits must reach single pixels: 2^ORB_CODE_DEPTH-1 >= larger border-relative extent
    if (std::max(w, h) - 2 * ORB_BORDER > (1 << (ORB_CODE_DEPTH - 1))) {
        orb_set_error("frame side %d exceeds the %d-pixel limit of the quadtree path code", std::max(w, h),
                      (1 << (ORB_CODE_DEPTH - 1)) + 2 * ORB_BORDER);
        return ORB_ERR_UNSUPPORTED;
    }
    // a cell's slots start 16-byte aligned and number at least 16: the quadtree's counting pass reads the first 16
    // of every cell as four-word loads
    cellCap = std::max(16, (cellCap + 3) & ~3);
    for (int l = 0; l < g.nlevels; l++) {
        if ((int64_t)g.lv[l].nCols * g.lv[l].nRows * cellCap > (int64_t)1 << 24) {
            orb_set_error("frame too large: level %d needs more than 2^24 key ranks", l);
            return ORB_ERR_UNSUPPORTED;
        }
    }
    // quadtree path words: per level one word per column and one per row of the area FAST keys lie in
    // (coordinates relative to the 16-pixel border, src/ORBextractor.cc:864-865)
    std::vector<uint32_t> ct;
    for (int l = 0; l < g.nlevels; l++) {
        OrbLevel& L = g.lv[l];
        const int width = L.maxBorderX - ORB_BORDER, height = L.maxBorderY - ORB_BORDER;
        L.codeTabX = (int)ct.size();
        for (int x = 0; x < width; x++) ct.push_back(qt_code_x(x, L.nIni, L.hX));
        L.codeTabY = (int)ct.size();
        for (int y = 0; y < height; y++) ct.push_back(qt_code_y(y, height));
        // buckets of the quadtree's histogram form: 4096 at most where several workgroups share a CU's LDS,
        // 16384 for the levels of large frames, whose workgroups take a CU for themselves
        L.histDepth = qt_hist_depth(L.nIni, L.quota, g.lv[0].keyCap > ORB_QT_BIG_KEYCAP ? 16384 : 4096);
        L.histDepthFirst = std::min(L.histDepth, qt_hist_depth(L.nIni, L.quota, 4096, 1));
    }
    g.cellsPerFrame = cellBase;
    g.cellCap = cellCap;
    g.outCapFrame = outBase;
    // dividends: workgroup number / 8 (frames go to the eight XCD lanes in turn), key rank within a level
    const uint64_t framesPerLane = ((uint64_t)std::max(p.max_batch, 1) + ORB_XCDS - 1) / ORB_XCDS;
    uint64_t maxRank = 0;
    for (int l = 0; l < g.nlevels; l++)
        maxRank = std::max<uint64_t>(maxRank, (uint64_t)g.lv[l].nCols * g.lv[l].nRows * cellCap);
    g.cellsPerFrameMagic = orb_div_magic((uint32_t)cellBase, framesPerLane * cellBase);
    for (int cls = 0; cls < 2; cls++) {
        g.fastCells[cls] = fastCells[cls];
        g.fastCellsMagic[cls] = orb_div_magic((uint32_t)fastCells[cls], framesPerLane * fastCells[cls]);
        g.fastFirst[cls] = -1;
        for (int l = g.nlevels - 1; l >= 0; l--)
            if (g.lv[l].fastClass == cls) g.fastFirst[cls] = l;
    }
    g.outCapFrameMagic = orb_div_magic((uint32_t)outBase, framesPerLane * outBase);
    g.cellCapMagic = orb_div_magic((uint32_t)cellCap, maxRank);
    g.keysPerFrame = keyBase;
    g.nodesPerFrame = nodeBase;
    g.areaTapsMax = 1;
    for (const OrbAreaTap& t : at)
        for (int j = 1; j < ORB_MAX_TAPS; j++)
            if (t.a[j] != 0.f) g.areaTapsMax = std::max(g.areaTapsMax, j + 1);
    build_chain(g, at, lt);
    c->geom = g;
    c->areaTaps.swap(at);
    c->linTaps.swap(lt);
    c->codeTab.swap(ct);
    c->geomValid = true;
    c->lastFrames = 0;  // per-batch level offsets must be rebuilt
    int rc;
    if ((rc = c->dGeom.ensure(sizeof(OrbGeom)))) return rc;
    if ((rc = c->dAreaTaps.ensure(std::max<size_t>(1, c->areaTaps.size()) * sizeof(OrbAreaTap)))) return rc;
    if ((rc = c->dLinTaps.ensure(std::max<size_t>(1, c->linTaps.size()) * sizeof(OrbLinTap)))) return rc;
    if ((rc = c->dCodeTab.ensure(c->codeTab.size() * sizeof(uint32_t)))) return rc;
    // (synchronous copies: geometry only changes when the frame size does)
    HIP_TRY(hipDeviceSynchronize());
    if (!c->areaTaps.empty())
        HIP_TRY(hipMemcpy(c->dAreaTaps.p, c->areaTaps.data(), c->areaTaps.size() * sizeof(OrbAreaTap),
                          hipMemcpyHostToDevice));
    if (!c->linTaps.empty())
        HIP_TRY(hipMemcpy(c->dLinTaps.p, c->linTaps.data(), c->linTaps.size() * sizeof(OrbLinTap),
                          hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->dCodeTab.p, c->codeTab.data(), c->codeTab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return ORB_OK;
}

// Pyramid levels are stored level-major: all frames of level 1, then level 2, ...
// so the per-level offsets depend on the batch size; the device copy of the
// geometry is refreshed when either changes.
static int finalize_geom_for_batch(orb_ctx* c, int nframes) {
    OrbGeom& g = c->geom;
    if (c->lastFrames == nframes) return ORB_OK;
    int64_t off = 0;
    for (int l = 1; l < g.nlevels; l++) {
        g.lv[l].off = off;
        off += g.lv[l].frameStride * nframes;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(c->dGeom.p, &g, sizeof(OrbGeom), hipMemcpyHostToDevice));
    c->lastFrames = nframes;
    return ORB_OK;
}

static int ensure_buffers(orb_ctx* c, int n) {
    const OrbGeom& g = c->geom;
    int rc;
    int64_t pyr = 0;
    for (int l = 1; l < g.nlevels; l++) pyr += g.lv[l].frameStride * n;
    if ((rc = c->dPyr.ensure((size_t)std::max<int64_t>(pyr, 64)))) return rc;
    if ((rc = c->dCellKeys.ensure((size_t)n * g.cellsPerFrame * g.cellCap * 4))) return rc;
    if ((rc = c->dCellCount.ensure((size_t)n * g.cellsPerFrame * 4))) return rc;
    if ((rc = c->dSort.ensure((size_t)n * g.keysPerFrame * 16 + 16))) return rc;
    if ((rc = c->dNodes.ensure((size_t)n * g.nodesPerFrame * 2 * sizeof(OrbNode)))) return rc;
    if ((rc = c->dInts.ensure(((size_t)n * g.nodesPerFrame * 9 + (size_t)n * g.nlevels * 4) * 4))) return rc;
    if ((rc = c->dSelXY.ensure((size_t)n * g.outCapFrame * 4))) return rc;
    if ((rc = c->dSelScore.ensure((size_t)n * g.outCapFrame * 4))) return rc;
    if ((rc = c->dLevelCount.ensure((size_t)n * g.nlevels * 4))) return rc;
    if ((rc = c->dLevelKeys.ensure((size_t)n * g.nlevels * 4))) return rc;
    {   // the quadtree's second-pass lists: three words (count, workgroups done, last count) and the entries per frame
        // group; the kernels leave count and done at zero, so the words only need clearing when the block is made
        const size_t need = ((size_t)n * (g.nlevels + 1) + 4) * 4;
        const void* before = c->dQtRetry.p;
        if ((rc = c->dQtRetry.ensure(need))) return rc;
        if (c->dQtRetry.p != before) HIP_TRY(hipMemset(c->dQtRetry.p, 0, c->dQtRetry.bytes));
    }
    if (!c->dHostOut.p) {
        const size_t frames = (size_t)std::max(n, c->prm.max_batch), cap = (size_t)orb_max_keypoints_per_frame(c);
        if ((rc = c->dHostOut.ensure(64 + 2 * ((frames * 4 + 63) & ~(size_t)63) + frames * cap * (sizeof(orb_keypoint) + 32)))) return rc;
        HIP_TRY(hipMemset(c->dHostOut.p, 0, 64));
    }
    return ORB_OK;
}

extern "C" int orb_create(const orb_params* prm, orb_ctx** out) {
    if (!prm || !out) {
        orb_set_error("orb_create: null argument");
        return ORB_ERR_INVALID;
    }
    *out = nullptr;
    if (prm->nlevels < 1 || prm->nlevels > ORB_MAX_LEVELS) {
        orb_set_error("nlevels %d outside 1..%d", prm->nlevels, ORB_MAX_LEVELS);
        return ORB_ERR_UNSUPPORTED;
    }
    if (prm->nfeatures < 1 || !(prm->scale_factor > 1.0f) || prm->max_batch < 1 || prm->max_width < 1 ||
        prm->max_height < 1) {
        orb_set_error("orb_create: nfeatures, max sizes and max_batch must be >= 1, scale_factor > 1");
        return ORB_ERR_INVALID;
    }
    if (prm->min_th_fast < 1 || prm->ini_th_fast < prm->min_th_fast || prm->ini_th_fast > 255) {
        orb_set_error("FAST thresholds must satisfy 1 <= minThFAST <= iniThFAST <= 255");
        return ORB_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (prm->device < 0 || prm->device >= ndev) {
        orb_set_error("device %d not present (%d devices)", prm->device, ndev);
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(prm->device));
    orb_ctx* c = new (std::nothrow) orb_ctx();
    if (!c) return ORB_ERR_NOMEM;
    c->prm = *prm;
    build_tables(c);
    // a failing HIP call from here on must not leak the half-built context
#define CREATE_TRY(call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            orb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            orb_destroy(c);                                                                           \
            return ORB_ERR_HIP;                                                                       \
        }                                                                                             \
    } while (0)
    CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // one extra stream per frame group beyond the first: a process has few hardware queues (four by default),
    // and streams that share one serialise, so a context takes no more than it will use
    if (const char* e = getenv("ORBHIP_PARTS")) c->parts = c->hostParts = std::max(1, std::min(ORB_MAX_PARTS, atoi(e)));
    if (const char* e = getenv("ORBHIP_HOST_PARTS")) c->hostParts = std::max(1, std::min(ORB_MAX_PARTS, atoi(e)));
    if (const char* e = getenv("ORBHIP_STAGE_THREADS")) c->stageThreads = std::max(1, std::min(ORB_STAGE_THREADS, atoi(e)));
    for (int i = 0; i < std::max(c->parts, c->hostParts) - 1; i++) {
        CREATE_TRY(hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking));
        CREATE_TRY(hipEventCreateWithFlags(&c->evJoin[i], hipEventDisableTiming));
    }
    CREATE_TRY(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
    for (int i = 1; i < ORB_MAX_PARTS; i++) CREATE_TRY(hipEventCreateWithFlags(&c->evIn[i], hipEventDisableTiming));
#undef CREATE_TRY
    // size everything for the largest announced frame and batch now, so steady-state calls allocate nothing
    int rc = compute_geom(c, prm->max_width, prm->max_height);
    if (rc == ORB_OK) rc = ensure_buffers(c, prm->max_batch);
    if (rc == ORB_OK) {
        c->inputStride = (prm->max_width + 63) & ~63;
        rc = c->dInput.ensure((size_t)prm->max_batch * c->inputStride * prm->max_height);
    }
    if (rc != ORB_OK) {
        orb_destroy(c);
        return rc;
    }
    *out = c;
    return ORB_OK;
}

extern "C" void orb_destroy(orb_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->prm.device);
    (void)hipDeviceSynchronize();
    DevBuf* bufs[] = {&c->dGeom, &c->dAreaTaps, &c->dLinTaps, &c->dCodeTab, &c->dInput, &c->dPyr, &c->dCellKeys, &c->dCellCount,
                      &c->dSort, &c->dNodes, &c->dInts, &c->dSelXY, &c->dSelScore,
                      &c->dLevelCount, &c->dLevelKeys, &c->dQtRetry, &c->dHostOut, &c->dTmpA,
                      &c->dTmpB};
    for (DevBuf* b : bufs) b->release();
    c->pinIn.release();
    c->pinOut.release();
    delete c->pool;
    c->pool = nullptr;
    if (c->matchState) orb_match_state_free(c->matchState);
    if (c->bowState) orb_bow_state_free(c->bowState);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    for (int i = 0; i < ORB_MAX_PARTS - 1; i++) {
        if (c->evJoin[i]) (void)hipEventDestroy(c->evJoin[i]);
        if (c->aux[i]) (void)hipStreamDestroy(c->aux[i]);
    }
    if (c->evFork) (void)hipEventDestroy(c->evFork);
    for (hipEvent_t e : c->evIn)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int orb_max_keypoints_per_frame(const orb_ctx* c) {
    if (!c) return 0;
    // independent of the frame size except through the number of quadtree roots (at most 16)
    int cap = 0;
    for (int l = 0; l < c->prm.nlevels; l++) cap += std::max(c->perLevel[l] + 3, 4 * 16);
    return cap;
}

extern "C" int orb_get_tables(const orb_ctx* c, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                              int* per_level, int* umax) {
    if (!c) {
        orb_set_error("null context");
        return ORB_ERR_INVALID;
    }
    for (int i = 0; i < c->prm.nlevels; i++) {
        if (scale) scale[i] = c->scale[i];
        if (inv_scale) inv_scale[i] = c->invScale[i];
        if (sigma2) sigma2[i] = c->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = c->invSigma2[i];
        if (per_level) per_level[i] = c->perLevel[i];
    }
    if (umax)
        for (int i = 0; i < 16; i++) umax[i] = c->umax[i];
    return ORB_OK;
}

extern "C" int orb_compute_tables(const orb_params* prm, float* scale, float* inv_scale, float* sigma2,
                                  float* inv_sigma2, int* per_level, int* umax) {
    if (!prm || prm->nlevels < 1 || prm->nlevels > ORB_MAX_LEVELS || !(prm->scale_factor > 1.f) || prm->nfeatures < 1) {
        orb_set_error("orb_compute_tables: bad parameters");
        return ORB_ERR_INVALID;
    }
    orb_ctx tmp;  // host members only; no device call is made
    tmp.prm = *prm;
    build_tables(&tmp);
    return orb_get_tables(&tmp, scale, inv_scale, sigma2, inv_sigma2, per_level, umax);
}

// Frame groups a batch of n frames is enqueued as (group i = frames [n i / parts, n (i + 1) / parts)).
static int batch_parts(const orb_ctx* c, int n) { return c->timing ? 1 : std::max(1, std::min(c->parts, n / ORB_MIN_PART_FRAMES)); }

// Enqueue the whole extraction for n frames whose level 0 is at d_img0.  groupInput, when given, holds per frame group
// i >= 1 an event after which that group's frames are in place (the first group's are by the time the stream reaches
// its kernels): those groups then wait for their events instead of for the stream's earlier work.
static int run_batch(orb_ctx* c, const uint8_t* d_img0, int n, int w, int h, int stride, int64_t frameStride,
                     orb_keypoint* d_kps, uint8_t* d_desc, int cap, int* d_nOut, hipStream_t st, int* d_status = nullptr,
                     const hipEvent_t* groupInput = nullptr, const int* groupStart = nullptr, int ngroups = 0) {
    if (n < 1 || n > c->prm.max_batch) {
        orb_set_error("batch of %d frames outside 1..max_batch=%d", n, c->prm.max_batch);
        return ORB_ERR_INVALID;
    }
    if (w > c->prm.max_width || h > c->prm.max_height) {
        orb_set_error("frame %dx%d larger than the context's %dx%d", w, h, c->prm.max_width, c->prm.max_height);
        return ORB_ERR_INVALID;
    }
    if (stride < w || cap < 1) {
        orb_set_error("bad stride or capacity");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->prm.device));
    int rc;
    if ((rc = compute_geom(c, w, h))) return rc;
    if ((rc = ensure_buffers(c, n))) return rc;
    if ((rc = finalize_geom_for_batch(c, n))) return rc;
    c->lastImg0 = d_img0;
    c->lastStride0 = stride;
    c->lastFrameStride0 = frameStride;
    // rows that do not start on 4-byte boundaries: one copy into the context's aligned input block (sized at
    // orb_create for the largest batch), which every stage then reads with whole dwords
    if (((unsigned long long)d_img0 | (unsigned long long)stride | (unsigned long long)frameStride) & 3ull) {
        if ((int64_t)n * h * ((w + 15) / 16) > 0x7fffffffll * 128) {
            orb_set_error("batch too large for the row realignment");
            return ORB_ERR_UNSUPPORTED;
        }
        if ((rc = orb_launch_realign(d_img0, w, h, stride, frameStride, (uint8_t*)c->dInput.p, c->inputStride, n, st))) return rc;
        d_img0 = (const uint8_t*)c->dInput.p;
        stride = c->inputStride;
        frameStride = (int64_t)c->inputStride * h;
    }
    OrbBatch b;
    memset(&b, 0, sizeof(b));
    b.d_geom = (const OrbGeom*)c->dGeom.p;
    b.h_geom = &c->geom;
    b.d_img0 = d_img0;
    b.stride0 = stride;
    b.frameStride0 = frameStride;
    b.frame0 = 0;
    b.nframes = n;
    b.d_pyr = (uint8_t*)c->dPyr.p;
    b.d_areaTaps = (const OrbAreaTap*)c->dAreaTaps.p;
    b.d_linTaps = (const OrbLinTap*)c->dLinTaps.p;
    b.d_codeTab = (const uint32_t*)c->dCodeTab.p;
    b.d_cellKeys = (uint32_t*)c->dCellKeys.p;
    b.d_cellCount = (int*)c->dCellCount.p;
    b.d_sort = (unsigned long long*)c->dSort.p;
    b.d_nodes = (OrbNode*)c->dNodes.p;
    b.d_ints = (int*)c->dInts.p;
    b.d_selXY = (uint32_t*)c->dSelXY.p;
    b.d_selScore = (int*)c->dSelScore.p;
    b.d_levelCount = (int*)c->dLevelCount.p;
    b.d_levelKeys = (int*)c->dLevelKeys.p;
    b.d_qtRetry = (int*)c->dQtRetry.p;
    b.d_error = c->dErr();
    b.d_status = d_status;
    b.d_kps = d_kps;
    b.d_desc = d_desc;
    b.capPerFrame = cap;
    b.d_nOut = d_nOut;
    b.stream = st;
    if (c->timing) {  // stage timing: the whole batch as one group, stages back to back on the caller's stream
        hipEvent_t* ev = &c->events[(size_t)(c->evCount % ORB_EV_RING) * 5];
        c->evCount++;
        c->lastGroups = orb_quadtree_two_passes(c->geom, n) ? 1 : 0;
        c->lastGroupFrame0[0] = 0;
        HIP_TRY(hipEventRecord(ev[0], st));
        if ((rc = orb_launch_pyramid(b))) return rc;
        HIP_TRY(hipEventRecord(ev[1], st));
        if ((rc = orb_launch_fast(b))) return rc;
        HIP_TRY(hipEventRecord(ev[2], st));
        if ((rc = orb_launch_quadtree(b))) return rc;
        HIP_TRY(hipEventRecord(ev[3], st));
        if ((rc = orb_launch_describe(b))) return rc;
        HIP_TRY(hipEventRecord(ev[4], st));
        return ORB_OK;
    }
    // groups: the caller's (groupStart[0 .. ngroups], first 0, last n) or equal shares
    const int parts = groupStart ? ngroups : batch_parts(c, n);
    if (parts > 1 && !groupInput) HIP_TRY(hipEventRecord(c->evFork, st));
    // stage by stage across the groups, so that the queues hold comparable work at any time
    OrbBatch pb[ORB_MAX_PARTS];
    for (int i = 0; i < parts; i++) {
        pb[i] = b;
        pb[i].frame0 = (int)((int64_t)n * i / parts);
        pb[i].nframes = (int)((int64_t)n * (i + 1) / parts) - pb[i].frame0;
        if (groupStart) {
            pb[i].frame0 = groupStart[i];
            pb[i].nframes = groupStart[i + 1] - groupStart[i];
        }
        if (i > 0) {
            pb[i].stream = c->aux[i - 1];
            HIP_TRY(hipStreamWaitEvent(c->aux[i - 1], groupInput ? groupInput[i] : c->evFork, 0));
        }
    }
    c->lastGroups = 0;
    for (int i = 0; i < parts; i++)
        if (orb_quadtree_two_passes(c->geom, pb[i].nframes)) c->lastGroupFrame0[c->lastGroups++] = pb[i].frame0;
    for (int i = 0; i < parts; i++)
        if ((rc = orb_launch_pyramid(pb[i]))) return rc;
    for (int i = 0; i < parts; i++)
        if ((rc = orb_launch_fast(pb[i]))) return rc;
    for (int i = 0; i < parts; i++)
        if ((rc = orb_launch_quadtree(pb[i]))) return rc;
    for (int i = 0; i < parts; i++)
        if ((rc = orb_launch_describe(pb[i]))) return rc;
    for (int i = 1; i < parts; i++) {
        HIP_TRY(hipEventRecord(c->evJoin[i - 1], c->aux[i - 1]));
        HIP_TRY(hipStreamWaitEvent(st, c->evJoin[i - 1], 0));
    }
    return ORB_OK;
}

int orb_match_check_error(orb_ctx* c);  // kernels_match.hip

static int read_error_flag(orb_ctx* c) {
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, c->dErr(), 4, hipMemcpyDeviceToHost));
    if (flag) {
        HIP_TRY(hipMemset(c->dErr(), 0, 4));
        orb_set_error("extraction hit an internal limit (flags 0x%x:%s%s%s%s%s)", flag,
                      (flag & 4) ? " key capacity" : "", (flag & 8) ? " node pool" : "",
                      (flag & 16) ? " quadtree depth" : "", (flag & 32) ? " level output" : "",
                      (flag & 64) ? " caller keypoint capacity" : "");
        return ORB_ERR_CAPACITY;
    }
    return ORB_OK;
}

extern "C" int orb_enable_stage_timing(orb_ctx* c, int on) {
    if (!c) {
        orb_set_error("null context");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->prm.device));
    if (on && c->events.empty()) {
        c->events.resize((size_t)ORB_EV_RING * 5);
        for (hipEvent_t& e : c->events) HIP_TRY(hipEventCreate(&e));
    }
    c->timing = on != 0;
    c->evCount = 0;
    return ORB_OK;
}

extern "C" int orb_get_stage_timings(orb_ctx* c, float ms[ORB_NUM_STAGES], int* nbatches) {
    if (!c || !ms || !nbatches) {
        orb_set_error("orb_get_stage_timings: null argument");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->prm.device));
    HIP_TRY(hipDeviceSynchronize());
    const int n = std::min(c->evCount, ORB_EV_RING);
    for (int s = 0; s < ORB_NUM_STAGES; s++) ms[s] = 0.f;
    for (int i = 0; i < n; i++) {
        hipEvent_t* ev = &c->events[(size_t)i * 5];
        for (int s = 0; s < ORB_NUM_STAGES; s++) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, ev[s], ev[s + 1]));
            ms[s] += t;
        }
    }
    if (n > 0)
        for (int s = 0; s < ORB_NUM_STAGES; s++) ms[s] /= n;
    *nbatches = n;
    c->evCount = 0;
    return ORB_OK;
}

extern "C" int orb_check_last_batch(orb_ctx* c) {
    if (!c) {
        orb_set_error("null context");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->prm.device));
    HIP_TRY(hipDeviceSynchronize());
    const int rc = read_error_flag(c);
    const int rcMatch = orb_match_check_error(c);  // asynchronous matcher calls report through the same function
    return rc ? rc : rcMatch;
}

extern "C" int orb_quadtree_second_pass_levels(orb_ctx* c, int* levels) {
    if (!c || !levels) {
        orb_set_error("orb_quadtree_second_pass_levels: null argument");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->prm.device));
    HIP_TRY(hipDeviceSynchronize());
    int total = 0;
    for (int i = 0; i < c->lastGroups; i++) {
        int n = 0;
        HIP_TRY(hipMemcpy(&n, (const int*)c->dQtRetry.p + (int64_t)c->lastGroupFrame0[i] * (c->geom.nlevels + 1) + 2, sizeof(int),
                          hipMemcpyDeviceToHost));  // (the count of the last launch, kept by the second pass)
        total += n;
    }
    *levels = total;
    return ORB_OK;
}

extern "C" int orb_extract_batch_device(orb_ctx* c, const uint8_t* d_imgs, int n, int w, int h, int stride,
                                        int64_t frame_stride, orb_keypoint* d_kps, uint8_t* d_desc, int cap_per_frame,
                                        int* d_n_out, void* stream) {
    if (!c || !d_imgs || !d_kps || !d_desc || !d_n_out) {
        orb_set_error("orb_extract_batch_device: null argument");
        return ORB_ERR_INVALID;
    }
    if (w <= 0 || h <= 0) {
        orb_set_error("empty frames are not accepted by the device entry point");
        return ORB_ERR_INVALID;
    }
    return run_batch(c, d_imgs, n, w, h, stride, frame_stride, d_kps, d_desc, cap_per_frame, d_n_out,
                     (hipStream_t)stream);
}

// Are the n frames (each `bytes` long) in page-locked host memory (hipHostMalloc, hipHostRegister, orb_alloc_host)?
// The DMA engine can then fetch them where they are.  One range query covers every frame of the same allocation.
static bool frames_page_locked(const uint8_t* const* imgs, int n, size_t bytes) {
    const uint8_t *lo = nullptr, *hi = nullptr;  // the page-locked range found last
    for (int f = 0; f < n; f++) {
        const uint8_t* p = imgs[f];
        if (lo && p >= lo && p + bytes <= hi) continue;
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) {
            (void)hipGetLastError();  // (an ordinary pointer is reported as an error: not one of ours)
            return false;
        }
        if (a.type != hipMemoryTypeHost) return false;
        void* base = nullptr;
        size_t size = 0;
        if (hipPointerGetAttribute(&base, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)p) != hipSuccess ||
            hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, (hipDeviceptr_t)p) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        lo = (const uint8_t*)base;
        hi = lo + size;
        if (!(p >= lo && p + bytes <= hi)) return false;
    }
    return true;
}

// First half of a host-pointer batch: the frames' way to the device, the extraction and the copy of the result block
// are enqueued on the context's stream; nothing is waited for.  pinnedHint: 1 / 0 when the caller has already
// looked where the frames live, -1 to find out here.
int orb_launch_undistort_frames(const float K[4], const float dist[5], const orb_keypoint* d_in, orb_keypoint* d_out,
                                const int* d_count, int cap, int nframes, void* stream);  // kernels_frame.hip
static int extract_batch_begin(orb_ctx* c, const uint8_t* const* imgs, int n, int w, int h, int stride, bool wantKps,
                               bool wantDesc, int pinnedHint, const float* camK = nullptr, const float* camDist = nullptr) {
    if (n < 1 || n > c->prm.max_batch || w > c->prm.max_width || h > c->prm.max_height || stride < w) {
        orb_set_error("orb_extract_batch: batch %d / frame %dx%d stride %d outside the context's limits", n, w, h,
                      stride);
        return ORB_ERR_INVALID;
    }
    if (c->hostBatch.open) {
        orb_set_error("orb_extract_batch: the context's previous host-pointer batch has not been collected");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->prm.device));
    int rc = compute_geom(c, w, h);
    if (rc) return rc;
    const int cap = orb_max_keypoints_per_frame(c);
    const int dstride = c->inputStride;
    const int64_t fstride = (int64_t)dstride * h;
    for (int f = 0; f < n; f++)
        if (!imgs[f]) {
            orb_set_error("frame %d is null", f);
            return ORB_ERR_INVALID;
        }
    const size_t frameBytes = (size_t)stride * (h - 1) + w;
    const bool pinned = pinnedHint >= 0 ? pinnedHint != 0 : frames_page_locked(imgs, n, frameBytes);
    const bool few = n < ORB_STAGE_SLICE;
    static const bool trace = getenv("ORBHIP_HOST_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double T0 = now();
    const hipEvent_t* groupIn = nullptr;
    // Frame groups: the frames of the later groups travel first, group by group, each group starts extracting when
    // its own frames have arrived, and its results are unpacked on the host while the groups behind it still run.
    // What remains once the last frame has arrived is the first group's extraction.  Three equal groups by default
    // (64 page-locked frames: 0.70 ms against 0.735 with two; a smaller first group, ORBHIP_HOST_FIRST_PCT, gained
    // nothing: consecutive copies leave some 35 us between them and small groups are bound by their launches).
    const int parts = (few || c->timing) ? 1 : std::max(1, std::min(c->hostParts, n / ORB_MIN_PART_FRAMES));
    int gs[ORB_MAX_PARTS + 1] = {0};
    {
        static const int pct = getenv("ORBHIP_HOST_FIRST_PCT") ? atoi(getenv("ORBHIP_HOST_FIRST_PCT")) : 0;
        const int first = parts > 1 ? std::max(ORB_MIN_PART_FRAMES, std::min(n - ORB_MIN_PART_FRAMES * (parts - 1), pct > 0 ? n * pct / 100 : n / parts)) : n;
        for (int i = 1; i <= parts; i++) gs[i] = first + (int)((int64_t)(n - first) * (i - 1) / std::max(1, parts - 1));
        gs[parts] = n;
    }
    // A few frames (the reference's one-frame-per-call use) are fetched by a kernel that reads the page-locked memory
    // over the link: a DMA copy of one 640x480 frame takes 11 us and the first kernel behind it starts 12 us later,
    // the kernel takes 8 us and the next one follows at once.  ORBHIP_UPLOAD_DMA=1 keeps the copy engine.
    static const bool uploadDma = getenv("ORBHIP_UPLOAD_DMA") != nullptr;
    auto fetch_by_kernel = [&](const uint8_t* hostSrc, int srcStride, int64_t srcFrame, int f0, int count) -> int {
        uint8_t* dsrc = nullptr;  // the frames as the device addresses them
        HIP_TRY(hipHostGetDevicePointer((void**)&dsrc, (void*)hostSrc, 0));
        return orb_launch_realign(dsrc, w, h, srcStride, srcFrame, (uint8_t*)c->dInput.p + (size_t)f0 * fstride, dstride, count, c->stream);
    };
    if (pinned && few && !uploadDma) {
        for (int f = 0; f < n; f++)
            if ((rc = fetch_by_kernel(imgs[f], stride, 0, f, 1))) return rc;
    } else if (pinned) {
        // Page-locked frames go to the device from where they are: no staging copy on the host (which runs at the
        // host memory's rate, and at that rate for all contexts of a node together).  Frames that follow each other
        // at a fixed distance travel as one copy per frame group, the later groups' first (those groups extract while
        // the first group's frames are still on the link); frames scattered over the heap as one copy each.
        auto copy_run = [&](int f0, int f1) -> hipError_t {
            bool uniform = true;
            for (int f = f0; f + 1 < f1 && uniform; f++) uniform = imgs[f + 1] - imgs[f] == (ptrdiff_t)stride * h;
            uint8_t* dst = (uint8_t*)c->dInput.p + (size_t)f0 * fstride;
            if (uniform && stride == dstride)
                return hipMemcpyAsync(dst, imgs[f0], (size_t)(f1 - f0 - 1) * fstride + frameBytes, hipMemcpyHostToDevice, c->stream);
            if (uniform)
                return hipMemcpy2DAsync(dst, (size_t)dstride, imgs[f0], (size_t)stride, (size_t)w, (size_t)(f1 - f0) * h,
                                        hipMemcpyHostToDevice, c->stream);
            for (int f = f0; f < f1; f++) {
                uint8_t* df = (uint8_t*)c->dInput.p + (size_t)f * fstride;
                const hipError_t e = stride == dstride
                                         ? hipMemcpyAsync(df, imgs[f], frameBytes, hipMemcpyHostToDevice, c->stream)
                                         : hipMemcpy2DAsync(df, (size_t)dstride, imgs[f], (size_t)stride, (size_t)w, (size_t)h,
                                                            hipMemcpyHostToDevice, c->stream);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        };
        for (int i = 1; i < parts; i++) {
            HIP_TRY(copy_run(gs[i], gs[i + 1]));
            HIP_TRY(hipEventRecord(c->evIn[i], c->stream));
        }
        HIP_TRY(copy_run(0, parts > 1 ? gs[1] : n));
        if (parts > 1) groupIn = c->evIn;
    } else {
        // Pageable frames -> page-locked staging in the device layout (row copies on the host) -> device.  A single
        // host thread copies at a fraction of the link rate, so a batch is staged by up to four threads (the context
        // keeps them), slice by slice, and every slice is handed to the DMA engine as soon as it is staged (host
        // copies and DMA overlap).  The copies go to ONE stream, in a fixed order (copies on several streams at once
        // share the link badly: 17 instead of 56 GB/s), in slices of about 5 MB (consecutive copies on a stream leave
        // 18 us between them and cost the enqueuing thread 15 us each: 1.2 MB slices ran at 35 GB/s; a kernel that
        // reads the staging block over the link itself has no such gaps but is slower, 0.89 against 0.82 ms for 64
        // frames).  The frames of the later frame groups go first and an event marks their arrival: those groups
        // start extracting while the first group's frames, the last to arrive, are still on the link.
        if ((rc = c->pinIn.ensure((size_t)n * fstride))) return rc;
        auto stage_frames = [&](int f0, int f1) {
            for (int f = f0; f < f1; f++) {
                uint8_t* dst = (uint8_t*)c->pinIn.p + f * fstride;
                if (stride == dstride) {
                    memcpy(dst, imgs[f], frameBytes);
                } else {
                    for (int r = 0; r < h; r++) memcpy(dst + (size_t)r * dstride, imgs[f] + (size_t)r * stride, (size_t)w);
                }
            }
        };
        if (few) {
            // a few frames (the reference's one-frame-per-call use): one host thread, one copy
            stage_frames(0, n);
            if (uploadDma) HIP_TRY(hipMemcpyAsync(c->dInput.p, c->pinIn.p, (size_t)n * fstride, hipMemcpyHostToDevice, c->stream));
            else if ((rc = fetch_by_kernel((const uint8_t*)c->pinIn.p, dstride, fstride, 0, n))) return rc;
        } else {
            const int per = (int)std::max<int64_t>(1, std::min<int64_t>(n, ORB_STAGE_SLICE_BYTES / fstride));
            int sl0[ORB_MAX_STAGE_SLICES], sl1[ORB_MAX_STAGE_SLICES], slEv[ORB_MAX_STAGE_SLICES], nsl = 0;
            // (a slice holds at least n / 25 frames: the runs below, one per group and each ending in a shorter slice,
            // make at most 25 + ORB_MAX_PARTS slices)
            const int perCapped = std::max(per, (n + 24) / 25);
            for (int i = 1; i <= parts; i++) {  // groups 1 .. parts - 1, then group 0
                const int gi = i < parts ? i : 0, f0 = gs[gi], f1 = gs[gi + 1];
                for (int s0 = f0; s0 < f1; s0 += perCapped) {
                    sl0[nsl] = s0;
                    sl1[nsl] = std::min(f1, s0 + perCapped);
                    slEv[nsl++] = (gi > 0 && s0 + perCapped >= f1) ? gi : 0;  // the group's last slice records its event
                }
            }
            // every slice is staged by all the threads together, a share of its frames each; whoever finishes a slice's
            // last share hands the slice to the DMA engine.  A thread takes the slices in order, so the last share of
            // slice k is finished after its owner has enqueued every earlier slice it completed: copies are enqueued
            // in slice order without any waiting.
            const int nthreads = std::max(1, std::min(c->stageThreads, n / ORB_STAGE_SLICE));
            hipError_t terr[ORB_STAGE_THREADS];
            std::atomic<int> arrived[ORB_MAX_STAGE_SLICES];
            for (int k = 0; k < nsl; k++) arrived[k].store(0, std::memory_order_relaxed);
            const std::function<void(int)> stage = [&](int t) {
                terr[t] = t ? hipSetDevice(c->prm.device) : hipSuccess;
                for (int k = 0; k < nsl; k++) {
                    const int m = sl1[k] - sl0[k];
                    stage_frames(sl0[k] + m * t / nthreads, sl0[k] + m * (t + 1) / nthreads);
                    if (arrived[k].fetch_add(1, std::memory_order_acq_rel) + 1 < nthreads || terr[t] != hipSuccess) continue;
                    const size_t off = (size_t)sl0[k] * fstride;
                    terr[t] = hipMemcpyAsync((uint8_t*)c->dInput.p + off, (uint8_t*)c->pinIn.p + off, (size_t)m * fstride,
                                             hipMemcpyHostToDevice, c->stream);
                    if (terr[t] == hipSuccess && slEv[k]) terr[t] = hipEventRecord(c->evIn[slEv[k]], c->stream);
                }
            };
            if (nthreads > 1 && !c->pool) c->pool = new StagePool();
            if (nthreads > 1) c->pool->run(nthreads, stage);
            else stage(0);
            for (int t = 0; t < nthreads; t++) HIP_TRY(terr[t]);
            if (parts > 1) groupIn = c->evIn;
        }
    }
    // results: [flag words, one per frame | counts | keypoints (cap per frame) | descriptors] as one block in
    // page-locked host memory, which the last kernel writes over the link itself (a copy after the kernels, and the
    // gap before it, cost more than the 64 KB per frame take), with the flag word as the last stage saw it per
    // frame.  ORBHIP_HOST_RESULT_COPY=1 keeps the block on the device and fetches it with one copy instead.
    const double T1 = now();
    const size_t offCnt = std::max<size_t>(64, ((size_t)n * 4 + 63) & ~(size_t)63), offKps = offCnt + (((size_t)n * 4 + 63) & ~(size_t)63);
    const size_t offDesc = offKps + (size_t)n * cap * sizeof(orb_keypoint);
    // (with a camera: the undistorted keypoints behind the descriptors, made by one more kernel on the same stream)
    const bool undist = camK && camDist && camDist[0] != 0.0f;
    const size_t offUn = (offDesc + (size_t)n * cap * 32 + 63) & ~(size_t)63;
    const size_t total = undist ? offUn + (size_t)n * cap * sizeof(orb_keypoint) : offDesc + (size_t)n * cap * 32;
    if ((rc = c->pinOut.ensure(total))) return rc;
    uint8_t* po = (uint8_t*)c->pinOut.p;
    static const bool resultCopy = getenv("ORBHIP_HOST_RESULT_COPY") != nullptr;
    const bool direct = few || !resultCopy || undist;
    if (direct) {
        uint8_t* dpo = nullptr;  // the block as the device addresses it
        HIP_TRY(hipHostGetDevicePointer((void**)&dpo, po, 0));
        rc = run_batch(c, (const uint8_t*)c->dInput.p, n, w, h, dstride, fstride, (orb_keypoint*)(dpo + offKps), dpo + offDesc,
                       cap, (int*)(dpo + offCnt), c->stream, (int*)dpo, groupIn, parts > 1 ? gs : nullptr, parts);
        if (rc) return rc;
        if (undist && (rc = orb_launch_undistort_frames(camK, camDist, (const orb_keypoint*)(dpo + offKps), (orb_keypoint*)(dpo + offUn),
                                                        (const int*)(dpo + offCnt), cap, n, c->stream)))
            return rc;
    } else {
        if (c->dHostOut.bytes < total) {
            orb_set_error("internal: result block of %zu bytes beyond the context's", total);
            return ORB_ERR_CAPACITY;
        }
        uint8_t* dOut = (uint8_t*)c->dHostOut.p;
        rc = run_batch(c, (const uint8_t*)c->dInput.p, n, w, h, dstride, fstride, (orb_keypoint*)(dOut + offKps), dOut + offDesc,
                       cap, (int*)(dOut + offCnt), c->stream, nullptr, groupIn, parts > 1 ? gs : nullptr, parts);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(po, dOut, (wantKps || wantDesc) ? (wantDesc ? total : offDesc) : offKps, hipMemcpyDeviceToHost, c->stream));
    }
    c->hostBatch.t0 = T0;
    c->hostBatch.open = true;
    c->hostBatch.direct = direct;
    c->hostBatch.ngroups = (direct && !c->timing) ? parts : 1;
    for (int i = 0; i <= parts; i++) c->hostBatch.groupStart[i] = gs[i];
    c->hostBatch.n = n;
    c->hostBatch.cap = cap;
    c->hostBatch.offCnt = offCnt;
    c->hostBatch.offKps = offKps;
    c->hostBatch.offDesc = offDesc;
    c->hostBatch.total = total;
    c->hostBatch.offUn = undist ? offUn : 0;
    if (trace) fprintf(stderr, "host batch %d (%s frames): input enqueued after %.0f us, kernels and result copy after %.0f us\n", n, pinned ? "page-locked" : "pageable", T1 - T0, now() - T0);
    return ORB_OK;
}

// Second half: waits for the context's stream and unpacks the result block into the caller's arrays.
static int extract_batch_finish(orb_ctx* c, orb_keypoint* kps, uint8_t* desc, int cap_per_frame, int* n_out,
                                orb_keypoint* kpsUn = nullptr) {
    if (!c->hostBatch.open) {
        orb_set_error("orb_extract_batch: no batch to collect");
        return ORB_ERR_INVALID;
    }
    c->hostBatch.open = false;
    HIP_TRY(hipSetDevice(c->prm.device));
    const int n = c->hostBatch.n, cap = c->hostBatch.cap, ngroups = c->hostBatch.ngroups;
    const int* gs = c->hostBatch.groupStart;
    const size_t offCnt = c->hostBatch.offCnt, offKps = c->hostBatch.offKps, offDesc = c->hostBatch.offDesc;
    uint8_t* po = (uint8_t*)c->pinOut.p;
    auto unpack = [&](int f0, int f1) -> int {
        memcpy(n_out + f0, po + offCnt + (size_t)f0 * 4, (size_t)(f1 - f0) * 4);
        for (int f = f0; f < f1; f++) {
            const int m = n_out[f];
            if (m > cap_per_frame) {
                orb_set_error("frame %d produced %d keypoints, caller capacity is %d", f, m, cap_per_frame);
                return ORB_ERR_CAPACITY;
            }
            if (m > 0) {
                if (kps)
                    memcpy(kps + (size_t)f * cap_per_frame, po + offKps + (size_t)f * cap * sizeof(orb_keypoint),
                           (size_t)m * sizeof(orb_keypoint));
                if (desc) memcpy(desc + (size_t)f * cap_per_frame * 32, po + offDesc + (size_t)f * cap * 32, (size_t)m * 32);
                if (kpsUn)  // (no distortion: mvKeysUn = mvKeys)
                    memcpy(kpsUn + (size_t)f * cap_per_frame,
                           po + (c->hostBatch.offUn ? c->hostBatch.offUn : offKps) + (size_t)f * cap * sizeof(orb_keypoint),
                           (size_t)m * sizeof(orb_keypoint));
            }
        }
        return ORB_OK;
    };
    auto flagged = [&](int f0, int f1) {
        int any = 0;
        for (int f = f0; f < f1; f++) any |= ((const int*)po)[f];
        return any != 0;
    };
    static const bool trace = getenv("ORBHIP_HOST_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tA = 0, tB = 0, tC = 0;
    int rc = ORB_OK;
    // the later groups are done first (their frames arrived first), in turn: their results are unpacked while the
    // first group still extracts
    for (int i = 1; i < ngroups && rc == ORB_OK; i++) {
        HIP_TRY(hipEventSynchronize(c->evJoin[i - 1]));
        if (i == 1) tA = now();
        if (!flagged(gs[i], gs[i + 1])) rc = unpack(gs[i], gs[i + 1]);
        tB = now();
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    tC = now();
    if (c->hostBatch.direct ? flagged(0, n) : *(const int*)po != 0) return read_error_flag(c);
    if (rc) return rc;
    rc = unpack(0, ngroups > 1 ? gs[1] : n);
    if (trace)
        fprintf(stderr, "host batch %d: first of the later groups done at %.0f us, last of them unpacked at %.0f, stream done at %.0f, all unpacked at %.0f\n", n,
                tA - c->hostBatch.t0, tB - c->hostBatch.t0, tC - c->hostBatch.t0, now() - c->hostBatch.t0);
    return rc;
}

extern "C" int orb_extract_batch(orb_ctx* c, const uint8_t* const* imgs, int n, int w, int h, int stride,
                                 orb_keypoint* kps, uint8_t* desc, int cap_per_frame, int* n_out) {
    if (!c || !imgs || !n_out) {
        orb_set_error("orb_extract_batch: null argument");
        return ORB_ERR_INVALID;
    }
    if (w <= 0 || h <= 0) {  // empty image: silent return, src/ORBextractor.cc:451-452
        for (int f = 0; f < n; f++) n_out[f] = 0;
        return ORB_OK;
    }
    int rc = extract_batch_begin(c, imgs, n, w, h, stride, kps != nullptr, desc != nullptr, -1);
    if (rc) {
        // whatever was enqueued before the failure reads the staging blocks: let it finish before they are reused
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        return rc;
    }
    return extract_batch_finish(c, kps, desc, cap_per_frame, n_out);
}

// The extraction of one frame and the undistortion of its keypoints (the Frame constructor's ExtractORB +
// UndistortKeyPoints, src/Frame.cc:67-74, include/Frame.h:121) as one enqueue and one wait.
extern "C" int orb_extract_undistorted(orb_ctx* c, const uint8_t* img, int w, int h, int stride, const float K[4],
                                       const float dist[5], orb_keypoint* kps, orb_keypoint* kps_un, uint8_t* desc, int cap,
                                       int* n_out) {
    if (!c || !n_out || !K || !dist || !kps || !kps_un) {
        orb_set_error("orb_extract_undistorted: null argument");
        return ORB_ERR_INVALID;
    }
    if (!img || w <= 0 || h <= 0) {  // empty image: silent return, src/ORBextractor.cc:451-452
        *n_out = 0;
        return ORB_OK;
    }
    const uint8_t* one[1] = {img};
    int rc = extract_batch_begin(c, one, 1, w, h, stride, true, desc != nullptr, -1, K, dist);
    if (rc) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        return rc;
    }
    return extract_batch_finish(c, kps, desc, cap, n_out, kps_un);
}

// Frame blocks over several contexts (one per device), host pointers.  Page-locked frames: the calling thread
// enqueues every context's block (copies, kernels, result copy: a fraction of a millisecond of host time each) and
// then collects them in turn, so the devices work side by side without any thread being made.  Pageable frames have
// to be copied into staging by the host first, which is the slow part: one host thread per context.
int orb_extract_batch_over_contexts(orb_ctx* const* ctxs, int nctx, const uint8_t* const* imgs, int n, int w, int h, int stride,
                                    orb_keypoint* kps, uint8_t* desc, int cap_per_frame, int* n_out) {
    if (w <= 0 || h <= 0) {
        for (int f = 0; f < n; f++) n_out[f] = 0;
        return ORB_OK;
    }
    for (int f = 0; f < n; f++)
        if (!imgs[f]) {
            orb_set_error("frame %d is null", f);
            return ORB_ERR_INVALID;
        }
    auto lo = [&](int r) { return (int)((int64_t)n * r / nctx); };
    auto outK = [&](int f) { return kps ? kps + (size_t)f * cap_per_frame : nullptr; };
    auto outD = [&](int f) { return desc ? desc + (size_t)f * cap_per_frame * 32 : nullptr; };
    if (stride >= w && frames_page_locked(imgs, n, (size_t)stride * (h - 1) + w)) {
        int rc = ORB_OK, begun = 0;
        for (int r = 0; r < nctx && rc == ORB_OK; r++, begun++)
            if (lo(r + 1) > lo(r)) rc = extract_batch_begin(ctxs[r], imgs + lo(r), lo(r + 1) - lo(r), w, h, stride, kps != nullptr, desc != nullptr, 1);
        if (rc != ORB_OK) {  // leave no context with work in flight or a batch open
            char msg[512];
            snprintf(msg, sizeof(msg), "%s", orb_last_error());
            for (int r = 0; r < begun; r++) {
                (void)hipSetDevice(ctxs[r]->prm.device);
                if (ctxs[r]->stream) (void)hipStreamSynchronize(ctxs[r]->stream);
                ctxs[r]->hostBatch.open = false;
            }
            orb_set_error("%s", msg);
            return rc;
        }
        for (int r = 0; r < nctx; r++) {
            if (lo(r + 1) == lo(r)) continue;
            const int rr = extract_batch_finish(ctxs[r], outK(lo(r)), outD(lo(r)), cap_per_frame, n_out + lo(r));
            if (rr != ORB_OK && rc == ORB_OK) {
                char msg[400];
                snprintf(msg, sizeof(msg), "%s", orb_last_error());
                orb_set_error("frame block %d: %s", r, msg);
                rc = rr;
            }
        }
        return rc;
    }
    std::vector<int> rcs(nctx, ORB_OK);
    std::vector<std::vector<char> > msgs(nctx);
    auto work = [&](int r) {
        if (lo(r + 1) == lo(r)) return;
        rcs[r] = orb_extract_batch(ctxs[r], imgs + lo(r), lo(r + 1) - lo(r), w, h, stride, outK(lo(r)), outD(lo(r)), cap_per_frame,
                                   n_out + lo(r));
        if (rcs[r]) {  // the message is per thread: hand it to the caller's
            const char* m = orb_last_error();
            msgs[r].assign(m, m + strlen(m) + 1);
        }
    };
    std::vector<std::thread> th;
    for (int r = 1; r < nctx; r++) th.emplace_back(work, r);
    work(0);
    for (std::thread& t : th) t.join();
    for (int r = 0; r < nctx; r++)
        if (rcs[r]) {
            orb_set_error("frame block %d: %s", r, msgs[r].data());
            return rcs[r];
        }
    return ORB_OK;
}

// The same with every block resident on its context's device: one enqueue per context from the calling thread.
int orb_extract_batch_over_contexts_device(orb_ctx* const* ctxs, int nctx, const uint8_t* const* d_imgs, int n, int w, int h,
                                           int stride, int64_t frame_stride, orb_keypoint* const* d_kps, uint8_t* const* d_desc,
                                           int cap_per_frame, int* const* d_n_out, void* const* streams) {
    for (int r = 0; r < nctx; r++) {
        const int f0 = (int)((int64_t)n * r / nctx), f1 = (int)((int64_t)n * (r + 1) / nctx);
        if (f1 == f0) continue;
        if (!d_imgs[r] || !d_kps[r] || !d_desc[r] || !d_n_out[r]) {
            orb_set_error("orb_extract_batch_multi_device: block %d has a null buffer", r);
            return ORB_ERR_INVALID;
        }
        const int rc = run_batch(ctxs[r], d_imgs[r], f1 - f0, w, h, stride, frame_stride, d_kps[r], d_desc[r], cap_per_frame,
                                 d_n_out[r], streams ? (hipStream_t)streams[r] : ctxs[r]->stream);
        if (rc) {
            char msg[400];
            snprintf(msg, sizeof(msg), "%s", orb_last_error());
            orb_set_error("frame block %d: %s", r, msg);
            return rc;
        }
    }
    if (!streams)  // the contexts' own streams: the call is complete when it returns
        for (int r = 0; r < nctx; r++) {
            HIP_TRY(hipSetDevice(ctxs[r]->prm.device));
            HIP_TRY(hipStreamSynchronize(ctxs[r]->stream));
        }
    return ORB_OK;
}

// Page-locked host memory for callers that do not link the HIP runtime themselves: frames read into such a block
// (or any memory registered with hipHostRegister) reach the device without a staging copy.
extern "C" int orb_alloc_host(size_t bytes, void** out) {
    if (!out || bytes == 0) {
        orb_set_error("orb_alloc_host: bad arguments");
        return ORB_ERR_INVALID;
    }
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        *out = nullptr;
        orb_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return ORB_ERR_NOMEM;
    }
    return ORB_OK;
}
extern "C" void orb_free_host(void* p) {
    if (p) (void)hipHostFree(p);
}

extern "C" int orb_extract(orb_ctx* c, const uint8_t* img, int w, int h, int stride, orb_keypoint* kps, uint8_t* desc,
                           int cap, int* n_out) {
    if (!n_out) {
        orb_set_error("orb_extract: null n_out");
        return ORB_ERR_INVALID;
    }
    if (!img || w <= 0 || h <= 0) {  // empty image: silent return, src/ORBextractor.cc:451-452
        *n_out = 0;
        return ORB_OK;
    }
    const uint8_t* one[1] = {img};
    return orb_extract_batch(c, one, 1, w, h, stride, kps, desc, cap, n_out);
}

// Levels [l0, l1) of one frame of the last batch to host memory.  A two-dimensional copy into pageable memory
// goes row by row through the runtime (2 ms per level); instead every level's pitched block is sent to page-locked
// staging by the DMA engine, all levels in flight together, and the rows are unpacked on the host.
static int download_levels(orb_ctx* c, int frame, int l0, int l1, uint8_t* const* dst, const int* dstStride) {
    const OrbGeom& g = c->geom;
    HIP_TRY(hipSetDevice(c->prm.device));
    HIP_TRY(hipDeviceSynchronize());  // the batch may have run on the caller's stream
    size_t total = 0;
    size_t off[ORB_MAX_LEVELS];
    for (int l = l0; l < l1; l++) {
        const OrbLevel& L = g.lv[l];
        const size_t pitch = l == 0 ? (size_t)c->lastStride0 : (size_t)L.pitch;
        off[l] = total;
        total += (pitch * (size_t)(L.h - 1) + (size_t)L.w + 255) & ~(size_t)255;
    }
    int rc = c->pinOut.ensure(total);
    if (rc) return rc;
    for (int l = l0; l < l1; l++) {
        const OrbLevel& L = g.lv[l];
        const uint8_t* src = l == 0 ? c->lastImg0 + (int64_t)frame * c->lastFrameStride0
                                    : (const uint8_t*)c->dPyr.p + L.off + (int64_t)frame * L.frameStride;
        const size_t pitch = l == 0 ? (size_t)c->lastStride0 : (size_t)L.pitch;
        HIP_TRY(hipMemcpyAsync((uint8_t*)c->pinOut.p + off[l], src, pitch * (size_t)(L.h - 1) + (size_t)L.w,
                               hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int l = l0; l < l1; l++) {
        const OrbLevel& L = g.lv[l];
        const size_t pitch = l == 0 ? (size_t)c->lastStride0 : (size_t)L.pitch;
        const uint8_t* ps = (const uint8_t*)c->pinOut.p + off[l];
        for (int r = 0; r < L.h; r++) memcpy(dst[l] + (size_t)r * dstStride[l], ps + (size_t)r * pitch, (size_t)L.w);
    }
    return ORB_OK;
}

extern "C" int orb_get_pyramid_level(orb_ctx* c, int frame, int level, uint8_t* dst, int dst_stride, int* w, int* h) {
    if (!c || !c->geomValid || !c->lastImg0) {
        orb_set_error("no batch has been extracted yet");
        return ORB_ERR_INVALID;
    }
    const OrbGeom& g = c->geom;
    if (level < 0 || level >= g.nlevels || frame < 0 || frame >= c->lastFrames) {
        orb_set_error("frame %d / level %d out of range", frame, level);
        return ORB_ERR_INVALID;
    }
    const OrbLevel& L = g.lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (!dst) return ORB_OK;
    if (dst_stride < L.w) {
        orb_set_error("dst_stride %d < level width %d", dst_stride, L.w);
        return ORB_ERR_INVALID;
    }
    uint8_t* dsts[ORB_MAX_LEVELS] = {};
    int strides[ORB_MAX_LEVELS] = {};
    dsts[level] = dst;
    strides[level] = dst_stride;
    return download_levels(c, frame, level, level + 1, dsts, strides);
}

extern "C" int orb_get_pyramid(orb_ctx* c, int frame, uint8_t* const* dst, const int* dst_stride) {
    if (!c || !c->geomValid || !c->lastImg0 || !dst || !dst_stride) {
        orb_set_error(!c || !dst || !dst_stride ? "orb_get_pyramid: null argument" : "no batch has been extracted yet");
        return ORB_ERR_INVALID;
    }
    const OrbGeom& g = c->geom;
    if (frame < 0 || frame >= c->lastFrames) {
        orb_set_error("frame %d out of range", frame);
        return ORB_ERR_INVALID;
    }
    for (int l = 0; l < g.nlevels; l++)
        if (!dst[l] || dst_stride[l] < g.lv[l].w) {
            orb_set_error("level %d: null destination or stride %d < width %d", l, dst_stride[l], g.lv[l].w);
            return ORB_ERR_INVALID;
        }
    return download_levels(c, frame, 0, g.nlevels, dst, dst_stride);
}

extern "C" int orb_debug_get_fast_keys(orb_ctx* c, int frame, int level, int* xs, int* ys, int* scores, int cap,
                                       int* n_out) {
    if (!c || !c->geomValid || !n_out) {
        orb_set_error("orb_debug_get_fast_keys: bad arguments");
        return ORB_ERR_INVALID;
    }
    const OrbGeom& g = c->geom;
    if (level < 0 || level >= g.nlevels || frame < 0 || frame >= c->lastFrames) {
        orb_set_error("frame %d / level %d out of range", frame, level);
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->prm.device));
    HIP_TRY(hipDeviceSynchronize());
    const OrbLevel& L = g.lv[level];
    const int nCells = L.nCols * L.nRows;
    std::vector<int> cnt(nCells);
    std::vector<uint32_t> keys((size_t)nCells * g.cellCap);
    const int64_t cell0 = (int64_t)frame * g.cellsPerFrame + L.cellBase;
    HIP_TRY(hipMemcpy(cnt.data(), (int*)c->dCellCount.p + cell0, (size_t)nCells * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(keys.data(), (uint32_t*)c->dCellKeys.p + cell0 * g.cellCap, keys.size() * 4,
                      hipMemcpyDeviceToHost));
    int n = 0;
    for (int ci = 0; ci < nCells; ci++) {
        const int i = ci / L.nCols, j = ci % L.nCols;
        for (int k = 0; k < cnt[ci]; k++) {
            const uint32_t key = keys[(size_t)ci * g.cellCap + k];
            if (n < cap) {
                if (xs) xs[n] = 3 + j * L.wCell + (int)(key & 0xff);
                if (ys) ys[n] = 3 + i * L.hCell + (int)((key >> 8) & 0xff);
                if (scores) scores[n] = (int)(key >> 16);
            }
            n++;
        }
    }
    *n_out = n;
    return ORB_OK;
}

extern "C" int orb_resize(orb_ctx* c, const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                          int dstride, int area) {
    if (!c || !src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1 || sstride < sw || dstride < dw) {
        orb_set_error("orb_resize: bad arguments");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->prm.device));
    std::vector<OrbAreaTap> at;
    std::vector<OrbLinTap> lt;
    int rc;
    if (area) {
        if ((rc = area_taps(sw, dw, at))) return rc;
        if ((rc = area_taps(sh, dh, at))) return rc;
        if ((rc = check_tile_spans(sw, dw, ORB_RS_TW, &at[0], nullptr))) return rc;
        if ((rc = check_tile_spans(sh, dh, ORB_RS_TH, &at[dw], nullptr))) return rc;
    } else {
        linear_taps(sw, dw, true, lt);
        linear_taps(sh, dh, false, lt);
        if ((rc = check_tile_spans(sw, dw, ORB_RS_TW, nullptr, &lt[0]))) return rc;
        if ((rc = check_tile_spans(sh, dh, ORB_RS_TH, nullptr, &lt[dw]))) return rc;
    }
    DevBuf dS, dD, dT;
    const size_t tabBytes = area ? at.size() * sizeof(OrbAreaTap) : lt.size() * sizeof(OrbLinTap);
    rc = dS.ensure((size_t)sstride * sh);
    if (!rc) rc = dD.ensure((size_t)dstride * dh);
    if (!rc) rc = dT.ensure(tabBytes);
    if (!rc) {
        hipError_t e = hipMemcpy(dS.p, src, (size_t)sstride * (sh - 1) + sw, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dT.p, area ? (void*)at.data() : (void*)lt.data(), tabBytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            rc = orb_launch_resize_single((const uint8_t*)dS.p, sw, sh, sstride, (uint8_t*)dD.p, dw, dh, dstride,
                                          (const OrbAreaTap*)dT.p, (const OrbAreaTap*)dT.p + dw, (const OrbLinTap*)dT.p,
                                          (const OrbLinTap*)dT.p + dw, area, nullptr);
            if (!rc) e = hipDeviceSynchronize();
            if (!rc && e == hipSuccess) e = hipMemcpy2D(dst, dstride, dD.p, dstride, dw, dh, hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) {
            orb_set_error("orb_resize: %s", hipGetErrorString(e));
            rc = ORB_ERR_HIP;
        }
    }
    dS.release();
    dD.release();
    dT.release();
    return rc;
}

extern "C" int orb_gaussian_blur(orb_ctx* c, const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    if (!c || !src || !dst || w < 1 || h < 1 || sstride < w || dstride < w) {
        orb_set_error("orb_gaussian_blur: bad arguments");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->prm.device));
    DevBuf dS, dD;
    int rc = dS.ensure((size_t)sstride * h);
    if (!rc) rc = dD.ensure((size_t)dstride * h);
    if (!rc) {
        hipError_t e = hipMemcpy(dS.p, src, (size_t)sstride * (h - 1) + w, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            rc = orb_launch_blur_single((const uint8_t*)dS.p, w, h, sstride, (uint8_t*)dD.p, dstride,
                                        (c->prm.quirks & ORB_Q_GAUSS_LEGACY) ? 1 : 0, nullptr);
            if (!rc) e = hipDeviceSynchronize();
            if (!rc && e == hipSuccess) e = hipMemcpy2D(dst, dstride, dD.p, dstride, w, h, hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) {
            orb_set_error("orb_gaussian_blur: %s", hipGetErrorString(e));
            rc = ORB_ERR_HIP;
        }
    }
    dS.release();
    dD.release();
    return rc;
}

// ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:460-476: population count of the 256-bit XOR.
extern "C" int orb_hamming(const uint8_t a[32], const uint8_t b[32]) {
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) +
           __builtin_popcountll(x[3] ^ y[3]);
}
