/* liborbhip — C ABI of the MI355X (gfx950) ORB front end.
 *
 * The reference (JiahangWu/ORB_SLAM2) has no FFI layer: its boundary for this
 * path is the C++ class API called from Tracking.  The drop-in C++ classes in
 * orb_slam2_amd/cpp/ (same names and signatures as include/ORBextractor.h,
 * include/ORBmatcher.h, include/Frame.h of the reference) call the functions
 * below; so do the ctypes binding orb_slam2_amd/binding.py and the tests.
 *
 * Conventions: every function returns 0 (ORB_OK) or a negative orb_status;
 * orb_last_error() gives the message of the calling thread's last failure.
 * No C++ types, exceptions or torch types cross this boundary.  A context is
 * single-threaded (like one ORBextractor object, which owns its pyramid);
 * distinct contexts may be used from distinct threads / on distinct devices.
 * Pointers named d_* are device pointers on the context's device, all others
 * are host pointers.  `stream` is a hipStream_t passed as void* (NULL = the
 * null stream).  The *_device entry points enqueue work and return without
 * synchronising.  The host-pointer entry points work on a stream of the
 * context's own (non-blocking: it does not wait for the null stream) and
 * return when their results are in the caller's arrays.
 */
#ifndef ORB_HIP_H
#define ORB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ORB_OK = 0,
    ORB_ERR_INVALID = -1,     /* bad argument */
    ORB_ERR_UNSUPPORTED = -2, /* configuration the kernels do not cover (stated in the message) */
    ORB_ERR_HIP = -3,         /* a HIP runtime call failed */
    ORB_ERR_CAPACITY = -4,    /* an output or internal buffer was too small; nothing was truncated silently */
    ORB_ERR_NOMEM = -5
} orb_status;

/* Behaviour switches: where the fork's text compiles but differs from upstream
 * ORB-SLAM2, the fork's behaviour is selectable.  ORB_QUIRKS_FORK reproduces
 * the reference as written, ORB_QUIRKS_CANONICAL upstream. */
enum {
    ORB_Q_RESIZE_AREA = 1u << 0,     /* src/ORBextractor.cc:931 INTER_AREA (upstream INTER_LINEAR) */
    ORB_Q_UMAX_FORK = 1u << 1,       /* src/ORBextractor.cc:431 */
    ORB_Q_DIVIDE_N3 = 1u << 2,       /* src/ORBextractor.cc:567-568 */
    ORB_Q_GRID_EXCLUSIVE = 1u << 3,  /* src/Frame.cc:101,110,112 */
    ORB_Q_WINDOW_CIRCULAR = 1u << 4, /* src/Frame.cc:133 */
    ORB_Q_GAUSS_LEGACY = 1u << 5,    /* OpenCV <= 4.1 Gaussian taps {18,34,49,55,49,34,18} */
    ORB_QUIRKS_FORK = 0x1f,
    ORB_QUIRKS_CANONICAL = 0
};

typedef struct orb_ctx orb_ctx; /* extractor + windowed matcher state on one device */
typedef struct orb_bf orb_bf;   /* one shard of a brute-force descriptor database */

/* Constructor arguments of ORBextractor (include/ORBextractor.h:36) plus sizing. */
typedef struct {
    int nfeatures;
    float scale_factor;
    int nlevels;
    int ini_th_fast;
    int min_th_fast;
    int max_width, max_height; /* largest frame the context will be given */
    int max_batch;             /* largest number of frames (or frame pairs) per call */
    int device;                /* HIP device ordinal */
    unsigned quirks;
} orb_params;

/* Same layout as cv::KeyPoint (28 bytes). */
typedef struct {
    float x, y, size, angle, response;
    int octave, class_id;
} orb_keypoint;

/* ---- ORBextractor ------------------------------------------------------- */

/* ORBextractor::ORBextractor, src/ORBextractor.cc:369-436. */
int orb_create(const orb_params* params, orb_ctx** out);
void orb_destroy(orb_ctx* ctx);

/* Upper bound of keypoints one frame can produce (sum over levels of quota+3,
 * the quadtree never trims): size kps/desc buffers with it. */
int orb_max_keypoints_per_frame(const orb_ctx* ctx);

/* ORBextractor::operator()(image, mask, keypoints, descriptors),
 * src/ORBextractor.cc:448-509 (mask ignored, as there).  8-bit single channel
 * image in host memory; keypoints level-major in the reference's order,
 * descriptors cap x 32 bytes.  An empty image (w or h == 0) yields *n_out = 0. */
int orb_extract(orb_ctx* ctx, const uint8_t* img, int w, int h, int stride, orb_keypoint* kps, uint8_t* desc, int cap,
                int* n_out);

/* The same over n independent frames of one size (n <= max_batch).  Frame f
 * writes kps[f*cap_per_frame ..] and desc[f*cap_per_frame*32 ..], n_out[f].
 * Frames in pageable memory are first copied into page-locked staging by host
 * threads the context keeps (the host's memory rate bounds that copy, for all
 * contexts of a machine together); frames that already lie in page-locked
 * memory (orb_alloc_host, hipHostMalloc, hipHostRegister - the call finds out
 * by itself) are fetched by the DMA engine from where they are, as one copy per
 * frame group when they follow each other stride*h bytes apart. */
int orb_extract_batch(orb_ctx* ctx, const uint8_t* const* imgs, int n, int w, int h, int stride, orb_keypoint* kps,
                      uint8_t* desc, int cap_per_frame, int* n_out);

/* Page-locked host memory for frames (no counterpart in the reference): what a
 * capture loop reads its images into so that orb_extract_batch needs no staging
 * copy.  Usable from every device. */
int orb_alloc_host(size_t bytes, void** out);
void orb_free_host(void* p);

/* The same with every buffer resident in device memory: frame f starts at
 * d_imgs + f*frame_stride.  Asynchronous on `stream`. */
int orb_extract_batch_device(orb_ctx* ctx, const uint8_t* d_imgs, int n, int w, int h, int stride,
                             int64_t frame_stride, orb_keypoint* d_kps, uint8_t* d_desc, int cap_per_frame,
                             int* d_n_out, void* stream);

/* Diagnostics.  A batch of about a hundred frames or more distributes its keypoints (DistributeOctTree,
 * src/ORBextractor.cc:571-796) in two passes: the first with coarse per-level histograms, which settles the levels
 * whose keypoints are spread as in ordinary images, the second with finer ones for the levels that crowd their
 * keypoints into a small part of the frame.  Results do not depend on the pass; throughput does.  Waits for the
 * device and reports how many (frame, level) pairs of the context's last batch went to the second pass (0 for
 * smaller batches and for frames above ~2K, which run one pass). */
int orb_quadtree_second_pass_levels(orb_ctx* ctx, int* levels);

/* After *_device calls have completed: 0 if no internal capacity or depth
 * limit was hit by the extraction batches and windowed-matcher calls since the
 * last check, else ORB_ERR_CAPACITY (synchronises).
 *
 * One context = one set of scratch buffers.  Calls on a context must not
 * overlap: a *_device call still running on the caller's stream when another
 * entry point of the same context starts (host-pointer ones included, which
 * use streams of the context) would have its scratch overwritten.  Wait for
 * the stream, or use one context per stream. */
int orb_check_last_batch(orb_ctx* ctx);

/* ORBextractor::mvImagePyramid[level] (include/ORBextractor.h:71) of frame
 * `frame` of the last batch, copied to host memory (ROI pixels, no padding). */
int orb_get_pyramid_level(orb_ctx* ctx, int frame, int level, uint8_t* dst, int dst_stride, int* w, int* h);

/* All levels of that frame with one synchronisation: dst[l] receives level l
 * (rows dst_stride[l] bytes apart; sizes from orb_get_pyramid_level with a NULL
 * destination). */
int orb_get_pyramid(orb_ctx* ctx, int frame, uint8_t* const* dst, const int* dst_stride);

/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:50-68), the per-level
 * quotas mnFeaturesPerLevel and the 16-entry umax table.  NULL = skip. */
int orb_get_tables(const orb_ctx* ctx, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* per_level,
                   int* umax);

/* The same tables from the constructor arguments alone (nfeatures, scale_factor,
 * nlevels, quirks of *params): host arithmetic of src/ORBextractor.cc:377-435,
 * no device, no context. */
int orb_compute_tables(const orb_params* params, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                       int* per_level, int* umax);

/* Stage access for parity tests: keys handed to DistributeOctTree for
 * (frame, level) of the last batch, in the reference's order
 * (src/ORBextractor.cc:862-867): x, y relative to the 16-pixel border and the
 * FAST score.  Returns the count through *n_out (writes at most cap). */
int orb_debug_get_fast_keys(orb_ctx* ctx, int frame, int level, int* xs, int* ys, int* scores, int cap, int* n_out);

/* Stand-alone stages (host buffers), the kernels of the pipeline run alone:
 * cv::resize as called at src/ORBextractor.cc:926-931 (area != 0: INTER_AREA),
 * cv::GaussianBlur as called at :493. */
int orb_resize(orb_ctx* ctx, const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride,
               int area);
int orb_gaussian_blur(orb_ctx* ctx, const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);

/* ---- Frame constructor tail ---------------------------------------------- */

/* Frame::UndistortKeyPoints (declared at include/Frame.h:121; body as in
 * upstream ORB-SLAM2): cv::undistortPoints(pts, pts, mK, mDistCoef, noArray(),
 * mK) on the keypoint positions, every other field copied.  K = fx, fy, cx, cy
 * of mK; dist = k1, k2, p1, p2, k3 of mDistCoef
 * (Examples/Monocular/TUM1.yaml:8-17).  k1 == 0 copies the keypoints, as there. */
int orb_undistort_keypoints(orb_ctx* ctx, const float K[4], const float dist[5], const orb_keypoint* in,
                            orb_keypoint* out, int n);
/* ExtractORB and UndistortKeyPoints of the Frame constructor (src/Frame.cc:67-74,
 * include/Frame.h:121) for one frame in host memory, as one enqueue and one wait
 * (orb_extract followed by orb_undistort_keypoints costs a second round trip to
 * the device): kps as orb_extract writes them, kps_un the same keypoints with
 * undistorted positions (equal to kps when k1 == 0). */
int orb_extract_undistorted(orb_ctx* ctx, const uint8_t* img, int w, int h, int stride, const float K[4],
                            const float dist[5], orb_keypoint* kps, orb_keypoint* kps_un, uint8_t* desc, int cap,
                            int* n_out);
/* The same on n device-resident keypoints (e.g. a whole batch, n = frames * cap). */
int orb_undistort_keypoints_device(orb_ctx* ctx, const float K[4], const float dist[5], const orb_keypoint* d_in,
                                   orb_keypoint* d_out, long long n, void* stream);

/* Frame::ComputeImageBounds (include/Frame.h:123): bounds = mnMinX, mnMaxX,
 * mnMinY, mnMaxY of the undistorted image corners (the image rectangle when
 * k1 == 0).  The reference keeps them in static members of Frame, set from the
 * first frame; here the context keeps them, and the windowed matchers
 * (orb_match_window*, orb_match_projected) build their 64x48 grid over them
 * from then on.  orb_set_image_bounds installs given bounds, NULL returns to
 * "the image rectangle of each call". */
int orb_compute_image_bounds(orb_ctx* ctx, int cols, int rows, const float K[4], const float dist[5], float bounds[4]);
int orb_set_image_bounds(orb_ctx* ctx, const float bounds[4]);

/* ---- ORBmatcher ---------------------------------------------------------- */

/* ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:460-476 (host, pure). */
int orb_hamming(const uint8_t a[32], const uint8_t b[32]);

/* ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:26-136, with the grid
 * query Frame::GetFeaturesInArea (src/Frame.cc:86-144) inside.  Frames are
 * given as keypoints + descriptors (mvKeysUn == mvKeys: no distortion) and the
 * image size that defines the 64x48 grid.  prev_xy (2*n1 floats) is
 * vbPrevMatched in/out, matches12 (n1 ints) is vnMatches12 out. */
int orb_match_window(orb_ctx* ctx, const orb_keypoint* k1, const uint8_t* d1, int n1, const orb_keypoint* k2,
                     const uint8_t* d2, int n2, int img_w, int img_h, float* prev_xy, int* matches12, int window,
                     float nnratio, int check_ori, int* nmatches);

/* The same over npairs frame pairs resident in device memory (npairs <=
 * max_batch).  Pair p reads d_k1 + p*stride1 keypoints, d_d1 + p*stride1*32
 * descriptor bytes, d_n1[p] valid entries (likewise frame 2), and
 * reads/writes d_prev_xy + p*stride1*2, d_matches12 + p*stride1, d_nmatches[p]. */
int orb_match_window_batch_device(orb_ctx* ctx, int npairs, const orb_keypoint* d_k1, const uint8_t* d_d1,
                                  const int* d_n1, int stride1, const orb_keypoint* d_k2, const uint8_t* d_d2,
                                  const int* d_n2, int stride2, int img_w, int img_h, float* d_prev_xy,
                                  int* d_matches12, int window, float nnratio, int check_ori, int* d_nmatches,
                                  void* stream);

/* Device memory of the windowed matchers (orb_match_window*, orb_match_projected,
 * orb_match_local_map), grown on demand and kept by the context: per pair the
 * searched frame's bucket-sorted grid (13 bytes per keypoint + 12 KB), per
 * query 13 bytes of bookkeeping and a candidate list of min(stride2, 256)
 * entries of 4 bytes (e.g. 64 pairs of 2070-keypoint frames: 136 MB), and one
 * candidate pool for what windows hold beyond their first 256 candidates: 4 Mi
 * entries (16 MiB) to start with, never more than the worst case.  A call whose
 * windows overrun the pool fails with ORB_ERR_CAPACITY and the pool is grown to
 * what that call needed before the next one.  The host-pointer entry points
 * write no output in that case, grow the pool and run again by themselves.  The
 * asynchronous entry point cannot: it reports through orb_check_last_batch, and
 * in every pair that held a query whose window did not fit, that query's list
 * was cut at 256 candidates, so the pair's d_matches12 / d_prev_xy are void and
 * its d_nmatches is -1.  The report stays pending until orb_check_last_batch
 * collects it, whatever host-pointer calls are made on the context meanwhile
 * (they keep error words of their own).  orb_match_reserve_candidates sets
 * the pool size (entries) ahead of time, e.g. before asynchronous calls with
 * whole-image windows: the worst case is pairs * stride1 * (stride2 - 256). */
int orb_match_reserve_candidates(orb_ctx* ctx, long long entries);

/* Kernel-level form of ORBmatcher::SearchByProjection(Frame&, const Frame&,
 * th, bMono), src/ORBmatcher.cc:279-408: the caller projects the last frame's
 * map points (lines :302-331 are pose algebra on host objects) and passes per
 * query the pixel (u,v), the search radius, the octave range of the window
 * query, the keypoint angle, the map point's descriptor and whether the map
 * point has observations; occupied2[i2] != 0 marks current-frame keypoints
 * that already hold an observed map point.  match2[i2] = index of the query
 * whose map point keypoint i2 holds afterwards, -1 if the keypoint was never
 * assigned, -2 if it was assigned and then cleared by the rotation check (the
 * reference stores NULL there, :400). */
int orb_match_projected(orb_ctx* ctx, const float* u, const float* v, const float* radius, const int* min_level,
                        const int* max_level, const float* qangle, const uint8_t* qdesc, const uint8_t* qobs, int nq,
                        const orb_keypoint* k2, const uint8_t* d2, int n2, int img_w, int img_h,
                        const uint8_t* occupied2, int* match2, int th_high, int check_ori, int* nmatches);

/* The overloads below are declared by the reference (include/ORBmatcher.h) and
 * defined nowhere in it; their candidate loops follow upstream ORB-SLAM2, of
 * which the fork is a copy, and the steps listed in the header's comments.
 * As with orb_match_projected the caller does the pose algebra and passes
 * pixels, radii and level ranges; the window query is Frame::GetFeaturesInArea
 * as the context's quirks configure it.
 *
 * Other overloads map onto orb_match_projected as it stands:
 * SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)
 * (include/ORBmatcher.h:75): levels [l-1, l+1], occupied2 = keypoints that
 * hold any map point, qobs all 1, th_high = ORBdist, check_ori as configured;
 * SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (:90): levels
 * [l-1, l], occupied2 = entries of vpMatched already set, qobs all 1,
 * th_high = TH_LOW, check_ori 0. */

/* Kernel-level form of SearchByProjection(Frame&, const vector<MapPoint*>&, th)
 * (include/ORBmatcher.h:24-39, tracking of the local map): per query the
 * projection (u,v) = (mTrackProjX, mTrackProjY), radius = r * scale factor of the
 * predicted level, levels [predicted-1, predicted], the map point's descriptor
 * and qobs = "has observations".  Keypoints that hold an observed map point are
 * skipped (occupied2 on entry, then whatever earlier queries stored); best and
 * second-best distance are kept with their levels; a best distance <= th_high
 * is accepted unless both lie on the same level and best > nnratio * second.
 * No rotation check.  match2 as in orb_match_projected (-1 or the query). */
int orb_match_local_map(orb_ctx* ctx, const float* u, const float* v, const float* radius, const int* min_level,
                        const int* max_level, const uint8_t* qdesc, const uint8_t* qobs, int nq, const orb_keypoint* k2,
                        const uint8_t* d2, int n2, int img_w, int img_h, const uint8_t* occupied2, int* match2,
                        int th_high, float nnratio, int* nmatches);

/* Candidate loop shared by Fuse(KeyFrame*, vpMapPoints, th), Fuse(KeyFrame*,
 * Scw, vpPoints, th, vpReplacePoint) (include/ORBmatcher.h:163-184) and
 * SearchBySim3 (:146-160): every query on its own takes the keypoint of
 * smallest descriptor distance in its window and level range (first in window
 * order among equals); best_idx[i] = that keypoint when the distance is <= th
 * (TH_LOW for Fuse, TH_HIGH for SearchBySim3), else -1; best_dist[i] (may be
 * NULL) = the distance, 256 when the window was empty.  inv_level_sigma2 != NULL
 * (nlevels entries, mvInvLevelSigma2) adds the monocular reprojection gate of
 * the first Fuse overload: a candidate whose squared pixel distance to (u,v)
 * times the inverse sigma^2 of its level exceeds chi2_max (5.99) is left out.
 * What to do with the result (replace, add observation, mutual agreement of
 * the two SearchBySim3 directions) is map surgery on host objects. */
int orb_match_best(orb_ctx* ctx, const float* u, const float* v, const float* radius, const int* min_level,
                   const int* max_level, const uint8_t* qdesc, int nq, const orb_keypoint* k2, const uint8_t* d2, int n2,
                   int img_w, int img_h, const float* inv_level_sigma2, int nlevels, double chi2_max, int th,
                   int* best_idx, int* best_dist);

/* MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:158-222, for npoints
 * map points at once.  desc holds every point's observation descriptors back
 * to back (32 bytes each), point p owning rows offsets[p] .. offsets[p+1]-1 in
 * the order the reference collects them (:176-181).  best_index[p] = row,
 * within the point's list, of the descriptor with the least median distance
 * to the others (first on ties, :208-212), -1 for an empty list (the reference
 * returns without touching mDescriptor); best_median (may be NULL) receives
 * that median.  self_distance != 0 reproduces the fork's line :195, which
 * compares every descriptor with itself (all distances 0, row 0 wins);
 * 0 is upstream ORB-SLAM2 (descriptor i against descriptor j).  Lists of up to
 * 65535 observations. */
int orb_distinctive_descriptors(orb_ctx* ctx, const uint8_t* desc, const int* offsets, int npoints, int self_distance,
                                int* best_index, int* best_median);
/* The same on device-resident arrays (d_desc 4-byte aligned); a list longer
 * than 65535 yields best_index -2. */
int orb_distinctive_descriptors_device(orb_ctx* ctx, const uint8_t* d_desc, const int* d_offsets, int npoints,
                                       int self_distance, int* d_best_index, int* d_best_median, void* stream);

/* ---- brute-force top-2 (relocalisation / loop-candidate scale) ----------- */

/* Dense scan with the best / second-best rule of src/ORBmatcher.cc:53-76:
 * best = smallest distance, lowest database index among equals; second =
 * second-smallest distance counted with multiplicity. */
int orb_bf_create(int device, orb_bf** out);
void orb_bf_destroy(orb_bf* bf);

/* Load this shard's rows (n x 32 bytes, host memory); global_base = index of
 * its first row in the whole database. */
int orb_bf_set_database(orb_bf* bf, const uint8_t* db, int64_t n, int64_t global_base);
/* The same from device memory; the shard keeps the pointer (no copy). */
int orb_bf_set_database_device(orb_bf* bf, const uint8_t* d_db, int64_t n, int64_t global_base);

/* Whole query on one shard, host buffers.  Row numbers come back as int32_t
 * (-1 = none): a shard whose last global row is above 2^31 - 1 is refused
 * (ORB_ERR_UNSUPPORTED); the packed form below carries 32 bits. */
int orb_bf_query_top2(orb_bf* bf, const uint8_t* q, int nq, int32_t* best_idx, uint16_t* best_dist,
                      uint16_t* second_dist);

/* Per-shard result in packed form, device buffers, asynchronous:
 * d_packed[i] = best_dist << 48 | second_dist << 32 | global index of best
 * (0xFFFF / 0xFFFFFFFF when the shard holds fewer than 1 / 2 rows). */
int orb_bf_query_packed_device(orb_bf* bf, const uint8_t* d_q, int nq, uint64_t* d_packed, void* stream);

/* orb_bf_query_packed_device for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid
 * it).  No counterpart in the reference.  Same arguments and result, but the
 * per-chunk partial results (and the one word per query through which the
 * chunks tell each other their thresholds, cleared on `stream` by the call)
 * live in a fixed 32 MiB block inside the library's device image instead of a
 * buffer grown on demand, and the call makes no hipMalloc / hipFree.  At most ORB_BF_NOALLOC_MAX_QUERIES queries per call
 * (more: ORB_ERR_CAPACITY, nothing is launched).  All calls on one device share
 * the block: calls that overlap in time on the same device race, whichever
 * handles or streams they use. */
#define ORB_BF_NOALLOC_MAX_QUERIES 262144
int orb_bf_query_packed_noalloc(orb_bf* bf, const uint8_t* d_q, int nq, uint64_t* d_packed, void* stream);

/* Measurement aid (no counterpart in the reference): the engine clock the scan
 * kernel ran at.  enable != 0 makes the following scans of this shard add up
 * shader-clock ticks and ticks of the constant 100 MHz counter over their
 * workgroups; every call waits for the device, returns through *mhz (may be
 * NULL) the ratio seen since the previous call (0 if nothing ran) and clears the
 * sums.  enable == 0 switches the probe off again. */
int orb_bf_clock_probe(orb_bf* bf, int enable, double* mhz);

/* Merge nshards packed results per query (d_gathered[s*nq + i], the layout an
 * all-gather produces) into the final answer; equals a scan of the whole
 * database.  Device buffers, asynchronous. */
int orb_bf_merge_packed_device(int device, const uint64_t* d_gathered, int nshards, int nq, int32_t* d_best_idx,
                               uint16_t* d_best_dist, uint16_t* d_second_dist, void* stream);

/* ---- several GPUs of one node from one process ------------------------------------
 * (No counterpart in the reference, whose front end runs on one thread.)
 *
 * Brute force: the database is split into contiguous row blocks, one per entry
 * of `devices`; every device scans its block for all queries, the per-query
 * packed words are exchanged with one all-gather of nq x 8 bytes per device
 * (RCCL's ncclAllGather over xGMI; librccl.so.1 is opened when the object is
 * created) and merged on every device.  The result equals orb_bf_query_top2 on
 * the whole database, ties across blocks included.  A device may be listed more
 * than once (then, or with ORBHIP_EXCHANGE=copy in the environment, the words
 * travel by device-to-device copies instead of RCCL): that is how one GPU
 * stands in for several in tests.  Calls on one object must not overlap. */
typedef struct orb_bf_multi orb_bf_multi;
int orb_bf_multi_create(const int* devices, int ndev, orb_bf_multi** out);
void orb_bf_multi_destroy(orb_bf_multi* m);
/* 1: RCCL all-gather, 0: device-to-device copies. */
int orb_bf_multi_exchange(const orb_bf_multi* m);
/* n x 32 bytes in host memory (n <= 2^31 - 1: results are int32_t row numbers); block r = rows
 * [r n / R, (r + 1) n / R), uploaded by one host thread per device. */
int orb_bf_multi_set_database(orb_bf_multi* m, const uint8_t* db, int64_t n);
int orb_bf_multi_query_top2(orb_bf_multi* m, const uint8_t* q, int nq, int32_t* best_idx, uint16_t* best_dist,
                            uint16_t* second_dist);

/* Frames: orb_extract_batch over n frames split into contiguous blocks
 * [r n / nctx, (r + 1) n / nctx), block r on ctxs[r] (distinct contexts, one per
 * device, each created with max_batch >= its block), one host thread per
 * context, no exchange.  Outputs as orb_extract_batch: the result does not
 * depend on nctx. */
int orb_extract_batch_multi(orb_ctx* const* ctxs, int nctx, const uint8_t* const* imgs, int n, int w, int h, int stride,
                            orb_keypoint* kps, uint8_t* desc, int cap_per_frame, int* n_out);
/* (Page-locked frames: the calling thread enqueues every context's block and
 * collects them in turn, no host thread is made and no host copy stands between
 * the frames and the link.  Pageable frames: one host thread per context.)
 *
 * The same split with block r resident on ctxs[r]'s device: d_imgs[r],
 * d_kps[r], d_desc[r], d_n_out[r] are that block's buffers (frame f of the
 * block at d_imgs[r] + f*frame_stride, outputs as orb_extract_batch_device).
 * One enqueue per context from the calling thread, no host thread, no copy.
 * streams[r] is the stream block r runs on and the call returns at once;
 * streams == NULL runs every block on its context's own stream and waits for
 * all of them. */
int orb_extract_batch_multi_device(orb_ctx* const* ctxs, int nctx, const uint8_t* const* d_imgs, int n, int w, int h,
                                   int stride, int64_t frame_stride, orb_keypoint* const* d_kps, uint8_t* const* d_desc,
                                   int cap_per_frame, int* const* d_n_out, void* const* streams);

/* ---- bag of words: DBoW2 vocabulary tree and SearchByBoW ----------------------- */

/* TemplatedVocabulary<FORB::TDescriptor, FORB> (ORBVocabulary, include/ORBVocabulary.h;
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h): the tree lives in device memory,
 * the per-feature descent runs there; the per-word bookkeeping (std::map in the
 * reference) is host code.  scoring: 0 L1_NORM .. 5 DOT_PRODUCT, weighting:
 * 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (BowVector.h:24-45). */
typedef struct orb_voc orb_voc;

/* loadFromTextFile, TemplatedVocabulary.h:1368-1470: first line "k L scoring
 * weighting", then one line per node (id = line order from 1): parent id, leaf
 * flag, 32 descriptor bytes, weight.  Empty lines are skipped. */
int orb_voc_load_text(const char* path, int device, orb_voc** out);
/* The same tree from arrays: entry i describes node i + 1 (parent[i] <= i). */
int orb_voc_create(int k, int L, int scoring, int weighting, const int* parent, const uint8_t* is_leaf,
                   const uint8_t* desc, const double* weight, int nnodes, int device, orb_voc** out);
void orb_voc_destroy(orb_voc* voc);
/* info[6] = k, L, scoring, weighting, nodes (without the root), words. */
int orb_voc_info(const orb_voc* voc, int info[6]);

/* transform(feature, word_id, weight, nid, levelsup), TemplatedVocabulary.h:1240-1289,
 * for n descriptors at once: descent by smallest Hamming distance, first child
 * on ties.  node_id = the node `levelsup` levels above the leaves on the path
 * (0 = root when levelsup >= L); when the descent ends in a leaf above that
 * level the reference leaves the value unwritten, here it is the leaf.  Any of
 * the three outputs may be NULL. */
int orb_voc_transform(orb_voc* voc, const uint8_t* desc, int n, int levelsup, int* word_id, double* weight,
                      int* node_id);
/* Device-resident descriptors (4-byte aligned) and outputs; asynchronous on `stream`. */
int orb_voc_transform_device(orb_voc* voc, const uint8_t* d_desc, int n, int levelsup, int* d_word_id,
                             double* d_weight, int* d_node_id, void* stream);

/* transform(features, BowVector&, FeatureVector&, levelsup),
 * TemplatedVocabulary.h:1136-1206 (KeyFrame::ComputeBoW, src/KeyFrame.cc:145-152,
 * calls it with levelsup 4).  BowVector: *n_bow (word id ascending, value)
 * pairs, weighted and normalised as the vocabulary's weighting / scoring
 * prescribe.  FeatureVector in CSR form: *n_fv node ids ascending, fv_offsets
 * (n_fv + 1), fv_index = feature numbers in increasing order per node.  Output
 * arrays hold n entries (fv_offsets n + 1). */
int orb_voc_bow(orb_voc* voc, const uint8_t* desc, int n, int levelsup, int* bow_id, double* bow_value, int* n_bow,
                int* fv_node, int* fv_offsets, int* fv_index, int* n_fv);

/* L1Scoring::score, ScoringObject.cpp:23-68 (the score ORBvoc.txt is built
 * for), of two normalised BowVectors; host, pure. */
double orb_bow_score_l1(const int* id1, const double* value1, int n1, const int* id2, const double* value2, int n2);

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&),
 * src/ORBmatcher.cc:138-268.  kf_* describe the keyframe: undistorted keypoints
 * (only .angle is read, :212-216), descriptors, kf_valid[i] != 0 where the
 * keyframe's map point i exists and is not bad (:170-176), FeatureVector in the
 * CSR form of orb_voc_bow; f_* the frame.  match_kf[j] (nf entries) = keyframe
 * feature whose map point frame feature j receives, -1 for none
 * (vpMapPointMatches).  Up to 8192 features per side. */
int orb_match_bow(orb_ctx* ctx, const orb_keypoint* kf_keys_un, const uint8_t* kf_desc, const uint8_t* kf_valid, int nkf,
                  const int* kf_fv_node, const int* kf_fv_offsets, const int* kf_fv_index, int kf_nfv,
                  const orb_keypoint* f_keys, const uint8_t* f_desc, int nf, const int* f_fv_node,
                  const int* f_fv_offsets, const int* f_fv_index, int f_nfv, float nnratio, int check_ori,
                  int* match_kf, int* nmatches);

/* SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (include/ORBmatcher.h:110,
 * loop detection; declared only, body as upstream ORB-SLAM2): as orb_match_bow
 * with three differences: features of the second keyframe take part only where
 * valid2 says they hold a good map point, the distance test is the strict
 * bestDist1 < TH_LOW, and the result is indexed by the first keyframe's
 * features: matches12[i] = feature of keyframe 2 whose map point feature i of
 * keyframe 1 is matched to, -1 for none (n1 entries). */
int orb_match_bow_keyframes(orb_ctx* ctx, const orb_keypoint* keys_un1, const uint8_t* desc1, const uint8_t* valid1,
                            int n1, const int* fv_node1, const int* fv_offsets1, const int* fv_index1, int nfv1,
                            const orb_keypoint* keys_un2, const uint8_t* desc2, const uint8_t* valid2, int n2,
                            const int* fv_node2, const int* fv_offsets2, const int* fv_index2, int nfv2, float nnratio,
                            int check_ori, int* matches12, int* nmatches);

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo)
 * with CheckDistEpipolarLine (include/ORBmatcher.h:133-144,196; declared only,
 * loops as upstream ORB-SLAM2, monocular members).  free1 / free2 mark the
 * features without a map point.  Under every common vocabulary node a free
 * feature of keyframe 1 takes the free feature of keyframe 2 that is at most
 * TH_LOW away, lies farther than 10 * sqrt(scale_factors[octave]) pixels from
 * the epipole (epipole = projection of camera centre 1 into keyframe 2; NULL
 * skips the test) and within 3.84 * level_sigma2[octave] of the epipolar line
 * x1' F12 (F12 row-major); smallest distance, the last candidate among equals.
 * mark_matched = 0 is upstream ORB-SLAM2, which never marks a second-side
 * feature as used; 1 gives every second-side feature away once.  matches12[i]
 * = feature of keyframe 2 or -1 (vMatchedPairs = the pairs (i, matches12[i])),
 * after the rotation check when check_ori != 0. */
int orb_match_triangulation(orb_ctx* ctx, const orb_keypoint* keys_un1, const uint8_t* desc1, const uint8_t* free1,
                            int n1, const int* fv_node1, const int* fv_offsets1, const int* fv_index1, int nfv1,
                            const orb_keypoint* keys_un2, const uint8_t* desc2, const uint8_t* free2, int n2,
                            const int* fv_node2, const int* fv_offsets2, const int* fv_index2, int nfv2,
                            const float F12[9], const float epipole[2], const float* scale_factors,
                            const float* level_sigma2, int nlevels, int mark_matched, int check_ori, int* matches12,
                            int* nmatches);

/* ---- measurement ------------------------------------------------------------ */

/* Stage timing with HIP events recorded on the launch stream around the four
 * stages of every extraction batch (pyramid, FAST cells, quadtree, describe).
 * orb_get_stage_timings synchronises, writes the mean milliseconds per batch of
 * each stage over the batches recorded since the last call (at most the last
 * 64) into ms[0..3], their number into *nbatches, and clears the record.
 * (No counterpart in the reference, which takes no timings.) */
#define ORB_NUM_STAGES 4
int orb_enable_stage_timing(orb_ctx* ctx, int on);
int orb_get_stage_timings(orb_ctx* ctx, float ms[ORB_NUM_STAGES], int* nbatches);

/* ---- misc ------------------------------------------------------------------ */
const char* orb_last_error(void);
const char* orb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ORB_HIP_H */
