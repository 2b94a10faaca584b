/* eao_fusion.h -- C-ABI of libeaofusion_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for EAO-Fusion's ORB front-end + local-BA hot path.  The reference has no FFI/plugin
 * layer: its boundary is the C++ class surface ORB_SLAM2::{ORBextractor, ORBmatcher, Optimizer}.  The
 * header-only adapters in include/eaofusion/ keep those signatures and call the functions below; every
 * entry point cites the reference interface it stands behind.  Plain pointers and sizes only -- no C++,
 * OpenCV, Eigen or torch types cross this ABI.
 *
 * Conventions
 *  - every function returns eao_status (0 = ok, < 0 = error) and never throws; eao_last_error() gives the
 *    thread-local message of the last failure.
 *  - the caller allocates all outputs; the library owns device memory inside handles.
 *  - "_device" variants take pointers that are already resident in HBM and a hipStream_t passed as void*
 *    (NULL = the null stream); they enqueue work on that stream and do NOT synchronise.
 *  - a handle is single-threaded (one HIP stream each); distinct handles are independent; the stateless
 *    functions are thread-safe.
 *  - there is NO CPU fallback: without a usable HIP device every compute entry point fails with
 *    EAO_ERR_NO_DEVICE.
 */
#ifndef EAO_FUSION_H_
#define EAO_FUSION_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t eao_status;
enum {
    EAO_OK = 0,
    EAO_ERR_INVALID = -1,    /* bad argument / unsupported geometry */
    EAO_ERR_NO_DEVICE = -2,  /* no HIP device or HIP runtime failure */
    EAO_ERR_CAPACITY = -3,   /* caller-provided output capacity too small */
    EAO_ERR_INTERNAL = -4
};

const char* eao_last_error(void);
/* 0 when a gfx950-class device is usable by this process, else EAO_ERR_NO_DEVICE (never initialises a context
 * beyond hipGetDeviceCount / hipGetDeviceProperties). */
eao_status eao_device_check(void);
const char* eao_version(void);

/* ------------------------------------------------------------------------------------------------
 * ORB extraction -- replaces ORB_SLAM2::ORBextractor
 *   ctor        reference include/ORBextractor.h:52-53, src/ORBextractor.cc:410-470
 *   operator()  reference include/ORBextractor.h:60-62, src/ORBextractor.cc:1043-1105
 * ------------------------------------------------------------------------------------------------ */
typedef struct eao_orb eao_orb; /* opaque handle */

typedef struct {
    int32_t nfeatures;   /* ORBextractor.nFeatures  */
    float scale_factor;  /* ORBextractor.scaleFactor */
    int32_t nlevels;     /* ORBextractor.nLevels (1..16) */
    int32_t ini_th_fast; /* ORBextractor.iniThFAST */
    int32_t min_th_fast; /* ORBextractor.minThFAST */
} eao_orb_cfg;

/* cv::KeyPoint POD mirror (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} eao_keypoint;

eao_status eao_orb_create(const eao_orb_cfg* cfg, eao_orb** out);
void eao_orb_destroy(eao_orb* h);

/* scale tables exposed by the reference getters (include/ORBextractor.h:63-83); arrays of nlevels floats/ints,
 * any pointer may be NULL */
eao_status eao_orb_tables(const eao_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                          int32_t* features_per_level);

/* Upper bound on keypoints returned per frame of the given size (the quad-tree can return a few more than
 * nfeatures: it stops at the first list size >= quota).  Use it to size kps/desc. */
eao_status eao_orb_max_keypoints(eao_orb* h, int32_t width, int32_t height, int32_t* cap);

/* One frame from host memory, synchronous: operator()(image, mask, keypoints, descriptors).
 * img: 8-bit single channel, `stride` bytes per row.  kps/desc: capacity `cap` keypoints (desc = cap*32 bytes).
 * *n receives the number of keypoints.  Empty image (img NULL or w/h <= 0) => *n = 0, outputs untouched
 * (reference src/ORBextractor.cc:1046-1047). */
eao_status eao_orb_extract(eao_orb* h, const uint8_t* img, int32_t width, int32_t height, int32_t stride,
                           eao_keypoint* kps, uint8_t* desc, int32_t cap, int32_t* n);

/* Batch of `batch` same-sized frames from host memory, synchronous.  Frame f starts at img + f*frame_stride.
 * kps: batch*cap entries, desc: batch*cap*32 bytes, n: batch counts. */
eao_status eao_orb_extract_batch(eao_orb* h, const uint8_t* img, int32_t width, int32_t height, int32_t stride,
                                 int64_t frame_stride, int32_t batch, eao_keypoint* kps, uint8_t* desc, int32_t cap,
                                 int32_t* n);

/* Streaming host API: what feeds Frame::ExtractORB are HOST images (reference src/Frame.cc:616-622, behind the Frame constructors
 * :192-194), and a sequence reader has the next batch ready while this one is extracted.  The handle owns `nslots` slots of PINNED
 * host memory: the producer writes up to `batch` frames into eao_orb_slot.frames (rows of `stride` bytes, frames `frame_stride`
 * apart), eao_orb_stream_submit is ASYNCHRONOUS (upload -> extraction -> download on three streams, chained by events: the upload
 * of one slot and the download of another overlap the extraction of a third), eao_orb_stream_wait blocks until the slot's results
 * are in its pinned kps / desc / n arrays (frame f: kps + f * cap, desc + f * cap * 32, n[f]).  Results are those of
 * eao_orb_extract_batch on the same frames.  A slot may be refilled once it has been waited for; submit on a slot that is still in
 * flight waits for it first.  One producer thread per handle. */
typedef struct {
    uint8_t* frames;          /* pinned: batch x height rows of `stride` bytes */
    int32_t stride;           /* row pitch of `frames` (>= width, a multiple of 64) */
    int64_t frame_stride;     /* stride * height */
    eao_keypoint* kps;        /* pinned: batch x cap */
    uint8_t* desc;            /* pinned: batch x cap x 32 */
    int32_t* n;               /* pinned: batch */
    int32_t cap;              /* = eao_orb_max_keypoints */
} eao_orb_slot;
eao_status eao_orb_stream_create(eao_orb* h, int32_t width, int32_t height, int32_t batch, int32_t nslots);
eao_status eao_orb_stream_slot(eao_orb* h, int32_t slot, eao_orb_slot* out);
eao_status eao_orb_stream_submit(eao_orb* h, int32_t slot, int32_t batch);
eao_status eao_orb_stream_wait(eao_orb* h, int32_t slot);

/* Same, all pointers device-resident (HBM), asynchronous on `stream` (a hipStream_t; NULL = the null stream, exactly as the
 * Hamming entry points read it): whatever the caller enqueues on that stream afterwards is ordered behind the extraction.
 * The main chain of the call's kernels is enqueued on `stream` ITSELF (a private side stream of the handle, forked from and
 * joined back into it with events, carries the part that runs beside it), so consecutive calls on one stream follow each
 * other without a cross-stream hop.  Calls on one handle are serialised with each other even when they come in on different
 * streams (they share the handle's pyramid and candidate scratch): a call on another stream than the previous one first drains
 * the device on the host (hipDeviceSynchronize -- the previous stream's handle is never used again, so its owner is free to
 * destroy it once its own work is done).  NULL is the legacy null stream (rounds 1 - 2a read NULL as "the handle's private
 * stream"; callers that relied on that pass their own stream now).  d_n: batch int32 on the device. */
eao_status eao_orb_extract_batch_device(eao_orb* h, const uint8_t* d_img, int32_t width, int32_t height,
                                        int32_t stride, int64_t frame_stride, int32_t batch, eao_keypoint* d_kps,
                                        uint8_t* d_desc, int32_t cap, int32_t* d_n, void* stream);

/* Pyramid level of frame `frame` of the LAST extract call, copied to host (tight rows, w*h bytes): backs the
 * public member ORBextractor::mvImagePyramid (include/ORBextractor.h:85).  which = 0 level image, 1 = its
 * 7x7 Gaussian-blurred copy.  Either of w/h/dst may be NULL to query sizes only. */
eao_status eao_orb_level(eao_orb* h, int32_t frame, int32_t level, int32_t which, int32_t* w, int32_t* hgt, uint8_t* dst);

/* ALL levels of frame `frame` of the last extraction in ONE call, each with a `border`-pixel BORDER_REFLECT_101 frame physically around it -- the form
 * upstream keeps in mvImagePyramid (EDGE_THRESHOLD = 19: src/ORBextractor.cc:1113-1128; read by Frame::ComputeStereoMatches only, src/Frame.cc:846, 936-953).
 * One launch writes the bordered levels into a pinned host block the handle owns; levels[l] (nlevels entries) receives the address of the level's pixel
 * (0, 0) inside it, its size and its row pitch (rows -border .. height + border - 1 and columns -border .. width + border - 1 are addressable around it).
 * The block is valid until the next eao_orb_pyramid call on the handle or its destruction.  After a host-API extraction (eao_orb_extract*) nothing but the
 * handle's own stream is synchronised; after eao_orb_extract_batch_device on a caller's stream the device is drained first. */
typedef struct {
    uint8_t* data;            /* pixel (0, 0) of the level */
    int32_t width, height;
    int32_t step;             /* bytes per row */
} eao_orb_level_view;
eao_status eao_orb_pyramid(eao_orb* h, int32_t frame, int32_t border, eao_orb_level_view* levels);
/* border >= 0: every eao_orb_extract_ref call also exports the frame's bordered pyramid behind the extraction, on the same stream and under the same single
 * synchronisation, and the eao_orb_pyramid call that follows only hands out the views; -1 (the default): off. */
eao_status eao_orb_set_keep_pyramid(eao_orb* h, int32_t border);

/* eao_orb_extract without the copy into caller arrays: *kps / *desc point into the pinned block the device wrote the frame's results to (*n keypoints,
 * *n x 32 descriptor bytes), valid until the next call on the handle.  What the class-surface adapter uses: its cv::KeyPoint vector and descriptor matrix
 * are filled straight from there. */
eao_status eao_orb_extract_ref(eao_orb* h, const uint8_t* img, int32_t width, int32_t height, int32_t stride, const eao_keypoint** kps,
                               const uint8_t** desc, int32_t* n);

/* Stage taps for parity tests (device -> host copies of intermediate products of the LAST extract call).
 * candidates: FAST corners of (frame, level) in reference order before the quad-tree, as (x, y, response)
 * float triples relative to the (16,16) level border origin; returns the count in *n (cap in triples). */
eao_status eao_orb_level_candidates(eao_orb* h, int32_t frame, int32_t level, float* xyr, int32_t cap, int32_t* n);

/* HIP-event timing of the pipeline stages, in milliseconds, in pipeline order:
 * [0] pyramid  [1] fast  [2] quadtree  [3] blur  [4] orient+describe  [5] whole pipeline,
 * AVERAGED over every extract call made since eao_orb_set_profiling(h, 1) (events are recorded on the stream the
 * kernels run on); reading blocks until the events have completed and restarts the average.  The blur runs on a side
 * stream concurrently with FAST + quad-tree, so the stage times may add up to more than [5].
 * eao_orb_lanes() returns the maximum number of concurrent slices a batch can be cut into (EAO_ORB_LANES, default 1). */
eao_status eao_orb_set_profiling(eao_orb* h, int32_t on);
eao_status eao_orb_last_timing(eao_orb* h, float ms[6]);
int32_t eao_orb_lanes(int32_t batch);

/* ------------------------------------------------------------------------------------------------
 * Hamming matching -- replaces ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:1649-1665) and
 * the candidate loops of the Search* routines (e.g. :83-115, :1402-1426)
 * ------------------------------------------------------------------------------------------------ */
/* D[i*nb + j] = popcount(A_i xor B_j), A: na x 32 bytes, B: nb x 32 bytes (host pointers, synchronous) */
eao_status eao_hamming_matrix(const uint8_t* A, int32_t na, const uint8_t* B, int32_t nb, uint16_t* D);

typedef struct {
    int32_t best;    /* smallest distance, 256 if no candidate */
    int32_t second;  /* second smallest (reference bestDist2), 256 if none */
    int32_t idx;     /* column of best, -1 if none; first column wins ties (strict '<' upstream) */
    int32_t idx2;    /* column of second, -1 if none */
} eao_best2;
/* per row of A the best two over the columns j with mask[i*nb + j] != 0 (mask NULL = all) */
eao_status eao_hamming_best2(const uint8_t* A, int32_t na, const uint8_t* B, int32_t nb, const uint8_t* mask,
                             eao_best2* out);

/* device-resident variants; `pairs` independent (A,B) problems laid out back to back (pair p: A + p*na*32, ...) */
eao_status eao_hamming_matrix_device(const uint8_t* d_A, int32_t na, const uint8_t* d_B, int32_t nb, int32_t pairs,
                                     uint16_t* d_D, void* stream);
eao_status eao_hamming_best2_device(const uint8_t* d_A, int32_t na, const uint8_t* d_B, int32_t nb, int32_t pairs,
                                    const uint8_t* d_mask, eao_best2* d_out, void* stream);

/* The consecutive-frame matcher of a device-resident batch (BASELINE configs[4]): pair f = (frame f - 1,
 * frame f) for every frame of the descriptor block d_desc [batch][cap][32] an eao_orb_extract_batch_device call left on
 * the device, with the per-frame keypoint counts read ON THE DEVICE from d_counts (that call's d_n) -- no host round trip
 * between extraction and matching.  Result of pair f, row i < count[f - 1]: d_out[f * cap + i] = the two nearest
 * descriptors of frame f (same definition as eao_hamming_best2).  Pair 0 matches the halo frame (d_halo_desc, halo_n rows:
 * the last frame of the previous shard) against frame 0, or is skipped when halo_n == 0.  Rows beyond a pair's count are
 * not written. */
eao_status eao_hamming_best2_sequence_device(const uint8_t* d_desc, int32_t cap, const int32_t* d_counts, int32_t batch,
                                             const uint8_t* d_halo_desc, int32_t halo_n, eao_best2* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Guided matching -- ORBmatcher::SearchByProjection, the two per-frame tracking variants
 *   (Frame&, const vector<MapPoint*>&, th)          reference src/ORBmatcher.cc:45-129   (TrackLocalMap)
 *   (Frame& Cur, const Frame& Last, th, bMono)      reference src/ORBmatcher.cc:1328-1472 (TrackWithMotionModel)
 * with Frame::GetFeaturesInArea / PosInGrid (src/Frame.cc:696-761) as the candidate generator.  The device builds,
 * for every query, the candidate list (keypoint index, Hamming distance) in the reference's candidate ORDER (grid cell
 * column-major, then insertion order) with every geometric / level / stereo gate applied; the greedy assignment
 * ("skip keypoints that already hold a map point", best / second-best, ratio test, rotation histogram) is replayed
 * on the host inside the library, query by query, exactly as upstream's loops run.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n;                   /* Frame::N */
    const float* kp_x;           /* mvKeysUn[i].pt.x */
    const float* kp_y;
    const int32_t* kp_octave;    /* mvKeysUn[i].octave */
    const float* kp_angle;       /* mvKeysUn[i].angle (rotation histogram) */
    const float* u_right;        /* mvuRight[i] */
    const uint8_t* descriptors;  /* mDescriptors: n x 32 */
    const uint8_t* occupied;     /* n: mvpMapPoints[i] != NULL && Observations() > 0 on entry (may be NULL = none) */
    float min_x, min_y, max_x, max_y;   /* mnMinX, mnMinY, mnMaxX, mnMaxY */
    float grid_inv_w, grid_inv_h;       /* mfGridElementWidthInv, mfGridElementHeightInv */
    int32_t grid_cols, grid_rows;       /* FRAME_GRID_COLS, FRAME_GRID_ROWS (64 x 48 upstream) */
    const float* scale_factors;         /* mvScaleFactors */
    int32_t nlevels;
    /* only read by the entry points that say so (may be 0 / NULL otherwise) */
    float log_scale_factor;             /* mfLogScaleFactor */
    const float* level_sigma2;          /* mvLevelSigma2 */
    const float* inv_level_sigma2;      /* mvInvLevelSigma2 */
} eao_frame_view;

/* SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th): map points that are in view.
 * Per map point: mTrackProjX/Y/XR, mTrackViewCos, mnTrackScaleLevel, GetDescriptor(), and skip[i] != 0 where upstream
 * `continue`s (!mbTrackInView || isBad()).  match_kp[i] receives the keypoint index the point was assigned to or -1;
 * the return value of the reference is *nmatches.  A keypoint assigned during the call counts as occupied afterwards
 * (upstream: F.mvpMapPoints[idx]->Observations() > 0 for a point of the local map). */
eao_status eao_search_by_projection_points(const eao_frame_view* F, int32_t n_mp, const float* proj_x, const float* proj_y,
                                           const float* proj_xr, const float* view_cos, const int32_t* pred_level,
                                           const uint8_t* mp_desc, const uint8_t* skip, float th, float nnratio,
                                           int32_t* match_kp, int32_t* nmatches);

/* SearchByProjection(Frame& Cur, const Frame& Last, th, bMono).  Per last-frame keypoint i: valid[i] != 0 where
 * LastFrame.mvpMapPoints[i] != NULL && !mvbOutlier[i]; Xw = GetWorldPos(); mp_desc = GetDescriptor(); last_octave =
 * LastFrame.mvKeys[i].octave; last_angle = LastFrame.mvKeysUn[i].angle.  Tcw_cur / Tcw_last: 16 floats row-major.
 * cur_match[k] (Cur->n entries) receives the last-frame index assigned to current keypoint k or -1 (after the rotation
 * consistency filter when check_orientation != 0). */
eao_status eao_search_by_projection_frames(const eao_frame_view* Cur, const float* Tcw_cur, const float* Tcw_last,
                                           int32_t n_last, const uint8_t* valid, const float* Xw, const uint8_t* mp_desc,
                                           const int32_t* last_octave, const float* last_angle, float fx, float fy, float cx,
                                           float cy, float mbf, float mb, float th, int32_t mono, int32_t check_orientation,
                                           int32_t* cur_match, int32_t* nmatches);

/* Map points as plain arrays (what the remaining searches read through MapPoint's accessors). */
typedef struct {
    int32_t n;
    const uint8_t* active;        /* n: 0 where upstream `continue`s before looking at the point (NULL pointer, isBad(),
                                     member of the caller's "already found" set, IsInKeyFrame(pKF), ...) */
    const float* Xw;              /* n*3  GetWorldPos() */
    const float* normal;          /* n*3  GetNormal()  (NULL for the searches that have no viewing-angle test) */
    const float* min_dist_inv;    /* n    GetMinDistanceInvariance() */
    const float* max_dist_inv;    /* n    GetMaxDistanceInvariance() */
    const float* max_dist;        /* n    mfMaxDistance, the numerator of PredictScale (src/MapPoint.cc:385-394) */
    const uint8_t* desc;          /* n*32 GetDescriptor() */
} eao_map_points;

/* DBoW2::FeatureVector of one frame: nodes in ascending id, each with its keypoint indices (std::map order). */
typedef struct {
    int32_t n_nodes;
    const uint32_t* node_id;      /* n_nodes, strictly ascending */
    const int32_t* node_start;    /* n_nodes + 1 offsets into index */
    const uint32_t* index;        /* keypoint indices, in the order of the node's vector */
} eao_feature_vector;

/* a13  SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th)
 * -- src/ORBmatcher.cc:290-403 (loop detection).  KF->occupied[k] != 0 where vpMatched[k] != NULL on entry; active[i] = 0
 * for bad points and members of spAlreadyFound.  Reads KF->log_scale_factor.  kp_match[k] receives the index of the point
 * assigned to keypoint k during the call, or -1. */
eao_status eao_search_by_projection_sim3(const eao_frame_view* KF, const float* Scw, float fx, float fy, float cx, float cy,
                                         const eao_map_points* pts, int32_t th, int32_t* kp_match, int32_t* nmatches);

/* a13  SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
 * -- src/ORBmatcher.cc:1474-1601 (relocalisation).  pts = pKF->GetMapPointMatches() (active[i] = 0 for NULL, bad or already
 * found), kf_angle[i] = pKF->mvKeysUn[i].angle; Cur->occupied = CurrentFrame.mvpMapPoints[k] != NULL.  cur_match[k]
 * receives the map-point index assigned to current keypoint k or -1 (after the rotation filter, whose histogram factor
 * is 1/HISTO_LENGTH here, :1486). */
eao_status eao_search_by_projection_kf(const eao_frame_view* Cur, const float* Tcw, float fx, float fy, float cx, float cy,
                                       const eao_map_points* pts, const float* kf_angle, float th, int32_t orb_dist,
                                       int32_t check_orientation, int32_t* cur_match, int32_t* nmatches);

/* a14  SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) -- src/ORBmatcher.cc:159-288 (mode 0) and
 *      SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) -- :522-655 (mode 1).
 * Side 1 is the keyframe whose map points drive the search: valid1[i] != 0 where its map point exists and is not bad.
 * mode 0: side 2 is the frame; a frame keypoint taken during the call is skipped afterwards; accept best <= TH_LOW.
 * mode 1: valid2[j] != 0 where pKF2's map point exists and is not bad; accept best < TH_LOW.
 * Both: best < nnratio * second.  match12[i] receives the side-2 index matched to side-1 keypoint i or -1. */
eao_status eao_search_by_bow(int32_t mode, int32_t n1, const uint8_t* desc1, const float* angle1, const uint8_t* valid1,
                             const eao_feature_vector* fv1, int32_t n2, const uint8_t* desc2, const float* angle2,
                             const uint8_t* valid2, const eao_feature_vector* fv2, float nnratio, int32_t check_orientation,
                             int32_t* match12, int32_t* nmatches);

/* a15  SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) -- src/ORBmatcher.cc:657-823, with
 * CheckDistEpipolarLine :140-157.  K1 / K2: keypoints (kp_x, kp_y, kp_octave, kp_angle, u_right, descriptors;
 * occupied[k] != 0 where the keyframe already has a map point at k); K2->level_sigma2 and K2->scale_factors are read.
 * F12: 9 floats row-major; (ex, ey) the epipole in image 2 (:663-670).  match12[i] = index in K2 or -1. */
eao_status eao_search_for_triangulation(const eao_frame_view* K1, const eao_feature_vector* fv1, const eao_frame_view* K2,
                                        const eao_feature_vector* fv2, const float* F12, float ex, float ey,
                                        int32_t only_stereo, int32_t check_orientation, int32_t* match12, int32_t* nmatches);

/* The same for ALL neighbour keyframes of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:211-290 calls SearchForTriangulation once per neighbour,
 * 10 or 20 per new keyframe) in one call: one upload of K1 and of every neighbour, one launch, one copy back.  K2s / fv2s: n_nb pointers; F12s: 9 floats
 * per neighbour; exs / eys: the epipoles; match12: n_nb x K1->n (row k = the single call's match12 against neighbour k), nmatches: n_nb.  Equal to n_nb
 * single calls, entry for entry (tests/test_gpu_search.py). */
eao_status eao_search_for_triangulation_batch(const eao_frame_view* K1, const eao_feature_vector* fv1, int32_t n_nb, const eao_frame_view* const* K2s,
                                              const eao_feature_vector* const* fv2s, const float* F12s, const float* exs, const float* eys,
                                              int32_t only_stereo, int32_t check_orientation, int32_t* match12, int32_t* nmatches);

/* a15  SearchForInitialization(Frame& F1, Frame& F2, vbPrevMatched, vnMatches12, windowSize) -- src/ORBmatcher.cc:405-520.
 * F1: n1 keypoints (octave, angle, descriptors); prev_matched: n1 x 2 floats, updated in place like vbPrevMatched. */
eao_status eao_search_for_initialization(int32_t n1, const int32_t* octave1, const float* angle1, const uint8_t* desc1,
                                         const eao_frame_view* F2, float* prev_matched, int32_t window, float nnratio,
                                         int32_t check_orientation, int32_t* match12, int32_t* nmatches);

/* a15  the search half of Fuse(KeyFrame*, const vector<MapPoint*>&, th) -- src/ORBmatcher.cc:825-975 (use_sim3 = 0:
 * pose = Rcw (9) | tcw (3) | Ow (3) as 15 floats; reads KF->inv_level_sigma2 and the stereo / monocular chi2 gates) and of
 * Fuse(KeyFrame*, cv::Mat Scw, vpPoints, th, vpReplacePoint) -- :977-1100 (use_sim3 = 1: pose = Scw, 16 floats).
 * best_kp[i] receives the keypoint the point would be fused into (best distance <= TH_LOW) or -1; replacing or adding the
 * observation is the caller's part (it mutates the map), in index order, re-checking isBad() as upstream would. */
eao_status eao_fuse_search(const eao_frame_view* KF, int32_t use_sim3, const float* pose, float fx, float fy, float cx, float cy,
                           float bf, const eao_map_points* pts, float th, int32_t* best_kp, int32_t* nfused);

/* The same points against ALL target keyframes of LocalMapping::SearchInNeighbors (src/LocalMapping.cc:458-520 calls Fuse once per target) in one call: the
 * points travel once, every target's frame and windows in the same copy, one synchronisation.  KFs: n_kf pointers; poses: 15 (use_sim3 = 0) or 16 floats per
 * target; best_kp: n_kf x pts->n; nfused: n_kf.  Equal to n_kf single calls ON THE SAME MAP STATE: an earlier target's fusions change the map, so the caller,
 * applying the targets in order, re-checks isBad() and IsInKeyFrame(pKFi) (upstream's `continue`s, src/ORBmatcher.cc:851-861) before it uses a later
 * target's candidate -- and searches again, against the remaining targets, every point whose DESCRIPTOR an applied fusion changed (pMPinKF->Replace(pMP) ends in
 * pMP->ComputeDistinctiveDescriptors(), src/MapPoint.cc:177-215): eaofusion::ORBmatcher::FuseBatch does both. */
eao_status eao_fuse_search_batch(int32_t n_kf, const eao_frame_view* const* KFs, int32_t use_sim3, const float* poses, float fx, float fy, float cx, float cy,
                                 float bf, const eao_map_points* pts, float th, int32_t* best_kp, int32_t* nfused);

/* a15  SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) -- src/ORBmatcher.cc:1102-1326.  pts1 / pts2: the map
 * points of the two keyframes by keypoint index (active = exists, not bad, not already matched); T1w / T2w: 16 floats.
 * match12[i] = keypoint of KF2 whose map point agrees in both directions, or -1 (already matched entries are not touched
 * by the caller). */
eao_status eao_search_by_sim3(const eao_frame_view* K1, const float* T1w, const eao_map_points* pts1, const eao_frame_view* K2,
                              const float* T2w, const eao_map_points* pts2, float fx, float fy, float cx, float cy, float s12,
                              const float* R12, const float* t12, float th, int32_t* match12, int32_t* nfound);

/* ---- keyframe handles (round 5): upload once, search many -------------------------------------------------------------------
 * LocalMapping::CreateNewMapPoints / SearchInNeighbors (reference src/LocalMapping.cc:211-290, 458-520) and LoopClosing search the SAME keyframes again and
 * again; the entry points above upload every frame with every call.  An eao_keyframe keeps what the guided searches read of a KeyFrame -- or of a Frame that is
 * a search target -- in HBM: the view's keypoints / octaves / angles / uRight / descriptors, the grid order GetFeaturesInArea walks (built once), the scale
 * tables, the DBoW2 feature vector (fv, may be NULL: the handle then serves the window searches only) and the occupancy view->occupied (NULL = none).  The
 * arrays are copied: the caller's may go away.  A handle is immutable but for its occupancy; searches only read it, so any number of threads may search the
 * same handle (each thread has its own stream and scratch).  `_kf` variants return exactly what their host-array counterparts above return on the same data. */
typedef struct eao_keyframe eao_keyframe;
eao_status eao_keyframe_create(const eao_frame_view* view, const eao_feature_vector* fv, eao_keyframe** out);
/* occupied[k] != 0 where the keyframe holds a map point at keypoint k (GetMapPoint(k) != NULL) -- what SearchForTriangulation skips on both sides
 * (src/ORBmatcher.cc:696-700, 712-716); NULL = no keypoint occupied.  Call it when map points were added to / erased from the keyframe -- from the thread that owns the
 * keyframe's map points (LocalMapping), and not while another thread searches this handle: the update is the one writer, ordered by the caller like upstream's mMutexFeatures. */
eao_status eao_keyframe_update_points(eao_keyframe* kf, const uint8_t* occupied);
void eao_keyframe_destroy(eao_keyframe* kf);
int32_t eao_keyframe_size(const eao_keyframe* kf);      /* number of keypoints, -1 for NULL */

/* a14 on handles, selection included ON THE DEVICE (a wavefront per common vocabulary node; two launches, one small copy back): arguments as
 * eao_search_by_bow.  valid1 / valid2 change with the map and travel per call (n bytes each). */
eao_status eao_kf_search_by_bow(int32_t mode, const eao_keyframe* kf1, const uint8_t* valid1, const eao_keyframe* kf2, const uint8_t* valid2, float nnratio,
                                int32_t check_orientation, int32_t* match12, int32_t* nmatches);
/* a15 on handles, all neighbours of LocalMapping::CreateNewMapPoints in one call, selection on the device: arguments as eao_search_for_triangulation_batch
 * (n_nb = 1: the single search); the occupancy of both sides is the handles' (eao_keyframe_update_points).  match12: n_nb x size(kf1). */
eao_status eao_kf_search_for_triangulation(const eao_keyframe* kf1, int32_t n_nb, const eao_keyframe* const* kf2s, const float* F12s, const float* exs,
                                           const float* eys, int32_t only_stereo, int32_t check_orientation, int32_t* match12, int32_t* nmatches);
/* a15 on handles, all targets of LocalMapping::SearchInNeighbors (or one: n_kf = 1) in one call: projection, window walk and gated best candidate per
 * (target, point) on the device; arguments as eao_fuse_search_batch.  The points change from call to call and travel each time (70 bytes per point). */
eao_status eao_kf_fuse_search(int32_t n_kf, const eao_keyframe* const* kfs, int32_t use_sim3, const float* poses, float fx, float fy, float cx, float cy,
                              float bf, const eao_map_points* pts, float th, int32_t* best_kp, int32_t* nfused);
/* a13 / a15: the list-based searches with the searched frame taken from a handle (only the queries travel; the selection is replayed on the host as before).
 * `occupied` replaces view->occupied of the host-array form (it differs from search to search: vpMatched, CurrentFrame.mvpMapPoints); NULL = none. */
eao_status eao_kf_search_by_projection_sim3(const eao_keyframe* kf, const uint8_t* occupied, const float* Scw, float fx, float fy, float cx, float cy,
                                            const eao_map_points* pts, int32_t th, int32_t* kp_match, int32_t* nmatches);
eao_status eao_kf_search_by_projection_kf(const eao_keyframe* cur, const uint8_t* occupied, const float* Tcw, float fx, float fy, float cx, float cy,
                                          const eao_map_points* pts, const float* kf_angle, float th, int32_t orb_dist, int32_t check_orientation,
                                          int32_t* cur_match, int32_t* nmatches);
eao_status eao_kf_search_for_initialization(int32_t n1, const int32_t* octave1, const float* angle1, const uint8_t* desc1, const eao_keyframe* f2,
                                            float* prev_matched, int32_t window, float nnratio, int32_t check_orientation, int32_t* match12, int32_t* nmatches);
eao_status eao_kf_search_by_sim3(const eao_keyframe* k1, const float* T1w, const eao_map_points* pts1, const eao_keyframe* k2, const float* T2w,
                                 const eao_map_points* pts2, float fx, float fy, float cx, float cy, float s12, const float* R12, const float* t12, float th,
                                 int32_t* match12, int32_t* nfound);

/* ---- f1  Frame glue either side of the matcher ------------------------------------------------------------------ */

/* What Frame::isInFrustum reads of the frame: mTcw (16 floats row-major: mRcw, mtcw), mOw, the static intrinsics and image
 * bounds (fx, fy, cx, cy, mbf, mnMinX .. mnMaxY) and mfLogScaleFactor. */
typedef struct {
    float Tcw[16];
    float Ow[3];
    float fx, fy, cx, cy, mbf;
    float min_x, max_x, min_y, max_y;
    float log_scale_factor;
} eao_frustum_frame;

/* Frame::isInFrustum(MapPoint*, viewingCosLimit) -- src/Frame.cc:638-695 -- for every point of pts, i.e. the projection
 * loop of Tracking::SearchLocalPoints (src/Tracking.cc:2612-2626; the caller keeps its own skip rules: mnLastFrameSeen,
 * isBad()).  pts: Xw, normal, min_dist_inv / max_dist_inv (Get{Min,Max}DistanceInvariance()), max_dist (mfMaxDistance, the
 * numerator of MapPoint::PredictScale, src/MapPoint.cc:385-394 -- not clamped in this fork); active and desc are not read.
 * in_view[i] = the return value = mbTrackInView; proj_x / proj_y / proj_xr / view_cos / pred_level receive mTrackProjX,
 * mTrackProjY, mTrackProjXR, mTrackViewCos, mnTrackScaleLevel where in_view[i] = 1 and keep the caller's values elsewhere
 * (upstream leaves those members untouched).  The outputs are exactly the inputs of eao_search_by_projection_points. */
eao_status eao_frame_is_in_frustum(const eao_frustum_frame* F, const eao_map_points* pts, float viewing_cos_limit, uint8_t* in_view,
                                   float* proj_x, float* proj_y, float* proj_xr, float* view_cos, int32_t* pred_level);

/* Frame::AssignFeaturesToGrid() with PosInGrid -- src/Frame.cc:597-614, 751-761 -- over mvKeysUn.  mGrid comes back as a CSR:
 * cell c = ix * rows + iy is mGrid[ix][iy]; cell_start has cols * rows + 1 entries; items[cell_start[c] .. cell_start[c+1])
 * are the keypoint indices of the cell in push_back (= ascending) order.  items needs room for n entries. */
eao_status eao_assign_features_to_grid(int32_t n, const float* kp_x, const float* kp_y, float min_x, float min_y, float grid_inv_w,
                                       float grid_inv_h, int32_t cols, int32_t rows, int32_t* cell_start, int32_t* items);

/* Frame::ComputeStereoFromRGBD(imDepth) -- src/Frame.cc:1016-1037.  kp_x / kp_y: mvKeys[i].pt (distorted), kpu_x:
 * mvKeysUn[i].pt.x; depth: the CV_32F depth image (`pitch` floats per row), a host pointer or, with depth_on_device != 0,
 * a device pointer (a depth map that is already resident).  u_right / out_depth receive mvuRight / mvDepth (-1 where the
 * depth is not positive). */
eao_status eao_compute_stereo_from_rgbd(int32_t n, const float* kp_x, const float* kp_y, const float* kpu_x, const float* depth,
                                        int32_t width, int32_t height, int32_t pitch, int32_t depth_on_device, float mbf,
                                        float* u_right, float* out_depth);

/* Frame::UndistortKeyPoints() -- src/Frame.cc:773-806: mvKeysUn[i].pt = cv::undistortPoints(mvKeys[i].pt, mK, mDistCoef, cv::Mat(), mK) (OpenCV 3.x:
 * five fixed-point iterations of the inverse distortion model in double, result rounded to float).  dist_coef: mDistCoef as the reference fills it
 * (src/Tracking.cc:90-101): k1, k2, p1, p2 and, with n_coef == 5, k3.  dist_coef[0] == 0 (or n_coef == 0): the coordinates are copied, as upstream copies
 * mvKeys (:775-779).  One thread per keypoint on the device; host arrays in and out. */
eao_status eao_undistort_keypoints(int32_t n, const float* kp_x, const float* kp_y, float fx, float fy, float cx, float cy,
                                   const float* dist_coef, int32_t n_coef, float* out_x, float* out_y);
/* Frame::ComputeImageBounds(imLeft) -- src/Frame.cc:808-842: bounds = { mnMinX, mnMaxX, mnMinY, mnMaxY } from the undistorted image corners, clamped to
 * the image ("make sure it is inside image", :828-832). */
eao_status eao_compute_image_bounds(int32_t cols, int32_t rows, float fx, float fy, float cx, float cy, const float* dist_coef, int32_t n_coef,
                                    float* bounds);

/* f4  Frame::ComputeStereoMatches() -- src/Frame.cc:841-1013.  `left` / `right` are the two extractor handles
 * (mpORBextractorLeft / Right) right after they extracted the stereo pair: the image pyramids of frame `frame` of their
 * last batch are still on the device and are read in place (the 19-px reflect-101 border of mvImagePyramid is evaluated
 * on the fly).  kps / desc: mvKeys + mDescriptors and mvKeysRight + mDescriptorsRight.  mb, mbf: Frame::mb, mbf.
 * u_right[i] / depth[i] receive mvuRight / mvDepth (-1 = no match), after the median-based outlier rejection. */
eao_status eao_compute_stereo_matches(eao_orb* left, eao_orb* right, int32_t frame, int32_t nl, const eao_keypoint* kps_l,
                                      const uint8_t* desc_l, int32_t nr, const eao_keypoint* kps_r, const uint8_t* desc_r,
                                      float mb, float mbf, float* u_right, float* depth);

/* f4  MapPoint::ComputeDistinctiveDescriptors() -- src/MapPoint.cc:242-307, batched over map points.  Set s holds the
 * descriptors of the (non-bad) keyframes observing point s, in std::map<KeyFrame*, size_t> iteration order:
 * desc[32 * k] for set_start[s] <= k < set_start[s + 1].  best[s] receives the position inside the set of the descriptor
 * with the least median Hamming distance to the others (median = sorted[(int)(0.5 * (N - 1))], the distance to itself
 * included; first index wins ties), or -1 for an empty set. */
eao_status eao_distinctive_descriptors(int32_t n_sets, const int32_t* set_start, const uint8_t* desc, int32_t* best);

/* ------------------------------------------------------------------------------------------------
 * Optimizer::PoseOptimization(Frame*) -- reference include/Optimizer.h:56, src/Optimizer.cc:325-673
 * (point / stereo edges and the plane edges of src/Optimizer.cc:456-535)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n;               /* matched map points (pFrame->mvpMapPoints[i] != NULL), in index order */
    const float* Tcw;        /* 16 floats, row-major 4x4: pFrame->mTcw */
    const float* Xw;         /* n*3: MapPoint::GetWorldPos() */
    const float* obs;        /* n*3: mvKeysUn[i].pt.x, .pt.y, mvuRight[i]  (mvuRight < 0 => monocular edge) */
    const float* inv_sigma2; /* n: mvInvLevelSigma2[mvKeysUn[i].octave] */
    float fx, fy, cx, cy, bf;
    /* plane edges (src/Optimizer.cc:456-535, 626-658; src/g2oAddition/EdgePlane.h, Plane3D.h): one per associated map
     * plane (pFrame->mvpMapPlanes[i] != NULL), in index order; n_planes = 0 and NULL pointers when there are none */
    int32_t n_planes;           /* at most 32 */
    const float* plane_world;   /* n_planes*4: MapPlane::GetWorldPos() */
    const float* plane_obs;     /* n_planes*4: pFrame->mvPlaneCoefficients[i] */
    const uint8_t* plane_seen;  /* n_planes:   MapPlane::mbSeen (unseen planes get twice the information) */
} eao_pose_problem;

typedef struct {
    float Tcw[16];      /* optimised pose as Converter::toCvMat would deliver it */
    uint8_t* outlier;   /* n flags: pFrame->mvbOutlier (caller-allocated) */
    int32_t n_inliers;  /* return value of PoseOptimization: nInitialCorrespondences - nBad */
    int32_t lm_iterations; /* outer LM iterations executed over the 4 rounds */
    uint8_t* plane_outlier; /* n_planes flags: pFrame->mvbPlaneOutlier (caller-allocated; may be NULL when n_planes = 0) */
} eao_pose_result;

eao_status eao_pose_optimization(const eao_pose_problem* p, eao_pose_result* r);

/* `n` independent PoseOptimization problems in one call: the candidate loop of Tracking::Relocalization
 * (reference src/Tracking.cc:2786-2940: the first Optimizer::PoseOptimization (:2885) of every candidate keyframe whose PnP converged)
 * and offline replays.  One workgroup per frame, one upload, one launch, one synchronisation; results[i] is
 * bit-identical to what eao_pose_optimization(&problems[i], &results[i]) returns.  Frames with more than 2048
 * correspondences are run one by one through that entry point. */
eao_status eao_pose_optimization_batch(const eao_pose_problem* problems, int32_t n, eao_pose_result* results);

/* ------------------------------------------------------------------------------------------------
 * Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*) -- reference include/Optimizer.h:55,
 * src/Optimizer.cc:675-1138.  The adapter flattens the local window: cameras in ascending KeyFrame::mnId
 * (free = lLocalKeyFrames, fixed = lFixedCameras or mnId == 0), points in ascending MapPoint::mnId, edges
 * in insertion order.  At most one edge per (camera, point).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_cams, n_points, n_edges;
    const float* cam_Tcw;         /* n_cams*16 */
    const uint8_t* cam_fixed;     /* n_cams */
    const float* points;          /* n_points*3 */
    const int32_t* edge_cam;      /* n_edges */
    const int32_t* edge_point;    /* n_edges */
    const float* edge_obs;        /* n_edges*3: u, v, ur (ur < 0 => monocular) */
    const float* edge_inv_sigma2; /* n_edges */
    float fx, fy, cx, cy, bf;
    int32_t its_first, its_second; /* 5 and 10 upstream (src/Optimizer.cc:966,1027) */
} eao_ba_problem;

typedef struct {
    float* cam_Tcw;        /* n_cams*16 out */
    float* points;         /* n_points*3 out */
    uint8_t* edge_outlier; /* n_edges: observation to erase (chi2 > 5.991/7.815 or depth <= 0), src/Optimizer.cc:1040-1068 */
    int32_t iters[2];      /* outer LM iterations executed in the two passes */
    int32_t aborted;       /* 1 when *stop was set on entry (src/Optimizer.cc:961-963): outputs = inputs */
    double chi2[2];        /* robust chi2 after each pass */
} eao_ba_result;

/* stop may be NULL; it is polled on the host between LM iterations like g2o's forceStopFlag */
eao_status eao_local_ba(const eao_ba_problem* p, const volatile uint8_t* stop, eao_ba_result* r);

/* The same call for n INDEPENDENT windows at once (BASELINE configs[4]: the 25 local-BA windows of a batched sequence; in
 * a live system: the windows of several maps / agents).  Every window is exactly one eao_local_ba call -- same inputs, same
 * outputs, same LM schedule (src/Optimizer.cc:675-1138 per window; independence per SURVEY.md s8e) -- but the window is the
 * z dimension of every kernel launch, so one enqueue serves them all.  Windows the batched path does not take (more than 30
 * free keyframes, nothing to optimise) run one after the other inside the same call.  `stop` is shared by all windows.
 * eao_last_lm_timing then reports the device time of the whole batch and the linearisations summed over the windows. */
eao_status eao_local_ba_batch(const eao_ba_problem* problems, int32_t n, const volatile uint8_t* stop, eao_ba_result* results);

/* f3  Optimizer::BundleAdjustment(vpKFs, vpMP, vpMPl, nIterations, pbStopFlag, nLoopKF, bRobust) --
 * src/Optimizer.cc:55-323: the same flattening as eao_local_ba (cameras in ascending mnId with
 * cam_fixed = (mnId == 0), points in ascending mnId, points without an edge simply stay where they are),
 * ONE optimize(p->its_first) call (its_second is ignored), Huber kernels (delta sqrt(5.99) / sqrt(7.815), :94-95) only
 * when robust != 0, no outlier pass: r->edge_outlier may be NULL and comes back all zero, r->iters[1] = 0.
 * Up to 30 free keyframes run on the register-tile solver of eao_local_ba; beyond that (whole maps after a loop
 * closure, up to 8192 free keyframes) the reduced camera system is kept as block-sparse 64 x 64 tiles in HBM, in a nested-dissection
 * order, and factorised by the whole chip (csrc/gba.hip, k_bal_*); oversized windows of eao_local_ba take the same path. */
eao_status eao_bundle_adjustment(const eao_ba_problem* p, int32_t robust, const volatile uint8_t* stop, eao_ba_result* r);

/* The MapPlane vertices / EdgePlane edges of the same function (src/Optimizer.cc:203-252; src/g2oAddition/VertexPlane.h,
 * EdgePlane.h, Plane3D.h): every non-bad map plane is a marginalised 3-dof vertex (Plane3D::oplus), every observation by
 * a keyframe of the graph an edge with information diag(3282.8, 3282.8, 1e4), ALWAYS a Huber kernel (delta sqrt(300))
 * and g2o's numeric Jacobians on both vertices.  planes == NULL or n_planes == 0: same as eao_bundle_adjustment. */
typedef struct {
    int32_t n_planes;
    const float* plane_world;     /* n_planes*4: MapPlane::GetWorldPos() */
    int32_t n_pedges;
    const int32_t* pedge_plane;   /* n_pedges: index into plane_world */
    const int32_t* pedge_cam;     /* n_pedges: index into p->cam_Tcw */
    const float* pedge_obs;       /* n_pedges*4: KeyFrame::mvPlaneCoefficients[observation index] */
} eao_ba_planes;
eao_status eao_bundle_adjustment_planes(const eao_ba_problem* p, const eao_ba_planes* planes, int32_t robust, const volatile uint8_t* stop,
                                        eao_ba_result* r, float* planes_out /* n_planes*4: Converter::toCvMat(vPlane->estimate()) */);

/* (diagnostic, host only -- needs no device) The elimination order, tile structure and launch schedule the map-scale path derives from a covisibility pattern
 * (round 6; csrc/gba.hip gba_build_plan: one level of nested dissection of the keyframe graph, the block form of the symbolic phase of the reference's
 * SimplicialLDLT + AMD ordering, Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:95-112).  pair_a[k] <= pair_b[k]: the covisible pairs of the n_free free keyframes,
 * the diagonal pairs (i, i) included.  force_segments: 0 = the library's choice, 1 = natural order, p > 1 = p segments.  Arrays the caller passes as NULL (or whose
 * capacity is too small) are left out; `info` always comes back, so a second call can size them.  tests/test_gba_plan.py replays the schedule on the CPU against a
 * dense solve. */
typedef struct {
    int32_t n_free, n_rows /* N: padded system, a multiple of 64; row N = the right-hand side */, n_tile_rows /* T = N / 64 + 1 */, n_panels /* N / 32 */, n_tiles;
    int32_t n_segments, n_separator, separator_start /* first row of the separator block */, rcm /* 1: the line is a reverse Cuthill-McKee order */, bandwidth;
    int32_t chain_natural /* factorisation launches of the natural order */, chain_estimate /* this plan's launches, factorisation + back substitution */;
    int32_t n_launches, n_work /* record pairs */, n_diag, n_sb, n_sb_launches;
} eao_gba_plan_info;
eao_status eao_bundle_adjustment_plan(int32_t n_free, int32_t n_pairs, const int32_t* pair_a, const int32_t* pair_b, int32_t force_segments, eao_gba_plan_info* info,
                                      int32_t* row_of /* n_free */, int32_t* tile_map /* T * T */, int32_t cap_tile_map, int32_t* work /* 8 per record pair */, int32_t cap_work,
                                      int32_t* launches /* 4 each: work offset, count, diag offset, count */, int32_t cap_launches, int32_t* diag_list, int32_t cap_diag,
                                      int32_t* sb /* 4 each: first column, width, chunk lo, chunk hi */, int32_t cap_sb, int32_t* sb_launches /* 3 each: offset, count, grid.x */,
                                      int32_t cap_sb_launches);

/* LM trace of the last eao_local_ba / eao_pose_optimization call made by this thread (for parity tests):
 * up to cap entries of (lambda after the iteration, robust chi2, trials). Returns the count in *n. */
eao_status eao_last_lm_trace(double* lambda, double* chi2, int32_t* trials, int32_t cap, int32_t* n);

/* HIP-event time (ms) spent in device work by the last eao_local_ba / eao_pose_optimization on this thread,
 * and the number of linearisations (buildSystem calls) it made. */
eao_status eao_last_lm_timing(float* device_ms, int32_t* linearizations);

/* ---- f1, second half: the device-resident tracked frame ------------------------------------------------------------
 * Tracking::TrackLocalMap's data path (reference src/Tracking.cc:1717-2231, 2587-2641) chained on the device: the
 * extractor's device outputs of ONE frame go through Frame::ComputeStereoFromRGBD + AssignFeaturesToGrid (src/Frame.cc:
 * 599-614, 751-761, 1016-1037), Frame::isInFrustum over the local map (:638-695; viewingCosLimit 0.5 as in
 * Tracking::SearchLocalPoints), ORBmatcher::SearchByProjection(Frame&, local map points, th) (src/ORBmatcher.cc:45-137)
 * and Optimizer::PoseOptimization (src/Optimizer.cc:325-673, point edges) without returning to the host in between; one
 * copy brings back the pose, mvpMapPoints, mvbOutlier (and mvuRight / mvDepth).  Results are those of the host-hop calls
 * eao_compute_stereo_from_rgbd -> eao_frame_is_in_frustum -> eao_search_by_projection_points -> eao_pose_optimization on
 * the same data, bit for bit (tests/test_gpu_track.py).  Distortion-free camera (mvKeysUn = mvKeys), as the reference's
 * RGB-D configuration (ros_test/config/TUM3.yaml:13-16). */
typedef struct eao_tracker eao_tracker;
typedef struct {
    float fx, fy, cx, cy, mbf;                 /* Frame::fx .. mbf */
    float min_x, max_x, min_y, max_y;          /* mnMinX .. mnMaxY */
    int32_t grid_cols, grid_rows;              /* FRAME_GRID_COLS x FRAME_GRID_ROWS = 64 x 48 (include/Frame.h:89-90) */
    int32_t nlevels;
    const float* scale_factors;                /* nlevels: mvScaleFactors (copied) */
    const float* inv_level_sigma2;             /* nlevels: mvInvLevelSigma2 (copied) */
    float log_scale_factor;                    /* mfLogScaleFactor */
    int32_t max_keypoints;                     /* per frame, <= 4096; >= eao_orb_max_keypoints of the extractor */
    int32_t max_map_points;                    /* of a local map, <= 16384 */
} eao_tracker_cfg;
eao_status eao_tracker_create(const eao_tracker_cfg* cfg, eao_tracker** out);
void eao_tracker_destroy(eao_tracker* h);
/* The local map (Tracking::mvpLocalMapPoints) as plain arrays; uploaded once and kept in HBM until replaced.  active[i] = 0
 * where upstream skips the point (isBad()). */
eao_status eao_tracker_set_local_map(eao_tracker* h, const eao_map_points* pts);
typedef struct {
    float Tcw[16];            /* the optimised pose (the prior when fewer than 3 correspondences, src/Optimizer.cc:453-454) */
    int32_t n_keypoints;      /* N of the frame */
    int32_t n_matches;        /* return value of SearchByProjection */
    int32_t n_edges;          /* nInitialCorrespondences of PoseOptimization */
    int32_t n_inliers;        /* its return value */
    int32_t* kp_map_point;    /* caller array [max_keypoints]: mvpMapPoints as map-point index or -1 */
    uint8_t* kp_outlier;      /* caller array [max_keypoints]: mvbOutlier */
    float* kp_u_right;        /* optional caller arrays [max_keypoints]: mvuRight, mvDepth (NULL: not copied) */
    float* kp_depth;
    uint8_t* map_in_view;     /* optional caller array [max_map_points] (NULL: not copied): Frame::isInFrustum(pMP, 0.5) of every ACTIVE
                               * local map point -- what Tracking::SearchLocalPoints needs for IncreaseVisible() (src/Tracking.cc:2621-2625);
                               * the caller ignores the entries of points a prior match names (upstream never projects those) */
} eao_track_result;
/* d_kps / d_desc / d_n: ONE frame's slice of the device outputs of eao_orb_extract_batch_device (d_n points at that frame's
 * count); d_depth: the float depth image on the device (rows of depth_pitch floats) or NULL (monocular: mvuRight = -1);
 * Tcw_prior: the pose the frame enters TrackLocalMap with; prior_kp_map_point: mvpMapPoints as it stands (host array
 * [max_keypoints], NULL = none) -- those keypoints are occupied, their map points are not searched again, and they are
 * edges of the pose optimisation.  Entries: -1 = no map point; m in [0, n) = point m of the local map -- if that point is
 * inactive (isBad()) the prior is DROPPED and the keypoint is free, as upstream sets such an entry to NULL
 * (src/Tracking.cc:2596-2599), and the returned kp_map_point shows it; -2 = a map point that is not in the local map
 * (a temporal point of the RGB-D odometry, a point the local map lost): the keypoint stays occupied and its edge is built
 * from prior_kp_Xw[3k..3k+2] (host array [3 * max_keypoints], may be NULL when no entry is -2), kp_map_point returns -2;
 * anything else (an index >= n, e.g. a table older than the last eao_tracker_set_local_map) fails with EAO_ERR_INVALID before
 * any kernel runs.  `stream`: the stream the extraction was enqueued on -- the chain is enqueued on
 * that stream itself (ordered behind the extraction without an event hand-over); the call returns when the results are on the host. */
eao_status eao_tracker_track_local_map(eao_tracker* h, const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                       const float* d_depth, int32_t depth_pitch, int32_t width, int32_t height, const float* Tcw_prior,
                                       const int32_t* prior_kp_map_point, const float* prior_kp_Xw, float th, float nnratio, eao_track_result* out,
                                       void* stream);

/* One-shot options of the NEXT eao_tracker_track_* call on the handle (round 5; NULL or never called: defaults).
 *  - plane edges: in this fork Map::AssociatePlanesByBoundary runs BEFORE Optimizer::PoseOptimization in all three stages (src/Tracking.cc:1587, before :2181,
 *    TrackLocalMap), so the frame's associated planes are edges of the optimisation (src/Optimizer.cc:456-535, 626-658).  A caller whose frame carries them
 *    hands them over exactly as in eao_pose_problem (n_planes <= 32; plane_world = MapPlane::GetWorldPos(), plane_obs = mvPlaneCoefficients[i], plane_seen =
 *    mbSeen) and gets mvbPlaneOutlier back in plane_outlier (caller array, valid until the track call returns); the chained pose optimisation is then the
 *    plane instantiation of the same kernel -- same result as eao_pose_optimization with those planes on the chain's correspondences.
 *  - min_matches (motion-model / reference-keyframe stages): when the search returns fewer matches, the pose optimisation and the outlier discard are
 *    SKIPPED -- upstream repeats the search with 2 * th (src/Tracking.cc:1756-1763) or returns false (:1580-1581) before it optimises anything; the result then
 *    carries the search's tables, n_edges = n_inliers = 0 and the prior pose, and the caller touches no map point.  0 = always optimise. */
typedef struct {
    int32_t min_matches;
    int32_t n_planes;
    const float* plane_world;      /* n_planes*4 */
    const float* plane_obs;        /* n_planes*4 */
    const uint8_t* plane_seen;     /* n_planes */
    uint8_t* plane_outlier;        /* out: n_planes */
} eao_track_options;
eao_status eao_tracker_set_options(eao_tracker* h, const eao_track_options* opt);
/* The camera's lens distortion (mDistCoef; round 5).  With a non-zero k1 the frame set-up of every eao_tracker_track_* call undistorts the keypoints on the
 * device first (Frame::UndistortKeyPoints, as eao_undistort_keypoints): mvuRight, the grid, the searches and the pose edges then read mvKeysUn, the depth image
 * is looked up at the DISTORTED keypoint -- upstream's own split (src/Frame.cc:1016-1037).  cfg.min_x .. max_y are the caller's mnMinX .. mnMaxY
 * (eao_compute_image_bounds).  Persistent for the handle; n_coef == 0 or dist_coef[0] == 0: distortion-free (the default). */
eao_status eao_tracker_set_distortion(eao_tracker* h, const float* dist_coef, int32_t n_coef);

/* Tracking::TrackWithMotionModel's data path (reference src/Tracking.cc:1717-2231) on the same chain, ahead of TrackLocalMap: the frame set-up as
 * above, ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) (src/ORBmatcher.cc:1328-1472, with this fork's rotation-histogram
 * factor HISTO_LENGTH / 360, :1337) against the LAST frame's map points, Optimizer::PoseOptimization from the predicted pose, and -- when
 * discard_outliers != 0 -- the "Discard outliers" loop of src/Tracking.cc:2188-2207; one copy back.  The last frame travels as host arrays per
 * last-frame keypoint i (n_last <= max_keypoints entries): valid[i] != 0 where mvpMapPoints[i] != NULL && !mvbOutlier[i]; Xw = GetWorldPos();
 * mp_desc = GetDescriptor() (32 bytes); last_octave = mvKeys[i].octave; last_angle = mvKeysUn[i].angle.  Tcw_cur: the predicted pose
 * (mVelocity * mLastFrame.mTcw, :1726), Tcw_last: mLastFrame.mTcw; 16 floats row-major each, finite (EAO_ERR_INVALID otherwise).
 * Result (zero-initialise the struct and set the array pointers; map_in_view is not written): kp_map_point[k] = the LAST-FRAME INDEX whose map point
 * keypoint k took, or -1; n_matches = the search's return value (what upstream tests against 20 before it retries with 2 th -- the caller repeats the
 * call); n_edges = correspondences of the pose optimisation; n_inliers = n_edges minus the outliers = the matches left after the discard;
 * kp_outlier = mvbOutlier after PoseOptimization (all zero after the discard).  A keypoint a point has taken counts as occupied for the points behind
 * it, as for map points with Observations() > 0 (the same rule eao_search_by_projection_frames applies).  Same results as
 * eao_compute_stereo_from_rgbd -> eao_search_by_projection_frames -> eao_pose_optimization on the same data. */
eao_status eao_tracker_track_with_motion_model(eao_tracker* h, const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                               const float* d_depth, int32_t depth_pitch, int32_t width, int32_t height, const float* Tcw_cur,
                                               const float* Tcw_last, int32_t n_last, const uint8_t* valid, const float* Xw, const uint8_t* mp_desc,
                                               const int32_t* last_octave, const float* last_angle, float th, int32_t mono, int32_t check_orientation,
                                               int32_t discard_outliers, eao_track_result* out, void* stream);

/* Tracking::TrackReferenceKeyFrame's data path (reference src/Tracking.cc:1568-1631) on the same chain: the frame set-up as above,
 * ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vpMapPointMatches) (src/ORBmatcher.cc:159-288: ORBmatcher(0.7, true) at :1574, rotation-histogram factor
 * 1 / HISTO_LENGTH :171), Optimizer::PoseOptimization from the last frame's pose (:1584-1590), and -- when discard_outliers != 0 -- the "Discard outliers" loop
 * (:1593-1612); one copy back.  The reference keyframe travels as host arrays per keyframe keypoint i (n_kf <= max_keypoints entries): valid[i] != 0 where
 * GetMapPointMatches()[i] != NULL && !isBad(); Xw = GetWorldPos(); kf_desc = pKF->mDescriptors.row(i); kf_angle = pKF->mvKeysUn[i].angle.  fv_kf = pKF->mFeatVec,
 * fv_cur = mCurrentFrame.mFeatVec: the frame's own vector is what Frame::ComputeBoW (:1571) makes of its descriptors on the host -- DBoW2's vocabulary tree is
 * not part of this library, so this stage costs the descriptors one trip to the host before the call.  A keypoint index lies in ONE node (DBoW2 files a feature
 * under its single ancestor at levelsup); a vector that lists an index twice, out of range, or whose node ids do not ascend fails with EAO_ERR_INVALID before
 * any kernel runs, an index of fv_cur beyond the *d_n keypoints the extractor left fails after the chain.  Tcw_last: mLastFrame.mTcw, 16 floats row-major, finite.
 * Result as for eao_tracker_track_with_motion_model, with kp_map_point[k] = the KEYFRAME KEYPOINT INDEX whose map point frame keypoint k took, or -1, and n_matches =
 * SearchByBoW's return value (what upstream tests against 15 / 10).  Same results as eao_compute_stereo_from_rgbd -> eao_search_by_bow(mode 0) ->
 * eao_pose_optimization on the same data. */
eao_status eao_tracker_track_reference_keyframe(eao_tracker* h, const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                                const float* d_depth, int32_t depth_pitch, int32_t width, int32_t height, const float* Tcw_last,
                                                int32_t n_kf, const uint8_t* valid, const float* Xw, const uint8_t* kf_desc, const float* kf_angle,
                                                const eao_feature_vector* fv_kf, const eao_feature_vector* fv_cur, float nnratio, int32_t check_orientation,
                                                int32_t discard_outliers, eao_track_result* out, void* stream);

/* The value of EAO_ABI_VERSION the library was built with.  Bumped whenever an entry point's parameter list or a struct's layout changes (round 3
 * changed eao_tracker_track_local_map and eao_track_result in place); a caller compiled against another version must not call into the library.
 * Result structs are zero-initialised by the caller (`eao_track_result R = {0};`) before their array pointers are set: a pointer member the
 * caller's header does not know yet then reads as NULL = "not wanted". */
#define EAO_ABI_VERSION 6
int32_t eao_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* EAO_FUSION_H_ */
