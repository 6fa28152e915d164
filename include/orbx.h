/* orbx -- C ABI of the MI355X (gfx950) ORB feature extractor.
 *
 * This is the boundary a host program binds to replace ORB_SLAM2::ORBextractor
 * (reference include/ORBextractor.h:45-111, src/ORBextractor.cc).  Plain pointers and sizes
 * only; no C++ or torch types.  All functions return 0 (ORBX_OK) or a negative error code
 * and never throw.  Outputs are allocated by the caller.
 *
 * A context owns one HIP stream and all device scratch; contexts are independent and may
 * be used from different host threads at the same time (the reference runs two extractor
 * instances on two threads for stereo, src/Frame.cc:78-81).  One context is not re-entrant.
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_MAX_LEVELS 16

enum {
    ORBX_OK = 0,
    ORBX_E_INVALID = -1,   /* bad argument */
    ORBX_E_HIP = -2,       /* a HIP runtime call failed (see orbx_last_hip_error) */
    ORBX_E_NO_DEVICE = -3, /* no usable GPU */
    ORBX_E_CAPACITY = -4,  /* image/batch larger than the context was created for, or kp_cap too small */
    ORBX_E_GEOMETRY = -5,  /* pyramid level too small for the 30-px cell grid (under 62 px), wider or higher than
                              4127 px, or with a feature quota above 8189 */
    ORBX_E_STATE = -6,     /* no extraction has run yet (pyramid read-back), or the intermediate asked for is gone */
    ORBX_E_DEVICE_DATA = -7 /* a kernel read a count or a coordinate out of device memory that lay outside the array it
                              indexes (memory the library does not own was overwritten, or a producer failed).  The
                              kernel went on with the value cut to its bound -- nothing was accessed out of range, the
                              GPU does not fault -- and the context reports this code from the next call that waits
                              for the work (orbx_synchronize, a call with sync != 0, a host-memory entry point), once;
                              the results of the batches enqueued since the last such call are invalid.  The reference
                              has no counterpart (its lists are std::vectors on the host). */
};

/* Same field order and size (28 bytes) as cv::KeyPoint. */
typedef struct orbx_kp {
    float x, y;      /* level-0 pixel coordinates (level coords * scale factor of the octave) */
    float size;      /* float(int(31 * scale)) */
    float angle;     /* degrees, [0, 360) */
    float response;  /* FAST score */
    int32_t octave;
    int32_t class_id; /* -1 */
} orbx_kp;

/* Replaces the constructor arguments of ORBextractor::ORBextractor
 * (reference include/ORBextractor.h:51-52, src/ORBextractor.cc:416-423) plus sizing. */
typedef struct orbx_params {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t device_id;   /* HIP device ordinal */
    int32_t max_width;   /* largest image the context must accept */
    int32_t max_height;
    int32_t max_batch;   /* largest number of frames per call */
    uint32_t flags;      /* reserved, 0 */
} orbx_params;

typedef struct orbx_level_geom {
    int32_t width, height;     /* level image size (reference src/ORBextractor.cc:1220-1221) */
    int32_t n_cols, n_rows;    /* FAST cell grid (:871-877) */
    int32_t w_cell, h_cell;
    int32_t n_cells;           /* cells actually visited (:885-895 skip rules applied) */
    int32_t n_ini;             /* quadtree roots (:576) */
    int32_t quota;             /* mnFeaturesPerLevel[level] (:447-463) */
} orbx_level_geom;

typedef struct orbx_ctx orbx_ctx;

/* ---- host-only helpers (no GPU needed) ------------------------------------------------ */

/* Constructor tables: mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2,
 * mnFeaturesPerLevel (each nlevels long) and umax (16).  Replaces what the getters
 * GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares return (reference include/ORBextractor.h:69-83).
 * Any output pointer may be NULL. */
int orbx_host_tables(int nfeatures, float scale_factor, int nlevels,
                     float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                     int32_t* quota, int32_t* umax16);

/* Per-level geometry for a w x h input (reference src/ORBextractor.cc:862-897, 576, 1220). */
int orbx_host_level_geometry(int nfeatures, float scale_factor, int nlevels,
                             int width, int height, int level, orbx_level_geom* out);

/* Fixed-point bilinear taps for one axis (cv::resize INTER_LINEAR 8UC1, call site
 * src/ORBextractor.cc:1237): for each of dn destination indices the source index and the two
 * 11-bit coefficients.  clamp_index != 0 applies the horizontal rule (index clamped, weight
 * zeroed); 0 applies the vertical rule (index left unclamped, may be -1 or sn-1). */
int orbx_host_resize_taps(int sn, int dn, int clamp_index, int32_t* ofs, int16_t* c0, int16_t* c1);

/* The byte selector (v_perm_b32 convention: values 4..7 pick from the dword that holds columns hi_col..hi_col+3,
 * values 0..3 from the one that holds lo_col..lo_col+3) with which the blur kernel builds the dword of columns
 * target_col..target_col+3 of a row of `width` pixels at the image's left and right rim, where cv::GaussianBlur's
 * BORDER_REFLECT_101 (call site src/ORBextractor.cc:1188) mirrors the columns.  Host-side copy of the device
 * function, exported so that the rim arithmetic can be tested without a GPU. */
uint32_t orbx_host_blur_rim_selector(int target_col, int hi_col, int lo_col, int width);

/* Upper bound on keypoints returned per image for these parameters and image size. */
int orbx_host_max_keypoints(int nfeatures, float scale_factor, int nlevels, int width, int height);

/* The frames [*first, *last) that worker `rank` of `world` takes of a stream of n_frames frames: contiguous runs
 * of stereo pairs (frames 2k, 2k + 1 stay together, so that the association of src/Frame.cc:90 can follow on the
 * same device), sizes differing by at most one pair.  The multi-GPU paths (orbx_pool_extract_stream, the
 * process-per-GPU benchmark) cut their streams with it; the reference has one extractor per image and no
 * counterpart (src/Frame.cc:78-81). */
int orbx_host_frame_shard(int n_frames, int rank, int world, int* first, int* last);

/* ---- context -------------------------------------------------------------------------- */

/* Replaces `new ORBextractor(...)` (reference src/Tracking.cc:121,125,128). */
int orbx_create(const orbx_params* params, orbx_ctx** out);
/* Releases stream and device memory (the reference destructor is trivial,
 * include/ORBextractor.h:54). */
void orbx_destroy(orbx_ctx* ctx);

/* ---- extraction ----------------------------------------------------------------------- */

/* One 8-bit gray image in host memory.  Replaces ORBextractor::operator()
 * (reference include/ORBextractor.h:59-61, src/ORBextractor.cc:1137-1211); the mask argument
 * of the reference is ignored there and has no counterpart here.
 * kps[kp_cap], desc[kp_cap*32]; *n_out receives the keypoint count.
 * kp_cap must be >= orbx_host_max_keypoints(...) for this image size.
 * The call returns when the results are in kps / desc.  Upload, all kernel launches and the
 * download are captured once per image size into a HIP graph and replayed as one submission
 * per call (the reference calls the extractor once per image, src/Frame.cc:252, so the cost of
 * a call is launch overhead as much as kernel time). */
int orbx_extract(orbx_ctx* ctx, const uint8_t* img, int width, int height, int stride,
                 orbx_kp* kps, int kp_cap, uint8_t* desc, int* n_out);

/* A stereo pair in host memory as ONE call on ONE context.  Replaces what the stereo Frame constructor does with
 * its two extractor instances on two threads plus the association that follows (reference src/Frame.cc:78-81:
 * thread threadLeft(&Frame::ExtractORB, this, 0, imLeft), threadRight(..., 1, imRight), join; :90
 * ComputeStereoMatches): both images go through the extractor in the same launches (frame 0 = left, 1 = right),
 * the stereo association of orbx_stereo_match runs behind them on the device, and upload, all launches and the
 * downloads are one captured HIP graph per image size, replayed per call.  kps[2 * kp_cap], desc[2 * kp_cap * 32]
 * and n_out[2] are laid out as orbx_extract_batch lays out two frames; u_right[kp_cap] and depth[kp_cap] receive
 * mvuRight and mvDepth of the n_out[0] left keypoints (-1 = no match), or pass both NULL to skip the
 * association (mb, mbf are then ignored; else mb = baseline in metres, mbf = baseline * fx, src/Frame.cc:114).
 * The context must have been created with max_batch >= 2.  The pyramids of both images stay readable
 * (orbx_pyramid_level, image 0 and 1). */
int orbx_extract_pair(orbx_ctx* ctx, const uint8_t* left, int stride_left, const uint8_t* right, int stride_right,
                      int width, int height, orbx_kp* kps, int kp_cap, uint8_t* desc, int n_out[2], float mb, float mbf,
                      float* u_right, float* depth);

/* n images of one size in host memory; image i starts at imgs[i] with row stride strides[i].
 * kps[n*kp_cap], desc[n*kp_cap*32], n_out[n].  The throughput entry point: the reference
 * processes one image per call (src/Frame.cc:249-255); batching has no counterpart there. */
int orbx_extract_batch(orbx_ctx* ctx, int n, const uint8_t* const* imgs, int width, int height,
                       const int* strides, orbx_kp* kps, int kp_cap, uint8_t* desc, int* n_out);

/* ---- several GPUs in one process: a pool of workers -----------------------------------------
 * One host worker thread and one context per entry of devices[] (a device may be listed more than once: its
 * workers then share it).  params as for orbx_create (device_id is ignored; max_batch >= 2 is the number of frames a
 * worker hands its GPU per orbx_extract_batch call).  orbx_pool_extract_stream is orbx_extract_batch for a stream
 * of n host frames over all workers: worker r takes the frames orbx_host_frame_shard(n, r, workers) names, runs
 * them through its context's chunked, pipelined host path (pinned upload and result slots, copies overlapped with
 * the kernels) and writes their results into the caller's arrays at the frames' own positions, so that kps, desc
 * and n_out are in frame order when the call returns.  No data crosses devices.  Returns the first error of any
 * worker (the arrays are then not to be used).  Has no counterpart in the reference, which extracts one image per
 * call on the calling thread (src/Frame.cc:249-255); it is the in-process form of sharding a stream by frame. */
typedef struct orbx_pool orbx_pool;
int orbx_pool_create(const orbx_params* params, const int* devices, int n_workers, orbx_pool** out);
void orbx_pool_destroy(orbx_pool* pool);
int orbx_pool_workers(const orbx_pool* pool);
int orbx_pool_extract_stream(orbx_pool* pool, int n, const uint8_t* const* imgs, int width, int height,
                             const int* strides, orbx_kp* kps, int kp_cap, uint8_t* desc, int* n_out);

/* As orbx_extract_batch with every buffer already in device memory (HBM-resident input,
 * HBM-resident results).  d_imgs holds n images, image i at d_imgs + i*frame_stride, rows
 * `stride` bytes apart.  d_kps[n*kp_cap], d_desc[n*kp_cap*32], d_n_out[n] are device
 * pointers.  The work is enqueued on streams the context owns (three of them, ordered among
 * themselves by events; none is exposed).  If `sync` is non-zero the call returns after all of
 * it has finished.  With sync = 0 it returns at once, and until orbx_synchronize -- or a later
 * orbx_* call on the same context, each of which orders itself behind this batch -- has
 * returned:
 *   - d_imgs must stay valid and unmodified: an input whose pointer, stride and frame stride
 *     are multiples of 4 is read in place (level 0 of the pyramid is the caller's buffer) by
 *     the detection, the first pyramid level and the orientation stage, which keep reading
 *     after the call has returned;
 *   - d_kps, d_desc and d_n_out must not be read, written or freed: they are written last, by
 *     the descriptor stage, on a stream of its own.
 * Work the caller enqueues on streams of its own is not ordered against the batch: synchronise
 * it first.  In particular d_imgs must be complete in device memory when the call is made,
 * unless an orbx_* call on this context produced it (orbx_gray_batch_device,
 * orbx_remap_batch_device: those are ordered in front of the batch).  The library relies on
 * that: a batch of resident images builds its pyramid as soon as the arenas are free, which
 * may be before the previous batch has been detected.  Several batches may be enqueued back to
 * back with sync = 0 (each with its own output arrays); their results are those of the same
 * calls made one at a time, and their stages overlap where they allow. */
int orbx_extract_batch_device(orbx_ctx* ctx, int n, const uint8_t* d_imgs, int width, int height,
                              int stride, size_t frame_stride, orbx_kp* d_kps, int kp_cap,
                              uint8_t* d_desc, int32_t* d_n_out, int sync);

int orbx_synchronize(orbx_ctx* ctx);

/* ---- the stage before the extractor: input staging in device memory ---------------------
 * Both calls work on n equally sized frames (frame i at base + i*frame_stride, rows `stride`
 * bytes apart), enqueue on the context's stream and so order themselves before a following
 * orbx_extract_batch_device on the same context. */

/* Replaces cv::cvtColor(im, im, CV_{RGB,BGR,RGBA,BGRA}2GRAY) of Tracking::GrabImageStereo /
 * GrabImageRGBD / GrabImageMonocular (reference src/Tracking.cc:175-199, 213-231, 250-268):
 * OpenCV 3.4.5's 8-bit form (B*1868 + G*9617 + R*4899 + 8192) >> 14.  channels = 3 or 4 (alpha
 * ignored); blue_idx = 0 for BGR(A) byte order, 2 for RGB(A) (the reference's mbRGB). */
int orbx_gray_batch_device(orbx_ctx* ctx, int n, const uint8_t* d_src, int width, int height,
                           int src_stride, size_t src_frame_stride, int channels, int blue_idx,
                           uint8_t* d_dst, int dst_stride, size_t dst_frame_stride, int sync);

/* Replaces cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) of the stereo examples (reference
 * Examples/Stereo/stereo_euroc.cc:137-138; maps from cv::initUndistortRectifyMap, :98-99, stay
 * a host-side set-up step): 8UC1 image, CV_32FC1 maps of dst_width x dst_height floats (rows
 * map_stride floats apart, shared by all frames), constant border 0, OpenCV 3.4.5's fixed-point
 * form (coordinates rounded to 1/32 px, 15-bit weights). */
int orbx_remap_batch_device(orbx_ctx* ctx, int n, const uint8_t* d_src, int src_width, int src_height,
                            int src_stride, size_t src_frame_stride, const float* d_mapx,
                            const float* d_mapy, int map_stride, uint8_t* d_dst, int dst_width,
                            int dst_height, int dst_stride, size_t dst_frame_stride, int sync);

/* ---- right after the extractor: undistorted keypoints and the frame grid ------------------
 * Replaces Frame::UndistortKeyPoints (reference src/Frame.cc:421-451: cv::undistortPoints with
 * R = I, P = K, OpenCV 3.4.5's five double-precision iterations; dist[0] == 0 copies) and
 * Frame::AssignFeaturesToGrid / PosInGrid (reference src/Frame.cc:232-247, 398-408) for every
 * frame of a batch in device memory.  K = {fx, fy, cx, cy}, dist = {k1, k2, p1, p2, k3} (host
 * pointers); min_x, min_y, inv_w, inv_h are Frame's mnMinX, mnMinY, mfGridElementWidthInv,
 * mfGridElementHeightInv (src/Frame.cc:101-102, 452-482).  d_kps_un[n*kp_cap] receives the
 * keypoints with undistorted coordinates (mvKeysUn); the 64 x 48 grid comes back in CSR form:
 * cell gx*48 + gy of frame f holds d_cell_idx[f*kp_cap + s .. e) with s, e =
 * d_cell_start[f*3073 + cell], [.. + cell + 1], indices ascending (mGrid's push_back order). */
#define ORBX_GRID_COLS 64
#define ORBX_GRID_ROWS 48
int orbx_undistort_grid_batch_device(orbx_ctx* ctx, int n, const orbx_kp* d_kps, int kp_cap,
                                     const int32_t* d_n, const float K[4], const float dist[5],
                                     float min_x, float min_y, float inv_w, float inv_h,
                                     orbx_kp* d_kps_un, int32_t* d_cell_start, int32_t* d_cell_idx,
                                     int sync);

/* ---- RGB-D frames: depth per keypoint ------------------------------------------------------
 * Replaces Frame::ComputeStereoFromRGBD (reference src/Frame.cc:662-682) together with the depth map's
 * conversion in Tracking::GrabImageRGBD (src/Tracking.cc:230-231, imDepth.convertTo(imDepth, CV_32F,
 * mDepthMapFactor)) for every frame of a batch: feature i samples the depth map at the truncated
 * coordinates of mvKeys[i], d = sample * depth_map_factor (float32; pass Tracking's mDepthMapFactor, i.e.
 * 1.0f / the settings' DepthMapFactor, or 1), and d > 0 gives d_depth_out = d, d_u_right =
 * mvKeysUn[i].pt.x - mbf / d; other features and the slots from d_n[f] to kp_cap get -1.  The depth
 * frames are 16-bit unsigned (depth_is_f32 = 0, the TUM PNGs) or float32 samples in device memory, rows
 * stride_bytes apart.  Only the sampled values are converted: the converted map has no other reader. */
int orbx_rgbd_depth_batch_device(orbx_ctx* ctx, int n, const void* d_depth, int depth_is_f32, int width, int height,
                                 int stride_bytes, size_t frame_stride_bytes, float depth_map_factor, const orbx_kp* d_kps,
                                 const orbx_kp* d_kps_un, int kp_cap, const int32_t* d_n, float mbf, float* d_u_right,
                                 float* d_depth_out, int sync);

/* ---- descriptor consumer: the bag-of-words transform -------------------------------------
 * Replaces Frame::ComputeBoW (reference src/Frame.cc:411-418), i.e.
 * ORBVocabulary::transform(descriptors, mBowVec, mFeatVec, 4) of the vendored DBoW2 (reference
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1194; one feature down the tree :1218-1260;
 * FORB::distance, FORB.cpp:81-101; BowVector.cpp:25-84; FeatureVector.cpp:31-45), for every frame
 * of a batch whose descriptors are in device memory.
 *
 * A vocabulary is a tree: node 0 is the root, node i > 0 has parent[i] < i, a 32-byte centre, a
 * weight and the is-leaf flag of its line in the text file; word ids go to the flagged nodes in
 * node order, children keep node order (loadFromTextFile, :1338-1425).  scoring / weighting are
 * the header's enum values (ORBvoc.txt: "10 6 0 0" = k, L, L1_NORM, TF_IDF). */
typedef struct orbx_voc orbx_voc;
int orbx_voc_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                    const uint8_t* is_leaf, const uint8_t* desc /* n_nodes x 32 */,
                    const double* weight, orbx_voc** out);
/* The text format ORB_SLAM2 ships its vocabulary in (System.cc loads it with loadFromTextFile). */
int orbx_voc_load_text(const char* path, orbx_voc** out);
void orbx_voc_destroy(orbx_voc* voc);
int orbx_voc_info(const orbx_voc* voc, int* k, int* L, int* n_nodes, int* n_words);
/* One node of the tree as the library holds it (host side): number of children, the first child's
 * node id (0 if none), centre, weight, word id.  For checking a loaded file. */
int orbx_voc_node(const orbx_voc* voc, int node, int* n_children, uint32_t* first_child, uint8_t desc[32],
                  double* weight, uint32_t* word);
/* d_desc[n][kp_cap][32], d_n[n]: as the extractor's batch entry points leave them.  Per feature
 * (arrays of n*kp_cap): d_word, d_node (the node met at level L - levelsup; 0 when that level is
 * not positive or the leaf is met earlier), d_weight.  Per frame f: the BowVector in word order in
 * d_bow_id / d_bow_val[f*kp_cap .. + d_bow_n[f]) and the FeatureVector as (node, feature index)
 * pairs, nodes ascending and indices ascending inside a node, in d_fv_node / d_fv_idx[f*kp_cap
 * .. + d_fv_n[f]).  Features whose word weight is 0 (stopped words) are left out of both, as in
 * the reference.  kp_cap <= 16384.  The vocabulary is copied to the context's device on first
 * use and stays there. */
int orbx_bow_transform_batch_device(orbx_ctx* ctx, orbx_voc* voc, int n, const uint8_t* d_desc, int kp_cap,
                                    const int32_t* d_n, int levelsup, uint32_t* d_word, uint32_t* d_node,
                                    double* d_weight, uint32_t* d_bow_id, double* d_bow_val, int32_t* d_bow_n,
                                    uint32_t* d_fv_node, uint32_t* d_fv_idx, int32_t* d_fv_n, int sync);

/* ---- descriptor consumer: best match inside a window ------------------------------------
 * Replaces Frame::GetFeaturesInArea (reference src/Frame.cc:338-396) together with the candidate
 * loop of ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (reference
 * src/ORBmatcher.cc:1427-1455) for many projected map points per frame at once, on the outputs
 * of orbx_undistort_grid_batch_device (d_kps_un, d_cell_start, d_cell_idx) and the extractor's
 * descriptors.  A query is a projected point: window centre (u, v), half-size radius, level
 * range as GetFeaturesInArea takes it (max_level < 0: no upper limit), the predicted right
 * coordinate ur, and the map point's descriptor.  d_u_right (mvuRight, may be null): features with
 * a value > 0 must lie within radius of ur.  d_taken (may be null): non-zero entries are skipped
 * -- the reference's "mvpMapPoints[i2] && Observations() > 0" as the caller holds it.  The
 * reference updates that state inside its loop over the map points; here it is an input, so the
 * caller orders dependent rounds.  Per query: d_best_idx = feature index of the smallest
 * distance below 256, first in grid order, or -1; d_best_dist = that distance (256 if none). */
typedef struct orbx_window_query {
    float u, v, radius;
    int32_t min_level, max_level;
    float ur;
    uint8_t desc[32];
} orbx_window_query; /* 56 bytes */
int orbx_search_window_batch_device(orbx_ctx* ctx, int n, const orbx_kp* d_kps_un, int kp_cap,
                                    const int32_t* d_cell_start, const int32_t* d_cell_idx, const uint8_t* d_desc,
                                    const float* d_u_right, const uint8_t* d_taken, float min_x, float min_y,
                                    float inv_w, float inv_h, const orbx_window_query* d_queries, int q_cap,
                                    const int32_t* d_nq, int32_t* d_best_idx, int32_t* d_best_dist, int sync);


/* Replaces ORBVocabulary::score(v1, v2) for ORB-SLAM2's vocabulary (L1 norm: DBoW2 L1Scoring::score, reference
 * Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68; called once per candidate keyframe from
 * KeyFrameDatabase::DetectLoopCandidates, src/KeyFrameDatabase.cc:133, DetectRelocalizationCandidates, :249, and
 * LoopClosing::DetectLoop, src/LoopClosing.cc:137) for a batch of pairs of BowVectors as
 * orbx_bow_transform_batch_device leaves them (ascending word ids with their values): pair p scores vector
 * d_pairs[2p] of the first set (d_bow_*_1, rows of cap_1 entries) against vector d_pairs[2p + 1] of the second set
 * (which may be the same arrays).  d_score[p] is the reference's double, bit for bit (its callers narrow it to
 * float). */
int orbx_bow_score_l1_batch_device(orbx_ctx* ctx, int n_pairs, const uint32_t* d_bow_id_1, const double* d_bow_val_1,
                                   const int32_t* d_bow_n_1, int cap_1, const uint32_t* d_bow_id_2, const double* d_bow_val_2,
                                   const int32_t* d_bow_n_2, int cap_2, const int32_t* d_pairs, double* d_score, int sync);

/* Replaces the search inside ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th) (reference
 * src/ORBmatcher.cc:929-986; KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:575-614; called from
 * LocalMapping::SearchInNeighbors, src/LocalMapping.cc:573 and :601) for a batch of keyframes: query i of keyframe f
 * is a map point that passed the tests of :873-922, with u, v (:892-893), ur (:899), radius (:927) and
 * min_level = nPredictedLevel - 1, max_level = nPredictedLevel (:948).  Among the keyframe's keypoints in the
 * window and on those levels whose reprojection error passes the chi-square test of :951-975 (5.99, or 7.8 with
 * d_u_right[idx] >= 0; d_u_right may be null; mvInvLevelSigma2 is the context's), d_best_idx receives the one of
 * smallest descriptor distance, first in grid order (-1: none), d_best_dist that distance (256: none).  The
 * caller applies bestDist <= TH_LOW and the map updates of :989-1008 in query order.  The second overload,
 * Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:1014-1137), and both directions of SearchBySim3 (:1139-1363)
 * search without the chi-square test: they are orbx_search_window_batch_device with d_u_right = d_taken = null. */
int orbx_fuse_search_batch_device(orbx_ctx* ctx, int n, const orbx_kp* d_kps_un, int kp_cap, const int32_t* d_cell_start,
                                  const int32_t* d_cell_idx, const uint8_t* d_desc, const float* d_u_right, float min_x,
                                  float min_y, float inv_w, float inv_h, const orbx_window_query* d_queries, int q_cap,
                                  const int32_t* d_nq, int32_t* d_best_idx, int32_t* d_best_dist, int sync);

/* ---- descriptor consumer: ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) ----
 * Replaces the whole function (reference src/ORBmatcher.cc:1365-1507, ComputeThreeMaxima
 * :1638-1679) for a batch of (current frame, last frame) pairs: the last frame's map points are
 * projected with the current pose (:1396-1413), searched in their window on the outputs of
 * orbx_undistort_grid_batch_device and the extractor's descriptors (:1415-1461), matches at
 * distance <= 100 are written to the current frame so that later points skip features held by a
 * point with observations (:1440-1442, :1463-1466), and the rotation histogram clears the matches
 * outside its three largest bins (:1468-1504).  The result is that of the reference's sequential
 * loop; the points are searched concurrently in rounds (see k_search_projection).
 * d_queries[f*q_cap + i], i < d_nq[f]: feature i of the last frame.  d_poses[f*24 ..]: rows of the
 * 3x4 [R|t] of the current frame (Tcw), then of the last frame (Tlw).  d_taken (may be null):
 * features that hold a map point with observations before the call.  d_u_right may be null.
 * Scale factors are the context's (mvScaleFactors).  Outputs: d_match[f*kp_cap + k] = index i of
 * the map point feature k ends with, -1 = none or cleared by the rotation check;
 * d_query_match[f*q_cap + i] = the feature point i was written to (before the rotation check),
 * -1 = none; d_nmatches[f] = the reference's return value. */
typedef struct orbx_map_query {
    float x, y, z;     /* MapPoint::GetWorldPos */
    int32_t octave;    /* LastFrame.mvKeys[i].octave */
    float angle;       /* LastFrame.mvKeysUn[i].angle */
    int32_t flags;     /* bit 0: Observations() > 0; bit 1: skip (no map point, or mvbOutlier[i]) */
    uint8_t desc[32];  /* MapPoint::GetDescriptor */
} orbx_map_query; /* 56 bytes */
typedef struct orbx_projection_params {
    float fx, fy, cx, cy, mbf, mb;                   /* Frame::fx .. mb */
    float min_x, min_y, max_x, max_y, inv_w, inv_h;  /* mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElement{Width,Height}Inv */
    float th;                                        /* window half-size at level 0 */
    int32_t mono;                                    /* bMono */
    int32_t check_orientation;                       /* ORBmatcher::mbCheckOrientation */
} orbx_projection_params;
int orbx_search_by_projection_batch_device(orbx_ctx* ctx, int n, const orbx_kp* d_kps_un, int kp_cap,
                                           const int32_t* d_cell_start, const int32_t* d_cell_idx,
                                           const uint8_t* d_desc, const float* d_u_right, const uint8_t* d_taken,
                                           const orbx_projection_params* params, const float* d_poses,
                                           const orbx_map_query* d_queries, int q_cap, const int32_t* d_nq,
                                           int32_t* d_match, int32_t* d_query_match, int32_t* d_nmatches, int sync);

/* ---- descriptor consumer: ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) ----
 * Replaces the local-map search of Tracking::SearchLocalPoints (reference src/ORBmatcher.cc:45-129,
 * RadiusByViewingCos :131-137) for a batch of frames: every map point that Frame::isInFrustum found in
 * view comes with its projection; its window (radius 2.5 or 4.0 by the viewing cosine, times th when
 * th != 1, times the scale factor of the predicted level; levels predicted - 1 .. predicted) is searched
 * for the best and the second best descriptor distance, a best distance <= 100 is accepted unless both
 * lie on one level and best > nn_ratio * second (mfNNratio), and the match is written to the frame at
 * once, so that later points skip features held by a point with observations (:87-89).  The result is
 * that of the reference's sequential loop; the points are searched concurrently in rounds.
 * d_queries[f*q_cap + i], i < d_nq[f]: map point i.  d_taken (may be null): features that hold a map
 * point with observations before the call.  d_u_right (mvuRight) may be null.  Scale factors are the
 * context's.  Outputs: d_match[f*kp_cap + k] = index i of the map point feature k ends with (-1: the
 * call wrote none); d_query_match[f*q_cap + i] = the feature point i was written to (-1: none);
 * d_nmatches[f] = the reference's return value. */
typedef struct orbx_local_query {
    float u, v, ur;    /* MapPoint::mTrackProjX, mTrackProjY, mTrackProjXR */
    int32_t level;     /* mnTrackScaleLevel */
    float view_cos;    /* mTrackViewCos */
    int32_t flags;     /* bit 0: Observations() > 0; bit 1: skip (!mbTrackInView or isBad()) */
    uint8_t desc[32];  /* MapPoint::GetDescriptor */
} orbx_local_query; /* 56 bytes */
int orbx_search_local_map_batch_device(orbx_ctx* ctx, int n, const orbx_kp* d_kps_un, int kp_cap,
                                       const int32_t* d_cell_start, const int32_t* d_cell_idx, const uint8_t* d_desc,
                                       const float* d_u_right, const uint8_t* d_taken, float min_x, float min_y,
                                       float inv_w, float inv_h, float th, float nn_ratio,
                                       const orbx_local_query* d_queries, int q_cap, const int32_t* d_nq,
                                       int32_t* d_match, int32_t* d_query_match, int32_t* d_nmatches, int sync);

/* ---- in front of the searches: projecting map points ------------------------------------------
 * orbx_is_in_frustum_batch_device replaces Frame::isInFrustum (reference src/Frame.cc:271-327) with
 * MapPoint::PredictScale (src/MapPoint.cc:402-417) as Tracking::SearchLocalPoints calls it for every local map point
 * (src/Tracking.cc:1340-1353), for a batch of frames: point i of frame f is moved into the camera frame of
 * d_poses[f*12 ..] (rows of [Rcw | tcw]), must have a depth that is not negative, project inside
 * [min_x, max_x] x [min_y, max_y], lie between 0.8 * min_dist and 1.2 * max_dist from the camera centre
 * d_centres[f*3 ..] (Frame::mOw) and be seen at a cosine of at least view_cos_limit against its normal.  The record
 * written for it is what orbx_search_local_map_batch_device reads: mTrackProjX / Y / XR, mnTrackScaleLevel,
 * mTrackViewCos, the descriptor, flags bit 0 copied and bit 1 set when the point is skipped on input or not in
 * view (mbTrackInView false).  d_n_in_view[f] = the points in view (nToMatch of src/Tracking.cc:1351).
 *
 * orbx_fuse_project_batch_device replaces the tests in front of the search of ORBmatcher::Fuse(pKF, vpMapPoints, th)
 * (src/ORBmatcher.cc:881-927; KeyFrame::IsInImage, src/KeyFrame.cc:616-619): the same steps with that function's
 * arithmetic (x / z first, half-open image bounds, PO . Pn >= 0.5 * dist in double), d_centres = pKF->GetCameraCenter().
 * Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) applies the same tests (:1046-1086) to the pose taken out of Scw (:1023-1027).
 * The records are the queries of orbx_fuse_search_batch_device: u, v, ur, radius = th * mvScaleFactors[level],
 * levels level - 1 .. level; a point that is skipped on input (flags bit 1: NULL, isBad() or IsInKeyFrame(pKF)) or
 * fails a test gets radius -1, an empty window, so its result is "no match".  d_n_projected[f] counts the others.
 *
 * The logarithm of PredictScale is the float nearest to log(max_dist / dist), which a C library's logf returns
 * except for arguments within a fraction of an ulp of a rounding boundary.  log_scale_factor is
 * Frame::mfLogScaleFactor (src/Frame.cc:71); the scale factors and the number of levels are the context's. */
typedef struct orbx_frustum_point {
    float x, y, z;            /* MapPoint::GetWorldPos */
    float nx, ny, nz;         /* MapPoint::GetNormal */
    float min_dist, max_dist; /* MapPoint::mfMinDistance, mfMaxDistance */
    int32_t flags;            /* bit 0: Observations() > 0 (copied to the output); bit 1: skip */
    uint8_t desc[32];         /* MapPoint::GetDescriptor */
} orbx_frustum_point; /* 68 bytes */
typedef struct orbx_frustum_params {
    float fx, fy, cx, cy, mbf;           /* Frame::fx .. mbf */
    float min_x, max_x, min_y, max_y;    /* mnMinX, mnMaxX, mnMinY, mnMaxY */
    float view_cos_limit;                /* 0.5 in Tracking::SearchLocalPoints; not used by the Fuse variant */
    float log_scale_factor;              /* mfLogScaleFactor */
    float th;                            /* Fuse's th; not used by isInFrustum */
} orbx_frustum_params;
int orbx_is_in_frustum_batch_device(orbx_ctx* ctx, int n, const orbx_frustum_params* params, const float* d_poses,
                                    const float* d_centres, const orbx_frustum_point* d_points, int q_cap,
                                    const int32_t* d_nq, orbx_local_query* d_queries_out, int32_t* d_n_in_view, int sync);
int orbx_fuse_project_batch_device(orbx_ctx* ctx, int n, const orbx_frustum_params* params, const float* d_poses,
                                   const float* d_centres, const orbx_frustum_point* d_points, int q_cap,
                                   const int32_t* d_nq, orbx_window_query* d_queries_out, int32_t* d_n_projected, int sync);

/* Replaces the projection of either direction of ORBmatcher::SearchBySim3 (reference src/ORBmatcher.cc:1195-1226 and
 * :1275-1306) for a batch of keyframe pairs: the map points of one keyframe go through that keyframe's pose and the
 * similarity between the cameras, d_poses[f*24 ..] = rows of [R1w | t1w], then rows of [sR21 | t21] (:1156-1158; the
 * other direction passes [R2w | t2w] and [sR12 | t12]); depth not negative, KeyFrame::IsInImage of the target keyframe,
 * the distance region and PredictScale on the length of the camera-frame point (:1216), no viewing test.  The records
 * are window queries as orbx_fuse_project_batch_device writes them (radius -1 for a point skipped on input -- NULL,
 * isBad() or already matched -- or dropped); the search is orbx_search_window_batch_device with d_u_right = d_taken =
 * null and the caller's bestDist <= TH_HIGH.  The normals of the points, mbf and view_cos_limit are not read. */
int orbx_sim3_project_batch_device(orbx_ctx* ctx, int n, const orbx_frustum_params* params, const float* d_poses,
                                   const orbx_frustum_point* d_points, int q_cap, const int32_t* d_nq,
                                   orbx_window_query* d_queries_out, int32_t* d_n_projected, int sync);

/* Replaces the matching loops of ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th,
 * ORBdist) (reference src/ORBmatcher.cc:1509-1636; Tracking::Relocalization, src/Tracking.cc:1654 and :1670) and of
 * ORBmatcher::SearchByProjection(KeyFrame* pKF, Scw, vpPoints, vpMatched, th) (:290-403; LoopClosing::ComputeSim3,
 * src/LoopClosing.cc:378) for a batch of frames / keyframes.  The map points come projected (:1534-1563, :321-360: pose,
 * image bounds, distance invariance, viewing angle, PredictScale -- orbx_project_guided_batch_device below does it on
 * the device), one record per point, in the reference's order, with flags bit 1 set for a point the reference skips and bit 0 set otherwise.  Each point searches its
 * window on levels predicted - 1 .. predicted + level_up (1 for the first function, 0 for the second) among the
 * features that hold no map point -- d_taken before the call (mvpMapPoints[i] / vpMatched[i] != NULL), and those an
 * earlier point of this call took -- and takes the most similar one when its distance is <= max_dist (ORBdist;
 * TH_LOW = 50 for the second function).  check_orientation applies the rotation histogram of :1597-1633 with the
 * angle in the record against the feature's.  The result is that of the reference's sequential loop.  Outputs as
 * orbx_search_local_map_batch_device: d_match[f*kp_cap + k] = the point feature k receives (-1: none),
 * d_query_match[f*q_cap + i] = the feature point i was written to (-1: none), d_nmatches[f] = the return value. */
typedef struct orbx_guided_query {
    float u, v;        /* the projection */
    float angle;       /* pKF->mvKeysUn[i].angle of the map point's keypoint (rotation check only) */
    int32_t level;     /* nPredictedLevel */
    float radius;      /* th * mvScaleFactors[nPredictedLevel] */
    int32_t flags;     /* bit 0: set (the point holds what it takes); bit 1: skip */
    uint8_t desc[32];  /* MapPoint::GetDescriptor */
} orbx_guided_query; /* 56 bytes */
int orbx_search_guided_batch_device(orbx_ctx* ctx, int n, const orbx_kp* d_kps_un, int kp_cap, const int32_t* d_cell_start,
                                    const int32_t* d_cell_idx, const uint8_t* d_desc, const uint8_t* d_taken, float min_x,
                                    float min_y, float inv_w, float inv_h, const orbx_guided_query* d_queries, int q_cap,
                                    const int32_t* d_nq, int level_up, int max_dist, int check_orientation, int32_t* d_match,
                                    int32_t* d_query_match, int32_t* d_nmatches, int sync);

/* Replaces the projections in front of the two guided searches, for a batch of frames / keyframes; the records written
 * are the queries of orbx_search_guided_batch_device, radius = th * mvScaleFactors[level], flags 1 for
 * a projected point and 3 for one that is skipped on input (flags bit 1) or dropped, angle = d_angles[f*q_cap + i]
 * (pKF->mvKeysUn[i].angle for the rotation check; d_angles may be null: 0).  d_n_projected[f] counts the projected ones.
 * relocalisation = 0: ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (reference
 * src/ORBmatcher.cc:321-360), the tests of orbx_fuse_project_batch_device; d_poses / d_centres hold the pose taken out
 * of the similarity (Rcw = sRcw / scw, tcw / scw, Ow: :299-303).  relocalisation = 1:
 * ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:1534-1563): no depth and no viewing
 * test, the Frame's closed image bounds, the distance region, PredictScale. */
int orbx_project_guided_batch_device(orbx_ctx* ctx, int n, const orbx_frustum_params* params, const float* d_poses,
                                     const float* d_centres, const orbx_frustum_point* d_points, const float* d_angles,
                                     int q_cap, const int32_t* d_nq, int relocalisation, orbx_guided_query* d_queries_out,
                                     int32_t* d_n_projected, int sync);

/* ---- descriptor consumer: matching by vocabulary node -------------------------------------
 * Replaces ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (reference
 * src/ORBmatcher.cc:159-290; called from Tracking::TrackReferenceKeyFrame, src/Tracking.cc:894, and
 * Tracking::Relocalization, :1563) for a batch of (keyframe, frame) pairs: for every vocabulary node both
 * FeatureVectors hold, every keyframe feature with a good map point looks for its best and second best
 * descriptor distance among the frame's features of that node that no earlier keyframe feature took,
 * accepts a best distance <= 50 (TH_LOW) below nn_ratio (mfNNratio) times the second, and the rotation
 * histogram keeps the three main bins when check_orientation is set.  The FeatureVectors are what
 * orbx_bow_transform_batch_device leaves ((node, feature index) pairs, nodes ascending): pair f reads
 * d_fv_*_kf[f*kp_cap_kf ..], d_fv_*_f[f*kp_cap_f ..]; d_kps_* supply the angles (pKF->mvKeysUn, F.mvKeys);
 * d_valid_kf[f*kp_cap_kf + i] != 0 when the keyframe's feature i holds a map point that is not bad.
 * Outputs: d_match[f*kp_cap_f + k] = index of the keyframe feature whose map point vpMapPointMatches[k]
 * becomes (-1: NULL), d_nmatches[f] = the return value.  kp_cap_kf, kp_cap_f <= 16384. */
int orbx_search_by_bow_batch_device(orbx_ctx* ctx, int n, const uint32_t* d_fv_node_kf, const uint32_t* d_fv_idx_kf,
                                    const int32_t* d_fv_n_kf, const uint8_t* d_desc_kf, const orbx_kp* d_kps_kf,
                                    const uint8_t* d_valid_kf, int kp_cap_kf, const uint32_t* d_fv_node_f,
                                    const uint32_t* d_fv_idx_f, const int32_t* d_fv_n_f, const uint8_t* d_desc_f,
                                    const orbx_kp* d_kps_f, int kp_cap_f, float nn_ratio, int check_orientation,
                                    int32_t* d_match, int32_t* d_nmatches, int sync);

/* Replaces ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (reference
 * src/ORBmatcher.cc:551-684; called from LoopClosing::ComputeSim3, src/LoopClosing.cc:268): the same node-by-node
 * matching between two keyframes.  Features of the second keyframe need a good map point too (d_valid_2), the
 * best distance must be below TH_LOW (< 50, not <=), and the result is indexed by the first keyframe's
 * features: d_match12[f*kp_cap_1 + i] = the feature of the second keyframe whose map point vpMatches12[i]
 * becomes (-1: NULL).  Everything else as orbx_search_by_bow_batch_device. */
int orbx_search_by_bow_keyframes_batch_device(orbx_ctx* ctx, int n, const uint32_t* d_fv_node_1, const uint32_t* d_fv_idx_1,
                                              const int32_t* d_fv_n_1, const uint8_t* d_desc_1, const orbx_kp* d_kps_1,
                                              const uint8_t* d_valid_1, int kp_cap_1, const uint32_t* d_fv_node_2,
                                              const uint32_t* d_fv_idx_2, const int32_t* d_fv_n_2, const uint8_t* d_desc_2,
                                              const orbx_kp* d_kps_2, const uint8_t* d_valid_2, int kp_cap_2, float nn_ratio,
                                              int check_orientation, int32_t* d_match12, int32_t* d_nmatches, int sync);

/* Replaces ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (reference
 * src/ORBmatcher.cc:686-852 with CheckDistEpipolarLine, :140-157; called from LocalMapping::CreateNewMapPoints,
 * src/LocalMapping.cc:329) for a batch of keyframe pairs: between the features that hold no map point yet
 * (d_has_mp_* == 0), node by node of the FeatureVectors, every feature of the first keyframe takes the candidate
 * of smallest descriptor distance <= 50 (TH_LOW; the last of equals) that is far enough from the epipole (when
 * neither feature has a stereo measurement: d_u_right_* < 0 or null) and within 3.84 * mvLevelSigma2 of the
 * epipolar line of F12; a feature of the second keyframe may serve several of the first, as in the reference,
 * which never sets vbMatched2.  d_kps_un_*: mvKeysUn; d_f12[f*9 ..]: F12 row-major as ComputeF12 returns it
 * (src/LocalMapping.cc:624); d_epipole[f*2 ..]: (ex, ey) of :693-699; scale factors and level sigmas are the
 * context's.  Outputs: d_match12[f*kp_cap_1 + i] = the second keyframe's feature (-1: none), so that
 * vMatchedPairs is every (i, d_match12[i] >= 0) in index order (:841-849); d_nmatches[f] = the return value.
 * kp_cap_1, kp_cap_2 <= 16384; descriptors 16-byte aligned. */
int orbx_search_for_triangulation_batch_device(orbx_ctx* ctx, int n, const uint32_t* d_fv_node_1, const uint32_t* d_fv_idx_1,
                                               const int32_t* d_fv_n_1, const uint8_t* d_desc_1, const orbx_kp* d_kps_un_1,
                                               const uint8_t* d_has_mp_1, const float* d_u_right_1, int kp_cap_1,
                                               const uint32_t* d_fv_node_2, const uint32_t* d_fv_idx_2, const int32_t* d_fv_n_2,
                                               const uint8_t* d_desc_2, const orbx_kp* d_kps_un_2, const uint8_t* d_has_mp_2,
                                               const float* d_u_right_2, int kp_cap_2, const float* d_f12,
                                               const float* d_epipole, int only_stereo, int check_orientation,
                                               int32_t* d_match12, int32_t* d_nmatches, int sync);

/* ---- descriptor consumer: the matcher of the monocular initialisation ----------------------
 * Replaces ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched,
 * vector<int>& vnMatches12, int windowSize) (reference src/ORBmatcher.cc:418-549; called from
 * Tracking::MonocularInitialization, src/Tracking.cc:667, with windowSize 100 and an ORBmatcher(0.9, true))
 * for a batch of (F1, F2) pairs.  Every level-0 keypoint i of F1 searches the square window of half side
 * window_size around d_prev_matched[f*kp_cap_1 + i] (x, y) among F2's level-0 features (F2's undistorted
 * keypoints, grid and descriptors as orbx_undistort_grid_batch_device leaves them; the grid geometry as in
 * orbx_search_window_batch_device), skips features an earlier keypoint holds with a distance at least as
 * small, accepts a best distance <= 50 (TH_LOW) below nn_ratio times the second best, and takes the
 * feature away from its previous holder; the rotation histogram keeps the three main bins when
 * check_orientation is set.  Outputs: d_match12[f*kp_cap_1 + i] = vnMatches12[i] (-1 also beyond
 * d_n_1[f]), d_nmatches[f] = the return value, and d_prev_matched is updated in place with the matched
 * F2 keypoints' positions (:544-546).  kp_cap_1, kp_cap_2 <= 8192; descriptors 16-byte aligned. */
int orbx_search_for_initialization_batch_device(orbx_ctx* ctx, int n, const orbx_kp* d_kps_un_1, const uint8_t* d_desc_1,
                                                const int32_t* d_n_1, int kp_cap_1, float* d_prev_matched,
                                                const orbx_kp* d_kps_un_2, const int32_t* d_cell_start_2,
                                                const int32_t* d_cell_idx_2, const uint8_t* d_desc_2, int kp_cap_2,
                                                float min_x, float min_y, float inv_w, float inv_h, int window_size,
                                                float nn_ratio, int check_orientation, int32_t* d_match12,
                                                int32_t* d_nmatches, int sync);

/* ---- the stage after the extractor: stereo association ---------------------------------
 * Replaces Frame::ComputeStereoMatches (reference src/Frame.cc:485-659, descriptor distance
 * src/ORBmatcher.cc:1684-1700): for every left keypoint the best right keypoint in its row
 * band by Hamming distance, refined by an 11x11 window distance over 11 shifts on the
 * keypoint's pyramid level and a parabola fit, then the cut at 1.5*1.4*median.  It reads the
 * pyramids where the extractions left them in device memory, so in stereo mode no pyramid has
 * to be copied to the host (the reference reads mvImagePyramid at :492,582,594,599).
 * mb = stereo baseline in metres, mbf = baseline * fx (src/Frame.cc:114); outputs are the
 * reference's mvuRight and mvDepth: -1 where a left keypoint has no match. */

/* Left image `img_left` of `left`'s last orbx_extract / orbx_extract_batch against right
 * image `img_right` of `right`'s (the reference's two extractor instances; both contexts on
 * one device, same parameters and image size; `left` and `right` may be the same context).
 * Writes *n_left values to the host arrays u_right and depth (capacity `cap` each). */
int orbx_stereo_match(orbx_ctx* left, int img_left, orbx_ctx* right, int img_right, float mb,
                      float mbf, float* u_right, float* depth, int cap, int* n_left);

/* Throughput form, everything in device memory: pair p = images (2p, 2p+1) of ctx's last
 * orbx_extract_batch_device; d_kps, d_desc, d_n are the arrays that call filled (kp_cap as
 * passed there, and its level-0 input must still be valid if it was used in place);
 * d_u_right and d_depth hold n_pairs * kp_cap floats, row p for the left image of pair p. */
int orbx_stereo_match_batch_device(orbx_ctx* ctx, int n_pairs, const orbx_kp* d_kps, int kp_cap,
                                   const uint8_t* d_desc, const int32_t* d_n, float mb, float mbf,
                                   float* d_u_right, float* d_depth, int sync);

/* orbx_stereo_match_batch_device for callers that cannot allocate device memory: code under HIP
 * graph capture elsewhere in the process, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the
 * reference.  Same arguments, same results; the kernels' scratch (row table, bucketed right
 * keypoints, window distances) lives in a fixed array inside the library instead of buffers the
 * context grows on demand, and the call makes no hipMalloc, hipFree or other allocating call.
 * Limit: n_pairs * (4 * (image rows + 1) + 16 * kp_cap + 4) bytes, each of the four parts rounded
 * up to 256, must fit ORBX_NOALLOC_SCRATCH_BYTES (8 MiB: 248 KITTI-shape pairs at kp_cap 2008);
 * beyond that, or kp_cap > 16384, the call returns ORBX_E_CAPACITY and does nothing.  Each GPU
 * the library runs on holds one copy of the array until the process ends, and every call on
 * that GPU -- from any context -- uses it, one after the other: a call waits on its stream for
 * the kernels of the previous call on that GPU (an event per GPU, recorded behind each call's
 * kernels), so two contexts on one GPU take turns on the array and never overlap on it.  Nothing
 * in the array has to start at zero; it keeps the last call's contents.  The launches go to the
 * context's own stream like those of every other entry point: a caller's stream capture does not
 * see them. */
#define ORBX_NOALLOC_SCRATCH_BYTES ((size_t)8 << 20)
int orbx_stereo_match_batch_device_noalloc(orbx_ctx* ctx, int n_pairs, const orbx_kp* d_kps, int kp_cap,
                                           const uint8_t* d_desc, const int32_t* d_n, float mb, float mbf,
                                           float* d_u_right, float* d_depth, int sync);

/* Copies pyramid level `level` of image `img_idx` of the LAST extraction to host memory
 * (un-padded, rows dst_stride apart).  Replaces reads of the public member mvImagePyramid
 * (reference include/ORBextractor.h:85; read by src/Frame.cc:492,582,594,599). */
int orbx_pyramid_level(orbx_ctx* ctx, int img_idx, int level, uint8_t* dst, int dst_stride,
                       int* width, int* height);

/* mvImagePyramid as ComputePyramid leaves it (reference src/ORBextractor.cc:1227-1245: every level is the ROI at
 * (border, border) of a buffer with a `border`-pixel BORDER_REFLECT_101 rim, EDGE_THRESHOLD = 19 there), for callers
 * that read the public member on the host (the reference's own Frame::ComputeStereoMatches, src/Frame.cc:582-599).
 * orbx_set_keep_pyramid(ctx, border > 0): from now on every single-image extraction (orbx_extract, orbx_extract_batch
 * with n = 1) also builds the padded levels on the device and copies them, inside the same submission, into pinned
 * host memory the context owns; border = 0 switches that off again (the default).  orbx_padded_level then gives the
 * address of padded level `level` of the last such extraction: *base points at the buffer's first byte (the rim's
 * corner), rows are *stride = width + 2 * border bytes apart, the level itself starts at base + border * stride +
 * border.  The memory stays valid and unchanged until the next extraction, orbx_set_keep_pyramid or orbx_destroy on
 * the context: read it in place (no copy), or copy it out if it must outlive the next call. */
int orbx_set_keep_pyramid(orbx_ctx* ctx, int border);
int orbx_padded_level(orbx_ctx* ctx, int level, const uint8_t** base, int* width, int* height, int* stride);

/* ---- introspection for tests and measurement ------------------------------------------- */

enum {
    ORBX_DBG_BLURRED = 1,     /* level image after the 7x7 Gaussian, w*h bytes               */
    ORBX_DBG_CANDIDATES = 2,  /* int32 triplets (x, y, response), (cell row, cell col, y, x)  */
    ORBX_DBG_LEVEL_KPS = 3    /* int32 triplets (x, y, response) of retained points, level
                                 coordinates, quadtree list order                             */
};
/* Reads an intermediate of the last extraction into dst (capacity cap_bytes); *n_bytes
 * receives the size.  No counterpart in the reference (its intermediates are locals).
 * After an orbx_extract_batch call that ran in chunks (16 frames or more) the candidate and
 * retained-point lists have been reused chunk by chunk: those two reads return ORBX_E_STATE;
 * the blurred levels and the pyramid exist for every frame of the batch. */
int orbx_debug_read(orbx_ctx* ctx, int what, int img_idx, int level, void* dst, size_t cap_bytes,
                    size_t* n_bytes);

/* Diagnostic switches.  None changes results; each lets a test or a measurement reach a path
 * ordinary inputs seldom take.  Their initial values come from the environment, read once in
 * orbx_create (ORBX_NO_OVERLAP, ORBX_RESIZE_TILES, ORBX_STEREO_BORDER_PATH, ORBX_TRACE,
 * ORBX_NO_GRAPH, ORBX_ORDERED: set = 1; ORBX_PIPE_CHUNK=<frames>); this call changes one of them on a live
 * context after waiting for its work in flight.  No counterpart in the reference. */
enum {
    ORBX_FLAG_NO_OVERLAP = 1,         /* all stages on one stream, in stage order                   */
    ORBX_FLAG_RESIZE_TILES = 2,       /* every pyramid level through the LDS-tiled resize kernel     */
    ORBX_FLAG_STEREO_BORDER_PATH = 3, /* every stereo window through the reflecting byte loader      */
    ORBX_FLAG_TRACE = 4,              /* timing prints of the host batch path on stderr              */
    ORBX_FLAG_NO_GRAPH = 5,           /* orbx_extract issues its launches one by one, no HIP graph   */
    ORBX_FLAG_PIPE_CHUNK = 6,         /* frames per chunk of the pipelined host batch path (0: auto) */
    ORBX_FLAG_INIT_LIST = 7,          /* candidates per list of orbx_search_for_initialization_batch_device (0: default) */
    ORBX_FLAG_ORDERED = 9,            /* orbx_extract_batch_device: no batch's pyramid and blur run ahead of the
                                         main stream (the order every other entry point uses)         */
    ORBX_FLAG_POISON = 8              /* test hook: from the next batch on, the device-side count of frame 0's first
                                         FAST cell (1), of its first level's retained points (2), or its first
                                         candidate point (3) is overwritten with an out-of-range value after its
                                         producer has run; 0 switches it off.  The batch must end in ORBX_E_DEVICE_DATA. */
};
int orbx_set_debug_flag(orbx_ctx* ctx, int flag, int value);

/* Stage timing with HIP events on the context's stream.  Stages, in order:
 * 0 pyramid, 1 fast, 2 quadtree, 3 blur, 4 orient+describe. */
#define ORBX_NUM_STAGES 5
int orbx_set_profiling(orbx_ctx* ctx, int enable);
/* Milliseconds per stage of the last extraction (valid after it has been synchronised). */
int orbx_get_stage_ms(orbx_ctx* ctx, float ms[ORBX_NUM_STAGES]);
const char* orbx_stage_name(int stage);

const char* orbx_strerror(int code);
/* hipError_t of the last failed HIP call on this context (0 if none). */
int orbx_last_hip_error(const orbx_ctx* ctx);
const char* orbx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
