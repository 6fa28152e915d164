/* CPU oracle for the ORB extractor hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A single-threaded, OpenCV-free restatement of ORB_SLAM2::ORBextractor
 * (reference: src/ORBextractor.cc, include/ORBextractor.h) plus the five OpenCV 3.4.5
 * primitives it calls (cv::resize INTER_LINEAR 8UC1, cv::FAST 9/16 with NMS,
 * cv::GaussianBlur 7x7 sigma 2 on 8U, cv::fastAtan2, cvRound).
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this
 * path, and OpenCV is not installed in the build environment, so the reference itself
 * cannot be compiled or run.  The restatement is pinned only by hand-derivable
 * known-answer tests (tests/test_oracle_kat.py) and by its own committed golden vectors.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the CPU baseline -- never on the product path.
 *
 * Pinned decisions where the reference is not bit-deterministic:
 *   D1  descriptor rotation uses the FMA-contracted form the reference's -march=native
 *       build produces: col = rint(fmaf(x, a, -(y*b))), row = rint(fmaf(x, b, y*a)).
 *   D2  quadtree node sort ties (pointer values in the reference) break on creation
 *       sequence: among equal sizes the most recently created node is split first.
 *   D3  fastAtan2 polynomial in IEEE float32, separate multiply and add.
 *   D5  Gaussian taps 18,34,49,55,49,34,18 (sum 257, OpenCV 3.4.5 fixed-point path).
 *   D7  stereo matching (src/Frame.cc:485-659): float32 statements evaluated as written, no
 *       contraction across statements (bestuR = scale*(...) is rounded before uL - bestuR);
 *       an empty match list leaves every output at -1 (the reference indexes an empty
 *       vector at :646); right keypoints whose row band leaves [0, rows) only fill the rows
 *       that exist (the reference writes out of bounds at :511).
 *   D8  input staging restates OpenCV 3.4.5's 8-bit paths: gray = 14-bit fixed point with
 *       coefficients 1868/9617/4899; remap = maps rounded to 1/32 px (round half to even),
 *       15-bit bilinear weights.  Later OpenCV releases changed the gray coefficients.
 *   D9  cv::undistortPoints as in OpenCV 3.4.5: exactly 5 iterations, IEEE double, no contraction.
 *   D10 bag of words: the vocabulary loader of the reference appends, for a text file ending in a
 *       newline, one more child of the root whose centre is uninitialised memory; that node is not
 *       reproduced.  The node id of a feature whose branch ends above level L - levelsup is 0 (the
 *       reference leaves it unset).  Sums and norms are IEEE double in the reference's order.
 *   D11 SearchByProjection (src/ORBmatcher.cc:1365-1507): the pose products go through cv::gemm as
 *       OpenCV 3.4.5 evaluates them -- Rcw*x3Dw+tcw and Rlw*twc+tlw on its 3x3 float path (products
 *       summed left to right in float32, then the addend), -Rcw.t()*tcw on its general path (sums in
 *       double, scaled by -1, rounded to float32); the other float32 statements as written, no
 *       contraction.  A feature that two map points take (possible only when the first has no
 *       observations) ends with the later one, and is cleared when either entry falls in a rejected
 *       rotation bin, as the reference's loops leave it.
 *   D6  a = cos, b = sin of the float32 radian angle are float32 roundings of a
 *       double-precision evaluation (fixed range reduction + fdlibm kernels, explicit
 *       fma) -- the correctly rounded value on every input checked.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16

typedef struct orc_kp {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_kp; /* 28 bytes, field order of cv::KeyPoint */

typedef struct orc_cand {
    int32_t x, y;      /* relative to the level's (16,16) origin, as DistributeOctTree sees them */
    int32_t response;  /* FAST score */
} orc_cand;

typedef struct orc_params {
    int nfeatures;
    float scale_factor;
    int nlevels;
    int ini_th_fast;
    int min_th_fast;
    /* derived (reference ctor, src/ORBextractor.cc:416-494) */
    float scale[ORC_MAX_LEVELS], inv_scale[ORC_MAX_LEVELS];
    float sigma2[ORC_MAX_LEVELS], inv_sigma2[ORC_MAX_LEVELS];
    int quota[ORC_MAX_LEVELS];
    int umax[16];
} orc_params;

typedef struct orc_ctx orc_ctx; /* holds every intermediate of the last extraction */

/* constructor arithmetic; returns 0 or a negative error */
int orc_params_init(orc_params* p, int nfeatures, float scale_factor, int nlevels,
                    int ini_th, int min_th);
void orc_level_size(const orc_params* p, int w0, int h0, int level, int* w, int* h);

/* primitives */
void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride,
                       uint8_t* dst, int dw, int dh, int dstride);
void orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
/* cv::FAST(img, kps, threshold, nonmax=true); returns count, writes up to cap (x,y,score) */
int orc_fast(const uint8_t* img, int w, int h, int stride, int threshold,
             orc_cand* out, int cap);
/* per-level candidate list in (cell row, cell col, y, x) order; returns count */
int orc_level_candidates(const uint8_t* img, int w, int h, int stride, int ini_th, int min_th,
                         orc_cand* out, int cap);
/* DistributeOctTree: writes indices (into cands) of retained points in list order */
int orc_distribute_octree(const orc_cand* cands, int n, int minX, int maxX, int minY, int maxY,
                          int N, int32_t* out_idx, int cap);
/* measurement only: tie order of equal-size nodes, 0 = decision D2 (default), 1 = reverse, 2 = pseudo-random */
void orc_set_d2_mode(int mode);
/* as above; *d2_sensitive = 1 when the retained set depends on the pinned tie-break D2 */
int orc_distribute_octree_ex(const orc_cand* cands, int n, int minX, int maxX, int minY, int maxY,
                             int N, int32_t* out_idx, int cap, int* d2_sensitive);
float orc_fast_atan2(float y, float x);
float orc_ic_angle(const uint8_t* img, int stride, int x, int y, const int* umax);
void orc_sincos_deg(float angle_deg, float* cos_out, float* sin_out);
void orc_descriptor(const uint8_t* blurred, int stride, int x, int y, float angle_deg,
                    uint8_t desc[32]);
/* the descriptor under the alternatives to decisions D1 (rotation) and D6 (sincos), see the source */
void orc_descriptor_variant(const uint8_t* blurred, int stride, int x, int y, float angle_deg,
                            int rotation, int sincos, uint8_t desc[32]);
const int8_t* orc_pattern(void); /* 1024 values */

/* whole path */
orc_ctx* orc_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
void orc_destroy(orc_ctx* c);
const orc_params* orc_get_params(const orc_ctx* c);
/* returns number of keypoints, or negative error */
int orc_extract(orc_ctx* c, const uint8_t* img, int w, int h, int stride);
int orc_num_keypoints(const orc_ctx* c);
void orc_get_keypoints(const orc_ctx* c, orc_kp* out);
void orc_get_descriptors(const orc_ctx* c, uint8_t* out);
/* intermediates of the last orc_extract */
void orc_get_level_dims(const orc_ctx* c, int level, int* w, int* h);
void orc_get_level(const orc_ctx* c, int level, uint8_t* out);          /* w*h, un-blurred */
void orc_get_blurred(const orc_ctx* c, int level, uint8_t* out);        /* w*h */
int orc_get_num_candidates(const orc_ctx* c, int level);
void orc_get_candidates(const orc_ctx* c, int level, orc_cand* out);
int orc_get_num_level_keypoints(const orc_ctx* c, int level);
/* 19-pixel reflect-101 padded level as ComputePyramid builds it: (w+38)*(h+38) */
void orc_get_level_padded(const orc_ctx* c, int level, uint8_t* out);


/* ---- next stage after the extractor: stereo association ---------------------------------- */
/* ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:1684-1700): Hamming distance of
   two 32-byte descriptors. */
int orc_descriptor_distance(const uint8_t a[32], const uint8_t b[32]);
/* Frame::ComputeStereoMatches (reference src/Frame.cc:485-659) on the keypoints, descriptors and
   pyramids left by the last orc_extract of `left` and `right` (same extractor parameters).
   mb = baseline in metres, mbf = baseline * fx (src/Frame.cc:114).  u_right, depth: one value
   per left keypoint, -1 where there is no match.  sad (may be null): the window distance of
   each accepted match before the median cut, -1 elsewhere.  Returns the matches kept. */
int orc_stereo_match(const orc_ctx* left, const orc_ctx* right, float mb, float mbf,
                     float* u_right, float* depth, int32_t* sad);


/* ---- the stage before the extractor: input staging ---------------------------------------- */
/* cv::cvtColor(src, dst, CV_{BGR,RGB,BGRA,RGBA}2GRAY) on 8-bit images, the call of
   Tracking::GrabImage* (reference src/Tracking.cc:175-199, 213-231, 250-268).  OpenCV 3.4.5
   fixed-point form: (B*1868 + G*9617 + R*4899 + 8192) >> 14.  channels = 3 or 4 (alpha ignored),
   blue_idx = 0 for BGR(A) order, 2 for RGB(A). */
void orc_cvt_gray(const uint8_t* src, int w, int h, int sstride, int channels, int blue_idx,
                  uint8_t* dst, int dstride);
/* cv::remap(src, dst, mapx, mapy, INTER_LINEAR) with CV_32FC1 maps, 8UC1 image, BORDER_CONSTANT 0:
   the rectification of the stereo examples (reference Examples/Stereo/stereo_euroc.cc:137-138).
   OpenCV 3.4.5 fixed-point form: map coordinates rounded to 1/32 px, bilinear weights
   (32-fx)(32-fy)*32 ... summing to 32768, result (sum + 16384) >> 15.  Maps are dw x dh floats,
   rows map_stride floats apart. */
void orc_remap_linear(const uint8_t* src, int sw, int sh, int sstride, const float* mapx,
                      const float* mapy, int map_stride, uint8_t* dst, int dw, int dh, int dstride);


/* ---- right after the extractor: undistorted keypoints and the frame grid -------------------- */
/* Frame::UndistortKeyPoints (reference src/Frame.cc:421-451): cv::undistortPoints(pts, pts, K, D,
   Mat(), K) of OpenCV 3.4.5 -- normalise, 5 fixed-point iterations of the inverse distortion
   model in double precision, re-project with K, round to float.  K = {fx, fy, cx, cy},
   dist = {k1, k2, p1, p2, k3}.  dist[0] == 0 copies the input (:423-427).  xy: n (x, y) pairs. */
void orc_undistort_points(const float* xy_in, int n, const float K[4], const float dist[5], float* xy_out);
/* Frame::AssignFeaturesToGrid + PosInGrid (reference src/Frame.cc:232-247, 398-408): cell
   (round((x-min_x)*inv_w), round((y-min_y)*inv_h)) of a 64 x 48 grid; points outside are
   dropped.  CSR output: cell c = gx*48 + gy holds idx[cell_start[c] .. cell_start[c+1]) in
   ascending keypoint order (the push_back order).  cell_start has 64*48+1 entries. */
void orc_assign_grid(const float* xy, int n, float min_x, float min_y, float inv_w, float inv_h,
                     int32_t* cell_start, int32_t* idx);

/* ---- descriptor consumer: the bag-of-words transform ---------------------------------------- */
/* Frame::ComputeBoW (reference src/Frame.cc:411-418) = TemplatedVocabulary::transform(features,
   BowVector&, FeatureVector&, levelsup = 4) of the vendored DBoW2
   (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1194, per feature :1218-1260;
   FORB::distance, FORB.cpp:81-101; BowVector::addWeight / addIfNotExist / normalize,
   BowVector.cpp:25-84; FeatureVector::addFeature, FeatureVector.cpp:31-45).
   The vocabulary is a tree: node 0 is the root, node i > 0 has parent[i] < i, a 32-byte centre,
   a weight, and is_leaf[i] as the text file of loadFromTextFile (:1338-1425) states it.  Children
   are kept in node order (the push_back order); word ids are given to the nodes flagged is_leaf in
   node order.  scoring / weighting are the file header's enum values (ORBvoc.txt: 0 0 = L1_NORM,
   TF_IDF).  The ORB-SLAM2 vocabulary file itself is not in the reference checkout; the tests use
   hand-built and seeded random trees. */
typedef struct orc_voc orc_voc;
orc_voc* orc_voc_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                        const uint8_t* is_leaf, const uint8_t* desc /* n_nodes x 32 */, const double* weight);
void orc_voc_destroy(orc_voc* v);
int orc_voc_num_words(const orc_voc* v);
/* One feature down the tree (:1218-1260): the child with the smallest Hamming distance at each
   level, the first one on ties; *node = the node met at level L - levelsup (0 when that level is
   not positive; the reference leaves it unset when the leaf is met earlier, here it stays 0). */
void orc_voc_transform_feature(const orc_voc* v, const uint8_t desc[32], int levelsup, uint32_t* word,
                               uint32_t* node, double* weight);
/* All n features of a frame (:1127-1194).  bow_id / bow_val: the BowVector in word order (n
   entries of room); fv_node / fv_idx: the FeatureVector as (node, feature index) pairs in node
   order, indices ascending inside a node (n entries of room). */
void orc_voc_transform(const orc_voc* v, const uint8_t* descs, int n, int levelsup, uint32_t* bow_id,
                       double* bow_val, int* n_bow, uint32_t* fv_node, uint32_t* fv_idx, int* n_fv);

/* ---- descriptor consumer: best match inside a window ------------------------------------- */
/* Frame::GetFeaturesInArea (reference src/Frame.cc:338-396) followed by the candidate loop of
   ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (reference
   src/ORBmatcher.cc:1427-1455): among the frame's features inside the square window, on the
   allowed pyramid levels, not `taken` (the reference's test mvpMapPoints[i2] &&
   Observations() > 0, as the caller sees it at call time) and, when the feature has a right
   coordinate (u_right[i2] > 0), within `radius` of the predicted one (`ur`), the one at the
   smallest descriptor distance below 256, the first such in the order the grid lists them.
   kps_xy: n (x, y) pairs (mvKeysUn), octave: n levels, desc: n x 32, cell_start / cell_idx: the
   CSR grid of orc_assign_grid.  u_right and taken may be NULL.  Returns the feature index or -1;
   *best_dist receives the distance (256 when none). */
int orc_search_window(const float* kps_xy, const int32_t* octave, const uint8_t* desc, const float* u_right,
                      const uint8_t* taken, const int32_t* cell_start, const int32_t* cell_idx, float min_x,
                      float min_y, float inv_w, float inv_h, float u, float v, float radius, int min_level,
                      int max_level, float ur, const uint8_t q_desc[32], int* best_dist);

/* ---- descriptor consumer: ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) ---- */
/* The whole function (reference src/ORBmatcher.cc:1365-1507): every map point of the last frame is
   projected with the current pose, searched in its window (orc_search_window), and a match at
   distance <= TH_HIGH (100) is written to the current frame at once, so that later points skip
   features held by a point with observations; then the rotation histogram (30 bins,
   ComputeThreeMaxima :1638-1679) clears the matches outside the three largest bins.
   q[i]: the last frame's feature i -- world position and descriptor of its map point, the octave
   of mvKeys[i], the angle of mvKeysUn[i]; flags bit 0 = the map point has Observations() > 0,
   bit 1 = skip (no map point, or mvbOutlier).  angle: the current frame's mvKeysUn angles.
   taken0 (may be NULL): features that hold a map point with observations before the call.
   Tcw / Tlw: rows of the 3x4 [R|t] of the current / last frame.  scale_factors: mvScaleFactors.
   match[n_kp]: per feature the index i of the map point written last, -1 = none or cleared;
   query_match[nq]: per i the feature it was written to (before the rotation check), or -1.
   Returns nmatches. */
typedef struct orc_map_query {
    float x, y, z;
    int32_t octave;
    float angle;
    int32_t flags;
    uint8_t desc[32];
} orc_map_query; /* 56 bytes */
typedef struct orc_proj_params {
    float fx, fy, cx, cy, mbf, mb;
    float min_x, min_y, max_x, max_y, inv_w, inv_h; /* mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElement{Width,Height}Inv */
    float th;
    int32_t mono, check_orientation;
} orc_proj_params;
int orc_search_by_projection(const float* kps_xy, const int32_t* octave, const float* angle, const uint8_t* desc,
                             const float* u_right, const uint8_t* taken0, int n_kp, const int32_t* cell_start,
                             const int32_t* cell_idx, const float* scale_factors, const orc_proj_params* P,
                             const float Tcw[12], const float Tlw[12], const orc_map_query* q, int nq,
                             int32_t* match, int32_t* query_match);

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:45-129) */
typedef struct orc_local_query {
    float u, v, ur;    /* MapPoint::mTrackProjX, mTrackProjY, mTrackProjXR */
    int32_t level;     /* mnTrackScaleLevel */
    float view_cos;    /* mTrackViewCos */
    int32_t flags;     /* bit 0: Observations() > 0; bit 1: skip (!mbTrackInView or isBad()) */
    uint8_t desc[32];  /* GetDescriptor() */
} orc_local_query;
int orc_search_local_map(const float* kps_xy, const int32_t* octave, const uint8_t* desc, const float* u_right,
                         const uint8_t* taken0, int n_kp, const int32_t* cell_start, const int32_t* cell_idx,
                         const float* scale_factors, float min_x, float min_y, float inv_w, float inv_h, float th,
                         float nn_ratio, const orc_local_query* q, int nq, int32_t* match, int32_t* query_match);

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (reference src/ORBmatcher.cc:159-290).
 * FeatureVectors as (node, feature index) pairs sorted by node, then index. */
int orc_search_by_bow(const uint32_t* fv_node_kf, const uint32_t* fv_idx_kf, int fv_n_kf, const uint8_t* desc_kf,
                      const float* angle_kf, const uint8_t* valid_kf, const uint32_t* fv_node_f, const uint32_t* fv_idx_f,
                      int fv_n_f, const uint8_t* desc_f, const float* angle_f, int n_f, float nn_ratio,
                      int check_orientation, int32_t* match);

/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (reference src/ORBmatcher.cc:551-684). */
int orc_search_by_bow_keyframes(const uint32_t* fv_node_1, const uint32_t* fv_idx_1, int fv_n_1, const uint8_t* desc_1,
                                const float* angle_1, const uint8_t* valid_1, int n_1, const uint32_t* fv_node_2,
                                const uint32_t* fv_idx_2, int fv_n_2, const uint8_t* desc_2, const float* angle_2,
                                const uint8_t* valid_2, int n_2, float nn_ratio, int check_orientation, int32_t* match12);

/* Frame::ComputeStereoFromRGBD (reference src/Frame.cc:662-682) incl. the depth conversion of
 * Tracking::GrabImageRGBD (src/Tracking.cc:230-231). */
void orc_stereo_from_rgbd(const void* depth, int is_f32, int width, int height, int stride_bytes, float factor,
                          const float* kps_xy, const float* un_x, int n, float mbf, float* u_right, float* depth_out);

/* ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:418-549); prev_matched is vbPrevMatched, in and out. */
int orc_search_for_initialization(const int32_t* octave1, const float* angle1, const uint8_t* desc1, int n1,
                                  float* prev_matched, const float* kps2_xy, const int32_t* octave2, const float* angle2,
                                  const uint8_t* desc2, int n2, const int32_t* cell_start2, const int32_t* cell_idx2,
                                  float min_x, float min_y, float inv_w, float inv_h, int window_size, float nn_ratio,
                                  int check_orientation, int32_t* match12);

/* ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:686-852, CheckDistEpipolarLine :140-157). */
int orc_search_for_triangulation(const uint32_t* fv_node_1, const uint32_t* fv_idx_1, int fv_n_1, const uint8_t* desc_1,
                                 const float* kps1_xy, const float* angle_1, const uint8_t* has_mp_1, const float* u_right_1,
                                 int n_1, const uint32_t* fv_node_2, const uint32_t* fv_idx_2, int fv_n_2, const uint8_t* desc_2,
                                 const float* kps2_xy, const float* angle_2, const int32_t* octave_2, const uint8_t* has_mp_2,
                                 const float* u_right_2, int n_2, const float* f12, float ex, float ey,
                                 const float* scale_factors, const float* level_sigma2, int only_stereo,
                                 int check_orientation, int32_t* match12);

/* The search of ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) for one projected map point (reference
 * src/ORBmatcher.cc:929-986). */
int orc_fuse_search(const float* kps_xy, const int32_t* octave, const uint8_t* desc, const float* u_right,
                    const int32_t* cell_start, const int32_t* cell_idx, float min_x, float min_y, float inv_w, float inv_h,
                    float u, float v, float ur, float radius, int nPredictedLevel, const float* inv_level_sigma2,
                    const uint8_t q_desc[32], int* best_dist);

/* The matching loops of ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (reference
 * src/ORBmatcher.cc:1565-1632) and SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (:362-398) on projected
 * map points; the records are read as (u, v, angle, predicted level, radius, flags, descriptor). */
int orc_search_guided(const float* kps_xy, const int32_t* octave, const float* angle, const uint8_t* desc, int n_kp,
                      const uint8_t* taken0, const int32_t* cell_start, const int32_t* cell_idx, float min_x, float min_y,
                      float inv_w, float inv_h, const orc_local_query* q, int n_q, int level_up, int max_dist,
                      int check_orientation, int32_t* match, int32_t* query_match);

/* DBoW2 L1Scoring::score (reference Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68). */
double orc_bow_score_l1(const uint32_t* id1, const double* v1, int n1, const uint32_t* id2, const double* v2, int n2);

/* log(x) of a positive finite float in double from IEEE operations only; (float) of it is the correctly rounded logf. */
double orc_log_pinned(float x);

/* Frame::isInFrustum (reference src/Frame.cc:271-327) with MapPoint::PredictScale (src/MapPoint.cc:402-417). */
int orc_is_in_frustum(const float Tcw[12], const float Ow[3], const float cam[5], const float bounds[4],
                      float viewing_cos_limit, float log_scale_factor, int n_levels, const float P[3], const float Pn[3],
                      float min_distance, float max_distance, float* proj_x, float* proj_y, float* proj_xr, int* scale_level,
                      float* view_cos);

/* The tests in front of the search of ORBmatcher::Fuse(pKF, vpMapPoints, th) (reference src/ORBmatcher.cc:881-927). */
int orc_fuse_project(const float Tcw[12], const float Ow[3], const float cam[5], const float bounds[4],
                     float log_scale_factor, int n_levels, const float* scale_factors, float th, const float P[3],
                     const float Pn[3], float min_distance, float max_distance, float* u_out, float* v_out, float* ur_out,
                     float* radius, int* predicted_level);

/* The projection of ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (reference
 * src/ORBmatcher.cc:1534-1563). */
int orc_reloc_project(const float Tcw[12], const float Ow[3], const float cam[5], const float bounds[4], float log_scale_factor,
                      int n_levels, const float* scale_factors, float th, const float P[3], float min_distance,
                      float max_distance, float* u_out, float* v_out, float* radius, int* predicted_level);

/* The projection of one direction of ORBmatcher::SearchBySim3 (reference src/ORBmatcher.cc:1195-1226). */
int orc_sim3_project(const float T1w[12], const float S21[12], const float cam[5], const float bounds[4], float log_scale_factor,
                     int n_levels, const float* scale_factors, float th, const float P[3], float min_distance,
                     float max_distance, float* u_out, float* v_out, float* radius, int* predicted_level);

#ifdef __cplusplus
}
#endif
#endif
