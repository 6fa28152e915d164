/* Drop-in replacement for ORB_SLAM2::ORBextractor on AMD MI355X (gfx950).
 *
 * Same namespace, class name, constructor, call operator, getters and public member as the
 * reference header (reference include/ORBextractor.h:29-113), so that Frame.cc, Tracking.cc and
 * KeyFrame.h compile against it unchanged.  The work is done by liborbx (include/orbx.h); this
 * class is a thin host wrapper: it owns one device context, created on the first call for the
 * image size it sees (and re-created if a larger image arrives).
 *
 * Differences a maintainer should know about:
 *  - the header includes <opencv2/core.hpp> and <opencv2/features2d.hpp> instead of the legacy
 *    <opencv/cv.h> of the reference (line 26), which OpenCV 4 no longer ships;
 *  - protected members of the reference (pattern, umax, ComputePyramid, ...) are gone: they
 *    were never part of the contract;
 *  - environment: ORBX_DEVICE selects the HIP device ordinal (default 0);
 *    ORBX_KEEP_PYRAMID=0 skips the device-to-host copy that keeps mvImagePyramid valid (only
 *    stereo matching reads it, reference src/Frame.cc:492-599);
 *  - the matrices in mvImagePyramid are headers over memory the instance owns and refills on every
 *    call (the reference allocates new buffers per call): a cv::Mat copied out of the member shows the
 *    next image after the next call -- clone() it to keep it;
 *  - a GPU failure is fatal (message on stderr, abort): the reference API has no error channel
 *    and there is deliberately no CPU fallback.
 */
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <list>
#include <vector>

#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>

struct orbx_ctx;

namespace ORB_SLAM2 {

/* Kept for source compatibility (reference include/ORBextractor.h:32-43); the GPU quadtree does
 * not use it. */
class ExtractorNode {
public:
    ExtractorNode() : bNoMore(false) {}

    void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4);

    std::vector<cv::KeyPoint> vKeys;
    cv::Point2i UL, UR, BL, BR;
    std::list<ExtractorNode>::iterator lit;
    bool bNoMore;
};

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();

    // Compute the ORB features and descriptors on an image.  The mask is ignored, as in the
    // reference.
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                    cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    std::vector<cv::Mat> mvImagePyramid;

    // Extension, not in the reference header: the body of Frame::ComputeStereoMatches
    // (reference src/Frame.cc:485-659) evaluated on the GPU from the pyramids, keypoints and
    // descriptors this (left) extractor and `right` produced in their last operator() calls.
    // vuRight / vDepth are the reference's mvuRight / mvDepth (-1 = no match); mb and mbf are the
    // Frame members of the same names.  With this call in place Frame no longer reads
    // mvImagePyramid, so the pyramid copy to the host can be switched off (ORBX_KEEP_PYRAMID=0).
    void ComputeStereoMatches(ORBextractor& right, float mb, float mbf, std::vector<float>& vuRight,
                              std::vector<float>& vDepth);

    // Extension, not in the reference header: what the stereo Frame constructor does with two extractor
    // instances on two threads and the association behind them (reference src/Frame.cc:78-81, :90), as ONE
    // call on this one instance and one submission to the GPU: both images are extracted in the same
    // launches and Frame::ComputeStereoMatches follows on the device.  Outputs as operator() gives them per
    // image, plus mvuRight / mvDepth of the left keypoints.  mvImagePyramid is not touched (nothing reads it
    // once the association runs on the device).
    void ExtractStereoPair(cv::InputArray imLeft, cv::InputArray imRight, std::vector<cv::KeyPoint>& keysLeft,
                           cv::OutputArray descLeft, std::vector<cv::KeyPoint>& keysRight, cv::OutputArray descRight,
                           float mb, float mbf, std::vector<float>& vuRight, std::vector<float>& vDepth);

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;

    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;

private:
    ORBextractor(const ORBextractor&);
    ORBextractor& operator=(const ORBextractor&);
    void EnsureContext(int width, int height);
    int EnsureBuffers(int width, int height, int frames);

    std::vector<unsigned char> mvKpBuf, mvDescBuf;  // results as the library writes them, kept across calls

    orbx_ctx* mpCtx;
    int mnCtxWidth, mnCtxHeight;    // size the device context was created for
    int mnLastWidth, mnLastHeight;  // size of the last image
    int mnDevice;
    bool mbKeepPyramid;
};

}  // namespace ORB_SLAM2

#endif
