// Test harness for the drop-in C++ class (include/ORBextractor.h).
//   harness tables <nfeatures> <scale> <nlevels>                 -> prints the getter tables
//   harness extract <nf> <scale> <nl> <ini> <min> <w> <h> <stride> <in.raw> <out.bin>
//   harness stereo <nf> <scale> <nl> <w> <h> <left.raw> <right.raw> <mb> <mbf> <out.bin>
//       out.bin: int32 N, N floats mvuRight, N floats mvDepth
//   harness threads <nf> <scale> <nl> <w> <h> <pairs> <frames.raw> <mb> <mbf> <out.bin> <threaded 0|1>
//       frames.raw: 2*pairs frames (left, right, left, ...).  Two extractor instances live for the whole
//       run; per pair each one is called from a thread of its own that is created for the call and
//       joined after it (threaded = 1; the calling thread does both calls when 0), then the left
//       instance matches against the right; threaded = 2: ONE instance, ExtractStereoPair per pair.
//       Prints the median time of a pair (extractions + association, no file output) on stdout.  out.bin per pair: int32 Kl, Kl*28 + Kl*32 bytes, int32 Kr,
//       Kr*28 + Kr*32 bytes, int32 N, N floats mvuRight, N floats mvDepth.
// extract's out.bin: int32 K, K*28 bytes keypoints (cv::KeyPoint field order), K*32 bytes descriptors,
// then per level: int32 w, int32 h, (w+38)*(h+38) bytes of the padded pyramid buffer.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ORBextractor.h"

namespace {
struct Extraction {
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
};
// The body of Frame::ExtractORB (reference src/Frame.cc:249-255).
void ExtractORB(ORB_SLAM2::ORBextractor* ex, const cv::Mat* im, Extraction* out) { (*ex)(*im, cv::Mat(), out->kps, out->desc); }
void WriteExtraction(FILE* o, const Extraction& e) {
    int K = (int)e.kps.size();
    std::fwrite(&K, 4, 1, o);
    for (int i = 0; i < K; i++) {
        float v[5] = {e.kps[i].pt.x, e.kps[i].pt.y, e.kps[i].size, e.kps[i].angle, e.kps[i].response};
        int q[2] = {e.kps[i].octave, e.kps[i].class_id};
        std::fwrite(v, 4, 5, o);
        std::fwrite(q, 4, 2, o);
    }
    for (int i = 0; i < K; i++) std::fwrite(e.desc.ptr(i), 1, 32, o);
}
}  // namespace

int main(int argc, char** argv) {
    if (argc >= 5 && !std::strcmp(argv[1], "tables")) {
        ORB_SLAM2::ORBextractor ex(std::atoi(argv[2]), (float)std::atof(argv[3]), std::atoi(argv[4]), 20, 7);
        std::printf("levels %d scale %.9g\n", ex.GetLevels(), ex.GetScaleFactor());
        std::vector<float> a = ex.GetScaleFactors(), b = ex.GetInverseScaleFactors(), c = ex.GetScaleSigmaSquares(),
                           d = ex.GetInverseScaleSigmaSquares();
        for (size_t i = 0; i < a.size(); i++) std::printf("%.9g %.9g %.9g %.9g\n", a[i], b[i], c[i], d[i]);
        return 0;
    }
    if (argc >= 12 && !std::strcmp(argv[1], "extract")) {
        const int nf = std::atoi(argv[2]), nl = std::atoi(argv[4]), ini = std::atoi(argv[5]), mn = std::atoi(argv[6]);
        const float sc = (float)std::atof(argv[3]);
        const int w = std::atoi(argv[7]), h = std::atoi(argv[8]), stride = std::atoi(argv[9]);
        std::vector<unsigned char> raw((size_t)stride * h);
        FILE* f = std::fopen(argv[10], "rb");
        if (!f || std::fread(&raw[0], 1, raw.size(), f) != raw.size()) return 2;
        std::fclose(f);
        cv::Mat img(h, w, CV_8UC1, &raw[0], (size_t)stride);
        ORB_SLAM2::ORBextractor ex(nf, sc, nl, ini, mn);
        std::vector<cv::KeyPoint> kps;
        cv::Mat desc;
        ex(cv::Mat(), cv::Mat(), kps, desc);  // empty image: outputs untouched
        if (!kps.empty() || !desc.empty()) return 3;
        ex(img, cv::Mat(), kps, desc);
        ex(img, cv::Mat(), kps, desc);  // second call on the same context
        FILE* o = std::fopen(argv[11], "wb");
        if (!o) return 4;
        int K = (int)kps.size();
        std::fwrite(&K, 4, 1, o);
        for (int i = 0; i < K; i++) {
            float v[5] = {kps[i].pt.x, kps[i].pt.y, kps[i].size, kps[i].angle, kps[i].response};
            int q[2] = {kps[i].octave, kps[i].class_id};
            std::fwrite(v, 4, 5, o);
            std::fwrite(q, 4, 2, o);
        }
        for (int i = 0; i < K; i++) std::fwrite(desc.ptr(i), 1, 32, o);
        for (int l = 0; l < nl; l++) {
            const cv::Mat& roi = ex.mvImagePyramid[l];
            int dims[2] = {roi.cols, roi.rows};
            std::fwrite(dims, 4, 2, o);
            const unsigned char* base = roi.data - 19 * roi.step - 19;
            for (int y = 0; y < roi.rows + 38; y++) std::fwrite(base + (size_t)y * roi.step, 1, roi.cols + 38, o);
        }
        std::fclose(o);
        return 0;
    }
    if (argc >= 12 && !std::strcmp(argv[1], "stereo")) {
        const int nf = std::atoi(argv[2]), nl = std::atoi(argv[4]);
        const float sc = (float)std::atof(argv[3]);
        const int w = std::atoi(argv[5]), h = std::atoi(argv[6]);
        std::vector<unsigned char> rawl((size_t)w * h), rawr((size_t)w * h);
        FILE* fl = std::fopen(argv[7], "rb");
        FILE* fr = std::fopen(argv[8], "rb");
        if (!fl || !fr || std::fread(&rawl[0], 1, rawl.size(), fl) != rawl.size() ||
            std::fread(&rawr[0], 1, rawr.size(), fr) != rawr.size())
            return 2;
        std::fclose(fl);
        std::fclose(fr);
        cv::Mat iml(h, w, CV_8UC1, &rawl[0], (size_t)w), imr(h, w, CV_8UC1, &rawr[0], (size_t)w);
        // two instances, as Tracking constructs them (reference src/Tracking.cc:121-125)
        ORB_SLAM2::ORBextractor left(nf, sc, nl, 20, 7), right(nf, sc, nl, 20, 7);
        std::vector<cv::KeyPoint> kl, kr;
        cv::Mat dl, dr;
        left(iml, cv::Mat(), kl, dl);
        right(imr, cv::Mat(), kr, dr);
        std::vector<float> u, d;
        left.ComputeStereoMatches(right, (float)std::atof(argv[9]), (float)std::atof(argv[10]), u, d);
        if (u.size() != kl.size() || d.size() != kl.size()) return 5;
        FILE* o = std::fopen(argv[11], "wb");
        if (!o) return 4;
        int N = (int)u.size();
        std::fwrite(&N, 4, 1, o);
        if (N) {
            std::fwrite(&u[0], 4, (size_t)N, o);
            std::fwrite(&d[0], 4, (size_t)N, o);
        }
        std::fclose(o);
        return 0;
    }
    if (argc >= 13 && !std::strcmp(argv[1], "threads")) {
        const int nf = std::atoi(argv[2]), nl = std::atoi(argv[4]);
        const float sc = (float)std::atof(argv[3]);
        const int w = std::atoi(argv[5]), h = std::atoi(argv[6]), pairs = std::atoi(argv[7]);
        const int mode = std::atoi(argv[12]);
        const bool threaded = mode == 1;
        const float mb = (float)std::atof(argv[9]), mbf = (float)std::atof(argv[10]);
        std::vector<double> ms;
        std::vector<unsigned char> raw((size_t)w * h * 2 * pairs);
        FILE* f = std::fopen(argv[8], "rb");
        if (!f || std::fread(&raw[0], 1, raw.size(), f) != raw.size()) return 2;
        std::fclose(f);
        FILE* o = std::fopen(argv[11], "wb");
        if (!o) return 4;
        // the two instances Tracking keeps for the STEREO sensor (reference src/Tracking.cc:121-125)
        ORB_SLAM2::ORBextractor left(nf, sc, nl, 20, 7), right(nf, sc, nl, 20, 7);
        for (int p = 0; p < pairs; p++) {
            cv::Mat iml(h, w, CV_8UC1, &raw[(size_t)(2 * p) * w * h], (size_t)w);
            cv::Mat imr(h, w, CV_8UC1, &raw[(size_t)(2 * p + 1) * w * h], (size_t)w);
            Extraction el, er;
            std::vector<float> u, d;
            const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
            if (mode == 2) {
                left.ExtractStereoPair(iml, imr, el.kps, el.desc, er.kps, er.desc, mb, mbf, u, d);
            } else if (threaded) {
                // Frame::Frame (stereo), reference src/Frame.cc:78-81: two threads per frame, joined
                std::thread tl(ExtractORB, &left, &iml, &el);
                std::thread tr(ExtractORB, &right, &imr, &er);
                tl.join();
                tr.join();
            } else {
                ExtractORB(&left, &iml, &el);
                ExtractORB(&right, &imr, &er);
            }
            if (mode != 2) left.ComputeStereoMatches(right, mb, mbf, u, d);
            ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            if (u.size() != el.kps.size() || d.size() != el.kps.size()) return 5;
            WriteExtraction(o, el);
            WriteExtraction(o, er);
            int N = (int)u.size();
            std::fwrite(&N, 4, 1, o);
            if (N) {
                std::fwrite(&u[0], 4, (size_t)N, o);
                std::fwrite(&d[0], 4, (size_t)N, o);
            }
        }
        std::fclose(o);
        if (ms.size() > 4) ms.erase(ms.begin(), ms.begin() + 2);  // the first calls create contexts and capture graphs
        std::sort(ms.begin(), ms.end());
        std::printf("pair_ms median %.4f p10 %.4f p90 %.4f pairs %d mode %d\n", ms[ms.size() / 2], ms[ms.size() / 10],
                    ms[ms.size() * 9 / 10], (int)ms.size(), mode);
        return 0;
    }
    std::fprintf(stderr, "usage: see source\n");
    return 1;
}
