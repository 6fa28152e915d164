// TEST-ONLY stand-in for the few OpenCV types include/ORBextractor.h touches, so that the
// drop-in class can be compiled and exercised in environments without OpenCV.  It is never
// used to build reference code and is not part of the product; with a real OpenCV on the
// include path this directory is simply left out.
#ifndef ORBX_TEST_CV_STUB_CORE_HPP
#define ORBX_TEST_CV_STUB_CORE_HPP

#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

namespace cv {

typedef unsigned char uchar;

template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T _x, T _y) : x(_x), y(_y) {}
};
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;

struct Rect {
    int x, y, width, height;
    Rect(int _x, int _y, int _w, int _h) : x(_x), y(_y), width(_w), height(_h) {}
};
struct Size {
    int width, height;
    Size(int w, int h) : width(w), height(h) {}
    bool operator!=(const Size& o) const { return width != o.width || height != o.height; }
};

class KeyPoint {
public:
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};

class Mat {
public:
    Mat() : rows(0), cols(0), step(0), data(0) {}
    Mat(int r, int c, int /*type*/) : rows(0), cols(0), step(0), data(0) { create(r, c, CV_8UC1); }
    Mat(int r, int c, int /*type*/, void* ext, size_t st) : rows(r), cols(c), step(st), data((uchar*)ext) {}
    void create(int r, int c, int /*type*/) {
        if (r == rows && c == cols && data && step == (size_t)c) return;
        buf.reset(new std::vector<uchar>((size_t)r * c));
        rows = r;
        cols = c;
        step = (size_t)c;
        data = buf->empty() ? 0 : &(*buf)[0];
    }
    void release() {
        buf.reset();
        rows = cols = 0;
        step = 0;
        data = 0;
    }
    bool empty() const { return data == 0 || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    bool isContinuous() const { return step == (size_t)cols || rows <= 1; }
    Size size() const { return Size(cols, rows); }
    uchar* ptr(int i = 0) { return data + (size_t)i * step; }
    const uchar* ptr(int i = 0) const { return data + (size_t)i * step; }
    Mat operator()(const Rect& r) const {
        Mat m;
        m.buf = buf;
        m.rows = r.height;
        m.cols = r.width;
        m.step = step;
        m.data = data + (size_t)r.y * step + r.x;
        return m;
    }
    int rows, cols;
    size_t step;
    uchar* data;

private:
    std::shared_ptr<std::vector<uchar> > buf;
};

class _InputArray {
public:
    _InputArray() : m(0) {}
    _InputArray(const Mat& mm) : m(&mm) {}
    bool empty() const { return !m || m->empty(); }
    Mat getMat() const { return m ? *m : Mat(); }

private:
    const Mat* m;
};
class _OutputArray {
public:
    _OutputArray(Mat& mm) : m(&mm) {}
    void create(int r, int c, int type) const { m->create(r, c, type); }
    void release() const { m->release(); }
    Mat getMat() const { return *m; }

private:
    Mat* m;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv
#endif
