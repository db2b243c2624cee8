// Minimal stand-ins for the handful of OpenCV types the drop-in classes expose
// (cv::Mat, cv::KeyPoint, cv::Point_, InputArray/OutputArray).  Used ONLY when
// the OpenCV headers are not installed; with OpenCV present the real types are
// used and this file defines nothing.
#pragma once
#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#define ORB_HAVE_OPENCV 1
#endif
#endif

#ifdef ORB_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_32FC1 5

namespace cv {

typedef unsigned char uchar;

template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;

struct KeyPoint {  // field order and size as in OpenCV (28 bytes)
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1)
        : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};

// Dense 2-D matrix, 8-bit or 32-bit float, single channel, reference-counted
// storage, row views.  Only what the ORB front end's signatures need.
class Mat {
public:
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    size_t step = 0;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t stp = 0)  // wraps caller memory
        : rows(r), cols(c), data((uchar*)ext), step(stp ? stp : (size_t)c * esz(type)), type_(type) {}

    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && data && owner_) return;
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * esz(type);
        owner_.reset(new std::vector<uchar>((size_t)r * step));
        data = owner_->data();
    }
    void release() {
        owner_.reset();
        data = nullptr;
        rows = cols = 0;
        step = 0;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    int channels() const { return 1; }
    size_t elemSize() const { return esz(type_); }
    Mat clone() const {
        Mat m;
        if (empty()) return m;
        m.create(rows, cols, type_);
        for (int r = 0; r < rows; r++) memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * esz(type_));
        return m;
    }
    Mat row(int r) const {
        Mat m;
        m.rows = 1;
        m.cols = cols;
        m.type_ = type_;
        m.step = step;
        m.data = data + (size_t)r * step;
        m.owner_ = owner_;
        return m;
    }
    template <typename T>
    T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <typename T>
    const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    template <typename T>
    T& at(int r, int c) { return ((T*)(data + (size_t)r * step))[c]; }
    template <typename T>
    const T& at(int r, int c) const { return ((const T*)(data + (size_t)r * step))[c]; }
    template <typename T>
    T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <typename T>
    const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    Mat getMat() const { return *this; }

private:
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    int type_ = CV_8U;
    std::shared_ptr<std::vector<uchar>> owner_;
};

typedef const Mat& InputArray;
typedef Mat& OutputArray;

}  // namespace cv
#endif  // ORB_HAVE_OPENCV
