// cv_compat.h -- the handful of OpenCV value types the adapters touch.
//
// In a real EAO-Fusion checkout OpenCV is present and this header only forwards to it.  This build image has no
// OpenCV, so for compile- and run-testing the adapters a minimal stand-in with the same member names is provided
// (cv::Mat for CV_8UC1 / CV_32FC1, cv::KeyPoint, cv::Point2f, InputArray/OutputArray as Mat references).  It is NOT
// used to build any reference source -- only this repository's own adapters and tests.
#pragma once

#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>) && !defined(EAOFUSION_FORCE_CV_COMPAT)
#define EAOFUSION_HAVE_OPENCV 1
#endif
#endif

#ifdef EAOFUSION_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else

#include <cassert>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_32FC1 5

namespace cv {

struct Point2f {
    float x = 0, y = 0;
    Point2f() {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct KeyPoint {  // same field order and size (28 bytes) as OpenCV's
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
    KeyPoint() {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t stp = 0) : rows(r), cols(c), data((unsigned char*)ext), type_(type) {
        step = stp ? stp : (size_t)c * elemSize();
    }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type;
        step = (size_t)c * elemSize();
        buf_ = std::shared_ptr<std::vector<unsigned char>>(new std::vector<unsigned char>((size_t)r * step));
        data = buf_->data();
    }
    void release() { buf_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * elemSize());
        return m;
    }
    void copyTo(Mat& dst) const {      // (OpenCV: dst.create(size, type) keeps a buffer of the right shape, then the rows are copied)
        if (dst.data == nullptr || dst.rows != rows || dst.cols != cols || dst.type_ != type_ || dst.data == data) { if (dst.data != data || dst.data == nullptr) dst = clone(); return; }
        for (int y = 0; y < rows; y++) std::memcpy(dst.data + (size_t)y * dst.step, data + (size_t)y * step, (size_t)cols * elemSize());
    }
    static Mat zeros(int r, int c, int type) { Mat m(r, c, type); std::memset(m.data, 0, (size_t)r * m.step); return m; }
    static Mat eye(int r, int c, int type) {
        Mat m = zeros(r, c, type);
        for (int i = 0; i < r && i < c; i++) m.at<float>(i, i) = 1.f;
        return m;
    }
    template <typename T> T* ptr(int y = 0) { return reinterpret_cast<T*>(data + (size_t)y * step); }
    template <typename T> const T* ptr(int y = 0) const { return reinterpret_cast<const T*>(data + (size_t)y * step); }
    unsigned char* ptr(int y = 0) { return data + (size_t)y * step; }
    const unsigned char* ptr(int y = 0) const { return data + (size_t)y * step; }
    template <typename T> T& at(int y, int x) { return ptr<T>(y)[x]; }
    template <typename T> const T& at(int y, int x) const { return ptr<T>(y)[x]; }
    template <typename T> T& at(int i) { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
    template <typename T> const T& at(int i) const { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
    // view of the w x h block at (x, y) sharing this matrix' storage (upstream: temp(Rect(x, y, w, h)))
    Mat roi(int x, int y, int w, int h) const { Mat m(h, w, type_, const_cast<unsigned char*>(data) + (size_t)y * step + (size_t)x * elemSize(), step); m.buf_ = buf_; return m; }
    Mat row(int y) const { Mat m(1, cols, type_, const_cast<unsigned char*>(data) + (size_t)y * step, step); m.buf_ = buf_; return m; }

private:
    int type_ = CV_8U;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};

// just enough of the proxy types for ORBextractor::operator()
class _InputArray {
public:
    _InputArray() {}
    _InputArray(const Mat& m) : m_(&m) {}
    bool empty() const { return !m_ || m_->empty(); }
    Mat getMat() const { return m_ ? *m_ : Mat(); }
private:
    const Mat* m_ = nullptr;
};
class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int r, int c, int type) const { m_->create(r, c, type); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
private:
    Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
inline Mat noArray() { return Mat(); }

}  // namespace cv
#endif  // EAOFUSION_HAVE_OPENCV
