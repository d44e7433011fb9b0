// Minimal stand-in for the OpenCV 2.4 types the reference's three classes use at their boundary (cv::Mat, cv::KeyPoint,
// cv::InputArray / OutputArray): only what INTEGRATION.md's replacement bodies touch.  OpenCV does not exist in this image;
// tests/test_integration_snippets.py compiles the documented code blocks against this header so that they cannot rot.
// Written for this repository -- nothing here comes from OpenCV's sources.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32FC1 5

namespace cv {
struct Point2f { float x = 0, y = 0; };
struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1; };
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint is 28 bytes");

class Mat {
  public:
    unsigned char *data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;
    Mat() {}
    Mat(int r, int c, int t) { create(r, c, t); }
    void create(int r, int c, int t)
    {
        type_ = t;
        rows = r, cols = c;
        step = (size_t)c * elem(t);
        buf_ = std::shared_ptr<unsigned char>(new unsigned char[std::max<size_t>(step * (size_t)r, 1)], std::default_delete<unsigned char[]>());
        data = buf_.get();
    }
    void release() { buf_.reset(), data = nullptr, rows = cols = 0, step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    Mat clone() const { Mat m(rows, cols, type_); if (data) std::memcpy(m.data, data, step * (size_t)rows); return m; }
    template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + step * (size_t)r); }
    template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + step * (size_t)r); }
    template <typename T> T &at(int r, int c) { return ptr<T>(r)[c]; }

  private:
    static size_t elem(int t) { return t == CV_32F ? 4 : t == CV_8UC3 ? 3 : 1; }
    std::shared_ptr<unsigned char> buf_;
    int type_ = 0;
};

// InputArray / OutputArray as the reference's operator() sees them: a view of a Mat the caller owns
class _InputArray {
  public:
    _InputArray() {}
    _InputArray(const Mat &m) : m_(const_cast<Mat *>(&m)) {}
    Mat getMat() const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }

  protected:
    Mat *m_ = nullptr;
};
class _OutputArray : public _InputArray {
  public:
    _OutputArray(Mat &m) { m_ = &m; }
    void create(int r, int c, int t) const { m_->create(r, c, t); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
} // namespace cv
