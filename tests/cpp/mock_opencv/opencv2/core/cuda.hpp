// TEST-ONLY stand-in for the slice of OpenCV the reference-signature branch of include/orbfe_adaptor.hpp touches
// (cv::Mat header, cv::cuda::HostMem, cv::Point2f, CV_8UC1), so that the branch is compiled and type-checked in an image
// without OpenCV.  Not OpenCV, not shipped: plain host memory behind the same member names.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <memory>

#define CV_8UC1 0

namespace cv {

struct Point2f {
    float x, y;
};

class Mat {
public:
    Mat() = default;
    Mat(int r, int c, int /*type*/, void* d, size_t s) : rows(r), cols(c), data(static_cast<uint8_t*>(d)), step(s) {}
    uint8_t* ptr(int i) { return data + (size_t)i * step; }
    const uint8_t* ptr(int i) const { return data + (size_t)i * step; }
    int rows = 0, cols = 0;
    uint8_t* data = nullptr;
    size_t step = 0;
};

namespace cuda {

class HostMem {
public:
    enum class AllocType { PAGE_LOCKED = 1, SHARED = 2, WRITE_COMBINED = 4 };
    HostMem() = default;
    HostMem(int r, int c, int type, AllocType = AllocType::PAGE_LOCKED) : rows(r), cols(c), type_(type)
    {
        step = ((size_t)c + 63) / 64 * 64;  // a padded pitch, like a real allocation may have
        buf_ = std::shared_ptr<uint8_t>(static_cast<uint8_t*>(std::calloc((size_t)(r > 0 ? r : 1), step)), std::free);
        data = buf_.get();
    }
    Mat createMatHeader() const { return Mat(rows, cols, type_, data, step); }
    int rows = 0, cols = 0;
    uint8_t* data = nullptr;
    size_t step = 0;

private:
    int type_ = 0;
    std::shared_ptr<uint8_t> buf_;  // reference-counted like the real one
};

}  // namespace cuda
}  // namespace cv
