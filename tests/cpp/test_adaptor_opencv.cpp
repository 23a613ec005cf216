// The exact reference signatures of the adaptor -- extractFeatures(const cv::cuda::HostMem&) (include/ORBextractor.h:62) and
// ConvertImageToGPU(const cv::Mat&) (image_grabber.hpp:96-110) -- compiled against tests/cpp/mock_opencv (no OpenCV in
// this image).  Without arguments: type check + link only.  With "<W> <H> <gray.raw>": both overloads on the GPU must
// return the same keypoints and descriptor bytes.
#define ORBFE_WITH_OPENCV
#include <cstdio>
#include <cstring>
#include <fstream>
#include <optional>
#include <tuple>
#include <type_traits>

#include "orbfe_adaptor.hpp"

using ORB_SLAM3::KeyPoint;
using ORB_SLAM3::ORBextractor;

// the reference's declaration, checked at compile time
using RefResult = std::optional<std::tuple<std::shared_ptr<std::vector<KeyPoint>>, cv::cuda::HostMem>>;
static_assert(std::is_same<decltype(std::declval<ORBextractor&>().extractFeatures(std::declval<const cv::cuda::HostMem&>())), RefResult>::value,
              "extractFeatures(const cv::cuda::HostMem&) must keep the reference's return type");
static_assert(std::is_same<decltype(std::declval<ORB_SLAM3::ImagePreparer&>().ConvertImageToGPU(std::declval<const cv::Mat&>())),
                           cv::cuda::HostMem>::value,
              "ConvertImageToGPU(const cv::Mat&) must keep the reference's return type");

int main(int argc, char** argv)
{
    std::printf("%s\n", orbfe_version());
    if (argc < 4) return 0;
    const int W = std::atoi(argv[1]), H = std::atoi(argv[2]);
    cv::cuda::HostMem im(H, W, CV_8UC1, cv::cuda::HostMem::AllocType::SHARED);
    std::vector<uint8_t> raw((size_t)W * H);
    std::ifstream(argv[3], std::ios::binary).read(reinterpret_cast<char*>(raw.data()), (std::streamsize)raw.size());
    cv::Mat m = im.createMatHeader();
    for (int y = 0; y < H; y++) std::memcpy(m.ptr(y), raw.data() + (size_t)y * W, (size_t)W);
    ORBextractor ex(1000, 40000, 1.2f, 8, 20, 7, W, H);
    auto a = ex.extractFeatures(im);
    auto b = ex.extractFeatures(ORB_SLAM3::GrayImageView{raw.data(), W});
    if (!a || !b) return 2;
    auto& [ka, da] = *a;
    auto& [kb, db] = *b;
    if (ka->size() != kb->size() || std::memcmp(ka->data(), kb->data(), ka->size() * sizeof(KeyPoint)) != 0) return 3;
    const cv::Mat dm = da.createMatHeader();
    if (dm.rows != (int)ka->size() || dm.cols != 32) return 4;
    for (int i = 0; i < dm.rows; i++)
        if (std::memcmp(dm.ptr(i), db.data() + (size_t)i * 32, 32) != 0) return 5;
    std::printf("hostmem overload ok: %zu keypoints\n", ka->size());
    return 0;
}
