// ORBextractor.h -- drop-in replacement for the reference's include/ORBextractor.h + src/ORBextractor.cc.
//
// Same namespace, class name, constructor, operator(), getters and public mvImagePyramid member as
// reference include/ORBextractor.h:45-111, so Frame.cc (:616-622, :846-953) and Tracking.cc (:143-149, 296, 386-397)
// compile unchanged; all compute happens in libeaofusion_hip.so through the C-ABI (include/eao_fusion.h).
// Header-only: add -I<repo>/include and link -leaofusion_hip; remove src/ORBextractor.cc from the build.
//
// Error behaviour mirrors upstream: an empty image returns without touching the outputs; zero keypoints releases
// the descriptor matrix; a non-CV_8UC1 image asserts.  Library failures (no MI355X, unsupported geometry) throw
// std::runtime_error -- upstream has no error channel here and silently continuing on a CPU path is not an option.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <cassert>
#include <list>
#include <stdexcept>
#include <string>
#include <vector>

#include "../eao_fusion.h"
#include "cv_compat.h"

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures_, float scaleFactor_, int nlevels_, int iniThFAST_, int minThFAST_)
        : nfeatures(nfeatures_), scaleFactor(scaleFactor_), nlevels(nlevels_), iniThFAST(iniThFAST_), minThFAST(minThFAST_) {
        eao_orb_cfg cfg = {nfeatures_, scaleFactor_, nlevels_, iniThFAST_, minThFAST_};
        check(eao_orb_create(&cfg, &h_), "eao_orb_create");
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels);
        check(eao_orb_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                             mnFeaturesPerLevel.data()), "eao_orb_tables");
        mvImagePyramid.resize(nlevels);
    }
    ~ORBextractor() { eao_orb_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // Compute the ORB features and descriptors on an image; the mask is ignored, as upstream (include/ORBextractor.h:58).
    void operator()(cv::InputArray image_, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors_) {
        if (image_.empty()) return;
        cv::Mat image = image_.getMat();
        assert(image.type() == CV_8UC1);
        int cap = 0;
        check(eao_orb_max_keypoints(h_, image.cols, image.rows, &cap), "eao_orb_max_keypoints");
        kp_.resize(cap);
        desc_.resize((size_t)cap * 32);
        int n = 0;
        static_assert(sizeof(cv::KeyPoint) == sizeof(eao_keypoint), "cv::KeyPoint must be the 28-byte POD the ABI mirrors");
        check(eao_orb_extract(h_, image.ptr(0), image.cols, image.rows, (int)image.step, kp_.data(), desc_.data(), cap, &n), "eao_orb_extract");
        keypoints.clear();
        if (n == 0) {
            descriptors_.release();
        } else {
            descriptors_.create(n, 32, CV_8U);
            cv::Mat d = descriptors_.getMat();
            for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), &desc_[(size_t)i * 32], 32);
            keypoints.resize(n);
            std::memcpy((void*)keypoints.data(), kp_.data(), (size_t)n * sizeof(eao_keypoint));
        }
        refreshPyramid();
    }

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Level images WITH the 19 px BORDER_REFLECT_101 frame, as upstream stores them (src/ORBextractor.cc:1113-1128).
    // Only Frame::ComputeStereoMatches reads them (stereo sensor); set keepPyramid = false to skip the copy otherwise.
    std::vector<cv::Mat> mvImagePyramid;
    bool keepPyramid = true;

protected:
    static void check(eao_status st, const char* what) {
        if (st != EAO_OK) throw std::runtime_error(std::string(what) + ": " + eao_last_error());
    }
    void refreshPyramid() {
        if (!keepPyramid) return;
        const int E = 19;
        for (int l = 0; l < nlevels; l++) {
            int w = 0, h = 0;
            check(eao_orb_level(h_, 0, l, 0, &w, &h, nullptr), "eao_orb_level");
            tight_.resize((size_t)w * h);
            check(eao_orb_level(h_, 0, l, 0, nullptr, nullptr, tight_.data()), "eao_orb_level");
            cv::Mat whole(h + 2 * E, w + 2 * E, CV_8UC1);
            for (int y = -E; y < h + E; y++) {
                const int sy = y < 0 ? -y : (y >= h ? 2 * h - 2 - y : y);
                unsigned char* dst = whole.ptr(y + E);
                const unsigned char* src = &tight_[(size_t)sy * w];
                for (int x = -E; x < w + E; x++) dst[x + E] = src[x < 0 ? -x : (x >= w ? 2 * w - 2 - x : x)];
            }
#ifdef EAOFUSION_HAVE_OPENCV
            mvImagePyramid[l] = whole(cv::Rect(E, E, w, h));   // w x h view with the border physically around it, as upstream
#else
            mvImagePyramid[l] = whole.roi(E, E, w, h);
#endif
        }
    }

    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

public:
    // the library handle (its image pyramid of the last extraction stays on the device: eaofusion::ComputeStereoMatches)
    eao_orb* handle() const { return h_; }

private:
    eao_orb* h_ = nullptr;
    std::vector<eao_keypoint> kp_;
    std::vector<unsigned char> desc_, tight_;
};

}  // namespace ORB_SLAM2

namespace eaofusion {

// Frame::ComputeStereoMatches() -- reference src/Frame.cc:841-1013.  The body of the member becomes
//   eaofusion::ComputeStereoMatches(*this);
// right after the two extractors ran on the stereo pair (as upstream: Frame's stereo constructor, src/Frame.cc:113-125):
// their pyramids are read in place on the device.  Members used: N, mvKeys, mvKeysRight, mDescriptors, mDescriptorsRight,
// mpORBextractorLeft / Right, mb, mbf, mvuRight, mvDepth.
template <class FrameT>
void ComputeStereoMatches(FrameT& F) {
    static_assert(sizeof(cv::KeyPoint) == sizeof(eao_keypoint), "cv::KeyPoint is the 28-byte POD the C-ABI mirrors");
    const int N = F.N, Nr = (int)F.mvKeysRight.size();
    F.mvuRight = std::vector<float>(N, -1.0f);
    F.mvDepth = std::vector<float>(N, -1.0f);
    if (N == 0 || Nr == 0) return;
    const eao_status st = eao_compute_stereo_matches(F.mpORBextractorLeft->handle(), F.mpORBextractorRight->handle(), 0, N,
                                                     reinterpret_cast<const eao_keypoint*>(F.mvKeys.data()), F.mDescriptors.ptr(0), Nr,
                                                     reinterpret_cast<const eao_keypoint*>(F.mvKeysRight.data()), F.mDescriptorsRight.ptr(0),
                                                     F.mb, F.mbf, F.mvuRight.data(), F.mvDepth.data());
    if (st != EAO_OK) throw std::runtime_error(std::string("eao_compute_stereo_matches: ") + eao_last_error());
}

}  // namespace eaofusion

#endif  // ORBEXTRACTOR_H
