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
#include <cstring>
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
    // One library call; the results are copied ONCE, from the pinned block the device wrote them to into the caller's cv types.
    void operator()(cv::InputArray image_, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors_) {
        if (image_.empty()) return;
        cv::Mat image = image_.getMat();
        assert(image.type() == CV_8UC1);
        static_assert(sizeof(cv::KeyPoint) == sizeof(eao_keypoint), "cv::KeyPoint must be the 28-byte POD the ABI mirrors");
        if (keepPyramid != keptPyramid_) {
            check(eao_orb_set_keep_pyramid(h_, keepPyramid ? 19 : -1), "eao_orb_set_keep_pyramid");      // EDGE_THRESHOLD, src/ORBextractor.cc:67
            keptPyramid_ = keepPyramid;
        }
        const eao_keypoint* kp = nullptr;
        const uint8_t* desc = nullptr;
        int n = 0;
        check(eao_orb_extract_ref(h_, image.ptr(0), image.cols, image.rows, (int)image.step, &kp, &desc, &n), "eao_orb_extract_ref");
        keypoints.resize((size_t)n);
        if (n == 0) {
            descriptors_.release();
        } else {
            descriptors_.create(n, 32, CV_8U);
            cv::Mat d = descriptors_.getMat();
            if (d.isContinuous()) std::memcpy(d.ptr(0), desc, (size_t)n * 32);
            else for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), desc + (size_t)i * 32, 32);
            std::memcpy((void*)keypoints.data(), kp, (size_t)n * sizeof(eao_keypoint));
        }
        pyramidStale_ = true;
        if (keepPyramid) ImagePyramid();
    }

    // mvImagePyramid of the frame extracted last, materialised ON DEMAND: upstream fills it in every call (src/ORBextractor.cc:1107-1131), but only
    // Frame::ComputeStereoMatches ever reads it (src/Frame.cc:846, 936-953) -- and eaofusion::ComputeStereoMatches below does not need it either: it reads the
    // pyramids where they are, on the device.  A caller that keeps upstream's ComputeStereoMatches leaves keepPyramid at its default (true) and finds
    // the member filled after every call as before; anyone else sets keepPyramid = false and calls ImagePyramid() when it wants the images.  Either way the levels come
    // in ONE library call (eao_orb_pyramid): w x h views with the 19-pixel BORDER_REFLECT_101 frame physically around them, as upstream lays them out, inside a
    // pinned block the extractor's handle owns -- valid until the next operator() call.
    std::vector<cv::Mat>& ImagePyramid() {
        if (pyramidStale_) {
            eao_orb_level_view lv[16];
            check(eao_orb_pyramid(h_, 0, 19, lv), "eao_orb_pyramid");
            for (int l = 0; l < nlevels; l++) mvImagePyramid[l] = cv::Mat(lv[l].height, lv[l].width, CV_8UC1, lv[l].data, (size_t)lv[l].step);
            pyramidStale_ = false;
        }
        return mvImagePyramid;
    }

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Level images WITH the 19 px BORDER_REFLECT_101 frame around them, as upstream stores them (src/ORBextractor.cc:1113-1128): see ImagePyramid().
    // keepPyramid = true (the DEFAULT since round 6, ADVICE r5: an integrator who keeps upstream's Frame::ComputeStereoMatches must find the member filled
    // without knowing about this flag) fills the member in every operator() call (+ ~0.02 ms per 640 x 480 frame: one more launch behind the extraction, 1.3 MB
    // over PCIe); a caller that has replaced the stereo matcher or runs RGB-D / monocular sets it to false once and leaves the images to
    // ImagePyramid().
    std::vector<cv::Mat> mvImagePyramid;
    bool keepPyramid = true;

protected:
    static void check(eao_status st, const char* what) {
        if (st != EAO_OK) throw std::runtime_error(std::string(what) + ": " + eao_last_error());
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
    bool pyramidStale_ = true, keptPyramid_ = false;
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
