/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  *** parity unpinned *** (see orc_common.h)
 *
 * SURVEY 8(f) rank 4, the image side of the dataset reader:
 *   orc_undistort          PhotometricUndistorter::processFrame (util/Undistort.cpp:214-251) followed by the remap loop of Undistort::undistort
 *                          (:435-530): data = G[raw] (* vignetteMapInv) or factor * raw, out = bilinear(data, remapX, remapY), 0 where remapX < 0;
 *                          passthrough (remapX == NULL) copies data. The benchmark noise / blur branches (benchmark_varNoise, applyBlurNoise :536-635) are off
 *                          by default (util/settings.cpp:214-216) and not restated.
 *   orc_resize_nearest_u8  cv::resize(.., INTER_NEAREST) as IOWrap::resizeMask / resizeColor call it (IOWrapper/OpenCV/ImageRW_OpenCV.cpp:55-85);
 *                          OpenCV is a third-party dependency that is absent here: restated from its published resizeNN
 *                          (sx = min(cvFloor(x * ifx), ssize.width - 1), ifx = 1 / inv_scale_x, inv_scale_x = (double)dsize.width / ssize.width).
 */
#include "orc_common.h"

void orc_undistort(const void* raw, int bpp, int wOrg, int hOrg, const float* G, const float* vinv, int photometric, float factor,
                   const float* remapX, const float* remapY, int w, int h, float* out) {
    const int wh = wOrg * hOrg;
    float* data = (float*)malloc(sizeof(float) * wh);
    for (int i = 0; i < wh; i++) {
        const unsigned v = bpp == 1 ? ((const uint8_t*)raw)[i] : ((const uint16_t*)raw)[i];
        if (photometric == 0) data[i] = factor * v;
        else { data[i] = G[v]; if (photometric == 2) data[i] *= vinv[i]; }
    }
    if (!remapX) { memcpy(out, data, sizeof(float) * w * h); free(data); return; }
    for (int idx = w * h - 1; idx >= 0; idx--) {
        float xx = remapX[idx], yy = remapY[idx];
        if (xx < 0) out[idx] = 0;
        else {
            int xxi = xx, yyi = yy;
            xx -= xxi; yy -= yyi;
            float xxyy = xx * yy;
            const float* src = data + xxi + yyi * wOrg;
            out[idx] = xxyy * src[1 + wOrg] + (yy - xxyy) * src[wOrg] + (xx - xxyy) * src[1] + (1 - xx - yy + xxyy) * src[0];
        }
    }
    free(data);
}

void orc_resize_nearest_u8(const uint8_t* src, int wOrg, int hOrg, int channels, uint8_t* dst, int w, int h) {
    const double ifx = 1.0 / ((double)w / wOrg), ify = 1.0 / ((double)h / hOrg);
    for (int y = 0; y < h; y++) {
        int sy = (int)floor(y * ify); if (sy > hOrg - 1) sy = hOrg - 1;
        for (int x = 0; x < w; x++) {
            int sx = (int)floor(x * ifx); if (sx > wOrg - 1) sx = wOrg - 1;
            for (int k = 0; k < channels; k++) dst[((size_t)y * w + x) * channels + k] = src[((size_t)sy * wOrg + sx) * channels + k];
        }
    }
}
