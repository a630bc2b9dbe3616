/* On-disk formats of huziqi/NALO-SLAM, restated as plain C entry points of libnalo_gpu.so. Host code only (text files,
 * no device work); each function cites the reference code whose bytes / parsed values it reproduces. All return 0 on success, a negative
 * NALO_IO_* code otherwise. */
#ifndef NALO_IO_H
#define NALO_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { NALO_IO_OK = 0, NALO_IO_ERR_ARG = -1, NALO_IO_ERR_FILE = -2, NALO_IO_ERR_FORMAT = -3 };

/* result.txt -- FullSystem::printResult (FullSystem/FullSystem.cpp:445-499): TUM lines `timestamp tx ty tz qx qy qz qw`, stream precision 15, one
 * line per frame of allFrameHistory sorted by timestamp (FrameSort, :267). A frame with poseValid == 0 repeats the pose of the frame BEFORE it in the
 * sorted order (its own stored pose, valid or not), the first frame of the file prints seven zeros instead. The translation goes through Eigen's
 * operator<< of a row vector (default IOFormat: the three coefficients are right-aligned to their common width), reproduced here.
 * t: [n][3] camToWorld.translation(), q: [n][4] camToWorld.so3().unit_quaternion() as x y z w. */
int nalo_io_write_result(const char* path, int n, const double* timestamp, const uint8_t* poseValid, const double* t, const double* q);

/* pcl_data_tmp.pcd -- SampleOutputWrapper::publishKeyframes (IOWrapper/OutputWrapper/SampleOutputWrapper.h:110-134, 150-176): bare `x y z` lines
 * (no PCD header), default stream precision, of the back-projected points: depth = 1/idepth (float), x = (u*fxi + cxi)*depth, y = (v*fyi + cyi)*depth,
 * z = depth*(1 + 2*fxi) (float arithmetic, the reference's z term included), world = camToWorld(3x4, double) * (x, y, z, 1).
 * append != 0 opens the file for appending (the wrapper keeps one stream open over the run). calib_inv = {fxi, fyi, cxi, cyi}. */
int nalo_io_write_pcd_points(const char* path, int append, int n, const float* u, const float* v, const float* idepth, const float calib_inv[4],
                             const double camToWorld[12]);

/* camera.txt -- Undistort::getUndistorterForFile (util/Undistort.cpp:266-370) + Undistort::readFromFile (:765-933): line 1 `[Pinhole |RadTan |FOV |EquiDistant |
 * KannalaBrandt ]` + 5 (FOV, Pinhole) or 8 parameters; the prefix-less legacy forms are 8 values = RadTan, 5 values = FOV, or Pinhole when the 5th is 0; line 2 `wOrg hOrg`, line 3 `crop` | `full` | `none` | five floats,
 * line 4 `w h`. Relative calibrations (cx < 1 and cy < 1) are rescaled: fx*wOrg, fy*hOrg, cx*wOrg - 0.5, cy*hOrg - 0.5. */
enum { NALO_CAM_PINHOLE = 0, NALO_CAM_RADTAN = 1, NALO_CAM_FOV = 2, NALO_CAM_EQUIDISTANT = 3, NALO_CAM_KANNALABRANDT = 4 };
typedef struct {
    int model, n_pars;                 /* 5 or 8 */
    double pars[8];                    /* parsOrg after the relative rescale */
    int w_org, h_org, w, h;
    int rect_mode;                     /* -1 crop, -2 full, -3 none, 0 = explicit output calibration in out_calib */
    float out_calib[5];
} nalo_camera_file;
int nalo_io_read_camera(const char* path, nalo_camera_file* out);

/* pcalib.txt -- PhotometricUndistorter (util/Undistort.cpp:68-110): first line, >= 256 strictly increasing floats, rescaled to 0..255 with the
 * reference's expression 255.0 * (G[i] - min) / (max - min). G: capacity cap floats; *n = GDepth. */
int nalo_io_read_pcalib(const char* path, int cap, float* G, int* n);

/* times.txt -- ImageFolderReader::loadTimestamps (util/DatasetReader.h:317-380): lines `id stamp [exposure]`; a zero exposure is replaced by the mean of
 * its positive neighbours; if the line count differs from n_images both lists are dropped (*n_stamps = *n_exposures = 0); exposures are dropped when
 * any stays zero. cap = capacity of both arrays. */
int nalo_io_read_times(const char* path, int n_images, int cap, double* stamps, float* exposures, int* n_stamps, int* n_exposures);

/* Rectification -- the part of Undistort::readFromFile that follows the parse (util/Undistort.cpp:911-1005) with makeOptimalK_crop (:637-757) and the five
 * distortCoordinates models (:1018-1292: FOV, RadTan, EquiDistant, KannalaBrandt, Pinhole): from a parsed camera.txt to the rectified camera matrix
 * K_out = {fx, fy, cx, cy} and the per-pixel lookup remapX / remapY [w*h] (position in the original image, -1 = outside) that Undistort::undistort and
 * nalo_undist_set consume. rect_mode crop: the largest axis-aligned normalised rectangle whose border stays inside the original image (the reference's
 * 0.995 shrink iteration, <= 500 rounds); explicit: K = out_calib scaled by w, h (-0.5 on the centre); none: K = parsOrg, *passthrough = 1 (no table is needed;
 * remap is still filled). full (makeOptimalK_full) is an assert(false) in the reference: NALO_IO_ERR_FORMAT. Of the reference's two slips in the border
 * clean-up (:979-982) `if(iy == hOrg-1) ix = hOrg-1.001` is kept; the `iy < wOrg-1` test is kept AND completed by `iy < hOrg-1`: on a landscape sensor the reference
 * leaves entries with hOrg-1 <= iy < wOrg-1 "valid" and Undistort::undistort reads behind the image for them (undefined there); here they are -1 (pixel value 0),
 * so that the table is always one nalo_undist_set accepts. */
int nalo_io_make_rectification(const nalo_camera_file* cam, double K_out[4], float* remapX, float* remapY, int* passthrough);

/* PNG images -- the reference reads them through cv::imread (IOWrapper/OpenCV/ImageRW_OpenCV.cpp:33-53, 88-175): a self-contained decoder on zlib's
 * inflate (non-interlaced files; grey / grey+alpha / RGB / RGBA / palette; 1-16 bits), returning what the reference's wrappers return:
 *   NALO_PNG_GRAY8     readImageBW_8U / readMask_8U  (cv::IMREAD_GRAYSCALE): 8-bit grey; 16-bit samples keep their high byte (png_set_strip_16), colour goes
 *                      through libpng's png_set_rgb_to_gray(1, 0.299, 0.587) integer weights (9798 R + 19235 G + 3735 B + 16384) >> 15, alpha is dropped
 *   NALO_PNG_BGR8      readImageRGB_8U               (cv::IMREAD_COLOR): 8-bit B,G,R triplets (grey is replicated)
 *   NALO_PNG_UNCHANGED readImageBW_16U               (cv::IMREAD_UNCHANGED): samples as stored, 16-bit in host byte order; the wrapper accepts the file
 *                      only if it is single-channel 16-bit -- check *channels == 1 && *depth == 16 as it does (:160-164)
 * data: malloc'ed, w*h*channels samples of depth/8 bytes; free with nalo_io_free. */
enum { NALO_PNG_GRAY8 = 0, NALO_PNG_BGR8 = 1, NALO_PNG_UNCHANGED = 2 };
int nalo_io_read_png(const char* path, int mode, int* w, int* h, int* channels, int* depth, void** data);
void nalo_io_free(void* p);

/* vignette.png -- PhotometricUndistorter (util/Undistort.cpp:119-174): vignetteMap[i] = v[i] / max(v) (float division by the float maximum) from the
 * 16-bit image when readImageBW_16U accepts the file, else from the 8-bit one; vignetteMapInv[i] = 1.0f / vignetteMap[i]. px: n samples of `depth` (8|16) bits. */
int nalo_io_make_vignette(const void* px, int depth, int n, float* vignetteMap, float* vignetteMapInv);

/* masks / colour images -- IOWrap::resizeMask / resizeColor (IOWrapper/OpenCV/ImageRW_OpenCV.cpp:55-85): cv::resize(.., INTER_NEAREST) of an 8-bit image with
 * `channels` interleaved channels: dst(x, y) = src(min(floor(x * ifx), wOrg-1), min(floor(y * ify), hOrg-1)), ifx = 1 / ((double)w / wOrg) (OpenCV's resizeNN). */
int nalo_io_resize_nearest_u8(const uint8_t* src, int wOrg, int hOrg, int channels, uint8_t* dst, int w, int h);

#ifdef __cplusplus
}
#endif
#endif
