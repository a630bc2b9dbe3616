// On-disk formats: host-only text I/O behind the C ABI of include/nalo_io.h. Reference paths relative to src/.
#include "../../include/nalo_io.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>
#include <cmath>
#include <cstdlib>
#include <zlib.h>

namespace {

// Eigen's operator<< for a 1x3 row vector with the default IOFormat (Core/IO.h print_matrix): every coefficient is formatted with the stream's
// precision, the common width is the longest of them, each is written with that width, separated by one space.
void put_row3(std::ostream& os, const double* v) {
    size_t width = 0;
    for (int i = 0; i < 3; ++i) {
        std::ostringstream s;
        s.copyfmt(os);
        s << v[i];
        width = std::max(width, s.str().size());
    }
    for (int i = 0; i < 3; ++i) {
        if (i) os << " ";
        os.width((std::streamsize)width);
        os << v[i];
    }
}

}  // namespace

extern "C" {

int nalo_io_write_result(const char* path, int n, const double* timestamp, const uint8_t* poseValid, const double* t, const double* q) {
    if (!path || n < 0 || (n > 0 && (!timestamp || !poseValid || !t || !q))) return NALO_IO_ERR_ARG;
    std::ofstream f(path);
    if (!f.good()) return NALO_IO_ERR_FILE;
    f << std::setprecision(15);
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return timestamp[a] < timestamp[b]; });        // FrameSort (FullSystem.cpp:267)
    for (int i = 0; i < n; ++i) {
        const int s = order[i];
        if (!poseValid[s] && i == 0) { f << timestamp[s] << " " << 0 << " " << 0 << " " << 0 << " " << 0 << " " << 0 << " " << 0 << " " << 0 << "\n"; continue; }
        const int p = poseValid[s] ? s : order[i - 1];          // invalid: the pose of the previous frame in the sorted history (:474-480)
        f << timestamp[s] << " ";
        put_row3(f, t + 3 * p);
        f << " " << q[4 * p] << " " << q[4 * p + 1] << " " << q[4 * p + 2] << " " << q[4 * p + 3] << "\n";
    }
    f.close();
    return f.fail() ? NALO_IO_ERR_FILE : NALO_IO_OK;
}

int nalo_io_write_pcd_points(const char* path, int append, int n, const float* u, const float* v, const float* idepth, const float calib_inv[4],
                             const double m[12]) {
    if (!path || n < 0 || !calib_inv || !m || (n > 0 && (!u || !v || !idepth))) return NALO_IO_ERR_ARG;
    std::ofstream f(path, append ? std::ios::app : std::ios::trunc);
    if (!f.good()) return NALO_IO_ERR_FILE;
    const float fxi = calib_inv[0], fyi = calib_inv[1], cxi = calib_inv[2], cyi = calib_inv[3];
    for (int i = 0; i < n; ++i) {
        const float depth = 1.0f / idepth[i];
        const float x = (u[i] * fxi + cxi) * depth, y = (v[i] * fyi + cyi) * depth, z = depth * (1 + 2 * fxi);       // SampleOutputWrapper.h:113-116
        const double c[4] = {x, y, z, 1.0};
        double wp[3];
        for (int r = 0; r < 3; ++r) wp[r] = ((m[4 * r] * c[0] + m[4 * r + 1] * c[1]) + m[4 * r + 2] * c[2]) + m[4 * r + 3] * c[3];
        f << wp[0] << " " << wp[1] << " " << wp[2] << "\n";
    }
    f.close();
    return f.fail() ? NALO_IO_ERR_FILE : NALO_IO_OK;
}

int nalo_io_read_camera(const char* path, nalo_camera_file* out) {
    if (!path || !out) return NALO_IO_ERR_ARG;
    std::ifstream in(path);
    if (!in.good()) return NALO_IO_ERR_FILE;
    std::string l1, l2, l3, l4;
    std::getline(in, l1); std::getline(in, l2); std::getline(in, l3); std::getline(in, l4);
    std::memset(out, 0, sizeof(*out));
    // getUndistorterForFile (Undistort.cpp:266-370): the prefix picks the model; the prefix-less legacy forms are RadTan (8 values) or FOV / pinhole (5)
    static const struct { const char* prefix; int model, npars; } kForms[] = {
        {"RadTan ", NALO_CAM_RADTAN, 8}, {"EquiDistant ", NALO_CAM_EQUIDISTANT, 8}, {"KannalaBrandt ", NALO_CAM_KANNALABRANDT, 8},
        {"FOV ", NALO_CAM_FOV, 5},       {"Pinhole ", NALO_CAM_PINHOLE, 5},         {"", NALO_CAM_RADTAN, 8},                     {"", NALO_CAM_FOV, 5}};
    bool ok = false;
    for (const auto& fm : kForms) {
        const size_t pl = std::strlen(fm.prefix);
        if (l1.compare(0, pl, fm.prefix) != 0) continue;
        double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int got = std::sscanf(l1.c_str() + pl, "%lf %lf %lf %lf %lf %lf %lf %lf", &p[0], &p[1], &p[2], &p[3], &p[4], &p[5], &p[6], &p[7]);
        if (got < fm.npars) continue;                           // (prefix-less line with 5 values: sscanf stops at 5, the 8-value form is skipped)
        out->model = fm.model; out->n_pars = fm.npars;
        if (pl == 0 && fm.npars == 5 && p[4] == 0) out->model = NALO_CAM_PINHOLE;      // legacy 5-value form: omega == 0 is a pinhole (Undistort.cpp:299-313)
        for (int i = 0; i < fm.npars; ++i) out->pars[i] = p[i];
        ok = true;
        break;
    }
    if (!ok || std::sscanf(l2.c_str(), "%d %d", &out->w_org, &out->h_org) != 2) return NALO_IO_ERR_FORMAT;
    if (out->pars[2] < 1 && out->pars[3] < 1) {                 // relative calibration (:838-855)
        out->pars[0] = out->pars[0] * out->w_org; out->pars[1] = out->pars[1] * out->h_org;
        out->pars[2] = out->pars[2] * out->w_org - 0.5; out->pars[3] = out->pars[3] * out->h_org - 0.5;
    }
    if (l3 == "crop") out->rect_mode = -1;
    else if (l3 == "full") out->rect_mode = -2;
    else if (l3 == "none") out->rect_mode = -3;
    else if (std::sscanf(l3.c_str(), "%f %f %f %f %f", &out->out_calib[0], &out->out_calib[1], &out->out_calib[2], &out->out_calib[3], &out->out_calib[4]) == 5) out->rect_mode = 0;
    else return NALO_IO_ERR_FORMAT;
    if (std::sscanf(l4.c_str(), "%d %d", &out->w, &out->h) != 2) return NALO_IO_ERR_FORMAT;
    return NALO_IO_OK;
}

int nalo_io_read_pcalib(const char* path, int cap, float* G, int* n) {
    if (!path || !G || !n || cap < 256) return NALO_IO_ERR_ARG;
    std::ifstream f(path);
    if (!f.good()) return NALO_IO_ERR_FILE;
    std::string line;
    std::getline(f, line);
    std::istringstream l1i(line);
    const std::vector<float> Gvec((std::istream_iterator<float>(l1i)), std::istream_iterator<float>());
    const int depth = (int)Gvec.size();
    if (depth < 256 || depth > cap) return NALO_IO_ERR_FORMAT;
    for (int i = 0; i < depth - 1; ++i) if (Gvec[i + 1] <= Gvec[i]) return NALO_IO_ERR_FORMAT;     // strictly increasing (:98-105)
    const float mn = Gvec[0], mx = Gvec[depth - 1];
    for (int i = 0; i < depth; ++i) G[i] = 255.0 * (Gvec[i] - mn) / (mx - mn);                      // :109
    *n = depth;
    return NALO_IO_OK;
}

int nalo_io_read_times(const char* path, int n_images, int cap, double* stamps, float* exposures, int* n_stamps, int* n_exposures) {
    if (!path || !stamps || !exposures || !n_stamps || !n_exposures || cap < 0) return NALO_IO_ERR_ARG;
    std::ifstream tr(path);
    if (!tr.good()) return NALO_IO_ERR_FILE;
    std::vector<double> ts; std::vector<float> ex;
    while (!tr.eof() && tr.good()) {
        char buf[1000];
        tr.getline(buf, 1000);
        int id; double stamp; float exposure = 0;
        if (3 == std::sscanf(buf, "%d %lf %f", &id, &stamp, &exposure)) { ts.push_back(stamp); ex.push_back(exposure); }
        else if (2 == std::sscanf(buf, "%d %lf", &id, &stamp)) { ts.push_back(stamp); ex.push_back(exposure); }
    }
    bool good = (int)ex.size() == n_images;
    for (int i = 0; i < (int)ex.size(); ++i) {
        if (ex[i] == 0) {
            float sum = 0, num = 0;
            if (i > 0 && ex[i - 1] > 0) { sum += ex[i - 1]; num++; }
            if (i + 1 < (int)ex.size() && ex[i + 1] > 0) { sum += ex[i + 1]; num++; }
            if (num > 0) ex[i] = sum / num;
        }
        if (ex[i] == 0) good = false;
    }
    if (n_images != (int)ts.size()) { ex.clear(); ts.clear(); }
    if (n_images != (int)ex.size() || !good) ex.clear();
    if ((int)ts.size() > cap || (int)ex.size() > cap) return NALO_IO_ERR_ARG;
    std::copy(ts.begin(), ts.end(), stamps); std::copy(ex.begin(), ex.end(), exposures);
    *n_stamps = (int)ts.size(); *n_exposures = (int)ex.size();
    return NALO_IO_OK;
}

// ------------------------------------------------------------------------------------------------ PNG
namespace {
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
}

static int read_png_impl(const char* path, int mode, int* w_out, int* h_out, int* channels_out, int* depth_out, void** data_out);
int nalo_io_read_png(const char* path, int mode, int* w_out, int* h_out, int* channels_out, int* depth_out, void** data_out) {
    if (!path || !w_out || !h_out || !channels_out || !depth_out || !data_out || mode < 0 || mode > 2) return NALO_IO_ERR_ARG;
    try { return read_png_impl(path, mode, w_out, h_out, channels_out, depth_out, data_out); }
    catch (...) { return NALO_IO_ERR_FORMAT; }                                                   // no exception crosses the C ABI (a damaged header can ask for any size)
}
static int read_png_impl(const char* path, int mode, int* w_out, int* h_out, int* channels_out, int* depth_out, void** data_out) {
    std::ifstream f(path, std::ios::binary);
    if (!f.good()) return NALO_IO_ERR_FILE;
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8)) return NALO_IO_ERR_FORMAT;
    int w = 0, h = 0, bitdepth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    size_t pos = 8; bool have_hdr = false, done = false;
    while (!done && pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]); const uint8_t* type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) return NALO_IO_ERR_FORMAT;
        const uint8_t* d = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return NALO_IO_ERR_FORMAT;
            w = (int)be32(d); h = (int)be32(d + 4); bitdepth = d[8]; ctype = d[9]; interlace = d[12]; have_hdr = true;
            if (d[10] != 0 || d[11] != 0) return NALO_IO_ERR_FORMAT;
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!std::memcmp(type, "IEND", 4)) done = true;
        pos += 12 + (size_t)len;
    }
    if (!have_hdr || w <= 0 || h <= 0 || interlace != 0) return NALO_IO_ERR_FORMAT;             // Adam7 files are not supported
    int nch;
    switch (ctype) { case 0: nch = 1; break; case 2: nch = 3; break; case 3: nch = 1; break; case 4: nch = 2; break; case 6: nch = 4; break; default: return NALO_IO_ERR_FORMAT; }
    if (!(bitdepth == 8 || bitdepth == 16 || ((ctype == 0 || ctype == 3) && (bitdepth == 1 || bitdepth == 2 || bitdepth == 4))) || (ctype == 3 && bitdepth == 16)) return NALO_IO_ERR_FORMAT;
    if (ctype == 3 && plte.size() < 3) return NALO_IO_ERR_FORMAT;
    const size_t bpp_bits = (size_t)nch * bitdepth, stride = ((size_t)w * bpp_bits + 7) / 8, bpp = std::max<size_t>(1, bpp_bits / 8);
    // a damaged IHDR must not turn into a huge allocation: deflate cannot expand beyond ~1032:1, and frames here are far below 2^15 pixels a side
    if (w > 32768 || h > 32768 || (stride + 1) * (size_t)h > idat.size() * 1100 + 4096) return NALO_IO_ERR_FORMAT;
    std::vector<uint8_t> rawbuf((stride + 1) * (size_t)h);
    {
        z_stream zs; std::memset(&zs, 0, sizeof(zs));
        if (inflateInit(&zs) != Z_OK) return NALO_IO_ERR_FORMAT;
        zs.next_in = idat.data(); zs.avail_in = (uInt)idat.size(); zs.next_out = rawbuf.data(); zs.avail_out = (uInt)rawbuf.size();
        const int zr = inflate(&zs, Z_FINISH);
        const size_t got = rawbuf.size() - zs.avail_out;
        inflateEnd(&zs);
        if ((zr != Z_STREAM_END && zr != Z_OK && zr != Z_BUF_ERROR) || got != rawbuf.size()) return NALO_IO_ERR_FORMAT;
    }
    // un-filter (PNG specification, 9.2)
    std::vector<uint8_t> img(stride * (size_t)h);
    for (int y = 0; y < h; ++y) {
        const uint8_t ft = rawbuf[(stride + 1) * y]; const uint8_t* in = &rawbuf[(stride + 1) * y + 1];
        uint8_t* cur = &img[stride * y]; const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = in[i];
            switch (ft) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break; default: return NALO_IO_ERR_FORMAT; }
            cur[i] = (uint8_t)v;
        }
    }
    // samples -> 16-bit working values per channel (palette and sub-byte depths expanded the way libpng's png_set_expand does)
    const size_t npx = (size_t)w * h;
    int och = nch; const int odepth = bitdepth == 16 ? 16 : 8;
    std::vector<uint16_t> px;
    if (ctype == 3) {
        och = 3; px.resize(npx * 3);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
            const uint8_t* row = &img[stride * y]; unsigned idx;
            if (bitdepth == 8) idx = row[x]; else { const int per = 8 / bitdepth, sh = (per - 1 - x % per) * bitdepth; idx = (row[x / per] >> sh) & ((1u << bitdepth) - 1); }
            if ((size_t)idx * 3 + 2 >= plte.size()) return NALO_IO_ERR_FORMAT;
            for (int k = 0; k < 3; ++k) px[((size_t)y * w + x) * 3 + k] = plte[idx * 3 + k];
        }
    } else {
        px.resize(npx * nch);
        for (int y = 0; y < h; ++y) {
            const uint8_t* row = &img[stride * y];
            for (int x = 0; x < w; ++x) for (int k = 0; k < nch; ++k) {
                uint16_t v;
                if (bitdepth == 16) v = (uint16_t)((row[((size_t)x * nch + k) * 2] << 8) | row[((size_t)x * nch + k) * 2 + 1]);
                else if (bitdepth == 8) v = row[(size_t)x * nch + k];
                else { const int per = 8 / bitdepth, sh = (per - 1 - x % per) * bitdepth; const unsigned s = (row[x / per] >> sh) & ((1u << bitdepth) - 1); v = (uint16_t)(s * 255u / ((1u << bitdepth) - 1)); }
                px[((size_t)y * w + x) * nch + k] = v;
            }
        }
    }
    void* out = nullptr; int rch, rdepth;
    if (mode == NALO_PNG_UNCHANGED) {
        rch = och; rdepth = odepth;
        if (rdepth == 16) { uint16_t* o = (uint16_t*)std::malloc(npx * rch * 2); if (!o) return NALO_IO_ERR_FILE; std::memcpy(o, px.data(), npx * rch * 2); out = o; }
        else { uint8_t* o = (uint8_t*)std::malloc(npx * rch); if (!o) return NALO_IO_ERR_FILE; for (size_t i = 0; i < npx * rch; ++i) o[i] = (uint8_t)px[i]; out = o; }
        if (rdepth == 8 && rch >= 3) { uint8_t* o = (uint8_t*)out; for (size_t i = 0; i < npx; ++i) std::swap(o[i * rch], o[i * rch + 2]); }      // OpenCV stores B,G,R(,A)
        if (rdepth == 16 && rch >= 3) { uint16_t* o = (uint16_t*)out; for (size_t i = 0; i < npx; ++i) std::swap(o[i * rch], o[i * rch + 2]); }
    } else {
        const int sh = odepth == 16 ? 8 : 0;                                                       // png_set_strip_16: the high byte
        const bool colour = och >= 3;
        rdepth = 8; rch = mode == NALO_PNG_GRAY8 ? 1 : 3;
        uint8_t* o = (uint8_t*)std::malloc(npx * rch); if (!o) return NALO_IO_ERR_FILE;
        for (size_t i = 0; i < npx; ++i) {
            const uint16_t* p = &px[i * och];
            if (mode == NALO_PNG_GRAY8) {
                if (colour) { const unsigned r = p[0] >> sh, g = p[1] >> sh, b = p[2] >> sh; o[i] = (uint8_t)((9798u * r + 19235u * g + 3735u * b + 16384u) >> 15); }
                else o[i] = (uint8_t)(p[0] >> sh);
            } else {
                if (colour) { o[3 * i] = (uint8_t)(p[2] >> sh); o[3 * i + 1] = (uint8_t)(p[1] >> sh); o[3 * i + 2] = (uint8_t)(p[0] >> sh); }
                else o[3 * i] = o[3 * i + 1] = o[3 * i + 2] = (uint8_t)(p[0] >> sh);
            }
        }
        out = o;
    }
    *w_out = w; *h_out = h; *channels_out = rch; *depth_out = rdepth; *data_out = out;
    return NALO_IO_OK;
}
void nalo_io_free(void* p) { std::free(p); }

int nalo_io_make_vignette(const void* px, int depth, int n, float* vignetteMap, float* vignetteMapInv) {
    if (!px || (depth != 8 && depth != 16) || n <= 0 || !vignetteMap || !vignetteMapInv) return NALO_IO_ERR_ARG;
    float maxV = 0;                                                                              // Undistort.cpp:137-142 / 156-161
    for (int i = 0; i < n; ++i) { const float v = depth == 16 ? (float)((const uint16_t*)px)[i] : (float)((const uint8_t*)px)[i]; if (v > maxV) maxV = v; }
    for (int i = 0; i < n; ++i) { const float v = depth == 16 ? (float)((const uint16_t*)px)[i] : (float)((const uint8_t*)px)[i]; vignetteMap[i] = v / maxV; }
    for (int i = 0; i < n; ++i) vignetteMapInv[i] = 1.0f / vignetteMap[i];                       // :176-177
    return NALO_IO_OK;
}

int nalo_io_resize_nearest_u8(const uint8_t* src, int wOrg, int hOrg, int channels, uint8_t* dst, int w, int h) {
    if (!src || !dst || wOrg <= 0 || hOrg <= 0 || w <= 0 || h <= 0 || channels <= 0) return NALO_IO_ERR_ARG;
    const double ifx = 1.0 / ((double)w / wOrg), ify = 1.0 / ((double)h / hOrg);
    for (int y = 0; y < h; ++y) {
        const int sy = std::min((int)std::floor(y * ify), hOrg - 1);
        for (int x = 0; x < w; ++x) {
            const int sx = std::min((int)std::floor(x * ifx), wOrg - 1);
            for (int k = 0; k < channels; ++k) dst[((size_t)y * w + x) * channels + k] = src[((size_t)sy * wOrg + sx) * channels + k];
        }
    }
    return NALO_IO_OK;
}

// ------------------------------------------------------------------------------------------------ rectification tables
// The per-frame work of the rectification (the remap itself) runs on the device (ingest_kernel); what is built here, once per camera, is its table: for every
// pixel of the rectified pinhole image the sensor position it is sampled from. The structure is our own; the ARITHMETIC per point is the reference's, in its
// float / double mix and association order, because the table decides which sensor texels every later stage reads (tests/test_io_cpu.py holds it to the oracle's
// restatement bit for bit, and to closed forms that involve no restatement at all):
//   * a lens model = one function "normalised pinhole point -> sensor pixel" out of a table indexed by nalo_camera_model (util/Undistort.cpp:1018-1283),
//   * View = a centred pinhole view in normalised coordinates; edges_off_sensor() probes its four border lines through the lens,
//   * the `crop` mode (makeOptimalK_crop, :637-757) = per-axis extent of the sensor along the centre lines, a 1 % head start, then shrinking the View by 0.5 % per
//     round on the sides that leave the sensor (the longer axis first when both do) until all four borders land on it.
namespace {
struct LensPars { float fx, fy, cx, cy, k[4]; };                       // parsOrg as floats (every model of the reference casts its doubles first)
typedef void (*LensFn)(const LensPars& L, float ix, float iy, float& sx, float& sy);
// unsuffixed literals are doubles ON PURPOSE: the reference's expressions promote there
void lens_pinhole(const LensPars& L, float ix, float iy, float& sx, float& sy) { sx = L.fx * ix + L.cx; sy = L.fy * iy + L.cy; }
void lens_fov(const LensPars& L, float ix, float iy, float& sx, float& sy) {
    const float dist = L.k[0], d2t = 2.0f * tan(dist / 2.0f);
    const float r = sqrtf(ix * ix + iy * iy);
    const float fac = (r == 0 || dist == 0) ? 1 : atanf(r * d2t) / (dist * r);
    sx = L.fx * fac * ix + L.cx; sy = L.fy * fac * iy + L.cy;
}
void lens_radtan(const LensPars& L, float ix, float iy, float& sx, float& sy) {
    const float k1 = L.k[0], k2 = L.k[1], r1 = L.k[2], r2 = L.k[3];
    const float mx2 = ix * ix, my2 = iy * iy, mxy = ix * iy, rho2 = mx2 + my2;
    const float rad = k1 * rho2 + k2 * rho2 * rho2;
    const float xd = ix + ix * rad + 2.0 * r1 * mxy + r2 * (rho2 + 2.0 * mx2);
    const float yd = iy + iy * rad + 2.0 * r2 * mxy + r1 * (rho2 + 2.0 * my2);
    sx = L.fx * xd + L.cx; sy = L.fy * yd + L.cy;
}
void lens_equidistant(const LensPars& L, float ix, float iy, float& sx, float& sy) {
    const float r = sqrt(ix * ix + iy * iy);                           // sqrt / atan: double functions of float arguments
    const float theta = atan(r), t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const float thetad = theta * (1 + L.k[0] * t2 + L.k[1] * t4 + L.k[2] * t6 + L.k[3] * t8);
    const float scaling = (r > 1e-8) ? thetad / r : 1.0;
    sx = L.fx * ix * scaling + L.cx; sy = L.fy * iy * scaling + L.cy;
}
void lens_kannala_brandt(const LensPars& L, float ix, float iy, float& sx, float& sy) {
    const float sq = sqrtf(ix * ix + iy * iy);
    const float theta = atan2f(sq, 1), t2 = theta * theta, t3 = t2 * theta, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
    const float r = theta + L.k[0] * t3 + L.k[1] * t5 + L.k[2] * t7 + L.k[3] * t9;
    if (sq < 1e-6) { sx = L.fx * ix + L.cx; sy = L.fy * iy + L.cy; }
    else { sx = (r / sq) * L.fx * ix + L.cx; sy = (r / sq) * L.fy * iy + L.cy; }
}
LensFn lens_of(int model) {
    static const struct { int id; LensFn fn; } table[] = {{NALO_CAM_PINHOLE, lens_pinhole}, {NALO_CAM_RADTAN, lens_radtan}, {NALO_CAM_FOV, lens_fov},
                                                          {NALO_CAM_EQUIDISTANT, lens_equidistant}, {NALO_CAM_KANNALABRANDT, lens_kannala_brandt}};
    for (const auto& e : table) if (e.id == model) return e.fn;
    return lens_pinhole;
}
struct Lens {
    LensPars L; LensFn fn;
    explicit Lens(const nalo_camera_file& cam) : fn(lens_of(cam.model)) {
        L.fx = (float)cam.pars[0]; L.fy = (float)cam.pars[1]; L.cx = (float)cam.pars[2]; L.cy = (float)cam.pars[3];
        for (int i = 0; i < 4; ++i) L.k[i] = (float)cam.pars[4 + i];
    }
    // n pixels of the pinhole view K = {fx, fy, cx, cy} -> sensor pixels, in place
    void to_sensor(const double K[4], float* px, float* py, int n) const {
        const float ofx = (float)K[0], ofy = (float)K[1], ocx = (float)K[2], ocy = (float)K[3];
        for (int i = 0; i < n; ++i) fn(L, (px[i] - ocx) / ofx, (py[i] - ocy) / ofy, px[i], py[i]);
    }
};
const double kUnitView[4] = {1, 1, 0, 0};                               // K = identity: view pixels ARE normalised coordinates
struct View { float lo[2], hi[2]; };                                    // [0] = x, [1] = y, normalised coordinates
enum : unsigned { kOffLeft = 1, kOffRight = 2, kOffTop = 4, kOffBottom = 8 };

// the part of the centre line of one axis that lands strictly inside the sensor: first and last of 100 000 samples over [-5, 5) (the first sample is taken while
// the bound still reads 0, as in the reference)
void centre_line_extent(const Lens& lens, int axis, int sensor_size, float& lo, float& hi) {
    const int N = 100000;
    std::vector<float> a(N), b(N, 0.f);
    for (int i = 0; i < N; ++i) a[i] = (i - 50000.0f) / 10000.0f;
    if (axis == 0) lens.to_sensor(kUnitView, a.data(), b.data(), N); else lens.to_sensor(kUnitView, b.data(), a.data(), N);
    lo = hi = 0;
    for (int i = 0; i < N; ++i) if (a[i] > 0 && a[i] < sensor_size - 1) { const float t = (i - 50000.0f) / 10000.0f; if (lo == 0) lo = t; hi = t; }
}
// which borders of the view (sampled at the h rows / w columns of the rectified image) leave the sensor; sx, sy: scratch of 2 max(w, h) floats
unsigned edges_off_sensor(const Lens& lens, const View& v, int w, int h, int wOrg, int hOrg, float* sx, float* sy) {
    unsigned off = 0;
    for (int y = 0; y < h; ++y) { sx[2 * y] = v.lo[0]; sx[2 * y + 1] = v.hi[0]; sy[2 * y] = sy[2 * y + 1] = v.lo[1] + (v.hi[1] - v.lo[1]) * (float)y / ((float)h - 1.0f); }
    lens.to_sensor(kUnitView, sx, sy, 2 * h);
    for (int y = 0; y < h; ++y) {
        if (!(sx[2 * y] > 0 && sx[2 * y] < wOrg - 1)) off |= kOffLeft;
        if (!(sx[2 * y + 1] > 0 && sx[2 * y + 1] < wOrg - 1)) off |= kOffRight;
    }
    for (int x = 0; x < w; ++x) { sy[2 * x] = v.lo[1]; sy[2 * x + 1] = v.hi[1]; sx[2 * x] = sx[2 * x + 1] = v.lo[0] + (v.hi[0] - v.lo[0]) * (float)x / ((float)w - 1.0f); }
    lens.to_sensor(kUnitView, sx, sy, 2 * w);
    for (int x = 0; x < w; ++x) {
        if (!(sy[2 * x] > 0 && sy[2 * x] < hOrg - 1)) off |= kOffTop;
        if (!(sy[2 * x + 1] > 0 && sy[2 * x + 1] < hOrg - 1)) off |= kOffBottom;
    }
    return off;
}
// the largest centred view all of whose borders land on the sensor -> K of the rectified image; false = no such view after 500 rounds (the reference exits)
bool crop_view(const Lens& lens, int w, int h, int wOrg, int hOrg, double K[4], float* sx, float* sy) {
    View v;
    centre_line_extent(lens, 0, wOrg, v.lo[0], v.hi[0]);
    centre_line_extent(lens, 1, hOrg, v.lo[1], v.hi[1]);
    for (int a = 0; a < 2; ++a) { v.lo[a] *= 1.01; v.hi[a] *= 1.01; }      // head start: the shrink loop below comes back from outside
    for (int round = 0;; ++round) {
        unsigned off = edges_off_sensor(lens, v, w, h, wOrg, hOrg, sx, sy);
        if (!off) break;
        if ((off & (kOffLeft | kOffRight)) && (off & (kOffTop | kOffBottom)))  // both axes overflow: only the longer one shrinks this round
            off &= (v.hi[0] - v.lo[0]) > (v.hi[1] - v.lo[1]) ? (kOffLeft | kOffRight) : (kOffTop | kOffBottom);
        if (off & kOffLeft) v.lo[0] *= 0.995;
        if (off & kOffRight) v.hi[0] *= 0.995;
        if (off & kOffTop) v.lo[1] *= 0.995;
        if (off & kOffBottom) v.hi[1] *= 0.995;
        if (round >= 500) return false;
    }
    K[0] = ((float)w - 1.0f) / (v.hi[0] - v.lo[0]); K[1] = ((float)h - 1.0f) / (v.hi[1] - v.lo[1]);
    K[2] = -v.lo[0] * K[0]; K[3] = -v.lo[1] * K[1];
    return true;
}
}  // namespace

int nalo_io_make_rectification(const nalo_camera_file* cam, double K_out[4], float* remapX, float* remapY, int* passthrough) {
    if (!cam || !K_out || !remapX || !remapY || cam->w <= 1 || cam->h <= 1 || cam->w_org <= 1 || cam->h_org <= 1) return NALO_IO_ERR_ARG;
    const Lens lens(*cam);
    const int w = cam->w, h = cam->h, wOrg = cam->w_org, hOrg = cam->h_org;
    double K[4];
    int pass = 0;
    switch (cam->rect_mode) {
    case -1: if (!crop_view(lens, w, h, wOrg, hOrg, K, remapX, remapY)) return NALO_IO_ERR_FORMAT; break;       // the table buffers double as the probe's scratch (w h >= 2 max(w, h))
    case -2: return NALO_IO_ERR_FORMAT;                                                                        // makeOptimalK_full: assert(false) in the reference
    case -3:
        if (w != wOrg || h != hOrg) return NALO_IO_ERR_FORMAT;
        for (int i = 0; i < 4; ++i) K[i] = cam->pars[i];
        pass = 1;
        break;
    default: K[0] = cam->out_calib[0] * w; K[1] = cam->out_calib[1] * h; K[2] = cam->out_calib[2] * w - 0.5; K[3] = cam->out_calib[3] * h - 0.5;
    }
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) { remapX[x + y * w] = x; remapY[x + y * w] = y; }
    lens.to_sensor(K, remapX, remapY, h * w);
    for (size_t i = 0, n = (size_t)w * h; i < n; ++i) {                                           // "make rounding resistant" (:972-996), the `ix = hOrg-1.001` slip included
        float ix = remapX[i], iy = remapY[i];
        if (ix == 0) ix = 0.001;
        if (iy == 0) iy = 0.001;
        if (ix == wOrg - 1) ix = wOrg - 1.001;
        if (iy == hOrg - 1) ix = hOrg - 1.001;
        // DEVIATION from the reference: its test is `iy < wOrg-1` (util/Undistort.cpp:980). On a landscape sensor an entry with hOrg-1 <= iy < wOrg-1 stays
        // "valid" there and Undistort::undistort then reads rows behind the image (:497-512: an out-of-bounds read). Such entries are outside here (-1 -> pixel 0),
        // which is also what nalo_undist_set demands of a table; everything the reference defines is unchanged (tests/test_io_cpu.py, tests/test_ingest_gpu.py).
        const bool inside = ix > 0 && iy > 0 && ix < wOrg - 1 && iy < wOrg - 1 && iy < hOrg - 1;
        remapX[i] = inside ? ix : -1; remapY[i] = inside ? iy : -1;
    }
    for (int i = 0; i < 4; ++i) K_out[i] = K[i];
    if (passthrough) *passthrough = pass;
    return NALO_IO_OK;
}

}  // extern "C"
