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
namespace {
struct Rectifier {
    int model; float p[8]; double K[4];                       // parsOrg as floats (VecX parsOrg holds doubles; every model casts to float first), K = fx fy cx cy
    void distort(const float* in_x, const float* in_y, float* out_x, float* out_y, int n) const {
        const float fx = p[0], fy = p[1], cx = p[2], cy = p[3];
        const float ofx = (float)K[0], ofy = (float)K[1], ocx = (float)K[2], ocy = (float)K[3];
        for (int i = 0; i < n; ++i) {
            float ix = (in_x[i] - ocx) / ofx, iy = (in_y[i] - ocy) / ofy;
            switch (model) {
            case NALO_CAM_FOV: {                                                                 // Undistort.cpp:1018-1055
                const float dist = p[4], d2t = 2.0f * tan(dist / 2.0f);
                const float r = sqrtf(ix * ix + iy * iy);
                const float fac = (r == 0 || dist == 0) ? 1 : atanf(r * d2t) / (dist * r);
                out_x[i] = fx * fac * ix + cx; out_y[i] = fy * fac * iy + cy;
            } break;
            case NALO_CAM_RADTAN: {                                                              // :1074-1116 (2.0 literals: double intermediate)
                const float k1 = p[4], k2 = p[5], r1 = p[6], r2 = p[7];
                const float mx2 = ix * ix, my2 = iy * iy, mxy = ix * iy, rho2 = mx2 + my2;
                const float rad = k1 * rho2 + k2 * rho2 * rho2;
                const float xd = ix + ix * rad + 2.0 * r1 * mxy + r2 * (rho2 + 2.0 * mx2);
                const float yd = iy + iy * rad + 2.0 * r2 * mxy + r1 * (rho2 + 2.0 * my2);
                out_x[i] = fx * xd + cx; out_y[i] = fy * yd + cy;
            } break;
            case NALO_CAM_EQUIDISTANT: {                                                         // :1134-1176 (sqrt / atan: double functions of float arguments)
                const float k1 = p[4], k2 = p[5], k3 = p[6], k4 = p[7];
                const float r = sqrt(ix * ix + iy * iy);
                const float theta = atan(r), t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
                const float thetad = theta * (1 + k1 * t2 + k2 * t4 + k3 * t6 + k4 * t8);
                const float scaling = (r > 1e-8) ? thetad / r : 1.0;
                out_x[i] = fx * ix * scaling + cx; out_y[i] = fy * iy * scaling + cy;
            } break;
            case NALO_CAM_KANNALABRANDT: {                                                       // :1193-1240
                const float k0 = p[4], k1 = p[5], k2 = p[6], k3 = p[7];
                const float ss = ix * ix + iy * iy, sq = sqrtf(ss);
                const float theta = atan2f(sq, 1), t2 = theta * theta, t3 = t2 * theta, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
                const float r = theta + k0 * t3 + k1 * t5 + k2 * t7 + k3 * t9;
                if (sq < 1e-6) { out_x[i] = fx * ix + cx; out_y[i] = fy * iy + cy; }
                else { out_x[i] = (r / sq) * fx * ix + cx; out_y[i] = (r / sq) * fy * iy + cy; }
            } break;
            default:                                                                             // Pinhole :1258-1283
                out_x[i] = fx * ix + cx; out_y[i] = fy * iy + cy;
            }
        }
    }
};
}  // namespace

int nalo_io_make_rectification(const nalo_camera_file* cam, double K_out[4], float* remapX, float* remapY, int* passthrough) {
    if (!cam || !K_out || !remapX || !remapY || cam->w <= 1 || cam->h <= 1 || cam->w_org <= 1 || cam->h_org <= 1) return NALO_IO_ERR_ARG;
    Rectifier R; R.model = cam->model;
    for (int i = 0; i < 8; ++i) R.p[i] = (float)cam->pars[i];
    const int w = cam->w, h = cam->h, wOrg = cam->w_org, hOrg = cam->h_org;
    int pass = 0;
    if (cam->rect_mode == -1) {                                                                    // makeOptimalK_crop (:637-757)
        R.K[0] = R.K[1] = 1; R.K[2] = R.K[3] = 0;                                                  // K.setIdentity()
        std::vector<float> tgX(100000), tgY(100000);
        float minX = 0, maxX = 0, minY = 0, maxY = 0;
        for (int x = 0; x < 100000; ++x) { tgX[x] = (x - 50000.0f) / 10000.0f; tgY[x] = 0; }
        R.distort(tgX.data(), tgY.data(), tgX.data(), tgY.data(), 100000);
        for (int x = 0; x < 100000; ++x) if (tgX[x] > 0 && tgX[x] < wOrg - 1) { if (minX == 0) minX = (x - 50000.0f) / 10000.0f; maxX = (x - 50000.0f) / 10000.0f; }
        for (int y = 0; y < 100000; ++y) { tgY[y] = (y - 50000.0f) / 10000.0f; tgX[y] = 0; }
        R.distort(tgX.data(), tgY.data(), tgX.data(), tgY.data(), 100000);
        for (int y = 0; y < 100000; ++y) if (tgY[y] > 0 && tgY[y] < hOrg - 1) { if (minY == 0) minY = (y - 50000.0f) / 10000.0f; maxY = (y - 50000.0f) / 10000.0f; }
        minX *= 1.01; maxX *= 1.01; minY *= 1.01; maxY *= 1.01;
        bool oobLeft = true, oobRight = true, oobTop = true, oobBottom = true;
        int iteration = 0;
        while (oobLeft || oobRight || oobTop || oobBottom) {
            oobLeft = oobRight = oobTop = oobBottom = false;
            for (int y = 0; y < h; ++y) { remapX[y * 2] = minX; remapX[y * 2 + 1] = maxX; remapY[y * 2] = remapY[y * 2 + 1] = minY + (maxY - minY) * (float)y / ((float)h - 1.0f); }
            R.distort(remapX, remapY, remapX, remapY, 2 * h);
            for (int y = 0; y < h; ++y) {
                if (!(remapX[2 * y] > 0 && remapX[2 * y] < wOrg - 1)) oobLeft = true;
                if (!(remapX[2 * y + 1] > 0 && remapX[2 * y + 1] < wOrg - 1)) oobRight = true;
            }
            for (int x = 0; x < w; ++x) { remapY[x * 2] = minY; remapY[x * 2 + 1] = maxY; remapX[x * 2] = remapX[x * 2 + 1] = minX + (maxX - minX) * (float)x / ((float)w - 1.0f); }
            R.distort(remapX, remapY, remapX, remapY, 2 * w);
            for (int x = 0; x < w; ++x) {
                if (!(remapY[2 * x] > 0 && remapY[2 * x] < hOrg - 1)) oobTop = true;
                if (!(remapY[2 * x + 1] > 0 && remapY[2 * x + 1] < hOrg - 1)) oobBottom = true;
            }
            if ((oobLeft || oobRight) && (oobTop || oobBottom)) { if ((maxX - minX) > (maxY - minY)) oobBottom = oobTop = false; else oobLeft = oobRight = false; }
            if (oobLeft) minX *= 0.995;
            if (oobRight) maxX *= 0.995;
            if (oobTop) minY *= 0.995;
            if (oobBottom) maxY *= 0.995;
            if (++iteration > 500) return NALO_IO_ERR_FORMAT;                                     // the reference exit(1)s here
        }
        R.K[0] = ((float)w - 1.0f) / (maxX - minX); R.K[1] = ((float)h - 1.0f) / (maxY - minY);
        R.K[2] = -minX * R.K[0]; R.K[3] = -minY * R.K[1];
    } else if (cam->rect_mode == -2) return NALO_IO_ERR_FORMAT;                                   // makeOptimalK_full: assert(false)
    else if (cam->rect_mode == -3) {
        if (w != wOrg || h != hOrg) return NALO_IO_ERR_FORMAT;
        for (int i = 0; i < 4; ++i) R.K[i] = cam->pars[i];
        pass = 1;
    } else {
        R.K[0] = cam->out_calib[0] * w; R.K[1] = cam->out_calib[1] * h; R.K[2] = cam->out_calib[2] * w - 0.5; R.K[3] = cam->out_calib[3] * h - 0.5;
    }
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) { remapX[x + y * w] = x; remapY[x + y * w] = y; }
    R.distort(remapX, remapY, remapX, remapY, h * w);
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {                                     // "make rounding resistant" (:972-996), the `ix = hOrg-1.001` slip included
        float ix = remapX[x + y * w], iy = remapY[x + y * w];
        if (ix == 0) ix = 0.001;
        if (iy == 0) iy = 0.001;
        if (ix == wOrg - 1) ix = wOrg - 1.001;
        if (iy == hOrg - 1) ix = hOrg - 1.001;
        // DEVIATION from the reference: its test is `iy < wOrg-1` (util/Undistort.cpp:980). On a landscape sensor an entry with hOrg-1 <= iy < wOrg-1 stays
        // "valid" there and Undistort::undistort then reads rows behind the image (:497-512: an out-of-bounds read). Such entries are outside here (-1 -> pixel 0),
        // which is also what nalo_undist_set demands of a table; everything the reference defines is unchanged (tests/test_io_cpu.py, tests/test_ingest_gpu.py).
        if (ix > 0 && iy > 0 && ix < wOrg - 1 && iy < wOrg - 1 && iy < hOrg - 1) { remapX[x + y * w] = ix; remapY[x + y * w] = iy; }
        else { remapX[x + y * w] = -1; remapY[x + y * w] = -1; }
    }
    for (int i = 0; i < 4; ++i) K_out[i] = R.K[i];
    if (passthrough) *passthrough = pass;
    return NALO_IO_OK;
}

}  // extern "C"
