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

}  // extern "C"
