/* CPU oracle -- TEST INFRASTRUCTURE ONLY (see orc_common.h). Parity unpinned: the reference ships no tests/fixtures for this path.
 *
 * the candidate-pixel selection of PixelSelector (FullSystem/PixelSelector2.cpp), restated as plain scalar C:
 *   orc_pixsel_select          PixelSelector::select            (:564-711)  three-level grid selection (pot, 2pot, 4pot cells)
 *   orc_pixsel_make_maps       PixelSelector::makeMaps          (:144-291)  select + potential adaptation (<= recursionsLeft re-selects) + random sub-selection
 *   orc_pixsel_fuse_mask       PixelSelector::FusedWithMask     (:431-560)  mask histogram / quantile and the rand()-driven status changes
 *   orc_pixsel_make_maps_lidar PixelSelector::makeMaps_lidar    (:293-428)  select + FusedWithMask (the adaptation code is commented out in the fork)
 *   orc_pixsel_libc_tables     PixelSelector ctor (:40-45) and FusedWithMask's srand(3141592)/rand() stream (:496-501): both are libc streams,
 *                              so they are INPUTS of the path (the caller owns them); this helper draws them from the libc the tests run on.
 * Scan order matters: the direction of every cell is directions[randomPattern[n2] & 15] with n2 = number of level-1 selections made so far.
 * thsSmoothed: the reference allocates (w/32)*(h/32)+100 floats and fills the first (w/32)*(h/32); select() indexes it with (x>>5) + (y>>5)*(w/32), which
 * for image sizes that are not multiples of 32 (KITTI's 1224x368 included) runs into the next row (x) or into the uninitialised tail (y >= 32*(h/32)).
 * The restatement keeps the indexing and DEFINES the tail as 0: callers pass nb + 100 floats with a zero tail. */
#include "orc_common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_pixsel_make_hists(const float* absg0, int w, int h, float* ths, float* thsSmoothed);

static const float kDirs[16][2] = {{0.f, 1.0000f},     {0.3827f, 0.9239f},  {0.1951f, 0.9808f},  {0.9239f, 0.3827f}, {0.7071f, 0.7071f},  {0.3827f, -0.9239f},
                                   {0.8315f, 0.5556f}, {0.8315f, -0.5556f}, {0.5556f, -0.8315f}, {0.9808f, 0.1951f}, {0.9239f, -0.3827f}, {0.7071f, -0.7071f},
                                   {0.5556f, 0.8315f}, {0.9808f, -0.1951f}, {1.0000f, 0.0000f},  {0.1951f, -0.9808f}};   /* PixelSelector2.cpp:581-597 */

void orc_pixsel_libc_tables(int n, unsigned char* randomPattern, int* draws) {
    srand(3141592);
    for (int i = 0; i < n; i++) randomPattern[i] = (unsigned char)(rand() & 0xFF);
    srand(3141592);
    for (int i = 0; i < n; i++) draws[i] = rand();
}

/* dI: [w*h][3] level-0 texels {I, dx, dy}; absg0/1/2: absSquaredGrad of levels 0..2 (w1 = w/2, w2 = w/4). map_out: 0 / 1 / 2 / 4 per pixel. */
void orc_pixsel_select(const float* dI, const float* absg0, const float* absg1, const float* absg2, int w, int h, const float* thsSmoothed,
                       const unsigned char* randomPattern, int pot, float thFactor, float* map_out, int* n_out) {
    const int w1 = w / 2, w2 = w / 4, thsStep = w / 32;
    const float dw1 = SETTING_GRAD_DOWNWEIGHT_PER_LEVEL, dw2 = dw1 * dw1;   /* settings.cpp:156 */
    memset(map_out, 0, sizeof(float) * (size_t)w * h);
    int n2 = 0, n3 = 0, n4 = 0;
    for (int y4 = 0; y4 < h; y4 += 4 * pot) for (int x4 = 0; x4 < w; x4 += 4 * pot) {
        const int my3 = (4 * pot < h - y4) ? 4 * pot : h - y4, mx3 = (4 * pot < w - x4) ? 4 * pot : w - x4;
        int best4 = -1; float val4 = 0;
        const float* dir4 = kDirs[randomPattern[n2] & 0xF];
        for (int y3 = 0; y3 < my3; y3 += 2 * pot) for (int x3 = 0; x3 < mx3; x3 += 2 * pot) {
            const int x34 = x3 + x4, y34 = y3 + y4;
            const int my2 = (2 * pot < h - y34) ? 2 * pot : h - y34, mx2 = (2 * pot < w - x34) ? 2 * pot : w - x34;
            int best3 = -1; float val3 = 0;
            const float* dir3 = kDirs[randomPattern[n2] & 0xF];
            for (int y2 = 0; y2 < my2; y2 += pot) for (int x2 = 0; x2 < mx2; x2 += pot) {
                const int x234 = x2 + x34, y234 = y2 + y34;
                const int my1 = (pot < h - y234) ? pot : h - y234, mx1 = (pot < w - x234) ? pot : w - x234;
                int best2 = -1; float val2 = 0;
                const float* dir2 = kDirs[randomPattern[n2] & 0xF];
                for (int y1 = 0; y1 < my1; y1++) for (int x1 = 0; x1 < mx1; x1++) {
                    const int xf = x1 + x234, yf = y1 + y234, idx = xf + w * yf;
                    if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
                    const float th0 = thsSmoothed[(xf >> 5) + (yf >> 5) * thsStep], th1 = th0 * dw1, th2 = th1 * dw2;
                    const float gx = dI[3 * idx + 1], gy = dI[3 * idx + 2];
                    if (absg0[idx] > th0 * thFactor) {
                        const float dn = fabsf(gx * dir2[0] + gy * dir2[1]);
                        if (dn > val2) { val2 = dn; best2 = idx; best3 = -2; best4 = -2; }
                    }
                    if (best3 == -2) continue;
                    if (absg1[(int)(xf * 0.5f + 0.25f) + (int)(yf * 0.5f + 0.25f) * w1] > th1 * thFactor) {
                        const float dn = fabsf(gx * dir3[0] + gy * dir3[1]);
                        if (dn > val3) { val3 = dn; best3 = idx; best4 = -2; }
                    }
                    if (best4 == -2) continue;
                    if (absg2[(int)(xf * 0.25f + 0.125) + (int)(yf * 0.25f + 0.125) * w2] > th2 * thFactor) {
                        const float dn = fabsf(gx * dir4[0] + gy * dir4[1]);
                        if (dn > val4) { val4 = dn; best4 = idx; }
                    }
                }
                if (best2 > 0) { map_out[best2] = 1; val3 = 1e10f; n2++; }
            }
            if (best3 > 0) { map_out[best3] = 2; val4 = 1e10f; n3++; }
        }
        if (best4 > 0) { map_out[best4] = 4; n4++; }
    }
    n_out[0] = n2; n_out[1] = n3; n_out[2] = n4;
}

/* makeMaps: returns numHaveSub; *potential is PixelSelector::currentPotential (in/out). ths/thsSmoothed: scratch of (w/32)*(h/32) floats. */
int orc_pixsel_make_maps(const float* dI, const float* absg0, const float* absg1, const float* absg2, int w, int h, const unsigned char* randomPattern,
                         float density, int recursionsLeft, float thFactor, int* potential, float* map_out) {
    const int nb = (w / 32) * (h / 32), nbp = nb + 100;
    float* ths = (float*)calloc(2 * (size_t)nbp, sizeof(float));
    orc_pixsel_make_hists(absg0, w, h, ths, ths + nbp);
    int cur = *potential, ideal, n[3];
    float numHave, quotia;
    for (;;) {
        orc_pixsel_select(dI, absg0, absg1, absg2, w, h, ths + nbp, randomPattern, cur, thFactor, map_out, n);
        numHave = (float)(n[0] + n[1] + n[2]);
        quotia = density / numHave;
        const float K = numHave * (cur + 1) * (cur + 1);
        ideal = (int)(sqrtf(K / density) - 1);
        if (ideal < 1) ideal = 1;
        if (recursionsLeft > 0 && quotia > 1.25 && cur > 1) { if (ideal >= cur) ideal = cur - 1; cur = ideal; recursionsLeft--; continue; }
        if (recursionsLeft > 0 && quotia < 0.25) { if (ideal <= cur) ideal = cur + 1; cur = ideal; recursionsLeft--; continue; }
        break;
    }
    int numHaveSub = (int)numHave;
    if (quotia < 0.95) {                                       /* double literal, as in the reference */
        const unsigned char charTH = (unsigned char)(255 * quotia);
        int rn = 0;
        for (int i = 0; i < w * h; i++)
            if (map_out[i] != 0) { if (randomPattern[rn] > charTH) { map_out[i] = 0; numHaveSub--; } rn++; }
    }
    *potential = ideal;
    free(ths);
    return numHaveSub;
}

/* FusedWithMask. draws[i] = the i-th rand() after srand(3141592). mhist has 257 entries here: the reference reads mhist[256] (one past its
 * 256-int array) on the last quantile iteration; the restatement defines that entry as 0. Outputs n = {n1, n2, 0}, quantile, max_mas. */
void orc_pixsel_fuse_mask(const float* mask, const int* draws, int wh, float* map, int* n_out, int* qm_out) {
    int mhist[257]; memset(mhist, 0, sizeof(mhist));
    for (int i = 0; i < wh; i++) if (mask[i] != 0) { mhist[0]++; mhist[(int)mask[i]]++; }
    int th_ = (int)(mhist[0] * 0.5 + 0.5f), quantile = 255, max_mas = 0;
    for (int i = 0; i < 256; i++) { th_ -= mhist[i + 1]; if (th_ < 0) { quantile = i; break; } }
    for (int i = 255; i > 0; i--) { max_mas = i; if (mhist[i] != 0) break; }
    int n1 = 0, n2 = 0;
    for (int i = 0; i < wh; i++) {
        const float rs = draws[i] % 1000 / (float)(1000.0);
        if (map[i] == 1) { n1++; if (rs > 0.5 && mask[i] < quantile / 3) { map[i] = 2; n1--; n2++; } }
        else if (map[i] == 2) { n2++; if (rs < 0.6 && mask[i] > quantile + (max_mas - quantile) / 2) { n2--; n1++; map[i] = 1; } }
        else if (rs < 0.01 && mask[i] > quantile) { n1++; map[i] = 1; }
    }
    n_out[0] = n1; n_out[1] = n2; n_out[2] = 0;
    if (qm_out) { qm_out[0] = quantile; qm_out[1] = max_mas; }
}

int orc_pixsel_make_maps_lidar(const float* dI, const float* absg0, const float* absg1, const float* absg2, int w, int h, const unsigned char* randomPattern,
                               const float* mask, const int* draws, float thFactor, int potential, float* map_out) {
    const int nb = (w / 32) * (h / 32), nbp = nb + 100;
    float* ths = (float*)calloc(2 * (size_t)nbp, sizeof(float));
    orc_pixsel_make_hists(absg0, w, h, ths, ths + nbp);
    int n[3], m[3];
    orc_pixsel_select(dI, absg0, absg1, absg2, w, h, ths + nbp, randomPattern, potential, thFactor, map_out, n);
    orc_pixsel_fuse_mask(mask, draws, w * h, map_out, m, 0);
    free(ths);
    return m[0] + m[1] + m[2];
}
