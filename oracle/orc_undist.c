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

/* ---- rectification: Undistort::readFromFile after the parse (util/Undistort.cpp:911-1005), makeOptimalK_crop (:637-757), distortCoordinates of the five models
 * (:1018-1292). model: 0 Pinhole, 1 RadTan, 2 FOV, 3 EquiDistant, 4 KannalaBrandt. rect_mode: -1 crop, -3 none, 0 explicit (out_calib). Returns 0, or -3 where the
 * reference exits / asserts. K: fx fy cx cy (doubles, Mat33 K). */
static void undist_distort(int model, const float* p, const double* K, const float* in_x, const float* in_y, float* out_x, float* out_y, int n) {
    float fx=p[0], fy=p[1], cx=p[2], cy=p[3];
    float ofx=K[0], ofy=K[1], ocx=K[2], ocy=K[3];
    for (int i=0;i<n;i++) {
        float x=in_x[i], y=in_y[i];
        float ix=(x-ocx)/ofx, iy=(y-ocy)/ofy;
        if (model==2) { float dist=p[4]; float d2t=2.0f*tan(dist/2.0f); float r=sqrtf(ix*ix+iy*iy); float fac=(r==0||dist==0)?1:atanf(r*d2t)/(dist*r);
            ix=fx*fac*ix+cx; iy=fy*fac*iy+cy; out_x[i]=ix; out_y[i]=iy; }
        else if (model==1) { float k1=p[4],k2=p[5],r1=p[6],r2=p[7]; float mx2_u=ix*ix, my2_u=iy*iy, mxy_u=ix*iy, rho2_u=mx2_u+my2_u; float rad_dist_u=k1*rho2_u+k2*rho2_u*rho2_u;
            float x_dist=ix+ix*rad_dist_u+2.0*r1*mxy_u+r2*(rho2_u+2.0*mx2_u); float y_dist=iy+iy*rad_dist_u+2.0*r2*mxy_u+r1*(rho2_u+2.0*my2_u);
            out_x[i]=fx*x_dist+cx; out_y[i]=fy*y_dist+cy; }
        else if (model==3) { float k1=p[4],k2=p[5],k3=p[6],k4=p[7]; float r=sqrt(ix*ix+iy*iy); float theta=atan(r); float theta2=theta*theta, theta4=theta2*theta2, theta6=theta4*theta2, theta8=theta4*theta4;
            float thetad=theta*(1+k1*theta2+k2*theta4+k3*theta6+k4*theta8); float scaling=(r>1e-8)?thetad/r:1.0; out_x[i]=fx*ix*scaling+cx; out_y[i]=fy*iy*scaling+cy; }
        else if (model==4) { float k0=p[4],k1=p[5],k2=p[6],k3=p[7]; float Xsq_plus_Ysq=ix*ix+iy*iy, sqrt_Xsq_Ysq=sqrtf(Xsq_plus_Ysq); float theta=atan2f(sqrt_Xsq_Ysq,1);
            float theta2=theta*theta, theta3=theta2*theta, theta5=theta3*theta2, theta7=theta5*theta2, theta9=theta7*theta2; float r=theta+k0*theta3+k1*theta5+k2*theta7+k3*theta9;
            if (sqrt_Xsq_Ysq<1e-6) { out_x[i]=fx*ix+cx; out_y[i]=fy*iy+cy; } else { out_x[i]=(r/sqrt_Xsq_Ysq)*fx*ix+cx; out_y[i]=(r/sqrt_Xsq_Ysq)*fy*iy+cy; } }
        else { ix=fx*ix+cx; iy=fy*iy+cy; out_x[i]=ix; out_y[i]=iy; }
    }
}
int orc_make_rectification(int model, const double* pars8, int wOrg, int hOrg, int w, int h, int rect_mode, const float* out_calib, double* K, float* remapX, float* remapY, int* passthrough) {
    float p[8]; for (int i=0;i<8;i++) p[i]=(float)pars8[i];
    *passthrough=0;
    if (rect_mode==-1) {
        K[0]=K[1]=1; K[2]=K[3]=0;
        float* tgX=(float*)malloc(4*100000); float* tgY=(float*)malloc(4*100000);
        float minX=0,maxX=0,minY=0,maxY=0;
        for (int x=0;x<100000;x++) { tgX[x]=(x-50000.0f)/10000.0f; tgY[x]=0; }
        undist_distort(model,p,K,tgX,tgY,tgX,tgY,100000);
        for (int x=0;x<100000;x++) if (tgX[x]>0 && tgX[x]<wOrg-1) { if (minX==0) minX=(x-50000.0f)/10000.0f; maxX=(x-50000.0f)/10000.0f; }
        for (int y=0;y<100000;y++) { tgY[y]=(y-50000.0f)/10000.0f; tgX[y]=0; }
        undist_distort(model,p,K,tgX,tgY,tgX,tgY,100000);
        for (int y=0;y<100000;y++) if (tgY[y]>0 && tgY[y]<hOrg-1) { if (minY==0) minY=(y-50000.0f)/10000.0f; maxY=(y-50000.0f)/10000.0f; }
        free(tgX); free(tgY);
        minX*=1.01; maxX*=1.01; minY*=1.01; maxY*=1.01;
        int oobLeft=1,oobRight=1,oobTop=1,oobBottom=1, iteration=0;
        while (oobLeft||oobRight||oobTop||oobBottom) {
            oobLeft=oobRight=oobTop=oobBottom=0;
            for (int y=0;y<h;y++) { remapX[y*2]=minX; remapX[y*2+1]=maxX; remapY[y*2]=remapY[y*2+1]=minY+(maxY-minY)*(float)y/((float)h-1.0f); }
            undist_distort(model,p,K,remapX,remapY,remapX,remapY,2*h);
            for (int y=0;y<h;y++) { if (!(remapX[2*y]>0 && remapX[2*y]<wOrg-1)) oobLeft=1; if (!(remapX[2*y+1]>0 && remapX[2*y+1]<wOrg-1)) oobRight=1; }
            for (int x=0;x<w;x++) { remapY[x*2]=minY; remapY[x*2+1]=maxY; remapX[x*2]=remapX[x*2+1]=minX+(maxX-minX)*(float)x/((float)w-1.0f); }
            undist_distort(model,p,K,remapX,remapY,remapX,remapY,2*w);
            for (int x=0;x<w;x++) { if (!(remapY[2*x]>0 && remapY[2*x]<hOrg-1)) oobTop=1; if (!(remapY[2*x+1]>0 && remapY[2*x+1]<hOrg-1)) oobBottom=1; }
            if ((oobLeft||oobRight)&&(oobTop||oobBottom)) { if ((maxX-minX)>(maxY-minY)) oobBottom=oobTop=0; else oobLeft=oobRight=0; }
            if (oobLeft) minX*=0.995; if (oobRight) maxX*=0.995; if (oobTop) minY*=0.995; if (oobBottom) maxY*=0.995;
            iteration++;
            if (iteration>500) return -3;
        }
        K[0]=((float)w-1.0f)/(maxX-minX); K[1]=((float)h-1.0f)/(maxY-minY); K[2]=-minX*K[0]; K[3]=-minY*K[1];
    } else if (rect_mode==-2) return -3;
    else if (rect_mode==-3) { if (w!=wOrg||h!=hOrg) return -3; for (int i=0;i<4;i++) K[i]=pars8[i]; *passthrough=1; }
    else { K[0]=out_calib[0]*w; K[1]=out_calib[1]*h; K[2]=out_calib[2]*w-0.5; K[3]=out_calib[3]*h-0.5; }
    for (int y=0;y<h;y++) for (int x=0;x<w;x++) { remapX[x+y*w]=x; remapY[x+y*w]=y; }
    undist_distort(model,p,K,remapX,remapY,remapX,remapY,h*w);
    for (int y=0;y<h;y++) for (int x=0;x<w;x++) {
        float ix=remapX[x+y*w], iy=remapY[x+y*w];
        if (ix==0) ix=0.001; if (iy==0) iy=0.001;
        if (ix==wOrg-1) ix=wOrg-1.001;
        if (iy==hOrg-1) ix=hOrg-1.001;
        if (ix>0 && iy>0 && ix<wOrg-1 && iy<wOrg-1) { remapX[x+y*w]=ix; remapY[x+y*w]=iy; } else { remapX[x+y*w]=-1; remapY[x+y*w]=-1; }
    }
    return 0;
}
