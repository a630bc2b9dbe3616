/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h). parity unpinned.
 * Restates DenseMapping::makeMap pixel loop (FullSystem/MapPoint.cpp:334-407) and the per-cluster
 * bounding-box scan of DenseMapping::updateMap (MapPoint.cpp:300-310). The plane (pi1..pi4) is an INPUT:
 * it comes from PCL RANSAC in the reference (MapPoint.cpp:522-613), third-party and unseeded. */
#include "orc_common.h"
#include <limits.h>
#include <float.h>

/* MapPoint.cpp:287-310: rect = {minx,maxx,miny,maxy} over 2<=x<w-2, 2<=y<h-2 with mask==value */
void orc_dense_bbox(const float* mask, int w, int h, float value, int rect[4]) {
    int minx=INT_MAX, miny=INT_MAX, maxx=INT_MIN, maxy=INT_MIN;
    for (int x=2;x<w-2;x++) for (int y=2;y<h-2;y++) {
        if (mask[x+y*w]!=value) continue;
        if (x>maxx) maxx=x; if (x<minx) minx=x; if (y>maxy) maxy=y; if (y<miny) miny=y;
    }
    rect[0]=minx; rect[1]=maxx; rect[2]=miny; rect[3]=maxy;
}
/* MapPoint.cpp:362-404. dI: level-0 AoS{I,dx,dy}; bgr: 3 bytes/pixel; camToWorld: 3x4 fp64.
 * out: u,v (int pixel), idepth, color, bgr; returns count; *accept = extent test (:403) incl. the
 * reference's bbox typos (`if(mP[1]>miny) maxy=..`, maxima seeded with FLT_MIN). */
int orc_dense_make_map(const float* mask, const float* dI, const uint8_t* bgr, int w, int h,
                       const float plane[4], float pcolor, const int rect[4],
                       float fxi, float fyi, float cx, float cy, const double camToWorld[12],
                       int* out_u, int* out_v, float* out_idepth, float* out_color, uint8_t* out_bgr, int* accept) {
    *accept=0;
    if (pcolor==0) return 0;
    float Ki[9]={fxi,0,-cx*fxi, 0,fyi,-cy*fyi, 0,0,1};   /* Ki[0] = K^-1 of the pinhole K (DenseMapping::makeK, MapPoint.cpp:409-443) */
    float minx=FLT_MAX, maxx=FLT_MIN, miny=FLT_MAX, maxy=FLT_MIN, minz=FLT_MAX, maxz=FLT_MIN;
    int n=0;
    for (int i=rect[2]; i<rect[3]; i++) for (int j=rect[0]; j<rect[1]; j++) {
        float color=mask[j+i*w];
        if (color!=pcolor) continue;
        if (i%3==0 || j%3==0) {
            float ddepth = plane[0]*(j*fxi-cx*fxi) + plane[1]*(i*fyi-cy*fyi) + plane[2];
            if (ddepth==0) continue;
            float depth = -plane[3]/ddepth;
            if (depth==0) continue;
            float idepth = 1/depth;
            out_u[n]=j; out_v[n]=i; out_idepth[n]=idepth; out_color[n]=dI[3*(j+i*w)];
            out_bgr[3*n]=bgr[3*(j+i*w)]; out_bgr[3*n+1]=bgr[3*(j+i*w)+1]; out_bgr[3*n+2]=bgr[3*(j+i*w)+2];
            n++;
            float cP[3]={Ki[0]*j+Ki[1]*i+Ki[2], Ki[3]*j+Ki[4]*i+Ki[5], Ki[6]*j+Ki[7]*i+Ki[8]};
            cP[0]/=idepth; cP[1]/=idepth; cP[2]/=idepth;
            double mP[3];
            for (int r=0;r<3;r++) mP[r]=camToWorld[r*4]*(double)cP[0]+camToWorld[r*4+1]*(double)cP[1]+camToWorld[r*4+2]*(double)cP[2]+camToWorld[r*4+3];
            if (mP[0]<minx) minx=mP[0]; if (mP[0]>maxx) maxx=mP[0];
            if (mP[1]<miny) miny=mP[1]; if (mP[1]>miny) maxy=mP[1];
            if (mP[2]<minz) minz=mP[2]; if (mP[2]>minz) maxz=mP[2];
        }
    }
    *accept = (maxx-minx<30 && maxy-miny<30 && maxz-minz<30);
    return n;
}
