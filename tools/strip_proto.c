// Prototype of the strip tables of the hole fill (ken-burns-effect_amd/csrc/kbe_holes.hip, build_strips): how many
// (hole, direction) pairs pass the test, and is it conservative -- does any direction that completes in a brute-force
// walk get skipped?  (dev aid)   Input: the validity mask of a 1024 x 1024 frame, one byte per pixel (existing > 0).
//   gcc -O2 -ffp-contract=off -o /tmp/strip_proto tools/strip_proto.c -lm && /tmp/strip_proto mask.u8
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#define W 1024
#define H 1024
#define TH 16
#define TW 32
static uint8_t m[H][W];
int main(int argc,char**argv){
    FILE*f=fopen(argv[1],"rb"); fread(m,1,W*H,f); fclose(f);
    const float dx[16]={-1,0,1,1,-1,1,2,2,-2,-1,1,2,3,3,3,3}, dy[16]={1,1,1,0,2,2,1,-1,3,3,3,3,2,1,-1,-2};
    // the box of every tile's valid pixels (what the tile launch leaves in its bbox table)
    static int bb[H/TH][W/TW][4];
    for(int ty=0;ty<H/TH;ty++)for(int tx=0;tx<W/TW;tx++){bb[ty][tx][0]=1<<20;bb[ty][tx][1]=1<<20;bb[ty][tx][2]=-1;bb[ty][tx][3]=-1;}
    int bx0=W,bx1=-1,by0=H,by1=-1;
    for(int y=0;y<H;y++)for(int x=0;x<W;x++) if(m[y][x]){ int*q=bb[y/TH][x/TW]; if(x<q[0])q[0]=x; if(y<q[1])q[1]=y; if(x>q[2])q[2]=x; if(y>q[3])q[3]=y;
        if(x<bx0)bx0=x; if(x>bx1)bx1=x; if(y<by0)by0=y; if(y>by1)by1=y; }
    const int NB=2*(W+H)+8, OFF=W+H+4; const float M=1.0f;
    static float lo[16][2*(W+H)+8], hi[16][2*(W+H)+8];
    // as build_strips (round 5): per tile row (tile column for a flat direction) the tiles under the strip, the strip clipped to each tile's own box
    for(int d=0;d<16;d++){ float n=sqrtf(dx[d]*dx[d]+dy[d]*dy[d]); float ux=dx[d]/n, uy=dy[d]/n;
        for(int b=0;b<NB;b++){ float c0=(float)(b-OFF)-M, c1=(float)(b-OFF)+1.0f+M; float l=1e9f,h=-1e9f;
            if(fabsf(uy)>=fabsf(ux)){ for(int ty=0;ty<H/TH;ty++){ float ya=ty*TH, yb=ty*TH+TH-1;
                    float xs[4]={(ux*ya-c0)/uy,(ux*ya-c1)/uy,(ux*yb-c0)/uy,(ux*yb-c1)/uy}; float xa=xs[0],xb=xs[0]; for(int k=1;k<4;k++){ if(xs[k]<xa)xa=xs[k]; if(xs[k]>xb)xb=xs[k]; }
                    int t0=(int)floorf((xa-0.01f)/TW), t1=(int)floorf((xb+0.01f)/TW); if(t0<0)t0=0; if(t1>=W/TW)t1=W/TW-1;
                    for(int tx=t0;tx<=t1;tx++){ int*q=bb[ty][tx]; if(q[2]<0) continue; float y0=q[1], y1=q[3];
                        float zs[4]={(ux*y0-c0)/uy,(ux*y0-c1)/uy,(ux*y1-c0)/uy,(ux*y1-c1)/uy}; float za=zs[0],zb=zs[0]; for(int k=1;k<4;k++){ if(zs[k]<za)za=zs[k]; if(zs[k]>zb)zb=zs[k]; }
                        float ia=fmaxf(za-0.01f,(float)q[0]), ib=fminf(zb+0.01f,(float)q[2]); if(ia>ib) continue;
                        float ts[4]={ux*ia+uy*y0,ux*ia+uy*y1,ux*ib+uy*y0,ux*ib+uy*y1}; for(int k=0;k<4;k++){ if(ts[k]<l)l=ts[k]; if(ts[k]>h)h=ts[k]; } } }
            } else { for(int tx=0;tx<W/TW;tx++){ float xa=tx*TW, xb=tx*TW+TW-1;
                    float ys[4]={(c0+uy*xa)/ux,(c1+uy*xa)/ux,(c0+uy*xb)/ux,(c1+uy*xb)/ux}; float ya=ys[0],yb=ys[0]; for(int k=1;k<4;k++){ if(ys[k]<ya)ya=ys[k]; if(ys[k]>yb)yb=ys[k]; }
                    int t0=(int)floorf((ya-0.01f)/TH), t1=(int)floorf((yb+0.01f)/TH); if(t0<0)t0=0; if(t1>=H/TH)t1=H/TH-1;
                    for(int ty=t0;ty<=t1;ty++){ int*q=bb[ty][tx]; if(q[2]<0) continue; float x0=q[0], x1=q[2];
                        float zs[4]={(c0+uy*x0)/ux,(c1+uy*x0)/ux,(c0+uy*x1)/ux,(c1+uy*x1)/ux}; float za=zs[0],zb=zs[0]; for(int k=1;k<4;k++){ if(zs[k]<za)za=zs[k]; if(zs[k]>zb)zb=zs[k]; }
                        float ia=fmaxf(za-0.01f,(float)q[1]), ib=fminf(zb+0.01f,(float)q[3]); if(ia>ib) continue;
                        float ts[4]={ux*x0+uy*ia,ux*x0+uy*ib,ux*x1+uy*ia,ux*x1+uy*ib}; for(int k=0;k<4;k++){ if(ts[k]<l)l=ts[k]; if(ts[k]>h)h=ts[k]; } } } }
            lo[d][b]=l; hi[d][b]=h; } }
    long steps_loop=0; long holes=0, pairs=0, complete=0, survive=0, falsekill=0, steps_all=0, steps_surv=0, holes_any=0;
    for(int y=by0;y<=by1;y++)for(int x=bx0;x<=bx1;x++){ if(m[y][x]) continue; holes++; int any=0;
        for(int d=0;d<16;d++){ float n=sqrtf(dx[d]*dx[d]+dy[d]*dy[d]); volatile float ux=dx[d]/n, uy=dy[d]/n; pairs++;
            // truth
            volatile float fx=x,fy=y; int okA=0,okB=0; long st=0;
            for(;;){ fx-=ux; fy-=uy; int ix=(int)roundf(fx),iy=(int)roundf(fy); st++; if(ix<0||ix>=W||iy<0||iy>=H) break; if(m[iy][ix]){okA=1;break;} }
            fx=x;fy=y;
            for(;;){ fx+=ux; fy+=uy; int ix=(int)roundf(fx),iy=(int)roundf(fy); st++; if(ix<0||ix>=W||iy<0||iy>=H) break; if(m[iy][ix]){okB=1;break;} }
            steps_all+=st; int comp=okA&&okB; complete+=comp;
            float c=-uy*x+ux*y, t=ux*x+uy*y; int b=(int)floorf(c)+OFF; int surv = !(lo[d][b] > t+1.0f || hi[d][b] < t-1.0f);
            survive+=surv; if(surv){steps_surv+=st; any=1;
                // steps with the in-loop test: an end stops once it is past everything on its line
                volatile float gx=x,gy=y; long s2=0; int dead=0;
                for(;;){ gx-=ux; gy-=uy; int ix=(int)roundf(gx),iy=(int)roundf(gy); s2++; if(ix<0||ix>=W||iy<0||iy>=H){dead=1;break;} if(m[iy][ix])break; if(lo[d][b] > (ux*ix+uy*iy)+1.0f){dead=1;break;} }
                if(!dead){ gx=x;gy=y; for(;;){ gx+=ux; gy+=uy; int ix=(int)roundf(gx),iy=(int)roundf(gy); s2++; if(ix<0||ix>=W||iy<0||iy>=H)break; if(m[iy][ix])break; if(hi[d][b] < (ux*ix+uy*iy)-1.0f)break; } }
                steps_loop+=s2; } if(comp&&!surv){ if(falsekill<6) printf("FK x=%d y=%d d=%d c=%.3f t=%.3f b=%d lo=%.3f hi=%.3f\n",x,y,d,c,t,b,lo[d][b],hi[d][b]); falsekill++;} }
        holes_any+=any; }
    printf("steps with in-loop test and A-first order: %.1f/hole\n",(double)steps_loop/holes); printf("holes in box %ld, pairs %ld, complete %ld (%.2f/hole), survive strip test %ld (%.2f/hole), false kills %ld; holes with any survivor %ld; steps all %.1f/hole, of survivors %.1f/hole\n",
        holes,pairs,complete,(double)complete/holes,survive,(double)survive/holes,falsekill,holes_any,(double)steps_all/holes,(double)steps_surv/holes);
    return 0; }
