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
    // as build_strips (round 5): per tile row (tile column for a flat direction) the tiles under the strip, the strip clipped to each tile's own
    // box; the tile whose box reaches farthest towards either end of the strip is then looked at row by row in the validity bitmask itself: its
    // valid pixels of the strip give that end's bound, unless another tile's box reaches farther than they do (then that box's reach does)
    static uint32_t bits[H][W/32];
    for(int y=0;y<H;y++)for(int x=0;x<W;x++) if(m[y][x]) bits[y][x>>5]|=1u<<(x&31);
    long row_tests=0, box_tests=0;
    for(int d=0;d<16;d++){ float n=sqrtf(dx[d]*dx[d]+dy[d]*dy[d]); float ux=dx[d]/n, uy=dy[d]/n;
        const int steep=fabsf(uy)>=fabsf(ux); const int nn=steep?H/TH:W/TW, mm=steep?W/TW:H/TH;
        const float ua=steep?ux:uy, ub=steep?uy:ux, inv=1.0f/ub; const int sa=steep?TW:TH, sb=steep?TH:TW;
        for(int b=0;b<NB;b++){ float c0=(float)(b-OFF)-M, c1=(float)(b-OFF)+1.0f+M; float l=1e9f,h=-1e9f;
            for(int pass=0;pass<2;pass++){
                float best=pass==0?1e9f:-1e9f, second=best; int bty=-1,btx=-1;
                for(int i=0;i<nn;i++){ const float b0=i*sb, b1=i*sb+sb-1;
                    float v[4]={steep?(ua*b0-c0)*inv:(c0+ua*b0)*inv, steep?(ua*b0-c1)*inv:(c1+ua*b0)*inv, steep?(ua*b1-c0)*inv:(c0+ua*b1)*inv, steep?(ua*b1-c1)*inv:(c1+ua*b1)*inv};
                    float a0=v[0],a1=v[0]; for(int k=1;k<4;k++){ if(v[k]<a0)a0=v[k]; if(v[k]>a1)a1=v[k]; } a0-=0.01f; a1+=0.01f;
                    int j0=(int)floorf(a0/sa), j1=(int)floorf(a1/sa); if(j0<0)j0=0; if(j1>=mm)j1=mm-1;
                    for(int j=j0;j<=j1;j++){ const int ty=steep?i:j, tx=steep?j:i; int*q=bb[ty][tx]; box_tests++; if(q[2]<0) continue;
                        const float q0=steep?q[1]:q[0], q1=steep?q[3]:q[2];
                        float z[4]={steep?(ua*q0-c0)*inv:(c0+ua*q0)*inv, steep?(ua*q0-c1)*inv:(c1+ua*q0)*inv, steep?(ua*q1-c0)*inv:(c0+ua*q1)*inv, steep?(ua*q1-c1)*inv:(c1+ua*q1)*inv};
                        float p0=z[0],p1=z[0]; for(int k=1;k<4;k++){ if(z[k]<p0)p0=z[k]; if(z[k]>p1)p1=z[k]; } p0-=0.01f; p1+=0.01f;
                        p0=fmaxf(p0,(float)(steep?q[0]:q[1])); p1=fminf(p1,(float)(steep?q[2]:q[3])); if(p0>p1) continue;
                        const float t4[4]={ua*p0+ub*q0,ua*p0+ub*q1,ua*p1+ub*q0,ua*p1+ub*q1}; float tmin=t4[0],tmax=t4[0]; for(int k=1;k<4;k++){ if(t4[k]<tmin)tmin=t4[k]; if(t4[k]>tmax)tmax=t4[k]; }
                        const float key=pass==0?tmin:tmax;
                        if(pass==0 ? key<best : key>best){ second=best; best=key; bty=ty; btx=tx; } else if(pass==0 ? key<second : key>second) second=key; } }
                float exact=pass==0?1e9f:-1e9f;
                if(bty>=0){ int*q=bb[bty][btx]; const int tx=btx;
                    for(int y=q[1];y<=q[3];y++){ row_tests++;
                        float xa,xb;
                        if(fabsf(uy)>=1e-6f){ const float e0=(ux*y-c0)/uy, e1=(ux*y-c1)/uy; xa=fminf(e0,e1)-0.01f; xb=fmaxf(e0,e1)+0.01f; }
                        else { const float c=ux*y; if(c<c0-0.01f||c>c1+0.01f) continue; xa=-1e9f; xb=1e9f; }
                        int xl=(int)ceilf(fmaxf(xa,(float)q[0])), xr=(int)floorf(fminf(xb,(float)q[2])); if(xl>xr) continue;
                        const int x0=tx*TW; uint32_t wbits=bits[y][tx]&(0xFFFFFFFFu<<(xl-x0))&(0xFFFFFFFFu>>(31-(xr-x0))); if(!wbits) continue;
                        const int xf=x0+__builtin_ctz(wbits), xt=x0+31-__builtin_clz(wbits);
                        const float ta=ux*xf+uy*y, tb=ux*xt+uy*y;
                        if(pass==0){ if(ta<exact)exact=ta; if(tb<exact)exact=tb; } else { if(ta>exact)exact=ta; if(tb>exact)exact=tb; } } }
                if(pass==0) l=fminf(exact,second); else h=fmaxf(exact,second); }
            lo[d][b]=l; hi[d][b]=h; } }
    fprintf(stderr,"box tests per bin %.1f, row tests per bin %.1f\n",(double)box_tests/(16.0*NB),(double)row_tests/(16.0*NB));
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
