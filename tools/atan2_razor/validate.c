#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "tab.h"
static uint64_t s=88172645463325252ull;
static inline uint64_t rnd(){ s^=s<<13; s^=s>>7; s^=s<<17; return s; }
static inline double u01(){ return (rnd()>>11)*(1.0/9007199254740992.0); }
/* atan2 for (y, x) whose bearing is within ~1e-9 degrees of k degrees, k = 0..180 on |y|: theta = phi_k + N/D with
   N = |y| cos - x sin (double-double: the products exactly, by FMA), D = x cos + |y| sin */
static double razor(double y, double x, int k){
  const double ay=fabs(y);
  const double ph=kDegDD[k][0], pl=kDegDD[k][1], ch=kDegDD[k][2], cl=kDegDD[k][3], sh=kDegDD[k][4], sl=kDegDD[k][5];
  /* N = ay*c - x*s */
  double p1=ay*ch, e1=fma(ay,ch,-p1);
  double p2=x*sh,  e2=fma(x,sh,-p2);
  double d=p1-p2;                      /* nearly cancels: exact when within a factor 2 (Sterbenz) */
  double bb=d-p1; double err=(p1-(d-bb))+(-p2-bb); /* two_sum(p1,-p2) error term */
  double lo=err+(e1-e2)+(ay*cl-x*sl);
  double N=d+lo;
  double D=x*ch+ay*sh;
  double delta=N/D;
  double r=ph+(pl+delta);
  return copysign(r,y);
}
int main(int argc,char**argv){
  long n=argc>1?atol(argv[1]):4000000, bad=0, tested=0, differs_fast=0;
  for(long i=0;i<n;i++){
    int k=(int)(rnd()%181);
    double rad=10.0+u01()*700.0;
    double th=(double)k*M_PI/180.0;
    double x=rad*cos(th), y=rad*sin(th)*((rnd()&1)?1:-1);
    /* perturb by a few ulps / tiny relative amounts, and sometimes snap to a coarse lattice */
    int m=rnd()%4;
    if(m==0){ x=nextafter(x,(rnd()&1)?1e9:-1e9); }
    else if(m==1){ y*= (1.0+ldexp((double)((int)(rnd()%17)-8),-52)); }
    else if(m==2){ x*=(1.0+ (u01()-0.5)*1e-12); }
    if(y==0.0) continue;
    double g=atan2(y,x);
    double deg=fabs(g)/M_PI*180.0;
    if(fabs(deg-rint(deg))>=1e-9) continue;
    int kk=(int)rint(deg);
    double r=razor(y,x,kk);
    tested++;
    if(memcmp(&g,&r,8)!=0){ if(bad<10) printf("k=%d y=%a x=%a glibc=%a razor=%a\n",kk,y,x,g,r); bad++; }
  }
  printf("tested %ld mismatches %ld\n",tested,bad);
  return bad!=0;
}
