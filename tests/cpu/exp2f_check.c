#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
static inline uint64_t asu64(double d){uint64_t u;memcpy(&u,&d,8);return u;}
static inline double asd(uint64_t u){double d;memcpy(&d,&u,8);return d;}
static uint64_t T[32];
static float my_exp2f(float x, int use_fma){
  if (x <= -150.0f) return 0.0f;
  const double C0=0x1.c6af84b912394p-5, C1=0x1.ebfce50fac4f3p-3, C2=0x1.62e42ff0c52d6p-1;
  const double SHIFT = 0x1.8p+52/32;
  double xd=(double)x; double kd=xd+SHIFT; uint64_t ki=asu64(kd); kd-=SHIFT; double r=xd-kd;
  uint64_t t=T[ki%32]; t+= ki<<(52-5); double s=asd(t);
  double z,r2,y;
  if(use_fma){ z=fma(C0,r,C1); r2=r*r; y=fma(C2,r,1.0); y=fma(z,r2,y); y=y*s; }
  else { z=C0*r+C1; r2=r*r; y=C2*r+1; y=z*r2+y; y=y*s; }
  return (float)y;
}
int main(){
  for(int i=0;i<32;i++){ double v=(double)exp2l((long double)i/32.0L); T[i]=asu64(v)-((uint64_t)i<<47); }
  srand(1); long bad0=0,bad1=0,n=20000000;
  for(long i=0;i<n;i++){ float x=-(float)(rand()/(double)RAND_MAX*160.0); if(i%3==0) x=-(float)(rand()/(double)RAND_MAX*8.0);
    volatile float xv=x; float e=exp2f(xv); float a=my_exp2f(x,0), b=my_exp2f(x,1);
    if(memcmp(&e,&a,4)) bad0++; if(memcmp(&e,&b,4)) bad1++; }
  printf("n=%ld mismatches nofma=%ld fma=%ld\n",n,bad0,bad1);
}
