/* logf_check.c — TEST INFRASTRUCTURE.  The device-side .ply row encoder (mesh2splat_amd/csrc/m2s_export.hip) must produce
 * the bytes the reference's writers produce with std::log(float), i.e. with the C library's logf, which is not correctly
 * rounded.  This program restates glibc's logf (Arm optimized-routines algorithm; constants read out of libm.so.6's
 * __logf_data) exactly as the device code does and compares it with the host's logf over positive finite floats:
 *     logf_check [stride]      (stride 1 = all 2 139 095 039 of them, ~20 CPU-seconds)
 * prints the number of mismatches without and with fused multiply-adds; exit status 0 iff both are zero. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <omp.h>
static const double T[16][2] = {
{0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2},{0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},{0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2},{0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
{0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3},{0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},{0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4},{0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
{0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},{0x1p+0, 0x0p+0},{0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},{0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
{0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},{0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},{0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},{0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
static const double Ln2 = 0x1.62e42fefa39efp-1, A0=-0x1.00ea348b88334p-2, A1=0x1.5575b0be00b6ap-2, A2=-0x1.ffffef20a4123p-2;
static inline uint32_t asu(float f){uint32_t u;memcpy(&u,&f,4);return u;} static inline float asf(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static float mylog(float x, int usefma){
  uint32_t ix=asu(x);
  if(ix==0x3f800000) return 0;
  if(ix-0x00800000u >= 0x7f800000u-0x00800000u){
    if(ix*2==0) return -INFINITY; if(ix==0x7f800000) return x; if((ix&0x80000000u)|| ix*2>=0xff000000u) return NAN;
    ix=asu(x*0x1p23f); ix-=23u<<23; }
  uint32_t tmp=ix-0x3f330000u; int i=(tmp>>19)%16; int k=(int32_t)tmp>>23; uint32_t iz=ix-(tmp&0xff800000u);
  double invc=T[i][0], logc=T[i][1], z=(double)asf(iz);
  double r,y0,r2,y;
  if(usefma){ r=fma(z,invc,-1.0); y0=fma((double)k,Ln2,logc); r2=r*r; y=fma(A1,r,A2); y=fma(A0,r2,y); y=fma(y,r2,y0+r);} 
  else { r=z*invc-1; y0=logc+(double)k*Ln2; r2=r*r; y=A1*r+A2; y=A0*r2+y; y=y*r2+(y0+r);} 
  return (float)y;
}
int main(int argc,char**argv){ long stride = argc>1 ? atol(argv[1]) : 1; if(stride<1) stride=1;
  long bad0=0,bad1=0; 
  #pragma omp parallel for reduction(+:bad0,bad1) schedule(static)
  for(long u=1; u<0x7f800000L; u+=stride){ float x=asf((uint32_t)u); float ref=logf(x); 
     if(asu(mylog(x,0))!=asu(ref)) bad0++; if(asu(mylog(x,1))!=asu(ref)) bad1++; }
  printf("mismatch nofma=%ld fma=%ld stride=%ld\n",bad0,bad1,stride);
  return (bad0||bad1)?1:0; }
